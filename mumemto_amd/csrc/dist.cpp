// dist.cpp -- the multi-GPU exchange in C++ against RCCL (one process per GPU, xGMI underneath).
//
// What the reference does between partitions with files and a second tool (README.md:124-141: PREFIX.mums +
// PREFIX.athresh per partition, then `anchor_merge`, src/merge_candidates.cpp:170-255) happens here between ranks:
// every rank has run the single-GPU path on {anchor} + its share of the documents with merge metadata on; the row
// tables and thresholds of ranks 1 .. G-1 travel HBM -> HBM to rank 0 (ncclSend / ncclRecv, one group: the tables are
// ragged, an all-gather would pad them to the largest partition and leave copies nobody reads on every rank), rank 0
// folds them on its GPU (merge.cpp) and re-sorts into direct-run order.  For the modes without a partition merge (mmt_engine_set_scan_shard) the ranks'
// output bytes are gathered to rank 0 in rank order.
//
// RCCL is bound at run time: a process that already holds a copy (PyTorch ships its own librccl.so) must use that one,
// a plain C++ host gets /opt/rocm/lib/librccl.so.  Nothing here falls back to another transport.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "engine.hpp"
#include "merge.hpp"
#include "dist.hpp"

namespace mmt {

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle) return r;
    void* h = nullptr;
    // MUMEMTO_RCCL_LIB: this library and no other (the tests' transport double, tests/fake_rccl: ranks = processes that
    // share one GPU -- the only way the exchange below runs with more than one rank on a one-GPU box)
    if (const char* path = std::getenv("MUMEMTO_RCCL_LIB")) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) throw std::runtime_error(std::string("cannot load MUMEMTO_RCCL_LIB: ") + dlerror());
    }
    // a copy that is already in the process first (RTLD_NOLOAD), then the ROCm one
    if (!h)
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        if (!p) throw std::runtime_error(std::string("librccl lacks ") + n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.handle = h;
    return r;
}

void check(ncclResult_t e, const char* what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + what + ": " + rccl().GetErrorString(e));
}
#define MMT_NCCL(x) check((x), #x)

// One message of `count` elements as pieces of at most 2^29 BYTES (MUMEMTO_RCCL_CHUNK = elements per piece: tests) -- sender and
// receiver cut the same way, so the pieces pair up in order.  A rank's threshold column over a 3.05 Gbp anchor is 3.05 G elements
// / 12 GB: counts are size_t in the interface, but not every layer below it holds what the interface promises.  MEASURED in round 6
// (tests/micro/rccl_sizes.py, profiles/round6_rccl_piece_sizes.log; RCCL 2.26.6 of PyTorch 2.10 + ROCm 7.0, the rank as its own
// peer): a piece of 1 GiB arrives whole; of a piece of 2 GiB, 3.16 GB or 4 GiB -- whatever the element type -- HALF the elements
// arrive different, without an error.  Rounds 4 and 5 cut at 2^30 ELEMENTS (4 - 8 GiB a piece).  Half a gibibyte leaves a margin;
// two dozen pieces of a 12 GB column cost nothing inside a group.
static size_t rccl_chunk_elements(size_t width) {
    static const size_t env = [] { const char* e = std::getenv("MUMEMTO_RCCL_CHUNK"); return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0; }();
    if (env) return env;
    return ((size_t)1 << 29) / width;
}
template <typename T>
void send_pieces(const T* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    const size_t C = rccl_chunk_elements(sizeof(T));
    for (size_t at = 0; at < count || at == 0; at += C) {
        MMT_NCCL(rccl().Send(buf + at, std::min(C, count - at), t, peer, comm, st));
        if (count <= at + C) break;
    }
}
template <typename T>
void recv_pieces(T* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
    const size_t C = rccl_chunk_elements(sizeof(T));
    for (size_t at = 0; at < count || at == 0; at += C) {
        MMT_NCCL(rccl().Recv(buf + at, std::min(C, count - at), t, peer, comm, st));
        if (count <= at + C) break;
    }
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    Engine* engine = nullptr;
    DevBuf<uint64_t> d_meta;                 // 4 words per rank: rows, documents, output bytes, spare
    std::vector<std::unique_ptr<DevBuf<uint32_t>>> len;
    std::vector<std::unique_ptr<DevBuf<int64_t>>> off;
    std::vector<std::unique_ptr<DevBuf<uint8_t>>> st, text;
    std::vector<std::unique_ptr<DevBuf<uint32_t>>> th;      // thresholds travel at 32 bits
};

void comm_unique_id(uint8_t out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    ncclUniqueId id;
    MMT_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(out, &id, 128);
}

Comm* comm_create(Engine& e, int rank, int world, const uint8_t id_bytes[128]) {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("rank / world size out of range");
    MMT_HIP(hipSetDevice(e.device()));
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world; c->engine = &e;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, 128);
    MMT_NCCL(rccl().CommInitRank(&c->comm, world, id, rank));
    c->d_meta.ensure((size_t)world * 4);
    c->len.resize(world); c->off.resize(world); c->st.resize(world); c->th.resize(world); c->text.resize(world);
    for (int r = 0; r < world; r++) {
        c->len[r].reset(new DevBuf<uint32_t>()); c->off[r].reset(new DevBuf<int64_t>());
        c->st[r].reset(new DevBuf<uint8_t>()); c->th[r].reset(new DevBuf<uint32_t>()); c->text[r].reset(new DevBuf<uint8_t>());
    }
    return c.release();
}

void comm_destroy(Comm* c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
}

// every rank's four words on every rank
static std::vector<uint64_t> exchange_meta(Comm& c, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3 = 0) {
    hipStream_t st = c.engine->stream();
    uint64_t mine[4] = {w0, w1, w2, w3};
    DevBuf<uint64_t> d_mine;
    d_mine.ensure(4);
    MMT_HIP(hipMemcpyAsync(d_mine.get(), mine, 32, hipMemcpyHostToDevice, st));
    MMT_NCCL(rccl().AllGather(d_mine.get(), c.d_meta.get(), 4, ncclUint64, c.comm, st));
    std::vector<uint64_t> all((size_t)c.world * 4);
    MMT_HIP(hipMemcpyAsync(all.data(), c.d_meta.get(), all.size() * 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    return all;
}

int comm_world(const Comm& c) { return c.world; }

static MergedRows merge_on_rank0(Comm& c, uint32_t min_len, const std::vector<uint64_t>& meta);
static MergedRows merge_by_ranges(Comm& c, uint32_t min_len, const std::vector<uint64_t>& meta);

// Strict multi-MUMs.  The engine's last run must have been this rank's partition with merge metadata on.
// Returns the merged rows on rank 0 (already in direct-run order), an empty MergedRows elsewhere.
// route: 0 = rank 0 folds everything, 1 = every rank folds its slice of the anchor, -1 = automatic (slices from four ranks
// on; MUMEMTO_RANGE_FOLD=0/1 overrides).  The route is a collective decision: every rank sends its wish with the table
// sizes and all of them follow RANK 0's -- ranks whose environments differ must not end up in different collectives.
static MergedRows merge_routed(Comm& c, uint32_t min_len, bool* is_root, int route) {
    Engine& e = *c.engine;
    MMT_HIP(hipSetDevice(e.device()));
    if (is_root) *is_root = c.rank == 0;
    const HostRows& R = e.rows_meta();
    if (!R.mum_mode || !e.thresh_len()) throw std::runtime_error("the exchange needs a multi-MUM run with merge metadata");
    if (route < 0) {
        const char* env = std::getenv("MUMEMTO_RANGE_FOLD");
        route = env ? (std::string(env) == "1" ? 1 : 0) : (c.world >= 4 ? 1 : 0);
    }
    const uint32_t* my_len; const int64_t* my_off; const uint8_t* my_st;
    e.rows_mum_device(&my_len, &my_off, &my_st);
    const uint32_t my_longest = longest_row(e, my_len, R.n_rows, true);
    const std::vector<uint64_t> meta = exchange_meta(c, R.n_rows, R.n_docs, my_longest, (uint64_t)route);
    for (int r = 1; r < c.world; r++)
        if (meta[(size_t)r * 4 + 1] == 0) throw std::runtime_error("a rank without documents in the exchange");
    return meta[3] ? merge_by_ranges(c, min_len, meta) : merge_on_rank0(c, min_len, meta);
}

MergedRows dist_merge(Comm& c, uint32_t min_len, bool* is_root) { return merge_routed(c, min_len, is_root, -1); }
MergedRows dist_merge_ranges(Comm& c, uint32_t min_len, bool* is_root) { return merge_routed(c, min_len, is_root, 1); }

static MergedRows merge_on_rank0(Comm& c, uint32_t min_len, const std::vector<uint64_t>& meta) {
    Engine& e = *c.engine;
    hipStream_t st = e.stream();
    const HostRows& R = e.rows_meta();
    const uint64_t L = e.doc_len()[0] + 1;
    const uint32_t* my_len; const int64_t* my_off; const uint8_t* my_st;
    e.rows_mum_device(&my_len, &my_off, &my_st);
    // Only rank 0 folds: every other rank SENDS its four tables to rank 0 (point-to-point over xGMI, one group) and keeps
    // nothing of the others -- with a whole genome as the anchor a threshold column is 6 GB, and broadcasting every
    // rank's to every rank (round 2) put 8 x 6 GB into each rank's HBM for nothing.  Rank 0's own tables stay where the
    // engine has them.
    MMT_NCCL(rccl().GroupStart());
    if (c.rank != 0) {
        const size_t rows = R.n_rows, cells = rows * R.n_docs;
        if (rows) {
            send_pieces(my_len, rows, ncclUint32, 0, c.comm, st);
            send_pieces(my_off, cells, ncclInt64, 0, c.comm, st);
            send_pieces(my_st, cells, ncclUint8, 0, c.comm, st);
        }
        send_pieces(e.thresh_device32(), L, ncclUint32, 0, c.comm, st);
    } else {
        for (int r = 1; r < c.world; r++) {
            const size_t rows = meta[(size_t)r * 4], docs = meta[(size_t)r * 4 + 1], cells = rows * docs;
            c.len[r]->ensure(rows + 1); c.off[r]->ensure(cells + 1); c.st[r]->ensure(cells + 1); c.th[r]->ensure(L);
            if (rows) {
                recv_pieces(c.len[r]->get(), rows, ncclUint32, r, c.comm, st);
                recv_pieces(c.off[r]->get(), cells, ncclInt64, r, c.comm, st);
                recv_pieces(c.st[r]->get(), cells, ncclUint8, r, c.comm, st);
            }
            recv_pieces(c.th[r]->get(), L, ncclUint32, r, c.comm, st);
        }
    }
    MMT_NCCL(rccl().GroupEnd());
    MMT_HIP(hipStreamSynchronize(st));
    if (c.rank != 0) return MergedRows();
    std::vector<mmt_partition> parts((size_t)c.world);
    for (int r = 0; r < c.world; r++) {
        mmt_partition& p = parts[(size_t)r];
        p.n_rows = meta[(size_t)r * 4]; p.n_docs = meta[(size_t)r * 4 + 1];
        if (r == 0) { p.length = my_len; p.offsets = my_off; p.strands = my_st; p.thresh = reinterpret_cast<const uint16_t*>(e.thresh_device32()); }
        else { p.length = c.len[r]->get(); p.offsets = c.off[r]->get(); p.strands = c.st[r]->get(); p.thresh = reinterpret_cast<const uint16_t*>(c.th[r]->get()); }
        p.thresh_len = L; p.thresh_on_device = 1; p.rows_on_device = 1; p.thresh_bits = 32;
    }
    if (c.world == 1) {
        // one partition: nothing to fold; the rows are the engine's own, already in direct-run order
        MergedRows m;
        m.n_rows = parts[0].n_rows; m.n_docs = parts[0].n_docs; m.thresh_len = L;
        m.d_length.ensure(m.n_rows + 1); m.d_offsets.ensure(m.n_rows * m.n_docs + 1); m.d_strands.ensure(m.n_rows * m.n_docs + 1);
        m.d_thresh.ensure(L);
        if (m.n_rows) {
            MMT_HIP(hipMemcpyAsync(m.d_length.get(), parts[0].length, m.n_rows * 4, hipMemcpyDeviceToDevice, st));
            MMT_HIP(hipMemcpyAsync(m.d_offsets.get(), parts[0].offsets, m.n_rows * m.n_docs * 8, hipMemcpyDeviceToDevice, st));
            MMT_HIP(hipMemcpyAsync(m.d_strands.get(), parts[0].strands, m.n_rows * m.n_docs, hipMemcpyDeviceToDevice, st));
        }
        MMT_HIP(hipMemcpyAsync(m.d_thresh.get(), parts[0].thresh, L * 4, hipMemcpyDeviceToDevice, st));
        MMT_HIP(hipStreamSynchronize(st));
        m.on_host = false;
        return m;
    }
    MergedRows m = anchor_merge(e, parts.data(), parts.size(), min_len);
    sort_like_direct(e, m);
    return m;
}

// The same result by coordinate ranges (merge.cpp): every rank folds ITS slice of the anchor.  Rank q needs,
// from every rank, the rows that start in [base[q], hi[q]) and the thresholds of that range -- so both travel as an
// all-to-all of slices: world x (world - 1) messages of ncclSend / ncclRecv in one group, over all xGMI links at once
// instead of everything into rank 0's.  (Round 3 broadcast every rank's whole row table to every rank: with 94 whole
// genomes a rank's table is 30 million rows x 13 columns = 3.5 GB, and seven of them arrived on every rank to be filtered
// down to an eighth.)  The pieces go to rank 0 in rank order = anchor order.  Rank 0's work drops from world - 1 fold steps
// over the whole anchor to world - 1 steps over 1 / world of it.
static MergedRows merge_by_ranges(Comm& c, uint32_t min_len, const std::vector<uint64_t>& meta) {
    Engine& e = *c.engine;
    hipStream_t st = e.stream();
    const HostRows& R = e.rows_meta();
    const uint64_t L = e.doc_len()[0] + 1;
    const int W = c.world;
    const uint32_t* my_len; const int64_t* my_off; const uint8_t* my_st;
    e.rows_mum_device(&my_len, &my_off, &my_st);
    uint32_t longest = 0;
    for (int r = 0; r < W; r++) longest = std::max<uint32_t>(longest, (uint32_t)meta[(size_t)r * 4 + 2]);
    const uint64_t margin = fold_margin((size_t)W, longest);
    std::vector<uint64_t> lo((size_t)W), hi((size_t)W), base((size_t)W);
    for (int r = 0; r < W; r++) fold_slice_bounds(L, W, r, margin, &lo[r], &hi[r], &base[r]);
    const uint64_t my_span = hi[c.rank] - base[c.rank];
    if (W == 1) {
        mmt_partition one;
        one.n_rows = R.n_rows; one.n_docs = R.n_docs; one.length = my_len; one.offsets = my_off; one.strands = my_st;
        one.thresh = reinterpret_cast<const uint16_t*>(e.thresh_device32()); one.thresh_bits = 32;
        one.thresh_len = L; one.thresh_on_device = 1; one.rows_on_device = 1;
        MergedRows piece = anchor_merge_slice(e, &one, 1, min_len, 0, L, 0, true);
        sort_like_direct(e, piece);
        return piece;
    }
    // my rows for every destination: those that start in [base[q], hi[q]) (global coordinates; the slice fold shifts them)
    struct Out { DevBuf<uint32_t> len; DevBuf<int64_t> off; DevBuf<uint8_t> str; uint32_t n = 0; };
    std::vector<std::unique_ptr<Out>> out((size_t)W);
    std::vector<uint64_t> my_counts((size_t)W, 0);
    for (int q = 0; q < W; q++) {
        out[q].reset(new Out());
        filter_rows(e, my_len, my_off, my_st, (uint32_t)R.n_rows, (uint32_t)R.n_docs, 0, (int64_t)base[q], (int64_t)hi[q], 0,
                    out[q]->len, out[q]->off, out[q]->str, &out[q]->n);
        my_counts[q] = out[q]->n;
    }
    // counts[r][q] = rows rank r has for rank q: one all-gather of W words per rank
    std::vector<uint64_t> counts((size_t)W * W);
    {
        DevBuf<uint64_t> d_mine, d_all;
        d_mine.ensure((size_t)W); d_all.ensure((size_t)W * W);
        MMT_HIP(hipMemcpyAsync(d_mine.get(), my_counts.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
        MMT_NCCL(rccl().AllGather(d_mine.get(), d_all.get(), (size_t)W, ncclUint64, c.comm, st));
        MMT_HIP(hipMemcpyAsync(counts.data(), d_all.get(), counts.size() * 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
    }
    MMT_NCCL(rccl().GroupStart());
    for (int r = 0; r < W; r++) {
        if (r == c.rank) continue;
        // to rank r: my rows of its range, my thresholds of its range
        const size_t s_rows = out[r]->n, s_cells = s_rows * R.n_docs;
        if (s_rows) {
            send_pieces(out[r]->len.get(), s_rows, ncclUint32, r, c.comm, st);
            send_pieces(out[r]->off.get(), s_cells, ncclInt64, r, c.comm, st);
            send_pieces(out[r]->str.get(), s_cells, ncclUint8, r, c.comm, st);
        }
        send_pieces(e.thresh_device32() + base[r], hi[r] - base[r], ncclUint32, r, c.comm, st);
        // from rank r: its rows of my range, its thresholds of my range
        const size_t rows = counts[(size_t)r * W + c.rank], docs = meta[(size_t)r * 4 + 1], cells = rows * docs;
        c.len[r]->ensure(rows + 1); c.off[r]->ensure(cells + 1); c.st[r]->ensure(cells + 1); c.th[r]->ensure(my_span + 1);
        if (rows) {
            recv_pieces(c.len[r]->get(), rows, ncclUint32, r, c.comm, st);
            recv_pieces(c.off[r]->get(), cells, ncclInt64, r, c.comm, st);
            recv_pieces(c.st[r]->get(), cells, ncclUint8, r, c.comm, st);
        }
        recv_pieces(c.th[r]->get(), my_span, ncclUint32, r, c.comm, st);
    }
    MMT_NCCL(rccl().GroupEnd());
    MMT_HIP(hipStreamSynchronize(st));
    // this rank's slice
    std::vector<mmt_partition> parts((size_t)W);
    for (int r = 0; r < W; r++) {
        mmt_partition& p = parts[(size_t)r];
        p.n_docs = meta[(size_t)r * 4 + 1];
        if (r == c.rank) {
            Out& mine = *out[(size_t)c.rank];
            p.n_rows = mine.n; p.length = mine.len.get(); p.offsets = mine.off.get(); p.strands = mine.str.get();
            p.thresh = reinterpret_cast<const uint16_t*>(e.thresh_device32() + base[c.rank]);
        } else {
            p.n_rows = counts[(size_t)r * W + c.rank];
            p.length = c.len[r]->get(); p.offsets = c.off[r]->get(); p.strands = c.st[r]->get();
            p.thresh = reinterpret_cast<const uint16_t*>(c.th[r]->get());
        }
        p.thresh_len = L; p.thresh_on_device = 1; p.rows_on_device = 1; p.thresh_bits = 32;
    }
    MergedRows piece = anchor_merge_slice(e, parts.data(), parts.size(), min_len, lo[c.rank], hi[c.rank], base[c.rank], true);
    out.clear();
    // the pieces to rank 0, in rank order
    const std::vector<uint64_t> pm = exchange_meta(c, piece.n_rows, piece.n_docs, 0);
    std::vector<MergedRows> pieces;
    MMT_NCCL(rccl().GroupStart());
    if (c.rank != 0) {
        const size_t rows = piece.n_rows, cells = rows * piece.n_docs;
        if (rows) {
            send_pieces(piece.d_length.get(), rows, ncclUint32, 0, c.comm, st);
            send_pieces(piece.d_offsets.get(), cells, ncclInt64, 0, c.comm, st);
            send_pieces(piece.d_strands.get(), cells, ncclUint8, 0, c.comm, st);
        }
        send_pieces(piece.d_thresh.get(), piece.thresh_len, ncclUint32, 0, c.comm, st);
    } else {
        pieces.resize((size_t)W);
        pieces[0] = std::move(piece);
        for (int r = 1; r < W; r++) {
            MergedRows& p = pieces[(size_t)r];
            p.n_rows = pm[(size_t)r * 4]; p.n_docs = pm[(size_t)r * 4 + 1]; p.thresh_len = hi[r] - lo[r];
            const size_t cells = p.n_rows * p.n_docs;
            p.d_length.ensure(p.n_rows + 1); p.d_offsets.ensure(cells + 1); p.d_strands.ensure(cells + 1); p.d_thresh.ensure(p.thresh_len + 1);
            if (p.n_rows) {
                recv_pieces(p.d_length.get(), p.n_rows, ncclUint32, r, c.comm, st);
                recv_pieces(p.d_offsets.get(), cells, ncclInt64, r, c.comm, st);
                recv_pieces(p.d_strands.get(), cells, ncclUint8, r, c.comm, st);
            }
            recv_pieces(p.d_thresh.get(), p.thresh_len, ncclUint32, r, c.comm, st);
        }
    }
    MMT_NCCL(rccl().GroupEnd());
    MMT_HIP(hipStreamSynchronize(st));
    if (c.rank != 0) return MergedRows();
    MergedRows m = concat_pieces(e, pieces);
    sort_like_direct(e, m);
    return m;
}

// The messages of merge_on_rank0 / merge_by_ranges with THIS rank as its own peer: the row tables and the 32-bit threshold column
// of the engine's last run travel through ncclSend / ncclRecv (one group, pieces of at most 2^30 elements), an all-gather of the
// meta words and a broadcast of the thresholds -- what a one-GPU box can show the real library of the exchange's message sizes
// (a rank's share of configs[3]: 3.05 G thresholds = 12.2 GB, 30 M x 13 rows), its size_t counts and its stream ordering.
// out: [0] bytes moved, [1] message pieces, [2] largest piece in bytes, [3] elements that arrived different, [4] microseconds,
// [5] rows, [6] row cells, [7] thresholds.
template <typename T>
__global__ void k_count_diff(const T* __restrict__ a, const T* __restrict__ b, size_t n, unsigned long long* __restrict__ diff) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mine += a[i] != b[i] ? 1 : 0;
    if (mine) atomicAdd(diff, mine);
}
void dist_loopback(Comm& c, uint64_t out[8]) {
    Engine& e = *c.engine;
    MMT_HIP(hipSetDevice(e.device()));
    hipStream_t st = e.stream();
    const HostRows& R = e.rows_meta();
    if (!R.mum_mode || !e.thresh_len()) throw std::runtime_error("the exchange needs a multi-MUM run with merge metadata");
    const uint64_t L = e.doc_len()[0] + 1;
    const uint32_t* my_len; const int64_t* my_off; const uint8_t* my_st;
    e.rows_mum_device(&my_len, &my_off, &my_st);
    const size_t rows = R.n_rows, cells = rows * R.n_docs;
    const int me = c.rank;
    c.len[me]->ensure(rows + 1); c.off[me]->ensure(cells + 1); c.st[me]->ensure(cells + 1); c.th[me]->ensure(L + 1);
    DevBuf<uint32_t> bc;
    bc.ensure(L + 1);
    DevBuf<unsigned long long> diff;
    diff.ensure(1);
    MMT_HIP(hipMemsetAsync(diff.get(), 0, 8, st));
    MMT_HIP(hipMemsetAsync(c.th[me]->get(), 0xa5, L * 4, st));             // (the stream orders the fill before the receive)
    MMT_HIP(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    (void)exchange_meta(c, rows, R.n_docs, L);                               // all-gather of the meta words
    MMT_NCCL(rccl().GroupStart());
    if (rows) {
        send_pieces(my_len, rows, ncclUint32, me, c.comm, st); recv_pieces(c.len[me]->get(), rows, ncclUint32, me, c.comm, st);
        send_pieces(my_off, cells, ncclInt64, me, c.comm, st); recv_pieces(c.off[me]->get(), cells, ncclInt64, me, c.comm, st);
        send_pieces(my_st, cells, ncclUint8, me, c.comm, st); recv_pieces(c.st[me]->get(), cells, ncclUint8, me, c.comm, st);
    }
    send_pieces(e.thresh_device32(), L, ncclUint32, me, c.comm, st); recv_pieces(c.th[me]->get(), L, ncclUint32, me, c.comm, st);
    MMT_NCCL(rccl().GroupEnd());
    // a broadcast in pieces as well (round 2's route; still what a one-to-all step would use)
    const size_t C = rccl_chunk_elements(4);
    uint64_t pieces = 0, largest = 0;
    MMT_HIP(hipMemcpyAsync(bc.get(), e.thresh_device32(), L * 4, hipMemcpyDeviceToDevice, st));
    for (size_t at = 0; at < L; at += C) {
        const size_t k = std::min<size_t>(C, L - at);
        MMT_NCCL(rccl().Broadcast(bc.get() + at, bc.get() + at, k, ncclUint32, 0, c.comm, st));
    }
    MMT_HIP(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    auto count = [&](size_t n, size_t width) { const size_t Cw = rccl_chunk_elements(width); for (size_t at = 0; at < n || at == 0; at += Cw) { pieces++; largest = std::max<uint64_t>(largest, std::min(Cw, n - at) * width); if (n <= at + Cw) break; } };
    if (rows) { count(rows, 4); count(cells, 8); count(cells, 1); }
    count(L, 4);
    if (rows) {
        hipLaunchKernelGGL(k_count_diff<uint32_t>, dim3(2048), dim3(256), 0, st, my_len, (const uint32_t*)c.len[me]->get(), rows, diff.get());
        hipLaunchKernelGGL(k_count_diff<int64_t>, dim3(2048), dim3(256), 0, st, my_off, (const int64_t*)c.off[me]->get(), cells, diff.get());
        hipLaunchKernelGGL(k_count_diff<uint8_t>, dim3(2048), dim3(256), 0, st, my_st, (const uint8_t*)c.st[me]->get(), cells, diff.get());
    }
    hipLaunchKernelGGL(k_count_diff<uint32_t>, dim3(2048), dim3(256), 0, st, e.thresh_device32(), (const uint32_t*)c.th[me]->get(), (size_t)L, diff.get());
    hipLaunchKernelGGL(k_count_diff<uint32_t>, dim3(2048), dim3(256), 0, st, e.thresh_device32(), (const uint32_t*)bc.get(), (size_t)L, diff.get());
    MMT_HIP(hipGetLastError());
    unsigned long long d = 0;
    MMT_HIP(hipMemcpyAsync(&d, diff.get(), 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    out[0] = rows * 4 + cells * 9 + L * 4 + L * 4; out[1] = pieces; out[2] = largest; out[3] = d; out[4] = (uint64_t)us;
    out[5] = rows; out[6] = cells; out[7] = L;
    c.len[me]->release(); c.off[me]->release(); c.st[me]->release(); c.th[me]->release();
}

// One message of `elements` elements of `width` bytes (1, 4 or 8) with this rank as its own peer, in the pieces send_pieces /
// recv_pieces cut: a pattern goes out, what arrives is compared.  What a one-GPU box can ask the real library about a message
// SIZE without a whole-genome run behind it.  out: [0] elements that arrived different, [1] pieces, [2] largest piece in bytes,
// [3] microseconds.
template <typename T>
__global__ void k_pattern(T* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = (T)(i * 2654435761ull + (i >> 29));
}
template <typename T>
static void selftest_typed(Comm& c, size_t n, ncclDataType_t t, uint64_t out[4]) {
    Engine& e = *c.engine;
    hipStream_t st = e.stream();
    DevBuf<T> src, dst;
    DevBuf<unsigned long long> diff;
    src.ensure(n + 1); dst.ensure(n + 1); diff.ensure(1);
    hipLaunchKernelGGL(k_pattern<T>, dim3(2048), dim3(256), 0, st, src.get(), n);
    MMT_HIP(hipMemsetAsync(dst.get(), 0xa5, n * sizeof(T), st));
    MMT_HIP(hipMemsetAsync(diff.get(), 0, 8, st));
    MMT_HIP(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    MMT_NCCL(rccl().GroupStart());
    send_pieces(src.get(), n, t, c.rank, c.comm, st);
    recv_pieces(dst.get(), n, t, c.rank, c.comm, st);
    MMT_NCCL(rccl().GroupEnd());
    MMT_HIP(hipStreamSynchronize(st));
    out[3] = (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipLaunchKernelGGL(k_count_diff<T>, dim3(2048), dim3(256), 0, st, (const T*)src.get(), (const T*)dst.get(), n, diff.get());
    MMT_HIP(hipGetLastError());
    unsigned long long d = 0;
    MMT_HIP(hipMemcpyAsync(&d, diff.get(), 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    const size_t C = rccl_chunk_elements(sizeof(T));
    out[0] = d; out[1] = n ? (n + C - 1) / C : 1; out[2] = std::min(C, n) * sizeof(T);
}
void dist_selftest(Comm& c, uint64_t elements, uint32_t width, uint64_t out[4]) {
    MMT_HIP(hipSetDevice(c.engine->device()));
    if (width == 1) selftest_typed<uint8_t>(c, (size_t)elements, ncclUint8, out);
    else if (width == 4) selftest_typed<uint32_t>(c, (size_t)elements, ncclUint32, out);
    else if (width == 8) selftest_typed<int64_t>(c, (size_t)elements, ncclInt64, out);
    else throw std::runtime_error("dist_selftest: width 1, 4 or 8");
}

// Modes without a partition merge: this rank's PREFIX.mums / .mems bytes (mmt_engine_set_scan_shard) to rank 0, in rank
// order -- sent to rank 0 only (round 3 broadcast every rank's bytes to every rank, and only rank 0 read them).  Returns the
// whole output on rank 0, an empty string elsewhere.
std::string dist_gather_text(Comm& c) {
    Engine& e = *c.engine;
    MMT_HIP(hipSetDevice(e.device()));
    hipStream_t st = e.stream();
    const HostRows& R = e.rows(Engine::ROWS_TEXT);
    const std::vector<uint64_t> meta = exchange_meta(c, R.n_rows, R.n_docs, R.text_len);
    DevBuf<uint8_t> mine;
    if (c.rank != 0) {
        mine.ensure(R.text_len + 1);
        if (R.text_len) MMT_HIP(hipMemcpyAsync(mine.get(), R.text, R.text_len, hipMemcpyHostToDevice, st));
    }
    MMT_NCCL(rccl().GroupStart());
    if (c.rank != 0) {
        if (R.text_len) send_pieces(mine.get(), R.text_len, ncclUint8, 0, c.comm, st);
    } else {
        for (int r = 1; r < c.world; r++) {
            const size_t bytes = meta[(size_t)r * 4 + 2];
            c.text[r]->ensure(bytes + 1);
            if (bytes) recv_pieces(c.text[r]->get(), bytes, ncclUint8, r, c.comm, st);
        }
    }
    MMT_NCCL(rccl().GroupEnd());
    MMT_HIP(hipStreamSynchronize(st));
    if (c.rank != 0) return std::string();
    size_t total = 0;
    for (int r = 0; r < c.world; r++) total += meta[(size_t)r * 4 + 2];
    std::string out(total, '\0');
    if (R.text_len) std::memcpy(&out[0], R.text, R.text_len);
    size_t at = R.text_len;
    for (int r = 1; r < c.world; r++) {
        const size_t bytes = meta[(size_t)r * 4 + 2];
        if (bytes) MMT_HIP(hipMemcpyAsync(&out[at], c.text[r]->get(), bytes, hipMemcpyDeviceToHost, st));
        at += bytes;
    }
    MMT_HIP(hipStreamSynchronize(st));
    return out;
}

}  // namespace mmt
