// merge.cpp -- see merge.hpp.  The whole fold runs in HBM: partitions that arrive in host memory
// are uploaded once, every fold step is a handful of launches plus one 8-byte read-back (the
// number of new rows), and the merged tables are materialised and formatted on the device.
#include "merge.hpp"
#include "fasta.hpp"

#include <algorithm>
#include <map>
#include <mutex>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "kernels.hpp"
#include "merge_kernels.hpp"
#include "pfp_kernels.hpp"
#include "prims.hpp"

namespace mmt {
namespace {

int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b ? b : 1; }

// One side of a fold step in anchor order (see merge_kernels.hpp).
struct SideBuf {
    DevBuf<uint64_t> start;
    DevBuf<uint32_t> len, src;
    DevBuf<int64_t> plus, minus;
    uint32_t n = 0, n_parts = 0;
    void ensure(size_t rows, size_t parts) {
        start.ensure(rows + 1); len.ensure(rows + 1);
        src.ensure(rows * parts + 1); plus.ensure(rows * parts + 1); minus.ensure(rows * parts + 1);
        n = (uint32_t)rows; n_parts = (uint32_t)parts;
    }
    mk::SideView view() const { return mk::SideView{start.get(), len.get(), src.get(), plus.get(), minus.get(), n, n_parts}; }
    void swap(SideBuf& o) {
        start.swap(o.start); len.swap(o.len); src.swap(o.src); plus.swap(o.plus); minus.swap(o.minus);
        std::swap(n, o.n); std::swap(n_parts, o.n_parts);
    }
};

// start flags over the anchor (what k_fold_step walks; the row of a position comes from a search in the starts)
struct SideIndex {
    DevBuf<uint8_t> bv;
    void build(const SideBuf& s, uint64_t L, hipStream_t st) {
        bv.ensure(L);
        MMT_HIP(hipMemsetAsync(bv.get(), 0, L, st));
        k::mark_starts(s.start.get(), s.n, bv.get(), st);
    }
};

struct PartUpload {
    DevBuf<uint32_t> length; DevBuf<int64_t> offsets; DevBuf<uint8_t> strands;
};

// device scratch of the fold, kept between calls (one merge per bench step on rank 0)
struct MergeScratch {
    DevBuf<uint32_t> nb_left, nb_right, nb_out;
    DevBuf<uint16_t> narrow;         // a 16-bit column on its way into the 32-bit ones
    SideIndex ia, ib;
    SideBuf left, right, out;
    DevBuf<uint64_t> d_pos, keys_a, keys_b;
    DevBuf<uint32_t> d_ra, d_rb, d_len, d_count, vals_a, vals_b;
    std::vector<std::unique_ptr<PartUpload>> uploads;
    DevBuf<mk::PartTable> d_parts;
    DevBuf<uint32_t> d_colpart;
};
// one scratch set per device and host thread (an engine on another GPU must not reuse buffers that live on the first one,
// and two engines on one GPU that merge from different threads must not share them); the sets are leaked on purpose: a
// static destructor would run after the HIP runtime has gone
std::mutex& scratch_mutex() { static std::mutex mu; return mu; }
std::map<std::pair<int, std::thread::id>, MergeScratch*>& scratch_sets() {
    static auto* sets = new std::map<std::pair<int, std::thread::id>, MergeScratch*>();
    return *sets;
}
MergeScratch& scratch(int device) {
    std::lock_guard<std::mutex> lock(scratch_mutex());
    MergeScratch*& s = scratch_sets()[std::make_pair(device, std::this_thread::get_id())];
    if (!s) s = new MergeScratch();
    return *s;
}

}  // namespace

// The fold's scratch back to the device heap (mmt_pool_trim: a long-lived process between two jobs).  A fold at the length of a
// human chromosome set leaves tens of GB here, scattered over the heap: the next job's 137 GB text then found no room in a
// heap of 235 GB with 6 GB live.  Nobody may be folding while this runs.
void merge_release_scratch() {
    std::lock_guard<std::mutex> lock(scratch_mutex());
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : scratch_sets()) {
        MergeScratch* M = kv.second;
        if (!M) continue;
        (void)hipSetDevice(kv.first.first);
        for (DevBuf<uint32_t>* b : {&M->nb_left, &M->nb_right, &M->nb_out, &M->d_ra, &M->d_rb, &M->d_len, &M->d_count, &M->vals_a, &M->vals_b,
                                    &M->d_colpart, &M->left.len, &M->left.src, &M->right.len, &M->right.src, &M->out.len, &M->out.src})
            b->release();
        for (DevBuf<uint64_t>* b : {&M->d_pos, &M->keys_a, &M->keys_b, &M->left.start, &M->right.start, &M->out.start}) b->release();
        for (DevBuf<int64_t>* b : {&M->left.plus, &M->left.minus, &M->right.plus, &M->right.minus, &M->out.plus, &M->out.minus}) b->release();
        M->narrow.release(); M->ia.bv.release(); M->ib.bv.release(); M->d_parts.release();
        M->uploads.clear();
    }
    (void)hipSetDevice(cur);
}

MergedRows anchor_merge(Engine& e, const mmt_partition* parts, size_t k, uint32_t min_len) {
    const bool dbg = std::getenv("MMT_MERGE_DEBUG") != nullptr;
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        MMT_HIP(hipStreamSynchronize(st));
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[merge] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    const uint64_t L = parts[0].thresh_len;
    size_t n_docs_out = 0;
    for (size_t i = 0; i < k; i++) {
        if (parts[i].thresh_len != L) throw std::runtime_error("partitions disagree on the anchor length");
        if (parts[i].n_rows >= 0xffffffffull) throw std::runtime_error("too many rows in a partition");
        if (parts[i].n_docs == 0) throw std::runtime_error("a partition without documents");
        n_docs_out += parts[i].n_docs - (i ? 1 : 0);
    }
    MergeScratch& M = scratch(e.device());
    DevBuf<uint8_t>& temp = e.scratch();

    // partition tables in HBM (uploaded when they arrive in host memory) + the column map of the result
    std::vector<mk::PartTable> tab(k);
    std::vector<uint32_t> col_part(n_docs_out);
    if (M.uploads.size() < k) M.uploads.resize(k);
    uint32_t col = 0;
    for (size_t g = 0; g < k; g++) {
        const mmt_partition& P = parts[g];
        mk::PartTable& t = tab[g];
        t.n_rows = (uint32_t)P.n_rows; t.n_docs = (uint32_t)P.n_docs; t.skip = g ? 1 : 0; t.first_col = col;
        for (uint32_t c = t.skip; c < t.n_docs; c++) col_part[col++] = (uint32_t)g;
        if (P.rows_on_device) {
            t.length = P.length; t.offsets = P.offsets; t.strands = P.strands;
        } else {
            if (!M.uploads[g]) M.uploads[g].reset(new PartUpload());
            PartUpload& U = *M.uploads[g];
            const size_t cells = (size_t)P.n_rows * P.n_docs;
            U.length.ensure(P.n_rows + 1); U.offsets.ensure(cells + 1); U.strands.ensure(cells + 1);
            if (P.n_rows) {
                MMT_HIP(hipMemcpyAsync(U.length.get(), P.length, P.n_rows * 4, hipMemcpyHostToDevice, st));
                MMT_HIP(hipMemcpyAsync(U.offsets.get(), P.offsets, cells * 8, hipMemcpyHostToDevice, st));
                MMT_HIP(hipMemcpyAsync(U.strands.get(), P.strands, cells, hipMemcpyHostToDevice, st));
            }
            t.length = U.length.get(); t.offsets = U.offsets.get(); t.strands = U.strands.get();
        }
    }
    M.d_parts.ensure(k); M.d_colpart.ensure(n_docs_out);
    MMT_HIP(hipMemcpyAsync(M.d_parts.get(), tab.data(), k * sizeof(mk::PartTable), hipMemcpyHostToDevice, st));
    MMT_HIP(hipMemcpyAsync(M.d_colpart.get(), col_part.data(), n_docs_out * 4, hipMemcpyHostToDevice, st));
    M.d_count.ensure(4);
    MMT_HIP(hipMemsetAsync(M.d_count.get(), 0, 16, st));     // [0] rows of this step, [1] input errors
    lap("tables");

    // thresholds are folded at 32 bits; a partition that brings the reference's 16-bit column (a
    // PREFIX.athresh file, saturated at 65535) is widened on arrival
    auto load_thresh = [&](const mmt_partition& p, DevBuf<uint32_t>& dst) {
        dst.ensure(L);
        if (p.thresh_bits == 32) {
            MMT_HIP(hipMemcpyAsync(dst.get(), p.thresh, L * 4, p.thresh_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        } else if (p.thresh_on_device) {
            k::thresh_widen(p.thresh, L, dst.get(), st);
        } else {
            M.narrow.ensure(L);
            MMT_HIP(hipMemcpyAsync(M.narrow.get(), p.thresh, L * 2, hipMemcpyHostToDevice, st));
            k::thresh_widen(M.narrow.get(), L, dst.get(), st);
        }
    };
    // parse_candidate(): rows in anchor order (merge_candidates.cpp:89-92)
    const int key_bits = bit_width_u64(L);
    auto make_leaf = [&](size_t g, SideBuf& side) {
        const mk::PartTable& t = tab[g];
        const uint32_t n = t.n_rows;
        side.ensure(n, 1);
        M.keys_a.ensure(n + 1); M.keys_b.ensure(n + 1); M.vals_a.ensure(n + 1); M.vals_b.ensure(n + 1);
        if (!n) return;
        mk::leaf_keys(t.offsets, n, t.n_docs, L, M.keys_a.get(), M.vals_a.get(), M.d_count.get() + 1, st);
        prims::sort_pairs_u64_u32(temp, M.keys_a.get(), M.keys_b.get(), M.vals_a.get(), M.vals_b.get(), n, 0, key_bits,
                                  st);
        mk::leaf_side(M.keys_b.get(), M.vals_b.get(), t.length, n, side.view(), st);
    };
    make_leaf(0, M.left);
    load_thresh(parts[0], M.nb_left);
    for (size_t pi = 1; pi < k; pi++) {
        make_leaf(pi, M.right);
        load_thresh(parts[pi], M.nb_right);
        M.nb_out.ensure(L);
        M.ia.build(M.left, L, st);
        M.ib.build(M.right, L, st);
        const size_t capacity = (size_t)M.left.n + M.right.n + 1;
        M.d_pos.ensure(capacity); M.d_ra.ensure(capacity); M.d_rb.ensure(capacity); M.d_len.ensure(capacity);
        MMT_HIP(hipMemsetAsync(M.d_count.get(), 0, 4, st));
        k::FoldArgs a;
        a.len = L; a.nb_a = M.nb_left.get(); a.nb_b = M.nb_right.get(); a.nb_out = M.nb_out.get();
        a.n_a = (uint32_t)M.left.n; a.n_b = (uint32_t)M.right.n;
        a.start_a = M.left.start.get(); a.start_b = M.right.start.get();
        a.len_a = M.left.len.get(); a.len_b = M.right.len.get();
        a.bv_a = M.ia.bv.get(); a.bv_b = M.ib.bv.get();
        a.out_pos = M.d_pos.get(); a.out_ra = M.d_ra.get(); a.out_rb = M.d_rb.get(); a.out_len = M.d_len.get();
        a.capacity = (uint32_t)capacity; a.d_count = M.d_count.get(); a.min_len = min_len;
        k::fold_step(a, st);
        uint32_t h[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(h, M.d_count.get(), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (h[1] & 1u) throw std::runtime_error("anchor offset outside the threshold array");
        const uint32_t found = h[0];
        if (found > capacity) throw std::runtime_error("anchor merge produced more rows than MUM starts");
        // new rows in anchor order (every anchor position starts at most one row)
        M.keys_b.ensure(found + 1); M.vals_a.ensure(found + 1); M.vals_b.ensure(found + 1);
        M.out.ensure(found, M.left.n_parts + 1);
        if (found) {
            pk::iota(M.vals_a.get(), found, st);
            prims::sort_pairs_u64_u32(temp, M.d_pos.get(), M.keys_b.get(), M.vals_a.get(), M.vals_b.get(), found, 0,
                                      key_bits, st);
            mk::fold_rows(M.vals_b.get(), M.d_pos.get(), M.d_ra.get(), M.d_rb.get(), M.d_len.get(), found,
                          M.left.view(), M.right.view(), M.out.view(), st);
        }
        M.left.swap(M.out);
        M.nb_left.swap(M.nb_out);
        lap("fold step");
    }
    if (k == 1) {
        uint32_t h[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(h, M.d_count.get(), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (h[1] & 1u) throw std::runtime_error("anchor offset outside the threshold array");
    }
    // materialise the surviving rows: partition 0's columns, then every other partition's without its anchor
    MergedRows m;
    m.n_docs = n_docs_out; m.n_rows = M.left.n; m.thresh_len = L;
    m.d_length.ensure(m.n_rows + 1); m.d_offsets.ensure(m.n_rows * n_docs_out + 1);
    m.d_strands.ensure(m.n_rows * n_docs_out + 1); m.d_thresh.ensure(L);
    mk::materialise(M.left.view(), nullptr, M.d_parts.get(), (uint32_t)n_docs_out, M.d_colpart.get(),
                    m.d_length.get(), m.d_offsets.get(), m.d_strands.get(), st);
    MMT_HIP(hipMemcpyAsync(m.d_thresh.get(), M.nb_left.get(), L * 4, hipMemcpyDeviceToDevice, st));
    MMT_HIP(hipStreamSynchronize(st));       // host partitions may be released by the caller from here on
    lap("materialise");
    return m;
}

// ---- the fold by coordinate ranges -----------------------------------------------------------------------
// A row of the fold that starts at anchor position x is decided by the rows and thresholds of every partition within
// (partitions - 1) x the longest row to its left (one row length per fold step) and up to its own end to the right: a slice
// [lo, hi) of the anchor folds by itself from the rows that start in [base, hi) and the thresholds [base, hi), base = lo -
// margin.  N ranks fold N slices at once, and of the thresholds -- 2 bytes per anchor position and partition, 6 GB for a whole
// genome -- every rank needs its slice only.
uint64_t fold_margin(size_t k, uint32_t longest) { return (uint64_t)(k ? k - 1 : 0) * longest + 1; }

void fold_slice_bounds(uint64_t L, int world, int r, uint64_t margin, uint64_t* lo, uint64_t* hi, uint64_t* base) {
    *lo = L * (uint64_t)r / (uint64_t)world;
    *hi = L * (uint64_t)(r + 1) / (uint64_t)world;
    *base = *lo > margin ? *lo - margin : 0;
}

// rows of `in` (device tables) whose anchor offset + shift lies in [lo, hi), anchor offsets moved by delta
void filter_rows(Engine& e, const uint32_t* len, const int64_t* off, const uint8_t* st, uint32_t n, uint32_t n_docs,
                        int64_t shift, int64_t lo, int64_t hi, int64_t delta, DevBuf<uint32_t>& o_len, DevBuf<int64_t>& o_off,
                        DevBuf<uint8_t>& o_st, uint32_t* kept) {
    hipStream_t st_ = e.stream();
    DevBuf<uint8_t> flags;
    DevBuf<uint32_t> idx, count;
    flags.ensure((size_t)n + 1); idx.ensure((size_t)n + 1); count.ensure(4);
    uint32_t m = 0;
    if (n) {
        mk::range_flags(off, n, n_docs, shift, lo, hi, flags.get(), st_);
        prims::select_indices(e.scratch(), flags.get(), idx.get(), count.get(), n, st_);
        MMT_HIP(hipMemcpyAsync(&m, count.get(), 4, hipMemcpyDeviceToHost, st_));
        MMT_HIP(hipStreamSynchronize(st_));
    }
    o_len.ensure((size_t)m + 1); o_off.ensure((size_t)m * n_docs + 1); o_st.ensure((size_t)m * n_docs + 1);
    if (m) {
        mk::permute_rows(idx.get(), m, n_docs, len, off, st, o_len.get(), o_off.get(), o_st.get(), st_);
        mk::shift_anchor(o_off.get(), m, n_docs, delta, st_);
    }
    MMT_HIP(hipStreamSynchronize(st_));
    *kept = m;
}

MergedRows anchor_merge_slice(Engine& e, const mmt_partition* parts, size_t k, uint32_t min_len, uint64_t lo, uint64_t hi,
                              uint64_t base, bool thresh_is_slice) {
    MMT_HIP(hipSetDevice(e.device()));
    hipStream_t st = e.stream();
    struct Held { DevBuf<uint32_t> len, in_len; DevBuf<int64_t> off, in_off; DevBuf<uint8_t> str, in_str, th; };
    std::vector<std::unique_ptr<Held>> held(k);
    std::vector<mmt_partition> sliced(k);
    for (size_t g = 0; g < k; g++) {
        const mmt_partition& P = parts[g];
        held[g].reset(new Held());
        Held& H = *held[g];
        if (P.n_rows >= 0xffffffffull) throw std::runtime_error("too many rows in a partition");
        const uint32_t n = (uint32_t)P.n_rows, nd = (uint32_t)P.n_docs;
        const uint32_t* len = P.length; const int64_t* off = P.offsets; const uint8_t* str = P.strands;
        if (!P.rows_on_device && n) {
            H.in_len.ensure(n); H.in_off.ensure((size_t)n * nd); H.in_str.ensure((size_t)n * nd);
            MMT_HIP(hipMemcpyAsync(H.in_len.get(), P.length, (size_t)n * 4, hipMemcpyHostToDevice, st));
            MMT_HIP(hipMemcpyAsync(H.in_off.get(), P.offsets, (size_t)n * nd * 8, hipMemcpyHostToDevice, st));
            MMT_HIP(hipMemcpyAsync(H.in_str.get(), P.strands, (size_t)n * nd, hipMemcpyHostToDevice, st));
            len = H.in_len.get(); off = H.in_off.get(); str = H.in_str.get();
        }
        uint32_t kept = 0;
        filter_rows(e, len, off, str, n, nd, 0, (int64_t)base, (int64_t)hi, -(int64_t)base, H.len, H.off, H.str, &kept);
        H.in_len.release(); H.in_off.release(); H.in_str.release();
        mmt_partition& Q = sliced[g];
        Q = P;
        Q.n_rows = kept; Q.length = H.len.get(); Q.offsets = H.off.get(); Q.strands = H.str.get(); Q.rows_on_device = 1;
        const size_t tw = P.thresh_bits == 32 ? 4 : 2;          // bytes per threshold as the partition brings them
        const uint8_t* th = reinterpret_cast<const uint8_t*>(P.thresh) + (thresh_is_slice ? 0 : base) * tw;
        if (!P.thresh_on_device) {
            H.th.ensure((hi - base) * tw);
            MMT_HIP(hipMemcpyAsync(H.th.get(), th, (hi - base) * tw, hipMemcpyHostToDevice, st));
            th = H.th.get();
        }
        Q.thresh = reinterpret_cast<const uint16_t*>(th); Q.thresh_len = hi - base; Q.thresh_on_device = 1;
    }
    MergedRows whole = anchor_merge(e, sliced.data(), k, min_len);
    held.clear();
    // the rows that start in [lo, hi), in the coordinates of the whole anchor; the thresholds of [lo, hi)
    MergedRows piece;
    piece.n_docs = whole.n_docs; piece.thresh_len = hi - lo;
    uint32_t kept = 0;
    filter_rows(e, whole.d_length.get(), whole.d_offsets.get(), whole.d_strands.get(), (uint32_t)whole.n_rows, (uint32_t)whole.n_docs,
                (int64_t)base, (int64_t)lo, (int64_t)hi, (int64_t)base, piece.d_length, piece.d_offsets, piece.d_strands, &kept);
    piece.n_rows = kept;
    piece.d_thresh.ensure(hi - lo + 1);
    MMT_HIP(hipMemcpyAsync(piece.d_thresh.get(), whole.d_thresh.get() + (lo - base), (hi - lo) * 4, hipMemcpyDeviceToDevice, st));
    MMT_HIP(hipStreamSynchronize(st));
    return piece;
}

MergedRows concat_pieces(Engine& e, std::vector<MergedRows>& pieces) {
    hipStream_t st = e.stream();
    MergedRows m;
    if (pieces.empty()) return m;
    m.n_docs = pieces[0].n_docs;
    for (const MergedRows& p : pieces) { m.n_rows += p.n_rows; m.thresh_len += p.thresh_len; }
    m.d_length.ensure(m.n_rows + 1); m.d_offsets.ensure(m.n_rows * m.n_docs + 1); m.d_strands.ensure(m.n_rows * m.n_docs + 1);
    m.d_thresh.ensure(m.thresh_len + 1);
    size_t row = 0, pos = 0;
    for (const MergedRows& p : pieces) {
        if (p.n_docs != m.n_docs) throw std::runtime_error("the pieces of a range fold disagree on the number of documents");
        if (p.n_rows) {
            MMT_HIP(hipMemcpyAsync(m.d_length.get() + row, p.d_length.get(), p.n_rows * 4, hipMemcpyDeviceToDevice, st));
            MMT_HIP(hipMemcpyAsync(m.d_offsets.get() + row * m.n_docs, p.d_offsets.get(), p.n_rows * m.n_docs * 8, hipMemcpyDeviceToDevice, st));
            MMT_HIP(hipMemcpyAsync(m.d_strands.get() + row * m.n_docs, p.d_strands.get(), p.n_rows * m.n_docs, hipMemcpyDeviceToDevice, st));
        }
        if (p.thresh_len)
            MMT_HIP(hipMemcpyAsync(m.d_thresh.get() + pos, p.d_thresh.get(), p.thresh_len * 4, hipMemcpyDeviceToDevice, st));
        row += p.n_rows; pos += p.thresh_len;
    }
    MMT_HIP(hipStreamSynchronize(st));
    return m;
}

// the longest row of a partition (device or host table)
uint32_t longest_row(Engine& e, const uint32_t* length, size_t n_rows, bool on_device) {
    if (!n_rows) return 0;
    std::vector<uint32_t> h;
    const uint32_t* p = length;
    if (on_device) {
        h.resize(n_rows);
        MMT_HIP(hipMemcpyAsync(h.data(), length, n_rows * 4, hipMemcpyDeviceToHost, e.stream()));
        MMT_HIP(hipStreamSynchronize(e.stream()));
        p = h.data();
    }
    return *std::max_element(p, p + n_rows);
}

// anchor_merge(parts) computed as `slices` independent slices of the anchor, one after the other on this device: what
// `slices` ranks do at once (dist.cpp dist_merge_ranges); the result is the same table
MergedRows anchor_merge_by_ranges(Engine& e, const mmt_partition* parts, size_t k, int slices, uint32_t min_len) {
    if (slices < 1) throw std::invalid_argument("a range fold needs at least one slice");
    const uint64_t L = parts[0].thresh_len;
    uint32_t longest = 0;
    for (size_t g = 0; g < k; g++) longest = std::max(longest, longest_row(e, parts[g].length, parts[g].n_rows, parts[g].rows_on_device != 0));
    const uint64_t margin = fold_margin(k, longest);
    // row tables that arrive in host memory go to the device once, not once per slice (eight shares of 94 whole genomes:
    // 28 GB of rows); the thresholds -- the O(L_0) columns -- stay where they are and travel slice by slice
    struct Up { DevBuf<uint32_t> len; DevBuf<int64_t> off; DevBuf<uint8_t> str; };
    std::vector<std::unique_ptr<Up>> up(k);
    std::vector<mmt_partition> dev(parts, parts + k);
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    for (size_t g = 0; g < k; g++) {
        const mmt_partition& P = parts[g];
        if (P.rows_on_device || !P.n_rows) continue;
        up[g].reset(new Up());
        const size_t cells = (size_t)P.n_rows * P.n_docs;
        up[g]->len.ensure(P.n_rows); up[g]->off.ensure(cells); up[g]->str.ensure(cells);
        MMT_HIP(hipMemcpyAsync(up[g]->len.get(), P.length, P.n_rows * 4, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemcpyAsync(up[g]->off.get(), P.offsets, cells * 8, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemcpyAsync(up[g]->str.get(), P.strands, cells, hipMemcpyHostToDevice, st));
        dev[g].length = up[g]->len.get(); dev[g].offsets = up[g]->off.get(); dev[g].strands = up[g]->str.get();
        dev[g].rows_on_device = 1;
    }
    MMT_HIP(hipStreamSynchronize(st));
    const bool dbg = std::getenv("MMT_MERGE_DEBUG") != nullptr;
    std::vector<MergedRows> pieces;
    for (int r = 0; r < slices; r++) {
        uint64_t lo, hi, base;
        fold_slice_bounds(L, slices, r, margin, &lo, &hi, &base);
        const auto t0 = std::chrono::steady_clock::now();
        pieces.push_back(anchor_merge_slice(e, dev.data(), k, min_len, lo, hi, base, false));
        if (dbg) std::fprintf(stderr, "[merge] slice %d of %d: anchor [%llu, %llu) from %llu, %zu rows, %.1f ms\n", r, slices,
                              (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)base, pieces.back().n_rows,
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    up.clear();
    return concat_pieces(e, pieces);
}

void sort_like_direct(Engine& e, MergedRows& m) {
    const size_t n = m.n_rows;
    if (!n) return;
    if (e.text_length() == 0 || !e.anchor_ranks_valid())
        throw std::runtime_error("engine holds no suffix ranks of the anchor: run it on a partition (with merge metadata) first");
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    MergeScratch& M = scratch(e.device());
    // anchor = document 0 of the engine's text, '+' strand starts at text offset 0
    DevBuf<uint32_t> key_a, key_b;
    key_a.ensure(n); key_b.ensure(n); M.vals_a.ensure(n + 1); M.vals_b.ensure(n + 1); M.d_count.ensure(4);
    MMT_HIP(hipMemsetAsync(M.d_count.get(), 0, 16, st));
    if (e.wide()) {
        DevBuf<uint64_t> k64_a, k64_b;
        k64_a.ensure(n); k64_b.ensure(n);
        mk::rank_keys64(m.d_offsets.get(), (uint32_t)n, (uint32_t)m.n_docs, e.isa_device64(), e.doc_len()[0], k64_a.get(),
                        M.vals_a.get(), M.d_count.get() + 1, st);
        prims::sort_pairs_u64_u32(e.scratch(), k64_a.get(), k64_b.get(), M.vals_a.get(), M.vals_b.get(), n, 0, 40, st);
        MMT_HIP(hipStreamSynchronize(st));
    } else {
        mk::rank_keys(m.d_offsets.get(), (uint32_t)n, (uint32_t)m.n_docs, e.isa_device(), e.doc_len()[0], key_a.get(),
                      M.vals_a.get(), M.d_count.get() + 1, st);
        prims::sort_pairs_u32_u32(e.scratch(), key_a.get(), key_b.get(), M.vals_a.get(), M.vals_b.get(), n, 0, 32, st);
    }
    DevBuf<uint32_t> len2; DevBuf<int64_t> off2; DevBuf<uint8_t> st2;
    len2.ensure(n + 1); off2.ensure(n * m.n_docs + 1); st2.ensure(n * m.n_docs + 1);
    mk::permute_rows(M.vals_b.get(), (uint32_t)n, (uint32_t)m.n_docs, m.d_length.get(), m.d_offsets.get(),
                     m.d_strands.get(), len2.get(), off2.get(), st2.get(), st);
    uint32_t bad = 0;
    MMT_HIP(hipMemcpyAsync(&bad, M.d_count.get() + 1, 4, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    if (bad) throw std::runtime_error("anchor offset outside the anchor");
    m.d_length.swap(len2); m.d_offsets.swap(off2); m.d_strands.swap(st2);
    m.on_host = false;
}

// .mums bytes of the merged rows in HBM; returns their number
static size_t format_merged_device(Engine& e, const MergedRows& m, DevBuf<char>& text) {
    const size_t n = m.n_rows;
    if (!n) return 0;
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    DevBuf<uint64_t> tlen, toff;
    tlen.ensure(n); toff.ensure(n);
    mk::table_measure(m.d_length.get(), m.d_offsets.get(), (uint32_t)n, (uint32_t)m.n_docs, tlen.get(), st);
    prims::exclusive_sum_u64(e.scratch(), tlen.get(), toff.get(), n, st);
    uint64_t last[2] = {0, 0};
    MMT_HIP(hipMemcpyAsync(&last[0], toff.get() + (n - 1), 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipMemcpyAsync(&last[1], tlen.get() + (n - 1), 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    const size_t bytes = (size_t)(last[0] + last[1]);
    text.ensure(bytes + 1);
    mk::table_write(m.d_length.get(), m.d_offsets.get(), m.d_strands.get(), (uint32_t)n, (uint32_t)m.n_docs, toff.get(), 0,
                    text.get(), st);
    return bytes;
}

std::string format_merged(Engine& e, const MergedRows& m) {
    DevBuf<char> text;
    const size_t bytes = format_merged_device(e, m, text);
    if (!bytes) return std::string();
    std::string out(bytes, '\0');
    MMT_HIP(hipMemcpyAsync(&out[0], text.get(), bytes, hipMemcpyDeviceToHost, e.stream()));
    MMT_HIP(hipStreamSynchronize(e.stream()));
    return out;
}

const char* stage_merged_text(Engine& e, const MergedRows& m, size_t* n_bytes) {
    DevBuf<char> text;
    const size_t bytes = format_merged_device(e, m, text);
    char* host = e.merge_text_staging(bytes + 1);       // (a fresh std::string of 900 MB costs more than the copy into it)
    if (bytes) {
        MMT_HIP(hipMemcpyAsync(host, text.get(), bytes, hipMemcpyDeviceToHost, e.stream()));
        MMT_HIP(hipStreamSynchronize(e.stream()));
    }
    if (n_bytes) *n_bytes = bytes;
    return host;
}

// Up to MERGED_TEXT_ONE_PIECE the bytes are formatted, staged and written as one piece (the C3 stand-in: 0.9 GB).  The
// merged table of 94 whole genomes is 40 million rows x 94 columns = tens of GB of text: it is formatted in pieces of
// whole rows -- row ranges cut where the running byte offset passes a multiple of the piece size --, every piece into one
// of two device buffers, copied to one of two page-locked blocks and written by a helper thread while the next piece is
// formatted (the shape of the text sink of a streamed run, engine.cpp).
static constexpr size_t MERGED_TEXT_ONE_PIECE = (size_t)3 << 30, MERGED_TEXT_PIECE = (size_t)1 << 30;

void write_merged_text(Engine& e, const MergedRows& m, const std::string& path) {
    const size_t n = m.n_rows;
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    size_t piece_bytes = MERGED_TEXT_PIECE;
    if (const char* c = std::getenv("MMT_MERGED_TEXT_PIECE")) piece_bytes = std::max<size_t>(std::strtoull(c, nullptr, 10), 4096);
    // ~ (digits + comma + strand + comma) per cell: an upper bound good enough to choose the route
    const bool one_piece = !std::getenv("MMT_MERGED_TEXT_PIECE") && (double)n * (double)m.n_docs * 24.0 < (double)MERGED_TEXT_ONE_PIECE;
    if (!n || one_piece) {
        size_t bytes = 0;
        const char* host = stage_merged_text(e, m, &bytes);
        write_file_bytes(path, host, bytes);
        return;
    }
    DevBuf<uint64_t> tlen, toff;
    tlen.ensure(n + 1); toff.ensure(n + 1);
    mk::table_measure(m.d_length.get(), m.d_offsets.get(), (uint32_t)n, (uint32_t)m.n_docs, tlen.get(), st);
    MMT_HIP(hipMemsetAsync(tlen.get() + n, 0, 8, st));
    prims::exclusive_sum_u64(e.scratch(), tlen.get(), toff.get(), n + 1, st);     // toff[n] = all bytes
    std::vector<uint64_t> h_off(n + 1);
    MMT_HIP(hipMemcpyAsync(h_off.data(), toff.get(), (n + 1) * 8, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    tlen.release();
    // pieces of whole rows, each at most piece_bytes (a single row longer than that is a piece of its own)
    std::vector<size_t> cut(1, 0);
    uint64_t longest = 0;
    for (size_t r = 0; r < n;) {
        const uint64_t lim = h_off[r] + piece_bytes;
        size_t q = (size_t)(std::upper_bound(h_off.begin() + r + 1, h_off.end(), lim) - h_off.begin()) - 1;
        if (q <= r) q = r + 1;
        longest = std::max<uint64_t>(longest, h_off[q] - h_off[r]);
        cut.push_back(q);
        r = q;
    }
    DevBuf<char> d_piece[2];
    PinnedBuf<char> h_piece[2];
    for (int b = 0; b < 2; b++) { d_piece[b].ensure(longest + 1); h_piece[b].ensure(longest + 1); }
    // the bytes go to PATH.tmp and take the final name when every piece is in (a short write, a full disk or a kill half way
    // must not leave a plausible but truncated PREFIX.mums); a device file -- /dev/null: runs that only time the formatting --
    // is written as it is
    struct stat sb;
    const bool special = ::stat(path.c_str(), &sb) == 0 && !S_ISREG(sb.st_mode);
    const std::string tmp = special ? path : path + ".tmp";
    const int fd = ::open(tmp.c_str(), special ? O_WRONLY : (O_CREAT | O_TRUNC | O_WRONLY), 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + tmp);
    // the writer: piece i of buffer i & 1 once its copy has landed
    struct Job { int buf; size_t bytes; };
    std::mutex mu; std::condition_variable cv; std::deque<Job> q; bool closing = false; int free_buf[2] = {1, 1};
    std::string error;
    std::thread writer([&]() {
        for (;;) {
            Job j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty() || closing; });
              if (q.empty()) return;
              j = q.front(); q.pop_front(); }
            const char* src = h_piece[j.buf].get();
            size_t at = 0;
            auto failed = [&]() { std::lock_guard<std::mutex> lk(mu); return !error.empty(); };     // (read and written under the mutex)
            while (at < j.bytes && !failed()) {
                const ssize_t w = ::write(fd, src + at, j.bytes - at);
                if (w <= 0) { std::lock_guard<std::mutex> lk(mu); error = "short write to " + tmp; break; }
                at += (size_t)w;
            }
            { std::lock_guard<std::mutex> lk(mu); free_buf[j.buf] = 1; }
            cv.notify_all();
        }
    });
    try {
        for (size_t i = 0; i + 1 < cut.size(); i++) {
            const int b = (int)(i & 1);
            const size_t r0 = cut[i], r1 = cut[i + 1];
            const uint64_t bytes = h_off[r1] - h_off[r0];
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return free_buf[b] == 1; }); free_buf[b] = 0;
              if (!error.empty()) throw std::runtime_error(error); }
            mk::table_write(m.d_length.get() + r0, m.d_offsets.get() + r0 * m.n_docs, m.d_strands.get() + r0 * m.n_docs,
                            (uint32_t)(r1 - r0), (uint32_t)m.n_docs, toff.get() + r0, h_off[r0], d_piece[b].get(), st);
            MMT_HIP(hipMemcpyAsync(h_piece[b].get(), d_piece[b].get(), bytes, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
            { std::lock_guard<std::mutex> lk(mu); q.push_back(Job{b, (size_t)bytes}); }
            cv.notify_all();
        }
    } catch (...) {
        { std::lock_guard<std::mutex> lk(mu); closing = true; }
        cv.notify_all(); writer.join(); ::close(fd);
        if (!special) ::unlink(tmp.c_str());
        throw;
    }
    { std::lock_guard<std::mutex> lk(mu); closing = true; }
    cv.notify_all();
    writer.join();
    std::string failure;
    { std::lock_guard<std::mutex> lk(mu); failure = error; }
    if (::close(fd) != 0 && failure.empty()) failure = "cannot close " + tmp;
    if (failure.empty() && !special && std::rename(tmp.c_str(), path.c_str()) != 0) failure = "cannot rename " + tmp;
    if (!failure.empty()) { if (!special) ::unlink(tmp.c_str()); throw std::runtime_error(failure); }
}

void download_merged(Engine& e, MergedRows& m) {
    if (m.on_host) return;
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    const size_t n = m.n_rows, cells = n * m.n_docs;
    m.length.resize(n); m.offsets.resize(cells); m.strands.resize(cells); m.thresh.resize(m.thresh_len);
    if (n) {
        MMT_HIP(hipMemcpyAsync(m.length.data(), m.d_length.get(), n * 4, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(m.offsets.data(), m.d_offsets.get(), cells * 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(m.strands.data(), m.d_strands.get(), cells, hipMemcpyDeviceToHost, st));
    }
    if (m.thresh_len) {         // the host copy is the PREFIX.athresh form: 16 bits, saturated (mem_finder.hpp:299)
        DevBuf<uint16_t> narrow;
        narrow.ensure(m.thresh_len);
        k::thresh_narrow(m.d_thresh.get(), m.thresh_len, narrow.get(), st);
        MMT_HIP(hipMemcpyAsync(m.thresh.data(), narrow.get(), m.thresh_len * 2, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
    }
    MMT_HIP(hipStreamSynchronize(st));
    m.on_host = true;
}

}  // namespace mmt
