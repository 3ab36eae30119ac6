// merge.cpp -- see merge.hpp.  Host side keeps the row tables (rows x docs);
// the device does everything that is proportional to the anchor length.
#include "merge.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

#include "prims.hpp"

namespace mmt {
namespace {

// One side of a fold step, in anchor order.  Rows are kept lazily: a merged row is (anchor start,
// length) plus, for every source partition g folded so far, the source row and the accumulated
// shifts of its '+' and '-' columns (fix_neg_strand, merge_candidates.cpp:97-104, applied once per
// fold step: '+' offsets move by the trim at the front, '-' offsets by the trim at the back).
// The columns themselves are materialised once, for the rows that survive every fold.
struct Side {
    size_t n_parts = 0;                 // partitions folded into this side
    std::vector<uint64_t> start;        // anchor offset of row i
    std::vector<uint32_t> len;
    std::vector<uint32_t> src;          // n_rows * n_parts: row index inside partition g
    std::vector<int64_t> plus, minus;   // n_rows * n_parts accumulated shifts
    size_t n_rows() const { return start.size(); }
};

// parse_candidate(): rows in anchor order (merge_candidates.cpp:89-92)
Side leaf_side(const mmt_partition& p, uint64_t L) {
    Side s;
    s.n_parts = 1;
    const size_t n = p.n_rows;
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++) sorted = p.offsets[(i - 1) * p.n_docs] <= p.offsets[i * p.n_docs];
    if (!sorted)
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return p.offsets[(size_t)a * p.n_docs] < p.offsets[(size_t)b * p.n_docs];
        });
    s.start.resize(n); s.len.resize(n); s.src.resize(n); s.plus.assign(n, 0); s.minus.assign(n, 0);
    for (size_t i = 0; i < n; i++) {
        const int64_t o = p.offsets[(size_t)order[i] * p.n_docs];
        if (o < 0 || (uint64_t)o >= L) throw std::runtime_error("anchor offset outside the threshold array");
        s.start[i] = (uint64_t)o; s.len[i] = p.length[order[i]]; s.src[i] = order[i];
    }
    return s;
}

struct DeviceSide {
    DevBuf<uint64_t> start;
    DevBuf<uint32_t> len, ones, rank;
    DevBuf<uint8_t> bv;
    void upload(const Side& s, uint64_t L, DevBuf<uint8_t>& temp, hipStream_t st) {
        const size_t n = s.n_rows();
        start.ensure(n + 1); len.ensure(n + 1); bv.ensure(L); ones.ensure(L); rank.ensure(L);
        if (n) {
            MMT_HIP(hipMemcpyAsync(start.get(), s.start.data(), n * 8, hipMemcpyHostToDevice, st));
            MMT_HIP(hipMemcpyAsync(len.get(), s.len.data(), n * 4, hipMemcpyHostToDevice, st));
        }
        MMT_HIP(hipMemsetAsync(bv.get(), 0, L, st));
        MMT_HIP(hipMemsetAsync(ones.get(), 0, L * 4, st));
        k::mark_starts(start.get(), (uint32_t)n, bv.get(), ones.get(), st);
        prims::exclusive_sum_u32(temp, ones.get(), rank.get(), L, st);
    }
};

// device scratch of the fold, kept between calls (one merge per bench step on rank 0)
struct MergeScratch {
    DevBuf<uint16_t> nb_left, nb_right, nb_out;
    DeviceSide da, db;
    DevBuf<uint64_t> d_pos;
    DevBuf<uint32_t> d_ra, d_rb, d_len, d_count;
};
MergeScratch& scratch() {
    static thread_local MergeScratch s;
    return s;
}

}  // namespace

MergedRows anchor_merge(Engine& e, const mmt_partition* parts, size_t k, uint32_t min_len) {
    const bool dbg = std::getenv("MMT_MERGE_DEBUG") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[merge] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    const uint64_t L = parts[0].thresh_len;
    for (size_t i = 0; i < k; i++) {
        if (parts[i].thresh_len != L) throw std::runtime_error("partitions disagree on the anchor length");
        if (parts[i].n_rows >= 0xffffffffull) throw std::runtime_error("too many rows in a partition");
    }
    MergeScratch& M = scratch();
    auto load_thresh = [&](const mmt_partition& p, DevBuf<uint16_t>& dst) {
        dst.ensure(L);
        MMT_HIP(hipMemcpyAsync(dst.get(), p.thresh, L * 2,
                               p.thresh_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    };
    Side left = leaf_side(parts[0], L);
    load_thresh(parts[0], M.nb_left);
    M.d_count.ensure(4);
    lap("view part 0");
    for (size_t pi = 1; pi < k; pi++) {
        Side right = leaf_side(parts[pi], L);
        load_thresh(parts[pi], M.nb_right);
        lap("view right");
        M.nb_out.ensure(L);
        M.da.upload(left, L, e.scratch(), st);
        M.db.upload(right, L, e.scratch(), st);
        lap("upload sides");
        const size_t capacity = left.n_rows() + right.n_rows() + 1;
        M.d_pos.ensure(capacity); M.d_ra.ensure(capacity); M.d_rb.ensure(capacity); M.d_len.ensure(capacity);
        MMT_HIP(hipMemsetAsync(M.d_count.get(), 0, 16, st));
        k::FoldArgs a;
        a.len = L; a.nb_a = M.nb_left.get(); a.nb_b = M.nb_right.get(); a.nb_out = M.nb_out.get();
        a.rank_a = M.da.rank.get(); a.rank_b = M.db.rank.get();
        a.start_a = M.da.start.get(); a.start_b = M.db.start.get();
        a.len_a = M.da.len.get(); a.len_b = M.db.len.get();
        a.bv_a = M.da.bv.get(); a.bv_b = M.db.bv.get();
        a.out_pos = M.d_pos.get(); a.out_ra = M.d_ra.get(); a.out_rb = M.d_rb.get(); a.out_len = M.d_len.get();
        a.capacity = (uint32_t)capacity; a.d_count = M.d_count.get(); a.min_len = min_len;
        k::fold_step(a, st);
        uint32_t found = 0;
        MMT_HIP(hipMemcpyAsync(&found, M.d_count.get(), 4, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (found > capacity) throw std::runtime_error("anchor merge produced more rows than MUM starts");
        lap("fold kernel");
        std::vector<uint64_t> h_pos; std::vector<uint32_t> h_ra, h_rb, h_len;
        d2h(h_pos, M.d_pos.get(), found, st); d2h(h_ra, M.d_ra.get(), found, st);
        d2h(h_rb, M.d_rb.get(), found, st); d2h(h_len, M.d_len.get(), found, st);
        std::vector<uint32_t> order(found);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return h_pos[x] < h_pos[y]; });
        // new rows, lazily: trims of this step added to the shifts of every source partition (:97-104, :142-151)
        Side out;
        const size_t np = left.n_parts + 1;
        out.n_parts = np;
        out.start.resize(found); out.len.resize(found);
        out.src.resize((size_t)found * np); out.plus.resize((size_t)found * np); out.minus.resize((size_t)found * np);
        for (uint32_t q = 0; q < found; q++) {
            const uint32_t t = order[q], nl = h_len[t], ra = h_ra[t], rb = h_rb[t];
            const int64_t i = (int64_t)h_pos[t];
            const int64_t d1 = i - (int64_t)left.start[ra], d2 = i - (int64_t)right.start[rb];
            const int64_t s1 = (int64_t)left.len[ra] - d1, s2 = (int64_t)right.len[rb] - d2;
            out.start[q] = (uint64_t)i; out.len[q] = nl;
            for (size_t g = 0; g < left.n_parts; g++) {
                out.src[q * np + g] = left.src[ra * left.n_parts + g];
                out.plus[q * np + g] = left.plus[ra * left.n_parts + g] + d1;
                out.minus[q * np + g] = left.minus[ra * left.n_parts + g] + (s1 - (int64_t)nl);
            }
            out.src[q * np + np - 1] = right.src[rb];
            out.plus[q * np + np - 1] = d2;
            out.minus[q * np + np - 1] = s2 - (int64_t)nl;
        }
        left = std::move(out);
        M.nb_left.swap(M.nb_out);
        lap("host rows");
    }
    // materialise the surviving rows: partition 0's columns, then every other partition's without its anchor
    MergedRows m;
    m.n_docs = 0;
    for (size_t g = 0; g < k; g++) m.n_docs += parts[g].n_docs - (g ? 1 : 0);
    const size_t n = left.n_rows();
    m.length.resize(n); m.offsets.resize(n * m.n_docs); m.strands.resize(n * m.n_docs);
    for (size_t i = 0; i < n; i++) {
        m.length[i] = left.len[i];
        size_t col = 0;
        for (size_t g = 0; g < left.n_parts; g++) {
            const mmt_partition& P = parts[g];
            const size_t r = left.src[i * left.n_parts + g];
            const int64_t ps = left.plus[i * left.n_parts + g], ms = left.minus[i * left.n_parts + g];
            for (size_t c = g ? 1 : 0; c < P.n_docs; c++, col++) {
                const uint8_t sd = P.strands[r * P.n_docs + c];
                m.offsets[i * m.n_docs + col] = P.offsets[r * P.n_docs + c] + (sd ? ps : ms);
                m.strands[i * m.n_docs + col] = sd;
            }
        }
    }
    d2h(m.thresh, M.nb_left.get(), L, st);
    lap("final copy");
    return m;
}

void sort_like_direct(Engine& e, MergedRows& m) {
    const size_t n = m.length.size();
    if (!n) return;
    if (e.text_length() == 0) throw std::runtime_error("engine holds no suffix ranks: run it on a partition first");
    hipStream_t st = e.stream();
    MMT_HIP(hipSetDevice(e.device()));
    // anchor = document 0 of the engine's text, '+' strand starts at text offset 0
    std::vector<uint64_t> h_idx(n);
    for (size_t r = 0; r < n; r++) {
        int64_t o = m.offsets[r * m.n_docs];
        if (o < 0 || (uint64_t)o >= e.doc_len()[0]) throw std::runtime_error("anchor offset outside the anchor");
        h_idx[r] = (uint64_t)o;
    }
    DevBuf<uint64_t> d_idx; DevBuf<uint32_t> d_key;
    d_idx.ensure(n); d_key.ensure(n);
    MMT_HIP(hipMemcpyAsync(d_idx.get(), h_idx.data(), n * 8, hipMemcpyHostToDevice, st));
    k::gather_u32(e.isa_device(), d_idx.get(), (uint32_t)n, d_key.get(), st);
    std::vector<uint32_t> key;
    d2h(key, d_key.get(), n, st);
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
    MergedRows o;
    o.n_docs = m.n_docs; o.thresh = std::move(m.thresh);
    o.length.resize(n); o.offsets.resize(n * m.n_docs); o.strands.resize(n * m.n_docs);
    for (size_t i = 0; i < n; i++) {
        size_t r = order[i];
        o.length[i] = m.length[r];
        std::copy_n(&m.offsets[r * m.n_docs], m.n_docs, &o.offsets[i * m.n_docs]);
        std::copy_n(&m.strands[r * m.n_docs], m.n_docs, &o.strands[i * m.n_docs]);
    }
    m = std::move(o);
}

std::string format_merged(const MergedRows& m) {
    std::string t;
    const size_t n = m.length.size();
    for (size_t r = 0; r < n; r++) {
        append_uint(t, m.length[r]); t.push_back('\t');
        for (size_t c = 0; c < m.n_docs; c++) {
            int64_t o = m.offsets[r * m.n_docs + c];
            if (o < 0) { t.push_back('-'); append_uint(t, (uint64_t)(-o)); } else append_uint(t, (uint64_t)o);
            if (c + 1 < m.n_docs) t.push_back(',');
        }
        t.push_back('\t');
        for (size_t c = 0; c < m.n_docs; c++) {
            t.push_back(m.strands[r * m.n_docs + c] ? '+' : '-');
            if (c + 1 < m.n_docs) t.push_back(',');
        }
        t.push_back('\n');
    }
    return t;
}

}  // namespace mmt
