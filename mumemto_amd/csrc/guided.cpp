// guided.cpp -- Engine::suffix_sort_guided: suffix array + BWT of a text whose prefix-free parse has a dictionary too
// large to be suffix-sorted (guided_kernels.hip).  Same result as suffix_sort_pfp / the reference's pfp_lcp
// (pfp_lcp_mum.hpp:115-231): the text suffixes ordered by (phrase suffix, rank of the following parse suffix).
//
//   pfp_parse (pfp.cpp)           phrases, distinct phrases (verified fingerprints), phrase id per parse position
//   phrase ranks                  the distinct phrases sorted as strings: one batch of the rounds below
//   parse                         32-bit doubling sort of the rank sequence -> rank of every parse suffix
//   batches of text suffixes      by leading characters; radix sort on 63 bits of characters, refinement rounds up to
//                                 the phrase end, parse ranks beyond it; suffix-array and BWT columns written in place
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine.hpp"
#include "guided_kernels.hpp"
#include "pfp_kernels.hpp"
#include "pool.hpp"
#include "prims.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

static uint32_t read_u32(const uint32_t* d, hipStream_t s) {
    uint32_t v = 0;
    MMT_HIP(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    return v;
}

// MMT_MEM_TRACE=1: live / peak bytes of the device heap at the stage boundaries (stderr)
static void mem_mark(int device, const char* what) {
    static const bool on = std::getenv("MMT_MEM_TRACE") != nullptr;
    if (!on) return;
    const pool::Stats s = pool::stats(device);
    std::fprintf(stderr, "[mem] %-28s live %7.2f GB  peak %7.2f GB  mapped %7.2f GB\n", what, s.live / 1e9, s.peak / 1e9, s.mapped / 1e9);
}

namespace {

// Karp-Rabin hash of a window of w equal bytes (newscan.hpp:84-114: prime 1999999973, base 256)
uint64_t kr_window_of_run(uint8_t c, uint32_t w) {
    uint64_t h = 0;
    for (uint32_t i = 0; i < w; i++) h = (h * 256 + c) % 1999999973ull;
    return h;
}

// scratch of one batch (capacity elements)
struct Batch {
    DevBuf<uint64_t> key_a, key_b, pos_a, pos_b, pos_c;
    DevBuf<uint32_t> slot_a, slot_b, ghead, hv, idx, bound, big_begin, big_end, count, seg;
    DevBuf<uint8_t> head, flags;
    uint32_t cap = 0;
    void reserve(uint32_t n) {
        cap = n;
        for (DevBuf<uint64_t>* b : {&key_a, &key_b, &pos_a, &pos_b, &pos_c}) b->ensure(n);
        for (DevBuf<uint32_t>* b : {&slot_a, &slot_b, &ghead, &hv, &idx}) b->ensure(n);
        head.ensure(n); flags.ensure(n);
        bound.ensure((size_t)n / 1024 + 4); big_begin.ensure(4096); big_end.ensure(4096); count.ensure(8);
    }
    static size_t bytes_per_element() { return 5 * 8 + 5 * 4 + 2 + 1; }
};

struct RoundStats { int rounds = 0; uint64_t active_sum = 0; uint32_t first_active = 0, small = 0; };

// (key_a, pos_a) hold B elements with their first keys: sorts them completely; the sorted element records end up in
// B.pos_b (suffix-array order of the batch).
// lcp_out (optional, B entries, indexed like pos_b): preset to 0xffffffff here; wherever the sort separates an element
// from its predecessor by a key of known depth or by a parse rank it leaves their LCP (gk::batch_lcp computes the rest).
RoundStats sort_batch(Batch& X, uint32_t B, const gk::Ctx& ctx, DevBuf<uint8_t>& temp, uint32_t* err, hipStream_t st,
                      uint32_t* lcp_out = nullptr, const RmqView* rmq = nullptr) {
    RoundStats rs;
    if (B == 0) return rs;
    if (lcp_out) MMT_HIP(hipMemsetAsync(lcp_out, 0xff, (size_t)B * 4, st));
    const bool in_b = prims::sort_pairs_u64_u64_inplace(temp, X.key_a.get(), X.key_b.get(), X.pos_a.get(), X.pos_b.get(), B, 0,
                                                        ctx.bits * ctx.chars, st);
    if (!in_b) { X.key_a.swap(X.key_b); X.pos_a.swap(X.pos_b); }          // from here on: sorted pairs in (key_b, pos_b)
    gk::heads0(X.key_b.get(), B, X.head.get(), X.flags.get(), lcp_out, ctx.bits, ctx.chars, st);
    prims::select_indices(temp, X.flags.get(), X.idx.get(), X.count.get(), B, st);
    uint32_t m = read_u32(X.count.get(), st);
    rs.first_active = m;
    if (!m) return rs;
    gk::gather_active(X.idx.get(), m, X.pos_b.get(), X.head.get(), X.slot_a.get(), X.pos_a.get(), X.ghead.get(), st);
    prims::inclusive_max_u32(temp, X.ghead.get(), X.ghead.get(), m, st);
    uint64_t offset = (uint64_t)ctx.chars;
    const uint32_t target = 1024, limit = gk::SORT_CAP - target;
    if (!std::getenv("MMT_GUIDED_NO_SMALL")) {                      // (the variable sends every group through the rounds: tests)
        gk::resolve_small(ctx, X.pos_a.get(), X.ghead.get(), X.slot_a.get(), m, offset, X.pos_b.get(), X.flags.get(), err, st,
                          std::getenv("MMT_GUIDED_NO_SMALL_LCP") ? nullptr : lcp_out);
        prims::select_indices(temp, X.flags.get(), X.idx.get(), X.count.get(), m, st);
        const uint32_t m2 = read_u32(X.count.get(), st);
        if (m2 && m2 < m) {
            gk::round_compact(X.idx.get(), m2, X.slot_a.get(), X.pos_a.get(), X.ghead.get(), X.slot_b.get(), X.pos_c.get(),
                              X.hv.get(), st);
            prims::inclusive_max_u32(temp, X.hv.get(), X.ghead.get(), m2, st);
            X.slot_a.swap(X.slot_b); X.pos_a.swap(X.pos_c);
        }
        rs.small = m - m2;
        m = m2;
        // ... and the groups of up to 128 elements (the copies of a position in 9 .. 128 haplotypes), one wave per group
        if (m && !std::getenv("MMT_GUIDED_NO_MEDIUM")) {
            // lists of (first element, size) pairs, one per size class, in key_b (dead since the first sort): m / 9 pairs at most
            const uint32_t lcap = m / 9 + 64;
            gk::medium_groups(X.ghead.get(), m, X.key_b.get(), lcap, X.count.get() + 4, st);
            uint32_t ng4[4];
            MMT_HIP(hipMemcpyAsync(ng4, X.count.get() + 4, 16, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
            const uint32_t ng = ng4[0] + ng4[1] + ng4[2] + ng4[3];
            if (ng) {
                MMT_HIP(hipMemsetAsync(X.flags.get(), 1, m, st));
                gk::resolve_medium(ctx, rmq ? *rmq : RmqView(), X.pos_a.get(), X.slot_a.get(), X.key_b.get(), lcap, ng4, offset,
                                   X.pos_b.get(), X.flags.get(), rmq ? lcp_out : nullptr, err, st);
                prims::select_indices(temp, X.flags.get(), X.idx.get(), X.count.get(), m, st);
                const uint32_t m3 = read_u32(X.count.get(), st);
                if (m3 && m3 < m) {
                    gk::round_compact(X.idx.get(), m3, X.slot_a.get(), X.pos_a.get(), X.ghead.get(), X.slot_b.get(), X.pos_c.get(),
                                      X.hv.get(), st);
                    prims::inclusive_max_u32(temp, X.hv.get(), X.ghead.get(), m3, st);
                    X.slot_a.swap(X.slot_b); X.pos_a.swap(X.pos_c);
                }
                rs.small += m - m3;
                m = m3;
            }
        }
    }
    const bool no_early_giant = std::getenv("MMT_GUIDED_NO_EARLY_GIANT") != nullptr;
    const bool no_tail = std::getenv("MMT_GUIDED_NO_TAIL") != nullptr;
    bool tail_done = false;
    while (m) {
        if (++rs.rounds > (1 << 22)) throw std::runtime_error("parse-guided suffix sort did not converge");
        rs.active_sum += m;
        // (expansion: groups whose members all lie in giant phrases take their order from the giant dictionary at once --
        // guided_kernels.hip k_giant_probe; idx and flags are dead until this round's heads are known)
        const bool early_giant = ctx.expand && ctx.g_n && offset < ctx.g_depth && !no_early_giant;
        if (early_giant) {
            gk::giant_probe(ctx, X.pos_a.get(), X.ghead.get(), m, offset, X.idx.get(), X.flags.get(), st);
            gk::round_keys(ctx, X.pos_a.get(), m, offset, X.key_a.get(), err, st, X.idx.get(), X.flags.get(), X.ghead.get());
        } else
            gk::round_keys(ctx, X.pos_a.get(), m, offset, X.key_a.get(), err, st);
        // sort inside the groups: (key_a, pos_a) -> (key_b, pos_c)
        const uint32_t n_tiles = (m + target - 1) / target;
        X.bound.ensure((size_t)n_tiles + 2);
        MMT_HIP(hipMemsetAsync(X.count.get() + 1, 0, 4, st));
        gk::tile_bounds(X.ghead.get(), m, target, limit, n_tiles, X.bound.get(), st);
        uint32_t big_cap = (uint32_t)X.big_begin.size();
        gk::local_sort(X.key_a.get(), X.pos_a.get(), X.ghead.get(), X.key_b.get(), X.pos_c.get(), X.bound.get(), n_tiles,
                       X.big_begin.get(), X.big_end.get(), X.count.get() + 1, big_cap, st);
        uint32_t big = read_u32(X.count.get() + 1, st);
        if (big > big_cap) {                                         // (rare) list too short: run the tile sort again
            X.big_begin.ensure(big); X.big_end.ensure(big);
            big_cap = big;
            MMT_HIP(hipMemsetAsync(X.count.get() + 1, 0, 4, st));
            gk::local_sort(X.key_a.get(), X.pos_a.get(), X.ghead.get(), X.key_b.get(), X.pos_c.get(), X.bound.get(), n_tiles,
                           X.big_begin.get(), X.big_end.get(), X.count.get() + 1, big_cap, st);
            big = read_u32(X.count.get() + 1, st);
        }
        if (big) {
            // ranges that hold a group longer than an LDS tile: their groups become the segments of ONE segmented sort
            // (a collection with satellite arrays has thousands of such ranges per round)
            std::vector<uint32_t> hb(big), he(big);
            MMT_HIP(hipMemcpyAsync(hb.data(), X.big_begin.get(), (size_t)big * 4, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(he.data(), X.big_end.get(), (size_t)big * 4, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
            uint64_t covered = 0;
            for (uint32_t r = 0; r < big; r++) covered += he[r] - hb[r];
            const size_t cap = (size_t)(covered / 2 + 2);                  // every group that is still active has two elements
            X.seg.ensure(2 * cap);
            gk::range_groups(X.ghead.get(), X.big_begin.get(), X.big_end.get(), big, X.seg.get(), X.seg.get() + cap,
                             X.count.get() + 2, st);
            const uint32_t segs = read_u32(X.count.get() + 2, st);
            if (segs > cap) throw std::runtime_error("guided sort: more groups than elements in the long ranges");
            prims::segmented_sort_pairs_u64_u64vals_ranges(temp, X.key_a.get(), X.key_b.get(), X.pos_a.get(), X.pos_c.get(), m,
                                                           segs, X.seg.get(), X.seg.get() + cap, 64, st);
        }
        gk::round_heads(ctx, X.key_b.get(), X.ghead.get(), m, X.hv.get(), err, st, std::getenv("MMT_GUIDED_NO_ROUND_LCP") ? nullptr : lcp_out,
                        X.slot_a.get(), offset, early_giant ? X.flags.get() : nullptr);
        prims::inclusive_max_u32(temp, X.hv.get(), X.hv.get(), m, st);
        gk::round_apply(X.pos_c.get(), X.hv.get(), X.slot_a.get(), m, X.pos_b.get(), X.flags.get(), st);
        prims::select_indices(temp, X.flags.get(), X.idx.get(), X.count.get(), m, st);
        const uint32_t m2 = read_u32(X.count.get(), st);
        if (m2) {
            gk::round_compact(X.idx.get(), m2, X.slot_a.get(), X.pos_c.get(), X.hv.get(), X.slot_b.get(), X.pos_a.get(),
                              X.ghead.get(), st);
            prims::inclusive_max_u32(temp, X.ghead.get(), X.ghead.get(), m2, st);
            X.slot_a.swap(X.slot_b);
        }
        m = m2;
        // (after the round that took its keys from the giant dictionary every alpha that is still tied is spent)
        offset = ctx.g_n && offset >= ctx.g_depth ? (1ull << 40) : offset + (uint64_t)ctx.chars;
        // The tail: a few thousand elements -- the rests of giant phrases the group kernels left undecided -- used to walk on 21
        // characters a round, a dozen launches and three round trips to the host each: 148 rounds a batch on a rank's share of
        // configs[4], ~6 of its 61 s for 0.003 element-rounds per representative.  Their groups are small: once, when the active
        // set has shrunk to that, every group of up to 1024 is finished by comparison (the giant dictionary answers what is long).
        if (m && m <= 32768 && !tail_done && offset < (1ull << 40) && !no_tail) {
            tail_done = true;
            gk::resolve_small(ctx, X.pos_a.get(), X.ghead.get(), X.slot_a.get(), m, offset, X.pos_b.get(), X.flags.get(), err, st,
                              std::getenv("MMT_GUIDED_NO_SMALL_LCP") ? nullptr : lcp_out, 1024);
            prims::select_indices(temp, X.flags.get(), X.idx.get(), X.count.get(), m, st);
            const uint32_t m3 = read_u32(X.count.get(), st);
            if (m3 && m3 < m) {
                gk::round_compact(X.idx.get(), m3, X.slot_a.get(), X.pos_a.get(), X.ghead.get(), X.slot_b.get(), X.pos_c.get(),
                                  X.hv.get(), st);
                prims::inclusive_max_u32(temp, X.hv.get(), X.ghead.get(), m3, st);
                X.slot_a.swap(X.slot_b); X.pos_a.swap(X.pos_c);
            }
            m = m3;
        }
    }
    return rs;
}

}  // namespace

// Tables of the bucket-wise producer: symbol codes, rank / successor structure over the phrase ends, ranks of the distinct
// phrases, the parse's suffix array with the LCP of adjacent parse suffixes, histogram of the suffixes' leading
// characters.  The suffixes themselves are sorted batch by batch while the scan consumes them (guided_stream).
void Engine::guided_prepare() {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    const bool W = wide_;
    hipStream_t st = stream_;
    const uint32_t w = S.w, m = S.n_phrases, D = S.n_distinct;
    const bool stats = std::getenv("MMT_GUIDED_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        MMT_HIP(hipStreamSynchronize(st));
        return std::chrono::duration<double, std::milli>(now() - t).count();
    };
    EventPair e3, e5;
    auto t_mark = now();
    auto mark = [&](const char* what) {               // (MMT_GUIDED_STATS: where the preparation's time goes)
        if (!stats) return;
        std::fprintf(stderr, "[guided] prepare: %s %.1f ms\n", what, ms_since(t_mark));
        t_mark = now();
    };

    // ---- symbol codes (ascending with the byte value; Dollar is the smallest symbol of V) ----
    e3.start(st);
    std::vector<uint64_t> hist;
    d2h(hist, d_hist_.get(), 256, st);
    uint8_t code[256];
    int sigma = 0;
    for (int c = 0; c < 256; c++) code[c] = (hist[c] || c == 2) ? (uint8_t)(++sigma) : 0;
    gk::Ctx& ctx = S.gctx;
    ctx = gk::Ctx{};
    ctx.bits = std::max(1, bit_width_u64((uint64_t)sigma));
    ctx.chars = std::min(63 / ctx.bits, 63);
    // Bins of leading characters: every interval the scan can report (LCP value >= the minimum match length) lies inside
    // one bin as long as a bin's prefix is not longer than that length -- SURVEY.md 8(e) -- so that the pieces of the
    // stream can be produced, scanned and dropped bin by bin, on any rank.
    int prefix_chars = std::max(1, std::min(12 / ctx.bits, ctx.chars));
    // (a bin must fit one batch of at most 2^30 suffixes: beyond 2^38 characters -- four leading bases hold n / 256 of the
    // suffixes of a DNA text -- the bins take one character more; MMT_GUIDED_PREFIX: tests)
    if (n >= (1ull << 38)) prefix_chars = std::max(prefix_chars, std::min(15 / ctx.bits, ctx.chars));
    if (const char* e = std::getenv("MMT_GUIDED_PREFIX")) prefix_chars = std::max(1, std::min(std::atoi(e), std::min(18 / ctx.bits, ctx.chars)));
    prefix_chars = std::max(1, std::min<int>(prefix_chars, (int)std::min<uint32_t>(stream_min_len_, 64u)));
    // (expansion: every occurrence of a phrase suffix must lie in the bin of its representative -- a phrase suffix has at
    // least w characters, so a bin's prefix must not be longer than that)
    if (S.expand) prefix_chars = std::max(1, std::min<int>(prefix_chars, (int)w));
    S.g_prefix = prefix_chars;
    d_code_.ensure(256);
    MMT_HIP(hipMemcpyAsync(d_code_.get(), code, 256, hipMemcpyHostToDevice, st));
    std::memcpy(S.g_code, code, 256); S.g_bits = ctx.bits; S.g_share_valid = false;
    ctx.T = text_ref(); ctx.n = n; ctx.w = w; ctx.code = d_code_.get(); ctx.m = m;
    for (int k = 0; k < 4; k++) ctx.acgt[k] = code[(uint8_t)"ACGT"[k]];
    ctx.acgt_lut = (uint32_t)ctx.acgt[0] | ((uint32_t)ctx.acgt[1] << 8) | ((uint32_t)ctx.acgt[2] << 16) | ((uint32_t)ctx.acgt[3] << 24);
    ctx.dense_ok = ctx.acgt[0] && ctx.acgt[1] && ctx.acgt[2] && ctx.acgt[3] && ctx.chars <= 32 && !std::getenv("MMT_GUIDED_NO_DENSE");

    // ---- phrase ends: rank directory and successor table over the cut bits ----
    const uint64_t n_words = S.tmask.size() * sizeof(uint16_t) / 8 / 64 * 64;     // whole blocks of 4096 positions
    const uint64_t n_blocks = n_words / 64, n_counts = n_words / 8;
    if (n_blocks < n / 4096 + 2) throw std::runtime_error("cut bit vector too short for the guided sort");
    const uint64_t* mask = reinterpret_cast<const uint64_t*>(S.tmask.get());
    // A packed text is a text that fills the device: the phrase ends are kept as a list then (two bytes per phrase + four per
    // block of 4096 positions) and the bit per position -- 72 GB on 573 G characters, with 9 GB of rank directory -- goes
    // (MMT_CUT_LIST=0 / 1 overrides: the tests run both forms)
    // (... and only then: a rank query on the bits is two independent lines -- directory word + the 512 positions' words --,
    // on the list a search of two or three dependent round trips; a rank's share of configs[3], 79 G characters packed to
    // 20 GB, keeps its 10 GB of bits)
    const bool cut_list = std::getenv("MMT_CUT_LIST") ? std::atoi(std::getenv("MMT_CUT_LIST")) != 0 : (packed_ && n >= (1ull << 37));
    if (cut_list) {
        DevBuf<uint32_t> bcount;
        bcount.ensure(n_blocks + 2); S.g_brank.ensure(n_blocks + 2);
        MMT_HIP(hipMemsetAsync(bcount.get(), 0, (n_blocks + 2) * 4, st));
        gk::block_cut_counts(mask, n_words, bcount.get(), n_blocks, st);
        prims::exclusive_sum_u32(d_temp_, bcount.get(), S.g_brank.get(), n_blocks + 2, st);
        S.g_coff.ensure((size_t)m + 64);
        gk::block_cut_offsets(mask, n_words, S.g_brank.get(), S.g_coff.get(), n_blocks, st);
        MMT_HIP(hipStreamSynchronize(st));
    } else {
        DevBuf<uint32_t> rcount;
        rcount.ensure(n_counts + 1); S.g_rdir.ensure(n_counts + 1);
        gk::rank_counts(mask, n_words, rcount.get(), n_counts, st);
        prims::exclusive_sum_u32(d_temp_, rcount.get(), S.g_rdir.get(), n_counts, st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    {
        S.g_nxt.ensure(n_blocks + 1);
        gk::block_first_cut(mask, n_words, S.g_nxt.get(), n_blocks, st);
        std::vector<uint64_t> h(n_blocks + 1);
        MMT_HIP(hipMemcpyAsync(h.data(), S.g_nxt.get(), n_blocks * 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        h[n_blocks] = n + w - 1;
        for (uint64_t b = n_blocks; b-- > 0;) if (h[b] == ~0ull || h[b] >= n) h[b] = h[b + 1];
        MMT_HIP(hipMemcpyAsync(S.g_nxt.get(), h.data(), (n_blocks + 1) * 8, hipMemcpyHostToDevice, st));
        MMT_HIP(hipStreamSynchronize(st));
    }
    ctx.mask = mask; ctx.rdir = S.g_rdir.get(); ctx.nxt = S.g_nxt.get();
    if (cut_list) {
        ctx.mask = nullptr; ctx.rdir = nullptr; ctx.coff = S.g_coff.get(); ctx.brank = S.g_brank.get();
        S.tmask.release();
    }

    mark("phrase-end tables");
    // ---- the bins of the text suffixes (leading characters) ----
    S.g_nbins = 1u << (ctx.bits * prefix_chars);
    {
        DevBuf<uint64_t> d_bins;
        d_bins.ensure(std::max<uint32_t>(S.g_nbins, 4096u));
        MMT_HIP(hipMemsetAsync(d_bins.get(), 0, (size_t)std::max<uint32_t>(S.g_nbins, 4096u) * 8, st));
        gk::bin_hist(ctx, prefix_chars, d_bins.get(), st);
        d2h(S.g_bins, d_bins.get(), std::max<uint32_t>(S.g_nbins, 4096u), st);
    }
    S.g_bins_rep.clear(); S.g_repbits.release();
    if (S.expand) {
        // expansion: which occurrence stands for its distinct phrase, and how many suffixes of each bin start in one
        S.g_repbits.ensure(((size_t)m + 31) / 32 + 2);
        gk::rep_bits(S.pid.get(), S.rep.get(), m, S.g_repbits.get(), st);
        DevBuf<uint64_t> d_bins;
        d_bins.ensure(std::max<uint32_t>(S.g_nbins, 4096u));
        MMT_HIP(hipMemsetAsync(d_bins.get(), 0, (size_t)std::max<uint32_t>(S.g_nbins, 4096u) * 8, st));
        ctx.repbits = S.g_repbits.get();
        gk::bin_hist(ctx, prefix_chars, d_bins.get(), st);
        ctx.repbits = nullptr;
        d2h(S.g_bins_rep, d_bins.get(), std::max<uint32_t>(S.g_nbins, 4096u), st);
    }
    S.err.ensure(16);
    MMT_HIP(hipMemsetAsync(S.err.get(), 0, 64, st));

    mark("bin histograms");
    build_giant(hist);
    mark("giant dictionary");

    // ---- lexicographic ranks of the distinct phrases (one batch of the sort below), the parse ----
    auto t0 = now();
    {
        Batch X;
        const uint64_t fit = (uint64_t)(0.85 * (double)pool::available(device_) / (double)Batch::bytes_per_element());
        if (D > fit || D >= 0xfffffff0u)
            throw std::runtime_error("guided sort: " + std::to_string(D) + " distinct phrases are more than one batch can hold "
                                     "on this device (" + std::to_string(fit) + ")");
        X.reserve(std::max<uint32_t>(D, 1024));
        ctx.skip = 1; ctx.isa_p = nullptr; ctx.pos_bits = 40; ctx.rec_rank = 0;
        gk::phrase_items(ctx, S.pstart.get(), W, S.rep.get(), D, X.key_a.get(), X.pos_a.get(), st);
        RoundStats r1 = sort_batch(X, D, ctx, d_temp_, S.err.get(), st);
        guided_check_errors("phrases");
        S.prank.ensure(D); S.parse.ensure(m);
        gk::phrase_ranks(ctx, X.pos_b.get(), D, S.pid.get(), S.prank.get(), st);
        pk::parse_ranks(S.pid.get(), S.prank.get(), m, S.parse.get(), st);
        MMT_HIP(hipStreamSynchronize(st));
        if (stats) std::fprintf(stderr, "[guided] %u distinct phrases ranked in %.1f ms (%d rounds, %u tied after the first sort)\n", D,
                                ms_since(t0), r1.rounds, r1.first_active);
    }
    e3.stop(st);

    // ---- suffix array of the parse (parse.hpp:85), LCP of adjacent parse suffixes (pfp.hpp:210-244) ----
    e5.start(st);
    t0 = now();
    S.sa_p.ensure(m); S.isa_p.ensure(m);
    {
        const int pbits = std::max(1, bit_width_u64((uint64_t)D));
        const int pchars = std::max(1, 64 / pbits);
        sorter_.reserve(m);
        pk::pack_keys_u32(S.parse.get(), m, pbits, pchars, sorter_.keys_in(), sorter_.vals_in(), st);
        S.rounds_parse = sorter_.sort(m, pbits * pchars, (uint64_t)pchars, S.sa_p.get(), S.isa_p.get(), d_temp_, st);
        MMT_HIP(hipStreamSynchronize(st));
        sorter_.release();
    }
    S.plcp.build(text_ref(), n + 1 + w, S.sa_p.get(), S.pid.get(), S.pstart.get(), W, m, d_temp_, st);
    if (S.expand) {
        // the inverted lists of the distinct phrases (parse.hpp:106-134), as for the emitter of the parse proper
        // (pfp.cpp::pfp_prepare_emitter): occurrences ordered by (phrase, rank of the following parse suffix)
        const int shift = bit_width_u64((uint64_t)m + 1);
        const uint32_t pos_bits = W ? (uint32_t)bit_width_u64(n + w + 1) : 32u;
        const bool rec12 = (uint32_t)shift + pos_bits > 64 || std::getenv("MMT_OCC_REC12") != nullptr;
        S.occ_start.ensure((size_t)D + 2);
        S.occ_ids.ensure((size_t)m + 1); S.occ_ts.ensure((size_t)m + 1);
        {
            DevBuf<uint32_t> k_in, v_in;
            k_in.ensure((size_t)m + 1); v_in.ensure((size_t)m + 1);
            pk::occ_sequence(S.sa_p.get(), S.pid.get(), m, D, k_in.get(), v_in.get(), st);
            prims::sort_pairs_u32_u32(d_temp_, k_in.get(), S.occ_ids.get(), v_in.get(), S.occ_ts.get(), (size_t)m + 1, 0,
                                      std::max(1, bit_width_u64((uint64_t)D)), st);
            if (rec12) {
                S.occ.release(); S.occ_sl.release();
                S.occ12.ensure(3 * (size_t)m + 4);
                pk::occ_finish12(S.occ_ids.get(), S.occ_ts.get(), S.sa_p.get(), S.pstart.get(), W, m, S.occ_start.get(),
                                 S.occ12.get(), S.plcp.sl.get(), st);
            } else {
                S.occ12.release();
                S.occ.ensure(m); S.occ_sl.ensure(m);
                pk::occ_finish(S.occ_ids.get(), S.occ_ts.get(), S.sa_p.get(), S.pstart.get(), m, S.occ_start.get(), S.occ.get(),
                               pos_bits, S.plcp.sl.get(), S.occ_sl.get(), W, st);
            }
            MMT_HIP(hipStreamSynchronize(st));
        }
        S.occ_ids.release(); S.occ_ts.release();
        S.ptab.ensure((size_t)D * 16 + 16);
        pk::phrase_table(S.occ_start.get(), S.plen.get(), S.rep.get(), D, S.ptab.get(), st);
        MMT_HIP(hipStreamSynchronize(st));
        S.occ_start.release();
        S.emit_pos_bits = pos_bits; S.emit_w = w;
        pfp_emit_codes(shift);
    }
    // (the distinct-phrase ids stay: two suffixes of a group that start at the same offset of the same distinct phrase spell
    // the same alpha -- gk::med_before asks the ids before it compares characters)
    S.sa_p.release(); S.parse.release(); S.rep.release(); S.prank.release(); S.pstart.release();
    S.plen.release(); S.dlen.release(); S.dstart.release();
    ctx.pid = std::getenv("MMT_GUIDED_NO_PID") && !S.expand ? nullptr : S.pid.get();
    e5.stop(st);
    if (stats) std::fprintf(stderr, "[guided] parse of %u phrases sorted in %.1f ms (%d rounds)\n", m, ms_since(t0), S.rounds_parse);
    ctx.skip = 0; ctx.isa_p = S.isa_p.get();
    ctx.expand = S.expand ? 1u : 0u;
    if (S.expand) { MMT_HIP(hipStreamSynchronize(st)); S.isa_p.release(); ctx.isa_p = nullptr; }    // (representatives never ask for a parse rank)
    if (!S.expand) {
        // the parse rank rides in the record when it fits next to the position (MMT_GUIDED_NO_RANK: never -- tests)
        // (representatives never ask for a parse rank: their records carry the length of alpha)
        const uint32_t pb = (uint32_t)bit_width_u64(n + w + 1);
        if (pb + (uint32_t)bit_width_u64(m) <= 64 && !std::getenv("MMT_GUIDED_NO_RANK")) { ctx.pos_bits = pb; ctx.rec_rank = 1; }
    }
    S.ms[2] = 0; S.ms[3] = e3.ms(); S.ms[4] = 0; S.ms[5] = e5.ms();
    S.bwt_ready = true;
    S.n_groups = 0; S.dict_len = 0;
}

// Giant phrases -- longer than 24 first-key lengths (504 bases): a run of N, a microsatellite, any stretch without a
// trigger of the parse (newscan.hpp:265-325) -- would be refined 21 characters per round by every suffix that starts in
// them.  Their suffixes are sorted ONCE instead, as a dictionary of their own with the machinery of the parse proper (unique
// terminators, prefix doubling: a megabase of N is 20 rounds), with its LCP array (irreducible entries + PLCP chain) and a
// range-minimum structure on it; a comparison that is undecided 504 characters into alpha continues on those ranks
// (gk::cmp_rest, k_round_keys), and the LCP of two such suffixes is a range minimum.
void Engine::build_giant(const std::vector<uint64_t>& hist) {
    PfpState& S = *pfp_;
    hipStream_t st = stream_;
    gk::Ctx& ctx = S.gctx;
    const bool W = wide_;
    const uint32_t m = S.n_phrases, D = S.n_distinct;
    // (the threshold follows the modulus of the parse: beyond 48 G characters the modulus grows with the text -- p = 157 at
    // 250 G characters -- and an ordinary phrase of a few hundred characters must not count as giant: twelve mean phrase
    // lengths leave e^-12 of the phrases, 24 key lengths at the moduli of the smaller texts as before)
    ctx.g_n = 0;
    ctx.g_depth = std::max<uint32_t>(24u, (12u * S.p + (uint32_t)ctx.chars - 1) / (uint32_t)ctx.chars) * (uint32_t)ctx.chars;
    S.gi_occ = S.gi_distinct = S.gi_chars = 0;
    if (std::getenv("MMT_GUIDED_NO_GIANT")) return;
    if (const char* c = std::getenv("MMT_GIANT_DEPTH")) ctx.g_depth = (uint32_t)std::max(1, std::atoi(c)) * (uint32_t)ctx.chars;
    DevBuf<uint32_t> flags, gids, count;
    count.ensure(4);
    // distinct phrases longer than g_depth (dlen counts the terminator) ...
    flags.ensure(std::max(D, m)); gids.ensure(D);
    gk::flag_greater(S.dlen.get(), D, ctx.g_depth + 1, flags.get(), st);
    // ... and the phrases NEXT to their occurrences in the parse (four steps either way).  A group of suffixes takes its order
    // from the giant dictionary at once when ALL its members lie in giant phrases (sort_batch, k_giant_probe); the suffixes inside
    // a microsatellite -- 1.3 G representatives of a rank's share of 13 realistic whole genomes: a phrase of 2.5 kb is private
    // to its haplotype -- share hundreds of characters with thousands of others, and one member in an ORDINARY phrase (two
    // mutations that made triggers a few hundred bases apart: 3 % of the positions of such an array) keeps its whole group walking
    // 21 characters a round to the giant depth.  Which phrases the giant dictionary holds is free as long as it holds every
    // phrase longer than g_depth: the short phrases between and beside giant ones are the ones that sit in those groups.
    // (MMT_GIANT_PROMOTE=<steps>, 0: none)
    DevBuf<uint32_t> occ_flag;                  // per phrase of the parse: lies in a phrase of the giant dictionary
    const int promote = std::getenv("MMT_GIANT_PROMOTE") ? std::max(0, std::atoi(std::getenv("MMT_GIANT_PROMOTE"))) : 4;
    bool promoted = false;
    uint32_t nG = 0;
    uint64_t nd64 = 0;
    DevBuf<uint32_t> which, glen, gstart, dmap;
    // (a dictionary that the neighbours push beyond 32 bits is built with fewer of them, then with none: the phrases longer than
    // g_depth are the ones it must hold)
    std::vector<int> attempts{promote};
    if (promote > 1) attempts.push_back(1);
    if (promote > 0) attempts.push_back(0);
    for (size_t a = 0; a < attempts.size(); a++) {
        const int steps = attempts[a];
        gk::flag_greater(S.dlen.get(), D, ctx.g_depth + 1, flags.get(), st);
        promoted = false;
        if (steps > 0 && S.pid.get()) {
            DevBuf<uint32_t> spread;
            occ_flag.ensure(m); spread.ensure(m);
            gk::flag_greater(S.plen.get(), m, ctx.g_depth, occ_flag.get(), st);
            MMT_HIP(hipMemsetAsync(count.get(), 0, 16, st));
            pk::sum_u32(occ_flag.get(), m, reinterpret_cast<uint64_t*>(count.get()), st);
            uint64_t any = 0;
            MMT_HIP(hipMemcpyAsync(&any, count.get(), 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
            if (any) {
                for (int h = 0; h < steps; h++) { gk::flag_spread(occ_flag.get(), m, spread.get(), st); occ_flag.swap(spread); }
                gk::flag_to_distinct(occ_flag.get(), S.pid.get(), m, flags.get(), st);
                promoted = true;
            }
        }
        prims::select_indices_u32flags(d_temp_, flags.get(), gids.get(), count.get(), D, st);
        nG = read_u32(count.get(), st);
        if (!nG) return;
        which.ensure(nG); glen.ensure(nG); gstart.ensure(nG);
        gk::giant_distinct(gids.get(), nG, S.rep.get(), S.dlen.get(), which.get(), glen.get(), st);
        DevBuf<uint64_t> total;
        total.ensure(1);
        pk::sum_u32(glen.get(), nG, total.get(), st);
        MMT_HIP(hipMemcpyAsync(&nd64, total.get(), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        nd64 += 1;
        if (std::getenv("MMT_GUIDED_STATS"))
            std::fprintf(stderr, "[guided] giant dictionary with the phrases within %d steps of a giant occurrence: %u phrases, %llu characters\n",
                         steps, nG, (unsigned long long)nd64);
        if (nd64 < 0xffffff00ull || a + 1 == attempts.size()) break;
    }
    dmap.ensure(D);
    if (nd64 >= 0xffffff00ull) throw std::runtime_error("giant phrases of " + std::to_string(nd64) + " characters in all exceed a 32-bit dictionary");
    const uint32_t nd = (uint32_t)nd64;
    prims::exclusive_sum_u32(d_temp_, glen.get(), gstart.get(), nG, st);
    MMT_HIP(hipMemsetAsync(dmap.get(), 0xff, (size_t)D * 4, st));
    gk::giant_map(gids.get(), gstart.get(), nG, dmap.get(), st);
    // the giant dictionary, its suffix array (unique terminators: prefix doubling as for the parse proper's dictionary)
    DevBuf<uint8_t> dict, code_d, ebw;
    DevBuf<uint64_t> dinfo;
    dict.ensure((size_t)nd + 64); dinfo.ensure(nd);
    MMT_HIP(hipMemsetAsync(dict.get() + nd, 0, 64, st));
    pk::copy_dict(text_ref(), S.pstart.get(), S.plen.get(), which.get(), gstart.get(), nG, dict.get(), dinfo.get(), nd, false, W, st);
    uint8_t code[256];
    int sigma = 0;
    // (0x00 -- once, the last byte of the dictionary -- and the phrase terminator 0x01 share code 0 with the padding behind
    // the end: nothing is compared behind a terminator, terminators are ordered by position.  Dollar, the document
    // separator, A C G T and N then fit three bits: 21 characters per 64-bit key instead of 15)
    for (int c = 0; c < 256; c++) code[c] = (c > 1 && (hist[c] || c == 2)) ? (uint8_t)(++sigma) : 0;
    const int bits = std::max(1, bit_width_u64((uint64_t)sigma)), chars = std::min(63 / bits, 63);
    code_d.ensure(256);
    MMT_HIP(hipMemcpyAsync(code_d.get(), code, 256, hipMemcpyHostToDevice, st));
    DevBuf<uint32_t> sa_g, esuf, ephr, plcp, cnt2, huge;
    sa_g.ensure(nd); S.gi_isa.ensure(nd);
    sorter_.reserve(nd);
    k::pack_keys(dict.get(), nd, code_d.get(), bits, chars, (uint32_t)code[1], sorter_.keys_in(), sorter_.vals_in(), st);
    const int rounds = sorter_.sort(nd, bits * chars + 1, (uint64_t)chars, sa_g.get(), S.gi_isa.get(), d_temp_, st, true);
    MMT_HIP(hipStreamSynchronize(st));
    sorter_.release();
    esuf.ensure(nd); ephr.ensure(nd); ebw.ensure(nd);
    pk::entry_info(sa_g.get(), dinfo.get(), dict.get(), nd, false, esuf.get(), ephr.get(), ebw.get(), st);
    // its LCP array (irreducible entries compared directly, the rest by the PLCP chain), groups of equal strings
    plcp.ensure((size_t)nd + 16); cnt2.ensure(4); S.gi_lcp.ensure(((size_t)nd + 3) / 4 * 4 + 16);
    {
        DevBuf<uint8_t> longs;
        uint32_t cap = std::max<uint32_t>(nd / 64 + 4096, 1u << 16), found = 0;
        for (int attempt = 0;; attempt++) {
            longs.ensure((size_t)cap * sizeof(k::LongLcpLim));
            pk::dict_irreducible(dict.get(), nd, sa_g.get(), esuf.get(), ebw.get(), plcp.get(), longs.get(), cnt2.get(), cap, st);
            found = read_u32(cnt2.get(), st);
            if (found <= cap) break;
            if (attempt) throw std::runtime_error("long-match list overflow in the giant dictionary's LCP construction");
            cap = found + 1024;
        }
        if (found) {
            huge.ensure((size_t)found + 1);
            k::long_lcp_lim(dict.get(), nd, longs.get(), found, plcp.get(), huge.get(), cnt2.get() + 1, st);
        }
        d_temp_.ensure(k::plcp_running_max_scratch(nd));
        k::plcp_running_max(plcp.get(), nd, d_temp_.get(), st);
        SaCol sd; sd.lo = sa_g.get(); sd.hi = nullptr;
        k::lcp_gather(plcp.get(), sd, 0, nd, S.gi_lcp.get(), st);
        pk::dict_lcp_clamp(S.gi_lcp.get(), esuf.get(), nd, st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    S.gi_grp.ensure(nd);
    gk::giant_group_flags(esuf.get(), S.gi_lcp.get(), nd, S.gi_grp.get(), st);
    prims::inclusive_sum_u32(d_temp_, S.gi_grp.get(), S.gi_grp.get(), nd, st);
    build_rmq(S.gi_lcp.get(), nd, S.gi_bmin, S.gi_nb, S.gi_levels, st);
    // the occurrences of giant phrases in the parse: phrase index (ascending), first V index, place in the dictionary
    if (promoted) {
        // (the occurrences of the dictionary's phrases: every occurrence of a promoted phrase, wherever it lies)
        DevBuf<uint32_t> dflag;
        dflag.ensure(D);
        MMT_HIP(hipMemsetAsync(dflag.get(), 0, (size_t)D * 4, st));
        gk::flag_scatter_ones(gids.get(), nG, dflag.get(), st);
        gk::flag_from_distinct(dflag.get(), S.pid.get(), m, flags.get(), st);
        MMT_HIP(hipStreamSynchronize(st));
    } else
    gk::flag_greater(S.plen.get(), m, ctx.g_depth, flags.get(), st);
    occ_flag.release();
    S.gi_bits.ensure(((size_t)m + 31) / 32 + 1);
    gk::giant_bits(flags.get(), m, S.gi_bits.get(), st);
    {
        // rank directory over those bits: giant occurrences before phrase 32 j (gk::giant_entry)
        const uint32_t nw = (uint32_t)(((size_t)m + 31) / 32);
        DevBuf<uint32_t> pc;
        pc.ensure(nw + 1); S.gi_rank.ensure(nw + 1);
        gk::popcount_words(S.gi_bits.get(), nw, pc.get(), st);
        prims::exclusive_sum_u32(d_temp_, pc.get(), S.gi_rank.get(), nw, st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    DevBuf<uint32_t> occ_idx;
    occ_idx.ensure(m);
    prims::select_indices_u32flags(d_temp_, flags.get(), occ_idx.get(), count.get(), m, st);
    const uint32_t nO = read_u32(count.get(), st);
    S.gi_k.ensure(std::max<uint32_t>(nO, 1)); S.gi_ps.ensure(std::max<uint32_t>(nO, 1)); S.gi_base.ensure(std::max<uint32_t>(nO, 1));
    if (nO) {
        MMT_HIP(hipMemcpyAsync(S.gi_k.get(), occ_idx.get(), (size_t)nO * 4, hipMemcpyDeviceToDevice, st));
        gk::giant_occurrences(S.gi_k.get(), nO, S.pid.get(), S.pstart.get(), W, dmap.get(), S.gi_ps.get(), S.gi_base.get(), st);
    }
    MMT_HIP(hipStreamSynchronize(st));
    S.gi_occ = nO; S.gi_distinct = nG; S.gi_chars = nd;
    ctx.g_k = S.gi_k.get(); ctx.g_ps = S.gi_ps.get(); ctx.g_base = S.gi_base.get(); ctx.g_n = nO;
    ctx.g_isa = S.gi_isa.get(); ctx.g_grp = S.gi_grp.get(); ctx.g_bits = S.gi_bits.get(); ctx.g_rank = S.gi_rank.get();
    ctx.g_rmq.sl = S.gi_lcp.get(); ctx.g_rmq.bmin = S.gi_bmin.get(); ctx.g_rmq.m = nd; ctx.g_rmq.nb = S.gi_nb;
    if (std::getenv("MMT_GUIDED_STATS"))
        std::fprintf(stderr, "[guided] %u giant distinct phrases (%u characters, sorted in %d rounds), %u occurrences in the parse\n",
                     nG, nd, rounds, nO);
}

// does the share of the stream the last run produced hold the suffixes that begin with this k-mer?  (tests: bigchecks.py)
int Engine::kmer_in_share(const uint8_t* kmer, size_t k) const {
    const PfpState& S = *pfp_;
    if (!S.guided || !S.g_share_valid || (size_t)S.g_prefix > k) return -1;
    uint32_t bin = 0;
    for (int c = 0; c < S.g_prefix; c++) bin = (bin << S.g_bits) | S.g_code[kmer[c]];
    return bin >= S.g_share_lo && bin < S.g_share_hi ? 1 : 0;
}

void Engine::guided_check_errors(const char* what) {
    PfpState& S = *pfp_;
    uint32_t e2[3];
    MMT_HIP(hipMemcpyAsync(e2, S.err.get(), 12, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    if (e2[0] || e2[1] || e2[2])
        throw std::runtime_error(std::string("guided sort (") + what + "): phrase suffixes are not prefix-free (" +
                                 std::to_string(e2[0]) + " equal distinct phrases, " + std::to_string(e2[1]) +
                                 " groups with spent and unspent members, " + std::to_string(e2[2]) +
                                 " neighbours equal up to the end of alpha without ascending parse ranks)");
}

// The stream, batch by batch.  A batch = whole bins of leading characters = one contiguous piece of the suffix array: its
// suffixes are collected in text order, sorted, written as a window of the columns (suffix array, BWT, LCP from the
// parse), scanned, and dropped; the accepted rows take their suffix-array entries along.  A rank of a sharded run takes
// the bins from the first one whose cumulative count reaches k n / count (every rank derives the same shares from the
// same histogram); nothing is exchanged but the rows.
void Engine::guided_stream(ScanState& SS, const mmt_params& p) {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    hipStream_t st = stream_;
    DevBuf<unsigned long long> prof;
    if (std::getenv("MMT_GUIDED_PROF")) {
        prof.ensure(16);
        MMT_HIP(hipMemsetAsync(prof.get(), 0, 16 * 8, st));
        S.gctx.prof = prof.get();
    }
    struct ProfOff { gk::Ctx& c; ~ProfOff() { c.prof = nullptr; } } prof_off{S.gctx};
    auto print_prof = [&]() {
        if (!prof.get()) return;
        std::vector<unsigned long long> h(16);
        MMT_HIP(hipMemcpyAsync(h.data(), prof.get(), 16 * 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        const double w = (double)std::max<unsigned long long>(1, h[7]);
        std::fprintf(stderr, "[guided] k_resolve_medium: %llu waves; ticks of the 100 MHz clock per wave: staging %.0f, reference %.0f, "
                     "first comparison %.0f, network %.0f, output %.0f; members with a difference %llu, undecided %llu, text "
                     "comparisons inside the network %llu; members %llu, of them in the reference's class %llu, groups whose best class has one member %llu, "
                     "groups where half the members differ from the reference %llu, members that repeat another's (place, character) %llu\n",
                     h[7], h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[8], h[9], h[10], h[12], h[11], h[13], h[14], h[15]);
    };
    if (pfp_->expand) { guided_stream_expand(SS, p); print_prof(); return; }
    const gk::Ctx& ctx = S.gctx;
    const int prefix_chars = S.g_prefix;
    const uint32_t n_bins = S.g_nbins;
    const std::vector<uint64_t>& bins = S.g_bins;
    const bool stats = std::getenv("MMT_GUIDED_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    if (p.min_match_len < (uint32_t)prefix_chars)
        throw std::runtime_error("guided producer: the bins were formed for another minimum match length");

    // ---- shares of the ranks: whole bins ----
    std::vector<uint64_t> pre(n_bins + 1, 0);                  // suffixes in the bins before bin b
    for (uint32_t bq = 0; bq < n_bins; bq++) pre[bq + 1] = pre[bq] + bins[bq];
    if (pre[n_bins] != n) throw std::runtime_error("guided sort: the histogram of leading characters does not cover the text");
    std::vector<uint32_t> cut(shard_count_ + 1, 0);
    cut[shard_count_] = n_bins;
    for (uint32_t k = 1; k < shard_count_; k++) {
        const uint64_t target = (uint64_t)((unsigned __int128)n * k / shard_count_);
        cut[k] = std::max<uint32_t>(cut[k - 1], (uint32_t)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin()));
        if (cut[k] > n_bins) cut[k] = n_bins;
    }
    sort_pieces_.clear();
    for (uint32_t q = 0; q < shard_count_; q++) sort_pieces_.emplace_back(pre[cut[q]], pre[cut[q + 1]] - pre[cut[q]]);
    const uint32_t bin_lo = cut[shard_index_], bin_hi = cut[shard_index_ + 1];
    S.g_share_lo = bin_lo; S.g_share_hi = bin_hi; S.g_share_valid = true;
    if (shard_count_ > 1 && p.merge_metadata)
        throw std::runtime_error("merge metadata needs the whole stream on one rank (partition the documents instead)");

    // ---- batch capacity: the batch scratch + two window sets (a batch with the tail of the one before) ----
    // what a batch's first closing position may need of the batch before: the last bin of that batch, as far as an
    // interval can reach
    const bool capped = SS.cap != 0;
    uint64_t largest = 0;
    for (uint32_t b = bin_lo; b < bin_hi; b++) largest = std::max(largest, bins[b]);
    const uint64_t head_room = capped ? std::min<uint64_t>(SS.ext0, largest) : largest;
    const double per_element = (double)Batch::bytes_per_element() + 2.0 * (wide_ ? 10.0 : 9.0);
    uint64_t cap64 = std::min<uint64_t>(std::max<uint64_t>(n, 1024), 1ull << 30);
    // (what else the batches allocate: the tile tables of the two text-order kernels, 8 bytes per 4096 text positions; the
    // candidates and rows of the scan; and the heap wants its blocks contiguous -- at 573 G characters 85 % of what was free
    // left the last allocation 9 GB short)
    const double avail = 0.80 * (double)pool::available(device_) - 2.0 * 10.0 * (double)head_room -
                         8.0 * (double)((n + gk::TILE - 1) / gk::TILE) - 6.0 * 1073741824.0;
    const uint64_t fit = avail > 0 ? (uint64_t)(avail / per_element) : 0;
    cap64 = std::min(cap64, std::max<uint64_t>(fit, 1u << 20));
    if (const char* c = std::getenv("MMT_GUIDED_BATCH")) cap64 = std::max<uint64_t>(1024, std::strtoull(c, nullptr, 10));
    if (largest > cap64) {
        if (largest > fit || largest >= 0xfffffff0ull)
            throw std::runtime_error("guided sort: " + std::to_string(largest) + " suffixes share their first " +
                                     std::to_string(prefix_chars) + " characters: more than one batch can hold on this "
                                     "device (" + std::to_string(fit) + ")");
        cap64 = largest;
    }
    // ---- several batches per pass over the text (gk::stage_fill): when the share takes more than two batches, part of the
    // memory becomes a list of the suffixes of the next batches (8 bytes each), filled by ONE pass over the text -- a pass per
    // batch was 126 of the 186 s the device was busy on a rank's share of 573 G characters.  MMT_GUIDED_STAGE=0 | 1 forces.
    // It pays when the passes it saves are long and many: text characters x batches from 2 x 10^13 on (configs[4]: 573 G x 127;
    // a share of configs[3], 79 G x 86, ran 38 - 45 s without the list and 42 - 48 s with it: smaller batches, keys looked up at
    // random instead of rolled along the tile).
    const uint64_t share = pre[bin_hi] - pre[bin_lo];
    bool staged = share > 2 * cap64 && (double)n * ((double)share / (double)cap64) >= 2e13;
    if (const char* c = std::getenv("MMT_GUIDED_STAGE")) staged = std::atoi(c) != 0;
    else if (std::getenv("MMT_GUIDED_BATCH")) staged = share > 2 * cap64;            // (tests: every run of several batches)
    uint64_t stage_cap = 0;
    if (staged) {
        if (std::getenv("MMT_GUIDED_BATCH")) stage_cap = std::min<uint64_t>(share, std::max<uint64_t>(4 * cap64, largest));   // (tests: a few batches per pass)
        else {
            // (what a full batch leaves, if that holds two batches' worth; else 40 % of what the batches may use)
            const double spare = avail - (double)cap64 * per_element;
            double stage_bytes = spare / 8.0 >= 2.0 * (double)cap64 ? spare : 0.4 * avail;
            uint64_t fit2 = (uint64_t)(std::max(avail - stage_bytes, 0.0) / per_element);
            if (fit2 < largest) { fit2 = largest; stage_bytes = avail - (double)largest * per_element; }
            stage_cap = stage_bytes > 0 ? (uint64_t)(stage_bytes / 4.0) : 0;        // (four bytes an entry since round 6)
            stage_cap = std::min<uint64_t>(std::min<uint64_t>(stage_cap, share), 0xfff00000ull);      // (32-bit offsets per tile)
            if (stage_cap < 2 * std::max<uint64_t>(largest, 1) || fit2 < (1u << 20)) staged = false;
            else cap64 = std::min<uint64_t>(cap64, std::max<uint64_t>(fit2, largest));
        }
        if (stage_cap < largest) staged = false;
    }
    Batch X;
    X.reserve((uint32_t)cap64);
    DevBuf<uint32_t> stage, blk_tile;
    DevBuf<uint32_t> blk_cnt, blk_off;
    if (staged) { stage.ensure(stage_cap + 16); blk_cnt.ensure(stage_cap / 4096 + 2); blk_off.ensure(stage_cap / 4096 + 2); blk_tile.ensure(stage_cap / 4096 + 3); }
    window_reserve(0, head_room + cap64 + 16);
    window_reserve(1, head_room + cap64 + 16);
    DevBuf<uint64_t> carry;
    carry.ensure(2);
    const uint32_t n_tiles = (uint32_t)((n + gk::TILE - 1) / gk::TILE);
    DevBuf<uint32_t> tile_cnt, tile_off;
    tile_cnt.ensure((size_t)n_tiles + 1); tile_off.ensure((size_t)n_tiles + 1);
    const uint64_t anchor = std::min<uint64_t>(doc_len_[0], n);

    uint64_t base = pre[bin_lo], active_sum = 0, small_sum = 0;
    const uint64_t piece_end = pre[bin_hi];
    int batches = 0, rounds_max = 0;
    uint64_t prev_len = 0;                 // entries of the window before (without a virtual closing entry)
    uint32_t prev_last_bin = 0;            // its last non-empty bin
    bool have_prev = false;
    uint32_t pass_end = bin_lo;            // staged: the bins [.., pass_end) are in the list
    uint32_t counted_lo = 0, counted_hi = 0;   // ... and tile_cnt holds the per-tile counts of the bins [counted_lo, counted_hi)
    uint32_t taken_lo = 0, taken_hi = 0;       // ... and blk_cnt the per-block counts of the list's entries of the bins [taken_lo, taken_hi)
    uint64_t n_staged = 0;
    int passes = 0;
    for (uint32_t b0 = bin_lo; b0 < bin_hi;) {
        if (staged && b0 == pass_end) {    // the next pass over the text: as many bins as the list holds
            n_staged = 0;
            while (pass_end < bin_hi && n_staged + bins[pass_end] <= stage_cap) n_staged += bins[pass_end++];
            if (pass_end == b0) throw std::runtime_error("guided sort: a bin exceeds the staging list");
            taken_lo = taken_hi = 0;               // (a new list: nothing of it is counted yet)
            if (n_staged) {
                // (the pass before counted this pass's suffixes per tile while it filled its own list)
                if (!(counted_lo == b0 && counted_hi == pass_end)) gk::batch_count(ctx, prefix_chars, b0, pass_end, tile_cnt.get(), st);
                prims::exclusive_sum_u32(d_temp_, tile_cnt.get(), tile_off.get(), n_tiles, st);
                uint32_t next_end = pass_end;
                uint64_t next_total = 0;
                while (next_end < bin_hi && next_total + bins[next_end] <= stage_cap) next_total += bins[next_end++];
                const bool more = next_total > 0;
                gk::stage_fill(ctx, prefix_chars, b0, pass_end, tile_off.get(), stage.get(), pass_end, next_end,
                               more ? tile_cnt.get() : nullptr, st);
                gk::stage_block_tiles(tile_off.get(), n_tiles, n_staged, blk_tile.get(), st);
                counted_lo = more ? pass_end : 0; counted_hi = more ? next_end : 0;
            }
            passes++;
        }
        uint64_t total = 0;
        uint32_t b1 = b0;
        const uint32_t b_stop = staged ? pass_end : bin_hi;
        while (b1 < b_stop && total + bins[b1] <= X.cap) total += bins[b1++];
        if (b1 == b0) throw std::runtime_error("guided sort: a bin exceeds the batch");
        if (total) {
            const uint32_t B = (uint32_t)total;
            const int set = batches & 1;
            EventPair& ee = next_range_event(SS, 3);
            ee.start(st);
            if (staged) {
                const uint32_t nb = (uint32_t)((n_staged + 4095) / 4096);
                // (the batch before counted this batch's entries per block while it took its own)
                if (!(taken_lo == b0 && taken_hi == b1)) gk::stage_count(stage.get(), n_staged, b0, b1, blk_cnt.get(), st);
                prims::exclusive_sum_u32(d_temp_, blk_cnt.get(), blk_off.get(), nb, st);
                uint32_t nb1 = b1;
                uint64_t next_total = 0;
                while (nb1 < pass_end && next_total + bins[nb1] <= X.cap) next_total += bins[nb1++];
                const bool more = next_total > 0;
                gk::stage_take(ctx, stage.get(), n_staged, b0, b1, blk_off.get(), X.key_a.get(), X.pos_a.get(), b1, nb1,
                               more ? blk_cnt.get() : nullptr, tile_off.get(), blk_tile.get(), st);
                taken_lo = more ? b1 : 0; taken_hi = more ? nb1 : 0;
            } else {
                // (the batch before counted this batch's suffixes per tile while it filled its own)
                if (!(counted_lo == b0 && counted_hi == b1)) gk::batch_count(ctx, prefix_chars, b0, b1, tile_cnt.get(), st);
                prims::exclusive_sum_u32(d_temp_, tile_cnt.get(), tile_off.get(), n_tiles, st);
                uint32_t nb1 = b1;
                uint64_t next_total = 0;
                while (nb1 < bin_hi && next_total + bins[nb1] <= X.cap) next_total += bins[nb1++];
                const bool more = next_total > 0;
                gk::batch_fill(ctx, prefix_chars, b0, b1, tile_off.get(), X.key_a.get(), X.pos_a.get(), b1, nb1,
                               more ? tile_cnt.get() : nullptr, st);
                counted_lo = more ? b1 : 0; counted_hi = more ? nb1 : 0;
            }
            // (the LCP values the sort finds on its way go straight into the window, behind the tail of the batch before)
            uint64_t ext = 0;
            if (have_prev) ext = std::min<uint64_t>(std::min<uint64_t>(bins[prev_last_bin], prev_len), capped ? SS.ext0 : ~0ull);
            if (ext > head_room) throw std::runtime_error("guided sort: window head room too small");
            const RmqView rmq = S.plcp.view();
            RoundStats rs = sort_batch(X, B, ctx, d_temp_, S.err.get(), st, w_lcp_[set].get() + ext, &rmq);
            // the window: [tail of the batch before | this batch | one virtual closing entry at the end of a rank's share]
            if (ext) {
                const int o = set ^ 1;
                const uint64_t from = prev_len - ext;
                MMT_HIP(hipMemcpyAsync(w_sa_[set].get(), w_sa_[o].get() + from, ext * 4, hipMemcpyDeviceToDevice, st));
                if (wide_) MMT_HIP(hipMemcpyAsync(w_hi_[set].get(), w_hi_[o].get() + from, ext, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(w_bwt_[set].get(), w_bwt_[o].get() + from, ext, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(w_lcp_[set].get(), w_lcp_[o].get() + from, ext * 4, hipMemcpyDeviceToDevice, st));
            }
            SaCol wsa; wsa.lo = w_sa_[set].get(); wsa.hi = wide_ ? w_hi_[set].get() : nullptr;
            gk::write_columns(ctx, X.pos_b.get(), B, ext, wsa, w_bwt_[set].get(), st);
            gk::batch_lcp(ctx, S.plcp.view(), X.pos_b.get(), B, carry.get(), have_prev, w_lcp_[set].get() + ext, S.err.get(), st);
            MMT_HIP(hipMemcpyAsync(carry.get(), X.pos_b.get() + (B - 1), 8, hipMemcpyDeviceToDevice, st));
            ee.stop(st);
            stream_entries_ += B;
            uint64_t len = ext + B;
            ColWindow w = window_view(set, base - ext, (uint32_t)len, (uint32_t)ext);
            w.more_left = false;          // nothing an interval of this window could reach lies further left (bins)
            keep_window(w);
            if (want_anchor_ranks_) {
                SaCol piece = w.sa; piece.lo += ext; if (piece.hi) piece.hi += ext;
                k::anchor_ranks(piece, base, B, anchor, wide_ ? (void*)d_rank64_.get() : (void*)d_rank_.get(), st);
            }
            const bool last_of_share = b1 == bin_hi || base + B == piece_end;
            if (last_of_share && base + B < n) {
                // the first entry of the next rank's share closes what is still open here: its LCP is below the bins'
                // prefix length, below every reportable value -- stand-in entry with LCP 0
                MMT_HIP(hipMemsetAsync(w_lcp_[set].get() + len, 0, 4, st));
                MMT_HIP(hipMemsetAsync(w_bwt_[set].get() + len, 0, 1, st));
                MMT_HIP(hipMemsetAsync(w_sa_[set].get() + len, 0, 4, st));
                if (wide_) MMT_HIP(hipMemsetAsync(w_hi_[set].get() + len, 0, 1, st));
                w.len = (uint32_t)(len + 1);
            }
            if (!scan_window(SS, w, p)) throw std::runtime_error("guided sort: a walk left its bin");
            sink_flush(SS);
            prev_len = len; have_prev = true;
            for (uint32_t b = b1; b-- > b0;) if (bins[b]) { prev_last_bin = b; break; }
            base += B; batches++; rounds_max = std::max(rounds_max, rs.rounds); active_sum += rs.active_sum; small_sum += rs.small;
        }
        b0 = b1;
    }
    guided_check_errors("text suffixes");
    print_prof();
    if (base != piece_end) throw std::runtime_error("guided sort: the batches do not cover the text exactly once");
    S.rounds_dict = rounds_max; S.emit_launches = (uint32_t)batches;
    run_slices_ = 0; text_passes_ = (uint32_t)(staged ? passes : batches); batches_ = (uint32_t)batches; staged_ = staged;
    MMT_HIP(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (stats && staged) std::fprintf(stderr, "[guided] %d passes over the text for those batches (a list of %llu suffixes)\n", passes,
                                      (unsigned long long)stage_cap);
    if (stats) std::fprintf(stderr, "[guided] %llu suffixes in %d batches of at most %u: %.1f ms with their scans; %.3f of them settled in "
                            "small groups by comparison, %.3f element-rounds per suffix in %d rounds at most\n",
                            (unsigned long long)(piece_end - pre[bin_lo]), batches, X.cap, ms,
                            (double)small_sum / (double)std::max<uint64_t>(1, piece_end - pre[bin_lo]),
                            (double)active_sum / (double)std::max<uint64_t>(1, piece_end - pre[bin_lo]), rounds_max);
    S.ms[6] = (float)ms;
    sort_rounds_ = rounds_max;
}

// The same stream when the collection is redundant (PfpState::expand): a batch collects, of its bins, only the suffixes that
// start in the REPRESENTATIVE occurrence of their distinct phrase -- one per valid suffix of the dictionary of the parse
// (include/dictionary.hpp:103-157), which is never built --, sorts them with the machinery above, and hands the sorted
// batch to the emitter of the parse proper (pfp_kernels.hip k_emit, pfp_lcp_mum.hpp:151-212) as its entry tables: every
// representative is expanded by the inverted list of its phrase, groups of equal phrase suffixes are merged by the rank of
// the following parse suffix.  {anchor + 12} whole-genome haplotypes: 79 G text suffixes, 14 G representatives.
void Engine::guided_stream_expand(ScanState& SS, const mmt_params& p) {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    const bool W = wide_;
    hipStream_t st = stream_;
    gk::Ctx& ctx = S.gctx;
    ctx.repbits = S.g_repbits.get();
    ctx.pid = S.pid.get();
    const int prefix_chars = S.g_prefix;
    const uint32_t n_bins = S.g_nbins;
    const std::vector<uint64_t>& bins = S.g_bins;
    const std::vector<uint64_t>& rbins = S.g_bins_rep;
    const bool stats = std::getenv("MMT_GUIDED_STATS") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    if (p.min_match_len < (uint32_t)prefix_chars)
        throw std::runtime_error("guided producer: the bins were formed for another minimum match length");
    if (!S.ptab.get() || (!S.occ.get() && !S.occ12.get())) throw std::runtime_error("expansion: the inverted lists are gone");

    // ---- shares of the ranks: whole bins, cut by the TEXT suffixes (the same shares as the plain producer's) ----
    std::vector<uint64_t> pre(n_bins + 1, 0);
    uint64_t reps_total = 0;
    for (uint32_t bq = 0; bq < n_bins; bq++) { pre[bq + 1] = pre[bq] + bins[bq]; reps_total += rbins[bq]; }
    if (pre[n_bins] != n) throw std::runtime_error("guided sort: the histogram of leading characters does not cover the text");
    std::vector<uint32_t> cut(shard_count_ + 1, 0);
    cut[shard_count_] = n_bins;
    for (uint32_t k = 1; k < shard_count_; k++) {
        const uint64_t target = (uint64_t)((unsigned __int128)n * k / shard_count_);
        cut[k] = std::max<uint32_t>(cut[k - 1], (uint32_t)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin()));
        if (cut[k] > n_bins) cut[k] = n_bins;
    }
    sort_pieces_.clear();
    for (uint32_t q = 0; q < shard_count_; q++) sort_pieces_.emplace_back(pre[cut[q]], pre[cut[q + 1]] - pre[cut[q]]);
    const uint32_t bin_lo = cut[shard_index_], bin_hi = cut[shard_index_ + 1];
    S.g_share_lo = bin_lo; S.g_share_hi = bin_hi; S.g_share_valid = true;
    if (shard_count_ > 1 && p.merge_metadata)
        throw std::runtime_error("merge metadata needs the whole stream on one rank (partition the documents instead)");

    // ---- capacities.  A BATCH is what is collected by one pass over the text and sorted at once: representatives of whole
    // bins, at most 2^30; the emitter's entry and group tables of the batch live in the sort's scratch, which is dead by then.
    // A WINDOW is what is emitted, scanned and dropped at once: whole bins of one batch, below 2^32 entries with the tail of
    // the window before (ONE window set: the tail -- as far as an interval can reach -- waits in a buffer of its own). ----
    const bool capped = SS.cap != 0;
    // A bin of ONE repeated symbol -- N^pc: the assembly gaps of every haplotype and strand, 1.5 G suffixes in a rank's share of
    // whole genomes with 60 Mbp of gaps each -- is not cut by more leading characters, but its order is known in closed form
    // (guided_kernels.hip RunSlice): when it exceeds a batch or a window it is produced in slices.  Only in the capped modes: an
    // interval of an uncapped mode may be as long as its bin, and a window must hold it.  (MMT_GUIDED_SLICE=<suffixes>: tests)
    const uint32_t smask = (1u << ctx.bits) - 1u;
    auto run_symbol = [&](uint32_t b) -> uint32_t {
        const uint32_t sym = b & smask;
        if (!sym) return 0u;
        for (int ch = 1; ch < prefix_chars; ch++) if (((b >> (ctx.bits * ch)) & smask) != sym) return 0u;
        return sym;
    };
    const uint64_t slice_env = std::getenv("MMT_GUIDED_SLICE") ? std::strtoull(std::getenv("MMT_GUIDED_SLICE"), nullptr, 10) : 0;
    // (the closed form needs every phrase suffix that begins inside a run to reach beyond the run's end -- all occurrences of a
    // representative then share (r, X0).  That holds unless the window of w equal symbols is itself a trigger of the parse
    // (newscan.hpp:106-114,321: hash of c^w divisible by p -- a phrase would end at EVERY position of the run); the automatic
    // parameters avoid such moduli (engine.cpp), a parse that was asked for with one keeps its bins whole)
    auto run_triggers = [&](uint32_t sym) {
        for (int c = 0; c < 256; c++)
            if (S.g_code[c] == sym) return kr_window_of_run((uint8_t)c, S.w) % S.p == 0;
        return true;
    };
    std::vector<char> keep_whole(n_bins, 0);        // run bins whose slices would not fit a batch either: produced as whole bins
    auto sliceable = [&](uint32_t b) {
        const uint32_t sym = run_symbol(b);
        return capped && sym != 0 && !keep_whole[b] && !run_triggers(sym) && !std::getenv("MMT_GUIDED_NO_SLICES");
    };
    uint64_t largest = 0, largest_rep = 0, share = 0, share_rep = 0, largest_any = 0;
    for (uint32_t b = bin_lo; b < bin_hi; b++) { largest_any = std::max(largest_any, bins[b]); share += bins[b]; share_rep += rbins[b]; }
    const uint64_t head_room = capped ? std::min<uint64_t>(SS.ext0, largest_any) : largest_any;      // (uncapped: no bin is sliced)
    const double per_rep = (double)Batch::bytes_per_element() + 12.0;              // batch scratch (holds the tables) + LCP, sege, fb_group
    const double per_out = (W ? 10.0 : 9.0) + 0.05;                                // one window set (+ the emitter's tile records)
    // (what the batches may take: four fifths of what is free, and no more than brings the heap to three quarters of the device --
    // a share of whole genomes holds 130 GB of text, tables of the parse, anchor ranks and thresholds before its first batch, and
    // the scan's candidates, the rows and the writers still come on top)
    // (a run whose rows leave with their windows -- the MEM modes of a rank of configs[4]: text and tables are 200 GB before the
    // first batch -- keeps nothing that grows: the heap may go to nine tenths.  MMT_EXPAND_HEAP_FRAC: tuning aid)
    const double free_now = (double)pool::available(device_), live_now = (double)pool::stats(device_).live;
    // (0.82, 0.76 until round 6: a share of 13 REALISTIC whole genomes holds 35 GB of giant-phrase tables more than the i.i.d. one before
    // its first batch, and at 0.76 its batches were 312 M representatives -- 64 passes over the text -- where 0.86 gave 723 M, 29
    // passes and 7 s less; 0.82 keeps the peak of such a share near 255 GB of the 288)
    double heap_frac = sink_active_ && sink_discard_ ? 0.90 : 0.82;
    if (const char* c = std::getenv("MMT_EXPAND_HEAP_FRAC")) heap_frac = std::min(0.95, std::max(0.3, std::atof(c)));
    const double avail = std::min(0.80 * free_now, heap_frac * (free_now + live_now) - live_now) - 2.0 * per_out * (double)head_room -
                         8.0 * (double)((n + gk::TILE - 1) / gk::TILE) - 4.0 * 1073741824.0;
    const uint64_t WIN_MAX = 1ull << 31;                                           // (a window and its tail stay well below 2^32 entries)
    uint64_t win_cap = 0, rep_cap = 0, stage_cap = 0;
    bool staged = false;
    struct Piece { uint32_t bin, blo, bhi; uint64_t count, reps; bool slice, first; };
    std::vector<Piece> vb;
    DevBuf<uint64_t> run_lead;
    DevBuf<uint8_t> run_first, run_follow;
    gk::RunSlice run_tab;                    // the tables of the tiles (sym, blo, bhi are set per batch)
    uint32_t sliced_bins = 0;
    const uint32_t n_tiles = (uint32_t)((n + gk::TILE - 1) / gk::TILE);
    for (int attempt = 0;; attempt++) {
    if (attempt > (int)n_bins + 1) throw std::runtime_error("guided sort (expansion): the slices of the run bins do not settle");
    largest = largest_rep = 0;
    for (uint32_t b = bin_lo; b < bin_hi; b++)
        if (!sliceable(b)) { largest = std::max(largest, bins[b]); largest_rep = std::max(largest_rep, rbins[b]); }
    win_cap = std::min<uint64_t>(std::max<uint64_t>(share, 1024), WIN_MAX);
    if ((double)win_cap * per_out > 0.5 * std::max(avail, 0.0)) win_cap = std::max<uint64_t>((uint64_t)(0.5 * std::max(avail, 0.0) / per_out), 1u << 20);
    rep_cap = avail > 0 ? (uint64_t)((avail - (double)win_cap * per_out) / per_rep) : 0;
    rep_cap = std::min<uint64_t>(std::max<uint64_t>(rep_cap, 1u << 20), 1ull << 30);
    rep_cap = std::min<uint64_t>(rep_cap, std::max<uint64_t>(share_rep, 1024));
    if (const char* c = std::getenv("MMT_GUIDED_BATCH")) {                         // (tests: small batches, two or three windows each)
        rep_cap = std::max<uint64_t>(1024, std::strtoull(c, nullptr, 10));
        const double ratio = (double)std::max<uint64_t>(share, 1) / (double)std::max<uint64_t>(share_rep, 1);
        win_cap = std::max<uint64_t>(1024, (uint64_t)(0.4 * (double)rep_cap * ratio));
    }
    // Several batches per pass over the text (gk::stage_fill, as in the plain producer): a batch of representatives is a pass over
    // the WHOLE text, and a rank's share of configs[4] -- 573 G characters, 64 batches -- spent 64 of its 88 s of batches there.
    // Part of the memory becomes a list of the representatives of the next batches' bins (8 bytes each), filled by one pass.
    // (1e12 since the entries are four bytes, 2e13 before: a rank's share of configs[3] -- 79 G characters, 15 - 30 batches -- gains a
    // second of its 14, and three of 29 with realistic content: 4 and 14 passes over the text instead of 15 and 36)
    staged = share_rep > 2 * rep_cap && (double)n * ((double)share_rep / (double)std::max<uint64_t>(rep_cap, 1)) >= 1e12;
    if (const char* c = std::getenv("MMT_GUIDED_STAGE")) staged = std::atoi(c) != 0;
    else if (std::getenv("MMT_GUIDED_BATCH")) staged = share_rep > 2 * rep_cap;      // (tests: every run of several batches)
    stage_cap = 0;
    if (staged) {
        if (std::getenv("MMT_GUIDED_BATCH")) stage_cap = std::min<uint64_t>(share_rep, std::max<uint64_t>(4 * rep_cap, largest_rep));
        else {
            const double for_reps = std::max(avail - (double)win_cap * per_out, 0.0);
            const double stage_bytes = 0.4 * for_reps;
            const uint64_t fit2 = (uint64_t)((for_reps - stage_bytes) / per_rep);
            stage_cap = std::min<uint64_t>(std::min<uint64_t>((uint64_t)(stage_bytes / 4.0), share_rep), 0xfff00000ull);   // (32-bit offsets per tile)
            if (fit2 < (1u << 20) || fit2 < largest_rep || stage_cap < 2 * std::max<uint64_t>(largest_rep, 1)) staged = false;
            else rep_cap = std::min<uint64_t>(rep_cap, fit2);
        }
        if (stage_cap < largest_rep) staged = false;
    }
    if (largest > win_cap || largest_rep > rep_cap) {
        const double need = per_out * (double)largest + per_rep * (double)largest_rep;
        if (need > std::max(avail, 0.0) + per_out * (double)(1u << 20) + per_rep * (double)(1u << 20) || largest >= 0xf0000000ull ||
            largest_rep >= 0xffffff00ull)                 // (the scratch is sized rep_cap + 64 in 32 bits: stay clear of the wrap)
            throw std::runtime_error("guided sort (expansion): " + std::to_string(largest) + " suffixes (" + std::to_string(largest_rep) +
                                     " representatives) share their first " + std::to_string(prefix_chars) +
                                     " characters: more than one batch can hold on this device");
        win_cap = std::max(win_cap, largest); rep_cap = std::max(rep_cap, largest_rep);
    }
    // ---- the pieces the batches and windows are made of: whole bins, and the slices of run bins that exceed a batch or a window ----
    vb.clear(); sliced_bins = 0;
    bool again = false;
    {
        const uint64_t lim_sfx = slice_env ? std::min<uint64_t>(slice_env, win_cap) : win_cap, lim_rep = slice_env ? std::min<uint64_t>(slice_env, rep_cap) : rep_cap;
        for (uint32_t b = bin_lo; b < bin_hi; b++) {
            const bool cut = sliceable(b) && (bins[b] > lim_sfx || rbins[b] > lim_rep);
            if (!cut) { vb.push_back(Piece{b, 0, 0, bins[b], rbins[b], false, true}); continue; }
            if (!run_tab.n_tiles) {
                // per tile: the run that begins at its first position -- within the tile from a pass over the text, across tiles
                // by a chain over the tiles that hold one symbol only (on the host: 19 M tiles of 79 G characters in 20 ms)
                DevBuf<uint16_t> lead16;
                lead16.ensure(n_tiles); run_first.ensure((size_t)n_tiles + 1); run_follow.ensure((size_t)n_tiles + 1); run_lead.ensure((size_t)n_tiles + 1);
                gk::Ctx plain = ctx;
                plain.repbits = nullptr;
                gk::tile_lead(plain, run_first.get(), lead16.get(), run_follow.get(), st);
                std::vector<uint16_t> l16(n_tiles);
                std::vector<uint8_t> fi(n_tiles), fo(n_tiles);
                MMT_HIP(hipMemcpyAsync(l16.data(), lead16.get(), (size_t)n_tiles * 2, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipMemcpyAsync(fi.data(), run_first.get(), n_tiles, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipMemcpyAsync(fo.data(), run_follow.get(), n_tiles, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipStreamSynchronize(st));
                std::vector<uint64_t> lead(n_tiles);
                for (uint32_t t = n_tiles; t-- > 0;) {
                    if (l16[t] < gk::TILE) lead[t] = l16[t];
                    else if (t + 1 < n_tiles && fi[t + 1] == fi[t]) { lead[t] = (uint64_t)gk::TILE + lead[t + 1]; fo[t] = fo[t + 1]; }
                    else { lead[t] = gk::TILE; fo[t] = t + 1 < n_tiles ? fi[t + 1] : (uint8_t)0; }
                }
                MMT_HIP(hipMemcpyAsync(run_lead.get(), lead.data(), (size_t)n_tiles * 8, hipMemcpyHostToDevice, st));
                MMT_HIP(hipMemcpyAsync(run_follow.get(), fo.data(), n_tiles, hipMemcpyHostToDevice, st));
                MMT_HIP(hipStreamSynchronize(st));
                run_tab.lead = run_lead.get(); run_tab.first = run_first.get(); run_tab.follow = run_follow.get(); run_tab.n_tiles = n_tiles;
            }
            const uint32_t NB = 2u * gk::RUN_BUCKETS_HALF;
            DevBuf<uint64_t> d_hist;
            d_hist.ensure(2 * (size_t)NB);
            MMT_HIP(hipMemsetAsync(d_hist.get(), 0, 2 * (size_t)NB * 8, st));
            gk::RunSlice rs = run_tab;
            rs.sym = run_symbol(b);
            gk::run_hist(ctx, prefix_chars, b, rs, d_hist.get(), st);
            std::vector<uint64_t> h;
            d2h(h, d_hist.get(), 2 * (size_t)NB, st);
            uint64_t sum = 0, sum_rep = 0;
            for (uint32_t q = 0; q < NB; q++) { sum += h[q]; sum_rep += h[NB + q]; }
            if (sum != bins[b] || sum_rep != rbins[b])
                throw std::runtime_error("guided sort: the slices of a run bin hold " + std::to_string(sum) + " suffixes (" + std::to_string(sum_rep) +
                                         " representatives), its bin " + std::to_string(bins[b]) + " (" + std::to_string(rbins[b]) + ")");
            // (a bucket -- suffixes with (nearly) the same length of their run left: every AAAA of a random text has four -- that
            // exceeds a batch by itself: the bin goes as a whole after all, and the capacities are worked out again with it)
            bool fits = true;
            for (uint32_t q = 0; q < NB; q++) fits = fits && h[q] <= lim_sfx && h[NB + q] <= lim_rep;
            if (!fits) { keep_whole[b] = 1; again = true; break; }
            Piece cur{b, 0, 0, 0, 0, true, true};
            for (uint32_t q = 0; q < NB; q++) {
                if (cur.count && (cur.count + h[q] > lim_sfx || cur.reps + h[NB + q] > lim_rep)) {
                    cur.bhi = q; vb.push_back(cur);
                    cur = Piece{b, q, 0, 0, 0, true, false};
                }
                cur.count += h[q]; cur.reps += h[NB + q];
            }
            cur.bhi = NB; vb.push_back(cur);
            sliced_bins++;
            if (stats) std::fprintf(stderr, "[guided] bin %u (one symbol, %llu suffixes, %llu representatives) goes in %zu slices by what is left of the runs\n",
                                    b, (unsigned long long)bins[b], (unsigned long long)rbins[b],
                                    (size_t)std::count_if(vb.begin(), vb.end(), [&](const Piece& x) { return x.bin == b; }));
        }
    }
    if (!again) break;
    }
    mem_mark(device_, "expansion: before the batches");
    Batch X;
    if (rep_cap > 0xffffff00ull) throw std::runtime_error("guided sort (expansion): a batch of " + std::to_string(rep_cap) + " representatives");
    X.reserve((uint32_t)std::max<uint64_t>(rep_cap, 1024) + 64);
    X.cap = (uint32_t)std::max<uint64_t>(rep_cap, 1024);
    const size_t C = (size_t)X.cap + 64;
    DevBuf<uint32_t> L;                      // LCP of every representative of the batch with the one before it
    L.ensure(C);
    // The tables of the emitter are views of the batch's sort scratch (only pos_b, the sorted records, outlives the sort); they
    // are forgotten again before the scratch goes (a DevBuf that borrows never frees, but a later run must not find them).
    struct Views {
        PfpState& S;
        ~Views() {
            S.ce_cnt.release(); S.ce_first.release(); S.ce_offm1.release(); S.ce_gs.release(); S.ce_hl.release(); S.ce_slen.release();
            S.gscan.release(); S.ce_bwt.release(); S.ce_eoff.release(); S.segb.release(); S.ghead.release();
            S.gctx.repbits = nullptr;
            // what only this stream needed goes with it, also when it fails (a run that does not fit is repeated as anchor
            // partitions: they must find the device as empty as the first attempt did)
            S.occ.release(); S.occ_sl.release(); S.occ12.release(); S.ptab.release(); S.g_repbits.release();
            S.sege.release(); S.fb_group.release(); S.fb_size.release(); S.fb_off.release(); S.fb_start.release();
            S.tile_first.release(); S.emit_plan.release(); S.xk_a.release(); S.xk_b.release(); S.xv_a.release(); S.xv_b.release();
            S.fb_rel.release(); S.fb_chunk0.release();
            S.emit_ready = false;
        }
    } views{S};
    auto borrow_tables = [&]() {
        S.ce_eoff.borrow(X.key_a.get(), C, W);
        S.segb.borrow(X.key_b.get(), C, W);
        S.ghead.borrow(reinterpret_cast<uint32_t*>(X.pos_a.get()), 2 * C);
        S.ce_slen.borrow(reinterpret_cast<uint32_t*>(X.pos_c.get()), C);
        S.gscan.borrow(reinterpret_cast<uint32_t*>(X.pos_c.get()) + C, C);
        S.ce_cnt.borrow(X.slot_a.get(), C); S.ce_first.borrow(X.slot_b.get(), C); S.ce_offm1.borrow(X.ghead.get(), C);
        S.ce_gs.borrow(X.hv.get(), C); S.ce_hl.borrow(X.idx.get(), C);
        S.ce_bwt.borrow(X.head.get(), C);
    };
    S.sege.ensure(C + 2); S.fb_group.ensure(C + 2);
    window_reserve(0, head_room + win_cap + 16);
    // the tail of the window before (its last `head_room` entries at most)
    DevBuf<uint32_t> t_sa, t_lcp;
    DevBuf<uint8_t> t_hi, t_bwt;
    t_sa.ensure(head_room + 16); t_lcp.ensure(head_room + 16); t_bwt.ensure(head_room + 16);
    if (W) t_hi.ensure(head_room + 16);
    DevBuf<uint64_t> carry;
    carry.ensure(2);
    DevBuf<uint32_t> tile_cnt, tile_off;
    tile_cnt.ensure((size_t)n_tiles + 1); tile_off.ensure((size_t)n_tiles + 1);
    DevBuf<uint32_t> stage, blk_tile;
    DevBuf<uint32_t> blk_cnt, blk_off;
    if (staged) { stage.ensure(stage_cap + 16); blk_cnt.ensure(stage_cap / 4096 + 2); blk_off.ensure(stage_cap / 4096 + 2); blk_tile.ensure(stage_cap / 4096 + 3); }
    const uint64_t anchor = std::min<uint64_t>(doc_len_[0], n);
    const RmqView rmq = S.plcp.view();
    S.emit_ready = true;

    const uint32_t nv = (uint32_t)vb.size();
    run_slices_ = 0;
    for (const Piece& x : vb) run_slices_ += x.slice ? 1 : 0;
    uint64_t base = pre[bin_lo], active_sum = 0, small_sum = 0, reps_done = 0;
    const uint64_t piece_end = pre[bin_hi];
    int batches = 0, windows = 0, rounds_max = 0;
    uint64_t prev_len = 0;                 // entries of the window before (without a virtual closing entry), the tail buffers hold its end
    uint64_t tail_len = 0;
    uint32_t prev_last_bin = 0;
    bool have_prev = false;
    uint32_t counted_lo = 0, counted_hi = 0;
    double ms_sort = 0, ms_emit = 0;
    // (pieces [b0, b1): whole bins, or slices of ONE run bin -- never both in a batch; `stop`: the end of the staged pieces)
    auto next_batch_end = [&](uint32_t b0, uint32_t stop, uint64_t& total, uint64_t& total_rep) {
        uint32_t b1 = b0;
        total = 0; total_rep = 0;
        while (b1 < stop && vb[b1].slice == vb[b0].slice && (!vb[b0].slice || vb[b1].bin == vb[b0].bin) &&
               total_rep + vb[b1].reps <= X.cap && total + vb[b1].count < (1ull << 40)) { total += vb[b1].count; total_rep += vb[b1].reps; b1++; }
        return b1;
    };
    // the kernels' view of pieces [b0, b1): a range of real bins + the range of buckets of a run bin's slices
    auto real_range = [&](uint32_t b0, uint32_t b1, uint32_t& lo, uint32_t& hi, gk::RunSlice& out) {
        lo = hi = 0; out = gk::RunSlice();
        if (b0 >= b1) return;
        lo = vb[b0].bin; hi = vb[b1 - 1].bin + 1;
        if (vb[b0].slice) { out = run_tab; out.sym = run_symbol(vb[b0].bin); out.blo = vb[b0].blo; out.bhi = vb[b1 - 1].bhi; }
    };
    uint32_t pass_end = 0;                     // staged: the whole-bin pieces [.., pass_end) are in the list
    uint32_t taken_lo = 0, taken_hi = 0;       // ... and blk_cnt holds the per-block counts of the list's entries of the pieces [taken_lo - 1, taken_hi - 1)
    uint64_t n_staged = 0;
    int passes = 0;
    for (uint32_t b0 = 0; b0 < nv;) {
        const bool from_list = staged && !vb[b0].slice;
        if (from_list && b0 >= pass_end) {     // the next pass over the text: as many whole bins as the list holds
            pass_end = b0; n_staged = 0;
            while (pass_end < nv && !vb[pass_end].slice && n_staged + vb[pass_end].reps <= stage_cap) n_staged += vb[pass_end++].reps;
            if (pass_end == b0) throw std::runtime_error("guided sort (expansion): a bin exceeds the staging list");
            taken_lo = taken_hi = 0;
            if (n_staged) {
                uint32_t lo, hi, nlo, nhi;
                gk::RunSlice none;
                real_range(b0, pass_end, lo, hi, none);
                if (!(counted_lo == b0 + 1 && counted_hi == pass_end + 1)) gk::batch_count(ctx, prefix_chars, lo, hi, tile_cnt.get(), st);
                prims::exclusive_sum_u32(d_temp_, tile_cnt.get(), tile_off.get(), n_tiles, st);
                uint32_t next_end = pass_end;
                uint64_t next_total = 0;
                while (next_end < nv && !vb[next_end].slice && next_total + vb[next_end].reps <= stage_cap) next_total += vb[next_end++].reps;
                const bool more = next_total > 0;
                real_range(pass_end, next_end, nlo, nhi, none);
                gk::stage_fill(ctx, prefix_chars, lo, hi, tile_off.get(), stage.get(), nlo, more ? nhi : nlo, more ? tile_cnt.get() : nullptr, st);
                gk::stage_block_tiles(tile_off.get(), n_tiles, n_staged, blk_tile.get(), st);
                counted_lo = more ? pass_end + 1 : 0; counted_hi = more ? next_end + 1 : 0;
                passes++;
            }
        }
        uint64_t total = 0, total_rep = 0;
        const uint32_t b_stop = from_list ? pass_end : nv;
        const uint32_t b1 = next_batch_end(b0, b_stop, total, total_rep);
        if (b1 == b0) throw std::runtime_error("guided sort (expansion): a bin exceeds the batch");
        if (total && !total_rep) throw std::runtime_error("guided sort (expansion): suffixes without a representative");
        if (!total) { b0 = b1; continue; }
        const uint32_t B = (uint32_t)total_rep;
        auto t_a = now();
        // ---- collect and sort the representatives of the pieces [b0, b1) ----
        uint32_t r_lo, r_hi, nr_lo, nr_hi;
        gk::RunSlice rsl, nrs;
        real_range(b0, b1, r_lo, r_hi, rsl);
        uint64_t nt = 0, ntr = 0;
        if (from_list) {
            const uint32_t nb = (uint32_t)((n_staged + 4095) / 4096);
            // (the batch before counted this batch's entries per block while it took its own)
            if (!(taken_lo == b0 + 1 && taken_hi == b1 + 1)) gk::stage_count(stage.get(), n_staged, r_lo, r_hi, blk_cnt.get(), st);
            prims::exclusive_sum_u32(d_temp_, blk_cnt.get(), blk_off.get(), nb, st);
            const uint32_t nb1 = b1 < pass_end ? next_batch_end(b1, pass_end, nt, ntr) : b1;
            real_range(b1, nb1, nr_lo, nr_hi, nrs);
            const bool more = ntr > 0;
            gk::stage_take(ctx, stage.get(), n_staged, r_lo, r_hi, blk_off.get(), X.key_a.get(), X.pos_a.get(), nr_lo, more ? nr_hi : nr_lo,
                           more ? blk_cnt.get() : nullptr, tile_off.get(), blk_tile.get(), st);
            taken_lo = more ? b1 + 1 : 0; taken_hi = more ? nb1 + 1 : 0;
        } else {
            // (the batch before counted this batch's representatives per tile while it collected its own)
            if (!(counted_lo == b0 + 1 && counted_hi == b1 + 1)) { gk::batch_count(ctx, prefix_chars, r_lo, r_hi, tile_cnt.get(), st, rsl); if (rsl.sym) passes++; }
            prims::exclusive_sum_u32(d_temp_, tile_cnt.get(), tile_off.get(), n_tiles, st);
            const uint32_t nb1 = b1 < nv ? next_batch_end(b1, nv, nt, ntr) : b1;
            real_range(b1, nb1, nr_lo, nr_hi, nrs);
            const bool more = nt > 0 && !rsl.sym && !nrs.sym && !staged;      // (a batch of slices counts for itself)
            gk::batch_fill(ctx, prefix_chars, r_lo, r_hi, tile_off.get(), X.key_a.get(), X.pos_a.get(), nr_lo, more ? nr_hi : nr_lo,
                           more ? tile_cnt.get() : nullptr, st, rsl);
            counted_lo = more ? b1 + 1 : 0; counted_hi = more ? nb1 + 1 : 0;
            passes++;
            if (vb[b0].slice) pass_end = b1;
        }
        RoundStats rs = sort_batch(X, B, ctx, d_temp_, S.err.get(), st, L.get(), &rmq);
        gk::batch_lcp(ctx, rmq, X.pos_b.get(), B, carry.get(), have_prev, L.get(), S.err.get(), st);
        MMT_HIP(hipMemcpyAsync(carry.get(), X.pos_b.get() + (B - 1), 8, hipMemcpyDeviceToDevice, st));
        // ---- the emitter's tables: one entry per representative, in suffix-array order of the phrase suffixes ----
        borrow_tables();
        gk::expand_entries(ctx, X.pos_b.get(), L.get(), B, S.ptab.get(), S.ce_cnt.get(), S.ce_first.get(), S.ce_offm1.get(),
                           S.ce_bwt.get(), S.ce_gs.get(), S.ce_hl.get(), S.ce_slen.get(), S.err.get(), st);
        guided_check_errors("representatives");            // (synchronises; the emitter's tables reuse the error words)
        prims::inclusive_sum_u32(d_temp_, S.ce_gs.get(), S.gscan.get(), B, st);
        const uint32_t G = read_u32(S.gscan.get() + (B - 1), st);
        gk::group_ids(S.ce_gs.get(), S.gscan.get(), B, st);
        if (W) prims::exclusive_sum_u32_to_u64(d_temp_, S.ce_cnt.get(), S.ce_eoff.p64(), B, st);
        else prims::exclusive_sum_u32(d_temp_, S.ce_cnt.get(), S.ce_eoff.p32(), B, st);
        const uint64_t out_lo = base + 1;                   // (stream entry j + 1 = suffix-array entry j; entry 0 is the end sentinel)
        gk::add_offset(S.ce_eoff.get(), W, B, out_lo, st);
        {
            const uint64_t expanded = S.ce_eoff.read(B - 1, st) + read_u32(S.ce_cnt.get() + (B - 1), st) - out_lo;
            if (expanded != total)
                throw std::runtime_error("expansion: the representatives of a batch stand for " + std::to_string(expanded) +
                                         " suffixes, its bins hold " + std::to_string(total));
        }
        uint32_t head0 = 0;
        if (have_prev) { MMT_HIP(hipMemcpyAsync(&head0, L.get(), 4, hipMemcpyDeviceToHost, st)); MMT_HIP(hipStreamSynchronize(st)); }
        pfp_group_tables(B, G, out_lo, out_lo + total, false);
        // (group 0 of a batch has a predecessor in the batch before: its LCP came with the carry)
        if (have_prev) MMT_HIP(hipMemcpyAsync(S.ghead.get() + 1, &head0, 4, hipMemcpyHostToDevice, st));
        if (stats) { MMT_HIP(hipStreamSynchronize(st)); ms_sort += std::chrono::duration<double, std::milli>(now() - t_a).count(); }
        auto t_b = now();
        // ---- the windows of the batch: [tail of the window before | whole bins | one virtual closing entry at the end of a rank's share] ----
        for (uint32_t s0 = b0; s0 < b1;) {
            uint64_t wtotal = 0;
            uint32_t s1 = s0;
            while (s1 < b1 && wtotal + vb[s1].count <= win_cap) wtotal += vb[s1++].count;
            if (s1 == s0) throw std::runtime_error("guided sort (expansion): a bin exceeds the window");
            if (!wtotal) { s0 = s1; continue; }
            EventPair& ee = next_range_event(SS, 3);
            ee.start(st);
            uint64_t ext = 0;
            // (what an interval of this window can reach of the window before: the rest of ITS bin -- of the whole run bin when the
            // window before ended in one of its slices)
            if (have_prev) ext = std::min<uint64_t>(std::min<uint64_t>(bins[vb[prev_last_bin].bin], prev_len), capped ? SS.ext0 : ~0ull);
            if (ext > head_room || ext > tail_len) throw std::runtime_error("guided sort: window head room too small");
            if (ext) {
                const uint64_t from = tail_len - ext;
                MMT_HIP(hipMemcpyAsync(w_sa_[0].get(), t_sa.get() + from, ext * 4, hipMemcpyDeviceToDevice, st));
                if (W) MMT_HIP(hipMemcpyAsync(w_hi_[0].get(), t_hi.get() + from, ext, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(w_bwt_[0].get(), t_bwt.get() + from, ext, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(w_lcp_[0].get(), t_lcp.get() + from, ext * 4, hipMemcpyDeviceToDevice, st));
            }
            S.first_tile.clear();
            // (stream entry base + 1 begins a bin or a slice, hence a group: its tile is where the window's groups begin; what the tile holds
            // of the window before is written once more, into the tail, with the same values)
            S.first_tile[base - ext] = (base + 1) / pk::emit_tile();
            pfp_emit_window(base - ext, base + wtotal, 0);
            ee.stop(st);
            {
                uint32_t e16[16];
                MMT_HIP(hipMemcpyAsync(e16, S.err.get(), 64, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipStreamSynchronize(st));
                if (e16[0]) {
                    char msg[400];
                    std::snprintf(msg, sizeof(msg), "expansion: the emitter's order is inconsistent in window %d: %u entries (text position past the end: "
                                  "%u in tile groups, %u / %u in oversized groups; neighbours of a group without ascending parse ranks: %u)",
                                  windows, e16[0], e16[5], e16[6], e16[7], e16[3]);
                    throw std::runtime_error(msg);
                }
            }
            stream_entries_ += wtotal;
            uint64_t len = ext + wtotal;
            // what the next window may need of this one, before anything else touches the window
            tail_len = std::min<uint64_t>(len, head_room);
            if (tail_len) {
                const uint64_t from = len - tail_len;
                MMT_HIP(hipMemcpyAsync(t_sa.get(), w_sa_[0].get() + from, tail_len * 4, hipMemcpyDeviceToDevice, st));
                if (W) MMT_HIP(hipMemcpyAsync(t_hi.get(), w_hi_[0].get() + from, tail_len, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(t_bwt.get(), w_bwt_[0].get() + from, tail_len, hipMemcpyDeviceToDevice, st));
                MMT_HIP(hipMemcpyAsync(t_lcp.get(), w_lcp_[0].get() + from, tail_len * 4, hipMemcpyDeviceToDevice, st));
            }
            ColWindow w = window_view(0, base - ext, (uint32_t)len, (uint32_t)ext);
            // nothing an interval of this window could reach lies further left (bins) -- but for a window that continues a run bin:
            // there the cap of the mode bounds the reach (ext0), as between the windows of the parse proper
            w.more_left = vb[s0].slice && !vb[s0].first;
            keep_window(w);
            if (want_anchor_ranks_) {
                SaCol piece = w.sa; piece.lo += ext; if (piece.hi) piece.hi += ext;
                k::anchor_ranks(piece, base, wtotal, anchor, wide_ ? (void*)d_rank64_.get() : (void*)d_rank_.get(), st);
            }
            const bool last_of_share = s1 == nv || base + wtotal == piece_end;
            if (last_of_share && base + wtotal < n) {
                MMT_HIP(hipMemsetAsync(w_lcp_[0].get() + len, 0, 4, st));
                MMT_HIP(hipMemsetAsync(w_bwt_[0].get() + len, 0, 1, st));
                MMT_HIP(hipMemsetAsync(w_sa_[0].get() + len, 0, 4, st));
                if (W) MMT_HIP(hipMemsetAsync(w_hi_[0].get() + len, 0, 1, st));
                w.len = (uint32_t)(len + 1);
            }
            if (!scan_window(SS, w, p)) throw std::runtime_error("guided sort: a walk left its bin");
            sink_flush(SS);
            prev_len = len; have_prev = true;
            for (uint32_t b = s1; b-- > s0;) if (vb[b].count) { prev_last_bin = b; break; }
            base += wtotal; windows++;
            s0 = s1;
        }
        MMT_HIP(hipMemsetAsync(S.err.get(), 0, 64, st));
        if (stats) { MMT_HIP(hipStreamSynchronize(st)); ms_emit += std::chrono::duration<double, std::milli>(now() - t_b).count(); }
        if (!batches) mem_mark(device_, "expansion: first batch done");
        reps_done += B; batches++;
        rounds_max = std::max(rounds_max, rs.rounds); active_sum += rs.active_sum; small_sum += rs.small;
        b0 = b1;
    }
    guided_check_errors("text suffixes");
    if (base != piece_end) throw std::runtime_error("guided sort: the batches do not cover the text exactly once");
    S.rounds_dict = rounds_max; S.emit_launches = (uint32_t)windows;
    text_passes_ = (uint32_t)passes; batches_ = (uint32_t)batches; staged_ = staged;
    MMT_HIP(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
    if (stats) std::fprintf(stderr, "[guided] expansion: %d passes over the text%s, %u slices of %u run bins\n", passes,
                            staged ? " (several batches per pass: staging list)" : "", run_slices_, sliced_bins);
    if (stats) std::fprintf(stderr, "[guided] expansion: %llu suffixes from %llu representatives (%llu in the whole text) in %d batches of at most %u "
                            "representatives, %d windows of at most %llu suffixes: %.1f ms (collect + sort + tables %.1f, emitter + scans %.1f); "
                            "%.3f of the representatives settled in small groups, %.3f element-rounds per representative in %d rounds at most\n",
                            (unsigned long long)(piece_end - pre[bin_lo]), (unsigned long long)reps_done, (unsigned long long)reps_total,
                            batches, X.cap, windows, (unsigned long long)win_cap, ms, ms_sort, ms_emit,
                            (double)small_sum / (double)std::max<uint64_t>(1, reps_done),
                            (double)active_sum / (double)std::max<uint64_t>(1, reps_done), rounds_max);
    S.ms[6] = (float)ms;
    sort_rounds_ = rounds_max;
}

}  // namespace mmt
