#include "sorter.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "kernels.hpp"
#include "prims.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

DoublingSorter::~DoublingSorter() {
    if (side_) (void)hipStreamDestroy(side_);
    if (ev_main_) (void)hipEventDestroy(ev_main_);
    if (ev_side_) (void)hipEventDestroy(ev_side_);
}

// Text-sized scratch: the two key columns and one value column, 20 bytes per suffix.  Everything else is either carved out
// of a key column that is dead at that point of sort() or sized by the active set (the suffixes the first sort left tied).
void DoublingSorter::reserve(uint32_t n) {
    keys_a_.ensure(n); keys_b_.ensure(n); sac_a_.ensure(n); count_.ensure(4);
}

// (keys_a_, sac_a_) -> (keys_b_, sac_b_), m active elements grouped by bucket: tiles between bucket boundaries are
// sorted in LDS, the few ranges holding a bucket longer than a tile by one segmented radix sort.
void DoublingSorter::release() {
    keys_a_.release(); keys_b_.release();
    for (DevBuf<uint32_t>* b : {&sac_a_, &sac_b_, &pos_a_, &pos_b_, &headc_, &headc_b_, &bound_, &big_begin_, &big_end_})
        b->release();
    hf_.release(); tile_big_.release();
}

void DoublingSorter::sort_round(uint32_t m, int shift, DevBuf<uint8_t>& temp, hipStream_t s) {
    static const bool global_sort = std::getenv("MMT_SORT_GLOBAL_ROUNDS") != nullptr;     // the old path (tests)
    if (global_sort || m < 4096) {
        prims::sort_pairs_u64_u32(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sac_b_.get(), m, 0,
                                  std::min(64, 2 * shift), s);
        return;
    }
    const uint32_t target = 1024, limit = k::ROUND_TILE_CAP;   // tests/round_sweep: 256 .. 1900 all within 2 %
    const uint32_t n_tiles = (m + target - 1) / target;
    bound_.ensure((size_t)n_tiles + 2);
    uint32_t big_cap = (uint32_t)std::max<size_t>(big_begin_.size(), 4096);
    big_begin_.ensure(big_cap); big_end_.ensure(big_cap);
    MMT_HIP(hipMemsetAsync(count_.get() + 1, 0, 4, s));
    k::round_tile_bounds(keys_a_.get(), m, shift, target, limit, n_tiles, bound_.get(), s);
    k::round_local_sort(keys_a_.get(), sac_a_.get(), keys_b_.get(), sac_b_.get(), bound_.get(), n_tiles, big_begin_.get(),
                        big_end_.get(), count_.get() + 1, big_cap, shift, s);
    uint32_t big = 0;
    MMT_HIP(hipMemcpyAsync(&big, count_.get() + 1, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    if (big > big_cap) {                                   // (rare) more long ranges than listed: sort the round globally
        prims::sort_pairs_u64_u32(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sac_b_.get(), m, 0,
                                  std::min(64, 2 * shift), s);
        return;
    }
    if (big)
        prims::segmented_sort_pairs_u64_ranges(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sac_b_.get(), m, big,
                                               big_begin_.get(), big_end_.get(), std::min(64, 2 * shift), s, true);
}

// The buckets c^chars of the first sort (sorted keys in keys_b_, suffixes in sa), each sorted once more by k_run_keys' key:
// the input keys (dead) take the new keys, the input values (dead) the sorted suffixes; the sorted new keys replace the bucket's
// equal keys, so that mark_heads cuts it where they differ (and at its two ends, `forced`: a new key may happen to equal the
// real key next to the bucket).  Suffixes with a set low bit are final, in position order: the sorts are stable.
void DoublingSorter::refine_runs(uint32_t n, const RunRefine& R, uint32_t* sa, std::vector<uint32_t>& forced,
                                 DevBuf<uint8_t>& temp, hipStream_t s) {
    // (MMT_RUN_BUCKET: the smallest bucket that is worth a sort of its own; the tests lower it)
    const uint32_t min_bucket = std::getenv("MMT_RUN_BUCKET") ? (uint32_t)std::atoi(std::getenv("MMT_RUN_BUCKET")) : 4096u;
    const bool trace = std::getenv("MMT_SORT_TRACE") != nullptr;
    const int S = std::min(R.sigma, 15);
    std::vector<uint64_t> probe(S);
    for (int c = 1; c <= S; c++) {
        uint64_t k = 0;
        for (int i = 0; i < R.chars; i++) k = (k << R.bits) | (uint64_t)c;
        probe[c - 1] = k << 1;                                // (k_pack_keys: no terminator in the window)
    }
    run_probe_.ensure(S); run_range_.ensure(2 * (size_t)S);
    MMT_HIP(hipMemcpyAsync(run_probe_.get(), probe.data(), (size_t)S * 8, hipMemcpyHostToDevice, s));
    k::equal_range_u64(keys_b_.get(), n, run_probe_.get(), (uint32_t)S, run_range_.get(), s);
    std::vector<uint32_t> range(2 * (size_t)S);
    MMT_HIP(hipMemcpyAsync(range.data(), run_range_.get(), range.size() * 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    for (int c = 0; c < S; c++) {
        const uint32_t lo = range[2 * c], hi = range[2 * c + 1];
        if (hi - lo < min_bucket) continue;
        const uint32_t cnt = hi - lo;
        k::run_keys(sa + lo, cnt, R.text, R.n, R.code, R.bits, R.chars, R.ends, R.n_ends, keys_a_.get() + lo, s);
        prims::sort_pairs_u64_u32(temp, keys_a_.get() + lo, keys_b_.get() + lo, sa + lo, sac_a_.get() + lo, cnt, 0, 64, s);
        MMT_HIP(hipMemcpyAsync(sa + lo, sac_a_.get() + lo, (size_t)cnt * 4, hipMemcpyDeviceToDevice, s));
        forced.push_back(lo); forced.push_back(hi);
        run_refined_ += cnt;
        if (trace) std::fprintf(stderr, "[sort] bucket of %u suffixes inside long runs of symbol %d: ordered by (end of the run, what follows)\n", cnt, c + 1);
    }
}

int DoublingSorter::sort(uint32_t n, int key_bits, uint64_t h0, uint32_t* sa, uint32_t* rank, DevBuf<uint8_t>& temp,
                         hipStream_t s, bool lsb_unique, const RunRefine* runs) {
    reserve(n);
    prims::sort_pairs_u64_u32(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sa, n, 0, std::min(64, key_bits), s);
    run_refined_ = 0;
    std::vector<uint32_t> forced;
    if (runs && runs->n_ends && lsb_unique && !std::getenv("MMT_NO_RUN_REFINE")) refine_runs(n, *runs, sa, forced, temp, s);
    // the input keys are dead: their column holds the head marks and the index list; the sorted keys die with mark_heads:
    // their column holds the heads and the flags (n entries each, at fixed places, for every later round as well)
    uint32_t* const headval = reinterpret_cast<uint32_t*>(keys_a_.get());
    uint32_t* const idx = headval + n;
    uint32_t* const head = reinterpret_cast<uint32_t*>(keys_b_.get());
    uint8_t* const flags = reinterpret_cast<uint8_t*>(head + n);
    if (!side_ && !std::getenv("MMT_SORT_ONE_STREAM")) {
        MMT_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
        MMT_HIP(hipEventCreateWithFlags(&ev_main_, hipEventDisableTiming));
        MMT_HIP(hipEventCreateWithFlags(&ev_side_, hipEventDisableTiming));
    }
    k::mark_heads(keys_b_.get(), n, headval, lsb_unique, s);
    if (!forced.empty()) {
        run_range_.ensure(forced.size());
        MMT_HIP(hipMemcpyAsync(run_range_.get(), forced.data(), forced.size() * 4, hipMemcpyHostToDevice, s));
        k::force_heads(headval, run_range_.get(), (uint32_t)forced.size(), n, s);
        MMT_HIP(hipStreamSynchronize(s));                     // (`forced` is host memory of this call)
    }
    prims::inclusive_max_u32(temp, headval, head, n, s);
    // every rank once (random stores: latency) on the second stream, beside the selection of the tied suffixes (streams)
    if (side_) {
        MMT_HIP(hipEventRecord(ev_main_, s));
        MMT_HIP(hipStreamWaitEvent(side_, ev_main_, 0));
        k::scatter_rank(sa, head, n, rank, side_);
        MMT_HIP(hipEventRecord(ev_side_, side_));
    } else k::scatter_rank(sa, head, n, rank, s);
    // (MMT_SORT_FLAG_ARRAY: the byte-flag pass + the selection over it, the two steps this replaced)
    static const bool flag_array = std::getenv("MMT_SORT_FLAG_ARRAY") != nullptr;
    if (flag_array) {
        k::flag_unsorted(head, n, flags, s);
        prims::select_indices(temp, flags, idx, count_.get(), n, s);
    } else prims::select_tied_heads(temp, head, idx, count_.get(), n, s);
    uint32_t m = 0;
    MMT_HIP(hipMemcpyAsync(&m, count_.get(), 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    if (m) {
        pos_a_.ensure(m); pos_b_.ensure(m); headc_.ensure(m); sac_b_.ensure(m);       // the active set only
        k::gather_active(idx, m, sa, head, pos_a_.get(), sac_a_.get(), headc_.get(), s);
    }
    if (side_) MMT_HIP(hipStreamWaitEvent(s, ev_side_, 0));                            // the ranks are in place

    const int shift = bit_width_u64(n);            // second key component holds values 0..n
    static const bool fused_ok = !std::getenv("MMT_SORT_GLOBAL_ROUNDS") &&
                                 !(std::getenv("MMT_SORT_FUSED") && std::atoi(std::getenv("MMT_SORT_FUSED")) == 0);
    static const bool trace = std::getenv("MMT_SORT_TRACE") != nullptr;      // tuning aid: the active set round by round
    if (trace) std::fprintf(stderr, "[sort] n %u, tied after the first pass %u (h = %llu)\n", n, m, (unsigned long long)h0);
    bool side_busy = false;
    uint64_t h = h0;
    int rounds = 0;
    while (m) {
        if (++rounds > 64) throw std::runtime_error("suffix sort did not converge");
        const uint32_t hh = h > 0xffffffffull ? 0xffffffffu : (uint32_t)h;
        const uint32_t* head_r = head;              // the new heads and the still-tied flags of this round
        const uint8_t* flags_r = flags;
        bool done = false;
        if (fused_ok && m >= 4096) {
            // one pass over the active list: gather, sort in LDS, new heads, SA entries (no key column); then the ranks
            const uint32_t limit = k::round_fused_cap(), target = limit / 2;
            const uint32_t n_tiles = (m + target - 1) / target;
            const size_t m4 = ((size_t)m + 15) & ~(size_t)15;
            hf_.ensure(m4 * 5 + 16); bound_.ensure((size_t)n_tiles + 2); tile_big_.ensure((size_t)n_tiles + 1);
            uint32_t* const head_w = reinterpret_cast<uint32_t*>(hf_.get());
            uint8_t* const flags_w = hf_.get() + m4 * 4;
            // room for the list of long ranges (satellite content: thousands per round); MMT_BIG_CAP: a small list, so that the
            // round of separate kernels takes over after the fused pass has run (tests)
            static const uint32_t big_cap_min = std::getenv("MMT_BIG_CAP") ? (uint32_t)std::max(1, std::atoi(std::getenv("MMT_BIG_CAP"))) : 65536u;
            const uint32_t big_cap = std::getenv("MMT_BIG_CAP") ? big_cap_min : (uint32_t)std::max<size_t>(big_begin_.size(), big_cap_min);
            big_begin_.ensure(big_cap); big_end_.ensure(big_cap);
            MMT_HIP(hipMemsetAsync(count_.get() + 1, 0, 4, s));
            MMT_HIP(hipMemsetAsync(tile_big_.get(), 0, (size_t)n_tiles + 1, s));
            k::round_head_bounds(headc_.get(), m, target, limit, n_tiles, bound_.get(), s);
            k::round_fused(sac_a_.get(), headc_.get(), pos_a_.get(), bound_.get(), n_tiles, rank, n, hh, shift, sa,
                           sac_b_.get(), head_w, flags_w, big_begin_.get(), big_end_.get(), count_.get() + 1, big_cap,
                           tile_big_.get(), s);
            uint32_t big = 0;
            MMT_HIP(hipMemcpyAsync(&big, count_.get() + 1, 4, hipMemcpyDeviceToHost, s));
            MMT_HIP(hipStreamSynchronize(s));
            if (trace) std::fprintf(stderr, "[sort]   round %d: h %u, tied %u, long ranges %u\n", rounds, hh, m, big);
            if (big <= big_cap) {                           // (more long ranges than listed: the round below, rare)
                if (big) {
                    // the long ranges: keys in the dead input-key column, sorted into the dead sorted-key column
                    k::round_big_keys(tile_big_.get(), bound_.get(), target, n_tiles, sac_a_.get(), headc_.get(), rank, n, hh,
                                      shift, keys_a_.get(), s);
                    prims::segmented_sort_pairs_u64_ranges(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sac_b_.get(), m,
                                                           big, big_begin_.get(), big_end_.get(), std::min(64, 2 * shift), s, true);
                    k::round_big_subheads(tile_big_.get(), bound_.get(), target, n_tiles, keys_b_.get(), pos_a_.get(), head_w, s);
                    prims::inclusive_max_u32(temp, head_w, head_w, m, s);
                    k::round_big_apply(tile_big_.get(), bound_.get(), target, n_tiles, m, sac_b_.get(), head_w, pos_a_.get(),
                                       sa, flags_w, s);
                }
                // the scatter of the ranks (random stores: latency) runs beside the compaction of the list (streams):
                // both read the sorted list, the compaction writes the other copies of its columns
                static const bool all_ranks = std::getenv("MMT_SORT_ALL_RANKS") != nullptr;        // (A/B: every rank rewritten)
                hipStream_t sb = s;
                if (side_) {
                    MMT_HIP(hipEventRecord(ev_main_, s));
                    MMT_HIP(hipStreamWaitEvent(side_, ev_main_, 0));
                    sb = side_;
                }
                if (all_ranks) k::scatter_rank(sac_b_.get(), head_w, m, rank, sb);
                else k::scatter_rank_changed(sac_b_.get(), head_w, headc_.get(), m, rank, sb);
                if (side_) { MMT_HIP(hipEventRecord(ev_side_, side_)); side_busy = true; }
                head_r = head_w; flags_r = flags_w;
                done = true;
            }
        }
        if (!done) {
            // round keys in keys_a_[0, m) -> sorted in keys_b_[0, m); then the marks over the dead round keys, the heads over
            // the dead sorted keys
            k::make_round_keys(sac_a_.get(), headc_.get(), m, rank, n, hh, shift, keys_a_.get(), s);
            sort_round(m, shift, temp, s);
            k::mark_subheads(keys_b_.get(), pos_a_.get(), m, headval, s);
            prims::inclusive_max_u32(temp, headval, head, m, s);
            k::apply_round(sac_b_.get(), head, pos_a_.get(), m, sa, rank, flags, s);
        }
        prims::select_indices(temp, flags_r, idx, count_.get(), m, s);
        uint32_t m2 = 0;
        MMT_HIP(hipMemcpyAsync(&m2, count_.get(), 4, hipMemcpyDeviceToHost, s));
        MMT_HIP(hipStreamSynchronize(s));
        if (m2) {
            headc_b_.ensure(m2);
            k::compact_round(idx, m2, pos_a_.get(), sac_b_.get(), head_r, pos_b_.get(), sac_a_.get(), headc_b_.get(), s);
            pos_a_.swap(pos_b_);
        }
        if (side_busy) { MMT_HIP(hipStreamWaitEvent(s, ev_side_, 0)); side_busy = false; }      // the ranks are in place
        if (m2) headc_.swap(headc_b_);
        m = m2;
        h *= 2;
    }
    return rounds;
}

}  // namespace mmt
