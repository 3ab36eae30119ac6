#include "sorter.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "kernels.hpp"
#include "prims.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

void DoublingSorter::reserve(uint32_t n) {
    keys_a_.ensure(n); keys_b_.ensure(n);
    sac_a_.ensure(n); sac_b_.ensure(n); pos_a_.ensure(n); pos_b_.ensure(n); headc_.ensure(n);
    headval_.ensure(n); head_.ensure(n); idx_.ensure(n); flags_.ensure(n); count_.ensure(4);
}

bool DoublingSorter::reserve_in(uint8_t* region, size_t bytes, uint32_t n) {
    const size_t A = 512;
    auto up = [&](size_t x) { return (x + A - 1) / A * A; };
    const size_t need = 2 * up((size_t)n * 8) + 8 * up((size_t)n * 4) + up(n) + A;
    if (!region || bytes < need) { reserve(n); return false; }
    uint8_t* at = reinterpret_cast<uint8_t*>(up(reinterpret_cast<uintptr_t>(region)));
    auto take64 = [&](DevBuf<uint64_t>& b) { b.borrow(reinterpret_cast<uint64_t*>(at), n); at += up((size_t)n * 8); };
    auto take32 = [&](DevBuf<uint32_t>& b) { b.borrow(reinterpret_cast<uint32_t*>(at), n); at += up((size_t)n * 4); };
    take64(keys_a_); take64(keys_b_);
    for (DevBuf<uint32_t>* b : {&sac_a_, &sac_b_, &pos_a_, &pos_b_, &headc_, &headval_, &head_, &idx_}) take32(*b);
    flags_.borrow(at, n);
    count_.ensure(4);
    return true;
}

// (keys_a_, sac_a_) -> (keys_b_, sac_b_), m active elements grouped by bucket: tiles between bucket boundaries are
// sorted in LDS, the few ranges holding a bucket longer than a tile by one segmented radix sort.
void DoublingSorter::release() {
    keys_a_.release(); keys_b_.release(); flags_.release();
    for (DevBuf<uint32_t>* b : {&sac_a_, &sac_b_, &pos_a_, &pos_b_, &headc_, &headval_, &head_, &idx_, &bound_, &big_begin_,
                                &big_end_})
        b->release();
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
                                               big_begin_.get(), big_end_.get(), std::min(64, 2 * shift), s);
}

int DoublingSorter::sort(uint32_t n, int key_bits, uint64_t h0, uint32_t* sa, uint32_t* rank, DevBuf<uint8_t>& temp,
                         hipStream_t s, bool lsb_unique) {
    reserve(n);
    prims::sort_pairs_u64_u32(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sa, n, 0, std::min(64, key_bits), s);
    k::mark_heads(keys_b_.get(), n, headval_.get(), lsb_unique, s);
    prims::inclusive_max_u32(temp, headval_.get(), head_.get(), n, s);
    k::scatter_rank(sa, head_.get(), n, rank, s);
    k::flag_unsorted(head_.get(), n, flags_.get(), s);
    prims::select_indices(temp, flags_.get(), idx_.get(), count_.get(), n, s);
    uint32_t m = 0;
    MMT_HIP(hipMemcpyAsync(&m, count_.get(), 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    if (m) k::gather_active(idx_.get(), m, sa, head_.get(), pos_a_.get(), sac_a_.get(), headc_.get(), s);

    const int shift = bit_width_u64(n);            // second key component holds values 0..n
    uint64_t h = h0;
    int rounds = 0;
    while (m) {
        if (++rounds > 64) throw std::runtime_error("suffix sort did not converge");
        const uint32_t hh = h > 0xffffffffull ? 0xffffffffu : (uint32_t)h;
        k::make_round_keys(sac_a_.get(), headc_.get(), m, rank, n, hh, shift, keys_a_.get(), s);
        sort_round(m, shift, temp, s);
        k::mark_subheads(keys_b_.get(), pos_a_.get(), m, headval_.get(), s);
        prims::inclusive_max_u32(temp, headval_.get(), head_.get(), m, s);
        k::apply_round(sac_b_.get(), head_.get(), pos_a_.get(), m, sa, rank, flags_.get(), s);
        prims::select_indices(temp, flags_.get(), idx_.get(), count_.get(), m, s);
        uint32_t m2 = 0;
        MMT_HIP(hipMemcpyAsync(&m2, count_.get(), 4, hipMemcpyDeviceToHost, s));
        MMT_HIP(hipStreamSynchronize(s));
        if (m2) {
            k::compact_round(idx_.get(), m2, pos_a_.get(), sac_b_.get(), head_.get(), pos_b_.get(), sac_a_.get(),
                             headc_.get(), s);
            pos_a_.swap(pos_b_);
        }
        m = m2;
        h *= 2;
    }
    return rounds;
}

}  // namespace mmt
