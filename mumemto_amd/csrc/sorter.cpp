#include "sorter.hpp"

#include <algorithm>
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
        prims::sort_pairs_u64_u32(temp, keys_a_.get(), keys_b_.get(), sac_a_.get(), sac_b_.get(), m, 0,
                                  std::min(64, 2 * shift), s);
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
