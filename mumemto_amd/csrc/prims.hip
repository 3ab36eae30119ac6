// prims.hip -- rocPRIM instantiations (see prims.hpp).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "prims.hpp"

namespace mmt { namespace prims {

template <typename F>
static void with_temp(DevBuf<uint8_t>& temp, F&& call) {
    size_t bytes = 0;
    MMT_HIP(call(nullptr, bytes));
    if (bytes == 0) bytes = 16;
    temp.ensure(bytes);
    MMT_HIP(call(temp.get(), bytes));
}

void sort_pairs_u64_u32(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::radix_sort_pairs(t, b, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
    });
}
void sort_pairs_u32_u32(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int begin_bit, int end_bit, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::radix_sort_pairs(t, b, kin, kout, vin, vout, n, (unsigned)begin_bit, (unsigned)end_bit, s);
    });
}
void inclusive_max_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::inclusive_scan(t, b, in, out, n, rocprim::maximum<uint32_t>(), s);
    });
}
void exclusive_sum_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::exclusive_scan(t, b, in, out, uint32_t(0), n, rocprim::plus<uint32_t>(), s);
    });
}
void inclusive_sum_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::inclusive_scan(t, b, in, out, n, rocprim::plus<uint32_t>(), s);
    });
}
void exclusive_sum_u64(DevBuf<uint8_t>& temp, const uint64_t* in, uint64_t* out, size_t n, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::exclusive_scan(t, b, in, out, uint64_t(0), n, rocprim::plus<uint64_t>(), s);
    });
}
void select_indices(DevBuf<uint8_t>& temp, const uint8_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                    hipStream_t s) {
    rocprim::counting_iterator<uint32_t> idx(0);
    with_temp(temp, [&](void* t, size_t& b) { return rocprim::select(t, b, idx, flags, out, d_count, n, s); });
}


void select_indices_u32flags(DevBuf<uint8_t>& temp, const uint32_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                             hipStream_t s) {
    rocprim::counting_iterator<uint32_t> idx(0);
    with_temp(temp, [&](void* t, size_t& b) { return rocprim::select(t, b, idx, flags, out, d_count, n, s); });
}
void segmented_sort_pairs_u32_ranges(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                     uint32_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                     const uint32_t* end, int end_bit, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::segmented_radix_sort_pairs(t, b, kin, kout, vin, vout, n, segments, begin, end, 0u,
                                                   (unsigned)end_bit, s);
    });
}


}}  // namespace mmt::prims
