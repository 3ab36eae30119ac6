// prims.hpp -- device-wide library primitives (rocPRIM) used as building blocks:
// LSD radix sort of (u64 key, u32 value) pairs, prefix max / sum, flagged
// select.  Everything domain-specific is hand-written in kernels.hip.
#pragma once
#include <cstddef>
#include <cstdint>

#include "device_utils.hpp"

namespace mmt { namespace prims {

void sort_pairs_u64_u32(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int begin_bit, int end_bit, hipStream_t s);
void sort_pairs_u32_u32(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int begin_bit, int end_bit, hipStream_t s);
// the pairs are sorted between the two buffer pairs (no library copy of the input); true: the result is in (kb, vb)
bool sort_pairs_u64_u64_inplace(DevBuf<uint8_t>& temp, uint64_t* ka, uint64_t* kb, uint64_t* va, uint64_t* vb, size_t n,
                                int begin_bit, int end_bit, hipStream_t s);
// out may be the same array as in
void inclusive_max_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s);
// elements (head flag << 32) | value: out[i] = (flag of the segment's head, minimum of the values from that head up to i); out may be in
void inclusive_segmin_u64(DevBuf<uint8_t>& temp, const uint64_t* in, uint64_t* out, size_t n, hipStream_t s);
void exclusive_sum_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s);
void inclusive_sum_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s);
void exclusive_sum_u32_to_u64(DevBuf<uint8_t>& temp, const uint32_t* in, uint64_t* out, size_t n, hipStream_t s);
void exclusive_sum_u64(DevBuf<uint8_t>& temp, const uint64_t* in, uint64_t* out, size_t n, hipStream_t s);
// out[k] = index i of the k-th set flag; *d_count = number of set flags
void select_indices(DevBuf<uint8_t>& temp, const uint8_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                    hipStream_t s);


// out[k] = k-th position j that is not a bucket of its own: not (head[j] == j and (j + 1 == n or head[j + 1] == j + 1)) --
// the suffixes the sorter's first pass left tied, selected from the head column itself (no flag array in between)
void select_tied_heads(DevBuf<uint8_t>& temp, const uint32_t* head, uint32_t* out, uint32_t* d_count, size_t n, hipStream_t s);
void select_indices_u32flags(DevBuf<uint8_t>& temp, const uint32_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                             hipStream_t s);
void segmented_sort_pairs_u32_ranges(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                     uint32_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                     const uint32_t* end, int end_bit, hipStream_t s);

void segmented_sort_pairs_u32_u64vals_ranges(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint64_t* vin,
                                             uint64_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                             const uint32_t* end, int end_bit, hipStream_t s);

// 64-bit keys, ranges [begin[i], end[i]) of one array; elements outside the ranges are not touched.
// keys_order_the_ranges: every key of a range is above every key of the ranges that begin before it (the sorter's round keys
// carry the bucket in their high bits): all ranges then go through ONE device-wide sort (prims.hip, sort_ranges_as_one)
void segmented_sort_pairs_u64_ranges(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                                     uint32_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                     const uint32_t* end, int end_bit, hipStream_t s, bool keys_order_the_ranges = false);

void segmented_sort_pairs_u64_u64vals_ranges(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint64_t* vin,
                                             uint64_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                             const uint32_t* end, int end_bit, hipStream_t s);

}}  // namespace mmt::prims
