// prims.hip -- rocPRIM instantiations (see prims.hpp).
#include <algorithm>
#include <cstdio>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <vector>
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
bool sort_pairs_u64_u64_inplace(DevBuf<uint8_t>& temp, uint64_t* ka, uint64_t* kb, uint64_t* va, uint64_t* vb, size_t n,
                                int begin_bit, int end_bit, hipStream_t s) {
    rocprim::double_buffer<uint64_t> keys(ka, kb), vals(va, vb);
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::radix_sort_pairs(t, b, keys, vals, n, (unsigned)begin_bit, (unsigned)end_bit, s);
    });
    return keys.current() == kb;
}
// Running maximum in two passes over the data (workgroup maxima, a small scan of those, then the scan proper with
// the carry-in): rocprim's single-pass look-back scan reaches 1.1 TB/s on 387 M elements with this operator.
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_block_max(const uint32_t* __restrict__ in, size_t n,
                                                     uint32_t* __restrict__ block_max) {
    __shared__ uint32_t s_w[BLOCK / 64];
    const size_t base = (size_t)blockIdx.x * BLOCK * ITEMS;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {                  // coalesced: consecutive threads read consecutive elements
        const size_t i = base + (size_t)q * BLOCK + threadIdx.x;
        if (i < n) { const uint32_t v = in[i]; m = v > m ? v : m; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t y = __shfl_xor(m, o, 64); m = y > m ? y : m; }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        for (int w = 0; w < BLOCK / 64; w++) r = s_w[w] > r ? s_w[w] : r;
        block_max[blockIdx.x] = r;
    }
}
template <int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_block_max_scan(const uint32_t* in, size_t n,
                                                          const uint32_t* __restrict__ carry /* exclusive */,
                                                          uint32_t* out /* may be `in`: every thread reads i, then writes i */) {
    __shared__ uint32_t s_w[BLOCK / 64];
    const size_t base = (size_t)blockIdx.x * BLOCK * ITEMS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = carry[blockIdx.x];
    for (int q = 0; q < ITEMS; q++) {                  // ITEMS rounds of BLOCK consecutive elements
        const size_t i = base + (size_t)q * BLOCK + threadIdx.x;
        uint32_t v = i < n ? in[i] : 0u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(v, o, 64); if (lane >= (uint32_t)o && y > v) v = y; }
        if (lane == 63) s_w[wave] = v;
        __syncthreads();
        uint32_t pre = run;
        for (uint32_t w2 = 0; w2 < wave; w2++) pre = s_w[w2] > pre ? s_w[w2] : pre;
        v = v > pre ? v : pre;
        if (i < n) out[i] = v;
        uint32_t tot = run;
        for (int w2 = 0; w2 < BLOCK / 64; w2++) tot = s_w[w2] > tot ? s_w[w2] : tot;
        run = tot;
        __syncthreads();
    }
}
void inclusive_max_u32(DevBuf<uint8_t>& temp, const uint32_t* in, uint32_t* out, size_t n, hipStream_t s) {
    constexpr int BLOCK = 256, ITEMS = 16;
    if (n < (size_t)1 << 22) {                          // small inputs: the library scan
        with_temp(temp, [&](void* t, size_t& b) {
            return rocprim::inclusive_scan(t, b, in, out, n, rocprim::maximum<uint32_t>(), s);
        });
        return;
    }
    const uint32_t blocks = (uint32_t)((n + (size_t)BLOCK * ITEMS - 1) / ((size_t)BLOCK * ITEMS));
    size_t scan_bytes = 0;
    MMT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, uint32_t(0), blocks,
                                    rocprim::maximum<uint32_t>(), s));
    const size_t head = ((size_t)blocks * 8 + 255) & ~(size_t)255;
    temp.ensure(head + scan_bytes + 256);
    uint32_t* bmax = reinterpret_cast<uint32_t*>(temp.get());
    uint32_t* carry = bmax + blocks;
    hipLaunchKernelGGL((k_block_max<BLOCK, ITEMS>), dim3(blocks), dim3(BLOCK), 0, s, in, n, bmax);
    MMT_HIP(rocprim::exclusive_scan(temp.get() + head, scan_bytes, bmax, carry, uint32_t(0), blocks,
                                    rocprim::maximum<uint32_t>(), s));
    hipLaunchKernelGGL((k_block_max_scan<BLOCK, ITEMS>), dim3(blocks), dim3(BLOCK), 0, s, in, n, carry, out);
    MMT_HIP(hipGetLastError());
}
// segmented minimum: elements are (head flag << 32) | value; a set flag starts a new segment
struct SegMinU64 {
    __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const {
        if (b >> 32) return b;
        const uint32_t x = (uint32_t)a, y = (uint32_t)b;
        return (a & 0xffffffff00000000ull) | (uint64_t)(y < x ? y : x);
    }
};
void inclusive_segmin_u64(DevBuf<uint8_t>& temp, const uint64_t* in, uint64_t* out, size_t n, hipStream_t s) {
    if (!n) return;
    with_temp(temp, [&](void* t, size_t& b) { return rocprim::inclusive_scan(t, b, in, out, n, SegMinU64(), s); });
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
struct WidenU32 {
    __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; }
};
void exclusive_sum_u32_to_u64(DevBuf<uint8_t>& temp, const uint32_t* in, uint64_t* out, size_t n, hipStream_t s) {
    auto it = rocprim::make_transform_iterator(in, WidenU32());
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::exclusive_scan(t, b, it, out, uint64_t(0), n, rocprim::plus<uint64_t>(), s);
    });
}
void exclusive_sum_u64(DevBuf<uint8_t>& temp, const uint64_t* in, uint64_t* out, size_t n, hipStream_t s) {
    with_temp(temp, [&](void* t, size_t& b) {
        return rocprim::exclusive_scan(t, b, in, out, uint64_t(0), n, rocprim::plus<uint64_t>(), s);
    });
}
// ---- stream compaction of flagged indices (hand-written: rocprim::select moves 0.2 TB/s on this shape) ----
// Pass 1: every thread turns ITEMS consecutive flags into a bit mask and the workgroup counts them; an exclusive
// scan of the workgroup counts follows; pass 2 rebuilds the masks and writes the indices in order.
// the sixteen flags of a work-item as a bit mask: 16-byte loads where the flags allow them (sixteen byte loads a work-item
// ran the byte-flag passes of the suffix sorter at 85 GB/s: 69 ms of kernels per C3 step)
template <typename F>
__device__ __forceinline__ uint32_t flags16(const F* __restrict__ flags, size_t i0, size_t n, bool aligned) {
    uint32_t mask = 0;
    if (aligned && i0 + 16 <= n) {
        if constexpr (sizeof(F) == 1) {
            const uint4 v = *reinterpret_cast<const uint4*>(flags + i0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t nz = (w[k] | ((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;      // high bit of every non-zero byte
                mask |= (((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u)) << (4 * k);
            }
            return mask;
        } else if constexpr (sizeof(F) == 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = reinterpret_cast<const uint4*>(flags + i0)[k];
                mask |= ((v.x ? 1u : 0u) | (v.y ? 2u : 0u) | (v.z ? 4u : 0u) | (v.w ? 8u : 0u)) << (4 * k);
            }
            return mask;
        }
    }
#pragma unroll
    for (int q = 0; q < 16; q++) if (i0 + q < n && flags[i0 + q]) mask |= 1u << q;
    return mask;
}
// what is selected: a flag array, or a predicate over another column (no flag array is written and read back then)
template <typename F>
struct FlagArray {
    const F* flags; size_t n;
    __device__ __forceinline__ uint32_t operator()(size_t i0) const { return flags16(flags, i0, n, (reinterpret_cast<uintptr_t>(flags) & 15u) == 0); }
};
// the suffix sorter's "still tied after the first sort": position j is NOT a bucket of its own, i.e. not (head[j] == j and
// (j + 1 == n or head[j + 1] == j + 1))
struct TiedHeads {
    const uint32_t* head; size_t n;
    __device__ __forceinline__ uint32_t operator()(size_t i0) const {
        uint32_t own = 0;                                       // bit q: head[i0 + q] == i0 + q (bit 16: the position behind the sixteen)
        if (i0 + 17 <= n && (reinterpret_cast<uintptr_t>(head) & 15u) == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 v = reinterpret_cast<const uint4*>(head + i0)[k];
                const uint32_t b = (uint32_t)i0 + 4u * k;
                own |= ((v.x == b ? 1u : 0u) | (v.y == b + 1 ? 2u : 0u) | (v.z == b + 2 ? 4u : 0u) | (v.w == b + 3 ? 8u : 0u)) << (4 * k);
            }
            own |= (head[i0 + 16] == (uint32_t)i0 + 16u ? 1u : 0u) << 16;
        } else {
#pragma unroll
            for (int q = 0; q <= 16; q++) {
                const size_t j = i0 + q;
                if (j < n ? head[j] == (uint32_t)j : j == n) own |= 1u << q;      // (the position behind the last one counts as its own bucket)
            }
        }
        uint32_t mask = ~(own & (own >> 1)) & 0xffffu;
        if (i0 + 16 > n) mask &= i0 < n ? (1u << (n - i0)) - 1u : 0u;
        return mask;
    }
};
template <typename Sel, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_flag_counts(const Sel sel, size_t n, uint32_t* __restrict__ block_count) {
    static_assert(ITEMS == 16, "flags16");
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const size_t i0 = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * ITEMS;
    uint32_t c = __popc(sel(i0));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s_cnt;
}
template <typename Sel, int BLOCK, int ITEMS>
__global__ __launch_bounds__(BLOCK) void k_flag_write(const Sel sel, size_t n,
                                                      const uint32_t* __restrict__ block_off,
                                                      const uint32_t* __restrict__ block_count, uint32_t n_blocks,
                                                      uint32_t* __restrict__ out, uint32_t* __restrict__ d_count) {
    __shared__ uint32_t s_wave[BLOCK / 64];
    const size_t i0 = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * ITEMS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t mask = sel(i0);
    const uint32_t c = __popc(mask);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t at = block_off[blockIdx.x] + inc - c;
    for (uint32_t wv = 0; wv < wave; wv++) at += s_wave[wv];
    while (mask) {
        const uint32_t b = __builtin_ctz(mask);
        out[at++] = (uint32_t)(i0 + b);
        mask &= mask - 1;
    }
    if (blockIdx.x == n_blocks - 1 && threadIdx.x == 0) *d_count = block_off[n_blocks - 1] + block_count[n_blocks - 1];
}
template <typename Sel>
static void select_flagged(DevBuf<uint8_t>& temp, const Sel sel, uint32_t* out, uint32_t* d_count, size_t n,
                           hipStream_t s) {
    constexpr int BLOCK = 256, ITEMS = 16;
    if (n == 0) { MMT_HIP(hipMemsetAsync(d_count, 0, 4, s)); return; }
    const uint32_t blocks = (uint32_t)((n + (size_t)BLOCK * ITEMS - 1) / ((size_t)BLOCK * ITEMS));
    // temp = [workgroup counts | their exclusive scan | scratch of the scan]
    size_t scan_bytes = 0;
    MMT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, uint32_t(0), blocks,
                                    rocprim::plus<uint32_t>(), s));
    const size_t head = ((size_t)blocks * 8 + 255) & ~(size_t)255;
    temp.ensure(head + scan_bytes + 256);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(temp.get());
    uint32_t* off = cnt + blocks;
    hipLaunchKernelGGL((k_flag_counts<Sel, BLOCK, ITEMS>), dim3(blocks), dim3(BLOCK), 0, s, sel, n, cnt);
    MMT_HIP(rocprim::exclusive_scan(temp.get() + head, scan_bytes, cnt, off, uint32_t(0), blocks,
                                    rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL((k_flag_write<Sel, BLOCK, ITEMS>), dim3(blocks), dim3(BLOCK), 0, s, sel, n, off, cnt, blocks, out,
                       d_count);
    MMT_HIP(hipGetLastError());
}
void select_indices(DevBuf<uint8_t>& temp, const uint8_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                    hipStream_t s) {
    select_flagged(temp, FlagArray<uint8_t>{flags, n}, out, d_count, n, s);
}
void select_indices_u32flags(DevBuf<uint8_t>& temp, const uint32_t* flags, uint32_t* out, uint32_t* d_count, size_t n,
                             hipStream_t s) {
    select_flagged(temp, FlagArray<uint32_t>{flags, n}, out, d_count, n, s);
}
void select_tied_heads(DevBuf<uint8_t>& temp, const uint32_t* head, uint32_t* out, uint32_t* d_count, size_t n, hipStream_t s) {
    select_flagged(temp, TiedHeads{head, n}, out, d_count, n, s);
}
// Ranges [begin[i], end[i]) of one array, each sorted by the key bits below end_bit.  rocPRIM's segmented sort gives a
// range to ONE workgroup however long it is (~30 ns per element: a homopolymer or a run of N puts millions of suffixes
// into one bucket of a doubling round, and such a round then takes hundreds of milliseconds), so the range list is read
// on the host and a range beyond GIANT elements gets a device-wide radix sort of its own; the others share one
// segmented sort as before.  MMT_GIANT_RANGE overrides the threshold (tests).
// Ranges whose keys already say which range an element belongs to -- the rounds of the suffix sorter: key = (head of the bucket
// << shift) | rank of the suffix h characters on, buckets in suffix-array order -- need no segmented sort at all: the elements
// of ALL ranges are gathered into one compact array, sorted by ONE device-wide radix sort, and written back range by range
// (range r keeps the slots [off[r], off[r + 1]) of the sorted array: every key of a range is above every key of the range
// before).  A round of a realistic dictionary lists thousands of long ranges of very different lengths; one sort per giant
// range plus a segmented sort for the rest was 2,400 radix passes, 13,000 merge kernels and 4,000 memsets per step (~390 ms
// of kernels, profiles/round5_realistic_*): this is three kernels and one sort per round.
template <typename K, typename V>
__global__ void k_ranges_gather(const K* __restrict__ kin, const V* __restrict__ vin, const uint32_t* __restrict__ begin,
                                const uint32_t* __restrict__ off, uint32_t ranges, uint32_t total, K* __restrict__ gk,
                                V* __restrict__ gv) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    uint32_t lo = 0, hi = ranges;                               // last range with off[r] <= c
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t src = begin[lo] + (c - off[lo]);
    gk[c] = kin[src]; gv[c] = vin[src];
}
template <typename K, typename V>
__global__ void k_ranges_scatter(const K* __restrict__ gk, const V* __restrict__ gv, const uint32_t* __restrict__ begin,
                                 const uint32_t* __restrict__ off, uint32_t ranges, uint32_t total, K* __restrict__ kout,
                                 V* __restrict__ vout) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    uint32_t lo = 0, hi = ranges;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t dst = begin[lo] + (c - off[lo]);
    kout[dst] = gk[c]; vout[dst] = gv[c];
}
template <typename K, typename V>
static bool sort_ranges_as_one(DevBuf<uint8_t>& temp, const K* kin, K* kout, const V* vin, V* vout,
                               std::vector<uint32_t>& hb, std::vector<uint32_t>& he, int end_bit, hipStream_t s) {
    const uint32_t R = (uint32_t)hb.size();
    std::vector<uint32_t> order(R);
    for (uint32_t i = 0; i < R; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hb[a] < hb[b]; });
    std::vector<uint32_t> tab(2 * (size_t)R + 1);             // begin[R] | off[R + 1]
    uint64_t total = 0;
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t r = order[i];
        if (i && hb[r] < he[order[i - 1]]) return false;      // (overlapping ranges: not this path)
        tab[i] = hb[r]; tab[R + i] = (uint32_t)total;
        total += he[r] - hb[r];
    }
    tab[2 * (size_t)R] = (uint32_t)total;
    static const uint64_t limit = std::getenv("MMT_RANGES_AS_ONE_MAX") ? std::strtoull(std::getenv("MMT_RANGES_AS_ONE_MAX"), nullptr, 10)
                                                                      : (uint64_t)400000000ull;
    if (total == 0 || total > limit) return false;
    const uint32_t T = (uint32_t)total;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t kb = up((size_t)T * sizeof(K)), vb = up((size_t)T * sizeof(V)), tb = up(tab.size() * 4);
    size_t bytes = 0;
    {
        rocprim::double_buffer<K> dk(nullptr, nullptr);
        rocprim::double_buffer<V> dv(nullptr, nullptr);
        MMT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, dk, dv, (size_t)T, 0u, (unsigned)end_bit, s));
    }
    const size_t head = up(bytes);
    temp.ensure(head + 2 * kb + 2 * vb + tb + 256);
    uint8_t* base = temp.get() + head;
    K* k0 = reinterpret_cast<K*>(base); K* k1 = reinterpret_cast<K*>(base + kb);
    V* v0 = reinterpret_cast<V*>(base + 2 * kb); V* v1 = reinterpret_cast<V*>(base + 2 * kb + vb);
    uint32_t* dtab = reinterpret_cast<uint32_t*>(base + 2 * kb + 2 * vb);
    MMT_HIP(hipMemcpyAsync(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    const uint32_t blocks = (T + 255) / 256;
    hipLaunchKernelGGL((k_ranges_gather<K, V>), dim3(blocks), dim3(256), 0, s, kin, vin, dtab, dtab + R, R, T, k0, v0);
    rocprim::double_buffer<K> dk(k0, k1);
    rocprim::double_buffer<V> dv(v0, v1);
    MMT_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, dk, dv, (size_t)T, 0u, (unsigned)end_bit, s));
    hipLaunchKernelGGL((k_ranges_scatter<K, V>), dim3(blocks), dim3(256), 0, s, dk.current(), dv.current(), dtab, dtab + R, R, T,
                       kout, vout);
    MMT_HIP(hipGetLastError());
    MMT_HIP(hipStreamSynchronize(s));                         // (`tab` is host memory of this call)
    return true;
}

// 32-bit keys that do NOT say which range they belong to (the oversized groups of the emitter: key = parse rank) get the
// range's ordinal in front of them for the one sort: 64-bit keys (ordinal << 32) | key in the compact array, the low word
// written back.
template <typename V>
__global__ void k_ranges_gather_tagged(const uint32_t* __restrict__ kin, const V* __restrict__ vin, const uint32_t* __restrict__ begin,
                                       const uint32_t* __restrict__ off, uint32_t ranges, uint32_t total, uint64_t* __restrict__ gk,
                                       V* __restrict__ gv) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    uint32_t lo = 0, hi = ranges;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t src = begin[lo] + (c - off[lo]);
    gk[c] = ((uint64_t)lo << 32) | kin[src]; gv[c] = vin[src];
}
template <typename V>
__global__ void k_ranges_scatter_tagged(const uint64_t* __restrict__ gk, const V* __restrict__ gv, const uint32_t* __restrict__ begin,
                                        const uint32_t* __restrict__ off, uint32_t ranges, uint32_t total, uint32_t* __restrict__ kout,
                                        V* __restrict__ vout) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    uint32_t lo = 0, hi = ranges;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
    const uint32_t dst = begin[lo] + (c - off[lo]);
    kout[dst] = (uint32_t)gk[c]; vout[dst] = gv[c];
}
template <typename V>
static bool sort_ranges_as_one_tagged(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const V* vin, V* vout,
                                      std::vector<uint32_t>& hb, std::vector<uint32_t>& he, int end_bit, hipStream_t s) {
    const uint32_t R = (uint32_t)hb.size();
    std::vector<uint32_t> order(R);
    for (uint32_t i = 0; i < R; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hb[a] < hb[b]; });
    std::vector<uint32_t> tab(2 * (size_t)R + 1);
    uint64_t total = 0;
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t r = order[i];
        if (i && hb[r] < he[order[i - 1]]) return false;
        tab[i] = hb[r]; tab[R + i] = (uint32_t)total;
        total += he[r] - hb[r];
    }
    tab[2 * (size_t)R] = (uint32_t)total;
    static const uint64_t limit = std::getenv("MMT_RANGES_AS_ONE_MAX") ? std::strtoull(std::getenv("MMT_RANGES_AS_ONE_MAX"), nullptr, 10)
                                                                      : (uint64_t)400000000ull;
    if (total == 0 || total > limit) return false;
    const uint32_t T = (uint32_t)total;
    int rbits = 1;
    while ((1ull << rbits) < (uint64_t)R) rbits++;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t kb = up((size_t)T * 8), vb = up((size_t)T * sizeof(V)), tb = up(tab.size() * 4);
    size_t bytes = 0;
    {
        rocprim::double_buffer<uint64_t> dk(nullptr, nullptr);
        rocprim::double_buffer<V> dv(nullptr, nullptr);
        MMT_HIP(rocprim::radix_sort_pairs(nullptr, bytes, dk, dv, (size_t)T, 0u, 32u + (unsigned)rbits, s));
    }
    const size_t head = up(bytes);
    temp.ensure(head + 2 * kb + 2 * vb + tb + 256);
    uint8_t* base = temp.get() + head;
    uint64_t* k0 = reinterpret_cast<uint64_t*>(base); uint64_t* k1 = reinterpret_cast<uint64_t*>(base + kb);
    V* v0 = reinterpret_cast<V*>(base + 2 * kb); V* v1 = reinterpret_cast<V*>(base + 2 * kb + vb);
    uint32_t* dtab = reinterpret_cast<uint32_t*>(base + 2 * kb + 2 * vb);
    MMT_HIP(hipMemcpyAsync(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    const uint32_t blocks = (T + 255) / 256;
    hipLaunchKernelGGL((k_ranges_gather_tagged<V>), dim3(blocks), dim3(256), 0, s, kin, vin, dtab, dtab + R, R, T, k0, v0);
    rocprim::double_buffer<uint64_t> dk(k0, k1);
    rocprim::double_buffer<V> dv(v0, v1);
    // (the key bits at and above end_bit are not part of the order inside a range: they are zero in the callers' keys or must
    // be ignored -- the bits between end_bit and 32 are left out of the sort by two calls when there are any)
    if (end_bit >= 32) {
        MMT_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, dk, dv, (size_t)T, 0u, 32u + (unsigned)rbits, s));
    } else {
        MMT_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, dk, dv, (size_t)T, 0u, (unsigned)end_bit, s));
        MMT_HIP(rocprim::radix_sort_pairs(temp.get(), bytes, dk, dv, (size_t)T, 32u, 32u + (unsigned)rbits, s));
    }
    hipLaunchKernelGGL((k_ranges_scatter_tagged<V>), dim3(blocks), dim3(256), 0, s, dk.current(), dv.current(), dtab, dtab + R, R, T,
                       kout, vout);
    MMT_HIP(hipGetLastError());
    MMT_HIP(hipStreamSynchronize(s));
    return true;
}

template <typename K, typename V>
static void sort_ranges(DevBuf<uint8_t>& temp, const K* kin, K* kout, const V* vin, V* vout, uint32_t n,
                        uint32_t segments, const uint32_t* begin, const uint32_t* end, int end_bit, hipStream_t s,
                        bool keys_order_the_ranges = false) {
    static const uint32_t GIANT = std::getenv("MMT_GIANT_RANGE") ? (uint32_t)std::atoi(std::getenv("MMT_GIANT_RANGE")) : 65536u;
    if (!segments) return;
    std::vector<uint32_t> hb(segments), he(segments);
    MMT_HIP(hipMemcpyAsync(hb.data(), begin, (size_t)segments * 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipMemcpyAsync(he.data(), end, (size_t)segments * 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    // (MMT_RANGES_AS_ONE=0: the older route -- a sort per giant range, a segmented sort for the rest --, A/B and tests)
    static const bool as_one = !(std::getenv("MMT_RANGES_AS_ONE") && std::atoi(std::getenv("MMT_RANGES_AS_ONE")) == 0);
    if (keys_order_the_ranges && as_one && sort_ranges_as_one(temp, kin, kout, vin, vout, hb, he, end_bit, s)) return;
    if constexpr (sizeof(K) == 4) {
        // many ranges, or any range one workgroup would be busy with for long: one sort of tagged keys
        bool any_long = false;
        for (uint32_t i = 0; i < segments && !any_long; i++) any_long = he[i] - hb[i] > GIANT / 2;
        if (!keys_order_the_ranges && as_one && any_long &&
            sort_ranges_as_one_tagged(temp, kin, kout, vin, vout, hb, he, end_bit, s)) return;
    }
    // a range is "giant" when one workgroup would still be busy with it long after the other ranges, which share the
    // chip a few hundred at a time, are done: beyond GIANT elements and four times the 256th longest range
    uint32_t thresh = GIANT;
    if (segments > 256 && !std::getenv("MMT_GIANT_RANGE")) {
        std::vector<uint32_t> len(segments);
        for (uint32_t i = 0; i < segments; i++) len[i] = he[i] - hb[i];
        std::nth_element(len.begin(), len.begin() + 255, len.end(), std::greater<uint32_t>());
        if (len[255] > thresh / 4) thresh = len[255] >= 0x3fffffffu ? 0xffffffffu : 4 * len[255];
    }
    std::vector<uint32_t> giant, rb, re;
    for (uint32_t i = 0; i < segments; i++)
        if (he[i] - hb[i] > thresh) giant.push_back(i);
    if (std::getenv("MMT_RANGE_STATS")) {                    // tuning aid
        uint64_t total = 0;
        uint32_t longest = 0;
        for (uint32_t i = 0; i < segments; i++) { total += he[i] - hb[i]; longest = std::max(longest, he[i] - hb[i]); }
        std::fprintf(stderr, "[ranges] %zu-byte keys: %u ranges, %llu elements, longest %u, threshold %u, %zu device-wide\n",
                     sizeof(K), segments, (unsigned long long)total, longest, thresh, giant.size());
    }
    if (giant.empty()) {
        with_temp(temp, [&](void* t, size_t& b) {
            return rocprim::segmented_radix_sort_pairs(t, b, kin, kout, vin, vout, n, segments, begin, end, 0u,
                                                       (unsigned)end_bit, s);
        });
        return;
    }
    for (uint32_t i : giant) {
        const uint32_t b0 = hb[i], len = he[i] - hb[i];
        with_temp(temp, [&](void* t, size_t& b) {
            return rocprim::radix_sort_pairs(t, b, kin + b0, kout + b0, vin + b0, vout + b0, len, 0u, (unsigned)end_bit, s);
        });
    }
    if (giant.size() == segments) return;
    size_t g = 0;
    for (uint32_t i = 0; i < segments; i++) {
        if (g < giant.size() && giant[g] == i) { g++; continue; }
        rb.push_back(hb[i]); re.push_back(he[i]);
    }
    // the range list without the giants sits behind rocPRIM's scratch in `temp`
    const uint32_t rest = (uint32_t)rb.size();
    size_t bytes = 0;
    MMT_HIP(rocprim::segmented_radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, rest, begin, end, 0u,
                                                (unsigned)end_bit, s));
    const size_t head = (bytes + 255) & ~(size_t)255;
    temp.ensure(head + (size_t)rest * 8 + 16);
    uint32_t* db = reinterpret_cast<uint32_t*>(temp.get() + head);
    uint32_t* de = db + rest;
    MMT_HIP(hipMemcpyAsync(db, rb.data(), (size_t)rest * 4, hipMemcpyHostToDevice, s));
    MMT_HIP(hipMemcpyAsync(de, re.data(), (size_t)rest * 4, hipMemcpyHostToDevice, s));
    MMT_HIP(hipStreamSynchronize(s));
    MMT_HIP(rocprim::segmented_radix_sort_pairs(temp.get(), bytes, kin, kout, vin, vout, n, rest, db, de, 0u,
                                                (unsigned)end_bit, s));
}

void segmented_sort_pairs_u32_ranges(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint32_t* vin,
                                     uint32_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                     const uint32_t* end, int end_bit, hipStream_t s) {
    sort_ranges(temp, kin, kout, vin, vout, n, segments, begin, end, end_bit, s);
}

void segmented_sort_pairs_u32_u64vals_ranges(DevBuf<uint8_t>& temp, const uint32_t* kin, uint32_t* kout, const uint64_t* vin,
                                             uint64_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                             const uint32_t* end, int end_bit, hipStream_t s) {
    sort_ranges(temp, kin, kout, vin, vout, n, segments, begin, end, end_bit, s);
}

void segmented_sort_pairs_u64_ranges(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                                     uint32_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                     const uint32_t* end, int end_bit, hipStream_t s, bool keys_order_the_ranges) {
    sort_ranges(temp, kin, kout, vin, vout, n, segments, begin, end, end_bit, s, keys_order_the_ranges);
}

void segmented_sort_pairs_u64_u64vals_ranges(DevBuf<uint8_t>& temp, const uint64_t* kin, uint64_t* kout, const uint64_t* vin,
                                             uint64_t* vout, uint32_t n, uint32_t segments, const uint32_t* begin,
                                             const uint32_t* end, int end_bit, hipStream_t s) {
    sort_ranges(temp, kin, kout, vin, vout, n, segments, begin, end, end_bit, s);
}

}}  // namespace mmt::prims
