// pfp_kernels.hip -- prefix-free parsing on the GPU (rows A2-A4).
//
// A2  newscan.hpp: KR_window::addchar (:106-114), trigger test (:321),
//     save_update_word (:265-307), finish_parse (:357-423).
// A3  dictionary.hpp:103-157 (SA + LCP of the dictionary), parse.hpp:77-146
//     (SA / ISA of the parse), pfp.hpp:171-244.
// A4  pfp_lcp_mum.hpp:115-231: order of the text suffixes = (rank of the proper
//     phrase suffix of length >= w, rank of the parse suffix that follows).
//
// All kernels address the virtual text V = Dollar . T . Dollar^w (V[i+1] = T[i]),
// exactly the string the reference parser consumes (newscan.hpp:248, :359).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "pfp_kernels.hpp"

namespace mmt { namespace pk {

// A launch may not have 2^32 work-items or more (HIP folds the product of grid and workgroup size into 32 bits: a
// larger launch silently runs a fraction of its workgroups).  Every kernel here uses workgroups of at most 256
// work-items with grid_for, so 2^24 workgroups is the limit; kernels over text-sized ranges handle 4 - 16 items per
// work-item and stay below it for any text that fits the device.
static inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}

// ---- A2: trigger positions ---------------------------------------------------------
// hash_i = sum_{k<w} T[i-k] * 256^k mod prime (T[<0] = 0): the value KR_window holds after
// addchar(T[i]) when its window starts zero-filled and is never reset (newscan.hpp:96-114).
// A phrase ends at i iff hash_i % p == 0 and the accumulated word is longer than w
// (newscan.hpp:266), which holds for every trigger with i >= w - 1.
// Pass 1: one 16-bit mask per thread (16 consecutive positions) + the number of triggers per workgroup.
// Pass 2 (after an exclusive scan of the workgroup counts): the trigger positions, ascending.
constexpr uint32_t KR_PRIME = 1999999973u;             // newscan.hpp:86 (compile-time: reductions become multiplies)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trigger_masks(const TextRef T, uint64_t n, uint32_t w,
                                                         uint32_t p, uint32_t pot,
                                                         uint16_t* __restrict__ masks,
                                                         uint32_t* __restrict__ block_count, uint32_t block0) {
    constexpr int PER = 16;
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint64_t blk = (uint64_t)blockIdx.x + block0;
    const uint64_t t = blk * BLOCK + threadIdx.x;
    const uint64_t i0 = t * PER;
    uint32_t mask = 0;
    if (i0 < n) {
        uint32_t h = 0;                                    // always < KR_PRIME < 2^31
        for (uint32_t k = 0; k < w; k++) {                 // window ending at i0
            int64_t pos = (int64_t)i0 - (int64_t)w + 1 + k;
            uint32_t c = pos >= 0 ? tx_byte(T, (uint64_t)pos + 1) : 0;
            h = (uint32_t)(((uint64_t)h * 256 + c) % KR_PRIME);
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint64_t i = i0 + q;
            if (i >= n) break;
            if (i + 1 >= w && h % p == 0) mask |= 1u << q;
            // roll to i + 1: drop T[i-w+1], add T[i+1]
            const int64_t drop = (int64_t)i - (int64_t)w + 1;
            const uint32_t out = drop >= 0 ? tx_byte(T, (uint64_t)drop + 1) : 0;
            const uint32_t in = i + 1 < n ? tx_byte(T, i + 2) : 0;
            h = (uint32_t)((h + KR_PRIME - (uint32_t)(((uint64_t)out * pot) % KR_PRIME)) % KR_PRIME);
            h = (uint32_t)(((uint64_t)h * 256 + in) % KR_PRIME);
        }
        masks[t] = (uint16_t)mask;
    }
    uint32_t c = __popc(mask);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blk] = s_cnt;
}
// The same with everything that made the first version compute-bound taken out of the per-character path (it ran at 0.4
// TB/s on a 1 B / character stream):
//   * the 16 + w + 1 bytes a work-item needs are four 16-byte loads (consecutive work-items on consecutive 16 bytes: fully
//     coalesced; the overlap is served by the caches) kept in registers -- W is a template parameter, every byte index is
//     a compile-time constant;
//   * no 64-bit modulo: (h - out * 256^(w-1)) * 256 + in  mod prime  needs out * 256^(w-1) mod prime and
//     (h >> 23) * 2^31 mod prime, both functions of ONE BYTE: two 256-entry tables in LDS; the rest is h < prime < 2^31
//     arithmetic with conditional subtractions;
//   * "h % p == 0" for the runtime modulus p = 2^e * q, q odd: the low e bits are zero and (h >> e) * q^-1 mod 2^32 <=
//     (2^32 - 1) / q (Granlund-Montgomery divisibility test): one multiply instead of a division.
template <int BLOCK, int W>
__global__ __launch_bounds__(BLOCK) void k_trigger_masks_fast(const TextRef T, uint64_t n, uint32_t pot,
                                                              uint32_t pe, uint32_t qinv, uint32_t qlim,
                                                              uint16_t* __restrict__ masks,
                                                              uint32_t* __restrict__ block_count, uint32_t block0) {
    constexpr int PER = 16;
    static_assert(W >= 1 && W <= 32, "window");
    __shared__ uint32_t s_cnt;
    __shared__ uint32_t s_out[256], s_hi[256];
    {
        const uint32_t x = threadIdx.x & 255u;
        if (threadIdx.x < 256) {
            s_out[x] = (uint32_t)(((uint64_t)x * pot) % KR_PRIME);
            s_hi[x] = (uint32_t)(((uint64_t)x << 31) % KR_PRIME);
        }
        if (threadIdx.x == 0) s_cnt = 0;
    }
    __syncthreads();
    const uint64_t blk = (uint64_t)blockIdx.x + block0;
    const uint64_t t = blk * BLOCK + threadIdx.x;
    const uint64_t i0 = t * PER;
    uint32_t mask = 0;
    if (i0 < n) {
        // by[32 + k] = T[i0 + k] for k = -32 .. 16 (0 before the text and from n on)
        union { uint4 v[4]; uint8_t b[64]; } u;
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        u.v[0] = i0 >= 32 ? tx_load16(T, i0 - 32) : zero;
        u.v[1] = i0 >= 16 ? tx_load16(T, i0 - 16) : zero;
        u.v[2] = tx_load16(T, i0);                          // (the text is padded behind its end)
        u.v[3] = i0 + 16 < n ? tx_load16(T, i0 + 16) : zero;
        const uint64_t left = n - i0;                       // characters of this work-item's 17 that exist
        uint32_t h = 0;                                     // always < KR_PRIME < 2^31
#pragma unroll
        for (int k = 0; k < W; k++) {                       // window ending at i0
            const uint32_t c = u.b[32 - W + 1 + k];
            const uint32_t x = s_hi[h >> 23] + ((h & 0x7fffffu) << 8) + c;     // = h * 256 + c (mod prime), < 2.1 prime
            const uint32_t y = x >= KR_PRIME ? x - KR_PRIME : x;
            h = y >= KR_PRIME ? y - KR_PRIME : y;
        }
        const uint32_t emask = (1u << pe) - 1u;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            if ((uint64_t)q < left) {
                const bool hit = (h & emask) == 0 && (h >> pe) * qinv <= qlim;
                if (hit && i0 + q + 1 >= (uint64_t)W) mask |= 1u << q;
                // roll to i + 1: drop T[i-w+1], add T[i+1]
                const uint32_t out = u.b[32 + q - W + 1];
                const uint32_t in = (uint64_t)(q + 1) < left ? u.b[32 + q + 1] : 0u;
                uint32_t g = h + KR_PRIME - s_out[out];
                g = g >= KR_PRIME ? g - KR_PRIME : g;
                const uint32_t x = s_hi[g >> 23] + ((g & 0x7fffffu) << 8) + in;
                const uint32_t y = x >= KR_PRIME ? x - KR_PRIME : x;
                h = y >= KR_PRIME ? y - KR_PRIME : y;
            }
        }
        masks[t] = (uint16_t)mask;
    }
    uint32_t c = __popc(mask);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blk] = s_cnt;
}
template <int BLOCK, typename P>
__global__ __launch_bounds__(BLOCK) void k_trigger_cuts(const uint16_t* __restrict__ masks, uint64_t n_threads,
                                                        const uint32_t* __restrict__ block_off,
                                                        P* __restrict__ cuts, uint32_t block0) {
    __shared__ uint32_t s_wave[BLOCK / 64];
    const uint64_t blk = (uint64_t)blockIdx.x + block0;
    const uint64_t t = blk * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t mask = t < n_threads ? masks[t] : 0u;
    const uint32_t c = __popc(mask);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t out = block_off[blk] + inc - c;
    for (uint32_t wv = 0; wv < wave; wv++) out += s_wave[wv];
    while (mask) {
        const uint32_t b = __builtin_ctz(mask);
        cuts[out++] = (P)(t * 16 + b);
        mask &= mask - 1;
    }
}
// (workgroups of 256 work-items, 16 positions each; launched in slices of 2^23 workgroups: a launch may not have 2^32
// work-items, and the anchor next to twelve whole-genome haplotypes is 79 G characters)
uint32_t trigger_blocks(uint64_t n) {
    const uint64_t b = ((n + 15) / 16 + 255) / 256;
    if (b >= 0xffffffffull) throw HipError("text too long for the trigger pass");
    return (uint32_t)(b ? b : 1);
}
template <typename F>
static void for_trigger_slices(uint64_t n, F&& launch) {
    const uint32_t blocks = trigger_blocks(n), SLICE = 1u << 23;
    for (uint32_t b0 = 0; b0 < blocks; b0 += SLICE) launch(b0, std::min<uint32_t>(SLICE, blocks - b0));
}
void trigger_masks(const TextRef& text, uint64_t n, uint32_t w, uint32_t p, uint16_t* masks, uint32_t* block_count,
                   hipStream_t s) {
    uint64_t pot = 1;
    for (uint32_t i = 1; i < w; i++) pot = (pot * 256) % KR_PRIME;
    // p = 2^e * q, q odd: q^-1 mod 2^32 by Newton steps, limit (2^32 - 1) / q
    uint32_t pe = 0, q = p;
    while ((q & 1u) == 0) { q >>= 1; pe++; }
    uint32_t qinv = q;                                       // correct to 3 bits
    for (int i = 0; i < 5; i++) qinv *= 2u - q * qinv;
    const uint32_t qlim = 0xffffffffu / q;
    // (text.v = V: the text itself begins one byte on, 16-byte aligned -- Engine::text_ptr)
    const bool fast = !getenv("MMT_TRIGGER_PLAIN") && (text.is_packed() || (reinterpret_cast<uintptr_t>(text.v + 1) & 15u) == 0);
#define MMT_TRIG(WW) for_trigger_slices(n, [&](uint32_t b0, uint32_t cnt) { \
        hipLaunchKernelGGL((k_trigger_masks_fast<256, WW>), dim3(cnt), dim3(256), 0, s, text, n, (uint32_t)pot, pe, qinv, qlim, \
                           masks, block_count, b0); })
    if (fast && w == 6) MMT_TRIG(6);
    else if (fast && w == 10) MMT_TRIG(10);
    else if (fast && w == 14) MMT_TRIG(14);
    else if (fast && w == 4) MMT_TRIG(4);
    else if (fast && w == 8) MMT_TRIG(8);
    else if (fast && w == 12) MMT_TRIG(12);
    else if (fast && w == 16) MMT_TRIG(16);
    else
        for_trigger_slices(n, [&](uint32_t b0, uint32_t cnt) {
            hipLaunchKernelGGL(k_trigger_masks<256>, dim3(cnt), dim3(256), 0, s, text, n, w, p, (uint32_t)pot, masks, block_count, b0);
        });
#undef MMT_TRIG
    MMT_HIP(hipGetLastError());
}
void trigger_cuts(const uint16_t* masks, uint64_t n, const uint32_t* block_off, void* cuts, bool wide, hipStream_t s) {
    for_trigger_slices(n, [&](uint32_t b0, uint32_t cnt) {
        if (wide)
            hipLaunchKernelGGL((k_trigger_cuts<256, uint64_t>), dim3(cnt), dim3(256), 0, s, masks, (n + 15) / 16, block_off,
                               static_cast<uint64_t*>(cuts), b0);
        else
            hipLaunchKernelGGL((k_trigger_cuts<256, uint32_t>), dim3(cnt), dim3(256), 0, s, masks, (n + 15) / 16, block_off,
                               static_cast<uint32_t*>(cuts), b0);
    });
    MMT_HIP(hipGetLastError());
}

// phrase k occupies V[a_k .. a_k + len_k - 1]; consecutive phrases overlap by w characters
template <typename P>
__global__ void k_phrase_bounds(const P* __restrict__ cuts, uint32_t n_cuts, uint64_t n, uint32_t w,
                                P* __restrict__ start, uint32_t* __restrict__ len) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_cuts) return;
    const uint64_t a = k == 0 ? 0u : (uint64_t)cuts[k - 1] - w + 2;    // V index of T[cut - w + 1]
    const uint64_t b = k < n_cuts ? (uint64_t)cuts[k] + 1 : n + w;      // V index of the last character
    start[k] = (P)a;
    len[k] = (uint32_t)(b - a + 1);
}
void phrase_bounds(const void* cuts, uint32_t n_cuts, uint64_t n, uint32_t w, void* start, uint32_t* len, bool wide,
                   hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_phrase_bounds<uint64_t>, dim3(grid_for((uint64_t)n_cuts + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint64_t*>(cuts), n_cuts, n, w, static_cast<uint64_t*>(start), len);
    else
        hipLaunchKernelGGL(k_phrase_bounds<uint32_t>, dim3(grid_for((uint64_t)n_cuts + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint32_t*>(cuts), n_cuts, n, w, static_cast<uint32_t*>(start), len);
    MMT_HIP(hipGetLastError());
}

// Two independent 64-bit polynomial fingerprints per phrase (the reference keys its std::map by
// one 55-bit KR hash with probing, newscan.hpp:133-142, 269-282; fingerprints only group equal
// phrases here and every merge is verified byte by byte in k_mark_distinct).
#define MMT_B1 0x9E3779B97F4A7C15ull
#define MMT_B2 0xC2B2AE3D27D4EB4Full
template <typename V>
__device__ __forceinline__ uint8_t v_at(const V& v, uint64_t i);
template <> __device__ __forceinline__ uint8_t v_at<const uint8_t*>(const uint8_t* const& v, uint64_t i) { return v[i]; }
template <> __device__ __forceinline__ uint8_t v_at<uint8_t*>(uint8_t* const& v, uint64_t i) { return v[i]; }
template <> __device__ __forceinline__ uint8_t v_at<TextRef>(const TextRef& v, uint64_t i) { return tx_byte(v, i); }
template <typename V>
__device__ __forceinline__ void hash_range(const V& v, uint64_t lo, uint64_t hi, uint64_t& h1,
                                           uint64_t& h2, uint64_t& p1, uint64_t& p2) {
    h1 = 0; h2 = 0; p1 = 1; p2 = 1;
    for (uint64_t i = lo; i < hi; i++) {
        const uint64_t c = (uint64_t)v_at<V>(v, i) + 1;
        h1 = h1 * MMT_B1 + c; h2 = h2 * MMT_B2 + c;
        p1 *= MMT_B1; p2 *= MMT_B2;
    }
}
// The per-phrase record is 16 bytes: 56 bits of the second fingerprint, the 40-bit start (its high byte rides in the
// top byte of the fingerprint word), the length.
constexpr uint32_t FP2_HI_MASK = 0x00ffffffu;
constexpr uint32_t HASH_SPAN = 4096;          // bytes of V one wave of k_phrase_hash stages in LDS
template <typename P>
__global__ void k_phrase_hash(const TextRef T, const P* __restrict__ start,
                              const uint32_t* __restrict__ len, uint32_t m, uint64_t* __restrict__ o1,
                              uint4* __restrict__ pinfo) {
    const uint8_t* const v = T.v;                      // (nullptr: packed text -- every lane reads its phrase through the accessor)
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const bool have = k < m;
    const uint64_t a = have ? (uint64_t)start[k] : 0;
    const uint32_t l = have ? len[k] : 0;
    const bool is_long = l > 2048;
    uint64_t h1 = 0, h2 = 0, p1, p2;
    // The 64 phrases of a wave are consecutive in V and overlap by w characters: when they span at most HASH_SPAN bytes
    // the wave brings the span into LDS with aligned 16-byte loads and every lane reads its phrase from there (one byte
    // load per character and lane from global memory ran at 0.4 TB/s).  The first wave (nothing readable before V) and
    // waves with a long phrase or a partial set of lanes read V directly.
    __shared__ __attribute__((aligned(16))) uint8_t s_span[(256 / 64) * (HASH_SPAN + 32)];
    {
        const uint32_t wave = threadIdx.x >> 6;
        uint8_t* const mine = s_span + wave * (HASH_SPAN + 32);
        const uint64_t a_first = __shfl(a, 0, 64), a_last = __shfl(a, 63, 64), l_last = __shfl(l, 63, 64);
        const bool whole = v != nullptr && __ballot(have) == ~0ull && __ballot(is_long) == 0 && k >= 64;
        const uint64_t lo16 = ((uint64_t)(uintptr_t)(v + a_first)) & ~(uint64_t)15;
        const uint64_t end = (uint64_t)(uintptr_t)(v + a_last + l_last);
        if (whole && end - lo16 <= HASH_SPAN) {
            const uint32_t chunks = (uint32_t)((end - lo16 + 15) / 16);
            for (uint32_t c = lane; c < chunks; c += 64)
                *reinterpret_cast<uint4*>(mine + 16 * c) = *reinterpret_cast<const uint4*>((const uint8_t*)(uintptr_t)lo16 + 16 * (uint64_t)c);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t o = (uint32_t)((uint64_t)(uintptr_t)(v + a) - lo16);
            hash_range<uint8_t*>(mine, o, (uint64_t)o + l, h1, h2, p1, p2);
        } else if (have && !is_long) hash_range<TextRef>(T, a, a + l, h1, h2, p1, p2);
    }
    // long phrases (no trigger inside a low-complexity run): the whole wave hashes one phrase
    uint64_t todo = __ballot(have && is_long);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint64_t A = __shfl(a, src, 64), L = __shfl(l, src, 64);
        const uint64_t chunk = (L + 63) / 64;
        const uint64_t lo = A + (uint64_t)lane * chunk < A + L ? A + (uint64_t)lane * chunk : A + L;
        const uint64_t hi = lo + chunk < A + L ? lo + chunk : A + L;
        uint64_t c1, c2, q1, q2;
        hash_range<TextRef>(T, lo, hi, c1, c2, q1, q2);
        uint64_t t1 = 0, t2 = 0;
        for (int s = 0; s < 64; s++) {                 // left-to-right combine, identical on every lane
            const uint64_t x1 = __shfl(c1, s, 64), x2 = __shfl(c2, s, 64);
            const uint64_t y1 = __shfl(q1, s, 64), y2 = __shfl(q2, s, 64);
            t1 = t1 * y1 + x1; t2 = t2 * y2 + x2;
        }
        if ((int)lane == src) { h1 = t1; h2 = t2; }
    }
    if (have) {
        o1[k] = h1 ^ ((uint64_t)l * 0xD6E8FEB86659FD93ull);
        const uint64_t g2 = h2 + ((uint64_t)l << 32);
        pinfo[k] = make_uint4((uint32_t)g2, ((uint32_t)(g2 >> 32) & FP2_HI_MASK) | ((uint32_t)(a >> 32) << 24), (uint32_t)a, l);
    }
}
void phrase_hash(const TextRef& v, const void* start, const uint32_t* len, uint32_t m, uint64_t* h1, void* pinfo,
                 bool wide, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_phrase_hash<uint64_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, v,
                           static_cast<const uint64_t*>(start), len, m, h1, static_cast<uint4*>(pinfo));
    else
        hipLaunchKernelGGL(k_phrase_hash<uint32_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, v,
                           static_cast<const uint32_t*>(start), len, m, h1, static_cast<uint4*>(pinfo));
    MMT_HIP(hipGetLastError());
}
// h2[k] out of the per-phrase records (only for the rare two-fingerprint ordering)
__global__ void k_second_fingerprint(const uint4* __restrict__ pinfo, uint32_t m, uint64_t* __restrict__ h2) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) { const uint4 p = pinfo[k]; h2[k] = ((uint64_t)(p.y & FP2_HI_MASK) << 32) | p.x; }
}
void second_fingerprint(const void* pinfo, uint32_t m, uint64_t* h2, hipStream_t s) {
    hipLaunchKernelGGL(k_second_fingerprint, dim3(grid_for(m, 256)), dim3(256), 0, s, static_cast<const uint4*>(pinfo), m,
                       h2);
    MMT_HIP(hipGetLastError());
}

__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }

// order[] lists the phrases sorted by their first fingerprint (h1s = those fingerprints in sorted order), or by
// both; flags[k] = 1 where a new distinct phrase starts.  Equal fingerprints + equal length are confirmed by
// comparing the bytes; a mismatch there (a 128-bit collision) raises err[0] instead of silently merging two
// different phrases.  The second fingerprint, start and length of a phrase sit in one 16-byte record.
__global__ void k_mark_distinct(const uint32_t* __restrict__ order, const uint64_t* __restrict__ h1s,
                                const uint4* __restrict__ pinfo, const TextRef T, uint32_t m,
                                uint32_t* __restrict__ flags, uint32_t* __restrict__ err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    // Every lane fetches ITS phrase once (record, then the bytes 8 at a time) and gets the predecessor's from the lane
    // below by shuffle; only lane 0 of a wave reads its predecessor itself.  Equal phrases are long runs in this
    // order (one per haplotype), so this halves the random reads of the verification.
    const bool have = k < m;
    const uint4 X = have ? pinfo[order[k]] : make_uint4(0, 0, 0, 0);
    uint4 Y;
    Y.x = __shfl_up(X.x, 1, 64); Y.y = __shfl_up(X.y, 1, 64); Y.z = __shfl_up(X.z, 1, 64); Y.w = __shfl_up(X.w, 1, 64);
    if (lane == 0 && have && k > 0) Y = pinfo[order[k - 1]];
    const bool same1 = have && k > 0 && h1s[k] == h1s[k - 1];
    const bool same2 = X.x == Y.x && (X.y & FP2_HI_MASK) == (Y.y & FP2_HI_MASK) && X.w == Y.w;
    bool same = same1 && same2;
    // equal first fingerprints of different phrases: when the order came from the first fingerprint alone, equal
    // phrases need not be adjacent any more -- the host then repeats the grouping with both fingerprints
    if (same1 && !same) atomicOr(err + 1, 1u);
    // byte verification; the loop runs while any lane of the wave still compares
    const uint64_t px = ((uint64_t)(X.y >> 24) << 32) | X.z;       // V indices
    const uint64_t py = ((uint64_t)(Y.y >> 24) << 32) | Y.z;
    const uint32_t l = X.w;
    bool verified = same;
    // (16 bytes per step: the two loads of a step are independent and in flight together)
    auto chunk = [&](uint64_t p, uint32_t i) -> uint64_t {
        uint64_t x = 0;
        if (i + 8 <= l) x = tx_load8(T, p + i);
        else for (uint32_t t = i; t < l; t++) x |= (uint64_t)tx_byte(T, p + t) << (8 * (t - i));
        return x;
    };
    for (uint32_t i = 0; __ballot(same && i < l) != 0; i += 16) {
        // every lane with bytes left loads its own chunk (the lane above may need it even if this lane is done comparing)
        uint64_t mine0 = 0, mine1 = 0;
        if (have && i < l) { mine0 = chunk(px, i); if (i + 8 < l) mine1 = chunk(px, i + 8); }
        uint64_t prev0 = __shfl_up(mine0, 1, 64), prev1 = __shfl_up(mine1, 1, 64);
        if (same && i < l) {
            if (lane == 0) { prev0 = chunk(py, i); prev1 = i + 8 < l ? chunk(py, i + 8) : 0; }
            if (mine0 != prev0 || mine1 != prev1) { same = false; verified = false; }
        }
    }
    if (have) {
        if (k == 0) flags[0] = 1;
        else {
            if (same1 && same2 && !verified) atomicAdd(err, 1u);   // collision of both fingerprints
            flags[k] = verified ? 0u : 1u;
        }
    }
}
void mark_distinct(const uint32_t* order, const uint64_t* h1s, const void* pinfo, const TextRef& v, uint32_t m,
                   uint32_t* flags, uint32_t* err, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_distinct, dim3(grid_for(m, 256)), dim3(256), 0, s, order, h1s,
                       static_cast<const uint4*>(pinfo), v, m, flags, err);
    MMT_HIP(hipGetLastError());
}

// scan[k] = inclusive sum of flags: distinct id of order[k] is scan[k] - 1
__global__ void k_assign_distinct(const uint32_t* __restrict__ order, const uint32_t* __restrict__ scan,
                                  const uint32_t* __restrict__ flags, const uint32_t* __restrict__ len, uint32_t m,
                                  uint32_t* __restrict__ pid, uint32_t* __restrict__ rep, uint32_t* __restrict__ dlen) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint32_t d = scan[k] - 1;
    pid[order[k]] = d;
    if (flags[k]) { rep[d] = order[k]; dlen[d] = len[order[k]] + 1; }   // + EndOfWord
}
void assign_distinct(const uint32_t* order, const uint32_t* scan, const uint32_t* flags, const uint32_t* len,
                     uint32_t m, uint32_t* pid, uint32_t* rep, uint32_t* dlen, hipStream_t s) {
    hipLaunchKernelGGL(k_assign_distinct, dim3(grid_for(m, 256)), dim3(256), 0, s, order, scan, flags, len, m, pid, rep,
                       dlen);
    MMT_HIP(hipGetLastError());
}

// dictionary text: phrase bytes, EndOfWord (1) after every phrase, EndOfDict (0) at the very end
// (dictionary file layout of newscan.hpp:386-397).  dinfo[pos] = (distinct phrase id << 32) | suffix
// word: length of the phrase suffix that starts at pos (0 on separators), bit 31 set on the first
// byte of a phrase.  One wave per phrase.
template <typename P>
__global__ void k_copy_dict(const TextRef v, const P* __restrict__ start,
                            const uint32_t* __restrict__ len, const uint32_t* __restrict__ which,
                            const uint32_t* __restrict__ dstart, uint32_t n_phr, uint8_t* __restrict__ dict,
                            uint64_t* __restrict__ dinfo, uint32_t dict_len, int pack_prev) {
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= n_phr) return;
    const uint32_t ph = which[wave], l = len[ph], o = dstart[wave];
    if ((uint64_t)o + l >= (uint64_t)dict_len) return;      // (cannot happen: the caller sized the dictionary in 64 bits)
    const uint64_t a = start[ph];
    // pack_prev (fewer than 2^24 distinct phrases): the byte before each position rides in the top byte of the
    // record, so that k_entry_info needs one random read per dictionary suffix instead of two.  The byte before
    // the first character of a phrase is the terminator of the phrase before it (padding for the first phrase).
    for (uint32_t i = lane; i < l; i += 64) {
        const uint8_t c = tx_byte(v, a + i);
        dict[o + i] = c;
        if (dinfo) {
            uint64_t hi = wave;
            if (pack_prev) hi |= (uint64_t)(i ? tx_byte(v, a + i - 1) : (wave ? (uint8_t)1 : (uint8_t)0)) << 24;
            dinfo[o + i] = (hi << 32) | (uint64_t)((l - i) | (i == 0 ? 0x80000000u : 0u));
        }
    }
    if (lane == 0) {
        dict[o + l] = 1;
        if (dinfo) {
            uint64_t hi = wave;
            if (pack_prev) hi |= (uint64_t)tx_byte(v, a + l - 1) << 24;
            dinfo[o + l] = hi << 32;
        }
        if (wave + 1 == n_phr) {
            dict[dict_len - 1] = 0;
            if (dinfo) { uint64_t hi = wave; if (pack_prev) hi |= (uint64_t)1 << 24; dinfo[dict_len - 1] = hi << 32; }
        }
    }
}
// sum of 32-bit lengths in 64 bits (a 32-bit prefix sum wraps silently)
__global__ void k_sum_u32(const uint32_t* __restrict__ x, uint32_t n, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += x[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
void sum_u32(const uint32_t* x, uint32_t n, uint64_t* d_out, hipStream_t s) {
    MMT_HIP(hipMemsetAsync(d_out, 0, 8, s));
    if (!n) return;
    hipLaunchKernelGGL(k_sum_u32, dim3(std::min<unsigned>(grid_for(n, 256), 4096u)), dim3(256), 0, s, x, n,
                       reinterpret_cast<unsigned long long*>(d_out));
    MMT_HIP(hipGetLastError());
}
void copy_dict(const TextRef& v, const void* start, const uint32_t* len, const uint32_t* which,
               const uint32_t* dstart, uint32_t n_phr, uint8_t* dict, uint64_t* dinfo, uint32_t dict_len,
               bool pack_prev, bool wide, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_copy_dict<uint64_t>, dim3(grid_for((uint64_t)n_phr * 64, 256)), dim3(256), 0, s, v,
                           static_cast<const uint64_t*>(start), len, which, dstart, n_phr, dict, dinfo, dict_len,
                           pack_prev ? 1 : 0);
    else
        hipLaunchKernelGGL(k_copy_dict<uint32_t>, dim3(grid_for((uint64_t)n_phr * 64, 256)), dim3(256), 0, s, v,
                           static_cast<const uint32_t*>(start), len, which, dstart, n_phr, dict, dinfo, dict_len,
                           pack_prev ? 1 : 0);
    MMT_HIP(hipGetLastError());
}

// ---- A3/A4: groups of equal proper phrase suffixes, in dictionary suffix-array order -----------
// One random pass brings everything that is known per dictionary position into suffix-array order;
// all later kernels read these columns coalesced.
__global__ void k_entry_info(const uint32_t* __restrict__ sa_d, const uint64_t* __restrict__ dinfo,
                             const uint8_t* __restrict__ dict, uint32_t nd, int pack_prev,
                             uint32_t* __restrict__ esuf, uint32_t* __restrict__ ephr, uint8_t* __restrict__ ebw) {
    // four entries per thread: the four gathers are in flight together, the columns are stored 16 / 16 / 4 bytes at a time
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r >= nd) return;
    auto one = [&](uint32_t pos, uint64_t e, uint32_t& su, uint32_t& ph) -> uint32_t {
        su = (uint32_t)e;
        uint8_t prev;
        if (pack_prev) { ph = (uint32_t)(e >> 32) & 0xffffffu; prev = (uint8_t)(e >> 56); }
        else { ph = (uint32_t)(e >> 32); prev = pos ? dict[pos - 1] : (uint8_t)0; }
        return prev == 2 ? 0u : (uint32_t)prev;               // Dollar before text position 0 -> bwt 0
    };
    if (r + 4 <= nd) {
        const uint4 p = *reinterpret_cast<const uint4*>(sa_d + r);
        const uint64_t e0 = dinfo[p.x], e1 = dinfo[p.y], e2 = dinfo[p.z], e3 = dinfo[p.w];
        uint4 su, ph;
        const uint32_t b0 = one(p.x, e0, su.x, ph.x), b1 = one(p.y, e1, su.y, ph.y), b2 = one(p.z, e2, su.z, ph.z),
                       b3 = one(p.w, e3, su.w, ph.w);
        *reinterpret_cast<uint4*>(esuf + r) = su;
        *reinterpret_cast<uint4*>(ephr + r) = ph;
        *reinterpret_cast<uint32_t*>(ebw + r) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    } else {
        for (uint64_t t = r; t < nd; t++) {
            const uint32_t pos = sa_d[t];
            uint32_t su, ph;
            ebw[t] = (uint8_t)one(pos, dinfo[pos], su, ph);
            esuf[t] = su; ephr[t] = ph;
        }
    }
}
void entry_info(const uint32_t* sa_d, const uint64_t* dinfo, const uint8_t* dict, uint32_t nd, bool pack_prev,
                uint32_t* esuf, uint32_t* ephr, uint8_t* ebw, hipStream_t s) {
    hipLaunchKernelGGL(k_entry_info, dim3(grid_for(nd, 1024)), dim3(256), 0, s, sa_d, dinfo, dict, nd, pack_prev ? 1 : 0,
                       esuf, ephr, ebw);
    MMT_HIP(hipGetLastError());
}

// LCP array of the dictionary (the reference gets it from gsacak, dictionary.hpp:133; every phrase terminator is a
// symbol of its own, so a match never runs past the end of the shorter phrase suffix) -- by the construction of the
// text-level column (kernels.hip): entries of the dictionary's suffix array whose preceding byte differs from their
// predecessor's are irreducible and compared directly (phrase starts always: the terminator before them is unique), every
// other value follows from PLCP[i] = PLCP[i - 1] - 1 along the dictionary.  A trigger-free run (a gap of a megabase of N, a
// microsatellite: one giant phrase, newscan.hpp:265-325) costs its length once, not once per suffix: comparing neighbours
// directly took 3.8 s of a 6.9 s step on a collection with such runs (group flags + group heads) and takes 0.1 s this way.
constexpr int DICT_IRR_STEPS = 16;
template <int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK) void k_dict_irr(const uint8_t* __restrict__ dict, uint32_t nd,
                                                    const uint32_t* __restrict__ sa_d, const uint32_t* __restrict__ esuf,
                                                    const uint8_t* __restrict__ ebw, uint32_t* __restrict__ plcp,
                                                    k::LongLcpLim* __restrict__ longs, uint32_t* __restrict__ long_count,
                                                    uint32_t long_cap) {
    constexpr int TILE = BLOCK * PER;
    __shared__ uint32_t s_q[TILE];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * TILE;
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t r = base + (uint32_t)q * BLOCK + threadIdx.x;
        bool irr = false;
        if (r > 0 && r < nd) {
            const uint32_t e = esuf[r], pe = esuf[r - 1];
            const uint32_t lim = min(e & 0x7fffffffu, pe & 0x7fffffffu);
            // (an entry that cannot match anything keeps the cleared value 0)
            irr = lim > 0 && ((e >> 31) || (pe >> 31) || ebw[r] != ebw[r - 1]);
        }
        const uint64_t m = __ballot(irr);
        uint32_t at = 0;
        if (lane == 0 && m) at = atomicAdd(&s_n, (uint32_t)__popcll(m));
        at = __shfl(at, 0, 64);
        if (irr) s_q[at + __popcll(m & ((1ull << lane) - 1))] = r - base;
    }
    __syncthreads();
    const uint32_t cnt = s_n;
    for (uint32_t wbase = 0; wbase < cnt; wbase += BLOCK) {
        const uint32_t wi = wbase + threadIdx.x;
        bool queue = false;
        uint32_t p = 0, qq = 0, h = 0, lim = 0;
        if (wi < cnt) {
            const uint32_t r = base + s_q[wi];
            p = sa_d[r]; qq = sa_d[r - 1];
            lim = min(esuf[r] & 0x7fffffffu, esuf[r - 1] & 0x7fffffffu);
            bool done = false;
            for (int step = 0; step < DICT_IRR_STEPS && h < lim; step++) {
                const uint64_t x = ld64(dict + p + h), y = ld64(dict + qq + h);
                if (x != y) { h += (uint32_t)(__builtin_ctzll(x ^ y) >> 3); done = true; break; }
                h += 8;
            }
            if (h >= lim) { h = lim; done = true; }
            if (done) plcp[p] = h;
            else queue = true;
        }
        const uint64_t m = __ballot(queue);
        if (m) {
            uint32_t slot0 = 0;
            const int leader = __builtin_ctzll(m);
            if ((int)lane == leader) slot0 = atomicAdd(long_count, (uint32_t)__popcll(m));
            slot0 = __shfl(slot0, leader, 64);
            if (queue) {
                const uint32_t slot = slot0 + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                if (slot < long_cap) { longs[slot].p = p; longs[slot].q = qq; longs[slot].h = h; longs[slot].lim = lim; }
            }
        }
    }
}
void dict_irreducible(const uint8_t* dict, uint32_t nd, const uint32_t* sa_d, const uint32_t* esuf, const uint8_t* ebw,
                      uint32_t* plcp, void* longs, uint32_t* long_count, uint32_t long_cap, hipStream_t s) {
    constexpr int B = 256, PER = 8;
    MMT_HIP(hipMemsetAsync(plcp, 0, (size_t)nd * 4, s));
    MMT_HIP(hipMemsetAsync(long_count, 0, 4, s));
    hipLaunchKernelGGL((k_dict_irr<B, PER>), dim3(grid_for(nd, B * PER)), dim3(B), 0, s, dict, nd, sa_d, esuf, ebw, plcp,
                       static_cast<k::LongLcpLim*>(longs), long_count, long_cap);
    MMT_HIP(hipGetLastError());
}
// the dictionary's PLCP values can exceed the characters a suffix has left only through the chain rule's slack: clamp
// (lcp[r] = min(PLCP[sa_d[r]], own length, predecessor's length))
__global__ void k_dict_lcp_clamp(uint32_t* __restrict__ lcp, const uint32_t* __restrict__ esuf, uint32_t nd) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    uint32_t v = r ? lcp[r] : 0u;
    const uint32_t a = esuf[r] & 0x7fffffffu, b = r ? esuf[r - 1] & 0x7fffffffu : 0u;
    v = min(v, min(a, b));
    lcp[r] = v;
}
void dict_lcp_clamp(uint32_t* lcp, const uint32_t* esuf, uint32_t nd, hipStream_t s) {
    hipLaunchKernelGGL(k_dict_lcp_clamp, dim3(grid_for(nd, 256)), dim3(256), 0, s, lcp, esuf, nd);
    MMT_HIP(hipGetLastError());
}

// Valid = proper suffix (not the whole phrase) of length >= w (pfp_lcp_mum.hpp:272-282).  Two
// neighbours of the dictionary SA spell the same string iff they have the same length and share all of it
// (:141-154 collects them as `same_suffix`, from lcpD like here).
// vflag = valid, gflag = first of its group, pflag = first byte of a phrase (their order gives the
// lexicographic phrase ranks).  seg[r] = (1 << 32 when the entry before r is valid) | lcp[r]: the segmented minimum of
// these is, at every valid entry, the LCP with the valid entry before it.
__global__ void k_group_flags(const uint32_t* __restrict__ esuf, const uint32_t* __restrict__ lcp_d, uint32_t nd, uint32_t w,
                              uint32_t* __restrict__ gflag, uint32_t* __restrict__ pflag,
                              uint32_t* __restrict__ vflag, uint64_t* __restrict__ seg) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    const uint32_t e = esuf[r];
    const uint32_t sl = e & 0x7fffffffu;
    const bool is_start = e >> 31;
    const bool valid = !is_start && sl >= w;
    bool fresh = valid, pvalid = false;
    const uint32_t l = lcp_d[r];
    if (r > 0) {
        const uint32_t pe = esuf[r - 1];
        pvalid = !(pe >> 31) && (pe & 0x7fffffffu) >= w;
        if (valid && pvalid && (pe & 0x7fffffffu) == sl && l >= sl) fresh = false;
    }
    gflag[r] = fresh ? 1u : 0u;
    pflag[r] = is_start ? 1u : 0u;
    vflag[r] = valid ? 1u : 0u;
    seg[r] = ((uint64_t)((pvalid || r == 0) ? 1u : 0u) << 32) | (uint64_t)l;
}
void group_flags(const uint32_t* esuf, const uint32_t* lcp_d, uint32_t nd, uint32_t w, uint32_t* gflag, uint32_t* pflag,
                 uint32_t* vflag, uint64_t* seg, hipStream_t s) {
    hipLaunchKernelGGL(k_group_flags, dim3(grid_for(nd, 256)), dim3(256), 0, s, esuf, lcp_d, nd, w, gflag, pflag, vflag, seg);
    MMT_HIP(hipGetLastError());
}

// prank[distinct phrase] = 1-based lexicographic rank = position of its first byte among the phrase
// starts of the dictionary suffix array
__global__ void k_phrase_ranks(const uint32_t* __restrict__ esuf, const uint32_t* __restrict__ ephr,
                               const uint32_t* __restrict__ pscan, uint32_t nd, uint32_t* __restrict__ prank) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    if (esuf[r] >> 31) prank[ephr[r]] = pscan[r];
}
void phrase_ranks(const uint32_t* esuf, const uint32_t* ephr, const uint32_t* pscan, uint32_t nd, uint32_t* prank,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_phrase_ranks, dim3(grid_for(nd, 256)), dim3(256), 0, s, esuf, ephr, pscan, nd, prank);
    MMT_HIP(hipGetLastError());
}

// parse[q] = 1-based rank of the q-th phrase (the reference's .parse, newscan.hpp:399-404);
// byrank[rank-1] = distinct id (for writing the sorted dictionary)
__global__ void k_parse_ranks(const uint32_t* __restrict__ pid, const uint32_t* __restrict__ prank, uint32_t m,
                              uint32_t* __restrict__ parse) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) parse[q] = prank[pid[q]];
}
void parse_ranks(const uint32_t* pid, const uint32_t* prank, uint32_t m, uint32_t* parse, hipStream_t s) {
    hipLaunchKernelGGL(k_parse_ranks, dim3(grid_for(m, 256)), dim3(256), 0, s, pid, prank, m, parse);
    MMT_HIP(hipGetLastError());
}
__global__ void k_invert_ranks(const uint32_t* __restrict__ prank, const uint32_t* __restrict__ rep,
                               const uint32_t* __restrict__ dlen, uint32_t n_distinct, uint32_t* __restrict__ which,
                               uint32_t* __restrict__ slen) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_distinct) return;
    which[prank[d] - 1] = rep[d];
    slen[prank[d] - 1] = dlen[d];
}
void invert_ranks(const uint32_t* prank, const uint32_t* rep, const uint32_t* dlen, uint32_t n_distinct,
                  uint32_t* which, uint32_t* slen, hipStream_t s) {
    hipLaunchKernelGGL(k_invert_ranks, dim3(grid_for(n_distinct, 256)), dim3(256), 0, s, prank, rep, dlen, n_distinct,
                       which, slen);
    MMT_HIP(hipGetLastError());
}

// keys for the parse suffix sort: `chars` consecutive ranks of `bits` bits each (0 past the end)
__global__ void k_pack_keys_u32(const uint32_t* __restrict__ parse, uint32_t m, int bits, int chars,
                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    uint64_t key = 0;
    for (int c = 0; c < chars; c++) {
        const uint64_t x = (uint64_t)q + c < m ? parse[q + c] : 0;
        key = (key << bits) | x;
    }
    keys[q] = key;
    vals[q] = q;
}
void pack_keys_u32(const uint32_t* parse, uint32_t m, int bits, int chars, uint64_t* keys, uint32_t* vals,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_pack_keys_u32, dim3(grid_for(m, 256)), dim3(256), 0, s, parse, m, bits, chars, keys, vals);
    MMT_HIP(hipGetLastError());
}

// ---- A4 without a global sort -----------------------------------------------------------
// The occurrences of every distinct phrase, ordered by the rank of the parse suffix that follows
// them, are the reference's inverted list (parse.hpp:106-134 compute_ilist).  Walking the
// dictionary suffix array and copying, for every valid phrase suffix, its phrase's list yields the
// text suffixes grouped by phrase suffix; only suffixes that spell the same string in several
// phrases still have to be merged -- a segmented sort of small segments instead of the
// reference's priority queue (pfp_lcp_mum.hpp:151-212).
// Inverted lists (parse.hpp:106-134): for every distinct phrase, its occurrences ordered by the rank of the parse
// suffix that follows.  Walking the parse suffix array in order visits the followers by rank, so the sequence
//   t = 0: the last phrase of the parse (nothing follows it: smallest key), t = r + 1: the phrase before suffix sa_p[r]
// only needs ONE stable sort by phrase id (no second key, no atomic counting); t itself is the key the emitter
// compares.  The suffix that starts the parse has no phrase before it: its slot carries the dummy id D.
__global__ void k_occ_sequence(const uint32_t* __restrict__ sa_p, const uint32_t* __restrict__ pid, uint32_t m,
                               uint32_t D, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > m) return;
    uint32_t id;
    if (t == 0) id = pid[m - 1];
    else { const uint32_t q1 = sa_p[t - 1]; id = q1 ? pid[q1 - 1] : D; }
    keys[t] = id;
    vals[t] = t;
}
void occ_sequence(const uint32_t* sa_p, const uint32_t* pid, uint32_t m, uint32_t D, uint32_t* keys, uint32_t* vals,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_occ_sequence, dim3(grid_for((uint64_t)m + 1, 256)), dim3(256), 0, s, sa_p, pid, m, D, keys, vals);
    MMT_HIP(hipGetLastError());
}
// after the sort: k-th occurrence overall = (phrase ids[k], t = ts[k]); occ_start[d] = first k of phrase d
// (occ_start[D] = m: the dummy sorts last); occ[k] = (t << pos_bits) | start of that phrase occurrence in V -- one
// 8-byte record, because the emitter reads the short list of a phrase at a random place and pays per 64-byte line
template <typename P>
__global__ void k_occ_finish(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ ts,
                             const uint32_t* __restrict__ sa_p, const P* __restrict__ pstart, uint32_t m,
                             uint32_t* __restrict__ occ_start, uint64_t* __restrict__ occ, uint32_t pos_bits,
                             const uint32_t* __restrict__ sl, uint32_t* __restrict__ occ_sl) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > m) return;
    const uint32_t id = ids[k];
    if (k == 0 || id != ids[k - 1]) occ_start[id] = k;
    if (k == m) return;                                    // the dummy
    const uint32_t t = ts[k];
    const uint32_t q = t ? sa_p[t - 1] - 1 : m - 1;
    occ[k] = ((uint64_t)t << pos_bits) | (uint64_t)pstart[q];
    // LCP of the following parse suffix with its predecessor in the parse's suffix array (parse_lcp.hpp): the emitter
    // needs it for the element's LCP and would otherwise fetch it after it has read this record
    occ_sl[k] = t ? sl[t - 1] : 0u;
}
// The same lists for texts whose positions leave no room for the parse rank in one 64-bit word (37 + 31 bits on a rank's
// share of whole genomes): ONE 12-byte record per occurrence -- t | low 32 bits of the position | its high byte + 24 bits of
// sl[t - 1], saturated (0xffffff: the emitter reads the parse's LCP array instead) -- so that a phrase's short list still
// costs its own lines only.
template <typename P>
__global__ void k_occ_finish12(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ ts,
                               const uint32_t* __restrict__ sa_p, const P* __restrict__ pstart, uint32_t m,
                               uint32_t* __restrict__ occ_start, uint32_t* __restrict__ occ12, const uint32_t* __restrict__ sl) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > m) return;
    const uint32_t id = ids[k];
    if (k == 0 || id != ids[k - 1]) occ_start[id] = k;
    if (k == m) return;                                    // the dummy
    const uint32_t t = ts[k];
    const uint32_t q = t ? sa_p[t - 1] - 1 : m - 1;
    const uint64_t pos = (uint64_t)pstart[q];
    const uint32_t v = t ? sl[t - 1] : 0u;
    occ12[3ull * k] = t;
    occ12[3ull * k + 1] = (uint32_t)pos;
    occ12[3ull * k + 2] = ((uint32_t)(pos >> 32) & 0xffu) | ((v < 0xffffffu ? v : 0xffffffu) << 8);
}
void occ_finish12(const uint32_t* ids, const uint32_t* ts, const uint32_t* sa_p, const void* pstart, bool wide, uint32_t m,
                  uint32_t* occ_start, uint32_t* occ12, const uint32_t* sl, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_occ_finish12<uint64_t>, dim3(grid_for((uint64_t)m + 1, 256)), dim3(256), 0, s, ids, ts, sa_p,
                           static_cast<const uint64_t*>(pstart), m, occ_start, occ12, sl);
    else
        hipLaunchKernelGGL(k_occ_finish12<uint32_t>, dim3(grid_for((uint64_t)m + 1, 256)), dim3(256), 0, s, ids, ts, sa_p,
                           static_cast<const uint32_t*>(pstart), m, occ_start, occ12, sl);
    MMT_HIP(hipGetLastError());
}
void occ_finish(const uint32_t* ids, const uint32_t* ts, const uint32_t* sa_p, const void* pstart, uint32_t m,
                uint32_t* occ_start, uint64_t* occ, uint32_t pos_bits, const uint32_t* sl, uint32_t* occ_sl, bool wide,
                hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_occ_finish<uint64_t>, dim3(grid_for((uint64_t)m + 1, 256)), dim3(256), 0, s, ids, ts, sa_p,
                           static_cast<const uint64_t*>(pstart), m, occ_start, occ, pos_bits, sl, occ_sl);
    else
        hipLaunchKernelGGL(k_occ_finish<uint32_t>, dim3(grid_for((uint64_t)m + 1, 256)), dim3(256), 0, s, ids, ts, sa_p,
                           static_cast<const uint32_t*>(pstart), m, occ_start, occ, pos_bits, sl, occ_sl);
    MMT_HIP(hipGetLastError());
}


// per distinct phrase: (occurrences, first slot in the inverted lists, length of the phrase) in one 16-byte record,
// so that an entry needs one random read instead of three
__global__ void k_phrase_table(const uint32_t* __restrict__ occ_start /* n_distinct + 1 */,
                               const uint32_t* __restrict__ plen, const uint32_t* __restrict__ rep, uint32_t n_distinct,
                               uint4* __restrict__ tab) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_distinct) tab[d] = make_uint4(occ_start[d + 1] - occ_start[d], occ_start[d], plen[rep[d]], 0u);
}
void phrase_table(const uint32_t* occ_start, const uint32_t* plen, const uint32_t* rep, uint32_t n_distinct, void* tab,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_phrase_table, dim3(grid_for(n_distinct, 256)), dim3(256), 0, s, occ_start, plen, rep,
                       n_distinct, static_cast<uint4*>(tab));
    MMT_HIP(hipGetLastError());
}

__global__ void k_entry_compact(const uint32_t* __restrict__ esuf, const uint32_t* __restrict__ ephr,
                                const uint8_t* __restrict__ ebw, const uint32_t* __restrict__ gflag,
                                const uint32_t* __restrict__ gscan, const uint32_t* __restrict__ vflag,
                                const uint32_t* __restrict__ vscan, const uint64_t* __restrict__ segmin,
                                const uint4* __restrict__ tab, uint32_t nd,
                                uint32_t* __restrict__ ce_cnt, uint32_t* __restrict__ ce_first,
                                uint32_t* __restrict__ ce_offm1, uint8_t* __restrict__ ce_bwt,
                                uint32_t* __restrict__ ce_gs, uint32_t* __restrict__ ce_hl,
                                uint32_t* __restrict__ ce_slen) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd || !vflag[r]) return;
    const uint32_t c = vscan[r];
    const uint4 t = tab[ephr[r]];
    const uint32_t sl = esuf[r] & 0x7fffffffu;
    ce_cnt[c] = t.x;
    ce_first[c] = t.y;
    ce_offm1[c] = t.z - sl - 1;                                // offset inside the phrase, minus one
    ce_bwt[c] = ebw[r];
    ce_gs[c] = gflag[r] ? gscan[r] : 0u;                       // first entry of group g: g + 1, every other entry: 0
    ce_hl[c] = (uint32_t)segmin[r];                             // LCP with the valid entry before (segmented minimum of lcpD)
    ce_slen[c] = sl;
}
void entry_compact(const uint32_t* esuf, const uint32_t* ephr, const uint8_t* ebw, const uint32_t* gflag,
                   const uint32_t* gscan, const uint32_t* vflag, const uint32_t* vscan, const uint64_t* segmin,
                   const void* tab, uint32_t nd, uint32_t* ce_cnt, uint32_t* ce_first, uint32_t* ce_offm1,
                   uint8_t* ce_bwt, uint32_t* ce_gs, uint32_t* ce_hl, uint32_t* ce_slen, hipStream_t s) {
    hipLaunchKernelGGL(k_entry_compact, dim3(grid_for(nd, 256)), dim3(256), 0, s, esuf, ephr, ebw, gflag, gscan, vflag,
                       vscan, segmin, static_cast<const uint4*>(tab), nd, ce_cnt, ce_first, ce_offm1, ce_bwt, ce_gs, ce_hl,
                       ce_slen);
    MMT_HIP(hipGetLastError());
}

// Per group of equal phrase suffixes: the length of alpha, and the LCP of alpha with the alpha of the group before -- the
// LCP of the first stream entry of the group (pfp_lcp_mum.hpp:176-186: between groups the reference takes the minimum of
// lcpD; so does this, through the segmented minimum that entry_compact copied to the group's first entry).
__global__ void k_group_heads(const uint32_t* __restrict__ sege, const uint32_t* __restrict__ ce_hl,
                              const uint32_t* __restrict__ ce_slen, uint32_t n_groups, uint2* __restrict__ ghead) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t e = sege[g];
    const uint32_t la = ce_slen[e];
    uint32_t h = g ? ce_hl[e] : 0u;
    if (g) { const uint32_t lb = ce_slen[e - 1]; const uint32_t lim = la < lb ? la : lb; h = h < lim ? h : lim; }
    ghead[g] = make_uint2(la, h);
}
void group_heads(const uint32_t* sege, const uint32_t* ce_hl, const uint32_t* ce_slen, uint32_t n_groups, void* ghead,
                 hipStream_t s) {
    hipLaunchKernelGGL(k_group_heads, dim3(grid_for(n_groups, 256)), dim3(256), 0, s, sege, ce_hl, ce_slen, n_groups,
                       static_cast<uint2*>(ghead));
    MMT_HIP(hipGetLastError());
}

// first index i in [0, n] with a[i] >= x (a non-decreasing), all lanes of the calling wave cooperate
template <typename T>
__device__ __forceinline__ uint32_t wave_lower_bound(const T* __restrict__ a, uint32_t n, T x) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t lo = 0, hi = n;                                 // answer in [lo, hi]
    while (hi - lo > 64) {
        const uint32_t step = (hi - lo + 63) / 64;
        const uint32_t probe = lo + (lane + 1) * step;       // probes lo+step .. lo+64*step
        const bool ge = probe >= hi ? true : a[probe] >= x;
        const uint64_t m = __ballot(ge);
        const uint32_t first = (uint32_t)__builtin_ctzll(m); // always at least the last lane
        const uint32_t nhi = lo + (first + 1) * step;
        lo = lo + first * step;
        hi = nhi < hi ? nhi : hi;
    }
    const uint32_t i = lo + lane;
    const bool ge = i >= hi ? true : a[i] >= x;
    const uint64_t m = __ballot(ge);
    return m ? lo + (uint32_t)__builtin_ctzll(m) : hi;
}

// One workgroup per tile of TILE output positions: takes the groups (of equal phrase suffixes) that
// START inside its tile, expands their entries' inverted lists into LDS and places every element at
// (group begin + number of group elements with a smaller following-parse-suffix rank) -- the k-way
// merge of pfp_lcp_mum.hpp:151-212 done by counting, since the lists of one group are sorted runs.
// Groups that do not fit CAP elements are expanded unsorted and queued for a segmented sort.
// P = type of the stream offsets and text positions (uint32_t narrow, uint64_t wide), SA = suffix-array accessor.
template <typename P, typename SA>
struct EmitArgsT {
    const P* segb; const uint32_t* sege; uint32_t n_groups;
    const P* ce_eoff; const uint32_t* ce_cnt; const uint32_t* ce_first; const uint32_t* ce_offm1;
    const uint8_t* ce_bwt; const uint32_t* ce_gs;
    const uint64_t* occ; const uint32_t* occ_sl; uint32_t pos_bits;
    const uint32_t* occ12;       // 12-byte occurrence records instead of occ / occ_sl (k_occ_finish12), or nullptr
    P n;
    SA sa; uint8_t* bwt;
    const uint32_t* fb_group; const P* fb_off; uint32_t n_fb; P fb_base;
    uint32_t* fb_keys; P* fb_vals;
    const uint8_t* bwt_code; uint32_t fb_bits;
    uint32_t* err;
    uint64_t tile_lo;
    // LCP column and the window of the stream this launch writes: entry j of the suffix array (stream entry j + 1) goes to
    // index j - out_base of sa / bwt / lcp when win_lo <= j < win_hi, and nowhere otherwise
    uint32_t* lcp; const uint2* ghead; RmqView rmq; uint32_t w;
    uint64_t out_base, win_lo, win_hi;
    uint32_t many_runs;          // a group of more runs than this: the piece is ranked by one sort in LDS (emit_piece)
    uint32_t abl2;               // k_emit2, timing only, WRONG OUTPUT (MMT_EMIT2_ABLATE): 1 no range minima, 2 no occurrence records, 4 no column stores, 8 no group heads
};
template <int BLOCK, int CAP, typename ES = uint32_t>
struct EmitShared {
    alignas(8) uint32_t efirst[CAP];   // these two also hold, once the elements have their positions, (key, sl[key - 1]) of
    uint32_t eoffm1[CAP];              // the element in every slot of the merged order, as CAP pairs
    uint32_t key[CAP];
    ES estart[CAP + 1];          // (16 bits where the entries of a piece begin inside it: k_emit; a chunk of an oversized group may begin before its piece)
    uint16_t egfirst[CAP];       // first entry of the entry's group (an index below CAP)
    uint16_t egs[CAP];           // group id + 1 - (first group of the piece) at the first entry of a group, 0 elsewhere
    uint16_t owner[CAP];
    uint8_t ebwt[CAP];
    uint32_t wmax[BLOCK / 64], gwmax[BLOCK / 64];
    uint32_t bound[4];
    uint32_t many;               // most runs (entries) in one group of the piece
    uint64_t origin;             // oversized group: output offset that maps to slot 0 of this launch's fallback arrays
};

// Expands entries [e0, e1) (<= CAP entries, L <= CAP elements starting at output offset clo) through
// LDS.  sorted = true: entries form whole groups; every element goes to its merged position in
// sa / bwt.  sorted = false: part of an oversized group; (key, position) go to the fallback arrays at
// (output offset - fb_origin).
template <int BLOCK, int CAP, typename P, typename SA, typename ES, int ABL = 0>
__device__ __forceinline__ void emit_piece(const EmitArgsT<P, SA>& a, EmitShared<BLOCK, CAP, ES>& sh, uint32_t e0, uint32_t e1,
                                           P clo, uint32_t L, bool sorted, P fb_origin, uint32_t gbase) {
    using EmitSh = EmitShared<BLOCK, CAP, ES>;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t E = e1 - e0;
    const uint64_t pos_mask = (1ull << a.pos_bits) - 1ull;
    __syncthreads();
    for (uint32_t i = tid; i < L; i += BLOCK) sh.owner[i] = 0;
    __syncthreads();
    for (uint32_t e = tid; e < E; e += BLOCK) {
        const uint32_t st = (uint32_t)(a.ce_eoff[e0 + e] - clo);
        sh.estart[e] = (ES)st;
        sh.efirst[e] = a.ce_first[e0 + e];
        sh.eoffm1[e] = a.ce_offm1[e0 + e];
        sh.ebwt[e] = a.ce_bwt[e0 + e];
        const uint32_t gs = a.ce_gs[e0 + e];               // group id + 1 at the first entry of a group
        sh.egs[e] = (uint16_t)(gs ? (sorted ? gs - gbase : 1u) : 0u);
        sh.owner[st < L ? st : 0u] = (uint16_t)e;            // (k_emit_big: entry 0 may begin before the piece -- st wraps, k = i - st holds)
    }
    if (tid == 0) { sh.estart[E] = (ES)L; sh.many = 0; }
    __syncthreads();
    if (ABL == 5) return;
    // owner[i] = last entry starting at or before i; egfirst[e] = last group-start entry at or before e
    // (running maxima over <= CAP items: each thread scans a contiguous slice, slices are stitched)
    {
        constexpr int PER = CAP / BLOCK;
        const uint32_t b0 = tid * PER;
        uint32_t run = 0, grun = 0;
        uint32_t loc[PER], gloc[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = b0 + q;
            if (i < L) { const uint32_t o = sh.owner[i]; run = o > run ? o : run; }
            loc[q] = run;
            if (i < E) { if (sh.egs[i]) grun = i; }
            gloc[q] = grun;
        }
        uint32_t inc = run, ginc = grun;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(inc, o, 64), gy = __shfl_up(ginc, o, 64);
            if (lane >= (uint32_t)o) { inc = y > inc ? y : inc; ginc = gy > ginc ? gy : ginc; }
        }
        if (lane == 63) { sh.wmax[wave] = inc; sh.gwmax[wave] = ginc; }
        __syncthreads();
        uint32_t pre = __shfl_up(inc, 1, 64), gpre = __shfl_up(ginc, 1, 64);
        if (lane == 0) { pre = 0; gpre = 0; }
        for (uint32_t wv = 0; wv < wave; wv++) {
            pre = sh.wmax[wv] > pre ? sh.wmax[wv] : pre;
            gpre = sh.gwmax[wv] > gpre ? sh.gwmax[wv] : gpre;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = b0 + q;
            if (i < L) sh.owner[i] = (uint16_t)(loc[q] > pre ? loc[q] : pre);
            if (i < E) {
                const uint32_t gf = gloc[q] > gpre ? gloc[q] : gpre;
                sh.egfirst[i] = (uint16_t)gf;
                if (sorted && (i + 1 == E || sh.egs[i + 1]) && i - gf >= a.many_runs) atomicMax(&sh.many, i - gf + 1);
            }
        }
    }
    __syncthreads();
    constexpr int PERX = CAP / BLOCK;
    P my_pos[PERX];
    uint32_t my_sl[PERX];
#pragma unroll
    for (int q = 0; q < PERX; q++) {
        const uint32_t i = tid + q * BLOCK;
        my_sl[q] = 0;
        if (i < L) {
            const uint32_t e = sh.owner[i], k = i - sh.estart[e];
            uint32_t key;
            if (a.occ12 && ABL != 3) {
                const uint32_t* r = a.occ12 + 3ull * ((uint64_t)sh.efirst[e] + k);
                const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
                key = r0;
                my_pos[q] = (P)(((uint64_t)r1 | ((uint64_t)(r2 & 0xffu) << 32)) + sh.eoffm1[e]);
                if (sorted) { const uint32_t v = r2 >> 8; my_sl[q] = v == 0xffffffu && r0 ? a.rmq.sl[r0 - 1] : v; }
            } else {
                const uint64_t kp = ABL == 3 ? ((uint64_t)(i + 1) << a.pos_bits) | 7u : a.occ[sh.efirst[e] + k];
                if (sorted && ABL != 3) my_sl[q] = a.occ_sl[sh.efirst[e] + k];
                key = (uint32_t)(kp >> a.pos_bits);
                my_pos[q] = (P)((kp & pos_mask) + sh.eoffm1[e]);
            }
            if (sorted) sh.key[i] = key;
            else {
                const uint32_t slot = (uint32_t)(clo - fb_origin) + i;
                a.fb_keys[slot] = a.fb_bits ? (key << a.fb_bits) | a.bwt_code[sh.ebwt[e]] : key;
                a.fb_vals[slot] = my_pos[q];
            }
        }
    }
    if (!sorted) return;
    __syncthreads();
    // What the LCP values need from global memory is requested before the merge ranks are counted in LDS: per element
    // sl[key - 1] (it came with the occurrence record) -- the last term of min(sl[t1 .. t2 - 1]), and all of it when the
    // two parse ranks are adjacent, the common case --, per slot |alpha| and the LCP at the head of its group.  (Element i
    // and slot i lie in the same group: a group keeps its slots.)
    uint2 g_head[PERX];
    uint32_t my_e[PERX], my_gf[PERX], my_gs[PERX];
#pragma unroll
    for (int q = 0; q < PERX; q++) {
        const uint32_t i = tid + q * BLOCK;
        g_head[q] = make_uint2(0u, 0u); my_e[q] = 0; my_gf[q] = 0; my_gs[q] = 0;
        if (i < L) {
            my_e[q] = sh.owner[i];
            my_gf[q] = sh.egfirst[my_e[q]];
            my_gs[q] = sh.estart[my_gf[q]];
            if (ABL != 4) g_head[q] = a.ghead[gbase + sh.egs[my_gf[q]] - 1];
        }
    }
    // merge rank of every element inside its group = its slot
    uint32_t my_slot[PERX];
    if (ABL == 2) {
#pragma unroll
        for (int q = 0; q < PERX; q++) my_slot[q] = tid + q * BLOCK;
    } else if (sh.many) {
        // A group of many runs -- a phrase suffix that ends hundreds of distinct phrases: the copies of a satellite monomer,
        // each with its own mutations -- would cost every element one binary search per run.  The piece is sorted instead,
        // once, by (first slot of the group, rank of the following parse suffix): groups keep their slots and the ranks
        // are distinct, so the place in the sorted order IS the slot.  (The entry tables in efirst / eoffm1 are dead.)
        uint64_t* const sk = reinterpret_cast<uint64_t*>(sh.efirst);
        uint32_t P2 = 64;
        while (P2 < L) P2 <<= 1;
#pragma unroll
        for (int q = 0; q < PERX; q++) {
            const uint32_t i = tid + q * BLOCK;
            if (i < P2) sk[i] = i < L ? ((uint64_t)my_gs[q] << 42) | ((uint64_t)sh.key[i] << 10) | (uint64_t)i : ~0ull;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < P2 / 2; t += BLOCK) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const bool up = (lo & k) == 0;
                    const uint64_t x = sk[lo], y = sk[hi];
                    if ((x > y) == up) { sk[lo] = y; sk[hi] = x; }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int q = 0; q < PERX; q++) {
            const uint32_t pl = tid + q * BLOCK;
            if (pl < L) sh.owner[(uint32_t)(sk[pl] & 1023u)] = (uint16_t)pl;       // (my_e holds what owner held)
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PERX; q++) {
            const uint32_t i = tid + q * BLOCK;
            my_slot[q] = i < L ? (uint32_t)sh.owner[i] : 0u;
        }
    } else {
#pragma unroll
    for (int q = 0; q < PERX; q++) {
        const uint32_t i = tid + q * BLOCK;
        my_slot[q] = 0;
        if (i < L) {
            const uint32_t e = my_e[q], key = sh.key[i];
            uint32_t rank = 0, e2 = my_gf[q];
            do {
                const uint32_t lo = sh.estart[e2], hi = sh.estart[e2 + 1];
                if (e2 == e) rank += i - lo;
                else {                                   // #keys of run e2 smaller than key
                    uint32_t x = lo, y = hi;
                    while (x < y) { const uint32_t mid = (x + y) >> 1; if (sh.key[mid] < key) x = mid + 1; else y = mid; }
                    rank += x - lo;
                }
                e2++;
            } while (e2 < E && !sh.egs[e2]);
            my_slot[q] = my_gs[q] + rank;
        }
    }
    }
    // The elements move to their slots INSIDE LDS -- (key, sl) over efirst / eoffm1, the text position over key[], its
    // high byte and the BWT byte over owner[]: nothing of the tile tables is read any more -- so that every column is
    // written to HBM in slot order, one coalesced store per column (the elements used to be scattered from where they
    // were expanded: four partial-line stores per element).
    static_assert(offsetof(EmitSh, eoffm1) == offsetof(EmitSh, efirst) + CAP * sizeof(uint32_t), "efirst and eoffm1 must be adjacent");
    uint2* const merged = reinterpret_cast<uint2*>(sh.efirst);
    uint32_t* const mpos = sh.key;
    uint16_t* const mhb = sh.owner;
    uint8_t my_bwt[PERX];
#pragma unroll
    for (int q = 0; q < PERX; q++) my_bwt[q] = tid + q * BLOCK < L ? sh.ebwt[my_e[q]] : (uint8_t)0;
    uint32_t my_key[PERX];
#pragma unroll
    for (int q = 0; q < PERX; q++) my_key[q] = tid + q * BLOCK < L ? sh.key[tid + q * BLOCK] : 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PERX; q++) {
        const uint32_t i = tid + q * BLOCK;
        if (i < L) {
            const uint32_t slot = my_slot[q];
            const uint64_t pos = (uint64_t)my_pos[q];
            merged[slot] = make_uint2(my_key[q], my_sl[q]);
            mpos[slot] = (uint32_t)pos;
            mhb[slot] = (uint16_t)(((uint32_t)(pos >> 32) & 0xffu) | ((uint32_t)my_bwt[q] << 8));
        }
    }
    __syncthreads();
    // slot i: suffix-array entry, BWT byte, and the LCP with the slot before it -- inside a group |alpha| - w + the LCP of
    // the two following parse suffixes (a range minimum over the parse's LCP array: pfp_lcp_mum.hpp:295-321), at the first
    // slot of a group the LCP of the two phrase suffixes themselves
#pragma unroll
    for (int q = 0; q < PERX; q++) {
        const uint32_t i = tid + q * BLOCK;
        if (i < L) {
            const uint64_t out = (uint64_t)clo + i;
            const uint32_t hb = mhb[i];
            const uint64_t pos = (uint64_t)mpos[i] | ((uint64_t)(hb & 0xffu) << 32);
            if (out == 0) {                              // entry 0 must be the end sentinel
                if (pos != (uint64_t)a.n) { atomicAdd(a.err, 1u); atomicAdd(a.err + 4, 1u); }
                continue;
            }
            if (pos >= (uint64_t)a.n) {
                atomicAdd(a.err, 1u);
                if (atomicAdd(a.err + 5, 1u) == 0) {       // first offender, for the error message
                    a.err[8] = (uint32_t)pos; a.err[9] = (uint32_t)(pos >> 32);
                    a.err[10] = (uint32_t)out; a.err[11] = (uint32_t)(out >> 32);
                }
                continue;
            }
            const uint64_t j = out - 1;
            if (j < a.win_lo || j >= a.win_hi) continue;
            uint32_t v;
            if (i == my_gs[q]) v = g_head[q].y;
            else {
                const uint2 cur = merged[i];
                const uint32_t t1 = merged[i - 1].x, t2 = cur.x;
                if (t1 == 0 || t2 <= t1) { atomicAdd(a.err, 1u); atomicAdd(a.err + 3, 1u); v = 0; }
                else {
                    uint32_t mn = cur.y;
                    if (t2 - t1 > 1) { const uint32_t rest = rmq_min(a.rmq, t1, t2 - 2); mn = rest < mn ? rest : mn; }
                    const uint64_t x = (uint64_t)g_head[q].x - a.w + mn;
                    v = x < (uint64_t)LCP_CAP ? (uint32_t)x : LCP_CAP;
                }
            }
            const uint64_t at = j - a.out_base;
            if (ABL == 1) { if (v == 0xfffffff3u) a.lcp[at] = v; continue; }
            a.sa.set(at, pos);
            a.bwt[at] = (uint8_t)(hb >> 8);
            a.lcp[at] = j == 0 ? 0u : v;
        }
    }
}

// tile_first[t] = first group whose begin offset is >= t * TILE (t = 0 .. tiles; n_groups past the end): one
// pass over the groups instead of a global binary search per workgroup of the emitter
template <typename P>
__global__ void k_tile_first(const P* __restrict__ segb, uint32_t n_groups, uint32_t tile, uint64_t tiles,
                             uint32_t* __restrict__ tile_first, uint64_t tile_base) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    // group g is the first one at or after t * tile for every t with begin(g-1) < t * tile <= begin(g);
    // the thread of g == n_groups closes the table for the tiles behind the last group
    // (tile_base: the tables of one batch of the stream -- guided.cpp, expansion -- begin at that tile: entry t - tile_base)
    const uint64_t lo = g ? (uint64_t)segb[g - 1] / tile + 1 : tile_base;
    const uint64_t hi = g < n_groups ? (uint64_t)segb[g] / tile : tiles;
    for (uint64_t t = lo; t <= hi && t <= tiles; t++) tile_first[t - tile_base] = g;
}
void tile_first(const void* segb, uint32_t n_groups, uint64_t tiles, uint32_t* out, bool wide, hipStream_t s, uint64_t tile_base) {
    if (wide)
        hipLaunchKernelGGL(k_tile_first<uint64_t>, dim3(grid_for((uint64_t)n_groups + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint64_t*>(segb), n_groups, emit_tile(), tiles, out, tile_base);
    else
        hipLaunchKernelGGL(k_tile_first<uint32_t>, dim3(grid_for((uint64_t)n_groups + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint32_t*>(segb), n_groups, emit_tile(), tiles, out, tile_base);
    MMT_HIP(hipGetLastError());
}

// What a workgroup needs to know about a tile before it can load anything of it: the first piece -- the maximal run of whole
// groups [g0, g_next) that begin in the tile and hold at most CAP elements --, its entries [e0, e1), its first output offset
// inside the tile and its length; and where the (rare) rest of the tile goes on: groups [g_next, g_end).  Round 3's k_emit
// found this by itself, per tile: tile_first -> a ballot search over segb -> sege, three dependent round trips in front of
// every tile's first useful load.  k_emit_plan does it for all tiles of a launch at once, one work-item per tile (a binary
// search: throughput, not latency), and k_emit begins with one 32-byte record.
struct EmitDesc { uint32_t g0, g_next, g_end, e0, e1, clo, L, pad; };
template <typename P, int TILE, int CAP>
__global__ void k_emit_plan(const P* __restrict__ segb, const uint32_t* __restrict__ sege, const uint32_t* __restrict__ tile_first,
                            uint64_t tile_base, uint64_t tile_lo, uint32_t n_tiles, EmitDesc* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t tile = tile_lo + t, tbase = tile * TILE;
    EmitDesc d;
    d.g0 = tile_first[tile - tile_base]; d.g_end = tile_first[tile - tile_base + 1]; d.g_next = d.g0;
    d.e0 = d.e1 = d.clo = d.L = d.pad = 0;
    if (d.g0 < d.g_end) {
        const uint64_t first = (uint64_t)segb[d.g0] - tbase, lim = first + CAP;
        // groups after g0 that still begin at or before lim (the begin of group g_end, the first one of the next tile, counts:
        // it is where the last group of this tile ends)
        uint32_t lo = d.g0 + 1, hi = d.g_end + 1;
        while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)segb[mid] - tbase <= lim) lo = mid + 1; else hi = mid; }
        const uint32_t g2 = lo - 1;
        if (g2 > d.g0) {
            d.clo = (uint32_t)first; d.L = (uint32_t)((uint64_t)segb[g2] - tbase - first);
            d.e0 = sege[d.g0]; d.e1 = sege[g2]; d.g_next = g2;
        } else d.g_next = d.g0 + 1;          // a single group larger than CAP: k_emit_big's
    }
    out[t] = d;
}

// Persistent workgroups: workgroup b takes tiles b, b + gridDim.x, ... of the launch; the record of its NEXT tile is
// requested before the current tile is expanded, so that it has arrived when the tile is done.
template <int BLOCK, int CAP, int TILE, typename P, typename SA, int ABL = 0>
__global__ __launch_bounds__(BLOCK, 7) void k_emit(EmitArgsT<P, SA> a, const EmitDesc* __restrict__ desc, uint32_t n_tiles) {
    __shared__ EmitShared<BLOCK, CAP, uint16_t> sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    EmitDesc d = desc[t];
    for (;;) {
        const uint32_t t_next = t + gridDim.x;
        EmitDesc dn = d;
        if (t_next < n_tiles) dn = desc[t_next];
        const uint64_t tile = a.tile_lo + t;
        const P tbase = (P)(tile * TILE);
        if (d.L) emit_piece<BLOCK, CAP, P, SA, uint16_t, ABL>(a, sh, d.e0, d.e1, tbase + d.clo, d.L, true, (P)0, d.g0);
        // what the first piece left of the tile (a last group that hangs over by more than CAP - TILE elements; oversized
        // groups, which k_emit_big expands chunk by chunk over many workgroups into the compact fallback arrays)
        uint32_t g = d.g_next;
        const uint32_t g_end = d.g_end;
        while (g < g_end) {
            __syncthreads();
            if (wave == 0) {
                const uint64_t first = (uint64_t)a.segb[g] - (uint64_t)tbase;
                const uint64_t lim = first + CAP;
                uint32_t c = 0;                                  // groups after g that still begin at or before lim
                for (uint32_t base = g + 1; base <= g_end; base += 64) {
                    const uint32_t idx = base + lane;
                    const bool ok = idx <= g_end && (uint64_t)a.segb[idx] - (uint64_t)tbase <= lim;
                    const uint64_t m = __ballot(ok);
                    c += (uint32_t)__popcll(m);
                    if (m != ~0ull) break;
                }
                if (lane == 0) {
                    sh.bound[2] = g + c;
                    sh.bound[0] = (uint32_t)first;
                    sh.bound[1] = (uint32_t)((uint64_t)a.segb[g + c] - (uint64_t)tbase - first);
                }
            }
            __syncthreads();
            const uint32_t g2 = sh.bound[2];
            if (g2 > g) {
                const uint32_t clo = sh.bound[0], L = sh.bound[1];
                emit_piece<BLOCK, CAP, P, SA, uint16_t>(a, sh, a.sege[g], a.sege[g2], tbase + clo, L, true, (P)0, g);
                g = g2;
                continue;
            }
            g = g + 1;
        }
        if (t_next >= n_tiles) break;
        t = t_next; d = dn;
    }
}

// ---- the emitter's tile kernel, second form (round 5) ----------------------------------------------------------------------
// Same pieces, same tables, same output as emit_piece(sorted = true); what changed is how an element finds its place in
// the piece.  The first form zeroed an owner array, scattered the entries' first elements into it and ran two running
// maxima over the piece (entry of every element, first entry of every entry's group): six workgroup barriers before the
// first occurrence record could be requested, ten per piece.  Here the entries leave two BIT MASKS in LDS -- "an entry
// begins at element i", "entry e is the first of its group" -- and every element finds its entry, the first entry of its
// group and the first entry of the next group with a count-leading-zeros on one or two mask words: four barriers per piece
// (entries in place / keys in place / elements in their slots / piece written), and the group-head records are requested
// together with the occurrence records.  The merged (key, sl) pairs and the high bytes overlay tables that are dead once
// the records are gathered, so no barrier separates the merge ranks from the move.  Runs of one element -- the copies of
// a phrase with a private mutation next to the long run of the unmutated phrase: the common shape of a group in a
// pangenome -- are ranked by one comparison instead of a binary search, and a group of one run keeps its order.
template <int BLOCK, int CAP>
struct EmitTile {
    alignas(8) uint32_t efirst[CAP];   // per entry: first slot of its inverted list | from the move on: key of the element in slot i ...
    uint32_t eoffm1[CAP];              // per entry: offset in the phrase - 1        | ... and its sl, as CAP pairs over both arrays
    uint32_t key[CAP];                 // key of element i, in expansion order
    uint32_t mpos[CAP];                // at the first entry of a group: group id + 1 | from the move on: text position (low word) in slot i
    uint16_t estart[CAP + 2];          // first element of entry e; estart[E] = L
    uint16_t owner_at[CAP];            // at the first element of an entry: the entry | from the move on: position high byte | BWT byte << 8
    uint8_t ebwt[CAP];
    uint32_t omask[CAP / 32];          // bit i: an entry begins at element i
    uint32_t gmask[CAP / 32 + 1];      // bit e: entry e is the first of its group; bit E closes the table
    uint32_t bound[4];
    uint32_t many;                     // some group of the piece has more runs than many_runs
};

template <int BLOCK, int CAP, typename P, typename SA>
__device__ __forceinline__ void emit_tile_piece(const EmitArgsT<P, SA>& a, EmitTile<BLOCK, CAP>& sh, uint32_t e0, uint32_t e1,
                                                P clo, uint32_t L) {
    using Sh = EmitTile<BLOCK, CAP>;
    constexpr int PER = CAP / BLOCK;
    constexpr uint32_t MW = CAP / 32;
    const uint32_t tid = threadIdx.x;
    const uint32_t E = e1 - e0;
    const uint64_t pos_mask = (1ull << a.pos_bits) - 1ull;
    __syncthreads();                                         // the piece before is written out; the masks are zero
    for (uint32_t e = tid; e < E; e += BLOCK) {
        const uint32_t st = (uint32_t)(a.ce_eoff[e0 + e] - clo);
        const uint32_t gs = a.ce_gs[e0 + e];                 // group id + 1 at the first entry of a group
        const uint32_t fi = a.ce_first[e0 + e], om = a.ce_offm1[e0 + e];
        const uint8_t bw = a.ce_bwt[e0 + e];
        sh.estart[e] = (uint16_t)st;
        sh.efirst[e] = fi;
        sh.eoffm1[e] = om;
        sh.ebwt[e] = bw;
        if (st < (uint32_t)CAP) {
            sh.owner_at[st] = (uint16_t)e;
            atomicOr(&sh.omask[st >> 5], 1u << (st & 31));
        }
        if (gs) { atomicOr(&sh.gmask[e >> 5], 1u << (e & 31)); sh.mpos[e] = gs; }
    }
    if (tid == 0) { sh.estart[E] = (uint16_t)L; atomicOr(&sh.gmask[E >> 5], 1u << (E & 31)); sh.many = 0; }
    __syncthreads();
    // every element: its entry, the runs of its group, its occurrence record, the head record of its group
    P my_pos[PER];
    uint32_t my_sl[PER], my_run[PER], my_gs[PER];          // my_run = entry | first entry of the group << 10 | first entry of the next group << 20
    uint2 g_head[PER];
    bool any_many = false;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = tid + q * BLOCK;
        my_sl[q] = 0; my_run[q] = 0; my_gs[q] = 0; my_pos[q] = 0; g_head[q] = make_uint2(0u, 0u);
        if (i < L) {
            uint32_t wd = i >> 5;
            uint32_t m = sh.omask[wd] & (0xffffffffu >> (31u - (i & 31u)));
            while (!m && wd) m = sh.omask[--wd];
            const uint32_t p = m ? (wd << 5) + 31u - (uint32_t)__clz(m) : 0u;
            const uint32_t e = sh.owner_at[p], k = i - p;
            wd = e >> 5;
            m = sh.gmask[wd] & (0xffffffffu >> (31u - (e & 31u)));
            while (!m && wd) m = sh.gmask[--wd];
            const uint32_t gf = m ? (wd << 5) + 31u - (uint32_t)__clz(m) : 0u;
            wd = e >> 5;
            m = sh.gmask[wd] & ~(0xffffffffu >> (31u - (e & 31u)));
            while (!m && wd < MW) m = sh.gmask[++wd];
            const uint32_t ge = m ? (wd << 5) + (uint32_t)__ffs(m) - 1u : E;
            const uint32_t fi = sh.efirst[e], om = sh.eoffm1[e];
            uint32_t key;
            if (a.abl2 & 2u) { key = i + 1; my_pos[q] = (P)7; my_sl[q] = 5; }
            else if (a.occ12) {
                const uint32_t* r = a.occ12 + 3ull * ((uint64_t)fi + k);
                const uint32_t r0 = r[0], r1 = r[1], r2 = r[2];
                key = r0;
                my_pos[q] = (P)(((uint64_t)r1 | ((uint64_t)(r2 & 0xffu) << 32)) + om);
                const uint32_t v = r2 >> 8;
                my_sl[q] = v == 0xffffffu && r0 ? a.rmq.sl[r0 - 1] : v;
            } else {
                const uint64_t kp = a.occ[fi + k];
                my_sl[q] = a.occ_sl[fi + k];
                key = (uint32_t)(kp >> a.pos_bits);
                my_pos[q] = (P)((kp & pos_mask) + om);
            }
            if (!(a.abl2 & 8u)) g_head[q] = a.ghead[sh.mpos[gf] - 1u];
            my_gs[q] = sh.estart[gf];
            my_run[q] = e | (gf << 10) | (ge << 20);
            sh.key[i] = key;
            if (ge - gf - 1u >= a.many_runs) any_many = true;
        }
    }
    if (any_many) sh.many = 1u;
    __syncthreads();                                         // keys in place; masks, owner_at, the entry rows and the group ids are dead
    if (tid < MW) sh.omask[tid] = 0u;
    if (tid <= MW) sh.gmask[tid] = 0u;
    // merge rank of every element inside its group = its slot
    uint32_t my_slot[PER];
    if (sh.many) {
        // (see emit_piece: a group of many runs is ranked by one sort of the piece in LDS)
        uint64_t* const sk = reinterpret_cast<uint64_t*>(sh.efirst);
        uint32_t P2 = 64;
        while (P2 < L) P2 <<= 1;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = tid + q * BLOCK;
            if (i < P2) sk[i] = i < L ? ((uint64_t)my_gs[q] << 42) | ((uint64_t)sh.key[i] << 10) | (uint64_t)i : ~0ull;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= P2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < P2 / 2; t += BLOCK) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const bool up = (lo & k) == 0;
                    const uint64_t x = sk[lo], y = sk[hi];
                    if ((x > y) == up) { sk[lo] = y; sk[hi] = x; }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t pl = tid + q * BLOCK;
            if (pl < L) sh.owner_at[(uint32_t)(sk[pl] & 1023u)] = (uint16_t)pl;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = tid + q * BLOCK;
            my_slot[q] = i < L ? (uint32_t)sh.owner_at[i] : 0u;
        }
        __syncthreads();                                     // (the sort's columns become the merged pairs, owner_at the high bytes)
    } else {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = tid + q * BLOCK;
            my_slot[q] = i;
            if (i < L) {
                const uint32_t e = my_run[q] & 1023u, gf = (my_run[q] >> 10) & 1023u, ge = my_run[q] >> 20;
                if (ge - gf > 1u && !(a.abl2 & 64u)) {
                    const uint32_t key = sh.key[i];
                    uint32_t rank = 0, lo = my_gs[q];
                    for (uint32_t e2 = gf; e2 < ge; e2++) {
                        const uint32_t hi = sh.estart[e2 + 1];
                        if (e2 == e) rank += i - lo;
                        else if (hi - lo == 1u) rank += sh.key[lo] < key ? 1u : 0u;
                        else if (!(a.abl2 & 32u)) {              // #keys of run e2 smaller than key
                            uint32_t x = lo, y = hi;
                            while (x < y) { const uint32_t mid = (x + y) >> 1; if (sh.key[mid] < key) x = mid + 1; else y = mid; }
                            rank += x - lo;
                        }
                        lo = hi;
                    }
                    my_slot[q] = my_gs[q] + rank;
                }
            }
        }
    }
    static_assert(offsetof(Sh, eoffm1) == offsetof(Sh, efirst) + CAP * sizeof(uint32_t), "efirst and eoffm1 must be adjacent");
    uint2* const merged = reinterpret_cast<uint2*>(sh.efirst);
    uint16_t* const mhb = sh.owner_at;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = tid + q * BLOCK;
        if (i < L) {
            const uint32_t slot = my_slot[q];
            const uint64_t pos = (uint64_t)my_pos[q];
            merged[slot] = make_uint2(sh.key[i], my_sl[q]);
            sh.mpos[slot] = (uint32_t)pos;
            mhb[slot] = (uint16_t)(((uint32_t)(pos >> 32) & 0xffu) | ((uint32_t)sh.ebwt[my_run[q] & 1023u] << 8));
        }
    }
    __syncthreads();
    // slot i (element i and slot i lie in the same group: a group keeps its slots): suffix-array entry, BWT byte, and the LCP
    // with the slot before it -- inside a group |alpha| - w + the LCP of the two following parse suffixes (a range minimum over
    // the parse's LCP array: pfp_lcp_mum.hpp:295-321), at the first slot of a group the LCP of the two phrase suffixes
    // themselves.  The minimum is min(sl[t1 .. t2 - 1]); sl[t2 - 1] came with the occurrence record, and when other parse
    // suffixes rank between the two -- copies of the locus that left the group through a mutation inside alpha: a few per
    // cent of the slots, but one in nearly every wave -- sl[t1] is requested for all of a work-item's slots BEFORE any of
    // them is used: one round trip per piece instead of one per slot (MMT_EMIT2_ABLATE=1: 52 of 180 ms per C3 pass).
    uint32_t t_lo[PER], t_hi[PER], mn0[PER], sl_first[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = tid + q * BLOCK;
        t_lo[q] = 0; t_hi[q] = 0; mn0[q] = 0; sl_first[q] = 0xffffffffu;
        if (i < L && i != my_gs[q]) {
            const uint2 cur = merged[i];
            t_lo[q] = merged[i - 1].x; t_hi[q] = cur.x; mn0[q] = cur.y;
            if (t_lo[q] != 0 && t_hi[q] > t_lo[q] + 1u && !(a.abl2 & 1u)) sl_first[q] = a.rmq.sl[t_lo[q]];
        }
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = tid + q * BLOCK;
        if (i < L) {
            const uint64_t out = (uint64_t)clo + i;
            const uint32_t hb = mhb[i];
            const uint64_t pos = (uint64_t)sh.mpos[i] | ((uint64_t)(hb & 0xffu) << 32);
            if (out == 0) {                              // entry 0 must be the end sentinel
                if (pos != (uint64_t)a.n) { atomicAdd(a.err, 1u); atomicAdd(a.err + 4, 1u); }
                continue;
            }
            if (pos >= (uint64_t)a.n) {
                atomicAdd(a.err, 1u);
                if (atomicAdd(a.err + 5, 1u) == 0) {       // first offender, for the error message
                    a.err[8] = (uint32_t)pos; a.err[9] = (uint32_t)(pos >> 32);
                    a.err[10] = (uint32_t)out; a.err[11] = (uint32_t)(out >> 32);
                }
                continue;
            }
            const uint64_t j = out - 1;
            if (j < a.win_lo || j >= a.win_hi) continue;
            uint32_t v;
            if (i == my_gs[q]) v = g_head[q].y;
            else {
                const uint32_t t1 = t_lo[q], t2 = t_hi[q];
                if (t1 == 0 || t2 <= t1) { atomicAdd(a.err, 1u); atomicAdd(a.err + 3, 1u); v = 0; }
                else {
                    uint32_t mn = mn0[q] < sl_first[q] ? mn0[q] : sl_first[q];
                    if (t2 - t1 > 2 && !(a.abl2 & 1u)) { const uint32_t rest = rmq_min8(a.rmq, t1 + 1, t2 - 2); mn = rest < mn ? rest : mn; }
                    const uint64_t x = (uint64_t)g_head[q].x - a.w + mn;
                    v = x < (uint64_t)LCP_CAP ? (uint32_t)x : LCP_CAP;
                }
            }
            const uint64_t at = j - a.out_base;
            if (a.abl2 & 4u) { if (v == 0xfffffff3u) a.lcp[at] = v; continue; }
            a.sa.set(at, pos);
            a.bwt[at] = (uint8_t)(hb >> 8);
            a.lcp[at] = j == 0 ? 0u : v;
        }
    }
}

template <int BLOCK, int CAP, int TILE, typename P, typename SA>
__global__ __launch_bounds__(BLOCK, 7) void k_emit2(EmitArgsT<P, SA> a, const EmitDesc* __restrict__ desc, uint32_t n_tiles) {
    __shared__ EmitTile<BLOCK, CAP> sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t t = blockIdx.x;
    if (t >= n_tiles) return;
    if (tid < CAP / 32) sh.omask[tid] = 0u;
    if (tid <= CAP / 32) sh.gmask[tid] = 0u;
    EmitDesc d = desc[t];
    for (;;) {
        const uint32_t t_next = t + gridDim.x;
        EmitDesc dn = d;
        if (t_next < n_tiles) dn = desc[t_next];
        const uint64_t tile = a.tile_lo + t;
        const P tbase = (P)(tile * TILE);
        if (d.L) emit_tile_piece<BLOCK, CAP, P, SA>(a, sh, d.e0, d.e1, tbase + d.clo, d.L);
        // what the first piece left of the tile (see k_emit)
        uint32_t g = d.g_next;
        const uint32_t g_end = d.g_end;
        while (g < g_end) {
            __syncthreads();
            if (wave == 0) {
                const uint64_t first = (uint64_t)a.segb[g] - (uint64_t)tbase;
                const uint64_t lim = first + CAP;
                uint32_t c = 0;                                  // groups after g that still begin at or before lim
                for (uint32_t base = g + 1; base <= g_end; base += 64) {
                    const uint32_t idx = base + lane;
                    const bool ok = idx <= g_end && (uint64_t)a.segb[idx] - (uint64_t)tbase <= lim;
                    const uint64_t m = __ballot(ok);
                    c += (uint32_t)__popcll(m);
                    if (m != ~0ull) break;
                }
                if (lane == 0) {
                    sh.bound[2] = g + c;
                    sh.bound[0] = (uint32_t)first;
                    sh.bound[1] = (uint32_t)((uint64_t)a.segb[g + c] - (uint64_t)tbase - first);
                }
            }
            __syncthreads();
            const uint32_t g2 = sh.bound[2];
            if (g2 > g) {
                const uint32_t clo = sh.bound[0], L = sh.bound[1];
                emit_tile_piece<BLOCK, CAP, P, SA>(a, sh, a.sege[g], a.sege[g2], tbase + clo, L);
                g = g2;
                continue;
            }
            g = g + 1;
        }
        if (t_next >= n_tiles) break;
        t = t_next; d = dn;
    }
}

uint32_t emit_tile() {
    static const uint32_t t = [] {
        const char* e = getenv("MMT_EMIT_TILE");
        const int v = e ? atoi(e) : 896;
        return (uint32_t)(v == 1024 || v == 768 ? v : 896);
    }();
    return t;
}
template <typename P, typename SA, int TILE, int BLOCK = 256>
static void emit_typed(const EmitArgs& a, const uint32_t* tile_first_tab, uint64_t tile_base, void* plan, uint64_t tile_lo, uint64_t tile_hi, hipStream_t s) {
    constexpr int CAP = (int)EMIT_CAP;
    EmitArgsT<P, SA> t;
    t.segb = static_cast<const P*>(a.segb); t.sege = a.sege; t.n_groups = a.n_groups;
    t.ce_eoff = static_cast<const P*>(a.ce_eoff); t.ce_cnt = a.ce_cnt; t.ce_first = a.ce_first; t.ce_offm1 = a.ce_offm1;
    t.ce_bwt = a.ce_bwt; t.ce_gs = a.ce_gs; t.occ = a.occ; t.occ_sl = a.occ_sl; t.pos_bits = a.pos_bits; t.occ12 = a.occ12; t.n = (P)a.n;
    t.sa = SA(a.sa); t.bwt = a.bwt;
    t.fb_group = a.fb_group; t.fb_off = static_cast<const P*>(a.fb_off); t.n_fb = a.n_fb; t.fb_base = (P)a.fb_base;
    t.fb_keys = a.fb_keys; t.fb_vals = static_cast<P*>(a.fb_vals);
    t.bwt_code = a.bwt_code; t.fb_bits = a.fb_bits; t.err = a.err; t.tile_lo = tile_lo;
    t.lcp = a.lcp; t.ghead = static_cast<const uint2*>(a.ghead); t.rmq = a.rmq; t.w = a.w;
    t.out_base = a.out_base; t.win_lo = a.win_lo; t.win_hi = a.win_hi;
    // (tests/micro: MMT_EMIT_MANY=0 sorts every piece, a large value none)
    static const uint32_t many = std::getenv("MMT_EMIT_MANY") ? (uint32_t)std::atoi(std::getenv("MMT_EMIT_MANY")) : 24u;
    t.many_runs = many;
    static const uint32_t abl2 = std::getenv("MMT_EMIT2_ABLATE") ? (uint32_t)std::atoi(std::getenv("MMT_EMIT2_ABLATE")) : 0u;
    t.abl2 = abl2;
    const uint32_t n_tiles = (uint32_t)(tile_hi - tile_lo);
    EmitDesc* desc = static_cast<EmitDesc*>(plan);
    hipLaunchKernelGGL((k_emit_plan<P, TILE, CAP>), dim3(grid_for(n_tiles, 256)), dim3(256), 0, s, t.segb, t.sege, tile_first_tab,
                       tile_base, tile_lo, n_tiles, desc);
    // persistent workgroups: seven per CU (the launch bounds), a few rounds of them so that the tail is short
    // (MMT_EMIT_GRID: workgroups of the launch, tests/micro; 0 = one per tile)
    static const uint32_t grid_env = std::getenv("MMT_EMIT_GRID") ? (uint32_t)std::atoi(std::getenv("MMT_EMIT_GRID")) : 256u * 7u * 4u;
    const uint32_t grid = grid_env ? std::min(grid_env, n_tiles) : n_tiles;
    // MMT_EMIT_ABLATE (tests/micro/emit_ablate.sh, wide 896-element tiles only): the kernel with one of its phases cut out
    // -- WRONG OUTPUT, timing only -- 1 no column stores, 2 no merge ranks, 3 no occurrence records, 4 no group heads,
    // 5 nothing after the entry rows
    static const int abl = std::getenv("MMT_EMIT_ABLATE") ? std::atoi(std::getenv("MMT_EMIT_ABLATE")) : 0;
    if constexpr (TILE == 896 && sizeof(P) == 8) {
        if (abl == 1) { hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA, 1>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles); return; }
        if (abl == 2) { hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA, 2>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles); return; }
        if (abl == 3) { hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA, 3>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles); return; }
        if (abl == 4) { hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA, 4>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles); return; }
        if (abl == 5) { hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA, 5>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles); return; }
    }
    // MMT_EMIT_V1: the first form of the tile kernel (A/B, and the ablations above)
    static const bool v1 = std::getenv("MMT_EMIT_V1") != nullptr;
    if (v1 || (abl && !abl2)) hipLaunchKernelGGL((k_emit<BLOCK, CAP, TILE, P, SA>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles);
    else hipLaunchKernelGGL((k_emit2<BLOCK, CAP, TILE, P, SA>), dim3(grid), dim3(BLOCK), 0, s, t, desc, n_tiles);
    MMT_HIP(hipGetLastError());
}
size_t emit_plan_bytes(uint64_t tiles) { return (size_t)tiles * sizeof(EmitDesc); }
void emit(const EmitArgs& a, const uint32_t* tile_first_tab, uint64_t tile_base, void* plan, uint64_t tile_lo, uint64_t tile_hi, hipStream_t s) {
    if (tile_hi <= tile_lo) return;
    const uint32_t tile = emit_tile();
    if (a.wide) {
        if (tile == 1024) emit_typed<uint64_t, Sa40, 1024>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
        else if (tile == 768) emit_typed<uint64_t, Sa40, 768>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
        else emit_typed<uint64_t, Sa40, 896>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
    } else {
        if (tile == 1024) emit_typed<uint32_t, Sa32, 1024>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
        else if (tile == 768) emit_typed<uint32_t, Sa32, 768>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
        else emit_typed<uint32_t, Sa32, 896>(a, tile_first_tab, tile_base, plan, tile_lo, tile_hi, s);
    }
}

// Oversized groups, unsorted, into the fallback arrays: workgroup b takes chunk (chunk0[f0] + b) of the launch, i.e. EMIT_BIG_CHUNK
// consecutive output offsets of one oversized group; the entries that cover them -- the first may begin before the chunk,
// the last may end behind it -- go through emit_piece's unsorted path.
template <int BLOCK, int CAP, typename P, typename SA>
__global__ __launch_bounds__(BLOCK) void k_emit_big(EmitArgsT<P, SA> a, const uint64_t* __restrict__ chunk0, uint32_t f0, uint32_t nf) {
    __shared__ EmitShared<BLOCK, CAP> sh;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t cb = chunk0[f0] + blockIdx.x;
    if (wave == 0) {
        const uint32_t idx = wave_lower_bound<uint64_t>(chunk0 + f0, nf, cb + 1);      // first group whose chunks begin behind cb
        if (lane == 0) sh.bound[0] = f0 + idx - 1;
    }
    __syncthreads();
    const uint32_t f = sh.bound[0];
    const uint32_t g = a.fb_group[f];
    const uint64_t gb = (uint64_t)a.segb[g], ge = (uint64_t)a.segb[g + 1];
    const uint64_t X0 = gb + (cb - chunk0[f]) * (uint64_t)EMIT_BIG_CHUNK;
    if (X0 >= ge) return;                                   // (cannot happen: chunk0 counts ceil(size / chunk) per group)
    const uint32_t L = ge - X0 < (uint64_t)EMIT_BIG_CHUNK ? (uint32_t)(ge - X0) : EMIT_BIG_CHUNK;
    const P fb_origin = (P)(gb - ((uint64_t)a.fb_off[f] - (uint64_t)a.fb_base));
    const uint32_t e_lo = a.sege[g], e_hi = a.sege[g + 1];
    if (wave == 0) {
        // last entry that begins at or before X0 .. last entry that begins before X0 + L
        const uint32_t first = wave_lower_bound<P>(a.ce_eoff + e_lo, e_hi - e_lo, (P)(X0 + 1));
        const uint32_t end = wave_lower_bound<P>(a.ce_eoff + e_lo, e_hi - e_lo, (P)(X0 + L));
        if (lane == 0) { sh.bound[1] = e_lo + first - 1; sh.bound[2] = e_lo + end; }
    }
    __syncthreads();
    const uint32_t e0 = sh.bound[1], e1 = sh.bound[2];
    emit_piece<BLOCK, CAP, P, SA, uint32_t>(a, sh, e0, e1, (P)X0, L, false, fb_origin, 0u);
}
template <typename P, typename SA>
static void emit_big_typed(const EmitArgs& a, const uint64_t* chunk0, uint32_t f0, uint32_t nf, uint32_t n_chunks, hipStream_t s) {
    constexpr int BLOCK = 256, CAP = (int)EMIT_CAP;
    static_assert(EMIT_BIG_CHUNK + 1 <= EMIT_CAP, "a chunk's entries (one per element + the one that began before it) must fit the tile tables");
    EmitArgsT<P, SA> t;
    t.segb = static_cast<const P*>(a.segb); t.sege = a.sege; t.n_groups = a.n_groups;
    t.ce_eoff = static_cast<const P*>(a.ce_eoff); t.ce_cnt = a.ce_cnt; t.ce_first = a.ce_first; t.ce_offm1 = a.ce_offm1;
    t.ce_bwt = a.ce_bwt; t.ce_gs = a.ce_gs; t.occ = a.occ; t.occ_sl = a.occ_sl; t.pos_bits = a.pos_bits; t.occ12 = a.occ12; t.n = (P)a.n;
    t.sa = SA(a.sa); t.bwt = a.bwt;
    t.fb_group = a.fb_group; t.fb_off = static_cast<const P*>(a.fb_off); t.n_fb = a.n_fb; t.fb_base = (P)a.fb_base;
    t.fb_keys = a.fb_keys; t.fb_vals = static_cast<P*>(a.fb_vals);
    t.bwt_code = a.bwt_code; t.fb_bits = a.fb_bits; t.err = a.err; t.tile_lo = 0;
    t.lcp = a.lcp; t.ghead = static_cast<const uint2*>(a.ghead); t.rmq = a.rmq; t.w = a.w;
    t.out_base = a.out_base; t.win_lo = a.win_lo; t.win_hi = a.win_hi; t.many_runs = 0xffffffffu; t.abl2 = 0;
    hipLaunchKernelGGL((k_emit_big<BLOCK, CAP, P, SA>), dim3(n_chunks), dim3(BLOCK), 0, s, t, chunk0, f0, nf);
    MMT_HIP(hipGetLastError());
}
void emit_big(const EmitArgs& a, const uint64_t* chunk0, uint32_t f0, uint32_t nf, uint64_t n_chunks, hipStream_t s) {
    if (!nf || !n_chunks) return;
    if (n_chunks > 0x7fffffffull) throw std::runtime_error("too many chunks of oversized suffix groups in one emitter launch");
    if (a.wide) emit_big_typed<uint64_t, Sa40>(a, chunk0, f0, nf, (uint32_t)n_chunks, s);
    else emit_big_typed<uint32_t, Sa32>(a, chunk0, f0, nf, (uint32_t)n_chunks, s);
}

// osize[g] = size of group g if it exceeds the emitter's LDS tile, else 0
template <typename P>
__global__ void k_oversize(const P* __restrict__ segb, uint32_t n_groups, uint32_t cap,
                           uint32_t* __restrict__ osize, uint32_t* __restrict__ err) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const uint64_t sz = (uint64_t)segb[g + 1] - (uint64_t)segb[g];
    if (sz >= 0xffffffffull) atomicAdd(err + 2, 1u);       // one suffix group of 2^32 elements: not supported
    osize[g] = sz > cap ? (uint32_t)sz : 0u;
}
void oversize(const void* segb, uint32_t n_groups, uint32_t* osize, uint32_t* err, bool wide, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_oversize<uint64_t>, dim3(grid_for(n_groups, 256)), dim3(256), 0, s,
                           static_cast<const uint64_t*>(segb), n_groups, EMIT_CAP, osize, err);
    else
        hipLaunchKernelGGL(k_oversize<uint32_t>, dim3(grid_for(n_groups, 256)), dim3(256), 0, s,
                           static_cast<const uint32_t*>(segb), n_groups, EMIT_CAP, osize, err);
    MMT_HIP(hipGetLastError());
}

template <typename P>
__global__ void k_gather_pos(const P* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n, P* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
void gather_pos(const void* src, const uint32_t* idx, uint32_t n, void* out, bool wide, hipStream_t s) {
    if (!n) return;
    if (wide)
        hipLaunchKernelGGL(k_gather_pos<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<const uint64_t*>(src),
                           idx, n, static_cast<uint64_t*>(out));
    else
        hipLaunchKernelGGL(k_gather_pos<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<const uint32_t*>(src),
                           idx, n, static_cast<uint32_t*>(out));
    MMT_HIP(hipGetLastError());
}

template <typename P>
__global__ void k_relative_offsets(const P* __restrict__ fb_off, uint32_t f0, uint32_t count, uint32_t* __restrict__ rel) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= count) rel[i] = (uint32_t)(fb_off[f0 + i] - fb_off[f0]);
}
void relative_offsets(const void* fb_off, uint32_t f0, uint32_t count, uint32_t* rel, bool wide, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_relative_offsets<uint64_t>, dim3(grid_for((uint64_t)count + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint64_t*>(fb_off), f0, count, rel);
    else
        hipLaunchKernelGGL(k_relative_offsets<uint32_t>, dim3(grid_for((uint64_t)count + 1, 256)), dim3(256), 0, s,
                           static_cast<const uint32_t*>(fb_off), f0, count, rel);
    MMT_HIP(hipGetLastError());
}

// oversized groups after their segmented sort: sa / bwt / lcp from the sorted (key, position) pairs
struct FinishLcp { uint32_t* lcp; const uint2* ghead; RmqView rmq; uint32_t w; uint64_t out_base, win_lo, win_hi; };
// One work-item per element of the launch's fallback arrays (a group of millions of elements -- every suffix of a
// satellite monomer in every haplotype -- used to be one workgroup's job: 0.84 s of a step); the element's group by a
// binary search over the groups' offsets.
template <typename P, typename SA>
__global__ void k_fallback_finish(const uint32_t* __restrict__ fb_group, const P* __restrict__ fb_off, uint32_t f0,
                                  uint32_t f1, P fb_base, const P* __restrict__ segb,
                                  const uint32_t* __restrict__ sorted_keys, const P* __restrict__ sorted_vals,
                                  uint32_t fb_bits, BwtDecode decode, const TextRef text, P n, SA sa,
                                  uint8_t* __restrict__ bwt, uint32_t* __restrict__ err, FinishLcp F, uint32_t total) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // group f: fb_off[f] - fb_base <= i < fb_off[f + 1] - fb_base
    uint32_t a = f0, b = f1;                                    // answer in [a, b)
    while (b - a > 1) {
        const uint32_t mid = (a + b) >> 1;
        if ((uint32_t)(fb_off[mid] - fb_base) <= i) a = mid; else b = mid;
    }
    const uint32_t f = a;
    const uint32_t lo = (uint32_t)(fb_off[f] - fb_base);
    const uint32_t g = fb_group[f];
    const P out0 = segb[g];
    const uint32_t mask = (1u << fb_bits) - 1u;
    const P p = sorted_vals[i], out = out0 + (i - lo);
    if (out == 0 || p >= n) {                                   // the sentinel never sits in an oversized group
        atomicAdd(err, 1u);
        if (atomicAdd(err + (out == 0 ? 6 : 7), 1u) == 0) {
            err[12] = (uint32_t)p; err[13] = (uint32_t)((uint64_t)p >> 32);
            err[14] = (uint32_t)out; err[15] = (uint32_t)((uint64_t)out >> 32);
        }
        return;
    }
    const uint64_t j = (uint64_t)out - 1;
    if (j < F.win_lo || j >= F.win_hi) return;
    const uint64_t at = j - F.out_base;
    sa.set(at, p);
    if (fb_bits) bwt[at] = decode.byte[sorted_keys[i] & mask];     // rode along in the key
    else bwt[at] = p ? tx_byte(text, (uint64_t)p) : (uint8_t)0;             // V[p] = T[p - 1]
    const uint2 head = F.ghead[g];
    uint32_t v = head.y;
    if (i > lo) {
        const uint32_t t1 = sorted_keys[i - 1] >> fb_bits, t2 = sorted_keys[i] >> fb_bits;
        if (t1 == 0 || t2 <= t1) { atomicAdd(err, 1u); atomicAdd(err + 3, 1u); v = 0; }
        else {
            const uint64_t x = (uint64_t)head.x - F.w + rmq_min(F.rmq, t1, t2 - 1);
            v = x < (uint64_t)LCP_CAP ? (uint32_t)x : LCP_CAP;
        }
    }
    F.lcp[at] = j == 0 ? 0u : v;
}
void fallback_finish(const uint32_t* fb_group, const void* fb_off, uint32_t f0, uint32_t f1, uint64_t fb_base,
                     const void* segb, const uint32_t* sorted_keys, const void* sorted_vals, uint32_t fb_bits,
                     const BwtDecode& decode, const TextRef& text, uint64_t n, const EmitArgs& ea, uint32_t total, bool wide,
                     hipStream_t s) {
    if (f1 <= f0 || !total) return;
    FinishLcp F{ea.lcp, static_cast<const uint2*>(ea.ghead), ea.rmq, ea.w, ea.out_base, ea.win_lo, ea.win_hi};
    if (wide)
        hipLaunchKernelGGL((k_fallback_finish<uint64_t, Sa40>), dim3(grid_for(total, 256)), dim3(256), 0, s, fb_group,
                           static_cast<const uint64_t*>(fb_off), f0, f1, (uint64_t)fb_base, static_cast<const uint64_t*>(segb),
                           sorted_keys, static_cast<const uint64_t*>(sorted_vals), fb_bits, decode, text, (uint64_t)n,
                           Sa40(ea.sa), ea.bwt, ea.err, F, total);
    else
        hipLaunchKernelGGL((k_fallback_finish<uint32_t, Sa32>), dim3(grid_for(total, 256)), dim3(256), 0, s, fb_group,
                           static_cast<const uint32_t*>(fb_off), f0, f1, (uint32_t)fb_base, static_cast<const uint32_t*>(segb),
                           sorted_keys, static_cast<const uint32_t*>(sorted_vals), fb_bits, decode, text, (uint32_t)n,
                           Sa32(ea.sa), ea.bwt, ea.err, F, total);
    MMT_HIP(hipGetLastError());
}

__global__ void k_iota(uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}
void iota(uint32_t* out, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n, 256)), dim3(256), 0, s, out, n);
    MMT_HIP(hipGetLastError());
}

__global__ void k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n,
                             uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
void gather_u64(const uint64_t* src, const uint32_t* idx, uint32_t n, uint64_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, n, out);
    MMT_HIP(hipGetLastError());
}

}}  // namespace mmt::pk
