// parse_lcp.hip -- construction of ParseLcp (parse_lcp.hpp): the LCP of adjacent parse suffixes in characters and its
// range-minimum structure (the reference's s_lcp_T / rmq_s_lcp_T, include/pfp.hpp:210-244).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "parse_lcp.hpp"
#include "prims.hpp"
#include "wide.hpp"

namespace mmt {

namespace {

inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}


// Irreducible entries of the parse's suffix array: the phrase before sa_p[r] differs from the phrase before sa_p[r - 1]
// (or one of the two suffixes starts the parse).  The irreducible ones are appended to a list of (parse position, parse
// position of the predecessor) pairs, and head[q] = q + 1 marks them in parse order (head and lirr arrive zeroed).
template <int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK) void k_parse_mark(const uint32_t* __restrict__ sa_p, const uint32_t* __restrict__ pid,
                                                      uint32_t m, uint32_t* __restrict__ head, uint2* __restrict__ list,
                                                      uint32_t* __restrict__ counts) {
    // one list slot allocation per workgroup (a counter word takes ~150 atomics per microsecond on this part: one per wave
    // of 64 entries was 6 M atomics = 40 ms)
    __shared__ uint32_t s_cnt[PER * (BLOCK / 64)];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t qa[PER], qb[PER];
    bool irr[PER];
    uint64_t mask[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t r = (blockIdx.x * PER + k) * BLOCK + threadIdx.x;
        irr[k] = false; qa[k] = 0; qb[k] = 0;
        if (r < m) {
            qa[k] = sa_p[r];
            qb[k] = r ? sa_p[r - 1] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t r = (blockIdx.x * PER + k) * BLOCK + threadIdx.x;
        if (r < m) {
            irr[k] = r == 0 || qa[k] == 0 || qb[k] == 0 || pid[qa[k] - 1] != pid[qb[k] - 1];
            if (irr[k]) head[qa[k]] = qa[k] + 1;
            if (r == 0) irr[k] = false;                        // (irreducible with LCP 0: nothing to compare)
        }
        mask[k] = __ballot(irr[k]);
        if (lane == 0) s_cnt[k * (BLOCK / 64) + wave] = (uint32_t)__popcll(mask[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int i = 0; i < PER * (BLOCK / 64); i++) { const uint32_t c = s_cnt[i]; s_cnt[i] = total; total += c; }
        s_base = total ? atomicAdd(counts + 1, total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (irr[k])
            list[s_base + s_cnt[k * (BLOCK / 64) + wave] + (uint32_t)__popcll(mask[k] & ((1ull << lane) - 1))] = make_uint2(qa[k], qb[k]);
}

// The comparisons: eight lanes per pair, 64 characters per step (aligned 8-byte words, funnelled), up to CMP_STEPS steps;
// what is still equal then goes to the long-match list (kernels.hip: one wave per match, 512 characters per step and up).
constexpr int CMP_STEPS = 8;
template <typename P>
__global__ void k_parse_cmp(const TextRef T, uint64_t nv, const uint2* __restrict__ list, uint32_t count,
                            const P* __restrict__ pstart, uint32_t* __restrict__ lirr, k::LongLcpDst* __restrict__ longs,
                            uint32_t* __restrict__ counts, uint32_t long_cap) {
    const uint32_t lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const uint32_t e = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const bool live = e < count;
    uint64_t p = 0, q = 0;
    uint32_t qa = 0, limit = 0;
    if (live) {
        const uint2 pr = list[e];
        qa = pr.x;
        p = (uint64_t)pstart[pr.x]; q = (uint64_t)pstart[pr.y];
        const uint64_t room = nv - (p > q ? p : q);
        limit = room < (uint64_t)LCP_CAP ? (uint32_t)room : LCP_CAP;
    }
    const uint8_t* const v = T.v;                              // (nullptr: packed text, textref.hpp)
    const uint8_t* pa = v + (p & ~7ull);
    const uint8_t* qb = v + (q & ~7ull);
    const uint32_t sp = (uint32_t)(p & 7u) * 8, sq = (uint32_t)(q & 7u) * 8;
    uint32_t h = 0;
    bool done = !live;
    for (int step = 0; step < CMP_STEPS; step++) {
        const uint32_t o = h + sub * 8;
        uint64_t d = 0;
        if (!done && o < limit) {
            if (v) {
                const uint64_t xl = *reinterpret_cast<const uint64_t*>(pa + o), xh = *reinterpret_cast<const uint64_t*>(pa + o + 8);
                const uint64_t yl = *reinterpret_cast<const uint64_t*>(qb + o), yh = *reinterpret_cast<const uint64_t*>(qb + o + 8);
                const uint64_t x = sp ? (xl >> sp) | (xh << (64 - sp)) : xl;
                const uint64_t y = sq ? (yl >> sq) | (yh << (64 - sq)) : yl;
                d = x ^ y;
            } else d = tx_load8(T, p + o) ^ tx_load8(T, q + o);
        }
        const uint64_t mall = __ballot(d != 0);
        const uint32_t mg = (uint32_t)(mall >> (grp * 8)) & 0xffu;       // the same for the eight lanes of a pair
        const int fl = mg ? __builtin_ctz(mg) : 0;
        const uint64_t dx = __shfl(d, (int)(grp * 8) + fl, 64);         // (every lane takes part: no shuffle under divergence)
        if (!done) {
            if (mg) { h += (uint32_t)fl * 8 + (uint32_t)(__builtin_ctzll(dx) >> 3); done = true; }
            else { h += 64; if (h >= limit) done = true; }
        }
        if (__ballot(!done) == 0) break;
    }
    if (h > limit) h = limit;
    const bool queue = live && !done;
    if (live && done && sub == 0) lirr[qa] = h;
    const uint64_t mq = __ballot(queue && sub == 0);
    if (mq) {
        uint32_t slot0 = 0;
        const int leader = __builtin_ctzll(mq);
        if ((int)lane == leader) slot0 = atomicAdd(counts, (uint32_t)__popcll(mq));
        slot0 = __shfl(slot0, leader, 64);
        if (queue && sub == 0) {
            const uint32_t slot = slot0 + (uint32_t)__popcll(mq & ((1ull << lane) - 1));
            if (slot < long_cap) { longs[slot].p = p; longs[slot].q = q; longs[slot].h = h; longs[slot].d = qa; }
        }
    }
}

// In parse order: value[q] = lirr[q*] + pstart[q*] - pstart[q], q* = src[q] - 1 = last irreducible position at or before q
// (in place: src becomes the value)
template <typename P>
__global__ void k_parse_values(uint32_t* __restrict__ src, const uint32_t* __restrict__ lirr, const P* __restrict__ pstart,
                               uint32_t m, uint32_t* __restrict__ err) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const uint32_t s = src[q];
    if (s == 0) { atomicAdd(err, 1u); return; }                // (position 0 is irreducible: cannot happen)
    const uint64_t val = (uint64_t)lirr[s - 1] + (uint64_t)pstart[s - 1];
    const uint64_t here = (uint64_t)pstart[q];
    if (val < here) { atomicAdd(err, 1u); src[q] = 0; return; }   // a reducible entry always keeps at least w characters
    const uint64_t d = val - here;
    src[q] = d < (uint64_t)LCP_CAP ? (uint32_t)d : LCP_CAP;
}
// sl[r] = value[sa_p[r]]
__global__ void k_parse_sl(const uint32_t* __restrict__ sa_p, const uint32_t* __restrict__ value, uint32_t m,
                           uint32_t* __restrict__ sl) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    sl[r] = r ? value[sa_p[r]] : 0u;
}

// minimum of every block of 64 entries: one wave per block
__global__ void k_block_min(const uint32_t* __restrict__ sl, uint32_t m, uint32_t* __restrict__ bmin, uint32_t nb) {
    const uint32_t b = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (b >= nb) return;
    const uint32_t i = (b << 6) + lane;
    uint32_t x = i < m ? sl[i] : 0xffffffffu;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const uint32_t y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
    if (lane == 0) bmin[b] = x;
}
__global__ void k_level_min(const uint32_t* __restrict__ below, uint32_t* __restrict__ above, uint32_t nb, uint32_t half) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t x = below[b];
    const uint32_t y = b + half < nb ? below[b + half] : 0xffffffffu;   // (ranges that run past the end are never queried)
    above[b] = y < x ? y : x;
}

}  // namespace

void build_rmq(const uint32_t* vals, uint32_t m, DevBuf<uint32_t>& bmin, uint32_t& nb, uint32_t& levels, hipStream_t s) {
    nb = (m + 63) / 64;
    levels = 1;
    while ((1u << levels) <= nb) levels++;
    bmin.ensure((size_t)levels * nb + 64);
    hipLaunchKernelGGL(k_block_min, dim3(grid_for((uint64_t)nb * 64, 256)), dim3(256), 0, s, vals, m, bmin.get(), nb);
    for (uint32_t k = 1; k < levels; k++)
        hipLaunchKernelGGL(k_level_min, dim3(grid_for(nb, 256)), dim3(256), 0, s, bmin.get() + (size_t)(k - 1) * nb,
                           bmin.get() + (size_t)k * nb, nb, 1u << (k - 1));
    MMT_HIP(hipGetLastError());
}

void ParseLcp::build(const TextRef& v, uint64_t nv, const uint32_t* sa_p, const uint32_t* pid, const void* pstart, bool wide,
                     uint32_t m_, DevBuf<uint8_t>& temp, hipStream_t s) {
    m = m_;
    nb = (m + 63) / 64;
    levels = 1;
    while ((1u << levels) <= nb) levels++;
    sl.ensure((size_t)m + 64);
    DevBuf<uint32_t> lirr, head, counts, huge;
    DevBuf<uint2> list;
    DevBuf<uint8_t> longs;
    lirr.ensure(m); head.ensure(m); counts.ensure(4); list.ensure((size_t)m + 64);
    MMT_HIP(hipMemsetAsync(lirr.get(), 0, (size_t)m * 4, s));
    MMT_HIP(hipMemsetAsync(head.get(), 0, (size_t)m * 4, s));
    MMT_HIP(hipMemsetAsync(counts.get(), 0, 16, s));
    hipLaunchKernelGGL((k_parse_mark<256, 4>), dim3(grid_for(m, 1024)), dim3(256), 0, s, sa_p, pid, m, head.get(), list.get(),
                       counts.get());
    MMT_HIP(hipGetLastError());
    MMT_HIP(hipMemcpyAsync(&n_irreducible, counts.get() + 1, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    uint32_t cap = std::max<uint32_t>(n_irreducible / 2 + 4096, 1u << 16);
    for (int attempt = 0;; attempt++) {
        longs.ensure((size_t)cap * sizeof(k::LongLcpDst));
        MMT_HIP(hipMemsetAsync(counts.get(), 0, 4, s));
        for (uint32_t first = 0; first < n_irreducible; first += 1u << 27) {      // (eight lanes per pair: slices of 2^27 pairs)
            const uint32_t part = std::min<uint32_t>(1u << 27, n_irreducible - first);
            if (wide)
                hipLaunchKernelGGL(k_parse_cmp<uint64_t>, dim3(grid_for((uint64_t)part * 8, 256)), dim3(256), 0, s, v, nv,
                                   list.get() + first, part, static_cast<const uint64_t*>(pstart), lirr.get(),
                                   reinterpret_cast<k::LongLcpDst*>(longs.get()), counts.get(), cap);
            else
                hipLaunchKernelGGL(k_parse_cmp<uint32_t>, dim3(grid_for((uint64_t)part * 8, 256)), dim3(256), 0, s, v, nv,
                                   list.get() + first, part, static_cast<const uint32_t*>(pstart), lirr.get(),
                                   reinterpret_cast<k::LongLcpDst*>(longs.get()), counts.get(), cap);
            MMT_HIP(hipGetLastError());
        }
        MMT_HIP(hipMemcpyAsync(&n_long, counts.get(), 4, hipMemcpyDeviceToHost, s));
        MMT_HIP(hipStreamSynchronize(s));
        if (n_long <= cap) break;
        if (attempt) throw std::runtime_error("long-match list overflow in the parse LCP construction");
        cap = n_long + 1024;                                   // rare: once more with the exact size
    }
    list.release();
    if (n_long) {
        huge.ensure((size_t)n_long + 1);
        k::long_lcp_dst(v, nv, longs.get(), n_long, lirr.get(), huge.get(), counts.get() + 2, s);
    }
    prims::inclusive_max_u32(temp, head.get(), head.get(), m, s);
    MMT_HIP(hipMemsetAsync(counts.get() + 3, 0, 4, s));
    if (wide)
        hipLaunchKernelGGL(k_parse_values<uint64_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, head.get(), lirr.get(),
                           static_cast<const uint64_t*>(pstart), m, counts.get() + 3);
    else
        hipLaunchKernelGGL(k_parse_values<uint32_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, head.get(), lirr.get(),
                           static_cast<const uint32_t*>(pstart), m, counts.get() + 3);
    hipLaunchKernelGGL(k_parse_sl, dim3(grid_for(m, 256)), dim3(256), 0, s, sa_p, head.get(), m, sl.get());
    MMT_HIP(hipGetLastError());
    build_rmq(sl.get(), m, bmin, nb, levels, s);
    uint32_t bad = 0;
    MMT_HIP(hipMemcpyAsync(&bad, counts.get() + 3, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    if (bad) throw std::runtime_error("parse LCP construction: " + std::to_string(bad) + " inconsistent entries");
    if (std::getenv("MMT_LCP_STATS"))
        std::fprintf(stderr, "[parse lcp] %u parse suffixes, %u irreducible, %u matches beyond %d characters\n", m, n_irreducible,
                     n_long, CMP_STEPS * 64);
}

}  // namespace mmt
