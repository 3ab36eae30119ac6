// parse_lcp.hip -- construction of ParseLcp (parse_lcp.hpp): the LCP of adjacent parse suffixes in characters and its
// range-minimum structure (the reference's s_lcp_T / rmq_s_lcp_T, include/pfp.hpp:210-244).
#include <hip/hip_runtime.h>

#include <cstdint>
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

__device__ __forceinline__ uint64_t load_u64(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }

constexpr int STEPS = 24;          // 192 characters in the first kernel, the rest through the long-match list

// One thread per entry of the parse's suffix array.  lirr[q] = LCP of parse suffix q with its predecessor when the entry
// is irreducible (0 otherwise), head[q] = q + 1 for irreducible entries (0 otherwise).
template <typename P>
__global__ void k_parse_irr(const uint8_t* __restrict__ v, uint64_t nv, const uint32_t* __restrict__ sa_p,
                            const uint32_t* __restrict__ pid, const P* __restrict__ pstart, uint32_t m,
                            uint32_t* __restrict__ lirr, uint32_t* __restrict__ head, k::LongLcpDst* __restrict__ longs,
                            uint32_t* __restrict__ counts, uint32_t long_cap) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    bool queue = false, irr = false;
    uint64_t p = 0, q = 0;
    uint32_t h = 0, qa = 0;
    if (r < m) {
        qa = sa_p[r];
        const uint32_t qb = r ? sa_p[r - 1] : 0u;
        irr = r == 0 || qa == 0 || qb == 0 || pid[qa - 1] != pid[qb - 1];
        if (irr) {
            if (r) {
                p = (uint64_t)pstart[qa]; q = (uint64_t)pstart[qb];
                const uint64_t room = nv - (p > q ? p : q);
                const uint32_t limit = room < (uint64_t)LCP_CAP ? (uint32_t)room : LCP_CAP;
                bool done = false;
                for (int step = 0; step < STEPS && h < limit; step++) {
                    const uint64_t x = load_u64(v + p + h), y = load_u64(v + q + h);
                    if (x != y) { h += (uint32_t)(__builtin_ctzll(x ^ y) >> 3); done = true; break; }
                    h += 8;
                }
                if (h >= limit) { h = limit; done = true; }
                queue = !done;
            }
            lirr[qa] = queue ? 0u : h;
            head[qa] = qa + 1;
        } else {
            lirr[qa] = 0; head[qa] = 0;
        }
    }
    const uint64_t mi = __ballot(irr);
    if (mi && lane == (uint32_t)__builtin_ctzll(mi)) atomicAdd(counts + 1, (uint32_t)__popcll(mi));
    const uint64_t mq = __ballot(queue);
    if (mq) {
        uint32_t slot0 = 0;
        const int leader = __builtin_ctzll(mq);
        if ((int)lane == leader) slot0 = atomicAdd(counts, (uint32_t)__popcll(mq));
        slot0 = __shfl(slot0, leader, 64);
        if (queue) {
            const uint32_t slot = slot0 + (uint32_t)__popcll(mq & ((1ull << lane) - 1));
            if (slot < long_cap) { longs[slot].p = p; longs[slot].q = q; longs[slot].h = h; longs[slot].d = qa; }
        }
    }
}

// sl[r] = lirr[q*] + pstart[q*] - pstart[q], q = sa_p[r], q* = src[q] - 1 (last irreducible parse position at or before q)
template <typename P>
__global__ void k_parse_sl(const uint32_t* __restrict__ sa_p, const uint32_t* __restrict__ src,
                           const uint32_t* __restrict__ lirr, const P* __restrict__ pstart, uint32_t m,
                           uint32_t* __restrict__ sl, uint32_t* __restrict__ err) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    if (r == 0) { sl[0] = 0; return; }
    const uint32_t q = sa_p[r];
    const uint32_t s = src[q];
    if (s == 0) { atomicAdd(err, 1u); sl[r] = 0; return; }       // (position 0 is irreducible: cannot happen)
    const uint64_t val = (uint64_t)lirr[s - 1] + (uint64_t)pstart[s - 1];
    const uint64_t here = (uint64_t)pstart[q];
    if (val < here) { atomicAdd(err, 1u); sl[r] = 0; return; }   // a reducible entry always keeps at least w characters
    const uint64_t d = val - here;
    sl[r] = d < (uint64_t)LCP_CAP ? (uint32_t)d : LCP_CAP;
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

void ParseLcp::build(const uint8_t* v, uint64_t nv, const uint32_t* sa_p, const uint32_t* pid, const void* pstart, bool wide,
                     uint32_t m_, DevBuf<uint8_t>& temp, hipStream_t s) {
    m = m_;
    nb = (m + 63) / 64;
    levels = 1;
    while ((1u << levels) <= nb) levels++;
    sl.ensure((size_t)m + 64);
    DevBuf<uint32_t> lirr, head, counts, huge;
    DevBuf<uint8_t> longs;
    lirr.ensure(m); head.ensure(m); counts.ensure(4);
    uint32_t cap = std::max<uint32_t>(m / 64 + 4096, 1u << 16);
    for (int attempt = 0;; attempt++) {
        longs.ensure((size_t)cap * sizeof(k::LongLcpDst));
        MMT_HIP(hipMemsetAsync(counts.get(), 0, 16, s));
        if (wide)
            hipLaunchKernelGGL(k_parse_irr<uint64_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, v, nv, sa_p, pid,
                               static_cast<const uint64_t*>(pstart), m, lirr.get(), head.get(),
                               reinterpret_cast<k::LongLcpDst*>(longs.get()), counts.get(), cap);
        else
            hipLaunchKernelGGL(k_parse_irr<uint32_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, v, nv, sa_p, pid,
                               static_cast<const uint32_t*>(pstart), m, lirr.get(), head.get(),
                               reinterpret_cast<k::LongLcpDst*>(longs.get()), counts.get(), cap);
        MMT_HIP(hipGetLastError());
        uint32_t back[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(back, counts.get(), 8, hipMemcpyDeviceToHost, s));
        MMT_HIP(hipStreamSynchronize(s));
        n_long = back[0]; n_irreducible = back[1];
        if (n_long <= cap) break;
        if (attempt) throw std::runtime_error("long-match list overflow in the parse LCP construction");
        cap = n_long + 1024;                                   // rare: once more with the exact size
    }
    if (n_long) {
        huge.ensure((size_t)n_long + 1);
        k::long_lcp_dst(v, nv, longs.get(), n_long, lirr.get(), huge.get(), counts.get() + 2, s);
    }
    prims::inclusive_max_u32(temp, head.get(), head.get(), m, s);
    MMT_HIP(hipMemsetAsync(counts.get() + 3, 0, 4, s));
    if (wide)
        hipLaunchKernelGGL(k_parse_sl<uint64_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, sa_p, head.get(), lirr.get(),
                           static_cast<const uint64_t*>(pstart), m, sl.get(), counts.get() + 3);
    else
        hipLaunchKernelGGL(k_parse_sl<uint32_t>, dim3(grid_for(m, 256)), dim3(256), 0, s, sa_p, head.get(), lirr.get(),
                           static_cast<const uint32_t*>(pstart), m, sl.get(), counts.get() + 3);
    MMT_HIP(hipGetLastError());
    bmin.ensure((size_t)levels * nb + 64);
    hipLaunchKernelGGL(k_block_min, dim3(grid_for((uint64_t)nb * 64, 256)), dim3(256), 0, s, sl.get(), m, bmin.get(), nb);
    for (uint32_t k = 1; k < levels; k++)
        hipLaunchKernelGGL(k_level_min, dim3(grid_for(nb, 256)), dim3(256), 0, s, bmin.get() + (size_t)(k - 1) * nb,
                           bmin.get() + (size_t)k * nb, nb, 1u << (k - 1));
    MMT_HIP(hipGetLastError());
    uint32_t bad = 0;
    MMT_HIP(hipMemcpyAsync(&bad, counts.get() + 3, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    if (bad) throw std::runtime_error("parse LCP construction: " + std::to_string(bad) + " inconsistent entries");
}

}  // namespace mmt
