// parse_lcp.hpp -- LCP, in characters, of the parse suffixes that are adjacent in the parse's suffix array, with a
// range-minimum structure on top: the reference's s_lcp_T + rmq_s_lcp_T (include/pfp.hpp:210-244, :54), which
// pfp_lcp::compute_lcp_suffix (include/pfp_lcp_mum.hpp:295-321) reads as `suffix_len + RMQ(s_lcp_T) - w`.
//
// With it every LCP value of the stream is LOCAL to the piece of the suffix array that is being produced: two
// neighbours with the same phrase suffix alpha share |alpha| - w + min(sl[t1 .. t2 - 1]) characters (t = 1 + rank of the
// parse suffix that follows), two neighbours with different alpha are compared directly and differ before the shorter
// alpha ends.  No PLCP column in text order, no gather through the suffix array (round 2: 4 B per text character +
// one random 64-byte line per suffix).
//
// Construction (parse_lcp.hip), everything O(#phrases):
//   * in suffix-array order of the parse, entry r is irreducible when the phrase before sa_p[r] differs from the
//     phrase before sa_p[r - 1] (or one of them starts the parse): its value is found by comparing V at the two phrase
//     starts (long matches: the wave / workgroup loops of the text-level construction, kernels.hip k_long_lcp);
//   * a reducible entry equals the value of the parse suffix one phrase to the left minus the characters that phrase
//     contributes (plen - w): per parse position q the value is lirr[q*] + pstart[q*] - pstart[q], q* = last
//     irreducible position at or before q -- one running maximum over indices (NOT over values: the sums are not
//     monotone on the parse, a match may end inside the w characters two phrases share);
//   * sl[r] = value at sa_p[r]; block minima over 64 entries and a sparse table over the blocks.
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

#include "device_utils.hpp"
#include "textref.hpp"

namespace mmt {

// what kernels need for a query
struct RmqView {
    const uint32_t* sl = nullptr;      // m entries; sl[0] = 0
    const uint32_t* bmin = nullptr;    // level k at bmin + k * nb: minimum of blocks [b, b + 2^k)
    uint32_t m = 0, nb = 0;
};

// min(sl[a .. b]), a <= b < m
__device__ __forceinline__ uint32_t rmq_min(const RmqView& R, uint32_t a, uint32_t b) {
    uint32_t best = 0xffffffffu;
    if (b - a < 128u) {
        for (uint32_t i = a; i <= b; i++) { const uint32_t x = R.sl[i]; best = x < best ? x : best; }
        return best;
    }
    const uint32_t ba = (a + 63u) >> 6, bb = (b + 1u) >> 6;           // whole blocks [ba, bb)
    for (uint32_t i = a; i < (ba << 6); i++) { const uint32_t x = R.sl[i]; best = x < best ? x : best; }
    for (uint32_t i = bb << 6; i <= b; i++) { const uint32_t x = R.sl[i]; best = x < best ? x : best; }
    if (bb > ba) {
        const uint32_t k = 31u - (uint32_t)__builtin_clz(bb - ba);
        const uint32_t x = R.bmin[(size_t)k * R.nb + ba], y = R.bmin[(size_t)k * R.nb + bb - (1u << k)];
        best = x < best ? x : best;
        best = y < best ? y : best;
    }
    return best;
}

// The same minimum with the loads of a stretch requested EIGHT AT A TIME (indices clamped to the stretch: a value read twice
// does not change a minimum): a lane of the emitter that needs sl[t1 .. t2 - 2] holds its whole wave for as many round trips
// as the loop has iterations, and the common ranges -- the ranks of a few copies of the locus that left the group through a
// mutation -- are a handful of entries in one or two lines.
__device__ __forceinline__ uint32_t rmq_stretch8(const uint32_t* __restrict__ sl, uint32_t a, uint32_t b, uint32_t best) {
    for (uint32_t i = a; i <= b; i += 8u) {
        uint32_t x[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) { const uint32_t j = i + u; x[u] = sl[j <= b ? j : b]; }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) best = x[u] < best ? x[u] : best;
        if (b - i < 8u) break;                                   // (i + 8 may wrap at the top of the range)
    }
    return best;
}
__device__ __forceinline__ uint32_t rmq_min8(const RmqView& R, uint32_t a, uint32_t b) {
    if (b - a < 128u) return rmq_stretch8(R.sl, a, b, 0xffffffffu);
    const uint32_t ba = (a + 63u) >> 6, bb = (b + 1u) >> 6;           // whole blocks [ba, bb)
    uint32_t best = 0xffffffffu;
    if (bb > ba) {
        const uint32_t k = 31u - (uint32_t)__builtin_clz(bb - ba);
        const uint32_t x = R.bmin[(size_t)k * R.nb + ba], y = R.bmin[(size_t)k * R.nb + bb - (1u << k)];
        best = x < y ? x : y;
    }
    if (a < (ba << 6)) best = rmq_stretch8(R.sl, a, (ba << 6) - 1u, best);
    if ((bb << 6) <= b) best = rmq_stretch8(R.sl, bb << 6, b, best);
    return best;
}

// block minima over 64 entries + sparse table over the blocks of any array of m values (bmin: levels * nb entries)
void build_rmq(const uint32_t* vals, uint32_t m, DevBuf<uint32_t>& bmin, uint32_t& nb, uint32_t& levels, hipStream_t s);

struct ParseLcp {
    DevBuf<uint32_t> sl, bmin;
    uint32_t m = 0, nb = 0, levels = 0;
    uint32_t n_irreducible = 0, n_long = 0;
    RmqView view() const { RmqView v; v.sl = sl.get(); v.bmin = bmin.get(); v.m = m; v.nb = nb; return v; }
    void release() { sl.release(); bmin.release(); m = nb = levels = 0; }
    // v: the string V = Dollar . T . Dollar^w (v[q], q = text position + 1), nv = n + 1 + w bytes, zero padded behind;
    // sa_p: suffix array of the parse (m entries); pid[q]: any id that is equal for equal phrases (distinct-phrase id or
    // rank); pstart: V index of the first character of phrase q (uint32_t entries, or uint64_t when `wide`)
    void build(const TextRef& v, uint64_t nv, const uint32_t* sa_p, const uint32_t* pid, const void* pstart, bool wide,
               uint32_t m, DevBuf<uint8_t>& temp, hipStream_t s);
};

}  // namespace mmt
