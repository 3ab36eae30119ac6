// guided_kernels.hip -- kernels of the parse-guided suffix sort (guided.cpp).
//
// The prefix-free parse orders the suffixes of the text by (phrase suffix alpha, rank of the parse suffix that
// follows): pfp_lcp_mum.hpp:115-231, pfp.hpp:171-244.  The reference -- and pfp.cpp here -- gets the first component
// from the suffix array of the dictionary (dictionary.hpp:133).  A collection with little redundancy (the anchor and
// one other whole genome: the two strands of a genome share nothing) has a dictionary as large as half the text, and
// its suffix array no longer fits; this file sorts the text suffixes themselves instead:
//   * the suffixes are dealt into batches by their leading characters (a histogram over at most 4096 bins), so that a
//     batch is one contiguous piece of the suffix array and fits the device next to the columns;
//   * a batch is sorted by its first 63 bits of characters (device radix sort), then groups of equal keys are refined
//     63 bits at a time -- but only up to the end of alpha, the phrase the suffix starts in: phrase suffixes are
//     prefix-free, so two suffixes still tied there have the same alpha and the ranks of the following parse suffixes
//     (parse.hpp:85: 32-bit doubling sort of the parse, sorter.cpp) decide.  Deep matches between the documents cost
//     nothing: they live in the parse.
// The same rounds sort the distinct phrases themselves (their ranks are the parse's alphabet, newscan.hpp:386-404).
//
// Positions are V indices (v[q], q = text position + 1; pfp_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "device_utils.hpp"
#include "guided_kernels.hpp"

namespace mmt { namespace gk {

static inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}

__device__ __forceinline__ uint64_t load_u64(const uint8_t* p) { uint64_t x; __builtin_memcpy(&x, p, 8); return x; }

// ---- phrase ends: rank and successor queries on the cut bits ------------------------------------------------------
__global__ void k_rank_counts(const uint64_t* __restrict__ mask, uint64_t n_words, uint32_t* __restrict__ counts,
                              uint64_t n_counts) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_counts) return;
    uint32_t c = 0;
    for (int t = 0; t < 8; t++) { const uint64_t wi = j * 8 + t; if (wi < n_words) c += (uint32_t)__popcll(mask[wi]); }
    counts[j] = c;
}
void rank_counts(const uint64_t* mask, uint64_t n_words, uint32_t* counts, uint64_t n_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_rank_counts, dim3(grid_for(n_counts, 256)), dim3(256), 0, s, mask, n_words, counts, n_counts);
    MMT_HIP(hipGetLastError());
}
__global__ void k_block_first_cut(const uint64_t* __restrict__ mask, uint64_t n_words, uint64_t* __restrict__ first,
                                  uint64_t n_blocks) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint64_t r = ~0ull;
    for (int t = 0; t < 64; t++) {
        const uint64_t wi = b * 64 + t;
        if (wi >= n_words) break;
        const uint64_t x = mask[wi];
        if (x) { r = wi * 64 + (uint64_t)__builtin_ctzll(x); break; }
    }
    first[b] = r;
}
// phrase ends per block of 4096 positions (64 words), then -- after an exclusive sum of those -- their offsets inside the block
__global__ void k_block_cut_counts(const uint64_t* __restrict__ mask, uint64_t n_words, uint32_t* __restrict__ counts, uint64_t n_blocks) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t c = 0;
    for (int t = 0; t < 64; t++) { const uint64_t wi = b * 64 + t; if (wi < n_words) c += (uint32_t)__popcll(mask[wi]); }
    counts[b] = c;
}
__global__ void k_block_cut_offsets(const uint64_t* __restrict__ mask, uint64_t n_words, const uint32_t* __restrict__ brank,
                                    uint16_t* __restrict__ coff, uint64_t n_blocks) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t at = brank[b];
    for (int t = 0; t < 64; t++) {
        const uint64_t wi = b * 64 + t;
        if (wi >= n_words) break;
        uint64_t x = mask[wi];
        while (x) { coff[at++] = (uint16_t)(t * 64 + __builtin_ctzll(x)); x &= x - 1; }
    }
}
void block_cut_counts(const uint64_t* mask, uint64_t n_words, uint32_t* counts, uint64_t n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_block_cut_counts, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, mask, n_words, counts, n_blocks);
    MMT_HIP(hipGetLastError());
}
void block_cut_offsets(const uint64_t* mask, uint64_t n_words, const uint32_t* brank, uint16_t* coff, uint64_t n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_block_cut_offsets, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, mask, n_words, brank, coff, n_blocks);
    MMT_HIP(hipGetLastError());
}
void block_first_cut(const uint64_t* mask, uint64_t n_words, uint64_t* first, uint64_t n_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_block_first_cut, dim3(grid_for(n_blocks, 256)), dim3(256), 0, s, mask, n_words, first, n_blocks);
    MMT_HIP(hipGetLastError());
}

// The phrase ends as a LIST instead of a bit per text position (Ctx::coff): the offsets inside their blocks of 4096 positions,
// two bytes per phrase, and the number of phrase ends before every block -- 3.8 GB instead of the 81 GB the bits and their
// rank directory take on 573 G characters (BASELINE configs[4]).  A block's entries are found by its two counts.
// (an 8-ary search: seven pivots asked for at once, then the last eight entries at once -- two or three round trips to memory
// for the 26 ... 140 phrase ends of a block where a binary search makes five to eight, each waiting for the one before)
__device__ __forceinline__ uint32_t list_lower(const Ctx& c, uint32_t lo, uint32_t hi, uint32_t t) {
    if (hi <= lo) return lo;
    while (hi - lo > 8) {
        const uint32_t step = (hi - lo + 7) >> 3;
        uint32_t pv[7];
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) { const uint32_t at = lo + (k + 1) * step; pv[k] = at < hi ? (uint32_t)c.coff[at] : 0xffffu; }
        uint32_t below = 0;                                       // (ascending: the pivots below t are the first `below`)
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) below += pv[k] < t ? 1u : 0u;
        const uint32_t nlo = below ? lo + below * step + 1 : lo;
        const uint32_t nhi = below < 7 && lo + (below + 1) * step < hi ? lo + (below + 1) * step : hi;
        lo = nlo; hi = nhi;
    }
    uint32_t less = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { const uint32_t v = lo + k < hi ? (uint32_t)c.coff[lo + k] : 0xffffu; less += v < t ? 1u : 0u; }
    return lo + less;                                             // first entry of [lo, hi) with offset >= t
}
__device__ __forceinline__ uint64_t list_next_cut(const Ctx& c, uint64_t x) {
    const uint64_t b = x >> 12;
    const uint32_t lo = c.brank[b], hi = c.brank[b + 1];
    const uint32_t k = list_lower(c, lo, hi, (uint32_t)(x & 4095));
    return k < hi ? (b << 12) + (uint64_t)c.coff[k] : c.nxt[b + 1];
}
// first phrase end at or after text position x; `first` = c.mask[x >> 6], loaded by the caller (x < n)
__device__ __forceinline__ uint64_t next_cut_from(const Ctx& c, uint64_t x, uint64_t first) {
    if (c.coff) return list_next_cut(c, x);
    uint64_t wi = x >> 6;
    uint64_t word = first & (~0ull << (x & 63));
    const uint64_t stop = ((x >> 12) + 1) << 6;                   // first word of the next block of 4096 positions
    for (;;) {
        if (word) return wi * 64 + (uint64_t)__builtin_ctzll(word);
        if (++wi == stop) return c.nxt[(x >> 12) + 1];
        word = c.mask[wi];
    }
}
// first phrase end at or after text position x
__device__ __forceinline__ uint64_t next_cut(const Ctx& c, uint64_t x) {
    if (x >= c.n) return c.n + c.w - 1;
    if (c.coff) return list_next_cut(c, x);
    uint64_t wi = x >> 6;
    uint64_t word = c.mask[wi] & (~0ull << (x & 63));
    const uint64_t stop = ((x >> 12) + 1) << 6;                   // first word of the next block of 4096 positions
    for (;;) {
        if (word) return wi * 64 + (uint64_t)__builtin_ctzll(word);
        if (++wi == stop) return c.nxt[(x >> 12) + 1];
        word = c.mask[wi];
    }
}
// number of phrase ends before text position x
__device__ __forceinline__ uint32_t rank1(const Ctx& c, uint64_t x) {
    if (x > c.n) x = c.n;                                         // no cut bits at or beyond n
    if (c.coff) { const uint64_t b = x >> 12; return list_lower(c, c.brank[b], c.brank[b + 1], (uint32_t)(x & 4095)); }
    uint32_t r = c.rdir[x >> 9];
    const uint64_t w0 = (x >> 9) << 3, w1 = x >> 6;
    for (uint64_t wi = w0; wi < w1; wi++) r += (uint32_t)__popcll(c.mask[wi]);
    if (x & 63) r += (uint32_t)__popcll(c.mask[w1] & ((1ull << (x & 63)) - 1));
    return r;
}
// the query point of an element: its phrase is rank1(x), its alpha ends at next_cut(x) (text coordinates)
__device__ __forceinline__ uint64_t query_point(const Ctx& c, uint64_t q) { return q + c.skip + c.w - 2; }
__device__ __forceinline__ uint64_t alpha_len(const Ctx& c, uint64_t q) { return next_cut(c, query_point(c, q)) + 2 - q; }

// An element record is the V index of its first character plus, above it, either the length of its alpha (saturated:
// looked up again) or -- text suffixes, when the parse rank fits the bits the position leaves -- the rank of the parse
// suffix that follows, which saves the three random lines of that lookup where pairs are compared.
__device__ __forceinline__ uint64_t rec_pos(const Ctx& c, uint64_t rec) { return rec & ((1ull << c.pos_bits) - 1); }
__device__ __forceinline__ uint64_t rec_len(const Ctx& c, uint64_t rec, uint64_t q) {
    if (c.rec_rank) return alpha_len(c, q);
    const uint64_t len = rec >> 40;
    return len == LEN_SAT ? alpha_len(c, q) : len;
}
__device__ __forceinline__ uint64_t rec_rank_key(const Ctx& c, uint64_t rec, uint64_t q) {
    if (c.expand) return q;                                       // representatives: any consistent order of a group will do
    if (c.rec_rank) return rec >> c.pos_bits;
    const uint32_t k = rank1(c, query_point(c, q));
    return k + 1 < c.m ? (uint64_t)c.isa_p[k + 1] : 0ull;
}
__device__ __forceinline__ uint64_t make_rec(const Ctx& c, uint64_t q) {
    if (c.rec_rank) {
        const uint32_t k = rank1(c, query_point(c, q));
        return q | ((uint64_t)(k + 1 < c.m ? c.isa_p[k + 1] : 0u) << c.pos_bits);
    }
    const uint64_t len = alpha_len(c, q);
    return q | ((len < LEN_SAT ? len : (uint64_t)LEN_SAT) << 40);
}

// ---- giant phrases -----------------------------------------------------------------------------------------------------
// the rest of alpha of the element at V index q, from `off` characters on, as an entry of the giant dictionary's suffix
// array; false when the element does not lie in a giant phrase
__device__ __forceinline__ bool giant_entry(const Ctx& c, uint64_t q, uint64_t off, uint32_t& r) {
    if (!c.g_n) return false;
    const uint32_t k = rank1(c, query_point(c, q));
    const uint32_t word = c.g_bits[k >> 5];
    if (!((word >> (k & 31)) & 1u)) return false;
    // (its place among the giant occurrences: a rank on the bits -- the binary search over the list of occurrences was twenty
    // dependent loads per lookup, and a comparison of two suffixes inside giant phrases asks twice)
    const uint32_t lo = c.g_rank[k >> 5] + (uint32_t)__popc(word & ((1u << (k & 31)) - 1u));
    if (lo >= c.g_n) return false;
    r = c.g_isa[(uint64_t)c.g_base[lo] + (q + off - c.g_ps[lo])];
    return true;
}
// The first t in [from, stop) with V[qa + t] != V[qb + t] (ca, cb = the two characters), or `stop`.  A walk is a chain of
// dependent round trips to memory, so each trip carries as much as a lane can ask for at once: 64 characters a side as
// words of codes when the text is packed, else 32 characters as four independent 8-byte loads a side (those that begin
// before `stop`: nothing is read that the loop of single words would not have read).
__device__ __forceinline__ uint64_t first_diff(const Ctx& c, uint64_t qa, uint64_t qb, uint64_t from, uint64_t stop,
                                               uint32_t& ca, uint32_t& cb) {
    uint64_t t = from;
    while (t < stop) {
        if (c.T.is_packed()) {
            // (64 characters a side whatever is left to compare: the words are there, a difference at or behind `stop` is none)
            uint64_t xl, xh, yl, yh;
            const bool oka = tx_codes64(c.T, qa + t, xl, xh), okb = tx_codes64(c.T, qb + t, yl, yh);
            if (oka && okb) {
                if (xl != yl || xh != yh) {
                    const bool low = xl != yl;
                    const uint64_t x = low ? xl : xh, y = low ? yl : yh;
                    const uint32_t k = (uint32_t)__builtin_ctzll(x ^ y) >> 1;
                    const uint64_t at = t + (low ? 0u : 32u) + k;
                    if (at >= stop) return stop;
                    ca = tx_code_char((uint32_t)(x >> (2 * k)) & 3u); cb = tx_code_char((uint32_t)(y >> (2 * k)) & 3u);
                    return at;
                }
                t += 64;
                continue;
            }
        }
        uint64_t x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool in = t + 8 * k < stop;
            x[k] = in ? tx_load8(c.T, qa + t + 8 * k) : 0ull; y[k] = in ? tx_load8(c.T, qb + t + 8 * k) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (x[k] != y[k]) {
                const uint32_t d = (uint32_t)__builtin_ctzll(x[k] ^ y[k]) >> 3;
                const uint64_t at = t + 8 * k + d;
                if (at >= stop) return stop;
                ca = (uint32_t)(x[k] >> (8 * d)) & 0xffu; cb = (uint32_t)(y[k] >> (8 * d)) & 0xffu;
                return at;
            }
        t += 32;
    }
    return stop;
}

// Order of alpha(qa) and alpha(qb) (la, lb characters), known to agree in their first `from` characters: -1 / +1, or 0 when
// they are the same phrase suffix (prefix-free: no difference before the shorter one ends).  *lcp = characters the two
// suffixes share when a character decided.  Beyond g_depth characters two alphas that both go on lie in giant phrases and
// the giant dictionary decides: one lookup instead of a comparison that may run for a megabase.
__device__ __forceinline__ int cmp_rest(const Ctx& c, uint64_t qa, uint64_t la, uint64_t qb, uint64_t lb, uint64_t from,
                                        uint64_t* lcp) {
    const uint64_t L = la < lb ? la : lb;
    // (64 characters are compared first -- most pairs differ there; two alphas that go on may both lie in giant phrases)
    const bool giant = c.g_n && from + 64 < L;
    const uint64_t stop = giant ? from + 64 : L;
    uint32_t ca = 0, cb = 0;
    if (from < stop) {
        const uint64_t at = first_diff(c, qa, qb, from, stop, ca, cb);
        if (at < stop) { if (lcp) *lcp = at; return ca < cb ? -1 : 1; }
    }
    if (!giant) return 0;
    uint32_t ra = 0, rb = 0;
    if (!giant_entry(c, qa, stop, ra) || !giant_entry(c, qb, stop, rb)) {
        if (stop < L) {                                         // one of them in an ordinary phrase: at most g_depth characters
            const uint64_t at = first_diff(c, qa, qb, stop, L, ca, cb);
            if (at < L) { if (lcp) *lcp = at; return ca < cb ? -1 : 1; }
        }
        return 0;
    }
    if (c.g_grp[ra] == c.g_grp[rb]) return 0;
    if (lcp) *lcp = stop + rmq_min(c.g_rmq, (ra < rb ? ra : rb) + 1, ra < rb ? rb : ra);
    return ra < rb ? -1 : 1;
}

// up to c.chars symbol codes of v[from ...], most significant first
__device__ __forceinline__ uint64_t pack_chars(const Ctx& c, const uint8_t* __restrict__ s_code, uint64_t from) {
    uint64_t key = 0;
    int done = 0;
    while (done < c.chars) {
        uint64_t x = tx_load8(c.T, from + done);
        const int take = c.chars - done < 8 ? c.chars - done : 8;
        for (int t = 0; t < take; t++) { key = (key << c.bits) | s_code[x & 0xff]; x >>= 8; }
        done += take;
    }
    return key;
}

// ---- text-order kernels: one workgroup per TILE text positions, 16 consecutive positions per work-item ----------------
// Every kernel stages the symbol codes of its tile (+ one key of lookahead) in LDS and rolls the key along.
template <typename F>
__device__ __forceinline__ void for_tile_keys(const Ctx& c, uint8_t* s_sym, F&& f) {
    constexpr int BLOCK = 256, PER = TILE / BLOCK;
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += BLOCK) s_code[i] = c.code[i];
    __syncthreads();
    const uint64_t base = ((uint64_t)blockIdx.x + c.tile0) * TILE; // text position of the tile's first suffix
    for (uint32_t i = threadIdx.x; i < TILE + 64; i += BLOCK) {
        const uint64_t p = base + i;
        s_sym[i] = p < c.n + 40 ? s_code[tx_byte(c.T, p + 1)] : (uint8_t)0;       // (the text is padded: Engine::text_ptr, textref.hpp)
    }
    __syncthreads();
    const int t0 = threadIdx.x * PER;
    const uint64_t kmask = (c.bits * c.chars >= 64) ? ~0ull : ((1ull << (c.bits * c.chars)) - 1);
    uint64_t key = 0;
    for (int ch = 0; ch < c.chars; ch++) key = (key << c.bits) | s_sym[t0 + ch];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint64_t i = base + t0 + q;
        f(q, i, i < c.n, key & kmask);
        key = ((key << c.bits) | s_sym[t0 + q + c.chars]) & kmask;
    }
}

// A launch may not have 2^32 work-items: the text-order kernels of a text beyond 2^36 characters (whole genomes: the
// anchor next to twelve haplotypes is 79 G) run as slices of 2^23 tiles, the first tile of a slice in Ctx::tile0.
template <typename F>
static void for_tile_slices(const Ctx& c, F&& launch) {
    const uint64_t tiles = (c.n + TILE - 1) / TILE, SLICE = 1ull << 23;
    for (uint64_t t0 = 0; t0 < tiles || t0 == 0; t0 += SLICE) {
        Ctx cs = c;
        cs.tile0 = (uint32_t)t0;
        const uint64_t cnt = std::min<uint64_t>(SLICE, tiles > t0 ? tiles - t0 : 1);
        launch(cs, (unsigned)cnt);
        if (tiles <= t0 + SLICE) break;
    }
}

// Which suffixes of a tile belong to the bins [bin_lo, bin_hi)?  Only the first `pc` symbols decide, so the pass over the
// text rolls a bin code of pc * bits bits (a batch re-reads the whole text: at 79 G characters and 86 batches the two
// text-order kernels were half of the run while they rolled full 63-bit keys for every position).
struct NoTileHook { __device__ __forceinline__ void operator()() const {} };
template <typename F, typename H = NoTileHook>
__device__ __forceinline__ void for_tile_bins(const Ctx& c, uint64_t tile, int pc, uint8_t* s_sym, F&& f, H&& staged = H()) {
    constexpr int BLOCK = 256, PER = TILE / BLOCK;
    static_assert(PER == 16, "one 16-byte load per work-item");
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += BLOCK) s_code[i] = c.code[i];
    __syncthreads();
    const uint64_t base = tile * TILE; // text position of the tile's first suffix
    // 16 consecutive characters per work-item in one load, turned into symbol codes in registers and staged as one
    // 16-byte LDS store (byte loads and byte stores made these kernels run at 0.4 - 0.8 TB/s of a 1 B / character stream)
    auto codes_at = [&](uint64_t p0) {
        union { uint4 v; uint8_t b[16]; } raw, out;
        raw.v = p0 + 16 <= c.n + 64 ? tx_load16(c.T, p0) : make_uint4(0u, 0u, 0u, 0u);      // (16-byte aligned, padded: Engine::text_ptr)
#pragma unroll
        for (int k = 0; k < 16; k++) out.b[k] = p0 + k < c.n + 40 ? s_code[raw.b[k]] : (uint8_t)0;
        return out.v;
    };
    const int t0 = threadIdx.x * PER;
    union { uint4 v; uint8_t b[16]; } mine;
    mine.v = codes_at(base + t0);
    *reinterpret_cast<uint4*>(s_sym + t0) = mine.v;
    if (threadIdx.x < 4) *reinterpret_cast<uint4*>(s_sym + TILE + 16 * threadIdx.x) = codes_at(base + TILE + 16 * threadIdx.x);
    __syncthreads();
    staged();                                   // (every work-item of the workgroup: the hook may hold barriers)
    const uint32_t bmask = (1u << (c.bits * pc)) - 1u;
    uint32_t bin = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        // bin of the suffix at t0 + q = symbols t0 + q .. t0 + q + pc - 1: roll in symbol t0 + q + pc - 1
        if (q == 0) { for (int ch = 0; ch < pc; ch++) bin = (bin << c.bits) | s_sym[t0 + ch]; }
        else bin = ((bin << c.bits) | s_sym[t0 + q + pc - 1]) & bmask;
        f(q, base + t0 + q < c.n, bin & bmask);
    }
}

// ---- tiles of a PACKED text that hold no exception: the suffixes straight from the words of 2-bit codes -------------------------
// A pass over the text is a pass over 573 G characters on every rank of configs[4], nineteen times: staging symbol codes in LDS
// (a table lookup and a byte store per character, a byte load per character and leading symbol, four workgroup barriers) was
// most of its cost.  A tile -- one block of the exception flags -- without an exception, with 64 plain characters behind it, is
// read as words: the dense code of a suffix (two bits per leading base, first base most significant) is a shift and a mask of
// the work-item's 64-bit window with its 2-bit groups reversed.  The batches' bin ranges are ranges of dense codes too
// (dense_lower: A < C < G < T in both numberings); only what is SELECTED turns its dense code back into symbol codes.
// What a tile's kernel needs from memory before anything can be computed, asked for TOGETHER at its top: a pass is a chain of
// round trips to memory per tile (exception flags -> words of text; phrase ends -> rank -> representative bits; list offset;
// phrase ends again for the records), eight tiles to a compute unit -- ten microseconds a tile, 0.3 TB/s of a text that is
// a quarter of a byte per character.  Loads that depend on nothing but the tile's number leave at once.
struct TileLoads {
    uint64_t w0 = 0, w1 = 0;          // the work-item's words of a packed text (meaningful when `plain`)
    uint32_t r0 = 0;                  // phrase ends before the tile (expansion)
    bool plain = false;
};
__device__ __forceinline__ TileLoads tile_loads(const Ctx& c, uint64_t tile) {
    TileLoads L;
    const uint64_t b = tile, base = b * TILE;
    const bool maybe = c.dense_ok && c.T.is_packed() && base + TILE + 64 <= c.n;
    uint64_t f0 = 0, f1 = 0;
    if (maybe) {
        const uint64_t p0 = base + threadIdx.x * 16u;
        const uint64_t* w = c.T.packed + (p0 >> 5);
        L.w0 = w[0]; L.w1 = w[1];
        f0 = c.T.excw[b >> 6]; f1 = c.T.excw[(b + 1) >> 6];
    }
    if (c.repbits) L.r0 = c.coff ? c.brank[b] : c.rdir[b * 8];
    L.plain = maybe && !(((f0 >> (b & 63)) | (f1 >> ((b + 1) & 63))) & 1ull);
    return L;
}
// f(q, dense): the suffix at tile position 16 * threadIdx.x + q and the dense code of its first pc bases
template <typename F>
__device__ __forceinline__ void for_tile_dense(const Ctx& c, int pc, const TileLoads& L, F&& f) {
    const uint64_t win = (threadIdx.x & 1u) ? (L.w0 >> 32) | (L.w1 << 32) : L.w0;   // bases p0 .. p0 + 31, base k in bits 2k, 2k + 1
    uint64_t r = __brevll(win);                                                 // ... base k in bits 62 - 2k, 63 - 2k, halves swapped
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    const uint32_t dmask = (1u << (2 * pc)) - 1u;
#pragma unroll
    for (int q = 0; q < 16; q++) f(q, (uint32_t)(r >> (2 * (32 - q - pc))) & dmask);
}
// the number of dense codes (pc bases) whose bin lies below `bin`
__device__ __forceinline__ uint32_t dense_lower(const Ctx& c, int pc, uint32_t bin) {
    if (c.bits * pc < 32 && bin >= (1u << (c.bits * pc))) return 1u << (2 * pc);
    const uint32_t smask = (1u << c.bits) - 1u;
    uint32_t d = 0;
    for (int i = 0; i < pc; i++) {
        const uint32_t sym = (bin >> (c.bits * (pc - 1 - i))) & smask;
        uint32_t less = 0, eq = 0;
        for (int a = 0; a < 4; a++) { less += c.acgt[a] < sym ? 1u : 0u; eq |= c.acgt[a] == sym ? 1u : 0u; }
        d += less << (2 * (pc - 1 - i));
        if (!eq) break;
    }
    return d;
}
__device__ __forceinline__ uint32_t dense_to_bin(const Ctx& c, int pc, uint32_t dense) {
    uint32_t bin = 0;
    for (int i = 0; i < pc; i++) bin = (bin << c.bits) | ((c.acgt_lut >> (8 * ((dense >> (2 * (pc - 1 - i))) & 3u))) & 0xffu);
    return bin;
}
// the first c.chars symbol codes of the suffix at text position p (plain bases all of them), most significant first
__device__ __forceinline__ uint64_t key_from_packed(const Ctx& c, uint64_t p) {
    const uint64_t* w = c.T.packed + (p >> 5);
    const uint32_t sh = 2 * (uint32_t)(p & 31);
    const uint64_t x = sh ? (w[0] >> sh) | (w[1] << (64 - sh)) : w[0];           // 32 bases from p on
    uint64_t key = 0;
    for (int i = 0; i < c.chars; i++) key = (key << c.bits) | ((c.acgt_lut >> (8 * (uint32_t)((x >> (2 * i)) & 3u))) & 0xffu);
    return key;
}

// ---- slices of a bin of ONE repeated symbol (RunSlice, guided.cpp) ---------------------------------------------------------
// The suffixes that begin with c^pc -- assembly gaps: runs of N of tens of megabases in every haplotype -- are one bin whatever
// the number of leading characters, and on whole genomes a bin of billions of suffixes that no batch holds.  Their order is
// known in closed form: a suffix is c^r X with X0 != c (r = what is left of its run), and
//     c^r X < c^r' X'  <=>  (X0 < c, r) before (X0' < c, r' > r);  every X0 < c before every X0 > c;  (X0 > c: larger r first),
// so K = (X0 < c ? r : 2^41 - r) is monotone along the bin and a range of K is a contiguous piece of the suffix array.  K is
// cut into buckets (run_bucket: exact below 256, then 8 bits of mantissa per power of two -- 0.4 % steps) and a slice is a
// range of buckets [blo, bhi).  What is left of a run beyond the tile comes from a table with one entry per tile: the length of
// the run that begins at the tile's first position, and the symbol behind it (k_tile_lead + a chain on the host).
__device__ __forceinline__ uint32_t run_f(uint64_t r) {
    if (r < 256) return (uint32_t)r;
    const int e = 63 - __clzll((long long)r);
    return (uint32_t)(e - 7) * 256u + (uint32_t)((r >> (e - 8)) & 0xffu);
}
__device__ __forceinline__ uint32_t run_bucket(uint64_t r, bool below) {
    if (r >= (1ull << 40)) r = (1ull << 40) - 1;
    const uint32_t f = run_f(r);
    return below ? f : 2u * RUN_BUCKETS_HALF - 1u - f;
}
// bit q of the result: the suffix at tile position 16 * threadIdx.x + q, IF it begins with pc symbols rs.sym, lies in a bucket
// of [rs.blo, rs.bhi); *bucket_out (optional, 16 entries): its bucket.  Called by the whole workgroup with the tile staged.
__device__ __forceinline__ uint32_t tile_run_mask(const Ctx& c, uint64_t tile, const RunSlice& rs, const uint8_t* s_sym, uint32_t* bucket_out) {
    __shared__ unsigned long long s_rn[257];
    __shared__ uint8_t s_fn[257], s_d[256];
    const uint32_t t = threadIdx.x, t0 = t * 16;
    uint32_t d = 0;
    bool any = false;
#pragma unroll
    for (int q = 0; q < 16; q++) { const bool is = s_sym[t0 + q] == rs.sym; any |= is; if (is && d == (uint32_t)q) d++; }
    s_d[t] = (uint8_t)d;
    if (!__syncthreads_or(any ? 1 : 0)) return 0u;
    if (t == 0) {
        unsigned long long rn = 0; uint8_t fn = 0;
        if (tile + 1 < rs.n_tiles) {
            const uint8_t f1 = rs.first[tile + 1];
            if (f1 == rs.sym) { rn = rs.lead[tile + 1]; fn = rs.follow[tile + 1]; } else fn = f1;
        }
        s_rn[256] = rn; s_fn[256] = fn;
        for (int k = 255; k >= 0; k--) {
            const uint32_t dk = s_d[k];
            if (dk == 16) rn += 16; else { rn = dk; fn = s_sym[k * 16 + dk]; }
            s_rn[k] = rn; s_fn[k] = fn;
        }
    }
    __syncthreads();
    unsigned long long r = s_rn[t + 1];
    uint8_t fo = s_fn[t + 1];
    uint32_t mask = 0;
#pragma unroll
    for (int q = 15; q >= 0; q--) {
        const uint8_t sy = s_sym[t0 + q];
        if (sy == rs.sym) r++; else { r = 0; fo = sy; }
        const uint32_t b = r ? run_bucket(r, fo < rs.sym) : 0u;
        if (bucket_out) bucket_out[q] = b;
        if (r && b >= rs.blo && b < rs.bhi) mask |= 1u << q;
    }
    __syncthreads();
    return mask;
}

// per tile: the symbol at its first position, how many of its first positions hold that symbol (TILE: all of them), and the
// symbol behind them
__global__ __launch_bounds__(256) void k_tile_lead(Ctx c, uint8_t* __restrict__ first, uint16_t* __restrict__ lead, uint8_t* __restrict__ follow) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    __shared__ uint32_t s_min;
    if (threadIdx.x == 0) s_min = TILE;
    for_tile_bins(c, (uint64_t)blockIdx.x + c.tile0, 1, s_sym, [&](int, bool, uint32_t) {});
    const uint8_t s0 = s_sym[0];
    uint32_t mine = TILE;
    for (int q = 15; q >= 0; q--) if (s_sym[threadIdx.x * 16 + q] != s0) mine = threadIdx.x * 16 + q;
    if (mine < TILE) atomicMin(&s_min, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t tile = (uint64_t)blockIdx.x + c.tile0;
        first[tile] = s0; lead[tile] = (uint16_t)s_min; follow[tile] = s_min < TILE ? s_sym[s_min] : (uint8_t)0;
    }
}
void tile_lead(const Ctx& c, uint8_t* first, uint16_t* lead, uint8_t* follow, hipStream_t s) {
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_tile_lead, dim3(blocks), dim3(256), 0, s, cs, first, lead, follow);
    });
    MMT_HIP(hipGetLastError());
}

// ---- expansion (guided.cpp, Ctx::repbits): only the text suffixes that start in the REPRESENTATIVE occurrence of their
// distinct phrase are collected and sorted -- one per (distinct phrase, offset), i.e. one per valid suffix of the parse's
// dictionary (pfp_lcp_mum.hpp:131-147), which nobody has to hold --, and the emitter of the parse proper (pfp_kernels.hip
// k_emit) expands each of them by the inverted list of its phrase.
// Bit q of the result: the suffix at tile position 16 * threadIdx.x + q starts in a representative occurrence.  The tile's
// phrase ends (4096 positions + the w - 1 the query points reach beyond it) are staged in LDS as bits with their running
// counts: the phrase of a suffix is a rank, and its bit says "representative".
__device__ __forceinline__ uint32_t tile_rep_keep(const Ctx& c, uint64_t tile, uint32_t r0, const unsigned long long** cut_words = nullptr) {
    if (cut_words) *cut_words = nullptr;
    if (!c.repbits) return 0xffffu;
    __shared__ unsigned long long s_cut[66];
    if (cut_words) *cut_words = s_cut;       // (bit x of the 65 words: a phrase ends at tile position x; valid until the kernel's next call)
    __shared__ uint32_t s_pre[66];
    const uint64_t b = tile;                                       // the tile = block b of 4096 text positions
    if (c.coff) {
        if (threadIdx.x < 66) s_cut[threadIdx.x] = 0ull;
        __syncthreads();
        const uint32_t lo = c.brank[b], hi = c.brank[b + 1], hi2 = c.brank[b + 2];
        for (uint32_t e = lo + threadIdx.x; e < hi; e += 256) { const uint32_t o = c.coff[e]; atomicOr(&s_cut[o >> 6], 1ull << (o & 63)); }
        if (threadIdx.x < 64 && hi + threadIdx.x < hi2) {          // (ascending: only the first 64 entries can lie below 64)
            const uint32_t o = c.coff[hi + threadIdx.x];
            if (o < 64) atomicOr(&s_cut[64], 1ull << o);
        }
    } else if (threadIdx.x < 66) s_cut[threadIdx.x] = threadIdx.x < 65 ? c.mask[b * 64 + threadIdx.x] : 0ull;     // (whole blocks, zero padded, one block to spare)
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t v = (uint32_t)__popcll(s_cut[threadIdx.x]);
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if ((threadIdx.x & 63) >= (uint32_t)o) inc += y; }
        s_pre[threadIdx.x] = inc - v;
        if (threadIdx.x == 63) s_pre[64] = inc;
    }
    __syncthreads();
    const uint32_t t0 = threadIdx.x * 16;
    // The sixteen suffixes of a work-item lie in the phrase of the first one unless a phrase ends among their query points: one
    // rank for the first, then one step per phrase end (a phrase is tens to hundreds of characters: most work-items see none).
    // (sixteen ranks -- two LDS reads and a population count each -- were a third of a pass over the text)
    const uint32_t x0 = t0 + c.w - 1;                                  // query point of the first suffix, relative to the tile
    const uint32_t wi = x0 >> 6, sh = x0 & 63;
    uint32_t k = r0 + s_pre[wi] + (uint32_t)__popcll(s_cut[wi] & ((1ull << sh) - 1ull));
    // phrase ends at the query points of suffixes 0 .. 14 (an end AT the query point of suffix q puts suffix q + 1 in the next phrase)
    uint32_t cuts = (uint32_t)((s_cut[wi] >> sh) | (sh > 48 ? s_cut[wi + 1] << (64 - sh) : 0ull)) & 0x7fffu;
    uint32_t cached = k >> 5, word = c.repbits[cached];
    uint32_t keep = 0, from = 0;
    for (;;) {
        const uint32_t to = cuts ? (uint32_t)__builtin_ctz(cuts) + 1u : 16u;      // suffixes [from, to) start in phrase k
        if ((word >> (k & 31)) & 1u) keep |= ((1u << to) - 1u) & ~((1u << from) - 1u);
        if (!cuts) break;
        cuts &= cuts - 1u;
        from = to; k++;
        if ((k >> 5) != cached) { cached = k >> 5; word = c.repbits[cached]; }
    }
    __syncthreads();
    return keep;
}

// the element record of the suffix at tile position o of a plain tile of an expansion pass: the end of its phrase from the tile's
// staged phrase ends (tile_rep_keep's words) when it lies there -- nearly always -- instead of two or three more round trips
__device__ __forceinline__ uint64_t make_rec_tile(const Ctx& c, uint64_t base, uint32_t o, const unsigned long long* cut_words) {
    const uint64_t q = base + o + 1;
    if (c.rec_rank || c.skip || !cut_words) return make_rec(c, q);
    const uint32_t xo = o + c.w - 1;                              // query point (text position q + w - 2), relative to the tile
    uint32_t wi = xo >> 6;
    unsigned long long word = wi < 65u ? cut_words[wi] & (~0ull << (xo & 63)) : 0ull;
    while (!word && ++wi < 65u) word = cut_words[wi];
    if (!word) return make_rec(c, q);
    const uint64_t len = base + (uint64_t)wi * 64 + (uint64_t)__builtin_ctzll(word) + 2 - q;
    return q | ((len < LEN_SAT ? len : (uint64_t)LEN_SAT) << 40);
}

// One pass whatever the number of bins: a bin whose leading characters are all of A C G T -- nearly every suffix -- is
// counted in LDS under its dense code (two bits a character: 4^pc <= 1024 counters), the others ('$', N, IUPAC codes, the
// Dollars at the end) straight in global memory.  (A tile used to count in 4096 LDS counters, one pass over the text per
// value of the bin code's bits above the low twelve: five leading characters at three bits each -- texts beyond 2^38
// characters -- were EIGHT passes rolling full 63-bit keys, 13 s of a rank's share of 573 G characters.)
// (a workgroup counts BIN_HIST_TILES consecutive tiles before its LDS counters leave for global memory: with five leading
// characters nearly all of the 1024 counters are hit in every tile, and 140 M tiles of a 573 G-character text each sending 1024
// atomics to the same 1024 words were 6 s per pass -- eight times a pass that only reads the text)
constexpr uint32_t BIN_HIST_TILES = 32;
__global__ __launch_bounds__(256) void k_bin_hist(Ctx c0, int pc, uint64_t* __restrict__ hist, uint32_t slice_tiles) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    __shared__ uint32_t s_hist[1024];
    __shared__ uint8_t s_acgt[256];                 // symbol code -> 0 .. 3, or 0xff
    __shared__ uint32_t s_sym_of[4];
    for (int i = threadIdx.x; i < 1024; i += 256) s_hist[i] = 0;
    s_acgt[threadIdx.x] = 0xff;
    __syncthreads();
    if (threadIdx.x < 4) { const uint8_t sym = c0.code[(uint8_t)"ACGT"[threadIdx.x]]; s_acgt[sym] = (uint8_t)threadIdx.x; s_sym_of[threadIdx.x] = sym; }
    __syncthreads();
    const uint32_t smask = (1u << c0.bits) - 1u;
    // (symbol code -> dense digit from a table in a register, four bits a symbol, 15 = not one of A C G T: five LDS lookups
    // per position made this pass 6 s over 573 G characters)
    uint64_t lut = ~0ull;
    const bool in_register = c0.bits <= 4;
    if (in_register)
        for (uint32_t k = 0; k < 4; k++) { const uint32_t sym = s_sym_of[k]; lut = (lut & ~(0xfull << (4 * sym))) | ((uint64_t)k << (4 * sym)); }
    for (uint32_t ti = 0; ti < BIN_HIST_TILES; ti++) {
    const uint32_t tile_in_slice = blockIdx.x * BIN_HIST_TILES + ti;
    if (tile_in_slice >= slice_tiles) break;
    const Ctx& c = c0;
    const uint64_t tile = (uint64_t)c0.tile0 + tile_in_slice;
    const TileLoads TL = tile_loads(c, tile);
    const uint32_t keep = tile_rep_keep(c, tile, TL.r0);
    // (the other bins: a work-item adds up consecutive suffixes of the same bin -- inside an assembly gap all sixteen are N^pc --
    // and the wave adds up its work-items' counts before ONE atomic leaves for global memory: 1.5 G suffixes of the gaps of 13 whole
    // genomes on one address were 6 s per pass at the ~250 atomics per microsecond one word takes)
    uint32_t pend_bin = 0xffffffffu, pend_cnt = 0;
    auto flush_wave = [&]() {
        unsigned long long todo = __ballot(pend_cnt != 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t b = (uint32_t)__shfl((int)pend_bin, leader, 64);
            const bool same = pend_cnt != 0 && pend_bin == b;
            const unsigned long long grp = __ballot(same);
            uint32_t v = same ? pend_cnt : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(reinterpret_cast<unsigned long long*>(hist + b), (unsigned long long)v);
            if (same) pend_cnt = 0;
            todo &= ~grp;
        }
    };
    if (TL.plain) {
        for_tile_dense(c, pc, TL, [&](int q, uint32_t dense) { if ((keep >> q) & 1u) atomicAdd(&s_hist[dense], 1u); });
        __syncthreads();
        continue;
    }
    for_tile_bins(c, tile, pc, s_sym, [&](int q, bool in, uint32_t bin) {
        if (!in || !((keep >> q) & 1u)) return;
        uint32_t dense = 0, bad = 0;
        for (int ch = 0; ch < pc; ch++) {
            const uint32_t sym = (bin >> (c.bits * (pc - 1 - ch))) & smask;
            const uint32_t a = in_register ? (uint32_t)(lut >> (4 * sym)) & 15u : (s_acgt[sym] == 0xffu ? 15u : (uint32_t)s_acgt[sym]);
            bad |= a >> 2;
            dense = (dense << 2) | (a & 3u);
        }
        if (!bad) atomicAdd(&s_hist[dense], 1u);
        else if (bin == pend_bin) pend_cnt++;
        else {
            if (pend_cnt) atomicAdd(reinterpret_cast<unsigned long long*>(hist + pend_bin), (unsigned long long)pend_cnt);
            pend_bin = bin; pend_cnt = 1;
        }
    });
    flush_wave();
    __syncthreads();
    }
    const Ctx& c = c0;
    const uint32_t n_dense = 1u << (2 * pc);
    for (uint32_t d = threadIdx.x; d < n_dense; d += 256) {
        if (!s_hist[d]) continue;
        uint32_t bin = 0;
        for (int ch = 0; ch < pc; ch++) bin = (bin << c.bits) | s_sym_of[(d >> (2 * (pc - 1 - ch))) & 3u];
        atomicAdd(reinterpret_cast<unsigned long long*>(hist + bin), (unsigned long long)s_hist[d]);
    }
}
void bin_hist(const Ctx& c, int prefix_chars, uint64_t* hist, hipStream_t s) {
    if (prefix_chars > 5) throw std::runtime_error("bin_hist: at most five leading characters");
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_bin_hist, dim3((blocks + BIN_HIST_TILES - 1) / BIN_HIST_TILES), dim3(256), 0, s, cs, prefix_chars, hist, blocks);
    });
    MMT_HIP(hipGetLastError());
}

// (the text-order kernels below take PASS_TILES consecutive tiles per workgroup, like k_bin_hist: a workgroup per tile -- 140 M of
// them per pass over 573 G characters, each fetching its kernel arguments and living six microseconds -- ran at 3 - 6 ns a tile
// where the histogram's loop runs at 1.8)
constexpr uint32_t PASS_TILES = 1;
#define MMT_PASS_LOOP(c, slice_tiles, tile)                                                                 \
    for (uint32_t ti_ = 0; ti_ < PASS_TILES; ti_++)                                                         \
        if (const uint32_t tile_in_slice_ = blockIdx.x * PASS_TILES + ti_; tile_in_slice_ < (slice_tiles))  \
            if (const uint64_t tile = (uint64_t)(c).tile0 + tile_in_slice_; true)
__device__ __forceinline__ void batch_count_tile(const Ctx& c, uint64_t tile, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                 uint32_t* __restrict__ tile_count, const RunSlice& rs) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    uint32_t mine = 0, slice = 0xffffu;
    const TileLoads TL = tile_loads(c, tile);
    const uint32_t keep = tile_rep_keep(c, tile, TL.r0);
    if (!rs.sym && TL.plain) {
        const uint32_t dlo = dense_lower(c, pc, bin_lo), dhi = dense_lower(c, pc, bin_hi);
        for_tile_dense(c, pc, TL, [&](int q, uint32_t d) { if (((keep >> q) & 1u) && d >= dlo && d < dhi) mine++; });
    } else
    for_tile_bins(c, tile, pc, s_sym, [&](int q, bool in, uint32_t b) { if (in && ((keep & slice) >> q & 1u) && b >= bin_lo && b < bin_hi) mine++; },
                  [&]() { if (rs.sym) slice = tile_run_mask(c, tile, rs, s_sym, nullptr); });
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[tile] = s_cnt;
}
__global__ __launch_bounds__(256) void k_batch_count(Ctx c, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                     uint32_t* __restrict__ tile_count, RunSlice rs, uint32_t slice_tiles) {
    MMT_PASS_LOOP(c, slice_tiles, tile) { batch_count_tile(c, tile, pc, bin_lo, bin_hi, tile_count, rs); __syncthreads(); }
}
void batch_count(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, uint32_t* tile_count, hipStream_t s, const RunSlice& rs) {
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_batch_count, dim3((blocks + PASS_TILES - 1) / PASS_TILES), dim3(256), 0, s, cs, prefix_chars, bin_lo, bin_hi,
                           tile_count, rs, blocks);
    });
    MMT_HIP(hipGetLastError());
}

// the suffixes of the run bin `bin` (= rs.sym repeated pc times) per bucket of K: hist[b] all of them, hist[2 * HALF + b] those the
// expansion keeps (c.repbits; the same numbers without it).  One pass; only tiles that hold the symbol do anything.
__global__ __launch_bounds__(256) void k_run_hist(Ctx c, int pc, uint32_t bin, RunSlice rs, unsigned long long* __restrict__ hist) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    uint32_t bucket[16];
    uint32_t inrun = 0;
    const uint64_t tile = (uint64_t)blockIdx.x + c.tile0;
    const uint32_t keep = tile_rep_keep(c, tile, c.repbits ? (c.coff ? c.brank[tile] : c.rdir[tile * 8]) : 0u);
    for_tile_bins(c, tile, pc, s_sym, [&](int q, bool in, uint32_t b) {
        if (!in || b != bin || !((inrun >> q) & 1u)) return;
        atomicAdd(hist + bucket[q], 1ull);
        if ((keep >> q) & 1u) atomicAdd(hist + 2u * RUN_BUCKETS_HALF + bucket[q], 1ull);
    }, [&]() { inrun = tile_run_mask(c, tile, rs, s_sym, bucket); });
}
void run_hist(const Ctx& c, int prefix_chars, uint32_t bin, const RunSlice& rs, uint64_t* hist, hipStream_t s) {
    RunSlice all = rs;
    all.blo = 0; all.bhi = 2u * RUN_BUCKETS_HALF;
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_run_hist, dim3(blocks), dim3(256), 0, s, cs, prefix_chars, bin, all, reinterpret_cast<unsigned long long*>(hist));
    });
    MMT_HIP(hipGetLastError());
}

// (next_count, optional: the same pass counts, per tile, the suffixes of the NEXT batch's bins [next_lo, next_hi): that batch
// then needs no k_batch_count of its own -- one pass over the text per batch instead of two)
__device__ __forceinline__ void batch_fill_tile(const Ctx& c, uint64_t tile, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                const uint32_t* __restrict__ tile_off, uint64_t* __restrict__ keys,
                                                uint64_t* __restrict__ pos, uint32_t next_lo, uint32_t next_hi,
                                                uint32_t* __restrict__ next_count, const RunSlice& rs) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    __shared__ uint32_t s_wave[4], s_next[4];
    __shared__ uint16_t s_sel[TILE];                               // tile offsets of the selected suffixes, in order
    constexpr int PER = TILE / 256;
    uint32_t sel = 0, nxt = 0, slice = 0xffffu;
    const TileLoads TL = tile_loads(c, tile);
    const uint64_t first = tile_off[tile];
    const unsigned long long* cut_words = nullptr;
    const uint32_t keep = tile_rep_keep(c, tile, TL.r0, &cut_words);
    const bool plain = !rs.sym && TL.plain;
    if (plain) {
        const uint32_t dlo = dense_lower(c, pc, bin_lo), dhi = dense_lower(c, pc, bin_hi);
        const uint32_t nlo = dense_lower(c, pc, next_lo), nhi = dense_lower(c, pc, next_hi);
        for_tile_dense(c, pc, TL, [&](int q, uint32_t d) {
            if (!((keep >> q) & 1u)) return;
            if (d >= dlo && d < dhi) sel |= 1u << q;
            if (d >= nlo && d < nhi) nxt++;
        });
    } else
    // (a batch of slices of a run bin never counts the next batch along: next_count is null then)
    for_tile_bins(c, tile, pc, s_sym, [&](int q, bool in, uint32_t b) {
        in = in && ((keep >> q) & 1u);
        if (in && b >= bin_lo && b < bin_hi && ((slice >> q) & 1u)) sel |= 1u << q;
        if (in && b >= next_lo && b < next_hi) nxt++;
    }, [&]() { if (rs.sym) slice = tile_run_mask(c, tile, rs, s_sym, nullptr); });
    // ordered compaction: exclusive prefix of the per-work-item counts over the workgroup
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = __popc(sel);
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nxt += __shfl_xor(nxt, o, 64);
    if (lane == 63) s_wave[wave] = inc;
    if (lane == 0) s_next[wave] = nxt;
    __syncthreads();
    if (next_count && threadIdx.x == 0) next_count[tile] = s_next[0] + s_next[1] + s_next[2] + s_next[3];
    uint32_t at = inc - cnt;
    uint32_t total = 0;
    for (uint32_t wv = 0; wv < 4; wv++) { if (wv < wave) at += s_wave[wv]; total += s_wave[wv]; }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        if (!(sel & (1u << q))) continue;
        s_sel[at++] = (uint16_t)(threadIdx.x * PER + q);
    }
    __syncthreads();
    // keys and records of the selected suffixes only, one per work-item at a time: the first `chars` symbol codes from the
    // staged tile; phrase-end / parse-rank lookups (inside the loop above each lookup would wait for the one before it:
    // sixteen latencies in a row per wave)
    const uint64_t base = tile * TILE;
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint32_t o = s_sel[i];
        uint64_t key = 0;
        if (plain) key = key_from_packed(c, base + o);
        else for (int ch = 0; ch < c.chars; ch++) key = (key << c.bits) | s_sym[o + ch];
        keys[first + i] = key;
        pos[first + i] = plain ? make_rec_tile(c, base, o, cut_words) : make_rec(c, base + o + 1);
    }
}
__global__ __launch_bounds__(256) void k_batch_fill(Ctx c, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                    const uint32_t* __restrict__ tile_off, uint64_t* __restrict__ keys,
                                                    uint64_t* __restrict__ pos, uint32_t next_lo, uint32_t next_hi,
                                                    uint32_t* __restrict__ next_count, RunSlice rs, uint32_t slice_tiles) {
    MMT_PASS_LOOP(c, slice_tiles, tile) {
        batch_fill_tile(c, tile, pc, bin_lo, bin_hi, tile_off, keys, pos, next_lo, next_hi, next_count, rs);
        __syncthreads();
    }
}
void batch_fill(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, const uint32_t* tile_off, uint64_t* keys,
                uint64_t* pos, uint32_t next_lo, uint32_t next_hi, uint32_t* next_count, hipStream_t s, const RunSlice& rs) {
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_batch_fill, dim3((blocks + PASS_TILES - 1) / PASS_TILES), dim3(256), 0, s, cs, prefix_chars, bin_lo, bin_hi,
                           tile_off, keys, pos, next_lo, next_hi, next_count, rs, blocks);
    });
    MMT_HIP(hipGetLastError());
}

// ---- several batches per pass over the text ------------------------------------------------------------------------------------
// A batch is the suffixes of a few bins, and finding them is a pass over the WHOLE text: 127 batches of a rank's share of
// 573 G characters were 127 x 2 passes, 126 of the 186 s the device was busy (k_batch_count + k_batch_fill + the histogram).
// One pass now serves as many batches as a staging list holds: the suffixes of bins [bin_lo, bin_hi) -- several batches --
// are written in text order as (V index | bin << 40), eight bytes each, and every batch then takes its own from that list (a
// pass over a few G entries instead of hundreds of G characters) and only there looks up keys and records.
// (the same pass also counts, per tile, the suffixes of the NEXT pass's bins [next_lo, next_hi): its k_batch_count is then not run)
__device__ __forceinline__ void stage_fill_tile(const Ctx& c, uint64_t tile, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ staged,
                                                uint32_t next_lo, uint32_t next_hi, uint32_t* __restrict__ next_count) {
    __shared__ __align__(16) uint8_t s_sym[TILE + 64];
    __shared__ uint32_t s_wave[4], s_next[4];
    __shared__ uint16_t s_sel[TILE];
    constexpr int PER = TILE / 256;
    uint32_t sel = 0, nxt = 0;
    const TileLoads TL = tile_loads(c, tile);
    const uint64_t first = tile_off[tile];
    const uint32_t keep = tile_rep_keep(c, tile, TL.r0);       // (expansion: the list holds representatives only)
    const bool plain = TL.plain;
    if (plain) {
        const uint32_t dlo = dense_lower(c, pc, bin_lo), dhi = dense_lower(c, pc, bin_hi);
        const uint32_t nlo = dense_lower(c, pc, next_lo), nhi = dense_lower(c, pc, next_hi);
        for_tile_dense(c, pc, TL, [&](int q, uint32_t d) {
            if (!((keep >> q) & 1u)) return;
            if (d >= dlo && d < dhi) sel |= 1u << q;
            if (d >= nlo && d < nhi) nxt++;
        });
    } else
    for_tile_bins(c, tile, pc, s_sym, [&](int q, bool in, uint32_t b) {
        in = in && ((keep >> q) & 1u);
        if (in && b >= bin_lo && b < bin_hi) sel |= 1u << q;
        if (in && b >= next_lo && b < next_hi) nxt++;
    });
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = __popc(sel);
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nxt += __shfl_xor(nxt, o, 64);
    if (lane == 63) s_wave[wave] = inc;
    if (lane == 0) s_next[wave] = nxt;
    __syncthreads();
    if (next_count && threadIdx.x == 0) next_count[tile] = s_next[0] + s_next[1] + s_next[2] + s_next[3];
    uint32_t at = inc - cnt, total = 0;
    for (uint32_t wv = 0; wv < 4; wv++) { if (wv < wave) at += s_wave[wv]; total += s_wave[wv]; }
    const uint64_t base = tile * TILE;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        if (!(sel & (1u << q))) continue;
        s_sel[at++] = (uint16_t)(threadIdx.x * PER + q);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint32_t o = s_sel[i];
        uint32_t bin = 0;
        if (plain) {
            // (the bin of a selected suffix from the words of the text again: they are in the cache, the selected are few)
            const uint64_t p = base + o;
            const uint64_t* w = c.T.packed + (p >> 5);
            const uint32_t sh = 2 * (uint32_t)(p & 31);
            const uint64_t x = sh ? (w[0] >> sh) | (w[1] << (64 - sh)) : w[0];
            for (int ch = 0; ch < pc; ch++) bin = (bin << c.bits) | ((c.acgt_lut >> (8 * (uint32_t)((x >> (2 * ch)) & 3u))) & 0xffu);
        } else
        for (int ch = 0; ch < pc; ch++) bin = (bin << c.bits) | s_sym[o + ch];
        staged[first + i] = o | (bin << 12);             // (the tile is where the entry lies in the list: tile_off)
    }
}
__global__ __launch_bounds__(256) void k_stage_fill(Ctx c, int pc, uint32_t bin_lo, uint32_t bin_hi,
                                                    const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ staged,
                                                    uint32_t next_lo, uint32_t next_hi, uint32_t* __restrict__ next_count,
                                                    uint32_t slice_tiles) {
    MMT_PASS_LOOP(c, slice_tiles, tile) {
        stage_fill_tile(c, tile, pc, bin_lo, bin_hi, tile_off, staged, next_lo, next_hi, next_count);
        __syncthreads();
    }
}
void stage_fill(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, const uint32_t* tile_off, uint32_t* staged,
                uint32_t next_lo, uint32_t next_hi, uint32_t* next_count, hipStream_t s) {
    for_tile_slices(c, [&](const Ctx& cs, unsigned blocks) {
        hipLaunchKernelGGL(k_stage_fill, dim3((blocks + PASS_TILES - 1) / PASS_TILES), dim3(256), 0, s, cs, prefix_chars, bin_lo, bin_hi,
                           tile_off, staged, next_lo, next_hi, next_count, blocks);
    });
    MMT_HIP(hipGetLastError());
}
// the last tile t of [lo, hi] with tile_off[t] <= idx: the tile whose entries hold entry idx of the list (empty tiles share their
// offset with the tile behind them: the last one owns the entries)
__device__ __forceinline__ uint32_t tile_of_entry(const uint32_t* __restrict__ tile_off, uint32_t lo, uint32_t hi, uint32_t idx) {
    while (lo < hi) { const uint32_t mid = lo + (hi - lo + 1) / 2; if (tile_off[mid] <= idx) lo = mid; else hi = mid - 1; }
    return lo;
}
// blk_tile[B] = the tile of entry 4096 B (B < blocks), blk_tile[blocks] = the last tile: a block of the list knows its tiles' range
__global__ void k_stage_block_tiles(const uint32_t* __restrict__ tile_off, uint32_t n_tiles, uint64_t n, uint32_t blocks,
                                    uint32_t* __restrict__ blk_tile) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > blocks) return;
    blk_tile[b] = b == blocks || (uint64_t)b * 4096 >= n ? n_tiles - 1 : tile_of_entry(tile_off, 0, n_tiles - 1, b * 4096u);
}
void stage_block_tiles(const uint32_t* tile_off, uint32_t n_tiles, uint64_t n, uint32_t* blk_tile, hipStream_t s) {
    if (!n) return;
    const uint32_t blocks = (uint32_t)((n + 4095) / 4096);
    hipLaunchKernelGGL(k_stage_block_tiles, dim3(grid_for((uint64_t)blocks + 1, 256)), dim3(256), 0, s, tile_off, n_tiles, n, blocks, blk_tile);
    MMT_HIP(hipGetLastError());
}
// blocks of 4096 staged entries, 16 consecutive ones per work-item
__global__ __launch_bounds__(256) void k_stage_count(const uint32_t* __restrict__ staged, uint64_t n, uint32_t b0, uint32_t b1,
                                                     uint32_t* __restrict__ block_count) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint64_t from = (uint64_t)blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (from + q < n) { const uint32_t b = staged[from + q] >> 12; mine += (b >= b0 && b < b1) ? 1u : 0u; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = s_cnt;
}
// (next_count, optional: the same pass counts, per block, the entries of the NEXT batch's bins [nb0, nb1))
// An entry is four bytes -- its offset inside its tile and its bin -- since round 6 (eight before: V index and bin): the same
// memory lists twice the suffixes, a share of 573 G characters takes ten passes over the text instead of nineteen.  The tile of an
// entry is where the entry lies in the list (tile_off, the prefix sums the pass filled it by); a block of the list knows the range
// of its tiles (blk_tile) and only the entries a batch TAKES look theirs up, in the block's staged offsets.
constexpr uint32_t TAKE_SPAN = 2048;
__global__ __launch_bounds__(256) void k_stage_take(Ctx c, const uint32_t* __restrict__ staged, uint64_t n, uint32_t b0, uint32_t b1,
                                                    const uint32_t* __restrict__ block_off, uint64_t* __restrict__ keys,
                                                    uint64_t* __restrict__ pos, uint32_t nb0, uint32_t nb1,
                                                    uint32_t* __restrict__ next_count, const uint32_t* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ blk_tile) {
    __shared__ uint8_t s_code[256];
    __shared__ uint32_t s_wave[4], s_next[4];
    __shared__ uint32_t s_sel[4096];
    __shared__ uint16_t s_idx[4096];
    __shared__ uint32_t s_off[TAKE_SPAN];
    for (int i = threadIdx.x; i < 256; i += 256) s_code[i] = c.code[i];
    const uint32_t t_lo = blk_tile[blockIdx.x], t_hi = blk_tile[blockIdx.x + 1];
    const bool in_lds = t_hi - t_lo < TAKE_SPAN;
    if (in_lds) for (uint32_t j = threadIdx.x; j <= t_hi - t_lo; j += 256) s_off[j] = tile_off[t_lo + j];
    const uint64_t from = (uint64_t)blockIdx.x * 4096 + threadIdx.x * 16;
    uint32_t mine[16];
    uint32_t sel = 0, nxt = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        mine[q] = from + q < n ? staged[from + q] : ~0u;
        const uint32_t b = mine[q] >> 12;
        if (from + q < n && b >= b0 && b < b1) sel |= 1u << q;
        if (from + q < n && b >= nb0 && b < nb1) nxt++;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cnt = __popc(sel);
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += y; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nxt += __shfl_xor(nxt, o, 64);
    if (lane == 63) s_wave[wave] = inc;
    if (lane == 0) s_next[wave] = nxt;
    __syncthreads();
    uint32_t at = inc - cnt, total = 0;
    for (uint32_t wv = 0; wv < 4; wv++) { if (wv < wave) at += s_wave[wv]; total += s_wave[wv]; }
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (sel & (1u << q)) { s_sel[at] = mine[q]; s_idx[at] = (uint16_t)(threadIdx.x * 16 + q); at++; }
    __syncthreads();
    const uint64_t first = block_off[blockIdx.x];
    if (next_count && threadIdx.x == 0) next_count[blockIdx.x] = s_next[0] + s_next[1] + s_next[2] + s_next[3];
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint32_t idx = blockIdx.x * 4096u + s_idx[i];
        uint32_t tile;
        if (in_lds) {
            uint32_t lo = 0, hi = t_hi - t_lo;
            while (lo < hi) { const uint32_t mid = lo + (hi - lo + 1) / 2; if (s_off[mid] <= idx) lo = mid; else hi = mid - 1; }
            tile = t_lo + lo;
        } else tile = tile_of_entry(tile_off, t_lo, t_hi, idx);
        const uint64_t q = (uint64_t)tile * TILE + (s_sel[i] & 4095u) + 1;
        keys[first + i] = pack_chars(c, s_code, q);
        pos[first + i] = make_rec(c, q);
    }
}
void stage_count(const uint32_t* staged, uint64_t n, uint32_t b0, uint32_t b1, uint32_t* block_count, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_stage_count, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, staged, n, b0, b1, block_count);
    MMT_HIP(hipGetLastError());
}
void stage_take(const Ctx& c, const uint32_t* staged, uint64_t n, uint32_t b0, uint32_t b1, const uint32_t* block_off, uint64_t* keys,
                uint64_t* pos, uint32_t nb0, uint32_t nb1, uint32_t* next_count, const uint32_t* tile_off, const uint32_t* blk_tile,
                hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_stage_take, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, c, staged, n, b0, b1, block_off, keys, pos, nb0,
                       nb1, next_count, tile_off, blk_tile);
    MMT_HIP(hipGetLastError());
}

template <typename P>
__global__ void k_phrase_items(Ctx c, const P* __restrict__ pstart, const uint32_t* __restrict__ rep, uint32_t D,
                               uint64_t* __restrict__ keys, uint64_t* __restrict__ pos) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = c.code[i];
    __syncthreads();
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint64_t q = (uint64_t)pstart[rep[d]];
    keys[d] = pack_chars(c, s_code, q);
    pos[d] = make_rec(c, q);
}
void phrase_items(const Ctx& c, const void* pstart, bool wide, const uint32_t* rep, uint32_t D, uint64_t* keys, uint64_t* pos,
                  hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_phrase_items<uint64_t>, dim3(grid_for(D, 256)), dim3(256), 0, s, c,
                           static_cast<const uint64_t*>(pstart), rep, D, keys, pos);
    else
        hipLaunchKernelGGL(k_phrase_items<uint32_t>, dim3(grid_for(D, 256)), dim3(256), 0, s, c,
                           static_cast<const uint32_t*>(pstart), rep, D, keys, pos);
    MMT_HIP(hipGetLastError());
}

// ---- refinement rounds -------------------------------------------------------------------------------------------
// lcp (optional, one entry per slot): where two neighbours differ in their first keys the number of leading symbols the
// keys share IS their LCP (symbol codes are injective; `bits` per symbol, `chars` symbols per key)
__global__ void k_heads0(const uint64_t* __restrict__ keys, uint32_t B, uint8_t* __restrict__ head,
                         uint8_t* __restrict__ active, uint32_t* __restrict__ lcp, int bits, int chars) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const uint64_t k = keys[j];
    const uint64_t kp = j ? keys[j - 1] : 0ull;
    const bool h = j == 0 || k != kp;
    const bool next_h = j + 1 == B || keys[j + 1] != k;
    head[j] = h ? 1 : 0;
    active[j] = (h && next_h) ? 0 : 1;
    if (lcp && j && h) lcp[j] = (uint32_t)((__builtin_clzll(k ^ kp) - (64 - bits * chars)) / bits);
}
void heads0(const uint64_t* keys, uint32_t B, uint8_t* head, uint8_t* active, uint32_t* lcp, int bits, int chars, hipStream_t s) {
    hipLaunchKernelGGL(k_heads0, dim3(grid_for(B, 256)), dim3(256), 0, s, keys, B, head, active, lcp, bits, chars);
    MMT_HIP(hipGetLastError());
}
__global__ void k_gather_active(const uint32_t* __restrict__ idx, uint32_t m, const uint64_t* __restrict__ pos_sorted,
                                const uint8_t* __restrict__ head, uint32_t* __restrict__ slot, uint64_t* __restrict__ pos,
                                uint32_t* __restrict__ headval) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    const uint32_t j = idx[c];
    slot[c] = j; pos[c] = pos_sorted[j];
    headval[c] = head[j] ? c : 0u;
}
void gather_active(const uint32_t* idx, uint32_t m, const uint64_t* pos_sorted, const uint8_t* head, uint32_t* slot,
                   uint64_t* pos, uint32_t* headval, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_active, dim3(grid_for(m, 256)), dim3(256), 0, s, idx, m, pos_sorted, head, slot, pos, headval);
    MMT_HIP(hipGetLastError());
}

// Early use of the giant dictionary (expansion only): a group ALL of whose members lie in giant phrases -- the suffixes inside the
// assembly gaps of whole genomes: 1.5 G of the 16.6 G representatives of a rank's share of 13 realistic genomes, one group of N^21
// after the first sort -- need not walk to g_depth 21 characters a round (30 rounds of 1.5 G elements each): the giant dictionary
// orders them at any depth.  k_giant_probe looks every member up (gr[e] = its entry at `offset`, or none) and clears the flag of a
// group that holds a member outside the giant phrases (or a spent one); k_round_keys then gives the members of the other groups
// their entries as keys.  (Only when the elements are representatives: a tie among them is final in any order, so a group that
// took its keys from the giant dictionary is done -- with several occurrences per phrase suffix the ties would go on to the
// parse ranks, which this round's offset does not reach.)
__global__ void k_giant_probe(Ctx c, const uint64_t* __restrict__ pos, const uint32_t* __restrict__ ghead, uint32_t m, uint64_t offset,
                              uint32_t* __restrict__ gr, uint8_t* __restrict__ pure) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const uint64_t rec = pos[e], q = rec_pos(c, rec);
    uint32_t r = 0xffffffffu;
    const uint64_t len = rec_len(c, rec, q);
    if (offset >= len || !giant_entry(c, q, offset, r)) r = 0xffffffffu;
    gr[e] = r;
    if (r == 0xffffffffu) pure[ghead[e]] = 0;
}
void giant_probe(const Ctx& c, const uint64_t* pos, const uint32_t* ghead, uint32_t m, uint64_t offset, uint32_t* gr, uint8_t* pure,
                 hipStream_t s) {
    MMT_HIP(hipMemsetAsync(pure, 1, m, s));
    hipLaunchKernelGGL(k_giant_probe, dim3(grid_for(m, 256)), dim3(256), 0, s, c, pos, ghead, m, offset, gr, pure);
    MMT_HIP(hipGetLastError());
}

// key of an element whose first `offset` characters are known to be shared by its whole group
// (gr / pure / ghead, optional: the entries and group flags of k_giant_probe)
__global__ void k_round_keys(Ctx c, const uint64_t* __restrict__ pos, uint32_t m, uint64_t offset,
                             uint64_t* __restrict__ keys, uint32_t* __restrict__ err, const uint32_t* __restrict__ gr,
                             const uint8_t* __restrict__ pure, const uint32_t* __restrict__ ghead) {
    __shared__ uint8_t s_code[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_code[i] = c.code[i];
    __syncthreads();
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    if (gr && pure[ghead[e]]) { keys[e] = GIANT_KEY | gr[e]; return; }
    const uint64_t rec = pos[e], q = rec_pos(c, rec);
    const uint64_t len = rec_len(c, rec, q);
    if (offset >= len) {
        // alpha is spent: the group shares it (phrase suffixes are prefix-free) and the following parse suffixes decide
        if (c.skip) { atomicAdd(err, 1u); keys[e] = RANK_KEY | e; return; }     // two equal distinct phrases
        keys[e] = RANK_KEY | rec_rank_key(c, rec, q);
        return;
    }
    if (c.g_n && offset >= c.g_depth) {
        // the group has shared g_depth characters and its alphas go on: giant phrases -- the rest by the giant dictionary
        uint32_t r = 0;
        // (the key is the ENTRY, not its group: entries order like groups, and k_round_heads reads the LCP of two of them off the
        // giant dictionary's LCP array instead of leaving the pair to k_batch_lcp's walk through the text)
        if (giant_entry(c, q, offset, r)) { keys[e] = GIANT_KEY | r; return; }
    }
    keys[e] = pack_chars(c, s_code, q + offset);
}
void round_keys(const Ctx& c, const uint64_t* pos, uint32_t m, uint64_t offset, uint64_t* keys, uint32_t* err, hipStream_t s,
                const uint32_t* gr, const uint8_t* pure, const uint32_t* ghead) {
    hipLaunchKernelGGL(k_round_keys, dim3(grid_for(m, 256)), dim3(256), 0, s, c, pos, m, offset, keys, err, gr, pure, ghead);
    MMT_HIP(hipGetLastError());
}

// Groups of at most SMALL elements are finished in one step: every element counts the members of its group that sort
// before it, comparing the rest of alpha 8 bytes at a time and then the parse ranks.  (Two haplotypes that agree for
// thousands of characters are such a pair for every one of their suffixes; the refinement rounds would walk their
// phrase 63 bits per round.)  flags[c] = 1 for the members of larger groups, which go through the rounds.
constexpr uint32_t SMALL = 8;
__global__ void k_resolve_small(Ctx c, const uint64_t* __restrict__ pos, const uint32_t* __restrict__ ghead,
                                const uint32_t* __restrict__ slot, uint32_t m, uint64_t offset,
                                uint64_t* __restrict__ out, uint8_t* __restrict__ flags, uint32_t* __restrict__ err,
                                uint32_t* __restrict__ lcp_out, uint32_t limit) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const uint32_t g0 = ghead[e];
    uint32_t end = e + 1;
    while (end < m && end - g0 <= limit && ghead[end] == g0) end++;
    if (end - g0 > limit) { flags[e] = 1; return; }
    const uint64_t rec = pos[e], q = rec_pos(c, rec);
    if (end - g0 == 2 && c.rec_rank) {
        // a pair (two haplotypes): every load of the common case is issued before anything depends on one -- both
        // phrase ends, 32 characters of both suffixes -- because random lines over tens of GB are latency, not bandwidth
        const uint32_t j = e == g0 ? g0 + 1 : g0;
        const uint64_t rj = pos[j], qj = rec_pos(c, rj);
        const uint64_t xe = query_point(c, q), xj = query_point(c, qj);
        const uint64_t me = xe < c.n && c.mask ? c.mask[xe >> 6] : 0, mj = xj < c.n && c.mask ? c.mask[xj >> 6] : 0;
        uint64_t a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { a[t] = tx_load8(c.T, q + offset + 8 * t); b[t] = tx_load8(c.T, qj + offset + 8 * t); }
        const uint64_t len = (xe < c.n ? next_cut_from(c, xe, me) : c.n + c.w - 1) + 2 - q;
        const uint64_t lj = (xj < c.n ? next_cut_from(c, xj, mj) : c.n + c.w - 1) + 2 - qj;
        const uint64_t L = len < lj ? len : lj;
        int cmp = 0;
        bool open = true;                                        // no difference found yet and alpha not yet spent
        // (share = what the two suffixes share, known here where a character decides: the second of the pair leaves it as its
        // LCP -- k_batch_lcp would walk the same characters again from the start of the suffixes)
        uint64_t share = ~0ull;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint64_t at = offset + 8 * t;
            if (open && at < L && a[t] != b[t]) {
                const uint32_t d = (uint32_t)__builtin_ctzll(a[t] ^ b[t]) >> 3;
                if (at + d < L) { cmp = ((b[t] >> (8 * d)) & 0xff) < ((a[t] >> (8 * d)) & 0xff) ? -1 : 1; share = at + d; }
                open = false;
            }
        }
        if (open && offset + 32 < L) cmp = -cmp_rest(c, q, len, qj, lj, offset + 32, &share);   // a longer phrase (cmp: -1 = j first)
        if (cmp == 0) {
            if (len != lj) atomicAdd(err + 1, 1u);
            const uint64_t mine = rec >> c.pos_bits, other = rj >> c.pos_bits;
            cmp = other < mine || (other == mine && j < e) ? -1 : 1;
            share = c.expand ? L : ~0ull;                        // (the same alpha: |alpha| is all the expansion asks; else the parse's LCP)
        }
        if (lcp_out && cmp < 0 && share != ~0ull) lcp_out[slot[g0] + 1u] = share < (uint64_t)LCP_CAP ? (uint32_t)share : LCP_CAP;
        out[slot[g0] + (cmp < 0 ? 1u : 0u)] = rec;
        flags[e] = 0;
        return;
    }
    const uint64_t len = rec_len(c, rec, q);
    uint64_t my_rank = ~0ull;                                    // looked up when first needed
    uint32_t before = 0;
    // (the element's LCP with its predecessor in the group = the most it shares with any member that sorts before it; a member
    // with the same alpha before it: |alpha| for the expansion, else left to k_batch_lcp and the parse's LCP array)
    uint64_t share_best = 0;
    bool same_before = false;
    for (uint32_t j = g0; j < end; j++) {
        if (j == e) continue;
        const uint64_t rj = pos[j], qj = rec_pos(c, rj);
        const uint64_t lj = rec_len(c, rj, qj);
        uint64_t share = 0;
        int cmp = -cmp_rest(c, q, len, qj, lj, offset, &share);   // -1: j sorts before e
        if (cmp < 0 && share > share_best) share_best = share;
        if (cmp == 0) {
            // the shorter alpha is a prefix of the other string: the two are the same phrase suffix (prefix-free)
            if (len != lj || c.skip) atomicAdd(err + (c.skip ? 0 : 1), 1u);
            if (c.skip) cmp = j < e ? -1 : 1;
            else {
                if (my_rank == ~0ull) my_rank = rec_rank_key(c, rec, q);
                const uint64_t other = rec_rank_key(c, rj, qj);
                cmp = other < my_rank || (other == my_rank && j < e) ? -1 : 1;
            }
            if (cmp < 0) same_before = true;
        }
        before += cmp < 0 ? 1u : 0u;
    }
    if (lcp_out && before) {
        const uint64_t v = same_before ? (c.expand ? len : ~0ull) : share_best;
        if (v != ~0ull) lcp_out[slot[g0] + before] = v < (uint64_t)LCP_CAP ? (uint32_t)v : LCP_CAP;
    }
    out[slot[g0] + before] = rec;
    flags[e] = 0;
}
void resolve_small(const Ctx& c, const uint64_t* pos, const uint32_t* ghead, const uint32_t* slot, uint32_t m, uint64_t offset,
                   uint64_t* out, uint8_t* flags, uint32_t* err, hipStream_t s, uint32_t* lcp_out, uint32_t limit) {
    hipLaunchKernelGGL(k_resolve_small, dim3(grid_for(m, 256)), dim3(256), 0, s, c, pos, ghead, slot, m, offset, out, flags, err, lcp_out,
                       limit ? limit : SMALL);
    MMT_HIP(hipGetLastError());
}

// Groups of SMALL < size <= MEDIUM elements -- the copies of one position in 9 .. 128 haplotypes -- are finished in one
// step as well: one wave per group stages, for every member, the next 64 characters behind the shared prefix, the length
// of its alpha and its parse rank in LDS, and every member counts the members that sort before it from there (the
// refinement rounds took 2.4 - 3.2 passes over such elements: 63 bits of alpha per round, then the parse ranks).  A group
// in which two members still agree after those 64 characters with alpha not yet spent is left to the rounds.
constexpr uint32_t MEDIUM = 128, MED_WORDS = 8;
// size classes of the medium groups: a wave takes 4 groups of <= 16, 2 of <= 32, 1 of <= 64 or 1 of <= 128 elements
__device__ __forceinline__ uint32_t medium_class(uint32_t size) { return size <= 16 ? 0u : (size <= 32 ? 1u : (size <= 64 ? 2u : 3u)); }
template <int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK) void k_medium_groups(const uint32_t* __restrict__ ghead, uint32_t m, uint2* __restrict__ list,
                                                         uint32_t cap, uint32_t* __restrict__ count) {
    // the LAST element of a group knows the group's size (no loop over the members); one list-slot allocation per
    // workgroup and class (a counter word takes ~150 atomics per microsecond: one per wave was 110 ms per batch)
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_base[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t first[PER], size[PER], at[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t e = (blockIdx.x * PER + k) * BLOCK + threadIdx.x;
        first[k] = 0; size[k] = 0; at[k] = 0;
        if (e < m) {
            const uint32_t f = ghead[e];
            if (e + 1 == m || ghead[e + 1] != f) {
                const uint32_t sz = e - f + 1;
                if (sz <= MEDIUM) { first[k] = f; size[k] = sz; at[k] = atomicAdd(&s_cnt[medium_class(sz)], 1u); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(count + threadIdx.x, s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (size[k]) {
            const uint32_t cl = medium_class(size[k]);
            list[(size_t)cl * cap + s_base[cl] + at[k]] = make_uint2(first[k], size[k]);
        }
}
void medium_groups(const uint32_t* ghead, uint32_t m, void* list, uint32_t cap, uint32_t* count4, hipStream_t s) {
    MMT_HIP(hipMemsetAsync(count4, 0, 16, s));
    hipLaunchKernelGGL((k_medium_groups<256, 4>), dim3(grid_for(m, 1024)), dim3(256), 0, s, ghead, m, static_cast<uint2*>(list), cap,
                       count4);
    MMT_HIP(hipGetLastError());
}

// the staged members of the groups one wave works on (W = 64 or 128 members)
template <int W>
struct MedStage {
    // (the class of up to 128 members -- the copies of a locus in the 94 haplotypes of BASELINE configs[3] / [4] -- stages 32
    // characters a member instead of 64: 56 KB of LDS a workgroup left two waves a SIMD)
    static constexpr uint32_t WORDS = W > 64 ? 4 : MED_WORDS;
    uint64_t w[W][WORDS];
    uint64_t rank[W];
    uint64_t rec[W];
    uint32_t len[W];
    uint32_t pid[W];                  // id of the distinct phrase the member starts in (0xffffffff: unknown)
    // every member against the REFERENCE member of its group: where the two alphas first differ (MED_SAME: they are the same
    // alpha; MED_UNKNOWN: not computed -- giant phrases, elements that are whole phrases), the member's and the reference's
    // character there
    uint32_t dref[W];
    uint16_t cref[W];
    // ... and, for the members that differ from the reference at the same place with the same character as an earlier member
    // (a mutation shared by descent; every member when the reference itself carries a private one): the first of them
    // (`lead`) and where the member's alpha first differs from THAT one's beyond the shared place, as above
    uint32_t d2[W];
    uint16_t c2[W];
    uint8_t lead[W];
    uint32_t gr[W], gg[W];            // members in giant phrases: entry of the giant dictionary's suffix array at `offset`, its group
    uint8_t idx[W];
    uint32_t bad;
    uint32_t text_cmps;              // MMT_GUIDED_PROF: comparisons the network sent to the text
};
// Order of two staged members (true: a sorts before b).  What the staged characters do not decide -- both alphas longer
// than what was staged and equal so far -- is compared in the text itself.  *lcp (optional) receives the number of
// characters the two suffixes share when the order was decided by a character; ~0 when they spell the same alpha.
constexpr uint32_t MED_SAME = 0xffffffffu, MED_UNKNOWN = 0xfffffffeu;
template <int W>
__device__ __forceinline__ bool med_before_chars(const Ctx& c, MedStage<W>& S, uint32_t a, uint32_t b, uint64_t offset, uint64_t* lcp);
// The members of a group are the copies of one locus: most spell the same alpha, the others differ from it in one place.
// Comparing two of them character by character costs a chain of dependent loads that runs up to that place -- and the
// sorting network asks log^2 times per member, every stage as slow as its slowest lane (with the phrases of a 250 G-character
// text, 171 characters on average, that was 5.1 of the 5.5 s a G suffixes took).  Each member is compared ONCE instead, with
// the REFERENCE member of its group (k_resolve_medium, after the staging: a member of the largest class of members that
// start at the same place of the same distinct phrase): d = the first position where its alpha differs from that one's.
// Then for two members a, b: d_a < d_b: b agrees with the reference at d_a, so a's character there against the reference's
// decides, and they share d_a characters; d_a = d_b: their own characters there decide.  If those are equal too -- a
// mutation two haplotypes share by descent, or the reference itself carrying a private one (then every other member) --
// both were compared a second time, with the first member of their kind (`lead`, d2, c2), and the same rule applies one
// level down; only a third shared difference sends the pair to the text (cmp_rest).
template <int W>
__device__ __forceinline__ bool med_before(const Ctx& c, MedStage<W>& S, uint32_t a, uint32_t b, uint64_t offset, uint64_t* lcp) {
    const uint32_t da = S.dref[a], db = S.dref[b];
    if (da == MED_UNKNOWN || db == MED_UNKNOWN) return med_before_chars<W>(c, S, a, b, offset, lcp);
    if (lcp) *lcp = ~0ull;
    if (da != db) {
        const bool a_first = da < db;
        const uint32_t x = a_first ? a : b;
        const uint32_t cc = S.cref[x];
        if (lcp) *lcp = a_first ? da : db;
        const bool x_small = (cc & 0xffu) < (cc >> 8);            // x's character against the first member's (= the other's)
        return a_first ? x_small : !x_small;
    }
    if (da != MED_SAME) {
        const uint32_t ca = S.cref[a] & 0xffu, cb = S.cref[b] & 0xffu;
        if (ca != cb) { if (lcp) *lcp = da; return ca < cb; }
        // the same place, the same character: both were compared with the first member of their kind
        const uint32_t ea = S.d2[a], eb = S.d2[b];
        uint64_t from = (uint64_t)da + 1;
        bool text = ea == MED_UNKNOWN || eb == MED_UNKNOWN;
        if (!text && ea != eb) {
            const bool a_first = ea < eb;
            const uint32_t cc = S.c2[a_first ? a : b];
            if (lcp) *lcp = a_first ? ea : eb;
            const bool x_small = (cc & 0xffu) < (cc >> 8);
            return a_first ? x_small : !x_small;
        }
        if (!text && ea != MED_SAME) {
            const uint32_t fa = S.c2[a] & 0xffu, fb = S.c2[b] & 0xffu;
            if (fa != fb) { if (lcp) *lcp = ea; return fa < fb; }
            text = true; from = (uint64_t)ea + 1;                  // (a third shared mutation: the text decides)
        }
        if (text) {
            if (c.prof) atomicAdd(&S.text_cmps, 1u);             // (counted in LDS, added up once per wave)
            const int r2 = cmp_rest(c, rec_pos(c, S.rec[a]), S.len[a], rec_pos(c, S.rec[b]), S.len[b], from, lcp);
            if (r2) return r2 < 0;
        }
    }
    if (S.len[a] != S.len[b]) S.bad = 1;
    const uint64_t ra = S.rank[a], rb = S.rank[b];
    return ra < rb || (ra == rb && a < b);
}
template <int W>
__device__ __forceinline__ bool med_before_chars(const Ctx& c, MedStage<W>& S, uint32_t a, uint32_t b, uint64_t offset, uint64_t* lcp) {
    const uint64_t la = S.len[a], lb = S.len[b];
    const uint64_t L = la < lb ? la : lb;
    if (lcp) *lcp = ~0ull;
    // The same distinct phrase and the same number of characters left of it: the same alpha, nothing to compare -- the copies
    // of one locus in the haplotypes that carry no mutation inside the phrase, i.e. most members of a group.  (With the
    // long phrases of a text of hundreds of G characters -- modulus 157 at 250 G -- nearly every pair used to run off the
    // staged characters into the text: 5.1 of 5.5 s per G suffixes.)
    const bool same_alpha = la == lb && S.pid[a] == S.pid[b] && S.pid[a] != 0xffffffffu;
    if (same_alpha) {
    } else if (S.gr[a] != 0xffffffffu && S.gr[b] != 0xffffffffu) {
        // two members in giant phrases: the giant dictionary knows their order and what they share
        const uint32_t ra = S.gr[a], rb = S.gr[b];
        if (S.gg[a] != S.gg[b]) {
            if (lcp) *lcp = offset + rmq_min(c.g_rmq, (ra < rb ? ra : rb) + 1, ra < rb ? rb : ra);
            return ra < rb;
        }
    } else
#pragma unroll
    for (uint32_t t = 0; t < MedStage<W>::WORDS; t++) {
        const uint64_t at = offset + 8 * t;
        if (at >= L) break;                                  // alpha of the shorter one is spent: same phrase suffix
        const uint64_t x = S.w[a][t], y = S.w[b][t];
        if (x != y) {
            const uint32_t d = (uint32_t)__builtin_ctzll(x ^ y) >> 3;
            if (at + d < L) { if (lcp) *lcp = at + d; return ((x >> (8 * d)) & 0xff) < ((y >> (8 * d)) & 0xff); }
            break;
        }
        if (t + 1 == MedStage<W>::WORDS && at + 8 < L) {     // (rare: alphas beyond the staged characters)
            const int r2 = cmp_rest(c, rec_pos(c, S.rec[a]), la, rec_pos(c, S.rec[b]), lb, at + 8, lcp);
            if (r2) return r2 < 0;
        }
    }
    if (la != lb || c.skip) S.bad = 1;
    if (c.skip) return a < b;
    const uint64_t ra = S.rank[a], rb = S.rank[b];
    return ra < rb || (ra == rb && a < b);
}
// SEG = slots per group (16 / 32 / 64 / 128), W = members per wave (64, or 128 with two per lane): W / SEG groups per wave
template <int SEG, int W>
__global__ __launch_bounds__(256) void k_resolve_medium(Ctx c, RmqView R, const uint64_t* __restrict__ pos,
                                                        const uint32_t* __restrict__ slot, const uint2* __restrict__ list,
                                                        uint32_t n_groups, uint64_t offset, uint64_t* __restrict__ out,
                                                        uint8_t* __restrict__ flags, uint32_t* __restrict__ lcp_out,
                                                        uint32_t* __restrict__ err) {
    constexpr int PERL = W / 64, GPW = W / SEG;              // members per lane, groups per wave
    __shared__ MedStage<W> s_st[4];
    __shared__ uint32_t s_g0[4][GPW], s_g[4][GPW], s_ref[4][GPW];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    MedStage<W>& S = s_st[wave];
    const uint64_t gw = ((uint64_t)blockIdx.x * 4 + wave) * GPW;     // first group of this wave
    if (lane < (uint32_t)GPW) {
        const bool have = gw + lane < n_groups;
        const uint2 grp = have ? list[gw + lane] : make_uint2(0u, 0u);
        s_g0[wave][lane] = grp.x; s_g[wave][lane] = have ? grp.y : 0u;
    }
    if (lane == 0) { S.bad = 0; S.text_cmps = 0; }
    unsigned long long tick = c.prof ? wall_clock64() : 0ull;
#define MMT_PROF(slot) do { if (c.prof) { const unsigned long long now_ = wall_clock64(); if (lane == 0) atomicAdd(c.prof + (slot), now_ - tick); tick = now_; } } while (0)
#define MMT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
    MMT_WAVE_SYNC();
#pragma unroll
    for (int h = 0; h < PERL; h++) {
        const uint32_t i = lane + 64 * h, seg = i / SEG, mem = i % SEG;
        S.idx[i] = (uint8_t)i;
        if (mem < s_g[wave][seg]) {
            const uint64_t rec = pos[s_g0[wave][seg] + mem];
            const uint64_t q = rec_pos(c, rec);
            constexpr uint32_t WORDS = MedStage<W>::WORDS;
            static_assert(WORDS == 8 || WORDS == 4, "the staging reads 64 or 32 characters");
            uint64_t c_lo = 0, c_hi = 0;
            if (c.T.is_packed() && tx_codes64(c.T, q + offset, c_lo, c_hi)) {       // (one round trip for the staged characters)
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) { S.w[i][t] = tx_expand8(c_lo >> (16 * t)); if (WORDS == 8) S.w[i][(4 + t) % WORDS] = tx_expand8(c_hi >> (16 * t)); }
            } else {
#pragma unroll
                for (uint32_t t = 0; t < WORDS; t++) S.w[i][t] = tx_load8(c.T, q + offset + 8 * t);
            }
            const uint64_t len = rec_len(c, rec, q);
            S.len[i] = len < 0xffffffffull ? (uint32_t)len : 0xffffffffu;
            uint32_t pid = 0xffffffffu;
            if (c.skip) S.rank[i] = 0ull;
            else if (c.expand) S.rank[i] = q;                        // (representatives: ties by position, no two share a phrase and an offset)
            else if (c.rec_rank) S.rank[i] = rec >> c.pos_bits;      // (no phrase lookup at all: the id is not worth three lines)
            else {                                                  // (one rank query serves both lookups)
                const uint32_t k = rank1(c, query_point(c, q));
                S.rank[i] = k + 1 < c.m ? (uint64_t)c.isa_p[k + 1] : 0ull;
                if (c.pid) pid = c.pid[k];
            }
            S.pid[i] = pid;
            S.rec[i] = rec;
            uint32_t r = 0xffffffffu;
            // (what the staged words cannot decide is looked up, not compared, when the member lies in a giant phrase)
            // (only an alpha that runs beyond giant depth needs it: a shorter rest of a giant phrase is compared in the text like
            // any other, at most g_depth characters -- and with phrases of 170 characters "longer than the staged 64" was nearly
            // every member, a rank query and a random line each)
            if (c.g_n && len > offset + 8 * WORDS && len > (uint64_t)c.g_depth && !giant_entry(c, q, offset, r)) r = 0xffffffffu;
            S.gr[i] = r;
            S.gg[i] = r != 0xffffffffu ? c.g_grp[r] : 0u;
        }
    }
    // (a wave works on its own stage: ordering its LDS traffic inside the wave is all the synchronisation there is)
    MMT_WAVE_SYNC();
    MMT_PROF(0);
    // The reference member of a group: one that starts in the group's MOST COMMON phrase at the same place -- the copies of a
    // locus without a mutation in their phrase -- so that the largest class of members needs no comparison at all and the
    // reference's alpha is the one most of the others spell too.  (The first member as the reference carries a private
    // mutation in every fourth group of a text with 170-character phrases: then EVERY other member differs from it at the
    // same place with the same character, and the sorting network below compared each pair of them in the text, log^2
    // times: 1.7 of the 2.1 s a G suffixes of such a text took.)
    if (lane < (uint32_t)GPW) s_ref[wave][lane] = 0u;
    MMT_WAVE_SYNC();
    if (c.pid && !c.skip && !c.rec_rank && !c.expand) {
#pragma unroll
        for (int h = 0; h < PERL; h++) {
            const uint32_t i = lane + 64 * h, seg = i / SEG, mem = i % SEG;
            const uint32_t g = s_g[wave][seg];
            if (mem >= g) continue;
            const uint32_t my_pid = S.pid[i], my_len = S.len[i];
            uint32_t same = 0;
            if (my_pid != 0xffffffffu && S.gr[i] == 0xffffffffu)
                for (uint32_t j = 0; j < g; j++) same += (S.pid[seg * SEG + j] == my_pid && S.len[seg * SEG + j] == my_len) ? 1u : 0u;
            atomicMax(&s_ref[wave][seg], (same << 8) | (uint32_t)(SEG - 1 - mem));
        }
        MMT_WAVE_SYNC();
    }
    MMT_PROF(1);
    if (c.prof && lane < (uint32_t)GPW && s_g[wave][lane]) {
        atomicAdd(c.prof + 11, (unsigned long long)(s_ref[wave][lane] >> 8));
        atomicAdd(c.prof + 12, (unsigned long long)s_g[wave][lane]);
        if ((s_ref[wave][lane] >> 8) <= 1) atomicAdd(c.prof + 13, 1ull);
    }
    // every member against the reference member of its group (med_before)
#pragma unroll
    for (int h = 0; h < PERL; h++) {
        const uint32_t i = lane + 64 * h, seg = i / SEG, mem = i % SEG;
        if (mem >= s_g[wave][seg]) continue;
        const uint32_t ref = seg * SEG + (s_ref[wave][seg] ? (uint32_t)(SEG - 1) - (s_ref[wave][seg] & 0xffu) : 0u);
        uint32_t d = MED_UNKNOWN, cc = 0;
        if (!c.skip && S.gr[i] == 0xffffffffu && S.gr[ref] == 0xffffffffu && !(c.g_n && S.len[ref] > offset + 8 * MedStage<W>::WORDS && S.len[i] > offset + 8 * MedStage<W>::WORDS && c.g_depth <= offset + 8 * MedStage<W>::WORDS)) {
            const uint64_t la = S.len[i], lb = S.len[ref], L = la < lb ? la : lb;
            if (i == ref || (la == lb && S.pid[i] == S.pid[ref] && S.pid[i] != 0xffffffffu)) d = MED_SAME;
            else {
                d = MED_SAME;
                bool decided = false;
#pragma unroll
                for (uint32_t t = 0; t < MedStage<W>::WORDS && !decided; t++) {
                    const uint64_t at = offset + 8 * t;
                    if (at >= L) { decided = true; break; }
                    const uint64_t x = S.w[i][t], y = S.w[ref][t];
                    if (x != y) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(x ^ y) >> 3;
                        if (at + k < L) { d = (uint32_t)(at + k); cc = (uint32_t)((x >> (8 * k)) & 0xff) | ((uint32_t)((y >> (8 * k)) & 0xff) << 8); }
                        decided = true;
                    }
                }
                if (!decided) {
                    const uint64_t qa = rec_pos(c, S.rec[i]), qb = rec_pos(c, S.rec[ref]);
                    // (a stretch that runs into giant depth is the giant dictionary's: left to the comparison of characters)
                    const uint64_t stop = c.g_n && L > (uint64_t)c.g_depth ? (uint64_t)c.g_depth : L;
                    const uint64_t t0 = offset + 8 * MedStage<W>::WORDS;
                    if (t0 < stop) {
                        uint32_t ca = 0, cb = 0;
                        const uint64_t at = first_diff(c, qa, qb, t0, stop, ca, cb);
                        if (at < stop) { d = (uint32_t)at; cc = ca | (cb << 8); decided = true; }
                    }
                    if (!decided && stop < L) d = MED_UNKNOWN;
                }
            }
        }
        S.dref[i] = d; S.cref[i] = (uint16_t)cc;
        if (c.prof) {                                           // (one atomic per wave and counter, not one per member)
            const unsigned long long diff = __ballot(d != MED_SAME && d != MED_UNKNOWN), unk = __ballot(d == MED_UNKNOWN);
            if (lane == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) { if (diff) atomicAdd(c.prof + 8, (unsigned long long)__popcll(diff)); if (unk) atomicAdd(c.prof + 9, (unsigned long long)__popcll(unk)); }
        }
    }
    MMT_WAVE_SYNC();
    // second level: the members that repeat an earlier member's (place, character) against the first of their kind
#pragma unroll
    for (int h = 0; h < PERL; h++) {
        const uint32_t i = lane + 64 * h, seg = i / SEG, mem = i % SEG;
        if (mem >= s_g[wave][seg]) continue;
        const uint32_t d = S.dref[i];
        uint32_t leader = i, e = MED_SAME, cc = 0;
        if (d != MED_SAME && d != MED_UNKNOWN) {
            const uint32_t mine = S.cref[i] & 0xffu;
            for (uint32_t j = seg * SEG; j < i; j++)
                if (S.dref[j] == d && (S.cref[j] & 0xffu) == mine) { leader = j; break; }
            if (leader != i) {
                const uint64_t la = S.len[i], lb = S.len[leader], L = la < lb ? la : lb;
                const uint64_t stop = c.g_n && L > (uint64_t)c.g_depth ? (uint64_t)c.g_depth : L;
                const uint64_t t0 = (uint64_t)d + 1;
                if (S.pid[i] != 0xffffffffu && S.pid[i] == S.pid[leader] && la == lb) {
                } else if (t0 < stop) {
                    uint32_t ca = 0, cb = 0;
                    const uint64_t at = first_diff(c, rec_pos(c, S.rec[i]), rec_pos(c, S.rec[leader]), t0, stop, ca, cb);
                    if (at < stop) { e = (uint32_t)at; cc = ca | (cb << 8); }
                    else if (stop < L) e = MED_UNKNOWN;
                } else if (stop < L) e = MED_UNKNOWN;
            }
        }
        S.lead[i] = (uint8_t)leader; S.d2[i] = e; S.c2[i] = (uint16_t)cc;
    }
    MMT_WAVE_SYNC();
    if (c.prof && lane < (uint32_t)GPW && s_g[wave][lane]) {
        uint32_t diff = 0, shared = 0;
        for (uint32_t j = 0; j < s_g[wave][lane]; j++) {
            const uint32_t dj = S.dref[lane * SEG + j];
            if (dj == MED_SAME || dj == MED_UNKNOWN) continue;
            diff++;
            for (uint32_t k2 = 0; k2 < j; k2++)
                if (S.dref[lane * SEG + k2] == dj && (S.cref[lane * SEG + k2] & 0xff) == (S.cref[lane * SEG + j] & 0xff)) { shared++; break; }
        }
        if (2 * diff >= s_g[wave][lane]) atomicAdd(c.prof + 14, 1ull);
        atomicAdd(c.prof + 15, (unsigned long long)shared);
    }
    MMT_PROF(2);
    // bitonic network over the SEG index slots of every group (members beyond the group sort last): log^2 steps of
    // compare-exchanges instead of g^2 / 2 comparisons
    for (uint32_t k = 2; k <= (uint32_t)SEG; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            bool swap[PERL];
            uint32_t a[PERL], b[PERL], lo[PERL];
#pragma unroll
            for (int h = 0; h < PERL; h++) {
                // pair number p of the wave (W / 2 pairs per step): its group, its place inside the group
                const uint32_t p = lane + 64 * h;
                swap[h] = false; a[h] = b[h] = lo[h] = 0;
                if (p < (uint32_t)W / 2) {
                    const uint32_t seg = p / (SEG / 2), pl = p % (SEG / 2);
                    const uint32_t ll = ((pl & ~(j - 1)) << 1) | (pl & (j - 1));
                    lo[h] = seg * SEG + ll;
                    const uint32_t g = s_g[wave][seg];
                    // (a group of at most k / 2 ... elements is sorted after fewer stages; the extra ones are no-ops on sorted data)
                    if (g > 1) {
                        const bool up = (ll & k) == 0;
                        a[h] = S.idx[lo[h]]; b[h] = S.idx[lo[h] | j];
                        const bool a_in = a[h] % SEG < g, b_in = b[h] % SEG < g;
                        bool a_first;                                // a sorts before b
                        if (a_in && b_in) a_first = med_before<W>(c, S, a[h], b[h], offset, nullptr);
                        else a_first = a_in || (!b_in && a[h] < b[h]);
                        swap[h] = up ? !a_first : a_first;
                    }
                }
            }
            MMT_WAVE_SYNC();
#pragma unroll
            for (int h = 0; h < PERL; h++)
                if (swap[h]) { S.idx[lo[h]] = (uint8_t)b[h]; S.idx[lo[h] | j] = (uint8_t)a[h]; }
            MMT_WAVE_SYNC();
        }
    }
#undef MMT_WAVE_SYNC
    MMT_PROF(3);
    if (S.bad && lane == 0) atomicAdd(err + (c.skip ? 0 : 1), 1u);
#pragma unroll
    for (int h = 0; h < PERL; h++) {
        const uint32_t i = lane + 64 * h, seg = i / SEG, mem = i % SEG;
        const uint32_t g = s_g[wave][seg], g0 = s_g0[wave][seg];
        if (mem >= g) continue;
        const uint32_t me = S.idx[i];
        const uint32_t at = slot[g0] + mem;
        out[at] = S.rec[me];
        flags[g0 + mem] = 0;
        if (lcp_out && mem > 0) {
            // LCP with the member before: decided by a character, or the same alpha and the parse ranks
            const uint32_t pr = S.idx[i - 1];
            uint64_t l = ~0ull;
            (void)med_before<W>(c, S, pr, me, offset, &l);
            if (l == ~0ull) {
                const uint64_t ra = S.rank[pr], rb = S.rank[me];
                l = rb > ra && !c.expand ? (uint64_t)S.len[me] - c.w + rmq_min(R, (uint32_t)ra + 1, (uint32_t)rb) : (uint64_t)S.len[me];
            }
            lcp_out[at] = l < (uint64_t)LCP_CAP ? (uint32_t)l : LCP_CAP;
        }
    }
    MMT_PROF(4);
    if (c.prof && lane == 0) { atomicAdd(c.prof + 7, 1ull); if (S.text_cmps) atomicAdd(c.prof + 10, (unsigned long long)S.text_cmps); }
#undef MMT_PROF
}
template <int SEG, int W>
static void resolve_medium_class(const Ctx& c, const RmqView& R, const uint64_t* pos, const uint32_t* slot, const uint2* list,
                                 uint32_t n_groups, uint64_t offset, uint64_t* out, uint8_t* flags, uint32_t* lcp_out, uint32_t* err,
                                 hipStream_t s) {
    constexpr uint32_t GPW = W / SEG;
    // (slices of 2^24 waves: a launch may not have 2^32 work-items)
    for (uint32_t first = 0; first < n_groups; first += GPW << 24) {
        const uint32_t part = std::min<uint32_t>(GPW << 24, n_groups - first);
        const uint64_t waves = (part + GPW - 1) / GPW;
        hipLaunchKernelGGL((k_resolve_medium<SEG, W>), dim3(grid_for(waves * 64, 256)), dim3(256), 0, s, c, R, pos, slot, list + first,
                           part, offset, out, flags, lcp_out, err);
    }
}
void resolve_medium(const Ctx& c, const RmqView& R, const uint64_t* pos, const uint32_t* slot, const void* list, uint32_t cap,
                    const uint32_t n_groups[4], uint64_t offset, uint64_t* out, uint8_t* flags, uint32_t* lcp_out, uint32_t* err,
                    hipStream_t s) {
    const uint2* l = static_cast<const uint2*>(list);
    if (n_groups[0]) resolve_medium_class<16, 64>(c, R, pos, slot, l, n_groups[0], offset, out, flags, lcp_out, err, s);
    if (n_groups[1]) resolve_medium_class<32, 64>(c, R, pos, slot, l + (size_t)cap, n_groups[1], offset, out, flags, lcp_out, err, s);
    if (n_groups[2]) resolve_medium_class<64, 64>(c, R, pos, slot, l + (size_t)2 * cap, n_groups[2], offset, out, flags, lcp_out, err, s);
    if (n_groups[3]) resolve_medium_class<128, 128>(c, R, pos, slot, l + (size_t)3 * cap, n_groups[3], offset, out, flags, lcp_out, err, s);
    MMT_HIP(hipGetLastError());
}

// tile t of the round sort begins at the first group head at or after t * target (NO_BOUND: none within `limit`)
__global__ void k_tile_bounds(const uint32_t* __restrict__ ghead, uint32_t m, uint32_t target, uint32_t limit,
                              uint32_t n_tiles, uint32_t* __restrict__ bound) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { bound[t] = m; return; }
    if (t == 0) { bound[0] = 0; return; }
    uint64_t c = (uint64_t)t * target;
    const uint64_t stop = c + limit < m ? c + limit : m;
    while (c < stop && ghead[c] != c) c++;
    bound[t] = c >= m ? m : (c == stop ? NO_BOUND : (uint32_t)c);
}
void tile_bounds(const uint32_t* ghead, uint32_t m, uint32_t target, uint32_t limit, uint32_t n_tiles, uint32_t* bound,
                 hipStream_t s) {
    hipLaunchKernelGGL(k_tile_bounds, dim3(grid_for((uint64_t)n_tiles + 1, 256)), dim3(256), 0, s, ghead, m, target, limit,
                       n_tiles, bound);
    MMT_HIP(hipGetLastError());
}

template <int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void k_local_sort(const uint64_t* __restrict__ kin, const uint64_t* __restrict__ pin,
                                                      const uint32_t* __restrict__ ghead, uint64_t* __restrict__ kout,
                                                      uint64_t* __restrict__ pout, const uint32_t* __restrict__ bound,
                                                      uint32_t n_tiles, uint32_t* __restrict__ big_begin,
                                                      uint32_t* __restrict__ big_end, uint32_t* __restrict__ big_count,
                                                      uint32_t big_cap) {
    __shared__ uint64_t s_k[CAP];
    __shared__ uint64_t s_p[CAP];
    __shared__ uint32_t s_h[CAP];
    __shared__ uint32_t s_long;
    const uint32_t t = blockIdx.x;
    const uint32_t b = bound[t];
    if (b == NO_BOUND) return;                              // this tile starts inside a group: an earlier tile owns it
    uint32_t u = t + 1;
    while (bound[u] == NO_BOUND) u++;                       // bound[n_tiles] = m always ends the search
    const uint32_t e = bound[u];
    if (e <= b) return;
    const uint32_t len = e - b;
    if (len > (uint32_t)CAP) {                              // holds a group longer than a tile: segmented sort (host)
        if (threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(big_count, 1u);
            if (slot < big_cap) { big_begin[slot] = b; big_end[slot] = e; }
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < len; i += BLOCK) { s_k[i] = kin[b + i]; s_p[i] = pin[b + i]; s_h[i] = ghead[b + i] - b; }
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    // groups are short: every element counts the members of its group that sort before it (stable)
    constexpr uint32_t SHORT = 64;
    constexpr int PER = CAP / BLOCK;
    uint32_t slot[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = threadIdx.x + q * BLOCK;
        slot[q] = i;
        if (i < len) {
            const uint32_t st = s_h[i];
            const uint64_t ki = s_k[i];
            uint32_t before = 0, j = st, cnt = 0;
            while (j < len && cnt <= SHORT) {
                if (s_h[j] != st) break;
                const uint64_t kj = s_k[j];
                before += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
                j++; cnt++;
            }
            if (cnt > SHORT) s_long = 1;
            slot[q] = st + before;
        }
    }
    __syncthreads();
    if (!s_long) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = threadIdx.x + q * BLOCK;
            if (i < len) { kout[b + slot[q]] = s_k[i]; pout[b + slot[q]] = s_p[i]; }
        }
        return;
    }
    // a longer group in the tile: bitonic network over (group, key, position in the tile)
    uint32_t P = 64;
    while (P < len) P <<= 1;
    for (uint32_t i = len + threadIdx.x; i < P; i += BLOCK) { s_k[i] = ~0ull; s_p[i] = 0; s_h[i] = 0xffffffffu; }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t ha = s_h[i], hc = s_h[l];
                    const uint64_t a = s_k[i], c2 = s_k[l];
                    const bool greater = ha != hc ? ha > hc : a > c2;
                    const bool up = (i & k) == 0;
                    if (greater == up && !(ha == hc && a == c2)) {
                        s_k[i] = c2; s_k[l] = a; s_h[i] = hc; s_h[l] = ha;
                        const uint64_t pa = s_p[i]; s_p[i] = s_p[l]; s_p[l] = pa;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < len; i += BLOCK) { kout[b + i] = s_k[i]; pout[b + i] = s_p[i]; }
}
void local_sort(const uint64_t* kin, const uint64_t* pin, const uint32_t* ghead, uint64_t* kout, uint64_t* pout,
                const uint32_t* bound, uint32_t n_tiles, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count,
                uint32_t big_cap, hipStream_t s) {
    hipLaunchKernelGGL((k_local_sort<256, (int)SORT_CAP>), dim3(n_tiles), dim3(256), 0, s, kin, pin, ghead, kout, pout, bound,
                       n_tiles, big_begin, big_end, big_count, big_cap);
    MMT_HIP(hipGetLastError());
}

// the groups inside the ranges [big_begin[r], big_end[r]) as segments of one segmented sort: the last element of a group
// appends (head, end) -- any order will do
__global__ void k_range_groups(const uint32_t* __restrict__ ghead, const uint32_t* __restrict__ big_begin,
                               const uint32_t* __restrict__ big_end, uint32_t* __restrict__ seg_begin,
                               uint32_t* __restrict__ seg_end, uint32_t* __restrict__ seg_count) {
    // (a range is cut into gridDim.y stretches: one range of a round is often a hundred times longer than the others)
    const uint32_t begin = big_begin[blockIdx.x], end = big_end[blockIdx.x];
    const uint32_t per = ((end - begin + gridDim.y - 1) / gridDim.y + 1023u) & ~1023u;
    const uint64_t lo64 = (uint64_t)begin + (uint64_t)blockIdx.y * per;
    if (lo64 >= end) return;
    const uint32_t lo = (uint32_t)lo64, hi = end - lo < per ? end : lo + per;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t c0 = lo; c0 < hi; c0 += blockDim.x) {
        const uint32_t c = c0 + threadIdx.x;
        const bool last = c < hi && (c + 1 == end || ghead[c + 1] == c + 1);
        const uint64_t bal = __ballot(last);
        if (!bal) continue;
        uint32_t at = 0;
        if (lane == (uint32_t)__builtin_ctzll(bal)) at = atomicAdd(seg_count, (uint32_t)__popcll(bal));
        at = __shfl(at, __builtin_ctzll(bal));
        if (last) {
            const uint32_t o = at + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
            seg_begin[o] = ghead[c];
            seg_end[o] = c + 1;
        }
    }
}
void range_groups(const uint32_t* ghead, const uint32_t* big_begin, const uint32_t* big_end, uint32_t big, uint32_t* seg_begin,
                  uint32_t* seg_end, uint32_t* seg_count, hipStream_t s) {
    MMT_HIP(hipMemsetAsync(seg_count, 0, 4, s));
    hipLaunchKernelGGL(k_range_groups, dim3(big, 64), dim3(1024), 0, s, ghead, big_begin, big_end, seg_begin, seg_end, seg_count);
    MMT_HIP(hipGetLastError());
}

// lcp_out (optional, indexed by batch slot): where two neighbours of a group part by the CHARACTERS of this round's keys -- both
// keys are symbol codes of the text behind the `offset` characters the group shares -- their suffixes share offset + the common
// symbols of the two keys, and so do the two sub-groups' final neighbours whatever the later rounds do inside them: the value is
// left for k_batch_lcp, which would otherwise walk those characters again from the start (satellite arrays: groups of millions
// that part after hundreds of characters -- 12 s of a 105 s share of realistic whole genomes)
__global__ void k_round_heads(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ ghead, uint32_t m,
                              uint32_t* __restrict__ headval, uint32_t* __restrict__ err, uint32_t* __restrict__ lcp_out,
                              const uint32_t* __restrict__ slot, uint64_t offset, int bits, int chars,
                              const uint32_t* __restrict__ g_grp, RmqView g_rmq, bool expand, bool giant_round,
                              const uint8_t* __restrict__ pure) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    const uint64_t k = keys[c];
    const uint32_t g0 = ghead[c];
    bool h = g0 == c;
    if (!h) {
        const uint64_t kp = keys[c - 1];
        if ((k >> 63) != (kp >> 63)) atomicAdd(err + 1, 1u);    // one group, spent and unspent phrase suffixes: not prefix-free
        h = (k >> 63) != 0 || k != kp;                           // parse ranks are distinct: such a group is done
        // (a key of characters may have bit 62 set as well -- 21 symbols of three bits: whether this group's keys are entries of
        // the giant dictionary is what the round, or k_giant_probe's flag of the group, says)
        const bool giant = (giant_round || (pure && pure[g0])) && (k >> 32) == (GIANT_KEY >> 32) && (kp >> 32) == (GIANT_KEY >> 32);
        if (giant) {
            // two entries of the giant dictionary (k_round_keys): one group of equal strings, or what its LCP array says
            const uint32_t r = (uint32_t)k, rp = (uint32_t)kp;
            const bool differ = g_grp[r] != g_grp[rp];
            h = differ || expand;             // (representatives that spell the same phrase suffix: final in any order)
            if (differ && lcp_out) {
                const uint64_t v = offset + (uint64_t)rmq_min(g_rmq, rp + 1, r);
                lcp_out[slot[c]] = v < (uint64_t)LCP_CAP ? (uint32_t)v : LCP_CAP;
            }
        } else if (lcp_out && k != kp && !((k | kp) >> 63) && !giant_round && !(pure && pure[g0])) {
            const uint64_t v = offset + (uint64_t)((__builtin_clzll(k ^ kp) - (64 - bits * chars)) / bits);
            lcp_out[slot[c]] = v < (uint64_t)LCP_CAP ? (uint32_t)v : LCP_CAP;
        }
    }
    headval[c] = h ? c : 0u;
}
void round_heads(const Ctx& ctx, const uint64_t* keys, const uint32_t* ghead, uint32_t m, uint32_t* headval, uint32_t* err, hipStream_t s,
                 uint32_t* lcp_out, const uint32_t* slot, uint64_t offset, const uint8_t* pure) {
    hipLaunchKernelGGL(k_round_heads, dim3(grid_for(m, 256)), dim3(256), 0, s, keys, ghead, m, headval, err, lcp_out, slot, offset,
                       ctx.bits, ctx.chars, ctx.g_grp, ctx.g_rmq, ctx.expand != 0, ctx.g_n != 0 && offset >= ctx.g_depth, pure);
    MMT_HIP(hipGetLastError());
}
__global__ void k_round_apply(const uint64_t* __restrict__ pos_sorted, const uint32_t* __restrict__ newhead,
                              const uint32_t* __restrict__ slot, uint32_t m, uint64_t* __restrict__ out,
                              uint8_t* __restrict__ flags) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    const bool single = newhead[c] == c && (c + 1 == m || newhead[c + 1] == c + 1);
    if (single) out[slot[c]] = pos_sorted[c];
    flags[c] = single ? 0 : 1;
}
void round_apply(const uint64_t* pos_sorted, const uint32_t* newhead, const uint32_t* slot, uint32_t m, uint64_t* out,
                 uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_round_apply, dim3(grid_for(m, 256)), dim3(256), 0, s, pos_sorted, newhead, slot, m, out, flags);
    MMT_HIP(hipGetLastError());
}
__global__ void k_round_compact(const uint32_t* __restrict__ idx, uint32_t m2, const uint32_t* __restrict__ slot,
                                const uint64_t* __restrict__ pos_sorted, const uint32_t* __restrict__ newhead,
                                uint32_t* __restrict__ slot_out, uint64_t* __restrict__ pos_out,
                                uint32_t* __restrict__ headval_out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m2) return;
    const uint32_t o = idx[c];
    slot_out[c] = slot[o]; pos_out[c] = pos_sorted[o];
    headval_out[c] = newhead[o] == o ? c : 0u;
}
void round_compact(const uint32_t* idx, uint32_t m2, const uint32_t* slot, const uint64_t* pos_sorted,
                   const uint32_t* newhead, uint32_t* slot_out, uint64_t* pos_out, uint32_t* headval_out, hipStream_t s) {
    hipLaunchKernelGGL(k_round_compact, dim3(grid_for(m2, 256)), dim3(256), 0, s, idx, m2, slot, pos_sorted, newhead,
                       slot_out, pos_out, headval_out);
    MMT_HIP(hipGetLastError());
}

// ---- the giant dictionary's bookkeeping (guided.cpp::build_giant) ---------------------------------------------------
__global__ void k_flag_greater(const uint32_t* __restrict__ v, uint32_t n, uint32_t thr, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = v[i] > thr ? 1u : 0u;
}
void flag_greater(const uint32_t* v, uint32_t n, uint32_t thr, uint32_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_greater, dim3(grid_for(n, 256)), dim3(256), 0, s, v, n, thr, flags);
    MMT_HIP(hipGetLastError());
}
// Promotion of the neighbours of giant occurrences (guided.cpp build_giant): out[k] = in[k - 1] | in[k] | in[k + 1];
// dflag[pid[k]] = 1 where an occurrence is flagged; flags[k] = dflag[pid[k]]
__global__ void k_flag_spread(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] | (i ? in[i - 1] : 0u) | (i + 1 < n ? in[i + 1] : 0u);
}
void flag_spread(const uint32_t* in, uint32_t n, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_spread, dim3(grid_for(n, 256)), dim3(256), 0, s, in, n, out);
    MMT_HIP(hipGetLastError());
}
__global__ void k_flag_to_distinct(const uint32_t* __restrict__ occ_flag, const uint32_t* __restrict__ pid, uint32_t m,
                                   uint32_t* __restrict__ dflag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m && occ_flag[k]) dflag[pid[k]] = 1u;
}
void flag_to_distinct(const uint32_t* occ_flag, const uint32_t* pid, uint32_t m, uint32_t* dflag, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_to_distinct, dim3(grid_for(m, 256)), dim3(256), 0, s, occ_flag, pid, m, dflag);
    MMT_HIP(hipGetLastError());
}
__global__ void k_flag_from_distinct(const uint32_t* __restrict__ dflag, const uint32_t* __restrict__ pid, uint32_t m,
                                     uint32_t* __restrict__ occ_flag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) occ_flag[k] = dflag[pid[k]];
}
void flag_from_distinct(const uint32_t* dflag, const uint32_t* pid, uint32_t m, uint32_t* occ_flag, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_from_distinct, dim3(grid_for(m, 256)), dim3(256), 0, s, dflag, pid, m, occ_flag);
    MMT_HIP(hipGetLastError());
}
__global__ void k_flag_scatter_ones(const uint32_t* __restrict__ ids, uint32_t n, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[ids[i]] = 1u;
}
void flag_scatter_ones(const uint32_t* ids, uint32_t n, uint32_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_scatter_ones, dim3(grid_for(n, 256)), dim3(256), 0, s, ids, n, flags);
    MMT_HIP(hipGetLastError());
}
__global__ void k_giant_distinct(const uint32_t* __restrict__ gids, uint32_t n, const uint32_t* __restrict__ rep,
                                 const uint32_t* __restrict__ dlen, uint32_t* __restrict__ which, uint32_t* __restrict__ glen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { which[i] = rep[gids[i]]; glen[i] = dlen[gids[i]]; }
}
void giant_distinct(const uint32_t* gids, uint32_t n, const uint32_t* rep, const uint32_t* dlen, uint32_t* which, uint32_t* glen,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_giant_distinct, dim3(grid_for(n, 256)), dim3(256), 0, s, gids, n, rep, dlen, which, glen);
    MMT_HIP(hipGetLastError());
}
__global__ void k_giant_map(const uint32_t* __restrict__ gids, const uint32_t* __restrict__ gstart, uint32_t n,
                            uint32_t* __restrict__ dmap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dmap[gids[i]] = gstart[i];
}
__global__ void k_giant_bits(const uint32_t* __restrict__ flags, uint32_t m, uint32_t* __restrict__ bits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)i * 32 >= m) return;
    uint32_t word = 0;
    for (uint32_t t = 0; t < 32 && (uint64_t)i * 32 + t < m; t++) word |= (flags[(uint64_t)i * 32 + t] ? 1u : 0u) << t;
    bits[i] = word;
}
__global__ void k_popcount_words(const uint32_t* __restrict__ bits, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)__popc(bits[i]);
}
void popcount_words(const uint32_t* bits, uint32_t n, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_popcount_words, dim3(grid_for(n, 256)), dim3(256), 0, s, bits, n, out);
    MMT_HIP(hipGetLastError());
}
void giant_bits(const uint32_t* flags, uint32_t m, uint32_t* bits, hipStream_t s) {
    hipLaunchKernelGGL(k_giant_bits, dim3(grid_for(((uint64_t)m + 31) / 32, 256)), dim3(256), 0, s, flags, m, bits);
    MMT_HIP(hipGetLastError());
}
void giant_map(const uint32_t* gids, const uint32_t* gstart, uint32_t n, uint32_t* dmap, hipStream_t s) {
    hipLaunchKernelGGL(k_giant_map, dim3(grid_for(n, 256)), dim3(256), 0, s, gids, gstart, n, dmap);
    MMT_HIP(hipGetLastError());
}
template <typename P>
__global__ void k_giant_occurrences(const uint32_t* __restrict__ gk, uint32_t n, const uint32_t* __restrict__ pid,
                                    const P* __restrict__ pstart, const uint32_t* __restrict__ dmap, uint64_t* __restrict__ gps,
                                    uint32_t* __restrict__ gbase) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t k = gk[j];
    gps[j] = (uint64_t)pstart[k];
    gbase[j] = dmap[pid[k]];
}
void giant_occurrences(const uint32_t* gk, uint32_t n, const uint32_t* pid, const void* pstart, bool wide, const uint32_t* dmap,
                       uint64_t* gps, uint32_t* gbase, hipStream_t s) {
    if (wide)
        hipLaunchKernelGGL(k_giant_occurrences<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, gk, n, pid,
                           static_cast<const uint64_t*>(pstart), dmap, gps, gbase);
    else
        hipLaunchKernelGGL(k_giant_occurrences<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, gk, n, pid,
                           static_cast<const uint32_t*>(pstart), dmap, gps, gbase);
    MMT_HIP(hipGetLastError());
}
// entry r of the giant dictionary's suffix array starts a new group unless it spells the string of entry r - 1 (same
// length, LCP >= length)
__global__ void k_giant_group_flags(const uint32_t* __restrict__ esuf, const uint32_t* __restrict__ lcp, uint32_t nd,
                                    uint32_t* __restrict__ flags) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nd) return;
    const uint32_t sl = esuf[r] & 0x7fffffffu;
    flags[r] = r == 0 || (esuf[r - 1] & 0x7fffffffu) != sl || lcp[r] < sl ? 1u : 0u;
}
void giant_group_flags(const uint32_t* esuf, const uint32_t* lcp, uint32_t nd, uint32_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_giant_group_flags, dim3(grid_for(nd, 256)), dim3(256), 0, s, esuf, lcp, nd, flags);
    MMT_HIP(hipGetLastError());
}

// ---- results -----------------------------------------------------------------------------------------------------
template <typename SA>
__global__ void k_write_columns(Ctx c, const uint64_t* __restrict__ pos, uint32_t B, uint64_t base, SA sa,
                                uint8_t* __restrict__ bwt) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const uint64_t q = rec_pos(c, pos[j]);                 // V index; text position q - 1
    sa.set(base + j, q - 1);
    bwt[base + j] = q == 1 ? (uint8_t)0 : tx_byte(c.T, q - 1);      // the Dollar before text position 0 reads as 0 (k_entry_info)
}
void write_columns(const Ctx& c, const uint64_t* pos, uint32_t B, uint64_t base, SaCol sa, uint8_t* bwt, hipStream_t s) {
    if (sa.wide())
        hipLaunchKernelGGL(k_write_columns<Sa40>, dim3(grid_for(B, 256)), dim3(256), 0, s, c, pos, B, base, Sa40(sa), bwt);
    else
        hipLaunchKernelGGL(k_write_columns<Sa32>, dim3(grid_for(B, 256)), dim3(256), 0, s, c, pos, B, base, Sa32(sa), bwt);
    MMT_HIP(hipGetLastError());
}
// LCP of every element of a sorted batch with the element before it (`carry`: the last element of the batch before).
// The two suffixes are compared up to the end of the shorter phrase suffix alpha; phrase suffixes are prefix-free, so a
// pair that is still equal there has the same alpha and shares |alpha| - w characters + the LCP of the two parse
// suffixes that follow: a range minimum over the LCP array of the parse (pfp_lcp_mum.hpp:295-321, parse_lcp.hpp).
__global__ void k_batch_lcp(Ctx c, RmqView R, const uint64_t* __restrict__ pos, uint32_t B, const uint64_t* __restrict__ carry,
                            int have_carry, uint32_t* __restrict__ lcp, uint32_t* __restrict__ err) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    if (j == 0 && !have_carry) { lcp[0] = 0; return; }
    if (lcp[j] != 0xffffffffu) return;                       // known since the sort separated the two (heads0, k_resolve_medium)
    const uint64_t ra = pos[j], rb = j ? pos[j - 1] : carry[0];
    const uint64_t qa = rec_pos(c, ra), qb = rec_pos(c, rb);
    const uint64_t la = rec_len(c, ra, qa), lb = rec_len(c, rb, qb);
    const uint64_t lim = la < lb ? la : lb;
    uint64_t v = 0;
    if (cmp_rest(c, qa, la, qb, lb, 0, &v) == 0) {
        if (c.expand) {
            // representatives of one group of equal phrase suffixes: in any order (the first keys may have ordered them by what
            // follows alpha, ties go by position), and "at least |alpha|" is all the group tables ask of their LCP
            if (la != lb) atomicAdd(err + 2, 1u);
            v = lim;
        } else {
            const uint64_t ka = rec_rank_key(c, ra, qa), kb = rec_rank_key(c, rb, qb);
            if (la != lb || kb >= ka) { atomicAdd(err + 2, 1u); v = lim; }
            else v = la - c.w + rmq_min(R, (uint32_t)kb + 1, (uint32_t)ka);
        }
    }
    lcp[j] = v < (uint64_t)LCP_CAP ? (uint32_t)v : LCP_CAP;
}
void batch_lcp(const Ctx& c, const RmqView& R, const uint64_t* pos, uint32_t B, const uint64_t* carry, bool have_carry,
               uint32_t* lcp, uint32_t* err, hipStream_t s) {
    if (!B) return;
    hipLaunchKernelGGL(k_batch_lcp, dim3(grid_for(B, 256)), dim3(256), 0, s, c, R, pos, B, carry, have_carry ? 1 : 0, lcp, err);
    MMT_HIP(hipGetLastError());
}

__global__ void k_phrase_ranks(Ctx c, const uint64_t* __restrict__ pos, uint32_t D, const uint32_t* __restrict__ pid,
                               uint32_t* __restrict__ prank) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const uint64_t q = rec_pos(c, pos[j]);
    prank[pid[rank1(c, query_point(c, q))]] = j + 1;
}
void phrase_ranks(const Ctx& c, const uint64_t* pos, uint32_t D, const uint32_t* pid, uint32_t* prank, hipStream_t s) {
    hipLaunchKernelGGL(k_phrase_ranks, dim3(grid_for(D, 256)), dim3(256), 0, s, c, pos, D, pid, prank);
    MMT_HIP(hipGetLastError());
}

// ---- expansion: a sorted batch of representative suffixes -> the entry tables of the emitter ---------------------------------
// Element e of the batch (suffix-array order of the phrase suffixes; lcp[e] = characters it shares with element e - 1, the
// carry of the batch before for e = 0) is one valid suffix of the parse's dictionary: its phrase k = rank of its query
// point, its distinct phrase d = pid[k], |alpha| from the next phrase end; tab[d] = (occurrences, first slot of the inverted
// list, length of the phrase).  Two neighbours spell the same alpha exactly when they share at least |alpha| characters
// (phrase suffixes are prefix-free): they belong to one group, whose lists the emitter merges by parse rank
// (pfp_lcp_mum.hpp:141-154 same_suffix).
__global__ void k_expand_entries(Ctx c, const uint64_t* __restrict__ pos, const uint32_t* __restrict__ lcp, uint32_t B,
                                 const uint4* __restrict__ tab, uint32_t* __restrict__ ce_cnt, uint32_t* __restrict__ ce_first,
                                 uint32_t* __restrict__ ce_offm1, uint8_t* __restrict__ ce_bwt, uint32_t* __restrict__ ce_gs,
                                 uint32_t* __restrict__ ce_hl, uint32_t* __restrict__ ce_slen, uint32_t* __restrict__ err) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const uint64_t rec = pos[e], q = rec_pos(c, rec);            // V index; text position q - 1
    const uint64_t x = query_point(c, q);
    const uint32_t k = rank1(c, x);
    const uint64_t slen = rec_len(c, rec, q);                    // (rides in the record unless it is beyond 2^24)
    const uint4 t = tab[c.pid[k]];
    const uint32_t l = lcp[e];
    if (slen >= (uint64_t)t.z || slen < (uint64_t)c.w) atomicAdd(err + 1, 1u);     // not a proper suffix of length >= w of its phrase
    ce_cnt[e] = t.x;
    ce_first[e] = t.y;
    ce_offm1[e] = t.z - (uint32_t)slen - 1u;
    ce_bwt[e] = q == 1 ? (uint8_t)0 : tx_byte(c.T, q - 1);
    ce_gs[e] = e == 0 || (uint64_t)l < slen ? 1u : 0u;
    ce_hl[e] = l;
    ce_slen[e] = (uint32_t)slen;
}
void expand_entries(const Ctx& c, const uint64_t* pos, const uint32_t* lcp, uint32_t B, const void* tab, uint32_t* ce_cnt,
                    uint32_t* ce_first, uint32_t* ce_offm1, uint8_t* ce_bwt, uint32_t* ce_gs, uint32_t* ce_hl, uint32_t* ce_slen,
                    uint32_t* err, hipStream_t s) {
    if (!B) return;
    hipLaunchKernelGGL(k_expand_entries, dim3(grid_for(B, 256)), dim3(256), 0, s, c, pos, lcp, B, static_cast<const uint4*>(tab),
                       ce_cnt, ce_first, ce_offm1, ce_bwt, ce_gs, ce_hl, ce_slen, err);
    MMT_HIP(hipGetLastError());
}
// ce_gs: flag -> group id + 1 at the first entry of a group (gscan = inclusive sum of the flags), 0 elsewhere
__global__ void k_group_ids(uint32_t* __restrict__ ce_gs, const uint32_t* __restrict__ gscan, uint32_t B) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B) ce_gs[e] = ce_gs[e] ? gscan[e] : 0u;
}
void group_ids(uint32_t* ce_gs, const uint32_t* gscan, uint32_t B, hipStream_t s) {
    if (!B) return;
    hipLaunchKernelGGL(k_group_ids, dim3(grid_for(B, 256)), dim3(256), 0, s, ce_gs, gscan, B);
    MMT_HIP(hipGetLastError());
}
// table[i] += add (stream offsets of a batch's entries: from batch-relative to absolute)
template <typename P>
__global__ void k_add_offset(P* __restrict__ t, uint32_t n, uint64_t add) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (P)((uint64_t)t[i] + add);
}
void add_offset(void* table, bool wide, uint32_t n, uint64_t add, hipStream_t s) {
    if (!n) return;
    if (wide) hipLaunchKernelGGL(k_add_offset<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<uint64_t*>(table), n, add);
    else hipLaunchKernelGGL(k_add_offset<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<uint32_t*>(table), n, add);
    MMT_HIP(hipGetLastError());
}
// bit k <=> phrase k of the parse is the representative occurrence of its distinct phrase (rep[pid[k]] == k)
__global__ void k_rep_bits(const uint32_t* __restrict__ pid, const uint32_t* __restrict__ rep, uint32_t m, uint32_t* __restrict__ bits) {
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)wi * 32 >= m) return;
    uint32_t v = 0;
    for (uint32_t b = 0; b < 32; b++) { const uint64_t k = (uint64_t)wi * 32 + b; if (k < m && rep[pid[k]] == (uint32_t)k) v |= 1u << b; }
    bits[wi] = v;
}
void rep_bits(const uint32_t* pid, const uint32_t* rep, uint32_t m, uint32_t* bits, hipStream_t s) {
    hipLaunchKernelGGL(k_rep_bits, dim3(grid_for(((uint64_t)m + 31) / 32, 256)), dim3(256), 0, s, pid, rep, m, bits);
    MMT_HIP(hipGetLastError());
}

}}  // namespace mmt::gk
