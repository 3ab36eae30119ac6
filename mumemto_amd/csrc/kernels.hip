// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
// Everything here is integer / byte work bounded by HBM bandwidth: loads are
// coalesced and, where a tile is re-read (keys, scan), staged through LDS.
// Reference behaviour restated by each kernel is cited at its definition.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include <cstdint>
#include <string>
#include <cstdlib>

#include "device_utils.hpp"
#include "kernels.hpp"
#include "wide.hpp"

namespace mmt { namespace k {

// A launch may not have 2^32 work-items or more (HIP folds the product of grid and workgroup size into 32 bits: a
// larger launch silently runs a fraction of its workgroups).  Every kernel here uses workgroups of at most 256
// work-items with grid_for, so 2^24 workgroups is the limit; kernels over text-sized ranges handle 4 - 16 items per
// work-item and stay below it for any text that fits the device.
static inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}

// ============================================================================
// A1  text layout -- RefBuilder::build_input_file (src/ref_builder.cpp:211-314)
//     and build_input_file_lib (:330-384): per document
//     UPPER(F) '$' [ revcomp(UPPER(F)) '$' ]; complement = the IUPAC table the
//     reference embeds (:29-38), letters only.
// ============================================================================
__device__ __forceinline__ uint8_t dev_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
__device__ __forceinline__ uint8_t dev_complement(uint8_t c) {
    switch (c) {
        case 'A': return 'T'; case 'B': return 'V'; case 'C': return 'G'; case 'D': return 'H';
        case 'G': return 'C'; case 'H': return 'D'; case 'K': return 'M'; case 'M': return 'K';
        case 'R': return 'Y'; case 'T': return 'A'; case 'U': return 'A'; case 'V': return 'B';
        case 'Y': return 'R'; default: return c;   // S W N and non-IUPAC bytes map to themselves
    }
}

// largest d in [0, n_docs] with start[d] <= p
__device__ __forceinline__ uint32_t doc_lookup(const uint64_t* __restrict__ start, uint32_t n_docs, uint64_t p) {
    uint32_t lo = 0, hi = n_docs;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (start[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// 16 output bytes per thread.  A thread whose 16 bytes lie inside one strand of one document (all
// but a handful per document) does two 8-byte loads, a table lookup per byte (upper-case, or
// complement of upper-case in reverse order) and one 16-byte store; the others go byte by byte.
// Alphabet histogram: A C G T N $ are counted in packed registers and reduced across the wave,
// anything else (rare) goes to the LDS histogram directly.
// 16 bytes from any address as two words, built from aligned 8-byte loads (a wave-wide load at a misaligned address
// is served lane by lane on this part).  Only words that hold at least one of the 16 bytes are touched.
__device__ __forceinline__ void load_16_bytes(const uint8_t* p, uint64_t& x, uint64_t& y) {
    const uint64_t* w = reinterpret_cast<const uint64_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)7);
    const uint32_t s = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 7u) * 8u;
    const uint64_t w0 = w[0], w1 = w[1];
    if (s == 0) { x = w0; y = w1; return; }
    const uint64_t w2 = w[2];
    x = (w0 >> s) | (w1 << (64 - s));
    y = (w1 >> s) | (w2 << (64 - s));
}
template <int BLOCK, int MAXD>
__global__ __launch_bounds__(BLOCK) void k_build_text(const uint8_t* __restrict__ raw,
                                                       const uint64_t* __restrict__ doc_base,
                                                       const uint64_t* __restrict__ doc_len,
                                                       const uint64_t* __restrict__ doc_start, uint32_t n_docs,
                                                       uint8_t* __restrict__ text, uint64_t n,
                                                       unsigned long long* __restrict__ hist) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint8_t s_up[256], s_rc[256], s_slot[256];
    __shared__ uint64_t s_start[MAXD + 1], s_base[MAXD + 1], s_len[MAXD + 1];
    for (int i = threadIdx.x; i < 256; i += BLOCK) {
        s_hist[i] = 0;
        const uint8_t u = dev_upper((uint8_t)i);
        s_up[i] = u; s_rc[i] = dev_complement(u);
        s_slot[i] = i == 'A' ? 0 : i == 'C' ? 1 : i == 'G' ? 2 : i == 'T' ? 3 : i == 'N' ? 4 : i == '$' ? 5 : 7;
    }
    const bool in_lds = n_docs <= (uint32_t)MAXD;
    if (in_lds)
        for (uint32_t i = threadIdx.x; i <= n_docs; i += BLOCK) {
            s_start[i] = doc_start[i]; s_base[i] = doc_base[i]; s_len[i] = i < n_docs ? doc_len[i] : 0;
        }
    __syncthreads();
    const uint64_t* st = in_lds ? s_start : doc_start;
    const uint64_t* bs = in_lds ? s_base : doc_base;
    const uint64_t* ln = in_lds ? s_len : doc_len;        // the documents need not be contiguous in `raw`
    // a workgroup walks over many tiles and adds its histogram to the global one once: with a workgroup per tile
    // the ~100,000 flushes queue up on five counter words (~90 atomics per microsecond each) and set the run time
    const uint64_t n_tiles = (n + (uint64_t)BLOCK * 16 - 1) / ((uint64_t)BLOCK * 16);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = (tile * BLOCK + threadIdx.x) * 16;
        uint8_t out[16];
        uint32_t valid = 0;
        if (p0 < n) {
            uint32_t d = doc_lookup(st, n_docs, p0);
            const uint64_t L = ln[d], local = p0 - st[d];
            if (local + 16 <= L) {                                              // forward strand
                const uint8_t* src = raw + bs[d] + local;
                uint64_t x, y;
                load_16_bytes(src, x, y);
#pragma unroll
                for (int b = 0; b < 8; b++) { out[b] = s_up[(x >> (8 * b)) & 0xff]; out[8 + b] = s_up[(y >> (8 * b)) & 0xff]; }
                valid = 0xffffu;
            } else if (local > L && local + 15 <= 2 * L) {                      // reverse strand
                const uint8_t* src = raw + bs[d] + (2 * L - local - 15);
                uint64_t x, y;
                load_16_bytes(src, x, y);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    out[15 - b] = s_rc[(x >> (8 * b)) & 0xff]; out[7 - b] = s_rc[(y >> (8 * b)) & 0xff];
                }
                valid = 0xffffu;
            } else {
#pragma unroll
                for (int b = 0; b < 16; b++) {
                    const uint64_t p = p0 + b;
                    uint8_t c = 0;
                    if (p < n) {
                        while (p >= st[d + 1]) d++;
                        const uint64_t Ld = ln[d], lo = p - st[d];
                        if (lo < Ld) c = s_up[raw[bs[d] + lo]];
                        else if (lo == Ld) c = '$';
                        else if (lo <= 2 * Ld) c = s_rc[raw[bs[d] + (2 * Ld - lo)]];
                        else c = '$';
                        valid |= 1u << b;
                    }
                    out[b] = c;
                }
            }
            uint4 w;
            uint32_t* wp = reinterpret_cast<uint32_t*>(&w);
#pragma unroll
            for (int q = 0; q < 4; q++)
                wp[q] = (uint32_t)out[4 * q] | ((uint32_t)out[4 * q + 1] << 8) | ((uint32_t)out[4 * q + 2] << 16) |
                        ((uint32_t)out[4 * q + 3] << 24);
            *reinterpret_cast<uint4*>(text + p0) = w;                           // buffer is padded past n
        }
        uint64_t pa = 0, pb = 0;      // 16-bit counters: pa = A C G T, pb = N $
#pragma unroll
        for (int b = 0; b < 16; b++) {
            if (!((valid >> b) & 1u)) continue;
            const uint32_t sl = s_slot[out[b]];
            const uint64_t inc = 1ull << ((sl & 3u) * 16);
            if (sl < 4) pa += inc; else if (sl < 6) pb += inc; else atomicAdd(&s_hist[out[b]], 1u);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { pa += __shfl_xor(pa, o, 64); pb += __shfl_xor(pb, o, 64); }
        if ((threadIdx.x & 63) == 0) {
            if (pa & 0xffffull) atomicAdd(&s_hist['A'], (uint32_t)(pa & 0xffff));
            if ((pa >> 16) & 0xffffull) atomicAdd(&s_hist['C'], (uint32_t)((pa >> 16) & 0xffff));
            if ((pa >> 32) & 0xffffull) atomicAdd(&s_hist['G'], (uint32_t)((pa >> 32) & 0xffff));
            if (pa >> 48) atomicAdd(&s_hist['T'], (uint32_t)(pa >> 48));
            if (pb & 0xffffull) atomicAdd(&s_hist['N'], (uint32_t)(pb & 0xffff));
            if ((pb >> 16) & 0xffffull) atomicAdd(&s_hist['$'], (uint32_t)((pb >> 16) & 0xffff));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += BLOCK)
        if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}

// ---- the text packed to two bits per character (textref.hpp) -----------------------------------------------------------------
// One work-item per packed word = 32 text positions, computed from the raw bases exactly as k_build_text computes its bytes
// (character by character here: the layout rules in one place, and this pass is a few per cent of a run).  What is not
// A C G T -- the '$' behind every strand, N, IUPAC codes -- is an EXCEPTION: code 0 in the word, and its maximal runs of one
// byte leave as events (start, byte) / (end) that the host pairs up into the sorted run list.
__device__ __forceinline__ uint8_t text_char_at(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ st,
                                                const uint64_t* __restrict__ bs, const uint64_t* __restrict__ ln, uint32_t n_docs,
                                                uint64_t p, const uint8_t* s_up, const uint8_t* s_rc) {
    const uint32_t d = doc_lookup(st, n_docs, p);
    const uint64_t Ld = ln[d], lo = p - st[d];
    if (lo < Ld) return s_up[raw[bs[d] + lo]];
    if (lo == Ld) return '$';
    if (lo <= 2 * Ld) return s_rc[raw[bs[d] + (2 * Ld - lo)]];
    return '$';
}
template <int BLOCK, int MAXD>
__global__ __launch_bounds__(BLOCK) void k_pack_text(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ doc_base,
                                                      const uint64_t* __restrict__ doc_len, const uint64_t* __restrict__ doc_start,
                                                      uint32_t n_docs, uint64_t* __restrict__ packed, uint64_t n,
                                                      unsigned long long* __restrict__ hist, uint64_t* __restrict__ ev_start,
                                                      uint64_t* __restrict__ ev_end, uint32_t* __restrict__ ev_count, uint32_t ev_cap,
                                                      uint64_t p_lo, uint64_t p_hi) {
    // [p_lo, p_hi): the text positions this launch packs -- everything, or the span of ONE document whose bases sit in a
    // staging buffer (`raw` is then offset so that raw + doc_base[d] is that buffer: a collection whose raw bases would not
    // fit the device next to its packed text is packed document by document, Engine::build_text).  A span begins and ends at
    // a document boundary: the character before it and the character behind it are not part of any of its runs ('$' ends
    // every document, no document is empty), and the two words it shares with its neighbours are OR-ed in.
    __shared__ uint32_t s_hist[256];
    __shared__ uint8_t s_up[256], s_rc[256];
    __shared__ uint64_t s_start[MAXD + 1], s_base[MAXD + 1], s_len[MAXD + 1];
    for (int i = threadIdx.x; i < 256; i += BLOCK) {
        s_hist[i] = 0;
        const uint8_t u = dev_upper((uint8_t)i);
        s_up[i] = u; s_rc[i] = dev_complement(u);
    }
    const bool in_lds = n_docs <= (uint32_t)MAXD;
    if (in_lds)
        for (uint32_t i = threadIdx.x; i <= n_docs; i += BLOCK) {
            s_start[i] = doc_start[i]; s_base[i] = doc_base[i]; s_len[i] = i < n_docs ? doc_len[i] : 0;
        }
    __syncthreads();
    const uint64_t* st = in_lds ? s_start : doc_start;
    const uint64_t* bs = in_lds ? s_base : doc_base;
    const uint64_t* ln = in_lds ? s_len : doc_len;
    const uint64_t w_lo = p_lo / 32, w_hi = (p_hi + 31) / 32;
    for (uint64_t W = w_lo + (uint64_t)blockIdx.x * BLOCK + threadIdx.x; W < w_hi; W += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t p0 = W * 32;
        uint32_t d = doc_lookup(st, n_docs, p0 > p_lo ? p0 : p_lo);
        uint64_t word = 0;
        uint32_t exc = 0;
        uint8_t ch[32];
        uint32_t cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const uint64_t p = p0 + b;
            uint8_t c = 0;
            if (p >= p_lo && p < p_hi) {
                while (p >= st[d + 1]) d++;
                const uint64_t Ld = ln[d], lo = p - st[d];
                if (lo < Ld) c = s_up[raw[bs[d] + lo]];
                else if (lo == Ld) c = '$';
                else if (lo <= 2 * Ld) c = s_rc[raw[bs[d] + (2 * Ld - lo)]];
                else c = '$';
                const uint32_t code = tx_code_of(c);
                if (code < 4) { word |= (uint64_t)code << (2 * b); cnt[code]++; }
                else { exc |= 1u << b; atomicAdd(&s_hist[c], 1u); }
            }
            ch[b] = c;
        }
        if (p0 >= p_lo && p0 + 32 <= p_hi) packed[W] = word;
        else atomicOr(reinterpret_cast<unsigned long long*>(packed + W), (unsigned long long)word);
        if (cnt[0]) atomicAdd(&s_hist['A'], cnt[0]);
        if (cnt[1]) atomicAdd(&s_hist['C'], cnt[1]);
        if (cnt[2]) atomicAdd(&s_hist['G'], cnt[2]);
        if (cnt[3]) atomicAdd(&s_hist['T'], cnt[3]);
        if (exc) {
            // runs of one exception byte: a start where the byte before differs, an end where the byte behind differs
            const uint8_t before = p0 > p_lo ? (uint8_t)((exc & 1u) ? text_char_at(raw, st, bs, ln, n_docs, p0 - 1, s_up, s_rc) : 0) : (uint8_t)0;
            const uint8_t behind = (p0 + 32 < p_hi && (exc >> 31)) ? text_char_at(raw, st, bs, ln, n_docs, p0 + 32, s_up, s_rc) : (uint8_t)0;
#pragma unroll
            for (int b = 0; b < 32; b++) {
                if (!((exc >> b) & 1u)) continue;
                const uint8_t prev = b ? ch[b ? b - 1 : 0] : before, next = b < 31 ? ch[b < 31 ? b + 1 : 31] : behind;
                if (prev != ch[b] || p0 + b == p_lo) {
                    const uint32_t slot = atomicAdd(ev_count, 1u);
                    if (slot < ev_cap) ev_start[slot] = ((p0 + b) << 8) | ch[b];
                }
                if (next != ch[b] || p0 + b + 1 == p_hi) {
                    const uint32_t slot = atomicAdd(ev_count + 1, 1u);
                    if (slot < ev_cap) ev_end[slot] = p0 + b + 1;
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += BLOCK)
        if (s_hist[i]) atomicAdd(&hist[i], (unsigned long long)s_hist[i]);
}
void pack_text(const uint8_t* raw, const uint64_t* d_doc_base, const uint64_t* d_doc_len, const uint64_t* d_doc_start, uint32_t n_docs,
               uint64_t* packed, uint64_t n, uint64_t* hist, uint64_t* ev_start, uint64_t* ev_end, uint32_t* ev_count, uint32_t ev_cap,
               uint64_t p_lo, uint64_t p_hi, hipStream_t s) {
    constexpr int B = 256;
    if (p_hi <= p_lo) return;
    const uint64_t words = (p_hi + 31) / 32 - p_lo / 32;
    const unsigned grid = (unsigned)std::min<uint64_t>((words + B - 1) / B ? (words + B - 1) / B : 1, 256u * 32u);
    hipLaunchKernelGGL((k_pack_text<B, 1023>), dim3(grid), dim3(B), 0, s, raw, d_doc_base, d_doc_len, d_doc_start, n_docs, packed, n,
                       reinterpret_cast<unsigned long long*>(hist), ev_start, ev_end, ev_count, ev_cap, p_lo, p_hi);
    MMT_HIP(hipGetLastError());
}
// a byte text (a handed-over text: Engine::set_text_host) packed the same way
__global__ void k_pack_bytes(const uint8_t* __restrict__ text, uint64_t n, uint64_t* __restrict__ packed, uint64_t* __restrict__ ev_start,
                             uint64_t* __restrict__ ev_end, uint32_t* __restrict__ ev_count, uint32_t ev_cap) {
    const uint64_t n_words = (n + 31) / 32;
    for (uint64_t W = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; W < n_words; W += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = W * 32;
        uint64_t word = 0;
        for (int b = 0; b < 32; b++) {
            const uint64_t p = p0 + b;
            if (p >= n) break;
            const uint8_t c = text[p];
            const uint32_t code = tx_code_of(c);
            if (code < 4) { word |= (uint64_t)code << (2 * b); continue; }
            const uint8_t prev = p ? text[p - 1] : (uint8_t)0, next = p + 1 < n ? text[p + 1] : (uint8_t)0;
            if (prev != c || p == 0) { const uint32_t slot = atomicAdd(ev_count, 1u); if (slot < ev_cap) ev_start[slot] = (p << 8) | c; }
            if (next != c || p + 1 == n) { const uint32_t slot = atomicAdd(ev_count + 1, 1u); if (slot < ev_cap) ev_end[slot] = p + 1; }
        }
        packed[W] = word;
    }
}
void pack_bytes(const uint8_t* text, uint64_t n, uint64_t* packed, uint64_t* ev_start, uint64_t* ev_end, uint32_t* ev_count,
                uint32_t ev_cap, hipStream_t s) {
    const uint64_t words = (n + 31) / 32;
    const unsigned grid = (unsigned)std::min<uint64_t>((words + 255) / 256 ? (words + 255) / 256 : 1, 256u * 32u);
    hipLaunchKernelGGL(k_pack_bytes, dim3(grid), dim3(256), 0, s, text, n, packed, ev_start, ev_end, ev_count, ev_cap);
    MMT_HIP(hipGetLastError());
}
// V[first .. first + count) of a packed text as bytes (Engine::copy_text, tests)
__global__ void k_unpack_text(const TextRef T, uint64_t first, uint64_t count, uint8_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = tx_byte(T, first + i);
}
void unpack_text(const TextRef& T, uint64_t first, uint64_t count, uint8_t* out, hipStream_t s) {
    if (!count) return;
    const uint64_t g = (count + 255) / 256;
    hipLaunchKernelGGL(k_unpack_text, dim3((unsigned)(g < 65536 ? g : 65536)), dim3(256), 0, s, T, first, count, out);
    MMT_HIP(hipGetLastError());
}

void build_text(const uint8_t* raw, const uint64_t* d_doc_base, const uint64_t* d_doc_len, const uint64_t* d_doc_start, uint32_t n_docs,
                bool /*revcomp*/, uint8_t* text, uint64_t n, uint64_t* hist, hipStream_t s) {
    constexpr int B = 256;
    const uint64_t tiles = (n + (uint64_t)B * 16 - 1) / ((uint64_t)B * 16);
    const unsigned grid = (unsigned)std::min<uint64_t>(tiles ? tiles : 1, 256u * 16u);
    hipLaunchKernelGGL((k_build_text<B, 1023>), dim3(grid), dim3(B), 0, s, raw, d_doc_base, d_doc_len, d_doc_start, n_docs, text, n,
                       reinterpret_cast<unsigned long long*>(hist));
    MMT_HIP(hipGetLastError());
}

// ============================================================================
// A8  direct suffix sort: prefix doubling over a radix sort.
//     Replaces gsacak(text) of include/direct_gsacak.hpp:62 (suffix array of
//     the whole text; the end of the text is smaller than every symbol).
// ============================================================================
// Tile of the text is converted to `bits`-wide symbol codes in LDS once; each
// thread then builds 4 consecutive keys with a rolling shift.
// sep_code != PACK_NO_SEP (PFP dictionary): every occurrence of that symbol is a UNIQUE terminator, ordered by position
// (its code may be 0, the code of the padding behind the end: nothing is compared behind a terminator) --
// the strings between terminators are what matters there, and a suffix is fully ordered as soon as its window
// reaches its terminator.  The key then is (symbols up to and including the first terminator, rest zeroed) << 1
// | 1; without a terminator in the window (symbols) << 1.  Equal keys with the low bit set stay in position
// order (stable radix sort), and k_mark_heads makes each of them its own bucket.
template <int BLOCK, int PER>
__global__ __launch_bounds__(BLOCK) void k_pack_keys(const uint8_t* __restrict__ text, uint32_t n,
                                                     const uint8_t* __restrict__ code, int bits, int chars,
                                                     uint32_t sep_code, uint64_t* __restrict__ keys,
                                                     uint32_t* __restrict__ vals, uint32_t* __restrict__ run_ends,
                                                     uint32_t* __restrict__ run_count, uint32_t run_cap, uint64_t rep) {
    constexpr int TILE = BLOCK * PER;
    __shared__ uint8_t s_code[256];
    __shared__ uint8_t s_sym[TILE + 64];
    for (int i = threadIdx.x; i < 256; i += BLOCK) s_code[i] = code[i];
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    for (int i = threadIdx.x; i < TILE + 64; i += BLOCK) {
        uint64_t p = base + i;
        s_sym[i] = p < n ? s_code[text[p]] : (uint8_t)0;
    }
    __syncthreads();
    const int t0 = threadIdx.x * PER;
    const uint64_t mask = (bits * chars >= 64) ? ~0ull : ((1ull << (bits * chars)) - 1);
    int next_sep[PER];                                   // offset of the first terminator in window q, or >= chars
    if (sep_code != PACK_NO_SEP) {
        int nx = 1 << 20;
        for (int idx = t0 + PER - 1 + chars - 1; idx >= t0; idx--) {
            if (s_sym[idx] == sep_code) nx = idx;
            if (idx < t0 + PER) next_sep[idx - t0] = nx - idx;
        }
    }
    uint64_t key = 0;
    for (int c = 0; c < chars; c++) key = (key << bits) | s_sym[t0 + c];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        uint64_t p = base + t0 + q;
        if (p < n) {
            uint64_t kq = key & mask;
            // the last window of a run of `chars` or more equal symbols (sorter.hpp, RunRefine): listed, in any order
            if (run_ends) {
                const uint32_t c0 = s_sym[t0 + q];
                if (c0 && kq == (uint64_t)c0 * rep && s_sym[t0 + q + chars] != c0) {
                    const uint32_t at = atomicAdd(run_count, 1u);
                    if (at < run_cap) run_ends[at] = (uint32_t)p + (uint32_t)chars - 1u;
                }
            }
            if (sep_code != PACK_NO_SEP) {
                const int d = next_sep[q];
                if (d < chars) kq = ((kq >> (bits * (chars - 1 - d))) << (bits * (chars - 1 - d)) << 1) | 1ull;
                else kq <<= 1;
            }
            keys[p] = kq; vals[p] = (uint32_t)p;
        }
        key = ((key << bits) | s_sym[t0 + q + chars]) & mask;
    }
}
void pack_keys(const uint8_t* text, uint32_t n, const uint8_t* d_code, int bits, int chars, uint32_t sep_code,
               uint64_t* keys, uint32_t* vals, hipStream_t s, uint32_t* run_ends, uint32_t* run_count, uint32_t run_cap) {
    constexpr int B = 256, PER = 4;
    uint64_t rep = 0;
    for (int c = 0; c < chars; c++) rep |= 1ull << (bits * c);
    hipLaunchKernelGGL((k_pack_keys<B, PER>), dim3(grid_for(n, B * PER)), dim3(B), 0, s, text, n, d_code, bits, chars,
                       sep_code, keys, vals, run_ends, run_count, run_cap, rep);
    MMT_HIP(hipGetLastError());
}

// ---- long runs of one symbol (sorter.hpp, RunRefine) -------------------------------------------------------------------------
// lo_hi[2 k], lo_hi[2 k + 1] = the range of sorted[] that equals probe[k]
__global__ void k_equal_range_u64(const uint64_t* __restrict__ sorted, uint32_t n, const uint64_t* __restrict__ probe,
                                  uint32_t n_probes, uint32_t* __restrict__ lo_hi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n_probes) return;
    const uint64_t x = probe[t >> 1];
    const bool upper = t & 1u;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t v = sorted[mid];
        if (upper ? v <= x : v < x) lo = mid + 1; else hi = mid;
    }
    lo_hi[t] = lo;
}
void equal_range_u64(const uint64_t* sorted, uint32_t n, const uint64_t* probe, uint32_t n_probes, uint32_t* lo_hi, hipStream_t s) {
    hipLaunchKernelGGL(k_equal_range_u64, dim3(grid_for(2 * n_probes, 64)), dim3(64), 0, s, sorted, n, probe, n_probes, lo_hi);
    MMT_HIP(hipGetLastError());
}

// The suffixes sa[0 .. cnt) all begin with `chars` copies of one symbol c.  Their order is decided by where the run ends and
// by what follows it: c^j X (X0 != c).  With X0 < c -- class 0 -- a shorter run is the smaller suffix, with X0 > c -- class 1,
// behind all of class 0 -- the longer one; equal j: the symbols behind the run.  key2 = class | j or its complement (24
// bits) | 12 symbols behind the run (zero behind a terminator) | 00 | "reached its terminator" -- the low bit of k_pack_keys.
// A run of 2^24 - 1 symbols or more: the symbols behind it are left out (such suffixes tie: they agree in that many
// characters).  ends: ascending last positions of the runs of `chars` symbols or more.
__global__ void k_run_keys(const uint32_t* __restrict__ sa, uint32_t cnt, const uint8_t* __restrict__ text, uint32_t n,
                           const uint8_t* __restrict__ code, int bits, int chars, const uint32_t* __restrict__ ends,
                           uint32_t n_ends, uint64_t* __restrict__ key2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t p = sa[i];
    const uint32_t want = p + (uint32_t)chars - 1u;                  // the run of p ends at the first listed end >= want
    uint32_t lo = 0, hi = n_ends;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (ends[mid] < want) lo = mid + 1; else hi = mid; }
    const uint32_t e = lo < n_ends ? ends[lo] : want;                // (cannot be missing: every run of that length is listed)
    const uint32_t c = code[text[p]];
    const uint32_t CAP = 0xffffffu;
    uint32_t j = e - p + 1u;
    const bool capped = j >= CAP;
    if (capped) j = CAP;
    const uint32_t x0 = e + 1u < n ? code[text[e + 1u]] : 0u;
    const uint64_t cls = x0 > c ? 1ull : 0ull;
    uint64_t k = (cls << 63) | ((uint64_t)(cls ? CAP - j : j) << 39);
    if (!capped) {
        uint64_t x = 0, fin = 0;
        for (uint32_t t = 0; t < 12u; t++) {
            const uint32_t q = e + 1u + t;
            const uint32_t sym = q < n ? code[text[q]] : 0u;
            x = (x << bits) | sym;
            if (sym == 0u) { x <<= bits * (11u - t); fin = 1; break; }     // a terminator: nothing behind it is compared
        }
        k |= (x << 3) | fin;
    }
    key2[i] = k;
}
void run_keys(const uint32_t* sa, uint32_t cnt, const uint8_t* text, uint32_t n, const uint8_t* code, int bits, int chars,
              const uint32_t* ends, uint32_t n_ends, uint64_t* key2, hipStream_t s) {
    if (bits > 3) throw std::runtime_error("run_keys: symbols wider than three bits");
    hipLaunchKernelGGL(k_run_keys, dim3(grid_for(cnt, 256)), dim3(256), 0, s, sa, cnt, text, n, code, bits, chars, ends, n_ends, key2);
    MMT_HIP(hipGetLastError());
}
__global__ void k_force_heads(uint32_t* __restrict__ headval, const uint32_t* __restrict__ at, uint32_t cnt, uint32_t n) {
    const uint32_t t = threadIdx.x;
    if (t < cnt && at[t] < n) headval[at[t]] = at[t];
}
void force_heads(uint32_t* headval, const uint32_t* at, uint32_t cnt, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_force_heads, dim3(1), dim3(64), 0, s, headval, at, cnt, n);
    MMT_HIP(hipGetLastError());
}

__global__ void k_mark_heads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ headval,
                             int lsb_unique) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t k = keys[j];
    headval[j] = (j == 0 || k != keys[j - 1] || (lsb_unique && (k & 1ull))) ? j : 0u;
}
void mark_heads(const uint64_t* keys, uint32_t n, uint32_t* headval, bool lsb_unique, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_heads, dim3(grid_for(n, 256)), dim3(256), 0, s, keys, n, headval, lsb_unique ? 1 : 0);
    MMT_HIP(hipGetLastError());
}

__global__ void k_scatter_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ head, uint32_t n,
                               uint32_t* __restrict__ rank) {
    // four entries per thread: 16-byte loads of both columns, four stores in flight
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j + 4 <= n) {
        const uint4 p = *reinterpret_cast<const uint4*>(sa + j), v = *reinterpret_cast<const uint4*>(head + j);
        rank[p.x] = v.x; rank[p.y] = v.y; rank[p.z] = v.z; rank[p.w] = v.w;
    } else {
        for (uint64_t t = j; t < n; t++) rank[sa[t]] = head[t];
    }
}
// the same for the sorted list of a round, where most ranks stay what they were: the elements of the first sub-bucket of
// every bucket keep the bucket's head.  Sorting permutes inside buckets only, so element c's old head is old_head[c].
__global__ void k_scatter_rank_changed(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ head,
                                       const uint32_t* __restrict__ old_head, uint32_t m, uint32_t* __restrict__ rank) {
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j + 4 <= m) {
        const uint4 v = *reinterpret_cast<const uint4*>(head + j), o = *reinterpret_cast<const uint4*>(old_head + j);
        if (v.x == o.x && v.y == o.y && v.z == o.z && v.w == o.w) return;
        const uint4 p = *reinterpret_cast<const uint4*>(sa + j);
        if (v.x != o.x) rank[p.x] = v.x;
        if (v.y != o.y) rank[p.y] = v.y;
        if (v.z != o.z) rank[p.z] = v.z;
        if (v.w != o.w) rank[p.w] = v.w;
    } else {
        for (uint64_t t = j; t < m; t++) if (head[t] != old_head[t]) rank[sa[t]] = head[t];
    }
}
void scatter_rank_changed(const uint32_t* sa, const uint32_t* head, const uint32_t* old_head, uint32_t m, uint32_t* rank,
                          hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_rank_changed, dim3(grid_for(m, 1024)), dim3(256), 0, s, sa, head, old_head, m, rank);
    MMT_HIP(hipGetLastError());
}
void scatter_rank(const uint32_t* sa, const uint32_t* head, uint32_t n, uint32_t* rank, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_rank, dim3(grid_for(n, 1024)), dim3(256), 0, s, sa, head, n, rank);
    MMT_HIP(hipGetLastError());
}

__global__ void k_flag_unsorted(const uint32_t* __restrict__ head, uint32_t n, uint8_t* __restrict__ flags) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    bool single = head[j] == j && (j + 1 == n || head[j + 1] == j + 1);
    flags[j] = single ? 0 : 1;
}
void flag_unsorted(const uint32_t* head, uint32_t n, uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_unsorted, dim3(grid_for(n, 256)), dim3(256), 0, s, head, n, flags);
    MMT_HIP(hipGetLastError());
}

__global__ void k_gather_active(const uint32_t* __restrict__ idx, uint32_t m, const uint32_t* __restrict__ sa,
                                const uint32_t* __restrict__ head, uint32_t* __restrict__ out_pos,
                                uint32_t* __restrict__ out_sa, uint32_t* __restrict__ out_head) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    uint32_t j = idx[c];
    out_pos[c] = j; out_sa[c] = sa[j]; out_head[c] = head[j];
}
void gather_active(const uint32_t* idx, uint32_t m, const uint32_t* sa, const uint32_t* head, uint32_t* out_pos,
                   uint32_t* out_sa, uint32_t* out_head, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_active, dim3(grid_for(m, 256)), dim3(256), 0, s, idx, m, sa, head, out_pos, out_sa,
                       out_head);
    MMT_HIP(hipGetLastError());
}

__global__ void k_make_round_keys(const uint32_t* __restrict__ sa_c, const uint32_t* __restrict__ head_c, uint32_t m,
                                  const uint32_t* __restrict__ rank, uint32_t n, uint32_t h, int shift,
                                  uint64_t* __restrict__ keys) {
    // four entries per thread: the gathers of rank[sa + h] are independent and all in flight together
    const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c + 4 <= m) {
        const uint4 p = *reinterpret_cast<const uint4*>(sa_c + c), hd = *reinterpret_cast<const uint4*>(head_c + c);
        const uint32_t pv[4] = {p.x, p.y, p.z, p.w}, hv[4] = {hd.x, hd.y, hd.z, hd.w};
        uint64_t second[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint64_t i = (uint64_t)pv[t] + h;
            second[t] = i < n ? (uint64_t)rank[i] + 1 : 0;   // past the end sorts first
        }
#pragma unroll
        for (int t = 0; t < 4; t++) keys[c + t] = ((uint64_t)hv[t] << shift) | second[t];
    } else {
        for (uint64_t t = c; t < m; t++) {
            const uint64_t i = (uint64_t)sa_c[t] + h;
            keys[t] = ((uint64_t)head_c[t] << shift) | (i < n ? (uint64_t)rank[i] + 1 : 0);
        }
    }
}
void make_round_keys(const uint32_t* sa_c, const uint32_t* head_c, uint32_t m, const uint32_t* rank, uint32_t n,
                     uint32_t h, int shift, uint64_t* keys, hipStream_t s) {
    hipLaunchKernelGGL(k_make_round_keys, dim3(grid_for(m, 1024)), dim3(256), 0, s, sa_c, head_c, m, rank, n, h, shift,
                       keys);
    MMT_HIP(hipGetLastError());
}

// ---- local sort of a doubling round -------------------------------------------------------------
// The active elements of a round arrive grouped by bucket (the high key bits): sorting them only permutes elements
// inside buckets.  Tiles that begin and end on bucket boundaries and fit LDS are sorted there (bitonic network);
// ranges that contain a bucket too long for a tile are listed for a segmented radix sort.
constexpr uint32_t ROUND_NO_BOUND = 0xffffffffu;
__global__ void k_round_tile_bounds(const uint64_t* __restrict__ keys, uint32_t m, int shift, uint32_t target,
                                    uint32_t limit, uint32_t n_tiles, uint32_t* __restrict__ bound) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) { bound[t] = m; return; }
    uint64_t c = (uint64_t)t * target;
    if (t == 0) { bound[0] = 0; return; }
    const uint64_t stop = c + limit < m ? c + limit : m;
    while (c < stop && (keys[c] >> shift) == (keys[c - 1] >> shift)) c++;
    bound[t] = c >= m ? m : (c == stop ? ROUND_NO_BOUND : (uint32_t)c);      // stop reached inside a long bucket
}
template <int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void k_round_local_sort(const uint64_t* __restrict__ kin,
                                                            const uint32_t* __restrict__ vin,
                                                            uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                            const uint32_t* __restrict__ bound, uint32_t n_tiles,
                                                            uint32_t* __restrict__ big_begin,
                                                            uint32_t* __restrict__ big_end,
                                                            uint32_t* __restrict__ big_count, uint32_t big_cap,
                                                            int shift) {
    __shared__ uint64_t s_k[CAP];
    __shared__ uint32_t s_v[CAP];
    __shared__ uint32_t s_long;
    const uint32_t t = blockIdx.x;
    const uint32_t b = bound[t];
    if (b == ROUND_NO_BOUND) return;                       // this tile starts inside a bucket: an earlier tile owns it
    uint32_t u = t + 1;
    while (bound[u] == ROUND_NO_BOUND) u++;                // bound[n_tiles] = m always ends the search
    const uint32_t e = bound[u];
    if (e <= b) return;
    const uint32_t len = e - b;
    if (len > (uint32_t)CAP) {
        if (threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(big_count, 1u);
            if (slot < big_cap) { big_begin[slot] = b; big_end[slot] = e; }
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < len; i += BLOCK) { s_k[i] = kin[b + i]; s_v[i] = vin[b + i]; }
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    // Buckets are short (a few elements on a pangenome): every element finds its bucket by walking left and right
    // and takes the slot "bucket start + number of bucket elements that sort before it".  A bucket of more than
    // SHORT elements anywhere in the tile sends the whole tile through the bitonic network instead.
    constexpr uint32_t SHORT = 128;
    constexpr int PER = CAP / BLOCK;
    uint32_t slot[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = threadIdx.x + q * BLOCK;
        slot[q] = i;
        if (i < len) {
            const uint64_t ki = s_k[i], hi = ki >> shift;
            uint32_t st = i, steps = 0;
            while (st > 0 && (s_k[st - 1] >> shift) == hi && steps <= SHORT) { st--; steps++; }
            uint32_t before = 0, j = st, cnt = 0;
            while (j < len && cnt <= SHORT) {
                const uint64_t kj = s_k[j];
                if ((kj >> shift) != hi) break;
                before += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
                j++; cnt++;
            }
            if (steps > SHORT || cnt > SHORT) s_long = 1;
            slot[q] = st + before;
        }
    }
    __syncthreads();
    if (!s_long) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t i = threadIdx.x + q * BLOCK;
            if (i < len) { kout[b + slot[q]] = s_k[i]; vout[b + slot[q]] = s_v[i]; }
        }
        return;
    }
    uint32_t P = 64;
    while (P < len) P <<= 1;
    for (uint32_t i = len + threadIdx.x; i < P; i += BLOCK) { s_k[i] = ~0ull; s_v[i] = 0u; }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = s_k[i], c = s_k[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) {
                        s_k[i] = c; s_k[l] = a;
                        const uint32_t va = s_v[i]; s_v[i] = s_v[l]; s_v[l] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < len; i += BLOCK) { kout[b + i] = s_k[i]; vout[b + i] = s_v[i]; }
}
void round_tile_bounds(const uint64_t* keys, uint32_t m, int shift, uint32_t target, uint32_t limit, uint32_t n_tiles,
                       uint32_t* bound, hipStream_t s) {
    hipLaunchKernelGGL(k_round_tile_bounds, dim3(grid_for((uint64_t)n_tiles + 1, 256)), dim3(256), 0, s, keys, m, shift,
                       target, limit, n_tiles, bound);
    MMT_HIP(hipGetLastError());
}
void round_local_sort(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, const uint32_t* bound,
                      uint32_t n_tiles, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count, uint32_t big_cap,
                      int shift, hipStream_t s) {
    hipLaunchKernelGGL((k_round_local_sort<256, (int)ROUND_TILE_CAP>), dim3(n_tiles), dim3(256), 0, s, kin, vin, kout, vout,
                       bound, n_tiles, big_begin, big_end, big_count, big_cap, shift);
    MMT_HIP(hipGetLastError());
}

__global__ void k_mark_subheads(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, uint32_t m,
                                uint32_t* __restrict__ headval) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    headval[c] = (c == 0 || keys[c] != keys[c - 1]) ? pos[c] : 0u;
}
void mark_subheads(const uint64_t* keys, const uint32_t* pos, uint32_t m, uint32_t* headval, hipStream_t s) {
    hipLaunchKernelGGL(k_mark_subheads, dim3(grid_for(m, 256)), dim3(256), 0, s, keys, pos, m, headval);
    MMT_HIP(hipGetLastError());
}

__global__ void k_apply_round(const uint32_t* __restrict__ sa_sorted, const uint32_t* __restrict__ newhead,
                              const uint32_t* __restrict__ pos, uint32_t m, uint32_t* __restrict__ sa,
                              uint32_t* __restrict__ rank, uint8_t* __restrict__ flags) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    uint32_t p = pos[c], i = sa_sorted[c], hd = newhead[c];
    sa[p] = i;
    rank[i] = hd;
    bool single = hd == p && (c + 1 == m || newhead[c + 1] == pos[c + 1]);
    flags[c] = single ? 0 : 1;
}
void apply_round(const uint32_t* sa_sorted, const uint32_t* newhead, const uint32_t* pos, uint32_t m, uint32_t* sa,
                 uint32_t* rank, uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_round, dim3(grid_for(m, 256)), dim3(256), 0, s, sa_sorted, newhead, pos, m, sa, rank,
                       flags);
    MMT_HIP(hipGetLastError());
}

__global__ void k_compact_round(const uint32_t* __restrict__ idx, uint32_t m2, const uint32_t* __restrict__ pos,
                                const uint32_t* __restrict__ sa_sorted, const uint32_t* __restrict__ newhead,
                                uint32_t* __restrict__ out_pos, uint32_t* __restrict__ out_sa,
                                uint32_t* __restrict__ out_head) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m2) return;
    uint32_t o = idx[c];
    out_pos[c] = pos[o]; out_sa[c] = sa_sorted[o]; out_head[c] = newhead[o];
}
void compact_round(const uint32_t* idx, uint32_t m2, const uint32_t* pos, const uint32_t* sa_sorted,
                   const uint32_t* newhead, uint32_t* out_pos, uint32_t* out_sa, uint32_t* out_head, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_round, dim3(grid_for(m2, 256)), dim3(256), 0, s, idx, m2, pos, sa_sorted, newhead,
                       out_pos, out_sa, out_head);
    MMT_HIP(hipGetLastError());
}

// ---- one doubling round in one pass over the active list -----------------------------------------
// The active list (pos, suffix, bucket head) is grouped by bucket, so the tiles between bucket boundaries are
// independent: a workgroup gathers rank[suffix + h] for its tile, sorts (head, rank) in LDS, finds the new
// sub-bucket heads with a scan inside the tile (a tile begins on a bucket boundary: no carry from the left), and
// writes the suffix-array entries, the sorted suffixes, the new heads and the "still tied" flags.  No key ever
// reaches HBM.  The ranks themselves must not change while other tiles still gather them (a tile that saw half of
// a refined bucket would order its elements wrongly), so they are scattered by a second launch (k_scatter_rank over
// the sorted list).  Ranges too long for a tile are listed, their tiles marked, and finished by the k_big_* kernels
// around one segmented radix sort.
__global__ void k_round_head_bounds(const uint32_t* __restrict__ headc, uint32_t m, uint32_t target, uint32_t limit,
                                    uint32_t n_tiles, uint32_t* __restrict__ bound) {
    // one wave per tile: 64 positions per step (a work-item walking its tile alone read up to `limit` heads one after the
    // other -- inside the long buckets of periodic sequence that was 2 ms per round)
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (t > n_tiles) return;
    if (t == n_tiles) { if (lane == 0) bound[t] = m; return; }
    if (t == 0) { if (lane == 0) bound[0] = 0; return; }
    const uint64_t c0 = (uint64_t)t * target;
    const uint64_t stop = c0 + limit < m ? c0 + limit : m;
    uint64_t found = stop;
    for (uint64_t c = c0; c < stop; c += 64) {
        const uint64_t i = c + lane;
        const bool is_bound = i < stop && headc[i] != headc[i - 1];
        const uint64_t hit = __ballot(is_bound);
        if (hit) { found = c + (uint64_t)__builtin_ctzll(hit); break; }
    }
    if (lane == 0) bound[t] = found >= m ? m : (found == stop ? ROUND_NO_BOUND : (uint32_t)found);
}
template <int BLOCK, int CAP>
__global__ __launch_bounds__(BLOCK) void k_round_fused(const uint32_t* __restrict__ sac,
                                                       const uint32_t* __restrict__ headc,
                                                       const uint32_t* __restrict__ pos,
                                                       const uint32_t* __restrict__ bound, uint32_t n_tiles,
                                                       const uint32_t* __restrict__ rank, uint32_t n, uint32_t h,
                                                       int shift, uint32_t* __restrict__ sa,
                                                       uint32_t* __restrict__ sac_out, uint32_t* __restrict__ head_out,
                                                       uint8_t* __restrict__ flags, uint32_t* __restrict__ big_begin,
                                                       uint32_t* __restrict__ big_end, uint32_t* __restrict__ big_count,
                                                       uint32_t big_cap, uint8_t* __restrict__ tile_big) {
    constexpr int PER = CAP / BLOCK, NW = BLOCK / 64;
    static_assert((CAP & (CAP - 1)) == 0, "the bitonic path pads a tile to the next power of two of its length: CAP must be one");
    // bucket head, rank of the suffix h further on (+ 1; 0 past the end), suffix: three 32-bit columns of the tile
    __shared__ uint32_t s_h[CAP];
    __shared__ uint32_t s_r[CAP];
    __shared__ uint32_t s_v[CAP];
    __shared__ uint32_t s_w[PER * NW];
    __shared__ uint32_t s_long;
    const uint32_t t = blockIdx.x;
    const uint32_t b = bound[t];
    if (b == ROUND_NO_BOUND) return;                       // this tile starts inside a bucket: an earlier tile owns it
    uint32_t u = t + 1;
    while (bound[u] == ROUND_NO_BOUND) u++;                // bound[n_tiles] = m always ends the search
    const uint32_t e = bound[u];
    if (e <= b) return;
    const uint32_t len = e - b;
    if (len > (uint32_t)CAP) {
        if (threadIdx.x == 0) {
            const uint32_t slot = atomicAdd(big_count, 1u);
            if (slot < big_cap) { big_begin[slot] = b; big_end[slot] = e; }
        }
        for (uint32_t x = t + threadIdx.x; x < u; x += BLOCK)
            tile_big[x] = (uint8_t)(1u | (x == t ? 2u : 0u) | (x + 1 == u ? 4u : 0u));
        return;
    }
    uint32_t hreg[PER], rreg[PER], vreg[PER], pj[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = threadIdx.x + q * BLOCK;
        hreg[q] = 0; rreg[q] = 0; vreg[q] = 0; pj[q] = 0;
        if (i < len) {
            const uint32_t v = sac[b + i];
            hreg[q] = headc[b + i]; pj[q] = pos[b + i];
            const uint64_t at = (uint64_t)v + h;
            rreg[q] = at < n ? rank[at] + 1u : 0u;           // past the end sorts first (rank + 1 <= n fits 32 bits)
            vreg[q] = v;
            s_h[i] = hreg[q]; s_r[i] = rreg[q]; s_v[i] = v;
        }
    }
    if (threadIdx.x == 0) s_long = 0;
    __syncthreads();
    // short buckets: every element counts the bucket members that sort before it.  A bucket keeps its slots and its
    // elements stand in suffix-array order, so the bucket of slot i begins at slot i - (pos - head): nothing to search.
    constexpr uint32_t SHORT = 128;
    uint32_t slot[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t i = threadIdx.x + q * BLOCK;
        slot[q] = i;
        if (i < len) {
            const uint32_t hi = hreg[q], ri = rreg[q];
            const uint32_t st = i - (pj[q] - hi);
            uint32_t before = 0, j = st, cnt = 0;
            while (j < len && cnt <= SHORT) {
                if (s_h[j] != hi) break;
                const uint32_t rj = s_r[j];
                before += (rj < ri || (rj == ri && j < i)) ? 1u : 0u;
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
            if (i < len) { s_h[slot[q]] = hreg[q]; s_r[slot[q]] = rreg[q]; s_v[slot[q]] = vreg[q]; }
        }
        __syncthreads();
    } else {
        // a bucket beyond SHORT elements somewhere in the tile: bitonic network over (head, rank) pairs
        uint32_t P = 64;
        while (P < len) P <<= 1;
        for (uint32_t i = len + threadIdx.x; i < P; i += BLOCK) { s_h[i] = 0xffffffffu; s_r[i] = 0xffffffffu; s_v[i] = 0u; }
        __syncthreads();
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < P; i += BLOCK) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const uint64_t a = ((uint64_t)s_h[i] << 32) | s_r[i], c = ((uint64_t)s_h[l] << 32) | s_r[l];
                        const bool up = (i & k) == 0;
                        if ((a > c) == up) {
                            s_h[i] = (uint32_t)(c >> 32); s_r[i] = (uint32_t)c; s_h[l] = (uint32_t)(a >> 32); s_r[l] = (uint32_t)a;
                            const uint32_t va = s_v[i]; s_v[i] = s_v[l]; s_v[l] = va;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    // new heads: position of the first element of every run of equal (head, rank), carried along the tile by a running
    // maximum (positions grow along the list); a tile begins with a head, so nothing comes in from the left
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t hv[PER];
    uint8_t tied[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t j = threadIdx.x + q * BLOCK;
        uint32_t x = 0;
        tied[q] = 0;
        if (j < len) {
            const uint32_t hj = s_h[j], rj = s_r[j];
            const bool ish = j == 0 || s_h[j - 1] != hj || s_r[j - 1] != rj;
            const bool nxt = j + 1 == len || s_h[j + 1] != hj || s_r[j + 1] != rj;
            x = ish ? pj[q] : 0u;
            tied[q] = (ish && nxt) ? 0 : 1;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= (uint32_t)o && y > x) x = y; }
        hv[q] = x;
        if (lane == 63) s_w[q * NW + wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint32_t j = threadIdx.x + q * BLOCK;
        if (j < len) {
            uint32_t nh = hv[q];
            for (uint32_t k2 = 0; k2 < (uint32_t)q * NW + wave; k2++) { const uint32_t y = s_w[k2]; nh = y > nh ? y : nh; }
            const uint32_t v = s_v[j];
            sa[pj[q]] = v; sac_out[b + j] = v; head_out[b + j] = nh; flags[b + j] = tied[q];
        }
    }
}
// the chunk of a marked tile: [c0, c1) of the active list (the first chunk of a range begins at its bound, the last one
// ends at the next bound)
__device__ __forceinline__ bool big_chunk(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t x,
                                          uint32_t& c0, uint32_t& c1, bool& first) {
    const uint32_t f = tile_big[x];
    if (!f) return false;
    first = (f & 2u) != 0;
    c0 = first ? bound[x] : x * target;
    c1 = (f & 4u) ? bound[x + 1] : (x + 1) * target;
    return true;
}
__global__ void k_big_keys(const uint8_t* __restrict__ tile_big, const uint32_t* __restrict__ bound, uint32_t target,
                           const uint32_t* __restrict__ sac, const uint32_t* __restrict__ headc,
                           const uint32_t* __restrict__ rank, uint32_t n, uint32_t h, int shift,
                           uint64_t* __restrict__ keys) {
    uint32_t c0, c1; bool first;
    if (!big_chunk(tile_big, bound, target, blockIdx.x, c0, c1, first)) return;
    for (uint32_t c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
        const uint64_t at = (uint64_t)sac[c] + h;
        keys[c] = ((uint64_t)headc[c] << shift) | (at < n ? (uint64_t)rank[at] + 1 : 0);
    }
}
__global__ void k_big_subheads(const uint8_t* __restrict__ tile_big, const uint32_t* __restrict__ bound, uint32_t target,
                               const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                               uint32_t* __restrict__ head) {
    uint32_t c0, c1; bool first;
    if (!big_chunk(tile_big, bound, target, blockIdx.x, c0, c1, first)) return;
    for (uint32_t c = c0 + threadIdx.x; c < c1; c += blockDim.x)
        head[c] = ((first && c == c0) || keys[c] != keys[c - 1]) ? pos[c] : 0u;
}
__global__ void k_big_apply(const uint8_t* __restrict__ tile_big, const uint32_t* __restrict__ bound, uint32_t target,
                            uint32_t m, const uint32_t* __restrict__ sa_sorted, const uint32_t* __restrict__ head,
                            const uint32_t* __restrict__ pos, uint32_t* __restrict__ sa, uint8_t* __restrict__ flags) {
    uint32_t c0, c1; bool first;
    if (!big_chunk(tile_big, bound, target, blockIdx.x, c0, c1, first)) return;
    for (uint32_t c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
        const uint32_t p = pos[c];
        sa[p] = sa_sorted[c];
        const bool single = head[c] == p && (c + 1 == m || head[c + 1] == pos[c + 1]);
        flags[c] = single ? 0 : 1;
    }
}
void round_head_bounds(const uint32_t* headc, uint32_t m, uint32_t target, uint32_t limit, uint32_t n_tiles,
                       uint32_t* bound, hipStream_t s) {
    hipLaunchKernelGGL(k_round_head_bounds, dim3(grid_for(((uint64_t)n_tiles + 1) * 64, 256)), dim3(256), 0, s, headc, m, target,
                       limit, n_tiles, bound);
    MMT_HIP(hipGetLastError());
}
void round_fused(const uint32_t* sac, const uint32_t* headc, const uint32_t* pos, const uint32_t* bound, uint32_t n_tiles,
                 const uint32_t* rank, uint32_t n, uint32_t h, int shift, uint32_t* sa, uint32_t* sac_out,
                 uint32_t* head_out, uint8_t* flags, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count,
                 uint32_t big_cap, uint8_t* tile_big, hipStream_t s) {
    const uint32_t cap = round_fused_cap();
    if (cap == 1024)
        hipLaunchKernelGGL((k_round_fused<256, 1024>), dim3(n_tiles), dim3(256), 0, s, sac, headc, pos, bound, n_tiles, rank, n,
                           h, shift, sa, sac_out, head_out, flags, big_begin, big_end, big_count, big_cap, tile_big);
    else
        hipLaunchKernelGGL((k_round_fused<256, 2048>), dim3(n_tiles), dim3(256), 0, s, sac, headc, pos, bound, n_tiles, rank, n,
                           h, shift, sa, sac_out, head_out, flags, big_begin, big_end, big_count, big_cap, tile_big);
    MMT_HIP(hipGetLastError());
}
// elements of one LDS tile of k_round_fused (MMT_ROUND_CAP = 1024 / 2048: tuning aid; a power of two -- the bitonic path
// of a tile pads to the next power of two of its length and the LDS columns hold CAP entries)
uint32_t round_fused_cap() {
    static const uint32_t cap = [] {
        const char* e = std::getenv("MMT_ROUND_CAP");
        const int v = e ? std::atoi(e) : 2048;
        return (uint32_t)(v == 1024 ? v : 2048);
    }();
    return cap;
}
void round_big_keys(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles, const uint32_t* sac,
                    const uint32_t* headc, const uint32_t* rank, uint32_t n, uint32_t h, int shift, uint64_t* keys,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_big_keys, dim3(n_tiles), dim3(256), 0, s, tile_big, bound, target, sac, headc, rank, n, h, shift,
                       keys);
    MMT_HIP(hipGetLastError());
}
void round_big_subheads(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles,
                        const uint64_t* keys, const uint32_t* pos, uint32_t* head, hipStream_t s) {
    hipLaunchKernelGGL(k_big_subheads, dim3(n_tiles), dim3(256), 0, s, tile_big, bound, target, keys, pos, head);
    MMT_HIP(hipGetLastError());
}
void round_big_apply(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles, uint32_t m,
                     const uint32_t* sa_sorted, const uint32_t* head, const uint32_t* pos, uint32_t* sa, uint8_t* flags,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_big_apply, dim3(n_tiles), dim3(256), 0, s, tile_big, bound, target, m, sa_sorted, head, pos, sa,
                       flags);
    MMT_HIP(hipGetLastError());
}

// 8 bytes from any address (gfx950 + amdhsa: one unaligned global_load_dwordx2)
__device__ __forceinline__ uint64_t load_u64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// ============================================================================
// LCP column: lcp[j] = LCP(suffix sa[j-1], suffix sa[j]), lcp[0] = 0 -- the array
// gsacak returns (direct_gsacak.hpp:62) / pfp_lcp emits (pfp_lcp_mum.hpp:197) --
// WITHOUT the inverse suffix array (the PFP emitter then has no 4-byte random
// store per suffix, and there is no text-order sweep with random stores either).
//   1. k_irr_lcp: in suffix-array order, the entries whose BWT byte differs from
//      the entry before are the irreducible ones; their LCP is computed by plain
//      comparison of the two suffixes (their sum is O(n log n), Karkkainen et
//      al.) and stored at the text position: plcp[sa[j]] = LCP (every other
//      position keeps 0).  Entries are compacted per workgroup first so that
//      every lane compares; matches longer than IRR_STEPS * 8 characters go to a
//      list for k_long_lcp.
//   2. PLCP[i] + i is non-decreasing along the text and PLCP[i] = PLCP[i-1] - 1
//      at every reducible position: PLCP[i] = max_{i' <= i}(plcp[i'] + i') - i,
//      one inclusive max-scan (plcp_running_max; the sums are formed in
//      registers, 64 bits wide when the text is, the column stays 32 bits).
//   3. k_lcp_gather: lcp[j] = PLCP[sa[j]] -- for any range of suffix-array
//      positions, so that a text beyond one LCP column is scanned range by range.
// Also records the suffix ranks of the anchor document (positions < anchor_len),
// which the multi-GPU re-sort needs.
// SA = suffix-array accessor (wide.hpp); its idx_t holds positions and ranks.
// ============================================================================
template <typename I>
struct LongLcpT {
    I p, q; uint32_t h;
    __device__ __forceinline__ uint64_t dst() const { return (uint64_t)p; }      // the value belongs to text position p
    __device__ __forceinline__ uint32_t limit(I n) const {                        // the shorter suffix ends first
        const I room = n - (p > q ? p : q);
        return room < (I)LCP_CAP ? (uint32_t)room : LCP_CAP;
    }
};
// the same with a destination of its own (LCP of adjacent parse suffixes: positions are V indices, values are stored per
// parse position -- kernels.hpp LongLcpDst)
struct LongLcpDstT {
    uint64_t p, q; uint32_t h, d;
    __device__ __forceinline__ uint64_t dst() const { return (uint64_t)d; }
    __device__ __forceinline__ uint32_t limit(uint64_t n) const {
        const uint64_t room = n - (p > q ? p : q);
        return room < (uint64_t)LCP_CAP ? (uint32_t)room : LCP_CAP;
    }
};
// ... and with a limit of its own (dictionary of the parse: a match ends at the terminator of the shorter phrase suffix)
struct LongLcpLimT {
    uint32_t p, q, h, lim;
    __device__ __forceinline__ uint64_t dst() const { return (uint64_t)p; }
    __device__ __forceinline__ uint32_t limit(uint32_t) const { return lim; }
};
constexpr int IRR_STEPS = 24;

// number of characters two suffixes at p and q can share at most, capped (wide.hpp)
template <typename I>
__device__ __forceinline__ uint32_t lcp_limit(I n, I p, I q) {
    const I room = n - (p > q ? p : q);
    return room < (I)LCP_CAP ? (uint32_t)room : LCP_CAP;
}

template <int BLOCK, int PER, typename SA>
__global__ __launch_bounds__(BLOCK) void k_irr_lcp(const uint8_t* __restrict__ text, typename SA::idx_t n, SA sa,
                                                   const uint8_t* __restrict__ bwt, uint32_t* __restrict__ plcp,
                                                   typename SA::idx_t* __restrict__ anchor_rank,
                                                   typename SA::idx_t anchor_len,
                                                   LongLcpT<typename SA::idx_t>* __restrict__ longs,
                                                   uint32_t* __restrict__ long_count, uint32_t long_cap) {
    using I = typename SA::idx_t;
    constexpr int TILE = BLOCK * PER;
    __shared__ uint32_t s_q[TILE];
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * TILE;
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const uint64_t j = base + (uint64_t)q * BLOCK + threadIdx.x;
        bool irr = false;
        if (j < n) {
            const uint8_t b = bwt[j];
            irr = j == 0 || b == 0 || b != bwt[j - 1];
            if (anchor_rank) { const I p = sa.get(j); if (p < anchor_len) anchor_rank[p] = (I)j; }
        }
        const uint64_t m = __ballot(irr);
        uint32_t at = 0;
        if (lane == 0 && m) at = atomicAdd(&s_n, (uint32_t)__popcll(m));
        at = __shfl(at, 0, 64);
        if (irr) s_q[at + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)(j - base);
    }
    __syncthreads();
    const uint32_t cnt = s_n;
    for (uint32_t wbase = 0; wbase < cnt; wbase += BLOCK) {          // uniform trip count: the list slots are handed out per wave
        const uint32_t wi = wbase + threadIdx.x;
        bool queue = false;
        I p = 0, qq = 0;
        uint32_t h = 0;
        if (wi < cnt) {
            const uint64_t j = base + s_q[wi];
            p = sa.get(j);
            if (j != 0) {                                            // j == 0: no predecessor, LCP 0 (the cleared value)
                qq = sa.get(j - 1);
                const uint32_t limit = lcp_limit<I>(n, p, qq);      // the shorter suffix ends first
                bool done = false;
                for (int step = 0; step < IRR_STEPS && h < limit; step++) {
                    const uint64_t x = load_u64(text + p + h), y = load_u64(text + qq + h);
                    if (x != y) { h += (uint32_t)(__builtin_ctzll(x ^ y) >> 3); done = true; break; }
                    h += 8;
                }
                if (h >= limit) { h = limit; done = true; }
                if (done) plcp[p] = h;
                else queue = true;
            }
        }
        // one counter update per wave (1.2 M long matches on the bench workload: one atomic each on a single word was
        // a queue of its own)
        const uint64_t m = __ballot(queue);
        if (m) {
            uint32_t slot0 = 0;
            const int leader = __builtin_ctzll(m);
            if ((int)lane == leader) slot0 = atomicAdd(long_count, (uint32_t)__popcll(m));
            slot0 = __shfl(slot0, leader, 64);
            if (queue) {
                const uint32_t slot = slot0 + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                if (slot < long_cap) { longs[slot].p = p; longs[slot].q = qq; longs[slot].h = h; }
            }
        }
    }
}

// Long matches.  A step of a comparison costs one memory round trip whatever its size, and a match of millions of
// characters (an identical stretch of two haplotypes) is a chain of such steps, so the step grows with the match:
//   k_long_lcp   one wave per match: eight steps of 512 characters (most "long" matches end here), then steps of 4 KB
//                (sixteen 8-byte loads per lane, all in flight before the first use) up to LONG_WAVE_MAX characters;
//   k_huge_lcp   what is still undecided: one workgroup of 16 waves per match, 64 KB per step, the waves agree on
//                the first mismatch through LDS.
constexpr int LONG_UNROLL = 8, HUGE_WAVES = 16;
constexpr uint32_t LONG_SLICE = LONG_UNROLL * 512, LONG_WAVE_MAX = 64 * 1024;

// first mismatch of text[p + base ..) and text[q + base ..) within one 4 KB slice (0xffffffff: none); wave-uniform
template <typename I>
__device__ __forceinline__ uint32_t slice_mismatch(const TextRef& T, I p, I q, uint32_t base,
                                                   uint32_t limit, uint32_t lane) {
    // 8-byte loads at addresses that are multiples of 8 (a misaligned wave-wide load is served lane by lane, ~230 ns per
    // instruction on this part); the suffix bytes are funnelled out of two neighbouring words
    uint64_t x[LONG_UNROLL], y[LONG_UNROLL];
    if (T.v) {
        const uint8_t* const text = T.v;
        const uint8_t* pa = text + (p & ~(I)7);
        const uint8_t* qa = text + (q & ~(I)7);
        const uint32_t sp = (uint32_t)(p & 7u) * 8, sq = (uint32_t)(q & 7u) * 8;
#pragma unroll
        for (int u = 0; u < LONG_UNROLL; u++) {
            const uint32_t o = base + (uint32_t)(u * 64 + lane) * 8;
            const uint32_t oc = o < limit ? o : limit;            // past the shorter suffix: the zero padding after the text
            const uint64_t xl = *reinterpret_cast<const uint64_t*>(pa + oc), xh = *reinterpret_cast<const uint64_t*>(pa + oc + 8);
            const uint64_t yl = *reinterpret_cast<const uint64_t*>(qa + oc), yh = *reinterpret_cast<const uint64_t*>(qa + oc + 8);
            x[u] = sp ? (xl >> sp) | (xh << (64 - sp)) : xl;
            y[u] = sq ? (yl >> sq) | (yh << (64 - sq)) : yl;
        }
    } else {                                                  // packed text (textref.hpp): positions are V indices
#pragma unroll
        for (int u = 0; u < LONG_UNROLL; u++) {
            const uint32_t o = base + (uint32_t)(u * 64 + lane) * 8;
            const uint32_t oc = o < limit ? o : limit;
            x[u] = tx_load8(T, (uint64_t)p + oc); y[u] = tx_load8(T, (uint64_t)q + oc);
        }
    }
    uint32_t first = 0xffffffffu;
#pragma unroll
    for (int u = LONG_UNROLL - 1; u >= 0; u--) {
        const uint32_t o = base + (uint32_t)(u * 64 + lane) * 8;
        const uint64_t d = o < limit ? x[u] ^ y[u] : 0;
        const uint64_t m = __ballot(d != 0);
        if (m) {
            const int fl = __builtin_ctzll(m);
            const uint64_t dx = __shfl(d, fl, 64);
            first = base + (uint32_t)(u * 64 + fl) * 8 + (uint32_t)(__builtin_ctzll(dx) >> 3);
        }
    }
    return first;
}

template <typename I, typename R>
__global__ void k_long_lcp(const TextRef T, I n, R* __restrict__ longs, uint32_t count,
                           uint32_t* __restrict__ plcp, uint32_t* __restrict__ huge_idx, uint32_t* __restrict__ huge_count,
                           uint32_t first) {
    const uint32_t w = first + (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w >= count) return;
    const I p = (I)longs[w].p, q = (I)longs[w].q;
    uint32_t h = longs[w].h;
    const uint32_t limit = longs[w].limit(n);
    const uint32_t stop = limit - h > LONG_WAVE_MAX ? h + LONG_WAVE_MAX : limit;
    bool found = false;
    for (int step = 0; step < 8 && h < limit && !found; step++) {
        const uint32_t o = h + lane * 8;
        uint64_t d = 0;
        if (o < limit) d = tx_load8(T, (uint64_t)p + o) ^ tx_load8(T, (uint64_t)q + o);       // text is zero padded by 64 bytes
        const uint64_t m = __ballot(d != 0);
        if (m) {
            const int fl = __builtin_ctzll(m);
            const uint64_t dx = __shfl(d, fl, 64);
            h += (uint32_t)fl * 8 + (uint32_t)(__builtin_ctzll(dx) >> 3);
            found = true;
        } else h += 512;
    }
    while (!found && h < stop) {
        const uint32_t first = slice_mismatch<I>(T, p, q, h, limit, lane);
        if (first != 0xffffffffu) { h = first; found = true; }
        else h += LONG_SLICE;
    }
    if (found || h >= limit) {
        if (lane == 0) plcp[longs[w].dst()] = h > limit ? limit : h;
    } else if (lane == 0) {
        longs[w].h = h;
        huge_idx[atomicAdd(huge_count, 1u)] = w;
    }
}

template <typename I, typename R>
__global__ __launch_bounds__(HUGE_WAVES * 64) void k_huge_lcp(const TextRef T, I n,
                                                              const R* __restrict__ longs,
                                                              const uint32_t* __restrict__ huge_idx,
                                                              const uint32_t* __restrict__ huge_count,
                                                              uint32_t* __restrict__ plcp) {
    __shared__ uint32_t s_first[HUGE_WAVES];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total = *huge_count;
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {
        const R L = longs[huge_idx[e]];
        const I p = (I)L.p, q = (I)L.q;
        const uint32_t limit = L.limit(n);
        uint32_t h = L.h;
        while (h < limit) {
            // slices past the cap (wide texts only) compare nothing: slice_mismatch clamps every offset to `limit`
            const uint32_t first = slice_mismatch<I>(T, p, q, h + wave * LONG_SLICE, limit, lane);
            if (lane == 0) s_first[wave] = first;
            __syncthreads();
            uint32_t best = 0xffffffffu;
#pragma unroll
            for (int w = 0; w < HUGE_WAVES; w++) best = s_first[w] < best ? s_first[w] : best;
            __syncthreads();
            if (best != 0xffffffffu) { h = best; break; }
            if (limit - h <= HUGE_WAVES * LONG_SLICE) { h = limit; break; }
            h += HUGE_WAVES * LONG_SLICE;
        }
        if (threadIdx.x == 0) plcp[L.dst()] = h > limit ? limit : h;
    }
}

// plcp[i] <- max_{i' <= i}(plcp[i'] + i') - i, in place.  Three passes: workgroup maxima of plcp[i] + i, a small scan of
// those, then the scan proper with the carry-in.  A = type of the sums (uint32_t while n < 2^32, else uint64_t).
template <int BLOCK, int ITEMS, typename A>
__global__ __launch_bounds__(BLOCK) void k_plcp_block_max(const uint32_t* __restrict__ plcp, uint64_t n,
                                                          A* __restrict__ block_max) {
    __shared__ A s_w[BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK * ITEMS;
    A m = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; q++) {                  // coalesced: consecutive threads read consecutive elements
        const uint64_t i = base + (uint64_t)q * BLOCK + threadIdx.x;
        if (i < n) { const A v = (A)plcp[i] + (A)i; m = v > m ? v : m; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const A y = __shfl_xor(m, o, 64); m = y > m ? y : m; }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        A r = 0;
        for (int w = 0; w < BLOCK / 64; w++) r = s_w[w] > r ? s_w[w] : r;
        block_max[blockIdx.x] = r;
    }
}
// exclusive running maximum of the workgroup maxima (a few hundred thousand entries at most): one workgroup
template <int BLOCK, typename A>
__global__ __launch_bounds__(BLOCK) void k_plcp_carry(const A* __restrict__ block_max, uint32_t blocks, A* __restrict__ carry) {
    __shared__ A s_w[BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    A run = 0;
    for (uint32_t b0 = 0; b0 < blocks; b0 += BLOCK) {
        const uint32_t b = b0 + threadIdx.x;
        const A own = b < blocks ? block_max[b] : (A)0;
        A v = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const A y = __shfl_up(v, o, 64); if (lane >= (uint32_t)o && y > v) v = y; }
        if (lane == 63) s_w[wave] = v;
        __syncthreads();
        A pre = run;
        for (uint32_t w2 = 0; w2 < wave; w2++) pre = s_w[w2] > pre ? s_w[w2] : pre;
        A excl = __shfl_up(v, 1, 64);
        if (lane == 0) excl = 0;
        excl = excl > pre ? excl : pre;
        if (b < blocks) carry[b] = excl;
        A tot = run;
        for (int w2 = 0; w2 < BLOCK / 64; w2++) tot = s_w[w2] > tot ? s_w[w2] : tot;
        run = tot;
        __syncthreads();
    }
}
template <int BLOCK, int ITEMS, typename A>
__global__ __launch_bounds__(BLOCK) void k_plcp_scan(uint32_t* plcp, uint64_t n, const A* __restrict__ carry) {
    __shared__ A s_w[BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK * ITEMS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    A run = carry[blockIdx.x];
    for (int q = 0; q < ITEMS; q++) {                  // ITEMS rounds of BLOCK consecutive elements
        const uint64_t i = base + (uint64_t)q * BLOCK + threadIdx.x;
        A v = i < n ? (A)plcp[i] + (A)i : (A)0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const A y = __shfl_up(v, o, 64); if (lane >= (uint32_t)o && y > v) v = y; }
        if (lane == 63) s_w[wave] = v;
        __syncthreads();
        A pre = run;
        for (uint32_t w2 = 0; w2 < wave; w2++) pre = s_w[w2] > pre ? s_w[w2] : pre;
        v = v > pre ? v : pre;
        if (i < n) plcp[i] = (uint32_t)(v - (A)i);
        A tot = run;
        for (int w2 = 0; w2 < BLOCK / 64; w2++) tot = s_w[w2] > tot ? s_w[w2] : tot;
        run = tot;
        __syncthreads();
    }
}
template <typename A>
static void plcp_running_max_typed(uint32_t* plcp, uint64_t n, void* scratch, hipStream_t s) {
    constexpr int BLOCK = 256, ITEMS = 16;
    const uint32_t blocks = (uint32_t)((n + (uint64_t)BLOCK * ITEMS - 1) / ((uint64_t)BLOCK * ITEMS));
    A* bmax = static_cast<A*>(scratch);
    A* carry = bmax + blocks;
    hipLaunchKernelGGL((k_plcp_block_max<BLOCK, ITEMS, A>), dim3(blocks), dim3(BLOCK), 0, s, plcp, n, bmax);
    hipLaunchKernelGGL((k_plcp_carry<1024, A>), dim3(1), dim3(1024), 0, s, bmax, blocks, carry);
    hipLaunchKernelGGL((k_plcp_scan<BLOCK, ITEMS, A>), dim3(blocks), dim3(BLOCK), 0, s, plcp, n, carry);
    MMT_HIP(hipGetLastError());
}
size_t plcp_running_max_scratch(uint64_t n) { return ((n + 4095) / 4096) * 16 + 64; }
void plcp_running_max(uint32_t* plcp, uint64_t n, void* scratch, hipStream_t s) {
    if (!n) return;
    if (n < NARROW_LIMIT) plcp_running_max_typed<uint32_t>(plcp, n, scratch, s);
    else plcp_running_max_typed<uint64_t>(plcp, n, scratch, s);
}

// lcp[t] = PLCP[sa[j0 + t]] for t in [0, count): four consecutive entries per thread (one 16-byte load of SA, four
// independent gathers in flight, one 16-byte store); j0 is a multiple of 4
template <typename SA>
__global__ void k_lcp_gather(const uint32_t* __restrict__ plcp, SA sa, uint64_t j0, uint64_t count,
                             uint32_t* __restrict__ lcp) {
    const uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t + 4 <= count) {
        typename SA::idx_t p[4];
        sa.get4(j0 + t, p);
        uint4 r;
        r.x = plcp[p[0]]; r.y = plcp[p[1]]; r.z = plcp[p[2]]; r.w = plcp[p[3]];
        *reinterpret_cast<uint4*>(lcp + t) = r;
    } else {
        for (uint64_t u = t; u < count; u++) lcp[u] = plcp[sa.get(j0 + u)];
    }
}

template <typename SA>
static void irreducible_lcp_typed(const uint8_t* text, uint64_t n, SaCol sa, const uint8_t* bwt, uint32_t* plcp,
                                  void* anchor_rank, uint64_t anchor_len, void* long_list, uint32_t* long_count,
                                  uint32_t long_cap, hipStream_t s) {
    using I = typename SA::idx_t;
    constexpr int B = 256, PER = 8;
    hipLaunchKernelGGL((k_irr_lcp<B, PER, SA>), dim3(grid_for(n, B * PER)), dim3(B), 0, s, text, (I)n, SA(sa), bwt, plcp,
                       static_cast<I*>(anchor_rank), (I)anchor_len, static_cast<LongLcpT<I>*>(long_list), long_count,
                       long_cap);
}
void irreducible_lcp(const uint8_t* text, uint64_t n, SaCol sa, const uint8_t* bwt, uint32_t* plcp, void* anchor_rank,
                     uint64_t anchor_len, void* long_list, uint32_t* long_count, uint32_t long_cap, hipStream_t s) {
    MMT_HIP(hipMemsetAsync(plcp, 0, (size_t)n * 4, s));
    MMT_HIP(hipMemsetAsync(long_count, 0, 4, s));
    if (sa.wide()) irreducible_lcp_typed<Sa40>(text, n, sa, bwt, plcp, anchor_rank, anchor_len, long_list, long_count, long_cap, s);
    else irreducible_lcp_typed<Sa32>(text, n, sa, bwt, plcp, anchor_rank, anchor_len, long_list, long_count, long_cap, s);
    MMT_HIP(hipGetLastError());
}
size_t long_lcp_record_bytes(bool wide) { return wide ? sizeof(LongLcpT<uint64_t>) : sizeof(LongLcpT<uint32_t>); }
template <typename I, typename R = LongLcpT<I>>
static void long_lcp_typed(const TextRef& text, uint64_t n, void* long_list, uint32_t count, uint32_t* plcp,
                           uint32_t* huge_idx, uint32_t* huge_count, hipStream_t s) {
    // (one wave per record; slices of 2^24 records: a launch may not have 2^32 work-items)
    for (uint32_t first = 0; first < count; first += 1u << 24) {
        const uint32_t part = std::min<uint32_t>(1u << 24, count - first);
        hipLaunchKernelGGL((k_long_lcp<I, R>), dim3(grid_for((uint64_t)part * 64, 256)), dim3(256), 0, s, text, (I)n,
                           static_cast<R*>(long_list), count, plcp, huge_idx, huge_count, first);
    }
    const uint32_t blocks = count < 1024u ? count : 1024u;       // the list is read on the device: no host round trip
    hipLaunchKernelGGL((k_huge_lcp<I, R>), dim3(blocks), dim3(HUGE_WAVES * 64), 0, s, text, (I)n,
                       static_cast<const R*>(long_list), huge_idx, huge_count, plcp);
}
void long_lcp_lim(const uint8_t* text, uint32_t n, void* long_list, uint32_t count, uint32_t* out, uint32_t* huge_idx,
                  uint32_t* huge_count, hipStream_t s) {
    static_assert(sizeof(LongLcpLimT) == sizeof(LongLcpLim), "record layout");
    if (!count) return;
    MMT_HIP(hipMemsetAsync(huge_count, 0, 4, s));
    TextRef T; T.v = text; T.n = n;
    long_lcp_typed<uint32_t, LongLcpLimT>(T, n, long_list, count, out, huge_idx, huge_count, s);
    MMT_HIP(hipGetLastError());
}
void long_lcp_dst(const TextRef& v, uint64_t nv, void* long_list, uint32_t count, uint32_t* out, uint32_t* huge_idx,
                  uint32_t* huge_count, hipStream_t s) {
    static_assert(sizeof(LongLcpDstT) == sizeof(LongLcpDst), "record layout");
    if (!count) return;
    MMT_HIP(hipMemsetAsync(huge_count, 0, 4, s));
    long_lcp_typed<uint64_t, LongLcpDstT>(v, nv, long_list, count, out, huge_idx, huge_count, s);
    MMT_HIP(hipGetLastError());
}
void long_lcp(const uint8_t* text, uint64_t n, bool wide, void* long_list, uint32_t count, uint32_t* plcp,
              uint32_t* huge_idx, uint32_t* huge_count, hipStream_t s) {
    if (!count) return;
    MMT_HIP(hipMemsetAsync(huge_count, 0, 4, s));
    TextRef T; T.v = text; T.n = n;
    if (wide) long_lcp_typed<uint64_t>(T, n, long_list, count, plcp, huge_idx, huge_count, s);
    else long_lcp_typed<uint32_t>(T, n, long_list, count, plcp, huge_idx, huge_count, s);
    MMT_HIP(hipGetLastError());
}
void lcp_gather(const uint32_t* plcp, SaCol sa, uint64_t j0, uint64_t count, uint32_t* lcp, hipStream_t s) {
    if (!count) return;
    if (sa.wide())
        hipLaunchKernelGGL(k_lcp_gather<Sa40>, dim3(grid_for(count, 1024)), dim3(256), 0, s, plcp, Sa40(sa), j0, count, lcp);
    else
        hipLaunchKernelGGL(k_lcp_gather<Sa32>, dim3(grid_for(count, 1024)), dim3(256), 0, s, plcp, Sa32(sa), j0, count, lcp);
    MMT_HIP(hipGetLastError());
}

// rank[p] = j0 + t for every entry t of a piece of the suffix array whose text position p lies in the anchor document
// (the multi-GPU re-sort orders merged rows by the suffix rank of their anchor occurrence)
template <typename SA>
__global__ void k_anchor_ranks(SA sa, uint64_t j0, uint64_t count, typename SA::idx_t anchor_len,
                               typename SA::idx_t* __restrict__ rank) {
    // 16 entries per work-item, consecutive work-items on consecutive entries
    const uint64_t t0 = (uint64_t)blockIdx.x * (blockDim.x * 16) + threadIdx.x;
#pragma unroll 4
    for (int q = 0; q < 16; q++) {
        const uint64_t t = t0 + (uint64_t)q * blockDim.x;
        if (t < count) {
            const typename SA::idx_t p = sa.get(t);
            if (p < anchor_len) rank[p] = (typename SA::idx_t)(j0 + t);
        }
    }
}
void anchor_ranks(SaCol piece, uint64_t j0, uint64_t count, uint64_t anchor_len, void* rank, hipStream_t s) {
    if (!count || !anchor_len) return;
    if (piece.wide())
        hipLaunchKernelGGL(k_anchor_ranks<Sa40>, dim3(grid_for(count, 256 * 16)), dim3(256), 0, s, Sa40(piece), j0, count,
                           (uint64_t)anchor_len, static_cast<uint64_t*>(rank));
    else
        hipLaunchKernelGGL(k_anchor_ranks<Sa32>, dim3(grid_for(count, 256 * 16)), dim3(256), 0, s, Sa32(piece), j0, count,
                           (uint32_t)anchor_len, static_cast<uint32_t*>(rank));
    MMT_HIP(hipGetLastError());
}

// bwt[j] = text[sa[j]-1], 0 for sa[j] = 0 (pfp_lcp_mum.hpp:268, direct_gsacak.hpp:66)
__global__ void k_bwt_from_sa(const uint8_t* __restrict__ text, uint32_t n, const uint32_t* __restrict__ sa,
                              uint8_t* __restrict__ bwt) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t p = sa[j];
    bwt[j] = p ? text[p - 1] : (uint8_t)0;
}
void bwt_from_sa(const uint8_t* text, uint32_t n, const uint32_t* sa, uint8_t* bwt, hipStream_t s) {
    hipLaunchKernelGGL(k_bwt_from_sa, dim3(grid_for(n, 256)), dim3(256), 0, s, text, n, sa, bwt);
    MMT_HIP(hipGetLastError());
}

// ============================================================================
// A5  match scan -- mem_finder::update / update_mems (include/mem_finder.hpp:
//     161-170, 304-355), order-independent form:
//     position j closes every LCP interval [s, j-1] whose value
//     l = min(lcp[s+1..j-1]) satisfies lcp[s] < l and lcp[j] < l.  The thread of
//     j walks left from j-1 keeping the running minimum; every time the minimum
//     drops it has found one interval, longest first -- the pop order of the
//     reference's stack.  Intervals with l < min_len are never produced
//     (:350-353); an interval is produced only if some later entry closes it
//     (no flush, pfp_lcp_mum.hpp:223-230) which holds by construction as j <= n-1.
//     Left-maximality (check_bwt_range, :189-192) = some t in (s, j-1] with
//     bwt[t] != bwt[t-1], accumulated during the same walk.
//     LCP and BWT tiles (+ left halo of `cap` entries) are staged in LDS; the
//     positions that can close something are first compacted into a work queue
//     (one per wave) so that every lane of a wave walks a real candidate.
// ============================================================================
//
// What keeps the kernel off the memory-latency, instruction-issue and atomic limits:
//  * two column buffers per workgroup: the next tile of the workgroup arrives by LDS-DMA
//    (global_load_lds_dwordx4) while the current one is processed; barriers wait for LDS traffic only.
//  * every reportable interval has >= num_distinct entries, so the first
//    w = num_distinct - 1 steps of the walk are replaced by one range-min /
//    range-or query on sparse tables T_k[i] = min(lcp[i .. i+2^k-1]) (and the
//    same with OR over the "BWT changes here" bits), built per tile in LDS by
//    k = floor(log2 w) doubling passes.  A position whose window minimum is not
//    above its own LCP closes nothing and is dropped without any walk; for
//    strict multi-MUMs (cap == num_distinct) the walk is a single step.
//  * candidates are collected in an LDS buffer per workgroup and flushed with
//    ONE global atomic per flush: a single counter word saturates at ~90
//    returning atomics per microsecond on MI355X, which bounded the first
//    version of this kernel (profiles/round1_a).
//  * all table work is done on groups of 4 consecutive entries (16-byte LDS
//    accesses, conflict-free for consecutive lanes); the first min(k,3) levels
//    are fused into one register pass over 12 staged values.
__device__ __forceinline__ uint32_t umin32(uint32_t x, uint32_t y) { return x < y ? x : y; }
__device__ __forceinline__ uint4 umin4(uint4 x, uint4 y) {
    return make_uint4(umin32(x.x, y.x), umin32(x.y, y.y), umin32(x.z, y.z), umin32(x.w, y.w));
}
// entries r..r+3 of the 8 values {lo, hi}; r is uniform over the kernel
__device__ __forceinline__ uint4 shift4(uint4 lo, uint4 hi, uint32_t r) {
    switch (r) {
        case 0: return lo;
        case 1: return make_uint4(lo.y, lo.z, lo.w, hi.x);
        case 2: return make_uint4(lo.z, lo.w, hi.x, hi.y);
        default: return make_uint4(lo.w, hi.x, hi.y, hi.z);
    }
}

// bytes r..r+3 of the 8 bytes {lo, hi}; r is uniform over the kernel
__device__ __forceinline__ uint32_t shift4b(uint32_t lo, uint32_t hi, uint32_t r) {
    return r ? (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * r)) : lo;
}
// the byte before each byte of `cur`, given the word before it
__device__ __forceinline__ uint32_t prev_bytes(uint32_t cur, uint32_t before) { return (cur << 8) | (before >> 24); }

// K0 = min(klev, 3): number of levels fused into the first pass.
// "The BWT bytes of the first w + 1 entries are not all equal" is a second window query -- on a sparse OR-table C
// over the change bytes bwt[t] ^ bwt[t-1], built next to T with the same levels (four bytes per word).
// EXACT: cap == num_distinct and the BWT test applies (strict multi-MUMs, every -k run without merge metadata).
// An interval then has exactly w + 1 entries: the C query moves into phase 1 and the walk disappears -- a queued
// position is a candidate as soon as lcp[s] < l.
// LDS byte address of a pointer into shared memory (what M0 / ds instructions take)
__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 64 lanes x 16 bytes from per-lane global addresses to LDS at (wave-uniform) dst + lane * 16.  hipcc does not
// count this load: it is retired by the explicit vmcnt(0) at the top of the tile that consumes the buffer.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// Workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also drains vmcnt, i.e. the DMA).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DPP moves inside rows of 16 lanes / the wave (gfx9 controls): lanes without a source keep `old`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_move(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

// VH (strict multi-MUMs with 64 <= w < 128, the 94-document shape): the window tables are not built level by level.
// A window of 64 entries is min(suf[i], pre[i + 63]) over blocks of 64 entries (van Herk / Gil-Werman); sixteen lanes x
// four entries are such a block, so pre / suf are DPP scans inside a row of lanes, `pre` makes one trip through LDS and
// the table T_6 is written once.  "A BWT change inside the window" is one comparison with the running maximum of the
// change positions (exclusive, 16 bits per entry), a wave scan + one carry per wave.  One LDS write + two reads + one
// write per group of four entries and three barriers, instead of the fused pass + two doubling passes (nine reads,
// three writes, five barriers) of the level-wise tables.
// ALL (with EXACT): merge metadata -- the intervals whose BWT bytes agree are candidates too (compile-time, so that the
// kernel of the plain strict mode carries none of the dense-list code).
// ONEPASS (general modes without merge metadata -- partial multi-MUMs, multi-MEMs: the configs[4] instantiation): the queue is
// walked ONCE and the intervals go through the workgroup's LDS buffer, flushed with one global atomic when half full, as in
// the exact mode.  The two-pass form below -- count, one allocation per workgroup and tile, write -- pays when every other
// position closes an interval (merge metadata on two haplotypes: 576 M candidates per 1.2 G positions); in these modes one
// position in two thousand does, and the second walk of up to cap - w steps per queued position is saved.  (A slot
// allocation per wave and walk step straight in the global list was tried first: 6.7 M returning atomics on one counter word
// in 60 ms are the word's whole rate -- 75.7 ms instead of 60.1.)
template <int BLOCK, int VG, int K0, int OUT_CAP, bool EXACT, bool VH = false, bool ALL = false, bool ONEPASS = false>
__global__ __launch_bounds__(BLOCK) void k_scan(ScanArgs a, uint32_t halo, uint32_t n_tiles, uint32_t w,
                                                uint32_t klev) {
    constexpr int TILE = BLOCK * VG * 4;
    constexpr int MAXG = VG + 1;                        // groups of 4 per thread incl. halo (halo <= 4 * BLOCK)
    constexpr bool USE_C = EXACT || K0 > 0;             // a window of one entry needs no table for its BWT test
    const uint32_t span = halo + TILE;
    extern __shared__ __align__(16) uint8_t smem[];
    // two column buffers: while one tile is processed the next one of this workgroup lands in the other by LDS-DMA
    uint32_t* const lcp_buf = reinterpret_cast<uint32_t*>(smem);           // 2 x (span + 16)
    uint32_t* const s_T = lcp_buf + 2 * (span + 16);                        // span + 16: T_k[i] = min(lcp[i..i+2^k-1])
    uint8_t* const bwt_buf = reinterpret_cast<uint8_t*>(s_T + span + 16);   // 2 x (16 + span + 16): [16 + i] = bwt[lds_lo + i]
    uint16_t* s_queue = reinterpret_cast<uint16_t*>(bwt_buf + 2 * (span + 32));   // TILE
    uint32_t* s_lcp = lcp_buf;
    uint8_t* s_bwt = bwt_buf;
    Cand* s_out = reinterpret_cast<Cand*>(s_queue + TILE);                  // OUT_CAP
    uint32_t* s_C = reinterpret_cast<uint32_t*>(s_out + OUT_CAP);           // span / 4 + 8 words of 4 change bytes
    uint16_t* s_L = reinterpret_cast<uint16_t*>(s_C);                       // VH: span + 16 running maxima instead
    __shared__ uint32_t s_wtot[(VG + 1) * (BLOCK / 64)];
    __shared__ uint32_t s_wcnt[BLOCK / 64 + 1];
    __shared__ uint32_t s_on;
    if (threadIdx.x == 0) s_on = 0;
    const uint32_t lane = threadIdx.x & 63;
    uint16_t* my_queue = s_queue + (threadIdx.x >> 6) * (VG * 256);       // VG passes x 64 lanes x 4 positions
    const uint32_t wstep = 1u << klev;
    // misalignment of the two query windows; the right one is aligned as soon as its step is a multiple of 4
    const uint32_t rl = (0u - w) & 3u, rr = K0 >= 2 ? 0u : (0u - wstep) & 3u;
    const uint32_t* tbl = s_T;
    bool by_dma = false;
    uint32_t cur = 0;
    const uint32_t wave = threadIdx.x >> 6;

    // T (and C) of one group of 4 entries from 12 staged values: fused levels 0..K0-1
    auto fuse_group = [&](uint32_t g) {
        if (K0 > 0) {
            const uint4* l4 = reinterpret_cast<const uint4*>(s_lcp);
            const uint4 A = l4[g], B = l4[g + 1], C = l4[g + 2];             // entries i..i+11 (tail is padding)
            uint4 r;
            if (K0 == 1) {
                r = make_uint4(umin32(A.x, A.y), umin32(A.y, A.z), umin32(A.z, A.w), umin32(A.w, B.x));
            } else if (K0 == 2) {
                const uint32_t m12 = umin32(A.y, A.z), m23 = umin32(A.z, A.w), m01b = umin32(B.x, B.y);
                r = make_uint4(umin32(umin32(A.x, m12), A.w), umin32(umin32(m12, A.w), B.x),
                               umin32(m23, m01b), umin32(umin32(A.w, m01b), B.z));
            } else {
                const uint32_t mid = umin32(umin32(A.w, B.x), umin32(umin32(B.y, B.z), B.w));  // i+3..i+7
                const uint32_t a12 = umin32(A.y, A.z), c01 = umin32(C.x, C.y);
                r = make_uint4(umin32(umin32(A.x, a12), mid), umin32(umin32(a12, mid), C.x),
                               umin32(umin32(A.z, mid), c01), umin32(umin32(mid, c01), C.z));
            }
            reinterpret_cast<uint4*>(s_T)[g] = r;
        }
        if (USE_C) {
            // change bytes x[t] = bwt[t] ^ bwt[t-1] of entries i..i+11, then OR over windows of 2^K0 bytes
            const uint32_t* b32 = reinterpret_cast<const uint32_t*>(s_bwt + 12) + g;   // word before the group
            const uint32_t wm = b32[0], w0 = b32[1], w1 = b32[2], w2 = b32[3];
            const uint32_t x0 = w0 ^ prev_bytes(w0, wm), x1 = w1 ^ prev_bytes(w1, w0), x2 = w2 ^ prev_bytes(w2, w1);
            uint32_t c;
            if (K0 == 0) c = x0;
            else {
                const uint32_t y0 = x0 | shift4b(x0, x1, 1), y1 = x1 | shift4b(x1, x2, 1);
                if (K0 == 1) c = y0;
                else {
                    const uint32_t z0 = y0 | shift4b(y0, y1, 2);
                    if (K0 == 2) c = z0;
                    else {
                        const uint32_t y2 = x2 | (x2 >> 8);
                        const uint32_t z1 = y1 | shift4b(y1, y2, 2);
                        c = z0 | z1;
                    }
                }
            }
            s_C[g] = c;
        }
    };

    // closers below a.first belong to the range before this one (the columns then start with a left extension that
    // only serves the walks): tiles are counted from the start of the columns, the first ones are skipped
    const uint32_t jmin = a.first > 1u ? a.first : 1u;
    for (uint32_t tile = blockIdx.x + a.first / TILE; tile < n_tiles; tile += gridDim.x, cur ^= 1u) {
        s_lcp = lcp_buf + cur * (span + 16);
        s_bwt = bwt_buf + cur * (span + 32);
        tbl = K0 == 0 ? s_lcp : s_T;                                        // level-0 table is the column itself
        const uint64_t tile0 = (uint64_t)tile * TILE;
        const uint64_t lds_lo = tile0 >= halo ? tile0 - halo : 0;          // first staged index
        const uint32_t shift = (uint32_t)(tile0 - lds_lo);                  // LDS index of tile0
        const uint64_t hi = tile0 + TILE < a.n ? tile0 + TILE : a.n;       // one past last staged index
        // interior tiles (full halo, full length: all but the first and the last few) need no bounds checks
        const bool interior = tile0 >= halo && tile0 >= a.first && tile0 + TILE <= a.n;
        const uint32_t staged = (uint32_t)(hi - lds_lo);
        const uint32_t groups = (staged + 3) >> 2;
        uint32_t qn = 0;                                                    // entries in this wave's queue

        auto tile_body = [&](auto interior_tag) {
            constexpr bool INT = decltype(interior_tag)::value;
            // ---- stage the columns (range starts 16-element aligned: 16-byte loads) ----
            if (INT && by_dma) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of the DMA has landed
            } else {
                const uint4* g4 = reinterpret_cast<const uint4*>(a.lcp + lds_lo);
                uint4* l4 = reinterpret_cast<uint4*>(s_lcp);
                const uint4* b4 = reinterpret_cast<const uint4*>(a.bwt + lds_lo);
                uint4* lb4 = reinterpret_cast<uint4*>(s_bwt + 16);
                if (INT) {
#pragma unroll
                    for (int q = 0; q < VG; q++) l4[threadIdx.x + q * BLOCK] = g4[threadIdx.x + q * BLOCK];
                    if (threadIdx.x < (halo >> 2)) l4[VG * BLOCK + threadIdx.x] = g4[VG * BLOCK + threadIdx.x];
                    for (uint32_t i = threadIdx.x; i < (span >> 4); i += BLOCK) lb4[i] = b4[i];
                } else {
                    const uint32_t vec4 = staged >> 2;
                    for (uint32_t i = threadIdx.x; i < vec4; i += BLOCK) l4[i] = g4[i];
                    for (uint32_t i = (vec4 << 2) + threadIdx.x; i < staged; i += BLOCK) s_lcp[i] = a.lcp[lds_lo + i];
                    const uint32_t vec16 = staged >> 4;
                    for (uint32_t i = threadIdx.x; i < vec16; i += BLOCK) lb4[i] = b4[i];
                    for (uint32_t i = (vec16 << 4) + threadIdx.x; i < staged; i += BLOCK) s_bwt[16 + i] = a.bwt[lds_lo + i];
                }
                if (threadIdx.x == 0) s_bwt[15] = lds_lo ? a.bwt[lds_lo - 1] : (uint8_t)0;
            }
            lds_barrier();
            {   // send the next tile of this workgroup to the other buffer (interior tiles only)
                const uint32_t nt = tile + gridDim.x;
                const uint64_t nt0 = (uint64_t)nt * TILE;
                by_dma = nt < n_tiles && nt0 >= (uint64_t)halo + 16 && nt0 + TILE <= a.n;
                if (by_dma) {
                    const uint64_t nlo = nt0 - halo;
                    const uint32_t other = cur ^ 1u;
                    const uint8_t* gl = reinterpret_cast<const uint8_t*>(a.lcp + nlo);
                    const uint32_t l_dst = lds_offset(lcp_buf + other * (span + 16));
                    const uint32_t l_bytes = span * 4;
                    for (uint32_t off = wave * 1024; off < l_bytes; off += (BLOCK / 64) * 1024)
                        if (off + lane * 16 < l_bytes) glds16(gl + off + lane * 16, l_dst + off);
                    const uint8_t* gb = a.bwt + nlo - 16;                   // [15] = the byte before the staged range
                    const uint32_t b_dst = lds_offset(bwt_buf + other * (span + 32));
                    const uint32_t b_bytes = span + 16;
                    for (uint32_t off = wave * 1024; off < b_bytes; off += (BLOCK / 64) * 1024)
                        if (off + lane * 16 < b_bytes) glds16(gb + off + lane * 16, b_dst + off);
                }
            }
            if (VH) {
                const uint4* l4 = reinterpret_cast<const uint4*>(s_lcp);
                uint4* t4 = reinterpret_cast<uint4*>(s_T);
                const uint32_t* b32 = reinterpret_cast<const uint32_t*>(s_bwt + 12);
                uint4 sufv[MAXG];
                uint32_t cpre[MAXG][3], cexc[MAXG];
#pragma unroll
                for (int q = 0; q < MAXG; q++) {
                    const uint32_t g = threadIdx.x + q * BLOCK;
                    const bool in = INT ? (q < VG || threadIdx.x < (halo >> 2)) : g < groups;
                    const uint4 v = in ? l4[g] : make_uint4(~0u, ~0u, ~0u, ~0u);
                    // prefix minima inside the block of 64: own four entries, then the lanes to the left in the row
                    const uint32_t p1 = umin32(v.x, v.y), p2 = umin32(p1, v.z), gm = umin32(p2, v.w);
                    uint32_t r = gm;
                    r = umin32(r, dpp_move<DPP_ROW_SHR + 1>(~0u, r)); r = umin32(r, dpp_move<DPP_ROW_SHR + 2>(~0u, r));
                    r = umin32(r, dpp_move<DPP_ROW_SHR + 4>(~0u, r)); r = umin32(r, dpp_move<DPP_ROW_SHR + 8>(~0u, r));
                    const uint32_t before = dpp_move<DPP_ROW_SHR + 1>(~0u, r);
                    if (in) t4[g] = make_uint4(umin32(before, v.x), umin32(before, p1), umin32(before, p2), umin32(before, gm));
                    // suffix minima: the lanes to the right
                    const uint32_t s2 = umin32(v.z, v.w), s1 = umin32(v.y, s2);
                    uint32_t l = gm;
                    l = umin32(l, dpp_move<DPP_ROW_SHL + 1>(~0u, l)); l = umin32(l, dpp_move<DPP_ROW_SHL + 2>(~0u, l));
                    l = umin32(l, dpp_move<DPP_ROW_SHL + 4>(~0u, l)); l = umin32(l, dpp_move<DPP_ROW_SHL + 8>(~0u, l));
                    const uint32_t after = dpp_move<DPP_ROW_SHL + 1>(~0u, l);
                    sufv[q] = make_uint4(umin32(after, gm), umin32(after, s1), umin32(after, s2), umin32(after, v.w));
                    // change positions (LDS index + 1, 0 = none) and their running maximum over the wave
                    uint32_t c[4] = {0, 0, 0, 0};
                    if (in) {
                        const uint32_t wm = b32[g], w0 = b32[g + 1];
                        const uint32_t x0 = w0 ^ prev_bytes(w0, wm);
#pragma unroll
                        for (int t = 0; t < 4; t++) c[t] = ((x0 >> (8 * t)) & 0xffu) ? 4u * g + t + 1u : 0u;
                    }
                    const uint32_t c1 = c[0] > c[1] ? c[0] : c[1], c2 = c1 > c[2] ? c1 : c[2], c3 = c2 > c[3] ? c2 : c[3];
                    cpre[q][0] = c[0]; cpre[q][1] = c1; cpre[q][2] = c2;
                    uint32_t m = c3, u;
                    u = dpp_move<DPP_ROW_SHR + 1>(0u, m); m = m > u ? m : u; u = dpp_move<DPP_ROW_SHR + 2>(0u, m); m = m > u ? m : u;
                    u = dpp_move<DPP_ROW_SHR + 4>(0u, m); m = m > u ? m : u; u = dpp_move<DPP_ROW_SHR + 8>(0u, m); m = m > u ? m : u;
                    u = dpp_move<DPP_BCAST15, 0xa>(0u, m); m = m > u ? m : u;
                    u = dpp_move<DPP_BCAST31, 0xc>(0u, m); m = m > u ? m : u;
                    cexc[q] = dpp_move<DPP_WAVE_SHR1>(0u, m);
                    if (lane == 63) s_wtot[q * (BLOCK / 64) + wave] = m;
                }
                lds_barrier();
                uint4 t6[MAXG];
                uint32_t carry[MAXG];
#pragma unroll
                for (int q = 0; q < MAXG; q++) {
                    const uint32_t g = threadIdx.x + q * BLOCK;
                    const bool in = INT ? (q < VG || threadIdx.x < (halo >> 2)) : g < groups;
                    // T_6[i + t] = min(suf[i + t], pre[i + t + 63]); entries whose window leaves the staged range are
                    // never asked for
                    const uint32_t last = ((span + 16) >> 2) - 1;
                    const uint32_t ga = g + 15 < last ? g + 15 : last, gb = g + 16 < last ? g + 16 : last;
                    const uint4 A = t4[ga], B = t4[gb];
                    t6[q] = make_uint4(umin32(sufv[q].x, A.w), umin32(sufv[q].y, B.x), umin32(sufv[q].z, B.y), umin32(sufv[q].w, B.z));
                    uint32_t cy = cexc[q];
#pragma unroll
                    for (int x = 0; x < MAXG * (BLOCK / 64); x++) {
                        const uint32_t tot = s_wtot[x];
                        if ((uint32_t)x < q * (BLOCK / 64) + wave) cy = cy > tot ? cy : tot;
                    }
                    carry[q] = cy;
                    (void)in;
                }
                lds_barrier();
#pragma unroll
                for (int q = 0; q < MAXG; q++) {
                    const uint32_t g = threadIdx.x + q * BLOCK;
                    const bool in = INT ? (q < VG || threadIdx.x < (halo >> 2)) : g < groups;
                    if (in) {
                        t4[g] = t6[q];
                        const uint32_t e0 = carry[q], e1 = e0 > cpre[q][0] ? e0 : cpre[q][0], e2 = e0 > cpre[q][1] ? e0 : cpre[q][1],
                                       e3 = e0 > cpre[q][2] ? e0 : cpre[q][2];
                        reinterpret_cast<uint2*>(s_L)[g] = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
                    }
                }
                lds_barrier();
            }
            // ---- fused levels 0..K0-1 ----
            if (USE_C && !VH) {
                if (INT) {
#pragma unroll
                    for (int q = 0; q < VG; q++) fuse_group(threadIdx.x + q * BLOCK);
                    if (threadIdx.x < (halo >> 2)) fuse_group(VG * BLOCK + threadIdx.x);
                } else {
#pragma unroll
                    for (int q = 0; q < MAXG; q++) {
                        const uint32_t g = threadIdx.x + q * BLOCK;
                        if (g < groups) fuse_group(g);
                    }
                }
                lds_barrier();
            }
            // ---- remaining levels (step >= 8): T[i..i+3] = min(T[i..i+3], T[i+step..i+step+3]) ----
            // Two levels per pass where two are left (four reads, one write, two barriers instead of 2 x (two reads, one
            // write, two barriers)): a window of 93 entries (94 documents, levels 3 -> 5 -> 6) takes two passes, not three.
            for (uint32_t lev = K0; !VH && lev < klev;) {
                const uint32_t gstep = (1u << lev) >> 2;
                const bool two = lev + 2 <= klev;
                uint4* t4 = reinterpret_cast<uint4*>(s_T);
                uint4 rt[MAXG];
                uint32_t rc[MAXG];
#pragma unroll
                for (int q = 0; q < MAXG; q++) {
                    const uint32_t g = threadIdx.x + q * BLOCK;
                    if (g < groups) {
                        const uint32_t last = groups - 1;
                        const uint32_t g2 = g + gstep < groups ? g + gstep : last;
                        rt[q] = umin4(t4[g], t4[g2]);
                        rc[q] = s_C[g] | s_C[g2];
                        if (two) {
                            const uint32_t g3 = g + 2 * gstep < groups ? g + 2 * gstep : last;
                            const uint32_t g4 = g + 3 * gstep < groups ? g + 3 * gstep : last;
                            rt[q] = umin4(rt[q], umin4(t4[g3], t4[g4]));
                            rc[q] |= s_C[g3] | s_C[g4];
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int q = 0; q < MAXG; q++) {
                    const uint32_t g = threadIdx.x + q * BLOCK;
                    if (g < groups) { t4[g] = rt[q]; s_C[g] = rc[q]; }
                }
                lds_barrier();
                lev += two ? 2u : 1u;
            }

            // ---- phase 1: positions whose w-window minimum exceeds their own LCP close something ----
            {
                const uint4* l4 = reinterpret_cast<const uint4*>(s_lcp);
                const uint4* t4 = reinterpret_cast<const uint4*>(tbl);
#pragma unroll
                for (int q = 0; q < VG; q++) {
                    const uint32_t o = (threadIdx.x + q * BLOCK) * 4;               // offset in tile of 4 positions
                    const uint32_t lj = shift + o;                                  // LDS index, multiple of 4
                    uint32_t takes = 0;
                    if (INT || (tile0 + o < a.n && lj + 3 >= w)) {
                        const uint4 closing = l4[lj >> 2];
                        // windows start at lj - w + t (left) and lj - wstep + t (right), t = 0..3;
                        // floor((lj - w) / 4) without going negative: lj + 3 >= w guarantees lj - w >= -3
                        const int32_t s0 = (int32_t)lj - (int32_t)w, s1 = (int32_t)lj - (int32_t)wstep;
                        const int32_t g0 = (INT || s0 >= 0) ? (s0 >> 2) : -1, g1 = (INT || s1 >= 0) ? (s1 >> 2) : -1;
                        const uint4 lo = (INT || g0 >= 0) ? t4[g0] : make_uint4(0, 0, 0, 0);
                        const uint4 lo1 = (INT || g1 >= 0) ? t4[g1] : make_uint4(0, 0, 0, 0);
                        const uint4 M = umin4(shift4(lo, t4[g0 + 1], rl), K0 >= 2 ? lo1 : shift4(lo1, t4[g1 + 1], rr));
                        const uint32_t mv[4] = {M.x, M.y, M.z, M.w}, cv[4] = {closing.x, closing.y, closing.z, closing.w};
                        uint32_t chg4 = 0xffffffffu;
                        if (EXACT && VH) {
                            // a change at an index >= the window start?  (running maximum of index + 1 before position lj + t)
                            const uint2 L2 = reinterpret_cast<const uint2*>(s_L)[lj >> 2];
                            const uint32_t lv[4] = {L2.x & 0xffffu, L2.x >> 16, L2.y & 0xffffu, L2.y >> 16};
                            chg4 = 0;
#pragma unroll
                            for (int t = 0; t < 4; t++) chg4 |= ((int32_t)lv[t] > (int32_t)lj + t - (int32_t)w) ? (0xffu << (8 * t)) : 0u;
                        } else if (EXACT) {
                            const uint32_t cl = (INT || g0 >= 0) ? s_C[g0] : 0u, cr = (INT || g1 >= 0) ? s_C[g1] : 0u;
                            chg4 = shift4b(cl, s_C[g0 + 1], rl) | (K0 >= 2 ? cr : shift4b(cr, s_C[g1 + 1], rr));
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            bool ok = mv[t] > cv[t] && mv[t] >= a.min_len;
                            if (!INT) ok = ok && tile0 + o + t >= jmin && tile0 + o + t < a.n && lj + t >= w;
                            if (EXACT) ok = ok && (ALL || ((chg4 >> (8 * t)) & 0xffu));
                            takes |= ok ? (1u << t) : 0u;
                        }
                    }
                    if (__ballot(takes) == 0) continue;                             // nothing to queue in this wave
                    // compaction into this wave's queue: four ballots, the chained mbcnt of all four gives the
                    // number of taken positions in lower lanes; no LDS atomic, no cross-wave traffic
                    const uint64_t m0 = __ballot(takes & 1u), m1 = __ballot(takes & 2u), m2 = __ballot(takes & 4u),
                                   m3 = __ballot(takes & 8u);
                    uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, qn));
                    at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, at));
                    at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, at));
                    at = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, at));
                    if (takes & 1u) my_queue[at++] = (uint16_t)o;
                    if (takes & 2u) my_queue[at++] = (uint16_t)(o + 1);
                    if (takes & 4u) my_queue[at++] = (uint16_t)(o + 2);
                    if (takes & 8u) my_queue[at] = (uint16_t)(o + 3);
                    qn += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                }
            }
        };
        if (interior) tile_body(std::true_type{}); else tile_body(std::false_type{});
        __builtin_amdgcn_wave_barrier();     // the queue is private to the wave: LDS keeps its writes and reads in order

        // ---- phase 2: every wave drains its own queue ----
        // General modes walk their queue twice: first counting the intervals the walks produce, then -- after ONE slot
        // allocation per wave and tile in the global list -- writing them.  (With merge metadata on two haplotypes every
        // other position closes an interval: 5.9 G candidates per whole-genome partition, and an allocation per wave and
        // step of the walk ran at the rate of one counter word, 220 atomics per microsecond: 1.1 s per partition.)
        uint32_t wave_base = 0, wave_off = 0;
        // (strict multi-MUMs without merge metadata: one position in a thousand is queued and fewer still become candidates --
        // one pass, the few waves that have one ask for a slot as they go)
        constexpr bool rare = (EXACT && !ALL) || ONEPASS;
        for (int pass = rare ? 1 : 0; pass < 2; pass++) {
        uint32_t counted = 0;
        // one interval [s, e] (LDS indices) of value len: counted in the first pass, stored in the second at the wave's
        // next slot (the lanes that call this together share one ballot)
        auto put = [&](bool emit, uint32_t s_idx, uint32_t e_idx, uint32_t len, bool chg) {
            if (pass == 0) { counted += emit ? 1u : 0u; return; }
            if (rare) {
                // into the workgroup's LDS buffer, flushed with one global atomic when half full (a returning global
                // atomic per candidate would hold its wave -- and at the barrier the tile -- for its latency)
                if (emit) {
                    Cand c; c.start = (uint32_t)(lds_lo + s_idx); c.end = (uint32_t)(lds_lo + e_idx); c.len = len;
                    c.flags = (!ONEPASS || chg) ? CAND_LEFT_MAXIMAL : 0u;
                    const uint32_t slot = atomicAdd(&s_on, 1u);
                    if (slot < OUT_CAP) s_out[slot] = c;
                    else { const uint32_t g = atomicAdd(a.d_count, 1u); if (g < a.capacity) a.out[g] = c; }
                }
                return;
            }
            const uint64_t em = __ballot(emit);
            if (emit) {
                const uint32_t g = wave_base + wave_off + (uint32_t)__popcll(em & ((1ull << lane) - 1));
                if (g < a.capacity) {
                    Cand c; c.start = (uint32_t)(lds_lo + s_idx); c.end = (uint32_t)(lds_lo + e_idx); c.len = len;
                    c.flags = chg ? CAND_LEFT_MAXIMAL : 0u;
                    a.out[g] = c;
                }
            }
            wave_off += (uint32_t)__popcll(em);
        };
        auto drain = [&](uint32_t wi) {
            const uint32_t o = my_queue[wi];
            const uint32_t lj = shift + o;
            const uint32_t closing = s_lcp[lj];
            uint32_t m = umin32(tbl[lj - w], tbl[lj - wstep]);
            uint32_t lk = lj - w;                                           // LDS index of k; candidate start s = k - 1
            if (EXACT) {
                // the only interval this position can close has w + 1 entries; without merge metadata phase 1 queued it
                // only if its BWT bytes differ, with it (emit_all) that is looked up here for the flag
                bool chg = true;
                if (ALL) {
                    if (VH) chg = (int32_t)s_L[lj] > (int32_t)lj - (int32_t)w;
                    else chg = (reinterpret_cast<const uint8_t*>(s_C)[lj - w] | reinterpret_cast<const uint8_t*>(s_C)[lj - wstep]) != 0;
                }
                put(lk > 0 && s_lcp[lk - 1] < m, lk - 1, lj - 1, m, chg);
                return;
            }
            // general case: finish the walk from s = j - w - 1 leftwards
            // a BWT change among the entries k .. j-1?  Two lookups in the window table of the change bytes (the walk
            // below adds the entries further left one by one)
            const uint8_t* c8 = reinterpret_cast<const uint8_t*>(s_C);
            bool chg;
            if (VH) chg = (int32_t)s_L[lj] > (int32_t)lj - (int32_t)w;       // (the running maximum of the change positions: as in the exact mode)
            else chg = USE_C ? (c8[lj - w] | c8[lj - wstep]) != 0 : s_bwt[16 + lj - 1] != s_bwt[16 + lj - 2];
            bool done = false;
            while (lk > 0) {
                const uint32_t v = s_lcp[lk - 1];
                chg |= s_bwt[16 + lk] != s_bwt[16 + lk - 1];
                const uint32_t cnt = lj - lk + 1;
                const bool emit = v < m && cnt >= a.num_distinct && (a.cap == 0 || cnt <= a.cap) && (chg || a.emit_all);
                put(emit, lk - 1, lj - 1, m, chg);                          // (the lanes still walking take part)
                if (v < m) {
                    m = v;
                    if (m <= closing || m < a.min_len) { done = true; break; }
                }
                lk--;
                if (a.cap && lj - lk + 1 > a.cap) { done = true; break; }  // every further interval is too big
            }
            if (pass == 0) return;
            if (!done && lds_lo > 0) {
                // left the staged halo (uncapped modes / very large caps): continue in the global columns
                const uint64_t j = lds_lo + lj;
                uint64_t kpos = lds_lo;                                     // k = lds_lo, candidate start k - 1
                bool fin = false;
                while (kpos > 0) {
                    const uint32_t v = a.lcp[kpos - 1];
                    chg |= a.bwt[kpos] != a.bwt[kpos - 1];
                    if (v < m) {
                        const uint64_t cnt = j - (kpos - 1);
                        if (cnt >= a.num_distinct && (a.cap == 0 || cnt <= a.cap) && (chg || a.emit_all)) {
                            Cand c; c.start = (uint32_t)(kpos - 1); c.end = (uint32_t)(j - 1); c.len = m;
                            c.flags = chg ? CAND_LEFT_MAXIMAL : 0u;
                            uint32_t g = atomicAdd(a.d_count, 1u);
                            if (g < a.capacity) a.out[g] = c;
                        }
                        m = v;
                        if (m <= closing || m < a.min_len) { fin = true; break; }
                    }
                    kpos--;
                    if (a.cap && j - (kpos - 1) > a.cap) { fin = true; break; }
                }
                // ran off the left extension of a range that does not start the stream: the host repeats the range
                if (!fin && a.more_left) atomicAdd(a.d_count + 4, 1u);
            } else if (!done && a.more_left) atomicAdd(a.d_count + 4, 1u);
        };
        for (uint32_t w0 = 0; w0 < qn; w0 += 64) {               // uniform over the wave: the lanes meet again after each entry
            if (w0 + lane < qn) drain(w0 + lane);
            if (pass == 1 && !rare) {
                // the lane that walked longest saw every allocation of this round
#pragma unroll
                for (int x = 32; x >= 1; x >>= 1) { const uint32_t y = __shfl_xor(wave_off, x, 64); wave_off = y > wave_off ? y : wave_off; }
            }
        }
        if (pass == 0) {
            // the workgroup's share of the global list: one atomic per tile (the counter word takes about 200 of them
            // per microsecond, and a tile per wave would be four times as many)
#pragma unroll
            for (int x = 32; x >= 1; x >>= 1) counted += __shfl_xor(counted, x, 64);
            {
                if (lane == 0) s_wcnt[wave] = counted;
                lds_barrier();
                if (threadIdx.x == 0) {
                    uint32_t tot = 0;
                    for (int x = 0; x < BLOCK / 64; x++) tot += s_wcnt[x];
                    s_wcnt[BLOCK / 64] = tot ? atomicAdd(a.d_count, tot) : 0u;
                }
                lds_barrier();
                wave_base = s_wcnt[BLOCK / 64];
                for (uint32_t x = 0; x < wave; x++) wave_base += s_wcnt[x];
            }
            wave_off = 0;
        }
        }
        lds_barrier();                        // the tables of this tile are dead: the next tile of the workgroup may build its own
        if (rare) {
            const uint32_t filled = s_on < OUT_CAP ? s_on : OUT_CAP;
            const bool last = tile + gridDim.x >= n_tiles;
            if (filled >= OUT_CAP / 2 || (last && filled)) {
                if (threadIdx.x == 0) s_wcnt[BLOCK / 64] = atomicAdd(a.d_count, filled);
                lds_barrier();
                const uint32_t base = s_wcnt[BLOCK / 64];
                for (uint32_t i = threadIdx.x; i < filled; i += BLOCK)
                    if (base + i < a.capacity) a.out[base + i] = s_out[i];
                lds_barrier();
                if (threadIdx.x == 0) s_on = 0;
                lds_barrier();
            }
        }
    }
}

// ---- A5 for windows that do not fit an LDS tile (more than ~1000 documents) ---------------------------------------
// van Herk / Gil-Werman sliding minimum: cut the column into blocks of w entries; with pre[i] = min(lcp[block start .. i])
// and suf[i] = min(lcp[i .. block end]), min(lcp[k .. k+w-1]) = min(suf[k], pre[k+w-1]) for every k, whatever w is.
// One workgroup per block of w entries, a forward and a backward pass of workgroup-wide running minima.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_window_block_mins(const uint32_t* __restrict__ lcp, uint32_t n, uint32_t w,
                                                             uint32_t* __restrict__ pre, uint32_t* __restrict__ suf) {
    __shared__ uint32_t s_wave[BLOCK / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * w;
    const uint64_t e = b0 + w < n ? b0 + w : n;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int dir = 0; dir < 2; dir++) {
        uint32_t carry = 0xffffffffu;
        for (uint64_t c = 0; b0 + c < e; c += BLOCK) {
            const uint64_t t = c + threadIdx.x;                              // distance from the end the pass starts at
            const bool in = b0 + t < e;
            const uint64_t i = dir == 0 ? b0 + t : e - 1 - t;
            uint32_t v = in ? lcp[i] : 0xffffffffu;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(v, o, 64); if (lane >= (uint32_t)o) v = umin32(v, u); }
            if (lane == 63) s_wave[wave] = v;
            __syncthreads();
            uint32_t before = carry, all = carry;
#pragma unroll
            for (int x = 0; x < BLOCK / 64; x++) { if ((uint32_t)x < wave) before = umin32(before, s_wave[x]); all = umin32(all, s_wave[x]); }
            v = umin32(v, before);
            if (in) (dir == 0 ? pre : suf)[i] = v;
            __syncthreads();
            carry = all;
        }
    }
}

// chg[t] = t where the BWT byte differs from the one before (and at 0), else 0: its running maximum is the position of
// the last change at or before t, and "the BWT bytes of entries k-1 .. j-1 are not all equal" = lastchg[j-1] >= k
__global__ void k_mark_bwt_changes(const uint8_t* __restrict__ bwt, uint32_t n, uint32_t* __restrict__ chg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    chg[t] = (t > 0 && bwt[t] != bwt[t - 1]) ? t : 0u;
}

// One thread per closing position j; the window query is two loads, the rest of the walk (modes other than the
// exact-window one) runs in the cached global columns exactly like the tail of k_scan's phase 2.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_scan_wide(ScanArgs a, uint32_t w, int exact) {
    const uint64_t j64 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    bool emit = false;
    Cand c{};
    if (j64 > w && j64 >= a.first && j64 < a.n) {             // the candidate start k - 1 = j - w - 1 must exist
        const uint32_t j = (uint32_t)j64, k = j - w;
        uint32_t m = umin32(a.wide_suf[k], a.wide_pre[j - 1]);
        const uint32_t closing = a.lcp[j];
        if (m > closing && m >= a.min_len) {
            bool chg = a.wide_chg[j - 1] >= k;
            if (exact) {
                if (chg && a.lcp[k - 1] < m) {
                    emit = true;
                    c.start = k - 1; c.end = j - 1; c.len = m; c.flags = CAND_LEFT_MAXIMAL;
                }
            } else {
                uint32_t kpos = k;
                bool fin = false;
                while (kpos > 0) {
                    const uint32_t v = a.lcp[kpos - 1];
                    chg |= a.bwt[kpos] != a.bwt[kpos - 1];
                    if (v < m) {
                        const uint32_t cnt = j - (kpos - 1);
                        if (cnt >= a.num_distinct && (a.cap == 0 || cnt <= a.cap) && (chg || a.emit_all)) {
                            Cand d; d.start = kpos - 1; d.end = j - 1; d.len = m; d.flags = chg ? CAND_LEFT_MAXIMAL : 0u;
                            const uint32_t g = atomicAdd(a.d_count, 1u);
                            if (g < a.capacity) a.out[g] = d;
                        }
                        m = v;
                        if (m <= closing || m < a.min_len) { fin = true; break; }
                    }
                    kpos--;
                    if (a.cap && j - (kpos - 1) > a.cap) { fin = true; break; }
                }
                if (!fin && a.more_left) atomicAdd(a.d_count + 4, 1u);
            }
        }
    }
    const uint64_t mask = __ballot(emit);                     // one counter update per wave
    if (mask) {
        uint32_t base = 0;
        const int leader = __builtin_ctzll(mask);
        if ((int)lane == leader) base = atomicAdd(a.d_count, (uint32_t)__popcll(mask));
        base = __shfl(base, leader, 64);
        if (emit) {
            const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1));
            if (slot < a.capacity) a.out[slot] = c;
        }
    }
}

static uint32_t wide_threshold() {
    static const uint32_t t = getenv("MMT_SCAN_WIDE_AT") ? (uint32_t)atoi(getenv("MMT_SCAN_WIDE_AT")) : 1000u;
    return t;
}
bool scan_needs_wide_docs(size_t n_docs) { return n_docs > (size_t)wide_threshold() + 1; }
bool scan_needs_wide(const ScanArgs& a) {
    const uint32_t nd = a.num_distinct < 2 ? 2 : a.num_distinct;
    return nd - 1 > wide_threshold();
}
void scan_wide_prepare(const uint32_t* lcp, const uint8_t* bwt, uint32_t n, uint32_t num_distinct, uint32_t* pre,
                       uint32_t* suf, uint32_t* chg, hipStream_t s) {
    const uint32_t w = (num_distinct < 2 ? 2 : num_distinct) - 1;
    hipLaunchKernelGGL((k_window_block_mins<256>), dim3(grid_for(n, w)), dim3(256), 0, s, lcp, n, w, pre, suf);
    hipLaunchKernelGGL(k_mark_bwt_changes, dim3(grid_for(n, 256)), dim3(256), 0, s, bwt, n, chg);
    MMT_HIP(hipGetLastError());
}

template <int B, int VG, int OUT_CAP, int VH_OUT = 0>
static void launch_scan(const ScanArgs& a, hipStream_t s, unsigned blocks_per_cu) {
    constexpr int TILE = B * VG * 4;
    uint32_t nd = a.num_distinct < 2 ? 2 : a.num_distinct;
    uint32_t w = nd - 1;
    // intervals of exactly nd entries (strict multi-MUMs; with merge metadata the ones whose BWT bytes agree count too)
    bool exact = a.cap != 0 && a.cap == nd && nd == a.num_distinct;
    if (w > 1000) { w = 1; exact = false; }                // window tables need w <= halo <= 4 * BLOCK
    uint32_t klev = 0;
    while ((2u << klev) <= w) klev++;                      // floor(log2(w))
    uint32_t halo = a.cap ? a.cap + 1 : 256;
    if (halo < w + 1) halo = w + 1;
    halo = (halo + 15) & ~15u;
    if (halo > 4 * B) halo = 4 * B;                        // beyond this the walk reads the cached global columns
    if (halo < w + 1) exact = false;
    // the 94-document shape (64 <= w < 128, exact windows): block-wise window minima instead of level-wise tables
    static const bool no_vh = std::getenv("MMT_SCAN_NO_VH") != nullptr;
    // (... and the general modes with such a window -- 94 documents, -k -1: the configs[4] instantiation -- take the same tables)
    static const bool two_pass = std::getenv("MMT_SCAN_TWO_PASS") != nullptr;     // tests: the general modes' two-pass form
    const bool onepass = !exact && !a.emit_all && !two_pass;
    const bool vh = (exact || onepass) && klev == 6 && halo >= w + 1 && !no_vh && VH_OUT > 0;
    const size_t out_cap = vh ? (size_t)VH_OUT : (size_t)OUT_CAP;
    size_t lds = (size_t)(halo + TILE + 16) * 12 + (size_t)(halo + TILE + 32) * 2 + (size_t)TILE * 2 +
                 out_cap * sizeof(Cand) + (vh ? (size_t)(halo + TILE + 16) * 2 : (size_t)(halo + TILE + 32));
    uint32_t n_tiles = grid_for(a.n, TILE);
    const uint32_t todo = n_tiles - a.first / TILE;        // tiles below a.first are skipped (a.first < a.n)
    unsigned grid = todo < 256u * blocks_per_cu ? (todo ? todo : 1u) : 256u * blocks_per_cu;
    dim3 g(grid), b(B);
    auto go = [&](auto kernel) {
        MMT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        hipLaunchKernelGGL(kernel, g, b, lds, s, a, halo, n_tiles, w, klev);
    };
    const uint32_t k0 = klev < 3 ? klev : 3;
    if (vh && onepass) {
        go(k_scan<B, VG, 3, (VH_OUT > 0 ? VH_OUT : OUT_CAP), false, true, false, true>);
    } else if (vh && a.emit_all) {
        go(k_scan<B, VG, 3, (VH_OUT > 0 ? VH_OUT : OUT_CAP), true, true, true>);
    } else if (vh) {
        go(k_scan<B, VG, 3, (VH_OUT > 0 ? VH_OUT : OUT_CAP), true, true>);
    } else if (exact && a.emit_all) {
        switch (k0) {
            case 0: go(k_scan<B, VG, 0, OUT_CAP, true, false, true>); break;
            case 1: go(k_scan<B, VG, 1, OUT_CAP, true, false, true>); break;
            case 2: go(k_scan<B, VG, 2, OUT_CAP, true, false, true>); break;
            default: go(k_scan<B, VG, 3, OUT_CAP, true, false, true>); break;
        }
    } else if (exact) {
        switch (k0) {
            case 0: go(k_scan<B, VG, 0, OUT_CAP, true>); break;
            case 1: go(k_scan<B, VG, 1, OUT_CAP, true>); break;
            case 2: go(k_scan<B, VG, 2, OUT_CAP, true>); break;
            default: go(k_scan<B, VG, 3, OUT_CAP, true>); break;
        }
    } else if (onepass) {
        switch (k0) {
            case 0: go(k_scan<B, VG, 0, OUT_CAP, false, false, false, true>); break;
            case 1: go(k_scan<B, VG, 1, OUT_CAP, false, false, false, true>); break;
            case 2: go(k_scan<B, VG, 2, OUT_CAP, false, false, false, true>); break;
            default: go(k_scan<B, VG, 3, OUT_CAP, false, false, false, true>); break;
        }
    } else {
        switch (k0) {
            case 0: go(k_scan<B, VG, 0, OUT_CAP, false>); break;
            case 1: go(k_scan<B, VG, 1, OUT_CAP, false>); break;
            case 2: go(k_scan<B, VG, 2, OUT_CAP, false>); break;
            default: go(k_scan<B, VG, 3, OUT_CAP, false>); break;
        }
    }
    MMT_HIP(hipGetLastError());
}

void scan_intervals(const ScanArgs& a, hipStream_t s) {
    if (a.cap && a.cap < a.num_distinct) return;          // no interval can satisfy both bounds
    if (scan_needs_wide(a)) {
        if (!a.wide_pre || !a.wide_suf || !a.wide_chg) throw HipError("scan: the window tables of the wide path are missing");
        const uint32_t nd = a.num_distinct < 2 ? 2 : a.num_distinct;
        const bool exact = a.cap != 0 && a.cap == nd && !a.emit_all && nd == a.num_distinct;
        hipLaunchKernelGGL((k_scan_wide<256>), dim3(grid_for(a.n, 256)), dim3(256), 0, s, a, nd - 1, exact ? 1 : 0);
        MMT_HIP(hipGetLastError());
        return;
    }
    // measured on MI355X (tests/scan_sweep.sh, profiles/): workgroups of 256 threads x 8 positions (39.7 KB of LDS
    // with both column buffers: four resident workgroups per CU) and a grid of 16 per CU are best for 16 and for
    // 94 documents alike; 512 x 8 is 5 % slower, 512 x 12 leaves one workgroup per CU.
    static int variant = -1, bpc = 16;
    if (variant < 0) {
        const char* v = getenv("MMT_SCAN_VARIANT"); variant = v ? atoi(v) : 0;
        const char* g = getenv("MMT_SCAN_BPC"); if (g) bpc = atoi(g);
    }
    switch (variant) {
        case 1: launch_scan<512, 1, 256>(a, s, bpc); break;
        case 2: launch_scan<512, 2, 256>(a, s, bpc); break;
        default: launch_scan<256, 2, 256, 128>(a, s, bpc); break;
    }
}

// ============================================================================
// Candidate verification -- check_doc_range (mem_finder.hpp:265-289), the
// threshold record of update_mems (:326-336) and the row decision (:338-343).
// One wave per candidate; document ids are derived from SA by binary search
// over the document starts (rank of doc_ends, pfp_lcp_mum.hpp:194).
// ============================================================================
__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_min32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { uint32_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Candidates carry positions relative to the scanned range (a.base = suffix-array index of its entry 0; a.lcp is the
// LCP column of the range); accepted rows leave with absolute positions.
template <typename SA>
struct VerifyArgsT {
    const Cand* cand; uint32_t n_cand; SA sa; uint64_t base, sa_off; const uint32_t* lcp; const uint64_t* d_doc_start;
    uint32_t n_docs, num_distinct, max_doc_freq; int merge; uint32_t* thresh; Row* rows; uint32_t* d_row_count;
};
__device__ __forceinline__ Row make_row(const Cand& c, uint64_t base) {
    Row r; r.start = base + c.start; r.cnt = c.end - c.start + 1; r.len = c.len; return r;
}
template <int WAVES, typename SA>
__global__ __launch_bounds__(WAVES * 64) void k_verify(VerifyArgsT<SA> a, int use_counters) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* ctr = reinterpret_cast<uint32_t*>(smem) + (size_t)(threadIdx.x >> 6) * a.n_docs;
    __shared__ Row s_rows[WAVES][64];       // accepted rows of this wave, flushed 64 at a time
    Row* my_rows = s_rows[threadIdx.x >> 6];
    uint32_t n_my = 0;                        // wave-uniform
    const uint32_t lane = threadIdx.x & 63;
    if (use_counters) {
        for (uint32_t i = lane; i < a.n_docs; i += 64) ctr[i] = 0;
    }
    const uint64_t wave = (uint64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES;
    for (uint64_t ci = wave; ci < a.n_cand; ci += n_waves) {
        const Cand c = a.cand[ci];
        const uint32_t cnt = c.end - c.start + 1;
        bool ok;
        uint32_t first0 = 0xffffffffu;
        if (!use_counters) {
            // <= 64 documents, at most one occurrence each, interval fits one wave
            uint64_t bit = 0;
            if (lane < cnt) {
                uint32_t d = doc_lookup(a.d_doc_start, a.n_docs, a.sa.get(a.sa_off + c.start + lane));
                bit = 1ull << d;
                if (d == 0) first0 = c.start + lane;
            }
            uint64_t mask = wave_or64(bit);
            ok = (uint32_t)__popcll(mask) == cnt && cnt >= a.num_distinct;
        } else {
            uint32_t uniq = 0, fail = 0;
            for (uint32_t base = 0; base < cnt; base += 64) {
                if (base + lane < cnt) {
                    uint32_t kk = c.start + base + lane;
                    uint32_t d = doc_lookup(a.d_doc_start, a.n_docs, a.sa.get(a.sa_off + kk));
                    if (d >= a.n_docs) { fail = 1; }
                    else {
                        uint32_t old = atomicAdd(&ctr[d], 1u);
                        if (old == 0) uniq++;
                        if (a.max_doc_freq && old + 1 > a.max_doc_freq) fail = 1;
                        if (d == 0 && kk < first0) first0 = kk;
                    }
                }
            }
            uniq = wave_sum32(uniq);
            fail = wave_sum32(fail);
            for (uint32_t base = 0; base < cnt; base += 64) {               // undo the counters
                if (base + lane < cnt) {
                    uint32_t d = doc_lookup(a.d_doc_start, a.n_docs, a.sa.get(a.sa_off + c.start + base + lane));
                    if (d < a.n_docs) ctr[d] = 0;
                }
            }
            ok = fail == 0 && uniq >= a.num_distinct;
        }
        if (!ok) continue;
        if (a.merge) {
            first0 = wave_min32(first0);
            if (lane == 0 && first0 != 0xffffffffu) {
                uint32_t before = a.lcp[c.start], after = a.lcp[c.end + 1];
                // (the reference's column saturates at 65535, mem_finder.hpp:299,328; the width here is 32 bits and the
                // saturation happens where a 16-bit file or table is written: Engine::thresh_device / copy_thresh)
                const uint32_t nb = before > after ? before : after;
                a.thresh[(uint64_t)a.sa.get(a.sa_off + first0) - a.d_doc_start[0]] = nb;
            }
        }
        if (c.flags & CAND_LEFT_MAXIMAL) {
            if (lane == 0) my_rows[n_my] = make_row(c, a.base);
            n_my++;
            if (n_my == 64) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.d_row_count, 64u);
                base = __shfl(base, 0, 64);
                a.rows[base + lane] = my_rows[lane];
                n_my = 0;
            }
        }
    }
    if (n_my) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.d_row_count, n_my);
        base = __shfl(base, 0, 64);
        if (lane < n_my) a.rows[base + lane] = my_rows[lane];
    }
}

// The same for at most 32 documents with at most one occurrence each: an interval then has at most SUB entries, so a wave
// takes 64 / SUB candidates at a time, one per group of SUB lanes (document bitmap and first anchor entry by shuffles of
// width SUB).  The merge-metadata runs of the multi-GPU path verify every structural interval, ten times the
// candidates of a plain run.
template <int WAVES, int SUB, typename SA>
__global__ __launch_bounds__(WAVES * 64) void k_verify_packed(VerifyArgsT<SA> a) {
    constexpr int PER = 64 / SUB;
    __shared__ Row s_rows[WAVES][64];       // accepted rows of this wave
    Row* my_rows = s_rows[threadIdx.x >> 6];
    uint32_t n_my = 0;                        // wave-uniform
    const uint32_t lane = threadIdx.x & 63, sl = lane % SUB, grp = lane / SUB;
    const uint64_t wave = (uint64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * WAVES;
    auto flush = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.d_row_count, n_my);
        base = __shfl(base, 0, 64);
        if (lane < n_my) a.rows[base + lane] = my_rows[lane];
        n_my = 0;
    };
    for (uint64_t c0 = wave * PER; c0 < a.n_cand; c0 += n_waves * PER) {
        const uint64_t ci = c0 + grp;
        const bool live = ci < a.n_cand;
        Cand c{};
        if (live) c = a.cand[ci];
        const uint32_t cnt = live ? c.end - c.start + 1 : 0u;
        uint32_t bits = 0, first0 = 0xffffffffu;
        if (sl < cnt) {
            const uint32_t d = doc_lookup(a.d_doc_start, a.n_docs, a.sa.get(a.sa_off + c.start + sl));
            bits = 1u << d;
            if (d == 0) first0 = c.start + sl;
        }
#pragma unroll
        for (int o = SUB / 2; o >= 1; o >>= 1) {
            bits |= __shfl_xor(bits, o, SUB);
            const uint32_t f = __shfl_xor(first0, o, SUB);
            first0 = f < first0 ? f : first0;
        }
        // an interval longer than SUB cannot hold distinct documents only: its count exceeds the bitmap
        const bool ok = live && (uint32_t)__popc(bits) == cnt && cnt >= a.num_distinct;
        if (a.merge && ok && sl == 0 && first0 != 0xffffffffu) {
            const uint32_t before = a.lcp[c.start], after = a.lcp[c.end + 1];
            const uint32_t nb = before > after ? before : after;
            a.thresh[(uint64_t)a.sa.get(a.sa_off + first0) - a.d_doc_start[0]] = nb;
        }
        const bool take = ok && sl == 0 && (c.flags & CAND_LEFT_MAXIMAL);
        const uint64_t m = __ballot(take);
        if (m) {
            const uint32_t k = (uint32_t)__popcll(m);
            if (n_my + k > 64) flush();
            if (take) my_rows[n_my + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = make_row(c, a.base);
            n_my += k;
        }
    }
    if (n_my) flush();
}

template <typename SA>
static void verify_typed(const VerifyArgs& v, hipStream_t s) {
    VerifyArgsT<SA> a;
    a.cand = v.cand; a.n_cand = v.n_cand; a.sa = SA(v.sa); a.base = v.base; a.sa_off = v.sa_off; a.lcp = v.lcp; a.d_doc_start = v.d_doc_start;
    a.n_docs = v.n_docs; a.num_distinct = v.num_distinct; a.max_doc_freq = v.max_doc_freq; a.merge = v.merge;
    a.thresh = v.thresh; a.rows = v.rows; a.d_row_count = v.d_row_count;
    // every candidate interval has <= cap entries; the single-wave bitmap path needs
    // MUM mode, <= 64 documents (then an accepted interval has <= 64 entries)
    bool fast = a.max_doc_freq == 1 && a.n_docs <= 64;
    uint64_t waves_needed = a.n_cand;
    static const bool packed_off = getenv("MMT_VERIFY_UNPACKED") != nullptr;      // tests: the wave-per-candidate kernel
    if (fast && a.n_docs <= 32 && !packed_off) {
        constexpr int W = 4;
        const int sub = a.n_docs <= 8 ? 8 : (a.n_docs <= 16 ? 16 : 32);
        waves_needed = (a.n_cand + (uint64_t)(64 / sub) - 1) / (uint64_t)(64 / sub);
        unsigned grid = (unsigned)std::min<uint64_t>((waves_needed + W - 1) / W, 256u * 16u);
        if (sub == 8) hipLaunchKernelGGL((k_verify_packed<W, 8, SA>), dim3(grid), dim3(W * 64), 0, s, a);
        else if (sub == 16) hipLaunchKernelGGL((k_verify_packed<W, 16, SA>), dim3(grid), dim3(W * 64), 0, s, a);
        else hipLaunchKernelGGL((k_verify_packed<W, 32, SA>), dim3(grid), dim3(W * 64), 0, s, a);
    } else if (fast) {
        // intervals longer than 64 cannot be all-distinct with <= 64 docs, but the fast path
        // reads only 64 entries; they are rejected by the popcount == cnt test because cnt > 64
        constexpr int W = 4;
        unsigned grid = (unsigned)std::min<uint64_t>((waves_needed + W - 1) / W, 256u * 16u);
        hipLaunchKernelGGL((k_verify<W, SA>), dim3(grid), dim3(W * 64), 0, s, a, 0);
    } else {
        size_t per_wave = (size_t)a.n_docs * 4;
        int W = per_wave * 4 <= 144 * 1024 ? 4 : (per_wave * 2 <= 144 * 1024 ? 2 : 1);
        unsigned grid = (unsigned)std::min<uint64_t>((waves_needed + W - 1) / W, 256u * 8u);
        size_t lds = per_wave * W;
        auto launch = [&](auto kernel, int threads) {
            if (lds > 48 * 1024)      // more dynamic LDS than the default launch limit: opt in (160 KiB per CU on gfx950)
                MMT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, s, a, 1);
        };
        if (W == 4) launch(k_verify<4, SA>, 256);
        else if (W == 2) launch(k_verify<2, SA>, 128);
        else launch(k_verify<1, SA>, 64);
    }
    MMT_HIP(hipGetLastError());
}
void verify_candidates(const VerifyArgs& a, hipStream_t s) {
    if (a.n_cand == 0) return;
    if (a.sa.wide()) verify_typed<Sa40>(a, s); else verify_typed<Sa32>(a, s);
}

// ---- occurrences of accepted rows, kept while the columns are produced window by window ------------------------------
// The stream is not stored (pfp_lcp_mum.hpp:197: one update() per suffix, nothing kept): what the writers need of the
// suffix array -- the text positions of every accepted interval -- is copied out of the window while it exists.
__global__ void k_row_counts(const Row* __restrict__ rows, uint32_t n_rows, uint64_t* __restrict__ cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) cnt[i] = rows[i].cnt;
}
// one wave per row: pool[pool_base + off[i] + k] = sa[row.start - win_base + k]; out[i] = the row with the pool offset as start
template <typename SA>
__global__ void k_capture_rows(const Row* __restrict__ rows, uint32_t n_rows, const uint64_t* __restrict__ off,
                               uint64_t pool_base, SA win, uint64_t win_base, SA pool, Row* __restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (i >= n_rows) return;
    const Row r = rows[i];
    const uint64_t dst = pool_base + off[i], src = r.start - win_base;
    for (uint32_t k = lane; k < r.cnt; k += 64) pool.set(dst + k, win.get(src + k));
    if (lane == 0) { Row o; o.start = dst; o.cnt = r.cnt; o.len = r.len; out[i] = o; }
}
void row_counts(const Row* rows, uint32_t n_rows, uint64_t* cnt, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_row_counts, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, rows, n_rows, cnt);
    MMT_HIP(hipGetLastError());
}
void capture_rows(const Row* rows, uint32_t n_rows, const uint64_t* off, uint64_t pool_base, SaCol win, uint64_t win_base,
                  SaCol pool, Row* out, hipStream_t s) {
    if (!n_rows) return;
    if (win.wide() != pool.wide()) throw HipError("capture_rows: window and pool differ in width");
    if (win.wide())
        hipLaunchKernelGGL(k_capture_rows<Sa40>, dim3(grid_for((uint64_t)n_rows * 64, 256)), dim3(256), 0, s, rows, n_rows, off,
                           pool_base, Sa40(win), win_base, Sa40(pool), out);
    else
        hipLaunchKernelGGL(k_capture_rows<Sa32>, dim3(grid_for((uint64_t)n_rows * 64, 256)), dim3(256), 0, s, rows, n_rows, off,
                           pool_base, Sa32(win), win_base, Sa32(pool), out);
    MMT_HIP(hipGetLastError());
}

// ============================================================================
// A9  anchor merge, one fold step -- merge_partitions
//     (src/merge_candidates.cpp:106-157), one thread per anchor position.
//     The reference walks i = 0..L_0 keeping "the last MUM that started at or
//     before i" per side; that is row (#starts in [0,i]) - 1, a prefix count.
// ============================================================================
__global__ void k_mark_starts(const uint64_t* __restrict__ starts, uint32_t n_rows, uint8_t* __restrict__ bv) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    bv[starts[r]] = 1;
}
void mark_starts(const uint64_t* starts, uint32_t n_rows, uint8_t* bv, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_mark_starts, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, starts, n_rows, bv);
    MMT_HIP(hipGetLastError());
}

// number of rows that start at or before anchor position i (starts ascending, one row per position): asked only at
// the few positions where a row starts, so a binary search replaces a prefix-count column over the whole anchor
__device__ __forceinline__ uint32_t rows_started(const uint64_t* __restrict__ starts, uint32_t n_rows, uint64_t i) {
    uint32_t lo = 0, hi = n_rows;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= i) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void k_fold_step(FoldArgs a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.len) return;
    const uint32_t na = a.nb_a[i], nb = a.nb_b[i];
    const bool both = na > 0 && nb > 0;
    const uint32_t nbo = both ? (na > nb ? na : nb) : 0u;                 // :121-123
    a.nb_out[i] = nbo;
    const bool sa_ = a.bv_a[i] != 0, sb_ = a.bv_b[i] != 0;
    if (!(sa_ || sb_) || !both) return;                                    // :132
    // number of starts in [0, i] per side
    const uint32_t ca = rows_started(a.start_a, a.n_a, i), cb = rows_started(a.start_b, a.n_b, i);
    if (ca == 0 || cb == 0) return;                                        // cur_mum1 && cur_mum2
    const uint32_t ra = ca - 1, rb = cb - 1;
    const uint64_t d1 = i - a.start_a[ra], d2 = i - a.start_b[rb];
    if (d1 > a.len_a[ra] || d2 > a.len_b[rb]) return;                      // :135
    const uint32_t s1 = (uint32_t)(a.len_a[ra] - d1), s2 = (uint32_t)(a.len_b[rb] - d2);
    const uint32_t nl = s1 < s2 ? s1 : s2;
    if (nl > nbo && nl >= a.min_len) {                                            // :139-141
        uint32_t slot = atomicAdd(a.d_count, 1u);
        if (slot < a.capacity) { a.out_pos[slot] = i; a.out_ra[slot] = ra; a.out_rb[slot] = rb; a.out_len[slot] = nl; }
    }
}
void fold_step(const FoldArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_step, dim3(grid_for(a.len, 256)), dim3(256), 0, s, a);
    MMT_HIP(hipGetLastError());
}

// threshold columns: 32 bits inside the engine, the fold and the exchange; 16 bits, saturated at 65535 like the
// reference's (mem_finder.hpp:299), where PREFIX.athresh / .thresh or a 16-bit table is written
__global__ void k_thresh_narrow(const uint32_t* __restrict__ src, uint64_t n, uint16_t* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = src[i];
        dst[i] = (uint16_t)(v > 65535u ? 65535u : v);
    }
}
__global__ void k_thresh_widen(const uint16_t* __restrict__ src, uint64_t n, uint32_t* __restrict__ dst) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
void thresh_narrow(const uint32_t* src, uint64_t n, uint16_t* dst, hipStream_t s) {
    if (!n) return;
    const uint64_t g = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_thresh_narrow, dim3((unsigned)(g < 65536 ? g : 65536)), dim3(256), 0, s, src, n, dst);
    MMT_HIP(hipGetLastError());
}
void thresh_widen(const uint16_t* src, uint64_t n, uint32_t* dst, hipStream_t s) {
    if (!n) return;
    const uint64_t g = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_thresh_widen, dim3((unsigned)(g < 65536 ? g : 65536)), dim3(256), 0, s, src, n, dst);
    MMT_HIP(hipGetLastError());
}

__global__ void k_gather_u32_idx32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t n,
                                   uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
void gather_u32_idx32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_gather_u32_idx32, dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, n, out);
    MMT_HIP(hipGetLastError());
}

}}  // namespace mmt::k
