// textref.hpp -- how kernels read the text T (and the string V = Dollar . T . Dollar^w the prefix-free parse reads).
//
// Two layouts behind one accessor:
//   * BYTES (every run so far): one byte per character in HBM, V[q] = v[q], q = text position + 1, a Dollar byte (0x02) in
//     front of T, 32 Dollar bytes and then zeros behind it (Engine::text_ptr).
//   * PACKED (MMT_PACKED_TEXT=1, or automatically when one byte per character would not fit next to the tables of the
//     parse: BASELINE configs[4] holds 573 G characters on every rank): two bits per character
//     (A C G T -> 0 1 2 3: code order = byte order), 32 characters per 64-bit word, text position p in bits
//     [2 (p & 31), + 2) of word p >> 5; everything else -- the '$' between strands and documents, N, IUPAC codes -- is an
//     EXCEPTION: its packed code is 0 and a sorted list of runs (start, length, byte) says what it really is; one bit per
//     block of 4096 positions says "this block holds an exception" (17 MB for 573 G characters: it stays in the caches),
//     so the common path costs the packed word(s) and one cached bit.  The reference never packs the text (it streams it
//     through the parser and keeps dictionary + parse, include/newscan.hpp): this layout is what lets a GPU hold the
//     whole collection for random access.
// Every reader gets ASCII bytes back (tx_byte, tx_load8, tx_load16), so the comparison and hashing logic above is the
// same for both layouts; in packed mode 8 characters cost a handful of integer operations to expand.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MMT_TX __host__ __device__ __forceinline__
#else
#define MMT_TX inline
#endif

namespace mmt {

struct ExcRun { uint64_t start; uint32_t len; uint32_t byte; };        // text positions [start, start + len) hold `byte`

struct TextRef {
    const uint8_t* v = nullptr;          // BYTES: V (v[q], q = text position + 1); nullptr in packed mode
    const uint64_t* packed = nullptr;    // PACKED: 2 bits per text position (two words of padding behind the text)
    const uint64_t* excw = nullptr;      // PACKED: bit b of word b >> 6 set <=> text positions [4096 b, 4096 b + 4096) hold an exception
    const ExcRun* runs = nullptr;        // PACKED: exception runs, ascending, disjoint
    uint32_t n_runs = 0;
    uint64_t n = 0;                      // text length
    MMT_TX bool is_packed() const { return v == nullptr; }
};

constexpr uint32_t TX_BLOCK_SHIFT = 12;  // exception flags per 4096 positions

// 16 bits = 8 characters (character k in bits 2k, 2k + 1) -> 8 ASCII bytes, character k in byte k
MMT_TX uint64_t tx_expand8(uint64_t bits) {
    uint64_t x = bits & 0xffffull;
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    const uint64_t hi = (x >> 1) & 0x0101010101010101ull;       // code >> 1
    const uint64_t both = hi & x;                                // 1 for T
    // A 0x41, C 0x43, G 0x47, T 0x54 = 0x41 + 2 c + 2 (c >> 1) + 11 (c == 3); no byte overflows
    return 0x4141414141414141ull + 2 * x + 2 * hi + 11 * both;
}
// packed code of a base, or 4 for an exception (bytes are upper case: Engine::build_text)
MMT_TX uint32_t tx_code_of(uint8_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }

#if defined(__HIPCC__)
__device__ __forceinline__ bool tx_block_flag(const TextRef& T, uint64_t p) {
    const uint64_t b = p >> TX_BLOCK_SHIFT;
    return (T.excw[b >> 6] >> (b & 63)) & 1ull;
}
// the byte at text position p of a packed text, exceptions looked up (p < n)
__device__ __forceinline__ uint8_t tx_exception_or(const TextRef& T, uint64_t p, uint8_t plain) {
    uint32_t lo = 0, hi = T.n_runs;                               // first run that ends behind p
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (T.runs[mid].start + T.runs[mid].len <= p) lo = mid + 1; else hi = mid; }
    if (lo < T.n_runs && T.runs[lo].start <= p) return (uint8_t)T.runs[lo].byte;
    return plain;
}
__device__ __forceinline__ uint8_t tx_packed_char(const TextRef& T, uint64_t p) {
    const uint32_t c = (uint32_t)(T.packed[p >> 5] >> (2 * (p & 31))) & 3u;
    const uint8_t plain = (uint8_t)(0x41u + 2u * c + 2u * (c >> 1) + 11u * (c & (c >> 1)));
    return tx_block_flag(T, p) ? tx_exception_or(T, p, plain) : plain;
}
// V[q]: Dollar in front of the text, 32 Dollars behind it, zeros beyond
__device__ __forceinline__ uint8_t tx_byte(const TextRef& T, uint64_t q) {
    if (T.v) return T.v[q];
    if (q == 0) return 2;
    const uint64_t p = q - 1;
    if (p >= T.n) return p < T.n + 32 ? (uint8_t)2 : (uint8_t)0;
    return tx_packed_char(T, p);
}
// V[q .. q + 8) as one little-endian word (what an unaligned 8-byte load of the byte layout gives).  Packed: the flag words
// and the two words of codes are asked for together (a reader that walks a text pays one round trip to memory per call, not
// one for the flags and another for the codes).
__device__ __forceinline__ uint64_t tx_load8(const TextRef& T, uint64_t q) {
    if (T.v) { uint64_t x; __builtin_memcpy(&x, T.v + q, 8); return x; }
    if (q >= 1 && q + 7 <= T.n) {                                 // text positions p .. p + 7 all inside the text
        const uint64_t p = q - 1;
        const uint64_t b0 = p >> TX_BLOCK_SHIFT, b1 = (p + 7) >> TX_BLOCK_SHIFT;
        const uint64_t f0 = T.excw[b0 >> 6], f1 = T.excw[b1 >> 6];
        const uint64_t w0 = T.packed[p >> 5], w1 = T.packed[(p >> 5) + 1];
        if (!(((f0 >> (b0 & 63)) | (f1 >> (b1 & 63))) & 1ull)) {
            const uint32_t sh = 2 * (uint32_t)(p & 31);
            uint64_t bits = w0 >> sh;
            if (sh > 48) bits |= w1 << (64 - sh);
            return tx_expand8(bits);
        }
    }
    uint64_t x = 0;
#pragma unroll 1
    for (int k = 0; k < 8; k++) x |= (uint64_t)tx_byte(T, q + k) << (8 * k);
    return x;
}
// PACKED only: the codes of V[q .. q + 64) as two words (character k of a word in bits 2k, 2k + 1); false when one of them is
// not a plain base inside the text (the caller then reads bytes).  Two texts are compared 64 characters a step by the XOR of
// such words: the first set bit pair is the first difference.  All loads are issued before the first is looked at.
__device__ __forceinline__ bool tx_codes64(const TextRef& T, uint64_t q, uint64_t& lo, uint64_t& hi) {
    if (q < 1 || q + 63 > T.n) return false;
    const uint64_t p = q - 1;
    const uint64_t b0 = p >> TX_BLOCK_SHIFT, b1 = (p + 63) >> TX_BLOCK_SHIFT;
    const uint64_t f0 = T.excw[b0 >> 6], f1 = T.excw[b1 >> 6];
    const uint64_t* w = T.packed + (p >> 5);                       // (two words of padding behind the text)
    const uint64_t w0 = w[0], w1 = w[1], w2 = w[2];
    if (((f0 >> (b0 & 63)) | (f1 >> (b1 & 63))) & 1ull) return false;
    const uint32_t sh = 2 * (uint32_t)(p & 31);
    lo = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    hi = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
    return true;
}
__device__ __forceinline__ uint8_t tx_code_char(uint32_t c) { return (uint8_t)(0x41u + 2u * c + 2u * (c >> 1) + 11u * (c & (c >> 1))); }
// text positions p0 .. p0 + 15 (p0 a multiple of 16; positions from n on as V has them: Dollars, then zeros) as 16 bytes
__device__ __forceinline__ uint4 tx_load16(const TextRef& T, uint64_t p0) {
    if (T.v) return *reinterpret_cast<const uint4*>(T.v + 1 + p0);
    union { uint4 v; uint64_t w[2]; } out;
    if (p0 + 16 <= T.n && !tx_block_flag(T, p0)) {
        const uint64_t bits = T.packed[p0 >> 5] >> (2 * (uint32_t)(p0 & 31));       // 32 bits = 16 characters, one word
        out.w[0] = tx_expand8(bits); out.w[1] = tx_expand8(bits >> 16);
        return out.v;
    }
    out.w[0] = tx_load8(T, p0 + 1); out.w[1] = tx_load8(T, p0 + 9);
    return out.v;
}
#endif

}  // namespace mmt
