// pfp_kernels.hpp -- launch wrappers of pfp_kernels.hip (rows A2-A4).
//
// Tables of text positions / stream offsets are passed as untyped pointers together with `wide`
// (false: uint32_t entries, true: uint64_t entries; wide.hpp).  Phrase, dictionary and parse indices are
// always 32 bits (at most 2^31 - 2 distinct phrases in the reference too: newscan.hpp:44).
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

#include "parse_lcp.hpp"
#include "textref.hpp"
#include "wide.hpp"

namespace mmt { namespace pk {

// trigger positions in two passes: 16-bit masks (one per 16 text positions) + triggers per workgroup; then, given the
// exclusive scan of those counts, the positions themselves (ascending)
uint32_t trigger_blocks(uint64_t n);
void trigger_masks(const TextRef& text, uint64_t n, uint32_t w, uint32_t p, uint16_t* masks, uint32_t* block_count,
                   hipStream_t s);
void trigger_cuts(const uint16_t* masks, uint64_t n, const uint32_t* block_off, void* cuts, bool wide, hipStream_t s);
void phrase_bounds(const void* cuts, uint32_t n_cuts, uint64_t n, uint32_t w, void* start, uint32_t* len, bool wide,
                   hipStream_t s);
// h1: first fingerprint per phrase; pinfo: 16-byte record per phrase (56 bits of the second fingerprint, start (40
// bits), length)
void phrase_hash(const TextRef& v, const void* start, const uint32_t* len, uint32_t m, uint64_t* h1, void* pinfo,
                 bool wide, hipStream_t s);
void second_fingerprint(const void* pinfo, uint32_t m, uint64_t* h2, hipStream_t s);
void mark_distinct(const uint32_t* order, const uint64_t* h1_sorted, const void* pinfo, const TextRef& v, uint32_t m,
                   uint32_t* flags, uint32_t* err, hipStream_t s);
void assign_distinct(const uint32_t* order, const uint32_t* scan, const uint32_t* flags, const uint32_t* len,
                     uint32_t m, uint32_t* pid, uint32_t* rep, uint32_t* dlen, hipStream_t s);
// *d_out = x[0] + ... + x[n - 1] in 64 bits
void sum_u32(const uint32_t* x, uint32_t n, uint64_t* d_out, hipStream_t s);
void copy_dict(const TextRef& v, const void* start, const uint32_t* len, const uint32_t* which,
               const uint32_t* dstart, uint32_t n_phr, uint8_t* dict, uint64_t* dinfo, uint32_t dict_len,
               bool pack_prev, bool wide, hipStream_t s);
void entry_info(const uint32_t* sa_d, const uint64_t* dinfo, const uint8_t* dict, uint32_t nd, bool pack_prev,
                uint32_t* esuf, uint32_t* ephr, uint8_t* ebw, hipStream_t s);
// LCP array of the dictionary in suffix-array order (every phrase terminator a symbol of its own): dict_irreducible
// leaves plcp[position] for the irreducible entries (longs: long_cap records of k::LongLcpLim for matches beyond 128
// characters -> k::long_lcp_lim), the caller turns plcp into PLCP (k::plcp_running_max), gathers it through sa_d
// (k::lcp_gather) and clamps it (dict_lcp_clamp).
void dict_irreducible(const uint8_t* dict, uint32_t nd, const uint32_t* sa_d, const uint32_t* esuf, const uint8_t* ebw,
                      uint32_t* plcp, void* longs, uint32_t* long_count, uint32_t long_cap, hipStream_t s);
void dict_lcp_clamp(uint32_t* lcp, const uint32_t* esuf, uint32_t nd, hipStream_t s);
// seg[r] = (1 << 32 if the entry before r is valid, or r = 0) | lcp_d[r], to be turned into its segmented minimum
void group_flags(const uint32_t* esuf, const uint32_t* lcp_d, uint32_t nd, uint32_t w, uint32_t* gflag, uint32_t* pflag,
                 uint32_t* vflag, uint64_t* seg, hipStream_t s);
void phrase_ranks(const uint32_t* esuf, const uint32_t* ephr, const uint32_t* pscan, uint32_t nd, uint32_t* prank,
                  hipStream_t s);
// tab: n_distinct 16-byte records (phrase_table)
void phrase_table(const uint32_t* occ_start /* n_distinct + 1 */, const uint32_t* plen, const uint32_t* rep,
                  uint32_t n_distinct, void* tab, hipStream_t s);
// inverted lists from the parse suffix array (see pfp_kernels.hip): m + 1 (id, t) pairs to be stably sorted by id,
// then occ_start (n_distinct + 1 entries) and occ (m records of 8 bytes: (t << pos_bits) | start of the occurrence
// in V; the caller guarantees that t and the position fit 64 bits together)
void occ_sequence(const uint32_t* sa_p, const uint32_t* pid, uint32_t m, uint32_t D, uint32_t* keys, uint32_t* vals,
                  hipStream_t s);
// occ_sl[k] = sl[t - 1] of the same occurrence (sl: parse_lcp.hpp), 0 for t = 0
void occ_finish(const uint32_t* ids, const uint32_t* ts, const uint32_t* sa_p, const void* pstart, uint32_t m,
                uint32_t* occ_start, uint64_t* occ, uint32_t pos_bits, const uint32_t* sl, uint32_t* occ_sl, bool wide,
                hipStream_t s);
// 12-byte records (t | position low | position high byte + 24 bits of sl[t - 1], saturated) for wide texts
void occ_finish12(const uint32_t* ids, const uint32_t* ts, const uint32_t* sa_p, const void* pstart, bool wide, uint32_t m,
                  uint32_t* occ_start, uint32_t* occ12, const uint32_t* sl, hipStream_t s);
// gscan: inclusive sum of gflag.  ce_gs[c] = g + 1 at the first entry of group g, 0 elsewhere; ce_hl / ce_slen: LCP with
// the valid entry before (segmin: the segmented minimum of group_flags' seg) and length of the entry's phrase suffix
// (scratch of group_heads)
void entry_compact(const uint32_t* esuf, const uint32_t* ephr, const uint8_t* ebw, const uint32_t* gflag,
                   const uint32_t* gscan, const uint32_t* vflag, const uint32_t* vscan, const uint64_t* segmin,
                   const void* tab, uint32_t nd, uint32_t* ce_cnt, uint32_t* ce_first, uint32_t* ce_offm1,
                   uint8_t* ce_bwt, uint32_t* ce_gs, uint32_t* ce_hl, uint32_t* ce_slen, hipStream_t s);
// ghead[g] = (length of the phrase suffix of group g, its LCP with the phrase suffix of group g - 1 -- 0 for g = 0), two
// uint32_t per group; sege[g] = first compact entry of group g
void group_heads(const uint32_t* sege, const uint32_t* ce_hl, const uint32_t* ce_slen, uint32_t n_groups, void* ghead,
                 hipStream_t s);
void parse_ranks(const uint32_t* pid, const uint32_t* prank, uint32_t m, uint32_t* parse, hipStream_t s);
void invert_ranks(const uint32_t* prank, const uint32_t* rep, const uint32_t* dlen, uint32_t n_distinct,
                  uint32_t* which, uint32_t* slen, hipStream_t s);
void pack_keys_u32(const uint32_t* parse, uint32_t m, int bits, int chars, uint64_t* keys, uint32_t* vals,
                   hipStream_t s);
static const uint32_t EMIT_CAP = 1024;   // elements of one LDS tile of the emitter
// output positions per workgroup of the emitter: below EMIT_CAP, so that the groups that begin in a tile -- its own
// positions plus what the last group hangs over -- usually fit ONE pass through LDS (with 1024 there almost always was a
// second pass for a last group of a few dozen elements); 1024 / 896 / 768 are compiled, MMT_EMIT_TILE chooses (tests)
uint32_t emit_tile();
struct EmitArgs {
    bool wide;                  // entry type of segb, ce_eoff, fb_off, fb_vals and of the suffix-array column
    const void* segb;           // n_groups + 1 group begin offsets in the output (last = n + 1)
    const uint32_t* sege;       // n_groups + 1 compact entry index of every group's first entry
    uint32_t n_groups;
    const void* ce_eoff; const uint32_t* ce_cnt; const uint32_t* ce_first; const uint32_t* ce_offm1;
    const uint8_t* ce_bwt; const uint32_t* ce_gs;
    const uint64_t* occ;        // per phrase occurrence (t << pos_bits) | V position, grouped by phrase
    const uint32_t* occ_sl = nullptr;   // and sl[t - 1] of the same occurrence
    const uint32_t* occ12 = nullptr;    // instead of the two: 12-byte records (occ_finish12) when t and the position exceed 64 bits
    uint32_t pos_bits;
    uint64_t n;                 // text length; output stream has n + 1 entries, entry 0 = end sentinel
    SaCol sa; uint8_t* bwt;                       // n entries each (the sentinel entry is not stored)
    // oversized groups (more than EMIT_CAP suffixes): ids ascending, compact offsets (n_fb + 1 entries).  The fallback
    // arrays of one launch hold the groups whose offsets start at fb_base (the emitter runs over ranges of output tiles
    // so that these arrays stay bounded: pfp.cpp).
    const uint32_t* fb_group; const void* fb_off; uint32_t n_fb;
    uint64_t fb_base;
    uint32_t* fb_keys; void* fb_vals;
    // fb_bits > 0: a fallback key is (t << fb_bits) | bwt_code[byte before the suffix] -- the sort order is that of t
    // (distinct within a group) and fallback_finish gets the BWT byte back without a random read of the text
    const uint8_t* bwt_code; uint32_t fb_bits;
    uint32_t* err;                            // consistency errors
    // The LCP column, and the window of the stream a launch writes: suffix-array entry j (stream entry j + 1) lands at
    // index j - out_base of sa / bwt / lcp when win_lo <= j < win_hi and nowhere otherwise (a window is produced from
    // the output tiles that cover it; groups that begin in those tiles may reach beyond it on either side).
    uint32_t* lcp = nullptr;
    const void* ghead = nullptr;              // per group: (|alpha|, LCP with the group before), group_heads
    RmqView rmq;                              // LCP of adjacent parse suffixes (parse_lcp.hpp): keys t1 < t2 -> min sl[t1 .. t2 - 1]
    uint32_t w = 0;
    uint64_t out_base = 0, win_lo = 0, win_hi = ~0ull;
};
struct BwtDecode { uint8_t byte[16]; };       // code -> byte
// tile_first[t] = first group whose begin offset is >= t * EMIT_TILE (tiles + 1 entries, tiles = ceil(n_out / TILE))
// (tile_base: the table begins at that tile -- entry t - tile_base --: the tables of one batch)
void tile_first(const void* segb, uint32_t n_groups, uint64_t tiles, uint32_t* out, bool wide, hipStream_t s, uint64_t tile_base = 0);
// output tiles [tile_lo, tile_hi); plan: emit_plan_bytes(tile_hi - tile_lo) bytes of device scratch (one record per tile,
// written by a pre-pass of the launch: what a workgroup needs to know about a tile before it can load anything of it)
size_t emit_plan_bytes(uint64_t tiles);
void emit(const EmitArgs& a, const uint32_t* tile_first_tab, uint64_t tile_base, void* plan, uint64_t tile_lo, uint64_t tile_hi, hipStream_t s);
// the oversized groups fb_group[f0 .. f0 + nf) of that launch, unsorted, into the fallback arrays: chunk0[f] = number of
// chunks of EMIT_BIG_CHUNK output positions in all oversized groups before f (n_fb + 1 entries, device)
constexpr uint32_t EMIT_BIG_CHUNK = 896;
void emit_big(const EmitArgs& a, const uint64_t* chunk0, uint32_t f0, uint32_t nf, uint64_t n_chunks, hipStream_t s);
// osize[g] = size of group g if it exceeds the emitter's LDS tile, else 0; err[2] counts groups of 2^32 suffixes or more
void oversize(const void* segb, uint32_t n_groups, uint32_t* osize, uint32_t* err, bool wide, hipStream_t s);
// out[i] = segb[fb_group[i]] (begin offset of every oversized group)
void gather_pos(const void* src, const uint32_t* idx, uint32_t n, void* out, bool wide, hipStream_t s);
// rel[i] = fb_off[f0 + i] - fb_off[f0], i = 0 .. count (32-bit offsets into the fallback arrays of one launch)
void relative_offsets(const void* fb_off, uint32_t f0, uint32_t count, uint32_t* rel, bool wide, hipStream_t s);
// oversized groups [f0, f1) after their segmented sort (total = their elements: the used part of the fallback arrays)
void fallback_finish(const uint32_t* fb_group, const void* fb_off, uint32_t f0, uint32_t f1, uint64_t fb_base,
                     const void* segb, const uint32_t* sorted_keys, const void* sorted_vals, uint32_t fb_bits,
                     const BwtDecode& decode, const TextRef& text, uint64_t n, const EmitArgs& ea, uint32_t total, bool wide,
                     hipStream_t s);
void iota(uint32_t* out, uint32_t n, hipStream_t s);
void gather_u64(const uint64_t* src, const uint32_t* idx, uint32_t n, uint64_t* out, hipStream_t s);

}}  // namespace mmt::pk
