// pfp_kernels.hpp -- launch wrappers of pfp_kernels.hip (rows A2-A4).
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace mmt { namespace pk {

void make_vtext(const uint8_t* text, uint32_t n, uint32_t w, uint8_t* v, uint32_t vlen_padded, hipStream_t s);
// trigger positions in two passes: 16-bit masks (one per 16 text positions) + triggers per workgroup; then, given the
// exclusive scan of those counts, the positions themselves (ascending)
uint32_t trigger_blocks(uint32_t n);
void trigger_masks(const uint8_t* text, uint32_t n, uint32_t w, uint32_t p, uint16_t* masks, uint32_t* block_count,
                   hipStream_t s);
void trigger_cuts(const uint16_t* masks, uint32_t n, const uint32_t* block_off, uint32_t* cuts, hipStream_t s);
void phrase_bounds(const uint32_t* cuts, uint32_t n_cuts, uint32_t n, uint32_t w, uint32_t* start, uint32_t* len,
                   hipStream_t s);
// h1: first fingerprint per phrase; pinfo: 16-byte record (second fingerprint, start, length) per phrase
void phrase_hash(const uint8_t* v, const uint32_t* start, const uint32_t* len, uint32_t m, uint64_t* h1, void* pinfo,
                 hipStream_t s);
void second_fingerprint(const void* pinfo, uint32_t m, uint64_t* h2, hipStream_t s);
void mark_distinct(const uint32_t* order, const uint64_t* h1_sorted, const void* pinfo, const uint8_t* v, uint32_t m,
                   uint32_t* flags, uint32_t* err, hipStream_t s);
void assign_distinct(const uint32_t* order, const uint32_t* scan, const uint32_t* flags, const uint32_t* len,
                     uint32_t m, uint32_t* pid, uint32_t* rep, uint32_t* dlen, hipStream_t s);
void copy_dict(const uint8_t* v, const uint32_t* start, const uint32_t* len, const uint32_t* which,
               const uint32_t* dstart, uint32_t n_phr, uint8_t* dict, uint64_t* dinfo, uint32_t dict_len,
               bool pack_prev, hipStream_t s);
void entry_info(const uint32_t* sa_d, const uint64_t* dinfo, const uint8_t* dict, uint32_t nd, bool pack_prev,
                uint32_t* esuf, uint32_t* ephr, uint8_t* ebw, hipStream_t s);
void group_flags(const uint32_t* esuf, const uint32_t* sa_d, const uint8_t* dict, uint32_t nd, uint32_t w,
                 uint32_t* gflag, uint32_t* pflag, uint32_t* vflag, hipStream_t s);
void phrase_ranks(const uint32_t* esuf, const uint32_t* ephr, const uint32_t* pscan, uint32_t nd, uint32_t* prank,
                  hipStream_t s);
// tab: n_distinct 16-byte records (phrase_table)
void phrase_table(const uint32_t* occ_start /* n_distinct + 1 */, const uint32_t* plen, const uint32_t* rep,
                  uint32_t n_distinct, void* tab, hipStream_t s);
// inverted lists from the parse suffix array (see pfp_kernels.hip): m + 1 (id, t) pairs to be stably sorted by id,
// then occ_start (n_distinct + 1 entries) and occ (m records of 8 bytes: t, text position of the occurrence)
void occ_sequence(const uint32_t* sa_p, const uint32_t* pid, uint32_t m, uint32_t D, uint32_t* keys, uint32_t* vals,
                  hipStream_t s);
void occ_finish(const uint32_t* ids, const uint32_t* ts, const uint32_t* sa_p, const uint32_t* pstart, uint32_t m,
                uint32_t* occ_start, void* occ, hipStream_t s);
void entry_compact(const uint32_t* esuf, const uint32_t* ephr, const uint8_t* ebw, const uint32_t* gflag,
                   const uint32_t* vflag, const uint32_t* vscan, const void* tab, uint32_t nd, uint32_t* ce_cnt,
                   uint32_t* ce_first, uint32_t* ce_offm1, uint8_t* ce_bwt, uint32_t* ce_gs, hipStream_t s);
void parse_ranks(const uint32_t* pid, const uint32_t* prank, uint32_t m, uint32_t* parse, hipStream_t s);
void invert_ranks(const uint32_t* prank, const uint32_t* rep, const uint32_t* dlen, uint32_t n_distinct,
                  uint32_t* which, uint32_t* slen, hipStream_t s);
void pack_keys_u32(const uint32_t* parse, uint32_t m, int bits, int chars, uint64_t* keys, uint32_t* vals,
                   hipStream_t s);
static const uint32_t EMIT_CAP = 1024;   // elements of one LDS tile of the emitter
static const uint32_t EMIT_TILE = 1024;  // output positions per workgroup of the emitter
struct EmitArgs {
    const uint32_t* segb;       // n_groups + 1 group begin offsets in the output (last = n + 1)
    const uint32_t* sege;       // n_groups + 1 compact entry index of every group's first entry
    uint32_t n_groups;
    const uint32_t* ce_eoff; const uint32_t* ce_cnt; const uint32_t* ce_first; const uint32_t* ce_offm1;
    const uint8_t* ce_bwt; const uint32_t* ce_gs;
    const uint2* occ;           // per phrase occurrence (t, text position), grouped by phrase
    uint32_t n;                 // text length; output stream has n + 1 entries, entry 0 = end sentinel
    uint32_t* sa; uint8_t* bwt;                   // n entries each (the sentinel entry is not stored)
    // oversized groups (more than EMIT_CAP suffixes): ids ascending, compact offsets (n_fb + 1 entries)
    const uint32_t* fb_group; const uint32_t* fb_off; uint32_t n_fb;
    uint32_t* fb_keys; uint32_t* fb_vals;     // compact fallback arrays, fb_off[n_fb] entries
    // fb_bits > 0: a fallback key is (t << fb_bits) | bwt_code[byte before the suffix] -- the sort order is that of t
    // (distinct within a group) and fallback_finish gets the BWT byte back without a random read of the text
    const uint8_t* bwt_code; uint32_t fb_bits;
    uint32_t* err;                            // consistency errors
};
struct BwtDecode { uint8_t byte[16]; };       // code -> byte
// tile_first_buf: scratch of n_out / EMIT_TILE + 2 entries
void emit(const EmitArgs& a, uint32_t n_out, uint32_t* tile_first_buf, hipStream_t s);
void oversize(const uint32_t* segb, uint32_t n_groups, uint32_t* osize, hipStream_t s);
void fallback_finish(const uint32_t* fb_group, const uint32_t* fb_off, uint32_t n_fb, const uint32_t* segb,
                     const uint32_t* sorted_keys, const uint32_t* sorted_vals, uint32_t fb_bits, const BwtDecode& decode,
                     const uint8_t* text, uint32_t n, uint32_t* sa, uint8_t* bwt, uint32_t* err, hipStream_t s);
void iota(uint32_t* out, uint32_t n, hipStream_t s);
void gather_u64(const uint64_t* src, const uint32_t* idx, uint32_t n, uint64_t* out, hipStream_t s);

}}  // namespace mmt::pk
