// kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

#include "wide.hpp"
#include "textref.hpp"

namespace mmt { namespace k {

// A candidate LCP interval [start, end] (real-suffix index space) of value len.
struct Cand {
    uint32_t start, end, len, flags;  // flags bit0: BWT characters not all equal (left-maximal)
};
static const uint32_t CAND_LEFT_MAXIMAL = 1u;
// An accepted row: suffix-array interval [start, start + cnt) of value len (absolute positions, any text size).
struct Row {
    uint64_t start;
    uint32_t cnt, len;
};

// ---- A1 text layout ---------------------------------------------------------
// text[p] for p in [0,n): UPPER(F_d) '$' [revcomp(UPPER(F_d)) '$'] per document;
// hist[256] += byte counts of the text.  d_doc_base / d_doc_start: N+1 entries.
// Document d sits at raw[d_doc_base[d] .. + d_doc_len[d]) (any gaps between documents are ignored).
void build_text(const uint8_t* raw, const uint64_t* d_doc_base, const uint64_t* d_doc_len, const uint64_t* d_doc_start, uint32_t n_docs,
                bool revcomp, uint8_t* text, uint64_t n, uint64_t* hist, hipStream_t s);
// The same text packed to two bits per character (textref.hpp): packed[(n + 31) / 32 (+ padding)] words; maximal runs of one
// exception byte (anything but A C G T) leave as events -- ev_start[i] = (first position << 8) | byte, ev_end[i] = position
// behind the run, counts in ev_count[0 / 1], in no particular order: the host sorts and pairs them.
void pack_text(const uint8_t* raw, const uint64_t* d_doc_base, const uint64_t* d_doc_len, const uint64_t* d_doc_start, uint32_t n_docs,
               uint64_t* packed, uint64_t n, uint64_t* hist, uint64_t* ev_start, uint64_t* ev_end, uint32_t* ev_count, uint32_t ev_cap,
               uint64_t p_lo, uint64_t p_hi, hipStream_t s);      // text positions [p_lo, p_hi): all of it, or the span of whole documents
void pack_bytes(const uint8_t* text, uint64_t n, uint64_t* packed, uint64_t* ev_start, uint64_t* ev_end, uint32_t* ev_count,
                uint32_t ev_cap, hipStream_t s);
void unpack_text(const TextRef& T, uint64_t first, uint64_t count, uint8_t* out, hipStream_t s);

// ---- A8 direct suffix sort (prefix doubling) --------------------------------
// keys[i] = first `chars` symbols of suffix i, `bits` per symbol via code[256]
// (0 = past the end, smaller than every symbol); vals[i] = i.
// sep_code != PACK_NO_SEP: that symbol is a unique terminator (ordered by position); keys get a low "reached my terminator"
// bit and bits * chars + 1 <= 64 must hold (see kernels.hip).
static const uint32_t PACK_NO_SEP = 0xffffffffu;
void pack_keys(const uint8_t* text, uint32_t n, const uint8_t* d_code, int bits, int chars, uint32_t sep_code,
               uint64_t* keys, uint32_t* vals, hipStream_t s, uint32_t* run_ends = nullptr, uint32_t* run_count = nullptr,
               uint32_t run_cap = 0);
// run_ends / run_count (pack_keys): the last position of every run of `chars` or more equal symbols, in any order; *run_count
// counts them all (more than run_cap: the list is incomplete).  The helpers of DoublingSorter's RunRefine (sorter.hpp):
void equal_range_u64(const uint64_t* sorted, uint32_t n, const uint64_t* probe, uint32_t n_probes, uint32_t* lo_hi, hipStream_t s);
void run_keys(const uint32_t* sa, uint32_t cnt, const uint8_t* text, uint32_t n, const uint8_t* code, int bits, int chars,
              const uint32_t* ends, uint32_t n_ends, uint64_t* key2, hipStream_t s);
void force_heads(uint32_t* headval, const uint32_t* at, uint32_t cnt, uint32_t n, hipStream_t s);
// headval[j] = j if keys[j] != keys[j-1] (or j == 0) else 0
// lsb_unique: a key with its low bit set is a bucket of its own
void mark_heads(const uint64_t* keys, uint32_t n, uint32_t* headval, bool lsb_unique, hipStream_t s);
// rank[sa[j]] = head[j]
void scatter_rank(const uint32_t* sa, const uint32_t* head, uint32_t n, uint32_t* rank, hipStream_t s);
// rank[sa[c]] = head[c] where head[c] != old_head[c] (the sorted list of a doubling round against its heads before the round)
void scatter_rank_changed(const uint32_t* sa, const uint32_t* head, const uint32_t* old_head, uint32_t m, uint32_t* rank,
                          hipStream_t s);
// flags[j] = 1 unless bucket of j is a singleton
void flag_unsorted(const uint32_t* head, uint32_t n, uint8_t* flags, hipStream_t s);
void gather_active(const uint32_t* idx, uint32_t m, const uint32_t* sa, const uint32_t* head, uint32_t* out_pos,
                   uint32_t* out_sa, uint32_t* out_head, hipStream_t s);
// keys[c] = head[c] << shift | (rank[sa[c]+h] + 1, or 0 past the end)
void make_round_keys(const uint32_t* sa_c, const uint32_t* head_c, uint32_t m, const uint32_t* rank, uint32_t n,
                     uint32_t h, int shift, uint64_t* keys, hipStream_t s);
// Local sort of a doubling round (the active elements are grouped by bucket = the key bits from `shift` up):
// bound[t] (n_tiles + 1 entries) = first bucket start at or after t * target, ROUND_NO_BOUND when none within
// `limit` elements; round_local_sort sorts every range between consecutive bounds that fits ROUND_TILE_CAP
// elements in LDS and lists the longer ones (begin, end) for a segmented radix sort.
static const uint32_t ROUND_TILE_CAP = 2048;
void round_tile_bounds(const uint64_t* keys, uint32_t m, int shift, uint32_t target, uint32_t limit, uint32_t n_tiles,
                       uint32_t* bound, hipStream_t s);
void round_local_sort(const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, const uint32_t* bound,
                      uint32_t n_tiles, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count, uint32_t big_cap,
                      int shift, hipStream_t s);
// headval[c] = pos[c] if keys[c] != keys[c-1] (or c == 0) else 0
void mark_subheads(const uint64_t* keys, const uint32_t* pos, uint32_t m, uint32_t* headval, hipStream_t s);
// SA[pos[c]] = sa_sorted[c]; rank[sa_sorted[c]] = newhead[c]; flags[c] = still unsorted
void apply_round(const uint32_t* sa_sorted, const uint32_t* newhead, const uint32_t* pos, uint32_t m, uint32_t* sa,
                 uint32_t* rank, uint8_t* flags, hipStream_t s);
void compact_round(const uint32_t* idx, uint32_t m2, const uint32_t* pos, const uint32_t* sa_sorted,
                   const uint32_t* newhead, uint32_t* out_pos, uint32_t* out_sa, uint32_t* out_head, hipStream_t s);
// One doubling round in one pass over the active list (no key reaches HBM): round_head_bounds cuts the list at bucket
// boundaries (as round_tile_bounds, on the head column), round_fused gathers rank[suffix + h], sorts every tile in LDS and
// writes SA[pos], the sorted suffixes, the new heads and the still-tied flags; the ranks are scattered afterwards
// (scatter_rank over the sorted list: no tile may see half of a refined bucket).  Ranges beyond ROUND_TILE_CAP are
// listed (begin, end) and their tiles marked in tile_big (n_tiles bytes, cleared by the caller): round_big_keys writes
// their round keys, a segmented sort orders them, round_big_subheads marks their new heads (a running maximum over the
// whole head column follows) and round_big_apply writes their SA entries and flags.
uint32_t round_fused_cap();
void round_head_bounds(const uint32_t* headc, uint32_t m, uint32_t target, uint32_t limit, uint32_t n_tiles,
                       uint32_t* bound, hipStream_t s);
void round_fused(const uint32_t* sac, const uint32_t* headc, const uint32_t* pos, const uint32_t* bound, uint32_t n_tiles,
                 const uint32_t* rank, uint32_t n, uint32_t h, int shift, uint32_t* sa, uint32_t* sac_out,
                 uint32_t* head_out, uint8_t* flags, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count,
                 uint32_t big_cap, uint8_t* tile_big, hipStream_t s);
void round_big_keys(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles, const uint32_t* sac,
                    const uint32_t* headc, const uint32_t* rank, uint32_t n, uint32_t h, int shift, uint64_t* keys,
                    hipStream_t s);
void round_big_subheads(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles,
                        const uint64_t* keys, const uint32_t* pos, uint32_t* head, hipStream_t s);
void round_big_apply(const uint8_t* tile_big, const uint32_t* bound, uint32_t target, uint32_t n_tiles, uint32_t m,
                     const uint32_t* sa_sorted, const uint32_t* head, const uint32_t* pos, uint32_t* sa, uint8_t* flags,
                     hipStream_t s);

// ---- LCP / BWT columns of the stream ----------------------------------------
// text must be readable (zero padded) up to n + 64.
// ISA-free LCP construction (see kernels.hip): plcp (n entries, cleared here) receives the LCP at the text position of
// every irreducible suffix; matches longer than 192 characters are queued (long_lcp_record_bytes(wide) per record,
// long_cap of them) for long_lcp; plcp_running_max then turns the column into PLCP (scratch of
// plcp_running_max_scratch(n) bytes), and lcp_gather writes lcp[t] = PLCP[sa[j0 + t]] for any range of count
// suffix-array positions (j0 a multiple of 4).  anchor_rank (optional; uint32_t entries narrow, uint64_t wide)
// receives the suffix ranks of the text positions below anchor_len.
void irreducible_lcp(const uint8_t* text, uint64_t n, SaCol sa, const uint8_t* bwt, uint32_t* plcp, void* anchor_rank,
                     uint64_t anchor_len, void* long_list, uint32_t* long_count, uint32_t long_cap, hipStream_t s);
size_t long_lcp_record_bytes(bool wide);
// huge_idx (count entries) / huge_count: scratch for the matches that outgrow one wave (see kernels.hip)
void long_lcp(const uint8_t* text, uint64_t n, bool wide, void* long_list, uint32_t count, uint32_t* plcp,
              uint32_t* huge_idx, uint32_t* huge_count, hipStream_t s);
// The same comparison loops for matches whose value belongs somewhere else than at text position p (parse_lcp.hip: LCP
// of adjacent parse suffixes): records of (p, q, h, d) -- positions in the byte string `v` of nv bytes, h characters
// already known to match -- leave out[d] = LCP.  huge_idx: count entries of scratch.
struct LongLcpDst { uint64_t p, q; uint32_t h, d; };
// ... records of (p, q, h, lim): 32-bit positions, the match cannot exceed lim characters, out[p] = LCP (the dictionary of
// the parse: a match ends where the shorter phrase suffix ends)
struct LongLcpLim { uint32_t p, q, h, lim; };
void long_lcp_lim(const uint8_t* text, uint32_t n, void* long_list, uint32_t count, uint32_t* out, uint32_t* huge_idx,
                  uint32_t* huge_count, hipStream_t s);
void long_lcp_dst(const TextRef& v, uint64_t nv, void* long_list, uint32_t count, uint32_t* out, uint32_t* huge_idx,
                  uint32_t* huge_count, hipStream_t s);
size_t plcp_running_max_scratch(uint64_t n);
void plcp_running_max(uint32_t* plcp, uint64_t n, void* scratch, hipStream_t s);
void lcp_gather(const uint32_t* plcp, SaCol sa, uint64_t j0, uint64_t count, uint32_t* lcp, hipStream_t s);
void bwt_from_sa(const uint8_t* text, uint32_t n, const uint32_t* sa, uint8_t* bwt, hipStream_t s);
// rank[p] = j0 + t for the entries t < count of `piece` (a piece of the suffix array that starts at suffix-array index j0)
// whose text position p is below anchor_len; rank: uint32_t entries for a narrow piece, uint64_t for a wide one
void anchor_ranks(SaCol piece, uint64_t j0, uint64_t count, uint64_t anchor_len, void* rank, hipStream_t s);

// ---- A5 match scan -----------------------------------------------------------
struct ScanArgs {
    // The columns of one range of the suffix array: entry 0 may lie anywhere in the stream (a text beyond one LCP
    // column is scanned range by range); closing positions below `first` belong to the range before, the entries
    // before `first` only serve the walks to the left.  lcp and bwt + 0 must be 16-byte aligned.
    const uint32_t* lcp;
    const uint8_t* bwt;
    uint32_t n;
    uint32_t first = 0;
    int more_left = 0;      // entry 0 is not the start of the stream: a walk that reaches it is reported in d_count[4]
    uint32_t min_len;
    uint32_t num_distinct;  // interval size lower bound
    uint32_t cap;           // interval size upper bound, 0 = none
    int emit_all;           // 1: emit structural candidates regardless of the BWT test (merge mode)
    Cand* out;
    uint32_t capacity;
    uint32_t* d_count;      // [0] total candidates found (may exceed capacity); [4] walks that ran off the range
    // windows beyond one LDS tile (scan_needs_wide): block-wise prefix / suffix minima of the LCP column for the
    // window size num_distinct - 1 and the position of the last BWT change at or before every entry
    const uint32_t* wide_pre = nullptr;
    const uint32_t* wide_suf = nullptr;
    const uint32_t* wide_chg = nullptr;
};
void scan_intervals(const ScanArgs& a, hipStream_t s);
// More than ~1000 documents: the window of num_distinct - 1 entries does not fit k_scan's LDS tile.  The caller then
// fills pre / suf (n entries each) and chg (n entries, to be replaced by its inclusive running maximum) with
// scan_wide_prepare and hands them over in ScanArgs.
bool scan_needs_wide(const ScanArgs& a);
bool scan_needs_wide_docs(size_t n_docs);      // could any scan of this collection need them (num_distinct <= documents)
void scan_wide_prepare(const uint32_t* lcp, const uint8_t* bwt, uint32_t n, uint32_t num_distinct, uint32_t* pre,
                       uint32_t* suf, uint32_t* chg, hipStream_t s);

struct VerifyArgs {
    const Cand* cand;             // positions relative to the scanned range
    uint32_t n_cand;
    SaCol sa;                     // the suffix array, or the window of it that holds the scanned range
    uint64_t base;                // suffix-array index of entry 0 of the scanned range (rows leave with absolute positions)
    uint64_t sa_off = 0;          // index of that entry in `sa` (= base for the whole column, 0 for a window)
    const uint32_t* lcp;          // LCP column of the scanned range
    const uint64_t* d_doc_start;  // N+1
    uint32_t n_docs;
    uint32_t num_distinct;
    uint32_t max_doc_freq;        // 0 = unlimited
    int merge;                    // record thresholds
    uint32_t* thresh;             // 2*(L_0+1) entries (merge only); 32 bits: never saturated inside the engine
    Row* rows;                    // accepted + left-maximal, appended at *d_row_count
    uint32_t* d_row_count;
};
void verify_candidates(const VerifyArgs& a, hipStream_t s);
// The suffix-array entries of accepted rows, copied out of a window of the column: cnt[i] = rows[i].cnt (to be turned into
// exclusive offsets by the caller), then pool[pool_base + off[i] + k] = win[rows[i].start - win_base + k] and out[i] = the
// row with that pool offset as its start (what the writers index instead of the column).
void row_counts(const Row* rows, uint32_t n_rows, uint64_t* cnt, hipStream_t s);
void capture_rows(const Row* rows, uint32_t n_rows, const uint64_t* off, uint64_t pool_base, SaCol win, uint64_t win_base,
                  SaCol pool, Row* out, hipStream_t s);

// ---- A9 anchor merge, one fold step -------------------------------------------
// Per anchor position i (parallel): thresholds merged into nb_out; emits
// (i, row index in A, row index in B, new length) for every new MUM.
struct FoldArgs {
    uint64_t len;                       // L_0 + 1
    const uint32_t* nb_a; const uint32_t* nb_b; uint32_t* nb_out;
    uint32_t n_a, n_b;                                // rows per side (start_* ascending)
    const uint64_t* start_a; const uint64_t* start_b; // offsets[0] of row r (sorted)
    const uint32_t* len_a; const uint32_t* len_b;     // length of row r
    const uint8_t* bv_a; const uint8_t* bv_b;         // MUM start flags per position
    uint64_t* out_pos; uint32_t* out_ra; uint32_t* out_rb; uint32_t* out_len;
    uint32_t capacity; uint32_t* d_count;
    uint32_t min_len;                   // 20 in the reference (merge_candidates.cpp:141)
};
void fold_step(const FoldArgs& a, hipStream_t s);
void mark_starts(const uint64_t* starts, uint32_t n_rows, uint8_t* bv, hipStream_t s);
// threshold columns between their 32-bit form (engine, fold, exchange) and the reference's saturating 16-bit form (files)
void thresh_narrow(const uint32_t* src, uint64_t n, uint16_t* dst, hipStream_t s);
void thresh_widen(const uint16_t* src, uint64_t n, uint32_t* dst, hipStream_t s);

// out[i] = src[idx[i]]
void gather_u32_idx32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out, hipStream_t s);

}}  // namespace mmt::k
