// kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace mmt { namespace k {

// A candidate LCP interval [start, end] (real-suffix index space) of value len.
struct Cand {
    uint32_t start, end, len, flags;  // flags bit0: BWT characters not all equal (left-maximal)
};
static const uint32_t CAND_LEFT_MAXIMAL = 1u;

// ---- A1 text layout ---------------------------------------------------------
// text[p] for p in [0,n): UPPER(F_d) '$' [revcomp(UPPER(F_d)) '$'] per document;
// hist[256] += byte counts of the text.  d_doc_base / d_doc_start: N+1 entries.
void build_text(const uint8_t* raw, const uint64_t* d_doc_base, const uint64_t* d_doc_start, uint32_t n_docs,
                bool revcomp, uint8_t* text, uint64_t n, uint32_t* hist, hipStream_t s);

// ---- A8 direct suffix sort (prefix doubling) --------------------------------
// keys[i] = first `chars` symbols of suffix i, `bits` per symbol via code[256]
// (0 = past the end, smaller than every symbol); vals[i] = i.
// sep_code != 0: that symbol is a unique terminator (ordered by position); keys get a low "reached my terminator"
// bit and bits * chars + 1 <= 64 must hold (see kernels.hip).
void pack_keys(const uint8_t* text, uint32_t n, const uint8_t* d_code, int bits, int chars, uint32_t sep_code,
               uint64_t* keys, uint32_t* vals, hipStream_t s);
// headval[j] = j if keys[j] != keys[j-1] (or j == 0) else 0
// lsb_unique: a key with its low bit set is a bucket of its own
void mark_heads(const uint64_t* keys, uint32_t n, uint32_t* headval, bool lsb_unique, hipStream_t s);
// rank[sa[j]] = head[j]
void scatter_rank(const uint32_t* sa, const uint32_t* head, uint32_t n, uint32_t* rank, hipStream_t s);
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

// ---- LCP / BWT columns of the stream ----------------------------------------
// text must be readable (zero padded) up to n + 64.
// ISA-free LCP construction (see kernels.hip): K (n entries, cleared here) receives LCP + position at the
// irreducible suffixes; matches longer than 192 characters are queued (12-byte records, long_cap of them) for
// long_lcp; after an inclusive max-scan Ks of K, lcp_gather writes the column.  anchor_rank (optional) receives the
// suffix ranks of the text positions below anchor_len.
void irreducible_lcp(const uint8_t* text, uint32_t n, const uint32_t* sa, const uint8_t* bwt, uint32_t* K,
                     uint32_t* anchor_rank, uint32_t anchor_len, void* long_list, uint32_t* long_count,
                     uint32_t long_cap, hipStream_t s);
// huge_idx (count entries) / huge_count: scratch for the matches that outgrow one wave (see kernels.hip)
void long_lcp(const uint8_t* text, uint32_t n, void* long_list, uint32_t count, uint32_t* K, uint32_t* huge_idx,
              uint32_t* huge_count, hipStream_t s);
void lcp_gather(const uint32_t* Ks, const uint32_t* sa, uint32_t n, uint32_t* lcp, hipStream_t s);
void bwt_from_sa(const uint8_t* text, uint32_t n, const uint32_t* sa, uint8_t* bwt, hipStream_t s);

// ---- A5 match scan -----------------------------------------------------------
struct ScanArgs {
    const uint32_t* lcp;
    const uint8_t* bwt;
    uint32_t n;
    uint32_t min_len;
    uint32_t num_distinct;  // interval size lower bound
    uint32_t cap;           // interval size upper bound, 0 = none
    int emit_all;           // 1: emit structural candidates regardless of the BWT test (merge mode)
    Cand* out;
    uint32_t capacity;
    uint32_t* d_count;      // total candidates found (may exceed capacity)
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
    const Cand* cand;
    uint32_t n_cand;
    const uint32_t* sa;
    const uint32_t* lcp;
    const uint64_t* d_doc_start;  // N+1
    uint32_t n_docs;
    uint32_t num_distinct;
    uint32_t max_doc_freq;        // 0 = unlimited
    int merge;                    // record thresholds
    uint16_t* thresh;             // 2*(L_0+1) entries (merge only)
    Cand* rows;                   // accepted + left-maximal
    uint32_t* d_row_count;
};
void verify_candidates(const VerifyArgs& a, hipStream_t s);

// ---- A9 anchor merge, one fold step -------------------------------------------
// Per anchor position i (parallel): thresholds merged into nb_out; emits
// (i, row index in A, row index in B, new length) for every new MUM.
struct FoldArgs {
    uint64_t len;                       // L_0 + 1
    const uint16_t* nb_a; const uint16_t* nb_b; uint16_t* nb_out;
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

// out[i] = src[idx[i]]
void gather_u32_idx32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out, hipStream_t s);

}}  // namespace mmt::k
