// merge_kernels.hpp -- device side of the anchor partition merge (A9).
//
// The fold of src/merge_candidates.cpp:97-157 with every table kept in HBM: the rows of a side
// are (anchor start, length) plus, per source partition folded so far, the source row and the
// accumulated shifts of its '+' and '-' columns (fix_neg_strand, :97-104).  Columns are
// materialised once for the rows that survive every fold step.
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

namespace mmt { namespace mk {

// rows of one partition as they sit in HBM (row-major tables)
struct PartTable {
    const uint32_t* length;
    const int64_t* offsets;     // n_rows * n_docs, column 0 = anchor
    const uint8_t* strands;     // 1 = '+'
    uint32_t n_rows, n_docs;
    uint32_t first_col;         // first merged column taken from this partition
    uint32_t skip;              // 1 = its own column 0 (the anchor) is not repeated
};

struct SideView {
    uint64_t* start; uint32_t* len;       // n rows, anchor order
    uint32_t* src; int64_t* plus; int64_t* minus;   // n * n_parts
    uint32_t n, n_parts;
};

// keys[r] = offsets[r * n_docs] (anchor offset), vals[r] = r; *bad |= 1 when an offset lies outside [0, L),
// *bad |= 2 when the rows are not in ascending anchor order
void leaf_keys(const int64_t* offsets, uint32_t n_rows, uint32_t n_docs, uint64_t L, uint64_t* keys, uint32_t* vals,
               uint32_t* bad, hipStream_t s);
// side rows of a leaf: start / len / src from (sorted keys, row order); shifts are zero
void leaf_side(const uint64_t* keys, const uint32_t* order, const uint32_t* length, uint32_t n_rows, SideView out,
               hipStream_t s);
// rows of a fold step in anchor order: order[q] indexes the (pos, ra, rb, len) tuples of k_fold_step
void fold_rows(const uint32_t* order, const uint64_t* pos, const uint32_t* ra, const uint32_t* rb,
               const uint32_t* nl, uint32_t found, SideView left, SideView right, SideView out, hipStream_t s);
// merged tables: row i of the output = side row perm[i] (perm may be null = identity)
void materialise(SideView side, const uint32_t* perm, const PartTable* parts /* device */, uint32_t n_docs_out,
                 const uint32_t* col_part /* device, n_docs_out */, uint32_t* out_len, int64_t* out_off,
                 uint8_t* out_st, hipStream_t s);
// out row i = in row perm[i]
void permute_rows(const uint32_t* perm, uint32_t n, uint32_t n_docs, const uint32_t* in_len, const int64_t* in_off,
                  const uint8_t* in_st, uint32_t* out_len, int64_t* out_off, uint8_t* out_st, hipStream_t s);
// coordinate-range fold: flags[r] = offsets[r * n_docs] + shift in [lo, hi); offsets[r * n_docs] += delta
void range_flags(const int64_t* off, uint32_t n, uint32_t n_docs, int64_t shift, int64_t lo, int64_t hi, uint8_t* flags,
                 hipStream_t s);
void shift_anchor(int64_t* off, uint32_t n, uint32_t n_docs, int64_t delta, hipStream_t s);
// keys[r] = isa[offsets[r * n_docs]] (suffix rank of the anchor occurrence), vals[r] = r
void rank_keys(const int64_t* off, uint32_t n, uint32_t n_docs, const uint32_t* isa, uint64_t anchor_len,
               uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t s);
void rank_keys64(const int64_t* off, uint32_t n, uint32_t n_docs, const uint64_t* isa, uint64_t anchor_len,
                 uint64_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t s);
// mumsio::serialize_mum (include/mumsio.hpp:311-320): LEN \t offsets \t strands \n; one wave per row
void table_measure(const uint32_t* len, const int64_t* off, uint32_t n, uint32_t n_docs, uint64_t* text_len,
                   hipStream_t s);
void table_write(const uint32_t* len, const int64_t* off, const uint8_t* st, uint32_t n, uint32_t n_docs,
                 const uint64_t* text_off, uint64_t text_base, char* text, hipStream_t s);   // row r -> text + text_off[r] - text_base

}}  // namespace mmt::mk
