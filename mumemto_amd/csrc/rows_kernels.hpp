// rows_kernels.hpp -- launch wrappers of rows_kernels.hip (A6 on the device).
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

#include "kernels.hpp"
#include "textref.hpp"

namespace mmt { namespace rk {

struct RowArgs {
    const k::Row* rows;         // accepted, left-maximal intervals (any order)
    const uint32_t* order;      // order[r] = index into rows of the r-th row in pop order
    uint32_t n_rows;
    SaCol sa;
    const uint64_t* doc_start;  // N + 1
    const uint64_t* doc_len;    // N (bases per document)
    uint32_t n_docs;
    int revcomp;
};

// Keys for the pop order of the reference's stack (closing position ascending, longer first).  Closing positions
// below 2^32: one 64-bit key (end << 32 | ~len).  Beyond: two stable sorts, first by len_keys (~len), then by
// end_keys (end) with the order of the first sort as values.
void row_keys(const k::Row* rows, uint32_t n_rows, uint64_t* keys, uint32_t* vals, hipStream_t s);
void row_len_keys(const k::Row* rows, uint32_t n_rows, uint32_t* keys, uint32_t* vals, hipStream_t s);
void row_end_keys(const k::Row* rows, const uint32_t* order, uint32_t n_rows, uint64_t* keys, hipStream_t s);
void mum_measure(const RowArgs& a, int64_t* slot_off, uint8_t* slot_st, uint32_t* keep, uint32_t* text_len,
                 hipStream_t s);
void mum_write(const RowArgs& a, const int64_t* slot_off, const uint8_t* slot_st, const uint32_t* keep,
               const uint32_t* row_idx, const uint64_t* text_off, uint32_t* out_len, int64_t* out_off, uint8_t* out_st,
               char* text, hipStream_t s);
void mem_measure(const RowArgs& a, uint32_t* occ_cnt, uint32_t* text_len, uint32_t* w_pos, uint32_t* w_doc,
                 hipStream_t s);
void mem_write(const RowArgs& a, const uint64_t* occ_off, const uint64_t* text_off, const uint32_t* w_pos,
               const uint32_t* w_doc, uint32_t* out_len, int64_t* out_off, uint64_t* out_doc, uint8_t* out_st,
               char* text, hipStream_t s);
void widen(const uint32_t* in, uint32_t n, uint64_t* out, hipStream_t s);
// The row tap (Engine::set_row_tap): rows (pool-indexed, fresh from a window) whose match begins with one of n_kmers k-mers
// (2 words each, as tx_load8 reads their characters) leave a copy -- length, offset and count of their suffix-array entries in
// t_sa -- at slots taken from used[0] (rows) / used[1] (entries); nothing is written beyond the capacities, the counters go on.
void tap_rows(const k::Row* rows, uint32_t n_rows, SaCol pool, const TextRef& T, const uint64_t* kmers, uint32_t n_kmers, uint32_t k,
              uint32_t* t_len, uint64_t* t_off, uint32_t* t_cnt, uint64_t* t_sa, uint64_t* used, uint64_t cap_rows, uint64_t cap_occ,
              hipStream_t s);
// every text position whose suffix begins with one of the k-mers (position, which), unordered; used[0] counts them
void kmer_positions(const TextRef& T, const uint64_t* kmers, uint32_t n_kmers, uint32_t k, uint64_t* out_pos, uint32_t* out_which,
                    uint64_t* used, uint64_t cap, hipStream_t s);

}}  // namespace mmt::rk
