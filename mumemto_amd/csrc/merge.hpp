// merge.hpp -- anchor-based partition merge (src/merge_candidates.cpp), GPU fold.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "engine.hpp"
#include "merge_types.hpp"

namespace mmt {

// Left fold parts[0] (+) parts[1] (+) ... exactly like anchor_merge's main()
// (merge_candidates.cpp:208-219).  Partitions may hand their rows over in host memory or in HBM
// (mmt_partition.rows_on_device); everything after the upload runs on the device and the merged
// rows stay there.  min_len: minimum length of a merged MUM; the reference hard-codes 20.
MergedRows anchor_merge(Engine& e, const mmt_partition* parts, size_t k, uint32_t min_len = 20);
// The same fold by coordinate ranges: slice r of `world` near-equal slices [lo, hi) of the anchor is folded
// from the rows that start in [base, hi) and the thresholds [base, hi), base = lo - margin (merge.cpp).
uint64_t fold_margin(size_t k, uint32_t longest_row);
void fold_slice_bounds(uint64_t L, int world, int r, uint64_t margin, uint64_t* lo, uint64_t* hi, uint64_t* base);
uint32_t longest_row(Engine& e, const uint32_t* length, size_t n_rows, bool on_device);
// rows that start in [lo, hi) and thresholds [lo, hi) of anchor_merge(parts); thresh_is_slice: parts[g].thresh points at
// entry `base`, not at entry 0
MergedRows anchor_merge_slice(Engine& e, const mmt_partition* parts, size_t k, uint32_t min_len, uint64_t lo, uint64_t hi,
                              uint64_t base, bool thresh_is_slice);
// rows of a device table whose anchor offset + shift lies in [lo, hi), copied out with their anchor offsets moved by delta
void filter_rows(Engine& e, const uint32_t* len, const int64_t* off, const uint8_t* st, uint32_t n, uint32_t n_docs,
                 int64_t shift, int64_t lo, int64_t hi, int64_t delta, DevBuf<uint32_t>& o_len, DevBuf<int64_t>& o_off,
                 DevBuf<uint8_t>& o_st, uint32_t* kept);
MergedRows concat_pieces(Engine& e, std::vector<MergedRows>& pieces);
MergedRows anchor_merge_by_ranges(Engine& e, const mmt_partition* parts, size_t k, int slices, uint32_t min_len = 20);
// Direct-run order: sort by the suffix rank of the anchor occurrence.
void sort_like_direct(Engine& e, MergedRows& m);
// mumsio::write_mums / serialize_mum (include/mumsio.hpp:281-294, :311-320), formatted on the device
std::string format_merged(Engine& e, const MergedRows& m);
// the same bytes, formatted in HBM, staged in page-locked memory that stays with the engine, written to `path`
void write_merged_text(Engine& e, const MergedRows& m, const std::string& path);
// ... or left in that page-locked memory (valid until the next merged result is staged there); returns the bytes
const char* stage_merged_text(Engine& e, const MergedRows& m, size_t* bytes);
// host copies of the rows and thresholds (m.length / m.offsets / m.strands / m.thresh)
void download_merged(Engine& e, MergedRows& m);

// the fold's device scratch (kept between calls) back to the heap: mmt_pool_trim
void merge_release_scratch();

}  // namespace mmt
