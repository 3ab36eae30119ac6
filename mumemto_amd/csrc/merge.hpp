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

}  // namespace mmt
