// merge.hpp -- anchor-based partition merge (src/merge_candidates.cpp), GPU fold.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "engine.hpp"

namespace mmt {

struct MergedRows {
    size_t n_docs = 0;
    std::vector<uint32_t> length;
    std::vector<int64_t> offsets;   // n_rows * n_docs, column 0 = anchor
    std::vector<uint8_t> strands;   // 1 = '+'
    std::vector<uint16_t> thresh;   // merged .athresh, L_0 + 1 entries
};

// Left fold parts[0] (+) parts[1] (+) ... exactly like anchor_merge's main()
// (merge_candidates.cpp:208-219); each step's O(L_0) walk is one kernel launch.
MergedRows anchor_merge(Engine& e, const mmt_partition* parts, size_t k);
// Direct-run order: sort by the suffix rank of the anchor occurrence.
void sort_like_direct(Engine& e, MergedRows& m);
// mumsio::write_mums / serialize_mum (include/mumsio.hpp:281-294, :311-320)
std::string format_merged(const MergedRows& m);

}  // namespace mmt
