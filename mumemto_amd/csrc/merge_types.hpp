// merge_types.hpp -- result of the anchor merge (shared by engine.hpp and merge.hpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mmt {

struct MergedRows {
    size_t n_docs = 0;
    std::vector<uint32_t> length;
    std::vector<int64_t> offsets;   // n_rows * n_docs, column 0 = anchor
    std::vector<uint8_t> strands;   // 1 = '+'
    std::vector<uint16_t> thresh;   // merged .athresh, L_0 + 1 entries
};

}  // namespace mmt
