// merge_types.hpp -- result of the anchor merge (shared by engine.hpp and merge.hpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_utils.hpp"

namespace mmt {

// The merged rows live in HBM; host copies are made on demand (download_merged in merge.hpp).
struct MergedRows {
    size_t n_docs = 0, n_rows = 0, thresh_len = 0;
    DevBuf<uint32_t> d_length;
    DevBuf<int64_t> d_offsets;      // n_rows * n_docs, column 0 = anchor
    DevBuf<uint8_t> d_strands;      // 1 = '+'
    DevBuf<uint32_t> d_thresh;      // merged thresholds, L_0 + 1 entries, 32 bits (the 16-bit .athresh form: `thresh`, saturated)
    bool on_host = false;
    std::vector<uint32_t> length;
    std::vector<int64_t> offsets;
    std::vector<uint8_t> strands;
    std::vector<uint16_t> thresh;

    MergedRows() = default;
    MergedRows(MergedRows&& o) noexcept { *this = std::move(o); }
    MergedRows& operator=(MergedRows&& o) noexcept {
        n_docs = o.n_docs; n_rows = o.n_rows; thresh_len = o.thresh_len; on_host = o.on_host;
        d_length.swap(o.d_length); d_offsets.swap(o.d_offsets); d_strands.swap(o.d_strands); d_thresh.swap(o.d_thresh);
        length = std::move(o.length); offsets = std::move(o.offsets); strands = std::move(o.strands);
        thresh = std::move(o.thresh);
        return *this;
    }
};

}  // namespace mmt
