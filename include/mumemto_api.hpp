// mumemto_api.hpp -- native C++ API of libmumemto (MI355X build).
// Same two functions, default arguments and result types as the reference's
// mumemto_library/mumemto_api.hpp:29-57, so the pybind11 module
// (python_bindings/src/mumemto_pybind.cpp:92-117) and C++ callers link unchanged.
// Errors are reported by exception (std::invalid_argument for f <= 1 in
// mumemto_mem, std::runtime_error for device failures); there is no CPU fallback.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "mumsio.hpp"

#if defined(__GNUC__) || defined(__clang__)
#define MUMEMTO_CPP_API __attribute__((visibility("default")))
#else
#define MUMEMTO_CPP_API
#endif

namespace mumemto {

struct MumResult {
    std::vector<mumsio::Mum> matches;
    std::vector<std::vector<size_t>> lengths;   // lengths[doc][record], raw record lengths
};

struct MemResult {
    std::vector<mumsio::Mem> matches;
    std::vector<std::vector<size_t>> lengths;
};

MUMEMTO_CPP_API MumResult mumemto_mum(const std::vector<std::vector<std::string>>& sequences,
                                      std::uint32_t min_match_len = 20, bool use_revcomp = true,
                                      size_t num_distinct = 0, bool use_gsacak = false);

MUMEMTO_CPP_API MemResult mumemto_mem(const std::vector<std::vector<std::string>>& sequences,
                                      std::uint32_t min_match_len = 20, bool use_revcomp = true,
                                      size_t num_distinct = 0, size_t max_total_freq = 0, size_t max_doc_freq = 2,
                                      bool use_gsacak = false);

}  // namespace mumemto

#include "mumemto.h"
