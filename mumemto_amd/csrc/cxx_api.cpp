// cxx_api.cpp -- mumemto::mumemto_mum / mumemto::mumemto_mem on top of the C ABI
// (reference: mumemto_library/mumemto_api.cpp:332-411).
#include <stdexcept>

#include "../../include/mumemto_api.hpp"

namespace mumemto {
namespace {

struct Views {
    std::vector<std::vector<const char*>> ptrs;
    std::vector<mumemto_doc_view> docs;
    explicit Views(const std::vector<std::vector<std::string>>& seqs) {
        ptrs.resize(seqs.size());
        docs.resize(seqs.size());
        for (size_t d = 0; d < seqs.size(); d++) {
            for (const auto& r : seqs[d]) ptrs[d].push_back(r.c_str());
            docs[d].records = ptrs[d].empty() ? nullptr : ptrs[d].data();
            docs[d].num_records = ptrs[d].size();
        }
    }
};

std::vector<std::vector<size_t>> record_lengths_of(const std::vector<std::vector<std::string>>& seqs) {
    std::vector<std::vector<size_t>> out;
    for (const auto& d : seqs) { out.emplace_back(); for (const auto& r : d) out.back().push_back(r.size()); }
    return out;
}

}  // namespace

MumResult mumemto_mum(const std::vector<std::vector<std::string>>& sequences, std::uint32_t min_match_len,
                      bool use_revcomp, size_t num_distinct, bool use_gsacak) {
    MumResult res;
    if (sequences.empty()) return res;
    Views v(sequences);
    mumemto_mum_result* h = nullptr;
    int rc = ::mumemto_mum(v.docs.data(), sequences.size(), min_match_len, use_revcomp, num_distinct, use_gsacak, &h);
    if (rc) throw std::runtime_error(::mumemto_last_error());
    const size_t n = ::num_mums(h), N = ::num_docs(h);
    res.matches.resize(n);
    for (size_t i = 0; i < n; i++) {
        mumemto_mum_match_view mv = ::mum_at(h, i);
        res.matches[i].length = mv.length;
        res.matches[i].offsets.assign(mv.offsets, mv.offsets + N);
        res.matches[i].strands.assign(mv.strands, mv.strands + N);
    }
    ::mum_free(h);
    res.lengths = record_lengths_of(sequences);
    return res;
}

MemResult mumemto_mem(const std::vector<std::vector<std::string>>& sequences, std::uint32_t min_match_len,
                      bool use_revcomp, size_t num_distinct, size_t max_total_freq, size_t max_doc_freq,
                      bool use_gsacak) {
    MemResult res;
    if (sequences.empty()) return res;
    if (max_doc_freq <= 1)
        throw std::invalid_argument("per-sequence MEM frequency f must be > 1 (use mumemto_mum instead)");
    Views v(sequences);
    mumemto_mem_result* h = nullptr;
    int rc = ::mumemto_mem(v.docs.data(), sequences.size(), min_match_len, use_revcomp, num_distinct, max_total_freq,
                           max_doc_freq, use_gsacak, &h);
    if (rc) throw std::runtime_error(::mumemto_last_error());
    const size_t n = ::num_mems(h);
    res.matches.resize(n);
    for (size_t i = 0; i < n; i++) {
        mumemto_mem_match_view mv = ::mem_at(h, i);
        res.matches[i].length = mv.length;
        res.matches[i].offsets.assign(mv.offsets, mv.offsets + mv.occurrences);
        res.matches[i].seq_ids.assign(mv.seq_ids, mv.seq_ids + mv.occurrences);
        res.matches[i].strands.assign(mv.strands, mv.strands + mv.occurrences);
    }
    ::mem_free(h);
    res.lengths = record_lengths_of(sequences);
    return res;
}

}  // namespace mumemto
