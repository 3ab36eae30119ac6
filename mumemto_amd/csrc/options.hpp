// options.hpp -- command-line contract of mumemto_exec.
// Same option letters / long names and the same normalisation as the reference
// (src/pfp_mum.cpp:255-313 parse_build_options; include/pfp_mum.hpp:80-147
// BuildOptions::validate, :149-198 set_parameters).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace mmt {

struct CliError {
    std::string message;
    int code;
};

struct BuildOptions {
    std::string input_list;
    std::string output_prefix = "output";
    std::vector<std::string> files;
    bool use_rcomp = true;          // -r turns it OFF (pfp_mum.cpp:290)
    size_t pfp_w = 10;
    size_t hash_mod = 100;
    bool arrays_out = false;
    std::string arrays_in;
    bool arrays_in_flag = false;
    bool keep_temp = false;
    int num_distinct_docs = 0;
    bool overlap = true;
    std::string parse_prefix;
    bool from_parse_flag = false;
    size_t min_match_len = 20;
    int max_mem_freq = 0;
    int rare_freq = 1;
    bool binary = false;
    bool merge = false;
    bool anchor_merge = false;
    bool use_gsacak = false;
    bool only_parse = false;
    bool help = false;
    // --gpus N: one process per GPU of this node (cli_main.cpp).  --rank / --comm-file are what the launcher hands its ranks.
    int gpus = 1;
    int rank = -1;
    std::string comm_file;
    std::vector<std::string> notes;  // FORCE_LOG lines the reference would print

    // argv -> fields; throws CliError for unknown options.
    void parse(int argc, char** argv);
    // returns mum_mode (rare_freq == 1); throws CliError where the reference calls FATAL_ERROR.
    bool validate();
    // -k / -F normalisation against the number of documents.
    void set_parameters(size_t num_docs, bool mum_mode);
};

std::string usage_text();

// RefBuilder's file checks (src/ref_builder.cpp:52-138): existence, FASTA suffix,
// de-duplication (raw strings for a file-list, normalised absolute paths for
// positional arguments), at least two inputs.
std::vector<std::string> resolve_inputs(const BuildOptions& o);

}  // namespace mmt
