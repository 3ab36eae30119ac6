#include "options.hpp"

#include <getopt.h>

#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <unordered_set>

namespace fs = std::filesystem;

namespace mmt {

static bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
static bool is_fasta_name(const std::string& f) {
    for (const char* e : {".fa", ".fasta", ".fna", ".fa.gz", ".fasta.gz", ".fna.gz"})
        if (ends_with(f, e)) return true;
    return false;
}

void BuildOptions::parse(int argc, char** argv) {
    static const struct option longopts[] = {
        {"help", no_argument, nullptr, 'h'},          {"input", required_argument, nullptr, 'i'},
        {"output", required_argument, nullptr, 'o'},  {"revcomp", no_argument, nullptr, 'r'},
        {"minimum-genomes", required_argument, nullptr, 'k'}, {"no-overlap", no_argument, nullptr, 's'},
        {"modulus", required_argument, nullptr, 'm'}, {"from-parse", no_argument, nullptr, 'p'},
        {"only-parse", no_argument, nullptr, 'P'},    {"min-match-len", required_argument, nullptr, 'l'},
        {"max-freq", required_argument, nullptr, 'F'}, {"arrays-out", no_argument, nullptr, 'A'},
        {"arrays-in", required_argument, nullptr, 'a'}, {"keep-temp-files", no_argument, nullptr, 'K'},
        {"window", required_argument, nullptr, 'w'},  {"rare", required_argument, nullptr, 'f'},
        {"binary", no_argument, nullptr, 'b'},        {"merge", no_argument, nullptr, 'M'},
        {"anchor", no_argument, nullptr, 'n'},        {"use-gsacak", no_argument, nullptr, 'g'},
        {"gpus", required_argument, nullptr, 1000},   {"rank", required_argument, nullptr, 1001},
        {"comm-file", required_argument, nullptr, 1002},
        {nullptr, 0, nullptr, 0}};
    optind = 1;
    int c;
    while ((c = getopt_long(argc, argv, "hi:F:o:w:sl:ra:AKk:p:m:f:bgMnP", longopts, nullptr)) >= 0) {
        switch (c) {
            case 'h': help = true; break;
            case 'i': input_list = optarg; break;
            case 'o': output_prefix = optarg; break;
            case 'w': pfp_w = (size_t)std::atoi(optarg); break;
            case 'r': use_rcomp = false; break;
            case 's': overlap = false; break;
            case 'k': num_distinct_docs = std::atoi(optarg); break;
            case 'm': hash_mod = (size_t)std::atoi(optarg); break;
            case 'p': parse_prefix = optarg; from_parse_flag = true; break;
            case 'l': min_match_len = (size_t)std::atoi(optarg); break;
            case 'F': max_mem_freq = std::atoi(optarg); break;
            case 'A': arrays_out = true; break;
            case 'a': arrays_in = optarg; arrays_in_flag = true; break;
            case 'K': keep_temp = true; break;
            case 'f': rare_freq = std::atoi(optarg); break;
            case 'b': binary = true; break;
            case 'M': merge = true; break;
            case 'n': anchor_merge = true; break;
            case 'g': use_gsacak = true; break;
            case 'P': only_parse = true; break;
            case 1000: gpus = std::atoi(optarg); break;
            case 1001: rank = std::atoi(optarg); break;
            case 1002: comm_file = optarg; break;
            default: throw CliError{usage_text(), 1};
        }
    }
    for (int i = optind; i < argc; i++) files.push_back(argv[i]);
}

bool BuildOptions::validate() {
    if (!input_list.empty() && !fs::is_regular_file(input_list))
        throw CliError{"The provided file-list is not valid.", 1};
    if (!input_list.empty() && !files.empty()) {
        notes.push_back("Using filelist, ignoring positional args");
        files.clear();
    } else if (input_list.empty() && files.empty() && !from_parse_flag && !arrays_in_flag) {
        throw CliError{"Need to provide a file-list or files as positional args for processing.", 1};
    }
    for (const auto& f : files)
        if (!fs::is_regular_file(f)) throw CliError{"The following file path is not valid: " + f, 1};
    fs::path p(output_prefix);
    if (p.parent_path().string().empty() && !p.string().empty()) output_prefix = "./" + output_prefix;
    else if (!fs::exists(p.parent_path())) fs::create_directories(p.parent_path());
    if (only_parse && (use_gsacak || arrays_in_flag || from_parse_flag)) {
        only_parse = false;
        notes.push_back("only-parse flag is not supported with use-gsacak, arrays-in, or from-parse, ignoring flag");
    }
    if (use_gsacak && from_parse_flag) throw CliError{"--use-gsacak flag is incompatible with --from-parse flag", 1};
    if (use_gsacak && arrays_in_flag) throw CliError{"--use-gsacak flag is incompatible with --arrays-in flag", 1};
    if (from_parse_flag && arrays_in_flag) throw CliError{"--from-parse flag is incompatible with --arrays-in flag", 1};
    if (anchor_merge && !merge) merge = true;
    if (rare_freq < 0) throw CliError{"Per-sequence MEM frequency must be > 0 (or 0 for no limit).", 1};
    if (binary && rare_freq != 1) {
        notes.push_back("binary output is not supported for multi-MEMs, ignoring flag");
        binary = false;
    }
    return rare_freq == 1;
}

void BuildOptions::set_parameters(size_t num_docs, bool mum_mode) {
    const int N = (int)num_docs;
    const std::string kind = mum_mode ? "MUMs" : "MEMs";
    if (num_distinct_docs < -N) {
        notes.push_back("Too few number of sequences, defaulting to multi-" + kind + " in 2 or more sequences");
        num_distinct_docs = 2;
    } else if (num_distinct_docs <= 0) {
        num_distinct_docs = N + num_distinct_docs;
    } else if (num_distinct_docs == 1) {
        notes.push_back("Too few number of sequences, defaulting to multi-" + kind + " in 2 or more sequences");
        num_distinct_docs = 2;
    } else if (num_distinct_docs >= N) {
        notes.push_back("Too large number of sequences, defaulting to multi-" + kind + " in all sequences");
        num_distinct_docs = N;
    }
    if (merge && num_distinct_docs != N) throw CliError{"Merging not available for partial multi-MUM/MEMs", 1};
    if (merge && rare_freq != 1) throw CliError{"Merging not available for multi-MEMs", 1};
    if (max_mem_freq < -N || max_mem_freq == 1) {
        notes.push_back("Invalid maximum total MEM frequency, defaulting to no upper limit");
        max_mem_freq = 0;
    } else if (max_mem_freq < 0) {
        max_mem_freq = N + max_mem_freq;
    }
    // the per-document cap overrides the total cap
    if (rare_freq > 0 && (max_mem_freq == 0 || max_mem_freq > rare_freq * N)) max_mem_freq = rare_freq * N;
}

std::vector<std::string> resolve_inputs(const BuildOptions& o) {
    std::vector<std::string> out;
    std::unordered_set<std::string> seen;
    if (!o.input_list.empty()) {
        std::ifstream in(o.input_list);
        std::string line;
        while (std::getline(in, line)) {
            // first space-separated word of the line
            size_t a = line.find_first_not_of(' ');
            if (a == std::string::npos) continue;
            size_t b = line.find(' ', a);
            std::string f = line.substr(a, b == std::string::npos ? std::string::npos : b - a);
            if (!fs::is_regular_file(f)) throw CliError{"The following path in the input list is not valid: " + f, 1};
            if (!is_fasta_name(f)) throw CliError{"The following input-file is not a FASTA file: " + f, 1};
            if (seen.insert(f).second) out.push_back(f);
        }
    } else {
        for (const auto& f : o.files) {
            if (!fs::is_regular_file(f)) throw CliError{"The following file path is not valid: " + f, 1};
            if (!is_fasta_name(f)) throw CliError{"The following input-file is not a FASTA file: " + f, 1};
            std::string norm = fs::absolute(f).lexically_normal().string();
            if (seen.insert(norm).second) out.push_back(norm);
        }
    }
    if (out.size() <= 1)
        throw CliError{"Multiple FASTA inputs required. Perhaps split a multi-FASTA into multiple files?", 1};
    return out;
}

std::string usage_text() {
    return "\nmumemto_exec (MI355X build) - find maximal [unique | exact] matches across a collection.\n"
           "Usage: mumemto_exec [options] [input_fasta [...]]\n\n"
           "*** for all options, N = # of sequences ***\n"
           "I/O options:\n"
           "\t-h, --help                      prints this usage message\n"
           "\t-i, --input           [FILE]    path to a file-list of genomes to use (overrides positional args)\n"
           "\t-o, --output          [PREFIX]  output prefix path\n"
           "\t-r, --revcomp                   do NOT include the reverse complement of the sequences\n"
           "\t-b, --binary                    output binary .bumbl format (multi-MUMs only)\n"
           "\t-A, --arrays-out                write LCP, BWT, and SA to file (5-byte / 1-byte entries)\n"
           "\t-M, --merge                     output extra metadata to enable merging multi-MUMs\n"
           "\t-n, --anchor                    anchor-based merging metadata (PREFIX.athresh; implies -M)\n"
           "Exact match parameters:\n"
           "\t-l, --min-match-len   [INT]     minimum MUM or MEM length (default: 20)\n"
           "\t-k, --minimum-genomes [INT]     find matches in at least k sequences (k < 0: N - |k|; default: all)\n"
           "\t-f, --rare            [INT]     maximum number of occurences per sequence (0 = no limit; default 1)\n"
           "\t-F, --max-freq        [INT]     maximum number of total occurences (negative: relative to N)\n"
           "Several GPUs of this node (one process per GPU, exchange over RCCL):\n"
           "\t    --gpus            [INT]     strict multi-MUMs: anchor partitions, one per GPU, merged like `mumemto merge`;\n"
           "\t                                other modes: every GPU builds the stream and scans its share of it\n"
           "Accepted for compatibility (the GPU pipeline has no use for them):\n"
           "\t-g, --use-gsacak  -s, --no-overlap\n"
           "PFP options:\n"
           "\t-w, --window          [INT]     window size of the PFP files written by -P / -K (default: 10)\n"
           "\t-m, --modulus         [INT]     hash modulus of the PFP files written by -P / -K (default: 100)\n"
           "\t-P, --only-parse                only compute the prefix-free parse (PREFIX.dict, PREFIX.parse)\n"
           "\t-K, --keep-temp-files           also write PREFIX.dict and PREFIX.parse\n"
           "Stage checkpoints (need PREFIX.lengths next to the files):\n"
           "\t-p, --from-parse     [PREFIX]  start from PREFIX.parse and PREFIX.dict (made with the same -w and -r)\n"
           "\t-a, --arrays-in      [PREFIX]  start from PREFIX.sa, PREFIX.lcp and PREFIX.bwt as -A writes them\n";
}

}  // namespace mmt
