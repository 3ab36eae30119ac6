// cli_main.cpp -- mumemto_exec: the reference CLI's contract for the hot path
// (src/pfp_mum.cpp:31-159 build_main): FASTA files in, PREFIX.mums | .mems |
// .bumbl + PREFIX.lengths (+ .athresh | .thresh/.thresh_rev, + .sa/.lcp/.bwt
// with -A) out.  All compute runs through libmumemto's GPU engine.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>

#include "engine.hpp"
#include "fasta.hpp"
#include "options.hpp"

namespace fs = std::filesystem;
using namespace mmt;

static void log_line(const char* tag, const std::string& msg) {
    std::fprintf(stderr, "\033[32m[%s] \033[m%s\n", tag, msg.c_str());
}
static double secs_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static void write_file(const std::string& path, const void* data, size_t n) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f.write(static_cast<const char*>(data), (std::streamsize)n);
}

// RefBuilder::write_lengths_file (src/ref_builder.cpp:193-209)
static void write_lengths(const std::string& prefix, const std::vector<FastaDoc>& docs) {
    std::ofstream out(prefix + ".lengths");
    for (const auto& d : docs) {
        const std::string canon = fs::canonical(d.path).string();
        out << canon << " * " << d.total << std::endl;
        for (size_t r = 0; r < d.names.size(); r++) out << canon << " " << d.names[r] << " " << d.lengths[r] << std::endl;
    }
}

static void put40(std::vector<uint8_t>& b, uint64_t v) {
    for (int k = 0; k < 5; k++) b.push_back((uint8_t)(v >> (8 * k)));
}

int main(int argc, char** argv) {
    std::fprintf(stderr, "\nmumemto_exec (MI355X / gfx950 build of the mumemto 1.4.0 hot path)\n");
    if (argc == 1) { std::fprintf(stderr, "Usage: mumemto_exec [options] [input_fasta [...]]\n\t-h, --help  prints detailed usage message\n"); return 0; }
    BuildOptions o;
    try {
        o.parse(argc, argv);
        if (o.help) { std::fputs(usage_text().c_str(), stderr); return 0; }
        const bool mum_mode = o.validate();
        if (o.from_parse_flag || o.arrays_in_flag)
            throw CliError{"-p/--from-parse and -a/--arrays-in are not available in this build", 1};
        const std::vector<std::string> inputs = resolve_inputs(o);
        o.set_parameters(inputs.size(), mum_mode);
        for (const auto& n : o.notes) log_line("build_main", n);

        auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> bases;
        std::vector<FastaDoc> docs;
        std::vector<uint64_t> doc_len;
        for (const auto& f : inputs) {
            docs.push_back(read_fasta(f, bases));
            if (docs.back().total == 0) {           // ref_builder.cpp:249-252 + pfp_mum.cpp:68-71
                std::cerr << std::endl << "Empty input file found: " << f << std::endl;
                throw CliError{"Please check the input files and ensure that it contains valid FASTA files. Cleaning up...", 1};
            }
            doc_len.push_back(docs.back().total);
        }
        write_lengths(o.output_prefix, docs);
        std::fprintf(stderr, "\033[32m[build_main] \033[0mread %zu files, %zu bases ... done.  (%.3f sec)\n", docs.size(),
                     bases.size(), secs_since(t0));

        if (std::getenv("MUMEMTO_DRY_RUN")) {     // host-side checks only (tests on machines without a GPU)
            uint64_t h = 1469598103934665603ull;
            for (uint8_t b : bases) { h ^= b; h *= 1099511628211ull; }
            std::printf("docs=%zu bases=%zu fnv1a=%016llx num_distinct=%d max_doc_freq=%d max_total_freq=%d revcomp=%d "
                        "merge=%d anchor=%d binary=%d min_len=%zu\n", docs.size(), bases.size(), (unsigned long long)h,
                        o.num_distinct_docs, o.rare_freq, o.max_mem_freq, (int)o.use_rcomp, (int)o.merge,
                        (int)o.anchor_merge, (int)o.binary, o.min_match_len);
            return 0;
        }
        t0 = std::chrono::steady_clock::now();
        Engine eng(std::getenv("MUMEMTO_DEVICE") ? std::atoi(std::getenv("MUMEMTO_DEVICE")) : 0, nullptr);
        uint64_t text_chars = 0;
        for (uint64_t l : doc_len) text_chars += (o.use_rcomp ? 2 : 1) * (l + 1);
        const uint64_t max_text = std::getenv("MUMEMTO_MAX_TEXT") ? std::strtoull(std::getenv("MUMEMTO_MAX_TEXT"), nullptr, 10)
                                                                  : 0xfffff000ull - 1;
        const bool partitioned = text_chars > max_text;
        if (!partitioned) eng.set_input_host(bases.data(), doc_len.data(), doc_len.size());
        auto write_pfp_files = [&]() {              // PREFIX.dict / PREFIX.parse as newscan.hpp:406-419 writes them
            eng.parse_only(o.use_rcomp, (uint32_t)o.pfp_w, (uint32_t)o.hash_mod);
            std::vector<uint8_t> dict; std::vector<uint32_t> parse;
            eng.pfp_copy_dict(dict); eng.pfp_copy_parse(parse);
            write_file(o.output_prefix + ".dict", dict.data(), dict.size());
            write_file(o.output_prefix + ".parse", parse.data(), parse.size() * 4);
        };
        if (o.only_parse && partitioned) throw CliError{"-P is not available for inputs larger than one suffix array", 1};
        if (o.only_parse) {                         // -P: pfp_mum.cpp:125-127
            write_pfp_files();
            log_line("build_main", "wrote the prefix-free parse (.dict, .parse)");
            return 0;
        }
        mmt_params p{};
        p.min_match_len = (uint32_t)o.min_match_len;
        p.num_distinct = (uint64_t)o.num_distinct_docs;
        p.max_doc_freq = o.rare_freq;
        p.max_total_freq = o.max_mem_freq;
        p.use_revcomp = o.use_rcomp ? 1 : 0;
        p.merge_metadata = o.merge ? 1 : 0;
        if (partitioned) {      // larger than one suffix array: anchor partitions + merge on this GPU
            if (o.keep_temp || o.arrays_out || (o.merge && !o.anchor_merge))
                throw CliError{"-K, -A and -M (without -n) are not available for inputs larger than one suffix array", 1};
            eng.run_partitioned_host(bases.data(), doc_len.data(), doc_len.size(), p, max_text);
            log_line("build_main", "text of " + std::to_string(text_chars) + " characters processed as " +
                                       std::to_string(eng.partitions_used()) + " anchor partitions");
        } else {
            eng.run(p);
        }
        const HostRows& R = eng.rows();
        std::fprintf(stderr, "\033[32m[build_main] \033[0mfinding multi-%ss on the GPU ... done.  (%.3f sec)\n",
                     mum_mode ? "MUM" : "MEM", secs_since(t0));

        if (!mum_mode) write_file(o.output_prefix + ".mems", R.text, R.text_len);
        else if (o.binary) { const std::string& b = eng.bumbl(); write_file(o.output_prefix + ".bumbl", b.data(), b.size()); }
        else write_file(o.output_prefix + ".mums", R.text, R.text_len);

        if (o.anchor_merge && partitioned) {
            write_file(o.output_prefix + ".athresh", eng.merged_thresh().data(), (doc_len[0] + 1) * sizeof(uint16_t));
        } else if (o.anchor_merge) {                // mem_finder.hpp:110-115
            std::vector<uint16_t> th(eng.thresh_len());
            eng.copy_thresh(th.data());
            write_file(o.output_prefix + ".athresh", th.data(), (doc_len[0] + 1) * sizeof(uint16_t));
        } else if (o.merge) {                       // mem_finder.hpp:116-157
            std::vector<uint16_t> fwd, rev;
            eng.thresh_files(fwd, rev);
            write_file(o.output_prefix + ".thresh", fwd.data(), fwd.size() * 2);
            write_file(o.output_prefix + ".thresh_rev", rev.data(), rev.size() * 2);
        }
        if (o.arrays_out) {                         // pfp_lcp_mum.hpp:323-369: 40-bit SA / LCP, 1-byte BWT, n+1 entries
            const uint64_t n = eng.text_length();
            std::vector<uint32_t> sa(n), lcp(n);
            std::vector<uint8_t> bwt(n), text(n);
            eng.copy_sa(sa.data()); eng.copy_lcp(lcp.data()); eng.copy_bwt(bwt.data()); eng.copy_text(text.data());
            std::vector<uint8_t> fsa, flcp, fbwt;
            put40(fsa, n); put40(flcp, 0); fbwt.push_back(n ? text[n - 1] : 0);
            for (uint64_t j = 0; j < n; j++) { put40(fsa, sa[j]); put40(flcp, lcp[j]); fbwt.push_back(bwt[j]); }
            write_file(o.output_prefix + ".sa", fsa.data(), fsa.size());
            write_file(o.output_prefix + ".lcp", flcp.data(), flcp.size());
            write_file(o.output_prefix + ".bwt", fbwt.data(), fbwt.size());
        }
        log_line("build_main", "Found " + std::to_string(R.n_rows) + " matches!");
        if (o.keep_temp) write_pfp_files();         // -K: keep PREFIX.dict / PREFIX.parse
        const float* ms = eng.stage_ms();
        std::fprintf(stderr, "GPU stages (ms): text %.2f | suffix sort %.2f | lcp+bwt %.2f | scan %.2f | verify %.2f | rows %.2f | format %.2f\n\n",
                     ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6]);
        return 0;
    } catch (const CliError& e) {
        std::fprintf(stderr, "\n\033[31mError: \033[m%s\n\n", e.message.c_str());
        return e.code;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "\n\033[31mError: \033[m%s\n\n", e.what());
        return 1;
    }
}
