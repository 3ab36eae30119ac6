// extract_mums_main.cpp -- extract_mums: the sequences of the multi-MUMs of a run as a multi-FASTA
// (reference CLI contract: src/extract_mums.cpp:24-33,92-185).  One record per row of PREFIX.mums / PREFIX.bumbl in
// file order, `>mum_<i>`, the bases cut out of the first document named by PREFIX.lengths, followed by `#` unless -t.
// The string-based merge feeds these files back into mumemto_exec (mumemto_amd/merge_mums.py).  Host-only tool.
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <getopt.h>

#include "../../include/mumsio.hpp"
#include "fasta.hpp"

static bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static void usage() {
    std::fprintf(stderr, "\nextract_mums - extract MUMs from a MUM and length file\n");
    std::fprintf(stderr, "Usage: extract_mums [options] -m mum_file -o output_file\n\n");
    std::fprintf(stderr, "Options:\n");
    std::fprintf(stderr, "\t%-32sprints this usage message\n", "-h, --help");
    std::fprintf(stderr, "\t%-22s%-10spath to a mum file\n", "-m, --mums", "[FILE]");
    std::fprintf(stderr, "\t%-22s%-10spath to a length file (optional, uses associated length file if not provided)\n",
                 "-l, --lengths", "[FILE]");
    std::fprintf(stderr, "\t%-22s%-10soutput path\n", "-o, --output", "[FILE]");
    std::fprintf(stderr, "\t%-32sdo not add terminator (#) to end of each MUM sequence\n", "-t, --no-terminator");
}

int main(int argc, char** argv) {
    if (argc == 1) { usage(); return 1; }
    std::string mums, lengths, output;
    bool terminator = true;
    static struct option long_options[] = {{"help", no_argument, nullptr, 'h'},
                                           {"mums", required_argument, nullptr, 'm'},
                                           {"lengths", required_argument, nullptr, 'l'},
                                           {"output", required_argument, nullptr, 'o'},
                                           {"no-terminator", no_argument, nullptr, 't'},
                                           {nullptr, 0, nullptr, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "hm:l:o:t", long_options, nullptr)) >= 0) {
        switch (c) {
            case 'h': usage(); return 0;
            case 'm': mums = optarg; break;
            case 'l': lengths = optarg; break;
            case 'o': output = optarg; break;
            case 't': terminator = false; break;
            default: usage(); return 1;
        }
    }
    if (optind < argc && mums.empty()) mums = argv[optind];
    if (mums.empty()) { std::cerr << "Error: No mum file provided\n"; return 1; }
    if (!ends_with(mums, ".mums") && !ends_with(mums, ".bumbl")) mums += ".mums";
    if (!std::filesystem::is_regular_file(mums)) { std::cerr << "Error: Invalid mum file provided: " << mums << std::endl; return 1; }
    const std::string stem = mums.substr(0, mums.find_last_of('.'));
    if (lengths.empty()) lengths = stem + ".lengths";
    if (!std::filesystem::is_regular_file(lengths)) {
        std::cerr << "Error: Invalid lengths file: " << lengths << std::endl;
        std::cerr << "Provide a lengths file with the -l option or ensure the mum file has an associated lengths file" << std::endl;
        return 1;
    }
    if (output.empty()) output = stem + "_mums.fa";
    else if (!ends_with(output, ".fa")) output += ".fa";

    try {
        // the first document: the path is the first token of the first line of the lengths file
        std::ifstream lf(lengths);
        std::string line;
        if (!std::getline(lf, line)) { std::cerr << "Error: Length file is empty" << std::endl; return 1; }
        const std::string fasta = line.substr(0, line.find_first_of(" \t\n\r"));
        if (!std::filesystem::is_regular_file(fasta)) { std::cerr << "Error: Invalid FASTA path in length file" << std::endl; return 1; }
        std::vector<uint8_t> bases;
        mmt::read_fasta(fasta, bases);
        if (bases.empty()) { std::cerr << "Error: Empty input file found" << std::endl; return 1; }

        // partial multi-MUMs (a row without an offset in some document) cannot be merged: exit code 1, which the merge
        // driver reports as "Partial MUMs detected" (mumemto/merge_mums.py:156-162)
        const std::vector<mumsio::Mum> rows = ends_with(mums, ".bumbl") ? mumsio::parse_bumbl(mums, true) : mumsio::parse_mums(mums, true);
        std::string out;
        size_t count = 0;
        for (const mumsio::Mum& m : rows) {
            const uint64_t a = (uint64_t)m.offsets.at(0);
            const uint64_t b = std::min<uint64_t>(bases.size(), a + m.length);
            if (a > bases.size()) throw std::runtime_error("MUM " + std::to_string(count) + " starts beyond the first document");
            out += ">mum_" + std::to_string(count++) + "\n";
            out.append(reinterpret_cast<const char*>(bases.data()) + a, b - a);
            if (terminator) out += '#';
            out += '\n';
        }
        std::ofstream of(output, std::ios::binary);
        if (!of) { std::cerr << "Error: cannot write " << output << std::endl; return 1; }
        of.write(out.data(), (std::streamsize)out.size());
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
