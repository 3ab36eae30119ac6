// merge_main.cpp -- anchor_merge: `anchor_merge <inputs...> -o <out>[.mums|.bumbl] [-v]`
// (reference CLI contract: src/merge_candidates.cpp:170-255).  Inputs end in .mums
// or .bumbl and sit next to a PREFIX.athresh; output rows come out in anchor order
// with the merged PREFIX.athresh.  The O(anchor length) walk of every fold step
// runs on the GPU (mmt::anchor_merge).
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iostream>

#include "../../include/mumsio.hpp"
#include "merge.hpp"

namespace fs = std::filesystem;

static bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

struct Loaded {
    std::vector<uint32_t> length;
    std::vector<int64_t> offsets;
    std::vector<uint8_t> strands;
    std::vector<uint16_t> thresh;
    size_t n_docs = 0;
};

static Loaded load(const std::string& path) {
    const bool bumbl = ends_with(path, ".bumbl");
    const std::string base = path.substr(0, path.size() - (bumbl ? 6 : 5));
    Loaded L;
    std::ifstream t(base + ".athresh", std::ios::binary | std::ios::ate);
    if (!t) throw std::runtime_error("Could not find threshold file: " + base + ".athresh");
    std::streamsize sz = t.tellg();
    if (sz % 2) throw std::runtime_error("File size is not a multiple of uint16_t");
    L.thresh.resize((size_t)sz / 2);
    t.seekg(0);
    t.read(reinterpret_cast<char*>(L.thresh.data()), sz);
    std::vector<mumsio::Mum> rows = bumbl ? mumsio::parse_bumbl(path, true) : mumsio::parse_mums(path, true);
    for (const auto& m : rows) {
        if (!L.n_docs) L.n_docs = m.offsets.size();
        L.length.push_back(m.length);
        L.offsets.insert(L.offsets.end(), m.offsets.begin(), m.offsets.end());
        L.strands.insert(L.strands.end(), m.strands.begin(), m.strands.end());
    }
    return L;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "Usage: " << argv[0] << " <input_paths>... -o <output_prefix> [-l min_len] [-v]" << std::endl; return 1; }
    std::vector<std::string> paths;
    std::string output = "merged";
    bool verbose = false;
    // -l: minimum length of a merged MUM.  The reference hard-codes 20 (src/merge_candidates.cpp:141), which is only
    // right for partitions made with the default -l; partitions made with another -l need the same value here.
    uint32_t min_len = 20;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) output = argv[++i];
        else if (a == "-l" && i + 1 < argc) min_len = (uint32_t)std::stoul(argv[++i]);
        else if (a == "-v") verbose = true;
        else paths.push_back(a);
    }
    try {
        if (paths.size() < 2) throw std::runtime_error("requires at least two input files");
        for (const auto& p : paths) {
            if (!ends_with(p, ".mums") && !ends_with(p, ".bumbl"))
                throw std::runtime_error("Invalid input: " + p + ". Inputs must explicitly end with .mums or .bumbl.");
            if (!fs::exists(p)) throw std::runtime_error("Could not find MUMs file: " + p);
        }
        std::vector<Loaded> parts;
        for (const auto& p : paths) { if (verbose) std::cerr << "reading " << p << std::endl; parts.push_back(load(p)); }
        std::vector<mmt_partition> mp(parts.size());
        for (size_t i = 0; i < parts.size(); i++) {
            mp[i].n_rows = parts[i].length.size(); mp[i].n_docs = parts[i].n_docs;
            mp[i].length = parts[i].length.data(); mp[i].offsets = parts[i].offsets.data();
            mp[i].strands = parts[i].strands.data(); mp[i].thresh = parts[i].thresh.data();
            mp[i].thresh_len = parts[i].thresh.size(); mp[i].thresh_on_device = 0; mp[i].rows_on_device = 0;
        }
        mmt::Engine eng(std::getenv("MUMEMTO_DEVICE") ? std::atoi(std::getenv("MUMEMTO_DEVICE")) : 0, nullptr);
        mmt::MergedRows m = mmt::anchor_merge(eng, mp.data(), mp.size(), min_len);
        mmt::download_merged(eng, m);
        bool out_bumbl = ends_with(output, ".bumbl"), out_mums = ends_with(output, ".mums");
        std::string out_path = output;
        if (!out_bumbl && !out_mums) { out_path += ".mums"; out_mums = true; }
        const std::string prefix = out_path.substr(0, out_path.size() - (out_bumbl ? 6 : 5));
        if (out_bumbl) {
            std::vector<mumsio::Mum> rows(m.length.size());
            for (size_t r = 0; r < rows.size(); r++) {
                rows[r].length = m.length[r];
                rows[r].offsets.assign(m.offsets.begin() + r * m.n_docs, m.offsets.begin() + (r + 1) * m.n_docs);
                rows[r].strands.assign(m.strands.begin() + r * m.n_docs, m.strands.begin() + (r + 1) * m.n_docs);
            }
            mumsio::write_bumbl(rows, out_path);
        } else {
            std::string text = mmt::format_merged(eng, m);
            std::ofstream f(out_path, std::ios::binary);
            f.write(text.data(), (std::streamsize)text.size());
        }
        std::ofstream th(prefix + ".athresh", std::ios::binary);
        th.write(reinterpret_cast<const char*>(m.thresh.data()), (std::streamsize)(m.thresh.size() * 2));
        std::cerr << "done." << std::endl;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
}
