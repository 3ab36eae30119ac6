// Driver (ours) around the REFERENCE's header-only C++ wrapper
// mumemto_library/mumemto.hpp (+ its mumemto_api.hpp / mumemto.h / mumsio.hpp), compiled
// unmodified from /root/reference and linked against THIS repository's libmumemto.so:
// the drop-in claim for C++ callers.  Usage: dropin_wrapper <mode> <min_len> <revcomp> <out_path>
//   stdin: one document per line, records separated by ','.   mode: mum | mem
#include <mumemto.hpp>

#include <iostream>
#include <sstream>

int main(int argc, char** argv) {
    if (argc < 5) { std::cerr << "usage: dropin_wrapper mum|mem min_len revcomp out_path\n"; return 2; }
    const std::string mode = argv[1];
    const uint32_t min_len = (uint32_t)std::stoul(argv[2]);
    const bool revcomp = std::stoi(argv[3]) != 0;
    std::vector<std::vector<std::string>> docs;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::vector<std::string> recs;
        std::stringstream ss(line);
        std::string r;
        while (std::getline(ss, r, ',')) recs.push_back(r);
        docs.push_back(recs);
    }
    try {
        if (mode == "mum") {
            auto r = mumemto_cxx::mum(docs, min_len, revcomp);
            r.write_mums(argv[4]);
            std::cout << r.num_docs() << " " << r.num_matches() << "\n";
        } else {
            auto r = mumemto_cxx::mem(docs, min_len, revcomp, 0, 0, 2);
            r.write_mems(argv[4]);
            std::cout << r.num_docs() << " " << r.num_matches() << "\n";
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
