// Driver (ours) around the REFERENCE's include/newscan.hpp, compiled from
// /root/reference by oracle/Makefile into oracle/_ref/newscan_ref.
// Usage: newscan_ref <w> <p> <out_prefix> < records
// stdin: one record per line; a line "F" = process_string(rec) forward,
//        "R" = process_string_revcomp(rec); i.e. lines are "<F|R> <string>".
// Writes <out_prefix>.dict and <out_prefix>.parse (u32 LE), exactly the
// in-memory outputs of pfparser::finish_parse (newscan.hpp:357-423).
#include <newscan.hpp>
#include <cstdio>
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s w p out_prefix < lines\n", argv[0]); return 2; }
    size_t w = (size_t)atoi(argv[1]), p = (size_t)atoi(argv[2]);
    std::string prefix = argv[3];
    pfparser parser(prefix, w, p, true, false);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.size() < 2) { if (line == "F" || line == "R") continue; else continue; }
        std::string rec = line.substr(2);
        if (line[0] == 'F') parser.process_string(rec);
        else if (line[0] == 'R') parser.process_string_revcomp(rec);
    }
    parser.finish_parse();
    auto dict = parser.take_dict_data();
    auto parse = parser.take_parse_data();
    FILE *fd = fopen((prefix + ".dict").c_str(), "wb");
    fwrite(dict.data(), 1, dict.size(), fd); fclose(fd);
    FILE *fp = fopen((prefix + ".parse").c_str(), "wb");
    fwrite(parse.data(), 4, parse.size(), fp); fclose(fp);
    return 0;
}
