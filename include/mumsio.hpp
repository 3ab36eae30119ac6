// mumsio.hpp -- result types of the native C++ API and readers/writers for the
// .mums (text) and .bumbl (binary) formats.
//
// Type layout is that of the reference's include/mumsio.hpp:17-28 (Mum, Mem), since
// those structs are part of the C++ API boundary (mumemto_library/mumemto_api.hpp:31-43).
// File formats follow the reference writers/readers:
//   .mums  : "LEN\tOFF_0,...,OFF_{N-1}\tS_0,...,S_{N-1}\n", '+'/'-' strands, an absent
//            document = empty field (mem_finder.hpp:406-426; reader mumsio.hpp:40-94)
//   .bumbl : u16 flags (bit15 length32, bit14 coll_blocks, bit13 partial) | u64 n_seqs |
//            u64 n_mums | lengths (u32 or u16) | i64 starts[n_mums][n_seqs] (-1 absent) |
//            strand bits, MSB first, '+' = 1 | optional blocks   (mumsio.hpp:96-194, :328-385)
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mumsio {

struct Mum {
    uint32_t length;
    std::vector<int64_t> offsets;  // -1 = document absent (partial MUM)
    std::vector<uint8_t> strands;  // 1 = '+', 0 = '-'
};

struct Mem {
    uint32_t length;
    std::vector<int64_t> offsets;
    std::vector<size_t> seq_ids;
    std::vector<uint8_t> strands;
};

inline std::vector<std::string> split_keep_empty(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        size_t b = s.find(sep, a);
        if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

// noPartials: throw on an empty offset (the anchor merge needs strict multi-MUMs).
inline std::vector<Mum> parse_mums(const std::string& path, bool noPartials = true) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Failed to open MUMs file: " + path);
    std::vector<Mum> rows;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f = split_keep_empty(line, '\t');
        if (f.size() < 3 || f[0].empty()) throw std::runtime_error("Malformed MUMs line: " + line);
        Mum m;
        m.length = static_cast<uint32_t>(std::stoul(f[0]));
        std::vector<std::string> offs = split_keep_empty(f[1], ','), sts = split_keep_empty(f[2], ',');
        // a trailing absent document leaves a trailing ',' in both fields: sizes still agree
        if (offs.size() != sts.size()) throw std::runtime_error("Offsets and strands column size mismatch in MUMs file");
        for (size_t i = 0; i < offs.size(); i++) {
            if (offs[i].empty()) {
                if (noPartials)
                    throw std::runtime_error("Cannot parse partial MUMs: empty offset encountered. Filter to strict MUMs.");
                m.offsets.push_back(-1);
                m.strands.push_back(0);
            } else {
                m.offsets.push_back(static_cast<int64_t>(std::stoll(offs[i])));
                m.strands.push_back(sts[i] == "+" ? 1 : 0);
            }
        }
        rows.push_back(std::move(m));
    }
    return rows;
}

inline std::vector<Mum> parse_bumbl(const std::string& path, bool noPartials = true) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("Failed to open bumbl file: " + path);
    auto need = [&](void* dst, size_t n) {
        in.read(static_cast<char*>(dst), static_cast<std::streamsize>(n));
        if (static_cast<size_t>(in.gcount()) != n) throw std::runtime_error("Unexpected EOF while reading bumbl file");
    };
    uint16_t flags = 0; uint64_t n_seqs = 0, n_mums = 0;
    need(&flags, 2); need(&n_seqs, 8); need(&n_mums, 8);
    const bool partial = flags & (1u << 13), len32 = flags & (1u << 15);
    if (noPartials && partial) throw std::runtime_error("Cannot parse partial bumbl: header indicates partial MUMs");
    std::vector<uint32_t> len(n_mums);
    if (len32) { if (n_mums) need(len.data(), n_mums * 4); }
    else { std::vector<uint16_t> l16(n_mums); if (n_mums) need(l16.data(), n_mums * 2); for (size_t i = 0; i < n_mums; i++) len[i] = l16[i]; }
    const size_t total = static_cast<size_t>(n_mums * n_seqs);
    std::vector<int64_t> starts(total);
    std::vector<uint8_t> bits((total + 7) / 8);
    if (total) { need(starts.data(), total * 8); need(bits.data(), bits.size()); }
    std::vector<Mum> rows(n_mums);
    for (size_t r = 0; r < n_mums; r++) {
        rows[r].length = len[r];
        rows[r].offsets.assign(starts.begin() + r * n_seqs, starts.begin() + (r + 1) * n_seqs);
        rows[r].strands.resize(n_seqs);
        for (size_t c = 0; c < n_seqs; c++) {
            size_t i = r * n_seqs + c;
            if (noPartials && starts[i] == -1) throw std::runtime_error("Cannot parse partial bumbl: -1 start encountered");
            rows[r].strands[c] = (bits[i / 8] >> (7 - (i % 8))) & 1u;
        }
    }
    return rows;
}

inline std::string serialize_mum(const Mum& m) {
    std::string s = std::to_string(m.length) + "\t";
    for (size_t i = 0; i < m.offsets.size(); i++) { s += std::to_string(m.offsets[i]); if (i + 1 < m.offsets.size()) s += ","; }
    s += "\t";
    for (size_t i = 0; i < m.strands.size(); i++) { s += m.strands[i] ? "+" : "-"; if (i + 1 < m.strands.size()) s += ","; }
    return s;
}

inline std::string serialize_mem(const Mem& m) {
    std::string s = std::to_string(m.length) + "\t";
    for (size_t i = 0; i < m.offsets.size(); i++) { s += std::to_string(m.offsets[i]); if (i + 1 < m.offsets.size()) s += ","; }
    s += "\t";
    for (size_t i = 0; i < m.seq_ids.size(); i++) { s += std::to_string(m.seq_ids[i]); if (i + 1 < m.seq_ids.size()) s += ","; }
    s += "\t";
    for (size_t i = 0; i < m.strands.size(); i++) { s += m.strands[i] ? "+" : "-"; if (i + 1 < m.strands.size()) s += ","; }
    return s;
}

inline void write_mums(const std::vector<Mum>& rows, const std::string& path) {
    std::ofstream out(path);
    if (!out) throw std::runtime_error("Failed to open output file: " + path);
    for (const Mum& m : rows) out << serialize_mum(m) << "\n";
}

inline void write_bumbl(const std::vector<Mum>& rows, const std::string& path, bool partial = false,
                        bool coll_blocks = false) {
    const uint64_t n_mums = rows.size(), n_seqs = rows.empty() ? 0 : rows[0].offsets.size();
    std::vector<uint32_t> len(n_mums);
    std::vector<int64_t> starts(n_mums * n_seqs);
    std::vector<uint8_t> bits((n_mums * n_seqs + 7) / 8, 0);
    for (size_t r = 0; r < n_mums; r++) {
        len[r] = rows[r].length;
        for (size_t c = 0; c < n_seqs; c++) {
            size_t i = r * n_seqs + c;
            starts[i] = rows[r].offsets[c];
            if (starts[i] == -1) partial = true;
            if (rows[r].strands[c]) bits[i / 8] |= static_cast<uint8_t>(1u << (7 - (i % 8)));
        }
    }
    uint16_t flags = static_cast<uint16_t>(1u << 15);
    if (partial) flags |= static_cast<uint16_t>(1u << 13);
    if (coll_blocks) flags |= static_cast<uint16_t>(1u << 14);
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("Failed to open output file: " + path);
    out.write(reinterpret_cast<const char*>(&flags), 2);
    out.write(reinterpret_cast<const char*>(&n_seqs), 8);
    out.write(reinterpret_cast<const char*>(&n_mums), 8);
    out.write(reinterpret_cast<const char*>(len.data()), static_cast<std::streamsize>(n_mums * 4));
    out.write(reinterpret_cast<const char*>(starts.data()), static_cast<std::streamsize>(starts.size() * 8));
    out.write(reinterpret_cast<const char*>(bits.data()), static_cast<std::streamsize>(bits.size()));
}

}  // namespace mumsio
