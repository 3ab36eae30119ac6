#include "fasta.hpp"

#include <zlib.h>

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace mmt {
namespace {

class LineReader {
public:
    explicit LineReader(const std::string& path) : f_(gzopen(path.c_str(), "r")) {
        if (!f_) throw std::runtime_error("cannot open " + path);
        gzbuffer(f_, 1 << 20);
    }
    ~LineReader() { if (f_) gzclose(f_); }
    // next byte or -1
    int getc() {
        if (pos_ >= end_) {
            if (eof_) return -1;
            int n = gzread(f_, buf_, sizeof(buf_));
            if (n < 0) throw std::runtime_error("read error in compressed stream");
            if (n == 0) { eof_ = true; return -1; }
            pos_ = 0; end_ = n;
        }
        return (unsigned char)buf_[pos_++];
    }
    // appends the rest of the current line (without '\n') to out; returns false at EOF before any byte
    void rest_of_line(std::vector<uint8_t>* out, std::string* sout, uint64_t* count) {
        while (true) {
            if (pos_ >= end_) {
                if (eof_) return;
                int n = gzread(f_, buf_, sizeof(buf_));
                if (n < 0) throw std::runtime_error("read error in compressed stream");
                if (n == 0) { eof_ = true; return; }
                pos_ = 0; end_ = n;
            }
            const char* nlp = static_cast<const char*>(std::memchr(buf_ + pos_, '\n', (size_t)(end_ - pos_)));
            const int i = nlp ? (int)(nlp - buf_) : end_;
            if (out) out->insert(out->end(), buf_ + pos_, buf_ + i);
            if (sout) sout->append(buf_ + pos_, buf_ + i);
            if (count) *count += (uint64_t)(i - pos_);
            bool nl = i < end_;
            pos_ = nl ? i + 1 : i;
            if (nl) return;
        }
    }

    bool at_eof() { if (pos_ < end_) return false; int c = getc(); if (c < 0) return true; pos_--; return false; }

private:
    gzFile f_;
    char buf_[1 << 18];
    int pos_ = 0, end_ = 0;
    bool eof_ = false;
};

}  // namespace

FastaDoc read_fasta(const std::string& path, std::vector<uint8_t>& bases) {
    FastaDoc doc;
    doc.path = path;
    {   // one allocation instead of a doubling vector: a plain file holds at most its size in bases, a gzip'ed one
        // about four times that (2 bits of entropy per base)
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && st.st_size > 0) {
            unsigned char magic[2] = {0, 0};
            if (FILE* f = std::fopen(path.c_str(), "rb")) { (void)!std::fread(magic, 1, 2, f); std::fclose(f); }
            const bool gz = magic[0] == 0x1f && magic[1] == 0x8b;
            bases.reserve(bases.size() + (size_t)st.st_size * (gz ? 4 : 1) + 16);
        }
    }
    LineReader in(path);
    int c = in.getc();
    // skip to the first header line
    while (c >= 0 && c != '>' && c != '@') c = in.getc();
    while (c == '>' || c == '@') {
        std::string header;
        in.rest_of_line(nullptr, &header, nullptr);
        if (!header.empty() && header.back() == '\r') header.pop_back();
        size_t k = 0;
        while (k < header.size() && !isspace((unsigned char)header[k])) k++;
        doc.names.push_back(header.substr(0, k));
        const size_t start = bases.size();
        // sequence lines
        while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            bases.push_back((uint8_t)c);
            in.rest_of_line(&bases, nullptr, nullptr);
            if (bases.size() - start > 1 && bases.back() == '\r') bases.pop_back();
        }
        const uint64_t len = bases.size() - start;
        doc.lengths.push_back(len);
        doc.total += len;
        if (c == '+') {   // FASTQ: skip the '+' line and as many quality characters as bases
            in.rest_of_line(nullptr, nullptr, nullptr);
            uint64_t q = 0;
            while (q < len && !in.at_eof()) {
                std::string tmp;
                in.rest_of_line(nullptr, &tmp, nullptr);
                if (!tmp.empty() && tmp.back() == '\r') tmp.pop_back();
                q += tmp.size();
            }
            c = in.getc();
            while (c >= 0 && c != '>' && c != '@') c = in.getc();
        }
    }
    return doc;
}

}  // namespace mmt
