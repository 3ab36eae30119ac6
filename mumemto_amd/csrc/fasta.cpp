#include "fasta.hpp"

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <thread>

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
    template <class Sink>
    void rest_of_line(Sink* out, std::string* sout, uint64_t* count) {
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
            if (out) out->append(buf_ + pos_, buf_ + i);
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

// where the bases of a file go: a vector (grows) or a fixed slot of a larger buffer
struct VecSink {
    std::vector<uint8_t>& v;
    void append(const char* b, const char* e) { v.insert(v.end(), b, e); }
    void push(uint8_t c) { v.push_back(c); }
    size_t size() const { return v.size(); }
    uint8_t back() const { return v.back(); }
    void pop() { v.pop_back(); }
};
struct RawSink {
    uint8_t* p; size_t n, cap;
    // (the fast line copier of read_plain_fasta_blocks: `room` bytes may be written at the returned address, then advance())
    uint8_t* direct(size_t& room) { room = cap - n; return p + n; }
    void advance(size_t k) { n += k; }
    void need(size_t k) { if (n + k > cap) throw std::runtime_error("FASTA slot overflow (file changed while reading?)"); }
    void append(const char* b, const char* e) { const size_t k = (size_t)(e - b); need(k); std::memcpy(p + n, b, k); n += k; }
    void push(uint8_t c) { need(1); p[n++] = c; }
    size_t size() const { return n; }
    uint8_t back() const { return p[n - 1]; }
    void pop() { n--; }
};
struct NullSink { void append(const char*, const char*) {} };
// the chunked mode of ReadHooks: the buffer is swapped when the NEXT byte needs room, so that the last byte written is
// always still in `buf` (a line's trailing '\r' is taken back after the line is complete)
struct ChunkAbort {};
struct ChunkSink {
    ChunkTarget& T;
    size_t cap;                                    // bases the document's slot holds
    uint8_t* buf = nullptr;
    size_t fill = 0;
    uint64_t flushed = 0;
    void room() {
        if (buf && fill < T.chunk) return;
        uint8_t* next = T.swap(buf, fill, flushed, true);
        if (!next) throw ChunkAbort();
        flushed += fill; fill = 0; buf = next;
    }
    void append(const char* b, const char* e) {
        size_t n = (size_t)(e - b);
        if (flushed + fill + n > cap) throw std::runtime_error("FASTA slot overflow (file changed while reading?)");
        while (n) {
            room();
            const size_t k = std::min(n, T.chunk - fill);
            std::memcpy(buf + fill, b, k);
            fill += k; b += k; n -= k;
        }
    }
    void push(uint8_t c) { const char ch = (char)c; append(&ch, &ch + 1); }
    uint8_t* direct(size_t& room) {                // (only inside the current buffer, and never past the slot)
        if (!buf || fill == 0) { room = 0; return nullptr; }
        room = std::min(T.chunk - fill, cap - (size_t)flushed - fill);
        return buf + fill;
    }
    void advance(size_t k) { fill += k; }
    size_t size() const { return (size_t)flushed + fill; }
    uint8_t back() const { return buf[fill - 1]; }
    void pop() { fill--; }
    void restart() { fill = 0; flushed = 0; }      // (the reader begins again from the first byte of the file: same slot, from offset 0)
    void finish() { (void)T.swap(buf, fill, flushed, false); flushed += fill; fill = 0; buf = nullptr; }
};

template <class Reader, class Sink>
static FastaDoc read_fasta_with(const std::string& path, Sink& bases) {
    FastaDoc doc;
    doc.path = path;
    Reader in(path);
    int c = in.getc();
    // skip to the first header line
    while (c >= 0 && c != '>' && c != '@') c = in.getc();
    while (c == '>' || c == '@') {
        std::string header;
        in.rest_of_line((NullSink*)nullptr, &header, nullptr);
        if (!header.empty() && header.back() == '\r') header.pop_back();
        size_t k = 0;
        while (k < header.size() && !isspace((unsigned char)header[k])) k++;
        doc.names.push_back(header.substr(0, k));
        const size_t start = bases.size();
        // sequence lines
        while ((c = in.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            bases.push((uint8_t)c);
            in.rest_of_line(&bases, nullptr, nullptr);
            if (bases.size() - start > 1 && bases.back() == '\r') bases.pop();
        }
        const uint64_t len = bases.size() - start;
        doc.lengths.push_back(len);
        doc.total += len;
        if (c == '+') {   // FASTQ: skip the '+' line and as many quality characters as bases
            in.rest_of_line((NullSink*)nullptr, nullptr, nullptr);
            uint64_t q = 0;
            while (q < len && !in.at_eof()) {
                std::string tmp;
                in.rest_of_line((NullSink*)nullptr, &tmp, nullptr);
                if (!tmp.empty() && tmp.back() == '\r') tmp.pop_back();
                q += tmp.size();
            }
            c = in.getc();
            while (c >= 0 && c != '>' && c != '@') c = in.getc();
        }
    }
    return doc;
}

// Plain FASTA files: read() in blocks that stay in the cache, the bases of every line go straight to the file's slot
// (memchr + memcpy).  The stream reader above spends ~30 ms of CPU per 64 MB on its layers (zlib's pass-through copy, a
// refill check per character class) -- what matters when the process may use 16 cores' worth of time per 100 ms.
// Anything that is not plain multi-FASTA ('@' records, '+' lines) returns false and goes through the stream reader.
// Sequence lines, 32 bytes at a time: every vector is stored where the line's bases go and the write position moves on by
// the bytes before the first '\n' in it (the next store covers what was written past that point).  A 60-base line costs two
// loads, two compares and two stores instead of a memchr and a memcpy call: 98 M lines of a 6 GB collection were ~3.4 s of CPU
// in those calls, 0.2 s on each of sixteen reader threads -- the input phase of a one-shot run.  Leaves at anything that is
// not a plain sequence line ('>', '+', '@' at a line start), within 32 bytes of the end of the block or of the room at `w`.
// at_line_start in/out; rec_bytes = bases of the current record before `w`.  Returns the bytes consumed from q.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2")))
static size_t copy_sequence_lines_avx2(const char* q0, const char* e, uint8_t* w0, size_t room, bool& at_line_start,
                                       size_t rec_bytes, size_t& written) {
    const char* q = q0;
    uint8_t* w = w0;
    uint8_t* const w_end = w0 + (room >= 32 ? room - 32 : 0);
    const __m256i nl = _mm256_set1_epi8('\n');
    for (;;) {
        if (at_line_start) {
            if (q >= e) break;
            const char c = *q;
            if (c == '\n') { q++; continue; }
            if (c == '>' || c == '+' || c == '@') break;
            at_line_start = false;
        }
        if (q + 32 > e || w > w_end) break;
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(q));
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
        _mm256_storeu_si256(reinterpret_cast<__m256i*>(w), v);
        if (!m) { q += 32; w += 32; continue; }
        const uint32_t k = (uint32_t)__builtin_ctz(m);
        if (k == 0 && w == w0) break;                      // (the byte before this line end is not ours to look at: the caller's path)
        w += k; q += k + 1;
        if (rec_bytes + (size_t)(w - w0) > 1 && w[-1] == '\r') w--;
        at_line_start = true;
    }
    written = (size_t)(w - w0);
    return (size_t)(q - q0);
}
static bool have_avx2() { static const bool yes = __builtin_cpu_supports("avx2") && !std::getenv("MUMEMTO_NO_AVX2"); return yes; }
#else
static size_t copy_sequence_lines_avx2(const char*, const char*, uint8_t*, size_t, bool&, size_t, size_t& written) { written = 0; return 0; }
static bool have_avx2() { return false; }
#endif

// (Sink: RawSink -- the file's slot in the arena -- or ChunkSink)
template <class Sink>
static bool read_plain_fasta_blocks(const std::string& path, Sink& out, FastaDoc& doc) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    static thread_local std::vector<char> block(1u << 20);
    enum { BEFORE_FIRST, IN_HEADER, LINE_START, IN_LINE } state = BEFORE_FIRST;
    size_t rec = out.size();
    std::string header;
    bool open_record = false;
    doc = FastaDoc();
    doc.path = path;
    auto end_header = [&]() {
        if (!header.empty() && header.back() == '\r') header.pop_back();
        size_t k = 0;
        while (k < header.size() && !isspace((unsigned char)header[k])) k++;
        doc.names.push_back(header.substr(0, k));
        rec = out.size(); open_record = true;
    };
    auto end_record = [&]() {
        const uint64_t len = (uint64_t)(out.size() - rec);
        doc.lengths.push_back(len); doc.total += len; open_record = false;
    };
    auto end_line = [&]() { if (out.size() - rec > 1 && out.back() == '\r') out.pop(); };
    for (;;) {
        const ssize_t got = ::read(fd, block.data(), block.size());
        if (got < 0) return false;
        if (got == 0) break;
        const char* p = block.data();
        const char* const e = p + got;
        while (p < e) {
            if ((state == LINE_START || state == IN_LINE) && e - p >= 64 && have_avx2()) {
                size_t room = 0, written = 0;
                uint8_t* w = out.direct(room);
                if (w && room >= 96) {
                    bool at_start = state == LINE_START;
                    const size_t used = copy_sequence_lines_avx2(p, e, w, room, at_start, out.size() - rec, written);
                    out.advance(written);
                    p += used;
                    state = at_start ? LINE_START : IN_LINE;
                    if (p >= e) break;
                }
            }
            if (state == BEFORE_FIRST) {
                while (p < e && *p != '>' && *p != '@') p++;
                if (p == e) break;
                if (*p == '@') return false;
                header.clear(); state = IN_HEADER; p++;
            } else if (state == IN_HEADER) {
                const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(e - p)));
                header.append(p, nl ? nl : e);
                if (!nl) break;
                end_header(); state = LINE_START; p = nl + 1;
            } else if (state == LINE_START) {
                const char c = *p;
                if (c == '>') { end_record(); header.clear(); state = IN_HEADER; p++; }
                else if (c == '+' || c == '@') return false;
                else if (c == '\n') p++;
                else state = IN_LINE;
            } else {
                const char* nl = static_cast<const char*>(std::memchr(p, '\n', (size_t)(e - p)));
                const char* le = nl ? nl : e;
                if (out.size() + (size_t)(le - p) > out.cap) return false;
                out.append(p, le);
                if (!nl) break;
                end_line(); state = LINE_START; p = nl + 1;
            }
        }
    }
    if (state == IN_HEADER) end_header();
    if (state == IN_LINE) end_line();
    if (open_record) end_record();
    return true;
}

// size of the file and whether it is gzip-compressed
// (a FIFO / process substitution / /dev/stdin is not probed -- the probe would consume its first bytes -- and reports
// "compressed": it then goes through the stream reader, which handles both, into a vector of its own)
static bool file_info(const std::string& path, size_t& size) {
    struct stat st;
    size = 0;
    if (stat(path.c_str(), &st) == 0 && !S_ISREG(st.st_mode)) return true;
    if (stat(path.c_str(), &st) == 0 && st.st_size > 0) size = (size_t)st.st_size;
    if (size == 0) return true;                  // (an empty or unreadable file: the stream reader reports it)
    unsigned char magic[2] = {0, 0};
    if (FILE* f = std::fopen(path.c_str(), "rb")) { (void)!std::fread(magic, 1, 2, f); std::fclose(f); }
    return magic[0] == 0x1f && magic[1] == 0x8b;
}

FastaDoc read_fasta(const std::string& path, std::vector<uint8_t>& bases) {
    // one allocation instead of a doubling vector: a plain file holds at most its size in bases, a gzip'ed one about
    // four times that (2 bits of entropy per base)
    size_t size = 0;
    const bool gz = file_info(path, size);
    if (size) bases.reserve(bases.size() + size * (gz ? 4 : 1) + 16);
    VecSink sink{bases};
    return read_fasta_with<LineReader>(path, sink);
}

FastaDoc read_fasta_replace(const std::string& path, std::vector<uint8_t>& bases) {
    size_t size = 0;
    const bool gz = file_info(path, size);
    if (!gz && size && !std::getenv("MUMEMTO_STREAM_READER")) {
        bases.resize(size + 64);                    // (a plain file holds at most its size in bases)
        FastaDoc doc;
        RawSink raw{bases.data(), 0, bases.size()};
        if (read_plain_fasta_blocks(path, raw, doc)) { bases.resize(raw.n); return doc; }
    }
    bases.clear();
    return read_fasta(path, bases);
}

HostArena::~HostArena() { if (p_) munmap(p_, cap_); }
uint8_t* HostArena::ensure(size_t bytes) {
    if (bytes <= cap_) return p_;
    if (p_) { munmap(p_, cap_); p_ = nullptr; cap_ = 0; }
    const size_t want = (bytes + bytes / 16 + (2u << 20)) & ~(size_t)((2u << 20) - 1);
    void* m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::runtime_error("cannot allocate " + std::to_string(want >> 20) + " MiB of host memory");
    (void)madvise(m, want, MADV_HUGEPAGE);          // 2 MiB pages: 6 GB of bases are 3,000 page faults instead of 1.5 M
    p_ = static_cast<uint8_t*>(m); cap_ = want;
    return p_;
}

// CPUs this process may really use: the cgroup's CPU quota if there is one (a container with 16 CPUs' worth of time on a
// 256-thread host is throttled for the rest of every 100 ms period once 94 reader threads have spent it in 17 ms -- and
// with them the thread that feeds the GPU), else what the machine has.  MUMEMTO_READ_THREADS overrides.
size_t reader_threads() {
    if (const char* e = std::getenv("MUMEMTO_READ_THREADS")) return (size_t)std::max(1, std::atoi(e));
    size_t n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64] = {0};
        unsigned long long period = 0;
        if (std::fscanf(f, "%63s %llu", quota, &period) == 2 && period && std::strcmp(quota, "max") != 0) {
            const unsigned long long q = std::strtoull(quota, nullptr, 10);
            if (q) n = std::min<size_t>(n, (size_t)((q + period - 1) / period));
        }
        std::fclose(f);
    }
    return std::max<size_t>(n, 1);
}

long read_fasta_collection(const std::vector<std::string>& inputs, std::vector<FastaDoc>& docs, HostArena& arena,
                           HostDocs& out, const ReadHooks* hooks) {
    const size_t N = inputs.size();
    docs.assign(N, FastaDoc());
    out.ptr.assign(N, nullptr); out.len.assign(N, 0); out.owned.clear(); out.owned.resize(N);
    // plain files get a slot of their size in the arena (bases <= bytes of the file); compressed ones a vector
    std::vector<size_t> slot(N + 1, 0);
    std::vector<char> gz(N, 0);
    std::vector<size_t> fsize(N, 0);
    for (size_t i = 0; i < N; i++) {
        size_t size = 0;
        gz[i] = file_info(inputs[i], size) ? 1 : 0;
        fsize[i] = size;
        slot[i + 1] = slot[i] + (gz[i] ? 0 : ((size + 64 + 4095) & ~(size_t)4095));
    }
    bool all_plain = true;
    for (size_t i = 0; i < N; i++) all_plain = all_plain && !gz[i];
    const bool chunked = hooks && hooks->plan_chunks && hooks->chunks && all_plain && N > 0 &&
                         hooks->plan_chunks(slot[N], slot, all_plain);
    uint8_t* base = chunked ? nullptr : arena.ensure(slot[N] + 4096);
    if (!chunked && hooks && hooks->layout) hooks->layout(base, slot[N], slot, all_plain);
    std::vector<std::string> err(N);
    const size_t n_thr = std::min<size_t>(N, reader_threads());
    std::atomic<size_t> next{0};
    std::atomic<bool> aborted{false};
    auto work = [&]() {
        for (size_t i = next++; i < N; i = next++) {
            try {
                if (chunked) {
                    if (aborted.load()) continue;
                    ChunkTarget target = hooks->chunks(i);
                    ChunkSink sink{target, slot[i + 1] - slot[i]};
                    if (std::getenv("MUMEMTO_STREAM_READER") || !read_plain_fasta_blocks(inputs[i], sink, docs[i])) {
                        sink.restart();                  // ('@' records, '+' lines: once more through the stream reader)
                        docs[i] = read_fasta_with<LineReader>(inputs[i], sink);
                    }
                    sink.finish();
                    out.ptr[i] = nullptr; out.len[i] = sink.size();
                    if (hooks->ready) hooks->ready(i, out.len[i]);
                } else if (gz[i]) {
                    docs[i] = read_fasta(inputs[i], out.owned[i]);
                    out.ptr[i] = out.owned[i].data(); out.len[i] = out.owned[i].size();
                } else {
                    RawSink raw{base + slot[i], 0, slot[i + 1] - slot[i]};
                    if (!std::getenv("MUMEMTO_STREAM_READER") && read_plain_fasta_blocks(inputs[i], raw, docs[i])) {
                        out.ptr[i] = base + slot[i]; out.len[i] = raw.n;
                    } else {
                        RawSink sink{base + slot[i], 0, slot[i + 1] - slot[i]};
                        docs[i] = read_fasta_with<LineReader>(inputs[i], sink);
                        out.ptr[i] = sink.p; out.len[i] = sink.n;
                    }
                    if (hooks && hooks->ready) hooks->ready(i, out.len[i]);
                }
            } catch (const ChunkAbort&) { aborted.store(true);
            } catch (const std::exception& e) { err[i] = e.what(); }
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_thr; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (aborted.load()) return -2;                            // (the chunk target gave up: the caller reads again without chunks)
    for (size_t i = 0; i < N; i++) {
        if (!err[i].empty()) throw std::runtime_error(err[i]);
        if (docs[i].total == 0) return (long)i;             // ref_builder.cpp:249-252 + pfp_mum.cpp:68-71
    }
    return -1;
}

long read_fasta_files(const std::vector<std::string>& inputs, std::vector<FastaDoc>& docs, HostBytes& bases,
                      std::vector<uint64_t>& doc_len) {
    docs.assign(inputs.size(), FastaDoc());
    doc_len.clear();
    std::vector<std::vector<uint8_t>> part(inputs.size());
    std::vector<std::string> err(inputs.size());
    const size_t n_thr = std::min<size_t>(inputs.size(), reader_threads());
    auto on_all_threads = [&](const std::function<void(size_t)>& per_file) {
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t i = next++; i < inputs.size(); i = next++) per_file(i); };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_thr; t++) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    };
    on_all_threads([&](size_t i) {
        try { docs[i] = read_fasta(inputs[i], part[i]); }
        catch (const std::exception& e) { err[i] = e.what(); }
    });
    std::vector<size_t> at(inputs.size() + 1, 0);
    long empty = -1;
    for (size_t i = 0; i < inputs.size(); i++) {
        if (!err[i].empty()) throw std::runtime_error(err[i]);
        if (docs[i].total == 0 && empty < 0) empty = (long)i;      // ref_builder.cpp:249-252 + pfp_mum.cpp:68-71
        at[i + 1] = at[i] + part[i].size();
        doc_len.push_back(docs[i].total);
    }
    if (empty >= 0) return empty;
    bases.allocate(at.back());
    on_all_threads([&](size_t i) {
        if (!part[i].empty()) std::memcpy(bases.data() + at[i], part[i].data(), part[i].size());
        std::vector<uint8_t>().swap(part[i]);
    });
    return -1;
}

void write_file_bytes(const std::string& path, const void* data, size_t n) {
    // one writer: write() on a file is serialised by its inode lock, and filling a shared mapping of the file from
    // sixteen threads was slower than this loop on tmpfs (0.33 against 0.22 s for the 894 MB of the C3 stand-in)
    const int fd = ::open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + path);
    const char* src = static_cast<const char*>(data);
    size_t at = 0;
    while (at < n) {
        const ssize_t w = ::write(fd, src + at, std::min<size_t>(n - at, (size_t)1 << 30));
        if (w <= 0) { ::close(fd); throw std::runtime_error("short write to " + path); }
        at += (size_t)w;
    }
    if (::close(fd) != 0) throw std::runtime_error("cannot close " + path);
}

void write_lengths_file(const std::string& prefix, const std::vector<FastaDoc>& docs) {
    std::ofstream out(prefix + ".lengths");
    for (const auto& d : docs) {
        const std::string canon = std::filesystem::canonical(d.path).string();
        out << canon << " * " << d.total << std::endl;
        for (size_t r = 0; r < d.names.size(); r++) out << canon << " " << d.names[r] << " " << d.lengths[r] << std::endl;
    }
}

}  // namespace mmt
