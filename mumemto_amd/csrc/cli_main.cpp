// cli_main.cpp -- mumemto_exec: the reference CLI's contract for the hot path
// (src/pfp_mum.cpp:31-159 build_main): FASTA files in, PREFIX.mums | .mems |
// .bumbl + PREFIX.lengths (+ .athresh | .thresh/.thresh_rev, + .sa/.lcp/.bwt
// with -A) out.  All compute runs through libmumemto's GPU engine.
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <memory>
#include <iostream>
#include <thread>

#include <csignal>
#include <sys/wait.h>
#include <unistd.h>

#include "dist.hpp"
#include "engine.hpp"
#include "fasta.hpp"
#include "merge.hpp"
#include "options.hpp"

namespace fs = std::filesystem;
using namespace mmt;

static void log_line(const char* tag, const std::string& msg) {
    std::fprintf(stderr, "\033[32m[%s] \033[m%s\n", tag, msg.c_str());
}
static const auto g_start = std::chrono::steady_clock::now();
static double secs_since(std::chrono::steady_clock::time_point t0);
// MUMEMTO_TIMING=1: wall-clock marks (seconds since the process started) on stderr
static void mark(const char* what) {
    static const bool on = std::getenv("MUMEMTO_TIMING") != nullptr;
    if (on) std::fprintf(stderr, "[timing] %8.3f  %s\n", secs_since(g_start), what);
}
static double secs_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
static void write_file(const std::string& path, const void* data, size_t n) { write_file_bytes(path, data, n); }

// RefBuilder(prefix, use_rcomp) (src/ref_builder.cpp:140-169): document lengths from PREFIX.lengths --
// "path * total" lines (or the two-word "path total" form); per-record lines are skipped
static std::vector<uint64_t> read_lengths_file(const std::string& prefix) {
    const std::string name = prefix + ".lengths";
    std::ifstream in(name);
    if (!in) throw CliError{"Lengths file required for using intermediate files. File should match output prefix: " + name, 1};
    std::vector<uint64_t> lens;
    std::string line;
    while (std::getline(in, line)) {
        std::vector<std::string> words;
        size_t a = 0;
        while (a <= line.size()) {
            size_t b = line.find(' ', a);
            if (b == std::string::npos) b = line.size();
            if (b > a) words.push_back(line.substr(a, b - a));
            a = b + 1;
        }
        if (words.size() == 2) lens.push_back(std::stoull(words[1]));
        else if (words.size() == 3 && words[1] == "*") lens.push_back(std::stoull(words[2]));
    }
    if (lens.empty()) throw CliError{"no document lengths in " + name, 1};
    return lens;
}
static std::vector<uint8_t> read_whole_file(const std::string& path, const char* what) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw CliError{std::string("Error: opening ") + what + " file " + path, 1};
    std::vector<uint8_t> data((size_t)f.tellg());
    f.seekg(0);
    f.read(reinterpret_cast<char*>(data.data()), (std::streamsize)data.size());
    return data;
}
// -a PREFIX (src/pfp_mum.cpp:97-111, include/read_arrays.hpp:86-122): 40-bit little-endian SA / LCP, one BWT byte
// per entry.  The reference feeds the first |T| entries of the |T|+1 the writer produced to the match finder, i.e.
// the sentinel entry and all real suffixes but the last; the engine takes exactly those real suffixes.
static void load_stream_files(const std::string& prefix, uint64_t text_chars, std::vector<uint32_t>& sa,
                              std::vector<uint8_t>& sa_hi, std::vector<uint32_t>& lcp, std::vector<uint8_t>& bwt) {
    const std::vector<uint8_t> fsa = read_whole_file(prefix + ".sa", "SA"), flcp = read_whole_file(prefix + ".lcp", "LCP"),
                               fbwt = read_whole_file(prefix + ".bwt", "BWT");
    if (fsa.size() < text_chars * 5 || flcp.size() < text_chars * 5 || fbwt.size() < text_chars)
        throw CliError{"the arrays under " + prefix + " hold fewer than the " + std::to_string(text_chars) +
                       " entries the lengths file announces", 1};
    auto get40 = [](const std::vector<uint8_t>& b, uint64_t j) {
        uint64_t v = 0;
        for (int k = 4; k >= 0; k--) v = (v << 8) | b[j * 5 + k];
        return v;
    };
    const uint64_t entries = text_chars ? text_chars - 1 : 0;      // stream entries 1 .. |T|-1
    sa.resize(entries); lcp.resize(entries); bwt.resize(entries);
    const bool wide = text_chars >= NARROW_LIMIT;          // positions need the high byte (wide.hpp)
    if (wide) sa_hi.resize(entries);
    for (uint64_t j = 0; j < entries; j++) {
        const uint64_t s = get40(fsa, j + 1);
        uint64_t l = get40(flcp, j + 1);
        if (s > 0xffffffffull && !wide) throw CliError{"suffix array entry beyond the text", 1};
        if (l > LCP_CAP) l = LCP_CAP;                       // lengths are 32 bits in the results (mumsio.hpp:18,24)
        sa[j] = (uint32_t)s; lcp[j] = (uint32_t)l; bwt[j] = fbwt[j + 1];
        if (wide) sa_hi[j] = (uint8_t)(s >> 32);
    }
}
// -p PREFIX (src/pfp_mum.cpp:122-124, include/pfp.hpp:105-129): PREFIX.dict = sorted phrases, each closed by 0x01,
// the file by 0x00; PREFIX.parse = 1-based u32 ranks.  Consecutive phrases overlap by w characters; the parsed
// string is 0x02 T 0x02^w (include/newscan.hpp:357-423), so T falls out by concatenation.
static std::vector<uint8_t> text_from_parse(const std::string& prefix, size_t w) {
    const std::vector<uint8_t> dict = read_whole_file(prefix + ".dict", "dictionary"),
                               parse = read_whole_file(prefix + ".parse", "parse");
    std::vector<std::pair<size_t, size_t>> phrase;              // (offset, length) in dict
    size_t a = 0;
    for (size_t i = 0; i < dict.size(); i++) {
        if (dict[i] == 0x01) { phrase.emplace_back(a, i - a); a = i + 1; }
        else if (dict[i] == 0x00) break;
    }
    if (phrase.empty() || parse.size() % 4) throw CliError{"malformed " + prefix + ".dict / .parse", 1};
    std::vector<uint8_t> full;
    const size_t m = parse.size() / 4;
    for (size_t i = 0; i < m; i++) {
        uint32_t r;
        std::memcpy(&r, parse.data() + 4 * i, 4);
        if (r == 0 || r > phrase.size()) throw CliError{"parse rank outside the dictionary", 1};
        const auto& ph = phrase[r - 1];
        const size_t skip = i ? w : 0;
        if (ph.second < skip) throw CliError{"phrase shorter than the window: is -w the value the parse was made with?", 1};
        full.insert(full.end(), dict.begin() + ph.first + skip, dict.begin() + ph.first + ph.second);
    }
    if (full.size() < w + 1 || full[0] != 0x02) throw CliError{"the parse does not start with the padding symbol", 1};
    for (size_t i = 0; i < w; i++)
        if (full[full.size() - 1 - i] != 0x02) throw CliError{"the parse does not end with w padding symbols: is -w right?", 1};
    return std::vector<uint8_t>(full.begin() + 1, full.end() - w);
}

static void put40(std::vector<uint8_t>& b, uint64_t v) {
    for (int k = 0; k < 5; k++) b.push_back((uint8_t)(v >> (8 * k)));
}

// ---- --gpus N: one process per GPU -----------------------------------------------------------------------------------
// The launcher starts N copies of itself (--rank r --comm-file F, device r); rank 0 makes the RCCL id and leaves it in F.
// Strict multi-MUMs: the reference's own workflow (README.md:124-141: partitions that share the anchor, -M -n, then
// anchor_merge) with the files between the tools replaced by mmt::dist_merge -- rank r runs {anchor} + its share of the
// other documents (as partitions of its own if that share is larger than one suffix array), row tables and thresholds
// travel HBM -> HBM, rank 0 folds, re-sorts into direct-run order and writes PREFIX.mums / PREFIX.lengths (/ .athresh).
// Other modes (the reference refuses to merge them, include/pfp_mum.hpp:178-183): every rank builds the stream of the
// whole collection and scans its share of the suffix-array positions; rank 0 writes the concatenated bytes.
static bool want_streamed_input(const std::vector<std::string>& inputs, size_t copies);
static int launch_ranks(int argc, char** argv, const BuildOptions& o) {
    const std::string comm_file = o.output_prefix + ".comm." + std::to_string((long)getpid());
    std::remove(comm_file.c_str());
    // ONE decision for all ranks, taken before any of them exists: whether the input is streamed and whether the ranks of a
    // sharded run write pieces (no communicator) or gather (a communicator).  Each rank judging the host's free memory for
    // itself -- at another moment, while the others allocate -- could put ranks on both sides of a collective.
    {
        BuildOptions q = o;
        const bool mum_mode = q.validate();
        const std::vector<std::string> inputs = resolve_inputs(q);
        q.set_parameters(inputs.size(), mum_mode);
        const size_t N = inputs.size(), world = (size_t)o.gpus;
        const bool strict = mum_mode && (q.num_distinct_docs == 0 || (size_t)q.num_distinct_docs == N);
        std::vector<std::string> share = inputs;                  // the largest share: rank 0's (anchor + the first block)
        if (strict && N > 1) share.assign(inputs.begin(), inputs.begin() + 1 + ((N - 1) + world - 1) / world);
        const bool streamed = want_streamed_input(share, world);
        setenv("MUMEMTO_STREAM_INPUT", streamed ? "1" : "0", 0);            // (a value the user set stays)
        if (!strict) setenv("MUMEMTO_RANK_PIECES", streamed ? "1" : "0", 0);
    }
    // (pieces an earlier run that failed may have left behind must not be taken for this run's)
    for (int r = 0; r < o.gpus; r++)
        for (const char* ext : {".mums", ".mems", ".mums.tmp", ".mems.tmp"}) std::remove((o.output_prefix + ".rank" + std::to_string(r) + ext).c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < o.gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) throw CliError{"cannot start rank " + std::to_string(r), 1};
        if (pid == 0) {
            std::vector<std::string> args(argv, argv + argc);
            args.push_back("--rank"); args.push_back(std::to_string(r));
            args.push_back("--comm-file"); args.push_back(comm_file);
            std::vector<char*> av;
            for (auto& s : args) av.push_back(const_cast<char*>(s.c_str()));
            av.push_back(nullptr);
            if (!std::getenv("MUMEMTO_SHARE_DEVICE")) setenv("MUMEMTO_DEVICE", std::to_string(r).c_str(), 1);
            // the ranks share the node's CPUs (a container's quota): every reader takes its part
            if (!std::getenv("MUMEMTO_READ_THREADS"))
                setenv("MUMEMTO_READ_THREADS", std::to_string(std::max<size_t>(2, reader_threads() / (size_t)o.gpus)).c_str(), 1);
            execv("/proc/self/exe", av.data());
            std::_Exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (size_t left = kids.size(); left;) {
        int st = 0;
        const pid_t done = wait(&st);
        if (done < 0) break;
        left--;
        kids.erase(std::remove(kids.begin(), kids.end(), done), kids.end());     // (a reaped pid may be reused by another process)
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
        if (code && !rc) {                   // a rank failed: the others would wait for it in a collective for ever
            rc = code;
            for (pid_t k : kids) kill(k, SIGTERM);
        }
    }
    std::remove(comm_file.c_str());
    // the ranks of a sharded run (-k / -f modes) that wrote their rows as pieces, PREFIX.rankR.mems: rows come out in order of
    // their closing position and a rank's share is a range of the stream, so the pieces in rank order are the file of one GPU
    for (const char* ext : {".mums", ".mems"}) {
        const std::string first = o.output_prefix + ".rank0" + ext;
        if (rc || !fs::exists(first)) continue;
        // joined under a temporary name and renamed when every piece is in: a short write or a kill half way leaves no
        // plausible but truncated PREFIX.mems; the pieces go only after the joined file is complete
        const std::string final_name = o.output_prefix + ext, tmp = final_name + ".tmp";
        {
            std::ofstream all(tmp, std::ios::binary | std::ios::trunc);
            std::vector<char> buf(64u << 20);
            for (int r = 0; r < o.gpus && !rc; r++) {
                const std::string piece = o.output_prefix + ".rank" + std::to_string(r) + ext;
                std::ifstream in(piece, std::ios::binary);
                if (!in) { log_line("build_main", "the piece of rank " + std::to_string(r) + " is missing: " + piece); rc = 1; break; }
                while (in && all) { in.read(buf.data(), (std::streamsize)buf.size()); all.write(buf.data(), in.gcount()); }
                if (!all || in.bad()) { log_line("build_main", "could not write " + tmp + " (piece of rank " + std::to_string(r) + ")"); rc = 1; }
            }
            all.close();
            if (!rc && !all) { log_line("build_main", "could not write " + tmp); rc = 1; }
        }
        if (!rc && std::rename(tmp.c_str(), final_name.c_str()) != 0) { log_line("build_main", "could not rename " + tmp); rc = 1; }
        if (rc) { std::remove(tmp.c_str()); continue; }
        for (int r = 0; r < o.gpus; r++) std::remove((o.output_prefix + ".rank" + std::to_string(r) + ext).c_str());
    }
    return rc;
}

static void leave_id(const std::string& path, const uint8_t id[128]) {
    const std::string tmp = path + ".tmp";
    { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char*>(id), 128); }
    std::rename(tmp.c_str(), path.c_str());
}
static void fetch_id(const std::string& path, uint8_t id[128]) {
    for (int tries = 0; tries < 360000; tries++) {         // an hour: rank 0 writes it after its partition has run
        std::ifstream f(path, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(id), 128)) return;
        usleep(10000);
    }
    throw CliError{"rank 0 never left the communicator id in " + path, 1};
}

// ---- a collection that does not fit the HOST as bytes ----------------------------------------------------------------------
// The reference streams every FASTA file through its parser once and never holds the collection (src/ref_builder.cpp:211-314
// writes the text to a file, include/newscan.hpp:265-325 reads that).  Here: the lengths first (every file parsed once on the
// reader threads, the bases thrown away: PREFIX.lengths and the layout of the text need them), then the documents one at a
// time when the engine asks (Engine::run_supplied), the next files read ahead by one thread.  Chosen when the files together
// exceed half of what the host may still use (/proc/meminfo, the memory cgroup); MUMEMTO_STREAM_INPUT=1 | 0 forces either.
static uint64_t host_memory_available() {
    uint64_t avail = ~0ull;
    { std::ifstream f("/proc/meminfo"); std::string k; uint64_t v; std::string unit;
      while (f >> k >> v >> unit) if (k == "MemAvailable:") { avail = v * 1024; break; } }
    auto first_number = [](std::initializer_list<const char*> paths, uint64_t& out) {
        for (const char* q : paths) { std::ifstream f(q); std::string w; if (f >> w && w != "max") { out = std::strtoull(w.c_str(), nullptr, 10); return true; } }
        return false;
    };
    uint64_t cg_max = 0, cg_now = 0;
    if (first_number({"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"}, cg_max) && cg_max < (1ull << 60)) {
        (void)first_number({"/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"}, cg_now);
        avail = std::min<uint64_t>(avail, cg_max > cg_now ? cg_max - cg_now : 0);
    }
    return avail;
}
static bool want_streamed_input(const std::vector<std::string>& inputs, size_t copies) {       // copies: processes that each hold it
    if (const char* e = std::getenv("MUMEMTO_STREAM_INPUT")) return std::atoi(e) != 0;
    uint64_t bound = 0;                                      // bases at most: a compressed file counted four times
    for (const auto& p : inputs) {
        std::error_code ec;
        const uint64_t sz = (uint64_t)fs::file_size(p, ec);
        if (ec) continue;
        bound += (p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0) ? 4 * sz : sz;
    }
    return (double)bound * (double)copies > 0.5 * (double)host_memory_available();
}
struct StreamedInput {
    std::vector<std::string> paths;
    std::vector<FastaDoc> docs;
    std::vector<uint64_t> len;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    std::map<size_t, std::vector<uint8_t>> ready;               // documents read ahead
    size_t expect = 0;                                           // the document the engine will ask for next
    size_t next_doc = 0;                                         // the next document a reader takes
    size_t depth = 2;                                            // documents read ahead at most (= readers + 1)
    size_t copies = 1;                                           // processes on this host that read ahead like this one
    bool stop = false;
    std::exception_ptr error;

    // every file once: names, record lengths, total (the bases are dropped); index of the first file without bases, or -1
    long measure() {
        const size_t N = paths.size();
        docs.assign(N, FastaDoc()); len.assign(N, 0);
        std::atomic<size_t> next{0};
        std::exception_ptr first_error;
        std::mutex emu;
        const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(reader_threads(), 8), N));
        std::vector<std::thread> th;
        for (size_t t = 0; t < T; t++) th.emplace_back([&]() {
            std::vector<uint8_t> scratch;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= N) break;
                try { docs[i] = read_fasta_replace(paths[i], scratch); len[i] = docs[i].total; }
                catch (...) { std::lock_guard<std::mutex> lk(emu); if (!first_error) first_error = std::current_exception(); }
            }
        });
        for (auto& t : th) t.join();
        if (first_error) std::rethrow_exception(first_error);
        for (size_t i = 0; i < N; i++) if (!len[i]) return (long)i;
        return -1;
    }
    void halt() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& w : workers) if (w.joinable()) w.join();
        workers.clear();
        stop = false; ready.clear();
    }
    // readers (a file is parsed at about 1 GB/s by one thread: 94 whole genomes one after the other would take minutes) take
    // the documents in order and keep at most `depth` of them ahead of the one the engine asks for
    void start(size_t from) {
        halt();
        expect = next_doc = from;
        size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(reader_threads(), 8), paths.size() - std::min(paths.size(), from)));
        // (what is read ahead stays within a quarter of this process's part of the host's memory: eight ranks x nine whole
        // genomes would be 216 GB)
        uint64_t longest = 1;
        for (uint64_t l : len) longest = std::max(longest, l);
        const uint64_t room = host_memory_available() / 4 / std::max<size_t>(copies, 1);
        const size_t max_docs = (size_t)std::max<uint64_t>(2, room / longest);
        depth = std::min(T + 1, max_docs);
        T = std::min(T, depth - 1);
        for (size_t t = 0; t < T; t++) workers.emplace_back([this]() {
            try {
                for (;;) {
                    size_t d;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || next_doc >= paths.size() || next_doc < expect + depth; });
                        if (stop || next_doc >= paths.size()) return;
                        d = next_doc++;
                    }
                    std::vector<uint8_t> bases;
                    (void)read_fasta_replace(paths[d], bases);
                    std::lock_guard<std::mutex> lk(mu);
                    if (stop) return;
                    ready.emplace(d, std::move(bases));
                    cv.notify_all();
                }
            } catch (...) { std::lock_guard<std::mutex> lk(mu); if (!error) error = std::current_exception(); cv.notify_all(); }
        });
    }
    // Engine::DocSupplier
    static int supply(void* user, uint64_t d, uint8_t* dst, uint64_t n) {
        StreamedInput& S = *static_cast<StreamedInput*>(user);
        try {
            if (S.workers.empty() || S.expect != d) S.start((size_t)d);      // (the text is built a second time: from the top)
            std::vector<uint8_t> bases;
            {
                std::unique_lock<std::mutex> lk(S.mu);
                S.cv.wait(lk, [&] { return S.ready.count((size_t)d) || S.error; });
                if (S.error) return 1;
                auto it = S.ready.find((size_t)d);
                bases = std::move(it->second);
                S.ready.erase(it);
                S.expect = (size_t)d + 1;
            }
            S.cv.notify_all();
            if (bases.size() != n) return 1;                                    // (the file changed between the two passes)
            std::memcpy(dst, bases.data(), n);
            return 0;
        } catch (...) { return 1; }
    }
    ~StreamedInput() { halt(); }
};

// mumemto_exec over a streamed collection: FASTA -> PREFIX.mums | .mems | .bumbl + PREFIX.lengths (+ -n / -M files), one text
static int run_streamed(BuildOptions& o, const std::vector<std::string>& inputs, bool mum_mode) {
    if (o.keep_temp || o.only_parse || o.arrays_out)
        throw CliError{"-K, -P and -A need the collection on the host: not available when the input is streamed (MUMEMTO_STREAM_INPUT)", 1};
    auto t0 = std::chrono::steady_clock::now();
    StreamedInput in;
    in.paths = inputs;
    std::unique_ptr<Engine> engine;
    std::exception_ptr engine_error;
    std::thread engine_init([&]() {
        if (std::getenv("MUMEMTO_DRY_RUN")) return;
        try { engine.reset(new Engine(std::getenv("MUMEMTO_DEVICE") ? std::atoi(std::getenv("MUMEMTO_DEVICE")) : 0, nullptr)); }
        catch (...) { engine_error = std::current_exception(); }
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } engine_joiner{engine_init};
    const long empty = in.measure();
    if (empty >= 0) {
        std::cerr << std::endl << "Empty input file found: " << inputs[(size_t)empty] << std::endl;
        throw CliError{"Please check the input files and ensure that it contains valid FASTA files. Cleaning up...", 1};
    }
    write_lengths_file(o.output_prefix, in.docs);
    uint64_t n_bases = 0, text_chars = 0;
    for (uint64_t l : in.len) { n_bases += l; text_chars += (o.use_rcomp ? 2 : 1) * (l + 1); }
    if (std::getenv("MUMEMTO_DRY_RUN")) {          // host-side checks only: every document through the supplier, in order
        engine_init.join();
        uint64_t h = 1469598103934665603ull;
        std::vector<uint8_t> dst;
        for (size_t d = 0; d < in.len.size(); d++) {
            dst.resize(in.len[d]);
            if (StreamedInput::supply(&in, d, dst.data(), in.len[d]) != 0) {
                if (in.error) std::rethrow_exception(in.error);
                throw CliError{"the streamed reader failed at " + inputs[d], 1};
            }
            for (uint8_t b : dst) { h ^= b; h *= 1099511628211ull; }
        }
        std::printf("docs=%zu bases=%zu fnv1a=%016llx num_distinct=%d max_doc_freq=%d max_total_freq=%d revcomp=%d "
                    "merge=%d anchor=%d binary=%d min_len=%zu\n", in.len.size(), (size_t)n_bases, (unsigned long long)h,
                    o.num_distinct_docs, o.rare_freq, o.max_mem_freq, (int)o.use_rcomp, (int)o.merge,
                    (int)o.anchor_merge, (int)o.binary, o.min_match_len);
        return 0;
    }
    std::fprintf(stderr, "\033[32m[build_main] \033[0mmeasured %zu files, %llu bases (the documents are read again, one at a time, "
                 "when the device asks) ... done.  (%.3f sec)\n", in.docs.size(), (unsigned long long)n_bases, secs_since(t0));
    in.start(0);                                          // the first documents are read while the engine comes up
    engine_init.join();
    if (engine_error) std::rethrow_exception(engine_error);
    Engine& eng = *engine;
    t0 = std::chrono::steady_clock::now();
    mmt_params p{};
    p.min_match_len = (uint32_t)o.min_match_len;
    p.num_distinct = (uint64_t)o.num_distinct_docs;
    p.max_doc_freq = o.rare_freq;
    p.max_total_freq = o.max_mem_freq;
    p.use_revcomp = o.use_rcomp ? 1 : 0;
    p.merge_metadata = o.merge ? 1 : 0;
    // (-M without -n cuts PREFIX.thresh / .thresh_rev out of the rows afterwards: such a run keeps them)
    if (!o.binary) eng.set_text_sink(o.output_prefix + (mum_mode ? ".mums" : ".mems"), false, o.merge && !o.anchor_merge);
    try { eng.run_supplied(&StreamedInput::supply, &in, in.len.data(), in.len.size(), p); }
    catch (...) {
        eng.set_text_sink(std::string());
        if (in.error) std::rethrow_exception(in.error);  // (what the reader met, not "the supplier failed")
        throw;
    }
    eng.set_text_sink(std::string());
    in.halt();
    const HostRows& R = eng.rows_meta();
    std::fprintf(stderr, "\033[32m[build_main] \033[0mfinding multi-%ss on the GPU ... done.  (%.3f sec)\n", mum_mode ? "MUM" : "MEM",
                 secs_since(t0));
    if (!mum_mode) eng.write_text_file(o.output_prefix + ".mems");      // (nothing left to do when the run streamed it)
    else if (o.binary) { const std::string& b = eng.bumbl(); write_file(o.output_prefix + ".bumbl", b.data(), b.size()); }
    else eng.write_text_file(o.output_prefix + ".mums");
    if (o.anchor_merge) {                           // mem_finder.hpp:110-115
        std::vector<uint16_t> th(eng.thresh_len());
        eng.copy_thresh(th.data());
        write_file(o.output_prefix + ".athresh", th.data(), (in.len[0] + 1) * sizeof(uint16_t));
    } else if (o.merge) {                           // mem_finder.hpp:116-157
        std::vector<uint16_t> fwd, rev;
        eng.thresh_files(fwd, rev);
        write_file(o.output_prefix + ".thresh", fwd.data(), fwd.size() * 2);
        write_file(o.output_prefix + ".thresh_rev", rev.data(), rev.size() * 2);
    }
    log_line("build_main", "Found " + std::to_string(R.n_rows) + " matches!");
    const float* ms = eng.stage_ms();
    std::fprintf(stderr, "GPU stages (ms): text %.2f | suffix sort %.2f (stream windows %.2f of it) | lcp+bwt %.2f | scan %.2f | verify %.2f | rows %.2f\n\n",
                 ms[0], ms[1], ms[6], ms[2], ms[3], ms[4], ms[5]);
    std::fflush(stdout); std::fflush(stderr);
    if (!std::getenv("MUMEMTO_FULL_TEARDOWN")) std::_Exit(0);
    return 0;
}

static int run_rank(BuildOptions& o) {
    const int rank = o.rank, world = o.gpus;
    const bool mum_mode = o.validate();
    if (o.from_parse_flag || o.arrays_in_flag || o.only_parse || o.keep_temp || o.arrays_out || o.binary || (o.merge && !o.anchor_merge))
        throw CliError{"--gpus runs the plain FASTA -> .mums / .mems job (with -n if wanted); -p -a -P -K -A -b -M are single-GPU options", 1};
    const std::vector<std::string> inputs = resolve_inputs(o);
    o.set_parameters(inputs.size(), mum_mode);
    const size_t N = inputs.size();
    const bool strict = mum_mode && (o.num_distinct_docs == 0 || (size_t)o.num_distinct_docs == N);
    if (strict && N - 1 < (size_t)world) throw CliError{"--gpus: fewer documents next to the anchor than GPUs", 1};
    if (rank == 0) for (const auto& n : o.notes) log_line("build_main", n);
    // this rank's documents: the anchor + a contiguous block of the others (strict), or everything
    std::vector<std::string> mine;
    if (strict) {
        const size_t rest = N - 1, base = rest / world, extra = rest % world;
        const size_t first = 1 + (size_t)rank * base + std::min<size_t>(rank, extra), count = base + ((size_t)rank < extra ? 1 : 0);
        mine.push_back(inputs[0]);
        mine.insert(mine.end(), inputs.begin() + first, inputs.begin() + first + count);
    } else mine = inputs;
    // The modes that shard the STREAM hold the whole collection on every rank: when that is more than the host can take
    // (eight copies of 94 whole genomes), every rank reads its documents one at a time as its engine asks (StreamedInput); and
    // each writes its rows window by window to its own piece of the output, PREFIX.rankR.mems, which the launcher joins --
    // nothing is gathered over the links (a rank of BASELINE configs[4] writes 66 GB).  MUMEMTO_RANK_PIECES=1 | 0 forces either.
    // (a strict share -- the anchor + this rank's block -- is streamed the same way when all ranks' shares together are too
    // much for the host, as long as it runs as ONE text: anchor partitions inside a rank read their documents again and
    // again and keep them resident)
    bool streamed = want_streamed_input(mine, (size_t)world);
    const bool pieces = !strict && (std::getenv("MUMEMTO_RANK_PIECES") ? std::atoi(std::getenv("MUMEMTO_RANK_PIECES")) != 0 : streamed);
    Engine eng(std::getenv("MUMEMTO_DEVICE") ? std::atoi(std::getenv("MUMEMTO_DEVICE")) : 0, nullptr);
    HostArena arena;
    HostDocs hd;
    std::vector<FastaDoc> docs;
    StreamedInput in;
    in.copies = (size_t)world;
    long empty = -1;
    if (streamed) {
        in.paths = mine; empty = in.measure(); docs = in.docs; hd.len = in.len;
        uint64_t chars = 0;
        for (uint64_t l : in.len) chars += (o.use_rcomp ? 2 : 1) * (l + 1);
        if (strict && empty < 0 && (chars > eng.auto_max_text() || std::getenv("MUMEMTO_MAX_TEXT") || std::getenv("MMT_MAX_TEXT"))) {
            streamed = false; docs.clear(); hd.len.clear();
        }
    }
    if (!streamed) empty = read_fasta_collection(mine, docs, arena, hd);
    if (empty >= 0) throw CliError{"Empty input file found: " + mine[(size_t)empty], 1};
    // PREFIX.lengths in pieces: every rank describes the documents only it has read, rank 0 joins them after the exchange
    if (strict) {
        std::vector<FastaDoc> part(docs.begin() + (rank ? 1 : 0), docs.end());
        write_lengths_file(o.output_prefix + ".rank" + std::to_string(rank), part);
    } else if (rank == 0) write_lengths_file(o.output_prefix, docs);
    mmt_params p{};
    p.min_match_len = (uint32_t)o.min_match_len;
    p.num_distinct = (uint64_t)o.num_distinct_docs;
    p.max_doc_freq = o.rare_freq;
    p.max_total_freq = o.max_mem_freq;
    p.use_revcomp = o.use_rcomp ? 1 : 0;
    p.merge_metadata = strict ? 1 : 0;
    Comm* comm = nullptr;
    if (!pieces) {
        uint8_t id[128];
        if (rank == 0) { comm_unique_id(id); leave_id(o.comm_file, id); } else fetch_id(o.comm_file, id);
        comm = comm_create(eng, rank, world, id);
    }
    if (strict) { p.num_distinct = 0; p.max_total_freq = 0; }
    else {
        // every rank builds the tables of the parse and produces, scans and drops its own share of the stream (ranges of
        // the emitter's output, or -- MUMEMTO_PRODUCER=guided, automatic when the dictionary would not fit -- whole bins
        // of leading characters); only the rows travel
        eng.set_scan_shard((uint32_t)rank, (uint32_t)world);
    }
    const std::string piece = o.output_prefix + ".rank" + std::to_string(rank) + (mum_mode ? ".mums" : ".mems");
    if (pieces) eng.set_text_sink(piece, true);
    if (streamed) {
        in.start(0);
        try { eng.run_supplied(&StreamedInput::supply, &in, in.len.data(), in.len.size(), p); }
        catch (...) { if (in.error) std::rethrow_exception(in.error); throw; }
        in.halt();
    } else eng.run_partitioned_docs(hd.ptr.data(), hd.len.data(), hd.len.size(), p, 0);
    size_t rows = 0;
    if (strict) {
        bool root = false;
        MergedRows merged = dist_merge(*comm, (uint32_t)o.min_match_len, &root);
        if (root) {
            write_merged_text(eng, merged, o.output_prefix + ".mums");      // (in pieces: 94 whole genomes are 40 GB of rows)
            rows = merged.n_rows;
            if (o.anchor_merge) {
                download_merged(eng, merged);
                write_file(o.output_prefix + ".athresh", merged.thresh.data(), merged.thresh.size() * sizeof(uint16_t));
            }
            std::ofstream all(o.output_prefix + ".lengths", std::ios::binary);
            for (int r = 0; r < world; r++) {
                const std::string piece = o.output_prefix + ".rank" + std::to_string(r) + ".lengths";
                { std::ifstream in(piece, std::ios::binary); all << in.rdbuf(); }
                std::remove(piece.c_str());
            }
        }
    } else if (pieces) {
        eng.set_text_sink(std::string());
        eng.write_text_file(piece);                  // (nothing left to do when the run streamed it)
    } else {
        (void)eng.rows(Engine::ROWS_TEXT);
        const std::string text = dist_gather_text(*comm);
        if (rank == 0) write_file(o.output_prefix + (mum_mode ? ".mums" : ".mems"), text.data(), text.size());
    }
    if (comm) comm_destroy(comm);
    if (rank == 0) log_line("build_main", strict ? "Found " + std::to_string(rows) + " matches on " + std::to_string(world) + " GPUs!"
                                                  : "matches of " + std::to_string(world) + " GPUs written");
    std::fflush(stdout); std::fflush(stderr);
    std::_Exit(0);
}

int main(int argc, char** argv) {
    mark("main");
    std::fprintf(stderr, "\nmumemto_exec (MI355X / gfx950 build of the mumemto 1.4.0 hot path)\n");
    if (argc == 1) { std::fprintf(stderr, "Usage: mumemto_exec [options] [input_fasta [...]]\n\t-h, --help  prints detailed usage message\n"); return 0; }
    BuildOptions o;
    try {
        o.parse(argc, argv);
        if (o.help) { std::fputs(usage_text().c_str(), stderr); return 0; }
        if (o.gpus < 1) throw CliError{"--gpus needs a positive number", 1};
        if (o.rank >= 0) return run_rank(o);
        if (o.gpus > 1 || std::getenv("MUMEMTO_FORCE_RANKS")) { (void)o.validate(); return launch_ranks(argc, argv, o); }
        const bool mum_mode = o.validate();
        const bool checkpoint = o.from_parse_flag || o.arrays_in_flag;
        std::vector<uint64_t> doc_len;
        if (checkpoint) doc_len = read_lengths_file(o.from_parse_flag ? o.parse_prefix : o.arrays_in);
        const std::vector<std::string> inputs = checkpoint ? std::vector<std::string>() : resolve_inputs(o);
        o.set_parameters(checkpoint ? doc_len.size() : inputs.size(), mum_mode);
        for (const auto& n : o.notes) log_line("build_main", n);

        if (!checkpoint && want_streamed_input(inputs, 1)) return run_streamed(o, inputs, mum_mode);
        auto t0 = std::chrono::steady_clock::now();
        // the HIP runtime comes up (device, stream, code objects) while the host threads read the inputs
        const bool dry_run = std::getenv("MUMEMTO_DRY_RUN") != nullptr;
        std::unique_ptr<Engine> engine;
        std::exception_ptr engine_error;
        std::thread engine_init;
        std::promise<void> engine_up;
        std::shared_future<void> engine_ready = engine_up.get_future().share();
        // (a collection that will run as one suffix array through the parse proper peaks at ~10.5 bytes of device heap per text
        // character -- 123 GB mapped for the 12.03 G characters of the C3 stand-in --: the heap is mapped to that size while
        // the inputs are on their way, before any kernel runs.  Chunks mapped beside running work each waited ~30 ms for it:
        // 0.04 - 0.10 s of a 1.9 s process, and never the same twice.  MUMEMTO_NO_PREMAP switches it off.)
        size_t premap_bytes = 0, slots_bound = 0;
        if (!dry_run && !checkpoint && !std::getenv("MUMEMTO_NO_PREMAP")) {
            double file_bytes = 0;
            for (const auto& f : inputs) { std::error_code ec; const auto sz = std::filesystem::file_size(f, ec); if (!ec) file_bytes += (double)sz; }
            if (3.0 * 2.0 * file_bytes + 24.0 * 1073741824.0 <= 200.0 * 1073741824.0) {
                // (9.9, 10.5 until round 6: the measured peak is 9.69 B per character -- 116.5 GB on the stand-in -- and what is mapped
                // beyond the peak is memory the next process waits for while the driver scrubs it)
                premap_bytes = (size_t)(9.9 * (o.use_rcomp ? 2.0 : 1.0) * file_bytes) + ((size_t)1 << 30);
                slots_bound = (size_t)file_bytes + inputs.size() * 8192 + 4096;     // (the documents' slots: the readers must not wait for the mapping)
            }
        }
        if (!dry_run)
            engine_init = std::thread([&]() {
                try { engine.reset(new Engine(std::getenv("MUMEMTO_DEVICE") ? std::atoi(std::getenv("MUMEMTO_DEVICE")) : 0, nullptr)); }
                catch (...) { engine_error = std::current_exception(); }
                try { if (engine && slots_bound && slots_bound * 2 <= engine->auto_max_text()) (void)engine->begin_input_slots(slots_bound); } catch (...) {}
                engine_up.set_value();
                if (engine && premap_bytes) pool::premap(engine->device(), premap_bytes);
            });
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } engine_joiner{engine_init};
        HostArena arena;
        HostDocs hd;
        std::vector<FastaDoc> docs;
        // A collection that will run as one suffix array (judged by the file sizes, which bound the bases) goes to the device
        // document by document while the other files are still being read, as in mmt_engine_run_files: ONE copier thread
        // takes the documents in the order the readers finish them -- once the engine is up, which happens beside the
        // reading.  MUMEMTO_NO_UPLOAD_OVERLAP switches it off.
        struct Upload {
            std::mutex mu; std::condition_variable cv; std::deque<std::pair<size_t, uint64_t>> q; bool done = false;
            std::thread thread; std::exception_ptr error; std::vector<size_t> slot; bool on = false, skipped = false;
        } up;
        ReadHooks hooks;
        hooks.layout = [&](const uint8_t* arena_p, size_t bytes, const std::vector<size_t>& slot, bool all_in_arena) {
            if (dry_run || !all_in_arena || std::getenv("MUMEMTO_NO_UPLOAD_OVERLAP") || slot.size() < 2) return;
            up.slot = slot; up.on = true;
            up.thread = std::thread([&up, &engine, &engine_error, engine_ready, arena_p, bytes]() {
                try {
                    engine_ready.wait();
                    const uint64_t bound = 2 * ((uint64_t)bytes + up.slot.size());            // text characters at most
                    if (engine_error || !engine || bound > engine->auto_max_text()) { up.skipped = true; return; }
                    uint8_t* dev = engine->begin_input_slots(bytes);
                    MMT_HIP(hipSetDevice(engine->device()));
                    hipStream_t cs = nullptr;
                    MMT_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
                    for (;;) {
                        std::pair<size_t, uint64_t> job;
                        {
                            std::unique_lock<std::mutex> lk(up.mu);
                            up.cv.wait(lk, [&] { return !up.q.empty() || up.done; });
                            if (up.q.empty()) break;
                            job = up.q.front(); up.q.pop_front();
                        }
                        if (job.second)
                            MMT_HIP(hipMemcpyAsync(dev + up.slot[job.first], arena_p + up.slot[job.first], job.second,
                                                   hipMemcpyHostToDevice, cs));
                        MMT_HIP(hipStreamSynchronize(cs));
                    }
                    (void)hipStreamDestroy(cs);
                } catch (...) { up.error = std::current_exception(); }
            });
        };
        hooks.ready = [&](size_t i, uint64_t len) {
            if (!up.on) return;
            { std::lock_guard<std::mutex> lk(up.mu); up.q.emplace_back(i, len); }
            up.cv.notify_one();
        };
        auto finish_upload = [&]() {
            if (!up.thread.joinable()) return;
            { std::lock_guard<std::mutex> lk(up.mu); up.done = true; }
            up.cv.notify_one();
            up.thread.join();
        };
        struct UploadJoiner { std::function<void()> f; ~UploadJoiner() { f(); } } upload_joiner{finish_upload};
        // Chunked input (fasta.hpp): a one-shot run never holds the collection in host memory -- every reader thread sends
        // the bases it parses to the document's slot on the device through two page-locked 8 MB buffers of its own.  What
        // 6 GB of anonymous host memory cost this process: ~0.2 s of page faults while it was filled and 0.3 - 0.4 s between
        // _Exit and the parent's waitpid (tests/micro/exit_probe2.cpp; giving it back beside the run stalls the GPU's queues
        // for as long).  A run that turns out to need host copies (anchor partitions) reads the files again.
        struct ChunkUp {
            std::vector<size_t> slot; size_t bytes = 0; bool on = false;
            std::atomic<bool> skipped{false};
            std::once_flag once; uint8_t* dev = nullptr; int device = 0;
        } cu;
        // (MUMEMTO_INPUT_CHUNK_MB: tuning aid; page-locked memory costs 0.19 s per GB when it is made and 0.13 s per GB at the exit)
        const size_t CHUNK = (size_t)(std::getenv("MUMEMTO_INPUT_CHUNK_MB") ? std::max(1, std::atoi(std::getenv("MUMEMTO_INPUT_CHUNK_MB"))) : 8) << 20;
        // (MUMEMTO_DRY_RUN_CHUNK=<bytes>, with MUMEMTO_DRY_RUN: the chunked reader into host vectors, chunks of that many bytes
        // -- the host-side test of the chunk logic, tests/test_cli_host.py)
        const size_t dry_chunk = dry_run && std::getenv("MUMEMTO_DRY_RUN_CHUNK") ? (size_t)std::max(1, std::atoi(std::getenv("MUMEMTO_DRY_RUN_CHUNK"))) : 0;
        std::vector<std::vector<uint8_t>> dry_docs;
        hooks.plan_chunks = [&](size_t bytes, const std::vector<size_t>& slot, bool all_plain) {
            if (dry_chunk && all_plain) { dry_docs.assign(slot.size() - 1, std::vector<uint8_t>()); cu.on = true; return true; }
            if (dry_run || !all_plain || slot.size() < 2 || std::getenv("MUMEMTO_NO_UPLOAD_OVERLAP") ||
                std::getenv("MUMEMTO_NO_CHUNKED_INPUT") || std::getenv("MUMEMTO_KEEP_HOST_INPUT")) return false;
            // (only collections that will clearly run as one suffix array on a device of this class: Engine::auto_max_text's
            // formula with 200 GB in place of the free memory nobody has asked the driver for yet)
            if (3.0 * 2.0 * (double)(bytes + slot.size()) + 24.0 * 1073741824.0 > 200.0 * 1073741824.0) return false;
            cu.slot = slot; cu.bytes = bytes; cu.on = true;
            return true;
        };
        hooks.chunks = [&](size_t i) {
            ChunkTarget t;
            if (dry_chunk) {
                t.chunk = dry_chunk;
                t.swap = [&, i](uint8_t* filled, size_t n, uint64_t offset, bool more) -> uint8_t* {
                    static thread_local std::vector<uint8_t> two[2];
                    for (auto& v : two) v.resize(dry_chunk);
                    if (filled && n) {
                        if (dry_docs[i].size() < offset + n) dry_docs[i].resize(offset + n);
                        std::memcpy(dry_docs[i].data() + offset, filled, n);
                    }
                    if (!more) return nullptr;
                    return filled == two[0].data() ? two[1].data() : two[0].data();
                };
                return t;
            }
            t.chunk = CHUNK;
            t.swap = [&, i](uint8_t* filled, size_t n, uint64_t offset, bool more) -> uint8_t* {
                if (cu.skipped.load()) return nullptr;
                std::call_once(cu.once, [&]() {
                    engine_ready.wait();
                    const uint64_t bound = 2 * ((uint64_t)cu.bytes + cu.slot.size());
                    if (engine_error || !engine || bound > engine->auto_max_text()) { cu.skipped.store(true); return; }
                    cu.device = engine->device();
                    cu.dev = engine->begin_input_slots(cu.bytes);
                });
                if (cu.skipped.load()) return nullptr;
                // (a reader thread gives its page-locked buffers back when it ends, i.e. when the inputs are read: what is
                // still page-locked at the exit costs 0.13 s per GB there)
                struct Bufs { uint8_t* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false};
                              hipStream_t s = nullptr;
                              ~Bufs() { for (int k = 0; k < 2; k++) { if (buf[k]) (void)hipHostFree(buf[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); }
                                        if (s) (void)hipStreamDestroy(s); } };
                static thread_local Bufs tb;
                if (!tb.s) {
                    MMT_HIP(hipSetDevice(cu.device));
                    MMT_HIP(hipStreamCreateWithFlags(&tb.s, hipStreamNonBlocking));
                    for (int k = 0; k < 2; k++) {
                        MMT_HIP(hipHostMalloc(reinterpret_cast<void**>(&tb.buf[k]), CHUNK, hipHostMallocDefault));
                        MMT_HIP(hipEventCreateWithFlags(&tb.ev[k], hipEventDisableTiming));
                    }
                }
                const int k = filled == tb.buf[1] ? 1 : 0;
                if (filled && n) {
                    MMT_HIP(hipMemcpyAsync(cu.dev + cu.slot[i] + offset, filled, n, hipMemcpyHostToDevice, tb.s));
                    MMT_HIP(hipEventRecord(tb.ev[k], tb.s));
                    tb.busy[k] = true;
                }
                if (!more) {                        // the document is complete: on the device when this returns
                    MMT_HIP(hipStreamSynchronize(tb.s));
                    tb.busy[0] = tb.busy[1] = false;
                    return nullptr;
                }
                const int nk = filled ? 1 - k : 0;
                if (tb.busy[nk]) { MMT_HIP(hipEventSynchronize(tb.ev[nk])); tb.busy[nk] = false; }
                return tb.buf[nk];
            };
            return t;
        };
        if (!checkpoint) {
            long empty = read_fasta_collection(inputs, docs, arena, hd, &hooks);
            if (empty == -2) {                      // the device turned out too small for one suffix array: host copies after all
                cu.on = false;
                hooks.plan_chunks = nullptr; hooks.chunks = nullptr;
                empty = read_fasta_collection(inputs, docs, arena, hd, &hooks);
            }
            if (empty >= 0) {                       // ref_builder.cpp:249-252 + pfp_mum.cpp:68-71
                std::cerr << std::endl << "Empty input file found: " << inputs[(size_t)empty] << std::endl;
                throw CliError{"Please check the input files and ensure that it contains valid FASTA files. Cleaning up...", 1};
            }
            doc_len = hd.len;
            if (dry_chunk && cu.on)
                for (size_t d = 0; d < hd.ptr.size(); d++) { dry_docs[d].resize(hd.len[d]); hd.ptr[d] = dry_docs[d].data(); }
        }
        uint64_t n_bases = 0;
        for (uint64_t l : hd.len) n_bases += l;
        uint64_t text_chars = 0;
        for (uint64_t l : doc_len) text_chars += (o.use_rcomp ? 2 : 1) * (l + 1);
        // stage checkpoints: the text (from PREFIX.parse/.dict) or the stream (PREFIX.sa/.lcp/.bwt) comes from files
        std::vector<uint8_t> ck_text, ck_bwt, ck_sa_hi;
        std::vector<uint32_t> ck_sa, ck_lcp;
        if (o.from_parse_flag) {
            ck_text = text_from_parse(o.parse_prefix, o.pfp_w);
            if (ck_text.size() != text_chars)
                throw CliError{"the parse expands to " + std::to_string(ck_text.size()) + " characters, " + o.parse_prefix +
                               ".lengths announces " + std::to_string(text_chars) + " (is -r set as it was for the parse?)", 1};
            log_line("build_main", "text of " + std::to_string(text_chars) + " characters rebuilt from " + o.parse_prefix +
                                       ".parse / .dict");
        } else if (o.arrays_in_flag) {
            load_stream_files(o.arrays_in, text_chars, ck_sa, ck_sa_hi, ck_lcp, ck_bwt);
            log_line("build_main", "Using pre-computed LCP/BWT/SA arrays from files with prefix: " + o.arrays_in);
        } else {
            write_lengths_file(o.output_prefix, docs);
            std::fprintf(stderr, "\033[32m[build_main] \033[0mread %zu files, %llu bases ... done.  (%.3f sec)\n", docs.size(),
                         (unsigned long long)n_bases, secs_since(t0));
        }

        if (dry_run) {                            // host-side checks only (tests on machines without a GPU)
            uint64_t h = 1469598103934665603ull;
            for (size_t d = 0; d < hd.ptr.size(); d++)
                for (uint64_t i = 0; i < hd.len[d]; i++) { h ^= hd.ptr[d][i]; h *= 1099511628211ull; }
            for (uint8_t b : ck_text) { h ^= b; h *= 1099511628211ull; }
            for (uint32_t v : ck_sa) { h ^= v; h *= 1099511628211ull; }
            for (uint32_t v : ck_lcp) { h ^= v; h *= 1099511628211ull; }
            for (uint8_t b : ck_bwt) { h ^= b; h *= 1099511628211ull; }
            if (checkpoint)
                std::printf("checkpoint=%s text_chars=%llu entries=%zu ", o.from_parse_flag ? "parse" : "arrays",
                            (unsigned long long)text_chars, o.from_parse_flag ? ck_text.size() : ck_sa.size());
            std::printf("docs=%zu bases=%zu fnv1a=%016llx num_distinct=%d max_doc_freq=%d max_total_freq=%d revcomp=%d "
                        "merge=%d anchor=%d binary=%d min_len=%zu\n", doc_len.size(), (size_t)n_bases, (unsigned long long)h,
                        o.num_distinct_docs, o.rare_freq, o.max_mem_freq, (int)o.use_rcomp, (int)o.merge,
                        (int)o.anchor_merge, (int)o.binary, o.min_match_len);
            return 0;
        }
        mark("inputs read");
        bool host_released = false;
        t0 = std::chrono::steady_clock::now();
        engine_init.join();
        if (engine_error) std::rethrow_exception(engine_error);
        Engine& eng = *engine;
        // (measured, tests/micro/exit_ab.py + exit_probe.cpp: what a process pays between _Exit and its parent's waitpid is not
        // its mapped device memory -- 123 or 79 GB mapped at exit: the same 0.45 s; a bare HIP process 0.08 s; 100 GB of hipMalloc
        // 0.06 s -- but page-locked HOST memory, 0.13 s per GB: the sink's blocks are 64 MB now.  Handing the heap's free top back
        // early therefore buys nothing and costs the run ~0.08 s: opt-in)
        eng.set_one_shot(std::getenv("MUMEMTO_EARLY_UNMAP") != nullptr);
        // one suffix array while the text fits the device (40-bit positions beyond 2^32 characters); beyond that --
        // or beyond MUMEMTO_MAX_TEXT characters -- strict multi-MUMs run as anchor partitions + merge on this GPU
        // (the same estimate as the library: Engine::auto_max_text, MUMEMTO_MAX_TEXT overrides both)
        const bool explicit_limit = std::getenv("MUMEMTO_MAX_TEXT") || std::getenv("MMT_MAX_TEXT");
        const uint64_t max_text = eng.auto_max_text();
        const bool strict_mode = mum_mode && (o.num_distinct_docs == 0 || (size_t)o.num_distinct_docs == doc_len.size());
        // modes without a partition merge are tried as one run whatever the estimate says, like the library does
        bool partitioned = text_chars > max_text && doc_len.size() >= 3 && (strict_mode || explicit_limit);
        if (checkpoint && partitioned) throw CliError{"-p / -a are not available for inputs larger than one suffix array", 1};
        if (checkpoint && (o.keep_temp || o.arrays_out))
            throw CliError{"-K and -A write what -p / -a read: run them without a checkpoint", 1};
        mark("engine created");
        if (o.from_parse_flag) eng.set_text_host(ck_text.data(), ck_text.size(), doc_len.data(), doc_len.size(), o.use_rcomp);
        else if (o.arrays_in_flag)
            eng.set_stream_host40(ck_sa.data(), ck_sa_hi.empty() ? nullptr : ck_sa_hi.data(), ck_lcp.data(), ck_bwt.data(),
                                  ck_sa.size(), doc_len.data(), doc_len.size(), o.use_rcomp);
        else {
            finish_upload();
            if (up.error) std::rethrow_exception(up.error);
            const bool chunked = cu.on && !cu.skipped.load();
            const bool uploaded = chunked || (up.on && !up.skipped);
            if (chunked && partitioned) {            // (cannot happen: the chunk target checked the same estimate)
                if (read_fasta_collection(inputs, docs, arena, hd) >= 0) throw CliError{"an input file changed while the run was going on", 1};
            } else if (chunked) {
                eng.finish_input_slots(cu.slot, doc_len.data(), doc_len.size());
                host_released = true;               // there never were host copies: the repeat as anchor partitions reads the files again
            } else if (!partitioned && uploaded) eng.finish_input_slots(up.slot, doc_len.data(), doc_len.size());
            else if (!partitioned) eng.set_input_host_docs(hd.ptr.data(), doc_len.data(), doc_len.size());
        }
        mark("input on the device");
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
            eng.run_partitioned_docs(hd.ptr.data(), doc_len.data(), doc_len.size(), p, explicit_limit ? max_text : 0);
            log_line("build_main", "text of " + std::to_string(text_chars) + " characters processed as " +
                                       std::to_string(eng.partitions_used()) + " anchor partitions");
        } else {
            if (o.arrays_out) eng.set_keep_columns(1);          // -A dumps whole columns: keep them next to the windows
            // PREFIX.mums is written window by window while the run goes on (Engine::set_text_sink)
            if (!o.binary) eng.set_text_sink(o.output_prefix + (mum_mode ? ".mums" : ".mems"), false, o.merge && !o.anchor_merge);
            try {
                eng.run(p);
                eng.set_text_sink(std::string());
            } catch (const DeviceOom&) {
                // The estimate accepted the collection and its dictionary or its giant phrases still ran out of memory: a
                // strict multi-MUM run is repeated as anchor partitions from the host copies, as the library does
                // (Engine::run_partitioned_docs); the sink's PREFIX.mums.tmp is already gone (Engine::sink_close).
                eng.set_text_sink(std::string());
                const bool can = strict_mode && !checkpoint && doc_len.size() >= 3 && !o.keep_temp && !o.arrays_out &&
                                 !(o.merge && !o.anchor_merge);
                if (!can) throw;
                log_line("build_main", "one suffix array ran out of device memory: repeating the run as anchor partitions");
                eng.forget_last_run();
                if (host_released) {                             // the host copies went away beside the run: once more from the files
                    if (read_fasta_collection(inputs, docs, arena, hd) >= 0) throw CliError{"an input file changed while the run was going on", 1};
                    host_released = false;
                }
                eng.run_partitioned_docs(hd.ptr.data(), doc_len.data(), doc_len.size(), p, text_chars / 2);
                partitioned = true;
                log_line("build_main", "text of " + std::to_string(text_chars) + " characters processed as " +
                                           std::to_string(eng.partitions_used()) + " anchor partitions");
            }
        }
        mark("run done");
        const HostRows& R = eng.rows_meta();       // (write_text_file brings the bytes; .bumbl pulls the arrays itself)
        std::fprintf(stderr, "\033[32m[build_main] \033[0mfinding multi-%ss on the GPU ... done.  (%.3f sec)\n",
                     mum_mode ? "MUM" : "MEM", secs_since(t0));

        if (!mum_mode) eng.write_text_file(o.output_prefix + ".mems");      // (nothing left to do when the run streamed it)
        else if (o.binary) { const std::string& b = eng.bumbl(); write_file(o.output_prefix + ".bumbl", b.data(), b.size()); }
        else eng.write_text_file(o.output_prefix + ".mums");       // (nothing left to do when the run streamed it)

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
            std::vector<uint64_t> sa(n);
            std::vector<uint32_t> lcp(n);
            std::vector<uint8_t> bwt(n), text(n);
            eng.copy_sa64(sa.data()); eng.copy_lcp(lcp.data()); eng.copy_bwt(bwt.data()); eng.copy_text(text.data());
            std::vector<uint8_t> fsa, flcp, fbwt;
            put40(fsa, n); put40(flcp, 0); fbwt.push_back(n ? text[n - 1] : 0);
            for (uint64_t j = 0; j < n; j++) { put40(fsa, sa[j]); put40(flcp, lcp[j]); fbwt.push_back(bwt[j]); }
            write_file(o.output_prefix + ".sa", fsa.data(), fsa.size());
            write_file(o.output_prefix + ".lcp", flcp.data(), flcp.size());
            write_file(o.output_prefix + ".bwt", fbwt.data(), fbwt.size());
        }
        mark("outputs written");
        const pool::Stats heap = pool::stats(eng.device());
        if (std::getenv("MUMEMTO_TIMING"))
            std::fprintf(stderr, "[timing] device heap: %.2f GB live at the peak, %.2f GB mapped, %.3f s mapping it\n",
                         heap.peak / 1073741824.0, heap.mapped / 1073741824.0, heap.map_seconds);
        if (const char* sp = std::getenv("MUMEMTO_STATS")) {       // one JSON object for bench.py
            const float* sm = eng.stage_ms();
            const float* pm = eng.pfp_state().ms;
            std::ofstream js(sp);
            js << "{\"text_chars\": " << text_chars << ", \"wide\": " << (eng.wide() ? "true" : "false")
               << ", \"partitions\": " << eng.partitions_used() << ", \"scan_ranges\": " << eng.scan_ranges()
               << ", \"rows\": " << R.n_rows << ", \"candidates\": " << eng.n_candidates()
               << ", \"producer\": " << eng.producer_used() << ", \"stage_ms\": [";
            for (int i = 0; i < 8; i++) js << (i ? ", " : "") << sm[i];
            js << "], \"pfp_ms\": [";
            for (int i = 0; i < 8; i++) js << (i ? ", " : "") << pm[i];
            js << "], \"heap_peak_bytes\": " << heap.peak << ", \"heap_mapped_bytes\": " << heap.mapped
               << ", \"heap_map_seconds\": " << heap.map_seconds << ", \"seconds_since_start\": " << secs_since(g_start)
               << "}\n";
        }
        log_line("build_main", "Found " + std::to_string(R.n_rows) + " matches!");
        if (o.keep_temp) write_pfp_files();         // -K: keep PREFIX.dict / PREFIX.parse
        const float* ms = eng.stage_ms();
        std::fprintf(stderr, "GPU stages (ms): text %.2f | suffix sort %.2f (stream windows %.2f of it) | lcp+bwt %.2f | scan %.2f | verify %.2f | rows %.2f\n\n",
                     ms[0], ms[1], ms[6], ms[2], ms[3], ms[4], ms[5]);
        // Everything is on disk: leave without unloading the HIP runtime and freeing gigabytes of HBM buffer by buffer
        // (the driver reclaims them with the process).  MUMEMTO_FULL_TEARDOWN=1 keeps the orderly exit, which
        // profilers that flush at exit need.
        if (!std::getenv("MUMEMTO_FULL_TEARDOWN")) {
            std::fflush(stdout); std::fflush(stderr);
            std::_Exit(0);
        }
        return 0;
    } catch (const CliError& e) {
        std::fprintf(stderr, "\n\033[31mError: \033[m%s\n\n", e.message.c_str());
        return e.code;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "\n\033[31mError: \033[m%s\n\n", e.what());
        return 1;
    }
}
