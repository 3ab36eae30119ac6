// partitioned.cpp -- collections whose text does not fit the device as one suffix array.
//
// The reference scales the same way (README.md:124-141): split the documents into
// partitions that share document 0 (the anchor), run each partition with merge metadata
// (-M -n), fold the partitions (anchor_merge, src/merge_candidates.cpp).  Here the
// partitions run back to back on one GPU, the fold and the re-sort into direct-run order
//  run on the same GPU, and the result is byte-identical to a direct run.
// Like the reference's merge (include/pfp_mum.hpp:178-183) this is defined for strict
// multi-MUMs only.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <thread>

#include "engine.hpp"
#include "merge.hpp"

namespace mmt {

// What fits the device as one suffix array: 14 (94 x 64 Mbp at 0.1 % divergence) to 21 (36 x 60 Mbp at 0.2 %) bytes of
// device memory per character at the peak: text 1, suffix array 5, BWT 1, then either the emitter tables -- which grow
// with the dictionary, i.e. with the divergence -- or PLCP 4 + the scan range.  A single-array run that
// still runs out of memory is repeated as partitions when the mode allows it.
uint64_t Engine::auto_max_text() const {
    // The stream is never stored (windows of it are produced, scanned and dropped), so a run needs the text (1 byte per
    // character), the raw bases until the text exists (0.5), the cut bits and the per-phrase tables of the parse (about
    // 1.5 at their peak) and a fixed amount for one batch / window of the producer; when the tables of the dictionary
    // would not fit next to that, the run takes the bucket-wise producer by itself (pfp.cpp).
    constexpr double PEAK_BYTES_PER_CHAR = 3.0, FIXED = 24.0 * 1073741824.0;
    const double avail = 0.95 * (double)pool::available(device_) - FIXED;
    uint64_t max_text = avail > 0 ? (uint64_t)std::min<double>(avail / PEAK_BYTES_PER_CHAR, (double)((1ull << 40) - 1)) : 0;
    // (the fixed part is what a batch of a whole-genome run takes; a device another process has filled still runs small
    // collections: 1 GB + 30 bytes per character covers every table of the parse proper)
    const double small = (0.95 * (double)pool::available(device_) - 1073741824.0) / 30.0;
    if (small > 0 && (uint64_t)small > max_text) max_text = std::min<uint64_t>((uint64_t)small, 1ull << 30);
    // one variable for the library and the command line (MMT_MAX_TEXT: the older name)
    for (const char* name : {"MUMEMTO_MAX_TEXT", "MMT_MAX_TEXT"})
        if (const char* c = std::getenv(name)) { max_text = std::strtoull(c, nullptr, 10); break; }
    return max_text;
}

const std::vector<uint16_t>& Engine::merged_thresh() {
    if (merged_thresh_valid_ && !merged_.on_host) download_merged(*this, merged_);
    return merged_.thresh;
}

void Engine::forget_last_run() {
    release_columns();
    // ... and the results (thresholds and merged tables of a genome-sized anchor are tens of GB), and the input buffer
    rows_ = HostRows(); rows_pending_ = 0; merged_thresh_valid_ = false; thresh_len_ = 0; thresh16_valid_ = false;
    merged_ = MergedRows();
    d_thresh_.release(); d_thresh16_.release(); d_rows_.release(); d_otext_.release(); d_olen_.release(); d_ooffs_.release(); d_ost_.release();
    d_omdoc_.release(); d_bases_own_.release(); d_bases_ = nullptr; input_valid_ = false;
}

void Engine::run_once_dropping_input(const mmt_params& p) {
    partitions_used_ = 1;
    drop_input_after_text_ = true;
    try { run(p); } catch (...) { drop_input_after_text_ = false; throw; }
    drop_input_after_text_ = false;
}

void Engine::run_supplied(DocSupplier fn, void* user, const uint64_t* doc_len, size_t n_docs, const mmt_params& p) {
    MMT_HIP(hipSetDevice(device_));
    forget_last_run();
    set_input_supplier(fn, user, doc_len, n_docs);
    try { run_once_dropping_input(p); } catch (...) { supplier_ = nullptr; supplier_user_ = nullptr; throw; }
    supplier_ = nullptr; supplier_user_ = nullptr;
}

void Engine::run_partitioned_host(const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs, const mmt_params& p,
                                  uint64_t max_text) {
    std::vector<const uint8_t*> ptr(n_docs);
    uint64_t at = 0;
    for (size_t d = 0; d < n_docs; d++) { ptr[d] = h_bases + at; at += doc_len[d]; }
    run_partitioned_docs(ptr.data(), doc_len, n_docs, p, max_text);
}

void Engine::run_partitioned_docs(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs, const mmt_params& p,
                                  uint64_t max_text) {
    MMT_HIP(hipSetDevice(device_));
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t mult = p.use_revcomp ? 2 : 1;
    std::vector<uint64_t> base(n_docs + 1, 0);
    uint64_t total = 0;
    for (size_t d = 0; d < n_docs; d++) { base[d + 1] = base[d] + doc_len[d]; total += mult * (doc_len[d] + 1); }
    const bool strict = p.max_doc_freq == 1 && (p.num_distinct == 0 || p.num_distinct == n_docs) &&
                        (p.max_total_freq == 0 || (uint64_t)p.max_total_freq >= n_docs);
    forget_last_run();                       // what the previous run left behind counts as free memory below
    const bool auto_limit = max_text == 0;
    if (auto_limit) max_text = auto_max_text();
    partitions_used_ = 1;
    // (a non-strict mode has no partition merge: with the automatic limit it is tried as one suffix array anyway and
    // fails with "out of device memory" if that was too optimistic; an explicit limit is an order)
    if (total <= max_text || n_docs < 3 || (!strict && auto_limit)) {
        try {
            // (a text that will be packed -- two bits per character -- never has its raw bases on the device as a whole:
            // they go through a staging buffer document by document; MMT_INPUT_DEFERRED=1 forces that route for tests)
            const bool defer = std::getenv("MMT_INPUT_DEFERRED") ? std::atoi(std::getenv("MMT_INPUT_DEFERRED")) != 0
                                                                  : want_packed_text_of(total, true);
            if (defer) set_input_host_docs_deferred(doc_ptr, doc_len, n_docs);
            else set_input_host_docs(doc_ptr, doc_len, n_docs);
            run_once_dropping_input(p);
            host_docs_.clear();
            return;
        } catch (const DeviceOom&) {
            if (!strict || n_docs < 3 || !auto_limit) throw;
            release_columns();
            d_bases_own_.release();
            max_text = total / 2;            // partitions of at most half the text, and so on below
        }
    }
    // (the partitions' own rows are not the output: nothing is streamed to the output file while they run)
    struct SinkHold { std::string& path; std::string saved; explicit SinkHold(std::string& p) : path(p), saved(p) { path.clear(); }
                      ~SinkHold() { path = saved; } } sink_hold(sink_path_);
    if (!strict)
        throw std::runtime_error("the text (" + std::to_string(total) + " characters) does not fit the device as one "
                                 "suffix array (limit " + std::to_string(max_text) + ") and only strict multi-MUMs "
                                 "can be computed by partition + anchor merge (include/pfp_mum.hpp:178-183)");
    // A partition shares the device with what the sequence of partitions keeps: both input buffers (the next
    // partition is uploaded while this one runs: one byte per text character), the threshold columns (this run's, for
    // both strands; the partition's copy; the fold so far: 32 bits each since round 4), the anchor's suffix ranks (64 bits a
    // position beyond 2^32 characters) and the scratch of a fold step -- 32 bytes per anchor base (16 until round 5: a share
    // {anchor + 12} x 3.05 Gbp on a device with 200 GB free ran out of memory twice before its partitions were small enough,
    // tests/big_reserve.py; the attempt loop below catches what this formula still gets wrong).
    // (a partition itself: its text, the tables of its parse and one batch of the producer -- auto_max_text)
    if (auto_limit && !std::getenv("MMT_MAX_TEXT") && !std::getenv("MUMEMTO_MAX_TEXT")) {
        const double budget = 0.95 * (double)pool::available(device_) - 32.0 * (double)doc_len[0] - 24.0 * 1073741824.0;
        const uint64_t fit = budget > 0 ? (uint64_t)(budget / 4.0) : 0;
        max_text = std::min(max_text, std::max<uint64_t>(fit, 1));
    }
    // (one attempt at the whole sequence of partitions with a given limit; an attempt that runs out of device memory -- the
    // estimate above is a formula, the heap has a shape -- is repeated with smaller partitions, below)
    size_t G_used = 0;
    float acc[8] = {0};
    auto attempt = [&](uint64_t max_text) {
    for (float& x : acc) x = 0;
    // contiguous groups of documents 1..N-1, each together with the anchor within max_text
    const uint64_t anchor_chars = mult * (doc_len[0] + 1);
    std::vector<std::pair<size_t, size_t>> groups;      // [first, last) document indices
    for (size_t d = 1; d < n_docs;) {
        uint64_t chars = anchor_chars;
        size_t e = d;
        while (e < n_docs && chars + mult * (doc_len[e] + 1) <= max_text) { chars += mult * (doc_len[e] + 1); e++; }
        if (e == d) throw std::runtime_error("document " + std::to_string(d) + " does not fit next to the anchor");
        groups.emplace_back(d, e);
        d = e;
    }
    const size_t G = groups.size();
    const uint64_t L = doc_len[0] + 1;
    // Every partition is folded into the result so far as soon as it has run (the reference's pairwise fold,
    // merge_candidates.cpp:106-157, left to right): what stays in HBM between partitions is one table of rows and ONE
    // threshold column (2 bytes per anchor position: 6 GB for a human genome), however many partitions there are.
    struct Part {       // one partition's rows and thresholds, copied out of the engine's buffers for the fold
        DevBuf<uint32_t> length, thresh; DevBuf<int64_t> offsets; DevBuf<uint8_t> strands;
        size_t n_rows = 0, n_docs = 0;
    };
    Part first;                 // partition 0 until partition 1 arrives
    MergedRows folded;          // partitions 0 .. g folded, g >= 1
    bool have_folded = false;
    std::vector<uint64_t> sub_len;
    mmt_params q = p;
    q.merge_metadata = 1; q.num_distinct = 0; q.max_total_freq = 0;
    uint64_t max_group_bytes = 0;
    for (size_t g = 0; g < G; g++) max_group_bytes = std::max(max_group_bytes, base[groups[g].second] - base[groups[g].first]);
    // Two input buffers: while partition g is processed, a helper thread copies the documents of partition g + 1
    // from the caller's (pageable) memory into the other buffer on its own stream.  The anchor sits in front of both.
    DevBuf<uint8_t> alt;
    d_bases_own_.ensure(doc_len[0] + max_group_bytes + 16);
    if (G > 1) alt.ensure(doc_len[0] + max_group_bytes + 16);
    uint8_t* buf[2] = {d_bases_own_.get(), G > 1 ? alt.get() : d_bases_own_.get()};
    hipStream_t copy_stream = nullptr;
    MMT_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    std::thread uploader;
    std::exception_ptr upload_error;
    auto upload = [&](size_t g, int slot) {
        try {
            MMT_HIP(hipSetDevice(device_));
            const size_t a = groups[g].first, b = groups[g].second;
            uint64_t at = doc_len[0];
            for (size_t d = a; d < b; d++) {
                if (doc_len[d])
                    MMT_HIP(hipMemcpyAsync(buf[slot] + at, doc_ptr[d], doc_len[d], hipMemcpyHostToDevice, copy_stream));
                at += doc_len[d];
            }
            MMT_HIP(hipStreamSynchronize(copy_stream));
        } catch (...) { upload_error = std::current_exception(); }
    };
    auto finish_upload = [&]() {
        if (uploader.joinable()) uploader.join();
        if (upload_error) { std::exception_ptr e = upload_error; upload_error = nullptr; std::rethrow_exception(e); }
    };
    try {
        if (doc_len[0]) {
            MMT_HIP(hipMemcpyAsync(buf[0], doc_ptr[0], doc_len[0], hipMemcpyHostToDevice, stream_));
            if (G > 1) MMT_HIP(hipMemcpyAsync(buf[1], buf[0], doc_len[0], hipMemcpyDeviceToDevice, stream_));
            MMT_HIP(hipStreamSynchronize(stream_));
        }
        upload(0, 0);
        finish_upload();
        for (size_t g = 0; g < G; g++) {
            const size_t a = groups[g].first, b = groups[g].second;
            if (g + 1 < G) uploader = std::thread(upload, g + 1, (int)((g + 1) & 1));
            sub_len.assign(1, doc_len[0]);
            sub_len.insert(sub_len.end(), doc_len + a, doc_len + b);
            set_input_device(buf[g & 1], sub_len.data(), sub_len.size());
            run(q);
            // the columns of this partition are dead (the next run builds its own; only the last partition's anchor ranks
            // order the result): its windows, batch buffers and tables go BEFORE its rows and thresholds are copied out --
            // the bucket-wise producer sizes its batches by what the heap has left, and a 6 GB threshold column on top of
            // that is what a {anchor + 8 whole genomes} partition did not have room for
            for (int i = 0; i < 7; i++) acc[i] += stage_ms_[i];
            if (g + 1 < G) release_columns();
            else release_columns(true);
            Part later;
            Part& P = g == 0 ? first : later;
            P.n_docs = sub_len.size(); P.n_rows = rows_.n_rows;
            const uint32_t* dl; const int64_t* dof; const uint8_t* dst;
            rows_mum_device(&dl, &dof, &dst);
            const size_t cells = P.n_rows * P.n_docs;
            P.length.ensure(P.n_rows + 1); P.offsets.ensure(cells + 1); P.strands.ensure(cells + 1); P.thresh.ensure(L);
            if (P.n_rows) {
                MMT_HIP(hipMemcpyAsync(P.length.get(), dl, P.n_rows * 4, hipMemcpyDeviceToDevice, stream_));
                MMT_HIP(hipMemcpyAsync(P.offsets.get(), dof, cells * 8, hipMemcpyDeviceToDevice, stream_));
                MMT_HIP(hipMemcpyAsync(P.strands.get(), dst, cells, hipMemcpyDeviceToDevice, stream_));
            }
            MMT_HIP(hipMemcpyAsync(P.thresh.get(), d_thresh_.get(), L * 4, hipMemcpyDeviceToDevice, stream_));
            MMT_HIP(hipStreamSynchronize(stream_));
            if (g >= 1) {
                auto as_partition = [&](mmt_partition& m, size_t rows, size_t docs, const uint32_t* len, const int64_t* off,
                                        const uint8_t* st, const uint32_t* th) {
                    m.n_rows = rows; m.n_docs = docs; m.length = len; m.offsets = off; m.strands = st;
                    m.thresh = reinterpret_cast<const uint16_t*>(th); m.thresh_bits = 32;
                    m.thresh_len = L; m.thresh_on_device = 1; m.rows_on_device = 1;
                };
                mmt_partition two[2];
                if (have_folded)
                    as_partition(two[0], folded.n_rows, folded.n_docs, folded.d_length.get(), folded.d_offsets.get(),
                                 folded.d_strands.get(), folded.d_thresh.get());
                else
                    as_partition(two[0], first.n_rows, first.n_docs, first.length.get(), first.offsets.get(),
                                 first.strands.get(), first.thresh.get());
                as_partition(two[1], P.n_rows, P.n_docs, P.length.get(), P.offsets.get(), P.strands.get(), P.thresh.get());
                MergedRows next = anchor_merge(*this, two, 2, p.min_match_len);
                folded = std::move(next);
                have_folded = true;
                first.length.release(); first.offsets.release(); first.strands.release(); first.thresh.release();
            }
            finish_upload();
        }
    } catch (...) {
        if (uploader.joinable()) uploader.join();
        (void)hipStreamDestroy(copy_stream);
        throw;
    }
    (void)hipStreamDestroy(copy_stream);
    if (have_folded) merged_ = std::move(folded);
    else {                                       // a single partition: the fold of one (filters by the minimum length)
        mmt_partition one;
        one.n_rows = first.n_rows; one.n_docs = first.n_docs; one.length = first.length.get(); one.offsets = first.offsets.get();
        one.strands = first.strands.get(); one.thresh = reinterpret_cast<const uint16_t*>(first.thresh.get()); one.thresh_bits = 32;
        one.thresh_len = L; one.thresh_on_device = 1; one.rows_on_device = 1;
        merged_ = anchor_merge(*this, &one, 1, p.min_match_len);
    }
    G_used = G;
    };
    for (int tries = 0;; tries++) {
        try { attempt(max_text); break; }
        catch (const DeviceOom&) {
            uint64_t longest = 0;
            for (size_t d = 1; d < n_docs; d++) longest = std::max<uint64_t>(longest, doc_len[d]);
            const uint64_t smallest = mult * (doc_len[0] + 1) + mult * (longest + 1);      // the anchor and the longest other document
            if (!auto_limit || tries >= 4 || max_text <= smallest) throw;
            forget_last_run();                 // everything the failed attempt held goes back to the heap
            max_text = std::max<uint64_t>(smallest, (uint64_t)(0.6 * (double)max_text));
            if (std::getenv("MUMEMTO_TIMING") || std::getenv("MMT_MEM_TRACE"))
                std::fprintf(stderr, "[partitions] out of device memory: once more with at most %llu text characters a partition\n",
                             (unsigned long long)max_text);
        }
    }
    sort_like_direct(*this, merged_);          // the last partition's suffix ranks order the anchor positions
    size_t text_bytes = 0;
    const char* text = stage_merged_text(*this, merged_, &text_bytes);     // page-locked, stays with the engine
    merged_text_.clear();
    // publish as the result of this "run"; the per-partition input buffers are gone or about to go: a later run()
    // needs a new set_input
    d_bases_ = nullptr;
    input_valid_ = false;
    doc_len_.assign(doc_len, doc_len + n_docs);
    HostRows& R = rows_;
    R = HostRows();
    rows_pending_ = ROWS_ARRAYS;               // the merged tables stay in HBM until somebody asks (Engine::fetch_rows)
    R.mum_mode = true; R.n_docs = n_docs; R.n_rows = merged_.n_rows;
    h_occ_start_.ensure(2); h_occ_start_.get()[0] = 0; R.occ_start = h_occ_start_.get();
    R.text = text; R.text_len = text_bytes;
    bumbl_.clear();
    num_distinct_eff_ = n_docs;
    partitions_used_ = G_used;
    merged_thresh_valid_ = true;
    thresh16_valid_ = false;
    for (int i = 0; i < 7; i++) stage_ms_[i] = acc[i];
    stage_ms_[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace mmt
