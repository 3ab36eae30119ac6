// api.cpp -- the C ABI of libmumemto: the drop-in symbols of include/mumemto.h
// (reference: mumemto_library/mumemto_api.cpp:489-644) and the device-resident
// entry points of include/mumemto_gpu.h.
#include <chrono>
#include <thread>
#include <deque>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mumemto.h"
#include "../../include/mumemto_gpu.h"
#include "engine.hpp"
#include "dist.hpp"
#include "fasta.hpp"
#include "merge.hpp"

namespace {

thread_local std::string g_last_error;   // mumemto_api.cpp:440-448

int fail(int rc, const std::string& msg) { g_last_error = msg; return rc; }

// One engine per process for the host-string ABI; calls are serialised
// (SURVEY.md 8(b) "Threading": device use must be serialised or stream-safe).
std::mutex g_engine_mu;
std::unique_ptr<mmt::Engine> g_engine;

mmt::Engine& shared_engine() {
    if (!g_engine) {
        int dev = 0;
        if (const char* s = std::getenv("MUMEMTO_DEVICE")) dev = std::atoi(s);
        g_engine.reset(new mmt::Engine(dev, nullptr));
    }
    return *g_engine;
}

struct DocInput {
    std::vector<uint8_t> bases;
    std::vector<uint64_t> doc_len;
    std::vector<size_t> doc_record_offsets, record_lengths;
};

// build_sequences_from_docs + compute_record_lengths + flatten_lengths
// (mumemto_api.cpp:315-330, :450-485): records of a document are concatenated,
// a NULL record is the empty string, lengths are the raw record byte lengths.
DocInput gather_docs(const mumemto_doc_view* docs, size_t n) {
    DocInput in;
    in.doc_record_offsets.push_back(0);
    for (size_t d = 0; d < n; d++) {
        uint64_t total = 0;
        for (size_t r = 0; r < docs[d].num_records; r++) {
            const char* s = docs[d].records ? docs[d].records[r] : nullptr;
            size_t len = s ? std::strlen(s) : 0;
            in.record_lengths.push_back(len);
            if (len) in.bases.insert(in.bases.end(), reinterpret_cast<const uint8_t*>(s),
                                     reinterpret_cast<const uint8_t*>(s) + len);
            total += len;
        }
        in.doc_len.push_back(total);
        in.doc_record_offsets.push_back(in.record_lengths.size());
    }
    return in;
}

}  // namespace

struct mumemto_mum_result {
    size_t num_docs = 0;
    std::vector<size_t> doc_record_offsets, record_lengths;
    std::vector<uint32_t> length;
    std::vector<int64_t> offsets;
    std::vector<uint8_t> strands;
};
struct mumemto_mem_result {
    size_t num_docs = 0;
    std::vector<size_t> doc_record_offsets, record_lengths;
    std::vector<uint32_t> length;
    std::vector<uint64_t> occ_start;
    std::vector<int64_t> offsets;
    std::vector<size_t> seq_ids;
    std::vector<uint8_t> strands;
};
struct mmt_engine {
    std::unique_ptr<mmt::Engine> e;
    mmt::HostArena arena;              // host buffer of mmt_engine_run_files, kept between calls
};
struct mmt_comm {
    mmt::Comm* c = nullptr;
    int rank = 0;
    std::string text;                  // rank 0: the gathered output of the last mmt_dist_gather_text
};
struct mmt_merged {
    mmt::MergedRows rows;
    std::string text;
    mmt::Engine* engine = nullptr;     // the engine whose device and stream the rows live on
    bool text_valid = false;
};

extern "C" {

const char* mumemto_last_error(void) { return g_last_error.c_str(); }
const char* mmt_last_error(void) { return g_last_error.c_str(); }

int mumemto_mum(const mumemto_doc_view* docs, size_t n_docs, uint32_t min_match_len, uint8_t use_revcomp,
                size_t num_distinct, uint8_t /*use_gsacak*/, mumemto_mum_result** out_result) {
    if (!out_result) return fail(1, "out_result must be non-null");
    *out_result = nullptr;
    try {
        if (!docs && n_docs != 0) return fail(2, "docs must be non-null when num_docs != 0");
        DocInput in = gather_docs(docs, n_docs);
        std::unique_ptr<mumemto_mum_result> r(new mumemto_mum_result());
        r->num_docs = n_docs;
        r->doc_record_offsets = std::move(in.doc_record_offsets);
        r->record_lengths = std::move(in.record_lengths);
        if (n_docs) {
            std::lock_guard<std::mutex> lock(g_engine_mu);
            mmt::Engine& e = shared_engine();
            e.set_input_host(in.bases.data(), in.doc_len.data(), n_docs);
            mmt_params p{};
            p.min_match_len = min_match_len; p.num_distinct = num_distinct;
            p.max_doc_freq = 1; p.max_total_freq = 0;                 // mumemto_api.cpp:353
            p.use_revcomp = use_revcomp ? 1 : 0; p.merge_metadata = 0;
            e.run(p);
            const mmt::HostRows& R = e.rows();
            r->length.assign(R.length, R.length + R.n_rows);
            r->offsets.assign(R.mum_offsets, R.mum_offsets + R.n_rows * n_docs);
            r->strands.assign(R.mum_strands, R.mum_strands + R.n_rows * n_docs);
        }
        *out_result = r.release();
        return 0;
    } catch (const std::exception& ex) {
        return fail(3, ex.what());
    } catch (...) {
        return fail(4, "unknown exception");
    }
}

int mumemto_mem(const mumemto_doc_view* docs, size_t n_docs, uint32_t min_match_len, uint8_t use_revcomp,
                size_t num_distinct, size_t max_total_freq, size_t max_doc_freq, uint8_t /*use_gsacak*/,
                mumemto_mem_result** out_result) {
    if (!out_result) return fail(1, "out_result must be non-null");
    *out_result = nullptr;
    try {
        if (!docs && n_docs != 0) return fail(2, "docs must be non-null when num_docs != 0");
        DocInput in = gather_docs(docs, n_docs);
        std::unique_ptr<mumemto_mem_result> r(new mumemto_mem_result());
        r->num_docs = n_docs;
        r->doc_record_offsets = std::move(in.doc_record_offsets);
        r->record_lengths = std::move(in.record_lengths);
        if (n_docs) {
            if (max_doc_freq <= 1)                                     // mumemto_api.cpp:381-383
                throw std::invalid_argument("per-sequence MEM frequency f must be > 1 (use mumemto_mum instead)");
            std::lock_guard<std::mutex> lock(g_engine_mu);
            mmt::Engine& e = shared_engine();
            e.set_input_host(in.bases.data(), in.doc_len.data(), n_docs);
            mmt_params p{};
            p.min_match_len = min_match_len; p.num_distinct = num_distinct;
            p.max_doc_freq = (int64_t)max_doc_freq; p.max_total_freq = (int64_t)max_total_freq;
            p.use_revcomp = use_revcomp ? 1 : 0; p.merge_metadata = 0;
            e.run(p);
            const mmt::HostRows& R = e.rows();
            r->length.assign(R.length, R.length + R.n_rows);
            r->occ_start.assign(R.occ_start, R.occ_start + R.n_rows + 1);
            r->offsets.assign(R.mem_offsets, R.mem_offsets + R.n_occ);
            r->seq_ids.assign(R.mem_docs, R.mem_docs + R.n_occ);
            r->strands.assign(R.mem_strands, R.mem_strands + R.n_occ);
        }
        if (r->occ_start.empty()) r->occ_start.push_back(0);
        *out_result = r.release();
        return 0;
    } catch (const std::exception& ex) {
        return fail(3, ex.what());
    } catch (...) {
        return fail(4, "unknown exception");
    }
}

size_t num_docs(const mumemto_mum_result* r) { return r ? r->num_docs : 0; }
const size_t* doc_record_offsets(const mumemto_mum_result* r) { return r ? r->doc_record_offsets.data() : nullptr; }
const size_t* record_lengths(const mumemto_mum_result* r) { return r ? r->record_lengths.data() : nullptr; }
size_t num_mums(const mumemto_mum_result* r) { return r ? r->length.size() : 0; }
mumemto_mum_match_view mum_at(const mumemto_mum_result* r, size_t idx) {
    mumemto_mum_match_view v{};
    if (!r || idx >= r->length.size()) return v;
    v.length = r->length[idx];
    v.offsets = r->offsets.data() + idx * r->num_docs;
    v.strands = r->strands.data() + idx * r->num_docs;
    return v;
}
void mum_free(mumemto_mum_result* r) { delete r; }

size_t num_docs_mem(const mumemto_mem_result* r) { return r ? r->num_docs : 0; }
const size_t* doc_record_offsets_mem(const mumemto_mem_result* r) { return r ? r->doc_record_offsets.data() : nullptr; }
const size_t* record_lengths_mem(const mumemto_mem_result* r) { return r ? r->record_lengths.data() : nullptr; }
size_t num_mems(const mumemto_mem_result* r) { return r ? r->length.size() : 0; }
mumemto_mem_match_view mem_at(const mumemto_mem_result* r, size_t idx) {
    mumemto_mem_match_view v{};
    if (!r || idx >= r->length.size()) return v;
    v.length = r->length[idx];
    v.occurrences = (size_t)(r->occ_start[idx + 1] - r->occ_start[idx]);
    v.offsets = r->offsets.data() + r->occ_start[idx];
    v.seq_ids = r->seq_ids.data() + r->occ_start[idx];
    v.strands = r->strands.data() + r->occ_start[idx];
    return v;
}
void mem_free(mumemto_mem_result* r) { delete r; }

// ---------------------------------------------------------------------------------
// device-resident entry points
// ---------------------------------------------------------------------------------
#define MMT_TRY try {
#define MMT_CATCH                                                           \
    return 0;                                                                \
    } catch (const std::exception& ex) { return fail(3, ex.what()); }        \
    catch (...) { return fail(4, "unknown exception"); }

int mmt_engine_create(int device, void* hip_stream, mmt_engine** out) {
    if (!out) return fail(1, "out must be non-null");
    *out = nullptr;
    MMT_TRY
    std::unique_ptr<mmt_engine> h(new mmt_engine());
    h->e.reset(new mmt::Engine(device, reinterpret_cast<hipStream_t>(hip_stream)));
    *out = h.release();
    MMT_CATCH
}
void mmt_engine_destroy(mmt_engine* e) { delete e; }

int mmt_engine_set_input_device(mmt_engine* e, const uint8_t* d_bases, const uint64_t* doc_len, size_t n_docs) {
    if (!e) return fail(1, "engine must be non-null");
    MMT_TRY
    e->e->set_input_device(d_bases, doc_len, n_docs);
    MMT_CATCH
}
int mmt_engine_set_input_host(mmt_engine* e, const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs) {
    if (!e) return fail(1, "engine must be non-null");
    MMT_TRY
    e->e->set_input_host(h_bases, doc_len, n_docs);
    MMT_CATCH
}
int mmt_engine_set_text_host(mmt_engine* e, const uint8_t* text, uint64_t n, const uint64_t* doc_len, size_t n_docs,
                             int use_revcomp) {
    if (!e || (!text && n) || (!doc_len && n_docs)) return fail(1, "engine, text and doc_len must be non-null");
    MMT_TRY
    e->e->set_text_host(text, n, doc_len, n_docs, use_revcomp != 0);
    MMT_CATCH
}
int mmt_engine_set_stream_host(mmt_engine* e, const uint32_t* sa, const uint32_t* lcp, const uint8_t* bwt,
                               uint64_t entries, const uint64_t* doc_len, size_t n_docs, int use_revcomp) {
    if (!e || ((!sa || !lcp || !bwt) && entries) || (!doc_len && n_docs))
        return fail(1, "engine, columns and doc_len must be non-null");
    MMT_TRY
    e->e->set_stream_host(sa, lcp, bwt, entries, doc_len, n_docs, use_revcomp != 0);
    MMT_CATCH
}
int mmt_engine_run(mmt_engine* e, const mmt_params* p) {
    if (!e || !p) return fail(1, "engine and params must be non-null");
    MMT_TRY
    e->e->run(*p);
    MMT_CATCH
}

int mmt_engine_run_partitioned(mmt_engine* e, const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs,
                               const mmt_params* p, uint64_t max_text_chars) {
    if (!e || !p || (!doc_len && n_docs)) return fail(1, "engine, params and doc_len must be non-null");
    MMT_TRY
    e->e->run_partitioned_host(h_bases, doc_len, n_docs, *p, max_text_chars);
    MMT_CATCH
}
int mmt_engine_run_supplied(mmt_engine* e, mmt_doc_supplier supplier, void* user, const uint64_t* doc_len, size_t n_docs,
                            const mmt_params* p) {
    if (!e || !p || !supplier || (!doc_len && n_docs)) return fail(1, "engine, params, supplier and doc_len must be non-null");
    MMT_TRY
    e->e->run_supplied(supplier, user, doc_len, n_docs, *p);
    MMT_CATCH
}
// build_main (src/pfp_mum.cpp:31-159) in-process: FASTA files -> text -> stream -> scan -> PREFIX.mums | .mems +
// PREFIX.lengths.  The same reader and the same engine entry as mumemto_exec.
int mmt_engine_run_files(mmt_engine* e, const char* const* paths, size_t n_paths, const mmt_params* p,
                         const char* out_prefix, uint64_t max_text_chars, double seconds[4]) {
    if (!e || !p || (!paths && n_paths)) return fail(1, "engine, params and paths must be non-null");
    MMT_TRY
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<std::string> inputs(paths, paths + n_paths);
    std::vector<mmt::FastaDoc> docs;
    mmt::HostDocs hd;
    // the host buffer the files are parsed into stays with the engine handle: the next call reuses its pages
    // (sending every document to the device from the thread that parsed it, while the other files are still being
    // read, was tried: 94 concurrent copies from pageable memory slow the parsers down by more than the copies take --
    // read 0.18 -> 0.37 s, run 1.67 -> 1.57 s on the C3 stand-in)
    // A collection that will run as one suffix array (judged by the file sizes, which bound the bases) goes to the device
    // document by document while the other files are still being read: ONE copier thread takes the documents in the order
    // the readers finish them (every reader copying its own document was tried: 94 concurrent copies from pageable
    // memory slowed the parsers down by more than the copies take).  MUMEMTO_NO_UPLOAD_OVERLAP switches it off.
    struct Upload {
        std::mutex mu; std::condition_variable cv; std::deque<std::pair<size_t, uint64_t>> q; bool done = false;
        std::thread thread; std::exception_ptr error; std::vector<size_t> slot; bool on = false;
    } up;
    mmt::ReadHooks hooks;
    hooks.layout = [&](const uint8_t* arena, size_t bytes, const std::vector<size_t>& slot, bool all_in_arena) {
        const uint64_t bound = 2 * ((uint64_t)bytes + slot.size());               // text characters at most
        if (!all_in_arena || std::getenv("MUMEMTO_NO_UPLOAD_OVERLAP") || slot.size() < 2) return;
        e->e->forget_last_run();
        if (bound > (max_text_chars ? max_text_chars : e->e->auto_max_text())) return;
        uint8_t* dev = e->e->begin_input_slots(bytes);
        up.slot = slot; up.on = true;
        const int device = e->e->device();
        up.thread = std::thread([&up, dev, arena, device]() {
            try {
                MMT_HIP(hipSetDevice(device));
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
                        MMT_HIP(hipMemcpyAsync(dev + up.slot[job.first], arena + up.slot[job.first], job.second, hipMemcpyHostToDevice, cs));
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
    long empty = -1;
    try { empty = mmt::read_fasta_collection(inputs, docs, e->arena, hd, &hooks); }
    catch (...) { finish_upload(); throw; }
    finish_upload();
    if (up.error) std::rethrow_exception(up.error);
    if (empty >= 0) throw std::runtime_error("Empty input file found: " + inputs[(size_t)empty]);
    const double t_read = since();
    // the output file is written while the run goes on when the run streams its rows (Engine::set_text_sink)
    e->e->set_text_sink(out_prefix ? std::string(out_prefix) + (p->max_doc_freq == 1 ? ".mums" : ".mems") : std::string());
    struct SinkOff { mmt::Engine* e; ~SinkOff() { e->set_text_sink(std::string()); } } sink_off{e->e.get()};
    bool ran = false;
    if (up.on) {
        try {
            e->e->finish_input_slots(up.slot, hd.len.data(), hd.len.size());
            e->e->run_once_dropping_input(*p);
            ran = true;
        } catch (const mmt::DeviceOom&) {                      // did not fit after all: the route for any size, from the host copies
        }
    }
    if (!ran) e->e->run_partitioned_docs(hd.ptr.data(), hd.len.data(), hd.len.size(), *p, max_text_chars);
    const double t_run = since();
    if (out_prefix) {
        // (D2H into page-locked memory, then one write(): copying from HBM straight into a populated mapping of the file
        // was slower, 0.27 against 0.21 s for 894 MB on tmpfs)
        e->e->write_text_file(std::string(out_prefix) + (e->e->rows_meta().mum_mode ? ".mums" : ".mems"));
        mmt::write_lengths_file(out_prefix, docs);
    }
    if (seconds) { seconds[0] = t_read; seconds[1] = t_run - t_read; seconds[2] = since() - t_run; seconds[3] = since(); }
    MMT_CATCH
}
size_t mmt_partitions_used(const mmt_engine* e) { return e ? e->e->partitions_used() : 0; }
int mmt_copy_merged_thresh(const mmt_engine* e, uint16_t* out) {
    if (!e) return fail(1, "null");
    if (!e->e->last_run_partitioned()) return fail(3, "the last run was not partitioned");
    const std::vector<uint16_t>& t = e->e->merged_thresh();
    std::memcpy(out, t.data(), t.size() * 2);
    return 0;
}

size_t mmt_num_rows(const mmt_engine* e) { return e ? e->e->rows_meta().n_rows : 0; }
size_t mmt_num_docs(const mmt_engine* e) { return e ? e->e->n_docs() : 0; }
int mmt_rows_mum(const mmt_engine* e, uint32_t* length, int64_t* offsets, uint8_t* strands) {
    if (!e) return fail(1, "engine must be non-null");
    if (!e->e->rows_meta().mum_mode) return fail(3, "last run was not in MUM mode");
    MMT_TRY
    // (a run whose rows left with their windows -- Engine::set_text_sink over a text that fills the device -- throws here: the
    // file and the count are all it answers for)
    const mmt::HostRows& R = e->e->rows(mmt::Engine::ROWS_ARRAYS);
    if (R.n_rows && R.length && R.mum_offsets && R.mum_strands) {
        std::memcpy(length, R.length, R.n_rows * 4);
        std::memcpy(offsets, R.mum_offsets, R.n_rows * R.n_docs * 8);
        std::memcpy(strands, R.mum_strands, R.n_rows * R.n_docs);
    }
    MMT_CATCH
}
int mmt_rows_mum_device(const mmt_engine* e, const uint32_t** length, const int64_t** offsets,
                        const uint8_t** strands) {
    if (!e || !length || !offsets || !strands) return fail(1, "engine and outputs must be non-null");
    if (!e->e->rows_meta().mum_mode) return fail(3, "last run was not in MUM mode");
    e->e->rows_mum_device(length, offsets, strands);
    return 0;
}
size_t mmt_num_occ(const mmt_engine* e) { return e ? e->e->rows_meta().n_occ : 0; }
int mmt_rows_mem(const mmt_engine* e, uint32_t* length, uint64_t* occ_start, int64_t* offsets, uint64_t* seq_ids,
                 uint8_t* strands) {
    if (!e) return fail(1, "engine must be non-null");
    if (e->e->rows_meta().mum_mode) return fail(3, "last run was in MUM mode");
    MMT_TRY
    const mmt::HostRows& R = e->e->rows(mmt::Engine::ROWS_ARRAYS);
    occ_start[0] = 0;
    if (R.n_rows && R.length && R.occ_start && R.mem_offsets && R.mem_docs && R.mem_strands) {
        std::memcpy(length, R.length, R.n_rows * 4);
        std::memcpy(occ_start, R.occ_start, (R.n_rows + 1) * 8);
        std::memcpy(offsets, R.mem_offsets, R.n_occ * 8);
        std::memcpy(seq_ids, R.mem_docs, R.n_occ * 8);
        std::memcpy(strands, R.mem_strands, R.n_occ);
    }
    MMT_CATCH
}
const char* mmt_output_text(mmt_engine* e, size_t* len) {
    if (!e) { if (len) *len = 0; return nullptr; }
    try {
        const mmt::HostRows& R = e->e->rows(mmt::Engine::ROWS_TEXT);
        if (len) *len = R.text_len;
        return R.text;
    } catch (const std::exception& ex) { fail(3, ex.what()); if (len) *len = 0; return nullptr; }
}
const uint8_t* mmt_output_bumbl(mmt_engine* e, size_t* len) {
    if (!e) { if (len) *len = 0; return nullptr; }
    const std::string& t = e->e->bumbl();
    if (len) *len = t.size();
    return reinterpret_cast<const uint8_t*>(t.data());
}
size_t mmt_thresh_len(const mmt_engine* e) { return e ? e->e->thresh_len() : 0; }
int mmt_copy_thresh(const mmt_engine* e, uint16_t* out) {
    if (!e) return fail(1, "engine must be non-null");
    MMT_TRY
    e->e->copy_thresh(out);
    MMT_CATCH
}
const uint16_t* mmt_thresh_device(const mmt_engine* e) {
    if (!e) return nullptr;
    try { return e->e->thresh_device(); } catch (const std::exception& ex) { fail(3, ex.what()); return nullptr; }
}
int mmt_copy_thresh32(const mmt_engine* e, uint32_t* out) {
    if (!e) return fail(1, "engine must be non-null");
    MMT_TRY
    e->e->copy_thresh32(out);
    MMT_CATCH
}
const uint32_t* mmt_thresh_device32(const mmt_engine* e) { return e ? e->e->thresh_device32() : nullptr; }

uint64_t mmt_text_length(const mmt_engine* e) { return e ? e->e->text_length() : 0; }
int mmt_copy_text(const mmt_engine* e, uint8_t* out) { if (!e) return fail(1, "null"); MMT_TRY e->e->copy_text(out); MMT_CATCH }
int mmt_copy_sa(const mmt_engine* e, uint32_t* out) { if (!e) return fail(1, "null"); MMT_TRY e->e->copy_sa(out); MMT_CATCH }
int mmt_copy_lcp(const mmt_engine* e, uint32_t* out) { if (!e) return fail(1, "null"); MMT_TRY e->e->copy_lcp(out); MMT_CATCH }
int mmt_copy_bwt(const mmt_engine* e, uint8_t* out) { if (!e) return fail(1, "null"); MMT_TRY e->e->copy_bwt(out); MMT_CATCH }
size_t mmt_num_candidates(const mmt_engine* e) { return e ? e->e->n_candidates() : 0; }
int mmt_copy_candidates(const mmt_engine* e, uint32_t* out) {
    if (!e) return fail(1, "null");
    MMT_TRY
    e->e->copy_candidates(out);
    MMT_CATCH
}
int mmt_stage_ms(const mmt_engine* e, float out[8]) {
    if (!e) return fail(1, "null");
    std::memcpy(out, e->e->stage_ms(), 8 * sizeof(float));
    return 0;
}
int mmt_column_bytes(const mmt_engine* e, uint32_t out[3]) {
    if (!e) return fail(1, "null");
    out[0] = e->e->wide() ? 5 : 4; out[1] = 4; out[2] = 1;   // SA 32 (+8) bits, LCP u32, BWT u8 as stored
    return 0;
}
int mmt_copy_sa64(const mmt_engine* e, uint64_t* out) { if (!e) return fail(1, "null"); MMT_TRY e->e->copy_sa64(out); MMT_CATCH }
int mmt_is_wide(const mmt_engine* e) { return e && e->e->wide() ? 1 : 0; }
size_t mmt_scan_ranges(const mmt_engine* e) { return e ? e->e->scan_ranges() : 0; }
int mmt_engine_set_stream_host40(mmt_engine* e, const uint32_t* sa_lo, const uint8_t* sa_hi, const uint32_t* lcp,
                                 const uint8_t* bwt, uint64_t entries, const uint64_t* doc_len, size_t n_docs,
                                 int use_revcomp) {
    if (!e || ((!sa_lo || !sa_hi || !lcp || !bwt) && entries) || (!doc_len && n_docs))
        return fail(1, "engine, columns and doc_len must be non-null");
    MMT_TRY
    e->e->set_stream_host40(sa_lo, sa_hi, lcp, bwt, entries, doc_len, n_docs, use_revcomp != 0);
    MMT_CATCH
}
void mmt_pool_set_reserve(unsigned long long bytes) { mmt::pool::set_reserve(bytes == ~0ull ? ~(size_t)0 : (size_t)bytes); }
void mmt_pool_trim(void) {
    try { mmt::merge_release_scratch(); } catch (...) {}
    // (the engine the mumemto_library entry points share keeps its buffers between calls: while it lives the heap holds live
    // blocks and nothing is unmapped.  Its results have been copied out by the call that made them; the next call makes a new one.)
    try { std::lock_guard<std::mutex> lock(g_engine_mu); g_engine.reset(); } catch (...) {}
    mmt::pool::trim();
}
int mmt_engine_set_text_sink(mmt_engine* e, const char* path) {
    if (!e) return fail(1, "null");
    MMT_TRY
    e->e->set_text_sink(path ? std::string(path) : std::string());
    MMT_CATCH
}
int mmt_engine_release_columns(mmt_engine* e, int keep_anchor_ranks) {
    if (!e) return fail(1, "null");
    MMT_TRY
    e->e->release_columns(keep_anchor_ranks != 0);
    MMT_CATCH
}
int mmt_engine_set_scan_shard(mmt_engine* e, uint32_t index, uint32_t count) {
    if (!e) return fail(1, "null");
    MMT_TRY
    e->e->set_scan_shard(index, count);
    MMT_CATCH
}
int mmt_engine_keep_columns(mmt_engine* e, int on) {
    if (!e) return fail(1, "null");
    e->e->set_keep_columns(on);
    return 0;
}
int mmt_columns_kept(const mmt_engine* e) { return e && e->e->columns_kept() ? 1 : 0; }
int mmt_stream_stats(const mmt_engine* e, uint64_t out[4]) {
    if (!e) return fail(1, "null");
    e->e->stream_stats(out);
    return 0;
}
size_t mmt_sort_pieces(const mmt_engine* e, uint64_t* first, uint64_t* count, size_t capacity) {
    if (!e) return 0;
    const auto& p = e->e->sort_pieces();
    for (size_t i = 0; i < p.size() && i < capacity; i++) { if (first) first[i] = p[i].first; if (count) count[i] = p[i].second; }
    return p.size();
}
int mmt_device_memory(const mmt_engine* e, uint64_t out[4]) {
    if (!e) return fail(1, "null");
    const mmt::pool::Stats s = mmt::pool::stats(e->e->device());
    out[0] = s.mapped; out[1] = s.live; out[2] = s.peak; out[3] = (uint64_t)(s.map_seconds * 1e6);
    return 0;
}

int mmt_engine_set_producer(mmt_engine* e, int kind, uint32_t w, uint32_t p) {
    if (!e) return fail(1, "null");
    if (kind < 0 || kind > 4) return fail(3, "producer must be 0 (auto), 1 (direct), 2 (pfp), 3 (guided) or 4 (guided + expansion)");
    e->e->set_producer(kind, w, p);
    return 0;
}
int mmt_abi_version(void) { return 6; }
int mmt_text_sink_digest(const mmt_engine* e, uint64_t out[2]) {
    if (!e || !out) return fail(1, "null");
    e->e->text_sink_digest(out);
    return 0;
}
int mmt_kmer_in_share(const mmt_engine* e, const uint8_t* kmer, size_t k) { return e && kmer ? e->e->kmer_in_share(kmer, k) : -1; }
int mmt_engine_set_row_tap(mmt_engine* e, const uint8_t* kmers, size_t n, size_t k, size_t max_rows, size_t max_occ) {
    if (!e || (n && !kmers)) return fail(1, "engine and k-mers must be non-null");
    MMT_TRY
    e->e->set_row_tap(kmers, n, k, max_rows, max_occ);
    MMT_CATCH
}
int mmt_row_tap_counts(mmt_engine* e, uint64_t out[2]) {
    if (!e || !out) return fail(1, "null");
    MMT_TRY
    e->e->row_tap_counts(out);
    MMT_CATCH
}
int mmt_row_tap_get(mmt_engine* e, uint32_t* length, uint64_t* occ_start, uint64_t* sa) {
    if (!e || !length || !occ_start || !sa) return fail(1, "null");
    MMT_TRY
    e->e->row_tap_get(length, occ_start, sa);
    MMT_CATCH
}
int mmt_kmer_positions(mmt_engine* e, const uint8_t* kmers, size_t n, size_t k, uint64_t* pos, uint32_t* which, uint64_t cap,
                       uint64_t* found) {
    if (!e || !kmers || !pos || !which || !found) return fail(1, "null");
    MMT_TRY
    *found = e->e->kmer_positions(kmers, n, k, pos, which, cap);
    MMT_CATCH
}
int mmt_producer_used(const mmt_engine* e) { return e ? e->e->producer_used() : 0; }
int mmt_producer_expanded(const mmt_engine* e) { return e && e->e->producer_expanded() ? 1 : 0; }
int mmt_producer_stats(const mmt_engine* e, uint64_t out[4]) {
    if (!e || !out) return fail(1, "null");
    e->e->producer_stats(out);
    return 0;
}
int mmt_engine_parse_only(mmt_engine* e, uint8_t use_revcomp, uint32_t w, uint32_t p) {
    if (!e) return fail(1, "null");
    MMT_TRY
    e->e->parse_only(use_revcomp != 0, w ? w : 10, p ? p : 100);
    MMT_CATCH
}
int mmt_pfp_counts(const mmt_engine* e, uint64_t out[8]) {
    if (!e) return fail(1, "null");
    const mmt::PfpState& S = e->e->pfp_state();
    out[0] = S.n_phrases; out[1] = S.n_distinct; out[2] = S.dict_len; out[3] = S.n_groups;
    out[4] = (uint64_t)S.rounds_dict; out[5] = (uint64_t)S.rounds_parse; out[6] = S.n_entries; out[7] = S.n_fallback;
    return 0;
}
long long mmt_pfp_run_refined(const mmt_engine* e) {
    if (!e) return -1;
    return (long long)e->e->pfp_state().run_refined;
}
int mmt_pfp_copy_dict(mmt_engine* e, uint8_t* out) {
    if (!e) return fail(1, "null");
    MMT_TRY
    std::vector<uint8_t> v;
    e->e->pfp_copy_dict(v);
    std::memcpy(out, v.data(), v.size());
    MMT_CATCH
}
int mmt_pfp_copy_parse(mmt_engine* e, uint32_t* out) {
    if (!e) return fail(1, "null");
    MMT_TRY
    std::vector<uint32_t> v;
    e->e->pfp_copy_parse(v);
    std::memcpy(out, v.data(), v.size() * 4);
    MMT_CATCH
}
int mmt_pfp_stage_ms(const mmt_engine* e, float out[8]) {
    if (!e) return fail(1, "null");
    std::memcpy(out, e->e->pfp_state().ms, 8 * sizeof(float));
    return 0;
}

// ---- anchor merge --------------------------------------------------------------------
int mmt_anchor_merge(mmt_engine* e, const mmt_partition* parts, size_t k, mmt_merged** out) {
    return mmt_anchor_merge_min_len(e, parts, k, 20, out);      // src/merge_candidates.cpp:141
}
// mmt_partition::thresh_bits took a byte that used to be padding: a caller that fills the struct field by field without
// zeroing it hands over garbage there, and 32 by accident would read 4 L bytes from a 2 L-byte column
static bool bad_thresh_bits(const mmt_partition* parts, size_t k) {
    for (size_t i = 0; i < k; i++)
        if (parts[i].thresh_bits != 0 && parts[i].thresh_bits != 16 && parts[i].thresh_bits != 32) return true;
    return false;
}
int mmt_anchor_merge_min_len(mmt_engine* e, const mmt_partition* parts, size_t k, uint32_t min_len, mmt_merged** out) {
    if (!e || !parts || !out) return fail(1, "engine, parts and out must be non-null");
    if (bad_thresh_bits(parts, k)) return fail(3, "mmt_partition.thresh_bits must be 0, 16 or 32 (zero-initialise the struct)");
    *out = nullptr;
    MMT_TRY
    if (k < 2) throw std::invalid_argument("anchor merge requires at least two partitions");
    std::unique_ptr<mmt_merged> m(new mmt_merged());
    m->rows = mmt::anchor_merge(*e->e, parts, k, min_len);
    m->engine = e->e.get();
    *out = m.release();
    MMT_CATCH
}
int mmt_fold_slice_bounds(uint64_t thresh_len, int world, int r, size_t k, uint32_t longest, uint64_t bounds[3]) {
    if (!bounds || world < 1 || r < 0 || r >= world) return fail(1, "slice out of range");
    mmt::fold_slice_bounds(thresh_len, world, r, mmt::fold_margin(k, longest), &bounds[0], &bounds[1], &bounds[2]);
    return 0;
}
int mmt_anchor_merge_by_ranges(mmt_engine* e, const mmt_partition* parts, size_t k, int slices, uint32_t min_len,
                               mmt_merged** out) {
    if (!e || !parts || !out) return fail(1, "engine, parts and out must be non-null");
    if (bad_thresh_bits(parts, k)) return fail(3, "mmt_partition.thresh_bits must be 0, 16 or 32 (zero-initialise the struct)");
    *out = nullptr;
    MMT_TRY
    if (k < 2) throw std::invalid_argument("anchor merge requires at least two partitions");
    std::unique_ptr<mmt_merged> m(new mmt_merged());
    m->rows = mmt::anchor_merge_by_ranges(*e->e, parts, k, slices, min_len);
    m->engine = e->e.get();
    *out = m.release();
    MMT_CATCH
}
size_t mmt_merged_rows(const mmt_merged* m) { return m ? m->rows.n_rows : 0; }
size_t mmt_merged_docs(const mmt_merged* m) { return m ? m->rows.n_docs : 0; }
int mmt_merged_get(mmt_merged* m, uint32_t* length, int64_t* offsets, uint8_t* strands, uint16_t* thresh) {
    if (!m) return fail(1, "null");
    MMT_TRY
    mmt::download_merged(*m->engine, m->rows);
    const mmt::MergedRows& R = m->rows;
    if (!R.length.empty()) {
        std::memcpy(length, R.length.data(), R.length.size() * 4);
        std::memcpy(offsets, R.offsets.data(), R.offsets.size() * 8);
        std::memcpy(strands, R.strands.data(), R.strands.size());
    }
    if (thresh && !R.thresh.empty()) std::memcpy(thresh, R.thresh.data(), R.thresh.size() * 2);
    MMT_CATCH
}
int mmt_merged_device(const mmt_merged* m, const uint32_t** length, const int64_t** offsets, const uint8_t** strands,
                      const uint32_t** thresh) {
    if (!m) return fail(1, "null");
    if (length) *length = m->rows.d_length.get();
    if (offsets) *offsets = m->rows.d_offsets.get();
    if (strands) *strands = m->rows.d_strands.get();
    if (thresh) *thresh = m->rows.d_thresh.get();
    return 0;
}
// rows that were folded elsewhere (coordinate-range fold: every rank folds a slice of the anchor) as a merged result of
// this engine, so that the re-sort into direct-run order and the formatter apply
int mmt_merged_from_rows(mmt_engine* e, const uint32_t* length, const int64_t* offsets, const uint8_t* strands,
                         size_t n_rows, size_t n_docs, const uint16_t* thresh, size_t thresh_len, mmt_merged** out) {
    if (!e || !out || (n_rows && (!length || !offsets || !strands))) return fail(1, "engine, rows and out must be non-null");
    *out = nullptr;
    MMT_TRY
    std::unique_ptr<mmt_merged> m(new mmt_merged());
    mmt::MergedRows& R = m->rows;
    hipStream_t st = e->e->stream();
    MMT_HIP(hipSetDevice(e->e->device()));
    R.n_rows = n_rows; R.n_docs = n_docs; R.thresh_len = thresh_len;
    R.d_length.ensure(n_rows + 1); R.d_offsets.ensure(n_rows * n_docs + 1); R.d_strands.ensure(n_rows * n_docs + 1);
    R.d_thresh.ensure(thresh_len + 1);
    if (n_rows) {
        MMT_HIP(hipMemcpyAsync(R.d_length.get(), length, n_rows * 4, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemcpyAsync(R.d_offsets.get(), offsets, n_rows * n_docs * 8, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemcpyAsync(R.d_strands.get(), strands, n_rows * n_docs, hipMemcpyHostToDevice, st));
    }
    if (thresh_len && thresh) {
        mmt::DevBuf<uint16_t> narrow;
        narrow.ensure(thresh_len);
        MMT_HIP(hipMemcpyAsync(narrow.get(), thresh, thresh_len * 2, hipMemcpyHostToDevice, st));
        mmt::k::thresh_widen(narrow.get(), thresh_len, R.d_thresh.get(), st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    MMT_HIP(hipStreamSynchronize(st));
    R.on_host = false;
    m->engine = e->e.get();
    *out = m.release();
    MMT_CATCH
}
int mmt_merged_sort_like_direct(mmt_engine* e, mmt_merged* m) {
    if (!e || !m) return fail(1, "null");
    MMT_TRY
    mmt::sort_like_direct(*e->e, m->rows);
    m->text.clear(); m->text_valid = false;
    MMT_CATCH
}
// ---- multi-GPU exchange (dist.cpp) ---------------------------------------------------------
int mmt_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(1, "null");
    MMT_TRY
    mmt::comm_unique_id(id);
    MMT_CATCH
}
int mmt_comm_create(mmt_engine* e, int rank, int world, const uint8_t id[128], mmt_comm** out) {
    if (!e || !id || !out) return fail(1, "engine, id and out must be non-null");
    *out = nullptr;
    MMT_TRY
    std::unique_ptr<mmt_comm> c(new mmt_comm());
    c->c = mmt::comm_create(*e->e, rank, world, id);
    c->rank = rank;
    *out = c.release();
    MMT_CATCH
}
void mmt_comm_destroy(mmt_comm* c) { if (c) { mmt::comm_destroy(c->c); delete c; } }
int mmt_dist_merge(mmt_comm* c, mmt_engine* e, uint32_t min_len, mmt_merged** out) {
    if (!c || !e || !out) return fail(1, "comm, engine and out must be non-null");
    *out = nullptr;
    MMT_TRY
    bool root = false;
    mmt::MergedRows rows = mmt::dist_merge(*c->c, min_len, &root);
    if (root) {
        std::unique_ptr<mmt_merged> m(new mmt_merged());
        m->rows = std::move(rows);
        m->engine = e->e.get();
        *out = m.release();
    }
    MMT_CATCH
}
int mmt_dist_merge_ranges(mmt_comm* c, mmt_engine* e, uint32_t min_len, mmt_merged** out) {
    if (!c || !e || !out) return fail(1, "comm, engine and out must be non-null");
    *out = nullptr;
    MMT_TRY
    bool root = false;
    mmt::MergedRows rows = mmt::dist_merge_ranges(*c->c, min_len, &root);
    if (root) {
        std::unique_ptr<mmt_merged> m(new mmt_merged());
        m->rows = std::move(rows);
        m->engine = e->e.get();
        *out = m.release();
    }
    MMT_CATCH
}
int mmt_dist_gather_text(mmt_comm* c, const char** text, size_t* len) {
    if (!c || !text || !len) return fail(1, "null");
    MMT_TRY
    c->text = mmt::dist_gather_text(*c->c);
    *text = c->text.data(); *len = c->text.size();
    MMT_CATCH
}

int mmt_comm_selftest(mmt_comm* c, uint64_t elements, uint32_t width, uint64_t out[4]) {
    if (!c || !out) return fail(1, "null");
    MMT_TRY
    mmt::dist_selftest(*c->c, elements, width, out);
    MMT_CATCH
}

int mmt_comm_loopback(mmt_comm* c, uint64_t out[8]) {
    if (!c || !out) return fail(1, "null");
    MMT_TRY
    mmt::dist_loopback(*c->c, out);
    MMT_CATCH
}

const char* mmt_merged_text(mmt_merged* m, size_t* len) {
    if (!m) { if (len) *len = 0; return nullptr; }
    if (!m->text_valid) {
        try { m->text = mmt::format_merged(*m->engine, m->rows); m->text_valid = true; }
        catch (const std::exception& ex) { fail(2, ex.what()); if (len) *len = 0; return nullptr; }
    }
    if (len) *len = m->text.size();
    return m->text.data();
}
int mmt_merged_write_text(mmt_merged* m, const char* path) {
    if (!m || !path) return fail(1, "null");
    MMT_TRY
    if (m->text_valid) mmt::write_file_bytes(path, m->text.data(), m->text.size());
    else mmt::write_merged_text(*m->engine, m->rows, path);
    MMT_CATCH
}
void mmt_merged_free(mmt_merged* m) { delete m; }

}  // extern "C"
