// engine.cpp -- orchestration of the hot path on one GPU (no kernels here).
#include "engine.hpp"

#include <sys/stat.h>
#include "fasta.hpp"

#include <fcntl.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "prims.hpp"
#include "rows_kernels.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

Engine::Engine(int device, hipStream_t stream) : device_(device), stream_(stream) {
    lean_ = std::getenv("MUMEMTO_LEAN") != nullptr;      // tests: stage scratch is released between the stages
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        throw HipError("no usable HIP device (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) +
                       "); libmumemto has no CPU fallback");
    if (device < 0 || device >= count) throw HipError("device index out of range");
    MMT_HIP(hipSetDevice(device_));
    if (!stream_) { MMT_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)); own_stream_ = true; }
    for (auto& ev : ev_) ev.reset(new EventPair());
}

Engine::~Engine() {
    if (own_stream_) (void)hipStreamDestroy(stream_);
    if (sink_stream_) { (void)hipStreamDestroy(sink_stream_); for (auto ev : sink_copied_) if (ev) (void)hipEventDestroy(ev); }
}

void Engine::set_input_device(const uint8_t* d_bases, const uint64_t* doc_len, size_t n_docs) {
    MMT_HIP(hipSetDevice(device_));
    host_docs_.clear();
    supplier_ = nullptr; supplier_user_ = nullptr;
    d_bases_ = d_bases;
    preset_ = 0;
    input_valid_ = true;
    lcp_whole_ = false;
    doc_len_.assign(doc_len, doc_len + n_docs);
    doc_base_.assign(n_docs + 1, 0);
    for (size_t d = 0; d < n_docs; d++) doc_base_[d + 1] = doc_base_[d] + doc_len_[d];
}

void Engine::set_input_host(const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs) {
    MMT_HIP(hipSetDevice(device_));
    uint64_t total = 0;
    for (size_t d = 0; d < n_docs; d++) total += doc_len[d];
    d_bases_own_.ensure(total + 16);
    if (total) MMT_HIP(hipMemcpyAsync(d_bases_own_.get(), h_bases, total, hipMemcpyHostToDevice, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    set_input_device(d_bases_own_.get(), doc_len, n_docs);
}

uint8_t* Engine::begin_input_slots(size_t bytes) {
    MMT_HIP(hipSetDevice(device_));
    d_bases_own_.ensure(bytes + 64);
    return d_bases_own_.get();
}
void Engine::finish_input_slots(const std::vector<size_t>& slot, const uint64_t* doc_len, size_t n_docs) {
    d_bases_ = d_bases_own_.get();
    preset_ = 0; input_valid_ = true; lcp_whole_ = false;
    doc_len_.assign(doc_len, doc_len + n_docs);
    doc_base_.assign(n_docs + 1, 0);
    for (size_t d = 0; d <= n_docs; d++) doc_base_[d] = slot[d];        // (the text builder takes any offsets)
}

void Engine::set_input_host_docs(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs) {
    MMT_HIP(hipSetDevice(device_));
    uint64_t total = 0;
    for (size_t d = 0; d < n_docs; d++) total += doc_len[d];
    d_bases_own_.ensure(total + 16);
    uint64_t at = 0;
    for (size_t d = 0; d < n_docs; d++) {
        if (doc_len[d]) MMT_HIP(hipMemcpyAsync(d_bases_own_.get() + at, doc_ptr[d], doc_len[d], hipMemcpyHostToDevice, stream_));
        at += doc_len[d];
    }
    MMT_HIP(hipStreamSynchronize(stream_));
    set_input_device(d_bases_own_.get(), doc_len, n_docs);
}

void Engine::set_input_host_docs_deferred(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs) {
    MMT_HIP(hipSetDevice(device_));
    d_bases_own_.release();
    set_input_device(nullptr, doc_len, n_docs);
    host_docs_.assign(doc_ptr, doc_ptr + n_docs);
}

void Engine::set_input_supplier(DocSupplier fn, void* user, const uint64_t* doc_len, size_t n_docs) {
    if (!fn) throw std::runtime_error("set_input_supplier: no supplier");
    MMT_HIP(hipSetDevice(device_));
    d_bases_own_.release();
    set_input_device(nullptr, doc_len, n_docs);
    supplier_ = fn; supplier_user_ = user;
}

namespace {
// page-locked host memory for the documents a supplier writes
struct PinnedBytes {
    uint8_t* p = nullptr;
    explicit PinnedBytes(size_t n) { MMT_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), n ? n : 1, hipHostMallocDefault)); }
    ~PinnedBytes() { if (p) (void)hipHostFree(p); }
    PinnedBytes(const PinnedBytes&) = delete;
    PinnedBytes& operator=(const PinnedBytes&) = delete;
};
}  // namespace

// document layout of the text: starts of the documents, total length, device copy of the starts
void Engine::layout_docs(bool revcomp) {
    const size_t N = doc_len_.size();
    revcomp_ = revcomp;
    doc_start_.assign(N + 1, 0);
    for (size_t d = 0; d < N; d++) doc_start_[d + 1] = doc_start_[d] + (revcomp ? 2 : 1) * (doc_len_[d] + 1);
    n_ = doc_start_[N];
    if (n_ >= (1ull << 40))
        throw std::runtime_error("text of " + std::to_string(n_) + " characters exceeds 40-bit positions");
    // MMT_FORCE_WIDE: the 40-bit code path on small inputs (tests)
    wide_ = n_ >= NARROW_LIMIT || std::getenv("MMT_FORCE_WIDE") != nullptr;
    d_doc_start_.ensure(N + 1);
    MMT_HIP(hipMemcpyAsync(d_doc_start_.get(), doc_start_.data(), (N + 1) * 8, hipMemcpyHostToDevice, stream_));
}

void Engine::set_text_host(const uint8_t* text, uint64_t n, const uint64_t* doc_len, size_t n_docs, bool revcomp) {
    MMT_HIP(hipSetDevice(device_));
    d_bases_ = nullptr;
    doc_len_.assign(doc_len, doc_len + n_docs);
    doc_base_.assign(n_docs + 1, 0);
    layout_docs(revcomp);
    if (n != n_) throw std::runtime_error("the text has " + std::to_string(n) + " characters, the document lengths add "
                                          "up to " + std::to_string(n_));
    packed_ = false;                                    // (a handed-over text keeps its bytes)
    d_text_.ensure(TEXT_FRONT + n_ + TEXT_BACK);
    if (n_) MMT_HIP(hipMemcpyAsync(text_ptr(), text, n_, hipMemcpyHostToDevice, stream_));
    finish_text_padding();
    std::vector<uint64_t> hist(256, 0);
    for (uint64_t i = 0; i < n_; i++) hist[text[i]]++;
    d_hist_.ensure(256);
    MMT_HIP(hipMemcpyAsync(d_hist_.get(), hist.data(), 256 * 8, hipMemcpyHostToDevice, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    preset_ = 1;
    input_valid_ = true;
}

void Engine::set_stream_host(const uint32_t* sa, const uint32_t* lcp, const uint8_t* bwt, uint64_t entries,
                             const uint64_t* doc_len, size_t n_docs, bool revcomp) {
    set_stream_host40(sa, nullptr, lcp, bwt, entries, doc_len, n_docs, revcomp);
}

void Engine::set_stream_host40(const uint32_t* sa_lo, const uint8_t* sa_hi, const uint32_t* lcp, const uint8_t* bwt,
                               uint64_t entries, const uint64_t* doc_len, size_t n_docs, bool revcomp) {
    MMT_HIP(hipSetDevice(device_));
    d_bases_ = nullptr;
    doc_len_.assign(doc_len, doc_len + n_docs);
    doc_base_.assign(n_docs + 1, 0);
    layout_docs(revcomp);
    if (entries > n_) throw std::runtime_error("more stream entries than text characters");
    const uint64_t text_chars = n_;
    wide_ = sa_hi != nullptr;
    if (!wide_ && text_chars >= NARROW_LIMIT)
        throw std::runtime_error("a stream over " + std::to_string(text_chars) + " text characters needs 40-bit entries");
    for (uint64_t j = 0; j < entries; j++)
        if (((uint64_t)sa_lo[j] | (sa_hi ? (uint64_t)sa_hi[j] << 32 : 0)) >= text_chars)
            throw std::runtime_error("suffix array entry outside the text");
    n_ = entries;                                      // what the scan walks (a truncated stream is legal, see -a)
    d_sa_.ensure(n_ + 1); d_lcp_.ensure(n_ + 16); d_bwt_.ensure(n_ + 16);
    if (wide_) d_sa_hi_.ensure(n_ + 16);
    if (n_) {
        MMT_HIP(hipMemcpyAsync(d_sa_.get(), sa_lo, n_ * 4, hipMemcpyHostToDevice, stream_));
        if (wide_) MMT_HIP(hipMemcpyAsync(d_sa_hi_.get(), sa_hi, n_, hipMemcpyHostToDevice, stream_));
        MMT_HIP(hipMemcpyAsync(d_lcp_.get(), lcp, n_ * 4, hipMemcpyHostToDevice, stream_));
        MMT_HIP(hipMemcpyAsync(d_bwt_.get(), bwt, n_, hipMemcpyHostToDevice, stream_));
    }
    MMT_HIP(hipStreamSynchronize(stream_));
    lcp_whole_ = true;
    preset_ = 2;
    input_valid_ = true;
}

// ---- A1 ----------------------------------------------------------------------
TextRef Engine::text_ref() const {
    TextRef T;
    T.n = n_;
    if (packed_) { T.packed = d_packed_.get(); T.excw = d_excw_.get(); T.runs = d_runs_.get(); T.n_runs = (uint32_t)h_runs_.size(); }
    else T.v = text_ptr() ? text_ptr() - 1 : nullptr;
    return T;
}

// Two bits per character when asked for (MMT_PACKED_TEXT=1: the parity suite runs through it that way), or when the byte
// text would not leave room for the tables of the parse and one batch of the producer: SURVEY.md 8(e) row 2 -- every rank of
// BASELINE configs[4] holds all 573 G characters, 143 GB packed.  Only the bucket-wise producer reads a packed text, and the
// direct producer's inputs (bytes <= 0x02, four documents or fewer below 2^32 characters) keep the byte layout.
bool Engine::want_packed_text() const { return want_packed_text_of(n_); }
// (the same predicate says whether the raw bases of a host-fed run go through staging buffers: partitioned.cpp)
bool Engine::want_packed_text_of(uint64_t n, bool by_size_only) const {
    if (const char* c = std::getenv("MMT_PACKED_TEXT")) { if (!by_size_only) return std::atoi(c) != 0; }
    // A text below 2^32 characters is never packed for want of room: on a small or busy device (less than ~27 GB free) the
    // formula below packed every text, a few kilobases too, and sent it to the bucket-wise producer with 256 MB of event
    // buffers and row discard -- such a device runs small collections normally or fails for what really does not fit.
    if (n < NARROW_LIMIT) return false;
    // (1 byte per character + 0.5 for the raw bases while the text is made + ~1.4 at the parse's peak + a batch)
    const double avail = 0.95 * (double)pool::available(device_);
    if ((double)n * 2.9 + 24.0 * 1073741824.0 > avail) return true;
    // whole-genome shares (from ~48 G characters on the modulus of the parse grows with the text too): the bucket-wise producer
    // with expansion takes them anyway, and what the packed text leaves free becomes batches -- {anchor + 11} haplotypes kept
    // their 73 GB of bytes, had 287 M representatives a batch and 52 passes over the text where the packed {anchor + 12} had 14
    return n >= 48000000000ull;
}

// exception runs: events -> sorted run list + one flag per block of 4096 positions (textref.hpp)
void Engine::finish_packed_text(DevBuf<uint64_t>& ev_start, DevBuf<uint64_t>& ev_end, DevBuf<uint32_t>& ev_count, uint32_t ev_cap) {
    uint32_t cnt[2] = {0, 0};
    MMT_HIP(hipMemcpyAsync(cnt, ev_count.get(), 8, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    if (cnt[0] > ev_cap || cnt[1] > ev_cap)
        throw std::runtime_error("the text holds more than " + std::to_string(ev_cap) + " runs of characters other than A C G T: "
                                 "too many for the packed layout (MMT_PACKED_TEXT=0 keeps one byte per character)");
    if (cnt[0] != cnt[1]) throw std::runtime_error("packed text: run starts and ends do not pair up");
    std::vector<uint64_t> hs, he;
    d2h(hs, ev_start.get(), cnt[0], stream_);
    d2h(he, ev_end.get(), cnt[1], stream_);
    std::sort(hs.begin(), hs.end());
    std::sort(he.begin(), he.end());
    h_runs_.resize(cnt[0]);
    const uint64_t blocks = (n_ >> TX_BLOCK_SHIFT) + 2;
    std::vector<uint64_t> flags((blocks + 63) / 64 + 1, 0);
    for (uint32_t i = 0; i < cnt[0]; i++) {
        const uint64_t a = hs[i] >> 8, b = he[i];
        if (b <= a || (i && a < (hs[i - 1] >> 8))) throw std::runtime_error("packed text: inconsistent exception runs");
        if (b - a > 0xffffffffull) throw std::runtime_error("packed text: a run of one character of 2^32 positions or more");
        h_runs_[i] = ExcRun{a, (uint32_t)(b - a), (uint32_t)(hs[i] & 0xff)};
        for (uint64_t k = a >> TX_BLOCK_SHIFT; k <= (b - 1) >> TX_BLOCK_SHIFT; k++) flags[k >> 6] |= 1ull << (k & 63);
    }
    d_excw_.ensure(flags.size());
    d_runs_.ensure(h_runs_.size() + 1);
    MMT_HIP(hipMemcpyAsync(d_excw_.get(), flags.data(), flags.size() * 8, hipMemcpyHostToDevice, stream_));
    if (!h_runs_.empty())
        MMT_HIP(hipMemcpyAsync(d_runs_.get(), h_runs_.data(), h_runs_.size() * sizeof(ExcRun), hipMemcpyHostToDevice, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
}

void Engine::build_text(bool revcomp) {
    const size_t N = doc_len_.size();
    layout_docs(revcomp);
    d_doc_base_.ensure(N + 1); d_doc_len_.ensure(N + 1);
    MMT_HIP(hipMemcpyAsync(d_doc_base_.get(), doc_base_.data(), (N + 1) * 8, hipMemcpyHostToDevice, stream_));
    MMT_HIP(hipMemcpyAsync(d_doc_len_.get(), doc_len_.data(), N * 8, hipMemcpyHostToDevice, stream_));
    d_hist_.ensure(256);
    MMT_HIP(hipMemsetAsync(d_hist_.get(), 0, 256 * 8, stream_));
    packed_ = want_packed_text();
    const bool deferred = (!host_docs_.empty() || supplier_) && d_bases_ == nullptr;
    uint64_t longest = 0;
    for (size_t d = 0; d < N; d++) longest = std::max(longest, doc_len_[d]);
    auto supply = [&](size_t d, uint8_t* dst) {
        if (doc_len_[d] && supplier_(supplier_user_, d, dst, doc_len_[d]) != 0)
            throw std::runtime_error("the supplier of the documents failed at document " + std::to_string(d));
    };
    auto upload_deferred = [&]() {                     // the byte layout reads all raw bases from the device
        uint64_t total = 0;
        for (size_t d = 0; d < N; d++) total += doc_len_[d];
        d_bases_own_.ensure(total + 16);
        uint64_t at = 0;
        std::unique_ptr<PinnedBytes> host[2];
        if (supplier_) { host[0].reset(new PinnedBytes(longest)); host[1].reset(new PinnedBytes(longest)); }
        for (size_t d = 0; d < N; d++) {
            const uint8_t* src = supplier_ ? host[d & 1]->p : host_docs_[d];
            if (supplier_) {
                if (d >= 2) MMT_HIP(hipStreamSynchronize(stream_));          // (the copy of document d - 2 has left the buffer)
                supply(d, host[d & 1]->p);
            }
            if (doc_len_[d]) MMT_HIP(hipMemcpyAsync(d_bases_own_.get() + at, src, doc_len_[d], hipMemcpyHostToDevice, stream_));
            at += doc_len_[d];
        }
        MMT_HIP(hipStreamSynchronize(stream_));
        d_bases_ = d_bases_own_.get();
    };
    if (packed_) {
        d_text_.release();
        const size_t words = (size_t)((n_ + 31) / 32) + 8;
        d_packed_.ensure(words);
        MMT_HIP(hipMemsetAsync(d_packed_.get(), 0, words * 8, stream_));
        const uint32_t ev_cap = 1u << 24;
        DevBuf<uint64_t> ev_start, ev_end;
        DevBuf<uint32_t> ev_count;
        ev_start.ensure(ev_cap); ev_end.ensure(ev_cap); ev_count.ensure(2);
        MMT_HIP(hipMemsetAsync(ev_count.get(), 0, 8, stream_));
        if (deferred) {
            // document by document through two staging buffers: the copy of document d + 1 runs beside the packing of d
            DevBuf<uint8_t> stage[2];
            stage[0].ensure(longest + 64); stage[1].ensure(longest + 64);
            std::unique_ptr<PinnedBytes> host[2];      // supplied documents: written here while the device packs the one before
            if (supplier_) { host[0].reset(new PinnedBytes(longest)); host[1].reset(new PinnedBytes(longest)); }
            hipStream_t cs = nullptr;
            MMT_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipEvent_t copied[2], packed_ev[2];
            for (int k = 0; k < 2; k++) { MMT_HIP(hipEventCreateWithFlags(&copied[k], hipEventDisableTiming)); MMT_HIP(hipEventCreateWithFlags(&packed_ev[k], hipEventDisableTiming)); }
            try {
                for (size_t d = 0; d < N; d++) {
                    const int b = (int)(d & 1);
                    const uint8_t* src = supplier_ ? host[b]->p : host_docs_[d];
                    if (supplier_) {
                        if (d >= 2) MMT_HIP(hipEventSynchronize(copied[b]));            // the copy of document d - 2 has left host[b]
                        supply(d, host[b]->p);
                    }
                    if (d >= 2) MMT_HIP(hipStreamWaitEvent(cs, packed_ev[b], 0));          // the buffer is free again
                    if (doc_len_[d]) MMT_HIP(hipMemcpyAsync(stage[b].get(), src, doc_len_[d], hipMemcpyHostToDevice, cs));
                    MMT_HIP(hipEventRecord(copied[b], cs));
                    MMT_HIP(hipStreamWaitEvent(stream_, copied[b], 0));
                    // (raw + doc_base[d] = the staging buffer)
                    k::pack_text(stage[b].get() - doc_base_[d], d_doc_base_.get(), d_doc_len_.get(), d_doc_start_.get(), (uint32_t)N,
                                 d_packed_.get(), n_, d_hist_.get(), ev_start.get(), ev_end.get(), ev_count.get(), ev_cap,
                                 doc_start_[d], doc_start_[d + 1], stream_);
                    MMT_HIP(hipEventRecord(packed_ev[b], stream_));
                }
                MMT_HIP(hipStreamSynchronize(stream_));
            } catch (...) {
                (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs);
                for (int k = 0; k < 2; k++) { (void)hipEventDestroy(copied[k]); (void)hipEventDestroy(packed_ev[k]); }
                throw;
            }
            (void)hipStreamDestroy(cs);
            for (int k = 0; k < 2; k++) { (void)hipEventDestroy(copied[k]); (void)hipEventDestroy(packed_ev[k]); }
        } else
            k::pack_text(d_bases_, d_doc_base_.get(), d_doc_len_.get(), d_doc_start_.get(), (uint32_t)N, d_packed_.get(), n_,
                         d_hist_.get(), ev_start.get(), ev_end.get(), ev_count.get(), ev_cap, 0, n_, stream_);
        finish_packed_text(ev_start, ev_end, ev_count, ev_cap);
        // bytes the parse reserves (<= 0x02) are the direct producer's, which reads one byte per character
        std::vector<uint64_t> hist;
        d2h(hist, d_hist_.get(), 256, stream_);
        if (!(hist[0] || hist[1] || hist[2])) return;
        if (n_ >= NARROW_LIMIT) throw std::runtime_error("texts of 2^32 characters or more must not contain the bytes 0x00-0x02 "
                                                         "(reserved by the prefix-free parse)");
        packed_ = false;
        MMT_HIP(hipMemsetAsync(d_hist_.get(), 0, 256 * 8, stream_));
    }
    if (deferred) upload_deferred();
    d_packed_.release(); d_excw_.release(); d_runs_.release(); h_runs_.clear();
    d_text_.ensure(TEXT_FRONT + n_ + TEXT_BACK);
    k::build_text(d_bases_, d_doc_base_.get(), d_doc_len_.get(), d_doc_start_.get(), (uint32_t)N, revcomp, text_ptr(), n_,
                  d_hist_.get(), stream_);
    finish_text_padding();          // (the last work-item of the kernel stores up to 15 bytes past the text)
}

// Dollar in front of the text, 32 Dollars and then zeros behind it (see text_ptr())
void Engine::finish_text_padding() {
    MMT_HIP(hipMemsetAsync(d_text_.get(), 0, TEXT_FRONT - 1, stream_));
    MMT_HIP(hipMemsetAsync(d_text_.get() + TEXT_FRONT - 1, 2, 1, stream_));
    MMT_HIP(hipMemsetAsync(text_ptr() + n_, 2, 32, stream_));
    MMT_HIP(hipMemsetAsync(text_ptr() + n_ + 32, 0, TEXT_BACK - 32, stream_));
}

// ---- A8: suffix array by prefix doubling (narrow texts only) --------------------
void Engine::suffix_sort() {
    if (n_ >= NARROW_LIMIT)
        throw std::runtime_error("the direct suffix sort handles texts below 2^32 - 4096 characters; larger inputs go "
                                 "through prefix-free parsing, which reserves the bytes 0x00-0x02");
    wide_ = false;                                   // (MMT_FORCE_WIDE does not apply to this producer)
    const uint32_t n = (uint32_t)n_;
    // symbol codes: dense ranks of the bytes that occur, 0 reserved for "past the end"
    std::vector<uint64_t> hist;
    d2h(hist, d_hist_.get(), 256, stream_);
    uint8_t code[256];
    int sigma = 0;
    for (int c = 0; c < 256; c++) code[c] = hist[c] ? (uint8_t)(++sigma) : 0;
    int bits = std::max(1, bit_width_u64((uint64_t)sigma));
    int chars = std::min(64 / bits, 64);
    d_code_.ensure(256);
    MMT_HIP(hipMemcpyAsync(d_code_.get(), code, 256, hipMemcpyHostToDevice, stream_));

    d_sa_.ensure(n); d_rank_.ensure(n);
    sorter_.reserve(n);
    k::pack_keys(text_ptr(), n, d_code_.get(), bits, chars, k::PACK_NO_SEP, sorter_.keys_in(), sorter_.vals_in(), stream_);
    sort_rounds_ = sorter_.sort(n, bits * chars, (uint64_t)chars, d_sa_.get(), d_rank_.get(), d_temp_, stream_);
}

// BWT (direct producer) and the PLCP column: d_plcp_a_[i] = LCP of the suffix at text position i with its
// predecessor in suffix-array order.  The LCP column itself is gathered per scan range (Engine::scan).
void Engine::lcp_bwt() {
    const uint64_t n = n_;
    d_bwt_.ensure(n + 16);
    // The LCP column follows from the irreducible suffixes alone (kernels.hip, "LCP column WITHOUT the inverse
    // suffix array"): no 4-byte random store per suffix anywhere.  The PFP emitter wrote SA and BWT and keeps no
    // inverse suffix array at all; only the suffix ranks of the anchor document are recorded here (multi-GPU
    // re-sort).  The direct producer has the full array from its sort.
    const bool pfp = producer_used_ >= 2 && pfp_->bwt_ready;
    if (!pfp) k::bwt_from_sa(text_ptr(), (uint32_t)n, d_sa_.get(), d_bwt_.get(), stream_);
    const uint64_t anchor = std::min<uint64_t>(doc_len_[0], n);
    void* rank_out = nullptr;
    if (pfp) {
        if (wide_) { d_rank64_.ensure((size_t)anchor + 1); rank_out = d_rank64_.get(); }
        else { d_rank_.ensure((size_t)anchor + 1); rank_out = d_rank_.get(); }
    }
    if (lcp_col_ready_) {                              // the producer wrote the LCP column itself (pfp.cpp, guided.cpp)
        // (the suffix ranks of the anchor order merged rows like a direct run: only runs that record merge metadata feed a merge)
        if (want_anchor_ranks_) k::anchor_ranks(sa_col(), 0, n, anchor, rank_out, stream_);
        anchor_ranks_valid_ = want_anchor_ranks_;
        lcp_whole_ = false;
        return;
    }
    d_plcp_a_.ensure(n); d_count_.ensure(8);
    const size_t rec = k::long_lcp_record_bytes(wide_) + 4;     // one record + one index for the second tier
    uint64_t cap64 = std::max<uint64_t>(d_long_.size() / rec, n / 256 + 4096);
    if (const char* c = std::getenv("MMT_LONG_CAP")) cap64 = (uint64_t)std::max(1, std::atoi(c));   // tests: force the rerun
    uint32_t cap = (uint32_t)std::min<uint64_t>(cap64, 0x7fffffffull);
    for (int attempt = 0; attempt < 2; attempt++) {
        d_long_.ensure((size_t)cap * rec);
        k::irreducible_lcp(text_ptr(), n, sa_col(), d_bwt_.get(), d_plcp_a_.get(), rank_out, anchor, d_long_.get(),
                           d_count_.get() + 2, cap, stream_);
        uint32_t found = 0;
        MMT_HIP(hipMemcpyAsync(&found, d_count_.get() + 2, 4, hipMemcpyDeviceToHost, stream_));
        MMT_HIP(hipStreamSynchronize(stream_));
        if (std::getenv("MMT_LCP_STATS")) std::fprintf(stderr, "[lcp] %u matches beyond 192 characters (list capacity %u)\n", found, cap);
        if (found <= cap) {
            k::long_lcp(text_ptr(), n, wide_, d_long_.get(), found, d_plcp_a_.get(),
                        reinterpret_cast<uint32_t*>(d_long_.get() + (size_t)cap * (rec - 4)), d_count_.get() + 3, stream_);
            break;
        }
        if (attempt) throw std::runtime_error("long-match list overflow in the LCP construction");
        cap = found + 1024;                            // rare: rerun with the exact size
    }
    d_temp_.ensure(k::plcp_running_max_scratch(n));
    k::plcp_running_max(d_plcp_a_.get(), n, d_temp_.get(), stream_);      // PLCP everywhere, in place
    lcp_whole_ = false;
    anchor_ranks_valid_ = true;                        // (written by k_irr_lcp, or the direct producer's inverse suffix array)
}

// One-shot / tight-memory runs: the suffix-sort stage (doubling scratch, dictionary, parse, emitter tables: two thirds
// of all device bytes) is dead once SA and BWT exist.  Counters and stage times of the parse survive for the statistics.
void Engine::release_sort_scratch() {
    MMT_HIP(hipStreamSynchronize(stream_));
    sorter_.release();
    d_temp_.release();                               // (library scratch of the sorts: as large as their inputs)
    std::unique_ptr<PfpState> fresh(new PfpState());
    const PfpState& S = *pfp_;
    fresh->w = S.w; fresh->p = S.p; fresh->n_cuts = S.n_cuts; fresh->n_phrases = S.n_phrases; fresh->n_distinct = S.n_distinct;
    fresh->dict_len = S.dict_len; fresh->n_groups = S.n_groups; fresh->rounds_dict = S.rounds_dict; fresh->run_refined = S.run_refined;
    fresh->rounds_parse = S.rounds_parse; fresh->n_entries = S.n_entries; fresh->n_fallback = S.n_fallback;
    fresh->emit_launches = S.emit_launches;
    fresh->bwt_ready = S.bwt_ready;
    // (what the bucket-wise producer's share was: Engine::kmer_in_share answers after the run)
    fresh->guided = S.guided; fresh->expand = S.expand; fresh->g_prefix = S.g_prefix; fresh->g_bits = S.g_bits;
    fresh->g_share_lo = S.g_share_lo; fresh->g_share_hi = S.g_share_hi; fresh->g_share_valid = S.g_share_valid;
    std::memcpy(fresh->g_code, S.g_code, sizeof(S.g_code));
    std::memcpy(fresh->ms, S.ms, sizeof(S.ms));
    pfp_ = std::move(fresh);
}

void Engine::release_columns(bool keep_anchor_ranks) {
    MMT_HIP(hipStreamSynchronize(stream_));
    release_sort_scratch();
    d_text_.release(); d_bwt_.release(); d_sa_.release(); d_sa_hi_.release(); d_cols_.release();
    d_packed_.release(); d_excw_.release(); d_runs_.release();
    if (!keep_anchor_ranks) { d_rank_.release(); d_rank64_.release(); anchor_ranks_valid_ = false; }
    d_lcp_.release(); d_plcp_a_.release(); d_long_.release(); d_wpre_.release(); d_wsuf_.release(); d_wide_.release();
    d_cand_.release(); d_flags_.release();
    for (int k = 0; k < 2; k++) { w_sa_[k].release(); w_hi_[k].release(); w_bwt_[k].release(); w_lcp_[k].release(); }
    d_pool_lo_.release(); d_pool_hi_.release(); d_rows_pool_.release(); d_cap_cnt_.release(); d_cap_off_.release();
    lcp_whole_ = false; lcp_col_ready_ = false; columns_kept_ = false;
}

// Called after the suffix sort: the LCP stage is about to allocate the PLCP and LCP columns (8 bytes per text
// character, plus the candidate list).  When the device does not have that much left, the sort stage's scratch goes.
bool Engine::wants_lean() const {
    const size_t have = (d_plcp_a_.size() + d_lcp_.size()) * sizeof(uint32_t);
    const double need = 9.0 * (double)n_ - (double)have;
    return need > 0.95 * (double)pool::available(device_);
}

// ---- A5 ------------------------------------------------------------------------
// keeps the first `used` elements when the buffer has to grow
template <typename T>
static void grow_keep(DevBuf<T>& buf, size_t need, size_t used, hipStream_t s) {
    if (need <= buf.size()) return;
    DevBuf<T> bigger;
    bigger.ensure(need + need / 4);
    if (used) MMT_HIP(hipMemcpyAsync(bigger.get(), buf.get(), used * sizeof(T), hipMemcpyDeviceToDevice, s));
    MMT_HIP(hipStreamSynchronize(s));
    buf.swap(bigger);
}

// ---- the scan, window by window ---------------------------------------------------------------------------------
// The stream is scanned in windows of suffix-array positions.  A window carries a left extension long enough for every
// walk; candidates carry window-relative positions, accepted rows absolute ones.  The producers that emit the columns
// themselves (pfp.cpp, guided.cpp) hand over one window at a time and drop it (pfp_lcp_mum.hpp:197: the reference
// never stores the stream); columns that exist as a whole (the direct producer, a handed-over stream) are cut into
// windows here.
EventPair& Engine::next_range_event(ScanState& S, int kind) {      // kind: 0 LCP gather, 1 scan kernel, 2 verification, 3 production
    if (S.ev_at == range_ev_.size()) range_ev_.emplace_back(new EventPair());
    range_ev_kind_.push_back(kind);
    range_ev_[S.ev_at]->reset();
    return *range_ev_[S.ev_at++];
}

void Engine::scan_begin(const mmt_params& p, ScanState& S) {
    const size_t N = doc_len_.size();
    hipStream_t st = stream_;
    d_count_.ensure(8);
    num_distinct_eff_ = p.num_distinct ? p.num_distinct : N;     // mumemto_api.cpp:344-346
    // interval size cap: explicit total cap, else docs * per-doc cap (every accepted interval obeys it)
    uint64_t cap = 0;
    if (p.max_total_freq > 0) cap = (uint64_t)p.max_total_freq;
    if (p.max_doc_freq > 0) {
        uint64_t c2 = (uint64_t)p.max_doc_freq * N;
        cap = cap ? std::min(cap, c2) : c2;
    }
    if (cap > 0xfffffff0ull) cap = 0;
    if (N > 32768 && !(p.max_doc_freq == 1 && N <= 64))
        throw std::runtime_error("more than 32768 documents are not supported by the candidate verifier");
    S = ScanState();
    S.cap = cap;
    S.a.min_len = p.min_match_len;
    S.a.num_distinct = (uint32_t)std::min<uint64_t>(num_distinct_eff_, 0xffffffffu);
    S.a.cap = (uint32_t)cap;
    S.a.emit_all = p.merge_metadata ? 1 : 0;
    S.a.d_count = d_count_.get();
    S.merge = p.merge_metadata != 0;
    S.max_doc_freq = p.max_doc_freq > 0 ? (uint32_t)std::min<int64_t>(p.max_doc_freq, 0x7fffffff) : 0u;
    // thresholds (merge metadata) and the row list are shared by all windows
    thresh_len_ = 0;
    if (p.merge_metadata && N > 0) {
        thresh_len_ = 2 * (doc_len_[0] + 1);
        d_thresh_.ensure(thresh_len_);
        MMT_HIP(hipMemsetAsync(d_thresh_.get(), 0, thresh_len_ * 4, st));
    }
    MMT_HIP(hipMemsetAsync(d_count_.get() + 1, 0, 4, st));
    n_cand_ = 0;
    pool_used_ = 0;
    stream_entries_ = 0; window_bytes_peak_ = 0;
    // left extension of every window but the first: at least the largest interval (+1), the window of the wide-document
    // path, one LDS halo; uncapped modes start with 64 K entries and repeat a window whose walk ran off it
    const uint64_t ALIGN_R = 4096;
    uint64_t ext0 = cap ? cap + 1 : 65536;
    ext0 = std::max<uint64_t>(ext0, (uint64_t)S.a.num_distinct + 2);
    ext0 = std::max<uint64_t>(ext0, 1040);
    S.ext0 = (ext0 + ALIGN_R - 1) / ALIGN_R * ALIGN_R;
    scan_ranges_ = 0;
    range_ev_kind_.clear();
}

// this rank's share of the closing positions (set_scan_shard): [n k / count, n (k + 1) / count), cut at multiples of 4096
void Engine::shard_range(uint64_t& lo, uint64_t& hi) const {
    const uint64_t n = n_, ALIGN_R = 4096;
    lo = 0; hi = n;
    if (shard_count_ <= 1) return;
    auto cut = [&](uint64_t k) { return k >= shard_count_ ? n : (n / shard_count_ * k) / ALIGN_R * ALIGN_R; };
    lo = cut(shard_index_); hi = cut(shard_index_ + 1);
}

// One window: scan kernel, verification, thresholds; rows of a transient window take their suffix-array entries along.
// Returns false when a walk ran off the left edge of a window that does not start the stream (the caller repeats it with
// a longer extension); nothing of that attempt is kept.
bool Engine::scan_window(ScanState& S, const ColWindow& w, const mmt_params& p) {
    hipStream_t st = stream_;
    const size_t N = doc_len_.size();
    k::ScanArgs& a = S.a;
    a.lcp = w.lcp; a.bwt = w.bwt; a.n = w.len; a.first = w.first; a.more_left = w.more_left ? 1 : 0;
    size_t capacity = std::max<size_t>(1u << 20, p.merge_metadata ? w.len / 6 : w.len / 32);
    capacity = std::max(capacity, d_cand_.size());
    uint32_t found = 0, overflow = 0;
    EventPair& es = next_range_event(S, 1);
    for (int attempt = 0; attempt < 3; attempt++) {
        d_cand_.ensure(capacity);
        a.out = d_cand_.get(); a.capacity = (uint32_t)std::min<size_t>(capacity, 0xffffffffu);
        MMT_HIP(hipMemsetAsync(d_count_.get(), 0, 4, st));
        MMT_HIP(hipMemsetAsync(d_count_.get() + 4, 0, 4, st));
        es.start(st);
        if (k::scan_needs_wide(a)) {               // window tables in HBM, per window
            d_wpre_.ensure(w.len); d_wsuf_.ensure(w.len); d_wide_.ensure(w.len);
            k::scan_wide_prepare(a.lcp, a.bwt, a.n, a.num_distinct, d_wpre_.get(), d_wsuf_.get(), d_wide_.get(), st);
            prims::inclusive_max_u32(d_temp_, d_wide_.get(), d_wide_.get(), w.len, st);
            a.wide_pre = d_wpre_.get(); a.wide_suf = d_wsuf_.get(); a.wide_chg = d_wide_.get();
        }
        k::scan_intervals(a, st);
        es.stop(st);
        uint32_t back[5] = {0, 0, 0, 0, 0};
        MMT_HIP(hipMemcpyAsync(back, d_count_.get(), 20, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        found = back[0]; overflow = back[4];
        if (found <= capacity) break;
        if (attempt == 2) throw std::runtime_error("candidate list overflow in the scan");
        capacity = (size_t)found + 1024;       // rare: re-run with the exact size
    }
    if (overflow && w.more_left) return false;
    scan_ranges_++;
    n_cand_ += found;
    // verification + thresholds of this window's candidates
    EventPair& ev = next_range_event(S, 2);
    ev.start(st);
    grow_keep(d_rows_, std::max<size_t>(S.rows_used + found, 1), S.rows_used, st);
    k::VerifyArgs v;
    v.cand = d_cand_.get(); v.n_cand = found; v.sa = w.sa; v.base = w.base; v.sa_off = w.sa_off; v.lcp = w.lcp;
    v.d_doc_start = d_doc_start_.get(); v.n_docs = (uint32_t)N;
    v.num_distinct = a.num_distinct;
    v.max_doc_freq = S.max_doc_freq;
    v.merge = S.merge ? 1 : 0; v.thresh = d_thresh_.get();
    v.rows = d_rows_.get(); v.d_row_count = d_count_.get() + 1;
    k::verify_candidates(v, st);
    uint32_t r = 0;
    MMT_HIP(hipMemcpyAsync(&r, d_count_.get() + 1, 4, hipMemcpyDeviceToHost, st));
    MMT_HIP(hipStreamSynchronize(st));
    const size_t fresh = (size_t)r - S.rows_used;
    if (w.transient && fresh) {
        // the accepted intervals' suffix-array entries leave the window with their rows
        d_cap_cnt_.ensure(fresh + 1); d_cap_off_.ensure(fresh + 1);
        k::row_counts(d_rows_.get() + S.rows_used, (uint32_t)fresh, d_cap_cnt_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_cap_cnt_.get(), d_cap_off_.get(), fresh, st);
        uint64_t last[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(&last[0], d_cap_off_.get() + (fresh - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&last[1], d_cap_cnt_.get() + (fresh - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        const uint64_t total = last[0] + last[1];
        grow_keep(d_pool_lo_, (size_t)(pool_used_ + total + 16), (size_t)pool_used_, st);
        if (wide_) grow_keep(d_pool_hi_, (size_t)(pool_used_ + total + 16), (size_t)pool_used_, st);
        grow_keep(d_rows_pool_, std::max<size_t>(r, 1), S.rows_used, st);
        SaCol pool; pool.lo = d_pool_lo_.get(); pool.hi = wide_ ? d_pool_hi_.get() : nullptr;
        SaCol win = w.sa; win.lo += w.sa_off; if (win.hi) win.hi += w.sa_off;
        k::capture_rows(d_rows_.get() + S.rows_used, (uint32_t)fresh, d_cap_off_.get(), pool_used_, win, w.base, pool,
                        d_rows_pool_.get() + S.rows_used, st);
        pool_used_ += total;
        if (!tap_kmers_.empty()) tap_window(d_rows_pool_.get() + S.rows_used, (uint32_t)fresh, pool);
    } else if (fresh && !tap_kmers_.empty()) tap_window(d_rows_.get() + S.rows_used, (uint32_t)fresh, sa_col());
    ev.stop(st);
    S.rows_used = r;
    return true;
}

// ---- the row tap -----------------------------------------------------------------------------------------------------------
static void pack_kmers(const uint8_t* kmers, size_t n, size_t k, std::vector<uint64_t>& out) {
    if (k < 1 || k > 16) throw std::runtime_error("k-mers of 1 .. 16 characters");
    out.assign(2 * n, 0);
    for (size_t i = 0; i < n; i++)
        for (size_t c = 0; c < k; c++) out[2 * i + c / 8] |= (uint64_t)kmers[i * k + c] << (8 * (c % 8));
}
void Engine::set_row_tap(const uint8_t* kmers, size_t n, size_t k, size_t max_rows, size_t max_occ) {
    tap_kmers_.clear(); tap_k_ = 0; tap_cap_rows_ = tap_cap_occ_ = 0;
    d_tap_off_.release(); d_tap_sa_.release(); d_tap_len_.release(); d_tap_cnt_.release();
    if (!n) return;
    pack_kmers(kmers, n, k, tap_kmers_);
    tap_k_ = (uint32_t)k; tap_cap_rows_ = std::max<size_t>(max_rows, 1); tap_cap_occ_ = std::max<size_t>(max_occ, 1);
}
void Engine::tap_window(const k::Row* rows, uint32_t n_rows, SaCol pool) {
    hipStream_t st = stream_;
    if (!d_tap_len_.get()) {                    // first window of the run (Engine::run cleared the buffers)
        d_tap_kmers_.ensure(tap_kmers_.size()); d_tap_used_.ensure(2);
        d_tap_len_.ensure(tap_cap_rows_); d_tap_cnt_.ensure(tap_cap_rows_); d_tap_off_.ensure(tap_cap_rows_); d_tap_sa_.ensure(tap_cap_occ_);
        MMT_HIP(hipMemcpyAsync(d_tap_kmers_.get(), tap_kmers_.data(), tap_kmers_.size() * 8, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemsetAsync(d_tap_used_.get(), 0, 16, st));
    }
    rk::tap_rows(rows, n_rows, pool, text_ref(), d_tap_kmers_.get(), (uint32_t)(tap_kmers_.size() / 2), tap_k_, d_tap_len_.get(),
                 d_tap_off_.get(), d_tap_cnt_.get(), d_tap_sa_.get(), d_tap_used_.get(), tap_cap_rows_, tap_cap_occ_, st);
}
void Engine::row_tap_counts(uint64_t out[2]) {
    out[0] = out[1] = 0;
    if (!d_tap_len_.get()) return;
    MMT_HIP(hipSetDevice(device_));
    MMT_HIP(hipMemcpyAsync(out, d_tap_used_.get(), 16, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
}
void Engine::row_tap_get(uint32_t* length, uint64_t* occ_start, uint64_t* sa) {
    uint64_t used[2];
    row_tap_counts(used);
    if (used[0] > tap_cap_rows_ || used[1] > tap_cap_occ_)
        throw std::runtime_error("the row tap overflowed: " + std::to_string(used[0]) + " rows, " + std::to_string(used[1]) + " entries");
    occ_start[0] = 0;
    if (!used[0]) return;
    std::vector<uint64_t> off(used[0]), all(used[1]);
    std::vector<uint32_t> cnt(used[0]), len(used[0]);
    MMT_HIP(hipMemcpyAsync(len.data(), d_tap_len_.get(), used[0] * 4, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipMemcpyAsync(cnt.data(), d_tap_cnt_.get(), used[0] * 4, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipMemcpyAsync(off.data(), d_tap_off_.get(), used[0] * 8, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipMemcpyAsync(all.data(), d_tap_sa_.get(), used[1] * 8, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    uint64_t at = 0;                             // (slots were handed out by atomics: compact in slot order)
    for (uint64_t r = 0; r < used[0]; r++) {
        length[r] = len[r];
        std::memcpy(sa + at, all.data() + off[r], (size_t)cnt[r] * 8);
        at += cnt[r];
        occ_start[r + 1] = at;
    }
}
uint64_t Engine::kmer_positions(const uint8_t* kmers, size_t n, size_t k, uint64_t* pos, uint32_t* which, uint64_t cap) {
    if (!have_text()) throw std::runtime_error("no text on the device");
    MMT_HIP(hipSetDevice(device_));
    std::vector<uint64_t> packed;
    pack_kmers(kmers, n, k, packed);
    DevBuf<uint64_t> d_k, d_pos, d_used;
    DevBuf<uint32_t> d_which;
    d_k.ensure(packed.size()); d_pos.ensure(std::max<uint64_t>(cap, 1)); d_which.ensure(std::max<uint64_t>(cap, 1)); d_used.ensure(2);
    MMT_HIP(hipMemcpyAsync(d_k.get(), packed.data(), packed.size() * 8, hipMemcpyHostToDevice, stream_));
    MMT_HIP(hipMemsetAsync(d_used.get(), 0, 16, stream_));
    rk::kmer_positions(text_ref(), d_k.get(), (uint32_t)n, (uint32_t)k, d_pos.get(), d_which.get(), d_used.get(), cap, stream_);
    uint64_t found = 0;
    MMT_HIP(hipMemcpyAsync(&found, d_used.get(), 8, hipMemcpyDeviceToHost, stream_));
    MMT_HIP(hipStreamSynchronize(stream_));
    const uint64_t got = std::min(found, cap);
    std::vector<uint64_t> hp(got);
    std::vector<uint32_t> hw(got);
    if (got) {
        MMT_HIP(hipMemcpyAsync(hp.data(), d_pos.get(), got * 8, hipMemcpyDeviceToHost, stream_));
        MMT_HIP(hipMemcpyAsync(hw.data(), d_which.get(), got * 4, hipMemcpyDeviceToHost, stream_));
        MMT_HIP(hipStreamSynchronize(stream_));
    }
    std::vector<uint64_t> order(got);
    for (uint64_t i = 0; i < got; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return hp[a] < hp[b]; });
    for (uint64_t i = 0; i < got; i++) { pos[i] = hp[order[i]]; which[i] = hw[order[i]]; }
    return found;
}

void Engine::scan_end(ScanState& S) {
    if (scan_ranges_ == 0) scan_ranges_ = 1;
    range_ev_.resize(S.ev_at);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = 0; i < S.ev_at; i++) acc[range_ev_kind_[i]] += range_ev_[i]->ms();
    for (int i = 0; i < 4; i++) scan_ms_[i] = acc[i];
}

// Columns that exist as a whole (the direct producer; a stream that was handed over): cut into windows here.  The LCP
// column is gathered from PLCP through the suffix array window by window (whole at once for a text below 2^32).
void Engine::scan(const mmt_params& p) {
    const uint64_t n = n_;
    hipStream_t st = stream_;
    ScanState S;
    scan_begin(p, S);
    streamed_ = false;
    // window size: everything at once unless the text is wide (MMT_SCAN_RANGE: tests)
    const uint64_t ALIGN_R = 4096;
    // (a handed-over stream brings its whole LCP column: windows of it are pointer offsets)
    const bool whole_lcp = lcp_whole_ && preset_ == 2;
    uint64_t range = n;
    if (wide_ || n >= 0xffffe000ull) range = 1ull << 28;
    if (const char* c = std::getenv("MMT_SCAN_RANGE")) range = std::max<uint64_t>(1, std::strtoull(c, nullptr, 10));
    range = (range + ALIGN_R - 1) / ALIGN_R * ALIGN_R;
    uint64_t shard_lo = 0, shard_hi = n;
    if (shard_count_ > 1) {
        if (preset_ == 2) throw std::runtime_error("a handed-over stream cannot be scanned in shards");
        shard_range(shard_lo, shard_hi);
        range = std::min<uint64_t>(range, std::max<uint64_t>(ALIGN_R, (shard_hi - shard_lo + ALIGN_R - 1) / ALIGN_R * ALIGN_R));
    }
    const bool single = range >= n && shard_count_ == 1;
    for (uint64_t c0 = shard_lo; c0 < shard_hi; c0 += range) {
        const uint64_t c1 = std::min(shard_hi, c0 + range);
        uint64_t ext = c0 ? S.ext0 : 0;
        for (;;) {
            if (ext > c0) ext = c0;
            const uint64_t b0 = c0 - ext;
            const uint64_t len = c1 - b0;
            if (len >= 0xffffe000ull) throw std::runtime_error("scan range with its left extension exceeds 2^32 entries");
            EventPair& eg = next_range_event(S, 0);
            eg.start(st);
            const uint32_t* lcp_ptr = nullptr;
            if (lcp_col_ready_) lcp_ptr = d_plcp_a_.get() + b0;          // the column exists in suffix-array order
            else if (whole_lcp) lcp_ptr = d_lcp_.get() + b0;
            else {
                if (!(lcp_whole_ && single)) {
                    d_lcp_.ensure(len + 16);
                    k::lcp_gather(d_plcp_a_.get(), sa_col(), b0, len, d_lcp_.get(), st);
                    lcp_whole_ = single;
                }
                lcp_ptr = d_lcp_.get();
            }
            eg.stop(st);
            ColWindow w;
            w.sa = sa_col(); w.sa_off = b0; w.base = b0; w.bwt = d_bwt_.get() + b0; w.lcp = lcp_ptr;
            w.len = (uint32_t)len; w.first = (uint32_t)ext; w.more_left = b0 > 0; w.transient = false;
            if (scan_window(S, w, p)) break;
            ext = std::max<uint64_t>(ext * 4, S.ext0);                   // a walk ran off the extension
        }
        if (single) break;
    }
    scan_end(S);
}

// ---- the window buffers of the producers that emit the columns themselves ----------------------------------------------
void Engine::window_reserve(int set, uint64_t entries) {
    w_sa_[set].ensure(entries + 64); w_bwt_[set].ensure(entries + 64); w_lcp_[set].ensure(entries + 64);
    if (wide_) w_hi_[set].ensure(entries + 64);
    uint64_t bytes = 0;
    for (int k = 0; k < 2; k++) bytes += w_sa_[k].bytes() + w_bwt_[k].bytes() + w_lcp_[k].bytes() + w_hi_[k].bytes();
    window_bytes_peak_ = std::max(window_bytes_peak_, bytes);
}
ColWindow Engine::window_view(int set, uint64_t base, uint32_t len, uint32_t first) const {
    ColWindow w;
    w.sa.lo = w_sa_[set].get(); w.sa.hi = wide_ ? w_hi_[set].get() : nullptr;
    w.bwt = w_bwt_[set].get(); w.lcp = w_lcp_[set].get();
    w.base = base; w.sa_off = 0; w.len = len; w.first = first; w.more_left = base > 0; w.transient = true;
    return w;
}
// keep mode (set_keep_columns): the closing positions of a window also go into whole columns
void Engine::keep_window(const ColWindow& w) {
    if (!columns_kept_) return;
    const uint64_t at = w.base + w.first, cnt = (uint64_t)w.len - w.first;
    if (!cnt) return;
    hipStream_t st = stream_;
    MMT_HIP(hipMemcpyAsync(d_sa_.get() + at, w.sa.lo + w.sa_off + w.first, cnt * 4, hipMemcpyDeviceToDevice, st));
    if (wide_) MMT_HIP(hipMemcpyAsync(d_sa_hi_.get() + at, w.sa.hi + w.sa_off + w.first, cnt, hipMemcpyDeviceToDevice, st));
    MMT_HIP(hipMemcpyAsync(d_bwt_.get() + at, w.bwt + w.first, cnt, hipMemcpyDeviceToDevice, st));
    MMT_HIP(hipMemcpyAsync(d_plcp_a_.get() + at, w.lcp + w.first, cnt * 4, hipMemcpyDeviceToDevice, st));
}

// ---- A6: rows -> coordinates -> text ---------------------------------------------
void append_uint(std::string& s, uint64_t v) {
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) s.push_back(tmp[--k]);
}

// pop order of the reference's stack for `cnt` rows (closing position ascending, longer first): d_order_[k] = index of the
// k-th row
void Engine::order_rows(const k::Row* rows_abs, uint32_t cnt) {
    hipStream_t st = stream_;
    d_rkeys_a_.ensure(cnt); d_rkeys_b_.ensure(cnt); d_rvals_a_.ensure(cnt); d_order_.ensure(cnt);
    if (!wide_) {
        rk::row_keys(rows_abs, cnt, d_rkeys_a_.get(), d_rvals_a_.get(), st);
        prims::sort_pairs_u64_u32(d_temp_, d_rkeys_a_.get(), d_rkeys_b_.get(), d_rvals_a_.get(), d_order_.get(), cnt, 0, 64, st);
    } else {
        // closing positions beyond 32 bits: two stable sorts (by descending length, then by closing position)
        d_rvals_b_.ensure(cnt);
        uint32_t* len_keys = reinterpret_cast<uint32_t*>(d_rkeys_a_.get());
        uint32_t* len_keys_out = len_keys + cnt;
        rk::row_len_keys(rows_abs, cnt, len_keys, d_rvals_a_.get(), st);
        prims::sort_pairs_u32_u32(d_temp_, len_keys, len_keys_out, d_rvals_a_.get(), d_rvals_b_.get(), cnt, 0, 32, st);
        rk::row_end_keys(rows_abs, d_rvals_b_.get(), cnt, d_rkeys_a_.get(), st);
        prims::sort_pairs_u64_u32(d_temp_, d_rkeys_a_.get(), d_rkeys_b_.get(), d_rvals_b_.get(), d_order_.get(), cnt, 0, 40, st);
    }
}

// ---- the text sink: PREFIX.mums written while the run goes on ---------------------------------------------------------
void Engine::sink_open(bool mum_mode) {
    sink_active_ = false;
    sink_written_path_.clear();
    sink_mum_ = mum_mode;
    sink_total_rows_ = 0;
    if (sink_path_.empty() || std::getenv("MUMEMTO_NO_TEXT_SINK")) return;
    // Rows that have been written need not stay: with a sink, a run whose accepted rows would not fit the device next to
    // its text keeps nothing of a window once its bytes are on their way (BASELINE configs[4]: a rank's rows carry ~94
    // occurrences each -- tens of GB of suffix-array entries, offsets and text).  Such a run answers only for the file and
    // the number of rows.  MMT_SINK_DISCARD=0 / 1 overrides (tests).
    sink_discard_ = sink_force_discard_ ||
                    (!sink_keep_rows_ &&
                     (std::getenv("MMT_SINK_DISCARD") ? std::atoi(std::getenv("MMT_SINK_DISCARD")) != 0 : (packed_ || n_ >= (1ull << 37))));
    if (!mum_mode && !sink_discard_) return;          // (a MEM run that keeps its rows writes its file at the end, as before)
    // the bytes go to PREFIX.mums.tmp and take the final name when the run has succeeded (sink_close): a run that fails
    // after some windows -- out of memory, a consistency check at the end -- must not leave a plausible partial PREFIX.mums
    // ("/dev/null": the bytes are formatted, copied out, digested and dropped -- a full-size test run whose 66 GB of rows the
    // box has no room for)
    // (any path that exists and is not a regular file -- a FIFO, /dev/stdout -- is written in place as well: the same rule as
    // merge.cpp write_merged_text)
    {
        struct stat sb;
        sink_null_ = sink_path_ == "/dev/null" || (::stat(sink_path_.c_str(), &sb) == 0 && !S_ISREG(sb.st_mode));
    }
    sink_tmp_path_ = sink_null_ ? sink_path_ : sink_path_ + ".tmp";
    sink_digest_ = StreamDigest(); sink_written_ = 0; sink_digest_value_ = 0;
    sink_want_digest_ = std::getenv("MMT_SINK_DIGEST") != nullptr;
    sink_fd_ = ::open(sink_tmp_path_.c_str(), sink_null_ ? O_WRONLY : (O_CREAT | O_TRUNC | O_WRONLY), 0644);
    if (sink_fd_ < 0) throw std::runtime_error("cannot write " + sink_tmp_path_);
    sink_rows_done_ = 0; sink_bytes_ = 0; sink_block_at_ = 0; sink_block_used_ = 0;
    sink_block_pending_.assign(sink_blocks_.size(), 0);
    sink_closing_ = false; sink_error_.clear();
    sink_pieces_ = 0;
    if (!sink_stream_) {
        MMT_HIP(hipStreamCreateWithFlags(&sink_stream_, hipStreamNonBlocking));
        for (auto& ev : sink_copied_) MMT_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    sink_active_ = true;
    const int device = device_;
    sink_thread_ = std::thread([this, device]() {
        (void)hipSetDevice(device);
        for (;;) {
            SinkPiece pc;
            {
                std::unique_lock<std::mutex> lk(sink_mu_);
                sink_cv_.wait(lk, [&] { return !sink_q_.empty() || sink_closing_; });
                if (sink_q_.empty()) break;
                pc = sink_q_.front(); sink_q_.pop_front();
            }
            // (sink_error_ is read by the run's thread in sink_host_room / sink_close: both sides under the mutex)
            auto failed = [&]() { std::lock_guard<std::mutex> lk(sink_mu_); return !sink_error_.empty(); };
            auto fail_with = [&](const std::string& what) { std::lock_guard<std::mutex> lk(sink_mu_); if (sink_error_.empty()) sink_error_ = what; };
            if (hipEventSynchronize(pc.ready) != hipSuccess) fail_with("device copy of the output failed");
            (void)hipEventDestroy(pc.ready);
            for (size_t done = 0; !failed() && done < pc.n;) {
                const ssize_t w = ::write(sink_fd_, pc.p + done, pc.n - done);
                if (w <= 0) { fail_with("short write to " + sink_tmp_path_); break; }
                done += (size_t)w;
            }
            // a running digest of the bytes in file order (this one thread writes the pieces in order; whole words are carried
            // across piece boundaries, which fall where windows end and differ from box to box): what two runs of a file
            // nobody can keep are compared by
            if (sink_null_ || sink_want_digest_) sink_digest_.update(pc.p, pc.n);
            { std::lock_guard<std::mutex> lk(sink_mu_); sink_written_ += pc.n; }
            { std::lock_guard<std::mutex> lk(sink_mu_); sink_block_pending_[pc.block]--; }
            sink_cv_.notify_all();
        }
    });
}
// Page-locked room for a piece: a RING of blocks of 256 MB (or the piece) that stay with the engine.  A block is used
// again once the writer thread has written every piece in it, so the page-locked memory of a run is a few blocks however
// large the output is (round 3 kept every piece until the end of the run: 4.5 GB of pinned memory for a rank's share of
// whole genomes, and it stayed with the engine).
char* Engine::sink_host_room(size_t n, uint32_t* block) {
    // (64 MB: page-locking costs 0.19 s per GB when the block is made and 0.13 s per GB when the process ends -- the four
    // blocks of 256 MB were a quarter of a second of the bench workload's 2.2 s from process start to exit)
    static const size_t BLOCK = (size_t)(std::getenv("MMT_SINK_BLOCK_MB") ? std::max(1, std::atoi(std::getenv("MMT_SINK_BLOCK_MB"))) : 64) << 20;
    const size_t RING = 4;
    auto fits = [&](size_t b) { return std::max(BLOCK, sink_block_cap_[b]) >= n; };
    std::unique_lock<std::mutex> lk(sink_mu_);
    if (!sink_blocks_.empty() && sink_block_at_ < sink_blocks_.size() && fits(sink_block_at_) &&
        sink_block_used_ + n <= std::max(BLOCK, sink_block_cap_[sink_block_at_])) {
        char* p = sink_blocks_[sink_block_at_]->get() + sink_block_used_;
        sink_block_used_ += n; sink_block_pending_[sink_block_at_]++; *block = (uint32_t)sink_block_at_;
        return p;
    }
    // the next block of the ring: a new one while the ring is short, otherwise the oldest, once it has been written
    size_t next;
    if (sink_blocks_.size() < RING) {
        next = sink_blocks_.size();
        sink_blocks_.emplace_back(new PinnedBuf<char>());
        sink_block_cap_.push_back(0);
        sink_block_pending_.push_back(0);
    } else {
        next = (sink_block_at_ + 1) % sink_blocks_.size();
        sink_cv_.wait(lk, [&] { return sink_block_pending_[next] == 0 || !sink_error_.empty(); });
    }
    if (sink_block_cap_[next] < std::max(BLOCK, n)) {
        lk.unlock();
        sink_blocks_[next]->ensure(std::max(BLOCK, n));      // (nobody reads an idle block)
        lk.lock();
        sink_block_cap_[next] = std::max(BLOCK, n);
    }
    sink_block_at_ = next; sink_block_used_ = n; sink_block_pending_[next]++; *block = (uint32_t)next;
    return sink_blocks_[next]->get();
}
// the rows accepted since the last call, in pop order, as PREFIX.mums bytes -> helper thread
void Engine::sink_flush(ScanState& S) {
    if (!sink_active_) return;
    const size_t r0 = sink_rows_done_, r1 = S.rows_used;
    if (r1 <= r0) return;
    const uint32_t cnt = (uint32_t)(r1 - r0);
    const size_t N = doc_len_.size();
    hipStream_t st = stream_;
    order_rows(d_rows_.get() + r0, cnt);
    d_doc_len_.ensure(N + 1);
    MMT_HIP(hipMemcpyAsync(d_doc_len_.get(), doc_len_.data(), N * 8, hipMemcpyHostToDevice, st));
    rk::RowArgs a;
    a.rows = d_rows_pool_.get() + r0; a.order = d_order_.get(); a.n_rows = cnt;
    a.sa.lo = d_pool_lo_.get(); a.sa.hi = wide_ ? d_pool_hi_.get() : nullptr;
    a.doc_start = d_doc_start_.get(); a.doc_len = d_doc_len_.get(); a.n_docs = (uint32_t)N; a.revcomp = revcomp_ ? 1 : 0;
    d_tlen_.ensure(cnt); d_tlen64_.ensure(cnt); d_toff_.ensure(cnt); d_keep_.ensure(cnt);
    size_t kept = 0, tbytes = 0, occ = 0;
    if (sink_mum_) {
        const size_t slots = (size_t)cnt * N;
        d_slot_off_.ensure(slots); d_slot_st_.ensure(slots); d_ridx_.ensure(cnt);
        MMT_HIP(hipMemsetAsync(d_slot_off_.get(), 0xFF, slots * 8, st));     // -1 = document absent
        MMT_HIP(hipMemsetAsync(d_slot_st_.get(), 0, slots, st));
        rk::mum_measure(a, d_slot_off_.get(), d_slot_st_.get(), d_keep_.get(), d_tlen_.get(), st);
        prims::exclusive_sum_u32(d_temp_, d_keep_.get(), d_ridx_.get(), cnt, st);
        rk::widen(d_tlen_.get(), cnt, d_tlen64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_tlen64_.get(), d_toff_.get(), cnt, st);
        uint32_t k0 = 0, k1 = 0; uint64_t t0 = 0, t1 = 0;
        MMT_HIP(hipMemcpyAsync(&k0, d_ridx_.get() + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&k1, d_keep_.get() + (cnt - 1), 4, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&t0, d_toff_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&t1, d_tlen64_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        kept = (size_t)k0 + k1; tbytes = (size_t)(t0 + t1);
    } else {
        // PREFIX.mems rows (write_mem, mem_finder.hpp:210-263): every accepted row is written, occurrences in suffix-array order
        d_wpos_.ensure(cnt); d_wdoc_.ensure(cnt); d_occ64_.ensure(cnt); d_ooff_.ensure(cnt);
        rk::mem_measure(a, d_keep_.get() /* occurrences per row */, d_tlen_.get(), d_wpos_.get(), d_wdoc_.get(), st);
        rk::widen(d_keep_.get(), cnt, d_occ64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_occ64_.get(), d_ooff_.get(), cnt, st);
        rk::widen(d_tlen_.get(), cnt, d_tlen64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_tlen64_.get(), d_toff_.get(), cnt, st);
        uint64_t o0 = 0, o1 = 0, t0 = 0, t1 = 0;
        MMT_HIP(hipMemcpyAsync(&o0, d_ooff_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&o1, d_occ64_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&t0, d_toff_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipMemcpyAsync(&t1, d_tlen64_.get() + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        kept = cnt; occ = (size_t)(o0 + o1); tbytes = (size_t)(t0 + t1);
    }
    sink_rows_done_ = r1;
    sink_total_rows_ += kept;
    auto discard = [&]() {
        // the rows and their suffix-array entries are dead once their bytes are formatted: the next window starts at slot 0
        if (!sink_discard_) return;
        MMT_HIP(hipMemsetAsync(d_count_.get() + 1, 0, 4, st));
        S.rows_used = 0; sink_rows_done_ = 0; pool_used_ = 0;
    };
    if (!tbytes) { discard(); return; }
    // the piece is formatted on the run's stream and copied out on the copy stream (two device pieces in turn: the
    // formatting of a piece waits for the copy of the piece two flushes ago)
    const uint32_t slot_i = sink_pieces_ & 1u;
    DevBuf<char>& piece = d_piece_[slot_i];
    if (sink_pieces_ >= 2) MMT_HIP(hipStreamWaitEvent(st, sink_copied_[slot_i], 0));
    if (piece.size() < tbytes + 1) { MMT_HIP(hipStreamSynchronize(sink_stream_)); piece.ensure(tbytes + tbytes / 4 + 1); }
    if (sink_mum_) {
        d_olen_.ensure(kept + 1); d_ooffs_.ensure(kept * N + 1); d_ost_.ensure(kept * N + 1);
        rk::mum_write(a, d_slot_off_.get(), d_slot_st_.get(), d_keep_.get(), d_ridx_.get(), d_toff_.get(), d_olen_.get(),
                      d_ooffs_.get(), d_ost_.get(), piece.get(), st);
    } else {
        d_olen_.ensure(cnt + 1); d_ooffs_.ensure(occ + 1); d_omdoc_.ensure(occ + 1); d_ost_.ensure(occ + 1);
        rk::mem_write(a, d_ooff_.get(), d_toff_.get(), d_wpos_.get(), d_wdoc_.get(), d_olen_.get(), d_ooffs_.get(),
                      d_omdoc_.get(), d_ost_.get(), piece.get(), st);
    }
    discard();
    SinkPiece pc;
    pc.n = tbytes;
    char* h = sink_host_room(tbytes, &pc.block);
    pc.p = h;
    hipEvent_t formatted;
    MMT_HIP(hipEventCreateWithFlags(&formatted, hipEventDisableTiming));
    MMT_HIP(hipEventRecord(formatted, st));
    MMT_HIP(hipStreamWaitEvent(sink_stream_, formatted, 0));
    MMT_HIP(hipMemcpyAsync(h, piece.get(), tbytes, hipMemcpyDeviceToHost, sink_stream_));
    MMT_HIP(hipEventRecord(sink_copied_[slot_i], sink_stream_));
    MMT_HIP(hipEventCreateWithFlags(&pc.ready, hipEventDisableTiming));
    MMT_HIP(hipEventRecord(pc.ready, sink_stream_));
    (void)hipEventDestroy(formatted);
    sink_pieces_++;
    { std::lock_guard<std::mutex> lk(sink_mu_); sink_q_.push_back(pc); }
    sink_cv_.notify_one();
    sink_bytes_ += tbytes;
}
void Engine::sink_close(bool ok) {
    if (!sink_active_) return;
    { std::lock_guard<std::mutex> lk(sink_mu_); sink_closing_ = true; }
    sink_cv_.notify_one();
    if (sink_thread_.joinable()) sink_thread_.join();
    (void)hipStreamSynchronize(sink_stream_);
    std::string error;
    { std::lock_guard<std::mutex> lk(sink_mu_); error = sink_error_; }
    if (sink_fd_ >= 0 && ::close(sink_fd_) != 0 && error.empty()) error = "cannot close " + sink_tmp_path_;
    sink_fd_ = -1;
    sink_active_ = false;
    if (error.empty() && !ok) error = "the run failed";
    if (error.empty() && !sink_null_ && std::rename(sink_tmp_path_.c_str(), sink_path_.c_str()) != 0) error = "cannot rename " + sink_tmp_path_;
    if (!error.empty()) { if (!sink_null_) ::unlink(sink_tmp_path_.c_str()); throw std::runtime_error(error); }
    sink_digest_value_ = sink_digest_.final();
    sink_written_path_ = sink_path_;
    sink_discarded_ = sink_discard_;
}

void Engine::make_rows(const mmt_params& p) {
    // Rows stay on the device: sort into pop order, measure, place, write (rows_kernels.hip), then
    // one D2H of the library arrays and of the .mums / .mems bytes into page-locked host memory.
    const size_t N = doc_len_.size();
    hipStream_t st = stream_;
    ev_[5]->start(st);
    HostRows& R = rows_;
    R = HostRows();
    R.mum_mode = p.max_doc_freq == 1;                 // mem_finder.hpp:85
    R.n_docs = N;
    bumbl_.clear();
    if (sink_discarded_ && !sink_written_path_.empty()) {
        // the rows left with the windows that accepted them (sink_flush): the file and the count are what this run answers for
        R.n_rows = sink_total_rows_;
        h_occ_start_.ensure(2); h_occ_start_.get()[0] = 0; R.occ_start = h_occ_start_.get();
        rows_pending_ = 0;
        ev_[5]->stop(st);
        return;
    }
    const uint32_t n_rows = [&] {
        uint32_t v = 0;
        MMT_HIP(hipMemcpyAsync(&v, d_count_.get() + 1, 4, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        return v;
    }();
    h_occ_start_.ensure(2);
    h_occ_start_.get()[0] = 0;
    R.occ_start = h_occ_start_.get();
    if (n_rows == 0) { ev_[5]->stop(st); return; }

    // pop order of the reference's stack: closing position ascending, longer first
    order_rows(d_rows_.get(), n_rows);
    d_doc_len_.ensure(N + 1);
    MMT_HIP(hipMemcpyAsync(d_doc_len_.get(), doc_len_.data(), N * 8, hipMemcpyHostToDevice, st));
    rk::RowArgs a;
    a.rows = d_rows_.get(); a.order = d_order_.get(); a.n_rows = n_rows; a.sa = sa_col();
    if (streamed_) {            // the columns are gone: the rows index the pool of their own suffix-array entries
        a.rows = d_rows_pool_.get();
        a.sa.lo = d_pool_lo_.get(); a.sa.hi = wide_ ? d_pool_hi_.get() : nullptr;
    }
    a.doc_start = d_doc_start_.get(); a.doc_len = d_doc_len_.get(); a.n_docs = (uint32_t)N; a.revcomp = revcomp_ ? 1 : 0;
    d_tlen_.ensure(n_rows); d_tlen64_.ensure(n_rows); d_toff_.ensure(n_rows);
    auto last_u32 = [&](const uint32_t* d) {
        uint32_t v = 0; MMT_HIP(hipMemcpyAsync(&v, d + (n_rows - 1), 4, hipMemcpyDeviceToHost, st)); return v; };
    auto last_u64 = [&](const uint64_t* d) {
        uint64_t v = 0; MMT_HIP(hipMemcpyAsync(&v, d + (n_rows - 1), 8, hipMemcpyDeviceToHost, st)); return v; };

    if (R.mum_mode) {
        const size_t slots = (size_t)n_rows * N;
        d_slot_off_.ensure(slots); d_slot_st_.ensure(slots); d_keep_.ensure(n_rows); d_ridx_.ensure(n_rows);
        MMT_HIP(hipMemsetAsync(d_slot_off_.get(), 0xFF, slots * 8, st));     // -1 = document absent
        MMT_HIP(hipMemsetAsync(d_slot_st_.get(), 0, slots, st));
        rk::mum_measure(a, d_slot_off_.get(), d_slot_st_.get(), d_keep_.get(), d_tlen_.get(), st);
        prims::exclusive_sum_u32(d_temp_, d_keep_.get(), d_ridx_.get(), n_rows, st);
        rk::widen(d_tlen_.get(), n_rows, d_tlen64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_tlen64_.get(), d_toff_.get(), n_rows, st);
        const uint32_t k0 = last_u32(d_ridx_.get()), k1 = last_u32(d_keep_.get());
        const uint64_t t0 = last_u64(d_toff_.get()), t1 = last_u64(d_tlen64_.get());
        MMT_HIP(hipStreamSynchronize(st));
        const size_t kept = (size_t)k0 + k1, tbytes = (size_t)(t0 + t1);
        d_olen_.ensure(kept + 1); d_ooffs_.ensure(kept * N + 1); d_ost_.ensure(kept * N + 1); d_otext_.ensure(tbytes + 1);
        rk::mum_write(a, d_slot_off_.get(), d_slot_st_.get(), d_keep_.get(), d_ridx_.get(), d_toff_.get(),
                      d_olen_.get(), d_ooffs_.get(), d_ost_.get(), d_otext_.get(), st);
        // the tables and the bytes stay in HBM; fetch_rows() copies what a caller asks for (library arrays, file bytes)
        R.n_rows = kept; R.text_len = tbytes;
        rows_pending_ = kept ? (ROWS_ARRAYS | ROWS_TEXT) : 0;
    } else {
        d_keep_.ensure(n_rows); d_wpos_.ensure(n_rows); d_wdoc_.ensure(n_rows); d_occ64_.ensure(n_rows);
        d_ooff_.ensure(n_rows);
        rk::mem_measure(a, d_keep_.get() /* occurrences per row */, d_tlen_.get(), d_wpos_.get(), d_wdoc_.get(), st);
        rk::widen(d_keep_.get(), n_rows, d_occ64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_occ64_.get(), d_ooff_.get(), n_rows, st);
        rk::widen(d_tlen_.get(), n_rows, d_tlen64_.get(), st);
        prims::exclusive_sum_u64(d_temp_, d_tlen64_.get(), d_toff_.get(), n_rows, st);
        const uint64_t o0 = last_u64(d_ooff_.get()), o1 = last_u64(d_occ64_.get());
        const uint64_t t0 = last_u64(d_toff_.get()), t1 = last_u64(d_tlen64_.get());
        MMT_HIP(hipStreamSynchronize(st));
        const size_t occ = (size_t)(o0 + o1), tbytes = (size_t)(t0 + t1);
        d_olen_.ensure(n_rows); d_ooffs_.ensure(occ + 1); d_omdoc_.ensure(occ + 1); d_ost_.ensure(occ + 1);
        d_otext_.ensure(tbytes + 1);
        rk::mem_write(a, d_ooff_.get(), d_toff_.get(), d_wpos_.get(), d_wdoc_.get(), d_olen_.get(), d_ooffs_.get(),
                      d_omdoc_.get(), d_ost_.get(), d_otext_.get(), st);
        R.n_rows = n_rows; R.n_occ = occ; R.text_len = tbytes;
        rows_pending_ = ROWS_ARRAYS | ROWS_TEXT;
    }
    ev_[5]->stop(st);
}

// D2H of the last run's rows into page-locked host memory, on demand: the library arrays (ROWS_ARRAYS) and / or the
// bytes of PREFIX.mums / PREFIX.mems (ROWS_TEXT).
void Engine::write_text_file(const std::string& path) {
    if (!sink_written_path_.empty() && sink_written_path_ == path) return;      // written while the run went on (set_text_sink)
    if (!(rows_pending_ & ROWS_TEXT) || merged_thresh_valid_) {      // already on the host (or a merged result: staged)
        const HostRows& R = rows(ROWS_TEXT);
        write_file_bytes(path, R.text, R.text_len);
        return;
    }
    MMT_HIP(hipSetDevice(device_));
    HostRows& R = rows_;
    h_text_.ensure(R.text_len + 1);
    const size_t PIECE = (size_t)64 << 20;
    const size_t pieces = (R.text_len + PIECE - 1) / PIECE;
    std::vector<hipEvent_t> ev(pieces);
    for (size_t k = 0; k < pieces; k++) {
        const size_t at = k * PIECE, len = std::min(PIECE, R.text_len - at);
        MMT_HIP(hipMemcpyAsync(h_text_.get() + at, d_otext_.get() + at, len, hipMemcpyDeviceToHost, stream_));
        MMT_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        MMT_HIP(hipEventRecord(ev[k], stream_));
    }
    const int fd = ::open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    std::string error;
    if (fd < 0) error = "cannot write " + path;
    for (size_t k = 0; k < pieces; k++) {
        MMT_HIP(hipEventSynchronize(ev[k]));
        (void)hipEventDestroy(ev[k]);
        const size_t at = k * PIECE, len = std::min(PIECE, R.text_len - at);
        for (size_t done = 0; error.empty() && done < len;) {
            const ssize_t w = ::write(fd, h_text_.get() + at + done, len - done);
            if (w <= 0) { error = "short write to " + path; break; }
            done += (size_t)w;
        }
    }
    if (fd >= 0 && ::close(fd) != 0 && error.empty()) error = "cannot close " + path;
    R.text = h_text_.get();
    rows_pending_ &= ~ROWS_TEXT;
    if (!error.empty()) throw std::runtime_error(error);
}

void Engine::fetch_rows(int need) {
    // the rows of a run that wrote them window by window and dropped them (set_text_sink over a text that fills the device,
    // or the pieces of a sharded run) are in the file and nowhere else: say so instead of handing out null arrays
    if (need && sink_discarded_ && !sink_written_path_.empty())
        throw std::runtime_error("the rows of this run left the device window by window: they are in " + sink_written_path_ +
                                 " (row arrays, the bytes, .bumbl and .thresh need a run that keeps them: no text sink, or MMT_SINK_DISCARD=0)");
    need &= rows_pending_;
    if (!need) return;
    MMT_HIP(hipSetDevice(device_));
    if (merged_thresh_valid_) {                 // a partitioned run: the merged tables (partitioned.cpp)
        (void)merged_thresh();
        rows_.length = merged_.length.data(); rows_.mum_offsets = merged_.offsets.data(); rows_.mum_strands = merged_.strands.data();
        rows_pending_ &= ~need;
        return;
    }
    hipStream_t st = stream_;
    HostRows& R = rows_;
    const size_t N = R.n_docs, nr = R.n_rows;
    if (need & ROWS_TEXT) {
        h_text_.ensure(R.text_len + 1);
        if (R.text_len) MMT_HIP(hipMemcpyAsync(h_text_.get(), d_otext_.get(), R.text_len, hipMemcpyDeviceToHost, st));
        R.text = h_text_.get();
    }
    if (need & ROWS_ARRAYS) {
        if (R.mum_mode) {
            h_len_.ensure(nr + 1); h_offs_.ensure(nr * N + 1); h_st_.ensure(nr * N + 1);
            if (nr) {
                MMT_HIP(hipMemcpyAsync(h_len_.get(), d_olen_.get(), nr * 4, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipMemcpyAsync(h_offs_.get(), d_ooffs_.get(), nr * N * 8, hipMemcpyDeviceToHost, st));
                MMT_HIP(hipMemcpyAsync(h_st_.get(), d_ost_.get(), nr * N, hipMemcpyDeviceToHost, st));
            }
            R.length = h_len_.get(); R.mum_offsets = h_offs_.get(); R.mum_strands = h_st_.get();
        } else {
            const size_t occ = R.n_occ;
            h_len_.ensure(nr + 1); h_offs_.ensure(occ + 1); h_mdoc_.ensure(occ + 1); h_st_.ensure(occ + 1);
            h_occ_start_.ensure(nr + 2);
            MMT_HIP(hipMemcpyAsync(h_len_.get(), d_olen_.get(), nr * 4, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(h_occ_start_.get(), d_ooff_.get(), nr * 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(h_offs_.get(), d_ooffs_.get(), occ * 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(h_mdoc_.get(), d_omdoc_.get(), occ * 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(h_st_.get(), d_ost_.get(), occ, hipMemcpyDeviceToHost, st));
            R.length = h_len_.get(); R.occ_start = h_occ_start_.get();
            R.mem_offsets = h_offs_.get(); R.mem_docs = h_mdoc_.get(); R.mem_strands = h_st_.get();
        }
    }
    MMT_HIP(hipStreamSynchronize(st));
    if ((need & ROWS_ARRAYS) && !R.mum_mode) h_occ_start_.get()[nr] = R.n_occ;
    rows_pending_ &= ~need;
}

const std::string& Engine::bumbl() {
    // mem_finder.hpp:451-503: u16 flags | u64 n_seqs | u64 n_mums | u32 len[] | i64 starts | strand bits
    if (!bumbl_.empty() || !rows_.mum_mode) return bumbl_;
    fetch_rows(ROWS_ARRAYS);
    const HostRows& R = rows_;
    const uint64_t nm = R.n_rows, ns = R.n_docs, nbits = nm * ns;
    uint16_t flags = (uint16_t)(1u << 15);
    if (num_distinct_eff_ < ns) flags |= (uint16_t)(1u << 13);
    bumbl_.assign(2 + 16 + 4 * nm + 8 * nbits + (nbits + 7) / 8, '\0');
    char* p = &bumbl_[0];
    std::memcpy(p, &flags, 2); p += 2;
    std::memcpy(p, &ns, 8); p += 8;
    std::memcpy(p, &nm, 8); p += 8;
    if (nm) { std::memcpy(p, R.length, 4 * nm); p += 4 * nm; std::memcpy(p, R.mum_offsets, 8 * nbits); p += 8 * nbits; }
    for (uint64_t i = 0; i < nbits; i++)
        if (R.mum_strands[i]) p[i / 8] |= (char)(1u << (7 - (i % 8)));
    return bumbl_;
}

void Engine::thresh_files(std::vector<uint16_t>& fwd, std::vector<uint16_t>& rev) {
    // thresholds re-indexed by position inside each written MUM, MUMs in anchor order, 0-terminated
    fetch_rows(ROWS_ARRAYS);
    const HostRows& R = rows_;
    if (!R.mum_mode || !thresh_len_) throw std::runtime_error("thresholds need a multi-MUM run with merge metadata");
    std::vector<uint16_t> th(thresh_len_);
    copy_thresh(th.data());
    const size_t nr = R.n_rows, N = R.n_docs;
    std::vector<std::pair<uint64_t, uint64_t>> mp(nr);     // (offset in doc 0, length), mem_finder.hpp:394-397
    uint64_t total = 0;
    for (size_t r = 0; r < nr; r++) { mp[r] = {(uint64_t)R.mum_offsets[r * N], R.length[r]}; total += R.length[r] + 1; }
    std::sort(mp.begin(), mp.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    fwd.assign(total, 0); rev.assign(total, 0);
    const uint64_t half0 = doc_len_[0] + 1;
    uint64_t off = 0;
    for (size_t r = 0; r < nr; r++) {
        const uint64_t first = mp[r].first, len = mp[r].second;
        const uint64_t revpos = 2 * half0 - first - len - 1;
        for (uint64_t j = 0; j < len; j++, off++) {
            if (first + j < th.size() && th[first + j] < len - j) fwd[off] = th[first + j];
            if (revpos + j < th.size() && th[revpos + j] < len - j) rev[off] = th[revpos + j];
        }
        off++;   // terminator stays 0
    }
}

void Engine::run(const mmt_params& p) {
    MMT_HIP(hipSetDevice(device_));
    auto t0 = std::chrono::steady_clock::now();
    for (auto& ev : ev_) ev->reset();
    for (float& f : stage_ms_) f = 0.f;
    for (float& f : scan_ms_) f = 0.f;
    merged_thresh_valid_ = false;
    d_tap_len_.release(); d_tap_cnt_.release(); d_tap_off_.release(); d_tap_sa_.release();      // (the tap of the run before)
    sink_written_path_.clear();
    sink_discarded_ = false;
    lcp_col_ready_ = false;
    want_anchor_ranks_ = p.merge_metadata != 0;
    anchor_ranks_valid_ = false;
    rows_ = HostRows();
    rows_pending_ = 0;
    rows_.mum_mode = p.max_doc_freq == 1;
    rows_.n_docs = doc_len_.size();
    n_cand_ = 0; thresh_len_ = 0; thresh16_valid_ = false; bumbl_.clear();
    if (doc_len_.empty()) return;                       // mumemto_api.cpp:338-340
    if (!input_valid_)
        throw std::runtime_error("the engine holds no input (a partitioned run consumed it): call set_input first");
    if (preset_ && (p.use_revcomp != 0) != revcomp_)
        throw std::runtime_error("the text / stream handed over was laid out with the other strand setting");
    // merge metadata is defined for strict multi-MUMs only (include/pfp_mum.hpp:178-183): with partial or MEM
    // parameters nested accepted intervals share anchor entries and the threshold of a position is not unique
    if (p.merge_metadata && !(p.max_doc_freq == 1 && (p.num_distinct == 0 || p.num_distinct == doc_len_.size())))
        throw std::runtime_error("merge metadata (-M / -n) needs strict multi-MUM parameters (every document, one "
                                 "occurrence each)");
    auto finish = [&]() {
        for (int i = 0; i < 6; i++) stage_ms_[i] = ev_[i]->ms();
        stage_ms_[1] += scan_ms_[3];                    // the windows of the columns are produced between the scans
        stage_ms_[6] = scan_ms_[3];                     // ... and reported on their own as well (the emitter / the guided batches)
        stage_ms_[2] += scan_ms_[0];                    // the LCP column is gathered range by range inside the scan
        stage_ms_[3] = scan_ms_[1];                     // k_scan (+ the window tables of the wide-document path)
        stage_ms_[4] = scan_ms_[2];
        stage_ms_[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    if (preset_ == 2) {                                 // stream handed over: scan it as it is
        producer_used_ = 0;
        columns_kept_ = true;
        scan(p);
        make_rows(p);
        finish();
        return;
    }
    ev_[0]->start(stream_);
    if (preset_ == 0) build_text(p.use_revcomp != 0);
    ev_[0]->stop(stream_);
    if (drop_input_after_text_ && preset_ == 0 && d_bases_ && d_bases_ == d_bases_own_.get()) {
        MMT_HIP(hipStreamSynchronize(stream_));
        d_bases_own_.release(); d_bases_ = nullptr; input_valid_ = false;
    }
    // what the run before left on the device goes first: the stages of this run are sized by what the heap has left
    const bool slim = lean_ || wide_;
    if (slim) {
        MMT_HIP(hipStreamSynchronize(stream_));
        d_plcp_a_.release(); d_lcp_.release(); d_long_.release(); d_wpre_.release(); d_wsuf_.release(); d_wide_.release();
        d_cand_.release(); d_rank_.release(); d_rank64_.release();
        d_sa_.release(); d_sa_hi_.release(); d_bwt_.release(); d_cols_.release();
        lcp_whole_ = false;
    }
    int kind = producer_;
    {
        std::vector<uint64_t> hist;
        d2h(hist, d_hist_.get(), 256, stream_);
        const bool reserved = hist[0] || hist[1] || hist[2];
        if (kind == 0) {
            // automatic: prefix-free parsing, which wins by the redundancy of the collection (2.4x on the 16-haplotype
            // bench, 3x at 0.1 % divergence), unless the text holds bytes the parse reserves (<= 0x02) or there are
            // too few documents for the dictionary to be much smaller than the text (measured, 1 % divergence:
            // 3 x 4.6 Mbp 8.6 vs 12.7 ms, 4 x 30 Mbp 80 vs 108 ms for the direct sort; even at 6 documents)
            const char* env = std::getenv("MUMEMTO_PRODUCER");      // "direct" | "pfp" | "guided" override
            const bool forced_pfp = env && std::string(env) == "pfp";
            const bool few_docs = doc_len_.size() <= 4 && !forced_pfp;
            kind = (env && std::string(env) == "direct") || reserved || few_docs ? 1 : 2;
            if (env && (std::string(env) == "guided" || std::string(env) == "expand") && !reserved) kind = 3;
            // beyond one 32-bit suffix array only the parse works (MMT_FORCE_WIDE: the same choice, for tests)
            if (wide_ && !reserved && kind == 1) kind = 2;
        }
        if (packed_) kind = 3;                  // a packed text (textref.hpp) is read by the bucket-wise producer only
        if (kind == 1 && n_ >= NARROW_LIMIT)
            throw std::runtime_error(reserved ? "texts of 2^32 characters or more must not contain the bytes 0x00-0x02 "
                                                "(reserved by the prefix-free parse)"
                                              : "the direct suffix sort handles texts below 2^32 - 4096 characters");
    }
    if (kind == 1) {
        // ---- the direct producer: whole columns (narrow texts only), cut into windows by scan() ----
        if (!d_sa_.owned() || !d_bwt_.owned() || !d_sa_hi_.owned()) { d_sa_.release(); d_sa_hi_.release(); d_bwt_.release(); }
        ev_[1]->start(stream_);
        suffix_sort();
        producer_used_ = 1; producer_expanded_ = false;
        ev_[1]->stop(stream_);
        const bool lean = slim || wants_lean();
        if (lean) release_sort_scratch();
        ev_[2]->start(stream_); lcp_bwt(); ev_[2]->stop(stream_);
        if (lean) d_long_.release();
        columns_kept_ = true;
        scan(p);
        if (lean && lcp_whole_ && !lcp_col_ready_) d_plcp_a_.release();
        if (lean) { d_wpre_.release(); d_wsuf_.release(); d_wide_.release(); d_cand_.release(); }
        make_rows(p);
        finish();
        return;
    }
    // ---- prefix-free parsing: the tables of the parse, then the stream window by window (never stored) ----
    // the stream does not depend on (w, p): the automatic producer uses short phrases, which shrink the
    // dictionary 2.2x on the bench workload; beyond ~1 G characters a wider window keeps the groups of
    // short phrase suffixes (all occurrences of a trigger window) small (gpurun sweeps)
    // (94 x 64 Mbp, 12 G characters: w = 14 leaves no suffix group larger than an emitter tile -- every 14-mer is
    // rare enough -- and takes 1054 ms against 1111 for 10 / 30, 1171 for 10 / 50, 1981 for 10 / 100, 1376 for 8 / 30)
    const bool big = n_ >= NARROW_LIMIT;
    // (parse positions are 32 bits here: beyond ~48 G characters the modulus grows so that the parse keeps below 1.6 G phrases)
    const uint32_t auto_w = n_ < (1ull << 30) ? 6 : (big ? 14 : 10);
    uint32_t auto_p = n_ < (1ull << 30) ? 16 : (uint32_t)std::max<uint64_t>(30, n_ / 1600000000ull + 1);
    // (a modulus that divides the hash of w equal bases ends a phrase at EVERY position of a run of that base -- an assembly gap of
    // megabases becomes megabases of phrases, and the bucket-wise producer cannot slice its bin (guided.cpp): the next modulus)
    {
        auto kr = [](uint8_t c, uint32_t w) { uint64_t h = 0; for (uint32_t i = 0; i < w; i++) h = (h * 256 + c) % 1999999973ull; return h; };
        for (;; auto_p++) {
            bool bad = false;
            for (const char c : {'A', 'C', 'G', 'T', 'N'}) bad = bad || kr((uint8_t)c, auto_w) % auto_p == 0;
            if (!bad) break;
        }
    }
    if (kind == 3 || kind == 4) pfp_want_guided_ = true;
    run_slices_ = text_passes_ = batches_ = 0; staged_ = false;
    stream_min_len_ = p.min_match_len;
    ev_[1]->start(stream_);
    // (the stream does not depend on the parameters of the parse: a producer named without them gets the automatic ones)
    pfp_prepare(producer_ == 0 || !pfp_w_ ? auto_w : pfp_w_, producer_ == 0 || !pfp_p_ ? auto_p : pfp_p_);
    pfp_want_guided_ = false;
    ev_[1]->stop(stream_);
    producer_used_ = pfp_->guided ? 3 : 2;
    producer_expanded_ = pfp_->guided && pfp_->expand;
    // whole columns next to the windows when somebody wants to look at them afterwards
    columns_kept_ = keep_columns_ > 0 || (keep_columns_ < 0 && n_ < (1ull << 26));
    if (columns_kept_) {
        d_sa_.ensure(n_ + 16); d_bwt_.ensure(n_ + 16); d_plcp_a_.ensure(n_ + 16);
        if (wide_) d_sa_hi_.ensure(n_ + 16);
    }
    want_anchor_ranks_ = p.merge_metadata != 0;
    if (want_anchor_ranks_) {
        const uint64_t anchor = std::min<uint64_t>(doc_len_[0], n_);
        if (wide_) d_rank64_.ensure((size_t)anchor + 1); else d_rank_.ensure((size_t)anchor + 1);
    }
    ScanState S;
    scan_begin(p, S);
    streamed_ = true;
    sink_open(p.max_doc_freq == 1);
    try {
        if (pfp_->guided) guided_stream(S, p); else pfp_stream(S, p);
        sink_flush(S);
    } catch (...) {
        try { sink_close(false); } catch (...) {}
        sink_written_path_.clear();
        throw;
    }
    sink_close();
    scan_end(S);
    anchor_ranks_valid_ = want_anchor_ranks_;
    lcp_col_ready_ = columns_kept_;
    lcp_whole_ = false;
    if (slim) {
        release_sort_scratch();
        d_wpre_.release(); d_wsuf_.release(); d_wide_.release(); d_cand_.release();
        for (int k = 0; k < 2; k++) { w_sa_[k].release(); w_hi_[k].release(); w_bwt_[k].release(); w_lcp_[k].release(); }
        if (one_shot_) pool::shrink_async(device_);      // (the tables of the parse and the windows are gone: rows and outputs remain)
    }
    make_rows(p);
    finish();
}

void Engine::parse_only(bool revcomp, uint32_t w, uint32_t p) {
    MMT_HIP(hipSetDevice(device_));
    if (doc_len_.empty()) throw std::runtime_error("no input");
    if (!input_valid_) throw std::runtime_error("the engine holds no input: call set_input first");
    build_text(revcomp);
    pfp_parse(w, p, true);
}

void Engine::copy_text(uint8_t* out) const {
    if (packed_) {                                      // unpacked in pieces (tests; the packed text exists for texts the bytes do not fit)
        DevBuf<uint8_t> piece;
        const uint64_t PIECE = 1ull << 28;
        piece.ensure(std::min<uint64_t>(PIECE, n_) + 1);
        for (uint64_t at = 0; at < n_; at += PIECE) {
            const uint64_t len = std::min<uint64_t>(PIECE, n_ - at);
            k::unpack_text(text_ref(), at + 1, len, piece.get(), stream_);
            MMT_HIP(hipMemcpyAsync(out + at, piece.get(), len, hipMemcpyDeviceToHost, stream_));
            MMT_HIP(hipStreamSynchronize(stream_));
        }
        return;
    }
    MMT_HIP(hipMemcpy(out, text_ptr(), n_, hipMemcpyDeviceToHost));
}
static const char* NOT_KEPT = "the columns of this run were produced window by window and not kept "
                              "(mmt_engine_keep_columns before the run)";
void Engine::copy_sa(uint32_t* out) const {
    if (!columns_kept_) throw std::runtime_error(NOT_KEPT);
    if (wide_ && n_ >= NARROW_LIMIT) throw std::runtime_error("40-bit suffix array: use the 64-bit accessor");
    MMT_HIP(hipMemcpy(out, d_sa_.get(), n_ * 4, hipMemcpyDeviceToHost));
}
void Engine::copy_sa64(uint64_t* out) const {
    if (!columns_kept_) throw std::runtime_error(NOT_KEPT);
    std::vector<uint32_t> lo(n_);
    MMT_HIP(hipMemcpy(lo.data(), d_sa_.get(), n_ * 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> hi;
    if (wide_) { hi.resize(n_); MMT_HIP(hipMemcpy(hi.data(), d_sa_hi_.get(), n_, hipMemcpyDeviceToHost)); }
    for (uint64_t j = 0; j < n_; j++) out[j] = (uint64_t)lo[j] | (wide_ ? (uint64_t)hi[j] << 32 : 0);
}
void Engine::copy_lcp(uint32_t* out) {
    if (!columns_kept_) throw std::runtime_error(NOT_KEPT);
    if (lcp_col_ready_) { MMT_HIP(hipMemcpy(out, d_plcp_a_.get(), n_ * 4, hipMemcpyDeviceToHost)); return; }
    if (lcp_whole_) { MMT_HIP(hipMemcpy(out, d_lcp_.get(), n_ * 4, hipMemcpyDeviceToHost)); return; }
    // the run scanned the stream range by range: gather the column again, piece by piece
    if (!d_plcp_a_.get()) throw std::runtime_error("the LCP column of this run is gone");
    const uint64_t piece = 1ull << 26;
    DevBuf<uint32_t> buf;
    buf.ensure(piece + 16);
    for (uint64_t j0 = 0; j0 < n_; j0 += piece) {
        const uint64_t c = std::min(piece, n_ - j0);
        k::lcp_gather(d_plcp_a_.get(), sa_col(), j0, c, buf.get(), stream_);
        MMT_HIP(hipMemcpyAsync(out + j0, buf.get(), c * 4, hipMemcpyDeviceToHost, stream_));
        MMT_HIP(hipStreamSynchronize(stream_));
    }
}
void Engine::copy_bwt(uint8_t* out) const {
    if (!columns_kept_) throw std::runtime_error(NOT_KEPT);
    MMT_HIP(hipMemcpy(out, d_bwt_.get(), n_, hipMemcpyDeviceToHost));
}
void Engine::copy_candidates(uint32_t* out) const {
    if (scan_ranges_ > 1) throw std::runtime_error("candidates are kept for single-range scans only");
    if (n_cand_ && !d_cand_.get()) throw std::runtime_error("the candidate list of this run was released (lean mode)");
    if (n_cand_) MMT_HIP(hipMemcpy(out, d_cand_.get(), n_cand_ * sizeof(k::Cand), hipMemcpyDeviceToHost));
}
const uint16_t* Engine::thresh_device() {
    const size_t n = thresh_len();
    if (!n) return nullptr;
    if (!thresh16_valid_) {
        MMT_HIP(hipSetDevice(device_));
        d_thresh16_.ensure(n);
        k::thresh_narrow(thresh_device32(), n, d_thresh16_.get(), stream_);
        MMT_HIP(hipStreamSynchronize(stream_));
        thresh16_valid_ = true;
    }
    return d_thresh16_.get();
}
void Engine::copy_thresh32(uint32_t* out) const {
    if (thresh_len()) MMT_HIP(hipMemcpy(out, thresh_device32(), thresh_len() * 4, hipMemcpyDeviceToHost));
}
void Engine::copy_thresh(uint16_t* out) {
    if (thresh_len()) MMT_HIP(hipMemcpy(out, thresh_device(), thresh_len() * 2, hipMemcpyDeviceToHost));
}

}  // namespace mmt
