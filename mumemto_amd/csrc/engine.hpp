// engine.hpp -- the hot path as one object: text -> SA/LCP/BWT -> scan -> rows.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mumemto_gpu.h"
#include "device_utils.hpp"
#include "kernels.hpp"
#include "merge_types.hpp"
#include "pfp.hpp"
#include "sorter.hpp"
#include "textref.hpp"

namespace mmt {

// Host-side result of one run, in the shapes the reference's collectors use
// (mumemto_library/mumemto_api.cpp:137-166 MEM, :241-286 MUM).
struct HostRows {
    bool mum_mode = true;
    size_t n_docs = 0, n_rows = 0, n_occ = 0;
    const uint32_t* length = nullptr;
    // MUM mode: n_rows * n_docs, -1 = document absent; strands 1 = '+'
    const int64_t* mum_offsets = nullptr;
    const uint8_t* mum_strands = nullptr;
    // MEM mode: occ_start[n_rows + 1], flat occurrences in suffix-array order
    const uint64_t* occ_start = nullptr;
    const int64_t* mem_offsets = nullptr;
    const uint64_t* mem_docs = nullptr;
    const uint8_t* mem_strands = nullptr;
    const char* text = nullptr;        // PREFIX.mums / PREFIX.mems bytes (page-locked host memory)
    size_t text_len = 0;
};

// One window of the stream's columns, as the scan sees it: entries [base, base + len) of the suffix array / BWT / LCP
// columns; closing positions below `first` belong to the window before (those entries only serve the walks to the left).
struct ColWindow {
    SaCol sa;                 // entry t of the window is sa[sa_off + t]
    const uint8_t* bwt = nullptr;
    const uint32_t* lcp = nullptr;
    uint64_t base = 0, sa_off = 0;
    uint32_t len = 0, first = 0;
    bool more_left = false;   // entry 0 is not the start of the stream (a walk that reaches it is reported)
    bool transient = false;   // the columns are gone after the window: accepted rows keep their suffix-array entries
};
// what the windows of one run share (Engine::scan_begin / scan_window / scan_end)
struct ScanState {
    k::ScanArgs a;
    uint64_t cap = 0, ext0 = 0;
    size_t rows_used = 0, ev_at = 0;
    uint32_t max_doc_freq = 0;
    bool merge = false;
};

class Engine {
public:
    Engine(int device, hipStream_t stream);
    ~Engine();
    Engine(const Engine&) = delete;

    void set_input_device(const uint8_t* d_bases, const uint64_t* doc_len, size_t n_docs);
    void set_input_host(const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs);
    // one pass of the path on the input that is set, as run_partitioned_docs does for a single suffix array (the
    // raw bases are dropped once the text exists)
    void run_once_dropping_input(const mmt_params& p);
    // Input that arrives document by document while the caller still reads the others (mmt_engine_run_files): a device
    // buffer with one slot per document, filled by the caller (any thread, its own stream), then declared as the input.
    uint8_t* begin_input_slots(size_t bytes);
    void finish_input_slots(const std::vector<size_t>& slot, const uint64_t* doc_len, size_t n_docs);
    // what the last run left on the device -- columns, scratch, results -- goes back to the heap (a new run replaces it)
    void forget_last_run();
    uint64_t auto_max_text() const;
    // the documents in separate host buffers (no concatenation on the host: one H2D copy per document)
    void set_input_host_docs(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs);
    // The same, but nothing is uploaded yet: the host buffers must stay valid until the next run() has built its text.  A
    // collection whose raw bases would not fit the device next to its packed text (BASELINE configs[4]: 287 GB of bases,
    // 143 GB packed) is then packed document by document through a staging buffer (build_text); a text that keeps one byte
    // per character is uploaded as a whole when the run begins.
    void set_input_host_docs_deferred(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs);
    // The documents SUPPLIED one at a time: `fn(user, d, dst, doc_len[d])` writes the bases of document d to dst (page-locked,
    // doc_len[d] bytes) and returns 0; it is asked for the documents in order, for every one again when the text is built a
    // second time (a packed text that turns out to hold bytes the parse reserves is rebuilt with a byte per character).  The reference
    // streams its FASTA files through the parser and never holds the collection (include/newscan.hpp:265-325): this is the
    // entry for a collection that does not fit the HOST either (configs[4]: 287 GB of bases).  run_supplied runs it as ONE
    // text (no anchor partitions: those would read the documents again).
    using DocSupplier = int (*)(void* user, uint64_t doc, uint8_t* dst, uint64_t len);
    void set_input_supplier(DocSupplier fn, void* user, const uint64_t* doc_len, size_t n_docs);
    void run_supplied(DocSupplier fn, void* user, const uint64_t* doc_len, size_t n_docs, const mmt_params& p);
    // Stage checkpoints of the reference CLI (src/pfp_mum.cpp:97-111 `-a`, :122-124 `-p`): the caller hands over
    // the text T itself (UPPER(F) '$' [revcomp '$'] per document, e.g. rebuilt from PREFIX.parse/.dict) or the
    // whole stream (SA / LCP / BWT of the real suffixes, sentinel entry dropped); run() then skips the stages
    // that would have produced them.  `revcomp` says how the documents are laid out in that text.
    void set_text_host(const uint8_t* text, uint64_t n, const uint64_t* doc_len, size_t n_docs, bool revcomp);
    void set_stream_host(const uint32_t* sa, const uint32_t* lcp, const uint8_t* bwt, uint64_t entries,
                         const uint64_t* doc_len, size_t n_docs, bool revcomp);
    // the same for streams of any length: 40-bit suffix-array entries as low word + high byte (the layout of the
    // reference's .sa dump, include/pfp_lcp_mum.hpp:323-369, split into two arrays)
    void set_stream_host40(const uint32_t* sa_lo, const uint8_t* sa_hi, const uint32_t* lcp, const uint8_t* bwt,
                           uint64_t entries, const uint64_t* doc_len, size_t n_docs, bool revcomp);
    void run(const mmt_params& p);
    // Same job for host-resident input of any size: if the text exceeds max_text characters (0 = what fits the
    // device memory as one suffix array) the documents are processed as anchor partitions and merged
    // (strict multi-MUMs only, like the reference's merge).
    void run_partitioned_host(const uint8_t* h_bases, const uint64_t* doc_len, size_t n_docs, const mmt_params& p,
                              uint64_t max_text);
    void run_partitioned_docs(const uint8_t* const* doc_ptr, const uint64_t* doc_len, size_t n_docs, const mmt_params& p,
                              uint64_t max_text);
    size_t partitions_used() const { return partitions_used_; }
    // merged .athresh (L_0 + 1 entries) of the last partitioned run, empty otherwise
    // (host copy made when first asked for: the thresholds of a genome-sized anchor are 6 GB)
    const std::vector<uint16_t>& merged_thresh();
    // the MUM-mode rows of the last run as they sit in HBM (valid until the next run)
    void rows_mum_device(const uint32_t** len, const int64_t** off, const uint8_t** st) const {
        if (merged_thresh_valid_) { *len = merged_.d_length.get(); *off = merged_.d_offsets.get(); *st = merged_.d_strands.get(); }
        else { *len = d_olen_.get(); *off = d_ooffs_.get(); *st = d_ost_.get(); }
    }
    bool last_run_partitioned() const { return merged_thresh_valid_; }
    // SA/LCP/BWT producer: 0 = automatic, 1 = direct suffix sort of the text (A8), 2 = prefix-free parsing (A2-A4),
    // 3 = prefix-free parsing without the dictionary's suffix array (guided.cpp; automatic when that would not fit)
    void set_producer(int kind, uint32_t w, uint32_t p) { producer_ = kind; pfp_w_ = w; pfp_p_ = p; }   // 0: chosen by the size
    int producer_used() const { return producer_used_; }
    bool producer_expanded() const { return producer_expanded_; }   // producer 3 sorted representatives, the emitter expanded them
    // bucket-wise producer, last run: slices of run bins, passes over the text, batches, several batches per pass
    void producer_stats(uint64_t out[4]) const { out[0] = run_slices_; out[1] = text_passes_; out[2] = batches_; out[3] = staged_ ? 1 : 0; }
    // A2 alone (after build_text): phrases, dictionary, parse.  Used by -P / -K and the parity tests.
    void parse_only(bool revcomp, uint32_t w, uint32_t p);
    void pfp_copy_dict(std::vector<uint8_t>& out);
    void pfp_copy_parse(std::vector<uint32_t>& out);
    const PfpState& pfp_state() const { return *pfp_; }
    // One-shot use (the command line; automatically when the buffers of all stages together would not fit the
    // device): every stage gives its scratch back before the next one allocates, so that the footprint is the
    // largest stage instead of the sum -- at the price of hipMalloc calls in every run.
    void set_lean(bool on) { lean_ = on; }
    // this process runs ONE job and exits (mumemto_exec): once a stage's peak is over, the free top of the device heap goes
    // back to the driver on a helper thread (pool::shrink_async) instead of being torn down on the way out
    void set_one_shot(bool on) { one_shot_ = on; }
    // Multi-GPU runs of the modes the anchor merge cannot serve (partial multi-MUMs, multi-MEMs: the reference refuses
    // to merge them, include/pfp_mum.hpp:178-183): every rank builds the same stream and scans only its share of the
    // suffix-array positions -- closing positions in [n * index / count, n * (index + 1) / count), cut at multiples of
    // 4096 -- so that the outputs of the ranks, concatenated in rank order, are the output of one GPU (rows come out in
    // order of their closing position).  count = 1 switches it off.
    void set_scan_shard(uint32_t index, uint32_t count) {
        if (count == 0 || index >= count) throw std::runtime_error("scan shard index out of range");
        shard_index_ = index; shard_count_ = count;
    }
    // The columns of the stream exist one window at a time (the reference never stores them either:
    // pfp_lcp_mum.hpp:197).  on = 1: every window is also copied into whole columns, so that copy_sa / copy_lcp /
    // copy_bwt work after the run (tests, -A); 0: never; -1 (default): for texts below 2^26 characters.
    void set_keep_columns(int on) { keep_columns_ = on; }
    bool columns_kept() const { return columns_kept_; }
    // of the last run: [0] entries of the stream this engine produced (its share + the left extensions of its windows),
    // [1] bytes of the window buffers that held them (high-water mark), [2] windows, [3] bytes of the suffix-array entries
    // that left with accepted rows
    void stream_stats(uint64_t out[4]) const {
        out[0] = stream_entries_; out[1] = window_bytes_peak_; out[2] = scan_ranges_; out[3] = pool_used_ * (wide_ ? 5 : 4);
    }
    // The shares of the ranks of a sharded run as (first suffix-array entry, entries): ranges cut at multiples of 4096 for
    // the parse proper, whole bins of leading characters for the bucket-wise producer (guided.cpp) -- every rank derives
    // the same list.
    const std::vector<std::pair<uint64_t, uint64_t>>& sort_pieces() const { return sort_pieces_; }   // (first entry, entries)
    uint8_t* bwt_device() const { return d_bwt_.get(); }
    void release_sort_scratch();
    // gives every column and scratch buffer of the last run back to the device heap (results already downloaded stay)
    // keep_anchor_ranks: the suffix ranks of the anchor stay (a fold of this run's rows with other partitions' is still to
    // be put into direct-run order, merge.cpp sort_like_direct)
    void release_columns(bool keep_anchor_ranks = false);
    bool wants_lean() const;

    // Results of the last run.  rows_meta(): counts and mode only; rows(need): also the host copies asked for
    // (ROWS_ARRAYS = library arrays, ROWS_TEXT = PREFIX.mums / .mems bytes), downloaded from HBM on first use.
    enum { ROWS_ARRAYS = 1, ROWS_TEXT = 2 };
    const HostRows& rows_meta() const { return rows_; }
    const HostRows& rows(int need = ROWS_ARRAYS | ROWS_TEXT) { fetch_rows(need); return rows_; }
    void fetch_rows(int need);
    // PREFIX.mums of the NEXT run straight to `path` while the run goes on: the rows of a window are final when the window
    // has been verified (they come out in order of their closing position), so their bytes are formatted, copied to
    // page-locked memory and written by a helper thread while the later windows are produced (multi-MUM runs of the
    // producers that stream; any other run writes the file at its end as before).  "" switches it off.
    // drop_rows: the rows of a window are forgotten once they are written whatever the size of the text (the ranks of a
    // sharded run write their pieces of PREFIX.mems this way: nothing is gathered, mumemto_exec --gpus N)
    // keep_rows: never -- whatever the size of the text -- because the caller will ask for the rows afterwards (PREFIX.thresh /
    // .thresh_rev are cut out of the threshold column row by row: mem_finder.hpp:116-157)
    void set_text_sink(const std::string& path, bool drop_rows = false, bool keep_rows = false) {
        sink_path_ = path; sink_force_discard_ = drop_rows && !path.empty(); sink_keep_rows_ = keep_rows;
    }
    // PREFIX.mums / .mems of the last run straight to a file: the bytes leave HBM in pieces and every piece is written
    // while the next ones are still on their way (rows(ROWS_TEXT) + one write otherwise)
    int kmer_in_share(const uint8_t* kmer, size_t k) const;        // (guided.cpp)
    // (bytes written, digest of those bytes in file order) of the last run's sink; the digest is kept when the sink is
    // "/dev/null" -- a test run whose rows nobody can keep -- or MMT_SINK_DIGEST is set
    void text_sink_digest(uint64_t out[2]) const { out[0] = sink_written_; out[1] = sink_digest_value_; }
    void write_text_file(const std::string& path);
    // The row tap (tests/bigchecks.py check_bins_complete): every accepted interval of the NEXT runs whose match begins with
    // one of the given k-mers (k <= 16 characters each) leaves a copy of its length and all its suffix-array entries before
    // its window drops it -- what a full-size run that keeps nothing else can still be asked.  n = 0 switches it off.
    void set_row_tap(const uint8_t* kmers, size_t n, size_t k, size_t max_rows, size_t max_occ);
    // (rows, entries) the last run tapped; the copies: length[rows], occ_start[rows + 1], sa[entries] (rows in no particular order)
    void row_tap_counts(uint64_t out[2]);
    void row_tap_get(uint32_t* length, uint64_t* occ_start, uint64_t* sa);
    // every text position of the resident text whose suffix begins with one of the k-mers: (position, which k-mer), ascending
    // by position; returns the number found (may exceed cap: then only cap were written)
    uint64_t kmer_positions(const uint8_t* kmers, size_t n, size_t k, uint64_t* pos, uint32_t* which, uint64_t cap);
    const std::string& bumbl();
    // PREFIX.thresh / PREFIX.thresh_rev contents (mem_finder.hpp:116-157); needs a merge_metadata MUM run
    void thresh_files(std::vector<uint16_t>& fwd, std::vector<uint16_t>& rev);
    uint64_t text_length() const { return n_; }
    size_t n_docs() const { return doc_len_.size(); }
    const std::vector<uint64_t>& doc_len() const { return doc_len_; }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }

    // stage introspection
    void copy_text(uint8_t* out) const;
    void copy_sa(uint32_t* out) const;          // narrow runs only
    void copy_sa64(uint64_t* out) const;
    void copy_lcp(uint32_t* out);
    void copy_bwt(uint8_t* out) const;
    size_t n_candidates() const { return n_cand_; }
    void copy_candidates(uint32_t* out) const;
    // (after a run that went through anchor partitions: the merged thresholds, like the rows)
    size_t thresh_len() const { return merged_thresh_valid_ ? merged_.thresh_len : thresh_len_; }
    // Thresholds are 32 bits wide inside the engine, the fold and the exchange (the reference's 16-bit column
    // saturates at 65535, mem_finder.hpp:299,328, and a merged MUM beyond that whose next-best match is as long would be
    // accepted unproven).  The 16-bit forms -- PREFIX.athresh / .thresh, copy_thresh, thresh_device -- saturate like the
    // reference's and are made from the 32-bit column when asked for.
    void copy_thresh(uint16_t* out);
    void copy_thresh32(uint32_t* out) const;
    const uint16_t* thresh_device();
    const uint32_t* thresh_device32() const { return merged_thresh_valid_ ? merged_.d_thresh.get() : d_thresh_.get(); }
    const uint32_t* isa_device() const { return d_rank_.get(); }        // narrow runs
    const uint64_t* isa_device64() const { return d_rank64_.get(); }    // wide runs
    bool anchor_ranks_valid() const { return anchor_ranks_valid_; }
    bool wide() const { return wide_; }
    // The text buffer also is the string V = Dollar . T . Dollar^w the prefix-free parse reads (newscan.hpp:248, :359):
    // one Dollar byte sits in front of T (64 bytes of padding keep T 16-byte aligned), 32 Dollar bytes and then
    // zeros behind it, so V[i] = text_ptr()[i - 1] without a second copy of the text.  Nothing reads T[n ..) for its
    // value: every comparison of suffixes is clamped to the end of the shorter one.
    static constexpr size_t TEXT_FRONT = 64, TEXT_BACK = 128;
    uint8_t* text_ptr() const { return d_text_.get() ? d_text_.get() + TEXT_FRONT : nullptr; }
    // The text as the kernels of the parse and of the bucket-wise producer read it (textref.hpp): the byte buffer above, or
    // -- MMT_PACKED_TEXT=1, or automatically when one byte per character would not fit next to the tables of the parse --
    // two bits per character + a sorted list of exception runs.  Only the bucket-wise producer runs on a packed text.
    TextRef text_ref() const;
    bool have_text() const { return text_ptr() != nullptr || d_packed_.get() != nullptr; }
    bool packed_text() const { return packed_; }
    uint32_t exception_runs() const { return (uint32_t)h_runs_.size(); }
    void finish_text_padding();
    SaCol sa_col() const { SaCol c; c.lo = d_sa_.get(); c.hi = wide_ ? d_sa_hi_.get() : nullptr; return c; }
    size_t scan_ranges() const { return scan_ranges_; }
    const float* stage_ms() const { return stage_ms_; }
    DevBuf<uint8_t>& scratch() { return d_temp_; }
    // page-locked staging of a merged result's .mums bytes on their way to a file (grows only; merge.cpp)
    char* merge_text_staging(size_t n) { h_merge_text_.ensure(n); return h_merge_text_.get(); }

private:
    void layout_docs(bool revcomp);
    void build_text(bool revcomp);
    bool want_packed_text() const;
    bool want_packed_text_of(uint64_t n, bool by_size_only = false) const;
    void finish_packed_text(DevBuf<uint64_t>& ev_start, DevBuf<uint64_t>& ev_end, DevBuf<uint32_t>& ev_count, uint32_t ev_cap);
    void suffix_sort();
    void pfp_parse(uint32_t w, uint32_t p, bool keep_dict_inputs);
    void pfp_prepare(uint32_t w, uint32_t p);
    void pfp_prepare_emitter(uint32_t w);
    // group tables of the emitter from the entry tables at hand (pfp.cpp); the BWT codes of the oversized groups' sort keys
    void pfp_group_tables(uint32_t E, uint32_t G, uint64_t out_lo, uint64_t out_hi, bool slim);
    void pfp_emit_codes(int shift);
    void guided_prepare();
    void guided_check_errors(const char* what);
    void build_giant(const std::vector<uint64_t>& hist);
    void pfp_stream(ScanState& S, const mmt_params& p);
    void pfp_emit_window(uint64_t b0, uint64_t c1, int set);
    uint64_t pfp_first_tile(uint64_t b0);         // emitter tile in which the group covering stream entry b0 + 1 begins
    void guided_stream(ScanState& S, const mmt_params& p);
    void guided_stream_expand(ScanState& S, const mmt_params& p);
    void lcp_bwt();
    void scan(const mmt_params& p);
    void scan_begin(const mmt_params& p, ScanState& S);
    bool scan_window(ScanState& S, const ColWindow& w, const mmt_params& p);    // false: a walk ran off the left edge
    void scan_end(ScanState& S);
    void window_reserve(int set, uint64_t entries);
    ColWindow window_view(int set, uint64_t base, uint32_t len, uint32_t first) const;
    void keep_window(const ColWindow& w);
    void shard_range(uint64_t& lo, uint64_t& hi) const;
    EventPair& next_range_event(ScanState& S, int kind);
    void make_rows(const mmt_params& p);

    int device_;
    hipStream_t stream_;
    bool own_stream_ = false;

    // input
    const uint8_t* d_bases_ = nullptr;    // borrowed or = d_bases_own_
    DevBuf<uint8_t> d_bases_own_;
    std::vector<uint64_t> doc_len_, doc_base_, doc_start_;
    DevBuf<uint64_t> d_doc_base_, d_doc_start_;
    bool revcomp_ = true;
    uint64_t n_ = 0;
    bool wide_ = false;                   // this text runs with 40-bit positions (wide.hpp)
    bool input_valid_ = false;
    bool drop_input_after_text_ = false;  // host-fed runs upload the bases for every run: they are dead once T exists
    int preset_ = 0;                      // 0: build everything, 1: text handed over, 2: stream handed over

    // columns
    DevBuf<uint8_t> d_text_, d_bwt_, d_flags_, d_code_, d_temp_, d_sa_hi_;
    // the packed layout of the text (textref.hpp)
    DevBuf<uint64_t> d_packed_, d_excw_;
    DevBuf<ExcRun> d_runs_;
    std::vector<ExcRun> h_runs_;
    bool packed_ = false;
    std::vector<const uint8_t*> host_docs_;        // deferred host input (set_input_host_docs_deferred)
    DocSupplier supplier_ = nullptr;               // ... or its supplier (set_input_supplier)
    void* supplier_user_ = nullptr;
    // One-shot / wide runs: suffix array (low words, high bytes) and BWT are views into one block that is allocated
    // before any scratch (it then sits at the bottom of the device heap, and what is above it leaves one hole when it
    // goes).  The emitter writes these columns only after the dictionary and the parse are sorted, so until then the
    // block is the scratch of those sorts (DoublingSorter::reserve_in): 49 bytes per dictionary character that the run
    // does not hold twice.
    DevBuf<uint8_t> d_cols_;
    DevBuf<uint64_t> d_hist_, d_rank64_;
    DevBuf<uint32_t> d_sa_, d_rank_, d_lcp_, d_count_, d_plcp_a_;
    DevBuf<uint8_t> d_long_;
    // scan of more than ~1000 documents: window tables of the scanned range (block-wise prefix / suffix minima of the
    // LCP column, BWT change marks replaced by their running maximum)
    DevBuf<uint32_t> d_wpre_, d_wsuf_, d_wide_;
    bool lcp_whole_ = false;              // d_lcp_ holds the LCP column of the whole stream (one scan range)
    bool want_anchor_ranks_ = false, anchor_ranks_valid_ = false;
    bool lcp_col_ready_ = false;          // d_plcp_a_ holds the LCP column in suffix-array order (written by the producer)
    uint32_t shard_index_ = 0, shard_count_ = 1;
    std::vector<std::pair<uint64_t, uint64_t>> sort_pieces_;
    // the window of the columns that exists (two sets: the bucket-wise producer carries the tail of one into the next)
    DevBuf<uint32_t> w_sa_[2], w_lcp_[2];
    DevBuf<uint8_t> w_hi_[2], w_bwt_[2];
    int keep_columns_ = -1;
    bool columns_kept_ = false, streamed_ = false, pfp_want_guided_ = false;
    // suffix-array entries of the accepted rows (rows of a streamed run index this pool instead of the column)
    DevBuf<uint32_t> d_pool_lo_;
    DevBuf<uint8_t> d_pool_hi_;
    DevBuf<k::Row> d_rows_pool_;
    DevBuf<uint64_t> d_cap_cnt_, d_cap_off_;
    uint64_t pool_used_ = 0;
    // digest of a byte stream that arrives in pieces of any size (the text sink's bytes in file order)
    struct StreamDigest {
        uint64_t h = 0x6d756d656d746f35ull;
        uint8_t carry[8];
        uint32_t have = 0;
        void word(uint64_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
        void update(const char* p, size_t n) {
            size_t i = 0;
            while (have && have < 8 && i < n) carry[have++] = (uint8_t)p[i++];
            if (have == 8) { uint64_t w; std::memcpy(&w, carry, 8); word(w); have = 0; }
            for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, p + i, 8); word(w); }
            for (; i < n; i++) carry[have++] = (uint8_t)p[i];
        }
        uint64_t final() const { uint64_t x = h; for (uint32_t i = 0; i < have; i++) x = (x ^ carry[i]) * 0x100000001b3ull; return x ^ (x >> 31); }
    };
    // the row tap (set_row_tap)
    std::vector<uint64_t> tap_kmers_;            // 2 words per k-mer
    uint32_t tap_k_ = 0;
    uint64_t tap_cap_rows_ = 0, tap_cap_occ_ = 0;
    DevBuf<uint64_t> d_tap_kmers_, d_tap_off_, d_tap_sa_, d_tap_used_;
    DevBuf<uint32_t> d_tap_len_, d_tap_cnt_;
    void tap_window(const k::Row* rows, uint32_t n_rows, SaCol pool);
    // the text sink (set_text_sink)
    bool sink_force_discard_ = false, sink_keep_rows_ = false, sink_null_ = false, sink_want_digest_ = false;
    StreamDigest sink_digest_;
    uint64_t sink_digest_value_ = 0, sink_written_ = 0;
    struct SinkPiece { const char* p; size_t n; hipEvent_t ready; uint32_t block; };
    std::string sink_path_, sink_written_path_, sink_tmp_path_;
    bool sink_active_ = false, sink_mum_ = true, sink_discard_ = false, sink_discarded_ = false;
    uint64_t sink_total_rows_ = 0;
    int sink_fd_ = -1;
    size_t sink_rows_done_ = 0;
    uint64_t sink_bytes_ = 0;
    std::thread sink_thread_;
    std::mutex sink_mu_;
    std::condition_variable sink_cv_;
    std::deque<SinkPiece> sink_q_;
    bool sink_closing_ = false;
    std::string sink_error_;
    std::vector<std::unique_ptr<PinnedBuf<char>>> sink_blocks_;      // page-locked blocks, kept between runs
    std::vector<size_t> sink_block_cap_;
    std::vector<uint32_t> sink_block_pending_;     // pieces of a block the writer thread has not written yet (under sink_mu_)
    size_t sink_block_at_ = 0, sink_block_used_ = 0;
    DevBuf<char> d_piece_[2];                // two pieces: one is copied out on the copy stream while the next is formatted
    hipStream_t sink_stream_ = nullptr;
    hipEvent_t sink_copied_[2] = {nullptr, nullptr};
    uint32_t sink_pieces_ = 0;
    void sink_open(bool mum_mode);
    void sink_flush(ScanState& S);
    void sink_close(bool ok = true);        // ok: PREFIX.mums.tmp takes its name; otherwise it is removed
    char* sink_host_room(size_t n, uint32_t* block);
    void order_rows(const k::Row* rows_abs, uint32_t cnt);
    float emit_ms_ = 0.f;
    uint64_t stream_entries_ = 0, window_bytes_peak_ = 0;
    uint32_t stream_min_len_ = 20;      // minimum match length of the run in progress (the bins of the bucket-wise producer)
    size_t scan_ranges_ = 1;
    DoublingSorter sorter_;
    int sort_rounds_ = 0;
    std::unique_ptr<PfpState> pfp_{new PfpState()};
    bool lean_ = false;                   // release each stage's scratch before the next stage allocates
    bool one_shot_ = false;
    int producer_ = 0, producer_used_ = 1;
    bool producer_expanded_ = false;
    uint32_t run_slices_ = 0, text_passes_ = 0, batches_ = 0;
    bool staged_ = false;
    uint32_t pfp_w_ = 0, pfp_p_ = 0;
    // scan
    DevBuf<k::Cand> d_cand_;
    DevBuf<k::Row> d_rows_;
    DevBuf<uint32_t> d_thresh_;
    DevBuf<uint16_t> d_thresh16_;          // the saturated 16-bit form, made on demand (thresh_device)
    bool thresh16_valid_ = false;
    size_t n_cand_ = 0, thresh_len_ = 0;
    // A6 on the device
    DevBuf<uint64_t> d_doc_len_, d_rkeys_a_, d_rkeys_b_, d_tlen64_, d_toff_, d_occ64_, d_ooff_, d_omdoc_;
    DevBuf<uint32_t> d_rvals_a_, d_rvals_b_, d_order_, d_keep_, d_tlen_, d_ridx_, d_wpos_, d_wdoc_, d_olen_;
    DevBuf<int64_t> d_slot_off_, d_ooffs_;
    DevBuf<uint8_t> d_slot_st_, d_ost_;
    DevBuf<char> d_otext_;
    PinnedBuf<uint32_t> h_len_;
    PinnedBuf<int64_t> h_offs_;
    PinnedBuf<uint8_t> h_st_;
    PinnedBuf<uint64_t> h_occ_start_, h_mdoc_;
    PinnedBuf<char> h_text_, h_merge_text_;

    HostRows rows_;
    int rows_pending_ = 0;                // ROWS_* bits that still sit in HBM only
    MergedRows merged_;
    std::string merged_text_;
    size_t partitions_used_ = 1;
    bool merged_thresh_valid_ = false;
    std::string bumbl_;
    uint64_t num_distinct_eff_ = 0;
    float stage_ms_[8] = {0};
    std::unique_ptr<EventPair> ev_[6];
    std::vector<std::unique_ptr<EventPair>> range_ev_;      // per scan range: LCP gather, scan kernel, verification
    std::vector<int> range_ev_kind_;
    float scan_ms_[4] = {0, 0, 0, 0};       // LCP gather, scan kernel, verification, window production
};

// decimal formatting shared with the merge output
void append_uint(std::string& s, uint64_t v);

}  // namespace mmt
