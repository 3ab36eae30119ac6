/* mumemto_gpu.h -- device-resident entry points of libmumemto (MI355X).
 *
 * The drop-in ABI (mumemto.h) takes host strings, like the reference.  These
 * entry points expose the same hot path with inputs already in HBM, stage by
 * stage, for bench.py, the parity tests and multi-GPU drivers.  Plain C ABI:
 * pointers and sizes only.  Every call returns 0 on success; on failure the
 * message is in mmt_last_error() (thread-local) and nothing falls back to a
 * CPU path.
 *
 * Reference interfaces replaced (file:line under the reference tree):
 *   text layout           RefBuilder::build_input_file   src/ref_builder.cpp:211-314, :330-384
 *   SA/LCP/BWT stream     gsacak_lcp / pfp_lcp::process  include/direct_gsacak.hpp:50-116,
 *                                                        include/pfp_lcp_mum.hpp:115-231
 *   match scan            mem_finder::update             include/mem_finder.hpp:161-170, :304-355
 *   writers               write_mum / write_mem / close  include/mem_finder.hpp:104-158, :210-263, :357-428
 *   partition merge       merge_partitions               src/merge_candidates.cpp:97-157
 */
#ifndef MUMEMTO_GPU_H
#define MUMEMTO_GPU_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) || defined(__clang__)
#define MMT_API __attribute__((visibility("default")))
#else
#define MMT_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmt_engine mmt_engine;

/* Scan predicates, named as in mem_finder's constructor (mem_finder.hpp:53).
 * CLI flag normalisation (pfp_mum.hpp:149-198) is the caller's job.          */
typedef struct mmt_params {
    uint32_t min_match_len;   /* -l, default 20                               */
    uint64_t num_distinct;    /* matches must occur in >= this many docs; 0 = all */
    int64_t  max_doc_freq;    /* 1 = multi-MUM mode; 0 = unlimited            */
    int64_t  max_total_freq;  /* 0 = no cap                                   */
    uint8_t  use_revcomp;     /* text holds F$ R$ per document                */
    uint8_t  merge_metadata;  /* record candidate thresholds (-M / -n)        */
} mmt_params;

MMT_API const char* mmt_last_error(void);

/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or
 * NULL for a stream owned by the engine.                                      */
MMT_API int  mmt_engine_create(int device, void* hip_stream, mmt_engine** out);
MMT_API void mmt_engine_destroy(mmt_engine* e);

/* Input = concatenated raw forward bases of all documents (records of one
 * document concatenated, no separators; any case) + per-document lengths.
 * _device: d_bases is HBM memory borrowed until the next set_input/destroy.   */
MMT_API int mmt_engine_set_input_device(mmt_engine* e, const uint8_t* d_bases,
                                        const uint64_t* doc_len, size_t n_docs);
MMT_API int mmt_engine_set_input_host(mmt_engine* e, const uint8_t* h_bases,
                                      const uint64_t* doc_len, size_t n_docs);

/* Producer of the SA/LCP/BWT stream for the next runs:
 *   0 automatic | 1 direct suffix sort of the whole text (the reference's -g path, direct_gsacak.hpp)
 *   2 prefix-free parsing (the reference's default: newscan.hpp + pfp.hpp + pfp_lcp_mum.hpp)
 *   3 prefix-free parsing without the suffix array of the dictionary (guided.cpp: the text suffixes are sorted by their
 *     characters up to the phrase end, then by the parse; what 0 / 2 fall back to when the dictionary -- as large as
 *     half the text for two unrelated strands -- would not fit);
 *   4 the same with expansion: only one representative per (distinct phrase, offset) is sorted -- the valid suffixes of
 *     the dictionary -- and the emitter of 2 expands each by the inverted list of its phrase (pfp_lcp_mum.hpp:151-212);
 *     what 0 takes instead of 3 when the collection is redundant (a rank's share of whole genomes).
 *     mmt_producer_used reports 3 for both, mmt_producer_expanded says which;
 * w / p = PFP window and modulus (0 = chosen by the size of the text, as for the automatic producer; the reference's
 * defaults are 10 / 100).  The stream does not depend on them. */
MMT_API int mmt_engine_set_producer(mmt_engine* e, int kind, uint32_t w, uint32_t p);
/* Version of this device-resident ABI (the drop-in ABI of mumemto.h never changes).  Changes under an unchanged symbol name
 * bump it: 4 = mmt_merged_device hands out the 32-bit merged thresholds (uint16_t before) and mmt_partition gained
 * thresh_bits in former padding; 5 = producer 4 (expansion), mmt_producer_expanded, mmt_engine_set_text_sink keeps rows on
 * request.  A caller built against an older header checks this before it trusts the layout.                            */
MMT_API int mmt_abi_version(void);
MMT_API int mmt_producer_used(const mmt_engine* e);
MMT_API int mmt_producer_expanded(const mmt_engine* e);
/* Of the last run through the bucket-wise producer: out[0] = slices that bins of one repeated symbol (assembly gaps) were
 * produced in, out[1] = passes over the text that collected suffixes, out[2] = batches, out[3] = 1 when several batches
 * shared a pass (staging list).  Zeros for the other producers.                                                          */
MMT_API int mmt_producer_stats(const mmt_engine* e, uint64_t out[4]);

/* The row tap (test instrument of full-size runs that keep nothing else: tests/bigchecks.py check_bins_complete): every
 * accepted interval of the NEXT runs whose match begins with one of the n k-mers (n x k bytes, k <= 16) leaves a copy of its
 * length and of ALL its suffix-array entries (text positions) before its window drops it; n = 0 switches it off.
 * mmt_row_tap_counts: out[0] rows, out[1] entries tapped by the last run; mmt_row_tap_get: length[rows], occ_start[rows + 1],
 * sa[entries] (rows in no particular order; rc 3 when the capacities given here were exceeded).
 * mmt_kmer_positions: every position of the resident text whose suffix begins with one of the k-mers, ascending, with the
 * index of its k-mer; *found may exceed cap (only cap were written).  Together they give precision AND recall inside whole
 * bins of leading characters: the host sorts those suffixes, runs the oracle's scan over them and compares.                */
/* (bytes, digest) of what the last run's text sink wrote, in file order -- kept when the sink's path is "/dev/null" (the bytes
 * are formatted, copied out, digested and dropped: a full-size test run whose rows nobody can keep) or MMT_SINK_DIGEST is set */
MMT_API int mmt_text_sink_digest(const mmt_engine* e, uint64_t out[2]);
/* 1 when the suffixes that begin with this k-mer belong to the share of the stream the last run of the bucket-wise producer
 * produced (mmt_engine_set_scan_shard: whole bins of leading characters), 0 when not, -1 when that producer did not run */
MMT_API int mmt_kmer_in_share(const mmt_engine* e, const uint8_t* kmer, size_t k);
MMT_API int mmt_engine_set_row_tap(mmt_engine* e, const uint8_t* kmers, size_t n, size_t k, size_t max_rows, size_t max_occ);
MMT_API int mmt_row_tap_counts(mmt_engine* e, uint64_t out[2]);
MMT_API int mmt_row_tap_get(mmt_engine* e, uint32_t* length, uint64_t* occ_start, uint64_t* sa);
MMT_API int mmt_kmer_positions(mmt_engine* e, const uint8_t* kmers, size_t n, size_t k, uint64_t* pos, uint32_t* which,
                               uint64_t cap, uint64_t* found);

/* One pass of the hot path: text -> SA/LCP/BWT -> scan -> rows (+ thresholds). */
/* Stage checkpoints of the reference CLI (src/pfp_mum.cpp:97-111 `-a`, :122-124 `-p`): hand over the text T itself
 * (UPPER(F) '$' [revcomp(F) '$'] per document, n characters) or the stream of the real suffixes (sentinel entry
 * dropped; may be a prefix of the full stream).  mmt_engine_run then skips the stages that would have made them.  */
MMT_API int mmt_engine_set_text_host(mmt_engine* e, const uint8_t* text, uint64_t n, const uint64_t* doc_len,
                                     size_t n_docs, int use_revcomp);
MMT_API int mmt_engine_set_stream_host(mmt_engine* e, const uint32_t* sa, const uint32_t* lcp, const uint8_t* bwt,
                                       uint64_t entries, const uint64_t* doc_len, size_t n_docs, int use_revcomp);
MMT_API int mmt_engine_run(mmt_engine* e, const mmt_params* p);

/* The same job for host-resident input of any size.  When the text would exceed max_text_chars
 * (0 = the limit of one suffix array in this build, 2^32 - 4097 characters) the documents are
 * processed as partitions that share document 0 and merged like `anchor_merge` does
 * (README.md:124-141 of the reference); strict multi-MUMs only, byte-identical to a direct run.   */
MMT_API int mmt_engine_run_partitioned(mmt_engine* e, const uint8_t* h_bases, const uint64_t* doc_len,
                                       size_t n_docs, const mmt_params* p, uint64_t max_text_chars);
MMT_API size_t mmt_partitions_used(const mmt_engine* e);
/* The same job with the documents SUPPLIED one at a time instead of resident on the host: supplier(user, d, dst, doc_len[d])
 * writes the bases of document d to dst and returns 0 (non-zero aborts the run).  It is asked for the documents in order
 * (for all of them a second time if the text has to be rebuilt).  The reference streams its FASTA files through the parser
 * and never holds the collection (include/newscan.hpp:265-325); this entry is for collections that do not fit the host as
 * bytes either (BASELINE configs[4]: 287 GB of bases).  Always ONE text: no anchor partitions.                              */
typedef int (*mmt_doc_supplier)(void* user, uint64_t doc, uint8_t* dst, uint64_t len);
MMT_API int mmt_engine_run_supplied(mmt_engine* e, mmt_doc_supplier supplier, void* user, const uint64_t* doc_len,
                                    size_t n_docs, const mmt_params* p);
/* build_main in-process (src/pfp_mum.cpp:31-159): FASTA / FASTQ(.gz) files in (one document per file, read on all
 * host cores), PREFIX.mums | PREFIX.mems and PREFIX.lengths out (out_prefix NULL: nothing is written, the rows stay
 * available through the accessors below).  seconds (optional): [0] reading + parsing the files, [1] H2D + the whole
 * GPU path, [2] writing the outputs, [3] total.  mmt_params.merge_metadata is honoured as in mmt_engine_run.         */
MMT_API int mmt_engine_run_files(mmt_engine* e, const char* const* paths, size_t n_paths, const mmt_params* p,
                                 const char* out_prefix, uint64_t max_text_chars, double seconds[4]);
/* merged PREFIX.athresh (L_0 + 1 entries) after a partitioned run                                  */
MMT_API int mmt_copy_merged_thresh(const mmt_engine* e, uint16_t* out);

/* ---- results of the last run ------------------------------------------------
 * A run leaves its rows and the file bytes in HBM; each accessor below downloads what it
 * returns on first use (page-locked host memory owned by the engine, valid until the next run). */
MMT_API size_t mmt_num_rows(const mmt_engine* e);
MMT_API size_t mmt_num_docs(const mmt_engine* e);
/* MUM mode: length[n_rows], offsets[n_rows*n_docs] (-1 absent), strands (1 '+') */
MMT_API int mmt_rows_mum(const mmt_engine* e, uint32_t* length, int64_t* offsets, uint8_t* strands);
/* the same three tables where the run left them in HBM (valid until the next run): what a
 * multi-GPU exchange hands to RCCL without a host round trip                    */
MMT_API int mmt_rows_mum_device(const mmt_engine* e, const uint32_t** length, const int64_t** offsets,
                                const uint8_t** strands);
/* MEM mode: occ_start[n_rows+1]; flat offsets / doc ids / strands              */
MMT_API size_t mmt_num_occ(const mmt_engine* e);
MMT_API int mmt_rows_mem(const mmt_engine* e, uint32_t* length, uint64_t* occ_start,
                         int64_t* offsets, uint64_t* seq_ids, uint8_t* strands);
/* PREFIX.mums / PREFIX.mems bytes exactly as the CLI writes them.              */
MMT_API const char* mmt_output_text(mmt_engine* e, size_t* len);
/* PREFIX.bumbl bytes (MUM mode).                                               */
MMT_API const uint8_t* mmt_output_bumbl(mmt_engine* e, size_t* len);
/* candidate_thresh (u16, 2*(L_0+1) entries) when merge_metadata was set.       */
MMT_API size_t mmt_thresh_len(const mmt_engine* e);
MMT_API int mmt_copy_thresh(const mmt_engine* e, uint16_t* out);
/* Device pointers of the last run's thresholds / anchor ISA (for the merge).   */
MMT_API const uint16_t* mmt_thresh_device(const mmt_engine* e);
/* The engine's own threshold column: 32 bits per entry, never saturated (the 16-bit forms above are made from it and
 * saturate at 65535 like the reference's, include/mem_finder.hpp:299,328).  What partitions hand to the fold with
 * mmt_partition.thresh_bits = 32 and what the multi-GPU exchange carries. */
MMT_API int mmt_copy_thresh32(const mmt_engine* e, uint32_t* out);
MMT_API const uint32_t* mmt_thresh_device32(const mmt_engine* e);

/* ---- stage introspection (parity tests, bench roofline) ------------------- */
MMT_API uint64_t mmt_text_length(const mmt_engine* e);
MMT_API int mmt_copy_text(const mmt_engine* e, uint8_t* out);   /* n bytes    */
MMT_API int mmt_copy_sa(const mmt_engine* e, uint32_t* out);    /* n entries, real suffixes only */
MMT_API int mmt_copy_lcp(const mmt_engine* e, uint32_t* out);   /* lcp[0] = 0 */
MMT_API int mmt_copy_bwt(const mmt_engine* e, uint8_t* out);
/* Candidate lists of the last run, each entry {start, end, length, flags}
 * in real-suffix index space (= reference stream index - 1).                  */
MMT_API size_t mmt_num_candidates(const mmt_engine* e);
MMT_API int mmt_copy_candidates(const mmt_engine* e, uint32_t* out);
/* HIP-event times (ms) of the last run:
 * [0] text build  [1] suffix sort  [2] LCP+BWT  [3] scan kernel (roofline kernel)
 * [4] candidate verification + thresholds  [5] row gather + D2H  [6] the windows of the stream (emitter /
 * bucket batches; part of [1])  [7] whole run (host clock).                     */
MMT_API int mmt_stage_ms(const mmt_engine* e, float out[8]);
/* Bytes of the SA / LCP / BWT columns as stored (for the roofline model).     */
MMT_API int mmt_column_bytes(const mmt_engine* e, uint32_t out[3]);
/* Texts of 2^32 - 4096 characters or more run with 40-bit positions (the reference's width: include/common.hpp:59-60,
 * include/parse.hpp:45; dumps include/pfp_lcp_mum.hpp:323-369): the suffix-array column is stored as low word + high
 * byte, and the stream is scanned in ranges of suffix-array positions (mmt_scan_ranges of them in the last run).     */
MMT_API int mmt_is_wide(const mmt_engine* e);
MMT_API size_t mmt_scan_ranges(const mmt_engine* e);
MMT_API int mmt_copy_sa64(const mmt_engine* e, uint64_t* out);    /* n entries, any text size */
/* mmt_engine_set_stream_host for streams with 40-bit suffix-array entries (sa_hi = bits 32..39)                        */
MMT_API int mmt_engine_set_stream_host40(mmt_engine* e, const uint32_t* sa_lo, const uint8_t* sa_hi, const uint32_t* lcp,
                                         const uint8_t* bwt, uint64_t entries, const uint64_t* doc_len, size_t n_docs,
                                         int use_revcomp);
/* Device heap of the engine's GPU (pool.hpp): [0] bytes mapped from the driver, [1] bytes in use, [2] high-water mark
 * of [1], [3] microseconds spent in the driver mapping memory.                                                         */
MMT_API int mmt_device_memory(const mmt_engine* e, uint64_t out[4]);
/* Multi-GPU runs of partial multi-MUMs / multi-MEMs, which the anchor merge cannot serve (the reference refuses to
 * merge them: include/pfp_mum.hpp:178-183): every rank builds the tables of the parse and then produces, scans and drops
 * ONLY ITS SHARE of the stream -- a range of suffix-array positions cut at multiples of 4096 (the parse proper), or whole
 * bins of leading characters (the bucket-wise producer: SURVEY.md 8(e), every reportable interval lies inside one bin).
 * No column is exchanged; the outputs of ranks 0 .. count-1, concatenated, are byte for byte the output of one GPU
 * (mmt_dist_gather_text, mumemto_amd/dist.py::run_sharded).  count = 1 switches it off.  mmt_sort_pieces: first entry and
 * number of entries of every rank's share (after a run).                                                              */
MMT_API int mmt_engine_set_scan_shard(mmt_engine* e, uint32_t index, uint32_t count);
MMT_API size_t mmt_sort_pieces(const mmt_engine* e, uint64_t* first, uint64_t* count, size_t capacity);
/* The columns of the stream exist one window at a time (the reference does not store them either:
 * include/pfp_lcp_mum.hpp:197).  on = 1: every window is also copied into whole columns, so that mmt_copy_sa / _lcp /
 * _bwt work after the run (tests, stage dumps); 0: never; -1 (default): for texts below 2^26 characters.               */
MMT_API int mmt_engine_keep_columns(mmt_engine* e, int on);
MMT_API int mmt_columns_kept(const mmt_engine* e);
/* Of the last run: [0] entries of the stream this engine produced (its share of a sharded run + the left extensions of its
 * windows), [1] bytes of the window buffers that held them (high-water mark: independent of the text length), [2] number
 * of windows, [3] bytes of the suffix-array entries that left the windows with accepted rows.                          */
MMT_API int mmt_stream_stats(const mmt_engine* e, uint64_t out[4]);
/* Returns the heap's physical memory to the driver when no engine buffer is live (long-lived hosts between jobs).  The engine
   the mumemto_library entry points (mumemto.h) share is let go first -- the next such call makes a new one --; engines made
   with mmt_engine_create are the caller's to destroy.                                                                      */
MMT_API void mmt_pool_trim(void);
/* bytes of device memory the heap leaves alone from now on (what MUMEMTO_HEAP_RESERVE sets for a whole process); ~0ull: back to
   the environment's value.  The estimates (automatic text limit, batch sizes, packing) see a device that much smaller. */
MMT_API void mmt_pool_set_reserve(unsigned long long bytes);
/* Text, window buffers and sort scratch of the last run go back to the device heap (downloaded results stay).
 * keep_anchor_ranks != 0: the suffix ranks of the anchor stay, so that mmt_merged_sort_like_direct still works -- the state
 * of a rank between its own pass and the fold of everybody's rows. */
MMT_API int mmt_engine_release_columns(mmt_engine* e, int keep_anchor_ranks);
/* PREFIX.mums / PREFIX.mems of the NEXT runs straight to `path` while a run goes on (NULL or "": off): the rows of a window of
 * the stream are final when it has been verified, so their bytes are formatted and written while the later windows are
 * produced (src/pfp_mum.cpp writes the file row by row as well, include/mem_finder.hpp:357-428).  A run over a text that
 * fills the device (packed text, or 2^37 characters and more; MMT_SINK_DISCARD=0/1 overrides) then keeps nothing of a
 * window's rows: it answers for the file and for mmt_num_rows only -- mmt_rows_* / mmt_output_text hold nothing. */
MMT_API int mmt_engine_set_text_sink(mmt_engine* e, const char* path);

/* ---- PFP stage checkpoints (the reference's -P / -K: PREFIX.dict, PREFIX.parse) ------------ */
/* Text layout + prefix-free parse only (newscan.hpp pfparser: process_string ... finish_parse).  */
MMT_API int mmt_engine_parse_only(mmt_engine* e, uint8_t use_revcomp, uint32_t w, uint32_t p);
/* out[0] #phrases of the parse, [1] #distinct phrases, [2] dictionary bytes, [3] #groups of equal
 * proper phrase suffixes, [4] doubling rounds on the dictionary, [5] on the parse, [6] #valid
 * dictionary suffixes, [7] #groups too large for an LDS tile (sorted by the segmented fallback)   */
MMT_API int mmt_pfp_counts(const mmt_engine* e, uint64_t out[8]);
/* dictionary suffixes of the last run that were ordered by the long run of one symbol they begin in (runs of assembly
   gaps, homopolymers: sorter.hpp, RunRefine) instead of by prefix doubling; -1: null engine */
MMT_API long long mmt_pfp_run_refined(const mmt_engine* e);
/* PREFIX.dict bytes (sorted phrases, 0x01 after each, 0x00 at the end; out holds counts[2] bytes)  */
MMT_API int mmt_pfp_copy_dict(mmt_engine* e, uint8_t* out);
/* PREFIX.parse entries (1-based phrase ranks, u32; out holds counts[0] entries)                    */
MMT_API int mmt_pfp_copy_parse(mmt_engine* e, uint32_t* out);
/* ms: [0] triggers+phrases [1] distinct phrases [2] dictionary text [3] dictionary SA
 * [4] dictionary LCP + groups + ranks [5] parse SA [6] text keys + sort [7] total (host clock)    */
MMT_API int mmt_pfp_stage_ms(const mmt_engine* e, float out[8]);

/* ---- anchor partition merge (src/merge_candidates.cpp:97-157) -------------- */
/* ZERO-INITIALISE this struct (`mmt_partition p = {0};` / `memset`): thresh_bits occupies what was padding before round 4,
 * and the merge entry points reject any value other than 0, 16 or 32 (rc 3).                                        */
typedef struct mmt_partition {
    uint64_t n_rows, n_docs;
    const uint32_t* length;   /* host or device (see rows_on_device)           */
    const int64_t*  offsets;  /* n_rows * n_docs, column 0 = anchor            */
    const uint8_t*  strands;  /* 1 = '+'                                       */
    const uint16_t* thresh;   /* host or device (see thresh_on_device), L_0+1 entries; uint32_t entries behind the same
                                 pointer when thresh_bits == 32 */
    uint64_t thresh_len;
    uint8_t thresh_on_device;
    uint8_t rows_on_device;   /* length / offsets / strands are HBM pointers   */
    uint8_t thresh_bits;      /* 0 or 16: the reference's PREFIX.athresh width (saturated at 65535, mem_finder.hpp:299);
                                 32: the engine's own width (mmt_copy_thresh32 / mmt_thresh_device32), never saturated --
                                 what the multi-GPU exchange carries */
} mmt_partition;
typedef struct mmt_merged mmt_merged;
/* Left fold over parts[0..k) exactly as anchor_merge does; the per-position
 * whole fold runs on the GPU of `e` and the merged rows stay in its HBM until
 * mmt_merged_get / mmt_merged_text copy them out.                              */
MMT_API int mmt_anchor_merge(mmt_engine* e, const mmt_partition* parts, size_t k, mmt_merged** out);
/* The same with the minimum length of a merged MUM stated.  The reference's tool hard-codes 20
 * (src/merge_candidates.cpp:141), which equals a direct run only for partitions made with -l 20: a multi-GPU run with
 * another -l passes that value here.                                                                                 */
MMT_API int mmt_anchor_merge_min_len(mmt_engine* e, const mmt_partition* parts, size_t k, uint32_t min_len,
                                     mmt_merged** out);
/* The same table computed as `slices` independent slices of the anchor: slice [lo, hi) folds from the rows
 * that start in [lo - margin, hi) and the thresholds of that range, margin = (k - 1) x the longest row + 1.  On one engine
 * the slices run one after the other; mmt_dist_merge_ranges gives every rank of a communicator its own.               */
MMT_API int mmt_anchor_merge_by_ranges(mmt_engine* e, const mmt_partition* parts, size_t k, int slices, uint32_t min_len,
                                       mmt_merged** out);
/* The arithmetic of that fold, on the host (no device needed): slice r of `world` is [bounds[0], bounds[1]) and folds from
 * [bounds[2], bounds[1]), for an anchor of thresh_len = L_0 + 1 entries, k partitions and a longest row of `longest`.    */
MMT_API int mmt_fold_slice_bounds(uint64_t thresh_len, int world, int r, size_t k, uint32_t longest, uint64_t bounds[3]);
MMT_API size_t mmt_merged_rows(const mmt_merged* m);
MMT_API size_t mmt_merged_docs(const mmt_merged* m);
MMT_API int mmt_merged_get(mmt_merged* m, uint32_t* length, int64_t* offsets, uint8_t* strands,
                           uint16_t* thresh);
/* the merged tables in HBM (owned by m); thresholds at the engine's width, 32 bits  */
MMT_API int mmt_merged_device(const mmt_merged* m, const uint32_t** length, const int64_t** offsets,
                              const uint8_t** strands, const uint32_t** thresh);
/* Rows folded elsewhere (a coordinate-range fold: every rank folds its slice of the anchor) as a merged
 * result of this engine: host arrays in, HBM tables out, so that mmt_merged_sort_like_direct / mmt_merged_text apply.  */
MMT_API int mmt_merged_from_rows(mmt_engine* e, const uint32_t* length, const int64_t* offsets, const uint8_t* strands,
                                 size_t n_rows, size_t n_docs, const uint16_t* thresh, size_t thresh_len,
                                 mmt_merged** out);
/* Re-order merged rows into the order of a direct run (lexicographic by match
 * string) using the anchor suffix ranks of the engine's last run, whose
 * document 0 must be the anchor.                              */
MMT_API int mmt_merged_sort_like_direct(mmt_engine* e, mmt_merged* m);
MMT_API const char* mmt_merged_text(mmt_merged* m, size_t* len);
/* PREFIX.mums straight from the library (no copy of the bytes through the caller) */
MMT_API int mmt_merged_write_text(mmt_merged* m, const char* path);
MMT_API void mmt_merged_free(mmt_merged* m);

/* ---- multi-GPU exchange, one process per GPU (RCCL over xGMI) ------------------------------
 * Replaces the files + second tool between partitions of the reference's workflow (README.md:124-141: PREFIX.mums and
 * PREFIX.athresh per partition, then `anchor_merge`, src/merge_candidates.cpp:170-255; Python side
 * mumemto/merge_mums.py:141-183).  Rank 0 makes a unique id and hands it to the other ranks out of band (a file, an
 * environment variable, MPI, torch's store); every call below is collective over the communicator.  RCCL is bound at run
 * time (a copy already in the process, else /opt/rocm/lib/librccl.so).                                              */
typedef struct mmt_comm mmt_comm;
MMT_API int  mmt_comm_unique_id(uint8_t id[128]);
MMT_API int  mmt_comm_create(mmt_engine* e, int rank, int world, const uint8_t id[128], mmt_comm** out);
MMT_API void mmt_comm_destroy(mmt_comm* c);
/* Strict multi-MUMs: e's last run = this rank's partition {anchor} + its documents with merge metadata on.  Row tables
 * and thresholds of ranks 1 .. world-1 travel HBM -> HBM to rank 0 (ncclSend / ncclRecv, one group), rank 0 folds them on its GPU and
 * re-sorts into direct-run order: *out is the merged result on rank 0 (mmt_merged_text / _get / _free) and NULL on the
 * other ranks.  min_len = the run's -l (the reference's tool hard-codes 20).                                          */
MMT_API int  mmt_dist_merge(mmt_comm* c, mmt_engine* e, uint32_t min_len, mmt_merged** out);
/* The same result with the fold itself spread over the ranks (dist.cpp dist_merge_ranges): rows are broadcast, of the
 * thresholds -- 2 bytes per anchor position and rank -- every rank receives only its slice of the anchor from every other
 * rank (all-to-all), folds it, and sends its piece to rank 0.  mmt_dist_merge does this by itself from four ranks on
 * (MUMEMTO_RANGE_FOLD=0 / 1 overrides).                                                                                */
MMT_API int  mmt_dist_merge_ranges(mmt_comm* c, mmt_engine* e, uint32_t min_len, mmt_merged** out);
/* Modes without a partition merge (mmt_engine_set_scan_shard): the ranks' output bytes, concatenated in rank order, on
 * rank 0 (*len = 0 elsewhere); valid until the next call on this communicator.                                        */
MMT_API int  mmt_dist_gather_text(mmt_comm* c, const char** text, size_t* len);
/* Self-test of the exchange's message machinery with this rank as its own peer (dist.cpp dist_loopback): the row tables and
 * the 32-bit threshold column of the engine's last run (merge metadata on) through ncclSend / ncclRecv in one group, in pieces
 * of at most 2^30 elements, an all-gather and a broadcast; out = bytes moved, pieces, largest piece (bytes), elements that
 * arrived different (0 is the answer), microseconds, rows, row cells, thresholds.                                         */
MMT_API int  mmt_comm_loopback(mmt_comm* c, uint64_t out[8]);
/* One message of `elements` elements of `width` bytes (1, 4, 8) to this rank itself through ncclSend / ncclRecv in the pieces the
 * exchange cuts (MUMEMTO_RCCL_CHUNK): out = elements that arrived different, pieces, largest piece (bytes), microseconds.      */
MMT_API int  mmt_comm_selftest(mmt_comm* c, uint64_t elements, uint32_t width, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* MUMEMTO_GPU_H */
