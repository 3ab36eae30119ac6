/* mumemto_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of vikshiv/mumemto's hot path
 * (text layout -> SA/LCP/BWT stream -> LCP-interval match scan -> writers ->
 * anchor merge).  It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have an independent checker for the HIP path.
 * Nothing under mumemto_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  Parity pin status (see DESIGN.md "Oracle"):
 *   - scan/writers: pinned by the known-answer vectors of SURVEY.md 8(c)
 *     (tests/golden/toy_vectors.json) and by a brute-force definition checker
 *     (tests/bruteforce.py); the reference's mem_finder.hpp cannot be built
 *     here (needs sdsl + gsacak, un-vendored).
 *   - anchor merge: pinned against the real reference build
 *     oracle/_ref/anchor_merge (src/merge_candidates.cpp compiles stand-alone).
 *   - PFP parse/dictionary: pinned against oracle/_ref/newscan_ref
 *     (include/newscan.hpp compiles stand-alone).
 */
#ifndef MUMEMTO_ORACLE_H
#define MUMEMTO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- text layout (src/ref_builder.cpp:211-314, 330-384) ------------------ */
/* Length of the text for docs of L_i bases: sum (revcomp ? 2 : 1) * (L_i+1). */
int64_t mmo_text_length(const int64_t *doc_len, int64_t n_docs, int revcomp);
/* bases = concatenated raw forward bases of every doc (any case).  Writes T
 * (upper-cased F_i '$' [revcomp(F_i) '$']) and doc_start[n_docs+1]. */
void mmo_build_text(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs,
                    int revcomp, uint8_t *text, int64_t *doc_start);

/* ---- SA / LCP / BWT stream (include/direct_gsacak.hpp:50-116 restated with
 * one unique smallest end sentinel == the PFP stream of
 * include/pfp_lcp_mum.hpp:115-231) ------------------------- */
/* Stream has n+1 entries, j = 0 is the sentinel suffix (sa = n).            */
int mmo_build_stream(const uint8_t *text, int64_t n, int64_t *sa, int64_t *lcp,
                     uint8_t *bwt);
/* The same stream by way of the prefix-free parse, the reference's default route
 * (include/newscan.hpp, dictionary.hpp, parse.hpp, pfp.hpp, pfp_lcp_mum.hpp:115-231;
 * window w, modulus p; text bytes must be > 2).  stats (optional, 4 entries):
 * phrases, distinct phrases, dictionary bytes, stream entries.  0 on success.   */
int mmo_build_stream_pfp(const uint8_t *text, int64_t n, int64_t w, int64_t pmod, int64_t *sa,
                         int64_t *lcp, uint8_t *bwt, int64_t *stats);
/* doc[j] = #doc ends in [0, sa[j])  (pfp_lcp_mum.hpp:194)                    */
void mmo_doc_array(const int64_t *sa, int64_t m, const int64_t *doc_start,
                   int64_t n_docs, int32_t *doc);

/* ---- match scan (include/mem_finder.hpp:161-170, 304-355, 265-289) ------- */
typedef struct mmo_scan_params {
    int64_t min_len;        /* min_mem_length                               */
    int64_t num_distinct;   /* num_distinct                                 */
    int64_t max_doc_freq;   /* 1 => MUM mode; 0 => unlimited                */
    int64_t max_total_freq; /* 0 => no cap (no_max_freq)                    */
    int32_t revcomp;
    int32_t merge;          /* record candidate thresholds (-M / -n)        */
} mmo_scan_params;

typedef struct mmo_result mmo_result;

/* Runs the stack scan over the whole stream.  Result owns its storage. */
mmo_result *mmo_scan(const int64_t *sa, const int64_t *lcp, const uint8_t *bwt,
                     const int32_t *doc, int64_t m, const int64_t *doc_start,
                     int64_t n_docs, const mmo_scan_params *p);
void mmo_result_free(mmo_result *r);

/* Candidates that passed every predicate incl. the BWT test, in pop order,
 * BEFORE the writer-side drops of write_mum (mem_finder.hpp:372-391).       */
int64_t mmo_num_intervals(const mmo_result *r);
/* out[4*i..] = {start j, end j, length, closing j} in stream index space.   */
void mmo_get_intervals(const mmo_result *r, int64_t *out);
/* Candidates accepted before the BWT test (threshold recorders), same layout */
int64_t mmo_num_accepted(const mmo_result *r);
void mmo_get_accepted(const mmo_result *r, int64_t *out);

/* Rows as the library collectors see them (mumemto_api.cpp:137-166,241-286). */
int64_t mmo_num_rows(const mmo_result *r);
/* MUM mode: offsets[n_rows*n_docs] (-1 absent), strands u8 (1='+', absent 0) */
void mmo_get_mum_rows(const mmo_result *r, uint32_t *length, int64_t *offsets,
                      uint8_t *strands);
/* MEM mode: occ_start[n_rows+1] prefix sums; flat offsets/docs/strands.
 * The last '-' occurrence of a row uses the "- length" (no -1) rule of
 * mem_finder.hpp:248 / mumemto_api.cpp:152-155.                             */
int64_t mmo_num_occ(const mmo_result *r);
void mmo_get_mem_rows(const mmo_result *r, uint32_t *length, int64_t *occ_start,
                      int64_t *offsets, int64_t *docs, uint8_t *strands);
/* .mums / .mems text exactly as the CLI writes it (mem_finder.hpp:210-263,
 * 357-428).  Returns a malloc'd buffer (caller frees with mmo_free).        */
char *mmo_format_text(const mmo_result *r, int64_t *out_len);
/* .bumbl bytes (mem_finder.hpp:460-503)                                      */
uint8_t *mmo_format_bumbl(const mmo_result *r, int64_t *out_len);
/* candidate_thresh, size 2*(L_0+1) u16 (mem_finder.hpp:75-77,326-336).       */
int64_t mmo_thresh_len(const mmo_result *r);
void mmo_get_thresh(const mmo_result *r, uint16_t *out);
/* .thresh / .thresh_rev contents (mem_finder.hpp:116-157). which: 0 fwd 1 rev */
uint16_t *mmo_format_thresh(const mmo_result *r, int which, int64_t *out_len);
void mmo_free(void *p);

/* ---- anchor merge (src/merge_candidates.cpp:97-157) ---------------------- */
/* One partition: rows sorted by offsets[0]; thresholds nb[nb_len].           */
typedef struct mmo_partition {
    int64_t n_rows, n_docs;
    const uint32_t *length;
    const int64_t *offsets; /* n_rows * n_docs */
    const uint8_t *strands; /* n_rows * n_docs, 1 = '+' */
    const uint16_t *nb;
    int64_t nb_len;
} mmo_partition;
typedef struct mmo_merged mmo_merged;
/* Pairwise fold over parts[0..k) exactly like anchor_merge's main loop
 * (merge_candidates.cpp:208-219).  Rows of each part are sorted by
 * offsets[0] first (parse_candidate, :89-92).                               */
mmo_merged *mmo_anchor_merge(const mmo_partition *parts, int64_t k);
int64_t mmo_merged_rows(const mmo_merged *m);
int64_t mmo_merged_docs(const mmo_merged *m);
void mmo_merged_get(const mmo_merged *m, uint32_t *length, int64_t *offsets,
                    uint8_t *strands, uint16_t *nb);
void mmo_merged_free(mmo_merged *m);

/* ---- whole job, timed pieces (bench.py cpu_baseline) --------------------- */
/* Runs text -> stream -> scan -> .mums text; returns text length processed
 * and fills seconds per stage {text, sa+lcp+bwt, scan+format}.              */
int64_t mmo_run_job(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs,
                    const mmo_scan_params *p, double *stage_sec, char **out_text,
                    int64_t *out_len);
/* the same with the stream produced through the prefix-free parse (window pfp_w, modulus pfp_p) */
int64_t mmo_run_job_pfp(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs,
                        const mmo_scan_params *p, int64_t pfp_w, int64_t pfp_p, double *stage_sec,
                        char **out_text, int64_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
