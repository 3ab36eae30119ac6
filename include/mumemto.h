/* mumemto.h -- C ABI of libmumemto (MI355X-native build).
 *
 * Drop-in for the reference's mumemto_library/mumemto.h:33-94: the same 15
 * exported symbols with the same signatures, ownership and error codes, so a
 * C / FFI caller of the reference library links against this one unchanged.
 * Behind it the whole hot path (text layout -> suffix array / LCP / BWT ->
 * LCP-interval match scan) runs as hand-written HIP kernels on gfx950; there
 * is no CPU fallback -- when no GPU is usable the calls return rc 3 and
 * mumemto_last_error() says why.
 *
 * Contract restated from the reference implementation
 * (mumemto_library/mumemto_api.cpp:489-644):
 *   rc 0 ok | 1 out_result == NULL | 2 docs == NULL && num_docs != 0 |
 *   3 failure with message | 4 unknown failure.          (:503-536, :549-584)
 *   *out_result is set to NULL first.                      (:507, :553)
 *   num_distinct == 0 means "all documents".               (:344-346)
 *   mumemto_mem requires max_doc_freq > 1 (else rc 3).     (:381-383)
 *   empty docs -> empty result, rc 0.                      (:338-340)
 *   A NULL record is an empty string.                      (:450-465)
 *   use_gsacak selects the reference's alternative SA producer; both give the
 *   same matches, so it is accepted and ignored here.
 *   Views alias storage owned by the handle until mum_free / mem_free;
 *   accessors tolerate NULL handles and out-of-range indices. (:587-642)
 *   Matches come out in suffix-array (lexicographic) order of the match.
 */
#ifndef MUMEMTO_H            /* the reference header's guard on purpose: the two headers declare the same ABI */
#define MUMEMTO_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__) || defined(__clang__)
#define MUMEMTO_EXPORT __attribute__((visibility("default")))
#else
#define MUMEMTO_EXPORT
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- inputs ------------------------------------------------------------------------------------------------- */
/* One document = the records of one FASTA file; NUL-terminated strings borrowed for the call (reference :33-36). */
typedef struct mumemto_doc_view { const char* const* records; size_t num_records; } mumemto_doc_view;

/* ---- result handles and row views ----------------------------------------------------------------------------- */
typedef struct mumemto_mum_result mumemto_mum_result;        /* opaque, owned by the caller until mum_free (:38) */
typedef struct mumemto_mem_result mumemto_mem_result;        /* opaque, owned by the caller until mem_free (:39) */

typedef struct mumemto_mum_match_view {                       /* one multi-MUM row (:41-45)                       */
    uint32_t length;
    const int64_t* offsets;      /* num_docs entries, -1 = document absent           */
    const uint8_t* strands;      /* 1 = '+', 0 = '-' (0 for absent documents)        */
} mumemto_mum_match_view;

typedef struct mumemto_mem_match_view {                       /* one multi-MEM row (:47-53)                       */
    uint32_t length;
    size_t occurrences;          /* entries of the three arrays below                */
    const int64_t* offsets;
    const size_t* seq_ids;
    const uint8_t* strands;
} mumemto_mem_match_view;

/* ---- searches ------------------------------------------------------------------------------------------------- */
/* multi-MUMs, at most one occurrence per document (:59-66) */
MUMEMTO_EXPORT int mumemto_mum(const mumemto_doc_view* docs, size_t num_docs, uint32_t min_match_len, uint8_t use_revcomp,
                               size_t num_distinct, uint8_t use_gsacak, mumemto_mum_result** out_result);
/* multi-MEMs (:69-78) */
MUMEMTO_EXPORT int mumemto_mem(const mumemto_doc_view* docs, size_t num_docs, uint32_t min_match_len, uint8_t use_revcomp,
                               size_t num_distinct, size_t max_total_freq, size_t max_doc_freq, uint8_t use_gsacak,
                               mumemto_mem_result** out_result);
/* thread-local message of the last non-zero return code (:56) */
MUMEMTO_EXPORT const char* mumemto_last_error(void);

/* ---- accessors, MUM and MEM side by side (:81-86, :89-94) ------------------------------------------------------- */
MUMEMTO_EXPORT size_t num_docs(const mumemto_mum_result* r);
MUMEMTO_EXPORT size_t num_docs_mem(const mumemto_mem_result* r);
MUMEMTO_EXPORT const size_t* doc_record_offsets(const mumemto_mum_result* r);          /* num_docs(r) + 1 entries */
MUMEMTO_EXPORT const size_t* doc_record_offsets_mem(const mumemto_mem_result* r);
MUMEMTO_EXPORT const size_t* record_lengths(const mumemto_mum_result* r);              /* one per record          */
MUMEMTO_EXPORT const size_t* record_lengths_mem(const mumemto_mem_result* r);
MUMEMTO_EXPORT size_t num_mums(const mumemto_mum_result* r);
MUMEMTO_EXPORT size_t num_mems(const mumemto_mem_result* r);
MUMEMTO_EXPORT mumemto_mum_match_view mum_at(const mumemto_mum_result* r, size_t idx);  /* idx out of range: zeros  */
MUMEMTO_EXPORT mumemto_mem_match_view mem_at(const mumemto_mem_result* r, size_t idx);
MUMEMTO_EXPORT void mum_free(mumemto_mum_result* r);
MUMEMTO_EXPORT void mem_free(mumemto_mem_result* r);

#ifdef __cplusplus
}  /* extern "C" */
#endif
#endif /* MUMEMTO_H */
