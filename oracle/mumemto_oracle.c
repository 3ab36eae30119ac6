/* mumemto_oracle.c -- TEST INFRASTRUCTURE ONLY (see mumemto_oracle.h).
 *
 * CPU restatement, in plain C, of the reference hot path.  Written from the
 * behaviour of the cited reference lines; no reference source text is used.
 * Not linked into, nor called by, anything under mumemto_amd/.
 */
#include "mumemto_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* growable arrays                                                           */
/* ------------------------------------------------------------------------ */
typedef struct { int64_t *v; int64_t n, cap; } vec64;
static void v64_push(vec64 *a, int64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (int64_t *)realloc(a->v, (size_t)a->cap * sizeof(int64_t));
    }
    a->v[a->n++] = x;
}
typedef struct { char *v; int64_t n, cap; } vecc;
static void vc_reserve(vecc *a, int64_t extra) {
    if (a->n + extra > a->cap) {
        while (a->n + extra > a->cap) a->cap = a->cap ? a->cap * 2 : 4096;
        a->v = (char *)realloc(a->v, (size_t)a->cap);
    }
}
static void vc_putc(vecc *a, char c) { vc_reserve(a, 1); a->v[a->n++] = c; }
static void vc_putu(vecc *a, uint64_t x) {
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + x % 10); x /= 10; } while (x);
    vc_reserve(a, k);
    while (k) a->v[a->n++] = tmp[--k];
}

/* ------------------------------------------------------------------------ */
/* text layout: src/ref_builder.cpp:211-314 (CLI), :330-384 (library)        */
/* ------------------------------------------------------------------------ */
/* Complement rule of the seqtk table the reference embeds
 * (src/ref_builder.cpp:29-38): IUPAC-aware, letters only, case preserved.   */
static uint8_t complement_of(uint8_t c) {
    static const char *from = "ABCDGHKMRSTUVWYN";
    static const char *to   = "TVGHCDMKYSAABWRN";
    uint8_t up = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
    int lower = up != c;
    const char *p;
    if (up < 'A' || up > 'Z') return c;
    p = strchr(from, up);
    if (!p) return c; /* E F I J L O P Q X Z map to themselves */
    up = (uint8_t)to[p - from];
    return lower ? (uint8_t)(up + 32) : up;
}
static uint8_t upper_of(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

int64_t mmo_text_length(const int64_t *doc_len, int64_t n_docs, int revcomp) {
    int64_t n = 0, i;
    for (i = 0; i < n_docs; i++) n += (revcomp ? 2 : 1) * (doc_len[i] + 1);
    return n;
}

/* D_i = UPPER(F_i) '$' [ revcomp(UPPER(F_i)) '$' ]
 * (ref_builder.cpp:232-241 upper-casing, :257-268 forward + '$',
 *  :272-291 reverse complement of the records in reverse order + '$';
 *  seq_lengths[i] = 2(L_i+1), :294).                                         */
void mmo_build_text(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs,
                    int revcomp, uint8_t *text, int64_t *doc_start) {
    int64_t d, k, src = 0, dst = 0;
    for (d = 0; d < n_docs; d++) {
        int64_t L = doc_len[d];
        doc_start[d] = dst;
        for (k = 0; k < L; k++) text[dst++] = upper_of(bases[src + k]);
        text[dst++] = '$';
        if (revcomp) {
            for (k = L - 1; k >= 0; k--) text[dst++] = complement_of(upper_of(bases[src + k]));
            text[dst++] = '$';
        }
        src += L;
    }
    doc_start[n_docs] = dst;
}

/* ------------------------------------------------------------------------ */
/* Suffix array by induced sorting.  The reference's -g path calls gsacak    */
/* (include/direct_gsacak.hpp:62), an induced-sorting construction from an   */
/* un-vendored dependency (oma219/gsa-is, unpinned).  A suffix array is      */
/* unique, so any correct construction restates it; this is SA-IS            */
/* (Nong, Zhang, Chan 2009) on an int32 string whose last symbol is a unique */
/* smallest 0.                                                               */
/* ------------------------------------------------------------------------ */
#define TGET(t, i) (((t)[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(t, i, b) do { if (b) (t)[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)); \
                           else (t)[(i) >> 3] &= (uint8_t)~(1u << ((i) & 7)); } while (0)
#define IS_LMS(t, i) ((i) > 0 && TGET(t, i) && !TGET(t, (i) - 1))

static void bucket_bounds(const int32_t *s, int32_t *bkt, int32_t n, int32_t K, int ends) {
    int32_t i, sum = 0;
    for (i = 0; i <= K; i++) bkt[i] = 0;
    for (i = 0; i < n; i++) bkt[s[i]]++;
    for (i = 0; i <= K; i++) { sum += bkt[i]; bkt[i] = ends ? sum : sum - bkt[i]; }
}
static void induce_L(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int32_t n, int32_t K) {
    int32_t i, j;
    bucket_bounds(s, bkt, n, K, 0);
    for (i = 0; i < n; i++) { j = SA[i] - 1; if (j >= 0 && !TGET(t, j)) SA[bkt[s[j]]++] = j; }
}
static void induce_S(const uint8_t *t, int32_t *SA, const int32_t *s, int32_t *bkt, int32_t n, int32_t K) {
    int32_t i, j;
    bucket_bounds(s, bkt, n, K, 1);
    for (i = n - 1; i >= 0; i--) { j = SA[i] - 1; if (j >= 0 && TGET(t, j)) SA[--bkt[s[j]]] = j; }
}
static int sais_i32(const int32_t *s, int32_t *SA, int32_t n, int32_t K) {
    uint8_t *t; int32_t *bkt; int32_t i, j, n1 = 0, name = 0, prev = -1;
    int32_t *SA1, *s1;
    if (n == 1) { SA[0] = 0; return 0; }
    t = (uint8_t *)calloc((size_t)n / 8 + 1, 1);
    bkt = (int32_t *)malloc(((size_t)K + 1) * sizeof(int32_t));
    if (!t || !bkt) return -1;
    TSET(t, n - 2, 0); TSET(t, n - 1, 1);
    for (i = n - 3; i >= 0; i--)
        TSET(t, i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && TGET(t, i + 1))) ? 1 : 0);
    bucket_bounds(s, bkt, n, K, 1);
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 1; i < n; i++) if (IS_LMS(t, i)) SA[--bkt[s[i]]] = i;
    induce_L(t, SA, s, bkt, n, K);
    induce_S(t, SA, s, bkt, n, K);
    for (i = 0; i < n; i++) if (IS_LMS(t, SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; i++) SA[i] = -1;
    for (i = 0; i < n1; i++) {
        int32_t pos = SA[i], d; int diff = 0;
        for (d = 0; d < n; d++) {
            if (prev == -1 || s[pos + d] != s[prev + d] || TGET(t, pos + d) != TGET(t, prev + d)) { diff = 1; break; }
            else if (d > 0 && (IS_LMS(t, pos + d) || IS_LMS(t, prev + d))) break;
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; i--) if (SA[i] >= 0) SA[j--] = SA[i];
    SA1 = SA; s1 = SA + n - n1;
    if (name < n1) { if (sais_i32(s1, SA1, n1, name - 1)) return -1; }
    else for (i = 0; i < n1; i++) SA1[s1[i]] = i;
    bucket_bounds(s, bkt, n, K, 1);
    for (i = 1, j = 0; i < n; i++) if (IS_LMS(t, i)) s1[j++] = i;
    for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; i++) SA[i] = -1;
    for (i = n1 - 1; i >= 0; i--) { j = SA[i]; SA[i] = -1; SA[--bkt[s[j]]] = j; }
    induce_L(t, SA, s, bkt, n, K);
    induce_S(t, SA, s, bkt, n, K);
    free(bkt); free(t);
    return 0;
}

/* Stream of SURVEY.md 8(0): suffixes of T.sentinel in lexicographic order,
 * j = 0..n; lcp[j] = LCP(suffix j-1, suffix j) (Kasai et al. 2001, the same
 * array gsacak returns at direct_gsacak.hpp:62); bwt[j] = T[sa[j]-1], 0 when
 * sa[j] = 0 (pfp_lcp_mum.hpp:268; direct_gsacak.hpp:66 wraps to the 0 byte). */
int mmo_build_stream(const uint8_t *text, int64_t n, int64_t *sa, int64_t *lcp, uint8_t *bwt) {
    int32_t *s, *SA, *isa; int64_t i, h = 0, m = n + 1;
    if (n >= 0x7ffffff0LL) return -2;
    s = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    SA = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    isa = (int32_t *)malloc((size_t)m * sizeof(int32_t));
    if (!s || !SA || !isa) return -1;
    for (i = 0; i < n; i++) s[i] = (int32_t)text[i] + 1;
    s[n] = 0;
    if (sais_i32(s, SA, (int32_t)m, 257)) return -1;
    for (i = 0; i < m; i++) { sa[i] = SA[i]; isa[SA[i]] = (int32_t)i; }
    lcp[0] = 0;
    for (i = 0; i < n; i++) {           /* text order, amortised linear */
        int64_t r = isa[i], p;
        if (r == 0) { h = 0; continue; }
        p = SA[r - 1];
        while (i + h < n && p + h < n && text[i + h] == text[p + h]) h++;
        lcp[r] = h;
        if (h > 0) h--;
    }
    for (i = 0; i < m; i++) bwt[i] = sa[i] > 0 ? text[sa[i] - 1] : 0;
    free(s); free(SA); free(isa);
    return 0;
}

void mmo_doc_array(const int64_t *sa, int64_t m, const int64_t *doc_start, int64_t n_docs, int32_t *doc) {
    int64_t j;
    for (j = 0; j < m; j++) {      /* rank of doc_ends below sa: #docs fully before sa */
        int64_t lo = 0, hi = n_docs; /* largest d with doc_start[d] <= sa, capped at n_docs */
        while (lo < hi) { int64_t mid = (lo + hi + 1) / 2; if (doc_start[mid] <= sa[j]) lo = mid; else hi = mid - 1; }
        doc[j] = (int32_t)lo;
    }
}

/* ------------------------------------------------------------------------ */
/* match scan: include/mem_finder.hpp                                        */
/* ------------------------------------------------------------------------ */
struct mmo_result {
    int64_t n_docs; int32_t revcomp, mummode, merge; int64_t num_distinct;
    int64_t *doc_start; int64_t *half;      /* half[d] = L_d + 1                 */
    vec64 iv;       /* 4 per emitted interval: start, end, length, closing j      */
    vec64 acc;      /* 4 per accepted (pre-BWT) candidate                         */
    vec64 occ_sa, occ_doc, occ_first; /* flat occurrences of emitted intervals     */
    uint16_t *thresh; int64_t thresh_len;
    /* rows after the writer-side rules */
    int64_t n_rows; vec64 row_iv;  /* index into iv of each row                   */
    vec64 mum_pos;  /* (offset in doc 0, length) per written MUM, write order      */
};

static int doc_range_ok(const int32_t *doc, int64_t start, int64_t end, int64_t n_docs,
                        int64_t max_doc_freq, int64_t num_distinct, int64_t *cnt) {
    /* mem_finder.hpp:265-289: per-document counts over [start,end]; fail as
     * soon as one document exceeds max_doc_freq (0 = unlimited); finally
     * require #distinct documents >= num_distinct.                            */
    int64_t i, uniq = 0; int ok = 1;
    for (i = start; i <= end; i++) {
        int32_t d = doc[i];
        if (d < 0 || d >= n_docs) { ok = 0; break; }
        if (cnt[d] == 0) { uniq++; cnt[d] = 1; }
        else if (max_doc_freq && ++cnt[d] > max_doc_freq) { ok = 0; break; }
    }
    for (i = start; i <= end; i++) if (doc[i] >= 0 && doc[i] < n_docs) cnt[doc[i]] = 0;
    return ok && uniq >= num_distinct;
}

static void finish_rows(mmo_result *r);

mmo_result *mmo_scan(const int64_t *sa, const int64_t *lcp, const uint8_t *bwt,
                     const int32_t *doc, int64_t m, const int64_t *doc_start,
                     int64_t n_docs, const mmo_scan_params *p) {
    mmo_result *r = (mmo_result *)calloc(1, sizeof(*r));
    /* stack of (start, length, lcp before start): mem_finder.hpp:292 */
    int64_t *st_start, *st_len, *st_prev, sp = 0, cap = 1024;
    int64_t *cnt = (int64_t *)calloc((size_t)n_docs + 1, sizeof(int64_t));
    int64_t j, d, last_bwt_change = 0, prev_lcp = 0;
    int no_max_freq = p->max_total_freq == 0;   /* :91 */
    r->n_docs = n_docs; r->revcomp = p->revcomp; r->merge = p->merge;
    r->mummode = p->max_doc_freq == 1;         /* :85 */
    r->num_distinct = p->num_distinct;
    r->doc_start = (int64_t *)malloc(((size_t)n_docs + 1) * sizeof(int64_t));
    r->half = (int64_t *)malloc(((size_t)n_docs + 1) * sizeof(int64_t));
    memcpy(r->doc_start, doc_start, ((size_t)n_docs + 1) * sizeof(int64_t));
    for (d = 0; d < n_docs; d++) {             /* :67-79 */
        int64_t len = doc_start[d + 1] - doc_start[d];
        r->half[d] = p->revcomp ? len / 2 : len;
    }
    if (p->merge && n_docs > 0) {
        r->thresh_len = r->half[0] * 2;
        r->thresh = (uint16_t *)calloc((size_t)r->thresh_len, sizeof(uint16_t));
    }
    st_start = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    st_len = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    st_prev = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
    st_start[0] = 0; st_len[0] = 0; st_prev[0] = 0; sp = 1; /* init_stack :505-508 */

    for (j = 0; j < m; j++) {                   /* update(), :161-170 */
        int64_t cur = lcp[j], start = j - 1;
        while (cur < st_len[sp - 1]) {          /* update_mems(), :313-347 */
            int64_t is = st_start[sp - 1], il = st_len[sp - 1], prev = st_prev[sp - 1];
            sp--;
            if (il >= p->min_len && j - is >= p->num_distinct &&
                (no_max_freq || j - is <= p->max_total_freq) &&
                doc_range_ok(doc, is, j - 1, n_docs, p->max_doc_freq, p->num_distinct, cnt)) {
                v64_push(&r->acc, is); v64_push(&r->acc, j - 1); v64_push(&r->acc, il); v64_push(&r->acc, j);
                if (p->merge) {                  /* :326-336 */
                    int64_t nb = prev > cur ? prev : cur, i;
                    if (nb > 65535) nb = 65535;
                    for (i = is; i <= j - 1; i++)
                        if (doc[i] == 0) { r->thresh[sa[i] - doc_start[0]] = (uint16_t)nb; break; }
                }
                if (!(last_bwt_change <= is)) {  /* check_bwt_range, :189-192 */
                    int64_t i;
                    v64_push(&r->iv, is); v64_push(&r->iv, j - 1); v64_push(&r->iv, il); v64_push(&r->iv, j);
                    v64_push(&r->occ_first, r->occ_sa.n);
                    for (i = is; i <= j - 1; i++) { v64_push(&r->occ_sa, sa[i]); v64_push(&r->occ_doc, doc[i]); }
                }
            }
            start = is; prev_lcp = prev;
        }
        if (cur > st_len[sp - 1] && cur >= p->min_len) {   /* :349-353 */
            if (sp == cap) {
                cap *= 2;
                st_start = (int64_t *)realloc(st_start, (size_t)cap * sizeof(int64_t));
                st_len = (int64_t *)realloc(st_len, (size_t)cap * sizeof(int64_t));
                st_prev = (int64_t *)realloc(st_prev, (size_t)cap * sizeof(int64_t));
            }
            st_start[sp] = start; st_len[sp] = cur; st_prev[sp] = prev_lcp; sp++;
        }
        if (j == 0 || bwt[j - 1] != bwt[j]) last_bwt_change = j;  /* :165-166 */
        prev_lcp = cur;                                          /* :168 */
    }
    /* no flush of the stack after the last entry (pfp_lcp_mum.hpp:223-230) */
    v64_push(&r->occ_first, r->occ_sa.n);
    free(st_start); free(st_len); free(st_prev); free(cnt);
    finish_rows(r);
    return r;
}

/* coordinate of one occurrence: mem_finder.hpp:367-377 (write_mum) and
 * :224-231, 244-249 (write_mem).  Returns 0 if write_mum would drop the row. */
static int map_pos(const mmo_result *r, int64_t sa, int64_t d, int64_t len, int last_mem_rule,
                   int64_t *pos, int *minus) {
    int64_t cur = sa - r->doc_start[d], half = r->half[d];
    *minus = 0;
    if (r->revcomp && cur >= half) {
        *minus = 1;
        if (r->mummode && cur + len >= half + half) return 0;
        cur = half + half - cur - len - (last_mem_rule ? 0 : 1);
    }
    *pos = cur;
    return 1;
}

static void finish_rows(mmo_result *r) {
    int64_t n_iv = r->iv.n / 4, k;
    for (k = 0; k < n_iv; k++) {
        int64_t len = r->iv.v[4 * k + 2];
        int64_t o0 = r->occ_first.v[k], o1 = r->occ_first.v[k + 1], o;
        if (r->mummode) {                                   /* write_mum, :357-428 */
            int keep = 1; int64_t first_doc = -1, first_minus = 0, off0 = -1;
            int64_t last_doc_minus = 0, last_doc_present = 0;
            for (o = o0; o < o1 && keep; o++) {
                int64_t pos; int minus;
                if (!map_pos(r, r->occ_sa.v[o], r->occ_doc.v[o], len, 0, &pos, &minus)) keep = 0;
            }
            if (!keep) continue;
            /* first present doc among 0..N-2, else doc N-1 (:380-391) */
            for (o = o0; o < o1; o++) {
                int64_t d = r->occ_doc.v[o], pos; int minus;
                map_pos(r, r->occ_sa.v[o], d, len, 0, &pos, &minus);
                if (d == r->n_docs - 1) { last_doc_present = 1; last_doc_minus = minus; }
                else if (first_doc < 0 || d < first_doc) { first_doc = d; first_minus = minus; }
                if (d == 0) off0 = pos;
            }
            if (first_doc < 0) { if (last_doc_present && last_doc_minus) continue; }
            else if (first_minus) continue;
            v64_push(&r->mum_pos, off0); v64_push(&r->mum_pos, len);
        }
        v64_push(&r->row_iv, k);
    }
    r->n_rows = r->row_iv.n;
}

void mmo_result_free(mmo_result *r) {
    if (!r) return;
    free(r->doc_start); free(r->half); free(r->iv.v); free(r->acc.v); free(r->occ_sa.v);
    free(r->occ_doc.v); free(r->occ_first.v); free(r->thresh); free(r->row_iv.v); free(r->mum_pos.v);
    free(r);
}
void mmo_free(void *p) { free(p); }
int64_t mmo_num_intervals(const mmo_result *r) { return r->iv.n / 4; }
void mmo_get_intervals(const mmo_result *r, int64_t *out) { memcpy(out, r->iv.v, (size_t)r->iv.n * sizeof(int64_t)); }
int64_t mmo_num_accepted(const mmo_result *r) { return r->acc.n / 4; }
void mmo_get_accepted(const mmo_result *r, int64_t *out) { memcpy(out, r->acc.v, (size_t)r->acc.n * sizeof(int64_t)); }
int64_t mmo_num_rows(const mmo_result *r) { return r->n_rows; }

void mmo_get_mum_rows(const mmo_result *r, uint32_t *length, int64_t *offsets, uint8_t *strands) {
    int64_t i, o;                       /* mumemto_api.cpp:241-286 */
    for (i = 0; i < r->n_rows; i++) {
        int64_t k = r->row_iv.v[i], len = r->iv.v[4 * k + 2];
        length[i] = (uint32_t)len;
        for (o = 0; o < r->n_docs; o++) { offsets[i * r->n_docs + o] = -1; strands[i * r->n_docs + o] = 0; }
        for (o = r->occ_first.v[k]; o < r->occ_first.v[k + 1]; o++) {
            int64_t pos, d = r->occ_doc.v[o]; int minus;
            map_pos(r, r->occ_sa.v[o], d, len, 0, &pos, &minus);
            offsets[i * r->n_docs + d] = pos; strands[i * r->n_docs + d] = minus ? 0 : 1;
        }
    }
}
int64_t mmo_num_occ(const mmo_result *r) {
    int64_t i, t = 0;
    for (i = 0; i < r->n_rows; i++) { int64_t k = r->row_iv.v[i]; t += r->occ_first.v[k + 1] - r->occ_first.v[k]; }
    return t;
}
void mmo_get_mem_rows(const mmo_result *r, uint32_t *length, int64_t *occ_start, int64_t *offsets,
                      int64_t *docs, uint8_t *strands) {
    int64_t i, o, t = 0;               /* mumemto_api.cpp:137-166 */
    for (i = 0; i < r->n_rows; i++) {
        int64_t k = r->row_iv.v[i], len = r->iv.v[4 * k + 2];
        int64_t o0 = r->occ_first.v[k], o1 = r->occ_first.v[k + 1];
        length[i] = (uint32_t)len; occ_start[i] = t;
        for (o = o0; o < o1; o++, t++) {
            int64_t pos; int minus;
            map_pos(r, r->occ_sa.v[o], r->occ_doc.v[o], len, o == o1 - 1, &pos, &minus);
            offsets[t] = pos; docs[t] = r->occ_doc.v[o]; strands[t] = minus ? 0 : 1;
        }
    }
    occ_start[r->n_rows] = t;
}

char *mmo_format_text(const mmo_result *r, int64_t *out_len) {
    vecc b = {0, 0, 0}; int64_t i, o, d;
    int64_t *off = (int64_t *)malloc(((size_t)r->n_docs + 1) * sizeof(int64_t));
    char *sd = (char *)malloc((size_t)r->n_docs + 1);
    for (i = 0; i < r->n_rows; i++) {
        int64_t k = r->row_iv.v[i], len = r->iv.v[4 * k + 2];
        int64_t o0 = r->occ_first.v[k], o1 = r->occ_first.v[k + 1];
        vc_putu(&b, (uint64_t)len); vc_putc(&b, '\t');
        if (r->mummode) {               /* mem_finder.hpp:406-426 */
            for (d = 0; d < r->n_docs; d++) { off[d] = -1; sd[d] = 0; }
            for (o = o0; o < o1; o++) {
                int64_t pos; int minus;
                map_pos(r, r->occ_sa.v[o], r->occ_doc.v[o], len, 0, &pos, &minus);
                off[r->occ_doc.v[o]] = pos; sd[r->occ_doc.v[o]] = minus ? '-' : '+';
            }
            for (d = 0; d < r->n_docs - 1; d++) { if (off[d] >= 0) vc_putu(&b, (uint64_t)off[d]); vc_putc(&b, ','); }
            if (off[r->n_docs - 1] >= 0) vc_putu(&b, (uint64_t)off[r->n_docs - 1]);
            vc_putc(&b, '\t');
            for (d = 0; d < r->n_docs - 1; d++) { if (off[d] >= 0) vc_putc(&b, sd[d]); vc_putc(&b, ','); }
            if (off[r->n_docs - 1] >= 0) vc_putc(&b, sd[r->n_docs - 1]);
        } else {                        /* mem_finder.hpp:210-263 */
            int pass;
            for (pass = 0; pass < 3; pass++) {
                for (o = o0; o < o1; o++) {
                    int64_t pos; int minus;
                    map_pos(r, r->occ_sa.v[o], r->occ_doc.v[o], len, o == o1 - 1, &pos, &minus);
                    if (pass == 0) vc_putu(&b, (uint64_t)pos);
                    else if (pass == 1) vc_putu(&b, (uint64_t)r->occ_doc.v[o]);
                    else vc_putc(&b, minus ? '-' : '+');
                    if (o != o1 - 1) vc_putc(&b, ',');
                }
                if (pass != 2) vc_putc(&b, '\t');
            }
        }
        vc_putc(&b, '\n');
    }
    free(off); free(sd);
    vc_reserve(&b, 1); b.v[b.n] = 0;
    *out_len = b.n;
    return b.v;
}

uint8_t *mmo_format_bumbl(const mmo_result *r, int64_t *out_len) {
    /* mem_finder.hpp:451-503: u16 flags | u64 n_seqs | u64 n_mums | u32 len[] |
     * i64 start[n_mums][n_seqs] | strand bits MSB-first ('+' = 1)             */
    int64_t nm = r->n_rows, ns = r->n_docs, nbits = nm * ns, i;
    int64_t total = 2 + 8 + 8 + 4 * nm + 8 * nbits + (nbits + 7) / 8;
    uint8_t *buf = (uint8_t *)calloc((size_t)total + 1, 1), *p = buf;
    uint16_t flags = (uint16_t)(1u << 15);
    uint32_t *len = (uint32_t *)malloc(((size_t)nm + 1) * 4);
    int64_t *off = (int64_t *)malloc(((size_t)nbits + 1) * 8);
    uint8_t *sd = (uint8_t *)malloc((size_t)nbits + 1);
    uint64_t u;
    if (r->num_distinct < r->n_docs) flags |= (uint16_t)(1u << 13);
    mmo_get_mum_rows(r, len, off, sd);
    memcpy(p, &flags, 2); p += 2;
    u = (uint64_t)ns; memcpy(p, &u, 8); p += 8;
    u = (uint64_t)nm; memcpy(p, &u, 8); p += 8;
    memcpy(p, len, (size_t)nm * 4); p += nm * 4;
    memcpy(p, off, (size_t)nbits * 8); p += nbits * 8;
    for (i = 0; i < nbits; i++) if (sd[i]) p[i / 8] |= (uint8_t)(1u << (7 - (i % 8)));
    free(len); free(off); free(sd);
    *out_len = total;
    return buf;
}

int64_t mmo_thresh_len(const mmo_result *r) { return r->thresh_len; }
void mmo_get_thresh(const mmo_result *r, uint16_t *out) { memcpy(out, r->thresh, (size_t)r->thresh_len * 2); }

static int cmp_pair(const void *a, const void *b) {
    const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
    return x[0] < y[0] ? -1 : x[0] > y[0];
}
uint16_t *mmo_format_thresh(const mmo_result *r, int which, int64_t *out_len) {
    /* mem_finder.hpp:116-157 */
    int64_t n = r->mum_pos.n / 2, i, j, total = 0, off = 0;
    int64_t *mp = (int64_t *)malloc(((size_t)n * 2 + 2) * sizeof(int64_t));
    uint16_t *out;
    memcpy(mp, r->mum_pos.v, (size_t)n * 2 * sizeof(int64_t));
    for (i = 0; i < n; i++) total += mp[2 * i + 1] + 1;
    out = (uint16_t *)calloc((size_t)total + 1, 2);
    qsort(mp, (size_t)n, 2 * sizeof(int64_t), cmp_pair);
    for (i = 0; i < n; i++) {
        int64_t first = mp[2 * i], len = mp[2 * i + 1];
        int64_t revpos = r->half[0] + r->half[0] - first - len - 1;
        for (j = 0; j < len; j++) {
            int64_t idx = which ? revpos + j : first + j;
            if (idx >= 0 && idx < r->thresh_len && (int64_t)r->thresh[idx] < len - j) out[off] = r->thresh[idx];
            off++;
        }
        out[off++] = 0;
    }
    free(mp);
    *out_len = total;
    return out;
}

/* ------------------------------------------------------------------------ */
/* anchor merge: src/merge_candidates.cpp                                    */
/* ------------------------------------------------------------------------ */
typedef struct {
    int64_t n_rows, n_docs, nb_len;
    uint32_t *length; int64_t *offsets; uint8_t *strands; uint8_t *bv; uint16_t *nb;
} part_t;
struct mmo_merged { part_t p; };

static const int64_t *g_sort_off; static int64_t g_sort_nd;
static int cmp_row(const void *a, const void *b) {
    int64_t x = g_sort_off[*(const int64_t *)a * g_sort_nd], y = g_sort_off[*(const int64_t *)b * g_sort_nd];
    return x < y ? -1 : x > y;
}
static void part_free(part_t *p) { free(p->length); free(p->offsets); free(p->strands); free(p->bv); free(p->nb); }

/* parse_candidate(), merge_candidates.cpp:62-95 */
static part_t part_load(const mmo_partition *in) {
    part_t p; int64_t i, c, *order;
    p.n_rows = in->n_rows; p.n_docs = in->n_docs; p.nb_len = in->nb_len;
    p.length = (uint32_t *)malloc(((size_t)p.n_rows + 1) * 4);
    p.offsets = (int64_t *)malloc(((size_t)(p.n_rows * p.n_docs) + 1) * 8);
    p.strands = (uint8_t *)malloc((size_t)(p.n_rows * p.n_docs) + 1);
    p.bv = (uint8_t *)calloc((size_t)p.nb_len + 1, 1);
    p.nb = (uint16_t *)malloc(((size_t)p.nb_len + 1) * 2);
    memcpy(p.nb, in->nb, (size_t)p.nb_len * 2);
    order = (int64_t *)malloc(((size_t)p.n_rows + 1) * 8);
    for (i = 0; i < p.n_rows; i++) order[i] = i;
    g_sort_off = in->offsets; g_sort_nd = in->n_docs;
    qsort(order, (size_t)p.n_rows, 8, cmp_row);
    for (i = 0; i < p.n_rows; i++) {
        int64_t s = order[i];
        p.length[i] = in->length[s];
        for (c = 0; c < p.n_docs; c++) {
            p.offsets[i * p.n_docs + c] = in->offsets[s * p.n_docs + c];
            p.strands[i * p.n_docs + c] = in->strands[s * p.n_docs + c];
        }
        p.bv[in->offsets[s * p.n_docs]] = 1;
    }
    free(order);
    return p;
}

/* merge_partitions(), merge_candidates.cpp:106-157 (+ fix_neg_strand :97-104) */
static part_t part_merge(const part_t *a, const part_t *b) {
    part_t o; int64_t i, c, idx1 = 0, idx2 = 0, cur1 = -1, cur2 = -1, last1 = 0, last2 = 0, cap = 64;
    o.n_docs = a->n_docs + b->n_docs - 1; o.nb_len = a->nb_len; o.n_rows = 0;
    o.length = (uint32_t *)malloc((size_t)cap * 4);
    o.offsets = (int64_t *)malloc((size_t)(cap * o.n_docs) * 8);
    o.strands = (uint8_t *)malloc((size_t)(cap * o.n_docs));
    o.bv = (uint8_t *)calloc((size_t)o.nb_len + 1, 1);
    o.nb = (uint16_t *)calloc((size_t)o.nb_len + 1, 2);
    for (i = 0; i < a->nb_len; i++) {
        int both = a->nb[i] > 0 && b->nb[i] > 0;
        if (both) o.nb[i] = a->nb[i] > b->nb[i] ? a->nb[i] : b->nb[i];
        if (a->bv[i]) { cur1 = idx1++; last1 = i; }
        if (b->bv[i]) { cur2 = idx2++; last2 = i; }
        if (cur1 >= 0 && cur2 >= 0 && (a->bv[i] || b->bv[i]) && both) {
            int64_t d1 = i - last1, d2 = i - last2;
            uint32_t s1, s2, nl;
            if (d1 > (int64_t)a->length[cur1] || d2 > (int64_t)b->length[cur2]) continue;
            s1 = (uint32_t)(a->length[cur1] - d1); s2 = (uint32_t)(b->length[cur2] - d2);
            nl = s1 < s2 ? s1 : s2;
            if (nl > o.nb[i] && nl >= 20) {
                int64_t *ro; uint8_t *rs;
                if (o.n_rows == cap) {
                    cap *= 2;
                    o.length = (uint32_t *)realloc(o.length, (size_t)cap * 4);
                    o.offsets = (int64_t *)realloc(o.offsets, (size_t)(cap * o.n_docs) * 8);
                    o.strands = (uint8_t *)realloc(o.strands, (size_t)(cap * o.n_docs));
                }
                ro = o.offsets + o.n_rows * o.n_docs; rs = o.strands + o.n_rows * o.n_docs;
                for (c = 0; c < a->n_docs; c++) {
                    uint8_t st = a->strands[cur1 * a->n_docs + c];
                    ro[c] = a->offsets[cur1 * a->n_docs + c] + (st ? d1 : (int64_t)s1 - (int64_t)nl);
                    rs[c] = st;
                }
                for (c = 1; c < b->n_docs; c++) {
                    uint8_t st = b->strands[cur2 * b->n_docs + c];
                    ro[a->n_docs + c - 1] = b->offsets[cur2 * b->n_docs + c] + (st ? d2 : (int64_t)s2 - (int64_t)nl);
                    rs[a->n_docs + c - 1] = st;
                }
                o.length[o.n_rows] = nl;
                o.bv[ro[0]] = 1;
                o.n_rows++;
            }
        }
    }
    return o;
}

mmo_merged *mmo_anchor_merge(const mmo_partition *parts, int64_t k) {
    mmo_merged *m = (mmo_merged *)calloc(1, sizeof(*m));
    part_t left = part_load(&parts[0]); int64_t i;
    for (i = 1; i < k; i++) {           /* main(), :208-219 */
        part_t right = part_load(&parts[i]);
        part_t nxt = part_merge(&left, &right);
        part_free(&left); part_free(&right);
        left = nxt;
    }
    m->p = left;
    return m;
}
int64_t mmo_merged_rows(const mmo_merged *m) { return m->p.n_rows; }
int64_t mmo_merged_docs(const mmo_merged *m) { return m->p.n_docs; }
void mmo_merged_get(const mmo_merged *m, uint32_t *length, int64_t *offsets, uint8_t *strands, uint16_t *nb) {
    const part_t *p = &m->p;
    memcpy(length, p->length, (size_t)p->n_rows * 4);
    memcpy(offsets, p->offsets, (size_t)(p->n_rows * p->n_docs) * 8);
    memcpy(strands, p->strands, (size_t)(p->n_rows * p->n_docs));
    memcpy(nb, p->nb, (size_t)p->nb_len * 2);
}
void mmo_merged_free(mmo_merged *m) { if (m) { part_free(&m->p); free(m); } }

/* ------------------------------------------------------------------------ */
/* the stream by way of the prefix-free parse: the reference's default route  */
/* ------------------------------------------------------------------------ */
/* Restates, with plain arrays instead of sdsl bit vectors and with the SA-IS
 * above instead of gsacak / sacak_int:
 *   parse        include/newscan.hpp:106-114 (rolling hash), :265-325 (phrases:
 *                a phrase ends where hash % p == 0 and is longer than w; the next
 *                one begins with its last w characters), :357-423 (the text is
 *                Dollar . T . Dollar^w; ranks of the distinct phrases)
 *   dictionary   include/dictionary.hpp:103-157 (suffix array and LCP array of the
 *                phrases in rank order, each followed by EndOfWord -- a separator
 *                that is unique and ordered by position, the gsacak convention)
 *   parse        include/parse.hpp:77-146 (suffix array of the rank sequence,
 *                inverted lists: per phrase the parse ranks of the suffixes that
 *                FOLLOW its occurrences, ascending)
 *   pfp          include/pfp.hpp:171-244 (text position of every parse suffix;
 *                s_lcp_T: LCP in characters of adjacent parse suffixes + range minima)
 *   emitter      include/pfp_lcp_mum.hpp:115-231 (dictionary suffixes in order;
 *                the proper phrase suffixes of length >= w that are equal form a
 *                group; its occurrences are merged by the rank of the following
 *                parse suffix), :257-282 (inc / is_valid), :284-321 (the LCP
 *                values), :337-341 (suffix-array entry = text position of the
 *                following phrase - suffix length).
 * The stream does not depend on (w, p): it must equal mmo_build_stream's.      */
typedef struct { int32_t *blockmin; int32_t **tab; int64_t nb; int levels; const int32_t *a; int64_t n; } rmq_t;
#define RMQ_B 32
static int rmq_build(rmq_t *r, const int32_t *a, int64_t n) {
    int64_t nb = (n + RMQ_B - 1) / RMQ_B, i; int l, levels = 1;
    while (((int64_t)1 << levels) <= nb) levels++;
    r->a = a; r->n = n; r->nb = nb; r->levels = levels;
    r->tab = (int32_t **)calloc((size_t)levels, sizeof(int32_t *));
    if (!r->tab) return -1;
    r->tab[0] = (int32_t *)malloc((size_t)(nb ? nb : 1) * 4);
    if (!r->tab[0]) return -1;
    for (i = 0; i < nb; i++) {
        int64_t lo = i * RMQ_B, hi = lo + RMQ_B < n ? lo + RMQ_B : n, k; int32_t m = a[lo];
        for (k = lo + 1; k < hi; k++) if (a[k] < m) m = a[k];
        r->tab[0][i] = m;
    }
    for (l = 1; l < levels; l++) {
        int64_t span = (int64_t)1 << l, cnt = nb - span + 1;
        if (cnt <= 0) { r->levels = l; break; }
        r->tab[l] = (int32_t *)malloc((size_t)cnt * 4);
        if (!r->tab[l]) return -1;
        for (i = 0; i < cnt; i++) {
            int32_t x = r->tab[l - 1][i], y = r->tab[l - 1][i + span / 2];
            r->tab[l][i] = x < y ? x : y;
        }
    }
    return 0;
}
static int32_t rmq_min(const rmq_t *r, int64_t lo, int64_t hi) {      /* minimum of a[lo .. hi], lo <= hi */
    int64_t bl = lo / RMQ_B, bh = hi / RMQ_B, k; int32_t m = r->a[lo];
    if (bl == bh) { for (k = lo + 1; k <= hi; k++) if (r->a[k] < m) m = r->a[k]; return m; }
    for (k = lo + 1; k < (bl + 1) * RMQ_B; k++) if (r->a[k] < m) m = r->a[k];
    for (k = bh * RMQ_B; k <= hi; k++) if (r->a[k] < m) m = r->a[k];
    if (bh - bl > 1) {
        int64_t a0 = bl + 1, a1 = bh - 1, len = a1 - a0 + 1; int l = 0;
        while (((int64_t)2 << l) <= len) l++;
        { int32_t x = r->tab[l][a0], y = r->tab[l][a1 - ((int64_t)1 << l) + 1];
          if (x < m) m = x;
          if (y < m) m = y; }
    }
    return m;
}
static void rmq_free(rmq_t *r) { int l; if (r->tab) { for (l = 0; l < r->levels; l++) free(r->tab[l]); free(r->tab); } }

static const uint8_t *g_cmp_v; static const int64_t *g_cmp_start; static const int32_t *g_cmp_len;
static int cmp_phrase(const void *x, const void *y) {                /* lexicographic, the shorter one first on a tie */
    int32_t a = *(const int32_t *)x, b = *(const int32_t *)y;
    int32_t la = g_cmp_len[a], lb = g_cmp_len[b], l = la < lb ? la : lb;
    int c = memcmp(g_cmp_v + g_cmp_start[a], g_cmp_v + g_cmp_start[b], (size_t)l);
    if (c) return c;
    return la < lb ? -1 : (la > lb ? 1 : 0);
}
typedef struct { int64_t val; int64_t at, end; uint8_t bwt; } pfp_run;    /* one member's list: next value, cursor, end */
static void heap_sift(pfp_run *h, int64_t n, int64_t i) {
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i; pfp_run t;
        if (l < n && h[l].val < h[m].val) m = l;
        if (r < n && h[r].val < h[m].val) m = r;
        if (m == i) return;
        t = h[i]; h[i] = h[m]; h[m] = t; i = m;
    }
}

int mmo_build_stream_pfp(const uint8_t *text, int64_t n, int64_t w, int64_t pmod, int64_t *sa, int64_t *lcp,
                         uint8_t *bwt, int64_t *stats) {
    const uint64_t PRIME = 1999999973ull;               /* newscan.hpp:84 */
    int64_t nv = 1 + n + w, i, k, m = 0, cap = 1024, D = 0, nd = 0, j = 0;
    uint8_t *v; int64_t *pstart; int32_t *plen, *order, *rank_of, *parse, *saP, *isaP, *slcp, *ilist, *ilist_start;
    uint8_t *d; int32_t *ds, *saD, *isaD, *lcpD, *dphr, *dsuf; int64_t *posval;
    uint64_t *window, hash = 0, pot = 1; int64_t tot = 0;
    rmq_t rq; pfp_run *heap = NULL; int64_t heap_cap = 0;
    if (w < 1 || pmod < 1 || n < 1 || nv >= 0x7fffff00LL) return -2;
    for (i = 0; i < n; i++) if (text[i] <= 2) return -3;             /* newscan.hpp:318 */
    v = (uint8_t *)malloc((size_t)nv + 8);
    pstart = (int64_t *)malloc((size_t)cap * 8); plen = (int32_t *)malloc((size_t)cap * 4);
    window = (uint64_t *)calloc((size_t)w, 8);
    if (!v || !pstart || !plen || !window) return -1;
    v[0] = 2; memcpy(v + 1, text, (size_t)n); memset(v + 1 + n, 2, (size_t)w);
    for (i = 1; i < w; i++) pot = (pot * 256) % PRIME;
    /* phrases */
    {
        int64_t start = 0;
        for (i = 1; i <= n; i++) {
            const uint64_t c = v[i]; const int64_t slot = tot++ % w;
            hash += PRIME - (window[slot] * pot) % PRIME;
            hash = (256 * hash + c) % PRIME;
            window[slot] = c;
            if (hash % (uint64_t)pmod == 0 && i - start + 1 > w) {
                if (m == cap) { cap *= 2; pstart = (int64_t *)realloc(pstart, (size_t)cap * 8); plen = (int32_t *)realloc(plen, (size_t)cap * 4); if (!pstart || !plen) return -1; }
                pstart[m] = start; plen[m] = (int32_t)(i - start + 1); m++;
                start = i - w + 1;
            }
        }
        if (m == cap) { cap += 1; pstart = (int64_t *)realloc(pstart, (size_t)cap * 8); plen = (int32_t *)realloc(plen, (size_t)cap * 4); if (!pstart || !plen) return -1; }
        pstart[m] = start; plen[m] = (int32_t)(nv - start); m++;
    }
    free(window);
    /* ranks of the distinct phrases (newscan.hpp:386-419: the dictionary is sorted, the parse holds ranks from 1) */
    order = (int32_t *)malloc((size_t)m * 4); rank_of = (int32_t *)malloc((size_t)m * 4); parse = (int32_t *)malloc(((size_t)m + 1) * 4);
    if (!order || !rank_of || !parse) return -1;
    for (k = 0; k < m; k++) order[k] = (int32_t)k;
    g_cmp_v = v; g_cmp_start = pstart; g_cmp_len = plen;
    qsort(order, (size_t)m, 4, cmp_phrase);
    for (k = 0; k < m; k++) {
        if (k == 0 || cmp_phrase(&order[k - 1], &order[k]) != 0) { D++; nd += plen[order[k]] + 1; }
        rank_of[order[k]] = (int32_t)D;
    }
    nd += 1;                                                         /* EndOfDict */
    for (k = 0; k < m; k++) parse[k] = rank_of[k];
    parse[m] = 0;
    /* dictionary: text, per-position phrase and remaining length, suffix array with unique ordered separators, LCP */
    d = (uint8_t *)malloc((size_t)nd); ds = (int32_t *)malloc((size_t)nd * 4); saD = (int32_t *)malloc((size_t)nd * 4);
    isaD = (int32_t *)malloc((size_t)nd * 4); lcpD = (int32_t *)malloc((size_t)nd * 4);
    dphr = (int32_t *)malloc((size_t)nd * 4); dsuf = (int32_t *)malloc((size_t)nd * 4);
    if (!d || !ds || !saD || !isaD || !lcpD || !dphr || !dsuf) return -1;
    {
        int64_t at = 0; int32_t r = 0;
        for (k = 0; k < m; k++) {
            const int32_t ph = order[k];
            if (k && cmp_phrase(&order[k - 1], &order[k]) == 0) continue;
            r++;
            for (i = 0; i < plen[ph]; i++) {
                d[at] = v[pstart[ph] + i]; ds[at] = (int32_t)d[at] + (int32_t)D + 1;
                dphr[at] = r; dsuf[at] = plen[ph] - (int32_t)i; at++;
            }
            d[at] = 1; ds[at] = r; dphr[at] = r; dsuf[at] = 0; at++;         /* EndOfWord number r: unique, ordered by position */
        }
        d[at] = 0; ds[at] = 0; dphr[at] = 0; dsuf[at] = 0; at++;
        if (at != nd) return -4;
    }
    if (sais_i32(ds, saD, (int32_t)nd, (int32_t)D + 258)) return -1;
    for (i = 0; i < nd; i++) isaD[saD[i]] = (int32_t)i;
    {
        int64_t h = 0;
        lcpD[0] = 0;
        for (i = 0; i < nd; i++) {
            const int64_t r = isaD[i]; int64_t q;
            if (r == 0) { h = 0; continue; }
            q = saD[r - 1];
            while (i + h < nd && q + h < nd && ds[i + h] == ds[q + h]) h++;
            lcpD[r] = (int32_t)h;
            if (h > 0) h--;
        }
    }
    free(isaD); free(ds);
    /* parse: suffix array, inverse, inverted lists (parse.hpp:85, :107-132) */
    saP = (int32_t *)malloc(((size_t)m + 1) * 4); isaP = (int32_t *)malloc(((size_t)m + 1) * 4);
    ilist = (int32_t *)malloc(((size_t)m + 1) * 4); ilist_start = (int32_t *)calloc((size_t)D + 3, 4);
    posval = (int64_t *)malloc(((size_t)m + 1) * 8); slcp = (int32_t *)calloc((size_t)m + 1, 4);
    if (!saP || !isaP || !ilist || !ilist_start || !posval || !slcp) return -1;
    if (sais_i32(parse, saP, (int32_t)(m + 1), (int32_t)D + 1)) return -1;
    for (i = 0; i <= m; i++) isaP[saP[i]] = (int32_t)i;
    for (i = 0; i <= m; i++) ilist_start[parse[i] + 1]++;
    for (i = 0; i <= D; i++) ilist_start[i + 1] += ilist_start[i];
    {
        int32_t *fill = (int32_t *)calloc((size_t)D + 2, 4);
        if (!fill) return -1;
        for (i = 0; i <= m; i++) {                                    /* rank i follows an occurrence of the phrase before it */
            const int64_t before = (saP[i] == 0 ? m + 1 : saP[i]) - 1;
            const int32_t ph = parse[before];
            ilist[ilist_start[ph] + fill[ph]++] = (int32_t)i;
        }
        free(fill);
    }
    /* text position of the character after the w shared ones of the phrase that begins parse suffix q, minus 1 + w:
       an occurrence of a phrase suffix of length L before parse suffix q begins at text position posval - L
       (pfp.hpp:185-201 with pfp_lcp_mum.hpp:337-341, in the coordinates of T) */
    for (i = 0; i <= m; i++) {
        const int64_t q = saP[i];
        const int64_t s_q = q < m ? pstart[q] : nv - w;               /* the end of the text for the terminator of the parse */
        posval[i] = s_q + w - 1;
    }
    /* s_lcp_T (pfp.hpp:210-244): LCP in characters of parse suffixes adjacent in rank, the w shared characters counted once */
    {
        int64_t l = 0, lt = 0;
        for (i = 0; i < m; i++) {
            const int64_t r = isaP[i]; int64_t q, a, b, c = 0;
            if (r == 0) { l = 0; lt = 0; continue; }
            q = saP[r - 1];
            while (parse[i + l] == parse[q + l]) { lt += plen[i + l] - w; l++; }
            a = i + l; b = q + l;
            if (parse[a] != 0 && parse[b] != 0) {
                const uint8_t *x = v + pstart[a], *y = v + pstart[b];
                const int64_t lim = plen[a] < plen[b] ? plen[a] : plen[b];
                while (c < lim && x[c] == y[c]) c++;
            }
            slcp[r] = (int32_t)(lt + c);
            if (l > 0) { l--; lt -= plen[i] - w; }
        }
    }
    if (rmq_build(&rq, slcp, m + 1)) return -1;
    /* emitter */
    {
        int64_t cur = 1, prev_i = 0, prev_L = -1, prev_phrase = 0;
        while (cur < nd) {
            const int64_t sn = saD[cur]; const int32_t L = dsuf[sn];
            const int valid = L >= w && sn > 0 && d[sn - 1] != 1 && d[sn] > 1;      /* proper suffix of a phrase, at least w long */
            int64_t nxt, nmem = 0, lcp_first, prev_occ = 0; int first = 1;
            if (!valid) { cur++; continue; }
            /* members: the entries that follow with the same suffix (pfp_lcp_mum.hpp:128-139) */
            for (nxt = cur; nxt < nd && (nxt == cur || lcpD[nxt] >= L); nxt++) {
                const int64_t s2 = saD[nxt];
                if (dsuf[s2] != L) continue;
                if (nmem == heap_cap) { heap_cap = heap_cap ? heap_cap * 2 : 16; heap = (pfp_run *)realloc(heap, (size_t)heap_cap * sizeof(pfp_run)); if (!heap) return -1; }
                heap[nmem].at = ilist_start[dphr[s2]]; heap[nmem].end = ilist_start[dphr[s2] + 1];
                heap[nmem].val = ilist[heap[nmem].at];
                heap[nmem].bwt = d[s2 - 1] == 2 ? 0 : d[s2 - 1];      /* Dollar before text position 0 (pfp_lcp_mum.hpp:268) */
                nmem++;
            }
            /* LCP with the entry before the group (pfp_lcp_mum.hpp:295-321) */
            lcp_first = 0;
            if (j > 0) {
                int64_t t; int32_t mn = lcpD[cur];
                for (t = prev_i + 1; t < cur; t++) if (lcpD[t] < mn) mn = lcpD[t];
                lcp_first = mn;
                if (mn >= L && L == prev_L) {
                    int64_t left = ilist[ilist_start[dphr[sn]]], right = ilist[ilist_start[prev_phrase + 1] - 1];
                    if (left > right) { const int64_t t2 = left; left = right; right = t2; }
                    lcp_first += rmq_min(&rq, left + 1, right) - w;
                }
            }
            {
                int64_t last_member = cur;
                for (k = cur; k < nxt; k++) if (dsuf[saD[k]] == L) last_member = k;
                for (k = nmem / 2 - 1; k >= 0; k--) heap_sift(heap, nmem, k);
                while (nmem) {
                    const int64_t occ = heap[0].val; int64_t val;
                    if (first) val = lcp_first;
                    else {
                        int64_t lo = occ < prev_occ ? occ : prev_occ, hi = occ < prev_occ ? prev_occ : occ;
                        val = L + rmq_min(&rq, lo + 1, hi) - w;
                    }
                    first = 0;
                    if (j > n) return -5;
                    sa[j] = (posval[occ] - L) % (n + 1); lcp[j] = val; bwt[j] = heap[0].bwt; j++;
                    prev_occ = occ;
                    if (++heap[0].at != heap[0].end) heap[0].val = ilist[heap[0].at];
                    else { heap[0] = heap[nmem - 1]; nmem--; }
                    heap_sift(heap, nmem, 0);
                }
                prev_i = last_member; prev_L = L; prev_phrase = dphr[saD[last_member]];
            }
            cur = nxt;
        }
    }
    if (stats) { stats[0] = m; stats[1] = D; stats[2] = nd; stats[3] = j; }
    rmq_free(&rq); free(heap);
    free(v); free(pstart); free(plen); free(order); free(rank_of); free(parse); free(d); free(saD); free(lcpD);
    free(dphr); free(dsuf); free(saP); free(isaP); free(ilist); free(ilist_start); free(posval); free(slcp);
    return j == n + 1 ? 0 : -6;
}

/* ------------------------------------------------------------------------ */
/* whole job for the CPU baseline leg of bench.py                            */
/* ------------------------------------------------------------------------ */
static double now_sec(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int64_t run_job_route(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs, const mmo_scan_params *p,
                             int64_t pfp_w, int64_t pfp_p, double *stage_sec, char **out_text, int64_t *out_len);
int64_t mmo_run_job(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs,
                    const mmo_scan_params *p, double *stage_sec, char **out_text, int64_t *out_len) {
    return run_job_route(bases, doc_len, n_docs, p, 0, 0, stage_sec, out_text, out_len);
}
int64_t mmo_run_job_pfp(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs, const mmo_scan_params *p,
                        int64_t pfp_w, int64_t pfp_p, double *stage_sec, char **out_text, int64_t *out_len) {
    return run_job_route(bases, doc_len, n_docs, p, pfp_w, pfp_p, stage_sec, out_text, out_len);
}
static int64_t run_job_route(const uint8_t *bases, const int64_t *doc_len, int64_t n_docs, const mmo_scan_params *p,
                             int64_t pfp_w, int64_t pfp_p, double *stage_sec, char **out_text, int64_t *out_len) {
    int64_t n = mmo_text_length(doc_len, n_docs, p->revcomp), m = n + 1;
    uint8_t *text = (uint8_t *)malloc((size_t)n + 1), *bwt = (uint8_t *)malloc((size_t)m);
    int64_t *doc_start = (int64_t *)malloc(((size_t)n_docs + 1) * 8);
    int64_t *sa = (int64_t *)malloc((size_t)m * 8), *lcp = (int64_t *)malloc((size_t)m * 8);
    int32_t *doc = (int32_t *)malloc((size_t)m * 4);
    mmo_result *r; double t0 = now_sec(), t1, t2, t3;
    mmo_build_text(bases, doc_len, n_docs, p->revcomp, text, doc_start);
    t1 = now_sec();
    if (pfp_w > 0 ? mmo_build_stream_pfp(text, n, pfp_w, pfp_p, sa, lcp, bwt, NULL) : mmo_build_stream(text, n, sa, lcp, bwt)) return -1;
    mmo_doc_array(sa, m, doc_start, n_docs, doc);
    t2 = now_sec();
    r = mmo_scan(sa, lcp, bwt, doc, m, doc_start, n_docs, p);
    *out_text = mmo_format_text(r, out_len);
    t3 = now_sec();
    stage_sec[0] = t1 - t0; stage_sec[1] = t2 - t1; stage_sec[2] = t3 - t2;
    mmo_result_free(r);
    free(text); free(bwt); free(doc_start); free(sa); free(lcp); free(doc);
    return n;
}
