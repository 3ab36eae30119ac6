// rows_kernels.hip -- A6 on the device: accepted intervals -> coordinates ->
// library arrays and the exact bytes of PREFIX.mums / PREFIX.mems.
//
// Restates write_mum (include/mem_finder.hpp:357-428), write_mem (:210-263) and the
// library collectors (mumemto_library/mumemto_api.cpp:137-166, 241-286).  One wave per
// row; a "measure" pass sizes every row (and applies write_mum's two drop rules), two
// prefix sums place rows and bytes, a "write" pass emits them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "device_utils.hpp"
#include "rows_kernels.hpp"
#include "textref.hpp"

namespace mmt { namespace rk {

// A launch may not have 2^32 work-items or more (HIP folds the product of grid and workgroup size into 32 bits: a
// larger launch silently runs a fraction of its workgroups).  Every kernel here uses workgroups of at most 256
// work-items with grid_for, so 2^24 workgroups is the limit; kernels over text-sized ranges handle 4 - 16 items per
// work-item and stay below it for any text that fits the device.
static inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}

__device__ __forceinline__ uint32_t doc_lookup(const uint64_t* __restrict__ start, uint32_t n_docs, uint64_t p) {
    uint32_t lo = 0, hi = n_docs;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (start[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}
__device__ __forceinline__ uint32_t ndigits(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) { v /= 10; d++; }
    return d;
}
__device__ __forceinline__ void put_uint(char* dst, uint64_t v, uint32_t nd) {
    for (uint32_t i = nd; i-- > 0;) { dst[i] = (char)('0' + v % 10); v /= 10; }
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { uint32_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if (lane >= (uint32_t)o) x += y; }
    total = __shfl(x, 63, 64);
    return x - v;
}

// key for the pop order of the reference's stack: closing position ascending, longer first
__global__ void k_row_keys(const k::Row* __restrict__ rows, uint32_t n_rows, uint64_t* __restrict__ keys,
                           uint32_t* __restrict__ vals) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const k::Row c = rows[r];
    keys[r] = ((c.start + c.cnt - 1) << 32) | (uint32_t)(~c.len);
    vals[r] = r;
}
void row_keys(const k::Row* rows, uint32_t n_rows, uint64_t* keys, uint32_t* vals, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_row_keys, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, rows, n_rows, keys, vals);
    MMT_HIP(hipGetLastError());
}
__global__ void k_row_len_keys(const k::Row* __restrict__ rows, uint32_t n_rows, uint32_t* __restrict__ keys,
                               uint32_t* __restrict__ vals) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    keys[r] = ~rows[r].len;
    vals[r] = r;
}
void row_len_keys(const k::Row* rows, uint32_t n_rows, uint32_t* keys, uint32_t* vals, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_row_len_keys, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, rows, n_rows, keys, vals);
    MMT_HIP(hipGetLastError());
}
__global__ void k_row_end_keys(const k::Row* __restrict__ rows, const uint32_t* __restrict__ order, uint32_t n_rows,
                               uint64_t* __restrict__ keys) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const k::Row c = rows[order[r]];
    keys[r] = c.start + c.cnt - 1;
}
void row_end_keys(const k::Row* rows, const uint32_t* order, uint32_t n_rows, uint64_t* keys, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_row_end_keys, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, rows, order, n_rows, keys);
    MMT_HIP(hipGetLastError());
}

template <typename SA>
struct RowArgsT {
    const k::Row* rows; const uint32_t* order; uint32_t n_rows; SA sa; const uint64_t* doc_start; const uint64_t* doc_len;
    uint32_t n_docs; int revcomp;
    explicit RowArgsT(const RowArgs& a)
        : rows(a.rows), order(a.order), n_rows(a.n_rows), sa(SA(a.sa)), doc_start(a.doc_start), doc_len(a.doc_len),
          n_docs(a.n_docs), revcomp(a.revcomp) {}
};

// ---- MUM mode --------------------------------------------------------------------
// slot arrays (n_rows x n_docs) must be pre-set: offsets = -1, strands = 0.
template <typename SA>
__global__ void k_mum_measure(RowArgsT<SA> a, int64_t* __restrict__ slot_off, uint8_t* __restrict__ slot_st,
                              uint32_t* __restrict__ keep, uint32_t* __restrict__ text_len) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rows) return;
    const k::Row c = a.rows[a.order[r]];
    const uint32_t cnt = c.cnt, N = a.n_docs;
    const uint64_t len = c.len;
    uint32_t drop = 0, digits = 0, present = 0;
    uint32_t first_key = 0xffffffffu;       // (doc << 1 | minus), smallest doc among 0..N-2
    uint32_t last_minus = 0;                // strand of doc N-1 if present
    for (uint32_t base = 0; base < cnt; base += 64) {
        if (base + lane < cnt) {                                     // write_mum, mem_finder.hpp:365-380
            const uint64_t sa = a.sa.get(c.start + base + lane);
            const uint32_t d = doc_lookup(a.doc_start, N, sa);
            const uint64_t half = a.doc_len[d] + 1;
            uint64_t pos = sa - a.doc_start[d];
            uint32_t minus = 0;
            if (a.revcomp && pos >= half) {
                minus = 1;
                if (pos + len >= 2 * half) drop = 1;
                pos = 2 * half - pos - len - 1;
            }
            if (!drop) {
                slot_off[r * N + d] = (int64_t)pos;
                slot_st[r * N + d] = minus ? 0 : 1;
                digits += ndigits(pos); present++;
            }
            if (d + 1 < N) { uint32_t key = (d << 1) | minus; first_key = key < first_key ? key : first_key; }
            else last_minus = minus;
        }
    }
    drop = wave_sum(drop); digits = wave_sum(digits); present = wave_sum(present);
    first_key = wave_min(first_key); last_minus = wave_sum(last_minus);
    // first present document on '-' -> row not written (mem_finder.hpp:382-391)
    const bool first_minus = first_key != 0xffffffffu ? (first_key & 1u) : (last_minus != 0);
    const bool kept = drop == 0 && !first_minus;
    if (lane == 0) {
        keep[r] = kept ? 1u : 0u;
        // LEN \t offs(N-1 commas) \t strands(N-1 commas) \n
        text_len[r] = kept ? ndigits(len) + 1 + digits + (N - 1) + 1 + present + (N - 1) + 1 : 0u;
    }
}

template <typename SA>
__global__ void k_mum_write(RowArgsT<SA> a, const int64_t* __restrict__ slot_off, const uint8_t* __restrict__ slot_st,
                            const uint32_t* __restrict__ keep, const uint32_t* __restrict__ row_idx,
                            const uint64_t* __restrict__ text_off, uint32_t* __restrict__ out_len,
                            int64_t* __restrict__ out_off, uint8_t* __restrict__ out_st, char* __restrict__ text) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rows || !keep[r]) return;
    const uint32_t N = a.n_docs, ro = row_idx[r];
    const uint64_t len = a.rows[a.order[r]].len;
    char* t = text + text_off[r];
    const uint32_t nl = ndigits(len);
    if (lane == 0) { put_uint(t, len, nl); t[nl] = '\t'; out_len[ro] = (uint32_t)len; }
    uint32_t cur = nl + 1;
    for (uint32_t base = 0; base < N; base += 64) {                 // offsets field (mem_finder.hpp:406-421)
        const uint32_t d = base + lane;
        int64_t off = -1; uint8_t st = 0; uint32_t wd = 0, nd = 0;
        if (d < N) {
            off = slot_off[r * N + d]; st = slot_st[r * N + d];
            out_off[(uint64_t)ro * N + d] = off; out_st[(uint64_t)ro * N + d] = st;
            nd = off >= 0 ? ndigits((uint64_t)off) : 0;
            wd = nd + (d + 1 < N ? 1 : 0);
        }
        uint32_t total;
        const uint32_t at = wave_excl_sum(wd, lane, total);
        if (d < N) {
            if (nd) put_uint(t + cur + at, (uint64_t)off, nd);
            if (d + 1 < N) t[cur + at + nd] = ',';
        }
        cur += total;
    }
    if (lane == 0) t[cur] = '\t';
    cur += 1;
    for (uint32_t base = 0; base < N; base += 64) {                 // strands field
        const uint32_t d = base + lane;
        uint32_t wd = 0, pres = 0; uint8_t st = 0;
        if (d < N) { pres = slot_off[r * N + d] >= 0; st = slot_st[r * N + d]; wd = pres + (d + 1 < N ? 1 : 0); }
        uint32_t total;
        const uint32_t at = wave_excl_sum(wd, lane, total);
        if (d < N) {
            if (pres) t[cur + at] = st ? '+' : '-';
            if (d + 1 < N) t[cur + at + pres] = ',';
        }
        cur += total;
    }
    if (lane == 0) t[cur] = '\n';
}

// ---- MEM mode ---------------------------------------------------------------------
template <typename SA>
__device__ __forceinline__ void mem_occurrence(const RowArgsT<SA>& a, const k::Row& c, uint32_t k, uint64_t& pos,
                                               uint32_t& d, uint32_t& minus) {
    const uint32_t cnt = c.cnt;
    const uint64_t sa = a.sa.get(c.start + k);
    d = doc_lookup(a.doc_start, a.n_docs, sa);
    const uint64_t half = a.doc_len[d] + 1;
    pos = sa - a.doc_start[d];
    minus = 0;
    if (a.revcomp && pos >= half) {                                 // size_t arithmetic, may wrap (:229, :248)
        minus = 1;
        pos = 2 * half - pos - (uint64_t)c.len - (k + 1 == cnt ? 0 : 1);
    }
}

template <typename SA>
__global__ void k_mem_measure(RowArgsT<SA> a, uint32_t* __restrict__ occ_cnt, uint32_t* __restrict__ text_len,
                              uint32_t* __restrict__ w_pos, uint32_t* __restrict__ w_doc) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rows) return;
    const k::Row c = a.rows[a.order[r]];
    const uint32_t cnt = c.cnt;
    uint32_t dp = 0, dd = 0;
    for (uint32_t base = 0; base < cnt; base += 64) {
        if (base + lane < cnt) {
            uint64_t pos; uint32_t d, minus;
            mem_occurrence(a, c, base + lane, pos, d, minus);
            dp += ndigits(pos); dd += ndigits(d);
        }
    }
    dp = wave_sum(dp); dd = wave_sum(dd);
    if (lane == 0) {
        occ_cnt[r] = cnt;
        w_pos[r] = dp + (cnt - 1); w_doc[r] = dd + (cnt - 1);
        text_len[r] = ndigits(c.len) + 1 + dp + (cnt - 1) + 1 + dd + (cnt - 1) + 1 + cnt + (cnt - 1) + 1;
    }
}

template <typename SA>
__global__ void k_mem_write(RowArgsT<SA> a, const uint64_t* __restrict__ occ_off, const uint64_t* __restrict__ text_off,
                            const uint32_t* __restrict__ w_pos, const uint32_t* __restrict__ w_doc,
                            uint32_t* __restrict__ out_len, int64_t* __restrict__ out_off,
                            uint64_t* __restrict__ out_doc, uint8_t* __restrict__ out_st, char* __restrict__ text) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= a.n_rows) return;
    const k::Row c = a.rows[a.order[r]];
    const uint32_t cnt = c.cnt;
    char* t = text + text_off[r];
    const uint32_t nl = ndigits(c.len);
    if (lane == 0) { put_uint(t, c.len, nl); t[nl] = '\t'; out_len[r] = c.len; }
    const uint32_t p0 = nl + 1, d0 = p0 + w_pos[r] + 1, s0 = d0 + w_doc[r] + 1;
    if (lane == 0) { t[d0 - 1] = '\t'; t[s0 - 1] = '\t'; t[s0 + cnt + (cnt - 1)] = '\n'; }
    uint32_t cp = 0, cd = 0;
    const uint64_t ob = occ_off[r];
    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t k2 = base + lane;
        uint64_t pos = 0; uint32_t d = 0, minus = 0, wp = 0, wd = 0, np = 0, nd = 0;
        const bool have = k2 < cnt;
        if (have) {
            mem_occurrence(a, c, k2, pos, d, minus);
            np = ndigits(pos); nd = ndigits(d);
            const uint32_t comma = k2 + 1 < cnt ? 1 : 0;
            wp = np + comma; wd = nd + comma;
            out_off[ob + k2] = (int64_t)pos; out_doc[ob + k2] = d; out_st[ob + k2] = minus ? 0 : 1;
        }
        uint32_t tp, td;
        const uint32_t ap = wave_excl_sum(wp, lane, tp), ad = wave_excl_sum(wd, lane, td);
        if (have) {
            put_uint(t + p0 + cp + ap, pos, np);
            put_uint(t + d0 + cd + ad, d, nd);
            t[s0 + 2 * k2] = minus ? '-' : '+';
            if (k2 + 1 < cnt) { t[p0 + cp + ap + np] = ','; t[d0 + cd + ad + nd] = ','; t[s0 + 2 * k2 + 1] = ','; }
        }
        cp += tp; cd += td;
    }
}

void mum_measure(const RowArgs& a, int64_t* slot_off, uint8_t* slot_st, uint32_t* keep, uint32_t* text_len,
                 hipStream_t s) {
    if (!a.n_rows) return;
    if (a.sa.wide())
        hipLaunchKernelGGL(k_mum_measure<Sa40>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa40>(a), slot_off, slot_st, keep, text_len);
    else
        hipLaunchKernelGGL(k_mum_measure<Sa32>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa32>(a), slot_off, slot_st, keep, text_len);
    MMT_HIP(hipGetLastError());
}
void mum_write(const RowArgs& a, const int64_t* slot_off, const uint8_t* slot_st, const uint32_t* keep,
               const uint32_t* row_idx, const uint64_t* text_off, uint32_t* out_len, int64_t* out_off, uint8_t* out_st,
               char* text, hipStream_t s) {
    if (!a.n_rows) return;
    if (a.sa.wide())
        hipLaunchKernelGGL(k_mum_write<Sa40>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa40>(a), slot_off, slot_st, keep, row_idx, text_off, out_len, out_off, out_st, text);
    else
        hipLaunchKernelGGL(k_mum_write<Sa32>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa32>(a), slot_off, slot_st, keep, row_idx, text_off, out_len, out_off, out_st, text);
    MMT_HIP(hipGetLastError());
}
void mem_measure(const RowArgs& a, uint32_t* occ_cnt, uint32_t* text_len, uint32_t* w_pos, uint32_t* w_doc,
                 hipStream_t s) {
    if (!a.n_rows) return;
    if (a.sa.wide())
        hipLaunchKernelGGL(k_mem_measure<Sa40>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa40>(a), occ_cnt, text_len, w_pos, w_doc);
    else
        hipLaunchKernelGGL(k_mem_measure<Sa32>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa32>(a), occ_cnt, text_len, w_pos, w_doc);
    MMT_HIP(hipGetLastError());
}
void mem_write(const RowArgs& a, const uint64_t* occ_off, const uint64_t* text_off, const uint32_t* w_pos,
               const uint32_t* w_doc, uint32_t* out_len, int64_t* out_off, uint64_t* out_doc, uint8_t* out_st,
               char* text, hipStream_t s) {
    if (!a.n_rows) return;
    if (a.sa.wide())
        hipLaunchKernelGGL(k_mem_write<Sa40>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa40>(a), occ_off, text_off, w_pos, w_doc, out_len, out_off, out_doc, out_st, text);
    else
        hipLaunchKernelGGL(k_mem_write<Sa32>, dim3(grid_for((uint64_t)a.n_rows * 64, 256)), dim3(256), 0, s,
                           RowArgsT<Sa32>(a), occ_off, text_off, w_pos, w_doc, out_len, out_off, out_doc, out_st, text);
    MMT_HIP(hipGetLastError());
}

__global__ void k_widen(const uint32_t* __restrict__ in, uint32_t n, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
void widen(const uint32_t* in, uint32_t n, uint64_t* out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_widen, dim3(grid_for(n, 256)), dim3(256), 0, s, in, n, out);
    MMT_HIP(hipGetLastError());
}

// ---- the row tap (tests: precision AND recall inside whole bins of a full-size run) ----------------------------------------
// Every accepted interval whose match begins with one of a few given k-mers is copied out -- length and all suffix-array
// entries -- before its window drops it.  kmers: n_kmers x 2 words, the k <= 16 characters little-endian as tx_load8 reads them
// (second word zero beyond 8 characters).  One work-item per fresh row: it reads the text at the row's first occurrence.
__device__ __forceinline__ int tap_match(const TextRef& T, uint64_t pos, const uint64_t* __restrict__ kmers, uint32_t n_kmers, uint32_t k) {
    if (pos + k > T.n) return -1;
    uint64_t a = tx_load8(T, pos + 1), b = k > 8 ? tx_load8(T, pos + 9) : 0ull;       // V index = text position + 1
    if (k < 8) a &= (1ull << (8 * k)) - 1ull;
    if (k > 8 && k < 16) b &= (1ull << (8 * (k - 8))) - 1ull;
    for (uint32_t i = 0; i < n_kmers; i++)
        if (kmers[2 * i] == a && kmers[2 * i + 1] == b) return (int)i;
    return -1;
}
template <typename SA>
__global__ void k_tap_rows(const k::Row* __restrict__ rows, uint32_t n_rows, SA pool, TextRef T, const uint64_t* __restrict__ kmers,
                           uint32_t n_kmers, uint32_t k, uint32_t* __restrict__ t_len, uint64_t* __restrict__ t_off,
                           uint32_t* __restrict__ t_cnt, uint64_t* __restrict__ t_sa, unsigned long long* __restrict__ used,
                           uint64_t cap_rows, uint64_t cap_occ) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const k::Row row = rows[r];
    if (row.len < k || !row.cnt) return;
    const uint64_t first = (uint64_t)pool.get(row.start);
    if (tap_match(T, first, kmers, n_kmers, k) < 0) return;
    const unsigned long long slot = atomicAdd(used, 1ull);
    const unsigned long long off = atomicAdd(used + 1, (unsigned long long)row.cnt);
    if (slot >= cap_rows || off + row.cnt > cap_occ) return;       // (counted: the caller sees used > capacity)
    t_len[slot] = row.len; t_off[slot] = off; t_cnt[slot] = row.cnt;
    for (uint32_t i = 0; i < row.cnt; i++) t_sa[off + i] = (uint64_t)pool.get(row.start + i);
}
void tap_rows(const k::Row* rows, uint32_t n_rows, SaCol pool, const TextRef& T, const uint64_t* kmers, uint32_t n_kmers, uint32_t k,
              uint32_t* t_len, uint64_t* t_off, uint32_t* t_cnt, uint64_t* t_sa, uint64_t* used, uint64_t cap_rows, uint64_t cap_occ,
              hipStream_t s) {
    if (!n_rows) return;
    const unsigned grid = (n_rows + 255) / 256;
    if (pool.wide())
        hipLaunchKernelGGL(k_tap_rows<Sa40>, dim3(grid), dim3(256), 0, s, rows, n_rows, Sa40(pool), T, kmers, n_kmers, k, t_len, t_off,
                           t_cnt, t_sa, reinterpret_cast<unsigned long long*>(used), cap_rows, cap_occ);
    else
        hipLaunchKernelGGL(k_tap_rows<Sa32>, dim3(grid), dim3(256), 0, s, rows, n_rows, Sa32(pool), T, kmers, n_kmers, k, t_len, t_off,
                           t_cnt, t_sa, reinterpret_cast<unsigned long long*>(used), cap_rows, cap_occ);
    MMT_HIP(hipGetLastError());
}
// every text position whose suffix begins with one of the k-mers: (position, which k-mer), in no particular order; 16 positions
// per work-item, slices of 2^28 work-items (a launch may not have 2^32)
__global__ void k_kmer_positions(TextRef T, uint64_t first, uint64_t n_items, const uint64_t* __restrict__ kmers, uint32_t n_kmers,
                                 uint32_t k, uint64_t* __restrict__ out_pos, uint32_t* __restrict__ out_which,
                                 unsigned long long* __restrict__ used, uint64_t cap) {
    const uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const uint64_t p0 = (first + it) * 16;
    for (uint32_t q = 0; q < 16; q++) {
        const uint64_t p = p0 + q;
        if (p >= T.n) break;
        const int w = tap_match(T, p, kmers, n_kmers, k);
        if (w < 0) continue;
        const unsigned long long at = atomicAdd(used, 1ull);
        if (at < cap) { out_pos[at] = p; out_which[at] = (uint32_t)w; }
    }
}
void kmer_positions(const TextRef& T, const uint64_t* kmers, uint32_t n_kmers, uint32_t k, uint64_t* out_pos, uint32_t* out_which,
                    uint64_t* used, uint64_t cap, hipStream_t s) {
    const uint64_t items = (T.n + 15) / 16, SLICE = 1ull << 28;
    for (uint64_t f = 0; f < items; f += SLICE) {
        const uint64_t cnt = std::min<uint64_t>(SLICE, items - f);
        hipLaunchKernelGGL(k_kmer_positions, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, T, f, cnt, kmers, n_kmers, k, out_pos,
                           out_which, reinterpret_cast<unsigned long long*>(used), cap);
    }
    MMT_HIP(hipGetLastError());
}

}}  // namespace mmt::rk
