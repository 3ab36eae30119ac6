// merge_kernels.hip -- see merge_kernels.hpp.  Everything here is proportional to the number of
// MUM rows (10^5..10^6), not to the anchor length; the O(L_0) walk is k_fold_step in kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "device_utils.hpp"
#include "merge_kernels.hpp"

namespace mmt { namespace mk {

// A launch may not have 2^32 work-items or more (HIP folds the product of grid and workgroup size into 32 bits: a
// larger launch silently runs a fraction of its workgroups).  Every kernel here uses workgroups of at most 256
// work-items with grid_for, so 2^24 workgroups is the limit; kernels over text-sized ranges handle 4 - 16 items per
// work-item and stay below it for any text that fits the device.
static inline unsigned grid_for(uint64_t items, unsigned per_block) {
    uint64_t g = (items + per_block - 1) / per_block;
    if (g >= (1ull << 24)) throw HipError("kernel launch of 2^32 work-items or more (" + std::to_string(items) + " items)");
    return (unsigned)(g ? g : 1);
}
// for grid-stride kernels
static inline unsigned grid_capped(uint64_t items, unsigned per_block) {
    const uint64_t g = (items + per_block - 1) / per_block;
    return (unsigned)(g ? (g < (1ull << 20) ? g : (1ull << 20)) : 1);
}

__global__ void k_leaf_keys(const int64_t* __restrict__ offsets, uint32_t n_rows, uint32_t n_docs, uint64_t L,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t o = offsets[(uint64_t)r * n_docs];
    uint32_t b = 0;
    if (o < 0 || (uint64_t)o >= L) b |= 1u;
    if (r && offsets[(uint64_t)(r - 1) * n_docs] > o) b |= 2u;
    if (b) atomicOr(bad, b);
    keys[r] = (uint64_t)o;
    vals[r] = r;
}
void leaf_keys(const int64_t* offsets, uint32_t n_rows, uint32_t n_docs, uint64_t L, uint64_t* keys, uint32_t* vals,
               uint32_t* bad, hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_leaf_keys, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, offsets, n_rows, n_docs, L, keys,
                       vals, bad);
    MMT_HIP(hipGetLastError());
}

__global__ void k_leaf_side(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ order,
                            const uint32_t* __restrict__ length, uint32_t n_rows, SideView out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t s = order[r];
    out.start[r] = keys[r]; out.len[r] = length[s]; out.src[r] = s; out.plus[r] = 0; out.minus[r] = 0;
}
void leaf_side(const uint64_t* keys, const uint32_t* order, const uint32_t* length, uint32_t n_rows, SideView out,
               hipStream_t s) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_leaf_side, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, keys, order, length, n_rows, out);
    MMT_HIP(hipGetLastError());
}

// One thread per (new row, source partition).  The trims of this step are added to the shifts of every
// partition already folded into the left side (merge_candidates.cpp:97-104, :142-151).
__global__ void k_fold_rows(const uint32_t* __restrict__ order, const uint64_t* __restrict__ pos,
                            const uint32_t* __restrict__ ra_, const uint32_t* __restrict__ rb_,
                            const uint32_t* __restrict__ nl_, uint32_t found, SideView left, SideView right,
                            SideView out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t np = out.n_parts;
    if (t >= (uint64_t)found * np) return;
    const uint32_t q = (uint32_t)(t / np), g = (uint32_t)(t % np);
    const uint32_t e = order[q], nl = nl_[e], ra = ra_[e], rb = rb_[e];
    const int64_t i = (int64_t)pos[e];
    if (g == 0) { out.start[q] = (uint64_t)i; out.len[q] = nl; }
    if (g + 1 < np) {
        const int64_t d1 = i - (int64_t)left.start[ra];
        const int64_t s1 = (int64_t)left.len[ra] - d1;
        const uint64_t li = (uint64_t)ra * left.n_parts + g;
        out.src[t] = left.src[li];
        out.plus[t] = left.plus[li] + d1;
        out.minus[t] = left.minus[li] + (s1 - (int64_t)nl);
    } else {
        const int64_t d2 = i - (int64_t)right.start[rb];
        const int64_t s2 = (int64_t)right.len[rb] - d2;
        out.src[t] = right.src[rb];
        out.plus[t] = right.plus[rb] + d2;
        out.minus[t] = right.minus[rb] + (s2 - (int64_t)nl);
    }
}
void fold_rows(const uint32_t* order, const uint64_t* pos, const uint32_t* ra, const uint32_t* rb,
               const uint32_t* nl, uint32_t found, SideView left, SideView right, SideView out, hipStream_t s) {
    if (!found) return;
    hipLaunchKernelGGL(k_fold_rows, dim3(grid_for((uint64_t)found * out.n_parts, 256)), dim3(256), 0, s, order, pos, ra,
                       rb, nl, found, left, right, out);
    MMT_HIP(hipGetLastError());
}

__global__ void k_materialise(SideView side, const uint32_t* __restrict__ perm, const PartTable* __restrict__ parts,
                              uint32_t n_docs_out, const uint32_t* __restrict__ col_part,
                              uint32_t* __restrict__ out_len, int64_t* __restrict__ out_off,
                              uint8_t* __restrict__ out_st) {
    // grid-stride: 45 million rows x 94 documents are more cells than one launch has work-items
    const uint64_t total = (uint64_t)side.n * n_docs_out, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint32_t i = (uint32_t)(t / n_docs_out), col = (uint32_t)(t % n_docs_out);
        const uint32_t row = perm ? perm[i] : i;
        const uint32_t g = col_part[col];
        const PartTable P = parts[g];
        const uint32_t c = col - P.first_col + P.skip;
        const uint64_t li = (uint64_t)row * side.n_parts + g;
        const uint64_t cell = (uint64_t)side.src[li] * P.n_docs + c;
        const uint8_t sd = P.strands[cell];
        out_off[t] = P.offsets[cell] + (sd ? side.plus[li] : side.minus[li]);
        out_st[t] = sd;
        if (col == 0) out_len[i] = side.len[row];
    }
}
void materialise(SideView side, const uint32_t* perm, const PartTable* parts, uint32_t n_docs_out,
                 const uint32_t* col_part, uint32_t* out_len, int64_t* out_off, uint8_t* out_st, hipStream_t s) {
    if (!side.n) return;
    hipLaunchKernelGGL(k_materialise, dim3(grid_capped((uint64_t)side.n * n_docs_out, 256)), dim3(256), 0, s, side, perm,
                       parts, n_docs_out, col_part, out_len, out_off, out_st);
    MMT_HIP(hipGetLastError());
}

__global__ void k_permute_rows(const uint32_t* __restrict__ perm, uint32_t n, uint32_t n_docs,
                               const uint32_t* __restrict__ in_len, const int64_t* __restrict__ in_off,
                               const uint8_t* __restrict__ in_st, uint32_t* __restrict__ out_len,
                               int64_t* __restrict__ out_off, uint8_t* __restrict__ out_st) {
    const uint64_t total = (uint64_t)n * n_docs, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint32_t i = (uint32_t)(t / n_docs), col = (uint32_t)(t % n_docs);
        const uint64_t from = (uint64_t)perm[i] * n_docs + col;
        out_off[t] = in_off[from]; out_st[t] = in_st[from];
        if (col == 0) out_len[i] = in_len[perm[i]];
    }
}
void permute_rows(const uint32_t* perm, uint32_t n, uint32_t n_docs, const uint32_t* in_len, const int64_t* in_off,
                  const uint8_t* in_st, uint32_t* out_len, int64_t* out_off, uint8_t* out_st, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_permute_rows, dim3(grid_capped((uint64_t)n * n_docs, 256)), dim3(256), 0, s, perm, n, n_docs,
                       in_len, in_off, in_st, out_len, out_off, out_st);
    MMT_HIP(hipGetLastError());
}

// rows of a coordinate-range fold: flags[r] = the anchor offset of row r (+ shift) lies in [lo, hi)
__global__ void k_range_flags(const int64_t* __restrict__ off, uint32_t n, uint32_t n_docs, int64_t shift, int64_t lo, int64_t hi,
                              uint8_t* __restrict__ flags) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t a = off[(uint64_t)r * n_docs] + shift;
    flags[r] = a >= lo && a < hi ? 1 : 0;
}
void range_flags(const int64_t* off, uint32_t n, uint32_t n_docs, int64_t shift, int64_t lo, int64_t hi, uint8_t* flags,
                 hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_range_flags, dim3((n + 255) / 256), dim3(256), 0, s, off, n, n_docs, shift, lo, hi, flags);
    MMT_HIP(hipGetLastError());
}
__global__ void k_shift_anchor(int64_t* __restrict__ off, uint32_t n, uint32_t n_docs, int64_t delta) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) off[(uint64_t)r * n_docs] += delta;
}
void shift_anchor(int64_t* off, uint32_t n, uint32_t n_docs, int64_t delta, hipStream_t s) {
    if (!n || !delta) return;
    hipLaunchKernelGGL(k_shift_anchor, dim3((n + 255) / 256), dim3(256), 0, s, off, n, n_docs, delta);
    MMT_HIP(hipGetLastError());
}

__global__ void k_rank_keys(const int64_t* __restrict__ off, uint32_t n, uint32_t n_docs,
                            const uint32_t* __restrict__ isa, uint64_t anchor_len, uint32_t* __restrict__ keys,
                            uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t o = off[(uint64_t)r * n_docs];
    vals[r] = r;
    if (o < 0 || (uint64_t)o >= anchor_len) { atomicOr(bad, 1u); keys[r] = 0; return; }
    keys[r] = isa[o];
}
// the same with 40-bit suffix ranks (wide runs)
__global__ void k_rank_keys64(const int64_t* __restrict__ off, uint32_t n, uint32_t n_docs,
                              const uint64_t* __restrict__ isa, uint64_t anchor_len, uint64_t* __restrict__ keys,
                              uint32_t* __restrict__ vals, uint32_t* __restrict__ bad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t o = off[(uint64_t)r * n_docs];
    vals[r] = r;
    if (o < 0 || (uint64_t)o >= anchor_len) { atomicOr(bad, 1u); keys[r] = 0; return; }
    keys[r] = isa[o];
}
void rank_keys64(const int64_t* off, uint32_t n, uint32_t n_docs, const uint64_t* isa, uint64_t anchor_len,
                 uint64_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_rank_keys64, dim3(grid_for(n, 256)), dim3(256), 0, s, off, n, n_docs, isa, anchor_len, keys,
                       vals, bad);
    MMT_HIP(hipGetLastError());
}
void rank_keys(const int64_t* off, uint32_t n, uint32_t n_docs, const uint32_t* isa, uint64_t anchor_len,
               uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_rank_keys, dim3(grid_for(n, 256)), dim3(256), 0, s, off, n, n_docs, isa, anchor_len, keys,
                       vals, bad);
    MMT_HIP(hipGetLastError());
}

// ---- text ------------------------------------------------------------------------------------
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
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o, 64); if (lane >= (uint32_t)o) x += y; }
    total = __shfl(x, 63, 64);
    return x - v;
}
// width of one offsets cell: optional '-' + digits of |o|
__device__ __forceinline__ uint32_t cell_width(int64_t o) {
    return o < 0 ? 1u + ndigits((uint64_t)(-o)) : ndigits((uint64_t)o);
}

__global__ void k_table_measure(const uint32_t* __restrict__ len, const int64_t* __restrict__ off, uint32_t n,
                                uint32_t n_docs, uint64_t* __restrict__ text_len) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n) return;
    uint32_t w = 0;
    for (uint32_t d = lane; d < n_docs; d += 64) w += cell_width(off[r * n_docs + d]);
    w = wave_sum(w);
    // LEN \t offs(N-1 commas) \t strands(N chars, N-1 commas) \n
    if (lane == 0) text_len[r] = (uint64_t)ndigits(len[r]) + 1 + w + (n_docs - 1) + 1 + n_docs + (n_docs - 1) + 1;
}
void table_measure(const uint32_t* len, const int64_t* off, uint32_t n, uint32_t n_docs, uint64_t* text_len,
                   hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_table_measure, dim3(grid_for((uint64_t)n * 64, 256)), dim3(256), 0, s, len, off, n, n_docs,
                       text_len);
    MMT_HIP(hipGetLastError());
}

__global__ void k_table_write(const uint32_t* __restrict__ len, const int64_t* __restrict__ off,
                              const uint8_t* __restrict__ st, uint32_t n, uint32_t n_docs,
                              const uint64_t* __restrict__ text_off, uint64_t text_base, char* __restrict__ text) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= n) return;
    char* t = text + (text_off[r] - text_base);
    const uint32_t nl = ndigits(len[r]);
    if (lane == 0) { put_uint(t, len[r], nl); t[nl] = '\t'; }
    uint32_t cur = nl + 1;
    for (uint32_t base = 0; base < n_docs; base += 64) {
        const uint32_t d = base + lane;
        int64_t o = 0; uint32_t cw = 0, wd = 0;
        if (d < n_docs) { o = off[r * n_docs + d]; cw = cell_width(o); wd = cw + (d + 1 < n_docs ? 1 : 0); }
        uint32_t total;
        const uint32_t at = wave_excl_sum(wd, lane, total);
        if (d < n_docs) {
            char* c = t + cur + at;
            if (o < 0) { c[0] = '-'; put_uint(c + 1, (uint64_t)(-o), cw - 1); } else put_uint(c, (uint64_t)o, cw);
            if (d + 1 < n_docs) c[cw] = ',';
        }
        cur += total;
    }
    if (lane == 0) t[cur] = '\t';
    cur += 1;
    for (uint32_t d = lane; d < n_docs; d += 64) {
        t[cur + 2 * d] = st[r * n_docs + d] ? '+' : '-';
        if (d + 1 < n_docs) t[cur + 2 * d + 1] = ',';
    }
    if (lane == 0) t[cur + 2 * n_docs - 1] = '\n';
}
void table_write(const uint32_t* len, const int64_t* off, const uint8_t* st, uint32_t n, uint32_t n_docs,
                 const uint64_t* text_off, uint64_t text_base, char* text, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_table_write, dim3(grid_for((uint64_t)n * 64, 256)), dim3(256), 0, s, len, off, st, n, n_docs,
                       text_off, text_base, text);
    MMT_HIP(hipGetLastError());
}

}}  // namespace mmt::mk
