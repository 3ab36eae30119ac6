// wide.hpp -- text positions and suffix ranks beyond 32 bits.
//
// The reference keeps every text-sized quantity in 40 bits (include/common.hpp:59-60 SSABYTES / THRBYTES,
// include/parse.hpp:45 int_vector<40>, dumps include/pfp_lcp_mum.hpp:323-369).  Here a collection whose text has
// fewer than 2^32 - 4096 characters runs "narrow" (every position, rank and offset is a uint32_t, as in round 1);
// anything larger runs "wide":
//   * the suffix-array column is stored as 32 low bits + 8 high bits per entry (two arrays, 5 bytes per suffix);
//   * tables that hold text positions or stream offsets (trigger positions, phrase starts, group / entry offsets
//     of the emitter, suffix ranks of the anchor) hold uint64_t;
//   * LCP values stay uint32_t (result lengths are uint32_t in the reference: include/mumsio.hpp:18,24); an LCP
//     of 2^32 - 2^20 characters or more is recorded as 2^32 - 2^20.
// Kernels are templated on the accessor / position type, so the narrow instantiation is the round-1 code.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace mmt {

// one suffix-array column as the host code passes it around (hi == nullptr: narrow)
struct SaCol {
    uint32_t* lo = nullptr;
    uint8_t* hi = nullptr;
    bool wide() const { return hi != nullptr; }
};

struct Sa32 {
    using idx_t = uint32_t;                       // positions / ranks in registers
    static constexpr bool WIDE = false;
    uint32_t* lo;
    __host__ __device__ Sa32() : lo(nullptr) {}
    __host__ __device__ explicit Sa32(const SaCol& c) : lo(c.lo) {}
    __device__ __forceinline__ uint32_t get(uint64_t j) const { return lo[j]; }
    __device__ __forceinline__ void set(uint64_t j, uint64_t v) const { lo[j] = (uint32_t)v; }
    // four consecutive entries, j a multiple of 4
    __device__ __forceinline__ void get4(uint64_t j, uint32_t (&p)[4]) const {
        const uint4 v = *reinterpret_cast<const uint4*>(lo + j);
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
};

struct Sa40 {
    using idx_t = uint64_t;
    static constexpr bool WIDE = true;
    uint32_t* lo;
    uint8_t* hi;
    __host__ __device__ Sa40() : lo(nullptr), hi(nullptr) {}
    __host__ __device__ explicit Sa40(const SaCol& c) : lo(c.lo), hi(c.hi) {}
    __device__ __forceinline__ uint64_t get(uint64_t j) const { return (uint64_t)lo[j] | ((uint64_t)hi[j] << 32); }
    __device__ __forceinline__ void set(uint64_t j, uint64_t v) const { lo[j] = (uint32_t)v; hi[j] = (uint8_t)(v >> 32); }
    __device__ __forceinline__ void get4(uint64_t j, uint64_t (&p)[4]) const {
        const uint4 v = *reinterpret_cast<const uint4*>(lo + j);
        const uint32_t h = *reinterpret_cast<const uint32_t*>(hi + j);
        p[0] = (uint64_t)v.x | ((uint64_t)(h & 0xffu) << 32);
        p[1] = (uint64_t)v.y | ((uint64_t)((h >> 8) & 0xffu) << 32);
        p[2] = (uint64_t)v.z | ((uint64_t)((h >> 16) & 0xffu) << 32);
        p[3] = (uint64_t)v.w | ((uint64_t)(h >> 24) << 32);
    }
};

// LCP values are capped here (see above); the cap leaves room for the 8-byte steps of the comparison loops
constexpr uint32_t LCP_CAP = 0xfff00000u;

// first text length that runs wide
constexpr uint64_t NARROW_LIMIT = 0xfffff000ull;

}  // namespace mmt
