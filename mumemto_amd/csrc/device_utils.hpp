// device_utils.hpp -- HIP error handling, owned device buffers, event timers.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pool.hpp"

namespace mmt {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// the device has no room for an allocation (the device heap, or hipMalloc when the heap is switched off): the one error
// callers react to -- a run that does not fit as one array is repeated as anchor partitions
struct DeviceOom : HipError {
    using HipError::HipError;
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        throw HipError(std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " at " + file + ":" +
                       std::to_string(line));
    }
}
#define MMT_HIP(x) ::mmt::hip_check((x), #x, __FILE__, __LINE__)

// bytes of HBM this process holds through DevBuf (MUMEMTO_TIMING prints the high-water mark)
struct DevBytes {
    static size_t& live() { static size_t v = 0; return v; }
    static size_t& peak() { static size_t v = 0; return v; }
    static double& seconds() { static double v = 0; return v; }
    static bool log() { static const bool on = std::getenv("MUMEMTO_ALLOC_LOG") != nullptr; return on; }
};

// Device allocation that only grows (steps of the hot path are re-run by the
// bench with the same sizes), taken from the per-device heap of pool.hpp.
template <typename T>
class DevBuf {
public:
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void ensure(size_t n) {
        if (n <= cap_) { n_ = n; return; }
        release();                                   // (a borrowed view that is too small is dropped: own memory then)
        // a little slack, so that a re-run with slightly different sizes does not re-allocate; capped: a sixteenth of a
        // 48 GB column is 3 GB of device memory nobody uses
        size_t want = n + std::min<size_t>(n / 16, ((size_t)16 << 20) / sizeof(T)) + 64;
        const auto t0 = std::chrono::steady_clock::now();
        p_ = static_cast<T*>(pool::alloc(want * sizeof(T)));
        DevBytes::seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        cap_ = want; n_ = n;
        DevBytes::live() += want * sizeof(T);
        if (DevBytes::log() && want * sizeof(T) >= (64u << 20))
            std::fprintf(stderr, "[alloc] %8.2f GB  (%zu x %zu B)  live %.2f GB\n", want * sizeof(T) / 1073741824.0, want,
                         sizeof(T), DevBytes::live() / 1073741824.0);
        if (DevBytes::live() > DevBytes::peak()) DevBytes::peak() = DevBytes::live();
    }
    void release() {
        if (p_ && owned_) { pool::release(p_); DevBytes::live() -= cap_ * sizeof(T); }
        p_ = nullptr; owned_ = true;
        cap_ = n_ = 0;
    }
    // A view of memory that belongs to another buffer (stage scratch living inside columns that are written later):
    // not counted, not freed.
    void borrow(T* p, size_t n) { release(); p_ = p; cap_ = n_ = n; owned_ = false; }
    bool owned() const { return owned_; }
    T* get() const { return p_; }
    size_t size() const { return n_; }
    size_t bytes() const { return n_ * sizeof(T); }
    void swap(DevBuf& o) { std::swap(p_, o.p_); std::swap(cap_, o.cap_); std::swap(n_, o.n_); std::swap(owned_, o.owned_); }

private:
    T* p_ = nullptr;
    size_t cap_ = 0, n_ = 0;
    bool owned_ = true;
};

// A table of text positions / stream offsets: uint32_t entries in a narrow run, uint64_t entries in a wide one
// (wide.hpp).  The launch wrappers take the untyped pointer together with the `wide` flag.
class PosBuf {
public:
    void ensure(size_t n, bool wide) { wide_ = wide; n_ = n; b_.ensure(n * (wide ? 8 : 4)); }
    void release() { b_.release(); n_ = 0; }
    // a view of memory that belongs to another buffer (DevBuf::borrow): n entries of the given width
    void borrow(void* p, size_t n, bool wide) { wide_ = wide; n_ = n; b_.borrow(static_cast<uint8_t*>(p), n * (wide ? 8 : 4)); }
    bool wide() const { return wide_; }
    void* get() const { return b_.get(); }
    uint32_t* p32() const { return reinterpret_cast<uint32_t*>(b_.get()); }
    uint64_t* p64() const { return reinterpret_cast<uint64_t*>(b_.get()); }
    // entry i as an untyped pointer (for partial copies)
    void* at(size_t i) const { return b_.get() + i * (wide_ ? 8 : 4); }
    size_t size() const { return n_; }
    size_t elem() const { return wide_ ? 8 : 4; }
    void swap(PosBuf& o) { b_.swap(o.b_); std::swap(n_, o.n_); std::swap(wide_, o.wide_); }
    // one entry, read back (synchronises the stream)
    uint64_t read(size_t i, hipStream_t s) const {
        uint64_t v = 0;
        MMT_HIP(hipMemcpyAsync(&v, at(i), elem(), hipMemcpyDeviceToHost, s));
        MMT_HIP(hipStreamSynchronize(s));
        return v;                                  // little endian: a 4-byte entry lands in the low half
    }
    void write(size_t i, uint64_t v, hipStream_t s) const {
        MMT_HIP(hipMemcpyAsync(at(i), &v, elem(), hipMemcpyHostToDevice, s));
        MMT_HIP(hipStreamSynchronize(s));
    }

private:
    DevBuf<uint8_t> b_;
    size_t n_ = 0;
    bool wide_ = false;
};

// Page-locked host memory that only grows: D2H targets of the result rows / output bytes.
template <typename T>
class PinnedBuf {
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p_) (void)hipHostFree(p_); }
    void ensure(size_t n) {
        if (n <= cap_) return;
        if (p_) { (void)hipHostFree(p_); p_ = nullptr; }
        size_t want = n + n / 8 + 64;
        MMT_HIP(hipHostMalloc(reinterpret_cast<void**>(&p_), want * sizeof(T), hipHostMallocDefault));
        cap_ = want;
    }
    T* get() const { return p_; }

private:
    T* p_ = nullptr;
    size_t cap_ = 0;
};

class EventPair {
public:
    EventPair() { MMT_HIP(hipEventCreate(&a_)); MMT_HIP(hipEventCreate(&b_)); }
    ~EventPair() { (void)hipEventDestroy(a_); (void)hipEventDestroy(b_); }
    EventPair(const EventPair&) = delete;
    void start(hipStream_t s) { MMT_HIP(hipEventRecord(a_, s)); used_ = true; }
    void stop(hipStream_t s) { MMT_HIP(hipEventRecord(b_, s)); }
    float ms() {
        if (!used_) return 0.f;
        float t = 0.f;
        MMT_HIP(hipEventSynchronize(b_));
        MMT_HIP(hipEventElapsedTime(&t, a_, b_));
        return t;
    }
    void reset() { used_ = false; }

private:
    hipEvent_t a_{}, b_{};
    bool used_ = false;
};

template <typename T>
inline void d2h(std::vector<T>& dst, const T* src, size_t n, hipStream_t s) {
    dst.resize(n);
    if (n) MMT_HIP(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
}

}  // namespace mmt
