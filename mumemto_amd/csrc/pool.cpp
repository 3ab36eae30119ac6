// pool.cpp -- the device heap behind DevBuf (see pool.hpp).
#include "pool.hpp"

#include "device_utils.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace mmt { namespace pool {

namespace {

constexpr size_t ALIGN = 512;                    // every block starts on a 512-byte boundary (16-byte vector loads, LDS-DMA rows)
constexpr size_t GROW = (size_t)1 << 30;         // physical memory is mapped in chunks of 1 GiB

struct Heap {
    int device = 0;
    bool vmm = false;                            // false: plain hipMalloc / hipFree per block
    char* base = nullptr;
    size_t reserved = 0, top = 0;                // virtual range, mapped prefix [0, top)
    std::map<size_t, size_t> free_blocks;        // offset -> size, coalesced
    std::map<size_t, size_t> live_blocks;        // offset -> size
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<std::pair<size_t, size_t>> mapped;   // (offset, size) per handle
    size_t live_bytes = 0, peak_bytes = 0;
    double map_seconds = 0;
};

std::mutex g_mu;
std::vector<std::unique_ptr<Heap>> g_heaps;

bool enabled() {
    static const bool on = [] {
        const char* e = std::getenv("MUMEMTO_POOL");
        return !(e && std::string(e) == "0");
    }();
    return on;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

Heap& heap_for(int device) {
    for (auto& h : g_heaps) if (h->device == device) return *h;
    std::unique_ptr<Heap> h(new Heap());
    h->device = device;
    if (enabled()) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) {
            size_t want = ((tot + GROW - 1) / GROW) * GROW;
            void* va = nullptr;
            if (hipMemAddressReserve(&va, want, 0, nullptr, 0) == hipSuccess && va) {
                h->base = static_cast<char*>(va); h->reserved = want; h->vmm = true;
            } else (void)hipGetLastError();
        } else (void)hipGetLastError();
    }
    g_heaps.push_back(std::move(h));
    return *g_heaps.back();
}

// map `bytes` (multiple of GROW) more physical memory at the top of the heap, in chunks of exactly GROW bytes:
// hipMemSetAccess rejects ("invalid argument") some mappings when the chunks of one reservation differ in size
// (tests/micro/vmm_probe.cpp), uniform ones have never failed
bool grow(Heap& H, size_t bytes) {
    if (H.top + bytes > H.reserved) return false;
    const double t0 = now_s();
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = H.device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t done = 0;
    std::string why;
    while (done < bytes) {
        hipMemGenericAllocationHandle_t handle;
        hipError_t e = hipMemCreate(&handle, GROW, &prop, 0);
        if (e != hipSuccess) { why = std::string("hipMemCreate: ") + hipGetErrorString(e); (void)hipGetLastError(); break; }
        char* at = H.base + H.top + done;
        e = hipMemMap(at, GROW, 0, handle, 0);
        if (e != hipSuccess) {
            why = std::string("hipMemMap: ") + hipGetErrorString(e); (void)hipGetLastError(); (void)hipMemRelease(handle); break;
        }
        e = hipMemSetAccess(at, GROW, &acc, 1);
        if (e != hipSuccess) {
            why = std::string("hipMemSetAccess: ") + hipGetErrorString(e); (void)hipGetLastError();
            (void)hipMemUnmap(at, GROW); (void)hipMemRelease(handle); break;
        }
        H.handles.push_back(handle);
        H.mapped.emplace_back(H.top + done, GROW);
        done += GROW;
    }
    H.map_seconds += now_s() - t0;
    if (done) {
        // the new range joins the free list (coalesced with a free block that ends at the old top)
        size_t off = H.top, size = done;
        if (!H.free_blocks.empty()) {
            auto last = std::prev(H.free_blocks.end());
            if (last->first + last->second == H.top) { off = last->first; size += last->second; H.free_blocks.erase(last); }
        }
        H.free_blocks[off] = size;
        H.top += done;
    }
    if (DevBytes::log() || done < bytes)
        std::fprintf(stderr, "[pool] device %d: +%.2f GB mapped, heap %.2f GB (%.3f s in the driver so far)%s%s\n", H.device,
                     done / 1073741824.0, H.top / 1073741824.0, H.map_seconds, done < bytes ? "; stopped by " : "", why.c_str());
    return done == bytes;
}

void* take(Heap& H, size_t need) {
    // best fit
    auto best = H.free_blocks.end();
    for (auto it = H.free_blocks.begin(); it != H.free_blocks.end(); ++it)
        if (it->second >= need && (best == H.free_blocks.end() || it->second < best->second)) best = it;
    if (best == H.free_blocks.end()) return nullptr;
    const size_t off = best->first, size = best->second;
    H.free_blocks.erase(best);
    if (size > need) H.free_blocks[off + need] = size - need;
    H.live_blocks[off] = need;
    H.live_bytes += need;
    if (H.live_bytes > H.peak_bytes) H.peak_bytes = H.live_bytes;
    return H.base + off;
}

}  // namespace

void* alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    int device = 0;
    MMT_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_mu);
    Heap& H = heap_for(device);
    if (!H.vmm) {
        void* p = nullptr;
        MMT_HIP(hipMalloc(&p, bytes));
        return p;
    }
    const size_t need = (bytes + ALIGN - 1) / ALIGN * ALIGN;
    if (void* p = take(H, need)) return p;
    // not enough contiguous free space: map more at the top (a free block that ends at the top counts)
    size_t have = 0;
    if (!H.free_blocks.empty()) {
        auto last = std::prev(H.free_blocks.end());
        if (last->first + last->second == H.top) have = last->second;
    }
    const size_t extra = ((need - have) + GROW - 1) / GROW * GROW;
    if (!grow(H, extra))
        throw HipError("out of device memory: " + std::to_string(bytes >> 20) + " MiB requested, heap of " +
                       std::to_string(H.top >> 20) + " MiB with " + std::to_string(H.live_bytes >> 20) + " MiB live");
    void* p = take(H, need);
    if (!p) throw HipError("device heap: internal error after growing");
    return p;
}

void release(void* p) {
    if (!p) return;
    // like hipFree: work that may still read or write the block (on any stream) finishes first, so that the next owner
    // of these bytes -- possibly on another stream -- cannot race with it
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& hp : g_heaps) {
        Heap& H = *hp;
        if (!H.vmm) continue;
        const char* c = static_cast<const char*>(p);
        if (c < H.base || c >= H.base + H.reserved) continue;
        const size_t off = (size_t)(c - H.base);
        auto it = H.live_blocks.find(off);
        if (it == H.live_blocks.end()) return;             // not ours (double free): ignore
        size_t size = it->second;
        H.live_blocks.erase(it);
        H.live_bytes -= size;
        size_t o = off;
        auto next = H.free_blocks.lower_bound(off);
        if (next != H.free_blocks.end() && off + size == next->first) { size += next->second; next = H.free_blocks.erase(next); }
        if (next != H.free_blocks.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == off) { o = prev->first; size += prev->second; H.free_blocks.erase(prev); }
        }
        H.free_blocks[o] = size;
        return;
    }
    (void)hipFree(p);                                       // a block of the plain path
}

Stats stats(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    Stats s{};
    for (auto& hp : g_heaps)
        if (hp->device == device) {
            s.pooled = hp->vmm; s.mapped = hp->top; s.live = hp->live_bytes; s.peak = hp->peak_bytes;
            s.map_seconds = hp->map_seconds;
            for (auto& f : hp->free_blocks) s.largest_free = std::max(s.largest_free, f.second);
        }
    return s;
}

// bytes an allocation could still get: what the driver reports as free plus what the heap holds unused
size_t available(int device) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
    const Stats s = stats(device);
    return fr + (s.mapped - s.live);
}

void trim() {
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& hp : g_heaps) {
        Heap& H = *hp;
        if (!H.vmm || !H.live_blocks.empty()) continue;
        for (size_t i = 0; i < H.handles.size(); i++) {
            (void)hipMemUnmap(H.base + H.mapped[i].first, H.mapped[i].second);
            (void)hipMemRelease(H.handles[i]);
        }
        H.handles.clear(); H.mapped.clear(); H.free_blocks.clear(); H.top = 0;
    }
}

}}  // namespace mmt::pool
