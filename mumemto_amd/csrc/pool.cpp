// pool.cpp -- the device heap behind DevBuf (see pool.hpp).
#include "pool.hpp"

#include "device_utils.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mmt { namespace pool {

namespace {

constexpr size_t ALIGN = 512;                    // every block starts on a 512-byte boundary (16-byte vector loads, LDS-DMA rows)
constexpr size_t GROW = (size_t)1 << 30;         // physical memory is mapped in chunks of 1 GiB

// Two regions in one reservation: blocks of SMALL_LIMIT bytes or more grow upwards from offset 0, smaller ones live in
// a region that grows downwards from the end of the reservation -- a few long-lived kilobyte buffers (counters, document
// tables, rocPRIM scratch) in the middle of 48 GB columns would otherwise split every large hole and force the heap to
// map more physical memory than the run ever holds at once.
constexpr size_t SMALL_LIMIT = (size_t)32 << 20;

struct Heap {
    int device = 0;
    bool vmm = false;                            // false: plain hipMalloc / hipFree per block
    char* base = nullptr;
    size_t reserved = 0, top = 0;                // virtual range, mapped prefix [0, top)
    size_t small_bottom = 0;                     // mapped suffix [small_bottom, reserved) for the small blocks
    std::map<size_t, size_t> free_blocks;        // offset -> size, coalesced (large region)
    std::map<size_t, size_t> small_free;         // offset -> size, coalesced (small region)
    std::map<size_t, size_t> live_blocks;        // offset -> size
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<std::pair<size_t, size_t>> mapped;   // (offset, size) per handle
    size_t live_bytes = 0, peak_bytes = 0;
    double map_seconds = 0;
};

std::mutex g_mu;
std::vector<std::unique_ptr<Heap>> g_heaps;
// chunks on their way out (shrink_async): g_unmap_pending is set under g_mu BEFORE the heap's top is lowered and cleared by the
// helper thread when the last chunk is unmapped; grow() -- called with g_mu held -- waits for it (the helper never takes g_mu)
std::atomic<bool> g_unmap_pending{false};
std::mutex g_unmap_thread_mu;                       // guards g_unmap_thread itself
std::thread* g_unmap_thread = nullptr;

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
                h->base = static_cast<char*>(va); h->reserved = want; h->small_bottom = want; h->vmm = true;
            } else (void)hipGetLastError();
        } else (void)hipGetLastError();
    }
    g_heaps.push_back(std::move(h));
    return *g_heaps.back();
}

// maps one chunk of exactly GROW bytes at `offset`: hipMemSetAccess rejects ("invalid argument") some mappings when the
// chunks of one reservation differ in size (tests/micro/vmm_probe.cpp), uniform ones have never failed
bool map_chunk(Heap& H, size_t offset, std::string& why) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = H.device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t handle;
    hipError_t e = hipMemCreate(&handle, GROW, &prop, 0);
    if (e != hipSuccess) { why = std::string("hipMemCreate: ") + hipGetErrorString(e); (void)hipGetLastError(); return false; }
    char* at = H.base + offset;
    e = hipMemMap(at, GROW, 0, handle, 0);
    if (e != hipSuccess) {
        why = std::string("hipMemMap: ") + hipGetErrorString(e); (void)hipGetLastError(); (void)hipMemRelease(handle); return false;
    }
    e = hipMemSetAccess(at, GROW, &acc, 1);
    if (e != hipSuccess) {
        why = std::string("hipMemSetAccess: ") + hipGetErrorString(e); (void)hipGetLastError();
        (void)hipMemUnmap(at, GROW); (void)hipMemRelease(handle); return false;
    }
    H.handles.push_back(handle);
    H.mapped.emplace_back(offset, GROW);
    return true;
}

void add_free(std::map<size_t, size_t>& fl, size_t off, size_t size) {
    auto next = fl.lower_bound(off);
    if (next != fl.end() && off + size == next->first) { size += next->second; next = fl.erase(next); }
    if (next != fl.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) { off = prev->first; size += prev->second; fl.erase(prev); }
    }
    fl[off] = size;
}

size_t heap_limit() {
    static const size_t limit = [] { const char* e = std::getenv("MUMEMTO_HEAP_LIMIT"); return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0; }();
    return limit;
}

// MUMEMTO_HEAP_RESERVE (bytes, default 0): device memory the heap leaves to the driver and the runtime -- the estimates
// (pool::available) know about it, unlike MUMEMTO_HEAP_LIMIT, which exists to make an accepted run fail
std::atomic<size_t> g_reserve_set{~(size_t)0};        // pool::set_reserve (tests): overrides the environment while it is not ~0
size_t driver_reserve() {
    static const size_t keep = [] { const char* e = std::getenv("MUMEMTO_HEAP_RESERVE"); return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0; }();
    const size_t set = g_reserve_set.load();
    return set != ~(size_t)0 ? set : keep;
}

// map `bytes` (multiple of GROW) more physical memory at the top of the large region
bool grow(Heap& H, size_t bytes) {
    if (H.top + bytes > H.small_bottom) return false;
    // MUMEMTO_HEAP_LIMIT (bytes): the large region stops growing there -- how the tests make a run that the estimate
    // accepted run out of device memory
    const size_t limit = heap_limit();
    if (limit && H.top + bytes > limit) return false;
    if (const size_t keep = driver_reserve()) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
        if (fr < bytes + keep) return false;
    }
    while (g_unmap_pending.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (chunks on their way out may sit where this maps)
    const double t0 = now_s();
    size_t done = 0;
    std::string why;
    while (done < bytes && map_chunk(H, H.top + done, why)) done += GROW;
    H.map_seconds += now_s() - t0;
    if (done) { add_free(H.free_blocks, H.top, done); H.top += done; }
    if (DevBytes::log() || done < bytes)
        std::fprintf(stderr, "[pool] device %d: +%.2f GB mapped, heap %.2f GB (%.3f s in the driver so far)%s%s\n", H.device,
                     done / 1073741824.0, H.top / 1073741824.0, H.map_seconds, done < bytes ? "; stopped by " : "", why.c_str());
    return done == bytes;
}
// one more chunk below the small region
bool grow_small(Heap& H) {
    if (H.small_bottom < H.top + GROW) return false;
    const double t0 = now_s();
    std::string why;
    const bool ok = map_chunk(H, H.small_bottom - GROW, why);
    H.map_seconds += now_s() - t0;
    if (ok) { H.small_bottom -= GROW; add_free(H.small_free, H.small_bottom, GROW); }
    else std::fprintf(stderr, "[pool] device %d: small region stopped by %s\n", H.device, why.c_str());
    return ok;
}

void* take_from(Heap& H, std::map<size_t, size_t>& fl, size_t need) {
    // best fit
    auto best = fl.end();
    for (auto it = fl.begin(); it != fl.end(); ++it)
        if (it->second >= need && (best == fl.end() || it->second < best->second)) best = it;
    if (best == fl.end()) return nullptr;
    const size_t off = best->first, size = best->second;
    fl.erase(best);
    if (size > need) fl[off + need] = size - need;
    H.live_blocks[off] = need;
    H.live_bytes += need;
    if (H.live_bytes > H.peak_bytes) H.peak_bytes = H.live_bytes;
    return H.base + off;
}

void* take(Heap& H, size_t need) { return take_from(H, H.free_blocks, need); }

}  // namespace

void* alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    int device = 0;
    MMT_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_mu);
    Heap& H = heap_for(device);
    if (!H.vmm) {
        void* p = nullptr;
        const hipError_t me = hipMalloc(&p, bytes);
        if (me == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            throw DeviceOom("out of device memory: hipMalloc of " + std::to_string(bytes >> 20) + " MiB failed");
        }
        MMT_HIP(me);
        return p;
    }
    const size_t need = (bytes + ALIGN - 1) / ALIGN * ALIGN;
    if (need < SMALL_LIMIT) {
        if (void* p = take_from(H, H.small_free, need)) return p;
        if (grow_small(H))
            if (void* p = take_from(H, H.small_free, need)) return p;
        // (no room left for the small region: fall through to the large one)
    }
    if (void* p = take(H, need)) return p;
    // not enough contiguous free space: map more at the top (a free block that ends at the top counts)
    size_t have = 0;
    if (!H.free_blocks.empty()) {
        auto last = std::prev(H.free_blocks.end());
        if (last->first + last->second == H.top) have = last->second;
    }
    const size_t extra = ((need - have) + GROW - 1) / GROW * GROW;
    if (!grow(H, extra))
        throw DeviceOom("out of device memory: " + std::to_string(bytes >> 20) + " MiB requested, heap of " +
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
        add_free(off >= H.small_bottom ? H.small_free : H.free_blocks, off, size);
        return;
    }
    (void)hipFree(p);                                       // a block of the plain path
}

void set_reserve(size_t bytes) { g_reserve_set.store(bytes); }

void premap(int device, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
    std::lock_guard<std::mutex> lock(g_mu);
    Heap& H = heap_for(device);
    if (!H.vmm) return;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return; }
    const size_t spare = (size_t)8 << 30;
    const size_t want = std::min(bytes, H.top + (fr > spare ? fr - spare : 0));
    if (want <= H.top) return;
    (void)grow(H, (want - H.top + GROW - 1) / GROW * GROW);       // (as far as it gets: the rest grows on demand)
}

Stats stats(int device) {
    std::lock_guard<std::mutex> lock(g_mu);
    Stats s{};
    for (auto& hp : g_heaps)
        if (hp->device == device) {
            s.pooled = hp->vmm; s.mapped = hp->top + (hp->reserved - hp->small_bottom); s.live = hp->live_bytes;
            s.peak = hp->peak_bytes;
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
    const size_t keep = driver_reserve();
    fr = fr > keep ? fr - keep : 0;
    return fr + (s.mapped - s.live);
}

// ---- giving the free tail of the heap back while the run goes on (one-shot processes: mumemto_exec) ----------------------
// What a process holds mapped when it exits is torn down by the driver on its way out: 0.57 s for the 116 GB the bench
// workload's dictionary sort leaves mapped -- a quarter of the job's wall clock by SURVEY.md 8(d)'s definition (process start ->
// exit).  Once the peak stage is over, the chunks at the top of the heap that are wholly free are handed back by a helper
// thread while the GPU works on the stages that follow.  Not for a process that runs again: memory the driver gets back is
// scrubbed, and a heap that grows again waits for that.
void shrink_async(int device) {
    shrink_wait();
    std::vector<std::pair<char*, hipMemGenericAllocationHandle_t>> todo;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (auto& hp : g_heaps) {
            Heap& H = *hp;
            if (!H.vmm || hp->device != device || H.free_blocks.empty()) continue;
            auto last = std::prev(H.free_blocks.end());
            if (last->first + last->second != H.top) continue;
            // whole chunks inside the free tail [last->first, top): chunks sit at multiples of GROW
            const size_t first_chunk = (last->first + GROW - 1) / GROW * GROW;
            if (first_chunk >= H.top) continue;
            const size_t off = last->first, size = last->second;
            H.free_blocks.erase(last);
            if (first_chunk > off) H.free_blocks[off] = first_chunk - off;
            (void)size;
            for (size_t i = 0; i < H.mapped.size();) {
                if (H.mapped[i].first >= first_chunk && H.mapped[i].first < H.top) {
                    todo.emplace_back(H.base + H.mapped[i].first, H.handles[i]);
                    H.mapped.erase(H.mapped.begin() + (long)i); H.handles.erase(H.handles.begin() + (long)i);
                } else i++;
            }
            H.top = first_chunk;
        }
        if (!todo.empty()) g_unmap_pending.store(true, std::memory_order_release);    // (under g_mu, with the lowered top)
    }
    if (todo.empty()) return;
    // (a leaked pointer: a joinable std::thread with static storage would end the process in its destructor)
    std::lock_guard<std::mutex> tl(g_unmap_thread_mu);
    g_unmap_thread = new std::thread([todo, device]() {
        (void)hipSetDevice(device);
        // blocks are freed on the host, not in stream order: work that was enqueued before the release may still read a chunk
        (void)hipDeviceSynchronize();
        for (const auto& t : todo) { (void)hipMemUnmap(t.first, GROW); (void)hipMemRelease(t.second); }
        g_unmap_pending.store(false, std::memory_order_release);
    });
}
void shrink_wait() {
    std::lock_guard<std::mutex> tl(g_unmap_thread_mu);
    if (g_unmap_thread) { if (g_unmap_thread->joinable()) g_unmap_thread->join(); delete g_unmap_thread; g_unmap_thread = nullptr; }
}

void trim() {
    shrink_wait();
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto& hp : g_heaps) {
        Heap& H = *hp;
        if (!H.vmm || !H.live_blocks.empty()) continue;
        for (size_t i = 0; i < H.handles.size(); i++) {
            (void)hipMemUnmap(H.base + H.mapped[i].first, H.mapped[i].second);
            (void)hipMemRelease(H.handles[i]);
        }
        H.handles.clear(); H.mapped.clear(); H.free_blocks.clear(); H.small_free.clear(); H.top = 0;
        H.small_bottom = H.reserved;
    }
}

}}  // namespace mmt::pool
