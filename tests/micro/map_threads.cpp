// GPU box helper: what does mapping the device heap cost, and does it go faster from several threads?
// Dirty the memory first (allocate, write, free: what a box looks like after another job), then map 96 GiB in 1 GiB chunks
// with 1, 2, 4, 8 threads; per phase: hipMemCreate, hipMemMap, hipMemSetAccess.
//   hipcc -O2 --offload-arch=gfx950 tests/micro/map_threads.cpp -o /tmp/map_threads -lpthread && /tmp/map_threads
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define IGN(x) do { hipError_t _e = (x); (void)_e; } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    IGN(hipSetDevice(0)); IGN(hipFree(nullptr));
    const size_t GB = 1ull << 30, total = 96 * GB;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int dirty = 0; dirty < 2; dirty++)
    for (int T : {1, 2, 4, 8}) {
        if (dirty) {                       // what another process leaves behind
            void* p = nullptr;
            if (hipMalloc(&p, total + 8 * GB) == hipSuccess) { IGN(hipMemset(p, 5, total + 8 * GB)); IGN(hipDeviceSynchronize()); IGN(hipFree(p)); }
            else IGN(hipGetLastError());
        }
        void* va = nullptr;
        if (hipMemAddressReserve(&va, 300 * GB, 0, nullptr, 0) != hipSuccess) { std::printf("reserve failed\n"); return 1; }
        const size_t n = total / GB;
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        std::atomic<size_t> next{0};
        std::vector<double> tc(T, 0), tm(T, 0), ta(T, 0);
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t] {
            IGN(hipSetDevice(0));
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                double a = now();
                if (hipMemCreate(&hs[i], GB, &prop, 0) != hipSuccess) { std::printf("create failed\n"); return; }
                double b = now(); tc[t] += b - a;
                if (hipMemMap((char*)va + i * GB, GB, 0, hs[i], 0) != hipSuccess) { std::printf("map failed\n"); return; }
                double c = now(); tm[t] += c - b;
                if (hipMemSetAccess((char*)va + i * GB, GB, &acc, 1) != hipSuccess) { std::printf("access failed\n"); return; }
                ta[t] += now() - c;
            }
        });
        for (auto& x : th) x.join();
        const double wall = now() - t0;
        double c = 0, m = 0, a = 0;
        for (int t = 0; t < T; t++) { c += tc[t]; m += tm[t]; a += ta[t]; }
        const double t1 = now();
        IGN(hipMemset(va, 1, total)); IGN(hipDeviceSynchronize());
        const double touch = now() - t1;
        std::printf("%s box, %d thread(s): 96 GiB mapped in %.3f s wall (thread-seconds: create %.3f, map %.3f, access %.3f); first memset over it %.3f s\n",
                    dirty ? "dirtied" : "fresh", T, wall, c, m, a, touch);
        for (size_t i = 0; i < n; i++) { IGN(hipMemUnmap((char*)va + i * GB, GB)); IGN(hipMemRelease(hs[i])); }
        IGN(hipMemAddressFree(va, 300 * GB));
    }
    return 0;
}
