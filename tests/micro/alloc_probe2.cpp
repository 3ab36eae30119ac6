// GPU box helper: does hipMalloc scale over host threads, does it overlap with kernels, what does the
// virtual-memory API cost?  Build: hipcc --offload-arch=gfx950 -O2 -pthread -o alloc_probe2 alloc_probe2.cpp
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("ERR %s: %s\n", #x, hipGetErrorString(e)); std::fflush(stdout); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    const size_t GB = 1ull << 30;
    for (int T : {1, 2, 4, 8}) {
        std::vector<void*> ps(T, nullptr);
        std::vector<std::thread> th;
        double t = now();
        for (int i = 0; i < T; i++) th.emplace_back([&, i] { (void)hipSetDevice(0); (void)hipMalloc(&ps[i], (size_t)(10 + i) * GB / 2); });
        for (auto& x : th) x.join();
        double dt = now() - t;
        size_t tot = 0; for (int i = 0; i < T; i++) tot += (size_t)(10 + i) * GB / 2;
        std::printf("%d threads, %.1f GB total: %.3f s (%.1f ms/GB)\n", T, tot / 1.0737e9, dt, dt * 1e3 / (tot / 1.0737e9));
        std::fflush(stdout);
        // do not free: keep sizes distinct so nothing is served from a cache
    }
    // overlap with a kernel
    {
        uint32_t* buf; CK(hipMalloc(&buf, 3 * GB));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        std::atomic<bool> stop{false};
        std::atomic<int> launches{0};
        double t0 = now();
        std::thread kt([&] {
            (void)hipSetDevice(0);
            while (!stop) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, buf, 3 * GB / 4, 7u); (void)hipStreamSynchronize(st); launches++; }
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        int l0 = launches; double ta = now();
        void* p; CK(hipMalloc(&p, 21 * GB));
        double tm = now() - ta; int l1 = launches;
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        int l2 = launches; double tb = now();
        stop = true; kt.join();
        std::printf("kernel loop: %.1f launches/s before, %.1f during a 21 GB hipMalloc (%.3f s), %.1f after\n",
                    l0 / (ta - t0), (l1 - l0) / tm, tm, (l2 - l1) / (tb - ta - tm));
        std::fflush(stdout);
    }
    // virtual memory API
    {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        std::printf("VMM granularity %zu\n", gran);
        const size_t chunk = 2 * GB, total = 32 * GB;
        void* va = nullptr;
        double t = now();
        CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
        std::printf("reserve 32 GB %.4f s\n", now() - t);
        hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        double tc = 0, tm = 0, ta = 0;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            t = now(); CK(hipMemCreate(&h, chunk, &prop, 0)); tc += now() - t;
            t = now(); CK(hipMemMap((char*)va + off, chunk, 0, h, 0)); tm += now() - t;
            t = now(); CK(hipMemSetAccess((char*)va + off, chunk, &acc, 1)); ta += now() - t;
            hs.push_back(h);
        }
        std::printf("VMM 32 GB in 2 GB chunks: create %.3f s, map %.3f s, setaccess %.3f s (%.1f ms/GB total)\n", tc, tm, ta,
                    (tc + tm + ta) * 1e3 / 32);
        t = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)va, total / 4, 3u); CK(hipDeviceSynchronize());
        double dt = now() - t; std::printf("fill over the mapped range: %.3f s (%.0f GB/s)\n", dt, 34.4 / dt);
        t = now();
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)va, total / 4, 4u); CK(hipDeviceSynchronize());
        dt = now() - t; std::printf("again: %.3f s (%.0f GB/s)\n", dt, 34.4 / dt);
        // threads creating chunks in parallel
        for (int T : {4}) {
            std::vector<std::thread> th; std::vector<hipMemGenericAllocationHandle_t> h2(T * 4);
            t = now();
            for (int i = 0; i < T; i++) th.emplace_back([&, i] { (void)hipSetDevice(0); for (int q = 0; q < 4; q++) (void)hipMemCreate(&h2[i * 4 + q], chunk, &prop, 0); });
            for (auto& x : th) x.join();
            dt = now() - t; std::printf("hipMemCreate %d threads x 4 x 2 GB: %.3f s (%.1f ms/GB)\n", T, dt, dt * 1e3 / (T * 8));
        }
    }
    return 0;
}
