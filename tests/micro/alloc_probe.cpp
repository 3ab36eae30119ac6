// GPU box helper (not a test): what do hipMalloc / hipFree / hipHostMalloc / H2D cost on this stack at the sizes a
// 12 G-character collection needs?  Build: hipcc --offload-arch=gfx950 -O2 -o alloc_probe alloc_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    double t = now();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    std::printf("runtime init %.3f s\n", now() - t);
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    std::printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
    const size_t GB = 1ull << 30;
    // 1. many allocations in sequence, cumulative
    for (int round = 0; round < 2; round++) {
        std::vector<void*> ps;
        size_t sizes[] = {1, 2, 4, 8, 16, 32, 32, 32, 32, 32, 32};
        size_t cum = 0;
        for (size_t s : sizes) {
            void* p = nullptr;
            t = now();
            hipError_t e = hipMalloc(&p, s * GB);
            double dt = now() - t;
            if (e != hipSuccess) { std::printf("  malloc %zu GB failed\n", s); break; }
            cum += s;
            std::printf("round %d: hipMalloc %2zu GB  %.3f s (%.1f ms/GB), cumulative %zu GB\n", round, s, dt, dt * 1e3 / s, cum);
            ps.push_back(p);
        }
        t = now();
        for (void* p : ps) CK(hipFree(p));
        std::printf("round %d: hipFree all  %.3f s\n", round, now() - t);
    }
    // 2. one big block
    for (size_t s : {64, 128, 200, 240}) {
        void* p = nullptr;
        t = now();
        hipError_t e = hipMalloc(&p, s * GB);
        double dt = now() - t;
        if (e != hipSuccess) { std::printf("one block %zu GB failed: %s\n", s, hipGetErrorString(e)); continue; }
        std::printf("one block %3zu GB  %.3f s (%.1f ms/GB)", s, dt, dt * 1e3 / s);
        t = now();
        CK(hipMemset(p, 1, s * GB)); CK(hipDeviceSynchronize());
        dt = now() - t;
        std::printf(", first memset %.3f s (%.0f GB/s)", dt, s * 1.0737 / dt);
        t = now();
        CK(hipMemset(p, 2, s * GB)); CK(hipDeviceSynchronize());
        dt = now() - t;
        std::printf(", second %.3f s (%.0f GB/s)", dt, s * 1.0737 / dt);
        t = now();
        CK(hipFree(p));
        std::printf(", free %.3f s\n", now() - t);
    }
    // 3. stream-ordered pool
    {
        hipStream_t st; CK(hipStreamCreate(&st));
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int round = 0; round < 2; round++) {
            std::vector<void*> ps;
            t = now();
            for (int i = 0; i < 8; i++) { void* p; CK(hipMallocAsync(&p, 16 * GB, st)); ps.push_back(p); }
            CK(hipStreamSynchronize(st));
            double dt = now() - t;
            std::printf("pool round %d: 8 x 16 GB hipMallocAsync %.3f s", round, dt);
            t = now();
            for (void* p : ps) CK(hipFreeAsync(p, st));
            CK(hipStreamSynchronize(st));
            std::printf(", hipFreeAsync %.3f s\n", now() - t);
        }
        CK(hipMemPoolTrimTo(pool, 0));
    }
    // 4. host memory and transfers
    {
        const size_t S = 6 * GB;
        void* d; CK(hipMalloc(&d, S));
        t = now();
        char* pageable = (char*)std::malloc(S);
        std::memset(pageable, 1, S);
        std::printf("malloc+touch 6 GB pageable %.3f s\n", now() - t);
        t = now(); CK(hipMemcpy(d, pageable, S, hipMemcpyHostToDevice));
        double dt = now() - t; std::printf("H2D pageable 6 GB %.3f s (%.1f GB/s)\n", dt, 6.44 / dt);
        t = now(); CK(hipMemcpy(d, pageable, S, hipMemcpyHostToDevice));
        dt = now() - t; std::printf("H2D pageable again %.3f s (%.1f GB/s)\n", dt, 6.44 / dt);
        t = now(); CK(hipHostRegister(pageable, S, hipHostRegisterDefault));
        dt = now() - t; std::printf("hipHostRegister 6 GB %.3f s\n", dt);
        t = now(); CK(hipMemcpy(d, pageable, S, hipMemcpyHostToDevice));
        dt = now() - t; std::printf("H2D registered %.3f s (%.1f GB/s)\n", dt, 6.44 / dt);
        CK(hipHostUnregister(pageable));
        void* pinned;
        t = now(); CK(hipHostMalloc(&pinned, S, hipHostMallocDefault));
        dt = now() - t; std::printf("hipHostMalloc 6 GB %.3f s\n", dt);
        t = now(); std::memset(pinned, 2, S); dt = now() - t; std::printf("memset pinned %.3f s (%.1f GB/s)\n", dt, 6.44 / dt);
        t = now(); CK(hipMemcpy(d, pinned, S, hipMemcpyHostToDevice));
        dt = now() - t; std::printf("H2D pinned %.3f s (%.1f GB/s)\n", dt, 6.44 / dt);
        t = now(); CK(hipMemcpy(pinned, d, 1 * GB, hipMemcpyDeviceToHost));
        dt = now() - t; std::printf("D2H pinned 1 GB %.3f s (%.1f GB/s)\n", dt, 1.0737 / dt);
        t = now(); CK(hipMemcpy(pageable, d, 1 * GB, hipMemcpyDeviceToHost));
        dt = now() - t; std::printf("D2H pageable 1 GB %.3f s (%.1f GB/s)\n", dt, 1.0737 / dt);
        // file write
        t = now();
        FILE* f = std::fopen("/tmp/probe.bin", "wb"); std::fwrite(pinned, 1, 1 * GB, f); std::fclose(f);
        dt = now() - t; std::printf("fwrite 1 GB to /tmp %.3f s (%.1f GB/s)\n", dt, 1.0737 / dt);
        t = now();
        f = std::fopen("/dev/shm/probe.bin", "wb"); if (f) { std::fwrite(pinned, 1, 1 * GB, f); std::fclose(f); }
        dt = now() - t; std::printf("fwrite 1 GB to /dev/shm %.3f s (%.1f GB/s)\n", dt, 1.0737 / dt);
        std::remove("/tmp/probe.bin"); std::remove("/dev/shm/probe.bin");
        CK(hipHostFree(pinned)); std::free(pageable); CK(hipFree(d));
    }
    std::printf("cores: "); std::fflush(stdout); return std::system("nproc; free -g | head -2; df -h /tmp /dev/shm | tail -2");
}
