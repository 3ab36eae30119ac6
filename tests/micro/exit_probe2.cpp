// GPU box probe, second form: what a process pays between _Exit and its parent's waitpid, by the KIND of memory it holds.
// usage: exit_probe2 <mode> <GB>     modes: none | malloc | vmm (reserve + 1 GiB chunks created, mapped, touched: pool.cpp's heap)
//                                          | host (anonymous huge-page memory, filled and uploaded from) | vmm_unmap (vmm, unmapped and released before the exit)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "none";
    const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 0;
    const double t0 = now();
    if (hipSetDevice(0) != hipSuccess) return 2;
    hipStream_t s; hipStreamCreate(&s);
    std::vector<hipMemGenericAllocationHandle_t> handles;
    void* base = nullptr;
    if (mode == "malloc" && gb) {
        if (hipMalloc(&base, gb << 30) != hipSuccess) return 3;
        hipMemsetAsync(base, 1, gb << 30, s);
    } else if ((mode == "vmm" || mode == "vmm_unmap") && gb) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        if (hipMemAddressReserve(&base, (size_t)512 << 30, (size_t)1 << 30, nullptr, 0) != hipSuccess) return 4;
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t i = 0; i < gb; i++) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, (size_t)1 << 30, &prop, 0) != hipSuccess) return 5;
            if (hipMemMap((char*)base + (i << 30), (size_t)1 << 30, 0, h, 0) != hipSuccess) return 6;
            if (hipMemSetAccess((char*)base + (i << 30), (size_t)1 << 30, &acc, 1) != hipSuccess) return 7;
            handles.push_back(h);
        }
        hipMemsetAsync(base, 1, gb << 30, s);
    } else if (mode.rfind("host", 0) == 0 && gb) {
        // host: filled + uploaded from; hostonly: filled, never seen by HIP; host_small: filled with 4 KiB pages; host_unmap /
        // host_dontneed: given back before the exit (timed); host_pieces: uploaded in 64 MiB pieces
        void* d = nullptr;
        if (hipMalloc(&d, gb << 30) != hipSuccess) return 3;
        void* m = mmap(nullptr, gb << 30, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(m, gb << 30, mode == "host_small" ? MADV_NOHUGEPAGE : MADV_HUGEPAGE);
        memset(m, 7, gb << 30);
        if (mode == "host_pieces") {
            for (size_t o = 0; o < (gb << 30); o += (size_t)64 << 20) hipMemcpyAsync((char*)d + o, (char*)m + o, (size_t)64 << 20, hipMemcpyHostToDevice, s);
        } else if (mode != "hostonly") hipMemcpyAsync(d, m, gb << 30, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        const double t1 = now();
        if (mode == "host_unmap") munmap(m, gb << 30);
        if (mode == "host_dontneed") madvise(m, gb << 30, MADV_DONTNEED);
        if (mode == "host_unmap" || mode == "host_dontneed") fprintf(stderr, "giving it back: %.3f s; ", now() - t1);
    }
    hipStreamSynchronize(s);
    if (mode == "vmm_unmap") {
        for (size_t i = 0; i < handles.size(); i++) { hipMemUnmap((char*)base + (i << 30), (size_t)1 << 30); hipMemRelease(handles[i]); }
        hipMemAddressFree(base, (size_t)512 << 30);
    }
    fprintf(stderr, "%.3f\n", now() - t0);
    _Exit(0);
}
