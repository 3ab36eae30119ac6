// GPU box helper: which chunk size / alignment combinations does hipMemSetAccess accept on this stack?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define IGN(x) do { hipError_t _e = (x); (void)_e; } while (0)
int main() {
    IGN(hipSetDevice(0)); IGN(hipFree(nullptr));
    const size_t M = 1ull << 20, GB = 1ull << 30;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t align : {(size_t)0, GB, 2 * GB}) {
        for (size_t chunk : {256 * M, GB, 2 * GB}) {
            const size_t total = 24 * GB;
            void* va = nullptr;
            hipError_t e = hipMemAddressReserve(&va, 300 * GB, align, nullptr, 0);
            if (e != hipSuccess) { std::printf("reserve align %zu: %s\n", align, hipGetErrorString(e)); IGN(hipGetLastError()); continue; }
            std::vector<hipMemGenericAllocationHandle_t> hs;
            size_t top = 0; int fails = 0; size_t first_fail = 0;
            while (top < total) {
                hipMemGenericAllocationHandle_t h;
                e = hipMemCreate(&h, chunk, &prop, 0);
                if (e != hipSuccess) { std::printf("create failed: %s\n", hipGetErrorString(e)); IGN(hipGetLastError()); break; }
                e = hipMemMap((char*)va + top, chunk, 0, h, 0);
                if (e != hipSuccess) { std::printf("map failed at %zu: %s\n", top, hipGetErrorString(e)); IGN(hipGetLastError()); IGN(hipMemRelease(h)); break; }
                e = hipMemSetAccess((char*)va + top, chunk, &acc, 1);
                if (e != hipSuccess) { if (!fails) first_fail = top; fails++; IGN(hipGetLastError()); IGN(hipMemUnmap((char*)va + top, chunk)); IGN(hipMemRelease(h)); break; }
                hs.push_back(h); top += chunk;
            }
            hipError_t m = hipSuccess;
            if (top) { m = hipMemset(va, 3, top); if (m == hipSuccess) m = hipDeviceSynchronize(); }
            std::printf("va %p (align req %4zu MB) chunk %4zu MB: mapped %5.2f GB, first access failure at +%.2f GB (%d), memset over all: %s\n",
                        va, align / M, chunk / M, top / 1073741824.0, first_fail / 1073741824.0, fails, hipGetErrorString(m));
            for (size_t i = 0; i < hs.size(); i++) { IGN(hipMemUnmap((char*)va + i * chunk, chunk)); IGN(hipMemRelease(hs[i])); }
            IGN(hipMemAddressFree(va, 300 * GB));
        }
    }
    // non-uniform sizes at a 2 GB aligned base
    {
        void* va = nullptr;
        IGN(hipMemAddressReserve(&va, 300 * GB, 2 * GB, nullptr, 0));
        size_t sizes[] = {9 * 256 * M, 17 * 256 * M, 17 * 256 * M, 3 * 256 * M, 256 * M, 5 * GB};
        size_t top = 0;
        for (size_t s : sizes) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, s, &prop, 0);
            hipError_t e2 = e == hipSuccess ? hipMemMap((char*)va + top, s, 0, h, 0) : e;
            hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess((char*)va + top, s, &acc, 1) : e2;
            std::printf("non-uniform %5.2f GB at +%5.2f GB: %s / %s / %s\n", s / 1073741824.0, top / 1073741824.0, hipGetErrorString(e),
                        hipGetErrorString(e2), hipGetErrorString(e3));
            if (e3 != hipSuccess) { IGN(hipGetLastError()); break; }
            top += s;
        }
    }
    return 0;
}
