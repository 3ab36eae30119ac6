// GPU box probe: host-to-device copies straight out of a file mapping (tmpfs page cache) against an anonymous
// huge-page buffer.  build: hipcc -O2 -o /tmp/h2d_mmap_probe tests/micro/h2d_mmap_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t n = 6ull << 30;
    const char* path = "/dev/shm/h2d_probe.bin";
    int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (ftruncate(fd, n)) return 1;
    char* m = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    memset(m, 'A', n);
    munmap(m, n);
    void* d; CK(hipMalloc(&d, n));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = now();
        char* r = (char*)mmap(nullptr, n, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        auto t1 = now();
        CK(hipMemcpy(d, r, n, hipMemcpyHostToDevice));
        auto t2 = now();
        munmap(r, n);
        printf("file mapping (populate): map %.3f s, H2D %.3f s = %.1f GB/s\n", secs(t0, t1), secs(t1, t2), n / secs(t1, t2) / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        char* r = (char*)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        auto t1 = now();
        CK(hipMemcpy(d, r, n, hipMemcpyHostToDevice));
        auto t2 = now();
        munmap(r, n);
        printf("file mapping (lazy): map %.3f s, H2D %.3f s = %.1f GB/s\n", secs(t0, t1), secs(t1, t2), n / secs(t1, t2) / 1e9);
    }
    {
        char* a = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(a, n, MADV_HUGEPAGE);
        memset(a, 'C', n);
        for (int rep = 0; rep < 2; rep++) {
            auto t1 = now();
            CK(hipMemcpy(d, a, n, hipMemcpyHostToDevice));
            auto t2 = now();
            printf("anonymous huge pages: H2D %.3f s = %.1f GB/s\n", secs(t1, t2), n / secs(t1, t2) / 1e9);
        }
        // read() into the arena with 16 threads would be the alternative: time a plain single-thread read() for scale
        auto t0 = now();
        size_t at = 0; lseek(fd, 0, SEEK_SET);
        while (at < n) { ssize_t k = read(fd, a + at, 1 << 30); if (k <= 0) break; at += k; }
        printf("read() of the file into the buffer, one thread: %.3f s = %.1f GB/s\n", secs(t0, now()), n / secs(t0, now()) / 1e9);
    }
    unlink(path);
    return 0;
}
