// GPU box probe: what does a process pay between _Exit and its parent's waitpid?  usage: exit_probe [GB to map] [pinned GB] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <unistd.h>
#include <chrono>
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? atoi(argv[1]) : 0, pinned = argc > 2 ? atoi(argv[2]) : 0;
    const int threads = argc > 3 ? atoi(argv[3]) : 0;
    auto t0 = std::chrono::steady_clock::now();
    if (hipSetDevice(0) != hipSuccess) return 2;
    hipStream_t s; hipStreamCreate(&s);
    void* p = nullptr;
    if (gb) { if (hipMalloc(&p, gb << 30) != hipSuccess) return 3; hipMemsetAsync(p, 1, gb << 30, s); }
    void* h = nullptr;
    if (pinned) { if (hipHostMalloc(&h, pinned << 30, hipHostMallocDefault) != hipSuccess) return 4; }
    std::vector<std::thread> th;
    for (int i = 0; i < threads; i++) th.emplace_back([] { usleep(100000); });
    for (auto& t : th) t.join();
    hipStreamSynchronize(s);
    fprintf(stderr, "up %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    _Exit(0);
}
