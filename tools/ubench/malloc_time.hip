// how long hipMalloc / hipMemsetAsync / hipFree take against the size of the allocation, and what a free costs the NEXT allocation of a
// different size (the constructor's fp64 fronts are one allocation of 2.2 GB at 1M vertices and ~10 GB at 4M; a remesh frees one set and
// allocates a slightly different one): hipcc -O2 tools/ubench/malloc_time.hip -o tools/build/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; ++rep)
        for (double gb : {1.0, 4.0, 10.0}) {
            const size_t n = (size_t)(gb * (1ull << 30));
            void* p = nullptr;
            double t0 = now();
            if (hipMalloc(&p, n) != hipSuccess) { printf("%.1f GB: failed\n", gb); continue; }
            double t1 = now();
            (void)hipMemsetAsync(p, 0, n, 0); (void)hipStreamSynchronize(0);
            double t2 = now();
            (void)hipFree(p);
            double t3 = now();
            printf("same size again: rep %d  %5.1f GB: malloc %8.2f ms  memset %8.2f ms  free %8.2f ms\n", rep, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
    // a set of buffers like the constructor's at 4M (GB): freed, then a set 3 % larger
    const double set[] = {10.0, 2.5, 2.5, 1.2, 0.6, 0.6, 0.3};
    for (int rep = 0; rep < 4; ++rep) {
        std::vector<void*> ps;
        double t0 = now();
        for (double gb : set) { void* p = nullptr; (void)hipMalloc(&p, (size_t)(gb * (1.0 + 0.03 * rep) * (1ull << 30))); ps.push_back(p); }
        double t1 = now();
        for (size_t i = 0; i < ps.size(); ++i) (void)hipMemsetAsync(ps[i], 0, (size_t)(set[i] * (1.0 + 0.03 * rep) * (1ull << 30)), 0);
        (void)hipStreamSynchronize(0);
        double t2 = now();
        for (void* p : ps) (void)hipFree(p);
        double t3 = now();
        printf("set x %.2f: malloc %8.2f ms  memset %8.2f ms  free %8.2f ms\n", 1.0 + 0.03 * rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    return 0;
}
