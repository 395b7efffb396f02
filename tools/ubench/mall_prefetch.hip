// mall_prefetch.hip -- can a second stream pull the NEXT tree level's factor rows into the Infinity Cache while the current level's
// kernel runs? A chain of 9 dependent streaming kernels (35 MB each, the shape of the upper-level launches of a 1M-vertex re-solve:
// ~1000 workgroups of 256 threads, 16-byte loads) is timed alone, with all its buffers already cache resident, and with a chain of
// touch kernels (one dword per 64-byte line) on a second stream that starts with it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mall_prefetch.hip -o tools/build/mall_prefetch && tools/build/mall_prefetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// one "level": workgroup b reads its contiguous share and leaves one number; `dep` is read first (the previous level's output: the chain is dependent)
__global__ __launch_bounds__(256) void k_level(const f4* __restrict__ buf, size_t n4, const float* dep, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        f4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const size_t j = i + r * 256; v[r] = j < hi ? buf[j] : f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += v[r];
    }
    const float d = dep ? dep[blockIdx.x % 64] : 0.0f;
    float s = acc[0] + acc[1] + acc[2] + acc[3] + d;
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// touch: one dword of every `stride`-byte line
__global__ __launch_bounds__(256) void k_touch(const float* __restrict__ buf, size_t nlines, int stride_f, float* sink) {
    float s = 0.f;
    for (size_t l = (size_t)blockIdx.x * 256 + threadIdx.x; l < nlines; l += (size_t)gridDim.x * 256) s += buf[l * stride_f];
    if (s == 123.456f) sink[0] = s;
}

int main() {
    const int L = 9;
    const size_t bytes = 35u << 20, n4 = bytes / 16;
    std::vector<f4*> lv(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&lv[l], bytes)); CK(hipMemset(lv[l], 0, bytes)); }
    f4* flush; const size_t fbytes = (size_t)700 << 20; CK(hipMalloc(&flush, fbytes)); CK(hipMemset(flush, 0, fbytes));
    float *out[2], *sink; CK(hipMalloc(&out[0], 4096 * 4)); CK(hipMalloc(&out[1], 4096 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMemset(out[0], 0, 4096 * 4)); CK(hipMemset(out[1], 0, 4096 * 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int G = 1016;
    auto do_flush = [&] { hipLaunchKernelGGL(k_level, dim3(2048), dim3(256), 0, s1, flush, fbytes / 16, (const float*)nullptr, out[1]); };
    auto chain = [&](int first, int last) { for (int l = first; l < last; ++l) hipLaunchKernelGGL(k_level, dim3(G), dim3(256), 0, s1, lv[l], n4, (const float*)out[l & 1], out[(l + 1) & 1]); };
    auto run = [&](const char* what, int mode, int stride_b, int tgrid) {
        std::vector<float> t;
        for (int rep = 0; rep < 12; ++rep) {
            if (mode != 1) do_flush();
            if (mode == 1) chain(0, L);                       // warm: the chain itself ran just before (9 x 35 = 315 MB > 256 MB: only partly resident)
            CK(hipEventRecord(e0, s1));
            if (mode >= 2) {                                  // fork: the touch chain starts with the timed chain
                CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
                for (int l = (mode == 3 ? 1 : 0); l < L; ++l) hipLaunchKernelGGL(k_touch, dim3(tgrid), dim3(256), 0, s2, (const float*)lv[l], bytes / stride_b, stride_b / 4, sink);
                CK(hipEventRecord(ej, s2));
            }
            chain(0, L);
            if (mode >= 2) CK(hipStreamWaitEvent(s1, ej, 0));
            CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%-100s median %7.1f us  (min %7.1f)  = %.1f us per level\n", what, t[t.size() / 2], t[0], t[t.size() / 2] / L);
    };
    run("chain of 9 levels x 35 MB, cold (700 MB read in front)", 0, 0, 0);
    run("the same chain run twice in a row (second run timed: what is still cache resident)", 1, 0, 0);
    for (int tgrid : {64, 256, 1024}) {
        char name[160];
        snprintf(name, sizeof name, "cold + touch chain on a second stream (1 dword per 64 B, %d workgroups), all levels", tgrid);
        run(name, 2, 64, tgrid);
        snprintf(name, sizeof name, "cold + touch chain on a second stream (1 dword per 128 B, %d workgroups), all levels", tgrid);
        run(name, 2, 128, tgrid);
        snprintf(name, sizeof name, "cold + touch chain (1 dword per 64 B, %d workgroups), levels 1.. only", tgrid);
        run(name, 3, 64, tgrid);
    }
    // a single level: cold, and straight after itself
    {
        std::vector<float> tc, tw;
        for (int rep = 0; rep < 10; ++rep) {
            do_flush();
            CK(hipEventRecord(e0, s1)); chain(0, 1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tc.push_back(ms * 1e3f);
            CK(hipEventRecord(e0, s1)); chain(0, 1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); tw.push_back(ms * 1e3f);
        }
        std::sort(tc.begin(), tc.end()); std::sort(tw.begin(), tw.end());
        printf("one level (35 MB): cold %.1f us, straight after itself %.1f us\n", tc[5], tw[5]);
    }
    return 0;
}
