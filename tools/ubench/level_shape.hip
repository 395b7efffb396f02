// level_shape.hip -- how fast can ONE dependent launch stream a tree level's 35 MB (and 50 MB), as a function of its shape: workgroups, threads per
// workgroup, 16-byte loads in flight per thread. Chain of 9 such launches (each reads the previous one's output first), cold (700 MB read in front).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/level_shape.hip -o tools/build/level_shape && tools/build/level_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int T, int U>
__global__ __launch_bounds__(T) void k_level(const f4* __restrict__ buf, size_t n4, const float* dep, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)T * U) {
        f4 v[U];
#pragma unroll
        for (int r = 0; r < U; ++r) { const size_t j = i + (size_t)r * T; v[r] = j < hi ? buf[j] : f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < U; ++r) acc += v[r];
    }
    const float d = dep ? dep[blockIdx.x % 64] : 0.0f;
    float s = acc[0] + acc[1] + acc[2] + acc[3] + d;
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < T / 64; ++w) t += red[w]; out[blockIdx.x] = t; }
}

int main() {
    const int L = 9;
    hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f4* flush; const size_t fbytes = (size_t)700 << 20; CK(hipMalloc(&flush, fbytes)); CK(hipMemset(flush, 0, fbytes));
    float* out[2]; CK(hipMalloc(&out[0], 65536 * 4)); CK(hipMalloc(&out[1], 65536 * 4)); CK(hipMemset(out[0], 0, 65536 * 4)); CK(hipMemset(out[1], 0, 65536 * 4));
    for (size_t mb : {35, 50, 16}) {
        const size_t bytes = mb << 20, n4 = bytes / 16;
        std::vector<f4*> lv(L);
        for (int l = 0; l < L; ++l) { CK(hipMalloc(&lv[l], bytes)); CK(hipMemset(lv[l], 0, bytes)); }
        auto run = [&](const char* what, auto launch) {
            std::vector<float> t;
            for (int rep = 0; rep < 10; ++rep) {
                hipLaunchKernelGGL((k_level<256, 4>), dim3(2048), dim3(256), 0, s1, flush, fbytes / 16, (const float*)nullptr, out[1]);
                CK(hipEventRecord(e0, s1));
                for (int l = 0; l < L; ++l) launch(lv[l], n4, (const float*)out[l & 1], out[(l + 1) & 1]);
                CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) t.push_back(ms * 1e3f);
            }
            std::sort(t.begin(), t.end());
            printf("%3zu MB per level  %-44s %6.2f us per level (min %6.2f)  %.2f TB/s\n", mb, what, t[t.size() / 2] / L, t[0] / L, bytes / (t[t.size() / 2] / L) * 1e-6);
        };
#define SHAPE(T, U, G) { char nm[64]; snprintf(nm, sizeof nm, "%d workgroups x %d threads, %d loads in flight", G, T, U); \
        run(nm, [&](const f4* b, size_t n, const float* dep, float* o) { hipLaunchKernelGGL((k_level<T, U>), dim3(G), dim3(T), 0, s1, b, n, dep, o); }); }
        SHAPE(256, 4, 256) SHAPE(256, 4, 512) SHAPE(256, 4, 1024) SHAPE(256, 4, 2048) SHAPE(256, 4, 4096)
        SHAPE(256, 8, 512) SHAPE(256, 8, 1024) SHAPE(256, 8, 2048)
        SHAPE(256, 16, 256) SHAPE(256, 16, 512) SHAPE(256, 16, 1024)
        SHAPE(512, 4, 512) SHAPE(512, 8, 256) SHAPE(512, 8, 512) SHAPE(512, 4, 1024)
        SHAPE(1024, 4, 256) SHAPE(1024, 8, 256) SHAPE(1024, 2, 512)
        SHAPE(256, 2, 2048) SHAPE(256, 2, 4096) SHAPE(256, 1, 8192) SHAPE(64, 8, 4096) SHAPE(64, 16, 2048)
        for (int l = 0; l < L; ++l) CK(hipFree(lv[l]));
    }
    return 0;
}
