// fp64_rate.hip -- what a wave pays per fp64 instruction on gfx950: vector FMA (v_fma_f64) and matrix (v_mfma_f64_16x16x4_f64), one wave
// per SIMD, as a single workgroup (the in-register inverse of csrc/nd_factor.hip runs alone on the chip) and with every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fp64_rate.hip -o tools/build/fp64_rate && tools/build/fp64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(1024) void k_valu(double* out, int iters, long long* clk) {
    double a[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double x = 1.0000001, y = 1e-9 * (threadIdx.x + 1);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) a[i] = fma(a[i], x, y);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, long long* clk) {
    f64x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = 1e-3 * threadIdx.x, b = 1e-4 * (threadIdx.x + 3);
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    double* out; long long* clk;
    CK(hipMalloc(&out, sizeof(double) * 4096 * 1024)); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    auto time = [&](const char* what, auto launch, double ops_per_wave) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            if (rep == 2) printf("%-58s %8.3f ms  %7.2f ns per instruction and wave   clock64 %lld, wall_clock64 %lld ticks (%.1f / %.1f per us)\n", what, ms, ms * 1e6 / ops_per_wave,
                                 h[0], h[1], h[0] / (ms * 1e3), h[1] / (ms * 1e3));
        }
    };
    for (int blocks : {1, 256, 2048}) {
        char name[128];
        snprintf(name, sizeof name, "v_fma_f64, 16 independent, 256 threads x %d blocks", blocks);
        time(name, [&] { hipLaunchKernelGGL(k_valu<16>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 16.0 * iters);
        snprintf(name, sizeof name, "v_fma_f64, 64 independent, 256 threads x %d blocks", blocks);
        time(name, [&] { hipLaunchKernelGGL(k_valu<64>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 64.0 * iters);
        snprintf(name, sizeof name, "v_fma_f64, 16 independent, 1024 threads x %d blocks", blocks);
        time(name, [&] { hipLaunchKernelGGL(k_valu<16>, dim3(blocks), dim3(1024), 0, 0, out, iters, clk); }, 16.0 * iters);
        snprintf(name, sizeof name, "v_mfma_f64_16x16x4, 4 independent, 256 threads x %d blocks", blocks);
        time(name, [&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, clk); }, 4.0 * iters);
    }
    // the chain as the factorisation runs it: short single-workgroup kernels back to back
    {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_valu<64>, dim3(1), dim3(256), 0, 0, out, 200, clk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("32 launches of 1 workgroup x 200 x 64 v_fma_f64: %.1f us each, %.2f ns per instruction and wave\n", ms * 1e3 / 32, ms * 1e6 / 32 / (64.0 * 200));
    }
    return 0;
}
