// spd_inverse.hip -- the in-register Gauss-Jordan inverse of csrc/nd_factor.hip (one workgroup per SPD matrix of up to 128 rows) with its thread
// grid as a parameter: G x G threads, each owning an (NB / G) x (NB / G) block. The factorisation runs 32 of these one after the other at 1M vertices
// (89 us each with 16 x 16 threads and 8 x 8 doubles per thread: one wave per SIMD); more, smaller threads put 2 or 4 waves on a SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/spd_inverse.hip -o tools/build/spd_inverse && tools/build/spd_inverse
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NB, int GX, int GY>
__global__ __launch_bounds__(GX * GY) void k_inv(const double* __restrict__ M, double* __restrict__ X, int n) {
    constexpr int RX = NB / GX, RY = NB / GY;            // columns / rows per thread
    __shared__ double rowb[2][NB], colb[2][NB];
    const int bx = threadIdx.x % GX, by = threadIdx.x / GX;
    double a[RY][RX];
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int c = 0; c < RX; ++c) { const int i = by * RY + r, j = bx * RX + c; a[r][c] = (i < n && j < n) ? M[(size_t)i * n + j] : 0.0; }
    if (by == 0) {
#pragma unroll
        for (int c = 0; c < RX; ++c) rowb[0][bx * RX + c] = a[0][c];
    }
    if (bx == 0) {
#pragma unroll
        for (int r = 0; r < RY; ++r) colb[0][by * RY + r] = a[r][0];
    }
    __syncthreads();
    // the k loop is unrolled over lcm(RX, RY) steps so that the register that holds row / column k is a compile-time index
    constexpr int UN = RX > RY ? RX : RY;
    for (int kb = 0; kb * UN < n; ++kb) {
#pragma unroll
        for (int kk = 0; kk < UN; ++kk) {
            const int k = kb * UN + kk;
            if (k < n) {
                const int cur = kk & 1, nxt = cur ^ 1;
                const double p = colb[cur][k];
                const double ip = 1.0 / p;
                double rr[RX], ck[RY];
#pragma unroll
                for (int c = 0; c < RX; ++c) rr[c] = rowb[cur][bx * RX + c] * ip;
#pragma unroll
                for (int r = 0; r < RY; ++r) ck[r] = colb[cur][by * RY + r];
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int c = 0; c < RX; ++c) a[r][c] = fma(-ck[r], rr[c], a[r][c]);
                const int kx = k / RX, cx = kk % RX, ky = k / RY, ry = kk % RY;       // owner thread column / row, register inside
                if (bx == kx) {
#pragma unroll
                    for (int r = 0; r < RY; ++r) a[r][cx] = -ck[r] * ip;
                }
                if (by == ky) {
#pragma unroll
                    for (int c = 0; c < RX; ++c) a[ry][c] = rr[c];
                    if (bx == kx) a[ry][cx] = ip;
                }
                const int k1 = k + 1, kx1 = k1 / RX, cx1 = (kk + 1) % RX, ky1 = k1 / RY, ry1 = (kk + 1) % RY;
                if (by == ky1) {
#pragma unroll
                    for (int c = 0; c < RX; ++c) rowb[nxt][bx * RX + c] = a[ry1][c];
                }
                if (bx == kx1) {
#pragma unroll
                    for (int r = 0; r < RY; ++r) colb[nxt][by * RY + r] = a[r][cx1];
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int c = 0; c < RX; ++c) { const int i = by * RY + r, j = bx * RX + c; if (i < n && j < n) X[(size_t)i * n + j] = a[r][c]; }
}

int main() {
    const int n = 125, NB = 128;
    std::vector<double> A((size_t)n * n), B((size_t)n * n), X((size_t)n * n);
    srand(1);
    for (auto& v : B) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = i == j ? (double)n : 0.0; for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * B[(size_t)j * n + k]; A[(size_t)i * n + j] = s; }
    double *dA, *dX; CK(hipMalloc(&dA, sizeof(double) * n * n)); CK(hipMalloc(&dX, sizeof(double) * n * n));
    CK(hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* what, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 32; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(X.data(), dX, sizeof(double) * n * n, hipMemcpyDeviceToHost));
        double err = 0.0;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * X[(size_t)k * n + j]; err = fmax(err, fabs(s - (i == j ? 1.0 : 0.0))); }
        printf("%-46s %7.1f us per %d x %d inverse (32 back to back)   max |A X - I| %.2e\n", what, ms * 1e3 / 32, n, n, err);
    };
    (void)NB;
    run("16 x 16 threads, 8 x 8 per thread (product)", [&] { hipLaunchKernelGGL((k_inv<128, 16, 16>), dim3(1), dim3(256), 0, 0, dA, dX, n); });
    run("32 x 16 threads, 8 rows x 4 columns per thread", [&] { hipLaunchKernelGGL((k_inv<128, 32, 16>), dim3(1), dim3(512), 0, 0, dA, dX, n); });
    run("16 x 32 threads, 4 rows x 8 columns per thread", [&] { hipLaunchKernelGGL((k_inv<128, 16, 32>), dim3(1), dim3(512), 0, 0, dA, dX, n); });
    run("32 x 32 threads, 4 x 4 per thread", [&] { hipLaunchKernelGGL((k_inv<128, 32, 32>), dim3(1), dim3(1024), 0, 0, dA, dX, n); });
    return 0;
}
