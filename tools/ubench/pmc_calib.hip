// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 by access width (NOT part of the product):
// every kernel moves a known number of bytes with one access shape. MI355X_MICROARCH.md states that FETCH_SIZE reports HALF the
// bytes of wide (16 B per lane) coalesced streaming reads and calls other widths uncalibrated; the solver's kernels mix 16-byte
// factor streams with 4- and 12-byte vector accesses and index gathers.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pmc_calib.hip -o tools/build/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- tools/build/pmc_calib     (and --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr long long N = 32ll << 20;          // floats: 128 MiB per array

__global__ void read4(const float* __restrict__ a, float* out) {          // 4 B per lane, coalesced
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) s += a[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void read8(const float2* __restrict__ a, float* out) {
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N / 2; i += (long long)gridDim.x * blockDim.x) { const float2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678f) out[0] = s;
}
struct f3 { float x, y, z; };
__global__ void read12(const f3* __restrict__ a, float* out) {            // 12-byte rows, consecutive lanes consecutive rows
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N / 3; i += (long long)gridDim.x * blockDim.x) { const f3 v = a[i]; s += v.x + v.y + v.z; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void read16(const float4* __restrict__ a, float* out) {
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N / 4; i += (long long)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void gather12(const f3* __restrict__ a, const int* __restrict__ idx, long long n, float* out) {     // 12-byte rows at random positions
    float s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { const f3 v = a[idx[i]]; s += v.x + v.y + v.z; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void write4(float* a) { for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) a[i] = 1.0f; }
__global__ void write12(f3* a) { for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N / 3; i += (long long)gridDim.x * blockDim.x) a[i] = f3{1.f, 2.f, 3.f}; }
__global__ void write16(float4* a) { for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N / 4; i += (long long)gridDim.x * blockDim.x) a[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ void scatter12(f3* a, const int* __restrict__ idx, long long n) {       // a permutation: every row written once
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a[idx[i]] = f3{1.f, 2.f, 3.f};
}

int main() {
    float *a, *out; int* idx;
    const long long rows = N / 3;
    CK(hipMalloc(&a, N * 4 + 64)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, rows * 4));
    CK(hipMemset(a, 0, N * 4));
    std::vector<int> h(rows);
    for (long long i = 0; i < rows; ++i) h[i] = (int)i;
    unsigned long long s = 88172645463325252ull;
    for (long long i = rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const long long j = (long long)(s % (unsigned long long)(i + 1)); std::swap(h[i], h[j]); }
    CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    const int G = 2048, B = 256;
    printf("bytes per kernel: read4/8/16 %lld, read12 %lld, gather12 %lld data + %lld index, write4/16 %lld, write12 / scatter12 %lld (+ %lld index read)\n",
           N * 4, rows * 12, rows * 12, rows * 4, N * 4, rows * 12, rows * 4);
    hipLaunchKernelGGL(read4, dim3(G), dim3(B), 0, 0, a, out);
    hipLaunchKernelGGL(read8, dim3(G), dim3(B), 0, 0, (const float2*)a, out);
    hipLaunchKernelGGL(read12, dim3(G), dim3(B), 0, 0, (const f3*)a, out);
    hipLaunchKernelGGL(read16, dim3(G), dim3(B), 0, 0, (const float4*)a, out);
    hipLaunchKernelGGL(gather12, dim3(G), dim3(B), 0, 0, (const f3*)a, (const int*)idx, rows, out);
    hipLaunchKernelGGL(write4, dim3(G), dim3(B), 0, 0, a);
    hipLaunchKernelGGL(write12, dim3(G), dim3(B), 0, 0, (f3*)a);
    hipLaunchKernelGGL(write16, dim3(G), dim3(B), 0, 0, (float4*)a);
    hipLaunchKernelGGL(scatter12, dim3(G), dim3(B), 0, 0, (f3*)a, (const int*)idx, rows);
    CK(hipDeviceSynchronize());
    return 0;
}
