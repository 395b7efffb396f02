// spmv.hip -- y[V,k] = M x[V,k] on CSR (to_differential, largesteps/parameterize.py:30) + library plumbing.
#include "spmv_kernels.h"
#include <string.h>
#include <algorithm>

namespace ls {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();   // clear the sticky error so that later calls report their own
    return (int)e > 0 ? (int)e : 999;
}

// contiguous right-hand sides: x, y are (V,K) with leading dimension K
template <int K, int VARIANT>
__global__ __launch_bounds__(BLOCK) void k_spmv(CsrView A, const float* __restrict__ x, float* __restrict__ y, int64_t V,
                                                int T, int G) {
    __shared__ __attribute__((aligned(16))) int2 s_cv[VARIANT == 0 ? LDS_CAP : 1];
    const TileSched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t r0 = (int64_t)tile * TILE_ROWS, r1 = min(r0 + (int64_t)TILE_ROWS, V);
        const int64_t i = r0 + threadIdx.x;
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
        if (VARIANT == 0) row_csr_lds<K>(A, x, r0, r1, s_cv, acc);
        else if (i < r1) row_csr_direct<K>(A, x, i, acc);
        if (i < r1) {
            Vec<K> o;
#pragma unroll
            for (int q = 0; q < K; ++q) o.v[q] = acc[q];
            reinterpret_cast<Vec<K>*>(y)[i] = o;
        }
    }
}

// generic leading dimension: columns [c0, c0+K) of (V,ld) arrays (only used for k > 4)
template <int K>
__global__ __launch_bounds__(BLOCK) void k_spmv_strided(CsrView A, const float* __restrict__ x, float* __restrict__ y, int64_t V,
                                                        int ld, int c0) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= V) return;
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    const int s = A.rowptr[i], e = A.rowptr[i + 1];
    for (int j = s; j < e; ++j) {
        const float v = A.val[j];
        const float* xr = x + (size_t)A.col[j] * ld + c0;
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = fmaf(v, xr[q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < K; ++q) y[(size_t)i * ld + c0 + q] = acc[q];
}

template <int K>
static void launch_spmv(int variant, CsrView A, const float* x, float* y, int64_t V, hipStream_t st) {
    const int T = div_up(V, TILE_ROWS);
    const int G = T < 8 ? T : std::min(T & ~7, 4096);
    if (variant == 0) hipLaunchKernelGGL((k_spmv<K, 0>), dim3(G), dim3(BLOCK), 0, st, A, x, y, V, T, G);
    else hipLaunchKernelGGL((k_spmv<K, 1>), dim3(G), dim3(BLOCK), 0, st, A, x, y, V, T, G);
}

}  // namespace ls

using namespace ls;

extern "C" int ls_version(void) { return LS_VERSION; }
extern "C" const char* ls_last_error(void) { return g_err; }

extern "C" int ls_spmv(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V, int64_t nnz, const float* x,
                       float* y, int k, int variant, int device, void* stream) {
    LS_REQUIRE(V >= 0 && nnz >= 0 && rowptr && (nnz == 0 || (col && val)) && (V == 0 || (x && y)), LS_E_INVALID,
               "ls_spmv: null pointer or negative size");
    LS_REQUIRE(k >= 1 && k <= 64, LS_E_INVALID, "ls_spmv: k=%d outside [1,64]", k);
    LS_REQUIRE(variant == 0 || variant == 1, LS_E_INVALID, "ls_spmv: unknown variant %d", variant);
    LS_REQUIRE(x != y, LS_E_INVALID, "ls_spmv: x and y must not alias");
    if (V == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    CsrView A{rowptr, col, val};
    if (k <= 4) {
        switch (k) {
            case 1: launch_spmv<1>(variant, A, x, y, V, st); break;
            case 2: launch_spmv<2>(variant, A, x, y, V, st); break;
            case 3: launch_spmv<3>(variant, A, x, y, V, st); break;
            default: launch_spmv<4>(variant, A, x, y, V, st); break;
        }
    } else {
        const int grid = div_up(V, BLOCK);
        for (int c0 = 0; c0 < k; c0 += 4) {
            const int kk = std::min(4, k - c0);
            switch (kk) {
                case 1: hipLaunchKernelGGL(k_spmv_strided<1>, dim3(grid), dim3(BLOCK), 0, st, A, x, y, V, k, c0); break;
                case 2: hipLaunchKernelGGL(k_spmv_strided<2>, dim3(grid), dim3(BLOCK), 0, st, A, x, y, V, k, c0); break;
                case 3: hipLaunchKernelGGL(k_spmv_strided<3>, dim3(grid), dim3(BLOCK), 0, st, A, x, y, V, k, c0); break;
                default: hipLaunchKernelGGL(k_spmv_strided<4>, dim3(grid), dim3(BLOCK), 0, st, A, x, y, V, k, c0); break;
            }
        }
    }
    LS_HIP(hipGetLastError());
    return LS_OK;
}
