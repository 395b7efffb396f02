// experiments.hip -- A/B kernels used to choose the structure of the production PCG kernels (tools/ubench.py).
// Not on the product path: nothing in largesteps/ calls ls_experiment. K = 3 right-hand sides only.
#include "spmv_kernels.h"

namespace ls {
namespace ex {

struct f3 { float x, y, z; };

// ---- K3-like:  p = di * r + beta * p --------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(BS) void k3_row(const float* __restrict__ dinv, const float* __restrict__ r, float* __restrict__ p, int64_t V, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < V; i += (int64_t)gridDim.x * BS) {
        const f3 rv = reinterpret_cast<const f3*>(r)[i];
        f3 pv = reinterpret_cast<const f3*>(p)[i];
        const float di = dinv[i];
        pv.x = fmaf(beta, pv.x, di * rv.x); pv.y = fmaf(beta, pv.y, di * rv.y); pv.z = fmaf(beta, pv.z, di * rv.z);
        reinterpret_cast<f3*>(p)[i] = pv;
    }
}

// 4 rows (12 floats) per thread as three 16-byte accesses per vector
__device__ __forceinline__ void k3_chunk(const float4* __restrict__ d4, const float4* __restrict__ r4, float4* __restrict__ p4, int64_t c, float beta) {
    const float4 d = d4[c];
    const float4 r0 = r4[3 * c], r1 = r4[3 * c + 1], r2 = r4[3 * c + 2];
    float4 p0 = p4[3 * c], p1 = p4[3 * c + 1], p2 = p4[3 * c + 2];
    p0.x = fmaf(beta, p0.x, d.x * r0.x); p0.y = fmaf(beta, p0.y, d.x * r0.y); p0.z = fmaf(beta, p0.z, d.x * r0.z); p0.w = fmaf(beta, p0.w, d.y * r0.w);
    p1.x = fmaf(beta, p1.x, d.y * r1.x); p1.y = fmaf(beta, p1.y, d.y * r1.y); p1.z = fmaf(beta, p1.z, d.z * r1.z); p1.w = fmaf(beta, p1.w, d.z * r1.w);
    p2.x = fmaf(beta, p2.x, d.z * r2.x); p2.y = fmaf(beta, p2.y, d.w * r2.y); p2.z = fmaf(beta, p2.z, d.w * r2.z); p2.w = fmaf(beta, p2.w, d.w * r2.w);
    p4[3 * c] = p0; p4[3 * c + 1] = p1; p4[3 * c + 2] = p2;
}

template <int BS>
__global__ __launch_bounds__(BS) void k3_vec(const float* __restrict__ dinv, const float* __restrict__ r, float* __restrict__ p, int64_t V, float beta) {
    const int64_t C = V / 4;
    for (int64_t c = (int64_t)blockIdx.x * BS + threadIdx.x; c < C; c += (int64_t)gridDim.x * BS)
        k3_chunk(reinterpret_cast<const float4*>(dinv), reinterpret_cast<const float4*>(r), reinterpret_cast<float4*>(p), c, beta);
}

// same + a scalar hand-off prologue (reduce N partial arrays of G doubles, as the PCG kernels do)
template <int BS, int N>
__global__ __launch_bounds__(BS) void k3_vec_prologue(const float* __restrict__ dinv, const float* __restrict__ r, float* __restrict__ p, int64_t V,
                                                      const double* __restrict__ part, int G) {
    __shared__ double smem[(BS / WAVE + 1) * N];
    const int64_t C = V / 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = threadIdx.x < G ? part[(size_t)n * 1024 + threadIdx.x] : 0.0;
#pragma unroll
    for (int n = 0; n < N; ++n) { acc[n] = wave_sum(acc[n]); if (lane == 0) smem[N + w * N + n] = acc[n]; }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) { double v = lane < BS / WAVE ? smem[N + lane * N + n] : 0.0; v = wave_sum(v); if (lane == 0) smem[n] = v; }
    }
    __syncthreads();
    const float beta = (float)(smem[0] * 1e-30) + 0.5f;
    for (int64_t c = (int64_t)blockIdx.x * BS + threadIdx.x; c < C; c += (int64_t)gridDim.x * BS)
        k3_chunk(reinterpret_cast<const float4*>(dinv), reinterpret_cast<const float4*>(r), reinterpret_cast<float4*>(p), c, beta);
}

// loads first, prologue second (hand-off hidden under the loads), single chunk per thread
template <int BS, int N>
__global__ __launch_bounds__(BS) void k3_vec_hoist(const float* __restrict__ dinv, const float* __restrict__ r, float* __restrict__ p, int64_t V,
                                                   const double* __restrict__ part, int G) {
    __shared__ double smem[(BS / WAVE + 1) * N];
    const int64_t C = V / 4;
    const int64_t c = (int64_t)blockIdx.x * BS + threadIdx.x;
    const float4* d4 = reinterpret_cast<const float4*>(dinv);
    const float4* r4 = reinterpret_cast<const float4*>(r);
    float4* p4 = reinterpret_cast<float4*>(p);
    float4 d, r0, r1, r2, p0, p1, p2;
    const bool ok = c < C;
    if (ok) { d = d4[c]; r0 = r4[3 * c]; r1 = r4[3 * c + 1]; r2 = r4[3 * c + 2]; p0 = p4[3 * c]; p1 = p4[3 * c + 1]; p2 = p4[3 * c + 2]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = threadIdx.x < G ? part[(size_t)n * 1024 + threadIdx.x] : 0.0;
#pragma unroll
    for (int n = 0; n < N; ++n) { acc[n] = wave_sum(acc[n]); if (lane == 0) smem[N + w * N + n] = acc[n]; }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) { double v = lane < BS / WAVE ? smem[N + lane * N + n] : 0.0; v = wave_sum(v); if (lane == 0) smem[n] = v; }
    }
    __syncthreads();
    const float beta = (float)(smem[0] * 1e-30) + 0.5f;
    if (ok) {
        p0.x = fmaf(beta, p0.x, d.x * r0.x); p0.y = fmaf(beta, p0.y, d.x * r0.y); p0.z = fmaf(beta, p0.z, d.x * r0.z); p0.w = fmaf(beta, p0.w, d.y * r0.w);
        p1.x = fmaf(beta, p1.x, d.y * r1.x); p1.y = fmaf(beta, p1.y, d.y * r1.y); p1.z = fmaf(beta, p1.z, d.z * r1.z); p1.w = fmaf(beta, p1.w, d.z * r1.w);
        p2.x = fmaf(beta, p2.x, d.z * r2.x); p2.y = fmaf(beta, p2.y, d.w * r2.y); p2.z = fmaf(beta, p2.z, d.w * r2.z); p2.w = fmaf(beta, p2.w, d.w * r2.w);
        p4[3 * c] = p0; p4[3 * c + 1] = p1; p4[3 * c + 2] = p2;
    }
}

// plain float4 copy (the measured-achievable reference: read n, write n)
template <int BS>
__global__ __launch_bounds__(BS) void copy4(const float4* __restrict__ a, float4* __restrict__ b, int64_t n) {
    for (int64_t c = (int64_t)blockIdx.x * BS + threadIdx.x; c < n; c += (int64_t)gridDim.x * BS) b[c] = a[c];
}

// ---- K1-like: y = A x on SELL-64, K = 3 -------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(BS) void spmv_cur(SellView S, const float* __restrict__ x, float* __restrict__ y, int64_t V) {
    const int T = (int)((V + BS - 1) / BS);
    for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float acc[3] = {0.f, 0.f, 0.f};
        if ((i & ~(int64_t)63) < V) row_sell<3>(S, x, i, acc);
        if (i < V) { Vec<3> o; o.v[0] = acc[0]; o.v[1] = acc[1]; o.v[2] = acc[2]; reinterpret_cast<Vec<3>*>(y)[i] = o; }
    }
}

// matrix entries of the NEXT tile are requested before the gathers of the current one
template <int BS>
__global__ __launch_bounds__(BS) void spmv_prefetch(SellView S, const float* __restrict__ x, float* __restrict__ y, int64_t V) {
    const int T = (int)((V + BS - 1) / BS);
    const int lane = threadIdx.x & 63;
    int2 cur[8], nxt[8];
    int wcur = 0, wnxt = 0;
    auto fetch = [&](int tile, int2 (&c)[8], int& width) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        width = 0;
        if (tile < T && (i & ~(int64_t)63) < V) {
            const int slice = __builtin_amdgcn_readfirstlane((int)(i >> 6));
            const int off = __builtin_amdgcn_readfirstlane(S.slice_ptr[slice]);
            width = min(8, (__builtin_amdgcn_readfirstlane(S.slice_ptr[slice + 1]) - off) >> 6);
            const int2* __restrict__ q = S.cv + off + lane;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t < width) c[t] = q[(size_t)t * 64];
        }
    };
    int tile = blockIdx.x;
    fetch(tile, cur, wcur);
    while (tile < T) {
        fetch(tile + gridDim.x, nxt, wnxt);
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float acc[3] = {0.f, 0.f, 0.f};
        Vec<3> xv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < wcur) xv[t] = reinterpret_cast<const Vec<3>*>(x)[cur[t].x];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < wcur) {
            const float a = __int_as_float(cur[t].y);
            acc[0] = fmaf(a, xv[t].v[0], acc[0]); acc[1] = fmaf(a, xv[t].v[1], acc[1]); acc[2] = fmaf(a, xv[t].v[2], acc[2]);
        }
        if (i < V) { Vec<3> o; o.v[0] = acc[0]; o.v[1] = acc[1]; o.v[2] = acc[2]; reinterpret_cast<Vec<3>*>(y)[i] = o; }
#pragma unroll
        for (int t = 0; t < 8; ++t) cur[t] = nxt[t];
        wcur = wnxt;
        tile += gridDim.x;
    }
}

}  // namespace ex
}  // namespace ls

using namespace ls;

// which: 0 k3_row | 1 k3_vec | 2 k3_vec_prologue<6> | 3 k3_vec_hoist<6> | 4 copy4 (r -> p, 3V floats) | 10 spmv_cur | 11 spmv_prefetch
// bs in {256, 1024}; returns LS_E_INVALID for unknown combinations
extern "C" int ls_experiment(int which, int bs, int grid, int64_t V, const float* dinv, const float* r, float* p, const double* part,
                             const int32_t* slice_ptr, const void* cv, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const SellView S{slice_ptr, (const int2*)cv};
#define LS_EX(KERNEL, ...)                                                                               \
    do {                                                                                                 \
        if (bs == 1024) hipLaunchKernelGGL((KERNEL<1024>), dim3(grid), dim3(1024), 0, st, __VA_ARGS__);    \
        else if (bs == 256) hipLaunchKernelGGL((KERNEL<256>), dim3(grid), dim3(256), 0, st, __VA_ARGS__);  \
        else return LS_E_INVALID;                                                                        \
    } while (0)
    switch (which) {
        case 0: LS_EX(ex::k3_row, dinv, r, p, V, 0.5f); break;
        case 1: LS_EX(ex::k3_vec, dinv, r, p, V, 0.5f); break;
        case 2:
            if (bs == 1024) hipLaunchKernelGGL((ex::k3_vec_prologue<1024, 6>), dim3(grid), dim3(1024), 0, st, dinv, r, p, V, part, grid);
            else hipLaunchKernelGGL((ex::k3_vec_prologue<256, 6>), dim3(grid), dim3(256), 0, st, dinv, r, p, V, part, grid);
            break;
        case 3:
            if (bs == 1024) hipLaunchKernelGGL((ex::k3_vec_hoist<1024, 6>), dim3(grid), dim3(1024), 0, st, dinv, r, p, V, part, grid);
            else hipLaunchKernelGGL((ex::k3_vec_hoist<256, 6>), dim3(grid), dim3(256), 0, st, dinv, r, p, V, part, grid);
            break;
        case 4: LS_EX(ex::copy4, (const float4*)r, (float4*)p, (int64_t)(3 * V / 4)); break;
        case 10: LS_EX(ex::spmv_cur, S, r, p, V); break;
        case 11: LS_EX(ex::spmv_prefetch, S, r, p, V); break;
        default: return LS_E_INVALID;
    }
#undef LS_EX
    LS_HIP(hipGetLastError());
    return LS_OK;
}
