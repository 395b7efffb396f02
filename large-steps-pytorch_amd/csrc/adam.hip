// adam.hip -- AdamUniform step (largesteps/optimize.py:18-41): Adam whose per-element sqrt(v_hat) is
// replaced by one global max. Two kernels, no host sync: (1) moment update + per-workgroup max of
// sqrt(g2/(1-b2^t)), (2) every workgroup reduces the <= 1024 partial maxima and applies the step.
#include "common.h"
#include <algorithm>
#include <math.h>

namespace ls {

// max that keeps a NaN (fmaxf drops it): the reference's `m2.sqrt().max()` (optimize.py:40) is NaN as soon as one
// gradient is, which poisons every parameter and makes a divergence visible -- same here.
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }
__device__ __forceinline__ float wave_nan_max(float x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x = nan_max(x, __shfl_down(x, off, WAVE));
    return x;
}

// The element loops of both kernel pairs. 16-byte accesses when every array is 16-byte aligned (`vec`; the tail of < 4 elements goes
// to workgroup 0): 4-byte-per-lane streams level off near 3 TB/s on this chip, and the step is nothing but streams (round 5, 3M floats:
// 13.7 + 11.8 us -> see profiles/r05_adam.txt). Same operations per element in the same order, max is exact: same bits.
typedef float f4_adam __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float moments_one(float g, float& a, float& v, float b1, float b2, float inv_c2) {
    a = a * b1 + (1.0f - b1) * g;
    v = v * b2 + (1.0f - b2) * (g * g);
    return sqrtf(v * inv_c2);
}
__device__ __forceinline__ float moments_range(const float* __restrict__ grad, float* __restrict__ g1, float* __restrict__ g2, int64_t n,
                                               float b1, float b2, float inv_c2, bool vec) {
    float m = 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x, stride = (int64_t)gridDim.x * BLOCK;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            const f4_adam g = reinterpret_cast<const f4_adam*>(grad)[i];
            f4_adam a = reinterpret_cast<f4_adam*>(g1)[i], v = reinterpret_cast<f4_adam*>(g2)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float aq = a[q], vq = v[q];
                m = nan_max(m, moments_one(g[q], aq, vq, b1, b2, inv_c2));
                a[q] = aq; v[q] = vq;
            }
            reinterpret_cast<f4_adam*>(g1)[i] = a;
            reinterpret_cast<f4_adam*>(g2)[i] = v;
        }
        const int64_t i = (n4 << 2) + tid;
        if (i < n) {
            float a = g1[i], v = g2[i];
            m = nan_max(m, moments_one(grad[i], a, v, b1, b2, inv_c2));
            g1[i] = a; g2[i] = v;
        }
        return m;
    }
    for (int64_t i = tid; i < n; i += stride) {
        float a = g1[i], v = g2[i];
        m = nan_max(m, moments_one(grad[i], a, v, b1, b2, inv_c2));
        g1[i] = a; g2[i] = v;
    }
    return m;
}
__device__ __forceinline__ void apply_range(float* __restrict__ param, const float* __restrict__ g1, int64_t n, float lr, float inv_c1,
                                            float denom, bool vec) {
    const int64_t tid = (int64_t)blockIdx.x * BLOCK + threadIdx.x, stride = (int64_t)gridDim.x * BLOCK;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += stride) {
            f4_adam p = reinterpret_cast<f4_adam*>(param)[i];
            const f4_adam a = reinterpret_cast<const f4_adam*>(g1)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) p[q] = p[q] - lr * ((a[q] * inv_c1) / denom);
            reinterpret_cast<f4_adam*>(param)[i] = p;
        }
        const int64_t i = (n4 << 2) + tid;
        if (i < n) param[i] = param[i] - lr * ((g1[i] * inv_c1) / denom);
        return;
    }
    for (int64_t i = tid; i < n; i += stride) param[i] = param[i] - lr * ((g1[i] * inv_c1) / denom);
}

__global__ __launch_bounds__(BLOCK) void k_adam_moments(const float* __restrict__ grad, float* __restrict__ g1, float* __restrict__ g2,
                                                        int64_t n, float b1, float b2, float inv_c2, float* __restrict__ pmax, int vec) {
    __shared__ float s_max[BLOCK / WAVE];
    float m = moments_range(grad, g1, g2, n, b1, b2, inv_c2, vec != 0);
    m = wave_nan_max(m);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_max[threadIdx.x / WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 1; j < BLOCK / WAVE; ++j) m = nan_max(m, s_max[j]);
        pmax[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(BLOCK) void k_adam_apply(float* __restrict__ param, const float* __restrict__ g1, int64_t n, float lr,
                                                      float inv_c1, const float* __restrict__ pmax, int G, int vec) {
    __shared__ float s_max[BLOCK / WAVE];
    float m = 0.0f;
    for (int g = threadIdx.x; g < G; g += BLOCK) m = nan_max(m, pmax[g]);
    m = wave_nan_max(m);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_max[threadIdx.x / WAVE] = m;
    __syncthreads();
    m = s_max[0];
    for (int j = 1; j < BLOCK / WAVE; ++j) m = nan_max(m, s_max[j]);
    apply_range(param, g1, n, lr, inv_c1, 1e-8f + m, vec != 0);
}

// the same two kernels with the step count on the device (d_step[0] = steps done so far): nothing in the launch depends on the
// step number, so the pair can be replayed from a captured graph. d_step[1] carries the current step from the first kernel
// to the second, which publishes it as d_step[0] when it is done.
__device__ __forceinline__ float inv_bias(float beta, int t) { return (float)(1.0 / (1.0 - pow((double)beta, (double)t))); }

__global__ __launch_bounds__(BLOCK) void k_adam_moments_dev(const float* __restrict__ grad, float* __restrict__ g1, float* __restrict__ g2,
                                                            int64_t n, float b1, float b2, int* __restrict__ d_step, float* __restrict__ pmax, int vec) {
    __shared__ float s_max[BLOCK / WAVE];
    const int t = d_step[0] + 1;
    const float inv_c2 = inv_bias(b2, t);
    float m = moments_range(grad, g1, g2, n, b1, b2, inv_c2, vec != 0);
    m = wave_nan_max(m);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_max[threadIdx.x / WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 1; j < BLOCK / WAVE; ++j) m = nan_max(m, s_max[j]);
        pmax[blockIdx.x] = m;
        if (blockIdx.x == 0) d_step[1] = t;
    }
}

__global__ __launch_bounds__(BLOCK) void k_adam_apply_dev(float* __restrict__ param, const float* __restrict__ g1, int64_t n, float lr, float b1,
                                                          int* __restrict__ d_step, const float* __restrict__ pmax, int G, int vec) {
    __shared__ float s_max[BLOCK / WAVE];
    const int t = d_step[1];
    const float inv_c1 = inv_bias(b1, t);
    float m = 0.0f;
    for (int g = threadIdx.x; g < G; g += BLOCK) m = nan_max(m, pmax[g]);
    m = wave_nan_max(m);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_max[threadIdx.x / WAVE] = m;
    __syncthreads();
    m = s_max[0];
    for (int j = 1; j < BLOCK / WAVE; ++j) m = nan_max(m, s_max[j]);
    apply_range(param, g1, n, lr, inv_c1, 1e-8f + m, vec != 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) d_step[0] = t;          // nobody reads d_step[0] in this kernel
}

}  // namespace ls

using namespace ls;

static int adam_vec(const void* a, const void* b, const void* c, const void* d) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}
// at most 1024 workgroups (the partial maxima the second kernel reduces), each thread one or a few 16-byte (4-byte) elements
static int adam_grid(int64_t n, int vec) { return (int)std::min<int64_t>(div_up(std::max<int64_t>(vec ? (n + 3) / 4 : n, 1), BLOCK), 1024); }

extern "C" int ls_adam_uniform_step_device(float* param, const float* grad, float* g1, float* g2, int64_t n, float lr, float beta1,
                                           float beta2, int32_t* d_step, void* scratch, int device, void* stream) {
    LS_REQUIRE(n >= 0 && (n == 0 || (param && grad && g1 && g2)) && scratch && d_step, LS_E_INVALID, "ls_adam_uniform_step_device: null pointer or negative size");
    if (n == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    const int vec = adam_vec(param, grad, g1, g2), G = adam_grid(n, vec);
    hipLaunchKernelGGL(k_adam_moments_dev, dim3(G), dim3(BLOCK), 0, (hipStream_t)stream, grad, g1, g2, n, beta1, beta2, d_step, (float*)scratch, vec);
    hipLaunchKernelGGL(k_adam_apply_dev, dim3(G), dim3(BLOCK), 0, (hipStream_t)stream, param, g1, n, lr, beta1, d_step, (const float*)scratch, G, vec);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_adam_uniform_step(float* param, const float* grad, float* g1, float* g2, int64_t n, float lr, float beta1,
                                    float beta2, int step, void* scratch, int device, void* stream) {
    LS_REQUIRE(n >= 0 && (n == 0 || (param && grad && g1 && g2)) && scratch, LS_E_INVALID, "ls_adam_uniform_step: null pointer or negative size");
    LS_REQUIRE(step >= 1, LS_E_INVALID, "ls_adam_uniform_step: step must be >= 1");
    if (n == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    const int vec = adam_vec(param, grad, g1, g2), G = adam_grid(n, vec);
    const float inv_c1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    const float inv_c2 = (float)(1.0 / (1.0 - pow((double)beta2, (double)step)));
    hipLaunchKernelGGL(k_adam_moments, dim3(G), dim3(BLOCK), 0, (hipStream_t)stream, grad, g1, g2, n, beta1, beta2, inv_c2, (float*)scratch, vec);
    hipLaunchKernelGGL(k_adam_apply, dim3(G), dim3(BLOCK), 0, (hipStream_t)stream, param, g1, n, lr, inv_c1, (const float*)scratch, G, vec);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
