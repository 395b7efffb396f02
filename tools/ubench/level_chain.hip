// Micro-benchmark (NOT part of the product): what does one tree level of the direct solver's down sweep cost as a function of
// the number of DEPENDENT memory round trips in front of the arithmetic?  Synthetic level: n nodes, each an S x (S + B) fp32
// row-major block (zero padded to whole quads), a (S + B) x 3 vector per node, one output row per matrix row.
//   variant 0  record -> (matrix || vector) -> LDS -> FMA -> store        (the shipped kernels: a 64-byte record per tile first)
//   variant 1  (matrix || vector || record) -> LDS -> FMA -> store        (every address arithmetic in blockIdx)
//   variant 2  record -> index -> vector, matrix after record             (the shipped up sweep: perm -> b behind the record)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/level_chain.hip -o tools/ubench/build/level_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef CAPV
#define CAPV 16
#endif
constexpr int K = 3, CAP = CAPV, NW = 4;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
struct alignas(64) Rec { int node, row0, s, b; long long mat_off, vec_off; int pad[8]; };

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(m);
}
__device__ __forceinline__ float wave_sum63(float v) {
    v = dpp_step<0xB1, 0xf>(v); v = dpp_step<0x4E, 0xf>(v); v = dpp_step<0x141, 0xf>(v);
    v = dpp_step<0x140, 0xf>(v); v = dpp_step<0x142, 0xa>(v); v = dpp_step<0x143, 0xc>(v);
    return v;
}

template <int VARIANT>
__global__ __launch_bounds__(64 * NW) void k_level(const Rec* __restrict__ recs, const float* __restrict__ mat, const float* __restrict__ vecs,
                                                   const int* __restrict__ index, float* __restrict__ out, int S, int B, int R, int tiles_per_node) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int L = S + B, lpr = (L + 255) >> 8;
    int node = blockIdx.x / tiles_per_node, row0 = (blockIdx.x % tiles_per_node) * (NW * R);
    long long mat_off = (long long)node * S * L, vec_off = (long long)node * L;
    int s = S;
    if (VARIANT != 1) {
        const int wd = lane < 16 ? reinterpret_cast<const int*>(recs)[(size_t)blockIdx.x * 16 + lane] : 0;
        node = __builtin_amdgcn_readlane(wd, 0); row0 = __builtin_amdgcn_readlane(wd, 1); s = __builtin_amdgcn_readlane(wd, 2);
        mat_off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(wd, 5) << 32) | (unsigned)__builtin_amdgcn_readlane(wd, 4));
        vec_off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(wd, 7) << 32) | (unsigned)__builtin_amdgcn_readlane(wd, 6));
    }
    const int jw = row0 + w * R;
    const int wrows = max(0, min(R, s - jw));
    f4u a[CAP];
    {
        int row = 0, e = 0;
#pragma unroll
        for (int slot = 0; slot < CAP; ++slot) {
            const int tt = (e * 64 + lane) * 4;
            f4u z = {0.f, 0.f, 0.f, 0.f};
            a[slot] = (row < wrows && tt < L) ? *reinterpret_cast<const f4u*>(mat + mat_off + (size_t)(jw + row) * L + tt) : z;
            if (++e == lpr) { e = 0; ++row; }
        }
    }
    if (VARIANT == 1) s = recs[blockIdx.x].s;       // only needed for the store guard
    for (int u = threadIdx.x; u < L; u += blockDim.x) {
        size_t src = (size_t)vec_off + u;
        if (VARIANT == 2) src = (size_t)index[src];
#pragma unroll
        for (int q = 0; q < K; ++q) sm[u * K + q] = vecs[src * K + q];
    }
    __syncthreads();
    float mine[K] = {0.f, 0.f, 0.f}, acc[K] = {0.f, 0.f, 0.f};
    {
        int row = 0, e = 0;
#pragma unroll
        for (int slot = 0; slot < CAP; ++slot) {
            if (row < wrows) {
                const int tt = (e * 64 + lane) * 4;
                if (tt < L) {
                    const f4a* v4 = reinterpret_cast<const f4a*>(sm + (size_t)tt * K);
                    float v[4 * K];
#pragma unroll
                    for (int h = 0; h < K; ++h) { const f4a x = v4[h]; v[4 * h] = x[0]; v[4 * h + 1] = x[1]; v[4 * h + 2] = x[2]; v[4 * h + 3] = x[3]; }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < K; ++q) acc[q] = fmaf(a[slot][c], v[c * K + q], acc[q]);
                }
            }
            if (++e == lpr) {
                if (row < wrows) {
#pragma unroll
                    for (int q = 0; q < K; ++q) {
                        const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum63(acc[q])), 63));
                        if (lane == row) mine[q] = tot;
                        acc[q] = 0.f;
                    }
                }
                e = 0; ++row;
            }
        }
    }
    if (lane < wrows && jw + lane < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) out[((size_t)node * S + jw + lane) * K + q] = mine[q];
    }
}

// streaming floor: read the same bytes with nothing else
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ m, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = m[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    struct Shape { const char* name; int n, S, B; } shapes[] = {{"level4", 256, 128, 256}, {"level3", 64, 252, 504}, {"level2", 16, 504, 1004}, {"level1", 4, 1000, 1004}, {"level0", 1, 2000, 0}};
    const int COPIES = 8, REPS = 100;
    for (const Shape& sh : shapes) {
        const int L = sh.S + sh.B, lpr = (L + 255) / 256;
        const size_t mat_n = (size_t)sh.n * sh.S * L, vec_n = (size_t)sh.n * L;
        float *mat, *vecs, *out; int* index; 
        CK(hipMalloc(&mat, (mat_n * COPIES + 64) * 4)); CK(hipMalloc(&vecs, vec_n * K * 4)); CK(hipMalloc(&out, (size_t)sh.n * sh.S * K * 4)); CK(hipMalloc(&index, vec_n * 4));
        CK(hipMemset(mat, 0, (mat_n * COPIES + 64) * 4)); CK(hipMemset(vecs, 0, vec_n * K * 4));
        std::vector<int> idx(vec_n); for (size_t i = 0; i < vec_n; ++i) idx[i] = (int)((i * 7919) % vec_n);
        CK(hipMemcpy(index, idx.data(), vec_n * 4, hipMemcpyHostToDevice));
        printf("%s: %d nodes, S %d, B %d, %.1f MB per launch\n", sh.name, sh.n, sh.S, sh.B, mat_n * 4e-6);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        {
            for (int grid : {1024, 2048}) {
                for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const float4*)(mat + (r % COPIES) * mat_n), mat_n / 4, out);
                CK(hipEventRecord(e0));
                for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const float4*)(mat + (r % COPIES) * mat_n), mat_n / 4, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("   stream grid %4d: %6.2f us  %5.0f GB/s\n", grid, ms / REPS * 1e3, mat_n * 4 / (ms / REPS * 1e-3) * 1e-9);
            }
        }
        for (int R = 1; R * lpr <= CAP; R *= 2) {
            const int tpn = (sh.S + NW * R - 1) / (NW * R), tiles = sh.n * tpn;
            std::vector<Rec> recs(tiles);
            for (int t = 0; t < tiles; ++t) { Rec& r = recs[t]; r.node = t / tpn; r.row0 = (t % tpn) * NW * R; r.s = sh.S; r.b = sh.B; r.mat_off = (long long)r.node * sh.S * L; r.vec_off = (long long)r.node * L; }
            Rec* drec; CK(hipMalloc(&drec, sizeof(Rec) * tiles)); CK(hipMemcpy(drec, recs.data(), sizeof(Rec) * tiles, hipMemcpyHostToDevice));
            const size_t lds = ((size_t)L + 8) * K * 4;
            for (int variant = 0; variant < 3; ++variant) {
                auto launch = [&](int r) {
                    const float* m = mat + (size_t)(r % COPIES) * mat_n;
                    if (variant == 0) hipLaunchKernelGGL(k_level<0>, dim3(tiles), dim3(64 * NW), lds, 0, drec, m, vecs, index, out, sh.S, sh.B, R, tpn);
                    else if (variant == 1) hipLaunchKernelGGL(k_level<1>, dim3(tiles), dim3(64 * NW), lds, 0, drec, m, vecs, index, out, sh.S, sh.B, R, tpn);
                    else hipLaunchKernelGGL(k_level<2>, dim3(tiles), dim3(64 * NW), lds, 0, drec, m, vecs, index, out, sh.S, sh.B, R, tpn);
                };
                for (int r = 0; r < 5; ++r) launch(r);
                CK(hipEventRecord(e0));
                for (int r = 0; r < REPS; ++r) launch(r);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("   R %2d tiles %5d variant %d: %6.2f us per launch (back to back)  %5.0f GB/s\n", R, tiles, variant, ms / REPS * 1e3, mat_n * 4 / (ms / REPS * 1e-3) * 1e-9);
            }
            CK(hipFree(drec));
        }
        CK(hipFree(mat)); CK(hipFree(vecs)); CK(hipFree(out)); CK(hipFree(index));
    }
    return 0;
}
