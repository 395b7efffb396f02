// Micro-benchmark (NOT part of the product): the upper tree levels of a sweep as ONE persistent launch with a grid barrier per
// level and the next level's matrix rows requested BEFORE the barrier, against one launch per level.
// Synthetic 5-level down sweep of the 1M plane: level l has n nodes of S x (S + B) fp32; the vector of a level is made of
// the previous level's results (so the barrier carries real data).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/level_persist.hip -o tools/ubench/build/level_persist
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef CAPV
#define CAPV 32
#endif
constexpr int K = 3, CAP = CAPV, NW = 4, MAXL = 8;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
struct Lv { int n, S, B, R, tpn, tiles, prev_n, pad; long long mat_off, out_off, prev_off; };
struct Args { Lv lv[MAXL]; int nlv, grid; };

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
__device__ __forceinline__ void load_rows(const float* __restrict__ base, int L, int lpr, int nrows, f4u (&a)[CAP]) {
    const int lane = threadIdx.x & 63;
    int row = 0, e = 0;
#pragma unroll
    for (int slot = 0; slot < CAP; ++slot) {
        const int tt = (e * 64 + lane) * 4;
        f4u z = {0.f, 0.f, 0.f, 0.f};
        a[slot] = (row < nrows && tt < L) ? *reinterpret_cast<const f4u*>(base + (size_t)row * L + tt) : z;
        if (++e == lpr) { e = 0; ++row; }
    }
}
__device__ __forceinline__ void fma_rows(const f4u (&a)[CAP], int L, int lpr, int nrows, const float* __restrict__ sm, float (&mine)[K]) {
    const int lane = threadIdx.x & 63;
    float acc[K] = {0.f, 0.f, 0.f};
    int row = 0, e = 0;
#pragma unroll
    for (int slot = 0; slot < CAP; ++slot) {
        if (row < nrows) {
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
            if (row < nrows) {
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
// vector of node `node` of level v: entry u = previous level's result number (node * L + u) mod (prev_n) (level 0: a constant)
template <bool COHERENT>
__device__ __forceinline__ void stage_vec(const Lv& v, int node, float* outs, float* sm) {
    const int L = v.S + v.B;
    for (int u = threadIdx.x; u < L; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            float* src = outs + v.prev_off + ((size_t)((long long)node * L + u) % (v.prev_n ? v.prev_n : 1)) * K + q;
            sm[u * K + q] = !v.prev_n ? 1.0f : COHERENT ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
        }
    }
}
// no cache maintenance: the data that crosses the barrier is written and read with agent-scope (L2-coherent) accesses
__device__ __forceinline__ void grid_barrier_relaxed(unsigned* bar, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
        if (spins >= 2000000) bar[1] = 1u;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
        if (spins >= 2000000) bar[1] = 1u;     // timed out: not all workgroups are resident
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // every thread reads the other workgroups' results afterwards
}

// MODE 0: one level per launch (level index `only`); 1: persistent, loads after the barrier; 2: persistent, loads before the barrier
template <int MODE>
__global__ __launch_bounds__(64 * NW) void k_levels(Args A, const float* __restrict__ mat, float* outs, unsigned* bar, unsigned bar_base, int only) {
    static_assert(MODE >= 0 && MODE <= 3, "mode");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4u a[CAP];
    const int l_lo = MODE == 0 ? only : 0, l_hi = MODE == 0 ? only + 1 : A.nlv;
    bool have = false;
    for (int l = l_lo; l < l_hi; ++l) {
        const Lv v = A.lv[l];
        const int L = v.S + v.B, lpr = (L + 255) >> 8;
        for (int t = blockIdx.x; t < v.tiles; t += gridDim.x) {
            const int node = t / v.tpn, jw = (t % v.tpn) * (NW * v.R) + w * v.R;
            const int wrows = max(0, min(v.R, v.S - jw));
            if (!have) load_rows(mat + v.mat_off + ((size_t)node * v.S + jw) * L, L, lpr, wrows, a);
            have = false;
            stage_vec<MODE == 3>(v, node, outs, sm);
            __syncthreads();
            float mine[K] = {0.f, 0.f, 0.f};
            fma_rows(a, L, lpr, wrows, sm, mine);
            if (lane < wrows) {
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    float* dst = outs + v.out_off + ((size_t)node * v.S + jw + lane) * K + q;
                    if (MODE == 3) __hip_atomic_store(dst, mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = mine[q];
                }
            }
            __syncthreads();
        }
        if (MODE != 0 && l + 1 < l_hi) {
            if (MODE >= 2) {
                const Lv nx = A.lv[l + 1];
                const int t = blockIdx.x;
                if (t < nx.tiles) {
                    const int Ln = nx.S + nx.B, node = t / nx.tpn, jw = (t % nx.tpn) * (NW * nx.R) + w * nx.R;
                    load_rows(mat + nx.mat_off + ((size_t)node * nx.S + jw) * Ln, Ln, (Ln + 255) >> 8, max(0, min(nx.R, nx.S - jw)), a);
                    have = true;
                }
            }
            if (MODE == 3) grid_barrier_relaxed(bar, bar_base + (unsigned)(l + 1) * gridDim.x); else grid_barrier(bar, bar_base + (unsigned)(l + 1) * gridDim.x);
        }
    }
}

// MODE 4: ONE launch for all levels, a workgroup per tile in LEVEL ORDER (the dispatcher hands out workgroup ids in order, so
// every tile of a level is dispatched before any tile of the next one -- the forward-progress assumption of a decoupled
// look-back scan); a tile requests its matrix rows at once, then waits for ITS PARENT NODE only (a counter of finished
// tiles per node, agent-scope release / acquire), multiplies, stores, counts itself in. No device-wide barrier.
struct FlowArgs { Lv lv[MAXL]; int first[MAXL + 1]; int nlv, fan; };      // first[l]: first workgroup id of level l; fan: children per node
// COH: 0 = acquire / release, 1 = relaxed counters + per-element coherent (atomic) accesses of the hand-over vectors, 2 = relaxed counters +
// write-through stores / sc1 loads through raw buffer instructions (the recipe of csrc/nd_span.h, round 3)
constexpr int FLOW_SC1 = 16 | (int)0x80000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t flow_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
template <int COH>
__global__ __launch_bounds__(64 * NW) void k_flow(FlowArgs A, const float* __restrict__ mat, float* outs, unsigned* done, unsigned epoch) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int l = 0;
    while (l + 1 < A.nlv && (int)blockIdx.x >= A.first[l + 1]) ++l;
    const Lv v = A.lv[l];
    const int t = blockIdx.x - A.first[l];
    const int L = v.S + v.B, lpr = (L + 255) >> 8;
    const int node = t / v.tpn, jw = (t % v.tpn) * (NW * v.R) + w * v.R;
    const int wrows = max(0, min(v.R, v.S - jw));
    f4u a[CAP];
    load_rows(mat + v.mat_off + ((size_t)node * v.S + jw) * L, L, lpr, wrows, a);
    if (l > 0) {
        const Lv pv = A.lv[l - 1];
        const int parent = node / A.fan;
        if (threadIdx.x == 0) {
            unsigned* c = done + (size_t)(l - 1) * 4096 + parent;
            const unsigned target = epoch * (unsigned)pv.tpn;
            int spins = 0;
            while (__hip_atomic_load(c, COH ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 4000000) __builtin_amdgcn_s_sleep(COH == 2 ? 1 : 2);
            if (spins >= 4000000) done[MAXL * 4096] = 1u;
        }
        __syncthreads();
        if (!COH) __atomic_thread_fence(__ATOMIC_ACQUIRE); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // vector = the parent's results, repeated
        for (int u = threadIdx.x; u < L; u += blockDim.x) {
#pragma unroll
            for (int q = 0; q < K; ++q) {
                float* src = outs + pv.out_off + ((size_t)parent * pv.S + (u % pv.S)) * K + q;
                if (COH == 2) sm[u * K + q] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(flow_rsrc(outs), (int)((src - outs) * 4), 0, FLOW_SC1));
                else sm[u * K + q] = COH ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
            }
        }
    } else {
        for (int u = threadIdx.x; u < L; u += blockDim.x) {
#pragma unroll
            for (int q = 0; q < K; ++q) sm[u * K + q] = 1.0f;
        }
    }
    __syncthreads();
    float mine[K] = {0.f, 0.f, 0.f};
    fma_rows(a, L, lpr, wrows, sm, mine);
    if (lane < wrows) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            float* dst = outs + v.out_off + ((size_t)node * v.S + jw + lane) * K + q;
            if (COH == 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(mine[q]), flow_rsrc(outs), (int)((dst - outs) * 4), 0, FLOW_SC1);
            else if (COH) __hip_atomic_store(dst, mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = mine[q];
        }
    }
    if (l + 1 < A.nlv) {                       // the last level has no dependants
        if (COH) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done + (size_t)l * 4096 + node, 1u, COH ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the same data flow with one launch per level (reference for k_flow: the vector is the parent's results)
__global__ __launch_bounds__(64 * NW) void k_flow_level(FlowArgs A, const float* __restrict__ mat, float* outs, int l) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const Lv v = A.lv[l];
    const int t = blockIdx.x;
    const int L = v.S + v.B, lpr = (L + 255) >> 8;
    const int node = t / v.tpn, jw = (t % v.tpn) * (NW * v.R) + w * v.R;
    const int wrows = max(0, min(v.R, v.S - jw));
    f4u a[CAP];
    load_rows(mat + v.mat_off + ((size_t)node * v.S + jw) * L, L, lpr, wrows, a);
    for (int u = threadIdx.x; u < L; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sm[u * K + q] = l ? outs[A.lv[l - 1].out_off + ((size_t)(node / A.fan) * A.lv[l - 1].S + (u % A.lv[l - 1].S)) * K + q] : 1.0f;
    }
    __syncthreads();
    float mine[K] = {0.f, 0.f, 0.f};
    fma_rows(a, L, lpr, wrows, sm, mine);
    if (lane < wrows) {
#pragma unroll
        for (int q = 0; q < K; ++q) outs[v.out_off + ((size_t)node * v.S + jw + lane) * K + q] = mine[q];
    }
}

// the same level kernel for the nodes [n0, n0 + nn) of a level only: one of the independent subtrees below the root
__global__ __launch_bounds__(64 * NW) void k_flow_part(FlowArgs A, const float* __restrict__ mat, float* outs, int l, int n0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const Lv v = A.lv[l];
    const int t = blockIdx.x;
    const int L = v.S + v.B, lpr = (L + 255) >> 8;
    const int node = n0 + t / v.tpn, jw = (t % v.tpn) * (NW * v.R) + w * v.R;
    const int wrows = max(0, min(v.R, v.S - jw));
    f4u a[CAP];
    load_rows(mat + v.mat_off + ((size_t)node * v.S + jw) * L, L, lpr, wrows, a);
    for (int u = threadIdx.x; u < L; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sm[u * K + q] = l ? outs[A.lv[l - 1].out_off + ((size_t)(node / A.fan) * A.lv[l - 1].S + (u % A.lv[l - 1].S)) * K + q] : 1.0f;
    }
    __syncthreads();
    float mine[K] = {0.f, 0.f, 0.f};
    fma_rows(a, L, lpr, wrows, sm, mine);
    if (lane < wrows) {
#pragma unroll
        for (int q = 0; q < K; ++q) outs[v.out_off + ((size_t)node * v.S + jw + lane) * K + q] = mine[q];
    }
}

int main(int argc, char** argv) {
    int occ1 = 0, occ2 = 0, cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, k_levels<1>, 64 * NW, 40 * 1024));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, k_levels<3>, 64 * NW, 40 * 1024));
    const int want = argc > 1 ? atoi(argv[1]) : 2;
    const int per_cu = std::min(want, std::min(occ1, occ2));
    const int G = per_cu * cus;
    printf("CAP %d: %d CUs, occupancy %d / %d workgroups per CU -> persistent grid %d\n", CAP, cus, occ1, occ2, G);
    if (G <= 0) return 1;
    struct Shape { int n, S, B; } shapes[] = {{1, 2000, 0}, {4, 1000, 1004}, {16, 504, 1004}, {64, 252, 504}, {256, 128, 256}};
    const int NL = 5, COPIES = 6, REPS = 100;
    Args A; A.nlv = NL; A.grid = G;
    size_t mat_n = 0, out_n = 0;
    size_t lds = 0;
    for (int l = 0; l < NL; ++l) {
        Lv& v = A.lv[l];
        v.n = shapes[l].n; v.S = shapes[l].S; v.B = shapes[l].B;
        const int L = v.S + v.B, lpr = (L + 255) / 256;
        v.R = CAP / lpr;
        while (v.R > 1 && (long long)v.n * ((v.S + NW * (v.R - 1) - 1) / (NW * (v.R - 1))) <= G) --v.R;      // smallest R that still fits one tile per workgroup
        v.tpn = (v.S + NW * v.R - 1) / (NW * v.R); v.tiles = v.n * v.tpn;
        v.mat_off = (long long)mat_n; mat_n += (size_t)v.n * v.S * L;
        v.out_off = (long long)out_n; out_n += (size_t)v.n * v.S;
        v.prev_n = l ? A.lv[l - 1].n * A.lv[l - 1].S : 0; v.prev_off = l ? A.lv[l - 1].out_off * K : 0;
        v.out_off *= K;
        lds = std::max(lds, ((size_t)L + 8) * K * 4);
        printf("level %d: n %d S %d B %d  R %d tiles %d  %.1f MB\n", l, v.n, v.S, v.B, v.R, v.tiles, (size_t)v.n * v.S * L * 4e-6);
    }
    float *mat, *outs; unsigned* bar;
    CK(hipMalloc(&mat, (mat_n * COPIES + 64) * 4)); CK(hipMalloc(&outs, out_n * K * 4)); CK(hipMalloc(&bar, 8));
    {   // small values: products stay finite over the levels
        std::vector<float> h(mat_n);
        for (size_t i = 0; i < mat_n; ++i) h[i] = 1e-3f * (float)((i * 2654435761u) % 1000) / 1000.0f;
        for (int c = 0; c < COPIES; ++c) CK(hipMemcpy(mat + (size_t)c * mat_n, h.data(), mat_n * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(outs, 0, out_n * K * 4)); CK(hipMemset(bar, 0, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref(out_n * K), got(out_n * K);
    unsigned epoch = 0;
    for (int mode = 0; mode < 4; ++mode) {
        auto run = [&](int r) {
            const float* m = mat + (size_t)(r % COPIES) * mat_n;
            if (mode == 0) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(k_levels<0>, dim3(A.lv[l].tiles), dim3(64 * NW), lds, 0, A, m, outs, bar, 0u, l); }
            else {
                if (mode == 1) hipLaunchKernelGGL(k_levels<1>, dim3(G), dim3(64 * NW), lds, 0, A, m, outs, bar, epoch, 0);
                else if (mode == 3) hipLaunchKernelGGL(k_levels<3>, dim3(G), dim3(64 * NW), lds, 0, A, m, outs, bar, epoch, 0);
                else hipLaunchKernelGGL(k_levels<2>, dim3(G), dim3(64 * NW), lds, 0, A, m, outs, bar, epoch, 0);
                epoch += (unsigned)(NL - 1) * G;
            }
        };
        for (int r = 0; r < 5; ++r) run(r);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < REPS; ++r) run(r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy((mode == 0 ? ref : got).data(), outs, out_n * K * 4, hipMemcpyDeviceToHost));
        double diff = 0, mx = 0;
        if (mode) for (size_t i = 0; i < ref.size(); ++i) { diff = std::max(diff, (double)fabsf(ref[i] - got[i])); mx = std::max(mx, (double)fabsf(ref[i])); }
        unsigned hb[2]; CK(hipMemcpy(hb, bar, 8, hipMemcpyDeviceToHost));
        if (hb[1]) printf("  BARRIER TIMED OUT\n");
        printf("mode %d (%s): %7.2f us per sweep of %d levels, %.0f GB/s   max diff vs mode 0: %.3g (max |ref| %.3g)\n", mode,
               mode == 0 ? "one launch per level" : mode == 1 ? "persistent, loads after the barrier" : mode == 2 ? "persistent, loads before the barrier" : "persistent, relaxed barrier + coherent vector accesses, loads before the barrier",
               ms / REPS * 1e3, NL, mat_n * 4 / (ms / REPS * 1e-3) * 1e-9, diff, mx);
    }
    {   // ---- data flow in one launch (mode 4) against the same data flow with a launch per level
        FlowArgs F; F.nlv = NL; F.fan = 4;
        int first = 0;
        for (int l = 0; l < NL; ++l) {
            F.lv[l] = A.lv[l];
            // tiles small enough that a level has at least ~500 of them
            Lv& v = F.lv[l];
            const int L = v.S + v.B, lpr = (L + 255) / 256;
            v.R = std::max(1, std::min(CAP / lpr, v.S / NW));
            while (v.R > 1 && (long long)v.n * ((v.S + NW * v.R - 1) / (NW * v.R)) < 500) --v.R;
            v.tpn = (v.S + NW * v.R - 1) / (NW * v.R); v.tiles = v.n * v.tpn;
            F.first[l] = first; first += v.tiles;
            printf("flow level %d: R %d tiles %d (%d per node)\n", l, v.R, v.tiles, v.tpn);
        }
        F.first[NL] = first;
        unsigned* done; CK(hipMalloc(&done, (MAXL * 4096 + 4) * 4)); CK(hipMemset(done, 0, (MAXL * 4096 + 4) * 4));
        unsigned ep = 0;
        for (int mode = 0; mode < 4; ++mode) {
            auto run = [&](int r) {
                const float* m = mat + (size_t)(r % COPIES) * mat_n;
                if (mode == 0) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(k_flow_level, dim3(F.lv[l].tiles), dim3(64 * NW), lds, 0, F, m, outs, l); }
                else if (mode == 1) { ++ep; hipLaunchKernelGGL(k_flow<0>, dim3(first), dim3(64 * NW), lds, 0, F, m, outs, done, ep); }
                else if (mode == 2) { ++ep; hipLaunchKernelGGL(k_flow<1>, dim3(first), dim3(64 * NW), lds, 0, F, m, outs, done, ep); }
                else { ++ep; hipLaunchKernelGGL(k_flow<2>, dim3(first), dim3(64 * NW), lds, 0, F, m, outs, done, ep); }
            };
            for (int r = 0; r < 5; ++r) run(r);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < REPS; ++r) run(r);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy((mode == 0 ? ref : got).data(), outs, out_n * K * 4, hipMemcpyDeviceToHost));
            double diff = 0, mx = 0;
            if (mode) for (size_t i = 0; i < ref.size(); ++i) { diff = std::max(diff, (double)fabsf(ref[i] - got[i])); mx = std::max(mx, (double)fabsf(ref[i])); }
            unsigned flag = 0; CK(hipMemcpy(&flag, done + MAXL * 4096, 4, hipMemcpyDeviceToHost));
            if (flag) printf("  A WAIT TIMED OUT\n");
            printf("flow mode %d (%s): %7.2f us per sweep of %d levels, %.0f GB/s   max diff %.3g (max |ref| %.3g)\n", mode,
                   mode == 0 ? "one launch per level" : mode == 1 ? "ONE launch, tiles wait for their parent node (acquire / release)" : mode == 2 ? "ONE launch, tiles wait for their parent node (relaxed counters, L2-coherent vector accesses)" : "ONE launch, tiles wait for their parent node (relaxed counters, write-through stores / sc1 buffer loads)", ms / REPS * 1e3, NL, mat_n * 4 / (ms / REPS * 1e-3) * 1e-9, diff, mx);
        }
    }
    {   // ---- the four subtrees below the root as four concurrent chains of launches (streams), replayed as ONE graph
        FlowArgs F; F.nlv = NL; F.fan = 4;
        for (int l = 0; l < NL; ++l) {
            F.lv[l] = A.lv[l];
            Lv& v = F.lv[l];
            const int L = v.S + v.B, lpr = (L + 255) / 256;
            v.R = std::max(1, std::min(CAP / lpr, v.S / NW));
            while (v.R > 1 && (long long)v.n * ((v.S + NW * v.R - 1) / (NW * v.R)) < 500) --v.R;
            v.tpn = (v.S + NW * v.R - 1) / (NW * v.R); v.tiles = v.n * v.tpn;
        }
        hipStream_t main_s, sub[4];
        CK(hipStreamCreate(&main_s));
        for (int q = 0; q < 4; ++q) CK(hipStreamCreate(&sub[q]));
        hipEvent_t fork, join[4];
        CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        for (int q = 0; q < 4; ++q) CK(hipEventCreateWithFlags(&join[q], hipEventDisableTiming));
        const float* m = mat;
        for (int variant = 0; variant < 2; ++variant) {
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeGlobal));
            hipLaunchKernelGGL(k_flow_level, dim3(F.lv[0].tiles), dim3(64 * NW), lds, main_s, F, m, outs, 0);
            if (variant == 0) {
                for (int l = 1; l < NL; ++l) hipLaunchKernelGGL(k_flow_level, dim3(F.lv[l].tiles), dim3(64 * NW), lds, main_s, F, m, outs, l);
            } else {
                CK(hipEventRecord(fork, main_s));
                for (int q = 0; q < 4; ++q) {
                    CK(hipStreamWaitEvent(sub[q], fork, 0));
                    for (int l = 1; l < NL; ++l) {
                        const int nn = F.lv[l].n / 4;
                        hipLaunchKernelGGL(k_flow_part, dim3(nn * F.lv[l].tpn), dim3(64 * NW), lds, sub[q], F, m, outs, l, q * nn);
                    }
                    CK(hipEventRecord(join[q], sub[q]));
                    CK(hipStreamWaitEvent(main_s, join[q], 0));
                }
            }
            CK(hipStreamEndCapture(main_s, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, main_s));
            CK(hipStreamSynchronize(main_s));
            CK(hipEventRecord(e0, main_s));
            for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(exec, main_s));
            CK(hipEventRecord(e1, main_s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("graph variant %d (%s): %7.2f us per sweep of %d levels (same matrix copy every time)\n", variant,
                   variant == 0 ? "one chain of 5 launches" : "root, then 4 concurrent chains of 4 launches (one per subtree)", ms / REPS * 1e3, NL);
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
        }
    }
    return 0;
}
