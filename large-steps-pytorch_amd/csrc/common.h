// common.h -- shared host/device helpers for liblargesteps_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/largesteps_hip.h"

namespace ls {

constexpr int WAVE = 64;           // gfx950 wavefront
constexpr int BLOCK = 256;         // 4 waves, one per SIMD
constexpr int TILE_ROWS = 256;     // rows handled by one block pass (one row per thread)
constexpr int MAX_GRID = 2048;     // upper bound of reduction partials per kernel (8 blocks per CU on 256 CUs)

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define LS_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return ::ls::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define LS_REQUIRE(cond, code, ...)                                         \
    do {                                                                    \
        if (!(cond)) { ::ls::set_error(__VA_ARGS__); return (code); }       \
    } while (0)

struct DeviceGuard {  // every entry point pins the device itself: no thread-local state is assumed
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device) {
        // The process is shared with PyTorch (and with this library's own fire-and-forget frees): drop any stale
        // sticky error so that this entry point only reports failures of its own calls. LS_DEBUG=1 prints it.
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("LS_DEBUG"))
            fprintf(stderr, "[largesteps] stale HIP error %d (%s) found on entry\n", (int)stale, hipGetErrorString(stale));
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// direct.hip: a per-device pool of device buffers (>= 256 KB) that the direct solver's constructor and destructor hand back instead of
// hipFree -- see ls_release_scratch in the header. (Round 4: the threshold was 64 MB; a 1M-vertex construction also makes ~40 allocations of
// 1-30 MB -- index lists, tables, vectors -- whose hipMalloc / hipFree pairs cost the remesh loop milliseconds.) take: a pooled buffer of at least `bytes` (and not much more; *capacity = its real size,
// to be handed back to give) or nullptr;
// give: false when the pool did not take the buffer (the caller frees it). The buffer must be idle (its stream synchronised).
void* pool_take(int device, size_t bytes, size_t* capacity);
bool pool_give(int device, void* p, size_t bytes);
hipError_t pool_alloc(int device, void** p, size_t bytes);      // hipMalloc; out of memory: the pool of `device` is emptied and the call repeated once
constexpr size_t POOL_FROM = (size_t)256 << 10, POOL_SLACK = (size_t)1 << 20;      // smallest pooled buffer; a taken buffer is at most 1.5 x the request + POOL_SLACK
// assemble.hip: out[0 .. n] = exclusive scan of the int32 in[0 .. n), out[n] = total; bsum: scan_blocks(n) + 1 ints of scratch
int exclusive_scan(const int* in, int64_t n, int* out, int* bsum, hipStream_t st);
inline int64_t scan_blocks(int64_t n) { return (n + 2047) / 2048; }

// ---- device side ---------------------------------------------------------------------------------
#ifdef __HIPCC__

template <int K> struct Vec { float v[K]; };   // K interleaved right-hand-side columns of one vertex (4-byte aligned)

// Read-once factor streams. A solve reads every word of the dense factor blocks exactly once per sweep, while the vectors, slot
// records and index lists next to them are re-read by neighbouring tiles and by the next launch. When the factor does not fit the
// 256 MB Infinity Cache anyway, its loads carry the non-temporal policy (global_load ... nt: the line is not kept behind the read) and
// stop evicting what IS re-used: 1M vertices 214 -> 199 us per solve, 4M 722 -> 705 (profiles/r05_nt_policy.txt). Below that size the
// whole factor stays cache resident from solve to solve and nt costs 5-15 %: the policy is a template parameter the handle picks
// (ls_direct_create). The leaves' packed triangles never carry it: both sweeps read the same triangle.
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
typedef float ls_f4v __attribute__((ext_vector_type(4)));
typedef float ls_f2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float2 ld_stream2(const float2* p) {
    if constexpr (NT) {
        const ls_f2v v = __builtin_nontemporal_load(reinterpret_cast<const ls_f2v*>(p));
        return make_float2(v[0], v[1]);
    } else return *p;
}
template <bool NT>
__device__ __forceinline__ float4 ld_stream4(const float4* p) {
    if constexpr (NT) {
        const ls_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ls_f4v*>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else return *p;
}

// ---- DPP wave reductions ---------------------------------------------------------------------------
// __shfl_down lowers to ds_bpermute (an LDS-crossbar round trip, ~100+ cycles per dependent step); a
// 6-step fp64 butterfly per value made the dot-product hand-off of the PCG kernels cost ~4 us per launch
// (profiles/r01_ubench1_kernel_structure.txt). The DPP row operations below stay inside the VALU.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);   // lanes without a source (masked rows) add +0.0
}

// Sum over the 64 lanes of a wave, returned in EVERY lane (wave-uniform).
__device__ __forceinline__ double wave_sum_all(double v) {
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]   -> every lane: sum of its quad
    v = dpp_add<0x141, 0xf>(v);   // row_half_mirror       -> sum of its 8 lanes
    v = dpp_add<0x140, 0xf>(v);   // row_mirror            -> sum of its row of 16
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
    return x;   // valid in lane 0
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) x = fmaxf(x, __shfl_down(x, off, WAVE));
    return x;
}

// Block sum of N doubles per thread. Result valid in thread 0. `smem` holds (BLOCK/WAVE)*N doubles.
template <int N>
__device__ __forceinline__ void block_sum(double (&x)[N], double* smem) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i] = wave_sum(x[i]);
        if (lane == 0) smem[w * N + i] = x[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
            for (int j = 0; j < BLOCK / WAVE; ++j) s += smem[j * N + i];
            x[i] = s;
        }
    }
    __syncthreads();
}

// Deterministic reduction of a partial array part[n*G + g], g < G (written by the G blocks of the
// previous kernel) into out[n], n < N, broadcast to every thread of the block through smem.
// smem: N + (BLOCK/WAVE)*N doubles.
template <int N>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ part, int G, double (&out)[N], double* smem) {
    double acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        for (int g = threadIdx.x; g < G; g += BLOCK) s += part[(size_t)i * MAX_GRID + g];
        acc[i] = s;
    }
    block_sum<N>(acc, smem + N);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) smem[i] = acc[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = smem[i];
    __syncthreads();
}

// XCD-aware tile schedule. Workgroup b runs on XCD b % 8 (observed placement, used for L2 locality
// only, never for correctness). The T row tiles are cut into 8 contiguous ranges, one per XCD, and
// the G/8 workgroups of an XCD stride through their range, so neighbouring rows (and the vector
// entries they gather) stay in one XCD's L2 and the same rows meet the same L2 in every kernel.
struct TileSched {
    int first, step, end;
    __device__ __forceinline__ TileSched(int T, int G) {
        if (G >= 8 && (G & 7) == 0) {
            const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per = (T + 7) >> 3;
            const int lo = xcd * per;
            end = min(T, lo + per);
            first = lo + local;
            step = G >> 3;
        } else {
            first = blockIdx.x; step = G; end = T;
        }
    }
};

#endif  // __HIPCC__

}  // namespace ls
