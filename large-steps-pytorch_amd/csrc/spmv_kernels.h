// spmv_kernels.h -- device-side row products shared by spmv.hip (to_differential) and pcg.hip.
//
// Three storage/access schemes for y_i = sum_j M_ij x_j with K interleaved right-hand sides:
//   * CsrDirect : one thread per row walks col/val straight from global memory (uncoalesced matrix
//                 reads, kept as the A/B baseline).
//   * CsrLds    : the nnz range of a 256-row tile is contiguous in CSR; the workgroup copies it with
//                 fully coalesced loads into LDS as {col,val} pairs, then every thread walks its row
//                 from LDS with ds_read_b64 (lane stride = row length ~7 pairs: odd, conflict free).
//   * Sell64    : sliced ELLPACK with slice height 64 = one wavefront; entry t of the 64 rows of a
//                 slice is stored contiguously, so every matrix load is one coalesced 512-byte
//                 {col,val} wave access and no LDS / barrier is needed. Padding entries carry val=0.
// The gathers x[col] are 4*K-byte loads (global_load_dwordx3 for K=3); on banded meshes consecutive
// lanes hit consecutive vertices, so they coalesce as well.
#pragma once
#include "common.h"

namespace ls {

constexpr int LDS_CAP = 2560;   // {col,val} pairs staged per tile (20 KiB: eight workgroups per CU; a 256-row tile of a valence-6 mesh has ~1800); denser tiles fall back to direct reads

struct CsrView {
    const int* __restrict__ rowptr;
    const int* __restrict__ col;
    const float* __restrict__ val;
};

struct SellView {
    const int* __restrict__ slice_ptr;   // [S+1], in entries (multiples of 64)
    const int2* __restrict__ cv;         // {col, float bits}
};

template <int K>
__device__ __forceinline__ void fma_row(float (&acc)[K], float v, const float* __restrict__ x, int c) {
    const Vec<K> t = reinterpret_cast<const Vec<K>*>(x)[c];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = fmaf(v, t.v[q], acc[q]);
}

template <int K>
__device__ __forceinline__ void row_csr_direct(const CsrView& A, const float* __restrict__ x, int64_t i, float (&acc)[K]) {
    const int s = A.rowptr[i], e = A.rowptr[i + 1];
    for (int j = s; j < e; ++j) fma_row<K>(acc, A.val[j], x, A.col[j]);
}

// Workgroup-cooperative: every thread of the block must call this (barriers inside).
// r0/r1: tile row range; `active` = this thread owns row i = r0 + threadIdx.x < r1.
// Round 5: (1) the tile's col / val ranges are copied with 16-byte loads (4 entries per lane and request; the ranges start at any
// 4-byte offset, which global_load_dwordx4 accepts) instead of one 4-byte load per entry and array; (2) a row's gathers x[col] are all
// requested before the first product -- the loop `LDS read -> gather -> fma` had run a row's ~7 entries as 7 dependent round trips.
// The products stay in row order: the result is the same bit for bit.
typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4u_spmv __attribute__((ext_vector_type(4), aligned(4)));
template <int K>
__device__ __forceinline__ void row_csr_lds(const CsrView& A, const float* __restrict__ x, int64_t r0, int64_t r1,
                                            int2* __restrict__ s_cv, float (&acc)[K]) {
    const int64_t i = r0 + threadIdx.x;
    const bool active = i < r1;
    const int base = A.rowptr[r0];
    const int total = A.rowptr[r1] - base;
    int s = 0, e = 0;
    if (active) { s = A.rowptr[i]; e = A.rowptr[i + 1]; }
    if (total <= LDS_CAP) {
        for (int t = 4 * threadIdx.x; t < total; t += 4 * BLOCK) {
            if (t + 4 <= total) {
                const i4u c = *reinterpret_cast<const i4u*>(A.col + base + t);
                const f4u_spmv v = *reinterpret_cast<const f4u_spmv*>(A.val + base + t);
                int4* o = reinterpret_cast<int4*>(s_cv + t);           // (t is a multiple of 4: 32-byte aligned in LDS)
                o[0] = make_int4(c[0], __float_as_int(v[0]), c[1], __float_as_int(v[1]));
                o[1] = make_int4(c[2], __float_as_int(v[2]), c[3], __float_as_int(v[3]));
            } else {
                for (int u = t; u < total; ++u) s_cv[u] = make_int2(A.col[base + u], __float_as_int(A.val[base + u]));
            }
        }
        __syncthreads();
        const int2* __restrict__ p = s_cv + (s - base);
        const int n = e - s;
        constexpr int B = 8;                                            // gathers in flight per lane (valence + 1 <= 8 on a regular mesh)
        for (int j0 = 0; j0 < n; j0 += B) {
            int2 c[B];
            Vec<K> xv[B];
#pragma unroll
            for (int t = 0; t < B; ++t) c[t] = j0 + t < n ? p[j0 + t] : make_int2(0, 0);
#pragma unroll
            for (int t = 0; t < B; ++t) if (j0 + t < n) xv[t] = reinterpret_cast<const Vec<K>*>(x)[c[t].x];
#pragma unroll
            for (int t = 0; t < B; ++t) if (j0 + t < n) {
#pragma unroll
                for (int q = 0; q < K; ++q) acc[q] = fmaf(__int_as_float(c[t].y), xv[t].v[q], acc[q]);
            }
        }
        __syncthreads();   // the next tile overwrites s_cv
    } else {
        for (int j = s; j < e; ++j) fma_row<K>(acc, A.val[j], x, A.col[j]);
    }
}

template <int K>
__device__ __forceinline__ void row_sell(const SellView& A, const float* __restrict__ x, int64_t i, float (&acc)[K]) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int slice = __builtin_amdgcn_readfirstlane((int)(i >> 6));      // wave uniform
    const int off = __builtin_amdgcn_readfirstlane(A.slice_ptr[slice]);
    const int width = (__builtin_amdgcn_readfirstlane(A.slice_ptr[slice + 1]) - off) >> 6;
    const int2* __restrict__ p = A.cv + off + lane;
    if (width <= 8) {
        // the usual case (valence <= 7): every matrix load of the row, then every gather, is issued before
        // the first use -- up to 8 + 8 independent loads in flight per lane. `width` is wave uniform.
        int2 c[8];
        Vec<K> xv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < width) c[t] = p[(size_t)t * WAVE];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < width) xv[t] = reinterpret_cast<const Vec<K>*>(x)[c[t].x];
#pragma unroll
        for (int t = 0; t < 8; ++t) if (t < width) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(__int_as_float(c[t].y), xv[t].v[q], acc[q]);
        }
        return;
    }
    int t = 0;
    for (; t + 4 <= width; t += 4) {   // 4 matrix loads, then 4 gathers in flight per lane
        const int2 c0 = p[(size_t)(t + 0) * WAVE], c1 = p[(size_t)(t + 1) * WAVE];
        const int2 c2 = p[(size_t)(t + 2) * WAVE], c3 = p[(size_t)(t + 3) * WAVE];
        const Vec<K> x0 = reinterpret_cast<const Vec<K>*>(x)[c0.x], x1 = reinterpret_cast<const Vec<K>*>(x)[c1.x];
        const Vec<K> x2 = reinterpret_cast<const Vec<K>*>(x)[c2.x], x3 = reinterpret_cast<const Vec<K>*>(x)[c3.x];
#pragma unroll
        for (int q = 0; q < K; ++q) {
            acc[q] = fmaf(__int_as_float(c0.y), x0.v[q], acc[q]);
            acc[q] = fmaf(__int_as_float(c1.y), x1.v[q], acc[q]);
            acc[q] = fmaf(__int_as_float(c2.y), x2.v[q], acc[q]);
            acc[q] = fmaf(__int_as_float(c3.y), x3.v[q], acc[q]);
        }
    }
    for (; t < width; ++t) {
        const int2 cv = p[(size_t)t * WAVE];
        fma_row<K>(acc, __int_as_float(cv.y), x, cv.x);
    }
}

}  // namespace ls
