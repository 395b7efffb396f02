// pcg.hip -- Jacobi-preconditioned conjugate gradient for M x = b, K interleaved right-hand sides.
//
// Replaces the per-step solve of the reference: largesteps/solvers.py:26-39 (CholeskySolver ->
// cholespy/CHOLMOD triangular solves) and :41-126 (ConjugateGradientSolver: per column, ~10 torch
// kernels and one host sync per iteration).
//
// One iteration = three kernels, all scalars stay on the device, every column has its own alpha/beta:
//   K1  Ap = M p                      ; partial  p.Ap                          (matrix in SELL-64)
//   K2  x += a p ; r -= a Ap          ; partial  r.D^-1 r , r.r        (a  = rz / pAp)
//   K3  p = D^-1 r + b p              ; publishes rz, ||r||^2, the column mask and the stop flag  (b = rz'/rz)
// Dot products: fp32 products accumulated in fp64 per thread, wave shuffle + LDS block reduction,
// one partial per workgroup; the NEXT kernel reduces the <= 1024 partials in a fixed order in every
// workgroup (deterministic, no atomics, no extra launch). The loads of a workgroup's first row tile are
// issued BEFORE that reduction so that the scalar hand-off hides under HBM latency.
// Geometry: 1024-thread workgroups, two per CU (32 waves/CU) for large systems; 256-thread
// workgroups for small ones. Row tiles are dealt to workgroups XCD-aware (common.h TileSched).
// The host enqueues iterations in chunks and polls a stop flag one chunk behind the GPU; kernels of
// iterations past the stop point return immediately.
//
// The same kernels serve the vertex-block sharded solver (largesteps/distributed.py): the matrix of a
// shard is rectangular (n_rows owned rows, n_cols = owned + halo columns), p carries the halo entries,
// and the partial arrays are summed across ranks (RCCL all-reduce) between the kernels, which is why
// the kernels can also be launched one at a time (ls_solver_phase).
#include "spmv_kernels.h"
#include <algorithm>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <new>
#include <vector>
#include <map>
#include <type_traits>

namespace ls {

constexpr int KMAX = 4;
constexpr int MAXG = 1024;            // capacity of one partial array (>= grid size of any PCG kernel)
constexpr int PROF_MAX_ITERS = 512;
constexpr int PART_PAP = 0, PART_RZ = 1, PART_SLOTS = 4;   // slots: 0 p.Ap | 1 r.z | 2 r.r | 3 b.b

struct Scal {
    double rz[2][KMAX];      // r.z, ring indexed by iteration parity
    double thr2[KMAX];       // squared stop threshold per column
    double bb[KMAX];         // ||b||^2
    double rr[KMAX];         // latest ||r||^2
    int mask[2];             // active-column bit mask, ring indexed by iteration parity
    int stop_iter;           // kernels of iteration n run iff n < stop_iter
    int bad;                 // 1: non-finite residual, 2: p.Ap <= 0 (matrix not SPD)
};

// partial array n (0 <= n < N) of a group that starts at slot0 and holds K columns per slot
__device__ __forceinline__ size_t part_off(int slot0, int K, int n) { return (size_t)((slot0 + n / K) * KMAX + n % K) * MAXG; }

template <int K>
__device__ __forceinline__ Vec<K> ldv(const float* __restrict__ a, int64_t i) { return reinterpret_cast<const Vec<K>*>(a)[i]; }
template <int K>
__device__ __forceinline__ void stv(float* __restrict__ a, int64_t i, const Vec<K>& v) { reinterpret_cast<Vec<K>*>(a)[i] = v; }

// Sum of N doubles per thread over a BS-thread workgroup, result in EVERY thread: wave totals by DPP -> LDS ->
// every wave re-reduces the <= 16 wave totals by DPP (same order everywhere: deterministic, one barrier, and
// only N live registers; the trailing barrier lets smem be reused). smem: (BS/64)*N doubles.
template <int N, int BS>
__device__ __forceinline__ void wg_sum(double (&x)[N], double* smem) {
    constexpr int NW = BS / WAVE;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = wave_sum_all(x[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) smem[w * N + i] = x[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double v = lane < NW ? smem[lane * N + i] : 0.0;
        x[i] = wave_sum_all(v);
    }
    __syncthreads();
}

// Deterministic reduction of N partial arrays (written by the G workgroups of the previous kernel, possibly
// summed across ranks) to N scalars in every thread. Entries [G, MAXG) of every array are kept at zero by the
// host (memset whenever the grid changes), so the loads are unconditional; they are issued by load_partials()
// -- callers place it right after their first tile loads, before any scalar control flow -- and consumed by
// finish_partials().
template <int N, int K, int BS>
__device__ __forceinline__ void load_partials(const double* __restrict__ part, int slot0, double (&v)[N][MAXG / BS]) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const double* __restrict__ pp = part + part_off(slot0, K, n);
#pragma unroll
        for (int j = 0; j < MAXG / BS; ++j) v[n][j] = pp[threadIdx.x + j * BS];
    }
}

template <int N, int BS>
__device__ __forceinline__ void finish_partials(const double (&v)[N][MAXG / BS], double (&out)[N], double* smem) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        double s = v[n][0];
#pragma unroll
        for (int j = 1; j < MAXG / BS; ++j) s += v[n][j];
        out[n] = s;
    }
    wg_sum<N, BS>(out, smem);
}

template <int N, int K, int BS>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ part, int slot0, double (&out)[N], double* smem) {
    double v[N][MAXG / BS];
    load_partials<N, K, BS>(part, slot0, v);
    finish_partials<N, BS>(v, out, smem);
}

template <int N, int K, int BS>
__device__ __forceinline__ void write_partials(double (&acc)[N], double* __restrict__ part, int slot0, double* smem) {
    wg_sum<N, BS>(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) part[part_off(slot0, K, n) + blockIdx.x] = acc[n];
    }
}

// tile schedule with BS-row tiles (see common.h TileSched for the XCD reasoning)
struct Sched {
    int first, step, end;
    __device__ __forceinline__ Sched(int T, int G) {
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

// ---- setup of one solve ----------------------------------------------------------------------------
template <int K, int BS, bool WARM>
__global__ __launch_bounds__(BS) void k_init(SellView S, const float* __restrict__ dinv, const float* __restrict__ b,
                                             const float* __restrict__ x0, float* __restrict__ x, float* __restrict__ r,
                                             float* __restrict__ p, double* __restrict__ part, int64_t V, int T, int G) {
    __shared__ double s_red[(BS / WAVE) * 3 * K];
    double acc[3 * K];
#pragma unroll
    for (int n = 0; n < 3 * K; ++n) acc[n] = 0.0;
    const Sched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float ax[K];
#pragma unroll
        for (int q = 0; q < K; ++q) ax[q] = 0.0f;
        if (WARM && (i & ~(int64_t)(WAVE - 1)) < V) row_sell<K>(S, x0, i, ax);
        if (i < V) {
            const Vec<K> bv = ldv<K>(b, i);
            const float di = dinv[i];
            Vec<K> xv, rv, pv;
            if (WARM) xv = ldv<K>(x0, i);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                if (!WARM) xv.v[q] = 0.0f;
                rv.v[q] = bv.v[q] - ax[q];
                pv.v[q] = di * rv.v[q];
                acc[q] += (double)rv.v[q] * (double)pv.v[q];            // r.z
                acc[K + q] += (double)rv.v[q] * (double)rv.v[q];        // r.r
                acc[2 * K + q] += (double)bv.v[q] * (double)bv.v[q];    // b.b
            }
            stv<K>(x, i, xv);
            stv<K>(r, i, rv);
            stv<K>(p, i, pv);
        }
    }
    write_partials<3 * K, K, BS>(acc, part, PART_RZ, s_red);   // slots r.z, r.r, b.b
}

template <int K>
__global__ __launch_bounds__(1024) void k_init_scal(const double* __restrict__ part, Scal* __restrict__ sc, int G, double rtol2, double atol2) {
    __shared__ double s_red[3 * K * 17];
    double v[3 * K];
    reduce_partials<3 * K, K, 1024>(part, PART_RZ, v, s_red);
    if (threadIdx.x == 0) {
        int mask = 0, bad = 0;
        for (int q = 0; q < K; ++q) {
            const double rz = v[q], rr = v[K + q], bb = v[2 * K + q];
            const double thr2 = fmax(rtol2 * bb, atol2);
            sc->rz[0][q] = rz;
            sc->rz[1][q] = rz;
            sc->rr[q] = rr;
            sc->bb[q] = bb;
            sc->thr2[q] = thr2;
            if (!(rr == rr) || !(bb == bb) || rr > 1e300 || bb > 1e300) bad = 1;
            else if (rr > thr2) mask |= 1 << q;
        }
        sc->mask[0] = mask;
        sc->mask[1] = mask;
        sc->bad = bad;
        sc->stop_iter = (mask == 0 || bad) ? 0 : INT_MAX;
    }
}

// ---- K1: Ap = M p, partial p.Ap ----------------------------------------------------------------------
template <int K, int BS>
__global__ __launch_bounds__(BS) void k_spmv_dot(SellView S, const float* __restrict__ p, float* __restrict__ Ap,
                                                 double* __restrict__ part, const Scal* __restrict__ sc, int it, int64_t V, int T, int G) {
    __shared__ double s_red[(BS / WAVE) * K];
    if (it >= sc->stop_iter) return;
    double acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0;
    const Sched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float ap[K];
#pragma unroll
        for (int q = 0; q < K; ++q) ap[q] = 0.0f;
        if ((i & ~(int64_t)(WAVE - 1)) < V) row_sell<K>(S, p, i, ap);
        if (i < V) {
            const Vec<K> pv = ldv<K>(p, i);
            Vec<K> o;
#pragma unroll
            for (int q = 0; q < K; ++q) { o.v[q] = ap[q]; acc[q] += (double)pv.v[q] * (double)ap[q]; }
            stv<K>(Ap, i, o);
        }
    }
    write_partials<K, K, BS>(acc, part, PART_PAP, s_red);
}

// ---- K2: x += alpha p ; r -= alpha Ap ; partial r.z and r.r ------------------------------------------------
template <int K, int BS>
__global__ __launch_bounds__(BS) void k_update(const float* __restrict__ dvec, const float* __restrict__ p,
                                               const float* __restrict__ Ap, float* __restrict__ x, float* __restrict__ r,
                                               double* __restrict__ part, Scal* __restrict__ sc, int it, int64_t V, int T, int G) {
    __shared__ double s_red[(BS / WAVE) * 2 * K];
    const Sched sch(T, G);
    int tile = sch.first;
    int64_t i = 0;
    bool valid = false;
    Vec<K> pv, av, xv, rv;
    float di = 0.0f;
    auto load = [&](int t) {
        i = (int64_t)t * BS + threadIdx.x;
        valid = t < sch.end && i < V;
        if (valid) { pv = ldv<K>(p, i); av = ldv<K>(Ap, i); xv = ldv<K>(x, i); rv = ldv<K>(r, i); di = dvec[i]; }
    };
    load(tile);                                   // in flight while the scalars are reduced
    double pl[K][MAXG / BS];
    load_partials<K, K, BS>(part, PART_PAP, pl);
    if (it >= sc->stop_iter) return;
    double pAp[K];
    finish_partials<K, BS>(pl, pAp, s_red);
    const int slot = it & 1;
    const int mask = sc->mask[slot];
    float alpha[K];
    int bad = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const bool on = (mask >> q) & 1;
        if (on && !(pAp[q] > 0.0)) bad = 2;
        alpha[q] = (on && pAp[q] > 0.0) ? (float)(sc->rz[slot][q] / pAp[q]) : 0.0f;
    }
    if (bad && blockIdx.x == 0 && threadIdx.x == 0) sc->bad = bad;
    double acc[2 * K];
#pragma unroll
    for (int n = 0; n < 2 * K; ++n) acc[n] = 0.0;
    while (tile < sch.end) {
        if (valid) {
#pragma unroll
            for (int q = 0; q < K; ++q) {
                xv.v[q] = fmaf(alpha[q], pv.v[q], xv.v[q]);
                rv.v[q] = fmaf(-alpha[q], av.v[q], rv.v[q]);
                const double rq = (double)rv.v[q];
                acc[q] += rq * (double)(di * rv.v[q]);
                acc[K + q] += rq * rq;
            }
            stv<K>(x, i, xv);
            stv<K>(r, i, rv);
        }
        tile += sch.step;
        load(tile);
    }
    write_partials<2 * K, K, BS>(acc, part, PART_RZ, s_red);   // slots r.z, r.r
}

// ---- K3: p = D^-1 r + beta p ; publish scalars for the next iteration --------------------------------
template <int K, int BS>
__global__ __launch_bounds__(BS) void k_direction(const float* __restrict__ dinv, const float* __restrict__ r, float* __restrict__ p,
                                                  const double* __restrict__ part, Scal* __restrict__ sc, int it, int64_t V, int T, int G) {
    __shared__ double s_red[(BS / WAVE) * 2 * K];
    const Sched sch(T, G);
    int tile = sch.first;
    int64_t i = 0;
    bool valid = false;
    Vec<K> rv, pv;
    float di = 0.0f;
    auto load = [&](int t) {
        i = (int64_t)t * BS + threadIdx.x;
        valid = t < sch.end && i < V;
        if (valid) { rv = ldv<K>(r, i); pv = ldv<K>(p, i); di = dinv[i]; }
    };
    load(tile);
    double pl[2 * K][MAXG / BS];
    load_partials<2 * K, K, BS>(part, PART_RZ, pl);
    if (it >= sc->stop_iter) return;
    double red[2 * K];
    finish_partials<2 * K, BS>(pl, red, s_red);
    const int mask = sc->mask[it & 1];
    float beta[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const double rz_old = sc->rz[it & 1][q];
        beta[q] = (((mask >> q) & 1) && rz_old > 0.0) ? (float)(red[q] / rz_old) : 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int nmask = 0, bad = 0;
        for (int q = 0; q < K; ++q) {
            const double rr = red[K + q];
            if ((mask >> q) & 1) {
                sc->rz[(it + 1) & 1][q] = red[q];
                sc->rr[q] = rr;
                if (!(rr == rr) || rr > 1e300) bad = 1;
                else if (rr > sc->thr2[q]) nmask |= 1 << q;
            } else {
                sc->rz[(it + 1) & 1][q] = sc->rz[it & 1][q];
            }
        }
        sc->mask[(it + 1) & 1] = nmask;
        if (bad) sc->bad = bad;
        if (nmask == 0 || bad || sc->bad) sc->stop_iter = it + 1;
    }
    while (tile < sch.end) {
        if (valid) {
#pragma unroll
            for (int q = 0; q < K; ++q) pv.v[q] = fmaf(beta[q], pv.v[q], di * rv.v[q]);
            stv<K>(p, i, pv);
        }
        tile += sch.step;
        load(tile);
    }
}

// ==== Chebyshev-accelerated Jacobi iteration: no dot products, ONE kernel per iteration =====================
// x_{k+1} = x_k + c1_k (x_k - x_{k-1}) + c2_k D^-1 (b - M x_k), the coefficients being the Chebyshev recurrence
// for the enclosure [lmin, lmax] of spec(D^-1 M) (Saad, Iterative Methods for Sparse Linear Systems, alg. 12.1
// written in the 3-term form). lmax is Gershgorin's bound max_i sum_j |m_ij| / m_ii, lmin = a_min / max_i m_ii
// with a_min <= lambda_min(M) supplied by the assembler (M = a I + b L, L positive semi-definite => a_min = a).
// Per iteration and vertex: the matrix row, one gather of x_k, b, 1/diag, x_{k-1} read and x_{k+1} written in
// place -- 8 nnz + 4 + 16k bytes, about half of a CG iteration, and no scalar hand-off between launches, so
// nothing ever waits for a reduction (single GPU) or an all-reduce (sharded).  The residual b - M x_k is
// recomputed, never recurred, so it is the TRUE fp32 residual.
template <int K, int BS, bool FIRST>
__global__ __launch_bounds__(BS) void k_cheb(SellView S, const float* __restrict__ dinv, const float* __restrict__ b,
                                             const float* __restrict__ xc, float* __restrict__ xpn, float c1, float c2,
                                             int64_t V, int T, int G) {
    const Sched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float ax[K];
#pragma unroll
        for (int q = 0; q < K; ++q) ax[q] = 0.0f;
        if ((i & ~(int64_t)(WAVE - 1)) < V) row_sell<K>(S, xc, i, ax);
        if (i < V) {
            const Vec<K> bv = ldv<K>(b, i), xv = ldv<K>(xc, i);
            const float di = dinv[i];
            Vec<K> xn;
            if (FIRST) {
#pragma unroll
                for (int q = 0; q < K; ++q) xn.v[q] = fmaf(c2, di * (bv.v[q] - ax[q]), xv.v[q]);
            } else {
                const Vec<K> xp = ldv<K>(xpn, i);
#pragma unroll
                for (int q = 0; q < K; ++q) xn.v[q] = fmaf(c2, di * (bv.v[q] - ax[q]), fmaf(c1, xv.v[q] - xp.v[q], xv.v[q]));
            }
            stv<K>(xpn, i, xn);
        }
    }
}

// ---- uniform-Laplacian specialisation of the Chebyshev step: the matrix values are implicit -----------------
// M = a I + b L_uniform  =>  (M x)_i = M_ii x_i - b * sum_{j in N(i)} x_j : only the neighbour ids are read
// (SELL-64 of int32 columns without the diagonal, 4 B per entry instead of 8 B per {col,val} pair). Padding
// entries point at row V of the iterate buffers, which the solver keeps at zero.
template <int K, int BS, bool FIRST>
__global__ __launch_bounds__(BS) void k_cheb_uniform(const int* __restrict__ slice_ptr, const int* __restrict__ cols,
                                                     const float* __restrict__ dd, const float* __restrict__ b,
                                                     const float* __restrict__ xc, float* __restrict__ xpn, float c1, float c2,
                                                     float offdiag, int64_t V, int T, int G) {
    const int lane = threadIdx.x & (WAVE - 1);
    const Sched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float sum[K];
#pragma unroll
        for (int q = 0; q < K; ++q) sum[q] = 0.0f;
        if ((i & ~(int64_t)(WAVE - 1)) < V) {
            const int slice = __builtin_amdgcn_readfirstlane((int)(i >> 6));
            const int off = __builtin_amdgcn_readfirstlane(slice_ptr[slice]);
            const int width = (__builtin_amdgcn_readfirstlane(slice_ptr[slice + 1]) - off) >> 6;
            const int* __restrict__ p = cols + off + lane;
            if (width <= 8) {
                int c[8];
                Vec<K> xv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) if (t < width) c[t] = p[(size_t)t * WAVE];
#pragma unroll
                for (int t = 0; t < 8; ++t) if (t < width) xv[t] = ldv<K>(xc, c[t]);
#pragma unroll
                for (int t = 0; t < 8; ++t) if (t < width) {
#pragma unroll
                    for (int q = 0; q < K; ++q) sum[q] += xv[t].v[q];
                }
            } else {
                for (int t = 0; t < width; ++t) {
                    const Vec<K> xv = ldv<K>(xc, p[(size_t)t * WAVE]);
#pragma unroll
                    for (int q = 0; q < K; ++q) sum[q] += xv.v[q];
                }
            }
        }
        if (i < V) {
            const Vec<K> bv = ldv<K>(b, i), xv = ldv<K>(xc, i);
            const float di = dd[i];
            Vec<K> xn;
            if (FIRST) {
#pragma unroll
                for (int q = 0; q < K; ++q) xn.v[q] = fmaf(c2, (bv.v[q] - fmaf(offdiag, sum[q], di * xv.v[q])) / di, xv.v[q]);
            } else {
                const Vec<K> xp = ldv<K>(xpn, i);
#pragma unroll
                for (int q = 0; q < K; ++q)
                    xn.v[q] = fmaf(c2, (bv.v[q] - fmaf(offdiag, sum[q], di * xv.v[q])) / di, fmaf(c1, xv.v[q] - xp.v[q], xv.v[q]));
            }
            stv<K>(xpn, i, xn);
        }
    }
}

// off-diagonal column ids of the CSR rows as SELL-64 (padding -> row V, a zero row of the iterate buffers)
__global__ __launch_bounds__(BLOCK) void k_sell_cols_fill(CsrView A, int64_t V, const int* __restrict__ slice_ptr, int* __restrict__ cols) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t slice = i >> 6;
    if ((i & ~(int64_t)(WAVE - 1)) >= V) return;
    const int off = slice_ptr[slice], width = (slice_ptr[slice + 1] - off) >> 6;
    int s = 0, e = 0;
    if (i < V) { s = A.rowptr[i]; e = A.rowptr[i + 1]; }
    int t = 0;
    for (int j = s; j < e; ++j) {
        const int cj = A.col[j];
        if (cj != (int)i) { cols[(size_t)off + (size_t)t * WAVE + lane] = cj; ++t; }
    }
    for (; t < width; ++t) cols[(size_t)off + (size_t)t * WAVE + lane] = (int)V;
}

// partials of ||b - M x||^2 (slot r.z), ||x||^2 (slot r.r) and ||b||^2, laid out so that k_init_scal can be reused
template <int K, int BS, bool ZERO_X>
__global__ __launch_bounds__(BS) void k_resnorm(SellView S, const float* __restrict__ b, const float* __restrict__ x,
                                                double* __restrict__ part, int64_t V, int T, int G) {
    __shared__ double s_red[(BS / WAVE) * 3 * K];
    double acc[3 * K];
#pragma unroll
    for (int n = 0; n < 3 * K; ++n) acc[n] = 0.0;
    const Sched sch(T, G);
    for (int tile = sch.first; tile < sch.end; tile += sch.step) {
        const int64_t i = (int64_t)tile * BS + threadIdx.x;
        float ax[K];
#pragma unroll
        for (int q = 0; q < K; ++q) ax[q] = 0.0f;
        if (!ZERO_X && (i & ~(int64_t)(WAVE - 1)) < V) row_sell<K>(S, x, i, ax);
        if (i < V) {
            const Vec<K> bv = ldv<K>(b, i);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const double rq = (double)(bv.v[q] - ax[q]);
                acc[q] += rq * rq;
                acc[2 * K + q] += (double)bv.v[q] * (double)bv.v[q];
            }
            if (!ZERO_X) {
                const Vec<K> xv = ldv<K>(x, i);
#pragma unroll
                for (int q = 0; q < K; ++q) acc[K + q] += (double)xv.v[q] * (double)xv.v[q];
            }
        }
    }
    write_partials<3 * K, K, BS>(acc, part, PART_RZ, s_red);
}

// Gershgorin bound of spec(D^-1 M) and max diagonal (one-off per matrix)
__global__ __launch_bounds__(BLOCK) void k_gershgorin(CsrView A, int64_t V, const float* __restrict__ dinv, int* __restrict__ out /* [2] float bits */) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    float ratio = 0.0f, d = 0.0f;
    if (i < V) {
        float sum = 0.0f;
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) sum += fabsf(A.val[j]);
        ratio = sum * dinv[i];
        d = 1.0f / dinv[i];
    }
    ratio = wave_max(ratio);
    d = wave_max(d);
    if ((threadIdx.x & (WAVE - 1)) == 0) {     // positive floats order like their bit patterns
        atomicMax(&out[0], __float_as_int(ratio));
        atomicMax(&out[1], __float_as_int(d));
    }
}

// ==== LDS-resident, s-step Chebyshev on mesh patches (temporal blocking; plan: largesteps/patches.py) ========
// One workgroup owns a compact patch of ~2k vertices plus its ghost layers 1..s and keeps BOTH iterates of all those
// vertices in LDS for s consecutive Chebyshev steps: the neighbour gathers of the sparse product become LDS reads and
// HBM sees vectors and matrix once per s iterations. Ghost layers go stale one layer per step (they are recomputed
// redundantly, layer s is read-only); the patch's own vertices never do. Uniform Laplacian only (implicit values):
// the matrix is the ELL list of LOCAL neighbour ids (uint16), padding -> slot n_local which holds zeros.
constexpr int PATCH_MAX_DEPTH = 12;     // steps per launch upper bound (= MAX_DEPTH of largesteps/patches.py)
struct PatchCoef { float c1[PATCH_MAX_DEPTH]; float c2[PATCH_MAX_DEPTH]; int steps; };
constexpr int PATCH_TABLE_COLS = 8 + PATCH_MAX_DEPTH;   // 8 header ints + lim[] (largesteps/patches.py)
constexpr int PATCH_RPT_MAX = 8;  // rows per thread upper bound: a patch may compute at most 8 * PATCH_BS rows

// LDS holds one K-float slot per local vertex. (Padding K = 3 to 16-byte slots for single ds_read_b128 gathers was
// measured: no gain -- the step loop is latency bound, not LDS-issue bound -- and it shrinks the patches that fit.)
template <int K> struct PatchSlot { static constexpr int KP = K; };
template <int KP> struct LdsVec;
template <> struct LdsVec<1> { typedef float T; };
template <> struct LdsVec<2> { typedef float2 T; };
template <> struct LdsVec<3> { struct T { float x, y, z; }; };
template <> struct LdsVec<4> { typedef float4 T; };

// the packed neighbour ids are unpacked where they are used: hipcc otherwise hoists the 64 shifts / masks of a thread's 8 rows out of
// the step loop and keeps both halves of every register resident (+32 VGPRs: the 1024-thread forms then spilled to scratch)
__device__ __forceinline__ unsigned in_loop(unsigned v) { asm volatile("" : "+v"(v)); return v; }

template <int K, int PATCH_BS, int PATCH_RPT>
__global__ __launch_bounds__(PATCH_BS) void k_patch_cheb(const int* __restrict__ table, const int* __restrict__ ghost_gid,
                                                         const unsigned short* __restrict__ cols16, const float* __restrict__ diag,
                                                         const float* __restrict__ b, const float* __restrict__ in_cur,
                                                         const float* __restrict__ in_prev, float* __restrict__ out_cur,
                                                         float* __restrict__ out_prev, PatchCoef coef, float offdiag, int cap1) {
    constexpr int KP = PatchSlot<K>::KP;
    typedef typename LdsVec<KP>::T slot_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    slot_t* cur = reinterpret_cast<slot_t*>(smem);
    slot_t* oth = cur + cap1;
    const int* __restrict__ t = table + (size_t)blockIdx.x * PATCH_TABLE_COLS;
    const int own_start = t[0], n_own = t[1], n_rows = t[2], n_local = t[3], W = t[4], og = t[5], oc = t[6], od = t[7];
    const int* __restrict__ lim = t + 8;   // lim[m] = rows of layers <= m: all that S-1-j remaining steps can still carry to an own vertex
    auto pack = [](const Vec<K>& v) { slot_t o; float* f = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int q = 0; q < KP; ++q) f[q] = q < K ? v.v[q] : 0.0f;
        return o; };
    for (int l = threadIdx.x; l < n_local; l += PATCH_BS) {
        const int g = l < n_own ? own_start + l : ghost_gid[og + l - n_own];
        cur[l] = pack(ldv<K>(in_cur, g));
        oth[l] = pack(ldv<K>(in_prev, g));
    }
    if (threadIdx.x == 0) { Vec<K> z; for (int q = 0; q < K; ++q) z.v[q] = 0.0f; cur[n_local] = pack(z); oth[n_local] = pack(z); }
    // per-thread rows r = tid + j * BS: right-hand side, diagonal and the (<= 8) local neighbour ids stay in registers
    // for all steps of this launch (two uint16 ids per register); wider rows re-read their ids from L2 every step
    float bl[PATCH_RPT][K], dd_j[PATCH_RPT];
    unsigned nb[PATCH_RPT][4];
    const bool narrow = W <= 8;
#pragma unroll
    for (int j = 0; j < PATCH_RPT; ++j) {
        const int r = threadIdx.x + j * PATCH_BS;
        dd_j[j] = 1.0f;
#pragma unroll
        for (int q = 0; q < K; ++q) bl[j][q] = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) nb[j][e] = (unsigned)n_local | ((unsigned)n_local << 16);   // zero slot
        if (r < n_rows) {
            const int g = r < n_own ? own_start + r : ghost_gid[og + r - n_own];
            const Vec<K> bv = ldv<K>(b, g);
#pragma unroll
            for (int q = 0; q < K; ++q) bl[j][q] = bv.v[q];
            dd_j[j] = diag[od + r];
            if (narrow) {
                const unsigned short* __restrict__ cr = cols16 + oc + r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < W) {
                        const unsigned c = cr[(size_t)e * n_rows];
                        nb[j][e >> 1] = (e & 1) ? ((nb[j][e >> 1] & 0xffffu) | (c << 16)) : ((nb[j][e >> 1] & 0xffff0000u) | c);
                    }
                }
            }
        }
    }
    __syncthreads();
    // one Chebyshev step on this thread's rows; NW = number of neighbour slots actually read (patch-uniform ELL width)
    auto step_rows = [&](auto nw_tag, float c1, float c2, int rows) {
        constexpr int NW = decltype(nw_tag)::value;
#pragma unroll
        for (int j = 0; j < PATCH_RPT; ++j) {
            const int r = threadIdx.x + j * PATCH_BS;
            if (r < rows) {
                float sum[K];
#pragma unroll
                for (int q = 0; q < K; ++q) sum[q] = 0.0f;
                if (NW > 0) {
                    // gathers in batches of GB slots: all NW at once (8 x K registers on top of the 8 rows' resident b, diagonal and
                    // ids) spilled 16-37 registers to scratch in the 1024-thread form for K >= 3; the sum's order is unchanged
                    constexpr int GB = (K >= 3 && PATCH_BS == 1024) ? 4 : (NW > 0 ? NW : 1);
#pragma unroll
                    for (int e0 = 0; e0 < NW; e0 += GB) {
                        slot_t g8[GB];
#pragma unroll
                        for (int e = e0; e < e0 + GB; ++e)        // padding ids point at the zero slot: no branch needed
                            if (e < NW) { const unsigned pk = in_loop(nb[j][e >> 1]); g8[e - e0] = cur[(e & 1) ? (pk >> 16) : (pk & 0xffffu)]; }
#pragma unroll
                        for (int e = e0; e < e0 + GB; ++e) {
                            if (e < NW) {
                                const float* f = reinterpret_cast<const float*>(&g8[e - e0]);
#pragma unroll
                                for (int q = 0; q < K; ++q) sum[q] += f[q];
                            }
                        }
                    }
                } else {
                    const unsigned short* __restrict__ cr = cols16 + oc + r;
                    for (int e = 0; e < W; ++e) {
                        const slot_t gv = cur[cr[(size_t)e * n_rows]];
                        const float* f = reinterpret_cast<const float*>(&gv);
#pragma unroll
                        for (int q = 0; q < K; ++q) sum[q] += f[q];
                    }
                }
                const slot_t xcv = cur[r];
                slot_t xpv = oth[r];
                const float* xc = reinterpret_cast<const float*>(&xcv);
                float* xp = reinterpret_cast<float*>(&xpv);
                const float w = c2 / dd_j[j];           // one division per row and step: c2 D^-1 (b - M x)
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    const float ax = fmaf(offdiag, sum[q], dd_j[j] * xc[q]);
                    xp[q] = fmaf(w, bl[j][q] - ax, fmaf(c1, xc[q] - xp[q], xc[q]));
                }
                oth[r] = xpv;   // oth[r] is touched by this thread only during this step: update in place
            }
        }
    };
    for (int step = 0; step < coef.steps; ++step) {
        const float c1 = coef.c1[step], c2 = coef.c2[step];
        const int rows = lim[coef.steps - 1 - step];   // shrinking steps: stale outer layers are not recomputed
        if (!narrow) step_rows(std::integral_constant<int, 0>(), c1, c2, rows);
        else if (W <= 6) step_rows(std::integral_constant<int, 6>(), c1, c2, rows);
        else if (W == 7) step_rows(std::integral_constant<int, 7>(), c1, c2, rows);
        else step_rows(std::integral_constant<int, 8>(), c1, c2, rows);
        __syncthreads();
        slot_t* tmp = cur; cur = oth; oth = tmp;
    }
    for (int l = threadIdx.x; l < n_own; l += PATCH_BS) {
        Vec<K> a, c;
        const slot_t av = cur[l], cv = oth[l];
        const float* fa = reinterpret_cast<const float*>(&av);
        const float* fc = reinterpret_cast<const float*>(&cv);
#pragma unroll
        for (int q = 0; q < K; ++q) { a.v[q] = fa[q]; c.v[q] = fc[q]; }
        stv<K>(out_cur, own_start + l, a);
        stv<K>(out_prev, own_start + l, c);
    }
}

template <int K>
__global__ __launch_bounds__(BLOCK) void k_scatter_rows(const float* __restrict__ src, const int* __restrict__ idx, int64_t n, float* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) stv<K>(dst, idx[t], ldv<K>(src, t));
}

// ---- CSR -> SELL-64 ---------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_sell_widths(const int* __restrict__ rowptr, int64_t V, int S, int* __restrict__ width64, int minus) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int len = (i < V) ? max(rowptr[i + 1] - rowptr[i] - minus, 0) : 0;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) len = max(len, __shfl_down(len, off, WAVE));
    const int64_t slice = i >> 6;
    if ((threadIdx.x & (WAVE - 1)) == 0 && slice < S) width64[slice] = len * WAVE;
}

// exclusive scan of the slice sizes by one 1024-thread workgroup (S <= V/64; one-off per matrix)
__global__ __launch_bounds__(1024) void k_sell_scan(const int* __restrict__ width64, int S, int* __restrict__ slice_ptr) {
    __shared__ long long s_w[17];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    long long carry = 0;
    for (int base = 0; base < S; base += 1024) {
        const int s = base + threadIdx.x;
        const long long v = s < S ? width64[s] : 0;
        long long inc = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { long long y = __shfl_up(inc, off, WAVE); if (lane >= off) inc += y; }
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        if (threadIdx.x == 0) { long long run = 0; for (int j = 0; j < 16; ++j) { long long t = s_w[j]; s_w[j] = run; run += t; } s_w[16] = run; }
        __syncthreads();
        const long long ex = carry + s_w[w] + inc - v;
        if (s < S) slice_ptr[s] = ex > (long long)INT_MAX ? -1 : (int)ex;
        carry += s_w[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) slice_ptr[S] = carry > (long long)INT_MAX ? -1 : (int)carry;
}

__global__ __launch_bounds__(BLOCK) void k_sell_fill(CsrView A, int64_t V, const int* __restrict__ slice_ptr, int2* __restrict__ cv) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t slice = i >> 6;
    if ((i & ~(int64_t)(WAVE - 1)) >= V) return;
    const int off = slice_ptr[slice], width = (slice_ptr[slice + 1] - off) >> 6;
    int s = 0, len = 0;
    if (i < V) { s = A.rowptr[i]; len = A.rowptr[i + 1] - s; }
    const int own = (int)min(i, V - 1);
    for (int t = 0; t < width; ++t) {
        int2 e = make_int2(own, 0);                       // padding: val = 0, a valid (own) column
        if (t < len) e = make_int2(A.col[s + t], __float_as_int(A.val[s + t]));
        cv[(size_t)off + (size_t)t * WAVE + lane] = e;
    }
}

__global__ __launch_bounds__(BLOCK) void k_diag_inv(CsrView A, int64_t V, float* __restrict__ dinv, float* __restrict__ dd,
                                                    int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= V) return;
    float d = 0.0f;
    for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) if (A.col[j] == (int)i) d = A.val[j];
    if (!(d > 0.0f)) *flag = 1;
    dinv[i] = 1.0f / d;
    if (dd) dd[i] = d;
}

template <int K>
__global__ __launch_bounds__(BLOCK) void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, int64_t n, float* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) stv<K>(dst, t, ldv<K>(src, idx[t]));
}

}  // namespace ls

using namespace ls;

struct ls_solver {
    int device = 0;
    int64_t V = 0, ncols = 0, nnz = 0;
    int kmax = 0;
    CsrView csr{};
    SellView sell{};
    int* slice_ptr = nullptr;
    int2* sell_cv = nullptr;
    int64_t sell_entries = 0;
    float *dinv = nullptr, *r = nullptr, *p = nullptr, *Ap = nullptr;
    float* dd = nullptr;                  // square systems: the diagonal
    // uniform-Laplacian specialisation (ls_solver_set_uniform): column-only SELL, zero-padded iterate buffers
    int* slice_ptr_u = nullptr;
    int* cols_u = nullptr;
    float uni_offdiag = 0.0f;
    bool uni = false;
    // Chebyshev work buffers (square systems): iterates with one extra (zero) row, staged right-hand side; the n
    // launches of a solve are replayed as one hipGraph per (k, n, kernel flavour)
    float *xu0 = nullptr, *xu1 = nullptr, *bh = nullptr;
    hipStream_t cap = nullptr;
    std::map<uint64_t, hipGraphExec_t> graphs;
    int use_graph = 1;
    // patch plan of the LDS-resident s-step kernel (ls_solver_set_patches)
    struct {
        int n = 0, depth = 0, cap1 = 0, bs = 512, rpt = 8;
        int *table = nullptr, *gid = nullptr, *perm = nullptr;
        unsigned short* cols = nullptr;
        float *diag = nullptr, *bn = nullptr, *it[4] = {nullptr, nullptr, nullptr, nullptr};
    } patch;
    int use_patch = 1;
    int profile_eager = 0;        // 1: profile the one-step kernel even when a patch plan exists
    double a_min = 0.0;           // caller-certified lower bound of lambda_min(M); 0 = unknown (Chebyshev refused)
    double gersh = 0.0, dmax = 0.0;   // Gershgorin bound of spec(D^-1 M), max diagonal entry
    int last_G = -1;              // grid the partial arrays were last written with (tail must stay zero)
    double* part = nullptr;
    Scal* scal = nullptr;
    Scal* h_scal = nullptr;       // pinned, 2 polling slots + 1 final
    hipEvent_t ev[2] = {nullptr, nullptr};
    int check_every = 16, grid = 0, block = 0, last_iters = 0;
    bool own_p = true, own_part = true;   // false after ls_solver_bind: the caller owns those buffers
    size_t bytes = 0;
    // optional per-kernel timing with HIP events on the solve's own stream (ls_solver_set("profile", 1))
    int profile = 0;
    std::vector<hipEvent_t> pev;
    double prof_ms[3] = {0, 0, 0};   // accumulated K1, K2, K3 time of the last profiled solve
    int prof_iters = 0;
};

namespace {

struct Geometry { int bs, T, G; };

Geometry geometry(const ls_solver* s) {
    Geometry g;
    // 512-thread workgroups once there are enough rows to give every CU several of them, 256 below that
    g.bs = s->block ? s->block : (s->V >= 262144 ? 512 : 256);
    g.T = div_up(s->V, g.bs);
    if (s->grid > 0) {
        g.G = std::min(s->grid, MAXG);            // explicit: exactly this many workgroups (shards must agree on it)
    } else {
        const int cap = g.bs == 1024 ? 512 : 1024;   // 2 x 1024, 4 x 512 or 4 x 256 threads per CU
        // a multiple of 8 (XCD-aware schedule) that covers every tile in one pass whenever the cap allows it
        g.G = g.T < 8 ? std::max(g.T, 1) : std::min((g.T + 7) & ~7, cap);
    }
    return g;
}

template <typename T>
int dev_alloc(ls_solver* s, T** out, size_t n) {
    void* p = nullptr;
    const size_t b = std::max<size_t>(n * sizeof(T), 256);
    LS_HIP(hipMalloc(&p, b));
    s->bytes += b;
    *out = (T*)p;
    return LS_OK;
}

void free_solver(ls_solver* s) {
    if (!s) return;
    (void)hipFree(s->slice_ptr); (void)hipFree(s->sell_cv); (void)hipFree(s->dinv); (void)hipFree(s->r);
    (void)hipFree(s->slice_ptr_u); (void)hipFree(s->cols_u); (void)hipFree(s->xu0); (void)hipFree(s->xu1);
    (void)hipFree(s->dd);
    if (s->own_p) (void)hipFree(s->p);
    if (s->own_part) (void)hipFree(s->part);
    (void)hipFree(s->Ap); (void)hipFree(s->scal);
    if (s->h_scal) (void)hipHostFree(s->h_scal);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->pev) if (e) (void)hipEventDestroy(e);
    for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
    if (s->cap) (void)hipStreamDestroy(s->cap);
    (void)hipFree(s->bh);
    (void)hipFree(s->patch.table); (void)hipFree(s->patch.gid); (void)hipFree(s->patch.perm); (void)hipFree(s->patch.cols);
    (void)hipFree(s->patch.diag); (void)hipFree(s->patch.bn);
    for (float* q : s->patch.it) (void)hipFree(q);
    delete s;
}

int create_impl(ls_solver* s, hipStream_t st) {
    const int64_t V = s->V;
    const size_t vk = (size_t)std::max<int64_t>(V, 1) * s->kmax;
    const size_t pk = (size_t)std::max<int64_t>(s->ncols, 1) * s->kmax;
    int rc;
    if ((rc = dev_alloc(s, &s->dinv, (size_t)std::max<int64_t>(V, 1)))) return rc;
    if ((rc = dev_alloc(s, &s->r, vk))) return rc;
    if ((rc = dev_alloc(s, &s->p, pk))) return rc;
    if ((rc = dev_alloc(s, &s->Ap, vk))) return rc;
    if ((rc = dev_alloc(s, &s->part, (size_t)PART_SLOTS * KMAX * MAXG))) return rc;
    if ((rc = dev_alloc(s, &s->scal, 1))) return rc;
    LS_HIP(hipHostMalloc((void**)&s->h_scal, 3 * sizeof(Scal), hipHostMallocDefault));
    for (auto& e : s->ev) LS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    LS_HIP(hipMemsetAsync(s->part, 0, sizeof(double) * PART_SLOTS * KMAX * MAXG, st));
    LS_HIP(hipMemsetAsync(s->p, 0, sizeof(float) * pk, st));
    LS_HIP(hipMemsetAsync(s->scal, 0, sizeof(Scal), st));
    if (V == 0) return LS_OK;
    // Jacobi preconditioner
    int* flag = (int*)s->scal;   // scal is re-initialised by every solve; borrow its first word
    const bool square = s->ncols == s->V;
    if (square) {
        const size_t n1 = (size_t)(V + 1) * s->kmax;
        if ((rc = dev_alloc(s, &s->dd, (size_t)std::max<int64_t>(V, 1)))) return rc;
        if ((rc = dev_alloc(s, &s->xu0, n1))) return rc;
        if ((rc = dev_alloc(s, &s->xu1, n1))) return rc;
        if ((rc = dev_alloc(s, &s->bh, vk))) return rc;
        LS_HIP(hipMemsetAsync(s->xu0, 0, sizeof(float) * n1, st));
        LS_HIP(hipMemsetAsync(s->xu1, 0, sizeof(float) * n1, st));
        LS_HIP(hipStreamCreateWithFlags(&s->cap, hipStreamNonBlocking));
    }
    hipLaunchKernelGGL(k_diag_inv, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, s->csr, V, s->dinv, s->dd, flag);
    // SELL-64 copy of the matrix
    const int S = div_up(V, WAVE);
    int* width64 = nullptr;
    if ((rc = dev_alloc(s, &s->slice_ptr, (size_t)S + 1))) return rc;
    LS_HIP(hipMalloc((void**)&width64, sizeof(int) * (size_t)S));
    hipLaunchKernelGGL(k_sell_widths, dim3(div_up((int64_t)S * WAVE, BLOCK)), dim3(BLOCK), 0, st, s->csr.rowptr, V, S, width64, 0);
    hipLaunchKernelGGL(k_sell_scan, dim3(1), dim3(1024), 0, st, width64, S, s->slice_ptr);
    hipLaunchKernelGGL(k_gershgorin, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, s->csr, V, s->dinv, flag + 2);
    int h[5] = {0, 0, 0, 0, 0};   // SELL entries | diag flag, (unused), Gershgorin bits, max-diag bits
    LS_HIP(hipMemcpyAsync(&h[0], s->slice_ptr + S, sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipMemcpyAsync(&h[1], flag, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    (void)hipFree(width64);
    { float f; memcpy(&f, &h[3], 4); s->gersh = f; memcpy(&f, &h[4], 4); s->dmax = f; }
    LS_REQUIRE(h[1] == 0, LS_E_INVALID, "matrix has a missing or non-positive diagonal entry: not SPD, Jacobi-PCG refused");
    LS_REQUIRE(h[0] >= 0, LS_E_OVERFLOW, "SELL copy of the matrix overflows int32 entry offsets");
    s->sell_entries = h[0];
    if ((rc = dev_alloc(s, &s->sell_cv, (size_t)std::max(h[0], 1)))) return rc;
    hipLaunchKernelGGL(k_sell_fill, dim3(div_up((int64_t)S * WAVE, BLOCK)), dim3(BLOCK), 0, st, s->csr, V, s->slice_ptr, s->sell_cv);
    LS_HIP(hipMemsetAsync(s->scal, 0, sizeof(Scal), st));
    LS_HIP(hipGetLastError());
    s->sell = SellView{s->slice_ptr, s->sell_cv};
    return LS_OK;
}

// ---- kernel launchers (one per phase; also used one at a time by the sharded driver) -----------------
// PCG: 0 init, 1 init_scal, 2 K1, 3 K2, 4 K3 | Chebyshev: 9 step, 11 step (implicit uniform values) | 10 residual norms
template <int K, int BS>
void launch_phase(ls_solver* s, int phase, const float* b, const float* x0, float* x, double rtol, double atol, int it,
                  const Geometry& g, hipStream_t st) {
    const dim3 grid(g.G), block(BS);
    switch (phase) {
        case 0:
            if (x0) hipLaunchKernelGGL((k_init<K, BS, true>), grid, block, 0, st, s->sell, s->dinv, b, x0, x, s->r, s->p, s->part, s->V, g.T, g.G);
            else hipLaunchKernelGGL((k_init<K, BS, false>), grid, block, 0, st, s->sell, s->dinv, b, x0, x, s->r, s->p, s->part, s->V, g.T, g.G);
            break;
        case 1:
            hipLaunchKernelGGL(k_init_scal<K>, dim3(1), dim3(1024), 0, st, s->part, s->scal, g.G, rtol * rtol, atol * atol);
            break;
        case 2:
            hipLaunchKernelGGL((k_spmv_dot<K, BS>), grid, block, 0, st, s->sell, s->p, s->Ap, s->part, s->scal, it, s->V, g.T, g.G);
            break;
        case 3:
            hipLaunchKernelGGL((k_update<K, BS>), grid, block, 0, st, s->dinv, s->p, s->Ap, x, s->r, s->part, s->scal, it, s->V, g.T, g.G);
            break;
        case 4:
            hipLaunchKernelGGL((k_direction<K, BS>), grid, block, 0, st, s->dinv, s->r, s->p, s->part, s->scal, it, s->V, g.T, g.G);
            break;
        case 9:   // Chebyshev step: x0 = current iterate (gathered), x = x_{k-1} in / x_{k+1} out, rtol/atol carry c1/c2
            if (it == 0) hipLaunchKernelGGL((k_cheb<K, BS, true>), grid, block, 0, st, s->sell, s->dinv, b, x0, x, (float)rtol, (float)atol, s->V, g.T, g.G);
            else hipLaunchKernelGGL((k_cheb<K, BS, false>), grid, block, 0, st, s->sell, s->dinv, b, x0, x, (float)rtol, (float)atol, s->V, g.T, g.G);
            break;
        case 11:  // Chebyshev step, uniform-Laplacian specialisation (same argument convention as 9)
            if (it == 0) hipLaunchKernelGGL((k_cheb_uniform<K, BS, true>), grid, block, 0, st, s->slice_ptr_u, s->cols_u, s->dd, b, x0, x, (float)rtol, (float)atol, s->uni_offdiag, s->V, g.T, g.G);
            else hipLaunchKernelGGL((k_cheb_uniform<K, BS, false>), grid, block, 0, st, s->slice_ptr_u, s->cols_u, s->dd, b, x0, x, (float)rtol, (float)atol, s->uni_offdiag, s->V, g.T, g.G);
            break;
        default:  // 10: residual / rhs norms of x0 (nullptr: x = 0)
            if (x0) hipLaunchKernelGGL((k_resnorm<K, BS, false>), grid, block, 0, st, s->sell, b, x0, s->part, s->V, g.T, g.G);
            else hipLaunchKernelGGL((k_resnorm<K, BS, true>), grid, block, 0, st, s->sell, b, x0, s->part, s->V, g.T, g.G);
            break;
    }
}

int dispatch_phase(ls_solver* s, int k, int phase, const float* b, const float* x0, float* x, double rtol, double atol, int it,
                   const Geometry& g, hipStream_t st) {
    if ((phase == 0 || phase == 10) && g.G != s->last_G) {   // partial entries [G, MAXG) must read as zero
        LS_HIP(hipMemsetAsync(s->part, 0, sizeof(double) * PART_SLOTS * KMAX * MAXG, st));
        s->last_G = g.G;
    }
#define LS_CASE(KK)                                                                                          \
    case KK:                                                                                                 \
        if (g.bs == 1024) launch_phase<KK, 1024>(s, phase, b, x0, x, rtol, atol, it, g, st);                 \
        else if (g.bs == 512) launch_phase<KK, 512>(s, phase, b, x0, x, rtol, atol, it, g, st);              \
        else launch_phase<KK, 256>(s, phase, b, x0, x, rtol, atol, it, g, st);                               \
        break;
    switch (k) { LS_CASE(1) LS_CASE(2) LS_CASE(3) LS_CASE(4) default: break; }
#undef LS_CASE
    return LS_OK;
}

void fill_info(const Scal& f, int k, int n_enqueued, ls_solve_info* info, bool* conv_out, int* iters_out) {
    const bool conv = f.stop_iter != INT_MAX && f.bad == 0;
    const int iters = f.stop_iter != INT_MAX ? f.stop_iter : n_enqueued;
    if (info) {
        info->iterations = iters;
        info->converged = conv ? 1 : 0;
        for (int q = 0; q < 4; ++q) {
            info->rnorm[q] = q < k ? sqrt(f.rr[q]) : 0.0;
            info->bnorm[q] = q < k ? sqrt(f.bb[q]) : 0.0;
        }
    }
    *conv_out = conv;
    *iters_out = iters;
}

int solve_impl(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol, double atol, int max_iter,
               ls_solve_info* info, hipStream_t st) {
    const Geometry g = geometry(s);
    int rc0;
    if ((rc0 = dispatch_phase(s, k, 0, b, x0, x, rtol, atol, 0, g, st))) return rc0;
    dispatch_phase(s, k, 1, b, x0, x, rtol, atol, 0, g, st);
    if (s->profile && s->pev.empty()) {
        s->pev.assign(4 * PROF_MAX_ITERS, nullptr);
        for (auto& e : s->pev) LS_HIP(hipEventCreate(&e));
    }
    int n = 0, chunk_id = 0;
    bool stopped = false;
    // first chunk: as long as the previous solve on this matrix needed (iteration counts are stable)
    int chunk = std::max(s->check_every, std::min(s->last_iters, max_iter));
    while (n < max_iter && !stopped) {
        const int todo = std::min(chunk, max_iter - n);
        for (int j = 0; j < todo; ++j, ++n) {
            const bool prof = s->profile && n < PROF_MAX_ITERS;
            if (prof) LS_HIP(hipEventRecord(s->pev[4 * n + 0], st));
            dispatch_phase(s, k, 2, b, x0, x, rtol, atol, n, g, st);
            if (prof) LS_HIP(hipEventRecord(s->pev[4 * n + 1], st));
            dispatch_phase(s, k, 3, b, x0, x, rtol, atol, n, g, st);
            if (prof) LS_HIP(hipEventRecord(s->pev[4 * n + 2], st));
            dispatch_phase(s, k, 4, b, x0, x, rtol, atol, n, g, st);
            if (prof) LS_HIP(hipEventRecord(s->pev[4 * n + 3], st));
        }
        LS_HIP(hipGetLastError());
        const int slot = chunk_id & 1;
        LS_HIP(hipMemcpyAsync(&s->h_scal[slot], s->scal, sizeof(Scal), hipMemcpyDeviceToHost, st));
        LS_HIP(hipEventRecord(s->ev[slot], st));
        if (chunk_id >= 1) {   // look at the chunk before the one just enqueued: the GPU never idles
            const int prev = (chunk_id - 1) & 1;
            LS_HIP(hipEventSynchronize(s->ev[prev]));
            if (s->h_scal[prev].stop_iter != INT_MAX) stopped = true;
        }
        ++chunk_id;
        chunk = s->check_every;
    }
    LS_HIP(hipMemcpyAsync(&s->h_scal[2], s->scal, sizeof(Scal), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    const Scal& f = s->h_scal[2];
    bool conv;
    int iters;
    fill_info(f, k, n, info, &conv, &iters);
    if (conv) s->last_iters = iters;
    if (s->profile) {   // only iterations that really ran (kernels past the stop point return at once)
        s->prof_iters = std::min(iters, PROF_MAX_ITERS);
        for (double& m : s->prof_ms) m = 0.0;
        for (int i = 0; i < s->prof_iters; ++i)
            for (int kk = 0; kk < 3; ++kk) {
                float ms = 0.f;
                LS_HIP(hipEventElapsedTime(&ms, s->pev[4 * i + kk], s->pev[4 * i + kk + 1]));
                s->prof_ms[kk] += ms;
            }
    }
    if (!conv) {
        if (f.bad == 1) set_error("PCG: non-finite residual after %d iterations", iters);
        else if (f.bad == 2) set_error("PCG: p.Ap <= 0 after %d iterations: matrix is not positive definite", iters);
        else set_error("PCG: not converged after %d iterations (max_iter=%d)", iters, max_iter);
        return LS_E_NOT_CONVERGED;
    }
    return LS_OK;
}

// n Chebyshev steps as ceil(n / depth) launches of the patch kernel (patch-major numbering inside)
template <int K>
void launch_patch(ls_solver* s, const float* in_cur, const float* in_prev, float* out_cur, float* out_prev, const PatchCoef& coef, hipStream_t st) {
    const size_t lds = 2 * (size_t)s->patch.cap1 * PatchSlot<K>::KP * sizeof(float);
#define LS_PATCH_LAUNCH(BS, RPT)                                                                                        \
    hipLaunchKernelGGL((k_patch_cheb<K, BS, RPT>), dim3(s->patch.n), dim3(BS), lds, st, s->patch.table, s->patch.gid,    \
                       s->patch.cols, s->patch.diag, s->patch.bn, in_cur, in_prev, out_cur, out_prev, coef, s->uni_offdiag, \
                       s->patch.cap1)
    if (s->patch.bs == 1024) { if (s->patch.rpt <= 6) LS_PATCH_LAUNCH(1024, 6); else LS_PATCH_LAUNCH(1024, 8); }
    else { if (s->patch.rpt <= 6) LS_PATCH_LAUNCH(512, 6); else LS_PATCH_LAUNCH(512, 8); }
#undef LS_PATCH_LAUNCH
}

int solve_cheb_patched(ls_solver* s, const float* b, const float* x0, float* x, int k, int n, double theta, double delta,
                       double sigma1, hipStream_t st) {
    const int64_t V = s->V;
    const dim3 vg(div_up(V, BLOCK)), vb(BLOCK);
    const size_t bytes = sizeof(float) * (size_t)V * k;
    float** it = s->patch.it;      // pairs (it[0], it[1]) and (it[2], it[3]): (current, previous)
#define LS_K(KK, ...) switch (KK) { case 1: { constexpr int K = 1; __VA_ARGS__; } break; case 2: { constexpr int K = 2; __VA_ARGS__; } break; \
                                    case 3: { constexpr int K = 3; __VA_ARGS__; } break; default: { constexpr int K = 4; __VA_ARGS__; } break; }
    LS_K(k, hipLaunchKernelGGL(k_gather_rows<K>, vg, vb, 0, st, b, s->patch.perm, V, s->patch.bn));
    if (x0) { LS_K(k, hipLaunchKernelGGL(k_gather_rows<K>, vg, vb, 0, st, x0, s->patch.perm, V, it[0])); }
    else LS_HIP(hipMemsetAsync(it[0], 0, bytes, st));
    LS_HIP(hipMemsetAsync(it[1], 0, bytes, st));
    if (s->profile) LS_HIP(hipEventRecord(s->pev[0], st));
    double rho = 1.0 / sigma1;
    int src = 0;
    for (int it0 = 0; it0 < n; it0 += s->patch.depth) {
        PatchCoef coef;
        coef.steps = std::min(s->patch.depth, n - it0);
        for (int j = 0; j < coef.steps; ++j) {
            double c1 = 0.0, c2 = 1.0 / theta;
            if (it0 + j > 0) {
                const double rho_new = 1.0 / (2.0 * sigma1 - rho);
                c1 = rho_new * rho;
                c2 = 2.0 * rho_new / delta;
                rho = rho_new;
            }
            coef.c1[j] = (float)c1;
            coef.c2[j] = (float)c2;
        }
        LS_K(k, launch_patch<K>(s, it[src], it[src + 1], it[2 - src], it[3 - src], coef, st));
        src = 2 - src;
    }
    if (s->profile) LS_HIP(hipEventRecord(s->pev[1], st));
    LS_K(k, hipLaunchKernelGGL(k_scatter_rows<K>, vg, vb, 0, st, (const float*)it[src], (const int*)s->patch.perm, V, x));
#undef LS_K
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// Chebyshev-Jacobi solve: the iteration count follows from the spectral enclosure and the requested reduction,
// nothing is polled: n launches, one residual check, one sync.
int solve_cheb(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol, double atol, int max_iter,
               ls_solve_info* info, hipStream_t st) {
    LS_REQUIRE(s->a_min > 0.0 && s->gersh > 0.0 && s->dmax > 0.0, LS_E_STATE,
               "Chebyshev: no certified spectral enclosure for this matrix (ls_solver_set_spectrum)");
    const Geometry g = geometry(s);
    const double lmin = 0.98 * s->a_min / s->dmax, lmax = s->gersh * (1.0 + 1e-5);
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma1 = theta / delta;
    const double sk = sqrt(lmax / lmin), rate = (sk - 1.0) / (sk + 1.0);
    int rc;
    // requested residual reduction relative to the starting residual
    double target = rtol;      // cold start without atol: ||r0|| = ||b||
    double bb[4] = {0, 0, 0, 0};
    if (x0 || atol > 0.0) {
        if ((rc = dispatch_phase(s, k, 10, b, x0, x, rtol, atol, 0, g, st))) return rc;
        dispatch_phase(s, k, 1, b, x0, x, rtol, atol, 0, g, st);
        LS_HIP(hipMemcpyAsync(&s->h_scal[0], s->scal, sizeof(Scal), hipMemcpyDeviceToHost, st));
        LS_HIP(hipStreamSynchronize(st));
        target = 1.0;
        for (int q = 0; q < k; ++q) {
            const double r0 = sqrt(s->h_scal[0].rz[0][q]), thr = sqrt(s->h_scal[0].thr2[q]);
            bb[q] = s->h_scal[0].bb[q];
            if (r0 > thr) target = std::min(target, thr / r0);
        }
    }
    if (s->profile && s->pev.empty()) {
        s->pev.assign(4 * PROF_MAX_ITERS, nullptr);
        for (auto& e : s->pev) LS_HIP(hipEventCreate(&e));
    }
    int n = 0;
    if (target < 1.0) n = (int)ceil(log(2.0 / std::max(target, 1e-30)) / -log(rate));
    const bool capped = n > max_iter;
    n = std::min(n, max_iter);
    if (s->patch.n > 0 && s->use_patch && s->uni && !s->profile_eager && 2 * (size_t)s->patch.cap1 * k * sizeof(float) <= 160 * 1024) {
        if ((rc = solve_cheb_patched(s, b, x0, x, k, n, theta, delta, sigma1, st))) return rc;
    } else {
    // iterate `it` gathers from (it even ? Y : Z) and overwrites the other buffer; both are handle-owned with one
    // extra row (index V) kept at zero: the padding entries of the column-only SELL point at it
    const bool uni = s->uni;
    float *Y = s->xu0, *Z = s->xu1;
    const size_t bytes = sizeof(float) * (size_t)s->V * k;
    LS_HIP(hipMemsetAsync(Y + (size_t)s->V * k, 0, sizeof(float) * k, st));   // the zero row moves with k
    LS_HIP(hipMemsetAsync(Z + (size_t)s->V * k, 0, sizeof(float) * k, st));
    if (x0) LS_HIP(hipMemcpyAsync(Y, x0, bytes, hipMemcpyDeviceToDevice, st));
    else LS_HIP(hipMemsetAsync(Y, 0, bytes, st));
    if (s->profile && s->pev.empty()) {
        s->pev.assign(4 * PROF_MAX_ITERS, nullptr);
        for (auto& e : s->pev) LS_HIP(hipEventCreate(&e));
    }
    auto enqueue = [&](const float* rhs, hipStream_t stream) {   // the n dependent launches of this solve
        double rho = 1.0 / sigma1;
        for (int it = 0; it < n; ++it) {
            double c1 = 0.0, c2 = 1.0 / theta;
            if (it > 0) {
                const double rho_new = 1.0 / (2.0 * sigma1 - rho);
                c1 = rho_new * rho;
                c2 = 2.0 * rho_new / delta;
                rho = rho_new;
            }
            dispatch_phase(s, k, uni ? 11 : 9, rhs, (it & 1) ? Z : Y, (it & 1) ? Y : Z, c1, c2, it, g, stream);
        }
    };
    if (s->profile) LS_HIP(hipEventRecord(s->pev[0], st));
    if (s->use_graph && n > 0) {
        // One hipGraph per (k, n, flavour, geometry): all pointers inside are handle-owned, so b is staged once
        // (a 4kV-byte copy, <0.2 % of a solve). Replaying removes the per-kernel host launch cost that dominates
        // small meshes (a 70k-vertex step is ~2 us of GPU time against ~4 us of eager launch).
        const uint64_t key = ((uint64_t)n << 32) | ((uint64_t)g.G << 12) | ((uint64_t)(g.bs >> 8) << 8) | ((uint64_t)k << 4) | (uni ? 1u : 0u);
        auto found = s->graphs.find(key);
        if (found == s->graphs.end()) {
            if (s->graphs.size() >= 16) {           // warm starts produce many different n: keep the cache bounded
                for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
                s->graphs.clear();
            }
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            LS_HIP(hipStreamBeginCapture(s->cap, hipStreamCaptureModeThreadLocal));
            enqueue(s->bh, s->cap);
            const hipError_t e_end = hipStreamEndCapture(s->cap, &graph);
            if (e_end != hipSuccess || !graph) return hip_fail(e_end != hipSuccess ? e_end : hipErrorUnknown, "hipStreamEndCapture", __FILE__, __LINE__);
            const hipError_t e_inst = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e_inst != hipSuccess) return hip_fail(e_inst, "hipGraphInstantiate", __FILE__, __LINE__);
            found = s->graphs.emplace(key, exec).first;
        }
        LS_HIP(hipMemcpyAsync(s->bh, b, bytes, hipMemcpyDeviceToDevice, st));
        LS_HIP(hipGraphLaunch(found->second, st));
    } else {
        enqueue(b, st);
    }
    LS_HIP(hipMemcpyAsync(x, (n & 1) ? Z : Y, bytes, hipMemcpyDeviceToDevice, st));
    }
    if (s->profile) LS_HIP(hipEventRecord(s->pev[1], st));
    LS_HIP(hipGetLastError());
    // true residual of the returned iterate
    if ((rc = dispatch_phase(s, k, 10, b, x, x, rtol, atol, 0, g, st))) return rc;
    dispatch_phase(s, k, 1, b, x, x, rtol, atol, 0, g, st);
    LS_HIP(hipMemcpyAsync(&s->h_scal[2], s->scal, sizeof(Scal), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    const Scal& f = s->h_scal[2];
    if (s->profile) {   // n back-to-back launches between two events: average includes the launch gaps
        float ms = 0.f;
        LS_HIP(hipEventElapsedTime(&ms, s->pev[0], s->pev[1]));
        s->prof_ms[0] = ms; s->prof_ms[1] = s->prof_ms[2] = 0.0;
        s->prof_iters = n;
    }
    bool ok = !capped && f.bad == 0;
    if (info) {
        info->iterations = n;
        for (int q = 0; q < 4; ++q) { info->rnorm[q] = q < k ? sqrt(f.rz[0][q]) : 0.0; info->bnorm[q] = q < k ? sqrt(f.bb[q]) : 0.0; }
    }
    // Acceptance of the result: the a-priori count guarantees the reduction IF the enclosure holds; what can be
    // checked a posteriori in fp32 is that the TRUE residual is at its backward-stable level, ||b - M x|| <=
    // max(request, 8 eps32 ||M||_2 ||x||_2) with ||M||_2 <= lmax * max diag -- a violated enclosure (diverging
    // low modes) fails this test and the caller falls back to PCG.
    const double mnorm = lmax * s->dmax;
    for (int q = 0; q < k; ++q) {
        const double rr = f.rz[0][q], xx = f.rr[q];
        const double floor2 = 64.0 * 3.6e-15 * mnorm * mnorm * xx;          // (8 * 2^-24)^2 = 2.3e-13 -> 64 * eps^2
        if (!(rr <= std::max(f.thr2[q], floor2))) ok = false;
    }
    if (info) info->converged = ok ? 1 : 0;
    if (!ok) {
        set_error(capped ? "Chebyshev: %d iterations needed for the requested reduction exceed max_iter"
                         : "Chebyshev: residual check failed after %d iterations (spectral enclosure violated?)", n);
        return LS_E_NOT_CONVERGED;
    }
    return LS_OK;
}

}  // namespace

extern "C" int ls_solver_create_ext(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int64_t n_cols,
                                    int64_t nnz, int kmax, int device, void* stream, ls_solver** h_out) {
    LS_REQUIRE(h_out, LS_E_INVALID, "ls_solver_create: h_out is null");
    *h_out = nullptr;
    LS_REQUIRE(n_rows >= 0 && n_cols >= n_rows && nnz >= 0 && rowptr && (nnz == 0 || (col && val)), LS_E_INVALID,
               "ls_solver_create: null pointer, negative size or n_cols < n_rows");
    LS_REQUIRE(kmax >= 1 && kmax <= KMAX, LS_E_INVALID, "ls_solver_create: kmax=%d outside [1,%d]", kmax, KMAX);
    LS_REQUIRE(n_cols < (int64_t)2000000000 && nnz < (int64_t)2000000000, LS_E_OVERFLOW, "matrix too large for int32 indices");
    DeviceGuard g(device);
    LS_HIP(g.err);
    ls_solver* s = new (std::nothrow) ls_solver();
    LS_REQUIRE(s, LS_E_INVALID, "out of host memory");
    s->device = device; s->V = n_rows; s->ncols = n_cols; s->nnz = nnz; s->kmax = kmax;
    s->csr = CsrView{rowptr, col, val};
    const int rc = create_impl(s, (hipStream_t)stream);
    if (rc) { free_solver(s); return rc; }
    *h_out = s;
    return LS_OK;
}

extern "C" int ls_solver_create(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V, int64_t nnz, int kmax,
                                int device, void* stream, ls_solver** h_out) {
    return ls_solver_create_ext(rowptr, col, val, V, V, nnz, kmax, device, stream, h_out);
}

extern "C" int ls_solver_destroy(ls_solver* s) {
    if (!s) return LS_OK;
    DeviceGuard g(s->device);
    free_solver(s);
    return LS_OK;
}

extern "C" int ls_solver_set(ls_solver* s, const char* name, int value) {
    LS_REQUIRE(s && name, LS_E_INVALID, "ls_solver_set: null argument");
    if (!strcmp(name, "check_every")) { LS_REQUIRE(value >= 1 && value <= 4096, LS_E_INVALID, "check_every outside [1,4096]"); s->check_every = value; }
    else if (!strcmp(name, "profile")) { s->profile = value ? 1 : 0; }
    else if (!strcmp(name, "graph")) { s->use_graph = value ? 1 : 0; }
    else if (!strcmp(name, "patch")) { s->use_patch = value ? 1 : 0; }
    else if (!strcmp(name, "grid")) { LS_REQUIRE(value >= 0 && value <= MAXG, LS_E_INVALID, "grid outside [0,%d]", MAXG); s->grid = value; }
    else if (!strcmp(name, "block")) { LS_REQUIRE(value == 0 || value == 256 || value == 512 || value == 1024, LS_E_INVALID, "block must be 0 (auto), 256, 512 or 1024"); s->block = value; }
    else { set_error("ls_solver_set: unknown knob '%s'", name); return LS_E_INVALID; }
    return LS_OK;
}

extern "C" int ls_solver_set_spectrum(ls_solver* s, double a_min) {
    LS_REQUIRE(s && a_min >= 0.0, LS_E_INVALID, "ls_solver_set_spectrum: need a handle and a_min >= 0");
    s->a_min = a_min;
    return LS_OK;
}

extern "C" int ls_solver_set_uniform(ls_solver* s, float a, float b, void* stream) {
    LS_REQUIRE(s, LS_E_INVALID, "ls_solver_set_uniform: null handle");
    LS_REQUIRE(s->ncols == s->V, LS_E_STATE, "ls_solver_set_uniform: square systems only");
    if (s->uni || s->V == 0) return LS_OK;
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const int64_t V = s->V;
    const int S = div_up(V, WAVE);
    int rc;
    int* width64 = nullptr;
    if ((rc = dev_alloc(s, &s->slice_ptr_u, (size_t)S + 1))) return rc;
    LS_HIP(hipMalloc((void**)&width64, sizeof(int) * (size_t)S));
    hipLaunchKernelGGL(k_sell_widths, dim3(div_up((int64_t)S * WAVE, BLOCK)), dim3(BLOCK), 0, st, s->csr.rowptr, V, S, width64, 1);
    hipLaunchKernelGGL(k_sell_scan, dim3(1), dim3(1024), 0, st, width64, S, s->slice_ptr_u);
    int total = 0;
    LS_HIP(hipMemcpyAsync(&total, s->slice_ptr_u + S, sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    (void)hipFree(width64);
    LS_REQUIRE(total >= 0, LS_E_OVERFLOW, "SELL copy of the matrix overflows int32 entry offsets");
    if ((rc = dev_alloc(s, &s->cols_u, (size_t)std::max(total, 1)))) return rc;
    hipLaunchKernelGGL(k_sell_cols_fill, dim3(div_up((int64_t)S * WAVE, BLOCK)), dim3(BLOCK), 0, st, s->csr, V, s->slice_ptr_u, s->cols_u);
    LS_HIP(hipGetLastError());
    (void)a;
    s->uni_offdiag = -b;            // M_ij = fl(b * -1) for every edge (geometry.py:86-94 + :128/:132)
    s->uni = true;
    return LS_OK;
}

extern "C" int ls_solver_set_patches(ls_solver* s, const int32_t* h_table, int n_patches, const int32_t* h_ghost_gid, int64_t n_gid,
                                     const uint16_t* h_cols16, int64_t n_cols, const float* h_diag, int64_t n_diag,
                                     const int32_t* h_perm, int depth, int max_local, int max_rows, void* stream) {
    LS_REQUIRE(s && h_table && h_perm && n_patches > 0 && depth >= 1 && depth <= PATCH_MAX_DEPTH, LS_E_INVALID, "ls_solver_set_patches: bad argument");
    LS_REQUIRE(s->ncols == s->V && s->uni, LS_E_STATE, "ls_solver_set_patches: needs a square system declared uniform (ls_solver_set_uniform)");
    LS_REQUIRE(max_rows <= 1024 * PATCH_RPT_MAX && max_local < 65535, LS_E_INVALID, "ls_solver_set_patches: patch too large (rows %d, local %d)", max_rows, max_local);
    const size_t lds = 2 * (size_t)(max_local + 1) * sizeof(float);     // per right-hand-side column
    LS_REQUIRE(lds <= 160 * 1024, LS_E_INVALID, "ls_solver_set_patches: %zu bytes of LDS per patch and column exceed 160 KiB", lds);
    if (s->patch.n) return LS_OK;
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    auto up = [&](auto** dst, const auto* src, size_t n) -> int {
        if ((rc = dev_alloc(s, dst, std::max<size_t>(n, 1)))) return rc;
        if (n) LS_HIP(hipMemcpyAsync(*dst, src, n * sizeof(**dst), hipMemcpyHostToDevice, st));
        return LS_OK;
    };
    if ((rc = up(&s->patch.table, h_table, (size_t)n_patches * PATCH_TABLE_COLS))) return rc;
    if ((rc = up(&s->patch.gid, h_ghost_gid, (size_t)n_gid))) return rc;
    if ((rc = up(&s->patch.cols, h_cols16, (size_t)n_cols))) return rc;
    if ((rc = up(&s->patch.diag, h_diag, (size_t)n_diag))) return rc;
    if ((rc = up(&s->patch.perm, h_perm, (size_t)s->V))) return rc;
    const size_t vk = (size_t)s->V * s->kmax;
    if ((rc = dev_alloc(s, &s->patch.bn, vk))) return rc;
    for (auto& q : s->patch.it) if ((rc = dev_alloc(s, &q, vk))) return rc;
    // kernels with more than 64 KiB of dynamic LDS need the opt-in
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<1, 512, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<1, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<1, 1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<1, 1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<2, 512, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<2, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<2, 1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<2, 1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<3, 512, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<3, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<3, 1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<3, 1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<4, 512, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<4, 512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<4, 1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_patch_cheb<4, 1024, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    LS_HIP(hipStreamSynchronize(st));     // the host arrays may go away after return
    s->patch.depth = depth;
    s->patch.bs = max_rows > 512 * PATCH_RPT_MAX ? 1024 : 512;
    s->patch.rpt = div_up(max_rows, s->patch.bs);
    s->patch.cap1 = max_local + 1;
    s->patch.n = n_patches;
    return LS_OK;
}

extern "C" int ls_solver_chebyshev_iterations(const ls_solver* s, double reduction, int* h_n) {
    LS_REQUIRE(s && h_n && reduction > 0.0, LS_E_INVALID, "ls_solver_chebyshev_iterations: bad argument");
    LS_REQUIRE(s->a_min > 0.0 && s->gersh > 0.0 && s->dmax > 0.0, LS_E_STATE, "Chebyshev: no certified spectral enclosure for this matrix");
    const double lmin = 0.98 * s->a_min / s->dmax, lmax = s->gersh * (1.0 + 1e-5);
    const double sk = sqrt(lmax / lmin), rate = (sk - 1.0) / (sk + 1.0);
    *h_n = reduction >= 1.0 ? 0 : (int)std::min(2.0e9, ceil(log(2.0 / reduction) / -log(rate)));
    return LS_OK;
}

extern "C" int ls_solver_spectrum(const ls_solver* s, double* h_lmin, double* h_lmax) {
    LS_REQUIRE(s && h_lmin && h_lmax, LS_E_INVALID, "ls_solver_spectrum: null argument");
    *h_lmin = s->dmax > 0.0 ? s->a_min / s->dmax : 0.0;
    *h_lmax = s->gersh;
    return LS_OK;
}

extern "C" int ls_solver_solve_chebyshev(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol, double atol,
                                         int max_iter, ls_solve_info* h_info, void* stream) {
    LS_REQUIRE(s, LS_E_INVALID, "ls_solver_solve_chebyshev: null handle");
    LS_REQUIRE(k >= 1 && k <= s->kmax, LS_E_INVALID, "ls_solver_solve_chebyshev: k=%d outside [1,%d]", k, s->kmax);
    LS_REQUIRE(s->V == 0 || (b && x), LS_E_INVALID, "ls_solver_solve_chebyshev: null pointer");
    LS_REQUIRE(x != b, LS_E_INVALID, "ls_solver_solve_chebyshev: x must not alias b");
    LS_REQUIRE(s->ncols == s->V, LS_E_STATE, "ls_solver_solve_chebyshev: square systems only");
    LS_REQUIRE(rtol >= 0.0 && atol >= 0.0 && (rtol > 0.0 || atol > 0.0) && max_iter >= 0, LS_E_INVALID,
               "ls_solver_solve_chebyshev: need rtol, atol >= 0 (one of them > 0) and max_iter >= 0");
    if (h_info) memset(h_info, 0, sizeof(*h_info));
    if (s->V == 0) { if (h_info) h_info->converged = 1; return LS_OK; }
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    return solve_cheb(s, b, x0, x, k, rtol, atol, max_iter, h_info, (hipStream_t)stream);
}

extern "C" int ls_solver_profile(const ls_solver* s, double* h_ms3, int* h_iters) {
    LS_REQUIRE(s && h_ms3 && h_iters, LS_E_INVALID, "ls_solver_profile: null argument");
    for (int i = 0; i < 3; ++i) h_ms3[i] = s->prof_ms[i];
    *h_iters = s->prof_iters;
    return LS_OK;
}

extern "C" int ls_solver_workspace_bytes(const ls_solver* s, size_t* h_bytes) {
    LS_REQUIRE(s && h_bytes, LS_E_INVALID, "ls_solver_workspace_bytes: null argument");
    *h_bytes = s->bytes;
    return LS_OK;
}

extern "C" int ls_solver_solve(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol, double atol,
                               int max_iter, ls_solve_info* h_info, void* stream) {
    LS_REQUIRE(s, LS_E_INVALID, "ls_solver_solve: null handle");
    LS_REQUIRE(k >= 1 && k <= s->kmax, LS_E_INVALID, "ls_solver_solve: k=%d outside [1,%d]", k, s->kmax);
    LS_REQUIRE(s->V == 0 || (b && x), LS_E_INVALID, "ls_solver_solve: null pointer");
    LS_REQUIRE(x != b, LS_E_INVALID, "ls_solver_solve: x must not alias b");
    LS_REQUIRE(s->ncols == s->V, LS_E_STATE, "ls_solver_solve: handle is a shard (n_cols > n_rows): drive it with ls_solver_phase");
    LS_REQUIRE(rtol >= 0.0 && atol >= 0.0 && (rtol > 0.0 || atol > 0.0) && max_iter >= 0, LS_E_INVALID,
               "ls_solver_solve: need rtol, atol >= 0 (one of them > 0) and max_iter >= 0");
    if (h_info) memset(h_info, 0, sizeof(*h_info));
    if (s->V == 0) { if (h_info) h_info->converged = 1; return LS_OK; }
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    return solve_impl(s, b, x0, x, k, rtol, atol, max_iter, h_info, (hipStream_t)stream);
}

// ---- one kernel at a time (sharded driver) ----------------------------------------------------------
extern "C" int ls_solver_phase(ls_solver* s, int phase, const float* b, float* x, int k, double rtol, double atol, int it,
                               void* stream) {
    LS_REQUIRE(s, LS_E_INVALID, "ls_solver_phase: null handle");
    LS_REQUIRE(phase >= 0 && phase <= 4, LS_E_INVALID, "ls_solver_phase: phase %d outside [0,4]", phase);
    LS_REQUIRE(k >= 1 && k <= s->kmax, LS_E_INVALID, "ls_solver_phase: k=%d outside [1,%d]", k, s->kmax);
    LS_REQUIRE((phase != 0 || (b && x)) && (phase != 3 || x), LS_E_INVALID, "ls_solver_phase: null pointer");
    if (s->V == 0) return LS_OK;
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    const int rc = dispatch_phase(s, k, phase, b, nullptr, x, rtol, atol, it, geometry(s), (hipStream_t)stream);
    if (rc) return rc;
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_solver_buffers(ls_solver* s, float** h_p, double** h_part, int* h_grid, int* h_part_stride) {
    LS_REQUIRE(s && h_p && h_part && h_grid && h_part_stride, LS_E_INVALID, "ls_solver_buffers: null argument");
    *h_p = s->p;
    *h_part = s->part;
    *h_grid = geometry(s).G;
    *h_part_stride = MAXG;
    return LS_OK;
}

extern "C" int ls_solver_sell(ls_solver* s, const int32_t** h_slice_ptr, const void** h_cv, int64_t* h_entries) {
    LS_REQUIRE(s && h_slice_ptr && h_cv && h_entries, LS_E_INVALID, "ls_solver_sell: null argument");
    *h_slice_ptr = s->slice_ptr; *h_cv = s->sell_cv; *h_entries = s->sell_entries;
    return LS_OK;
}

// ---- sharded Chebyshev: `nsteps` iterations on the first n_rows rows of the shard (owned + computed ghost layers) ----
extern "C" int ls_shard_cheb_steps(ls_solver* s, const float* b, float* xa, float* xb, int k, int it0, int nsteps,
                                   const float* h_c1, const float* h_c2, int64_t n_rows, void* stream) {
    LS_REQUIRE(s && b && xa && xb && h_c1 && h_c2, LS_E_INVALID, "ls_shard_cheb_steps: null argument");
    LS_REQUIRE(k >= 1 && k <= s->kmax && it0 >= 0 && nsteps >= 0 && n_rows >= 0 && n_rows <= s->V, LS_E_INVALID,
               "ls_shard_cheb_steps: k, it0, nsteps or n_rows out of range");
    if (n_rows == 0 || nsteps == 0) return LS_OK;
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    Geometry geo = geometry(s);
    geo.T = div_up(n_rows, geo.bs);
    const int64_t V_saved = s->V;
    s->V = n_rows;                                   // the launchers read the row count from the handle
    for (int j = 0; j < nsteps; ++j) {
        const int it = it0 + j;
        dispatch_phase(s, k, 9, b, (it & 1) ? xb : xa, (it & 1) ? xa : xb, (double)h_c1[j], (double)h_c2[j], it, geo, (hipStream_t)stream);
    }
    s->V = V_saved;
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// partials of ||b - M x||^2 (slots 1,2) and ||b||^2 (slot 3) over the first n_rows rows
extern "C" int ls_shard_resnorm(ls_solver* s, const float* b, const float* x, int k, int64_t n_rows, void* stream) {
    LS_REQUIRE(s && b && x, LS_E_INVALID, "ls_shard_resnorm: null argument");
    LS_REQUIRE(k >= 1 && k <= s->kmax && n_rows >= 0 && n_rows <= s->V, LS_E_INVALID, "ls_shard_resnorm: k or n_rows out of range");
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    Geometry geo = geometry(s);
    geo.T = div_up(n_rows, geo.bs);
    const int64_t V_saved = s->V;
    s->V = n_rows;
    const int rc = dispatch_phase(s, k, 10, b, x, nullptr, 0.0, 0.0, 0, geo, (hipStream_t)stream);
    s->V = V_saved;
    if (rc) return rc;
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_solver_bind(ls_solver* s, float* p_ext, double* part) {
    LS_REQUIRE(s && p_ext && part, LS_E_INVALID, "ls_solver_bind: null argument");
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    if (s->own_p) (void)hipFree(s->p);
    if (s->own_part) (void)hipFree(s->part);
    s->p = p_ext; s->part = part;
    s->own_p = s->own_part = false;
    return LS_OK;
}

extern "C" int ls_solver_poll(ls_solver* s, int k, int n_enqueued, ls_solve_info* h_info, void* stream) {
    LS_REQUIRE(s && h_info, LS_E_INVALID, "ls_solver_poll: null argument");
    memset(h_info, 0, sizeof(*h_info));
    DeviceGuard g(s->device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    LS_HIP(hipMemcpyAsync(&s->h_scal[2], s->scal, sizeof(Scal), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    bool conv;
    int iters;
    fill_info(s->h_scal[2], k, n_enqueued, h_info, &conv, &iters);
    if (s->h_scal[2].stop_iter == INT_MAX) h_info->iterations = -1;   // still running
    if (s->h_scal[2].bad) { set_error("PCG: breakdown (code %d) after %d iterations", s->h_scal[2].bad, iters); return LS_E_NOT_CONVERGED; }
    return LS_OK;
}

extern "C" int ls_gather_rows(const float* src, const int32_t* idx, int64_t n, int k, float* dst, int device, void* stream) {
    LS_REQUIRE(n >= 0 && (n == 0 || (src && idx && dst)), LS_E_INVALID, "ls_gather_rows: null pointer or negative size");
    LS_REQUIRE(k >= 1 && k <= KMAX, LS_E_INVALID, "ls_gather_rows: k=%d outside [1,%d]", k, KMAX);
    if (n == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    const dim3 grid(div_up(n, BLOCK)), block(BLOCK);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL(k_gather_rows<1>, grid, block, 0, st, src, idx, n, dst); break;
        case 2: hipLaunchKernelGGL(k_gather_rows<2>, grid, block, 0, st, src, idx, n, dst); break;
        case 3: hipLaunchKernelGGL(k_gather_rows<3>, grid, block, 0, st, src, idx, n, dst); break;
        default: hipLaunchKernelGGL(k_gather_rows<4>, grid, block, 0, st, src, idx, n, dst); break;
    }
    LS_HIP(hipGetLastError());
    return LS_OK;
}
