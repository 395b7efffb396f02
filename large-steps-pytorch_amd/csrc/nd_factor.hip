// nd_factor.hip -- numeric multifrontal factorisation of the nested-dissection plan on the MI355X, and ls_direct_factor:
// matrix in, solver handle out, entirely behind the C ABI.
//
// Replaces the constructor of the reference's default solver (largesteps/solvers.py:26-34: cholespy / CHOLMOD analyse +
// factorise). Per tree level, deepest first, every node i with front F = [own | boundary] (fp64, s + b rows):
//     F      = A[front, front] (entries with a row or column in own_i)  +  the children's Schur complements (extend-add)
//     Finv   = F_ss^-1                       recursive 2 x 2 Schur inversion: only matrix products + a <= 64 x 64 in-LDS inverse
//     W      = F_bs Finv
//     U      = F_bb - W F_sb                 stays in the front's storage for the parent
// All nodes of a level go through the same launches (batched: grid.y = node). The fp64 results are written once, as fp32, in
// the layouts the solve kernels read (csrc/direct.hip: finv / wf / wb; csrc/nd_tier.h: quad-interleaved u4 / d4, packed
// triangles + sparse blocks of the leaves). Everything dense here is hand-written: a tiled fp64 batched GEMM with per-batch
// descriptors, the small SPD inverse, assembly / extend-add / conversion kernels. No rocSOLVER / rocBLAS / torch.
#include "common.h"
#include "nd_plan.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <string.h>
#include <string>
#include <thread>
#include <sys/resource.h>
#include <vector>

#ifdef LS_ND_EXPERIMENTS
#include "experiments/round_study.h"
#endif
namespace ls {

struct SpEnt { float val; int idx; };      // same layout as csrc/nd_tier.h

// ---- batched fp64 GEMM: C = alpha * op(A) op(B) + beta * C, row-major, per-batch descriptor ------------------------------------
struct GemmDesc {
    const double* A; const double* B; double* C;
    int M, N, K, lda, ldb, ldc, ta, tb;          // ta / tb: the operand is stored transposed (op(A)[m][k] = A[k * lda + m])
    double alpha, beta;
    int sym, pad;                                // sym: the product (and C) is symmetric, M == N: only the tiles on and below the diagonal are
};                                               // computed, the ones below are stored twice (C[m][n] and C[n][m]) -- half the work of the Schur updates

#ifndef LS_GEMM_GK
#define LS_GEMM_GK 16          // (32 was measured too, with the prefetch below: 5.3 against 5.0 ms of products per 1M construction, and 3x slower at 4M)
#endif
constexpr int GT = 64, GK = LS_GEMM_GK;          // 64 x 64 output tile per workgroup, GK-deep slices

typedef double f64x4 __attribute__((ext_vector_type(4)));

// The products run on the fp64 matrix instruction (v_mfma_f64_16x16x4_f64: D(16 x 16) += A(16 x 4) B(4 x 16), one double of A and
// of B per lane: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; D: 4 doubles per lane, register v = row (l >> 4) + 4 v,
// column l & 15 -- the f64 form has its own row map, cdna_hip_programming.md "Fragment layout"). A wave owns a 32 x 32 quadrant of
// the workgroup's tile = 2 x 2 instruction tiles: per 4-deep step four 8-byte LDS reads feed four instructions (4096 fused
// multiply-adds). The scalar version (a 4 x 4 block per thread, 8 LDS reads per 16 multiply-adds) was bound by LDS bandwidth at
// half the fp64 rate and reached 11-16 TFLOP/s.
// Round 4: the NEXT slice is requested into registers before the current one is multiplied (the top levels' products are single
// matrices of a few hundred tiles -- one workgroup per CU: with the loads of a slice issued, waited for and only then multiplied,
// every 16-deep slice cost a memory round trip): 9.1 -> 5.0 ms of products per construction at 1M vertices (118 launches; the largest
// 366 -> 276 us), constructor 0.039-0.042 -> 0.030-0.032 s (profiles/r04_gemm_prefetch.txt).
// TT = 64: the tile above. TT = 32 (a wave owns a 16 x 16 quadrant, ONE instruction per 4-deep step): for launches of a few tiles -- the 2 x 2
// Schur recursion of the top levels multiplies blocks of 125-500 rows one launch after the other, 4-64 tiles of 64 x 64 on 256 CUs; with
// quarter tiles four times as many workgroups share the work and a launch takes a third of the time (same sums in the same order: bit-identical).
template <int TT>
__global__ __launch_bounds__(256) void k_gemm_batched(const GemmDesc* __restrict__ descs) {
    constexpr int GT = TT, MI = TT / 32;                           // (shadows the 64 of the host side) instruction tiles per quadrant side
    const GemmDesc d = descs[blockIdx.y];
    const int tiles_n = (d.N + GT - 1) / GT, tiles_m = (d.M + GT - 1) / GT;
    if ((int)blockIdx.x >= tiles_m * tiles_n || d.M <= 0 || d.N <= 0) return;
    const int tm = (blockIdx.x / tiles_n) * GT, tn = (blockIdx.x % tiles_n) * GT;
    if (d.sym && tn > tm) return;
    const bool mirror = d.sym && tn < tm;
    __shared__ double sa[GK][GT + 1], sb[GK][GT + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (GT / 2), wn = (wave & 1) * (GT / 2);        // this wave's quadrant
    const int l15 = lane & 15, l4 = lane >> 4;
    f64x4 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    constexpr int PER = GK * GT / 256;                             // elements of either slice per thread
    double ra[PER], rb[PER];
    // element e of a slice: A (m, k), B (k, n); the index order follows the storage order so that consecutive threads read consecutive memory
    auto fetch = [&](int k0) {
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int e = threadIdx.x + t * 256;
            int m, k;
            if (d.ta) { m = e % GT; k = e / GT; } else { k = e % GK; m = e / GK; }
            const int gm = tm + m, gk = k0 + k;
            ra[t] = (gm < d.M && gk < d.K) ? (d.ta ? d.A[(size_t)gk * d.lda + gm] : d.A[(size_t)gm * d.lda + gk]) : 0.0;
            int n, kk;
            if (d.tb) { kk = e % GK; n = e / GK; } else { n = e % GT; kk = e / GT; }
            const int gn = tn + n, gk2 = k0 + kk;
            rb[t] = (gn < d.N && gk2 < d.K) ? (d.tb ? d.B[(size_t)gn * d.ldb + gk2] : d.B[(size_t)gk2 * d.ldb + gn]) : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < d.K; k0 += GK) {
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int e = threadIdx.x + t * 256;
            int m, k;
            if (d.ta) { m = e % GT; k = e / GT; } else { k = e % GK; m = e / GK; }
            sa[k][m] = ra[t];
            int n, kk;
            if (d.tb) { kk = e % GK; n = e / GK; } else { n = e % GT; kk = e / GT; }
            sb[kk][n] = rb[t];
        }
        __syncthreads();
        if (k0 + GK < d.K) fetch(k0 + GK);                          // in flight while this slice is multiplied
#pragma unroll
        for (int k4 = 0; k4 < GK; k4 += 4) {
            double av[MI], bv[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) { av[i] = sa[k4 + l4][wm + 16 * i + l15]; bv[i] = sb[k4 + l4][wn + 16 * i + l15]; }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gm = tm + wm + 16 * i + l4 + 4 * v;
            if (gm >= d.M) continue;
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int gn = tn + wn + 16 * j + l15;
                if (gn >= d.N) continue;
                double* c = d.C + (size_t)gm * d.ldc + gn;
                const double r = d.alpha * acc[i][j][v] + (d.beta != 0.0 ? d.beta * *c : 0.0);
                *c = r;
                if (mirror) d.C[(size_t)gn * d.ldc + gm] = r;
            }
        }
    }
}

// ---- X = M^-1 for SPD M, n <= 128, one workgroup per matrix, the matrix in REGISTERS -------------------------------------------------
struct InvDesc { const double* M; double* X; int n, ldm, ldx, pad; };

constexpr int INV_N = 128;

// Gauss-Jordan sweeps in place (no pivoting: the pivots of an SPD matrix are the diagonal of its Schur complements, all positive).
// G x G threads, thread (by, bx) owns the RB x RB block of rows by * RB.. and columns bx * RB.. (RB = NB / G: 4 x 4 doubles in both
// forms: 16 x 16 threads for NB = 64, 32 x 32 for NB = 128), rows and columns >= n padded with zeros (they stay zero). Step k needs row k and column
// k of the current matrix: their owners publish them to LDS at the end of step k - 1 (two buffers, by parity), so a step is ONE
// barrier, 2 RB + 1 LDS reads and RB^2 fused multiply-adds per thread out of registers. The loop over k is blocked by RB with the
// inner RB steps unrolled: which REGISTER holds row / column k is then known at compile time, only the owning thread is a run-time
// test. The upper levels of the tree run these one workgroup at a time between the products of the 2 x 2 Schur recursion
// (inverse_rec): the chain of launches, not the arithmetic, is what the factorisation waits for there -- the version with the matrix
// in LDS took ~110 us per block (every element read and written through LDS in every step), and its 64-row limit meant one more
// level of recursion (twice the chain).
// Round 4: the 128-row form of the CHAIN (launches of up to 64 blocks) runs 32 x 32 threads with 4 x 4 doubles each instead of 16 x 16 with 8 x 8: a lone wave per SIMD issues the
// straight-line code of a step at ~12 cycles per instruction (tools/ubench/fp64_rate.hip, spd_inverse.hip: it waits for its instruction
// fetches, not for the fp64 pipe), four waves per SIMD hide that: 80 -> 69 us per block, the same sums in the same order.
template <int NB, int G>
__global__ __launch_bounds__(G * G) void k_spd_inverse_reg(const InvDesc* __restrict__ descs, int* __restrict__ flag) {
    constexpr int RB = NB / G;
    static_assert(RB % 2 == 0, "the parity of k is the parity of kk");
    const InvDesc d = descs[blockIdx.x];
    const int n = d.n;
    if (n <= 0) return;
    __shared__ double rowb[2][NB], colb[2][NB], ipb[2];
    const int bx = threadIdx.x % G, by = threadIdx.x / G;
    double a[RB][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < RB; ++c) {
            const int i = by * RB + r, j = bx * RB + c;
            a[r][c] = (i < n && j < n) ? d.M[(size_t)i * d.ldm + j] : 0.0;
        }
    }
    if (by == 0) {
#pragma unroll
        for (int c = 0; c < RB; ++c) rowb[0][bx * RB + c] = a[0][c];
    }
    if (bx == 0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) colb[0][by * RB + r] = a[r][0];
    }
    bool bad = false;
    // 1 / pivot is formed ONCE, by the thread that owns the pivot, and published with the row and the column (round 5: as every
    // thread's own fp64 division -- ~35 instructions at a quarter of the fp32 rate, four waves per SIMD -- it was half of a step)
    if (bx == 0 && by == 0) {
        double p = a[0][0];
        if (!(p > 0.0)) { bad = true; p = 1.0; }
        ipb[0] = 1.0 / p;
    }
    __syncthreads();
    for (int kb = 0; kb * RB < n; ++kb) {
#pragma unroll
        for (int kk = 0; kk < RB; ++kk) {
            const int k = kb * RB + kk;
            if (k < n) {                                      // uniform over the workgroup
                const int cur = kk & 1, nxt = cur ^ 1;        // RB is even: the parity of k is the parity of kk
                const double ip = ipb[cur];
                double rr[RB], ck[RB];
#pragma unroll
                for (int c = 0; c < RB; ++c) rr[c] = rowb[cur][bx * RB + c] * ip;
#pragma unroll
                for (int r = 0; r < RB; ++r) ck[r] = colb[cur][by * RB + r];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
#pragma unroll
                    for (int c = 0; c < RB; ++c) a[r][c] = fma(-ck[r], rr[c], a[r][c]);
                }
                if (bx == kb) {                               // column k: -colk[i] / p
#pragma unroll
                    for (int r = 0; r < RB; ++r) a[r][kk] = -ck[r] * ip;
                }
                if (by == kb) {                               // row k: rowk[j] / p, the pivot itself 1 / p
#pragma unroll
                    for (int c = 0; c < RB; ++c) a[kk][c] = rr[c];
                    if (bx == kb) a[kk][kk] = ip;
                }
                // row / column k + 1 of the updated matrix for the next step
                const int kn = (kk + 1) % RB;                 // a compile-time register index once the kk loop is unrolled
                const int ob = kk + 1 < RB ? kb : kb + 1;
                if (by == ob) {
#pragma unroll
                    for (int c = 0; c < RB; ++c) rowb[nxt][bx * RB + c] = a[kn][c];
                }
                if (bx == ob) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) colb[nxt][by * RB + r] = a[r][kn];
                    if (by == ob && k + 1 < n) {               // the next pivot and its reciprocal
                        double p = a[kn][kn];
                        if (!(p > 0.0)) { bad = true; p = 1.0; }
                        ipb[nxt] = 1.0 / p;
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int c = 0; c < RB; ++c) {
            const int i = by * RB + r, j = bx * RB + c;
            if (i < n && j < n) d.X[(size_t)i * d.ldx + j] = a[r][c];
        }
    }
    if (bad) atomicExch(flag, 1);
}

// ---- assembly -------------------------------------------------------------------------------------------------------------------
struct FactorNode {              // device copy of what the assembly / conversion kernels need per node
    int s, b, own_start, parent;
    long long bnd_off, f_off, x_off, w_off;      // offsets into bnd / fronts (fp64) / Finv storage (fp64) / W storage (fp64)
    long long o_finv, o_w, o_tri;                // fp32 output offsets (plain: finv, wf/wb; quad: d4, u4; sparse leaf: tri)
    int layout, pad;                             // 0 plain, 1 quad, 2 sparse leaf
};

// every stored matrix entry whose row lives in node n and whose column is in the front of n
__global__ void k_assemble(int64_t nnz, const int* __restrict__ rowidx, const int* __restrict__ col, const float* __restrict__ val,
                           const int* __restrict__ inv, const int* __restrict__ node_of_new, const FactorNode* __restrict__ nodes,
                           const int* __restrict__ bnd, double* __restrict__ fronts, int* __restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int r = inv[rowidx[e]], c = inv[col[e]];
    const int n = node_of_new[r];
    const FactorNode nd = nodes[n];
    if (c < nd.own_start) return;                          // column in a descendant: the mirrored entry is assembled there
    const int m = nd.s + nd.b, rl = r - nd.own_start;
    double* F = fronts + nd.f_off;
    const double v = (double)val[e];
    if (c < nd.own_start + nd.s) { F[(size_t)rl * m + (c - nd.own_start)] = v; return; }
    const int* B = bnd + nd.bnd_off;
    int lo = 0, hi = nd.b;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (B[mid] < c) lo = mid + 1; else hi = mid; }
    if (lo >= nd.b || B[lo] != c) { atomicExch(flag, 2); return; }
    F[(size_t)rl * m + nd.s + lo] = v;
    F[(size_t)(nd.s + lo) * m + rl] = v;
}

// row index of every stored entry (the caller hands over CSR; the per-entry kernels want COO rows)
__global__ void k_expand_rows(int64_t V, const int* __restrict__ rowptr, int* __restrict__ rowidx) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) rowidx[p] = (int)v;
}

// ---- sparse leaves: the off-diagonal block A_bs of every leaf as two CSR lists (by boundary row, by own row) ---------------------
// One thread per stored matrix entry (row in a sparse-layout leaf, column in its boundary). MODE 0 counts the entries of the
// two rows an entry belongs to, MODE 1 drops it into both lists (slot order by atomic cursor -- k_leaf_sort then orders every
// row by its index field, which is unique within a row: the result does not depend on the order of arrival).
// Own rows of the leaves are the tree's numbering [0, rows_s) (deepest level first); their boundary rows are
// bnd[bnd0 .. n_bnd).
template <int MODE>
__global__ void k_leaf_entries(int64_t nnz, const int* __restrict__ rowidx, const int* __restrict__ col, const float* __restrict__ val,
                               const int* __restrict__ inv, const int* __restrict__ node_of_new, const FactorNode* __restrict__ nodes,
                               const int* __restrict__ bnd, long long bnd0, int* __restrict__ cnt_s, int* __restrict__ cnt_b,
                               const int* __restrict__ ptr_s, const int* __restrict__ ptr_b, SpEnt* __restrict__ ent, int n_ent) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int r = inv[rowidx[e]];
    const FactorNode nd = nodes[node_of_new[r]];
    if (nd.layout != 2) return;
    const int c = inv[col[e]];
    if (c < nd.own_start + nd.s) return;
    const int* B = bnd + nd.bnd_off;
    int lo = 0, hi = nd.b;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (B[mid] < c) lo = mid + 1; else hi = mid; }
    if (lo >= nd.b || B[lo] != c) return;                  // (k_assemble reports the broken pattern)
    const int j = r - nd.own_start, rs = r;
    const long long rb = nd.bnd_off - bnd0 + lo;
    if (MODE == 0) { atomicAdd(cnt_s + rs, 1); atomicAdd(cnt_b + rb, 1); return; }
    const int pb = ptr_b[rb] + atomicAdd(cnt_b + rb, 1);
    ent[pb] = SpEnt{val[e], j};
    const int ps = n_ent + ptr_s[rs] + atomicAdd(cnt_s + rs, 1);
    ent[ps] = SpEnt{val[e], lo};
}

__global__ void k_leaf_sort(int64_t rows, const int* __restrict__ ptr, int base, SpEnt* __restrict__ ent) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    SpEnt* a = ent + base + ptr[row];
    const int n = ptr[row + 1] - ptr[row];
    for (int i = 1; i < n; ++i) {
        const SpEnt x = a[i];
        int k = i - 1;
        while (k >= 0 && a[k].idx > x.idx) { a[k + 1] = a[k]; --k; }
        a[k + 1] = x;
    }
}

// the pointer lists the tier kernels read: per leaf b + 1 entries for its boundary rows (at off_b[leaf]) and s + 1 for its own
// rows (at off_s[leaf]); values are absolute offsets into the entry array (boundary-row lists first, then own-row lists)
__global__ void k_leaf_ptrs(int first_leaf, const FactorNode* __restrict__ nodes, long long bnd0, const int* __restrict__ off_b,
                            const int* __restrict__ off_s, const int* __restrict__ ptr_b, const int* __restrict__ ptr_s, int n_ent,
                            int* __restrict__ sp_ptr) {
    const int leaf = blockIdx.x;
    if (off_b[leaf] < 0) return;
    const FactorNode nd = nodes[first_leaf + leaf];
    for (int r = threadIdx.x; r <= nd.b; r += blockDim.x) sp_ptr[off_b[leaf] + r] = ptr_b[nd.bnd_off - bnd0 + r];
    for (int r = threadIdx.x; r <= nd.s; r += blockDim.x) sp_ptr[off_s[leaf] + r] = n_ent + ptr_s[nd.own_start + r];
}

// parent front += Schur complement of the children with sibling index cix (one launch per cix: no two writers per entry)
__global__ void k_extend_add(const int* __restrict__ kids, int n_kids, const FactorNode* __restrict__ nodes, const int* __restrict__ ppos,
                             double* __restrict__ fronts) {
    const int ch = kids[blockIdx.y];
    const FactorNode c = nodes[ch];
    const FactorNode p = nodes[c.parent];
    const int b = c.b, mc = c.s + c.b, mp = p.s + p.b;
    const int* pp = ppos + c.bnd_off;
    const double* U = fronts + c.f_off + (size_t)c.s * mc + c.s;
    double* F = fronts + p.f_off;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)b * b; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / b), j = (int)(e % b);
        F[(size_t)pp[i] * mp + pp[j]] += U[(size_t)i * mc + j];
    }
}

#ifdef LS_ND_EXPERIMENTS
#define LS_EXP_ROUND(v, nd) exp_round((v), (nd).layout, (nd).pad)
#else
#define LS_EXP_ROUND(v, nd) (v)
#endif
// fp64 results -> the fp32 arrays of the solve kernels
__global__ void k_convert(const int* __restrict__ ids, const FactorNode* __restrict__ nodes, const double* __restrict__ xs,
                          const double* __restrict__ ws, float* __restrict__ finv, float* __restrict__ wf, float* __restrict__ wb,
                          float* __restrict__ u4, float* __restrict__ d4, float* __restrict__ tri) {
    const FactorNode nd = nodes[ids[blockIdx.y]];
    const int s = nd.s, b = nd.b;
    const double* X = xs + nd.x_off;
    const double* W = ws + nd.w_off;                         // (b, s) row-major
    const int s4 = (s + 3) & ~3;
    const int64_t total = (int64_t)s * s + (int64_t)b * s;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < (int64_t)s * s) {
            const int j = (int)(e / s), t = (int)(e % s);
            const float v = LS_EXP_ROUND((float)(0.5 * (X[(size_t)j * s + t] + X[(size_t)t * s + j])), nd);
            if (nd.layout == 0) finv[nd.o_finv + (size_t)t * s + j] = v;
            else if (nd.layout == 1) d4[nd.o_finv + ((size_t)(t >> 2) * s + j) * 4 + (t & 3)] = v;
            else if (t <= j) tri[nd.o_tri + (size_t)j * (j + 1) / 2 + t] = v;
        } else if (nd.layout != 2) {
            const int64_t f = e - (int64_t)s * s;
            const int i = (int)(f / s), j = (int)(f % s);
            const float v = LS_EXP_ROUND((float)W[(size_t)i * s + j], nd);
            if (nd.layout == 0) {
                wb[nd.o_w + (size_t)i * s + j] = v; wf[nd.o_w + (size_t)j * b + i] = v;
            }
            else {
                u4[nd.o_w + ((size_t)(j >> 2) * b + i) * 4 + (j & 3)] = v;
                const int t = s4 + i;
                d4[nd.o_finv + ((size_t)(t >> 2) * s + j) * 4 + (t & 3)] = v;
            }
        }
    }
}

}  // namespace ls

using namespace ls;

// ---- host driver -------------------------------------------------------------------------------------------------------------------
namespace {

struct Blk { double* M; double* X; double* ws; int n, ldm, ldx; };

// The factorisation is a fixed sequence of batched launches (per level: extend-add per sibling index, the recursion of the inverse,
// two products, the conversion) whose descriptors depend on the plan only. They are RECORDED first (host, no device call), uploaded
// with three copies, and then launched back to back: the first version uploaded every launch's descriptors on its own (~180 small
// copies from pageable memory) and synchronised the stream ~40 times to reuse its staging buffers -- 5-6 of its 21 ms at 1M.
struct Cmd { int kind; size_t off; int n, gx, nmax; };        // kind: 0 product, 1 inverse, 2 extend-add, 3 convert
struct FactorCtx {
    std::vector<GemmDesc> gemm;
    std::vector<InvDesc> inv;
    std::vector<int> ids;
    std::vector<Cmd> cmds;
    int launches = 0;
};

void gemm_batched(FactorCtx& c, std::vector<GemmDesc>& v) {
    const size_t off = c.gemm.size();
    int tiles = 0, tiles32 = 0;
    int64_t all = 0;
    for (const GemmDesc& d : v)
        if (d.M > 0 && d.N > 0) {
            c.gemm.push_back(d);
            const int t64 = div_up(d.M, GT) * div_up(d.N, GT);
            tiles = std::max(tiles, t64); tiles32 = std::max(tiles32, div_up(d.M, 32) * div_up(d.N, 32));
            all += d.sym ? (t64 + div_up(d.M, GT)) / 2 : t64;
        }
    v.clear();
    // fewer 64 x 64 tiles than half the chip's CUs in the whole launch: quarter tiles (Cmd::nmax carries the tile size of a product)
    static const int small = [] { const char* e = getenv("LS_GEMM_SMALL_TILES"); return e ? atoi(e) : 128; }();
    const bool quarter = all < small;
    if (c.gemm.size() > off) c.cmds.push_back(Cmd{0, off, (int)(c.gemm.size() - off), quarter ? tiles32 : tiles, quarter ? 32 : 64});
}

void inverse_small(FactorCtx& c, const std::vector<Blk>& v) {
    const size_t off = c.inv.size();
    int nmax = 0;
    for (const Blk& b : v) if (b.n > 0) { c.inv.push_back(InvDesc{b.M, b.X, b.n, b.ldm, b.ldx, 0}); nmax = std::max(nmax, b.n); }
    if (c.inv.size() > off) c.cmds.push_back(Cmd{1, off, (int)(c.inv.size() - off), 0, nmax});
}

void id_launch(FactorCtx& c, int kind, const std::vector<int>& ids, int gx) {
    if (ids.empty()) return;
    c.cmds.push_back(Cmd{kind, c.ids.size(), (int)ids.size(), gx, 0});
    c.ids.insert(c.ids.end(), ids.begin(), ids.end());
}

// X = M^-1 for every block (SPD, any size): 2 x 2 Schur recursion `depth` times, then the in-register inverse. M is overwritten.
void inverse_rec(FactorCtx& c, const std::vector<Blk>& v, int depth) {
    if (depth == 0) { inverse_small(c, v); return; }
    const size_t N = v.size();
    std::vector<Blk> A(N), S(N);
    std::vector<GemmDesc> g;
    for (size_t i = 0; i < N; ++i) {
        const Blk& b = v[i];
        const int n1 = (b.n + 1) / 2, n2 = b.n - n1;
        A[i] = Blk{b.M, b.X, b.ws + (size_t)n2 * n1, n1, b.ldm, b.ldx};
        S[i] = Blk{b.M + (size_t)n1 * b.ldm + n1, b.X + (size_t)n1 * b.ldx + n1, b.ws + (size_t)n2 * n1, n2, b.ldm, b.ldx};
    }
    inverse_rec(c, A, depth - 1);                                     // X11 = A^-1
    for (size_t i = 0; i < N; ++i) {                                  // T = B X11
        const Blk& b = v[i]; const int n1 = A[i].n, n2 = S[i].n;
        g.push_back(GemmDesc{b.M + (size_t)n1 * b.ldm, b.X, b.ws, n2, n1, n1, b.ldm, b.ldx, n1, 0, 0, 1.0, 0.0});
    }
    gemm_batched(c, g);
    for (size_t i = 0; i < N; ++i) {                                  // S = C - T B^T (in place)
        const Blk& b = v[i]; const int n1 = A[i].n, n2 = S[i].n;
        g.push_back(GemmDesc{b.ws, b.M + (size_t)n1 * b.ldm, S[i].M, n2, n2, n1, n1, b.ldm, b.ldm, 0, 1, -1.0, 1.0, 1, 0});
    }
    gemm_batched(c, g);
    inverse_rec(c, S, depth - 1);                                     // X22 = S^-1
    for (size_t i = 0; i < N; ++i) {                                  // X21 = -X22 T
        const Blk& b = v[i]; const int n1 = A[i].n, n2 = S[i].n;
        g.push_back(GemmDesc{S[i].X, b.ws, b.X + (size_t)n1 * b.ldx, n2, n1, n2, b.ldx, n1, b.ldx, 0, 0, -1.0, 0.0});
    }
    gemm_batched(c, g);
    for (size_t i = 0; i < N; ++i) {                                  // X12 = -T^T X22 ; X11 -= T^T X21
        const Blk& b = v[i]; const int n1 = A[i].n, n2 = S[i].n;
        g.push_back(GemmDesc{b.ws, S[i].X, b.X + n1, n1, n2, n2, n1, b.ldx, b.ldx, 1, 0, -1.0, 0.0});
        g.push_back(GemmDesc{b.ws, b.X + (size_t)n1 * b.ldx, b.X, n1, n1, n2, n1, b.ldx, b.ldx, 1, 0, -1.0, 1.0, 1, 0});
    }
    gemm_batched(c, g);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// defined in direct.hip: builds the handle from plan + factor arrays and takes ownership of the device arrays
extern "C" int ls_direct_create(const ls_direct_arrays* A, int device, void* stream, ls_direct** out);
int ls_direct_adopt(ls_direct* d, void* const* owned, const size_t* owned_bytes, int n_owned, const double* seconds3, const double* quality4);
namespace ls { hipStream_t side_stream(int device, int which); extern std::atomic<long long> g_malloc_calls, g_malloc_us, g_malloc_bytes; }
bool direct_tier_fits(int levels, int arity, const int* s, const int* b, const int* own_start, int tier_levels, bool sparse_leaves, int waves);
bool direct_tier_full16(int64_t V, int arity, int levels, int tier_levels, int shard_count, int tier_waves);

// ---- the tree the library picks for a system of V unknowns (leaf_size <= 0 / arity <= 0 on entry = "pick"; explicit values stay) ----------
extern "C" int ls_direct_pick_tree(int64_t V, int* leaf_size_io, int* arity_io) {
    using namespace ls;
    LS_REQUIRE(leaf_size_io && arity_io && V > 0, LS_E_INVALID, "ls_direct_pick_tree: bad argument");
    int leaf_size = *leaf_size_io, arity = *arity_io;
    // leaf_size <= 0: picked by the size of the system. A re-solve of a small mesh is a chain of launches, ~8-11 us each however few
    // bytes they move, so small meshes want SHALLOW trees of big dense nodes (tools/leaf_sweep.py, profiles/r03_leaf_size_sweep.txt):
    // up to 1280 vertices ONE dense node (one launch for both sweeps: 11 us against 33-36 with leaves of 64), up to 32k vertices
    // leaves of up to 1024 (3-4 levels, 19-43 us against 37-53); beyond that the bytes of the dense leaves cost more than the launches
    // they save and the leaves are the 64-vertex sparse ones of the tier kernels.
    // Between 32k and 128k vertices a tree whose depth rounds the 64-vertex leaves down to < 32 vertices (the rounds come in pairs at
    // arity 4) is better off one level shallower with dense leaves of 65-128 rows (70k: 75 -> 69 us, 90k: 75.5 -> 70, 122k: 78 -> 76;
    // from 490k on the sparse leaves' bytes win: 151 against 184 us).
    // arity <= 0: picked as well. Between ~12k and ~300k vertices a tree that merges THREE bisection rounds per level (arity 8) has two
    // levels = four launches less than the arity-4 tree and wins although its nodes are larger (16k: 27.9 against 36.5 us, 70k plane 64.8 /
    // 68.3, the 70k and 250k configs 67.0 / 71.5 and 105.8 / 113.0, 250k plane 90.0 / 95.6); from ~490k on its extra factor entries cost
    // more than the launches (490k: 196 against 151 us, 1M: 358 against 229).
    if (arity <= 0) arity = (V > 12000 && V <= 300000) ? 8 : 4;
    if (leaf_size <= 0) {
        if (arity == 8) {
            // up to 36k vertices three levels of big dense nodes and no tier (16k: 28 us); up to 105k vertices four levels with dense
            // leaves of 70-205 rows walked by the tier's leaf launch alone (40k: 43 against 52 us, 50k: 48 / 59, 70k: 57 / 65, 100k: 63 /
            // 66); beyond, five levels with the sparse 64-vertex leaves
            leaf_size = V <= 1280 ? (int)V : V <= 36000 ? 1024 : V <= 105000 ? 256 : 64;
            // (the rounds come in threes: a depth that rounds the leaves down to < 12 vertices costs a whole level -- 300k: 209 us with
            // leaves of 9 against 126 with leaves of 73)
            if (leaf_size == 64 && (V >> nd_plan_rounds(V, 64, 8)) < 12) leaf_size = 128;
        }
        else {
            leaf_size = V <= 1280 ? (int)V : V <= 32768 ? 1024 : 64;
            if (V > 32768 && V <= 131072 && (V >> nd_plan_rounds(V, 64, arity)) < 32) leaf_size = 128;
        }
    }
    *leaf_size_io = leaf_size; *arity_io = arity;
    return LS_OK;
}

static int direct_factor_impl(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                              const float* d_positions, int leaf_size, int arity, int tier_levels, int sparse_leaves, int shard_rank,
                              int shard_count, int ordering_arg, int tier_waves, int device, void* stream, ls_direct** out);

extern "C" int ls_direct_factor(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                                const float* d_positions, int leaf_size, int arity, int tier_levels, int sparse_leaves, int shard_rank,
                                int shard_count, int device, void* stream, ls_direct** out) {
    return direct_factor_impl(d_rowptr, d_col, d_val, V, nnz, d_positions, leaf_size, arity, tier_levels, sparse_leaves, shard_rank, shard_count,
                              LS_ND_ORDER_AUTO, 0, device, stream, out);
}

// the same constructor with every choice as an ARGUMENT (round 6: `ordering` used to travel through the process environment)
extern "C" int ls_direct_factor_ex(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                                   const float* d_positions, const ls_direct_options* opt, int device, void* stream, ls_direct** out) {
    ls_direct_options o;
    ls_direct_options_default(&o);
    if (opt) {
        LS_REQUIRE(opt->struct_bytes >= 8 && opt->struct_bytes <= 4096, LS_E_INVALID, "ls_direct_factor_ex: options.struct_bytes is not set (ls_direct_options_default)");
        memcpy(&o, opt, std::min((size_t)opt->struct_bytes, sizeof(o)));      // fields the caller's header did not know keep their defaults
        o.struct_bytes = (int32_t)sizeof(o);
    }
    LS_REQUIRE(o.ordering >= LS_ND_ORDER_AUTO && o.ordering <= LS_ND_ORDER_MINSEP, LS_E_INVALID, "ls_direct_factor_ex: ordering must be LS_ND_ORDER_AUTO / _LONGEST / _MINSEP");
    LS_REQUIRE(o.tier_waves == 0 || o.tier_waves == 4 || o.tier_waves == 8 || o.tier_waves == 16, LS_E_INVALID, "ls_direct_factor_ex: tier_waves must be 0 (library's rule), 4, 8 or 16");
    return direct_factor_impl(d_rowptr, d_col, d_val, V, nnz, d_positions, o.leaf_size, o.arity, o.tier_levels, o.sparse_leaves, o.shard_rank,
                              o.shard_count, o.ordering, o.tier_waves, device, stream, out);
}

extern "C" int ls_direct_options_default(ls_direct_options* o) {
    LS_REQUIRE(o, LS_E_INVALID, "ls_direct_options_default: null argument");
    memset(o, 0, sizeof(*o));
    o->struct_bytes = (int32_t)sizeof(*o);
    o->leaf_size = 0; o->arity = 0; o->tier_levels = -1; o->sparse_leaves = 1; o->shard_rank = 0; o->shard_count = 1;
    o->ordering = LS_ND_ORDER_AUTO; o->tier_waves = 0;
    return LS_OK;
}

static int direct_factor_impl(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                              const float* d_positions, int leaf_size, int arity, int tier_levels, int sparse_leaves, int shard_rank,
                              int shard_count, int ordering_arg, int tier_waves, int device, void* stream, ls_direct** out) {
    LS_REQUIRE(out && d_rowptr && d_col && d_val && V > 0 && nnz > 0 && nnz < INT32_MAX, LS_E_INVALID, "ls_direct_factor: bad argument");
    *out = nullptr;
    { const int rc_pick = ls_direct_pick_tree(V, &leaf_size, &arity); if (rc_pick) return rc_pick; }      // leaf_size / arity <= 0: picked from V
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const double t0 = now_s();
    const bool timing = getenv("LS_PLAN_TIMING") != nullptr;
    auto faults = [] { struct rusage u; getrusage(RUSAGE_SELF, &u); return (long)u.ru_minflt; };
    const long f0 = timing ? faults() : 0;
    auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); fprintf(stderr, "[ls_direct_factor] %-30s %.3f s  (%ld page faults so far)\n", what, now_s() - t0, faults() - f0); } };
    // ---- symbolic analysis: the bisection rounds on the device (nd_bisect.hip), the tree / fronts / index lists on the host ------------
    uvec<int32_t> rowptr((size_t)V + 1), col((size_t)nnz);                 // the host's copy of the pattern, filled by the analysis (not zeroed first: 32 MB at 1M)
    NdPlan P;
    {
        // how the cutting directions are chosen: ND_ORDER_AUTO (nd_plan.h) unless the environment says otherwise (LS_ND_ORDER = 0: always
        // the longest axis of the embedding, 1: always the thinnest of six trial separators -- on the device since round 5: 5-10 % fewer
        // factor numbers on rough scans for 10-25 ms more constructor at 250k vertices)
        // an explicit argument (ls_direct_factor_ex) wins; "auto" lets the environment override the library's rule
        const char* oe = getenv("LS_ND_ORDER");
        const int ordering = ordering_arg != ND_ORDER_AUTO ? ordering_arg : oe ? std::max(-1, std::min(1, atoi(oe))) : ND_ORDER_AUTO;
        const std::string err = nd_plan_build_device(d_rowptr, d_col, d_positions, V, nnz, rowptr.data(), col.data(), leaf_size, arity, 4, st, P, ordering,
                                                     /* defer_push_lists = */ true);
        LS_REQUIRE(err.empty(), LS_E_INVALID, "%s", err.c_str());
    }
    const double t1 = now_s();
    lap("symbolic analysis");
    const int n_nodes = P.n_nodes, levels = P.levels;
    int max_front = 0;
    for (int i = 1; i <= n_nodes; ++i) max_front = std::max(max_front, P.s[i] + P.b[i]);
    LS_REQUIRE(max_front <= 8000, LS_E_WORKSPACE, "ls_direct_factor: a front of %d rows exceeds the solver's limit (the mesh does not dissect)", max_front);
    // ---- layouts -------------------------------------------------------------------------------------------------------------------
    // tier_levels < 0: chosen here so that about a thousand subtrees (4 workgroups per CU) are left at the tier's root level:
    // levels - 5 at arity 4 -- three at 1M vertices (8 levels), four at 4M (9 levels: 0.795 ms against 0.850 with three), two
    // below 8 levels (a tier of three on a 7-level tree leaves most CUs without a workgroup; tools/tier_sweep.py, 576 .. 4M vertices)
    const bool tier_auto = tier_levels < 0;
    if (tier_levels < 0) {
        tier_levels = std::max(2, std::min(4, levels - 5));
        if (shard_count > 1) {                  // the cut (first level with a subtree per rank) must not lie inside the tier
            int cut = 0;
            int64_t width = 1;
            while (width < shard_count && cut < levels) { width *= arity; ++cut; }
            tier_levels = std::max(0, std::min(tier_levels, levels - cut));
        }
    }
    tier_levels = std::max(0, std::min(std::min(tier_levels, levels), 6));
    if (tier_auto) {
        // the tier walks a node with ONE workgroup (a wave per 64-row chunk): right for leaves of <= 64 vertices, fine for the leaf level
        // alone up to ~200 rows (tools/leaf_sweep.py: arity 8, 70k vertices, leaves of 137: 57 us with a tier of one level, 97 with two,
        // 74 with none), 10-30x too slow for a leaf of many hundreds or thousands of rows (a caller's large leaf_size; the single dense node of a very small mesh) -- those go
        // through the level kernels, which spread a node over as many workgroups as it has row tiles
        int leaf_max = 0;
        for (int64_t i = P.level_off[levels - 1]; i < P.level_off[levels]; ++i) leaf_max = std::max(leaf_max, P.s[i]);
        const int64_t n_leaves = P.level_off[levels] - P.level_off[levels - 1];
        if (leaf_max > 256 || levels == 1) tier_levels = 0;         // (a single node: the root's launch does both sweeps)
        else if (leaf_max > 64) {                                   // dense leaves
            // many leaves of up to ~220 rows in a shallow tree: the tier's leaf launch alone (64 leaves of 256 rows are better off in the
            // level kernels: 16k vertices 28 against 37 us)
            if (levels <= 4 && n_leaves >= 256 && leaf_max <= 224) tier_levels = std::min(tier_levels, 1);
            else if (leaf_max > 128) tier_levels = 0;
        }
    }
    bool leaves_ok = tier_levels > 0 && sparse_leaves;
    for (int64_t i = P.level_off[levels - 1]; i < P.level_off[levels] && leaves_ok; ++i) leaves_ok = P.s[i] <= 64;
    // a tier the library picked itself never fails for lack of LDS: one level less until its subtrees fit a workgroup
    // (an explicit tier_levels that does not fit is reported by ls_direct_create: LS_E_WORKSPACE)
    if (tier_auto) {
        // large systems: a subtree one level taller per workgroup of sixteen waves, if its leaves and vectors fit the 160 KB of LDS
        const int taller = levels - 4;
        if (direct_tier_full16(V, arity, levels, taller, shard_count, tier_waves) && direct_tier_fits(levels, arity, P.s.data(), P.b.data(), P.own_start.data(), taller, leaves_ok, 16))
            tier_levels = taller;
        else
            while (tier_levels > 0 && !direct_tier_fits(levels, arity, P.s.data(), P.b.data(), P.own_start.data(), tier_levels, leaves_ok, 4)) --tier_levels;
    }
    if (tier_levels == 0) leaves_ok = false;
    const int tier_root = levels - tier_levels;
    std::vector<FactorNode> fn((size_t)n_nodes + 1);
    memset(fn.data(), 0, fn.size() * sizeof(FactorNode));
    std::vector<int64_t> hn((size_t)(n_nodes + 1) * LS_DIRECT_NODE_COLS, 0);
    int64_t f_tot = 0, x_tot = 0, w_tot = 0, o_finv = 0, o_w = 0, o_d4 = 0, o_u4 = 0, o_tri = 0;
    for (int i = 1; i <= n_nodes; ++i) {
        FactorNode& n = fn[i];
        const int s = P.s[i], b = P.b[i], lv = P.level_of[i];
        n.s = s; n.b = b; n.own_start = P.own_start[i]; n.parent = P.parent[i]; n.bnd_off = P.bnd_off[i];
        n.f_off = f_tot; f_tot += (int64_t)(s + b) * (s + b);
        n.x_off = x_tot; x_tot += (int64_t)s * s;
        n.w_off = w_tot; w_tot += (int64_t)s * b;
        const bool sparse = leaves_ok && lv == levels - 1 && s >= 1;
        const bool quad = !sparse && lv >= tier_root;
        n.layout = sparse ? 2 : quad ? 1 : 0; n.pad = lv;
        int64_t* r = hn.data() + (size_t)i * LS_DIRECT_NODE_COLS;
        r[0] = s; r[1] = b; r[2] = P.own_start[i]; r[3] = P.bnd_off[i]; r[4] = P.front_off[i]; r[7] = P.parent[i];
        r[8] = -1; r[9] = -1; r[10] = -1; r[11] = quad;
        const int64_t s4 = (s + 3) & ~3, b4 = (b + 3) & ~3;
        if (sparse) { n.o_tri = o_tri; r[8] = o_tri; o_tri += ((int64_t)s * (s + 1) / 2 + 3) & ~(int64_t)3; }
        else if (quad) { n.o_finv = o_d4; n.o_w = o_u4; r[5] = o_d4; r[6] = o_u4; o_d4 += (s4 + b4) * s; o_u4 += s4 * b; }
        else { n.o_finv = o_finv; n.o_w = o_w; r[5] = o_finv; r[6] = o_w; o_finv += (int64_t)s * s; o_w += (int64_t)s * b; }
    }
    LS_REQUIRE(o_finv + 2 * o_w + o_d4 + o_u4 + 2 * o_tri < (int64_t)4000000000, LS_E_WORKSPACE, "ls_direct_factor: the factor is too large");
    lap("layouts");
    // ---- sparse leaves: where every leaf's two pointer lists (A_bs by boundary row / by own row) start; the lists themselves are
    // built on the device below
    std::vector<int> off_b, off_s;
    int64_t n_sp_ptr = 0;
    const int64_t leaf0 = P.level_off[levels - 1], leaf1 = P.level_off[levels];
    if (leaves_ok) {
        off_b.assign((size_t)(leaf1 - leaf0), -1); off_s.assign((size_t)(leaf1 - leaf0), -1);
        for (int pass = 0; pass < 2; ++pass)                 // boundary rows of all leaves first, then own rows
            for (int64_t i = leaf0; i < leaf1; ++i) {
                if (P.s[i] < 1) continue;
                (pass == 0 ? off_b : off_s)[(size_t)(i - leaf0)] = (int)n_sp_ptr;
                hn[(size_t)i * LS_DIRECT_NODE_COLS + (pass == 0 ? 9 : 10)] = n_sp_ptr;
                n_sp_ptr += (pass == 0 ? P.b[i] : P.s[i]) + 1;
            }
    }
    const double t2 = now_s();
    // ---- device storage ---------------------------------------------------------------------------------------------------------------
    std::vector<void*> owned, scratch;
    std::vector<size_t> owned_bytes, scratch_bytes;
    int rc = LS_OK;
    auto dalloc = [&](void** p, size_t bytes, bool keep, bool zero) -> bool {
        const double ta = timing ? now_s() : 0.0;
        const size_t want = std::max<size_t>(bytes, 16) + 16;
        hipError_t e = hipSuccess;
        size_t cap = want;
        *p = pool_take(device, want, &cap);                          // large buffers: from the pool the previous solver's went to (direct.hip)
        const bool pooled = *p != nullptr;
        if (!pooled) e = pool_alloc(device, p, want);                // (out of memory: the pool is emptied and the call repeated once)
        const double tb = timing ? now_s() : 0.0;
        if (e == hipSuccess && zero) e = hipMemsetAsync(*p, 0, std::max<size_t>(bytes, 16) + 16, st);
        if (timing && bytes > ((size_t)256 << 20)) {
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[ls_direct_factor]   %.2f GB: %s %.1f ms, %s %.1f ms\n", bytes / 1073741824.0, pooled ? "from the pool" : "hipMalloc", (tb - ta) * 1e3, zero ? "zeroed in" : "no memset",
                    (now_s() - tb) * 1e3);
        }
        if (e != hipSuccess) { rc = hip_fail(e, "ls_direct_factor allocation", __FILE__, __LINE__); *p = nullptr; return false; }
        (keep ? owned : scratch).push_back(*p);
        (keep ? owned_bytes : scratch_bytes).push_back(cap);
        return true;
    };
    float *finv = nullptr, *wf = nullptr, *wb = nullptr, *u4 = nullptr, *d4 = nullptr, *tri = nullptr;
    int32_t* d_sp_ptr = nullptr; SpEnt* d_sp_ent = nullptr;
    double *fronts = nullptr, *xs = nullptr, *ws = nullptr, *work = nullptr;
    int *d_inv = nullptr, *d_non = nullptr, *d_bnd = nullptr, *d_ppos = nullptr, *d_rowidx = nullptr, *d_ids = nullptr;
    FactorNode* d_nodes = nullptr;
    FactorCtx ctx;
    int* d_flag = nullptr;
    GemmDesc* d_gemm = nullptr; InvDesc* d_invd = nullptr;
    int64_t work_tot = 0;
    for (int i = 1; i <= n_nodes; ++i) work_tot += ((int64_t)P.s[i] * P.s[i] + 1) / 2 + 64;
    bool ok = dalloc((void**)&finv, sizeof(float) * o_finv, true, false) && dalloc((void**)&wf, sizeof(float) * o_w, true, false) &&
              dalloc((void**)&wb, sizeof(float) * o_w, true, false) && dalloc((void**)&u4, sizeof(float) * o_u4, true, true) &&
              dalloc((void**)&d4, sizeof(float) * o_d4, true, true) && dalloc((void**)&tri, sizeof(float) * o_tri, true, true) &&
              dalloc((void**)&d_sp_ptr, sizeof(int32_t) * n_sp_ptr, true, false) &&
              dalloc((void**)&fronts, sizeof(double) * f_tot, false, true) && dalloc((void**)&xs, sizeof(double) * x_tot, false, false) &&
              dalloc((void**)&ws, sizeof(double) * w_tot, false, false) && dalloc((void**)&work, sizeof(double) * work_tot, false, false) &&
              dalloc((void**)&d_inv, sizeof(int) * V, false, false) && dalloc((void**)&d_non, sizeof(int) * V, false, false) &&
              dalloc((void**)&d_bnd, sizeof(int) * P.n_bnd, false, false) && dalloc((void**)&d_ppos, sizeof(int) * P.n_bnd, false, false) &&
              dalloc((void**)&d_rowidx, sizeof(int) * nnz, false, false) &&
              dalloc((void**)&d_nodes, sizeof(FactorNode) * (n_nodes + 1), false, false) && dalloc((void**)&d_flag, sizeof(int), false, true);
    auto cleanup = [&](bool all) {
        (void)hipStreamSynchronize(st);
        for (size_t i = 0; i < scratch.size(); ++i) if (!pool_give(device, scratch[i], scratch_bytes[i])) (void)hipFree(scratch[i]);
        scratch.clear(); scratch_bytes.clear();
        if (all) {
            for (size_t i = 0; i < owned.size(); ++i) if (!pool_give(device, owned[i], owned_bytes[i])) (void)hipFree(owned[i]);
            owned.clear(); owned_bytes.clear();
        }
    };
    if (!ok) { cleanup(true); return rc; }
    lap("device allocations");
    hipError_t e = hipSuccess;
    auto h2d = [&](void* dst, const void* src, size_t bytes) { if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); };
    h2d(d_inv, P.inv.data(), sizeof(int) * V); h2d(d_non, P.node_of_new.data(), sizeof(int) * V);
    h2d(d_bnd, P.bnd.data(), sizeof(int) * P.n_bnd); h2d(d_ppos, P.ppos.data(), sizeof(int) * P.n_bnd);
    h2d(d_nodes, fn.data(), sizeof(FactorNode) * (n_nodes + 1));
    if (e != hipSuccess) { cleanup(true); return hip_fail(e, "ls_direct_factor uploads", __FILE__, __LINE__); }
    hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)div_up(V, 256)), dim3(256), 0, st, V, d_rowptr, d_rowidx);
    int64_t n_ent = 0;
    if (leaves_ok) {
        // own rows of the leaves: the tree's numbering [0, rows_s); boundary rows: bnd[bnd0, n_bnd)
        const int64_t rows_s = levels > 1 ? P.own_start[P.level_off[levels - 2]] : V, bnd0 = P.bnd_off[leaf0], rows_b = P.n_bnd - bnd0;
        int *cnt_s = nullptr, *cnt_b = nullptr, *ptr_s = nullptr, *ptr_b = nullptr, *bsum = nullptr, *d_off_b = nullptr, *d_off_s = nullptr;
        ok = dalloc((void**)&cnt_s, sizeof(int) * (rows_s + 1), false, true) && dalloc((void**)&cnt_b, sizeof(int) * (rows_b + 1), false, true) &&
             dalloc((void**)&ptr_s, sizeof(int) * (rows_s + 1), false, false) && dalloc((void**)&ptr_b, sizeof(int) * (rows_b + 1), false, false) &&
             dalloc((void**)&bsum, sizeof(int) * (scan_blocks(std::max(rows_s, rows_b)) + 2), false, false) &&
             dalloc((void**)&d_off_b, sizeof(int) * off_b.size(), false, false) && dalloc((void**)&d_off_s, sizeof(int) * off_s.size(), false, false);
        if (!ok) { cleanup(true); return rc; }
        h2d(d_off_b, off_b.data(), sizeof(int) * off_b.size()); h2d(d_off_s, off_s.data(), sizeof(int) * off_s.size());
        const unsigned eg = (unsigned)div_up(nnz, 256);
        hipLaunchKernelGGL(k_leaf_entries<0>, dim3(eg), dim3(256), 0, st, nnz, d_rowidx, d_col, d_val, d_inv, d_non, d_nodes, d_bnd, (long long)bnd0, cnt_s,
                           cnt_b, (const int*)nullptr, (const int*)nullptr, (SpEnt*)nullptr, 0);
        int tot[2] = {0, 0};
        if ((rc = exclusive_scan(cnt_s, rows_s, ptr_s, bsum, st)) != LS_OK || (rc = exclusive_scan(cnt_b, rows_b, ptr_b, bsum, st)) != LS_OK) { cleanup(true); return rc; }
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[0], ptr_s + rows_s, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&tot[1], ptr_b + rows_b, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemsetAsync(cnt_s, 0, sizeof(int) * (rows_s + 1), st);
        if (e == hipSuccess) e = hipMemsetAsync(cnt_b, 0, sizeof(int) * (rows_b + 1), st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { cleanup(true); return hip_fail(e, "ls_direct_factor leaf lists", __FILE__, __LINE__); }
        if (tot[0] != tot[1]) { cleanup(true); set_error("ls_direct_factor: the leaf lists disagree (%d own-row entries, %d boundary-row entries)", tot[0], tot[1]); return LS_E_INVALID; }
        n_ent = tot[0];
        if (!dalloc((void**)&d_sp_ent, sizeof(SpEnt) * 2 * n_ent, true, true)) { cleanup(true); return rc; }
        hipLaunchKernelGGL(k_leaf_entries<1>, dim3(eg), dim3(256), 0, st, nnz, d_rowidx, d_col, d_val, d_inv, d_non, d_nodes, d_bnd, (long long)bnd0, cnt_s,
                           cnt_b, (const int*)ptr_s, (const int*)ptr_b, d_sp_ent, (int)n_ent);
        if (rows_b) hipLaunchKernelGGL(k_leaf_sort, dim3((unsigned)div_up(rows_b, 256)), dim3(256), 0, st, rows_b, (const int*)ptr_b, 0, d_sp_ent);
        if (rows_s) hipLaunchKernelGGL(k_leaf_sort, dim3((unsigned)div_up(rows_s, 256)), dim3(256), 0, st, rows_s, (const int*)ptr_s, (int)n_ent, d_sp_ent);
        hipLaunchKernelGGL(k_leaf_ptrs, dim3((unsigned)(leaf1 - leaf0)), dim3(64), 0, st, (int)leaf0, d_nodes, (long long)bnd0, (const int*)d_off_b,
                           (const int*)d_off_s, (const int*)ptr_b, (const int*)ptr_s, (int)n_ent, d_sp_ptr);
    } else if (!dalloc((void**)&d_sp_ent, 16, true, true)) { cleanup(true); return rc; }
    if (e != hipSuccess) { cleanup(true); return hip_fail(e, "ls_direct_factor uploads", __FILE__, __LINE__); }
    lap("uploads");
    // ---- numeric factorisation: recorded (host), uploaded, launched back to back ---------------------------------------------------------
    std::vector<int> ids;
    std::vector<int64_t> work_off((size_t)n_nodes + 1, 0);
    { int64_t o = 0; for (int i = 1; i <= n_nodes; ++i) { work_off[i] = o; o += ((int64_t)P.s[i] * P.s[i] + 1) / 2 + 64; } }
    for (int lv = levels - 1; lv >= 0; --lv) {
        const int64_t first = P.level_off[lv], last = P.level_off[lv + 1];
        if (lv + 1 < levels) {                                      // children's Schur complements, one sibling index per launch
            for (int c = 0; c < arity; ++c) {
                ids.clear();
                int bmax = 0;
                for (int64_t ch = P.level_off[lv + 1] + c; ch < P.level_off[lv + 2]; ch += arity)
                    if (P.b[ch] > 0) { ids.push_back((int)ch); bmax = std::max(bmax, P.b[ch]); }
                id_launch(ctx, 2, ids, std::min(64, div_up((int64_t)bmax * bmax, 256)));
            }
        }
        int smax = 0;
        for (int64_t i = first; i < last; ++i) smax = std::max(smax, P.s[i]);
        int depth = 0;
        while (((smax + (1 << depth) - 1) >> depth) > INV_N) ++depth;
        std::vector<Blk> blocks;
        ids.clear();
        for (int64_t i = first; i < last; ++i) {
            if (P.s[i] == 0) continue;
            const int m = P.s[i] + P.b[i];
            blocks.push_back(Blk{fronts + fn[i].f_off, xs + fn[i].x_off, work + work_off[i], P.s[i], m, P.s[i]});
            ids.push_back((int)i);
        }
        inverse_rec(ctx, blocks, depth);
        std::vector<GemmDesc> gd;
        for (int64_t i = first; i < last; ++i) {                    // W = F_bs Finv
            const int s = P.s[i], b = P.b[i], m = s + b;
            if (s && b) gd.push_back(GemmDesc{fronts + fn[i].f_off + (size_t)s * m, xs + fn[i].x_off, ws + fn[i].w_off, b, s, s, m, s, s, 0, 0, 1.0, 0.0});
        }
        gemm_batched(ctx, gd);
        for (int64_t i = first; i < last; ++i) {                    // U = F_bb - W F_sb  (F_sb = F_bs^T)
            const int s = P.s[i], b = P.b[i], m = s + b;
            if (s && b) gd.push_back(GemmDesc{ws + fn[i].w_off, fronts + fn[i].f_off + (size_t)s * m, fronts + fn[i].f_off + (size_t)s * m + s,
                                              b, b, s, s, m, m, 0, 1, -1.0, 1.0, 1, 0});
        }
        gemm_batched(ctx, gd);
        int64_t emax = 0;
        for (int i : ids) emax = std::max(emax, (int64_t)P.s[i] * (P.s[i] + P.b[i]));
        id_launch(ctx, 3, ids, std::min(256, div_up(emax, 256)));
    }
    ok = dalloc((void**)&d_gemm, sizeof(GemmDesc) * ctx.gemm.size(), false, false) && dalloc((void**)&d_invd, sizeof(InvDesc) * ctx.inv.size(), false, false) &&
         dalloc((void**)&d_ids, sizeof(int) * ctx.ids.size(), false, false);
    if (!ok) { cleanup(true); return rc; }
    h2d(d_gemm, ctx.gemm.data(), sizeof(GemmDesc) * ctx.gemm.size());
    h2d(d_invd, ctx.inv.data(), sizeof(InvDesc) * ctx.inv.size());
    h2d(d_ids, ctx.ids.data(), sizeof(int) * ctx.ids.size());
    if (e == hipSuccess)
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)div_up(nnz, 256)), dim3(256), 0, st, nnz, d_rowidx, d_col, d_val, d_inv, d_non, d_nodes, d_bnd,
                           fronts, d_flag);
    // The fp32 conversion of a finished level runs on the side stream, beside the next level's chain of small launches (the upper levels
    // are inverses of one workgroup and products of a few tiles: the chip is nearly empty there, and the conversions were 1.0 of the
    // 10.6 ms of kernels of a 1M-vertex factorisation). Events order it: after its level's products, before the stream's end.
    hipStream_t sc = side_stream(device, 1);
    std::vector<hipEvent_t> evs;
#ifdef LS_ND_EXPERIMENTS
    exp_round_configure();
#endif
    for (const Cmd& c : ctx.cmds) {
        if (e != hipSuccess) break;
        hipStream_t sk = st;
        if (c.kind == 3 && sc) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                evs.push_back(ev);
                if (hipEventRecord(ev, st) == hipSuccess && hipStreamWaitEvent(sc, ev, 0) == hipSuccess) sk = sc;
            }
        }
        for (int b0 = 0; b0 < c.n; b0 += 65535) {
            const int nb = std::min(65535, c.n - b0);
            switch (c.kind) {
            case 0:
                if (c.nmax == 32) hipLaunchKernelGGL(k_gemm_batched<32>, dim3(c.gx, nb), dim3(256), 0, st, (const GemmDesc*)(d_gemm + c.off + b0));
                else hipLaunchKernelGGL(k_gemm_batched<64>, dim3(c.gx, nb), dim3(256), 0, st, (const GemmDesc*)(d_gemm + c.off + b0));
                break;
            case 1:
                if (c.nmax <= 64) hipLaunchKernelGGL((k_spd_inverse_reg<64, 16>), dim3(nb), dim3(256), 0, st, (const InvDesc*)(d_invd + c.off + b0), d_flag);
                else if (nb <= 64) hipLaunchKernelGGL((k_spd_inverse_reg<128, 32>), dim3(nb), dim3(1024), 0, st, (const InvDesc*)(d_invd + c.off + b0), d_flag);
                else hipLaunchKernelGGL((k_spd_inverse_reg<128, 16>), dim3(nb), dim3(256), 0, st, (const InvDesc*)(d_invd + c.off + b0), d_flag);   // a block per CU and more: 256 blocks 132 us, with 1024 threads 197
                break;
            case 2: hipLaunchKernelGGL(k_extend_add, dim3(c.gx, nb), dim3(256), 0, st, (const int*)(d_ids + c.off + b0), nb, d_nodes, d_ppos, fronts); break;
            default:
                hipLaunchKernelGGL(k_convert, dim3(c.gx, nb), dim3(256), 0, sk, (const int*)(d_ids + c.off + b0), d_nodes, xs, ws, finv, wf, wb, u4, d4, tri);
            }
        }
        ++ctx.launches;
    }
    if (sc && !evs.empty()) {                                  // join: the caller's stream continues after the last conversion
        hipEvent_t ev = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (ev) evs.push_back(ev);
        if (e == hipSuccess) e = hipEventRecord(ev, sc);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ev, 0);
    }
    struct EventGuard { std::vector<hipEvent_t>& v; ~EventGuard() { for (hipEvent_t x : v) (void)hipEventDestroy(x); } } ev_guard{evs};
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { cleanup(true); return hip_fail(e, "ls_direct_factor kernels", __FILE__, __LINE__); }
    // ---- the solver handle: its tables are built on the host WHILE the device factorises (everything above is only enqueued) ------------
    if (timing) fprintf(stderr, "[ls_direct_factor]   %d launches enqueued %.3f s (host clock)\n", ctx.launches, now_s() - t0);
    nd_plan_push_lists(P);          // (left out of the analysis: only the solve needs them)
    if (timing) fprintf(stderr, "[ls_direct_factor]   push lists built %.3f s (host clock)\n", now_s() - t0);
    ls_direct_arrays A;
    memset(&A, 0, sizeof(A));
    A.V = V; A.levels = levels; A.arity = arity; A.h_nodes = hn.data(); A.h_perm = P.perm.data(); A.h_ppos = P.ppos.data(); A.n_bnd = P.n_bnd;
    A.h_push_ptr = P.push_ptr.data(); A.h_push_tgt = P.push_tgt.data(); A.n_front = P.n_front;
    A.d_finv = finv; A.d_wf = wf; A.d_wb = wb; A.d_u4 = u4; A.d_d4 = d4; A.d_tri = tri; A.d_sp_ptr = d_sp_ptr; A.d_sp_ent = d_sp_ent;
    A.n_sp_ptr = n_sp_ptr; A.n_sp_ent = 2 * n_ent;
    A.shard_rank = shard_rank; A.shard_count = shard_count; A.tier_waves = tier_waves;
    rc = ls_direct_create(&A, device, stream, out);
    int flag = 0;
    e = hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (rc != LS_OK || e != hipSuccess || flag) {
        if (rc == LS_OK) { (void)ls_direct_destroy(*out); *out = nullptr; }       // (the handle does not own the factor arrays yet)
        cleanup(true);
        if (rc != LS_OK) return rc;
        if (e != hipSuccess) return hip_fail(e, "ls_direct_factor kernels", __FILE__, __LINE__);
        set_error(flag == 2 ? "ls_direct_factor: the matrix pattern is not symmetric" : "ls_direct_factor: a front is not positive definite");
        return LS_E_INVALID;
    }
    cleanup(false);
    const double t3 = now_s();
    lap("numeric factorisation + solve tables (overlapped)");
    if (timing) {
        const long long nc = ls::g_malloc_calls.exchange(0), us = ls::g_malloc_us.exchange(0), by = ls::g_malloc_bytes.exchange(0);
        fprintf(stderr, "[ls_direct_factor]   allocations that missed the pool since the last report: %lld hipMalloc calls, %.1f MB, %.2f ms\n", nc, by / 1048576.0, us / 1e3);
    }
    const double secs[3] = {t1 - t0, t2 - t1, t3 - t2};
    const double quality[4] = {(double)P.ordering, P.words_per_vertex, P.spread, P.words_other};
    return ls_direct_adopt(*out, owned.data(), owned_bytes.data(), (int)owned.size(), secs, quality);
}

// ---- is a CSR matrix symmetric (pattern and values)? Replaces a sort-based check on the host side of the solver -------------------
namespace ls {
__global__ void k_csr_symmetric(int64_t nnz, const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
                                const int* __restrict__ rowidx_or_null, int64_t V, float tol, int* __restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int r;
    {   // row of entry e: binary search in rowptr
        int64_t lo = 0, hi = V;
        while (lo + 1 < hi) { const int64_t mid = (lo + hi) >> 1; if (rowptr[mid] <= e) lo = mid; else hi = mid; }
        r = (int)lo;
    }
    const int c = col[e];
    int lo = rowptr[c], hi = rowptr[c + 1];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (col[mid] < r) lo = mid + 1; else hi = mid; }
    if (lo >= rowptr[c + 1] || col[lo] != r || fabsf(val[lo] - val[e]) > tol) atomicExch(flag, 1);
}
}  // namespace ls

extern "C" int ls_csr_is_symmetric(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz, float tol,
                                   int* h_symmetric, int device, void* stream) {
    LS_REQUIRE(d_rowptr && d_col && d_val && h_symmetric && V > 0 && nnz >= 0, LS_E_INVALID, "ls_csr_is_symmetric: bad argument");
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    int* d_flag = nullptr;
    LS_HIP(hipMalloc((void**)&d_flag, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), st);
    if (e == hipSuccess && nnz) hipLaunchKernelGGL(k_csr_symmetric, dim3((unsigned)div_up(nnz, 256)), dim3(256), 0, st, nnz, d_rowptr, d_col, d_val,
                                                  (const int*)nullptr, V, tol, d_flag);
    int flag = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_flag);
    LS_HIP(e);
    *h_symmetric = flag ? 0 : 1;
    return LS_OK;
}
