// direct.hip -- re-solve phase of the nested-dissection multifrontal direct solver (gfx950, wave64).
//
// Replaces the two sparse triangular solves of the reference's default method (largesteps/solvers.py:36-39,
// cholespy/CHOLMOD `solver.solve(b, x)`). Plan and notation: largesteps/nested.py. The factor arrays (fp32) are
// produced once per matrix by largesteps/direct.py and stay resident in HBM:
//     Finv_i = F_ss^-1 (s x s, symmetric),  W_i = F_bs F_ss^-1 stored twice: wf[j*b + i] and wb[i*s + j] = W[i][j],
//     so that either sweep can read it along either index with 256-byte coalesced wave loads.
// One launch per tree level and sweep. Data flow between levels (no atomics, bitwise reproducible):
//     up    a child pushes its update for boundary vertex i into slot `child index` of the parent's front position
//           ppos[i]; the parent reads the slots of a position with one contiguous load, gated by a static bit mask
//     down  a parent pushes x of every front position into the boundary vectors of the children that contain that
//           vertex (static push lists), so a node reads its x_bnd contiguously
// A level is latency bound (launch + ~3 dependent memory round trips), hence the index-free critical paths, the
// prefetch of the first matrix rows before the right-hand side is assembled, and two kernel shapes:
//     k_nd_up / k_nd_down      a row per lane, NW waves split the reduction range and meet in LDS (short reductions)
//     k_nd_up_b / k_nd_down_b  lanes ALONG the reduction, a wave owns ND_ROWS rows, DPP butterfly (long reductions)
//     k_nd_up_s / k_nd_down_s  small nodes: the node's whole matrix staged in LDS by one coalesced burst, a row per thread
//     k_nd_up_p / k_nd_down_p  tiny nodes: up to 8 consecutive nodes share a wave (lane -> (node, row))
#include "common.h"
#include <vector>
#include <chrono>
#include <algorithm>
#include <atomic>
#include <string.h>
#include <mutex>

namespace ls {

struct NodeDesc { int s, b, own_start, bnd_off, front_off, parent; long long finv_off, w_off; };

// one workgroup's job: a range of rows (from row0) of one node; everything the kernels need in one 64-byte record
struct alignas(64) Tile {
    int store;                 // up sweep: 0 = compute rows, 1 = compute rows and store all of b', 2 = only store b' of [row0, row0 + blockDim)
    int row0, s, b, own_start, bnd_off, front_off;
    int leaf;                  // 1: no children (nothing was pushed into this node's slots)
    int pfront_off, cix;       // parent's front_off (-1 = root) and this node's index among its siblings
    int forward;               // down sweep: 1 = rows are boundary rows that only hand x down to the children
    int arity;
    long long finv_off, w_off;
};

// A tile record is fetched with ONE vector load (lane i takes dword i) and unpacked with v_readlane: a scalar load at the
// head of a workgroup queues behind the streaming vector traffic of its neighbours (~3 us next to a loaded CU).
static_assert(sizeof(Tile) == 64, "Tile layout");
__device__ __forceinline__ Tile load_tile(const Tile* __restrict__ tiles, int idx) {
    const int lane = threadIdx.x & 63;
    const int w = lane < 16 ? reinterpret_cast<const int*>(tiles)[(size_t)idx * 16 + lane] : 0;
    Tile t;
    t.store = __builtin_amdgcn_readlane(w, 0); t.row0 = __builtin_amdgcn_readlane(w, 1); t.s = __builtin_amdgcn_readlane(w, 2);
    t.b = __builtin_amdgcn_readlane(w, 3); t.own_start = __builtin_amdgcn_readlane(w, 4); t.bnd_off = __builtin_amdgcn_readlane(w, 5);
    t.front_off = __builtin_amdgcn_readlane(w, 6); t.leaf = __builtin_amdgcn_readlane(w, 7); t.pfront_off = __builtin_amdgcn_readlane(w, 8);
    t.cix = __builtin_amdgcn_readlane(w, 9); t.forward = __builtin_amdgcn_readlane(w, 10); t.arity = __builtin_amdgcn_readlane(w, 11);
    t.finv_off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w, 13) << 32) | (unsigned)__builtin_amdgcn_readlane(w, 12));
    t.w_off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w, 15) << 32) | (unsigned)__builtin_amdgcn_readlane(w, 14));
    return t;
}

#ifndef LS_ND_UNROLL
#define LS_ND_UNROLL 32       // (8: 17.1 / 17.1 us for the two row-per-lane up levels of a 1M-vertex solve, 16: 19.9 / 19.3, 32: 16.6 / 16.0)
#endif
// independent matrix loads in flight per lane of the row-per-lane kernels. Both kernels are compiled for 1024 threads (128 VGPRs):
// 32 was measured on k_nd_up<3> and fits there (K = 4: 8); k_nd_down keeps TWO such batches (Finv and W^T) and stays at the 8 it
// has always fitted with -- at 32 it spilled 42-114 registers to scratch (round 3). tests/test_abi_and_host.py reads the spill
// counts of every kernel out of the built library.
template <int K> struct NdUnroll { static constexpr int up = K <= 3 ? LS_ND_UNROLL : (LS_ND_UNROLL > 8 ? 8 : LS_ND_UNROLL), down = 8; };
#ifndef LS_ND_ROWS
#define LS_ND_ROWS 2
#endif
constexpr int ND_ROWS = LS_ND_ROWS;     // rows a wave processes together (lanes-along-the-reduction kernels)
#ifndef LS_ND_BW
#define LS_ND_BW 8
#endif
constexpr int ND_BW = LS_ND_BW;       // most waves per workgroup of the *_b kernels (a level runs 4 or ND_BW of them: ND_ROWS x waves x chunks rows per tile)

// Sum of the valid child slots of front position f: slots[(f * A + c) * K + q], valid iff bit c of m. All A slots are
// loaded unconditionally (contiguous; never-written ones hold garbage and are masked out): no load waits for the mask.
template <int K, int A>
__device__ __forceinline__ void load_slots(const float* slots, size_t f, float (&raw)[A * K]) {
#pragma unroll
    for (int e = 0; e < A * K; ++e) raw[e] = slots[f * (A * K) + e];
}
template <int K, int A>
__device__ __forceinline__ void sum_slots(const float (&raw)[A * K], unsigned m, float (&v)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = 0.0f;
#pragma unroll
    for (int c = 0; c < A; ++c) {
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] += ((m >> c) & 1u) ? raw[c * K + q] : 0.0f;
    }
}
template <int K>
__device__ __forceinline__ void pull_slots(const float* slots, const unsigned char* __restrict__ mask, size_t f, int A, float (&v)[K]) {
    const unsigned m = mask[f];
    if (A == 4) { float raw[4 * K]; load_slots<K, 4>(slots, f, raw); sum_slots<K, 4>(raw, m, v); }
    else if (A == 2) { float raw[2 * K]; load_slots<K, 2>(slots, f, raw); sum_slots<K, 2>(raw, m, v); }
    else { float raw[8 * K]; load_slots<K, 8>(slots, f, raw); sum_slots<K, 8>(raw, m, v); }
}

// b'_j = b_j - (slots at own position j) for rows [j_lo, j_hi) of the node: R rows per thread are in flight together
// (index loads, then value loads, then the arithmetic) -- the chain perm -> b is 2 round trips per batch, not per row.
template <int K, int A, int R>
__device__ __forceinline__ void fill_rows(const Tile& t, int j_lo, int j_hi, const int* __restrict__ perm,
                                          const unsigned char* __restrict__ mask, const float* slots,
                                          const float* __restrict__ b_in, float* __restrict__ bprime, float* __restrict__ sb) {
    for (int jb = j_lo + threadIdx.x; jb < j_hi; jb += blockDim.x * R) {
        size_t g[R];
        unsigned m[R];
        float raw[R][A * K];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = jb + r * blockDim.x;
            const bool ok = j < j_hi;
            g[r] = !ok ? 0 : perm ? (size_t)perm[t.own_start + j] : (size_t)(t.own_start + j);   // perm == nullptr: b_in is already in the tree's numbering
            m[r] = (ok && !t.leaf) ? mask[t.front_off + j] : 0u;
            if (ok && !t.leaf) load_slots<K, A>(slots, (size_t)(t.front_off + j), raw[r]);
        }
        float v[R][K];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = jb + r * blockDim.x;
#pragma unroll
            for (int q = 0; q < K; ++q) v[r][q] = (j < j_hi) ? b_in[g[r] * K + q] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = jb + r * blockDim.x;
            if (j < j_hi) {
                if (!t.leaf) {
                    float u[K];
                    sum_slots<K, A>(raw[r], m[r], u);
#pragma unroll
                    for (int q = 0; q < K; ++q) v[r][q] -= u[q];
                }
                if (sb) {
#pragma unroll
                    for (int q = 0; q < K; ++q) sb[j * K + q] = v[r][q];
                }
                if (bprime) {
#pragma unroll
                    for (int q = 0; q < K; ++q) bprime[(size_t)(t.own_start + j) * K + q] = v[r][q];
                }
            }
        }
    }
}

// all s rows of the node into LDS (compute tiles); `store` = 1 also keeps them for the down sweep
template <int K>
__device__ __forceinline__ void fill_bprime(const Tile& t, const int* __restrict__ perm, const unsigned char* __restrict__ mask,
                                            const float* slots, const float* __restrict__ b_in, float* __restrict__ bprime,
                                            float* __restrict__ sb) {
    float* keep = t.store == 1 ? bprime : nullptr;
    if (t.arity == 4) fill_rows<K, 4, 4>(t, 0, t.s, perm, mask, slots, b_in, keep, sb);
    else if (t.arity == 2) fill_rows<K, 2, 4>(t, 0, t.s, perm, mask, slots, b_in, keep, sb);
    else fill_rows<K, 8, 2>(t, 0, t.s, perm, mask, slots, b_in, keep, sb);
}

// store-only tiles of the up sweep (large nodes): b' of own rows [row0, row0 + blockDim.x), nothing else
template <int K>
__device__ __forceinline__ void store_bprime(const Tile& t, const int* __restrict__ perm, const unsigned char* __restrict__ mask,
                                             const float* slots, const float* __restrict__ b_in, float* __restrict__ bprime) {
    const int hi = min(t.s, t.row0 + (int)blockDim.x);
    if (t.arity == 4) fill_rows<K, 4, 1>(t, t.row0, hi, perm, mask, slots, b_in, bprime, nullptr);
    else if (t.arity == 2) fill_rows<K, 2, 1>(t, t.row0, hi, perm, mask, slots, b_in, bprime, nullptr);
    else fill_rows<K, 8, 1>(t, t.row0, hi, perm, mask, slots, b_in, bprime, nullptr);
}

// The root's up-sweep step is only  b'_s = b_s - (slots at own positions): the root's down-sweep tiles can form it themselves
// (rf.slots != nullptr) and the root needs no up-sweep launch at all.
struct RootFill { const int* perm; const unsigned char* mask; const float* slots; const float* b_in; };
template <int K>
__device__ __forceinline__ void root_bprime(const Tile& t, const RootFill& rf, float* __restrict__ sb) {
    if (t.arity == 4) fill_rows<K, 4, 4>(t, 0, t.s, rf.perm, rf.mask, rf.slots, rf.b_in, nullptr, sb);
    else if (t.arity == 2) fill_rows<K, 2, 4>(t, 0, t.s, rf.perm, rf.mask, rf.slots, rf.b_in, nullptr, sb);
    else fill_rows<K, 8, 2>(t, 0, t.s, rf.perm, rf.mask, rf.slots, rf.b_in, nullptr, sb);
}

// hand x of a front position down: xb[target] = v for every child boundary entry that is this vertex
template <int K>
__device__ __forceinline__ void push_down(const int* __restrict__ push_tgt, int p0, int p1, float* xb, const float (&v)[K]) {
    for (int p = p0; p < p1; ++p) {
        const size_t tgt = (size_t)push_tgt[p];
#pragma unroll
        for (int q = 0; q < K; ++q) xb[tgt * K + q] = v[q];
    }
}

// ---- a row per lane ---------------------------------------------------------------------------------------------
// acc += sum_{u in [u0, u1)} col[u * stride] * sv[u * K + q]; the first ND_UNROLL values were prefetched
template <int K, bool NT, int ND_UNROLL>
__device__ __forceinline__ void dot_strided(const float* __restrict__ col, size_t stride, int u0, int u1,
                                            const float* __restrict__ sv, const float (&pre)[ND_UNROLL], float (&acc)[K]) {
#pragma unroll
    for (int e = 0; e < ND_UNROLL; ++e) {
        if (u0 + e < u1) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(pre[e], sv[(u0 + e) * K + q], acc[q]);
        }
    }
    int u = u0 + ND_UNROLL;
    for (; u + ND_UNROLL <= u1; u += ND_UNROLL) {
        float a[ND_UNROLL];
#pragma unroll
        for (int e = 0; e < ND_UNROLL; ++e) a[e] = ld_stream<NT>(col + (size_t)(u + e) * stride);
#pragma unroll
        for (int e = 0; e < ND_UNROLL; ++e) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(a[e], sv[(u + e) * K + q], acc[q]);
        }
    }
    for (; u < u1; ++u) {
        const float a = ld_stream<NT>(col + (size_t)u * stride);
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = fmaf(a, sv[u * K + q], acc[q]);
    }
}

template <bool NT, int ND_UNROLL>
__device__ __forceinline__ void prefetch_strided(const float* __restrict__ col, size_t stride, int u0, int u1, float (&pre)[ND_UNROLL]) {
#pragma unroll
    for (int e = 0; e < ND_UNROLL; ++e) pre[e] = (u0 + e < u1) ? ld_stream<NT>(col + (size_t)(u0 + e) * stride) : 0.0f;
}

// sum of acc over the NW waves of the workgroup, result in wave 0 (red: (NW-1) * 64 * K floats)
template <int K>
__device__ __forceinline__ void reduce_waves(float (&acc)[K], float* __restrict__ red) {
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (nw == 1) return;
    if (w) {
#pragma unroll
        for (int q = 0; q < K; ++q) red[((w - 1) * 64 + lane) * K + q] = acc[q];
    }
    __syncthreads();
    if (w == 0) {
        for (int o = 0; o < nw - 1; ++o) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] += red[(o * 64 + lane) * K + q];
        }
    }
}

// Up sweep, one tree level:  b'_s = b_s - (slots at own_i)  [stored for the down sweep];
//                            upd_i = W_i b'_s + (slots at bnd_i)  -> the parent's slots
template <int K, bool NT>
__global__ __launch_bounds__(1024) void k_nd_up(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                const unsigned char* __restrict__ mask, const int* __restrict__ ppos,
                                                const float* __restrict__ wf, const float* __restrict__ b_in,
                                                float* __restrict__ bprime, float* slots, int s_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;
    float* red = sm + (size_t)s_cap * K;
    const Tile t = load_tile(tiles, blockIdx.x);
    if (t.store == 3) return;                   // padding of the XCD-aware tile order
    if (t.store == 2) { store_bprime<K>(t, perm, mask, slots, b_in, bprime); return; }
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = t.row0 + lane, s = t.s, b = t.b;
    const int chunk = (s + nw - 1) / nw, j0 = min(s, w * chunk), j1 = min(s, j0 + chunk);
    const bool row = i < b;
    // everything that does not depend on this level's arithmetic is requested up front
    const float* __restrict__ col = wf + t.w_off + (row ? i : 0);
    float pre[NdUnroll<K>::up];
    prefetch_strided<NT>(col, (size_t)b, j0, row ? j1 : j0, pre);
    int pp = 0;
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    if (w == 0 && row) {
        pp = ppos[t.bnd_off + i];
        if (!t.leaf) pull_slots<K>(slots, mask, (size_t)(t.front_off + s + i), t.arity, pass);
    }
    fill_bprime<K>(t, perm, mask, slots, b_in, bprime, sb);
    __syncthreads();
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    if (row) dot_strided<K, NT>(col, (size_t)b, j0, j1, sb, pre, acc);
    reduce_waves<K>(acc, red);
    if (w == 0 && row) {
        const size_t dst = ((size_t)(t.pfront_off + pp) * t.arity + t.cix) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) slots[dst + q] = acc[q] + pass[q];
    }
}

// boundary rows of the down sweep: xb_i -> children (forward tiles; blockDim.x rows per tile)
template <int K>
__device__ __forceinline__ void forward_rows(const Tile& t, const int* __restrict__ push_ptr, const int* __restrict__ push_tgt, float* xb) {
    const int i = t.row0 + threadIdx.x;
    if (i < t.b) {
        const size_t f = (size_t)(t.front_off + t.s + i);
        const int p0 = push_ptr[f], p1 = push_ptr[f + 1];
        float v[K];
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = xb[(size_t)(t.bnd_off + i) * K + q];
        push_down<K>(push_tgt, p0, p1, xb, v);
    }
}

// Down sweep, one tree level:  x_s = Finv_i b'_s - W_i^T xb_i;  x leaves in the caller's numbering and is pushed down
template <int K, bool NT>
__global__ __launch_bounds__(1024) void k_nd_down(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                  const int* __restrict__ push_ptr, const int* __restrict__ push_tgt,
                                                  const float* __restrict__ finv, const float* __restrict__ wb,
                                                  const float* __restrict__ bprime, float* xb, float* __restrict__ x_out,
                                                  int s_cap, int b_cap, RootFill rf) {   // xb: own rows read, children's rows written
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;
    float* sx = sm + (size_t)s_cap * K;
    float* red = sx + (size_t)b_cap * K;
    const Tile t = load_tile(tiles, blockIdx.x);
    if (t.forward == 2) return;                 // padding of the XCD-aware tile order
    if (t.forward) { forward_rows<K>(t, push_ptr, push_tgt, xb); return; }
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = t.s, b = t.b, L = s + b;
    const int j = t.row0 + lane;
    const bool row = j < s;
    const int chunk = (L + nw - 1) / nw, t0 = min(L, w * chunk), t1 = min(L, t0 + chunk);
    const float* __restrict__ fcol = finv + t.finv_off + (row ? j : 0);
    const float* __restrict__ wcol = wb + t.w_off + (row ? j : 0);
    const int f0 = min(t0, s), f1 = min(t1, s), g0 = max(t0, s) - s, g1 = max(t1, s) - s;   // Finv part, W part
    float pre_f[NdUnroll<K>::down], pre_w[NdUnroll<K>::down];
    prefetch_strided<NT>(fcol, (size_t)s, f0, row ? f1 : f0, pre_f);
    prefetch_strided<NT>(wcol, (size_t)s, g0, row ? g1 : g0, pre_w);
    int p0 = 0, p1 = 0;
    size_t g = 0;
    if (w == 0 && row) {
        g = (size_t)perm[t.own_start + j];
        if (!t.leaf) { p0 = push_ptr[t.front_off + j]; p1 = push_ptr[t.front_off + j + 1]; }
    }
    if (rf.slots && t.pfront_off < 0) root_bprime<K>(t, rf, sb);
    else
    for (int u = threadIdx.x; u < s; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sb[u * K + q] = bprime[(size_t)(t.own_start + u) * K + q];
    }
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sx[i * K + q] = -xb[(size_t)(t.bnd_off + i) * K + q];
    }
    __syncthreads();
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    if (row) {
        dot_strided<K, NT>(fcol, (size_t)s, f0, f1, sb, pre_f, acc);
        dot_strided<K, NT>(wcol, (size_t)s, g0, g1, sx, pre_w, acc);
    }
    reduce_waves<K>(acc, red);
    if (w == 0 && row) {
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[g * K + q] = acc[q];
        push_down<K>(push_tgt, p0, p1, xb, acc);
    }
}

// ---- lanes along the reduction ----------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_sum_step(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(m);
}
// sum over the 64 lanes, valid in lane 63 (fixed order: bitwise reproducible)
__device__ __forceinline__ float wave_sum63(float v) {
    v = dpp_sum_step<0xB1, 0xf>(v);
    v = dpp_sum_step<0x4E, 0xf>(v);
    v = dpp_sum_step<0x141, 0xf>(v);
    v = dpp_sum_step<0x140, 0xf>(v);
    v = dpp_sum_step<0x142, 0xa>(v);
    v = dpp_sum_step<0x143, 0xc>(v);
    return v;
}

// One batch = ND_E x 256 reduction steps of the wave's ND_ROWS rows (row r at base + r * stride_rows): every lane loads
// 16 bytes = 4 consecutive steps per request (1 KB contiguous per wave and request; rows are only 4-byte aligned, which
// global_load_dwordx4 accepts), all requests of a batch are independent.
#ifndef LS_ND_E
#define LS_ND_E 2
#endif
constexpr int ND_E = LS_ND_E;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool NT>
__device__ __forceinline__ void rows_load(const float* __restrict__ base, size_t stride_rows, int nrows, int len, int t0,
                                          f4u (&a)[ND_ROWS][ND_E]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < ND_ROWS; ++r) {
#pragma unroll
        for (int e = 0; e < ND_E; ++e) {
            const int t = t0 + (e * 64 + lane) * 4;        // (the last quad of a row may reach 12 bytes into the next row /
            f4u z = {0.f, 0.f, 0.f, 0.f};                  //  the array's slack: those components are never used)
            const f4u* p = reinterpret_cast<const f4u*>(base + (size_t)r * stride_rows + t);
            if constexpr (NT) a[r][e] = (r < nrows && t < len) ? __builtin_nontemporal_load(p) : z;
            else a[r][e] = (r < nrows && t < len) ? *p : z;
        }
    }
}
template <int K>
__device__ __forceinline__ void rows_fma(const f4u (&a)[ND_ROWS][ND_E], int len, int t0, const float* __restrict__ sv,
                                         float (&acc)[ND_ROWS][K]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < ND_E; ++e) {
        const int t = t0 + (e * 64 + lane) * 4;
        if (t < len) {
            // the lane's 4 consecutive vector entries = 4 K floats = K aligned 16-byte reads (t is a multiple of 4 and the vector starts on a
            // 16-byte boundary; the last quad may reach up to 3 entries past len: inside the workgroup's LDS, never used). With a test per
            // element the compiler read dword by dword at a lane stride of 12 K bytes: 4-way bank conflicts, 68 % of the kernel's LDS cycles
            // (SQ_LDS_BANK_CONFLICT, profiles/r04_pmc_sq_counters.txt).
            typedef float f4a __attribute__((ext_vector_type(4)));
            float v[4 * K];
            const f4a* __restrict__ pv = reinterpret_cast<const f4a*>(sv + (size_t)t * K);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const f4a w = pv[i];
                v[4 * i] = w[0]; v[4 * i + 1] = w[1]; v[4 * i + 2] = w[2]; v[4 * i + 3] = w[3];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (t + c < len) {
#pragma unroll
                    for (int r = 0; r < ND_ROWS; ++r) {
#pragma unroll
                        for (int q = 0; q < K; ++q) acc[r][q] = fmaf(a[r][e][c], v[c * K + q], acc[r][q]);
                    }
                }
            }
        }
    }
}
// acc[r][q] += sum_t row_r[t] * sv[t*K+q]; `first` holds the already loaded batch t0 = 0
template <int K, bool NT>
__device__ __forceinline__ void dot_rows(const float* __restrict__ base, size_t stride_rows, int nrows, int len,
                                         const float* __restrict__ sv, const f4u (&first)[ND_ROWS][ND_E], float (&acc)[ND_ROWS][K]) {
    rows_fma<K>(first, len, 0, sv, acc);
    for (int t0 = 256 * ND_E; t0 < len; t0 += 256 * ND_E) {
        f4u a[ND_ROWS][ND_E];
        rows_load<NT>(base, stride_rows, nrows, len, t0, a);
        rows_fma<K>(a, len, t0, sv, acc);
    }
}

// lane (lane0 + r) of the wave takes the total of row r
template <int K>
__device__ __forceinline__ void rows_to_lanes(const float (&acc)[ND_ROWS][K], int lane0, float (&mine)[K]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < ND_ROWS; ++r) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum63(acc[r][q])), 63));
            if (lane == lane0 + r) mine[q] = tot;
        }
    }
}

template <int K, bool NT>
__global__ __launch_bounds__(64 * ND_BW) void k_nd_up_b(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                        const unsigned char* __restrict__ mask, const int* __restrict__ ppos,
                                                        const float* __restrict__ wb, const float* __restrict__ b_in,
                                                        float* __restrict__ bprime, float* slots, int s_cap, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;
    const Tile t = load_tile(tiles, blockIdx.x);
    if (t.store == 3) return;                   // padding of the XCD-aware tile order
    if (t.store == 2) { store_bprime<K>(t, perm, mask, slots, b_in, bprime); return; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = t.s, b = t.b;
    const int iw = t.row0 + w * ND_ROWS * chunks;              // this wave's rows iw .. iw + ND_ROWS * chunks
    const int wrows = max(0, min(ND_ROWS * chunks, b - iw));
    const float* __restrict__ wrow = wb + t.w_off + (size_t)iw * s;
    f4u first[ND_ROWS][ND_E];
    rows_load<NT>(wrow, (size_t)s, min(wrows, ND_ROWS), s, 0, first);   // in flight while b' is assembled
    // per-row epilogue data (lane r of the wave serves row iw + r), requested before anything else
    int pp = 0;
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    if (lane < wrows) {
        pp = ppos[t.bnd_off + iw + lane];
        if (!t.leaf) pull_slots<K>(slots, mask, (size_t)(t.front_off + s + iw + lane), t.arity, pass);
    }
    fill_bprime<K>(t, perm, mask, slots, b_in, bprime, sb);
    __syncthreads();
    float mine[K];
#pragma unroll
    for (int q = 0; q < K; ++q) mine[q] = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        const int nrows = max(0, min(ND_ROWS, wrows - c * ND_ROWS));
        if (nrows == 0) break;
        float acc[ND_ROWS][K];
#pragma unroll
        for (int r = 0; r < ND_ROWS; ++r) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[r][q] = 0.0f;
        }
        const float* __restrict__ rowc = wrow + (size_t)c * ND_ROWS * s;
        if (c) rows_load<NT>(rowc, (size_t)s, nrows, s, 0, first);
        dot_rows<K, NT>(rowc, (size_t)s, nrows, s, sb, first, acc);
        rows_to_lanes<K>(acc, c * ND_ROWS, mine);
    }
    if (lane < wrows) {
        const size_t dst = ((size_t)(t.pfront_off + pp) * t.arity + t.cix) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) slots[dst + q] = mine[q] + pass[q];
    }
}

template <int K, bool NT>
__global__ __launch_bounds__(64 * ND_BW) void k_nd_down_b(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                          const int* __restrict__ push_ptr, const int* __restrict__ push_tgt,
                                                          const float* __restrict__ finv, const float* __restrict__ wf,
                                                          const float* __restrict__ bprime, float* xb, float* __restrict__ x_out,
                                                          int s_cap, int b_cap, int chunks, RootFill rf) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;
    float* sx = sm + (((size_t)s_cap * K + 3) & ~(size_t)3);      // (16-byte aligned: rows_fma reads the vectors in quads)
    const Tile t = load_tile(tiles, blockIdx.x);
    if (t.forward == 2) return;                 // padding of the XCD-aware tile order
    if (t.forward) { forward_rows<K>(t, push_ptr, push_tgt, xb); return; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = t.s, b = t.b;
    const int jw = t.row0 + w * ND_ROWS * chunks;
    const int wrows = max(0, min(ND_ROWS * chunks, s - jw));
    const float* __restrict__ frow = finv + t.finv_off + (size_t)jw * s;
    const float* __restrict__ wrow = wf + t.w_off + (size_t)jw * b;
    f4u first_f[ND_ROWS][ND_E], first_w[ND_ROWS][ND_E];
    rows_load<NT>(frow, (size_t)s, min(wrows, ND_ROWS), s, 0, first_f);          // in flight while the vectors are staged
    rows_load<NT>(wrow, (size_t)b, min(wrows, ND_ROWS), b, 0, first_w);
    int p0 = 0, p1 = 0;
    size_t g = 0;
    if (lane < wrows) {
        g = (size_t)perm[t.own_start + jw + lane];
        if (!t.leaf) { p0 = push_ptr[t.front_off + jw + lane]; p1 = push_ptr[t.front_off + jw + lane + 1]; }
    }
    if (rf.slots && t.pfront_off < 0) root_bprime<K>(t, rf, sb);
    else
    for (int u = threadIdx.x; u < s; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sb[u * K + q] = bprime[(size_t)(t.own_start + u) * K + q];
    }
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sx[i * K + q] = -xb[(size_t)(t.bnd_off + i) * K + q];
    }
    __syncthreads();
    float mine[K];
#pragma unroll
    for (int q = 0; q < K; ++q) mine[q] = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        const int nrows = max(0, min(ND_ROWS, wrows - c * ND_ROWS));
        if (nrows == 0) break;
        float acc[ND_ROWS][K];
#pragma unroll
        for (int r = 0; r < ND_ROWS; ++r) {
#pragma unroll
            for (int q = 0; q < K; ++q) acc[r][q] = 0.0f;
        }
        const float* __restrict__ fc = frow + (size_t)c * ND_ROWS * s;
        const float* __restrict__ wc = wrow + (size_t)c * ND_ROWS * b;
        if (c) { rows_load<NT>(fc, (size_t)s, nrows, s, 0, first_f); rows_load<NT>(wc, (size_t)b, nrows, b, 0, first_w); }
        dot_rows<K, NT>(fc, (size_t)s, nrows, s, sb, first_f, acc);
        dot_rows<K, NT>(wc, (size_t)b, nrows, b, sx, first_w, acc);
        rows_to_lanes<K>(acc, c * ND_ROWS, mine);
    }
    if (lane < wrows) {
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[g * K + q] = mine[q];
        push_down<K>(push_tgt, p0, p1, xb, mine);
    }
}

// ---- small nodes (the lower tree levels): the node's whole matrix staged in LDS ------------------------------------
// A leaf front is a few thousand numbers. Walking it row-per-lane from HBM means s (+ b) dependent, partly filled
// 256-byte loads per wave; instead the workgroup copies the node's contiguous matrix block(s) into LDS with one burst
// of independent, fully coalesced loads (every lane busy whatever s and b are) and multiplies out of LDS, a row per
// thread, conflict free (consecutive threads read consecutive words).
__device__ __forceinline__ void stage_block(const float* __restrict__ src, int n, float* __restrict__ dst) {
    constexpr int U = 16;     // loads in flight per thread: the copy is a burst, not a chain of load -> store pairs
    for (int e0 = threadIdx.x; e0 < n; e0 += blockDim.x * U) {
        float v[U];
#pragma unroll
        for (int r = 0; r < U; ++r) { const int e = e0 + r * blockDim.x; v[r] = e < n ? src[e] : 0.0f; }
#pragma unroll
        for (int r = 0; r < U; ++r) { const int e = e0 + r * blockDim.x; if (e < n) dst[e] = v[r]; }
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_nd_up_s(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                 const unsigned char* __restrict__ mask, const int* __restrict__ ppos,
                                                 const float* __restrict__ wf, const float* __restrict__ b_in,
                                                 float* __restrict__ bprime, float* slots, int s_cap, int b_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;                                  // s_cap * K
    float* mw = sm + (size_t)s_cap * K;              // s * b: W^T, mw[j * b + i]
    const Tile t = load_tile(tiles, blockIdx.x);
    const int s = t.s, b = t.b, i = threadIdx.x;
    stage_block(wf + t.w_off, s * b, mw);
    int pp = 0;
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    if (i < b) {
        pp = ppos[t.bnd_off + i];
        if (!t.leaf) pull_slots<K>(slots, mask, (size_t)(t.front_off + s + i), t.arity, pass);
    }
    fill_bprime<K>(t, perm, mask, slots, b_in, bprime, sb);
    __syncthreads();
    if (i < b) {
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
#pragma unroll 4
        for (int j = 0; j < s; ++j) {
            const float a = mw[j * b + i];
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(a, sb[j * K + q], acc[q]);
        }
        const size_t dst = ((size_t)(t.pfront_off + pp) * t.arity + t.cix) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) slots[dst + q] = acc[q] + pass[q];
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_nd_down_s(const Tile* __restrict__ tiles, const int* __restrict__ perm,
                                                   const int* __restrict__ push_ptr, const int* __restrict__ push_tgt,
                                                   const float* __restrict__ finv, const float* __restrict__ wb,
                                                   const float* __restrict__ bprime, float* xb, float* __restrict__ x_out,
                                                   int s_cap, int b_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;                                  // s_cap * K
    float* sx = sm + (size_t)s_cap * K;              // b_cap * K
    float* mf = sx + (size_t)b_cap * K;              // s * s: Finv, mf[t * s + j]
    const Tile t = load_tile(tiles, blockIdx.x);
    if (t.forward == 2) return;                 // padding of the XCD-aware tile order
    if (t.forward) { forward_rows<K>(t, push_ptr, push_tgt, xb); return; }
    const int s = t.s, b = t.b, j = threadIdx.x;
    float* mw = mf + (size_t)s * s;                  // b * s: W, mw[i * s + j]
    stage_block(finv + t.finv_off, s * s, mf);
    stage_block(wb + t.w_off, b * s, mw);
    int p0 = 0, p1 = 0;
    size_t g = 0;
    if (j < s) {
        g = (size_t)perm[t.own_start + j];
        if (!t.leaf) { p0 = push_ptr[t.front_off + j]; p1 = push_ptr[t.front_off + j + 1]; }
    }
    for (int u = threadIdx.x; u < s; u += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sb[u * K + q] = bprime[(size_t)(t.own_start + u) * K + q];
    }
    for (int i = threadIdx.x; i < b; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sx[i * K + q] = -xb[(size_t)(t.bnd_off + i) * K + q];
    }
    __syncthreads();
    // the boundary rows hand x down while the own rows multiply (leaf levels have nothing to hand down)
    if (!t.leaf) {
        for (int i = threadIdx.x; i < b; i += blockDim.x) {
            const size_t f = (size_t)(t.front_off + s + i);
            const int q0 = push_ptr[f], q1 = push_ptr[f + 1];
            float v[K];
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = -sx[i * K + q];
            push_down<K>(push_tgt, q0, q1, xb, v);
        }
    }
    if (j < s) {
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
#pragma unroll 4
        for (int u = 0; u < s; ++u) {
            const float a = mf[u * s + j];
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(a, sb[u * K + q], acc[q]);
        }
#pragma unroll 4
        for (int i = 0; i < b; ++i) {
            const float a = mw[i * s + j];
#pragma unroll
            for (int q = 0; q < K; ++q) acc[q] = fmaf(a, sx[i * K + q], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[g * K + q] = acc[q];
        push_down<K>(push_tgt, p0, p1, xb, acc);
    }
}

// ---- tiny nodes (the deepest levels): several nodes per wave --------------------------------------------------------
// A node with a dozen rows leaves most of a 64-lane wave idle and there are tens of thousands of them. A packed tile is
// a run of up to 8 consecutive nodes of one level whose rows together fill a wave: lane -> (node, row). Consecutive
// nodes own consecutive vertex ranges and consecutive boundary vectors, so b' / x_bnd of the whole tile are staged
// with contiguous loads.
constexpr int ND_PACK = 8;
struct NodeP { int s, b, own_start, bnd_off, front_off, pfront_off, cix, pad; long long finv_off, w_off; };
struct alignas(64) PackedTile {
    int n, leaf, arity, pad;
    int row0[ND_PACK + 1];      // prefix of the packed row dimension (up: boundary rows, down: own rows)
    int sb0[ND_PACK + 1];       // prefix of s: where a node's b' starts in the tile's LDS vector
    int sx0[ND_PACK + 1];       // prefix of b: where a node's x_bnd starts
    NodeP d[ND_PACK];
};

__device__ __forceinline__ int find_group(const int* __restrict__ prefix, int n, int r) {
    int g = 0;
#pragma unroll
    for (int c = 1; c < ND_PACK; ++c) g += (c < n && r >= prefix[c]) ? 1 : 0;
    return g;
}

template <int K, int A>
__device__ __forceinline__ void packed_fill(const PackedTile& t, const int* __restrict__ perm, const unsigned char* __restrict__ mask,
                                            const float* slots, const float* __restrict__ b_in, float* __restrict__ bprime,
                                            float* __restrict__ sb) {
    const int S = t.sb0[t.n], first = t.d[0].own_start;
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        const int h = find_group(t.sb0, t.n, r);
        const int j = r - t.sb0[h];
        const size_t g = perm ? (size_t)perm[first + r] : (size_t)(first + r);            // own ranges of consecutive nodes are consecutive
        float v[K];
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = b_in[g * K + q];
        if (!t.leaf) {
            const size_t f = (size_t)(t.d[h].front_off + j);
            float raw[A * K], u[K];
            load_slots<K, A>(slots, f, raw);
            sum_slots<K, A>(raw, mask[f], u);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] -= u[q];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) { sb[r * K + q] = v[q]; bprime[(size_t)(first + r) * K + q] = v[q]; }
    }
}

template <int K>
__global__ __launch_bounds__(64) void k_nd_up_p(const PackedTile* __restrict__ tiles, const int* __restrict__ perm,
                                                const unsigned char* __restrict__ mask, const int* __restrict__ ppos,
                                                const float* __restrict__ wf, const float* __restrict__ b_in,
                                                float* __restrict__ bprime, float* slots) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;
    const PackedTile& t = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const bool row = lane < t.row0[t.n];
    const int g = find_group(t.row0, t.n, lane);
    const NodeP d = t.d[g];
    const int i = lane - t.row0[g];
    const float* __restrict__ col = wf + d.w_off + (row ? i : 0);
    float pre[LS_ND_UNROLL];
    prefetch_strided<false>(col, (size_t)d.b, 0, row ? d.s : 0, pre);
    int pp = 0;
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    if (row) {
        pp = ppos[d.bnd_off + i];
        if (!t.leaf) pull_slots<K>(slots, mask, (size_t)(d.front_off + d.s + i), t.arity, pass);
    }
    if (t.arity == 4) packed_fill<K, 4>(t, perm, mask, slots, b_in, bprime, sb);
    else if (t.arity == 2) packed_fill<K, 2>(t, perm, mask, slots, b_in, bprime, sb);
    else packed_fill<K, 8>(t, perm, mask, slots, b_in, bprime, sb);
    __syncthreads();
    if (row) {
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
        dot_strided<K, false>(col, (size_t)d.b, 0, d.s, sb + (size_t)t.sb0[g] * K, pre, acc);
        const size_t dst = ((size_t)(d.pfront_off + pp) * t.arity + d.cix) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) slots[dst + q] = acc[q] + pass[q];
    }
}

template <int K>
__global__ __launch_bounds__(64) void k_nd_down_p(const PackedTile* __restrict__ tiles, const int* __restrict__ perm,
                                                  const int* __restrict__ push_ptr, const int* __restrict__ push_tgt,
                                                  const float* __restrict__ finv, const float* __restrict__ wb,
                                                  const float* __restrict__ bprime, float* xb, float* __restrict__ x_out, int s_sum_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb = sm;                                     // b' of the tile's own rows
    float* sx = sm + (size_t)s_sum_cap * K;             // -x_bnd of the tile's boundary rows
    const PackedTile& t = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const bool row = lane < t.row0[t.n];
    const int g = find_group(t.row0, t.n, lane);
    const NodeP d = t.d[g];
    const int j = lane - t.row0[g];
    const float* __restrict__ fcol = finv + d.finv_off + (row ? j : 0);
    const float* __restrict__ wcol = wb + d.w_off + (row ? j : 0);
    float pre_f[LS_ND_UNROLL], pre_w[LS_ND_UNROLL];            // (a 64-thread workgroup: registers are plentiful here)
    prefetch_strided<false>(fcol, (size_t)d.s, 0, row ? d.s : 0, pre_f);
    prefetch_strided<false>(wcol, (size_t)d.s, 0, row ? d.b : 0, pre_w);
    int p0 = 0, p1 = 0;
    size_t gx = 0;
    if (row) {
        gx = (size_t)perm[d.own_start + j];
        if (!t.leaf) { p0 = push_ptr[d.front_off + j]; p1 = push_ptr[d.front_off + j + 1]; }
    }
    const int S = t.sb0[t.n], B = t.sx0[t.n], first = t.d[0].own_start, bfirst = t.d[0].bnd_off;
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sb[r * K + q] = bprime[(size_t)(first + r) * K + q];
    }
    for (int r = threadIdx.x; r < B; r += blockDim.x) {
#pragma unroll
        for (int q = 0; q < K; ++q) sx[r * K + q] = -xb[(size_t)(bfirst + r) * K + q];
    }
    __syncthreads();
    if (!t.leaf) {              // boundary rows hand x down
        for (int r = threadIdx.x; r < B; r += blockDim.x) {
            const int h = find_group(t.sx0, t.n, r);
            const size_t f = (size_t)(t.d[h].front_off + t.d[h].s + (r - t.sx0[h]));
            const int q0 = push_ptr[f], q1 = push_ptr[f + 1];
            float v[K];
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = -sx[r * K + q];
            push_down<K>(push_tgt, q0, q1, xb, v);
        }
    }
    if (row) {
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
        dot_strided<K, false>(fcol, (size_t)d.s, 0, d.s, sb + (size_t)t.sb0[g] * K, pre_f, acc);
        dot_strided<K, false>(wcol, (size_t)d.s, 0, d.b, sx + (size_t)t.sx0[g] * K, pre_w, acc);
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[gx * K + q] = acc[q];
        push_down<K>(push_tgt, p0, p1, xb, acc);
    }
}

}  // namespace ls
#include "nd_tier.h"
// (The levels above the tier as ONE persistent launch -- tree-local barriers, write-through hand-offs, register prefetch -- were built
// and measured in round 3: 166-207 us against 103-120 us for nine launches at 1M vertices. Archived: tools/archive/lab/.)
namespace ls {

// down tiles of a level: compute tiles, then forward tiles
struct LevelPlan { int up_first = 0, up_tiles = 0, up_nw = 1, down_first = 0, down_tiles = 0, down_nw = 1, s_cap = 0, b_cap = 0, up_b = 0, down_b = 0, up_chunks = 1, down_chunks = 1, up_s = 0, down_s = 0,
                   up_p = 0, down_p = 0, up_p_first = 0, up_p_tiles = 0, down_p_first = 0, down_p_tiles = 0, up_p_lds = 0, down_p_s = 0, down_p_lds = 0; };

}  // namespace ls

using namespace ls;

struct ls_direct {
    int device = 0, levels = 0, arity = 2, n_nodes = 0, kmax = 4;
    int64_t V = 0, n_bnd = 0, n_front = 0;
    int *perm = nullptr, *ppos = nullptr, *push_ptr = nullptr, *push_tgt = nullptr;
    unsigned char* mask = nullptr;
    Tile* tiles = nullptr;
    PackedTile* ptiles = nullptr;
    const float *finv = nullptr, *wf = nullptr, *wb = nullptr;   // owned by the caller
    const float *u4 = nullptr, *d4 = nullptr;                    // dense tier nodes: quad-interleaved streams (caller)
    const float* tri = nullptr;                                  // sparse leaves: packed triangles (caller)
    const int* sp_ptr = nullptr;                                 // sparse leaves: row pointers (caller)
    const SpEnt* sp_ent = nullptr;                               // sparse leaves: entries (caller)
    float* braw = nullptr;                                       // tier kernels: gathered right-hand side of the inner-node rows (V, k)
    float *bp = nullptr, *slots = nullptr, *xb = nullptr;        // b' (V, k); up-sweep slots (n_front, arity, k); x at boundaries (n_bnd, k)
    // bottom tier: levels [tier_root, levels) run as one launch per sweep, one workgroup per subtree (nd_tier.h)
    bool fuse_root = true;              // LS_ND_NO_FUSE_ROOT: the root keeps its up-sweep launch
    int upper_lo = 0;                   // rows [upper_lo, V) of the tree's numbering belong to the levels above the tier
    int tier_root = 0, tier_phases = 0, tier_wgs = 0, tier_region = 0, tier_vec = 0, tier_tri = 0, tier_waves = TIER_WAVES;
    TierItem* d_items = nullptr;
    int* pull = nullptr;                // tier up sweep: (front position - pull_base, child) -> child boundary entry, front positions of the tier's inner nodes
    int64_t pull_base = 0;
    TierWG* d_wgs = nullptr;
    hipEvent_t busy = nullptr;          // recorded after every solve: a solve on another stream waits for it (one workspace)
    hipStream_t last_stream = nullptr;
    bool used = false;
    std::vector<std::pair<void*, size_t>> tables;      // index tables and vectors of the handle (pool_take / pool_alloc in ls_direct_create; back to the pool with the handle)
    std::vector<void*> owned;           // device arrays adopted from ls_direct_factor (handed to the buffer pool / freed with the handle)
    std::vector<size_t> owned_bytes;
    // subtree sharding (one process per GPU): this handle runs the subtrees [sub_lo, sub_hi) of level `cut` and, replicated on
    // every rank, the levels above; the ranks meet once per solve in a sum over the slots of level cut - 1 (exch_f0 .. exch_f1)
    int shard_rank = 0, shard_count = 1, cut = 0;
    int64_t exch_f0 = 0, exch_f1 = 0;
    std::vector<unsigned char> owned_rows;    // caller's numbering: 1 = this rank is the designated owner of the row's x
    int tier_xcd = 1;                   // LS_ND_XCD=0 at creation: plain workgroup -> subtree order in the tier kernels
    // cache policy of the read-once factor streams (common.h, ld_stream): non-temporal when the factor cannot stay cache resident anyway
    bool nt_levels = false, nt_tier = false, nt_rule[2] = {false, false};
    double factor_s[3] = {0, 0, 0};     // ls_direct_factor: symbolic analysis, layout / sparse tables, numeric factorisation
    double plan_q[4] = {0, 0, 0, 0};    // ls_direct_factor: the dissection's ordering rule, factor numbers per vertex, spread, the other plan's numbers (nd_plan.h)
    std::vector<LevelPlan> plan;
    int64_t factor_entries = 0, words_up = 0, words_down = 0;    // 4-byte words of factor data per solve / per sweep
    int profile = 0;
    std::vector<hipEvent_t> ev;
    // "profile" = 3: one event in front of every launch of a solve (+ one behind the last)
    struct LaunchMeta { int lo, hi, sweep; };         // tree levels [lo, hi] the launch runs; sweep 0 up, 1 down, 2 both
    std::vector<hipEvent_t> lev;
    std::vector<LaunchMeta> lmeta;
    std::vector<double> launch_ms;
    std::vector<int64_t> lvl_up, lvl_down;            // 4-byte words of factor data per tree level and sweep
    std::vector<int64_t> lvl_rows, lvl_bnd;           // own rows / boundary entries per tree level (the vectors' share of a launch's bytes)
    std::vector<int64_t> lvl_idx_up, lvl_idx_down;    // bytes of STATIC index data a level's launch reads per sweep besides perm: tile / item records, mask, ppos, pull, push lists
    double tier_balance[4] = {0, 0, 0, 0};            // factor words of the tier's subtrees (one workgroup each): max and mean, up sweep / down sweep
    double prof_ms[3] = {0, 0, 0};     // up sweep, down sweep, 0 (last profiled solve)
#ifdef LS_TIER_STAMPS
    long long* stamps = nullptr;       // experiments build: 2 sweeps x tier_wgs x tier_waves x TIER_STAMP_SLOTS clock stamps of the last solve with "profile" = 2
#endif
};

static int env_int(const char* name, int dflt) { const char* e = getenv(name); const int v = e ? atoi(e) : 0; return v > 0 ? v : dflt; }

// waves per workgroup of the row-per-lane kernels: about LS_ND_STEPS reduction steps per wave where 16 waves allow it
static int pick_nw(int len, bool up_sweep = false) {
    // read at every create: tests and tuning runs toggle it
    const int target = up_sweep ? env_int("LS_ND_STEPS_UP", 32) : env_int("LS_ND_STEPS", 128);
    int nw = 1;
    while (nw < 16 && len > nw * target) nw *= 2;
    return nw;
}

int ls_direct_adopt(ls_direct* d, void* const* owned, const size_t* owned_bytes, int n_owned, const double* seconds3, const double* quality4);

// ---- pool of large device buffers --------------------------------------------------------------------------------------------------
// A remesh loop (scripts/main.py:137-169) destroys a solver and constructs one of nearly the same size again and again. The runtime gives
// freed device memory back lazily: at 4M vertices every second or third construction of a process stalled 0.45-0.7 s inside ONE hipMalloc
// (the 9.3 GB of fp64 fronts, or the next buffer after it: profiles/r03_run5_constructor_times.txt) -- four times the whole constructor.
// The constructor's scratch (fronts and work arrays: 3-4 GB at 1M, 14 GB at 4M) and the handle's factor arrays therefore go back to this
// pool instead of hipFree, and allocations of >= 64 MB take the best fit (at most 1.5x + 64 MB larger). What the pool may hold is
// bounded PER DEVICE: LS_POOL_GB (default 24, 0 = no pool) and never more than a quarter of the device's memory (torch's caching
// allocator cannot see or reclaim what sits here); oldest out first. An allocation of the library that fails empties the pool of its
// device and is tried once more (pool_alloc below), ls_release_scratch() empties it on request, and the Python layer calls that and
// repeats the call once when torch itself runs out of memory inside compute_matrix / a solve / the normals (largesteps._native.retry_on_oom).
namespace ls {
// Two non-blocking streams per device for work that must not queue behind the caller's stream: which = 0 the handle's tables go up
// while that stream factorises; which = 1 the fp32 conversion of a finished tree level runs beside the chain of small launches of
// the level above (two streams: the uploads must not queue behind conversions that wait for the factorisation either).
// Created on first use, kept for the life of the process. Ordering against the caller's stream is always by events. The two streams are
// shared by every construction on the device: two host threads that construct on one device at the same moment queue their uploads and
// conversions behind each other's (results stay correct -- ordering is by events -- but the two constructors are coupled in time).
hipStream_t side_stream(int device, int which) {
    static std::mutex mu;
    static hipStream_t streams[64][2] = {{nullptr, nullptr}};
    if (device < 0 || device >= 64 || which < 0 || which > 1) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[device][which]) {
        DeviceGuard dg(device);
        if (dg.err != hipSuccess || hipStreamCreateWithFlags(&streams[device][which], hipStreamNonBlocking) != hipSuccess) streams[device][which] = nullptr;
    }
    return streams[device][which];
}
namespace {
struct DevicePool {
    std::mutex mu;
    struct Entry { void* p; size_t bytes; int device; };
    std::vector<Entry> held;
    static size_t cap(int device) {              // bytes the pool may hold on one device
        const char* e = getenv("LS_POOL_GB");
        double gb = e ? atof(e) : 24.0;        // (a 4M-vertex construction leaves 14 GB of scratch + 2.4 GB of factor: with 16 the fronts were evicted every time)
        if (!(gb > 0.0)) return 0;
        // the device's memory size is asked for once (hipMemGetInfo is a driver call, and every buffer handed back comes through here:
        // ~40 per construction since the small arrays are pooled too)
        static std::mutex mu;
        static double total_gb[64];
        double t = 0.0;
        if (device >= 0 && device < 64) {
            std::lock_guard<std::mutex> lk(mu);
            if (total_gb[device] == 0.0) {
                size_t free_b = 0, total_b = 0;
                DeviceGuard dg(device);
                total_gb[device] = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) ? (double)total_b / 1073741824.0 : -1.0;
            }
            t = total_gb[device];
        }
        if (t > 0.0) gb = std::min(gb, t / 4.0);
        return (size_t)(gb * 1073741824.0);
    }
};
DevicePool g_pool;
}  // namespace

void* pool_take(int device, size_t bytes, size_t* capacity) {
    if (bytes < POOL_FROM) return nullptr;
    std::lock_guard<std::mutex> g(g_pool.mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.held.size(); ++i) {
        const DevicePool::Entry& e = g_pool.held[(size_t)i];
        if (e.device == device && e.bytes >= bytes && e.bytes <= bytes + bytes / 2 + POOL_SLACK && (best < 0 || e.bytes < g_pool.held[(size_t)best].bytes)) best = i;
    }
    if (best < 0) return nullptr;
    void* p = g_pool.held[(size_t)best].p;
    if (capacity) *capacity = g_pool.held[(size_t)best].bytes;
    g_pool.held.erase(g_pool.held.begin() + best);
    return p;
}

bool pool_give(int device, void* p, size_t bytes) {
    if (!p || bytes < POOL_FROM) return false;
    const size_t limit = DevicePool::cap(device);
    if (bytes > limit) return false;
    std::lock_guard<std::mutex> g(g_pool.mu);
    size_t total = bytes;
    for (const DevicePool::Entry& e : g_pool.held) if (e.device == device) total += e.bytes;
    size_t count = 1;
    for (const DevicePool::Entry& e : g_pool.held) if (e.device == device) ++count;
    for (size_t i = 0; (total > limit || count > 512) && i < g_pool.held.size();) {    // this device's oldest out first
        if (g_pool.held[i].device != device) { ++i; continue; }
        --count;
        total -= g_pool.held[i].bytes;
        DeviceGuard dg(device);
        (void)hipFree(g_pool.held[i].p);
        g_pool.held.erase(g_pool.held.begin() + (long)i);
    }
    g_pool.held.push_back(DevicePool::Entry{p, bytes, device});
    return true;
}

// hipMalloc that gives the pool's memory back before it gives up: a remesh to a size outside the pool's fit window, or a torch /
// renderer allocation next to a full pool, must not fail for memory the library is only keeping warm
// LS_PLAN_TIMING: how many allocations missed the pool and what hipMalloc cost them (read and reset by ls_direct_factor's last lap)
std::atomic<long long> g_malloc_calls{0}, g_malloc_us{0}, g_malloc_bytes{0};
hipError_t pool_alloc(int device, void** p, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, bytes);
    g_malloc_calls += 1; g_malloc_bytes += (long long)bytes;
    g_malloc_us += (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (e != hipErrorOutOfMemory && e != hipErrorMemoryAllocation) return e;
    (void)hipGetLastError();
    (void)ls_release_scratch(device);
    return hipMalloc(p, bytes);
}
}  // namespace ls

extern "C" int ls_release_scratch(int device) {            // device < 0: every device
    std::lock_guard<std::mutex> g(ls::g_pool.mu);
    for (size_t i = 0; i < ls::g_pool.held.size();) {
        if (device < 0 || ls::g_pool.held[i].device == device) {
            ls::DeviceGuard dg(ls::g_pool.held[i].device);
            (void)hipFree(ls::g_pool.held[i].p);
            ls::g_pool.held.erase(ls::g_pool.held.begin() + (long)i);
        } else ++i;
    }
    return LS_OK;
}

static int env_int0(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// Bottom tier (nd_tier.h): cut the subtrees rooted at level `root` into wave-sized items. Returns the LDS floats a wave
// needs (0 = the tier does not fit), fills items / wgs.
static size_t plan_tier(const std::vector<NodeD>& nd, const std::vector<int64_t>& level_off, int levels, int arity, int root, int64_t q_lo,
                        int64_t q_hi, std::vector<TierItem>& items, std::vector<TierWG>& wgs, int& vec_floats, int& tri_floats, int waves = TIER_WAVES) {
    const int H = levels - root;
    items.clear(); wgs.assign((size_t)(q_hi - q_lo), TierWG());
    int vec_need = 0, pbuf_need = 0, leaf_need = 0, tri_cap = 0;
    for (int64_t i = level_off[levels - 1]; i < level_off[levels]; ++i)
        if (nd[i].flags & NODE_SPARSE) {
            // the LDS triangle area holds whole 16-column chunks: the mat-vec reads element (j, c) of the LAST chunk for every c up to
            // 16 ceil(s / 16) - 1 (columns >= s meet zero vector entries; nd_tier.h, tri_chunk)
            const int sc = (nd[i].s + 15) & ~15;
            tri_cap = std::max(tri_cap, (sc * (sc + 1) / 2 + 3) & ~3);
        }
    tri_floats = tri_cap;
    for (int64_t q = q_lo; q < q_hi; ++q) {
        TierWG& g = wgs[(size_t)(q - q_lo)];
        g.up_split = g.down_split = g.up_leaf = g.down_leaf = 0;
        g.n_dense = 0; g.pad = 0;
        for (int t = 0; t < 2 * TIER_MAX_H; ++t) g.dense_rng[t] = 0;
        for (int lv = root, t = 0; lv < levels - 1; ++lv, ++t) {  // own rows of the inner nodes: contiguous per level (leaf level: not gathered)
            int64_t span = 1;
            for (int u = root; u < lv; ++u) span *= arity;
            const int64_t first = level_off[lv] + q * span;
            int lo = -1, hi = -1;
            for (int64_t i = first; i < first + span; ++i)
                if (!(nd[i].flags & NODE_SPARSE) && nd[i].s > 0) { if (lo < 0) lo = nd[i].own_start; hi = nd[i].own_start + nd[i].s; }
            if (lo >= 0) { g.dense_rng[2 * t] = lo; g.dense_rng[2 * t + 1] = hi - lo; g.n_dense += hi - lo; }
        }
        for (int sweep = 0; sweep < 2; ++sweep) {
            const bool up = sweep == 0;
            int* off = up ? g.up_off : g.down_off;
            for (int ph = 0; ph < H; ++ph) {
                off[ph] = (int)items.size();
                const int lv = up ? levels - 1 - ph : root + ph;
                int64_t span = 1;
                for (int t = root; t < lv; ++t) span *= arity;
                const int64_t first = level_off[lv] + q * span;
                int base = 0;
                bool sparse = false;
                for (int64_t i = first; i < first + span; ++i) {
                    const NodeD& n = nd[i];
                    if (n.flags & NODE_SPARSE) { sparse = true; continue; }
                    if (n.s == 0 && n.b == 0) continue;
                    base += std::max(1, div_up(up ? n.b : n.s, WAVE));
                }
                bool all_sparse = span > 0;
                for (int64_t i = first; i < first + span; ++i) {
                    const NodeD& n = nd[i];
                    if (!(n.flags & NODE_SPARSE) && (n.s || n.b)) all_sparse = false;
                    TierItem it;
                    memset(&it, 0, sizeof(it));
                    it.s = n.s; it.b = n.b; it.own_start = n.own_start; it.bnd_off = n.bnd_off; it.front_off = n.front_off;
                    it.pfront_off = n.pfront_off; it.cix = n.cix; it.flags = n.flags | (lv > root ? NODE_UPC : 0); it.finv_off = n.finv_off; it.w_off = n.w_off;
                    it.spb_off = n.spb_off; it.sps_off = n.sps_off; it.nparts = 1;
                    if (n.flags & NODE_SPARSE) {
                        items.push_back(it);
                        leaf_need = std::max(leaf_need, std::max(tri_cap, (4 * n.b + 3) & ~3) + 256);     // x_bnd (down) is staged in the triangle area
                        continue;
                    }
                    if (n.s == 0 && n.b == 0) continue;
                    // reduction in the padded index space of the quad-interleaved streams, parts cut at multiples of 4
                    const int rows = up ? n.b : n.s, L = up ? ((n.s + 3) & ~3) : ((n.s + 3) & ~3) + ((n.b + 3) & ~3);
                    int nparts = 1;
                    if (!sparse && base < waves) nparts = std::max(1, std::min(waves / std::max(base, 1), L / 16));
                    for (int r = 0; r < std::max(rows, 1); r += WAVE)
                        for (int pt = 0; pt < nparts; ++pt) {
                            it.row0 = r; it.r0 = (int)((int64_t)(L / 4) * pt / nparts) * 4; it.r1 = (int)((int64_t)(L / 4) * (pt + 1) / nparts) * 4;
                            it.part = pt; it.nparts = nparts;
                            items.push_back(it);
                            vec_need = std::max(vec_need, 4 * (it.r1 - it.r0 + 32));             // read as registers of 16 entries (two per batch of 8 quads)
                        }
                    if (nparts > 1) { if (up) g.up_split |= 1u << ph; else g.down_split |= 1u << ph; }
                }
                if (all_sparse) { if (up) g.up_leaf |= 1u << ph; else g.down_leaf |= 1u << ph; }
                else if (sparse) return 0;        // a level mixing sparse and dense leaves is not supported (the caller stores all leaves alike)
                const int n_items = (int)items.size() - off[ph];
                if (((up ? g.up_split : g.down_split) >> ph) & 1u) pbuf_need = std::max(pbuf_need, div_up(n_items, waves) * 256);
            }
            off[H] = (int)items.size();
        }
    }
    vec_floats = (vec_need + 3) & ~3;
    return (size_t)std::max(vec_floats + pbuf_need, leaf_need);
}


// Host only, called by ls_direct_factor BEFORE it lays the factor out: would a tier of `tier_levels` levels (sparse or dense leaves)
// fit the tier kernels' LDS budget on this tree? (s, b, own_start: per node id, 1-based, level-major.)
static size_t direct_tier_lds(int levels, int arity, const int* s, const int* b, const int* own_start, int tier_levels, bool sparse_leaves, int waves);
bool direct_tier_fits(int levels, int arity, const int* s, const int* b, const int* own_start, int tier_levels, bool sparse_leaves, int waves) {
    if (tier_levels <= 0) return true;
    if (tier_levels > levels || tier_levels > TIER_MAX_H) return false;
    const size_t bytes = direct_tier_lds(levels, arity, s, b, own_start, tier_levels, sparse_leaves, waves);
    return bytes && bytes <= (size_t)(waves == TIER_WAVES_FULL ? 160 : 150) * 1024;
}

// Host only: the dynamic LDS in bytes a tier workgroup of `waves` waves needs for this tree (0 = the tier cannot be planned). Exported as
// ls_direct_tier_lds_bytes: the CPU tests hold the 1M-vertex closed scan against the 160 KB of a CU with it (round 6: its leaves' 83 boundary
// rows once made the 16-wave plan 2668 floats per wave -- 4 % over -- and the solve fell back, silently, to 11 launches).
static size_t direct_tier_lds(int levels, int arity, const int* s, const int* b, const int* own_start, int tier_levels, bool sparse_leaves, int waves) {
    std::vector<int64_t> level_off((size_t)levels + 1);
    int64_t cnt = 1, off = 1;
    for (int lv = 0; lv <= levels; ++lv) { level_off[lv] = off; off += cnt; cnt *= arity; }
    const int64_t n_nodes = level_off[levels] - 1;
    std::vector<NodeD> nd((size_t)n_nodes + 1);
    memset(nd.data(), 0, nd.size() * sizeof(NodeD));
    const int root = levels - tier_levels;
    for (int lv = root; lv < levels; ++lv)
        for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i) {
            NodeD& q = nd[(size_t)i];
            q.s = s[i]; q.b = b[i]; q.own_start = own_start[i];
            const bool sparse = sparse_leaves && lv == levels - 1 && q.s >= 1;
            q.flags = (lv + 1 >= levels ? NODE_LEAF : 0) | (sparse ? NODE_SPARSE : NODE_QUAD);
        }
    std::vector<TierItem> items;
    std::vector<TierWG> wgs;
    int vec = 0, tri = 0;
    const size_t region = plan_tier(nd, level_off, levels, arity, root, 0, level_off[root + 1] - level_off[root], items, wgs, vec, tri, waves);
    return region ? ((region + 3) & ~(size_t)3) * sizeof(float) * waves : 0;
}

extern "C" int ls_direct_tier_lds_bytes(int levels, int arity, const int32_t* h_s, const int32_t* h_b, const int32_t* h_own_start, int tier_levels,
                                        int sparse_leaves, int waves, size_t* h_bytes) {
    LS_REQUIRE(h_s && h_b && h_own_start && h_bytes && levels >= 1 && levels <= 30 && (arity == 2 || arity == 4 || arity == 8) &&
               tier_levels >= 1 && tier_levels <= levels && tier_levels <= TIER_MAX_H && (waves == TIER_WAVES || waves == TIER_WAVES_WIDE || waves == TIER_WAVES_FULL),
               LS_E_INVALID, "ls_direct_tier_lds_bytes: bad argument (1 <= tier_levels <= min(levels, %d); waves 4, 8 or 16)", TIER_MAX_H);
    *h_bytes = direct_tier_lds(levels, arity, h_s, h_b, h_own_start, tier_levels, sparse_leaves != 0, waves);
    return LS_OK;
}

// One workgroup of SIXTEEN waves per CU walking a subtree one level taller (round 5): at 1M vertices the 256 subtrees below level 4
// instead of the 1024 below level 5 -- two launches less, the level's nodes cut into parts over the sixteen waves. Measured through
// the C ABI on planes (profiles/r05_tier16.txt): 1M 201.5 -> 197.2 us, 1.44M 376 -> 359, 2M 428 -> 415, 4M 719-729 -> 714; between
// 300k and 722k vertices even to worse (640k: 150.8 = 150.8; 722k: 157-162 -> 164), 810k 177.4 -> 171.5, below 300k the arity-8 trees stay
// ahead. So: arity 4, at least 8 levels,
// from 800k vertices, one GPU (a rank of a sharded solve keeps its own rule). LS_ND_TIER_WAVES = 4 / 8 / 16 overrides.
bool direct_tier_full16(int64_t V, int arity, int levels, int tier_levels, int shard_count, int tier_waves) {
    const int env = tier_waves > 0 ? tier_waves : env_int0("LS_ND_TIER_WAVES", 0);       // an explicit argument wins over the environment
    if (env == TIER_WAVES_FULL) return tier_levels >= 2;
    if (env == TIER_WAVES || env == TIER_WAVES_WIDE) return false;
    return shard_count <= 1 && arity == 4 && levels >= 8 && V >= 800000 && tier_levels == levels - 4 && tier_levels <= TIER_MAX_H;
}


extern "C" int ls_direct_create(const ls_direct_arrays* A, int device, void* stream, ls_direct** out) {
    LS_REQUIRE(A && out, LS_E_INVALID, "ls_direct_create: null argument");
    const int64_t V = A->V, n_bnd = A->n_bnd, n_front = A->n_front;
    const int levels = A->levels, arity = A->arity;
    const int64_t* h_nodes = A->h_nodes;
    const int32_t *h_perm = A->h_perm, *h_ppos = A->h_ppos, *h_push_ptr = A->h_push_ptr, *h_push_tgt = A->h_push_tgt;
    LS_REQUIRE(h_nodes && h_perm && h_push_ptr && V > 0 && levels >= 1 && levels <= 30 && n_bnd >= 0 && n_front >= V &&
               (arity == 2 || arity == 4 || arity == 8), LS_E_INVALID, "ls_direct_create: bad argument");
    LS_REQUIRE(V < INT32_MAX && n_bnd < INT32_MAX && n_front * arity < INT32_MAX, LS_E_OVERFLOW, "ls_direct_create: plan exceeds int32 offsets");
    *out = nullptr;
    std::vector<int64_t> level_off((size_t)levels + 1);
    {
        int64_t cnt = 1, off = 1;
        for (int lv = 0; lv <= levels; ++lv) {
            level_off[lv] = off; off += cnt; cnt *= arity;
            LS_REQUIRE(off < ((int64_t)1 << 30), LS_E_OVERFLOW, "ls_direct_create: tree too large");
        }
    }
    const int n_nodes = (int)(level_off[levels] - 1);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const bool timing = getenv("LS_PLAN_TIMING") != nullptr;          // host clock only, no synchronisation: where the handle's construction spends its time
    const auto t_create = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) fprintf(stderr, "[ls_direct_create] %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count()); };
    ls_direct* d = new ls_direct();
    d->device = device; d->levels = levels; d->arity = arity; d->n_nodes = n_nodes; d->V = V; d->n_bnd = n_bnd; d->n_front = n_front;
    d->finv = A->d_finv; d->wf = A->d_wf; d->wb = A->d_wb;
    d->u4 = A->d_u4; d->d4 = A->d_d4;
    d->tri = A->d_tri; d->sp_ptr = A->d_sp_ptr; d->sp_ent = (const SpEnt*)A->d_sp_ent;
    // subtree sharding: the cut level is the first one with at least `count` subtrees; rank r takes a contiguous share of them
    const int n_ranks = std::max(1, (int)A->shard_count), rank = std::min(std::max(0, (int)A->shard_rank), n_ranks - 1);
    int cut = 0;
    while (cut + 1 < levels && level_off[cut + 1] - level_off[cut] < n_ranks) ++cut;
    if (n_ranks == 1) cut = 0;
    const int64_t n_sub = level_off[cut + 1] - level_off[cut];
    const int64_t sub_lo = n_sub * rank / n_ranks, sub_hi = n_sub * (rank + 1) / n_ranks;
    auto active = [&](int64_t i, int lv) -> bool {
        if (lv < cut) return true;
        int64_t q = i - level_off[lv];
        for (int t = cut; t < lv; ++t) q /= arity;
        return q >= sub_lo && q < sub_hi;
    };
    d->shard_rank = rank; d->shard_count = n_ranks; d->cut = cut;
    std::vector<NodeDesc> nodes((size_t)n_nodes + 1);
    std::vector<NodeD> nd((size_t)n_nodes + 1);
    memset(nd.data(), 0, nd.size() * sizeof(NodeD));
    int64_t fe = 0, fe_up = 0, fe_down = 0;
    for (int i = 1; i <= n_nodes; ++i) {
        const int64_t* r = h_nodes + (size_t)i * LS_DIRECT_NODE_COLS;
        NodeDesc& n = nodes[i];
        n.s = (int)r[0]; n.b = (int)r[1]; n.own_start = (int)r[2]; n.bnd_off = (int)r[3]; n.front_off = (int)r[4];
        n.finv_off = r[5]; n.w_off = r[6]; n.parent = (int)r[7];
        const bool sparse = r[8] >= 0, quad = r[11] != 0;
        if (n.s < 0 || n.b < 0 || n.own_start < 0 || (int64_t)n.own_start + n.s > V || (int64_t)n.bnd_off + n.b > n_bnd ||
            (int64_t)n.front_off + n.s + n.b > n_front || (i == 1 ? n.b != 0 : (n.parent < 1 || n.parent >= i)) ||
            (sparse && (i < level_off[levels - 1] || n.s > 64 || n.s < 1 || !A->d_tri || !A->d_sp_ptr || !A->d_sp_ent || (r[8] & 3))) ||
            (quad && (sparse || !A->d_u4 || !A->d_d4 || (n.finv_off & 3) || (n.w_off & 3)))) {
            delete d;
            set_error("ls_direct_create: node %d of the plan is inconsistent", i);
            return LS_E_INVALID;
        }
        const int64_t s4 = (n.s + 3) & ~3, b4 = (n.b + 3) & ~3, tri_w = ((int64_t)n.s * (n.s + 1) / 2 + 3) & ~(int64_t)3;
        const int64_t up_w = sparse ? tri_w : quad ? s4 * n.b : (int64_t)n.s * n.b;
        const int64_t down_w = sparse ? tri_w : quad ? (s4 + b4) * n.s : (int64_t)n.s * n.s + (int64_t)n.s * n.b;
        fe += up_w + down_w; fe_up += up_w; fe_down += down_w;
        int lv = 0;
        while (lv + 1 < levels && i >= level_off[lv + 1]) ++lv;
        if (d->lvl_up.empty()) { d->lvl_up.assign((size_t)levels, 0); d->lvl_down.assign((size_t)levels, 0); d->lvl_rows.assign((size_t)levels, 0); d->lvl_bnd.assign((size_t)levels, 0); }
        d->lvl_up[(size_t)lv] += up_w; d->lvl_down[(size_t)lv] += down_w;
        d->lvl_rows[(size_t)lv] += n.s; d->lvl_bnd[(size_t)lv] += n.b;
    }
    fe += 2 * A->n_sp_ent + A->n_sp_ptr;        // each CSR list is read by one sweep (8-byte entries, 4-byte pointers)
    fe_up += A->n_sp_ent + A->n_sp_ptr / 2; fe_down += A->n_sp_ent + A->n_sp_ptr - A->n_sp_ptr / 2;
    if (!d->lvl_up.empty()) { d->lvl_up[(size_t)levels - 1] += A->n_sp_ent + A->n_sp_ptr / 2; d->lvl_down[(size_t)levels - 1] += A->n_sp_ent + A->n_sp_ptr - A->n_sp_ptr / 2; }
    d->factor_entries = fe; d->words_up = fe_up; d->words_down = fe_down;
    // Cache policy of the read-once factor streams (common.h, ld_stream; profiles/r05_nt_policy.txt: us per solve, default / nt):
    //   70k 70 / 78   250k 88 / 96   490k 139 / 133 (level kernels only; + tier: 140)   1M 214 / 199   2M 450 / 431   4M 722 / 705
    // While the factor (<= ~200 MB) stays resident in the 256 MB Infinity Cache from solve to solve, nt throws that away; beyond, the
    // level kernels' rows go nt first (their vectors and slot records are what the L2 should keep), the tier's dense streams from ~400 MB on.
    // LS_ND_NT=0 / 1 forces it off / on (A/B), ls_direct_set(h, "nt", v) per handle.
    {
        const double mb = 4e-6 * (double)fe;
        d->nt_rule[0] = mb > (double)env_int("LS_ND_NT_LEVELS_MB", 256);
        d->nt_rule[1] = mb > (double)env_int("LS_ND_NT_TIER_MB", 400);
        const int force = env_int0("LS_ND_NT", -1);
        d->nt_levels = force < 0 ? d->nt_rule[0] : force != 0;
        d->nt_tier = force < 0 ? d->nt_rule[1] : force != 0;
    }
    for (int lv = 0; lv < levels; ++lv)
        for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i) {
            const NodeDesc& n = nodes[i];
            const int64_t* r = h_nodes + (size_t)i * LS_DIRECT_NODE_COLS;
            NodeD& q = nd[i];
            q.s = n.s; q.b = n.b; q.own_start = n.own_start; q.bnd_off = n.bnd_off; q.front_off = n.front_off;
            q.pfront_off = i > 1 ? nodes[n.parent].front_off : -1;
            q.cix = lv ? (int)((i - level_off[lv]) % arity) : 0;
            q.flags = (lv + 1 >= levels ? NODE_LEAF : 0) | (r[8] >= 0 ? NODE_SPARSE : 0) | (r[11] != 0 ? NODE_QUAD : 0);
            q.finv_off = r[8] >= 0 ? r[8] : n.finv_off; q.w_off = n.w_off;
            q.spb_off = (int)r[9]; q.sps_off = (int)r[10];
        }
    // which children contribute to a front position: bit c of mask[f]
    std::vector<unsigned char> mask((size_t)n_front, 0);
    for (int lv = 1; lv < levels; ++lv)
        for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i) {
            const NodeDesc& n = nodes[i];
            const int cix = (int)((i - level_off[lv]) % arity);
            const NodeDesc& p = nodes[n.parent];
            for (int k = 0; k < n.b; ++k) {
                const int pp = h_ppos[n.bnd_off + k];
                if (pp < 0 || pp >= p.s + p.b) {
                    delete d;
                    set_error("ls_direct_create: ppos out of range at node %lld", (long long)i);
                    return LS_E_INVALID;
                }
                mask[(size_t)p.front_off + pp] |= (unsigned char)(1u << cix);
            }
        }
    // bottom tier: the levels whose nodes the caller stored in the tier kernels' layouts (quad-interleaved / sparse leaves)
    // run as one launch per sweep; every level above is one launch per sweep (unpadded finv / wf / wb)
    std::vector<TierItem> items;
    std::vector<TierWG> wgs;
    {
        int root = levels;
        for (int lv = levels - 1; lv >= 0; --lv) {
            bool tiered = false, plain = false;
            for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i) {
                if (nd[i].s == 0 && nd[i].b == 0) continue;
                if (nd[i].flags & (NODE_SPARSE | NODE_QUAD)) tiered = true; else plain = true;
            }
            if (tiered && plain) { delete d; set_error("ls_direct_create: level %d mixes tier-layout and plain nodes", lv); return LS_E_INVALID; }
            if (plain) break;
            if (tiered && root != lv + 1) { delete d; set_error("ls_direct_create: tier-layout levels must be the deepest ones"); return LS_E_INVALID; }
            if (tiered) root = lv;
        }
        for (int lv = 0; lv < root; ++lv)
            for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i)
                if (nd[i].flags & (NODE_SPARSE | NODE_QUAD)) { delete d; set_error("ls_direct_create: tier-layout node above the tier"); return LS_E_INVALID; }
        d->tier_root = levels;
        if (root < levels) {
            const int H = levels - root;
            if (cut > root) { delete d; set_error("ls_direct_create: %d ranks need a cut below the tier's root level (tree too small)", n_ranks); return LS_E_INVALID; }
            int64_t span = 1;
            for (int t = cut; t < root; ++t) span *= arity;
            // the kernel that will run is the one whose budget is checked: sixteen waves first where the rule asks for them (a subtree one
            // level taller may fit 160 KB as 16 regions and not 150 KB as 4), the 4- / 8-wave plans otherwise
            size_t region = 0;
            if (H <= TIER_MAX_H && direct_tier_full16(V, arity, levels, H, n_ranks, A->tier_waves)) {
                int vec16 = 0, tri16 = 0;
                const size_t region16 = plan_tier(nd, level_off, levels, arity, root, sub_lo * span, sub_hi * span, items, wgs, vec16, tri16, TIER_WAVES_FULL);
                if (region16 && ((region16 + 3) & ~(size_t)3) * sizeof(float) * TIER_WAVES_FULL <= 160 * 1024) {
                    d->tier_vec = vec16; d->tier_tri = tri16; region = region16; d->tier_waves = TIER_WAVES_FULL;
                }
            }
            if (!region) {
                region = H <= TIER_MAX_H ? plan_tier(nd, level_off, levels, arity, root, sub_lo * span, sub_hi * span, items, wgs, d->tier_vec, d->tier_tri) : 0;
                if (!region || region * sizeof(float) * TIER_WAVES > 150 * 1024) {
                    delete d;
                    set_error("ls_direct_create: a tier of %d levels does not fit the kernel (LDS per wave: %zu floats)", H, region);
                    return LS_E_WORKSPACE;
                }
                // few subtrees (<= 768 workgroups: three per CU or less): 8 waves per workgroup, two workgroups per CU, if that fits the LDS
                // (a tier of the leaf level alone -- dense leaves of 2-4 row chunks -- does not gain: 40k vertices 44.3 against 42.7 us)
                if (H >= 2 && (int)wgs.size() <= 768 && (A->tier_waves > 0 ? A->tier_waves : env_int0("LS_ND_TIER_WAVES", TIER_WAVES_WIDE)) == TIER_WAVES_WIDE) {
                    std::vector<TierItem> items8;
                    std::vector<TierWG> wgs8;
                    int vec8 = 0, tri8 = 0;
                    const size_t region8 = plan_tier(nd, level_off, levels, arity, root, sub_lo * span, sub_hi * span, items8, wgs8, vec8, tri8, TIER_WAVES_WIDE);
                    if (region8 && ((region8 + 3) & ~(size_t)3) * sizeof(float) * TIER_WAVES_WIDE <= 80 * 1024) {
                        items.swap(items8); wgs.swap(wgs8); d->tier_vec = vec8; d->tier_tri = tri8; region = region8; d->tier_waves = TIER_WAVES_WIDE;
                    }
                }
            }
            d->tier_root = root; d->tier_phases = H; d->tier_wgs = (int)wgs.size(); d->tier_region = (int)((region + 3) & ~(size_t)3);
        }
    }
    d->fuse_root = getenv("LS_ND_NO_FUSE_ROOT") == nullptr;
    d->tier_xcd = env_int0("LS_ND_XCD_TIER", env_int0("LS_ND_XCD", 1)) != 0;
    d->upper_lo = (d->tier_root < levels && d->tier_root > 0) ? nodes[level_off[d->tier_root - 1]].own_start : (int)V;
    std::vector<int> pull;
    if (d->tier_root < levels) {
        // only the tier's inner nodes are parents inside the tier: their front positions are one range (fronts are numbered level by level)
        d->pull_base = nodes[level_off[d->tier_root]].front_off;
        const int64_t pull_end = nodes[level_off[levels - 1]].front_off;
        pull.assign((size_t)std::max<int64_t>(0, pull_end - d->pull_base) * arity, -1);
        for (int lv = d->tier_root + 1; lv < levels; ++lv)
            for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i) {
                const NodeDesc& n = nodes[i];
                const int cix = (int)((i - level_off[lv]) % arity);
                const int64_t pf = nodes[n.parent].front_off;
                for (int k = 0; k < n.b; ++k) pull[(size_t)(pf - d->pull_base + h_ppos[n.bnd_off + k]) * arity + cix] = n.bnd_off + k;
            }
    }
    // tiles of the upper levels: a range of rows of one node each
    std::vector<Tile> tiles;
    std::vector<PackedTile> ptiles;
    auto tile_of = [&](int64_t i, int r, int lv, int forward) {
        const NodeDesc& n = nodes[i];
        Tile t;
        t.store = 0; t.row0 = r; t.s = n.s; t.b = n.b; t.own_start = n.own_start; t.bnd_off = n.bnd_off; t.front_off = n.front_off;
        t.leaf = lv + 1 >= levels;
        t.pfront_off = i > 1 ? nodes[n.parent].front_off : -1;
        t.cix = lv ? (int)((i - level_off[lv]) % arity) : 0;
        t.forward = forward; t.arity = arity; t.finv_off = n.finv_off; t.w_off = n.w_off;
        return t;
    };
    d->plan.resize(levels);
    size_t lds_max = 0;
    // (measured over 576 .. 1M vertices, tools/tier_sweep.py: the down sweep of levels with s + b of 90 .. 250 runs 5-10 % of a
    //  whole solve faster with the lanes along the reduction; the up sweep of the s = 125 nodes of an 8-level tree loses 2-3 %)
    const int long_red = env_int("LS_ND_LONG", 64);             // down sweep (reduction s + b)
    const int long_up = env_int("LS_ND_LONG_UP", levels >= 8 ? 256 : 64);   // up sweep (reduction s): small trees gain 1-3 % from 64
    for (int lv = 0; lv < d->tier_root; ++lv) {
        LevelPlan& p = d->plan[lv];
        // this rank's nodes of the level: one contiguous range (all of them above the cut)
        int64_t a0 = level_off[lv], a1 = level_off[lv + 1];
        if (lv >= cut) { int64_t span = 1; for (int t = cut; t < lv; ++t) span *= arity; a0 = level_off[lv] + sub_lo * span; a1 = level_off[lv] + sub_hi * span; }
        int red_up = 0, red_down = 0;
        for (int64_t i = a0; i < a1; ++i) {
            p.s_cap = std::max(p.s_cap, nodes[i].s); p.b_cap = std::max(p.b_cap, nodes[i].b);
            red_up = std::max(red_up, nodes[i].s); red_down = std::max(red_down, nodes[i].s + nodes[i].b);
        }
        // long reductions: lanes along the reduction (k_nd_*_b), ND_ROWS * ND_BW rows per tile; short: a row per lane
        p.up_b = red_up >= long_up; p.down_b = red_down >= long_red;
        // *_b kernels, shape per level (measured at 1M vertices, profiles/r03_level_kernel_variants.txt):
        //  * every tile assembles its node's whole reduction vector -- long vectors (the top levels: 2000 entries, four 16-byte slot
        //    records each) want FEW, FAT workgroups: 8 waves from ~1400 entries on (root 15.0 -> 13.9 us, level 1 15.2 -> 14.3 / 17.3 -> 16.3);
        //  * below that 4 waves, and as many row chunks per wave (<= 4) as still leave ~500 tiles in the level (two per CU; 1M: level 4 down
        //    19.6 -> 17.2 us with 4 chunks, level 3 down 18.9 -> 15.7 with 2).
        // LS_ND_INFLIGHT (16-byte requests per lane) selects the chunks by requests in flight instead (the round-2 rule).
        const int bw_long = env_int("LS_ND_BW_LONG", 1400), tile_target = env_int("LS_ND_TILES", 500), inflight = env_int("LS_ND_INFLIGHT", 0);
        const int up_bw = std::min(ND_BW, red_up >= bw_long ? 8 : 4), down_bw = std::min(ND_BW, red_down >= bw_long ? 8 : 4);
        p.up_nw = p.up_b ? up_bw : pick_nw(red_up, true); p.down_nw = p.down_b ? down_bw : pick_nw(red_down);
        const int lpr_up = div_up(std::max(p.s_cap, 1), 4 * WAVE), lpr_down = lpr_up + div_up(std::max(p.b_cap, 1), 4 * WAVE);   // 16-byte requests per row and lane
        int64_t rows_up = 0, rows_down = 0;
        for (int64_t i = a0; i < a1; ++i) { rows_up += nodes[i].b; rows_down += nodes[i].s; }
        auto pick_chunks = [&](int64_t rows, int bw, int lpr) {
            if (inflight > 0) return std::min(4, std::max(1, div_up(inflight, ND_ROWS * lpr)));
            int c = 4;
            while (c > 1 && rows / (ND_ROWS * bw * c) < tile_target) --c;
            return c;
        };
        p.up_chunks = pick_chunks(rows_up, up_bw, lpr_up);
        p.down_chunks = pick_chunks(rows_down, down_bw, lpr_down);
        const int up_rows = p.up_b ? ND_ROWS * up_bw * p.up_chunks : WAVE, down_rows = p.down_b ? ND_ROWS * down_bw * p.down_chunks : WAVE;
        // small nodes: whole matrix staged in LDS, one workgroup per node, a row per thread
        const int small_lds = env_int("LS_ND_SMALL_KB", 40) * 1024;
        const size_t up_s_bytes = ((size_t)p.s_cap * p.b_cap + (size_t)p.s_cap * d->kmax) * sizeof(float);
        const size_t down_s_bytes = ((size_t)p.s_cap * (p.s_cap + p.b_cap) + (size_t)(p.s_cap + p.b_cap) * d->kmax) * sizeof(float);
        p.up_s = p.b_cap <= 256 && p.s_cap <= 256 && up_s_bytes <= (size_t)small_lds && !getenv("LS_ND_NO_SMALL");
        // (measured at 1M: staging pays for the up sweep -- W only, 42 + 18 + 15 us against 45 + 21 + 17 -- but not for the
        //  down sweep, whose Finv + W footprint leaves 6 single-wave workgroups per CU: 151 us against 57; LS_ND_SMALL_DOWN=1 enables it)
        p.down_s = p.s_cap <= 256 && down_s_bytes <= (size_t)small_lds && getenv("LS_ND_SMALL_DOWN") && !getenv("LS_ND_NO_SMALL");
        if (p.up_s) { p.up_b = 0; p.up_nw = std::max(1, div_up(p.b_cap, WAVE)); }
        if (p.down_s) { p.down_b = 0; p.down_nw = std::max(1, div_up(p.s_cap, WAVE)); }
        // tiny nodes: several nodes per wave (packed tiles), when at least two nodes of the level fit a wave
        const bool no_pack = getenv("LS_ND_NO_PACK") != nullptr;
        // largest row count of a level that is still packed (measured at 1M: packing pairs of ~31-row leaves helps the up
        // sweep, 42 -> 37 us, while packed down tiles only pay below half a wave)
        const int pack_up = env_int("LS_ND_PACK_ROWS_UP", WAVE), pack_down = env_int("LS_ND_PACK_ROWS", WAVE / 2);
        p.up_p = !no_pack && p.b_cap <= pack_up && p.s_cap <= 256 && lv > 0;
        p.down_p = !no_pack && p.s_cap <= pack_down;
        auto pack_level = [&](bool up_sweep, int& first, int& count, int& lds_rows_s, int& lds_rows_b) {
            first = (int)ptiles.size();
            lds_rows_s = lds_rows_b = 0;
            int64_t i = a0;
            while (i < a1) {
                PackedTile t;
                memset(&t, 0, sizeof(t));
                t.leaf = lv + 1 >= levels; t.arity = arity;
                int rows = 0;
                while (i < a1 && t.n < ND_PACK) {
                    const NodeDesc& nd = nodes[i];
                    const int r = up_sweep ? nd.b : nd.s;
                    if (t.n && rows + r > WAVE) break;
                    NodeP& q = t.d[t.n];
                    q.s = nd.s; q.b = nd.b; q.own_start = nd.own_start; q.bnd_off = nd.bnd_off; q.front_off = nd.front_off;
                    q.pfront_off = i > 1 ? nodes[nd.parent].front_off : -1;
                    q.cix = lv ? (int)((i - level_off[lv]) % arity) : 0; q.pad = 0; q.finv_off = nd.finv_off; q.w_off = nd.w_off;
                    t.row0[t.n + 1] = t.row0[t.n] + r;
                    t.sb0[t.n + 1] = t.sb0[t.n] + nd.s;
                    t.sx0[t.n + 1] = t.sx0[t.n] + nd.b;
                    rows += r; ++t.n; ++i;
                }
                for (int c = t.n + 1; c <= ND_PACK; ++c) { t.row0[c] = t.row0[t.n]; t.sb0[c] = t.sb0[t.n]; t.sx0[c] = t.sx0[t.n]; }
                lds_rows_s = std::max(lds_rows_s, t.sb0[t.n]); lds_rows_b = std::max(lds_rows_b, t.sx0[t.n]);
                ptiles.push_back(t);
            }
            count = (int)ptiles.size() - first;
        };
        int dummy = 0;
        if (p.up_p) { pack_level(true, p.up_p_first, p.up_p_tiles, p.up_p_lds, dummy); p.up_s = 0; p.up_b = 0; }
        if (p.down_p) { pack_level(false, p.down_p_first, p.down_p_tiles, p.down_p_s, p.down_p_lds); p.down_s = 0; p.down_b = 0; }
        p.up_first = (int)tiles.size();
        const int up_threads = WAVE * p.up_nw;
        std::vector<int> up_nodes;                                 // node (index within the level) of every tile pushed below
        for (int64_t i = a0; i < a1; ++i) {
            // b' of the own rows is kept for the down sweep: by the first compute tile (small nodes, or nodes whose only
            // tile exists for that purpose), or by store-only tiles of blockDim rows each (large nodes: the one tile
            // would walk s / blockDim dependent load chains)
            const bool store_tiles = !p.up_s && nodes[i].s > 2 * up_threads;
            const int rows = std::max(nodes[i].b, (nodes[i].s && !store_tiles) ? 1 : 0);
            for (int r = 0; r < rows; r += (p.up_s ? 1 << 30 : up_rows)) {
                tiles.push_back(tile_of(i, r, lv, 0));
                tiles.back().store = (r == 0 && !store_tiles) ? 1 : 0;
                up_nodes.push_back((int)(i - a0));
            }
            if (store_tiles)
                for (int r = 0; r < nodes[i].s; r += up_threads) { tiles.push_back(tile_of(i, r, lv, 0)); tiles.back().store = 2; up_nodes.push_back((int)(i - a0)); }
        }
        // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement, speed only) and every tile of a node re-assembles
        // the node's reduction vector (slots, masks, b) -- with a node's tiles on ONE XCD those reads hit that XCD's L2 instead of
        // being fetched once per XCD (PMC, round 3: 55 MB per launch of the row-per-lane up kernel for 29 MB of factor). Levels
        // with fewer than 8 nodes keep the plain order (a node must not be confined to the 32 CUs of one XCD).
        auto xcd_order = [&](int first, std::vector<int>& node_of_tile, bool up_sweep) {
            const int n = (int)tiles.size() - first;
            if (a1 - a0 < 8 || n < 16 || !env_int0("LS_ND_XCD", 1)) return;
            std::vector<std::vector<Tile>> bucket(8);
            for (int t = 0; t < n; ++t) bucket[(size_t)(node_of_tile[(size_t)t] & 7)].push_back(tiles[(size_t)(first + t)]);
            size_t longest = 0;
            for (auto& bk : bucket) longest = std::max(longest, bk.size());
            Tile idle;
            memset(&idle, 0, sizeof(idle));
            idle.store = 3; idle.forward = 2;                     // a tile that returns at once (pads the shorter buckets)
            tiles.resize((size_t)first);
            for (size_t r = 0; r < longest; ++r)
                for (int x = 0; x < 8; ++x) tiles.push_back(r < bucket[(size_t)x].size() ? bucket[(size_t)x][r] : idle);
            (void)up_sweep;
        };
        xcd_order(p.up_first, up_nodes, true);
        p.up_tiles = (int)tiles.size() - p.up_first;
        p.down_first = (int)tiles.size();
        std::vector<int> down_nodes;
        for (int64_t i = a0; i < a1; ++i)
            for (int r = 0; r < nodes[i].s; r += (p.down_s ? 1 << 30 : down_rows)) { tiles.push_back(tile_of(i, r, lv, 0)); down_nodes.push_back((int)(i - a0)); }
        if (lv + 1 < levels)      // forward tiles: boundary rows of every node (staged kernel: only of nodes without own rows)
            for (int64_t i = a0; i < a1; ++i)
                if (!p.down_s || !nodes[i].s)
                    for (int r = 0; r < nodes[i].b; r += WAVE * p.down_nw) { tiles.push_back(tile_of(i, r, lv, 1)); down_nodes.push_back((int)(i - a0)); }
        xcd_order(p.down_first, down_nodes, false);
        p.down_tiles = (int)tiles.size() - p.down_first;
        lds_max = std::max(lds_max, ((size_t)p.s_cap + p.b_cap + 16 * WAVE) * d->kmax * sizeof(float));
    }
    // ---- accounting (host only): the static index bytes of every level's launches, and how evenly the tier's subtrees are loaded -------
    {
        d->lvl_idx_up.assign((size_t)levels, 0); d->lvl_idx_down.assign((size_t)levels, 0);
        std::vector<int64_t> fpos((size_t)levels, 0), bsum((size_t)levels + 1, 0);
        for (int lv = 0; lv < levels; ++lv)
            for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i)
                if (active(i, lv)) { fpos[(size_t)lv] += nodes[i].s + nodes[i].b; bsum[(size_t)lv] += nodes[i].b; }
        for (int lv = 0; lv < d->tier_root; ++lv) {
            const LevelPlan& p = d->plan[lv];
            // up: tile records, the children mask of every front position, the parent positions of the boundary rows (+ perm when no
            // tier launch gathered b into the tree's numbering); down: tile records, row pointers of the push lists, the children's targets
            d->lvl_idx_up[(size_t)lv] = (p.up_p ? (int64_t)p.up_p_tiles * (int64_t)sizeof(PackedTile) : (int64_t)p.up_tiles * (int64_t)sizeof(Tile)) + fpos[(size_t)lv] + 4 * bsum[(size_t)lv];
            d->lvl_idx_down[(size_t)lv] = (p.down_p ? (int64_t)p.down_p_tiles * (int64_t)sizeof(PackedTile) : (int64_t)p.down_tiles * (int64_t)sizeof(Tile)) + 4 * fpos[(size_t)lv] +
                                          4 * bsum[(size_t)lv + 1];
        }
        if (d->tier_root < levels) {
            const int H = levels - d->tier_root;
            // item records by the level they belong to: phase ph of the up sweep is level levels - 1 - ph, of the down sweep tier_root + ph
            for (const TierWG& g : wgs)
                for (int ph = 0; ph < H; ++ph) {
                    d->lvl_idx_up[(size_t)(levels - 1 - ph)] += (int64_t)(g.up_off[ph + 1] - g.up_off[ph]) * (int64_t)sizeof(TierItem);
                    d->lvl_idx_down[(size_t)(d->tier_root + ph)] += (int64_t)(g.down_off[ph + 1] - g.down_off[ph]) * (int64_t)sizeof(TierItem);
                }
            d->lvl_idx_up[(size_t)d->tier_root] += (int64_t)wgs.size() * (int64_t)sizeof(TierWG);
            d->lvl_idx_down[(size_t)d->tier_root] += (int64_t)wgs.size() * (int64_t)sizeof(TierWG);
            for (int lv = d->tier_root; lv < levels; ++lv) {
                const bool inner = lv + 1 < levels;
                // up: an inner node pulls its children's updates through `arity` indices per front position; the tier's root level and the
                // leaves read the parent positions of their boundary rows. down: push-list pointers per front position + the children's targets
                if (inner) d->lvl_idx_up[(size_t)lv] += 4 * (int64_t)arity * fpos[(size_t)lv];
                if (lv == d->tier_root || !inner) d->lvl_idx_up[(size_t)lv] += 4 * bsum[(size_t)lv];
                if (inner) d->lvl_idx_down[(size_t)lv] += 4 * fpos[(size_t)lv] + 4 * bsum[(size_t)lv + 1];
            }
            // factor words per subtree = per workgroup: with one workgroup per CU the slowest subtree is the launch's time
            int64_t span0 = 1;
            for (int t = cut; t < d->tier_root; ++t) span0 *= arity;
            const int64_t q0 = sub_lo * span0;
            double mx[2] = {0, 0}, sum[2] = {0, 0};
            for (size_t w = 0; w < wgs.size(); ++w) {
                double wu = 0, wd = 0;
                int64_t span = 1;
                for (int lv = d->tier_root; lv < levels; ++lv) {
                    const int64_t first = level_off[lv] + (q0 + (int64_t)w) * span;
                    for (int64_t i = first; i < first + span; ++i) {
                        const NodeD& n = nd[i];
                        const double s4 = (n.s + 3) & ~3, b4 = (n.b + 3) & ~3, tri_w = ((int64_t)n.s * (n.s + 1) / 2 + 3) & ~(int64_t)3;
                        if (n.flags & NODE_SPARSE) { wu += tri_w + 3.0 * n.b; wd += tri_w + 3.0 * n.b; }      // (+ ~3 words per boundary row of sparse block and pointers)
                        else { wu += s4 * n.b; wd += (s4 + b4) * n.s; }
                    }
                    span *= arity;
                }
                mx[0] = std::max(mx[0], wu); mx[1] = std::max(mx[1], wd); sum[0] += wu; sum[1] += wd;
            }
            const double nw = (double)std::max<size_t>(wgs.size(), 1);
            d->tier_balance[0] = mx[0]; d->tier_balance[1] = sum[0] / nw; d->tier_balance[2] = mx[1]; d->tier_balance[3] = sum[1] / nw;
        }
    }
    if (lds_max > 150 * 1024) {
        delete d;
        set_error("ls_direct_create: a front needs %zu bytes of LDS (separator too large for this kernel)", lds_max);
        return LS_E_INVALID;
    }
    if (n_ranks > 1 && cut >= 1) {
        d->exch_f0 = nodes[level_off[cut - 1]].front_off;
        d->exch_f1 = nodes[level_off[cut]].front_off;
    }
    if (n_ranks > 1) {        // designated owner of every row of x: the rank that runs the row's subtree, rank 0 for the replicated levels
        d->owned_rows.assign((size_t)V, 0);
        for (int lv = 0; lv < levels; ++lv)
            for (int64_t i = level_off[lv]; i < level_off[lv + 1]; ++i)
                if (lv < cut ? rank == 0 : active(i, lv))
                    for (int r = 0; r < nodes[i].s; ++r) d->owned_rows[(size_t)h_perm[nodes[i].own_start + r]] = 1;
    }
    int rc = LS_OK;
    lap("tables built (host)");
    // The tables go up on a stream of their own: `st` is busy with the factorisation when ls_direct_factor calls this (a copy from
    // pageable memory queued behind it kept the HOST waiting until the last kernel was done, and ~50 MB of index lists then crossed
    // the bus one after the other while the device idled: 3.5 of the 28 ms of a 1M-vertex construction). `st` waits for them through an event.
    hipStream_t su = side_stream(device, 0);
    LS_REQUIRE(su, LS_E_STATE, "ls_direct_create: no side stream on this device");
    auto table = [&](void** dst, size_t bytes) -> hipError_t {          // from the pool when a destroyed handle left one of this size there
        size_t cap = bytes;
        *dst = pool_take(device, bytes, &cap);
        if (!*dst) { const hipError_t e = pool_alloc(device, dst, bytes); if (e != hipSuccess) return e; }
        d->tables.emplace_back(*dst, cap);
        return hipSuccess;
    };
    auto up = [&](auto** dst, const auto* src, size_t n) -> int {
        LS_HIP(table((void**)dst, std::max<size_t>(n, 1) * sizeof(**dst)));
        if (n) LS_HIP(hipMemcpyAsync(*dst, src, n * sizeof(**dst), hipMemcpyHostToDevice, su));
        return LS_OK;
    };
    if (!(rc = up(&d->tiles, tiles.data(), tiles.size())) && !(rc = up(&d->ptiles, ptiles.data(), ptiles.size())) &&
        !(rc = up(&d->perm, h_perm, (size_t)V)) &&
        !(rc = up(&d->ppos, h_ppos, (size_t)n_bnd)) && !(rc = up(&d->push_ptr, h_push_ptr, (size_t)n_front + 1)) &&
        !(rc = up(&d->push_tgt, h_push_tgt, (size_t)n_bnd)) && !(rc = up(&d->mask, mask.data(), mask.size())) &&
        !(rc = up(&d->d_items, items.data(), items.size())) && !(rc = up(&d->pull, pull.data(), pull.size())) &&
        !(rc = up(&d->d_wgs, wgs.data(), wgs.size()))) {
        hipError_t e = table((void**)&d->bp, sizeof(float) * (size_t)V * d->kmax);
        if (e == hipSuccess) e = table((void**)&d->braw, sizeof(float) * (size_t)V * d->kmax);
        if (e == hipSuccess) e = table((void**)&d->slots, sizeof(float) * (size_t)n_front * arity * d->kmax);
        if (e == hipSuccess) e = table((void**)&d->xb, sizeof(float) * (size_t)std::max<int64_t>(n_bnd, 1) * d->kmax);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&d->busy, hipEventDisableTiming);
        lap("uploads enqueued, vectors allocated");
        if (e == hipSuccess) e = hipEventRecord(d->busy, su);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, d->busy, 0);      // whatever follows on the caller's stream sees the tables
        if (e == hipSuccess) e = hipStreamSynchronize(su);               // the host vectors above go out of scope
        lap("uploads done");
        if (e != hipSuccess) rc = hip_fail(e, "ls_direct_create allocations", __FILE__, __LINE__);
    }
    if (rc != LS_OK) { ls_direct_destroy(d); return rc; }
    // kernels of the top levels may need more than 64 KiB of dynamic LDS
#define LS_OPTIN(KK)                                                                                                   \
    (void)hipFuncSetAttribute((const void*)k_nd_up<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    (void)hipFuncSetAttribute((const void*)k_nd_down<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
    (void)hipFuncSetAttribute((const void*)k_nd_up_b<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
    (void)hipFuncSetAttribute((const void*)k_nd_down_b<KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    (void)hipFuncSetAttribute((const void*)k_nd_up<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
    (void)hipFuncSetAttribute((const void*)k_nd_down<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
    (void)hipFuncSetAttribute((const void*)k_nd_up_b<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
    (void)hipFuncSetAttribute((const void*)k_nd_down_b<KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
    (void)hipFuncSetAttribute((const void*)k_nd_up_s<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
    (void)hipFuncSetAttribute((const void*)k_nd_down_s<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LS_OPTIN(1) LS_OPTIN(2) LS_OPTIN(3) LS_OPTIN(4)
#undef LS_OPTIN
#define LS_OPTIN(KK)                                                                                                       \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, true, TIER_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, false, TIER_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, true, TIER_WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, false, TIER_WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, true, TIER_WAVES_WIDE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, false, TIER_WAVES_WIDE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, true, TIER_WAVES_FULL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, false, TIER_WAVES_FULL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, true, TIER_WAVES_FULL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
    (void)hipFuncSetAttribute((const void*)k_nd_tier<KK, false, TIER_WAVES_FULL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LS_OPTIN(1) LS_OPTIN(2) LS_OPTIN(3) LS_OPTIN(4)
#undef LS_OPTIN
    lap("kernel attributes set");
    *out = d;
    return LS_OK;
}

extern "C" int ls_direct_destroy(ls_direct* d) {
    if (!d) return LS_OK;
    DeviceGuard g(d->device);
    if (d->busy) (void)hipEventDestroy(d->busy);
    (void)hipDeviceSynchronize();                                   // (what hipFree did implicitly: nothing of the handle is in flight any more)
    for (const auto& t : d->tables)                                 // tiles, perm, ppos, push lists, mask, items, pull, wgs; bp, braw, slots, xb
        if (!ls::pool_give(d->device, t.first, t.second)) (void)hipFree(t.first);
    for (size_t i = 0; i < d->owned.size(); ++i)
        if (!ls::pool_give(d->device, d->owned[i], i < d->owned_bytes.size() ? d->owned_bytes[i] : 0)) (void)hipFree(d->owned[i]);
    for (hipEvent_t e : d->ev) (void)hipEventDestroy(e);
#ifdef LS_TIER_STAMPS
    if (d->stamps) (void)hipFree(d->stamps);
#endif
    for (hipEvent_t e : d->lev) (void)hipEventDestroy(e);
    delete d;
    return LS_OK;
}

// part: -1 the whole solve; sharded handles: 0 = this rank's subtrees upwards, level cut - 1's slots -> exchange buffer;
//       1 = exchange buffer (summed over the ranks by the caller) -> slots, the replicated levels, everything downwards
template <int K>
static int direct_solve_k(ls_direct* d, const float* b, float* x, hipStream_t st, int part, float* exchange) {
    const int top = d->tier_root - 1;            // levels [0, tier_root) are one launch each, the rest is the bottom tier
    TierArgs ta;
    ta.items = d->d_items; ta.wgs = d->d_wgs; ta.perm = d->perm; ta.mask = d->mask; ta.ppos = d->ppos;
    ta.pull = d->pull ? d->pull - (size_t)d->pull_base * d->arity : nullptr; ta.push_ptr = d->push_ptr; ta.push_tgt = d->push_tgt; ta.u4 = d->u4; ta.d4 = d->d4; ta.tri = d->tri;
    ta.sp_ptr = d->sp_ptr; ta.sp_ent = d->sp_ent; ta.bprime = d->bp; ta.braw = d->braw; ta.slots = d->slots; ta.xb = d->xb;
    ta.arity = d->arity; ta.phases = d->tier_phases; ta.region_floats = d->tier_region; ta.vec_floats = d->tier_vec;
    ta.upper_lo = d->upper_lo; ta.upper_hi = (int)d->V;
    ta.xcd_order = d->tier_xcd;
#ifdef LS_TIER_STAMPS
    const size_t stamp_n = (size_t)d->tier_wgs * d->tier_waves * TIER_STAMP_SLOTS;
    if (d->profile == 2 && !d->stamps && stamp_n) LS_HIP(hipMalloc((void**)&d->stamps, 2 * stamp_n * sizeof(long long)));
    if (d->profile == 2 && d->stamps) LS_HIP(hipMemsetAsync(d->stamps, 0, 2 * stamp_n * sizeof(long long), st));
    ta.stamps = d->profile == 2 ? d->stamps : nullptr;
#endif
    const size_t tier_lds = (size_t)d->tier_region * d->tier_waves * sizeof(float);
    const bool nt_tier = d->nt_tier && d->tier_waves != TIER_WAVES_WIDE, nt = d->nt_levels;
    int n_mark = 0;
    auto mark = [&](int lo, int hi, int sweep) -> hipError_t {        // "profile" = 3: an event in front of every launch
        if (d->profile != 3) return hipSuccess;
        while ((int)d->lev.size() <= n_mark + 1) { hipEvent_t e; const hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; d->lev.push_back(e); }
        if ((int)d->lmeta.size() <= n_mark) d->lmeta.resize((size_t)n_mark + 1);
        d->lmeta[(size_t)n_mark] = {lo, hi, sweep};
        return hipEventRecord(d->lev[(size_t)n_mark++], st);
    };
    auto mark_end = [&]() -> hipError_t {
        if (d->profile != 3) return hipSuccess;
        hipError_t r = hipEventRecord(d->lev[(size_t)n_mark], st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        d->launch_ms.assign((size_t)n_mark, 0.0);
        d->lmeta.resize((size_t)n_mark);
        for (int i = 0; i < n_mark && r == hipSuccess; ++i) { float ms = 0; r = hipEventElapsedTime(&ms, d->lev[(size_t)i], d->lev[(size_t)i + 1]); d->launch_ms[(size_t)i] = ms; }
        return r;
    };
    const size_t exch_off = (size_t)d->exch_f0 * d->arity * K, exch_n = (size_t)(d->exch_f1 - d->exch_f0) * d->arity * K;
    if (part != 1) {
    if (d->profile) LS_HIP(hipEventRecord(d->ev[0], st));
    if (part == 0 && exch_n) LS_HIP(hipMemsetAsync(d->slots + exch_off, 0, exch_n * sizeof(float), st));
    if (d->tier_wgs) {
        LS_HIP(mark(d->tier_root, d->levels - 1, 0));
        if (d->tier_waves == TIER_WAVES_FULL && nt_tier) hipLaunchKernelGGL((k_nd_tier<K, true, TIER_WAVES_FULL, true>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_FULL), tier_lds, st, ta, b, x, d->tier_tri);
        else if (d->tier_waves == TIER_WAVES_FULL) hipLaunchKernelGGL((k_nd_tier<K, true, TIER_WAVES_FULL, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_FULL), tier_lds, st, ta, b, x, d->tier_tri);
        else if (d->tier_waves == TIER_WAVES_WIDE) hipLaunchKernelGGL((k_nd_tier<K, true, TIER_WAVES_WIDE, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_WIDE), tier_lds, st, ta, b, x, d->tier_tri);
        else if (nt_tier) hipLaunchKernelGGL((k_nd_tier<K, true, TIER_WAVES, true>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES), tier_lds, st, ta, b, x, d->tier_tri);
        else hipLaunchKernelGGL((k_nd_tier<K, true, TIER_WAVES, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES), tier_lds, st, ta, b, x, d->tier_tri);
    }
    }
    if (part == 1 && exch_n && exchange != d->slots + exch_off)
        LS_HIP(hipMemcpyAsync(d->slots + exch_off, exchange, exch_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    // the tier's up-sweep launch gathers b of the upper levels' rows into the tree's numbering (braw): their kernels skip perm -> b
    const int* up_perm = d->tier_wgs ? nullptr : d->perm;
    const float* up_b = d->tier_wgs ? d->braw : b;
    // the root (level 0, when it is a level of its own launches): b' is formed by its down-sweep tiles, no up-sweep launch
    const bool fuse_root = top >= 0 && d->fuse_root && !d->plan[0].down_p && !d->plan[0].down_s;
    const RootFill rf = fuse_root ? RootFill{up_perm, d->mask, d->slots, up_b} : RootFill{nullptr, nullptr, nullptr, nullptr};
    for (int lv = (part == 1 ? std::min(top, d->cut - 1) : top); lv >= (part == 0 ? d->cut : 0); --lv) {
        const LevelPlan& p = d->plan[lv];
        if (!(p.up_p ? p.up_p_tiles : p.up_tiles)) continue;
        if (lv == 0 && fuse_root) continue;
        LS_HIP(mark(lv, lv, 0));
        if (p.up_p)
            hipLaunchKernelGGL(k_nd_up_p<K>, dim3(p.up_p_tiles), dim3(WAVE), (size_t)std::max(p.up_p_lds, 1) * K * sizeof(float), st,
                               d->ptiles + p.up_p_first, up_perm, d->mask, d->ppos, d->wf, up_b, d->bp, d->slots);
        else if (p.up_s)
            hipLaunchKernelGGL(k_nd_up_s<K>, dim3(p.up_tiles), dim3(WAVE * p.up_nw), ((size_t)p.s_cap * p.b_cap + (size_t)p.s_cap * K) * sizeof(float), st,
                               d->tiles + p.up_first, up_perm, d->mask, d->ppos, d->wf, up_b, d->bp, d->slots, p.s_cap, p.b_cap);
        else if (p.up_b)
            hipLaunchKernelGGL((nt ? k_nd_up_b<K, true> : k_nd_up_b<K, false>), dim3(p.up_tiles), dim3(WAVE * p.up_nw), ((size_t)p.s_cap * K + 16) * sizeof(float), st,
                               d->tiles + p.up_first, up_perm, d->mask, d->ppos, d->wb, up_b, d->bp, d->slots, p.s_cap, p.up_chunks);
        else
            hipLaunchKernelGGL((nt ? k_nd_up<K, true> : k_nd_up<K, false>), dim3(p.up_tiles), dim3(WAVE * p.up_nw),
                               ((size_t)p.s_cap + (size_t)(p.up_nw - 1) * WAVE) * K * sizeof(float),
                               st, d->tiles + p.up_first, up_perm, d->mask, d->ppos, d->wf, up_b, d->bp, d->slots, p.s_cap);
    }
    if (part == 0) {
        if (exch_n && exchange != d->slots + exch_off)
            LS_HIP(hipMemcpyAsync(exchange, d->slots + exch_off, exch_n * sizeof(float), hipMemcpyDeviceToDevice, st));
        LS_HIP(hipGetLastError());
        return LS_OK;
    }
    if (d->profile) LS_HIP(hipEventRecord(d->ev[1], st));
    for (int lv = 0; lv <= top; ++lv) {
        const LevelPlan& p = d->plan[lv];
        if (!(p.down_p ? p.down_p_tiles : p.down_tiles)) continue;
        LS_HIP(mark(lv, lv, 1));
        if (p.down_p)
            hipLaunchKernelGGL(k_nd_down_p<K>, dim3(p.down_p_tiles), dim3(WAVE), (size_t)std::max(p.down_p_s + p.down_p_lds, 1) * K * sizeof(float), st,
                               d->ptiles + p.down_p_first, d->perm, d->push_ptr, d->push_tgt, d->finv, d->wb, (const float*)d->bp, d->xb, x,
                               p.down_p_s);
        else if (p.down_s)
            hipLaunchKernelGGL(k_nd_down_s<K>, dim3(p.down_tiles), dim3(WAVE * p.down_nw),
                               ((size_t)p.s_cap * (p.s_cap + p.b_cap) + (size_t)(p.s_cap + p.b_cap) * K) * sizeof(float), st, d->tiles + p.down_first,
                               d->perm, d->push_ptr, d->push_tgt, d->finv, d->wb, (const float*)d->bp, d->xb, x, p.s_cap, p.b_cap);
        else if (p.down_b)
            hipLaunchKernelGGL((nt ? k_nd_down_b<K, true> : k_nd_down_b<K, false>), dim3(p.down_tiles), dim3(WAVE * p.down_nw), (((size_t)p.s_cap + p.b_cap) * K + 32) * sizeof(float), st,
                               d->tiles + p.down_first, d->perm, d->push_ptr, d->push_tgt, d->finv, d->wf, (const float*)d->bp, d->xb, x,
                               p.s_cap, p.b_cap, p.down_chunks, lv == 0 ? rf : RootFill{nullptr, nullptr, nullptr, nullptr});
        else
            hipLaunchKernelGGL((nt ? k_nd_down<K, true> : k_nd_down<K, false>), dim3(p.down_tiles), dim3(WAVE * p.down_nw),
                               ((size_t)p.s_cap + p.b_cap + (size_t)(p.down_nw - 1) * WAVE) * K * sizeof(float), st, d->tiles + p.down_first,
                               d->perm, d->push_ptr, d->push_tgt, d->finv, d->wb, (const float*)d->bp, d->xb, x, p.s_cap, p.b_cap,
                               lv == 0 ? rf : RootFill{nullptr, nullptr, nullptr, nullptr});
    }
#ifdef LS_TIER_STAMPS
    if (ta.stamps) ta.stamps += stamp_n;
#endif
    if (d->tier_wgs) {
        LS_HIP(mark(d->tier_root, d->levels - 1, 1));
        if (d->tier_waves == TIER_WAVES_FULL && nt_tier) hipLaunchKernelGGL((k_nd_tier<K, false, TIER_WAVES_FULL, true>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_FULL), tier_lds, st, ta, b, x, d->tier_tri);
        else if (d->tier_waves == TIER_WAVES_FULL) hipLaunchKernelGGL((k_nd_tier<K, false, TIER_WAVES_FULL, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_FULL), tier_lds, st, ta, b, x, d->tier_tri);
        else if (d->tier_waves == TIER_WAVES_WIDE) hipLaunchKernelGGL((k_nd_tier<K, false, TIER_WAVES_WIDE, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES_WIDE), tier_lds, st, ta, b, x, d->tier_tri);
        else if (nt_tier) hipLaunchKernelGGL((k_nd_tier<K, false, TIER_WAVES, true>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES), tier_lds, st, ta, b, x, d->tier_tri);
        else hipLaunchKernelGGL((k_nd_tier<K, false, TIER_WAVES, false>), dim3(d->tier_wgs), dim3(WAVE * TIER_WAVES), tier_lds, st, ta, b, x, d->tier_tri);
    }
    if (d->profile) LS_HIP(hipEventRecord(d->ev[2], st));
    LS_HIP(hipGetLastError());
    LS_HIP(mark_end());
    if (d->profile) {
        LS_HIP(hipStreamSynchronize(st));
        float a = 0, c = 0;
        LS_HIP(hipEventElapsedTime(&a, d->ev[0], d->ev[1]));
        LS_HIP(hipEventElapsedTime(&c, d->ev[1], d->ev[2]));
        d->prof_ms[0] = a; d->prof_ms[1] = c; d->prof_ms[2] = 0.0;
    }
    return LS_OK;
}

extern "C" int ls_direct_solve(ls_direct* d, const float* b, float* x, int k, void* stream) {
    LS_REQUIRE(d && b && x && k >= 1 && k <= d->kmax, LS_E_INVALID, "ls_direct_solve: bad argument (1 <= k <= %d)", d ? d->kmax : 4);
    LS_REQUIRE(b != x, LS_E_INVALID, "ls_direct_solve: b and x must not alias");
    DeviceGuard g(d->device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    // one workspace (b', slots, boundary vectors) per handle: a solve issued on another stream than the previous one
    // waits for it on the device
    // While the stream is being captured into a graph (torch.cuda.graph around a whole optimisation step) the launches become
    // graph nodes in stream order and the handle's event stays out of it: an event recorded inside a capture cannot be
    // waited for outside. Replays of such a graph are NOT serialised against eager solves on other streams by the library.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) (void)hipStreamIsCapturing(st, &cap);
    const bool capturing = cap == hipStreamCaptureStatusActive;
    if (!capturing && d->used && st != d->last_stream) LS_HIP(hipStreamWaitEvent(st, d->busy, 0));
    int rc;
    switch (k) {
        case 1: rc = direct_solve_k<1>(d, b, x, st, -1, nullptr); break;
        case 2: rc = direct_solve_k<2>(d, b, x, st, -1, nullptr); break;
        case 3: rc = direct_solve_k<3>(d, b, x, st, -1, nullptr); break;
        default: rc = direct_solve_k<4>(d, b, x, st, -1, nullptr); break;
    }
    if (rc == LS_OK && !capturing) { LS_HIP(hipEventRecord(d->busy, st)); d->last_stream = st; d->used = true; }
    return rc;
}

extern "C" int ls_direct_solve_part(ls_direct* d, const float* b, float* x, int k, int part, float* exchange, void* stream) {
    LS_REQUIRE(d && b && x && k >= 1 && k <= d->kmax && (part == 0 || part == 1), LS_E_INVALID, "ls_direct_solve_part: bad argument");
    LS_REQUIRE(b != x, LS_E_INVALID, "ls_direct_solve_part: b and x must not alias");
    LS_REQUIRE(exchange || d->exch_f1 == d->exch_f0, LS_E_INVALID, "ls_direct_solve_part: the exchange buffer is missing");
    DeviceGuard g(d->device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    // one workspace per handle, as in ls_direct_solve: a part issued on another stream than the previous call waits for it
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) (void)hipStreamIsCapturing(st, &cap);
    const bool capturing = cap == hipStreamCaptureStatusActive;
    if (!capturing && d->used && st != d->last_stream) LS_HIP(hipStreamWaitEvent(st, d->busy, 0));
    int rc;
    switch (k) {
        case 1: rc = direct_solve_k<1>(d, b, x, st, part, exchange); break;
        case 2: rc = direct_solve_k<2>(d, b, x, st, part, exchange); break;
        case 3: rc = direct_solve_k<3>(d, b, x, st, part, exchange); break;
        default: rc = direct_solve_k<4>(d, b, x, st, part, exchange); break;
    }
    if (rc == LS_OK && !capturing) { LS_HIP(hipEventRecord(d->busy, st)); d->last_stream = st; d->used = true; }
    return rc;
}

// the part of the handle's slot array that the ranks of a sharded solve sum (k columns): ls_dist_direct_solve reduces it in place
extern "C" int ls_direct_exchange_region(ls_direct* d, int k, float** region, int64_t* floats) {
    LS_REQUIRE(d && region && floats && k >= 1 && k <= d->kmax, LS_E_INVALID, "ls_direct_exchange_region: bad argument");
    *region = d->slots + (size_t)d->exch_f0 * d->arity * k;
    *floats = (d->exch_f1 - d->exch_f0) * d->arity * k;
    return LS_OK;
}

extern "C" int ls_direct_shard_info(const ls_direct* d, int* h_rank, int* h_count, int* h_cut_level, int64_t* h_exchange_floats_per_column,
                                    unsigned char* h_owned_rows) {
    LS_REQUIRE(d, LS_E_INVALID, "ls_direct_shard_info: bad argument");
    if (h_rank) *h_rank = d->shard_rank;
    if (h_count) *h_count = d->shard_count;
    if (h_cut_level) *h_cut_level = d->cut;
    if (h_exchange_floats_per_column) *h_exchange_floats_per_column = (d->exch_f1 - d->exch_f0) * d->arity;
    if (h_owned_rows) { if (d->owned_rows.empty()) memset(h_owned_rows, 1, (size_t)d->V); else memcpy(h_owned_rows, d->owned_rows.data(), (size_t)d->V); }
    return LS_OK;
}

extern "C" int ls_direct_set(ls_direct* d, const char* name, int value) {
    LS_REQUIRE(d && name, LS_E_INVALID, "ls_direct_set: bad argument");
    if (!strcmp(name, "profile")) {
        DeviceGuard g(d->device);
        LS_HIP(g.err);
        // 0 off; 1 events around the two sweeps; 3 an event in front of every launch (2 was the tier kernels' clock stamps: archived)
#ifndef LS_TIER_STAMPS
        LS_REQUIRE(value != 2, LS_E_INVALID, "ls_direct_set: profile 2 (per-wave clock stamps of the tier kernels) exists in the experiments build only (tools/build_variant.sh stamps -DLS_TIER_STAMPS)");
#endif
        d->profile = value < 0 ? 0 : std::min(value, 3);
        while (d->profile && d->ev.size() < 4) { hipEvent_t e; LS_HIP(hipEventCreate(&e)); d->ev.push_back(e); }
        return LS_OK;
    }
    if (!strcmp(name, "nt")) {          // cache policy of the read-once factor streams: 0 default policy, 1 non-temporal, -1 back to the library's rule
        d->nt_levels = value < 0 ? d->nt_rule[0] : value != 0;
        d->nt_tier = value < 0 ? d->nt_rule[1] : value != 0;
        return LS_OK;
    }
    set_error("ls_direct_set: unknown option '%s'", name);
    return LS_E_INVALID;
}

int ls_direct_adopt(ls_direct* d, void* const* owned, const size_t* owned_bytes, int n_owned, const double* seconds3, const double* quality4) {
    d->owned.assign(owned, owned + n_owned);
    d->owned_bytes.assign(owned_bytes, owned_bytes + n_owned);
    for (int i = 0; i < 3; ++i) d->factor_s[i] = seconds3[i];
    for (int i = 0; i < 4; ++i) d->plan_q[i] = quality4[i];
    return LS_OK;
}

extern "C" int ls_direct_plan_quality(const ls_direct* d, int* h_ordering, double* h_words_per_vertex, double* h_spread, double* h_words_other) {
    LS_REQUIRE(d, LS_E_INVALID, "ls_direct_plan_quality: bad argument");
    if (h_ordering) *h_ordering = (int)d->plan_q[0];
    if (h_words_per_vertex) *h_words_per_vertex = d->plan_q[1];
    if (h_spread) *h_spread = d->plan_q[2];
    if (h_words_other) *h_words_other = d->plan_q[3];
    return LS_OK;
}

extern "C" int ls_direct_shape(const ls_direct* d, int* h_levels, int* h_arity, int* h_tier_levels, int* h_tier_workgroups,
                               int64_t* h_words_up, int64_t* h_words_down, int64_t* h_n_bnd) {
    LS_REQUIRE(d, LS_E_INVALID, "ls_direct_shape: bad argument");
    if (h_words_up) *h_words_up = d->words_up;
    if (h_words_down) *h_words_down = d->words_down;
    if (h_n_bnd) *h_n_bnd = d->n_bnd;
    if (h_levels) *h_levels = d->levels;
    if (h_arity) *h_arity = d->arity;
    if (h_tier_levels) *h_tier_levels = d->tier_phases;
    if (h_tier_workgroups) *h_tier_workgroups = d->tier_wgs;
    return LS_OK;
}

extern "C" int ls_direct_factor_seconds(const ls_direct* d, double* h_s3) {
    LS_REQUIRE(d && h_s3, LS_E_INVALID, "ls_direct_factor_seconds: bad argument");
    for (int i = 0; i < 3; ++i) h_s3[i] = d->factor_s[i];
    return LS_OK;
}

extern "C" int ls_direct_level_words(const ls_direct* d, int cap, int64_t* h_up, int64_t* h_down) {
    LS_REQUIRE(d && cap >= 0, LS_E_INVALID, "ls_direct_level_words: bad argument");
    for (int lv = 0; lv < cap && lv < d->levels; ++lv) {
        if (h_up) h_up[lv] = lv < (int)d->lvl_up.size() ? d->lvl_up[(size_t)lv] : 0;
        if (h_down) h_down[lv] = lv < (int)d->lvl_down.size() ? d->lvl_down[(size_t)lv] : 0;
    }
    return LS_OK;
}

extern "C" int ls_direct_level_rows(const ls_direct* d, int cap, int64_t* h_rows, int64_t* h_bnd) {
    LS_REQUIRE(d && cap >= 0, LS_E_INVALID, "ls_direct_level_rows: bad argument");
    for (int lv = 0; lv < cap && lv < d->levels; ++lv) {
        if (h_rows) h_rows[lv] = lv < (int)d->lvl_rows.size() ? d->lvl_rows[(size_t)lv] : 0;
        if (h_bnd) h_bnd[lv] = lv < (int)d->lvl_bnd.size() ? d->lvl_bnd[(size_t)lv] : 0;
    }
    return LS_OK;
}

#ifdef LS_TIER_STAMPS
// experiments build only (tools/build_variant.sh stamps "-DLS_TIER_STAMPS"; not declared in the header, not in the product library)
extern "C" int ls_direct_tier_stamps(ls_direct* d, long long* h_out, int64_t n) {
    LS_REQUIRE(d && h_out && d->stamps, LS_E_STATE, "ls_direct_tier_stamps: no stamps (solve with \"profile\" = 2 first)");
    const int64_t have = 2 * (int64_t)d->tier_wgs * d->tier_waves * TIER_STAMP_SLOTS;
    LS_REQUIRE(n >= have, LS_E_WORKSPACE, "ls_direct_tier_stamps: %lld entries needed", (long long)have);
    DeviceGuard g(d->device);
    LS_HIP(hipDeviceSynchronize());
    LS_HIP(hipMemcpy(h_out, d->stamps, (size_t)have * sizeof(long long), hipMemcpyDeviceToHost));
    return LS_OK;
}
#endif

extern "C" int ls_direct_level_index_bytes(const ls_direct* d, int cap, int64_t* h_up, int64_t* h_down) {
    LS_REQUIRE(d && cap >= 0, LS_E_INVALID, "ls_direct_level_index_bytes: bad argument");
    for (int lv = 0; lv < cap && lv < d->levels; ++lv) {
        if (h_up) h_up[lv] = lv < (int)d->lvl_idx_up.size() ? d->lvl_idx_up[(size_t)lv] : 0;
        if (h_down) h_down[lv] = lv < (int)d->lvl_idx_down.size() ? d->lvl_idx_down[(size_t)lv] : 0;
    }
    return LS_OK;
}

extern "C" int ls_direct_tier_balance(const ls_direct* d, double* h_words4) {
    LS_REQUIRE(d && h_words4, LS_E_INVALID, "ls_direct_tier_balance: bad argument");
    for (int i = 0; i < 4; ++i) h_words4[i] = d->tier_balance[i];
    return LS_OK;
}

extern "C" int ls_direct_launch_profile(const ls_direct* d, int cap, int* h_n, double* h_ms, int64_t* h_words, int32_t* h_level_lo,
                                        int32_t* h_level_hi, int32_t* h_sweep) {
    LS_REQUIRE(d && cap >= 0, LS_E_INVALID, "ls_direct_launch_profile: bad argument");
    const int n = (int)d->launch_ms.size();
    if (h_n) *h_n = n;
    for (int i = 0; i < n && i < cap; ++i) {
        const ls_direct::LaunchMeta m = d->lmeta[(size_t)i];
        int64_t w = 0;
        for (int lv = std::max(m.lo, 0); lv <= m.hi && lv < d->levels; ++lv) {
            if (m.sweep != 1) w += d->lvl_up[(size_t)lv];
            if (m.sweep != 0) w += d->lvl_down[(size_t)lv];
        }
        if (h_ms) h_ms[i] = d->launch_ms[(size_t)i];
        if (h_words) h_words[i] = w;
        if (h_level_lo) h_level_lo[i] = m.lo;
        if (h_level_hi) h_level_hi[i] = m.hi;
        if (h_sweep) h_sweep[i] = m.sweep;
    }
    return LS_OK;
}

extern "C" int ls_direct_info(const ls_direct* d, int64_t* h_factor_entries, int* h_launches, double* h_ms3) {
    LS_REQUIRE(d, LS_E_INVALID, "ls_direct_info: bad argument");
    if (h_factor_entries) *h_factor_entries = d->factor_entries;
    if (h_launches) {
        int n = 0;
        for (int lv = 0; lv < d->levels; ++lv) {
            const LevelPlan& p = d->plan[lv];
            const bool fused = lv == 0 && d->tier_root > 0 && d->fuse_root && !p.down_p && !p.down_s;     // the root's up step rides in its down tiles
            n += (((p.up_p ? p.up_p_tiles : p.up_tiles) && !fused) ? 1 : 0) + ((p.down_p ? p.down_p_tiles : p.down_tiles) ? 1 : 0);
        }
        *h_launches = n + (d->tier_wgs ? 2 : 0);
    }
    if (h_ms3) for (int i = 0; i < 3; ++i) h_ms3[i] = d->prof_ms[i];
    return LS_OK;
}
