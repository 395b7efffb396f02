// radix.h -- stable LSD radix sort of row ids by a key policy (hand-written: per pass a histogram kernel, the scan of assemble.hip
// and a scatter kernel in which ONE wave walks its chunk in order and ranks equal digits inside every 64-element tile with ballots:
// stable by construction, no atomics in the scatter). Used by remove_duplicates, the CSR transpose, the corner ranking
// (assemble.hip) and the coordinate orders of the device bisection (nd_bisect.hip).
#pragma once
#include "common.h"

namespace ls {

// elements per workgroup (histogram: 256 threads; scatter: ONE wave walks them 64 at a time, in order -- a serial chain of dependent
// gathers whose length is the chunk, whatever the input's size): the chunk follows the input so that small sorts still fill the chip
// (round 4, tools/time_dedup.py: remove_duplicates of the 250k config's soup 1.36 -> 0.94 ms, 70k 1.10 -> 0.53 ms with 1024 instead of
// 4096; the 6M-row soup of the 1M config is best at 4096: 3.15 against 3.56 ms). Workspace formulas use the same function.
__host__ __device__ constexpr int rs_chunk(int64_t n) { return n > ((int64_t)4 << 20) ? 4096 : n > ((int64_t)2 << 20) ? 2048 : 1024; }

// order-preserving map of a float to uint32; -0.0 is folded into +0.0 first (torch compares values)
__device__ __forceinline__ unsigned key_of(float x) {
    unsigned u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// key policies of the radix sort: the 12 bytes of a vertex row (x most significant) / the 4 bytes of an int32 key
struct KeyVerts {
    const float* verts;
    __device__ __forceinline__ unsigned digit(int row, int pass) const { return (key_of(verts[3 * (size_t)row + (2 - pass / 4)]) >> (8 * (pass & 3))) & 255u; }
};
struct KeyInt {
    const int* keys;
    __device__ __forceinline__ unsigned digit(int row, int pass) const { return ((unsigned)keys[row] >> (8 * pass)) & 255u; }
};

template <typename Key>
__global__ __launch_bounds__(256) void k_rs_hist(Key key, const int* __restrict__ order, int64_t n, int pass,
                                                 int nblocks, int* __restrict__ hist /* [256][nblocks] */) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int RS_CHUNK = rs_chunk(n);
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    for (int e = threadIdx.x; e < RS_CHUNK && base + e < n; e += 256) atomicAdd(&h[key.digit(order ? order[base + e] : (int)(base + e), pass)], 1);
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <typename Key>
__global__ __launch_bounds__(64) void k_rs_scatter(Key key, const int* __restrict__ order, int64_t n, int pass,
                                                   int nblocks, const int* __restrict__ offs /* scanned hist */, int* __restrict__ out) {
    __shared__ int run[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) run[d] = offs[(size_t)d * nblocks + blockIdx.x];
    __syncthreads();
    const int RS_CHUNK = rs_chunk(n);
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int t = 0; t < RS_CHUNK && base + t < n; t += 64) {
        const int64_t e = base + t + lane;
        const bool ok = e < n;
        const int row = ok ? (order ? order[e] : (int)e) : 0;
        const unsigned dg = ok ? key.digit(row, pass) : 0u;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long b = __ballot((dg >> bit) & 1u);
            peers &= ((dg >> bit) & 1u) ? b : ~b;
        }
        const int rank = __popcll(peers & lt), cnt = __popcll(peers);
        const int start = ok ? run[dg] : 0;
        __syncthreads();                       // one wave: orders the reads of run[] before the updates below
        if (ok) {
            out[start + rank] = row;
            if (rank == cnt - 1) run[dg] = start + cnt;
        }
        __syncthreads();
    }
}

// order-preserving map of a double to uint64; -0.0 is folded into +0.0 (the host compares values: std::pair<double, int>)
__device__ __forceinline__ unsigned long long key_of64(double x) {
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    if (u == 0x8000000000000000ull) u = 0ull;
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
// 8 key bytes: coordinate `axis` of the row's position (V x 3 doubles)
struct KeyF64 {
    const double* pos;
    int axis;
    __device__ __forceinline__ unsigned digit(int row, int pass) const { return (unsigned)(key_of64(pos[3 * (size_t)row + axis]) >> (8 * pass)) & 255u; }
};

}  // namespace ls

// order = the ids 0..n-1 sorted stably by `passes` key bytes; tmp: n ints; hist / offs: 256 nb + 16 ints each; returns where the result is
template <typename Key>
static inline int radix_argsort(Key key, int64_t n, int passes, int* ord_a, int* ord_b, int* hist, int* offs, int* bsum, hipStream_t st, const int** result) {
    const int nb = ls::div_up(n, ls::rs_chunk(n));
    const int* src = nullptr;                  // pass 0 reads the identity order
    int* dst = ord_a;
    for (int pass = 0; pass < passes; ++pass) {
        hipLaunchKernelGGL(ls::k_rs_hist<Key>, dim3(nb), dim3(256), 0, st, key, src, n, pass, nb, hist);
        int rc = ls::exclusive_scan(hist, 256 * (int64_t)nb, offs, bsum, st);
        if (rc) return rc;
        hipLaunchKernelGGL(ls::k_rs_scatter<Key>, dim3(nb), dim3(64), 0, st, key, src, n, pass, nb, (const int*)offs, dst);
        src = dst;
        dst = (dst == ord_a) ? ord_b : ord_a;
    }
    *result = src;
    return LS_OK;
}

