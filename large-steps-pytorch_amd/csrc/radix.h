// radix.h -- stable LSD radix sort of row ids by a key policy (hand-written: per pass a histogram kernel, the scan of assemble.hip
// and a scatter kernel in which ONE wave walks its chunk in order and ranks equal digits inside every 64-element tile with ballots:
// stable by construction, no atomics in the scatter). Used by remove_duplicates, the CSR transpose, the corner ranking
// (assemble.hip) and the coordinate orders of the device bisection (nd_bisect.hip).
#pragma once
#include "common.h"
#include <utility>

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
    __device__ __forceinline__ unsigned word(int row, int w) const { return key_of(verts[3 * (size_t)row + (2 - w)]); }      // 32-bit key word w, least significant first (digit(row, 4 w + i) = byte i of it)
};
struct KeyInt {
    const int* keys;
    __device__ __forceinline__ unsigned digit(int row, int pass) const { return ((unsigned)keys[row] >> (8 * pass)) & 255u; }
    __device__ __forceinline__ unsigned word(int row, int) const { return (unsigned)keys[row]; }
};

// lanes of the wave that hold the same digit as this lane (eight ballots); `ok` = the lane takes part
__device__ __forceinline__ unsigned long long rs_peers(unsigned dg, bool ok) {
    unsigned long long peers = __ballot(ok);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long b = __ballot((dg >> bit) & 1u);
        peers &= ((dg >> bit) & 1u) ? b : ~b;
    }
    return peers;
}

// Both kernels ask for RS_U tiles of 64 rows at once -- the row ids, then the key bytes behind them: two dependent round trips per RS_U
// tiles instead of per tile (the scatter is one wave walking its chunk in order: at 4096 rows its 64 tiles were a chain of 128 round
// trips, 111 us per pass over the 6M-row soup of a 1M-vertex remesh) -- and the histogram counts a wave's equal digits with ballots and
// adds ONCE per group (sorted-ish coordinates put most of a wave's rows into one or two bins: 95 % of its LDS cycles were same-address
// atomics, profiles/r04_pmc_sq_counters.txt).
constexpr int RS_U = 4;

template <typename Key>
__global__ __launch_bounds__(256) void k_rs_hist(Key key, const int* __restrict__ order, int64_t n, int pass,
                                                 int nblocks, int* __restrict__ hist /* [256][nblocks] */) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int RS_CHUNK = rs_chunk(n);
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int e0 = 0; e0 < RS_CHUNK && base + e0 < n; e0 += 256 * RS_U) {        // (uniform over the workgroup)
        int row[RS_U];
        unsigned dg[RS_U];
        bool ok[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const int e = e0 + u * 256 + threadIdx.x;
            ok[u] = e < RS_CHUNK && base + e < n;
            row[u] = ok[u] ? (order ? order[base + e] : (int)(base + e)) : 0;
        }
#pragma unroll
        for (int u = 0; u < RS_U; ++u) dg[u] = ok[u] ? key.digit(row[u], pass) : 0u;
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const unsigned long long peers = rs_peers(dg[u], ok[u]);
            if (ok[u] && (peers & lt) == 0ull) atomicAdd(&h[dg[u]], __popcll(peers));
        }
    }
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
    for (int t0 = 0; t0 < RS_CHUNK && base + t0 < n; t0 += 64 * RS_U) {
        int row[RS_U];
        unsigned dg[RS_U];
        bool ok[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const int t = t0 + u * 64;
            const int64_t e = base + t + lane;
            ok[u] = t < RS_CHUNK && e < n;
            row[u] = ok[u] ? (order ? order[e] : (int)e) : 0;
        }
#pragma unroll
        for (int u = 0; u < RS_U; ++u) dg[u] = ok[u] ? key.digit(row[u], pass) : 0u;
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {                       // the tiles in order: ranks inside a tile by ballots, running offsets per digit in LDS
            const unsigned long long peers = rs_peers(dg[u], ok[u]);
            const int rank = __popcll(peers & lt), cnt = __popcll(peers);
            const int start = ok[u] ? run[dg[u]] : 0;
            __syncthreads();                       // one wave: orders the reads of run[] before the updates below
            if (ok[u]) {
                out[start + rank] = row[u];
                if (rank == cnt - 1) run[dg[u]] = start + cnt;
            }
            __syncthreads();
        }
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
    __device__ __forceinline__ unsigned word(int row, int w) const { return (unsigned)(key_of64(pos[3 * (size_t)row + axis]) >> (32 * w)); }
};

// ---- the same sort with the keys CARRIED: a key that sits behind a gather (a vertex row, a position) costs every pass a random 4- or
// 8-byte read per row, i.e. a 64-byte sector -- 400 MB per kernel over the 6M-row soup of a 1M-vertex remesh, which is what the byte-wise
// passes above take their ~100 us for. Here a 32-bit key word is gathered ONCE into an array in the current order (k_rs_load) and its
// four byte passes read and move (key, id) pairs with unit stride; the next word is gathered through the ids again. Same digits in the
// same order as radix_argsort: the same permutation, bit for bit.
template <typename Key>
__global__ __launch_bounds__(256) void k_rs_load(Key key, const int* __restrict__ order, int64_t n, int w, unsigned* __restrict__ keys) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) keys[e] = key.word(order ? order[e] : (int)e, w);
}

template <int UNIT = 0>      // (a template only so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void k_rs_hist32(const unsigned* __restrict__ keys, int64_t n, int shift, int nblocks, int* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int RS_CHUNK = rs_chunk(n);
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int e0 = 0; e0 < RS_CHUNK && base + e0 < n; e0 += 256 * RS_U) {
        unsigned dg[RS_U];
        bool ok[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const int e = e0 + u * 256 + threadIdx.x;
            ok[u] = e < RS_CHUNK && base + e < n;
            dg[u] = ok[u] ? (keys[base + e] >> shift) & 255u : 0u;
        }
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const unsigned long long peers = rs_peers(dg[u], ok[u]);
            if (ok[u] && (peers & lt) == 0ull) atomicAdd(&h[dg[u]], __popcll(peers));
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// ids == nullptr: the identity (the very first pass); keys_out == nullptr: the word's last pass (its keys are not needed again)
template <int UNIT = 0>
__global__ __launch_bounds__(64) void k_rs_scatter32(const unsigned* __restrict__ keys, const int* __restrict__ ids, int64_t n, int shift, int nblocks,
                                                     const int* __restrict__ offs, unsigned* __restrict__ keys_out, int* __restrict__ ids_out) {
    __shared__ int run[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) run[d] = offs[(size_t)d * nblocks + blockIdx.x];
    __syncthreads();
    const int RS_CHUNK = rs_chunk(n);
    const int64_t base = (int64_t)blockIdx.x * RS_CHUNK;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int t0 = 0; t0 < RS_CHUNK && base + t0 < n; t0 += 64 * RS_U) {
        unsigned kv[RS_U];
        int id[RS_U];
        bool ok[RS_U];
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const int t = t0 + u * 64;
            const int64_t e = base + t + lane;
            ok[u] = t < RS_CHUNK && e < n;
            kv[u] = ok[u] ? keys[e] : 0u;
            id[u] = ok[u] ? (ids ? ids[e] : (int)e) : 0;
        }
#pragma unroll
        for (int u = 0; u < RS_U; ++u) {
            const unsigned dg = (kv[u] >> shift) & 255u;
            const unsigned long long peers = rs_peers(dg, ok[u]);
            const int rank = __popcll(peers & lt), cnt = __popcll(peers);
            const int start = ok[u] ? run[dg] : 0;
            __syncthreads();
            if (ok[u]) {
                ids_out[start + rank] = id[u];
                if (keys_out) keys_out[start + rank] = kv[u];
                if (rank == cnt - 1) run[dg] = start + cnt;
            }
            __syncthreads();
        }
    }
}

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

// the same order (bit for bit) by `words` 32-bit key words, keys carried; keys_a / keys_b: n unsigned each; last_bytes: byte passes of the LAST word
// (keys known to be small: fewer passes)
template <typename Key>
static inline int radix_argsort_words(Key key, int64_t n, int words, int* ord_a, int* ord_b, unsigned* keys_a, unsigned* keys_b, int* hist, int* offs, int* bsum,
                                      hipStream_t st, const int** result, int last_bytes = 4) {
    const int nb = ls::div_up(n, ls::rs_chunk(n));
    const int* src = nullptr;                  // word 0 is loaded in the identity order
    int* dst = ord_a;
    for (int w = 0; w < words; ++w) {
        hipLaunchKernelGGL(ls::k_rs_load<Key>, dim3(ls::div_up(n, 256)), dim3(256), 0, st, key, src, n, w, keys_a);
        unsigned* kin = keys_a;
        unsigned* kout = keys_b;
        const int nbytes = w + 1 == words ? last_bytes : 4;
        for (int byte = 0; byte < nbytes; ++byte) {
            hipLaunchKernelGGL(ls::k_rs_hist32<0>, dim3(nb), dim3(256), 0, st, (const unsigned*)kin, n, 8 * byte, nb, hist);
            int rc = ls::exclusive_scan(hist, 256 * (int64_t)nb, offs, bsum, st);
            if (rc) return rc;
            hipLaunchKernelGGL(ls::k_rs_scatter32<0>, dim3(nb), dim3(64), 0, st, (const unsigned*)kin, src, n, 8 * byte, nb, (const int*)offs,
                               byte + 1 < nbytes ? kout : (unsigned*)nullptr, dst);
            src = dst;
            dst = (dst == ord_a) ? ord_b : ord_a;
            std::swap(kin, kout);
        }
    }
    *result = src;
    return LS_OK;
}

