// assemble.hip -- sort-free CSR assembly of M = a*I + b*L (uniform / cotangent Laplacian) on gfx950.
//
// Replaces largesteps/geometry.py:3-133 of the reference, which builds the same matrix through
// torch.unique(dim=1) + sparse add + coalesce() (several device-wide radix sorts and host syncs).
// Here:   rank the face corners per vertex in LDS and reserve their slots (one returning atomic per vertex pair and workgroup)
//         -> one-launch scan of the row counters -> atomic-free scatter of the half-edges into per-row slots
//         -> per-row sort (in registers) + merge -> scan of the tile totals -> LDS-staged coalesced emit.
// Semantics reproduced exactly:
//   * uniform: every undirected edge once per direction (dedup), diag = #distinct neighbours,
//     off-diag = fl(b*(-1)), diag = fl(a + fl(b*deg)); unreferenced vertices keep only the a*I entry.
//   * cot: per face fp32 Heron area with the 1e-12 clamp, weights /4, contributions of the incident
//     faces are scaled first (fl(b*(-w))) and then summed; diag = fl(a + fl(b * sum_w)).
// The fp32 operation order of the cotangents is the one of oracle/laplacian.py (explicit fma chain
// in the edge norm, everything else unfused): this file is compiled with -ffp-contract=off.
#include "common.h"
#include "radix.h"
#include <algorithm>

#pragma STDC FP_CONTRACT OFF

namespace ls {

// ------------------------------------------------------------------------------------------------
// workspace layout (shared by ls_assemble_pattern and ls_assemble_fill; depends on V and F only)
// ------------------------------------------------------------------------------------------------
struct AsmLayout {
    size_t flags, chain_rows, cnt, slot_ptr, row_off, diag, tile_cnt, tile_off, corner_off, slot_col, slot_val, comp, total;
    int64_t nslots, tiles;
    AsmLayout(int64_t V, int64_t F) {
        auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
        nslots = 6 * F;
        tiles = (V + TILE_ROWS - 1) / TILE_ROWS;
        size_t o = 0;
        // what has to be ZERO when the kernels start comes first: one memset over [0, slot_ptr)
        flags = o;     o = al(o + 64);
        chain_rows = o;  o = al(o + 8 * (size_t)((V + 1) / 4096 + 4));   // single-pass scan of the row counters: ticket + one word per 4096 rows
        cnt = o;       o = al(o + 4 * (size_t)(V + 2));        // (k_count reserves for vertex pairs with 64-bit atomics: one entry of slack)
        slot_ptr = o;  o = al(o + 4 * (size_t)(V + 1));
        row_off = o;   o = al(o + 4 * (size_t)(V + 1));        // offset of a row's first entry inside its tile's compact block
        diag = o;      o = al(o + 4 * (size_t)(V + 1));
        tile_cnt = o;  o = al(o + 4 * (size_t)(tiles + 1));    // entries of a 256-row tile
        tile_off = o;  o = al(o + 4 * (size_t)(tiles + 1));    // ... and their exclusive scan: where the tile's rows start in the CSR arrays
        corner_off = o; o = al(o + 4 * (size_t)(3 * F) + 64);  // where a face corner's two slots sit inside its vertex' row (k_count)
        slot_col = o;  o = al(o + 4 * (size_t)nslots + 64);    // (+ 64: a row's last 16-byte quad may reach two slots past the array)
        slot_val = o;  o = al(o + 4 * (size_t)nslots + 64);
        comp = o;      o = al(o + 8 * (size_t)(nslots + V));   // final rows, tile by tile: tile t at entry slot_ptr[256 t] + 256 t ({col, value bits})
        total = o;
    }
};

// ------------------------------------------------------------------------------------------------
// exclusive scan of int32 (three kernels; the middle one is a single workgroup)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = BLOCK * SCAN_ITEMS;   // 2048 elements per workgroup

__device__ __forceinline__ int wave_inclusive_scan(int x) {
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        int y = __shfl_up(x, off, WAVE);
        if (lane >= off) x += y;
    }
    return x;
}

// exclusive scan of one value per thread across the block; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int x, int* total, int* smem /* BLOCK/WAVE + 1 */) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    int inc = wave_inclusive_scan(x);
    if (lane == WAVE - 1) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int j = 0; j < BLOCK / WAVE; ++j) { int t = smem[j]; smem[j] = run; run += t; }
        smem[BLOCK / WAVE] = run;
    }
    __syncthreads();
    int res = inc - x + smem[w];
    *total = smem[BLOCK / WAVE];
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(BLOCK) void k_scan_reduce(const int* __restrict__ in, int64_t n, int* __restrict__ bsum) {
    __shared__ int smem[BLOCK / WAVE + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
    int total;
    block_exclusive_scan(s, &total, smem);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLOCK) void k_scan_bsums(int* __restrict__ bsum, int nb) {   // <<<1, BLOCK>>>
    __shared__ int smem[BLOCK / WAVE + 1];
    int carry = 0;
    for (int base = 0; base < nb; base += BLOCK) {
        const int i = base + threadIdx.x;
        int v = (i < nb) ? bsum[i] : 0;
        int total;
        int ex = block_exclusive_scan(v, &total, smem);
        if (i < nb) bsum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) bsum[nb] = carry;   // grand total
}

__global__ __launch_bounds__(BLOCK) void k_scan_final(const int* __restrict__ in, int64_t n, const int* __restrict__ bsum,
                                                      int* __restrict__ out /* n + 1 entries */) {
    __shared__ int smem[BLOCK / WAVE + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
    int total;
    int run = block_exclusive_scan(s, &total, smem) + bsum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = bsum[gridDim.x];
}

// out[0..n] = exclusive scan of in[0..n); out[n] = total. `in` and `out` may not alias.
int exclusive_scan(const int* in, int64_t n, int* out, int* bsum, hipStream_t st) {      // declared in common.h (nd_factor.hip uses it too)
    if (n <= 0) { LS_HIP(hipMemsetAsync(out, 0, sizeof(int), st)); return LS_OK; }
    const int nb = div_up(n, SCAN_CHUNK);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(BLOCK), 0, st, in, n, bsum);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(BLOCK), 0, st, bsum, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(BLOCK), 0, st, in, n, bsum, out);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ------------------------------------------------------------------------------------------------
// single-pass exclusive scan (round 5): the workgroups hand their sums along a chain of status words ("decoupled look-back")
// instead of meeting at two kernel boundaries. state[0] = ticket counter, state[1 + k] = status << 62 | value of chunk k -- status 1:
// the chunk's own sum, 2: the sum of everything up to and including the chunk. ALL ZERO before the launch (the caller's memset).
// A workgroup takes its chunk from the ticket counter, so a chunk's predecessors are running or done whatever order the hardware
// starts workgroups in: the spin below always ends. Status and value share one 64-bit word: no fence between them.
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long CH_AGG = 1ull << 62, CH_INC = 2ull << 62;
__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int chain_ticket(unsigned long long* state, int* s_tk) {      // all threads; *s_tk in LDS
    if (threadIdx.x == 0) *s_tk = (int)atomicAdd(state, 1ull);
    __syncthreads();
    return *s_tk;
}
__device__ __forceinline__ void chain_publish(unsigned long long* state, int k, int agg) {   // one thread
    chain_store(state + 1 + k, (k == 0 ? CH_INC : CH_AGG) | (unsigned long long)(unsigned)agg);
}
// all 64 lanes of ONE wave, after chain_publish(k, agg): the sum of the chunks before k. A lane looks at CH_PER consecutive chunks
// (lane 0 the nearest), the window of 64 x CH_PER chunks moves back until one of them knows its inclusive sum. A hop of the chain is a
// store becoming visible to another CU's load (~1 us), so the chain pays for FEW chunks only: the row counters of 1M vertices are 245
// chunks = one window. (The same chain inside k_row_merge -- 3907 tiles handing their entry counts along, to drop k_tile_scan and
// k_rowptr -- made that kernel 67 us with a window of 64 and 88 us with 256 instead of 23: measured and taken out.)
constexpr int CH_PER = 4;
__device__ __forceinline__ int chain_lookback(unsigned long long* state, int k, int agg) {
    const int lane = threadIdx.x & (WAVE - 1);
    int prefix = 0;
    for (int j0 = k - 1 - lane * CH_PER;; j0 -= WAVE * CH_PER) {
        unsigned long long w[CH_PER];
#pragma unroll
        for (int e = 0; e < CH_PER; ++e) w[e] = j0 - e >= 0 ? chain_load(state + 1 + j0 - e) : CH_INC;      // before chunk 0: an inclusive sum of 0
        bool pending;
        do {
            pending = false;
#pragma unroll
            for (int e = 0; e < CH_PER; ++e) {
                if ((w[e] >> 62) == 0) { w[e] = chain_load(state + 1 + j0 - e); pending |= (w[e] >> 62) == 0; }
            }
            if (pending) __builtin_amdgcn_s_sleep(2);               // a predecessor is still summing: do not hammer its status word
        } while (pending);
        int v = 0;
        bool has = false;                                           // this lane holds a chunk that knows its inclusive sum
#pragma unroll
        for (int e = 0; e < CH_PER; ++e) {
            if (!has) { v += (int)(unsigned)w[e]; has = (w[e] >> 62) == 2; }
        }
        const unsigned long long inc = __ballot(has);
        const int first = inc ? __ffsll((long long)inc) - 1 : WAVE;   // the nearest lane with one
        v = lane <= first ? v : 0;
#pragma unroll
        for (int o = WAVE / 2; o; o >>= 1) v += __shfl_xor(v, o, WAVE);
        prefix += v;
        if (inc) break;
    }
    if (lane == 0 && k > 0) chain_store(state + 1 + k, CH_INC | (unsigned long long)(unsigned)(prefix + agg));
    return prefix;
}

constexpr int CH_ITEMS = 16;
constexpr int CH_CHUNK = BLOCK * CH_ITEMS;       // 4096 elements per workgroup
typedef int i4a_scan __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(BLOCK) void k_scan_chained(const int* __restrict__ in, int64_t n, int* __restrict__ out /* n + 1 */,
                                                        unsigned long long* __restrict__ state) {
    __shared__ int smem[BLOCK / WAVE + 1];
    __shared__ int s_tk, s_pref;
    const int k = chain_ticket(state, &s_tk);
    const int64_t base = (int64_t)k * CH_CHUNK + (int64_t)threadIdx.x * CH_ITEMS;
    const bool quad = base + CH_ITEMS <= n && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
    int v[CH_ITEMS];
    if (quad) {
#pragma unroll
        for (int q = 0; q < CH_ITEMS / 4; ++q) {
            const i4a_scan t = *reinterpret_cast<const i4a_scan*>(in + base + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH_ITEMS; ++i) v[i] = (base + i < n) ? in[base + i] : 0;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < CH_ITEMS; ++i) s += v[i];
    int total;
    int run = block_exclusive_scan(s, &total, smem);
    if (threadIdx.x < WAVE) {
        if (threadIdx.x == 0) chain_publish(state, k, total);
        const int p = chain_lookback(state, k, total);
        if (threadIdx.x == 0) s_pref = p;
    }
    __syncthreads();
    run += s_pref;
    if (quad) {
#pragma unroll
        for (int q = 0; q < CH_ITEMS / 4; ++q) {
            i4a_scan t;
            t.x = run; run += v[4 * q]; t.y = run; run += v[4 * q + 1]; t.z = run; run += v[4 * q + 2]; t.w = run; run += v[4 * q + 3];
            *reinterpret_cast<i4a_scan*>(out + base + 4 * q) = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH_ITEMS; ++i) {
            if (base + i < n) out[base + i] = run;
            run += v[i];
        }
    }
    if (threadIdx.x == 0 && (int64_t)(k + 1) * CH_CHUNK >= n) out[n] = s_pref + total;
}

// out[0..n] = exclusive scan of in[0..n) in ONE launch; `state`: 8 * (ceil(n / 4096) + 1) bytes, ZERO on entry
int exclusive_scan_chained(const int* in, int64_t n, int* out, unsigned long long* state, hipStream_t st) {
    if (n <= 0) { LS_HIP(hipMemsetAsync(out, 0, sizeof(int), st)); return LS_OK; }
    hipLaunchKernelGGL(k_scan_chained, dim3((unsigned)div_up(n, CH_CHUNK)), dim3(BLOCK), 0, st, in, n, out, state);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ------------------------------------------------------------------------------------------------
// 1. count half-edges per row, validate indices, and give every face corner its place inside its vertex' row.
//    Round 5: one global atomic per DISTINCT vertex pair of a workgroup's faces instead of one per corner. The corners are first ranked
//    per vertex in an LDS hash table (ds_cmpst / ds_add: open addressing; a corner that finds no entry within CNT_PROBES steps -- a
//    soup whose faces share no vertices -- reserves with its own atomic, the round-4 way), then every occupied entry reserves its vertex' slots with ONE returning atomic on the row counter, and a corner's offset = that base + 2 x its
//    rank goes to corner_off. The scatter needs no atomics at all. (Round 1-4: 6 M atomics in k_count + 6 M returning ones in k_scatter,
//    76 + 189 us of a 0.5 ms assembly at 1M vertices; meshes are stored coherently enough that a vertex' faces meet in few workgroups.)
//    Which slots a corner gets depends on the order of the atomics; the final matrix does not (k_row_merge sorts every row).
// ------------------------------------------------------------------------------------------------
// Workgroup size of the ranking, measured at 1M vertices (profiles/r05_assembler_count_variants.txt): the kernel's phases (face stream,
// LDS hashing, returning atomics) only overlap ACROSS workgroups, so many small ones win -- 1024 faces and a 4096-entry table (3 per
// CU, all 2048 resident at once and in step) 64 us, 512 faces / 2048 entries 59, 512 / 512 35, 256 faces / 512 entries 31.
constexpr int CNT_FPT = 1;                       // faces per thread
constexpr int CNT_HT = 512;                      // hash table entries (power of two); sized for meshes, see CNT_PROBES
constexpr int CNT_PROBES = 8;                    // entries a corner tries before it reserves directly
static_assert((CNT_HT & (CNT_HT - 1)) == 0 && CNT_HT <= 4096 && CNT_HT >= 64, "hash table of k_count");
template <typename IdxT>
__global__ __launch_bounds__(BLOCK) void k_count(const IdxT* __restrict__ faces, int64_t F, int64_t V,
                                                 int* __restrict__ cnt, int* __restrict__ corner_off, int* __restrict__ flags) {
    // an entry serves the vertex PAIR (2 p, 2 p + 1): neighbouring ids meet in the same faces, and one 64-bit atomic reserves for both
    // (the returning atomics are what this kernel waits for: ~30 ns each at the L2 however few bytes they carry)
    __shared__ int h_key[CNT_HT];
    __shared__ int h_cnt[2 * CNT_HT];            // corners of the pair's even / odd vertex in this workgroup, then the bases of their reservations
    for (int t = threadIdx.x; t < CNT_HT; t += BLOCK) { h_key[t] = -1; h_cnt[2 * t] = 0; h_cnt[2 * t + 1] = 0; }
    __syncthreads();
    const int64_t f0 = (int64_t)blockIdx.x * (BLOCK * CNT_FPT);
    int slot[3 * CNT_FPT], rank[3 * CNT_FPT];
#pragma unroll
    for (int j = 0; j < CNT_FPT; ++j) {
        const int64_t f = f0 + (int64_t)j * BLOCK + threadIdx.x;
        int64_t v[3] = {-1, -1, -1};
        if (f < F) {
            v[0] = (int64_t)faces[3 * f + 0]; v[1] = (int64_t)faces[3 * f + 1]; v[2] = (int64_t)faces[3 * f + 2];
            if (v[0] < 0 || v[1] < 0 || v[2] < 0 || v[0] >= V || v[1] >= V || v[2] >= V) { flags[0] = 1; v[0] = v[1] = v[2] = -1; }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            int h = -1, r = 0;
            if (v[e] >= 0) {
                const int key = (int)(v[e] >> 1);
                h = (int)(((unsigned)key * 2654435761u) >> 20) & (CNT_HT - 1);
                bool found = false;
#pragma unroll 1
                for (int probe = 0; probe < CNT_PROBES; ++probe) {
                    const int old = atomicCAS(&h_key[h], -1, key);
                    if (old == -1 || old == key) { found = true; break; }
                    h = (h + 1) & (CNT_HT - 1);
                }
                if (found) {
                    h = 2 * h + (int)(v[e] & 1);
                    r = atomicAdd(&h_cnt[h], 1);
                } else {
                    // the table is sized for meshes (a workgroup's faces share most of their vertices), not for the worst case: a corner
                    // that finds no entry reserves its two slots directly
                    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(cnt) + key, (v[e] & 1) ? (2ull << 32) : 2ull);
                    h = -2;
                    r = (int)(unsigned)((v[e] & 1) ? (old >> 32) : old);
                }
            }
            slot[3 * j + e] = h; rank[3 * j + e] = r;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CNT_HT; t += BLOCK) {
        const int key = h_key[t];
        if (key >= 0) {                                                    // each corner puts two half-edges into its vertex' row
            const unsigned long long add = (unsigned long long)(unsigned)(2 * h_cnt[2 * t]) | ((unsigned long long)(unsigned)(2 * h_cnt[2 * t + 1]) << 32);
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(cnt) + key, add);    // (cnt is 256-byte aligned, V + 2 entries)
            h_cnt[2 * t] = (int)(unsigned)old; h_cnt[2 * t + 1] = (int)(unsigned)(old >> 32);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CNT_FPT; ++j) {
        const int64_t f = f0 + (int64_t)j * BLOCK + threadIdx.x;
        if (f < F) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int sl = slot[3 * j + e];
                corner_off[3 * f + e] = sl >= 0 ? h_cnt[sl] + 2 * rank[3 * j + e] : (sl == -2 ? rank[3 * j + e] : -1);
            }
        }
    }
}

// fp32 cotangents of one face in the reference's operation order (geometry.py:20-41, oracle/laplacian.py)
__device__ __forceinline__ float edge_norm(float ax, float ay, float az, float bx, float by, float bz) {
    const float x = ax - bx, y = ay - by, z = az - bz;
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

__device__ __forceinline__ void face_cot(const float* __restrict__ verts, int64_t i0, int64_t i1, int64_t i2,
                                         float& cota, float& cotb, float& cotc) {
    const float x0 = verts[3 * i0], y0 = verts[3 * i0 + 1], z0 = verts[3 * i0 + 2];
    const float x1 = verts[3 * i1], y1 = verts[3 * i1 + 1], z1 = verts[3 * i1 + 2];
    const float x2 = verts[3 * i2], y2 = verts[3 * i2 + 1], z2 = verts[3 * i2 + 2];
    const float A = edge_norm(x1, y1, z1, x2, y2, z2);   // opposite v0
    const float B = edge_norm(x0, y0, z0, x2, y2, z2);   // opposite v1
    const float C = edge_norm(x0, y0, z0, x1, y1, z1);   // opposite v2
    const float s = 0.5f * ((A + B) + C);
    float prod = ((s * (s - A)) * (s - B)) * (s - C);
    prod = prod < 1e-12f ? 1e-12f : prod;                // clamp_(min=1e-12); NaN passes through like torch
    const float area = sqrtf(prod);
    const float A2 = A * A, B2 = B * B, C2 = C * C;
    cota = (((B2 + C2) - A2) / area) / 4.0f;
    cotb = (((A2 + C2) - B2) / area) / 4.0f;
    cotc = (((A2 + B2) - C2) / area) / 4.0f;
}

// ------------------------------------------------------------------------------------------------
// 2. scatter half-edges into the slots of their rows
// ------------------------------------------------------------------------------------------------
template <typename IdxT, bool COT>
__global__ __launch_bounds__(BLOCK) void k_scatter(const IdxT* __restrict__ faces, int64_t F, int64_t V,
                                                   const float* __restrict__ verts, const int* __restrict__ slot_ptr,
                                                   const int* __restrict__ corner_off, int* __restrict__ slot_col,
                                                   float* __restrict__ slot_val) {
    for (int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x; f < F; f += (int64_t)gridDim.x * BLOCK) {
        const int64_t i0 = (int64_t)faces[3 * f + 0], i1 = (int64_t)faces[3 * f + 1], i2 = (int64_t)faces[3 * f + 2];
        if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= V || i1 >= V || i2 >= V) continue;
        float wa = 1.0f, wb = 1.0f, wc = 1.0f;
        if (COT) face_cot(verts, i0, i1, i2, wa, wb, wc);
        // geometry.py:43-50: cota -> (f1,f2), cotb -> (f2,f0), cotc -> (f0,f1), then symmetrised: every vertex of the face
        // receives the two half-edges towards the other two, at the offset k_count gave the corner inside the vertex' row; rows hold
        // an even number of slots, so the pair is 8-byte aligned and goes out as one int2 / float2 store.
        const int64_t row[3] = {i0, i1, i2};
        const int c0[3] = {(int)i2, (int)i2, (int)i1}, c1[3] = {(int)i1, (int)i0, (int)i0};
        const float w0[3] = {wb, wa, wa}, w1[3] = {wc, wc, wb};
        int slot[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) slot[e] = slot_ptr[row[e]] + corner_off[3 * f + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            *reinterpret_cast<int2*>(slot_col + slot[e]) = make_int2(c0[e], c1[e]);
            if (COT) *reinterpret_cast<float2*>(slot_val + slot[e]) = make_float2(w0[e], w1[e]);     // (the uniform Laplacian carries no weights)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. per row: sort slots by (col, weight), merge duplicates -> the row's final entries (diagonal included, in column order), written
//    tile by tile into a compact array; 4. a streaming copy of the tiles to their place in the CSR / COO arrays.
//    Round 5 (was: insertion sort in LDS, sorted slots written back, a 1M-element scan, an emit kernel that walked the slot rows again --
//    every half-edge crossed the memory system four times, 870 MB for 220 MB of output at 1M vertices): a row of up to 16 slots is
//    sorted IN REGISTERS by a 60-comparator network, merged there, and only its final entries leave the kernel; the scan is over the
//    3907 tile totals, not the rows; the emit is a coalesced copy.
// ------------------------------------------------------------------------------------------------
// One row: c / w point at its n slots (global memory or an LDS copy). Returns the number of distinct off-diagonal
// columns (compacted to the front of the slots) and the diagonal value.
template <bool COT>
__device__ __forceinline__ int merge_row(int* c, float* w, int n, int i, float a, float b, float& d_out) {
    // insertion sort (rows are short: ~2 x valence); ties on col are ordered by weight so that the
    // summation order, hence the fp32 result, does not depend on the atomics' arrival order
    for (int k = 1; k < n; ++k) {
        const int ck = c[k];
        const float wk = COT ? w[k] : 0.0f;        // uniform: the value slots are not even written (k_scatter)
        int j = k - 1;
        while (j >= 0 && (c[j] > ck || (COT && c[j] == ck && w[j] > wk))) { c[j + 1] = c[j]; if (COT) w[j + 1] = w[j]; --j; }
        c[j + 1] = ck;
        if (COT) w[j + 1] = wk;
    }
    int nu = 0;          // distinct off-diagonal columns written so far
    int ndistinct = 0;   // distinct columns including a self loop (uniform degree bookkeeping)
    bool self = false;
    float wsum = 0.0f;   // cot: sum of all slot weights of this row (= column sum of W, geometry.py:59)
    float selfacc = 0.0f;
    int k = 0;
    while (k < n) {
        const int ck = c[k];
        float acc = 0.0f;
        int k2 = k;
        while (k2 < n && c[k2] == ck) {
            if (COT) { wsum = wsum + w[k2]; acc = acc + b * (-w[k2]); }
            ++k2;
        }
        ++ndistinct;
        if (ck == i) {
            self = true;
            selfacc = acc;
        } else {
            c[nu] = ck;                       // nu <= k: never overwrites unread slots
            w[nu] = COT ? acc : b * -1.0f;
            ++nu;
        }
        k = k2;
    }
    if (COT) {
        float t = b * wsum;
        if (self) t = t + selfacc;
        d_out = a + t;
    } else {
        // L_ii = (#distinct adjacency entries, a self loop included) - (1 if self loop)   [geometry.py:86-94]
        const float lii = (float)(ndistinct - (self ? 1 : 0));
        d_out = a + b * lii;
    }
    return nu;
}


// 16 (col, weight) pairs in registers, ascending by (col, weight): the 60-comparator, 10-layer network (verified over all 2^16 0/1 inputs
// by tools/check_sort16.py). The uniform Laplacian carries no weights.
template <bool COT>
__device__ __forceinline__ void sort16(int (&c)[16], float (&w)[16]) {
#define LS_CE(i, j)                                                                                                   \
    {                                                                                                                 \
        const bool sw = c[i] > c[j] || (COT && c[i] == c[j] && w[i] > w[j]);                                          \
        const int ci = sw ? c[j] : c[i], cj = sw ? c[i] : c[j];                                                       \
        c[i] = ci; c[j] = cj;                                                                                         \
        if (COT) { const float wi = sw ? w[j] : w[i], wj = sw ? w[i] : w[j]; w[i] = wi; w[j] = wj; }                  \
    }
    LS_CE(0, 13) LS_CE(1, 12) LS_CE(2, 15) LS_CE(3, 14) LS_CE(4, 8) LS_CE(5, 6) LS_CE(7, 11) LS_CE(9, 10)
    LS_CE(0, 5) LS_CE(1, 7) LS_CE(2, 9) LS_CE(3, 4) LS_CE(6, 13) LS_CE(8, 14) LS_CE(10, 15) LS_CE(11, 12)
    LS_CE(0, 1) LS_CE(2, 3) LS_CE(4, 5) LS_CE(6, 8) LS_CE(7, 9) LS_CE(10, 11) LS_CE(12, 13) LS_CE(14, 15)
    LS_CE(0, 2) LS_CE(1, 3) LS_CE(4, 10) LS_CE(5, 11) LS_CE(6, 7) LS_CE(8, 9) LS_CE(12, 14) LS_CE(13, 15)
    LS_CE(1, 2) LS_CE(3, 12) LS_CE(4, 6) LS_CE(5, 7) LS_CE(8, 10) LS_CE(9, 11) LS_CE(13, 14)
    LS_CE(1, 4) LS_CE(2, 6) LS_CE(5, 8) LS_CE(7, 10) LS_CE(9, 13) LS_CE(11, 14)
    LS_CE(2, 4) LS_CE(3, 6) LS_CE(9, 12) LS_CE(11, 13)
    LS_CE(3, 5) LS_CE(6, 8) LS_CE(7, 9) LS_CE(10, 12)
    LS_CE(3, 4) LS_CE(5, 6) LS_CE(7, 8) LS_CE(9, 10) LS_CE(11, 12)
    LS_CE(6, 7) LS_CE(8, 9)
#undef LS_CE
}

// One pass over a sorted register row (the summation order of merge_row: groups in column order, weights ascending inside a group).
// emit(col, value) is called once per distinct off-diagonal column, in column order. Returns what merge_row returns.
template <bool COT, typename Emit>
__device__ __forceinline__ void walk16(const int (&c)[16], const float (&w)[16], int n, int i, float a, float b, int& n_off, int& n_lt,
                                       float& d_out, Emit emit) {
    int ndistinct = 0;
    bool self = false;
    float wsum = 0.0f, selfacc = 0.0f, acc = 0.0f;
    n_off = 0; n_lt = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < n) {
            if (COT) { wsum = wsum + w[k]; acc = acc + b * (-w[k]); }
            const bool tail = k == 15 || k + 1 >= n || c[k < 15 ? k + 1 : 15] != c[k];
            if (tail) {
                ++ndistinct;
                if (c[k] == i) { self = true; selfacc = acc; }
                else { emit(c[k], COT ? acc : b * -1.0f, n_off); ++n_off; n_lt += c[k] < i ? 1 : 0; }
                acc = 0.0f;
            }
        }
    }
    if (COT) {
        float t = b * wsum;
        if (self) t = t + selfacc;
        d_out = a + t;
    } else {
        const float lii = (float)(ndistinct - (self ? 1 : 0));        // geometry.py:86-94 (see merge_row)
        d_out = a + b * lii;
    }
}

constexpr int OUT_CAP = 3328;   // final entries of a tile staged in LDS (256 rows x 13): 26 KiB; larger tiles store directly
typedef int i4u_asm __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4u_asm __attribute__((ext_vector_type(4), aligned(4)));

template <bool COT>
__global__ __launch_bounds__(BLOCK) void k_row_merge(int64_t V, const int* __restrict__ slot_ptr, int* __restrict__ slot_col,
                                                     float* __restrict__ slot_val, float a, float b, int2* __restrict__ comp,
                                                     int* __restrict__ row_off, int* __restrict__ tile_cnt, float* __restrict__ diag) {
    __shared__ __attribute__((aligned(16))) int2 s_out[OUT_CAP];
    __shared__ int s_scan[BLOCK / WAVE + 1];
    const int64_t t0 = (int64_t)blockIdx.x * TILE_ROWS;
    const int64_t t1 = min(V, t0 + TILE_ROWS);
    const int64_t i = t0 + threadIdx.x;
    const bool active = i < t1;
    const int s0 = active ? slot_ptr[i] : 0, n = active ? slot_ptr[i + 1] - s0 : 0;
    const bool big = n > 16;                      // valence > 8: the row is sorted and merged in place in global memory (merge_row)
    int c[16];
    float w[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // the row's slots: up to four 16-byte requests per array (slot ranges are 8-byte aligned)
        i4u_asm cq = {0, 0, 0, 0};
        f4u_asm wq = {0.f, 0.f, 0.f, 0.f};
        if (!big && 4 * q < n) {
            cq = *reinterpret_cast<const i4u_asm*>(slot_col + s0 + 4 * q);
            if (COT) wq = *reinterpret_cast<const f4u_asm*>(slot_val + s0 + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { c[4 * q + e] = (!big && 4 * q + e < n) ? cq[e] : 0x7fffffff; w[4 * q + e] = (COT && !big && 4 * q + e < n) ? wq[e] : 0.0f; }
    }
    int n_off = 0, n_lt = 0;
    float d = a;
    if (big) {
        n_off = merge_row<COT>(slot_col + s0, slot_val + s0, n, (int)i, a, b, d);
        for (int k = 0; k < n_off; ++k) n_lt += slot_col[s0 + k] < (int)i ? 1 : 0;
    } else if (active) {
        sort16<COT>(c, w);
        walk16<COT>(c, w, n, (int)i, a, b, n_off, n_lt, d, [](int, float, int) {});
    }
    int total;
    const int o = block_exclusive_scan(active ? n_off + 1 : 0, &total, s_scan);
    if (active) { row_off[i] = o; diag[i] = d; }
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
    const size_t cbase = (size_t)slot_ptr[t0] + (size_t)t0;
    const bool staged = total <= OUT_CAP;         // uniform per workgroup
    int2* dst = staged ? s_out + o : comp + cbase + o;
    if (active) {
        dst[n_lt] = make_int2((int)i, __float_as_int(d));                  // the diagonal sits behind the n_lt smaller columns
        if (big) {
            for (int k = 0; k < n_off; ++k) dst[k + (k >= n_lt ? 1 : 0)] = make_int2(slot_col[s0 + k], __float_as_int(slot_val[s0 + k]));
        } else {
            int n2, l2;
            float d2;
            walk16<COT>(c, w, n, (int)i, a, b, n2, l2, d2, [&](int ck, float v, int p) { dst[p + (ck > (int)i ? 1 : 0)] = make_int2(ck, __float_as_int(v)); });
        }
    }
    if (!staged) return;
    __syncthreads();
    for (int t = threadIdx.x; t < total; t += BLOCK) comp[cbase + t] = s_out[t];
}

// where the tiles' rows start: exclusive scan of the tile totals (one workgroup), then rowptr[i] = tile_off[tile] + row_off[i]
__global__ __launch_bounds__(1024) void k_tile_scan(const int* __restrict__ tile_cnt, int tiles, int* __restrict__ tile_off) {   // <<<1, 1024>>>
    __shared__ int s_wave[17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (tiles + 1023) / 1024, lo = min(tiles, (int)threadIdx.x * per), hi = min(tiles, lo + per);
    int sum = 0;
    for (int t = lo; t < hi; ++t) sum += tile_cnt[t];
    const int inc = wave_inclusive_scan(sum);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int k = 0; k < 16; ++k) { const int v = s_wave[k]; s_wave[k] = run; run += v; } s_wave[16] = run; }
    __syncthreads();
    int run = s_wave[w] + inc - sum;
    for (int t = lo; t < hi; ++t) { tile_off[t] = run; run += tile_cnt[t]; }
    if (threadIdx.x == 0) tile_off[tiles] = s_wave[16];
}
__global__ __launch_bounds__(BLOCK) void k_rowptr(int64_t V, const int* __restrict__ tile_off, const int* __restrict__ row_off, int* __restrict__ rowptr,
                                                  int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * TILE_ROWS + threadIdx.x;
    if (i < V) rowptr[i] = tile_off[blockIdx.x] + row_off[i];
    if (i == V - 1) { rowptr[V] = tile_off[gridDim.x]; flags[1] = tile_off[gridDim.x]; }      // (flags[0]: index check, flags[1]: nnz -- one copy back)
}

// ------------------------------------------------------------------------------------------------
// 4. emit CSR (+ COO int64 indices, + 1/diag): a tile's compact block is copied to its place with fully coalesced loads and stores;
//    the COO row of an entry is found by a binary search over the tile's 256 row offsets in LDS
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_emit(int64_t V, const int* __restrict__ slot_ptr, const int2* __restrict__ comp,
                                                const int* __restrict__ row_off, const float* __restrict__ diag,
                                                const int* __restrict__ rowptr, int* __restrict__ col, float* __restrict__ val,
                                                int64_t* __restrict__ coo_row, int64_t* __restrict__ coo_col,
                                                float* __restrict__ dinv) {
    __shared__ int s_ro[TILE_ROWS + 1];
    const int64_t r0 = (int64_t)blockIdx.x * TILE_ROWS;
    const int64_t r1 = min(r0 + (int64_t)TILE_ROWS, V);
    const int rows = (int)(r1 - r0);
    const int base = rowptr[r0], total = rowptr[r1] - base;
    const size_t cbase = (size_t)slot_ptr[r0] + (size_t)r0;
    const int64_t i = r0 + threadIdx.x;
    if (i < r1) {
        s_ro[threadIdx.x] = row_off[i];
        if (dinv) dinv[i] = 1.0f / diag[i];
    }
    if (threadIdx.x == 0) s_ro[rows] = total;
    __syncthreads();
    for (int t = threadIdx.x; t < total; t += BLOCK) {
        const int2 e = comp[cbase + t];
        col[base + t] = e.x;                      // (non-temporal stores measured: 55 against 48 us)
        val[base + t] = __int_as_float(e.y);
        if (coo_row) {
            int lo = 0, hi = rows;                // largest r with s_ro[r] <= t
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_ro[mid] <= t) lo = mid; else hi = mid; }
            coo_row[base + t] = r0 + lo;
            coo_col[base + t] = e.x;
        }
    }
}

// foreign COO (coalesced) -> CSR
__global__ __launch_bounds__(BLOCK) void k_coo_hist(const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
                                                    const float* __restrict__ vals, int64_t nnz, int64_t V,
                                                    int* __restrict__ cnt, int* __restrict__ col, float* __restrict__ diagv,
                                                    int* __restrict__ flags) {
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * BLOCK) {
        const int64_t r = rows[t], c = cols[t];
        if (r < 0 || r >= V || c < 0 || c >= V) { flags[0] = 1; continue; }
        if (t > 0) {   // must be row-major sorted and duplicate free (torch coalesce order)
            const int64_t pr = rows[t - 1], pc = cols[t - 1];
            if (pr > r || (pr == r && pc >= c)) flags[1] = 1;
        }
        atomicAdd(&cnt[r], 1);
        col[t] = (int)c;
        if (r == c && diagv) diagv[r] = vals[t];
    }
}

__global__ __launch_bounds__(BLOCK) void k_invert(float* __restrict__ d, int64_t V, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= V) return;
    const float v = d[i];
    if (!(v > 0.0f)) flags[2] = 1;    // Jacobi needs a positive diagonal (SPD matrix)
    d[i] = 1.0f / v;
}

}  // namespace ls

using namespace ls;

extern "C" int ls_assemble_workspace_bytes(int64_t V, int64_t F, size_t* h_bytes) {
    LS_REQUIRE(h_bytes && V >= 0 && F >= 0, LS_E_INVALID, "ls_assemble_workspace_bytes: bad argument");
    LS_REQUIRE(V < (int64_t)2000000000 && 6 * F < (int64_t)2000000000, LS_E_OVERFLOW,
               "mesh too large for the int32 index space (V=%lld F=%lld)", (long long)V, (long long)F);
    *h_bytes = AsmLayout(V, F).total;
    return LS_OK;
}

extern "C" int ls_assemble_pattern(const void* faces, int idx_bytes, int64_t F, int64_t V, const float* verts, int kind,
                                   float a, float b, void* workspace, size_t workspace_bytes, int32_t* rowptr,
                                   int64_t* h_nnz, int device, void* stream) {
    LS_REQUIRE(V >= 0 && F >= 0 && (F == 0 || faces) && workspace && rowptr && h_nnz, LS_E_INVALID,
               "ls_assemble_pattern: null pointer or negative size");
    LS_REQUIRE(idx_bytes == 4 || idx_bytes == 8, LS_E_INVALID, "faces must be int32 or int64 (idx_bytes=%d)", idx_bytes);
    LS_REQUIRE(kind == LS_LAPLACIAN_UNIFORM || kind == LS_LAPLACIAN_COT, LS_E_INVALID, "unknown Laplacian kind %d", kind);
    LS_REQUIRE(kind == LS_LAPLACIAN_UNIFORM || verts || F == 0, LS_E_INVALID, "cotangent Laplacian needs vertex positions");
    LS_REQUIRE(V < (int64_t)2000000000 && 6 * F < (int64_t)2000000000, LS_E_OVERFLOW, "mesh too large for int32 indices");
    const AsmLayout L(V, F);
    LS_REQUIRE(workspace_bytes >= L.total, LS_E_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* cnt = (int*)(ws + L.cnt);
    int* slot_ptr = (int*)(ws + L.slot_ptr);
    int* corner_off = (int*)(ws + L.corner_off);
    int* row_off = (int*)(ws + L.row_off);
    unsigned long long* chain_rows = (unsigned long long*)(ws + L.chain_rows);
    int* tile_cnt = (int*)(ws + L.tile_cnt);
    int* tile_off = (int*)(ws + L.tile_off);
    int2* comp = (int2*)(ws + L.comp);
    float* diag = (float*)(ws + L.diag);
    int* flags = (int*)(ws + L.flags);
    int* slot_col = (int*)(ws + L.slot_col);
    float* slot_val = (float*)(ws + L.slot_val);

    // flags, the scan's chain and cnt are contiguous: one memset
    LS_HIP(hipMemsetAsync(ws, 0, L.slot_ptr, st));
    if (V == 0) { LS_HIP(hipMemsetAsync(rowptr, 0, 4, st)); *h_nnz = 0; LS_HIP(hipStreamSynchronize(st)); return LS_OK; }
    const int fgrid = F ? (int)std::min<int64_t>(div_up(F, BLOCK), 8192) : 1;
    if (F) {
        const int cgrid = div_up(F, BLOCK * CNT_FPT);
        if (idx_bytes == 4) hipLaunchKernelGGL(k_count<int32_t>, dim3(cgrid), dim3(BLOCK), 0, st, (const int32_t*)faces, F, V, cnt, corner_off, flags);
        else hipLaunchKernelGGL(k_count<int64_t>, dim3(cgrid), dim3(BLOCK), 0, st, (const int64_t*)faces, F, V, cnt, corner_off, flags);
    }
    int rc = exclusive_scan_chained(cnt, V, slot_ptr, chain_rows, st);
    if (rc) return rc;
    if (F) {
        const bool cot = kind == LS_LAPLACIAN_COT;
        if (idx_bytes == 4) {
            if (cot) hipLaunchKernelGGL((k_scatter<int32_t, true>), dim3(fgrid), dim3(BLOCK), 0, st, (const int32_t*)faces, F, V, verts, slot_ptr, (const int*)corner_off, slot_col, slot_val);
            else hipLaunchKernelGGL((k_scatter<int32_t, false>), dim3(fgrid), dim3(BLOCK), 0, st, (const int32_t*)faces, F, V, verts, slot_ptr, (const int*)corner_off, slot_col, slot_val);
        } else {
            if (cot) hipLaunchKernelGGL((k_scatter<int64_t, true>), dim3(fgrid), dim3(BLOCK), 0, st, (const int64_t*)faces, F, V, verts, slot_ptr, (const int*)corner_off, slot_col, slot_val);
            else hipLaunchKernelGGL((k_scatter<int64_t, false>), dim3(fgrid), dim3(BLOCK), 0, st, (const int64_t*)faces, F, V, verts, slot_ptr, (const int*)corner_off, slot_col, slot_val);
        }
    }
    const int vgrid = div_up(V, TILE_ROWS);
    if (kind == LS_LAPLACIAN_COT) hipLaunchKernelGGL(k_row_merge<true>, dim3(vgrid), dim3(BLOCK), 0, st, V, slot_ptr, slot_col, slot_val, a, b, comp, row_off, tile_cnt, diag);
    else hipLaunchKernelGGL(k_row_merge<false>, dim3(vgrid), dim3(BLOCK), 0, st, V, slot_ptr, slot_col, slot_val, a, b, comp, row_off, tile_cnt, diag);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, (const int*)tile_cnt, vgrid, tile_off);
    hipLaunchKernelGGL(k_rowptr, dim3(vgrid), dim3(BLOCK), 0, st, V, (const int*)tile_off, (const int*)row_off, rowptr, flags);
    LS_HIP(hipGetLastError());
    int h[2] = {0, 0};
    LS_HIP(hipMemcpyAsync(h, flags, 8, hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    LS_REQUIRE(h[0] == 0, LS_E_INDEX, "a face index is outside [0, %lld)", (long long)V);
    *h_nnz = h[1];
    return LS_OK;
}

extern "C" int ls_assemble_fill(const void* workspace, size_t workspace_bytes, int64_t V, int64_t F, const int32_t* rowptr,
                                int32_t* col, float* val, int64_t* coo_idx, int64_t nnz, float* dinv, int device,
                                void* stream) {
    LS_REQUIRE(workspace && rowptr && V >= 0 && F >= 0 && nnz >= 0 && (nnz == 0 || (col && val)), LS_E_INVALID,
               "ls_assemble_fill: null pointer or negative size");
    const AsmLayout L(V, F);
    LS_REQUIRE(workspace_bytes >= L.total, LS_E_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, L.total);
    if (V == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    const char* ws = (const char*)workspace;
    hipLaunchKernelGGL(k_emit, dim3(div_up(V, TILE_ROWS)), dim3(BLOCK), 0, (hipStream_t)stream, V,
                       (const int*)(ws + L.slot_ptr), (const int2*)(ws + L.comp), (const int*)(ws + L.row_off),
                       (const float*)(ws + L.diag), rowptr, col, val, coo_idx, coo_idx ? coo_idx + nnz : nullptr, dinv);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_csr_from_coo(const int64_t* coo_rows, const int64_t* coo_cols, const float* vals, int64_t nnz, int64_t V,
                               int32_t* rowptr, int32_t* col, float* dinv, void* scratch, size_t scratch_bytes, int device,
                               void* stream) {
    LS_REQUIRE(V >= 0 && nnz >= 0 && rowptr && scratch && (nnz == 0 || (coo_rows && coo_cols && col)), LS_E_INVALID,
               "ls_csr_from_coo: null pointer or negative size");
    LS_REQUIRE(V < (int64_t)2000000000 && nnz < (int64_t)2000000000, LS_E_OVERFLOW, "matrix too large for int32 indices");
    const size_t need = 4 * (size_t)(V + 1) + 4 * (size_t)((V + 1) / 2048 + 4) + 64 + 512;
    LS_REQUIRE(scratch_bytes >= need, LS_E_WORKSPACE, "scratch too small: %zu < %zu", scratch_bytes, need);
    LS_REQUIRE(!dinv || vals, LS_E_INVALID, "dinv requested without values");
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    char* ws = (char*)scratch;
    int* cnt = (int*)ws;
    int* flags = (int*)(ws + al(4 * (size_t)(V + 1)));
    int* bsum = flags + 16;
    LS_HIP(hipMemsetAsync(ws, 0, al(4 * (size_t)(V + 1)) + 64, st));
    if (dinv && V) LS_HIP(hipMemsetAsync(dinv, 0, 4 * (size_t)V, st));
    if (nnz) hipLaunchKernelGGL(k_coo_hist, dim3((int)std::min<int64_t>(div_up(nnz, BLOCK), 8192)), dim3(BLOCK), 0, st,
                                coo_rows, coo_cols, vals, nnz, V, cnt, col, dinv, flags);
    if (V == 0) { LS_HIP(hipMemsetAsync(rowptr, 0, 4, st)); return LS_OK; }
    int rc = exclusive_scan(cnt, V, rowptr, bsum, st);
    if (rc) return rc;
    if (dinv) hipLaunchKernelGGL(k_invert, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, dinv, V, flags);
    LS_HIP(hipGetLastError());
    int h[3];
    LS_HIP(hipMemcpyAsync(h, flags, 12, hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    LS_REQUIRE(h[0] == 0, LS_E_INDEX, "COO index outside [0, %lld)", (long long)V);
    LS_REQUIRE(h[1] == 0, LS_E_INVALID, "COO matrix is not coalesced (row-major sorted, unique): call .coalesce() first");
    LS_REQUIRE(!dinv || h[2] == 0, LS_E_INVALID, "matrix has a missing or non-positive diagonal entry: not SPD");
    return LS_OK;
}

// ------------------------------------------------------------------------------------------------
// remove_duplicates (reference: scripts/geometry.py:3-11): unique vertex rows in lexicographic order of their VALUES
// (what torch.unique(v, dim=0) returns), the inverse map, and the faces re-indexed through it. Done once per remesh.
// Stable LSD radix sort of the row ids over the 12 key bytes (z low byte first, x high byte last), hand-written:
// per pass a histogram kernel, the scan above, and a scatter kernel in which ONE wave walks its chunk in order and ranks
// equal digits inside every 64-element tile with ballots (stable by construction, no atomics in the scatter).
// ------------------------------------------------------------------------------------------------
namespace ls {

__global__ __launch_bounds__(256) void k_dedup_flags(const float* __restrict__ verts, const int* __restrict__ order, int64_t n, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int f = 1;
    if (i) {
        const int a = order[i], b = order[i - 1];
        f = key_of(verts[3 * (size_t)a]) != key_of(verts[3 * (size_t)b]) || key_of(verts[3 * (size_t)a + 1]) != key_of(verts[3 * (size_t)b + 1]) ||
            key_of(verts[3 * (size_t)a + 2]) != key_of(verts[3 * (size_t)b + 2]);
    }
    flag[i] = f;
}

// uid[i] = exclusive scan of flag -> unique id of sorted position i is uid[i + 1] - 1
__global__ __launch_bounds__(256) void k_dedup_emit(const float* __restrict__ verts, const int* __restrict__ order, int64_t n, const int* __restrict__ flag,
                                                    const int* __restrict__ uid, float* __restrict__ unique_verts, int64_t* __restrict__ inverse) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int row = order[i], u = uid[i + 1] - 1;
    inverse[row] = u;
    if (flag[i]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) unique_verts[3 * (size_t)u + c] = verts[3 * (size_t)row + c];
    }
}

template <typename IdxT>
__global__ __launch_bounds__(256) void k_dedup_faces(const IdxT* __restrict__ faces, int64_t m, int64_t V, const int64_t* __restrict__ inverse,
                                                     int64_t* __restrict__ out, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t v = (int64_t)faces[i];
    if (v < 0 || v >= V) { *bad = 1; out[i] = 0; return; }
    out[i] = inverse[v];
}

}  // namespace ls

extern "C" int ls_remove_duplicates_workspace_bytes(int64_t V, size_t* h_bytes) {
    LS_REQUIRE(h_bytes && V >= 0, LS_E_INVALID, "ls_remove_duplicates_workspace_bytes: bad argument");
    const size_t nb = (size_t)div_up(std::max<int64_t>(V, 1), rs_chunk(V));
    // order a / b, flags, uid (V + 1), carried keys a / b, histogram + its scan (256 nb + 1 each), scan block sums
    *h_bytes = sizeof(int) * ((size_t)V * 6 + 16 + 2 * (256 * nb + 16) + (size_t)div_up(std::max<int64_t>(std::max<int64_t>(V, 256 * (int64_t)nb), 1), SCAN_CHUNK) + 64) + 256;
    return LS_OK;
}

extern "C" int ls_remove_duplicates(const float* verts, int64_t V, const void* faces, int idx_bytes, int64_t F, float* unique_verts,
                                    int64_t* inverse, int64_t* new_faces, int64_t* h_n_unique, void* workspace, size_t ws_bytes, int device,
                                    void* stream) {
    LS_REQUIRE(h_n_unique && V >= 0 && F >= 0 && V < INT32_MAX && (V == 0 || (verts && unique_verts && inverse)) &&
               (F == 0 || (faces && new_faces && (idx_bytes == 4 || idx_bytes == 8))), LS_E_INVALID, "ls_remove_duplicates: bad argument");
    size_t need = 0;
    ls_remove_duplicates_workspace_bytes(V, &need);
    LS_REQUIRE(workspace && ws_bytes >= need, LS_E_WORKSPACE, "ls_remove_duplicates: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    *h_n_unique = 0;
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    if (V == 0) { LS_REQUIRE(F == 0, LS_E_INDEX, "ls_remove_duplicates: faces without vertices"); return LS_OK; }
    const int nb = div_up(V, rs_chunk(V));
    int* w = (int*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int* ord_a = w;
    int* ord_b = ord_a + V;
    int* flag = ord_b + V;
    int* uid = flag + V;                       // V + 1
    unsigned* keys_a = (unsigned*)(uid + V + 16);
    unsigned* keys_b = keys_a + V;
    int* hist = (int*)(keys_b + V);            // 256 nb
    int* offs = hist + 256 * (size_t)nb + 16;  // 256 nb + 1
    int* bsum = offs + 256 * (size_t)nb + 16;
    const int* src = nullptr;
    {
        KeyVerts key{verts};                   // three 32-bit key words (z, y, x), each gathered once and carried through its four byte passes
        const int rc0 = radix_argsort_words(key, V, 3, ord_a, ord_b, keys_a, keys_b, hist, offs, bsum, st, &src);
        if (rc0) return rc0;
    }
    const int vg = div_up(V, 256);
    hipLaunchKernelGGL(k_dedup_flags, dim3(vg), dim3(256), 0, st, verts, src, V, flag);
    int rc = exclusive_scan(flag, V, uid, bsum, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_dedup_emit, dim3(vg), dim3(256), 0, st, verts, src, V, (const int*)flag, (const int*)uid, unique_verts, inverse);
    int* bad = flag;                           // flags are consumed: reuse one word as the range-check flag
    if (F) {
        LS_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
        const int64_t m = 3 * F;
        if (idx_bytes == 4) hipLaunchKernelGGL(k_dedup_faces<int32_t>, dim3(div_up(m, 256)), dim3(256), 0, st, (const int32_t*)faces, m, V, (const int64_t*)inverse, new_faces, bad);
        else hipLaunchKernelGGL(k_dedup_faces<int64_t>, dim3(div_up(m, 256)), dim3(256), 0, st, (const int64_t*)faces, m, V, (const int64_t*)inverse, new_faces, bad);
    }
    LS_HIP(hipGetLastError());
    int h[2] = {0, 0};
    LS_HIP(hipMemcpyAsync(&h[0], uid + V, sizeof(int), hipMemcpyDeviceToHost, st));
    if (F) LS_HIP(hipMemcpyAsync(&h[1], bad, sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    LS_REQUIRE(h[1] == 0, LS_E_INDEX, "a face index is outside [0, %lld)", (long long)V);
    *h_n_unique = h[0];
    return LS_OK;
}

// ------------------------------------------------------------------------------------------------
// Two more users of the radix sort (both replace stock-torch sorts that sat next to the product path):
//   ls_csr_transpose   CSR of M^T (for the backward pass of to_differential on an unsymmetric foreign matrix)
//   ls_corner_ranks    vertex-major ranking of the 3 F face corners (normals: corners of a vertex summed in ascending corner id)
// ------------------------------------------------------------------------------------------------
namespace ls {
__global__ __launch_bounds__(256) void k_count_keys(const int* __restrict__ keys, int64_t n, int64_t nkeys, int* __restrict__ cnt, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = keys[i];
    if (k < 0 || k >= nkeys) { *bad = 1; return; }
    atomicAdd(&cnt[k], 1);
}
__global__ __launch_bounds__(256) void k_transpose_emit(const int* __restrict__ order, int64_t nnz, const int* __restrict__ rowptr, int64_t V,
                                                        const float* __restrict__ val, int* __restrict__ t_col, float* __restrict__ t_val) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const int e = order[i];
    int64_t lo = 0, hi = V;                     // row of entry e
    while (lo + 1 < hi) { const int64_t mid = (lo + hi) >> 1; if (rowptr[mid] <= e) lo = mid; else hi = mid; }
    t_col[i] = (int)lo;
    t_val[i] = val[e];
}
template <typename IdxT>
__global__ __launch_bounds__(256) void k_faces_to_i32(const IdxT* __restrict__ f, int64_t m, int64_t V, int* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // the range check sees the ORIGINAL value: 2^32 + 3 must not wrap into a valid vertex (-1 fails k_count_keys' check)
    if (i < m) { const IdxT x = f[i]; out[i] = (x < 0 || (int64_t)x >= V) ? -1 : (int)x; }
}
__global__ __launch_bounds__(256) void k_invert_order(const int* __restrict__ order, int64_t n, int* __restrict__ rank) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rank[order[i]] = (int)i;
}
}  // namespace ls

static size_t argsort_ws_ints(int64_t n, int64_t nkeys) {
    const size_t nb = (size_t)div_up(std::max<int64_t>(n, 1), rs_chunk(n));
    return (size_t)n * 3 + (size_t)nkeys + 64 + 2 * (256 * nb + 16) + (size_t)div_up(std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(n, nkeys), 256 * (int64_t)nb), 1), SCAN_CHUNK) + 64;
}

extern "C" int ls_csr_transpose_workspace_bytes(int64_t V, int64_t nnz, size_t* h_bytes) {
    LS_REQUIRE(h_bytes && V >= 0 && nnz >= 0, LS_E_INVALID, "ls_csr_transpose_workspace_bytes: bad argument");
    *h_bytes = sizeof(int) * argsort_ws_ints(nnz, V + 1) + 256;
    return LS_OK;
}

extern "C" int ls_csr_transpose(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V, int64_t nnz, int32_t* t_rowptr,
                                int32_t* t_col, float* t_val, void* workspace, size_t ws_bytes, int device, void* stream) {
    LS_REQUIRE(rowptr && t_rowptr && V > 0 && nnz >= 0 && nnz < INT32_MAX && (nnz == 0 || (col && val && t_col && t_val)), LS_E_INVALID, "ls_csr_transpose: bad argument");
    size_t need = 0;
    ls_csr_transpose_workspace_bytes(V, nnz, &need);
    LS_REQUIRE(workspace && ws_bytes >= need, LS_E_WORKSPACE, "ls_csr_transpose: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = (size_t)div_up(std::max<int64_t>(nnz, 1), rs_chunk(nnz));
    int* w = (int*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int *ord_a = w, *ord_b = ord_a + nnz, *cnt = ord_b + nnz;          // cnt: V + 1 (last = range flag)
    int *hist = cnt + V + 16 + nnz, *offs = hist + 256 * nb + 16, *bsum = offs + 256 * nb + 16;
    LS_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (V + 1), st));
    if (nnz) hipLaunchKernelGGL(k_count_keys, dim3(div_up(nnz, 256)), dim3(256), 0, st, (const int*)col, nnz, V, cnt, cnt + V);
    int rc = exclusive_scan(cnt, V, t_rowptr, bsum, st);
    if (rc) return rc;
    if (nnz) {
        const int* src = nullptr;
        int passes = 1;
        while (passes < 4 && (V - 1) >> (8 * passes)) ++passes;
        KeyInt key{(const int*)col};
        rc = radix_argsort(key, nnz, passes, ord_a, ord_b, hist, offs, bsum, st, &src);
        if (rc) return rc;
        hipLaunchKernelGGL(k_transpose_emit, dim3(div_up(nnz, 256)), dim3(256), 0, st, src, nnz, (const int*)rowptr, V, val, (int*)t_col, t_val);
    }
    int bad = 0;
    LS_HIP(hipMemcpyAsync(&bad, cnt + V, sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    LS_HIP(hipGetLastError());
    LS_REQUIRE(!bad, LS_E_INDEX, "ls_csr_transpose: a column index is outside [0, %lld)", (long long)V);
    return LS_OK;
}

extern "C" int ls_corner_ranks_workspace_bytes(int64_t F, int64_t V, size_t* h_bytes) {
    LS_REQUIRE(h_bytes && F >= 0 && V >= 0, LS_E_INVALID, "ls_corner_ranks_workspace_bytes: bad argument");
    *h_bytes = sizeof(int) * (argsort_ws_ints(3 * F, V + 1) + (size_t)3 * F) + 256;
    return LS_OK;
}

extern "C" int ls_corner_ranks(const void* faces, int idx_bytes, int64_t F, int64_t V, int32_t* vptr, int32_t* cpos, void* workspace,
                               size_t ws_bytes, int device, void* stream) {
    LS_REQUIRE(vptr && V >= 0 && F >= 0 && 3 * F < INT32_MAX && (F == 0 || (faces && cpos && (idx_bytes == 4 || idx_bytes == 8))), LS_E_INVALID,
               "ls_corner_ranks: bad argument");
    size_t need = 0;
    ls_corner_ranks_workspace_bytes(F, V, &need);
    LS_REQUIRE(workspace && ws_bytes >= need, LS_E_WORKSPACE, "ls_corner_ranks: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = 3 * F;
    const size_t nb = (size_t)div_up(std::max<int64_t>(n, 1), rs_chunk(n));
    int* w = (int*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int *ord_a = w, *ord_b = ord_a + n, *cnt = ord_b + n;
    int *hist = cnt + V + 16 + n, *offs = hist + 256 * nb + 16, *bsum = offs + 256 * nb + 16;
    int* keys = bsum + div_up(std::max<int64_t>(std::max<int64_t>(std::max<int64_t>(n, V + 1), 256 * (int64_t)nb), 1), SCAN_CHUNK) + 64;
    LS_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (V + 1), st));
    if (n) {
        if (idx_bytes == 4) hipLaunchKernelGGL(k_faces_to_i32<int32_t>, dim3(div_up(n, 256)), dim3(256), 0, st, (const int32_t*)faces, n, V, keys);
        else hipLaunchKernelGGL(k_faces_to_i32<int64_t>, dim3(div_up(n, 256)), dim3(256), 0, st, (const int64_t*)faces, n, V, keys);
        hipLaunchKernelGGL(k_count_keys, dim3(div_up(n, 256)), dim3(256), 0, st, (const int*)keys, n, V, cnt, cnt + V);
    }
    int rc = exclusive_scan(cnt, V, (int*)vptr, bsum, st);
    if (rc) return rc;
    int bad = 0;
    LS_HIP(hipMemcpyAsync(&bad, cnt + V, sizeof(int), hipMemcpyDeviceToHost, st));
    LS_HIP(hipStreamSynchronize(st));
    LS_REQUIRE(!bad, LS_E_INDEX, "face index out of range for %lld vertices", (long long)V);
    if (n) {
        const int* src = nullptr;
        int passes = 1;
        while (passes < 4 && (std::max<int64_t>(V, 1) - 1) >> (8 * passes)) ++passes;
        KeyInt key{(const int*)keys};
        rc = radix_argsort(key, n, passes, ord_a, ord_b, hist, offs, bsum, st, &src);
        if (rc) return rc;
        hipLaunchKernelGGL(k_invert_order, dim3(div_up(n, 256)), dim3(256), 0, st, src, n, (int*)cpos);
    }
    LS_HIP(hipGetLastError());
    return LS_OK;
}
