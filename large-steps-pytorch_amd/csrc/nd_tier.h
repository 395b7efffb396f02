// nd_tier.h -- the bottom tier of the nested-dissection re-solve: ONE launch per sweep for the deepest H tree levels.
//
// A dependent kernel boundary is the cheapest device-wide synchronisation on this chip (~1.5-1.9 us), but every tree
// level that is its own launch also pays a chain of dependent memory round trips before its factor stream starts
// (tile record -> index lists -> right-hand side -> LDS), ~8-10 us per level at 1M vertices however few bytes the
// level holds. The deepest levels are where the nodes are small and many: a whole subtree of H levels (1 + A + A^2
// nodes for H = 3) is a few hundred KB of factor, so ONE workgroup walks it level by level with __syncthreads() between
// the levels -- no launch, no device-wide hand-off; the vectors it hands from level to level (child updates -> slots,
// parent x -> boundary vectors) stay in this CU's L1/L2. Waves are the unit of work inside the workgroup:
//     item      = (node, 64-row chunk, part of the reduction range); a wave runs its items of a level one after another;
//                 a node that is large for a wave (the tier's root) is cut into parts whose partial sums meet in LDS
//     leaves    (no children; >= 75 % of all vertices) are stored in their own format, see "sparse leaves" below
//
// Sparse leaves. A leaf front has no child contributions: F_ss = A_ss, F_bs = A_bs are blocks of the matrix itself, and
// A_bs is sparse (2-3 entries per boundary row). Instead of the dense W = A_bs F_ss^-1 (read in both sweeps) and the full
// symmetric inverse, a leaf keeps ONE triangle of Finv = A_ss^-1 (packed rows, s (s + 1) / 2 numbers) and the sparse
// block in two row orders:
//     up    y = Finv b_s (kept for the down sweep in place of b');  upd = A_bs y
//     down  x_s = y - Finv (A_sb x_bnd)
// 2 x s(s+1)/2 + O(b) numbers per leaf and solve instead of s^2 + 2 s b: 325 -> ~160 MB at 1M vertices. The triangle is
// copied into LDS with 16-byte loads (contiguous, fully coalesced) and read row-per-lane from there; the packed index
// T(j) + c, T(j) = j (j + 1) / 2, is conflict free over the 32 lanes of an LDS access group because j -> T(j) mod 32 is a
// permutation of 0..31 (the triangular-probing property), as is the transposed access T(c) + j (consecutive).
// A wave runs its leaves one after the other; only the small loads of the NEXT leaf (its record and the index lists that
// depend on it) are requested ahead -- a full software pipeline was measured slower (see leaf_phase), and so was a one-dword-per-line
// touch of the next triangle into the L2 (round 3: 227.7 against 224.0 us at 1M, 759 / 755 at 4M).
#pragma once

namespace ls {

constexpr int TIER_MAX_H = 6;            // tree levels one workgroup may walk
constexpr int TIER_WAVES = 4;            // waves per tier workgroup: 4 (four workgroups per CU), or TIER_WAVES_WIDE for trees with few subtrees
constexpr int TIER_WAVES_WIDE = 8;       // (<= 768 workgroups -- the arity-8 trees of 105k-300k vertices have 512: with 4 waves each that is half a
                                         //  CU's waves; 8 waves per workgroup, two workgroups per CU: -5 %)
constexpr int TIER_WAVES_FULL = 16;      // one workgroup per CU on a subtree one level taller (direct_tier_full16, direct.hip: from 800k vertices)
constexpr int TIER_TRI4 = 9;             // 16-byte loads per lane that hold a leaf triangle (s <= 64: 2080 floats = 520 float4)
constexpr int TIER_SPE = 4;              // sparse entries per row prefetched to registers (longer rows: loop)

enum : int { NODE_LEAF = 1, NODE_SPARSE = 2, NODE_QUAD = 4, NODE_UPC = 8 };   // UPC: the parent is in the tier too (compact update hand-off)

struct SpEnt { float val; int idx; };

struct NodeD {             // host-side node record the planner cuts into items
    int s, b, own_start, bnd_off, front_off, pfront_off, cix, flags;
    long long finv_off, w_off;
    int spb_off, sps_off;
};

// everything a wave needs for one item, fetched with scalar loads (the item index is wave-uniform)
struct alignas(32) TierItem {
    int s, b, own_start, bnd_off, front_off, pfront_off, cix, flags;
    long long finv_off, w_off;           // sparse leaf: finv_off = offset of the packed triangle in `tri`
    int spb_off, sps_off;                // sparse leaf: offsets of the row pointers (boundary rows / own rows) in sp_ptr
    int row0, r0, r1, part, nparts, pad0, pad1, pad2;
};
static_assert(sizeof(TierItem) == 96, "TierItem layout");

struct alignas(64) TierWG {
    int up_off[TIER_MAX_H + 1];          // item ranges per phase of the up sweep (deepest level first) into TierItem[]
    int down_off[TIER_MAX_H + 1];        // ... of the down sweep (tier root first)
    unsigned up_split, down_split;       // bit p: phase p has items cut into parts (needs the combine step)
    unsigned up_leaf, down_leaf;         // bit p: phase p consists of sparse leaves only (pipelined leaf loop)
    int n_dense, pad;                    // own rows of the subtree's dense nodes: TIER_MAX_H ranges (start, count) in the tree's numbering
    int dense_rng[2 * TIER_MAX_H];
};
static_assert(sizeof(TierWG) == 128, "TierWG layout (read as 32 dwords)");

struct TierArgs {
    const TierItem* items;
    const TierWG* wgs;
    const int* perm;
    const unsigned char* mask;
    const int* ppos;
    const int* pull;                                 // up sweep inside the tier: (front position, child) -> index of the child's boundary entry, -1 = none
    const int* push_ptr;
    const int* push_tgt;
    const float* u4;                                 // dense tier nodes, up sweep stream (quad-interleaved, see tier_dot)
    const float* d4;                                 // dense tier nodes, down sweep stream
    const float* tri;
    const int* sp_ptr;
    const SpEnt* sp_ent;
    float* bprime;
    int upper_lo, upper_hi;                          // rows of the levels above the tier (tree numbering): gathered into braw as well
    float* braw;                                     // b of the tier's inner-node rows in the tree's numbering (up sweep scratch)
    float* slots;
    float* xb;
    int arity, phases, region_floats, vec_floats;    // LDS per wave: [vec_floats | partial sums: rounds x 64 x 4]
    int xcd_order;                                   // 1: workgroup -> subtree map that keeps neighbouring subtrees on one XCD
#ifdef LS_TIER_STAMPS
    long long* stamps;                               // experiments build only: per wave TIER_STAMP_SLOTS clock stamps (100 MHz), nullptr = off
#endif
};

// Per-phase timeline of the tier kernels (build variant -DLS_TIER_STAMPS, tools/tier_stamps.py; never in the product library):
// slot 0 wave start, 1 header + right-hand-side gather done, 2 + 2 ph phase ph's work done, 3 + 2 ph its barrier passed,
// 16 + r the r-th leaf of the wave done (r < 8), 24 + ph items this wave ran in phase ph
constexpr int TIER_STAMP_SLOTS = 32;
#ifdef LS_TIER_STAMPS
#define LS_STAMP(slot) do { if (a.stamps && lane == 0) stamp_base[(slot)] = (long long)wall_clock64(); } while (0)
#define LS_STAMP_VAL(slot, v) do { if (a.stamps && lane == 0) stamp_base[(slot)] = (long long)(v); } while (0)
#else
#define LS_STAMP(slot) do { } while (0)
#define LS_STAMP_VAL(slot, v) do { } while (0)
#endif

// Item records and the workgroup header are fetched with VECTOR loads (lane i takes dword i) and unpacked with
// v_readlane: a scalar load on the critical path costs ~3 us next to a streaming CU (the scalar cache path queues behind
// the vector traffic), a vector load can be requested two items ahead at the price of one VGPR.
constexpr int ITEM_DWORDS = 24;
__device__ __forceinline__ int rec_load(const TierItem* items, int k, int lane) {
    return lane < ITEM_DWORDS ? reinterpret_cast<const int*>(items)[(size_t)k * ITEM_DWORDS + lane] : 0;
}
__device__ __forceinline__ int rl(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ TierItem rec_unpack(int r) {
    TierItem t;
    t.s = rl(r, 0); t.b = rl(r, 1); t.own_start = rl(r, 2); t.bnd_off = rl(r, 3); t.front_off = rl(r, 4); t.pfront_off = rl(r, 5);
    t.cix = rl(r, 6); t.flags = rl(r, 7);
    t.finv_off = (long long)(((unsigned long long)(unsigned)rl(r, 9) << 32) | (unsigned)rl(r, 8));
    t.w_off = (long long)(((unsigned long long)(unsigned)rl(r, 11) << 32) | (unsigned)rl(r, 10));
    t.spb_off = rl(r, 12); t.sps_off = rl(r, 13); t.row0 = rl(r, 14); t.r0 = rl(r, 15); t.r1 = rl(r, 16); t.part = rl(r, 17);
    t.nparts = rl(r, 18); t.pad0 = t.pad1 = t.pad2 = 0;
    return t;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the LDS queue is in order per wave, only the
// compiler has to be kept from moving the accesses across this point
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float bcast_lane(float v, int c) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c));
}

// Up-sweep hand-off INSIDE the tier: a child stores its update vector contiguously (entry i of its boundary list at
// upd[(bnd_off + i) * K], in the array the down sweep later uses for x_bnd) and the parent gathers per front position through
// a static pull list -- the slot scheme of the upper levels (arity x K floats per position, 12-byte pieces written by
// different workgroup phases) moved ~80 MB of partial cache lines per sweep at 1M vertices.
template <int K>
__device__ __forceinline__ void pull_compact(const TierArgs& a, size_t f, float (&v)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = 0.0f;
    const int* pl = a.pull + f * a.arity;
    for (int c = 0; c < a.arity; ++c) {
        const int idx = pl[c];
        if (idx >= 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] += a.xb[(size_t)idx * K + q];
        }
    }
}

// the same with the (static) pull indices already in registers: what is left behind a barrier is ONE round trip
template <int K>
__device__ __forceinline__ void pull_values(const TierArgs& a, const int (&pl)[4], float (&v)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (pl[c] >= 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] += a.xb[(size_t)pl[c] * K + q];
        }
    }
}
__device__ __forceinline__ void pull_indices(const TierArgs& a, size_t f, bool on, int (&pl)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) pl[c] = on ? a.pull[f * 4 + c] : -1;
}

// ---- sparse leaves --------------------------------------------------------------------------------------------------
struct LeafIdx {           // stage A: loads that depend on the item record only
    int g, p0, p1, pp;     // up: perm of own row `lane`; row pointers + parent position of boundary row `lane`
};                         // down: perm of own row `lane`; row pointers of own row `lane`
template <int K>
struct LeafDat {           // stage B: the triangle and the loads that depend on stage A
    float4 t[TIER_TRI4];
    float v[K];            // up: b of own row `lane`; down: y of own row `lane`
    float xv[K];           // down: x_bnd of boundary row `lane`
    SpEnt e[TIER_SPE];
};

template <bool UP>
__device__ __forceinline__ void leaf_idx(const TierArgs& a, const TierItem& n, int lane, LeafIdx& ix) {
    ix.g = lane < n.s ? a.perm[n.own_start + lane] : 0;
    if (UP) {
        const bool r = lane < n.b;
        ix.p0 = r ? a.sp_ptr[n.spb_off + lane] : 0;
        ix.p1 = r ? a.sp_ptr[n.spb_off + lane + 1] : 0;
        ix.pp = r ? a.ppos[n.bnd_off + lane] : 0;
    } else {
        const bool r = lane < n.s;
        ix.p0 = r ? a.sp_ptr[n.sps_off + lane] : 0;
        ix.p1 = r ? a.sp_ptr[n.sps_off + lane + 1] : 0;
        ix.pp = 0;
    }
}

template <int K, bool UP, bool NT>
__device__ __forceinline__ void leaf_dat(const TierArgs& a, const TierItem& n, const LeafIdx& ix, const float* __restrict__ b_in,
                                         int lane, LeafDat<K>& d) {
    // down sweep: x_bnd and the sparse entries are needed FIRST (the sparse product runs while the triangle is still in flight, see
    // leaf_phase), so they are requested first -- loads return in order
    if (!UP) {
#pragma unroll
        for (int q = 0; q < K; ++q) d.xv[q] = lane < n.b ? a.xb[(size_t)(n.bnd_off + lane) * K + q] : 0.0f;
    }
    // always a load from a valid address (a conditional load of a struct becomes a flat load through a scratch copy, and
    // flat loads would tie the LDS counter to these global loads): entry 0 of the array exists, unused values are zeroed
    const float2* __restrict__ ent = reinterpret_cast<const float2*>(a.sp_ent);
    if (!UP) {
#pragma unroll
        for (int t = 0; t < TIER_SPE; ++t) {
            const bool ok = ix.p0 + t < ix.p1;
            const float2 r = ld_stream2<NT>(ent + (ok ? ix.p0 + t : 0));
            d.e[t].val = ok ? r.x : 0.0f;
            d.e[t].idx = ok ? __float_as_int(r.y) : 0;
        }
    }
    const float4* __restrict__ p = reinterpret_cast<const float4*>(a.tri + n.finv_off);
    const int n4 = (n.s * (n.s + 1) / 2 + 3) >> 2;
#pragma unroll
    for (int e = 0; e < TIER_TRI4; ++e) {
        const int i = lane + e * 64;
        d.t[e] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
        if (UP) d.v[q] = lane < n.s ? b_in[(size_t)ix.g * K + q] : 0.0f;
        else d.v[q] = lane < n.s ? a.bprime[(size_t)(n.own_start + lane) * K + q] : 0.0f;
    }
    if (UP) {
#pragma unroll
        for (int t = 0; t < TIER_SPE; ++t) {
            const bool ok = ix.p0 + t < ix.p1;
            const float2 r = ld_stream2<NT>(ent + (ok ? ix.p0 + t : 0));
            d.e[t].val = ok ? r.x : 0.0f;
            d.e[t].idx = ok ? __float_as_int(r.y) : 0;
        }
    }
}

template <int K>
__device__ __forceinline__ void tri_stage(const LeafDat<K>& d, int s, int lane, float* stage) {
    float4* o = reinterpret_cast<float4*>(stage);
    const int n4 = (s * (s + 1) / 2 + 3) >> 2;
#pragma unroll
    for (int e = 0; e < TIER_TRI4; ++e) {
        const int i = lane + e * 64;
        if (i < n4) o[i] = d.t[e];
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The 4 x 4 x 1 matrix instruction in its 16-block form IS a row-per-lane mat-vec step for K <= 4 right-hand sides:
// block i = lanes 4i .. 4i + 3 computes D_i[m][n] += A_i[m] * B_i[n], and with A BROADCAST from one block (cbsz = 4,
// abid = e) every lane l gets  acc[m] += A_e[m] * B[l]:  A_e = the K values of vector entry e (lanes 4e .. 4e + 3 of a
// register that holds 16 consecutive entries of a [entry][4] vector -- ONE contiguous LDS read), B[l] = this lane's
// matrix element. One instruction per matrix element instead of K FMAs + K broadcasts; exact fp32 (an fmaf chain).
// Probed on the hardware: tools/mfma4x4_probe.hip.
template <int E>
__device__ __forceinline__ f32x4 mv_step(float vec16, float m, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(vec16, m, acc, 4, E, 0);
}

// acc[q] = sum_c Finv[row lane][c] v_c[q] over the packed triangle in LDS; vec4: the vector as [c][4] in LDS (entries
// c >= s must be ZERO, components q >= K finite). Two accumulators break the dependent chain; fixed order.
// Round 4 counted this loop in instructions (it is what a wave spends its issue slots on while 16 waves per CU stream triangles):
//   * element (lane j, column c) is stage[T(j) + c] for j >= c and stage[T(c) + j] otherwise: two LDS reads whose addresses are a
//     per-lane base + a COMPILE-TIME offset (no address arithmetic) and ONE select whose lane mask "j >= c" is a constant in a
//     scalar register pair (v_cndmask_b32_e64 with an SGPR mask: no per-element compare);
//   * every chunk of 16 columns runs this form, the last, partial one too: its columns c >= s meet zero vector entries, and what
//     they read -- inside the LDS triangle area, which the planner sizes for whole chunks (T(16 ceil(s_max / 16)) numbers) and the
//     kernel zero-fills once at its start, so that it only ever holds finite numbers -- contributes an exact 0. (Round 3 clamped
//     the column index of the partial chunk at run time: ~8 more instructions per element of it.)
// ~580 -> ~300 vector instructions per leaf and wave, the kernels' scalar-register spills 134 -> 4 (the three look-ahead records of
// the DMA variant and the masks of the columns >= 32 had been the rest); bitwise the same result. Measured (same box, C-ABI driver):
// 1M vertices 224-226 -> 219-221 us per solve, 4M 755-769 -> 723-739. The leaf rounds themselves did not get shorter (5.4 us per
// leaf and wave by the clock stamps, with half of the LDS reads removed as a timing experiment 4.6): they stream ~125 MB per sweep at
// 5.5+ TB/s -- see the note at LS_TIER_DMA below -- so neither instructions nor LDS nor the triangle's round trip is what they wait for.
template <int C>
__device__ __forceinline__ float tri_select(float lo, float hi) {       // lanes >= C: lo, lanes < C: hi
    if (C == 0) return lo;
    // the mask is formed where it is used, by ONE scalar instruction with inline constants (~0 << C): as a C++ constant the masks
    // of the columns >= 32 (two 32-bit literals each) were hoisted out of the leaf loop and spilled to vector lanes. The select itself
    // is the compiler's (inverse ballot: a scalar mask as a per-lane condition) -- it is a VALU write feeding a matrix instruction,
    // and the wait states between the two are the compiler's business (a v_cndmask inside an asm statement got none: wrong values
    // on some waves of some launches, measured)
    unsigned long long mask;
    asm volatile("s_lshl_b64 %0, -1, %1" : "=s"(mask) : "n"(C) : "scc");
    return __builtin_amdgcn_inverse_ballot_w64(mask) ? lo : hi;
}
template <int T, int E>
__device__ __forceinline__ float tri_element(const float* stage, int j, int tj) {
    constexpr int c = 16 * T + E;
    const float lo = stage[tj + c];
    if (c == 0) return lo;
    const float hi = stage[c * (c + 1) / 2 + j];
    return tri_select<c>(lo, hi);
}
template <int T>
__device__ __forceinline__ void tri_chunk(const float* stage, const float* vec4, int s, int lane, int j, int tj, f32x4& a0, f32x4& a1) {
    if (16 * T >= s) return;
    const float v16 = vec4[64 * T + lane];
    // all 32 LDS reads of the chunk are requested before the first matrix instruction (no branch per column: a branch ends
    // the basic block and with it the scheduler's freedom to hoist the reads)
    float m[16];
    {
    m[0] = tri_element<T, 0>(stage, j, tj);   m[1] = tri_element<T, 1>(stage, j, tj);   m[2] = tri_element<T, 2>(stage, j, tj);   m[3] = tri_element<T, 3>(stage, j, tj);
    m[4] = tri_element<T, 4>(stage, j, tj);   m[5] = tri_element<T, 5>(stage, j, tj);   m[6] = tri_element<T, 6>(stage, j, tj);   m[7] = tri_element<T, 7>(stage, j, tj);
    m[8] = tri_element<T, 8>(stage, j, tj);   m[9] = tri_element<T, 9>(stage, j, tj);   m[10] = tri_element<T, 10>(stage, j, tj); m[11] = tri_element<T, 11>(stage, j, tj);
    m[12] = tri_element<T, 12>(stage, j, tj); m[13] = tri_element<T, 13>(stage, j, tj); m[14] = tri_element<T, 14>(stage, j, tj); m[15] = tri_element<T, 15>(stage, j, tj);
    }
    a0 = mv_step<0>(v16, m[0], a0);   a1 = mv_step<1>(v16, m[1], a1);   a0 = mv_step<2>(v16, m[2], a0);   a1 = mv_step<3>(v16, m[3], a1);
    a0 = mv_step<4>(v16, m[4], a0);   a1 = mv_step<5>(v16, m[5], a1);   a0 = mv_step<6>(v16, m[6], a0);   a1 = mv_step<7>(v16, m[7], a1);
    a0 = mv_step<8>(v16, m[8], a0);   a1 = mv_step<9>(v16, m[9], a1);   a0 = mv_step<10>(v16, m[10], a0); a1 = mv_step<11>(v16, m[11], a1);
    a0 = mv_step<12>(v16, m[12], a0); a1 = mv_step<13>(v16, m[13], a1); a0 = mv_step<14>(v16, m[14], a0); a1 = mv_step<15>(v16, m[15], a1);
}
template <int K>
__device__ __forceinline__ void tri_matvec(const float* stage, const float* vec4, int s, int lane, float (&acc)[K]) {
    const int j = min(lane, s - 1);
    const int tj = j * (j + 1) / 2;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    tri_chunk<0>(stage, vec4, s, lane, j, tj, a0, a1);
    tri_chunk<1>(stage, vec4, s, lane, j, tj, a0, a1);
    tri_chunk<2>(stage, vec4, s, lane, j, tj, a0, a1);
    tri_chunk<3>(stage, vec4, s, lane, j, tj, a0, a1);
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = a0[q] + a1[q];
}

// row `lane` of the sparse block times a vector staged in LDS (4 floats per entry): prefetched entries, then the tail
template <int K>
__device__ __forceinline__ void sparse_row(const TierArgs& a, const LeafIdx& ix, const SpEnt (&e)[TIER_SPE], const float* vec4, float (&u)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) u[q] = 0.0f;
#pragma unroll
    for (int t = 0; t < TIER_SPE; ++t) {
        if (ix.p0 + t < ix.p1) {
#pragma unroll
            for (int q = 0; q < K; ++q) u[q] = fmaf(e[t].val, vec4[e[t].idx * 4 + q], u[q]);
        }
    }
    for (int p = ix.p0 + TIER_SPE; p < ix.p1; ++p) {
        const SpEnt z = a.sp_ent[p];
#pragma unroll
        for (int q = 0; q < K; ++q) u[q] = fmaf(z.val, vec4[z.idx * 4 + q], u[q]);
    }
}

// LDS of a leaf item: [triangle: tri_floats][y / t: 64 x 4]; the down sweep's x_bnd (b x 4) is staged in the triangle area before the triangle
template <int K>
__device__ __forceinline__ void leaf_up_compute(const TierArgs& a, const TierItem& n, const LeafIdx& ix, const float (&bj)[K],
                                                const SpEnt (&e)[TIER_SPE], float* region, int tri_floats) {
    const int lane = threadIdx.x & 63, s = n.s, b = n.b;
    float* yv = region + tri_floats;
    {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < s) { w.x = bj[0]; if (K > 1) w.y = bj[K > 1 ? 1 : 0]; if (K > 2) w.z = bj[K > 2 ? 2 : 0]; if (K > 3) w.w = bj[K > 3 ? 3 : 0]; }
        reinterpret_cast<float4*>(yv)[lane] = w;
    }
    wave_lds_sync();
    float y[K];
    tri_matvec<K>(region, yv, s, lane, y);
    wave_lds_sync();
    if (lane < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) { yv[lane * 4 + q] = y[q]; a.bprime[(size_t)(n.own_start + lane) * K + q] = y[q]; }
    }
    wave_lds_sync();
    if (n.pfront_off >= 0) {
        const bool upc = n.flags & NODE_UPC;
        if (lane < b) {
            float u[K];
            sparse_row<K>(a, ix, e, yv, u);
            const size_t dst = upc ? (size_t)(n.bnd_off + lane) * K : ((size_t)(n.pfront_off + ix.pp) * a.arity + n.cix) * K;
            float* out = upc ? a.xb : a.slots;
#pragma unroll
            for (int q = 0; q < K; ++q) out[dst + q] = u[q];
        }
        for (int i = lane + 64; i < b; i += 64) {          // leaves with more than 64 boundary rows: no prefetch
            const int p0 = a.sp_ptr[n.spb_off + i], p1 = a.sp_ptr[n.spb_off + i + 1], pp = a.ppos[n.bnd_off + i];
            float u[K];
#pragma unroll
            for (int q = 0; q < K; ++q) u[q] = 0.0f;
            for (int p = p0; p < p1; ++p) {
                const SpEnt z = a.sp_ent[p];
#pragma unroll
                for (int q = 0; q < K; ++q) u[q] = fmaf(z.val, yv[z.idx * 4 + q], u[q]);
            }
            const size_t dst = upc ? (size_t)(n.bnd_off + i) * K : ((size_t)(n.pfront_off + pp) * a.arity + n.cix) * K;
            float* out = upc ? a.xb : a.slots;
#pragma unroll
            for (int q = 0; q < K; ++q) out[dst + q] = u[q];
        }
    }
    wave_lds_sync();
}

// Down sweep of a leaf in two steps around the staging of its triangle. x_bnd lives in the TRIANGLE area of the wave's LDS region: it
// is dead once the sparse product t = A_sb x_bnd is in registers, and only then is the triangle written over it -- a leaf's LDS need
// is triangle + one vector whatever its boundary (round 6: a closed 1M-vertex scan has leaves of up to 83 boundary rows; with x_bnd
// BEHIND the triangle its 16-wave tier did not fit the 160 KB and the solve fell back to 11 launches on 3 workgroups per CU: 315 us).
// The arithmetic and its order are unchanged (bit-identical results).
template <int K>
__device__ __forceinline__ void leaf_down_head(const TierArgs& a, const TierItem& n, const LeafIdx& ix, const float (&xv)[K],
                                               const SpEnt (&e)[TIER_SPE], float* region, float (&t)[K]) {
    const int lane = threadIdx.x & 63, b = n.b;
    float* xbv = region;                                   // (4 b floats; the planner guarantees 4 b <= the triangle area)
    if (lane < b) {
#pragma unroll
        for (int q = 0; q < K; ++q) xbv[lane * 4 + q] = xv[q];
    }
    for (int i = lane + 64; i < b; i += 64) {
#pragma unroll
        for (int q = 0; q < K; ++q) xbv[i * 4 + q] = a.xb[(size_t)(n.bnd_off + i) * K + q];
    }
    wave_lds_sync();
    sparse_row<K>(a, ix, e, xbv, t);
    wave_lds_sync();                                       // every lane has read x_bnd: the triangle may overwrite it
}

template <int K>
__device__ __forceinline__ void leaf_down_vec(const TierItem& n, const float (&t)[K], float* region, int tri_floats) {
    const int lane = threadIdx.x & 63;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n.s) { w.x = t[0]; if (K > 1) w.y = t[K > 1 ? 1 : 0]; if (K > 2) w.z = t[K > 2 ? 2 : 0]; if (K > 3) w.w = t[K > 3 ? 3 : 0]; }
    reinterpret_cast<float4*>(region + tri_floats)[lane] = w;
}

template <int K>
__device__ __forceinline__ void leaf_down_tail(const TierItem& n, const LeafIdx& ix, const float (&yj)[K], float* __restrict__ x_out,
                                               float* region, int tri_floats) {
    const int lane = threadIdx.x & 63, s = n.s;
    float z[K];
    tri_matvec<K>(region, region + tri_floats, s, lane, z);
    if (lane < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[(size_t)ix.g * K + q] = yj[q] - z[q];
    }
    wave_lds_sync();
}

// The leaves of this wave in one phase: items k0, k0 + stride, ... < k1, one after the other. Only the small loads of the NEXT
// leaf are requested ahead (its item record and the index lists that depend on it: 5 registers). A full software pipeline
// (next leaf's triangle in flight during the multiply, 36 more registers) was measured slower: it spilled, and the leaf
// phase is bound by its dependent steps, not by bytes in flight.
template <int K, bool UP, int W, bool NT>
__device__ __forceinline__ void leaf_phase(const TierArgs& a, int k0, int k1, const float* __restrict__ b_in, float* __restrict__ x_out,
                                           float* region, int tri_floats, long long* stamp_base) {
    const int lane = threadIdx.x & 63;
    int leaf_no = 0;
    (void)leaf_no; (void)stamp_base;
    if (k0 >= k1) return;
    constexpr int S = W;
    TierItem it = rec_unpack(rec_load(a.items, k0, lane));
    int rec_n = k0 + S < k1 ? rec_load(a.items, k0 + S, lane) : 0;
    LeafIdx ix;
    leaf_idx<UP>(a, it, lane, ix);
    for (int k = k0; k < k1; k += S) {
        float v[K];
        SpEnt e[TIER_SPE];
        float t[K];
        {
            LeafDat<K> d;
            leaf_dat<K, UP, NT>(a, it, ix, b_in, lane, d);
            // down: the sparse product first (x_bnd staged in the triangle area), THEN the triangle over it
            if (!UP) leaf_down_head<K>(a, it, ix, d.xv, d.e, region, t);
            tri_stage<K>(d, it.s, lane, region);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = d.v[q];
#pragma unroll
            for (int u = 0; u < TIER_SPE; ++u) e[u] = d.e[u];
        }
        if (!UP) leaf_down_vec<K>(it, t, region, tri_floats);
        wave_lds_sync();
        // the next leaf's record has arrived by now: its index loads go out before this leaf is multiplied
        const bool more = k + S < k1;
        TierItem it_n = it;
        LeafIdx ix_n = ix;
        if (more) { it_n = rec_unpack(rec_n); leaf_idx<UP>(a, it_n, lane, ix_n); }
        rec_n = k + 2 * S < k1 ? rec_load(a.items, k + 2 * S, lane) : 0;
        if (UP) leaf_up_compute<K>(a, it, ix, v, e, region, tri_floats);
        else leaf_down_tail<K>(it, ix, v, x_out, region, tri_floats);
        if (leaf_no < 8) LS_STAMP(16 + leaf_no);
        ++leaf_no;
        it = it_n; ix = ix_n;
    }
}

// ---- dense nodes inside the tier ---------------------------------------------------------------------------------------
// Dense nodes of the tier keep their matrices QUAD-INTERLEAVED along the reduction: entry (row, t) of a stream sits at
// ((t / 4) * rows + row) * 4 + t % 4, reduction padded to a multiple of 4 with zeros. A lane owns a row and reads 16 bytes
// = 4 consecutive reduction entries per load; the 64 lanes of a wave read 1 KB contiguous -- the access the memory system
// wants (4-byte-per-lane row loads level off near 3 TB/s on this chip) -- and each loaded quad feeds 4 matrix instructions.
//     up    u4: rows = boundary rows i (b), reduction = own rows j (s -> s4)                 W[i][j]
//     down  d4: rows = own rows j (s), reduction = [own rows t (s -> s4) | boundary rows (b -> b4)]   [Finv | W^T][j][t]
#ifndef LS_TIER_Q
#define LS_TIER_Q 4
#endif
constexpr int TIER_Q = LS_TIER_Q;   // quads per lane and batch; two batches in flight

template <bool NT>
__device__ __forceinline__ void tier_prefetch(const float4* __restrict__ col, size_t rows, int q0, int q1, float4 (&cur)[TIER_Q]) {
#pragma unroll
    for (int e = 0; e < TIER_Q; ++e) cur[e] = (q0 + e < q1) ? ld_stream4<NT>(col + (size_t)(q0 + e) * rows) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// 4 quads (16 reduction entries) against one vector register; n = quads that exist
__device__ __forceinline__ void tier_mv16(float v16, const float4& a, const float4& b, const float4& c, const float4& d, int n, f32x4& acc) {
    acc = mv_step<0>(v16, a.x, acc); acc = mv_step<1>(v16, a.y, acc); acc = mv_step<2>(v16, a.z, acc); acc = mv_step<3>(v16, a.w, acc);
    if (n > 1) { acc = mv_step<4>(v16, b.x, acc); acc = mv_step<5>(v16, b.y, acc); acc = mv_step<6>(v16, b.z, acc); acc = mv_step<7>(v16, b.w, acc); }
    if (n > 2) { acc = mv_step<8>(v16, c.x, acc); acc = mv_step<9>(v16, c.y, acc); acc = mv_step<10>(v16, c.z, acc); acc = mv_step<11>(v16, c.w, acc); }
    if (n > 3) { acc = mv_step<12>(v16, d.x, acc); acc = mv_step<13>(v16, d.y, acc); acc = mv_step<14>(v16, d.z, acc); acc = mv_step<15>(v16, d.w, acc); }
}

// acc[m] += sum over the quads [q0, q1) of col[q * rows] (4 entries each) times the vector sv4[(4 q - base) * 4 + m].
// A = batch q0, already requested by the caller; B = batch q0 + TIER_Q. The two buffers form a ring: as soon as a batch
// is multiplied its registers are re-requested for the batch two ahead, so two batches are in flight all the time (the
// products of a batch are a few dozen cycles: with one batch in flight every batch cost a whole memory round trip).
__device__ __forceinline__ void tier_mv_batch(const float* sv4, int base, int q, int q1, const float4 (&c)[TIER_Q], f32x4& acc) {
    const int lane = threadIdx.x & 63;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TIER_Q == 2) tier_mv16(sv4[(4 * q - base) * 4 + lane], c[0], c[1], z, z, min(2, q1 - q), acc);
    if (TIER_Q >= 4) tier_mv16(sv4[(4 * q - base) * 4 + lane], c[0], c[1], c[TIER_Q >= 4 ? 2 : 0], c[TIER_Q >= 4 ? 3 : 0], q1 - q, acc);
    if (TIER_Q >= 8 && q + 4 < q1) tier_mv16(sv4[(4 * q + 16 - base) * 4 + lane], c[TIER_Q >= 8 ? 4 : 0], c[TIER_Q >= 8 ? 5 : 0],
                                             c[TIER_Q >= 8 ? 6 : 0], c[TIER_Q >= 8 ? 7 : 0], q1 - q - 4, acc);
}
// Round 6: B is requested by the CALLER too, at the item's entry -- most dense items of a 16-wave tier are 8 quads long, and with B
// requested here, behind the assembly of the vector, its round trip was a dependent step of every dense phase.
template <bool NT>
__device__ __forceinline__ void tier_dot(const float4* __restrict__ col, size_t rows, int q0, int q1, const float* sv4, int base,
                                         float4 (&A)[TIER_Q], float4 (&B)[TIER_Q], f32x4& acc) {
    for (int q = q0; q < q1; q += 2 * TIER_Q) {
        tier_mv_batch(sv4, base, q, q1, A, acc);
        tier_prefetch<NT>(col, rows, q + 2 * TIER_Q, q1, A);
        if (q + TIER_Q < q1) tier_mv_batch(sv4, base, q + TIER_Q, q1, B, acc);
        tier_prefetch<NT>(col, rows, q + 3 * TIER_Q, q1, B);
    }
}

// a split node's row chunk: the sum of its parts (part 0 already carries the children's hand-over) goes to the parent;
// pp = parent position of boundary row i (requested by the part-0 item before the parts met)
template <int K>
__device__ __forceinline__ void node_up_finish(const TierArgs& a, const TierItem& n, int i, int pp, const float (&acc)[K]) {
    if (i >= n.b || n.pfront_off < 0) return;
    if (n.flags & NODE_UPC) {
#pragma unroll
        for (int q = 0; q < K; ++q) a.xb[(size_t)(n.bnd_off + i) * K + q] = acc[q];
        return;
    }
    const size_t dst = ((size_t)(n.pfront_off + pp) * a.arity + n.cix) * K;
#pragma unroll
    for (int q = 0; q < K; ++q) a.slots[dst + q] = acc[q];
}

// What a dense item can request BEFORE the barrier that ends the previous phase (nothing here depends on that phase):
// the first batches of its matrix stream and static index / vector loads. The wave's first item of a phase gets this
// treatment; the round trips that remain behind the barrier are the children's slots (up) / the parent's x (down).
template <int K>
struct DensePre {
    float4 cur[TIER_Q];    // first batch of the matrix stream
    float v[K];            // down: b' of reduction entry r0 + lane (if it is an own row)
    int idx;               // up: parent position of this lane's boundary row; down: the caller's id of this lane's own row
    int plo[4], plb[4];    // up, arity 4: pull indices of own row r0 + lane and of boundary row row0 + lane (static lists)
    int fw0, fw1;          // down, inner node: push-list range of the boundary row this lane hands down (fwd_slice), fw0 > fw1: none
};

// The boundary rows of an inner node hand their x down to the children; that is independent of the node's own arithmetic (x_bnd is
// complete when the phase starts). Rounds 3-5: the node's first item did it for ALL boundary rows in a loop of two dependent round
// trips per 64 rows before its own work (level 4 at 1M: 234 rows = four iterations in front of the phase's critical wave -- round 6's
// stamps: that phase was the longest dense phase of the down sweep). Now EVERY item of the node takes a slice of the rows (<= 64 each
// when the node has at least b / 64 items), its row pointers are requested before the barrier, value and targets at the item's entry,
// and the stores go out after the item's product.
__device__ __forceinline__ void fwd_slice(const TierItem& it, int& f0, int& f1) {
    const int chunks = (max(it.s, 1) + 63) >> 6, total = chunks * it.nparts, ord = (it.row0 >> 6) * it.nparts + it.part;
    f0 = (int)((long long)it.b * ord / total);
    f1 = (int)((long long)it.b * (ord + 1) / total);
}

template <int K, bool NT>
__device__ __forceinline__ void node_up_pre(const TierArgs& a, const TierItem& it, DensePre<K>& P) {
    const int lane = threadIdx.x & 63, i = it.row0 + lane;
    const bool row = i < it.b;
    const float4* col = reinterpret_cast<const float4*>(a.u4 + it.w_off) + (row ? i : 0);
    tier_prefetch<NT>(col, (size_t)it.b, it.r0 >> 2, row ? it.r1 >> 2 : it.r0 >> 2, P.cur);
    // the item that finishes a (row chunk of a) node -- the only part, or part 0 of a split node -- adds what the children
    // hand to the boundary rows and knows where the result goes
    const bool fin = it.part == 0 && row && it.pfront_off >= 0;
    P.idx = (fin && !(it.flags & NODE_UPC)) ? a.ppos[it.bnd_off + i] : 0;
    const bool inner = !(it.flags & NODE_LEAF) && a.arity == 4;
    const int j = it.r0 + lane;
    pull_indices(a, (size_t)(it.front_off + j), inner && j < it.r1 && j < it.s, P.plo);
    // the boundary rows' hand-over (needed by the item that finishes the row chunk): its static indices too -- round 6's stamps show
    // dense phases of 8-quad items whose whole stream is in flight before the barrier; two dependent round trips behind it were
    // the phase, not something the stream hid
    // (k = 4 columns: the four index registers across the barrier would be the kernel's 129th-130th VGPR -- it keeps the two-trip form)
    if (K <= 3) pull_indices(a, (size_t)(it.front_off + it.s + i), inner && fin, P.plb);
}

// up: b'_j for the item's reduction range (stored by the row0 == 0 items), partial upd_i = sum_j W[i][j] b'_j.
// braw: the right-hand side of the tier's dense rows in the tree's numbering (gathered at kernel start, see k_nd_tier).
template <int K, bool NT>
__device__ __forceinline__ void node_up(const TierArgs& a, const TierItem& it, DensePre<K>& P, const float* __restrict__ b_in, float* region, float* pbuf) {
    const int lane = threadIdx.x & 63, b = it.b;
    const int i = it.row0 + lane;
    const bool row = i < b;
    float* sv4 = region;
    const float4* __restrict__ col = reinterpret_cast<const float4*>(a.u4 + it.w_off) + (row ? i : 0);
    float4 B[TIER_Q];                               // second batch of the matrix stream: in flight while the vector is assembled
    tier_prefetch<NT>(col, (size_t)b, (it.r0 >> 2) + TIER_Q, it.r1 >> 2, B);
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    const bool fin = it.part == 0 && row && it.pfront_off >= 0;        // this item adds the children's hand-over (and, unsplit, stores)
    const bool pre4 = a.arity == 4;
    if (fin && !(it.flags & NODE_LEAF)) {
        if (pre4 && K <= 3) pull_values<K>(a, P.plb, pass); else pull_compact<K>(a, (size_t)(it.front_off + it.s + i), pass);
    }
    for (int j = it.r0 + lane; j < it.r1; j += 64) {
        float v[K];
        if (j >= it.s) {                              // padding of the reduction to a multiple of 4
#pragma unroll
            for (int q = 0; q < K; ++q) sv4[(j - it.r0) * 4 + q] = 0.0f;
            continue;
        }
        if (it.flags & NODE_LEAF) {                   // a dense node on the leaf level (phase 0: nothing was gathered for it)
            const size_t g = (size_t)a.perm[it.own_start + j];
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = b_in[g * K + q];
        } else {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = a.braw[(size_t)(it.own_start + j) * K + q];        // gathered at kernel start
        }
        if (!(it.flags & NODE_LEAF)) {
            float u[K];
            if (pre4 && j == it.r0 + lane) pull_values<K>(a, P.plo, u); else pull_compact<K>(a, (size_t)(it.front_off + j), u);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] -= u[q];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) sv4[(j - it.r0) * 4 + q] = v[q];
        if (it.row0 == 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) a.bprime[(size_t)(it.own_start + j) * K + q] = v[q];
        }
    }
    wave_lds_sync();
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    tier_dot<NT>(col, (size_t)b, it.r0 >> 2, it.r1 >> 2, sv4, it.r0, P.cur, B, a4);   // every lane takes part (matrix instruction)
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = a4[q];
    if (it.nparts == 1) {
        if (fin) {
            const bool upc = it.flags & NODE_UPC;
            const size_t dst = upc ? (size_t)(it.bnd_off + i) * K : ((size_t)(it.pfront_off + P.idx) * a.arity + it.cix) * K;
            float* out = upc ? a.xb : a.slots;
#pragma unroll
            for (int q = 0; q < K; ++q) out[dst + q] = acc[q] + pass[q];
        }
    } else {                                        // part 0 carries the children's hand-over into the sum of the parts
#pragma unroll
        for (int q = 0; q < K; ++q) pbuf[lane * 4 + q] = acc[q] + pass[q];
    }
    wave_lds_sync();
}

// x of the own rows: to the caller's numbering and into the children's boundary vectors. g, [p0, p1) and the first four
// targets tg were requested before the arithmetic (static lists); a list longer than four entries walks the rest.
template <int K>
__device__ __forceinline__ void node_down_store(const TierArgs& a, const TierItem& it, int g, int p0, int p1, const int (&tg)[4],
                                                float* __restrict__ x_out, const float (&acc)[K]) {
    const int j = it.row0 + (threadIdx.x & 63);
    if (j >= it.s) return;
#pragma unroll
    for (int q = 0; q < K; ++q) x_out[(size_t)g * K + q] = acc[q];
    if (it.flags & NODE_LEAF) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (tg[c] >= 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) a.xb[(size_t)tg[c] * K + q] = acc[q];
        }
    }
    if (p1 - p0 > 4) push_down<K>(a.push_tgt, p0 + 4, p1, a.xb, acc);
}

// the same without anything requested ahead (a wave that finishes more than one split node in a phase)
template <int K>
__device__ __forceinline__ void node_down_finish(const TierArgs& a, const TierItem& it, float* __restrict__ x_out, const float (&acc)[K]) {
    const int j = it.row0 + (threadIdx.x & 63);
    if (j >= it.s) return;
    const size_t g = (size_t)a.perm[it.own_start + j];
#pragma unroll
    for (int q = 0; q < K; ++q) x_out[g * K + q] = acc[q];
    if (!(it.flags & NODE_LEAF)) push_down<K>(a.push_tgt, a.push_ptr[it.front_off + j], a.push_ptr[it.front_off + j + 1], a.xb, acc);
}

template <int K, bool NT>
__device__ __forceinline__ void node_down_pre(const TierArgs& a, const TierItem& it, DensePre<K>& P) {
    const int lane = threadIdx.x & 63, s = it.s, j = it.row0 + lane;
    const bool row = j < s;
    tier_prefetch<NT>(reinterpret_cast<const float4*>(a.d4 + it.finv_off) + (row ? j : 0), (size_t)s, it.r0 >> 2, row ? it.r1 >> 2 : it.r0 >> 2, P.cur);
    const int t = it.r0 + lane;
#pragma unroll
    for (int q = 0; q < K; ++q) P.v[q] = (t < it.r1 && t < s) ? a.bprime[(size_t)(it.own_start + t) * K + q] : 0.0f;
    // the item that finishes a row chunk (the only part, or part 0 of a split node): where x goes -- static, requested now
    const bool fin = it.part == 0 && row;
    const bool push = fin && !(it.flags & NODE_LEAF);
    P.idx = fin ? a.perm[it.own_start + j] : 0;
    P.plo[0] = push ? a.push_ptr[it.front_off + j] : 0;
    P.plo[1] = push ? a.push_ptr[it.front_off + j + 1] : 0;
    int f0, f1;
    fwd_slice(it, f0, f1);
    const bool fw = !(it.flags & NODE_LEAF) && f0 + lane < f1;
    P.fw0 = fw ? a.push_ptr[it.front_off + s + f0 + lane] : 1;
    P.fw1 = fw ? a.push_ptr[it.front_off + s + f0 + lane + 1] : 0;
}

// down: partial x_j = sum_{t in [r0, r1)} [Finv | -W^T][j][t] * [b' | x_bnd][t]
template <int K, bool NT>
__device__ __forceinline__ void node_down(const TierArgs& a, const TierItem& it, DensePre<K>& P, float* __restrict__ x_out, float* region, float* pbuf) {
    const int lane = threadIdx.x & 63, s = it.s;
    const int j = it.row0 + lane;
    const bool row = j < s;
    float* sv4 = region;
    const float4* __restrict__ col = reinterpret_cast<const float4*>(a.d4 + it.finv_off) + (row ? j : 0);
    float4 B[TIER_Q];                               // second batch of the matrix stream: in flight while the vector is assembled
    tier_prefetch<NT>(col, (size_t)s, (it.r0 >> 2) + TIER_Q, it.r1 >> 2, B);
    // this item's slice of the boundary rows to hand down: value and the first two targets requested now, stored after the product
    int f0, f1;
    fwd_slice(it, f0, f1);
    const bool fw = P.fw0 <= P.fw1;
    float fv[K];
    int ftg[2];
#pragma unroll
    for (int q = 0; q < K; ++q) fv[q] = fw ? a.xb[(size_t)(it.bnd_off + f0 + lane) * K + q] : 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) ftg[c] = (fw && P.fw0 + c < P.fw1) ? a.push_tgt[P.fw0 + c] : -1;
    const int s4 = (s + 3) & ~3;                  // reduction index space: [0, s4) own rows (padded), [s4, ..) boundary rows (padded)
    for (int t = it.r0 + lane; t < it.r1; t += 64) {
        float v[K];
        if (t < s) {
            if (t == it.r0 + lane) {
#pragma unroll
                for (int q = 0; q < K; ++q) v[q] = P.v[q];
            } else {
#pragma unroll
                for (int q = 0; q < K; ++q) v[q] = a.bprime[(size_t)(it.own_start + t) * K + q];
            }
        } else if (t >= s4 && t - s4 < it.b) {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = -a.xb[(size_t)(it.bnd_off + t - s4) * K + q];
        } else {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) sv4[(t - it.r0) * 4 + q] = v[q];
    }
    wave_lds_sync();
    // the push targets of this lane's own row (the row pointers arrived while the vector was assembled)
#pragma unroll
    for (int c = 0; c < 4; ++c) P.plb[c] = (it.nparts == 1 && P.plo[0] + c < P.plo[1]) ? a.push_tgt[P.plo[0] + c] : -1;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    tier_dot<NT>(col, (size_t)s, it.r0 >> 2, it.r1 >> 2, sv4, it.r0, P.cur, B, a4);
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = a4[q];
    if (it.nparts == 1) node_down_store<K>(a, it, P.idx, P.plo[0], P.plo[1], P.plb, x_out, acc);
    else {
#pragma unroll
        for (int q = 0; q < K; ++q) pbuf[lane * 4 + q] = acc[q];
    }
    // the hand-down of this item's slice of boundary rows (requested at entry)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (ftg[c] >= 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) a.xb[(size_t)ftg[c] * K + q] = fv[q];
        }
    }
    if (fw && P.fw1 - P.fw0 > 2) push_down<K>(a.push_tgt, P.fw0 + 2, P.fw1, a.xb, fv);
    if (!(it.flags & NODE_LEAF)) {
        for (int i = f0 + 64 + lane; i < f1; i += 64) {          // a slice of more than 64 rows (a node with fewer items than b / 64)
            const size_t f = (size_t)(it.front_off + s + i);
            const int p0 = a.push_ptr[f], p1 = a.push_ptr[f + 1];
            float v[K];
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = a.xb[(size_t)(it.bnd_off + i) * K + q];
            push_down<K>(a.push_tgt, p0, p1, a.xb, v);
        }
    }
    wave_lds_sync();
}

// One workgroup per subtree. UP: phases run leaves -> tier root. DOWN: tier root -> leaves.
template <int K, bool UP, int W, bool NT>
__global__ __launch_bounds__(64 * W, 16 / W) void k_nd_tier(TierArgs a, const float* __restrict__ b_in, float* __restrict__ x_out,
                                                              int tri_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float* region = sm + (size_t)wave * a.region_floats;
    // a wave's LDS region only ever holds finite numbers: zeros now, then factor data, vectors and partial sums. The leaves' mat-vec
    // relies on it (the columns of a partial chunk beyond the leaf's size read what lies there and multiply it by an exact zero)
    for (int i = lane; 4 * i < a.region_floats; i += 64) reinterpret_cast<float4*>(region)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_sync();
    // workgroup header: up_off[7] | down_off[7] | up_split down_split up_leaf down_leaf | n_dense pad | dense_rng[12]
    // XCD-aware subtree order: workgroup b runs on XCD b % 8 (observed placement, speed only); consecutive subtrees are spatial
    // neighbours (they share cache lines of b / x in the caller's numbering and of the hand-off arrays), so each XCD takes a
    // contiguous eighth of them: subtree = (b % 8) * (n / 8) + b / 8
    const int n_wg = (int)gridDim.x;
    const int sub = (a.xcd_order && (n_wg & 7) == 0) ? (int)(blockIdx.x & 7) * (n_wg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#ifdef LS_TIER_STAMPS
    long long* stamp_base = a.stamps ? a.stamps + ((size_t)sub * W + wave) * TIER_STAMP_SLOTS : nullptr;
#else
    long long* stamp_base = nullptr;
#endif
    LS_STAMP(0);
    const int hdr = lane < 32 ? reinterpret_cast<const int*>(a.wgs + sub)[lane] : 0;
    const int obase = UP ? 0 : TIER_MAX_H + 1;
    const unsigned split = (unsigned)rl(hdr, UP ? 14 : 15), leafy = (unsigned)rl(hdr, UP ? 16 : 17);
    if (UP) {
        // right-hand side of the tier's inner-node rows, gathered once into the tree's numbering (braw): the dense items
        // read it from a static address instead of walking perm -> b behind a barrier
        const int n_dense = rl(hdr, 18);
        for (int r = threadIdx.x; r < n_dense; r += blockDim.x) {
            // the dense rows of a subtree are one range per level; ranges are listed as (start, count) pairs
            int o = r, start = 0;
#pragma unroll
            for (int t = 0; t < TIER_MAX_H; ++t) {
                const int cnt = rl(hdr, 20 + 2 * t + 1);
                const bool here = o >= 0 && o < cnt;
                start = here ? rl(hdr, 20 + 2 * t) + o : start;
                o = here ? -1 : o - cnt;
            }
            const size_t g = (size_t)a.perm[start];
#pragma unroll
            for (int q = 0; q < K; ++q) a.braw[(size_t)start * K + q] = b_in[g * K + q];
        }
        // ... and the rows of the levels above the tier (a few per workgroup): their launches start from braw, not from perm -> b
        for (int r = a.upper_lo + blockIdx.x * blockDim.x + threadIdx.x; r < a.upper_hi; r += gridDim.x * blockDim.x) {
            const size_t g = (size_t)a.perm[r];
#pragma unroll
            for (int q = 0; q < K; ++q) a.braw[(size_t)r * K + q] = b_in[g * K + q];
        }
    }
    LS_STAMP(1);
    DensePre<K> pre;
#pragma unroll
    for (int e = 0; e < TIER_Q; ++e) pre.cur[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < K; ++q) pre.v[q] = 0.0f;
    pre.idx = 0; pre.fw0 = 1; pre.fw1 = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { pre.plo[c] = -1; pre.plb[c] = -1; }
    TierItem it_pre = rec_unpack(0);
    bool pre_valid = false;
    for (int ph = 0; ph < a.phases; ++ph) {
        const int i0 = rl(hdr, obase + ph), i1 = rl(hdr, obase + ph + 1);
        // the record of this wave's first item of the NEXT phase: requested now, used before this phase's barrier
        int rec_nx = 0;
        bool has_nx = false;
        if (ph + 1 < a.phases && !((leafy >> (ph + 1)) & 1u)) {
            const int j0 = rl(hdr, obase + ph + 1) + wave;
            has_nx = j0 < rl(hdr, obase + ph + 2);
            if (has_nx) rec_nx = rec_load(a.items, j0, lane);
        }
        if ((leafy >> ph) & 1u) {
            leaf_phase<K, UP, W, NT>(a, i0 + wave, i1, b_in, x_out, region, tri_floats, stamp_base);
        } else {
            int rec = (!pre_valid && i0 + wave < i1) ? rec_load(a.items, i0 + wave, lane) : 0;
            // the (last) split row chunk this wave will finish once its parts have met: record and static indices stay in
            // registers across that barrier, nothing is loaded behind it
            int head_k = -1, head_idx = 0, head_p0 = 0, head_p1 = 0, head_flags = 0;
            for (int k = i0 + wave; k < i1; k += W) {
                const int rec_next = k + W < i1 ? rec_load(a.items, k + W, lane) : 0;
                const bool first = k == i0 + wave && pre_valid;
                const TierItem it = first ? it_pre : rec_unpack(rec);
                rec = rec_next;
                float* pbuf = region + a.vec_floats + ((k - i0) / W) * 256;
                if (!first) { if (UP) node_up_pre<K, NT>(a, it, pre); else node_down_pre<K, NT>(a, it, pre); }
                if (UP) node_up<K, NT>(a, it, pre, b_in, region, pbuf);
                else node_down<K, NT>(a, it, pre, x_out, region, pbuf);
                if (it.nparts > 1 && it.part == 0) {
                    head_k = k; head_idx = pre.idx; head_p0 = pre.plo[0]; head_p1 = pre.plo[1]; head_flags = it.flags;
                }
            }
            if ((split >> ph) & 1u) {
                const int rec_head = head_k >= 0 ? rec_load(a.items, head_k, lane) : 0;      // arrives while the parts meet
                // ... and so do the first four push targets of the row this lane will store (static list; round 6: one round trip less
                // behind the combine barrier of the down sweep's split phases)
                int head_tg[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) head_tg[c] = (!UP && head_k >= 0 && head_p0 + c < head_p1 && !(head_flags & NODE_LEAF)) ? a.push_tgt[head_p0 + c] : -1;
                __syncthreads();
                for (int k = i0 + wave; k < i1; k += W) {
                    const bool mine = k == head_k;
                    const TierItem it = rec_unpack(mine ? rec_head : rec_load(a.items, k, lane));
                    if (it.nparts == 1 || it.part != 0) continue;
                    float acc[K];
#pragma unroll
                    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
                    for (int p = 0; p < it.nparts; ++p) {          // parts are consecutive items: fixed order of the sum
                        const int kp = k + p - i0;
                        const float* pb = sm + (size_t)(kp % W) * a.region_floats + a.vec_floats + (kp / W) * 256;
#pragma unroll
                        for (int q = 0; q < K; ++q) acc[q] += pb[lane * 4 + q];
                    }
                    if (UP) {
                        const int i = it.row0 + lane;
                        const int pp = mine ? head_idx : ((i < it.b && it.pfront_off >= 0 && !(it.flags & NODE_UPC)) ? a.ppos[it.bnd_off + i] : 0);
                        node_up_finish<K>(a, it, i, pp, acc);
                    } else if (mine) {
                        node_down_store<K>(a, it, head_idx, head_p0, head_p1, head_tg, x_out, acc);
                    }
                    else node_down_finish<K>(a, it, x_out, acc);
                }
            }
        }
        if (ph < 6) { LS_STAMP(2 + 2 * ph); LS_STAMP_VAL(24 + ph, (i1 - i0 - wave + W - 1) / W); }
        // every field of `pre` is rewritten here on every path: nothing of it stays live across a leaf phase
        pre_valid = has_nx;
        it_pre = rec_unpack(rec_nx);
        if (pre_valid) {
            if (UP) node_up_pre<K, NT>(a, it_pre, pre); else node_down_pre<K, NT>(a, it_pre, pre);
        } else {
#pragma unroll
            for (int e = 0; e < TIER_Q; ++e) pre.cur[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < K; ++q) pre.v[q] = 0.0f;
            pre.idx = 0; pre.fw0 = 1; pre.fw1 = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { pre.plo[c] = -1; pre.plb[c] = -1; }
        }
        __syncthreads();      // workgroup-scope release/acquire of the slots / boundary vectors written above (same CU)
        if (ph < 6) LS_STAMP(3 + 2 * ph);
    }
}

}  // namespace ls
