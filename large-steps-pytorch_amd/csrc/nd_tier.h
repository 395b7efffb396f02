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
#pragma once

namespace ls {

constexpr int TIER_MAX_H = 6;            // tree levels one workgroup may walk
constexpr int TIER_WAVES = 4;            // waves per tier workgroup
constexpr int TIER_TRI4 = 9;             // 16-byte loads per lane that hold a leaf triangle (s <= 64: 2080 floats = 520 float4)

enum : int { NODE_LEAF = 1, NODE_SPARSE = 2 };

struct alignas(64) NodeD {
    int s, b, own_start, bnd_off, front_off, pfront_off, cix, flags;
    long long finv_off, w_off;           // sparse leaf: finv_off = offset of the packed triangle in `tri`
    int spb_off, sps_off;                // sparse leaf: offsets of the row pointers (boundary rows / own rows) in sp_ptr
    int pad[2];
};

struct SpEnt { float val; int idx; };

struct alignas(32) TierItem {
    int node, row0, r0, r1, part, nparts, pad0, pad1;
};

struct alignas(64) TierWG {
    int up_off[TIER_MAX_H + 1];          // item ranges per phase of the up sweep (deepest level first) into TierItem[]
    int down_off[TIER_MAX_H + 1];        // ... of the down sweep (tier root first)
    unsigned up_split, down_split;       // bit p: phase p has items cut into parts (needs the combine step)
};

struct TierArgs {
    const NodeD* nodes;
    const TierItem* items;
    const TierWG* wgs;
    const int* perm;
    const unsigned char* mask;
    const int* ppos;
    const int* push_ptr;
    const int* push_tgt;
    const float* finv;
    const float* wf;
    const float* wb;
    const float* tri;
    const int* sp_ptr;
    const SpEnt* sp_ent;
    float* bprime;
    float* slots;
    float* xb;
    int arity, phases, region_floats, vec_floats;    // LDS per wave: [vec_floats | partial sums: rounds x 64 x 4]
};

// LDS written by some lanes of a wave and read by others of the SAME wave: the LDS queue is in order per wave, only the
// compiler has to be kept from moving the accesses across this point
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int K>
__device__ __forceinline__ float bcast_lane(float v, int c) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), c));
}

// ---- sparse leaves --------------------------------------------------------------------------------------------------
struct TriRegs { float4 t[TIER_TRI4]; };

__device__ __forceinline__ void tri_load(const float* __restrict__ tri, long long off, int s, int lane, TriRegs& r) {
    const float4* __restrict__ p = reinterpret_cast<const float4*>(tri + off);
    const int n4 = (s * (s + 1) / 2 + 3) >> 2;
#pragma unroll
    for (int e = 0; e < TIER_TRI4; ++e) {
        const int i = lane + e * 64;
        r.t[e] = i < n4 ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void tri_stage(const TriRegs& r, int s, int lane, float* stage) {
    float4* d = reinterpret_cast<float4*>(stage);
    const int n4 = (s * (s + 1) / 2 + 3) >> 2;
#pragma unroll
    for (int e = 0; e < TIER_TRI4; ++e) {
        const int i = lane + e * 64;
        if (i < n4) d[i] = r.t[e];
    }
}
// acc_j = sum_c Finv[j][c] v_c over the staged triangle; v lives one row per lane (lane c holds v_c)
template <int K>
__device__ __forceinline__ void tri_matvec(const float* stage, int s, int lane, const float (&v)[K], float (&acc)[K]) {
    const int j = min(lane, s - 1);
    const int tj = j * (j + 1) / 2;
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    int tc = 0;                                   // T(c)
    for (int c = 0; c < s; ++c) {
        const float a = stage[c <= j ? tj + c : tc + j];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = fmaf(a, bcast_lane<K>(v[q], c), acc[q]);
        tc += c + 1;
    }
}

// LDS of a leaf item: [triangle: tri_floats][vec: 64 x 4][xbv: b x 4]
template <int K>
__device__ __forceinline__ void leaf_up(const TierArgs& a, const NodeD& n, const float* __restrict__ b_in, float* region, int tri_floats) {
    const int lane = threadIdx.x & 63, s = n.s, b = n.b;
    float* stage = region;
    float* yv = region + tri_floats;
    TriRegs tr;
    tri_load(a.tri, n.finv_off, s, lane, tr);
    float bj[K];
    {
        const size_t g = lane < s ? (size_t)a.perm[n.own_start + lane] : 0;
#pragma unroll
        for (int q = 0; q < K; ++q) bj[q] = lane < s ? b_in[g * K + q] : 0.0f;
    }
    tri_stage(tr, s, lane, stage);
    wave_lds_sync();
    float y[K];
    tri_matvec<K>(stage, s, lane, bj, y);
    if (lane < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) { yv[lane * 4 + q] = y[q]; a.bprime[(size_t)(n.own_start + lane) * K + q] = y[q]; }
    }
    wave_lds_sync();
    if (n.pfront_off < 0) return;
    for (int i = lane; i < b; i += 64) {
        const int p0 = a.sp_ptr[n.spb_off + i], p1 = a.sp_ptr[n.spb_off + i + 1];
        const int pp = a.ppos[n.bnd_off + i];
        float u[K];
#pragma unroll
        for (int q = 0; q < K; ++q) u[q] = 0.0f;
        for (int p = p0; p < p1; ++p) {
            const SpEnt e = a.sp_ent[p];
#pragma unroll
            for (int q = 0; q < K; ++q) u[q] = fmaf(e.val, yv[e.idx * 4 + q], u[q]);
        }
        const size_t dst = ((size_t)(n.pfront_off + pp) * a.arity + n.cix) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) a.slots[dst + q] = u[q];
    }
}

template <int K>
__device__ __forceinline__ void leaf_down(const TierArgs& a, const NodeD& n, float* __restrict__ x_out, float* region, int tri_floats) {
    const int lane = threadIdx.x & 63, s = n.s, b = n.b;
    float* stage = region;
    float* xbv = region + tri_floats + 64 * 4;
    TriRegs tr;
    tri_load(a.tri, n.finv_off, s, lane, tr);
    float yj[K];
    size_t g = 0;
    int p0 = 0, p1 = 0;
    if (lane < s) {
        g = (size_t)a.perm[n.own_start + lane];
        p0 = a.sp_ptr[n.sps_off + lane]; p1 = a.sp_ptr[n.sps_off + lane + 1];
    }
#pragma unroll
    for (int q = 0; q < K; ++q) yj[q] = lane < s ? a.bprime[(size_t)(n.own_start + lane) * K + q] : 0.0f;
    for (int i = lane; i < b; i += 64) {
#pragma unroll
        for (int q = 0; q < K; ++q) xbv[i * 4 + q] = a.xb[(size_t)(n.bnd_off + i) * K + q];
    }
    tri_stage(tr, s, lane, stage);
    wave_lds_sync();
    float t[K];
#pragma unroll
    for (int q = 0; q < K; ++q) t[q] = 0.0f;
    for (int p = p0; p < p1; ++p) {
        const SpEnt e = a.sp_ent[p];
#pragma unroll
        for (int q = 0; q < K; ++q) t[q] = fmaf(e.val, xbv[e.idx * 4 + q], t[q]);
    }
    float z[K];
    tri_matvec<K>(stage, s, lane, t, z);
    if (lane < s) {
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[g * K + q] = yj[q] - z[q];
    }
    wave_lds_sync();
}

// ---- dense nodes inside the tier ---------------------------------------------------------------------------------------
constexpr int TIER_U = 8;        // strided matrix loads per lane and batch; two batches in flight

// acc += sum_{u in [u0, u1)} col[u * stride] * sv4[(u - base) * 4 + q]
template <int K>
__device__ __forceinline__ void tier_dot(const float* __restrict__ col, size_t stride, int u0, int u1, const float* sv4, int base, float (&acc)[K]) {
    float cur[TIER_U], nxt[TIER_U];
#pragma unroll
    for (int e = 0; e < TIER_U; ++e) cur[e] = (u0 + e < u1) ? col[(size_t)(u0 + e) * stride] : 0.0f;
    for (int u = u0; u < u1; u += TIER_U) {
#pragma unroll
        for (int e = 0; e < TIER_U; ++e) nxt[e] = (u + TIER_U + e < u1) ? col[(size_t)(u + TIER_U + e) * stride] : 0.0f;
#pragma unroll
        for (int e = 0; e < TIER_U; ++e) {
            if (u + e < u1) {
#pragma unroll
                for (int q = 0; q < K; ++q) acc[q] = fmaf(cur[e], sv4[(u + e - base) * 4 + q], acc[q]);
            }
        }
#pragma unroll
        for (int e = 0; e < TIER_U; ++e) cur[e] = nxt[e];
    }
}

template <int K>
__device__ __forceinline__ void node_up_finish(const TierArgs& a, const NodeD& n, int i, const float (&acc)[K]) {
    if (i >= n.b || n.pfront_off < 0) return;
    float pass[K];
#pragma unroll
    for (int q = 0; q < K; ++q) pass[q] = 0.0f;
    if (!(n.flags & NODE_LEAF)) pull_slots<K>(a.slots, a.mask, (size_t)(n.front_off + n.s + i), a.arity, pass);
    const int pp = a.ppos[n.bnd_off + i];
    const size_t dst = ((size_t)(n.pfront_off + pp) * a.arity + n.cix) * K;
#pragma unroll
    for (int q = 0; q < K; ++q) a.slots[dst + q] = acc[q] + pass[q];
}

// up: b'_j for the item's reduction range (stored by the row0 == 0 items), partial upd_i = sum_j W[i][j] b'_j
template <int K>
__device__ __forceinline__ void node_up(const TierArgs& a, const NodeD& n, const TierItem& it, const float* __restrict__ b_in,
                                        float* region, float* pbuf) {
    const int lane = threadIdx.x & 63, s = n.s, b = n.b;
    const int i = it.row0 + lane;
    const bool row = i < b;
    float* sv4 = region;
    for (int j = it.r0 + lane; j < it.r1; j += 64) {
        const size_t g = (size_t)a.perm[n.own_start + j];
        float v[K];
#pragma unroll
        for (int q = 0; q < K; ++q) v[q] = b_in[g * K + q];
        if (!(n.flags & NODE_LEAF)) {
            float u[K];
            pull_slots<K>(a.slots, a.mask, (size_t)(n.front_off + j), a.arity, u);
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] -= u[q];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) sv4[(j - it.r0) * 4 + q] = v[q];
        if (it.row0 == 0) {
#pragma unroll
            for (int q = 0; q < K; ++q) a.bprime[(size_t)(n.own_start + j) * K + q] = v[q];
        }
    }
    wave_lds_sync();
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    if (row) tier_dot<K>(a.wf + n.w_off + i, (size_t)b, it.r0, it.r1, sv4, it.r0, acc);
    if (it.nparts == 1) node_up_finish<K>(a, n, i, acc);
    else {
#pragma unroll
        for (int q = 0; q < K; ++q) pbuf[lane * 4 + q] = acc[q];
    }
    wave_lds_sync();
    (void)s;
}

template <int K>
__device__ __forceinline__ void node_down_finish(const TierArgs& a, const NodeD& n, const TierItem& it, float* __restrict__ x_out,
                                                 const float (&acc)[K]) {
    const int lane = threadIdx.x & 63;
    const int j = it.row0 + lane;
    const bool inner = !(n.flags & NODE_LEAF);
    if (j < n.s) {
        const size_t g = (size_t)a.perm[n.own_start + j];
#pragma unroll
        for (int q = 0; q < K; ++q) x_out[g * K + q] = acc[q];
        if (inner) push_down<K>(a.push_tgt, a.push_ptr[n.front_off + j], a.push_ptr[n.front_off + j + 1], a.xb, acc);
    }
    if (inner && it.row0 == 0) {          // the boundary rows hand x down too
        for (int i = lane; i < n.b; i += 64) {
            const size_t f = (size_t)(n.front_off + n.s + i);
            float v[K];
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = a.xb[(size_t)(n.bnd_off + i) * K + q];
            push_down<K>(a.push_tgt, a.push_ptr[f], a.push_ptr[f + 1], a.xb, v);
        }
    }
}

// down: partial x_j = sum_{t in [r0, r1)} [Finv | -W^T][j][t] * [b' | x_bnd][t]
template <int K>
__device__ __forceinline__ void node_down(const TierArgs& a, const NodeD& n, const TierItem& it, float* __restrict__ x_out,
                                          float* region, float* pbuf) {
    const int lane = threadIdx.x & 63, s = n.s;
    const int j = it.row0 + lane;
    const bool row = j < s;
    float* sv4 = region;
    for (int t = it.r0 + lane; t < it.r1; t += 64) {
        float v[K];
        if (t < s) {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = a.bprime[(size_t)(n.own_start + t) * K + q];
        } else {
#pragma unroll
            for (int q = 0; q < K; ++q) v[q] = -a.xb[(size_t)(n.bnd_off + t - s) * K + q];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) sv4[(t - it.r0) * 4 + q] = v[q];
    }
    wave_lds_sync();
    float acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = 0.0f;
    if (row) {
        const int f0 = min(it.r0, s), f1 = min(it.r1, s), g0 = max(it.r0, s), g1 = max(it.r1, s);
        tier_dot<K>(a.finv + n.finv_off + j, (size_t)s, f0, f1, sv4, it.r0, acc);
        tier_dot<K>(a.wb + n.w_off + j - (size_t)s * s, (size_t)s, g0, g1, sv4, it.r0, acc);     // row t of wb is boundary row t - s
    }
    if (it.nparts == 1) node_down_finish<K>(a, n, it, x_out, acc);
    else {
#pragma unroll
        for (int q = 0; q < K; ++q) pbuf[lane * 4 + q] = acc[q];
    }
    wave_lds_sync();
}

// One workgroup per subtree. UP: phases run leaves -> tier root. DOWN: tier root -> leaves.
template <int K, bool UP>
__global__ __launch_bounds__(64 * TIER_WAVES) void k_nd_tier(TierArgs a, const float* __restrict__ b_in, float* __restrict__ x_out,
                                                              int tri_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* region = sm + (size_t)wave * a.region_floats;
    const TierWG& g = a.wgs[blockIdx.x];
    const int* off = UP ? g.up_off : g.down_off;
    const unsigned split = UP ? g.up_split : g.down_split;
    for (int ph = 0; ph < a.phases; ++ph) {
        const int i0 = off[ph], i1 = off[ph + 1];
        for (int k = i0 + wave; k < i1; k += TIER_WAVES) {
            const TierItem it = a.items[k];
            const NodeD n = a.nodes[it.node];
            float* pbuf = region + a.vec_floats + ((k - i0) / TIER_WAVES) * 256;
            if (n.flags & NODE_SPARSE) {
                if (UP) leaf_up<K>(a, n, b_in, region, tri_floats);
                else leaf_down<K>(a, n, x_out, region, tri_floats);
            } else {
                if (UP) node_up<K>(a, n, it, b_in, region, pbuf);
                else node_down<K>(a, n, it, x_out, region, pbuf);
            }
        }
        if ((split >> ph) & 1u) {
            __syncthreads();
            for (int k = i0 + wave; k < i1; k += TIER_WAVES) {
                const TierItem it = a.items[k];
                if (it.nparts == 1 || it.part != 0) continue;
                const NodeD n = a.nodes[it.node];
                float acc[K];
#pragma unroll
                for (int q = 0; q < K; ++q) acc[q] = 0.0f;
                for (int p = 0; p < it.nparts; ++p) {          // parts are consecutive items: fixed order of the sum
                    const int kp = k + p - i0;
                    const float* pb = sm + (size_t)(kp % TIER_WAVES) * a.region_floats + a.vec_floats + (kp / TIER_WAVES) * 256;
#pragma unroll
                    for (int q = 0; q < K; ++q) acc[q] += pb[lane * 4 + q];
                }
                if (UP) node_up_finish<K>(a, n, it.row0 + lane, acc);
                else node_down_finish<K>(a, n, it, x_out, acc);
            }
        }
        __syncthreads();      // workgroup-scope release/acquire of the slots / boundary vectors written above (same CU)
    }
}

}  // namespace ls
