// nd_bisect.hip -- the bisection rounds of the nested-dissection analysis ON THE DEVICE (gfx950, wave64).
//
// The host version (csrc/nd_plan.cpp, nd_plan_build without a callback) walks the domains of a round with host threads: per
// domain a bounding box, a median selection (std::nth_element over (coordinate, id) pairs), the end points of the cut edges, a
// compaction -- 14 rounds x ~4 ms at 1M vertices, two thirds of the solver's constructor (the reference's constructor is one
// call, largesteps/solvers.py:34; here it is paid at every remesh, scripts/main.py:137-169). None of it needs a host:
//
//   * the ORDER of the vertices along every axis is fixed once (stable radix sort by (coordinate, id), three lists); a domain
//     is a contiguous segment of each list, and splitting a domain is a STABLE PARTITION of its three segments -- the lists stay
//     sorted inside every segment, so in every later round the bounding box of a domain is its segments' first and last entries and
//     its median is the entry in the middle: no reduction, no selection;
//   * a round is eight launches over the live vertices: axis + half per domain, side of every vertex (from its position in the
//     split axis' list), cut-edge end points (+ per-domain counts of the two sides' end points: integer atomics), the smaller
//     end-point set becomes the separator and the sizes of the 2 x n_dom child segments are scanned (one workgroup), then a
//     packed two-counter scan per list (reduce / block offsets / scan + scatter) moves every surviving vertex to
//     next_start[child] + its rank among the survivors of its side -- the stable partition;
//   * nothing is read back between rounds (the number of live vertices is read from the segment table on the device; grids are
//     sized for V). One copy at the end: node[v], the binary heap id of the domain v ended in.
//
// The result is bit-identical to the host rounds (the same total order (coordinate, id), the same tie rules: first longest axis,
// smaller end-point set with ties to side 0): tests/test_nested_gpu.py compares node[] round by round and the finished plans.
#include "common.h"
#include "radix.h"
#include "nd_plan.h"
#include <chrono>
#include <string>
#include <vector>

namespace ls {
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int BI = 8;                       // list positions per thread in the regroup kernels
constexpr int BCH = BLOCK * BI;             // list positions per workgroup

constexpr int NAX = 6;                      // most candidate directions per domain (trial cuts: three position axes + three graph distances)
struct Round {
    int n_dom, NA;                          // NA = 3 lists (longest-axis rule, or trial cuts without positions) or 6 (trial cuts with positions)
    const int* seg;                         // [n_dom + 1] segment starts of this round (seg[n_dom] = live vertices)
    int* seg_next;                          // [2 n_dom + 1]
    const int* L[NAX];                      // the lists, grouped by domain, sorted by their axis inside a domain
    int* Ln[NAX];
    const double* pos;                      // V x 3: axes 0-2
    const double* pos2;                     // V x 3: axes 3-5 (NA == 6)
    long long* state;                       // (heap id << 2) | side << 1 | fixed
    unsigned char* endp;
    int* ax; int* half; int* ecnt; int* use1; int* base0; int* base1;
    unsigned long long* bsum;               // [NA][nb] packed block sums / offsets
    int nb;
    // trial cuts (ND_ORDER_MINSEP): bit k of sidebits[u] = side of vertex u under direction k; bit k of cutbits[u] = u is an end point of
    // a cut edge under direction k; ecnt6[(d * NA + k) * 2 + side] = end points per (domain, direction, side)
    unsigned* sidebits; unsigned char* cutbits; int* ecnt6;
    __device__ __forceinline__ double coord(int u, int k) const { return (k < 3 ? pos : pos2)[3 * (size_t)u + (k < 3 ? k : k - 3)]; }
};

__global__ __launch_bounds__(BLOCK) void k_f32_to_f64(const float* __restrict__ in, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// one averaging pass over the matrix neighbours, the host's arithmetic to the letter (nd_plan.cpp "positions": sums in CSR order,
// then times 1.0 / count; the library is built with -ffp-contract=off)
__global__ __launch_bounds__(BLOCK) void k_smooth(const int* __restrict__ rowptr, const int* __restrict__ col, int64_t V,
                                                  const double* __restrict__ pos, double* __restrict__ nxt) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    double a = 0.0, b = 0.0, c = 0.0;
    const int p0 = rowptr[v], p1 = rowptr[v + 1];
    for (int p = p0; p < p1; ++p) {
        const size_t w = (size_t)col[p];
        a += pos[3 * w]; b += pos[3 * w + 1]; c += pos[3 * w + 2];
    }
    const double inv = 1.0 / (double)(p1 - p0);
    nxt[3 * v] = a * inv; nxt[3 * v + 1] = b * inv; nxt[3 * v + 2] = c * inv;
}

__global__ __launch_bounds__(BLOCK) void k_init(int64_t V, long long* __restrict__ state, int* __restrict__ seg) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v < V) state[v] = 4;                                        // heap id 1, side 0, live
    if (v == 0) { seg[0] = 0; seg[1] = (int)V; }
}

// per domain: the longest axis of its bounding box (first / last entry of its three segments), half of its size
__global__ __launch_bounds__(BLOCK) void k_axis(Round r) {
    const int d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= r.n_dom) return;
    const int a = r.seg[d], e = r.seg[d + 1], cnt = e - a;
    r.ecnt[2 * d] = 0; r.ecnt[2 * d + 1] = 0;
    r.half[d] = cnt / 2;
    int ax = 0;
    if (cnt > 0) {
        double ext[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ext[k] = r.pos[3 * (size_t)r.L[k][e - 1] + k] - r.pos[3 * (size_t)r.L[k][a] + k];
        for (int k = 1; k < 3; ++k) if (ext[k] > ext[ax]) ax = k;
    }
    r.ax[d] = ax;
}

// side of every live vertex: its position in the list of its domain's split axis (blockIdx.y = list)
__global__ __launch_bounds__(BLOCK) void k_side(Round r) {
    const int k = blockIdx.y, i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= r.seg[r.n_dom]) return;
    const int u = r.L[k][i];
    const long long node = r.state[u] >> 2;
    const int d = (int)(node - r.n_dom);
    if (r.ax[d] != k) return;
    const long long s = (i - r.seg[d]) >= r.half[d];
    r.state[u] = (node << 2) | (s << 1);
}

// end points of the cut edges (a neighbour in the same domain, live, on the other side) and their count per (domain, side)
__global__ __launch_bounds__(BLOCK) void k_cut(Round r, const int* __restrict__ rowptr, const int* __restrict__ col) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= r.seg[r.n_dom]) return;
    const int u = r.L[0][i];
    const long long mine = r.state[u];
    bool cut = false;
    const int p1 = rowptr[u + 1];
    for (int p = rowptr[u]; p < p1 && !cut; ++p) cut = (r.state[col[p]] ^ mine) == 2;
    r.endp[u] = cut;
    if (cut) atomicAdd(&r.ecnt[2 * (int)((mine >> 2) - r.n_dom) + (int)((mine >> 1) & 1)], 1);
}

// ---- trial cuts (ND_ORDER_MINSEP; the host statement: nd_plan.cpp, "every candidate axis of every domain is TRIED") ---------------------
// Every domain tries all NA directions -- a median split of its segment of that direction's list, the end points of the edges the split
// cuts counted per side -- and takes the direction whose smaller end-point set is smallest (ties: the lower direction; a direction that
// is constant over the domain would order by vertex id: never). With the lists sorted once, a trial is a bit per vertex and direction.
__global__ __launch_bounds__(BLOCK) void k_trial_init(Round r) {
    const int d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= r.n_dom) return;
    r.half[d] = (r.seg[d + 1] - r.seg[d]) / 2;
    for (int t = 0; t < 2 * r.NA; ++t) r.ecnt6[(size_t)d * 2 * r.NA + t] = 0;
}
__global__ __launch_bounds__(BLOCK) void k_trial_side(Round r) {          // grid (vertices, NA): list position -> side bit
    const int k = blockIdx.y, i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= r.seg[r.n_dom]) return;
    const int u = r.L[k][i];
    const int d = (int)((r.state[u] >> 2) - r.n_dom);
    if ((i - r.seg[d]) >= r.half[d]) atomicOr(&r.sidebits[u], 1u << k);
}
__global__ __launch_bounds__(BLOCK) void k_trial_cut(Round r, const int* __restrict__ rowptr, const int* __restrict__ col) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= r.seg[r.n_dom]) return;
    const int u = r.L[0][i];
    const long long node = r.state[u] >> 2;
    const unsigned mine = r.sidebits[u];
    unsigned diff = 0;
    const int p1 = rowptr[u + 1];
    for (int p = rowptr[u]; p < p1; ++p) {
        const int w = col[p];
        const long long sw = r.state[w];
        if ((sw >> 2) == node && !(sw & 1)) diff |= r.sidebits[w] ^ mine;          // same domain, live
    }
    diff &= (1u << r.NA) - 1u;
    r.cutbits[u] = (unsigned char)diff;
    const int d = (int)(node - r.n_dom);
    for (int k = 0; k < r.NA; ++k)
        if ((diff >> k) & 1u) atomicAdd(&r.ecnt6[((size_t)d * r.NA + k) * 2 + ((mine >> k) & 1u)], 1);
}
__global__ __launch_bounds__(BLOCK) void k_trial_pick(Round r) {
    const int d = blockIdx.x * BLOCK + threadIdx.x;
    if (d >= r.n_dom) return;
    const int a = r.seg[d], e = r.seg[d + 1], cnt = e - a;
    int best = 0, best_sep = 0;
    for (int k = 0; k < r.NA; ++k) {
        int sep = 0;
        if (cnt > 0) {
            const double lo = r.coord(r.L[k][a], k), hi = r.coord(r.L[k][e - 1], k);
            if (!(hi > lo) && cnt > 1) sep = INT32_MAX;
            else { const int e0 = r.ecnt6[((size_t)d * r.NA + k) * 2], e1 = r.ecnt6[((size_t)d * r.NA + k) * 2 + 1]; sep = e0 < e1 ? e0 : e1; }
        }
        if (k == 0 || sep < best_sep) { best = k; best_sep = sep; }
    }
    r.ax[d] = best;
    r.ecnt[2 * d] = r.ecnt6[((size_t)d * r.NA + best) * 2];
    r.ecnt[2 * d + 1] = r.ecnt6[((size_t)d * r.NA + best) * 2 + 1];
}
__global__ __launch_bounds__(BLOCK) void k_trial_apply(Round r) {          // the picked direction's side and end-point flag become the round's
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= r.seg[r.n_dom]) return;
    const int u = r.L[0][i];
    const long long node = r.state[u] >> 2;
    const int k = r.ax[(int)(node - r.n_dom)];
    const long long s = (r.sidebits[u] >> k) & 1u;
    r.state[u] = (node << 2) | (s << 1);
    r.endp[u] = (r.cutbits[u] >> k) & 1u;
}

// one workgroup: which side gives the separator, the sizes of the child segments and their scans:
// base0[d] / base1[d] = survivors on side 0 / 1 in the domains before d; seg_next[2 d] = base0 + base1, seg_next[2 d + 1] = + keep0[d]
constexpr int PLAN_T = 1024;
__global__ __launch_bounds__(PLAN_T) void k_plan(Round r) {
    __shared__ int tot0[PLAN_T], tot1[PLAN_T];
    const int t = threadIdx.x, per = (r.n_dom + PLAN_T - 1) / PLAN_T, d0 = t * per, d1 = min(r.n_dom, d0 + per);
    int s0 = 0, s1 = 0;
    for (int d = d0; d < d1; ++d) {
        const int cnt = r.seg[d + 1] - r.seg[d], n0 = r.half[d], n1 = cnt - n0, e0 = r.ecnt[2 * d], e1 = r.ecnt[2 * d + 1];
        const int u1 = e1 < e0;
        r.use1[d] = u1;
        s0 += n0 - (u1 ? 0 : e0);
        s1 += n1 - (u1 ? e1 : 0);
    }
    tot0[t] = s0; tot1[t] = s1;
    __syncthreads();
    if (t == 0) {                                                   // 1024 entries: a serial scan is microseconds
        int a = 0, b = 0;
        for (int j = 0; j < PLAN_T; ++j) { const int x = tot0[j], y = tot1[j]; tot0[j] = a; tot1[j] = b; a += x; b += y; }
    }
    __syncthreads();
    int b0 = tot0[t], b1 = tot1[t];
    for (int d = d0; d < d1; ++d) {
        const int cnt = r.seg[d + 1] - r.seg[d], n0 = r.half[d], n1 = cnt - n0, e0 = r.ecnt[2 * d], e1 = r.ecnt[2 * d + 1];
        const int u1 = e1 < e0, k0 = n0 - (u1 ? 0 : e0), k1 = n1 - (u1 ? e1 : 0);
        r.base0[d] = b0; r.base1[d] = b1;
        r.seg_next[2 * d] = b0 + b1;
        r.seg_next[2 * d + 1] = b0 + b1 + k0;
        b0 += k0; b1 += k1;
        if (d == r.n_dom - 1) r.seg_next[2 * r.n_dom] = b0 + b1;
    }
}

// what happens to list position i: 0 = leaves (separator), 1 / 2 = survives on side 0 / 1 -- packed as two 32-bit counters
__device__ __forceinline__ unsigned long long survivor(const Round& r, int u, int& d, int& side) {
    const long long st = r.state[u];
    d = (int)((st >> 2) - r.n_dom);
    side = (int)((st >> 1) & 1);
    const bool sep = r.endp[u] && side == r.use1[d];
    return sep ? 0ull : (side ? (1ull << 32) : 1ull);
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
    return x;
}

__global__ __launch_bounds__(BLOCK) void k_regroup_reduce(Round r) {
    __shared__ unsigned long long sm[BLOCK / WAVE];
    const int k = blockIdx.y, n_live = r.seg[r.n_dom];
    const int i0 = blockIdx.x * BCH + threadIdx.x * BI;
    unsigned long long acc = 0;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int i = i0 + j;
        if (i < n_live) { int d, s; acc += survivor(r, r.L[k][i], d, s); }
    }
    acc = wave_sum64(acc);
    if ((threadIdx.x & (WAVE - 1)) == 0) sm[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int j = 0; j < BLOCK / WAVE; ++j) t += sm[j];
        r.bsum[(size_t)k * r.nb + blockIdx.x] = t;
    }
}

// exclusive scan of the block sums of one list (one workgroup per list)
__global__ __launch_bounds__(PLAN_T) void k_regroup_offsets(Round r) {
    __shared__ unsigned long long tot[PLAN_T];
    unsigned long long* b = r.bsum + (size_t)blockIdx.x * r.nb;
    const int t = threadIdx.x, per = (r.nb + PLAN_T - 1) / PLAN_T, j0 = t * per, j1 = min(r.nb, j0 + per);
    unsigned long long s = 0;
    for (int j = j0; j < j1; ++j) s += b[j];
    tot[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long a = 0;
        for (int j = 0; j < PLAN_T; ++j) { const unsigned long long x = tot[j]; tot[j] = a; a += x; }
    }
    __syncthreads();
    unsigned long long run = tot[t];
    for (int j = j0; j < j1; ++j) { const unsigned long long x = b[j]; b[j] = run; run += x; }
}

// the stable partition: survivor of side s of domain d at list position i moves to seg_next[2 d + s] + (survivors of side s of d
// before i). List 0's thread of a vertex also writes its new state (the child's heap id, or the fixed bit of a separator vertex)
// while the threads of the other lists need the old one: the host launches lists 1 and 2 first (k0 = 1), list 0 after them.
__global__ __launch_bounds__(BLOCK) void k_regroup_scatter(Round r, int k0) {
    __shared__ unsigned long long sm[BLOCK / WAVE + 1];
    const int k = k0 + blockIdx.y, n_live = r.seg[r.n_dom];
    const int i0 = blockIdx.x * BCH + threadIdx.x * BI;
    int u[BI], d[BI], sd[BI];
    unsigned long long f[BI], mine = 0;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int i = i0 + j;
        u[j] = 0; d[j] = 0; sd[j] = 0; f[j] = 0;
        if (i < n_live) { u[j] = r.L[k][i]; f[j] = survivor(r, u[j], d[j], sd[j]); }
        mine += f[j];
    }
    // exclusive scan of `mine` over the workgroup
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    unsigned long long inc = mine;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const unsigned long long y = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += y;
    }
    if (lane == WAVE - 1) sm[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        for (int j = 0; j < BLOCK / WAVE; ++j) { const unsigned long long x = sm[j]; sm[j] = a; a += x; }
    }
    __syncthreads();
    unsigned long long run = r.bsum[(size_t)k * r.nb + blockIdx.x] + sm[w] + (inc - mine);
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int i = i0 + j;
        if (i < n_live) {
            if (f[j]) {
                const int rank = sd[j] ? (int)(run >> 32) - r.base1[d[j]] : (int)(run & 0xffffffffull) - r.base0[d[j]];
                r.Ln[k][r.seg_next[2 * d[j] + sd[j]] + rank] = u[j];
            }
            if (k == 0) {
                const long long st = r.state[u[j]];
                r.state[u[j]] = f[j] ? (((st >> 2) * 2 + sd[j]) << 2) : (st | 1);
            }
        }
        run += f[j];
    }
}

__global__ __launch_bounds__(BLOCK) void k_nodes(int64_t V, const long long* __restrict__ state, long long* __restrict__ node) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v < V) node[v] = state[v] >> 2;
}

}  // namespace

struct NdBisectDevice {
    const int32_t* d_rowptr; const int32_t* d_col; const float* d_pos; int64_t nnz; hipStream_t st;
    int32_t* h_col_pending;                 // not nullptr: the host copy of the column indices is still to be made -- WHILE the rounds run
    double* d_emb = nullptr;                // graph embedding formed ON THE DEVICE (nd_embed_device): nd_bisect_device reads it instead of `embedded`
    size_t d_emb_cap = 0;
    ~NdBisectDevice() {
        if (d_emb) { int dev = 0; (void)hipGetDevice(&dev); (void)hipStreamSynchronize(st); if (!pool_give(dev, d_emb, d_emb_cap)) (void)hipFree(d_emb); }
    }
};

// ---- the graph embedding on the device (round 6) -------------------------------------------------------------------------------------
// nd_plan.cpp's graph_embedding -- per connected component four breadth-first sweeps (seed -> p1 -> p2, p3 = far from both) -- took
// 70-80 ms of host time on a 1M-vertex closed scan (single thread, ~1000 levels of a few thousand vertices each), half of that mesh's
// constructor. A sweep is a chain of ~1000 dependent steps however many threads walk a level, so ONE workgroup runs many levels per
// launch (__syncthreads() between them, the frontier in two global queues, no launch and no grid barrier per level): ~4 us per level.
// Same numbers as the host's: distances are exact levels, "the last vertex reached" is the SMALLEST index of the last level, p3 the
// smallest index among the maxima of min(d1, d2).
namespace {
// One launch per level, a few dozen workgroups each: a level of a 1M-vertex surface mesh is ~1000 vertices = ~15 000 scattered addresses
// (row pointers, column indices, distances, claims) -- ONE workgroup walking it is bound by its CU's address rate (measured: 14 us per
// level as a single persistent workgroup with __syncthreads() between levels, 53 ms for the four sweeps; unrolled variants with more
// loads in flight 70-75 ms); spread over the chip a level costs about a dependent launch. cnt[l] = size of level l (zeroed per sweep),
// queues alternate; the host looks at the counters every BFS_CHUNK levels.
constexpr int BFS_CHUNK = 128, BFS_MAX_LEVELS = 60000, BFS_WG = 64;

__global__ __launch_bounds__(BLOCK) void k_bfs_init(int64_t V, int* __restrict__ dist, int start, int* __restrict__ q0, int* __restrict__ cnt, int n_cnt) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < V) dist[i] = i == start ? 0 : -1;
    if (i < n_cnt) cnt[i] = i == 0 ? 1 : 0;
    if (i == 0) q0[0] = start;
}

__global__ __launch_bounds__(BLOCK) void k_bfs_level(const int* __restrict__ rowptr, const int* __restrict__ col, int* dist, const int* __restrict__ qi,
                                                     int* __restrict__ qo, int* cnt, int level) {
    const int n = cnt[level];
    const int lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * BLOCK; i0 < n; i0 += gridDim.x * BLOCK) {
        const int i = i0 + (int)threadIdx.x;
        const int u = i < n ? qi[i] : -1;
        const int p0 = u >= 0 ? rowptr[u] : 0, p1 = u >= 0 ? rowptr[u + 1] : 0;
        int deg = p1 - p0;
        for (int o = 32; o > 0; o >>= 1) deg = max(deg, __shfl_xor(deg, o));        // the wave walks its longest row together (aggregated claims)
        for (int j = 0; j < deg; ++j) {
            const int w = p0 + j < p1 ? col[p0 + j] : -1;
            const bool win = w >= 0 && dist[w] < 0 && atomicCAS(&dist[w], -1, level + 1) == -1;
            const unsigned long long m = __ballot(win);
            if (m) {                                               // one counter update per wave
                const int leader = __ffsll((long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&cnt[level + 1], __popcll(m));
                base = __shfl(base, leader);
                if (win) qo[base + __popcll(m & ((1ull << lane) - 1ull))] = w;
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_bfs_last(const int* __restrict__ q, int n, int* out) {
    int v = 0x7fffffff;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) v = min(v, q[i]);
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) atomicMin(out, v);
}

// p3 = the smallest index among the maxima of min(d1, d2) over the component (d0 >= 0): packed (value << 32 | ~index) maximum
__global__ __launch_bounds__(BLOCK) void k_far_from_both(int64_t V, const int* __restrict__ d0, const int* __restrict__ d1, const int* __restrict__ d2,
                                                         unsigned long long* best) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long key = 0;
    if (v < V && d0[v] >= 0) key = ((unsigned long long)(unsigned)min(d1[v], d2[v]) << 32) | (unsigned)(0x7fffffff - (int)v);
    // wave maximum first: one atomic per wave
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other > key ? other : key; }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(best, key);
}

__global__ __launch_bounds__(BLOCK) void k_emb_write(int64_t V, const int* __restrict__ d0, const int* __restrict__ d1, const int* __restrict__ d2,
                                                     const int* __restrict__ d3, double offset, double* __restrict__ emb, unsigned char* __restrict__ done) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v < V && d0[v] >= 0) {
        emb[3 * v] = (double)d1[v] + offset; emb[3 * v + 1] = (double)d2[v]; emb[3 * v + 2] = (double)d3[v];
        done[v] = 1;
    }
}

__global__ __launch_bounds__(BLOCK) void k_first_undone(int64_t V, const unsigned char* __restrict__ done, int* first) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v < V && !done[v]) atomicMin(first, (int)v);
}
}  // namespace

// "" = the embedding is in ctx->d_emb (V x 3 doubles, device); "host" = this graph is one for the host's sweeps (a swarm of isolated
// vertices, a path-like graph of tens of thousands of levels); anything else = an error text.
static std::string nd_embed_device_impl(void* ctx, int64_t V);
std::string nd_embed_device(void* ctx, int64_t V) {
    NdBisectDevice& A = *(NdBisectDevice*)ctx;
    if (getenv("LS_ND_HOST_EMBED")) { }                    // A/B and tests: the host's sweeps
    else {
        const std::string r = nd_embed_device_impl(ctx, V);
        if (r != "host") return r;
        if (A.d_emb) { int dev = 0; (void)hipGetDevice(&dev); (void)hipStreamSynchronize(A.st); if (!pool_give(dev, A.d_emb, A.d_emb_cap)) (void)hipFree(A.d_emb); A.d_emb = nullptr; }
    }
    // the host walks the pattern itself: it needs the column indices NOW
    if (A.h_col_pending) {
        if (hipMemcpyAsync(A.h_col_pending, A.d_col, sizeof(int32_t) * A.nnz, hipMemcpyDeviceToHost, A.st) != hipSuccess || hipStreamSynchronize(A.st) != hipSuccess)
            return "nd_embed_device: copy of the pattern failed";
        A.h_col_pending = nullptr;
    }
    return "host";
}
static std::string nd_embed_device_impl(void* ctx, int64_t V) {
    NdBisectDevice& A = *(NdBisectDevice*)ctx;
    hipStream_t st = A.st;
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_d[4];
    for (int k = 0; k < 4; ++k) o_d[k] = take(sizeof(int) * V);
    const int n_cnt = BFS_MAX_LEVELS + 2 * BFS_CHUNK + 8;
    const size_t o_q0 = take(sizeof(int) * V), o_q1 = take(sizeof(int) * V), o_done = take(V), o_cnt = take(sizeof(int) * n_cnt), o_best = take(16), o_first = take(16);
    size_t cap = off, ecap = sizeof(double) * 3 * (size_t)V;
    char* base = (char*)pool_take(dev, off, &cap);
    if (!base && pool_alloc(dev, (void**)&base, off) != hipSuccess) return "nd_embed_device: out of device memory";
    struct Free { char* p; size_t bytes; int dev; hipStream_t st; ~Free() { (void)hipStreamSynchronize(st); if (!pool_give(dev, p, bytes)) (void)hipFree(p); } } guard{base, cap, dev, st};
    double* emb = (double*)pool_take(dev, ecap, &ecap);
    if (!emb && pool_alloc(dev, (void**)&emb, ecap) != hipSuccess) return "nd_embed_device: out of device memory";
    A.d_emb = emb; A.d_emb_cap = ecap;               // (owned by the context from here on, whatever this function returns)
    int* d[4];
    for (int k = 0; k < 4; ++k) d[k] = (int*)(base + o_d[k]);
    int* q[2] = {(int*)(base + o_q0), (int*)(base + o_q1)};
    unsigned char* done = (unsigned char*)(base + o_done);
    int* cnt = (int*)(base + o_cnt);
    unsigned long long* best = (unsigned long long*)(base + o_best);
    int* first = (int*)(base + o_first);
    const unsigned gv = (unsigned)div_up(std::max<int64_t>(V, n_cnt), BLOCK);
    bool failed = false, to_host = false;
    struct Sweep { int levels, last; int64_t visited; };           // levels = index of the last non-empty level
    std::vector<int> hc((size_t)BFS_CHUNK + 1);
    auto sweep = [&](int start, int* dist, Sweep& h) {              // one breadth-first sweep
        hipLaunchKernelGGL(k_bfs_init, dim3(gv), dim3(BLOCK), 0, st, V, dist, start, q[0], cnt, n_cnt);
        h.visited = 1; h.levels = 0; h.last = start;
        for (int lv = 0;; lv += BFS_CHUNK) {
            for (int l = lv; l < lv + BFS_CHUNK; ++l)
                hipLaunchKernelGGL(k_bfs_level, dim3(BFS_WG), dim3(BLOCK), 0, st, A.d_rowptr, A.d_col, dist, (const int*)q[l & 1], q[(l + 1) & 1], cnt, l);
            // sizes of levels lv + 1 .. lv + BFS_CHUNK
            if (hipMemcpyAsync(hc.data(), cnt + lv + 1, sizeof(int) * BFS_CHUNK, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { failed = true; return; }
            bool ended = false;
            for (int t = 0; t < BFS_CHUNK; ++t) {
                if (hc[(size_t)t] == 0) { ended = true; break; }
                h.visited += hc[(size_t)t]; h.levels = lv + 1 + t;
            }
            if (ended) break;
            if (lv + BFS_CHUNK > BFS_MAX_LEVELS) { to_host = true; return; }
        }
        // the smallest index of the last level (its queue is intact: the launches behind it found it empty and wrote nothing)
        int n_last = 1;
        if (h.levels > 0) {
            if (hipMemcpyAsync(&n_last, cnt + h.levels, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { failed = true; return; }
        }
        int init = 0x7fffffff;
        if (hipMemcpyAsync(first, &init, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { failed = true; return; }
        hipLaunchKernelGGL(k_bfs_last, dim3(8), dim3(BLOCK), 0, st, (const int*)q[h.levels & 1], n_last, first);
        if (hipMemcpyAsync(&h.last, first, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { failed = true; return; }
    };
    if (hipMemsetAsync(done, 0, (size_t)V, st) != hipSuccess || hipMemsetAsync(emb, 0, sizeof(double) * 3 * (size_t)V, st) != hipSuccess) return "nd_embed_device: memset failed";
    double offset = 0.0;
    int64_t remaining = V;
    int seed = 0;
    while (remaining > 0) {
        Sweep h0, h1, h2, h3;
        sweep(seed, d[0], h0);
        if (failed) return "nd_embed_device: a device call failed";
        if (to_host) return "host";
        const int64_t comp = h0.visited;
        if (comp <= 2 && remaining - comp > 4096) return "host";       // a swarm of isolated vertices: the host lays it out in index order
        sweep(h0.last, d[1], h1);
        if (!failed && !to_host) sweep(h1.last, d[2], h2);
        if (failed) return "nd_embed_device: a device call failed";
        if (to_host) return "host";
        unsigned long long hbest = 0;
        if (hipMemsetAsync(best, 0, sizeof(unsigned long long), st) != hipSuccess) return "nd_embed_device: memset failed";
        hipLaunchKernelGGL(k_far_from_both, dim3(gv), dim3(BLOCK), 0, st, V, (const int*)d[0], (const int*)d[1], (const int*)d[2], best);
        if (hipMemcpyAsync(&hbest, best, sizeof(hbest), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return "nd_embed_device: a device call failed";
        const int p3 = hbest ? 0x7fffffff - (int)(unsigned)(hbest & 0xffffffffu) : seed;
        sweep(p3, d[3], h3);
        if (failed) return "nd_embed_device: a device call failed";
        if (to_host) return "host";
        hipLaunchKernelGGL(k_emb_write, dim3(gv), dim3(BLOCK), 0, st, V, (const int*)d[0], (const int*)d[1], (const int*)d[2], (const int*)d[3], offset, emb, done);
        offset += (double)h1.levels + 2.0;                // (the largest d1 of the component = the last level of the sweep from p1)
        remaining -= comp;
        if (remaining > 0) {
            int hfirst = 0x7fffffff;
            if (hipMemcpyAsync(first, &hfirst, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return "nd_embed_device: a device call failed";
            hipLaunchKernelGGL(k_first_undone, dim3(gv), dim3(BLOCK), 0, st, V, (const unsigned char*)done, first);
            if (hipMemcpyAsync(&hfirst, first, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return "nd_embed_device: a device call failed";
            if (hfirst < 0 || hfirst >= V) return "nd_embed_device: inconsistent component count";
            seed = hfirst;
        }
    }
    if (hipGetLastError() != hipSuccess) return "nd_embed_device: a kernel launch failed";
    return "";
}

// ordering = ND_ORDER_LONGEST: the longest axis of the caller's positions (averaged `smooth` times), or of `embedded` as it is.
// ordering = ND_ORDER_MINSEP:  trial cuts over the position axes AND the three graph distances of `embedded` (both averaged `smooth`
//                              times), or over the embedding alone when there are no positions.
std::string nd_bisect_device(void* ctx, int64_t V, int D, int smooth, const double* embedded, int64_t* node, int ordering) {
    const NdBisectDevice& A = *(const NdBisectDevice*)ctx;
    hipStream_t st = A.st;
    if (D > 24) return "nd_bisect_device: more than 24 bisection rounds";
    if (D == 0) {                              // one domain: nothing to split (the host still gets its copy of the pattern)
        for (int64_t v = 0; v < V; ++v) node[v] = 1;
        if (A.h_col_pending && (hipMemcpyAsync(A.h_col_pending, A.d_col, sizeof(int32_t) * A.nnz, hipMemcpyDeviceToHost, st) != hipSuccess ||
                                hipStreamSynchronize(st) != hipSuccess))
            return "nd_bisect_device: copy of the pattern failed";
        return "";
    }
    const bool minsep = ordering == ND_ORDER_MINSEP, has_pos = A.d_pos != nullptr;
    if (A.d_emb) embedded = (const double*)A.d_emb;          // formed on the device (nd_embed_device): only ever used as a device-to-device source below
    if (!has_pos && !embedded) return "nd_bisect_device: neither positions nor an embedding";
    const int NA = (minsep && has_pos && embedded) ? 6 : 3;
    const int max_dom = 1 << (D - 1), nb = div_up(V, BCH);
    const size_t nbr = (size_t)div_up(V, rs_chunk(V));
    // one allocation: positions (two copies), 3 x 2 lists, state, node, end-point flags, per-domain tables, scan scratch, sort scratch
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pos = take(sizeof(double) * 3 * V), o_pos2 = take(sizeof(double) * 3 * V);
    const size_t o_pos3 = NA == 6 ? take(sizeof(double) * 3 * V) : 0;
    size_t o_L[2 * NAX];
    for (int k = 0; k < 2 * NA; ++k) o_L[k] = take(sizeof(int) * V);
    const size_t o_state = take(sizeof(long long) * V), o_node = take(sizeof(long long) * V), o_endp = take(V);
    const size_t o_seg0 = take(sizeof(int) * (2 * (size_t)max_dom + 2)), o_seg1 = take(sizeof(int) * (2 * (size_t)max_dom + 2));
    const size_t o_ax = take(sizeof(int) * max_dom), o_half = take(sizeof(int) * max_dom), o_ecnt = take(sizeof(int) * 2 * (size_t)max_dom),
                 o_use1 = take(sizeof(int) * max_dom), o_b0 = take(sizeof(int) * max_dom), o_b1 = take(sizeof(int) * max_dom);
    const size_t o_bsum = take(sizeof(unsigned long long) * NA * (size_t)nb);
    const size_t o_sideb = minsep ? take(sizeof(unsigned) * V) : 0, o_cutb = minsep ? take(V) : 0, o_ecnt6 = minsep ? take(sizeof(int) * 2 * NAX * (size_t)max_dom) : 0;
    const size_t o_keys = take(sizeof(unsigned) * 2 * (size_t)V);            // carried key words of the coordinate sorts
    const size_t o_hist = take(sizeof(int) * (256 * nbr + 16)), o_offs = take(sizeof(int) * (256 * nbr + 16)),
                 o_sb = take(sizeof(int) * ((size_t)scan_blocks(256 * (int64_t)nbr) + 64));
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t cap = off;
    char* base = (char*)pool_take(dev, off, &cap);                  // (the pool of large buffers of direct.hip: a remesh loop comes back with the same size)
    if (!base && pool_alloc(dev, (void**)&base, off) != hipSuccess) return "nd_bisect_device: out of device memory";
    struct Free {
        char* p; size_t bytes; int dev; hipStream_t st;
        ~Free() { (void)hipStreamSynchronize(st); if (!pool_give(dev, p, bytes)) (void)hipFree(p); }
    } guard{base, cap, dev, st};
    double* pos = (double*)(base + o_pos);
    double* pos2 = (double*)(base + o_pos2);
    int* L[2 * NAX];
    for (int k = 0; k < 2 * NA; ++k) L[k] = (int*)(base + o_L[k]);
    long long* state = (long long*)(base + o_state);
    long long* d_node = (long long*)(base + o_node);
    int* seg[2] = {(int*)(base + o_seg0), (int*)(base + o_seg1)};
    // ---- positions ------------------------------------------------------------------------------------------------------------------
    // axes 0-2: the caller's positions, averaged; without positions the embedding (averaged under the trial-cut rule only: the plain
    // embedding is what the longest-axis rounds are pinned to). Axes 3-5 (NA == 6): the embedding, averaged.
    double* posB = NA == 6 ? (double*)(base + o_pos3) : nullptr;
    auto smooth_passes = [&](double*& p, double*& scratch, int passes) {
        for (int it = 0; it < passes; ++it) {
            hipLaunchKernelGGL(k_smooth, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, A.d_rowptr, A.d_col, V, (const double*)p, scratch);
            std::swap(p, scratch);
        }
    };
    if (has_pos && !(embedded && !minsep)) {
        hipLaunchKernelGGL(k_f32_to_f64, dim3(div_up(3 * V, BLOCK)), dim3(BLOCK), 0, st, A.d_pos, 3 * V, pos);
        smooth_passes(pos, pos2, smooth);
        if (NA == 6) {
            if (hipMemcpyAsync(posB, A.d_emb ? (const void*)A.d_emb : (const void*)embedded, sizeof(double) * 3 * V, A.d_emb ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) != hipSuccess)
                return "nd_bisect_device: copy failed";
            smooth_passes(posB, pos2, smooth);            // (pos2 is scratch from here on)
        }
    } else {
        if (hipMemcpyAsync(pos, A.d_emb ? (const void*)A.d_emb : (const void*)embedded, sizeof(double) * 3 * V, A.d_emb ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st) != hipSuccess)
            return "nd_bisect_device: copy failed";
        if (minsep) smooth_passes(pos, pos2, smooth);
    }
    // ---- the three coordinate orders: stable LSD radix sort of the ids by the 8 key bytes -> (coordinate, id) order -----------------
    const int* Lc[NAX];
    int* Lo[NAX];
    for (int k = 0; k < NA; ++k) {
        const int* res = nullptr;
        const int rc = radix_argsort_words(KeyF64{k < 3 ? pos : posB, k < 3 ? k : k - 3}, V, 2, L[2 * k], L[2 * k + 1], (unsigned*)(base + o_keys), (unsigned*)(base + o_keys) + V, (int*)(base + o_hist),
                                           (int*)(base + o_offs), (int*)(base + o_sb), st, &res);
        if (rc) return "nd_bisect_device: sort failed";
        Lc[k] = res;
        Lo[k] = res == L[2 * k] ? L[2 * k + 1] : L[2 * k];
    }
    hipLaunchKernelGGL(k_init, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, V, state, seg[0]);
    // ---- D rounds ---------------------------------------------------------------------------------------------------------------------
    for (int r = 0; r < D; ++r) {
        Round R;
        R.n_dom = 1 << r; R.NA = NA;
        R.seg = seg[r & 1]; R.seg_next = seg[(r + 1) & 1];
        for (int k = 0; k < NAX; ++k) { R.L[k] = k < NA ? Lc[k] : nullptr; R.Ln[k] = k < NA ? Lo[k] : nullptr; }
        R.pos = pos; R.pos2 = posB; R.state = state; R.endp = (unsigned char*)(base + o_endp);
        R.sidebits = minsep ? (unsigned*)(base + o_sideb) : nullptr; R.cutbits = minsep ? (unsigned char*)(base + o_cutb) : nullptr;
        R.ecnt6 = minsep ? (int*)(base + o_ecnt6) : nullptr;
        R.ax = (int*)(base + o_ax); R.half = (int*)(base + o_half); R.ecnt = (int*)(base + o_ecnt); R.use1 = (int*)(base + o_use1);
        R.base0 = (int*)(base + o_b0); R.base1 = (int*)(base + o_b1);
        R.bsum = (unsigned long long*)(base + o_bsum); R.nb = nb;
        const int gv = div_up(V, BLOCK);
        if (minsep) {
            if (hipMemsetAsync(R.sidebits, 0, sizeof(unsigned) * V, st) != hipSuccess) return "nd_bisect_device: memset failed";
            hipLaunchKernelGGL(k_trial_init, dim3(div_up(R.n_dom, BLOCK)), dim3(BLOCK), 0, st, R);
            hipLaunchKernelGGL(k_trial_side, dim3(gv, NA), dim3(BLOCK), 0, st, R);
            hipLaunchKernelGGL(k_trial_cut, dim3(gv), dim3(BLOCK), 0, st, R, A.d_rowptr, A.d_col);
            hipLaunchKernelGGL(k_trial_pick, dim3(div_up(R.n_dom, BLOCK)), dim3(BLOCK), 0, st, R);
            hipLaunchKernelGGL(k_trial_apply, dim3(gv), dim3(BLOCK), 0, st, R);
        } else {
            hipLaunchKernelGGL(k_axis, dim3(div_up(R.n_dom, BLOCK)), dim3(BLOCK), 0, st, R);
            hipLaunchKernelGGL(k_side, dim3(gv, 3), dim3(BLOCK), 0, st, R);
            hipLaunchKernelGGL(k_cut, dim3(gv), dim3(BLOCK), 0, st, R, A.d_rowptr, A.d_col);
        }
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(PLAN_T), 0, st, R);
        hipLaunchKernelGGL(k_regroup_reduce, dim3(nb, NA), dim3(BLOCK), 0, st, R);
        hipLaunchKernelGGL(k_regroup_offsets, dim3(NA), dim3(PLAN_T), 0, st, R);
        // lists 1 .. NA - 1 first (their threads read the old states), then list 0, whose threads write the new ones
        hipLaunchKernelGGL(k_regroup_scatter, dim3(nb, NA - 1), dim3(BLOCK), 0, st, R, 1);
        hipLaunchKernelGGL(k_regroup_scatter, dim3(nb, 1), dim3(BLOCK), 0, st, R, 0);
        for (int k = 0; k < NA; ++k) { const int* t = Lc[k]; Lc[k] = Lo[k]; Lo[k] = (int*)t; }
    }
    hipLaunchKernelGGL(k_nodes, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, V, (const long long*)state, d_node);
    if (A.h_col_pending) {
        // everything above is only enqueued: the host's copy of the pattern (28 MB at 1M vertices, needed by the boundary sets after the
        // rounds) crosses the bus on a stream of its own while the device sorts and partitions
        hipStream_t s2 = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemcpyAsync(A.h_col_pending, A.d_col, sizeof(int32_t) * A.nnz, hipMemcpyDeviceToHost, s2);
        if (e == hipSuccess) e = hipStreamSynchronize(s2);
        if (s2) (void)hipStreamDestroy(s2);
        if (e != hipSuccess) { (void)hipStreamSynchronize(st); return "nd_bisect_device: copy of the pattern failed"; }
    }
    if (hipMemcpyAsync(node, d_node, sizeof(long long) * V, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
        hipGetLastError() != hipSuccess)
        return "nd_bisect_device: device error";
    return "";
}

}  // namespace ls

// the whole analysis with the rounds on the device: what ls_direct_factor runs (csrc/nd_factor.hip), and -- as a plan object -- what
// the GPU tests compare with the host-only ls_nd_plan_create
std::string ls::nd_plan_build_device(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                                     int32_t* h_rowptr, int32_t* h_col, int leaf_size, int arity, int smooth, void* stream, NdPlan& out,
                                     int ordering, bool defer_push_lists) {
    hipStream_t st = (hipStream_t)stream;
    // the host's copy of the pattern: the row pointers now (the analysis looks at them first), the column indices during the device
    // rounds -- unless there are no positions: the graph embedding walks the pattern on the host before anything else
    const bool host_trials = getenv("LS_ND_HOST_TRIALS") != nullptr;      // the trial cuts on host threads, as in round 4 (A/B, tests)
    // (Round 5 measured making the trial cuts the automatic choice between 12k and 300k vertices, where they find 5-10 % thinner separators
    // on rough closed surfaces: cfg3 0.1024 -> 0.0899 ms per solve, cfg2 0.0571 -> 0.0547 -- for +12 ms of constructor at 250k on a mesh in
    // generation order and +23 ms on a mesh fresh from remove_duplicates, whose lexicographic vertex order makes the host's breadth-first
    // sweeps three times slower; the eager optimisation step at these sizes is host-bound and does not get faster at all. Not the
    // default: LS_ND_ORDER=1 asks for it -- a long captured run on one mesh -- and costs 0.026-0.038 s instead of round 4's 0.10.
    // profiles/r05_trial_cuts_on_device.txt)
    const bool host_rounds = ordering == ND_ORDER_MINSEP && host_trials;
    // (the trial cuts need the graph distances, i.e. the pattern on the host, before the rounds start; so does a matrix without positions)
    // (round 6: the graph distances are breadth-first sweeps ON THE DEVICE, nd_embed_device -- the pattern crosses the bus during the rounds
    //  in every case; the rare graphs the device hands back to the host's sweeps fetch it then)
    const bool col_first = host_rounds;
    if (hipMemcpyAsync(h_rowptr, d_rowptr, sizeof(int32_t) * (V + 1), hipMemcpyDeviceToHost, st) != hipSuccess ||
        (col_first && hipMemcpyAsync(h_col, d_col, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess)
        return "the copy of the matrix pattern to the host failed";
    if (h_rowptr[0] != 0 || h_rowptr[V] != nnz) return "rowptr does not match nnz";
    std::vector<float> h_pos;
    auto fetch_positions = [&]() -> bool {
        if (!d_positions) return true;
        h_pos.resize((size_t)V * 3);
        return hipMemcpyAsync(h_pos.data(), d_positions, sizeof(float) * 3 * V, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
    };
    const double t0 = now_s();
    const bool timing = getenv("LS_PLAN_TIMING") != nullptr;
    if (host_rounds) {
        if (!fetch_positions()) return "the copy of the positions to the host failed";
        return nd_plan_build(V, h_rowptr, h_col, d_positions ? h_pos.data() : nullptr, leaf_size, arity, smooth, out, nullptr, nullptr, ND_ORDER_MINSEP, defer_push_lists);
    }
    NdBisectDevice ctx{d_rowptr, d_col, d_positions, nnz, st, col_first ? nullptr : h_col};
    const float given = 0.0f;                  // "positions were given": nd_plan_build only tests the pointer, the values are read on the device
    std::string err = nd_plan_build(V, h_rowptr, h_col, d_positions ? &given : nullptr, leaf_size, arity, smooth, out, nd_bisect_device, &ctx,
                                    ordering == ND_ORDER_MINSEP ? ND_ORDER_MINSEP : ND_ORDER_LONGEST, defer_push_lists, nd_embed_device);
    if (timing) fprintf(stderr, "[nd_plan] returned (pool joined, temporaries released) %.3f s after its start; %.1f factor numbers per vertex, spread %.2f\n",
                        now_s() - t0, out.words_per_vertex, out.spread);
    if (!err.empty() || ordering != ND_ORDER_AUTO || out.spread <= nd_plan_suspect()) return err;
    // Separators thicker than a surface's should be: the cutting planes cross several layers of a surface that is folded or rolled up
    // in space (or several components that lie inside each other). The rounds are run again with graph distances among the candidate
    // directions (on the device; the breadth-first sweeps that give the distances run on the host); the cheaper plan is kept.
    NdPlan B;
    if (host_trials) {
        if (!fetch_positions()) return "";
        err = nd_plan_build(V, h_rowptr, h_col, d_positions ? h_pos.data() : nullptr, leaf_size, arity, smooth, B, nullptr, nullptr, ND_ORDER_MINSEP, defer_push_lists);
    } else {
        NdBisectDevice ctx2{d_rowptr, d_col, d_positions, nnz, st, nullptr};         // (h_col is complete: the first set of rounds has returned)
        err = nd_plan_build(V, h_rowptr, h_col, d_positions ? &given : nullptr, leaf_size, arity, smooth, B, nd_bisect_device, &ctx2, ND_ORDER_MINSEP, defer_push_lists, nd_embed_device);
    }
    if (timing) fprintf(stderr, "[nd_plan] suspect dissection: tried graph distances as well: %.1f factor numbers per vertex, spread %.2f (%s), %.3f s after the start\n",
                        B.words_per_vertex, B.spread, err.empty() ? (B.words_per_vertex < out.words_per_vertex ? "taken" : "not taken") : err.c_str(), now_s() - t0);
    if (!err.empty()) return "";
    const double seconds = out.seconds + B.seconds;
    if (B.words_per_vertex < out.words_per_vertex) { B.words_other = out.words_per_vertex; out = std::move(B); }
    else out.words_other = B.words_per_vertex;
    out.seconds = seconds;
    return "";
}

extern "C" int ls_nd_plan_create_device(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                                        int leaf_size, int arity, int smooth, int device, void* stream, ls_nd_plan** out) {
    using namespace ls;
    LS_REQUIRE(out && d_rowptr && d_col && V > 0 && nnz > 0 && nnz < INT32_MAX, LS_E_INVALID, "ls_nd_plan_create_device: bad argument");
    *out = nullptr;
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    uvec<int32_t> rowptr((size_t)V + 1), col((size_t)nnz);
    ls_nd_plan* h = new ls_nd_plan();
    const std::string err = nd_plan_build_device(d_rowptr, d_col, d_positions, V, nnz, rowptr.data(), col.data(), leaf_size, arity, smooth, st, h->p);
    if (!err.empty()) { delete h; set_error("%s", err.c_str()); return LS_E_INVALID; }
    *out = h;
    return LS_OK;
}

extern "C" int ls_nd_plan_create_device_ordered(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                                                int leaf_size, int arity, int smooth, int ordering, int device, void* stream, ls_nd_plan** out) {
    using namespace ls;
    LS_REQUIRE(out && d_rowptr && d_col && V > 0 && nnz > 0 && nnz < INT32_MAX, LS_E_INVALID, "ls_nd_plan_create_device_ordered: bad argument");
    LS_REQUIRE(ordering >= ND_ORDER_AUTO && ordering <= ND_ORDER_MINSEP, LS_E_INVALID, "ls_nd_plan_create_device_ordered: ordering must be -1, 0 or 1");
    *out = nullptr;
    DeviceGuard g(device);
    LS_HIP(g.err);
    uvec<int32_t> rowptr((size_t)V + 1), col((size_t)nnz);
    ls_nd_plan* h = new ls_nd_plan();
    const std::string err = nd_plan_build_device(d_rowptr, d_col, d_positions, V, nnz, rowptr.data(), col.data(), leaf_size, arity, smooth, (hipStream_t)stream, h->p, ordering);
    if (!err.empty()) { delete h; set_error("%s", err.c_str()); return LS_E_INVALID; }
    *out = h;
    return LS_OK;
}
