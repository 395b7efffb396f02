// nd_span.h -- the levels ABOVE the tier of the nested-dissection re-solve as ONE persistent launch (both sweeps).
//
// Replaces (with csrc/nd_tier.h) the two sparse triangular solves of the reference's default method
// (largesteps/solvers.py:36-39, cholespy `solver.solve(b, x)`).
//
// One launch per tree level costs T_stream + ~4.5 us (kernel boundary, first-byte latency under load, reduction tail):
// with one round of workgroups nothing of a level overlaps with its neighbours, and the nine upper-level launches of a
// 1M-vertex solve spent 120 us on 310 MB (2.6 TB/s). Here the upper levels are phases of ONE launch:
//     up(T-1) ... up(1), root, down(1) ... down(T-1)            (T = first tier level)
//   * one 512-thread workgroup per CU, every workgroup works in every phase;
//   * the factor rows a workgroup needs in phase p+1 are requested into REGISTERS (2 x 16 x 16 bytes per lane: the whole
//     next level of a 1M-vertex mesh fits the chip's register files) before it waits for phase p to finish elsewhere,
//     so the memory system streams level p+1 while level p's results cross the chip;
//   * dependencies follow the TREE, not the grid: the elimination tree's nodes own nested ranges of workgroups (a
//     node's range is the union of its children's), so "the children of node n are done" is a barrier among the
//     workgroups of n's range only -- 4, 16, 64 workgroups for the lower phases, all of them only around the root.
//     Barrier = arrival counter (per 32 workgroups, then per node) whose last arriver publishes release flags; waiters
//     poll one flag with relaxed device-scope loads (MI355X_MICROARCH.md "barrier-xcd");
//   * everything that crosses a barrier (child updates -> pslots, b' -> bp4, x -> xt4; 16-byte entries) is stored
//     write-through (sc1) and read with sc1 loads: no release / acquire cache maintenance (cdna_hip_programming.md
//     guideline 16, form "sc1 stores and sc1 loads on both sides"); the storing waves drain (s_waitcnt vmcnt(0)) before
//     the workgroup arrives. Static data (factor, index lists) and data from the previous launch use plain loads.
// Results are bitwise reproducible (fixed reduction order, no atomics on data). A wait that exceeds ~0.5 s (a workgroup
// that never became resident: another process holds CUs) sets a host-visible flag and gives up; the host side then
// reports it and falls back to one launch per level (csrc/direct.hip).
//
// Matrix layouts of this kernel (fp32, rows 16-byte aligned, zero padded; written by csrc/nd_factor.hip k_convert):
//     pu: up sweep, node rows i < b:  W[i][0 .. s4)                                       upd_i = sum_j W[i][j] b'_j
//     pd: down sweep, node rows j < s: [Finv[j][0 .. s4) | W[0 .. b4)[j]]                 x_j = sum_t Finv[j][t] b'_t - sum_i W[i][j] x_bnd_i
// Lanes run ALONG the reduction (16 bytes = 4 entries per lane and request, 1 KB contiguous per wave); rows of at most
// 128 / 64 / 32 entries are packed 2 / 4 / 8 to a request. A wave reduces with DPP row operations.
#pragma once

namespace ls {

constexpr int SPAN_WAVES = 8, SPAN_THREADS = 64 * SPAN_WAVES;
constexpr int SPAN_NB = 16;        // 16-byte requests per lane and half buffer (two halves form a ring)
constexpr int SPAN_VR = 4;         // vector positions per thread handled as one batch of independent loads
constexpr int SPAN_DOMAIN = 32;    // workgroups per first-level arrival counter

enum : int { SPAN_UP = 0, SPAN_ROOT = 1, SPAN_DOWN = 2 };
enum : int { SPAN_F_LEAF = 1,      // the node has no children: b' = b
             SPAN_F_CHTIER = 2,    // its children are tier nodes: their updates are in the tier's slot array (previous launch)
             SPAN_F_LAST = 4 };    // deepest level above the tier: x is pushed into the tier's boundary vectors

struct alignas(128) SpanJob {      // one workgroup's share of one node in one phase (read as 32 dwords)
    int kind, flags, s, b;
    int own_start, bnd_off, front_off, pfront_off;
    int cix, row0, nrows, len;     // rows [row0, row0 + nrows) of the phase's row space (up: boundary rows, else own rows); len = padded reduction
    int v0, v1, s4, pad0;          // vector positions this workgroup keeps (up: b' -> bp4) / forwards (last down level: boundary rows)
    long long mat_off;             // first float of the node's matrix in pu (up) / pd (root, down)
    int pad1[14];
};
static_assert(sizeof(SpanJob) == 128, "SpanJob layout");

struct alignas(64) SpanSync {      // per (phase, workgroup), read as 16 dwords
    int job0, job1;                // this workgroup's jobs of the phase
    int wait_flag;                 // word to poll before the phase's first vector is assembled (-1: nothing to wait for)
    int arr_ctr, arr_size;         // counter to arrive on after the phase (-1: none)
    int top_ctr, top_size;         // second-level counter the last first-level arriver arrives on (-1: none)
    int rel_flag0, rel_n;          // flags the last arriver publishes
    int njob0, njob1;              // the NEXT phase's jobs (its first record is requested a phase ahead)
    int pad[5];
};
static_assert(sizeof(SpanSync) == 64, "SpanSync layout");

struct SpanArgs {
    const SpanJob* jobs;
    const SpanSync* sync;
    unsigned* words;               // counters and flags, 16 dwords apart, zero at launch
    unsigned* fail;                // host-visible: set when a wait timed out
    int phases, grid, lcap, arity, upper_lo;
    const float* pu;
    const float* pd;
    const float* braw;             // b in the tree's numbering (rows >= upper_lo), stride K, from the previous launch
    const float* tslots;           // the tier's slot array (stride K): updates of the tier's root nodes
    float* txb;                    // the tier's boundary vectors (stride K)
    const unsigned char* mask;
    const int* ppos;
    const int* perm;
    const int* bnd;                // tree-numbering vertex id of every boundary entry
    const int* push_ptr;
    const int* push_tgt;
    float* pslots;                 // child updates per front position of the upper levels: [(f * arity + c) * 4]
    float* bp4;                    // b' of the upper rows: [(row - upper_lo) * 4]
    float* xt4;                    // x of the upper rows in the tree's numbering
    long long* dbg;                // experiments build: shader-clock stamps, 8 per (workgroup, phase)
};

// experiments builds (-DLS_ND_EXPERIMENTS), profile = 2: shader-clock stamps, 8 per (workgroup, phase):
//   0 phase entered (before the wait)  1 wait over  2 vector in LDS  3 products done  4 jobs done  5 stores drained  6 arrived  7 next phase requested
#ifdef LS_ND_EXPERIMENTS
__device__ __forceinline__ void span_stamp(const SpanArgs& a, int slot_base, int k) {
    if (a.dbg && slot_base >= 0 && threadIdx.x == 0) a.dbg[(size_t)slot_base * 8 + k] = (long long)__builtin_amdgcn_s_memtime();
}
#else
__device__ __forceinline__ void span_stamp(const SpanArgs&, int, int) {}
#endif

typedef int span_i4 __attribute__((ext_vector_type(4)));
typedef float span_f4 __attribute__((ext_vector_type(4)));

// 16-byte device-coherent accesses: write-through stores / L1-bypassing loads (sc1), volatile for the compiler
constexpr int SPAN_SC1 = 16 | (int)0x80000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t span_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ span_f4 ld_sc1(__amdgpu_buffer_rsrc_t r, int entry) {
    const span_i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, entry * 16, 0, SPAN_SC1);
    span_f4 f;
    f[0] = __int_as_float(v[0]); f[1] = __int_as_float(v[1]); f[2] = __int_as_float(v[2]); f[3] = __int_as_float(v[3]);
    return f;
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int entry, span_f4 f) {
    span_i4 v;
    v[0] = __float_as_int(f[0]); v[1] = __float_as_int(f[1]); v[2] = __float_as_int(f[2]); v[3] = __float_as_int(f[3]);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, entry * 16, 0, SPAN_SC1);
}

// a job record is fetched with ONE vector load (lane i takes dword i) and unpacked with v_readlane (see load_tile in direct.hip)
__device__ __forceinline__ int span_job_rec(const SpanJob* jobs, int idx) {
    const int lane = threadIdx.x & 63;
    return lane < 18 ? reinterpret_cast<const int*>(jobs)[(size_t)idx * 32 + lane] : 0;
}
__device__ __forceinline__ SpanJob span_job_unpack(int w) {
    SpanJob t;
    t.kind = __builtin_amdgcn_readlane(w, 0); t.flags = __builtin_amdgcn_readlane(w, 1); t.s = __builtin_amdgcn_readlane(w, 2);
    t.b = __builtin_amdgcn_readlane(w, 3); t.own_start = __builtin_amdgcn_readlane(w, 4); t.bnd_off = __builtin_amdgcn_readlane(w, 5);
    t.front_off = __builtin_amdgcn_readlane(w, 6); t.pfront_off = __builtin_amdgcn_readlane(w, 7); t.cix = __builtin_amdgcn_readlane(w, 8);
    t.row0 = __builtin_amdgcn_readlane(w, 9); t.nrows = __builtin_amdgcn_readlane(w, 10); t.len = __builtin_amdgcn_readlane(w, 11);
    t.v0 = __builtin_amdgcn_readlane(w, 12); t.v1 = __builtin_amdgcn_readlane(w, 13); t.s4 = __builtin_amdgcn_readlane(w, 14); t.pad0 = 0;
    t.mat_off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w, 17) << 32) | (unsigned)__builtin_amdgcn_readlane(w, 16));
    return t;
}
__device__ __forceinline__ int span_load_sync(const SpanSync* sync, size_t idx) {
    const int lane = threadIdx.x & 63;
    return lane < 11 ? reinterpret_cast<const int*>(sync)[idx * 16 + lane] : 0;
}

// how a wave walks its rows of a job: rows [r_lo, r_lo + n_my), `pack` rows per 1 KB request, lpr requests per row
struct SpanRows { int r_lo, n_my, packsh, lpr; };
__device__ __forceinline__ SpanRows span_rows(const SpanJob& J, int wave) {
    SpanRows R;
    const int rpw = (J.nrows + SPAN_WAVES - 1) / SPAN_WAVES;
    R.r_lo = J.row0 + wave * rpw;
    R.n_my = max(0, min(rpw, J.row0 + J.nrows - R.r_lo));
    R.packsh = J.len <= 32 ? 3 : J.len <= 64 ? 2 : J.len <= 128 ? 1 : 0;
    R.lpr = R.packsh ? 1 : (J.len + 255) >> 8;
    return R;
}

// requests [g0, g0 + SPAN_NB) of the row chunk that starts at row c0 (n_c rows, request g = (row group g / lpr, piece g % lpr)).
// The node's matrix is addressed through a buffer descriptor (wave-uniform base, one 32-bit offset register per request).
__device__ __forceinline__ void span_load_half(span_f4 (&buf)[SPAN_NB], __amdgpu_buffer_rsrc_t mat, const SpanJob& J, const SpanRows& R,
                                               int c0, int n_c, int nreq, int g0) {
    const int lane = threadIdx.x & 63;
    const int segsh = 6 - R.packsh, sub = lane >> segsh, col = lane & ((1 << segsh) - 1);
    int grp = g0 / R.lpr, e = g0 - grp * R.lpr;
#pragma unroll
    for (int i = 0; i < SPAN_NB; ++i) {
        const int row = (grp << R.packsh) + sub;
        const int t = R.packsh ? col * 4 : (e * 64 + lane) * 4;
        const bool ok = g0 + i < nreq && row < n_c && t < J.len;
        const span_f4 z = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const span_i4 v = __builtin_amdgcn_raw_buffer_load_b128(mat, ((c0 + row) * J.len + t) * 4, 0, 0);
            buf[i][0] = __int_as_float(v[0]); buf[i][1] = __int_as_float(v[1]); buf[i][2] = __int_as_float(v[2]); buf[i][3] = __int_as_float(v[3]);
        } else buf[i] = z;
        if (++e == R.lpr) { e = 0; ++grp; }
    }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float span_dpp(float v) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(m);
}

// multiply the requests [g0, g0 + SPAN_NB) with the vector in LDS (K arrays of lcap floats); a finished row group is reduced over
// its lanes and the row sums go to `outw` ([row of the chunk][4]) of this wave
template <int K>
__device__ __forceinline__ void span_fma_half(const span_f4 (&buf)[SPAN_NB], const float* __restrict__ vec, int lcap, const SpanJob& J,
                                              const SpanRows& R, int n_c, int nreq, int g0, float (&acc)[K], float* __restrict__ outw) {
    const int lane = threadIdx.x & 63;
    const int segsh = 6 - R.packsh, sub = lane >> segsh, col = lane & ((1 << segsh) - 1);
    int grp = g0 / R.lpr, e = g0 - grp * R.lpr;
#pragma unroll
    for (int i = 0; i < SPAN_NB; ++i) {
        if (g0 + i < nreq) {
            const int t = R.packsh ? col * 4 : (e * 64 + lane) * 4;
            if (t < J.len) {
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    const span_f4 v = *reinterpret_cast<const span_f4*>(vec + (size_t)q * lcap + t);
                    acc[q] = fmaf(buf[i][0], v[0], acc[q]); acc[q] = fmaf(buf[i][1], v[1], acc[q]);
                    acc[q] = fmaf(buf[i][2], v[2], acc[q]); acc[q] = fmaf(buf[i][3], v[3], acc[q]);
                }
            }
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // bounds the LDS reads in flight (registers)
            if (e + 1 == R.lpr) {                  // the row group is complete: sum over the lanes of every row (fixed order)
                const int row = (grp << R.packsh) + sub;
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    float v = acc[q];
                    v = span_dpp<0xB1, 0xf>(v); v = span_dpp<0x4E, 0xf>(v); v = span_dpp<0x141, 0xf>(v);      // 8 lanes
                    if (R.packsh <= 2) v = span_dpp<0x140, 0xf>(v);                                           // 16
                    if (R.packsh <= 1) v = span_dpp<0x142, 0xa>(v);                                           // 32 (in rows 1, 3)
                    if (R.packsh == 0) v = span_dpp<0x143, 0xc>(v);                                           // 64 (in lane 63)
                    if (col == (1 << segsh) - 1 && row < n_c) outw[row * 4 + q] = v;
                    acc[q] = 0.0f;
                }
            }
        }
        if (++e == R.lpr) { e = 0; ++grp; }
    }
}

// what is requested for a job BEFORE the workgroup waits for the previous phase (nothing here depends on that phase)
struct SpanPre {
    span_f4 A[SPAN_NB], B[SPAN_NB];
    int eidx, p0, p1;              // epilogue of this lane's row of chunk 0: up: parent position; else: caller's row id, push list
};

template <int K>
__device__ __forceinline__ void span_pre(const SpanArgs& a, const SpanJob& J, int wave, SpanPre& P) {
    const int lane = threadIdx.x & 63;
    const SpanRows R = span_rows(J, wave);
    const int n_c = min(64, R.n_my), nreq = ((n_c + (1 << R.packsh) - 1) >> R.packsh) * R.lpr;
    const __amdgpu_buffer_rsrc_t mat = span_rsrc((J.kind == SPAN_UP ? a.pu : a.pd) + J.mat_off);
    span_load_half(P.A, mat, J, R, R.r_lo, n_c, nreq, 0);
    span_load_half(P.B, mat, J, R, R.r_lo, n_c, nreq, SPAN_NB);
    const int r = R.r_lo + lane;
    const bool mine = lane < n_c;
    P.p0 = P.p1 = 0;
    if (J.kind == SPAN_UP) P.eidx = mine ? a.ppos[J.bnd_off + r] : 0;
    else {
        P.eidx = mine ? a.perm[J.own_start + r] : 0;
        if ((J.flags & SPAN_F_LAST) && mine) { P.p0 = a.push_ptr[J.front_off + r]; P.p1 = a.push_ptr[J.front_off + r + 1]; }
    }
}

// The children's updates at front position f (mask: which children contribute). Two steps so that the loads of SEVERAL positions
// are in flight together: span_slots_load requests (A slots of 16 bytes, or A x K floats of the tier's array), span_slots_sum adds.
template <int K, int A>
struct SpanSlots { span_f4 r[A]; unsigned m; };
template <int K, int A>
__device__ __forceinline__ void span_slots_load(const SpanArgs& a, const SpanJob& J, __amdgpu_buffer_rsrc_t rs, int f, bool on, SpanSlots<K, A>& S) {
    const span_f4 z = {0.f, 0.f, 0.f, 0.f};
    S.m = on ? a.mask[f] : 0u;
    if (J.flags & SPAN_F_CHTIER) {
#pragma unroll
        for (int c = 0; c < A; ++c) {
            S.r[c] = z;
#pragma unroll
            for (int q = 0; q < K; ++q) S.r[c][q] = on ? a.tslots[((size_t)f * A + c) * K + q] : 0.0f;
        }
    } else {
#pragma unroll
        for (int c = 0; c < A; ++c) S.r[c] = on ? ld_sc1(rs, f * A + c) : z;
    }
}
template <int K, int A>
__device__ __forceinline__ void span_slots_sum(const SpanSlots<K, A>& S, float (&u)[K]) {
#pragma unroll
    for (int q = 0; q < K; ++q) u[q] = 0.0f;
#pragma unroll
    for (int c = 0; c < A; ++c) {
#pragma unroll
        for (int q = 0; q < K; ++q) u[q] += ((S.m >> c) & 1u) ? S.r[c][q] : 0.0f;
    }
}

__device__ __forceinline__ void span_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The reduction vector of a job, K arrays of lcap floats in LDS:
//   up / root   b'_t = b_t - (children's updates at own position t)          [a slice of it is kept for the down sweep]
//   down        [b'_t | -x of the boundary vertices]                          [last level: the boundary rows' x goes on to the tier]
template <int K, int A>
__device__ __forceinline__ void span_vector(const SpanArgs& a, const SpanJob& J, float* __restrict__ vec) {
    const __amdgpu_buffer_rsrc_t r_slots = span_rsrc(a.pslots), r_bp = span_rsrc(a.bp4), r_xt = span_rsrc(a.xt4);
    const int lcap = a.lcap;
    if (J.kind != SPAN_DOWN) {
        const bool leaf = J.flags & SPAN_F_LEAF;
        for (int t0 = threadIdx.x; t0 < J.s4; t0 += SPAN_THREADS * SPAN_VR) {
            float v[SPAN_VR][K];
            SpanSlots<K, A> S[SPAN_VR];
#pragma unroll
            for (int r = 0; r < SPAN_VR; ++r) {
                const int t = t0 + r * SPAN_THREADS;
#pragma unroll
                for (int q = 0; q < K; ++q) v[r][q] = t < J.s ? a.braw[(size_t)(J.own_start + t) * K + q] : 0.0f;
                span_slots_load<K, A>(a, J, r_slots, J.front_off + t, t < J.s && !leaf, S[r]);
            }
#pragma unroll
            for (int r = 0; r < SPAN_VR; ++r) {
                const int t = t0 + r * SPAN_THREADS;
                if (t < J.s4) {
                    float u[K];
                    span_slots_sum<K, A>(S[r], u);
#pragma unroll
                    for (int q = 0; q < K; ++q) { v[r][q] -= u[q]; vec[(size_t)q * lcap + t] = v[r][q]; }
                    if (J.kind == SPAN_UP && t >= J.v0 && t < J.v1) {
                        span_f4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < K; ++q) w[q] = v[r][q];
                        st_sc1(r_bp, J.own_start + t - a.upper_lo, w);
                    }
                }
            }
        }
        return;
    }
    for (int t0 = threadIdx.x; t0 < J.s4; t0 += SPAN_THREADS * SPAN_VR) {
        span_f4 w[SPAN_VR];
#pragma unroll
        for (int r = 0; r < SPAN_VR; ++r) {
            const int t = t0 + r * SPAN_THREADS;
            const span_f4 z = {0.f, 0.f, 0.f, 0.f};
            w[r] = t < J.s ? ld_sc1(r_bp, J.own_start + t - a.upper_lo) : z;
        }
#pragma unroll
        for (int r = 0; r < SPAN_VR; ++r) {
            const int t = t0 + r * SPAN_THREADS;
            if (t < J.s4) {
#pragma unroll
                for (int q = 0; q < K; ++q) vec[(size_t)q * lcap + t] = w[r][q];
            }
        }
    }
    const int b4 = J.len - J.s4;
    for (int i0 = threadIdx.x; i0 < b4; i0 += SPAN_THREADS * SPAN_VR) {
        int idx[SPAN_VR];
#pragma unroll
        for (int r = 0; r < SPAN_VR; ++r) { const int i = i0 + r * SPAN_THREADS; idx[r] = i < J.b ? a.bnd[J.bnd_off + i] : -1; }
        span_f4 w[SPAN_VR];
#pragma unroll
        for (int r = 0; r < SPAN_VR; ++r) {
            const span_f4 z = {0.f, 0.f, 0.f, 0.f};
            w[r] = idx[r] >= 0 ? ld_sc1(r_xt, idx[r] - a.upper_lo) : z;
        }
#pragma unroll
        for (int r = 0; r < SPAN_VR; ++r) {
            const int i = i0 + r * SPAN_THREADS;
            if (i < b4) {
#pragma unroll
                for (int q = 0; q < K; ++q) vec[(size_t)q * lcap + J.s4 + i] = -w[r][q];
                if ((J.flags & SPAN_F_LAST) && i >= J.v0 && i < J.v1) {          // hand the boundary rows' x on to the tier's root nodes
                    const int f = J.front_off + J.s + i;
                    for (int p = a.push_ptr[f]; p < a.push_ptr[f + 1]; ++p) {
                        const size_t tgt = (size_t)a.push_tgt[p];
#pragma unroll
                        for (int q = 0; q < K; ++q) a.txb[tgt * K + q] = w[r][q];
                    }
                }
            }
        }
    }
}

// what happens to the sum of row `r` (lane's row of the current chunk)
template <int K>
__device__ __forceinline__ void span_epilogue(const SpanArgs& a, const SpanJob& J, int r, int eidx, int p0, int p1, const float (&pass)[K],
                                              const float* __restrict__ outw, int lane, float* __restrict__ x_out) {
    float v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = outw[lane * 4 + q];
    span_f4 w = {0.f, 0.f, 0.f, 0.f};
    if (J.kind == SPAN_UP) {
#pragma unroll
        for (int q = 0; q < K; ++q) w[q] = v[q] + pass[q];
        st_sc1(span_rsrc(a.pslots), (J.pfront_off + eidx) * a.arity + J.cix, w);
        return;
    }
#pragma unroll
    for (int q = 0; q < K; ++q) { w[q] = v[q]; x_out[(size_t)eidx * K + q] = v[q]; }
    if (!(J.flags & SPAN_F_LAST)) { st_sc1(span_rsrc(a.xt4), J.own_start + r - a.upper_lo, w); return; }
    for (int p = p0; p < p1; ++p) {
        const size_t tgt = (size_t)a.push_tgt[p];
#pragma unroll
        for (int q = 0; q < K; ++q) a.txb[tgt * K + q] = v[q];
    }
}

template <int K, int A>
__device__ __forceinline__ void span_job(const SpanArgs& a, const SpanJob& J, SpanPre& P, bool pre_loaded, int wave, float* __restrict__ vec,
                                         float* __restrict__ outw, float* __restrict__ x_out, int stamp_wg) {
    const int lane = threadIdx.x & 63;
    const SpanRows R = span_rows(J, wave);
    const __amdgpu_buffer_rsrc_t mat = span_rsrc((J.kind == SPAN_UP ? a.pu : a.pd) + J.mat_off);
    for (int c = 0; c == 0 || c < R.n_my; c += 64) {                // row chunks of the wave (one for all but very large shares)
        const int c0 = R.r_lo + c, n_c = max(0, min(64, R.n_my - c));
        const int nreq = ((n_c + (1 << R.packsh) - 1) >> R.packsh) * R.lpr;
        const bool have = c == 0 && pre_loaded;                      // the first two half buffers and the epilogue indices are on their way
        int eidx = P.eidx, p0 = P.p0, p1 = P.p1;
        if (!have) {
            const bool mine = lane < n_c;
            p0 = p1 = 0;
            if (J.kind == SPAN_UP) eidx = mine ? a.ppos[J.bnd_off + c0 + lane] : 0;
            else {
                eidx = mine ? a.perm[J.own_start + c0 + lane] : 0;
                if ((J.flags & SPAN_F_LAST) && mine) { p0 = a.push_ptr[J.front_off + c0 + lane]; p1 = a.push_ptr[J.front_off + c0 + lane + 1]; }
            }
        }
        // up: what the children hand to this lane's boundary row travels on with the row's own sum
        SpanSlots<K, A> PS;
        span_slots_load<K, A>(a, J, span_rsrc(a.pslots), J.front_off + J.s + c0 + lane, J.kind == SPAN_UP && lane < n_c && !(J.flags & SPAN_F_LEAF), PS);
        if (c == 0) { span_vector<K, A>(a, J, vec); __syncthreads(); span_stamp(a, stamp_wg, 2); }
        float pass[K];
        span_slots_sum<K, A>(PS, pass);
        float acc[K];
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0f;
        // ring of two half buffers; a chunk that was not requested ahead starts one (virtual) round early, with the requests only
        for (int g = have ? 0 : -2 * SPAN_NB; g < nreq; g += 2 * SPAN_NB) {
            if (g >= 0) span_fma_half<K>(P.A, vec, a.lcap, J, R, n_c, nreq, g, acc, outw);
            if (g + 2 * SPAN_NB < nreq) span_load_half(P.A, mat, J, R, c0, n_c, nreq, g + 2 * SPAN_NB);
            if (g >= 0 && g + SPAN_NB < nreq) span_fma_half<K>(P.B, vec, a.lcap, J, R, n_c, nreq, g + SPAN_NB, acc, outw);
            if (g + 3 * SPAN_NB < nreq) span_load_half(P.B, mat, J, R, c0, n_c, nreq, g + 3 * SPAN_NB);
        }
        span_wave_lds_sync();
        if (c == 0) span_stamp(a, stamp_wg, 3);
        if (lane < n_c) span_epilogue<K>(a, J, c0 + lane, eidx, p0, p1, pass, outw, lane, x_out);
        span_wave_lds_sync();
    }
}

template <int K, int A>
__global__ __launch_bounds__(SPAN_THREADS) void k_nd_span(SpanArgs a, float* __restrict__ x_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* vec = sm;                                                   // K x lcap
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* outw = sm + (size_t)K * a.lcap + (size_t)wave * 256;        // this wave's row sums: 64 x 4
    const int G = a.grid;
    // consecutive logical ids share an XCD (workgroup b runs on XCD b % 8 -- observed placement, used for speed only)
    const int w = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int rec = span_load_sync(a.sync, (size_t)w);
    SpanPre P;
    bool pre_loaded = false;
    SpanJob Jpre = span_job_unpack(0);
    int jrec_next = 0, nj0 = 0, nj1 = 0;
    int job0 = 0, job1 = 0, wait_flag = -1, arr_ctr = -1, arr_size = 0, top_ctr = -1, top_size = 0, rel_flag0 = 0, rel_n = 0;
    for (int ph = -1; ph < a.phases; ++ph) {          // ph = -1: only the requests for phase 0
        const int sb = ph >= 0 ? w * a.phases + ph : -1;
        if (ph >= 0) {
            span_stamp(a, sb, 0);
            if (wait_flag >= 0) {
                if (threadIdx.x == 0) {
                    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
                    while (__hip_atomic_load(a.words + (size_t)wait_flag * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(4);
                        if ((long long)__builtin_amdgcn_s_memtime() - t0 > 1000000000ll) {      // ~0.5 s of shader clocks: a workgroup is not resident
                            __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
                __syncthreads();
            }
            span_stamp(a, sb, 1);
            for (int j = job0; j < job1; ++j) {
                const bool first = j == job0 && pre_loaded;
                const SpanJob J = first ? Jpre : span_job_unpack(span_job_rec(a.jobs, j));
                span_job<K, A>(a, J, P, first, wave, vec, outw, x_out, j == job0 ? sb : -1);
                __syncthreads();                                        // the vector in LDS is rewritten by the next job
            }
            pre_loaded = false;
            span_stamp(a, sb, 4);
            if (ph + 1 == a.phases) break;
            // every store of this phase has left the CU before the workgroup counts as arrived
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            span_stamp(a, sb, 5);
            if (arr_ctr >= 0 && threadIdx.x == 0) {
                bool last = __hip_atomic_fetch_add(a.words + (size_t)arr_ctr * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(arr_size - 1);
                if (last && top_ctr >= 0)
                    last = __hip_atomic_fetch_add(a.words + (size_t)top_ctr * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(top_size - 1);
                if (last)
                    for (int f = 0; f < rel_n; ++f) __hip_atomic_store(a.words + (size_t)(rel_flag0 + f) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        span_stamp(a, sb, 6);
        // the next phase: its first job's matrix rows stream while the rest of the chip finishes this one
        if (ph >= 0) { Jpre = span_job_unpack(jrec_next); }
        job0 = __builtin_amdgcn_readlane(rec, 0); job1 = __builtin_amdgcn_readlane(rec, 1); wait_flag = __builtin_amdgcn_readlane(rec, 2);
        arr_ctr = __builtin_amdgcn_readlane(rec, 3); arr_size = __builtin_amdgcn_readlane(rec, 4); top_ctr = __builtin_amdgcn_readlane(rec, 5);
        top_size = __builtin_amdgcn_readlane(rec, 6); rel_flag0 = __builtin_amdgcn_readlane(rec, 7); rel_n = __builtin_amdgcn_readlane(rec, 8);
        nj0 = __builtin_amdgcn_readlane(rec, 9); nj1 = __builtin_amdgcn_readlane(rec, 10);
        if (ph < 0 && job0 < job1) Jpre = span_job_unpack(span_job_rec(a.jobs, job0));
        if (job0 < job1) { span_pre<K>(a, Jpre, wave, P); pre_loaded = true; }
        if (ph + 2 < a.phases) rec = span_load_sync(a.sync, (size_t)(ph + 2) * G + w);
        jrec_next = nj0 < nj1 ? span_job_rec(a.jobs, nj0) : 0;          // the record of the phase after: arrives while the next one runs
        span_stamp(a, sb, 7);
    }
}

}  // namespace ls
