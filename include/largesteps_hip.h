/*
 * largesteps_hip.h -- C ABI of liblargesteps_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the `largesteps` parameterization hot path. Every entry point takes plain
 * device pointers, sizes, a HIP device ordinal and a hipStream_t passed as void*; no torch types, no
 * C++ types, nothing is thrown across the boundary. Every function returns an int status:
 *       0            success
 *      <0            LS_E_* (invalid argument / state, see below)
 *      >0            a hipError_t raised by the runtime
 * ls_last_error() returns a thread-local human readable message for the last non-zero status.
 *
 * All pointers are DEVICE pointers unless the parameter name starts with `h_` (host).
 * Work is enqueued on `stream`; functions that have to return a size to the host (marked SYNC)
 * synchronise that stream once.
 *
 * Reference interface each entry point replaces (paths relative to rgl-epfl/large-steps-pytorch):
 *   ls_assemble_*        largesteps/geometry.py:65-94  (laplacian_uniform), :3-63 (laplacian_cot),
 *                        :96-133 (compute_matrix: M = a I + b L, coalesced)
 *   ls_csr_from_coo      the implicit COO->CSR conversion inside torch's sparse mm
 *                        (largesteps/parameterize.py:30) for a matrix that was not built by ls_assemble_*
 *   ls_spmv              largesteps/parameterize.py:30  (to_differential: u = M @ v)
 *   ls_solver_*          largesteps/solvers.py:26-39 (CholeskySolver.__init__/solve -> cholespy) and
 *                        :41-126 (ConjugateGradientSolver); one handle == one cached solver object of
 *                        largesteps/parameterize.py:48-59
 *   ls_solver_phase/...  no reference counterpart: the reference is single process / single GPU
 *   ls_direct_*          largesteps/solvers.py:36-39 (CholeskySolver.solve: the two triangular solves of the cholespy /
 *                        CHOLMOD factorisation built at :34) -- re-solve phase of the nested-dissection direct solver
 *   ls_adam_uniform_step largesteps/optimize.py:18-41
 *   ls_face_normals* / ls_vertex_normals*   scripts/geometry.py:91-110 (compute_face_normals), :115-147
 *                        (compute_vertex_normals) and the autograd graph torch records through them -- the consumer
 *                        right after from_differential in every optimisation step (scripts/main.py:178-179)
 */
#ifndef LARGESTEPS_HIP_H
#define LARGESTEPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS_VERSION 110 /* 0.1.1: ls_direct_factor_ex / ls_direct_options (round 6), ls_direct_arrays.tier_waves */

#define LS_OK 0
#define LS_E_INVALID (-1)      /* bad argument (null pointer, negative size, unsupported k, ...) */
#define LS_E_INDEX (-2)        /* a face index is outside [0, V) */
#define LS_E_WORKSPACE (-3)    /* workspace too small */
#define LS_E_OVERFLOW (-4)     /* problem does not fit the int32 index space of the kernels */
#define LS_E_STATE (-5)        /* call sequence violated (e.g. fill before pattern) */
#define LS_E_NOT_CONVERGED (-6) /* solver hit max_iter or a non-finite residual; x holds the last iterate */

#define LS_LAPLACIAN_UNIFORM 0
#define LS_LAPLACIAN_COT 1

int ls_version(void);
const char* ls_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Assembly of M = a*I + b*L as CSR (int32 rowptr/col, fp32 val) + the row-major sorted, duplicate
 * free COO index list torch's coalesce() would produce (geometry.py:133).
 *
 *   1. ls_assemble_workspace_bytes -> size the caller must allocate (device memory)
 *   2. ls_assemble_pattern  (SYNC)  -> fills rowptr[V+1], returns nnz through h_nnz
 *   3. caller allocates col[nnz], val[nnz], coo_idx[2*nnz] (int64: row indices then column indices)
 *   4. ls_assemble_fill             -> writes them (same V, F, workspace); optionally dinv[V] = 1/M_ii
 *
 * faces: (F,3) row-major, index width idx_bytes in {4, 8}. verts: (V,3) fp32 row-major, only read
 * for LS_LAPLACIAN_COT (may be NULL otherwise). `a` and `b` are the already fp32-rounded scalars
 * (a=1,b=lambda  or  a=1-alpha,b=alpha).
 * --------------------------------------------------------------------------------------------- */
int ls_assemble_workspace_bytes(int64_t V, int64_t F, size_t* h_bytes);
int ls_assemble_pattern(const void* faces, int idx_bytes, int64_t F, int64_t V, const float* verts,
                        int kind, float a, float b, void* workspace, size_t workspace_bytes,
                        int32_t* rowptr, int64_t* h_nnz, int device, void* stream);
int ls_assemble_fill(const void* workspace, size_t workspace_bytes, int64_t V, int64_t F, const int32_t* rowptr,
                     int32_t* col, float* val, int64_t* coo_idx, int64_t nnz, float* dinv, int device,
                     void* stream);

/* Foreign matrix: coalesced (row-major sorted, unique) COO int64 -> CSR int32. rowptr[V+1], col[nnz];
 * val is shared with the COO tensor (same order) so only indices are produced. dinv[V] = 1/M_ii optional.
 * scratch: >= 4*(V+1) + 4*((V+1)/2048+4) + 1024 bytes. SYNC (validates sortedness / index range). */
int ls_csr_from_coo(const int64_t* coo_rows, const int64_t* coo_cols, const float* vals, int64_t nnz,
                    int64_t V, int32_t* rowptr, int32_t* col, float* dinv, void* scratch,
                    size_t scratch_bytes, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * y[V,k] = M x[V,k]   (row-major, leading dimension = k, fp32). 1 <= k <= 64.
 * variant: 0 = default (LDS-staged CSR), 1 = thread-per-row direct CSR (kept for A/B measurements).
 * --------------------------------------------------------------------------------------------- */
int ls_spmv(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V, int64_t nnz,
            const float* x, float* y, int k, int variant, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Jacobi-preconditioned conjugate gradient on M (SPD). One handle per matrix; owns a SELL-64 copy of the
 * matrix and its workspace (dinv, r, p, Ap, reduction scratch) sized for `kmax` right-hand-side columns
 * (1..4). The CSR arrays are only read during creation. SYNC (once, to size the SELL copy).
 * --------------------------------------------------------------------------------------------- */
typedef struct ls_solver ls_solver;

typedef struct ls_solve_info {
    int32_t iterations;   /* iterations executed until every column met its threshold (-1: still running, ls_solver_poll) */
    int32_t converged;    /* 1 if every column met its threshold */
    double  rnorm[4];     /* final ||r||_2 per column (recursively updated residual) */
    double  bnorm[4];     /* ||b||_2 per column */
} ls_solve_info;

int ls_solver_create(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V,
                     int64_t nnz, int kmax, int device, void* stream, ls_solver** h_out);
int ls_solver_destroy(ls_solver* s);
/* x0 may be NULL (cold start, x0 = 0). b, x, x0: (V,k) fp32 row-major contiguous; x may alias x0 but not b.
 * A column is converged when ||r||_2 <= max(rtol*||b||_2, atol). SYNC (the host polls convergence).
 * Returns LS_E_NOT_CONVERGED (info still filled) if max_iter is hit. */
int ls_solver_solve(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol,
                    double atol, int max_iter, ls_solve_info* h_info, void* stream);
/* Chebyshev-accelerated Jacobi iteration on the same handle: one kernel per iteration, no dot products. Needs
 * ls_solver_set_spectrum(s, a_min) with a_min <= lambda_min(M) (M = a I + b L with L positive semi-definite: a);
 * the upper end of spec(D^-1 M) is the handle's own Gershgorin bound. The iteration count is fixed a priori:
 * ceil(log(2/t) / log((sqrt(kappa)+1)/(sqrt(kappa)-1))) for the requested residual reduction t (rtol, or
 * max(rtol||b||, atol)/||r0|| after one residual evaluation when x0 or atol is given). h_info->rnorm is the TRUE
 * fp32 residual of the returned x. SYNC once at the end. LS_E_STATE without a spectrum, LS_E_NOT_CONVERGED if the
 * count exceeds max_iter or the final check fails: ||b - M x|| must be <= max(request, 8 eps32 ||M|| ||x||), the
 * backward-stable fp32 level (callers fall back to ls_solver_solve). */
int ls_solver_set_spectrum(ls_solver* s, double a_min);
/* Declare M = a I + b L_uniform (every off-diagonal entry equals -b): the Chebyshev solver then reads only the
 * neighbour ids (4 B per entry instead of 8 B) -- values are implicit. SYNC once (sizes the column-only SELL copy). */
int ls_solver_set_uniform(ls_solver* s, float a, float b, void* stream);
int ls_solver_spectrum(const ls_solver* s, double* h_lmin, double* h_lmax);
/* Patch plan of the LDS-resident s-step Chebyshev kernel (host arrays, built by largesteps/patches.py; needs
 * ls_solver_set_uniform first): the vertices are renumbered patch-major (h_perm[new] = old); patch p owns the new ids
 * [table[20p], table[20p] + table[20p+1]); one workgroup keeps both iterates of the patch and of its ghost layers 1..depth
 * in LDS and advances `depth` Chebyshev steps per launch, so HBM sees vectors and matrix once per `depth` iterations.
 * table: 20 int32 per patch {own_start, n_own, n_rows, n_local, ell_width, off_gid, off_cols, off_diag, lim[0..11]}
 * (depth <= 12) with
 * lim[m] = number of rows in layers <= m (lim[0] = n_own, lim[m >= depth-1] = n_rows): step j of an S-step launch only
 * recomputes rows < lim[S-1-j], the outer layers are stale by then and never reach an own vertex; h_ghost_gid:
 * new global ids of the local vertices >= n_own; h_cols16: per patch (ell_width, n_rows) uint16 local neighbour ids
 * (padding = n_local); h_diag: per patch the n_rows diagonal entries. SYNC (copies the arrays). */
int ls_solver_set_patches(ls_solver* s, const int32_t* h_table, int n_patches, const int32_t* h_ghost_gid, int64_t n_gid,
                          const uint16_t* h_cols16, int64_t n_cols, const float* h_diag, int64_t n_diag,
                          const int32_t* h_perm, int depth, int max_local, int max_rows, void* stream);
/* iterations the Chebyshev solver will run for a residual reduction `reduction` (e.g. rtol from a cold start) */
int ls_solver_chebyshev_iterations(const ls_solver* s, double reduction, int* h_n);
int ls_solver_solve_chebyshev(ls_solver* s, const float* b, const float* x0, float* x, int k, double rtol,
                              double atol, int max_iter, ls_solve_info* h_info, void* stream);
/* knobs for measurements: name in {"check_every", "grid" (workgroups per kernel, 0 = auto), "block" (0 = auto,
 * 256, 512 or 1024 threads per workgroup), "graph" (1 = replay the Chebyshev launches of a solve as one hipGraph,
 * default; 0 = eager launches), "patch" (1 = use the patch plan if one was set, default), "profile"}; unknown name -> LS_E_INVALID */
int ls_solver_set(ls_solver* s, const char* name, int value);
/* With "profile"=1 every solve brackets its three kernels per iteration with HIP events on the solve's
 * stream; this returns the accumulated milliseconds of K1 (SpMV+dot), K2 (update), K3 (direction) over the h_iters
 * iterations that really ran in the last PCG solve; after a Chebyshev solve: [0] = total of the h_iters launches. */
int ls_solver_profile(const ls_solver* s, double* h_ms3, int* h_iters);
/* the handle's SELL-64 copy of the matrix: slice_ptr[V/64+1] (entry offsets), cv[entries] = {col, fp32 bits} */
int ls_solver_sell(ls_solver* s, const int32_t** h_slice_ptr, const void** h_cv, int64_t* h_entries);
/* bytes the handle allocated on the device */
int ls_solver_workspace_bytes(const ls_solver* s, size_t* h_bytes);

/* ------------------------------------------------------------------------------------------------
 * Vertex-block shard of the same solver (one process per GPU, largesteps/distributed.py). The shard owns
 * rows [0, n_rows) of its block; columns are local ids: [0, n_rows) owned, [n_rows, n_cols) halo. The host
 * driver launches one kernel at a time and, between them, exchanges the halo rows of p and sums the
 * reduction partials across ranks (RCCL):
 *   phase 0  r = b, p = D^-1 r, x = 0 ; partials r.z, r.r, b.b      -> all-reduce partials [1..3]
 *   phase 1  thresholds / column mask from the summed partials
 *   phase 2  K1: Ap = M p_ext ; partial p.Ap  (needs the halo of p)   -> all-reduce partials [0]
 *   phase 3  K2: x += a p ; r -= a Ap ; partials r.z, r.r             -> all-reduce partials [1..2]
 *   phase 4  K3: p = D^-1 r + b p ; publishes the stop flag            -> halo exchange of p
 * ls_solver_buffers exposes p ((n_cols,k) fp32; the halo rows start at p + n_rows*k) and the partial array
 * (4 slots x 4 columns x h_part_stride doubles; slot s, column c, workgroup g at ((s*4+c)*stride + g)); every
 * rank must use the same "grid" so that the partial arrays line up.
 * --------------------------------------------------------------------------------------------- */
int ls_solver_create_ext(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                         int64_t n_cols, int64_t nnz, int kmax, int device, void* stream, ls_solver** h_out);
int ls_solver_phase(ls_solver* s, int phase, const float* b, float* x, int k, double rtol, double atol,
                    int it, void* stream);
int ls_solver_buffers(ls_solver* s, float** h_p, double** h_part, int* h_grid, int* h_part_stride);
/* Replace the handle's p ((n_cols*kmax) floats) and partial array (4*4*stride doubles, zero-initialised by the
 * caller) by caller-owned device buffers, so that the host driver can hand them to its communication library. */
int ls_solver_bind(ls_solver* s, float* p_ext, double* part);
/* Sharded Chebyshev (largesteps/distributed.py ShardedChebyshev): `nsteps` iterations it0..it0+nsteps-1 on the first
 * n_rows rows of the shard (owned rows + redundantly computed ghost layers). xa / xb: (n_cols,k) ping-pong buffers,
 * iteration `it` gathers from (it even ? xa : xb) and overwrites the other; h_c1/h_c2: HOST arrays of the nsteps
 * Chebyshev coefficients (it = 0: x1 = x0 + c2 D^-1 (b - M x0)). No collective is needed between these launches. */
int ls_shard_cheb_steps(ls_solver* s, const float* b, float* xa, float* xb, int k, int it0, int nsteps,
                        const float* h_c1, const float* h_c2, int64_t n_rows, void* stream);
/* partials of ||b - M x||^2 (slots 1 and 2) and ||b||^2 (slot 3) over the first n_rows rows -> all-reduce ->
 * ls_solver_phase(1) -> ls_solver_poll gives the global norms */
int ls_shard_resnorm(ls_solver* s, const float* b, const float* x, int k, int64_t n_rows, void* stream);
/* SYNC: copies the device scalars; h_info->iterations = -1 while no stop was published. */
int ls_solver_poll(ls_solver* s, int k, int n_enqueued, ls_solve_info* h_info, void* stream);
/* dst[t,:] = src[idx[t],:] for t < n (halo send buffer packing), k in 1..4 */
int ls_gather_rows(const float* src, const int32_t* idx, int64_t n, int k, float* dst, int device, void* stream);

/* ---- factor-once / re-solve direct solver (nested dissection, multifrontal; symbolic analysis: csrc/nd_plan.cpp,
 *      numeric factorisation: csrc/nd_factor.hip, re-solve: csrc/direct.hip + csrc/nd_tier.h) ---------
 * The elimination tree is a complete `arity`-ary tree (2, 4 or 8) of `levels` levels; node ids are 1-based and
 * level-major (level l: arity^l nodes, node (l, q) has the children (l+1, arity*q + c)). The vertices are renumbered
 * deepest level first (h_perm[new] = old); node i owns the new ids [own_start, own_start + s) and has b boundary
 * vertices in its ancestors; its front is [own | boundary] and front_off is its offset in the concatenation of all
 * fronts (n_front positions). h_ppos[bnd_off + i] = position of boundary vertex i in the PARENT's front (n_bnd entries).
 * h_push_ptr (n_front + 1) / h_push_tgt (n_bnd): CSR lists, front position -> indices (into the concatenated boundary
 * vectors) of the children's boundary entries that are this vertex.
 * Factor arrays (DEVICE, fp32, owned by the caller and kept alive for the handle's lifetime; the kernels read rows with 16-byte
 * loads that are only 4-byte aligned, so d_finv / d_wf / d_wb need at least 12 readable bytes of slack behind their last entry):
 * d_finv[finv_off + t*s + j] = (F_ss^-1)[t][j]; W = F_bs F_ss^-1 twice: d_wf[w_off + j*b + i] = d_wb[w_off + i*s + j] = W[i][j].
 * h_nodes: (n_nodes + 1, 8) int64, row i = {s, b, own_start, bnd_off, front_off, finv_off, w_off, parent} (row 0 unused).
 * One solve = one launch per upper level upwards
 *     b'_s = b_s - (children's updates at own_i);   upd_i = W_i b'_s + (children's updates at bnd_i)
 * (children push into one slot per child of every parent front position), one launch per upper level downwards
 *     x_s = F_ss^-1 b'_s - W_i^T x[bnd_i]          (parents push x into the children's boundary vectors);
 * the levels stored in the tier layouts run as ONE launch per sweep, a workgroup per subtree (csrc/nd_tier.h).
 * b is read and x written in the caller's numbering. No atomics: bitwise reproducible. A handle owns one workspace:
 * solves issued on different streams are serialised on the device (event wait), concurrent host threads must not
 * share a handle. ls_direct_create is SYNC (copies the host tables). */
/* ---- host-side analysis for the LDS-resident s-step Chebyshev kernel (ls_solver_set_patches; csrc/patch_plan.cpp, host threads, no
 * device): the mesh cut into 2^m equally sized compact patches (recursive coordinate bisection of h_positions, (V, 3)), every patch with
 * its ghost layers 1..depth and patch-local uint16 neighbour ids. h_rowptr / h_col: CSR pattern of M (diagonal included), h_diag: its
 * diagonal. The deepest plan <= depth whose patches keep <= cap_local local vertices (LDS) and <= cap_rows computed rows is built;
 * *out stays NULL (status LS_OK) when even min_depth does not fit. Arrays: table (n_patches x 20 int32: own_start, n_own, n_rows,
 * n_local, W, off_gid, off_cols, off_diag, lim[12]), ghost_gid, cols16 ((W, n_rows) per patch), diag, perm (new -> old, V). */
typedef struct ls_patch_plan ls_patch_plan;
int ls_patch_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_diag, const float* h_positions,
                         int patch_size, int depth, int cap_local, int min_depth, int cap_rows, ls_patch_plan** out);
int ls_patch_plan_destroy(ls_patch_plan* p);
int ls_patch_plan_info(const ls_patch_plan* p, int* n_patches, int* depth, int* max_local, int* max_rows, int* max_width, int64_t* n_gid,
                       int64_t* n_cols, int64_t* n_diag, double* seconds);
int ls_patch_plan_arrays(const ls_patch_plan* p, int32_t* table, int32_t* ghost_gid, uint16_t* cols16, float* diag, int32_t* perm);

/* ---- host-side analysis of a vertex-block shard of the iterative solvers (csrc/shard_plan.cpp; one process per GPU, halo exchange):
 * rank `rank` of P owns the rows [rank V / P, (rank + 1) V / P) of the CSR matrix (h_rowptr, h_col, h_val); with depth s it also computes
 * the ghost layers 1..s-1 redundantly and reads layer s. Local column ids: [owned | computed ghosts | read-only ghosts], each ghost group
 * sorted by global id. Arrays: local rowptr (n_own + n_inner + 1) / col / val (n_entries), ghosts (global ids, n_ghosts), recv3 (n_recv x
 * {source rank, offset in the ghost region, count}), send lists as CSR: send_ptr (n_send + 1), send_dst (n_send), send_ids (owned-row
 * indices, message for message in the RECEIVER's order). ls_shard_layer_sizes: sizes of the ghost layers 1..depth of a block [lo, hi). */
typedef struct ls_shard_plan ls_shard_plan;
int ls_shard_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_val, int P, int rank, int depth,
                         ls_shard_plan** out);
int ls_shard_plan_destroy(ls_shard_plan* s);
int ls_shard_plan_info(const ls_shard_plan* s, int64_t* lo, int64_t* hi, int64_t* n_inner, int64_t* n_ghosts, int64_t* n_entries,
                       int* n_recv, int* n_send, int64_t* n_send_ids);
int ls_shard_plan_arrays(const ls_shard_plan* s, int32_t* rowptr, int32_t* col, float* val, int32_t* ghosts, int32_t* recv3,
                         int32_t* send_ptr, int32_t* send_dst, int32_t* send_ids);
int ls_shard_layer_sizes(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, int64_t lo, int64_t hi, int depth, int64_t* h_sizes);

/* Symbolic analysis alone, on the HOST (no device is touched): the elimination tree and everything static of the solver above
 * for the CSR pattern (h_rowptr, h_col; structurally symmetric) of a V x V matrix. h_positions: (V, 3) vertex positions the
 * geometric bisection runs on (only their spatial order matters), or NULL: graph-distance pseudo-positions are derived from
 * the pattern. smooth: neighbour-averaging passes applied to the positions first (4 is the solver's default). */
typedef struct ls_nd_plan ls_nd_plan;
int ls_nd_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_positions, int leaf_size,
                      int arity, int smooth, ls_nd_plan** out);
/* The same plan as ls_direct_factor forms it: the positions' smoothing and the bisection rounds run ON THE DEVICE (csrc/nd_bisect.hip:
 * three coordinate orders by radix sort, a stable partition per round), the rest on the host. d_positions may be NULL. Bit-identical
 * to ls_nd_plan_create on the same input -- the GPU tests compare the two. SYNC. */
int ls_nd_plan_create_device(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                             int leaf_size, int arity, int smooth, int device, void* stream, ls_nd_plan** out);
/* How the cutting direction of a domain is chosen (replaces the graph-based fill-reducing ordering CHOLMOD runs behind the reference's
 * constructor, largesteps/solvers.py:34 -- which does not depend on how the surface lies in space):
 *   LS_ND_ORDER_LONGEST (0)  median cut along the longest axis of the domain's bounding box in the embedding (positions, or graph
 *                            distances without positions) -- what ls_nd_plan_create / ls_nd_plan_create_device run;
 *   LS_ND_ORDER_MINSEP  (1)  every domain tries the three position axes AND three graph distances and takes the thinnest separator
 *                            (host threads): a surface that is folded or rolled up in space, or shells inside each other, dissect
 *                            like the flat sheet (a cutting plane crosses every layer, a level set of a graph distance crosses one);
 *   LS_ND_ORDER_AUTO   (-1)  LONGEST first; if its separators are thicker than a surface's should be (spread > 1.3, below), MINSEP
 *                            too, and the plan with fewer factor numbers -- what ls_direct_factor runs (environment LS_ND_ORDER
 *                            overrides, LS_ND_SUSPECT moves the threshold).
 * ls_nd_plan_quality / ls_direct_plan_quality: the rule that built the plan, factor numbers per vertex (sum over the nodes of
 * s^2 + 2 s b, / V), spread = sum of s^2 over the inner nodes / (A x sum of their subtrees' vertex counts) with A = (sum_{j < log2
 * arity} 2^(j/2))^2 -- ~0.7 on a flat sheet at ANY size, 1.0-1.2 on closed surfaces, 1.4 on a sheet folded once, 5-75 on scrolls --,
 * and the factor numbers per vertex of the plan AUTO built and did not take (0: it built one). Any pointer may be NULL. Host only. */
#define LS_ND_ORDER_AUTO (-1)
#define LS_ND_ORDER_LONGEST 0
#define LS_ND_ORDER_MINSEP 1
int ls_nd_plan_create_ordered(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_positions, int leaf_size,
                              int arity, int smooth, int ordering, ls_nd_plan** out);
/* ls_nd_plan_create_device with the ordering rule as an argument: LS_ND_ORDER_MINSEP runs the trial cuts ON THE DEVICE (six sorted lists, a
 * side bit and a cut bit per vertex and direction; the graph distances are breadth-first sweeps on the host) and is bit-identical to
 * ls_nd_plan_create_ordered(..., LS_ND_ORDER_MINSEP) on the same input; LS_ND_ORDER_AUTO is what ls_direct_factor runs. SYNC. */
int ls_nd_plan_create_device_ordered(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                                     int leaf_size, int arity, int smooth, int ordering, int device, void* stream, ls_nd_plan** out);
int ls_nd_plan_quality(const ls_nd_plan* p, int* h_ordering, double* h_words_per_vertex, double* h_spread, double* h_words_other);
int ls_nd_plan_destroy(ls_nd_plan* p);
int ls_nd_plan_info(const ls_nd_plan* p, int* levels, int* arity, int* n_nodes, int64_t* n_bnd, int64_t* n_front, double* seconds);
/* copies out: perm (V), s / b / own_start / parent (n_nodes + 1 each, node ids are 1-based), bnd / ppos / push_tgt (n_bnd),
 * push_ptr (n_front + 1); any pointer may be NULL */
int ls_nd_plan_arrays(const ls_nd_plan* p, int32_t* perm, int32_t* s, int32_t* b, int32_t* own_start, int32_t* parent,
                      int32_t* bnd, int32_t* ppos, int32_t* push_ptr, int32_t* push_tgt);

typedef struct ls_direct ls_direct;
/* The plan and the factor of one matrix, as plain arrays. h_* live on the host and are copied; d_* are DEVICE arrays
 * owned by the caller and kept alive for the handle's lifetime.
 * h_nodes: (n_nodes + 1, LS_DIRECT_NODE_COLS) int64, row i = {s, b, own_start, bnd_off, front_off, finv_off, w_off, parent,
 *          tri_off, spb_off, sps_off, quad} (row 0 unused). The deepest levels may be stored in the layouts of the tier
 *          kernels (csrc/nd_tier.h), which then run them as ONE launch per sweep, a workgroup per subtree:
 *          quad != 0: a dense node whose two matrix streams are quad-interleaved along the reduction (a lane reads 16 bytes
 *              = 4 consecutive reduction entries; reductions padded to multiples of 4 with zeros; s4 = s rounded up to 4):
 *              d_u4[w_off + ((j / 4) * b + i) * 4 + j % 4] = W[i][j]                                        (up sweep)
 *              d_d4[finv_off + ((t / 4) * s + j) * 4 + t % 4] = [F_ss^-1 (padded to s4 columns) | W^T][j][t]  (down sweep)
 *              (finv_off, w_off multiples of 4 floats);
 *          tri_off >= 0: a SPARSE LEAF (last level, 1 <= s <= 64; all leaves with s >= 1 alike): instead of dense arrays
 *              it owns  d_tri[tri_off + r (r + 1) / 2 + c] = (F_ss^-1)[r][c], c <= r  (packed rows of the lower triangle,
 *              tri_off a multiple of 4 floats), and its off-diagonal block A_bs of the matrix itself as two CSR lists in
 *              d_sp_ptr / d_sp_ent: boundary row i owns the entries d_sp_ptr[spb_off + i] .. d_sp_ptr[spb_off + i + 1]
 *              ({value, own-row index}), own row j the entries d_sp_ptr[sps_off + j] .. [sps_off + j + 1] ({value, boundary
 *              index}). Up sweep  y = F_ss^-1 b_s, update = A_bs y;  down sweep  x_s = y - F_ss^-1 (A_sb x_bnd).
 *          Tier-layout levels must be the deepest ones, complete levels, at most 6 of them. */
#define LS_DIRECT_NODE_COLS 12
typedef struct ls_direct_arrays {
    int64_t V;
    int32_t levels, arity;
    const int64_t* h_nodes;
    const int32_t* h_perm;
    const int32_t* h_ppos;
    int64_t n_bnd;
    const int32_t* h_push_ptr;
    const int32_t* h_push_tgt;
    int64_t n_front;
    const float* d_finv;
    const float* d_wf;
    const float* d_wb;
    const float* d_u4;
    const float* d_d4;
    const float* d_tri;
    const int32_t* d_sp_ptr;
    const void* d_sp_ent;          /* {float value; int32 index} pairs */
    int64_t n_sp_ptr, n_sp_ent;    /* lengths of the two sparse-leaf arrays (accounting only) */
    int32_t shard_rank, shard_count;   /* subtree sharding over `shard_count` processes (0 or 1: none), see ls_direct_solve_part */
    int32_t tier_waves;                /* waves per tier workgroup: 0 = the library's rule (see ls_direct_options), 4 / 8 / 16 */
} ls_direct_arrays;
int ls_direct_create(const ls_direct_arrays* arrays, int device, void* stream, ls_direct** out);
/* The tree ls_direct_factor picks for a V x V system: on entry *leaf_size / *arity <= 0 mean "pick" (explicit values are kept), on return
 * both hold what the factorisation will use. Host only (no device is touched): the rule is a table of measured crossovers
 * (profiles/r03_leaf_size_sweep.txt), documented in DESIGN.md section 2.3 (table: docs/history_rounds_1_4.md). */
int ls_direct_pick_tree(int64_t V, int* leaf_size, int* arity);
/* Host only: the dynamic LDS (bytes) one workgroup of the tier kernels needs to walk a subtree of the `tier_levels` deepest levels of a plan
 * (h_s / h_b / h_own_start: per node id, 1-based, level-major -- what ls_nd_plan_arrays returns), with `waves` = 4, 8 or 16 waves per workgroup;
 * 0 = such a tier cannot be planned (a level mixing sparse and dense leaves). ls_direct_factor takes the 16-wave tier only while this is <= 160 KB
 * and the 4-wave one while it is <= 150 KB. sparse_leaves != 0: leaves of at most 64 rows are stored as triangle + sparse block. */
int ls_direct_tier_lds_bytes(int levels, int arity, const int32_t* h_s, const int32_t* h_b, const int32_t* h_own_start, int tier_levels,
                             int sparse_leaves, int waves, size_t* h_bytes);
/* Matrix in, solver out: symbolic analysis (bisection rounds on the device, csrc/nd_bisect.hip; tree, fronts and index lists on
 * host threads), numeric multifrontal factorisation in fp64 on the device with hand-written kernels (csrc/nd_factor.hip: products
 * on the fp64 matrix instruction, SPD inverses in registers), fp32 factor in the solve kernels' layouts, handle. This one call is the
 * constructor of the reference's default solver (largesteps/solvers.py:34, CholeskySolverF(n, ii, jj, x, MatrixType.COO)).
 * d_rowptr / d_col / d_val: CSR of the symmetric positive definite matrix (DEVICE, original numbering, column-sorted rows);
 * d_positions: (V, 3) fp32 vertex positions (DEVICE) or NULL (graph-distance pseudo-positions);
 * arity 2 / 4 / 8 = children per tree node (1 / 2 / 3 bisection rounds per level) and leaf_size = the largest leaf, or <= 0 = chosen
 *   by the library from V (ls_direct_pick_tree above; small and medium systems are bound by their chain of launches, not by bytes):
 *   one dense node up to 1280 unknowns (ONE launch per re-solve), arity 4 with leaves of up to 1024 up to 12k, arity 8 between 12k and
 *   300k (three levels of dense nodes up to 36k, four levels with dense leaves of 70-205 rows up to 105k, five levels with 64-vertex
 *   sparse leaves beyond), arity 4 with 64-vertex sparse leaves -- the large-mesh setting -- above 300k;
 * tier_levels deepest levels go into the tier layouts (-1 = chosen by the library: tree levels - 5, at least 2 and at most 4; from
 *   800k unknowns on an arity-4 tree of at least 8 levels, one GPU: tree levels - 4, walked by one 16-wave workgroup per CU; the leaf
 *   level alone for many dense leaves of 65-224 rows in a tree of at most 4 levels; none for leaves of more than 256 rows or a single
 *   node; 0 = none); sparse_leaves != 0 stores leaves of at most 64 rows as packed triangle + sparse block;
 * shard_rank / shard_count: subtree sharding (0 / 1: none; every rank factorises the whole matrix, the re-solve is sharded, see
 *   ls_direct_solve_part).
 * The host half of the analysis runs on a pool of threads (environment LS_PLAN_THREADS, default 32, at most the host's cores divided
 * by LOCAL_WORLD_SIZE) that is created by the first call and kept, asleep, for the life of the process; a call made while another
 * thread's call holds the pool, or from a forked child, uses threads of its own.
 * SYNC. Errors: LS_E_INVALID (not symmetric / not positive definite / bad arguments), LS_E_WORKSPACE (fronts or factor beyond the
 * solver's limits; or an EXPLICIT tier_levels whose subtrees do not fit a workgroup's LDS -- tier_levels = -1 lowers its own choice
 * until it fits and never fails for that reason). */
int ls_direct_factor(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                     const float* d_positions, int leaf_size, int arity, int tier_levels, int sparse_leaves, int shard_rank,
                     int shard_count, int device, void* stream, ls_direct** out);
/* The same constructor with every choice as an argument (round 6; `ordering` used to reach the library through the process
 * environment only). Fill the struct with ls_direct_options_default, change what you need:
 *   leaf_size, arity, tier_levels, sparse_leaves, shard_rank, shard_count: as ls_direct_factor's arguments (defaults 0, 0, -1, 1, 0, 1);
 *   ordering    how the bisection picks its cutting directions: LS_ND_ORDER_AUTO (the library's rule: trial cuts where the separators
 *               are thicker than a surface's should be), LS_ND_ORDER_LONGEST, LS_ND_ORDER_MINSEP (trial cuts always: 5-10 % fewer factor
 *               numbers on rough closed scans for 10-25 ms more constructor). An explicit value wins; AUTO lets the environment
 *               variable LS_ND_ORDER override the rule (A/B runs);
 *   tier_waves  waves per workgroup of the tier kernels: 0 = the library's rule (16 on one workgroup per CU and a subtree one level
 *               taller from 800k unknowns, 8 for <= 768 subtrees, 4 otherwise), or 4 / 8 / 16. An explicit value wins; 0 lets
 *               LS_ND_TIER_WAVES override.
 * struct_bytes = sizeof(ls_direct_options) of the CALLER's header: fields a newer library knows beyond it take their defaults.
 * opt == NULL: all defaults (= ls_direct_factor(..., 0, 0, -1, 1, 0, 1, ...)). */
typedef struct ls_direct_options {
    int32_t struct_bytes;
    int32_t leaf_size, arity, tier_levels, sparse_leaves, shard_rank, shard_count;
    int32_t ordering;
    int32_t tier_waves;
} ls_direct_options;
int ls_direct_options_default(ls_direct_options* opt);
int ls_direct_factor_ex(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz,
                        const float* d_positions, const ls_direct_options* opt, int device, void* stream, ls_direct** out);
/* The direct solver keeps its device buffers (>= 256 KB: the constructor's fp64 fronts and work arrays -- 3-4 GB at 1M vertices, 14 GB
 * at 4M --, the analysis' scratch, a destroyed handle's factor arrays, index tables and vectors) in a per-process pool instead of freeing them: a remesh loop
 * (scripts/main.py:137-169) destroys a solver and constructs one of nearly the same size again and again, and the runtime gives freed
 * memory back lazily -- at the 4M size every second or third construction stalled 0.45-0.7 s inside one hipMalloc. The pool holds at most
 * LS_POOL_GB per device (environment, default 24; 0 = no pool; oldest out first) and never more than a quarter of the device's memory. An
 * allocation of the library that fails for lack of memory empties the pool of its device and is repeated once. This call frees what the
 * pool holds (device < 0: on every device) -- for callers whose OWN allocator ran out (torch's caching allocator cannot see the pool). */
int ls_release_scratch(int device);
/* tree levels, arity, levels run by the tier kernels and their workgroup count, 4-byte words of factor data the up / the
 * down sweep reads, total boundary entries (any pointer may be NULL) */
int ls_direct_shape(const ls_direct* d, int* h_levels, int* h_arity, int* h_tier_levels, int* h_tier_workgroups,
                    int64_t* h_words_up, int64_t* h_words_down, int64_t* h_n_bnd);
/* 4-byte words of factor data each tree level reads in the up / the down sweep (h_up, h_down: `cap` entries, level 0 = root;
 * the sparse leaves' lists are counted with the last level). Host only. */
int ls_direct_level_words(const ls_direct* d, int cap, int64_t* h_up, int64_t* h_down);
/* own rows (vertices) and boundary entries of each tree level (level 0 = root): what the vectors a level's launch moves are made
 * of -- per sweep b / b' / x rows of its vertices and the boundary vectors of its nodes. Host only. */
int ls_direct_level_rows(const ls_direct* d, int cap, int64_t* h_rows, int64_t* h_bnd);
/* bytes of STATIC index data each tree level's launch reads per sweep, besides the factor words (ls_direct_level_words) and the vectors
 * (ls_direct_level_rows): tile / item records, children masks, parent positions, the tier's pull lists (arity indices per front position
 * of an inner node), push-list pointers and targets. perm (4 bytes per own row and sweep) is counted with the vectors. Host only. */
int ls_direct_level_index_bytes(const ls_direct* d, int cap, int64_t* h_up, int64_t* h_down);
/* how evenly the tier's subtrees (one workgroup each) are loaded: h_words4 = {max, mean} factor words of a subtree in the up sweep,
 * {max, mean} in the down sweep. With one workgroup per CU the heaviest subtree is the launch's time. Zeros without a tier. Host only. */
int ls_direct_tier_balance(const ls_direct* d, double* h_words4);
/* After a solve with "profile" = 3 (an event in front of every launch; the solve synchronises): number of launches and, for
 * the first `cap` of them, duration in ms, factor words read, tree levels [lo, hi] covered, sweep (0 up, 1 down, 2 both).
 * Any pointer may be NULL. */
int ls_direct_launch_profile(const ls_direct* d, int cap, int* h_n, double* h_ms, int64_t* h_words, int32_t* h_level_lo,
                             int32_t* h_level_hi, int32_t* h_sweep);
/* the dissection behind a handle made by ls_direct_factor (see ls_nd_plan_quality) */
int ls_direct_plan_quality(const ls_direct* d, int* h_ordering, double* h_words_per_vertex, double* h_spread, double* h_words_other);
/* seconds of the three constructor stages of a handle made by ls_direct_factor: symbolic analysis, layout tables, numeric */
int ls_direct_factor_seconds(const ls_direct* d, double* h_s3);
/* SYNC: *h_symmetric = 1 iff every stored entry (r, c, v) has a stored mirror (c, r, v') with |v - v'| <= tol */
int ls_csr_is_symmetric(const int32_t* d_rowptr, const int32_t* d_col, const float* d_val, int64_t V, int64_t nnz, float tol,
                        int* h_symmetric, int device, void* stream);
int ls_direct_destroy(ls_direct* d);
/* x = M^-1 b for k <= 4 interleaved columns ((V, k) row-major, b != x) */
int ls_direct_solve(ls_direct* d, const float* b, float* x, int k, void* stream);
/* Subtree sharding, one process per GPU (largesteps/distributed.py ShardedDirect): a handle created with shard_count = N > 1
 * runs the subtrees of the first tree level that has >= N of them ("cut" level; rank r takes a contiguous share) and,
 * replicated on every rank, the levels above the cut. One solve =
 *     part 0   this rank's subtrees upwards; their updates for level cut - 1 land in `exchange` (floats_per_column x k floats,
 *              zero where another rank's subtree contributes)
 *     the caller SUMS `exchange` over the ranks (one all-reduce of a few hundred KB; every entry has exactly one non-zero
 *     contributor, so the sum is exact and order independent)
 *     part 1   the replicated levels up and down, then this rank's subtrees downwards: x rows of the rank's subtrees and of
 *              the replicated levels are written, other rows of x are left untouched.
 * h_owned_rows (V bytes, may be NULL): 1 where this rank is the designated owner of the row (replicated levels: rank 0). */
int ls_direct_solve_part(ls_direct* d, const float* b, float* x, int k, int part, float* exchange, void* stream);
int ls_direct_shard_info(const ls_direct* d, int* h_rank, int* h_count, int* h_cut_level, int64_t* h_exchange_floats_per_column,
                         unsigned char* h_owned_rows);
/* the region of the handle's own workspace that holds the exchange of a k-column solve (what part 0 leaves and part 1 reads):
 * passing it as `exchange` to ls_direct_solve_part skips the staging copies -- the caller then reduces it in place */
int ls_direct_exchange_region(ls_direct* d, int k, float** h_region, int64_t* h_floats);

/* ---- the collective of the sharded solve behind the C ABI (one process per GPU; RCCL over xGMI, looked up at run time) --------------
 * ls_dist_unique_id: rank 0 creates a 128-byte communicator id and ships it to the other ranks (any transport: the caller's);
 * ls_dist_create: COLLECTIVE, every rank calls it with the same id (ncclCommInitRank on `device`);
 * ls_dist_allreduce_sum: in-place sum of n floats over the ranks, ASYNC on `stream`;
 * ls_dist_direct_solve: one sharded solve = ls_direct_solve_part(0), the all-reduce of the exchange region in place, part (1),
 *   all on `stream` with no host round trip in between (a consumer needs no Python and no all-reduce of its own).
 * LS_E_STATE: librccl.so is not loadable; values >= 2000: 2000 + ncclResult_t. */
typedef struct ls_dist ls_dist;
int ls_dist_unique_id(void* h_id128);
int ls_dist_create(const void* h_id128, int rank, int world, int device, ls_dist** out);
int ls_dist_destroy(ls_dist* c);
int ls_dist_allreduce_sum(ls_dist* c, float* d_buf, int64_t n, void* stream);
/* what the RCCL communicator itself reports (ncclCommUserRank / ncclCommCount), not what the caller passed to ls_dist_create */
int ls_dist_info(const ls_dist* c, int* h_rank, int* h_world);
int ls_dist_direct_solve(ls_dist* c, ls_direct* d, const float* b, float* x, int k, void* stream);
/* knobs: "profile" (1: the next solves time the up sweep and the down sweep with HIP events and synchronise; 3: an event in
 * front of every launch, read back by ls_direct_launch_profile);
 * "nt" (cache policy of the read-once factor streams: 1 non-temporal loads, 0 default policy, -1 the library's rule -- non-temporal
 * once the factor exceeds what the 256 MB Infinity Cache keeps from solve to solve; results are bit-identical either way) */
int ls_direct_set(ls_direct* d, const char* name, int value);
/* host-only: 4-byte words of factor data one solve reads, kernel launches per solve, {up sweep, down sweep, 0} ms of the
 * last profiled solve (any pointer may be NULL) */
int ls_direct_info(const ls_direct* d, int64_t* h_factor_entries, int* h_launches, double* h_ms3);

/* ---- remove_duplicates (reference scripts/geometry.py:3-11) --------------------------------------
 * unique_verts = the distinct rows of verts ((V, 3) fp32) in lexicographic order of their VALUES (-0.0 == 0.0), exactly what
 * torch.unique(v, dim=0, return_inverse=True) returns; inverse (V) int64: verts[i] == unique_verts[inverse[i]];
 * new_faces (F, 3) int64 = inverse[faces] (faces int32 / int64 by idx_bytes; F may be 0). unique_verts needs room for V rows;
 * *h_n_unique receives the number of rows written. Hand-written stable LSD radix sort (no library sort). SYNC.
 * A face index outside [0, V) -> LS_E_INDEX. */
int ls_remove_duplicates_workspace_bytes(int64_t V, size_t* h_bytes);
int ls_remove_duplicates(const float* verts, int64_t V, const void* faces, int idx_bytes, int64_t F, float* unique_verts,
                         int64_t* inverse, int64_t* new_faces, int64_t* h_n_unique, void* workspace, size_t ws_bytes, int device,
                         void* stream);

/* CSR of the transpose (rows of M^T sorted by column = original row), for the backward pass of to_differential on a matrix
 * that is not symmetric. Hand-written radix sort of the entry ids by column. SYNC (range check of the column indices). */
int ls_csr_transpose_workspace_bytes(int64_t V, int64_t nnz, size_t* h_bytes);
int ls_csr_transpose(const int32_t* rowptr, const int32_t* col, const float* val, int64_t V, int64_t nnz, int32_t* t_rowptr,
                     int32_t* t_col, float* t_val, void* workspace, size_t ws_bytes, int device, void* stream);
/* Vertex-major ranking of the 3 F face corners (see the normals below): vptr (V + 1) = first rank of every vertex, cpos (3 F) =
 * rank of corner 3 f + i; the corners of a vertex are ranked in ascending corner id. A face index outside [0, V) -> LS_E_INDEX.
 * SYNC. */
int ls_corner_ranks_workspace_bytes(int64_t F, int64_t V, size_t* h_bytes);
int ls_corner_ranks(const void* faces, int idx_bytes, int64_t F, int64_t V, int32_t* vptr, int32_t* cpos, void* workspace,
                    size_t ws_bytes, int device, void* stream);

/* ---- normals ----------------------------------------------------------------------
 * faces: (F, 3) int32 or int64 (idx_bytes 4 / 8), verts (V, 3) fp32. Face normals are (3, F) like the reference returns
 * them. Semantics of scripts/geometry.py kept to the letter: a degenerate face / an unreferenced vertex gives NaN, and
 * the corner "angles" are acos(clamp(e_a . e_b / (||E_a||_F ||E_b||_F))) with the FROBENIUS norms of the whole edge
 * matrices (geometry.py:138-141), not per-face lengths. Indices are not range checked (the host wrapper does that).
 * vptr (V + 1) / cpos (3 F): corner 3 f + i has rank cpos[3 f + i] in the vertex-major order of all corners and vertex
 * v owns the ranks [vptr[v], vptr[v + 1]); the per-face kernels store one vector per corner at its rank and a
 * per-vertex kernel sums its contiguous range in rank order -- no atomics, bitwise reproducible for a given ranking. */
int ls_normals_workspace_bytes(int64_t F, int64_t V, size_t* h_bytes);
int ls_face_normals(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, float* fn, int device, void* stream);
/* grad_verts (V, 3) = d sum(g_fn * fn) / d verts; overwritten */
int ls_face_normals_backward(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                             const int32_t* cpos, const float* g_fn, float* grad_verts, void* workspace, size_t ws_bytes,
                             int device, void* stream);
/* out (V, 3) normalised vertex normals; raw (V, 3) the unnormalised sums and norms[3] = {||E01||, ||E02||, ||E12||} are
 * kept by the caller for the backward pass */
int ls_vertex_normals(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                      const int32_t* cpos, const float* fn, float* out, float* raw, float* norms, void* workspace,
                      size_t ws_bytes, int device, void* stream);
/* grad_verts (V, 3) and grad_fn (3, F) of sum(g_out * out) with the face normals an independent input; both overwritten */
int ls_vertex_normals_backward(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                               const int32_t* cpos, const float* fn, const float* raw, const float* norms, const float* g_out,
                               float* grad_verts, float* grad_fn, void* workspace, size_t ws_bytes, int device, void* stream);
/* The PAIR compute_face_normals -> compute_vertex_normals on one mesh (scripts/main.py:178-179: the face normals handed to
 * compute_vertex_normals ARE the normalised cross products of the same vertices). Then the two are one function of the
 * vertices and the passes share work: the face-normal pass also reduces the three edge norms, later passes recompute n_f from
 * the positions they load anyway instead of reading (3, F), and the backward stores the corner buffer once:
 *   forward    ls_face_normals_with_norms (fn, norms[3])  ->  ls_vertex_normals_from_norms (out, raw)
 *   backward   ls_normals_pair_backward_faces: g_raw (V, 3) = gradient of the unnormalised sums, gN[3] = dL/d(edge norms),
 *              grad_fn (3, F) = what reaches the face normals through the vertex normals -- an OUTPUT of the pair, the
 *              caller adds what other consumers of the face normals contribute -- then
 *              ls_normals_pair_backward_verts: grad_verts (V, 3) of everything, g_fn = that total (nullptr: none).
 * Same values as the separate calls up to the order of a few additions. g_raw / gN are the caller's (they live between the
 * two backward calls); workspace as above.
 * ls_vertex_normals_gathered = ls_vertex_normals_from_norms bit for bit, vertex-major: a thread per vertex walks its corners in
 * rank order and recomputes their contributions (no corner buffer, no workspace, one launch instead of two: 32 against 46 us at
 * 1M vertices). order[3 F] is the inverse permutation of cpos (order[cpos[c]] = c: rank -> corner id 3 f + i).
 * PRECONDITION of ls_vertex_normals_gathered (NOT checked on the device, unlike ls_corner_ranks / ls_assemble_*, which validate and
 * return LS_E_INDEX): vptr / order must be what ls_corner_ranks returned FOR THESE faces, i.e. every face index lies in [0, V) and
 * every order entry in [0, 3 F) -- the kernel indexes verts through them without a range check. */
int ls_face_normals_with_norms(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, float* fn, float* norms,
                               void* workspace, size_t ws_bytes, int device, void* stream);
int ls_vertex_normals_from_norms(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                 const int32_t* cpos, const float* norms, float* out, float* raw, void* workspace, size_t ws_bytes,
                                 int device, void* stream);
int ls_vertex_normals_gathered(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                               const int32_t* order, const float* norms, float* out, float* raw, int device, void* stream);
int ls_normals_pair_backward_faces(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const float* raw,
                                   const float* norms, const float* g_out, float* g_raw, float* gN, float* grad_fn, void* workspace,
                                   size_t ws_bytes, int device, void* stream);
int ls_normals_pair_backward_verts(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                   const int32_t* cpos, const float* norms, const float* g_raw, const float* gN, const float* g_fn,
                                   float* grad_verts, void* workspace, size_t ws_bytes, int device, void* stream);

/* ------------------------------------------------------------------------------------------------
 * AdamUniform step (optimize.py:18-41) on n contiguous fp32 elements, two kernels, no host sync:
 *   g1 = b1 g1 + (1-b1) g ; g2 = b2 g2 + (1-b2) g^2 ; p -= lr * (g1/(1-b1^t)) / (1e-8 + max sqrt(g2/(1-b2^t)))
 * scratch: at least 4096 bytes of device memory owned by the caller. ASYNC.
 * --------------------------------------------------------------------------------------------- */
int ls_adam_uniform_step(float* param, const float* grad, float* g1, float* g2, int64_t n, float lr,
                         float beta1, float beta2, int step, void* scratch, int device, void* stream);
/* The same step with the step count ON THE DEVICE: d_step[0] = steps done so far (int32, zero before the first call; the call
 * increments it), d_step[1] = scratch. No argument changes from step to step, so the two launches can sit in a captured
 * graph (torch.cuda.graph around a whole optimisation step) and be replayed. ASYNC. */
int ls_adam_uniform_step_device(float* param, const float* grad, float* g1, float* g2, int64_t n, float lr, float beta1,
                                float beta2, int32_t* d_step, void* scratch, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LARGESTEPS_HIP_H */
