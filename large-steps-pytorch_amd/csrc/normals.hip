// normals.hip -- per-face and per-vertex normals with their gradients (gfx950, wave64).
//
// the consumer right after from_differential in every optimisation step
// (rgl-epfl/large-steps-pytorch scripts/main.py:178-179 -> scripts/geometry.py:91-147). The reference builds these from
// ~40 stock torch kernels per call and lets autograd replay them; here the forward is 4 launches and the backward 4.
//
// Reference semantics kept to the letter (scripts/geometry.py):
//   compute_face_normals :91-110     n_f = c / |c|, c = (v1 - v0) x (v2 - v0), returned as (3, F); a degenerate face gives NaN
//   compute_vertex_normals :115-147  corner i of face f: d0 = v[i+1] - v[i], d1 = v[i+2] - v[i], each divided by the
//                                    FROBENIUS norm of the whole (3, F) edge matrix (`d0 / torch.norm(d0)`, :138-141: a
//                                    global scalar, not a per-face length), angle = acos(clamp(sum(d0 * d1), -1, 1)),
//                                    normals[f_i] += n_f * angle; result normalised per vertex, (V, 3); an unreferenced
//                                    vertex gives NaN (0 / 0), as in the reference.
// Three distinct global norms exist: ||E01||, ||E02||, ||E12|| (E_ab = all faces' edges v_b - v_a), each used by two corners.
// No atomics: per-face kernels write one 3-vector per CORNER (coalesced), a per-vertex kernel sums the corners of a vertex
// through a vertex-major corner ranking built once per face tensor (measured at 2M faces: 18M fp32 atomics took 0.44 ms per
// scatter, the two-pass form ~0.05 ms) -- and the result is bitwise reproducible, unlike the reference's index_add_.
#include "common.h"
#include <algorithm>

namespace ls {

constexpr int NRM_MAXG = 1024;      // partial sums per reduction (grid of the reducing kernels is capped to this)

template <typename IDX>
__device__ __forceinline__ void load_face(const IDX* __restrict__ faces, int64_t f, const float* __restrict__ verts,
                                          int (&id)[3], float (&p)[3][3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        id[c] = (int)faces[f * 3 + c];
#pragma unroll
        for (int q = 0; q < 3; ++q) p[c][q] = verts[(size_t)id[c] * 3 + q];
    }
}

__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&c)[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_face_normals(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                        float* __restrict__ fn) {
    const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3], a[3], b[3], c[3];
    load_face(faces, f, verts, id, p);
#pragma unroll
    for (int q = 0; q < 3; ++q) { a[q] = p[1][q] - p[0][q]; b[q] = p[2][q] - p[0][q]; }
    cross3(a, b, c);
    const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) fn[(size_t)q * F + f] = c[q] / len;
}

// d/dv of sum(g * n), n = c / |c|, c = a x b: g_c = (g - n (n.g)) / |c|, dL/da = b x g_c, dL/db = g_c x a
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_face_normals_bwd(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                            const float* __restrict__ g_fn, const int* __restrict__ cpos,
                                                            float* __restrict__ corner) {
    const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3], a[3], b[3], c[3], g[3], gc[3], ga[3], gb[3];
    load_face(faces, f, verts, id, p);
#pragma unroll
    for (int q = 0; q < 3; ++q) { a[q] = p[1][q] - p[0][q]; b[q] = p[2][q] - p[0][q]; g[q] = g_fn[(size_t)q * F + f]; }
    cross3(a, b, c);
    const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    float n[3], ng = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) { n[q] = c[q] / len; ng += n[q] * g[q]; }
#pragma unroll
    for (int q = 0; q < 3; ++q) gc[q] = (g[q] - n[q] * ng) / len;
    cross3(b, gc, ga);
    cross3(gc, a, gb);
    const size_t p0 = (size_t)cpos[f * 3] * 3, p1 = (size_t)cpos[f * 3 + 1] * 3, p2 = (size_t)cpos[f * 3 + 2] * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        corner[p1 + q] = ga[q];
        corner[p2 + q] = gb[q];
        corner[p0 + q] = -ga[q] - gb[q];
    }
}

// dst[v] = sum of the corner vectors of vertex v: the per-face kernels store the vector of corner 3 f + i at slot
// cpos[3 f + i], the corner's rank in vertex-major order, so a vertex's vectors are the contiguous slots
// [vptr[v], vptr[v + 1]) -- consecutive threads read consecutive memory. normalize: also writes the unit vector to `out`.
__global__ __launch_bounds__(BLOCK) void k_gather_corners(const int* __restrict__ vptr, const float* __restrict__ corner, int64_t V,
                                                          float* __restrict__ dst, float* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    const int e0 = vptr[v], e1 = vptr[v + 1];
    // the first GC corners (a vertex of a triangle mesh has ~6) are requested together, from clamped addresses (no branch between
    // the loads); the sum runs over them in rank order as before
    constexpr int GC = 8;
    float c[GC][3];
    const int last = max(e1 - 1, e0);            // e1 == e0 (unreferenced vertex): a valid address, the value is not used
#pragma unroll
    for (int t = 0; t < GC; ++t) {
        const size_t e = (size_t)min(e0 + t, last);
#pragma unroll
        for (int q = 0; q < 3; ++q) c[t][q] = corner[e * 3 + q];
    }
    float x = 0.0f, y = 0.0f, z = 0.0f;
#pragma unroll
    for (int t = 0; t < GC; ++t) {
        if (e0 + t < e1) { x += c[t][0]; y += c[t][1]; z += c[t][2]; }
    }
    for (int e = e0 + GC; e < e1; ++e) {
        x += corner[(size_t)e * 3]; y += corner[(size_t)e * 3 + 1]; z += corner[(size_t)e * 3 + 2];
    }
    dst[v * 3] = x; dst[v * 3 + 1] = y; dst[v * 3 + 2] = z;
    if (out) {
        const float len = sqrtf(x * x + y * y + z * z);
        out[v * 3] = x / len; out[v * 3 + 1] = y / len; out[v * 3 + 2] = z / len;
    }
}

// sum over the workgroup of three doubles (result in thread 0)
__device__ __forceinline__ void block_sum3(double (&x)[3], double* smem) {
    block_sum<3>(x, smem);
}

// partial sums of |e01|^2, |e02|^2, |e12|^2 over the faces of this workgroup's stride
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_edge_norm_partials(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                              double* __restrict__ part) {
    __shared__ double smem[3 * (BLOCK / WAVE)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x; f < F; f += (int64_t)gridDim.x * BLOCK) {
        int id[3];
        float p[3][3];
        load_face(faces, f, verts, id, p);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float e01 = p[1][q] - p[0][q], e02 = p[2][q] - p[0][q], e12 = p[2][q] - p[1][q];
            acc[0] += (double)(e01 * e01); acc[1] += (double)(e02 * e02); acc[2] += (double)(e12 * e12);
        }
    }
    block_sum3(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) part[(size_t)i * NRM_MAXG + blockIdx.x] = acc[i];
    }
}

// out[i] = (take_sqrt ? sqrt : id)(sum of the G partials of slot i), i < 3, one workgroup of FIN threads, fixed order; P = slots per
// row of `part`. Everything a thread adds is requested before the first addition (the launch is one round trip + a tree).
constexpr int FIN = 1024, FIN_U = 4;
__global__ __launch_bounds__(FIN) void k_finish3(const double* __restrict__ part, int G, int P, int take_sqrt, float* __restrict__ out) {
    __shared__ double smem[3 * (FIN / WAVE)];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int g0 = threadIdx.x; g0 < G; g0 += FIN * FIN_U) {
        double v[FIN_U][3];
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
            const int g = g0 + u * FIN;
#pragma unroll
            for (int i = 0; i < 3; ++i) v[u][i] = g < G ? part[(size_t)i * P + g] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < FIN_U; ++u) {
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] += v[u][i];
        }
    }
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc[i] = wave_sum(acc[i]);
        if (lane == 0) smem[w * 3 + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double t = 0.0;
            for (int j = 0; j < FIN / WAVE; ++j) t += smem[j * 3 + i];
            out[i] = take_sqrt ? sqrtf((float)t) : (float)t;
        }
    }
}

// corner i of a face: edges e_a = v[i+1] - v[i], e_b = v[i+2] - v[i]; the global norms they are divided by
// (index into norms[]: 0 = ||E01||, 1 = ||E02||, 2 = ||E12||); s = sum((e_a / N_a) * (e_b / N_b)), theta = acos(clamp(s)).
// The kernels are bound by their arithmetic, not by bytes (2M faces x 3 corners x ~25 IEEE divisions were a third of it): the
// three reciprocal norms are formed once per thread and s = (e_a . e_b) * (1 / N_a) * (1 / N_b) -- the reference's value up to a
// few ulp of s (its own fp32 sum has the same freedom), far inside the 1e-6 the parity tests allow on the normals.
struct InvNorms { float N[3], inv[3]; };
__device__ __forceinline__ InvNorms inv_norms(const float* __restrict__ norms) {
    InvNorms n;
#pragma unroll
    for (int i = 0; i < 3; ++i) { n.N[i] = norms[i]; n.inv[i] = 1.0f / n.N[i]; }
    return n;
}
struct Corner { float ea[3], eb[3], iab, s, theta; int na, nb; };     // iab = 1 / (N_a N_b)
__device__ __forceinline__ Corner corner_of(const float (&p)[3][3], int i, const InvNorms& nr) {
    Corner c;
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    c.na = i == 0 ? 0 : (i == 1 ? 2 : 1);
    c.nb = i == 0 ? 1 : (i == 1 ? 0 : 2);
    c.iab = nr.inv[c.na] * nr.inv[c.nb];
    float d = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        c.ea[q] = p[i1][q] - p[i][q];
        c.eb[q] = p[i2][q] - p[i][q];
        d += c.ea[q] * c.eb[q];
    }
    c.s = d * c.iab;
    c.theta = acosf(fminf(fmaxf(c.s, -1.0f), 1.0f));
    return c;
}

template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_vertex_normals_scatter(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                                  const float* __restrict__ fn, const float* __restrict__ norms,
                                                                  const int* __restrict__ cpos, float* __restrict__ corner) {
    const InvNorms nr = inv_norms(norms);
    const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3], n[3];
    load_face(faces, f, verts, id, p);
#pragma unroll
    for (int q = 0; q < 3; ++q) n[q] = fn[(size_t)q * F + f];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Corner c = corner_of(p, i, nr);
#pragma unroll
        for (int q = 0; q < 3; ++q) corner[(size_t)cpos[f * 3 + i] * 3 + q] = n[q] * c.theta;
    }
}

// g_raw = (g - o (o.g)) / |raw|, o = raw / |raw|
__global__ __launch_bounds__(BLOCK) void k_normalize_rows_bwd(const float* __restrict__ raw, const float* __restrict__ g, int64_t V,
                                                              float* __restrict__ g_raw) {
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    const float x = raw[v * 3], y = raw[v * 3 + 1], z = raw[v * 3 + 2];
    const float len = sqrtf(x * x + y * y + z * z);
    const float o[3] = {x / len, y / len, z / len};
    const float gg[3] = {g[v * 3], g[v * 3 + 1], g[v * 3 + 2]};
    const float og = o[0] * gg[0] + o[1] * gg[1] + o[2] * gg[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) g_raw[v * 3 + q] = (gg[q] - o[q] * og) / len;
}

// d theta / d s through acos(clamp(s, -1, 1)): -1 / sqrt(1 - s^2) inside the clamp, 0 outside
__device__ __forceinline__ float dtheta_ds(float s) { return fabsf(s) < 1.0f ? -rsqrtf(1.0f - s * s) : 0.0f; }

// pass 1 of the vertex-normal backward: grad of the face normals (no scatter) and the partial sums of dL/dN for the
// three global norms
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_vertex_normals_bwd1(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                               const float* __restrict__ fn, const float* __restrict__ norms,
                                                               const float* __restrict__ g_raw, float* __restrict__ grad_fn,
                                                               double* __restrict__ part) {
    const InvNorms nr = inv_norms(norms);
    __shared__ double smem[3 * (BLOCK / WAVE)];
    double gN[3] = {0.0, 0.0, 0.0};
    for (int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x; f < F; f += (int64_t)gridDim.x * BLOCK) {
        int id[3];
        float p[3][3], n[3], gf[3] = {0.0f, 0.0f, 0.0f};
        load_face(faces, f, verts, id, p);
#pragma unroll
        for (int q = 0; q < 3; ++q) n[q] = fn[(size_t)q * F + f];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const Corner c = corner_of(p, i, nr);
            float gth = 0.0f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float gr = g_raw[(size_t)id[i] * 3 + q];
                gf[q] += c.theta * gr;
                gth += n[q] * gr;
            }
            const float gs = gth * dtheta_ds(c.s);
            gN[c.na] += (double)(gs * (-c.s * nr.inv[c.na]));
            gN[c.nb] += (double)(gs * (-c.s * nr.inv[c.nb]));
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) grad_fn[(size_t)q * F + f] = gf[q];
    }
    block_sum3(gN, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) part[(size_t)i * NRM_MAXG + blockIdx.x] = gN[i];
    }
}

// pass 2: gradient of the vertices -- through the corner dot products and through the three global norms (gN)
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_vertex_normals_bwd2(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                               const float* __restrict__ fn, const float* __restrict__ norms,
                                                               const float* __restrict__ g_raw, const float* __restrict__ gN,
                                                               const int* __restrict__ cpos, float* __restrict__ corner) {
    const InvNorms nr = inv_norms(norms);
    const int64_t f = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3], n[3], gv[3][3];
    load_face(faces, f, verts, id, p);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) gv[i][q] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) n[q] = fn[(size_t)q * F + f];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Corner c = corner_of(p, i, nr);
        float gth = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q) gth += n[q] * g_raw[(size_t)id[i] * 3 + q];
        const float w = gth * dtheta_ds(c.s) * c.iab;
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float ga = w * c.eb[q], gb = w * c.ea[q];      // d s / d e_a = e_b / (Na Nb), d s / d e_b = e_a / (Na Nb)
            gv[i1][q] += ga; gv[i2][q] += gb; gv[i][q] -= ga + gb;
        }
    }
    // N_ab = sqrt(sum over all faces |v_b - v_a|^2): dN/d(v_b - v_a) = (v_b - v_a) / N
    const float w01 = gN[0] * nr.inv[0], w02 = gN[1] * nr.inv[1], w12 = gN[2] * nr.inv[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float e01 = p[1][q] - p[0][q], e02 = p[2][q] - p[0][q], e12 = p[2][q] - p[1][q];
        gv[1][q] += w01 * e01; gv[0][q] -= w01 * e01;
        gv[2][q] += w02 * e02; gv[0][q] -= w02 * e02;
        gv[2][q] += w12 * e12; gv[1][q] -= w12 * e12;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) corner[(size_t)cpos[f * 3 + i] * 3 + q] = gv[i][q];
    }
}

// ---- the pair compute_face_normals -> compute_vertex_normals on ONE mesh (scripts/main.py:178-179) -------------------------
// When the face normals handed to compute_vertex_normals ARE the normalised cross products of the same vertices, the pair is
// one function of the vertices and the passes can share work: the face-normal pass also accumulates the three edge norms; the
// later passes recompute n_f from the positions they load anyway (the library is built with -ffp-contract=off: the same
// expression gives the same bits as the stored array) instead of reading (3, F); and the backward writes the corner buffer
// ONCE -- the vertex-normal pass and the face-normal pass of the chain rule run as one kernel over the faces.
//
// Workgroup -> block of faces: workgroup b runs on XCD b % 8 (observed placement, speed only). Neighbouring face blocks share
// vertices (gathers) and corner-buffer lines (the 6 corners of a vertex are 72 contiguous bytes written by up to 6 faces), so
// every XCD takes a contiguous eighth of the blocks and those lines meet in ONE L2.
__device__ __forceinline__ int64_t xcd_block() {
    const int per = (int)gridDim.x >> 3, b = (int)blockIdx.x;
    return b < 8 * per ? (int64_t)(b & 7) * per + (b >> 3) : (int64_t)b;
}

struct FaceGeo { float a[3], b[3], n[3], len; };
__device__ __forceinline__ FaceGeo face_geo(const float (&p)[3][3]) {
    FaceGeo g;
    float c[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { g.a[q] = p[1][q] - p[0][q]; g.b[q] = p[2][q] - p[0][q]; }
    cross3(g.a, g.b, c);
    g.len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) g.n[q] = c[q] / g.len;
    return g;
}

// The two face passes that end in a reduction take PF faces per thread (face j of a thread: block * PF * BLOCK + j * BLOCK + thread):
// the workgroup reduction (3 doubles through DPP + LDS + two barriers) costs as much as a face's arithmetic, and the loads of
// the PF faces are requested together (clamped addresses, no branch in between).
constexpr int PF = 4;

// face normals + the partial sums of |e01|^2, |e02|^2, |e12|^2 of this workgroup's faces (slot = its block of faces)
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_face_normals_norms(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                              float* __restrict__ fn, double* __restrict__ part, int P) {
    __shared__ double smem[3 * (BLOCK / WAVE)];
    const int64_t blk = xcd_block(), f0 = blk * (PF * BLOCK) + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    int id[PF][3];
    float p[PF][3][3];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_face(faces, min(f0 + j * BLOCK, F - 1), verts, id[j], p[j]);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int64_t f = f0 + j * BLOCK;
        if (f < F) {
            const FaceGeo g = face_geo(p[j]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                fn[(size_t)q * F + f] = g.n[q];
                const float e12 = p[j][2][q] - p[j][1][q];
                acc[0] += (double)(g.a[q] * g.a[q]); acc[1] += (double)(g.b[q] * g.b[q]); acc[2] += (double)(e12 * e12);
            }
        }
    }
    block_sum3(acc, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) part[(size_t)i * P + blk] = acc[i];
    }
}

// the corner vectors n_f * theta_i with n_f recomputed
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_vertex_normals_scatter_geo(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                                      const float* __restrict__ norms, const int* __restrict__ cpos,
                                                                      float* __restrict__ corner) {
    const InvNorms nr = inv_norms(norms);
    const int64_t f = xcd_block() * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3];
    load_face(faces, f, verts, id, p);
    const int c0 = cpos[f * 3], c1 = cpos[f * 3 + 1], c2 = cpos[f * 3 + 2];
    const FaceGeo g = face_geo(p);
    const int cp[3] = {c0, c1, c2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Corner c = corner_of(p, i, nr);
#pragma unroll
        for (int q = 0; q < 3; ++q) corner[(size_t)cp[i] * 3 + q] = g.n[q] * c.theta;
    }
}

// first half of the pair's backward: the gradient that reaches the face normals (an output of the pair: other consumers
// may add to it) and the partial sums of dL/dN for the three global norms; nothing is scattered
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_pair_bwd_face(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                         const float* __restrict__ norms, const float* __restrict__ g_raw,
                                                         float* __restrict__ grad_fn, double* __restrict__ part, int P) {
    const InvNorms nr = inv_norms(norms);
    __shared__ double smem[3 * (BLOCK / WAVE)];
    const int64_t blk = xcd_block(), f0 = blk * (PF * BLOCK) + threadIdx.x;
    double gN[3] = {0.0, 0.0, 0.0};
    int id[PF][3];
    float p[PF][3][3], gr[PF][3][3];
#pragma unroll
    for (int j = 0; j < PF; ++j) load_face(faces, min(f0 + j * BLOCK, F - 1), verts, id[j], p[j]);
#pragma unroll
    for (int j = 0; j < PF; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int q = 0; q < 3; ++q) gr[j][i][q] = g_raw[(size_t)id[j][i] * 3 + q];
        }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int64_t f = f0 + j * BLOCK;
        if (f < F) {
            float gf[3] = {0.0f, 0.0f, 0.0f};
            const FaceGeo g = face_geo(p[j]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const Corner c = corner_of(p[j], i, nr);
                float gth = 0.0f;
#pragma unroll
                for (int q = 0; q < 3; ++q) { gf[q] += c.theta * gr[j][i][q]; gth += g.n[q] * gr[j][i][q]; }
                const float gs = gth * dtheta_ds(c.s);
                gN[c.na] += (double)(gs * (-c.s * nr.inv[c.na]));
                gN[c.nb] += (double)(gs * (-c.s * nr.inv[c.nb]));
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) grad_fn[(size_t)q * F + f] = gf[q];
        }
    }
    block_sum3(gN, smem);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) part[(size_t)i * P + blk] = gN[i];
    }
}

// second half: everything that reaches the vertices, one corner vector per corner -- through the corner angles and the three
// global norms (the vertex-normal pass, as k_vertex_normals_bwd2) and through n_f = c / |c| with the TOTAL gradient of the
// face normals g_fn (the face-normal pass, as k_face_normals_bwd; nullptr = no gradient reached them)
template <typename IDX>
__global__ __launch_bounds__(BLOCK) void k_pair_bwd_verts(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t F,
                                                          const float* __restrict__ norms, const float* __restrict__ g_raw,
                                                          const float* __restrict__ gN, const float* __restrict__ g_fn,
                                                          const int* __restrict__ cpos, float* __restrict__ corner) {
    const InvNorms nr = inv_norms(norms);
    const int64_t f = xcd_block() * BLOCK + threadIdx.x;
    if (f >= F) return;
    int id[3];
    float p[3][3], gr[3][3], gv[3][3], gt[3] = {0.0f, 0.0f, 0.0f};
    load_face(faces, f, verts, id, p);
    const int cp[3] = {cpos[f * 3], cpos[f * 3 + 1], cpos[f * 3 + 2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { gr[i][q] = g_raw[(size_t)id[i] * 3 + q]; gv[i][q] = 0.0f; }
    }
    if (g_fn) {
#pragma unroll
        for (int q = 0; q < 3; ++q) gt[q] = g_fn[(size_t)q * F + f];
    }
    const FaceGeo g = face_geo(p);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const Corner c = corner_of(p, i, nr);
        float gth = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; ++q) gth += g.n[q] * gr[i][q];
        const float w = gth * dtheta_ds(c.s) * c.iab;
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float ga = w * c.eb[q], gb = w * c.ea[q];
            gv[i1][q] += ga; gv[i2][q] += gb; gv[i][q] -= ga + gb;
        }
    }
    const float w01 = gN[0] * nr.inv[0], w02 = gN[1] * nr.inv[1], w12 = gN[2] * nr.inv[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float e12 = p[2][q] - p[1][q];
        gv[1][q] += w01 * g.a[q]; gv[0][q] -= w01 * g.a[q];
        gv[2][q] += w02 * g.b[q]; gv[0][q] -= w02 * g.b[q];
        gv[2][q] += w12 * e12; gv[1][q] -= w12 * e12;
    }
    if (g_fn) {
        float ng = 0.0f, gc[3], ga[3], gb[3];
        const float il = 1.0f / g.len;
#pragma unroll
        for (int q = 0; q < 3; ++q) ng += g.n[q] * gt[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) gc[q] = (gt[q] - g.n[q] * ng) * il;
        cross3(g.b, gc, ga);
        cross3(gc, g.a, gb);
#pragma unroll
        for (int q = 0; q < 3; ++q) { gv[1][q] += ga[q]; gv[2][q] += gb[q]; gv[0][q] += -ga[q] - gb[q]; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) corner[(size_t)cp[i] * 3 + q] = gv[i][q];
    }
}

// ---- the pair, vertex-major (round 5) ----------------------------------------------------------------------------------------
// The face-major passes above move one 3-vector per CORNER through memory (72 MB written + 72 MB read per direction at 2M faces) to
// sum it per vertex without atomics. The forward sums can be formed where they are needed: a thread per VERTEX walks its corners in
// rank order (order[] = the inverse of cpos: rank -> corner id 3 f + i), loads that face's three vertices (its neighbours: L2 hits)
// and recomputes the face's contribution -- three times the face arithmetic, no corner buffer, one launch instead of two. The
// statements per corner are those of k_vertex_normals_scatter_geo and the sum runs in the same rank order: the same bits.
// 1M vertices: 32-33 us against 31.9 + 14.4 (profiles/r05_normals_vertex_major.txt). Bound by its occupancy (four dependent round
// trips per thread, 96 VGPRs): 12-byte loads and 3 / 4 / 8 corners in flight change nothing. The BACKWARD built the same way (the
// whole face gradient recomputed per corner, 120 VGPRs) takes 58 us against 35.5 + 14.4 and was not kept.

// a 3-vector at a 4-byte aligned address as ONE 12-byte access (global_load_dwordx3): 24 gather instructions per thread instead of 72
typedef float f3_nrm __attribute__((ext_vector_type(3), aligned(4)));
typedef int i3_nrm __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void ld3(const float* __restrict__ base, size_t row, float (&v)[3]) {
    const f3_nrm t = *reinterpret_cast<const f3_nrm*>(base + row * 3);
    v[0] = t.x; v[1] = t.y; v[2] = t.z;
}
__device__ __forceinline__ void ld_ids(const int32_t* __restrict__ faces, int64_t f, int (&id)[3]) {
    const i3_nrm t = *reinterpret_cast<const i3_nrm*>(faces + f * 3);
    id[0] = t.x; id[1] = t.y; id[2] = t.z;
}
__device__ __forceinline__ void ld_ids(const int64_t* __restrict__ faces, int64_t f, int (&id)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) id[c] = (int)faces[f * 3 + c];
}

__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// theta of corner i (known at run time only) of a face: corner_of's arithmetic on operands picked by selects (no indexed registers)
__device__ __forceinline__ float corner_theta_rt(const float (&p)[3][3], int i, const InvNorms& nr) {
    const float ina = sel3(i, nr.inv[0], nr.inv[2], nr.inv[1]), inb = sel3(i, nr.inv[1], nr.inv[0], nr.inv[2]);   // corner_of: na, nb
    const float iab = ina * inb;
    float d = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float pa = sel3(i, p[0][q], p[1][q], p[2][q]), pb = sel3(i, p[1][q], p[2][q], p[0][q]), pc = sel3(i, p[2][q], p[0][q], p[1][q]);
        const float ea = pb - pa, eb = pc - pa;
        d += ea * eb;
    }
    const float sdot = d * iab;
    return acosf(fminf(fmaxf(sdot, -1.0f), 1.0f));
}

template <typename IDX, int VG>
__global__ __launch_bounds__(BLOCK) void k_vertex_normals_gather_geo(const float* __restrict__ verts, const IDX* __restrict__ faces, int64_t V,
                                                                     const float* __restrict__ norms, const int* __restrict__ vptr,
                                                                     const int* __restrict__ order, float* __restrict__ raw,
                                                                     float* __restrict__ out) {
    const InvNorms nr = inv_norms(norms);
    const int64_t v = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (v >= V) return;
    const int e0 = vptr[v], e1 = vptr[v + 1];
    float x = 0.0f, y = 0.0f, z = 0.0f;
    for (int eb = e0; eb < e1; eb += VG) {
        int cid[VG], id[VG][3];
        float p[VG][3][3];
#pragma unroll
        for (int t = 0; t < VG; ++t) cid[t] = order[min(eb + t, e1 - 1)];
#pragma unroll
        for (int t = 0; t < VG; ++t) ld_ids(faces, (int64_t)(cid[t] / 3), id[t]);
#pragma unroll
        for (int t = 0; t < VG; ++t) {
#pragma unroll
            for (int c = 0; c < 3; ++c) ld3(verts, (size_t)id[t][c], p[t][c]);
        }
#pragma unroll
        for (int t = 0; t < VG; ++t) {
            if (eb + t < e1) {
                const FaceGeo g = face_geo(p[t]);
                const float th = corner_theta_rt(p[t], cid[t] - 3 * (cid[t] / 3), nr);
                x += g.n[0] * th; y += g.n[1] * th; z += g.n[2] * th;
            }
        }
    }
    raw[v * 3] = x; raw[v * 3 + 1] = y; raw[v * 3 + 2] = z;
    const float len = sqrtf(x * x + y * y + z * z);
    out[v * 3] = x / len; out[v * 3 + 1] = y / len; out[v * 3 + 2] = z / len;
}

static int reduce_grid(int64_t F) { return (int)std::min<int64_t>(NRM_MAXG, std::max<int64_t>(1, div_up(F, BLOCK))); }
// slots per row of the partial-sum array: the looping reductions use NRM_MAXG, the per-block ones one per block of faces
static int64_t part_slots(int64_t F) { return std::max<int64_t>(NRM_MAXG, div_up(std::max<int64_t>(F, 1), BLOCK)); }

}  // namespace ls

using namespace ls;

#define LS_IDX(bytes, ...)                                                                     \
    do {                                                                                       \
        if ((bytes) == 8) { typedef int64_t IDX; __VA_ARGS__; } else { typedef int32_t IDX; __VA_ARGS__; } \
    } while (0)

extern "C" int ls_normals_workspace_bytes(int64_t F, int64_t V, size_t* h_bytes) {
    LS_REQUIRE(h_bytes && F >= 0 && V >= 0, LS_E_INVALID, "ls_normals_workspace_bytes: bad argument");
    // reduction partials (one slot per block of faces) | 4 floats | g_raw (V, 3) | one 3-vector per corner (3 F, 3)
    *h_bytes = sizeof(double) * 3 * (size_t)part_slots(F) + sizeof(float) * 4 + sizeof(float) * 3 * (size_t)std::max<int64_t>(V, 1) +
               sizeof(float) * 9 * (size_t)std::max<int64_t>(F, 1);
    return LS_OK;
}

static int check_mesh_args(const void* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const char* who) {
    LS_REQUIRE(verts && (faces || F == 0) && (idx_bytes == 4 || idx_bytes == 8) && F >= 0 && V > 0 && V < INT32_MAX && 3 * F < INT32_MAX,
               LS_E_INVALID, "%s: bad argument (faces must be int32 or int64, V and 3 F < 2^31)", who);
    return LS_OK;
}

struct NormalsWs { double* part; float* gN; float* g_raw; float* corner; };
static NormalsWs carve(void* workspace, int64_t V, int64_t F) {
    NormalsWs w;
    w.part = (double*)workspace;
    w.gN = (float*)(w.part + 3 * part_slots(F));
    w.g_raw = w.gN + 4;
    w.corner = w.g_raw + 3 * (size_t)std::max<int64_t>(V, 1);
    return w;
}

extern "C" int ls_face_normals(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, float* fn, int device,
                               void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_face_normals");
    if (rc) return rc;
    LS_REQUIRE(fn || F == 0, LS_E_INVALID, "ls_face_normals: null output");
    if (F == 0) return LS_OK;
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    LS_IDX(idx_bytes, hipLaunchKernelGGL(k_face_normals<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, fn));
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_face_normals_backward(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                        const int32_t* cpos, const float* g_fn, float* grad_verts, void* workspace, size_t ws_bytes,
                                        int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_face_normals_backward");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(grad_verts && vptr && workspace && ((g_fn && cpos) || F == 0), LS_E_INVALID, "ls_face_normals_backward: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_face_normals_backward: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    if (F > 0)
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_face_normals_bwd<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, g_fn,
                                             cpos, w.corner));
    hipLaunchKernelGGL(k_gather_corners, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, vptr, (const float*)w.corner, V, grad_verts,
                       (float*)nullptr);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_vertex_normals(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                 const int32_t* cpos, const float* fn, float* out, float* raw, float* norms, void* workspace,
                                 size_t ws_bytes, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_vertex_normals");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(out && raw && norms && workspace && vptr && ((fn && cpos) || F == 0), LS_E_INVALID, "ls_vertex_normals: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_vertex_normals: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    if (F > 0) {
        const int G = reduce_grid(F);
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_edge_norm_partials<IDX>, dim3(G), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, w.part));
        hipLaunchKernelGGL(k_finish3, dim3(1), dim3(FIN), 0, st, (const double*)w.part, G, NRM_MAXG, 1, norms);
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_vertex_normals_scatter<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces,
                                             F, fn, (const float*)norms, cpos, w.corner));
    } else {
        LS_HIP(hipMemsetAsync(norms, 0, sizeof(float) * 3, st));
    }
    hipLaunchKernelGGL(k_gather_corners, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, vptr, (const float*)w.corner, V, raw, out);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_vertex_normals_backward(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                          const int32_t* cpos, const float* fn, const float* raw, const float* norms, const float* g_out,
                                          float* grad_verts, float* grad_fn, void* workspace, size_t ws_bytes, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_vertex_normals_backward");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(raw && norms && g_out && grad_verts && workspace && vptr && ((fn && grad_fn && cpos) || F == 0), LS_E_INVALID,
               "ls_vertex_normals_backward: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_vertex_normals_backward: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    if (F > 0) {
        hipLaunchKernelGGL(k_normalize_rows_bwd, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, raw, g_out, V, w.g_raw);
        const int G = reduce_grid(F);
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_vertex_normals_bwd1<IDX>, dim3(G), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, fn, norms,
                                             (const float*)w.g_raw, grad_fn, w.part));
        hipLaunchKernelGGL(k_finish3, dim3(1), dim3(FIN), 0, st, (const double*)w.part, G, NRM_MAXG, 0, w.gN);
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_vertex_normals_bwd2<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, fn,
                                             norms, (const float*)w.g_raw, (const float*)w.gN, cpos, w.corner));
    }
    hipLaunchKernelGGL(k_gather_corners, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, vptr, (const float*)w.corner, V, grad_verts,
                       (float*)nullptr);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- the pair on one mesh (see k_face_normals_norms) ---------------------------------------------------------------------
extern "C" int ls_face_normals_with_norms(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, float* fn, float* norms,
                                          void* workspace, size_t ws_bytes, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_face_normals_with_norms");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE((fn || F == 0) && norms && workspace, LS_E_INVALID, "ls_face_normals_with_norms: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_face_normals_with_norms: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    if (F == 0) { LS_HIP(hipMemsetAsync(norms, 0, sizeof(float) * 3, st)); return LS_OK; }
    const NormalsWs w = carve(workspace, V, F);
    const int G = (int)div_up(F, (int64_t)PF * BLOCK), P = (int)part_slots(F);
    LS_IDX(idx_bytes, hipLaunchKernelGGL(k_face_normals_norms<IDX>, dim3(G), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, fn, w.part, P));
    hipLaunchKernelGGL(k_finish3, dim3(1), dim3(FIN), 0, st, (const double*)w.part, G, P, 1, norms);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_vertex_normals_from_norms(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                            const int32_t* cpos, const float* norms, float* out, float* raw, void* workspace,
                                            size_t ws_bytes, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_vertex_normals_from_norms");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(out && raw && norms && workspace && vptr && (cpos || F == 0), LS_E_INVALID, "ls_vertex_normals_from_norms: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_vertex_normals_from_norms: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    if (F > 0)
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_vertex_normals_scatter_geo<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces,
                                             F, norms, cpos, w.corner));
    hipLaunchKernelGGL(k_gather_corners, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, vptr, (const float*)w.corner, V, raw, out);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_normals_pair_backward_faces(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const float* raw,
                                              const float* norms, const float* g_out, float* g_raw, float* gN, float* grad_fn,
                                              void* workspace, size_t ws_bytes, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_normals_pair_backward_faces");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(raw && norms && g_out && g_raw && gN && workspace && (grad_fn || F == 0), LS_E_INVALID,
               "ls_normals_pair_backward_faces: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_normals_pair_backward_faces: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    hipLaunchKernelGGL(k_normalize_rows_bwd, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, raw, g_out, V, g_raw);
    if (F > 0) {
        const int G = (int)div_up(F, (int64_t)PF * BLOCK), P = (int)part_slots(F);
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_pair_bwd_face<IDX>, dim3(G), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, norms, (const float*)g_raw,
                                             grad_fn, w.part, P));
        hipLaunchKernelGGL(k_finish3, dim3(1), dim3(FIN), 0, st, (const double*)w.part, G, P, 0, gN);
    } else {
        LS_HIP(hipMemsetAsync(gN, 0, sizeof(float) * 3, st));
    }
    LS_HIP(hipGetLastError());
    return LS_OK;
}

extern "C" int ls_normals_pair_backward_verts(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                              const int32_t* cpos, const float* norms, const float* g_raw, const float* gN,
                                              const float* g_fn, float* grad_verts, void* workspace, size_t ws_bytes, int device,
                                              void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_normals_pair_backward_verts");
    if (rc) return rc;
    size_t need = 0;
    ls_normals_workspace_bytes(F, V, &need);
    LS_REQUIRE(norms && g_raw && gN && grad_verts && workspace && vptr && (cpos || F == 0), LS_E_INVALID,
               "ls_normals_pair_backward_verts: null argument");
    LS_REQUIRE(ws_bytes >= need, LS_E_WORKSPACE, "ls_normals_pair_backward_verts: workspace too small (%zu < %zu bytes)", ws_bytes, need);
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    const NormalsWs w = carve(workspace, V, F);
    if (F > 0)
        LS_IDX(idx_bytes, hipLaunchKernelGGL(k_pair_bwd_verts<IDX>, dim3(div_up(F, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces, F, norms,
                                             g_raw, gN, g_fn, cpos, w.corner));
    hipLaunchKernelGGL(k_gather_corners, dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, vptr, (const float*)w.corner, V, grad_verts,
                       (float*)nullptr);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- the pair's forward, vertex-major: no corner buffer, no workspace (order = the inverse permutation of cpos) ---------------------
extern "C" int ls_vertex_normals_gathered(const float* verts, const void* faces, int idx_bytes, int64_t F, int64_t V, const int32_t* vptr,
                                          const int32_t* order, const float* norms, float* out, float* raw, int device, void* stream) {
    int rc = check_mesh_args(verts, faces, idx_bytes, F, V, "ls_vertex_normals_gathered");
    if (rc) return rc;
    LS_REQUIRE(out && raw && norms && vptr && (order || F == 0), LS_E_INVALID, "ls_vertex_normals_gathered: null argument");
    DeviceGuard g(device);
    LS_HIP(g.err);
    hipStream_t st = (hipStream_t)stream;
    LS_IDX(idx_bytes, hipLaunchKernelGGL((k_vertex_normals_gather_geo<IDX, 6>), dim3(div_up(V, BLOCK)), dim3(BLOCK), 0, st, verts, (const IDX*)faces, V,
                                         norms, vptr, order, raw, out));
    LS_HIP(hipGetLastError());
    return LS_OK;
}
