// patch_plan.cpp -- host-side analysis for the LDS-resident, s-step Chebyshev kernel (csrc/pcg.hip, k_patch_cheb): the mesh cut into
// 2^m equally sized, spatially compact patches, every patch with its ghost layers 1..s and patch-local uint16 neighbour ids in an
// ELL-by-columns layout. Host only (no HIP), C++ threads over the patches.
//
// This is setup work of the ITERATIVE stand-in for the reference's default solver (largesteps/solvers.py:26-39; the direct solver is
// csrc/nd_*.{cpp,hip}): a Chebyshev step x_{k+1} = x_k + c1 (x_k - x_{k-1}) + c2 D^-1 (b - M x_k) couples mesh neighbours only, so a
// patch plus its ghost layers 1..s can advance s steps out of LDS without talking to anyone (temporal blocking).
//
// Per patch p (arrays concatenated, offsets in `table`, TABLE_COLS ints per patch):
//     own_start, n_own, n_rows, n_local, W, off_gid, off_cols, off_diag | lim[0 .. MAX_DEPTH)
//     local ids   [0, n_own) own | [n_own, n_rows) ghost layers 1..s-1 (recomputed) | [n_rows, n_local) layer s (read only)
//     ghost_gid   new global id of every local vertex >= n_own
//     cols16      (W, n_rows) uint16: local id of the t-th off-diagonal neighbour of row r (padding -> n_local, a zero slot in LDS)
//     diag        (n_rows) fp32
//     lim[m]      rows of layers <= m (lim[0] = n_own, lim[depth-1] = n_rows): with S steps left in a launch only the layers <= S-1
//                 still influence the own vertices, so step j (0-based) of S computes rows < lim[S-1-j]
// The vertices are renumbered patch-major (perm: new -> old), scan-line order inside a patch (stride-1 LDS gathers).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include "../../include/largesteps_hip.h"
#include "nd_plan.h"

namespace ls { void set_error(const char* fmt, ...); }

namespace {

constexpr int MAX_DEPTH = 12, TABLE_COLS = 8 + MAX_DEPTH;

uint64_t spread3(uint64_t x) {
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

int n_threads() {
    const char* e = getenv("LS_PLAN_THREADS");
    const int want = e ? atoi(e) : 32;
    int hw = (int)std::thread::hardware_concurrency();
    const char* lw = getenv("LOCAL_WORLD_SIZE");
    if (!lw) lw = getenv("WORLD_SIZE");
    const int ranks = lw ? atoi(lw) : 1;
    if (hw > 0 && ranks > 1) hw = std::max(1, hw / ranks);
    return std::max(1, std::min(want, hw > 0 ? hw : 1));
}

template <typename F>
void parallel_for(int64_t n, int threads, F&& f) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    if (threads == 1) { for (int64_t i = 0; i < n; ++i) f(i, 0); return; }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] { for (int64_t i; (i = next.fetch_add(1)) < n;) f(i, t); });
    for (auto& x : th) x.join();
}

struct PatchOut {               // one patch of one depth attempt
    std::vector<int32_t> gid;
    std::vector<uint16_t> cols;
    std::vector<float> diag;
    int32_t row[TABLE_COLS];
    bool ok = false;
};

}  // namespace

struct ls_patch_plan {
    int64_t V = 0;
    int depth = 0, n_patches = 0, max_local = 0, max_rows = 0, max_width = 0, patch_size = 0;
    std::vector<int32_t> table, ghost_gid, perm;
    std::vector<uint16_t> cols16;
    std::vector<float> diag;
    double seconds = 0.0;
};

extern "C" int ls_patch_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_diag, const float* h_positions,
                                    int patch_size, int depth, int cap_local, int min_depth, int cap_rows, ls_patch_plan** out) {
    if (!out || !h_rowptr || !h_col || !h_diag || !h_positions || V <= 0 || patch_size < 1 || depth < 1 || depth > MAX_DEPTH || min_depth < 1 ||
        cap_local < 1 || cap_rows < 1) {
        ls::set_error("ls_patch_plan_create: bad argument (1 <= depth <= %d)", MAX_DEPTH);
        return LS_E_INVALID;
    }
    *out = nullptr;
    if (const char* bad = csr_pattern_problem(V, h_rowptr, h_col, h_positions)) { ls::set_error("ls_patch_plan_create: %s", bad); return LS_E_INVALID; }
    const auto t_begin = std::chrono::steady_clock::now();
    // (two V-sized marker arrays per thread below: 32 threads at 4M vertices would hold 1 GB of host memory for markers alone)
    const int threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads(), std::max<int64_t>(1, ((int64_t)256 << 20) / (8 * std::max<int64_t>(V, 1)))));
    // ---- 2^m equal patches by recursive coordinate bisection (median along the longest axis of every box) ---------------------------
    int levels = 0;
    while ((V + (1ll << levels) - 1) / (1ll << levels) > patch_size) ++levels;
    while ((1 << levels) < 256 && V / (1ll << (levels + 1)) >= 192) ++levels;      // a patch per CU as long as patches do not become tiny
    std::vector<int32_t> ids((size_t)V);
    for (int64_t i = 0; i < V; ++i) ids[(size_t)i] = (int32_t)i;
    std::vector<int64_t> start{0, V};
    const float* P = h_positions;
    for (int lv = 0; lv < levels; ++lv) {
        std::vector<int64_t> nxt((size_t)2 * (start.size() - 1) + 1);
        parallel_for((int64_t)start.size() - 1, threads, [&](int64_t bx, int) {
            const int64_t a = start[(size_t)bx], b = start[(size_t)bx + 1], n = b - a, half = n / 2;
            nxt[(size_t)2 * bx] = a; nxt[(size_t)2 * bx + 1] = a + half;
            if (n <= 1) { nxt[(size_t)2 * bx + 1] = b; return; }      // (numpy statement: [idx, empty])
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int64_t i = a; i < b; ++i)
                for (int q = 0; q < 3; ++q) { const float x = P[(size_t)ids[(size_t)i] * 3 + q]; lo[q] = std::min(lo[q], x); hi[q] = std::max(hi[q], x); }
            int axis = 0;
            for (int q = 1; q < 3; ++q) if (hi[q] - lo[q] > hi[axis] - lo[axis]) axis = q;
            std::nth_element(ids.begin() + a, ids.begin() + a + half, ids.begin() + b, [&](int32_t x, int32_t y) {
                const float kx = P[(size_t)x * 3 + axis], ky = P[(size_t)y * 3 + axis];
                return kx < ky || (kx == ky && x < y);          // (key, id): the plan does not depend on the selection's internals
            });
        });
        nxt.back() = V;
        start.swap(nxt);
    }
    // drop empty boxes
    std::vector<int64_t> bs, be;
    for (size_t i = 0; i + 1 < start.size(); ++i) if (start[i + 1] > start[i]) { bs.push_back(start[i]); be.push_back(start[i + 1]); }
    const int n_patches = (int)bs.size();
    // ---- patches along a Morton curve of their centres; scan-line order inside a patch --------------------------------------------------
    float glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < V; ++i)
        for (int q = 0; q < 3; ++q) { glo[q] = std::min(glo[q], P[(size_t)i * 3 + q]); ghi[q] = std::max(ghi[q], P[(size_t)i * 3 + q]); }
    double ext = 1e-30;
    for (int q = 0; q < 3; ++q) ext = std::max(ext, (double)ghi[q] - glo[q]);
    auto morton = [&](const double (&p)[3]) {
        uint64_t k = 0;
        for (int q = 0; q < 3; ++q) {
            double t = (p[q] - glo[q]) / ext * (double)((1 << 21) - 1);
            t = std::min(std::max(t, 0.0), (double)((1 << 21) - 1));
            k |= spread3((uint64_t)t) << q;
        }
        return k;
    };
    std::vector<std::pair<uint64_t, int>> pk((size_t)n_patches);
    parallel_for(n_patches, threads, [&](int64_t p, int) {
        double c[3] = {0, 0, 0};
        for (int64_t i = bs[(size_t)p]; i < be[(size_t)p]; ++i) for (int q = 0; q < 3; ++q) c[q] += P[(size_t)ids[(size_t)i] * 3 + q];
        for (int q = 0; q < 3; ++q) c[q] /= (double)(be[(size_t)p] - bs[(size_t)p]);
        pk[(size_t)p] = {morton(c), (int)p};
    });
    std::stable_sort(pk.begin(), pk.end());
    ls_patch_plan* pl = new ls_patch_plan();
    pl->V = V; pl->patch_size = patch_size; pl->n_patches = n_patches;
    pl->perm.resize((size_t)V);
    std::vector<int64_t> starts((size_t)n_patches + 1, 0);
    for (int p = 0; p < n_patches; ++p) starts[(size_t)p + 1] = starts[(size_t)p] + (be[(size_t)pk[(size_t)p].second] - bs[(size_t)pk[(size_t)p].second]);
    parallel_for(n_patches, threads, [&](int64_t p, int) {
        const int src = pk[(size_t)p].second;
        const int64_t a = bs[(size_t)src], n = be[(size_t)src] - a;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t i = 0; i < n; ++i)
            for (int q = 0; q < 3; ++q) { const float x = P[(size_t)ids[(size_t)(a + i)] * 3 + q]; lo[q] = std::min(lo[q], x); hi[q] = std::max(hi[q], x); }
        int ax[3] = {0, 1, 2};
        std::stable_sort(ax, ax + 3, [&](int x, int y) { return hi[x] - lo[x] > hi[y] - lo[y]; });
        const int a1 = ax[0], a2 = ax[1];
        const double delta = std::max(sqrt(std::max((double)(hi[a1] - lo[a1]) * (hi[a2] - lo[a2]), 1e-300) / (double)std::max<int64_t>(n, 1)), 1e-30);
        struct Key { int64_t row; float x; uint64_t fine; int32_t id; };
        std::vector<Key> keys((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            const int32_t id = ids[(size_t)(a + i)];
            const double pp[3] = {P[(size_t)id * 3], P[(size_t)id * 3 + 1], P[(size_t)id * 3 + 2]};
            keys[(size_t)i] = {(int64_t)floor((pp[a2] - lo[a2]) / delta + 0.5), P[(size_t)id * 3 + a1], morton(pp), id};
        }
        std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) {
            if (x.row != y.row) return x.row < y.row;
            if (x.x != y.x) return x.x < y.x;
            if (x.fine != y.fine) return x.fine < y.fine;
            return x.id < y.id;
        });
        for (int64_t i = 0; i < n; ++i) pl->perm[(size_t)(starts[(size_t)p] + i)] = keys[(size_t)i].id;
    });
    // ---- the matrix in the new numbering ----------------------------------------------------------------------------------------------------
    std::vector<int32_t> inv((size_t)V);
    for (int64_t i = 0; i < V; ++i) inv[(size_t)pl->perm[(size_t)i]] = (int32_t)i;
    std::vector<int64_t> nrp((size_t)V + 1, 0);
    for (int64_t i = 0; i < V; ++i) nrp[(size_t)i + 1] = nrp[(size_t)i] + (h_rowptr[pl->perm[(size_t)i] + 1] - h_rowptr[pl->perm[(size_t)i]]);
    std::vector<int32_t> ncol((size_t)nrp[(size_t)V]);
    parallel_for(V, threads, [&](int64_t i, int) {
        const int32_t o = pl->perm[(size_t)i];
        int64_t w = nrp[(size_t)i];
        for (int32_t e = h_rowptr[o]; e < h_rowptr[o + 1]; ++e) ncol[(size_t)w++] = inv[(size_t)h_col[e]];
    });
    // ---- ghost layers and patch-local matrices: the deepest plan that fits --------------------------------------------------------------------
    std::vector<PatchOut> po((size_t)n_patches);
    std::vector<std::vector<int32_t>> stamp((size_t)threads), lut((size_t)threads);
    for (int t = 0; t < threads; ++t) { stamp[(size_t)t].assign((size_t)V, -1); lut[(size_t)t].assign((size_t)V, -1); }
    bool found = false;
    for (int d = depth; d >= min_depth && !found; --d) {
        std::atomic<bool> fits{true};
        parallel_for(n_patches, threads, [&](int64_t p, int t) {
            PatchOut& o = po[(size_t)p];
            o.ok = false; o.gid.clear(); o.cols.clear(); o.diag.clear();
            if (!fits.load(std::memory_order_relaxed)) return;
            std::vector<int32_t>& seen = stamp[(size_t)t];
            std::vector<int32_t>& loc = lut[(size_t)t];
            const int32_t s0 = (int32_t)starts[(size_t)p], s1 = (int32_t)starts[(size_t)p + 1], n_own = s1 - s0;
            const int32_t tag = (int32_t)((depth - d) * (int64_t)n_patches + p);      // unique per (attempt, patch): the marker arrays are never cleared
            for (int32_t v = s0; v < s1; ++v) seen[(size_t)v] = tag;
            std::vector<int32_t> local;                         // [own | L1 | ... | Ld]
            local.reserve((size_t)n_own * 2);
            for (int32_t v = s0; v < s1; ++v) local.push_back(v);
            std::vector<int32_t> layer_size;
            size_t f0 = 0, f1 = local.size();
            bool too_big = false;
            for (int l = 0; l < d; ++l) {
                const size_t before = local.size();
                for (size_t i = f0; i < f1; ++i) {
                    const int32_t v = local[i];
                    for (int64_t e = nrp[(size_t)v]; e < nrp[(size_t)v + 1]; ++e) {
                        const int32_t c = ncol[(size_t)e];
                        if (seen[(size_t)c] != tag) { seen[(size_t)c] = tag; local.push_back(c); }
                    }
                }
                std::sort(local.begin() + (int64_t)before, local.end());
                layer_size.push_back((int32_t)(local.size() - before));
                f0 = before; f1 = local.size();
                if ((int64_t)local.size() > cap_local) { too_big = true; break; }
            }
            int64_t n_rows = n_own;
            for (size_t l = 0; l + 1 < layer_size.size(); ++l) n_rows += layer_size[l];
            if (too_big || n_rows > cap_rows) { fits.store(false, std::memory_order_relaxed); return; }
            const int32_t n_local = (int32_t)local.size();
            for (int32_t i = 0; i < n_local; ++i) loc[(size_t)local[(size_t)i]] = i;
            int W = 0;
            for (int64_t r = 0; r < n_rows; ++r) {
                const int32_t v = local[(size_t)r];
                int deg = 0;
                for (int64_t e = nrp[(size_t)v]; e < nrp[(size_t)v + 1]; ++e) deg += ncol[(size_t)e] != v;
                W = std::max(W, deg);
            }
            o.cols.assign((size_t)W * (size_t)n_rows, (uint16_t)n_local);
            o.diag.resize((size_t)n_rows);
            for (int64_t r = 0; r < n_rows; ++r) {
                const int32_t v = local[(size_t)r];
                int slot = 0;
                for (int64_t e = nrp[(size_t)v]; e < nrp[(size_t)v + 1]; ++e) {
                    const int32_t c = ncol[(size_t)e];
                    if (c == v) continue;
                    o.cols[(size_t)slot * (size_t)n_rows + (size_t)r] = (uint16_t)loc[(size_t)c];
                    ++slot;
                }
                o.diag[(size_t)r] = h_diag[pl->perm[(size_t)v]];
            }
            for (int32_t i = 0; i < n_local; ++i) loc[(size_t)local[(size_t)i]] = -1;
            o.gid.assign(local.begin() + n_own, local.end());
            memset(o.row, 0, sizeof(o.row));
            o.row[0] = s0; o.row[1] = n_own; o.row[2] = (int32_t)n_rows; o.row[3] = n_local; o.row[4] = W;
            int32_t lim = n_own;
            for (int m = 0; m < MAX_DEPTH; ++m) {
                o.row[8 + m] = m < (int)layer_size.size() ? lim : (int32_t)n_rows;
                if (m + 1 < (int)layer_size.size()) lim += layer_size[(size_t)m];
            }
            o.ok = true;
        });
        if (!fits.load()) continue;
        bool all = true;
        for (const PatchOut& o : po) all = all && o.ok;
        if (!all) continue;
        int64_t off_gid = 0, off_cols = 0, off_diag = 0;
        for (PatchOut& o : po) {
            o.row[5] = (int32_t)off_gid; o.row[6] = (int32_t)off_cols; o.row[7] = (int32_t)off_diag;
            off_gid += (int64_t)o.gid.size(); off_cols += (int64_t)o.cols.size(); off_diag += (int64_t)o.diag.size();
        }
        if (off_cols >= (1ll << 31) || off_gid >= (1ll << 31)) continue;
        pl->depth = d;
        pl->table.resize((size_t)n_patches * TABLE_COLS);
        pl->ghost_gid.resize((size_t)off_gid); pl->cols16.resize((size_t)off_cols); pl->diag.resize((size_t)off_diag);
        for (int p = 0; p < n_patches; ++p) {
            const PatchOut& o = po[(size_t)p];
            memcpy(&pl->table[(size_t)p * TABLE_COLS], o.row, sizeof(o.row));
            std::copy(o.gid.begin(), o.gid.end(), pl->ghost_gid.begin() + o.row[5]);
            std::copy(o.cols.begin(), o.cols.end(), pl->cols16.begin() + o.row[6]);
            std::copy(o.diag.begin(), o.diag.end(), pl->diag.begin() + o.row[7]);
            pl->max_local = std::max(pl->max_local, (int)o.row[3]);
            pl->max_rows = std::max(pl->max_rows, (int)o.row[2]);
            pl->max_width = std::max(pl->max_width, (int)o.row[4]);
        }
        found = true;
    }
    if (!found) { delete pl; return LS_OK; }                    // *out stays NULL: no depth >= min_depth fits (the caller keeps the one-step kernel)
    pl->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    *out = pl;
    return LS_OK;
}

extern "C" int ls_patch_plan_destroy(ls_patch_plan* p) { delete p; return LS_OK; }

extern "C" int ls_patch_plan_info(const ls_patch_plan* p, int* n_patches, int* depth, int* max_local, int* max_rows, int* max_width, int64_t* n_gid,
                                  int64_t* n_cols, int64_t* n_diag, double* seconds) {
    if (!p) { ls::set_error("ls_patch_plan_info: null plan"); return LS_E_INVALID; }
    if (n_patches) *n_patches = p->n_patches;
    if (depth) *depth = p->depth;
    if (max_local) *max_local = p->max_local;
    if (max_rows) *max_rows = p->max_rows;
    if (max_width) *max_width = p->max_width;
    if (n_gid) *n_gid = (int64_t)p->ghost_gid.size();
    if (n_cols) *n_cols = (int64_t)p->cols16.size();
    if (n_diag) *n_diag = (int64_t)p->diag.size();
    if (seconds) *seconds = p->seconds;
    return LS_OK;
}

extern "C" int ls_patch_plan_arrays(const ls_patch_plan* p, int32_t* table, int32_t* ghost_gid, uint16_t* cols16, float* diag, int32_t* perm) {
    if (!p) { ls::set_error("ls_patch_plan_arrays: null plan"); return LS_E_INVALID; }
    if (table) memcpy(table, p->table.data(), p->table.size() * sizeof(int32_t));
    if (ghost_gid) memcpy(ghost_gid, p->ghost_gid.data(), p->ghost_gid.size() * sizeof(int32_t));
    if (cols16) memcpy(cols16, p->cols16.data(), p->cols16.size() * sizeof(uint16_t));
    if (diag) memcpy(diag, p->diag.data(), p->diag.size() * sizeof(float));
    if (perm) memcpy(perm, p->perm.data(), p->perm.size() * sizeof(int32_t));
    return LS_OK;
}
