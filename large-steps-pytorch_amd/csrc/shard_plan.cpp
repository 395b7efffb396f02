// shard_plan.cpp -- host-side analysis of the vertex-block shards of the ITERATIVE solvers (largesteps/distributed.py: ShardedPCG,
// ShardedChebyshev): the block of rank `rank` of P, its ghost layers 1..depth (breadth-first on the matrix pattern), the local matrix
// with columns renumbered to [owned | computed ghosts (layers < depth) | read-only ghosts (layer depth)], and the receive / send lists of
// the halo exchange. Host only (no HIP), C++ threads over the ranks whose ghost sets are needed for the send lists.
//
// The reference has no multi-GPU path (largesteps/solvers.py is one process, one device); this serves the north star's "meshes shard by
// vertex blocks across the GPUs of one node with halo exchange". numpy statement of the same plan: tests/shard_plan_statement.py.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/largesteps_hip.h"
#include "nd_plan.h"

namespace ls { void set_error(const char* fmt, ...); }

namespace {

int64_t bound(int64_t V, int P, int q) { return ((int64_t)q * V) / P; }

// ghost layers 1..depth of the block [lo, hi): layer l = vertices at graph distance l, each layer sorted ascending
void layers_of(int64_t V, const int32_t* rowptr, const int32_t* col, int64_t lo, int64_t hi, int depth, std::vector<unsigned char>& seen,
               std::vector<std::vector<int32_t>>& layers) {
    seen.assign((size_t)V, 0);
    for (int64_t v = lo; v < hi; ++v) seen[(size_t)v] = 1;
    layers.assign((size_t)depth, {});
    std::vector<int32_t> frontier;
    for (int l = 0; l < depth; ++l) {
        std::vector<int32_t>& out = layers[(size_t)l];
        auto visit = [&](int32_t v) {
            for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                const int32_t c = col[e];
                if (!seen[(size_t)c]) { seen[(size_t)c] = 1; out.push_back(c); }
            }
        };
        if (l == 0) for (int64_t v = lo; v < hi; ++v) visit((int32_t)v);
        else for (int32_t v : frontier) visit(v);
        std::sort(out.begin(), out.end());
        frontier = out;
    }
}

struct Groups { std::vector<int32_t> inner, outer; };       // computed ghosts (layers < depth, sorted), read-only ghosts (layer depth)

void groups_of(int64_t V, const int32_t* rowptr, const int32_t* col, int P, int q, int depth, Groups& g) {
    std::vector<unsigned char> seen;
    std::vector<std::vector<int32_t>> layers;
    layers_of(V, rowptr, col, bound(V, P, q), bound(V, P, q + 1), depth, seen, layers);
    g.inner.clear();
    for (int l = 0; l + 1 < depth; ++l) g.inner.insert(g.inner.end(), layers[(size_t)l].begin(), layers[(size_t)l].end());
    std::sort(g.inner.begin(), g.inner.end());
    g.outer = layers[(size_t)depth - 1];
}

}  // namespace

struct ls_shard_plan {
    int rank = 0, P = 1, depth = 1;
    int64_t lo = 0, hi = 0, n_inner = 0;
    std::vector<int32_t> rowptr, col, ghosts, recv, send_ptr, send_dst, send_ids;
    std::vector<float> val;
};

extern "C" int ls_shard_layer_sizes(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, int64_t lo, int64_t hi, int depth, int64_t* h_sizes) {
    if (!h_rowptr || !h_col || !h_sizes || V <= 0 || lo < 0 || hi > V || lo > hi || depth < 1) { ls::set_error("ls_shard_layer_sizes: bad argument"); return LS_E_INVALID; }
    if (const char* bad = csr_pattern_problem(V, h_rowptr, h_col)) { ls::set_error("ls_shard_layer_sizes: %s", bad); return LS_E_INVALID; }
    std::vector<unsigned char> seen;
    std::vector<std::vector<int32_t>> layers;
    layers_of(V, h_rowptr, h_col, lo, hi, depth, seen, layers);
    for (int l = 0; l < depth; ++l) h_sizes[l] = (int64_t)layers[(size_t)l].size();
    return LS_OK;
}

extern "C" int ls_shard_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_val, int P, int rank, int depth,
                                    ls_shard_plan** out) {
    if (!out || !h_rowptr || !h_col || !h_val || V < 0) { ls::set_error("ls_shard_plan_create: bad argument"); return LS_E_INVALID; }
    *out = nullptr;
    if (P < 1 || rank < 0 || rank >= P) { ls::set_error("invalid rank %d of %d", rank, P); return LS_E_INVALID; }
    if ((int64_t)P > std::max<int64_t>(V, 1)) { ls::set_error("cannot cut %lld vertices into %d non-empty blocks", (long long)V, P); return LS_E_INVALID; }
    if (depth < 1) { ls::set_error("halo depth must be >= 1"); return LS_E_INVALID; }
    if (const char* bad = csr_pattern_problem(V, h_rowptr, h_col)) { ls::set_error("ls_shard_plan_create: %s", bad); return LS_E_INVALID; }
    ls_shard_plan* s = new ls_shard_plan();
    s->rank = rank; s->P = P; s->depth = depth; s->lo = bound(V, P, rank); s->hi = bound(V, P, rank + 1);
    // every rank's ghost groups (mine for the local matrix and the receive list, the others' for the send lists)
    std::vector<Groups> G((size_t)P);
    {
        const char* e = getenv("LS_PLAN_THREADS");
        int threads = std::max(1, std::min(e ? atoi(e) : 16, (int)std::thread::hardware_concurrency()));
        threads = std::min(threads, P);
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        auto work = [&] { for (int q; (q = next.fetch_add(1)) < P;) groups_of(V, h_rowptr, h_col, P, q, depth, G[(size_t)q]); };
        for (int t = 1; t < threads; ++t) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
    }
    const Groups& me = G[(size_t)rank];
    const int64_t n_own = s->hi - s->lo;
    s->n_inner = (int64_t)me.inner.size();
    s->ghosts = me.inner;
    s->ghosts.insert(s->ghosts.end(), me.outer.begin(), me.outer.end());
    std::vector<int32_t> lut((size_t)std::max<int64_t>(V, 1), -1);
    for (int64_t i = 0; i < n_own; ++i) lut[(size_t)(s->lo + i)] = (int32_t)i;
    for (size_t i = 0; i < s->ghosts.size(); ++i) lut[(size_t)s->ghosts[i]] = (int32_t)(n_own + (int64_t)i);
    const int64_t n_rows = n_own + s->n_inner;
    s->rowptr.assign((size_t)n_rows + 1, 0);
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t g = r < n_own ? s->lo + r : s->ghosts[(size_t)(r - n_own)];
        s->rowptr[(size_t)r + 1] = s->rowptr[(size_t)r] + (h_rowptr[g + 1] - h_rowptr[g]);
    }
    s->col.resize((size_t)s->rowptr[(size_t)n_rows]);
    s->val.resize(s->col.size());
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t g = r < n_own ? s->lo + r : s->ghosts[(size_t)(r - n_own)];
        int64_t w = s->rowptr[(size_t)r];
        for (int32_t e = h_rowptr[g]; e < h_rowptr[g + 1]; ++e, ++w) {
            const int32_t lc = lut[(size_t)h_col[e]];
            if (lc < 0) { delete s; ls::set_error("ls_shard_plan_create: a computed row references a column outside the halo"); return LS_E_INVALID; }
            s->col[(size_t)w] = lc; s->val[(size_t)w] = h_val[e];
        }
    }
    // receive list: per ghost group, one contiguous range per owner (the groups are sorted by global id = by owner first)
    auto owner_of = [&](int32_t g) { int q = (int)(((int64_t)g * P + P - 1) / std::max<int64_t>(V, 1)); q = std::min(q, P - 1); while (q > 0 && bound(V, P, q) > g) --q; while (q + 1 < P && bound(V, P, q + 1) <= g) ++q; return q; };
    for (int grp = 0; grp < 2; ++grp) {
        const std::vector<int32_t>& g = grp == 0 ? me.inner : me.outer;
        const int64_t base = grp == 0 ? 0 : s->n_inner;
        size_t i = 0;
        while (i < g.size()) {
            const int q = owner_of(g[i]);
            size_t j = i;
            while (j < g.size() && owner_of(g[j]) == q) ++j;
            s->recv.push_back(q); s->recv.push_back((int32_t)(base + (int64_t)i)); s->recv.push_back((int32_t)(j - i));
            i = j;
        }
    }
    // send lists: what the others need from me, in THEIR order (group by group, ids ascending)
    s->send_ptr.push_back(0);
    for (int grp = 0; grp < 2; ++grp)
        for (int q = 0; q < P; ++q) {
            if (q == rank) continue;
            const std::vector<int32_t>& g = grp == 0 ? G[(size_t)q].inner : G[(size_t)q].outer;
            const size_t before = s->send_ids.size();
            for (int32_t id : g) if (id >= s->lo && id < s->hi) s->send_ids.push_back((int32_t)(id - s->lo));
            if (s->send_ids.size() > before) { s->send_dst.push_back(q); s->send_ptr.push_back((int32_t)s->send_ids.size()); }
        }
    *out = s;
    return LS_OK;
}

extern "C" int ls_shard_plan_destroy(ls_shard_plan* s) { delete s; return LS_OK; }

extern "C" int ls_shard_plan_info(const ls_shard_plan* s, int64_t* lo, int64_t* hi, int64_t* n_inner, int64_t* n_ghosts, int64_t* n_entries,
                                  int* n_recv, int* n_send, int64_t* n_send_ids) {
    if (!s) { ls::set_error("ls_shard_plan_info: null plan"); return LS_E_INVALID; }
    if (lo) *lo = s->lo;
    if (hi) *hi = s->hi;
    if (n_inner) *n_inner = s->n_inner;
    if (n_ghosts) *n_ghosts = (int64_t)s->ghosts.size();
    if (n_entries) *n_entries = (int64_t)s->col.size();
    if (n_recv) *n_recv = (int)(s->recv.size() / 3);
    if (n_send) *n_send = (int)s->send_dst.size();
    if (n_send_ids) *n_send_ids = (int64_t)s->send_ids.size();
    return LS_OK;
}

extern "C" int ls_shard_plan_arrays(const ls_shard_plan* s, int32_t* rowptr, int32_t* col, float* val, int32_t* ghosts, int32_t* recv3,
                                    int32_t* send_ptr, int32_t* send_dst, int32_t* send_ids) {
    if (!s) { ls::set_error("ls_shard_plan_arrays: null plan"); return LS_E_INVALID; }
    auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(rowptr, s->rowptr); cp(col, s->col); cp(val, s->val); cp(ghosts, s->ghosts); cp(recv3, s->recv); cp(send_ptr, s->send_ptr);
    cp(send_dst, s->send_dst); cp(send_ids, s->send_ids);
    return LS_OK;
}
