// nd_plan.cpp -- see nd_plan.h. Integer / geometric work only, parallel over vertices or tree nodes with a few host threads.
#include "nd_plan.h"
#include <pthread.h>
#include <new>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <unistd.h>
#include <sys/resource.h>

namespace ls {
namespace {

int n_threads() {          // read at every build: tests compare the one-thread and the many-thread paths
    const char* e = getenv("LS_PLAN_THREADS");
    const int want = e ? atoi(e) : 32;
    int hw = (int)std::thread::hardware_concurrency();
    // one process per GPU (torchrun): the N ranks of a node analyse their matrices at the same moment on the same host cores --
    // every rank takes its share (LOCAL_WORLD_SIZE is set by torch.distributed.run; WORLD_SIZE as a fallback on one node)
    const char* lw = getenv("LOCAL_WORLD_SIZE");
    if (!lw) lw = getenv("WORLD_SIZE");
    const int ranks = lw ? atoi(lw) : 1;
    if (hw > 0 && ranks > 1) hw = std::max(1, hw / ranks);
    return std::max(1, std::min(want, hw > 0 ? hw : 1));
}

// A pool of host threads: a round of the bisection issues a handful of short parallel passes, and creating 32 threads for each of
// them costs more than the passes themselves. One pool is kept for the life of the process (PoolLease below): creating and joining
// its threads cost every nd_plan_build call 1-2 ms of the ~19 ms a 1M-vertex analysis takes.
class Pool {
public:
    explicit Pool(int threads) : limit_(std::max(1, threads)) {
        for (int t = 1; t < threads; ++t) th_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return limit_; }                       // threads a pass may count on (<= capacity())
    int capacity() const { return (int)th_.size() + 1; }
    void set_limit(int threads) { limit_ = std::max(1, std::min(threads, capacity())); }
    // fn(chunk, begin, end) for `chunks` contiguous pieces of [0, n); the calling thread takes part
    void run(int64_t n, int chunks, const std::function<void(int, int64_t, int64_t)>& fn) {
        if (n <= 0) return;
        chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, n));
        if (chunks == 1 || th_.empty() || limit_ == 1) {
            const int64_t step = (n + chunks - 1) / chunks;
            for (int c = 0; c < chunks; ++c) { const int64_t lo = c * step, hi = std::min(n, lo + step); if (lo < hi) fn(c, lo, hi); }
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; chunks_ = chunks; step_ = (n + chunks - 1) / chunks; next_ = 0; pending_ = chunks; ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            const std::function<void(int, int64_t, int64_t)>* fn;
            int c;
            int64_t lo, hi;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= chunks_) return;
                c = next_++; fn = fn_; lo = c * step_; hi = std::min(n_, lo + step_);
            }
            if (lo < hi) (*fn)(c, lo, hi);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int64_t, int64_t)>* fn_ = nullptr;
    int64_t n_ = 0, step_ = 0;
    int chunks_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    int limit_ = 1;
    bool stop_ = false;
};

thread_local Pool* g_pool = nullptr;      // the pool of the nd_plan_build call running on this thread

// The process-wide pool when nobody else is using it (two threads that analyse two matrices at the same moment -- one per device of
// a multi-device solver -- : the second one gets a pool of its own, as every call did before). The shared pool is never destroyed: its
// threads sleep on a condition variable between calls and end with the process; a forked child (which has none of the parent's
// threads) notices the changed process id and starts its own.
std::mutex g_shared_m;
Pool* g_shared = nullptr;
long g_shared_pid = 0;
bool g_shared_busy = false;
// fork(): the child has none of the pool's threads, and g_shared_m may have been held by a thread that does not exist there. The child
// handler gives it a fresh mutex (placement new over the old object: its state is never looked at again) and forgets the parent's pool
// (leaked on purpose: destroying it would join threads the child does not have). The library must not be dlclose'd while the workers
// sleep in its code -- nothing in the package unloads it.
void shared_pool_after_fork_in_child() {
    new (&g_shared_m) std::mutex();
    g_shared = nullptr; g_shared_busy = false; g_shared_pid = 0;
}
struct ForkHandler { ForkHandler() { (void)pthread_atfork(nullptr, nullptr, shared_pool_after_fork_in_child); } } g_fork_handler;
struct PoolLease {
    Pool* pool = nullptr;
    bool shared = false;
    explicit PoolLease(int threads) {
        {
            std::lock_guard<std::mutex> lk(g_shared_m);
            if (g_shared && g_shared_pid != (long)getpid()) { g_shared = nullptr; g_shared_busy = false; }      // after fork(): the old object is left alone
            if (!g_shared_busy) {
                if (g_shared && g_shared->capacity() < threads) { delete g_shared; g_shared = nullptr; }
                if (!g_shared) { g_shared = new Pool(threads); g_shared_pid = (long)getpid(); }
                g_shared_busy = true; shared = true; pool = g_shared;
            }
        }
        if (shared) pool->set_limit(threads); else pool = new Pool(threads);
    }
    ~PoolLease() {
        if (!shared) { delete pool; return; }
        std::lock_guard<std::mutex> lk(g_shared_m);
        if (pool == g_shared) g_shared_busy = false;
    }
};

// body(begin, end) over [0, n) in contiguous chunks of at least `grain`
void parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& body) {
    const int T = (int)std::min<int64_t>(g_pool ? g_pool->size() : 1, std::max<int64_t>(1, n / std::max<int64_t>(grain, 1)));
    if (T <= 1 || !g_pool) { body(0, n); return; }
    g_pool->run(n, T, [&](int, int64_t lo, int64_t hi) { body(lo, hi); });
}
// fn(chunk, begin, end): the chunk index lets a pass keep per-chunk partial results
void parallel_chunks(int64_t n, int chunks, const std::function<void(int, int64_t, int64_t)>& fn) {
    if (!g_pool) { Pool one(1); one.run(n, chunks, fn); return; }
    g_pool->run(n, chunks, fn);
}

// level-synchronous BFS; dist < 0 = unreached. Returns the last vertex reached.
int bfs(int64_t V, const int32_t* rowptr, const int32_t* col, int start, std::vector<float>& dist) {
    std::vector<int> frontier{start}, next;
    dist[start] = 0.0f;
    int last = start;
    float d = 0.0f;
    while (!frontier.empty()) {
        // "the last vertex reached" = the SMALLEST index of the last level: independent of the order a level is discovered in (the device's
        // sweeps, csrc/nd_bisect.hip nd_embed_device, fill their queues through atomics), and what tests/nd_plan_statement.py states
        last = *std::min_element(frontier.begin(), frontier.end());
        next.clear();
        d += 1.0f;
        for (int u : frontier)
            for (int p = rowptr[u]; p < rowptr[u + 1]; ++p) {
                const int w = col[p];
                if (dist[w] < 0.0f) { dist[w] = d; next.push_back(w); }
            }
        frontier.swap(next);
    }
    return last;
}

// Pseudo-positions for a matrix that comes without vertex positions: graph distances from three mutually far landmarks
// (double-sweep BFS) per connected component, components laid out side by side. Level sets of a graph distance are
// separators as thin as the mesh allows; only the ORDER these coordinates induce matters.
void graph_embedding(int64_t V, const int32_t* rowptr, const int32_t* col, std::vector<double>& pos) {
    pos.assign((size_t)V * 3, 0.0);
    std::vector<char> done((size_t)V, 0);
    std::vector<float> d0((size_t)V), d1((size_t)V), d2((size_t)V), d3((size_t)V);
    double offset = 0.0;
    int64_t remaining = V, seed = 0;
    while (remaining > 0) {
        while (done[seed]) ++seed;
        std::fill(d0.begin(), d0.end(), -1.0f);
        const int p1 = bfs(V, rowptr, col, (int)seed, d0);
        std::fill(d1.begin(), d1.end(), -1.0f);
        const int p2 = bfs(V, rowptr, col, p1, d1);
        std::fill(d2.begin(), d2.end(), -1.0f);
        bfs(V, rowptr, col, p2, d2);
        int p3 = (int)seed;
        float best = -1.0f;
        for (int64_t v = 0; v < V; ++v)
            if (d0[v] >= 0.0f) { const float m = std::min(d1[v], d2[v]); if (m > best) { best = m; p3 = (int)v; } }
        std::fill(d3.begin(), d3.end(), -1.0f);
        bfs(V, rowptr, col, p3, d3);
        int64_t comp = 0;
        float dmax = 0.0f;
        for (int64_t v = 0; v < V; ++v)
            if (d0[v] >= 0.0f) {
                pos[3 * v] = d1[v] + offset; pos[3 * v + 1] = d2[v]; pos[3 * v + 2] = d3[v];
                done[v] = 1; ++comp; dmax = std::max(dmax, d1[v]);
            }
        offset += dmax + 2.0;
        remaining -= comp;
        if (comp <= 2 && remaining > 4096) {        // a swarm of isolated vertices: lay the rest out in index order
            for (int64_t v = 0; v < V; ++v)
                if (!done[v]) { pos[3 * v] = offset; offset += 1.0; done[v] = 1; }
            remaining = 0;
        }
    }
}

}  // namespace

int nd_plan_rounds(int64_t V, int leaf_size, int arity) {
    const int m = arity == 2 ? 1 : arity == 4 ? 2 : 3;
    int D = 0;
    while ((V >> D) > leaf_size) ++D;
    return (D + m - 1) / m * m;
}

std::string nd_plan_build(int64_t V, const int32_t* rowptr, const int32_t* col, const float* pos_in, int leaf_size, int arity,
                          int smooth, NdPlan& P, NdBisectFn bisect, void* bisect_ctx, int ordering, bool defer_push_lists, NdEmbedFn embed) {
    const auto t_start = std::chrono::steady_clock::now();
    const bool timing = getenv("LS_PLAN_TIMING") != nullptr;
    auto faults = [] { struct rusage u; getrusage(RUSAGE_SELF, &u); return (long)u.ru_minflt; };
    const long f_start = timing ? faults() : 0;
    auto lap = [&](const char* what) { if (timing) fprintf(stderr, "[nd_plan] %-28s %.3f s  (%ld page faults so far)\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), faults() - f_start); };
    if (V <= 0 || V >= INT32_MAX) return "nd_plan_build: bad vertex count";
    // (with `bisect` the column indices are still on their way from the device -- see nd_plan_build_device -- and were bounds-checked when
    // the matrix was made: a coalesced torch tensor / ls_assemble_*; the host-only entry points take arbitrary arrays)
    if (!bisect) { if (const char* bad = csr_pattern_problem(V, rowptr, col, pos_in)) return std::string("nd_plan_build: ") + bad; }
    if (arity != 2 && arity != 4 && arity != 8) return "nd_plan_build: arity must be 2, 4 or 8";
    if (leaf_size < 1) return "nd_plan_build: leaf_size must be positive";
    PoolLease lease(n_threads());
    Pool& pool = *lease.pool;
    struct PoolScope { PoolScope(Pool* p) { g_pool = p; } ~PoolScope() { g_pool = nullptr; } } pool_scope(&pool);
    const int T = pool.size();
    const int m = arity == 2 ? 1 : arity == 4 ? 2 : 3;
    int D = 0;
    while ((V >> D) > leaf_size) ++D;
    D = (D + m - 1) / m * m;
    if (D > 40) return "nd_plan_build: tree too deep";
    uvec<int64_t> node((size_t)V);                    // binary heap id of the domain a vertex lives in / became a separator of
    if (!bisect) std::fill(node.begin(), node.end(), (int64_t)1);
    if (bisect) {
        // positions and the D rounds run elsewhere (csrc/nd_bisect.hip: on the device); only the graph embedding -- of a matrix that
        // comes without positions, or as the trial cuts' second set of directions -- is formed here
        std::vector<double> emb;
        const bool minsep = ordering == ND_ORDER_MINSEP;
        std::atomic<bool> empty_row(false);
        if (pos_in || minsep)
            parallel_for(V, 65536, [&](int64_t lo, int64_t hi) { for (int64_t v = lo; v < hi; ++v) if (rowptr[v + 1] <= rowptr[v]) { empty_row = true; break; } });
        const bool all_rows = !empty_row;
        bool emb_on_device = false;
        if (!pos_in || minsep) {
            const std::string how = embed ? embed(bisect_ctx, V) : std::string("host");
            if (how.empty()) emb_on_device = true;
            else if (how == "host") graph_embedding(V, rowptr, col, emb);
            else return how;
        }
        lap("positions");
        const int passes = minsep ? (all_rows ? smooth : 0) : ((pos_in && all_rows) ? smooth : 0);
        const std::string err = bisect(bisect_ctx, V, D, passes, ((!pos_in || minsep) && !emb_on_device) ? emb.data() : nullptr, node.data(), minsep ? ND_ORDER_MINSEP : ND_ORDER_LONGEST);
        if (!err.empty()) return err;
    } else {
        // ---- coordinates: NA candidate axes per vertex. Axes 0-2: the caller's positions (averaged `smooth` times over the matrix
        // neighbours: a rough surface bisects badly otherwise) or, without positions, the graph-distance embedding. ordering ==
        // ND_ORDER_MINSEP with positions: axes 3-5 = the graph-distance embedding as well -- a surface that is folded or rolled up
        // in space (cloth, a scroll, two shells close to each other) has layers that are neighbours in space and not on the
        // surface; a cutting plane then crosses every layer, a level set of a graph distance crosses one.
        const bool minsep = ordering == ND_ORDER_MINSEP;
        const int NA = (minsep && pos_in) ? 6 : 3;
        std::vector<double> pos((size_t)V * NA);
        auto smooth_axes = [&](int a0, int passes) {          // axes a0 .. a0 + 2, sums in CSR order (the device rounds repeat this to the letter)
            std::vector<double> nxt((size_t)V * 3);
            for (int it = 0; it < passes; ++it) {
                parallel_for(V, 4096, [&](int64_t lo, int64_t hi) {
                    for (int64_t v = lo; v < hi; ++v) {
                        double a = 0, b = 0, c = 0;
                        for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) { const size_t w = (size_t)col[p] * NA + a0; a += pos[w]; b += pos[w + 1]; c += pos[w + 2]; }
                        const double inv = 1.0 / (rowptr[v + 1] - rowptr[v]);
                        nxt[3 * v] = a * inv; nxt[3 * v + 1] = b * inv; nxt[3 * v + 2] = c * inv;
                    }
                });
                parallel_for(V, 65536, [&](int64_t lo, int64_t hi) {
                    for (int64_t v = lo; v < hi; ++v) for (int k = 0; k < 3; ++k) pos[(size_t)v * NA + a0 + k] = nxt[3 * v + k];
                });
            }
        };
        bool all_rows = true;
        for (int64_t v = 0; v < V && all_rows; ++v) all_rows = rowptr[v + 1] > rowptr[v];
        if (pos_in) {
            for (int64_t v = 0; v < V; ++v) for (int k = 0; k < 3; ++k) pos[(size_t)v * NA + k] = pos_in[3 * v + k];
            if (smooth > 0 && all_rows) smooth_axes(0, smooth);
        }
        if (!pos_in || NA == 6) {
            std::vector<double> emb;
            graph_embedding(V, rowptr, col, emb);
            const int a0 = pos_in ? 3 : 0;
            for (int64_t v = 0; v < V; ++v) for (int k = 0; k < 3; ++k) pos[(size_t)v * NA + a0 + k] = emb[3 * v + k];
            // integer distances tie by the thousand: a median inside a level set would be cut by vertex id. Averaging makes the
            // level sets smooth curves (only with ND_ORDER_MINSEP: the plain embedding stays what the device rounds are pinned to)
            if (minsep && smooth > 0 && all_rows) smooth_axes(a0, smooth);
        }
        lap("positions");
        // ---- D rounds of bisection ------------------------------------------------------------------------------------------
        std::vector<char> fixed((size_t)V, 0), side((size_t)V, 0), endp((size_t)V, 0);
        // (domain << 2 | side << 1 | fixed) of every vertex in ONE word: the cut detection touches one cache line per neighbour
        std::vector<int64_t> state((size_t)V, 4);
        std::vector<int> live((size_t)V);                 // live vertices grouped by domain
        std::iota(live.begin(), live.end(), 0);
        int64_t n_live = V;
        std::vector<int64_t> seg_start;
        std::vector<int> tmp((size_t)V);
        // `live` holds the live vertices grouped by domain: domain d of the round owns live[seg_start[d] .. seg_start[d + 1]).
        // A median split partitions the segment in place, so the next round's grouping is the two halves minus the separator
        // vertices -- compacted per half with per-domain counts: every pass of a round is parallel over domains or vertices.
        seg_start.assign(2, 0);
        seg_start[1] = V;
        std::vector<int64_t> next_start, keep_cnt;
        std::vector<int> cnt0, cnt1;
        // Rounds with fewer domains than threads run their passes parallel INSIDE a domain (the first round is one domain of V
        // vertices); later rounds run one domain per thread. Both produce the same sets: which vertices land left of the median
        // is decided by the (key, id) order alone, the arrangement inside `live` is irrelevant.
        typedef std::pair<double, int> KV;
        std::vector<KV> kv_a, kv_b;
        constexpr int NB = 2048;                          // buckets of the parallel selection
        std::vector<int> hist;
        std::vector<int> pick, trial;                     // ND_ORDER_MINSEP: split axis per domain, separator size per (domain, axis)
        std::vector<std::vector<char>> cside;             // ... the side of every vertex under each candidate axis
        struct Piece { int64_t d, lo, hi; int n0, e0, n1, e1; int64_t w0, w1; };
        std::vector<Piece> pieces;
        for (int r = 0; r < D; ++r) {
            const int64_t n_dom = (int64_t)1 << r;
            const bool inside = 4 * n_dom <= T && n_live >= 65536;       // (with a thread for every second domain or more, whole domains per thread win)
            auto split_serial = [&](int64_t d) {
                const int64_t a = seg_start[d], e = seg_start[d + 1], cnt = e - a;
                if (cnt <= 0) return;
                double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
                for (int64_t i = a; i < e; ++i)
                    for (int k = 0; k < 3; ++k) { const double x = pos[(size_t)NA * live[i] + k]; mn[k] = std::min(mn[k], x); mx[k] = std::max(mx[k], x); }
                int ax = 0;
                for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
                if (minsep) ax = pick[(size_t)d];
                const int64_t half = cnt / 2;
                // selection on contiguous (key, id) pairs: the comparator must not chase pos[] through the index array
                std::vector<KV> kv((size_t)cnt);
                for (int64_t i = a; i < e; ++i) kv[(size_t)(i - a)] = {pos[(size_t)NA * live[i] + ax], live[i]};
                std::nth_element(kv.begin(), kv.begin() + half, kv.end());
                for (int64_t i = a; i < e; ++i) {
                    const int u = kv[(size_t)(i - a)].second;
                    live[i] = u; side[u] = i - a >= half;
                    state[u] = (node[u] << 2) | ((int64_t)side[u] << 1);
                }
            };
            auto split_parallel = [&](int64_t d) {
                const int64_t a = seg_start[d], e = seg_start[d + 1], cnt = e - a;
                if (cnt < 32768) { split_serial(d); return; }
                const int C = (int)std::min<int64_t>(T, cnt / 8192);
                std::vector<double> part((size_t)C * 12);
                for (int c = 0; c < C; ++c) for (int k = 0; k < 6; ++k) { part[(size_t)c * 12 + k] = 1e300; part[(size_t)c * 12 + 6 + k] = -1e300; }
                parallel_chunks(cnt, C, [&](int c, int64_t lo, int64_t hi) {
                    double mn[6] = {1e300, 1e300, 1e300, 1e300, 1e300, 1e300}, mx[6] = {-1e300, -1e300, -1e300, -1e300, -1e300, -1e300};
                    for (int64_t i = a + lo; i < a + hi; ++i)
                        for (int k = 0; k < NA; ++k) { const double x = pos[(size_t)NA * live[i] + k]; mn[k] = std::min(mn[k], x); mx[k] = std::max(mx[k], x); }
                    for (int k = 0; k < NA; ++k) { part[(size_t)c * 12 + k] = mn[k]; part[(size_t)c * 12 + 6 + k] = mx[k]; }
                });
                double mn[6] = {1e300, 1e300, 1e300, 1e300, 1e300, 1e300}, mx[6] = {-1e300, -1e300, -1e300, -1e300, -1e300, -1e300};
                for (int c = 0; c < C; ++c) for (int k = 0; k < NA; ++k) { mn[k] = std::min(mn[k], part[(size_t)c * 12 + k]); mx[k] = std::max(mx[k], part[(size_t)c * 12 + 6 + k]); }
                int ax = 0;
                for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
                if (minsep) ax = pick[(size_t)d];
                const int64_t half = cnt / 2;
                // bucket = a monotone function of the key: a smaller bucket means a smaller key, so only the median's bucket needs a
                // real selection
                const double lo_key = mn[ax], scale = mx[ax] > mn[ax] ? NB / (mx[ax] - mn[ax]) : 0.0;
                auto bucket = [&](double x) { const int q = (int)((x - lo_key) * scale); return q < 0 ? 0 : q >= NB ? NB - 1 : q; };
                if ((int64_t)kv_a.size() < cnt) { kv_a.resize((size_t)cnt); kv_b.resize((size_t)cnt); }
                hist.assign((size_t)C * NB, 0);
                parallel_chunks(cnt, C, [&](int c, int64_t lo, int64_t hi) {
                    int* h = hist.data() + (size_t)c * NB;
                    for (int64_t i = lo; i < hi; ++i) {
                        const int u = live[a + i];
                        const double x = pos[(size_t)NA * u + ax];
                        kv_a[(size_t)i] = {x, u};
                        ++h[bucket(x)];
                    }
                });
                std::vector<int64_t> total((size_t)NB, 0);
                for (int c = 0; c < C; ++c) for (int q = 0; q < NB; ++q) total[q] += hist[(size_t)c * NB + q];
                int bm = 0;
                int64_t before = 0;
                while (bm + 1 < NB && before + total[bm] <= half) { before += total[bm]; ++bm; }
                const int64_t n_left = before, n_mid = total[bm];
                // write offsets of every chunk's three classes (left of / in / right of the median's bucket)
                std::vector<int64_t> off((size_t)C * 3);
                {
                    int64_t wl = 0, wm = n_left, wr = n_left + n_mid;
                    for (int c = 0; c < C; ++c) {
                        int64_t cl = 0, cr = 0;
                        const int* h = hist.data() + (size_t)c * NB;
                        for (int q = 0; q < bm; ++q) cl += h[q];
                        for (int q = bm + 1; q < NB; ++q) cr += h[q];
                        off[(size_t)c * 3] = wl; off[(size_t)c * 3 + 1] = wm; off[(size_t)c * 3 + 2] = wr;
                        wl += cl; wm += h[bm]; wr += cr;
                    }
                }
                parallel_chunks(cnt, C, [&](int c, int64_t lo, int64_t hi) {
                    int64_t wl = off[(size_t)c * 3], wm = off[(size_t)c * 3 + 1], wr = off[(size_t)c * 3 + 2];
                    for (int64_t i = lo; i < hi; ++i) {
                        const KV x = kv_a[(size_t)i];
                        const int q = bucket(x.first);
                        kv_b[(size_t)(q < bm ? wl++ : q == bm ? wm++ : wr++)] = x;
                    }
                });
                std::nth_element(kv_b.begin() + n_left, kv_b.begin() + half, kv_b.begin() + n_left + n_mid);
                parallel_chunks(cnt, C, [&](int, int64_t lo, int64_t hi) {
                    for (int64_t i = lo; i < hi; ++i) {
                        const int u = kv_b[(size_t)i].second;
                        live[a + i] = u; side[u] = i >= half;
                        state[u] = (node[u] << 2) | ((int64_t)side[u] << 1);
                    }
                });
            };
            const auto tr0 = std::chrono::steady_clock::now();
            if (minsep) {
                // every candidate axis of every domain is TRIED: median split along it, count the end points of the cut edges on
                // either side; the axis whose smaller end-point set is smallest wins (ties: the lower axis -- positions before
                // graph distances). Tasks = (domain, axis); a task's sides live in that axis' own array.
                pick.assign((size_t)n_dom, 0);
                trial.assign((size_t)n_dom * NA, 0);
                if (cside.empty()) cside.assign((size_t)NA, std::vector<char>((size_t)V, 0));
                parallel_for(n_dom * NA, 1, [&](int64_t lo, int64_t hi) {
                    std::vector<KV> kv;
                    for (int64_t t = lo; t < hi; ++t) {
                        const int64_t d = t / NA;
                        const int ax = (int)(t % NA);
                        const int64_t a = seg_start[d], e = seg_start[d + 1], cnt = e - a, half = cnt / 2;
                        if (cnt <= 0) continue;
                        kv.resize((size_t)cnt);
                        double lo_k = 1e300, hi_k = -1e300;
                        for (int64_t i = a; i < e; ++i) {
                            const double x = pos[(size_t)NA * live[i] + ax];
                            kv[(size_t)(i - a)] = {x, live[i]};
                            lo_k = std::min(lo_k, x); hi_k = std::max(hi_k, x);
                        }
                        if (!(hi_k > lo_k) && cnt > 1) { trial[(size_t)t] = INT32_MAX; continue; }      // a constant axis orders by vertex id: never
                        std::nth_element(kv.begin(), kv.begin() + half, kv.end());
                        char* cs = cside[(size_t)ax].data();
                        for (int64_t i = 0; i < cnt; ++i) cs[kv[(size_t)i].second] = i >= half;
                        int e0 = 0, e1 = 0;
                        for (int64_t i = a; i < e; ++i) {
                            const int u = live[i];
                            const int64_t nu = node[u];
                            const char su = cs[u];
                            bool cut = false;
                            for (int p = rowptr[u]; p < rowptr[u + 1] && !cut; ++p) { const int w = col[p]; cut = node[w] == nu && !fixed[w] && cs[w] != su; }
                            if (cut) { if (su) ++e1; else ++e0; }
                        }
                        trial[(size_t)t] = std::min(e0, e1);
                    }
                });
                for (int64_t d = 0; d < n_dom; ++d) {
                    int best = 0;
                    for (int k = 1; k < NA; ++k) if (trial[(size_t)d * NA + k] < trial[(size_t)d * NA + best]) best = k;
                    pick[(size_t)d] = best;
                }
            }
            // median split of every domain along the longest axis of its bounding box (ND_ORDER_MINSEP: along the axis picked above)
            if (inside) { for (int64_t d = 0; d < n_dom; ++d) split_parallel(d); }
            else parallel_for(n_dom, 1, [&](int64_t lo, int64_t hi) { for (int64_t d = lo; d < hi; ++d) split_serial(d); });
            const auto tr1 = std::chrono::steady_clock::now();
            // end points of the cut edges
            parallel_for(n_live, 4096, [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) {
                    const int u = live[i];
                    bool cut = false;
                    const int64_t mine = state[u];          // live vertex of domain node[u] on side side[u]
                    for (int p = rowptr[u]; p < rowptr[u + 1] && !cut; ++p) cut = (state[col[p]] ^ mine) == 2;   // same domain, live, other side
                    endp[u] = cut;
                }
            });
            const auto tr2 = std::chrono::steady_clock::now();
            // per domain: the smaller end-point set is the separator; what stays in either half moves on, in order, to the next
            // round's grouping. The passes run over pieces of domains (a domain is one piece once there are enough domains).
            pieces.clear();
            {
                const int per_dom = inside ? (int)std::max<int64_t>(1, 2 * T / n_dom) : 1;
                for (int64_t d = 0; d < n_dom; ++d) {
                    const int64_t a = seg_start[d], e = seg_start[d + 1];
                    const int np = (int)std::max<int64_t>(1, std::min<int64_t>(per_dom, (e - a) / 4096));
                    const int64_t step = (e - a + np - 1) / np;
                    for (int q = 0; q < np; ++q) {
                        Piece pc{d, a + q * step, std::min(e, a + (q + 1) * step), 0, 0, 0, 0, 0, 0};
                        if (pc.lo < pc.hi || q == 0) pieces.push_back(pc);
                    }
                }
            }
            const int64_t n_pieces = (int64_t)pieces.size();
            parallel_for(n_pieces, 1, [&](int64_t lo, int64_t hi) {
                for (int64_t q = lo; q < hi; ++q) {
                    Piece& pc = pieces[(size_t)q];
                    const int64_t a = seg_start[pc.d], half = (seg_start[pc.d + 1] - a) / 2;
                    int n0 = 0, e0 = 0, n1 = 0, e1 = 0;
                    for (int64_t i = pc.lo; i < pc.hi; ++i) {
                        const bool ep = endp[live[i]];
                        if (i - a >= half) { ++n1; e1 += ep; } else { ++n0; e0 += ep; }
                    }
                    pc.n0 = n0; pc.e0 = e0; pc.n1 = n1; pc.e1 = e1;
                }
            });
            cnt0.assign((size_t)n_dom, 0); cnt1.assign((size_t)n_dom, 0);
            keep_cnt.assign((size_t)2 * n_dom, 0);
            for (const Piece& pc : pieces) { cnt0[(size_t)pc.d] += pc.e0; cnt1[(size_t)pc.d] += pc.e1; }
            for (const Piece& pc : pieces) {
                const bool use1 = cnt1[(size_t)pc.d] < cnt0[(size_t)pc.d];
                keep_cnt[2 * (size_t)pc.d] += pc.n0 - (use1 ? 0 : pc.e0);
                keep_cnt[2 * (size_t)pc.d + 1] += pc.n1 - (use1 ? pc.e1 : 0);
            }
            next_start.assign((size_t)2 * n_dom + 1, 0);
            for (int64_t h = 0; h < 2 * n_dom; ++h) next_start[h + 1] = next_start[h] + keep_cnt[h];
            {
                std::vector<int64_t> w(next_start.begin(), next_start.end() - 1);
                for (Piece& pc : pieces) {
                    const bool use1 = cnt1[(size_t)pc.d] < cnt0[(size_t)pc.d];
                    pc.w0 = w[2 * (size_t)pc.d]; pc.w1 = w[2 * (size_t)pc.d + 1];
                    w[2 * (size_t)pc.d] += pc.n0 - (use1 ? 0 : pc.e0);
                    w[2 * (size_t)pc.d + 1] += pc.n1 - (use1 ? pc.e1 : 0);
                }
            }
            parallel_for(n_pieces, 1, [&](int64_t lo, int64_t hi) {
                for (int64_t q = lo; q < hi; ++q) {
                    const Piece& pc = pieces[(size_t)q];
                    const int64_t a = seg_start[pc.d], half = (seg_start[pc.d + 1] - a) / 2;
                    const bool use1 = cnt1[(size_t)pc.d] < cnt0[(size_t)pc.d];
                    int64_t w0 = pc.w0, w1 = pc.w1;
                    for (int64_t i = pc.lo; i < pc.hi; ++i) {
                        const int u = live[i];
                        const bool s1 = i - a >= half;
                        if (endp[u] && s1 == use1) { fixed[u] = 1; state[u] |= 1; continue; }      // node[u] stays: the domain it separates
                        tmp[(size_t)(s1 ? w1++ : w0++)] = u;
                    }
                }
            });
            n_live = next_start[2 * n_dom];
            parallel_for(n_live, 65536, [&](int64_t lo, int64_t hi) {
                for (int64_t i = lo; i < hi; ++i) { const int u = tmp[i]; live[i] = u; node[u] = 2 * node[u] + side[u]; }
            });
            seg_start.swap(next_start);
            if (timing) {
                const auto tr3 = std::chrono::steady_clock::now();
                auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
                fprintf(stderr, "[nd_plan]   round %2d (%6lld domains%s): split %.2f ms, cut edges %.2f ms, regroup %.2f ms\n", r, (long long)n_dom, inside ? ", inside" : "",
                        ms(tr0, tr1), ms(tr1, tr2), ms(tr2, tr3));
            }
        }
    }
    lap("bisection");
    // ---- merged tree: log2(arity) bisection rounds per level, the leaf domains are the last level -------------------------
    const int levels = D / m + 1;
    P = NdPlan();
    P.V = V; P.levels = levels; P.arity = arity; P.rounds = D;
    P.level_off.resize((size_t)levels + 1);
    {
        int64_t off = 1, cnt = 1;
        for (int l = 0; l <= levels; ++l) { P.level_off[l] = off; off += cnt; cnt *= arity; if (off > ((int64_t)1 << 30)) return "nd_plan_build: tree too large"; }
    }
    const int n_nodes = (int)(P.level_off[levels] - 1);
    P.n_nodes = n_nodes;
    P.parent.assign((size_t)n_nodes + 1, 0); P.level_of.assign((size_t)n_nodes + 1, 0); P.child_ix.assign((size_t)n_nodes + 1, 0);
    for (int l = 0; l < levels; ++l)
        for (int64_t i = P.level_off[l]; i < P.level_off[l + 1]; ++i) {
            P.level_of[i] = l;
            if (l) { P.parent[i] = (int)(P.level_off[l - 1] + (i - P.level_off[l]) / arity); P.child_ix[i] = (int)((i - P.level_off[l]) % arity); }
        }
    uvec<int> node_id((size_t)V);
    parallel_for(V, 65536, [&](int64_t lo, int64_t hi) {
        for (int64_t v = lo; v < hi; ++v) {
            const int64_t h = node[v];
            int lb = 0;
            while ((h >> (lb + 1)) != 0) ++lb;
            const int lvl = lb == D ? D / m : lb / m;
            const int64_t anc = lb == D ? h : h >> (lb - m * (lb / m));
            node_id[v] = (int)(P.level_off[lvl] + (anc - ((int64_t)1 << (m * lvl))));
        }
    });
    // ---- ordering: deepest level first, node by node, original id inside a node ----------------------------------------------
    P.s.assign((size_t)n_nodes + 1, 0); P.b.assign((size_t)n_nodes + 1, 0); P.own_start.assign((size_t)n_nodes + 1, 0);
    {
        // stable counting sort of the vertices by node id, chunk by chunk: counts per (chunk, node), offsets = the node's start +
        // what earlier chunks hold of that node
        const int C = (int)std::max<int64_t>(1, std::min<int64_t>(T, V / 32768));
        std::vector<int> cnt((size_t)C * (n_nodes + 1), 0);
        parallel_chunks(V, C, [&](int c, int64_t lo, int64_t hi) {
            int* h = cnt.data() + (size_t)c * (n_nodes + 1);
            for (int64_t v = lo; v < hi; ++v) ++h[node_id[v]];
        });
        parallel_for(n_nodes + 1, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) { int t = 0; for (int c = 0; c < C; ++c) t += cnt[(size_t)c * (n_nodes + 1) + i]; P.s[i] = t; }
        });
        int64_t off = 0;
        for (int l = levels - 1; l >= 0; --l)
            for (int64_t i = P.level_off[l]; i < P.level_off[l + 1]; ++i) { P.own_start[i] = (int)off; off += P.s[i]; }
        parallel_for(n_nodes + 1, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                int t = P.own_start[i];
                for (int c = 0; c < C; ++c) { int& x = cnt[(size_t)c * (n_nodes + 1) + i]; const int n = x; x = t; t += n; }
            }
        });
        P.perm.resize((size_t)V); P.inv.resize((size_t)V); P.node_of_new.resize((size_t)V);
        parallel_chunks(V, C, [&](int c, int64_t lo, int64_t hi) {
            int* cur = cnt.data() + (size_t)c * (n_nodes + 1);
            for (int64_t v = lo; v < hi; ++v) { const int nw = cur[node_id[v]]++; P.perm[nw] = (int)v; P.inv[v] = nw; P.node_of_new[nw] = node_id[v]; }
        });
    }
    lap("ordering");
    // ---- boundary sets, deepest level first (a node's set needs its children's) ------------------------------------------------
    // Sets live in per-(level, chunk) arenas -- one growing array per thread and level instead of one heap block per node (21845
    // of them at 1M vertices, allocated and freed by 32 threads at once); a level's arenas no longer move once the level is done,
    // which is when the parents read them.
    struct Span { const int* p = nullptr; int n = 0; };
    std::vector<Span> bset((size_t)n_nodes + 1);
    std::vector<std::vector<uvec<int>>> arena((size_t)levels);
    std::atomic<bool> bad(false);
    for (int l = levels - 1; l >= 1; --l) {
        const int64_t first = P.level_off[l], cnt = P.level_off[l + 1] - first;
        const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(T, cnt));
        arena[l].resize((size_t)chunks);
        parallel_chunks(cnt, chunks, [&](int c, int64_t lo, int64_t hi) {
            uvec<int>& A = arena[l][c];
            std::vector<int64_t> at((size_t)(hi - lo) + 1, 0);
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t i = first + k;
                const size_t a0 = A.size();
                const int o = P.own_start[i], oe = o + P.s[i];
                for (int nw = o; nw < oe; ++nw) {
                    const int v = P.perm[nw];
                    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) { const int w = P.inv[col[p]]; if (w >= oe) A.push_back(w); }
                }
                if (l + 1 < levels)
                    for (int ch = 0; ch < arity; ++ch) {
                        const Span& S = bset[P.level_off[l + 1] + (i - first) * arity + ch];
                        for (int e = 0; e < S.n; ++e) if (S.p[e] >= oe) A.push_back(S.p[e]);
                    }
                std::sort(A.begin() + a0, A.end());
                A.erase(std::unique(A.begin() + a0, A.end()), A.end());
                at[(size_t)(k - lo) + 1] = (int64_t)A.size();
            }
            for (int64_t k = lo; k < hi; ++k) {
                Span& S = bset[first + k];
                S.p = A.data() + at[(size_t)(k - lo)]; S.n = (int)(at[(size_t)(k - lo) + 1] - at[(size_t)(k - lo)]);
            }
        });
    }
    lap("boundary sets");
    P.bnd_off.assign((size_t)n_nodes + 2, 0); P.front_off.assign((size_t)n_nodes + 2, 0);
    for (int i = 1; i <= n_nodes; ++i) {
        P.b[i] = bset[i].n;
        P.bnd_off[i + 1] = P.bnd_off[i] + P.b[i];
        P.front_off[i + 1] = P.front_off[i] + P.s[i] + P.b[i];
    }
    P.bnd_off[1] = 0; P.front_off[1] = 0;
    P.n_bnd = P.bnd_off[n_nodes + 1]; P.n_front = P.front_off[n_nodes + 1];
    if (P.n_bnd >= INT32_MAX || P.n_front * arity >= INT32_MAX) return "nd_plan_build: plan exceeds int32 offsets";
    P.bnd.resize((size_t)P.n_bnd); P.ppos.resize((size_t)P.n_bnd);
    // position of every boundary vertex in its parent's front [own | boundary]
    parallel_for(n_nodes, 64, [&](int64_t lo, int64_t hi) {
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t i = k + 1;
            if (i < 2) continue;
            const int par = P.parent[i];
            const int po = P.own_start[par], pe = po + P.s[par];
            const Span PB = bset[par];
            for (int k = 0; k < P.b[i]; ++k) {
                const int w = bset[i].p[k];
                P.bnd[(size_t)P.bnd_off[i] + k] = w;
                int pp;
                if (w < pe) { if (w < po) bad = true; pp = w - po; }
                else {
                    const int* it = std::lower_bound(PB.p, PB.p + PB.n, w);
                    if (it == PB.p + PB.n || *it != w) { bad = true; pp = 0; } else pp = P.s[par] + (int)(it - PB.p);
                }
                P.ppos[(size_t)P.bnd_off[i] + k] = pp;
            }
        }
    });
    if (bad) return "nd_plan_build: separator property violated (the matrix pattern is not symmetric?)";
    if (P.b[1] != 0) return "nd_plan_build: the root has a boundary";
    lap("parent positions");
    if (!defer_push_lists) {
        nd_plan_push_lists(P);
        lap("push lists");
    }
    P.ordering = ordering == ND_ORDER_MINSEP ? ND_ORDER_MINSEP : ND_ORDER_LONGEST;
    nd_plan_quality(P);
    lap("quality");
    P.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return "";
}

// ---- push lists of the down sweep: front position -> boundary entries of the children that are this vertex. Needed by the solve
// only, not by the factorisation: ls_direct_factor builds them while the device factorises (defer_push_lists) ---------------------------
void nd_plan_push_lists(NdPlan& P) {
    if (!P.push_ptr.empty()) return;
    const int levels = P.levels, arity = P.arity;
    PoolLease* own = nullptr;
    if (!g_pool) { own = new PoolLease(n_threads()); g_pool = own->pool; }
    P.push_ptr.assign((size_t)P.n_front + 1, 0);
    P.push_tgt.resize((size_t)P.n_bnd);
    // a parent's front positions receive entries from its own children only: counts and fills run parent by parent
    const int64_t n_inner = P.level_off[levels - 1] - 1;             // nodes 1 .. n_inner have children
    auto children = [&](int64_t par, int64_t& first) { const int l = P.level_of[par]; first = P.level_off[l + 1] + (par - P.level_off[l]) * arity; };
    parallel_for(n_inner, 16, [&](int64_t lo, int64_t hi) {
        for (int64_t par = lo + 1; par <= hi; ++par) {
            int64_t first; children(par, first);
            const int64_t pf = P.front_off[par];
            for (int64_t i = first; i < first + arity; ++i)
                for (int k = 0; k < P.b[i]; ++k) ++P.push_ptr[(size_t)(pf + P.ppos[(size_t)P.bnd_off[i] + k]) + 1];
        }
    });
    for (int64_t f = 0; f < P.n_front; ++f) P.push_ptr[f + 1] += P.push_ptr[f];
    parallel_for(n_inner, 16, [&](int64_t lo, int64_t hi) {
        std::vector<int> cur;
        for (int64_t par = lo + 1; par <= hi; ++par) {
            int64_t first; children(par, first);
            const int64_t pf = P.front_off[par], fl = P.s[par] + P.b[par];
            cur.assign(P.push_ptr.begin() + pf, P.push_ptr.begin() + pf + fl);
            for (int64_t i = first; i < first + arity; ++i)
                for (int k = 0; k < P.b[i]; ++k) P.push_tgt[(size_t)cur[(size_t)P.ppos[(size_t)P.bnd_off[i] + k]]++] = (int)(P.bnd_off[i] + k);
        }
    });
    if (own) { g_pool = nullptr; delete own; }
}

void nd_plan_quality(NdPlan& P) {
    const int n = P.n_nodes;
    double words = 0.0;
    for (int i = 1; i <= n; ++i) words += (double)P.s[i] * P.s[i] + 2.0 * P.s[i] * P.b[i];
    P.words_per_vertex = P.V > 0 ? words / (double)P.V : 0.0;
    std::vector<double> sub((size_t)n + 1, 0.0);
    for (int i = n; i >= 1; --i) { sub[i] += P.s[i]; if (i > 1) sub[P.parent[i]] += sub[i]; }
    const int m = P.arity == 2 ? 1 : P.arity == 4 ? 2 : 3;
    double a = 0.0;
    for (int j = 0; j < m; ++j) a += std::sqrt((double)(1 << j));
    double s2 = 0.0, nv = 0.0;
    const int64_t n_inner = P.levels > 1 ? P.level_off[P.levels - 1] - 1 : 0;
    for (int64_t i = 1; i <= n_inner; ++i) { s2 += (double)P.s[i] * P.s[i]; nv += sub[i]; }
    P.spread = nv > 0.0 ? s2 / (a * a * nv) : 0.0;
}

double nd_plan_suspect() {
    const char* e = getenv("LS_ND_SUSPECT");
    const double v = e ? atof(e) : 1.3;
    return v > 0.0 ? v : 1.3;
}

std::string nd_plan_build_auto(int64_t V, const int32_t* rowptr, const int32_t* col, const float* pos, int leaf_size, int arity,
                               int smooth, NdPlan& out) {
    std::string err = nd_plan_build(V, rowptr, col, pos, leaf_size, arity, smooth, out, nullptr, nullptr, ND_ORDER_LONGEST);
    if (!err.empty() || out.spread <= nd_plan_suspect()) return err;
    NdPlan B;
    err = nd_plan_build(V, rowptr, col, pos, leaf_size, arity, smooth, B, nullptr, nullptr, ND_ORDER_MINSEP);
    if (!err.empty()) return "";                     // the first plan stands
    const double seconds = out.seconds + B.seconds;
    if (B.words_per_vertex < out.words_per_vertex) { B.words_other = out.words_per_vertex; out = std::move(B); }
    else out.words_other = B.words_per_vertex;
    out.seconds = seconds;
    return "";
}

}  // namespace ls

// ---- host-only C entry points (no device needed: the CPU tests check the plan's invariants through these) ---------------------
#include "../../include/largesteps_hip.h"

namespace ls { void set_error(const char* fmt, ...); }

extern "C" int ls_nd_plan_create(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_positions, int leaf_size,
                                 int arity, int smooth, ls_nd_plan** out) {
    if (!out || !h_rowptr || !h_col) { ls::set_error("ls_nd_plan_create: null argument"); return LS_E_INVALID; }
    *out = nullptr;
    ls_nd_plan* h = new ls_nd_plan();
    const std::string err = ls::nd_plan_build(V, h_rowptr, h_col, h_positions, leaf_size, arity, smooth, h->p);
    if (!err.empty()) { delete h; ls::set_error("%s", err.c_str()); return LS_E_INVALID; }
    *out = h;
    return LS_OK;
}

extern "C" int ls_nd_plan_create_ordered(int64_t V, const int32_t* h_rowptr, const int32_t* h_col, const float* h_positions, int leaf_size,
                                         int arity, int smooth, int ordering, ls_nd_plan** out) {
    if (!out || !h_rowptr || !h_col) { ls::set_error("ls_nd_plan_create_ordered: null argument"); return LS_E_INVALID; }
    *out = nullptr;
    if (ordering < ls::ND_ORDER_AUTO || ordering > ls::ND_ORDER_MINSEP) { ls::set_error("ls_nd_plan_create_ordered: ordering must be -1, 0 or 1"); return LS_E_INVALID; }
    ls_nd_plan* h = new ls_nd_plan();
    const std::string err = ordering == ls::ND_ORDER_AUTO
        ? ls::nd_plan_build_auto(V, h_rowptr, h_col, h_positions, leaf_size, arity, smooth, h->p)
        : ls::nd_plan_build(V, h_rowptr, h_col, h_positions, leaf_size, arity, smooth, h->p, nullptr, nullptr, ordering);
    if (!err.empty()) { delete h; ls::set_error("%s", err.c_str()); return LS_E_INVALID; }
    *out = h;
    return LS_OK;
}

extern "C" int ls_nd_plan_quality(const ls_nd_plan* h, int* ordering, double* words_per_vertex, double* spread, double* words_other) {
    if (!h) { ls::set_error("ls_nd_plan_quality: null handle"); return LS_E_INVALID; }
    if (ordering) *ordering = h->p.ordering;
    if (words_per_vertex) *words_per_vertex = h->p.words_per_vertex;
    if (spread) *spread = h->p.spread;
    if (words_other) *words_other = h->p.words_other;
    return LS_OK;
}

extern "C" int ls_nd_plan_destroy(ls_nd_plan* h) { delete h; return LS_OK; }

extern "C" int ls_nd_plan_info(const ls_nd_plan* h, int* levels, int* arity, int* n_nodes, int64_t* n_bnd, int64_t* n_front, double* seconds) {
    if (!h) { ls::set_error("ls_nd_plan_info: null handle"); return LS_E_INVALID; }
    if (levels) *levels = h->p.levels;
    if (arity) *arity = h->p.arity;
    if (n_nodes) *n_nodes = h->p.n_nodes;
    if (n_bnd) *n_bnd = h->p.n_bnd;
    if (n_front) *n_front = h->p.n_front;
    if (seconds) *seconds = h->p.seconds;
    return LS_OK;
}

// copies: perm (V), s / b / own_start / parent (n_nodes + 1 each), bnd / ppos / push_tgt (n_bnd), push_ptr (n_front + 1); any pointer may be NULL
extern "C" int ls_nd_plan_arrays(const ls_nd_plan* h, int32_t* perm, int32_t* s, int32_t* b, int32_t* own_start, int32_t* parent,
                                 int32_t* bnd, int32_t* ppos, int32_t* push_ptr, int32_t* push_tgt) {
    if (!h) { ls::set_error("ls_nd_plan_arrays: null handle"); return LS_E_INVALID; }
    const ls::NdPlan& p = h->p;
    auto cp = [](int32_t* dst, const auto& src) { if (dst) std::copy(src.begin(), src.end(), dst); };
    cp(perm, p.perm); cp(s, p.s); cp(b, p.b); cp(own_start, p.own_start); cp(parent, p.parent);
    cp(bnd, p.bnd); cp(ppos, p.ppos); cp(push_ptr, p.push_ptr); cp(push_tgt, p.push_tgt);
    return LS_OK;
}
