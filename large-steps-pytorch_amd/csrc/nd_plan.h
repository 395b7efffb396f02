// nd_plan.h -- symbolic analysis of the nested-dissection direct solver (host only, no HIP): elimination tree by geometric
// bisection, vertex ordering, fronts, and the static push lists of the two sweeps. Shared by ls_direct_factor
// (csrc/nd_factor.hip) and the host-only C entry points ls_nd_plan_* (tests run them without a GPU).
//
// This is the factorisation-time half of what the reference gets from cholespy / CHOLMOD's analysis phase
// (largesteps/solvers.py:34, `CholeskySolverF(n, ii, jj, x, MatrixType.COO)`).
#pragma once
#include <limits.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace ls {

// std::vector whose resize(n) leaves the new elements uninitialised: the analysis overwrites every entry of its large arrays (4-28 MB
// each at 1M vertices), and zero-filling them first costs the constructor milliseconds. (assign(n, v) and resize(n, v) still fill.)
template <class T>
struct NoInitAlloc {
    typedef T value_type;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
    void deallocate(T* p, size_t) { ::operator delete(p); }
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(static_cast<A0&&>(a0), static_cast<A&&>(a)...); }
    template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T> using uvec = std::vector<T, NoInitAlloc<T>>;

struct NdPlan {
    int64_t V = 0;
    int levels = 0, arity = 4, n_nodes = 0, rounds = 0;      // rounds = bisection rounds D (levels = D / log2(arity) + 1)
    std::vector<int64_t> level_off;                          // node ids are 1-based, level-major: level l = [level_off[l], level_off[l+1])
    std::vector<int> parent, level_of, child_ix;             // per node id (slot 0 unused)
    std::vector<int> s, b, own_start;                        // own block size, boundary size, first new vertex id
    std::vector<int64_t> bnd_off, front_off;                 // prefix sums of b and of s + b
    uvec<int> perm, inv;                                     // perm[new] = old, inv[old] = new
    uvec<int> node_of_new;                                   // node id of every new vertex id
    uvec<int> bnd;                                           // concatenated boundary lists (new vertex ids, ascending per node)
    uvec<int> ppos;                                          // position of every boundary entry in the PARENT's front [own | boundary]
    uvec<int> push_ptr, push_tgt;                            // CSR: front position -> the children's boundary entries that are this vertex
    int64_t n_bnd = 0, n_front = 0;
    double seconds = 0.0;
    // quality of the dissection (nd_plan_quality): factor numbers per vertex, sum over the nodes of s^2 + 2 s b; `spread` = sum of
    // s^2 over the inner nodes / (A x sum of their subtrees' vertex counts), A = (sum_{j < log2 arity} 2^(j/2))^2 -- the squared
    // separator size per vertex of the domain it separates, scale free: ~0.7 on a flat sheet at any size, 1.0-1.2 on closed
    // surfaces, 1.4+ when a cutting plane crosses several layers of a folded surface. ordering = the ND_ORDER_* that built the
    // plan; words_other = words_per_vertex of the plan the automatic choice built and did NOT take (0: it built only one).
    int ordering = 0;
    double words_per_vertex = 0.0, spread = 0.0, words_other = 0.0;
};

// rowptr / col: CSR pattern of a structurally symmetric matrix (int32, original numbering); pos: V x 3 positions (any scale;
// only their spatial order matters), nullptr = derive pseudo-positions from graph distances. Returns "" or an error text.
//
// bisect: when given, the positions' smoothing and the D = nd_plan_rounds(...) rounds of bisection are done by the callee
// (csrc/nd_bisect.hip runs them on the device): it fills node[v] = binary heap id of the domain vertex v ended in (a separator
// vertex: the domain it separates; everything else: its leaf, an id >= 2^D). embedded: nullptr = take the caller's positions and
// average them `smooth` times over the matrix neighbours; otherwise V x 3 doubles formed here from graph distances (smooth = 0).
// With ordering == ND_ORDER_MINSEP the callee gets BOTH where there are positions (its own copy of them and `embedded`), averages both
// `smooth` times and runs the trial cuts over the six directions (without positions: over the averaged embedding's three).
// The host's own rounds (bisect == nullptr) are what the host-only entry points and the CPU tests run, and what the device
// rounds are checked against (tests/test_nested_gpu.py): both must produce the same node[] bit for bit.
// How a domain's cutting direction is chosen (the `ordering` argument below):
//   ND_ORDER_LONGEST  median cut along the longest axis of the domain's bounding box in the embedding it is given (the caller's
//                     positions, or graph distances when there are none): what the device rounds run, a few milliseconds
//   ND_ORDER_MINSEP   every domain TRIES six directions -- the three position axes and three graph distances (level sets of a
//                     graph distance do not care how the surface lies in space) -- and takes the thinnest separator. On the device
//                     too since round 5 (six sorted lists, a side bit and a cut bit per vertex and direction; the graph distances
//                     themselves are breadth-first sweeps on the host); LS_ND_HOST_TRIALS=1 keeps the host rounds (A/B, tests).
//   ND_ORDER_AUTO     LONGEST first; when its separators are thicker than a surface's should be (NdPlan::spread above
//                     nd_plan_suspect(), 1.3), MINSEP as well, and the cheaper plan of the two.
// CHOLMOD's ordering behind the reference's constructor (largesteps/solvers.py:34) is graph based and has no such dependence on
// the embedding; ND_ORDER_AUTO is what ls_direct_factor runs.
enum { ND_ORDER_AUTO = -1, ND_ORDER_LONGEST = 0, ND_ORDER_MINSEP = 1 };
double nd_plan_suspect();                                    // the threshold on NdPlan::spread (environment LS_ND_SUSPECT)
typedef std::string (*NdBisectFn)(void* ctx, int64_t V, int D, int smooth, const double* embedded, int64_t* node, int ordering);
// embed: when given (with bisect), the graph-distance embedding is first asked of the callee too (csrc/nd_bisect.hip: breadth-first sweeps on
// the device, the result stays there and the bisect callee reads it): "" = done, "host" = this graph is one for the host's own sweeps (the
// callee has made sure rowptr / col are complete on the host), anything else = an error text.
typedef std::string (*NdEmbedFn)(void* ctx, int64_t V);
int nd_plan_rounds(int64_t V, int leaf_size, int arity);
std::string nd_plan_build(int64_t V, const int32_t* rowptr, const int32_t* col, const float* pos, int leaf_size, int arity,
                          int smooth, NdPlan& out, NdBisectFn bisect = nullptr, void* bisect_ctx = nullptr, int ordering = ND_ORDER_LONGEST,
                          bool defer_push_lists = false, NdEmbedFn embed = nullptr);
// the push lists of the down sweep (push_ptr / push_tgt): the last stage of nd_plan_build, or -- with defer_push_lists -- called by
// ls_direct_factor while the device factorises (the factorisation does not need them: ~3 ms off the constructor's critical path at 1M)
void nd_plan_push_lists(NdPlan& P);
void nd_plan_quality(NdPlan& P);                             // fills words_per_vertex and spread
// ND_ORDER_AUTO on the host: `pos` are real host positions (or nullptr)
std::string nd_plan_build_auto(int64_t V, const int32_t* rowptr, const int32_t* col, const float* pos, int leaf_size, int arity,
                               int smooth, NdPlan& out);

// The same analysis with the positions' smoothing and the bisection rounds on the device (csrc/nd_bisect.hip). The matrix pattern is
// needed on both sides: d_* on the device, h_rowptr (V + 1) / h_col (nnz) = the CALLER'S BUFFERS for the host copy, filled here (the
// column indices cross the bus while the device rounds run). d_positions may be nullptr (graph embedding, formed on the host).
std::string nd_plan_build_device(const int32_t* d_rowptr, const int32_t* d_col, const float* d_positions, int64_t V, int64_t nnz,
                                 int32_t* h_rowptr, int32_t* h_col, int leaf_size, int arity, int smooth, void* stream, NdPlan& out,
                                 int ordering = ND_ORDER_LONGEST, bool defer_push_lists = false);

}  // namespace ls

// Shared by the host-side planners (nd_plan.cpp, patch_plan.cpp, shard_plan.cpp): a CSR pattern they index arrays with must be checked
// ONCE up front -- a malformed matrix would otherwise write out of bounds in native code where the numpy statements raised IndexError,
// and a NaN coordinate breaks the strict weak order of nth_element / sort. Returns nullptr or what is wrong.
inline const char* csr_pattern_problem(int64_t V, const int32_t* rowptr, const int32_t* col, const float* positions = nullptr) {
    if (V < 0 || V >= INT32_MAX) return "bad vertex count";
    if (rowptr[0] != 0) return "rowptr[0] != 0";
    for (int64_t v = 0; v < V; ++v) if (rowptr[v + 1] < rowptr[v]) return "rowptr is not monotone";
    const int64_t nnz = rowptr[V];
    for (int64_t e = 0; e < nnz; ++e) if (col[e] < 0 || col[e] >= V) return "a column index is out of range";
    if (positions)
        for (int64_t i = 0; i < 3 * V; ++i) if (!(positions[i] - positions[i] == 0.0f)) return "a position is not finite";
    return nullptr;
}

struct ls_nd_plan { ls::NdPlan p; };          // the opaque plan object of the C ABI (ls_nd_plan_create / ls_nd_plan_create_device)
