// nd_plan.h -- symbolic analysis of the nested-dissection direct solver (host only, no HIP): elimination tree by geometric
// bisection, vertex ordering, fronts, and the static push lists of the two sweeps. Shared by ls_direct_factor
// (csrc/nd_factor.hip) and the host-only C entry points ls_nd_plan_* (tests run them without a GPU).
//
// This is the factorisation-time half of what the reference gets from cholespy / CHOLMOD's analysis phase
// (largesteps/solvers.py:34, `CholeskySolverF(n, ii, jj, x, MatrixType.COO)`).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace ls {

struct NdPlan {
    int64_t V = 0;
    int levels = 0, arity = 4, n_nodes = 0, rounds = 0;      // rounds = bisection rounds D (levels = D / log2(arity) + 1)
    std::vector<int64_t> level_off;                          // node ids are 1-based, level-major: level l = [level_off[l], level_off[l+1])
    std::vector<int> parent, level_of, child_ix;             // per node id (slot 0 unused)
    std::vector<int> s, b, own_start;                        // own block size, boundary size, first new vertex id
    std::vector<int64_t> bnd_off, front_off;                 // prefix sums of b and of s + b
    std::vector<int> perm, inv;                              // perm[new] = old, inv[old] = new
    std::vector<int> node_of_new;                            // node id of every new vertex id
    std::vector<int> bnd;                                    // concatenated boundary lists (new vertex ids, ascending per node)
    std::vector<int> ppos;                                   // position of every boundary entry in the PARENT's front [own | boundary]
    std::vector<int> push_ptr, push_tgt;                     // CSR: front position -> the children's boundary entries that are this vertex
    int64_t n_bnd = 0, n_front = 0;
    double seconds = 0.0;
};

// rowptr / col: CSR pattern of a structurally symmetric matrix (int32, original numbering); pos: V x 3 positions (any scale;
// only their spatial order matters), nullptr = derive pseudo-positions from graph distances. Returns "" or an error text.
std::string nd_plan_build(int64_t V, const int32_t* rowptr, const int32_t* col, const float* pos, int leaf_size, int arity,
                          int smooth, NdPlan& out);

}  // namespace ls
