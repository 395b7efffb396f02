// nd_drive.cpp -- C++ driver of the direct solver through the C ABI only (no Python, starts in a second): factorises the
// uniform-Laplacian system M = I + lambda L of an n x n plane (BASELINE.json configs[3] at n = 1000), solves it, checks the solution
// against the known one and the residual on the host, prints an FNV hash of the solution (variant builds must agree bit for bit) and
// times the solve with HIP events (whole solve; per sweep; ND_DRIVE_TABLE=1: every launch). ND_DRIVE_NT=0/1 forces the cache policy.
//   build: hipcc -O2 -std=c++17 tools/nd_drive.cpp -Iinclude -L large-steps-pytorch_amd/lib -llargesteps_hip -Wl,-rpath,'$ORIGIN/../../large-steps-pytorch_amd/lib' -o tools/build/nd_drive
//   run:   tools/build/nd_drive [n = 1000] [solves = 200] [k = 3] [tier levels = -1]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "largesteps_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LS(x) do { int r_ = (x); if (r_ != 0) { printf("largesteps error %d at line %d: %s\n", r_, __LINE__, ls_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000, solves = argc > 2 ? atoi(argv[2]) : 200, k = argc > 3 ? atoi(argv[3]) : 3;
    const int tier = argc > 4 ? atoi(argv[4]) : -1, max_mode = 0;          // tier levels (-1: the library picks)
    const float lambda = 50.0f;
    const int64_t V = (int64_t)n * n;
    // uniform Laplacian of the plane's triangulation (cell (x, y): triangles (i, i+1, i+n+1), (i, i+n+1, i+n)): neighbours E, W, N, S, NE, SW
    std::vector<int32_t> rowptr(V + 1, 0), col;
    std::vector<float> val, pos((size_t)V * 3);
    col.reserve(V * 7); val.reserve(V * 7);
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const int64_t i = (int64_t)y * n + x;
            int64_t nb[6]; int cnt = 0;
            if (y > 0 && x > 0) nb[cnt++] = i - n - 1;
            if (y > 0) nb[cnt++] = i - n;
            if (x > 0) nb[cnt++] = i - 1;
            if (x + 1 < n) nb[cnt++] = i + 1;
            if (y + 1 < n) nb[cnt++] = i + n;
            if (y + 1 < n && x + 1 < n) nb[cnt++] = i + n + 1;
            bool diag_done = false;
            for (int c = 0; c <= cnt; ++c) {
                if (!diag_done && (c == cnt || nb[c] > i)) { col.push_back((int32_t)i); val.push_back(1.0f + lambda * cnt); diag_done = true; }
                if (c < cnt) { col.push_back((int32_t)nb[c]); val.push_back(-lambda); }
            }
            rowptr[i + 1] = (int32_t)col.size();
            pos[i * 3] = (float)x / (n - 1); pos[i * 3 + 1] = (float)y / (n - 1); pos[i * 3 + 2] = 0.1f * sinf(6.2831853f * x / (n - 1));
        }
    const int64_t nnz = (int64_t)col.size();
    std::vector<float> xtrue((size_t)V * k), b((size_t)V * k, 0.0f);
    unsigned seed = 12345u;
    for (auto& v : xtrue) { seed = seed * 1664525u + 1013904223u; v = (float)((seed >> 8) & 0xffff) / 65536.0f; }
    for (int64_t i = 0; i < V; ++i)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
            for (int q = 0; q < k; ++q) b[i * k + q] += val[p] * xtrue[(size_t)col[p] * k + q];
    int32_t *d_rowptr, *d_col; float *d_val, *d_pos, *d_b, *d_x;
    CK(hipMalloc(&d_rowptr, (V + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 4)); CK(hipMalloc(&d_pos, V * 12));
    CK(hipMalloc(&d_b, V * k * 4)); CK(hipMalloc(&d_x, V * k * 4));
    CK(hipMemcpy(d_rowptr, rowptr.data(), (V + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pos, pos.data(), V * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b.data(), V * k * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    ls_direct* h = nullptr;
    int leaf = getenv("ND_DRIVE_LEAF") ? atoi(getenv("ND_DRIVE_LEAF")) : 64, ar = getenv("ND_DRIVE_ARITY") ? atoi(getenv("ND_DRIVE_ARITY")) : 4;
    if (getenv("ND_DRIVE_PICK")) { leaf = ar = 0; LS(ls_direct_pick_tree(V, &leaf, &ar)); }       // the tree the library picks for this size
    LS(ls_direct_factor(d_rowptr, d_col, d_val, V, nnz, d_pos, leaf, ar, tier, 1, 0, 1, 0, st, &h));
    int levels, arity, tl, tw, launches; int64_t wu, wd, nb;
    LS(ls_direct_shape(h, &levels, &arity, &tl, &tw, &wu, &wd, &nb));
    printf("plane %d x %d: V %lld nnz %lld, arity %d, %d levels, tier of %d (%d workgroups), factor %.1f MB up + %.1f MB down\n", n, n, (long long)V, (long long)nnz,
           arity, levels, tl, tw, wu * 4e-6, wd * 4e-6);
    std::vector<float> x0((size_t)V * k), x1((size_t)V * k);
    double span_us = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode <= max_mode; ++mode) {
        if (getenv("ND_DRIVE_NT")) LS(ls_direct_set(h, "nt", atoi(getenv("ND_DRIVE_NT"))));
        LS(ls_direct_info(h, nullptr, &launches, nullptr));
        CK(hipMemsetAsync(d_x, 0xff, V * k * 4, st));
        for (int r = 0; r < 3; ++r) LS(ls_direct_solve(h, d_b, d_x, k, st));
        CK(hipStreamSynchronize(st));
        std::vector<float>& x = mode ? x1 : x0;
        CK(hipMemcpy(x.data(), d_x, V * k * 4, hipMemcpyDeviceToHost));
        double err = 0, res = 0, bn = 0;
        for (size_t i = 0; i < x.size(); ++i) err = std::max(err, (double)fabsf(x[i] - xtrue[i]));
        for (int64_t i = 0; i < V; ++i)
            for (int q = 0; q < k; ++q) {
                double r = b[i * k + q];
                for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) r -= (double)val[p] * x[(size_t)col[p] * k + q];
                res += r * r; bn += (double)b[i * k + q] * b[i * k + q];
            }
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < solves; ++r) LS(ls_direct_solve(h, d_b, d_x, k, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (getenv("ND_DRIVE_GRAPH")) {          // the same solve captured once and replayed: what the launches cost the HOST at small sizes
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            LS(ls_direct_solve(h, d_b, d_x, k, st));
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < solves; ++r) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float gms; CK(hipEventElapsedTime(&gms, e0, e1));
            printf("replayed as a graph: %8.2f us per solve (launched one by one: %8.2f)\n", gms / solves * 1e3, ms / solves * 1e3);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        LS(ls_direct_set(h, "profile", 1));
        double pm[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
        for (int r = 0; r < 10; ++r) { LS(ls_direct_solve(h, d_b, d_x, k, st)); LS(ls_direct_info(h, nullptr, nullptr, pm)); for (int t = 0; t < 3; ++t) acc[t] += pm[t] / 10; }
        LS(ls_direct_set(h, "profile", 0));
        if (mode) span_us = acc[2] * 1e3;
        {   // every launch of a solve: an event in front of each ("profile" 3), mean of 10 solves
            LS(ls_direct_set(h, "profile", 3));
            int nl = 0;
            double ms[64], sum[64] = {0};
            int64_t words[64]; int32_t lo[64], hi[64], sw[64];
            for (int r = 0; r < 10; ++r) {
                LS(ls_direct_solve(h, d_b, d_x, k, st));
                LS(ls_direct_launch_profile(h, 64, &nl, ms, words, lo, hi, sw));
                for (int t = 0; t < nl && t < 64; ++t) sum[t] += ms[t] / 10;
            }
            LS(ls_direct_set(h, "profile", 0));
            if (getenv("ND_DRIVE_TABLE")) {
                double tot = 0;
                for (int t = 0; t < nl && t < 64; ++t) {
                    tot += sum[t] * 1e3;
                    printf("    levels %d-%d %-4s %8.1f MB %7.2f us %6.2f TB/s\n", lo[t], hi[t], sw[t] == 0 ? "up" : sw[t] == 1 ? "down" : "both", words[t] * 4e-6,
                           sum[t] * 1e3, words[t] * 4e-6 / (sum[t] * 1e3));
                }
                printf("    sum of the launches (with an event between each two) %.1f us\n", tot);
            }
        }
        {   // FNV-1a over the solution's bytes: variant builds of the library must agree bit for bit
            unsigned long long hsh = 1469598103934665603ull;
            const unsigned char* pb = (const unsigned char*)x.data();
            for (size_t i = 0; i < x.size() * 4; ++i) { hsh ^= pb[i]; hsh *= 1099511628211ull; }
            printf("solution hash %016llx\n", hsh);
        }
        printf("solve (mode %d): %2d launches  %8.2f us per solve   max |x - x*| %.2e   ||b - M x|| / ||b|| %.2e   events: first part %.1f us, last part %.1f us, middle %.1f us\n",
               mode, launches, ms / solves * 1e3, err, sqrt(res / bn), acc[0] * 1e3, acc[1] * 1e3, acc[2] * 1e3);
    }
    LS(ls_direct_destroy(h));
    return 0;
}
