// experiments/round_study.h -- reduced-precision STORAGE of the factor, emulated: k_convert rounds the fp32 values it writes to what a
// shorter format would hold, every solve kernel stays as it is. A study of the error a format costs before any kernel is written for it
// (round 5, tools/precision_study.py -> docs/measurements.md). Compiled only with -DLS_ND_EXPERIMENTS; the product stores fp32.
//   LS_ND_ROUND_BITS   explicit mantissa bits kept, round to nearest even (23 = fp32; 15 = fp32 without its low byte, "24-bit storage";
//                      10 = fp16's mantissa at fp32's range; 7 = bf16); -16 = IEEE fp16 itself (range and subnormals included)
//   LS_ND_ROUND_MASK   which arrays: 1 leaf triangles, 2 dense tier streams (u4 / d4), 4 upper levels (finv / wf / wb)
//   LS_ND_ROUND_FROM   only nodes of tree level >= this (0 = all): "16 bits below level L, fp32 above"
#pragma once
#include <hip/hip_fp16.h>
namespace ls {
__device__ int g_round_cfg[3] = {23, 0, 0};
__device__ __forceinline__ float exp_round(float v, int layout, int level) {
    const int bits = g_round_cfg[0], cls = layout == 2 ? 1 : layout == 1 ? 2 : 4;
    if (!(g_round_cfg[1] & cls) || level < g_round_cfg[2] || bits >= 23) return v;
    if (bits == -16) return __half2float(__float2half_rn(v));
    const int drop = 23 - bits;
    unsigned u = __float_as_uint(v);
    u += ((1u << (drop - 1)) - 1u) + ((u >> drop) & 1u);
    u &= ~((1u << drop) - 1u);
    return __uint_as_float(u);
}
inline void exp_round_configure() {
    auto env = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    const int cfg[3] = {env("LS_ND_ROUND_BITS", 23), env("LS_ND_ROUND_MASK", 0), env("LS_ND_ROUND_FROM", 0)};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_round_cfg), cfg, sizeof(cfg));
}
}  // namespace ls
