// probe: operand / result layout of v_mfma_f32_4x4x1_16b_f32 with A broadcast (cbsz = 4, abid = block)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ABID>
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}
int main() {
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) { ha[l] = 1000.f * (l / 4) + 100.f * (l % 4 + 1); hb[l] = (float)(l + 1); }
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    // expectation: D[l][i] = A_blk5[i] * B[l] = (5000 + 100 (i + 1)) * (l + 1)
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (hd[l * 4 + i] != (5000.f + 100.f * (i + 1)) * (l + 1)) ++bad;
    printf("mfma 4x4x1 cbsz=4 abid=5: %d mismatches; lane 0: %g %g %g %g; lane 7: %g %g %g %g\n", bad, hd[0], hd[1], hd[2], hd[3], hd[28], hd[29], hd[30], hd[31]);
    return bad != 0;
}
