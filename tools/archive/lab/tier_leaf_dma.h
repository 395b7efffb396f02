// ARCHIVED (not compiled). The leaf loop of the tier kernels as a software pipeline around LDS-DMA (global_load_lds), round 4:
// built, bit-identical, measured NOT faster (221-224 us against 219-221 at 1M vertices; docs/measurements.md "Tier leaf loop").
// Last built and measured at commit 9f0a197 as nd_tier.h under -DLS_TIER_DMA=1; it replaced leaf_phase<K, UP, W> there.
#ifndef LS_TIER_DMA
#define LS_TIER_DMA 0
#endif
#ifndef LS_TIER_DMA_AUX
#define LS_TIER_DMA_AUX 0      // cache policy bits of the LDS-DMA requests (2 = nt: streamed once by one CU)
#endif
#if LS_TIER_DMA
// ---- the leaf loop as a software pipeline around LDS-DMA (round 4: BUILT, BIT-IDENTICAL, MEASURED, NOT FASTER -> a build variant) -----
// make EXTRA=-DLS_TIER_DMA=1. The judge's round-3 item: round 3's loop stages every triangle global -> 36 VGPRs -> ds_write -> LDS and
// requests nothing of the NEXT leaf but its record and index lists. global_load_lds (the gfx950 LDS-DMA: 16 bytes per lane, the
// destination is a wave-uniform LDS base + lane x 16, i.e. the packed triangle lands exactly as tri_stage writes it) needs no staging
// registers and no ds_write pass; the freed registers hold the next leaf's small operands, requested a whole leaf ahead; the triangle
// buffer is free as soon as leaf k's mat-vec has read it, so leaf k + 1's DMA is issued THERE and flies under what leaf k still has to
// do (up: y -> LDS, sparse product, update store; down: leaf k + 1's own sparse product, which needs no triangle):
//   record k + 3 -> index lists k + 2 -> small operands k + 1 -> [triangle k + 1 by DMA] -> leaf k
// Ordering of the DMA is by hand (hipcc orders neither a ds_read behind a pending DMA -- it hoisted one above the wait in a probe --
// nor a DMA behind pending ds_reads): RAW: s_waitcnt vmcnt(0) + a wave-level fence at the top of a leaf; WAR: s_waitcnt lgkmcnt(0)
// between the mat-vec's last LDS read and the DMA. No ordinary load is USED between the DMA and that wait (hipcc would wait vmcnt(0)
// for it and drain the DMA): the operands of leaf k + 1 are requested before leaf k's mat-vec and pinned right before the DMA.
// Result (profiles/r04_tier_leaf_variants.txt; 1M / 4M vertices, same box, solutions identical bit for bit): 221-224 / 748-777 us per
// solve with this loop, 219-221 / 723-739 with round 3's loop, default and nt cache policy alike; per-wave clock stamps: a leaf takes
// 5.4 us per wave in either. WHY: the four leaf rounds of a sweep move ~125 MB (105 algorithmic + the partial lines of the 12-byte
// perm -> b gathers) in 22 us = 5.5+ TB/s -- the leaf rounds already stream at the rate the chip sustains, 16 waves per CU are enough
// to cover the round trip; what the tier loses against its 3.4 TB/s average is its start (three dependent round trips and the burst of
// 4096 first triangles: the first leaf is done after 12 us, the next ones every 5.4) and its two dense levels, not the leaf loop.
typedef __attribute__((address_space(3))) void tier_lds_void;
typedef __attribute__((address_space(1))) const void tier_glb_cvoid;
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }        // vmcnt(0)   (expcnt, lgkmcnt: no wait)
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }      // lgkmcnt(0) (vmcnt, expcnt: no wait)
template <typename T> __device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ void tri_dma(const TierArgs& a, const TierItem& n, int lane, float* region) {
    const float4* __restrict__ p = reinterpret_cast<const float4*>(a.tri + n.finv_off);
    const int n4 = LS_ABLATE(a, 16) ? 0 : (n.s * (n.s + 1) / 2 + 3) >> 2;
#pragma unroll
    for (int e = 0; e < TIER_TRI4; ++e) {
        const int i = lane + e * 64;
        if (i < n4) __builtin_amdgcn_global_load_lds((tier_glb_cvoid*)(p + i), (tier_lds_void*)(region + e * 256), 16, 0, LS_TIER_DMA_AUX);
    }
}

template <int K>
struct LeafSmall {         // the operands of a leaf that are not its triangle
    float v[K];            // up: b of own row `lane`; down: y of own row `lane`
    float xv[K];           // down: x_bnd of boundary row `lane`
    SpEnt e[TIER_SPE];
};
template <int K, bool UP>
__device__ __forceinline__ void leaf_small(const TierArgs& a, const TierItem& n, const LeafIdx& ix, const float* __restrict__ b_in, int lane, LeafSmall<K>& d) {
#pragma unroll
    for (int q = 0; q < K; ++q) {
        if (UP) { d.v[q] = lane < n.s ? b_in[(size_t)ix.g * K + q] : 0.0f; d.xv[q] = 0.0f; }
        else {
            d.v[q] = lane < n.s ? a.bprime[(size_t)(n.own_start + lane) * K + q] : 0.0f;
            d.xv[q] = lane < n.b ? a.xb[(size_t)(n.bnd_off + lane) * K + q] : 0.0f;
        }
    }
    const float2* __restrict__ ent = reinterpret_cast<const float2*>(a.sp_ent);       // (always a valid address, see leaf_dat)
#pragma unroll
    for (int t = 0; t < TIER_SPE; ++t) {
        const bool ok = ix.p0 + t < ix.p1;
        const float2 r = ent[ok ? ix.p0 + t : 0];
        d.e[t].val = ok ? r.x : 0.0f;
        d.e[t].idx = ok ? __float_as_int(r.y) : 0;
    }
}
template <int K, bool UP>
__device__ __forceinline__ void pin_small(LeafSmall<K>& d) {
#pragma unroll
    for (int q = 0; q < K; ++q) { pin(d.v[q]); if (!UP) pin(d.xv[q]); }
#pragma unroll
    for (int t = 0; t < TIER_SPE; ++t) { pin(d.e[t].val); pin(d.e[t].idx); }
}

// down sweep, the part of a leaf that needs no triangle:  t = A_sb x_bnd  as [row][4] in LDS behind the triangle
template <int K>
__device__ __forceinline__ void leaf_down_sparse(const TierArgs& a, const TierItem& n, const LeafIdx& ix, const LeafSmall<K>& d, float* region, int tri_floats) {
    const int lane = threadIdx.x & 63, s = n.s, b = n.b;
    float* xbv = region + tri_floats + 64 * 4;
    if (lane < b) {
#pragma unroll
        for (int q = 0; q < K; ++q) xbv[lane * 4 + q] = d.xv[q];
    }
    for (int i = lane + 64; i < b; i += 64) {
#pragma unroll
        for (int q = 0; q < K; ++q) xbv[i * 4 + q] = a.xb[(size_t)(n.bnd_off + i) * K + q];
    }
    wave_lds_sync();
    float t[K];
    if (LS_ABLATE(a, 2)) { for (int q = 0; q < K; ++q) t[q] = d.v[q]; } else
    sparse_row<K>(a, ix, d.e, xbv, t);
    float* tv = region + tri_floats;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < s) { w.x = t[0]; if (K > 1) w.y = t[K > 1 ? 1 : 0]; if (K > 2) w.z = t[K > 2 ? 2 : 0]; if (K > 3) w.w = t[K > 3 ? 3 : 0]; }
    reinterpret_cast<float4*>(tv)[lane] = w;
    wave_lds_sync();
}

// The records of the leaves ahead stay PACKED in one vector register each (r1: leaf k + 1, r2: leaf k + 2) and are unpacked with
// v_readlane where a field is needed (rec_at: the pin keeps the compiler from merging the sites): carrying three unpacked records in
// scalar registers next to the kernel's ~20 pointers spilled ~100 of them to vector lanes -- 42 v_writelane + 40 v_readlane per leaf
// in a loop that is bound by instruction issue.
__device__ __forceinline__ TierItem rec_at(int r) { pin(r); return rec_unpack(r); }

template <int K, bool UP, int W>
__device__ __forceinline__ void leaf_phase(const TierArgs& a, int k0, int k1, const float* __restrict__ b_in, float* __restrict__ x_out,
                                           float* region, int tri_floats) {
    const int lane = threadIdx.x & 63;
    if (k0 >= k1) return;
    constexpr int S = W;
    tier_stamp(a, 24);
    float* yv = region + tri_floats;
    // prologue: three dependent round trips (record -> index lists -> operands), as before; then the first triangle goes out
    TierItem it = rec_unpack(rec_load(a.items, k0, lane));
    LeafIdx ix;
    leaf_idx<UP>(a, it, lane, ix);
    int r1 = k0 + S < k1 ? rec_load(a.items, k0 + S, lane) : 0;
    int r2 = k0 + 2 * S < k1 ? rec_load(a.items, k0 + 2 * S, lane) : 0;
    LeafSmall<K> sd;
    leaf_small<K, UP>(a, it, ix, b_in, lane, sd);
    LeafIdx ix1 = ix;
    if (k0 + S < k1) leaf_idx<UP>(a, rec_at(r1), lane, ix1);
    pin_small<K, UP>(sd);                                  // (landed: nothing ordinary is waited for behind the DMA)
    tri_dma(a, it, lane, region);
    if (!UP) leaf_down_sparse<K>(a, it, ix, sd, region, tri_floats);
    tier_stamp(a, 25);
    for (int k = k0; k < k1; k += S) {
        const bool more = k + S < k1, more2 = k + 2 * S < k1;
        // the wait for leaf k's triangle comes first (everything this wave has in flight is needed now), the requests for the leaves
        // ahead right behind it: operands of leaf k + 1 (its index lists arrived a leaf ago) ...
        wait_vm0();
        wave_lds_sync();
        if (k == k0) tier_stamp(a, 26);
        LeafSmall<K> sd1 = sd;
        if (more) leaf_small<K, UP>(a, rec_at(r1), ix1, b_in, lane, sd1);
        LeafIdx ix2 = ix1;
        int r3 = 0;
        // ... index lists of leaf k + 2 and the record of leaf k + 3: AFTER the mat-vec, where 32 LDS reads are in flight and registers
        // are scarce (they are first used behind the next leaf's wait)
        auto request_ahead = [&]() {
            if (more2) leaf_idx<UP>(a, rec_at(r2), lane, ix2);
            r3 = k + 3 * S < k1 ? rec_load(a.items, k + 3 * S, lane) : 0;
        };
        const int s = it.s, b = it.b;
        if (UP) {
            {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < s) { w.x = sd.v[0]; if (K > 1) w.y = sd.v[K > 1 ? 1 : 0]; if (K > 2) w.z = sd.v[K > 2 ? 2 : 0]; if (K > 3) w.w = sd.v[K > 3 ? 3 : 0]; }
                reinterpret_cast<float4*>(yv)[lane] = w;
            }
            wave_lds_sync();
            float y[K];
            if (LS_ABLATE(a, 1)) { for (int q = 0; q < K; ++q) y[q] = sd.v[q]; } else
            tri_matvec<K>(region, yv, s, lane, y);
            tier_stamp(a, 29);
            if (lane < s) {
#pragma unroll
                for (int q = 0; q < K; ++q) a.bprime[(size_t)(it.own_start + lane) * K + q] = y[q];
            }
            request_ahead();
            // the triangle has been read: the next one may land on it
            wait_lgkm0();
            wave_lds_sync();
            if (more) { pin_small<K, UP>(sd1); tri_dma(a, rec_at(r1), lane, region); }
            if (lane < s) {
#pragma unroll
                for (int q = 0; q < K; ++q) yv[lane * 4 + q] = y[q];
            }
            wave_lds_sync();
            if (it.pfront_off >= 0 && !LS_ABLATE(a, 2)) {
                const bool upc = it.flags & NODE_UPC;
                float* out = upc ? a.xb : a.slots;
                if (lane < b) {
                    float u[K];
                    sparse_row<K>(a, ix, sd.e, yv, u);
                    const size_t dst = upc ? (size_t)(it.bnd_off + lane) * K : ((size_t)(it.pfront_off + ix.pp) * a.arity + it.cix) * K;
#pragma unroll
                    for (int q = 0; q < K; ++q) out[dst + q] = u[q];
                }
                for (int i = lane + 64; i < b; i += 64) {          // leaves with more than 64 boundary rows: no prefetch
                    const int p0 = a.sp_ptr[it.spb_off + i], p1 = a.sp_ptr[it.spb_off + i + 1], pp = a.ppos[it.bnd_off + i];
                    float u[K];
#pragma unroll
                    for (int q = 0; q < K; ++q) u[q] = 0.0f;
                    for (int p = p0; p < p1; ++p) {
                        const SpEnt z = a.sp_ent[p];
#pragma unroll
                        for (int q = 0; q < K; ++q) u[q] = fmaf(z.val, yv[z.idx * 4 + q], u[q]);
                    }
                    const size_t dst = upc ? (size_t)(it.bnd_off + i) * K : ((size_t)(it.pfront_off + pp) * a.arity + it.cix) * K;
#pragma unroll
                    for (int q = 0; q < K; ++q) out[dst + q] = u[q];
                }
            }
            wave_lds_sync();
        } else {
            float z[K];
            if (LS_ABLATE(a, 1)) { for (int q = 0; q < K; ++q) z[q] = 0.0f; } else
            tri_matvec<K>(region, yv, s, lane, z);                 // yv holds t = A_sb x_bnd of this leaf (leaf_down_sparse, a leaf ago)
            tier_stamp(a, 29);
            if (lane < s) {
#pragma unroll
                for (int q = 0; q < K; ++q) x_out[(size_t)ix.g * K + q] = sd.v[q] - z[q];
            }
            request_ahead();
            wait_lgkm0();
            wave_lds_sync();
            if (more) {
                pin_small<K, UP>(sd1);
                const TierItem t1 = rec_at(r1);
                tri_dma(a, t1, lane, region);
                leaf_down_sparse<K>(a, t1, ix1, sd1, region, tri_floats);        // flies under the DMA: needs operands and LDS vectors only
            }
        }
        if (more) it = rec_at(r1);
        ix = ix1; sd = sd1; ix1 = ix2;
        r1 = r2; r2 = r3;
        tier_stamp(a, 16 + min(7, (k - k0) / S));
    }
    // the last DMA of this wave was waited for at the top of its last leaf: nothing of it is pending when the phase's barrier comes
}
