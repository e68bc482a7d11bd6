// r2l_dw_head16.hip — head weight gradient of the default (fp16) training trio:
//     dWh[o][k] = sum_r Gh[r][o] * PE[r][k]   (k in 1008, padded to 1024),   dbh[o] = sum_r Gh[r][o]
// Gh = gx[0] = dL/d(head pre-activation) [N,256] fp32 (written by the dX chain), PE = the 1008-d positional encoding of the
// ray's 16 sample points, recomputed on the fly from (rays_o, rays_d, t_rand, ztab) exactly as the forward did (never stored:
// 4 KB/ray).  Replaces the head part of loss.backward() (/root/reference/main.py:1403-1404) like r2l_dw_head_kernel
// (r2l_backward.hip), on the fp16 matrix pipe: both operands are rounded to fp16 (PE values lie in [-1, 1] or are point
// coordinates; Gh carries the chain's power-of-two scale so that it sits in fp16's range) and ONE v_mfma_f32_32x32x16_f16
// takes the place of eight v_mfma_f32_32x32x2_f32 — the same argument as r2l_dw16.hip: a reduction over ~10^5 rays of
// independently rounded products.  The fp32 kernel took 0.71 ms of an 8.3 ms step (46 % of the fp32 MFMA peak).
//
// Workgroup (kq, slice); wave w of it owns 32 column PAIRS x all 256 output rows (eight 32-row tiles): 16 accumulator tiles.
// A pair is (sin, cos) of one frequency of one point coordinate — one r2l_sincos evaluation feeds both of the lane's B
// operands — or two identity columns (pairs 480 .. 503), or padding (504 .. 511): pair P = 128 kq + 32 w + m, tile 0 holds
// the pair's first column, tile 1 its second.  One k-step = 16 rays: lane (m, kg) supplies, for rays 8 kg .. 8 kg + 7 of the
// step, Gh[ray][32 e + m] (A operands, e = 0..7: coalesced 128-byte rows) and its pair's two encoding values (B operands).
// The kernel is VALU-bound (the encoding), not matrix-bound.  Rows past the end of a slice are out-of-range buffer loads
// (zero fill): no tails, no predicates.  Per-slice partials go to the slab and are added in slice order by
// r2l_head_reduce_kernel, like the fp32 kernel's.
#include "r2l_f2.h"
#include "r2l_dw.h"

struct Head16G {  // gradient rows of one k-step as loaded: [tile e][ray i of this lane's k half]
    float g[8][8];
};
struct Head16Rays {  // ray data of one k-step for this lane's column pair: coordinate A (and B: identity pairs only)
    float oa[8], da[8], ua[8], ob[8], db[8], ub[8];
};
// the column pair of a lane: encoding columns ka / kb (-1: padding), the point coordinate(s) they read, the frequency
struct Head16Pair {
    int ka, kb;        // columns of head.0.weight (PositionalEmbedder order: 21 per coordinate: 10 sin, 10 cos, x)
    int smp_a, ax_a;   // coordinate of ka (and of kb for a trig pair)
    int smp_b, ax_b;   // coordinate of kb for an identity pair
    float scale;       // 2^f (trig pairs)
    bool trig;
};
__device__ __forceinline__ Head16Pair head16_pair(int P) {
    Head16Pair c;
    if (P < 480) {
        const int co = P / 10, f = P - 10 * co;
        c.ka = 21 * co + f; c.kb = 21 * co + 10 + f;
        c.smp_a = c.smp_b = co / 3; c.ax_a = c.ax_b = co % 3;
        c.scale = (float)(1 << f); c.trig = true;
    } else {
        const int i = P - 480, ca = 2 * i, cb = 2 * i + 1;  // 48 identity columns = 24 pairs; 504 ..: padding
        const bool live = i < 24;
        c.ka = live ? 21 * ca + 20 : -1; c.kb = live ? 21 * cb + 20 : -1;
        c.smp_a = live ? ca / 3 : 0; c.ax_a = live ? ca % 3 : 0;
        c.smp_b = live ? cb / 3 : 0; c.ax_b = live ? cb % 3 : 0;
        c.scale = 1.f; c.trig = false;
    }
    return c;
}

typedef __amdgpu_buffer_rsrc_t h16_rsrc_t;
__device__ __forceinline__ float h16_load(h16_rsrc_t rs, unsigned voff, unsigned imm) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + imm, 0, 0));
}
// raw buffer (no stride, 32-bit offsets) over `bytes` bytes at base: loads at offsets >= bytes return 0
__device__ __forceinline__ h16_rsrc_t h16_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// EXACT (r2l_config.dw_mode = R2L_DW_EXACT): both operands as fp16 hi + mid, mid*hi + hi*mid + hi*hi per tile (48 MFMAs per
// k-step instead of 16; the kernel stays VALU-bound).
template <bool JITTER, bool EXACT>
__global__ __launch_bounds__(256, 1) void r2l_dw_head16_kernel(const R2LDwHeadArgs a) {
    if (a.run_unless != nullptr && __builtin_nontemporal_load(a.run_unless) != 0u) return;  // the fp32 kernel behind does it
    const float gscale = a.scale_dev != nullptr ? a.scale_dev[0] : a.gscale;
    const float unscale = a.scale_dev != nullptr ? a.scale_dev[1] : a.unscale;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = lane & 31, kg = lane >> 5;
    const int kq = blockIdx.x & 3;
    const int64_t slice = blockIdx.x >> 2;
    const int64_t r0 = slice * a.rays_per_wg;
    int64_t r1 = r0 + a.rays_per_wg;
    if (r1 > a.N) r1 = a.N;
    if (r0 >= r1) return;
    const int nrays = (int)(r1 - r0);
    const Head16Pair pc = head16_pair(kq * 128 + wave * 32 + m);
    const bool trig_wave = kq * 128 + wave * 32 + 31 < 480;  // wave-uniform: every lane of this wave holds a (sin, cos) pair
    const float zla = a.ztab[pc.smp_a], zsa = JITTER ? a.ztab[16 + pc.smp_a] : 0.f;
    const float zlb = a.ztab[pc.smp_b], zsb = JITTER ? a.ztab[16 + pc.smp_b] : 0.f;

    f32x16 acc[8][2];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[e][ei][c] = 0.f;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // dbh partials (kq == 0, wave 0): row 32 e + m, this lane's rays

    // raw buffers over the slice's rows: loads past its last ray return zero (the range check sees voffset + immediate, which
    // is why the k-step advance is added to the VGPR offsets, not passed as a scalar offset)
    const h16_rsrc_t grs = h16_rsrc(a.gh + r0 * R2L_W, (unsigned)nrays * (R2L_W * 4));
    const h16_rsrc_t ors = h16_rsrc(a.rays_o + r0 * 3, (unsigned)nrays * 12u);
    const h16_rsrc_t drs = h16_rsrc(a.rays_d + r0 * 3, (unsigned)nrays * 12u);
    const h16_rsrc_t urs = h16_rsrc(JITTER ? a.t_rand + r0 * 16 : a.rays_o, JITTER ? (unsigned)nrays * 64u : 0u);
    unsigned vg = (unsigned)((8 * kg * R2L_W + m) * 4);          // Gh[8kg + i][32 e + m]: + i*1024 + e*128
    unsigned vp0 = (unsigned)((8 * kg * 3 + pc.ax_a) * 4), vp1 = (unsigned)((8 * kg * 3 + pc.ax_b) * 4);     // + i*12
    unsigned vu0 = (unsigned)((8 * kg * 16 + pc.smp_a) * 4), vu1 = (unsigned)((8 * kg * 16 + pc.smp_b) * 4);  // + i*64
    const int nsteps = (nrays + 15) / 16;

    // the k-step the offsets point at, then advance them by 16 rays
    auto ld_g = [&](Head16G& v) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v.g[e][i] = (i < 4) ? h16_load(grs, vg, (unsigned)(i * 1024 + e * 128))
                                    : h16_load(grs, vg + 4096u, (unsigned)((i - 4) * 1024 + e * 128));
        vg += 16u * R2L_W * 4u;
    };
    auto ld_rays = [&](Head16Rays& v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v.oa[i] = h16_load(ors, vp0, (unsigned)i * 12u);
            v.da[i] = h16_load(drs, vp0, (unsigned)i * 12u);
            v.ua[i] = JITTER ? h16_load(urs, vu0, (unsigned)i * 64u) : 0.f;
            if (!trig_wave) {  // identity pairs read a second coordinate (a (sin, cos) pair shares one)
                v.ob[i] = h16_load(ors, vp1, (unsigned)i * 12u);
                v.db[i] = h16_load(drs, vp1, (unsigned)i * 12u);
                v.ub[i] = JITTER ? h16_load(urs, vu1, (unsigned)i * 64u) : 0.f;
            } else {
                v.ob[i] = v.db[i] = v.ub[i] = 0.f;
            }
        }
        vp0 += 16u * 12u; vp1 += 16u * 12u;
        vu0 += 16u * 64u; vu1 += 16u * 64u;
    };
    auto pk = [](float x0, float x1) { return F2Side<false, F3None>::pk(x0, x1); };
    // One set of raw registers each: a k-step first turns its gradient rows into fp16 A operands and re-issues the loads of the
    // next step into the same registers (they land under the encoding work), then evaluates its encoding columns and
    // re-issues the ray loads (they land under the MFMAs and the next step's conversions).
    Head16G rg;
    Head16Rays rr;
    ld_g(rg);
    ld_rays(rr);
    for (int s = 0; s < nsteps; ++s) {
        f16x8 ga[8], gm[EXACT ? 8 : 1];
        const float sc = gscale;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u32x4 uh;
#pragma unroll
            for (int d = 0; d < 4; ++d) uh[d] = pk(rg.g[e][2 * d] * sc, rg.g[e][2 * d + 1] * sc);
            ga[e] = __builtin_bit_cast(f16x8, uh);
            if (EXACT) {
                u32x4 um;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    um[d] = pk(f2_res_lo(uh[d], rg.g[e][2 * d] * sc), f2_res_hi(uh[d], rg.g[e][2 * d + 1] * sc));
                gm[e] = __builtin_bit_cast(f16x8, um);
            }
            if (kq == 0 && wave == 0)
                bsum[e] += ((rg.g[e][0] + rg.g[e][1]) + (rg.g[e][2] + rg.g[e][3])) + ((rg.g[e][4] + rg.g[e][5]) + (rg.g[e][6] + rg.g[e][7]));
        }
        __builtin_amdgcn_sched_barrier(0);
        ld_g(rg);  // (one step past the end: every load out of range, zeros)
        __builtin_amdgcn_sched_barrier(0);
        // B operands: the lane's column pair for its 8 rays (point = fl(o + fl(d*z)), as the forward)
        float p0[8], p1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xa = rr.oa[i] + rr.da[i] * (JITTER ? zla + zsa * rr.ua[i] : zla);
            if (trig_wave) {
                r2l_sincos(xa * pc.scale, p0[i], p1[i]);
            } else {
                const float xb = rr.ob[i] + rr.db[i] * (JITTER ? zlb + zsb * rr.ub[i] : zlb);
                float sn, cs;
                r2l_sincos(xa * pc.scale, sn, cs);  // (pairs 480 .. 511, one wave: identity columns and padding; pc.trig is false)
                p0[i] = pc.trig ? sn : (pc.ka >= 0 ? xa : 0.f);
                p1[i] = pc.trig ? cs : (pc.kb >= 0 ? xb : 0.f);
            }
        }
        u32x4 u0, u1, m0, m1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            u0[d] = pk(p0[2 * d], p0[2 * d + 1]);
            u1[d] = pk(p1[2 * d], p1[2 * d + 1]);
            if (EXACT) {
                m0[d] = pk(f2_res_lo(u0[d], p0[2 * d]), f2_res_hi(u0[d], p0[2 * d + 1]));
                m1[d] = pk(f2_res_lo(u1[d], p1[2 * d]), f2_res_hi(u1[d], p1[2 * d + 1]));
            }
        }
        const f16x8 b0 = __builtin_bit_cast(f16x8, u0), b1 = __builtin_bit_cast(f16x8, u1);
        __builtin_amdgcn_sched_barrier(0);
        ld_rays(rr);
        __builtin_amdgcn_sched_barrier(0);
        if (EXACT) {  // small terms first
            const f16x8 bm0 = __builtin_bit_cast(f16x8, m0), bm1 = __builtin_bit_cast(f16x8, m1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gm[e], b0, acc[e][0], 0, 0, 0);
                acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gm[e], b1, acc[e][1], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[e], bm0, acc[e][0], 0, 0, 0);
                acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[e], bm1, acc[e][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[e], b0, acc[e][0], 0, 0, 0);
            acc[e][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[e], b1, acc[e][1], 0, 0, 0);
        }
    }
    // flush: D row 8 (c>>2) + 4 kg + (c&3) of tile e -> output row o = 32 e + that; column = the pair's ka (tile 0) / kb (tile 1)
    float* sl = a.slab ? a.slab + slice * (int64_t)(R2L_W * 1024) : nullptr;
#pragma unroll
    for (int ei = 0; ei < 2; ++ei) {
        const int k = ei == 0 ? pc.ka : pc.kb;  // the real column of this lane's tile-ei values
        if (k < 0) continue;                    // padding pairs
        if (sl) {
            float* p = sl + (4 * kg) * 1024 + k;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int c = 0; c < 16; ++c) p[(32 * e + 8 * (c >> 2) + (c & 3)) * 1024] = acc[e][ei][c] * unscale;
        } else {
            float* p = a.grads + (int64_t)(4 * kg) * R2L_IN + k;
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int c = 0; c < 16; ++c) atomicAdd(p + (32 * e + 8 * (c >> 2) + (c & 3)) * R2L_IN, acc[e][ei][c] * unscale);
        }
    }
    if (kq == 0 && wave == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = bsum[e] + __shfl_xor(bsum[e], 32);
            if (kg == 0) {
                if (sl) sl[(32 * e + m) * 1024 + R2L_IN] = s;  // the slab row's padding column 1008 carries the bias partial
                else atomicAdd(a.grads + b_off_head_b() + 32 * e + m, s);
            }
        }
    }
}

int r2l_dw_head16_launch(const R2LDwHeadArgs& a, int64_t slices, hipStream_t stream) {
    const dim3 grid((unsigned)(slices * 4)), block(256);
    if (a.exact) {
        if (a.t_rand != nullptr) hipLaunchKernelGGL((r2l_dw_head16_kernel<true, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((r2l_dw_head16_kernel<false, true>), grid, block, 0, stream, a);
    } else {
        if (a.t_rand != nullptr) hipLaunchKernelGGL((r2l_dw_head16_kernel<true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((r2l_dw_head16_kernel<false, false>), grid, block, 0, stream, a);
    }
    R2L_CHECK(hipGetLastError());
    return 0;
}
