// r2l_render.hip — the volumetric-render glue of the NeRF teacher for gfx950: HBM-sized work (1.5 - 3.9 KB per ray) whose cost is
// cross-lane arithmetic, so the teacher path's shapes run a QUARTER wave per ray (scans, reductions and the sorting network inside
// 16-lane DPP rows, at VALU rate) and every other shape a wave per ray; no host round trip (the reference moves sample_pdf to the
// CPU: /root/reference/utils/create_data.py:506-511).
//   r2l_stratified_z   : z_vals = near(1-t)+far t, stratified jitter            (create_data.py:457-482)
//   r2l_raw2outputs    : alpha compositing                                       (create_data.py:335-402)
//   r2l_sample_pdf_sort: inverse-CDF importance sampling + merge-sort of depths  (helpers:283-330, create_data.py:505-515)
#include "r2l_common.h"
#include <math.h>

#define MAX_CH 4  // samples per lane: supports S <= 256

// ---------------------------------------------------------------------------------------------------------------
__global__ void r2l_stratified_z_kernel(const float* __restrict__ near, const float* __restrict__ far, int nf_stride,
                                        const float* __restrict__ ttab, const float* __restrict__ t_rand,
                                        float* __restrict__ z_out, int64_t R, int S) {
    // ttab[0..S) = t_vals, ttab[S..2S) = 1 - t_vals (both computed by the host exactly as torch does)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R * S; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / S;
        const int s = (int)(i % S);
        const float n = near[r * nf_stride], f = far[r * nf_stride];
        auto zv = [&](int k) {
            float a = n * ttab[S + k];
            r2l_no_pack(a);  // (no packed multiply + swizzled add for the pair of products: r2l_common.h)
            return a + f * ttab[k];
        };
        float z = zv(s);
        if (t_rand != nullptr) {
            const float lo = s == 0 ? z : .5f * (z + zv(s - 1));
            const float up = s == S - 1 ? z : .5f * (zv(s + 1) + z);
            z = lo + (up - lo) * t_rand[i];
        }
        z_out[i] = z;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// raw2outputs: one wave per ray, lane l owns samples [l*CH, l*CH+CH).  HBM-bound: S*20 + 12 B read, 24 (+ 4 S) B written per
// ray.  Round 5: a wave takes RPW rays and issues the loads of ALL of them before the first is composited
// (round 4's one-ray waves spent half their life in the scan / reduction with nothing in flight: 2.7 - 4.4 TB/s), and the
// scan and the five sums run on DPP row shifts / broadcasts (VALU, ~2 cycles each) instead of 37 ds_bpermute round trips per
// ray through the LDS crossbar.  The association of the product scan and of the sums differs from round 4's (and from
// torch's cumprod / sum) at the 1e-7 level, inside the 1e-4 bar (tests: rtol 2e-5).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {  // (sample_pdf kernel below: every lane needs the total)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// DPP controls (gfx9): row_shr:n = 0x110 + n (lane i reads lane i - n of its 16-lane row), row_bcast:15 = 0x142 (lane 15 of a
// row -> every lane of the next row), row_bcast:31 = 0x143 (lane 31 -> rows 2, 3), wave_shr:1 = 0x138, wave_shl:1 = 0x130.
// bound_ctrl = false: a lane without a source (or masked out by row_mask) receives `old` — the identity of the operation.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float r2l_dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
// inclusive prefix over the 64 lanes (lane 63 = the total): the scan LLVM's own atomic optimiser emits for gfx9
__device__ __forceinline__ float wave_scan_add(float v) {
    v += r2l_dpp<0x111, 0xf>(0.f, v);
    v += r2l_dpp<0x112, 0xf>(0.f, v);
    v += r2l_dpp<0x114, 0xf>(0.f, v);
    v += r2l_dpp<0x118, 0xf>(0.f, v);
    v += r2l_dpp<0x142, 0xa>(0.f, v);
    v += r2l_dpp<0x143, 0xc>(0.f, v);
    return v;
}
// five sums at once (rgb, depth, acc of one ray): the same six steps as v_add_f32 WITH the DPP modifier (one instruction per step
// and value instead of v_mov_b32_dpp + v_add_f32), step-major over the five values so that a register written by one step is read
// by DPP four instructions later (the VALU-write -> DPP-read hazard needs two wait states; the hazard recognizer does not look into
// inline asm: s_nop in front for the values the compiler wrote last).  bound_ctrl: lanes without a source add 0.
__device__ __forceinline__ void wave_scan_add5(float& a, float& b, float& c, float& d, float& e) {
#define R2O_STEP(CTRL)                                \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"           \
    "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"           \
    "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"           \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"           \
    "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 R2O_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 R2O_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 R2O_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 R2O_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                 R2O_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 R2O_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
#undef R2O_STEP
}
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= r2l_dpp<0x111, 0xf>(1.f, v);
    v *= r2l_dpp<0x112, 0xf>(1.f, v);
    v *= r2l_dpp<0x114, 0xf>(1.f, v);
    v *= r2l_dpp<0x118, 0xf>(1.f, v);
    v *= r2l_dpp<0x142, 0xa>(1.f, v);
    v *= r2l_dpp<0x143, 0xc>(1.f, v);
    return v;
}

template <int CH, int RPW>
__global__ __launch_bounds__(256) void r2l_raw2outputs_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                              const float* __restrict__ rays_d,
                                                              const float* __restrict__ noise, int white_bkgd,
                                                              float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                                              float* __restrict__ acc_map, float* __restrict__ weights,
                                                              float* __restrict__ depth_map, int64_t R, int S) {
    const int lane = threadIdx.x & 63;
    const int64_t ray0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (ray0 >= R) return;
    // ---- every load of the wave's RPW rays first (bytes in flight: RPW x (20 S + 12) per wave) ----
    f32x4 v[RPW][CH];
    float zz[RPW][CH], dd[RPW][3];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t ray = ray0 + r < R ? ray0 + r : R - 1;  // (a tail wave re-reads the last ray; nothing is stored for it)
        const float* zr = z + ray * S;
        const float* rr = raw + ray * (int64_t)S * 4;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int s = lane * CH + c;
            v[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            zz[r][c] = 0.f;
            if (s < S) {
                v[r][c] = *reinterpret_cast<const f32x4*>(rr + 4 * s);
                zz[r][c] = zr[s];
                if (noise != nullptr) v[r][c][3] += noise[ray * S + s];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dd[r][k] = rays_d[ray * 3 + k];
    }
    // ---- composite ray by ray ----
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t ray = ray0 + r;
        if (ray >= R) break;  // wave-uniform
        const float dn = sqrtf(dd[r][0] * dd[r][0] + dd[r][1] * dd[r][1] + dd[r][2] * dd[r][2]);  // torch.norm(rays_d[..., None, :], dim=-1)
        const float z_next_lane = r2l_dpp<0x130, 0xf>(0.f, zz[r][0]);  // lane l + 1's first sample = sample (l + 1) CH
        float al[CH], col[CH][3], pl = 1.0f;  // pl = product of (1-alpha+1e-10) over this lane's chunk
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int s = lane * CH + c;
            al[c] = 0.f; col[c][0] = col[c][1] = col[c][2] = 0.f;
            if (s < S) {
                const float z0 = zz[r][c];
                const float zn = c + 1 < CH ? zz[r][c + 1 < CH ? c + 1 : c] : z_next_lane;
                float dist = (s + 1 < S) ? zn - z0 : 1e10f;
                dist = dist * dn;
                // (round 5: v_exp_f32 / v_rcp_f32 — 1 ulp each — instead of expf and IEEE division: the kernel was VALU-bound,
                // 1200 VALU instructions per wave of four rays, 80 of them division steps, 17 us for 51 MB; the results move by
                // ~1e-7, the parity bar on the maps is 1e-4)
                al[c] = 1.0f - __expf(-fmaxf(v[r][c][3], 0.f) * dist);
#pragma unroll
                for (int k = 0; k < 3; ++k) col[c][k] = __builtin_amdgcn_rcpf(1.0f + __expf(-v[r][c][k]));
                pl *= (1.0f - al[c]) + 1e-10f;
            }
        }
        // exclusive product scan across lanes
        float T = r2l_dpp<0x138, 0xf>(1.0f, wave_scan_mul(pl));
        float sr = 0.f, sg_ = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int s = lane * CH + c;
            if (s < S) {
                const float w = al[c] * T;
                if (weights != nullptr) weights[ray * S + s] = w;
                sr += w * col[c][0]; sg_ += w * col[c][1]; sb += w * col[c][2];
                sd += w * zz[r][c]; sa += w;
                T *= (1.0f - al[c]) + 1e-10f;
            }
        }
        wave_scan_add5(sr, sg_, sb, sd, sa);
        if (lane == 63) {  // (the inclusive scans' last lane holds the totals)
            const float q = sd * __builtin_amdgcn_rcpf(sa);   // (0 * inf = NaN for the empty ray, as 0 / 0)
            const float m = (q != q) ? q : fmaxf(1e-10f, q);  // torch.max propagates NaN (empty ray: 0/0)
            disp_map[ray] = __builtin_amdgcn_rcpf(m);
            acc_map[ray] = sa;
            depth_map[ray] = sd;
            const float bg = white_bkgd ? (1.0f - sa) : 0.f;
            rgb_map[ray * 3 + 0] = sr + bg;
            rgb_map[ray * 3 + 1] = sg_ + bg;
            rgb_map[ray * 3 + 2] = sb + bg;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// raw2outputs for S = 16 * ROWS (64, 128, 192, 256: every sample count the teacher path uses): a QUARTER wave per ray.
// The 64-lanes-per-ray kernel above is VALU-issue bound, not HBM bound (round-5 PMC: 649 VALU instructions per wave of four
// rays, two thirds of them the 6-step wave scans and their register moves, HBM traffic = the algorithmic bytes at 0.44 of
// peak).  Here lane l of a 16-lane DPP row owns samples l, l + 16, l + 32 ... of its ray, so that
//   * every load / store instruction of the wave still covers whole 64 B .. 256 B runs (row c of a ray = 16 consecutive samples),
//   * the transmittance is a 4-step scan INSIDE a DPP row (row_shr never crosses a row: no masks, no identity moves) times a
//     carry that every lane of the row holds (row total by 4 butterfly steps: quad_perm, row_half_mirror, row_mirror),
//   * the five sums are reduced once per ray over 16 lanes instead of 64,
//   * the scans of the ROWS rows are independent of each other: issued step-major over four rows at a time, so that no DPP
//     instruction reads a register written less than two instructions earlier (the VALU-write -> DPP-read hazard).
// ~230 VALU instructions per wave of four rays at S = 64 instead of 649.  Association of products / sums differs from the
// kernel above and from torch at the 1e-7 level (tests: rtol 2e-5 on the maps, the bar is 1e-4).
// ---------------------------------------------------------------------------------------------------------------
#define R2O_DPP4(OP, CTRL)                                                                   \
    OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define R2O_DPP4B(OP, CTRL)                                                                  \
    OP " %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                 \
    OP " %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
// e[0..3]: in-row inclusive product scan (a lane without a source keeps its value: no bound_ctrl); t[0..3]: in-row product of
// all 16 lanes, left in every lane.
__device__ __forceinline__ void r2o_row_scan_total4(float* e, float* t) {
    asm volatile("s_nop 1\n\t"
                 R2O_DPP4("v_mul_f32_dpp", "row_shr:1") R2O_DPP4B("v_mul_f32_dpp", "quad_perm:[1,0,3,2]")
                 R2O_DPP4("v_mul_f32_dpp", "row_shr:2") R2O_DPP4B("v_mul_f32_dpp", "quad_perm:[2,3,0,1]")
                 R2O_DPP4("v_mul_f32_dpp", "row_shr:4") R2O_DPP4B("v_mul_f32_dpp", "row_half_mirror")
                 R2O_DPP4("v_mul_f32_dpp", "row_shr:8") R2O_DPP4B("v_mul_f32_dpp", "row_mirror")
                 "s_nop 1"
                 : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
}
// five in-row sums, left in every lane of the row
__device__ __forceinline__ void r2o_row_total5(float& a, float& b, float& c, float& d, float& e) {
#define R2O_STEP5(CTRL)                                                    \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t"
                 R2O_STEP5("quad_perm:[1,0,3,2]") R2O_STEP5("quad_perm:[2,3,0,1]") R2O_STEP5("row_half_mirror") R2O_STEP5("row_mirror")
                 "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
#undef R2O_STEP5
}
#undef R2O_DPP4
#undef R2O_DPP4B

template <int ROWS, bool WEIGHTS>
__global__ __launch_bounds__(256) void r2l_raw2outputs16_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                const float* __restrict__ rays_d,
                                                                const float* __restrict__ noise, int white_bkgd,
                                                                float* __restrict__ rgb_map, float* __restrict__ disp_map,
                                                                float* __restrict__ acc_map, float* __restrict__ weights,
                                                                float* __restrict__ depth_map, int64_t R) {
    static_assert(ROWS % 4 == 0, "rows are scanned four at a time");
    constexpr int S = 16 * ROWS;
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    const int64_t ray_w = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = ray_w < R;
    const int64_t ray = live ? ray_w : R - 1;  // (a tail row re-reads the last ray; nothing is stored for it)
    // ---- every load of the wave's four rays first ----
    const float* zr = z + ray * S + l16;
    const float* rr = raw + (ray * S + l16) * 4;
    f32x4 v[ROWS];
    float zz[ROWS];
#pragma unroll
    for (int c = 0; c < ROWS; ++c) {
        v[c] = *reinterpret_cast<const f32x4*>(rr + 64 * c);
        zz[c] = zr[16 * c];
    }
    if (noise != nullptr) {  // (wave-uniform)
#pragma unroll
        for (int c = 0; c < ROWS; ++c) v[c][3] += noise[ray * S + l16 + 16 * c];
    }
    const float d0 = rays_d[ray * 3 + 0], d1 = rays_d[ray * 3 + 1], d2 = rays_d[ray * 3 + 2];
    const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);  // torch.norm(rays_d[..., None, :], dim=-1)
    // ---- alpha and the per-sample transmittance factor of every row ----
    float al[ROWS], e[ROWS], tot[ROWS];
#pragma unroll
    for (int c = 0; c < ROWS; ++c) {
        // z of sample s + 1: the next lane of the row; for lane 15 the row's lane 0 of the NEXT row of samples (row_ror:15)
        float zn = r2l_dpp<0x12f, 0xf>(0.f, zz[c]);
        if (c + 1 < ROWS) {
            const float zw = r2l_dpp<0x12f, 0xf>(0.f, zz[c + 1]);
            zn = l16 == 15 ? zw : zn;
        }
        float dist = zn - zz[c];
        if (c + 1 == ROWS) dist = l16 == 15 ? 1e10f : dist;
        dist = dist * dn;
        al[c] = 1.0f - __expf(-fmaxf(v[c][3], 0.f) * dist);
        tot[c] = (1.0f - al[c]) + 1e-10f;
        e[c] = r2l_dpp<0x111, 0xf>(1.0f, tot[c]);  // shifted by one lane: the scan below comes out EXCLUSIVE
    }
#pragma unroll
    for (int c = 0; c < ROWS; c += 4) r2o_row_scan_total4(e + c, tot + c);
    // ---- weights and the five sums ----
    float carry = 1.0f, sr = 0.f, sg_ = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int c = 0; c < ROWS; ++c) {
        const float w = al[c] * (carry * e[c]);
        carry *= tot[c];
        if (WEIGHTS) {
            if (live) weights[ray * S + l16 + 16 * c] = w;
        }
        sr += w * __builtin_amdgcn_rcpf(1.0f + __expf(-v[c][0]));
        sg_ += w * __builtin_amdgcn_rcpf(1.0f + __expf(-v[c][1]));
        sb += w * __builtin_amdgcn_rcpf(1.0f + __expf(-v[c][2]));
        sd += w * zz[c];
        sa += w;
    }
    r2o_row_total5(sr, sg_, sb, sd, sa);
    if (l16 == 0 && live) {
        const float q = sd * __builtin_amdgcn_rcpf(sa);   // (0 * inf = NaN for the empty ray, as 0 / 0)
        const float m = (q != q) ? q : fmaxf(1e-10f, q);  // torch.max propagates NaN (empty ray: 0/0)
        disp_map[ray] = __builtin_amdgcn_rcpf(m);
        acc_map[ray] = sa;
        depth_map[ray] = sd;
        const float bg = white_bkgd ? (1.0f - sa) : 0.f;
        rgb_map[ray * 3 + 0] = sr + bg;
        rgb_map[ray * 3 + 1] = sg_ + bg;
        rgb_map[ray * 3 + 2] = sb + bg;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sample_pdf + sort-merge.  One wave per ray.  S coarse samples (S <= 64), NI new samples (NI <= 192, S+NI <= 256).
// Round 5 (second form).  The first form did everything through LDS behind block barriers — the cdf as a 62-step loop of ONE lane
// over LDS, the merge as a 36-stage bitonic network of LDS compare-exchanges with a __syncthreads per stage — and took 90 us per
// 32 768-ray chunk (2.3 KB per ray: 0.65 TB/s), every step latency-bound.  Now:
//   * cdf: torch.cumsum's left-to-right fp32 order is kept (the det sample at u = 1 sits on cdf[-1] to the ulp: tests), as 63
//     dependent `v_add_f32 ... wave_shr:1` steps in registers — lane k ends with ((p0 + p1) + p2) ... + pk — ~8 clocks a step
//     instead of an LDS round trip;
//   * the sort runs in registers: lane l holds elements 4l .. 4l+3 of the 256-element network (all comparisons "lower index gets the
//     smaller": each merge phase opens with the mirror step i <-> i ^ (k - 1)); partners in other lanes come through DPP
//     (quad_perm, row_half_mirror, row_mirror, row_ror:8) and, for the three lane distances that leave a 16-lane row, ds_bpermute:
//     12 LDS-crossbar operations per ray instead of ~600 LDS accesses, no barrier;
//   * the sorted row leaves as one 16-byte store per lane.
// Only the inverse-cdf search still reads LDS (cdf and bin edges of the wave's own ray; one block barrier in front of it).
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float r2o_perm(float v) {  // (every lane has a source: no `old` operand to set up)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float r2o_xlane(int byte_addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float r2o_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float r2o_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void r2o_ce(float& a, float& b) {
    const float lo = r2o_min(a, b), hi = r2o_max(a, b);
    a = lo; b = hi;
}
__device__ __forceinline__ float r2o_keep(bool keep_min, float a, float p) {
    const float lo = r2o_min(a, p), hi = r2o_max(a, p);  // (both, then a select: a ternary over the asm statements becomes branches)
    return keep_min ? lo : hi;
}

__global__ __launch_bounds__(256) void r2l_sample_pdf_sort_kernel(const float* __restrict__ z, const float* __restrict__ wts,
                                                                  const float* __restrict__ u, int64_t u_stride,
                                                                  float* __restrict__ z_samples, float* __restrict__ z_all,
                                                                  float* __restrict__ z_std, int64_t R, int S, int NI) {
    __shared__ float s_cdf[4][64];
    __shared__ float s_bins[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    const bool live = ray < R;  // idle waves of the last block run on a clamped ray (they meet the block barrier) and store nothing
    if (!live) ray = R - 1;
    const float* zr = z + ray * S;
    const int nb = S - 1;   // bins = z_mid (S-1 edges)
    const int nw = S - 2;   // weights[..., 1:-1]
    float* cdf = s_cdf[wv];
    float* bins = s_bins[wv];
    // ---- loads: this lane's coarse depth, inner weight and its (up to three) u's ----
    const float z0 = lane < S ? zr[lane] : INFINITY;
    float wl = 0.f, uu[3];
    if (lane < nw) wl = wts[ray * S + lane + 1] + 1e-5f;
#pragma unroll
    for (int m = 0; m < 3; ++m) uu[m] = lane + 64 * m < NI ? u[ray * u_stride + lane + 64 * m] : 0.f;
    // ---- bins, pdf, cdf ----
    const float z1 = r2l_dpp<0x130, 0xf>(0.f, z0);  // wave_shl:1: the next lane's depth
    if (lane < nb) bins[lane] = .5f * (z1 + z0);
    const float total = wave_sum(wl);
    const float pdf = wl / total;
    float run = pdf;  // -> inclusive prefix, summed left to right like torch.cumsum (lanes >= nw add 0)
    asm volatile("s_nop 1\n\t"
                 ".rept 63\n\t"
                 "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 ".endr"
                 : "+&v"(run) : "v"(pdf));  // (early clobber: `run` starts as a copy of `pdf` and would share its register)
    if (lane == 0) cdf[0] = 0.f;
    if (lane < nw) cdf[lane + 1] = run;
    __syncthreads();
    // ---- inverse CDF for this lane's u's ----
    float sum1 = 0.f;
    float samp[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int i = lane + 64 * m;
        samp[m] = 0.f;
        if (i < NI) {
            // searchsorted(cdf[0..nb), uu, right=True): number of entries <= uu
            int lo = 0, hi = nb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] <= uu[m]) lo = mid + 1; else hi = mid;
            }
            const int below = lo - 1 > 0 ? lo - 1 : 0;
            const int above = lo < nb - 1 ? lo : nb - 1;
            const float c0 = cdf[below], c1 = cdf[above];
            const float b0 = bins[below], b1 = bins[above];
            float denom = c1 - c0;
            denom = denom < 1e-5f ? 1.0f : denom;
            const float t = (uu[m] - c0) / denom;
            samp[m] = b0 + t * (b1 - b0);
            if (live) z_samples[ray * NI + i] = samp[m];
            sum1 += samp[m];
        }
    }
    // z_std = std(z_samples, unbiased=False)
    const float mean = wave_sum(sum1) / (float)NI;
    float sq = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if (lane + 64 * m < NI) sq += (samp[m] - mean) * (samp[m] - mean);
    sq = wave_sum(sq);
    if (live && lane == 0 && z_std != nullptr) z_std[ray] = sqrtf(sq / (float)NI);
    // ---- sort [z (S), samples (NI), +inf padding]: 256-element network, element 4 lane + r in register v_r ----
    float v0 = z0;
    float v1 = lane < NI ? samp[0] : INFINITY;
    float v2 = lane + 64 < NI ? samp[1] : INFINITY;
    float v3 = lane + 128 < NI ? samp[2] : INFINITY;
    const bool k0 = (lane & 1) == 0, k1 = (lane & 2) == 0, k2 = (lane & 4) == 0, k3 = (lane & 8) == 0, k4 = (lane & 16) == 0,
               k5 = (lane & 32) == 0;
    const int a16 = (lane ^ 16) << 2, a31 = (lane ^ 31) << 2, a63 = (lane ^ 63) << 2;
#define R2O_TAIL() r2o_ce(v0, v2); r2o_ce(v1, v3); r2o_ce(v0, v1); r2o_ce(v2, v3)   /* partner distances 2, 1: inside the lane */
#define R2O_MIRROR(F, KM)  /* i <-> i ^ (k - 1): the other lane's registers in reverse */                  \
    { const float p0 = F(v3), p1 = F(v2), p2 = F(v1), p3 = F(v0);                                           \
      v0 = r2o_keep(KM, v0, p0); v1 = r2o_keep(KM, v1, p1); v2 = r2o_keep(KM, v2, p2); v3 = r2o_keep(KM, v3, p3); }
#define R2O_XOR(F, KM)     /* i <-> i ^ j, j >= 4: the same register of lane ^ (j / 4) */                  \
    { const float p0 = F(v0), p1 = F(v1), p2 = F(v2), p3 = F(v3);                                           \
      v0 = r2o_keep(KM, v0, p0); v1 = r2o_keep(KM, v1, p1); v2 = r2o_keep(KM, v2, p2); v3 = r2o_keep(KM, v3, p3); }
#define X1(x) r2o_perm<0xB1>(x)                   /* quad_perm:[1,0,3,2]  lane ^ 1  */
#define X2(x) r2o_perm<0x4E>(x)                   /* quad_perm:[2,3,0,1]  lane ^ 2  */
#define X3(x) r2o_perm<0x1B>(x)                   /* quad_perm:[3,2,1,0]  lane ^ 3  */
#define X7(x) r2o_perm<0x141>(x)                  /* row_half_mirror      lane ^ 7  */
#define X4(x) r2o_perm<0x1B>(r2o_perm<0x141>(x))  /* (lane ^ 3) ^ 7                 */
#define X8(x) r2o_perm<0x128>(x)                  /* row_ror:8            lane ^ 8  */
#define X15(x) r2o_perm<0x140>(x)                 /* row_mirror           lane ^ 15 */
#define X16(x) r2o_xlane(a16, x)
#define X31(x) r2o_xlane(a31, x)
#define X63(x) r2o_xlane(a63, x)
    r2o_ce(v0, v1); r2o_ce(v2, v3);                                                           // k = 2
    r2o_ce(v0, v3); r2o_ce(v1, v2); r2o_ce(v0, v1); r2o_ce(v2, v3);                           // k = 4
    R2O_MIRROR(X1, k0) R2O_TAIL();                                                            // k = 8
    R2O_MIRROR(X3, k1) R2O_XOR(X1, k0) R2O_TAIL();                                            // k = 16
    R2O_MIRROR(X7, k2) R2O_XOR(X2, k1) R2O_XOR(X1, k0) R2O_TAIL();                            // k = 32
    R2O_MIRROR(X15, k3) R2O_XOR(X4, k2) R2O_XOR(X2, k1) R2O_XOR(X1, k0) R2O_TAIL();           // k = 64
    R2O_MIRROR(X31, k4) R2O_XOR(X8, k3) R2O_XOR(X4, k2) R2O_XOR(X2, k1) R2O_XOR(X1, k0) R2O_TAIL();                    // k = 128
    R2O_MIRROR(X63, k5) R2O_XOR(X16, k4) R2O_XOR(X8, k3) R2O_XOR(X4, k2) R2O_XOR(X2, k1) R2O_XOR(X1, k0) R2O_TAIL();   // k = 256
#undef X1
#undef X2
#undef X3
#undef X4
#undef X7
#undef X8
#undef X15
#undef X16
#undef X31
#undef X63
#undef R2O_XOR
#undef R2O_MIRROR
#undef R2O_TAIL
    const int tot = S + NI;
    if (live) {
        float* out = z_all + ray * tot + 4 * lane;
        if ((tot & 3) == 0) {  // (wave-uniform) rows are 16-byte aligned: one store per lane
            if (4 * lane < tot) *reinterpret_cast<f32x4*>(out) = f32x4{v0, v1, v2, v3};
        } else {
            if (4 * lane + 0 < tot) out[0] = v0;
            if (4 * lane + 1 < tot) out[1] = v1;
            if (4 * lane + 2 < tot) out[2] = v2;
            if (4 * lane + 3 < tot) out[3] = v3;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sample_pdf + sort-merge for the reference's configuration (S = 64 coarse depths, NI = 128 new ones): a QUARTER wave per ray.
// The one-ray-per-wave kernel above is VALU-issue bound (~900 instructions per ray: 47 us per 32 768-ray chunk, 0.2 of the HBM
// roofline its 2.3 KB per ray would allow).  With 16 lanes per ray and 16 elements of the 256-element network per lane,
//   * the sort (round 6): the 128 samples are sorted alone — 28 stages on the lane's 8 consecutive samples, 18 of them min / max pairs
//     between the lane's own registers, the other 10 reaching their partner inside the 16-lane DPP row (quad_perm, row_half_mirror,
//     row_mirror: no ds_bpermute, a compare + select per element) — and then MERGED with the 64 coarse depths, which arrive
//     ascending: the last phase of the 256-element network only (8 stages on 16 values per lane).  2816 compare-exchanges per ray
//     instead of the 4608 of round 5's full network, which stays in the kernel for waves whose coarse depths do not ascend (checked);
//     waves whose samples already ascend (perturb = 0: the inverse cdf is monotone in the linspace u) skip the sample sort too;
//   * the left-to-right cdf runs for four rays at once: 16 steps of (carry from the previous lane by row_shr:1, four dependent
//     adds) — the same association as torch.cumsum, element for element;
//   * sums over a ray (pdf normalisation, z_std) follow wave_sum's butterfly association (k ^ 32, 16, ... 1 over the 64 positions),
//     so this kernel and the generic one return the same bits.
// Loads and stores are 16 bytes per lane, contiguous per ray.
// ---------------------------------------------------------------------------------------------------------------
#define X1(x) r2o_perm<0xB1>(x)                   /* quad_perm:[1,0,3,2]  lane ^ 1  */
#define X2(x) r2o_perm<0x4E>(x)                   /* quad_perm:[2,3,0,1]  lane ^ 2  */
#define X3(x) r2o_perm<0x1B>(x)                   /* quad_perm:[3,2,1,0]  lane ^ 3  */
#define X7(x) r2o_perm<0x141>(x)                  /* row_half_mirror      lane ^ 7  */
#define X4(x) r2o_perm<0x1B>(r2o_perm<0x141>(x))  /* (lane ^ 3) ^ 7                 */
#define X8(x) r2o_perm<0x128>(x)                  /* row_ror:8            lane ^ 8  */
#define X15(x) r2o_perm<0x140>(x)                 /* row_mirror           lane ^ 15 */
#define DPP_X1 "quad_perm:[1,0,3,2]"
#define DPP_X2 "quad_perm:[2,3,0,1]"
#define DPP_X3 "quad_perm:[3,2,1,0]"
#define DPP_X7 "row_half_mirror"
#define DPP_X8 "row_ror:8"
#define DPP_X15 "row_mirror"
// sum over the 64 positions k = 4 l16 + j of a ray in wave_sum's association; every lane of the row gets the total
__device__ __forceinline__ float r2o_row_sum64(float a0, float a1, float a2, float a3) {
    a0 += X8(a0); a1 += X8(a1); a2 += X8(a2); a3 += X8(a3);  // k ^ 32
    a0 += X4(a0); a1 += X4(a1); a2 += X4(a2); a3 += X4(a3);  // k ^ 16
    a0 += X2(a0); a1 += X2(a1); a2 += X2(a2); a3 += X2(a3);  // k ^ 8
    a0 += X1(a0); a1 += X1(a1); a2 += X1(a2); a3 += X1(a3);  // k ^ 4
    const float b0 = a0 + a2, b1 = a1 + a3;                  // k ^ 2
    return b0 + b1;                                          // k ^ 1
}
// the same sum for the layout of the round-6 kernel: position k = 8 l' + q in register a[q] of lane l' = l16 & 7 (the two half rows
// hold the same 64 values); same association, every lane gets the total
__device__ __forceinline__ float r2o_row_sum64_8(float (&a)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += X4(a[q]);  // k ^ 32
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += X2(a[q]);  // k ^ 16
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += X1(a[q]);  // k ^ 8
    const float b0 = a[0] + a[4], b1 = a[1] + a[5], b2 = a[2] + a[6], b3 = a[3] + a[7];  // k ^ 4
    const float c0 = b0 + b2, c1 = b1 + b3;                                                // k ^ 2
    return c0 + c1;                                                                        // k ^ 1
}
template <int MASK>
__device__ __forceinline__ void r2o_inlane8(float (&v)[8]) {  // compare-exchange r <-> r ^ MASK among a lane's 8, smaller first
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r ^ MASK) > r) r2o_ce(v[r], v[r ^ MASK]);
}
__device__ __forceinline__ void r2o_tail8(float (&v)[8]) { r2o_inlane8<4>(v); r2o_inlane8<2>(v); r2o_inlane8<1>(v); }
template <int MASK>
__device__ __forceinline__ void r2o_inlane(float (&v)[16]) {  // compare-exchange r <-> r ^ MASK inside the lane, smaller first
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((r ^ MASK) > r) r2o_ce(v[r], v[r ^ MASK]);
}
__device__ __forceinline__ void r2o_tail16(float (&v)[16]) { r2o_inlane<8>(v); r2o_inlane<4>(v); r2o_inlane<2>(v); r2o_inlane<1>(v); }
// `take`: the lanes of the wave-uniform 64-bit mask upper_ hold the LARGER index of the pair (they keep the larger value).  The
// predicate is kept in SGPRs end to end — v_cmp -> vcc, s_xor_b64 with the lane mask, v_cndmask on the result —: as a per-lane
// bool, hipcc turns the `!= upper` into compare + select + compare on the VALU once the value crosses a branch (round 6: the
// sort / merge paths sit behind wave-uniform branches), five VALU instructions per element instead of two + the DPP move.
#define R2O_TAKE(v_, p_, upper_) { const float pp = (p_);                                                   \
    const unsigned long long c = __builtin_amdgcn_ballot_w64(pp < (v_)) ^ (upper_);                          \
    (v_) = __builtin_amdgcn_inverse_ballot_w64(c) ? pp : (v_); }
// Cross-lane steps whose "upper" lanes are whole DPP BANKS (lane bits 2, 3 of the row: partners lane ^ 4, lane ^ 8, and the mirror
// steps lane ^ 7, lane ^ 15) need no compare-select at all (round 6): v_min_f32_dpp writes the lower banks, v_max_f32_dpp the upper
// ones — the DPP bank_mask disables the other lanes' writes — both reading the partner through their DPP operand: two VALU
// instructions per element, into fresh registers, instead of a DPP move + compare + select (lane ^ 4: two moves).  (VOPC has no
// DPP form on gfx9, so the steps inside a bank — lane ^ 1, ^ 2, ^ 3 — keep the move + v_cmp + s_xor + v_cndmask form.)
// (hand-placed s_nop 1 at both ends of a block: a VALU write of a register needs two wait states before a DPP read of it, and the
// hazard recogniser does not look into inline asm — neither at what the block reads first nor at what it wrote last.)
#define R2O_ASM_BANK1(O, A, B, LO, HI, BLO, BHI)                                                          \
    "v_min_f32_dpp %" #O ", %" #A ", %" #B " " LO " row_mask:0xf bank_mask:" BLO "\n\t"                  \
    "v_max_f32_dpp %" #O ", %" #A ", %" #B " " HI " row_mask:0xf bank_mask:" BHI "\n\t"
// x[r] <-> the same register of the partner lane (r = 0 .. 7)
#define R2O_ASM_BANK_XOR8(x, LO, HI, BLO, BHI) { float t_[8];                                                                             \
    asm volatile("s_nop 1\n\t" R2O_ASM_BANK1(0, 8, 8, LO, HI, BLO, BHI) R2O_ASM_BANK1(1, 9, 9, LO, HI, BLO, BHI)                           \
                 R2O_ASM_BANK1(2, 10, 10, LO, HI, BLO, BHI) R2O_ASM_BANK1(3, 11, 11, LO, HI, BLO, BHI)                                     \
                 R2O_ASM_BANK1(4, 12, 12, LO, HI, BLO, BHI) R2O_ASM_BANK1(5, 13, 13, LO, HI, BLO, BHI)                                     \
                 R2O_ASM_BANK1(6, 14, 14, LO, HI, BLO, BHI) R2O_ASM_BANK1(7, 15, 15, LO, HI, BLO, BHI) "s_nop 1"                           \
                 : "=&v"(t_[0]), "=&v"(t_[1]), "=&v"(t_[2]), "=&v"(t_[3]), "=&v"(t_[4]), "=&v"(t_[5]), "=&v"(t_[6]), "=&v"(t_[7])           \
                 : "v"((x)[0]), "v"((x)[1]), "v"((x)[2]), "v"((x)[3]), "v"((x)[4]), "v"((x)[5]), "v"((x)[6]), "v"((x)[7]));               \
    _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) (x)[r_] = t_[r_]; }
// x[r] <-> register 7 - r of the partner lane
#define R2O_ASM_BANK_MIRROR8(x, CTRL, BLO, BHI) { float t_[8];                                                                            \
    asm volatile("s_nop 1\n\t" R2O_ASM_BANK1(0, 15, 8, CTRL, CTRL, BLO, BHI) R2O_ASM_BANK1(1, 14, 9, CTRL, CTRL, BLO, BHI)                 \
                 R2O_ASM_BANK1(2, 13, 10, CTRL, CTRL, BLO, BHI) R2O_ASM_BANK1(3, 12, 11, CTRL, CTRL, BLO, BHI)                             \
                 R2O_ASM_BANK1(4, 11, 12, CTRL, CTRL, BLO, BHI) R2O_ASM_BANK1(5, 10, 13, CTRL, CTRL, BLO, BHI)                             \
                 R2O_ASM_BANK1(6, 9, 14, CTRL, CTRL, BLO, BHI) R2O_ASM_BANK1(7, 8, 15, CTRL, CTRL, BLO, BHI) "s_nop 1"                     \
                 : "=&v"(t_[0]), "=&v"(t_[1]), "=&v"(t_[2]), "=&v"(t_[3]), "=&v"(t_[4]), "=&v"(t_[5]), "=&v"(t_[6]), "=&v"(t_[7])           \
                 : "v"((x)[0]), "v"((x)[1]), "v"((x)[2]), "v"((x)[3]), "v"((x)[4]), "v"((x)[5]), "v"((x)[6]), "v"((x)[7]));               \
    _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) (x)[r_] = t_[r_]; }
// static form (the merge's first step: P <-> P ^ 255): the lower half (registers a[0..7]) keeps the smaller of a[r] and the partner
// lane's b[7 - r], the upper half (b[0..7]) the larger of b[r] and the partner's a[7 - r] — one DPP instruction per element, new
// values into fresh registers
#define R2O_ASM_MM1(OP, O, A, B) OP " %" #O ", %" #A ", %" #B " row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define R2O_ASM_MERGE_MIRROR(na, nb, a, b)                                                                                                \
    asm volatile("s_nop 1\n\t" R2O_ASM_MM1("v_min_f32_dpp", 0, 23, 8) R2O_ASM_MM1("v_min_f32_dpp", 1, 22, 9)                               \
                 R2O_ASM_MM1("v_min_f32_dpp", 2, 21, 10) R2O_ASM_MM1("v_min_f32_dpp", 3, 20, 11)                                           \
                 R2O_ASM_MM1("v_min_f32_dpp", 4, 19, 12) R2O_ASM_MM1("v_min_f32_dpp", 5, 18, 13)                                           \
                 R2O_ASM_MM1("v_min_f32_dpp", 6, 17, 14) R2O_ASM_MM1("v_min_f32_dpp", 7, 16, 15) "s_nop 1"                                 \
                 : "=&v"((na)[0]), "=&v"((na)[1]), "=&v"((na)[2]), "=&v"((na)[3]), "=&v"((na)[4]), "=&v"((na)[5]), "=&v"((na)[6]), "=&v"((na)[7]) \
                 : "v"((a)[0]), "v"((a)[1]), "v"((a)[2]), "v"((a)[3]), "v"((a)[4]), "v"((a)[5]), "v"((a)[6]), "v"((a)[7]),                 \
                   "v"((b)[0]), "v"((b)[1]), "v"((b)[2]), "v"((b)[3]), "v"((b)[4]), "v"((b)[5]), "v"((b)[6]), "v"((b)[7]));               \
    asm volatile("s_nop 1\n\t" R2O_ASM_MM1("v_max_f32_dpp", 0, 15, 16) R2O_ASM_MM1("v_max_f32_dpp", 1, 14, 17)                             \
                 R2O_ASM_MM1("v_max_f32_dpp", 2, 13, 18) R2O_ASM_MM1("v_max_f32_dpp", 3, 12, 19)                                           \
                 R2O_ASM_MM1("v_max_f32_dpp", 4, 11, 20) R2O_ASM_MM1("v_max_f32_dpp", 5, 10, 21)                                           \
                 R2O_ASM_MM1("v_max_f32_dpp", 6, 9, 22) R2O_ASM_MM1("v_max_f32_dpp", 7, 8, 23) "s_nop 1"                                   \
                 : "=&v"((nb)[0]), "=&v"((nb)[1]), "=&v"((nb)[2]), "=&v"((nb)[3]), "=&v"((nb)[4]), "=&v"((nb)[5]), "=&v"((nb)[6]), "=&v"((nb)[7]) \
                 : "v"((a)[0]), "v"((a)[1]), "v"((a)[2]), "v"((a)[3]), "v"((a)[4]), "v"((a)[5]), "v"((a)[6]), "v"((a)[7]),                 \
                   "v"((b)[0]), "v"((b)[1]), "v"((b)[2]), "v"((b)[3]), "v"((b)[4]), "v"((b)[5]), "v"((b)[6]), "v"((b)[7]))
#define R2O_MIRROR16(F, UPPER) { float q[16];                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) q[r] = F(v[15 - r]);                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) R2O_TAKE(v[r], q[r], UPPER) }
#define R2O_XOR16(F, UPPER) { _Pragma("unroll") for (int r = 0; r < 16; ++r) R2O_TAKE(v[r], F(v[r]), UPPER) }
#define R2O_MIRROR8(F, UPPER) { float q[8];                                                   \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) q[r] = F(sm[7 - r]);                        \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) R2O_TAKE(sm[r], q[r], UPPER) }
#define R2O_XOR8(F, UPPER) { _Pragma("unroll") for (int r = 0; r < 8; ++r) R2O_TAKE(sm[r], F(sm[r]), UPPER) }

__global__ __launch_bounds__(256) void r2l_sample_pdf_sort16_kernel(const float* __restrict__ z, const float* __restrict__ wts,
                                                                    const float* __restrict__ u, int64_t u_stride,
                                                                    float* __restrict__ z_samples, float* __restrict__ z_all,
                                                                    float* __restrict__ z_std, int64_t R) {
    constexpr int S = 64, NI = 128, NB = S - 1, NW = S - 2;
    // (row stride 68 words: the four rays of a wave — and the two of a 32-lane LDS group — read the SAME index of their tables in the
    // first steps of the search; at stride 64 those are 4 addresses in one bank.  PMC, round 6: 68 % of the kernel's LDS cycles were
    // bank conflicts)
    __shared__ __attribute__((aligned(16))) float s_cdf[16][68];
    __shared__ __attribute__((aligned(16))) float s_bins[16][68];
    const int lane = threadIdx.x & 63, l16 = lane & 15, slot = threadIdx.x >> 4;
    int64_t ray = (int64_t)blockIdx.x * 16 + slot;
    const bool live = ray < R;  // rows past the end run on a clamped ray (they meet the block barrier) and store nothing
    if (!live) ray = R - 1;
    float* cdf = s_cdf[slot];
    float* bins = s_bins[slot];
    // ---- loads: four coarse depths, four weights and eight u's per lane ----
    const f32x4 z4 = *reinterpret_cast<const f32x4*>(z + ray * S + 4 * l16);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wts + ray * S + 4 * l16);
    // (samples 8 l16 .. 8 l16 + 7: the lane's eight are consecutive in u order — the order the sort below starts from)
    const f32x4 ua = *reinterpret_cast<const f32x4*>(u + ray * u_stride + 8 * l16);
    const f32x4 ub = *reinterpret_cast<const f32x4*>(u + ray * u_stride + 8 * l16 + 4);
    // the lane's eight coarse depths for the merge (lanes 0 - 7; the others hold the +inf padding)
    f32x4 za = *reinterpret_cast<const f32x4*>(z + ray * S + 8 * (l16 & 7));
    f32x4 zb = *reinterpret_cast<const f32x4*>(z + ray * S + 8 * (l16 & 7) + 4);
    // ---- bins (position k = 4 l16 + j: .5 (z[k+1] + z[k]); k = 63 is never read) ----
    const float zn = r2l_dpp<0x101, 0xf>(0.f, z4[0]);  // row_shl:1: the next lane's first depth
    *reinterpret_cast<f32x4*>(bins + 4 * l16) = f32x4{.5f * (z4[1] + z4[0]), .5f * (z4[2] + z4[1]), .5f * (z4[3] + z4[2]), .5f * (zn + z4[3])};
    // ---- pdf of the inner weights: position k holds weights[k + 1] + 1e-5 for k < NW ----
    const float wn = r2l_dpp<0x101, 0xf>(0.f, w4[0]);
    float wl[4] = {w4[1] + 1e-5f, w4[2] + 1e-5f, w4[3] + 1e-5f, wn + 1e-5f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * l16 + j >= NW) wl[j] = 0.f;
    const float total = r2o_row_sum64(wl[0], wl[1], wl[2], wl[3]);
    const float p0 = wl[0] / total, p1 = wl[1] / total, p2 = wl[2] / total, p3 = wl[3] / total;
    // ---- cdf: inclusive prefix, summed left to right like torch.cumsum.  Step t makes lane t final (its carry, lane t - 1's last
    //      prefix, became final in step t - 1); a lane that is final recomputes the same values.  (s_nop: VALU write -> DPP read) ----
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    asm volatile("s_nop 1\n\t"
                 ".rept 16\n\t"
                 "v_add_f32_dpp %0, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_add_f32 %1, %0, %5\n\t"
                 "v_add_f32 %2, %1, %6\n\t"
                 "v_add_f32 %3, %2, %7\n\t"
                 "s_nop 1\n\t"
                 ".endr"
                 : "+&v"(c0), "+&v"(c1), "+&v"(c2), "+&v"(c3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
    if (l16 == 0) cdf[0] = 0.f;
    {
        const float cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * l16 + j < NW) cdf[4 * l16 + j + 1] = cc[j];
    }
    __syncthreads();
    // ---- inverse CDF: sample i = 8 l16 + q ----
    float samp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float uu = q < 4 ? ua[q & 3] : ub[q & 3];
        // searchsorted(cdf[0..NB), uu, right=True) = number of entries <= uu (NB = 63 = 32 + 16 + ... + 1: six fixed steps)
        int pos = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            pos += cdf[pos + step - 1] <= uu ? step : 0;
        const int below = pos - 1 > 0 ? pos - 1 : 0;
        const int above = pos < NB - 1 ? pos : NB - 1;
        const float k0 = cdf[below], k1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float denom = k1 - k0;
        denom = denom < 1e-5f ? 1.0f : denom;
        const float t = (uu - k0) / denom;
        samp[q] = b0 + t * (b1 - b0);
    }
    if (live) {
        *reinterpret_cast<f32x4*>(z_samples + ray * NI + 8 * l16) = f32x4{samp[0], samp[1], samp[2], samp[3]};
        *reinterpret_cast<f32x4*>(z_samples + ray * NI + 8 * l16 + 4) = f32x4{samp[4], samp[5], samp[6], samp[7]};
    }
    // ---- z_std = std(z_samples, unbiased=False), in the generic kernel's association: position k = 0 .. 63 sums its samples k and
    //      k + 64 — lane l16 < 8 holds sample k = 8 l16 + q, lane l16 + 8 its partner (row_ror:8; the sum commutes, so both half
    //      rows end up with the same 64 values) — then the butterfly over the positions ----
    float t8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t8[q] = samp[q] + X8(samp[q]);
    const float mean = r2o_row_sum64_8(t8) / (float)NI;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float e = (samp[q] - mean) * (samp[q] - mean);
        t8[q] = e + X8(e);
    }
    const float sqs = r2o_row_sum64_8(t8);
    if (live && l16 == 0 && z_std != nullptr) z_std[ray] = sqrtf(sqs / (float)NI);
    // ---- z_all = sort([z (64), samples (128)]).  Round 6: SORT the 128 samples (28 stages on 8 values per lane), then MERGE them
    //      with the 64 coarse depths, which arrive ascending (one bitonic merge phase of the 256-element network: 8 stages on 16
    //      values per lane) — 2816 compare-exchanges instead of the 4608 of sorting all 256 from scratch.  Both shortcuts are
    //      CHECKED, per wave: a wave whose rays' coarse depths are not ascending (never, for the z the render stack produces:
    //      stratified depths are monotone) takes the full network below; a wave whose samples already ascend (perturb = 0: u is
    //      a linspace and the inverse cdf is monotone) skips their sort.  The result is the sorted multiset either way: same bits
    //      as torch.sort.
    // lanes whose l16 has bit b set, as wave-uniform masks (l16 = lane & 15: the same four rows in every wave)
    const unsigned long long u0 = 0xAAAAAAAAAAAAAAAAull, u1 = 0xCCCCCCCCCCCCCCCCull, u2 = 0xF0F0F0F0F0F0F0F0ull, u3 = 0xFF00FF00FF00FF00ull;
    if (l16 >= 8) { za = f32x4{INFINITY, INFINITY, INFINITY, INFINITY}; zb = za; }
    const float z_next = r2l_dpp<0x101, 0xf>(INFINITY, za[0]);  // row_shl:1: the next lane's first (lane 15: none -> +inf)
    const bool z_asc = za[0] <= za[1] && za[1] <= za[2] && za[2] <= za[3] && za[3] <= zb[0] && zb[0] <= zb[1] && zb[1] <= zb[2] &&
                       zb[2] <= zb[3] && zb[3] <= z_next;
    if (__builtin_amdgcn_ballot_w64(z_asc) == ~0ull) {
        float sm[8] = {samp[0], samp[1], samp[2], samp[3], samp[4], samp[5], samp[6], samp[7]};
        const float s_next = r2l_dpp<0x101, 0xf>(INFINITY, sm[0]);
        const bool s_asc = sm[0] <= sm[1] && sm[1] <= sm[2] && sm[2] <= sm[3] && sm[3] <= sm[4] && sm[4] <= sm[5] && sm[5] <= sm[6] &&
                           sm[6] <= sm[7] && sm[7] <= s_next;
        if (__builtin_amdgcn_ballot_w64(s_asc) != ~0ull) {  // sort the samples: element 8 l16 + r of a 128-element network in sm[r]
            r2o_inlane8<1>(sm);                                                                        // k = 2
            r2o_inlane8<3>(sm); r2o_inlane8<1>(sm);                                                    // k = 4
            r2o_inlane8<7>(sm); r2o_inlane8<2>(sm); r2o_inlane8<1>(sm);                                // k = 8
            R2O_MIRROR8(X1, u0) r2o_tail8(sm);                                                                      // k = 16
            R2O_MIRROR8(X3, u1) R2O_XOR8(X1, u0) r2o_tail8(sm);                                                     // k = 32
            R2O_ASM_BANK_MIRROR8(sm, "row_half_mirror", "0x5", "0xa") R2O_XOR8(X2, u1) R2O_XOR8(X1, u0) r2o_tail8(sm);  // k = 64
            R2O_ASM_BANK_MIRROR8(sm, "row_mirror", "0x3", "0xc") R2O_ASM_BANK_XOR8(sm, "row_shl:4", "row_shr:4", "0x5", "0xa")
            R2O_XOR8(X2, u1) R2O_XOR8(X1, u0) r2o_tail8(sm);                                                        // k = 128
        }
        // merge: position P = 128 half + 8 l16 + r; half 0 = the sorted samples, half 1 = [z ascending, +inf x 64]
        float v[16] = {sm[0], sm[1], sm[2], sm[3], sm[4], sm[5], sm[6], sm[7], za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
        {   // P <-> P ^ 255: the other half, lane ^ 15, register 7 - r; the lower half keeps the smaller
            float na[8], nb[8];
            float* lo = v; float* hi = v + 8;
            R2O_ASM_MERGE_MIRROR(na, nb, lo, hi);
#pragma unroll
            for (int r = 0; r < 8; ++r) { v[r] = na[r]; v[8 + r] = nb[r]; }
        }
        {   // P ^ 64, 32: lane ^ 8, lane ^ 4 (bank steps); P ^ 16, 8: lane ^ 2, lane ^ 1
            float* lo = v; float* hi = v + 8;
            R2O_ASM_BANK_XOR8(lo, "row_ror:8", "row_ror:8", "0x3", "0xc") R2O_ASM_BANK_XOR8(hi, "row_ror:8", "row_ror:8", "0x3", "0xc")
            R2O_ASM_BANK_XOR8(lo, "row_shl:4", "row_shr:4", "0x5", "0xa") R2O_ASM_BANK_XOR8(hi, "row_shl:4", "row_shr:4", "0x5", "0xa")
            R2O_XOR16(X2, u1) R2O_XOR16(X1, u0)
        }
        r2o_inlane<4>(v); r2o_inlane<2>(v); r2o_inlane<1>(v);                        // P ^ 4, 2, 1: inside each half of the lane
        if (live) {  // lane l16: z_all[8 l16 .. + 7] and, of the upper half, z_all[128 + 8 l16 .. + 7] (l16 < 8: 192 depths)
            float* out = z_all + ray * (S + NI) + 8 * l16;
            *reinterpret_cast<f32x4*>(out) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(out + 4) = f32x4{v[4], v[5], v[6], v[7]};
            if (l16 < 8) {
                *reinterpret_cast<f32x4*>(out + 128) = f32x4{v[8], v[9], v[10], v[11]};
                *reinterpret_cast<f32x4*>(out + 132) = f32x4{v[12], v[13], v[14], v[15]};
            }
        }
        return;
    }
    // ---- full network (coarse depths not ascending): [z (64), samples (128), +inf (64)], element 16 l16 + r in register v[r] ----
    float v[16] = {z4[0], z4[1], z4[2], z4[3], samp[0], samp[1], samp[2], samp[3], samp[4], samp[5], samp[6], samp[7],
                   INFINITY, INFINITY, INFINITY, INFINITY};
    r2o_inlane<1>(v);                                                                     // k = 2
    r2o_inlane<3>(v); r2o_inlane<1>(v);                                                   // k = 4
    r2o_inlane<7>(v); r2o_inlane<2>(v); r2o_inlane<1>(v);                                 // k = 8
    r2o_inlane<15>(v); r2o_inlane<4>(v); r2o_inlane<2>(v); r2o_inlane<1>(v);              // k = 16
    R2O_MIRROR16(X1, u0) r2o_tail16(v);                                                   // k = 32
    R2O_MIRROR16(X3, u1) R2O_XOR16(X1, u0) r2o_tail16(v);                                 // k = 64
    R2O_MIRROR16(X7, u2) R2O_XOR16(X2, u1) R2O_XOR16(X1, u0) r2o_tail16(v);               // k = 128
    R2O_MIRROR16(X15, u3) R2O_XOR16(X4, u2) R2O_XOR16(X2, u1) R2O_XOR16(X1, u0) r2o_tail16(v);   // k = 256
    if (live && l16 < 12) {  // 192 = 12 lanes x 16 sorted depths
        float* out = z_all + ray * (S + NI) + 16 * l16;
#pragma unroll
        for (int r = 0; r < 16; r += 4) *reinterpret_cast<f32x4*>(out + r) = f32x4{v[r], v[r + 1], v[r + 2], v[r + 3]};
    }
}
#undef X1
#undef X2
#undef X3
#undef X4
#undef X7
#undef X8
#undef X15
#undef R2O_TAKE
#undef R2O_MIRROR16
#undef R2O_XOR16
#undef R2O_MIRROR8
#undef R2O_XOR8
#undef R2O_ASM_BANK1
#undef R2O_ASM_BANK_XOR8
#undef R2O_ASM_BANK_MIRROR8
#undef R2O_ASM_MM1
#undef R2O_ASM_MERGE_MIRROR

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
extern "C" int r2l_stratified_z(const float* near, const float* far, int nf_stride, const float* ttab,
                                const float* t_rand, float* z_out, int64_t R, int S, void* stream) {
    if (R <= 0) return 0;
    R2L_REQUIRE(near && far && ttab && z_out && S >= 1, "r2l_stratified_z: NULL pointer or S < 1");
    int64_t blocks = (R * S + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(r2l_stratified_z_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, near, far,
                       nf_stride, ttab, t_rand, z_out, R, S);
    R2L_CHECK(hipGetLastError());
    return 0;
}

extern "C" int r2l_raw2outputs(const float* raw, const float* z, const float* rays_d, const float* noise, int white_bkgd,
                               float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                               int64_t R, int S, void* stream) {
    if (R <= 0) return 0;
    if (S < 1 || S > 64 * MAX_CH) { r2l_set_error("r2l_raw2outputs: S out of range [1,256]", hipErrorInvalidValue); return (int)hipErrorInvalidValue; }
    R2L_REQUIRE(raw && z && rays_d && rgb_map && disp_map && acc_map && depth_map, "r2l_raw2outputs: a required pointer is NULL (only noise and weights are optional)");
    if (S % 64 == 0) {  // the teacher path's sample counts: a quarter wave per ray
#define R2O16_LAUNCH(ROWS_, W_)                                                                                                 \
    hipLaunchKernelGGL((r2l_raw2outputs16_kernel<ROWS_, W_>), dim3((unsigned)((R + 15) / 16)), dim3(256), 0, (hipStream_t)stream, \
                       raw, z, rays_d, noise, white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map, R)
#define R2O16_BOTH(ROWS_) do { if (weights != nullptr) R2O16_LAUNCH(ROWS_, true); else R2O16_LAUNCH(ROWS_, false); } while (0)
        if (S == 64) R2O16_BOTH(4);
        else if (S == 128) R2O16_BOTH(8);
        else if (S == 192) R2O16_BOTH(12);
        else R2O16_BOTH(16);
#undef R2O16_BOTH
#undef R2O16_LAUNCH
        R2L_CHECK(hipGetLastError());
        return 0;
    }
    const int CH = (S + 63) / 64;
#define R2O_LAUNCH(CH_, RPW_)                                                                                                  \
    hipLaunchKernelGGL((r2l_raw2outputs_kernel<CH_, RPW_>), dim3((unsigned)((R + 4 * RPW_ - 1) / (4 * RPW_))), dim3(256), 0,     \
                       (hipStream_t)stream, raw, z, rays_d, noise, white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map, R, S)
    if (CH == 1) R2O_LAUNCH(1, 4);
    else if (CH == 2) R2O_LAUNCH(2, 4);
    else if (CH == 3) R2O_LAUNCH(3, 2);
    else R2O_LAUNCH(4, 2);
#undef R2O_LAUNCH
    R2L_CHECK(hipGetLastError());
    return 0;
}

extern "C" int r2l_sample_pdf_sort(const float* z, const float* weights, const float* u, int64_t u_stride,
                                   float* z_samples, float* z_all, float* z_std, int64_t R, int S, int NI,
                                   void* stream) {
    if (R <= 0) return 0;
    if (S < 3 || S > 64 || NI < 1 || NI > 192 || S + NI > 256) {
        r2l_set_error("r2l_sample_pdf_sort: need 3<=S<=64, 1<=NI<=192, S+NI<=256", hipErrorInvalidValue);
        return (int)hipErrorInvalidValue;
    }
    R2L_REQUIRE(z && weights && u && z_samples && z_all && u_stride >= 0, "r2l_sample_pdf_sort: a required pointer is NULL (only z_std is optional) or u_stride < 0");
    const bool aligned = ((((uintptr_t)z | (uintptr_t)weights | (uintptr_t)u | (uintptr_t)z_samples | (uintptr_t)z_all) & 15) == 0) && (u_stride & 3) == 0;
    if (S == 64 && NI == 128 && aligned)  // the reference's configuration: a quarter wave per ray
        hipLaunchKernelGGL(r2l_sample_pdf_sort16_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, (hipStream_t)stream, z,
                           weights, u, u_stride, z_samples, z_all, z_std, R);
    else
        hipLaunchKernelGGL(r2l_sample_pdf_sort_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z,
                           weights, u, u_stride, z_samples, z_all, z_std, R, S, NI);
    R2L_CHECK(hipGetLastError());
    return 0;
}
