// r2l_dw16.hip — weight gradients of the 2*n_block body layers for the default (fp16) training trio:
//     dW[l][o][i] = sum_r G_l[r][o] * A_l[r][i],   db[l][o] = sum_r G_l[r][o]      (reduction over rays)
//   layer (b,0): G = gt[b] (masked u), A = x_b ;  layer (b,2): G = gx[b+1] (g), A = relu(t_b)
// replacing the weight-gradient half of loss.backward() (/root/reference/main.py:1403-1404) for the R2L body.
//
// Operands: the fp16 `hi` stage pieces the chains stashed (r2l_f2.h): 2 bytes per value instead of 4, ONE fp16 MFMA product
// per fp32 product instead of three.  This GEMM is a reduction over ~10^5 rays of products whose operands are rounded to 11
// bits independently: the rounding errors average out and dW moves by ~5e-5 (relative L2) — the distance between two fp32
// evaluations of this 88-layer net (ReLU-mask flips of near-zero pre-activations), measured in DESIGN.md §2.  The kernel it
// replaces read 17.3 GB per 98 304-ray step at 4.9 TB/s (HBM-bound); this one reads 8.7 GB.
//
// Roofline: HBM.  Every stash byte is read exactly once (a workgroup accumulates the full 256 x 256 tile set for its rays):
// 1 KiB per ray and layer (2 operands x 256 features x 2 B); arithmetic intensity 128 FLOP/B < the chip's ~300.
//
//   * one step = one 32-ray tile: 16 KiB per operand = its 16 stage pieces.  The four waves DMA them straight into LDS
//     (buffer_load ... lds, no VALU, no registers), a quarter each, four stages deep (96 KiB in flight per CU: the kernel's job
//     is to keep HBM busy).  The per-lane SOURCE offset permutes the 16-byte units on the way: the LDS image of a tile is
//         unit(T, rq, g, row, h) = T*128 + rq*16 + g*8 + row*2 + h   <-   stage piece kb = 2T + g, chain lane h*32 + 4rq + row
//     (T: 32-feature tile, rq: ray quad, row: ray in the quad, g / h: 16- / 4-feature sub-blocks), sources in 64-byte runs.
//   * MFMA operands (lane = feature, 8 consecutive rays in its 16 bytes) come out of the image through ds_read_b64_tr_b16.
//     The 32 lanes of a read pass (2 feature blocks g x 4 rays x 4 quads) address one contiguous 256-byte run of the image:
//     bank-conflict free by construction.
//   * wave (wo, wi) owns output features [128 wo, +128) x input features [128 wi, +128) as 4 x 4 MFMA tiles (256 accumulator
//     registers); db rides along on the VALU: the 8 rays of a gradient fragment are summed into one fp32 register per tile
//     (8 x v_fma_mix_f32, exact; waves with wi = 0).
//   * one barrier per tile, in its middle: behind it the next tile's image is published and the current one's buffer is handed
//     to the DMA of the tile four steps ahead.
// Work split, slab partials and the reduce are those of r2l_backward.hip (r2l_dw.h).
#include "r2l_f2.h"
#include "r2l_dw.h"

#define DW16_OP_BYTES 16384
// EXACT (r2l_config.dw_mode = R2L_DW_EXACT): hi AND mid halves of both operands, 48 MFMAs per 16 rays instead of 16:
// mid*hi + hi*mid + hi*hi, small terms first — the three-product scheme of the chains (r2l_f2.h), so dW carries fp32-grade
// products for twice the stash traffic (still HBM-bound).  The ring unit is then a HALF tile (one k-step = 16 rays) of
// [G hi | A hi | G mid | A mid] = 32 KiB, four units deep like the default kernel's four tiles (a DMA instruction moves 16
// rays of two stage pieces either way), one barrier per k-step: the same 96 KiB in flight per CU and the same continuous
// stream, its requests spread evenly over the k-step's MFMAs (round 6).  (First version: whole tiles of 64 KiB, two deep,
// vmcnt(0) per tile: 4.5 TB/s; units requested in a lump behind the barrier: 4.9 - 5.0; one request per 6 MFMAs: 5.1 - 5.2.)
// (A/B: -DDW16_NT streams the stash — read exactly once — past the caches: `nt` on the LDS-DMA loads of the exact kernel)
#ifdef DW16_NT
__device__ __forceinline__ void dw16_dma16_nt(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
#define DW16_DMA dw16_dma16_nt
#else
#define DW16_DMA f3_dma16
#endif
#define DW16_SLOT(q) ((q) & 3)  // ring of four 32 KiB buffers (five = all 160 KiB of LDS measured no gain: profiles/r06_exact_dw_ab.txt)
template <bool EXACT> struct Dw16Cfg {
    static constexpr int NB = 4;
    static constexpr unsigned STAGE_BYTES = 32768u;
};

#define DW16_WAIT_PRO() asm volatile("s_waitcnt vmcnt(24)" ::: "memory")   // the oldest of 4 x 8 requests landed
#define DW16_WAIT_LOOP() asm volatile("s_waitcnt vmcnt(16)" ::: "memory")  // 2 x 8 requests behind the awaited one
typedef short dw16_s16x4 __attribute__((ext_vector_type(4)));
typedef short dw16_s16x8 __attribute__((ext_vector_type(8)));

// 8 rays of this lane's feature: two transposing reads (ray quads rq, rq + 1: 256 bytes apart); `off` is a constant after
// unrolling and lands in the instructions' offset fields
__device__ __forceinline__ f16x8 dw16_frag(unsigned base, unsigned off) {
    typedef __attribute__((address_space(3))) dw16_s16x4 lds_v;
    const dw16_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(size_t)(base + off));
    const dw16_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(size_t)(base + off + 256u));
    return __builtin_bit_cast(f16x8, dw16_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// acc + fp16 half of a packed pair in ONE instruction, exact: v_fma_mix_f32 (the half as a mixed-precision operand times 1.0
// plus acc), as r2l_f2.h's residuals.  (v_dot2c_f32_f16 against (1, 1) would halve the count, but its bf16 sibling was
// measured NOT to return the fp32 sum of the two products on this chip: profiles/r01_summary.md.)
__device__ __forceinline__ float dw16_add_lo(unsigned h, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(acc));
    return r;
}
__device__ __forceinline__ float dw16_add_hi(unsigned h, float acc) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(acc));
    return r;
}

template <bool EXACT>
struct Dw16Frags {  // operands of one k-step (16 rays): four 32-feature tiles of each operand (EXACT: hi and mid halves)
    f16x8 g[4], x[4];
    f16x8 gm[EXACT ? 4 : 1], xm[EXACT ? 4 : 1];
};

template <bool EXACT>
__global__ __launch_bounds__(256, 1) void r2l_dw16_kernel(const R2LDwArgs a, const unsigned* run_unless) {
    constexpr int DW16_NB = Dw16Cfg<EXACT>::NB;
    constexpr unsigned DW16_STAGE_BYTES = Dw16Cfg<EXACT>::STAGE_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char img[DW16_NB][DW16_STAGE_BYTES];
    if (run_unless != nullptr && __builtin_nontemporal_load(run_unless) != 0u) {
        // this step's stash is the bf16x3 trio's (fp32): hand the launch to the kernel behind this one
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.status != nullptr) atomicOr(a.status, 1u);
        return;
    }
    const float unscale = a.scale_dev != nullptr ? a.scale_dev[1] : a.unscale;
    const float wscale = a.act_scale != nullptr ? unscale * a.act_scale[0] : unscale;  // powers of two: exact
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wo = wave >> 1, wi = wave & 1;
    const int64_t total = a.units_per_layer * a.n_layers;
    int64_t u0 = (int64_t)blockIdx.x * a.units_per_wg;
    int64_t u1 = u0 + a.units_per_wg;
    if (u1 > total) u1 = total;
    if (u0 >= u1) return;

    f32x16 acc[4][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};  // db partials: feature 128 wo + 32 eo + (lane & 31), rays of this lane's k half
#pragma unroll
    for (int eo = 0; eo < 4; ++eo)
#pragma unroll
        for (int ei = 0; ei < 4; ++ei)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[eo][ei][c] = 0.f;

    const int64_t Np = R2L_PAD_ROWS(a.N);
    const int64_t slot = R2L_TRIO_SLOT(Np);
    const unsigned img_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&img[0][0];
    // DMA: lane l of a piece fills image unit (rq & 3 = l >> 4, g = (l >> 3) & 1, row = (l >> 1) & 3, h = l & 1) from the
    // stage piece 2T + g, chain lane h*32 + 4 rq + row; this wave moves pieces 4 wave .. 4 wave + 3 of both operands
    const unsigned dvoff = (unsigned)(((lane >> 3) & 1) * 1024 + (lane & 1) * 512 + (lane >> 4) * 64 + ((lane >> 1) & 3) * 16);
    const unsigned wsrc = (unsigned)wave * 4096u;   // pieces 4w..: source offset (2w)*2048
    const unsigned wdst = img_lds + (unsigned)wave * 4096u;
    // transposing reads: lane = (kg = rays 8kg.., g = 16-feature block, row, q): see the header
    unsigned rl;
    {
        const int kg = lane >> 5, g = (lane >> 4) & 1, L = lane & 15, row = L >> 2, q = L & 3;
        rl = (unsigned)(kg * 512 + g * 128 + row * 32 + (q & 1) * 16 + (q >> 1) * 8);
    }
    const unsigned gl = img_lds + rl + (unsigned)wo * 8192u;
    const unsigned al = img_lds + DW16_OP_BYTES + rl + (unsigned)wi * 8192u;

    int64_t u = u0;
    const int first_layer = a.layer0 + (int)(u0 / a.units_per_layer);
    while (u < u1) {
        const int layer = a.layer0 + (int)(u / a.units_per_layer);
        const int64_t cu = u % a.units_per_layer;
        int64_t cend = cu + (u1 - u);
        if (cend > a.units_per_layer) cend = a.units_per_layer;
        const int b = layer >> 1;
        const float* G = (layer & 1) ? a.gx + (int64_t)(b + 1) * slot : a.gt + (int64_t)b * slot;
        const float* A = (layer & 1) ? a.save_t + (int64_t)b * slot : a.save_x + (int64_t)b * slot;
        const int64_t r0 = cu * DW_CHUNK;  // a multiple of 64 rays: two whole tiles
        int64_t r1 = cend * DW_CHUNK;
        if (r1 > Np) r1 = Np;
        const int ntiles = (int)((r1 - r0) / R2L_TILE_RAYS);
        // descriptors based at the first tile of the segment (16 KiB of fp16 stage pieces per tile and operand)
        const unsigned long long ga = (unsigned long long)G + (unsigned long long)(r0 / R2L_TILE_RAYS) * DW16_OP_BYTES;
        const unsigned long long aa = (unsigned long long)A + (unsigned long long)(r0 / R2L_TILE_RAYS) * DW16_OP_BYTES;
        const u32x4 grs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        const u32x4 ars = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)aa),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(aa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        if constexpr (EXACT) {
            // unit q = (tile q >> 1, k-step q & 1) -> ring buffer q & 3: [G hi | A hi | G mid | A mid], 8 KiB each (8 feature
            // tiles T x 1 KiB: T*64 + rq*16 + g*8 + row*2 + h in 16-byte units, rq = ray quad 0..3); 8 loads per wave and unit.
            // Units past the end are clamped to the last one (harmless reloads: the vmcnt waits stay uniform).
            const int nunits = 2 * ntiles;
            auto issue = [&](int q) {
                const int qc = q < nunits ? q : nunits - 1;
                const unsigned so = (unsigned)(qc >> 1) * (unsigned)DW16_OP_BYTES + wsrc + (unsigned)(qc & 1) * 256u;
                const unsigned ld = img_lds + (unsigned)DW16_SLOT(q) * DW16_STAGE_BYTES + (unsigned)wave * 2048u;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {  // the wave's two piece pairs (pieces 4w + 2pr, 4w + 2pr + 1), 16 rays of each
                    const unsigned po = so + (unsigned)pr * 2048u, lp = ld + (unsigned)pr * 1024u;
                    DW16_DMA(grs, dvoff, po, lp);
                    DW16_DMA(ars, dvoff, po, lp + 8192u);
                    DW16_DMA(grs, dvoff, po + a.mid_off, lp + 16384u);
                    DW16_DMA(ars, dvoff, po + a.mid_off, lp + 24576u);
                }
            };
            const unsigned gq = img_lds + rl + (unsigned)wo * 4096u, aq = img_lds + 8192u + rl + (unsigned)wi * 4096u;
            auto read = [&](Dw16Frags<EXACT>& R, int q) {
                const unsigned bo = (unsigned)DW16_SLOT(q) * DW16_STAGE_BYTES;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    R.g[e] = dw16_frag(gq + bo, (unsigned)e * 1024u);
                    R.x[e] = dw16_frag(aq + bo, (unsigned)e * 1024u);
                    R.gm[e] = dw16_frag(gq + bo, 16384u + (unsigned)e * 1024u);
                    R.xm[e] = dw16_frag(aq + bo, 16384u + (unsigned)e * 1024u);
                }
            };
            Dw16Frags<EXACT> R0, R1;
#pragma unroll
            for (int q = 0; q < DW16_NB; ++q) issue(q);
            DW16_WAIT_PRO();  // unit 0 landed ((NB - 1) x 8 loads behind it)
            __syncthreads();
            read(R0, 0);
            // Steady state: the 8 requests of unit q + NB go out ONE AT A TIME, behind every 6th of the k-step's 48 MFMAs (round 6:
            // in a lump behind the barrier the kernel took 3.52 ms per 98 304 rays, two per 12 MFMAs 3.42, one per 6 3.38; requesting
            // the two 256-byte halves of a tile's 512-byte runs back to back instead: +8 %.  profiles/r06_exact_dw_ab.txt).  Unit q+1
            // landed (the 16 requests of units q+2, q+3 behind it); behind the barrier everybody's share is visible and nobody reads
            // unit q's buffer any more (its fragments are in registers): it goes to the DMA of unit q+NB.  MFMAs: small terms first.
            auto issue_one = [&](int q, int k) {  // request k of 8: k = pr * 4 + {G hi, A hi, G mid, A mid}
                const int qc = q < nunits ? q : nunits - 1;
                const unsigned so = (unsigned)(qc >> 1) * (unsigned)DW16_OP_BYTES + wsrc + (unsigned)(qc & 1) * 256u;
                const unsigned ld = img_lds + (unsigned)DW16_SLOT(q) * DW16_STAGE_BYTES + (unsigned)wave * 2048u;
                const int pr = k >> 2, w = k & 3;
                const unsigned po = so + (unsigned)pr * 2048u + ((w & 2) ? a.mid_off : 0u);
                const unsigned lp = ld + (unsigned)pr * 1024u + (unsigned)w * 8192u;
                if (w & 1) DW16_DMA(ars, dvoff, po, lp);
                else DW16_DMA(grs, dvoff, po, lp);
            };
            auto mma_issue = [&](const Dw16Frags<EXACT>& R, int qn) {
                issue_one(qn, 0);
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    const int grp = i >> 4, eo = (i >> 2) & 3, ei = i & 3;
                    if (grp == 0) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.gm[eo], R.x[ei], acc[eo][ei], 0, 0, 0);
                    else if (grp == 1) acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.g[eo], R.xm[ei], acc[eo][ei], 0, 0, 0);
                    else acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.g[eo], R.x[ei], acc[eo][ei], 0, 0, 0);
                    if ((i + 1) % 6 == 0 && i + 1 < 48) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_one(qn, (i + 1) / 6);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (wi == 0) {  // db: hi and mid halves of the 8 rays of a gradient fragment, added up exactly in fp32
#pragma unroll
                    for (int eo = 0; eo < 4; ++eo) {
                        const u32x4 w = __builtin_bit_cast(u32x4, R.g[eo]), wm = __builtin_bit_cast(u32x4, R.gm[eo]);
#pragma unroll
                        for (int d = 0; d < 4; ++d) bsum[eo] = dw16_add_hi(w[d], dw16_add_lo(w[d], bsum[eo]));
#pragma unroll
                        for (int d = 0; d < 4; ++d) bsum[eo] = dw16_add_hi(wm[d], dw16_add_lo(wm[d], bsum[eo]));
                    }
                }
            };
            for (int q = 0; q < nunits; q += 2) {
                DW16_WAIT_LOOP();
                __syncthreads();
                read(R1, q + 1);
                mma_issue(R0, q + DW16_NB);
                DW16_WAIT_LOOP();
                __syncthreads();
                read(R0, q + 2);
                mma_issue(R1, q + 1 + DW16_NB);
            }
        } else {
            // tile s of the segment -> ring buffer s & 3: 8 pieces per wave.  Tiles past the end are clamped to the last one
            // (harmless reloads into a dead buffer: every step issues exactly 8 loads, which keeps the vmcnt waits uniform)
            auto issue = [&](int s) {
                const int sc = s < ntiles ? s : ntiles - 1;
                const unsigned so = (unsigned)sc * (unsigned)DW16_OP_BYTES + wsrc;
                const unsigned ld = wdst + (unsigned)DW16_SLOT(s) * DW16_STAGE_BYTES;
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned po = (unsigned)((i >> 1) * 2048 + (i & 1) * 256);
                    f3_dma16(grs, dvoff, so + po, ld + (unsigned)i * 1024u);
                    f3_dma16(ars, dvoff, so + po, ld + DW16_OP_BYTES + (unsigned)i * 1024u);
                    if (EXACT) {
                        f3_dma16(grs, dvoff, so + po + a.mid_off, ld + 2 * DW16_OP_BYTES + (unsigned)i * 1024u);
                        f3_dma16(ars, dvoff, so + po + a.mid_off, ld + 3 * DW16_OP_BYTES + (unsigned)i * 1024u);
                    }
                }
            };
            auto read = [&](Dw16Frags<EXACT>& R, int s, int ks) {
                const unsigned bo = (unsigned)DW16_SLOT(s) * DW16_STAGE_BYTES + (unsigned)ks * 1024u;
                const unsigned gp = gl + bo, ap = al + bo;
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    R.g[e] = dw16_frag(gp, (unsigned)e * 2048u);
                    R.x[e] = dw16_frag(ap, (unsigned)e * 2048u);
                    if (EXACT) {
                        R.gm[e] = dw16_frag(gp, 2 * DW16_OP_BYTES + (unsigned)e * 2048u);
                        R.xm[e] = dw16_frag(ap, 2 * DW16_OP_BYTES + (unsigned)e * 2048u);
                    }
                }
            };
            auto mma = [&](const Dw16Frags<EXACT>& R) {
                if (EXACT) {  // small terms first
    #pragma unroll
                    for (int eo = 0; eo < 4; ++eo)
    #pragma unroll
                        for (int ei = 0; ei < 4; ++ei)
                            acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.gm[eo], R.x[ei], acc[eo][ei], 0, 0, 0);
    #pragma unroll
                    for (int eo = 0; eo < 4; ++eo)
    #pragma unroll
                        for (int ei = 0; ei < 4; ++ei)
                            acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.g[eo], R.xm[ei], acc[eo][ei], 0, 0, 0);
                }
    #pragma unroll
                for (int eo = 0; eo < 4; ++eo)
    #pragma unroll
                    for (int ei = 0; ei < 4; ++ei)
                        acc[eo][ei] = __builtin_amdgcn_mfma_f32_32x32x16_f16(R.g[eo], R.x[ei], acc[eo][ei], 0, 0, 0);
                if (wi == 0) {  // db: the 8 rays of a gradient fragment added up on the VALU, exactly, in fp32
    #pragma unroll
                    for (int eo = 0; eo < 4; ++eo) {
                        const u32x4 w = __builtin_bit_cast(u32x4, R.g[eo]);
    #pragma unroll
                        for (int d = 0; d < 4; ++d) bsum[eo] = dw16_add_hi(w[d], dw16_add_lo(w[d], bsum[eo]));
                        if (EXACT) {
                            const u32x4 wm = __builtin_bit_cast(u32x4, R.gm[eo]);
    #pragma unroll
                            for (int d = 0; d < 4; ++d) bsum[eo] = dw16_add_hi(wm[d], dw16_add_lo(wm[d], bsum[eo]));
                        }
                    }
                }
            };
            Dw16Frags<EXACT> R0, R1;
            // prologue: tiles 0 .. NB-1 requested, tile 0 published (latency exposed once per segment)
    #pragma unroll
            for (int s = 0; s < DW16_NB; ++s) issue(s);
            DW16_WAIT_PRO();  // tile 0 landed ((NB - 1) x 8 loads behind it)
            __syncthreads();
            read(R0, 0, 0);
            for (int s = 0; s < ntiles; ++s) {
                read(R1, s, 1);
                mma(R0);
                // tile s+1 landed (its 8 loads have the 16 of tiles s+2, s+3 behind them; EXACT: nothing behind its 16); behind the
                // barrier everybody's share is visible and nobody reads the image of tile s any more (its fragments are in registers)
                DW16_WAIT_LOOP();
                __syncthreads();
                // (the 8 requests in a lump: spreading them over the second k-step's MFMAs, which pays in the EXACT kernel, measured 0 here)
                issue(s + DW16_NB);
                read(R0, s + 1, 0);
                mma(R1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the images are dead (and the clamped reloads landed) before the next segment refills them

        // flush: D[m][n] of tile (eo, ei): m = 8 (c>>2) + 4 hh + (c&3) -> output feature wo*128 + 32 eo + m, n = lane & 31 ->
        // input feature wi*128 + 32 ei + n
        {
            const int n = lane & 31, hh = lane >> 5;
            float* sl = (a.slab != nullptr) ? a.slab + ((int64_t)blockIdx.x * 2 + (layer - first_layer)) * DW_SLAB_FLOATS : nullptr;
            float* gw = a.grads + b_off_body_w(layer);
            float* gbias = a.grads + b_off_body_b(layer);
            float* rowp = (sl != nullptr ? sl : gw) + (wo * 128 + 4 * hh) * R2L_W + wi * 128 + n;
            if (sl != nullptr) {
#pragma unroll
                for (int eo = 0; eo < 4; ++eo)
#pragma unroll
                    for (int ei = 0; ei < 4; ++ei)
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            rowp[(32 * eo + 8 * (c >> 2) + (c & 3)) * R2L_W + 32 * ei] = acc[eo][ei][c] * wscale;
                            acc[eo][ei][c] = 0.f;
                        }
            } else {
#pragma unroll
                for (int eo = 0; eo < 4; ++eo)
#pragma unroll
                    for (int ei = 0; ei < 4; ++ei)
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            atomicAdd(rowp + (32 * eo + 8 * (c >> 2) + (c & 3)) * R2L_W + 32 * ei, acc[eo][ei][c] * wscale);
                            acc[eo][ei][c] = 0.f;
                        }
            }
            if (wi == 0) {  // db: the two k halves of a feature sit in lanes n and n + 32
#pragma unroll
                for (int eo = 0; eo < 4; ++eo) {
                    const float v = (bsum[eo] + __shfl_xor(bsum[eo], 32)) * unscale;
                    if (hh == 0) {
                        if (sl != nullptr) sl[R2L_W * R2L_W + wo * 128 + 32 * eo + n] = v;
                        else atomicAdd(gbias + wo * 128 + 32 * eo + n, v);
                    }
                    bsum[eo] = 0.f;
                }
            }
        }
        u += cend - cu;
    }
}

int r2l_dw16_launch(const R2LDwArgs& a, int64_t wgs, const unsigned* run_unless, hipStream_t stream) {
    if (a.mid_off != 0u) hipLaunchKernelGGL(r2l_dw16_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, stream, a, run_unless);
    else hipLaunchKernelGGL(r2l_dw16_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, stream, a, run_unless);
    R2L_CHECK(hipGetLastError());
    return 0;
}
