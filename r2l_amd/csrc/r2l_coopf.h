// r2l_coopf.h — machinery of the COOPERATIVE fp16x2 chains (r2l_coopf_fwd.hip, r2l_coopf_bwd.hip): the default fp16 trio for
// launches too small to fill the chip with one wave per 32-ray tile.
//
// The one-wave-per-tile kernels (r2l_fwd2 / r2l_bwd2) need 32 768 rays to occupy the 1024 SIMDs, and every workgroup streams
// the whole 25 MB of packed weights for its 128 rays.  Here the FOUR waves of a workgroup share ONE 32-ray tile: wave w owns
// output tiles 2w, 2w+1 (64 of the 256 features) of every layer, so 4096 rays already give 128 workgroups x 4 waves and a
// step of <= 8192 rays is one round.  From 8192 rays on a workgroup takes NT = 2 ray tiles (64 rays) that share every
// weight load: half the L2 stream per ray.  Everything else is the fp16 trio's: the same packed stage streams (16 KiB stages,
// [split][tile][lane][8 fp16]; a wave reads the four 1 KiB pieces of its two tiles), the same 32x32x16 MFMA fragments, three
// fp16 products per fp32 product, the same fp16 stage-piece stash, mask words and range-guard protocol — so r2l_dw16 /
// r2l_dw_head16 and the bf16x3 fallbacks serve these launches unchanged.
//   * B operands (the layer input, all 256 features x 32 rays) are shared through LDS: wave w's output fragments of tiles
//     2w, 2w+1 ARE, lane for lane, the B operands of stages 4w .. 4w+3 of the next layer (fragment registers c = 8r .. 8r+7 of
//     tile T = stage 2T + r), so each wave converts its own 32 values per lane to (hi, mid), writes them to the image
//     [stage][split][lane][16 B] and — training — stores the hi quad as the stash piece; ONE barrier per layer.
//   * A operands come straight from L2 into a register ring of R = 4 stages (16 loads in flight per wave; template
//     parameter), no LDS: a wave reads only its own quarter of a stage.  The ring slot of every stage is a
//     compile-time constant: a layer has 17 stages = 1 mod R, so the phase advances by one per layer and the body is
//     unrolled over R / 2 blocks.
//   * The kernel is bound by that weight stream (256 KiB per layer and workgroup from L2), not by the matrix pipe.
#pragma once
#include "r2l_f2.h"
#include <stdio.h>

// register-ring depth (stages of A operands in flight per wave), per tiles-per-workgroup.  A one-tile workgroup has the CU
// to itself and registers to spare, but a ring of 8 (32 loads in flight per wave, no spills) measured 0.264 ms against 0.259
// for the 4096-ray forward: the stream is bound by what a CU pulls from L2 (~46 B/clk), not by latency
#ifndef FC_RING_ONE
#define FC_RING_ONE 4
#endif
#define FC_RING_TWO 4
template <int NT> struct FcRingOf { static constexpr int value = NT == 1 ? FC_RING_ONE : FC_RING_TWO; };
#define FC_BOP_BYTES 32768   // one B-operand image: 16 stages x (hi, mid) x 1 KiB
// ONE one-tile workgroup per CU, as a guarantee.  Round 2 found that with two workgroups of the one-tile training forward on
// a CU (two waves per SIMD) single rays came back with a wrong red channel; round 3 pinned it down (DESIGN.md §2,
// tools/coopf_forensics.py, profiles/r03_coresidency.md): exactly ONE term of the tail's 32-term dot product is missing, always
// in the low lane of a `v_pk_fma_f32 ... op_sel:[0,1,0]` (low lane fed by src1's HIGH dword), always in lanes 48 - 63, with
// bit-exact inputs — independent of what produced the operands and when (s_nop / reordered v_movs / vmcnt(0) in front of it do
// not help), never in the op_sel_hi-only form of the same FMA, never with one wave per SIMD; armed by a co-resident wave that
// issues MFMAs (tools/pk_opsel_mfma_probe.hip reproduces it outside the library).  The tails are now written so
// that hipcc cannot form that instruction (r2l_no_pack), the build refuses any packed-fp32 op with a low-lane src1 / src2
// op_sel (r2l_amd/build.py: ISA audit), and — belt and braces, because the trigger is a property of the silicon that only
// the absence of a second wave is known to avoid — every one-tile launch still carries FC_SOLO_LDS_BYTES of dynamic LDS
// (64 - 72 KiB static + 24 KiB > half of the CU's 160 KiB), checked per kernel with hipOccupancyMaxActiveBlocksPerMultiprocessor
// before its first launch (fc_check_solo).  It costs nothing: these launches have at most one tile per CU by construction
// (r2l_coopf_two_tiles), and two-tile workgroups (143 KiB) cannot share a CU at all.  Only the reproducer builds
// (tools/build_variant.sh ... -DFC_ALLOW_SHARE_CU) drop the padding.
#ifdef FC_ALLOW_SHARE_CU
#define FC_SOLO_LDS_BYTES 0
#else
#define FC_SOLO_LDS_BYTES 24576
#endif
// the launch would put at most one such workgroup on a CU?  (asked once per kernel; error text in r2l_last_error)
template <class K>
static inline int fc_check_solo(K kernel, const char* name, int* cached) {
#ifndef FC_ALLOW_SHARE_CU
    if (*cached == 0) {
        int nb = 0;
        R2L_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, (size_t)FC_SOLO_LDS_BYTES));
        *cached = nb == 1 ? 1 : -nb - 1;
    }
    if (*cached != 1) {
        char msg[192];
        snprintf(msg, sizeof(msg), "%s: %d workgroups per CU possible, the one-tile cooperative kernels require exactly 1 "
                 "(r2l_coopf.h FC_SOLO_LDS_BYTES)", name, -*cached - 1);
        r2l_set_error_msg(msg);
        return (int)hipErrorLaunchFailure;
    }
#endif
    return 0;
}
typedef __attribute__((address_space(3))) u32x4 fc_lds_u32x4;
__device__ __forceinline__ u32x4 fc_lds_read(unsigned addr) { return *(fc_lds_u32x4*)(size_t)addr; }
__device__ __forceinline__ void fc_lds_write(unsigned addr, u32x4 v) { *(fc_lds_u32x4*)(size_t)addr = v; }

template <int R>
struct FcRing {  // R stages x (hi tile 0, hi tile 1, mid tile 0, mid tile 1) of this wave
    u32x4 a[R][4];
};
struct FcStream {
    u32x4 rs;       // descriptor of the stage stream
    unsigned voff;  // lane*16 + wave*2048: tile 2w of split 0; + 1024: tile 2w+1; + 8192: split 1
    unsigned g;     // next stage to LOAD
#ifdef FC_LATE_MODE  // diagnostic builds (tools/coopf_coresidency.py): the late workgroup of a CU runs a thinned-out body
    bool late;
#endif
};

template <int IMM>
__device__ __forceinline__ void fc_load(u32x4& dst, u32x4 rs, unsigned voff, unsigned soff) {
    // untracked by the compiler (it would drain vmcnt at every barrier and loop header): consumers wait with fc_wait
    // (no "memory" clobber: the stream is read-only, and LDS reads of the B operands may be scheduled across it)
    // (cache policy: default.  Measured on the 4096-ray forward: `nt` 0.40 ms against 0.25 — the stream every workgroup
    // shares stops being kept in L2 —, sc0 / sc1 / sc0 sc1: no change)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(dst) : "v"(voff), "s"(rs), "s"(soff), "n"(IMM));
}
__device__ __forceinline__ void fc_issue(u32x4 (&a)[4], FcStream& p) {
#if defined(FC_LATE_MODE) && FC_LATE_MODE == 2  // the late workgroup loads no weights (one stage over and over: L1 hits)
    const unsigned so = p.late ? 0u : p.g * (unsigned)F2_STAGE_BYTES;
#else
    const unsigned so = p.g * (unsigned)F2_STAGE_BYTES;
#endif
    fc_load<0>(a[0], p.rs, p.voff, so);
    fc_load<1024>(a[1], p.rs, p.voff, so);
    fc_load<0>(a[2], p.rs, p.voff + 8192u, so);
    fc_load<1024>(a[3], p.rs, p.voff + 8192u, so);
    ++p.g;
}
// the four loads of the oldest stage in flight have landed: the R - 1 younger stages (4 loads each) may still fly; vmcnt
// retires in order, and anything else in the queue (stash stores) only makes the wait stricter
template <int R>
__device__ __forceinline__ void fc_wait(u32x4 (&a)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(4 * (R - 1)));
}
__device__ __forceinline__ void fc_barrier() {  // LDS writes of this wave done, then the workgroup barrier (no vmcnt drain)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// stash / ride-along global stores the compiler does not track either.  (The hazard recognizer does not look into inline
// asm: a VALU write to the data registers of a > 64-bit store needs wait states behind it, hence the s_nop.)
__device__ __forceinline__ void fc_store_nt(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void fc_store_b32(void* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}

// acc[ray tile][2 feature tiles] (+)= stage: BIAS: one MFMA per tile (bias hi / mid in k slots 0, 1 against ones), else the three
// products, small terms first; the NT ray tiles of the workgroup share the A operands.  Then the slot is refilled with the
// stage R positions ahead.
template <int SLOT, bool BIAS, bool ZERO, int NT, int R>
__device__ __forceinline__ void fc_stage(f32x16 (&acc)[NT][2], FcRing<R>& W, FcStream& p, const f16x8 (&bh)[NT], const f16x8 (&bm)[NT]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    fc_wait<R>(W.a[SLOT]);
    const f16x8 h0 = __builtin_bit_cast(f16x8, W.a[SLOT][0]), h1 = __builtin_bit_cast(f16x8, W.a[SLOT][1]);
    const f16x8 m0 = __builtin_bit_cast(f16x8, W.a[SLOT][2]), m1 = __builtin_bit_cast(f16x8, W.a[SLOT][3]);
#if defined(FC_LATE_MODE) && FC_LATE_MODE == 1  // the late workgroup issues no MFMAs
    if (p.late) {
        fc_issue(W.a[SLOT], p);
        return;
    }
#endif
    if (BIAS) {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, bh[rt], ZERO ? zero : acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, bh[rt], ZERO ? zero : acc[rt][1], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m0, bh[rt], ZERO ? zero : acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(m1, bh[rt], ZERO ? zero : acc[rt][1], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, bm[rt], acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, bm[rt], acc[rt][1], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, bh[rt], acc[rt][0], 0, 0, 0);
            acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, bh[rt], acc[rt][1], 0, 0, 0);
        }
    }
    fc_issue(W.a[SLOT], p);
}

// One layer: its bias (or zero) stage in ring slot PH, then the 16 k-stages against the B-operand images at `bop` + rt *
// FC_BOP_BYTES (LDS byte address of this lane's 16 bytes of stage 0, split 0, ray tile 0).  ZERO_FIRST: the bias stage
// initialises acc (C = 0).  B operands are read one stage ahead.
template <int PH, int KB, int NT, int R>
__device__ __forceinline__ void fc_layer_k(f32x16 (&acc)[NT][2], FcRing<R>& W, FcStream& p, unsigned bop, u32x4 (&nh)[NT], u32x4 (&nm)[NT]) {
    if constexpr (KB < 16) {
        f16x8 bh[NT], bm[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            bh[rt] = __builtin_bit_cast(f16x8, nh[rt]);
            bm[rt] = __builtin_bit_cast(f16x8, nm[rt]);
            if (KB < 15) {
                nh[rt] = fc_lds_read(bop + (unsigned)rt * FC_BOP_BYTES + (unsigned)(KB + 1) * 2048u);
                nm[rt] = fc_lds_read(bop + (unsigned)rt * FC_BOP_BYTES + (unsigned)(KB + 1) * 2048u + 1024u);
            }
        }
        fc_stage<(PH + 1 + KB) % R, false, false>(acc, W, p, bh, bm);
        fc_layer_k<PH, KB + 1>(acc, W, p, bop, nh, nm);
    }
}
template <int PH, bool ZERO_FIRST, int NT, int R>
__device__ __forceinline__ void fc_layer(f32x16 (&acc)[NT][2], FcRing<R>& W, FcStream& p, unsigned bop, const f16x8& ones) {
    u32x4 nh[NT], nm[NT];
    f16x8 o1[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        nh[rt] = fc_lds_read(bop + (unsigned)rt * FC_BOP_BYTES);
        nm[rt] = fc_lds_read(bop + (unsigned)rt * FC_BOP_BYTES + 1024u);
        o1[rt] = ones;
    }
    fc_stage<PH, true, ZERO_FIRST>(acc, W, p, o1, o1);
    fc_layer_k<PH, 0>(acc, W, p, bop, nh, nm);
}

// Eight values of one stage -> (hi, mid) fp16 quads (r2l_f2.h's split: hi = fp16(x), mid = fp16(x - hi)); amax tracks the
// largest |value| for the range guard
__device__ __forceinline__ void fc_split8(const float (&v)[8], u32x4& uh, u32x4& um, float& amax) {
    typedef F2Side<false, F3None> S;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[2 * k])), __builtin_fabsf(v[2 * k + 1]));
        uh[k] = S::pk(v[2 * k], v[2 * k + 1]);
        um[k] = S::pk(f2_res_lo(uh[k], v[2 * k]), f2_res_hi(uh[k], v[2 * k + 1]));
    }
}

// This wave's output fragments (tiles 2w, 2w+1) become the B operands of stages 4w .. 4w+3 of the next GEMM: convert, write
// to the image at `bopw` (LDS byte address of this lane's 16 bytes of stage 4w, split 0), stash the hi quads (training) at
// hst[64 * i], i = 0..3.  RELU: values are max(x, 0) (and, MASK, bit tt*16 + c of *mword = [frag[tt][c] > 0]).
struct FcIdentity {
    __device__ __forceinline__ float operator()(float v, int, int) const { return v; }
};
// (MID: the mid quads are stashed as well, mid_units 16-byte units behind the hi pieces — exact weight gradients; a template
// parameter so that the default kernels keep their register allocation)
template <bool RELU, bool SAVE, bool MASK, class Sel = FcIdentity, bool MID = false>
__device__ __forceinline__ void fc_produce(const f32x16 (&frag)[2], unsigned bopw, u32x4* hst, unsigned* mword, float& amax,
                                           const Sel& sel = Sel(), int64_t mid_units = 0) {
    unsigned mw = 0u;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float x = sel(frag[tt][8 * r + s], tt, 8 * r + s);
                v[s] = RELU ? fmaxf(x, 0.f) : x;
                if (MASK) mw |= (x > 0.f ? 1u : 0u) << (tt * 16 + 8 * r + s);
            }
            u32x4 uh, um;
            fc_split8(v, uh, um, amax);
            fc_lds_write(bopw + (unsigned)(2 * tt + r) * 2048u, uh);
            fc_lds_write(bopw + (unsigned)(2 * tt + r) * 2048u + 1024u, um);
            if (SAVE) {
                fc_store_nt(hst + 64 * (2 * tt + r), uh);
                if (MID) fc_store_nt(hst + 64 * (2 * tt + r) + mid_units, um);
            }
        }
    if (MASK) *mword = mw;
}

// ray tiles per workgroup: 1 while that keeps the launch within one workgroup per CU, else 2.  A third policy, 3 = the MIXED
// grid for n_cu < tiles < 2 n_cu (tiles - n_cu two-tile workgroups and 2 n_cu - tiles one-tile ones = one workgroup on every CU,
// where ceil(tiles / 2) two-tile workgroups leave CUs idle; r2l_coopf_fwd.hip), is OPT-IN: r2l_config.coop_tiles = 3, else
// R2L_COOPF_TILES=3 (outside its band: 1 below, 2 above).  Built and measured in round 6 as VERDICT r5 #2 asked: bit-identical
// to both plain forms, and SLOWER than the two-tile launch at 12 288 rays (forward 461 vs 438 us, dX chain 412 vs 375 us with
// the roles dealt evenly over the XCDs; 497 / 443 us with the two-tile roles filling whole XCDs) — these chains are bound by the
// weight stream each XCD's L2 serves its CUs (~1.2 - 1.6 TB/s per XCD at 24 - 32 streaming workgroups), so putting the idle
// CUs to work on one more stream each slows every workgroup down by more than the shorter two-tile queue gains
// (profiles/r06_mixed_coopf_ab.txt).  AUTO therefore keeps the round-5 policy.
static inline int r2l_coopf_n_cu() {
    static int n_cu = 0;  // one device type per process
    if (n_cu == 0) {
        int dev = 0, v = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return n_cu;
}
static inline int r2l_coopf_policy(int64_t tiles) {
    int want = g_r2l_cfg.coop_tiles;
    if (want == 0) {
        if (const char* e = getenv("R2L_COOPF_TILES")) {
            if (e[0] >= '1' && e[0] <= '3') want = e[0] - '0';
        }
    }
    if (want == 1 || want == 2) return want;
    const int n_cu = r2l_coopf_n_cu();
    if (tiles <= n_cu) return 1;
    if (tiles >= 2 * (int64_t)n_cu) return 2;
#ifndef FC_MIXED_AUTO  // (A/B builds with -DFC_MIXED_AUTO: AUTO takes the mixed grid in its band)
    if (want == 0) return 2;
#endif
    return 3;
}
static inline bool r2l_coopf_two_tiles(int64_t tiles) { return r2l_coopf_policy(tiles) == 2; }
// two-tile workgroups of a MIXED launch (its grid has tiles - this many workgroups), or 0: not a mixed launch
static inline int r2l_coopf_mixed_two(int64_t tiles) { return r2l_coopf_policy(tiles) == 3 ? (int)(tiles - r2l_coopf_n_cu()) : 0; }

// Position of this workgroup in the role order of a MIXED launch.  The two roles run at different paces (one tile: ~3.1 us per
// layer, two tiles: ~4.5 us), and the workgroups of one XCD share the weight stream through that XCD's 4 MiB L2 only while they
// stay within a few blocks (0.57 MB each) of each other: with the roles interleaved over the XCDs (position = blockIdx) the
// one-tile workgroups run ahead.  xcd_major (A/B knob R2L_MIXED_MAP=1): position = (blockIdx % 8) * (grid / 8) + blockIdx / 8 —
// workgroups are dealt to the 8 XCDs round-robin, so the two-tile roles fill whole XCDs and the one-tile roles the others (at
// most one XCD holds both).  Measured: WORSE (forward 497 vs 461 us at 12 288 rays) — 32 two-tile streams on one L2 are slower
// than 16 + 16; both are slower than the plain two-tile launch (438 us): profiles/r06_mixed_coopf_ab.txt.  Default: by blockIdx.
__device__ __forceinline__ int fc_mixed_index(int xcd_major) {
    const int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int g = (int)gridDim.x;
    if (!xcd_major || (g & 7) != 0) return b;
    return (b & 7) * (g >> 3) + (b >> 3);
}
static inline int r2l_coopf_mixed_xcd_major() {
    const char* e = getenv("R2L_MIXED_MAP");  // A/B knob: 0 (default) = roles by blockIdx, 1 = XCD-major
    return (e && e[0] == '1') ? 1 : 0;
}

// launchers (called from r2l_fwd2_forward / r2l_bwd2_backward when the launch is small: r2l_use_coopf)
int r2l_coopf_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab, const float* c2w_host12,
                      int H, int W, float focal, const float* wstream2, const float* params, int n_block, float* rgb,
                      float* save_x, float* save_t, int64_t N, hipStream_t stream);
int r2l_coopf_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                       const float* wstream_bwd2, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                       float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale, unsigned* status,
                       const float* scale_dev, int b_start = -1, int b_end = 0);  // blocks b_start (-1: the last) down to b_end
