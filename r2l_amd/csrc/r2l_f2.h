// r2l_f2.h — the stage machinery of r2l_f3.h for TWO fp16 splits instead of three bf16 ones (r2l_fwd2.hip).
//
// An fp32 value x is written hi + mid with hi = fp16(x), mid = fp16(x - hi): 11 + 11 mantissa bits, i.e. x to ~2^-22
// relative as long as both parts stay inside fp16's exponent range (|x| < 65504; parts below 2^-14 lose bits but are then
// below 2^-25 absolutely).  A product a*b is taken as the three fp16 products mid*hi + hi*mid + hi*hi (fp32 accumulate in
// the MFMA): relative error ~2^-21 instead of the ~2^-24 of the six-product bf16 scheme, for HALF the matrix work and two
// thirds of the operand bytes.  Forward-only (render / evaluation): 1e-4 on RGB is the parity bar, measured ~1e-6.
// Range guard: every lane tracks the largest |B value| it converts; a kernel that saw one near the end of fp16's range
// raises a status word in device memory and the caller reruns the launch on the bf16x3 kernel (r2l_fwd2.hip).
//
// Everything else is r2l_f3.h's design: 16 KiB stages [split][tile][lane][8 fp16] DMA'd into LDS by the four waves (a
// quarter each = 4 pieces), six buffers, the barrier that publishes stage k+1 in the middle of stage k, side work (A-operand
// LDS reads, DMA pieces, gather + split of the next stage's B values) hand-interleaved with the MFMA groups.
#pragma once
#include "r2l_f3.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define F2_STAGE_BYTES 16384  // 2 splits x 8 tiles x 64 lanes x 16 B
#ifndef F2_NBUF
#define F2_NBUF 6
#endif

// x - fp16 half of a packed pair in ONE instruction: v_fma_mix_f32 (the half as a mixed-precision operand times -1.0 plus
// x; exact).  The compiler's own form is v_cvt_f32_f16 + v_sub_f32 (checked bit-identical on the device: tools/_bin/mix_probe)
__device__ __forceinline__ float f2_res_lo(unsigned h, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
__device__ __forceinline__ float f2_res_hi(unsigned h, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}

// LDS-DMA pieces of one request share ONE M0 write: the instruction's immediate offset moves the memory address AND the LDS
// address (measured: tools/dma_probe.hip), and consecutive pieces advance both by 1 KiB.  Nothing else in these kernels
// writes M0 between the pieces of a request (LDS instructions do not use it on gfx9+; no dynamic register indexing).
__device__ __forceinline__ void f2_dma_base(unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds_addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void f2_dma_piece(u32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" : : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
}
__device__ __forceinline__ void f2_dma_piece_i(int i, u32x4 rsrc, unsigned voff, unsigned soff) {
    if (i == 0) f2_dma_piece<0>(rsrc, voff, soff);
    else if (i == 1) f2_dma_piece<1024>(rsrc, voff, soff);
    else if (i == 2) f2_dma_piece<2048>(rsrc, voff, soff);
    else f2_dma_piece<3072>(rsrc, voff, soff);
}

struct F2Split {
    f16x8 h, m;
};

// ---- range control of the fp16x2 forward (round 4) ---------------------------------------------------------------------------
// fp16 ends at 65504, and a trained 88-layer residual stream is not bounded a priori.  A ReLU net is positively homogeneous:
// dividing X_0 = relu(head) and every BODY bias by s divides every activation (x_b and t_b alike) by s, with the body weights
// untouched; the tail multiplies its dot products by s again.  For s a power of two all of this is exact in fp32, so the body
// bias stages are PACKED for the scale s the previous launches asked for, the kernels multiply the head's fp32 accumulators by
// 1 / s where X_0 is formed, and by s where values leave the chain (tail, y slot of the stash; the weight-gradient GEMMs that
// read the stash of x / s multiply dW by s at their flush).  (Round 4 divided the head's WEIGHTS by s before their fp16 split:
// fp16 has no spare exponent range — weights of ~0.03 lose their mid half from s = 2^4 on and flush to zero near 2^24, ADVICE
// r4.  The head stages and the head bias are packed unscaled now: the head GEMM runs on the encoding, which is bounded.)
// Everything lives in the 16 status words behind the stage stream, on the device — no host round trip:
//   FLAG      != 0: the fp16x2 launch in flight saw |value| >= R2L_F2_RANGE (or s is exhausted): the bf16x3 kernel behind redoes it
//   AMAX      largest |B value| (scaled units, float bits; atomicMax) of the launches since the scale was last committed
//   SCALE/INV s and 1/s of the packed stream (floats)
//   MAGIC     the area has been initialised by a pack of this library (else: s = 1, nothing known)
//   TRIPS / RESCALES   launches that fell back / commits that changed s (telemetry)
//   PEAK      unscaled amax of the epoch closed by the last commit (telemetry; host reports max(PEAK, AMAX * SCALE))
//   DONE / GO fallback protocol: workgroups of the fallback pack that have finished; the bf16x3 forward runs iff GO != 0
//   REFINE    the last re-scale was a blind jump (AMAX was inf): the kernel behind the next launch re-packs once more, for the
//             scale that launch's AMAX — now the truth — asks for (no fallback involved)
// Policy (f2_next_scale): keep AMAX / s within [2^10, 2^13] — s = 1 for every net whose activations stay below 8192, so the
// default-init and the measured trained nets run bit-for-bit as before.  A launch that trips is redone by the bf16x3 kernel
// ONCE; the fallback's pack kernel re-packs the scaled stages (head + bias stages, 2.4 MB) for the new s and its last
// workgroup commits it, so the next launch is back on the fp16 kernels: the guard is per launch, not sticky.
// (the word indices F2S_* live in r2l_common.h, next to the status area's offset)
#define F2_SCALE_MAX_EXP 24   // s <= 2^24: activations up to ~1e11 stay on the fp16 kernels
struct F2Next {
    float s, inv;   // the scale to pack for
    float peak;     // unscaled amax of the epoch this decision closes
    bool tripped;   // FLAG was set
    bool stuck;     // tripped with s already at its maximum: the flag stays (sticky fallback, as before round 4)
    bool changed;
    bool blind;     // tripped, and AMAX was not usable: s jumped by 2^8 (REFINE is set)
};
// does the kernel launched behind an fp16 launch have work?  FLAG (redo + re-scale), or a pending refinement with a fresh AMAX
__device__ __forceinline__ bool f2_rescale_due(const unsigned* st, bool& tripped) {
    tripped = __builtin_nontemporal_load(st + F2S_FLAG) != 0u;
    return tripped || (st[F2S_REFINE] != 0u && st[F2S_AMAX] != 0u);
}
__device__ __forceinline__ int f2_ceil_log2(float a) {  // smallest e with a < 2^e (a > 0, finite)
    int e = 0;
    (void)frexpf(a, &e);
    return e;
}
__device__ __forceinline__ F2Next f2_next_scale(const unsigned* st) {
    F2Next r;
    const bool valid = st[F2S_MAGIC] == F2_MAGIC;
    int es = 0;
    if (valid) {
        int e = 0;
        const float m = frexpf(__builtin_bit_cast(float, st[F2S_SCALE]), &e);
        if (m == 0.5f && e >= 1 && e <= F2_SCALE_MAX_EXP + 1) es = e - 1;
    }
    const float a = valid ? __builtin_bit_cast(float, st[F2S_AMAX]) : 0.f;  // scaled units; inf once an operand overflowed
    r.tripped = valid && st[F2S_FLAG] != 0u;
    r.blind = false;
    int en = es;
    if (r.tripped) {
        // a value in [32768, 65504) tripped the guard while everything was still finite: AMAX is the truth; beyond that an
        // operand became inf and AMAX only says "too large": jump 2^8 and let the next launch's AMAX refine it
        r.blind = !(a < 60000.f);
        en = r.blind ? es + 8 : es + (f2_ceil_log2(a) - 13 > 1 ? f2_ceil_log2(a) - 13 : 1);
    } else if (a >= 8192.f) {
        en = es + f2_ceil_log2(a) - 13;
    } else if (a > 0.f && a < 1024.f && es > 0) {
        en = es + f2_ceil_log2(a) - 13;
        if (en < 0) en = 0;
    }
    if (en > F2_SCALE_MAX_EXP) en = F2_SCALE_MAX_EXP;
    r.stuck = r.tripped && en == es;
    r.changed = en != es;
    r.s = ldexpf(1.0f, en);
    r.inv = ldexpf(1.0f, -en);
    r.peak = a * ldexpf(1.0f, es);
    return r;
}
// one thread: close the epoch.  fallback: called by the last workgroup of the fallback pack (the bf16x3 forward behind it runs)
__device__ __forceinline__ void f2_commit_scale(unsigned* st, const F2Next& nx, bool fallback) {
    const bool valid = st[F2S_MAGIC] == F2_MAGIC;
    const float old_peak = valid ? __builtin_bit_cast(float, st[F2S_PEAK]) : 0.f;
    st[F2S_TRIPS] = (valid ? st[F2S_TRIPS] : 0u) + ((fallback && nx.tripped) ? 1u : 0u);
    st[F2S_RESCALES] = (valid ? st[F2S_RESCALES] : 0u) + (nx.changed ? 1u : 0u);
    st[F2S_PEAK] = __builtin_bit_cast(unsigned, fallback ? fmaxf(old_peak, nx.peak) : (nx.peak > 0.f ? nx.peak : old_peak));
    st[F2S_SCALE] = __builtin_bit_cast(unsigned, nx.s);
    st[F2S_INV] = __builtin_bit_cast(unsigned, nx.inv);
    st[F2S_AMAX] = 0u;
    st[F2S_DONE] = 0u;
    st[F2S_GO] = (fallback && nx.tripped) ? 1u : 0u;
    st[F2S_REFINE] = (nx.blind && !nx.stuck) ? 1u : 0u;
    st[F2S_MAGIC] = F2_MAGIC;
    st[F2S_FLAG] = nx.stuck ? 1u : 0u;
}
// end of a chain kernel: the wave's largest |B value| goes to AMAX, the guard to FLAG (one atomic pair per wave)
template <int AMAX_WORD = F2S_AMAX>
__device__ __forceinline__ void f2_report_amax(unsigned* st, float amax, int lane) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) {
        atomicMax(st + AMAX_WORD, __builtin_bit_cast(unsigned, amax));
        if (!(amax < R2L_F2_RANGE)) atomicOr(st, 1u);  // (FLAG is word 0 of both status areas)
    }
}

// ---- pack: flat fp32 parameters -> fp16x2 forward stage stream (shared by r2l_fwd2.hip's pack and the fallback pack of
// r2l_fwd3.hip).  A stage is [split sp (2)][tile t][lane (i,h)][slot s] fp16; bias stage: split region 0 only, slots 0, 1 of
// half 0 = hi, mid.  The BODY biases are multiplied by inv_s (a power of two: exact); head weights and head bias are packed as
// they are (the kernels scale the head's fp32 accumulators).  only_scaled: just the stages that depend on the scale (the body's
// bias stages).
__host__ __device__ static inline int64_t f2_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t f2_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t f2_off_body_b(int layer) { return f2_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t f2_off_tail_w(int n_block) { return f2_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t f2_off_tail_b(int n_block) { return f2_off_tail_w(n_block) + 3 * R2L_W; }
__device__ __forceinline__ unsigned short f2_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }
// one element (stage g, index `within` = ((tile * 64) + lane) * 8 + s of its 4096) of the stream
__device__ __forceinline__ void f2_pack_fwd_element(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                                    float inv_s, int64_t stages, int64_t g, int within) {
    const int s = within & 7, lane = (within >> 3) & 63, tile = (within >> 9) & 7;
    const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
    unsigned short* st = out + g * (F2_STAGE_BYTES / 2);
    unsigned short v0 = 0, v1 = 0;
    if (g < stages) {
        bool bias_stage = false;
        float w = 0.f;
        if (g == 0) {
            bias_stage = true;
            w = params[f2_off_head_b() + o];
        } else if (g < 64) {
            const int v = 8 * (int)(g - 1) + s;
            int col;
            if (v < 480) {
                const int ci = v / 20, within20 = v % 20, f = within20 >> 1;
                col = 21 * (3 * (8 * h + ci / 3) + ci % 3) + ((within20 & 1) ? 10 + f : f);
            } else {
                const int e = v - 480;
                col = 21 * (3 * (8 * h + e / 3) + e % 3) + 20;
            }
            w = params[(int64_t)o * R2L_IN + col];
        } else {
            const int layer = (int)((g - 64) / 17), r17 = (int)((g - 64) % 17);
            if (r17 == 0) {
                bias_stage = true;
                w = params[f2_off_body_b(layer) + o] * inv_s;
            } else {
                const int kb = r17 - 1, T = kb >> 1, r = kb & 1;
                const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
                w = params[f2_off_body_w(layer) + (int64_t)o * R2L_W + in];
            }
        }
        const _Float16 hi = (_Float16)w;
        const _Float16 mid = (_Float16)(w - (float)hi);
        if (bias_stage) {
            v0 = (h == 0) ? (s == 0 ? f2_bits(hi) : (s == 1 ? f2_bits(mid) : (unsigned short)0)) : (unsigned short)0;
        } else {
            v0 = f2_bits(hi); v1 = f2_bits(mid);
        }
    }
    st[within] = v0;
    st[8 * 64 * 8 + within] = v1;
}
__device__ __forceinline__ void f2_pack_fwd_elements(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                                     float inv_s, bool only_scaled, int64_t first, int64_t stride) {
    const int64_t stages = r2l_fwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int64_t g = idx >> 12;
        if (only_scaled && !(g >= 64 && g < stages && (g - 64) % 17 == 0)) continue;
        f2_pack_fwd_element(params, out, n_block, inv_s, stages, g, (int)(idx & 4095));
    }
}
// the stages that hold no body WEIGHT: the head's 64 and the 2 n_block body bias stages (r2l_adam_step_packed: the body weight
// stages were written by the optimizer kernel itself)
__device__ __forceinline__ void f2_pack_fwd_nonbody(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block,
                                                    float inv_s, int64_t first, int64_t stride) {
    const int64_t stages = r2l_fwd3_stages(n_block);
    const int64_t total = (int64_t)(64 + 2 * n_block) * 4096;
    for (int64_t idx = first; idx < total; idx += stride) {
        const int64_t sel = idx >> 12;
        const int64_t g = sel < 64 ? sel : 64 + 17 * (sel - 64);
        f2_pack_fwd_element(params, out, n_block, inv_s, stages, g, (int)(idx & 4095));
    }
}

// ---- fp16 stash of the training trio (r2l_fwd2<SAVE> / r2l_bwd2 -> r2l_dw16.hip) ------------------------------------------
// What the weight-gradient GEMMs read of a layer input / output gradient is its fp16 `hi` part only (dW = G_hi^T A_hi, one
// fp16 MFMA product per fp32 product: the rounding of both operands to 11 bits moves dW by ~5e-5 relative, the level at
// which two fp32 evaluations of this 88-layer net differ).  The hi halves ARE the B operand of the chain's next
// stage, so a chain stashes a stage's B operand as it stands: ONE 16-byte store per lane and stage, the 64 lanes writing one
// contiguous KiB.  A slot of the buffers then holds, per 32-ray tile, 16 stage pieces of 1 KiB:
//     byte  tile*16384 + kb*1024 + lane*16 + 2*s   <->   ray 32*tile + (lane & 31),
//                                                        feature 16*kb + 8*(s >> 2) + 4*(lane >> 5) + (s & 3)
// (half the bytes of the fp32 stash of the bf16x3 trio, which keeps the slot size: the range-guard fallback rewrites a slot
// in that format).  Word R2L_STASH_FMT_WORD of save_x says which format the forward left behind (0: this one).
#define R2L_H16_TILE_UNITS 1024  // 16-byte units per tile and slot
// Exact weight gradients (r2l_config.dw_mode = R2L_DW_EXACT): the chains also stash the MID halves of the same operands, as a
// second set of stage pieces R2L_H16_MID_BYTES(Np) behind the first in the slot (the half of its data area the hi pieces leave
// free), and r2l_dw16 / r2l_dw_head16 take hi*hi + hi*mid + mid*hi — fp32-grade products, twice the stash traffic.  A stage
// stores the mid half of its OWN B operand (f2_stage's hmid), the hi half of the NEXT stage's (hst).
#define R2L_H16_MID_BYTES(Np) ((int64_t)(Np) * 512)
struct F2Hst {                   // where this lane's 16 B of the NEXT stage's B operand go (on == false: no stash)
    bool on;
    float* slot;    // the stash slot (uniform: a scalar register pair; the chain's launcher advances it per block)
    unsigned voff;  // this lane's byte offset of stage piece 0 in the slot: (tile * 1024 + lane) * 16
    unsigned soff;  // + the stage piece, 1024 * kb (uniform: a scalar register, no address arithmetic on the VALU)
};
__device__ __forceinline__ void f2_hst_store(const F2Hst& h, u32x4 v) {
    // raw buffer store, no bounds (the padding rows of the last tile are written too).  Whole 128-byte lines, written once,
    // read back milliseconds later by another kernel: non-temporal (aux bit 1 = nt)
#ifndef F2_HST_AUX
#define F2_HST_AUX 18  // cache policy of the stash stores: nt + sc1 (same-box A/B of 0 plain, 2 nt, 3 nt + sc0, 18: profiles/r05_stash_store_ab.txt)
#endif
#ifndef F2_HST_OFF    // (timing build: no stash store at all — the MFMA operands are unchanged, the gradients are garbage)
    __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc(h.slot, 0, -1, 0x00020000), h.voff, h.soff, F2_HST_AUX);
#endif
}
struct F2A4 {  // A operands (fp16 pairs) of four output tiles
    f16x8 h[4], m[4];
};

// side work of one half stage in six steps, two per group of four MFMAs:
//   A operand reads of the NEXT half stage: 8 (2 per step, steps 0..3); DMA pieces: 4 (steps 0..3);
//   four B values of the next stage: gather (0), hi (1), residual (2), mid (3)
template <bool BIAS_A, class Gather>
struct F2Side {
    F2A4& a;
    const unsigned char* lb;  // lane base of the stage buffer the A operands come from
    int half;
    Gather gather;
    bool want_b;
    F3Dma dma;
    float& amax;  // running max |B value| of this lane (range guard: fp16 ends at 65504)
    F3Dma extra = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u};  // one more 1 KiB piece, issued with step 5 (backward: mask tile)
    F2Hst hst = F2Hst{false, nullptr, 0u, 0u};  // second half stage only: stash the assembled hi operand (step 4)
    const unsigned* uh_first = nullptr;  // the first half stage's packed hi pairs (values 0-3 of the operand)
#ifdef F2_MID_LATE  // (A/B: the mid-half store of exact weight gradients as side step 5 of the second half stage)
    F2Hst hmid_late = F2Hst{false, nullptr, 0u, 0u};
    u32x4 mid_bits = u32x4{0u, 0u, 0u, 0u};
#endif
    float x[4];
    unsigned uh[2], um[2];  // packed fp16 pairs: values (0,1) and (2,3)
    static __device__ __forceinline__ unsigned pk(float a0, float a1) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, h2));
    }
    static __device__ __forceinline__ float lo(unsigned u) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return (float)__builtin_bit_cast(h2, u)[0];
    }
    static __device__ __forceinline__ float hi(unsigned u) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return (float)__builtin_bit_cast(h2, u)[1];
    }
    __device__ __forceinline__ void loads(int i) {
#pragma unroll
        for (int k = 2 * i; k < 2 * i + 2; ++k) {
            if (k >= (BIAS_A ? 4 : 8)) continue;
            const int tt = BIAS_A ? k : k / 2, sp = BIAS_A ? 0 : k % 2;
            const f16x8 v = *reinterpret_cast<const f16x8*>(lb + (sp * 8 + 4 * half + tt) * 1024);
            if (sp == 0) a.h[tt] = v;
            else a.m[tt] = v;
        }
    }
    __device__ __forceinline__ void step(int i) {
        loads(i);
        if (dma.on && i < 4) {
            if (i == 0) f2_dma_base(dma.la);
            f2_dma_piece_i(i, dma.rs, dma.voff, dma.so);
        }
        if (i == 5 && extra.on) f3_dma16(extra.rs, extra.voff, extra.so, extra.la);
#ifdef F2_MID_LATE
        if (i == 5 && hmid_late.on) f2_hst_store(hmid_late, mid_bits);
#endif
        if (!want_b) return;
        if (i == 0) {
            gather(x);
#ifndef F2_NO_AMAX
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));  // v_max3_f32
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
#endif
        } else if (i == 1) {
            uh[0] = pk(x[0], x[1]);
            uh[1] = pk(x[2], x[3]);
        } else if (i == 2) {
            x[0] = f2_res_lo(uh[0], x[0]); x[1] = f2_res_hi(uh[0], x[1]);
            x[2] = f2_res_lo(uh[1], x[2]); x[3] = f2_res_hi(uh[1], x[3]);
        } else if (i == 3) {
            um[0] = pk(x[0], x[1]);
            um[1] = pk(x[2], x[3]);
        } else if (i == 4) {
            if (hst.on) f2_hst_store(hst, u32x4{uh_first[0], uh_first[1], uh[0], uh[1]});
        }
    }
};

// acc[4 tiles of `half`] (+)= W . b: three fp16 MFMAs per tile, small terms first, term-major; after every group of four
// MFMAs two steps of `side`.  BIAS stage: one MFMA per tile (hi, mid of the bias in k slots 0, 1 against ones).
template <bool BIAS, bool ZERO_INIT, class Side>
__device__ __forceinline__ void f2_mfma_half(f32x16 (&acc)[R2L_NT], int half, const F2A4& a, const F2Split& b, Side& side) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (BIAS) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h[t], b.h, ZERO_INIT ? zero : acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) side.step(i);
        return;
    }
#define F2_GROUP(AA, BB, I)                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[4 * half + t] =                                                   \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(AA[t], BB, acc[4 * half + t], 0, 0, 0);                                   \
    side.step(I);                                                                                                       \
    side.step(I + 1);                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
    if (ZERO_INIT) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.m[t], b.h, zero, 0, 0, 0);
        side.step(0);
        side.step(1);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        F2_GROUP(a.m, b.h, 0)
    }
    F2_GROUP(a.h, b.m, 2)
    F2_GROUP(a.h, b.h, 4)
#undef F2_GROUP
}

// ---- the head's sin / cos, software-pipelined (round 4) ----------------------------------------------------------------------
// The 63 encoding stages of a tile gathered their B values with r2l_sincos in side step 0: a DEPENDENT chain of ~20 VALU
// instructions (range reduction -> polynomial -> quadrant) between two groups of four MFMAs, with nothing else on the SIMD: the
// matrix pipe drained every half stage (timing build without sin / cos: 38.0 -> 36.2 ms per render launch, 4.8 % of the kernel;
// halving the NUMBER of evaluations by angle doubling bought 0.4 %: it is the chain's latency, not its instruction count).
// F2TrigPre cuts the evaluation into four phases that ride in side steps 2 - 5 of the PREVIOUS stage's half (a few instructions
// each, short chains), writing four registers that the next stage's gather (F2TakeRegs) merely copies.  Values are r2l_sincos's,
// bit for bit (same operations in the same order); the second pair of a gather is the angle doubling of the first (r2l_common.h).
struct F2TrigPre {
    float x;
    int f0;
    float (&out)[4];  // (sin, cos)(x 2^f0), (sin, cos)(x 2^(f0+1)): complete after phase 3
    float (&st)[5];   // in flight between the phases
    __device__ __forceinline__ void phase(int ph) const {
        if (ph == 0) {
            const float a = x * (float)(1 << f0);
            const float n = rintf(a * 0.63661977236758134f);
            float r = __builtin_fmaf(-n, 1.57079637050628662109375f, a);
            r = __builtin_fmaf(-n, -4.37113900018624283e-8f, r);
            r = __builtin_fmaf(-n, -1.71512449512872556e-15f, r);
            st[0] = n; st[1] = r;
        } else if (ph == 1) {
            const float r = st[1], r2 = r * r;
            float ps = __builtin_fmaf(r2, 2.718311493989822e-6f, -1.9839334836096632e-4f);
            ps = __builtin_fmaf(ps, r2, 8.3333293858894632e-3f);
            ps = __builtin_fmaf(ps, r2, -1.6666666641626524e-1f);
            float pc = __builtin_fmaf(r2, 2.439044879627741e-5f, -1.388676377460993e-3f);
            pc = __builtin_fmaf(pc, r2, 4.1666623323739063e-2f);
            pc = __builtin_fmaf(pc, r2, -0.5f);
            st[2] = r2; st[3] = ps; st[4] = pc;
        } else if (ph == 2) {
            const float r = st[1], r2 = st[2];
            st[3] = __builtin_fmaf(st[3] * r2, r, r);    // sin(r)
            st[4] = __builtin_fmaf(st[4], r2, 1.0f);     // cos(r)
        } else {
            const int q = (int)st[0];
            const float sr = st[3], cr = st[4];
            const float s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
            out[0] = (q & 2) ? -s1 : s1;
            out[1] = ((q + 1) & 2) ? -c1 : c1;
            r2l_sincos_double(out[0], out[1], out[2], out[3]);
        }
    }
};
struct F2NoPre {
    __device__ __forceinline__ void phase(int) const {}
};
struct F2TakeRegs {  // gather = the four values a F2TrigPre finished one stage ago
    const float (&r)[4];
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = r[s];
    }
};
template <class Side, class Pre>
struct F2SideWithPre {  // a F2Side whose steps 2 .. 5 also advance a pre-computation by one phase each
    Side s;
    Pre pre;
    __device__ __forceinline__ void step(int i) {
        s.step(i);
        if (i >= 2) pre.phase(i - 2);
    }
};

// state of the weight-staging pipeline (r2l_f3.h's F3PipeT for 16 KiB stages: 4 pieces per wave and stage)
struct F2Pipe {
    u32x4 rs;
    unsigned lds0;
    unsigned voff, wq;  // lane * 16 ; this wave's quarter of a stage (4096 * wave)
    const unsigned char* base;
    int lane;
    int gb;
    int gq, gqb;
    const unsigned char* lb;
    F2A4 a1, a2;
    F2Split sb;
    F2Split ones;
    float amax;  // max |B value| seen by this lane
    __device__ __forceinline__ void issue() {
        const unsigned so = (unsigned)gq * F2_STAGE_BYTES + wq;
        const unsigned la = lds0 + (unsigned)gqb * F2_STAGE_BYTES + wq;
        f2_dma_base(la);
#pragma unroll
        for (int i = 0; i < 4; ++i) f2_dma_piece_i(i, rs, voff, so);
        ++gq;
        gqb = (gqb == F2_NBUF - 1) ? 0 : gqb + 1;
    }
    // vmcnt retires in order: the own loads of stage k+1 have the 4 x 3 loads of stages k+2..k+4 behind them
    __device__ __forceinline__ void sync_next() {
        // F2_NBUF buffers: stages k+2 .. k+F2_NBUF-2 (4 loads each) may still fly behind the loads of stage k+1
        if (F2_NBUF == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        static_assert(F2_NBUF == 6 || F2_NBUF == 8, "wait immediates written for 6 or 8 buffers");
#ifndef F2_TIMING_NO_BARRIER  // (timing experiment only: results are wrong without the barrier)
        __syncthreads();
#endif
        gb = (gb == F2_NBUF - 1) ? 0 : gb + 1;
        lb = base + gb * F2_STAGE_BYTES + lane * 16;
    }
    __device__ __forceinline__ F3Dma request() {
        F3Dma d{true, rs, voff, (unsigned)gq * F2_STAGE_BYTES + wq, lds0 + (unsigned)gqb * F2_STAGE_BYTES + wq};
        ++gq;
        gqb = (gqb == F2_NBUF - 1) ? 0 : gqb + 1;
        return d;
    }
};

// One stage (see f3_stage): acc (+)= stage k; glo / ghi fill the B values 0-3 / 4-7 of stage k+1.
template <bool BIAS_K, bool ZERO_K, bool BIAS_NEXT, class GLo, class GHi>
__device__ __forceinline__ void f2_stage(f32x16 (&acc)[R2L_NT], F2Pipe& P, GLo glo, GHi ghi,
                                         F3Dma extra_a = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u},
                                         F3Dma extra_b = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u},
                                         F2Hst hst = F2Hst{false, nullptr, 0u, 0u},
                                         F2Hst hmid = F2Hst{false, nullptr, 0u, 0u}) {
    // exact weight gradients: the MID half of THIS stage's B operand goes to its own piece (hmid: the stage's piece +
    // R2L_H16_MID_BYTES).  It is the register quad the MFMAs of the stage read: no copy, nothing extra kept alive.
#ifndef F2_MID_LATE
    if (!BIAS_K && hmid.on) f2_hst_store(hmid, __builtin_bit_cast(u32x4, P.sb.m));
#endif
    F2Side<BIAS_K, GLo> sa{P.a2, P.lb, 1, glo, !BIAS_NEXT, F3Dma{false, P.rs, 0u, 0u, 0u}, P.amax, extra_a};
    f2_mfma_half<BIAS_K, ZERO_K>(acc, 0, P.a1, P.sb, sa);
    __builtin_amdgcn_sched_barrier(0);
    P.sync_next();
#ifdef F2_MID_LATE
    F2Side<BIAS_NEXT, GHi> sb2{P.a1, P.lb, 0, ghi, !BIAS_NEXT, P.request(), P.amax, extra_b, hst, sa.uh,
                               F2Hst{!BIAS_K && hmid.on, hmid.slot, hmid.voff, hmid.soff}, __builtin_bit_cast(u32x4, P.sb.m)};
#else
    F2Side<BIAS_NEXT, GHi> sb2{P.a1, P.lb, 0, ghi, !BIAS_NEXT, P.request(), P.amax, extra_b, hst, sa.uh};
#endif
    f2_mfma_half<BIAS_K, ZERO_K>(acc, 1, P.a2, P.sb, sb2);
    __builtin_amdgcn_sched_barrier(0);
    if (BIAS_NEXT) {
        P.sb = P.ones;
    } else {
        P.sb.h = __builtin_bit_cast(f16x8, u32x4{sa.uh[0], sa.uh[1], sb2.uh[0], sb2.uh[1]});
        P.sb.m = __builtin_bit_cast(f16x8, u32x4{sa.um[0], sa.um[1], sb2.um[0], sb2.um[1]});
    }
}

// f2_stage with a pre-computation riding in each half (head of r2l_fwd2.hip): plo / phi advance in the first / second half stage
template <bool BIAS_K, bool ZERO_K, class GLo, class GHi, class PLo, class PHi>
__device__ __forceinline__ void f2_stage_pre(f32x16 (&acc)[R2L_NT], F2Pipe& P, GLo glo, GHi ghi, PLo plo, PHi phi) {
    typedef F2Side<BIAS_K, GLo> SA;
    typedef F2Side<false, GHi> SB;
    F2SideWithPre<SA, PLo> sa{SA{P.a2, P.lb, 1, glo, true, F3Dma{false, P.rs, 0u, 0u, 0u}, P.amax}, plo};
    f2_mfma_half<BIAS_K, ZERO_K>(acc, 0, P.a1, P.sb, sa);
    __builtin_amdgcn_sched_barrier(0);
    P.sync_next();
    F2SideWithPre<SB, PHi> sb2{SB{P.a1, P.lb, 0, ghi, true, P.request(), P.amax}, phi};
    f2_mfma_half<BIAS_K, ZERO_K>(acc, 1, P.a2, P.sb, sb2);
    __builtin_amdgcn_sched_barrier(0);
    P.sb.h = __builtin_bit_cast(f16x8, u32x4{sa.s.uh[0], sa.s.uh[1], sb2.s.uh[0], sb2.s.uh[1]});
    P.sb.m = __builtin_bit_cast(f16x8, u32x4{sa.s.um[0], sa.s.um[1], sb2.s.um[0], sb2.s.um[1]});
}
