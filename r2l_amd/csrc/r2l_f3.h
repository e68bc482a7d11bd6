// r2l_f3.h — machinery shared by the kernels that run fp32-accurate GEMM chains on the bf16 matrix pipe (r2l_fwd3.hip,
// r2l_bwd3.hip, r2l_teacher3.hip): bf16 (hi, mid, lo) triples, the LDS-DMA weight staging pipeline and the stage routine
// with its hand-interleaved side work.  See the header comment of r2l_fwd3.hip for the scheme.  The gatherers at the end
// (layer inputs with their ride-along stash stores, positional-encoding values) are shared with the fp16x2 kernels (r2l_f2.h).
#pragma once
#include "r2l_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define F3_STAGE_BYTES 24576  // 3 splits x 8 tiles x 64 lanes x 16 B
#define F3_NBUF 6


// ---- bf16 helpers (round to nearest even; NaN / inf are not expected in weights or activations) -----------------------
__host__ __device__ static inline unsigned short f3_bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ static inline float f3_bf16_to_f(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}


// one LDS-DMA load: 64 lanes x 16 B from (rsrc, voff + soff) to LDS at lds_addr + 16*lane.  Inline asm on purpose: the
// compiler's waitcnt insertion treats the builtin form conservatively (vmcnt(0) before every LDS read), which would
// collapse the multi-stage prefetch; the waits are placed by hand (F3Pipe::sync_next).
__device__ __forceinline__ void f3_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct F3Split {
    bf16x8 h, m, l;
};
struct F3A4 {  // A operands (bf16 triples) of four output tiles
    bf16x8 h[4], m[4], l[4];
};

// The work that rides along the 24 MFMAs of one half stage, cut into six steps (one per group of four MFMAs):
//   * the twelve LDS reads of the A operands needed NEXT (two per step),
//   * four B values of the next stage and their split into bf16 (hi, mid, lo): gather, cvt, sub, cvt, sub, cvt.
// hipcc would issue all of it as one burst in front of the MFMAs (with one wave per SIMD nothing else feeds the matrix
// pipe meanwhile) and its sched_group_barrier solver does not terminate on this kernel, so the interleave is written out
// and fenced with sched_barrier(0).
struct F3Dma {  // one stage request, issued one 1 KiB piece per step (dma.on: this side issues it)
    bool on;
    u32x4 rs;
    unsigned voff, so, la;
};
template <bool BIAS_A, class Gather>
struct F3Side {
    F3A4& a;                  // destination of the A operands
    const unsigned char* lb;  // lane base of the stage buffer they come from
    int half;                 // which four tiles
    Gather gather;            // fills v[4] with the next stage's B values (lo or hi half)
    bool want_b;
    F3Dma dma;
    F3Dma extra = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u};  // one more 1 KiB piece, issued with step 5 (backward: mask tile)
    float x[4], r1[4];
    unsigned uh[2], um[2], ul[2];  // packed bf16 pairs: values (0,1) and (2,3)
    // two fp32 -> one dword of two bf16 (v_cvt_pk_bf16_f32), and back (shift / mask)
    static __device__ __forceinline__ unsigned pk(float a0, float a1) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
    }
    static __device__ __forceinline__ float lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
    static __device__ __forceinline__ float hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
    __device__ __forceinline__ void loads(int i) {  // A operand loads 2i, 2i+1 of the twelve (bias stage: of the four)
#pragma unroll
        for (int k = 2 * i; k < 2 * i + 2; ++k) {
            if (BIAS_A && k >= 4) continue;
            const int tt = BIAS_A ? k : k / 3, sp = BIAS_A ? 0 : k % 3;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(lb + (sp * 8 + 4 * half + tt) * 1024);
            if (sp == 0) a.h[tt] = v;
            else if (sp == 1) a.m[tt] = v;
            else a.l[tt] = v;
        }
    }
    __device__ __forceinline__ void step(int i) {
        loads(i);
        if (dma.on) f3_dma16(dma.rs, dma.voff, dma.so + i * 1024u, dma.la + i * 1024u);
        if (i == 5 && extra.on) f3_dma16(extra.rs, extra.voff, extra.so, extra.la);
        if (!want_b) return;
        if (i == 0) {
            gather(x);
        } else if (i == 1) {
            uh[0] = pk(x[0], x[1]);
            uh[1] = pk(x[2], x[3]);
        } else if (i == 2) {
            r1[0] = x[0] - lo(uh[0]); r1[1] = x[1] - hi(uh[0]);
            r1[2] = x[2] - lo(uh[1]); r1[3] = x[3] - hi(uh[1]);
        } else if (i == 3) {
            um[0] = pk(r1[0], r1[1]);
            um[1] = pk(r1[2], r1[3]);
        } else if (i == 4) {
            r1[0] -= lo(um[0]); r1[1] -= hi(um[0]);
            r1[2] -= lo(um[1]); r1[3] -= hi(um[1]);
        } else {
            ul[0] = pk(r1[0], r1[1]);
            ul[1] = pk(r1[2], r1[3]);
        }
    }
};

// acc[4 tiles of `half`] (+)= W . b: six bf16 MFMAs per tile, small terms first, term-major so that an accumulator is
// touched every fourth MFMA; after every group of four MFMAs one step of `side`.  BIAS stage: one MFMA per tile (hi,
// mid, lo of the bias in k slots 0..2 against ones), then all the side work.
template <bool BIAS, bool ZERO_INIT, class Side>
__device__ __forceinline__ void f3_mfma_half(f32x16 (&acc)[R2L_NT], int half, const F3A4& a, const F3Split& b, Side& side) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (BIAS) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[t], b.h, ZERO_INIT ? zero : acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) side.step(i);
        return;
    }
#define F3_GROUP(AA, BB, I)                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[4 * half + t] =                                                   \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(AA[t], BB, acc[4 * half + t], 0, 0, 0);                                  \
    side.step(I);                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);
    if (ZERO_INIT) {  // first k-block of a GEMM without bias: C = 0 (inline constant) in the first group
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[t], b.h, zero, 0, 0, 0);
        side.step(0);
        __builtin_amdgcn_sched_barrier(0);
    } else {
        F3_GROUP(a.l, b.h, 0)
    }
    F3_GROUP(a.h, b.l, 1)
    F3_GROUP(a.m, b.m, 2)
    F3_GROUP(a.m, b.h, 3)
    F3_GROUP(a.h, b.m, 4)
    F3_GROUP(a.h, b.h, 5)
#undef F3_GROUP
}

// state of the weight-staging pipeline (everything wave-uniform except lane-derived offsets)
template <int NBUF>
struct F3PipeT {
    u32x4 rs;            // buffer descriptor of the stage stream
    unsigned lds0;       // LDS address of buffer 0
    unsigned voff, wq;   // lane * 16 ; this wave's quarter of a stage
    const unsigned char* base;  // generic pointer to buffer 0
    int lane;
    int gb;              // buffer of the stage being consumed
    int gq, gqb;         // next stage to request and its buffer
    const unsigned char* lb;    // this lane's base in the current stage's buffer
    F3A4 a1, a2;         // A operands: first / second half of the current stage
    F3Split sb;          // B triple of the current stage
    F3Split ones;
    __device__ __forceinline__ void issue() {
        const unsigned so = (unsigned)gq * F3_STAGE_BYTES + wq;
        const unsigned la = lds0 + (unsigned)gqb * F3_STAGE_BYTES + wq;
#pragma unroll
        for (int i = 0; i < 6; ++i) f3_dma16(rs, voff, so + i * 1024u, la + i * 1024u);
        ++gq;
        gqb = (gqb == NBUF - 1) ? 0 : gqb + 1;
    }
    // publish the next stage (k+1), refill the buffer everybody has left, advance the buffer cursor.  vmcnt retires in
    // order: `vmcnt(18)` (at most 18 outstanding) covers the own loads of stage k+1, which have the 18 loads of stages
    // k+2..k+4 behind them; loads the compiler knows about only make its own waits stricter.
    // (training: the ride-along stash stores sit between the DMA loads in the queue; `vmcnt(18)` stays sufficient — it then
    // also waits for a few of the oldest of them — and never becomes too weak, whatever their number)
    __device__ __forceinline__ void sync_next() {
        // NBUF buffers: stages k+2 .. k+NBUF-2 (6 loads each) may still fly behind the loads of stage k+1
        if (NBUF == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        static_assert(NBUF == 6 || NBUF == 5, "wait immediates are written for 6 or 5 buffers");
        __syncthreads();
        gb = (gb == NBUF - 1) ? 0 : gb + 1;
        lb = base + gb * F3_STAGE_BYTES + lane * 16;
    }
    // the next request as six pieces for the side work of the second half stage (instead of a burst behind the barrier)
    __device__ __forceinline__ F3Dma request() {
        F3Dma d{true, rs, voff, (unsigned)gq * F3_STAGE_BYTES + wq, lds0 + (unsigned)gqb * F3_STAGE_BYTES + wq};
        ++gq;
        gqb = (gqb == NBUF - 1) ? 0 : gqb + 1;
        return d;
    }
};

// One stage: acc (+)= stage k.  Entry: P.a1 = A(tiles 0-3) and P.sb = B triple of stage k.  glo / ghi fill the B values
// 0-3 / 4-7 of stage k+1; BIAS_NEXT: stage k+1 is a bias stage (only the `hi` A operands exist, B = ones).
// The barrier that publishes stage k+1 sits in the MIDDLE of stage k: behind it the first-half A operands of stage k+1
// are read from LDS while the second half of stage k still feeds the matrix pipe.
typedef F3PipeT<F3_NBUF> F3Pipe;

template <bool BIAS_K, bool ZERO_K, bool BIAS_NEXT, class Pipe, class GLo, class GHi>
__device__ __forceinline__ void f3_stage(f32x16 (&acc)[R2L_NT], Pipe& P, GLo glo, GHi ghi, F3Dma extra_a = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u},
                                         F3Dma extra_b = F3Dma{false, u32x4{0u, 0u, 0u, 0u}, 0u, 0u, 0u}) {
    F3Side<BIAS_K, GLo> sa{P.a2, P.lb, 1, glo, !BIAS_NEXT, F3Dma{false, P.rs, 0u, 0u, 0u}, extra_a};
    f3_mfma_half<BIAS_K, ZERO_K>(acc, 0, P.a1, P.sb, sa);
    __builtin_amdgcn_sched_barrier(0);
    P.sync_next();
    F3Side<BIAS_NEXT, GHi> sb2{P.a1, P.lb, 0, ghi, !BIAS_NEXT, P.request(), extra_b};
    f3_mfma_half<BIAS_K, ZERO_K>(acc, 1, P.a2, P.sb, sb2);
    __builtin_amdgcn_sched_barrier(0);
    if (BIAS_NEXT) {
        P.sb = P.ones;
    } else {
        P.sb.h = __builtin_bit_cast(bf16x8, u32x4{sa.uh[0], sa.uh[1], sb2.uh[0], sb2.uh[1]});
        P.sb.m = __builtin_bit_cast(bf16x8, u32x4{sa.um[0], sa.um[1], sb2.um[0], sb2.um[1]});
        P.sb.l = __builtin_bit_cast(bf16x8, u32x4{sa.ul[0], sa.ul[1], sb2.ul[0], sb2.ul[1]});
    }
}

// gatherers of four B values
// STASH: training launch (RELU: the block's ReLU mask bits are collected); STORE: the fp32 values go to the chunked stash
// here (the bf16x3 trio; the fp16 trio stashes the assembled fp16 operand instead: r2l_f2.h F2Hst)
template <bool RELU, bool STASH = false, bool STORE = STASH>
struct F3Take4 {  // four consecutive fragment registers c0 .. c0+3 of one tile (tile T given for the stash address)
    const f32x16& frag;
    int c0;
    float* stash;  // training: this lane's base in the chunked stash slot of the layer input (r2l_chunk_lane), or nullptr
    int T;         // piece (T, c0/4)
    unsigned* mword = nullptr;  // training, RELU: the block's mask word T>>1 of this lane (bits shifted in MSB-first)
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = RELU ? fmaxf(frag[c0 + s], 0.f) : frag[c0 + s];
        // the B values ARE the layer input (x_b, relu(t_b)): the stash store rides along, one 16-byte piece per half stage
        // (unconditional when STASH: a data-dependent branch per piece would cut the half stage's schedule in two)
        // chunked layout: the 64 lanes of a piece write one contiguous KiB (whole 128-byte lines, written once: non-temporal)
        if (STORE) r2l_chunk_store(stash + R2L_CHUNK_PIECE * (4 * T + (c0 >> 2)), f32x4{v[0], v[1], v[2], v[3]});
        if (STASH && RELU) {  // relu'(t) for the backward: word = 2*word + [t > 0]  (v_cmp + v_addc per value)
            unsigned w = *mword;
#pragma unroll
            for (int s = 0; s < 4; ++s) w = w + w + (frag[c0 + s] > 0.f ? 1u : 0u);
            *mword = w;
        }
    }
};
struct F3None {
    __device__ __forceinline__ void operator()(float (&v)[4]) const { v[0] = v[1] = v[2] = v[3] = 0.f; }
};

// gatherers of the head: four values of the positional encoding (shared by r2l_fwd3.hip and r2l_fwd2.hip)
struct F3Trig2 {  // (sin, cos) of x * 2^f0 and x * 2^(f0+1)
    float x;
    int f0;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#ifdef F3_TIME_NOSINCOS  // (timing experiment only: results are wrong)
        v[0] = x * (float)(1 << f0) * 1e-3f; v[1] = v[0] + 0.5f; v[2] = v[0] * 2.f; v[3] = v[2] + 0.5f;
#else
        // (both pairs from r2l_sincos here: this gatherer serves the bf16x3 kernels, whose products are fp32-exact and whose
        // training family is held to the strict 2e-5 Adam bar — the angle doubling of the fp16x2 kernels' head, r2l_f2.h
        // F2TrigPre, moved one weight of the three-Adam-steps parity test (bf16x3 trio) by 1.9e-4 when it was tried here)
        r2l_sincos(x * (float)(1 << f0), v[0], v[1]);
        r2l_sincos(x * (float)(1 << (f0 + 1)), v[2], v[3]);
#endif
    }
};
struct F3Ident4 {  // identity features: coordinates e0 .. e0+3 of this half-wave (point = o + d * z)
    const float (&o)[3];
    const float (&d)[3];
    const float (&z)[8];
    int e0;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = o[(e0 + s) % 3] + d[(e0 + s) % 3] * z[(e0 + s) / 3];
    }
};
struct F3TrigOrIdent {  // last stage of a head trip: first block of the next trip, or the first identity block
    bool ident;
    F3Trig2 tr;
    F3Ident4 id;
    __device__ __forceinline__ void operator()(float (&v)[4]) const {
        float w[4];
        tr(v);
        id(w);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = ident ? w[s] : v[s];
    }
};
