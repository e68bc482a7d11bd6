// r2l_fwd3.hip — the R2L student forward at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value is the exact sum of three bf16 numbers
// (hi + mid + lo: 8 + 8 + 8 mantissa bits, same exponent range as fp32), so a product a*b of fp32 operands is
// reproduced to ~2^-24 relative by the six bf16 products  hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi  (each exact
// in the MFMA's fp32 accumulator; the three dropped terms are below 2^-24).  Six bf16 MFMAs of K = 16 replace sixteen
// fp32 MFMAs of K = 2 at the same result quality: 2.67x fewer matrix-pipe cycles for the same 11.79 MFLOP/ray.
//
// Structure: the register-resident activation chain of r2l_forward.hip (one wave = 32 rays for the whole network, C/D
// fragment of layer n = B operand of layer n+1: with the k slots of a K=16 block numbered so that lane-half h supplies the
// eight features {8q+4h+e}, the eight B values of a block are eight CONSECUTIVE fragment registers) — but
//   * weights are pre-split into bf16 triples by r2l_pack_fwd3_kernel and streamed in 24 KiB stages (one k-block of 16 x
//     256 outputs x 3 splits) straight into LDS with `buffer_load_dwordx4 ... lds` (no registers), four buffers deep
//     (three stages ~1.9 us of latency cover), shared by the four waves of the workgroup: one barrier per stage;
//   * the eight B values of a block are split into (hi, mid, lo) on the VALU in the shadow of the block's 48 MFMAs;
//   * the bias of a layer is ONE MFMA per output tile: hi, mid, lo of the bias sit in three k slots against B = 1;
//   * X_0 stays in registers (the weight ring no longer needs them) for the outer residual.
// Used for forward-only launches (render / evaluation); training keeps the fp32-MFMA kernels.
#include "r2l_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define F3_STAGE_BYTES 24576  // 3 splits x 8 tiles x 64 lanes x 16 B
#define F3_NBUF 4

__host__ __device__ static inline int64_t f3_off_head_b() { return (int64_t)R2L_IN * R2L_W; }
__host__ __device__ static inline int64_t f3_off_body_w(int layer) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)layer * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t f3_off_body_b(int layer) { return f3_off_body_w(layer) + R2L_W * R2L_W; }
__host__ __device__ static inline int64_t f3_off_tail_w(int n_block) { return f3_off_body_w(2 * n_block); }
__host__ __device__ static inline int64_t f3_off_tail_b(int n_block) { return f3_off_tail_w(n_block) + 3 * R2L_W; }

// ---- bf16 helpers (round to nearest even; NaN / inf are not expected in weights or activations) -----------------------
__host__ __device__ static inline unsigned short f3_bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ static inline float f3_bf16_to_f(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}

// =================================================================================================================
// pack: flat fp32 parameters -> stage stream.  Stage g: 0 = head bias, 1..63 = head k-blocks, then per body layer
// [bias stage, 16 k-block stages].  A stage is [split sp][tile t][lane (i,h)][slot s] bf16: split sp of
// W[32t + i][feature(stage, h, s)].
//   head k-block b, half h, slot s:  v = 8b + s indexes the half's 504 encoding values: v < 480: coordinate ci = v/20 of
//     the half (sample 8h + ci/3, axis ci%3), frequency f = (v%20)/2, sin (even) / cos (odd);  v >= 480: identity of
//     coordinate v - 480.
//   body k-block kb = 2T + r: feature 32T + 8(2r + (s>>2)) + 4h + (s&3)  — fragment registers 8r .. 8r+7 of tile T.
//   bias stage: split region 0 only: slots 0,1,2 of half 0 = hi, mid, lo of bias[32t + i]; everything else 0.
// =================================================================================================================
__global__ void r2l_pack_fwd3_kernel(const float* __restrict__ params, unsigned short* __restrict__ out, int n_block) {
    const int64_t stages = r2l_fwd3_stages(n_block);
    const int64_t total = (stages + R2L_F3_PAD_STAGES) * 8 * 64 * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), tile = (int)((idx >> 9) & 7);
        const int64_t g = idx >> 12;
        const int i = lane & 31, h = lane >> 5, o = 32 * tile + i;
        unsigned short* st = out + g * (F3_STAGE_BYTES / 2);
        unsigned short v0 = 0, v1 = 0, v2 = 0;
        if (g < stages) {
            bool bias_stage = false;
            float w = 0.f;
            if (g == 0) {
                bias_stage = true;
                w = params[f3_off_head_b() + o];
            } else if (g < 64) {
                const int v = 8 * (int)(g - 1) + s;
                int col;
                if (v < 480) {
                    const int ci = v / 20, within = v % 20, f = within >> 1;
                    col = 21 * (3 * (8 * h + ci / 3) + ci % 3) + ((within & 1) ? 10 + f : f);
                } else {
                    const int e = v - 480;
                    col = 21 * (3 * (8 * h + e / 3) + e % 3) + 20;
                }
                w = params[(int64_t)o * R2L_IN + col];
            } else {
                const int layer = (int)((g - 64) / 17), r17 = (int)((g - 64) % 17);
                if (r17 == 0) {
                    bias_stage = true;
                    w = params[f3_off_body_b(layer) + o];
                } else {
                    const int kb = r17 - 1, T = kb >> 1, r = kb & 1;
                    const int in = 32 * T + 8 * (2 * r + (s >> 2)) + 4 * h + (s & 3);
                    w = params[f3_off_body_w(layer) + (int64_t)o * R2L_W + in];
                }
            }
            const unsigned short hi = f3_bf16_rne(w);
            const float r1 = w - f3_bf16_to_f(hi);
            const unsigned short mid = f3_bf16_rne(r1);
            const unsigned short lo = f3_bf16_rne(r1 - f3_bf16_to_f(mid));
            if (bias_stage) {
                v0 = (h == 0) ? (s == 0 ? hi : (s == 1 ? mid : (s == 2 ? lo : (unsigned short)0))) : (unsigned short)0;
            } else {
                v0 = hi; v1 = mid; v2 = lo;
            }
        }
        const int64_t e = ((int64_t)tile * 64 + lane) * 8 + s;
        st[e] = v0;
        st[8 * 64 * 8 + e] = v1;
        st[2 * 8 * 64 * 8 + e] = v2;
    }
}

// =================================================================================================================
// forward
// =================================================================================================================
struct F3Args {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    int H, Wimg;
    float focal;
    const unsigned char* stream;  // fwd3 stage stream
    const float* params;
    int n_block;
    float* rgb;
    int64_t N;
};

// one LDS-DMA load: 64 lanes x 16 B from (rsrc, voff + soff) to LDS at lds_addr + 16*lane.  Inline asm on purpose: the
// compiler's waitcnt insertion treats the builtin form conservatively (vmcnt(0) before every LDS read), which would
// collapse the three-stage prefetch; the waits are placed by hand (F3_STAGE_BEGIN).
__device__ __forceinline__ void f3_dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

struct F3Split {
    bf16x8 h, m, l;
};
// v = h + m + l exactly (three bf16 values per fp32 value), round-to-nearest-even at each step
__device__ __forceinline__ F3Split f3_split8(const float (&v)[8]) {
    F3Split r;
    f32x8 x;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = v[k];
    r.h = __builtin_convertvector(x, bf16x8);
    const f32x8 r1 = x - __builtin_convertvector(r.h, f32x8);
    r.m = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(r.m, f32x8);
    r.l = __builtin_convertvector(r2, bf16x8);
    return r;
}

// acc[8 tiles] += W_block . b   for one k-block (16 features): per output tile six bf16 MFMAs, small terms first.
// lb: this lane's base inside the stage buffer (buffer + 16*lane); A operand of (split sp, tile T) at lb + (8 sp + T) KiB.
__device__ __forceinline__ void f3_kblock(f32x16 (&acc)[R2L_NT], const float (&bv)[8], const unsigned char* lb) {
    const F3Split b = f3_split8(bv);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        bf16x8 ah[4], am[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int T = 4 * half + t;
            ah[t] = *reinterpret_cast<const bf16x8*>(lb + (0 * 8 + T) * 1024);
            am[t] = *reinterpret_cast<const bf16x8*>(lb + (1 * 8 + T) * 1024);
            al[t] = *reinterpret_cast<const bf16x8*>(lb + (2 * 8 + T) * 1024);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], b.h, acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], b.l, acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], b.m, acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], b.h, acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], b.m, acc[4 * half + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * half + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], b.h, acc[4 * half + t], 0, 0, 0);
    }
}

// acc (=|+=) bias: one MFMA per tile (hi, mid, lo of the bias in k slots 0..2 of half 0 against ones)
template <bool ZERO_INIT>
__device__ __forceinline__ void f3_bias(f32x16 (&acc)[R2L_NT], const unsigned char* lb, const bf16x8& ones_h0) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lb + T * 1024);
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones_h0, ZERO_INIT ? zero : acc[T], 0, 0, 0);
    }
}

// (sin, cos) of x * 2^f for four consecutive frequencies f0 .. f0+3 -> 8 B values in stream order
__device__ __forceinline__ void f3_trig4(float x, int f0, float (&out)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r2l_sincos(x * (float)(1 << (f0 + k)), out[2 * k], out[2 * k + 1]);
}

template <bool POSE>
__global__ __launch_bounds__(256, 1) void r2l_fwd3_kernel(const F3Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[F3_NBUF][F3_STAGE_BYTES];

    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    // every wave of the workgroup takes part in the weight staging and the barriers, also when its tile is past the end
    const int64_t ray = tile * R2L_TILE_RAYS + (lane & 31);
    const bool valid = ray < a.N;
    const int64_t rc = valid ? ray : a.N - 1;

    // ---- weight staging ----------------------------------------------------------------------------------------------
    const unsigned long long sa = (unsigned long long)a.stream;
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu,
                      0x00020000u};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&wbuf[0][0];
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned wq = (unsigned)wave * 6144u;  // this wave stages a quarter of every stage
    auto issue = [&](int gq) {
        const unsigned so = (unsigned)gq * F3_STAGE_BYTES + wq;
        const unsigned la = lds0 + (unsigned)(gq & (F3_NBUF - 1)) * F3_STAGE_BYTES + wq;
#pragma unroll
        for (int i = 0; i < 6; ++i) f3_dma16(rs, voff, so + i * 1024u, la + i * 1024u);
    };
    int g = 0;  // stage counter (wave-uniform)
    issue(0);
    issue(1);
    issue(2);
// own stage-g loads have landed (12 = the loads of stages g+1, g+2 may still fly), everybody's have after the barrier;
// then the buffer read during stage g-1 is refilled with stage g+3
#define F3_STAGE_BEGIN()                                                                       \
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                          \
    __syncthreads();                                                                           \
    issue(g + 3);                                                                              \
    const unsigned char* lb = &wbuf[0][0] + (g & (F3_NBUF - 1)) * F3_STAGE_BYTES + lane * 16;  \
    ++g;

    // ---- rays -----------------------------------------------------------------------------------------------------------
    float o[3], d[3];
    if constexpr (!POSE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[rc * 3 + k];
            d[k] = a.rays_d[rc * 3 + k];
        }
    } else {
        const int pj = (int)(rc / a.Wimg), pi = (int)(rc % a.Wimg);
        const float dx = ((float)pi - (float)a.Wimg * 0.5f) / a.focal;
        const float dy = -(((float)pj - (float)a.H * 0.5f) / a.focal);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = (dx * a.c2w[4 * k + 0] + dy * a.c2w[4 * k + 1]) + (-1.0f) * a.c2w[4 * k + 2];
            o[k] = a.c2w[4 * k + 3];
        }
    }
    float z[8];  // the 8 sample depths of this half-wave (samples 8h .. 8h+7)
    {
        const f32x4 lo0 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h);
        const f32x4 lo1 = *reinterpret_cast<const f32x4*>(a.ztab + 8 * h + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { z[k] = lo0[k]; z[4 + k] = lo1[k]; }
        if (a.t_rand != nullptr) {
            const f32x4 sp0 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h);
            const f32x4 sp1 = *reinterpret_cast<const f32x4*>(a.ztab + 16 + 8 * h + 4);
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(a.t_rand + rc * 16 + 8 * h + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { z[k] = lo0[k] + sp0[k] * u0[k]; z[4 + k] = lo1[k] + sp1[k] * u1[k]; }
        }
    }
    // vmcnt bookkeeping: counters retire in order, so `vmcnt(12)` at a stage start (at most 12 operations outstanding)
    // always covers the stage's own DMA loads, which have at least the 12 loads of the next two stages behind them; loads
    // the compiler knows about (rays above, tail weights below) only make its own waits stricter.

    bf16x8 ones_h0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones_h0[k] = (__bf16)((h == 0 && k < 3) ? 1.0f : 0.0f);

    f32x16 x[R2L_NT], t[R2L_NT], x0[R2L_NT];
    // ---- head ---------------------------------------------------------------------------------------------------------
    {
        F3_STAGE_BEGIN()
        f3_bias<true>(x, lb, ones_h0);
    }
    auto zsel = [&](int s) {
        float zz = z[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) zz = (s == k) ? z[k] : zz;
        return zz;
    };
#pragma unroll 1
    for (int it2 = 0; it2 < 4; ++it2) {  // two samples = six coordinates = three pairs = 15 k-blocks per trip
        const float za = zsel(2 * it2), zb = zsel(2 * it2 + 1);
        float xc[6];
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) xc[ci] = o[ci % 3] + d[ci % 3] * (ci / 3 == 0 ? za : zb);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const float xa = xc[2 * p], xb = xc[2 * p + 1];
            float bv[8];
            {
                f3_trig4(xa, 0, bv);
                F3_STAGE_BEGIN()
                f3_kblock(x, bv, lb);
            }
            {
                f3_trig4(xa, 4, bv);
                F3_STAGE_BEGIN()
                f3_kblock(x, bv, lb);
            }
            {
                r2l_sincos(xa * 256.0f, bv[0], bv[1]);
                r2l_sincos(xa * 512.0f, bv[2], bv[3]);
                r2l_sincos(xb, bv[4], bv[5]);
                r2l_sincos(xb * 2.0f, bv[6], bv[7]);
                F3_STAGE_BEGIN()
                f3_kblock(x, bv, lb);
            }
            {
                f3_trig4(xb, 2, bv);
                F3_STAGE_BEGIN()
                f3_kblock(x, bv, lb);
            }
            {
                f3_trig4(xb, 6, bv);
                F3_STAGE_BEGIN()
                f3_kblock(x, bv, lb);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // identity features: coordinates 8j .. 8j+7 of the half
        float bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int e = 8 * j + s;
            bv[s] = o[e % 3] + d[e % 3] * z[e / 3];
        }
        F3_STAGE_BEGIN()
        f3_kblock(x, bv, lb);
    }
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            x[T][c] = fmaxf(x[T][c], 0.f);  // X_0 = relu(head)
            x0[T][c] = x[T][c];
        }

    // ---- body -----------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int b = 0; b < a.n_block; ++b) {
        {  // t = W1 x + b1   (its ReLU is applied where t is consumed)
            F3_STAGE_BEGIN()
            f3_bias<true>(t, lb, ones_h0);
        }
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            float bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) bv[s] = x[kb >> 1][8 * (kb & 1) + s];
            F3_STAGE_BEGIN()
            f3_kblock(t, bv, lb);
        }
        {  // x += W2 relu(t) + b2
            F3_STAGE_BEGIN()
            f3_bias<false>(x, lb, ones_h0);
        }
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            float bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) bv[s] = fmaxf(t[kb >> 1][8 * (kb & 1) + s], 0.f);
            F3_STAGE_BEGIN()
            f3_kblock(x, bv, lb);
        }
    }
#undef F3_STAGE_BEGIN

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt) on the VALU -------------------------------------------------------------
    const float* tw = a.params + f3_off_tail_w(a.n_block) + 4 * h;
    float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = x[T][4 * q + e] + x0[T][4 * q + e];
#pragma unroll
                for (int c = 0; c < 3; ++c) p3[c] = __builtin_fmaf(wv[c][e], y, p3[c]);
            }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) p3[c] += __shfl_xor(p3[c], 32);
    if (valid && h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = p3[c] + a.params[f3_off_tail_b(a.n_block) + c];
            a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
int r2l_fwd3_pack(const float* params, int n_block, float* wstream3, hipStream_t stream) {
    hipLaunchKernelGGL(r2l_pack_fwd3_kernel, dim3(2048), dim3(256), 0, stream, params,
                       reinterpret_cast<unsigned short*>(wstream3), n_block);
    R2L_CHECK(hipGetLastError());
    return 0;
}

int r2l_fwd3_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream3, const float* params,
                     int n_block, float* rgb, int64_t N, hipStream_t stream) {
    F3Args a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab;
    a.stream = reinterpret_cast<const unsigned char*>(wstream3); a.params = params;
    a.n_block = n_block; a.rgb = rgb; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (c2w_host12) hipLaunchKernelGGL((r2l_fwd3_kernel<true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((r2l_fwd3_kernel<false>), grid, block, 0, stream, a);
    R2L_CHECK(hipGetLastError());
    return 0;
}
