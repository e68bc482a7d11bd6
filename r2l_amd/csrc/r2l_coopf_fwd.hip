// r2l_coopf_fwd.hip — cooperative fp16x2 forward of the R2L student (see r2l_coopf.h): rgb[N,3] =
// NeRF_v3_2.forward(PositionalEmbedder(10)(PointSampler.sample_train / sample_test(...))), the op sequence of
// /root/reference/model/nerf_raybased.py:94-126, 198-208, 461-465, 539-544, with ONE 32-ray tile per workgroup (four waves,
// 64 output features each) instead of one per wave: the kernel of the default fp16 trio for launches of a few thousand
// rays (BASELINE configs[2] / [3] read literally: 4096 rays per step; the per-GPU share of a strong-scaling step).
// Numerically it is r2l_fwd2.hip: same packed stream, same three fp16 products per fp32 product, same stash.
#include "r2l_coopf.h"
#include <type_traits>

__host__ __device__ static inline int64_t cf_off_tail_w(int n_block) {
    return (int64_t)R2L_IN * R2L_W + R2L_W + (int64_t)2 * n_block * (R2L_W * R2L_W + R2L_W);
}
__host__ __device__ static inline int64_t cf_off_tail_b(int n_block) { return cf_off_tail_w(n_block) + 3 * R2L_W; }

struct CfFwdArgs {
    const float* rays_o;
    const float* rays_d;
    const float* t_rand;
    const float* ztab;
    float c2w[12];
    int H, Wimg;
    float focal;
    const unsigned char* stream;  // fwd2 stage stream
    unsigned* status;             // range-guard word behind the stream (r2l_fwd2.hip's)
    const float* params;
    int n_block;
    float* rgb;
    float* save_x;  // training stash (fp16 stage pieces, r2l_f2.h) or nullptr
    float* save_t;
    int64_t N;
    int64_t mid_units;  // != 0: mid quads stashed this many 16-byte units behind the hi pieces (exact weight gradients)
    int n_two;          // mixed launch: workgroups 0 .. n_two - 1 (in fc_mixed_index order) take two ray tiles, the others one
    int xcd_major;
};

// The chain of one workgroup on the NT ray tiles tile0 .. tile0 + NT - 1.  LDS comes from the kernel (the mixed launch below
// runs the NT = 1 and the NT = 2 body from ONE kernel's allocation): bop_lds = LDS byte address of the B-operand images
// [x | relu(t)][ray tile][16 stages x (hi, mid) x 1 KiB], pts = the tiles' 16 x 3 point coordinates per ray (row stride 49:
// conflict-free column reads), red = [4 waves][NT * 32 rays][3] partial tail sums.
template <bool POSE, bool SAVE, int NT, bool MID>
__device__ __forceinline__ void cf_fwd_body(const CfFwdArgs& a, const int64_t tile0, const unsigned bop_lds, float (*pts)[49],
                                            float (*red)[3]) {
    if (__builtin_nontemporal_load(a.status) != 0u) return;  // these weights left fp16's range before: the bf16x3 kernel behind
#ifdef FC_SKEW  // diagnostic builds: the second workgroup of every CU starts FC_SKEW x ~4 us late (which neighbour phase arms the fault?)
    if (blockIdx.x >= 256)
        for (int i = 0; i < FC_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_tiles = (a.N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    // ray tiles of this workgroup; one past the end (odd tile count, NT = 2) recomputes the last live tile: identical values
    // to identical addresses, nothing in the chain is conditional
    int64_t tile[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        tile[rt] = tile0 + rt;
        if (tile[rt] > n_tiles - 1) tile[rt] = n_tiles - 1;
    }
    const int64_t Np = R2L_PAD_ROWS(a.N);
    const int64_t slot = R2L_TRIO_SLOT(Np);

    // ---- the tiles' sample points: thread (ray = tid & 31, group = tid >> 5) evaluates coordinates 6 grp .. 6 grp + 5 ----------
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int pr = threadIdx.x & 31, grp = threadIdx.x >> 5;
        const int64_t ray = tile[rt] * R2L_TILE_RAYS + pr;
        const int64_t rc = ray < a.N ? ray : a.N - 1;
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
                float dxr = dx * a.c2w[4 * k + 0];
                r2l_no_pack(dxr);  // (no packed multiply + swizzled add for the pair of products: r2l_common.h)
                d[k] = (dxr + dy * a.c2w[4 * k + 1]) + (-1.0f) * a.c2w[4 * k + 2];
                o[k] = a.c2w[4 * k + 3];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int smp = 2 * grp + s2;
            float z = a.ztab[smp];
            if (a.t_rand != nullptr) z = z + a.ztab[16 + smp] * a.t_rand[rc * 16 + smp];
#pragma unroll
            for (int k = 0; k < 3; ++k) pts[rt * 32 + pr][3 * smp + k] = o[k] + d[k] * z;  // fl(o + fl(d*z)): -ffp-contract=off
        }
    }

    // ---- weight ring: stages 0 .. 3 requested -------------------------------------------------------------------------------
    constexpr int R = FcRingOf<NT>::value;
    FcRing<R> W;
    FcStream P;
    {
        const unsigned long long sa = (unsigned long long)a.stream;
        P.rs = u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sa),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sa >> 32)) & 0xffffu, 0xffffffffu, 0x00020000u};
        P.voff = (unsigned)lane * 16u + (unsigned)wave * 2048u;
#ifdef FC_LATE_MODE
        P.late = blockIdx.x >= 256;
#endif
        P.g = 0u;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) fc_issue(W.a[k], P);
    f16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (_Float16)((h == 0 && k < 2) ? 1.0f : 0.0f);

    constexpr unsigned KIND = NT * FC_BOP_BYTES;                                        // x images -> relu(t) images
    const unsigned bop_rd = bop_lds + (unsigned)lane * 16u;                            // + kind*KIND + rt*32768 + kb*2048 (+1024)
    const unsigned bop_wr = bop_lds + (unsigned)lane * 16u + (unsigned)wave * 8192u;    // stage 4w of x image 0
    float amax = 0.f;
    __syncthreads();  // pts complete (nothing of the ring is compiler-tracked, so no vmcnt drain here)

    // ---- head: 63 stages of positional-encoding B values, produced four chunks of 16 stages at a time ---------------------------
    // stage q + 1 (q = 0..62) = chunk q / 16, position kb = q % 16, produced by wave kb / 4.  Values 8q .. 8q+7 of half h:
    // v < 480: coordinate ci = v / 20 of the half's 24, frequency (v % 20) / 2, (sin, cos) alternating; else identity 24h + (v - 480)
    auto produce_pe = [&](int c) {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 16 * c + 4 * wave + i;
                if (q > 62) continue;  // (wave-uniform)
                float v8[8];
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const int v = 8 * q + 2 * p2;
                    if (v < 480) {
                        const int ci = v / 20, f = (v - 20 * ci) >> 1;
                        // (odd pairs: frequency f of the coordinate whose frequency f - 1 is the pair before: angle doubling, as
                        // F3Trig2 of the one-wave-per-tile kernels — same values, same instruction count: r2l_common.h)
                        if ((p2 & 1) && f > 0) {
                            r2l_sincos_double(v8[2 * p2 - 2], v8[2 * p2 - 1], v8[2 * p2], v8[2 * p2 + 1]);
                        } else {
                            const float x = pts[rt * 32 + j][24 * h + ci];
                            r2l_sincos(x * (float)(1 << f), v8[2 * p2], v8[2 * p2 + 1]);
                        }
                    } else {
                        const int e = v - 480;
                        v8[2 * p2] = pts[rt * 32 + j][24 * h + e];
                        v8[2 * p2 + 1] = pts[rt * 32 + j][24 * h + e + 1];
                    }
                }
                u32x4 uh, um;
                fc_split8(v8, uh, um, amax);
                const unsigned wa = bop_wr + (unsigned)(c & 1) * KIND + (unsigned)rt * FC_BOP_BYTES + (unsigned)i * 2048u;
                fc_lds_write(wa, uh);
                fc_lds_write(wa + 1024u, um);
            }
    };
    f32x16 x[NT][2], t[NT][2], x0[NT][2];
    // stage q + 1 = 16 c + kb + 1 sits in ring slot (kb + 1) % R (R divides 16)
    auto head_stage = [&](auto kb_tag, int c, unsigned rb) {
        constexpr int KB = decltype(kb_tag)::value;
        if (c == 3 && KB == 15) return;
        f16x8 bh[NT], bm[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            bh[rt] = __builtin_bit_cast(f16x8, fc_lds_read(rb + (unsigned)rt * FC_BOP_BYTES + (unsigned)KB * 2048u));
            bm[rt] = __builtin_bit_cast(f16x8, fc_lds_read(rb + (unsigned)rt * FC_BOP_BYTES + (unsigned)KB * 2048u + 1024u));
        }
        fc_stage<(KB + 1) % R, false, false>(x, W, P, bh, bm);
    };
    produce_pe(0);
    fc_barrier();
    {
        f16x8 o1[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) o1[rt] = ones;
        fc_stage<0, true, true>(x, W, P, o1, o1);  // head bias
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < 3) produce_pe(c + 1);  // into the other images (their last readers passed the barrier that closed chunk c - 1)
        const unsigned rb = bop_rd + (unsigned)(c & 1) * KIND;
        head_stage(std::integral_constant<int, 0>{}, c, rb);   head_stage(std::integral_constant<int, 1>{}, c, rb);
        head_stage(std::integral_constant<int, 2>{}, c, rb);   head_stage(std::integral_constant<int, 3>{}, c, rb);
        head_stage(std::integral_constant<int, 4>{}, c, rb);   head_stage(std::integral_constant<int, 5>{}, c, rb);
        head_stage(std::integral_constant<int, 6>{}, c, rb);   head_stage(std::integral_constant<int, 7>{}, c, rb);
        head_stage(std::integral_constant<int, 8>{}, c, rb);   head_stage(std::integral_constant<int, 9>{}, c, rb);
        head_stage(std::integral_constant<int, 10>{}, c, rb);  head_stage(std::integral_constant<int, 11>{}, c, rb);
        head_stage(std::integral_constant<int, 12>{}, c, rb);  head_stage(std::integral_constant<int, 13>{}, c, rb);
        head_stage(std::integral_constant<int, 14>{}, c, rb);  head_stage(std::integral_constant<int, 15>{}, c, rb);
        if (c < 3) fc_barrier();
    }
    // the head ran on unscaled weights (r2l_f2.h range control): the chain continues on X_0 / act_s (exact: a power of two)
    // (uniform words, kept in SGPRs: as VGPR values hipcc pairs them with a neighbour and broadcasts with the packed-multiply
    // op_sel form the ISA audit refuses, r2l_amd/build.py)
    const float act_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f2_act_inv(a.status))));
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                x[rt][tt][c] = fmaxf(x[rt][tt][c], 0.f) * act_inv;  // X_0 = relu(head) / act_s
                x0[rt][tt][c] = x[rt][tt][c];
            }

    // ---- body: x B operands live in the kind-0 images, relu(t) in the kind-1 images; a barrier after each production ----------------
    u32x4* hx[NT];
    u32x4* ht[NT];
    unsigned* mwp[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        hx[rt] = SAVE ? reinterpret_cast<u32x4*>(a.save_x) + tile[rt] * R2L_H16_TILE_UNITS + lane + 256 * wave : nullptr;
        ht[rt] = SAVE ? reinterpret_cast<u32x4*>(a.save_t) + tile[rt] * R2L_H16_TILE_UNITS + lane + 256 * wave : nullptr;
        mwp[rt] = SAVE ? reinterpret_cast<unsigned*>(a.save_t + R2L_MASK_OFFSET(Np) + tile[rt] * 256 + lane * 4) + wave : nullptr;
    }
    // the activation scale this stream was packed for (r2l_f2.h range control): what the chain holds is x / act_s
    const float act_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f2_act_scale(a.status))));
    if (SAVE && blockIdx.x == 0 && threadIdx.x == 0) {  // stash format word: fp16 stage pieces (a fallback launch overwrites it),
        reinterpret_cast<unsigned*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np)] = 0u;  // and the scale of the stashed x, relu(t)
        reinterpret_cast<float*>(a.save_x)[R2L_STASH_FMT_WORD(a.n_block, Np) + 1] = act_s;
    }
    // (the kind-0 images were last read in chunk 2 of the head, two barriers ago)
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) fc_produce<false, SAVE, false, FcIdentity, MID>(x[rt], bop_wr + (unsigned)rt * FC_BOP_BYTES, hx[rt], nullptr, amax, FcIdentity(), a.mid_units);
    fc_barrier();
    auto block = [&](auto ph_tag, bool last) {
        constexpr int PH = decltype(ph_tag)::value;
        // t = W1 x + b1
        fc_layer<PH, true>(t, W, P, bop_rd, ones);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            unsigned mw = 0u;
            fc_produce<true, SAVE, SAVE, FcIdentity, MID>(t[rt], bop_wr + KIND + (unsigned)rt * FC_BOP_BYTES, ht[rt], &mw, amax, FcIdentity(), a.mid_units);
            if (SAVE) fc_store_b32(mwp[rt], mw);
        }
        fc_barrier();
        // x += W2 relu(t) + b2
        fc_layer<(PH + 1) % R, false>(x, W, P, bop_rd + KIND, ones);
        if (SAVE) {
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) {
                hx[rt] += slot / 4;
                ht[rt] += slot / 4;
                mwp[rt] += slot;
            }
        }
        if (!last) {
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) fc_produce<false, SAVE, false, FcIdentity, MID>(x[rt], bop_wr + (unsigned)rt * FC_BOP_BYTES, hx[rt], nullptr, amax, FcIdentity(), a.mid_units);
            fc_barrier();
        }
    };
    // block b starts in phase 2 b mod R: unrolled over R / 2 blocks
#pragma unroll 1
    for (int b = 0; b < a.n_block; b += R / 2) {
        block(std::integral_constant<int, 0>{}, b == a.n_block - 1);
        if constexpr (R >= 4) {
            if (b + 1 < a.n_block) block(std::integral_constant<int, 2 % R>{}, b + 1 == a.n_block - 1);
        }
        if constexpr (R >= 8) {
            if (b + 2 < a.n_block) block(std::integral_constant<int, 4 % R>{}, b + 2 == a.n_block - 1);
            if (b + 3 < a.n_block) block(std::integral_constant<int, 6 % R>{}, b + 3 == a.n_block - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead loads (stream padding) and the stash stores

    f2_report_amax(a.status, amax, lane);  // range control (r2l_f2.h): AMAX, and FLAG if this launch belongs to the bf16x3 kernel

    // ---- tail: rgb = sigmoid(Wt (x + X_0) + bt): per-wave partial dot products over its 64 features, summed through LDS ----------
    const float* tw = a.params + cf_off_tail_w(a.n_block) + 4 * h;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int64_t ray = tile[rt] * R2L_TILE_RAYS + j;
        float p3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int T = 2 * wave + tt;
                f32x4 wv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) wv[c] = *reinterpret_cast<const f32x4*>(tw + c * R2L_W + 32 * T + 8 * q);
                f32x4 yv;
#if defined(FC_TAIL_ASM) && FC_TAIL_ASM > 0
                // diagnostic builds (tools/coopf_forensics.py, tools/build_variant.sh): the (R, G) accumulation as hand-placed
                // v_mov / v_pk_fma_f32 sequences in the form hipcc used to generate here, one thing varied per build (DESIGN.md §2,
                // co-residency fault).  0: no asm, the compiler's own packed chain (the round-2 kernel)
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[e] = x[rt][tt][4 * q + e] + x0[rt][tt][4 * q + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    typedef float fc_f32x2 __attribute__((ext_vector_type(2)));
                    fc_f32x2 acc = {p3[0], p3[1]};
#if FC_TAIL_ASM == 4
                    const fc_f32x2 yp = {yv[e], yv[e]};
                    const bool odd_form = false;
#else
                    const fc_f32x2 yp = {yv[e & ~1], yv[e | 1]};
                    const bool odd_form = e & 1;
#endif
#if FC_TAIL_ASM == 5  // nothing of this wave in flight in the vector-memory pipe while the packed FMAs run
#define FC_MOVS "s_waitcnt vmcnt(0)\n\tv_mov_b32 v251, %2\n\tv_mov_b32 v250, %1\n\t"
#elif FC_TAIL_ASM == 3
#define FC_MOVS "v_mov_b32 v250, %1\n\tv_mov_b32 v251, %2\n\t"
#elif FC_TAIL_ASM == 2
#define FC_MOVS "v_mov_b32 v251, %2\n\tv_mov_b32 v250, %1\n\ts_nop 0\n\t"
#else
#define FC_MOVS "v_mov_b32 v251, %2\n\tv_mov_b32 v250, %1\n\t"
#endif
                    if (odd_form)
                        asm volatile(FC_MOVS "v_pk_fma_f32 %0, v[250:251], %3, %0 op_sel:[0,1,0]"
                                     : "+v"(acc) : "v"(wv[0][e]), "v"(wv[1][e]), "v"(yp) : "v250", "v251");
                    else
                        asm volatile(FC_MOVS "v_pk_fma_f32 %0, v[250:251], %3, %0 op_sel_hi:[1,0,1]"
                                     : "+v"(acc) : "v"(wv[0][e]), "v"(wv[1][e]), "v"(yp) : "v250", "v251");
                    p3[0] = acc[0];
                    p3[1] = acc[1];
                    p3[2] = __builtin_fmaf(wv[2][e], yv[e], p3[2]);
                }
#else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yv[e] = x[rt][tt][4 * q + e] + x0[rt][tt][4 * q + e];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        p3[c] = __builtin_fmaf(wv[c][e], yv[e], p3[c]);
#ifndef FC_TAIL_ASM
                        r2l_no_pack(p3[c]);  // three scalar v_fmac chains: no v_pk_fma_f32 op_sel:[0,1,0] (r2l_coopf.h)
#endif
                    }
                }
#endif
                // slot n of save_x: y = x_n + x_0, row-major (the tail weight gradient reads nothing else)
                if (SAVE) *reinterpret_cast<f32x4*>(a.save_x + (int64_t)a.n_block * slot + ray * R2L_W + 32 * T + 8 * q + 4 * h) = yv * act_s;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) p3[c] += __shfl_xor(p3[c], 32);
        if (h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) red[wave * (NT * 32) + rt * 32 + j][c] = p3[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const int64_t ray = tile[rt] * R2L_TILE_RAYS + j;
        if (wave == 0 && h == 0 && ray < a.N) {
            const int rj = rt * 32 + j;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = ((red[rj][c] + red[NT * 32 + rj][c]) + (red[2 * NT * 32 + rj][c] + red[3 * NT * 32 + rj][c])) * act_s +
                                a.params[cf_off_tail_b(a.n_block) + c];
                a.rgb[ray * 3 + c] = 1.0f / (1.0f + expf(-v));
            }
        }
    }
}

template <bool POSE, bool SAVE, int NT, bool MID = false>
__global__ __launch_bounds__(256, NT == 1 ? 2 : 1) void r2l_coopf_fwd_kernel(const CfFwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char bop[2][NT][FC_BOP_BYTES];
    __shared__ float pts[NT * 32][49];
    __shared__ float red[4 * NT * 32][3];
    cf_fwd_body<POSE, SAVE, NT, MID>(a, (int64_t)blockIdx.x * NT,
                                     (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&bop[0][0][0], pts, red);
}

// MIXED launch (tile counts between one and two per CU, n_cu < tiles <= 2 n_cu: the per-GPU share of the README's 98 304-ray
// step at 8 GPUs is 12 288 rays = 384 tiles): ONE grid of n_cu workgroups, the first a.n_two of them with two ray tiles
// (tiles 2b, 2b + 1), the others with one (tile 2 n_two + (b - n_two)) — instead of ceil(tiles / 2) two-tile workgroups that
// leave n_cu - tiles / 2 CUs idle for the whole chain.  Each tile takes exactly the path it takes in the NT = 1 / NT = 2 kernels
// (same streams, stash, summation order): results are bit-identical to both.  The NT = 2 allocation (143 KiB) keeps it to one
// workgroup per CU (r2l_coopf.h: the one-tile body must not share a SIMD with a second wave).
template <bool SAVE, bool MID>
__global__ __launch_bounds__(256, 1) void r2l_coopf_fwd_mixed_kernel(const CfFwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char bop[2][2][FC_BOP_BYTES];
    __shared__ float pts[2 * 32][49];
    __shared__ float red[4 * 2 * 32][3];
    const unsigned bop_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)&bop[0][0][0];
    const int b = fc_mixed_index(a.xcd_major);
    if (b < a.n_two) cf_fwd_body<false, SAVE, 2, MID>(a, (int64_t)2 * b, bop_lds, pts, red);
    else cf_fwd_body<false, SAVE, 1, MID>(a, (int64_t)a.n_two + b, bop_lds, pts, red);  // 2 n_two + (b - n_two)
}

int r2l_coopf_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab, const float* c2w_host12,
                      int H, int W, float focal, const float* wstream2, const float* params, int n_block, float* rgb,
                      float* save_x, float* save_t, int64_t N, hipStream_t stream) {
    CfFwdArgs a{};
    a.rays_o = rays_o; a.rays_d = rays_d; a.t_rand = t_rand; a.ztab = ztab;
    a.stream = reinterpret_cast<const unsigned char*>(wstream2); a.params = params;
    a.status = reinterpret_cast<unsigned*>(const_cast<float*>(wstream2) + r2l_fwd2_status_offset(n_block));
    a.n_block = n_block; a.rgb = rgb; a.save_x = save_x; a.save_t = save_t; a.N = N; a.H = H; a.Wimg = W; a.focal = focal;
    if (c2w_host12) for (int i = 0; i < 12; ++i) a.c2w[i] = c2w_host12[i];
    a.mid_units = (save_x != nullptr && r2l_dw_exact()) ? R2L_H16_MID_BYTES(R2L_PAD_ROWS(N)) / 16 : 0;
    const int64_t tiles = (N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS;
    // up to one workgroup per CU: one ray tile each; beyond, two tiles per workgroup share every weight load; between one and
    // two tiles per CU: the mixed grid (explicit rays only: the single-pose render launches are far above this band)
    if (const int n_two = r2l_coopf_mixed_two(tiles); n_two > 0 && !c2w_host12) {
        a.n_two = n_two;
        a.xcd_major = r2l_coopf_mixed_xcd_major();
        const dim3 grid((unsigned)(tiles - n_two)), block(256);
        if (save_x && a.mid_units != 0) hipLaunchKernelGGL((r2l_coopf_fwd_mixed_kernel<true, true>), grid, block, 0, stream, a);
        else if (save_x) hipLaunchKernelGGL((r2l_coopf_fwd_mixed_kernel<true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((r2l_coopf_fwd_mixed_kernel<false, false>), grid, block, 0, stream, a);
        R2L_CHECK(hipGetLastError());
        return 0;
    }
    const bool two = r2l_coopf_two_tiles(tiles);
    const dim3 grid((unsigned)(two ? (tiles + 1) / 2 : tiles)), block(256);
    static int solo_ok[4] = {0, 0, 0, 0};  // one-tile kernels: at most one workgroup per CU, verified before the first launch
    if (c2w_host12) {
        if (two) hipLaunchKernelGGL((r2l_coopf_fwd_kernel<true, false, 2>), grid, block, 0, stream, a);
        else {
            if (int e = fc_check_solo(r2l_coopf_fwd_kernel<true, false, 1>, "r2l_coopf_fwd_kernel<pose>", &solo_ok[0])) return e;
            hipLaunchKernelGGL((r2l_coopf_fwd_kernel<true, false, 1>), grid, block, FC_SOLO_LDS_BYTES, stream, a);
        }
    } else if (save_x && a.mid_units != 0) {  // exact weight gradients: the mid halves are stashed too
        if (two) hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, true, 2, true>), grid, block, 0, stream, a);
        else {
            if (int e = fc_check_solo(r2l_coopf_fwd_kernel<false, true, 1, true>, "r2l_coopf_fwd_kernel<save, mid>", &solo_ok[3])) return e;
            hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, true, 1, true>), grid, block, FC_SOLO_LDS_BYTES, stream, a);
        }
    } else if (save_x) {
        if (two) hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, true, 2>), grid, block, 0, stream, a);
        else {
            if (int e = fc_check_solo(r2l_coopf_fwd_kernel<false, true, 1>, "r2l_coopf_fwd_kernel<save>", &solo_ok[1])) return e;
            hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, true, 1>), grid, block, FC_SOLO_LDS_BYTES, stream, a);
        }
    } else {
        if (two) hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, false, 2>), grid, block, 0, stream, a);
        else {
            if (int e = fc_check_solo(r2l_coopf_fwd_kernel<false, false, 1>, "r2l_coopf_fwd_kernel<rays>", &solo_ok[2])) return e;
            hipLaunchKernelGGL((r2l_coopf_fwd_kernel<false, false, 1>), grid, block, FC_SOLO_LDS_BYTES, stream, a);
        }
    }
    R2L_CHECK(hipGetLastError());
    return 0;
}

// which chain kernels a fp16-trio launch of N rays takes: 0 = one wave per tile (r2l_fwd2 / r2l_bwd2), 1 / 2 = cooperative with
// that many ray tiles per workgroup (host-side decision, no device work; include/r2l_hip.h)
extern "C" int r2l_coop_tiles_for_cfg(int64_t N, int n_block, const r2l_config* cfg) {
    R2L_CFG_QUERY(cfg);
    if (!r2l_use_coopf(N, n_block)) return 0;
    return r2l_coopf_policy((N + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS);
}
extern "C" int r2l_coop_tiles_for(int64_t N, int n_block) { return r2l_coop_tiles_for_cfg(N, n_block, nullptr); }
