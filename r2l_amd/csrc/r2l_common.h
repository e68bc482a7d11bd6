// r2l_common.h — shared device-side building blocks for the R2L ResMLP chain kernels (gfx950 only).
//
// Register-resident activation chain
// ----------------------------------
// One wavefront owns a tile of 32 rays for the WHOLE network.  A [256 features x 32 rays] activation
// is held as 8 x f32x16 = 128 VGPR/AGPRs in the C/D fragment layout of v_mfma_f32_32x32x2_f32:
//
//     reg (T, c) of lane l  <->  feature 32*T + 8*(c>>2) + 4*(l>>5) + (c&3),   ray (l & 31)
//
// Computing  out^T[256 x 32] = W[256 x 256] . in^T[256 x 32]  with the weights as the MFMA A operand
// makes the output fragment of layer n directly usable as the B operand of layer n+1 (B wants
// B[k = l>>5][j = l&31]: register (T,c) supplies k-pair {f_lo, f_lo+4}), so activations never leave
// the register file: no LDS round trip, no transposes, no barriers.  The weights are re-packed on the
// device (r2l_pack.hip) into the exact per-lane A-operand order, as ONE contiguous stream in
// consumption order, so every wave streams them with fully coalesced 1 KiB (16 B per lane) buffer loads.
#pragma once
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "r2l_hip.h"  // r2l_config

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define R2L_W 256             // network width (features)
#define R2L_NT 8              // 32-feature row tiles per activation
#define R2L_TILE_RAYS 32      // rays per wavefront tile
#define R2L_NSAMPLE 16        // sample points per ray
#define R2L_L 10              // positional-encoding frequencies
#define R2L_IN (R2L_NSAMPLE * 3 * (2 * R2L_L + 1))  // 1008
#define R2L_GROUP_FLOATS (R2L_NT * 64 * 4)           // one weight "group": 8 tiles x 64 lanes x float4 = 2048 floats
#define R2L_LAYER_GROUPS 32                          // groups per 256x256 layer
#define R2L_LAYER_FLOATS (R2L_LAYER_GROUPS * R2L_GROUP_FLOATS)  // 65536
#define R2L_HEAD_TRIG_GROUPS 120                     // 8 samples x 3 axes x 5 groups of 4 trig features
#define R2L_HEAD_ID_GROUPS 6                         // 24 identity features per half-wave
#define R2L_HEAD_GROUPS (R2L_HEAD_TRIG_GROUPS + R2L_HEAD_ID_GROUPS)
#define R2L_HEAD_FLOATS (R2L_HEAD_GROUPS * R2L_GROUP_FLOATS)    // 258048 = 1008*256
// forward stream: every layer is preceded by ONE bias group (component 0 of each float4 = bias of the tile row, for the
// k=0 half-wave only); the kernel turns it into 8 MFMAs against the constant B operand [1, 0], so the accumulators are
// initialised by the matrix pipe (C = 0 inline) instead of 32 bias loads + 128 register writes per layer.
#define R2L_FWD_HEAD_GROUPS (1 + R2L_HEAD_GROUPS)
#define R2L_FWD_LAYER_GROUPS (1 + R2L_LAYER_GROUPS)
#define R2L_PAD_ROWS(n) ((((int64_t)(n)) + R2L_TILE_RAYS - 1) / R2L_TILE_RAYS * R2L_TILE_RAYS)
#define R2L_STREAM_PAD (10 * R2L_GROUP_FLOATS)       // prefetchers run up to 9 groups past the end (r2l_coop.hip ring)

// ---------------------------------------------------------------------------------------------
// Weight stream reader.  A wave walks the packed stream one group (8 tiles x float4 per lane) at a time through a ring
// of D register buffers: a group is consumed tile-major (4 MFMAs on tile t's accumulator, then tile t+1, ...) and the
// registers of tile t are reloaded with tile t of the group D positions ahead right after its 4 MFMAs have issued.
// Every load is therefore issued 32*D - 4 MFMAs before its first use (D=1: 1792 cycles, D=2: 3840 cycles).  D = 2 is
// used by the student chains: vmcnt retires in order, so the (slow, HBM-latency) stash stores / mask loads that ride
// along in training would otherwise stall the weight loads queued behind them.  The ring slot of every group is a
// compile-time constant (SLOT), which is why loops over groups are arranged to have even trip lengths.
// ---------------------------------------------------------------------------------------------
// The stream is read through a BUFFER DESCRIPTOR (4 SGPRs) + one constant per-lane byte offset (VGPR) + a wave-uniform
// scalar offset: `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen offset:imm`.  Measured on gfx950: the 64-bit
// VGPR-address form (`global_load_dwordx4 v, v[addr:addr+1], off`) costs ~16 cycles of MFMA issue per load, this
// form none — the largest single lever of round 1.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct WPtr {
    __amdgpu_buffer_rsrc_t rsrc;  // buffer descriptor of the whole stream (4 SGPRs, wave-uniform by construction)
    unsigned voff;                // lane * 16
    unsigned soff;                // wave-uniform byte position of the next group to LOAD
    // tile t of the next group (i = t * 64).  The constant part must land in the instruction's 12-bit offset field
    // (tiles 4..7 go through soffset + 4096); `opaque()` keeps hipcc from hoisting voff + const into 8 loop-invariant
    // VGPRs instead.
    __device__ __forceinline__ f32x4 operator[](int i) const {
        const unsigned byte = (unsigned)i * 16u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (byte & 4095u), soff + (byte & ~4095u), 0);
        return __builtin_bit_cast(f32x4, v);
    }
    __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(voff)); }
    __device__ __forceinline__ WPtr& operator+=(int n) {  // advance by n float4 slots (n = 512: one group)
        soff += (unsigned)n * 16u;
        return *this;
    }
};

template <int D>
struct WRingT {
    WPtr p;  // the next group to LOAD
    f32x4 w[D][R2L_NT];
    __device__ __forceinline__ void init(const float* stream, int lane) {
        // 0x00020000: raw buffer, 32-bit offsets (cdna_hip_programming.md T8); num_records = 4 GiB - 1 (no clamping)
        p.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, 0xffffffff, 0x00020000);
        p.voff = (unsigned)lane * 16u;
        p.soff = 0u;
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int t = 0; t < R2L_NT; ++t) w[d][t] = p[t * 64];
            p += R2L_NT * 64;
        }
    }
};
typedef WRingT<1> WStream;
typedef WRingT<2> WRing2;

// acc[256x32] += W_group . b  for the four k-pairs of the group in ring slot SLOT (b0..b3 = B-operand registers), and
// start streaming the group D positions ahead into the same registers.  Schedule pinned: [4 MFMA, 1 VMEM read
// (, VPT VALU)] x 8 — hipcc otherwise sinks the prefetch loads next to their use and the wave eats the L2 latency with
// nothing else resident on the SIMD to hide it.  EXTRA_RD / EXTRA_WR: VMEM reads / writes a hook issued just before
// (mask prefetch, stash store) go FIRST.  VPT: non-MFMA VALU instructions to slot in after each tile (used by the
// head to hide the next coordinate's sin/cos evaluation under the MFMAs).
template <int SLOT = 0, int EXTRA_RD = 0, int EXTRA_WR = 0, int VPT = 0, bool ZERO_C = false, int D>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[R2L_NT], WRingT<D>& ws, float b0, float b1, float b2, float b3) {
    static_assert(SLOT >= 0 && SLOT < D, "ring slot");
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ws.p.opaque();
#pragma unroll
    for (int t = 0; t < R2L_NT; ++t) {
        // ZERO_C: first group of a GEMM without bias: the accumulator is initialised by C = 0 (inline constant)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][t][0], b0, ZERO_C ? zero : acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][t][1], b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][t][2], b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][t][3], b3, acc[t], 0, 0, 0);
        ws.w[SLOT][t] = ws.p[t * 64];
    }
    ws.p += R2L_NT * 64;
    if (EXTRA_WR > 0) __builtin_amdgcn_sched_group_barrier(0x040, EXTRA_WR, 0);
    if (EXTRA_RD > 0) __builtin_amdgcn_sched_group_barrier(0x020, EXTRA_RD, 0);
#pragma unroll
    for (int i = 0; i < R2L_NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
        if (VPT > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Same for a layer with 4 output tiles: one stream load-group carries two k-groups (slots 0-3: k-group A, 4-7: B).
template <int SLOT = 0, int D>
__device__ __forceinline__ void mfma_group4x2(f32x16 (&acc)[4], WRingT<D>& ws, const float (&ba)[4], const float (&bb)[4]) {
    ws.p.opaque();
#pragma unroll
    for (int s8 = 0; s8 < R2L_NT; ++s8) {
        const int tt = s8 & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][s8][j], s8 < 4 ? ba[j] : bb[j], acc[tt], 0, 0, 0);
        ws.w[SLOT][s8] = ws.p[s8 * 64];
    }
    ws.p += R2L_NT * 64;
#pragma unroll
    for (int i = 0; i < R2L_NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

struct IdentityAct {
    __device__ __forceinline__ float operator()(float v, int, int) const { return v; }
};
struct ReluAct {
    __device__ __forceinline__ float operator()(float v, int, int) const { return fmaxf(v, 0.f); }
};
// ---- per-group hooks: memory traffic that rides along a GEMM instead of bursting between GEMMs ---------------------
struct NoHook {
    static constexpr int RD = 0, WR = 0;
    __device__ __forceinline__ void at(int) {}
};

// Stores the fragment that is the GEMM's B operand (so its registers stay live and unmodified for the whole GEMM) to a
// row-major [N][256] tensor, one 16-byte piece per group: lane (ray j, half h) writes row*1 KiB + (32T + 8q + 4h)*4.
// Stash tensors have r2l_padded_rows(N) = ceil(N/32)*32 rows per slot, so the lanes of a ragged last tile store to
// their own padding rows: no predicate, no branch inside the GEMM.
#ifndef R2L_HOOK_BUFFER
#define R2L_HOOK_BUFFER 0  // ride-along stores / mask loads: 64-bit pointers (0) or buffer descriptors (1); same-box A/B: 0 is 0.3 % faster
#endif
// Stash / gradient stores.  Non-temporal stores (`global_store ... nt`) are acknowledged sooner, which matters because
// stores retire through the same in-order vmcnt queue as the weight prefetches — but only for stores that cover whole
// cache lines: the 32- / 64-byte fragment pieces of the chain kernels rely on L2 to merge partial lines (same-box A/B
// with nt on those: 98 304-ray step 25.2 -> 34 ms).  Whole-row stores (r2l_coop16.hip): 462 -> 446 us with nt.
#ifdef R2L_STASH_ST_AUX  // A/B builds (tools/mkvar.sh ... -DR2L_STASH_ST_AUX=1|2|3|4): cache policy of the fragment-piece stores
#if R2L_STASH_ST_AUX == 1
#define R2L_STASH_ST_POLICY "sc1"
#elif R2L_STASH_ST_AUX == 2
#define R2L_STASH_ST_POLICY "nt"
#elif R2L_STASH_ST_AUX == 3
#define R2L_STASH_ST_POLICY "sc1 nt"
#else
#define R2L_STASH_ST_POLICY "sc0 sc1"
#endif
__device__ __forceinline__ void r2l_stash_store(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off " R2L_STASH_ST_POLICY : : "v"(p), "v"(v) : "memory");
}
#elif defined(R2L_TIMING_NO_STASH_STORE)  // timing builds only: what the ride-along stores cost (results are wrong)
__device__ __forceinline__ void r2l_stash_store(float* p, const f32x4& v) {
    if (__builtin_expect(v[0] == 1.2345e-33f, 0)) *reinterpret_cast<f32x4*>(p) = v;
}
#else
__device__ __forceinline__ void r2l_stash_store(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
#endif
__device__ __forceinline__ void r2l_stash_store_nt(float* p, const f32x4& v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}
// ---- chunked stash layout (the bf16x3 training trio r2l_fwd3 / r2l_bwd3 / r2l_dw_body3c) ----------------------------------
// A [Np,256] fp32 slot is stored as [tile = ray/32][chunk = feature/8][ray%32][feature%8]: float index
//     tile*8192 + chunk*256 + (ray%32)*8 + feature%8 .
// The one-wave-per-tile chains hold, per lane (j, h) and fragment piece pi = 4T + c0/4, the features 8*pi + 4h .. +3 of ray j:
// the 64 lanes of one piece store 64 x 16 B = one contiguous KiB (row-major, the same store scatters 32-byte fragments over
// 32 rows and every 128-byte line is written by four different instructions).  The weight-gradient kernel loads the half
// tile of a chunk (16 rays x 32 B) as one 512-byte run.
#define R2L_CHUNK_PIECE 256   // floats per (tile, chunk) piece
#define R2L_CHUNK_TILE 8192   // floats per tile
// A slot of the trio's buffers is Np*256 floats of chunked data followed by Np*8 floats of ReLU mask words (save_t slots:
// per tile 64 lanes x 4 words; bit (T&1)*16 + c of word T>>1 of lane (j, h) = [t > 0] for the lane's fragment register (T, c),
// written by the forward chain, read back by the dX chain with ONE 1 KiB LDS-DMA piece per block)
#define R2L_TRIO_SLOT(Np) ((int64_t)(Np) * (R2L_W + 8))
#define R2L_MASK_OFFSET(Np) ((int64_t)(Np) * R2L_W)
// slot n_block of save_x (y = x_n + x_0, row-major) has no mask words: the first word of that area tells the backward which
// format the forward left in the other slots: 0 = fp16 stage pieces (r2l_f2.h: the default fp16 trio), 1 = chunked fp32 (the
// bf16x3 forward, also when it ran as the range-guard fallback of the fp16 one)
#define R2L_STASH_FMT_WORD(n_block, Np) ((int64_t)(n_block) * R2L_TRIO_SLOT(Np) + R2L_MASK_OFFSET(Np))
__device__ __forceinline__ int64_t r2l_chunk_lane(int64_t tile, int j, int h) {
    return tile * R2L_CHUNK_TILE + j * 8 + 4 * h;
}
__device__ __forceinline__ void r2l_chunk_store(float* p, const f32x4& v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}
template <bool RELU = false, class Act = IdentityAct>
struct StoreHookT {
    static constexpr int RD = 0, WR = 1;
#if R2L_HOOK_BUFFER
    __amdgpu_buffer_rsrc_t rsrc;  // descriptor of the destination slot (wave-uniform base)
    unsigned voff;                // ray*1024 + 16*h (per lane)
#else
    float* row;  // base + ray*256 + 4*h (per lane)
#endif
    const f32x16 (&src)[R2L_NT];
    Act act;  // applied to the stored values (e.g. the backward's ReLU mask), like the consumer GEMM applies it
    __device__ __forceinline__ StoreHookT(float* base, int64_t ray, int h, const f32x16 (&s)[R2L_NT], Act a = Act())
#if R2L_HOOK_BUFFER
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffff, 0x00020000)),
          voff((unsigned)(ray * (R2L_W * 4) + 16 * h)), src(s), act(a) {}
#else
        : row(base + ray * R2L_W + 4 * h), src(s), act(a) {}
#endif
    __device__ __forceinline__ void at(int G) {
        const int T = G >> 2, q = (G & 3) * 4;
        f32x4 v = {act(src[T][q + 0], T, q + 0), act(src[T][q + 1], T, q + 1), act(src[T][q + 2], T, q + 2),
                   act(src[T][q + 3], T, q + 3)};
        if (RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
#if R2L_HOOK_BUFFER
        asm volatile("" : "+v"(voff));  // keep voff + const out of loop-invariant VGPRs: it folds into offset:imm
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                               voff + (unsigned)(32 * T + 8 * (G & 3)) * 4u, 0, 0);
#else
        r2l_stash_store(row + 32 * T + 8 * (G & 3), v);
#endif
    }
};
typedef StoreHookT<false> StoreHook;

// ReLU mask words of the fp32 one-wave-per-tile training chains (round 6).  The forward (r2l_fwd_kernel<*, SAVE>) leaves, per
// block and 32-ray tile, 64 lanes x 4 words = 1 KiB behind the n_block dense [Np][256] slots of save_t (a slot is
// R2L_TRIO_SLOT(Np) = Np * 264 floats for every caller, include/r2l_hip.h: the row-major family used Np * 256 of them): bit
// 31 - ((T & 1) * 16 + c) of word T >> 1 of lane (j, h) = [t > 0] for the lane's fragment register (T, c).  The dX chain
// (r2l_bwd_chain_kernel) reads ONE 16-byte piece per lane and block instead of all of relu(t) for its signs (rounds 1 - 5:
// 32 loads per lane and block, 4.3 GB per 98 304-ray step; same-box A/B 8.21 -> 7.56 ms for the chain,
// profiles/r06_graded_step_ab.txt).
__host__ __device__ static inline int64_t r2l_mask32_offset(int n_block, int64_t Np, int b) {
    return (int64_t)n_block * Np * R2L_W + (int64_t)b * Np * 8;  // floats from save_t; + tile * 256 + lane * 4
}
#define R2L_MASK32_BIT(T, c) (31 - (((T) & 1) * 16 + (c)))
// StoreHookT<true> (relu(t) rides along GEMM 2 of a block) that also folds the signs of the pieces it stores into mb[4]: two
// VALU instructions per value — the sign bit of 0 - t is [t > 0] exactly (+-0 -> +0, t < 0 -> positive), shifted in from the
// right with v_alignbit_b32 ({w, y} >> 31 = (w << 1) | y[31]), so the value folded i-th into a word ends at bit 31 - i.
struct StoreMaskHook {
    static constexpr int RD = 0, WR = 1;
    StoreHookT<true> st;
    unsigned (&mb)[4];
    __device__ __forceinline__ StoreMaskHook(float* base, int64_t ray, int h, const f32x16 (&s)[R2L_NT], unsigned (&m)[4])
        : st(base, ray, h, s), mb(m) {}
    __device__ __forceinline__ void at(int G) {
        st.at(G);
        const int T = G >> 2, q = (G & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = 0.f - st.src[T][q + j];
            mb[G >> 3] = __builtin_amdgcn_alignbit(mb[G >> 3], __builtin_bit_cast(unsigned, y), 31);
        }
    }
};

// Consume one bias group of the forward stream: acc (+)= bias x [1,0]^T  — 8 MFMAs, one per tile.
template <bool ZERO_INIT, int SLOT = 0, int D>
__device__ __forceinline__ void mfma_bias_group(f32x16 (&acc)[R2L_NT], WRingT<D>& ws, float one_h0) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ws.p.opaque();
#pragma unroll
    for (int t = 0; t < R2L_NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws.w[SLOT][t][0], one_h0, ZERO_INIT ? zero : acc[t], 0, 0, 0);
        ws.w[SLOT][t] = ws.p[t * 64];
    }
    ws.p += R2L_NT * 64;
#pragma unroll
    for (int i = 0; i < R2L_NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// acc += W[256x256] . act(in)   (one full layer, 32 groups, 1024 MFMAs).  The activation is applied lazily to the four
// B-operand registers of each group (VALU work hidden in the MFMA shadow) instead of a 384-instruction burst between
// the GEMMs; `in` itself keeps the raw values.  Act: IdentityAct, ReluAct (forward) or a bit-mask (backward).
// BASE: ring slot of the first group.  ZERO_FIRST: acc is initialised by the first group's MFMAs (C = 0).
template <class Act, int BASE = 0, bool ZERO_FIRST = false, class Hook, int D>
__device__ __forceinline__ void gemm256a(f32x16 (&acc)[R2L_NT], const f32x16 (&in)[R2L_NT], WRingT<D>& ws, Hook& hook,
                                         const Act& act) {
    static_assert(R2L_LAYER_GROUPS % D == 0, "layer groups must be a multiple of the ring depth");
#pragma unroll
    for (int G2 = 0; G2 < R2L_LAYER_GROUPS; G2 += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            const int G = G2 + dd;
            hook.at(G);
            const int T = G >> 2, q = (G & 3) * 4;
            float b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = act(in[T][q + j], T, q + j);
            constexpr int S1 = D > 1 ? 1 : 0;
            if (G == 0 && ZERO_FIRST) {
                if ((BASE + dd) % D == 0) mfma_group<0, Hook::RD, Hook::WR, 0, true>(acc, ws, b[0], b[1], b[2], b[3]);
                else mfma_group<S1, Hook::RD, Hook::WR, 0, true>(acc, ws, b[0], b[1], b[2], b[3]);
            } else {
                if ((BASE + dd) % D == 0) mfma_group<0, Hook::RD, Hook::WR>(acc, ws, b[0], b[1], b[2], b[3]);
                else mfma_group<S1, Hook::RD, Hook::WR>(acc, ws, b[0], b[1], b[2], b[3]);
            }
        }
    }
}
template <bool RELU_IN, int BASE = 0, class Hook, int D>
__device__ __forceinline__ void gemm256x(f32x16 (&acc)[R2L_NT], const f32x16 (&in)[R2L_NT], WRingT<D>& ws, Hook& hook) {
    if (RELU_IN) gemm256a<ReluAct, BASE>(acc, in, ws, hook, ReluAct());
    else gemm256a<IdentityAct, BASE>(acc, in, ws, hook, IdentityAct());
}
__device__ __forceinline__ void relu_inplace(f32x16 (&a)[R2L_NT]) {
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int c = 0; c < 16; ++c) a[T][c] = fmaxf(a[T][c], 0.0f);
}

// Store / load a fragment to a row-major [N][256] fp32 tensor (saved activations / gradients).
// Lane (ray j, half h) writes 16 B at row*1 KiB + (32T + 8q + 4h)*4: lanes j and j+32 complete a
// 32-byte sector; the 32 (T,q) stores of one call complete every 128-byte line of the 32 rows.
__device__ __forceinline__ void store_frag(float* __restrict__ base, int64_t row, int h, const f32x16 (&a)[R2L_NT]) {
    float* r = base + row * R2L_W + 4 * h;
#pragma unroll
    for (int T = 0; T < R2L_NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {a[T][4 * q + 0], a[T][4 * q + 1], a[T][4 * q + 2], a[T][4 * q + 3]};
            r2l_stash_store(r + 32 * T + 8 * q, v);
        }
}

// ---- accurate sin & cos for |x| < ~8e3 (the encoder's arguments are 2^k * x, |x| < ~8, k <= 9) ---------------
// Cody-Waite reduction by pi/2 with a 3-term fp32 split and FMAs, then the classic degree-7/8 minimax kernels on
// [-pi/4, pi/4].  Max error vs fp64 sin/cos over |x| <= 4096: < 1.5 ulp (tests/test_sincos_gpu).  Branch free.
__device__ __forceinline__ void r2l_sincos(float x, float& s_out, float& c_out) {
    const float n = rintf(x * 0.63661977236758134f);  // round(x * 2/pi)
    float r = __builtin_fmaf(-n, 1.57079637050628662109375f, x);
    r = __builtin_fmaf(-n, -4.37113900018624283e-8f, r);
    r = __builtin_fmaf(-n, -1.71512449512872556e-15f, r);
    const float r2 = r * r;
    // sin(r) ~ r + r^3 * (S1 + r2*(S2 + r2*(S3 + r2*S4)))
    float ps = __builtin_fmaf(r2, 2.718311493989822e-6f, -1.9839334836096632e-4f);
    ps = __builtin_fmaf(ps, r2, 8.3333293858894632e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666666641626524e-1f);
    const float sr = __builtin_fmaf(ps * r2, r, r);
    // cos(r) ~ 1 - r2/2 + r2^2 * (C1 + r2*(C2 + r2*C3))
    float pc = __builtin_fmaf(r2, 2.439044879627741e-5f, -1.388676377460993e-3f);
    pc = __builtin_fmaf(pc, r2, 4.1666623323739063e-2f);
    pc = __builtin_fmaf(pc, r2, -0.5f);
    const float cr = __builtin_fmaf(pc, r2, 1.0f);
    const int q = (int)n;
    const float s1 = (q & 1) ? cr : sr;
    const float c1 = (q & 1) ? sr : cr;
    s_out = (q & 2) ? -s1 : s1;
    c_out = ((q + 1) & 2) ? -c1 : c1;
}

// (sin, cos) of 2x from (sin, cos) of x: three VALU instructions instead of the ~27 of r2l_sincos.  The encoder's frequencies are
// consecutive powers of two of the same coordinate, so every second (sin, cos) pair of the fp16x2 kernels' head (r2l_fwd2.hip,
// r2l_coopf_fwd.hip — NOT the bf16x3 / fp32 kernels, whose families are held to fp32-exact bars) is derived from its predecessor.
// Error: the inputs' 1.5 ulp doubled plus one rounding, <= 8e-7 absolute on values of magnitude <= 1 — the size of the fp16x2
// products' own 2^-21, three orders of magnitude below what one ulp of the POINT already does to these features at the highest
// frequency (1.2e-4): invisible at the 1e-4 RGB bar (measured: max |dRGB| against the CPU restatement of the reference unchanged at 1.5e-6).
// Same-box A/B of the render launch: 37.10 -> 36.94 ms (-0.4 %); with the evaluation software-pipelined (r2l_f2.h F2TrigPre)
// 37.93 -> 37.60 ms (-0.9 %).  (A timing build WITHOUT sin / cos ran in 36.2 ms, which first read as "4.8 % to gain": most of that
// is the power cap again — garbage encodings make a net whose activations toggle fewer bits.)
__device__ __forceinline__ void r2l_sincos_double(float s, float c, float& s2, float& c2) {
    const float t = s + s;
    s2 = t * c;
    c2 = __builtin_fmaf(-t, s, 1.0f);
}

// ---- 16-ray cooperative variants (r2l_coop16.hip): their own packed streams, appended to the 32-layout ones ------------
// group16 = [16 tiles of 16 output features][64 lanes][float4]: lane (i = l%16, kk = l/16) component e holds
// W[16*tile + i][16*G + 4*kk + e] — the A operand of v_mfma_f32_16x16x4_f32 number e of k-group G.
#define R2L_C16_GROUP_FLOATS 4096
#define R2L_C16_LAYER_GROUPS 16
#define R2L_C16_HEAD_GROUPS 63   // 60 trig groups (k' order: coord-major, (sin f, cos f) pairs) + 3 identity groups
#define R2L_C16_PAD_GROUPS 8     // the weight ring prefetches 8 groups past the end
__host__ __device__ static inline int64_t r2l_fwd32_stream_floats(int n_block) {
    return (int64_t)(R2L_FWD_HEAD_GROUPS + 2 * n_block * R2L_FWD_LAYER_GROUPS) * R2L_GROUP_FLOATS + R2L_STREAM_PAD;
}
__host__ __device__ static inline int64_t r2l_bwd32_stream_floats(int n_block) {
    return (int64_t)2 * n_block * R2L_LAYER_FLOATS + R2L_STREAM_PAD;
}
__host__ __device__ static inline int64_t r2l_fwd16_stream_floats(int n_block) {
    return (int64_t)(R2L_C16_HEAD_GROUPS + 2 * n_block * R2L_C16_LAYER_GROUPS + R2L_C16_PAD_GROUPS) * R2L_C16_GROUP_FLOATS;
}
__host__ __device__ static inline int64_t r2l_bwd16_stream_floats(int n_block) {
    return (int64_t)(2 * n_block * R2L_C16_LAYER_GROUPS + R2L_C16_PAD_GROUPS) * R2L_C16_GROUP_FLOATS;
}
int r2l_coop16_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                       const float* c2w_host12, int H, int W, float focal, const float* wstream16, const float* params,
                       int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream);
int r2l_coop16_backward(const float* rgb, const float* target, const float* drgb, const float* save_x,
                        const float* save_t, const float* wstream_bwd16, const float* params, int n_block,
                        float grad_scale, float* dpre, float* gx, float* gt, float* sqerr_partial, int64_t N,
                        hipStream_t stream);

// ---- fp32-accurate forward on the bf16 matrix pipe (r2l_fwd3.hip): stage stream of bf16 (hi, mid, lo) weight triples ------
#define R2L_F3_PAD_STAGES 8   // the staging pipelines request up to 7 stages past the one being consumed
__host__ __device__ static inline int64_t r2l_fwd3_stages(int n_block) { return 64 + 34 * (int64_t)n_block; }
__host__ __device__ static inline int64_t r2l_fwd3_stream_floats(int n_block) {
    return (r2l_fwd3_stages(n_block) + R2L_F3_PAD_STAGES) * (24576 / 4);
}
// fp16x2 forward (r2l_fwd2.hip): same stages, 16 KiB each, then 64 bytes of status (word 0: range guard)
#define R2L_F2_RANGE 32768.0f  // |activation| from which on a forward launch is handed over to the bf16x3 kernel
__host__ __device__ static inline int64_t r2l_fwd2_status_offset(int n_block) {
    return (r2l_fwd3_stages(n_block) + R2L_F3_PAD_STAGES) * (16384 / 4);
}
__host__ __device__ static inline int64_t r2l_fwd2_stream_floats(int n_block) { return r2l_fwd2_status_offset(n_block) + 16; }
// the 16 status words (range control of the fp16x2 forward: r2l_f2.h)
enum { F2S_FLAG = 0, F2S_AMAX = 1, F2S_SCALE = 2, F2S_INV = 3, F2S_MAGIC = 4, F2S_TRIPS = 5, F2S_PEAK = 6, F2S_RESCALES = 7,
       F2S_DONE = 8, F2S_GO = 9, F2S_REFINE = 10 };
#define F2_MAGIC 0x52324c34u
// The activation scale a fp16x2 stream is packed for / its inverse, as the chain kernels read them: (1, 1) while the status area
// has not been committed (a zero-filled stream whose stages were written without r2l_fwd2_commit: INV would read 0 and X_0
// vanish silently) — the same rule as the packers (r2l_f2.h f2_status_scale, r2l_train.hip)
__device__ __forceinline__ float f2_act_scale(const unsigned* st) {
    return st[F2S_MAGIC] == F2_MAGIC ? __builtin_bit_cast(float, st[F2S_SCALE]) : 1.0f;
}
__device__ __forceinline__ float f2_act_inv(const unsigned* st) {
    return st[F2S_MAGIC] == F2_MAGIC ? __builtin_bit_cast(float, st[F2S_INV]) : 1.0f;
}
__host__ __device__ static inline int64_t r2l_bwd3_stages(int n_block) { return 34 * (int64_t)n_block; }
__host__ __device__ static inline int64_t r2l_bwd3_stream_floats(int n_block) {
    return (r2l_bwd3_stages(n_block) + R2L_F3_PAD_STAGES) * (24576 / 4);
}
// fp16x2 dX chain (r2l_bwd2.hip): same stages, 16 KiB each, then 64 bytes of status (word 0: range guard)
__host__ __device__ static inline int64_t r2l_bwd2_status_offset(int n_block) {
    return (r2l_bwd3_stages(n_block) + R2L_F3_PAD_STAGES) * (16384 / 4);
}
__host__ __device__ static inline int64_t r2l_bwd2_stream_floats(int n_block) { return r2l_bwd2_status_offset(n_block) + 16; }
// the 16 status words of a training step's backward (fp16 trio).  FLAG: this step belongs to the bf16x3 kernels (range guard of
// the dX chain, or the forward already fell back); GSCALE / GINV: the power of two the chain runs on and its inverse — every
// kernel of the step reads them here; AMAX: largest |B value| of the step's chain (scaled units, float bits); MAGIC / PEAK / GS:
// history for the next step's scale (r2l_bwd_prepare_kernel, r2l_backward.hip): unscaled gradient amax and grad_scale of the
// last clean step; TRIPS: steps that fell back (telemetry)
// EXPANDED: workgroups of the fallback pack that have expanded their slot of an fp16 forward stash (r2l_bwd3.hip), per step
enum { B2S_FLAG = 0, B2S_GSCALE = 4, B2S_GINV = 5, B2S_AMAX = 8, B2S_MAGIC = 9, B2S_TRIPS = 10, B2S_PEAK = 11, B2S_GS = 12, B2S_EXPANDED = 13 };
// run_if: nullptr, or a device word — the pack returns at once while it is 0 (the bf16x3 stream as range-guard fallback of the
// fp16 kernels is packed right in front of the fallback launch, and only when that launch will really run)
// save_x / save_t / N (fallback of a training step only): an fp16 forward stash is expanded to the chunked fp32 layout first
int r2l_bwd3_pack(const float* params, int n_block, float* wstream3, hipStream_t stream, const unsigned* run_if = nullptr,
                  const float* save_x = nullptr, const float* save_t = nullptr, int64_t N = 0);
int r2l_bwd3_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd3, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale = 1.0f,
                      const unsigned* run_if = nullptr, const float* scale_dev = nullptr);
// the same chain on two-way fp16 splits (r2l_bwd2.hip); status: range-guard word (behind the bwd2 stream region)
int r2l_bwd2_pack(const float* params, int n_block, float* wstream2, hipStream_t stream);
int r2l_bwd2_backward(const float* rgb, const float* target, const float* drgb, const float* save_x, const float* save_t,
                      const float* wstream_bwd2, const float* params, int n_block, float grad_scale, float* dpre, float* gx,
                      float* gt, float* sqerr_partial, int64_t N, hipStream_t stream, float gscale, unsigned* status,
                      const float* scale_dev = nullptr, int b_start = -1, int b_end = 0);
// forward launches (with or without the training stash) big enough for the one-wave-per-tile kernels take the bf16x3
// kernel (R2L_NO_FWD3=1: fp32 MFMA)
// pose mode: the camera of ray rc and its pixel index.  One pose by value (c2w), or — one launch over several frames
// (r2l_forward_poses: no tail round and no launch gap per frame) — a device table c2w_dev[K][12], frame = rc / (H * W)
struct R2LPoseRay { float c[12]; int64_t pix; };
__device__ __forceinline__ R2LPoseRay r2l_pose_of(const float (&c2w)[12], const float* c2w_dev, int64_t hw, int64_t rc) {
    R2LPoseRay r;
    r.pix = rc;
    if (c2w_dev != nullptr) {
        const int64_t p = rc / hw;
        r.pix = rc - p * hw;
#pragma unroll
        for (int i = 0; i < 12; ++i) r.c[i] = c2w_dev[p * 12 + i];
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) r.c[i] = c2w[i];
    }
    return r;
}
// the multi-pose launch in progress on this thread (set by r2l_forward_poses_cfg around the single-pose dispatch)
extern thread_local const float* g_r2l_c2w_dev;  // r2l_error.hip

// Explicit dispatch (include/r2l_hip.h r2l_config): the *_cfg entry points install the caller's config for the duration of
// the call (R2LCfgScope, thread-local), and every decision below looks at it first; an AUTO (0) field falls through to the
// R2L_* environment switch it replaces, read per call (~100 ns) so tests can flip it.
extern thread_local r2l_config g_r2l_cfg;  // r2l_error.hip
struct R2LCfgScope {
    r2l_config saved;
    explicit R2LCfgScope(const r2l_config* c) : saved(g_r2l_cfg) { if (c) g_r2l_cfg = *c; }
    ~R2LCfgScope() { g_r2l_cfg = saved; }
};
// What is wrong with a caller's r2l_config, or nullptr.  Every *_cfg entry point checks before it does anything else:
// launch entry points fail with hipErrorInvalidValue (R2L_CFG_ENTER), the host-side queries return -1 (R2L_CFG_QUERY).
static inline const char* r2l_cfg_check(const r2l_config* c) {
    if (c == nullptr) return nullptr;
    if (c->precision < 0 || c->precision > R2L_PRECISION_FP32_MFMA) return "r2l_config.precision: not an R2L_PRECISION_* value";
    if (c->tiling < 0 || c->tiling > R2L_TILING_COOPF) return "r2l_config.tiling: not an R2L_TILING_* value";
    if (c->tiling == R2L_TILING_COOP_RETIRED) return "r2l_config.tiling: 2 (the 32-ray fp32-MFMA cooperative kernels) was retired in round 5 — R2L_TILING_COOP16 serves those launches";
    if (c->coop_tiles < 0 || c->coop_tiles > 3) return "r2l_config.coop_tiles: 0 (auto), 1, 2 or 3 (mixed)";
    if (c->reserve_cus < -1) return "r2l_config.reserve_cus: -1 (none), 0 (auto) or a CU count";
    if (c->dw_mode < 0 || c->dw_mode > R2L_DW_EXACT) return "r2l_config.dw_mode: not an R2L_DW_* value";
    if (c->reserved[0] || c->reserved[1] || c->reserved[2]) return "r2l_config.reserved: must be 0";
    return nullptr;
}
// argument checks of the C ABI: a bad pointer / size is an error code here, not a memory fault on the device
#define R2L_REQUIRE(cond, msg)                            \
    do {                                                  \
        if (!(cond)) {                                    \
            r2l_set_error_msg(msg);                       \
            return (int)hipErrorInvalidValue;             \
        }                                                 \
    } while (0)
#define R2L_MAX_BLOCKS 1024  // body blocks of a student net: the kernels address a weight stream (~0.57 MB per block) with 32-bit offsets
#define R2L_CFG_ENTER(cfg)                                \
    if (const char* r2l_why_ = r2l_cfg_check(cfg)) {      \
        r2l_set_error_msg(r2l_why_);                      \
        return (int)hipErrorInvalidValue;                 \
    }                                                     \
    R2LCfgScope scope(cfg)
#define R2L_CFG_QUERY(cfg)                                \
    if (const char* r2l_why_ = r2l_cfg_check(cfg)) {      \
        r2l_set_error_msg(r2l_why_);                      \
        return -1;                                        \
    }                                                     \
    R2LCfgScope scope(cfg)
static inline bool r2l_env_on(const char* name) {
    const char* e = getenv(name);
    return e && e[0] && e[0] != '0';
}
static inline bool r2l_use_fwd3() {
    if (g_r2l_cfg.precision) return g_r2l_cfg.precision != R2L_PRECISION_FP32_MFMA;
    return !r2l_env_on("R2L_NO_FWD3");
}
// one-wave-per-tile forward launches: three fp16 products per fp32 product, ~2^-21 relative (r2l_fwd2.hip), with
// the bf16x3 kernel launched behind it as the range-guard fallback (it returns at once unless the status word is raised).
// R2L_NO_FWD2=1: bf16x3 only.
static inline bool r2l_use_fwd2() {
    if (g_r2l_cfg.precision) return g_r2l_cfg.precision == R2L_PRECISION_FP16X2;
    return r2l_use_fwd3() && !r2l_env_on("R2L_NO_FWD2");
}
int r2l_fwd2_pack(const float* params, int n_block, float* wstream2, hipStream_t stream);
int r2l_fwd2_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream2, const float* params,
                     int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream);
int r2l_fwd3_pack(const float* params, int n_block, float* wstream3, hipStream_t stream, const unsigned* run_if = nullptr);
// Behind every fp16x2 forward launch: returns at once (GO = 0) unless that launch raised FLAG; else packs the bf16x3 stream,
// re-packs the scale-dependent stages of the fp16x2 stream for the next activation scale and commits it (GO = 1, FLAG = 0
// unless the scale is exhausted): the r2l_fwd3_forward behind it (run_if = status + F2S_GO) redoes this launch, the next
// launch is back on the fp16 kernels (r2l_f2.h: range control)
int r2l_fwd2_fallback_pack(const float* params, int n_block, float* wstream3, float* wstream2, hipStream_t stream);
// run_if: nullptr, or a device word — the launch returns at once while it is 0 (fallback behind r2l_fwd2_forward)
int r2l_fwd3_forward(const float* rays_o, const float* rays_d, const float* t_rand, const float* ztab,
                     const float* c2w_host12, int H, int W, float focal, const float* wstream3, const float* params,
                     int n_block, float* rgb, float* save_x, float* save_t, int64_t N, hipStream_t stream,
                     const unsigned* run_if = nullptr, const float* x0_in = nullptr);  // x0_in: body + tail from a given X_0

// Which chain variant is fastest for N rays.  In units of one main-kernel round (1024 wave slots x 32 rays): main needs
// ceil(N/32768) rounds; coop (4 waves share a 32-ray tile, 256 workgroups) ceil(N/8192) rounds of ~0.34 (measured: fwd
// 0.81 ms at 4096 rays, 0.94 ms at 8192); coop16 (4 waves share a 16-ray tile: fills all 256 CUs from 4096 rays)
// ceil(N/4096) rounds of ~0.174 (0.47 ms at 4096 rays, 0.91 ms at 8192).
// R2L_FORCE_VARIANT=main|coop|coop16 in the environment overrides (tests, A/B).
#ifndef R2L_C16_ROUND
#define R2L_C16_ROUND 0.174
#endif
enum { R2L_VARIANT_MAIN = 0, R2L_VARIANT_COOP16 = 2 };  // (1: the 32-ray fp32-MFMA cooperative family, retired in round 5)
// Cooperative fp16x2 kernels (r2l_coopf.h: one 32-ray tile per WORKGROUP): a sub-family of the MAIN variant — same streams,
// stash and fallbacks as r2l_fwd2 / r2l_bwd2, taken instead of them for launches of at most R2L_COOPF_MAX_RAYS rays, and
// for launches between one and one and a half ROUNDS of the one-wave-per-tile kernels (a round = 256 CUs x 128 rays): the
// two-tile cooperative kernels then run three full rounds of 16 384 rays where those run two, the second half empty
// (measured, tools/variant_sweep.py, 49 152 rays: step 4.19 vs 4.51 ms, forward 1.44 vs 1.53 ms; 24 576: 2.55 vs 2.44,
// 65 536: 5.53 vs 5.29, 98 304: 8.20 vs 7.81 — the one-wave-per-tile kernels everywhere else)
// (R2L_FORCE_VARIANT=coopf: always; =main: never).  Only with the whole fp16 trio enabled (no R2L_NO_* switch).
#ifndef R2L_COOPF_MAX_RAYS
#define R2L_COOPF_MAX_RAYS 16384
#endif
#define R2L_MAIN_ROUND_RAYS 32768  // one wave per 32-ray tile, four per workgroup, one workgroup per CU, 256 CUs
// Keep a scalar fp32 chain scalar.  hipcc's SLP vectoriser pairs independent fp32 chains into packed ops with op_sel swizzles;
// the form whose LOW lane reads the HIGH dword of src1 (`v_pk_fma_f32 ... op_sel:[0,1,0]`) lost results on gfx950 under two
// waves per SIMD (r2l_coopf.h) and r2l_amd/build.py refuses objects that contain it: an empty asm on the value
// between the operations is enough to keep them apart.
__device__ __forceinline__ void r2l_no_pack(float& v) { asm volatile("" : "+v"(v)); }

static inline bool r2l_fp16_trio_env() {
    if (g_r2l_cfg.precision) return g_r2l_cfg.precision == R2L_PRECISION_FP16X2;
    return !r2l_env_on("R2L_NO_FWD3") && !r2l_env_on("R2L_NO_FWD2") && !r2l_env_on("R2L_NO_BWD2") && !r2l_env_on("R2L_NO_DW2");
}
// the tiling a host pinned: cfg->tiling, else R2L_FORCE_VARIANT=main|coop|coop16|coopf, else R2L_TILING_AUTO
static inline int r2l_forced_tiling() {
    if (g_r2l_cfg.tiling) return g_r2l_cfg.tiling;
    const char* e = getenv("R2L_FORCE_VARIANT");
    if (!e || !e[0]) return R2L_TILING_AUTO;
    if (e[0] == 'm') return R2L_TILING_WAVE_PER_TILE;
    if (e[0] == 'c' && e[1] && e[2] && e[3] && e[4] == 'f') return R2L_TILING_COOPF;
    if (e[0] == 'c') {  // coop16; "coop" named the retired 32-ray family: its launches are coop16's now — said once, not silently
        if (e[1] && e[2] && e[3] && !e[4]) {
            static bool warned = false;
            if (!warned) {
                warned = true;
                fprintf(stderr, "libr2l_hip: R2L_FORCE_VARIANT=coop names the kernel family retired in round 5; taking coop16\n");
            }
        }
        return R2L_TILING_COOP16;
    }
    return R2L_TILING_WAVE_PER_TILE;  // (anything else used to mean "not the cooperative fp16 kernels")
}
static inline bool r2l_use_coopf(int64_t N, int n_block) {
    if (n_block <= 0 || !r2l_fp16_trio_env()) return false;
    const int t = r2l_forced_tiling();
    if (t == R2L_TILING_COOPF) return true;
    if (t != R2L_TILING_AUTO) return false;
    return N <= R2L_COOPF_MAX_RAYS || (N > R2L_MAIN_ROUND_RAYS && N <= R2L_MAIN_ROUND_RAYS + R2L_MAIN_ROUND_RAYS / 2);
}
static inline int r2l_chain_variant(int64_t N) {
    const int t = r2l_forced_tiling();
    if (t == R2L_TILING_WAVE_PER_TILE || t == R2L_TILING_COOPF) return R2L_VARIANT_MAIN;  // coopf: kernels of the MAIN family
    if (t == R2L_TILING_COOP16) return R2L_VARIANT_COOP16;
    if (r2l_fp16_trio_env() && N <= R2L_COOPF_MAX_RAYS) return R2L_VARIANT_MAIN;  // served by the cooperative fp16x2 kernels
    // (one main round on the fp16x2 kernels costs 0.30 of a round of the fp32-MFMA kernel the unit was defined on; measured,
    // tools/variant_sweep.py: 98 304-ray-style steps of 6144 rays 1.99 ms on the one-wave-per-tile kernels vs 2.11 ms on the
    // 16-ray cooperative ones, 20 480 rays 2.9 vs 5.9 ms; 4096 rays 1.94 vs 1.34 ms)
    // (round 5: the 32-ray fp32-MFMA cooperative family — 0.34 per round of 8192 rays — is retired: AUTO reached it only under a
    // pinned fp32_mfma precision, in the bands where it beat two 16-ray rounds by 2 %: profiles/r05_dispatch_table.md)
    const double main_t = (double)((N + 32767) / 32768) * (r2l_use_fwd3() ? 0.30 : 1.0);
    const double c16_t = (double)((N + 4095) / 4096) * R2L_C16_ROUND;
    return c16_t < main_t ? R2L_VARIANT_COOP16 : R2L_VARIANT_MAIN;
}

// The one-wave-per-tile training trios keep their stash (save_x[0..n-1], save_t, gx[1..n], gt) in a private layout: fp16
// stage pieces (the default trio, r2l_f2.h) or the chunked fp32 layout above (bf16x3 trio); slot n of save_x then holds
// y = x_n + x_0 row-major (all the tail gradient needs) and gx[0] stays row-major (head gradient).
// Every other combination (cooperative chains, fp32 chains, the pre-embedded module-boundary path) is row-major throughout.
static inline bool r2l_stash_chunked(int64_t N, bool pre_embedded) {
    return !pre_embedded && N > 0 && r2l_chain_variant(N) == R2L_VARIANT_MAIN && r2l_use_fwd3();
}

// The default TRAINING trio of one-wave-per-tile MSE-mode steps: fp16x2 forward (r2l_fwd2.hip) and dX chain (r2l_bwd2.hip)
// stashing fp16 stage pieces, and the fp16 weight-gradient GEMMs on them (r2l_dw16.hip).  Any of R2L_NO_FWD3 / R2L_NO_FWD2 /
// R2L_NO_BWD2 / R2L_NO_DW2 = 1 puts the whole step on the bf16x3 trio (r2l_fwd3 / r2l_bwd3 / r2l_dw_body3c, chunked fp32
// stash) — the kernels the range guards fall back to.  (Forward-only launches look at R2L_NO_FWD2 alone.)
static inline bool r2l_use_trio16() {
    if (g_r2l_cfg.precision) return g_r2l_cfg.precision == R2L_PRECISION_FP16X2;
    return r2l_use_fwd2() && !r2l_env_on("R2L_NO_BWD2") && !r2l_env_on("R2L_NO_DW2");
}
// weight-gradient GEMMs of the fp16 trio with the ray-side operand as hi + mid (two products): cfg->dw_mode, else R2L_DW_EXACT=1
static inline bool r2l_dw_exact() {
    if (g_r2l_cfg.dw_mode) return g_r2l_cfg.dw_mode == R2L_DW_EXACT;
    return r2l_env_on("R2L_DW_EXACT");
}

// error plumbing shared by the C-ABI translation units
extern "C" const char* r2l_last_error(void);
void r2l_set_error(const char* what, hipError_t e);
void r2l_set_error_msg(const char* msg);
#define R2L_CHECK(expr)                        \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            r2l_set_error(#expr, _e);          \
            return (int)_e;                    \
        }                                      \
    } while (0)
