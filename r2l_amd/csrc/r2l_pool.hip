// Hard-ray pool on the device: the three data movements of /root/reference/main.py:1325-1347 (augment: n_hard_out random pool
// rows appended to every batch) and :1410-1425 (update: the hard_ratio * B rays with the largest per-ray error enter the pool,
// replacing the rows that were handed out) as ONE kernel each, plus the row choice.
//
// The reference draws `np.random.permutation(pool_rows)[:n_out]` on the host per step (1.6 M entries at the README sizes: a
// whole MI355X training step); rounds 1 - 3 used torch.randperm on the device — a 1.6 M-key radix sort, 0.16 ms per step, the
// largest single item of the CLI loop's overhead over the bare step (profiles/r04_e2e_train.txt).  All that is needed is n_out
// DISTINCT rows, every row equally likely: r2l_pool_pick evaluates a keyed bijection of [0, n_rows) — a 4-round Feistel network
// on the next even number of bits, cycle-walked back into the range — at i = 0 .. n_out-1.  A fresh key per step (host counter
// through a mixer) gives a fresh permutation; distinctness is by construction.  HBM-bound trivia otherwise: 36 B per row.
#include "r2l_common.h"

namespace {

__device__ __forceinline__ unsigned pool_mix(unsigned x) {  // murmur3 finalizer
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// bijection of [0, 2^(2*half_bits)): 4 Feistel rounds with round keys k[r]
__device__ __forceinline__ unsigned long long pool_feistel(unsigned long long x, int half_bits, const unsigned (&k)[4]) {
    const unsigned mask = (1u << half_bits) - 1u;
    unsigned l = (unsigned)(x >> half_bits) & mask, r = (unsigned)x & mask;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned f = pool_mix(r ^ k[i]) & mask;
        const unsigned nl = r;
        r = l ^ f;
        l = nl;
    }
    return ((unsigned long long)l << half_bits) | r;
}

__global__ void r2l_pool_pick_kernel(int64_t* __restrict__ out, int64_t n_out, int64_t n_rows, int half_bits, unsigned long long key) {
    const unsigned k[4] = {pool_mix((unsigned)key), pool_mix((unsigned)(key >> 32) ^ 0x9e3779b9u), pool_mix((unsigned)key ^ 0x7f4a7c15u),
                           pool_mix((unsigned)(key >> 32) + 0x6a09e667u)};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)i;
        do {
            x = pool_feistel(x, half_bits, k);  // cycle walking: the domain is < 4 n_rows, ~2 trips on average at worst
        } while (x >= (unsigned long long)n_rows);
        out[i] = (int64_t)x;
    }
}

// rows [0, B): the batch; rows [B, B + n_out): pool rows idx[i]; three contiguous [B + n_out, 3] outputs
__global__ void r2l_pool_augment_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ t, int64_t so,
                                        int64_t sd, int64_t st, const float* __restrict__ pool, const int64_t* __restrict__ idx, int64_t B,
                                        int64_t n_out, float* __restrict__ oo, float* __restrict__ od, float* __restrict__ ot) {
    const int64_t total = (B + n_out) * 9;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / 9;
        const int c = (int)(e - row * 9);
        float v;
        if (row < B) v = c < 3 ? o[row * so + c] : (c < 6 ? d[row * sd + c - 3] : t[row * st + c - 6]);
        else v = pool[idx[row - B] * 9 + c];
        float* dst = c < 3 ? oo : (c < 6 ? od : ot);
        dst[row * 3 + (c % 3)] = v;
    }
}

// pool[dst(i)] = [o, d, t][hard[i]]  for i < n_in;  dst(i) = dst_idx[i] or dst0 + i
__global__ void r2l_pool_store_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ t, int64_t so,
                                      int64_t sd, int64_t st, const int64_t* __restrict__ hard, float* __restrict__ pool,
                                      const int64_t* __restrict__ dst_idx, int64_t dst0, int64_t n_in) {
    const int64_t total = n_in * 9;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / 9;
        const int c = (int)(e - i * 9);
        const int64_t src = hard[i];
        const float v = c < 3 ? o[src * so + c] : (c < 6 ? d[src * sd + c - 3] : t[src * st + c - 6]);
        pool[(dst_idx != nullptr ? dst_idx[i] : dst0 + i) * 9 + c] = v;
    }
}

unsigned grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int r2l_pool_pick(int64_t* idx_out, int64_t n_out, int64_t n_rows, uint64_t key, void* stream) {
    R2L_REQUIRE(n_out >= 0 && n_rows >= 0 && n_out <= n_rows && n_rows < ((int64_t)1 << 60), "r2l_pool_pick: need 0 <= n_out <= n_rows");
    if (n_out == 0) return 0;
    R2L_REQUIRE(idx_out != nullptr, "r2l_pool_pick: idx_out is NULL");
    int bits = 2;
    while (((int64_t)1 << bits) < n_rows) bits += 2;  // even, 2^bits >= n_rows, < 4 n_rows
    hipLaunchKernelGGL(r2l_pool_pick_kernel, dim3(grid_for(n_out)), dim3(256), 0, (hipStream_t)stream, idx_out, n_out, n_rows, bits / 2,
                       (unsigned long long)key);
    R2L_CHECK(hipGetLastError());
    return 0;
}

extern "C" int r2l_pool_augment(const float* rays_o, const float* rays_d, const float* target, int64_t stride_o, int64_t stride_d,
                                int64_t stride_t, const float* pool, const int64_t* idx, int64_t B, int64_t n_out, float* out_o,
                                float* out_d, float* out_t, void* stream) {
    R2L_REQUIRE(B >= 0 && n_out >= 0, "r2l_pool_augment: negative B / n_out");
    if (B + n_out == 0) return 0;
    R2L_REQUIRE((B == 0 || (rays_o && rays_d && target)) && (n_out == 0 || (pool && idx)) && out_o && out_d && out_t,
                "r2l_pool_augment: a required pointer is NULL");
    hipLaunchKernelGGL(r2l_pool_augment_kernel, dim3(grid_for((B + n_out) * 9)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, target,
                       stride_o, stride_d, stride_t, pool, idx, B, n_out, out_o, out_d, out_t);
    R2L_CHECK(hipGetLastError());
    return 0;
}

extern "C" int r2l_pool_store(const float* rays_o, const float* rays_d, const float* target, int64_t stride_o, int64_t stride_d,
                              int64_t stride_t, const int64_t* hard, float* pool, const int64_t* dst_idx, int64_t dst0, int64_t n_in,
                              void* stream) {
    R2L_REQUIRE(n_in >= 0 && dst0 >= 0, "r2l_pool_store: negative n_in / dst0");
    if (n_in == 0) return 0;
    R2L_REQUIRE(rays_o && rays_d && target && hard && pool, "r2l_pool_store: a required pointer is NULL");
    hipLaunchKernelGGL(r2l_pool_store_kernel, dim3(grid_for(n_in * 9)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, target, stride_o,
                       stride_d, stride_t, hard, pool, dst_idx, dst0, n_in);
    R2L_CHECK(hipGetLastError());
    return 0;
}
