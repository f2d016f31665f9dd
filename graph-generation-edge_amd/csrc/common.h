// common.h - shared device helpers for libedgehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "edgehip.h"

#define EH_WAVE 64

#define EH_CHECK_ARG(cond) do { if (!(cond)) return EDGEHIP_ERR_ARG; } while (0)
#define EH_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return EDGEHIP_ERR_LAUNCH; } while (0)

static inline hipStream_t eh_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int eh_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// riders of the selection launch inside the fused reverse step (state.hip: eh_active_select_step)
struct eh_select_extra { int32_t* t_out; int32_t t; int32_t* clear; int clear_n; };
int eh_active_select_step(const edgehip_layout* L, const int32_t* d0, const int32_t* deg, int32_t t, int32_t* t_out,
                          const float* post_table, int32_t nmax, const float* uniforms, const edgehip_rng* rng,
                          const float* gumbel_lut, uint8_t* flags, int32_t* act_nodes, int32_t* act_count,
                          int32_t* clear, int clear_n, hipStream_t s);

// inclusive prefix sum over the 64 lanes on the DPP path (row shifts 1, 2, 4, 8 with zero fill, then the row totals
// handed to the next rows): six vector instructions instead of six LDS-crossbar shuffles with their index arithmetic
__device__ static inline int eh_wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xA, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xC, 0xF, true);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based.  One call gives 4 x 32 random bits.
// ---------------------------------------------------------------------------------------------
struct eh_u4 { uint32_t x, y, z, w; };

__host__ __device__ static inline eh_u4 eh_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return eh_u4{c0, c1, c2, c3};
}

// 24-bit uniform on the grid torch.rand (CPU, fp32) emits: k * 2^-24, k in [0, 2^24).
__host__ __device__ static inline float eh_u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// RNG stream ids (counter word 3 = step * 8 + stream)
#define EH_STREAM_NODE 0
#define EH_STREAM_PAIR 1
#define EH_STREAM_QTM1 2
#define EH_STREAM_QT 3

// ---------------------------------------------------------------------------------------------
// Gumbel transform g(u) = -log(-log(u + 1e-30) + 1e-30)   (diffusion/diffusion_base.py:177)
// With a LUT (2^24 entries built by the host libm) the value is bit-identical to the host's for
// every u on the 2^-24 grid; off-grid u (or no LUT) uses the device logf (<= 1 ulp each).
// ---------------------------------------------------------------------------------------------
__device__ static inline float eh_gumbel(float u, const float* __restrict__ lut) {
    if (lut) {
        float kf = u * 16777216.0f;
        uint32_t k = (uint32_t)kf;
        if (kf >= 0.0f && kf < 16777216.0f && (float)k == kf) return lut[k];
    }
    return -logf(-logf(u + 1e-30f) + 1e-30f);
}

// argmax over two classes with ties -> class 0 (torch.argmax returns the first maximum).
__device__ static inline int eh_decide(float u0, float u1, float lp0, float lp1, const float* __restrict__ lut) {
    float a0 = eh_gumbel(u0, lut) + lp0;
    float a1 = eh_gumbel(u1, lut) + lp1;
    return a1 > a0 ? 1 : 0;
}

// pair index of (i<j) in torch.triu_indices(n, n, 1) order
__host__ __device__ static inline int64_t eh_pair_index(int64_t n, int64_t i, int64_t j) {
    return i * (2 * n - i - 1) / 2 + (j - i - 1);
}

// decode k-th pair (0-based) of the strict upper triangle of a K x K matrix: k -> (p<q)
__device__ static inline void eh_pair_decode(int64_t k, int64_t K, int& p, int& q) {
    double b = 2.0 * (double)K - 1.0;
    int64_t pp = (int64_t)((b - sqrt(b * b - 8.0 * (double)k)) * 0.5);
    if (pp < 0) pp = 0;
    if (pp > K - 2) pp = K - 2;
    // fix rounding: need off(pp) <= k < off(pp+1), off(p) = p(2K-p-1)/2
    while (pp > 0 && pp * (2 * K - pp - 1) / 2 > k) --pp;
    while ((pp + 1) * (2 * K - pp - 2) / 2 <= k) ++pp;
    p = (int)pp;
    q = (int)(k - pp * (2 * K - pp - 1) / 2 + pp + 1);
}

__device__ static inline void eh_adj_set(uint32_t* row_i, int j) { atomicOr(row_i + (j >> 5), 1u << (j & 31)); }
__device__ static inline void eh_adj_clear(uint32_t* row_i, int j) { atomicAnd(row_i + (j >> 5), ~(1u << (j & 31))); }
__device__ static inline int eh_adj_get(const uint32_t* row_i, int j) { return (row_i[j >> 5] >> (j & 31)) & 1; }

__device__ static inline float eh_silu(float x) { return x / (1.0f + expf(-x)); }
// Hardware-rate forms for the per-element epilogues of the hot kernels: v_exp_f32 / v_rcp_f32 are ~1 ulp
// each; absolute error ~2e-7 on O(1) activations, far inside the 1e-4 bound on the logits.
__device__ static inline float eh_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ static inline float eh_silu_fast(float x) { return x * eh_sigmoid_fast(x); }
__device__ static inline float eh_tanh_fast(float x) {
    float e = __expf(-2.0f * fabsf(x));                       // in (0, 1]: no overflow
    float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}
