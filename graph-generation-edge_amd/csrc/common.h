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
// Gumbel transform g(u) = -log(-log(u + 1e-30) + 1e-30)   (diffusion/diffusion_base.py:177) and the decision
// argmax(g(u0) + lp0, g(u1) + lp1), bit-identical to the HOST's for every u on the 2^-24 grid torch.rand emits -
// without the 64 MB table of all 2^24 host values the first two rounds gathered from (one 64-byte line per lookup).
//
// Three evaluations of g:
//   fast   two v_log_f32 (hardware log2, ~1 ulp) - a few vector instructions;
//   exact  each logarithm in fp64, rounded to fp32 (the correctly rounded fp32 logarithm up to double rounding);
//   host   what the host's torch computes (its vector math library is within 1 ulp but not always correctly
//          rounded: 15 862 of the 2^24 grid values differ from `exact` where MKL runs its Intel path, 2.38 M on an
//          AMD EPYC host).
// The "patch" blob (built once per process and device by edgehip/tables.py from an EXHAUSTIVE comparison of the
// grid, edgehip_gumbel_grid) holds the grid points where exact != host with the host's values, plus the largest
// |fast - host| over the whole grid, eps_fast.  A decision first uses `fast`: if |a1 - a0| exceeds
// 2 eps_fast + the rounding of the two sums, the host's decision is the same (its g differ from `fast` by at most
// eps_fast each) and nothing else is evaluated.  Otherwise (a near-tie: ~1e-5 of the decisions) both g are
// re-evaluated as `exact`, looked up in the patch list (binary search over its sorted keys) and the decision is
// taken on what are now the host's own values.  Off-grid u (only possible with supplied uniforms) or no blob:
// `exact`, no guarantee against the host (none existed before either).
//
// Blob layout (32-bit words): [0] magic, [1] n, [2] eps_fast (float), [3] reserved, [4 .. 4+n) keys k ascending
// (u = k 2^-24), [4+n .. 4+2n) host values (float).
// ---------------------------------------------------------------------------------------------
#define EH_GUMBEL_MAGIC 0x45484750u        // 'EHGP'
// (no fma contraction in any of these: the copies compiled into different translation units must give the same bits,
// and every decision is a plain fp32 add + compare as on the host)
__device__ static inline float eh_gumbel_fast(float u) {
#pragma clang fp contract(off)
    const float l1 = __builtin_amdgcn_logf(u + 1e-30f) * 0.69314718055994531f;          // v_log_f32 is log2
    return -(__builtin_amdgcn_logf(-l1 + 1e-30f) * 0.69314718055994531f);
}
__device__ static inline float eh_gumbel_exact(float u) {
#pragma clang fp contract(off)
    const float l1 = (float)log((double)(u + 1e-30f));
    return -(float)log((double)(-l1 + 1e-30f));
}
// the host's g(u): patched where the exhaustive comparison found exact != host
__device__ static inline float eh_gumbel_host(float u, const uint32_t* __restrict__ patch) {
    const float g = eh_gumbel_exact(u);
    const float kf = u * 16777216.0f;
    const uint32_t k = (uint32_t)kf;
    if (!(kf >= 0.0f && kf < 16777216.0f && (float)k == kf)) return g;                    // off the grid
    const int n = (int)patch[1];
    const uint32_t* keys = patch + 4;
    int lo = 0, hi = n;                                                                   // first key >= k
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
    if (lo < n && keys[lo] == k) return __uint_as_float(keys[n + lo]);
    return g;
}

// argmax over two classes with ties -> class 0 (torch.argmax returns the first maximum).
__device__ static inline int eh_decide(float u0, float u1, float lp0, float lp1, const float* __restrict__ patch_f) {
#pragma clang fp contract(off)
    const uint32_t* patch = reinterpret_cast<const uint32_t*>(patch_f);
    if (patch == nullptr) return (eh_gumbel_exact(u1) + lp1) > (eh_gumbel_exact(u0) + lp0) ? 1 : 0;
    float a0 = eh_gumbel_fast(u0) + lp0;
    float a1 = eh_gumbel_fast(u1) + lp1;
    const float eps = __uint_as_float(patch[2]);
    // |a_host - a_fast| <= eps + half an ulp of each sum's rounding (2^-23 |a| bounds a whole ulp)
    const float thr = 2.0f * eps + 2.5e-7f * (fabsf(a0) + fabsf(a1));
    if (!(fabsf(a1 - a0) > thr)) {                  // near-tie (or not finite): the host's own values decide
        a0 = eh_gumbel_host(u0, patch) + lp0;
        a1 = eh_gumbel_host(u1, patch) + lp1;
    }
    return a1 > a0 ? 1 : 0;
}

// evaluation of the device transforms on the grid u = (k0 + i) 2^-24 (patch construction and its tests); defined once
// per translation unit that takes decisions (state.hip: which = 0, gnn.hip: which = 1) so that a test can show the
// copies agree bit for bit
int eh_gumbel_grid_gnn(float* exact_out, float* fast_out, int64_t k0, int64_t n, hipStream_t s);
#define EH_DEFINE_GUMBEL_GRID_KERNEL(NAME)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float* __restrict__ exact_out, float* __restrict__ fast_out,        \
                                                long long k0, long long n) {                                        \
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;                                              \
        if (i >= n) return;                                                                                         \
        const float u = (float)(k0 + i) * 5.9604644775390625e-08f;                                                  \
        if (exact_out) exact_out[i] = eh_gumbel_exact(u);                                                           \
        if (fast_out) fast_out[i] = eh_gumbel_fast(u);                                                              \
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
