// gemv.h - block-cooperative GEMV for per-graph vectors (shared by the sampler's per-graph update and the
// training step's per-graph MLPs).
#pragma once
#include "common.h"
#include "gemm.h"

// 8-lane group reductions with DPP (no LDS crossbar): quad xor-1, quad xor-2, then half-row mirror.
__device__ static inline float eh_dpp_f(float v, const int ctrl_sel) {
    int t;
    if (ctrl_sel == 0) t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
    else if (ctrl_sel == 1) t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);
    else t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);
    return __builtin_bit_cast(float, t);
}

// Block-cooperative GEMV for the per-graph vectors: y[o] = bias[o] + W1[o,:K1] . x1 (+ W2[o,:64] . x2).
// 16 lanes per output row, one float4 per lane and 64-wide chunk: every wave-level load covers four
// complete 256-byte rows (thread-per-row reads touched 64 cache lines per instruction and made this
// latency-bound kernel 3x slower).  x vectors live in LDS.  Outputs o = first + 4*NW*pass + 4*wave + sub for a block of
// NW waves (the arithmetic of an output does not depend on NW).
__device__ static inline float eh_row16_sum(float v) {
    v += eh_dpp_f(v, 0);                                                                    // xor 1
    v += eh_dpp_f(v, 1);                                                                    // xor 2
    v += eh_dpp_f(v, 2);                                                                    // half-row mirror (8)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row mirror (16)
    return v;
}
template <int K1, bool TWO, int NW = 4>
__device__ static inline void eh_block_gemv(const float* __restrict__ W1, int ld1, const float* __restrict__ x1,
                                            const float* __restrict__ W2, int ld2, const float* __restrict__ x2,
                                            const float* __restrict__ bias, int first, int count, float* __restrict__ y,
                                            int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, kq = lane & 15;
    f32x4 xa[K1 / 64], xb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K1 / 64; ++j) xa[j] = *reinterpret_cast<const f32x4*>(x1 + 64 * j + 4 * kq);
    if (TWO) xb = *reinterpret_cast<const f32x4*>(x2 + 4 * kq);
    // PB passes (16 outputs each) per trip, all their loads (16 float4 per lane) issued before the first
    // reduction: a trip costs one L2 round trip
    constexpr int PB = 16 / (K1 / 64 + (TWO ? 1 : 0));
    for (int o0 = first + 4 * wave + sub; o0 < first + count; o0 += 4 * NW * PB) {
        f32x4 w1[PB][K1 / 64], w2[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int o = o0 + 4 * NW * p;
            const int oc = o < first + count ? o : o0;            // clamp: a valid row, result discarded
#pragma unroll
            for (int j = 0; j < K1 / 64; ++j)
                w1[p][j] = *reinterpret_cast<const f32x4*>(W1 + (size_t)oc * ld1 + 64 * j + 4 * kq);
            if (TWO) w2[p] = *reinterpret_cast<const f32x4*>(W2 + (size_t)oc * ld2 + 4 * kq);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int o = o0 + 4 * NW * p;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < K1 / 64; ++j)
                s += w1[p][j][0] * xa[j][0] + w1[p][j][1] * xa[j][1] + w1[p][j][2] * xa[j][2] + w1[p][j][3] * xa[j][3];
            if (TWO) s += w2[p][0] * xb[0] + w2[p][1] * xb[1] + w2[p][2] * xb[2] + w2[p][3] * xb[3];
            s = eh_row16_sum(s);
            if (kq == 0 && o < first + count) {
                s += bias ? bias[o] : 0.f;
                y[o] = act ? eh_silu(s) : s;
            }
        }
    }
}

