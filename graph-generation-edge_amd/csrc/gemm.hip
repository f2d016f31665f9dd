// gemm.hip - generic fp32 MFMA linear layer: Y = act(A W^T + bias + per-graph bias).
// Skinny-K (64..576), tall-M (one row per node) shapes: each wave keeps a 32-row A fragment in
// registers and streams W rows from L2; see gemm.h for the operand scheme.
#include "gemm.h"
#include "prof.h"

// One block = 4 waves = 4 row tiles sharing the weight tiles; the (column group, k chunk) trips of a block run
// as ONE software-pipelined sequence: the next weight tile (global -> registers) and the next A fragment are
// requested before the MFMAs of the current trip and land in the other LDS buffer after them, one barrier per
// trip.  The earlier form (barrier, stage, barrier, MFMA per trip) left the matrix pipe idle during every
// staging round trip: measured on the training step in DESIGN.md 5b.  With the split-bf16 products it is also
// the faster form for launches of many blocks per CU (32 polblogs graphs 11.2 -> 10.6 ms per step, 96: 30.4 -> 28.4),
// so it is the only one.
template <int NACC>
__device__ static inline void eh_gemm_lds_body(const eh_gemm_desc& d, int ng_per_block, int bx, int by, unsigned* wt) {
    constexpr int TSZ = NACC * 3 * EH_W3_PLANE;    // dwords per split weight tile; wt holds two
    constexpr int NLD = NACC * 32 * 16 / 256;      // float4 per thread and tile (256 threads)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    int tile = bx * 4 + wave;
    int ntiles;
    int64_t M = 0;
    if (d.tile_row0) {
        ntiles = d.num_tiles;
    } else {
        M = d.m_dev ? (int64_t)(*d.m_dev) : d.M;
        ntiles = (int)((M + 31) / 32);
    }
    if (bx * 4 >= ntiles) return;                  // block-uniform: nothing to do
    const bool live = tile < ntiles;               // idle waves still stage weights and hit barriers
    if (!live) tile = ntiles - 1;
    int row0, rows, g = 0;
    if (d.tile_row0) {
        row0 = d.tile_row0[tile]; rows = d.tile_rows[tile]; g = d.tile_graph[tile];
    } else {
        row0 = tile * 32; rows = (int)((M - row0) < 32 ? (M - row0) : 32);
    }
    const int r = lane & 31, hh = lane >> 5;
    const int64_t arow = row0 + (r < rows ? r : rows - 1);
    const float* ap = d.A + arow * d.lda + hh * 32;
    const float* ap2 = d.A2 ? d.A2 + arow * d.lda2 + hh * 32 : nullptr;
    const int ng_total = d.N / (32 * NACC);
    const int ng0 = by * ng_per_block;
    const int ng1 = (ng0 + ng_per_block < ng_total) ? ng0 + ng_per_block : ng_total;
    const int KC = d.K >> 6;
    const int nit = (ng1 - ng0) * KC;
    if (nit <= 0) return;
    const int srow = threadIdx.x >> 4, sc4 = (threadIdx.x & 15) * 4;       // staging coordinates (+16 rows per i)
    f32x4 wr[NLD];
    auto wload = [&](int ng, int kci) {
        const float* src = d.W + (size_t)(ng * NACC * 32 + srow) * d.K + kci * 64 + sc4;
#pragma unroll
        for (int i = 0; i < NLD; ++i) wr[i] = *reinterpret_cast<const f32x4*>(src + (size_t)i * 16 * d.K);
    };
    auto wstore = [&](unsigned* buf) {          // split into the three bf16 planes on the way into LDS
#pragma unroll
        for (int i = 0; i < NLD; ++i) eh_stage3_quad(buf, srow + 16 * i, sc4, wr[i]);
    };
    auto aload = [&](eh_frag& f, int kci) { eh_load_frag(f, (kci * 64 < d.K1) ? ap + kci * 64 : ap2 + (kci * 64 - d.K1)); };
    eh_frag an;
    eh_frag3 a;
    wload(ng0, 0);
    aload(an, 0);
    wstore(wt);
    eh_split3(a, an);
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = eh_zero16();
    int ng = ng0, kci = 0;
    for (int it = 0; it < nit; ++it) {
        const bool more = it + 1 < nit;
        int ngn = ng, kcn = kci + 1;
        if (kcn == KC) { kcn = 0; ++ngn; }
        if (more) {
            wload(ngn, kcn);
            if (KC > 1) aload(an, kcn);
        }
        const unsigned* cur = wt + (it & 1) * TSZ;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            eh_frag3 b;
            eh_load_frag3_lds(b, cur, j, r, hh);
            eh_mma64(acc[j], a, b);
        }
        if (kci == KC - 1) {
            if (live) {
#pragma unroll
                for (int j = 0; j < NACC; ++j) {
                    const int col = (ng * NACC + j) * 32 + r;
                    float add0 = d.bias ? d.bias[col] : 0.0f;
                    if (d.gbias0) add0 += d.gbias0[(int64_t)g * d.ldg + col];
                    const float add1 = d.gbias1 ? d.gbias1[(int64_t)g * d.ldg + col] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int rr = eh_crow(i, lane);
                        if (rr < rows) {
                            int64_t grow = row0 + rr;
                            float v = acc[j][i] + add0;
                            if (d.gbias1 && d.rowflag[grow] > 0) v += add1;
                            if (d.act == 1) v = eh_silu(v);
                            if (d.dact) {
                                const float x = d.dact[grow * d.ldy + col];
                                const float sg = 1.0f / (1.0f + expf(-x));
                                v *= sg * (1.0f + x * (1.0f - sg));
                            }
                            if (d.accum == 2) { atomicAdd(d.Y + grow * d.ldy + col, v); continue; }
                            if (d.accum) v += d.Y[grow * d.ldy + col];
                            d.Y[grow * d.ldy + col] = v;
                            if (d.Y2) d.Y2[grow * d.ldy + col] = eh_silu(v);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = eh_zero16();
        }
        if (more) {
            wstore(wt + ((it + 1) & 1) * TSZ);
            if (KC > 1) eh_split3(a, an);
        }
        ng = ngn; kci = kcn;
        __syncthreads();
    }
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void k_gemm(eh_gemm_desc d, int ng_per_block) {
    __shared__ __attribute__((aligned(16))) unsigned wt[2 * NACC * 3 * EH_W3_PLANE];
    eh_gemm_lds_body<NACC>(d, ng_per_block, blockIdx.x, blockIdx.y, wt);
}

// Several independent linear layers in ONE launch: blockIdx.z selects the descriptor.  At the training shapes
// (a few thousand rows, K <= 512) a launch is latency-bound, so the q/k/v/skip/gru products of one layer
// cost about as much together as the largest of them alone.
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_gemm_group(eh_gemm_group g) {
    __shared__ __attribute__((aligned(16))) unsigned wt[2 * NACC * 3 * EH_W3_PLANE];
    const int z = blockIdx.z;
    if ((int)blockIdx.x >= g.gx[z] || (int)blockIdx.y >= g.split[z]) return;
    eh_gemm_lds_body<NACC>(g.d[z], g.per[z], blockIdx.x, blockIdx.y, wt);
}

template <int NACC>
__global__ __launch_bounds__(256) void k_gemm_packed(eh_gemm_desc d, int ng_per_block) {
    eh_gemm_packed_body<NACC>(d, ng_per_block, blockIdx.x, blockIdx.y);
}

void eh_gemm_grid(const eh_gemm_desc& d, int& gx, int& split, int& per, int& nacc) {
    int64_t tiles = d.tile_row0 ? d.num_tiles : (d.M + 31) / 32;
    gx = (int)((tiles + 3) / 4);
    nacc = (d.N % 64 == 0) ? 2 : 1;
    const int ng_total = d.N / (32 * nacc);
    int want = (2048 + gx - 1) / gx;
    split = want < 1 ? 1 : (want > ng_total ? ng_total : want);
    per = (ng_total + split - 1) / split;
    split = (ng_total + per - 1) / per;
}

// Launch shape of the LDS-staged kernel: 4 waves per block sharing a weight tile, 64 columns per wave (32 when
// N is not a multiple of 64).  Cutting small problems finer (one wave per block, 32 columns per wave: 4x-8x more,
// shorter blocks) was measured on the training step and is SLOWER (polblogs B=4: 6.44 vs 5.08 ms per step;
// B=32: 13.75 vs 12.59): every wave then stages its own weight tile.
struct eh_gemm_cfg { int nacc, gx, split, per; };
static eh_gemm_cfg eh_gemm_config(const eh_gemm_desc& d) {
    const int64_t tiles = d.tile_row0 ? d.num_tiles : (d.M + 31) / 32;
    eh_gemm_cfg c;
    c.nacc = (d.N % 64 == 0) ? 2 : 1;
    c.gx = (int)((tiles + 3) / 4);
    const int ng_total = d.N / (32 * c.nacc);
    int want = (2048 + c.gx - 1) / c.gx;
    c.split = want < 1 ? 1 : (want > ng_total ? ng_total : want);
    c.per = (ng_total + c.split - 1) / c.split;
    c.split = (ng_total + c.per - 1) / c.per;
    return c;
}

int eh_launch_gemm(const eh_gemm_desc& d, hipStream_t stream) {
    if (!d.A || !d.W || !d.Y) return EDGEHIP_ERR_ARG;
    if (d.w_packed && (d.accum || d.Y2 || d.dact)) return EDGEHIP_ERR_UNSUPPORTED;
    if (d.K <= 0 || d.N <= 0 || (d.K & 63) || (d.N & 31)) return EDGEHIP_ERR_UNSUPPORTED;
    if (d.A2 ? ((d.K1 & 63) || d.K1 <= 0 || d.K1 >= d.K) : (d.K1 != d.K)) return EDGEHIP_ERR_ARG;
    if ((d.lda & 3) || (d.A2 && (d.lda2 & 3))) return EDGEHIP_ERR_ARG;
    int64_t tiles = d.tile_row0 ? d.num_tiles : (d.M + 31) / 32;
    if (tiles <= 0) return EDGEHIP_OK;
    if (d.w_packed) {
        int gx, split, per, nacc;
        eh_gemm_grid(d, gx, split, per, nacc);
        if (nacc == 2)
            EH_PROF(d.prof_id, stream, hipLaunchKernelGGL(k_gemm_packed<2>, dim3(gx, split), dim3(256), 0, stream, d, per));
        else
            EH_PROF(d.prof_id, stream, hipLaunchKernelGGL(k_gemm_packed<1>, dim3(gx, split), dim3(256), 0, stream, d, per));
    } else {
        const eh_gemm_cfg c = eh_gemm_config(d);
        if (c.nacc == 2)
            EH_PROF(d.prof_id, stream, hipLaunchKernelGGL(k_gemm<2>, dim3(c.gx, c.split), dim3(256), 0, stream, d, c.per));
        else
            EH_PROF(d.prof_id, stream, hipLaunchKernelGGL(k_gemm<1>, dim3(c.gx, c.split), dim3(256), 0, stream, d, c.per));
    }
    if (hipGetLastError() != hipSuccess) return EDGEHIP_ERR_LAUNCH;
    return EDGEHIP_OK;
}

int eh_launch_gemm_group(const eh_gemm_desc* ds, int n, hipStream_t stream) {
    if (n <= 0) return EDGEHIP_OK;
    if (n > EH_GROUP_MAX) return EDGEHIP_ERR_ARG;
    eh_gemm_group g = {};
    int mx = 0, my = 0, m = 0;
    const int nw = 4;
    for (int i = 0; i < n; ++i) {
        const eh_gemm_desc& d = ds[i];
        if (!d.A || !d.W || !d.Y) return EDGEHIP_ERR_ARG;
        if (d.K <= 0 || d.N <= 0 || (d.K & 63) || (d.N & 63) || d.w_packed) return EDGEHIP_ERR_UNSUPPORTED;
        if (d.A2 ? ((d.K1 & 63) || d.K1 <= 0 || d.K1 >= d.K) : (d.K1 != d.K)) return EDGEHIP_ERR_ARG;
        if ((d.lda & 3) || (d.A2 && (d.lda2 & 3))) return EDGEHIP_ERR_ARG;
        int64_t tiles = d.tile_row0 ? d.num_tiles : (d.M + 31) / 32;
        if (tiles <= 0) continue;
        const int gx = (int)((tiles + nw - 1) / nw), ng_total = d.N / 64;
        int want = (2048 + gx - 1) / gx;
        int split = want < 1 ? 1 : (want > ng_total ? ng_total : want);
        const int per = (ng_total + split - 1) / split;
        split = (ng_total + per - 1) / per;
        g.d[m] = d; g.gx[m] = gx; g.split[m] = split; g.per[m] = per;
        mx = gx > mx ? gx : mx; my = split > my ? split : my;
        ++m;
    }
    if (m == 0) return EDGEHIP_OK;
    EH_PROF(ds[0].prof_id, stream, hipLaunchKernelGGL(k_gemm_group<2>, dim3(mx, my, m), dim3(256), 0, stream, g));
    if (hipGetLastError() != hipSuccess) return EDGEHIP_ERR_LAUNCH;
    return EDGEHIP_OK;
}

extern "C" int edgehip_linear_fwd(const float* X, int64_t M, int32_t K, const float* W, const float* bias, int32_t N,
                                  int32_t act, float* Y, void* stream) {
    EH_CHECK_ARG(X && W && Y && M >= 0);
    if (M == 0) return EDGEHIP_OK;
    eh_gemm_desc d = {};
    d.A = X; d.lda = K; d.K1 = K; d.W = W; d.K = K; d.N = N; d.bias = bias; d.Y = Y; d.ldy = N; d.act = act; d.M = M; d.prof_id = EH_K_MISC;
    return eh_launch_gemm(d, eh_stream(stream));
}
