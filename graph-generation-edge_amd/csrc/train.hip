// train.hip - training ops of the path: forward kernels that keep what the backward needs, and their
// backward twins ("_bwd twins for 6-11 in training").  The training step is a composition
// of these ops by edgehip/train.py (one torch.autograd.Function per op, buffers owned by the caller);
// unlike the sampling path nothing is folded, so every gradient lands directly on a reference parameter.
// Batches are small here (polblogs: 4 graphs), so the ops favour plain, checkable kernels.
#include "common.h"
#include "gemm.h"
#include "gemv.h"
#include "prof.h"

#define TR_D 64

// ------------------------------------------------------------------------------------------------
// wave-wide sum over 64 lanes, result in every lane
// ------------------------------------------------------------------------------------------------
__device__ static inline float tr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// A. input embeddings  X0[i] = [ w_t*s_t+b_t | w_0*s_0+b_0 | emb_sel[sel] ]   (layers.py:326-336)
// ------------------------------------------------------------------------------------------------
__device__ static inline void tr_scalars(const int32_t* deg, const int32_t* d0, int i, int max_degree, float& st, float& s0) {
    int dt = deg[i];
    if (dt > max_degree + 1) dt = max_degree + 1;
    st = (float)dt / (float)max_degree;
    s0 = (float)d0[i] / (float)max_degree;
}

__global__ void k_tr_embed_fwd(int Nn, const int32_t* deg, const int32_t* d0, const uint8_t* flags, int max_degree,
                               const float* wt, const float* bt, const float* w0, const float* b0, const float* wsel,
                               float* X0) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Nn * 192) return;
    int i = idx / 192, c = idx % 192;
    float st, s0;
    tr_scalars(deg, d0, i, max_degree, st, s0);
    float v;
    if (c < 64) v = wt[c] * st + bt[c];
    else if (c < 128) v = w0[c - 64] * s0 + b0[c - 64];
    else v = wsel[(flags[i] ? 64 : 0) + c - 128];
    X0[idx] = v;
}

// one block = 64 rows; thread c (0..191) walks the rows and adds its column sums with one atomic each
__global__ __launch_bounds__(192) void k_tr_embed_bwd(int Nn, const int32_t* deg, const int32_t* d0, const uint8_t* flags,
                                                      int max_degree, const float* dX0, float* dwt, float* dbt, float* dw0,
                                                      float* db0, float* dwsel) {
    const int c = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int r1 = r0 + 64 < Nn ? r0 + 64 : Nn;
    float a = 0.f, b = 0.f;
    for (int i = r0; i < r1; ++i) {
        float g = dX0[(int64_t)i * 192 + c];
        if (c < 128) {
            float st, s0;
            tr_scalars(deg, d0, i, max_degree, st, s0);
            a += g * (c < 64 ? st : s0);
            b += g;
        } else {
            if (flags[i]) b += g; else a += g;
        }
    }
    if (c < 64) { atomicAdd(dwt + c, a); atomicAdd(dbt + c, b); }
    else if (c < 128) { atomicAdd(dw0 + c - 64, a); atomicAdd(db0 + c - 64, b); }
    else { atomicAdd(dwsel + c - 128, a); atomicAdd(dwsel + 64 + c - 128, b); }
}

// ------------------------------------------------------------------------------------------------
// B. linear backward.  dX = dY W is edgehip_linear_fwd on the transposed weight (k_tr_transpose);
//    dW[n][k] = sum_m dY[m][n] X[m][k] and db[n] = sum_m dY[m][n] run on the fp32 MFMA with the ROW index
//    as the reduction dimension: in step s lanes 0-31 feed row m0+2s, lanes 32-63 row m0+2s+1, each lane
//    one column of dY (operand A) and one of X (operand B).  Row chunks are split over blockIdx.z and
//    the four waves of a block; partial tiles are added with float atomics (dW, db zero-filled by the caller).
// ------------------------------------------------------------------------------------------------
__global__ void k_tr_transpose(const float* W, int N, int K, float* WT) {
    __shared__ float t[32][33];
    int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < N && k0 + tx < K) t[r][tx] = W[(int64_t)(n0 + r) * K + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K && n0 + tx < N) WT[(int64_t)(k0 + r) * N + n0 + tx] = t[tx][r];
}

__global__ __launch_bounds__(256) void k_tr_gemm_tn(const float* __restrict__ dY, const float* __restrict__ X, int64_t M,
                                                    int N, int K, int rows_per_block, float* dW, float* db) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int c = lane & 31, half = lane >> 5;
    const int rows_per_wave = rows_per_block / 4;          // multiple of 2
    int64_t m0 = (int64_t)blockIdx.z * rows_per_block + (int64_t)wave * rows_per_wave;
    int64_t m1 = m0 + rows_per_wave;
    if (m1 > M) m1 = M;
    if (m0 >= m1) return;
    f32x16 acc = eh_zero16();
    float bsum = 0.f;
    const bool ncol = n0 + c < N, kcol = k0 + c < K;
    for (int64_t m = m0; m < m1; m += 16) {
        float a[8], b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int64_t r = m + 2 * s + half;
            bool ok = r < m1;
            a[s] = (ok && ncol) ? dY[r * N + n0 + c] : 0.f;
            b[s] = (ok && kcol) ? X[r * K + k0 + c] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            bsum += a[s];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int n = n0 + eh_crow(i, lane);
        if (n < N && kcol) atomicAdd(dW + (int64_t)n * K + k0 + c, acc[i]);
    }
    if (db && blockIdx.y == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (half == 0 && ncol) atomicAdd(db + n0 + c, bsum);
    }
}

// The same product for many rows (M >= TR_TN64_MIN_ROWS, N and K multiples of 64).  k_tr_gemm_tn re-reads a 32-column
// strip of dY for every 32 columns of X and the other way round, 4 bytes per lane: M N K / 4 bytes from L2 per launch
// (640 MB for the q/k/v weights of 32 polblogs graphs), which is what its time follows there.  Here a block owns a
// 64 x 64 tile of dW: 32-row chunks of both operands go through LDS once per block (16-byte loads, half the bytes
// per flop), each wave multiplies one 32 x 32 quarter, the next chunk is requested before the products of the
// current one (two LDS buffers, one barrier per chunk).
#ifndef TR_TN64_MIN_ROWS
#define TR_TN64_MIN_ROWS 8192
#endif
#define TR_TNP 68
__global__ __launch_bounds__(256) void k_tr_gemm_tn64(const float* __restrict__ dY, const float* __restrict__ X, int64_t M,
                                                      int N, int K, int rows_per_block, float* dW, float* db) {
    __shared__ __attribute__((aligned(16))) float ys[2][32 * TR_TNP], xs[2][32 * TR_TNP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int c = lane & 31, half = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.z * rows_per_block;
    int64_t m1 = m0 + rows_per_block;
    if (m1 > M) m1 = M;
    if (m0 >= m1) return;                                   // block-uniform
    const int srow = tid >> 4, sc4 = (tid & 15) * 4;        // staging: rows srow and srow + 16, 4 columns
    f32x4 ry[2], rx[2];
    auto fetch = [&](int64_t m) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t r = m + srow + 16 * i;
            const bool ok = r < m1;
            ry[i] = ok ? *reinterpret_cast<const f32x4*>(dY + r * N + n0 + sc4) : f32x4{0.f, 0.f, 0.f, 0.f};
            rx[i] = ok ? *reinterpret_cast<const f32x4*>(X + r * K + k0 + sc4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&ys[buf][(srow + 16 * i) * TR_TNP + sc4]) = ry[i];
            *reinterpret_cast<f32x4*>(&xs[buf][(srow + 16 * i) * TR_TNP + sc4]) = rx[i];
        }
    };
    f32x16 acc = eh_zero16();
    float bsum = 0.f;
    fetch(m0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m = m0; m < m1; m += 32) {
        const bool more = m + 32 < m1;
        if (more) fetch(m + 32);
        const float* ya = &ys[buf][half * TR_TNP + wn * 32 + c];
        const float* xb = &xs[buf][half * TR_TNP + wk * 32 + c];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = ya[2 * s * TR_TNP], b = xb[2 * s * TR_TNP];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            bsum += a;
        }
        if (more) stash(buf ^ 1);
        buf ^= 1;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        atomicAdd(dW + (int64_t)(n0 + wn * 32 + eh_crow(i, lane)) * K + k0 + wk * 32 + c, acc[i]);
    if (db && blockIdx.y == 0 && wk == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (half == 0) atomicAdd(db + n0 + wn * 32 + c, bsum);
    }
}

// ------------------------------------------------------------------------------------------------
// C. activations (kind 1 SiLU layers.py act, 2 Mish layers.py:40-42)
// ------------------------------------------------------------------------------------------------
__device__ static inline float tr_act(int kind, float x) {
    if (kind == 1) return x / (1.0f + expf(-x));
    float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}
__device__ static inline float tr_act_grad(int kind, float x) {
    float s = 1.0f / (1.0f + expf(-x));
    if (kind == 1) return s * (1.0f + x * (1.0f - s));
    float sp = x > 20.f ? x : log1pf(expf(x));
    float th = tanhf(sp);
    return th + x * (1.0f - th * th) * s;
}
__global__ void k_tr_act_fwd(int kind, const float* x, int64_t n, float* y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = tr_act(kind, x[i]);
}
__global__ void k_tr_act_bwd(int kind, const float* x, const float* dy, int64_t n, float* dx) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * tr_act_grad(kind, x[i]);
}

// ------------------------------------------------------------------------------------------------
// D/E. assemble  out[i] = [ h[i] + ctx[g] | tg[g] or tr[i] ]  (layers.py:350-353,366,371,377) and the
//      per-graph reductions behind scatter(mean) / repeat_interleave (layers.py:342-345,367-369)
// ------------------------------------------------------------------------------------------------
__global__ void k_tr_assemble_fwd(int Nn, const int32_t* node_graph, const float* h, const float* ctx, const float* tg,
                                  const float* tr, int Kt, float* out) {
    const int W = 64 + Kt;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Nn * W) return;
    int i = (int)(idx / W), c = (int)(idx % W);
    int g = node_graph[i];
    float v;
    if (c < 64) v = h[(int64_t)i * 64 + c] + (ctx ? ctx[g * 64 + c] : 0.f);
    else v = tg ? tg[g * 64 + c - 64] : tr[(int64_t)i * Kt + c - 64];
    out[idx] = v;
}

// column sums of src[rows of graph g][col0 .. col0+64) ADDED into out[g][64] (scaled): blockIdx.y = graph,
// blockIdx.x = chunk of TR_SEG_ROWS rows, one float atomic per (block, column); `out` is zero-filled (or holds
// the value to accumulate onto) before the launch.  One block per graph left 4 blocks on 256 CUs (70 us).
#define TR_SEG_ROWS 32
__global__ __launch_bounds__(256) void k_tr_segment_sum(const int32_t* node_ptr, const float* src, int ld, int col0,
                                                        const float* add, int mean, float* out) {
    __shared__ float part[4][64];
    const int g = blockIdx.y, c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i0 = node_ptr[g], i1 = node_ptr[g + 1];
    const int a = i0 + (int)blockIdx.x * TR_SEG_ROWS;
    if (a >= i1 && blockIdx.x != 0) return;                 // chunk 0 still delivers `add` for an empty graph
    const int b = a + TR_SEG_ROWS < i1 ? a + TR_SEG_ROWS : i1;
    float s = 0.f;
    for (int i = a + r; i < b; i += 4) s += src[(int64_t)i * ld + col0 + c];
    part[r][c] = s;
    __syncthreads();
    if (r == 0) {
        s = part[0][c] + part[1][c] + part[2][c] + part[3][c];
        if (mean) s = (i1 > i0) ? s / (float)(i1 - i0) : 0.f;
        if (add && blockIdx.x == 0) s += add[g * 64 + c];
        atomicAdd(out + g * 64 + c, s);
    }
}
// backward of `assemble` for a [Nn,128] gradient in ONE launch: dh[i] += src[i][:64], sumA[g] += column sums of
// src[:, :64] (optional), sumB[g] += column sums of src[:, 64:] (both onto zero-filled / running buffers)
__global__ __launch_bounds__(256) void k_tr_assemble_bwd128(const int32_t* node_ptr, const float* __restrict__ src,
                                                            float* __restrict__ dh, float* sumA, float* sumB) {
    __shared__ float pa[4][64], pb[4][64];
    const int g = blockIdx.y, c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i0 = node_ptr[g], i1 = node_ptr[g + 1];
    const int a = i0 + (int)blockIdx.x * TR_SEG_ROWS;
    if (a >= i1) return;
    const int b = a + TR_SEG_ROWS < i1 ? a + TR_SEG_ROWS : i1;
    float sa = 0.f, sb = 0.f;
    for (int i = a + r; i < b; i += 4) {
        const float va = src[(int64_t)i * 128 + c], vb = src[(int64_t)i * 128 + 64 + c];
        dh[(int64_t)i * 64 + c] += va;
        sa += va; sb += vb;
    }
    pa[r][c] = sa; pb[r][c] = sb;
    __syncthreads();
    if (r == 0) {
        if (sumA) atomicAdd(sumA + g * 64 + c, pa[0][c] + pa[1][c] + pa[2][c] + pa[3][c]);
        atomicAdd(sumB + g * 64 + c, pb[0][c] + pb[1][c] + pb[2][c] + pb[3][c]);
    }
}
static inline dim3 tr_seg_grid(const edgehip_layout* L) {
    int chunks = (L->max_nodes + TR_SEG_ROWS - 1) / TR_SEG_ROWS;
    return dim3(chunks < 1 ? 1 : chunks, L->num_graphs);
}

// dst[i][0..ncol) (+)= src[i][col0 .. col0+ncol)   (strided slice -> contiguous)
__global__ void k_tr_slice(int64_t M, const float* src, int ld, int col0, int ncol, float* dst, int acc) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * ncol) return;
    int64_t i = idx / ncol;
    int c = (int)(idx % ncol);
    float v = src[i * ld + col0 + c];
    dst[idx] = acc ? dst[idx] + v : v;
}

// a += b + c + d  (float4 per thread)
__global__ void k_tr_add4(int64_t n4, float* a, const float* b, const float* c, const float* d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 va = reinterpret_cast<f32x4*>(a)[i];
    const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i], vc = reinterpret_cast<const f32x4*>(c)[i],
                vd = reinterpret_cast<const f32x4*>(d)[i];
    va += vb + vc + vd;
    reinterpret_cast<f32x4*>(a)[i] = va;
}

// dx[i][c] (+)= dout[g(i)][c] / n_g
__global__ void k_tr_segment_mean_bwd(int Nn, const int32_t* node_graph, const int32_t* node_ptr, const float* dout,
                                      float* dx, int acc) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Nn * 64) return;
    int i = idx >> 6, c = idx & 63;
    int g = node_graph[i];
    float v = dout[g * 64 + c] / (float)(node_ptr[g + 1] - node_ptr[g]);
    dx[idx] = acc ? dx[idx] + v : v;
}

// ------------------------------------------------------------------------------------------------
// F. attention (pyg TransformerConv, concat=False, root_weight: SURVEY App. B; call site layers.py:354)
//    One block per target node, one wave per head, lane = channel.  The bit row of the node is its
//    neighbour list.  Forward keeps the per-head outputs O and the log-sum-exp of the scores.
// ------------------------------------------------------------------------------------------------
struct tr_row {
    const uint32_t* bits;
    int words, node0;
};
__device__ static inline tr_row tr_row_of(const edgehip_layout& L, const uint32_t* adj, int i) {
    int g = L.node_graph[i];
    int n0 = L.node_ptr[g];
    int st = L.adj_stride[g];
    tr_row r;
    r.bits = adj + L.adj_ptr[g] + (int64_t)(i - n0) * st;
    r.words = (L.node_ptr[g + 1] - n0 + 31) >> 5;
    r.node0 = n0;
    return r;
}

// Neighbour walk shared by the three attention kernels.  The bit row is decoded 64 words at a time into an
// LDS list by wave 0 (popcount + wave prefix sum); every wave (= head) then visits FOUR neighbours per step:
// lane = (slot 0..3, 4 channels), the 64-channel dot products are 4 adds + 4 exchanges inside a 16-lane
// group, and the per-slot partial results meet at the end (the one-neighbour-per-step form spent its time
// in 64-lane reductions: 140 us per launch on 4 polblogs graphs).
#define TR_LIST 2048
__device__ static inline int tr_decode_chunk(const tr_row& row, int w0, int* list, int* cnt) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int w = w0 + lane;
        uint32_t bits = (w < row.words) ? row.bits[w] : 0u;
        const int pc = __popc(bits);
        int pre = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(pre, o, 64);
            if (lane >= o) pre += t;
        }
        int off = pre - pc;
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            list[off++] = row.node0 + w * 32 + b;
        }
        if (lane == 63) *cnt = pre;
    }
    __syncthreads();
    return *cnt;
}
__device__ static inline float tr_sum16(float v) {          // sum over the 16 lanes of a slot, in every lane
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ static inline float tr_dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
__device__ static inline f32x4 tr_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ static inline f32x4 tr_slots_sum(f32x4 v) {       // sum of the four slots, in every lane
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[c] += __shfl_xor(v[c], 16, 64);
        v[c] += __shfl_xor(v[c], 32, 64);
    }
    return v;
}

template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_fwd(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                        const float* V, const float* skip, float* out, float* O, float* lse) {
    __shared__ int list[TR_LIST];
    __shared__ int cnt;
    __shared__ float red[H][64];
    const int i = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int slot = lane >> 4, c4 = (lane & 15) * 4;
    const tr_row row = tr_row_of(L, adj, i);
    const int HD = H * 64;
    const f32x4 q = tr_ld4(Q + (int64_t)i * HD + h * 64 + c4);
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < row.words; w0 += 64) {
        const int n = tr_decode_chunk(row, w0, list, &cnt);
        for (int e0 = 0; e0 < n; e0 += 8) {                 // two neighbours per slot and trip: four row loads in flight
            const bool oka = e0 + slot < n, okb = e0 + 4 + slot < n;
            const int64_t ja = list[oka ? e0 + slot : e0], jb = list[okb ? e0 + 4 + slot : e0];
            const f32x4 ka = tr_ld4(K + ja * HD + h * 64 + c4), va = tr_ld4(V + ja * HD + h * 64 + c4);
            const f32x4 kb = tr_ld4(K + jb * HD + h * 64 + c4), vb = tr_ld4(V + jb * HD + h * 64 + c4);
            const float sa = tr_sum16(tr_dot4(q, ka)) * 0.125f;
            float sb = tr_sum16(tr_dot4(q, kb)) * 0.125f;
            if (oka) {                                       // okb implies oka
                sb = okb ? sb : -INFINITY;
                const float mn = fmaxf(m, fmaxf(sa, sb));
                const float r = expf(m - mn), pa = expf(sa - mn), pb = expf(sb - mn);
                l = l * r + pa + pb;
                acc = acc * r + pa * va + pb * vb;
                m = mn;
            }
        }
        __syncthreads();
    }
    // the four slots meet
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        f32x4 a2;
#pragma unroll
        for (int c = 0; c < 4; ++c) a2[c] = __shfl_xor(acc[c], o, 64);
        const float mn = fmaxf(m, m2);
        const float r1 = l > 0.f ? expf(m - mn) : 0.f, r2 = l2 > 0.f ? expf(m2 - mn) : 0.f;
        l = l * r1 + l2 * r2;
        acc = acc * r1 + a2 * r2;
        m = mn;
    }
    f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
    if (l > 0.f) o4 = acc * (1.0f / (l + 1e-16f));            // pyg softmax: w / (sum + 1e-16)
    if (slot == 0) {
        *reinterpret_cast<f32x4*>(O + (int64_t)i * HD + h * 64 + c4) = o4;
        *reinterpret_cast<f32x4*>(&red[h][c4]) = o4;
    }
    if (lane == 0) lse[(int64_t)i * H + h] = l > 0.f ? m + logf(l + 1e-16f) : 0.f;
    __syncthreads();
    if (h == 0) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) sum += red[k][lane];
        out[(int64_t)i * 64 + lane] = sum * (1.0f / H) + skip[(int64_t)i * 64 + lane];
    }
}

// dQ and the row terms D[i][h] = dO_i^h . O_i^h
template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_bwd_q(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                          const float* V, const float* O, const float* lse,
                                                          const float* dout, float* Dbuf, float* dQ) {
    __shared__ int list[TR_LIST];
    __shared__ int cnt;
    const int i = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int slot = lane >> 4, c4 = (lane & 15) * 4;
    const tr_row row = tr_row_of(L, adj, i);
    const int HD = H * 64;
    const f32x4 q = tr_ld4(Q + (int64_t)i * HD + h * 64 + c4);
    const f32x4 go = tr_ld4(dout + (int64_t)i * 64 + c4) * (1.0f / H);
    const float D = tr_sum16(tr_dot4(go, tr_ld4(O + (int64_t)i * HD + h * 64 + c4)));
    const float ls = lse[(int64_t)i * H + h];
    if (lane == 0) Dbuf[(int64_t)i * H + h] = D;
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < row.words; w0 += 64) {
        const int n = tr_decode_chunk(row, w0, list, &cnt);
        for (int e0 = 0; e0 < n; e0 += 8) {
            const bool oka = e0 + slot < n, okb = e0 + 4 + slot < n;
            const int64_t ja = list[oka ? e0 + slot : e0], jb = list[okb ? e0 + 4 + slot : e0];
            const f32x4 ka = tr_ld4(K + ja * HD + h * 64 + c4), va = tr_ld4(V + ja * HD + h * 64 + c4);
            const f32x4 kb = tr_ld4(K + jb * HD + h * 64 + c4), vb = tr_ld4(V + jb * HD + h * 64 + c4);
            const float sa = tr_sum16(tr_dot4(q, ka)) * 0.125f, da = tr_sum16(tr_dot4(go, va));
            const float sb = tr_sum16(tr_dot4(q, kb)) * 0.125f, db = tr_sum16(tr_dot4(go, vb));
            if (oka) dq += (expf(sa - ls) * (da - D) * 0.125f) * ka;
            if (okb) dq += (expf(sb - ls) * (db - D) * 0.125f) * kb;
        }
        __syncthreads();
    }
    dq = tr_slots_sum(dq);
    if (slot == 0) *reinterpret_cast<f32x4*>(dQ + (int64_t)i * HD + h * 64 + c4) = dq;
}

// dK, dV of node j: the adjacency is symmetric, so the targets that attend to j are row j's bits
template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_bwd_kv(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                           const float* V, const float* lse, const float* Dbuf,
                                                           const float* dout, float* dK, float* dV) {
    __shared__ int list[TR_LIST];
    __shared__ int cnt;
    const int j = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int slot = lane >> 4, c4 = (lane & 15) * 4;
    const tr_row row = tr_row_of(L, adj, j);
    const int HD = H * 64;
    const f32x4 kj = tr_ld4(K + (int64_t)j * HD + h * 64 + c4), vj = tr_ld4(V + (int64_t)j * HD + h * 64 + c4);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < row.words; w0 += 64) {
        const int n = tr_decode_chunk(row, w0, list, &cnt);
        for (int e0 = 0; e0 < n; e0 += 8) {
            const bool oka = e0 + slot < n, okb = e0 + 4 + slot < n;
            const int64_t ia = list[oka ? e0 + slot : e0], ib = list[okb ? e0 + 4 + slot : e0];
            const f32x4 qa = tr_ld4(Q + ia * HD + h * 64 + c4), qb = tr_ld4(Q + ib * HD + h * 64 + c4);
            const f32x4 ga = tr_ld4(dout + ia * 64 + c4) * (1.0f / H), gb = tr_ld4(dout + ib * 64 + c4) * (1.0f / H);
            const float la = lse[ia * H + h], lb = lse[ib * H + h], Da = Dbuf[ia * H + h], Db = Dbuf[ib * H + h];
            const float sa = tr_sum16(tr_dot4(qa, kj)) * 0.125f, da = tr_sum16(tr_dot4(ga, vj));
            const float sb = tr_sum16(tr_dot4(qb, kj)) * 0.125f, db = tr_sum16(tr_dot4(gb, vj));
            if (oka) {
                const float a = expf(sa - la);
                dk += (a * (da - Da) * 0.125f) * qa;
                dv += a * ga;
            }
            if (okb) {
                const float a = expf(sb - lb);
                dk += (a * (db - Db) * 0.125f) * qb;
                dv += a * gb;
            }
        }
        __syncthreads();
    }
    dk = tr_slots_sum(dk);
    dv = tr_slots_sum(dv);
    if (slot == 0) {
        *reinterpret_cast<f32x4*>(dK + (int64_t)j * HD + h * 64 + c4) = dk;
        *reinterpret_cast<f32x4*>(dV + (int64_t)j * HD + h * 64 + c4) = dv;
    }
}

// ------------------------------------------------------------------------------------------------
// G. GRU gates (torch.nn.GRU one step, gates r,z,n; layers.py:360)
// ------------------------------------------------------------------------------------------------
__device__ static inline float tr_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void k_tr_gru_fwd(int Nn, const float* gi, const float* gh, const float* h, float* hout) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Nn * 64) return;
    int i = idx >> 6, c = idx & 63;
    const float* a = gi + (int64_t)i * 192;
    const float* b = gh + (int64_t)i * 192;
    float r = tr_sigmoid(a[c] + b[c]);
    float z = tr_sigmoid(a[64 + c] + b[64 + c]);
    float n = tanhf(a[128 + c] + r * b[128 + c]);
    hout[idx] = (1.0f - z) * n + z * h[idx];
}

__global__ void k_tr_gru_bwd(int Nn, const float* gi, const float* gh, const float* h, const float* dhout, float* dgi,
                             float* dgh, float* dh) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Nn * 64) return;
    int i = idx >> 6, c = idx & 63;
    const float* a = gi + (int64_t)i * 192;
    const float* b = gh + (int64_t)i * 192;
    float r = tr_sigmoid(a[c] + b[c]);
    float z = tr_sigmoid(a[64 + c] + b[64 + c]);
    float hn = b[128 + c];
    float n = tanhf(a[128 + c] + r * hn);
    float g = dhout[idx];
    float dn = g * (1.0f - z), dz = g * (h[idx] - n);
    float dnp = dn * (1.0f - n * n);
    float drp = dnp * hn * r * (1.0f - r);
    float dzp = dz * z * (1.0f - z);
    float* da = dgi + (int64_t)i * 192;
    float* db = dgh + (int64_t)i * 192;
    da[c] = drp; db[c] = drp;
    da[64 + c] = dzp; db[64 + c] = dzp;
    da[128 + c] = dnp; db[128 + c] = dnp * r;
    dh[idx] = g * z;
}

// ------------------------------------------------------------------------------------------------
// H. pair gather-add  e[k] = z[row_k] + z[col_k]  (layers.py:380)
// ------------------------------------------------------------------------------------------------
__global__ void k_tr_pair_fwd(int64_t A, const int64_t* row, const int64_t* col, const float* z, int W, float* e) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A * W) return;
    int64_t k = idx / W;
    int c = (int)(idx % W);
    e[idx] = z[row[k] * W + c] + z[col[k] * W + c];
}
__global__ void k_tr_pair_bwd(int64_t A, const int64_t* row, const int64_t* col, const float* de, int W, float* dz) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A * W) return;
    int64_t k = idx / W;
    int c = (int)(idx % W);
    float g = de[idx];
    atomicAdd(dz + row[k] * W + c, g);
    atomicAdd(dz + col[k] * W + c, g);
}

// target bit of each active pair in x_{t-1} and the pair's graph (diffusion_binomial_active.py:165-170)
__global__ void k_tr_pair_targets(edgehip_layout L, const uint32_t* adj_tm1, int64_t A, const int64_t* row,
                                  const int64_t* col, uint8_t* tgt, int32_t* pair_graph) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A) return;
    int i = (int)row[k], j = (int)col[k];
    int g = L.node_graph[i];
    int n0 = L.node_ptr[g];
    const uint32_t* r = adj_tm1 + L.adj_ptr[g] + (int64_t)(i - n0) * L.adj_stride[g];
    tgt[k] = (uint8_t)eh_adj_get(r, j - n0);
    pair_graph[k] = g;
}

// ------------------------------------------------------------------------------------------------
// I. last linear (64 -> 2) + log_softmax + cross-entropy, one wave per pair
//    (final_out.4 layers.py:381; _compute_MC_KL_joint diffusion_binomial_active.py:157-181)
// ------------------------------------------------------------------------------------------------
// a wave owns TR_CE_CHUNK consecutive pairs (pairs are graph-major): it keeps the running sum of the current graph
// and adds it with ONE atomic when the graph changes (one atomic per pair serialised on B addresses: 170 us)
#define TR_CE_CHUNK 32
__global__ __launch_bounds__(256) void k_tr_ce_fwd(int64_t A, const float* e2, const float* W4, const float* b4,
                                                   const uint8_t* tgt, const int32_t* pair_graph, float* logits,
                                                   float* ce_graph) {
    const int64_t k0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * TR_CE_CHUNK;
    const int c = threadIdx.x & 63;
    if (k0 >= A) return;
    const int64_t k1 = k0 + TR_CE_CHUNK < A ? k0 + TR_CE_CHUNK : A;
    const float w0 = W4[c], w1 = W4[64 + c], bb0 = b4[0], bb1 = b4[1];
    int gcur = pair_graph[k0];
    float sum = 0.f;
    for (int64_t k = k0; k < k1; ++k) {
        float x = e2[k * 64 + c];
        float l0 = tr_wave_sum(x * w0) + bb0;
        float l1 = tr_wave_sum(x * w1) + bb1;
        int g = pair_graph[k];
        if (g != gcur) {
            if (c == 0) atomicAdd(ce_graph + gcur, sum);
            gcur = g; sum = 0.f;
        }
        float m = fmaxf(l0, l1);
        float lz = m + logf(expf(l0 - m) + expf(l1 - m));
        sum += -((tgt[k] ? l1 : l0) - lz);
        if (c == 0) { logits[k * 2] = l0; logits[k * 2 + 1] = l1; }
    }
    if (c == 0) atomicAdd(ce_graph + gcur, sum);
}

__global__ __launch_bounds__(256) void k_tr_ce_bwd(int64_t A, const float* e2, const float* W4, const float* b4,
                                                   const uint8_t* tgt, const int32_t* pair_graph, const float* gkl,
                                                   float* de2, float* dW4, float* db4) {
    __shared__ float sw[4][2][64];
    __shared__ float sb[4][2];
    const int wv = threadIdx.x >> 6, c = threadIdx.x & 63;
    float w0 = 0.f, w1 = 0.f, b0 = 0.f, b1 = 0.f;
    // each wave walks a strided set of pairs so that the dW4 partials are reduced once per block
    for (int64_t k = (int64_t)blockIdx.x * 4 + wv; k < A; k += (int64_t)gridDim.x * 4) {
        float x = e2[k * 64 + c];
        float l0 = tr_wave_sum(x * W4[c]) + b4[0];
        float l1 = tr_wave_sum(x * W4[64 + c]) + b4[1];
        float m = fmaxf(l0, l1);
        float e0 = expf(l0 - m), e1 = expf(l1 - m);
        float inv = 1.0f / (e0 + e1);
        float g = gkl[pair_graph[k]];
        float d0 = g * (e0 * inv - (tgt[k] ? 0.f : 1.f));
        float d1 = g * (e1 * inv - (tgt[k] ? 1.f : 0.f));
        de2[k * 64 + c] = d0 * W4[c] + d1 * W4[64 + c];
        w0 += d0 * x; w1 += d1 * x;
        b0 += d0; b1 += d1;
    }
    sw[wv][0][c] = w0; sw[wv][1][c] = w1;
    if (c == 0) { sb[wv][0] = b0; sb[wv][1] = b1; }
    __syncthreads();
    if (wv < 2) {
        float s = sw[0][wv][c] + sw[1][wv][c] + sw[2][wv][c] + sw[3][wv][c];
        atomicAdd(dW4 + wv * 64 + c, s);
        if (c == 0) atomicAdd(db4 + wv, sb[0][wv] + sb[1][wv] + sb[2][wv] + sb[3][wv]);
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
#define TR_GRID(n, b) dim3((unsigned)(((n) + (b) - 1) / (b)))

extern "C" int edgehip_train_embed_fwd(const edgehip_layout* L, const int32_t* deg, const int32_t* d0, const uint8_t* flags,
                                       int32_t max_degree, const float* wt, const float* bt, const float* w0,
                                       const float* b0, const float* wsel, float* X0, void* stream) {
    EH_CHECK_ARG(L && deg && d0 && flags && wt && bt && w0 && b0 && wsel && X0 && max_degree > 0);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    int64_t n = (int64_t)L->num_nodes * 192;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_embed_fwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream),
                       L->num_nodes, deg, d0, flags, max_degree, wt, bt, w0, b0, wsel, X0));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_train_embed_bwd(const edgehip_layout* L, const int32_t* deg, const int32_t* d0, const uint8_t* flags,
                                       int32_t max_degree, const float* dX0, float* dwt, float* dbt, float* dw0,
                                       float* db0, float* dwsel, void* stream) {
    EH_CHECK_ARG(L && deg && d0 && flags && dX0 && dwt && dbt && dw0 && db0 && dwsel && max_degree > 0);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_embed_bwd, TR_GRID(L->num_nodes, 64), dim3(192), 0,
                       eh_stream(stream), L->num_nodes, deg, d0, flags, max_degree, dX0, dwt, dbt, dw0, db0, dwsel));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_linear_bwd_input(const float* dY, int64_t M, int32_t N, const float* W, int32_t K, float* WT_scratch,
                                        float* dX, void* stream) {
    EH_CHECK_ARG(dY && W && WT_scratch && dX && M >= 0 && N > 0 && K > 0);
    if ((N & 63) || (K & 31)) return EDGEHIP_ERR_UNSUPPORTED;
    if (M == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_transpose, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0,
                       eh_stream(stream), W, N, K, WT_scratch));
    EH_LAUNCH_CHECK();
    return edgehip_linear_fwd(dY, M, N, WT_scratch, nullptr, K, 0, dX, stream);
}

extern "C" int edgehip_linear_bwd_weight(const float* dY, const float* X, int64_t M, int32_t N, int32_t K, float* dW,
                                         float* db, void* stream) {
    EH_CHECK_ARG(dY && X && dW && M >= 0 && N > 0 && K > 0);
    if (M == 0) return EDGEHIP_OK;
    if (M >= TR_TN64_MIN_ROWS && (N & 63) == 0 && (K & 63) == 0) {
        const int tiles64 = (N / 64) * (K / 64);
        int64_t want = (1024 + tiles64 - 1) / tiles64;               // ~1024 blocks in all
        int64_t rpb = (M + want - 1) / want;
        rpb = ((rpb + 31) / 32) * 32;
        if (rpb < 256) rpb = 256;
        const int gz = (int)((M + rpb - 1) / rpb);
        EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_gemm_tn64, dim3(N / 64, K / 64, gz), dim3(256), 0,
                           eh_stream(stream), dY, X, M, N, K, (int)rpb, dW, db));
        EH_LAUNCH_CHECK();
        return EDGEHIP_OK;
    }
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
#ifndef TR_TN_BLOCKS
#define TR_TN_BLOCKS 512
#endif
    int64_t want = (TR_TN_BLOCKS + tiles - 1) / tiles;               // blocks along M for ~TR_TN_BLOCKS blocks in all
    int64_t rpb = (M + want - 1) / want;
    rpb = ((rpb + 63) / 64) * 64;                                    // 4 waves x multiple of 16 rows
    if (rpb < 64) rpb = 64;
    int gz = (int)((M + rpb - 1) / rpb);
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_gemm_tn, dim3((N + 31) / 32, (K + 31) / 32, gz), dim3(256), 0,
                       eh_stream(stream), dY, X, M, N, K, (int)rpb, dW, db));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_act_fwd(int32_t kind, const float* x, int64_t n, float* y, void* stream) {
    EH_CHECK_ARG(x && y && n >= 0 && (kind == 1 || kind == 2));
    if (n == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_act_fwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream), kind, x, n, y));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_act_bwd(int32_t kind, const float* x, const float* dy, int64_t n, float* dx, void* stream) {
    EH_CHECK_ARG(x && dy && dx && n >= 0 && (kind == 1 || kind == 2));
    if (n == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_act_bwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream), kind, x, dy, n, dx));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_assemble_fwd(const edgehip_layout* L, const float* h, const float* ctx, const float* tail_graph,
                                    const float* tail_rows, int32_t Kt, float* out, void* stream) {
    EH_CHECK_ARG(L && h && out && Kt >= 0);
    EH_CHECK_ARG(Kt == 0 || (tail_graph && Kt == 64 && !tail_rows) || (tail_rows && !tail_graph));
    if (L->num_nodes == 0) return EDGEHIP_OK;
    int64_t n = (int64_t)L->num_nodes * (64 + Kt);
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_assemble_fwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream),
                       L->num_nodes, L->node_graph, h, ctx, tail_graph, tail_rows, Kt, out));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_assemble_bwd(const edgehip_layout* L, const float* dout, int32_t Kt, float* dh, float* dctx,
                                    float* dtail_graph, float* dtail_rows, void* stream) {
    EH_CHECK_ARG(L && dout && dh && Kt >= 0);
    EH_CHECK_ARG(!dtail_graph || Kt == 64);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    hipStream_t s = eh_stream(stream);
    const int ld = 64 + Kt;
    int64_t n = (int64_t)L->num_nodes * 64;
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_slice, TR_GRID(n, 256), dim3(256), 0, s, (int64_t)L->num_nodes, dout, ld, 0, 64, dh, 0));
    if (dctx) {
        if (hipMemsetAsync(dctx, 0, (size_t)L->num_graphs * 64 * sizeof(float), s) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_sum, tr_seg_grid(L), dim3(256), 0, s, L->node_ptr, dout, ld, 0,
                           (const float*)nullptr, 0, dctx));
    }
    if (dtail_graph) {
        if (hipMemsetAsync(dtail_graph, 0, (size_t)L->num_graphs * 64 * sizeof(float), s) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_sum, tr_seg_grid(L), dim3(256), 0, s, L->node_ptr, dout, ld, 64,
                           (const float*)nullptr, 0, dtail_graph));
    }
    if (dtail_rows && Kt > 0) {
        int64_t m = (int64_t)L->num_nodes * Kt;
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_slice, TR_GRID(m, 256), dim3(256), 0, s, (int64_t)L->num_nodes, dout, ld, 64, Kt,
                           dtail_rows, 0));
    }
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_segment_mean_fwd(const edgehip_layout* L, const float* x, const float* add, float* out, void* stream) {
    EH_CHECK_ARG(L && x && out);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    if (hipMemsetAsync(out, 0, (size_t)L->num_graphs * 64 * sizeof(float), eh_stream(stream)) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_segment_sum, tr_seg_grid(L), dim3(256), 0, eh_stream(stream),
                       L->node_ptr, x, 64, 0, add, 1, out));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_segment_mean_bwd(const edgehip_layout* L, const float* dout, float* dx, void* stream) {
    EH_CHECK_ARG(L && dout && dx);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    int64_t n = (int64_t)L->num_nodes * 64;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_segment_mean_bwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream),
                       L->num_nodes, L->node_graph, L->node_ptr, dout, dx, 0));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

template <int H>
static int tr_attn_fwd(const edgehip_layout* L, const uint32_t* adj, const float* q, const float* k, const float* v,
                       const float* skip, float* out, float* O, float* lse, hipStream_t s) {
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_attn_fwd<H>, dim3(L->num_nodes), dim3(64 * H), 0, s, *L, adj, q, k, v, skip, out, O, lse));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
template <int H>
static int tr_attn_bwd(const edgehip_layout* L, const uint32_t* adj, const float* q, const float* k, const float* v,
                       const float* O, const float* lse, const float* dout, float* Dbuf, float* dq, float* dk, float* dv,
                       hipStream_t s) {
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_attn_bwd_q<H>, dim3(L->num_nodes), dim3(64 * H), 0, s, *L, adj, q, k, v, O, lse, dout, Dbuf, dq));
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_attn_bwd_kv<H>, dim3(L->num_nodes), dim3(64 * H), 0, s, *L, adj, q, k, v, lse, Dbuf, dout, dk, dv));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_attn_fwd(const edgehip_layout* L, const uint32_t* adj, int32_t heads, const float* q, const float* k,
                                const float* v, const float* skip, float* out, float* O, float* lse, void* stream) {
    EH_CHECK_ARG(L && adj && q && k && v && skip && out && O && lse);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    hipStream_t s = eh_stream(stream);
    switch (heads) {
        case 1: return tr_attn_fwd<1>(L, adj, q, k, v, skip, out, O, lse, s);
        case 2: return tr_attn_fwd<2>(L, adj, q, k, v, skip, out, O, lse, s);
        case 4: return tr_attn_fwd<4>(L, adj, q, k, v, skip, out, O, lse, s);
        case 8: return tr_attn_fwd<8>(L, adj, q, k, v, skip, out, O, lse, s);
    }
    return EDGEHIP_ERR_UNSUPPORTED;
}

extern "C" int edgehip_attn_bwd(const edgehip_layout* L, const uint32_t* adj, int32_t heads, const float* q, const float* k,
                                const float* v, const float* O, const float* lse, const float* dout, float* Dbuf,
                                float* dq, float* dk, float* dv, void* stream) {
    EH_CHECK_ARG(L && adj && q && k && v && O && lse && dout && Dbuf && dq && dk && dv);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    hipStream_t s = eh_stream(stream);
    switch (heads) {
        case 1: return tr_attn_bwd<1>(L, adj, q, k, v, O, lse, dout, Dbuf, dq, dk, dv, s);
        case 2: return tr_attn_bwd<2>(L, adj, q, k, v, O, lse, dout, Dbuf, dq, dk, dv, s);
        case 4: return tr_attn_bwd<4>(L, adj, q, k, v, O, lse, dout, Dbuf, dq, dk, dv, s);
        case 8: return tr_attn_bwd<8>(L, adj, q, k, v, O, lse, dout, Dbuf, dq, dk, dv, s);
    }
    return EDGEHIP_ERR_UNSUPPORTED;
}

extern "C" int edgehip_gru_gate_fwd(const float* gi, const float* gh, const float* h, int64_t Nn, float* hout, void* stream) {
    EH_CHECK_ARG(gi && gh && h && hout && Nn >= 0);
    if (Nn == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_gru_fwd, TR_GRID(Nn * 64, 256), dim3(256), 0, eh_stream(stream),
                       (int)Nn, gi, gh, h, hout));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_gru_gate_bwd(const float* gi, const float* gh, const float* h, const float* dhout, int64_t Nn,
                                    float* dgi, float* dgh, float* dh, void* stream) {
    EH_CHECK_ARG(gi && gh && h && dhout && dgi && dgh && dh && Nn >= 0);
    if (Nn == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_gru_bwd, TR_GRID(Nn * 64, 256), dim3(256), 0, eh_stream(stream),
                       (int)Nn, gi, gh, h, dhout, dgi, dgh, dh));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_pair_gather_add_fwd(const float* z, int32_t width, const int64_t* row, const int64_t* col, int64_t A,
                                           float* e, void* stream) {
    EH_CHECK_ARG(z && row && col && e && A >= 0 && width > 0);
    if (A == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_pair_fwd, TR_GRID(A * width, 256), dim3(256), 0, eh_stream(stream),
                       A, row, col, z, width, e));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_pair_gather_add_bwd(const float* de, int32_t width, const int64_t* row, const int64_t* col, int64_t A,
                                           float* dz, void* stream) {
    EH_CHECK_ARG(de && row && col && dz && A >= 0 && width > 0);
    if (A == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_pair_bwd, TR_GRID(A * width, 256), dim3(256), 0, eh_stream(stream),
                       A, row, col, de, width, dz));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_pair_targets(const edgehip_layout* L, const uint32_t* adj_tm1, const int64_t* row, const int64_t* col,
                                    int64_t A, uint8_t* tgt, int32_t* pair_graph, void* stream) {
    EH_CHECK_ARG(L && adj_tm1 && row && col && tgt && pair_graph && A >= 0);
    if (A == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_pair_targets, TR_GRID(A, 256), dim3(256), 0, eh_stream(stream),
                       *L, adj_tm1, A, row, col, tgt, pair_graph));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_edge_ce_fwd(const float* e2, const float* W4, const float* b4, const uint8_t* tgt,
                                   const int32_t* pair_graph, int64_t A, float* logits, float* ce_graph, void* stream) {
    EH_CHECK_ARG(e2 && W4 && b4 && tgt && pair_graph && logits && ce_graph && A >= 0);
    if (A == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_ce_fwd, TR_GRID(A, 4 * TR_CE_CHUNK), dim3(256), 0, eh_stream(stream),
                       A, e2, W4, b4, tgt, pair_graph, logits, ce_graph));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_edge_ce_bwd(const float* e2, const float* W4, const float* b4, const uint8_t* tgt,
                                   const int32_t* pair_graph, const float* gkl, int64_t A, float* de2, float* dW4,
                                   float* db4, void* stream) {
    EH_CHECK_ARG(e2 && W4 && b4 && tgt && pair_graph && gkl && de2 && dW4 && db4 && A >= 0);
    if (A == 0) return EDGEHIP_OK;
    int64_t blocks = (A + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_ce_bwd, dim3((unsigned)blocks), dim3(256), 0, eh_stream(stream),
                       A, e2, W4, b4, tgt, pair_graph, gkl, de2, dW4, db4));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ================================================================================================
// I2. dropout (torch.nn.Dropout in training mode, layers/layers.py:286,357,361) on [Nn,64] rows.
//     y = x * keep / (1 - p).  keep comes from supplied 0/1 masks (parity runs: the reference is patched to use the
//     same masks) or from Philox keyed (seed; graph global id, node, channel group, slot), so that a graph's masks do
//     not depend on how the batch is sharded.  The backward is the same op applied to the gradient.
//     One thread = 4 consecutive channels of one node.
// ------------------------------------------------------------------------------------------------
#define EH_STREAM_DROPOUT 5
__global__ __launch_bounds__(256) void k_tr_dropout(edgehip_layout L, const float* __restrict__ x, float p, float scale,
                                                    uint64_t seed, int slot, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (node, 4-channel group)
    const int64_t node = idx >> 4;
    const int c4 = (int)(idx & 15);
    if (node >= L.num_nodes) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + node * 64 + c4 * 4);
    float k[4];
    if (mask) {
        const uint8_t* m = mask + node * 64 + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) k[e] = m[e] ? scale : 0.f;
    } else {
        const int b = L.node_graph[node];
        const uint32_t li = (uint32_t)(node - L.node_ptr[b]);
        const eh_u4 r = eh_philox(seed, li * 16u + (uint32_t)c4, (uint32_t)slot, (uint32_t)(L.gid_base + b), EH_STREAM_DROPOUT);
        k[0] = eh_u24(r.x) >= p ? scale : 0.f; k[1] = eh_u24(r.y) >= p ? scale : 0.f;
        k[2] = eh_u24(r.z) >= p ? scale : 0.f; k[3] = eh_u24(r.w) >= p ? scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(y + node * 64 + c4 * 4) = f32x4{v[0] * k[0], v[1] * k[1], v[2] * k[2], v[3] * k[3]};
}

static bool tr_dp_on(const edgehip_dropout* dp) { return dp != nullptr && dp->p > 0.f; }

extern "C" int edgehip_dropout_fwd(const edgehip_layout* L, const float* x, const edgehip_dropout* dp, int32_t slot,
                                   float* y, void* stream) {
    EH_CHECK_ARG(L && x && y && dp && slot >= 0 && slot < 2 * EDGEHIP_MAX_LAYERS);
    if (!(dp->p >= 0.f && dp->p < 1.f)) return EDGEHIP_ERR_ARG;
    if (L->num_nodes == 0) return EDGEHIP_OK;
    const float scale = 1.0f / (1.0f - dp->p);
    const uint8_t* mask = dp->masks ? dp->masks + (size_t)slot * (size_t)L->num_nodes * 64 : nullptr;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_dropout, TR_GRID((int64_t)L->num_nodes * 16, 256), dim3(256), 0,
                       eh_stream(stream), *L, x, dp->p, scale, dp->seed, slot, mask, y));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ================================================================================================
// J. The whole differentiable network in two calls.
//    edgehip_train_forward / edgehip_train_backward queue the SAME ops as above, in the order of
//    TGNN_degree_guided.forward (layers/layers.py:315-383), from one host call each: ~190 / ~330 launches are
//    issued back to back by the library instead of one Python autograd node per op (which bounded the step
//    at ~25 us per op on the host).  Everything the backward needs stays in the caller's workspace between
//    the two calls; gradients are ADDED into the buffers of `G` (a parameter struct whose pointers address
//    gradient storage), so the caller zero-fills one flat gradient buffer per step and nothing is copied.
// ================================================================================================
#define TR_RC(x) do { int rc__ = (x); if (rc__ != EDGEHIP_OK) return rc__; } while (0)
#define TR_HIP(x) do { if ((x) != hipSuccess) return EDGEHIP_ERR_LAUNCH; } while (0)

struct tr_bump {
    float* base; size_t off;
    float* take(size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return base ? base + o : nullptr; }
};

struct tr_plan {
    // saved by the forward
    float *X0, *a0, *hs[EDGEHIP_MAX_LAYERS + 1];
    float* hd[EDGEHIP_MAX_LAYERS + 1];      // dropout(hs[i]) = the GRU hidden state carried into layer i (layers.py:361); hd[0] = hs[0]
    float *pe, *m1, *m1s, *m2, *mt;
    float *gin[EDGEHIP_MAX_LAYERS + 1], *g1[EDGEHIP_MAX_LAYERS + 1], *g1s[EDGEHIP_MAX_LAYERS + 1], *cx[EDGEHIP_MAX_LAYERS + 1];
    float *tl[EDGEHIP_MAX_LAYERS], *x[EDGEHIP_MAX_LAYERS], *q[EDGEHIP_MAX_LAYERS], *k[EDGEHIP_MAX_LAYERS], *v[EDGEHIP_MAX_LAYERS];
    float *sk[EDGEHIP_MAX_LAYERS], *O[EDGEHIP_MAX_LAYERS], *lse[EDGEHIP_MAX_LAYERS], *conv[EDGEHIP_MAX_LAYERS], *xs[EDGEHIP_MAX_LAYERS];
    float *gi[EDGEHIP_MAX_LAYERS], *gh[EDGEHIP_MAX_LAYERS], *cin[EDGEHIP_MAX_LAYERS], *c1[EDGEHIP_MAX_LAYERS], *c1s[EDGEHIP_MAX_LAYERS];
    float *nctx[EDGEHIP_MAX_LAYERS];
    float *zin, *n1, *n1s, *n2, *n2s, *z, *e, *f1, *f1s, *f2, *f2s;
    // backward temporaries
    float *wt_node_in, *wt_mlp2, *wt_g0, *wt_g2, *wt_c0, *wt_c2, *wt_ih, *wt_hh, *wt_time[EDGEHIP_MAX_LAYERS];
    float *wt_q[EDGEHIP_MAX_LAYERS], *wt_k[EDGEHIP_MAX_LAYERS], *wt_v[EDGEHIP_MAX_LAYERS], *wt_sk[EDGEHIP_MAX_LAYERS];
    float *wt_no0, *wt_no2, *wt_no4, *wt_f0, *wt_f2;
    // gradient buffers.  Every buffer a weight-gradient product reads is written ONCE per backward call, so those
    // products can run on the side stream while the main stream walks on (no write-after-read hazards)
    float *dA0, *dA1, *dA2, *dz, *dn, *dn2, *dzin, *dh, *dh2, *dX0, *Dbuf;
    float *dcxj[EDGEHIP_MAX_LAYERS + 1], *dg1[EDGEHIP_MAX_LAYERS + 1];
    float *dnctx[EDGEHIP_MAX_LAYERS], *dc1[EDGEHIP_MAX_LAYERS], *dcin[EDGEHIP_MAX_LAYERS], *dgi[EDGEHIP_MAX_LAYERS];
    float *dgh[EDGEHIP_MAX_LAYERS], *dconv[EDGEHIP_MAX_LAYERS], *dq[EDGEHIP_MAX_LAYERS], *dk[EDGEHIP_MAX_LAYERS];
    float *dv[EDGEHIP_MAX_LAYERS], *dx[EDGEHIP_MAX_LAYERS], *dtl[EDGEHIP_MAX_LAYERS];
    float *dmt, *dm1, *dgin0, *dxp;
    size_t floats;
};

static void tr_make_plan(tr_plan& p, const edgehip_layout* L, const edgehip_gnn_params* P, int64_t A, float* base) {
    tr_bump b{base, 0};
    const size_t Nn = L->num_nodes, B = L->num_graphs, nl = P->num_layers;
    int hmax = 1;
    for (size_t i = 0; i < nl; ++i) hmax = P->heads[i] > hmax ? P->heads[i] : hmax;
    p.X0 = b.take(Nn * 192); p.a0 = b.take(Nn * 64);
    for (size_t i = 0; i <= nl; ++i) p.hs[i] = b.take(Nn * 64);
    p.hd[0] = p.hs[0]; p.hd[nl] = nullptr;
    for (size_t i = 1; i < nl; ++i) p.hd[i] = b.take(Nn * 64);
    p.pe = b.take(B * 64); p.m1 = b.take(B * 256); p.m1s = b.take(B * 256); p.m2 = b.take(B * 64); p.mt = b.take(B * 64);
    for (size_t i = 0; i <= nl; ++i) {
        p.gin[i] = b.take(B * 64); p.g1[i] = b.take(B * 256); p.g1s[i] = b.take(B * 256); p.cx[i] = b.take(B * 64);
    }
    for (size_t i = 0; i < nl; ++i) {
        const size_t hd = (size_t)P->heads[i] * 64;
        p.tl[i] = b.take(B * 64); p.x[i] = b.take(Nn * 128);
        p.q[i] = b.take(Nn * hd); p.k[i] = b.take(Nn * hd); p.v[i] = b.take(Nn * hd); p.sk[i] = b.take(Nn * 64);
        p.O[i] = b.take(Nn * hd); p.lse[i] = b.take(Nn * P->heads[i]); p.conv[i] = b.take(Nn * 64); p.xs[i] = b.take(Nn * 64);
        p.gi[i] = b.take(Nn * 192); p.gh[i] = b.take(Nn * 192); p.cin[i] = b.take(Nn * 128);
        p.c1[i] = b.take(Nn * 256); p.c1s[i] = b.take(Nn * 256); p.nctx[i] = b.take(Nn * 64);
    }
    p.zin = b.take(Nn * 256); p.n1 = b.take(Nn * 128); p.n1s = b.take(Nn * 128); p.n2 = b.take(Nn * 128); p.n2s = b.take(Nn * 128);
    p.z = b.take(Nn * 128);
    p.e = b.take(A * 128); p.f1 = b.take(A * 128); p.f1s = b.take(A * 128); p.f2 = b.take(A * 64); p.f2s = b.take(A * 64);
    p.wt_node_in = b.take(64 * 192); p.wt_mlp2 = b.take(64 * 256); p.wt_g0 = b.take(256 * 64); p.wt_g2 = b.take(64 * 256);
    p.wt_c0 = b.take(256 * 128); p.wt_c2 = b.take(64 * 256); p.wt_ih = b.take(192 * 64); p.wt_hh = b.take(192 * 64);
    for (size_t i = 0; i < nl; ++i) {
        const size_t hd = (size_t)P->heads[i] * 64;
        p.wt_time[i] = b.take(64 * 64); p.wt_q[i] = b.take(hd * 128); p.wt_k[i] = b.take(hd * 128); p.wt_v[i] = b.take(hd * 128);
        p.wt_sk[i] = b.take(64 * 128);
    }
    p.wt_no0 = b.take(128 * 256); p.wt_no2 = b.take(128 * 128); p.wt_no4 = b.take(128 * 128); p.wt_f0 = b.take(128 * 128);
    p.wt_f2 = b.take(64 * 128);
    p.dA0 = b.take(A * 64); p.dA1 = b.take(A * 128); p.dA2 = b.take(A * 128);
    p.dz = b.take(Nn * 128); p.dn = b.take(Nn * 128); p.dn2 = b.take(Nn * 128); p.dzin = b.take(Nn * 256);
    p.dh = b.take(Nn * 64); p.dh2 = b.take(Nn * 64); p.dX0 = b.take(Nn * 192); p.Dbuf = b.take(Nn * hmax);
    for (size_t i = 0; i <= nl; ++i) { p.dcxj[i] = b.take(B * 64); p.dg1[i] = b.take(B * 256); }
    for (size_t i = 0; i < nl; ++i) {
        const size_t hd = (size_t)P->heads[i] * 64;
        p.dnctx[i] = b.take(Nn * 64); p.dc1[i] = b.take(Nn * 256); p.dcin[i] = b.take(Nn * 128);
        p.dgi[i] = b.take(Nn * 192); p.dgh[i] = b.take(Nn * 192); p.dconv[i] = b.take(Nn * 64);
        p.dq[i] = b.take(Nn * hd); p.dk[i] = b.take(Nn * hd); p.dv[i] = b.take(Nn * hd);
        p.dx[i] = b.take(Nn * 128); p.dtl[i] = b.take(B * 64);
    }
    p.dmt = b.take(B * 64); p.dm1 = b.take(B * 256); p.dgin0 = b.take(B * 64);
    p.dxp = b.take(Nn * 128 * 3);
    p.floats = b.off;
}

static int tr_check_net(const edgehip_gnn_params* P) {
    if (!P || P->dim != 64 || P->num_layers < 1 || P->num_layers > EDGEHIP_MAX_LAYERS || P->max_degree <= 0) return EDGEHIP_ERR_UNSUPPORTED;
    for (int i = 0; i < P->num_layers; ++i) {
        int h = P->heads[i];
        if (h != 1 && h != 2 && h != 4 && h != 8) return EDGEHIP_ERR_UNSUPPORTED;
    }
    return EDGEHIP_OK;
}

extern "C" size_t edgehip_train_workspace_bytes(const edgehip_layout* L, const edgehip_gnn_params* P, int64_t A) {
    if (!L || tr_check_net(P) != EDGEHIP_OK || A < 0) return 0;
    tr_plan p;
    tr_make_plan(p, L, P, A, nullptr);
    return p.floats * sizeof(float) + 256;
}

static float* tr_ws_base(void* ws) { return reinterpret_cast<float*>(((uintptr_t)ws + 255) & ~(uintptr_t)255); }

__global__ void k_tr_pos_gather(int B, const int32_t* t_graph, const float* pos_emb, int T, float* pe) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 64) return;
    int g = idx >> 6, c = idx & 63;
    int t = t_graph[g];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    pe[idx] = pos_emb[(int64_t)t * 64 + c];
}

// Y = X W^T + b (pre-activation kept), optional Y2 = SiLU(Y)
static eh_gemm_desc tr_desc_lin(const float* X, int64_t M, int K, const float* W, const float* b, int N, float* Y, float* Y2) {
    eh_gemm_desc d = {};
    d.A = X; d.lda = K; d.K1 = K; d.W = W; d.K = K; d.N = N; d.bias = b; d.Y = Y; d.ldy = N; d.M = M; d.prof_id = EH_K_TRAIN;
    d.Y2 = Y2;
    return d;
}
static int tr_lin(const float* X, int64_t M, int K, const float* W, const float* b, int N, float* Y, float* Y2, hipStream_t s) {
    if (M == 0) return EDGEHIP_OK;
    return eh_launch_gemm(tr_desc_lin(X, M, K, W, b, N, Y, Y2), s);
}
// dX[M,K] (+)= dY[M,N] W[N,K], WT = W^T as [K,N]; accum 0 store, 1 add, 2 atomic add
static eh_gemm_desc tr_desc_dx(const float* dY, int64_t M, int N, const float* WT, int K, float* dX, int accum,
                               const float* silu_pre = nullptr) {
    eh_gemm_desc d = {};
    d.dact = silu_pre;
    d.A = dY; d.lda = N; d.K1 = N; d.W = WT; d.K = N; d.N = K; d.Y = dX; d.ldy = K; d.M = M; d.prof_id = EH_K_TRAIN;
    d.accum = accum;
    return d;
}
// silu_pre: dX is the gradient of SiLU(pre) -> multiply by SiLU'(pre) in the epilogue (accum must be 0)
static int tr_lin_dx(const float* dY, int64_t M, int N, const float* WT, int K, float* dX, int accum, hipStream_t s,
                     const float* silu_pre = nullptr) {
    if (M == 0) return EDGEHIP_OK;
    return eh_launch_gemm(tr_desc_dx(dY, M, N, WT, K, dX, accum, silu_pre), s);
}
// every weight of the network in one launch: blockIdx.z selects the matrix
struct tr_tentry { const float* W; float* WT; int N, K; };
#define TR_TMAX (13 + 5 * EDGEHIP_MAX_LAYERS)
struct tr_tlist { tr_tentry e[TR_TMAX]; int n; };
__global__ void k_tr_transpose_many(tr_tlist tl) {
    __shared__ float t[32][33];
    const tr_tentry e = tl.e[blockIdx.z];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    if (n0 >= e.N || k0 >= e.K) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < e.N && k0 + tx < e.K) t[r][tx] = e.W[(int64_t)(n0 + r) * e.K + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < e.K && n0 + tx < e.N) e.WT[(int64_t)(k0 + r) * e.N + n0 + tx] = t[tx][r];
}
static int tr_transpose(const float* W, int N, int K, float* WT, hipStream_t s) {
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_transpose, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, s, W, N, K, WT));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
static int tr_slice(int64_t M, const float* src, int ld, int col0, int ncol, float* dst, int acc, hipStream_t s) {
    if (M == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_slice, TR_GRID(M * ncol, 256), dim3(256), 0, s, M, src, ld, col0, ncol, dst, acc));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
// out[B,64] (+)= per-graph column sums of src[:, col0:col0+64]
static int tr_segsum(const edgehip_layout* L, const float* src, int ld, int col0, float* out, int acc, hipStream_t s) {
    if (L->num_graphs == 0) return EDGEHIP_OK;
    if (!acc) TR_HIP(hipMemsetAsync(out, 0, (size_t)L->num_graphs * 64 * sizeof(float), s));
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_sum, tr_seg_grid(L), dim3(256), 0, s, L->node_ptr, src, ld, col0,
                       (const float*)nullptr, 0, out));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
static int tr_segmean_bwd(const edgehip_layout* L, const float* dout, float* dx, int acc, hipStream_t s) {
    if (L->num_nodes == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_mean_bwd, TR_GRID((int64_t)L->num_nodes * 64, 256), dim3(256), 0, s,
                       L->num_nodes, L->node_graph, L->node_ptr, dout, dx, acc));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// global_mlp on the B per-graph rows as ONE launch each way (block per graph, block-cooperative GEMVs of gemv.h):
// as MFMA products these were two launches of a single 4-row tile each, 9-13 us of pure latency apiece.
__global__ __launch_bounds__(256) void k_tr_global_fwd(const float* __restrict__ gin, const float* __restrict__ W0,
                                                       const float* __restrict__ b0, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, float* __restrict__ g1,
                                                       float* __restrict__ g1s, float* __restrict__ cx) {
    __shared__ __attribute__((aligned(16))) float v[64], pre[256], hid[256], out[64];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) v[tid] = gin[(size_t)g * 64 + tid];
    __syncthreads();
    eh_block_gemv<64, false>(W0, 64, v, nullptr, 0, nullptr, b0, 0, 256, pre, 0);
    __syncthreads();
    {
        const float x = pre[tid], y = tr_act(1, x);
        g1[(size_t)g * 256 + tid] = x; g1s[(size_t)g * 256 + tid] = y; hid[tid] = y;
    }
    __syncthreads();
    eh_block_gemv<256, false>(W2, 256, hid, nullptr, 0, nullptr, b2, 0, 64, out, 0);
    __syncthreads();
    if (tid < 64) cx[(size_t)g * 64 + tid] = out[tid];
}
// d cx -> d g1 (pre-activation gradient, kept for the weight-gradient product) -> d gin; WT2 = W2^T [256,64], WT0 = W0^T [64,256]
__global__ __launch_bounds__(256) void k_tr_global_bwd(const float* __restrict__ dcx, const float* __restrict__ g1,
                                                       const float* __restrict__ WT2, const float* __restrict__ WT0,
                                                       float* __restrict__ dg1, float* __restrict__ dgin) {
    __shared__ __attribute__((aligned(16))) float v[64], pre[256], hid[256], out[64];
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) v[tid] = dcx[(size_t)g * 64 + tid];
    __syncthreads();
    eh_block_gemv<64, false>(WT2, 64, v, nullptr, 0, nullptr, nullptr, 0, 256, pre, 0);
    __syncthreads();
    {
        const float d = pre[tid] * tr_act_grad(1, g1[(size_t)g * 256 + tid]);
        dg1[(size_t)g * 256 + tid] = d; hid[tid] = d;
    }
    __syncthreads();
    eh_block_gemv<256, false>(WT0, 256, hid, nullptr, 0, nullptr, nullptr, 0, 64, out, 0);
    __syncthreads();
    if (tid < 64) dgin[(size_t)g * 64 + tid] = out[tid];
}

// global_mlp(gin) -> cx   (layers.py:343,368)
static int tr_global_fwd(const edgehip_gnn_params* P, const tr_plan& p, int j, int64_t B, hipStream_t s) {
    if (B == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_global_fwd, dim3((unsigned)B), dim3(256), 0, s, p.gin[j], P->global_w0,
                       P->global_b0, P->global_w2, P->global_b2, p.g1[j], p.g1s[j], p.cx[j]));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
// Side stream for the weight-gradient products.  Nothing on the main stream waits for dW/db until the step ends,
// so they are queued on a second HIP stream behind an event of the launch that produced their dY and overlap
// with the latency-bound chain of input-gradient products (one per device, created on first use).
#include <mutex>
struct tr_aux_t { hipStream_t side = nullptr; hipEvent_t ev[160]; int nev = 0; bool ok = false; bool own = false; };
static tr_aux_t g_tr_aux[16];
static std::mutex g_tr_aux_mu;
static bool tr_aux_events(tr_aux_t& a) {
    for (; a.nev < 160; ++a.nev)
        if (hipEventCreateWithFlags(&a.ev[a.nev], hipEventDisableTiming) != hipSuccess) return false;
    return true;
}
static tr_aux_t* tr_aux() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_tr_aux_mu);
    tr_aux_t& a = g_tr_aux[dev];
    if (!a.ok) {
        if (!a.side) {
            if (hipStreamCreateWithFlags(&a.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
            a.own = true;
        }
        if (!tr_aux_events(a)) return nullptr;
        a.ok = true;
    }
    return &a;
}
// A process has few hardware queues (4 by default) and HIP deals streams to them in creation order: a caller that
// keeps side streams of its own (the Python host: the sampler's sub-batch streams) hands one of them over, so that
// training and sampling together do not create more streams than there are queues - two streams on one queue run
// one after the other (measured: three-stream sampling after training in the same process 162 -> 114 graphs/s).
extern "C" int edgehip_train_set_side_stream(void* stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return EDGEHIP_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_tr_aux_mu);
    tr_aux_t& a = g_tr_aux[dev];
    if (a.side && a.own && a.side != (hipStream_t)stream) {
        (void)hipStreamSynchronize(a.side);
        (void)hipStreamDestroy(a.side);
    }
    a.side = (hipStream_t)stream;          // NULL: the library creates its own at the next backward
    a.own = false;
    a.ok = false;
    return EDGEHIP_OK;
}
struct tr_dw_ctx {
    hipStream_t main; tr_aux_t* aux; int used;
    // dW[N,K] += dY^T X, db[N] += colsum(dY) on the side stream, ordered after everything queued on main so far
    int dw(const float* dY, const float* X, int64_t M, int N, int K, const float* dW, const float* db) {
        hipStream_t s = main;
        if (aux && used < aux->nev - 1) {               // the last event is kept for join()
            if (hipEventRecord(aux->ev[used], main) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
            if (hipStreamWaitEvent(aux->side, aux->ev[used], 0) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
            ++used;
            s = aux->side;
        }
        return edgehip_linear_bwd_weight(dY, X, M, N, K, (float*)dW, (float*)db, (void*)s);
    }
    int join() {                       // main waits for every product queued on the side stream
        if (!aux || used == 0) return EDGEHIP_OK;
        hipEvent_t e = aux->ev[aux->nev - 1];
        if (hipEventRecord(e, aux->side) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        if (hipStreamWaitEvent(main, e, 0) != hipSuccess) return EDGEHIP_ERR_LAUNCH;
        return EDGEHIP_OK;
    }
};

// d cx[j] (p.dcxj[j]) -> d gin[j] (into ds); weight gradients accumulate
static int tr_global_bwd(const edgehip_gnn_params* G, const tr_plan& p, int j, int64_t B, float* ds, tr_dw_ctx& W, void* st) {
    hipStream_t s = eh_stream(st);
    const float* dcx = p.dcxj[j];
    TR_RC(W.dw(dcx, p.g1s[j], B, 64, 256, G->global_w2, G->global_b2));
    if (B > 0) {
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_global_bwd, dim3((unsigned)B), dim3(256), 0, s, dcx, p.g1[j], p.wt_g2, p.wt_g0,
                           p.dg1[j], ds));
        EH_LAUNCH_CHECK();
    }
    return W.dw(p.dg1[j], p.gin[j], B, 256, 64, G->global_w0, G->global_b0);
}

// node part of the forward: everything up to the per-node output z.  It needs no active pairs, only a CAPACITY A_cap
// that fixes the workspace layout; a caller can queue it before it knows how many pairs the step has.
static int tr_forward_nodes(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t, const int32_t* deg_t,
                            const int32_t* d0, const uint8_t* flags, const int32_t* t_graph, int64_t A_cap, void* ws,
                            size_t ws_bytes, void* stream, const edgehip_dropout* dp = nullptr) {
    const bool drop = tr_dp_on(dp);
    if (dp && !(dp->p >= 0.f && dp->p < 1.f)) return EDGEHIP_ERR_ARG;
    EH_CHECK_ARG(L && P && adj_t && deg_t && d0 && flags && t_graph && ws && A_cap >= 0);
    TR_RC(tr_check_net(P));
    if (ws_bytes < edgehip_train_workspace_bytes(L, P, A_cap)) return EDGEHIP_ERR_WORKSPACE;
    hipStream_t s = eh_stream(stream);
    const int64_t Nn = L->num_nodes, B = L->num_graphs;
    const int nl = P->num_layers;
    if (Nn == 0 || B == 0) return EDGEHIP_OK;
    tr_plan p;
    tr_make_plan(p, L, P, A_cap, tr_ws_base(ws));
    // input embeddings, node_in (layers.py:326-336)
    TR_RC(edgehip_train_embed_fwd(L, deg_t, d0, flags, P->max_degree, P->embedding_t_w, P->embedding_t_b, P->embedding_0_w,
                                  P->embedding_0_b, P->embedding_sel_w, p.X0, stream));
    TR_RC(tr_lin(p.X0, Nn, 192, P->node_in_w, P->node_in_b, 64, p.a0, p.hs[0], s));
    // time embedding: Mish(mlp(sinusoid))  (layers.py:338-339)
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_pos_gather, TR_GRID(B * 64, 256), dim3(256), 0, s, (int)B, t_graph, P->pos_emb,
                       P->num_steps, p.pe));
    EH_LAUNCH_CHECK();
    TR_RC(tr_lin(p.pe, B, 64, P->mlp_w0, P->mlp_b0, 256, p.m1, p.m1s, s));
    TR_RC(tr_lin(p.m1s, B, 256, P->mlp_w2, P->mlp_b2, 64, p.m2, nullptr, s));
    TR_RC(edgehip_act_fwd(2, p.m2, B * 64, p.mt, stream));
    // first context (layers.py:342-343)
    TR_RC(edgehip_segment_mean_fwd(L, p.hs[0], nullptr, p.gin[0], stream));
    TR_RC(tr_global_fwd(P, p, 0, B, s));
    {   // the per-layer time embeddings depend on mt only: one grouped launch (layers.py:349)
        eh_gemm_desc ds[EDGEHIP_MAX_LAYERS];
        for (int i = 0; i < nl; ++i) ds[i] = tr_desc_lin(p.mt, B, 64, P->time_w[i], P->time_b[i], 64, p.tl[i], nullptr);
        TR_RC(eh_launch_gemm_group(ds, nl, s));
    }
    for (int i = 0; i < nl; ++i) {
        const int h = P->heads[i], hd = h * 64;
        TR_RC(edgehip_assemble_fwd(L, p.hs[i], i > 0 ? p.cx[i] : nullptr, p.tl[i], nullptr, 64, p.x[i], stream));
        {   // q, k, v, skip read x; the hidden half of the GRU reads h: five independent products, one launch
            eh_gemm_desc ds[5] = {tr_desc_lin(p.x[i], Nn, 128, P->query_w[i], P->query_b[i], hd, p.q[i], nullptr),
                                  tr_desc_lin(p.x[i], Nn, 128, P->key_w[i], P->key_b[i], hd, p.k[i], nullptr),
                                  tr_desc_lin(p.x[i], Nn, 128, P->value_w[i], P->value_b[i], hd, p.v[i], nullptr),
                                  tr_desc_lin(p.x[i], Nn, 128, P->skip_w[i], P->skip_b[i], 64, p.sk[i], nullptr),
                                  tr_desc_lin(drop ? p.hd[i] : p.hs[i], Nn, 64, P->gru_w_hh, P->gru_b_hh, 192, p.gh[i], nullptr)};
            TR_RC(eh_launch_gemm_group(ds, 5, s));
        }
        TR_RC(edgehip_attn_fwd(L, adj_t, h, p.q[i], p.k[i], p.v[i], p.sk[i], p.conv[i], p.O[i], p.lse[i], stream));
        TR_RC(edgehip_act_fwd(1, p.conv[i], Nn * 64, p.xs[i], stream));
        if (drop) TR_RC(edgehip_dropout_fwd(L, p.xs[i], dp, 2 * i, p.xs[i], stream));                  // layers.py:357
        TR_RC(tr_lin(p.xs[i], Nn, 64, P->gru_w_ih, P->gru_b_ih, 192, p.gi[i], nullptr, s));
        TR_RC(edgehip_gru_gate_fwd(p.gi[i], p.gh[i], drop ? p.hd[i] : p.hs[i], Nn, p.hs[i + 1], stream));
        // layers.py:361: only the hidden state carried to the next layer is dropped, not the GRU output `nodes`
        // (the last layer's dropped state is never read)
        if (drop && i + 1 < nl) TR_RC(edgehip_dropout_fwd(L, p.hs[i + 1], dp, 2 * i + 1, p.hd[i + 1], stream));
        TR_RC(edgehip_assemble_fwd(L, p.hs[i + 1], nullptr, p.cx[i], nullptr, 64, p.cin[i], stream));
        TR_RC(tr_lin(p.cin[i], Nn, 128, P->context_w0, P->context_b0, 256, p.c1[i], p.c1s[i], s));
        TR_RC(tr_lin(p.c1s[i], Nn, 256, P->context_w2, P->context_b2, 64, p.nctx[i], nullptr, s));
        TR_RC(edgehip_segment_mean_fwd(L, p.nctx[i], p.cx[i], p.gin[i + 1], stream));
        TR_RC(tr_global_fwd(P, p, i + 1, B, s));
    }
    // node_out_mlp on cat[nodes + contexts, X0] (layers.py:371-378), pair gather, final_out, CE
    TR_RC(edgehip_assemble_fwd(L, p.hs[nl], p.cx[nl], nullptr, p.X0, 192, p.zin, stream));
    TR_RC(tr_lin(p.zin, Nn, 256, P->node_out_w0, P->node_out_b0, 128, p.n1, p.n1s, s));
    TR_RC(tr_lin(p.n1s, Nn, 128, P->node_out_w2, P->node_out_b2, 128, p.n2, p.n2s, s));
    return tr_lin(p.n2s, Nn, 128, P->node_out_w4, P->node_out_b4, 128, p.z, nullptr, s);
}

// pair part: gather of the active pairs, final_out, CE (layers.py:380-381, active.py:157-175); A <= A_cap
static int tr_forward_pairs(const edgehip_layout* L, const edgehip_gnn_params* P, const int64_t* row, const int64_t* col,
                            int64_t A, int64_t A_cap, const uint8_t* tgt, const int32_t* pair_graph, float* ce_graph,
                            float* logits, void* ws, size_t ws_bytes, void* stream) {
    EH_CHECK_ARG(L && P && ce_graph && ws && A >= 0 && A <= A_cap);
    EH_CHECK_ARG(A == 0 || (row && col && tgt && pair_graph && logits));
    TR_RC(tr_check_net(P));
    if (ws_bytes < edgehip_train_workspace_bytes(L, P, A_cap)) return EDGEHIP_ERR_WORKSPACE;
    hipStream_t s = eh_stream(stream);
    TR_HIP(hipMemsetAsync(ce_graph, 0, (size_t)L->num_graphs * sizeof(float), s));
    if (L->num_nodes == 0 || L->num_graphs == 0 || A == 0) return EDGEHIP_OK;
    tr_plan p;
    tr_make_plan(p, L, P, A_cap, tr_ws_base(ws));
    TR_RC(edgehip_pair_gather_add_fwd(p.z, 128, row, col, A, p.e, stream));
    TR_RC(tr_lin(p.e, A, 128, P->final_w0, P->final_b0, 128, p.f1, p.f1s, s));
    TR_RC(tr_lin(p.f1s, A, 128, P->final_w2, P->final_b2, 64, p.f2, p.f2s, s));
    return edgehip_edge_ce_fwd(p.f2s, P->final_w4, P->final_b4, tgt, pair_graph, A, logits, ce_graph, stream);
}

extern "C" int edgehip_train_forward_nodes(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                                           const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                           const int32_t* t_graph, int64_t A_cap, void* ws, size_t ws_bytes, void* stream) {
    return tr_forward_nodes(L, P, adj_t, deg_t, d0, flags, t_graph, A_cap, ws, ws_bytes, stream);
}
extern "C" int edgehip_train_forward_nodes_dp(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                                              const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                              const int32_t* t_graph, int64_t A_cap, const edgehip_dropout* dp, void* ws,
                                              size_t ws_bytes, void* stream) {
    return tr_forward_nodes(L, P, adj_t, deg_t, d0, flags, t_graph, A_cap, ws, ws_bytes, stream, dp);
}
extern "C" int edgehip_train_forward_pairs(const edgehip_layout* L, const edgehip_gnn_params* P, const int64_t* row,
                                           const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                                           const int32_t* pair_graph, float* ce_graph, float* logits, void* ws,
                                           size_t ws_bytes, void* stream) {
    return tr_forward_pairs(L, P, row, col, A, A_cap, tgt, pair_graph, ce_graph, logits, ws, ws_bytes, stream);
}
extern "C" int edgehip_train_forward(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                                     const int32_t* deg_t, const int32_t* d0, const uint8_t* flags, const int32_t* t_graph,
                                     const int64_t* row, const int64_t* col, int64_t A, const uint8_t* tgt,
                                     const int32_t* pair_graph, float* ce_graph, float* logits, void* ws, size_t ws_bytes,
                                     void* stream) {
    EH_CHECK_ARG(ce_graph && A >= 0);
    TR_RC(tr_forward_nodes(L, P, adj_t, deg_t, d0, flags, t_graph, A, ws, ws_bytes, stream));
    return tr_forward_pairs(L, P, row, col, A, A, tgt, pair_graph, ce_graph, logits, ws, ws_bytes, stream);
}

extern "C" int edgehip_train_backward_cap(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                                          const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                          const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                                          const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream);
extern "C" int edgehip_train_backward(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                                      const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                      const int64_t* row, const int64_t* col, int64_t A, const uint8_t* tgt,
                                      const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream) {
    return edgehip_train_backward_cap(L, P, G, adj_t, deg_t, d0, flags, row, col, A, A, tgt, pair_graph, gce, ws, ws_bytes, stream);
}
static int tr_backward(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                       const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                       const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                       const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream,
                       const edgehip_dropout* dp);
extern "C" int edgehip_train_backward_cap(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                                          const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                          const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                                          const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream) {
    return tr_backward(L, P, G, adj_t, deg_t, d0, flags, row, col, A, A_cap, tgt, pair_graph, gce, ws, ws_bytes, stream, nullptr);
}
extern "C" int edgehip_train_backward_dp(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                                         const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                                         const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                                         const int32_t* pair_graph, const float* gce, const edgehip_dropout* dp, void* ws,
                                         size_t ws_bytes, void* stream) {
    return tr_backward(L, P, G, adj_t, deg_t, d0, flags, row, col, A, A_cap, tgt, pair_graph, gce, ws, ws_bytes, stream, dp);
}
static int tr_backward(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                       const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                       const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                       const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream,
                       const edgehip_dropout* dp) {
    const bool drop = tr_dp_on(dp);
    if (dp && !(dp->p >= 0.f && dp->p < 1.f)) return EDGEHIP_ERR_ARG;
    EH_CHECK_ARG(L && P && G && adj_t && deg_t && d0 && flags && gce && ws && A >= 0 && A <= A_cap);
    EH_CHECK_ARG(A == 0 || (row && col && tgt && pair_graph));
    TR_RC(tr_check_net(P));
    if (ws_bytes < edgehip_train_workspace_bytes(L, P, A_cap)) return EDGEHIP_ERR_WORKSPACE;
    hipStream_t s = eh_stream(stream);
    const int64_t Nn = L->num_nodes, B = L->num_graphs;
    const int nl = P->num_layers;
    if (Nn == 0 || B == 0 || A == 0) return EDGEHIP_OK;          // no active pair: the loss does not depend on the network
    tr_plan p;
    tr_make_plan(p, L, P, A_cap, tr_ws_base(ws));
#define GW(f) ((float*)G->f)
    {   // weights transposed once per call, all in one launch (dX = dY W runs as a linear layer on W^T)
        tr_tlist tl = {};
        auto add = [&tl](const float* W, int N, int K, float* WT) { tl.e[tl.n++] = tr_tentry{W, WT, N, K}; };
        add(P->node_in_w, 64, 192, p.wt_node_in); add(P->mlp_w2, 64, 256, p.wt_mlp2);
        add(P->global_w0, 256, 64, p.wt_g0); add(P->global_w2, 64, 256, p.wt_g2);
        add(P->context_w0, 256, 128, p.wt_c0); add(P->context_w2, 64, 256, p.wt_c2);
        add(P->gru_w_ih, 192, 64, p.wt_ih); add(P->gru_w_hh, 192, 64, p.wt_hh);
        for (int i = 0; i < nl; ++i) {
            const int hd = P->heads[i] * 64;
            add(P->time_w[i], 64, 64, p.wt_time[i]); add(P->query_w[i], hd, 128, p.wt_q[i]); add(P->key_w[i], hd, 128, p.wt_k[i]);
            add(P->value_w[i], hd, 128, p.wt_v[i]); add(P->skip_w[i], 64, 128, p.wt_sk[i]);
        }
        add(P->node_out_w0, 128, 256, p.wt_no0); add(P->node_out_w2, 128, 128, p.wt_no2); add(P->node_out_w4, 128, 128, p.wt_no4);
        add(P->final_w0, 128, 128, p.wt_f0); add(P->final_w2, 64, 128, p.wt_f2);
        int mk = 0, mn = 0;
        for (int i = 0; i < tl.n; ++i) { mk = tl.e[i].K > mk ? tl.e[i].K : mk; mn = tl.e[i].N > mn ? tl.e[i].N : mn; }
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_transpose_many, dim3((mk + 31) / 32, (mn + 31) / 32, tl.n), dim3(256), 0, s, tl));
        EH_LAUNCH_CHECK();
    }
    tr_dw_ctx W{s, tr_aux(), 0};
    // CE + final_out (layers.py:381, active.py:157-175)
    TR_RC(edgehip_edge_ce_bwd(p.f2s, P->final_w4, P->final_b4, tgt, pair_graph, gce, A, p.dA0, GW(final_w4), GW(final_b4), stream));
    TR_RC(edgehip_act_bwd(1, p.f2, p.dA0, A * 64, p.dA0, stream));
    TR_RC(W.dw(p.dA0, p.f1s, A, 64, 128, G->final_w2, G->final_b2));
    TR_RC(tr_lin_dx(p.dA0, A, 64, p.wt_f2, 128, p.dA1, 0, s, p.f1));
        TR_RC(W.dw(p.dA1, p.e, A, 128, 128, G->final_w0, G->final_b0));
    TR_RC(tr_lin_dx(p.dA1, A, 128, p.wt_f0, 128, p.dA2, 0, s));                       // d e
    TR_HIP(hipMemsetAsync(p.dz, 0, (size_t)Nn * 128 * sizeof(float), s));
    TR_RC(edgehip_pair_gather_add_bwd(p.dA2, 128, row, col, A, p.dz, stream));
    // node_out_mlp
    TR_RC(W.dw(p.dz, p.n2s, Nn, 128, 128, G->node_out_w4, G->node_out_b4));
    TR_RC(tr_lin_dx(p.dz, Nn, 128, p.wt_no4, 128, p.dn, 0, s, p.n2));
        TR_RC(W.dw(p.dn, p.n1s, Nn, 128, 128, G->node_out_w2, G->node_out_b2));
    TR_RC(tr_lin_dx(p.dn, Nn, 128, p.wt_no2, 128, p.dn2, 0, s, p.n1));
        TR_RC(W.dw(p.dn2, p.zin, Nn, 128, 256, G->node_out_w0, G->node_out_b0));
    TR_RC(tr_lin_dx(p.dn2, Nn, 128, p.wt_no0, 256, p.dzin, 0, s));
    // zin = [h + cx | X0]
    float *dh = p.dh, *dh2 = p.dh2;
    TR_RC(tr_slice(Nn, p.dzin, 256, 0, 64, dh, 0, s));
    TR_RC(tr_segsum(L, p.dzin, 256, 0, p.dcxj[nl], 0, s));
    TR_RC(tr_slice(Nn, p.dzin, 256, 64, 192, p.dX0, 0, s));
    for (int i = nl - 1; i >= 0; --i) {
        const int h = P->heads[i], hd = h * 64;
        float* ds = p.dcxj[i];                                                         // running d cx[i]
        // cx[i+1] = global_mlp(mean(nctx_i) + cx[i])
        TR_RC(tr_global_bwd(G, p, i + 1, B, ds, W, stream));
        TR_RC(tr_segmean_bwd(L, ds, p.dnctx[i], 0, s));
        TR_RC(W.dw(p.dnctx[i], p.c1s[i], Nn, 64, 256, G->context_w2, G->context_b2));
        TR_RC(tr_lin_dx(p.dnctx[i], Nn, 64, p.wt_c2, 256, p.dc1[i], 0, s, p.c1[i]));
                TR_RC(W.dw(p.dc1[i], p.cin[i], Nn, 256, 128, G->context_w0, G->context_b0));
        TR_RC(tr_lin_dx(p.dc1[i], Nn, 256, p.wt_c0, 128, p.dcin[i], 0, s));          // d cin = [d h' | d cx[i] rows]
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_assemble_bwd128, tr_seg_grid(L), dim3(256), 0, s, L->node_ptr, p.dcin[i], dh,
                           (float*)nullptr, ds));
        EH_LAUNCH_CHECK();
        // GRU (layers.py:360)
        const float* hid = drop ? p.hd[i] : p.hs[i];                                 // the hidden state the GRU saw
        TR_RC(edgehip_gru_gate_bwd(p.gi[i], p.gh[i], hid, dh, Nn, p.dgi[i], p.dgh[i], dh2, stream));
        TR_RC(W.dw(p.dgi[i], p.xs[i], Nn, 192, 64, G->gru_w_ih, G->gru_b_ih));
        TR_RC(W.dw(p.dgh[i], hid, Nn, 192, 64, G->gru_w_hh, G->gru_b_hh));
        {
            eh_gemm_desc ds[2] = {tr_desc_dx(p.dgi[i], Nn, 192, p.wt_ih, 64, p.dconv[i], 0, p.conv[i]),   // through act{i}
                                  tr_desc_dx(p.dgh[i], Nn, 192, p.wt_hh, 64, dh2, 1)};
            TR_RC(eh_launch_gemm_group(ds, 2, s));
        }
        if (drop) {
            // d conv = d xs * keep/(1-p) * SiLU'(conv): the mask factor commutes with the SiLU' of the epilogue above;
            // dh2 so far is the gradient of the DROPPED hidden state hd[i]: back through its mask to hs[i] (i > 0)
            TR_RC(edgehip_dropout_fwd(L, p.dconv[i], dp, 2 * i, p.dconv[i], stream));
            if (i > 0) TR_RC(edgehip_dropout_fwd(L, dh2, dp, 2 * (i - 1) + 1, dh2, stream));
        }
        // TransformerConv (layers.py:354)
        TR_RC(edgehip_attn_bwd(L, adj_t, h, p.q[i], p.k[i], p.v[i], p.O[i], p.lse[i], p.dconv[i], p.Dbuf, p.dq[i], p.dk[i],
                               p.dv[i], stream));
        TR_RC(W.dw(p.dq[i], p.x[i], Nn, hd, 128, G->query_w[i], G->query_b[i]));
        TR_RC(W.dw(p.dk[i], p.x[i], Nn, hd, 128, G->key_w[i], G->key_b[i]));
        TR_RC(W.dw(p.dv[i], p.x[i], Nn, hd, 128, G->value_w[i], G->value_b[i]));
        TR_RC(W.dw(p.dconv[i], p.x[i], Nn, 64, 128, G->skip_w[i], G->skip_b[i]));
        {   // the four input-gradient products in one launch, each into its own buffer, then one pass adds them
            // (float atomics onto a shared dx were 2.5 M atomics per layer: slower than the products themselves)
            float* part1 = p.dxp, *part2 = p.dxp + (size_t)Nn * 128, *part3 = p.dxp + (size_t)Nn * 256;
            eh_gemm_desc ds[4] = {tr_desc_dx(p.dq[i], Nn, hd, p.wt_q[i], 128, p.dx[i], 0),
                                  tr_desc_dx(p.dk[i], Nn, hd, p.wt_k[i], 128, part1, 0),
                                  tr_desc_dx(p.dv[i], Nn, hd, p.wt_v[i], 128, part2, 0),
                                  tr_desc_dx(p.dconv[i], Nn, 64, p.wt_sk[i], 128, part3, 0)};
            TR_RC(eh_launch_gemm_group(ds, 4, s));
            const int64_t n4 = Nn * 32;
            EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_add4, TR_GRID(n4, 256), dim3(256), 0, s, n4, p.dx[i], part1, part2, part3));
            EH_LAUNCH_CHECK();
        }
        // x = [h + cx (i > 0) | tl_g]
        TR_HIP(hipMemsetAsync(p.dtl[i], 0, (size_t)B * 64 * sizeof(float), s));
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_assemble_bwd128, tr_seg_grid(L), dim3(256), 0, s, L->node_ptr, p.dx[i], dh2,
                           i > 0 ? ds : (float*)nullptr, p.dtl[i]));
        EH_LAUNCH_CHECK();
        TR_RC(W.dw(p.dtl[i], p.mt, B, 64, 64, G->time_w[i], G->time_b[i]));
        float* t0 = dh; dh = dh2; dh2 = t0;
    }
    {   // d mt = sum_i d tl_i W_time_i: the per-layer products are independent, one grouped launch (atomics onto zero)
        TR_HIP(hipMemsetAsync(p.dmt, 0, (size_t)B * 64 * sizeof(float), s));
        eh_gemm_desc ds[EDGEHIP_MAX_LAYERS];
        for (int i = 0; i < nl; ++i) ds[i] = tr_desc_dx(p.dtl[i], B, 64, p.wt_time[i], 64, p.dmt, 2);
        TR_RC(eh_launch_gemm_group(ds, nl, s));
    }
    // first context, node_in, embeddings
    float* dgin0 = p.dgin0;
    TR_RC(tr_global_bwd(G, p, 0, B, dgin0, W, stream));
    TR_RC(tr_segmean_bwd(L, dgin0, dh, 1, s));
    TR_RC(edgehip_act_bwd(1, p.a0, dh, Nn * 64, dh, stream));
    TR_RC(W.dw(dh, p.X0, Nn, 64, 192, G->node_in_w, G->node_in_b));
    TR_RC(tr_lin_dx(dh, Nn, 64, p.wt_node_in, 192, p.dX0, 1, s));
    TR_RC(edgehip_train_embed_bwd(L, deg_t, d0, flags, P->max_degree, p.dX0, GW(embedding_t_w), GW(embedding_t_b),
                                  GW(embedding_0_w), GW(embedding_0_b), GW(embedding_sel_w), stream));
    // time mlp
    TR_RC(edgehip_act_bwd(2, p.m2, p.dmt, B * 64, p.dmt, stream));
    TR_RC(W.dw(p.dmt, p.m1s, B, 64, 256, G->mlp_w2, G->mlp_b2));
    TR_RC(tr_lin_dx(p.dmt, B, 64, p.wt_mlp2, 256, p.dm1, 0, s, p.m1));
        TR_RC(W.dw(p.dm1, p.pe, B, 256, 64, G->mlp_w0, G->mlp_b0));
    return W.join();
#undef GW
}
