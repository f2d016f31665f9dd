// train.hip - training ops of the path: forward kernels that keep what the backward needs, and their
// backward twins ("_bwd twins for 6-11 in training").  The training step is a composition
// of these ops by edgehip/train.py (one torch.autograd.Function per op, buffers owned by the caller);
// unlike the sampling path nothing is folded, so every gradient lands directly on a reference parameter.
// Batches are small here (polblogs: 4 graphs), so the ops favour plain, checkable kernels.
#include "common.h"
#include "gemm.h"
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

// column sums of src[rows of graph g][col0 .. col0+64) into out[g][64] (scaled); one block per graph
__global__ __launch_bounds__(256) void k_tr_segment_sum(const int32_t* node_ptr, const float* src, int ld, int col0,
                                                        const float* add, int mean, float* out) {
    __shared__ float part[4][64];
    const int g = blockIdx.x, c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i0 = node_ptr[g], i1 = node_ptr[g + 1];
    float s = 0.f;
    for (int i = i0 + r; i < i1; i += 4) s += src[(int64_t)i * ld + col0 + c];
    part[r][c] = s;
    __syncthreads();
    if (r == 0) {
        s = part[0][c] + part[1][c] + part[2][c] + part[3][c];
        if (mean) s = (i1 > i0) ? s / (float)(i1 - i0) : 0.f;
        if (add) s += add[g * 64 + c];
        out[g * 64 + c] = s;
    }
}

// dst[i][0..ncol) = src[i][col0 .. col0+ncol)   (strided slice -> contiguous)
__global__ void k_tr_slice(int64_t M, const float* src, int ld, int col0, int ncol, float* dst) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * ncol) return;
    int64_t i = idx / ncol;
    int c = (int)(idx % ncol);
    dst[idx] = src[i * ld + col0 + c];
}

// dx[i][c] = dout[g(i)][c] / n_g
__global__ void k_tr_segment_mean_bwd(int Nn, const int32_t* node_graph, const int32_t* node_ptr, const float* dout,
                                      float* dx) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Nn * 64) return;
    int i = idx >> 6, c = idx & 63;
    int g = node_graph[i];
    dx[idx] = dout[g * 64 + c] / (float)(node_ptr[g + 1] - node_ptr[g]);
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

template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_fwd(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                        const float* V, const float* skip, float* out, float* O, float* lse) {
    __shared__ float red[H][64];
    const int i = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = threadIdx.x & 63;
    const tr_row row = tr_row_of(L, adj, i);
    const int HD = H * 64;
    const float q = Q[(int64_t)i * HD + h * 64 + c];
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int w = 0; w < row.words; ++w) {
        uint32_t bits = row.bits[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            int64_t j = row.node0 + w * 32 + b;
            float kj = K[j * HD + h * 64 + c], vj = V[j * HD + h * 64 + c];
            float s = tr_wave_sum(q * kj) * 0.125f;
            float mn = fmaxf(m, s);
            float sc = expf(m - mn), p = expf(s - mn);
            l = l * sc + p;
            acc = acc * sc + p * vj;
            m = mn;
        }
    }
    float o = l > 0.f ? acc / (l + 1e-16f) : 0.f;          // pyg softmax: w / (sum + 1e-16)
    O[(int64_t)i * HD + h * 64 + c] = o;
    if (c == 0) lse[(int64_t)i * H + h] = l > 0.f ? m + logf(l + 1e-16f) : 0.f;
    red[h][c] = o;
    __syncthreads();
    if (h == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) s += red[k][c];
        out[(int64_t)i * 64 + c] = s * (1.0f / H) + skip[(int64_t)i * 64 + c];
    }
}

// dQ and the row terms D[i][h] = dO_i^h . O_i^h
template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_bwd_q(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                          const float* V, const float* O, const float* lse,
                                                          const float* dout, float* Dbuf, float* dQ) {
    const int i = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = threadIdx.x & 63;
    const tr_row row = tr_row_of(L, adj, i);
    const int HD = H * 64;
    const float q = Q[(int64_t)i * HD + h * 64 + c];
    const float go = dout[(int64_t)i * 64 + c] * (1.0f / H);
    const float D = tr_wave_sum(go * O[(int64_t)i * HD + h * 64 + c]);
    const float ls = lse[(int64_t)i * H + h];
    if (c == 0) Dbuf[(int64_t)i * H + h] = D;
    float dq = 0.f;
    for (int w = 0; w < row.words; ++w) {
        uint32_t bits = row.bits[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            int64_t j = row.node0 + w * 32 + b;
            float kj = K[j * HD + h * 64 + c], vj = V[j * HD + h * 64 + c];
            float s = tr_wave_sum(q * kj) * 0.125f;
            float a = expf(s - ls);
            float dov = tr_wave_sum(go * vj);
            dq += a * (dov - D) * kj * 0.125f;
        }
    }
    dQ[(int64_t)i * HD + h * 64 + c] = dq;
}

// dK, dV of node j: the adjacency is symmetric, so the targets that attend to j are row j's bits
template <int H>
__global__ __launch_bounds__(64 * H) void k_tr_attn_bwd_kv(edgehip_layout L, const uint32_t* adj, const float* Q, const float* K,
                                                           const float* V, const float* lse, const float* Dbuf,
                                                           const float* dout, float* dK, float* dV) {
    const int j = blockIdx.x;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = threadIdx.x & 63;
    const tr_row row = tr_row_of(L, adj, j);
    const int HD = H * 64;
    const float kj = K[(int64_t)j * HD + h * 64 + c], vj = V[(int64_t)j * HD + h * 64 + c];
    float dk = 0.f, dv = 0.f;
    for (int w = 0; w < row.words; ++w) {
        uint32_t bits = row.bits[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            int64_t i = row.node0 + w * 32 + b;
            float qi = Q[i * HD + h * 64 + c];
            float go = dout[i * 64 + c] * (1.0f / H);
            float s = tr_wave_sum(qi * kj) * 0.125f;
            float a = expf(s - lse[i * H + h]);
            float dov = tr_wave_sum(go * vj);
            dk += a * (dov - Dbuf[i * H + h]) * qi * 0.125f;
            dv += a * go;
        }
    }
    dK[(int64_t)j * HD + h * 64 + c] = dk;
    dV[(int64_t)j * HD + h * 64 + c] = dv;
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
__global__ __launch_bounds__(256) void k_tr_ce_fwd(int64_t A, const float* e2, const float* W4, const float* b4,
                                                   const uint8_t* tgt, const int32_t* pair_graph, float* logits,
                                                   float* ce_graph) {
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (k >= A) return;
    float x = e2[k * 64 + c];
    float l0 = tr_wave_sum(x * W4[c]) + b4[0];
    float l1 = tr_wave_sum(x * W4[64 + c]) + b4[1];
    if (c == 0) {
        logits[k * 2] = l0; logits[k * 2 + 1] = l1;
        float m = fmaxf(l0, l1);
        float lz = m + logf(expf(l0 - m) + expf(l1 - m));
        float ce = -((tgt[k] ? l1 : l0) - lz);
        atomicAdd(ce_graph + pair_graph[k], ce);
    }
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
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
    int64_t want = (2048 + tiles - 1) / tiles;                       // blocks along M for ~2048 blocks in all
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
    EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_slice, TR_GRID(n, 256), dim3(256), 0, s, (int64_t)L->num_nodes, dout, ld, 0, 64, dh));
    if (dctx)
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_sum, dim3(L->num_graphs), dim3(256), 0, s, L->node_ptr, dout, ld, 0,
                           (const float*)nullptr, 0, dctx));
    if (dtail_graph)
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_segment_sum, dim3(L->num_graphs), dim3(256), 0, s, L->node_ptr, dout, ld, 64,
                           (const float*)nullptr, 0, dtail_graph));
    if (dtail_rows && Kt > 0) {
        int64_t m = (int64_t)L->num_nodes * Kt;
        EH_PROF(EH_K_TRAIN, s, hipLaunchKernelGGL(k_tr_slice, TR_GRID(m, 256), dim3(256), 0, s, (int64_t)L->num_nodes, dout, ld, 64, Kt,
                           dtail_rows));
    }
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_segment_mean_fwd(const edgehip_layout* L, const float* x, const float* add, float* out, void* stream) {
    EH_CHECK_ARG(L && x && out);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_segment_sum, dim3(L->num_graphs), dim3(256), 0, eh_stream(stream),
                       L->node_ptr, x, 64, 0, add, 1, out));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_segment_mean_bwd(const edgehip_layout* L, const float* dout, float* dx, void* stream) {
    EH_CHECK_ARG(L && dout && dx);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    int64_t n = (int64_t)L->num_nodes * 64;
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_segment_mean_bwd, TR_GRID(n, 256), dim3(256), 0, eh_stream(stream),
                       L->num_nodes, L->node_graph, L->node_ptr, dout, dx));
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
    EH_PROF(EH_K_TRAIN, eh_stream(stream), hipLaunchKernelGGL(k_tr_ce_fwd, TR_GRID(A, 4), dim3(256), 0, eh_stream(stream),
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
