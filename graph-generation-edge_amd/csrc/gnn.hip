// gnn.hip - TGNN_degree_guided forward (layers/layers.py:315-383 of the reference) on the
// bit-matrix state, plus the active-pair tail (final_out + Bernoulli re-sampling / cross-entropy).
//
// Algebra (DESIGN.md "Algebra"): with x_i = h_i + c_g (node state + per-graph context) and the
// layer input cat[x_i, te_g] (te_g: per-graph time embedding), every product with a per-graph
// vector becomes a per-graph bias, and
//   score(i,j,hd) = q_i,hd . k_j,hd / 8  ==  qt_i,hd . h_j + (terms constant in j, softmax-invariant)
//   qt_i,hd = (1/8) Wk_hd^T (Wq_hd x_i + ..)            -> one 64 -> 64h GEMM (folded weight Mcat)
//   out_i   = mean_hd Wv_hd (sum_j alpha h_j) + ..      -> one (64h+64) -> 64 GEMM (folded Ocat)
// so keys/values are never materialised and the attention gathers 256-byte rows of h.
#include "gemm.h"
#include "prof.h"

#define EH_D 64

// --------------------------------------------------------------------------------------------
// folded-weight buffer layout (floats)
// --------------------------------------------------------------------------------------------
struct eh_fold_layer { size_t Mcat, Mtcat, mbcat, Ocat, Vbbar, bvbar, Vsum; };
struct eh_fold_map {
    size_t nin_at, nin_a0, nin_as;   // [64], [64], [2][64]
    size_t W4f, b4f;                 // [128,128], [128]
    eh_fold_layer layer[EDGEHIP_MAX_LAYERS];
    size_t total;
};

static eh_fold_map eh_make_fold_map(const edgehip_gnn_params* P) {
    eh_fold_map m;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    m.nin_at = take(64); m.nin_a0 = take(64); m.nin_as = take(128);
    m.W4f = take(128 * 128); m.b4f = take(128);
    for (int l = 0; l < P->num_layers && l < EDGEHIP_MAX_LAYERS; ++l) {
        size_t H = (size_t)P->heads[l] * EH_D;
        m.layer[l].Mcat = take(H * 64); m.layer[l].Mtcat = take(H * 64); m.layer[l].mbcat = take(H);
        m.layer[l].Ocat = take(64 * (H + 64)); m.layer[l].Vbbar = take(64 * 64); m.layer[l].bvbar = take(64);
        m.layer[l].Vsum = take(64 * 64);
    }
    m.total = o;
    return m;
}

static int eh_check_params(const edgehip_gnn_params* P) {
    if (!P) return EDGEHIP_ERR_ARG;
    if (P->dim != EH_D || P->num_layers < 1 || P->num_layers > EDGEHIP_MAX_LAYERS || P->max_degree < 1 ||
        P->num_steps < 1)
        return EDGEHIP_ERR_UNSUPPORTED;
    for (int l = 0; l < P->num_layers; ++l) {
        int h = P->heads[l];
        if (!(h == 1 || h == 2 || h == 4 || h == 8)) return EDGEHIP_ERR_UNSUPPORTED;
    }
    return EDGEHIP_OK;
}

extern "C" size_t edgehip_gnn_folded_floats(const edgehip_gnn_params* P) {
    if (eh_check_params(P) != EDGEHIP_OK) return 0;
    return eh_make_fold_map(P).total;
}

// Mcat / Mtcat / mbcat : one block per output row o = hd*64 + d, 64 threads (e)
__global__ __launch_bounds__(64) void k_fold_attn(const float* __restrict__ qw, const float* __restrict__ qb,
                                                  const float* __restrict__ kw, float* __restrict__ Mcat,
                                                  float* __restrict__ Mtcat, float* __restrict__ mbcat) {
    int o = blockIdx.x, e = threadIdx.x;
    int hd = o >> 6, d = o & 63;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < 64; ++c) {
        float kk = kw[(size_t)(hd * 64 + c) * 128 + d];
        s0 += kk * qw[(size_t)(hd * 64 + c) * 128 + e];
        s1 += kk * qw[(size_t)(hd * 64 + c) * 128 + 64 + e];
        s2 += kk * qb[hd * 64 + c];
    }
    Mcat[(size_t)o * 64 + e] = 0.125f * s0;     // 1/sqrt(64)
    Mtcat[(size_t)o * 64 + e] = 0.125f * s1;
    if (e == 0) mbcat[o] = 0.125f * s2;
}

// Ocat / Vbbar / bvbar / Vsum : one block per output channel c, 64 threads (e)
__global__ __launch_bounds__(64) void k_fold_out(int h, const float* __restrict__ vw, const float* __restrict__ vb,
                                                 const float* __restrict__ sw, float* __restrict__ Ocat,
                                                 float* __restrict__ Vbbar, float* __restrict__ bvbar,
                                                 float* __restrict__ Vsum) {
    int c = blockIdx.x, e = threadIdx.x;
    int H = h * 64;
    float inv = 1.0f / (float)h;   // h is a power of two: exact
    float sb = 0.f, sa = 0.f, bb = 0.f;
    for (int hd = 0; hd < h; ++hd) {
        float wa = vw[(size_t)(hd * 64 + c) * 128 + e];
        Ocat[(size_t)c * (H + 64) + hd * 64 + e] = wa * inv;
        sa += wa;
        sb += vw[(size_t)(hd * 64 + c) * 128 + 64 + e];
        bb += vb[hd * 64 + c];
    }
    Ocat[(size_t)c * (H + 64) + H + e] = sw[(size_t)c * 128 + e];
    Vbbar[c * 64 + e] = sb * inv;
    Vsum[c * 64 + e] = sa * inv;
    if (e == 0) bvbar[c] = bb * inv;
}

// node_in fold: nodes0 = silu(at*x_t + a0*x_0 + as[sel])
__global__ __launch_bounds__(64) void k_fold_nin(edgehip_gnn_params P, float* __restrict__ at, float* __restrict__ a0,
                                                 float* __restrict__ as) {
    int o = threadIdx.x;
    const float* W = P.node_in_w + (size_t)o * 192;
    float st = 0.f, s0 = 0.f, sc = P.node_in_b[o], e0 = 0.f, e1 = 0.f;
    for (int c = 0; c < 64; ++c) {
        st += W[c] * P.embedding_t_w[c];
        sc += W[c] * P.embedding_t_b[c];
        s0 += W[64 + c] * P.embedding_0_w[c];
        sc += W[64 + c] * P.embedding_0_b[c];
        e0 += W[128 + c] * P.embedding_sel_w[c];
        e1 += W[128 + c] * P.embedding_sel_w[64 + c];
    }
    at[o] = st; a0[o] = s0; as[o] = e0 + sc; as[64 + o] = e1 + sc;
}

// W4f = final_w0 . node_out_w4 ; b4f = final_w0 . node_out_b4   (128 blocks x 128 threads)
__global__ __launch_bounds__(128) void k_fold_tail(const float* __restrict__ f0, const float* __restrict__ w4,
                                                   const float* __restrict__ b4, float* __restrict__ W4f,
                                                   float* __restrict__ b4f) {
    int i = blockIdx.x, j = threadIdx.x;
    float s = 0.f, sb = 0.f;
    for (int c = 0; c < 128; ++c) {
        float a = f0[i * 128 + c];
        s += a * w4[c * 128 + j];
        sb += a * b4[c];
    }
    W4f[i * 128 + j] = s;
    if (j == 0) b4f[i] = sb;
}

extern "C" int edgehip_gnn_prepare(const edgehip_gnn_params* P, float* folded, void* stream) {
    int rc = eh_check_params(P);
    if (rc != EDGEHIP_OK) return rc;
    EH_CHECK_ARG(folded);
    eh_fold_map m = eh_make_fold_map(P);
    hipStream_t s = eh_stream(stream);
    hipLaunchKernelGGL(k_fold_nin, dim3(1), dim3(64), 0, s, *P, folded + m.nin_at, folded + m.nin_a0, folded + m.nin_as);
    hipLaunchKernelGGL(k_fold_tail, dim3(128), dim3(128), 0, s, P->final_w0, P->node_out_w4, P->node_out_b4,
                       folded + m.W4f, folded + m.b4f);
    for (int l = 0; l < P->num_layers; ++l) {
        int h = P->heads[l];
        hipLaunchKernelGGL(k_fold_attn, dim3(h * 64), dim3(64), 0, s, P->query_w[l], P->query_b[l], P->key_w[l],
                           folded + m.layer[l].Mcat, folded + m.layer[l].Mtcat, folded + m.layer[l].mbcat);
        hipLaunchKernelGGL(k_fold_out, dim3(64), dim3(64), 0, s, h, P->value_w[l], P->value_b[l], P->skip_w[l],
                           folded + m.layer[l].Ocat, folded + m.layer[l].Vbbar, folded + m.layer[l].bvbar,
                           folded + m.layer[l].Vsum);
    }
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// --------------------------------------------------------------------------------------------
// workspace carving
// --------------------------------------------------------------------------------------------
struct eh_ws {
    float *hA, *hB, *Qt, *Z, *Mb, *part, *te, *ctx, *qbias, *obias0, *obias1, *cbias, *Xact, *U1, *U2;
    // step-level scratch
    int32_t *deg, *act_nodes, *act_count, *kptr, *tptr, *t_graph, *deg2;
    int64_t* aptr;
    uint8_t* flags;
    float *node_repr, *tile_partial;
    size_t bytes;
};

static eh_ws eh_carve(const edgehip_layout* L, const edgehip_gnn_params* P, void* base) {
    eh_ws w;
    char* p = (char*)base;
    size_t o = 0;
    auto take = [&](size_t bytes) { void* r = p ? (void*)(p + o) : nullptr; o += (bytes + 255) & ~(size_t)255; return r; };
    size_t Nn = (size_t)L->num_nodes, B = (size_t)L->num_graphs, NT = (size_t)L->num_tiles;
    int hmax = 1;
    for (int l = 0; l < P->num_layers; ++l) hmax = P->heads[l] > hmax ? P->heads[l] : hmax;
    size_t H = (size_t)hmax * EH_D;
    size_t NnP = Nn + 32;   // slack row block so clamped fragment loads stay inside
    w.hA = (float*)take(NnP * 64 * 4); w.hB = (float*)take(NnP * 64 * 4);
    w.Qt = (float*)take(NnP * H * 4); w.Z = (float*)take(NnP * H * 4);
    w.Mb = (float*)take(NnP * 64 * 4);
    w.part = (float*)take((NT + 1) * 256 * 4);
    w.te = (float*)take((B + 1) * EDGEHIP_MAX_LAYERS * 64 * 4);
    w.ctx = (float*)take((B + 1) * 64 * 4);
    w.qbias = (float*)take((B + 1) * H * 4);
    w.obias0 = (float*)take((B + 1) * 64 * 4); w.obias1 = (float*)take((B + 1) * 64 * 4);
    w.cbias = (float*)take((B + 1) * 256 * 4);
    w.Xact = (float*)take(NnP * 256 * 4);
    w.U1 = (float*)take(NnP * 128 * 4); w.U2 = (float*)take(NnP * 128 * 4);
    w.deg = (int32_t*)take(NnP * 4); w.deg2 = (int32_t*)take(NnP * 4);
    w.act_nodes = (int32_t*)take(NnP * 4);
    w.act_count = (int32_t*)take((B + 1) * 4); w.kptr = (int32_t*)take((B + 2) * 4);
    w.tptr = (int32_t*)take((B + 2) * 4); w.t_graph = (int32_t*)take((B + 1) * 4);
    w.aptr = (int64_t*)take((B + 2) * 8);
    w.flags = (uint8_t*)take(NnP);
    w.node_repr = (float*)take(NnP * 128 * 4);
    // pair tiles: sum ceil(A_b/32) <= P/32 + B ; bounded here by the dense pair count
    size_t maxpairs = 0;
    (void)maxpairs;
    w.tile_partial = nullptr;
    w.bytes = o;
    return w;
}

extern "C" size_t edgehip_gnn_workspace_bytes(const edgehip_layout* L, const edgehip_gnn_params* P) {
    if (!L || eh_check_params(P) != EDGEHIP_OK) return 0;
    return eh_carve(L, P, nullptr).bytes;
}

// --------------------------------------------------------------------------------------------
// per-graph time embedding: te[l] = W_l mish(mlp(pos_emb[t])) + b_l     (layers.py:338-339,349)
// grid = 1 (shared t) or B
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_time_embed(edgehip_gnn_params P, const int32_t* __restrict__ t_graph,
                                                    float* __restrict__ te) {
    __shared__ float pe[64], hid[256], tt[64];
    int g = blockIdx.x, tid = threadIdx.x;
    int t = t_graph[g];
    if (tid < 64) pe[tid] = P.pos_emb[(size_t)t * 64 + tid];
    __syncthreads();
    {
        const float* w = P.mlp_w0 + (size_t)tid * 64;
        float s = P.mlp_b0[tid];
        for (int k = 0; k < 64; ++k) s += w[k] * pe[k];
        hid[tid] = eh_silu(s);
    }
    __syncthreads();
    if (tid < 64) {
        const float* w = P.mlp_w2 + (size_t)tid * 256;
        float s = P.mlp_b2[tid];
        for (int k = 0; k < 256; ++k) s += w[k] * hid[k];
        // Mish: x * tanh(softplus(x)), softplus with torch's threshold 20
        float sp = s > 20.0f ? s : log1pf(expf(s));
        tt[tid] = s * tanhf(sp);
    }
    __syncthreads();
    for (int o = tid; o < P.num_layers * 64; o += 256) {
        int l = o >> 6, d = o & 63;
        const float* w = P.time_w[l] + (size_t)d * 64;
        float s = P.time_b[l][d];
        for (int k = 0; k < 64; ++k) s += w[k] * tt[k];
        te[((size_t)g * EDGEHIP_MAX_LAYERS + l) * 64 + d] = s;
    }
}

// --------------------------------------------------------------------------------------------
// node_in: h0 = silu(at*x_t + a0*x_0 + as[sel]) and per-tile column sums (layers.py:326-342)
// one wave per node tile, lane = channel
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_in(edgehip_layout L, int max_degree, const float* __restrict__ at,
                                                 const float* __restrict__ a0, const float* __restrict__ as,
                                                 const int32_t* __restrict__ deg, const int32_t* __restrict__ d0,
                                                 const uint8_t* __restrict__ flags, float* __restrict__ h0,
                                                 float* __restrict__ part) {
    int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (tile >= L.num_tiles) return;
    int row0 = L.tile_row0[tile], rows = L.tile_rows[tile];
    float vat = at[lane], va0 = a0[lane], vs0 = as[lane], vs1 = as[64 + lane];
    float md = (float)max_degree;
    float sum = 0.f;
    for (int r = 0; r < rows; ++r) {
        int i = row0 + r;
        int dt = deg[i];
        dt = dt > max_degree + 1 ? max_degree + 1 : dt;
        float xt = (float)dt / md, x0 = (float)d0[i] / md;
        float v = eh_silu(vat * xt + va0 * x0 + (flags[i] ? vs1 : vs0));
        h0[(size_t)i * 64 + lane] = v;
        sum += v;
    }
    part[(size_t)tile * 256 + lane] = sum;
}

// --------------------------------------------------------------------------------------------
// per-graph update between layers (block per graph, 256 threads):
//   l == 0 : ctx = global_mlp(mean_i h0)                                   (layers.py:342-343)
//   l >= 1 : ctx = global_mlp(ctx + W_c2 mean_i s_i + b_c2), s_i from k_ctx (layers.py:366-368)
//   then the per-graph bias vectors of layer l (if l < num_layers)
// --------------------------------------------------------------------------------------------
struct eh_layer_ptrs {
    const float *Mcat, *Mtcat, *mbcat, *Vbbar, *bvbar, *Vsum, *skip_w, *skip_b;
    int H;
};

__global__ __launch_bounds__(256) void k_graph_update(edgehip_layout L, edgehip_gnn_params P, int l, eh_layer_ptrs lp,
                                                      const float* __restrict__ part, const float* __restrict__ te,
                                                      int te_shared, float* __restrict__ ctx, float* __restrict__ qbias,
                                                      int ldq, float* __restrict__ obias0, float* __restrict__ obias1,
                                                      float* __restrict__ cbias) {
    __shared__ float mean[256], v[64], hid[256], cn[64], tev[64];
    int g = blockIdx.x, tid = threadIdx.x;
    int n = L.node_ptr[g + 1] - L.node_ptr[g];
    int t0 = L.tile_ptr[g], t1 = L.tile_ptr[g + 1];
    int width = (l == 0) ? 64 : 256;
    if (tid < width) {
        float s = 0.f;
        for (int t = t0; t < t1; ++t) s += part[(size_t)t * 256 + tid];
        mean[tid] = s / (float)(n > 0 ? n : 1);
    }
    __syncthreads();
    if (tid < 64) {
        if (l == 0) {
            v[tid] = mean[tid];
        } else {
            const float* w = P.context_w2 + (size_t)tid * 256;
            float s = P.context_b2[tid];
            for (int k = 0; k < 256; ++k) s += w[k] * mean[k];
            v[tid] = ctx[(size_t)g * 64 + tid] + s;
        }
    }
    __syncthreads();
    {
        const float* w = P.global_w0 + (size_t)tid * 64;
        float s = P.global_b0[tid];
        for (int k = 0; k < 64; ++k) s += w[k] * v[k];
        hid[tid] = eh_silu(s);
    }
    __syncthreads();
    if (tid < 64) {
        const float* w = P.global_w2 + (size_t)tid * 256;
        float s = P.global_b2[tid];
        for (int k = 0; k < 256; ++k) s += w[k] * hid[k];
        cn[tid] = s;
        ctx[(size_t)g * 64 + tid] = s;
        if (l < P.num_layers) tev[tid] = te[((size_t)(te_shared ? 0 : g) * EDGEHIP_MAX_LAYERS + l) * 64 + tid];
    }
    __syncthreads();
    if (l >= P.num_layers) return;
    const bool addx = l > 0;    // layer 0 sees x = h0 (no context added yet)
    for (int o = tid; o < lp.H; o += 256) {
        const float* m = lp.Mcat + (size_t)o * 64;
        const float* mt = lp.Mtcat + (size_t)o * 64;
        float s = lp.mbcat[o];
        for (int k = 0; k < 64; ++k) s += mt[k] * tev[k];
        if (addx) for (int k = 0; k < 64; ++k) s += m[k] * cn[k];
        qbias[(size_t)g * ldq + o] = s;
    }
    if (tid < 64) {
        const float* sw = lp.skip_w + (size_t)tid * 128;
        float s = lp.skip_b[tid];
        for (int k = 0; k < 64; ++k) s += sw[64 + k] * tev[k];
        if (addx) for (int k = 0; k < 64; ++k) s += sw[k] * cn[k];
        obias0[(size_t)g * 64 + tid] = s;
    } else if (tid < 128) {
        int c = tid - 64;
        float s = lp.bvbar[c];
        for (int k = 0; k < 64; ++k) s += lp.Vbbar[c * 64 + k] * tev[k];
        if (addx) for (int k = 0; k < 64; ++k) s += lp.Vsum[c * 64 + k] * cn[k];
        obias1[(size_t)g * 64 + c] = s;
    }
    {
        const float* w = P.context_w0 + (size_t)tid * 128 + 64;
        float s = P.context_b0[tid];
        for (int k = 0; k < 64; ++k) s += w[k] * cn[k];
        cbias[(size_t)g * 256 + tid] = s;
    }
}

// --------------------------------------------------------------------------------------------
// attention over the bit-matrix row of each target node (one wave per node).
// lane = 8*group + chunk: chunk c owns channels 8c..8c+7; group = slot*HEADS + head.  HEADS=8: the
// 8 groups are the 8 heads of ONE neighbour; HEADS<8: 8/HEADS neighbours are processed at once and
// their online-softmax states merged at the end.  Neighbour ids are popped from the row words with
// scalar bit ops (ascending j, like the reference's edge order).
// --------------------------------------------------------------------------------------------
// 8-lane group reductions with DPP (no LDS crossbar): quad xor-1, quad xor-2, then half-row mirror.
__device__ static inline float eh_dpp_f(float v, const int ctrl_sel) {
    int t;
    if (ctrl_sel == 0) t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
    else if (ctrl_sel == 1) t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true);
    else t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);
    return __builtin_bit_cast(float, t);
}
__device__ static inline float eh_group8_sum(float v) {
    v += eh_dpp_f(v, 0);
    v += eh_dpp_f(v, 1);
    v += eh_dpp_f(v, 2);
    return v;
}

// One target node: walk the set bits of its bit-matrix row (wave-uniform iterator, ascending j),
// keep U neighbours per slot in flight, online softmax.  q: this lane's 8 channels of qt for its head.
// out: this lane's 8 channels of z = sum_j alpha_j h_j (valid in every slot after the merge).
template <int HEADS>
__device__ static inline void eh_attend(const uint32_t* __restrict__ rowp, int stride, int node0, int self_node,
                                        const float* __restrict__ hsrc, const float (&q)[8], int lane, float (&out)[8]) {
    constexpr int NSLOT = 8 / HEADS;
    constexpr int U = 4;                 // neighbours in flight per slot: hubs are a latency chain otherwise
    const int grp = lane >> 3, c = lane & 7;
    const int slot = grp / HEADS;
    float acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    int w0 = -64;
    uint32_t word = 0u, wv = 0u;
    unsigned long long nz = 0ull;
    int jbase = 0;
    bool done = false;
    auto next = [&]() -> int {
        while (wv == 0u) {
            if (nz == 0ull) {
                if (done || w0 + 64 >= stride) { done = true; return -1; }
                w0 += 64;
                word = (w0 + lane < stride) ? rowp[w0 + lane] : 0u;
                nz = __ballot(word != 0u);
                continue;
            }
            int src = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
            wv = (uint32_t)__builtin_amdgcn_readlane((int)word, src);
            jbase = node0 + (w0 + src) * 32;
        }
        int j = jbase + __ffs((int)wv) - 1;
        wv &= wv - 1;
        return j;
    };
    while (true) {
        int myj[U];
        int first = -1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            myj[u] = -1;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                int j = next();
                if (u == 0 && s == 0) first = j;
                if (slot == s) myj[u] = j;
            }
        }
        if (first < 0) break;
        f32x4 xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4* xp = reinterpret_cast<const f32x4*>(hsrc + (size_t)(myj[u] >= 0 ? myj[u] : self_node) * 64 + c * 8);
            xa[u] = xp[0]; xb[u] = xp[1];
        }
        float sc[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t = 0.f;
#pragma unroll
            for (int d = 0; d < 4; ++d) t += q[d] * xa[u][d] + q[4 + d] * xb[u][d];
            t = eh_group8_sum(t);
            sc[u] = (myj[u] >= 0) ? t : -INFINITY;
            mn = fmaxf(mn, sc[u]);
        }
        if (mn > -INFINITY) {           // this slot saw at least one neighbour so far
            float resc = __expf(m - mn);      // m = -inf before the first neighbour: exp(-inf) = 0
            float p[U], ps = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) { p[u] = __expf(sc[u] - mn); ps += p[u]; }
            lsum = lsum * resc + ps;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float a0 = acc[d] * resc, a1 = acc[4 + d] * resc;
#pragma unroll
                for (int u = 0; u < U; ++u) { a0 += p[u] * xa[u][d]; a1 += p[u] * xb[u][d]; }
                acc[d] = a0; acc[4 + d] = a1;
            }
            m = mn;
        }
    }
    if (NSLOT > 1) {
#pragma unroll
        for (int off = 8 * HEADS; off < 64; off <<= 1) {
            float m2 = __shfl_xor(m, off), l2 = __shfl_xor(lsum, off);
            float mn = fmaxf(m, m2);
            float s1 = (m == -INFINITY) ? 0.f : __expf(m - mn);
            float s2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
            lsum = lsum * s1 + l2 * s2;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float a2 = __shfl_xor(acc[d], off);
                acc[d] = acc[d] * s1 + a2 * s2;
            }
            m = mn;
        }
    }
    float inv = 1.0f / (lsum + 1e-16f);
#pragma unroll
    for (int d = 0; d < 8; ++d) out[d] = acc[d] * inv;
}

template <int HEADS>
__global__ __launch_bounds__(256) void k_attention(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                   const float* __restrict__ hsrc, const float* __restrict__ Qt,
                                                   float* __restrict__ Z) {
    constexpr int H = HEADS * 64;
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; readfirstlane tells the compiler, so the row walk stays in SGPRs /
    // scalar branches instead of exec-masked vector code
    const int node = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (node >= L.num_nodes) return;
    const int b = L.node_graph[node];
    const int node0 = L.node_ptr[b];
    const int stride = L.adj_stride[b];
    const uint32_t* rowp = adj + L.adj_ptr[b] + (int64_t)(node - node0) * stride;
    const int grp = lane >> 3, c = lane & 7;
    const int head = grp % HEADS, slot = grp / HEADS;
    float q[8], z[8];
    {
        const f32x4* qp = reinterpret_cast<const f32x4*>(Qt + (size_t)node * H + head * 64 + c * 8);
        f32x4 q0 = qp[0], q1 = qp[1];
#pragma unroll
        for (int d = 0; d < 4; ++d) { q[d] = q0[d]; q[4 + d] = q1[d]; }
    }
    eh_attend<HEADS>(rowp, stride, node0, node, hsrc, q, lane, z);
    if (slot == 0) {
        f32x4 o0, o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) { o0[d] = z[d]; o1[d] = z[4 + d]; }
        f32x4* zp = reinterpret_cast<f32x4*>(Z + (size_t)node * H + head * 64 + c * 8);
        zp[0] = o0; zp[1] = o1;
    }
}

// --------------------------------------------------------------------------------------------
// GRU cell (torch.nn.GRU(64,64), one step; layers.py:360): one wave per node tile, MFMA for the six
// 64x64 products, gate math in the accumulator layout.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gru(edgehip_layout L, const float* __restrict__ w_ih,
                                             const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                             const float* __restrict__ b_hh, const float* __restrict__ xin,
                                             const float* __restrict__ hprev, float* __restrict__ hnew) {
    // weight fragments straight from L2 (each is used once per wave; staging them through LDS with two
    // barriers per column tile measured 15 % slower here)
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (tile >= L.num_tiles) return;
    const int row0 = L.tile_row0[tile], rows = L.tile_rows[tile];
    const int r = lane & 31, hh = lane >> 5;
    const int64_t arow = row0 + (r < rows ? r : rows - 1);
    eh_frag am, ah;
    eh_load_frag(am, xin + arow * 64 + hh * 32);
    eh_load_frag(ah, hprev + arow * 64 + hh * 32);
#pragma unroll 1
    for (int ct = 0; ct < 2; ++ct) {
        f32x16 ar = eh_zero16(), az = eh_zero16(), ain = eh_zero16(), ahn = eh_zero16();
        eh_frag bf;
        const int wr = ct * 32 + r;
        eh_load_frag(bf, w_ih + (size_t)(wr) * 64 + hh * 32);        eh_mma64(ar, am, bf);
        eh_load_frag(bf, w_hh + (size_t)(wr) * 64 + hh * 32);        eh_mma64(ar, ah, bf);
        eh_load_frag(bf, w_ih + (size_t)(64 + wr) * 64 + hh * 32);   eh_mma64(az, am, bf);
        eh_load_frag(bf, w_hh + (size_t)(64 + wr) * 64 + hh * 32);   eh_mma64(az, ah, bf);
        eh_load_frag(bf, w_ih + (size_t)(128 + wr) * 64 + hh * 32);  eh_mma64(ain, am, bf);
        eh_load_frag(bf, w_hh + (size_t)(128 + wr) * 64 + hh * 32);  eh_mma64(ahn, ah, bf);
        const int col = ct * 32 + r;
        const float bir = b_ih[col] + b_hh[col], biz = b_ih[64 + col] + b_hh[64 + col];
        const float bin = b_ih[128 + col], bhn = b_hh[128 + col];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int rr = eh_crow(i, lane);
            if (rr < rows) {
                size_t o = (size_t)(row0 + rr) * 64 + col;
                float rg = 1.0f / (1.0f + __expf(-(ar[i] + bir)));
                float zg = 1.0f / (1.0f + __expf(-(az[i] + biz)));
                float ng = tanhf((ain[i] + bin) + rg * (ahn[i] + bhn));
                float hp = hprev[o];
                hnew[o] = (1.0f - zg) * ng + zg * hp;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// context pre-activation: s_i = silu(W_c1[:, :64] h_i + cbias_g), per-tile column sums (layers.py:366-367)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ctx(edgehip_layout L, const float* __restrict__ w0 /*[256,128]*/,
                                             const float* __restrict__ cbias, const float* __restrict__ h,
                                             float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float wt[64 * EH_WPITCH];
    const int lane = threadIdx.x & 63;
    int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool live = tile < L.num_tiles;
    if (!live) tile = L.num_tiles - 1;
    const int row0 = L.tile_row0[tile], rows = L.tile_rows[tile], g = L.tile_graph[tile];
    const int r = lane & 31, hh = lane >> 5;
    const int64_t arow = row0 + (r < rows ? r : rows - 1);
    eh_frag a;
    eh_load_frag(a, h + arow * 64 + hh * 32);
#pragma unroll 1
    for (int np = 0; np < 4; ++np) {
        __syncthreads();
        eh_stage_w<64>(wt, w0, 128, np * 64, 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 acc = eh_zero16();
            eh_frag bf;
            eh_load_frag_lds(bf, wt, j * 32 + r, hh);
            eh_mma64(acc, a, bf);
            const int col = np * 64 + j * 32 + r;
            const float cb = cbias[(size_t)g * 256 + col];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rr = eh_crow(i, lane);
                float v = eh_silu(acc[i] + cb);
                s += (rr < rows) ? v : 0.f;
            }
            s += __shfl_xor(s, 32);
            if (live && lane < 32) part[(size_t)tile * 256 + col] = s;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Fused layer: for one node tile (32 nodes of one graph, one 256-thread block)
//   A  qt   = Mcat h + qbias_g                      -> LDS            (MFMA, 2h column tiles over 4 waves)
//   B  z    = attention over the bit rows           -> LDS in place   (8 targets per wave)
//   C  conv = silu(Ocat [z | h] + biases)           -> LDS            (MFMA, K split over the waves + reduce)
//   D  h'   = GRU(conv, h)                          -> global + LDS   (MFMA, gates split over the waves)
//   E  part = column sums of silu(Wc1 h' + cbias_g) -> global         (MFMA)
// qt and z (2 x 2 KB per node and layer) never touch HBM; per layer the kernel reads h (256 B/node),
// the bit row and the L2-resident neighbour rows, and writes h'.
// --------------------------------------------------------------------------------------------
struct eh_layer_args {
    const uint32_t* adj; const float* hprev; const int32_t* deg;
    const float* Mcat; const float* qbias; int ldq;
    const float* Ocat; const float* obias0; const float* obias1;
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    const float* cw0; const float* cbias;
    float* hnew; float* part;
};

#define EH_TP 68            // pitch (floats) of the 64-wide LDS tiles
#define EH_TSZ (32 * EH_TP)

template <int HEADS>
__global__ __launch_bounds__(256) void k_layer(edgehip_layout L, eh_layer_args a) {
    constexpr int H = HEADS * 64, QP = H + 4;
    constexpr int R1 = (32 * QP > 6 * EH_TSZ) ? 32 * QP : 6 * EH_TSZ;
    __shared__ __attribute__((aligned(16))) float lds[R1 + EH_TSZ];
    float* qz = lds;                    // [32][QP]   qt, then z in place              (phases A-C)
    float* pt = lds;                    // [4][32][68] K-split partials                (phase C end)
    float* xg = lds;                    // [2][2][16][64] GRU gate exchange            (phase D)
    float* mt = lds + 4 * EH_TSZ;       // [32][68]   conv
    float* ht = lds + 5 * EH_TSZ;       // [32][68]   h'
    float* hp = lds + R1;               // [32][68]   h (layer input / GRU hidden)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int tile = blockIdx.x;
    const int row0 = L.tile_row0[tile], rows = L.tile_rows[tile], g = L.tile_graph[tile];
    const int r = lane & 31, hh = lane >> 5;

    // ---- phase 0: h tile -> LDS (full-line loads)
#pragma unroll
    for (int i = tid; i < 512; i += 256) {
        int row = i >> 4, c4 = i & 15;
        int src = row0 + (row < rows ? row : rows - 1);
        *reinterpret_cast<f32x4*>(hp + row * EH_TP + c4 * 4) =
            *reinterpret_cast<const f32x4*>(a.hprev + (size_t)src * 64 + c4 * 4);
    }
    __syncthreads();

    // ---- phase A: qt
    {
        eh_frag af;
        eh_load_frag_lds(af, hp, r, hh);
        for (int nt = wave; nt < 2 * HEADS; nt += 4) {
            eh_frag bf;
            eh_load_frag(bf, a.Mcat + (size_t)(nt * 32 + r) * 64 + hh * 32);
            f32x16 acc = eh_zero16();
            eh_mma64(acc, af, bf);
            const int col = nt * 32 + r;
            const float qb = a.qbias[(size_t)g * a.ldq + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) qz[eh_crow(i, lane) * QP + col] = acc[i] + qb;
        }
    }
    __syncthreads();

    // ---- phase B: attention, 8 targets per wave, z overwrites qt
    {
        const int node0 = L.node_ptr[g];
        const int stride = L.adj_stride[g];
        const uint32_t* gadj = a.adj + L.adj_ptr[g];
        const int grp = lane >> 3, c = lane & 7;
        const int head = grp % HEADS, slot = grp / HEADS;
        for (int k = 0; k < 8; ++k) {
            const int rr = wave * 8 + k;
            if (rr >= rows) break;
            const int node = row0 + rr;
            float* qp = qz + rr * QP + head * 64 + c * 8;
            float q[8], z[8];
            {
                f32x4 q0 = *reinterpret_cast<const f32x4*>(qp), q1 = *reinterpret_cast<const f32x4*>(qp + 4);
#pragma unroll
                for (int d = 0; d < 4; ++d) { q[d] = q0[d]; q[4 + d] = q1[d]; }
            }
            eh_attend<HEADS>(gadj + (int64_t)(node - node0) * stride, stride, node0, node, a.hprev, q, lane, z);
            if (slot == 0) {
                f32x4 o0, o1;
#pragma unroll
                for (int d = 0; d < 4; ++d) { o0[d] = z[d]; o1[d] = z[4 + d]; }
                *reinterpret_cast<f32x4*>(qp) = o0;
                *reinterpret_cast<f32x4*>(qp + 4) = o1;
            }
        }
    }
    __syncthreads();

    // ---- phase C: conv = silu(Ocat [z | h] + obias0 + [deg>0] obias1); K chunks split over the waves
    {
        f32x16 acc0 = eh_zero16(), acc1 = eh_zero16();
        for (int ch = wave; ch < HEADS + 1; ch += 4) {
            eh_frag af, bf;
            if (ch < HEADS) {
                const float* p = qz + r * QP + ch * 64 + hh * 32;
#pragma unroll
                for (int i = 0; i < 8; ++i) af.v[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
            } else {
                eh_load_frag_lds(af, hp, r, hh);
            }
            eh_load_frag(bf, a.Ocat + (size_t)r * (H + 64) + ch * 64 + hh * 32);
            eh_mma64(acc0, af, bf);
            eh_load_frag(bf, a.Ocat + (size_t)(32 + r) * (H + 64) + ch * 64 + hh * 32);
            eh_mma64(acc1, af, bf);
        }
        __syncthreads();                      // every wave is done reading qz
        float* mine = pt + wave * EH_TSZ;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int rr = eh_crow(i, lane);
            mine[rr * EH_TP + r] = acc0[i];
            mine[rr * EH_TP + 32 + r] = acc1[i];
        }
        __syncthreads();
        const int row = tid >> 3, c0 = (tid & 7) * 8;
        const bool hasnbr = a.deg[row0 + (row < rows ? row : rows - 1)] > 0;
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(pt + row * EH_TP + c0 + 4 * q4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                f32x4 t = *reinterpret_cast<const f32x4*>(pt + w * EH_TSZ + row * EH_TP + c0 + 4 * q4);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int col = c0 + 4 * q4 + e;
                float x = v[e] + a.obias0[(size_t)g * 64 + col];
                if (hasnbr) x += a.obias1[(size_t)g * 64 + col];
                v[e] = eh_silu(x);
            }
            *reinterpret_cast<f32x4*>(mt + row * EH_TP + c0 + 4 * q4) = v;
        }
    }
    __syncthreads();

    // ---- phase D: GRU.  wave = (column tile ct, gate part): part 0 -> r and i_n, part 1 -> z and h_n
    {
        const int ct = wave & 1, part = wave >> 1;
        const int wr = ct * 32 + r, col = ct * 32 + r;
        eh_frag am, ah, bf;
        eh_load_frag_lds(am, mt, r, hh);
        eh_load_frag_lds(ah, hp, r, hh);
        f32x16 g0 = eh_zero16(), g1 = eh_zero16();
        float* xz = xg + (ct * 2 + 0) * 1024;
        float* xh = xg + (ct * 2 + 1) * 1024;
        if (part == 1) {
            eh_load_frag(bf, a.w_ih + (size_t)(64 + wr) * 64 + hh * 32);  eh_mma64(g0, am, bf);
            eh_load_frag(bf, a.w_hh + (size_t)(64 + wr) * 64 + hh * 32);  eh_mma64(g0, ah, bf);
            eh_load_frag(bf, a.w_hh + (size_t)(128 + wr) * 64 + hh * 32); eh_mma64(g1, ah, bf);
            const float biz = a.b_ih[64 + col] + a.b_hh[64 + col], bhn = a.b_hh[128 + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                xz[i * 64 + lane] = 1.0f / (1.0f + __expf(-(g0[i] + biz)));
                xh[i * 64 + lane] = g1[i] + bhn;
            }
        } else {
            eh_load_frag(bf, a.w_ih + (size_t)(wr) * 64 + hh * 32);       eh_mma64(g0, am, bf);
            eh_load_frag(bf, a.w_hh + (size_t)(wr) * 64 + hh * 32);       eh_mma64(g0, ah, bf);
            eh_load_frag(bf, a.w_ih + (size_t)(128 + wr) * 64 + hh * 32); eh_mma64(g1, am, bf);
        }
        __syncthreads();
        if (part == 0) {
            const float bir = a.b_ih[col] + a.b_hh[col], bin = a.b_ih[128 + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rr = eh_crow(i, lane);
                float zg = xz[i * 64 + lane], hn = xh[i * 64 + lane];
                float rg = 1.0f / (1.0f + __expf(-(g0[i] + bir)));
                float ng = tanhf((g1[i] + bin) + rg * hn);
                float hv = hp[rr * EH_TP + col];
                float o = (1.0f - zg) * ng + zg * hv;
                ht[rr * EH_TP + col] = o;
                if (rr < rows) a.hnew[(size_t)(row0 + rr) * 64 + col] = o;
            }
        }
    }
    __syncthreads();

    // ---- phase E: context pre-activation column sums
    {
        eh_frag af;
        eh_load_frag_lds(af, ht, r, hh);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = wave * 2 + j;
            eh_frag bf;
            eh_load_frag(bf, a.cw0 + (size_t)(nt * 32 + r) * 128 + hh * 32);
            f32x16 acc = eh_zero16();
            eh_mma64(acc, af, bf);
            const int col = nt * 32 + r;
            const float cb = a.cbias[(size_t)g * 256 + col];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = eh_silu(acc[i] + cb);
                sum += (eh_crow(i, lane) < rows) ? v : 0.f;
            }
            sum += __shfl_xor(sum, 32);
            if (lane < 32) a.part[(size_t)tile * 256 + col] = sum;
        }
    }
}

// --------------------------------------------------------------------------------------------
// tail input of the active nodes: [h + ctx | emb_t(x_t) | emb_0(x_0) | emb_sel[1]]  (layers.py:377)
// compact rows in active-node order; one wave per node tile (local index k of graph g).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_active(edgehip_layout L, edgehip_gnn_params P,
                                                       const int32_t* __restrict__ act_nodes,
                                                       const int32_t* __restrict__ kptr, const float* __restrict__ h,
                                                       const float* __restrict__ ctx, const int32_t* __restrict__ deg,
                                                       const int32_t* __restrict__ d0, float* __restrict__ X) {
    int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (tile >= L.num_tiles) return;
    int g = L.tile_graph[tile];
    int node0 = L.node_ptr[g];
    int k0 = L.tile_row0[tile] - node0;
    int K = kptr[g + 1] - kptr[g];
    if (k0 >= K) return;
    int kend = k0 + L.tile_rows[tile];
    kend = kend < K ? kend : K;
    float cg = ctx[(size_t)g * 64 + lane];
    float wt = P.embedding_t_w[lane], bt = P.embedding_t_b[lane];
    float w0 = P.embedding_0_w[lane], b0 = P.embedding_0_b[lane];
    float es = P.embedding_sel_w[64 + lane];
    float md = (float)P.max_degree;
    for (int k = k0; k < kend; ++k) {
        int i = act_nodes[node0 + k];
        int dt = deg[i];
        dt = dt > P.max_degree + 1 ? P.max_degree + 1 : dt;
        float xt = (float)dt / md, x0 = (float)d0[i] / md;
        float* o = X + (size_t)(kptr[g] + k) * 256;
        o[lane] = h[(size_t)i * 64 + lane] + cg;
        o[64 + lane] = wt * xt + bt;
        o[128 + lane] = w0 * x0 + b0;
        o[192 + lane] = es;
    }
}

// --------------------------------------------------------------------------------------------
// gnn forward driver
// --------------------------------------------------------------------------------------------
static int eh_gnn_forward_impl(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                               const uint32_t* adj, const int32_t* deg, const int32_t* d0, const int32_t* t_graph,
                               int t_shared, const uint8_t* flags, const int32_t* act_nodes, const int32_t* kptr,
                               const edgehip_counts* counts, float* node_repr, float* h_out, const eh_ws& w,
                               hipStream_t s) {
    eh_fold_map fm = eh_make_fold_map(P);
    const int B = L->num_graphs, Nn = L->num_nodes, NT = L->num_tiles;
    if (B == 0 || Nn == 0) return EDGEHIP_OK;
    const int tile_blocks = (NT + 3) / 4;
    int hmax = 1;
    for (int l = 0; l < P->num_layers; ++l) hmax = P->heads[l] > hmax ? P->heads[l] : hmax;
    const int ldq = hmax * 64;

    EH_PROF(EH_K_TIME_EMBED, s, hipLaunchKernelGGL(k_time_embed, dim3(t_shared ? 1 : B), dim3(256), 0, s, *P, t_graph, w.te));
    EH_PROF(EH_K_NODE_IN, s, hipLaunchKernelGGL(k_node_in, dim3(tile_blocks), dim3(256), 0, s, *L, P->max_degree, folded + fm.nin_at,
                       folded + fm.nin_a0, folded + fm.nin_as, deg, d0, flags, w.hA, w.part));
    float* hcur = w.hA;
    float* hnext = w.hB;
    for (int l = 0; l <= P->num_layers; ++l) {
        eh_layer_ptrs lp = {};
        if (l < P->num_layers) {
            const eh_fold_layer& f = fm.layer[l];
            lp.Mcat = folded + f.Mcat; lp.Mtcat = folded + f.Mtcat; lp.mbcat = folded + f.mbcat;
            lp.Vbbar = folded + f.Vbbar; lp.bvbar = folded + f.bvbar; lp.Vsum = folded + f.Vsum;
            lp.skip_w = P->skip_w[l]; lp.skip_b = P->skip_b[l]; lp.H = P->heads[l] * 64;
        }
        EH_PROF(EH_K_GRAPH_UPDATE, s, hipLaunchKernelGGL(k_graph_update, dim3(B), dim3(256), 0, s, *L, *P, l, lp, w.part, w.te, t_shared, w.ctx,
                           w.qbias, ldq, w.obias0, w.obias1, w.cbias));
        if (l == P->num_layers) break;
        const int h = P->heads[l];
        eh_layer_args la = {};
        la.adj = adj; la.hprev = hcur; la.deg = deg;
        la.Mcat = lp.Mcat; la.qbias = w.qbias; la.ldq = ldq;
        la.Ocat = folded + fm.layer[l].Ocat; la.obias0 = w.obias0; la.obias1 = w.obias1;
        la.w_ih = P->gru_w_ih; la.w_hh = P->gru_w_hh; la.b_ih = P->gru_b_ih; la.b_hh = P->gru_b_hh;
        la.cw0 = P->context_w0; la.cbias = w.cbias; la.hnew = hnext; la.part = w.part;
        switch (h) {
            case 8: EH_PROF(EH_K_LAYER, s, hipLaunchKernelGGL(k_layer<8>, dim3(NT), dim3(256), 0, s, *L, la)); break;
            case 4: EH_PROF(EH_K_LAYER, s, hipLaunchKernelGGL(k_layer<4>, dim3(NT), dim3(256), 0, s, *L, la)); break;
            case 2: EH_PROF(EH_K_LAYER, s, hipLaunchKernelGGL(k_layer<2>, dim3(NT), dim3(256), 0, s, *L, la)); break;
            default: EH_PROF(EH_K_LAYER, s, hipLaunchKernelGGL(k_layer<1>, dim3(NT), dim3(256), 0, s, *L, la)); break;
        }
        float* t = hcur; hcur = hnext; hnext = t;
    }
    if (h_out) {
        // debug: reference `nodes` after the last layer = h + ctx
        // (written by the gather kernel's formula for every node would need another kernel; copy h and
        //  let the caller add ctx)  -> keep simple: copy h
        (void)hipMemcpyAsync(h_out, hcur, (size_t)Nn * 64 * 4, hipMemcpyDeviceToDevice, s);
    }
    // tail on the active nodes
    EH_PROF(EH_K_GATHER, s, hipLaunchKernelGGL(k_gather_active, dim3(tile_blocks), dim3(256), 0, s, *L, *P, act_nodes, kptr, hcur, w.ctx, deg,
                       d0, w.Xact));
    eh_gemm_desc g0 = {};
    g0.A = w.Xact; g0.lda = 256; g0.K1 = 256; g0.W = P->node_out_w0; g0.K = 256; g0.N = 128; g0.bias = P->node_out_b0;
    g0.Y = w.U1; g0.ldy = 128; g0.act = 1; g0.M = Nn; g0.m_dev = &counts->k_total;
    g0.prof_id = EH_K_GEMM_TAIL;
    int rc = eh_launch_gemm(g0, s);
    if (rc != EDGEHIP_OK) return rc;
    eh_gemm_desc g1 = g0;
    g1.A = w.U1; g1.lda = 128; g1.K1 = 128; g1.W = P->node_out_w2; g1.K = 128; g1.bias = P->node_out_b2; g1.Y = w.U2;
    rc = eh_launch_gemm(g1, s);
    if (rc != EDGEHIP_OK) return rc;
    eh_gemm_desc g2 = g1;
    g2.A = w.U2; g2.W = folded + fm.W4f; g2.bias = folded + fm.b4f; g2.Y = node_repr; g2.act = 0;
    rc = eh_launch_gemm(g2, s);
    if (rc != EDGEHIP_OK) return rc;
    if (hipGetLastError() != hipSuccess) return EDGEHIP_ERR_LAUNCH;
    return EDGEHIP_OK;
}

extern "C" int edgehip_gnn_forward(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                                   const uint32_t* adj, const int32_t* deg, const int32_t* d0, const int32_t* t_graph,
                                   int32_t t_shared, const uint8_t* flags, const int32_t* act_nodes, const int32_t* kptr,
                                   const edgehip_counts* counts, float* node_repr, float* h_out, void* ws,
                                   size_t ws_bytes, void* stream) {
    int rc = eh_check_params(P);
    if (rc != EDGEHIP_OK) return rc;
    EH_CHECK_ARG(L && folded && adj && deg && d0 && t_graph && flags && act_nodes && kptr && counts && node_repr && ws);
    eh_ws w = eh_carve(L, P, ws);
    if (ws_bytes < w.bytes) return EDGEHIP_ERR_WORKSPACE;
    return eh_gnn_forward_impl(L, P, folded, adj, deg, d0, t_graph, t_shared, flags, act_nodes, kptr, counts, node_repr,
                               h_out, w, eh_stream(stream));
}

// --------------------------------------------------------------------------------------------
// active-pair tail.  One wave per pair tile (32 pairs of one graph); block = 4 tiles.
//   e   = silu(v_a + v_b + final_b0)                 [32 x 128]   (A operand built on the fly)
//   hid = silu(e W2^T + b2)                           [32 x 64]    MFMA
//   logit_c = hid . W4[c] + b4[c]                     via LDS, lane = (pair, class)
// MODE 0: log_softmax -> Gumbel decision -> overwrite the state bits.   MODE 1: CE against x_{t-1}.
// --------------------------------------------------------------------------------------------
struct eh_pair_args {
    const float* node_repr; const int32_t* act_nodes; const int32_t* kptr; const int64_t* aptr; const int32_t* tptr;
    edgehip_counts* counts;
    const float *fb0, *fw2, *fb2, *fw4, *fb4;
    const float* logits_in; int logits_are_logprobs;
    const float* uniforms; edgehip_rng rng; int use_rng; int step; const float* lut;
    uint32_t* adj;               // MODE 0: state (rw); MODE 1: x_{t-1} (ro)
    float* logits_out; uint8_t* bits_out; int64_t pair_cap;
    float* tile_partial; int64_t tile_cap;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_pair(edgehip_layout L, eh_pair_args a) {
    __shared__ float hid[4][32][65];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int total = a.counts->pair_tiles;
    const int B = L.num_graphs;
    for (int tg = blockIdx.x; tg * 4 < total; tg += gridDim.x) {
        const int tile = tg * 4 + wave;
        const bool valid = tile < total;
        int b = 0, K = 2, node0 = 0, kbase = 0;
        int64_t A = 1, k0 = 0, abase = 0;
        if (valid) {
            int lo = 0, hi = B - 1;
            while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (a.tptr[mid] <= tile) lo = mid; else hi = mid - 1; }
            b = lo;
            kbase = a.kptr[b];
            K = a.kptr[b + 1] - kbase;
            A = (int64_t)K * (K - 1) / 2;
            k0 = (int64_t)(tile - a.tptr[b]) * EDGEHIP_PAIR_TILE;
            node0 = L.node_ptr[b];
            abase = a.aptr[b];
        }
        const int r = lane & 31, hh = lane >> 5;
        int p = 0, q = 1;
        if (valid) {
            int64_t k = k0 + r;
            if (k >= A) k = A - 1;
            eh_pair_decode(k, K, p, q);
        }
        if (valid && !a.logits_in) {
            const float* va = a.node_repr + (size_t)(kbase + p) * 128;
            const float* vb = a.node_repr + (size_t)(kbase + q) * 128;
            f32x16 acc0 = eh_zero16(), acc1 = eh_zero16();
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int ko = ch * 64 + hh * 32;
                eh_frag fa, fb, fc, e;
                eh_load_frag(fa, va + ko); eh_load_frag(fb, vb + ko); eh_load_frag(fc, a.fb0 + ko);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) e.v[i][c] = eh_silu(fa.v[i][c] + fb.v[i][c] + fc.v[i][c]);
                eh_frag w;
                eh_load_frag(w, a.fw2 + (size_t)r * 128 + ko);        eh_mma64(acc0, e, w);
                eh_load_frag(w, a.fw2 + (size_t)(32 + r) * 128 + ko); eh_mma64(acc1, e, w);
            }
            const float b20 = a.fb2[r], b21 = a.fb2[32 + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rr = eh_crow(i, lane);
                hid[wave][rr][r] = eh_silu(acc0[i] + b20);
                hid[wave][rr][32 + r] = eh_silu(acc1[i] + b21);
            }
        }
        __syncthreads();
        // lane = (pair pr, class c)
        const int pr = lane >> 1, c = lane & 1;
        const int64_t k = k0 + pr;
        const bool live = valid && k < A;
        float logit = 0.f;
        if (live) {
            if (a.logits_in) {
                logit = a.logits_in[(abase + k) * 2 + c];
            } else {
                const float* w4 = a.fw4 + c * 64;
                float s = a.fb4[c];
#pragma unroll 8
                for (int d = 0; d < 64; ++d) s += hid[wave][pr][d] * w4[d];
                logit = s;
            }
        }
        __syncthreads();
        float other = __shfl_xor(logit, 1);
        float l0 = c == 0 ? logit : other, l1 = c == 0 ? other : logit;
        // the pair of lane pr (decoded by lane pr < 32): fetch (p,q) of pair `pr`
        int pp = __shfl(p, pr), qq = __shfl(q, pr);
        float ce = 0.f;
        if (live) {
            int64_t o = abase + k;
            if (a.logits_out) {
                if (o < a.pair_cap) a.logits_out[o * 2 + c] = logit;
                else atomicOr(&a.counts->status, EDGEHIP_STATUS_CAPACITY);
            }
            if (c == 0) {
                float lp0 = l0, lp1 = l1;
                if (!a.logits_are_logprobs) {    // F.log_softmax(dim=1): x - max - log(sum exp(x - max))
                    float mx = fmaxf(l0, l1);
                    float t0 = l0 - mx, t1 = l1 - mx;
                    float ls = logf(expf(t0) + expf(t1));
                    lp0 = t0 - ls; lp1 = t1 - ls;
                }
                int gi = a.act_nodes[node0 + pp] - node0, gj = a.act_nodes[node0 + qq] - node0;
                int stride = L.adj_stride[b];
                uint32_t* g = a.adj + L.adj_ptr[b];
                uint32_t* ri = g + (int64_t)gi * stride;
                uint32_t* rj = g + (int64_t)gj * stride;
                if (MODE == 0) {
                    float u0, u1;
                    if (a.use_rng) {
                        int64_t n = L.node_ptr[b + 1] - node0;
                        int64_t pidx = eh_pair_index(n, gi, gj);
                        eh_u4 rr = eh_philox(a.rng.seed, (uint32_t)pidx, (uint32_t)(pidx >> 32),
                                             (uint32_t)(L.gid_base + b), (uint32_t)a.step * 8u + EH_STREAM_PAIR);
                        u0 = eh_u24(rr.x); u1 = eh_u24(rr.y);
                    } else {
                        u0 = a.uniforms[o * 2]; u1 = a.uniforms[o * 2 + 1];
                    }
                    int bit = eh_decide(u0, u1, lp0, lp1, a.lut);
                    int old = eh_adj_get(ri, gj);
                    if (bit != old) {
                        if (bit) { eh_adj_set(ri, gj); eh_adj_set(rj, gi); }
                        else { eh_adj_clear(ri, gj); eh_adj_clear(rj, gi); }
                    }
                    if (a.bits_out && o < a.pair_cap) a.bits_out[o] = (uint8_t)bit;
                } else {
                    // -sum_c exp(log-one-hot target_c) * logp_c   (active.py:165; exp(log 1e-30) = 1e-30)
                    int tgt = eh_adj_get(ri, gj);
                    float w0 = tgt ? 1e-30f : 1.0f, w1 = tgt ? 1.0f : 1e-30f;
                    ce = -(w0 * lp0 + w1 * lp1);
                }
            }
        }
        if (MODE == 1) {
            // deterministic butterfly sum over the wave (odd lanes and dead lanes contribute 0)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) ce += __shfl_xor(ce, d);
            if (valid && lane == 0) {
                if (tile < a.tile_cap) a.tile_partial[tile] = ce;
                else atomicOr(&a.counts->status, EDGEHIP_STATUS_CAPACITY);
            }
        }
    }
}

static void eh_fill_pair_weights(eh_pair_args& a, const edgehip_gnn_params* P) {
    a.fb0 = P->final_b0; a.fw2 = P->final_w2; a.fb2 = P->final_b2; a.fw4 = P->final_w4; a.fb4 = P->final_b4;
}

extern "C" int edgehip_edge_resample(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                                     const float* node_repr, const int32_t* act_nodes, const int32_t* kptr,
                                     const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                                     const float* logits_in, int32_t logits_are_logprobs, const float* uniforms,
                                     const edgehip_rng* rng, int32_t step, const float* gumbel_lut, uint32_t* adj,
                                     float* logits_out, uint8_t* bits_out, int64_t pair_cap, void* stream) {
    (void)folded;
    EH_CHECK_ARG(L && act_nodes && kptr && aptr && tptr && counts && adj);
    EH_CHECK_ARG(uniforms || rng);
    EH_CHECK_ARG(logits_in || (node_repr && P));
    if (!logits_in) { int rc = eh_check_params(P); if (rc != EDGEHIP_OK) return rc; }
    if (L->num_graphs == 0) return EDGEHIP_OK;
    eh_pair_args a = {};
    a.node_repr = node_repr; a.act_nodes = act_nodes; a.kptr = kptr; a.aptr = aptr; a.tptr = tptr; a.counts = counts;
    if (P) eh_fill_pair_weights(a, P);
    a.logits_in = logits_in; a.logits_are_logprobs = logits_are_logprobs;
    a.uniforms = uniforms; a.rng = rng ? *rng : edgehip_rng{0}; a.use_rng = uniforms ? 0 : 1; a.step = step;
    a.lut = gumbel_lut; a.adj = adj; a.logits_out = logits_out; a.bits_out = bits_out; a.pair_cap = pair_cap;
    EH_PROF(EH_K_PAIR, eh_stream(stream), hipLaunchKernelGGL(k_pair<0>, dim3(2048), dim3(256), 0, eh_stream(stream), *L, a));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// kl[b] = sum of the graph's tile partials + 6.9078e-29 * (#inactive pairs) + 6.9078e-29 * #pairs
__global__ __launch_bounds__(64) void k_kl_reduce(edgehip_layout L, const int32_t* __restrict__ tptr,
                                                  const int64_t* __restrict__ aptr, const float* __restrict__ partial,
                                                  float* __restrict__ kl) {
    int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= L.num_graphs) return;
    float s = 0.f;
    for (int t = tptr[b]; t < tptr[b + 1]; ++t) s += partial[t];
    float n = (float)(L.node_ptr[b + 1] - L.node_ptr[b]);
    float npairs = floorf(n * (n - 1.0f) / 2.0f);          // active.py:175, float floor-division
    float nact = (float)(aptr[b + 1] - aptr[b]);
    s += 6.9078e-29f * (npairs - nact);
    float ent = 6.9078e-29f * (float)(L.pair_ptr[b + 1] - L.pair_ptr[b]);
    kl[b] = s + ent;
}

extern "C" int edgehip_edge_ce_loss(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                                    const float* node_repr, const int32_t* act_nodes, const int32_t* kptr,
                                    const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                                    const uint32_t* adj_tm1, float* tile_partial, int64_t tile_cap, float* kl,
                                    float* logits_out, int64_t pair_cap, void* stream) {
    (void)folded;
    int rc = eh_check_params(P);
    if (rc != EDGEHIP_OK) return rc;
    EH_CHECK_ARG(L && node_repr && act_nodes && kptr && aptr && tptr && counts && adj_tm1 && tile_partial && kl);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    eh_pair_args a = {};
    a.node_repr = node_repr; a.act_nodes = act_nodes; a.kptr = kptr; a.aptr = aptr; a.tptr = tptr; a.counts = counts;
    eh_fill_pair_weights(a, P);
    a.adj = const_cast<uint32_t*>(adj_tm1); a.logits_out = logits_out; a.pair_cap = pair_cap;
    a.tile_partial = tile_partial; a.tile_cap = tile_cap;
    EH_PROF(EH_K_PAIR, eh_stream(stream), hipLaunchKernelGGL(k_pair<1>, dim3(2048), dim3(256), 0, eh_stream(stream), *L, a));
    hipLaunchKernelGGL(k_kl_reduce, dim3((L->num_graphs + 63) / 64), dim3(64), 0, eh_stream(stream), *L, tptr, aptr,
                       tile_partial, kl);
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// --------------------------------------------------------------------------------------------
// one reverse step (p_sample) for the whole batch, no host sync
// --------------------------------------------------------------------------------------------
__global__ void k_fill_i32(int32_t* p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" int edgehip_sample_step(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                                   uint32_t* adj, const int32_t* d0, int32_t t, const float* post_table, int32_t nmax,
                                   const float* u_node, const float* u_pair, const edgehip_rng* rng,
                                   const float* gumbel_lut, edgehip_counts* counts, void* ws, size_t ws_bytes,
                                   void* stream) {
    int rc = eh_check_params(P);
    if (rc != EDGEHIP_OK) return rc;
    EH_CHECK_ARG(L && folded && adj && d0 && post_table && counts && ws && t >= 0 && t < P->num_steps);
    EH_CHECK_ARG((u_node && u_pair) || rng);
    eh_ws w = eh_carve(L, P, ws);
    if (ws_bytes < w.bytes) return EDGEHIP_ERR_WORKSPACE;
    if (L->num_graphs == 0 || L->num_nodes == 0) return EDGEHIP_OK;
    hipStream_t s = eh_stream(stream);
    hipLaunchKernelGGL(k_fill_i32, dim3((L->num_graphs + 255) / 256), dim3(256), 0, s, w.t_graph, L->num_graphs, t);
    rc = edgehip_degree_from_adj(L, adj, w.deg, stream);
    if (rc) return rc;
    rc = edgehip_active_select_sample(L, d0, w.deg, w.t_graph, post_table, nmax, u_node, rng, gumbel_lut, w.flags,
                                      w.act_nodes, w.act_count, stream);
    if (rc) return rc;
    rc = edgehip_active_offsets(L, w.act_count, w.kptr, w.aptr, w.tptr, counts, stream);
    if (rc) return rc;
    rc = eh_gnn_forward_impl(L, P, folded, adj, w.deg, d0, w.t_graph, 1, w.flags, w.act_nodes, w.kptr, counts,
                             w.node_repr, nullptr, w, s);
    if (rc) return rc;
    return edgehip_edge_resample(L, P, folded, w.node_repr, w.act_nodes, w.kptr, w.aptr, w.tptr, counts, nullptr, 0,
                                 u_pair, rng, t, gumbel_lut, adj, nullptr, nullptr, 0, stream);
}
