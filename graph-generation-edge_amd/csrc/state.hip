// state.hip - bit-matrix graph state: degrees, degree-guided active-node selection, stream
// compaction, forward noising, edge extraction.  Integer / bit work, HBM- or latency-bound.
// Compiled with -ffp-contract=off: every Gumbel decision is a plain fp32 add + compare.
#include "common.h"
#include "prof.h"

// ---------------------------------------------------------------------------------------------
// a8  degrees: deg[i] = popcount(row i).  8 lanes per row, strided words, shuffle reduce.
// ---------------------------------------------------------------------------------------------
template <bool UPPER>
__global__ __launch_bounds__(256) void k_row_popcount(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                      int32_t* __restrict__ out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int row = gid >> 3, sub = gid & 7;
    int cnt = 0;
    if (row < L.num_nodes) {
        int b = L.node_graph[row];
        int li = row - L.node_ptr[b];
        int stride = L.adj_stride[b];
        const uint32_t* p = adj + L.adj_ptr[b] + (int64_t)li * stride;
        for (int w = sub; w < stride; w += 8) {
            uint32_t v = p[w];
            if (UPPER) {   // keep only columns j > li
                int lo = w * 32;
                if (lo + 31 <= li) v = 0;
                else if (lo <= li) v &= ~((2u << (li - lo)) - 1u);   // clear bits 0..li-lo
            }
            cnt += __popc(v);
        }
    }
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    cnt += __shfl_xor(cnt, 4);
    if (row < L.num_nodes && sub == 0) out[row] = cnt;
}

extern "C" int edgehip_degree_from_adj(const edgehip_layout* L, const uint32_t* adj, int32_t* deg, void* stream) {
    EH_CHECK_ARG(L && adj && deg);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    int blocks = eh_cdiv((int64_t)L->num_nodes * 8, 256);
    EH_PROF(EH_K_DEGREE, eh_stream(stream), hipLaunchKernelGGL(k_row_popcount<false>, dim3(blocks), dim3(256), 0, eh_stream(stream), *L, adj, deg));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// a3  the device's Gumbel transforms on the grid (common.h): what edgehip/tables.py compares with the host's table
// ---------------------------------------------------------------------------------------------
EH_DEFINE_GUMBEL_GRID_KERNEL(k_gumbel_grid_state)

extern "C" int edgehip_gumbel_grid(float* exact_out, float* fast_out, int64_t k0, int64_t n, int32_t which, void* stream) {
    EH_CHECK_ARG((exact_out || fast_out) && k0 >= 0 && n >= 0 && k0 + n <= ((int64_t)1 << 24) && (which == 0 || which == 1));
    if (n == 0) return EDGEHIP_OK;
    if (which == 1) return eh_gumbel_grid_gnn(exact_out, fast_out, k0, n, eh_stream(stream));
    hipLaunchKernelGGL(k_gumbel_grid_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, eh_stream(stream), exact_out, fast_out,
                       (long long)k0, (long long)n);
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// a9/a10/a11  active-node selection + stable per-graph compaction.  One 256-thread block per graph.
// ---------------------------------------------------------------------------------------------
template <int MODE>   // 0: sample (posterior table + Gumbel), 1: train (deg_tm1 > deg_t)
__global__ __launch_bounds__(256) void k_select(edgehip_layout L, const int32_t* __restrict__ d0,
                                                const int32_t* __restrict__ deg, const int32_t* __restrict__ t_graph,
                                                const float* __restrict__ post, int nmax,
                                                const float* __restrict__ uniforms, edgehip_rng rng, int use_rng,
                                                const float* __restrict__ lut, uint8_t* __restrict__ flags,
                                                int32_t* __restrict__ act_nodes, int32_t* __restrict__ act_count,
                                                eh_select_extra x) {
    __shared__ int wave_cnt[4];
    __shared__ int running;
    int b = blockIdx.x;
    int node0 = L.node_ptr[b], n = L.node_ptr[b + 1] - node0;
    int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int t = (MODE == 0) ? (x.t_out ? x.t : t_graph[b]) : 0;
    // the fused reverse step: its two small fills ride along (one t for every graph; counters of the launches that
    // follow start at zero) instead of being launches of their own
    if (x.t_out && tid == 0) x.t_out[b] = x.t;
    if (x.clear) for (int i = b * 256 + tid; i < x.clear_n; i += gridDim.x * 256) x.clear[i] = 0;
    if (tid == 0) running = 0;
    // Graphs of up to 2 048 nodes: the flags of all 256-node chunks first, the wave totals of every chunk to LDS, ONE
    // barrier, then every thread adds up what lies before its own wave (the chunk-by-chunk form below takes three
    // barriers per chunk, and this launch is nothing but latency).
    constexpr int SEL_CH = 8;
    __shared__ int wcs[SEL_CH][4];
    if (n <= 256 * SEL_CH) {
        unsigned fbits = 0;
#pragma unroll
        for (int c = 0; c < SEL_CH; ++c) {
            if (256 * c >= n) break;
            const int li = 256 * c + tid;
            int flag = 0;
            if (li < n) {
                int i = node0 + li;
                if (MODE == 0) {
                    int nt = d0[i] - deg[i];
                    nt = nt < 0 ? 0 : (nt > nmax ? nmax : nt);
                    const float* lp = post + ((int64_t)t * (nmax + 1) + nt) * 2;
                    float u0, u1;
                    if (use_rng) {
                        eh_u4 r = eh_philox(rng.seed, (uint32_t)li, (uint32_t)(L.gid_base + b), 0u,
                                            (uint32_t)t * 8u + EH_STREAM_NODE);
                        u0 = eh_u24(r.x); u1 = eh_u24(r.y);
                    } else {
                        u0 = uniforms[2 * (int64_t)i]; u1 = uniforms[2 * (int64_t)i + 1];
                    }
                    flag = eh_decide(u0, u1, lp[0], lp[1], lut);
                } else {
                    flag = d0[i] > deg[i];   // here d0 = deg_{t-1}, deg = deg_t
                }
                flags[i] = (uint8_t)flag;
            }
            const unsigned long long m = __ballot(flag);
            if (lane == 0) wcs[c][wv] = __popcll(m);
            fbits |= (flag ? 1u : 0u) << c;
        }
        __syncthreads();
        int run = 0;
#pragma unroll
        for (int c = 0; c < SEL_CH; ++c) {
            if (256 * c >= n) break;
            const int w0 = wcs[c][0], w1 = wcs[c][1], w2 = wcs[c][2], w3 = wcs[c][3];
            const bool flag = (fbits >> c) & 1u;
            const unsigned long long m = __ballot(flag);
            const int off = run + (wv > 0 ? w0 : 0) + (wv > 1 ? w1 : 0) + (wv > 2 ? w2 : 0) + __popcll(m & ((1ull << lane) - 1ull));
            if (flag) act_nodes[node0 + off] = node0 + 256 * c + tid;
            run += w0 + w1 + w2 + w3;
        }
        if (tid == 0) act_count[b] = run;
        return;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        int li = base + tid;
        int flag = 0;
        if (li < n) {
            int i = node0 + li;
            if (MODE == 0) {
                int nt = d0[i] - deg[i];
                nt = nt < 0 ? 0 : (nt > nmax ? nmax : nt);
                const float* lp = post + ((int64_t)t * (nmax + 1) + nt) * 2;
                float u0, u1;
                if (use_rng) {
                    eh_u4 r = eh_philox(rng.seed, (uint32_t)li, (uint32_t)(L.gid_base + b), 0u,
                                        (uint32_t)t * 8u + EH_STREAM_NODE);
                    u0 = eh_u24(r.x); u1 = eh_u24(r.y);
                } else {
                    u0 = uniforms[2 * (int64_t)i]; u1 = uniforms[2 * (int64_t)i + 1];
                }
                flag = eh_decide(u0, u1, lp[0], lp[1], lut);
            } else {
                flag = d0[i] > deg[i];   // here d0 = deg_{t-1}, deg = deg_t
            }
            flags[i] = (uint8_t)flag;
        }
        unsigned long long m = __ballot(flag);
        int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wv; ++w) off += wave_cnt[w];
        if (flag) act_nodes[node0 + off + before] = node0 + li;
        __syncthreads();
        if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) act_count[b] = running;
}

extern "C" int edgehip_active_select_sample(const edgehip_layout* L, const int32_t* d0, const int32_t* deg,
                                            const int32_t* t_graph, const float* post_table, int32_t nmax,
                                            const float* uniforms, const edgehip_rng* rng, const float* gumbel_lut,
                                            uint8_t* flags, int32_t* act_nodes, int32_t* act_count, void* stream) {
    EH_CHECK_ARG(L && d0 && deg && t_graph && post_table && flags && act_nodes && act_count && nmax >= 0);
    EH_CHECK_ARG(uniforms || rng);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    edgehip_rng r = rng ? *rng : edgehip_rng{0};
    EH_PROF(EH_K_SELECT, eh_stream(stream), hipLaunchKernelGGL(k_select<0>, dim3(L->num_graphs), dim3(256), 0, eh_stream(stream), *L, d0, deg, t_graph,
                       post_table, nmax, uniforms, r, uniforms ? 0 : 1, gumbel_lut, flags, act_nodes, act_count, eh_select_extra{}));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// the same selection inside edgehip_sample_step: one t for all graphs (written to t_out for the kernels that follow)
// and `clear_n` ints at `clear` set to zero by the same launch
int eh_active_select_step(const edgehip_layout* L, const int32_t* d0, const int32_t* deg, int32_t t, int32_t* t_out,
                          const float* post_table, int32_t nmax, const float* uniforms, const edgehip_rng* rng,
                          const float* gumbel_lut, uint8_t* flags, int32_t* act_nodes, int32_t* act_count,
                          int32_t* clear, int clear_n, hipStream_t s) {
    EH_CHECK_ARG(L && d0 && deg && t_out && post_table && flags && act_nodes && act_count && nmax >= 0);
    EH_CHECK_ARG(uniforms || rng);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    edgehip_rng r = rng ? *rng : edgehip_rng{0};
    eh_select_extra x = {t_out, t, clear, clear_n};
    EH_PROF(EH_K_SELECT, s, hipLaunchKernelGGL(k_select<0>, dim3(L->num_graphs), dim3(256), 0, s, *L, d0, deg, (const int32_t*)nullptr,
                       post_table, nmax, uniforms, r, uniforms ? 0 : 1, gumbel_lut, flags, act_nodes, act_count, x));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" int edgehip_active_select_train(const edgehip_layout* L, const int32_t* deg_tm1, const int32_t* deg_t,
                                           uint8_t* flags, int32_t* act_nodes, int32_t* act_count, void* stream) {
    EH_CHECK_ARG(L && deg_tm1 && deg_t && flags && act_nodes && act_count);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    hipLaunchKernelGGL(k_select<1>, dim3(L->num_graphs), dim3(256), 0, eh_stream(stream), *L, deg_tm1, deg_t,
                       (const int32_t*)nullptr, (const float*)nullptr, 0, (const float*)nullptr, edgehip_rng{0}, 0,
                       (const float*)nullptr, flags, act_nodes, act_count, eh_select_extra{});
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// offsets: prefix sums over the B graphs (single block; B is small next to Nn).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_offsets(int B, const int32_t* __restrict__ act_count, int32_t* __restrict__ kptr,
                                                 int64_t* __restrict__ aptr, int32_t* __restrict__ tptr,
                                                 edgehip_counts* __restrict__ counts) {
    __shared__ long long sk[256], sa[256], st[256];
    int tid = threadIdx.x;
    // each thread owns a contiguous chunk of graphs
    int per = (B + 255) / 256;
    int g0 = tid * per, g1 = min(B, g0 + per);
    long long ck = 0, ca = 0, ct = 0;
    for (int g = g0; g < g1; ++g) {
        long long K = act_count[g];
        long long A = K * (K - 1) / 2;
        ck += K; ca += A; ct += (A + EDGEHIP_PAIR_TILE - 1) / EDGEHIP_PAIR_TILE;
    }
    sk[tid] = ck; sa[tid] = ca; st[tid] = ct;
    __syncthreads();
    // exclusive scan of the 256 chunk sums (Hillis-Steele)
    for (int d = 1; d < 256; d <<= 1) {
        long long vk = 0, va = 0, vt = 0;
        if (tid >= d) { vk = sk[tid - d]; va = sa[tid - d]; vt = st[tid - d]; }
        __syncthreads();
        sk[tid] += vk; sa[tid] += va; st[tid] += vt;
        __syncthreads();
    }
    long long ok = sk[tid] - ck, oa = sa[tid] - ca, ot = st[tid] - ct;
    for (int g = g0; g < g1; ++g) {
        kptr[g] = (int32_t)ok; aptr[g] = oa; tptr[g] = (int32_t)ot;
        long long K = act_count[g];
        long long A = K * (K - 1) / 2;
        ok += K; oa += A; ot += (A + EDGEHIP_PAIR_TILE - 1) / EDGEHIP_PAIR_TILE;
    }
    if (tid == 255) {
        kptr[B] = (int32_t)sk[255]; aptr[B] = sa[255]; tptr[B] = (int32_t)st[255];
        counts->k_total = (int32_t)sk[255];
        counts->a_total = sa[255];
        counts->pair_tiles = (int32_t)st[255];
        counts->status = 0;
        counts->reserved = 0;
    }
}

// Upper bound of the active-pair tiles of a layout (every node of every graph active).  tptr / counts->pair_tiles and
// the tile loops of the pair kernels are 32-bit: a layout whose bound does not fit is refused by every entry point
// that forms the offsets, instead of wrapping on the device.  20 000-node graphs: 6.25 Mi tiles each, 85 per batch.
extern "C" int64_t edgehip_layout_pair_tile_bound(const edgehip_layout* L) {
    if (!L || L->num_graphs < 0 || L->max_nodes < 0) return -1;
    const int64_t n = L->max_nodes;
    return (int64_t)L->num_graphs * ((n * (n - 1) / 2 + EDGEHIP_PAIR_TILE - 1) / EDGEHIP_PAIR_TILE + 1);
}

extern "C" int edgehip_active_offsets(const edgehip_layout* L, const int32_t* act_count, int32_t* kptr, int64_t* aptr,
                                      int32_t* tptr, edgehip_counts* counts, void* stream) {
    EH_CHECK_ARG(L && act_count && kptr && aptr && tptr && counts);
    if (edgehip_layout_pair_tile_bound(L) > EDGEHIP_MAX_PAIR_TILES) return EDGEHIP_ERR_TOO_LARGE;
    EH_PROF(EH_K_OFFSETS, eh_stream(stream), hipLaunchKernelGGL(k_offsets, dim3(1), dim3(256), 0, eh_stream(stream), L->num_graphs, act_count, kptr, aptr, tptr,
                       counts));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// a10  materialise the reference's index lists (API parity / tests; the hot path never does)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_enum_nodes(edgehip_layout L, const int32_t* __restrict__ act_nodes,
                                                    const int32_t* __restrict__ kptr, edgehip_counts* counts,
                                                    int64_t* __restrict__ out, int64_t cap) {
    int b = blockIdx.x;
    int K = kptr[b + 1] - kptr[b];
    int node0 = L.node_ptr[b];
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int64_t o = (int64_t)kptr[b] + k;
        if (o < cap) out[o] = act_nodes[node0 + k];
        else atomicOr(&counts->status, EDGEHIP_STATUS_CAPACITY);
    }
}

__global__ __launch_bounds__(256) void k_enum_pairs(edgehip_layout L, const int32_t* __restrict__ act_nodes,
                                                    const int32_t* __restrict__ kptr, const int64_t* __restrict__ aptr,
                                                    const int32_t* __restrict__ tptr, edgehip_counts* counts,
                                                    int64_t* __restrict__ idx, int64_t* __restrict__ row,
                                                    int64_t* __restrict__ col, int64_t cap) {
    int total_tiles = counts->pair_tiles;
    int B = L.num_graphs;
    for (int tile = blockIdx.x * 8 + (threadIdx.x >> 5); tile < total_tiles; tile += gridDim.x * 8) {
        // graph of this tile: last b with tptr[b] <= tile
        int lo = 0, hi = B - 1;
        while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (tptr[mid] <= tile) lo = mid; else hi = mid - 1; }
        int b = lo;
        int64_t K = kptr[b + 1] - kptr[b];
        int64_t A = K * (K - 1) / 2;
        int64_t k = (int64_t)(tile - tptr[b]) * EDGEHIP_PAIR_TILE + (threadIdx.x & 31);
        if (k < A) {
            int p, q;
            eh_pair_decode(k, K, p, q);
            int node0 = L.node_ptr[b];
            int gi = act_nodes[node0 + p], gj = act_nodes[node0 + q];
            int64_t n = L.node_ptr[b + 1] - node0;
            int64_t o = aptr[b] + k;
            if (o < cap) {
                if (idx) idx[o] = L.pair_ptr[b] + eh_pair_index(n, gi - node0, gj - node0);
                if (row) row[o] = gi;
                if (col) col[o] = gj;
            } else {
                atomicOr(&counts->status, EDGEHIP_STATUS_CAPACITY);
            }
        }
    }
}

extern "C" int edgehip_active_pairs_enumerate(const edgehip_layout* L, const int32_t* act_nodes, const int32_t* kptr,
                                              const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                                              int64_t* active_node_indices, int64_t node_cap,
                                              int64_t* active_edge_indices, int64_t* row, int64_t* col, int64_t pair_cap,
                                              void* stream) {
    EH_CHECK_ARG(L && act_nodes && kptr && aptr && tptr && counts && node_cap >= 0 && pair_cap >= 0);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    if (active_node_indices) {
        hipLaunchKernelGGL(k_enum_nodes, dim3(L->num_graphs), dim3(256), 0, eh_stream(stream), *L, act_nodes, kptr,
                           counts, active_node_indices, node_cap);
        EH_LAUNCH_CHECK();
    }
    if (active_edge_indices || row || col) {
        hipLaunchKernelGGL(k_enum_pairs, dim3(1024), dim3(256), 0, eh_stream(stream), *L, act_nodes, kptr, aptr, tptr,
                           counts, active_edge_indices, row, col, pair_cap);
        EH_LAUNCH_CHECK();
    }
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// a4-a7  forward noising of the E0 true edges.  8 lanes per row of x0; only bits j > i are visited,
// both directions are written with atomicOr (several rows share words of the transposed bit).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_q_sample(edgehip_layout L, const uint32_t* __restrict__ adj0,
                                                  const int32_t* __restrict__ t_graph, const float* __restrict__ q_cum,
                                                  const float* __restrict__ q_one, const float* __restrict__ u_tm1,
                                                  const float* __restrict__ u_t, edgehip_rng rng, int use_rng,
                                                  const float* __restrict__ lut, uint32_t* __restrict__ adj_tm1,
                                                  uint32_t* __restrict__ adj_t) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int row = gid >> 3, sub = gid & 7;
    if (row >= L.num_nodes) return;
    int b = L.node_graph[row];
    int node0 = L.node_ptr[b];
    int li = row - node0;
    int64_t n = L.node_ptr[b + 1] - node0;
    int stride = L.adj_stride[b];
    int64_t base = L.adj_ptr[b];
    const uint32_t* p = adj0 + base + (int64_t)li * stride;
    int t = t_graph[b];
    int tm1 = t > 0 ? t - 1 : 0;
    float c0 = q_cum[2 * tm1], c1 = q_cum[2 * tm1 + 1];
    float o0 = q_one[2 * t], o1 = q_one[2 * t + 1];
    for (int w = sub; w < stride; w += 8) {
        uint32_t v = p[w];
        int lo = w * 32;
        if (lo + 31 <= li) continue;
        if (lo <= li) v &= ~((2u << (li - lo)) - 1u);
        while (v) {
            int j = lo + __ffs(v) - 1;
            v &= v - 1;
            int64_t pidx = eh_pair_index(n, li, j);
            float a0, a1, b0, b1;
            if (use_rng) {
                eh_u4 r = eh_philox(rng.seed, (uint32_t)pidx, (uint32_t)(pidx >> 32), (uint32_t)(L.gid_base + b),
                                    (uint32_t)t * 8u + EH_STREAM_QTM1);
                a0 = eh_u24(r.x); a1 = eh_u24(r.y); b0 = eh_u24(r.z); b1 = eh_u24(r.w);
            } else {
                int64_t g = (L.pair_ptr[b] + pidx) * 2;
                a0 = u_tm1[g]; a1 = u_tm1[g + 1]; b0 = u_t[g]; b1 = u_t[g + 1];
            }
            int keep1 = (t == 0) ? 1 : eh_decide(a0, a1, c0, c1, lut);   // vanilla.py:206-207: x_{t-1} := x0 at t = 0
            if (!keep1) continue;
            uint32_t* r1i = adj_tm1 + base + (int64_t)li * stride;
            uint32_t* r1j = adj_tm1 + base + (int64_t)j * stride;
            eh_adj_set(r1i, j); eh_adj_set(r1j, li);
            if (eh_decide(b0, b1, o0, o1, lut)) {
                uint32_t* r2i = adj_t + base + (int64_t)li * stride;
                uint32_t* r2j = adj_t + base + (int64_t)j * stride;
                eh_adj_set(r2i, j); eh_adj_set(r2j, li);
            }
        }
    }
}

extern "C" int edgehip_q_sample_edges(const edgehip_layout* L, const uint32_t* adj0, const int32_t* t_graph,
                                      const float* q_cum, const float* q_one, const float* u_tm1, const float* u_t,
                                      const edgehip_rng* rng, const float* gumbel_lut, uint32_t* adj_tm1,
                                      uint32_t* adj_t, void* stream) {
    EH_CHECK_ARG(L && adj0 && t_graph && q_cum && q_one && adj_tm1 && adj_t);
    EH_CHECK_ARG((u_tm1 && u_t) || rng);
    if (L->num_nodes == 0) return EDGEHIP_OK;
    edgehip_rng r = rng ? *rng : edgehip_rng{0};
    int blocks = eh_cdiv((int64_t)L->num_nodes * 8, 256);
    EH_PROF(EH_K_QSAMPLE, eh_stream(stream), hipLaunchKernelGGL(k_q_sample, dim3(blocks), dim3(256), 0, eh_stream(stream), *L, adj0, t_graph, q_cum, q_one,
                       u_tm1, u_t, r, (u_tm1 && u_t) ? 0 : 1, gumbel_lut, adj_tm1, adj_t));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// a17 epilogue  edge list in ascending pair order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_edge_slice(edgehip_layout L, const int32_t* __restrict__ row_upper,
                                                     int64_t* __restrict__ edge_slice) {
    __shared__ long long red[16];
    __shared__ long long total;
    int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { total = 0; edge_slice[0] = 0; }
    __syncthreads();
    for (int b = 0; b < L.num_graphs; ++b) {
        int n0 = L.node_ptr[b], n1 = L.node_ptr[b + 1];
        long long s = 0;
        for (int i = n0 + tid; i < n1; i += 1024) s += row_upper[i];
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) red[wv] = s;
        __syncthreads();
        if (tid == 0) {
            long long a = 0;
            for (int w = 0; w < 16; ++w) a += red[w];
            total += a;
            edge_slice[b + 1] = total;
        }
        __syncthreads();
    }
}

extern "C" int edgehip_count_edges(const edgehip_layout* L, const uint32_t* adj, int32_t* row_upper, int64_t* edge_slice,
                                   void* stream) {
    EH_CHECK_ARG(L && adj && row_upper && edge_slice);
    if (L->num_nodes > 0) {
        int blocks = eh_cdiv((int64_t)L->num_nodes * 8, 256);
        hipLaunchKernelGGL(k_row_popcount<true>, dim3(blocks), dim3(256), 0, eh_stream(stream), *L, adj, row_upper);
        EH_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_edge_slice, dim3(1), dim3(1024), 0, eh_stream(stream), *L, row_upper, edge_slice);
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

__global__ __launch_bounds__(256) void k_extract(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                 const int32_t* __restrict__ row_upper,
                                                 const int64_t* __restrict__ edge_slice, int64_t* __restrict__ edge_index,
                                                 int64_t cap) {
    __shared__ int scan[256];
    __shared__ long long carry;
    int b = blockIdx.x, tid = threadIdx.x;
    int node0 = L.node_ptr[b], n = L.node_ptr[b + 1] - node0;
    int stride = L.adj_stride[b];
    const uint32_t* g = adj + L.adj_ptr[b];
    if (tid == 0) carry = edge_slice[b];
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        int li = base + tid;
        int c = li < n ? row_upper[node0 + li] : 0;
        scan[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            int v = tid >= d ? scan[tid - d] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        long long off = carry + scan[tid] - c;
        if (li < n) {
            const uint32_t* p = g + (int64_t)li * stride;
            for (int w = li >> 5; w < stride; ++w) {
                uint32_t v = p[w];
                int lo = w * 32;
                if (lo <= li) v &= ~((2u << (li - lo)) - 1u);
                while (v) {
                    int j = lo + __ffs(v) - 1;
                    v &= v - 1;
                    if (off < cap) { edge_index[off] = node0 + li; edge_index[cap + off] = node0 + j; }
                    ++off;
                }
            }
        }
        __syncthreads();
        if (tid == 255) carry += scan[255];
        __syncthreads();
    }
}

extern "C" int edgehip_extract_edges(const edgehip_layout* L, const uint32_t* adj, const int32_t* row_upper,
                                     const int64_t* edge_slice, int64_t* edge_index, int64_t edge_cap, void* stream) {
    EH_CHECK_ARG(L && adj && row_upper && edge_slice && edge_cap >= 0);
    if (L->num_graphs == 0 || edge_cap == 0) return EDGEHIP_OK;
    EH_CHECK_ARG(edge_index);
    hipLaunchKernelGGL(k_extract, dim3(L->num_graphs), dim3(256), 0, eh_stream(stream), *L, adj, row_upper, edge_slice,
                       edge_index, edge_cap);
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// x0 bit matrix from an edge list (global int32 ids, any orientation)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adj_from_edges(edgehip_layout L, const int32_t* __restrict__ edges, int64_t E,
                                                        uint32_t* __restrict__ adj) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int i = edges[2 * e], j = edges[2 * e + 1];
    if (i < 0 || j < 0 || i >= L.num_nodes || j >= L.num_nodes || i == j) return;
    int b = L.node_graph[i];
    if (L.node_graph[j] != b) return;
    int node0 = L.node_ptr[b];
    int stride = L.adj_stride[b];
    uint32_t* g = adj + L.adj_ptr[b];
    eh_adj_set(g + (int64_t)(i - node0) * stride, j - node0);
    eh_adj_set(g + (int64_t)(j - node0) * stride, i - node0);
}

extern "C" int edgehip_adj_from_edges(const edgehip_layout* L, const int32_t* edges, int64_t num_edges, uint32_t* adj,
                                      void* stream) {
    EH_CHECK_ARG(L && adj && num_edges >= 0);
    if (num_edges == 0) return EDGEHIP_OK;
    EH_CHECK_ARG(edges);
    hipLaunchKernelGGL(k_adj_from_edges, dim3(eh_cdiv(num_edges, 256)), dim3(256), 0, eh_stream(stream), *L, edges,
                       num_edges, adj);
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

extern "C" const char* edgehip_version(void) { return "edgehip 0.1 (gfx950)"; }
