// stats.hip - graph statistics of the sampled graphs straight from the bit matrix (the step
// after the path, datasets/evaluator.py:27-37 -> eval_utils/graph_statistics.py).  Every statistic the
// reference derives from degrees, triangles and degree products is an exact integer sum here:
//   edges, wedges sum d(d-1)/2, claws sum d(d-1)(d-2)/6, 3 x triangles = sum over edges {i,j} of |N(i) & N(j)|
//   (AND + popcount of two bit rows), and the edge sums behind the degree assortativity.
// The host turns them into the reference's floats in fp64 (edgehip/stats.py).
#include "common.h"
#include "prof.h"

// one wave per row i: the row's words stay in registers (lane w holds words w, w+64, ..), the set bits j > i
// are walked in ascending order; per neighbour every lane ANDs its words of row j and adds the popcount to a
// private counter (one wave reduction at the end, one 64-bit atomic per statistic and wave).
#define ST_MAXW 16          // words per lane: rows up to 64*16*32 = 32768 nodes

__global__ __launch_bounds__(256) void k_graph_stats(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                     const int32_t* __restrict__ deg, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= L.num_nodes) return;
    const int g = L.node_graph[i];
    const int node0 = L.node_ptr[g];
    const int stride = L.adj_stride[g];
    const int words = (L.node_ptr[g + 1] - node0 + 31) >> 5;
    const uint32_t* base = adj + L.adj_ptr[g];
    const uint32_t* rowi = base + (int64_t)(i - node0) * stride;
    uint32_t wi[ST_MAXW];
#pragma unroll
    for (int k = 0; k < ST_MAXW; ++k) wi[k] = (lane + 64 * k < words) ? rowi[lane + 64 * k] : 0u;
    const long long di = deg[i];
    unsigned long long common = 0;                      // per lane
    long long s_dd = 0, s_sum = 0, s_sq = 0, ecount = 0;   // wave-uniform (computed redundantly by every lane)
    const int li = i - node0;
    for (int w = li >> 5; w < words; ++w) {
        uint32_t bits = rowi[w];                        // uniform address: broadcast load
        if (w == (li >> 5)) bits &= ~((2u << (li & 31)) - 1u);      // neighbours j > i only
        while (bits) {
            const int b = __builtin_ctz(bits);
            bits &= bits - 1;
            const int lj = w * 32 + b;
            const uint32_t* rowj = base + (int64_t)lj * stride;
#pragma unroll
            for (int k = 0; k < ST_MAXW; ++k)
                if (64 * k < words) common += (lane + 64 * k < words) ? __popc(wi[k] & rowj[lane + 64 * k]) : 0;
            const long long dj = deg[node0 + lj];
            s_dd += di * dj; s_sum += di + dj; s_sq += di * di + dj * dj; ++ecount;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) common += __shfl_xor(common, o, 64);
    if (lane == 0) {
        unsigned long long* o = out + (size_t)g * EDGEHIP_NSTAT;
        const unsigned long long d = (unsigned long long)di;
        atomicAdd(o + 0, (unsigned long long)ecount);
        atomicAdd(o + 1, d * (d - (d > 0)) / 2);
        atomicAdd(o + 2, d >= 3 ? d * (d - 1) * (d - 2) / 6 : 0ull);
        atomicAdd(o + 3, common);
        atomicAdd(o + 4, (unsigned long long)s_dd);
        atomicAdd(o + 5, (unsigned long long)s_sum);
        atomicAdd(o + 6, (unsigned long long)s_sq);
        atomicMax(o + 7, d);
        atomicMin(o + 8, d);
        atomicAdd(o + 9, d);
    }
}

__global__ void k_stats_init(int B, unsigned long long* out) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * EDGEHIP_NSTAT) return;
    out[idx] = (idx % EDGEHIP_NSTAT == 8) ? ~0ull : 0ull;          // d_min starts at +inf
}

extern "C" int edgehip_graph_stats(const edgehip_layout* L, const uint32_t* adj, const int32_t* deg, int64_t* out,
                                   void* stream) {
    EH_CHECK_ARG(L && adj && deg && out);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    if (L->max_nodes > 64 * ST_MAXW * 32) return EDGEHIP_ERR_UNSUPPORTED;
    hipStream_t s = eh_stream(stream);
    unsigned long long* o = reinterpret_cast<unsigned long long*>(out);
    hipLaunchKernelGGL(k_stats_init, dim3((L->num_graphs * EDGEHIP_NSTAT + 255) / 256), dim3(256), 0, s, L->num_graphs, o);
    if (L->num_nodes > 0)
        EH_PROF(EH_K_MISC, s, hipLaunchKernelGGL(k_graph_stats, dim3((L->num_nodes + 3) / 4), dim3(256), 0, s, *L, adj, deg, o));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// shared edges of two states of the same batch layout: popcount(A & B) / 2 per graph   (edge_overlap,
// eval_utils/graph_statistics.py:287-299)
__global__ __launch_bounds__(256) void k_edge_overlap(edgehip_layout L, const uint32_t* __restrict__ a,
                                                      const uint32_t* __restrict__ b, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= L.num_nodes) return;
    const int g = L.node_graph[i];
    const int node0 = L.node_ptr[g];
    const int stride = L.adj_stride[g];
    const int words = (L.node_ptr[g + 1] - node0 + 31) >> 5;
    const int64_t off = L.adj_ptr[g] + (int64_t)(i - node0) * stride;
    unsigned long long c = 0;
    for (int w = lane; w < words; w += 64) c += __popc(a[off + w] & b[off + w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0 && c) atomicAdd(out + g, c);
}

extern "C" int edgehip_edge_overlap(const edgehip_layout* L, const uint32_t* adj_a, const uint32_t* adj_b, int64_t* out,
                                    void* stream) {
    EH_CHECK_ARG(L && adj_a && adj_b && out);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    hipStream_t s = eh_stream(stream);
    (void)hipMemsetAsync(out, 0, sizeof(int64_t) * L->num_graphs, s);
    if (L->num_nodes > 0)
        EH_PROF(EH_K_MISC, s, hipLaunchKernelGGL(k_edge_overlap, dim3((L->num_nodes + 3) / 4), dim3(256), 0, s, *L, adj_a, adj_b,
                           reinterpret_cast<unsigned long long*>(out)));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The statistics that need more than degrees and per-edge intersections (eval_utils/graph_statistics.py:60-73 LCC,
// :126-143 square_count, :233-246 cpl), still as exact integer sums from the bit matrix:
//   ext[g][0] = sum over node pairs i < j of c_ij (c_ij - 1), c_ij = |N(i) & N(j)|   (square_count = that / 4)
//   ext[g][1] = sum of shortest-path lengths over ordered pairs (i, j), j != i, j reachable from i
//   ext[g][2] = number of such pairs                                               (cpl = ext1 / ext2)
//   ext[g][3] = size of the largest connected component
// ---------------------------------------------------------------------------------------------
#define EDGEHIP_NSTAT_EXT 4

// common neighbours of every pair: wave per row i, lane per partner j > i (row i's words are wave-uniform loads)
__global__ __launch_bounds__(256) void k_stats_squares(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                       unsigned long long* __restrict__ ext) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= L.num_nodes) return;
    const int g = L.node_graph[i];
    const int node0 = L.node_ptr[g], n = L.node_ptr[g + 1] - node0;
    const int stride = L.adj_stride[g];
    const int words = (n + 31) >> 5;
    const uint32_t* base = adj + L.adj_ptr[g];
    const int li = i - node0;
    const uint32_t* rowi = base + (int64_t)li * stride;
    unsigned long long acc = 0;
    for (int lj = li + 1 + lane; lj < n; lj += 64) {
        const uint32_t* rowj = base + (int64_t)lj * stride;
        unsigned c = 0;
        for (int w = 0; w < words; ++w) c += __popc(rowi[w] & rowj[w]);
        acc += (unsigned long long)c * (c - (c > 0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0 && acc) atomicAdd(ext + (size_t)g * EDGEHIP_NSTAT_EXT + 0, acc);
}

// all-pairs shortest path sums by bit-parallel BFS: wave per source; visited / frontier / next are bitsets in
// registers (lane w holds words w, w+64, ..); a level ORs the rows of the frontier's nodes (coalesced row reads)
__global__ __launch_bounds__(64) void k_stats_paths(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                    unsigned long long* __restrict__ ext) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int g = L.node_graph[i];
    const int node0 = L.node_ptr[g], n = L.node_ptr[g + 1] - node0;
    const int stride = L.adj_stride[g];
    const int words = (n + 31) >> 5;
    const uint32_t* base = adj + L.adj_ptr[g];
    const int li = i - node0;
    uint32_t vis[ST_MAXW], fr[ST_MAXW], nx[ST_MAXW];
#pragma unroll
    for (int k = 0; k < ST_MAXW; ++k) {
        const int w = lane + 64 * k;
        vis[k] = fr[k] = (w == (li >> 5)) ? (1u << (li & 31)) : 0u;
    }
    unsigned long long dsum = 0, dcnt = 0;
    for (int level = 1; level <= n; ++level) {
#pragma unroll
        for (int k = 0; k < ST_MAXW; ++k) nx[k] = 0u;
#pragma unroll
        for (int k = 0; k < ST_MAXW; ++k) {
            if (64 * k >= words) break;
            unsigned long long m = __ballot(fr[k] != 0u);
            while (m) {
                const int l = __builtin_ctzll(m);                   // wave-uniform
                m &= m - 1;
                uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)fr[k], l);
                const int jbase = (l + 64 * k) * 32;
                while (word) {
                    const int b = __builtin_ctz(word);
                    word &= word - 1;
                    const uint32_t* rowj = base + (int64_t)(jbase + b) * stride;
#pragma unroll
                    for (int kk = 0; kk < ST_MAXW; ++kk)
                        if (64 * kk < words) nx[kk] |= (lane + 64 * kk < words) ? rowj[lane + 64 * kk] : 0u;
                }
            }
        }
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < ST_MAXW; ++k) { nx[k] &= ~vis[k]; c += __popc(nx[k]); vis[k] |= nx[k]; fr[k] = nx[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (c == 0) break;                                          // wave-uniform
        dsum += (unsigned long long)level * c;
        dcnt += c;
    }
    if (lane == 0 && dcnt) {
        atomicAdd(ext + (size_t)g * EDGEHIP_NSTAT_EXT + 1, dsum);
        atomicAdd(ext + (size_t)g * EDGEHIP_NSTAT_EXT + 2, dcnt);
    }
}

// connected components by min-label propagation with pointer jumping: one block per graph, labels (local node ids,
// label[x] <= x) in caller scratch.  Labels only decrease, so racy reads are harmless; the loop ends when a whole
// sweep changes nothing (at most n sweeps).  Then the component sizes are counted and the largest is written.
__global__ __launch_bounds__(256) void k_stats_components(edgehip_layout L, const uint32_t* __restrict__ adj,
                                                          int32_t* __restrict__ label, int32_t* __restrict__ cnt,
                                                          unsigned long long* __restrict__ ext) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const int node0 = L.node_ptr[g], n = L.node_ptr[g + 1] - node0;
    const int stride = L.adj_stride[g];
    const int words = (n + 31) >> 5;
    const uint32_t* base = adj + L.adj_ptr[g];
    volatile int32_t* lab = label + node0;
    int32_t* c = cnt + node0;
    for (int i = tid; i < n; i += 256) { lab[i] = i; c[i] = 0; }
    __syncthreads();
    for (int sweep = 0; sweep <= n; ++sweep) {
        int changed = 0;
        for (int i = tid; i < n; i += 256) {
            int m = lab[i];
            const int before = m;
            const uint32_t* row = base + (int64_t)i * stride;
            for (int w = 0; w < words; ++w) {
                uint32_t bits = row[w];
                while (bits) {
                    const int j = w * 32 + __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int lj = lab[j];
                    m = lj < m ? lj : m;
                }
            }
            const int mm = lab[m];                                  // pointer jump
            m = mm < m ? mm : m;
            if (m < before) { lab[i] = m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }
    // every label now names the smallest node of its component (a fixed point of the sweep); flatten once more
    for (int i = tid; i < n; i += 256) atomicAdd(&c[lab[i]], 1);
    __syncthreads();
    __shared__ int best[256];
    int b = 0;
    for (int i = tid; i < n; i += 256) b = c[i] > b ? c[i] : b;
    best[tid] = b;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) best[tid] = best[tid + d] > best[tid] ? best[tid + d] : best[tid];
        __syncthreads();
    }
    if (tid == 0) ext[(size_t)g * EDGEHIP_NSTAT_EXT + 3] = (unsigned long long)best[0];
}

extern "C" int edgehip_graph_stats_ext(const edgehip_layout* L, const uint32_t* adj, int32_t* labels, int32_t* work,
                                       int64_t* ext, void* stream) {
    EH_CHECK_ARG(L && adj && labels && work && ext);
    if (L->num_graphs == 0) return EDGEHIP_OK;
    if (L->max_nodes > 64 * ST_MAXW * 32) return EDGEHIP_ERR_UNSUPPORTED;
    hipStream_t s = eh_stream(stream);
    unsigned long long* o = reinterpret_cast<unsigned long long*>(ext);
    (void)hipMemsetAsync(ext, 0, sizeof(int64_t) * EDGEHIP_NSTAT_EXT * L->num_graphs, s);
    if (L->num_nodes > 0) {
        EH_PROF(EH_K_MISC, s, hipLaunchKernelGGL(k_stats_squares, dim3((L->num_nodes + 3) / 4), dim3(256), 0, s, *L, adj, o));
        EH_PROF(EH_K_MISC, s, hipLaunchKernelGGL(k_stats_paths, dim3(L->num_nodes), dim3(64), 0, s, *L, adj, o));
    }
    EH_PROF(EH_K_MISC, s, hipLaunchKernelGGL(k_stats_components, dim3(L->num_graphs), dim3(256), 0, s, *L, adj, labels, work, o));
    EH_LAUNCH_CHECK();
    return EDGEHIP_OK;
}
