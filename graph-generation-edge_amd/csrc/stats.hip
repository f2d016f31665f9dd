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
