/* edgehip.h - C ABI of libedgehip.so: EDGE's degree-guided binomial diffusion hot
 * path as hand-written gfx950 (MI355X) HIP kernels.
 *
 * Every entry point replaces a function of the reference (tufts-ml/graph-generation-EDGE);
 * the citation after "replaces:" is file:line in the reference tree.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory; no entry point allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - return value: EDGEHIP_OK or a negative error code; nothing throws;
 *   - data-dependent sizes (active nodes K, active pairs A, edges E) stay on the device:
 *     kernels read them from `edgehip_counts` and launch over fixed upper bounds, so the
 *     sampling loop needs no host synchronisation.
 *
 * Process state: the data path keeps none.  Two documented exceptions, both per device and created on
 * first use: (1) edgehip_train_backward* queue the weight-gradient products on a non-blocking side stream
 * (library-owned unless the caller supplies one, edgehip_train_set_side_stream) joined back into `stream`
 * before the call returns its work (csrc/train.hip tr_aux; it never outlives an entry point's ordering: the
 * caller sees one stream); (2) the measurement hooks at
 * the end of this file (off by default).  edgehip_edge_ce_loss_static additionally owns a fixed device
 * scratch array (see there).
 *
 * SURVEY.md 8b lists an op 6 `csr_build`: it is intentionally ABSENT.  The bit-matrix state below IS the
 * adjacency structure (a row is a sorted neighbour list), so no CSR is ever built.
 *
 * State layout (replaces the reference's dense log-one-hot `[P,2]` state and its
 * int64 `full_edge_index [2,P]`, datasets/data_utils.py:53-56): one symmetric BIT MATRIX
 * per graph.  Graph b owns `n_b` rows of `adj_stride[b]` 32-bit words starting at word
 * `adj_ptr[b]`; bit j of row i is 1 iff {i,j} is an edge.  Pair k of the reference's
 * enumeration (`torch.triu_indices(n,n,1)`, row-major) is (i,j), k = i(2n-i-1)/2 + (j-i-1).
 */
#ifndef EDGEHIP_H
#define EDGEHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EDGEHIP_OK 0
#define EDGEHIP_ERR_ARG (-1)          /* null pointer, negative size, misaligned buffer */
#define EDGEHIP_ERR_UNSUPPORTED (-2)  /* configuration outside what the kernels implement */
#define EDGEHIP_ERR_WORKSPACE (-3)    /* workspace too small */
#define EDGEHIP_ERR_LAUNCH (-4)       /* hipGetLastError() != hipSuccess after a launch */
#define EDGEHIP_ERR_TOO_LARGE (-5)    /* input beyond a fixed capacity: the static scratch of an entry point that cannot
                                       * allocate, or a layout whose 32-bit device counters could wrap (below) */

#define EDGEHIP_MAX_LAYERS 8
#define EDGEHIP_TILE_ROWS 32          /* node tile: 32 consecutive nodes of ONE graph */
#define EDGEHIP_PAIR_TILE 32          /* active-pair tile: 32 consecutive active pairs of ONE graph */

/* Batch layout: the native form of the fields the path reads from a PyG Batch
 * (datasets/data_utils.py:69-75).  Built once per batch. */
typedef struct edgehip_layout {
    int32_t num_graphs;          /* B */
    int32_t num_nodes;           /* Nn = sum n_b */
    int32_t num_tiles;           /* NT = sum ceil(n_b / 32) */
    int32_t max_nodes;           /* max n_b */
    int32_t gid_base;            /* global id of graph 0 (keys the counter-based RNG; rank offset) */
    int32_t reserved;
    const int32_t* node_ptr;     /* [B+1] first node of each graph */
    const int32_t* node_graph;   /* [Nn]  graph of each node ('batch') */
    const int64_t* adj_ptr;      /* [B+1] first word of each graph's bit matrix */
    const int32_t* adj_stride;   /* [B]   words per row (>= ceil(n_b/32)) */
    const int64_t* pair_ptr;     /* [B+1] prefix of n_b(n_b-1)/2 ('edges_per_graph' cumsum) */
    const int32_t* tile_ptr;     /* [B+1] first tile of each graph */
    const int32_t* tile_graph;   /* [NT] */
    const int32_t* tile_row0;    /* [NT] first (global) node of the tile */
    const int32_t* tile_rows;    /* [NT] rows in the tile, 1..32 */
} edgehip_layout;

/* Device-side counters written by edgehip_active_offsets and read by later kernels. */
typedef struct edgehip_counts {
    int32_t k_total;      /* active nodes in the batch */
    int32_t pair_tiles;   /* total active-pair tiles */
    int64_t a_total;      /* active pairs in the batch */
    int32_t status;       /* 0, or a bit mask of EDGEHIP_STATUS_* raised by kernels */
    int32_t reserved;
} edgehip_counts;
#define EDGEHIP_STATUS_CAPACITY 1   /* an optional output buffer was too small (output truncated) */

/* Counter-based uniforms (Philox4x32-10) for throughput runs.  u = (x >> 8) * 2^-24, the
 * grid torch.rand emits on CPU.  Keyed (seed; graph global id, step, stream, element), so
 * results do not depend on how graphs are sharded over GPUs. */
typedef struct edgehip_rng {
    uint64_t seed;
} edgehip_rng;

/* ---- a8: degrees ---------------------------------------------------------------------
 * replaces: BinomialDiffusionActive._compute_degree, diffusion/diffusion_binomial_active.py:92-95
 * deg[i] = popcount(row i).  int32, exact. */
int edgehip_degree_from_adj(const edgehip_layout* L, const uint32_t* adj, int32_t* deg, void* stream);

/* ---- a9+a10: degree-guided active-node selection (sampling) ---------------------------
 * replaces: _q_posterior_actives + log_sample_categorical + nonzero,
 *           diffusion/diffusion_binomial_active.py:97-126, diffusion/diffusion_base.py:175-180
 * n = max(d0 - deg, 0); log-probs = post_table[t_graph[b]][n] (host-built with the reference's
 * exact torch op sequence, [T][nmax+1][2] fp32); node i is active iff
 * g(u[i][1]) + lp1 > g(u[i][0]) + lp0, g(u) = -log(-log(u+1e-30)+1e-30).
 * uniforms: [Nn,2] fp32, or NULL to use `rng` (stream 0).  gumbel_patch: NULL, or the patch blob described
 * under edgehip_gumbel_grid below (decisions bit-identical to a CPU reference for every u on the 2^-24 grid).
 * Outputs: flags[Nn] (0/1); act_nodes[Nn]: graph b's active nodes, ascending, as global ids in
 * act_nodes[node_ptr[b] ...]; act_count[B]. */
int edgehip_active_select_sample(const edgehip_layout* L, const int32_t* d0, const int32_t* deg,
                                 const int32_t* t_graph, const float* post_table, int32_t nmax,
                                 const float* uniforms, const edgehip_rng* rng, const float* gumbel_patch,
                                 uint8_t* flags, int32_t* act_nodes, int32_t* act_count, void* stream);

/* ---- a3: the Gumbel transform of every random decision -----------------------------------
 * replaces: log_sample_categorical, diffusion/diffusion_base.py:175-180 (gumbel_noise = -log(-log(u+1e-30)+1e-30))
 * Every decision of the path is g(u1) + lp1 > g(u0) + lp0 in fp32 (ties -> class 0, torch.argmax).  The host's g is
 * its math library's, within an ulp of the correctly rounded value but not always equal to it, so a device
 * logarithm alone cannot reproduce a CPU run's near-ties.  fp32 uniforms from torch.rand (and from edgehip_rng) lie
 * on the grid k 2^-24: rounds 1-2 shipped the host's g at all 2^24 grid points (64 MB, one cache line fetched per
 * decision).  Now g is ARITHMETIC on the device plus a small patch list:
 *   edgehip_gumbel_grid evaluates, for u = (k0 + i) 2^-24, i < n, the device's two forms of g - `exact` (both
 *   logarithms in fp64, rounded to fp32) and `fast` (two hardware log2) - into exact_out / fast_out (either may be
 *   NULL).  `which` selects the copy compiled into the active-selection / forward-noising kernels (0) or the one in
 *   the pair kernel (1); they are the same instructions and a test shows them equal.
 *   The caller (edgehip/tables.py, once per process and device) compares `exact` with the host's own table over
 *   ALL 2^24 points and builds the blob the `gumbel_patch` arguments take: 32-bit words [0] 0x45484750, [1] n,
 *   [2] eps_fast = max |fast - host| over the grid (float), [3] 0, then n ascending keys k with exact != host and
 *   the n host values (float).  n depends on the host's math library: 15 862 (127 KB) where torch's MKL runs its
 *   Intel code path, 2 380 501 (19 MB) on an AMD EPYC host (MKL's generic path is one ulp off the correctly rounded
 *   fp32 logarithm at 7.5 % of the grid); the list is only read at near-ties.
 * A decision uses `fast`; if |a1 - a0| <= 2 eps_fast + rounding it is re-taken on the host's own values (`exact`,
 * patched).  Both outcomes are the host's decision for every on-grid u: exhaustively by construction, not
 * statistically.  gumbel_patch == NULL: `exact` only (device logarithms; no guarantee against a CPU run). */
int edgehip_gumbel_grid(float* exact_out, float* fast_out, int64_t k0, int64_t n, int32_t which, void* stream);

/* ---- a11: active nodes in training ----------------------------------------------------
 * replaces: _q_set_actives, diffusion/diffusion_binomial_active.py:134-140
 * active iff deg_tm1[i] > deg_t[i]. Outputs as above. */
int edgehip_active_select_train(const edgehip_layout* L, const int32_t* deg_tm1, const int32_t* deg_t,
                                uint8_t* flags, int32_t* act_nodes, int32_t* act_count, void* stream);

/* ---- offsets of the compacted sets -----------------------------------------------------
 * replaces: the implicit sizes of `.nonzero()` results, diffusion_binomial_active.py:126-128
 * kptr[B+1] prefix of K_b; aptr[B+1] prefix of A_b = K_b(K_b-1)/2; tptr[B+1] prefix of
 * ceil(A_b/32) pair tiles; totals into *counts (status is cleared). */
int edgehip_active_offsets(const edgehip_layout* L, const int32_t* act_count, int32_t* kptr, int64_t* aptr,
                           int32_t* tptr, edgehip_counts* counts, void* stream);

/* tptr / counts->pair_tiles and the tile loops of the pair kernels are 32-bit.  The reference's index lists are
 * int64 (`nonzero`, diffusion_binomial_active.py:126-128), so there a batch is only bounded by memory; here a layout
 * whose WORST-CASE tile count (every node of every graph active: B * (ceil(n_max(n_max-1)/2 / 32) + 1), what
 * edgehip_layout_pair_tile_bound returns) exceeds EDGEHIP_MAX_PAIR_TILES is refused with EDGEHIP_ERR_TOO_LARGE by
 * edgehip_active_offsets and by everything that calls it (edgehip_sample_step), before anything is launched.
 * 20 000-node graphs: 6.25 Mi tiles each, up to 85 graphs per batch. */
#define EDGEHIP_MAX_PAIR_TILES ((int64_t)1 << 29)
int64_t edgehip_layout_pair_tile_bound(const edgehip_layout* L);

/* ---- a10: materialise the reference's index lists ---------------------------------------
 * replaces: active_node_indices / active_edge_indices, diffusion_binomial_active.py:126-128,140-143
 * active_node_indices[K] int64 ascending; active_edge_indices[A] int64 ascending (indices into
 * the reference's P enumeration), with global row/col ids.  Any output may be NULL.
 * Capacities in elements; truncation raises EDGEHIP_STATUS_CAPACITY in counts->status. */
int edgehip_active_pairs_enumerate(const edgehip_layout* L, const int32_t* act_nodes, const int32_t* kptr,
                                   const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                                   int64_t* active_node_indices, int64_t node_cap,
                                   int64_t* active_edge_indices, int64_t* row, int64_t* col, int64_t pair_cap,
                                   void* stream);

/* ---- a4-a7: forward noising --------------------------------------------------------------
 * replaces: _q_sample_and_set_xtmin1_xt_given_x0, diffusion/diffusion_binomial_vanilla.py:192-212
 *           (q_sample diffusion_base.py:154-163, _q_pred :272-288, _q_pred_one_timestep :226-243)
 * For every edge of x0 (only those can survive: SURVEY App. A5): kept in x_{t-1} iff t==0 or
 * g(u1)+q_cum[t-1][1] > g(u0)+q_cum[t-1][0]; kept in x_t iff kept in x_{t-1} and
 * g(v1)+q_one[t][1] > g(v0)+q_one[t][0].  q_cum/q_one: [T][2] host-built log-probs of a source
 * bit 1.  u_tm1/u_t: dense [P,2] uniforms addressed by pair index, or NULL to use rng (streams 2,3).
 * adj_tm1/adj_t must be zero-filled by the caller. */
int edgehip_q_sample_edges(const edgehip_layout* L, const uint32_t* adj0, const int32_t* t_graph,
                           const float* q_cum, const float* q_one, const float* u_tm1, const float* u_t,
                           const edgehip_rng* rng, const float* gumbel_patch, uint32_t* adj_tm1,
                           uint32_t* adj_t, void* stream);

/* ---- a17 epilogue: edge list of the state ------------------------------------------------
 * replaces: the epilogue of DiffusionBase.sample, diffusion/diffusion_base.py:209-228
 * count: edges_per_graph[B] and edge_slice[B+1] (prefix) from the upper triangle.
 * extract: edge_index [2,E] int64 (row 0 = i, row 1 = j, global ids, i<j, ascending pair order),
 * `edge_cap` = E the caller allocated (read edge_slice[B] after a sync). */
int edgehip_count_edges(const edgehip_layout* L, const uint32_t* adj, int32_t* row_upper, int64_t* edge_slice,
                        void* stream);
int edgehip_extract_edges(const edgehip_layout* L, const uint32_t* adj, const int32_t* row_upper,
                          const int64_t* edge_slice, int64_t* edge_index, int64_t edge_cap, void* stream);

/* set bits of an edge list into a zero-filled bit matrix (host helper for x0; both directions).
 * edges: [E,2] int32 global node ids. */
int edgehip_adj_from_edges(const edgehip_layout* L, const int32_t* edges, int64_t num_edges, uint32_t* adj,
                           void* stream);

/* ---- a12-a15: the denoising network -------------------------------------------------------
 * Parameters of TGNN_degree_guided (layers/layers.py:250-313), torch layouts ([out,in] row-major
 * fp32), named after its state_dict. */
typedef struct edgehip_gnn_params {
    int32_t dim;                 /* must be 64 */
    int32_t num_layers;          /* len(num_heads) <= 8 */
    int32_t heads[EDGEHIP_MAX_LAYERS];   /* each in {1,2,4,8} */
    int32_t max_degree;
    int32_t num_steps;           /* T */
    const float *embedding_t_w, *embedding_t_b, *embedding_0_w, *embedding_0_b, *embedding_sel_w;
    const float *node_in_w, *node_in_b;
    const float *mlp_w0, *mlp_b0, *mlp_w2, *mlp_b2;               /* time mlp */
    const float *global_w0, *global_b0, *global_w2, *global_b2;   /* global_mlp */
    const float *context_w0, *context_b0, *context_w2, *context_b2;
    const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh;
    const float *time_w[EDGEHIP_MAX_LAYERS], *time_b[EDGEHIP_MAX_LAYERS];
    const float *query_w[EDGEHIP_MAX_LAYERS], *query_b[EDGEHIP_MAX_LAYERS];
    const float *key_w[EDGEHIP_MAX_LAYERS], *key_b[EDGEHIP_MAX_LAYERS];
    const float *value_w[EDGEHIP_MAX_LAYERS], *value_b[EDGEHIP_MAX_LAYERS];
    const float *skip_w[EDGEHIP_MAX_LAYERS], *skip_b[EDGEHIP_MAX_LAYERS];
    const float *node_out_w0, *node_out_b0, *node_out_w2, *node_out_b2, *node_out_w4, *node_out_b4;
    const float *final_w0, *final_b0, *final_w2, *final_b2, *final_w4, *final_b4;
    const float* pos_emb;        /* [T,64] sinusoidal table built on the host (layers.py:45-60) */
} edgehip_gnn_params;

/* Folded weights: products of the raw matrices that let the kernels skip work the reference
 * does per node (DESIGN.md "Algebra").  `folded` is a device buffer of
 * edgehip_gnn_folded_floats(P) floats; edgehip_gnn_prepare fills it and must be called again
 * whenever the raw parameters change. */
size_t edgehip_gnn_folded_floats(const edgehip_gnn_params* P);
int edgehip_gnn_prepare(const edgehip_gnn_params* P, float* folded, void* stream);

/* Workspace (bytes) for edgehip_gnn_forward / edgehip_sample_step / edgehip_loss_forward. */
size_t edgehip_gnn_workspace_bytes(const edgehip_layout* L, const edgehip_gnn_params* P);

/* replaces: TGNN_degree_guided.forward, layers/layers.py:315-383 (+ TransformerConv :354,
 *           GRU :360, scatter-mean :342,367)
 * Inputs: state bit matrix, its degrees, d0, per-graph t, the active set (flags, act_nodes, kptr,
 * aptr, tptr, counts).  Output: `node_repr` [K_total,128] (compact, active-node order): the
 * per-active-node vector v with  edge_logits(a,b) = final_tail(silu(v_a + v_b + final_b0)).
 * Optional debug output h_out [Nn,64]: node state after the last layer (reference `nodes` at
 * layers.py:371). */
int edgehip_gnn_forward(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                        const uint32_t* adj, const int32_t* deg, const int32_t* d0, const int32_t* t_graph,
                        int32_t t_shared, const uint8_t* flags, const int32_t* act_nodes, const int32_t* kptr,
                        const edgehip_counts* counts, float* node_repr, float* h_out, void* ws,
                        size_t ws_bytes, void* stream);

/* ---- a16: Bernoulli re-sampling of the active pairs ---------------------------------------
 * replaces: final_out on active pairs layers.py:380-381, F.log_softmax active:153-154,
 *           log_sample_categorical + in-place overwrite active:83-88
 * For active pair k of graph b (a_p < a_q the p-th/q-th active node): logits -> log_softmax ->
 * bit = g(u1)+lp1 > g(u0)+lp0; the state bit of {a_p,a_q} is OVERWRITTEN (both directions).
 * uniforms: [A,2] in active-pair order, or NULL for rng (stream 1, keyed by the pair's index in
 * the P enumeration).  Optional outputs (NULL to skip): logits_out [A,2], bits_out [A] (cap pair_cap).
 * logits_in: NULL, or [A,2] externally supplied logits that replace the network's (parity tests:
 * "sampled edges bit-exact given logits + uniforms"). `logits_are_logprobs` != 0 skips log_softmax.
 * adj == NULL: logits-only call (the denoiser contract `denoise_fn(batch, t_node, t_edge) -> edge_logits`,
 * active:148): logits_out is filled, nothing is drawn and no state is touched (uniforms/rng may be NULL). */
int edgehip_edge_resample(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                          const float* node_repr, const int32_t* act_nodes, const int32_t* kptr,
                          const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                          const float* logits_in, int32_t logits_are_logprobs, const float* uniforms,
                          const edgehip_rng* rng, int32_t step, const float* gumbel_patch, uint32_t* adj,
                          float* logits_out, uint8_t* bits_out, int64_t pair_cap, void* stream);

/* ---- a18: cross-entropy of the active pairs against x_{t-1} --------------------------------
 * replaces: _compute_MC_KL_joint, diffusion/diffusion_binomial_active.py:157-181
 * kl[b] = sum_k CE_k + 6.9078e-29*(n(n-1)/2 - A_b) + 6.9078e-29*n(n-1)/2.  tile_partial: scratch of
 * counts->pair_tiles floats (capacity tile_cap).  Optional logits_out as above. */
int edgehip_edge_ce_loss(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                         const float* node_repr, const int32_t* act_nodes, const int32_t* kptr,
                         const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                         const uint32_t* adj_tm1, float* tile_partial, int64_t tile_cap, float* kl,
                         float* logits_out, int64_t pair_cap, void* stream);

/* ---- the same loss without any caller scratch -------------------------------------------------
 * NO COUNTERPART in the reference.  For callers that cannot allocate device memory or size a workspace:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Identical to
 * edgehip_edge_ce_loss (same kernels, same results bit for bit) except that the per-tile partial sums live in
 * a fixed device array owned by the library: edgehip_edge_ce_loss_static_max_tiles() floats (2 Mi = 8 MiB) per
 * GPU the library runs on, held until the process ends.  The call allocates and frees nothing.
 * Limit: num_graphs * (ceil(max_nodes*(max_nodes-1)/2 / 32) + 1) must not exceed the maximum (64 polblogs
 * graphs fit), else EDGEHIP_ERR_TOO_LARGE and nothing is launched.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams) race. */
int64_t edgehip_edge_ce_loss_static_max_tiles(void);
int edgehip_edge_ce_loss_static(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                                const float* node_repr, const int32_t* act_nodes, const int32_t* kptr,
                                const int64_t* aptr, const int32_t* tptr, edgehip_counts* counts,
                                const uint32_t* adj_tm1, float* kl, float* logits_out, int64_t pair_cap,
                                void* stream);

/* ---- a16+a10+a12 fused: one reverse step for the whole batch --------------------------------
 * replaces: BinomialDiffusionActive.p_sample, diffusion/diffusion_binomial_active.py:72-90
 * Sequence: degree -> select -> offsets -> gnn_forward -> edge_resample, all on `stream`, no host
 * sync.  `t` is the shared step; post_table as in edgehip_active_select_sample.  u_node [Nn,2] and
 * u_pair [A,2] may be NULL (rng).  The scratch arrays live in `ws`. */
int edgehip_sample_step(const edgehip_layout* L, const edgehip_gnn_params* P, const float* folded,
                        uint32_t* adj, const int32_t* d0, int32_t t, const float* post_table, int32_t nmax,
                        const float* u_node, const float* u_pair, const edgehip_rng* rng,
                        const float* gumbel_patch, edgehip_counts* counts, void* ws, size_t ws_bytes,
                        void* stream);

/* ---- generic fp32 MFMA linear layer (exposed for tests / composition) -----------------------
 * replaces: torch.nn.Linear calls of layers/layers.py:257-313
 * Y[M,N] = act(X[M,K] W[N,K]^T + bias[N]); K % 64 == 0, N % 32 == 0; act 0 none, 1 SiLU. */
int edgehip_linear_fwd(const float* X, int64_t M, int32_t K, const float* W, const float* bias, int32_t N,
                       int32_t act, float* Y, void* stream);

/* ================================================================================================
 * Training ops: forward kernels that keep what the backward needs + their backward twins
 * (SURVEY.md 8b "ops ... _bwd twins in training").  edgehip/train.py composes them, one
 * torch.autograd.Function per op; nothing is folded here, so gradients land on the reference's own
 * parameters.  All buffers fp32 row-major unless noted; "accumulates" = float atomics into a buffer the
 * caller zero-filled.
 * ================================================================================================ */

/* replaces: embedding_t / embedding_0 / embedding_sel + cat, layers/layers.py:326-336
 * X0[Nn,192] = [ w_t*min(deg,dmax+1)/dmax + b_t | w_0*d0/dmax + b_0 | emb_sel[flag] ].
 * bwd accumulates dwt,dbt,dw0,db0 [64] and dwsel [2,64]. */
int edgehip_train_embed_fwd(const edgehip_layout* L, const int32_t* deg, const int32_t* d0, const uint8_t* flags,
                            int32_t max_degree, const float* wt, const float* bt, const float* w0, const float* b0,
                            const float* wsel, float* X0, void* stream);
int edgehip_train_embed_bwd(const edgehip_layout* L, const int32_t* deg, const int32_t* d0, const uint8_t* flags,
                            int32_t max_degree, const float* dX0, float* dwt, float* dbt, float* dw0, float* db0,
                            float* dwsel, void* stream);

/* replaces: autograd of torch.nn.Linear (every Linear of layers/layers.py:257-313)
 * bwd_input:  dX[M,K] = dY[M,N] W[N,K]   (N % 64 == 0, K % 32 == 0; WT_scratch: K*N floats)
 * bwd_weight: dW[N,K] += dY^T X, db[N] += column sums of dY (db may be NULL); accumulates. */
int edgehip_linear_bwd_input(const float* dY, int64_t M, int32_t N, const float* W, int32_t K, float* WT_scratch,
                             float* dX, void* stream);
int edgehip_linear_bwd_weight(const float* dY, const float* X, int64_t M, int32_t N, int32_t K, float* dW, float* db,
                              void* stream);

/* replaces: nn.SiLU (kind 1) and Mish (kind 2, layers/layers.py:40-42) and their autograd */
int edgehip_act_fwd(int32_t kind, const float* x, int64_t n, float* y, void* stream);
int edgehip_act_bwd(int32_t kind, const float* x, const float* dy, int64_t n, float* dx, void* stream);

/* replaces: `nodes + contexts`, torch.cat with the repeat_interleave'd per-graph rows,
 *           layers/layers.py:345,350-353,366,369-371,377
 * out[Nn,64+Kt] = [ h[i] + ctx[g(i)] | tail_graph[g(i)] (Kt = 64) or tail_rows[i] (any Kt) ]; ctx may be NULL.
 * bwd: dh[Nn,64]; dctx[B,64], dtail_graph[B,64] per-graph column sums; dtail_rows[Nn,Kt] (any may be NULL). */
int edgehip_assemble_fwd(const edgehip_layout* L, const float* h, const float* ctx, const float* tail_graph,
                         const float* tail_rows, int32_t Kt, float* out, void* stream);
int edgehip_assemble_bwd(const edgehip_layout* L, const float* dout, int32_t Kt, float* dh, float* dctx,
                         float* dtail_graph, float* dtail_rows, void* stream);

/* replaces: torch_scatter.scatter(reduce='mean') over `batch`, layers/layers.py:342,367
 * out[B,64] = add[B,64] (or 0) + per-graph mean of x[Nn,64];  bwd: dx[i] = dout[g(i)] / n_g. */
int edgehip_segment_mean_fwd(const edgehip_layout* L, const float* x, const float* add, float* out, void* stream);
int edgehip_segment_mean_bwd(const edgehip_layout* L, const float* dout, float* dx, void* stream);

/* replaces: pyg.nn.TransformerConv message passing + softmax + mean over heads + root skip
 *           (call site layers/layers.py:354) over the bit-matrix neighbourhoods.
 * q,k,v [Nn,heads*64]; skip,out [Nn,64]; saved for backward: O [Nn,heads*64] per-head outputs,
 * lse [Nn,heads].  bwd: Dbuf [Nn,heads] scratch; dq,dk,dv [Nn,heads*64] (dskip = dout). */
int edgehip_attn_fwd(const edgehip_layout* L, const uint32_t* adj, int32_t heads, const float* q, const float* k,
                     const float* v, const float* skip, float* out, float* O, float* lse, void* stream);
int edgehip_attn_bwd(const edgehip_layout* L, const uint32_t* adj, int32_t heads, const float* q, const float* k,
                     const float* v, const float* O, const float* lse, const float* dout, float* Dbuf, float* dq,
                     float* dk, float* dv, void* stream);

/* replaces: the gate arithmetic of torch.nn.GRU(64,64), one step, layers/layers.py:360
 * gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh ([Nn,192], gates r,z,n) -> h' [Nn,64]. */
int edgehip_gru_gate_fwd(const float* gi, const float* gh, const float* h, int64_t Nn, float* hout, void* stream);
int edgehip_gru_gate_bwd(const float* gi, const float* gh, const float* h, const float* dhout, int64_t Nn, float* dgi,
                         float* dgh, float* dh, void* stream);

/* replaces: nodes[row] + nodes[col] on the active pairs, layers/layers.py:380 (bwd accumulates into dz) */
int edgehip_pair_gather_add_fwd(const float* z, int32_t width, const int64_t* row, const int64_t* col, int64_t A, float* e,
                                void* stream);
int edgehip_pair_gather_add_bwd(const float* de, int32_t width, const int64_t* row, const int64_t* col, int64_t A,
                                float* dz, void* stream);

/* replaces: log_full_edge_attr_tmin1[active_edge_indices] and batch[row], diffusion_binomial_active.py:165-170
 * tgt[A] = bit of pair (row,col) in x_{t-1}; pair_graph[A] = graph of the pair. */
int edgehip_pair_targets(const edgehip_layout* L, const uint32_t* adj_tm1, const int64_t* row, const int64_t* col, int64_t A,
                         uint8_t* tgt, int32_t* pair_graph, void* stream);

/* replaces: final_out[4] (64 -> 2) layers/layers.py:381 + log_softmax + cross-entropy + per-graph sum,
 *           _compute_MC_KL_joint diffusion_binomial_active.py:157-175
 * fwd: logits[A,2]; ce_graph[B] accumulates sum_k CE_k.  bwd: gkl[B] = dLoss/dce_graph;
 * de2[A,64]; dW4[2,64], db4[2] accumulate. */
int edgehip_edge_ce_fwd(const float* e2, const float* W4, const float* b4, const uint8_t* tgt, const int32_t* pair_graph,
                        int64_t A, float* logits, float* ce_graph, void* stream);
int edgehip_edge_ce_bwd(const float* e2, const float* W4, const float* b4, const uint8_t* tgt, const int32_t* pair_graph,
                        const float* gkl, int64_t A, float* de2, float* dW4, float* db4, void* stream);

/* ---- dropout of the training forward ----------------------------------------------------------------
 * replaces: self.dropout = torch.nn.Dropout(p=dropout) in training mode, layers/layers.py:286, applied to the
 *           conv output (:357, slot 2i) and to the GRU hidden state carried to the next layer (:361, slot 2i+1)
 * y[Nn,64] = x * keep / (1 - p) (in place allowed).  keep: dp->masks ([2*num_layers][Nn][64] uint8 0/1, slot-major;
 * parity runs give the reference the same masks) or, when masks == NULL, Philox keyed (seed; graph global id, node,
 * channel group, slot): shard-invariant, reproducible.  The backward of the op is the op itself on the gradient. */
typedef struct edgehip_dropout {
    float p;                  /* drop probability, 0 <= p < 1 (0: the op is the identity) */
    int32_t reserved;
    uint64_t seed;
    const uint8_t* masks;     /* device pointer or NULL */
} edgehip_dropout;
int edgehip_dropout_fwd(const edgehip_layout* L, const float* x, const edgehip_dropout* dp, int32_t slot, float* y,
                        void* stream);

/* ---- the whole differentiable network in two calls ------------------------------------------------
 * replaces: TGNN_degree_guided.forward layers/layers.py:315-383 + _compute_MC_KL_joint
 *           diffusion_binomial_active.py:157-175 under autograd, and loss.backward() through them
 *           (diffusion/experiment.py:29-33).
 * The library queues the ops above itself (same kernels, same order), so a training step costs two host
 * calls instead of one autograd node per op.  `row`,`col` [A] int64: the active pairs (global node ids,
 * edgehip_active_pairs_enumerate); tgt/pair_graph from edgehip_pair_targets.  forward writes
 * ce_graph[B] = sum_k CE_k and logits[A,2] and leaves every activation the backward needs in `ws`
 * (edgehip_train_workspace_bytes(L,P,A) bytes, caller-owned, must stay untouched until backward returns).
 * backward: gce[B] = dLoss/d ce_graph; G has the layout of P, its pointers address GRADIENT buffers of the
 * parameters' shapes; gradients are ADDED (float atomics / read-modify-write), so the caller zero-fills them
 * once per step.  pos_emb and the integer fields of G are ignored. */
size_t edgehip_train_workspace_bytes(const edgehip_layout* L, const edgehip_gnn_params* P, int64_t A);
int edgehip_train_forward(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                          const int32_t* deg_t, const int32_t* d0, const uint8_t* flags, const int32_t* t_graph,
                          const int64_t* row, const int64_t* col, int64_t A, const uint8_t* tgt,
                          const int32_t* pair_graph, float* ce_graph, float* logits, void* ws, size_t ws_bytes,
                          void* stream);
/* The same step in three calls for callers that queue work before they know the number of active pairs: the node
 * part (everything up to the per-node output) only needs a CAPACITY A_cap >= A that fixes the workspace layout
 * (edgehip_train_workspace_bytes(L,P,A_cap)); the pair part and the backward then take the real A and the same
 * A_cap.  edgehip_train_forward / _backward are these with A_cap = A. */
int edgehip_train_forward_nodes(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                                const int32_t* deg_t, const int32_t* d0, const uint8_t* flags, const int32_t* t_graph,
                                int64_t A_cap, void* ws, size_t ws_bytes, void* stream);
int edgehip_train_forward_pairs(const edgehip_layout* L, const edgehip_gnn_params* P, const int64_t* row,
                                const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                                const int32_t* pair_graph, float* ce_graph, float* logits, void* ws, size_t ws_bytes,
                                void* stream);
int edgehip_train_backward_cap(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                               const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                               const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                               const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream);
/* The node part and the backward with training-mode dropout (--dp_rate > 0, README.md:49,79); dp == NULL or
 * dp->p == 0 is edgehip_train_forward_nodes / edgehip_train_backward_cap.  The same `dp` must be given to both. */
int edgehip_train_forward_nodes_dp(const edgehip_layout* L, const edgehip_gnn_params* P, const uint32_t* adj_t,
                                   const int32_t* deg_t, const int32_t* d0, const uint8_t* flags, const int32_t* t_graph,
                                   int64_t A_cap, const edgehip_dropout* dp, void* ws, size_t ws_bytes, void* stream);
int edgehip_train_backward_dp(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                              const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                              const int64_t* row, const int64_t* col, int64_t A, int64_t A_cap, const uint8_t* tgt,
                              const int32_t* pair_graph, const float* gce, const edgehip_dropout* dp, void* ws,
                              size_t ws_bytes, void* stream);
int edgehip_train_backward(const edgehip_layout* L, const edgehip_gnn_params* P, const edgehip_gnn_params* G,
                           const uint32_t* adj_t, const int32_t* deg_t, const int32_t* d0, const uint8_t* flags,
                           const int64_t* row, const int64_t* col, int64_t A, const uint8_t* tgt,
                           const int32_t* pair_graph, const float* gce, void* ws, size_t ws_bytes, void* stream);
/* The side stream of edgehip_train_backward* on the CURRENT device (see "Process state" at the top): by default the
 * library creates one on first use.  A caller that keeps side streams of its own passes one here (NULL: back to a
 * library-owned stream) so that the process does not hold more streams than it has hardware queues; the stream must
 * outlive every later backward on this device.  No counterpart in the reference (single stream). */
int edgehip_train_set_side_stream(void* stream);


/* ---- statistics of the sampled graphs (next row after the path) ------------------------
 * replaces: the degree / triangle / assortativity part of compute_graph_statistics,
 *           eval_utils/graph_statistics.py:9-123,202-230,249-285 (consumer datasets/evaluator.py:27-37)
 * out [B][EDGEHIP_NSTAT] int64, exact integer sums per graph:
 *   0 edges E; 1 wedges sum d(d-1)/2; 2 claws sum d(d-1)(d-2)/6; 3 sum over edges {i,j} of |N(i) & N(j)|
 *   (= 3 x triangles); 4 sum_edges d_i d_j; 5 sum_edges (d_i + d_j); 6 sum_edges (d_i^2 + d_j^2);
 *   7 d_max; 8 d_min; 9 sum d.   deg: edgehip_degree_from_adj of the same state. */
#define EDGEHIP_NSTAT 10
int edgehip_graph_stats(const edgehip_layout* L, const uint32_t* adj, const int32_t* deg, int64_t* out, void* stream);

/* replaces: LCC (eval_utils/graph_statistics.py:60-73), square_count (:126-143) and cpl (:233-246) of
 *           compute_graph_statistics (:249-285), again as exact integer sums from the bit matrix.
 * ext [B][4] int64: 0 sum over node pairs i<j of c(c-1), c = common neighbours (square_count = that / 4);
 *   1 sum of shortest-path lengths over ordered pairs (i,j), j != i reachable from i; 2 the number of such pairs
 *   (cpl = 1 / 2); 3 size of the largest connected component.
 * labels [Nn] int32 out: component label of every node = LOCAL id of the smallest node of its component
 * (min-label propagation with pointer jumping, one block per graph); work [Nn] int32 scratch. */
int edgehip_graph_stats_ext(const edgehip_layout* L, const uint32_t* adj, int32_t* labels, int32_t* work, int64_t* ext,
                            void* stream);

/* replaces: edge_overlap, eval_utils/graph_statistics.py:287-299.  out[B] = popcount(A & B) per graph
 * (= 2 x shared edges) for two states of the same layout. */
int edgehip_edge_overlap(const edgehip_layout* L, const uint32_t* adj_a, const uint32_t* adj_b, int64_t* out,
                         void* stream);

/* ---- measurement hooks (no counterpart in the reference; used by bench.py's roofline leg) -------
 * When enabled, every kernel launch of the path is bracketed by HIP events on its launch stream.
 * collect() synchronises on those events and returns, per kernel class, the summed device time (ms)
 * and the number of launches since the last collect; arrays are HOST memory of
 * edgehip_prof_num_kernels() entries.  Off by default (one branch per launch). */
int edgehip_prof_enable(int on);
int edgehip_prof_collect(double* ms_host, int64_t* launches_host, int32_t n);
const char* edgehip_prof_kernel_name(int32_t id);
/* node rows the fused layer kernel processed since the last reset (graphs without active pairs are skipped);
 * synchronises.  bench.py prices the roofline with the work that was executed, not the batch size. */
int edgehip_prof_layer_rows(unsigned long long* out_host, int reset);
/* (target, neighbour, head) visits of the same kernel's attention phase since the last reset: 256 fp32 flops each
 * (64-channel score + 64-channel weighted sum); synchronises. */
int edgehip_prof_layer_visits(unsigned long long* out_host, int reset);
int32_t edgehip_prof_num_kernels(void);
/* what the LAST edgehip_sample_step on workspace `ws` worked on; out_dev: int64[8] on the device =
 * {active nodes K, active pairs A, graphs with >= 2 active nodes (the rest were skipped), live node rows after the
 * isolated-class compaction, rows of those graphs with more than 48 neighbours, rows walked by a whole tile together
 * (more than 192), sum of degrees of the state the step started from (2 E_t),
 * nodes of the non-skipped graphs}.  Asynchronous on `stream`. */
int edgehip_prof_step_stats(const edgehip_layout* L, const edgehip_gnn_params* P, void* ws, size_t ws_bytes,
                            const edgehip_counts* counts, int64_t* out_dev, void* stream);

const char* edgehip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EDGEHIP_H */
