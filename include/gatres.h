/*
 * gatres.h -- C-ABI of the MI355X (gfx950) GATRes message-passing engine.
 *
 * Drop-in boundary for ONE hot path of DiTEC-project/gnn-pressure-estimation: the
 * GATConv-with-residual stack `GATResMeanConv` (reference gnn_pressure_estimation/GraphModels.py:454-494),
 * forward + backward, and the training step around it (train.py:159-190).  The reference has no FFI of its
 * own for this path (it is pure Python over torch_geometric); these entry points are what a ctypes binding
 * on the reference side would call (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - feature matrices are row-major fp32, [rows, cols] with cols contiguous;
 *   - graph index arrays are int32;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no allocation, no
 *     synchronisation) so a caller may capture any sequence of calls into a hipGraph;
 *   - return value: 0 = ok; < 0 = argument error (GATRES_E_*); > 0 = hipError_t from the launch.
 *
 * Supported widths: nc (hidden channels) a power of two, 4 <= nc <= 128  (gatres_small: 32, gatres_large: 128).
 */
#ifndef GATRES_H
#define GATRES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GATRES_ABI_VERSION 6

#define GATRES_E_BADARG      (-1)  /* null pointer, negative size, misaligned pointer            */
#define GATRES_E_UNSUPPORTED (-2)  /* width not supported by the gfx950 kernels                  */
#define GATRES_E_GRAPH       (-3)  /* edge endpoint out of range while building the graph plan   */

#define GATRES_DTYPE_F32  0        /* storage types of activation-sized tensors (gatres_model_t.act_dtype) */
#define GATRES_DTYPE_BF16 1

/* --------------------------------------------------------------------------------------------------------
 * Graph plan (K0).  Replaces what PyG re-derives inside every GATConv.forward call --
 * remove_self_loops + add_self_loops (torch_geometric.utils.loop; reference call sites
 * GraphModels.py:464-465) -- and the scatter indices of GATConv / SimpleConv (GraphModels.py:464-466).
 * Built once per topology.
 *   gat CSR   : edges of edge_index with src != dst, then one self loop per node appended (PyG order),
 *               stably sorted by destination.              rowptr[N+1], col[E'] = source node.
 *   gat CSR^T : the same edge list stably sorted by source. t_rowptr[N+1], t_eid[E'] = position of the
 *               edge in the destination-sorted list, t_dst[E'] = destination node.
 *   mean CSR  : the ORIGINAL edge_index (self loops kept, none added) sorted by destination (SimpleConv).
 *   mean CSR^T: sorted by source.  mt_dst[E] = destination node.
 * ------------------------------------------------------------------------------------------------------ */
typedef struct gatres_graph {
  int32_t num_nodes;        /* N                                         */
  int32_t num_edges_gat;    /* E' = #(src != dst) + N                    */
  int32_t num_edges_mean;   /* E  = edge_index.shape[1]                  */
  int32_t num_segments;     /* 0 = no segment table (per-op kernels only) */
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* t_rowptr;
  const int32_t* t_eid;
  const int32_t* t_dst;
  const int32_t* m_rowptr;
  const int32_t* m_col;
  const int32_t* mt_rowptr;
  const int32_t* mt_dst;
  /* Segments: contiguous node ranges [seg_ptr[s], seg_ptr[s+1]) that no edge leaves -- the snapshots of a PyG
   * Batch (train.py:302-303).  The fused kernels give each segment to one workgroup. */
  const int32_t* seg_ptr;   /* [num_segments + 1]                         */
  int32_t max_segment_nodes;
  int32_t max_segment_edges_gat;   /* most GATConv edges (self loops included) inside one segment */
  int32_t max_segment_edges_mean;  /* most SimpleConv edges inside one segment                    */
  int32_t flags;            /* GATRES_GRAPH_* bits from gatres_graph_flags_host (0 = nothing known)        */
  /* Row windows of split segments, from gatres_graph_windows_host: window[M - 2] = {rows, GATConv edges, SimpleConv
   * edges} for M = 2 .. 8 workgroups per segment.  All zero = unknown (the fused kernels then size their LDS tables by
   * the whole segment). */
  int32_t window[7][3];
  int32_t reserved2;
  /* Node relabelling of the plan (gatres_graph_reorder_host): every array above is built on the RELABELLED graph and
   * perm[new id] = the caller's node id.  Only the arrays the caller hands over or gets back -- x, y, mask, out, g_out,
   * g_x -- are indexed through it, inside lin0 / lin1 / the loss, so the nn.Module contract is unchanged.  NULL = identity.
   * A relabelling permutes nodes, never a row's edge list: inside a destination row the edges keep PyG's order, so every
   * fp32 sum associates exactly as without it. */
  const int32_t* perm;
  /* Split segments: the most edges one part has to a row of ANOTHER part (in- or out-edges of the GATConv graph, whichever
   * is more), halo[M - 2] for M = 2 .. 8 workgroups per segment; from gatres_graph_windows_host.  The window kernel keeps
   * its halo lists (one entry per such edge) in LDS and is only taken when they fit. */
  int32_t halo[7];
  int32_t reserved3;
  /* Part tables of split segments (gatres_graph_part_tables_host), for part_tables_m workgroups per segment: one record of
   * part_tables_stride int32 words per (segment, part), device memory.  NULL, or an m other than the one a launch uses: the
   * window kernel derives the same tables from the CSR arrays in its prologues (correct, ~15 us slower per launch). */
  const int32_t* part_tables;
  int32_t part_tables_m;
  int32_t part_tables_stride;
} gatres_graph_t;

/* Header words of one part-table record (the tables follow; offsets are in int32 words from the record's start) */
enum {
  GATRES_PT_MAGIC = 0, GATRES_PT_M, GATRES_PT_N0, GATRES_PT_N, GATRES_PT_E0, GATRES_PT_EM0, GATRES_PT_T0, GATRES_PT_MT0,
  GATRES_PT_LO, GATRES_PT_HI, GATRES_PT_WLO, GATRES_PT_WHI, GATRES_PT_ELO, GATRES_PT_OEG, GATRES_PT_EWLO, GATRES_PT_WEG,
  GATRES_PT_MELO, GATRES_PT_OEM, GATRES_PT_TLO, GATRES_PT_OTG, GATRES_PT_MTLO, GATRES_PT_OTM,
  GATRES_PT_F_IMG, GATRES_PT_F_IMG_WORDS, GATRES_PT_F_HLIST, GATRES_PT_F_HCNT, GATRES_PT_F_ELIST, GATRES_PT_F_ECNT,
  GATRES_PT_B_IMG, GATRES_PT_B_IMG_WORDS, GATRES_PT_B_HROW, GATRES_PT_B_HRCNT, GATRES_PT_B_HEDGE, GATRES_PT_B_HECNT,
  GATRES_PT_B_EROW, GATRES_PT_B_ERCNT, GATRES_PT_B_EEDGE, GATRES_PT_B_EECNT,
  GATRES_PT_HEADER = 48
};
#define GATRES_PT_MAGIC_VALUE 0x47545031

/* gatres_graph_t.flags */
#define GATRES_GRAPH_SYMMETRIC 1   /* every edge (u, v), u != v, has its reverse (v, u) in edge_index -- what
                                    * pgu.from_networkx gives for an undirected water network (utils/DataLoader.py:29).  The parts
                                    * of a split segment then owe each other halo rows in BOTH directions at every hand-off,
                                    * which paces them without the heartbeat granules (k_fused_dev.h: xch_heartbeat). */

#define GATRES_GRAPH_DEG_LE32 2    /* no node has more than 31 edges into it or out of it in edge_index: every row of the
                                    * plan's CSRs (GATConv's with its self loop, SimpleConv's, their transposes) has at most
                                    * 32 entries.  The blocked kernels (gatres_bf16_*_proj / _src_dx) walk a row with its own
                                    * lane group and are only taken for such plans; others keep the per-op kernels, whose hub
                                    * rows are reduced by whole waves. */

#define GATRES_GRAPH_DEG_LE6 4     /* no node has more than 5 edges into it or out of it in edge_index: every row of the plan's
                                    * CSRs has at most 6 entries (the per-snapshot kernels' slot width, MAXD).  The window kernel
                                    * then runs an instantiation without the edge-at-a-time paths of its stages (k_window.hip). */

/* Host-side plan builder (runs on the CPU, once per topology).  edge_index_host: int64 [2, E] row-major as
 * torch stores it.  Step 1: count -> E'.  Step 2: fill caller-allocated HOST arrays sized from that count. */
int gatres_graph_count_host(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes,
                            int64_t* num_edges_gat_out);
int gatres_graph_build_host(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes,
                            int32_t* rowptr, int32_t* col, int32_t* t_rowptr, int32_t* t_eid, int32_t* t_dst,
                            int32_t* m_rowptr, int32_t* m_col, int32_t* mt_rowptr, int32_t* mt_dst);

/* Properties of the edge list the kernels may rely on (host): *flags_out = OR of GATRES_GRAPH_* bits. */
int gatres_graph_flags_host(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes, int32_t* flags_out);

/* Part tables (host): see gatres_graph_t.part_tables.  Inputs: the plan's HOST arrays (as gatres_graph_build_host filled
 * them) and its segment table; m = workgroups per segment (2 .. 8, gatres_fused_cus_per_segment).  Call once with
 * out = NULL to learn the record size (*stride_words_out, a multiple of 4), then with an array of
 * num_segments * m * stride words. */
int gatres_graph_part_tables_host(const int32_t* rowptr, const int32_t* col, const int32_t* t_rowptr, const int32_t* t_eid,
                                  const int32_t* t_dst, const int32_t* m_rowptr, const int32_t* m_col,
                                  const int32_t* mt_rowptr, const int32_t* mt_dst, const int32_t* seg_ptr,
                                  int32_t num_segments, int32_t m, int32_t* out, int64_t stride_words,
                                  int64_t* stride_words_out);

/* Segment table (host): the finest partition of [0, N) into contiguous ranges closed under the edges, adjacent
 * ranges merged while the result stays <= merge_upto nodes.  seg_ptr_host must hold N + 1 entries. */
int gatres_graph_segments_host(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes,
                               int32_t merge_upto, int32_t* seg_ptr_host, int32_t* num_segments_out,
                               int32_t* max_segment_nodes_out, int32_t* max_segment_edges_gat_out,
                               int32_t* max_segment_edges_mean_out);

/* Row windows of split segments (see gatres_graph_t.window): for M = 2 .. 8 workgroups per segment, the largest
 * contiguous row range a part needs (own rows + the rows adjacent to them), in rows / GATConv edges / SimpleConv edges,
 * followed by gatres_graph_t.halo: out28 = {rows2, gat2, mean2, ..., rows8, gat8, mean8, halo2, ..., halo8}. */
int gatres_graph_windows_host(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, const int32_t* seg_ptr,
                              int32_t num_segments, int32_t* out28);

/* Bandwidth-reducing relabelling (host): reverse Cuthill-McKee inside every segment (nodes never leave their segment, so
 * seg_ptr stays valid).  perm_new2old_host[new] = old, N entries.  The fused kernels' row windows (above) are what it
 * shrinks: real .inp junction orders are not locality-preserving (utils/DataLoader.py:28-37 keeps the file order). */
int gatres_graph_reorder_host(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes,
                              const int32_t* seg_ptr_host, int32_t num_segments, int32_t* perm_new2old_host);

/* LZ4 block decompression on the host (the Blosc-lz4 chunks of the reference's zarr stores, utils/DataLoader.py:212-242).
 * Returns the number of bytes written to dst, or GATRES_E_BADARG on malformed input / a dst that is too small. */
int64_t gatres_lz4_decompress_host(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap);

/* The library reads its GATRES_* tuning / fallback switches from the environment once, at first use, never on a launch
 * path.  A process that changes them afterwards calls this to have them read again.  Always returns 0. */
int gatres_knobs_reload(void);

/* 64-bit content hash of an int64 [2,E] DEVICE edge_index (for plan caching); hash_out: device uint64[1],
 * must be zeroed by the caller on the same stream before the call. */
int gatres_edge_index_hash(const int64_t* edge_index, int64_t num_edges, uint64_t* hash_out, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * Per-op kernels.  Each replaces the PyG op sequence named in its comment.
 * ------------------------------------------------------------------------------------------------------ */

/* lin0: PyG Linear(1, nc) (GraphModels.py:477,487).  out[n,c] = xm[n]*w[c] + b[c], where xm[n] = 0 if
 * mask != NULL and mask[n] != 0 (the caller-side `data.x[batch_mask] = 0`, train.py:174). */
int gatres_lin0_fwd(const float* x, const uint8_t* mask, const float* w, const float* b, float* out,
                    int32_t num_nodes, int32_t nc, void* stream);

/* K1: GATConv's lin_src projection + attention logits (GATConv.forward:
 *   h = lin(x).view(N,H,C); a_src = (h*att_src).sum(-1); a_dst = (h*att_dst).sum(-1)).
 * fp32 MFMA (v_mfma_f32_16x16x4_f32).  W is [H*C, K] row-major (PyG Linear.weight). */
int gatres_proj_attn_fwd(const float* x, const float* W, const float* att_src, const float* att_dst,
                         float* h, float* a_src, float* a_dst,
                         int32_t num_nodes, int32_t K, int32_t H, int32_t C, void* stream);

/* K2: GATConv edge scoring + per-destination softmax + weighted neighbour sum + bias (+ReLU)
 * (GATConv.edge_update / torch_geometric.utils.softmax / message / aggregate; heads concatenated;
 * heads=1, concat=False is the H=1 case).  alpha[E',H] is written for the backward pass. */
int gatres_gat_aggregate_fwd(const gatres_graph_t* g, const float* h, const float* a_src, const float* a_dst,
                             const float* bias, float* out, float* alpha,
                             int32_t H, int32_t C, int32_t apply_relu, void* stream);

/* K3: SimpleConv(aggr="mean")(y) + x0, then ReLU (GraphModels.py:466-467). */
int gatres_mean_residual_relu_fwd(const gatres_graph_t* g, const float* y, const float* x0, float* out,
                                  int32_t C, void* stream);

/* lin1: PyG Linear(nc, 1) (GraphModels.py:484,492).  out[n] = sum_c x[n,c]*w[c] + b[0]. */
int gatres_lin1_fwd(const float* x, const float* w, const float* b, float* out,
                    int32_t num_nodes, int32_t nc, void* stream);

/* ---- backward -------------------------------------------------------------------------------------- */

/* Gradient partial sums.  Every reduction over nodes is done by `num_slabs` waves, each writing its partial
 * into its own slab of a [num_slabs, slab_stride] fp32 buffer at the parameter's flat offset; a final
 * gatres_reduce_slabs sums the slabs in a fixed order (bitwise reproducible, no atomics). */
int gatres_reduce_slabs(const float* slabs, int32_t num_slabs, int64_t slab_stride, int64_t count,
                        float* out, void* stream);

/* lin1 backward.  g_x[n,c] = g_out[n]*w[c], masked by (x[n,c] > 0) when relu_mask != 0 (x is the last block's
 * post-ReLU output).  slab_w / slab_b: this parameter's position inside slab 0. */
int gatres_lin1_bwd(const float* g_out, const float* x, const float* w, float* g_x,
                    float* slab_w, float* slab_b, int32_t num_slabs, int64_t slab_stride,
                    int32_t num_nodes, int32_t nc, int32_t relu_mask, void* stream);

/* K3 backward.  g_pre is the gradient w.r.t. the block's pre-ReLU sum (already ReLU-masked by its producer);
 * g_y[j] = sum over out-edges (j->i) of g_pre[i] / max(indeg(i), 1). */
int gatres_mean_bwd(const gatres_graph_t* g, const float* g_pre, float* g_y, int32_t C, void* stream);

/* K2 backward, destination-major pass: softmax / LeakyReLU backward per in-edge.
 * g_e[E',H] (grad w.r.t. the raw score a_src[j]+a_dst[i]) and g_a_dst[N,H]. */
int gatres_gat_aggregate_bwd_dst(const gatres_graph_t* g, const float* g_out, const float* h,
                                 const float* alpha, const float* a_src, const float* a_dst,
                                 float* g_e, float* g_a_dst, int32_t H, int32_t C, void* stream);

/* K2 backward, source-major pass over CSR^T (atomics-free):
 * g_a_src[j,h] = sum_out-edges g_e;  g_h[j] = sum alpha_e*g_out[i] + g_a_src (x) att_src + g_a_dst (x) att_dst. */
int gatres_gat_aggregate_bwd_src(const gatres_graph_t* g, const float* g_out, const float* alpha,
                                 const float* g_e, const float* g_a_dst, const float* att_src,
                                 const float* att_dst, float* g_h, float* g_a_src,
                                 int32_t H, int32_t C, void* stream);

/* K1 backward, data gradient: g_x = g_h @ W (fp32 MFMA), W given TRANSPOSED as Wt[K, HC] row-major.
 * Optional fused epilogue: g_x = (g_x + resid) masked by (relu_ref > 0)  (resid / relu_ref may be NULL). */
int gatres_proj_bwd_dx(const float* g_h, const float* Wt, const float* resid, const float* relu_ref,
                       float* g_x, int32_t num_nodes, int32_t K, int32_t HC, void* stream);

/* K1 backward, weight gradient partials: slab[s][c*K+k] = sum over slab s's nodes of g_h[n,c]*x[n,k]. */
int gatres_proj_bwd_dw(const float* g_h, const float* x, float* slab_W, int32_t num_slabs,
                       int64_t slab_stride, int32_t num_nodes, int32_t K, int32_t HC, void* stream);

/* Column reductions of one GATConv: partials of g_att_src[h,c] = sum_n g_a_src[n,h]*h[n,h,c], g_att_dst likewise,
 * g_bias[hc] = sum_n g_out[n,hc]. */
int gatres_conv_param_grads(const float* h, const float* g_a_src, const float* g_a_dst, const float* g_out,
                            float* slab_att_src, float* slab_att_dst, float* slab_bias, int32_t num_slabs,
                            int64_t slab_stride, int32_t num_nodes, int32_t H, int32_t C, void* stream);

/* lin0 backward partials: g_w[c] = sum_n g[n,c]*xm[n], g_b[c] = sum_n g[n,c]. */
int gatres_lin0_bwd(const float* g, const float* x, const uint8_t* mask, float* slab_w, float* slab_b,
                    int32_t num_slabs, int64_t slab_stride, int32_t num_nodes, int32_t nc, void* stream);

/* Relabelled plans (gatres_graph_t.perm), per-op path: dst[i] = src[perm[i]] (scatter == 0) or dst[perm[i]] = src[i]. */
int gatres_permute_f32(const float* src, const int32_t* perm, float* dst, int32_t num_nodes, int32_t scatter,
                       void* stream);
int gatres_gather_u8(const uint8_t* src, const int32_t* perm, uint8_t* dst, int32_t num_nodes, void* stream);

/* out[c*rows + r] = in[r*cols + c] for every GATConv weight of the flat parameter vector (see layout below). */
int gatres_transpose_conv_weights(const float* params, float* wt, int32_t num_blocks, int32_t nc, void* stream);

/* ---- caller side of the step (train.py:174-188) ---------------------------------------------------- */

/* Per-graph random node mask with EXACTLY int(n_g*rate) ones per graph (utils/auxil.py:143-182), generated
 * on the device.  node_ptr[B+1]: node offsets of the graphs; step_counter: device uint64[1] (read; combined
 * with seed so a replayed hipGraph draws a fresh mask each step when the caller bumps the counter). */
int gatres_mask_generate(const int32_t* node_ptr, int32_t num_graphs, double mask_rate, uint64_t seed,
                         const uint64_t* step_counter, uint8_t* mask, void* stream);

/* The same sampler AND the staging of a device-resident batch in ONE launch: x_dst[0..N) = x_src (and y_dst = y_src when
 * both are given; x_src may equal x_dst: nothing is copied then).  What `collate + mask` of train.py:159-173 amounts to
 * once snapshots live on the device: one small kernel and one kernel boundary less per step than a copy followed by
 * gatres_mask_generate. */
int gatres_stage_batch_mask(const float* x_src, const float* y_src, float* x_dst, float* y_dst, int64_t num_nodes,
                            const int32_t* node_ptr, int32_t num_graphs, double mask_rate, uint64_t seed,
                            const uint64_t* step_counter, uint8_t* mask, void* stream);

/* The same with the batch COLLATED on the way (utils/DataLoader.py:191-204 + the PyG DataLoader of train.py:302-303 for a
 * static topology): data is the device-resident snapshot matrix [S][nodes_per_graph], rows[num_graphs] (device, int64) the
 * snapshots of this batch; x_dst[g * nodes_per_graph + v] = data[rows[g]][v] (and y_dst likewise when given: targets are
 * the inputs before masking, utils/auxil.py:84-98). */
int gatres_stage_rows_mask(const float* data, const int64_t* rows, int32_t nodes_per_graph, float* x_dst, float* y_dst,
                           const int32_t* node_ptr, int32_t num_graphs, double mask_rate, uint64_t seed,
                           const uint64_t* step_counter, uint8_t* mask, void* stream);

/* MSELoss(mean) over masked nodes + its gradient: loss[0] = mean_{mask}(out-y)^2, g_out[n] = mask ? 2(out-y)/M : 0. */
int gatres_masked_mse(const float* out, const float* y, const uint8_t* mask, float* loss, float* g_out,
                      int32_t num_nodes, void* stream);

/* torch.optim.Adam step (weight_decay folded into the gradient), step counter on the device:
 * step_counter is device uint64[2] = {step, internal ticket (must start at 0)}; the kernel uses step+1 for the
 * bias corrections and then increments step.  grad_scale multiplies the gradient first (1/world_size). */
int gatres_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                     uint64_t* step_counter, int64_t count, double lr, double beta1, double beta2, double eps,
                     double weight_decay, float grad_scale, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * Typed per-op kernels: the per-op kernels above with the storage type of the activation-sized tensors as an argument
 * (`dtype`: GATRES_DTYPE_F32 or GATRES_DTYPE_BF16, defined with gatres_model_t below).  `void*` arguments are the tensors
 * whose element type `dtype` selects; every other pointer is fp32 / int32 exactly as in the fp32 entry point of the same
 * name, which is the GATRES_DTYPE_F32 instance.  bf16: loads widen to fp32, sums / softmax / accumulators stay fp32,
 * stores round to nearest-even; the projections run on v_mfma_f32_16x16x32_bf16 and need nc >= 32.
 * ------------------------------------------------------------------------------------------------------ */
/* k_aggregate.hip */
int gatres_t_gat_aggregate_fwd(const gatres_graph_t* g, const void* h, const float* a_src, const float* a_dst,
                               const float* bias, void* out, float* alpha, int H, int C, int apply_relu, int dtype,
                               void* stream);
int gatres_t_gat_aggregate_bwd_dst(const gatres_graph_t* g, const void* g_out, const void* h, const float* alpha,
                                   const float* a_src, const float* a_dst, float* g_e, float* g_a_dst, int H, int C,
                                   int dtype, void* stream);
int gatres_t_gat_aggregate_bwd_src(const gatres_graph_t* g, const void* g_out, const float* alpha, const float* g_e,
                                   const float* g_a_dst, const float* att_src, const float* att_dst, void* g_h,
                                   float* g_a_src, int H, int C, int dtype, void* stream);
int gatres_t_mean_residual_relu_fwd(const gatres_graph_t* g, const void* y, const void* x0, void* out, int C, int dtype,
                                    void* stream);
int gatres_t_mean_bwd(const gatres_graph_t* g, const void* g_pre, void* g_y, int C, int dtype, void* stream);

/* k_misc.hip */
int gatres_t_lin0_fwd(const float* x, const uint8_t* mask, const float* w, const float* b, void* out, int num_nodes, int nc,
                      int dtype, void* stream);
int gatres_t_lin0_bwd(const void* g, const float* x, const uint8_t* mask, float* slab_w, float* slab_b, int num_slabs,
                      int64_t slab_stride, int num_nodes, int nc, int dtype, void* stream);
int gatres_t_lin1_fwd(const void* x, const float* w, const float* b, float* out, int num_nodes, int nc, int dtype,
                      void* stream);
int gatres_t_lin1_bwd(const float* g_out, const void* x, const float* w, void* g_x, float* slab_w, float* slab_b,
                      int num_slabs, int64_t slab_stride, int num_nodes, int nc, int relu_mask, int dtype, void* stream);
int gatres_t_conv_param_grads(const void* h, const float* g_a_src, const float* g_a_dst, const void* g_out,
                              float* slab_att_src, float* slab_att_dst, float* slab_bias, int num_slabs,
                              int64_t slab_stride, int num_nodes, int H, int C, int dtype, void* stream);
/* A convolution's parameter-gradient partials as the per-op backward launches them (model_driver.hip): the weight
 * partials (gatres_t_proj_bwd_dw) and the attention-vector / bias column sums (gatres_t_conv_param_grads) as ONE launch where
 * a co-launching kernel applies, else one after the other.  w_slabs: slab rows the weight partials use (== num_slabs, or
 * fewer for the two-dimensional partials of wide bf16 models). */
int gatres_t_conv_partials(const void* g_h, const void* x, float* slab_W, int w_slabs, int64_t slab_stride, int num_nodes,
                           int K, int HC, const void* h, const float* g_a_src, const float* g_a_dst, const void* g_out,
                           float* slab_att_src, float* slab_att_dst, float* slab_bias, int num_slabs, int H, int C, int dtype,
                           void* stream);
/* bf16 copies of every GATConv weight and of its transpose: wb block layout [W1 : 2nc x nc][W2 : nc x 2nc] */
/* [W1^T : nc x 2nc][W2^T : 2nc x nc] (4 * 2nc^2 bf16 per block) */
int gatres_convert_conv_weights_bf16(const float* params, void* wb, int num_blocks, int nc, void* stream);

/* k_proj.hip.  W / Wt: fp32 for GATRES_DTYPE_F32, bf16 (from gatres_convert_conv_weights_bf16) for GATRES_DTYPE_BF16. */
int gatres_t_proj_attn_fwd(const void* x, const void* W, const float* att_src, const float* att_dst, void* h, float* a_src,
                           float* a_dst, int num_nodes, int K, int H, int C, int dtype, void* stream);
int gatres_t_proj_bwd_dx(const void* g_h, const void* Wt, const void* resid, const void* relu_ref, void* g_x, int num_nodes,
                         int K, int HC, int dtype, void* stream);
int gatres_t_proj_bwd_dw(const void* g_h, const void* x, float* slab_W, int num_slabs, int64_t slab_stride, int num_nodes,
                         int K, int HC, int dtype, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * Whole-network drivers: enqueue every kernel of GATResMeanConv.forward / its backward natively.
 *
 * Flat parameter layout (fp32, state_dict order; GraphModels.py:472-484, :455-460):
 *   lin0.weight[nc]  lin0.bias[nc]
 *   per block: conv1.att_src[2nc] conv1.att_dst[2nc] conv1.bias[2nc] conv1.lin.weight[2nc, nc]
 *              conv2.att_src[nc]  conv2.att_dst[nc]  conv2.bias[nc]  conv2.lin.weight[nc, 2nc]
 *   lin1.weight[nc]  lin1.bias[1]
 * ------------------------------------------------------------------------------------------------------ */
typedef struct gatres_model {
  int32_t num_blocks;
  int32_t nc;
  /* Storage type of the activation-sized tensors between kernels (saved activations, their gradients) and operand type of
   * the projections' MFMAs.  GATRES_DTYPE_F32: exact fp32 everywhere (the 1e-5 parity path).  GATRES_DTYPE_BF16 (BASELINE
   * config 3): bf16 in HBM, v_mfma_f32_16x16x32_bf16 for lin(x) and its data gradient with fp32 accumulation; attention
   * logits, softmax, every neighbour sum, parameter gradients, master weights and Adam stay fp32.  Per-op kernels only
   * (the fused per-snapshot path is fp32).  Buffers keep their fp32 SIZES (gatres_saved_floats / gatres_scratch_floats):
   * a bf16 tensor occupies the first half of its fp32 slot.  x, y, out, g_out, g_x, params, grads are always fp32. */
  int32_t act_dtype;
  int32_t flags;            /* GATRES_MODEL_* bits (0 = none) */
} gatres_model_t;

/* gatres_model_t.flags */
#define GATRES_MODEL_INFERENCE 1   /* forward-only calls whose `saved` buffer nobody will read (evaluation): the per-snapshot
                                    * kernel still needs the buffer (pass gatres_saved_floats floats) but may leave it unwritten --
                                    * the window kernel then runs an instantiation without the saved-activation stores */

/* --------------------------------------------------------------------------------------------------------
 * Blocked kernels (round 5; k_blocked.hip): a SPARSE stage and the dense projection that consumes its output
 * as ONE launch -- bf16 storage, nc = 128 (gatres_large: ConfigModels.py:22-32; block GraphModels.py:462-468),
 * plans flagged GATRES_GRAPH_DEG_LE32.  A 512-thread workgroup aggregates a block of 32 rows (the arithmetic
 * of the gatres_t_* sparse kernels, statement for statement), stores them (later launches read them) and runs
 * them through the matrix cores from LDS against W fragments held in registers.  Each call gives what the two
 * per-op calls it replaces give (the next convolution's attention logits are summed in another order).
 *   gatres_blocked_supported   1 when gatres_model_forward/backward_per_op take these kernels for (m, g)
 *   gatres_bf16_agg_proj_fwd   = gatres_t_gat_aggregate_fwd(H = 2, ReLU) + gatres_t_proj_attn_fwd(K = 2nc, H = 1)
 *   gatres_bf16_mean_proj_fwd  = gatres_t_mean_residual_relu_fwd + gatres_t_proj_attn_fwd(K = nc, H = 2)
 *   gatres_bf16_src_dx_bwd     = gatres_t_gat_aggregate_bwd_src(H) + gatres_t_proj_bwd_dx
 * W_next / Wt: bf16 [M, K] row-major as gatres_convert_conv_weights_bf16 leaves them. */
int gatres_blocked_supported(const gatres_model_t* m, const gatres_graph_t* g);
int gatres_bf16_agg_proj_fwd(const gatres_graph_t* g, const void* h, const float* a_src, const float* a_dst,
                             const float* bias, void* o, float* alpha, const void* W_next, const float* att_src_next,
                             const float* att_dst_next, void* h_next, float* a_src_next, float* a_dst_next, int32_t nc,
                             void* stream);
int gatres_bf16_mean_proj_fwd(const gatres_graph_t* g, const void* y, const void* x0, void* x_next, const void* W_next,
                              const float* att_src_next, const float* att_dst_next, void* h_next, float* a_src_next,
                              float* a_dst_next, int32_t nc, void* stream);
int gatres_bf16_src_dx_bwd(const gatres_graph_t* g, const void* g_out, const float* alpha, const float* g_e,
                           const float* g_a_dst, const float* att_src, const float* att_dst, void* g_h, float* g_a_src,
                           const void* Wt, const void* resid, const void* relu_ref, void* g_x, int32_t H, int32_t nc,
                           void* stream);

int64_t gatres_param_count(int32_t num_blocks, int32_t nc);
/* floats of forward state kept for backward / of scratch needed by forward+backward / slabs used. */
int64_t gatres_saved_floats(const gatres_model_t* m, const gatres_graph_t* g);
int64_t gatres_scratch_floats(const gatres_model_t* m, const gatres_graph_t* g);
int32_t gatres_num_slabs(const gatres_model_t* m, int32_t num_nodes);

/* out[N] = GATResMeanConv(x).  saved == NULL: inference (activations are not kept; scratch is reused per block). */
int gatres_model_forward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                         const float* x, const uint8_t* mask, float* out, float* saved, float* scratch,
                         void* stream);

/* grads[P] = d loss / d params given g_out[N] = d loss / d out; g_x (may be NULL) = d loss / d x. */
int gatres_model_backward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                          const float* x, const uint8_t* mask, const float* g_out, const float* saved,
                          float* scratch, float* grads, float* g_x, void* stream);

/* The same two entry points restricted to the per-op kernels above (any graph, any segment size). */
int gatres_model_forward_per_op(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                const float* x, const uint8_t* mask, float* out, float* saved, float* scratch,
                                void* stream);
int gatres_model_backward_per_op(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                 const float* x, const uint8_t* mask, const float* g_out, const float* saved,
                                 float* scratch, float* grads, float* g_x, void* stream);

/* One piece of gatres_model_backward_per_op: [lin1 backward] + blocks b_hi-1 .. b_lo + [lin0 backward].  FIRST must
 * come with b_hi == num_blocks, LAST with b_lo == 0; pieces are issued in that order on one stream.  With REDUCE the
 * piece ends by summing its slab partials into grads[lo, hi), lo = LAST ? 0 : 2nc + b_lo*(9nc + 4nc^2),
 * hi = FIRST ? P : 2nc + b_hi*(9nc + 4nc^2) -- final values, so a data-parallel caller can start the all-reduce of that
 * range while the next piece runs (gradient buckets in reverse block order).  Without REDUCE nothing is summed and the
 * slab partials stay in scratch in a layout private to the engine (wide bf16 models keep fewer rows for the weight
 * matrices than for the other parameters): gatres_model_reduce_grads sums them after the LAST piece, as
 * gatres_model_backward_per_op does. */
#define GATRES_PART_FIRST  1
#define GATRES_PART_LAST   2
#define GATRES_PART_REDUCE 4
int gatres_model_backward_per_op_part(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                      const float* x, const uint8_t* mask, const float* g_out, const float* saved,
                                      float* scratch, float* grads, float* g_x, int32_t b_hi, int32_t b_lo,
                                      int32_t flags, void* stream);

/* grads[0, P) = the sum of the per-op backward's slab partials in scratch (fixed order, bitwise reproducible). */
int gatres_model_reduce_grads(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * Fused per-snapshot path (k_fused.hip): ONE launch carries every segment through the selected phases --
 * GATRES_PHASE_FORWARD (lin0 .. lin1), GATRES_PHASE_LOSS (masked MSE + d loss/d out; needs mask, y) and
 * GATRES_PHASE_BACKWARD (all gradients, as per-segment slabs in scratch) -- one workgroup per segment, neighbour
 * gathers of the forward pass served from LDS when a segment fits (C-Town at nc=32 does).  The backward phase
 * leaves the GATConv weight / attention-vector gradients to gatres_fused_param_grads (they are off the backward's
 * dependency chain: one workgroup per (segment, block, conv) over the kept g_h tables in scratch), which must run
 * between gatres_fused_run(BACKWARD) and gatres_fused_finish.  gatres_fused_finish
 * sums the slabs into grads[P] and optionally applies Adam and finalises the loss in the same pass.
 * gatres_model_forward/backward and gatres_train_step take this path whenever gatres_fused_supported().
 * A segment may be carried by up to 8 workgroups on 8 CUs (hand-offs through scratch; the split is chosen so
 * that the whole grid is resident; environment GATRES_FUSED_SPLIT=1..8 sets it).  The barrier epochs persist
 * in `scratch` from launch to launch: ZERO the scratch buffer once after allocating it and do not share it between
 * streams.  A partner that never arrives (e.g. the GPU is shared and the grid is not resident) turns out[first node
 * of the segment] / the lin1 bias gradient into NaN instead of hanging.
 * loss_part: device float[8 * num_segments + 1].  GATRES_PHASE_BACKWARD reads the transposed conv weights from scratch:
 * call gatres_fused_prepare_backward (one small launch) after the parameters last changed.
 * ------------------------------------------------------------------------------------------------------ */
#define GATRES_PHASE_LOSS 16
int gatres_fused_supported(const gatres_model_t* m, const gatres_graph_t* g);
/* Workgroups (CUs) that carry one segment in the fused launches (1, 2, 4 or 8); 0 if the fused path is unavailable. */
int gatres_fused_cus_per_segment(const gatres_model_t* m, const gatres_graph_t* g);
/* 1 if launches that are given a `saved` buffer take the window kernel (LDS tables sized by the parts' row windows,
 * gatres_graph_t.window).  It is faster than the whole-segment-table kernel even for inference, so a caller that only
 * wants predictions may pass a throw-away `saved` buffer of gatres_saved_floats() to gatres_model_forward. */
int gatres_fused_window_kernel(const gatres_model_t* m, const gatres_graph_t* g);
/* Zero the split-segment barrier state inside `scratch` (as in a freshly zeroed buffer).  Only needed after an aborted
 * launch or when the buffer comes from elsewhere; not inside a captured graph that also holds fused launches. */
int gatres_fused_reset_sync(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, void* stream);
/* Two split launches must not run concurrently on one device (each needs its whole grid resident).  gatres_fused_run
 * orders itself behind the previous split launch when the stream changes; a captured launch cannot do that, so call
 * this on `stream` before REPLAYING a hipGraph that contains fused launches. */
int gatres_fused_serialize(void* stream);
/* How does the current device place the workgroups of a launch?  Runs ONCE per device (a 1-KB allocation, one launch of one
 * workgroup per CU, a synchronisation of `stream`: NOT inside a stream capture -- it then returns what is cached, 0 if nothing)
 * and caches: 1 = workgroups whose ids are 8 apart run on the same XCD (round-robin dispatch: every full MI355X), 0 = not.
 * With 1 the parts of a split segment take their shared L2 for granted and the launch starts WITHOUT its first cross-CU
 * barrier (round 6: 7 us per launch; every part still records its XCD id and compares at the end of the launch -- a mismatch
 * faults the launch, which is then dropped and counted like any other fault).  A caller that never asks gets the barrier. */
int gatres_probe_xcd_dispatch(void* stream);
/* Float index inside `scratch` of the split-launch status words {uint32 fault (set by a launch whose partner never
 * arrived; cleared by gatres_fused_finish / the next forward launch's epilogue), uint32 faults seen so far, uint32
 * internal}; -1 if the fused path does not apply.  A faulted training step is dropped: loss = NaN, gradients = NaN,
 * no Adam update, no step count. */
int64_t gatres_fused_status_offset(const gatres_model_t* m, const gatres_graph_t* g);
/* Diagnostic only: per-stage wall-clock stamps (100 MHz) of segment 0 for the following fused launches. */
int gatres_fused_set_stamps(uint64_t* stamps, int32_t capacity);
int gatres_fused_prepare_backward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                  float* scratch, void* stream);
int gatres_fused_run(const gatres_model_t* m, const gatres_graph_t* g, const float* params, const float* x,
                     const uint8_t* mask, const float* y, float* out, float* g_out, float* loss_part, float* g_x,
                     float* saved, float* scratch, int32_t phases, void* stream);
int gatres_fused_param_grads(const gatres_model_t* m, const gatres_graph_t* g, const float* saved, float* scratch,
                             void* stream);
int gatres_fused_finish(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                        const float* loss_part, float* loss, int32_t do_adam, float* params, float* exp_avg,
                        float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1, double beta2,
                        double eps, double weight_decay, float grad_scale, void* stream);
/* The same with the hyper-parameters read from `hparams` (device double[5] {lr, beta1, beta2, eps, weight_decay}) when it
 * is not NULL. */
int gatres_fused_finish_hp(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                           const float* loss_part, float* loss, int32_t do_adam, float* params, float* exp_avg,
                           float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1, double beta2,
                           double eps, double weight_decay, const double* hparams, float grad_scale,
                           const int32_t* mask_node_ptr, int32_t mask_graphs, double mask_rate, uint64_t mask_seed,
                           uint8_t* mask_next, void* stream);
/* mask_* : reserved for gatres_train_step's own sequence (GATRES_FLAG_MASK_NEXT: extra workgroups of the update launch sample
 * the device mask of the NEXT step -- what gatres_mask_generate would produce when called after this update); through this
 * entry point pass NULL / 0 (a non-NULL mask_next is refused with GATRES_E_UNSUPPORTED: the sampling tail reads a snapshot
 * that only the train step's parameter-gradient launch writes). */
/* gatres_fused_param_grads + gatres_fused_finish as ONE launch, for blocks [block_lo, block_hi) of the model (the whole
 * model: 0, num_blocks): every (block, conv) column's last workgroup sums the column's slab rows, writes grads and (do_adam)
 * applies Adam to those parameters; lin0's gradient rides on block 0, lin1's on the last block, so the ranges
 * [0, p(block_hi)) ... tile the flat vector like gatres_model_backward_per_op_part's.  do_adam needs the whole model in one
 * call.  Returns GATRES_E_UNSUPPORTED where the two-launch form has to be used (nc other than 16 / 32, or parameter
 * gradients formed on consumer workgroups of the backward launch: batches that leave CUs free). */
/* 1 if gatres_fused_param_grads_finish applies to this model / plan (else it returns GATRES_E_UNSUPPORTED). */
int gatres_fused_finish_folds(const gatres_model_t* m, const gatres_graph_t* g);
int gatres_fused_param_grads_finish(const gatres_model_t* m, const gatres_graph_t* g, const float* saved, float* scratch,
                                    float* grads, const float* loss_part, float* loss, int32_t do_adam, float* params,
                                    float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
                                    double beta2, double eps, double weight_decay, const double* hparams,
                                    float grad_scale, int32_t block_lo, int32_t block_hi, void* stream);

/* --------------------------------------------------------------------------------------------------------
 * One reference training iteration (train.py:159-190) enqueued natively:
 *   PHASE_MASK     device mask sampler (utils/auxil.py:166-182) -- skipped when node_ptr == NULL (caller's mask)
 *   PHASE_FORWARD  x[mask] = 0 ; out = model(x) ; loss = MSE(out[mask], y[mask]) and d loss / d out
 *   PHASE_BACKWARD grads = d loss / d params
 *   PHASE_ADAM     torch.optim.Adam(lr, weight_decay) update, step counter on the device
 * `phases` selects which of them a call enqueues, so a data-parallel caller can run its gradient all-reduce
 * between BACKWARD and ADAM.  Everything is stream-ordered and allocation-free (hipGraph capturable).
 * ------------------------------------------------------------------------------------------------------ */
#define GATRES_PHASE_MASK     1
#define GATRES_PHASE_FORWARD  2
#define GATRES_PHASE_BACKWARD 4
#define GATRES_PHASE_ADAM     8
#define GATRES_PHASE_ALL      15

typedef struct gatres_train_step {
  gatres_model_t model;
  const gatres_graph_t* graph;     /* HOST pointer to the plan struct (its members are device pointers) */
  float* params;                   /* [P] flat parameters (layout above)                 */
  float* grads;                    /* [P]                                                */
  float* exp_avg;                  /* [P] Adam first moment                              */
  float* exp_avg_sq;               /* [P] Adam second moment                             */
  uint64_t* step_counter;          /* device uint64[2]: {optimizer step, ticket}         */
  const float* x;                  /* [N] node inputs (z-normed pressures)               */
  const float* y;                  /* [N] targets (== x before masking, train.py:162-163)*/
  uint8_t* mask;                   /* [N] 1 = masked node                                */
  const int32_t* node_ptr;         /* [B+1] graph node offsets, or NULL                  */
  int32_t num_graphs;
  int32_t phases;
  double mask_rate;
  uint64_t seed;
  float* out;                      /* [N] predictions                                    */
  float* g_out;                    /* [N] d loss / d out                                 */
  float* loss;                     /* [1]                                                */
  float* saved;                    /* gatres_saved_floats()                              */
  float* scratch;                  /* gatres_scratch_floats()                            */
  double lr, beta1, beta2, eps, weight_decay;
  float grad_scale;                /* multiplies grads inside Adam (1/world_size)        */
  int32_t flags;                   /* GATRES_FLAG_PER_OP: force the per-op kernels       */
  const double* hparams;           /* device double[5] {lr, beta1, beta2, eps, weight_decay}, or NULL.  When set, the update
                                    * kernels read the hyper-parameters from it INSTEAD of the five host values above, so a
                                    * captured step follows a learning-rate schedule (train.py:349-350,510) without being
                                    * captured again                                                                     */
  int32_t block_lo, block_hi;      /* GATRES_FLAG_GRADS_ONLY: the blocks whose parameter gradients this call forms       */
  uint8_t* mask_next;              /* GATRES_FLAG_MASK_NEXT: [N] where the update launch leaves the NEXT step's mask (another
                                    * buffer than `mask`, which keeps the mask this step used); NULL: no sampling ahead     */
} gatres_train_step_t;
#define GATRES_FLAG_PER_OP 1
/* Fused path, PHASE_BACKWARD: stop after the backward chain; the caller forms the parameter gradients itself, range by
 * range, with gatres_fused_param_grads_finish (the data-parallel step: a bucket's all-reduce starts between two ranges). */
#define GATRES_FLAG_GRADS_DEFERRED 4
/* Fused path, PHASE_BACKWARD: ONLY the parameter gradients of blocks [block_lo, block_hi) from the kept tables of a
 * GRADS_DEFERRED backward (gatres_fused_param_grads_finish, no Adam; the launch that holds the last block also writes the
 * loss).  Needs gatres_fused_finish_folds(). */
#define GATRES_FLAG_GRADS_ONLY 8
/* Fused path, PHASE_ADAM in the same call as PHASE_BACKWARD, node_ptr and mask_next given: the update launch also samples
 * the mask of the NEXT step into `mask_next` (gatres_fused_finish_hp), so the next call -- with that buffer as its `mask` --
 * may leave PHASE_MASK out. */
#define GATRES_FLAG_MASK_NEXT 16
/* The transposed conv weights in `scratch` already match `params`: true right after a fused PHASE_ADAM step on the
 * same scratch (its Adam pass rewrites them) as long as nobody else has touched the parameters; skips the transposes. */
#define GATRES_FLAG_WT_VALID 2

int gatres_train_step(const gatres_train_step_t* ts, void* stream);

const char* gatres_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GATRES_H */
