// Whole-network drivers: enqueue GATResMeanConv.forward (GraphModels.py:486-494) and its backward as a fixed
// sequence of kernel launches on one stream, natively, so a training step costs two host calls (or one hipGraph
// replay) instead of ~2400 framework dispatches.  No allocation, no synchronisation: graph-capture safe.
//
// Block b (GResBlockMeanConv.forward, GraphModels.py:462-468):
//   xin -> K1(conv1) h1,a1 -> K2(conv1)+bias+ReLU out1 -> K1(conv2) h2,a2 -> K2(conv2)+bias y2
//       -> K3 mean(y2)+xin, ReLU -> xin of block b+1
#include <mutex>
#include "gatres_common.h"
#include "gatres_layout.h"

namespace {

#define RC(call)            \
  do {                      \
    const int rc_ = (call); \
    if (rc_) return rc_;    \
  } while (0)

}  // namespace

extern "C" int64_t gatres_param_count(int32_t num_blocks, int32_t nc) {
  return 2LL * nc + (int64_t)num_blocks * (9LL * nc + 4LL * nc * nc) + nc + 1;
}

extern "C" int64_t gatres_saved_floats(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!g) return GATRES_E_BADARG;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  int64_t total = L.saved_total;
  if (gatres_fused_supported(m, g)) {       // the fused kernels keep a segment-major copy instead (gatres_layout.h)
    const SegLayout S = make_seg_layout(L.nb, L.nc, g->max_segment_nodes, g->max_segment_edges_gat);
    const int64_t fused_total = (int64_t)g->num_segments * S.total;
    if (fused_total > total) total = fused_total;
  }
  return total;
}

extern "C" int64_t gatres_scratch_floats(const gatres_model_t* m, const gatres_graph_t* g) {
  Layout L;
  if (!g) return GATRES_E_BADARG;
  return make_layout_g(m, g, &L) ? L.scratch_total
                                                                             : (int64_t)GATRES_E_UNSUPPORTED;
}

extern "C" int32_t gatres_num_slabs(const gatres_model_t* m, int32_t num_nodes) {
  Layout L;
  return make_layout(m, num_nodes, 0, 0, 0, 0, &L) ? L.num_slabs : GATRES_E_UNSUPPORTED;
}

// "gatres-gfx950 abi<N> build <id>": <id> is the hash of the sources the library was built from (_build.py passes it; the
// Python loader refuses a library whose id differs from the sources beside it)
#ifndef GATRES_BUILD_ID
#define GATRES_BUILD_ID "unknown"
#endif
#define GATRES_STR2(x) #x
#define GATRES_STR(x) GATRES_STR2(x)
extern "C" const char* gatres_version(void) { return "gatres-gfx950 abi" GATRES_STR(GATRES_ABI_VERSION) " build " GATRES_BUILD_ID; }

extern "C" int gatres_model_forward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                    const float* x, const uint8_t* mask, float* out, float* saved, float* scratch,
                                    void* stream) {
  if (!m || !g || !params || !x || !out || !scratch) return GATRES_E_BADARG;
  if (!gatres_aligned16(params) || !gatres_aligned16(saved) || !gatres_aligned16(scratch)) return GATRES_E_BADARG;
  if (gatres_fused_supported(m, g))
    return gatres_fused_run(m, g, params, x, mask, nullptr, out, nullptr, nullptr, nullptr, saved, scratch,
                            GATRES_PHASE_FORWARD, stream);
  return gatres_model_forward_per_op(m, g, params, x, mask, out, saved, scratch, stream);
}

extern "C" int gatres_model_forward_per_op(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                           const float* x, const uint8_t* mask, float* out, float* saved,
                                           float* scratch, void* stream) {
  if (!m || !g || !params || !x || !out || !scratch) return GATRES_E_BADARG;
  if (!gatres_aligned16(params) || !gatres_aligned16(saved) || !gatres_aligned16(scratch)) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  const int N = g->num_nodes, nc = L.nc, dt = m->act_dtype;
  if (dt != GATRES_DTYPE_F32 && (dt != GATRES_DTYPE_BF16 || nc < 32)) return GATRES_E_UNSUPPORTED;
  // Activation tensors keep their fp32-sized slots in either mode (a bf16 tensor fills the first half), so every offset
  // below is mode-independent; the typed launchers (gatres_t_*) reinterpret the pointers.
  float* y2 = scratch + L.sc_y2;
  float* xa = scratch + L.sc_xa;
  float* xb = scratch + L.sc_xb;
  float* xcur = saved ? saved + L.s_xin : xa;
  float* out_caller = out;
  if (g->perm) {          // relabelled plan: x / mask into plan order, predictions back into the caller's order at the end
    RC(gatres_permute_f32(x, g->perm, scratch + L.sc_px, N, 0, stream));
    x = scratch + L.sc_px;
    if (mask) {
      uint8_t* pm = reinterpret_cast<uint8_t*>(scratch + L.sc_pmask);
      RC(gatres_gather_u8(mask, g->perm, pm, N, stream));
      mask = pm;
    }
    out = scratch + L.sc_pout;
  }
  const int64_t w = 2LL * nc * nc;
  if (dt == GATRES_DTYPE_BF16) RC(gatres_convert_conv_weights_bf16(params, scratch + L.sc_wb, L.nb, nc, stream));
  RC(gatres_t_lin0_fwd(x, mask, params + L.p_lin0_w, params + L.p_lin0_b, xcur, N, nc, dt, stream));
  // Blocked launches (k_blocked.hip; bf16, nc = 128, training -- `saved` -- only): conv1's aggregation carries conv2's
  // projection, K3 carries the next block's conv1 projection: 3 launches per block instead of 5.
  const bool blocked = saved && gatres_blocked_supported(m, g);
  bool have_proj1 = false;            // (the previous block's K3 launch has projected this block's conv1 already)
  for (int b = 0; b < L.nb; ++b) {
    float* base = saved ? saved + (int64_t)b * L.s_stride : scratch + L.sc_ev;
    float* xnext = saved ? saved + (int64_t)(b + 1) * L.s_stride + L.s_xin : (xcur == xa ? xb : xa);
    const float* pb = params + L.p_block0 + (int64_t)b * L.p_block_stride;
    // bf16: W1 / W2 of block b are the first two of its four 2nc^2-element bf16 matrices (= w/2 floats each)
    const float* wbb = scratch + L.sc_wb + (int64_t)b * 2 * w;
    const void* W1 = dt == GATRES_DTYPE_BF16 ? (const void*)wbb : (const void*)(pb + L.c1_W);
    const void* W2 = dt == GATRES_DTYPE_BF16 ? (const void*)(wbb + w / 2) : (const void*)(pb + L.c2_W);
    if (!have_proj1)
      RC(gatres_t_proj_attn_fwd(xcur, W1, pb + L.c1_as, pb + L.c1_ad, base + L.s_h1, base + L.s_as1, base + L.s_ad1, N, nc,
                                2, nc, dt, stream));
    if (blocked) {
      RC(gatres_bf16_agg_proj_fwd(g, base + L.s_h1, base + L.s_as1, base + L.s_ad1, pb + L.c1_b, base + L.s_o1,
                                  base + L.s_al1, W2, pb + L.c2_as, pb + L.c2_ad, base + L.s_h2, base + L.s_as2,
                                  base + L.s_ad2, nc, stream));
    } else {
      RC(gatres_t_gat_aggregate_fwd(g, base + L.s_h1, base + L.s_as1, base + L.s_ad1, pb + L.c1_b, base + L.s_o1,
                                    base + L.s_al1, 2, nc, 1, dt, stream));
      RC(gatres_t_proj_attn_fwd(base + L.s_o1, W2, pb + L.c2_as, pb + L.c2_ad, base + L.s_h2, base + L.s_as2,
                                base + L.s_ad2, N, 2 * nc, 1, nc, dt, stream));
    }
    RC(gatres_t_gat_aggregate_fwd(g, base + L.s_h2, base + L.s_as2, base + L.s_ad2, pb + L.c2_b, y2, base + L.s_al2, 1,
                                  nc, 0, dt, stream));
    if (blocked && b + 1 < L.nb) {
      float* nbase = saved + (int64_t)(b + 1) * L.s_stride;
      const float* pn = pb + L.p_block_stride;
      RC(gatres_bf16_mean_proj_fwd(g, y2, xcur, xnext, wbb + 2 * w, pn + L.c1_as, pn + L.c1_ad, nbase + L.s_h1,
                                   nbase + L.s_as1, nbase + L.s_ad1, nc, stream));
      have_proj1 = true;
    } else {
      RC(gatres_t_mean_residual_relu_fwd(g, y2, xcur, xnext, nc, dt, stream));
      have_proj1 = false;
    }
    xcur = xnext;
  }
  RC(gatres_t_lin1_fwd(xcur, params + L.p_lin1_w, params + L.p_lin1_b, out, N, nc, dt, stream));
  if (g->perm) RC(gatres_permute_f32(out, g->perm, out_caller, N, 1, stream));
  return 0;
}

extern "C" int gatres_model_backward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                     const float* x, const uint8_t* mask, const float* g_out, const float* saved,
                                     float* scratch, float* grads, float* g_x, void* stream) {
  if (!m || !g || !params || !x || !g_out || !saved || !scratch || !grads) return GATRES_E_BADARG;
  if (!gatres_aligned16(params) || !gatres_aligned16(saved) || !gatres_aligned16(scratch)) return GATRES_E_BADARG;
  if (gatres_fused_supported(m, g)) {
    RC(gatres_fused_prepare_backward(m, g, params, scratch, stream));
    RC(gatres_fused_run(m, g, params, x, mask, nullptr, nullptr, const_cast<float*>(g_out), nullptr, g_x,
                        const_cast<float*>(saved), scratch, GATRES_PHASE_BACKWARD, stream));
    RC(gatres_fused_param_grads(m, g, saved, scratch, stream));
    return gatres_fused_finish(m, g, scratch, grads, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0,
                               0, 0, 1.f, stream);
  }
  return gatres_model_backward_per_op(m, g, params, x, mask, g_out, saved, scratch, grads, g_x, stream);
}

extern "C" __attribute__((visibility("hidden"))) int gatres_reduce_slabs_regions(const float* slabs, int32_t num_slabs,
                                                                                 int32_t w_slabs, int64_t slab_stride,
                                                                                 int64_t lo_abs, int64_t count, int32_t nc,
                                                                                 int32_t nb, float* out, void* stream);
// Slab rows the GATConv weight gradients occupy: the bf16 dW kernel of wide models forms two-dimensional partials (64 rows of
// 64 x 64 blocks instead of one whole [2nc, nc] matrix per workgroup): 4x less slab traffic in dW and in the final sum
static int dw_slab_rows(const gatres_model_t* m, const Layout& L) {
  if (m->act_dtype == GATRES_DTYPE_BF16 && L.nc == 128 && L.num_slabs > 64 && !gatres_knobs()->dw_1d) {
    {                                                              // (tuning experiments: 32 / 64 / 128 row groups)
      const int v = gatres_knobs()->dw_slab_rows;
      if (v >= 8 && v <= 128 && v % 8 == 0 && v < L.num_slabs) return v;
    }
    return 64;
  }
  return L.num_slabs;
}

extern "C" int gatres_model_backward_per_op(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                            const float* x, const uint8_t* mask, const float* g_out,
                                            const float* saved, float* scratch, float* grads, float* g_x,
                                            void* stream) {
  if (!m) return GATRES_E_BADARG;
  RC(gatres_model_backward_per_op_part(m, g, params, x, mask, g_out, saved, scratch, grads, g_x, m->num_blocks, 0,
                                       GATRES_PART_FIRST | GATRES_PART_LAST, stream));
  return gatres_model_reduce_grads(m, g, scratch, grads, stream);
}

extern "C" int gatres_model_reduce_grads(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                                         void* stream) {
  if (!m || !g || !scratch || !grads) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  return gatres_reduce_slabs_regions(scratch + L.sc_slabs, L.num_slabs, dw_slab_rows(m, L), L.slab_stride, 0, L.P, L.nc, L.nb,
                                     grads, stream);
}

extern "C" __attribute__((visibility("hidden"))) int gatres_proj_bwd_dw_with_conv_grads(
    const void* g_h, const void* x, float* slab_W, int w_slabs, int64_t slab_stride, int num_nodes, int K, int HC,
    const void* h, const float* g_a_src, const float* g_a_dst, const void* g_out, float* slab_att_src, float* slab_att_dst,
    float* slab_bias, int num_slabs, int H, int C, int dtype, void* stream);
// A convolution's two partial-sum launches (attention-vector / bias partials, weight partials): ONE launch (extra workgroups
// of the weight-gradient kernel run the column sums) where a co-launching kernel applies, else one after the other.
static int conv_partials(const void* gh, const void* xin, float* slab_W, int Sw, int64_t st, int N, int K, int HC,
                         const void* h, const float* gas, const float* gad, const void* gout, float* s_as, float* s_ad,
                         float* s_b, int S, int H, int C, int dt, void* stream) {
  {
    const int rc = gatres_proj_bwd_dw_with_conv_grads(gh, xin, slab_W, Sw, st, N, K, HC, h, gas, gad, gout, s_as, s_ad, s_b,
                                                      S, H, C, dt, stream);
    if (rc != GATRES_E_UNSUPPORTED) return rc;
  }
  RC(gatres_t_conv_param_grads(h, gas, gad, gout, s_as, s_ad, s_b, S, st, N, H, C, dt, stream));
  return gatres_t_proj_bwd_dw(gh, xin, slab_W, Sw, st, N, K, HC, dt, stream);
}

extern "C" int gatres_t_conv_partials(const void* g_h, const void* x, float* slab_W, int w_slabs, int64_t slab_stride,
                                      int num_nodes, int K, int HC, const void* h, const float* g_a_src, const float* g_a_dst,
                                      const void* g_out, float* slab_att_src, float* slab_att_dst, float* slab_bias,
                                      int num_slabs, int H, int C, int dtype, void* stream) {
  if (!g_h || !x || !slab_W || !h || !g_a_src || !g_a_dst || !g_out || !slab_att_src || !slab_att_dst || !slab_bias)
    return GATRES_E_BADARG;
  return conv_partials(g_h, x, slab_W, w_slabs, slab_stride, num_nodes, K, HC, h, g_a_src, g_a_dst, g_out, slab_att_src,
                       slab_att_dst, slab_bias, num_slabs, H, C, dtype, stream);
}

// One piece of the per-op backward: [lin1 backward] blocks b_hi-1 .. b_lo [lin0 backward].  With GATRES_PART_REDUCE the
// slab partials of exactly the parameters this piece finishes are summed into `grads` right away, so a data-parallel
// caller can start the all-reduce of that range while the next piece runs (gradient buckets in reverse block order).
extern "C" int gatres_model_backward_per_op_part(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                                 const float* x, const uint8_t* mask, const float* g_out,
                                                 const float* saved, float* scratch, float* grads, float* g_x,
                                                 int32_t b_hi, int32_t b_lo, int32_t flags, void* stream) {
  if (!m || !g || !params || !x || !g_out || !saved || !scratch || !grads) return GATRES_E_BADARG;
  if (!gatres_aligned16(params) || !gatres_aligned16(saved) || !gatres_aligned16(scratch)) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout_g(m, g, &L)) return GATRES_E_UNSUPPORTED;
  const bool first = flags & GATRES_PART_FIRST, last = flags & GATRES_PART_LAST;
  if (b_lo < 0 || b_hi > L.nb || b_lo > b_hi || (first && b_hi != L.nb) || (last && b_lo != 0)) return GATRES_E_BADARG;
  const int N = g->num_nodes, nc = L.nc, S = L.num_slabs, dt = m->act_dtype, Sw = dw_slab_rows(m, L);
  if (dt != GATRES_DTYPE_F32 && (dt != GATRES_DTYPE_BF16 || nc < 32)) return GATRES_E_UNSUPPORTED;
  const int64_t st = L.slab_stride, w = 2LL * nc * nc;
  // g_pre ping-pongs between two buffers, one swap per block: where it stands depends only on the blocks done so far
  const bool odd = ((L.nb - b_hi) & 1) != 0;
  float* gp_cur = scratch + (odd ? L.sc_gpb : L.sc_gpa);
  float* gp_nxt = scratch + (odd ? L.sc_gpa : L.sc_gpb);
  float* gy2 = scratch + L.sc_gy2;
  float* ge = scratch + L.sc_ge;
  float* gad = scratch + L.sc_gad;
  float* gas = scratch + L.sc_gas;
  float* gh = scratch + L.sc_gh;
  float* go1 = scratch + L.sc_go1;
  // conv2's own g_h / g_a_src / g_a_dst tables: the parameter-gradient launches of a convolution read them on the side
  // stream while the chain goes on into the next convolution (which writes the other set)
  float* gh2 = scratch + L.sc_gh2;
  float* gas2 = scratch + L.sc_gas2;
  float* gad2 = scratch + L.sc_gad2;
  float* wt = scratch + L.sc_wt;
  float* slabs = scratch + L.sc_slabs;

  float* g_x_caller = g_x;
  if (g->perm) {          // relabelled plan: caller-order vectors into plan order (see the forward driver)
    if (first) {
      RC(gatres_permute_f32(g_out, g->perm, scratch + L.sc_pgout, N, 0, stream));
      RC(gatres_permute_f32(x, g->perm, scratch + L.sc_px, N, 0, stream));
      if (mask) RC(gatres_gather_u8(mask, g->perm, reinterpret_cast<uint8_t*>(scratch + L.sc_pmask), N, stream));
    }
    g_out = scratch + L.sc_pgout;
    x = scratch + L.sc_px;
    if (mask) mask = reinterpret_cast<const uint8_t*>(scratch + L.sc_pmask);
    if (g_x) g_x = scratch + L.sc_pgx;
  }
  if (first) {
    if (dt == GATRES_DTYPE_BF16) RC(gatres_convert_conv_weights_bf16(params, scratch + L.sc_wb, L.nb, nc, stream));
    else                         RC(gatres_transpose_conv_weights(params, wt, L.nb, nc, stream));
    const float* xfinal = saved + (int64_t)L.nb * L.s_stride + L.s_xin;
    RC(gatres_t_lin1_bwd(g_out, xfinal, params + L.p_lin1_w, gp_cur, slabs + L.p_lin1_w, slabs + L.p_lin1_b, S, st, N, nc,
                         L.nb > 0 ? 1 : 0, dt, stream));
  }
  // Fork / join.  A convolution's partial sums (dW, attention vectors, bias) feed only the optimizer: they are launched on
  // the library's side stream behind an event of the chain, and the chain waits for them only where it is about to
  // overwrite what they read -- conv2's before the next block's K3 backward (g_y2, set 2), conv1's before the next block's
  // dX2 (g_out1, set 1) -- and at the end of the piece.  Inside a stream capture the events become graph edges, and an
  // edge between two queues costs microseconds: measured on gatres_large, C-Town, bs 128 (hipGraph replay): fp32 13.54 ->
  // 13.16 ms / step, bf16 5.37 -> 5.85 (its launches are a third as long, and the side launches take the CUs the chain's
  // next launch needs) -- so the fork is taken for fp32 at nc >= 128 only, unless GATRES_SIDE_STREAM says otherwise.
  const int want_side = gatres_knobs()->side_stream;
  gatres_side_t* side = (want_side == 1 || (want_side < 0 && dt == GATRES_DTYPE_F32 && nc >= 128)) ? gatres_side() : nullptr;
  hipStream_t main_st = gatres_stream(stream);
  const bool blocked = gatres_blocked_supported(m, g) != 0;
  std::unique_lock<std::mutex> side_lock;          // (two host threads enqueueing backward pieces must not interleave on the events)
  if (side) side_lock = std::unique_lock<std::mutex>(*static_cast<std::mutex*>(side->mu));
  // Join on EVERY exit: an error return between a fork and its join would leave the side stream forked -- inside
  // a stream capture that invalidates the capture with an unrelated error, in eager mode the next call could overwrite
  // gy2 / gh2 / go1 while the side stream still reads them.  open_*: forked, the done event not recorded yet.
  struct SideJoin {
    gatres_side_t* side;
    hipStream_t main_st;
    bool pend_a = false, pend_b = false, open_a = false, open_b = false;
    ~SideJoin() {
      if (!side) return;
      if (open_a) { (void)hipEventRecord(side->done_a, side->stream); pend_a = true; }
      if (open_b) { (void)hipEventRecord(side->done_b, side->stream); pend_b = true; }
      if (pend_a) (void)hipStreamWaitEvent(main_st, side->done_a, 0);
      if (pend_b) (void)hipStreamWaitEvent(main_st, side->done_b, 0);
    }
  } sj{side, main_st};
  bool &pend_a = sj.pend_a, &pend_b = sj.pend_b;
#define HIPRC(call_) do { if ((call_) != hipSuccess) return (int)hipGetLastError(); } while (0)
  for (int b = b_hi - 1; b >= b_lo; --b) {
    const float* base = saved + (int64_t)b * L.s_stride;
    const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
    const float* pb = params + po;
    float* sb = slabs + po;
    // W1^T [nc, 2nc] and W2^T [2nc, nc]: fp32 transposes, or the last two of the block's four bf16 matrices
    const float* wbb = scratch + L.sc_wb + (int64_t)b * 2 * w;
    const void* wt1 = dt == GATRES_DTYPE_BF16 ? (const void*)(wbb + w) : (const void*)(wt + (int64_t)b * 2 * w);
    const void* wt2 = dt == GATRES_DTYPE_BF16 ? (const void*)(wbb + w + w / 2) : (const void*)(wt + (int64_t)b * 2 * w + w);
    // K3 backward: gradient w.r.t. conv2's output
    if (pend_a) { HIPRC(hipStreamWaitEvent(main_st, side->done_a, 0)); pend_a = false; }
    RC(gatres_t_mean_bwd(g, gp_cur, gy2, nc, dt, stream));
    // conv2 (H = 1, C = nc, K = 2nc); its output has no ReLU
    RC(gatres_t_gat_aggregate_bwd_dst(g, gy2, base + L.s_h2, base + L.s_al2, base + L.s_as2, base + L.s_ad2, ge, gad2, 1,
                                      nc, dt, stream));
    // Blocked launches (k_blocked.hip): the source-major pass carries the input gradient -- the weight-gradient launch that
    // reads g_h then follows the pair instead of standing between them (not on the side stream: it now reads what the chain's
    // next launch has not overwritten yet either way, but the fork is an fp32 choice and these kernels are bf16).
    if (blocked) {
      if (pend_b) { HIPRC(hipStreamWaitEvent(main_st, side->done_b, 0)); pend_b = false; }
      RC(gatres_bf16_src_dx_bwd(g, gy2, base + L.s_al2, ge, gad2, pb + L.c2_as, pb + L.c2_ad, gh2, gas2, wt2, nullptr,
                                base + L.s_o1, go1, 1, nc, stream));
    } else
    RC(gatres_t_gat_aggregate_bwd_src(g, gy2, base + L.s_al2, ge, gad2, pb + L.c2_as, pb + L.c2_ad, gh2, gas2, 1, nc, dt,
                                      stream));
    void* pst = stream;
    if (side) {
      HIPRC(hipEventRecord(side->fork_a, main_st));
      HIPRC(hipStreamWaitEvent(side->stream, side->fork_a, 0));
      pst = side->stream;
      sj.open_a = true;
    }
    RC(conv_partials(gh2, base + L.s_o1, sb + L.c2_W, Sw, st, N, 2 * nc, nc, base + L.s_h2, gas2, gad2, gy2, sb + L.c2_as,
                     sb + L.c2_ad, sb + L.c2_b, S, 1, nc, dt, pst));
    if (side) { sj.open_a = false; HIPRC(hipEventRecord(side->done_a, side->stream)); pend_a = true; }
    if (!blocked) {
      if (pend_b) { HIPRC(hipStreamWaitEvent(main_st, side->done_b, 0)); pend_b = false; }
      RC(gatres_t_proj_bwd_dx(gh2, wt2, nullptr, base + L.s_o1, go1, N, 2 * nc, nc, dt, stream));   // ReLU mask of conv1
    }
    // conv1 (H = 2, C = nc, K = nc)
    RC(gatres_t_gat_aggregate_bwd_dst(g, go1, base + L.s_h1, base + L.s_al1, base + L.s_as1, base + L.s_ad1, ge, gad, 2,
                                      nc, dt, stream));
    // d/d xin = conv1 path + residual; masked by the previous block's ReLU (block 0's input is lin0, no ReLU)
    if (blocked)
      RC(gatres_bf16_src_dx_bwd(g, go1, base + L.s_al1, ge, gad, pb + L.c1_as, pb + L.c1_ad, gh, gas, wt1, gp_cur,
                                b > 0 ? base + L.s_xin : nullptr, gp_nxt, 2, nc, stream));
    else
    RC(gatres_t_gat_aggregate_bwd_src(g, go1, base + L.s_al1, ge, gad, pb + L.c1_as, pb + L.c1_ad, gh, gas, 2, nc, dt,
                                      stream));
    pst = stream;
    if (side) {
      HIPRC(hipEventRecord(side->fork_b, main_st));
      HIPRC(hipStreamWaitEvent(side->stream, side->fork_b, 0));
      pst = side->stream;
      sj.open_b = true;
    }
    RC(conv_partials(gh, base + L.s_xin, sb + L.c1_W, Sw, st, N, nc, 2 * nc, base + L.s_h1, gas, gad, go1, sb + L.c1_as,
                     sb + L.c1_ad, sb + L.c1_b, S, 2, nc, dt, pst));
    if (side) { sj.open_b = false; HIPRC(hipEventRecord(side->done_b, side->stream)); pend_b = true; }
    if (!blocked)
      RC(gatres_t_proj_bwd_dx(gh, wt1, gp_cur, b > 0 ? base + L.s_xin : nullptr, gp_nxt, N, nc, 2 * nc, dt, stream));
    float* t = gp_cur; gp_cur = gp_nxt; gp_nxt = t;
  }
  if (pend_a) { HIPRC(hipStreamWaitEvent(main_st, side->done_a, 0)); pend_a = false; }      // join: the piece's slabs are complete
  if (pend_b) { HIPRC(hipStreamWaitEvent(main_st, side->done_b, 0)); pend_b = false; }
#undef HIPRC
  if (last) {
    RC(gatres_t_lin0_bwd(gp_cur, x, mask, slabs + L.p_lin0_w, slabs + L.p_lin0_b, S, st, N, nc, dt, stream));
    if (g_x) {
      RC(gatres_t_lin1_fwd(gp_cur, params + L.p_lin0_w, nullptr, g_x, N, nc, dt, stream));
      if (g->perm) RC(gatres_permute_f32(g_x, g->perm, g_x_caller, N, 1, stream));
    }
  }
  if (flags & GATRES_PART_REDUCE) {
    const int64_t lo = last ? 0 : L.p_block0 + (int64_t)b_lo * L.p_block_stride;
    const int64_t hi = first ? L.P : L.p_block0 + (int64_t)b_hi * L.p_block_stride;
    if (hi > lo) RC(gatres_reduce_slabs_regions(slabs + lo, S, Sw, st, lo, hi - lo, nc, L.nb, grads + lo, stream));
  }
  return 0;
}
