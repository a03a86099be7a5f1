// Flat parameter / saved-activation / scratch layouts shared by the per-op driver and the fused per-snapshot kernels.
#pragma once
#include "gatres_common.h"

static inline __host__ __device__ int64_t r4(int64_t v) { return (v + 3) & ~(int64_t)3; }
#define GATRES_UREC_WORDS 128      // window kernel: words of one workgroup's geometry record (forward half | backward half)

struct Layout {
  int nb, nc;
  int64_t N, Eg;
  // flat parameters
  int64_t p_lin0_w, p_lin0_b, p_block0, p_block_stride, p_lin1_w, p_lin1_b, P;
  int64_t c1_as, c1_ad, c1_b, c1_W, c2_as, c2_ad, c2_b, c2_W;   // inside a block
  // saved activations of one block (s_xin == 0 so that "xin of block nb" is the final activation)
  int64_t s_xin, s_h1, s_as1, s_ad1, s_al1, s_o1, s_h2, s_as2, s_ad2, s_al2, s_stride, saved_total;
  // scratch.  The *2 arrays are conv2's private copies for the fused kernels: there, workgroups are at different
  // stages at the same time, so conv1 ([row*2], [row*2nc]) and conv2 ([row], [row*nc]) must not share an array.
  int64_t sc_y2, sc_ev, sc_xa, sc_xb, sc_gpa, sc_gpb, sc_gy2, sc_ge, sc_gad, sc_gas, sc_gh, sc_go1, sc_wt,
      sc_ge2, sc_gad2, sc_gas2, sc_gh2, sc_slabs, sc_loss_part, scratch_total;
  // per-op path of a relabelled plan: plan-order copies of x, mask (bytes), out, g_out, g_x
  int64_t sc_px, sc_pmask, sc_pout, sc_pgout, sc_pgx;
  // bf16 mode (gatres_model_t.act_dtype): bf16 copies of every conv weight and its transpose, rewritten from the fp32
  // master parameters at the start of every forward / backward (k_misc.hip: convert_conv_weights_bf16_kernel)
  int64_t sc_wb;
  // window kernel, split segments: tagged 8-byte granules {value, epoch} through which the parts of a segment hand each
  // other the rows / edge values their neighbours need (k_fused.hip: xch_*), one region per segment
  int64_t sc_xch, xch_stride;        // in floats (a granule = 2 floats)
  int64_t sc_urec;                   // window kernel: GATRES_UREC_WORDS words per workgroup
  // fused path only: every block keeps its g_h / g_alpha tables until the deferred parameter-gradient launch
  // (k_fused.hip: param_grads_kernel) has consumed them.  Block-major, global node index.
  int64_t sc_keep, keep_stride, k_gh1, k_gh2, k_gas1, k_gad1, k_gas2, k_gad2;
  // fused path, segments split over several CUs: flag lines (8 x 32 words per segment, + one error word at the
  // end) and one partial slab row per (segment, part)
  int split_max;
  int64_t sc_flags, flag_words, ready_words, col_words, sc_part_slabs;    // (col_words: arrival counters of param_grads_finish_kernel)
  int num_slabs;      // node-range slabs of the per-op path
  int slab_rows;      // slabs allocated = max(num_slabs, segments)
  int64_t slab_stride;
};

// Node-range slabs of the per-op parameter-gradient kernels: one workgroup per slab, so their number is the
// parallelism of those kernels.  Bounded by 2 GB of slab storage and 1024 (gatres_large: 6.7 MB per slab -> 305 slabs; a 64-MB
// bound left it 16 workgroups, and dW + the column sums took 77 % of a 50k-node step).
static inline int num_slabs_for(int64_t P, int64_t N) {
  int64_t cap = (2048LL << 20) / (4 * (P > 0 ? P : 1));
  if (cap > 1024) cap = 1024;      // (256 left the dW / column-sum kernels of narrow models on 800k-row batches at a quarter of the chip)
  if (cap < 16) cap = 16;
  int64_t s = (N + 63) / 64;
  if (s > cap) s = cap;
  // whole rounds of the 256 CUs: 305 slabs (gatres_large) ran as one full round + a 49-workgroup tail that took as long
  if (s > 256) s = (s / 256) * 256;
  if (s < 1) s = 1;
  return (int)s;
}

// Most workgroups (CUs) one segment may be split over: at most 8, and such that every workgroup of the grid (segments
// rounded up to a multiple of 8, times the split) is resident at once on the 256 CUs of an MI355X -- the parts of a
// segment wait for each other, so none may be left undispatched.
// Batches of 49 .. 96 segments are carried in TWO ROUNDS of at most 48 segments (k_fused_host.hip: launch_fused), at the
// parts per segment that fit the chip for one round: 8 (49 .. 64 segments), 6 (65 .. 80), 5 (81 .. 96).  0: no rounds.
static inline int rounds_parts_for(int num_segments) {
  const int padded = ((num_segments + 7) / 8) * 8;
  if (padded <= 48 || padded > 96) return 0;
  const int per = ((padded / 2 + 7) / 8) * 8;
  const int m = 256 / per;
  return m > 8 ? 8 : m;
}
static inline int split_max_for(int num_segments) {
  const int padded = ((num_segments + 7) / 8) * 8;
  if (padded <= 0 || padded > 256) return 1;
  if (const int rp = rounds_parts_for(num_segments)) return rp;
  const int m = 256 / padded;
  return m > 8 ? 8 : (m < 1 ? 1 : m);
}

// fused_nodes: the plan's largest segment if the fused per-snapshot path can take it (fused_nodes_of), else 0
// Granule exchange regions of one segment (offsets in granules; mn / me: the plan's largest segment in rows / GATConv
// edges).  Row tables are indexed [local row][feature], edge tables [local edge][head]; one table per exchange point,
// so a cell is rewritten only six exchanges later.  hb: one heartbeat granule per (exchange point, part).
struct XchLayout {
  int64_t f1h, f1a, f2h, f2a, f3, b1, b2y, b2e, b3o, b3e, hb, total;
};
static inline __host__ __device__ XchLayout make_xch_layout(int nc, int64_t mn, int64_t me) {
  XchLayout X;
  int64_t o = 0;
  X.f1h = o; o += mn * 2 * nc;
  X.f1a = o; o += mn * 2;
  X.f2h = o; o += mn * nc;
  X.f2a = o; o += mn;
  X.f3 = o;  o += mn * nc;
  X.b1 = o;  o += mn * nc;
  X.b2y = o; o += mn * nc;
  X.b2e = o; o += me;
  X.b3o = o; o += mn * 2 * nc;
  X.b3e = o; o += me * 2;
  X.hb = o;  o += 6 * 8;
  X.total = (o + 15) & ~(int64_t)15;
  return X;
}

static inline bool make_layout(const gatres_model_t* m, int64_t N, int64_t Eg, int num_segments, int fused_nodes,
                               int fused_edges, Layout* L) {
  if (!m || m->num_blocks < 0 || N <= 0 || Eg < 0) return false;
  const int nc = m->nc, nb = m->num_blocks;
  if (nc < 4 || nc > 128 || !gatres_is_pow2(nc)) return false;
  L->nb = nb; L->nc = nc; L->N = N; L->Eg = Eg;
  const int64_t w = 2LL * nc * nc;
  L->p_lin0_w = 0; L->p_lin0_b = nc; L->p_block0 = 2LL * nc;
  L->c1_as = 0; L->c1_ad = 2LL * nc; L->c1_b = 4LL * nc; L->c1_W = 6LL * nc;
  L->c2_as = 6LL * nc + w; L->c2_ad = L->c2_as + nc; L->c2_b = L->c2_ad + nc; L->c2_W = L->c2_b + nc;
  L->p_block_stride = 9LL * nc + 2 * w;
  L->p_lin1_w = L->p_block0 + nb * L->p_block_stride;
  L->p_lin1_b = L->p_lin1_w + nc;
  L->P = L->p_lin1_b + 1;

  int64_t o = 0;
  L->s_xin = o; o += r4(N * nc);
  L->s_h1 = o;  o += r4(N * 2 * nc);
  L->s_as1 = o; o += r4(N * 2);
  L->s_ad1 = o; o += r4(N * 2);
  L->s_al1 = o; o += r4(Eg * 2);
  L->s_o1 = o;  o += r4(N * 2 * nc);
  L->s_h2 = o;  o += r4(N * nc);
  L->s_as2 = o; o += r4(N);
  L->s_ad2 = o; o += r4(N);
  L->s_al2 = o; o += r4(Eg);
  L->s_stride = o;
  L->saved_total = nb * L->s_stride + r4(N * nc);

  L->num_slabs = num_slabs_for(L->P, N);
  L->slab_stride = r4(L->P);
  o = 0;
  L->sc_y2 = o;  o += r4(N * nc);
  L->sc_ev = o;  o += L->s_stride;
  L->sc_xa = o;  o += r4(N * nc);
  L->sc_xb = o;  o += r4(N * nc);
  L->sc_gpa = o; o += r4(N * nc);
  L->sc_gpb = o; o += r4(N * nc);
  L->sc_gy2 = o; o += r4(N * nc);
  L->sc_ge = o;  o += r4(Eg * 2);
  L->sc_gad = o; o += r4(N * 2);
  L->sc_gas = o; o += r4(N * 2);
  L->sc_gh = o;  o += r4(N * 2 * nc);
  L->sc_go1 = o; o += r4(N * 2 * nc);
  L->sc_wt = o;  o += r4((int64_t)nb * 2 * w);
  L->sc_ge2 = o;  o += r4(Eg);
  L->sc_gad2 = o; o += r4(N);
  L->sc_gas2 = o; o += r4(N);
  L->sc_gh2 = o;  o += r4(N * nc);
  L->slab_rows = L->num_slabs > num_segments ? L->num_slabs : num_segments;
  L->sc_slabs = o; o += (int64_t)L->slab_rows * L->slab_stride;
  L->split_max = fused_nodes > 0 ? split_max_for(num_segments) : 1;
  {
    const int64_t lp = (int64_t)num_segments * L->split_max;
    L->sc_loss_part = o; o += r4((lp > L->slab_rows ? lp : L->slab_rows) + 1);
  }
  L->sc_keep = o;
  int64_t k = 0;
  L->k_gh1 = k;  k += r4(N * 2 * nc);
  L->k_gh2 = k;  k += r4(N * nc);
  L->k_gas1 = k; k += r4(N * 2);
  L->k_gad1 = k; k += r4(N * 2);
  L->k_gas2 = k; k += r4(N);
  L->k_gad2 = k; k += r4(N);
  L->keep_stride = k;
  if (fused_nodes > 0) o += (int64_t)nb * k;
  L->sc_flags = o;
  L->flag_words = r4((int64_t)((num_segments + 7) / 8) * 8 * 8 * 32 + 32);
  L->ready_words = r4((int64_t)num_segments * 4 * 32);            // consumer hand-off lines, behind the flags
  L->col_words = r4(2LL * nb + 8);                                 // one counter per (block, conv) column, behind the ready lines
  L->sc_part_slabs = o + L->flag_words + L->ready_words + L->col_words;
  if (fused_nodes > 0) o += L->flag_words + L->ready_words + L->col_words;
  if (L->split_max > 1) o += (int64_t)num_segments * L->split_max * L->slab_stride;
  L->sc_xch = o;
  L->xch_stride = 0;
  if (L->split_max > 1 && nc <= 32) {       // (the window kernel only exists for nc <= 32)
    const XchLayout X = make_xch_layout(nc, fused_nodes, fused_edges);
    L->xch_stride = 2 * X.total;
    o += (int64_t)num_segments * L->xch_stride;
  }
  // window kernel: one record of per-part geometry (LDS table offsets, counts) per workgroup, written by the workgroup's
  // prologue and re-read per stage through the constant cache (k_window.hip: FwdRec / BwdRec)
  L->sc_urec = o;
  if (L->xch_stride > 0) o += (int64_t)((num_segments + 7) / 8) * 8 * 8 * GATRES_UREC_WORDS;
  L->sc_wb = o;    o += r4((int64_t)nb * 4 * nc * nc);        // nb * 4 matrices * 2nc^2 bf16 = nb * 4 nc^2 floats
  L->sc_px = o;    o += r4(N);
  L->sc_pmask = o; o += r4((N + 3) / 4);
  L->sc_pout = o;  o += r4(N);
  L->sc_pgout = o; o += r4(N);
  L->sc_pgx = o;   o += r4(N);
  L->scratch_total = o;
  return true;
}

static inline int fused_nodes_of(const gatres_graph_t* g) {
  return (g && g->num_segments > 0 && g->seg_ptr && g->max_segment_nodes > 0 && g->max_segment_nodes <= 4096)
             ? g->max_segment_nodes : 0;
}
static inline bool make_layout_g(const gatres_model_t* m, const gatres_graph_t* g, Layout* L) {
  return g && make_layout(m, g->num_nodes, g->num_edges_gat, g->num_segments, fused_nodes_of(g),
                          g->max_segment_edges_gat, L);
}



// Saved activations of the FUSED path: segment-major.  Every segment (snapshot) owns one contiguous slot of
// `total` floats laid out [block 0 | block 1 | ... | final x], each block holding the same ten tables as the
// per-op layout but indexed by LOCAL node / edge ids.  A workgroup then walks ~5 MB of contiguous memory instead of
// ten arrays per block spread over 150 MB (which cost a TLB miss on almost every access).  Slots are sized for the
// largest segment of the plan.
struct SegLayout {
  int64_t xin, h1, as1, ad1, al1, o1, h2, as2, ad2, al2, bstride, total;
};

static inline __host__ __device__ SegLayout make_seg_layout(int nb, int nc, int64_t mn, int64_t me) {
  SegLayout S;
  int64_t o = 0;
  S.xin = o; o += r4(mn * nc);
  S.h1 = o;  o += r4(mn * 2 * nc);
  S.as1 = o; o += r4(mn * 2);
  S.ad1 = o; o += r4(mn * 2);
  S.al1 = o; o += r4(me * 2);
  S.o1 = o;  o += r4(mn * 2 * nc);
  S.h2 = o;  o += r4(mn * nc);
  S.as2 = o; o += r4(mn);
  S.ad2 = o; o += r4(mn);
  S.al2 = o; o += r4(me);
  S.bstride = o;
  S.total = (int64_t)nb * o + r4(mn * nc);
  S.total = (S.total + 63) & ~(int64_t)63;          // 256-B aligned slots
  return S;
}
