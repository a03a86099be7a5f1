// Whole-segment-table form of the fused per-snapshot kernel (segments whose parts have no compact row window, one
// workgroup per segment, or forward-only launches).  Device code and commentary: k_fused_dev.h.
#include "k_fused_dev.h"

namespace {

// ------------------------------------------------------------------------------------------ the kernel
template <int NC, int THREADS, bool CACHE>
__global__ __launch_bounds__(THREADS) void gatres_fused_kernel(const FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds_raw);
  const Layout& L = a.L;
  // workgroup id -> (segment, part): ids of one segment are 8 apart (same XCD)
  const int M = a.M;
  {
    const int F = ((a.seg_cnt + 7) / 8) * 8 * M;       // per-snapshot workgroups come first, consumers after
    if ((int)blockIdx.x >= F) {
      if constexpr (THREADS == 1024) consumer_main<NC, THREADS>(a, (int)blockIdx.x - F, ldsf);
      return;
    }
  }
  const int within = blockIdx.x % (8 * M);
  const int seg = a.seg0 + (blockIdx.x / (8 * M)) * 8 + (within & 7), part = within >> 3;
  if (seg >= a.seg0 + a.seg_cnt) return;
  const bool split = M > 1;
  const int n0 = a.seg_ptr[seg], n = a.seg_ptr[seg + 1] - n0;
  const int e0 = a.rowptr[n0], eg = a.rowptr[n0 + n] - e0;            // GATConv edges of this segment
  const int em0 = a.m_rowptr[n0], em = a.m_rowptr[n0 + n] - em0;      // SimpleConv edges
  const int tid = threadIdx.x;
  Rows rw;                                                            // own rows: whole 16-row tiles
  {
    const int tiles = (n + 15) >> 4;
    rw.lo = 16 * (int)((long long)tiles * part / M);
    rw.hi = min(n, 16 * (int)((long long)tiles * (part + 1) / M));
  }
  Group grp;
  grp.flags = a.flags + (size_t)seg * 8 * FLAG_STRIDE; grp.M = M; grp.part = part; grp.err = a.err;
  group_init<THREADS>(grp);
  if (a.safe_sync) grp.local = false;
  // rows per lane group per trip: 16 waves x 128 VGPRs cannot hold more than this without spilling; 8 waves x 256 can
  constexpr int UF = THREADS <= 512 ? 4 : 2;     // forward gathers
  constexpr int UB = THREADS <= 512 ? 2 : 1;     // backward sparse stages
  const float* P = a.params;
  float* sc = a.scratch;
  [[maybe_unused]] int stamp_i = 0;
  STAMP();
  if (STAMPS_PTR && blockIdx.x == 0 && threadIdx.x == 0) STAMPS_PTR[a.stamp_cap] = clock64();   // shader-clock probe

  if (a.phases & GATRES_PHASE_FORWARD) {
    // LDS map: [hA n x 2NC | hB n x NC | sa 2n | sd 2n] (CACHE) then the 16-bit topology
    float* hA = ldsf;
    float* hB = hA + (size_t)n * 2 * NC;
    float* sa = hB + (size_t)n * NC;
    float* sd = sa + (size_t)n * 2;
    u16* tp = reinterpret_cast<u16*>(CACHE ? (sd + (size_t)n * 2) : ldsf);
    u16* rp = tp;              tp += even(n + 1);
    u16* col = tp;             tp += even(eg);
    u16* mrp = tp;             tp += even(n + 1);
    u16* mcol = tp;
    tp += even(em);
    // LDS staging for a projection's W + att (seg_proj): with the tables cached, proj1 borrows the h2 region and
    // proj2 the h1 region (each is dead exactly then); otherwise a slot behind the topology
    constexpr int WL_FLOATS = 2 * NC * (2 * NC + 4) + 4 * NC;           // max over the block's projections
    constexpr bool WLDS = WL_FLOATS * 4 <= 40960;
    // The slot is always a real LDS address chosen by OFFSET (a nullable LDS pointer makes the compiler build flat
    // addresses with null checks, hoist them out of the block loop and spill them).  If the cached tables leave
    // room, a private slot at the end of LDS; otherwise proj1 borrows the h2 table and proj2 the h1 table.
    const int used_b = (int)(reinterpret_cast<unsigned char*>(tp) - lds_raw);
    const bool priv = used_b + WL_FLOATS * 4 + 16 <= LDS_BYTES;
    const int slot_b = LDS_BYTES - ((WL_FLOATS * 4 + 15) & ~15);
    float* wl1 = reinterpret_cast<float*>(lds_raw + (priv ? slot_b : (int)((unsigned char*)hB - lds_raw)));
    float* wl2 = reinterpret_cast<float*>(lds_raw + (priv ? slot_b : 0));
    copy_rowptr16<THREADS>(rp, a.rowptr, n0, n, e0);
    copy_idx16<THREADS>(col, a.col, e0, eg, n0);
    copy_rowptr16<THREADS>(mrp, a.m_rowptr, n0, n, em0);
    copy_idx16<THREADS>(mcol, a.m_col, em0, em, n0);

    // saved tables: this segment's contiguous slot with LOCAL indices (training), or the shared evaluation block of
    // the scratch area with global indices (inference)
    const SegLayout& SL = a.SL;
    float* segbase = a.saved ? a.saved + (int64_t)seg * SL.total : nullptr;
    const int nbS = a.saved ? 0 : n0, ebS = a.saved ? 0 : e0;
    const int64_t o_h1 = a.saved ? SL.h1 : L.s_h1, o_as1 = a.saved ? SL.as1 : L.s_as1, o_ad1 = a.saved ? SL.ad1 : L.s_ad1,
                  o_al1 = a.saved ? SL.al1 : L.s_al1, o_o1 = a.saved ? SL.o1 : L.s_o1, o_h2 = a.saved ? SL.h2 : L.s_h2,
                  o_as2 = a.saved ? SL.as2 : L.s_as2, o_ad2 = a.saved ? SL.ad2 : L.s_ad2,
                  o_al2 = a.saved ? SL.al2 : L.s_al2;
    float* xa = sc + L.sc_xa;
    float* xb = sc + L.sc_xb;
    float* xcur = a.saved ? segbase + SL.xin : xa;
    {  // lin0 (+ the caller-side x[mask] = 0)
      const float* w = P + L.p_lin0_w;
      const float* b = P + L.p_lin0_b;
      for (int idx = rw.lo * (NC / 4) + tid; idx < rw.hi * (NC / 4); idx += THREADS) {
        const int r = idx / (NC / 4), c0 = (idx % (NC / 4)) * 4;
        const int node = ext_id(a.perm, n0 + r);
        const float xv = (a.mask && a.mask[node]) ? 0.f : a.x[node];
        const float4 wv = ld4(w + c0), bv = ld4(b + c0);
        float4 o;
        o.x = xv * wv.x + bv.x; o.y = xv * wv.y + bv.y; o.z = xv * wv.z + bv.z; o.w = xv * wv.w + bv.w;
        st4(xcur + (unsigned)((nbS + r) * NC + c0), o);
      }
    }
    __syncthreads();
    // forward halo list in the LDS left over behind the topology (the W slot only lives there when it is `priv`)
    u16* hlist = tp + 2;
    int hcnt = 0;
    bool halo = false;
    if (CACHE && split) {
      const int cap = ((priv ? slot_b : LDS_BYTES) - used_b - 8) / 2;
      if (cap > 0 && !(a.no_halo & 1)) {
        hcnt = build_halo<THREADS>(rp, col, nullptr, rw, hlist, nullptr, cap, reinterpret_cast<int*>(tp));
        halo = hcnt <= cap;
      }
    }
    STAMP();
    for (int b = 0; b < L.nb; ++b) {
      float* base = a.saved ? segbase + (int64_t)b * SL.bstride : sc + L.sc_ev;
      float* xnext = a.saved ? segbase + (int64_t)(b + 1) * SL.bstride + SL.xin : (xcur == xa ? xb : xa);
      const float* pb = P + L.p_block0 + (int64_t)b * L.p_block_stride;
      float* y2g = sc + L.sc_y2;
      // conv1: K1, then K2 (+bias+ReLU)
      seg_proj<NC, 2 * NC, 2, EPI_ATT, THREADS, WLDS>(rw, xcur, nbS, pb + L.c1_W, base + o_h1, nbS, CACHE ? hA : nullptr, 0,
                                                 pb + L.c1_as, pb + L.c1_ad, base + o_as1, base + o_ad1, nbS,
                                                 CACHE ? sa : nullptr, CACHE ? sd : nullptr, nullptr, 0, nullptr, 0, wl1);
      group_sync<THREADS>(grp);                   // the gathers below read every row of h1 / a_src
      if (CACHE && split) {
        if (halo) {
          pull_list_rows<2 * NC, THREADS>(hA, base + o_h1 + (size_t)nbS * 2 * NC, hlist, hcnt);
          pull_list_small<2, THREADS>(sa, base + o_as1 + (size_t)nbS * 2, hlist, hcnt);
        } else {
          pull_rows4<THREADS>(hA, base + o_h1 + (size_t)nbS * 2 * NC, 2 * NC, rw, n);
          pull_flat<THREADS>(sa, base + o_as1 + (size_t)nbS * 2, rw.lo * 2, rw.hi * 2, n * 2);
        }
        __syncthreads();
      }
      STAMP();
      // K2 conv1: softmax (alpha -> HBM + LDS: the h2 table is dead now), then the gather
      if (CACHE && 2 * eg <= n * NC) {           // (wave-uniform) the alpha table fits the borrowed region
        seg_softmax<2, true, THREADS>(rw, rp, col, sa, sd, 0, base + o_al1, ebS, hB);
        __syncthreads();
        seg_gather<true, 2, NC, THREADS, UF>(rw, rp, col, hA, 0, hB, 0, pb + L.c1_b, base + o_o1, nbS);
      } else if (CACHE) {
        seg_softmax<2, false, THREADS>(rw, rp, col, sa, sd, 0, base + o_al1, ebS, nullptr);
        __syncthreads();
        seg_gather<true, 2, NC, THREADS, UF>(rw, rp, col, hA, 0, base + o_al1, ebS, pb + L.c1_b, base + o_o1, nbS);
      } else {
        seg_softmax<2, false, THREADS>(rw, rp, col, base + o_as1, base + o_ad1, nbS, base + o_al1, ebS, nullptr);
        __syncthreads();
        seg_gather<true, 2, NC, THREADS, UF>(rw, rp, col, base + o_h1, nbS, base + o_al1, ebS, pb + L.c1_b, base + o_o1,
                                         nbS);
      }
      __syncthreads();
      STAMP();
      // conv2
      seg_proj<2 * NC, NC, 1, EPI_ATT, THREADS, WLDS>(rw, base + o_o1, nbS, pb + L.c2_W, base + o_h2, nbS,
                                                 CACHE ? hB : nullptr, 0, pb + L.c2_as, pb + L.c2_ad, base + o_as2,
                                                 base + o_ad2, nbS, CACHE ? sa : nullptr, CACHE ? sd : nullptr,
                                                 nullptr, 0, nullptr, 0, wl2);
      group_sync<THREADS>(grp);
      if (CACHE && split) {
        if (halo) {
          pull_list_rows<NC, THREADS>(hB, base + o_h2 + (size_t)nbS * NC, hlist, hcnt);
          pull_list_small<1, THREADS>(sa, base + o_as2 + (size_t)nbS, hlist, hcnt);
        } else {
          pull_rows4<THREADS>(hB, base + o_h2 + (size_t)nbS * NC, NC, rw, n);
          pull_flat<THREADS>(sa, base + o_as2 + (size_t)nbS, rw.lo, rw.hi, n);
        }
        __syncthreads();
      }
      STAMP();
      // K2 conv2: alpha's LDS table sits in the upper half of the h1 region (y2 is written to the lower half)
      float* y2pub = split ? y2g : nullptr;       // partners read y2 from the global copy
      if (CACHE && eg <= n * NC) {
        float* al2L = hA + (size_t)n * NC;
        seg_softmax<1, true, THREADS>(rw, rp, col, sa, sd, 0, base + o_al2, ebS, al2L);
        __syncthreads();
        seg_gather<false, 1, NC, THREADS, UF>(rw, rp, col, hB, 0, al2L, 0, pb + L.c2_b, hA, 0, y2pub, n0);
      } else if (CACHE) {
        seg_softmax<1, false, THREADS>(rw, rp, col, sa, sd, 0, base + o_al2, ebS, nullptr);
        __syncthreads();
        seg_gather<false, 1, NC, THREADS, UF>(rw, rp, col, hB, 0, base + o_al2, ebS, pb + L.c2_b, hA, 0, y2pub, n0);
      } else {
        seg_softmax<1, false, THREADS>(rw, rp, col, base + o_as2, base + o_ad2, nbS, base + o_al2, ebS, nullptr);
        __syncthreads();
        seg_gather<false, 1, NC, THREADS, UF>(rw, rp, col, base + o_h2, nbS, base + o_al2, ebS, pb + L.c2_b, y2g, n0);
      }
      group_sync<THREADS>(grp);                   // K3 averages y2 over neighbours
      if (CACHE && split) {
        if (halo) pull_list_rows<NC, THREADS>(hA, y2g + (size_t)n0 * NC, hlist, hcnt);
        else      pull_rows4<THREADS>(hA, y2g + (size_t)n0 * NC, NC, rw, n);
        __syncthreads();
      }
      STAMP();
      // K3
      if (CACHE)
        seg_mean_fwd<NC, THREADS, UF>(rw, em, mrp, mcol, hA, 0, xcur, nbS, xnext, nbS);
      else
        seg_mean_fwd<NC, THREADS, UF>(rw, em, mrp, mcol, y2g, n0, xcur, nbS, xnext, nbS);
      __syncthreads();
      STAMP();
      xcur = xnext;
    }
    {  // lin1
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin1_w + (tid % G) * 4);
      const float bias = P[L.p_lin1_b];
      const int rounds = (rw.hi - rw.lo + THREADS / G - 1) / (THREADS / G);
      for (int it = 0; it < rounds; ++it) {
        int r = rw.lo + it * (THREADS / G) + tid / G;
        const bool valid = r < rw.hi;
        if (!valid) r = rw.hi - 1;
        const float4 xv = ld4(xcur + (unsigned)((nbS + r) * NC + (tid % G) * 4));
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) a.out[ext_id(a.perm, n0 + r)] = d + bias;
      }
    }
    __syncthreads();
    STAMP();
  }

  if (a.phases & PH_LOSS) {
    // M = number of masked nodes in the WHOLE batch (every workgroup counts them itself: N bytes from L2)
    float cnt = 0.f;
    for (int i = tid; i < a.N; i += THREADS) cnt += a.mask[i] ? 1.f : 0.f;
    const float Mn = block_sum<THREADS>(cnt, ldsf);
    float part_sum = 0.f;
    for (int r = rw.lo + tid; r < rw.hi; r += THREADS) {
      const int node = ext_id(a.perm, n0 + r);
      if (a.mask[node]) {
        const float d = a.out[node] - a.y[node];
        part_sum = fmaf(d, d, part_sum);
      }
    }
    part_sum = block_sum<THREADS>(part_sum, ldsf);
    if (tid == 0) {
      a.loss_part[seg * M + part] = part_sum;
      if (seg == 0 && part == 0) a.loss_part[a.num_segments * M] = Mn;
    }
    const float scale = Mn > 0.f ? 2.f / Mn : 0.f;
    for (int r = rw.lo + tid; r < rw.hi; r += THREADS) {
      const int node = ext_id(a.perm, n0 + r);
      a.g_out[node] = a.mask[node] ? (a.out[node] - a.y[node]) * scale : 0.f;
    }
    __syncthreads();
    STAMP();
  }

  if (a.phases & GATRES_PHASE_BACKWARD) {
    // LDS map: [red 3*THREADS] ( [RA n x 2NC : g_pre | g_y2, later g_out1] [ge 2*eg] [gad 2n] [spare 2n] ) topology
    float* red = ldsf;
    float* RA = red + 3 * THREADS;
    float* geL = RA + (size_t)n * 2 * NC;
    float* gadL = geL + 2 * (size_t)even(eg);
    float* spareL = gadL + 2 * (size_t)n;
    u16* tp = reinterpret_cast<u16*>(CACHE ? (spareL + 2 * (size_t)n) : RA);
    u16* rp = tp;              tp += even(n + 1);
    u16* col = tp;             tp += even(eg);
    u16* trp = tp;             tp += even(n + 1);
    u16* teid = tp;            tp += even(eg);
    u16* tdst = tp;            tp += even(eg);
    u16* mrp = tp;             tp += even(n + 1);
    u16* mtrp = tp;            tp += even(n + 1);
    u16* mtdst = tp;
    tp += even(em);
    constexpr int WL_FLOATS = 2 * NC * (2 * NC + 4) + 4 * NC;
    constexpr bool WLDS = WL_FLOATS * 4 <= 40960;
    float* wlB = reinterpret_cast<float*>(lds_raw + ((reinterpret_cast<unsigned char*>(tp) - lds_raw + 15) & ~15));
    __syncthreads();           // forward's LDS contents are dead from here
    copy_rowptr16<THREADS>(rp, a.rowptr, n0, n, e0);
    copy_idx16<THREADS>(col, a.col, e0, eg, n0);
    copy_rowptr16<THREADS>(trp, a.t_rowptr, n0, n, e0);
    copy_idx16<THREADS>(teid, a.t_eid, e0, eg, e0);
    copy_idx16<THREADS>(tdst, a.t_dst, e0, eg, n0);
    copy_rowptr16<THREADS>(mrp, a.m_rowptr, n0, n, em0);
    copy_rowptr16<THREADS>(mtrp, a.mt_rowptr, n0, n, em0);
    copy_idx16<THREADS>(mtdst, a.mt_dst, em0, em, n0);

    const SegLayout& SL = a.SL;
    const float* segbase = a.saved + (int64_t)seg * SL.total;
    float* gp_cur = sc + L.sc_gpa;       // global copies of g_pre (row-wise residual reads, partners' gathers)
    float* gp_nxt = sc + L.sc_gpb;
    // gathered / small backward tables: LDS when CACHE, else global scratch (conv2 has private arrays: see Layout).
    // With a split segment the LDS tables are completed from the *_pub global copies after each flag barrier.
    const bool pub = CACHE && split;
    float* gpT = CACHE ? RA : nullptr;                                   // g_pre, LDS copy for the K3 gather
    float* gy2T = CACHE ? RA + (size_t)n * NC : sc + L.sc_gy2;  const int gy2b = CACHE ? 0 : n0;
    float* go1T = CACHE ? RA : sc + L.sc_go1;                   const int go1b = CACHE ? 0 : n0;
    float* ge1T = CACHE ? geL : sc + L.sc_ge;                   const int ge_b = CACHE ? 0 : e0;
    float* ge2T = CACHE ? geL : sc + L.sc_ge2;
    float* gad1T = CACHE ? gadL : sc + L.sc_gad;                const int gd_b = CACHE ? 0 : n0;
    float* gad2T = CACHE ? gadL : sc + L.sc_gad2;
    float* slab = split ? a.part_slabs + ((int64_t)seg * M + part) * L.slab_stride
                        : a.slabs + (int64_t)seg * L.slab_stride;
    const int64_t w = 2LL * NC * NC;
    const float* xfinal = segbase + (int64_t)L.nb * SL.bstride + SL.xin;
    seg_lin1_bwd<NC, THREADS>(rw, n0, a.perm, a.g_out, xfinal, P + L.p_lin1_w, gp_cur, gpT, slab + L.p_lin1_w,
                              slab + L.p_lin1_b, L.nb > 0 ? 1 : 0, red);
    // backward halo list (remote dst rows + edge ids of own out-edges) in the LDS behind the W slot
    u16* hrow = reinterpret_cast<u16*>(wlB + (WLDS ? ((WL_FLOATS + 3) & ~3) : 0)) + 2;
    int hcnt = 0;
    bool halo = false;
    if (pub) {
      const int cap = (int)((lds_raw + LDS_BYTES - reinterpret_cast<unsigned char*>(hrow)) / 4) - 4;
      if (cap > 0 && !(a.no_halo & 1)) {
        hcnt = build_halo<THREADS>(trp, tdst, teid, rw, hrow, hrow + cap, cap, reinterpret_cast<int*>(hrow - 2));
        halo = hcnt <= cap;
      }
    }
    const u16* hedge = hrow + ((int)((lds_raw + LDS_BYTES - reinterpret_cast<unsigned char*>(hrow)) / 4) - 4);
    STAMP();
    for (int b = L.nb - 1; b >= 0; --b) {
      const float* base = segbase + (int64_t)b * SL.bstride;
      const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
      const float* pb = P + po;
      float* sb = slab + po;
      const float* wt1 = a.wt + (int64_t)b * 2 * w;
      const float* wt2 = wt1 + w;
      group_sync<THREADS>(grp);                  // K3 backward gathers g_pre of neighbours
      publish_items<THREADS>(a, seg, part, 2 * (L.nb - 1 - b), grp.local || !split, false);   // the blocks above are kept
      const int elo = rp[rw.lo], ehi = rp[rw.hi];       // own in-edge range (edges are dst-sorted)
      if (pub) {
        if (halo) pull_list_rows<NC, THREADS>(gpT, gp_cur + (size_t)n0 * NC, hrow, hcnt);
        else      pull_rows4<THREADS>(gpT, gp_cur + (size_t)n0 * NC, NC, rw, n);
        __syncthreads();
      }
      // K3 backward
      if (CACHE) seg_mean_bwd<NC, THREADS, UB>(rw, em, mrp, mtrp, mtdst, gpT, 0, gy2T, gy2b, pub ? sc + L.sc_gy2 : nullptr, n0);
      else       seg_mean_bwd<NC, THREADS, UB>(rw, em, mrp, mtrp, mtdst, gp_cur, n0, gy2T, gy2b);
      __syncthreads();
      STAMP();
      // conv2.  g_h2 / g_alpha tables go to this block's kept area: dx2 reads g_h2 back, the deferred
      // parameter-gradient launch reads all of them (the dW / att gradients are off the critical path).
      float* keep = sc + L.sc_keep + (int64_t)b * L.keep_stride;
      float* gh = keep + L.k_gh1;
      float* gh2 = keep + L.k_gh2;
      seg_edge_dots<1, NC, THREADS, 2>(rw, 0, rp, col, gy2T, gy2b, base + SL.h2, ge2T, ge_b);
      __syncthreads();
      seg_bias_part<NC, THREADS>(rw, gy2T, gy2b, red);
      seg_softmax_bwd<1, THREADS>(rw, 0, 0, rp, col, base + SL.al2, base + SL.as2, base + SL.ad2, ge2T, ge_b, gad2T,
                                  gd_b, pub ? sc + L.sc_ge2 : nullptr, e0, nullptr, 0);
      group_sync<THREADS>(grp);                  // the source-major stage reads g_y2 / g_e / g_a_dst of every dst
      if (pub) {                 // (g_a_dst is only read for own rows: no pull)
        if (halo) {
          pull_list_rows<NC, THREADS>(gy2T, sc + L.sc_gy2 + (size_t)n0 * NC, hrow, hcnt);
          pull_list_small<1, THREADS>(ge2T, sc + L.sc_ge2 + e0, hedge, hcnt);
        } else {
          pull_rows4<THREADS>(gy2T, sc + L.sc_gy2 + (size_t)n0 * NC, NC, rw, n);
          pull_flat<THREADS>(ge2T, sc + L.sc_ge2 + e0, elo, ehi, eg);
        }
        __syncthreads();
      }
      STAMP();
      seg_bias_finish<NC, THREADS>(red, sb + L.c2_b);
      seg_agg_bwd_src<1, NC, THREADS, UB>(rw, 0, trp, teid, tdst, gy2T, gy2b, base + SL.al2, ge2T, ge_b, gad2T, gd_b,
                                      pb + L.c2_as, pb + L.c2_ad, gh2, n0, keep + L.k_gas2, keep + L.k_gad2);
      __syncthreads();         // g_y2 (RA) is dead: dx2 overwrites RA with g_out1
      STAMP();
      seg_proj<NC, 2 * NC, 1, EPI_RESID_MASK, THREADS, WLDS>(rw, gh2, n0, wt2, go1T, go1b, pub ? sc + L.sc_go1 : nullptr,
                                                        n0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                                        nullptr, 0, base + SL.o1, 0, wlB);
      __syncthreads();
      STAMP();
      // conv1
      seg_edge_dots<2, NC, THREADS, 2>(rw, 0, rp, col, go1T, go1b, base + SL.h1, ge1T, ge_b);
      __syncthreads();
      seg_bias_part<2 * NC, THREADS>(rw, go1T, go1b, red);
      seg_softmax_bwd<2, THREADS>(rw, 0, 0, rp, col, base + SL.al1, base + SL.as1, base + SL.ad1, ge1T, ge_b, gad1T,
                                  gd_b, pub ? sc + L.sc_ge : nullptr, e0, nullptr, 0);
      group_sync<THREADS>(grp);
      publish_items<THREADS>(a, seg, part, 2 * (L.nb - 1 - b) + 1, grp.local || !split, false);   // conv2 tables complete
      if (pub) {
        if (halo) {
          pull_list_rows<2 * NC, THREADS>(go1T, sc + L.sc_go1 + (size_t)n0 * 2 * NC, hrow, hcnt);
          pull_list_small<2, THREADS>(ge1T, sc + L.sc_ge + (size_t)e0 * 2, hedge, hcnt);
        } else {
          pull_rows4<THREADS>(go1T, sc + L.sc_go1 + (size_t)n0 * 2 * NC, 2 * NC, rw, n);
          pull_flat<THREADS>(ge1T, sc + L.sc_ge + (size_t)e0 * 2, elo * 2, ehi * 2, eg * 2);
        }
        __syncthreads();
      }
      STAMP();
      seg_bias_finish<2 * NC, THREADS>(red, sb + L.c1_b);
      seg_agg_bwd_src<2, NC, THREADS, UB>(rw, 0, trp, teid, tdst, go1T, go1b, base + SL.al1, ge1T, ge_b, gad1T, gd_b,
                                      pb + L.c1_as, pb + L.c1_ad, gh, n0, keep + L.k_gas1, keep + L.k_gad1);
      __syncthreads();         // g_out1 (RA) is dead: dx1 writes the next g_pre into RA's low half
      STAMP();
      // d/d xin = conv1 path + residual, masked by the previous block's ReLU (block 0's input is lin0: no ReLU)
      seg_proj<2 * NC, NC, 1, EPI_RESID_MASK, THREADS, WLDS>(rw, gh, n0, wt1, gp_nxt, n0, gpT, 0, nullptr, nullptr, nullptr,
                                                        nullptr, 0, nullptr, nullptr, gp_cur, n0,
                                                        b > 0 ? base + SL.xin : nullptr, 0, wlB);
      STAMP();
      float* t = gp_cur; gp_cur = gp_nxt; gp_nxt = t;
    }
    group_sync<THREADS>(grp);
    publish_items<THREADS>(a, seg, part, 2 * L.nb, grp.local || !split, !split);
    seg_lin0_bwd<NC, THREADS>(rw, n0, a.perm, gp_cur, a.x, a.mask, slab + L.p_lin0_w, slab + L.p_lin0_b, red);
    if (split && a.C > 0) {                   // last item: fold the lin0 / lin1 partial rows
      group_sync<THREADS>(grp);
      publish_items<THREADS>(a, seg, part, 2 * L.nb + 1, grp.local, true);
    }
    if (a.g_x) {
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin0_w + (tid % G) * 4);
      const int rounds = (rw.hi - rw.lo + THREADS / G - 1) / (THREADS / G);
      for (int it = 0; it < rounds; ++it) {
        int r = rw.lo + it * (THREADS / G) + tid / G;
        const bool valid = r < rw.hi;
        if (!valid) r = rw.hi - 1;
        const float4 xv = ld4(gp_cur + ((size_t)n0 + r) * NC + (tid % G) * 4);
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) a.g_x[ext_id(a.perm, n0 + r)] = d;
      }
    }
  }
  if (split && tid == 0 && *a.err) {        // a partner never arrived: make the failure visible in the results
    if ((a.phases & GATRES_PHASE_FORWARD) && a.out) a.out[n0] = NAN;
    if (a.phases & GATRES_PHASE_BACKWARD) a.slabs[(int64_t)seg * L.slab_stride + L.p_lin1_b] = NAN;
  }
  if (STAMPS_PTR && blockIdx.x == 0 && threadIdx.x == 0) {
    STAMPS_PTR[a.stamp_cap + 1] = clock64();
    STAMPS_PTR[a.stamp_cap + 2] = wall_clock64();
  }
}

}  // namespace

#define GATRES_LAUNCH_WHOLE(NC, TH)                                                                                   \
  do {                                                                                                                \
    if (cache) hipLaunchKernelGGL((gatres_fused_kernel<NC, TH, true>), dim3(grid), dim3(TH), 0, st, a);               \
    else       hipLaunchKernelGGL((gatres_fused_kernel<NC, TH, false>), dim3(grid), dim3(TH), 0, st, a);              \
    return gatres_launch_status();                                                                                    \
  } while (0)

extern "C" __attribute__((visibility("hidden"))) int gatres_fused_launch_whole(const void* args, int nc, int threads, int cache,
                                                                              unsigned grid, void* stream) {
  const FusedArgs& a = *static_cast<const FusedArgs*>(args);
  hipStream_t st = gatres_stream(stream);
  if (nc == 4 && threads == 1024) GATRES_LAUNCH_WHOLE(4, 1024);
  if (nc == 8 && threads == 1024) GATRES_LAUNCH_WHOLE(8, 1024);
  if (nc == 16 && threads == 1024) GATRES_LAUNCH_WHOLE(16, 1024);
  if (nc == 32 && threads == 1024) GATRES_LAUNCH_WHOLE(32, 1024);
  if (nc == 32 && threads == 512) GATRES_LAUNCH_WHOLE(32, 512);
#ifdef GATRES_DIAG_BUILD
  // wide models on the per-snapshot kernels (GATRES_FUSED_WIDE=1): 1 000 VGPR spills, occupancy 1 -- kept only to be measured
  if (nc == 64 && threads == 512) GATRES_LAUNCH_WHOLE(64, 512);
  if (nc == 128 && threads == 256) GATRES_LAUNCH_WHOLE(128, 256);
#endif
  return GATRES_E_UNSUPPORTED;
}
