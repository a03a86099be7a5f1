// Blocked kernels for wide models in bf16 storage (gatres_large, BASELINE configs 3 and 5): a SPARSE stage and the dense
// projection that consumes its output, one launch.
//
// The per-op path runs a block of GResBlockMeanConv (GraphModels.py:462-468) as 14 launches of 7 - 18 us, and every one of
// them is mostly its own latency chain: ~3 us from the graph node's start to the first instruction, 4 - 7 us until the first
// MFMA of a projection (every workgroup staging W), the matrix work itself 2 - 3 us (DESIGN.md section 3.3,
// profiles/r03_proj_probe.txt).  The projections' inputs are exactly the row blocks the sparse kernels in front of them
// produce, so here ONE 1024-thread workgroup per CU
//   1. aggregates a block of 64 / 128 destination (or source) rows with the sparse kernels' own slot arithmetic (k_aggregate.hip:
//      same operands, same order -- the same bits), two row sets per wave in flight together, the next block's row pointers and
//      neighbour ids requested one block ahead; the rows go to HBM (later launches need them: weight gradients, ReLU masks) AND
//      into a double-buffered LDS tile,
//   2. runs the block through the matrix cores with proj_bf16_tile_kernel's own inner loop (W staged once per launch in that
//      kernel's LDS image, the B fragments from the LDS tile): the same accumulator-to-feature map, logits and epilogue, so a
//      blocked launch gives the BITS of the two per-op launches it replaces.
// MEASURED (profiles/r05_blocked_probe.txt): bit-identical, and 5 - 15 % SLOWER than the per-op pairs in four forms of this
// kernel, at 194, 390 and 780 rows per CU alike -- a launch is a chain of memory round trips either way, and fusing puts the
// sparse chain and the dense chain of a row block in series inside one workgroup where the per-op kernels run 20 independent
// waves per CU.  The per-op drivers therefore take these kernels only on request (GATRES_BLOCKED=1).
// Three sparse stages x two shapes:
//   agg_proj   GATConv aggregation (softmax, weighted sum, bias, ReLU)  -> next GATConv's projection + logits  (conv1 -> conv2)
//   mean_proj  SimpleConv mean + residual + ReLU                        -> next block's conv1 projection + logits
//   src_dx     GATConv backward, source-major (g_h, g_a_src)            -> input gradient g_h W (+ residual, ReLU mask)
// Rows of more than 6 edges take an edge-at-a-time loop; plans with a row of more than 32 edges (GATRES_GRAPH_DEG_LE32 not
// set) keep the per-op kernels, whose hub rows are reduced by whole waves.
#include <type_traits>
#include "gatres_common.h"

namespace {

typedef gatres_bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef gatres_bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { BK_GAT_FWD = 0, BK_MEAN_FWD = 1, BK_GAT_BWD_SRC = 2 };
enum { BE_ATT = 0, BE_RESID_MASK = 1 };

struct BlockedArgs {
  // ---- sparse stage
  const int* rowptr;            // GAT forward: rowptr / col; mean: m_rowptr / m_col; backward: t_rowptr / t_dst
  const int* col;
  const int* eid;               // backward: t_eid
  const gatres_bf16* src;       // the gathered table [N][K]: h (forward), y2 (mean), g_out (backward)
  const float* a_src;           // GAT forward: logits [N][HIN]
  const float* a_dst;
  const float* bias;            // GAT forward: [K]
  float* alpha;                 // GAT forward: written [E'][HIN]; backward: read
  const float* g_e;             // backward: [E'][HIN]
  const float* g_a_dst;         // backward: [N][HIN]
  const float* att_s_in;        // backward: this convolution's att_src / att_dst [K]
  const float* att_d_in;
  float* g_a_src;               // backward: written [N][HIN]
  const gatres_bf16* x0;        // mean: the residual rows [N][K]
  gatres_bf16* mid;             // the sparse stage's output [N][K]: o1 / x_next / g_h
  // ---- dense stage
  const gatres_bf16* W;         // [M][K] row-major (forward: W; backward: W^T)
  gatres_bf16* out;             // [N][M]
  const float* att_s_out;       // BE_ATT: the NEXT convolution's att_src / att_dst [M], logits [N][HOUT]
  const float* att_d_out;
  float* a_src_out;
  float* a_dst_out;
  const gatres_bf16* resid;     // BE_RESID_MASK: [N][M] or null
  const gatres_bf16* relu_ref;  // BE_RESID_MASK: [N][M] or null
  int N;
};

__device__ __forceinline__ float4 widen_lo(const uint4 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 widen_hi(const uint4 u) {
  return make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                     __uint_as_float(u.w & 0xffff0000u));
}
struct Row8 { float4 lo, hi; };
__device__ __forceinline__ Row8 row8_zero() { return {f4zero(), f4zero()}; }
__device__ __forceinline__ uint4 ld_raw8(const gatres_bf16* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void axpy8(Row8& a, float s, const uint4 raw) {
  gatres_axpy4(a.lo, s, widen_lo(raw));
  gatres_axpy4(a.hi, s, widen_hi(raw));
}
__device__ __forceinline__ void add4(float4& a, const float4 v) { a.x = a.x + v.x; a.y = a.y + v.y; a.z = a.z + v.z; a.w = a.w + v.w; }
__device__ __forceinline__ void add8(Row8& a, const uint4 raw) { add4(a.lo, widen_lo(raw)); add4(a.hi, widen_hi(raw)); }
__device__ __forceinline__ bf16x8 pack8(const Row8& r) {
  bf16x8 b;
  b[0] = (gatres_bf16)r.lo.x; b[1] = (gatres_bf16)r.lo.y; b[2] = (gatres_bf16)r.lo.z; b[3] = (gatres_bf16)r.lo.w;
  b[4] = (gatres_bf16)r.hi.x; b[5] = (gatres_bf16)r.hi.y; b[6] = (gatres_bf16)r.hi.z; b[7] = (gatres_bf16)r.hi.w;
  return b;
}

// ---------------------------------------------------------------------------------------------- sparse stages, one row per lane group
// Eight features per lane (one 16-byte access), K / 8 lanes per row: k_aggregate.hip's W = 8 instances restated for a row
// that the CALLER chose (a block's rows instead of the grid's); arithmetic statement for statement theirs.
//
// A row's work is a chain of three dependent memory round trips: rowptr -> col (neighbour ids) -> neighbour rows.  The
// per-op kernels hide it behind 20 waves per CU; here a CU has 16 and they stop at two barriers per phase, so the chain is cut:
// the INDEX part of a row (bk_idx_*: row pointers, then ids -- `Idx`) is loaded one phase AHEAD, under the current phase's
// arithmetic, barriers and matrix work, and a phase starts with the neighbour rows themselves.  `after_issue()` is called
// where a row's loads have just been issued: the caller issues the next phase's row-pointer loads there.

struct IdxGat { int beg, deg; float adst; int jj[6]; int jm; };
template <int K, int HIN>
__device__ __forceinline__ void bk_idx1_gat(const BlockedArgs& a, int row, int lr, IdxGat& x) {
  constexpr int C = K / HIN;
  x.beg = a.rowptr[row]; x.deg = a.rowptr[row + 1];       // (deg holds `end` until bk_idx2)
  x.adst = a.a_dst[row * HIN + (lr * 8) / C];
}
template <int K, int HIN>
__device__ __forceinline__ void bk_idx2_gat(const BlockedArgs& a, int lr, IdxGat& x) {
  constexpr int C = K / HIN, LH = C / 8;
  x.deg = x.deg - x.beg;
  const int kk = lr & (LH - 1);
#pragma unroll
  for (int k = 0; k < 6; ++k) x.jj[k] = a.col[x.beg + min(k, x.deg - 1)];
  x.jm = a.col[x.beg + min(min(kk, 5), x.deg - 1)];
}

// GATConv forward (gat_aggregate_fwd_kernel): softmax over the in-edges, weighted sum, bias (+ ReLU); writes alpha.
template <int K, int HIN, bool RELU, class F>
__device__ __forceinline__ Row8 bk_gat_fwd(const BlockedArgs& a, const IdxGat& x, bool valid, int lr, F&& after_issue) {
  constexpr int C = K / HIN, LH = C / 8;                 // lanes per head
  static_assert(LH == 8 || LH == 16, "heads of 64 or 128 features");
  const int c0 = lr * 8, hd = c0 / C, kk = lr & (LH - 1);
  const int beg = x.beg, deg = x.deg, end = beg + deg;
  const float adst = x.adst;
  const gatres_bf16* h = a.src + c0;
  Row8 acc = row8_zero();
  auto slots_lane = [&](auto KC) {
    constexpr int MAXD = decltype(KC)::value;
    // (slot kk's source: x.jm was clamped to slot 5; with four slots the lanes beyond slot 3 are padding either way)
    const int jm = MAXD == 6 ? x.jm : (kk == 0 ? x.jj[0] : kk == 1 ? x.jj[1] : kk == 2 ? x.jj[2] : x.jj[3]);
    uint4 v[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) v[k] = ld_raw8(h + (size_t)x.jj[k] * K);
    const float asv = a.a_src[jm * HIN + hd];
    after_issue();
    const float sv = gatres_leaky(asv + adst);
    const float so = kk < deg ? sv : -INFINITY;
    float m = so;
    m = fmaxf(m, gatres_dpp<0xB1>(m));
    m = fmaxf(m, gatres_dpp<0x4E>(m));
    m = fmaxf(m, gatres_dpp<0x141>(m));
    if constexpr (LH == 16) m = fmaxf(m, gatres_dpp<0x140>(m));
    const float ex = expf(so - m);
    auto bcast = [&](float xv, auto Kc) {
      constexpr int pat = (0x1f & ~(LH - 1)) | (decltype(Kc)::value << 5);
      return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(xv), pat));
    };
    float e[MAXD];
    e[0] = bcast(ex, std::integral_constant<int, 0>{}); e[1] = bcast(ex, std::integral_constant<int, 1>{});
    e[2] = bcast(ex, std::integral_constant<int, 2>{}); e[3] = bcast(ex, std::integral_constant<int, 3>{});
    if constexpr (MAXD > 4) { e[4] = bcast(ex, std::integral_constant<int, 4>{}); e[5] = bcast(ex, std::integral_constant<int, 5>{}); }
    float Z = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) Z = Z + e[k];
    Z = Z + GATRES_SOFTMAX_EPS;
    const float alm = ex / Z;
    if (valid && kk < deg) a.alpha[(size_t)(beg + kk) * HIN + hd] = alm;
    float al[MAXD];
    al[0] = bcast(alm, std::integral_constant<int, 0>{}); al[1] = bcast(alm, std::integral_constant<int, 1>{});
    al[2] = bcast(alm, std::integral_constant<int, 2>{}); al[3] = bcast(alm, std::integral_constant<int, 3>{});
    if constexpr (MAXD > 4) { al[4] = bcast(alm, std::integral_constant<int, 4>{}); al[5] = bcast(alm, std::integral_constant<int, 5>{}); }
#pragma unroll
    for (int k = 0; k < MAXD; ++k)
      if (k < deg) axpy8(acc, al[k], v[k]);
  };
  if (__builtin_expect(__ballot(deg > 6) == 0ULL, 1)) {        // (wave-uniform: every row of the wave takes the slot path)
    if (__ballot(deg > 4) == 0ULL) slots_lane(std::integral_constant<int, 4>{});
    else slots_lane(std::integral_constant<int, 6>{});
  } else {
    after_issue();
    if (deg <= 6) {
      // (the lanes of a head share the row: the group stays whole under the divergent branch)
      constexpr int MAXD = 6;
      uint4 v[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) v[k] = ld_raw8(h + (size_t)x.jj[k] * K);
      const float sv = gatres_leaky(a.a_src[x.jm * HIN + hd] + adst);
      const float so = kk < deg ? sv : -INFINITY;
      float m = so;
      m = fmaxf(m, gatres_dpp<0xB1>(m));
      m = fmaxf(m, gatres_dpp<0x4E>(m));
      m = fmaxf(m, gatres_dpp<0x141>(m));
      if constexpr (LH == 16) m = fmaxf(m, gatres_dpp<0x140>(m));
      const float ex = expf(so - m);
      auto bcast = [&](float xv, auto Kc) {
        constexpr int pat = (0x1f & ~(LH - 1)) | (decltype(Kc)::value << 5);
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(xv), pat));
      };
      float e[MAXD];
      e[0] = bcast(ex, std::integral_constant<int, 0>{}); e[1] = bcast(ex, std::integral_constant<int, 1>{});
      e[2] = bcast(ex, std::integral_constant<int, 2>{}); e[3] = bcast(ex, std::integral_constant<int, 3>{});
      e[4] = bcast(ex, std::integral_constant<int, 4>{}); e[5] = bcast(ex, std::integral_constant<int, 5>{});
      float Z = 0.f;
#pragma unroll
      for (int k = 0; k < MAXD; ++k) Z = Z + e[k];
      Z = Z + GATRES_SOFTMAX_EPS;
      const float alm = ex / Z;
      if (valid && kk < deg) a.alpha[(size_t)(beg + kk) * HIN + hd] = alm;
      float al[MAXD];
      al[0] = bcast(alm, std::integral_constant<int, 0>{}); al[1] = bcast(alm, std::integral_constant<int, 1>{});
      al[2] = bcast(alm, std::integral_constant<int, 2>{}); al[3] = bcast(alm, std::integral_constant<int, 3>{});
      al[4] = bcast(alm, std::integral_constant<int, 4>{}); al[5] = bcast(alm, std::integral_constant<int, 5>{});
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < deg) axpy8(acc, al[k], v[k]);
    } else {                                                   // up to 32 in-edges: edge after edge (k_aggregate.hip's loop form)
      const bool leader = valid && (c0 & (C - 1)) == 0;
      float m = -INFINITY;
      for (int e = beg; e < end; ++e) m = fmaxf(m, gatres_leaky(a.a_src[a.col[e] * HIN + hd] + adst));
      float Z = 0.f;
      for (int e = beg; e < end; ++e) Z = Z + expf(gatres_leaky(a.a_src[a.col[e] * HIN + hd] + adst) - m);
      Z = Z + GATRES_SOFTMAX_EPS;
      for (int e = beg; e < end; ++e) {
        const int j = a.col[e];
        const float al = expf(gatres_leaky(a.a_src[j * HIN + hd] + adst) - m) / Z;
        if (leader) a.alpha[(size_t)e * HIN + hd] = al;
        axpy8(acc, al, ld_raw8(h + (size_t)j * K));
      }
    }
  }
  const float4 b0 = ld4(a.bias + c0), b1 = ld4(a.bias + c0 + 4);
  add4(acc.lo, b0); add4(acc.hi, b1);
  if (RELU) {
    acc.lo.x = fmaxf(acc.lo.x, 0.f); acc.lo.y = fmaxf(acc.lo.y, 0.f); acc.lo.z = fmaxf(acc.lo.z, 0.f); acc.lo.w = fmaxf(acc.lo.w, 0.f);
    acc.hi.x = fmaxf(acc.hi.x, 0.f); acc.hi.y = fmaxf(acc.hi.y, 0.f); acc.hi.z = fmaxf(acc.hi.z, 0.f); acc.hi.w = fmaxf(acc.hi.w, 0.f);
  }
  return acc;
}

// SimpleConv mean + residual + ReLU (mean_residual_relu_fwd_kernel)
struct IdxMean { int beg, end0; int jj[4]; };
__device__ __forceinline__ void bk_idx1_mean(const BlockedArgs& a, int row, IdxMean& x) {
  x.beg = a.rowptr[row]; x.end0 = a.rowptr[row + 1];
}
__device__ __forceinline__ void bk_idx2_mean(const BlockedArgs& a, int row, IdxMean& x) {
  const int deg = x.end0 - x.beg;
#pragma unroll
  for (int k = 0; k < 4; ++k) x.jj[k] = k < deg ? a.col[x.beg + k] : row;
}
template <int K, class F>
__device__ __forceinline__ Row8 bk_mean_fwd(const BlockedArgs& a, const IdxMean& x, int row, int lr, F&& after_issue) {
  const int c0 = lr * 8;
  const int beg = x.beg, end0 = x.end0;
  const gatres_bf16* y = a.src + c0;
  const uint4 r = ld_raw8(a.x0 + (size_t)row * K + c0);
  Row8 acc = row8_zero();
  int end = end0;
  if (__ballot(end - beg > 4) == 0ULL) {
    const int deg = end - beg;
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld_raw8(y + (size_t)x.jj[k] * K);
    after_issue();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < deg) add8(acc, v[k]);
    end = beg;
  } else {
    after_issue();
  }
  int e = beg;
  for (; e + 1 < end; e += 2) {
    const uint4 v0 = ld_raw8(y + (size_t)a.col[e] * K), v1 = ld_raw8(y + (size_t)a.col[e + 1] * K);
    add8(acc, v0);
    add8(acc, v1);
  }
  if (e < end) add8(acc, ld_raw8(y + (size_t)a.col[e] * K));
  const float cnt = (float)max(end0 - beg, 1);
  const float4 rl = widen_lo(r), rh = widen_hi(r);
  Row8 o;
  o.lo.x = fmaxf(acc.lo.x / cnt + rl.x, 0.f); o.lo.y = fmaxf(acc.lo.y / cnt + rl.y, 0.f);
  o.lo.z = fmaxf(acc.lo.z / cnt + rl.z, 0.f); o.lo.w = fmaxf(acc.lo.w / cnt + rl.w, 0.f);
  o.hi.x = fmaxf(acc.hi.x / cnt + rh.x, 0.f); o.hi.y = fmaxf(acc.hi.y / cnt + rh.y, 0.f);
  o.hi.z = fmaxf(acc.hi.z / cnt + rh.z, 0.f); o.hi.w = fmaxf(acc.hi.w / cnt + rh.w, 0.f);
  return o;
}

// GATConv backward, source-major (gat_aggregate_bwd_src_kernel): g_h and g_a_src of a source row
struct IdxSrc { int beg, end0; float gad; int ee[4], ii[4]; };
template <int K, int HIN>
__device__ __forceinline__ void bk_idx1_src(const BlockedArgs& a, int row, int lr, IdxSrc& x) {
  constexpr int C = K / HIN;
  x.beg = a.rowptr[row]; x.end0 = a.rowptr[row + 1];
  x.gad = a.g_a_dst[row * HIN + (lr * 8) / C];
}
__device__ __forceinline__ void bk_idx2_src(const BlockedArgs& a, int row, IdxSrc& x) {
  const int deg = x.end0 - x.beg;
#pragma unroll
  for (int k = 0; k < 4; ++k) { x.ee[k] = k < deg ? a.eid[x.beg + k] : 0; x.ii[k] = k < deg ? a.col[x.beg + k] : row; }
}
template <int K, int HIN, class F>
__device__ __forceinline__ Row8 bk_gat_bwd_src(const BlockedArgs& a, const IdxSrc& x, int row, bool valid, int lr, F&& after_issue) {
  constexpr int C = K / HIN;
  const int c0 = lr * 8, hd = c0 / C;
  const bool leader = valid && (c0 & (C - 1)) == 0;
  const int beg = x.beg, end0 = x.end0;
  const gatres_bf16* go = a.src + c0;
  Row8 acc = row8_zero();
  float gas = 0.f;
  int end = end0;
  if (__ballot(end - beg > 4) == 0ULL) {
    const int deg = end - beg;
    uint4 gv[4];
    float al[4], ge[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gv[k] = ld_raw8(go + (size_t)x.ii[k] * K);
      al[k] = a.alpha[(size_t)x.ee[k] * HIN + hd];
      ge[k] = a.g_e[(size_t)x.ee[k] * HIN + hd];
    }
    after_issue();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < deg) { gas = gas + ge[k]; axpy8(acc, al[k], gv[k]); }
    end = beg;
  } else {
    after_issue();
  }
  int t = beg;
  for (; t + 1 < end; t += 2) {
    const int e0 = a.eid[t], e1 = a.eid[t + 1];
    const int i0 = a.col[t], i1 = a.col[t + 1];
    const uint4 v0 = ld_raw8(go + (size_t)i0 * K), v1 = ld_raw8(go + (size_t)i1 * K);
    const float al0 = a.alpha[(size_t)e0 * HIN + hd], al1 = a.alpha[(size_t)e1 * HIN + hd];
    gas = gas + a.g_e[(size_t)e0 * HIN + hd];
    gas = gas + a.g_e[(size_t)e1 * HIN + hd];
    axpy8(acc, al0, v0);
    axpy8(acc, al1, v1);
  }
  if (t < end) {
    const int e0 = a.eid[t], i0 = a.col[t];
    const uint4 v0 = ld_raw8(go + (size_t)i0 * K);
    const float al0 = a.alpha[(size_t)e0 * HIN + hd];
    gas = gas + a.g_e[(size_t)e0 * HIN + hd];
    axpy8(acc, al0, v0);
  }
  if (leader) a.g_a_src[row * HIN + hd] = gas;
  const float gad = x.gad;
  gatres_axpy4(acc.lo, gas, ld4(a.att_s_in + c0));     gatres_axpy4(acc.lo, gad, ld4(a.att_d_in + c0));
  gatres_axpy4(acc.hi, gas, ld4(a.att_s_in + c0 + 4)); gatres_axpy4(acc.hi, gad, ld4(a.att_d_in + c0 + 4));
  return acc;
}

// ---------------------------------------------------------------------------------------------- the kernel
// ONE 1024-thread workgroup per CU.  LDS: W [M][K + 8] bf16 in proj_bf16_tile_kernel's permuted layout (staged once), the
// next convolution's attention vectors, and a DOUBLE-BUFFERED tile of the phase's rows.
// Sparse stage: K / 8 lanes per row, RW = 64 / (K / 8) rows per wave at a time; a wave gathers TWO such row sets per phase with
// the loads of both in flight together (a CU holds ~200 rows of a C-Town batch of 128: what a phase costs is one memory round
// trip, so rows per round trip is what counts), 16 waves -> a PHASE is 64 (K = 256) or 128 (K = 128) rows = RT = 4 or 8 row tiles.
// The next phase's row pointers are requested behind this phase's row loads, its neighbour ids behind this phase's arithmetic.
// Dense stage (behind ONE barrier per phase): wave w < RT takes row tile w through proj_bf16_tile_kernel's own inner loop --
// B fragments from the LDS tile instead of HBM, passes of 128 columns, the same accumulator-to-feature map, logits and
// epilogue -- so a blocked launch gives the BITS of the per-op pair; the other waves go on to the next phase's rows (the tile a
// phase writes is the one whose readers passed the barrier in between).
template <int KIND, int K, int M, int HIN, bool RELU, int EPI, int HOUT>
__global__ __launch_bounds__(1024) void blocked_kernel(const BlockedArgs a) {
  constexpr int KS = K / 32, KP = K + 8, LPR = K / 8, RW = 64 / LPR, SETROWS = 16 * RW, ROWS = 2 * SETROWS, RT = ROWS / 16;
  constexpr int WC = 128, C = M / HOUT;
  static_assert((K == 128 || K == 256) && (M == 128 || M == 256), "gatres_large shapes");
  static_assert(EPI != BE_ATT || C == 128, "a pass of 128 columns is one head");
  __shared__ __attribute__((aligned(16))) gatres_bf16 wl[M * KP];
  __shared__ __attribute__((aligned(16))) gatres_bf16 xt[2][ROWS * KP];
  __shared__ __attribute__((aligned(16))) float attl[2 * M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int N = a.N, nph = (N + ROWS - 1) / ROWS;
  const int wb = gatres_xcd_block(blockIdx.x, gridDim.x);
  const int p_lo = (int)((long long)nph * wb / gridDim.x), p_hi = (int)((long long)nph * (wb + 1) / gridDim.x);
  if (p_lo >= p_hi) return;
  const int lr = lane & (LPR - 1);
  const int rr0 = wave * RW + lane / LPR, rr1 = SETROWS + rr0;          // this lane's two rows within a phase
  using Idx = std::conditional_t<KIND == BK_GAT_FWD, IdxGat, std::conditional_t<KIND == BK_MEAN_FWD, IdxMean, IdxSrc>>;
  auto idx1 = [&](int row, Idx& x) {
    if constexpr (KIND == BK_GAT_FWD) bk_idx1_gat<K, HIN>(a, row, lr, x);
    else if constexpr (KIND == BK_MEAN_FWD) bk_idx1_mean(a, row, x);
    else bk_idx1_src<K, HIN>(a, row, lr, x);
  };
  auto idx2 = [&](int row, Idx& x) {
    if constexpr (KIND == BK_GAT_FWD) bk_idx2_gat<K, HIN>(a, lr, x);
    else if constexpr (KIND == BK_MEAN_FWD) bk_idx2_mean(a, row, x);
    else bk_idx2_src(a, row, x);
  };
  auto row_of = [&](int ph, int rr) { return min(min(ph, p_hi - 1) * ROWS + rr, N - 1); };
  Idx c0, c1, n0, n1;
  idx1(row_of(p_lo, rr0), c0);
  idx1(row_of(p_lo, rr1), c1);
  {
    // W -> LDS, proj_bf16_tile_kernel's image: LDS row l = 128 p + 16 t + a holds W row 128 p + 32 (t >> 1) + 8 (a >> 2) +
    // 4 (t & 1) + (a & 3); four 16-byte loads in flight per thread, then their LDS stores
    constexpr int CH = M * (K / 8), PER = CH / 1024;
    static_assert(CH % 1024 == 0 && PER <= 8, "W chunks per thread");
    uint4 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = threadIdx.x + 1024 * u;
      const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
      const int lw = l % WC, t = lw >> 4, aa = lw & 15;
      const int m = (l / WC) * WC + 32 * (t >> 1) + 8 * (aa >> 2) + 4 * (t & 1) + (aa & 3);
      v[u] = *reinterpret_cast<const uint4*>(a.W + (size_t)m * K + k8);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = threadIdx.x + 1024 * u;
      const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
      *reinterpret_cast<uint4*>(wl + l * KP + k8) = v[u];
    }
    if constexpr (EPI == BE_ATT) {
      if ((int)threadIdx.x < 2 * (M / 4)) {
        const int which = (int)threadIdx.x / (M / 4), c4 = ((int)threadIdx.x % (M / 4)) * 4;
        st4(attl + which * M + c4, ld4((which ? a.att_d_out : a.att_s_out) + c4));
      }
    }
  }
  idx2(row_of(p_lo, rr0), c0);
  idx2(row_of(p_lo, rr1), c1);
  const float* attS = attl;
  const float* attD = attl + M;
  for (int ph = p_lo; ph < p_hi; ++ph) {
    const int row0 = ph * ROWS, buf = (ph - p_lo) & 1;
    gatres_bf16* xtb = xt[buf];
    // ---- sparse stage: both row sets of the phase -> HBM and the LDS tile
    const int r0 = row_of(ph, rr0), r1 = row_of(ph, rr1), nr0 = row_of(ph + 1, rr0), nr1 = row_of(ph + 1, rr1);
    const bool valid0 = row0 + rr0 < N, valid1 = row0 + rr1 < N;
    Row8 o0, o1;
#ifdef BK_PROBE_NO_GATHER            // (probe build, WRONG results: what a launch costs without its sparse stage)
    o0 = row8_zero(); o0.lo.x = (float)r0; o1 = o0; n0 = c0; n1 = c1;
#else
    {
      auto last = [&] { idx1(nr0, n0); idx1(nr1, n1); };          // (behind the issue of BOTH sets' row loads)
      auto second = [&] {
        if constexpr (KIND == BK_GAT_FWD) o1 = bk_gat_fwd<K, HIN, RELU>(a, c1, valid1, lr, last);
        else if constexpr (KIND == BK_MEAN_FWD) o1 = bk_mean_fwd<K>(a, c1, r1, lr, last);
        else o1 = bk_gat_bwd_src<K, HIN>(a, c1, r1, valid1, lr, last);
      };
      if constexpr (KIND == BK_GAT_FWD) o0 = bk_gat_fwd<K, HIN, RELU>(a, c0, valid0, lr, second);
      else if constexpr (KIND == BK_MEAN_FWD) o0 = bk_mean_fwd<K>(a, c0, r0, lr, second);
      else o0 = bk_gat_bwd_src<K, HIN>(a, c0, r0, valid0, lr, second);
    }
    idx2(nr0, n0);
    idx2(nr1, n1);
#endif
    {
      const bf16x8 b0 = pack8(o0), b1 = pack8(o1);
      if (valid0) *reinterpret_cast<bf16x8*>(a.mid + (size_t)r0 * K + lr * 8) = b0;
      if (valid1) *reinterpret_cast<bf16x8*>(a.mid + (size_t)r1 * K + lr * 8) = b1;
      *reinterpret_cast<bf16x8*>(xtb + rr0 * KP + lr * 8) = b0;       // (rows beyond N: a copy of the last row, never stored)
      *reinterpret_cast<bf16x8*>(xtb + rr1 * KP + lr * 8) = b1;
    }
    c0 = n0; c1 = n1;
#ifdef BK_PROBE_NO_DENSE             // (probe build, WRONG results: the sparse stage alone, phase after phase)
    continue;
#endif
    __syncthreads();
    // ---- dense stage: proj_bf16_tile_kernel's inner loop.  A work ITEM is (row tile, chunk of NTW column tiles): NTW = 8 -- one
    // head, so the logits reduce inside the wave -- for the projections with logits, 4 for the input gradients (their epilogue
    // holds a residual and a ReLU-reference row piece per pair of tiles: half the registers); items go round the 16 waves.
    constexpr int NTW = EPI == BE_ATT ? 8 : 4, CHUNKS = M / (16 * NTW), ITEMS = RT * CHUNKS;
    for (int it = wave; it < ITEMS; it += 16) {
      const int rt = it / CHUNKS, tb = (it % CHUNKS) * NTW;            // first column tile of the chunk
      const int n = row0 + rt * 16 + i;
      const bool nok = n < N;
      bf16x8 xf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xtb + (rt * 16 + i) * KP + s * 32 + q * 8);
      const size_t rowo = (size_t)min(n, N - 1) * M + 16 * tb + q * 8;
      const gatres_bf16* wbase = wl + (size_t)(16 * tb + i) * KP + q * 8;
      uint4 rraw[EPI == BE_RESID_MASK ? NTW / 2 : 1], mraw[EPI == BE_RESID_MASK ? NTW / 2 : 1];
      if constexpr (EPI == BE_RESID_MASK) {
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
          if (a.resid) rraw[t / 2] = *reinterpret_cast<const uint4*>(a.resid + rowo + 16 * t);
          if (a.relu_ref) mraw[t / 2] = *reinterpret_cast<const uint4*>(a.relu_ref + rowo + 16 * t);
        }
      }
      f32x4 acc[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 wf[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(t * 16) * KP + s * 32);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[s], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[t][reg] = output feature 16 (tb + t - ((tb + t) & 1)) + 8 q + 4 ((tb + t) & 1) + reg of row n  (tb is even)
      if constexpr (EPI == BE_ATT) {
        const int hd = tb / 8;                           // (a chunk is one head)
        float ps = 0.f, pd = 0.f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int mb = 16 * tb + 32 * (t >> 1) + 8 * q + 4 * (t & 1);
          const float4 as = ld4(attS + mb), ad = ld4(attD + mb);
          ps += fmaf(acc[t][3], as.w, fmaf(acc[t][2], as.z, fmaf(acc[t][1], as.y, acc[t][0] * as.x)));
          pd += fmaf(acc[t][3], ad.w, fmaf(acc[t][2], ad.z, fmaf(acc[t][1], ad.y, acc[t][0] * ad.x)));
        }
        ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
        pd += __shfl_xor(pd, 16); pd += __shfl_xor(pd, 32);
        if (q == 0 && nok) { a.a_src_out[n * HOUT + hd] = ps; a.a_dst_out[n * HOUT + hd] = pd; }
      }
      if (nok) {
#pragma unroll
        for (int t = 0; t < NTW; t += 2) {
          Row8 ov;
          ov.lo = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
          ov.hi = make_float4(acc[t + 1][0], acc[t + 1][1], acc[t + 1][2], acc[t + 1][3]);
          if constexpr (EPI == BE_RESID_MASK) {
            if (a.resid) { add4(ov.lo, widen_lo(rraw[t / 2])); add4(ov.hi, widen_hi(rraw[t / 2])); }
            if (a.relu_ref) {
              const float4 ra = widen_lo(mraw[t / 2]), rb = widen_hi(mraw[t / 2]);
              ov.lo.x = ra.x > 0.f ? ov.lo.x : 0.f; ov.lo.y = ra.y > 0.f ? ov.lo.y : 0.f;
              ov.lo.z = ra.z > 0.f ? ov.lo.z : 0.f; ov.lo.w = ra.w > 0.f ? ov.lo.w : 0.f;
              ov.hi.x = rb.x > 0.f ? ov.hi.x : 0.f; ov.hi.y = rb.y > 0.f ? ov.hi.y : 0.f;
              ov.hi.z = rb.z > 0.f ? ov.hi.z : 0.f; ov.hi.w = rb.w > 0.f ? ov.hi.w : 0.f;
            }
          }
          *reinterpret_cast<bf16x8*>(a.out + (size_t)n * M + 16 * (tb + t) + q * 8) = pack8(ov);
        }
      }
    }
  }
}

static inline bool blocked_shape_ok(const gatres_graph_t* g, int nc) {
  return g && nc == 128 && (g->flags & GATRES_GRAPH_DEG_LE32) && g->num_nodes >= 64 &&
         (long long)(g->num_nodes + 2) * 256 * 4 < (1LL << 31) && (long long)(g->num_edges_gat + 2) * 2 * 4 < (1LL << 31);
}

static inline unsigned blocked_grid(int N) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int phases = (N + 127) / 128;                          // one workgroup per CU; never more workgroups than 128-row phases
  return (unsigned)(phases < cus ? phases : cus);
}

}  // namespace

extern "C" int gatres_blocked_supported(const gatres_model_t* m, const gatres_graph_t* g) {
  return (m && m->act_dtype == GATRES_DTYPE_BF16 && blocked_shape_ok(g, m->nc) && gatres_knobs()->blocked) ? 1 : 0;
}

// conv1's aggregation (H = 2, +bias, ReLU) -> o1, alpha1; conv2's projection of o1 -> h2, a_src2, a_dst2
extern "C" int gatres_bf16_agg_proj_fwd(const gatres_graph_t* g, const void* h, const float* a_src, const float* a_dst,
                                        const float* bias, void* o, float* alpha, const void* W_next,
                                        const float* att_src_next, const float* att_dst_next, void* h_next,
                                        float* a_src_next, float* a_dst_next, int32_t nc, void* stream) {
  if (!g || !h || !a_src || !a_dst || !bias || !o || !alpha || !W_next || !att_src_next || !att_dst_next || !h_next ||
      !a_src_next || !a_dst_next)
    return GATRES_E_BADARG;
  if (!gatres_aligned16(h) || !gatres_aligned16(o) || !gatres_aligned16(W_next) || !gatres_aligned16(h_next) ||
      !gatres_aligned16(bias) || !gatres_aligned16(att_src_next) || !gatres_aligned16(att_dst_next))
    return GATRES_E_BADARG;
  if (!blocked_shape_ok(g, nc)) return GATRES_E_UNSUPPORTED;
  BlockedArgs a{};
  a.rowptr = g->rowptr; a.col = g->col; a.src = (const gatres_bf16*)h; a.a_src = a_src; a.a_dst = a_dst; a.bias = bias;
  a.alpha = alpha; a.mid = (gatres_bf16*)o; a.W = (const gatres_bf16*)W_next; a.out = (gatres_bf16*)h_next;
  a.att_s_out = att_src_next; a.att_d_out = att_dst_next; a.a_src_out = a_src_next; a.a_dst_out = a_dst_next; a.N = g->num_nodes;
  hipLaunchKernelGGL((blocked_kernel<BK_GAT_FWD, 256, 128, 2, true, BE_ATT, 1>), dim3(blocked_grid(a.N)), dim3(1024), 0,
                     gatres_stream(stream), a);
  return gatres_launch_status();
}

// K3 (mean of y2 over the in-neighbours + residual x0, ReLU) -> x_next; the next block's conv1 projection -> h1, a_src1, a_dst1
extern "C" int gatres_bf16_mean_proj_fwd(const gatres_graph_t* g, const void* y, const void* x0, void* x_next,
                                         const void* W_next, const float* att_src_next, const float* att_dst_next,
                                         void* h_next, float* a_src_next, float* a_dst_next, int32_t nc, void* stream) {
  if (!g || !y || !x0 || !x_next || !W_next || !att_src_next || !att_dst_next || !h_next || !a_src_next || !a_dst_next)
    return GATRES_E_BADARG;
  if (!gatres_aligned16(y) || !gatres_aligned16(x0) || !gatres_aligned16(x_next) || !gatres_aligned16(W_next) ||
      !gatres_aligned16(h_next) || !gatres_aligned16(att_src_next) || !gatres_aligned16(att_dst_next))
    return GATRES_E_BADARG;
  if (!blocked_shape_ok(g, nc)) return GATRES_E_UNSUPPORTED;
  BlockedArgs a{};
  a.rowptr = g->m_rowptr; a.col = g->m_col; a.src = (const gatres_bf16*)y; a.x0 = (const gatres_bf16*)x0;
  a.mid = (gatres_bf16*)x_next; a.W = (const gatres_bf16*)W_next; a.out = (gatres_bf16*)h_next;
  a.att_s_out = att_src_next; a.att_d_out = att_dst_next; a.a_src_out = a_src_next; a.a_dst_out = a_dst_next; a.N = g->num_nodes;
  hipLaunchKernelGGL((blocked_kernel<BK_MEAN_FWD, 128, 256, 1, false, BE_ATT, 2>), dim3(blocked_grid(a.N)), dim3(1024), 0,
                     gatres_stream(stream), a);
  return gatres_launch_status();
}

// a convolution's source-major backward (g_h, g_a_src) and its input gradient  g_x = relu_mask(g_h W + resid):
// H = 1: conv2 (g_h [N, nc] -> g_x [N, 2nc]);  H = 2: conv1 (g_h [N, 2nc] -> g_x [N, nc])
extern "C" int gatres_bf16_src_dx_bwd(const gatres_graph_t* g, const void* g_out, const float* alpha, const float* g_e,
                                      const float* g_a_dst, const float* att_src, const float* att_dst, void* g_h,
                                      float* g_a_src, const void* Wt, const void* resid, const void* relu_ref, void* g_x,
                                      int32_t H, int32_t nc, void* stream) {
  if (!g || !g_out || !alpha || !g_e || !g_a_dst || !att_src || !att_dst || !g_h || !g_a_src || !Wt || !g_x)
    return GATRES_E_BADARG;
  if (!gatres_aligned16(g_out) || !gatres_aligned16(g_h) || !gatres_aligned16(att_src) || !gatres_aligned16(att_dst) ||
      !gatres_aligned16(Wt) || !gatres_aligned16(resid) || !gatres_aligned16(relu_ref) || !gatres_aligned16(g_x))
    return GATRES_E_BADARG;
  if (!blocked_shape_ok(g, nc) || (H != 1 && H != 2)) return GATRES_E_UNSUPPORTED;
  BlockedArgs a{};
  a.rowptr = g->t_rowptr; a.col = g->t_dst; a.eid = g->t_eid; a.src = (const gatres_bf16*)g_out; a.alpha = const_cast<float*>(alpha);
  a.g_e = g_e; a.g_a_dst = g_a_dst; a.att_s_in = att_src; a.att_d_in = att_dst; a.g_a_src = g_a_src; a.mid = (gatres_bf16*)g_h;
  a.W = (const gatres_bf16*)Wt; a.out = (gatres_bf16*)g_x; a.resid = (const gatres_bf16*)resid;
  a.relu_ref = (const gatres_bf16*)relu_ref; a.N = g->num_nodes;
  if (H == 1)
    hipLaunchKernelGGL((blocked_kernel<BK_GAT_BWD_SRC, 128, 256, 1, false, BE_RESID_MASK, 1>), dim3(blocked_grid(a.N)), dim3(1024), 0,
                       gatres_stream(stream), a);
  else
    hipLaunchKernelGGL((blocked_kernel<BK_GAT_BWD_SRC, 256, 128, 2, false, BE_RESID_MASK, 1>), dim3(blocked_grid(a.N)), dim3(1024), 0,
                       gatres_stream(stream), a);
  return gatres_launch_status();
}
