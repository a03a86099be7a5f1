// Blocked kernels for wide models in bf16 storage (gatres_large, BASELINE configs 3 and 5): a SPARSE stage and the dense
// projection that consumes its output, one launch.
//
// The per-op path runs a block of GResBlockMeanConv (GraphModels.py:462-468) as 14 launches of 7 - 18 us, and every one of
// them is mostly its own latency chain: ~3 us from the graph node's start to the first instruction, 4 - 7 us until the first
// MFMA of a projection (every workgroup staging W), the matrix work itself 2 - 3 us (DESIGN.md section 3.3,
// profiles/r03_proj_probe.txt).  The projections' inputs are exactly the row blocks the sparse kernels in front of them
// produce, so here a 512-thread workgroup
//   1. aggregates a block of 32 destination (or source) rows with the sparse kernels' own slot arithmetic (k_aggregate.hip:
//      same operands, same order -- the same bits), writes them to HBM (later launches need them: weight gradients, ReLU
//      masks) AND into an LDS tile,
//   2. runs the block through the matrix cores against a slice of W that its waves hold in REGISTERS for the whole launch
//      (32 VGPRs per wave: loaded once, in the shadow of the first block's gathers; no LDS copy of W, so two workgroups share
//      a CU and their phases interleave), k steps in the projection kernels' order,
//   3. passes the result through an fp32 LDS tile so that every row leaves in 16-byte pieces; the next convolution's attention
//      logits are summed from that tile in proj_bf16_tile_kernel's association: a blocked launch gives the BITS of the two
//      per-op launches it replaces.
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

#ifdef BK_STAMPS       // (probe build: in-kernel time stamps of one workgroup's wave 0, tests/micro/blocked_probe.py --stamps)
__device__ unsigned long long g_bk_stamps[4096];
__device__ __forceinline__ unsigned long long bk_time(float dep) {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  return t;
}
#define BKSTAMP(k, dep) do { if (stamp_on) g_bk_stamps[stamp_base + (k)] = bk_time(dep); } while (0)
#else
#define BKSTAMP(k, dep) do {} while (0)
#endif
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
// 512 threads = 8 waves, two workgroups per CU.  K / 8 lanes per row, so a wave gathers RW = 2 (K = 256) or 4 (K = 128) rows at
// a time and a PHASE is the 8 RW = 16 or 32 rows of one such pass: RT = 1 or 2 row tiles of 16.  MFMA work of a phase: RT row
// tiles x M / 16 column tiles; wave w owns CTW = M / 128 column tiles for every row tile -- its W fragments (CTW x K / 32 x 16
// bytes per lane = 32 VGPRs) stay in registers from the start of the launch.  A workgroup takes a contiguous share of the phases
// (XCD-contiguous: neighbour rows lie near their own).
//
// A workgroup's life is a CHAIN of phases, and what a phase costs is latency, not throughput (a CU has ~200 rows per launch):
//   * three levels of loads are in flight at the top of a phase -- this phase's neighbour rows, the next phase's neighbour ids,
//     the row pointers of the phase after that -- so a phase waits for ONE memory round trip (it was three);
//   * ONE barrier per phase: the LDS row tile and the logit partials are double-buffered, so the tile a phase writes is the
//     one its slowest reader left two barriers ago;
//   * the B operands of a phase's whole k loop are read from LDS in one batch in front of the MFMA chain;
//   * results leave from the accumulators (no output tile): with M = 256 the A operand's rows are permuted so that a lane's two
//     column tiles hold 8 CONSECUTIVE features -- feature 32 w + 8 q + 4 c + r for wave w, lane group q, tile c, register r --
//     one 16-byte store per lane and row, 64 contiguous bytes per row and wave (proj_bf16_tile_kernel's map); M = 128: 8 bytes;
//   * the next convolution's attention logits in proj_bf16_tile_kernel's association (so a blocked launch gives the bits of
//     the per-op pair): every lane leaves its 4-term fma dots in LDS, and one phase LATER (behind that phase's barrier) a
//     thread per (row, head, quarter) adds a quarter's eight dots in that kernel's order, the quarters as (q0 + q1) + (q2 + q3).
template <int KIND, int K, int M, int HIN, bool RELU, int EPI, int HOUT>
__global__ __launch_bounds__(512, 4) void blocked_kernel(const BlockedArgs a) {
  constexpr int KS = K / 32, KP = K + 8, LPR = K / 8, RW = 64 / LPR, ROWS = 8 * RW, RT = ROWS / 16, CTW = M / 128, NE = 32 * CTW;
  static_assert((K == 128 || K == 256) && (M == 128 || M == 256), "gatres_large shapes");
  static_assert(EPI != BE_ATT || M / HOUT == 128, "heads of 128 columns");
  __shared__ __attribute__((aligned(16))) gatres_bf16 xt[2][ROWS * KP];            // a phase's rows (MFMA B operand), double-buffered
  __shared__ float dl[EPI == BE_ATT ? 2 : 1][2][EPI == BE_ATT ? ROWS * NE : 1];    // BE_ATT: [buffer][src | dst][row][dot]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int N = a.N, nph = (N + ROWS - 1) / ROWS;
  const int wb = gatres_xcd_block(blockIdx.x, gridDim.x);
  const int p_lo = (int)((long long)nph * wb / gridDim.x), p_hi = (int)((long long)nph * (wb + 1) / gridDim.x);
  if (p_lo >= p_hi) return;
  const int lr = lane & (LPR - 1);
  const int rr = wave * RW + lane / LPR;                  // this lane's row within a phase
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
  auto row_of = [&](int ph) { return min(min(ph, p_hi - 1) * ROWS + rr, N - 1); };
  Idx cur, nxt, nn;
  idx1(row_of(p_lo), cur);
  idx1(row_of(p_lo + 1), nxt);
  // W fragments: A operand of v_mfma_f32_16x16x32_bf16 -- lane (i, q) holds row i of the column tile, k chunk q of step s.
  // Row m of tile c is W's row for the feature the accumulators of (tile c, m) stand for (see the header).
  bf16x8 wreg[CTW][KS];
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    const int wrow = CTW == 2 ? 32 * wave + 8 * (i >> 2) + 4 * c + (i & 3) : 16 * wave + i;
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[c][s] = *reinterpret_cast<const bf16x8*>(a.W + (size_t)wrow * K + s * 32 + q * 8);
  }
  idx2(row_of(p_lo), cur);
#ifdef BK_STAMPS
  const bool stamp_on = (wb == 0 || wb == 100) && threadIdx.x == 0;
  int stamp_base = (wb == 0 ? 0 : 2048);
  if (stamp_on) g_bk_stamps[stamp_base + 1000] = bk_time(0.f);
#endif
  // the attention logits of the phase BEFORE `ph` (rows from prow0), from the dots that phase left in dl[pb]
  auto logits_of = [&](int prow0, int pb) {
    if constexpr (EPI == BE_ATT) {
      if ((int)threadIdx.x < ROWS * HOUT * 4) {
        const int qq = (int)threadIdx.x & 3, hd = ((int)threadIdx.x >> 2) % HOUT, lrr = (int)threadIdx.x / (4 * HOUT);
        const int lrow = prow0 + lrr;
        float ps = 0.f, pd = 0.f;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const int jb = tt >> 1, hh = tt & 1;
          // the dot of features 128 hd + 32 jb + 8 qq + 4 hh + [0, 4): which (wave, lane group, tile) formed it
          const int e = CTW == 2 ? ((hd * 4 + jb) * 4 + qq) * 2 + hh : (2 * jb + (qq >> 1)) * 4 + 2 * (qq & 1) + hh;
          ps += dl[pb][0][lrr * NE + e];
          pd += dl[pb][1][lrr * NE + e];
        }
        ps += __shfl_xor(ps, 1); ps += __shfl_xor(ps, 2);
        pd += __shfl_xor(pd, 1); pd += __shfl_xor(pd, 2);
        if (qq == 0 && lrow < N) { a.a_src_out[lrow * HOUT + hd] = ps; a.a_dst_out[lrow * HOUT + hd] = pd; }
      }
    }
  };
  for (int ph = p_lo; ph < p_hi; ++ph) {
    const int row0 = ph * ROWS, buf = (ph - p_lo) & 1;
    gatres_bf16* xtb = xt[buf];
    // ---- sparse stage: this phase's rows -> HBM and the LDS tile; behind the issue of their loads, the next phase's
    // neighbour ids and the row pointers of the phase after it
    const int row = row_of(ph), nrow1 = row_of(ph + 1), nrow2 = row_of(ph + 2);
    const bool valid = row0 + rr < N;
    auto after_issue = [&] { idx2(nrow1, nxt); idx1(nrow2, nn); };
    BKSTAMP(0, 0.f);
    Row8 o;
#ifdef BK_PROBE_NO_GATHER            // (probe build, WRONG results: what a launch costs without its sparse stage)
    o = row8_zero(); o.lo.x = (float)row; (void)after_issue; nn = nxt;
#else
    if constexpr (KIND == BK_GAT_FWD) o = bk_gat_fwd<K, HIN, RELU>(a, cur, valid, lr, after_issue);
    else if constexpr (KIND == BK_MEAN_FWD) o = bk_mean_fwd<K>(a, cur, row, lr, after_issue);
    else o = bk_gat_bwd_src<K, HIN>(a, cur, row, valid, lr, after_issue);
#endif
    BKSTAMP(1, o.lo.x + o.hi.w);
    const bf16x8 ob = pack8(o);
    if (valid) *reinterpret_cast<bf16x8*>(a.mid + (size_t)row * K + lr * 8) = ob;
    *reinterpret_cast<bf16x8*>(xtb + rr * KP + lr * 8) = ob;         // (rows beyond N: a copy of the last row, never stored)
    cur = nxt; nxt = nn;
#ifdef BK_PROBE_NO_DENSE             // (probe build, WRONG results: the sparse stage alone, phase after phase)
    continue;
#endif
    // the epilogue's operands (8 or 16 bytes per lane and row tile), requested in front of the barrier
    const int fb = CTW == 2 ? 32 * wave + 8 * q : 16 * wave + 4 * q;          // this lane's first feature
    uint4 rraw[RT], mraw[RT];
    if constexpr (EPI == BE_RESID_MASK) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const size_t off = (size_t)min(row0 + rt * 16 + i, N - 1) * M + fb;
        rraw[rt] = make_uint4(0u, 0u, 0u, 0u); mraw[rt] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (CTW == 2) {
          if (a.resid) rraw[rt] = ld_raw8(a.resid + off);
          if (a.relu_ref) mraw[rt] = ld_raw8(a.relu_ref + off);
        } else {
          if (a.resid) { const uint2 u = *reinterpret_cast<const uint2*>(a.resid + off); rraw[rt].x = u.x; rraw[rt].y = u.y; }
          if (a.relu_ref) { const uint2 u = *reinterpret_cast<const uint2*>(a.relu_ref + off); mraw[rt].x = u.x; mraw[rt].y = u.y; }
        }
      }
    }
    BKSTAMP(2, 0.f);
    __syncthreads();
    BKSTAMP(3, 0.f);
    if (ph > p_lo) logits_of(row0 - ROWS, buf ^ 1);
    BKSTAMP(4, 0.f);
    // ---- dense stage: every B fragment of the phase in one batch, then the MFMA chains
    bf16x8 xf[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[rt][s] = *reinterpret_cast<const bf16x8*>(xtb + (rt * 16 + i) * KP + s * 32 + q * 8);
    f32x4 acc[RT][CTW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < CTW; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[c][s], xf[rt][s], acc[rt][c], 0, 0, 0);
    // acc[rt][c][r] = output feature fb + 4 c + r of row rt 16 + i
    BKSTAMP(5, acc[0][0][0] + acc[RT - 1][CTW - 1][3]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int orow = row0 + rt * 16 + i;
      if constexpr (EPI == BE_ATT) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const float4 as = ld4(a.att_s_out + fb + 4 * c), ad = ld4(a.att_d_out + fb + 4 * c);
          const int e = CTW == 2 ? (wave * 4 + q) * 2 + c : wave * 4 + q;
          dl[buf][0][(rt * 16 + i) * NE + e] = fmaf(acc[rt][c][3], as.w, fmaf(acc[rt][c][2], as.z, fmaf(acc[rt][c][1], as.y, acc[rt][c][0] * as.x)));
          dl[buf][1][(rt * 16 + i) * NE + e] = fmaf(acc[rt][c][3], ad.w, fmaf(acc[rt][c][2], ad.z, fmaf(acc[rt][c][1], ad.y, acc[rt][c][0] * ad.x)));
        }
      }
      float4 o4[CTW];
#pragma unroll
      for (int c = 0; c < CTW; ++c) o4[c] = make_float4(acc[rt][c][0], acc[rt][c][1], acc[rt][c][2], acc[rt][c][3]);
      if constexpr (EPI == BE_RESID_MASK) {
        if (a.resid) {
          add4(o4[0], widen_lo(rraw[rt]));
          if constexpr (CTW == 2) add4(o4[1], widen_hi(rraw[rt]));
        }
        if (a.relu_ref) {
          const float4 ra = widen_lo(mraw[rt]);
          o4[0].x = ra.x > 0.f ? o4[0].x : 0.f; o4[0].y = ra.y > 0.f ? o4[0].y : 0.f;
          o4[0].z = ra.z > 0.f ? o4[0].z : 0.f; o4[0].w = ra.w > 0.f ? o4[0].w : 0.f;
          if constexpr (CTW == 2) {
            const float4 rb = widen_hi(mraw[rt]);
            o4[1].x = rb.x > 0.f ? o4[1].x : 0.f; o4[1].y = rb.y > 0.f ? o4[1].y : 0.f;
            o4[1].z = rb.z > 0.f ? o4[1].z : 0.f; o4[1].w = rb.w > 0.f ? o4[1].w : 0.f;
          }
        }
      }
      if (orow < N) {
        if constexpr (CTW == 2) {
          Row8 ov; ov.lo = o4[0]; ov.hi = o4[1];
          *reinterpret_cast<bf16x8*>(a.out + (size_t)orow * M + fb) = pack8(ov);
        } else {
          bf16x4 b4;
          b4[0] = (gatres_bf16)o4[0].x; b4[1] = (gatres_bf16)o4[0].y; b4[2] = (gatres_bf16)o4[0].z; b4[3] = (gatres_bf16)o4[0].w;
          *reinterpret_cast<bf16x4*>(a.out + (size_t)orow * M + fb) = b4;
        }
      }
    }
    BKSTAMP(6, 0.f);
#ifdef BK_STAMPS
    stamp_base += 8;
#endif
  }
#ifdef BK_STAMPS
  if (stamp_on) g_bk_stamps[(wb == 0 ? 0 : 2048) + 1001] = bk_time(0.f);
#endif
#ifndef BK_PROBE_NO_DENSE
  if constexpr (EPI == BE_ATT) {
    __syncthreads();                                      // the last phase's dots
    logits_of((p_hi - 1) * ROWS, (p_hi - 1 - p_lo) & 1);
  }
#endif
}

static inline bool blocked_shape_ok(const gatres_graph_t* g, int nc) {
  return g && nc == 128 && (g->flags & GATRES_GRAPH_DEG_LE32) && g->num_nodes >= 64 &&
         (long long)(g->num_nodes + 2) * 256 * 4 < (1LL << 31) && (long long)(g->num_edges_gat + 2) * 2 * 4 < (1LL << 31);
}

static inline unsigned blocked_grid(int N) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const int phases = (N + 31) / 32, want = 2 * cus;            // two workgroups per CU; never more workgroups than 32-row phases
  return (unsigned)(phases < want ? phases : want);
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
  hipLaunchKernelGGL((blocked_kernel<BK_GAT_FWD, 256, 128, 2, true, BE_ATT, 1>), dim3(blocked_grid(a.N)), dim3(512), 0,
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
  hipLaunchKernelGGL((blocked_kernel<BK_MEAN_FWD, 128, 256, 1, false, BE_ATT, 2>), dim3(blocked_grid(a.N)), dim3(512), 0,
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
    hipLaunchKernelGGL((blocked_kernel<BK_GAT_BWD_SRC, 128, 256, 1, false, BE_RESID_MASK, 1>), dim3(blocked_grid(a.N)), dim3(512), 0,
                       gatres_stream(stream), a);
  else
    hipLaunchKernelGGL((blocked_kernel<BK_GAT_BWD_SRC, 256, 128, 2, false, BE_RESID_MASK, 1>), dim3(blocked_grid(a.N)), dim3(512), 0,
                       gatres_stream(stream), a);
  return gatres_launch_status();
}

#ifdef BK_STAMPS
extern "C" int gatres_probe_bk_stamps(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_bk_stamps), sizeof(unsigned long long) * 4096);
}
#endif
