// Sparse message-passing kernels for gfx950 (wave64): GATConv edge scoring + per-destination softmax +
// weighted neighbour sum (K2) and SimpleConv mean + residual + ReLU (K3), forward and backward.
//
// Layout: feature rows are row-major fp32; a row of width HC is owned by G = HC/4 adjacent lanes, one float4
// per lane, so a neighbour-row gather is one 16-B load per lane and G*16 contiguous bytes per row.  A wave holds
// 64/G destination rows.  WDN graphs have in-degree ~2-5, so the reduction over a row's edges is sequential per
// lane in CSR order (= the order PyG's scatter visits them: original edges first, self loop last); there are no
// atomics anywhere and results are bitwise reproducible.
//
// Reference semantics restated (torch_geometric >= 2.3, call sites GraphModels.py:464-466):
//   s_e   = LeakyReLU_0.2(a_src[j] + a_dst[i])                 GATConv.edge_update
//   m_i   = max_e s_e;  p_e = exp(s_e - m_i);  Z_i = sum p_e + 1e-16;  alpha_e = p_e / Z_i     utils.softmax
//   out_i = sum_e alpha_e * h[j] + bias                        GATConv.message / aggregate(sum) / bias
//   mean  : out_i = (sum_{j->i} y[j]) / max(indeg(i), 1)       SimpleConv(aggr="mean")
#include "gatres_common.h"
#include <type_traits>

namespace {

struct RowGeom {
  int HC, C, lgC, H, G, lgG;
};

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static inline bool make_geom(int H, int C, RowGeom* g, int W = 4) {      // W = features per lane: a row has G = HC / W lanes
  if (H < 1 || C < W || !gatres_is_pow2(C) || !gatres_is_pow2(H)) return false;
  int HC = H * C;
  if (HC > 256) return false;
  g->HC = HC; g->C = C; g->lgC = ilog2(C); g->H = H;
  g->G = HC / W; g->lgG = ilog2(g->G);
  return true;
}

// Features per lane W (a row of `row_width` features has row_width / W lanes, a head of C features C / W of them).  The lanes
// of a row all repeat its scalar work -- index / logit / coefficient loads, softmax, address arithmetic -- and that, not
// bytes, bounds these kernels at model widths: rows of 64 features and more take 8 features per lane (bf16: one
// 16-byte access, fp32: two).  Measured, gatres_large bs 128: bf16 7.51 -> 6.72 ms/step, fp32 13.96 -> 13.67; 16 per lane
// gives nothing more (6.72) and is not instantiated.  GATRES_AGG_LANE_FEATURES=4|8 overrides the choice.
static inline int lane_features(int row_width, int C) {
  int w = row_width >= 64 ? 8 : 4;
  if (const int v = gatres_knobs()->agg_lane_features) w = v;
  while (w > 4 && (C < w || row_width < 4 * w)) w >>= 1;       // at least one lane per head, four per row
  return w;
}

// ------------------------------------------------------------------------------------------------------
// Hub rows: wavefront-level segmented reductions with LDS-staged partial sums.
// A destination (or, in the source-major pass, a source) with more edges than the slot paths take is not walked edge by
// edge by its G lanes: the WHOLE wave turns to it.  The 64 lanes form S = 64/G edge slots of G feature lanes; slot k
// takes edges beg + k, beg + k + S, ...; every reduction over the row's edges (softmax max, softmax sum, weighted
// neighbour sum, ...) is formed as S partial results that are staged in LDS (hub_lds: one [S][4G] float tile per wave)
// and summed in slot order by the row's own lanes.  Deterministic; the association differs from the edge-by-edge order
// (fp32 reassociation only).  Water networks never take this path (degree <= 6); power-law graphs do, for rows with more than HUB_MIN_DEGREE edges.
// ------------------------------------------------------------------------------------------------------
constexpr int HUB_FLOATS_PER_WAVE = 64 * 4;
constexpr int HUB_MIN_DEGREE = 32;             // up to here a row's own G lanes walk its edges (4 to 16 rows per wave in flight)
__device__ __forceinline__ float* hub_tile(float* hub_lds) { return hub_lds + (threadIdx.x >> 6) * HUB_FLOATS_PER_WAVE; }
// one scalar per lane -> the row's value for head-feature lane f: combine(slot 0, slot 1, ...) in slot order
template <bool MAX>
__device__ __forceinline__ float hub_reduce1(float v, float* tile, int G, int S, int f) {
  const int lane = threadIdx.x & 63;
  tile[lane] = v;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own LDS writes have landed
  float r = tile[f];
  for (int k = 1; k < S; ++k) r = MAX ? fmaxf(r, tile[k * G + f]) : r + tile[k * G + f];
  __builtin_amdgcn_wave_barrier();
  return r;
}
__device__ __forceinline__ float4 hub_reduce4(float4 v, float* tile, int G, int S, int f) {
  const int lane = threadIdx.x & 63;
  st4(tile + lane * 4, v);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float4 r = ld4(tile + f * 4);
  for (int k = 1; k < S; ++k) {
    const float4 t = ld4(tile + (k * G + f) * 4);
    r.x = r.x + t.x; r.y = r.y + t.y; r.z = r.z + t.z; r.w = r.w + t.w;
  }
  __builtin_amdgcn_wave_barrier();
  return r;
}

// Addressing.  Every table of these kernels is a kernel argument (a wave-uniform base) indexed by a node or edge id
// times a power-of-two row width: offsets are formed in IDX = 32-bit BYTE arithmetic with shifts -- one or two full-rate
// VALU operations and the `global_load v, voffset, s[base]` form -- instead of the 64-bit multiply-adds (quarter rate,
// five instructions per load) that `base[(size_t)j * HC + c]` compiles to and that bounded these kernels (83 % VALU busy
// on address arithmetic alone).  IDX = size_t is the instance for tables of 4 GiB and more (launchers: offsets_fit_32).
template <typename IDX, typename T>
__device__ __forceinline__ const T* gatres_at(const T* base, IDX elem) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem * (IDX)sizeof(T));
}
template <typename IDX, typename T>
__device__ __forceinline__ T* gatres_at_w(T* base, IDX elem) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + elem * (IDX)sizeof(T));
}
// rowld(tab, j, c): four features of row j from column c;  hval / hptr(tab, i, hh): per-(row or edge, head) scalars;
// ival(tab, i): an index table entry
#define GATRES_AGG_ADDRESSING(LG_ROW, LG_H)                                                                              \
  auto rowld = [&](const T* tab, int j, int c) { return ldrow4(gatres_at<IDX>(tab, ((IDX)j << (LG_ROW)) + (IDX)c)); };  \
  auto hval = [&](const float* tab, int i, int hh) { return *gatres_at<IDX>(tab, ((IDX)i << (LG_H)) + (IDX)hh); };      \
  auto hptr = [&](float* tab, int i, int hh) { return gatres_at_w<IDX>(tab, ((IDX)i << (LG_H)) + (IDX)hh); };           \
  auto ival = [&](const int* tab, int i) { return *gatres_at<IDX>(tab, (IDX)i); };                                      \
  (void)hval; (void)hptr; (void)rowld; (void)ival;

// ------------------------------------------------------------------------------------------------------
// K2 forward
// ------------------------------------------------------------------------------------------------------
// W = features per lane (4; 8 for bf16 rows of 64 features and more: one 16-byte access per lane, half the lanes per row --
// the lanes of a row all repeat its scalar work, which is what bounds these kernels)
template <bool RELU, typename T, typename IDX, int W>
__global__ __launch_bounds__(256, W == 16 ? 3 : W == 8 ? 5 : 8) void gat_aggregate_fwd_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const T* __restrict__ h,
    const float* __restrict__ a_src, const float* __restrict__ a_dst, const float* __restrict__ bias,
    T* __restrict__ out, float* __restrict__ alpha, int N, RowGeom gm) {
  __shared__ __attribute__((aligned(16))) float hub_lds[4 * HUB_FLOATS_PER_WAVE];
  constexpr int LGW = W == 16 ? 4 : W == 8 ? 3 : 2, Q = W / 4;
  const int lgHC = gm.lgG + LGW, lgH = lgHC - gm.lgC;
  GATRES_AGG_ADDRESSING(lgHC, lgH)
  auto rowldv = [&](const T* tab, int j, int c) { return ldrowv<W>(gatres_at<IDX>(tab, ((IDX)j << lgHC) + (IDX)c)); };
  auto axpyv = [&](gatres_rowv<W>& a, float s, const gatres_rowv<W>& v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) gatres_axpy4(a.v[q], s, v.v[q]);
  };
  (void)axpyv; (void)rowldv;
  const int tid = gatres_xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (XCD-local row ranges)
  int row = tid >> gm.lgG;
  const bool valid = row < N;                  // (every lane stays: hub rows are processed by the whole wave)
  if (!valid) row = N - 1;
  const int c0 = (tid & (gm.G - 1)) * W;
  const int hd = c0 >> gm.lgC;
  const bool leader = valid && (c0 & (gm.C - 1)) == 0;
  const int H = gm.H, HC = gm.HC;
  const int beg = ival(rowptr, row), end = ival(rowptr, row + 1);
  const float adst = hval(a_dst, row, hd);
  gatres_rowv<W> acc = rowv_zero<W>();
  constexpr int MAXD = 6;                      // rows with <= 6 in-edges (every water-network row): slot path
  const bool hub = valid && end - beg > HUB_MIN_DEGREE;
  const unsigned long long hubs = __ballot(hub && c0 == 0);         // first lane of every hub row in this wave
  if (__builtin_expect(hubs != 0ULL, 0)) {
    const int lane = threadIdx.x & 63, G = gm.G, S = 64 >> gm.lgG;
    const int f = lane & (G - 1), slot = lane >> gm.lgG;
    float* tile = hub_tile(hub_lds);
    unsigned long long todo = hubs;
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;                 // first lane of the hub row's lane group
      todo &= todo - 1;
      const int hrow = __shfl(row, owner), hbeg = __shfl(beg, owner), hend = __shfl(end, owner);
      const int fc0 = f * W, fhd = fc0 >> gm.lgC;
      const bool fleader = (fc0 & (gm.C - 1)) == 0;
      const float hadst = hval(a_dst, hrow, fhd);
      float m = -INFINITY;
      _Pragma("unroll 2") for (int e = hbeg + slot; e < hend; e += S) m = fmaxf(m, gatres_leaky(hval(a_src, ival(col, e), fhd) + hadst));
      m = hub_reduce1<true>(m, tile, G, S, f);
      float z = 0.f;
      _Pragma("unroll 2") for (int e = hbeg + slot; e < hend; e += S) z = z + expf(gatres_leaky(hval(a_src, ival(col, e), fhd) + hadst) - m);
      const float Z = hub_reduce1<false>(z, tile, G, S, f) + GATRES_SOFTMAX_EPS;
      gatres_rowv<W> part = rowv_zero<W>();
      _Pragma("unroll 2") for (int e = hbeg + slot; e < hend; e += S) {
        const int j = ival(col, e);
        const float al = expf(gatres_leaky(hval(a_src, j, fhd) + hadst) - m) / Z;
        if (fleader) *hptr(alpha, e, fhd) = al;
        axpyv(part, al, rowldv(h, j, fc0));
      }
      gatres_rowv<W> sum;
      _Pragma("unroll") for (int q = 0; q < Q; ++q) sum.v[q] = hub_reduce4(part.v[q], tile, G, S, f);
      if ((lane >> gm.lgG) == (owner >> gm.lgG)) acc = sum;            // the row's own lanes keep it (f == their feature lane)
    }
  }
  if (hub) {
    // (acc was formed above)
  } else if (__builtin_expect(end - beg <= MAXD, 1)) {
    // every neighbour index, logit and feature row of this destination is requested at once: one dependent round
    // trip (col -> {a_src, h}) instead of three passes of edge-at-a-time chains.  Statement for statement the
    // arithmetic of the loop form below (and of the fused kernels' seg_softmax / seg_gather): bit-identical.
    // (slot count: 4 when every row of the wave has <= 4 in-edges, else MAXD; padding slots only ever add exact zeros)
    const int deg = end - beg;
    auto slots = [&](auto KC) {
    constexpr int MAXD = decltype(KC)::value;
    int jj[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) jj[k] = ival(col, beg + min(k, deg - 1));
    float so[MAXD];
    gatres_rowv<W> v[MAXD];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      v[k] = rowldv(h, jj[k], c0);
      const float sv = gatres_leaky(hval(a_src, jj[k], hd) + adst);
      so[k] = k < deg ? sv : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < MAXD; ++k) m = fmaxf(m, so[k]);
    float Z = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      so[k] = expf(so[k] - m);                 // exp(-inf) = 0 on padding slots
      Z = Z + so[k];
    }
    Z = Z + GATRES_SOFTMAX_EPS;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const float al = so[k] / Z;
      if (k < deg) {
        if (leader) *hptr(alpha, (beg + k), hd) = al;
        axpyv(acc, al, v[k]);
      }
    }
    };
    // The same, one edge slot per LANE of the head's lane group (heads of 8 or 16 lanes: 64 / 128 features at 8 per lane):
    // a lane forms its slot's logit, ONE exp and ONE divide; the maximum runs over the group by DPP (exact in any order),
    // the exponentials and then the coefficients go round the group by ds_swizzle, and every lane adds them up in CSR
    // order -- bit for bit the sums of the form above, at a sixth of its exp / divide work (k_window_stages.h: win_fwd_agg).
    auto slots_lane = [&](auto KC, auto LHC) {
      constexpr int MAXD = decltype(KC)::value, LH = decltype(LHC)::value;
      const int kk = (c0 >> LGW) & (LH - 1);                      // this lane's place in its head's lane group = its slot
      int jj[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) jj[k] = ival(col, beg + min(k, deg - 1));
      const int jm = ival(col, beg + min(min(kk, MAXD - 1), deg - 1));
      gatres_rowv<W> v[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) v[k] = rowldv(h, jj[k], c0);
      const float sv = gatres_leaky(hval(a_src, jm, hd) + adst);
      const float so = kk < deg ? sv : -INFINITY;                 // (deg <= MAXD: slots beyond it are padding)
      float m = so;
      m = fmaxf(m, gatres_dpp<0xB1>(m));                          // quad_perm [1,0,3,2]
      m = fmaxf(m, gatres_dpp<0x4E>(m));                          // quad_perm [2,3,0,1]
      m = fmaxf(m, gatres_dpp<0x141>(m));                         // row_half_mirror
      if constexpr (LH == 16) m = fmaxf(m, gatres_dpp<0x140>(m)); // row_mirror
      const float ex = expf(so - m);                              // exp(-inf) = 0 on padding slots
      auto bcast = [&](float x, auto K) {                         // lane K of the lane group (32-lane swizzle: and | or << 5)
        constexpr int pat = (0x1f & ~(LH - 1)) | (decltype(K)::value << 5);
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), pat));
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
      if (valid && kk < deg) *hptr(alpha, (beg + kk), hd) = alm;
      float al[MAXD];
      al[0] = bcast(alm, std::integral_constant<int, 0>{}); al[1] = bcast(alm, std::integral_constant<int, 1>{});
      al[2] = bcast(alm, std::integral_constant<int, 2>{}); al[3] = bcast(alm, std::integral_constant<int, 3>{});
      if constexpr (MAXD > 4) { al[4] = bcast(alm, std::integral_constant<int, 4>{}); al[5] = bcast(alm, std::integral_constant<int, 5>{}); }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < deg) axpyv(acc, al[k], v[k]);
    };
    const int lgLH = gm.lgC - LGW;
    const bool few = __ballot(deg > 4) == 0ULL;
    if (lgLH == 4) {
      if (few) slots_lane(std::integral_constant<int, 4>{}, std::integral_constant<int, 16>{});
      else slots_lane(std::integral_constant<int, MAXD>{}, std::integral_constant<int, 16>{});
    } else if (lgLH == 3) {
      if (few) slots_lane(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});
      else slots_lane(std::integral_constant<int, MAXD>{}, std::integral_constant<int, 8>{});
    } else if (few) slots(std::integral_constant<int, 4>{});
    else slots(std::integral_constant<int, MAXD>{});
  } else {                                     // up to HUB_MIN_DEGREE in-edges: the row's own lanes, edge after edge
  float m = -INFINITY;
  for (int e = beg; e < end; ++e) {
    const float s = gatres_leaky(hval(a_src, ival(col, e), hd) + adst);
    m = fmaxf(m, s);
  }
  float Z = 0.f;
  for (int e = beg; e < end; ++e) {
    const float s = gatres_leaky(hval(a_src, ival(col, e), hd) + adst);
    Z = Z + expf(s - m);
  }
  Z = Z + GATRES_SOFTMAX_EPS;

  int e = beg;
  // two edges per trip so both neighbour rows are in flight together
  for (; e + 1 < end; e += 2) {
    const int j0 = ival(col, e), j1 = ival(col, e + 1);
    const gatres_rowv<W> v0 = rowldv(h, j0, c0);
    const gatres_rowv<W> v1 = rowldv(h, j1, c0);
    const float al0 = expf(gatres_leaky(hval(a_src, j0, hd) + adst) - m) / Z;
    const float al1 = expf(gatres_leaky(hval(a_src, j1, hd) + adst) - m) / Z;
    if (leader) { *hptr(alpha, e, hd) = al0; *hptr(alpha, (e + 1), hd) = al1; }
    axpyv(acc, al0, v0);
    axpyv(acc, al1, v1);
  }
  if (e < end) {
    const int j0 = ival(col, e);
    const gatres_rowv<W> v0 = rowldv(h, j0, c0);
    const float al0 = expf(gatres_leaky(hval(a_src, j0, hd) + adst) - m) / Z;
    if (leader) *hptr(alpha, e, hd) = al0;
    axpyv(acc, al0, v0);
  }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float4 b = ld4(bias + c0 + 4 * q);
    float4& a = acc.v[q];
    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
    if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  }
  if (valid) strowv<W>(gatres_at_w<IDX>(out, ((IDX)row << lgHC) + (IDX)c0), acc);
}

// ------------------------------------------------------------------------------------------------------
// K2 backward, destination-major: per in-edge e = (j -> i)
//   ga_e  = <g_out[i,h,:], h[j,h,:]>            (reduced over the C/4 lanes of the head)
//   S_i   = sum_e alpha_e * ga_e
//   gs_e  = alpha_e * (ga_e - S_i)              softmax backward (max is detached, eps is a constant)
//   g_e   = gs_e * (raw_e > 0 ? 1 : 0.2)        LeakyReLU backward,  raw_e = a_src[j] + a_dst[i]
//   g_a_dst[i] = sum_e g_e
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float head_reduce_any(float d, int lanes_per_head) {
  switch (lanes_per_head) {                      // wave-uniform
    case 1: return d;
    case 2: return gatres_head_reduce<2>(d);
    case 4: return gatres_head_reduce<4>(d);
    case 8: return gatres_head_reduce<8>(d);
    case 16: return gatres_head_reduce<16>(d);
    default: return gatres_head_reduce<32>(d);
  }
}

// LHT: the lanes per head (C / W) as a compile-time constant for the model widths (8, 16, 32), 0 = any (wave-uniform switch
// per dot: ~40 scalar branches per row)
template <typename T, typename IDX, int W, int LHT>
__global__ __launch_bounds__(256, W == 16 ? 2 : W == 8 ? 4 : 7) void gat_aggregate_bwd_dst_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const T* __restrict__ g_out,
    const T* __restrict__ h, const float* __restrict__ alpha, const float* __restrict__ a_src,
    const float* __restrict__ a_dst, float* __restrict__ g_e, float* __restrict__ g_a_dst, int N, RowGeom gm) {
  __shared__ __attribute__((aligned(16))) float hub_lds[4 * HUB_FLOATS_PER_WAVE];
  constexpr int LGW = W == 16 ? 4 : W == 8 ? 3 : 2, Q = W / 4;
  const int lgHC = gm.lgG + LGW, lgH = lgHC - gm.lgC;
  GATRES_AGG_ADDRESSING(lgHC, lgH)
  auto rowldv = [&](const T* tab, int j, int c) { return ldrowv<W>(gatres_at<IDX>(tab, ((IDX)j << lgHC) + (IDX)c)); };
  auto axpyv = [&](gatres_rowv<W>& a, float s, const gatres_rowv<W>& v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) gatres_axpy4(a.v[q], s, v.v[q]);
  };
  (void)axpyv; (void)rowldv;
  const int tid = gatres_xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (XCD-local row ranges)
  int row = tid >> gm.lgG;
  const bool valid = row < N;
  if (!valid) row = N - 1;              // keep every lane alive for the shuffles; stores are predicated
  const int c0 = (tid & (gm.G - 1)) * W;
  const int hd = c0 >> gm.lgC;
  const bool leader = valid && (c0 & (gm.C - 1)) == 0;
  const int H = gm.H, HC = gm.HC, LH = gm.C >> LGW;      // lanes per head
  auto hdot = [&](const gatres_rowv<W>& a, const gatres_rowv<W>& b) {
    float d = gatres_head_dot4(a.v[0], b.v[0]);
#pragma unroll
    for (int q = 1; q < Q; ++q) {
      d = fmaf(a.v[q].x, b.v[q].x, d); d = fmaf(a.v[q].y, b.v[q].y, d);
      d = fmaf(a.v[q].z, b.v[q].z, d); d = fmaf(a.v[q].w, b.v[q].w, d);
    }
    if constexpr (LHT != 0) return gatres_head_reduce<LHT>(d);
    else return head_reduce_any(d, LH);
  };
  const int beg = ival(rowptr, row), end = ival(rowptr, row + 1);
  float S = 0.f, gad = 0.f;
  // hub rows: the whole wave, S edge slots of G lanes, partial sums through LDS (see hub_reduce1)
  const unsigned long long hubs = __ballot(valid && end - beg > HUB_MIN_DEGREE && c0 == 0);
  if (__builtin_expect(hubs != 0ULL, 0)) {
    const int lane = threadIdx.x & 63, G = gm.G, S = 64 >> gm.lgG;
    const int f = lane & (G - 1), slot = lane >> gm.lgG;
    float* tile = hub_tile(hub_lds);
    unsigned long long todo = hubs;
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int hrow = __shfl(row, owner), hbeg = __shfl(beg, owner), hend = __shfl(end, owner);
      const int fc0 = f * W, fhd = fc0 >> gm.lgC;
      const bool fleader = (fc0 & (gm.C - 1)) == 0;
      const gatres_rowv<W> hgo = rowldv(g_out, hrow, fc0);
      const float hadst = hval(a_dst, hrow, fhd);
      float sp = 0.f;
      for (int e = hbeg + slot; e < hend; e += S) {
        const float ga = hdot(hgo, rowldv(h, ival(col, e), fc0));
        sp = fmaf(hval(alpha, e, fhd), ga, sp);
      }
      const float Ss = hub_reduce1<false>(sp, tile, G, S, f);
      float gp = 0.f;
      for (int e = hbeg + slot; e < hend; e += S) {
        const int j = ival(col, e);
        const float ga = hdot(hgo, rowldv(h, j, fc0));
        const float gs = hval(alpha, e, fhd) * (ga - Ss);
        const float raw = hval(a_src, j, fhd) + hadst;
        const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
        if (fleader) *hptr(g_e, e, fhd) = ge;
        gp = gp + ge;
      }
      const float gsum = hub_reduce1<false>(gp, tile, G, S, f);
      if ((lane >> gm.lgG) == (owner >> gm.lgG)) gad = gsum;
    }
  }
  const gatres_rowv<W> go = rowldv(g_out, row, c0);
  const float adst = hval(a_dst, row, hd);
  if (end - beg <= 8) {                        // the common case: every load of the row issued together (slot path)
    const int deg = end - beg;
    // K edge slots: 4 when every row of the wave has <= 4 in-edges (three of four waves of a water network), else 8.
    // Padding slots re-load the row's last neighbour and are masked out of every sum: the same bits either way.
    auto slots = [&](auto KC) {
      constexpr int K = decltype(KC)::value;
      int jj[K];
#pragma unroll
      for (int k = 0; k < K; ++k) jj[k] = ival(col, beg + min(k, deg - 1));
      gatres_rowv<W> hv[K];
      float al[K], as[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        hv[k] = rowldv(h, jj[k], c0);
        al[k] = hval(alpha, (beg + min(k, deg - 1)), hd);
        as[k] = hval(a_src, jj[k], hd);
      }
      float ga[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        ga[k] = hdot(go, hv[k]);
        if (k < deg) S = fmaf(al[k], ga[k], S);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k < deg) {
          const float gs = al[k] * (ga[k] - S);
          const float raw = as[k] + adst;
          const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
          if (leader) *hptr(g_e, (beg + k), hd) = ge;
          gad = gad + ge;
        }
      }
    };
    // The same with ONE edge slot per lane of the head's lane group for everything that is per EDGE rather than per feature
    // (heads of 8 lanes and more): lane k loads alpha_k / a_src_k, the coefficients go round the group by ds_swizzle for the
    // sum S (every lane holds all the head-reduced dots already), lane k forms gs_k / g_e_k and stores it, and the g_e go round
    // once more so that g_a_dst is summed in CSR order: the same bits, 2 K fewer loads and ~5 K fewer VALU instructions per lane.
    auto slots_lane = [&](auto KC) {
      constexpr int K = decltype(KC)::value;
      constexpr int LHL = LHT >= 8 ? LHT : 8;
      static_assert(K <= LHL, "a slot per lane");
      const int kk = (c0 >> LGW) & (LHL - 1);                     // this lane's place in its head's lane group = its slot
      int jj[K];
#pragma unroll
      for (int k = 0; k < K; ++k) jj[k] = ival(col, beg + min(k, deg - 1));
      const int em = beg + min(min(kk, K - 1), deg - 1);
      const int jm = ival(col, em);
      gatres_rowv<W> hv[K];
#pragma unroll
      for (int k = 0; k < K; ++k) hv[k] = rowldv(h, jj[k], c0);
      const float al_m = hval(alpha, em, hd), as_m = hval(a_src, jm, hd);
      auto bcast = [&](float x, auto KK) {                        // lane KK of the lane group (32-lane swizzle: and | or << 5)
        constexpr int pat = (0x1f & ~(LHL - 1)) | (decltype(KK)::value << 5);
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), pat));
      };
      float al[K], ga[K];
      al[0] = bcast(al_m, std::integral_constant<int, 0>{}); al[1] = bcast(al_m, std::integral_constant<int, 1>{});
      al[2] = bcast(al_m, std::integral_constant<int, 2>{}); al[3] = bcast(al_m, std::integral_constant<int, 3>{});
      if constexpr (K > 4) {
        al[4] = bcast(al_m, std::integral_constant<int, 4>{}); al[5] = bcast(al_m, std::integral_constant<int, 5>{});
        al[6] = bcast(al_m, std::integral_constant<int, 6>{}); al[7] = bcast(al_m, std::integral_constant<int, 7>{});
      }
      float ga_m = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        ga[k] = hdot(go, hv[k]);
        if (k < deg) S = fmaf(al[k], ga[k], S);
        ga_m = kk == k ? ga[k] : ga_m;
      }
      const float gs = al_m * (ga_m - S);
      const float raw = as_m + adst;
      const float ge_m = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
      if (valid && kk < deg) *hptr(g_e, (beg + kk), hd) = ge_m;
      float ge[K];
      ge[0] = bcast(ge_m, std::integral_constant<int, 0>{}); ge[1] = bcast(ge_m, std::integral_constant<int, 1>{});
      ge[2] = bcast(ge_m, std::integral_constant<int, 2>{}); ge[3] = bcast(ge_m, std::integral_constant<int, 3>{});
      if constexpr (K > 4) {
        ge[4] = bcast(ge_m, std::integral_constant<int, 4>{}); ge[5] = bcast(ge_m, std::integral_constant<int, 5>{});
        ge[6] = bcast(ge_m, std::integral_constant<int, 6>{}); ge[7] = bcast(ge_m, std::integral_constant<int, 7>{});
      }
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (k < deg) gad = gad + ge[k];
    };
    const bool few = W == 8 && __ballot(deg > 4) == 0ULL;
    if constexpr (LHT >= 8) {
      if (few) slots_lane(std::integral_constant<int, 4>{});
      else slots_lane(std::integral_constant<int, 8>{});
    } else {
      if (few) slots(std::integral_constant<int, 4>{});
      else slots(std::integral_constant<int, 8>{});
    }
  } else if (end - beg <= HUB_MIN_DEGREE) {    // the row's own lanes, edge after edge; the dots are recomputed in the second pass
    for (int e = beg; e < end; ++e) {
      const float ga = hdot(go, rowldv(h, ival(col, e), c0));
      S = fmaf(hval(alpha, e, hd), ga, S);
    }
    for (int e = beg; e < end; ++e) {
      const int j = ival(col, e);
      const float ga = hdot(go, rowldv(h, j, c0));
      const float gs = hval(alpha, e, hd) * (ga - S);
      const float raw = hval(a_src, j, hd) + adst;
      const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
      if (leader) *hptr(g_e, e, hd) = ge;
      gad = gad + ge;
    }
  }
  if (leader) *hptr(g_a_dst, row, hd) = gad;
}

// ------------------------------------------------------------------------------------------------------
// K2 backward, source-major over CSR^T:
//   g_a_src[j] = sum_{e out of j} g_e
//   g_h[j]     = sum_{e=(j->i)} alpha_e * g_out[i]  +  g_a_src[j] (x) att_src  +  g_a_dst[j] (x) att_dst
// ------------------------------------------------------------------------------------------------------
template <typename T, typename IDX, int W>
__global__ __launch_bounds__(256, W == 16 ? 4 : W == 8 ? 6 : 8) void gat_aggregate_bwd_src_kernel(
    const int* __restrict__ t_rowptr, const int* __restrict__ t_eid, const int* __restrict__ t_dst,
    const T* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ g_e,
    const float* __restrict__ g_a_dst, const float* __restrict__ att_src, const float* __restrict__ att_dst,
    T* __restrict__ g_h, float* __restrict__ g_a_src, int N, RowGeom gm) {
  __shared__ __attribute__((aligned(16))) float hub_lds[4 * HUB_FLOATS_PER_WAVE];
  constexpr int LGW = W == 16 ? 4 : W == 8 ? 3 : 2, Q = W / 4;
  const int lgHC = gm.lgG + LGW, lgH = lgHC - gm.lgC;
  GATRES_AGG_ADDRESSING(lgHC, lgH)
  auto rowldv = [&](const T* tab, int j, int c) { return ldrowv<W>(gatres_at<IDX>(tab, ((IDX)j << lgHC) + (IDX)c)); };
  auto axpyv = [&](gatres_rowv<W>& a, float s, const gatres_rowv<W>& v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) gatres_axpy4(a.v[q], s, v.v[q]);
  };
  (void)axpyv; (void)rowldv;
  const int tid = gatres_xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (XCD-local row ranges)
  int row = tid >> gm.lgG;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  const int c0 = (tid & (gm.G - 1)) * W;
  const int hd = c0 >> gm.lgC;
  const bool leader = valid && (c0 & (gm.C - 1)) == 0;
  const int H = gm.H, HC = gm.HC;
  const int beg = ival(t_rowptr, row), end0 = ival(t_rowptr, row + 1);
  gatres_rowv<W> acc = rowv_zero<W>();
  float gas = 0.f;
  // hub sources (more than HUB_MIN_DEGREE out-edges): the whole wave, S edge slots of G lanes, partial sums through LDS
  const bool hub = valid && end0 - beg > HUB_MIN_DEGREE;
  const unsigned long long hubs = __ballot(hub && c0 == 0);
  if (__builtin_expect(hubs != 0ULL, 0)) {
    const int lane = threadIdx.x & 63, G = gm.G, S = 64 >> gm.lgG;
    const int f = lane & (G - 1), slot = lane >> gm.lgG;
    float* tile = hub_tile(hub_lds);
    unsigned long long todo = hubs;
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int hbeg = __shfl(beg, owner), hend = __shfl(end0, owner);
      const int fc0 = f * W, fhd = fc0 >> gm.lgC;
      gatres_rowv<W> part = rowv_zero<W>();
      float gp = 0.f;
      _Pragma("unroll 2") for (int tt = hbeg + slot; tt < hend; tt += S) {
        const int e = ival(t_eid, tt), i = ival(t_dst, tt);
        gp = gp + hval(g_e, e, fhd);
        axpyv(part, hval(alpha, e, fhd), rowldv(g_out, i, fc0));
      }
      gatres_rowv<W> sum;
      _Pragma("unroll") for (int q = 0; q < Q; ++q) sum.v[q] = hub_reduce4(part.v[q], tile, G, S, f);
      const float gsum = hub_reduce1<false>(gp, tile, G, S, f);
      if ((lane >> gm.lgG) == (owner >> gm.lgG)) { acc = sum; gas = gsum; }
    }
  }
  int end = hub ? beg : end0;                  // (a hub's edges are done)
  // every row of the wave has <= 4 out-edges (most waves of a water network): the rows' loads are all issued together --
  // one dependent round trip {t_eid, t_dst} -> {alpha, g_e, g_out} instead of one per edge pair -- and summed in edge
  // order: the same bits as the loop below
  if (__ballot(end - beg > 4) == 0ULL) {
    const int deg = end - beg;
    int ee[4], ii[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ee[k] = k < deg ? ival(t_eid, beg + k) : 0; ii[k] = k < deg ? ival(t_dst, beg + k) : row; }   // (padding: valid entries, masked out below)
    gatres_rowv<W> gv[4];
    float al[4], ge[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gv[k] = rowldv(g_out, ii[k], c0);
      al[k] = hval(alpha, ee[k], hd);
      ge[k] = hval(g_e, ee[k], hd);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < deg) { gas = gas + ge[k]; axpyv(acc, al[k], gv[k]); }
    end = beg;
  }
  int t = beg;
  for (; t + 1 < end; t += 2) {
    const int e0 = ival(t_eid, t), e1 = ival(t_eid, t + 1);
    const int i0 = ival(t_dst, t), i1 = ival(t_dst, t + 1);
    const gatres_rowv<W> v0 = rowldv(g_out, i0, c0);
    const gatres_rowv<W> v1 = rowldv(g_out, i1, c0);
    const float al0 = hval(alpha, e0, hd), al1 = hval(alpha, e1, hd);
    gas = gas + hval(g_e, e0, hd);
    gas = gas + hval(g_e, e1, hd);
    axpyv(acc, al0, v0);
    axpyv(acc, al1, v1);
  }
  if (t < end) {
    const int e0 = ival(t_eid, t), i0 = ival(t_dst, t);
    const gatres_rowv<W> v0 = rowldv(g_out, i0, c0);
    const float al0 = hval(alpha, e0, hd);
    gas = gas + hval(g_e, e0, hd);
    axpyv(acc, al0, v0);
  }
  if (leader) *hptr(g_a_src, row, hd) = gas;
  const float gad = hval(g_a_dst, row, hd);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float4 as = ld4(att_src + c0 + 4 * q), ad = ld4(att_dst + c0 + 4 * q);
    gatres_axpy4(acc.v[q], gas, as);
    gatres_axpy4(acc.v[q], gad, ad);
  }
  if (valid) strowv<W>(gatres_at_w<IDX>(g_h, ((IDX)row << lgHC) + (IDX)c0), acc);
}

// ------------------------------------------------------------------------------------------------------
// K3: out_i = relu( (sum_{j->i} y[j]) / max(indeg(i),1) + x0_i )       and its backward
// ------------------------------------------------------------------------------------------------------
// W = features per lane (4, or 8 for bf16 rows of 64 features and more: one 16-byte access, half the lanes per row).
template <typename T, typename IDX, int W>
__global__ __launch_bounds__(256, W == 16 ? 5 : 8) void mean_residual_relu_fwd_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const T* __restrict__ y,
    const T* __restrict__ x0, T* __restrict__ out, int N, int C, int G, int lgG) {
  __shared__ __attribute__((aligned(16))) float hub_lds[4 * HUB_FLOATS_PER_WAVE];
  constexpr int LGW = W == 16 ? 4 : W == 8 ? 3 : 2, Q = W / 4;
  const int lgHC = lgG + LGW;
  GATRES_AGG_ADDRESSING(lgHC, 0)
  auto rowldv = [&](const T* tab, int j, int c) { return ldrowv<W>(gatres_at<IDX>(tab, ((IDX)j << lgHC) + (IDX)c)); };
  const int tid = gatres_xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (XCD-local row ranges)
  int row = tid >> lgG;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  const int c0 = (tid & (G - 1)) * W;
  const int beg = ival(rowptr, row), end0 = ival(rowptr, row + 1);
  gatres_rowv<W> acc = rowv_zero<W>();
  auto add = [&](gatres_rowv<W>& a, const gatres_rowv<W>& v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) { a.v[q].x = a.v[q].x + v.v[q].x; a.v[q].y = a.v[q].y + v.v[q].y; a.v[q].z = a.v[q].z + v.v[q].z; a.v[q].w = a.v[q].w + v.v[q].w; }
  };
  const bool hub = valid && end0 - beg > HUB_MIN_DEGREE;                    // hub rows: whole wave + LDS-staged partial sums
  const unsigned long long hubs = __ballot(hub && c0 == 0);
  if (__builtin_expect(hubs != 0ULL, 0)) {
    const int lane = threadIdx.x & 63, S = 64 >> lgG;
    const int f = lane & (G - 1), slot = lane >> lgG;
    float* tile = hub_tile(hub_lds);
    unsigned long long todo = hubs;
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int hbeg = __shfl(beg, owner), hend = __shfl(end0, owner);
      gatres_rowv<W> part = rowv_zero<W>();
      _Pragma("unroll 2") for (int e = hbeg + slot; e < hend; e += S) add(part, rowldv(y, ival(col, e), f * W));
      gatres_rowv<W> sum;
#pragma unroll
      for (int q = 0; q < Q; ++q) sum.v[q] = hub_reduce4(part.v[q], tile, G, S, f);
      if ((lane >> lgG) == (owner >> lgG)) acc = sum;
    }
  }
  int end = hub ? beg : end0;
  if (__ballot(end - beg > 4) == 0ULL) {       // (all rows of the wave: <= 4 neighbours -> their rows requested together, summed in edge order)
    const int deg = end - beg;
    int jj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) jj[k] = k < deg ? ival(col, beg + k) : row;
    gatres_rowv<W> v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = rowldv(y, jj[k], c0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < deg) add(acc, v[k]);
    end = beg;
  }
  int e = beg;
  for (; e + 1 < end; e += 2) {
    const gatres_rowv<W> v0 = rowldv(y, ival(col, e), c0);
    const gatres_rowv<W> v1 = rowldv(y, ival(col, e + 1), c0);
    add(acc, v0);
    add(acc, v1);
  }
  if (e < end) add(acc, rowldv(y, ival(col, e), c0));
  const float cnt = (float)max(end0 - beg, 1);
  const gatres_rowv<W> r = rowldv(x0, row, c0);
  gatres_rowv<W> o;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    o.v[q].x = fmaxf(acc.v[q].x / cnt + r.v[q].x, 0.f);
    o.v[q].y = fmaxf(acc.v[q].y / cnt + r.v[q].y, 0.f);
    o.v[q].z = fmaxf(acc.v[q].z / cnt + r.v[q].z, 0.f);
    o.v[q].w = fmaxf(acc.v[q].w / cnt + r.v[q].w, 0.f);
  }
  if (valid) strowv<W>(gatres_at_w<IDX>(out, ((IDX)row << lgHC) + (IDX)c0), o);
}

template <typename T, typename IDX, int W>
__global__ __launch_bounds__(256, W == 16 ? 5 : 8) void mean_bwd_kernel(
    const int* __restrict__ m_rowptr, const int* __restrict__ mt_rowptr, const int* __restrict__ mt_dst,
    const T* __restrict__ g_pre, T* __restrict__ g_y, int N, int C, int G, int lgG) {
  __shared__ __attribute__((aligned(16))) float hub_lds[4 * HUB_FLOATS_PER_WAVE];
  constexpr int LGW = W == 16 ? 4 : W == 8 ? 3 : 2, Q = W / 4;
  const int lgHC = lgG + LGW;
  GATRES_AGG_ADDRESSING(lgHC, 0)
  auto rowldv = [&](const T* tab, int j, int c) { return ldrowv<W>(gatres_at<IDX>(tab, ((IDX)j << lgHC) + (IDX)c)); };
  const int tid = gatres_xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (XCD-local row ranges)
  int row = tid >> lgG;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  const int c0 = (tid & (G - 1)) * W;
  const int beg = ival(mt_rowptr, row), end0 = ival(mt_rowptr, row + 1);
  gatres_rowv<W> acc = rowv_zero<W>();
  // fp32 storage: g / cnt per element, the reference's arithmetic bit for bit.  bf16 storage: ONE division per edge and a
  // multiply per element -- the product differs from the quotient by at most an fp32 ulp, far below the bf16 rounding of the
  // stored sum, and the kernel's VALU work per edge drops from 10 W to 10 + W instructions.
  auto add_div = [&](gatres_rowv<W>& a, const gatres_rowv<W>& v, float cnt) {
    if constexpr (sizeof(T) == 2) {
      const float inv = 1.f / cnt;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        a.v[q].x = a.v[q].x + v.v[q].x * inv; a.v[q].y = a.v[q].y + v.v[q].y * inv;
        a.v[q].z = a.v[q].z + v.v[q].z * inv; a.v[q].w = a.v[q].w + v.v[q].w * inv;
      }
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        a.v[q].x = a.v[q].x + v.v[q].x / cnt; a.v[q].y = a.v[q].y + v.v[q].y / cnt;
        a.v[q].z = a.v[q].z + v.v[q].z / cnt; a.v[q].w = a.v[q].w + v.v[q].w / cnt;
      }
    }
  };
  const bool hub = valid && end0 - beg > HUB_MIN_DEGREE;                    // hub sources: whole wave + LDS-staged partial sums
  const unsigned long long hubs = __ballot(hub && c0 == 0);
  if (__builtin_expect(hubs != 0ULL, 0)) {
    const int lane = threadIdx.x & 63, S = 64 >> lgG;
    const int f = lane & (G - 1), slot = lane >> lgG;
    float* tile = hub_tile(hub_lds);
    unsigned long long todo = hubs;
    while (todo) {
      const int owner = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int hbeg = __shfl(beg, owner), hend = __shfl(end0, owner);
      gatres_rowv<W> part = rowv_zero<W>();
      _Pragma("unroll 2") for (int t = hbeg + slot; t < hend; t += S) {
        const int i = ival(mt_dst, t);
        const float cnt = (float)max(ival(m_rowptr, i + 1) - ival(m_rowptr, i), 1);
        add_div(part, rowldv(g_pre, i, f * W), cnt);
      }
      gatres_rowv<W> sum;
#pragma unroll
      for (int q = 0; q < Q; ++q) sum.v[q] = hub_reduce4(part.v[q], tile, G, S, f);
      if ((lane >> lgG) == (owner >> lgG)) acc = sum;
    }
  }
  int end = hub ? beg : end0;
  // bf16 storage (one reciprocal per edge, light arithmetic): rows of up to four out-edges issue every load of the row together --
  // one dependent round trip mt_dst -> {m_rowptr x 2, g_pre row} instead of one per edge -- and sum in edge order: the same
  // bits as the loop.  (With fp32's per-element divisions the slot form measured slower: 13.6 vs 12.8 us; it keeps the loop.)
  if constexpr (sizeof(T) == 2) {
    if (__ballot(end - beg > 4) == 0ULL) {
      const int deg = end - beg;
      int ii[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ii[k] = k < deg ? ival(mt_dst, beg + k) : row;        // (padding: a valid row, masked out below)
      gatres_rowv<W> gv[4];
      int c0r[4], c1r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gv[k] = rowldv(g_pre, ii[k], c0);
        c0r[k] = ival(m_rowptr, ii[k]); c1r[k] = ival(m_rowptr, ii[k] + 1);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < deg) add_div(acc, gv[k], (float)max(c1r[k] - c0r[k], 1));
      end = beg;
    }
  }
  for (int t = beg; t < end; ++t) {
    const int i = ival(mt_dst, t);
    const float cnt = (float)max(ival(m_rowptr, i + 1) - ival(m_rowptr, i), 1);
    add_div(acc, rowldv(g_pre, i, c0), cnt);
  }
  if (valid) strowv<W>(gatres_at_w<IDX>(g_y, ((IDX)row << lgHC) + (IDX)c0), acc);
}

static inline bool graph_ok(const gatres_graph_t* g) {
  return g && g->num_nodes > 0 && g->rowptr && g->col && g->t_rowptr && g->t_eid && g->t_dst && g->m_rowptr &&
         g->m_col && g->mt_rowptr && g->mt_dst;
}

static inline int grid_rows(int N, int G) {
  const long long threads = (long long)N * G;
  return (int)((threads + 255) / 256);
}

}  // namespace

// Typed launchers (gatres_t_*): `dtype` says what the void* activation tensors hold.  The C-ABI entry points below are
// their fp32 instances.
#define GATRES_DISPATCH_T(dtype_, fit32_, CALL_)                                                       \
  switch (dtype_) {                                                                                    \
    case GATRES_DTYPE_F32:                                                                             \
      if (fit32_) { using T = float; using IDX = unsigned; CALL_; }                                    \
      else { using T = float; using IDX = size_t; CALL_; }                                             \
      break;                                                                                           \
    case GATRES_DTYPE_BF16:                                                                            \
      if (fit32_) { using T = gatres_bf16; using IDX = unsigned; CALL_; }                              \
      else { using T = gatres_bf16; using IDX = size_t; CALL_; }                                       \
      break;                                                                                           \
    default: return GATRES_E_UNSUPPORTED;                                                              \
  }

// storage type x offset width x features per lane: T, IDX and the constant WV inside CALL_  (flat: CALL_ holds kernel
// launches, whose expansion must not pass through another macro's argument list)
#define GATRES_DISPATCH_TIW(dtype_, fit32_, w_, CALL_)                                                  \
  switch (dtype_) {                                                                                    \
    case GATRES_DTYPE_F32:                                                                             \
      if ((w_) == 8) { if (fit32_) { using T = float; using IDX = unsigned; constexpr int WV = 8; CALL_; } else { using T = float; using IDX = size_t; constexpr int WV = 8; CALL_; } } \
      else { if (fit32_) { using T = float; using IDX = unsigned; constexpr int WV = 4; CALL_; } else { using T = float; using IDX = size_t; constexpr int WV = 4; CALL_; } } \
      break;                                                                                           \
    case GATRES_DTYPE_BF16:                                                                            \
      if ((w_) == 8) { if (fit32_) { using T = gatres_bf16; using IDX = unsigned; constexpr int WV = 8; CALL_; } else { using T = gatres_bf16; using IDX = size_t; constexpr int WV = 8; CALL_; } } \
      else { if (fit32_) { using T = gatres_bf16; using IDX = unsigned; constexpr int WV = 4; CALL_; } else { using T = gatres_bf16; using IDX = size_t; constexpr int WV = 4; CALL_; } } \
      break;                                                                                           \
    default: return GATRES_E_UNSUPPORTED;                                                              \
  }

// every byte offset into a [rows or edges] x width table of 4-byte elements fits 32 bits (the IDX = unsigned instances)
static inline bool offsets_fit_32(const gatres_graph_t* g, int width) {
  long long most = g->num_nodes;
  if (g->num_edges_gat > most) most = g->num_edges_gat;
  if (g->num_edges_mean > most) most = g->num_edges_mean;
  return (most + 2) * (long long)width * 4 < (1LL << 32) && !gatres_knobs()->agg_wide_offsets;
}

extern "C" int gatres_t_gat_aggregate_fwd(const gatres_graph_t* g, const void* h, const float* a_src, const float* a_dst,
                               const float* bias, void* out, float* alpha, int H, int C, int apply_relu, int dtype,
                               void* stream) {
  if (!graph_ok(g) || !h || !a_src || !a_dst || !bias || !out || !alpha) return GATRES_E_BADARG;
  if (!gatres_aligned16(h) || !gatres_aligned16(out) || !gatres_aligned16(bias)) return GATRES_E_BADARG;
  const int N = g->num_nodes, W = lane_features(H * C, C);
  RowGeom gm;
  if (!make_geom(H, C, &gm, W)) return GATRES_E_UNSUPPORTED;
  const bool fit32 = offsets_fit_32(g, gm.HC);
  dim3 grid(grid_rows(N, gm.G)), block(256);
  GATRES_DISPATCH_TIW(dtype, fit32, W, {
    if (apply_relu)
      hipLaunchKernelGGL((gat_aggregate_fwd_kernel<true, T, IDX, WV>), grid, block, 0, gatres_stream(stream), g->rowptr,
                         g->col, (const T*)h, a_src, a_dst, bias, (T*)out, alpha, N, gm);
    else
      hipLaunchKernelGGL((gat_aggregate_fwd_kernel<false, T, IDX, WV>), grid, block, 0, gatres_stream(stream), g->rowptr,
                         g->col, (const T*)h, a_src, a_dst, bias, (T*)out, alpha, N, gm);
  })
  return gatres_launch_status();
}

template <typename T, typename IDX, int W>
static void launch_bwd_dst(const gatres_graph_t* g, const T* g_out, const T* h, const float* alpha, const float* a_src,
                           const float* a_dst, float* g_e, float* g_a_dst, int N, const RowGeom& gm, void* stream) {
#define GATRES_BWD_DST_LAUNCH(LHT_)                                                                                      \
  hipLaunchKernelGGL((gat_aggregate_bwd_dst_kernel<T, IDX, W, LHT_>), dim3(grid_rows(N, gm.G)), dim3(256), 0,            \
                     gatres_stream(stream), g->rowptr, g->col, g_out, h, alpha, a_src, a_dst, g_e, g_a_dst, N, gm)
  switch (gm.C / W) {                          // lanes per head
    case 2: GATRES_BWD_DST_LAUNCH(2); break;
    case 4: GATRES_BWD_DST_LAUNCH(4); break;
    case 8: GATRES_BWD_DST_LAUNCH(8); break;
    case 16: GATRES_BWD_DST_LAUNCH(16); break;
    case 32: GATRES_BWD_DST_LAUNCH(32); break;
    default: GATRES_BWD_DST_LAUNCH(0); break;
  }
#undef GATRES_BWD_DST_LAUNCH
}

extern "C" int gatres_t_gat_aggregate_bwd_dst(const gatres_graph_t* g, const void* g_out, const void* h, const float* alpha,
                                   const float* a_src, const float* a_dst, float* g_e, float* g_a_dst, int H, int C,
                                   int dtype, void* stream) {
  if (!graph_ok(g) || !g_out || !h || !alpha || !a_src || !a_dst || !g_e || !g_a_dst) return GATRES_E_BADARG;
  if (!gatres_aligned16(h) || !gatres_aligned16(g_out)) return GATRES_E_BADARG;
  RowGeom gm;
  const int N = g->num_nodes, W = lane_features(H * C, C);
  if (!make_geom(H, C, &gm, W)) return GATRES_E_UNSUPPORTED;
  const bool fit32 = offsets_fit_32(g, gm.HC);
  GATRES_DISPATCH_TIW(dtype, fit32, W, {
    (launch_bwd_dst<T, IDX, WV>)(g, (const T*)g_out, (const T*)h, alpha, a_src, a_dst, g_e, g_a_dst, N, gm, stream);
  })
  return gatres_launch_status();
}

extern "C" int gatres_t_gat_aggregate_bwd_src(const gatres_graph_t* g, const void* g_out, const float* alpha, const float* g_e,
                                   const float* g_a_dst, const float* att_src, const float* att_dst, void* g_h,
                                   float* g_a_src, int H, int C, int dtype, void* stream) {
  if (!graph_ok(g) || !g_out || !alpha || !g_e || !g_a_dst || !att_src || !att_dst || !g_h || !g_a_src)
    return GATRES_E_BADARG;
  if (!gatres_aligned16(g_out) || !gatres_aligned16(g_h) || !gatres_aligned16(att_src) ||
      !gatres_aligned16(att_dst))
    return GATRES_E_BADARG;
  RowGeom gm;
  const int N = g->num_nodes, W = lane_features(H * C, C);
  if (!make_geom(H, C, &gm, W)) return GATRES_E_UNSUPPORTED;
  const bool fit32 = offsets_fit_32(g, gm.HC);
  GATRES_DISPATCH_TIW(dtype, fit32, W, {
    hipLaunchKernelGGL((gat_aggregate_bwd_src_kernel<T, IDX, WV>), dim3(grid_rows(N, gm.G)), dim3(256), 0,
                       gatres_stream(stream), g->t_rowptr, g->t_eid, g->t_dst, (const T*)g_out, alpha, g_e, g_a_dst,
                       att_src, att_dst, (T*)g_h, g_a_src, N, gm);
  })
  return gatres_launch_status();
}

extern "C" int gatres_t_mean_residual_relu_fwd(const gatres_graph_t* g, const void* y, const void* x0, void* out, int C, int dtype,
                                    void* stream) {
  if (!graph_ok(g) || !y || !x0 || !out) return GATRES_E_BADARG;
  if (!gatres_aligned16(y) || !gatres_aligned16(x0) || !gatres_aligned16(out)) return GATRES_E_BADARG;
  if (C < 4 || !gatres_is_pow2(C) || C > 256) return GATRES_E_UNSUPPORTED;
  const int N = g->num_nodes, W = lane_features(C, C), G = C / W;
  const bool fit32 = offsets_fit_32(g, C);
  GATRES_DISPATCH_TIW(dtype, fit32, W, {
    hipLaunchKernelGGL((mean_residual_relu_fwd_kernel<T, IDX, WV>), dim3(grid_rows(N, G)), dim3(256), 0,
                       gatres_stream(stream), g->m_rowptr, g->m_col, (const T*)y, (const T*)x0, (T*)out, N, C, G, ilog2(G));
  })
  return gatres_launch_status();
}

extern "C" int gatres_t_mean_bwd(const gatres_graph_t* g, const void* g_pre, void* g_y, int C, int dtype, void* stream) {
  if (!graph_ok(g) || !g_pre || !g_y) return GATRES_E_BADARG;
  if (!gatres_aligned16(g_pre) || !gatres_aligned16(g_y)) return GATRES_E_BADARG;
  if (C < 4 || !gatres_is_pow2(C) || C > 256) return GATRES_E_UNSUPPORTED;
  const int N = g->num_nodes, W = lane_features(C, C), G = C / W;
  const bool fit32 = offsets_fit_32(g, C);
  GATRES_DISPATCH_TIW(dtype, fit32, W, {
    hipLaunchKernelGGL((mean_bwd_kernel<T, IDX, WV>), dim3(grid_rows(N, G)), dim3(256), 0, gatres_stream(stream),
                       g->m_rowptr, g->mt_rowptr, g->mt_dst, (const T*)g_pre, (T*)g_y, N, C, G, ilog2(G));
  })
  return gatres_launch_status();
}

extern "C" int gatres_gat_aggregate_fwd(const gatres_graph_t* g, const float* h, const float* a_src,
                                        const float* a_dst, const float* bias, float* out, float* alpha,
                                        int32_t H, int32_t C, int32_t apply_relu, void* stream) {
  return gatres_t_gat_aggregate_fwd(g, h, a_src, a_dst, bias, out, alpha, H, C, apply_relu, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_gat_aggregate_bwd_dst(const gatres_graph_t* g, const float* g_out, const float* h,
                                            const float* alpha, const float* a_src, const float* a_dst,
                                            float* g_e, float* g_a_dst, int32_t H, int32_t C, void* stream) {
  return gatres_t_gat_aggregate_bwd_dst(g, g_out, h, alpha, a_src, a_dst, g_e, g_a_dst, H, C, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_gat_aggregate_bwd_src(const gatres_graph_t* g, const float* g_out, const float* alpha,
                                            const float* g_e, const float* g_a_dst, const float* att_src,
                                            const float* att_dst, float* g_h, float* g_a_src, int32_t H,
                                            int32_t C, void* stream) {
  return gatres_t_gat_aggregate_bwd_src(g, g_out, alpha, g_e, g_a_dst, att_src, att_dst, g_h, g_a_src, H, C,
                                        GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_mean_residual_relu_fwd(const gatres_graph_t* g, const float* y, const float* x0,
                                             float* out, int32_t C, void* stream) {
  return gatres_t_mean_residual_relu_fwd(g, y, x0, out, C, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_mean_bwd(const gatres_graph_t* g, const float* g_pre, float* g_y, int32_t C, void* stream) {
  return gatres_t_mean_bwd(g, g_pre, g_y, C, GATRES_DTYPE_F32, stream);
}
