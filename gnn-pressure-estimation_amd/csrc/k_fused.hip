// Fused per-snapshot GATRes kernels for gfx950: ONE workgroup owns ONE graph segment (a snapshot of the batch)
// and carries it through lin0 -> num_blocks x [K1 conv1, K2 conv1, K1 conv2, K2 conv2, K3] -> lin1 -> masked-MSE ->
// the whole backward, inside a single launch.
//
// Why (measured on MI355X, profiles/r01_unfused_*): as ~230 separate launches every stage is a dependent kernel
// whose inputs were just written by other XCDs, so each costs ~5 us of launch + Infinity-Cache latency however
// little it moves; the step was 2.35 ms for 1.6 GB of algorithmic traffic.  A PyG batch is block-diagonal
// (train.py:302-303; `batch` is not even passed to the model, train.py:167), so a snapshot never needs another
// workgroup's data: the only synchronisation left is __syncthreads().
//
// Data placement: a 388-node C-Town snapshot at nc=32 needs 388 x (64 + 32) x 4 B = 149 KB for the two tables that
// are gathered by neighbour index (h of conv1 / y2, h of conv2) plus the attention logits -- it fits the CU's
// 160 KB LDS, so every neighbour gather of the forward pass is an LDS read (16-B per lane, conflict-free for
// 256-B rows under the ds_read_b128 lane groups).  Activations that the backward pass needs are written once to
// HBM (coalesced float4 rows); backward gathers are L2/Infinity-Cache hits.  Dense projections run on the matrix
// cores (v_mfma_f32_16x16x4_f32, exact fp32), 16 waves per workgroup, one 16-node tile per wave per trip.
//
// Arithmetic is statement-for-statement the same as the per-op kernels (k_aggregate.hip, k_proj.hip), so fused
// and per-op forward/backward agree bitwise except for the parameter-gradient slabs, which are per-segment here.
#include "gatres_common.h"
#include "gatres_layout.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_FLOATS_MAX = 40000;   // 160,000 B of the CU's 163,840 B

struct FusedArgs {
  // graph plan
  const int* seg_ptr;
  const int *rowptr, *col, *t_rowptr, *t_eid, *t_dst, *m_rowptr, *m_col, *mt_rowptr, *mt_dst;
  int N;
  // model
  const float* params;
  const float* wt;          // transposed conv weights (backward)
  // io
  const float* x;
  const uint8_t* mask;      // may be null (no masking of x; loss phase needs it)
  const float* y;
  float* out;
  float* g_out;             // written by the loss phase, read by the backward phase
  float* loss_part;         // [num_segments + 1]: per-segment sum of squared errors; last = masked-node count
  float* g_x;               // may be null
  float* saved;             // may be null for inference
  float* scratch;
  float* slabs;
  Layout L;
  int phases;               // GATRES_PHASE_FORWARD | _BACKWARD, bit 16: loss
  unsigned long long* stamps;   // diagnostic: segment 0 stamps the wall clock at every stage boundary
  int stamp_cap;
};

enum { PH_LOSS = 16 };

static unsigned long long* g_stamps = nullptr;
static int g_stamp_cap = 0;

#define STAMP()                                                                                  \
  do {                                                                                           \
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && stamp_i < a.stamp_cap)                \
      a.stamps[stamp_i++] = wall_clock64();                                                      \
  } while (0)

// ------------------------------------------------------------------------------------------ small helpers
template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < THREADS / 64; ++w) s += red[w];
  return s;
}

template <int KQ>
__device__ __forceinline__ void load_frag(const float* __restrict__ p, float (&f)[KQ]) {
  if constexpr (KQ % 4 == 0) {
#pragma unroll
    for (int s = 0; s < KQ; s += 4) {
      const float4 v = ld4(p + s);
      f[s] = v.x; f[s + 1] = v.y; f[s + 2] = v.z; f[s + 3] = v.w;
    }
  } else if constexpr (KQ % 2 == 0) {
#pragma unroll
    for (int s = 0; s < KQ; s += 2) {
      const float2 v = *reinterpret_cast<const float2*>(p + s);
      f[s] = v.x; f[s + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int s = 0; s < KQ; ++s) f[s] = p[s];
  }
}

enum { EPI_NONE = 0, EPI_ATT = 1, EPI_RESID_MASK = 2 };

// ------------------------------------------------------------------------------------------ K1 (MFMA)
// OUT[n0+r, :] = X[n0+r, :] @ Wm^T for r in [0, n).  Optional copies into LDS (local row index) for the gathers
// that follow.  Same tile / lane map / k order as proj_kernel in k_proj.hip.
template <int K, int M, int H, int EPI, int THREADS>
__device__ __forceinline__ void seg_proj(int n0, int n, const float* __restrict__ X, const float* __restrict__ Wm,
                                         float* __restrict__ OUT, float* OUT_L, const float* __restrict__ att_src,
                                         const float* __restrict__ att_dst, float* __restrict__ a_src_g,
                                         float* __restrict__ a_dst_g, float* a_src_l, float* a_dst_l,
                                         const float* __restrict__ resid, const float* __restrict__ relu_ref) {
  constexpr int KQ = K / 4, NT = (M + 15) / 16;
  constexpr int SC = (KQ % 4 == 0) ? 4 : ((KQ % 2 == 0) ? 2 : 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int ntiles = (n + 15) >> 4;
  for (int t = wave; t < ntiles; t += THREADS / 64) {
    const int r = t * 16 + i;
    const bool rok = r < n;
    const size_t node = (size_t)n0 + (rok ? r : n - 1);
    float xf[KQ];
    load_frag<KQ>(X + node * K + q * KQ, xf);
    f32x4 acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KQ; s += SC) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int m = tt * 16 + i;
        const bool mok = (M % 16 == 0) || (m < M);
        float wf[SC];
        load_frag<SC>(Wm + (size_t)(mok ? m : 0) * K + q * KQ + s, wf);
#pragma unroll
        for (int u = 0; u < SC; ++u)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(mok ? wf[u] : 0.f, xf[s + u], acc[tt], 0, 0, 0);
      }
    }
    if constexpr (EPI == EPI_ATT) {
      constexpr int C = M / H;
      float ps[H], pd[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int mb = tt * 16 + q * 4;
        if ((M % 16 == 0) || (mb < M)) {
          const float4 as = ld4(att_src + mb), ad = ld4(att_dst + mb);
          const float ds = fmaf(acc[tt][3], as.w, fmaf(acc[tt][2], as.z, fmaf(acc[tt][1], as.y, acc[tt][0] * as.x)));
          const float dd = fmaf(acc[tt][3], ad.w, fmaf(acc[tt][2], ad.z, fmaf(acc[tt][1], ad.y, acc[tt][0] * ad.x)));
          const int hd = mb / C;
#pragma unroll
          for (int hh = 0; hh < H; ++hh)
            if (hh == hd) { ps[hh] += ds; pd[hh] += dd; }
        }
      }
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        ps[hh] += __shfl_xor(ps[hh], 16); ps[hh] += __shfl_xor(ps[hh], 32);
        pd[hh] += __shfl_xor(pd[hh], 16); pd[hh] += __shfl_xor(pd[hh], 32);
      }
      if (q == 0 && rok) {
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          a_src_g[node * H + hh] = ps[hh];
          a_dst_g[node * H + hh] = pd[hh];
          if (a_src_l) { a_src_l[r * H + hh] = ps[hh]; a_dst_l[r * H + hh] = pd[hh]; }
        }
      }
    }
    if (rok) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int mb = tt * 16 + q * 4;
        if ((M % 16 == 0) || (mb < M)) {
          float4 o = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
          if constexpr (EPI == EPI_RESID_MASK) {
            if (resid) {
              const float4 rr = ld4(resid + node * M + mb);
              o.x = o.x + rr.x; o.y = o.y + rr.y; o.z = o.z + rr.z; o.w = o.w + rr.w;
            }
            if (relu_ref) {
              const float4 rr = ld4(relu_ref + node * M + mb);
              o.x = rr.x > 0.f ? o.x : 0.f; o.y = rr.y > 0.f ? o.y : 0.f;
              o.z = rr.z > 0.f ? o.z : 0.f; o.w = rr.w > 0.f ? o.w : 0.f;
            }
          }
          st4(OUT + node * M + mb, o);
          if (OUT_L) st4(OUT_L + (size_t)r * M + mb, o);
        }
      }
    }
  }
}

// dW partial of this segment: slab[c*K + k] = sum_r G[n0+r, c] * X[n0+r, k].  One 16x16 output tile per wave-trip.
template <int HC, int K, int THREADS>
__device__ __forceinline__ void seg_dw(int n0, int n, const float* __restrict__ G, const float* __restrict__ X,
                                       float* __restrict__ slab) {
  constexpr int NCT = (HC + 15) / 16, NKT = (K + 15) / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  for (int u = wave; u < NCT * NKT; u += THREADS / 64) {
    const int c = (u / NKT) * 16 + i, k = (u % NKT) * 16 + i;
    const bool cok = c < HC, kok = k < K;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int nb = 0; nb < n; nb += 4) {
      const int r = nb + q;
      const bool rok = r < n;
      const size_t node = (size_t)n0 + (rok ? r : 0);
      const float a = (rok && cok) ? G[node * HC + c] : 0.f;
      const float b = (rok && kok) ? X[node * K + k] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int cr = (u / NKT) * 16 + 4 * q + rr;
      if (cr < HC && kok) slab[(size_t)cr * K + k] = acc[rr];
    }
  }
}

// ------------------------------------------------------------------------------------------ K2 forward
// Row r of the segment is node n0 + r.  Gathered tables are addressed as base + (j - joff) * width so the same
// code reads either an LDS copy (joff = n0) or the global array (joff = 0).
template <bool RELU, int H, int C, int THREADS>
__device__ __forceinline__ void seg_agg_fwd(int n0, int n, const int* __restrict__ rowptr,
                                            const int* __restrict__ col, const float* hsrc, int hoff,
                                            const float* asrc, const float* adst_t, int aoff,
                                            const float* __restrict__ bias, float* out, int ooff,
                                            float* __restrict__ alpha) {
  constexpr int HC = H * C, G = HC / 4;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const bool leader = (c0 % C) == 0;
  for (int r = threadIdx.x / G; r < n; r += THREADS / G) {
    const int row = n0 + r;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float adst = adst_t[(row - aoff) * H + hd];
    float m = -INFINITY;
    for (int e = beg; e < end; ++e) m = fmaxf(m, gatres_leaky(asrc[(col[e] - aoff) * H + hd] + adst));
    float Z = 0.f;
    for (int e = beg; e < end; ++e) Z = Z + expf(gatres_leaky(asrc[(col[e] - aoff) * H + hd] + adst) - m);
    Z = Z + GATRES_SOFTMAX_EPS;
    float4 acc = f4zero();
    int e = beg;
    for (; e + 1 < end; e += 2) {
      const int j0 = col[e], j1 = col[e + 1];
      const float4 v0 = ld4(hsrc + (size_t)(j0 - hoff) * HC + c0);
      const float4 v1 = ld4(hsrc + (size_t)(j1 - hoff) * HC + c0);
      const float al0 = expf(gatres_leaky(asrc[(j0 - aoff) * H + hd] + adst) - m) / Z;
      const float al1 = expf(gatres_leaky(asrc[(j1 - aoff) * H + hd] + adst) - m) / Z;
      if (leader) { alpha[(size_t)e * H + hd] = al0; alpha[(size_t)(e + 1) * H + hd] = al1; }
      acc.x = acc.x + al0 * v0.x; acc.y = acc.y + al0 * v0.y; acc.z = acc.z + al0 * v0.z; acc.w = acc.w + al0 * v0.w;
      acc.x = acc.x + al1 * v1.x; acc.y = acc.y + al1 * v1.y; acc.z = acc.z + al1 * v1.z; acc.w = acc.w + al1 * v1.w;
    }
    if (e < end) {
      const int j0 = col[e];
      const float4 v0 = ld4(hsrc + (size_t)(j0 - hoff) * HC + c0);
      const float al0 = expf(gatres_leaky(asrc[(j0 - aoff) * H + hd] + adst) - m) / Z;
      if (leader) alpha[(size_t)e * H + hd] = al0;
      acc.x = acc.x + al0 * v0.x; acc.y = acc.y + al0 * v0.y; acc.z = acc.z + al0 * v0.z; acc.w = acc.w + al0 * v0.w;
    }
    const float4 b = ld4(bias + c0);
    acc.x = acc.x + b.x; acc.y = acc.y + b.y; acc.z = acc.z + b.z; acc.w = acc.w + b.w;
    if (RELU) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    st4(out + (size_t)(row - ooff) * HC + c0, acc);
  }
}

// K3 forward
template <int C, int THREADS>
__device__ __forceinline__ void seg_mean_fwd(int n0, int n, const int* __restrict__ rowptr,
                                             const int* __restrict__ col, const float* y, int yoff,
                                             const float* __restrict__ x0, float* __restrict__ out) {
  constexpr int G = C / 4;
  const int c0 = (threadIdx.x % G) * 4;
  for (int r = threadIdx.x / G; r < n; r += THREADS / G) {
    const int row = n0 + r;
    const int beg = rowptr[row], end = rowptr[row + 1];
    float4 acc = f4zero();
    int e = beg;
    for (; e + 1 < end; e += 2) {
      const float4 v0 = ld4(y + (size_t)(col[e] - yoff) * C + c0);
      const float4 v1 = ld4(y + (size_t)(col[e + 1] - yoff) * C + c0);
      acc.x = acc.x + v0.x; acc.y = acc.y + v0.y; acc.z = acc.z + v0.z; acc.w = acc.w + v0.w;
      acc.x = acc.x + v1.x; acc.y = acc.y + v1.y; acc.z = acc.z + v1.z; acc.w = acc.w + v1.w;
    }
    if (e < end) {
      const float4 v0 = ld4(y + (size_t)(col[e] - yoff) * C + c0);
      acc.x = acc.x + v0.x; acc.y = acc.y + v0.y; acc.z = acc.z + v0.z; acc.w = acc.w + v0.w;
    }
    const float cnt = (float)max(end - beg, 1);
    const float4 rr = ld4(x0 + (size_t)row * C + c0);
    float4 o;
    o.x = fmaxf(acc.x / cnt + rr.x, 0.f); o.y = fmaxf(acc.y / cnt + rr.y, 0.f);
    o.z = fmaxf(acc.z / cnt + rr.z, 0.f); o.w = fmaxf(acc.w / cnt + rr.w, 0.f);
    st4(out + (size_t)row * C + c0, o);
  }
}

// ------------------------------------------------------------------------------------------ backward stages
template <int C, int THREADS>
__device__ __forceinline__ void seg_mean_bwd(int n0, int n, const int* __restrict__ m_rowptr,
                                             const int* __restrict__ mt_rowptr, const int* __restrict__ mt_dst,
                                             const float* __restrict__ g_pre, float* __restrict__ g_y) {
  constexpr int G = C / 4;
  const int c0 = (threadIdx.x % G) * 4;
  for (int r = threadIdx.x / G; r < n; r += THREADS / G) {
    const int row = n0 + r;
    const int beg = mt_rowptr[row], end = mt_rowptr[row + 1];
    float4 acc = f4zero();
    for (int t = beg; t < end; ++t) {
      const int i = mt_dst[t];
      const float cnt = (float)max(m_rowptr[i + 1] - m_rowptr[i], 1);
      const float4 v = ld4(g_pre + (size_t)i * C + c0);
      acc.x = acc.x + v.x / cnt; acc.y = acc.y + v.y / cnt; acc.z = acc.z + v.z / cnt; acc.w = acc.w + v.w / cnt;
    }
    st4(g_y + (size_t)row * C + c0, acc);
  }
}

__device__ __forceinline__ float head_dot(const float4 a, const float4 b, int lanes_per_head) {
  float d = a.x * b.x;
  d = fmaf(a.y, b.y, d);
  d = fmaf(a.z, b.z, d);
  d = fmaf(a.w, b.w, d);
  for (int off = lanes_per_head >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
  return d;
}

template <int H, int C, int THREADS>
__device__ __forceinline__ void seg_agg_bwd_dst(int n0, int n, const int* __restrict__ rowptr,
                                                const int* __restrict__ col, const float* __restrict__ g_out,
                                                const float* __restrict__ h, const float* __restrict__ alpha,
                                                const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                float* __restrict__ g_e, float* __restrict__ g_a_dst) {
  constexpr int HC = H * C, G = HC / 4, LH = C / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const int rounds = (n + RPP - 1) / RPP;
  for (int it = 0; it < rounds; ++it) {
    int r = it * RPP + threadIdx.x / G;
    const bool valid = r < n;              // every lane stays in the loop: head_dot shuffles across the head's lanes
    if (!valid) r = n - 1;
    const bool leader = valid && (c0 % C) == 0;
    const int row = n0 + r;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float4 go = ld4(g_out + (size_t)row * HC + c0);
    const float adst = a_dst[row * H + hd];
    float S = 0.f, gad = 0.f;
    if (end - beg <= 8) {
      float ga[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ga[k] = 0.f;
        if (beg + k < end) {
          ga[k] = head_dot(go, ld4(h + (size_t)col[beg + k] * HC + c0), LH);
          S = S + alpha[(size_t)(beg + k) * H + hd] * ga[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (beg + k < end) {
          const int e = beg + k;
          const float gs = alpha[(size_t)e * H + hd] * (ga[k] - S);
          const float raw = a_src[col[e] * H + hd] + adst;
          const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
          if (leader) g_e[(size_t)e * H + hd] = ge;
          gad = gad + ge;
        }
      }
    } else {
      for (int e = beg; e < end; ++e) {
        const float ga = head_dot(go, ld4(h + (size_t)col[e] * HC + c0), LH);
        S = S + alpha[(size_t)e * H + hd] * ga;
      }
      for (int e = beg; e < end; ++e) {
        const int j = col[e];
        const float ga = head_dot(go, ld4(h + (size_t)j * HC + c0), LH);
        const float gs = alpha[(size_t)e * H + hd] * (ga - S);
        const float raw = a_src[j * H + hd] + adst;
        const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
        if (leader) g_e[(size_t)e * H + hd] = ge;
        gad = gad + ge;
      }
    }
    if (leader) g_a_dst[row * H + hd] = gad;
  }
}

template <int H, int C, int THREADS>
__device__ __forceinline__ void seg_agg_bwd_src(int n0, int n, const int* __restrict__ t_rowptr,
                                                const int* __restrict__ t_eid, const int* __restrict__ t_dst,
                                                const float* __restrict__ g_out, const float* __restrict__ alpha,
                                                const float* __restrict__ g_e, const float* __restrict__ g_a_dst,
                                                const float* __restrict__ att_src,
                                                const float* __restrict__ att_dst, float* __restrict__ g_h,
                                                float* __restrict__ g_a_src) {
  constexpr int HC = H * C, G = HC / 4;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const bool leader = (c0 % C) == 0;
  const float4 as = ld4(att_src + c0), ad = ld4(att_dst + c0);
  for (int r = threadIdx.x / G; r < n; r += THREADS / G) {
    const int row = n0 + r;
    const int beg = t_rowptr[row], end = t_rowptr[row + 1];
    float4 acc = f4zero();
    float gas = 0.f;
    int t = beg;
    for (; t + 1 < end; t += 2) {
      const int e0 = t_eid[t], e1 = t_eid[t + 1];
      const int i0 = t_dst[t], i1 = t_dst[t + 1];
      const float4 v0 = ld4(g_out + (size_t)i0 * HC + c0);
      const float4 v1 = ld4(g_out + (size_t)i1 * HC + c0);
      const float al0 = alpha[(size_t)e0 * H + hd], al1 = alpha[(size_t)e1 * H + hd];
      gas = gas + g_e[(size_t)e0 * H + hd];
      gas = gas + g_e[(size_t)e1 * H + hd];
      acc.x = acc.x + al0 * v0.x; acc.y = acc.y + al0 * v0.y; acc.z = acc.z + al0 * v0.z; acc.w = acc.w + al0 * v0.w;
      acc.x = acc.x + al1 * v1.x; acc.y = acc.y + al1 * v1.y; acc.z = acc.z + al1 * v1.z; acc.w = acc.w + al1 * v1.w;
    }
    if (t < end) {
      const int e0 = t_eid[t], i0 = t_dst[t];
      const float4 v0 = ld4(g_out + (size_t)i0 * HC + c0);
      const float al0 = alpha[(size_t)e0 * H + hd];
      gas = gas + g_e[(size_t)e0 * H + hd];
      acc.x = acc.x + al0 * v0.x; acc.y = acc.y + al0 * v0.y; acc.z = acc.z + al0 * v0.z; acc.w = acc.w + al0 * v0.w;
    }
    if (leader) g_a_src[row * H + hd] = gas;
    const float gad = g_a_dst[row * H + hd];
    acc.x = acc.x + gas * as.x; acc.y = acc.y + gas * as.y; acc.z = acc.z + gas * as.z; acc.w = acc.w + gas * as.w;
    acc.x = acc.x + gad * ad.x; acc.y = acc.y + gad * ad.y; acc.z = acc.z + gad * ad.z; acc.w = acc.w + gad * ad.w;
    st4(g_h + (size_t)row * HC + c0, acc);
  }
}

// column sums of one GATConv for this segment (att_src / att_dst / bias gradients) -> the segment's slab.
// thread = (column c, row group rg); partials are combined across row groups through LDS in a fixed order.
template <int H, int C, int THREADS>
__device__ __forceinline__ void seg_conv_param_grads(int n0, int n, const float* __restrict__ h,
                                                     const float* __restrict__ g_a_src,
                                                     const float* __restrict__ g_a_dst,
                                                     const float* __restrict__ g_out, float* __restrict__ slab_as,
                                                     float* __restrict__ slab_ad, float* __restrict__ slab_b,
                                                     float* red) {
  constexpr int HC = H * C, R = THREADS / HC;
  const int c = threadIdx.x % HC, rg = threadIdx.x / HC;
  const int hd = c / C;
  float as = 0.f, ad = 0.f, ab = 0.f;
  for (int r = rg; r < n; r += R) {
    const size_t node = (size_t)n0 + r;
    const float hv = h[node * HC + c];
    as = fmaf(g_a_src[node * H + hd], hv, as);
    ad = fmaf(g_a_dst[node * H + hd], hv, ad);
    ab += g_out[node * HC + c];
  }
  __syncthreads();
  red[threadIdx.x] = as; red[THREADS + threadIdx.x] = ad; red[2 * THREADS + threadIdx.x] = ab;
  __syncthreads();
  if (rg == 0) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < R; ++k) {
      s0 += red[k * HC + c]; s1 += red[THREADS + k * HC + c]; s2 += red[2 * THREADS + k * HC + c];
    }
    slab_as[c] = s0; slab_ad[c] = s1; slab_b[c] = s2;
  }
}

// lin1 backward for this segment: g_x = g_out (x) w (ReLU-masked), slab partials of g_w, g_b.
template <int NC, int THREADS>
__device__ __forceinline__ void seg_lin1_bwd(int n0, int n, const float* __restrict__ g_out,
                                             const float* __restrict__ x, const float* __restrict__ w,
                                             float* __restrict__ g_x, float* __restrict__ slab_w,
                                             float* __restrict__ slab_b, int relu_mask, float* red) {
  constexpr int R = THREADS / NC;
  const int c = threadIdx.x % NC, rg = threadIdx.x / NC;
  const float wv = w[c];
  float aw = 0.f, ab = 0.f;
  for (int r = rg; r < n; r += R) {
    const size_t node = (size_t)n0 + r;
    const float go = g_out[node];
    const float xv = x[node * NC + c];
    aw = fmaf(go, xv, aw);
    ab += go;
    g_x[node * NC + c] = (relu_mask && !(xv > 0.f)) ? 0.f : go * wv;
  }
  __syncthreads();
  red[threadIdx.x] = aw; red[THREADS + threadIdx.x] = ab;
  __syncthreads();
  if (rg == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < R; ++k) { s0 += red[k * NC + c]; s1 += red[THREADS + k * NC + c]; }
    slab_w[c] = s0;
    if (c == 0) slab_b[0] = s1;
  }
}

template <int NC, int THREADS>
__device__ __forceinline__ void seg_lin0_bwd(int n0, int n, const float* __restrict__ g, const float* __restrict__ x,
                                             const uint8_t* __restrict__ mask, float* __restrict__ slab_w,
                                             float* __restrict__ slab_b, float* red) {
  constexpr int R = THREADS / NC;
  const int c = threadIdx.x % NC, rg = threadIdx.x / NC;
  float aw = 0.f, ab = 0.f;
  for (int r = rg; r < n; r += R) {
    const size_t node = (size_t)n0 + r;
    const float xv = (mask && mask[node]) ? 0.f : x[node];
    const float gv = g[node * NC + c];
    aw = fmaf(gv, xv, aw);
    ab += gv;
  }
  __syncthreads();
  red[threadIdx.x] = aw; red[THREADS + threadIdx.x] = ab;
  __syncthreads();
  if (rg == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < R; ++k) { s0 += red[k * NC + c]; s1 += red[THREADS + k * NC + c]; }
    slab_w[c] = s0; slab_b[c] = s1;
  }
}

// ------------------------------------------------------------------------------------------ the kernel
template <int NC, int THREADS, bool CACHE>
__global__ __launch_bounds__(THREADS) void gatres_fused_kernel(const FusedArgs a) {
  // forward LDS map (CACHE): hA [n, 2NC] (h of conv1, then y2 [n, NC]) | hB [n, NC] (h of conv2) | sa, sd [n, 2]
  // backward / loss: the first 3*THREADS floats are reduction scratch
  constexpr int LDSF = CACHE ? LDS_FLOATS_MAX : 3 * THREADS;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  const Layout& L = a.L;
  const int seg = blockIdx.x;
  const int n0 = a.seg_ptr[seg], n = a.seg_ptr[seg + 1] - n0;
  const int tid = threadIdx.x;
  float* hA = lds;
  float* hB = hA + (size_t)n * 2 * NC;
  float* sa = hB + (size_t)n * NC;
  float* sd = sa + (size_t)n * 2;
  const float* P = a.params;
  float* sc = a.scratch;
  int stamp_i = 0;
  STAMP();

  if (a.phases & GATRES_PHASE_FORWARD) {
    float* xa = sc + L.sc_xa;
    float* xb = sc + L.sc_xb;
    float* xcur = a.saved ? a.saved + L.s_xin : xa;
    {  // lin0 (+ the caller-side x[mask] = 0)
      const float* w = P + L.p_lin0_w;
      const float* b = P + L.p_lin0_b;
      for (int idx = tid; idx < n * (NC / 4); idx += THREADS) {
        const int r = idx / (NC / 4), c0 = (idx % (NC / 4)) * 4;
        const size_t node = (size_t)n0 + r;
        const float xv = (a.mask && a.mask[node]) ? 0.f : a.x[node];
        const float4 wv = ld4(w + c0), bv = ld4(b + c0);
        float4 o;
        o.x = xv * wv.x + bv.x; o.y = xv * wv.y + bv.y; o.z = xv * wv.z + bv.z; o.w = xv * wv.w + bv.w;
        st4(xcur + node * NC + c0, o);
      }
    }
    __syncthreads();
    STAMP();
    for (int b = 0; b < L.nb; ++b) {
      float* base = a.saved ? a.saved + (int64_t)b * L.s_stride : sc + L.sc_ev;
      float* xnext = a.saved ? a.saved + (int64_t)(b + 1) * L.s_stride + L.s_xin : (xcur == xa ? xb : xa);
      const float* pb = P + L.p_block0 + (int64_t)b * L.p_block_stride;
      float* y2g = sc + L.sc_y2;
      // conv1: K1 then K2(+bias+ReLU)
      seg_proj<NC, 2 * NC, 2, EPI_ATT, THREADS>(n0, n, xcur, pb + L.c1_W, base + L.s_h1, CACHE ? hA : nullptr,
                                                 pb + L.c1_as, pb + L.c1_ad, base + L.s_as1, base + L.s_ad1,
                                                 CACHE ? sa : nullptr, CACHE ? sd : nullptr, nullptr, nullptr);
      __syncthreads();
      STAMP();
      if (CACHE)
        seg_agg_fwd<true, 2, NC, THREADS>(n0, n, a.rowptr, a.col, hA, n0, sa, sd, n0, pb + L.c1_b, base + L.s_o1, 0,
                                          base + L.s_al1);
      else
        seg_agg_fwd<true, 2, NC, THREADS>(n0, n, a.rowptr, a.col, base + L.s_h1, 0, base + L.s_as1, base + L.s_ad1, 0,
                                          pb + L.c1_b, base + L.s_o1, 0, base + L.s_al1);
      __syncthreads();
      STAMP();
      // conv2
      seg_proj<2 * NC, NC, 1, EPI_ATT, THREADS>(n0, n, base + L.s_o1, pb + L.c2_W, base + L.s_h2, CACHE ? hB : nullptr,
                                                 pb + L.c2_as, pb + L.c2_ad, base + L.s_as2, base + L.s_ad2,
                                                 CACHE ? sa : nullptr, CACHE ? sd : nullptr, nullptr, nullptr);
      __syncthreads();
      STAMP();
      if (CACHE)
        seg_agg_fwd<false, 1, NC, THREADS>(n0, n, a.rowptr, a.col, hB, n0, sa, sd, n0, pb + L.c2_b, hA, n0,
                                           base + L.s_al2);
      else
        seg_agg_fwd<false, 1, NC, THREADS>(n0, n, a.rowptr, a.col, base + L.s_h2, 0, base + L.s_as2, base + L.s_ad2, 0,
                                           pb + L.c2_b, y2g, 0, base + L.s_al2);
      __syncthreads();
      STAMP();
      // K3
      if (CACHE)
        seg_mean_fwd<NC, THREADS>(n0, n, a.m_rowptr, a.m_col, hA, n0, xcur, xnext);
      else
        seg_mean_fwd<NC, THREADS>(n0, n, a.m_rowptr, a.m_col, y2g, 0, xcur, xnext);
      __syncthreads();
      STAMP();
      xcur = xnext;
    }
    {  // lin1
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin1_w + (tid % G) * 4);
      const float bias = P[L.p_lin1_b];
      const int rounds = (n + THREADS / G - 1) / (THREADS / G);
      for (int it = 0; it < rounds; ++it) {
        int r = it * (THREADS / G) + tid / G;
        const bool valid = r < n;
        if (!valid) r = n - 1;
        const float4 xv = ld4(xcur + ((size_t)n0 + r) * NC + (tid % G) * 4);
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) a.out[n0 + r] = d + bias;
      }
    }
    __syncthreads();
    STAMP();
  }

  if (a.phases & PH_LOSS) {
    // M = number of masked nodes in the WHOLE batch (every workgroup counts them itself: N bytes from L2)
    float cnt = 0.f;
    for (int i = tid; i < a.N; i += THREADS) cnt += a.mask[i] ? 1.f : 0.f;
    const float M = block_sum<THREADS>(cnt, lds);
    float part = 0.f;
    for (int r = tid; r < n; r += THREADS) {
      const int node = n0 + r;
      if (a.mask[node]) {
        const float d = a.out[node] - a.y[node];
        part = fmaf(d, d, part);
      }
    }
    part = block_sum<THREADS>(part, lds);
    if (tid == 0) {
      a.loss_part[seg] = part;
      if (seg == 0) a.loss_part[gridDim.x] = M;
    }
    const float scale = M > 0.f ? 2.f / M : 0.f;
    for (int r = tid; r < n; r += THREADS) {
      const int node = n0 + r;
      a.g_out[node] = a.mask[node] ? (a.out[node] - a.y[node]) * scale : 0.f;
    }
    __syncthreads();
    STAMP();
  }

  if (a.phases & GATRES_PHASE_BACKWARD) {
    const float* saved = a.saved;
    float* gp_cur = sc + L.sc_gpa;
    float* gp_nxt = sc + L.sc_gpb;
    float* gy2 = sc + L.sc_gy2;
    float* ge = sc + L.sc_ge;
    float* gad = sc + L.sc_gad;
    float* gas = sc + L.sc_gas;
    float* gh = sc + L.sc_gh;
    float* go1 = sc + L.sc_go1;
    float* ge2 = sc + L.sc_ge2;
    float* gad2 = sc + L.sc_gad2;
    float* gas2 = sc + L.sc_gas2;
    float* gh2 = sc + L.sc_gh2;
    float* slab = a.slabs + (int64_t)seg * L.slab_stride;
    const int64_t w = 2LL * NC * NC;
    const float* xfinal = saved + (int64_t)L.nb * L.s_stride + L.s_xin;
    seg_lin1_bwd<NC, THREADS>(n0, n, a.g_out, xfinal, P + L.p_lin1_w, gp_cur, slab + L.p_lin1_w, slab + L.p_lin1_b,
                              L.nb > 0 ? 1 : 0, lds);
    __syncthreads();
    STAMP();
    for (int b = L.nb - 1; b >= 0; --b) {
      const float* base = saved + (int64_t)b * L.s_stride;
      const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
      const float* pb = P + po;
      float* sb = slab + po;
      const float* wt1 = a.wt + (int64_t)b * 2 * w;
      const float* wt2 = wt1 + w;
      seg_mean_bwd<NC, THREADS>(n0, n, a.m_rowptr, a.mt_rowptr, a.mt_dst, gp_cur, gy2);
      __syncthreads();
      STAMP();
      // conv2
      seg_agg_bwd_dst<1, NC, THREADS>(n0, n, a.rowptr, a.col, gy2, base + L.s_h2, base + L.s_al2, base + L.s_as2,
                                      base + L.s_ad2, ge2, gad2);
      __syncthreads();
      STAMP();
      seg_agg_bwd_src<1, NC, THREADS>(n0, n, a.t_rowptr, a.t_eid, a.t_dst, gy2, base + L.s_al2, ge2, gad2, pb + L.c2_as,
                                      pb + L.c2_ad, gh2, gas2);
      __syncthreads();
      STAMP();
      seg_conv_param_grads<1, NC, THREADS>(n0, n, base + L.s_h2, gas2, gad2, gy2, sb + L.c2_as, sb + L.c2_ad,
                                           sb + L.c2_b, lds);
      STAMP();
      seg_dw<NC, 2 * NC, THREADS>(n0, n, gh2, base + L.s_o1, sb + L.c2_W);
      STAMP();
      seg_proj<NC, 2 * NC, 1, EPI_RESID_MASK, THREADS>(n0, n, gh2, wt2, go1, nullptr, nullptr, nullptr, nullptr,
                                                        nullptr, nullptr, nullptr, nullptr, base + L.s_o1);
      __syncthreads();
      STAMP();
      // conv1
      seg_agg_bwd_dst<2, NC, THREADS>(n0, n, a.rowptr, a.col, go1, base + L.s_h1, base + L.s_al1, base + L.s_as1,
                                      base + L.s_ad1, ge, gad);
      __syncthreads();
      STAMP();
      seg_agg_bwd_src<2, NC, THREADS>(n0, n, a.t_rowptr, a.t_eid, a.t_dst, go1, base + L.s_al1, ge, gad, pb + L.c1_as,
                                      pb + L.c1_ad, gh, gas);
      __syncthreads();
      STAMP();
      seg_conv_param_grads<2, NC, THREADS>(n0, n, base + L.s_h1, gas, gad, go1, sb + L.c1_as, sb + L.c1_ad,
                                           sb + L.c1_b, lds);
      STAMP();
      seg_dw<2 * NC, NC, THREADS>(n0, n, gh, base + L.s_xin, sb + L.c1_W);
      STAMP();
      seg_proj<2 * NC, NC, 1, EPI_RESID_MASK, THREADS>(n0, n, gh, wt1, gp_nxt, nullptr, nullptr, nullptr, nullptr,
                                                        nullptr, nullptr, nullptr, gp_cur,
                                                        b > 0 ? base + L.s_xin : nullptr);
      __syncthreads();
      STAMP();
      float* t = gp_cur; gp_cur = gp_nxt; gp_nxt = t;
    }
    seg_lin0_bwd<NC, THREADS>(n0, n, gp_cur, a.x, a.mask, slab + L.p_lin0_w, slab + L.p_lin0_b, lds);
    if (a.g_x) {
      constexpr int G = NC / 4;
      const float4 wv = ld4(P + L.p_lin0_w + (tid % G) * 4);
      const int rounds = (n + THREADS / G - 1) / (THREADS / G);
      for (int it = 0; it < rounds; ++it) {
        int r = it * (THREADS / G) + tid / G;
        const bool valid = r < n;
        if (!valid) r = n - 1;
        const float4 xv = ld4(gp_cur + ((size_t)n0 + r) * NC + (tid % G) * 4);
        float d = xv.x * wv.x;
        d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (valid && (tid % G) == 0) a.g_x[n0 + r] = d;
      }
    }
  }
}

// grads = sum of segment slabs (fixed order) ; optionally the Adam update and the loss finalisation in the same pass
__global__ __launch_bounds__(256) void reduce_adam_kernel(const float* __restrict__ slabs, int num_slabs,
                                                          long long stride, long long count,
                                                          float* __restrict__ grads, const float* loss_part,
                                                          float* loss, int do_adam, float* __restrict__ p,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          unsigned long long* __restrict__ step_counter, double lr,
                                                          double b1, double b2, double eps, double wd,
                                                          float grad_scale) {
  __shared__ float s_step_size, s_bc2_sqrt;
  if (do_adam && threadIdx.x == 0) {
    const double t = (double)(step_counter[0] + 1ULL);
    s_step_size = (float)(lr / (1.0 - pow(b1, t)));
    s_bc2_sqrt = (float)sqrt(1.0 - pow(b2, t));
  }
  if (loss_part && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < num_slabs; ++k) s += loss_part[k];
    loss[0] = s / loss_part[num_slabs];
  }
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < count) {
    float acc = 0.f;
    for (int s = 0; s < num_slabs; ++s) acc += slabs[(size_t)s * stride + idx];
    grads[idx] = acc;
    if (do_adam) {
      const float pv = p[idx];
      float gv = acc * grad_scale;
      gv = gv + (float)wd * pv;
      float mv = m[idx];
      mv = mv + (float)(1.0 - b1) * (gv - mv);
      const float vv = (float)b2 * v[idx] + (float)(1.0 - b2) * gv * gv;
      const float denom = sqrtf(vv) / s_bc2_sqrt + (float)eps;
      p[idx] = pv + (-s_step_size * mv) / denom;
      m[idx] = mv;
      v[idx] = vv;
    }
  }
  if (do_adam) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned long long done = atomicAdd(&step_counter[1], 1ULL);
      if (done == (unsigned long long)gridDim.x - 1ULL) {
        step_counter[1] = 0ULL;
        atomicAdd(&step_counter[0], 1ULL);
      }
    }
  }
}

template <int NC, int THREADS>
static int launch_fused(const FusedArgs& a, int num_segments, int max_seg, hipStream_t st) {
  const bool cache = (long long)max_seg * (3 * NC + 4) <= LDS_FLOATS_MAX && 3 * THREADS <= LDS_FLOATS_MAX &&
                     !getenv("GATRES_FUSED_NOCACHE");
  if (cache)
    hipLaunchKernelGGL((gatres_fused_kernel<NC, THREADS, true>), dim3(num_segments), dim3(THREADS), 0, st, a);
  else
    hipLaunchKernelGGL((gatres_fused_kernel<NC, THREADS, false>), dim3(num_segments), dim3(THREADS), 0, st, a);
  return gatres_launch_status();
}

}  // namespace

// C-ABI ------------------------------------------------------------------------------------------------------

extern "C" int gatres_fused_supported(const gatres_model_t* m, const gatres_graph_t* g) {
  if (!m || !g || g->num_segments <= 0 || !g->seg_ptr) return 0;
  if (g->max_segment_nodes > 4096) return 0;      // beyond this a snapshot should be spread over many CUs
  return (m->nc >= 4 && m->nc <= 128 && gatres_is_pow2(m->nc)) ? 1 : 0;
}

// Diagnostic: segment 0 of the next fused launches writes a 100 MHz wall-clock stamp at every stage boundary into
// stamps[0..capacity) (device memory).  Pass nullptr to switch it off.
extern "C" int gatres_fused_set_stamps(uint64_t* stamps, int32_t capacity) {
  g_stamps = reinterpret_cast<unsigned long long*>(stamps);
  g_stamp_cap = stamps ? capacity : 0;
  return 0;
}

extern "C" int gatres_fused_prepare_backward(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                             float* scratch, void* stream) {
  if (!m || !g || !params || !scratch) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout(m, g->num_nodes, g->num_edges_gat, g->num_segments, &L)) return GATRES_E_UNSUPPORTED;
  return gatres_transpose_conv_weights(params, scratch + L.sc_wt, L.nb, L.nc, stream);
}

extern "C" int gatres_fused_run(const gatres_model_t* m, const gatres_graph_t* g, const float* params,
                                const float* x, const uint8_t* mask, const float* y, float* out, float* g_out,
                                float* loss_part, float* g_x, float* saved, float* scratch, int32_t phases,
                                void* stream) {
  if (!m || !g || !params || !x || !scratch) return GATRES_E_BADARG;
  if (!gatres_fused_supported(m, g)) return GATRES_E_UNSUPPORTED;
  FusedArgs a;
  if (!make_layout(m, g->num_nodes, g->num_edges_gat, g->num_segments, &a.L)) return GATRES_E_UNSUPPORTED;
  if ((phases & GATRES_PHASE_FORWARD) && !out) return GATRES_E_BADARG;
  if ((phases & GATRES_PHASE_LOSS) && (!mask || !y || !out || !g_out || !loss_part)) return GATRES_E_BADARG;
  if ((phases & GATRES_PHASE_BACKWARD) && (!saved || !g_out)) return GATRES_E_BADARG;
  a.seg_ptr = g->seg_ptr;
  a.rowptr = g->rowptr; a.col = g->col; a.t_rowptr = g->t_rowptr; a.t_eid = g->t_eid; a.t_dst = g->t_dst;
  a.m_rowptr = g->m_rowptr; a.m_col = g->m_col; a.mt_rowptr = g->mt_rowptr; a.mt_dst = g->mt_dst;
  a.N = g->num_nodes;
  a.params = params; a.wt = scratch + a.L.sc_wt;
  a.x = x; a.mask = mask; a.y = y; a.out = out; a.g_out = g_out; a.loss_part = loss_part; a.g_x = g_x;
  a.saved = saved; a.scratch = scratch; a.slabs = scratch + a.L.sc_slabs;
  a.stamps = g_stamps; a.stamp_cap = g_stamp_cap;
  a.phases = (phases & (GATRES_PHASE_FORWARD | GATRES_PHASE_BACKWARD)) | ((phases & GATRES_PHASE_LOSS) ? PH_LOSS : 0);
  hipStream_t st = gatres_stream(stream);
  const int S = g->num_segments, mx = g->max_segment_nodes;
  switch (m->nc) {
    case 4: return launch_fused<4, 1024>(a, S, mx, st);
    case 8: return launch_fused<8, 1024>(a, S, mx, st);
    case 16: return launch_fused<16, 1024>(a, S, mx, st);
    case 32: return launch_fused<32, 1024>(a, S, mx, st);
    case 64: return launch_fused<64, 512>(a, S, mx, st);
    case 128: return launch_fused<128, 256>(a, S, mx, st);
  }
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_fused_finish(const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads,
                                   const float* loss_part, float* loss, int32_t do_adam, float* params,
                                   float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, float grad_scale, void* stream) {
  if (!m || !g || !scratch || !grads) return GATRES_E_BADARG;
  if (do_adam && (!params || !exp_avg || !exp_avg_sq || !step_counter)) return GATRES_E_BADARG;
  if ((loss_part == nullptr) != (loss == nullptr)) return GATRES_E_BADARG;
  Layout L;
  if (!make_layout(m, g->num_nodes, g->num_edges_gat, g->num_segments, &L)) return GATRES_E_UNSUPPORTED;
  hipLaunchKernelGGL(reduce_adam_kernel, dim3((unsigned)((L.P + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     scratch + L.sc_slabs, g->num_segments, (long long)L.slab_stride, (long long)L.P, grads,
                     loss_part, loss, do_adam, params, exp_avg, exp_avg_sq,
                     reinterpret_cast<unsigned long long*>(step_counter), lr, beta1, beta2, eps, weight_decay,
                     grad_scale);
  return gatres_launch_status();
}
