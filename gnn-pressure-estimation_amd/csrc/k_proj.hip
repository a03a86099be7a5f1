// Dense per-node projections on the gfx950 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32 == a k-ordered
// fmaf chain; gfx950 has no TF32/XF32 path, so this is what holds 1e-5 parity).
//
//   proj_kernel      OUT[n, m] = sum_k X[n, k] * Wm[m, k]          (K1 forward with Wm = lin.weight [HC, K];
//                                                                   K1 backward-dx with Wm = W^T [K, HC])
//                    + optional attention-logit epilogue  a_src[n,h] = sum_c OUT[n,hC+c]*att_src[hC+c]   (GATConv)
//                    + optional residual-add / ReLU-mask epilogue (backward)
//   dw_kernel        slab[s][c, k] = sum_{n in slab s} G[n, c] * X[n, k]   (K1 backward-dW partials)
//
// Tiling for wave64: the product is computed TRANSPOSED, D[m][n] = Wm . X^T, so that in the 16x16 accumulator
// layout (col = lane&15, row = 4*(lane>>4)+reg) a lane owns ONE node and 4 consecutive output features:
// the store is a float4 per lane and the attention dot product over features is a per-lane sum plus two
// cross-lane adds.  The four lane groups q = lane>>4 each own a contiguous quarter of K, so both operands are
// fed from per-lane contiguous float4 loads (the k order inside the MFMA chain is permuted, which a sum allows).
// One wave = 16 nodes x all M outputs: N = 12.4k nodes gives 776 waves, enough to cover the 1024 SIMDs' worth
// of a launch-bound problem better than 32-row tiles would.
#include <cstdlib>

#include "gatres_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_NONE = 0, EPI_ATT = 1, EPI_RESID_MASK = 2 };

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

template <int K, int M, int H, int EPI>
__global__ __launch_bounds__(256) void proj_kernel(const float* __restrict__ X, const float* __restrict__ Wm,
                                                   float* __restrict__ OUT, int N,
                                                   // EPI_ATT
                                                   const float* __restrict__ att_src,
                                                   const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                   float* __restrict__ a_dst,
                                                   // EPI_RESID_MASK (either may be null)
                                                   const float* __restrict__ resid,
                                                   const float* __restrict__ relu_ref) {
  constexpr int KQ = K / 4;                 // k values per lane group
  constexpr int NT = (M + 15) / 16;         // 16-row output tiles
  constexpr int SC = (KQ % 4 == 0) ? 4 : ((KQ % 2 == 0) ? 2 : 1);   // k values per W load
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int n0 = wave * 16;
  if (n0 >= N) return;
  const int i = lane & 15, q = lane >> 4;
  const int n = n0 + i;
  const int nl = n < N ? n : N - 1;

  float xf[KQ];
  load_frag<KQ>(X + (size_t)nl * K + q * KQ, xf);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < KQ; s += SC) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int m = t * 16 + i;
      const bool mok = (M % 16 == 0) || (m < M);
      float wf[SC];
      load_frag<SC>(Wm + (size_t)(mok ? m : 0) * K + q * KQ + s, wf);
#pragma unroll
      for (int u = 0; u < SC; ++u)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(mok ? wf[u] : 0.f, xf[s + u], acc[t], 0, 0, 0);
    }
  }

  if constexpr (EPI == EPI_ATT) {
    constexpr int C = M / H;
    float ps[H], pd[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int mb = t * 16 + q * 4;
      if ((M % 16 == 0) || (mb < M)) {
        const float4 as = ld4(att_src + mb), ad = ld4(att_dst + mb);
        const float ds = fmaf(acc[t][3], as.w, fmaf(acc[t][2], as.z, fmaf(acc[t][1], as.y, acc[t][0] * as.x)));
        const float dd = fmaf(acc[t][3], ad.w, fmaf(acc[t][2], ad.z, fmaf(acc[t][1], ad.y, acc[t][0] * ad.x)));
        const int hd = mb / C;               // C is a power of two >= 4, so a float4 never straddles heads
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
    if (q == 0 && n < N) {
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { a_src[n * H + hh] = ps[hh]; a_dst[n * H + hh] = pd[hh]; }
    }
  }

  if (n < N) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int mb = t * 16 + q * 4;
      if ((M % 16 == 0) || (mb < M)) {
        float4 o = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if constexpr (EPI == EPI_RESID_MASK) {
          if (resid) {
            const float4 r = ld4(resid + (size_t)n * M + mb);
            o.x = o.x + r.x; o.y = o.y + r.y; o.z = o.z + r.z; o.w = o.w + r.w;
          }
          if (relu_ref) {
            const float4 r = ld4(relu_ref + (size_t)n * M + mb);
            o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f;
            o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
          }
        }
        st4(OUT + (size_t)n * M + mb, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same projection for wide models on large graphs (nc >= 64, tens of thousands of rows): persistent workgroups,
// one per CU, that stage W (+ the attention vectors) in LDS ONCE and then walk their share of 16-row tiles.  proj_kernel
// re-reads the whole of W from L1/L2 for every tile (128 KB per 16 rows at nc = 128: five times the activation
// traffic) and ran at a quarter of the fp32 matrix rate; here W fragments come from LDS, every wave has a SIMD to
// itself (four waves, up to 512 VGPRs), and the MFMAs of one k step are issued across all output tiles before the next
// k step so consecutive instructions are independent.  Lane map and k order per output are those of proj_kernel:
// bit-identical results.
// ------------------------------------------------------------------------------------------------------
template <int K, int M, int H, int EPI>
__global__ __launch_bounds__(256) void proj_lds_kernel(const float* __restrict__ X, const float* __restrict__ Wm,
                                                       float* __restrict__ OUT, int N,
                                                       const float* __restrict__ att_src,
                                                       const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                       float* __restrict__ a_dst, const float* __restrict__ resid,
                                                       const float* __restrict__ relu_ref, int rows_per_wg) {
  constexpr int KQ = K / 4, NT = M / 16, SC = 4, KP = K + 4;        // K, M multiples of 16 here
  __shared__ __attribute__((aligned(16))) float wl[M * KP + 2 * M];
  for (int idx = threadIdx.x; idx < M * (K / 4); idx += 256) {
    const int m = idx / (K / 4), k4 = (idx % (K / 4)) * 4;
    st4(wl + m * KP + k4, ld4(Wm + (size_t)m * K + k4));
  }
  if constexpr (EPI == EPI_ATT) {
    for (int idx = threadIdx.x; idx < 2 * (M / 4); idx += 256) {
      const int which = idx / (M / 4), c4 = (idx % (M / 4)) * 4;
      st4(wl + M * KP + which * M + c4, ld4((which ? att_dst : att_src) + c4));
    }
  }
  __syncthreads();
  const float* attS = wl + M * KP;
  const float* attD = attS + M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(N, r0 + rows_per_wg);
  float xf[KQ], xn[KQ];
  if (r0 + wave * 16 < r1) load_frag<KQ>(X + (size_t)min(r0 + wave * 16 + i, r1 - 1) * K + q * KQ, xf);
  for (int n0 = r0 + wave * 16; n0 < r1; n0 += 64) {
    const int n = n0 + i;
    const bool nok = n < r1;
    // the next tile's x fragment is in flight while this tile's 8 * NT * ... MFMAs run
    if (n0 + 64 < r1) load_frag<KQ>(X + (size_t)min(n + 64, r1 - 1) * K + q * KQ, xn);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KQ; s += SC) {
      float wf[NT][SC];
#pragma unroll
      for (int t = 0; t < NT; ++t) load_frag<SC>(wl + (t * 16 + i) * KP + q * KQ + s, wf[t]);
#pragma unroll
      for (int u = 0; u < SC; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][u], xf[s + u], acc[t], 0, 0, 0);
    }
    if constexpr (EPI == EPI_ATT) {
      constexpr int C = M / H;
      float ps[H], pd[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int mb = t * 16 + q * 4;
        const float4 as = ld4(attS + mb), ad = ld4(attD + mb);
        const float ds = fmaf(acc[t][3], as.w, fmaf(acc[t][2], as.z, fmaf(acc[t][1], as.y, acc[t][0] * as.x)));
        const float dd = fmaf(acc[t][3], ad.w, fmaf(acc[t][2], ad.z, fmaf(acc[t][1], ad.y, acc[t][0] * ad.x)));
        const int hd = mb / C;
#pragma unroll
        for (int hh = 0; hh < H; ++hh)
          if (hh == hd) { ps[hh] += ds; pd[hh] += dd; }
      }
#pragma unroll
      for (int hh = 0; hh < H; ++hh) {
        ps[hh] += __shfl_xor(ps[hh], 16); ps[hh] += __shfl_xor(ps[hh], 32);
        pd[hh] += __shfl_xor(pd[hh], 16); pd[hh] += __shfl_xor(pd[hh], 32);
      }
      if (q == 0 && nok) {
#pragma unroll
        for (int hh = 0; hh < H; ++hh) { a_src[n * H + hh] = ps[hh]; a_dst[n * H + hh] = pd[hh]; }
      }
    }
    if (nok) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int mb = t * 16 + q * 4;
        float4 o = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        if constexpr (EPI == EPI_RESID_MASK) {
          if (resid) {
            const float4 r = ld4(resid + (size_t)n * M + mb);
            o.x = o.x + r.x; o.y = o.y + r.y; o.z = o.z + r.z; o.w = o.w + r.w;
          }
          if (relu_ref) {
            const float4 r = ld4(relu_ref + (size_t)n * M + mb);
            o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f;
            o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
          }
        }
        st4(OUT + (size_t)n * M + mb, o);
      }
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) xf[k] = xn[k];
  }
}

// ------------------------------------------------------------------------------------------------------
// dW partials.  One wave = one (slab, output block) pair; a slab is a contiguous range of nodes.  Per MFMA
// step the four lane groups q supply four consecutive nodes; lane i of a group loads VC consecutive features
// of G (A operand, rows c) and VK consecutive features of X (B operand, cols k), which are spread over VC*VK
// accumulator tiles with the permuted index maps  c = cb0 + VC*i + tc,  k = kb0 + VK*j + tk.
// ------------------------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, bool ok, float (&f)[V]) {
  if (!ok) {
#pragma unroll
    for (int v = 0; v < V; ++v) f[v] = 0.f;
    return;
  }
  load_frag<V>(p, f);
}

template <int HC, int K>
__global__ __launch_bounds__(256) void dw_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                 float* __restrict__ slab, int num_slabs, long long slab_stride,
                                                 int N, int nodes_per_slab) {
  constexpr int CB = HC < 64 ? (HC < 16 ? 16 : HC) : 64;   // rows of g_W per wave
  constexpr int KB = K < 64 ? (K < 16 ? 16 : K) : 64;      // cols of g_W per wave
  constexpr int VC = CB / 16, VK = KB / 16;
  constexpr int NCB = (HC + CB - 1) / CB, NKB = (K + KB - 1) / KB;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int s = wave / (NCB * NKB);
  if (s >= num_slabs) return;
  const int blk = wave % (NCB * NKB);
  const int cb0 = (blk / NKB) * CB, kb0 = (blk % NKB) * KB;
  const int i = lane & 15, q = lane >> 4;
  const int cl = cb0 + VC * i, kl = kb0 + VK * i;
  const bool cok = cl < HC, kok = kl < K;

  f32x4 acc[VC][VK];
#pragma unroll
  for (int a = 0; a < VC; ++a)
#pragma unroll
    for (int b = 0; b < VK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nbeg = s * nodes_per_slab;
  const int nend = min(N, nbeg + nodes_per_slab);

  constexpr int STEPS = 4;               // 16 rows of operands in flight per wave (same accumulation order as one by one)
  for (int nb = nbeg; nb < nend; nb += 4 * STEPS) {
    float a[STEPS][VC], b[STEPS][VK];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int n = nb + 4 * st + q;
      const bool nok = n < nend;
      load_vec<VC>(G + (size_t)(nok ? n : 0) * HC + cl, nok && cok, a[st]);
      load_vec<VK>(X + (size_t)(nok ? n : 0) * K + kl, nok && kok, b[st]);
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
      for (int tc = 0; tc < VC; ++tc)
#pragma unroll
        for (int tk = 0; tk < VK; ++tk)
          acc[tc][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][tc], b[st][tk], acc[tc][tk], 0, 0, 0);
  }

  float* out = slab + (size_t)s * slab_stride;
#pragma unroll
  for (int tc = 0; tc < VC; ++tc)
#pragma unroll
    for (int tk = 0; tk < VK; ++tk) {
      const int k = kb0 + VK * i + tk;        // accumulator column j = lane & 15
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cb0 + VC * (4 * q + r) + tc;
        if (c < HC && k < K) out[(size_t)c * K + k] = acc[tc][tk][r];
      }
    }
}

template <int K, int M, int H, int EPI>
int launch_proj(const float* X, const float* Wm, float* OUT, int N, const float* att_src, const float* att_dst,
                float* a_src, float* a_dst, const float* resid, const float* relu_ref, hipStream_t st) {
  const int waves = (N + 15) / 16;
  if constexpr (K % 16 == 0 && M % 16 == 0 && K * M >= 64 * 128 && (M * (K + 4) + 2 * M) * 4 <= 160 * 1024) {
    if (N >= 16384 && !getenv("GATRES_NO_PROJ_LDS")) {        // W staging per workgroup must amortise over many tiles
      const int grid = 256;                                   // one persistent workgroup per CU
      int rows = (N + grid - 1) / grid;
      rows = (rows + 63) & ~63;
      hipLaunchKernelGGL((proj_lds_kernel<K, M, H, EPI>), dim3((N + rows - 1) / rows), dim3(256), 0, st, X, Wm, OUT, N,
                         att_src, att_dst, a_src, a_dst, resid, relu_ref, rows);
      return gatres_launch_status();
    }
  }
  hipLaunchKernelGGL((proj_kernel<K, M, H, EPI>), dim3((waves + 3) / 4), dim3(256), 0, st, X, Wm, OUT, N, att_src,
                     att_dst, a_src, a_dst, resid, relu_ref);
  return gatres_launch_status();
}

template <int HC, int K>
int launch_dw(const float* G, const float* X, float* slab, int num_slabs, long long stride, int N, hipStream_t st) {
  constexpr int CB = HC < 64 ? (HC < 16 ? 16 : HC) : 64;
  constexpr int KB = K < 64 ? (K < 16 ? 16 : K) : 64;
  constexpr int NBLK = ((HC + CB - 1) / CB) * ((K + KB - 1) / KB);
  int nps = (N + num_slabs - 1) / num_slabs;
  nps = (nps + 3) & ~3;
  const long long waves = (long long)num_slabs * NBLK;
  hipLaunchKernelGGL((dw_kernel<HC, K>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, G, X, slab, num_slabs,
                     stride, N, nps);
  return gatres_launch_status();
}

}  // namespace

// (K, M) shapes that occur for nc in {4..128}: (nc, 2nc) and (2nc, nc)
#define GATRES_FOR_SHAPES(X_)                                                                     \
  X_(4, 8) X_(8, 4) X_(8, 16) X_(16, 8) X_(16, 32) X_(32, 16) X_(32, 64) X_(64, 32) X_(64, 128)   \
  X_(128, 64) X_(128, 256) X_(256, 128)

extern "C" int gatres_proj_attn_fwd(const float* x, const float* W, const float* att_src, const float* att_dst,
                                    float* h, float* a_src, float* a_dst, int32_t num_nodes, int32_t K, int32_t H,
                                    int32_t C, void* stream) {
  if (!x || !W || !att_src || !att_dst || !h || !a_src || !a_dst || num_nodes <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(x) || !gatres_aligned16(W) || !gatres_aligned16(h) || !gatres_aligned16(att_src) ||
      !gatres_aligned16(att_dst))
    return GATRES_E_BADARG;
  const int M = H * C;
  hipStream_t st = gatres_stream(stream);
#define CASE_(K_, M_)                                                                                         \
  if (K == K_ && M == M_) {                                                                                   \
    if (H == 1) return launch_proj<K_, M_, 1, EPI_ATT>(x, W, h, num_nodes, att_src, att_dst, a_src, a_dst,   \
                                                         nullptr, nullptr, st);                               \
    if (H == 2) return launch_proj<K_, M_, 2, EPI_ATT>(x, W, h, num_nodes, att_src, att_dst, a_src, a_dst,   \
                                                         nullptr, nullptr, st);                               \
  }
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_proj_bwd_dx(const float* g_h, const float* Wt, const float* resid, const float* relu_ref,
                                  float* g_x, int32_t num_nodes, int32_t K, int32_t HC, void* stream) {
  if (!g_h || !Wt || !g_x || num_nodes <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(g_h) || !gatres_aligned16(Wt) || !gatres_aligned16(g_x) || !gatres_aligned16(resid) ||
      !gatres_aligned16(relu_ref))
    return GATRES_E_BADARG;
  hipStream_t st = gatres_stream(stream);
  // reduction length = HC, outputs = K
#define CASE_(K_, M_)                                                                                          \
  if (HC == K_ && K == M_)                                                                                     \
    return launch_proj<K_, M_, 1, EPI_RESID_MASK>(g_h, Wt, g_x, num_nodes, nullptr, nullptr, nullptr, nullptr, \
                                                  resid, relu_ref, st);
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_proj_bwd_dw(const float* g_h, const float* x, float* slab_W, int32_t num_slabs,
                                  int64_t slab_stride, int32_t num_nodes, int32_t K, int32_t HC, void* stream) {
  if (!g_h || !x || !slab_W || num_nodes <= 0 || num_slabs <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(g_h) || !gatres_aligned16(x)) return GATRES_E_BADARG;
  hipStream_t st = gatres_stream(stream);
#define CASE_(K_, M_) \
  if (K == K_ && HC == M_) return launch_dw<M_, K_>(g_h, x, slab_W, num_slabs, slab_stride, num_nodes, st);
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}
