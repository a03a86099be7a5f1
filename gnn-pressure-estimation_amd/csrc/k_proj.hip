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
#include "k_conv_grads.h"

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
  {
    // eight 16-byte loads in flight per thread, then their LDS stores (one load-store pair per trip made the staging a chain
    // of M * K / 1024 dependent L2 round trips per workgroup)
    constexpr int CH = M * (K / 4), PER = (CH + 255) / 256, UB = PER < 8 ? PER : 8;
    for (int b0 = 0; b0 < PER; b0 += UB) {
      float4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 256 * (b0 + u);
        const int m = idx / (K / 4), k4 = (idx % (K / 4)) * 4;
        v[u] = (b0 + u < PER && idx < CH) ? ld4(Wm + (size_t)m * K + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 256 * (b0 + u);
        const int m = idx / (K / 4), k4 = (idx % (K / 4)) * 4;
        if (b0 + u < PER && idx < CH) st4(wl + m * KP + k4, v[u]);
      }
    }
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
template <int V>
__device__ __forceinline__ void load_vec(const gatres_bf16* __restrict__ p, bool ok, float (&f)[V]) {
  if (!ok) {
#pragma unroll
    for (int v = 0; v < V; ++v) f[v] = 0.f;
    return;
  }
  if constexpr (V % 4 == 0) {
#pragma unroll
    for (int v = 0; v < V; v += 4) {
      const float4 x = ldrow4(p + v);
      f[v] = x.x; f[v + 1] = x.y; f[v + 2] = x.z; f[v + 3] = x.w;
    }
  } else if constexpr (V % 2 == 0) {
#pragma unroll
    for (int v = 0; v < V; v += 2) {
      const unsigned u = *reinterpret_cast<const unsigned*>(p + v);
      f[v] = __uint_as_float(u << 16); f[v + 1] = __uint_as_float(u & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) f[v] = (float)p[v];
  }
}

// (T: storage type of G and X; the products and the slab stay fp32 -- v_mfma_f32_16x16x4_f32 on widened operands)
template <int HC, int K, typename T>
__global__ __launch_bounds__(256) void dw_kernel(const T* __restrict__ G, const T* __restrict__ X,
                                                 float* __restrict__ slab, int num_slabs, long long slab_stride,
                                                 int N, int nodes_per_slab, int dw_grid, gatres_conv_grads_co<T> cg) {
  if ((int)blockIdx.x >= dw_grid) {               // co-launched attention-vector / bias partials (see dw2d_bf16_kernel)
    conv_param_grads_co_run((int)blockIdx.x - dw_grid, cg, N);
    return;
  }
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

// ------------------------------------------------------------------------------------------------------
// bf16 projections (gatres_model_t.act_dtype == GATRES_DTYPE_BF16; BASELINE config 3): X, W, OUT (and resid / relu_ref)
// are bf16 in HBM, the products run on v_mfma_f32_16x16x32_bf16 with fp32 accumulators, the attention logits are taken
// from the fp32 accumulators BEFORE the output is rounded.  Same transposed formulation as above (D = W . X^T: a lane
// owns one node and four consecutive output features per 16-row output tile), so the epilogues carry over; per k step of
// 32 a lane feeds 8 consecutive k of its W row (A operand, LDS) and 8 consecutive k of its node's row (B operand, one
// 16-byte global load).  Persistent workgroups stage W once in LDS with rows padded by 16 bytes (row stride K*2 + 16:
// the 16 lanes of a ds_read_b128 group land on 16 different 16-byte slots).  At 16x the fp32 matrix rate the kernel is
// memory-bound: x is read once, W once per workgroup, h written once.
// ------------------------------------------------------------------------------------------------------
typedef gatres_bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Feature permutation inside the kernel: output tile t, accumulator row 4*qq + reg  <->  feature qq*(M/4) + 4*t + reg,
// i.e. lane group q owns the CONTIGUOUS quarter [q*M/4, (q+1)*M/4) of a node's output row and consecutive tiles are
// consecutive 4-feature pieces of it: two tiles make one 16-byte bf16 store, the four lane groups of a node write a whole
// row between them (with the natural map, feature 16*t + 4*q + reg, every store was an 8-byte piece 32 bytes apart).
// The permutation is applied where W rows are picked for the A operand; the arithmetic per output is unchanged.
template <int K, int M, int H, int EPI>
__global__ __launch_bounds__(256) void proj_bf16_kernel(const gatres_bf16* __restrict__ X,
                                                        const gatres_bf16* __restrict__ Wm,
                                                        gatres_bf16* __restrict__ OUT, int N,
                                                        const float* __restrict__ att_src,
                                                        const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                        float* __restrict__ a_dst, const gatres_bf16* __restrict__ resid,
                                                        const gatres_bf16* __restrict__ relu_ref, int rows_per_wg) {
  constexpr int KS = K / 32, NT = M / 16, KP = K + 8, MQ = M / 4;  // K, M multiples of 32 here; NT even
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gatres_bf16* wl = reinterpret_cast<gatres_bf16*>(smem);
  float* attl = reinterpret_cast<float*>(smem + (size_t)M * KP * 2);
  // LDS row 16*t + i holds W row (i>>2)*MQ + 4*t + (i&3): the permutation is applied while staging, so the 16 lanes of an
  // A-operand read touch 16 CONSECUTIVE LDS rows (stride K*2 + 16 bytes: conflict-free)
  {
    // eight 16-byte loads in flight per thread, then their LDS stores: written as one load-store pair per trip, the staging
    // was M * K / 2048 dependent L2 round trips per workgroup (16 for gatres_large), most of a two-tile workgroup's life
    constexpr int CH = M * (K / 8), PER = (CH + 255) / 256, UB = PER < 8 ? PER : 8;
    for (int b0 = 0; b0 < PER; b0 += UB) {
      uint4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 256 * (b0 + u);
        const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
        const int t = l >> 4, ii = l & 15;
        const int m = (ii >> 2) * MQ + 4 * t + (ii & 3);
        v[u] = (b0 + u < PER && idx < CH) ? *reinterpret_cast<const uint4*>(Wm + (size_t)m * K + k8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 256 * (b0 + u);
        const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
        if (b0 + u < PER && idx < CH) *reinterpret_cast<uint4*>(wl + l * KP + k8) = v[u];
      }
    }
  }
  if constexpr (EPI == EPI_ATT) {
    for (int idx = threadIdx.x; idx < 2 * (M / 4); idx += 256) {
      const int which = idx / (M / 4), c4 = (idx % (M / 4)) * 4;
      st4(attl + which * M + c4, ld4((which ? att_dst : att_src) + c4));
    }
  }
  __syncthreads();
  const float* attS = attl;
  const float* attD = attl + M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(N, r0 + rows_per_wg);
  bf16x8 xf[KS], xn[KS];
  auto load_x = [&](int n, bf16x8 (&f)[KS]) {
    const gatres_bf16* p = X + (size_t)n * K + q * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const bf16x8*>(p + s * 32);
  };
  if (r0 + wave * 16 < r1) load_x(min(r0 + wave * 16 + i, r1 - 1), xf);
  for (int n0 = r0 + wave * 16; n0 < r1; n0 += 64) {
    const int n = n0 + i;
    const bool nok = n < r1;
    if (n0 + 64 < r1) load_x(min(n + 64, r1 - 1), xn);          // next tile's rows in flight behind this tile's MFMAs
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (t * 16 + i) * KP + s * 32 + q * 8);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s], acc[t], 0, 0, 0);
      }
    }
    // acc[t][reg] = output feature q*MQ + 4*t + reg of node n
    if constexpr (EPI == EPI_ATT) {
      constexpr int C = M / H;
      float ps[H], pd[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int mb = q * MQ + 4 * t;
        const float4 as = ld4(attS + mb), ad = ld4(attD + mb);
        const float ds = fmaf(acc[t][3], as.w, fmaf(acc[t][2], as.z, fmaf(acc[t][1], as.y, acc[t][0] * as.x)));
        const float dd = fmaf(acc[t][3], ad.w, fmaf(acc[t][2], ad.z, fmaf(acc[t][1], ad.y, acc[t][0] * ad.x)));
        const int hd = mb / C;               // (a lane group's quarter row lies inside one head: H <= 4)
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
      for (int t = 0; t < NT; t += 2) {
        const int mb = q * MQ + 4 * t;               // 8 consecutive features: tiles t and t + 1
        float4 o0 = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        float4 o1 = make_float4(acc[t + 1][0], acc[t + 1][1], acc[t + 1][2], acc[t + 1][3]);
        if constexpr (EPI == EPI_RESID_MASK) {
          if (resid) {
            const float4 ra = ldrow4(resid + (size_t)n * M + mb), rb = ldrow4(resid + (size_t)n * M + mb + 4);
            o0.x += ra.x; o0.y += ra.y; o0.z += ra.z; o0.w += ra.w;
            o1.x += rb.x; o1.y += rb.y; o1.z += rb.z; o1.w += rb.w;
          }
          if (relu_ref) {
            const float4 ra = ldrow4(relu_ref + (size_t)n * M + mb), rb = ldrow4(relu_ref + (size_t)n * M + mb + 4);
            o0.x = ra.x > 0.f ? o0.x : 0.f; o0.y = ra.y > 0.f ? o0.y : 0.f;
            o0.z = ra.z > 0.f ? o0.z : 0.f; o0.w = ra.w > 0.f ? o0.w : 0.f;
            o1.x = rb.x > 0.f ? o1.x : 0.f; o1.y = rb.y > 0.f ? o1.y : 0.f;
            o1.z = rb.z > 0.f ? o1.z : 0.f; o1.w = rb.w > 0.f ? o1.w : 0.f;
          }
        }
        bf16x8 ob;
        ob[0] = (gatres_bf16)o0.x; ob[1] = (gatres_bf16)o0.y; ob[2] = (gatres_bf16)o0.z; ob[3] = (gatres_bf16)o0.w;
        ob[4] = (gatres_bf16)o1.x; ob[5] = (gatres_bf16)o1.y; ob[6] = (gatres_bf16)o1.z; ob[7] = (gatres_bf16)o1.w;
        *reinterpret_cast<bf16x8*>(OUT + (size_t)n * M + mb) = ob;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = xn[s];
  }
}

// The same product for wide outputs (M a multiple of 128: gatres_large), retiled in round 3.  proj_bf16_kernel gives a wave
// 16 rows x ALL M columns at once: 64 accumulator registers + operands = 256 VGPRs + AGPRs, ONE wave per SIMD, one
// 256-thread workgroup per CU and 194 of 256 CUs busy on C-Town batches of 128 -- nothing hid an HBM round trip
// (1.6 - 2.7 TB/s).  Here a wave still owns a 16-row tile, but walks its M columns in PASSES of 128 (32 accumulator
// registers, ~110 VGPRs: four waves per SIMD), re-using the tile's x fragment; a 512-thread workgroup shares ONE LDS copy
// of W, two workgroups share a CU, and the row tiles are dealt in contiguous, equal shares to all 512 workgroups: 3 104
// tiles = 6 or 7 per workgroup, one per wave, ONE round on every CU (16-row x 128-column wave tiles dealt round a grid were
// 1.5 rounds: half the chip idle in the second).  The tile's rows are requested before W is staged, the epilogue's operands
// before the MFMA chain: a wave's life is a chain of round trips (W, x, epilogue operands, store) and they now overlap.
// A pass covers whole heads for nc = 128 (C = 128): the attention logits reduce inside the wave, pass by pass.
// Feature permutation within a pass: accumulator tile pair (2 j, 2 j + 1) of lane group q holds the 8 features
// [128 p + 32 j + 8 q, + 8), so ONE 16-byte store / epilogue-operand load per lane covers 64 contiguous bytes of a row
// across the row's four lanes.  (Lane group q owning a contiguous 32 features, as in proj_bf16_kernel above, made every
// store a 16-byte write request of its own: 1.6 M requests per launch, one per L2-channel clock -- the store phase took
// 7 of the kernel's 17 us whether the output sat in HBM or in cache; tests/micro/proj_probe.py.)
template <int K, int M, int H, int EPI>
__global__ __launch_bounds__(512, 4) void proj_bf16_tile_kernel(const gatres_bf16* __restrict__ X,
                                                                const gatres_bf16* __restrict__ Wm,
                                                                gatres_bf16* __restrict__ OUT, int N,
                                                                const float* __restrict__ att_src,
                                                                const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                                float* __restrict__ a_dst,
                                                                const gatres_bf16* __restrict__ resid,
                                                                const gatres_bf16* __restrict__ relu_ref) {
  constexpr int KS = K / 32, KP = K + 8, WC = 128, NT = WC / 16, CG = M / WC, QF = WC / 4, C = M / H;
  static_assert(M % WC == 0 && K % 32 == 0, "passes of 128 columns");
  static_assert(EPI != EPI_ATT || (C % QF == 0 && (C <= WC ? WC % C == 0 : false)), "a pass covers whole heads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gatres_bf16* wl = reinterpret_cast<gatres_bf16*>(smem);
  float* attl = reinterpret_cast<float*>(smem + (size_t)M * KP * 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int ntiles = (N + 15) >> 4;
  // this workgroup's contiguous share of the row tiles, one tile per wave and trip
  const int t_lo = (int)((long long)ntiles * blockIdx.x / gridDim.x), t_hi = (int)((long long)ntiles * (blockIdx.x + 1) / gridDim.x);
  bf16x8 xf[KS];
  auto load_x = [&](int tile, bf16x8 (&f)[KS]) {
    const int n = min(tile * 16 + i, N - 1);
    const gatres_bf16* p = X + (size_t)n * K + q * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const bf16x8*>(p + s * 32);
  };
  int tile = t_lo + wave;
  if (tile < t_hi) load_x(tile, xf);                          // in flight while W is staged
  {
    // LDS row l = 128 p + 16 t + a holds W row 128 p + 32 (t >> 1) + 8 (a >> 2) + 4 (t & 1) + (a & 3) (see the header);
    // eight 16-byte loads in flight per thread, then their LDS stores
    constexpr int CH = M * (K / 8), PER = (CH + 511) / 512, UB = PER < 8 ? PER : 8;
    for (int b0 = 0; b0 < PER; b0 += UB) {
      uint4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 512 * (b0 + u);
        const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
        const int lw = l % WC, t = lw >> 4, a = lw & 15;
        const int m = (l / WC) * WC + 32 * (t >> 1) + 8 * (a >> 2) + 4 * (t & 1) + (a & 3);
        v[u] = (b0 + u < PER && idx < CH) ? *reinterpret_cast<const uint4*>(Wm + (size_t)m * K + k8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = threadIdx.x + 512 * (b0 + u);
        const int l = idx / (K / 8), k8 = (idx % (K / 8)) * 8;
        if (b0 + u < PER && idx < CH) *reinterpret_cast<uint4*>(wl + l * KP + k8) = v[u];
      }
    }
  }
  if constexpr (EPI == EPI_ATT) {
    for (int idx = threadIdx.x; idx < 2 * (M / 4); idx += 512) {
      const int which = idx / (M / 4), c4 = (idx % (M / 4)) * 4;
      st4(attl + which * M + c4, ld4((which ? att_dst : att_src) + c4));
    }
  }
  __syncthreads();
  const float* attS = attl;
  const float* attD = attl + M;
  bool first = true;
  for (; tile < t_hi; tile += 8) {
    const int n = tile * 16 + i;
    const bool nok = n < N;
    if (!first) load_x(tile, xf);
    first = false;
    const size_t rowo = (size_t)min(n, N - 1) * M + q * 8;
#pragma unroll
    for (int p = 0; p < CG; ++p) {
      const gatres_bf16* wbase = wl + (size_t)(p * WC + i) * KP + q * 8;
      // the epilogue's operands (ReLU reference / residual rows: 16 bytes per pair of tiles) are requested BEFORE the MFMA
      // chain: loaded in the epilogue they were one more exposed round trip per pass
      uint4 rraw[EPI == EPI_RESID_MASK ? NT / 2 : 1], mraw[EPI == EPI_RESID_MASK ? NT / 2 : 1];
      if constexpr (EPI == EPI_RESID_MASK) {
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
          if (resid) rraw[t / 2] = *reinterpret_cast<const uint4*>(resid + rowo + p * WC + 16 * t);
          if (relu_ref) mraw[t / 2] = *reinterpret_cast<const uint4*>(relu_ref + rowo + p * WC + 16 * t);
        }
      }
      f32x4 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // one k step: its eight W fragments (one batch of LDS reads), then its eight MFMAs.  The scheduling barrier keeps
        // the compiler from hoisting the reads of ALL k steps in front of the first MFMA (K / 32 x 32 registers: it did,
        // and spilled 260 of them at the 128-VGPR budget of four waves per SIMD)
        bf16x8 wf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)(t * 16) * KP + s * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], xf[s], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[t][reg] = output feature 128 p + 32 (t >> 1) + 8 q + 4 (t & 1) + reg of node n
      if constexpr (EPI == EPI_ATT) {
        constexpr int HP = WC / C;                   // heads per pass (1 for nc = 128)
        float ps[HP], pd[HP];
#pragma unroll
        for (int hh = 0; hh < HP; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int mb = p * WC + 32 * (t >> 1) + 8 * q + 4 * (t & 1);
          const float4 as = ld4(attS + mb), ad = ld4(attD + mb);
          ps[(32 * (t >> 1)) / C] += fmaf(acc[t][3], as.w, fmaf(acc[t][2], as.z, fmaf(acc[t][1], as.y, acc[t][0] * as.x)));
          pd[(32 * (t >> 1)) / C] += fmaf(acc[t][3], ad.w, fmaf(acc[t][2], ad.z, fmaf(acc[t][1], ad.y, acc[t][0] * ad.x)));
        }
#pragma unroll
        for (int hh = 0; hh < HP; ++hh) {
          ps[hh] += __shfl_xor(ps[hh], 16); ps[hh] += __shfl_xor(ps[hh], 32);
          pd[hh] += __shfl_xor(pd[hh], 16); pd[hh] += __shfl_xor(pd[hh], 32);
        }
        if (q == 0 && nok) {
#pragma unroll
          for (int hh = 0; hh < HP; ++hh) { a_src[n * H + p * HP + hh] = ps[hh]; a_dst[n * H + p * HP + hh] = pd[hh]; }
        }
      }
      if (nok) {
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
          float4 o0 = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
          float4 o1 = make_float4(acc[t + 1][0], acc[t + 1][1], acc[t + 1][2], acc[t + 1][3]);
          if constexpr (EPI == EPI_RESID_MASK) {
            auto widen = [](const uint4 u, float4& lo, float4& hi) {
              lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                               __uint_as_float(u.y & 0xffff0000u));
              hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                               __uint_as_float(u.w & 0xffff0000u));
            };
            if (resid) {
              float4 ra, rb;
              widen(rraw[t / 2], ra, rb);
              o0.x += ra.x; o0.y += ra.y; o0.z += ra.z; o0.w += ra.w;
              o1.x += rb.x; o1.y += rb.y; o1.z += rb.z; o1.w += rb.w;
            }
            if (relu_ref) {
              float4 ra, rb;
              widen(mraw[t / 2], ra, rb);
              o0.x = ra.x > 0.f ? o0.x : 0.f; o0.y = ra.y > 0.f ? o0.y : 0.f;
              o0.z = ra.z > 0.f ? o0.z : 0.f; o0.w = ra.w > 0.f ? o0.w : 0.f;
              o1.x = rb.x > 0.f ? o1.x : 0.f; o1.y = rb.y > 0.f ? o1.y : 0.f;
              o1.z = rb.z > 0.f ? o1.z : 0.f; o1.w = rb.w > 0.f ? o1.w : 0.f;
            }
          }
          bf16x8 ob;
          ob[0] = (gatres_bf16)o0.x; ob[1] = (gatres_bf16)o0.y; ob[2] = (gatres_bf16)o0.z; ob[3] = (gatres_bf16)o0.w;
          ob[4] = (gatres_bf16)o1.x; ob[5] = (gatres_bf16)o1.y; ob[6] = (gatres_bf16)o1.z; ob[7] = (gatres_bf16)o1.w;
          *reinterpret_cast<bf16x8*>(OUT + (size_t)n * M + p * WC + 16 * t + q * 8) = ob;
        }
      }
    }
  }
}

#if GATRES_DIAG      // (measured, not faster than proj_bf16_tile_kernel: DESIGN.md section 3.3; kept as the measured alternative)

#ifdef PROJ_STAMPS
__device__ unsigned long long g_pstamps[512 * 8 * 8];
#define PSTAMP(k) do { if (lane == 0) g_pstamps[(blockIdx.x * 8 + wave) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PSTAMP(k) do {} while (0)
#endif
// Round 3, second retile: the STREAMING form for K = 128 / 256 (gatres_large).  What the stamps inside
// proj_bf16_tile_kernel's successor showed (tests/micro/proj_probe.py --stamps; profiles/r03_proj_probe.txt): of a launch's
// 15 us, 3 are dispatch, and 4 - 7 go by before the first MFMA because sixteen waves per CU each fetch W (or their slice of it)
// from L2 -- 128 - 256 KB per CU, 33 - 66 MB chip-wide, every CU asking for the same 64 KB at the same moment -- while the
// rows themselves (12.7 - 25 MB from HBM) land in the shadow of that; the stores drain within 0.3 us of the last MFMA.
// So: ONE 512-thread workgroup per CU, and
//   * a wave owns a 32-COLUMN slice of W for the whole launch, in REGISTERS (2 column tiles x K / 32 fragments = K / 4
//     VGPRs, loaded once: 64 KB per CU for M = 256, 128 KB for M = 128, where two groups of four waves take even / odd
//     tiles), and walks the row tiles of its workgroup: per tile K / 32 LDS reads of x, 2 K / 32 MFMAs, one 16-byte store
//     per lane (the four lanes of a row write 64 contiguous bytes);
//   * the workgroup's x tiles arrive by LDS-DMA in STAGES of eight (tile j by wave j % 8): the first stage is waited for
//     with a counted vmcnt while the second is still in flight, and is computed while that lands;
//   * the LDS image is swizzled on the SOURCE side: the DMA writes 64 lanes x 16 bytes lane-linearly, so lane L fetches the
//     16-byte chunk that belongs in its slot -- slot (row r, c') holds chunk c' ^ (r & 15) -- and the B-operand read of
//     lane (i, q) at k step s takes slot (4 s + q) ^ i: conflict-free for ds_read_b128's lane groups (rows are 256 or 512
//     bytes: unswizzled, all 16 rows of a fragment sit on one bank group);
//   * the epilogue operands of the dX forms (residual / ReLU reference rows, 16 bytes per lane and tile) of a stage are
//     requested before the stage's wait, all at once (two waves per SIMD: 256 registers each);
//   * attention logits: a head spans C / 32 waves; each wave leaves its 32-column partial dots in LDS and, behind the
//     round's closing barrier, one thread per (row, head) adds the head's partials in wave order.
// Feature map inside a wave's slice: accumulator (ct, reg) of lane group q = feature cb + 8 q + 4 ct + reg.
// Shares longer than 16 tiles (50 k-node graphs) go round by round.
template <int K, int M, int H, int EPI>
__global__ __launch_bounds__(512, 2) void proj_bf16_stream_kernel(const gatres_bf16* __restrict__ X,
                                                                  const gatres_bf16* __restrict__ Wm,
                                                                  gatres_bf16* __restrict__ OUT, int N,
                                                                  const float* __restrict__ att_src,
                                                                  const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                                  float* __restrict__ a_dst,
                                                                  const gatres_bf16* __restrict__ resid,
                                                                  const gatres_bf16* __restrict__ relu_ref) {
  constexpr int KS = K / 32, G = M / 32, TG = 8 / G, ROWB = K * 2, TILEB = 16 * ROWB, C = M / H, WPH = C / 32;
  constexpr int TMAX = 16, SPW = 8 / TG;            // tiles per round; tiles of one stage per wave
  static_assert(K % 128 == 0 && (M == 128 || M == 256) && C % 32 == 0, "gatres_large's widths");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* part = reinterpret_cast<float2*>(smem + (size_t)TMAX * TILEB);       // [tile][wave][16]: partial attention dots
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int ntiles = (N + 15) >> 4;
  const int t_lo = (int)((long long)ntiles * blockIdx.x / gridDim.x), t_hi = (int)((long long)ntiles * (blockIdx.x + 1) / gridDim.x);
  const int tg = wave / G, cb = (wave % G) * 32;                   // tile group; first column of this wave's slice
  PSTAMP(0);
  // the W slice: A-operand row a = lane i of column tile ct  <->  feature cb + 8 (i >> 2) + 4 ct + (i & 3)
  bf16x8 wr[2][KS];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const gatres_bf16* wp = Wm + (size_t)(cb + 8 * (i >> 2) + 4 * ct + (i & 3)) * K + q * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) wr[ct][s] = *reinterpret_cast<const bf16x8*>(wp + s * 32);
  }
  float4 aS[2], aD[2];
  if constexpr (EPI == EPI_ATT) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { aS[ct] = ld4(att_src + cb + 8 * q + 4 * ct); aD[ct] = ld4(att_dst + cb + 8 * q + 4 * ct); }
  }
  constexpr int ROWS = 1024 / ROWB, PER = TILEB / 1024, CPR = ROWB / 16;      // rows / DMA instructions per tile, chunks per row
  for (int base = t_lo; base < t_hi; base += TMAX) {
    const int T = min(TMAX, t_hi - base);
    // LDS-DMA of tile j (by wave j % 8): 1 KB (ROWS rows) per instruction, source-side swizzle
    auto dma_tile = [&](int j) {
      unsigned char* dst = smem + (size_t)j * TILEB;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int r = u * ROWS + lane / CPR, cs = lane % CPR;              // slot (r, cs)
        const int n = min((base + j) * 16 + r, N - 1);
        const gatres_bf16* src = X + (size_t)n * K + ((cs ^ (r & 15)) * 8);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), reinterpret_cast<float*>(dst + u * 1024), 16, 0, 0);
      }
    };
    uint4 rr[SPW], mr[SPW];
    auto epi_load = [&](int stage) {                 // this wave's tiles of the stage: j = 8 stage + tg + TG k
      if constexpr (EPI == EPI_RESID_MASK) {
#pragma unroll
        for (int k = 0; k < SPW; ++k) {
          const int j = 8 * stage + tg + TG * k;
          const size_t o = (size_t)min((base + min(j, T - 1)) * 16 + i, N - 1) * M + cb + 8 * q;
          if (resid) rr[k] = *reinterpret_cast<const uint4*>(resid + o);
          if (relu_ref) mr[k] = *reinterpret_cast<const uint4*>(relu_ref + o);
        }
      }
    };
    if (wave < T) dma_tile(wave);
    epi_load(0);
    const bool second = wave + 8 < T;                // (wave-uniform)
    if (second) dma_tile(wave + 8);
    // everything but this wave's second-stage tile has landed (loads return in order)
    if (second) { if constexpr (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(PER == 4 || PER == 8, "the counted wait above");
    PSTAMP(1);
    __builtin_amdgcn_s_barrier();                    // (bare: a fence here would drain the second stage's DMA)
    PSTAMP(2);
    for (int stage = 0; stage * 8 < T; ++stage) {
      if (stage == 1) {
        epi_load(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PSTAMP(3);
      }
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const int j = 8 * stage + tg + TG * k;
        if (j >= T) break;
        const int n = (base + j) * 16 + i;
        const bool nok = n < N;
        const unsigned char* xrow = smem + j * TILEB + i * ROWB;
        bf16x8 xf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xrow + (((4 * s + q) ^ i) * 16));
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[0][s], xf[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[1][s], xf[s], acc[1], 0, 0, 0);
        }
        if constexpr (EPI == EPI_ATT) {
          float ps = fmaf(acc[0][3], aS[0].w, fmaf(acc[0][2], aS[0].z, fmaf(acc[0][1], aS[0].y, acc[0][0] * aS[0].x)));
          float pd = fmaf(acc[0][3], aD[0].w, fmaf(acc[0][2], aD[0].z, fmaf(acc[0][1], aD[0].y, acc[0][0] * aD[0].x)));
          ps += fmaf(acc[1][3], aS[1].w, fmaf(acc[1][2], aS[1].z, fmaf(acc[1][1], aS[1].y, acc[1][0] * aS[1].x)));
          pd += fmaf(acc[1][3], aD[1].w, fmaf(acc[1][2], aD[1].z, fmaf(acc[1][1], aD[1].y, acc[1][0] * aD[1].x)));
          ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
          pd += __shfl_xor(pd, 16); pd += __shfl_xor(pd, 32);
          if (q == 0) part[(j * 8 + wave) * 16 + i] = make_float2(ps, pd);
        }
        float4 o0 = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
        float4 o1 = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        if constexpr (EPI == EPI_RESID_MASK) {
          auto widen = [](const uint4 u, float4& lo, float4& hi) {
            lo = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xffff0000u));
            hi = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                             __uint_as_float(u.w & 0xffff0000u));
          };
          if (resid) {
            float4 ra, rb;
            widen(rr[k], ra, rb);
            o0.x += ra.x; o0.y += ra.y; o0.z += ra.z; o0.w += ra.w;
            o1.x += rb.x; o1.y += rb.y; o1.z += rb.z; o1.w += rb.w;
          }
          if (relu_ref) {
            float4 ra, rb;
            widen(mr[k], ra, rb);
            o0.x = ra.x > 0.f ? o0.x : 0.f; o0.y = ra.y > 0.f ? o0.y : 0.f;
            o0.z = ra.z > 0.f ? o0.z : 0.f; o0.w = ra.w > 0.f ? o0.w : 0.f;
            o1.x = rb.x > 0.f ? o1.x : 0.f; o1.y = rb.y > 0.f ? o1.y : 0.f;
            o1.z = rb.z > 0.f ? o1.z : 0.f; o1.w = rb.w > 0.f ? o1.w : 0.f;
          }
        }
        if (nok) {
          bf16x8 ob;
          ob[0] = (gatres_bf16)o0.x; ob[1] = (gatres_bf16)o0.y; ob[2] = (gatres_bf16)o0.z; ob[3] = (gatres_bf16)o0.w;
          ob[4] = (gatres_bf16)o1.x; ob[5] = (gatres_bf16)o1.y; ob[6] = (gatres_bf16)o1.z; ob[7] = (gatres_bf16)o1.w;
          *reinterpret_cast<bf16x8*>(OUT + (size_t)n * M + cb + 8 * q) = ob;
        }
      }
    }
    PSTAMP(4);
    __syncthreads();                       // the partial dots are complete; the x tiles are dead (the next round overwrites them)
    if constexpr (EPI == EPI_ATT) {
      for (int idx = threadIdx.x; idx < T * 16 * H; idx += 512) {
        const int hd = idx % H, r = (idx / H) % 16, j = idx / (16 * H);
        const int n = (base + j) * 16 + r;
        const float2* pp = part + (j * 8 + (j % TG) * G + hd * WPH) * 16 + r;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < WPH; ++w) { s0 += pp[w * 16].x; s1 += pp[w * 16].y; }
        if (n < N) { a_src[n * H + hd] = s0; a_dst[n * H + hd] = s1; }
      }
      if (base + TMAX < t_hi) __syncthreads();
    }
  }
  PSTAMP(6);
}

#endif      // GATRES_DIAG

template <int K, int M, int H, int EPI>
int launch_proj_bf16(const gatres_bf16* X, const gatres_bf16* Wm, gatres_bf16* OUT, int N, const float* att_src,
                     const float* att_dst, float* a_src, float* a_dst, const gatres_bf16* resid,
                     const gatres_bf16* relu_ref, hipStream_t st) {
  if constexpr (K % 128 == 0 && (M == 128 || M == 256) && (M / H) % 32 == 0) {
#if GATRES_DIAG
    // gatres_large's widths: W slices in registers, x tiles streamed through LDS (proj_bf16_stream_kernel)
    if (!gatres_knobs()->proj_rows && gatres_knobs()->proj_stream) {
      constexpr int TILEB = 16 * K * 2, G = M / 32, TG = 8 / G;
      const int ntiles = (N + 15) / 16;
      int grid = (ntiles + TG - 1) / TG;
      if (grid > 256) grid = 256;                 // one workgroup per CU, contiguous equal shares of the row tiles
      constexpr size_t lds = (size_t)16 * TILEB + (EPI == EPI_ATT ? (size_t)16 * 8 * 16 * 8 : 0);
      static bool attr_set = false;
      if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_bf16_stream_kernel<K, M, H, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          (void)hipGetLastError();
        attr_set = true;
      }
      hipLaunchKernelGGL((proj_bf16_stream_kernel<K, M, H, EPI>), dim3(grid), dim3(512), lds, st, X, Wm, OUT, N, att_src,
                         att_dst, a_src, a_dst, resid, relu_ref);
      return gatres_launch_status();
    }
#endif
  }
  if constexpr (K % 32 == 0 && M % 128 == 0 && M <= 256 && (EPI != EPI_ATT || (M / H) == 128 || (M / H) == 64 || (M / H) == 32)) {
    // wide outputs: 16-row wave tiles walked in passes of 128 columns, 512-thread workgroups, two per CU (proj_bf16_tile_kernel)
    if (!gatres_knobs()->proj_rows) {
      constexpr size_t lds = (size_t)M * (K + 8) * 2 + 2 * M * 4;
      static bool attr_set = false;
      if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_bf16_tile_kernel<K, M, H, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          (void)hipGetLastError();
        attr_set = true;
      }
      const int ntiles = (N + 15) / 16;
      int grid = (ntiles + 7) / 8;
      if (grid > 512) grid = 512;                 // two resident workgroups per CU, contiguous equal shares of the row tiles
      hipLaunchKernelGGL((proj_bf16_tile_kernel<K, M, H, EPI>), dim3(grid), dim3(512), lds, st, X, Wm, OUT, N, att_src,
                         att_dst, a_src, a_dst, resid, relu_ref);
      return gatres_launch_status();
    }
  }
  if constexpr (K % 32 == 0 && M % 32 == 0 && (H == 1 || H == 2 || H == 4)) {   // (a lane group's quarter row inside one head)
    constexpr size_t lds = (size_t)M * (K + 8) * 2 + 2 * M * 4;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_bf16_kernel<K, M, H, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        (void)hipGetLastError();
      attr_set = true;
    }
    // persistent workgroups: W staging must amortise over many 16-row tiles, and small batches still want every CU
    // (at most one workgroup per CU: 388 workgroups of 128 rows -- two on some CUs, one on others -- ran 3.5 % of a
    //  gatres_large step slower than 194 of 256 rows)
    int grid = (N + 63) / 64;
    if (grid > 256) grid = 256;
    int rows = (N + grid - 1) / grid;
    rows = (rows + 63) & ~63;
    if (const int pr = gatres_knobs()->proj_rows) rows = pr;      // (tuning experiments)
    if (rows < 64) rows = 64;
    hipLaunchKernelGGL((proj_bf16_kernel<K, M, H, EPI>), dim3((N + rows - 1) / rows), dim3(256), lds, st, X, Wm, OUT, N,
                       att_src, att_dst, a_src, a_dst, resid, relu_ref, rows);
    return gatres_launch_status();
  } else {
    return GATRES_E_UNSUPPORTED;      // bf16 MFMA needs a reduction length that is a multiple of 32 (nc >= 32)
  }
}

template <int K, int M, int H, int EPI>
int launch_proj(const float* X, const float* Wm, float* OUT, int N, const float* att_src, const float* att_dst,
                float* a_src, float* a_dst, const float* resid, const float* relu_ref, hipStream_t st) {
  const int waves = (N + 15) / 16;
  if constexpr (K % 16 == 0 && M % 16 == 0 && K * M >= 64 * 128 && (M * (K + 4) + 2 * M) * 4 <= 160 * 1024) {
    if (N >= 16384 && !gatres_knobs()->no_proj_lds) {        // W staging per workgroup must amortise over many tiles
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

// ------------------------------------------------------------------------------------------------------
// dW on the bf16 matrix cores: slab[s][c, k] = sum_{n in slab s} G[n, c] * X[n, k] with both operands bf16 in HBM.
// The reduction runs over NODES, the row index of both row-major operands, while v_mfma_f32_16x16x32_bf16 wants 8
// consecutive reduction indices per lane for A (= G^T) and for B (= X): a column access.  So a workgroup stages 32-node
// chunks of G and X TRANSPOSED in LDS ([feature][32 nodes], rows padded to 80 bytes: the 16 lanes of an operand read
// fall on 16 different 16-byte slots), coalesced 16-byte global loads in, 2-byte LDS stores out; every operand fragment
// is then one ds_read_b128.  One workgroup per node slab (the slabs of the fp32 kernel: same partition, same
// fixed-order slab reduction afterwards), wave w owns a CTW x KTW block of the [HC/16] x [K/16] output tiles in
// registers for the whole slab.  fp32 accumulation; at 16x the fp32 matrix rate the kernel is bound by its reads.
// ------------------------------------------------------------------------------------------------------
template <int HC, int K>
__global__ __launch_bounds__(256) void dw_bf16_kernel(const gatres_bf16* __restrict__ G, const gatres_bf16* __restrict__ X,
                                                      float* __restrict__ slab, long long slab_stride, int N,
                                                      int nodes_per_slab) {
  constexpr int NCT = HC / 16, NKT = K / 16, TOT = NCT * NKT, TPW = TOT / 4;      // tiles per wave (4 waves)
  constexpr int KTW = NKT < TPW ? NKT : TPW, CTW = TPW / KTW;                     // a wave's block of tiles
  // The reduction runs over NODES, the row index of both operands, so an MFMA lane needs 8 consecutive nodes of ONE
  // feature.  The 32-node chunk stays node-major in LDS, exactly as it lies in HBM (16-byte stores, no conflicts), and
  // the operands come out of it through gfx950's transposing LDS read (ds_read_b64_tr_b16: per 16-lane group a block of
  // 4 rows x 16 columns, delivered column-major): two reads per fragment.  (Before: the chunk was transposed on the way
  // IN, 2 bytes at a time -- with feature rows a multiple of 32 dwords apart all 32 lanes of a store hit one bank.)
  // Rows are padded by 16 elements: the four rows of a block fall into disjoint bank windows.
  constexpr int GP = HC + 16, XP = K + 16;
  constexpr int GLC = (32 * (HC / 8) + 255) / 256, XLC = (32 * (K / 8) + 255) / 256;   // 16-byte loads per thread and chunk
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  __shared__ __attribute__((aligned(16))) gatres_bf16 gimg[32 * GP];
  __shared__ __attribute__((aligned(16))) gatres_bf16 ximg[32 * XP];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int nbeg = s * nodes_per_slab, nend = min(N, nbeg + nodes_per_slab);
  const int t0 = wave * TPW, ct0 = t0 / NKT, kt0 = t0 % NKT;                      // block origin (KTW divides NKT)
  f32x4 acc[CTW][KTW];
#pragma unroll
  for (int a = 0; a < CTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 gv[GLC], xv[XLC];
  auto fetch = [&](int n0) {                         // the chunk's 16-byte pieces of this thread: registers
#pragma unroll
    for (int u = 0; u < GLC; ++u) {
      const int p = tid + 256 * u, nl = p / (HC / 8), c8 = (p % (HC / 8)) * 8, n = n0 + nl;
      gv[u] = (p < 32 * (HC / 8) && n < nend) ? *reinterpret_cast<const uint4*>(G + (size_t)n * HC + c8) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < XLC; ++u) {
      const int p = tid + 256 * u, nl = p / (K / 8), k8 = (p % (K / 8)) * 8, n = n0 + nl;
      xv[u] = (p < 32 * (K / 8) && n < nend) ? *reinterpret_cast<const uint4*>(X + (size_t)n * K + k8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  // this lane's block address inside a 16-column tile: lane 4qq + pp of the group supplies row qq, columns 4pp .. 4pp + 3
  const int qq = i >> 2, pp = i & 3;
  gatres_bf16* gbase = gimg + (8 * q + qq) * GP + ct0 * 16 + 4 * pp;
  gatres_bf16* xbase = ximg + (8 * q + qq) * XP + kt0 * 16 + 4 * pp;
  if (nbeg < nend) fetch(nbeg);
  for (int n0 = nbeg; n0 < nend; n0 += 32) {
#pragma unroll
    for (int u = 0; u < GLC; ++u) {
      const int p = tid + 256 * u, nl = p / (HC / 8), c8 = (p % (HC / 8)) * 8;
      if (p < 32 * (HC / 8)) *reinterpret_cast<uint4*>(gimg + nl * GP + c8) = gv[u];
    }
#pragma unroll
    for (int u = 0; u < XLC; ++u) {
      const int p = tid + 256 * u, nl = p / (K / 8), k8 = (p % (K / 8)) * 8;
      if (p < 32 * (K / 8)) *reinterpret_cast<uint4*>(ximg + nl * XP + k8) = xv[u];
    }
    __syncthreads();
    if (n0 + 32 < nend) fetch(n0 + 32);              // the next chunk's loads fly during this chunk's matrix work
    bf16x8 af[CTW], bfr[KTW];
#pragma unroll
    for (int a = 0; a < CTW; ++a) {                  // nodes 8q .. 8q + 3 | 8q + 4 .. 8q + 7 of feature (ct0 + a) * 16 + i
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(gbase + a * 16));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(gbase + 4 * GP + a * 16));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      af[a] = __builtin_bit_cast(bf16x8, both);
    }
#pragma unroll
    for (int b = 0; b < KTW; ++b) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xbase + b * 16));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xbase + 4 * XP + b * 16));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      bfr[b] = __builtin_bit_cast(bf16x8, both);
    }
#pragma unroll
    for (int a = 0; a < CTW; ++a)
#pragma unroll
      for (int b = 0; b < KTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    __syncthreads();
  }
  float* out = slab + (size_t)s * slab_stride;
#pragma unroll
  for (int a = 0; a < CTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)((ct0 + a) * 16 + 4 * q + r) * K + (kt0 + b) * 16 + i] = acc[a][b][r];
}
// The same product cut in TWO dimensions: workgroup (rg, ob) forms the 64 x 64 output block ob over the rows of row group
// rg and writes it into slab row rg.  With R row groups x (HC / 64)(K / 64) blocks filling the chip, a dW call writes R
// partial matrices instead of one per workgroup, and the final reduction reads R rows: for gatres_large (25 x 128, C-Town
// bs 128) the 256-row form wrote 33 MB per call and the reduction read 1.7 GB per step -- 1.6 of a 9-ms step.  Every
// operand row is read by (K / 64) resp. (HC / 64) workgroups, which run at the same time: L2 absorbs the re-reads.
// Co-launch: with cg.h set, workgroups beyond the first `dw_grid` form the convolution's attention-vector / bias gradient
// partials (conv_param_grads_bf16_body, slab cg.first + ...).  The two partial-sum launches of a convolution's backward
// read the same tables, write disjoint slab regions and each leave most of the chip waiting on memory: as one launch they
// overlap (as two parallel branches of the captured graph they cost more than they return).
typedef gatres_conv_grads_co<gatres_bf16> ConvGradsCo;

template <int HC, int K>
__global__ __launch_bounds__(256) void dw2d_bf16_kernel(const gatres_bf16* __restrict__ G, const gatres_bf16* __restrict__ X,
                                                        float* __restrict__ slab, long long slab_stride, int N,
                                                        int nodes_per_group, int dw_grid, ConvGradsCo cg) {
  if ((int)blockIdx.x >= dw_grid) {               // (workgroup-uniform)
    conv_param_grads_co_run((int)blockIdx.x - dw_grid, cg, N);
    return;
  }
  constexpr int OBK = K / 64, OB = (HC / 64) * OBK, CH = 64;          // 64-node chunks: two MFMA k steps
  constexpr int RP = 64 + 16;                                         // LDS row: 64 features + pad (see dw_bf16_kernel)
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  __shared__ __attribute__((aligned(16))) gatres_bf16 gimg[CH * RP];
  __shared__ __attribute__((aligned(16))) gatres_bf16 ximg[CH * RP];
  // The OB workgroups of a row group re-read its rows (each G row K / 64 times, each X row HC / 64 times): they are given
  // ids 8 apart, i.e. ONE XCD under round-robin dispatch, so that the re-reads hit its L2 (with consecutive ids the eight
  // sat on eight XCDs and every one of them fetched the rows from HBM: 110 MB per call instead of ~45).  Speed only.
  int rg = blockIdx.x / OB, ob = blockIdx.x % OB;
  {
    const int groups = dw_grid / OB;
    if ((groups & 7) == 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      rg = (j / OB) * 8 + xcd; ob = j % OB;
    }
  }
  const int cb = ob / OBK, kb = ob % OBK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4, qq = i >> 2, pp = i & 3;
  const int nbeg = rg * nodes_per_group, nend = min(N, nbeg + nodes_per_group);
  const gatres_bf16* Gc = G + cb * 64;
  const gatres_bf16* Xc = X + kb * 64;
  f32x4 acc[4];                                                       // c tile `wave` of the block x its four k tiles
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 gv[2], xv[2];                                                 // 64 nodes x 8 pieces of 16 bytes = 2 per thread and table
  auto fetch = [&](int n0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = tid + 256 * u, nl = p >> 3, c8 = (p & 7) * 8, n = n0 + nl;
      gv[u] = n < nend ? *reinterpret_cast<const uint4*>(Gc + (size_t)n * HC + c8) : make_uint4(0u, 0u, 0u, 0u);
      xv[u] = n < nend ? *reinterpret_cast<const uint4*>(Xc + (size_t)n * K + c8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  gatres_bf16* gbase = gimg + (8 * q + qq) * RP + wave * 16 + 4 * pp;
  gatres_bf16* xbase = ximg + (8 * q + qq) * RP + 4 * pp;
  if (nbeg < nend) fetch(nbeg);
  for (int n0 = nbeg; n0 < nend; n0 += CH) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = tid + 256 * u, nl = p >> 3, c8 = (p & 7) * 8;
      *reinterpret_cast<uint4*>(gimg + nl * RP + c8) = gv[u];
      *reinterpret_cast<uint4*>(ximg + nl * RP + c8) = xv[u];
    }
    __syncthreads();
    if (n0 + CH < nend) fetch(n0 + CH);
#pragma unroll
    for (int ks = 0; ks < CH / 32; ++ks) {
      const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(gbase + (32 * ks) * RP));
      const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(gbase + (32 * ks + 4) * RP));
      const bf16x8 af = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xbase + (32 * ks) * RP + b * 16));
        const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xbase + (32 * ks + 4) * RP + b * 16));
        const bf16x8 bf = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* out = slab + (size_t)rg * slab_stride;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      out[(size_t)(cb * 64 + wave * 16 + 4 * q + r) * K + kb * 64 + b * 16 + i] = acc[b][r];
}

// cg / cg_slabs: the co-launched attention-vector / bias partials (nullptr: none).  GATRES_E_UNSUPPORTED with a co-launch
// asked for means "launch the two separately".
template <int HC, int K>
int launch_dw_bf16(const gatres_bf16* G, const gatres_bf16* X, float* slab, int num_slabs, long long stride, int N,
                   hipStream_t st, const ConvGradsCo* cg = nullptr, int cg_slabs = 0) {
  if constexpr (HC % 64 == 0 && K % 64 == 0) {
    // few slab rows asked for (the per-op driver does for wide models): rows x output blocks instead of rows only
    constexpr int OB = (HC / 64) * (K / 64);
    if (num_slabs * OB <= 1024 && num_slabs <= 128 && !gatres_knobs()->dw_1d) {
      int npg = (N + num_slabs - 1) / num_slabs;
      npg = (npg + 3) & ~3;
      ConvGradsCo co{};
      if (cg) co = *cg;
      const int dw_grid = num_slabs * OB;
      hipLaunchKernelGGL((dw2d_bf16_kernel<HC, K>), dim3(dw_grid + (cg ? cg_slabs : 0)), dim3(256), 0, st, G, X, slab,
                         stride, N, npg, dw_grid, co);
      return gatres_launch_status();
    }
  }
  if (cg) return GATRES_E_UNSUPPORTED;
  if constexpr (HC % 16 == 0 && K % 16 == 0 && (HC / 16) * (K / 16) >= 4 && ((HC / 16) * (K / 16)) % 4 == 0 &&
                (HC + K) * 80 <= 160 * 1024) {
    int nps = (N + num_slabs - 1) / num_slabs;
    nps = (nps + 3) & ~3;                         // the slab boundaries of every other parameter-gradient kernel
    hipLaunchKernelGGL((dw_bf16_kernel<HC, K>), dim3(num_slabs), dim3(256), 0, st, G, X, slab, stride, N, nps);
    return gatres_launch_status();
  } else {
    return GATRES_E_UNSUPPORTED;
  }
}

template <int HC, int K, typename T>
int launch_dw(const T* G, const T* X, float* slab, int num_slabs, long long stride, int N, hipStream_t st,
              const gatres_conv_grads_co<T>* cg = nullptr, int cg_slabs = 0) {
  constexpr int CB = HC < 64 ? (HC < 16 ? 16 : HC) : 64;
  constexpr int KB = K < 64 ? (K < 16 ? 16 : K) : 64;
  constexpr int NBLK = ((HC + CB - 1) / CB) * ((K + KB - 1) / KB);
  int nps = (N + num_slabs - 1) / num_slabs;
  nps = (nps + 3) & ~3;
  const long long waves = (long long)num_slabs * NBLK;
  gatres_conv_grads_co<T> co{};
  if (cg) co = *cg;
  const int dw_grid = (int)((waves + 3) / 4);
  hipLaunchKernelGGL((dw_kernel<HC, K, T>), dim3((unsigned)(dw_grid + (cg ? cg_slabs : 0))), dim3(256), 0, st, G, X, slab,
                     num_slabs, stride, N, nps, dw_grid, co);
  return gatres_launch_status();
}

}  // namespace

// (K, M) shapes that occur for nc in {4..128}: (nc, 2nc) and (2nc, nc)
#define GATRES_FOR_SHAPES(X_)                                                                     \
  X_(4, 8) X_(8, 4) X_(8, 16) X_(16, 8) X_(16, 32) X_(32, 16) X_(32, 64) X_(64, 32) X_(64, 128)   \
  X_(128, 64) X_(128, 256) X_(256, 128)

extern "C" int gatres_t_proj_attn_fwd(const void* x, const void* W, const float* att_src, const float* att_dst, void* h, float* a_src,
                           float* a_dst, int num_nodes, int K, int H, int C, int dtype, void* stream) {
  if (!x || !W || !att_src || !att_dst || !h || !a_src || !a_dst || num_nodes <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(x) || !gatres_aligned16(W) || !gatres_aligned16(h) || !gatres_aligned16(att_src) ||
      !gatres_aligned16(att_dst))
    return GATRES_E_BADARG;
  const int M = H * C;
  hipStream_t st = gatres_stream(stream);
  if (dtype == GATRES_DTYPE_BF16) {
    const gatres_bf16* xb = (const gatres_bf16*)x; const gatres_bf16* Wb = (const gatres_bf16*)W; gatres_bf16* hb = (gatres_bf16*)h;
#define CASE_(K_, M_)                                                                                               \
  if (K == K_ && M == M_) {                                                                                         \
    if (H == 1) return launch_proj_bf16<K_, M_, 1, EPI_ATT>(xb, Wb, hb, num_nodes, att_src, att_dst, a_src, a_dst, \
                                                              nullptr, nullptr, st);                                \
    if (H == 2) return launch_proj_bf16<K_, M_, 2, EPI_ATT>(xb, Wb, hb, num_nodes, att_src, att_dst, a_src, a_dst, \
                                                              nullptr, nullptr, st);                                \
  }
    GATRES_FOR_SHAPES(CASE_)
#undef CASE_
    return GATRES_E_UNSUPPORTED;
  }
  if (dtype != GATRES_DTYPE_F32) return GATRES_E_UNSUPPORTED;
  const float* xf = (const float*)x; const float* Wf = (const float*)W; float* hf = (float*)h;
#define CASE_(K_, M_)                                                                                          \
  if (K == K_ && M == M_) {                                                                                    \
    if (H == 1) return launch_proj<K_, M_, 1, EPI_ATT>(xf, Wf, hf, num_nodes, att_src, att_dst, a_src, a_dst, \
                                                         nullptr, nullptr, st);                                \
    if (H == 2) return launch_proj<K_, M_, 2, EPI_ATT>(xf, Wf, hf, num_nodes, att_src, att_dst, a_src, a_dst, \
                                                         nullptr, nullptr, st);                                \
  }
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_t_proj_bwd_dx(const void* g_h, const void* Wt, const void* resid, const void* relu_ref, void* g_x, int num_nodes,
                         int K, int HC, int dtype, void* stream) {
  if (!g_h || !Wt || !g_x || num_nodes <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(g_h) || !gatres_aligned16(Wt) || !gatres_aligned16(g_x) || !gatres_aligned16(resid) ||
      !gatres_aligned16(relu_ref))
    return GATRES_E_BADARG;
  hipStream_t st = gatres_stream(stream);
  // reduction length = HC, outputs = K
  if (dtype == GATRES_DTYPE_BF16) {
#define CASE_(K_, M_)                                                                                                 \
  if (HC == K_ && K == M_)                                                                                            \
    return launch_proj_bf16<K_, M_, 1, EPI_RESID_MASK>((const gatres_bf16*)g_h, (const gatres_bf16*)Wt, (gatres_bf16*)g_x, \
                                                       num_nodes, nullptr, nullptr, nullptr, nullptr,                 \
                                                       (const gatres_bf16*)resid, (const gatres_bf16*)relu_ref, st);
    GATRES_FOR_SHAPES(CASE_)
#undef CASE_
    return GATRES_E_UNSUPPORTED;
  }
  if (dtype != GATRES_DTYPE_F32) return GATRES_E_UNSUPPORTED;
#define CASE_(K_, M_)                                                                                          \
  if (HC == K_ && K == M_)                                                                                     \
    return launch_proj<K_, M_, 1, EPI_RESID_MASK>((const float*)g_h, (const float*)Wt, (float*)g_x, num_nodes, \
                                                  nullptr, nullptr, nullptr, nullptr, (const float*)resid,     \
                                                  (const float*)relu_ref, st);
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_t_proj_bwd_dw(const void* g_h, const void* x, float* slab_W, int num_slabs, int64_t slab_stride, int num_nodes,
                         int K, int HC, int dtype, void* stream) {
  if (!g_h || !x || !slab_W || num_nodes <= 0 || num_slabs <= 0) return GATRES_E_BADARG;
  if (!gatres_aligned16(g_h) || !gatres_aligned16(x)) return GATRES_E_BADARG;
  hipStream_t st = gatres_stream(stream);
  if (dtype == GATRES_DTYPE_BF16) {
#define CASE_(K_, M_)                                                                                                  \
  if (K == K_ && HC == M_) {                                                                                           \
    if (K_ >= 32 && M_ >= 32 && !gatres_knobs()->dw_fp32)                                                             \
      return launch_dw_bf16<M_, K_>((const gatres_bf16*)g_h, (const gatres_bf16*)x, slab_W, num_slabs, slab_stride,   \
                                    num_nodes, st);                                                                    \
    return launch_dw<M_, K_, gatres_bf16>((const gatres_bf16*)g_h, (const gatres_bf16*)x, slab_W, num_slabs,          \
                                          slab_stride, num_nodes, st);                                                 \
  }
    GATRES_FOR_SHAPES(CASE_)
#undef CASE_
    return GATRES_E_UNSUPPORTED;
  }
  if (dtype != GATRES_DTYPE_F32) return GATRES_E_UNSUPPORTED;
#define CASE_(K_, M_) \
  if (K == K_ && HC == M_) return launch_dw<M_, K_, float>((const float*)g_h, (const float*)x, slab_W, num_slabs, slab_stride, num_nodes, st);
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

// (not part of include/gatres.h: the per-op backward driver's co-launch of a convolution's two partial-sum kernels -- bf16
//  storage, two-dimensional weight partials.  GATRES_E_UNSUPPORTED = launch gatres_t_proj_bwd_dw and
//  gatres_t_conv_param_grads separately.)
extern "C" __attribute__((visibility("hidden"))) int gatres_proj_bwd_dw_with_conv_grads(
    const void* g_h, const void* x, float* slab_W, int w_slabs, int64_t slab_stride, int num_nodes, int K, int HC,
    const void* h, const float* g_a_src, const float* g_a_dst, const void* g_out, float* slab_att_src, float* slab_att_dst,
    float* slab_bias, int num_slabs, int H, int C, int dtype, void* stream) {
  if (!g_h || !x || !slab_W || !h || !g_a_src || !g_a_dst || !g_out || !slab_att_src || !slab_att_dst || !slab_bias ||
      num_nodes <= 0 || w_slabs <= 0 || num_slabs <= 0)
    return GATRES_E_BADARG;
  if (!gatres_aligned16(g_h) || !gatres_aligned16(x)) return GATRES_E_BADARG;
  if (H * C != HC || HC > 256 || gatres_knobs()->no_co_launch) return GATRES_E_UNSUPPORTED;
  int nps = (num_nodes + num_slabs - 1) / num_slabs;
  nps = (nps + 3) & ~3;                          // (nodes_per_slab of k_misc.hip: the slab boundaries of every other kernel)
  hipStream_t st = gatres_stream(stream);
  if (dtype == GATRES_DTYPE_BF16) {
    if ((HC % 2) || (C % 2) || gatres_knobs()->dw_fp32) return GATRES_E_UNSUPPORTED;
    ConvGradsCo cg{(const gatres_bf16*)h, g_a_src, g_a_dst, (const gatres_bf16*)g_out, slab_att_src, slab_att_dst, slab_bias,
                   (long long)slab_stride, H, C, nps};
#define CASE_(K_, M_)                                                                                                 \
  if (K == K_ && HC == M_) {                                                                                          \
    if (K_ >= 32 && M_ >= 32)                                                                                         \
      return launch_dw_bf16<M_, K_>((const gatres_bf16*)g_h, (const gatres_bf16*)x, slab_W, w_slabs, slab_stride,     \
                                    num_nodes, st, &cg, num_slabs);                                                   \
    return GATRES_E_UNSUPPORTED;                                                                                      \
  }
    GATRES_FOR_SHAPES(CASE_)
#undef CASE_
    return GATRES_E_UNSUPPORTED;
  }
  if (dtype != GATRES_DTYPE_F32) return GATRES_E_UNSUPPORTED;
  // fp32: the per-wave weight-gradient kernel fills the chip by itself on large tables -- measured: gatres_small per-op
  // (12 k rows) 1.21 -> 1.04 ms/step with the co-launch, gatres_large on 50 k / 100 k rows 13.53 -> 13.54 / 25.0 -> 25.4
  if (num_nodes > 32768 && !gatres_knobs()->co_launch_always) return GATRES_E_UNSUPPORTED;
  gatres_conv_grads_co<float> cf{(const float*)h, g_a_src, g_a_dst, (const float*)g_out, slab_att_src, slab_att_dst,
                                 slab_bias, (long long)slab_stride, H, C, nps};
#define CASE_(K_, M_) \
  if (K == K_ && HC == M_) return launch_dw<M_, K_, float>((const float*)g_h, (const float*)x, slab_W, w_slabs, slab_stride, num_nodes, st, &cf, num_slabs);
  GATRES_FOR_SHAPES(CASE_)
#undef CASE_
  return GATRES_E_UNSUPPORTED;
}

extern "C" int gatres_proj_attn_fwd(const float* x, const float* W, const float* att_src, const float* att_dst,
                                    float* h, float* a_src, float* a_dst, int32_t num_nodes, int32_t K, int32_t H,
                                    int32_t C, void* stream) {
  return gatres_t_proj_attn_fwd(x, W, att_src, att_dst, h, a_src, a_dst, num_nodes, K, H, C, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_proj_bwd_dx(const float* g_h, const float* Wt, const float* resid, const float* relu_ref,
                                  float* g_x, int32_t num_nodes, int32_t K, int32_t HC, void* stream) {
  return gatres_t_proj_bwd_dx(g_h, Wt, resid, relu_ref, g_x, num_nodes, K, HC, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_proj_bwd_dw(const float* g_h, const float* x, float* slab_W, int32_t num_slabs,
                                  int64_t slab_stride, int32_t num_nodes, int32_t K, int32_t HC, void* stream) {
  return gatres_t_proj_bwd_dw(g_h, x, slab_W, num_slabs, slab_stride, num_nodes, K, HC, GATRES_DTYPE_F32, stream);
}

#ifdef PROJ_STAMPS
extern "C" int gatres_probe_pstamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 512 * 8 * 8);
}
#endif
