// Device code shared by the per-snapshot GATRes kernels (k_fused.hip: whole-segment tables, k_window.hip: row windows,
// k_fused_host.hip: the deferred parameter gradients and the C-ABI).  Everything here has internal linkage.
// Fused per-snapshot GATRes kernels for gfx950: ONE workgroup owns ONE graph segment (a snapshot of the batch)
// and carries it through lin0 -> num_blocks x [K1 conv1, K2 conv1, K1 conv2, K2 conv2, K3] -> lin1 -> masked-MSE ->
// the whole backward, inside a single launch.
//
// Why (measured on MI355X, profiles/r01_perop_kernel_stats.csv): as ~230 separate launches every stage is a
// dependent kernel whose inputs were just written by other XCDs, so each costs ~5 us of launch + Infinity-Cache
// latency however little it moves; the step was 2.35 ms for 1.6 GB of algorithmic traffic.  A PyG batch is
// block-diagonal (train.py:302-303; `batch` is not even passed to the model, train.py:167), so a snapshot never
// needs another workgroup's data: the only synchronisation left is __syncthreads().
//
// A single workgroup cannot hide latency with parallelism, so the kernel is organised around dependent-load depth
// (profiles/r01_fused_v0_stage_times.txt: a naive port of the per-op stages ran 3.4 ms because every col[e] ->
// gather hop was a serial ~0.3 us trip):
//   * the segment's topology lives in LDS as 16-bit LOCAL indices (rowptr / col / transposed / mean CSR);
//   * per destination row all (<= 8) neighbour loads are issued together, then reduced in CSR order, so the sums
//     are bit-identical to the per-op kernels' edge-at-a-time loops;
//   * forward: the tables gathered by neighbour index (h of conv1 / y2, h of conv2, attention logits) stay in LDS
//     (C-Town at nc=32: 388 x (64+32+4) x 4 B = 155 KB of the CU's 160 KB); backward: g_pre | g_y2, g_out1, g_e,
//     g_a_src, g_a_dst stay in LDS;  only what the backward pass needs later is written to HBM (coalesced float4);
//   * dense projections on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32), two 16-node tiles per wave
//     sharing each W fragment; dW partials with 8 operand pairs in flight per wave.
#pragma once
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "gatres_common.h"
#include "gatres_layout.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int LDS_BYTES = 163840;   // the whole CU LDS; one workgroup per CU
constexpr int MAXD = 6;             // neighbour loads issued together per row (rows with more edges take a loop);
                                    // C-Town's largest in-degree is 5 (+1 self loop)

struct FusedArgs {
  const int* seg_ptr;
  const int *rowptr, *col, *t_rowptr, *t_eid, *t_dst, *m_rowptr, *m_col, *mt_rowptr, *mt_dst;
  int N;
  const float* params;
  const float* wt;          // transposed conv weights (backward)
  const int* perm;          // plan's node relabelling (new id -> caller's id) for x / y / mask / out / g_out / g_x; null = identity
  const float* x;
  const uint8_t* mask;      // may be null (no masking of x; the loss phase needs it)
  const float* y;
  float* out;
  float* g_out;             // written by the loss phase, read by the backward phase
  float* loss_part;         // [num_segments + 1]: per-segment sum of squared errors; last = masked-node count
  float* g_x;               // may be null
  float* saved;             // may be null for inference
  float* scratch;
  float* slabs;
  float* part_slabs;        // split segments: one slab row per (segment, part), folded by param_grads_kernel
  unsigned* flags;          // split segments: 8 flag lines per segment
  int* err;
  int num_segments;
  int seg0, seg_cnt;        // the segments THIS launch carries: [seg0, seg0 + seg_cnt) -- batches beyond 32 snapshots at 8
                            // parts go round by round (k_fused_host.hip: launch_fused), everything else is one launch
  int M;                    // workgroups (CUs) per segment
  int safe_sync;            // diagnostic (GATRES_FUSED_SAFE_SYNC=1): always use agent-scope barriers
  int keep_lds;             // window kernel, forward + backward in one launch: ReLU sign masks and own-row g_pre stay in LDS
  int no_halo;              // diagnostic (GATRES_FUSED_NO_HALO=1): always take the bulk-pull fallback
  int C;                    // consumer workgroups per segment (deferred parameter gradients on spare CUs), 0 = none
  int sym;                  // the plan's GATRES_GRAPH_SYMMETRIC: partners owe each other halo rows in both directions
  int facts;                // window kernel, host side only: what the launch may take as compile-time facts (k_window.hip: 0x400
                            // no row with more than MAXD entries, 0x1000 no part with more than 64 rows, 0x2000 a forward-only launch whose saved
                            // activations nobody reads) and, read by the kernel itself, 0x4000: parts 8 ids apart share an XCD
                            // (gatres_probe_xcd_dispatch: the launch starts without its first cross-CU barrier)
  const int* ptab;          // the plan's part tables (gatres_graph_t.part_tables) and what they were built for; may be null
  int ptab_m, ptab_stride;
  unsigned* ready;          // [segment][4] lines: items published by the segment's part 0 for each consumer
  unsigned long long* xch;  // window kernel: granule exchange regions, one per segment (Layout::sc_xch)
  int* urec;                // window kernel: GATRES_UREC_WORDS words per workgroup (Layout::sc_urec), written by its prologue
  XchLayout XL;
  Layout L;
  SegLayout SL;             // segment-major saved activations (training)
  int phases;               // GATRES_PHASE_FORWARD | _BACKWARD, bit 16: loss
  unsigned long long* stamps;   // diagnostic: segment 0 stamps the wall clock at every stage boundary
  int stamp_cap;
};

enum { PH_LOSS = 16 };

// Stage stamps and the knobs that produce WRONG results (never-wait exchanges, unmasked dX epilogues) exist only in the
// diagnostic build of the library (-DGATRES_DIAG_BUILD: _build.build_native(diag=True) -> lib/libgatres_hip_diag.so,
// loaded when GATRES_DIAG_LIB=1); the product build carries neither the code nor the registers they cost.
#if GATRES_DIAG
#define STAMPS_PTR (a.stamps)
#define STAMP()                                                                                  \
  do {                                                                                           \
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && stamp_i < a.stamp_cap)                \
      a.stamps[stamp_i++] = wall_clock64();                                                      \
  } while (0)
#else
#define STAMPS_PTR (static_cast<unsigned long long*>(nullptr))
#define STAMP() do {} while (0)
#endif

// finer diagnostic of ONE backward block of segment 0 (tests/stage_profile.py --xstamps): wave 0 and the last wave of every
// part stamp the steps of the block's three exchanges; slots [2048 + (part * 2 + who) * 64 + step]
#if GATRES_DIAG
#define XSTAMP()                                                                                 \
  do {                                                                                           \
    if (xs_on && (threadIdx.x == 0 || threadIdx.x == THREADS - 64) && xs_i < 64)                 \
      a.stamps[2048 + (part * 2 + (threadIdx.x ? 1 : 0)) * 64 + xs_i] = wall_clock64();         \
    ++xs_i;                                                                                      \
  } while (0)
#else
#define XSTAMP() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------ LDS budget (host+device)
__host__ __device__ inline int even(int v) { return (v + 1) & ~1; }
__host__ __device__ inline int topo_fwd_bytes(int n, int eg, int em) {
  return 2 * (2 * even(n + 1) + even(eg) + even(em));
}
__host__ __device__ inline int topo_bwd_bytes(int n, int eg, int em) {
  return 2 * (4 * even(n + 1) + 3 * even(eg) + 2 * even(em));
}
static long long wl_bytes(int nc) {
  const long long f = 4LL * (2LL * nc * (2 * nc + 4) + 4 * nc);
  return f <= 40960 ? f + 16 : 0;          // W staging slot of seg_proj (not used for nc = 128: W does not fit)
}
static bool cache_fits(int nc, int threads, int n, int eg, int em) {
  const long long fwd = 4LL * n * (3 * nc + 4) + topo_fwd_bytes(n, eg, em);
  const long long bwd = 12LL * threads + 4LL * n * 2 * nc + 8LL * even(eg) + 16LL * n + topo_bwd_bytes(n, eg, em) +
                        wl_bytes(nc);
  return fwd <= LDS_BYTES && bwd <= LDS_BYTES && n <= 65535 && eg <= 65535 && em <= 65535;
}
static bool nocache_fits(int nc, int threads, int n, int eg, int em) {
  return 12LL * threads + topo_bwd_bytes(n, eg, em) + wl_bytes(nc) <= LDS_BYTES && n <= 65535 && eg <= 65535 &&
         em <= 65535;
}

// Rows [lo, hi) of the segment that THIS workgroup computes.  One workgroup per segment owns all rows; when a segment
// is split over several CUs (see group_sync) each part owns a 16-aligned window and the tables it gathers from are
// completed from the partners' global copies after a flag barrier.
struct Rows { int lo, hi; };
// workgroup-uniform values that reach the kernel through vector loads (LDS words, global CSR entries the compiler does not
// scalarise): pinned to SGPRs, so everything derived from them -- table bases, counts, loop bounds -- is scalar as well
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// workgroup barrier that waits for the wave's LDS operations only (not for its global stores / LDS-DMA in flight)
__device__ __forceinline__ void lds_barrier_raw() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// caller-side index of plan node `node` (gatres_graph_t.perm): only x / y / mask / out / g_out / g_x are indexed with it
__device__ __forceinline__ int ext_id(const int* __restrict__ perm, int node) { return perm ? perm[node] : node; }

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

// streaming read (nt): the deferred-gradient items read every kept / saved table exactly once; keep them from
// evicting the L2 lines the per-snapshot workgroups on the same XCD are working with
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
template <int KQ>
__device__ __forceinline__ void load_frag_nt(const float* p, float (&f)[KQ]) {
  if constexpr (KQ % 4 == 0) {
#pragma unroll
    for (int s = 0; s < KQ; s += 4) {
      const float4 v = ld4_nt(p + s);
      f[s] = v.x; f[s + 1] = v.y; f[s + 2] = v.z; f[s + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < KQ; ++s) f[s] = __builtin_nontemporal_load(p + s);
  }
}

template <int KQ>
__device__ __forceinline__ void load_frag(const float* p, float (&f)[KQ]) {
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

__device__ __forceinline__ void add4(float4& acc, const float4 v) {
  acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
}

template <int THREADS>
__device__ __forceinline__ void copy_rowptr16(u16* dst, const int* __restrict__ src, int n0, int n, int e0) {
  for (int r = threadIdx.x; r <= n; r += THREADS) dst[r] = (u16)(src[n0 + r] - e0);
}
template <int THREADS>
__device__ __forceinline__ void copy_idx16(u16* dst, const int* __restrict__ src, int e0, int cnt, int sub) {
  for (int k = threadIdx.x; k < cnt; k += THREADS) dst[k] = (u16)(src[e0 + k] - sub);
}

enum { EPI_NONE = 0, EPI_ATT = 1, EPI_RESID_MASK = 2 };

// ReLU sign mask of one row: G lanes hold the row's float4 column groups; the word packs four fields FS bits apart, field c
// bit l = (component c of lane l's float4 is positive).  The forward stages of the window kernel leave these in LDS so
// the dX epilogues of the backward phase mask without reading the saved activations back from HBM.  (Every lane of the
// wave calls this.)
template <int G, int FS>
__device__ __forceinline__ unsigned long long relu_bits(const float4 o) {
  static_assert(G <= FS && 4 * FS <= 64 && 64 % G == 0, "field layout");
  const int sh = ((threadIdx.x & 63) / G) * G;
  const unsigned long long fm = (1ull << G) - 1;
  unsigned long long w = (__ballot(o.x > 0.f) >> sh) & fm;
  w |= ((__ballot(o.y > 0.f) >> sh) & fm) << FS;
  w |= ((__ballot(o.z > 0.f) >> sh) & fm) << (2 * FS);
  w |= ((__ballot(o.w > 0.f) >> sh) & fm) << (3 * FS);
  return w;
}

// ------------------------------------------------------------------------------------------ K1 (MFMA)
// OUT[ob + r, :] = X[xb + r, :] @ Wm^T for r in [0, n).  Same lane map / k order as proj_kernel (k_proj.hip), so
// results are bit-identical; each wave takes TWO 16-node tiles per trip and feeds both from one W fragment load.
template <int K, int M, int H, int EPI, int THREADS, bool WLDS, bool WPRE = false>
__device__ __forceinline__ void seg_proj(Rows rw, const float* X, int xb, const float* __restrict__ Wm, float* OUT,
                                         int ob, float* OUT2, int o2b, const float* __restrict__ att_src,
                                         const float* __restrict__ att_dst, float* as_g, float* ad_g, int ag_b,
                                         float* as_l, float* ad_l, const float* resid, int rb,
                                         const float* relu_ref, int mb_, float* wl,
                                         const float* resid_l = nullptr, float* OUT3 = nullptr,
                                         const unsigned long long* m64 = nullptr, const unsigned* m32 = nullptr) {
  // (resid_l / OUT3: LDS [row][M] copies used instead of resid / in addition to OUT; m64 / m32: relu_bits words per row,
  //  fields 16 / 8 bits apart, used instead of relu_ref)
  constexpr int KQ = K / 4, NT = (M + 15) / 16, NW = THREADS / 64;
  constexpr int KP = K + 4;                  // padded LDS row: 16 lanes x float4 at stride KP hit distinct banks
  constexpr int SC = (KQ % 4 == 0) ? 4 : ((KQ % 2 == 0) ? 2 : 1);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  if (rw.hi <= rw.lo) return;                                  // (workgroup-uniform) an empty part of a split segment
  const int tlo = rw.lo >> 4, ntiles = (rw.hi + 15) >> 4;      // rw.lo is 16-aligned

  // the first tile's x fragment: in flight while W is staged
  float xcur[KQ], xnxt[KQ];
  {
    const int r0 = (tlo + wave) * 16 + i;
    load_frag<KQ>(X + (unsigned)((xb + min(r0, rw.hi - 1)) * K + q * KQ), xcur);
  }
  // Stage W (rows padded to KP) and the attention vectors into LDS once per stage: every wave then feeds its MFMA
  // chain from 28-ns LDS reads instead of ~0.3-us L2 trips that the in-order wave exposes one after another.
  // (Prefetching the W share into registers before the preceding barrier was measured: no gain, more spills.)
  const float* attS = att_src;
  const float* attD = att_dst;
  if constexpr (WLDS && WPRE) {          // W (+ att) was put into the slot by w_prefetch during an earlier stage
    if constexpr (EPI == EPI_ATT) { attS = wl + M * KP; attD = wl + M * KP + M; }
  } else if constexpr (WLDS) {
    for (int idx = threadIdx.x; idx < M * (K / 4); idx += THREADS) {
      const int m = idx / (K / 4), k4 = (idx % (K / 4)) * 4;
      st4(wl + m * KP + k4, ld4(Wm + (unsigned)(m * K + k4)));
    }
    if constexpr (EPI == EPI_ATT) {
      float* al = wl + M * KP;
      for (int idx = threadIdx.x; idx < 2 * (M / 4); idx += THREADS) {
        const int which = idx / (M / 4), c4 = (idx % (M / 4)) * 4;
        st4(al + which * M + c4, ld4((which ? att_dst : att_src) + c4));
      }
      attS = al; attD = al + M;
    }
    __syncthreads();
  }

  auto epilogue = [&](f32x4(&acc)[NT], int r, bool rok) {
    if constexpr (EPI == EPI_ATT) {
      constexpr int C = M / H;
      float ps[H], pd[H];
#pragma unroll
      for (int hh = 0; hh < H; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int mb = tt * 16 + q * 4;
        if ((M % 16 == 0) || (mb < M)) {
          const float4 as = ld4(attS + mb), ad = ld4(attD + mb);
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
          as_g[(unsigned)((ag_b + r) * H + hh)] = ps[hh];
          ad_g[(unsigned)((ag_b + r) * H + hh)] = pd[hh];
          if (as_l) { as_l[r * H + hh] = ps[hh]; ad_l[r * H + hh] = pd[hh]; }
        }
      }
    }
    if (rok) {
      unsigned long long mw = 0;
      int fs = 16;
      if constexpr (EPI == EPI_RESID_MASK) {
        if (m64) mw = m64[r];
        else if (m32) { mw = m32[r]; fs = 8; }
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int mb = tt * 16 + q * 4;
        if ((M % 16 == 0) || (mb < M)) {
          float4 o = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
          if constexpr (EPI == EPI_RESID_MASK) {
            if (resid_l) add4(o, ld4(resid_l + (unsigned)(r * M + mb)));
            else if (resid) add4(o, ld4(resid + (unsigned)((rb + r) * M + mb)));
            if (m64 || m32) {
              const unsigned long long b = mw >> (mb >> 2);
              o.x = (b & 1) ? o.x : 0.f;                 o.y = ((b >> fs) & 1) ? o.y : 0.f;
              o.z = ((b >> (2 * fs)) & 1) ? o.z : 0.f;   o.w = ((b >> (3 * fs)) & 1) ? o.w : 0.f;
            } else if (relu_ref) {
              const float4 rr = ld4(relu_ref + (unsigned)((mb_ + r) * M + mb));
              o.x = rr.x > 0.f ? o.x : 0.f; o.y = rr.y > 0.f ? o.y : 0.f;
              o.z = rr.z > 0.f ? o.z : 0.f; o.w = rr.w > 0.f ? o.w : 0.f;
            }
          }
          st4(OUT + (unsigned)((ob + r) * M + mb), o);
          if (OUT2) st4(OUT2 + (unsigned)((o2b + r) * M + mb), o);
          if (OUT3) st4(OUT3 + (unsigned)(r * M + mb), o);
        }
      }
    }
  };

  // W fragments: loaded ONCE per stage into registers when they fit (one load latency instead of NT*KQ/SC
  // dependent ones); lane (i, q) needs Wm[16*tt + i][q*KQ .. q*KQ+KQ) for every tile row tt.
  constexpr bool HOIST = false;   // holding W in 32+ registers spills at the 128-VGPR budget; W re-reads hit L1
  float wreg[HOIST ? NT : 1][HOIST ? KQ : 1];
  if constexpr (HOIST) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int m = tt * 16 + i;
      const bool mok = (M % 16 == 0) || (m < M);
      load_frag<KQ>(Wm + (size_t)(mok ? m : 0) * K + q * KQ, wreg[tt]);
      if (!mok) {
#pragma unroll
        for (int s = 0; s < KQ; ++s) wreg[tt][s] = 0.f;
      }
    }
  }
  // one 16-node tile per trip; the NEXT trip's x fragment is loaded before this trip's MFMA chain so its latency
  // hides behind the matrix work (a two-tiles-per-trip variant spilled at the 128-VGPR budget of a 16-wave workgroup)
  for (int t0 = tlo + wave; t0 < ntiles; t0 += NW) {
    const int rA = t0 * 16 + i;
    const bool okA = rA < rw.hi;
    const int rN = (t0 + NW) * 16 + i;
    if (t0 + NW < ntiles) load_frag<KQ>(X + (unsigned)((xb + min(rN, rw.hi - 1)) * K + q * KQ), xnxt);   // uniform branch
    f32x4 acc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KQ; s += SC) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float wf[SC];
        if constexpr (HOIST) {
#pragma unroll
          for (int u = 0; u < SC; ++u) wf[u] = wreg[tt][s + u];
        } else {
          const int m = tt * 16 + i;
          const bool mok = (M % 16 == 0) || (m < M);
          if constexpr (WLDS) load_frag<SC>(wl + (mok ? m : 0) * KP + q * KQ + s, wf);
          else                load_frag<SC>(Wm + (unsigned)((mok ? m : 0) * K + q * KQ + s), wf);
          if (!mok) {
#pragma unroll
            for (int u = 0; u < SC; ++u) wf[u] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < SC; ++u)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], xcur[s + u], acc[tt], 0, 0, 0);
      }
    }
    epilogue(acc, rA, okA);
#pragma unroll
    for (int s = 0; s < KQ; ++s) xcur[s] = xnxt[s];
  }
}

// dW partial of this segment: slab[c*K + k] = sum_r G[gb + r, c] * X[xb + r, k].  One 16x16 output tile per wave
// (x 2 node halves when there are waves to spare), 8 operand pairs in flight, halves combined through LDS.
template <int HC, int K, int THREADS>
__device__ __forceinline__ void seg_dw(int n, const float* G, int gb, const float* X, int xb,
                                       float* __restrict__ slab, float* red) {
  constexpr int NCT = (HC + 15) / 16, NKT = (K + 15) / 16, NTILE = NCT * NKT, NW = THREADS / 64;
  constexpr int HALVES = (NW >= 2 * NTILE) ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int nhalf = HALVES == 2 ? (((n + 1) / 2 + 3) & ~3) : n;
  for (int u0 = 0; u0 < NTILE * HALVES; u0 += NW) {
    const int u = u0 + wave;
    const bool active = u < NTILE * HALVES;              // wave-uniform
    const int tile = active ? u % NTILE : 0, half = active ? u / NTILE : 0;
    const int c = (tile / NKT) * 16 + i, k = (tile % NKT) * 16 + i;
    const bool cok = c < HC, kok = k < K;
    const int rbeg = half * nhalf, rend = min(n, rbeg + nhalf);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (active) {
      for (int nb = rbeg; nb < rend; nb += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = nb + 4 * j + q;
          const bool rok = r < rend;
          av[j] = (rok && cok) ? G[(size_t)(gb + (rok ? r : 0)) * HC + c] : 0.f;
          bv[j] = (rok && kok) ? X[(size_t)(xb + (rok ? r : 0)) * K + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
      }
    }
    if constexpr (HALVES == 2) {
      __syncthreads();
      if (active && half == 1) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[(tile * 64 + lane) * 4 + rr] = acc[rr];
      }
      __syncthreads();
      if (active && half == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[rr] += red[(tile * 64 + lane) * 4 + rr];
      }
    }
    if (active && half == 0) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int cr = (tile / NKT) * 16 + 4 * q + rr;
        if (cr < HC && kok) slab[(size_t)cr * K + k] = acc[rr];
      }
    }
  }
}

// dW partial, block form (HC, K in {16, 32, 64}): ONE wave computes the whole [HC, K] gradient block for a range
// of rows, with the permuted output map of k_proj.hip's dw_kernel (c = VC*i + tc, k = VK*j + tk) so both MFMA
// operands come from per-lane VECTOR loads (float4 / float2) of contiguous features instead of 4-byte column
// slices; R row ranges run on R waves, their partial blocks meet in LDS (`part`, R*HC*K floats: the backward's idle
// g_pre|g_y2 / g_out1 region) and are summed in range order -> the segment's slab.  Deterministic.
template <int HC, int K, int THREADS, bool NT = false>
__device__ __forceinline__ void seg_dw_blk(int n, int R, const float* G, int gb, const float* X, int xb,
                                           float* __restrict__ slab, float* part) {
  constexpr int VC = HC / 16, VK = K / 16, STEPS = 5;              // 5 steps (20 rows) of loads in flight
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  if (wave < R) {                                                   // wave-uniform
    const int chunk = (((n + R - 1) / R) + 3) & ~3;
    const int rbeg = wave * chunk, rend = min(n, rbeg + chunk);
    f32x4 acc[VC][VK];
#pragma unroll
    for (int a = 0; a < VC; ++a)
#pragma unroll
      for (int b = 0; b < VK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int nb = rbeg; nb < rend; nb += 4 * STEPS) {
      float av[STEPS][VC], bv[STEPS][VK];
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int r = nb + 4 * st + q;
        const bool ok = r < rend;
        const int rr = ok ? r : rbeg;
        if constexpr (NT) {           // streaming reads when running beside the per-snapshot workgroups
          load_frag_nt<VC>(G + (unsigned)((gb + rr) * HC + VC * i), av[st]);
          load_frag_nt<VK>(X + (unsigned)((xb + rr) * K + VK * i), bv[st]);
        } else {
          load_frag<VC>(G + (unsigned)((gb + rr) * HC + VC * i), av[st]);
          load_frag<VK>(X + (unsigned)((xb + rr) * K + VK * i), bv[st]);
        }
        if (!ok) {
#pragma unroll
          for (int a = 0; a < VC; ++a) av[st][a] = 0.f;
        }
      }
#pragma unroll
      for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int a = 0; a < VC; ++a)
#pragma unroll
          for (int b = 0; b < VK; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][a], bv[st][b], acc[a][b], 0, 0, 0);
    }
    float* mine = part + wave * (HC * K);
#pragma unroll
    for (int a = 0; a < VC; ++a)
#pragma unroll
      for (int b = 0; b < VK; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mine[(VC * (4 * q + rr) + a) * K + VK * i + b] = acc[a][b][rr];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < HC * K; idx += THREADS) {
    float sum = 0.f;
    for (int w = 0; w < R; ++w) sum += part[w * (HC * K) + idx];
    slab[idx] = sum;
  }
}

// ------------------------------------------------------------------------------------------ sparse stages
// Conventions.  Row r of the segment; tables are addressed as  base + (tb + local_index) * width  with tb = 0 for an
// LDS copy and tb = n0 for the global array; rp/col/...: LDS, 16-bit local indices.
// Rows with <= MAXD edges (every WDN row) take the slot path: all neighbour loads are issued together and the
// reductions run in CSR order, so results are bit-identical to the per-op kernels' edge-at-a-time loops.  Slots
// beyond a row's degree are made HARMLESS instead of predicated (weight 0, a valid address), and slots beyond the
// wave's maximum degree are skipped with wave-uniform branches: no per-lane exec-mask juggling in the hot loop.

__device__ __forceinline__ int wave_max_deg(int deg) {
  int d = 0;
#pragma unroll
  for (int k = 0; k < MAXD; ++k)
    if (__ballot(deg > k)) d = k + 1;            // scalar condition
  return d;
}

template <int U>
__device__ __forceinline__ int max_of(const int (&d)[U]) {
  int m = d[0];
#pragma unroll
  for (int u = 1; u < U; ++u) m = max(m, d[u]);
  return m;
}

constexpr int SPIN_LIMIT = 1 << 20;             // a lost partner poisons the results instead of hanging the GPU
// ---- granule primitives of the window kernel's exchange (described at xch_export below)
typedef unsigned long long u64;
// local: every part of the segment runs on ONE XCD (verified from HW_REG_XCC_ID at the launch's first barrier).  Then a
// plain 8-byte store is enough: it writes through the CU's L1 into the XCD's L2 and KEEPS the line there, where the
// partner's sc1 load (L1 bypassed) finds it at L2 latency.  Anywhere else the store is an agent-scope one (sc1: written
// through to memory, line dropped from this L2), which any XCD's sc1 load observes -- slower, never wrong.
__device__ __forceinline__ void gran_store(u64* p, float v, unsigned tag, bool local) {
  const u64 g = ((u64)tag << 32) | (u64)__float_as_uint(v);
  if (local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  else       __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Xch {
  u64* base;            // this segment's region
  u64* base_hb;         // its heartbeat granules, one per part (the first row of XchLayout.hb)
  unsigned ep;          // epoch of the current exchange
  int M, part;
  int* err;
  bool dead;
  bool local;           // all parts on one XCD (Group::local)
};

// K2 forward, sub-stage A: attention coefficients.  ONE thread per (row, head): no cross-lane traffic, one exp and
// one divide per edge.  (A single-stage K2 in which every lane of a head recomputes the coefficients was measured
// twice -- one CU per snapshot and ~100 rows per part of a split snapshot: the redundant exp / divide work costs more
// than the barrier and the LDS hop it removes, 2.8 -> 4.7 us for conv1 at 4 CUs per snapshot.)  alpha goes to HBM (saved for the backward pass) and, when ALDS, to an LDS table for sub-stage B.
template <int H, bool ALDS, int THREADS>
__device__ __forceinline__ void seg_softmax(Rows rw, const u16* rp, const u16* col, const float* asrc,
                                            const float* adst_t, int ab, float* __restrict__ alpha_g, int eb,
                                            float* alpha_l) {
  for (int idx = rw.lo * H + threadIdx.x; idx < rw.hi * H; idx += THREADS) {
    const int r = idx / H, hd = idx % H;
    const int beg = rp[r], deg = (int)rp[r + 1] - beg;            // deg >= 1 (self loop)
    const float adst = adst_t[(unsigned)((ab + r) * H + hd)];
    if (__builtin_expect(deg <= MAXD, 1)) {
      float so[MAXD];
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        const int jj = col[beg + min(k, deg - 1)];
        const float sv = gatres_leaky(asrc[(unsigned)((ab + jj) * H + hd)] + adst);
        so[k] = k < deg ? sv : -INFINITY;
        m = fmaxf(m, so[k]);
      }
      float Z = 0.f;
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        so[k] = expf(so[k] - m);                                   // exp(-inf) = 0 on padding slots
        Z = Z + so[k];
      }
      Z = Z + GATRES_SOFTMAX_EPS;
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < deg) {
          const float al = so[k] / Z;
          alpha_g[(unsigned)((eb + beg + k) * H + hd)] = al;
          if constexpr (ALDS) alpha_l[(beg + k) * H + hd] = al;
        }
    } else {
      const int end = beg + deg;
      auto at = [&](int e) -> float { return asrc[(unsigned)((ab + col[e]) * H + hd)]; };
      float m = -INFINITY;
      for (int e = beg; e < end; ++e) m = fmaxf(m, gatres_leaky(at(e) + adst));
      float Z = 0.f;
      for (int e = beg; e < end; ++e) Z = Z + expf(gatres_leaky(at(e) + adst) - m);
      Z = Z + GATRES_SOFTMAX_EPS;
      for (int e = beg; e < end; ++e) {
        const float al = expf(gatres_leaky(at(e) + adst) - m) / Z;
        alpha_g[(unsigned)((eb + e) * H + hd)] = al;
        if constexpr (ALDS) alpha_l[e * H + hd] = al;
      }
    }
  }
}

// K2 forward, sub-stage B: out[r] = sum_e alpha_e * h[src(e)] + bias (+ReLU).  HC/4 lanes per row; the per-row chain is
// rowptr -> col -> {alpha, h rows} -> fma.  alpha: [ab2 + e] (LDS table with ab2 = 0, or the global array with ab2 = eb).
// TWO rows per lane group per trip (UR): one workgroup has no spare parallelism to hide the dependent LDS/L2 hops,
// so the two rows' loads are issued together and their chains overlap.
template <bool RELU, int H, int C, int THREADS, int UR = 2>
__device__ __forceinline__ void seg_gather(Rows rw, const u16* rp, const u16* col, const float* hsrc, int hb,
                                           const float* alpha, int ab2, const float* __restrict__ bias, float* out,
                                           int ob, float* out_pub = nullptr, int opb = 0,
                                           unsigned long long* mask64 = nullptr) {
  constexpr int HC = H * C, G = HC / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const float4 b = ld4(bias + c0);
  const int rounds = (rw.hi - rw.lo + RPP * UR - 1) / (RPP * UR);
  for (int it = 0; it < rounds; ++it) {
    // waves whose rows of this trip all lie beyond the part skip it (wave-uniform): an idle wave that walks the trip with
    // clamped rows costs the busy waves of its SIMD their issue slots
    if (rw.lo + it * UR * RPP + (int)((threadIdx.x & ~63u) / G) >= rw.hi) continue;
    int r[UR], beg[UR], deg[UR];
    bool valid[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      r[u] = rw.lo + (it * UR + u) * RPP + threadIdx.x / G;
      valid[u] = r[u] < rw.hi;
      if (!valid[u]) r[u] = rw.hi - 1;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) { beg[u] = rp[r[u]]; deg[u] = (int)rp[r[u] + 1] - beg[u]; }
    const int dmax = wave_max_deg(max_of<UR>(deg));
    float4 acc[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) acc[u] = f4zero();
    if (__builtin_expect(max_of<UR>(deg) <= MAXD, 1)) {      // (the edge-at-a-time path is cold: laid out of line)
      float4 v[UR][MAXD];
      float al[UR][MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const int e = beg[u] + min(k, deg[u] - 1);
            v[u][k] = ld4(hsrc + (unsigned)((hb + (int)col[e]) * HC + c0));
            al[u][k] = alpha[(unsigned)((ab2 + e) * H + hd)];
          }
        }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) gatres_axpy4(acc[u], k < deg[u] ? al[u][k] : 0.f, v[u][k]);   // 0 on padding
        }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u)
        for (int e = beg[u]; e < beg[u] + deg[u]; ++e)
          gatres_axpy4(acc[u], alpha[(unsigned)((ab2 + e) * H + hd)],
                       ld4(hsrc + (unsigned)((hb + (int)col[e]) * HC + c0)));
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      add4(acc[u], b);
      if (RELU) {
        acc[u].x = fmaxf(acc[u].x, 0.f); acc[u].y = fmaxf(acc[u].y, 0.f);
        acc[u].z = fmaxf(acc[u].z, 0.f); acc[u].w = fmaxf(acc[u].w, 0.f);
      }
      if (valid[u]) {
        st4(out + (unsigned)((ob + r[u]) * HC + c0), acc[u]);
        if (out_pub) st4(out_pub + (unsigned)((opb + r[u]) * HC + c0), acc[u]);
      }
      if constexpr (RELU && G <= 16) {
        if (mask64) {                                             // (workgroup-uniform)
          const unsigned long long w = relu_bits<G, 16>(acc[u]);
          if (valid[u] && threadIdx.x % G == 0) mask64[r[u]] = w;
        }
      }
    }
  }
}

// K3 forward: out = relu(mean_{j->r} y[j] + x0[r]).  UR rows per lane group per trip.
template <int C, int THREADS, int UR = 2>
__device__ __forceinline__ void seg_mean_fwd(Rows rw, int em, const u16* mrp, const u16* mcol, const float* y, int yb,
                                             const float* x0, int xb, float* out, int ob, float* out2 = nullptr,
                                             int o2b = 0, unsigned* mask32 = nullptr) {
  constexpr int G = C / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int rounds = (rw.hi - rw.lo + RPP * UR - 1) / (RPP * UR);
  const int elast = max(em - 1, 0);
  for (int it = 0; it < rounds; ++it) {
    // waves whose rows of this trip all lie beyond the part skip it (wave-uniform): an idle wave that walks the trip with
    // clamped rows costs the busy waves of its SIMD their issue slots
    if (rw.lo + it * UR * RPP + (int)((threadIdx.x & ~63u) / G) >= rw.hi) continue;
    int r[UR], beg[UR], deg[UR];
    bool valid[UR];
    float4 rr[UR], acc[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      r[u] = rw.lo + (it * UR + u) * RPP + threadIdx.x / G;
      valid[u] = r[u] < rw.hi;
      if (!valid[u]) r[u] = rw.hi - 1;
      acc[u] = f4zero();
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      beg[u] = mrp[r[u]]; deg[u] = (int)mrp[r[u] + 1] - beg[u];
      rr[u] = ld4(x0 + (unsigned)((xb + r[u]) * C + c0));
    }
    const int dmax = wave_max_deg(max_of<UR>(deg));
    if (__builtin_expect(max_of<UR>(deg) <= MAXD, 1)) {      // (the edge-at-a-time path is cold: laid out of line)
      float4 v[UR][MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const int jj = k < deg[u] ? (int)mcol[min(beg[u] + k, elast)] : r[u];
            v[u][k] = ld4(y + (unsigned)((yb + jj) * C + c0));
          }
        }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) gatres_axpy4(acc[u], k < deg[u] ? 1.f : 0.f, v[u][k]);   // fma(1,v,acc) = acc+v
        }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u)
        for (int e = beg[u]; e < beg[u] + deg[u]; ++e) add4(acc[u], ld4(y + (unsigned)((yb + mcol[e]) * C + c0)));
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const float cnt = (float)max(deg[u], 1);
      float4 o;
      o.x = fmaxf(acc[u].x / cnt + rr[u].x, 0.f); o.y = fmaxf(acc[u].y / cnt + rr[u].y, 0.f);
      o.z = fmaxf(acc[u].z / cnt + rr[u].z, 0.f); o.w = fmaxf(acc[u].w / cnt + rr[u].w, 0.f);
      if (valid[u]) {
        st4(out + (unsigned)((ob + r[u]) * C + c0), o);
        if (out2) st4(out2 + (unsigned)((o2b + r[u]) * C + c0), o);
      }
      if constexpr (G <= 8) {
        if (mask32) {                                             // (workgroup-uniform)
          const unsigned w = (unsigned)relu_bits<G, 8>(o);
          if (valid[u] && threadIdx.x % G == 0) mask32[r[u]] = w;
        }
      }
    }
  }
}

// K3 backward: g_y[r] = sum over out-edges (r -> i) of g_pre[i] / max(indeg(i), 1)
template <int C, int THREADS, int UR = 1>
__device__ __forceinline__ void seg_mean_bwd(Rows rw, int em, const u16* mrp, const u16* mtrp, const u16* mtdst,
                                             const float* g_pre, int pb, float* g_y, int yb,
                                             float* g_y_pub = nullptr, int ypb = 0) {
  constexpr int G = C / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int rounds = (rw.hi - rw.lo + RPP * UR - 1) / (RPP * UR);
  const int elast = max(em - 1, 0);
  for (int it = 0; it < rounds; ++it) {
    // waves whose rows of this trip all lie beyond the part skip it (wave-uniform): an idle wave that walks the trip with
    // clamped rows costs the busy waves of its SIMD their issue slots
    if (rw.lo + it * UR * RPP + (int)((threadIdx.x & ~63u) / G) >= rw.hi) continue;
    int r[UR], beg[UR], deg[UR];
    bool valid[UR];
    float4 acc[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      r[u] = rw.lo + (it * UR + u) * RPP + threadIdx.x / G;
      valid[u] = r[u] < rw.hi;
      if (!valid[u]) r[u] = rw.hi - 1;
      acc[u] = f4zero();
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) { beg[u] = mtrp[r[u]]; deg[u] = (int)mtrp[r[u] + 1] - beg[u]; }
    const int dmax = wave_max_deg(max_of<UR>(deg));
    if (__builtin_expect(max_of<UR>(deg) <= MAXD, 1)) {      // (the edge-at-a-time path is cold: laid out of line)
      float4 v[UR][MAXD];
      float cnt[UR][MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const int ii = k < deg[u] ? (int)mtdst[min(beg[u] + k, elast)] : r[u];
            cnt[u][k] = (float)max((int)mrp[ii + 1] - (int)mrp[ii], 1);
            v[u][k] = ld4(g_pre + (unsigned)((pb + ii) * C + c0));
          }
        }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const float w = k < deg[u] ? 1.f : 0.f;                      // x + 0 * q == x exactly
            acc[u].x = fmaf(w, v[u][k].x / cnt[u][k], acc[u].x); acc[u].y = fmaf(w, v[u][k].y / cnt[u][k], acc[u].y);
            acc[u].z = fmaf(w, v[u][k].z / cnt[u][k], acc[u].z); acc[u].w = fmaf(w, v[u][k].w / cnt[u][k], acc[u].w);
          }
        }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u)
        for (int t = beg[u]; t < beg[u] + deg[u]; ++t) {
          const int ii = mtdst[t];
          const float cnt = (float)max((int)mrp[ii + 1] - (int)mrp[ii], 1);
          const float4 v = ld4(g_pre + (unsigned)((pb + ii) * C + c0));
          acc[u].x = acc[u].x + v.x / cnt; acc[u].y = acc[u].y + v.y / cnt;
          acc[u].z = acc[u].z + v.z / cnt; acc[u].w = acc[u].w + v.w / cnt;
        }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u)
      if (valid[u]) {
        st4(g_y + (unsigned)((yb + r[u]) * C + c0), acc[u]);
        if (g_y_pub) st4(g_y_pub + (unsigned)((ypb + r[u]) * C + c0), acc[u]);
      }
  }
}

// K2 backward, destination-major, in two sub-stages (same arithmetic and order as the per-op kernel):
//   A  seg_edge_dots   : ga_e = <g_out[i,h,:], h[j,h,:]> for every in-edge, HC/4 lanes per row, UR rows per trip,
//                        head sum by DPP row operations; ga goes to the g_e table ([eb2 + e], LDS or scratch).
//   B  seg_softmax_bwd : ONE thread per (row, head): S = sum alpha*ga ; g_e = alpha*(ga - S) * LeakyReLU' ;
//                        g_a_dst = sum g_e.  Overwrites ga with g_e in place.
template <int H, int C, int THREADS, int UR>
__device__ __forceinline__ void seg_edge_dots(Rows rw, int n0, const u16* rp, const u16* col, const float* g_out, int gb,
                                              const float* __restrict__ h, float* g_e, int eb2) {
  constexpr int HC = H * C, G = HC / 4, LH = C / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const int rounds = (rw.hi - rw.lo + RPP * UR - 1) / (RPP * UR);
  for (int it = 0; it < rounds; ++it) {
    // waves whose rows of this trip all lie beyond the part skip it (wave-uniform): an idle wave that walks the trip with
    // clamped rows costs the busy waves of its SIMD their issue slots
    if (rw.lo + it * UR * RPP + (int)((threadIdx.x & ~63u) / G) >= rw.hi) continue;
    int r[UR], beg[UR], deg[UR];
    bool leader[UR];
    float4 go[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      r[u] = rw.lo + (it * UR + u) * RPP + threadIdx.x / G;
      const bool valid = r[u] < rw.hi;         // every lane stays in the loop: the head reduction spans the head's lanes
      if (!valid) r[u] = rw.hi - 1;
      leader[u] = valid && (c0 % C) == 0;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      beg[u] = rp[r[u]]; deg[u] = (int)rp[r[u] + 1] - beg[u];
      go[u] = ld4(g_out + (unsigned)((gb + r[u]) * HC + c0));
    }
    const int dmax = wave_max_deg(max_of<UR>(deg));
    if (__builtin_expect(max_of<UR>(deg) <= MAXD, 1)) {      // (the edge-at-a-time path is cold: laid out of line)
      float4 hv[UR][MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u)
            hv[u][k] = ld4(h + (unsigned)((n0 + (int)col[beg[u] + min(k, deg[u] - 1)]) * HC + c0));
        }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const float ga = gatres_head_reduce<LH>(gatres_head_dot4(go[u], hv[u][k]));
            if (leader[u] && k < deg[u]) g_e[(unsigned)((eb2 + beg[u] + k) * H + hd)] = ga;
          }
        }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u)
        for (int e = beg[u]; e < beg[u] + deg[u]; ++e) {
          const float ga = gatres_head_reduce<LH>(gatres_head_dot4(go[u], ld4(h + (unsigned)((n0 + col[e]) * HC + c0))));
          if (leader[u]) g_e[(unsigned)((eb2 + e) * H + hd)] = ga;
        }
    }
  }
}

template <int H, int THREADS>
__device__ __forceinline__ void seg_softmax_bwd(Rows rw, int n0, int e0, const u16* rp, const u16* col,
                                                const float* __restrict__ alpha, const float* __restrict__ a_src,
                                                const float* __restrict__ a_dst, float* g_e, int eb2,
                                                float* g_a_dst, int db, float* g_e_pub, int epb, float* g_ad_pub,
                                                int dpb) {
  for (int idx = rw.lo * H + threadIdx.x; idx < rw.hi * H; idx += THREADS) {
    const int r = idx / H, hd = idx % H;
    const int beg = rp[r], deg = (int)rp[r + 1] - beg;
    const float adst = a_dst[(unsigned)((n0 + r) * H + hd)];
    float S = 0.f, gad = 0.f;
    if (__builtin_expect(deg <= MAXD, 1)) {
      float al[MAXD], ga[MAXD], raw[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        const int kk = min(k, deg - 1);
        al[k] = alpha[(unsigned)((e0 + beg + kk) * H + hd)];
        ga[k] = g_e[(unsigned)((eb2 + beg + kk) * H + hd)];
        raw[k] = a_src[(unsigned)((n0 + (int)col[beg + kk]) * H + hd)] + adst;
      }
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        al[k] = k < deg ? al[k] : 0.f;                                   // padding slots weigh nothing
        S = fmaf(al[k], ga[k], S);
      }
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        const float gs = al[k] * (ga[k] - S);
        const float ge = raw[k] > 0.f ? gs : gs * GATRES_NEG_SLOPE;
        if (k < deg) {
          g_e[(unsigned)((eb2 + beg + k) * H + hd)] = ge;
          if (g_e_pub) g_e_pub[(unsigned)((epb + beg + k) * H + hd)] = ge;      // global copy for the partner CUs
        }
        gad = gad + ge;                                                  // ge == 0 on padding slots
      }
    } else {
      const int end = beg + deg;
      for (int e = beg; e < end; ++e)
        S = fmaf(alpha[(unsigned)((e0 + e) * H + hd)], g_e[(unsigned)((eb2 + e) * H + hd)], S);
      for (int e = beg; e < end; ++e) {
        const float gs = alpha[(unsigned)((e0 + e) * H + hd)] * (g_e[(unsigned)((eb2 + e) * H + hd)] - S);
        const float rw = a_src[(unsigned)((n0 + (int)col[e]) * H + hd)] + adst;
        const float ge = rw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
        g_e[(unsigned)((eb2 + e) * H + hd)] = ge;
        if (g_e_pub) g_e_pub[(unsigned)((epb + e) * H + hd)] = ge;
        gad = gad + ge;
      }
    }
    g_a_dst[(unsigned)((db + r) * H + hd)] = gad;
    if (g_ad_pub) g_ad_pub[(unsigned)((dpb + r) * H + hd)] = gad;
  }
}

// K2 backward, source-major over CSR^T.  UR rows per lane group per trip.
template <int H, int C, int THREADS, int UR = 1>
__device__ __forceinline__ void seg_agg_bwd_src(Rows rw, int e0, const u16* trp, const u16* teid, const u16* tdst,
                                                const float* g_out, int gb, const float* __restrict__ alpha,
                                                const float* g_e, int eb2, const float* g_a_dst, int db,
                                                const float* __restrict__ att_src,
                                                const float* __restrict__ att_dst, float* g_h, int hb,
                                                float* keep_gas, float* keep_gad, float* g_h2 = nullptr,
                                                int h2b = 0) {
  constexpr int HC = H * C, G = HC / 4, RPP = THREADS / G;
  const int c0 = (threadIdx.x % G) * 4;
  const int hd = c0 / C;
  const float4 as = ld4(att_src + c0), ad = ld4(att_dst + c0);
  const int rounds = (rw.hi - rw.lo + RPP * UR - 1) / (RPP * UR);
  for (int it = 0; it < rounds; ++it) {
    // waves whose rows of this trip all lie beyond the part skip it (wave-uniform): an idle wave that walks the trip with
    // clamped rows costs the busy waves of its SIMD their issue slots
    if (rw.lo + it * UR * RPP + (int)((threadIdx.x & ~63u) / G) >= rw.hi) continue;
    int r[UR], beg[UR], deg[UR];
    bool valid[UR];
    float4 acc[UR];
    float gas[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      r[u] = rw.lo + (it * UR + u) * RPP + threadIdx.x / G;
      valid[u] = r[u] < rw.hi;
      if (!valid[u]) r[u] = rw.hi - 1;
      acc[u] = f4zero();
      gas[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) { beg[u] = trp[r[u]]; deg[u] = (int)trp[r[u] + 1] - beg[u]; }   // deg >= 1
    const int dmax = wave_max_deg(max_of<UR>(deg));
    if (__builtin_expect(max_of<UR>(deg) <= MAXD, 1)) {      // (the edge-at-a-time path is cold: laid out of line)
      float al[UR][MAXD], ge[UR][MAXD];
      float4 v[UR][MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const int kk = beg[u] + min(k, deg[u] - 1);
            const int e = teid[kk], ii = tdst[kk];
            v[u][k] = ld4(g_out + (unsigned)((gb + ii) * HC + c0));
            al[u][k] = alpha[(unsigned)((e0 + e) * H + hd)];
            ge[u][k] = g_e[(unsigned)((eb2 + e) * H + hd)];
          }
        }
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < dmax) {
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const bool ok = k < deg[u];
            gas[u] = gas[u] + (ok ? ge[u][k] : 0.f);
            gatres_axpy4(acc[u], ok ? al[u][k] : 0.f, v[u][k]);
          }
        }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u)
        for (int t = beg[u]; t < beg[u] + deg[u]; ++t) {
          const int e = teid[t], ii = tdst[t];
          gas[u] = gas[u] + g_e[(unsigned)((eb2 + e) * H + hd)];
          gatres_axpy4(acc[u], alpha[(unsigned)((e0 + e) * H + hd)], ld4(g_out + (unsigned)((gb + ii) * HC + c0)));
        }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const bool leader = valid[u] && (c0 % C) == 0;
      const float gad = g_a_dst[(unsigned)((db + r[u]) * H + hd)];
      if (leader) {                      // kept (global row hb + r) for the deferred att_src / att_dst gradients
        keep_gas[(unsigned)((hb + r[u]) * H + hd)] = gas[u];
        keep_gad[(unsigned)((hb + r[u]) * H + hd)] = gad;
      }
      gatres_axpy4(acc[u], gas[u], as);
      gatres_axpy4(acc[u], gad, ad);
      if (valid[u]) {
        st4(g_h + (unsigned)((hb + r[u]) * HC + c0), acc[u]);
        if (g_h2) st4(g_h2 + (unsigned)((h2b + r[u]) * HC + c0), acc[u]);
      }
    }
  }
}

// att_src / att_dst gradients of one GATConv for this segment -> the segment's slab (deferred launch).
// thread = (column c, row group rg); 4 rows in flight; partials combined across row groups through LDS.
template <int H, int C, int THREADS, bool NT = false>
__device__ __forceinline__ void seg_att_grads(int n, const float* __restrict__ h, int hb,
                                              const float* __restrict__ g_a_src, const float* __restrict__ g_a_dst,
                                              int db, float* __restrict__ slab_as, float* __restrict__ slab_ad,
                                              float* red) {
  constexpr int HC = H * C, R = THREADS / HC, U = 16;      // U rows in flight per thread: the loop is latency-bound
  const int c = threadIdx.x % HC, rg = threadIdx.x / HC;
  const int hd = c / C;
  float as = 0.f, ad = 0.f;
  if (rg < R) {
    for (int r0 = rg; r0 < n; r0 += U * R) {
      float hv[U], gs[U], gd[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int r = r0 + k * R;
        const bool ok = r < n;
        const int rr = ok ? r : 0;
        hv[k] = !ok ? 0.f : (NT ? __builtin_nontemporal_load(h + (size_t)(hb + rr) * HC + c) : h[(size_t)(hb + rr) * HC + c]);
        gs[k] = ok ? g_a_src[(size_t)(db + rr) * H + hd] : 0.f;
        gd[k] = ok ? g_a_dst[(size_t)(db + rr) * H + hd] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        as = fmaf(gs[k], hv[k], as);
        ad = fmaf(gd[k], hv[k], ad);
      }
    }
  }
  __syncthreads();
  red[threadIdx.x] = as; red[THREADS + threadIdx.x] = ad;
  __syncthreads();
  if (threadIdx.x < HC) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < R; ++k) { s0 += red[k * HC + c]; s1 += red[THREADS + k * HC + c]; }
    slab_as[c] = s0; slab_ad[c] = s1;
  }
}

// bias gradient of one GATConv = column sums of its g_out table.  Two halves that ride on barriers the backward
// already has: `part` (per-thread partial -> red) runs inside the softmax-backward stage, `finish` (HC threads sum
// the row groups -> slab) at the start of the source-major stage.
template <int HC, int THREADS>
__device__ __forceinline__ void seg_bias_part(Rows rw, const float* g_out, int gb, float* red) {
  constexpr int R = THREADS / HC;
  const int c = threadIdx.x % HC, rg = threadIdx.x / HC;
  float ab = 0.f;
  if (rg < R) {
    for (int r0 = rw.lo + rg; r0 < rw.hi; r0 += 4 * R) {
      float go[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * R;
        go[k] = r < rw.hi ? g_out[(unsigned)((gb + (r < rw.hi ? r : rw.lo)) * HC + c)] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) ab += go[k];
    }
  }
  red[threadIdx.x] = ab;
}
// The same partials (thread t's of seg_bias_part for every t in [0, THREADS)) formed by the threads [T0, T0 + NT) only --
// by waves that hold no rows in the sparse stage they ride on -- and the finish by the HC threads from T0 on.
template <int HC, int THREADS, int T0, int NT>
__device__ __forceinline__ void seg_bias_part_by(Rows rw, const float* g_out, int gb, float* red) {
  constexpr int R = THREADS / HC;
  const int t0 = (int)threadIdx.x - T0;
  if (t0 < 0 || t0 >= NT) return;
  for (int idx = t0; idx < THREADS; idx += NT) {
    const int c = idx % HC, rg = idx / HC;
    float ab = 0.f;
    for (int r0 = rw.lo + rg; r0 < rw.hi; r0 += 4 * R) {
      float go[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * R;
        go[k] = r < rw.hi ? g_out[(unsigned)((gb + (r < rw.hi ? r : rw.lo)) * HC + c)] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) ab += go[k];
    }
    red[idx] = ab;
  }
}
template <int HC, int THREADS, int T0>
__device__ __forceinline__ void seg_bias_finish_by(const float* red, float* __restrict__ slab_b) {
  constexpr int R = THREADS / HC;
  const int t = (int)threadIdx.x - T0;
  if (t >= 0 && t < HC) {
    float s = 0.f;
    for (int k = 0; k < R; ++k) s += red[k * HC + t];
    slab_b[t] = s;
  }
}
template <int HC, int THREADS>
__device__ __forceinline__ void seg_bias_finish(const float* red, float* __restrict__ slab_b) {
  constexpr int R = THREADS / HC;
  if (threadIdx.x < HC) {
    float s = 0.f;
    for (int k = 0; k < R; ++k) s += red[k * HC + threadIdx.x];
    slab_b[threadIdx.x] = s;
  }
}

// lin1 backward for this segment: g_x = g_out (x) w (ReLU-masked), slab partials of g_w, g_b.
template <int NC, int THREADS>
__device__ __forceinline__ void seg_lin1_bwd(Rows rw, int n0, const int* __restrict__ perm, const float* __restrict__ g_out,
                                             const float* __restrict__ x, const float* __restrict__ w, float* g_x,
                                             float* g_x2, float* __restrict__ slab_w, float* __restrict__ slab_b,
                                             int relu_mask, float* red, float* g_x3 = nullptr) {
  constexpr int R = THREADS / NC;
  const int c = threadIdx.x % NC, rg = threadIdx.x / NC;
  const float wv = w[c];
  float aw = 0.f, ab = 0.f;
  for (int r = rw.lo + rg; r < rw.hi; r += R) {
    const size_t node = (size_t)n0 + r;
    const float go = g_out[ext_id(perm, n0 + r)];
    const float xv = x[(unsigned)(r * NC + c)];          // x: the segment's saved final activation, local rows
    aw = fmaf(go, xv, aw);
    ab += go;
    const float gv = (relu_mask && !(xv > 0.f)) ? 0.f : go * wv;
    g_x[node * NC + c] = gv;
    if (g_x2) g_x2[(size_t)r * NC + c] = gv;
    if (g_x3) g_x3[(size_t)r * NC + c] = gv;
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
__device__ __forceinline__ void seg_lin0_bwd(Rows rw, int n0, const int* __restrict__ perm, const float* __restrict__ g, const float* __restrict__ x,
                                             const uint8_t* __restrict__ mask, float* __restrict__ slab_w,
                                             float* __restrict__ slab_b, float* red, const float* g_local = nullptr,
                                             const unsigned* x_local = nullptr) {
  // g_local: the same gradient rows in LDS, indexed by LOCAL row (window kernel: the last dX stage left them there)
  // x_local: the masked input of the rows as lin0 saw it, float bits in LDS indexed by local row (window kernel, keep-in-LDS)
  constexpr int R = THREADS / NC;
  const int c = threadIdx.x % NC, rg = threadIdx.x / NC;
  float aw = 0.f, ab = 0.f;
  for (int r = rw.lo + rg; r < rw.hi; r += R) {
    const size_t node = (size_t)n0 + r;
    float xv;
    if (x_local) xv = __uint_as_float(x_local[r]);
    else {
      const int en = ext_id(perm, n0 + r);
      xv = (mask && mask[en]) ? 0.f : x[en];
    }
    const float gv = g_local ? g_local[(size_t)r * NC + c] : g[node * NC + c];
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

// ------------------------------------------------------------------------------------------ deferred gradients
// Deferred parameter gradients of the GATConvs.  Nothing on the backward's dependency chain reads dW / g_att, so
// the per-snapshot workgroups only keep g_h and g_alpha_src / g_alpha_dst per block; one ITEM = (segment, block,
// conv) turns them into slab entries (and folds the bias partials of a split segment).  Items run either on spare
// CUs of the same launch (consumer workgroups, see gatres_fused_kernel) or in a second launch with one workgroup
// per item (param_grads_kernel).  Writes are disjoint slab ranges: deterministic.
struct ParamGradArgs {
  const int* seg_ptr;
  const float* saved;
  const float* keep;
  float* slabs;
  const float* part_slabs;
  int M;
  Layout L;
  SegLayout SL;
  const float* wt = nullptr;                  // (consumer_item_dma: the transposed conv weights, gatres_fused_prepare_backward)
  // param_grads_stream_kernel, training step: workgroup 0 copies the optimizer's step count and the split launches' fault word
  // into snap[0..1] -- stable words for the sampling tail of the update launch that follows (reduce_adam_kernel)
  const unsigned long long* step_counter = nullptr;
  const unsigned* status = nullptr;
  unsigned long long* snap = nullptr;
  unsigned long long* dstamps = nullptr;      // diagnostic (consumer 0 under gatres_fused_set_stamps): steps inside an item
};

// A slab entry.  AG: the entry is read by ANOTHER workgroup of the same launch (the column's last arriver of
// param_grads_finish_kernel, k_fused_host.hip): an agent-scope store (sc1: written through, visible from every XCD).
template <bool AG>
__device__ __forceinline__ void slab_st(float* p, float v) {
  if constexpr (AG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// slab[off .. off+cnt) of a segment = sum over its parts' partial rows (fixed order)
template <int THREADS, bool AG = false>
__device__ __forceinline__ void fold_parts(const ParamGradArgs& a, int seg, int64_t off, int cnt) {
  for (int idx = threadIdx.x; idx < cnt; idx += THREADS) {
    float sum = 0.f;
    for (int p = 0; p < a.M; ++p) sum += a.part_slabs[((int64_t)seg * a.M + p) * a.L.slab_stride + off + idx];
    slab_st<AG>(a.slabs + (int64_t)seg * a.L.slab_stride + off + idx, sum);
  }
}

// part: NW * 2*NC*NC floats of LDS when the block form applies (nc 16 / 32), red: 3 * THREADS floats
template <int NC, int THREADS, bool NT>
__device__ __forceinline__ void param_grads_item(const ParamGradArgs& a, int seg, int b, int conv, bool fold_lin,
                                                 float* part, float* red) {
  constexpr bool BLK = NC >= 16 && NC <= 32;
  constexpr int NW = THREADS / 64;
  const Layout& L = a.L;
  const SegLayout& SL = a.SL;
  const int n0 = uni(a.seg_ptr[seg]), n = uni(a.seg_ptr[seg + 1]) - n0;
  const float* base = a.saved + (int64_t)seg * SL.total + (int64_t)b * SL.bstride;
  const float* keep = a.keep + (int64_t)b * L.keep_stride;
  const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
  float* sb = a.slabs + (int64_t)seg * L.slab_stride + po;
  if (a.M > 1) {          // bias / lin0 / lin1 partials of a split segment
    if (conv == 0) fold_parts<THREADS>(a, seg, po + L.c1_b, 2 * NC);
    else           fold_parts<THREADS>(a, seg, po + L.c2_b, NC);
    if (fold_lin) {
      fold_parts<THREADS>(a, seg, L.p_lin0_w, 2 * NC);
      fold_parts<THREADS>(a, seg, L.p_lin1_w, NC + 1);
    }
  }
  if (conv == 0) {
    if constexpr (BLK) seg_dw_blk<2 * NC, NC, THREADS, NT>(n, NW, keep + L.k_gh1, n0, base + SL.xin, 0, sb + L.c1_W, part);
    else               seg_dw<2 * NC, NC, THREADS>(n, keep + L.k_gh1, n0, base + SL.xin, 0, sb + L.c1_W, red);
    seg_att_grads<2, NC, THREADS, NT>(n, base + SL.h1, 0, keep + L.k_gas1, keep + L.k_gad1, n0, sb + L.c1_as,
                                  sb + L.c1_ad, red);
  } else {
    if constexpr (BLK) seg_dw_blk<NC, 2 * NC, THREADS, NT>(n, NW, keep + L.k_gh2, n0, base + SL.o1, 0, sb + L.c2_W, part);
    else               seg_dw<NC, 2 * NC, THREADS>(n, keep + L.k_gh2, n0, base + SL.o1, 0, sb + L.c2_W, red);
    seg_att_grads<1, NC, THREADS, NT>(n, base + SL.h2, 0, keep + L.k_gas2, keep + L.k_gad2, n0, sb + L.c2_as,
                                  sb + L.c2_ad, red);
  }
}

template <int NC>
__global__ __launch_bounds__(512) void param_grads_stream_kernel(const ParamGradArgs a);

template <int NC, int THREADS>
__global__ __launch_bounds__(THREADS) void param_grads_kernel(const ParamGradArgs a) {
  constexpr bool BLK = NC >= 16 && NC <= 32;
  constexpr int NW = THREADS / 64;
  __shared__ __attribute__((aligned(16))) float part[BLK ? NW * 2 * NC * NC : 4];
  __shared__ float red[3 * THREADS];
  const int conv = blockIdx.x & 1, b = (blockIdx.x >> 1) % a.L.nb, seg = (blockIdx.x >> 1) / a.L.nb;
  param_grads_item<NC, THREADS, false>(a, seg, b, conv, b == 0 && conv == 0, part, red);
}


// ---- consumer items, streamed form (nc 16 / 32, 1024 threads): the item's tables flow through four LDS chunk buffers of
// 64 rows.  The LAST FOUR waves only issue LDS-DMA (global_load_lds: no registers, no waits in the issuing wave) two
// chunks ahead of the TWELVE compute waves, which read operands from LDS only -- an item then costs its bytes over
// what one CU streams from HBM beside 192 busy ones (~40 GB/s measured) instead of five dependent rounds of register
// loads (13 - 15 us per item before; profiles/r02_consumer_items.txt).  One barrier per chunk.
//   dW:  wave w takes 4-row steps w, w + 12 of a chunk and keeps the whole [HC, K] block in MFMA accumulators (the
//        permuted operand map of seg_dw_blk: vector LDS reads of contiguous features); the twelve partial blocks meet
//        in LDS after the last chunk and are summed in wave order -> slab.  Deterministic.
//   attention-vector gradients WITHOUT reading h (a third of the item's bytes): with h = x W^T,
//        g_att_src[c] = sum_r g_a_src[r, hd(c)] h[r, c] = sum_k W[c, k] T[hd(c), k],  T = [g_a_src | g_a_dst]^T x  ([2H, K]),
//        and T is one more MFMA tile row on the x operands the dW already has in registers.  Same value up to fp32
//        reassociation (sum over rows first, then over k); the parameter-gradient tests hold it to their usual tolerance.
constexpr int CI_DMA_WAVES = 4, CI_NBUF = 4, CI_DEPTH = CI_NBUF - 1, CI_CR = 64;   // chunks in flight ahead of the compute waves: DEPTH - 1
template <int NC, int CONV>
struct CiGeom {
  static constexpr int HC = CONV == 0 ? 2 * NC : NC, K = CONV == 0 ? NC : 2 * NC, H = CONV == 0 ? 2 : 1;
  static constexpr int ROW = HC + K + 2 * H;                     // floats of one row over the four tables
  static constexpr int CB = CI_CR * ROW;                         // floats of one chunk buffer
  // LDS-DMA instructions one DMA wave issues per chunk (16-byte form: 256 floats per instruction; the two small tables
  // share one 4-byte-form instruction per wave)
  static constexpr int N_G = CI_CR * HC / 256 / CI_DMA_WAVES, N_X = CI_CR * K / 256 / CI_DMA_WAVES;
  static constexpr int PER_WAVE = N_G + N_X + 1;
  static_assert(CI_CR * HC % (256 * CI_DMA_WAVES) == 0 && CI_CR * K % (256 * CI_DMA_WAVES) == 0, "chunk / wave split");
  static_assert(2 * CI_CR * H <= 64 * CI_DMA_WAVES && 2 * CI_CR * H % CI_DMA_WAVES == 0, "small tables: one 4-byte DMA instruction per wave");
};

// (NBUF chunk buffers: four in the consumers of the window kernel, which own a CU's whole LDS; three in the stand-alone
//  launch, whose 512-thread workgroups then fit a CU in pairs -- one streams while the other folds its partial blocks)
template <int NC, int THREADS, int CONV, int NBUF = CI_NBUF, bool AG = false>
__device__ __forceinline__ void consumer_item_dma(const ParamGradArgs& a, int seg, int b, float* lds) {
  using Gm = CiGeom<NC, CONV>;
  constexpr int DEPTH = NBUF - 1;
  constexpr int HC = Gm::HC, K = Gm::K, H = Gm::H, CB = Gm::CB, CR = CI_CR;
  constexpr int NW = THREADS / 64, CW = NW - CI_DMA_WAVES;                         // compute waves
  constexpr int VC = HC / 16, VK = K / 16;
  const Layout& L = a.L;
  const SegLayout& SL = a.SL;
  const int n0 = uni(a.seg_ptr[seg]), n = uni(a.seg_ptr[seg + 1]) - n0;
  const float* base = a.saved + (int64_t)seg * SL.total + (int64_t)b * SL.bstride;
  const float* keep = a.keep + (int64_t)b * L.keep_stride;
  const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
  float* sb = a.slabs + (int64_t)seg * L.slab_stride + po;
  const float* gG = keep + (CONV == 0 ? L.k_gh1 : L.k_gh2) + (int64_t)n0 * HC;      // [n][HC]
  const float* gX = base + (CONV == 0 ? SL.xin : SL.o1);                              // [n][K]
  const float* gS = keep + (CONV == 0 ? L.k_gas1 : L.k_gas2) + (int64_t)n0 * H;      // [n][H]
  const float* gD = keep + (CONV == 0 ? L.k_gad1 : L.k_gad2) + (int64_t)n0 * H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunks = (n + CR - 1) / CR;
  // chunk buffer: [G CR x HC | X CR x K | g_a_src CR x H | g_a_dst CR x H]
  auto bufp = [&](int c) { return lds + (c % NBUF) * CB; };
  // [CW][4][K]: the waves' T blocks, after the loop (three buffers: behind the partial dW blocks, inside the dead buffers)
  float* tred = NBUF < CI_NBUF ? lds + CW * HC * K : lds + NBUF * CB;
  static_assert(NBUF >= CI_NBUF || CW * HC * K + CW * 4 * K + 4 * K <= NBUF * CB, "the item's epilogue lives in the chunk buffers");
  float* tsum = tred + CW * 4 * K;                               // [4][K]
#define ISTAMP(k) do { if (GATRES_DIAG && a.dstamps && threadIdx.x == 0) a.dstamps[k] = wall_clock64(); } while (0)
  ISTAMP(0);
  if (wave >= CW) {
    // ------------------------------------------------------------------ DMA waves
    const int dw = wave - CW;
    auto issue = [&](int c) {
      float* B = bufp(c);
      const int r0 = c * CR;
      // rows beyond the segment are read from its last row (never used: the compute waves zero them) so that every
      // chunk costs the same number of instructions -- the waits below count them
      auto rows16 = [&](const float* src, float* dst, int W, int cnt) {
        for (int k = 0; k < cnt; ++k) {
          const int f = (dw * cnt + k) * 256 + lane * 4;                 // float index inside the chunk's table
          const int r = min(r0 + f / W, n - 1);
          __builtin_amdgcn_global_load_lds(src + (size_t)r * W + f % W, dst + (dw * cnt + k) * 256, 16, 0, 0);
        }
      };
      rows16(gG, B, HC, Gm::N_G);
      rows16(gX, B + CR * HC, K, Gm::N_X);
      {  // g_a_src | g_a_dst: 2 * CR * H floats, 4 bytes per lane, the same share (and ONE instruction) for every DMA wave
        constexpr int SH = 2 * CR * H / CI_DMA_WAVES;
        const int f = dw * SH + min(lane, SH - 1);
        const int which = f / (CR * H), e = f % (CR * H);
        const int r = min(r0 + e / H, n - 1);
        const float* src = (which ? gD : gS) + (size_t)r * H + e % H;
        if (lane < SH) __builtin_amdgcn_global_load_lds(src, B + CR * (HC + K) + dw * SH, 4, 0, 0);
      }
    };
    for (int c = 0; c < min(chunks, DEPTH); ++c) issue(c);
    for (int c = 0; c < chunks; ++c) {
      // chunk c has landed when at most the instructions of the younger chunks in flight are outstanding
      static_assert((DEPTH == 3 || DEPTH == 2) && 2 * Gm::PER_WAVE < 64, "the waits below");
      switch (min(chunks - 1 - c, DEPTH - 1)) {                  // (wave-uniform)
        case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Gm::PER_WAVE) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * Gm::PER_WAVE) : "memory"); break;
      }
      lds_barrier_raw();
      if (c + DEPTH < chunks) issue(c + DEPTH);                  // into the buffer of chunk c - 1: every compute wave is past it
    }
    lds_barrier_raw();                                            // (the compute waves' "chunk buffers are dead" barrier)
  } else {
    // ------------------------------------------------------------------ compute waves
    // bias partials of a split segment: loaded now, folded after the loop (their latency rides on the first chunks)
    constexpr int MAXM = 8;
    float bp[MAXM];
    const bool folds = a.M > 1 && (int)threadIdx.x < HC;
    const int64_t boff = po + (CONV == 0 ? L.c1_b : L.c2_b) + threadIdx.x;
#pragma unroll
    for (int p = 0; p < MAXM; ++p)
      bp[p] = (folds && p < a.M) ? a.part_slabs[((int64_t)seg * a.M + p) * L.slab_stride + boff] : 0.f;
    const int i = lane & 15, q = lane >> 4;
    f32x4 acc[VC][VK], accT[VK];
#pragma unroll
    for (int y = 0; y < VK; ++y) {
      accT[y] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int x = 0; x < VC; ++x) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    ISTAMP(1);
    for (int c = 0; c < chunks; ++c) {
      lds_barrier_raw();
      if (c == 0) ISTAMP(2);
      if (c == 1) ISTAMP(3);
      if (c == chunks - 1) ISTAMP(4);
      const float* B = bufp(c);
      const float* Sc = B + CR * (HC + K);                        // [CR][H] g_a_src, then [CR][H] g_a_dst
      const int r0 = c * CR;
      for (int st = wave; st < CR / 4; st += CW) {                // (wave-uniform)
        const int r = 4 * st + q;
        float av[VC], bv[VK];
        load_frag<VC>(B + r * HC + VC * i, av);
        load_frag<VK>(B + CR * HC + r * K + VK * i, bv);
        // the extra tile row: column i < H is g_a_src[r, i], H <= i < 2H is g_a_dst[r, i - H]
        float ae = i < 2 * H ? Sc[(i < H ? 0 : CR * H) + r * H + (i < H ? i : i - H)] : 0.f;
        if (r0 + r >= n) {
          ae = 0.f;
#pragma unroll
          for (int x = 0; x < VC; ++x) av[x] = 0.f;
        }
#pragma unroll
        for (int y = 0; y < VK; ++y) {
          accT[y] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, bv[y], accT[y], 0, 0, 0);
#pragma unroll
          for (int x = 0; x < VC; ++x) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], bv[y], acc[x][y], 0, 0, 0);
        }
      }
    }
    ISTAMP(5);
    lds_barrier_raw();                                            // every chunk buffer is dead: the partial blocks meet in them
    float* mine = lds + wave * (HC * K);
#pragma unroll
    for (int x = 0; x < VC; ++x)
#pragma unroll
      for (int y = 0; y < VK; ++y)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mine[(VC * (4 * q + rr) + x) * K + VK * i + y] = acc[x][y][rr];
    if (q == 0) {                                                 // T rows 0 .. 3 live in the q = 0 lanes
#pragma unroll
      for (int y = 0; y < VK; ++y)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) tred[(wave * 4 + rr) * K + VK * i + y] = accT[y][rr];
    }
    if (folds) {
      float sum = 0.f;
#pragma unroll
      for (int p = 0; p < MAXM; ++p)
        if (p < a.M) sum += bp[p];
      slab_st<AG>(a.slabs + (int64_t)seg * L.slab_stride + boff, sum);
    }
  }
  lds_barrier_raw();
  ISTAMP(6);
  for (int idx = threadIdx.x; idx < HC * K; idx += THREADS) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < CW; ++w) sum += lds[w * (HC * K) + idx];
    slab_st<AG>(sb + (CONV == 0 ? L.c1_W : L.c2_W) + idx, sum);
  }
  if ((int)threadIdx.x < 4 * K) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < CW; ++w) sum += tred[w * 4 * K + threadIdx.x];
    tsum[threadIdx.x] = sum;
  }
  lds_barrier_raw();
  {
    // g_att[c] = sum_k W^T[k][c] T[hd(c)][k]: W^T [K][HC] is the scratch copy the dX stages use (L2-resident).  Thread
    // (c, kg) takes k = kg, kg + KG, ... (coalesced over c), the KG partial sums meet in LDS and are added in order.
    constexpr int KG = THREADS / HC, KPT = (K + KG - 1) / KG;
    const int c = threadIdx.x % HC, kg = threadIdx.x / HC, hd = c / (HC / H);
    const float* Wt = a.wt + ((int64_t)b * 2 + CONV) * (2LL * NC * NC) + c;
    float wv[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) wv[j] = (kg + j * KG < K) ? Wt[(kg + j * KG) * HC] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = min(kg + j * KG, K - 1);
      s0 = fmaf(wv[j], tsum[hd * K + k], s0);
      s1 = fmaf(wv[j], tsum[(H + hd) * K + k], s1);
    }
    float* ared = lds;                                             // (the dW partial blocks are folded: reuse)  [2][KG][HC]
    ared[kg * HC + c] = s0; ared[(KG + kg) * HC + c] = s1;
    lds_barrier_raw();
    if ((int)threadIdx.x < HC) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll 4
      for (int g = 0; g < KG; ++g) { t0 += ared[g * HC + c]; t1 += ared[(KG + g) * HC + c]; }
      slab_st<AG>(sb + (CONV == 0 ? L.c1_as : L.c2_as) + c, t0);
      slab_st<AG>(sb + (CONV == 0 ? L.c1_ad : L.c2_ad) + c, t1);
    }
  }
  ISTAMP(7);
#undef ISTAMP
}

// the second-launch form of the streamed items (8 parts per snapshot leave no CU for consumers): one item per workgroup,
// 512 threads (4 DMA + 4 compute waves) and three chunk buffers (77 KB): TWO workgroups per CU, so that a CU keeps
// streaming while one of its items is in its epilogue (one 1024-thread workgroup per CU streamed 64 % of the time)
constexpr int PGS_THREADS = 512, PGS_NBUF = 3;
template <int NC>
__global__ __launch_bounds__(PGS_THREADS) void param_grads_stream_kernel(const ParamGradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PGS_NBUF * CI_CR * (3 * NC + 4)];
  // Item of this workgroup.  The kept g_h tables an item reads were written by the window kernel's parts of its segment -- on
  // XCD (segment % 8), their ids being 8 apart -- and the LAST blocks the backward chain walked (b = 0, 1, ...) were written
  // last: with a whole number of segments per XCD the item goes to a workgroup of THAT XCD (round-robin dispatch: id % 8) and
  // the items are dealt block-major from b = 0, so the first wave of items finds its freshest tables still in its XCD's L2
  // (4 MB hold the kept tables of four to five blocks of an XCD's four snapshots).  Speed only: any map is a bijection.
  int conv = blockIdx.x & 1, b = (blockIdx.x >> 1) % a.L.nb, seg = (blockIdx.x >> 1) / a.L.nb;
#ifndef GATRES_PROBE_PG_OLD_MAP
  {
    const int S = (int)(gridDim.x / (2 * a.L.nb));
    if ((S & 7) == 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (S >> 3) * 2;       // (segment, conv) pairs of one XCD
      b = j / per; conv = (j % per) & 1; seg = ((j % per) >> 1) * 8 + xcd;
    }
  }
#endif
  if (a.snap && blockIdx.x == 0 && threadIdx.x == 0) {
    a.snap[0] = a.step_counter ? __hip_atomic_load(a.step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
    a.snap[1] = a.status ? (unsigned long long)__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
  }
  if (conv == 0) consumer_item_dma<NC, PGS_THREADS, 0, PGS_NBUF>(a, seg, b, lds);
  else           consumer_item_dma<NC, PGS_THREADS, 1, PGS_NBUF>(a, seg, b, lds);
  if (a.M > 1 && b == 0 && conv == 0) {
    fold_parts<PGS_THREADS>(a, seg, a.L.p_lin0_w, 2 * NC);
    fold_parts<PGS_THREADS>(a, seg, a.L.p_lin1_w, NC + 1);
  }
}

// ---- items streamed through REGISTERS (round 5; the stand-alone launch for nc 16 / 32).  The LDS-DMA form above keeps two
// 64-row chunks in flight per workgroup and paces its waves by one barrier per chunk; its launch moves 161 MB at 4.2 TB/s with the
// matrix pipe 42 % busy -- neither bound.  Here an item is one 256-thread workgroup whose four waves take the item's 4-row steps
// round-robin (wave w: steps w, w + 4, ...), every lane loads its MFMA operands straight from the tables (the permuted operand map
// of seg_dw_blk: 16 / 8 bytes of contiguous features per lane, a wave-load covers four whole rows) into a ring of P register
// slots -- P steps in flight per wave, no barrier, no LDS until the four partial blocks meet -- and ALL 960 items of a bs-32 step
// are resident at once (35 KB of LDS and 16 waves for the four workgroups of a CU).  T = [g_a_src | g_a_dst]^T x (the attention-
// vector gradients without reading h, see above) is formed on the VALU beside the matrix pipe: 2H x VK fused multiply-adds per lane
// and step, the four row phases of a wave summed by two shuffles at the end.  Every sum has a fixed order: deterministic.
constexpr int PGR_THREADS = 256;
#define PGR_LDS_FLOATS(NC) ((PGR_THREADS / 64) * 2 * (NC) * (NC) + (PGR_THREADS / 64 + 1) * 4 * (NC))
#ifndef PGR_SLOTS
#define PGR_SLOTS 4          // steps in flight per wave (3 .. 6 measure the same; 7 and non-temporal loads are slower)
#endif
// a lane's operand fragment of W consecutive floats as ONE register tuple; requested by hand-written global_load (the compiler
// does not count these: pgr_wait names the tuples it makes valid, so no use can move in front of the wait)
template <int W> struct PgrVec;
template <> struct PgrVec<1> { typedef float type; };
template <> struct PgrVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct PgrVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int W>
__device__ __forceinline__ void pgr_load(typename PgrVec<W>::type& d, const float* p) {
#ifdef PGR_PROBE_NOLOAD
  asm volatile("" : "=v"(d) : "v"(p));
  return;
#endif
#ifdef PGR_NT
#define PGR_MOD " nt"
#else
#define PGR_MOD ""
#endif
  // "+v": the request lands in the registers the slot already occupies -- the ring stays in place, the compiler cannot rename
  // a slot between its request and its wait, and the tuple is never a free register while a load is in flight
  if constexpr (W == 4)      asm volatile("global_load_dwordx4 %0, %1, off" PGR_MOD : "+v"(d) : "v"(p) : "memory");
  else if constexpr (W == 2) asm volatile("global_load_dwordx2 %0, %1, off" PGR_MOD : "+v"(d) : "v"(p) : "memory");
  else                       asm volatile("global_load_dword %0, %1, off" PGR_MOD : "+v"(d) : "v"(p) : "memory");
}
template <int W>
__device__ __forceinline__ float pgr_at(const typename PgrVec<W>::type& v, int k) {
  if constexpr (W == 1) return v;
  else return v[k];
}
template <int N, class A, class B, class C, class D>
__device__ __forceinline__ void pgr_wait(A& a, B& b, C& c, D& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
// (AG: slab entries by agent-scope stores, for param_grads_finish_kernel's last arriver)
template <int NC, int CONV, bool AG = false>
__device__ __forceinline__ void param_grads_item_reg(const ParamGradArgs& a, int seg, int b, float* lds) {
  using Gm = CiGeom<NC, CONV>;
  constexpr int HC = Gm::HC, K = Gm::K, H = Gm::H, P = PGR_SLOTS;
  constexpr int NW = PGR_THREADS / 64, VC = HC / 16, VK = K / 16;
  static_assert(4 * (P - 1) < 64, "vmcnt");
  const Layout& L = a.L;
  const SegLayout& SL = a.SL;
  const int n0 = uni(a.seg_ptr[seg]), n = uni(a.seg_ptr[seg + 1]) - n0;
  const float* base = a.saved + (int64_t)seg * SL.total + (int64_t)b * SL.bstride;
  const float* keep = a.keep + (int64_t)b * L.keep_stride;
  const int64_t po = L.p_block0 + (int64_t)b * L.p_block_stride;
  float* sb = a.slabs + (int64_t)seg * L.slab_stride + po;
  const float* gG = keep + (CONV == 0 ? L.k_gh1 : L.k_gh2) + (int64_t)n0 * HC;      // [n][HC]
  const float* gX = base + (CONV == 0 ? SL.xin : SL.o1);                              // [n][K]
  const float* gS = keep + (CONV == 0 ? L.k_gas1 : L.k_gas2) + (int64_t)n0 * H;      // [n][H]
  const float* gD = keep + (CONV == 0 ? L.k_gad1 : L.k_gad2) + (int64_t)n0 * H;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int steps = (n + 3) >> 2;
  const int cnt = steps > wave ? (steps - wave + NW - 1) / NW : 0;                    // this wave's steps (wave-uniform)
  float* part = lds;                                  // [NW][HC * K]
  float* tpart = lds + NW * HC * K;                   // [NW][2H][K]
  float* tsum = tpart + NW * 2 * H * K;               // [2H][K]

  f32x4 acc[VC][VK];
  float accT[2 * H][VK];
#pragma unroll
  for (int y = 0; y < VK; ++y) {
#pragma unroll
    for (int x = 0; x < VC; ++x) acc[x][y] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2 * H; ++j) accT[j][y] = 0.f;
  }
  typename PgrVec<VC>::type gv[P] = {};
  typename PgrVec<VK>::type xv[P] = {};
  typename PgrVec<H>::type sv[P] = {}, dv[P] = {};
  // step t of this wave = rows 4 (wave + t NW) + q; a row beyond the segment reads the last row and counts as zero.  EVERY slot
  // is re-requested unconditionally, four instructions each (steps beyond the wave's last re-read its last step: cache hits),
  // so "slot j has landed" is always "at most the 4 (P - 1) younger requests are outstanding"
  const int last = max(cnt - 1, 0);
#define PGR_LOAD(slot, t)                                                                        \
  do {                                                                                           \
    const int r_ = max(min(4 * (wave + min((t), last) * NW) + q, n - 1), 0);                     \
    pgr_load<VC>(gv[slot], gG + (unsigned)(r_ * HC + VC * i));                                   \
    pgr_load<VK>(xv[slot], gX + (unsigned)(r_ * K + VK * i));                                    \
    pgr_load<H>(sv[slot], gS + (unsigned)(r_ * H));                                              \
    pgr_load<H>(dv[slot], gD + (unsigned)(r_ * H));                                              \
  } while (0)
#pragma unroll
  for (int j = 0; j < P; ++j) PGR_LOAD(j, j);
  for (int t0 = 0; t0 < cnt; t0 += P) {
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int t = t0 + j;
      pgr_wait<4 * (P - 1)>(gv[j], xv[j], sv[j], dv[j]);
#ifdef PGR_PROBE_NOMFMA
      if (t < cnt && n < 0) {
#else
      if (t < cnt) {                                                                    // (wave-uniform)
#endif
        const bool ok = 4 * (wave + t * NW) + q < n;
        float ae[2 * H], af[VC], bf[VK];
#pragma unroll
        for (int x = 0; x < VC; ++x) af[x] = ok ? pgr_at<VC>(gv[j], x) : 0.f;
#pragma unroll
        for (int y = 0; y < VK; ++y) bf[y] = pgr_at<VK>(xv[j], y);
#pragma unroll
        for (int k = 0; k < H; ++k) { ae[k] = ok ? pgr_at<H>(sv[j], k) : 0.f; ae[H + k] = ok ? pgr_at<H>(dv[j], k) : 0.f; }
#pragma unroll
        for (int y = 0; y < VK; ++y) {
#pragma unroll
          for (int x = 0; x < VC; ++x) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[x], bf[y], acc[x][y], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 2 * H; ++k) accT[k][y] = fmaf(ae[k], bf[y], accT[k][y]);
        }
      }
      PGR_LOAD(j, t + P);
    }
  }
#undef PGR_LOAD
  // the re-requests behind the last step: nothing reads them, but their registers stay named until they have landed (to the
  // compiler an unused asm output is a free register at once)
#pragma unroll
  for (int j = 0; j < P; ++j) pgr_wait<0>(gv[j], xv[j], sv[j], dv[j]);
#ifdef PGR_PROBE_NOEPI
  if (n > 0) { if (acc[0][0][0] == 12345.f && accT[0][0] == 1.f) sb[0] = 1.f; return; }
#endif
  // operands of the epilogue, requested before the partial blocks meet: the bias partials of a split segment and this thread's
  // share of W^T (the scratch copy the dX stages use, L2-resident)
  constexpr int MAXM = 8;
  float bp[MAXM];
  const bool folds = a.M > 1 && (int)threadIdx.x < HC;
  const int64_t boff = po + (CONV == 0 ? L.c1_b : L.c2_b) + threadIdx.x;
#pragma unroll
  for (int p = 0; p < MAXM; ++p)
    bp[p] = (folds && p < a.M) ? a.part_slabs[((int64_t)seg * a.M + p) * L.slab_stride + boff] : 0.f;
  // g_att[c] = sum_k W^T[k][c] T[hd(c)][k]: wave w takes the columns [CPW w, CPW (w + 1)), lane (cl, kg) the k = kg, kg + KGW, ...
  // of column CPW w + cl; the KGW partial sums of a column meet by shuffles inside the wave (no LDS, no barrier)
  constexpr int CPW = HC / NW, KGW = 64 / CPW, KPT = K / KGW;
  static_assert(HC % NW == 0 && 64 % CPW == 0 && K % KGW == 0 && CPW >= 1, "attention-vector columns per wave");
  const int c = CPW * wave + lane % CPW, kg = lane / CPW, hd = c / (HC / H);
  float wv[KPT];
  {
    const float* Wt = a.wt + ((int64_t)b * 2 + CONV) * (2LL * NC * NC) + c;
#pragma unroll
    for (int j = 0; j < KPT; ++j) wv[j] = Wt[(kg + j * KGW) * HC];
  }
  float* mine = part + wave * (HC * K);
#pragma unroll
  for (int x = 0; x < VC; ++x)
#pragma unroll
    for (int y = 0; y < VK; ++y)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) mine[(VC * (4 * q + rr) + x) * K + VK * i + y] = acc[x][y][rr];
#pragma unroll
  for (int k = 0; k < 2 * H; ++k)
#pragma unroll
    for (int y = 0; y < VK; ++y) {
      float v = accT[k][y];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q == 0) tpart[(wave * 2 * H + k) * K + VK * i + y] = v;
    }
  if (folds) {
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < MAXM; ++p)
      if (p < a.M) sum += bp[p];
    slab_st<AG>(a.slabs + (int64_t)seg * L.slab_stride + boff, sum);
  }
  lds_barrier_raw();
  static_assert(HC * K % (4 * PGR_THREADS) == 0 || HC * K == 2 * PGR_THREADS, "fold: 16 bytes per thread and trip");
  if constexpr (HC * K % (4 * PGR_THREADS) == 0) {
#pragma unroll
    for (int idx = 4 * (int)threadIdx.x; idx < HC * K; idx += 4 * PGR_THREADS) {
      float4 sum = ld4(part + idx);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 v = ld4(part + w * (HC * K) + idx);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      float* dst = sb + (CONV == 0 ? L.c1_W : L.c2_W) + idx;
      if constexpr (AG) { slab_st<AG>(dst, sum.x); slab_st<AG>(dst + 1, sum.y); slab_st<AG>(dst + 2, sum.z); slab_st<AG>(dst + 3, sum.w); }
      else st4(dst, sum);
    }
  } else {
    for (int idx = threadIdx.x; idx < HC * K; idx += PGR_THREADS) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += part[w * (HC * K) + idx];
      slab_st<AG>(sb + (CONV == 0 ? L.c1_W : L.c2_W) + idx, sum);
    }
  }
  if ((int)threadIdx.x < 2 * H * K) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += tpart[w * 2 * H * K + threadIdx.x];
    tsum[threadIdx.x] = sum;
  }
  lds_barrier_raw();
  {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int k = kg + j * KGW;
      s0 = fmaf(wv[j], tsum[hd * K + k], s0);
      s1 = fmaf(wv[j], tsum[(H + hd) * K + k], s1);
    }
#pragma unroll
    for (int off = CPW; off < 64; off <<= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if (kg == 0) {
      slab_st<AG>(sb + (CONV == 0 ? L.c1_as : L.c2_as) + c, s0);
      slab_st<AG>(sb + (CONV == 0 ? L.c1_ad : L.c2_ad) + c, s1);
    }
  }
}

template <int NC>
__global__ __launch_bounds__(PGR_THREADS) void param_grads_reg_kernel(const ParamGradArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[PGR_LDS_FLOATS(NC)];
  // item of this workgroup: on the XCD whose window-kernel parts wrote its kept tables (param_grads_stream_kernel's map)
  int conv = blockIdx.x & 1, b = (blockIdx.x >> 1) % a.L.nb, seg = (blockIdx.x >> 1) / a.L.nb;
  {
    const int S = (int)(gridDim.x / (2 * a.L.nb));
    if ((S & 7) == 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (S >> 3) * 2;       // (segment, conv) pairs of one XCD
      b = j / per; conv = (j % per) & 1; seg = ((j % per) >> 1) * 8 + xcd;
    }
  }
  if (a.snap && blockIdx.x == 0 && threadIdx.x == 0) {
    a.snap[0] = a.step_counter ? __hip_atomic_load(a.step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
    a.snap[1] = a.status ? (unsigned long long)__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL;
  }
#ifdef PGR_PROBE_EMPTY
  if (a.M > 0) return;
#endif
  if (conv == 0) param_grads_item_reg<NC, 0>(a, seg, b, lds);
  else           param_grads_item_reg<NC, 1>(a, seg, b, lds);
  if (a.M > 1 && b == 0 && conv == 0) {
    fold_parts<PGR_THREADS>(a, seg, a.L.p_lin0_w, 2 * NC);
    fold_parts<PGR_THREADS>(a, seg, a.L.p_lin1_w, NC + 1);
  }
}

// ------------------------------------------------------------------------------------------ split segments
// One segment may be carried by M workgroups on M CUs (M a power of two, chosen by the host so that every workgroup
// of the grid is resident at once).  Part p owns a 16-aligned row window; the dense stages touch own rows only, the
// sparse stages gather from neighbours, so at the six points per block where a stage reads rows that a partner
// produced the parts meet in a flag barrier through global memory (~0.7 us on MI355X, tests/micro/xcu_sync.hip) and
// then complete their LDS tables from the partners' global copies.  Workgroup ids of one segment differ by multiples
// of 8: consecutive ids are dispatched round-robin over the 8 XCDs, so partners share an XCD and its L2.
constexpr int FLAG_STRIDE = 32;                 // one 128-byte line per flag
// buffer_inv completes asynchronously; the s_waitcnt after it keeps the issuing wave (and, through the barrier that
// follows, the workgroup) from loading a partner's lines before this CU's stale L1 copies are gone
// (cdna_hip_programming.md Guideline 16: fence, that lane's vmcnt(0), barrier, then the other waves' loads).
#ifndef GATRES_NO_INV_WAIT
#define INV_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define INV_WAIT() do {} while (0)
#endif

struct Group {
  unsigned* flags;        // this segment's M flag lines: word 0 = epoch, word 1 = the XCD the part runs on.  Epochs
                          // keep counting across launches (all parts of a segment pass the same number of barriers
                          // per launch, so they stay equal between launches; the scratch buffer starts zeroed)
  int M, part;
  unsigned epoch;
  int* err;
  bool dead;              // (per thread) this poller gave up once: never spin again
  bool local;             // all parts run on one XCD (one L2): no L2 write-back / invalidate needed
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

// Barrier over the M workgroups of a segment, making every global store issued before it visible to every part
// after it.  Measured in tests/micro/xcu_sync.hip: with agent-scope release/acquire (L2 write-back + invalidate; the
// only correct form when parts sit on different XCDs) every barrier costs microseconds once all 256 CUs do it; parts on
// the SAME XCD share the L2, so after the workgroup barrier (s_waitcnt vmcnt(0): the stores are in L2) a relaxed
// flag store, relaxed polling and one "buffer_inv sc1" (drops the CU's stale L1 lines; "sc0" does not) suffice.
template <int THREADS>
__device__ __forceinline__ void group_sync(Group& g) {
  // EVERY wave drains its own global stores (and LDS-DMA) before the workgroup barrier: s_barrier waits for no counter,
  // and hipcc's workgroup-scope release before it only waits lgkmcnt -- without this wait the flag below could overtake
  // another wave's halo / table stores (MI355X_MICROARCH.md, Valid forms: "every storing wave's s_waitcnt vmcnt(0)")
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (g.M == 1) return;
  ++g.epoch;
  if (threadIdx.x == 0) {
    if (!g.local) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the fence's own wait)
    }
    __hip_atomic_store(g.flags + g.part * FLAG_STRIDE, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < 64) {                        // wave 0: lanes 0..M-1 poll one partner each
    if ((int)threadIdx.x < g.M && (int)threadIdx.x != g.part && !g.dead) {
      int spin = 0;
      while ((int)(__hip_atomic_load(g.flags + threadIdx.x * FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                   g.epoch) < 0)
        if (++spin > SPIN_LIMIT) { *g.err = 1; g.dead = true; break; }
    }
    if (g.local) asm volatile("buffer_inv sc1" ::: "memory");
    else         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    INV_WAIT();                                  // hold the barrier until the invalidate has completed
  }
  __syncthreads();
}

// First barrier of a launch: publishes the XCD ids (always with the safe agent-scope form) and decides `local`.
// assume_local (the host probed the device's dispatch: gatres_probe_xcd_dispatch): the parts take their shared XCD for granted
// and do NOT meet here -- the barrier made every part wait until the last one of its segment had been dispatched and had
// crossed an agent-scope flag round trip: 7 us per launch.  The ids are still recorded; group_verify_local compares them at
// the end of the launch.  (Wrong placement cannot go unnoticed earlier either: a granule that never arrives in this part's L2
// ends its sweep with the fault word set.)
template <int THREADS>
__device__ __forceinline__ void group_init(Group& g, bool assume_local = false) {
  g.dead = false; g.local = false;
  if (g.M == 1) { g.epoch = 0u; return; }
  g.epoch = __hip_atomic_load(g.flags + g.part * FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0)
    __hip_atomic_store(g.flags + g.part * FLAG_STRIDE + 1, xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (assume_local) { g.local = true; return; }
  group_sync<THREADS>(g);
  bool same = true;
  const unsigned mine = xcc_id();
  for (int p = 0; p < g.M; ++p)
    same = same && (__hip_atomic_load(g.flags + p * FLAG_STRIDE + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine);
  g.local = same;
}

// End of a launch that started with assume_local: every partner this part exchanged rows with has recorded its XCD id long ago.
// (lanes 0 .. M-1 of wave 0, one partner each: ONE L2 round trip, not M dependent ones)
__device__ __forceinline__ void group_verify_local(const Group& g) {
  if (g.M <= 1 || threadIdx.x >= 64) return;
  const unsigned mine = xcc_id();
  unsigned v = mine;
  if ((int)threadIdx.x < g.M)
    v = __hip_atomic_load(g.flags + threadIdx.x * FLAG_STRIDE + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__any(v != mine) && threadIdx.x == 0) *g.err = 1;
}

// dst[k] = src[k] for k in [0, a) and [b, total): the partners' part of a table whose own part is [a, b).
template <int THREADS>
__device__ __forceinline__ void pull_flat(float* dst, const float* src, int a, int b, int total) {
  const int cnt = a + (total - b);
  for (int k = threadIdx.x; k < cnt; k += THREADS) {
    const int j = k < a ? k : k - a + b;
    dst[j] = src[j];
  }
}
// same for rows of W floats (W % 4 == 0, 16-byte aligned tables), float4 at a time
template <int THREADS>
__device__ __forceinline__ void pull_rows4(float* dst, const float* src, int W, Rows rw, int n) {
  const int a = rw.lo * (W / 4), b = rw.hi * (W / 4), total = n * (W / 4);
  const int cnt = a + (total - b);
  for (int k = threadIdx.x; k < cnt; k += THREADS) {
    const int j = k < a ? k : k - a + b;
    st4(dst + 4 * j, ld4(src + 4 * j));
  }
}

// Halo lists.  A part only ever gathers partner rows that are NEIGHBOURS of its own rows, and the topology is the
// same for every block, so each phase builds the list once in spare LDS: the forward list holds the remote sources
// of own in-edges, the backward list the remote destinations (and edge ids) of own out-edges.  Duplicates are kept
// (copies are idempotent).  With a locality-preserving node order (water networks are near-planar) the halo is a few
// percent of the segment; if the list overflows the spare LDS the bulk pulls above are used instead.
template <int THREADS>
__device__ __forceinline__ int build_halo(const u16* ptr, const u16* idx, const u16* eid, Rows rw, u16* list,
                                          u16* elist, int cap, int* counter) {
  if (threadIdx.x == 0) *counter = 0;
  __syncthreads();
  const int beg = ptr[rw.lo], end = ptr[rw.hi];
  for (int t = beg + threadIdx.x; t < end; t += THREADS) {
    const int j = idx[t];
    if (j < rw.lo || j >= rw.hi) {
      const int pos = atomicAdd(counter, 1);
      if (pos < cap) {
        list[pos] = (u16)j;
        if (elist) elist[pos] = eid[t];
      }
    }
  }
  __syncthreads();
  const int total = *counter;
  __syncthreads();               // (the caller may reuse the counter at once)
  return total;
}
template <int W, int THREADS>
__device__ __forceinline__ void pull_list_rows(float* dst, const float* src, const u16* list, int cnt) {
  constexpr int G = W / 4;
  const int c0 = (threadIdx.x % G) * 4;
  for (int k = threadIdx.x / G; k < cnt; k += THREADS / G) {
    const int j = list[k];
    st4(dst + j * W + c0, ld4(src + (unsigned)(j * W + c0)));
  }
}
template <int H, int THREADS>
__device__ __forceinline__ void pull_list_small(float* dst, const float* src, const u16* list, int cnt) {
  for (int k = threadIdx.x; k < cnt * H; k += THREADS) {
    const int j = list[k / H] * H + k % H;
    dst[j] = src[j];
  }
}

// ------------------------------------------------------------------------------------------ granule exchange
// Window kernel: the parts of a split segment hand each other halo rows as tagged 8-byte GRANULES {value, epoch}
// (cdna_hip_programming.md Guideline 16, form R2: "the data IS the flag").  The producer stores a granule with one
// relaxed agent-scope 8-byte store (sc1: write-through, visible wherever the consumer runs); the consumer re-reads its
// granules with relaxed agent-scope loads (sc1: never served from its L1) until every tag equals the epoch of THIS
// exchange.  No flag, no fence, no cache invalidate, no store drain in front of the hand-off: one memory round trip
// instead of the drain -> flag -> poll -> invalidate -> pull chain of group_sync (2.2 us -> ~1 us per exchange, six
// exchanges per block).  Placement-independent: nothing relies on the parts sharing an XCD.
//   * one table per exchange point (XchLayout), indexed by the segment's local row / edge id, so a cell is rewritten only
//     six exchanges later; the HEARTBEAT granules (one per point and part, swept by every part at every exchange) keep
//     the parts within one exchange of each other, so a cell is never rewritten before its readers are done, whatever
//     the graph looks like (a directed edge list may make the data dependencies one-sided);
//   * epochs count exchanges and persist in the segment's flag lines across launches (word 2 of the part's line); every
//     part passes the same number of exchanges per launch;
//   * a producer that never delivers ends the sweep after SPIN_LIMIT rounds with the error word set (results poisoned,
//     the step dropped by gatres_fused_finish), never a hang.
// table[(idx)*W + c] of every listed row / edge -> granules.  src: LDS table addressed by the same index.
template <int W, int THREADS>
__device__ __forceinline__ void xch_export(const Xch& x, const u16* list, int cnt, const float* src, u64* dst) {
  for (int k = threadIdx.x; k < cnt * W; k += THREADS) {
    const int o = (int)list[k / W] * W + (k % W);
    gran_store(dst + o, src[o], x.ep, x.local);
  }
}

// granules of every listed row / edge -> table, re-read until their tags carry this exchange's epoch
// (w0: the sweep is carried by the waves from w0 up -- the ones below are busy with the softmax of the same exchange)
template <int W, int THREADS>
__device__ __forceinline__ void xch_import(Xch& x, const u16* list, int cnt, u64* src, float* dst, int w0 = 0) {
  constexpr int U = 2;
  const int total = cnt * W;
  const int t = (int)threadIdx.x - 64 * w0, PT = THREADS - 64 * w0;
  if (t < 0) return;                                                     // (wave-uniform)
  for (int base = 0; base < total; base += U * PT) {
    int o[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = base + u * PT + t;
      valid[u] = k < total;
      o[u] = valid[u] ? (int)list[k / W] * W + (k % W) : 0;
    }
    if (!__any(valid[0])) continue;                                   // (wave-uniform)
    u64 v[U];
    int spin = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = valid[u] ? gran_load(src + o[u]) : ((u64)x.ep << 32);
#pragma unroll
      for (int u = 0; u < U; ++u) ok = ok && (unsigned)(v[u] >> 32) == x.ep;
      if (__all(ok || x.dead)) break;
      if (++spin > SPIN_LIMIT) { *x.err = 1; x.dead = true; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (valid[u]) dst[o[u]] = __uint_as_float((unsigned)v[u]);
  }
}

// Every exchange, after the part's own sweep: announce "exchange ep is behind me" (one heartbeat granule per part, its
// tag only ever grows) and make sure every other part has announced exchange ep - 1.  That is all the pacing the tables
// need: a table is rewritten three exchanges after it was read (the same point of the next block), the writer has seen
// the reader's announcement of the exchange in between, and the reader announced that one after the sweep in question.
// Nobody waits for the slowest part of THIS exchange (round 2 until here: all parts met at every exchange, 0.5 - 1 us
// each, profiles/r02_exchange_steps.txt) -- a late part only delays the neighbours whose halo rows it owes.
// Then EVERY wave drains its own outstanding global stores -- the sweeping waves did so while waiting for their loads,
// the others wait here, beside them -- so that after the barrier that follows the exchange all stores issued before it
// (saved activations, kept gradient tables) are complete: publish_items relies on that.
template <int THREADS>
__device__ __forceinline__ void xch_heartbeat(Xch& x, int /*point*/) {
  u64* hb = x.base_hb;
  if (threadIdx.x == 0) gran_store(hb + x.part, 0.f, x.ep, x.local);
  const int lane = (int)threadIdx.x - (THREADS - 64);                 // the last wave polls: lanes 0 .. M-1, one part each
  if (lane >= 0) {
    const bool mine = lane < x.M && lane != x.part;
    int spin = 0;
    for (;;) {
      const bool ok = !mine || x.dead || (int)((unsigned)(gran_load(hb + lane) >> 32) - (x.ep - 1u)) >= 0;
      if (__all(ok)) break;
      if (++spin > SPIN_LIMIT) { *x.err = 1; x.dead = true; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// own rows with an out-edge to (FWD: a destination) / an in-edge from (BWD: a source) outside [lo, hi): the rows whose
// values some partner's halo list names.  ptr / idx: global CSR arrays of the relevant direction; no duplicates.
template <int THREADS>
__device__ __forceinline__ int build_export_rows(const int* __restrict__ ptr, const int* __restrict__ idx, int n0, Rows rw,
                                                 u16* list, int cap, int* counter) {
  if (threadIdx.x == 0) *counter = 0;
  __syncthreads();
  for (int r = rw.lo + threadIdx.x; r < rw.hi; r += THREADS) {
    bool remote = false;
    for (int t = ptr[n0 + r]; t < ptr[n0 + r + 1]; ++t) {
      const int j = idx[t] - n0;
      remote = remote || j < rw.lo || j >= rw.hi;
    }
    if (remote) {
      const int pos = atomicAdd(counter, 1);
      if (pos < cap) list[pos] = (u16)r;
    }
  }
  __syncthreads();
  const int total = *counter;
  __syncthreads();
  return total;
}

// Deferred-gradient items of a segment, in production order: item 2*(nb-1-b) is (block b, conv2), the next one
// (block b, conv1); a split segment adds one last item that folds the lin0 / lin1 partials.  Part 0 publishes the
// number of finished items after a barrier that all parts have passed; consumer c takes items c, c + C, ...
// The published word also says where the data is: bits 16..19 the XCD of part 0, bit 24 "every part runs on that
// XCD".  A consumer on the same XCD reads the tables through the shared L2 as soon as they are published; the LAST
// publication is preceded by an agent-scope release, so a consumer that finds itself anywhere else waits for it,
// acquires at agent scope and only then works through its items (correct wherever the workgroups were placed).
constexpr unsigned PUB_COUNT_MASK = 0xffffu, PUB_LOCAL = 1u << 24;
// EVERY part publishes its own progress (word `part` of each consumer's line): the tables of an item are complete when all
// M parts have published its number.  Called right after a point where every wave of the workgroup has drained its
// global stores (group_sync, or the granule import of the window kernel) and a workgroup barrier has been passed.
template <int THREADS>
__device__ __forceinline__ void publish_items(const FusedArgs& a, int seg, int part, int count, bool parts_local,
                                              bool last) {
  if (a.C > 0 && (int)threadIdx.x < 64) {
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if ((int)threadIdx.x < a.C)
      __hip_atomic_store(a.ready + ((size_t)seg * 4 + threadIdx.x) * FLAG_STRIDE + part,
                         (unsigned)count | (xcc_id() << 16) | (parts_local ? PUB_LOCAL : 0u), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NC, int THREADS>
__device__ __forceinline__ void consumer_main(const FusedArgs& a, int cid, float* ldsf) {
  const int C = a.C;
  const int within = cid % (8 * C);
  const int seg = a.seg0 + (cid / (8 * C)) * 8 + (within & 7), c = within >> 3;
  if (seg >= a.seg0 + a.seg_cnt) return;
  ParamGradArgs pg;
  pg.seg_ptr = a.seg_ptr; pg.saved = a.saved; pg.keep = a.scratch + a.L.sc_keep; pg.slabs = a.slabs;
  pg.part_slabs = a.part_slabs; pg.M = a.M; pg.L = a.L; pg.SL = a.SL;
  pg.wt = a.wt;
  pg.dstamps = (STAMPS_PTR && cid == 0 && a.stamp_cap >= 4096) ? STAMPS_PTR + 3072 : nullptr;
  unsigned* my = a.ready + ((size_t)seg * 4 + c) * FLAG_STRIDE;       // words 0 .. M-1: one per producing part
  float* part = ldsf;
  float* red = ldsf + (LDS_BYTES / 4 - 3 * THREADS);
  int* mode = reinterpret_cast<int*>(red + 3 * THREADS - 1);   // LDS word: 0 = hand-off through the shared L2, 1 = remote
  const int nb = a.L.nb, items = 2 * nb + (a.M > 1 ? 1 : 0);
  bool dead = false, remote = false;
  const int lane = threadIdx.x;
  const unsigned mine_xcc = xcc_id();
  int cst = 1024;            // diagnostic stamps of consumer 0 (gatres_fused_set_stamps): [item available, item done] pairs
#define CSTAMP()                                                                                    \
  do {                                                                                              \
    if (STAMPS_PTR && cid == 0 && threadIdx.x == 0 && cst < a.stamp_cap) STAMPS_PTR[cst++] = wall_clock64(); \
  } while (0)
  for (int i = c; i < items; i += C) {
    if (threadIdx.x < 64) {
      // lanes 0 .. M-1 poll one part each; the wave leaves the loop together
      bool need_remote = false;
      if (!remote) {
        int spin = 0;
        for (;;) {
          unsigned w = PUB_COUNT_MASK;
          if (lane < a.M && !dead) w = __hip_atomic_load(my + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ok = lane >= a.M || dead || (int)(w & PUB_COUNT_MASK) >= i + 1;
          if (ok && lane < a.M && !dead && (!(w & PUB_LOCAL) || ((w >> 16) & 0xfu) != mine_xcc || a.safe_sync))
            need_remote = true;
          if (__all(ok)) break;
          if (++spin > SPIN_LIMIT) { if (lane == 0) *a.err = 1; dead = true; }
        }
        if (__any(need_remote)) {
          // not behind every producer's L2: wait for the released, final publications
          remote = true;
          int spin2 = 0;
          for (;;) {
            unsigned w = PUB_COUNT_MASK;
            if (lane < a.M && !dead) w = __hip_atomic_load(my + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(lane >= a.M || dead || (int)(w & PUB_COUNT_MASK) >= items)) break;
            if (++spin2 > SPIN_LIMIT) { if (lane == 0) *a.err = 1; dead = true; }
          }
        }
        if (lane == 0) *mode = remote ? 1 : 0;
      }
      if (threadIdx.x == 0 && remote) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      else                            asm volatile("buffer_inv sc1" ::: "memory");
      INV_WAIT();
    }
    __syncthreads();
    CSTAMP();
    if (i < 2 * nb) {
      if constexpr (THREADS == 1024 && (NC == 16 || NC == 32)) {
        if (pg.dstamps) pg.dstamps += 8;
        if (i & 1) consumer_item_dma<NC, THREADS, 0>(pg, seg, nb - 1 - i / 2, ldsf);
        else       consumer_item_dma<NC, THREADS, 1>(pg, seg, nb - 1 - i / 2, ldsf);
      } else {
        if (*mode) param_grads_item<NC, THREADS, false>(pg, seg, nb - 1 - i / 2, (i & 1) ? 0 : 1, false, part, red);
        else       param_grads_item<NC, THREADS, true>(pg, seg, nb - 1 - i / 2, (i & 1) ? 0 : 1, false, part, red);
      }
    } else {
      fold_parts<THREADS>(pg, seg, a.L.p_lin0_w, 2 * NC);
      fold_parts<THREADS>(pg, seg, a.L.p_lin1_w, NC + 1);
    }
    __syncthreads();
    CSTAMP();
  }
#undef CSTAMP
  if (threadIdx.x < 64) {
    // reset the line for the next launch -- only after every producer's LAST publication, or that one would survive
    int spin = 0;
    for (;;) {
      unsigned w = PUB_COUNT_MASK;
      if (lane < a.M && !dead) w = __hip_atomic_load(my + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(lane >= a.M || dead || (int)(w & PUB_COUNT_MASK) >= items)) break;
      if (++spin > SPIN_LIMIT) { if (lane == 0) *a.err = 1; dead = true; }
    }
    if (lane < a.M) __hip_atomic_store(my + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ------------------------------------------------------------------------------------------ window kernel
// The same per-snapshot pipeline for split segments whose parts have compact ROW WINDOWS (gatres_graph_t.window: own
// rows + the rows adjacent to them form a contiguous range that is a fraction of the segment when the node order is
// local, as in water networks).  Every LDS table covers the window only (plain index shift: pointer - wlo * width),
// which frees more than half of the LDS.  The space buys what the one-table-per-segment layout has no room for:
//   * the x operand of every MFMA stage comes from an LDS copy of the producer stage's own rows, and W / W^T sit in
//     two slots filled by LDS-DMA (global_load_lds) one or more stages ahead: a projection starts computing at once
//     instead of after a W staging pass and a global round trip (~1.9 us each, four per block);
//   * the backward's gathers of SAVED tables (h, a_src, alpha: independent of the backward chain) read LDS copies of
//     the window that LDS-DMA prefetches a stage ahead, instead of going to L2 / Infinity Cache per neighbour.
// Stage functions, arithmetic and barrier protocol are exactly those of gatres_fused_kernel.
// LDS-DMA helpers.  Only waves w0 .. NW-1 issue: the compiler cannot prove that a wave's later LDS reads do not alias
// the DMA's destination (every table is a run-time offset into one LDS array), so it makes an issuing wave wait for
// its DMA (s_waitcnt vmcnt(0)) before its next ds_read -- which would serialise the copy with the stage it is meant to
// hide behind.  The MFMA stages of a split segment leave most waves idle (one 16-row tile per wave): those issue.
template <int THREADS>
__device__ __forceinline__ void dma_copy16(float* dst, const float* src, int nfloat, int w0) {   // nfloat % 4 == 0, 16-B aligned
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < w0) return;
  for (int c = (wave - w0) * 256; c < nfloat; c += (THREADS / 64 - w0) * 256)
    if (c + lane * 4 < nfloat) __builtin_amdgcn_global_load_lds(src + c + lane * 4, dst + c, 16, 0, 0);
}
template <int THREADS>
__device__ __forceinline__ void dma_copy4(float* dst, const float* src, int nfloat, int w0) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < w0) return;
  for (int c = (wave - w0) * 64; c < nfloat; c += (THREADS / 64 - w0) * 64)
    if (c + lane < nfloat) __builtin_amdgcn_global_load_lds(src + c + lane, dst + c, 4, 0, 0);
}
// W [M][K] (+ the two attention vectors) -> LDS slot in seg_proj's padded layout, one LDS-DMA instruction per row
template <int K, int M, int EPI, int THREADS>
__device__ __forceinline__ void w_prefetch(float* wl, const float* __restrict__ Wm, const float* __restrict__ att_src,
                                           const float* __restrict__ att_dst, int w0) {
  constexpr int KP = K + 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < w0) return;
  for (int m = wave - w0; m < M; m += THREADS / 64 - w0)
    if (lane < K / 4) __builtin_amdgcn_global_load_lds(Wm + m * K + lane * 4, wl + m * KP, 16, 0, 0);
  if constexpr (EPI == EPI_ATT) {
    if (wave == THREADS / 64 - 1 && lane < M / 4) __builtin_amdgcn_global_load_lds(att_src + lane * 4, wl + M * KP, 16, 0, 0);
    if (wave == THREADS / 64 - 2 && lane < M / 4) __builtin_amdgcn_global_load_lds(att_dst + lane * 4, wl + M * KP + M, 16, 0, 0);
  }
}

// LDS bytes of the two phases for given maxima (host + device): wr window rows, ow own rows, ge / gm window edges
// (hl: gatres_graph_t.halo -- the forward phase keeps two u16 lists of up to hl entries, the backward phase four)
__host__ __device__ inline long long win_fwd_bytes(int nc, int wr, int ow, int ge, int gm, int hl) {
  const long long wlf = 2LL * nc * (2 * nc + 4) + 4 * nc;
  return 4LL * (wr * (3LL * nc + 2) + ow * (3LL * nc + 2 + 8)) + 2 * (4 * wlf + 16) + 2LL * (2 * even(ow + 2) + even(ge) + even(gm)) +
         64 + 4LL * (hl + 4) + 32LL * ow + 16 + ow + 16;      // (+ the padded edge descriptors of k_window_stages.h: 2 x 16 B per own row, + export flags)
}
__host__ __device__ inline long long win_bwd_bytes(int nc, int threads, int wr, int ow, int ge, int gm, int hl) {
  const long long wlf = 2LL * nc * (2 * nc + 4) + 4 * nc;
  return 12LL * threads + 4LL * (wr * (4LL * nc + 2) + ow * (2LL * nc + 4 + 4) + 4LL * even(ge)) + 2 * (4 * wlf + 16) +
         2LL * (3 * even(ow + 2) + 3 * even(ge) + even(wr + 2) + even(gm)) + 64 + 8LL * (hl + 4) + 80LL * ow + 16 + ow + 16;   // (16 + 32 + 32 B of descriptors per own row, + export flags)
}

// Persistent LDS (top of the array, forward -> backward of one launch): per block and own row one 64-bit relu_bits word
// of conv1's output and one 32-bit word of the block input; the backward phase adds an own-row table of g_pre.
__host__ __device__ inline long long win_keep_bytes(int nb, int ow) { return (12LL * nb * ow + 15) & ~15LL; }

// Barrier between two stages of the window kernel whose hand-off goes through LDS only: waits for the wave's LDS
// operations, NOT for its outstanding global stores / LDS-DMA (s_waitcnt vmcnt), so the saved-activation stores of
// one stage drain while the next stages run.  Every stage that reads global data of the previous stage, or LDS data
// that a DMA delivers, is preceded by a full __syncthreads() / group_sync instead.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


}  // namespace
