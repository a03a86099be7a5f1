// Small dense / reduction kernels around the message-passing stack: lin0 / lin1 (PyG Linear(1,nc), Linear(nc,1);
// GraphModels.py:477,484,487,492), parameter-gradient column reductions, the slab reducer, weight transposes,
// and the caller-side pieces of the training step (device mask sampler, masked MSE, Adam; train.py:174-188).
//
// Reductions over nodes: `num_slabs` waves each own a contiguous node range and write their partial sums to
// their own slab at the parameter's flat offset; gatres_reduce_slabs adds the slabs in index order.  No atomics,
// bitwise reproducible.
#include "gatres_common.h"
#include "k_conv_grads.h"
#include "k_mask.h"

// ---------------------------------------------------------------------------------------------------- environment knobs
#include <atomic>
#include <cstdlib>
#include <mutex>
namespace {
gatres_knobs_t g_knobs;
std::atomic<bool> g_knobs_ready{false};
std::mutex g_knobs_mu;
int env_flag(const char* n) { const char* e = getenv(n); return (e && *e && !(e[0] == '0' && !e[1])) ? 1 : 0; }
int env_int(const char* n, int dflt) { const char* e = getenv(n); return (e && *e) ? atoi(e) : dflt; }
void read_knobs(gatres_knobs_t* k) {
  // product switches
  k->fused_split = env_int("GATRES_FUSED_SPLIT", 0);
  k->fused_safe_sync = env_flag("GATRES_FUSED_SAFE_SYNC");
  k->fused_no_halo = env_flag("GATRES_FUSED_NO_HALO");
  // consumer workgroups on a launch's spare CUs are opt-in since round 5: the stand-alone parameter-gradient launch
  // (param_grads_reg_kernel) is faster at every batch size that leaves CUs free (bs 8 / 16 / 24: 0.317 / 0.322 / 0.328 against
  // 0.347 / 0.349 / 0.352 ms/step), and it lets the update launch sample the next mask
  k->fused_no_consumers = env_flag("GATRES_FUSED_WITH_CONSUMERS") ? 0 : 1;
  k->fused_no_rounds = env_flag("GATRES_FUSED_NO_ROUNDS");
  const int lf = env_int("GATRES_AGG_LANE_FEATURES", 0);
  k->agg_lane_features = (lf == 4 || lf == 8) ? lf : 0;
  k->lin_bwd_wave = env_flag("GATRES_LIN_BWD_WAVE");
  k->no_proj_lds = env_flag("GATRES_NO_PROJ_LDS");
  k->dw_1d = env_flag("GATRES_DW_1D");
  k->side_stream = env_int("GATRES_SIDE_STREAM", -1);
  k->blocked = env_flag("GATRES_BLOCKED");
  k->window_runtime_phases = env_flag("GATRES_WINDOW_RUNTIME_PHASES");
  k->window_ph_mask = env_int("GATRES_WINDOW_PH_MASK", 0xffff);
  k->window_sync_start = env_flag("GATRES_WINDOW_SYNC_START");
  k->fused_no_keep = env_flag("GATRES_FUSED_NO_KEEP");
  // diagnostic build only
  k->agg_wide_offsets = 0; k->fused_threads = 1024; k->fused_no_window = 0; k->fused_prefer_consumers = 0;
  k->fused_consumers_cap = 2; k->fused_nocache = 0; k->fused_wide = 0; k->fused_heartbeat = 0;
  k->param_grads_no_stream = 0; k->proj_rows = 0; k->proj_stream = 0; k->dw_fp32 = 0; k->no_co_launch = 0;
  k->co_launch_always = 0; k->dw_slab_rows = 0; k->xch_nowait = 0; k->diag_nomask = 0;
#ifdef GATRES_DIAG_BUILD
  k->agg_wide_offsets = env_flag("GATRES_AGG_WIDE_OFFSETS");
  k->fused_threads = env_int("GATRES_FUSED_THREADS", 1024) == 512 ? 512 : 1024;
  k->fused_no_window = env_flag("GATRES_FUSED_NO_WINDOW");
  k->fused_prefer_consumers = env_flag("GATRES_FUSED_PREFER_CONSUMERS");
  { const int c = env_int("GATRES_FUSED_CONSUMERS", 2); k->fused_consumers_cap = c < 4 ? c : 4; }
  k->fused_nocache = env_flag("GATRES_FUSED_NOCACHE");
  k->fused_wide = env_flag("GATRES_FUSED_WIDE");
  k->fused_heartbeat = env_flag("GATRES_FUSED_HEARTBEAT");
  k->param_grads_no_stream = env_flag("GATRES_PARAM_GRADS_NO_STREAM");
  k->proj_rows = (env_int("GATRES_PROJ_ROWS", 0) + 63) & ~63;
  k->proj_stream = env_flag("GATRES_PROJ_STREAM");
  k->dw_fp32 = env_flag("GATRES_DW_FP32");
  k->no_co_launch = env_flag("GATRES_NO_CO_LAUNCH");
  k->co_launch_always = env_flag("GATRES_CO_LAUNCH_ALWAYS");
  k->dw_slab_rows = env_int("GATRES_DW_SLAB_ROWS", 0);
  k->xch_nowait = env_flag("GATRES_XCH_NOWAIT");
  k->diag_nomask = env_flag("GATRES_DIAG_NOMASK");
#endif
}
}  // namespace

extern "C" __attribute__((visibility("hidden"))) const gatres_knobs_t* gatres_knobs() {
  if (!g_knobs_ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs_ready.load(std::memory_order_relaxed)) {
      read_knobs(&g_knobs);
      g_knobs_ready.store(true, std::memory_order_release);
    }
  }
  return &g_knobs;
}
// Re-read the environment (a process that changed GATRES_* variables after the first call into the library).
extern "C" int gatres_knobs_reload(void) {
  std::lock_guard<std::mutex> lk(g_knobs_mu);
  read_knobs(&g_knobs);
  g_knobs_ready.store(true, std::memory_order_release);
  return 0;
}

static gatres_side_t* side_of_device(bool create) {
  constexpr int MAXDEV = 64;
  static gatres_side_t sides[MAXDEV];
  static int state[MAXDEV];                // 0: not tried, 1: ready, -1: failed
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(mu);
  if (state[dev] == 0 && create) {
    gatres_side_t& s = sides[dev];
    bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t* ev[4] = {&s.fork_a, &s.fork_b, &s.done_a, &s.done_b};
    for (int k = 0; k < 4 && ok; ++k) ok = hipEventCreateWithFlags(ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    s.mu = new std::mutex();
    state[dev] = ok ? 1 : -1;
  }
  return state[dev] == 1 ? &sides[dev] : nullptr;
}
extern "C" __attribute__((visibility("hidden"))) gatres_side_t* gatres_side() { return side_of_device(true); }
// the side stream if one was ever created on this device (never creates one)
extern "C" __attribute__((visibility("hidden"))) gatres_side_t* gatres_side_peek() { return side_of_device(false); }

namespace {

static inline int nodes_per_slab(int N, int num_slabs) {
  int nps = (N + num_slabs - 1) / num_slabs;
  return (nps + 3) & ~3;     // same rounding as the dW kernel so every reduction shares slab boundaries
}

// ---------------------------------------------------------------------------------------------------- lin0
template <typename T>
__global__ __launch_bounds__(256) void lin0_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       T* __restrict__ out, int N, int nc4) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (tid >= (long long)N * nc4) return;
  const int n = (int)(tid / nc4), c0 = (int)(tid % nc4) * 4;
  const float xv = (mask && mask[n]) ? 0.f : x[n];
  const float4 wv = ld4(w + c0), bv = ld4(b + c0);
  float4 o;
  o.x = xv * wv.x + bv.x; o.y = xv * wv.y + bv.y; o.z = xv * wv.z + bv.z; o.w = xv * wv.w + bv.w;
  strow4(out + (size_t)n * nc4 * 4 + c0, o);
}

// The same two sums with 256 threads per slab, for nc a power of two >= 4: G4 = nc / 4 lanes per node row (four columns
// each: one 16- or 8-byte load) and R = 256 / G4 rows per trip, four trips' loads in flight.  A thread sums its rows
// (nbeg + r, + R, ...) in order; the R partial sums of a column are then added in r order: deterministic.  The
// one-wave-per-slab form below moved 128 bytes per load instruction with at most 64 loads outstanding -- 120 us for
// 50 k x 128 bf16 -- and stays as the instance for other nc.
template <typename T>
__global__ __launch_bounds__(256) void lin0_bwd_rows_kernel(const T* __restrict__ g, const float* __restrict__ x,
                                                            const uint8_t* __restrict__ mask, float* __restrict__ slab_w,
                                                            float* __restrict__ slab_b, long long stride, int N, int nc,
                                                            int nps, int lgG4) {
  __shared__ __attribute__((aligned(16))) float red[256 * 8];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int G4 = 1 << lgG4, f = tid & (G4 - 1), r = tid >> lgG4, R = 256 >> lgG4;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  float4 aw = f4zero(), ab = f4zero();
  auto take = [&](float4 gv, float xv) {
    aw.x = fmaf(gv.x, xv, aw.x); aw.y = fmaf(gv.y, xv, aw.y); aw.z = fmaf(gv.z, xv, aw.z); aw.w = fmaf(gv.w, xv, aw.w);
    ab.x += gv.x; ab.y += gv.y; ab.z += gv.z; ab.w += gv.w;
  };
  int n = nbeg + r;
  for (; n + 3 * R < nend; n += 4 * R) {
    float4 gv[4];
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = n + u * R;
      gv[u] = ldrow4(g + (size_t)m * nc + f * 4);
      const float xr = x[m];
      xv[u] = (mask && mask[m]) ? 0.f : xr;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(gv[u], xv[u]);
  }
  for (; n < nend; n += R) take(ldrow4(g + (size_t)n * nc + f * 4), (mask && mask[n]) ? 0.f : x[n]);
  st4(red + tid * 8, aw); st4(red + tid * 8 + 4, ab);
  __syncthreads();
  if (tid < G4) {
    for (int k = 1; k < R; ++k) {
      const float4 w = ld4(red + (k * G4 + f) * 8), b = ld4(red + (k * G4 + f) * 8 + 4);
      aw.x += w.x; aw.y += w.y; aw.z += w.z; aw.w += w.w;
      ab.x += b.x; ab.y += b.y; ab.z += b.z; ab.w += b.w;
    }
    float* ow = slab_w + (size_t)s * stride + f * 4;
    float* ob = slab_b + (size_t)s * stride + f * 4;
    ow[0] = aw.x; ow[1] = aw.y; ow[2] = aw.z; ow[3] = aw.w;
    ob[0] = ab.x; ob[1] = ab.y; ob[2] = ab.z; ob[3] = ab.w;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lin1_bwd_rows_kernel(const float* __restrict__ g_out, const T* __restrict__ x,
                                                            const float* __restrict__ w, T* __restrict__ g_x,
                                                            float* __restrict__ slab_w, float* __restrict__ slab_b,
                                                            long long stride, int N, int nc, int nps, int relu_mask,
                                                            int lgG4) {
  __shared__ __attribute__((aligned(16))) float red[256 * 5];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int G4 = 1 << lgG4, f = tid & (G4 - 1), r = tid >> lgG4, R = 256 >> lgG4;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  const float wv[4] = {w[f * 4], w[f * 4 + 1], w[f * 4 + 2], w[f * 4 + 3]};
  float4 aw = f4zero();
  float ab = 0.f;
  auto take = [&](int m, float go, float4 xv) {
    ab += go;
    aw.x = fmaf(go, xv.x, aw.x); aw.y = fmaf(go, xv.y, aw.y); aw.z = fmaf(go, xv.z, aw.z); aw.w = fmaf(go, xv.w, aw.w);
    float4 o;
    o.x = (relu_mask && !(xv.x > 0.f)) ? 0.f : go * wv[0];
    o.y = (relu_mask && !(xv.y > 0.f)) ? 0.f : go * wv[1];
    o.z = (relu_mask && !(xv.z > 0.f)) ? 0.f : go * wv[2];
    o.w = (relu_mask && !(xv.w > 0.f)) ? 0.f : go * wv[3];
    strow4(g_x + (size_t)m * nc + f * 4, o);
  };
  int n = nbeg + r;
  for (; n + 3 * R < nend; n += 4 * R) {
    float4 xv[4];
    float go[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { xv[u] = ldrow4(x + (size_t)(n + u * R) * nc + f * 4); go[u] = g_out[n + u * R]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(n + u * R, go[u], xv[u]);
  }
  for (; n < nend; n += R) take(n, g_out[n], ldrow4(x + (size_t)n * nc + f * 4));
  st4(red + tid * 4, aw);
  red[1024 + tid] = ab;
  __syncthreads();
  if (tid < G4) {
    for (int k = 1; k < R; ++k) {
      const float4 t = ld4(red + (k * G4 + f) * 4);
      aw.x += t.x; aw.y += t.y; aw.z += t.z; aw.w += t.w;
      ab += red[1024 + k * G4];                   // (the f == 0 thread of row lane k)
    }
    float* ow = slab_w + (size_t)s * stride + f * 4;
    ow[0] = aw.x; ow[1] = aw.y; ow[2] = aw.z; ow[3] = aw.w;
    if (tid == 0) slab_b[(size_t)s * stride] = ab;
  }
}

// g_w[c] = sum_n g[n,c]*xm[n] ; g_b[c] = sum_n g[n,c]        (one wave per slab, lane = column)
template <typename T>
__global__ __launch_bounds__(64) void lin0_bwd_kernel(const T* __restrict__ g, const float* __restrict__ x,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ slab_w,
                                                      float* __restrict__ slab_b, long long stride, int N, int nc,
                                                      int nps) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n = nbeg; n < nend; ++n) {
    const float xv = (mask && mask[n]) ? 0.f : x[n];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = lane + 64 * cc;
      if (c < nc) {
        const float gv = ldval(g + (size_t)n * nc + c);
        aw[cc] = fmaf(gv, xv, aw[cc]);
        ab[cc] += gv;
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int c = lane + 64 * cc;
    if (c < nc) {
      slab_w[(size_t)s * stride + c] = aw[cc];
      slab_b[(size_t)s * stride + c] = ab[cc];
    }
  }
}

// ---------------------------------------------------------------------------------------------------- lin1
// out[n] = sum_c x[n,c]*w[c] + b   (G = nc/4 lanes per row)
template <typename T>
__global__ __launch_bounds__(256) void lin1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, int N,
                                                       int nc, int G, int lgG) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  int row = tid >> lgG;
  const bool valid = row < N;
  if (!valid) row = N - 1;
  const int c0 = (tid & (G - 1)) * 4;
  const float4 xv = ldrow4(x + (size_t)row * nc + c0), wv = ld4(w + c0);
  float d = xv.x * wv.x;
  d = fmaf(xv.y, wv.y, d); d = fmaf(xv.z, wv.z, d); d = fmaf(xv.w, wv.w, d);
  for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off);
  if (valid && c0 == 0) out[row] = d + (b ? b[0] : 0.f);
}

// g_x[n,c] = g_out[n]*w[c] (ReLU-masked by x>0);  slabs: g_w[c] = sum_n g_out[n]*x[n,c], g_b = sum_n g_out[n]
template <typename T>
__global__ __launch_bounds__(64) void lin1_bwd_kernel(const float* __restrict__ g_out, const T* __restrict__ x,
                                                      const float* __restrict__ w, T* __restrict__ g_x,
                                                      float* __restrict__ slab_w, float* __restrict__ slab_b,
                                                      long long stride, int N, int nc, int nps, int relu_mask) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, wv[4];
  float ab = 0.f;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) wv[cc] = (lane + 64 * cc < nc) ? w[lane + 64 * cc] : 0.f;
  for (int n = nbeg; n < nend; ++n) {
    const float go = g_out[n];
    ab += go;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = lane + 64 * cc;
      if (c < nc) {
        const float xv = ldval(x + (size_t)n * nc + c);
        aw[cc] = fmaf(go, xv, aw[cc]);
        stval(g_x + (size_t)n * nc + c, (relu_mask && !(xv > 0.f)) ? 0.f : go * wv[cc]);
      }
    }
  }
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int c = lane + 64 * cc;
    if (c < nc) slab_w[(size_t)s * stride + c] = aw[cc];
  }
  if (lane == 0) slab_b[(size_t)s * stride] = ab;
}

// -------------------------------------------------------------------------------- GATConv parameter grads
// g_att_src[hc] = sum_n g_a_src[n,h]*h[n,hc];  g_att_dst likewise;  g_bias[hc] = sum_n g_out[n,hc]
// (generic storage type: k_conv_grads.h)
template <typename T>
__global__ __launch_bounds__(256) void conv_param_grads_kernel(
    const T* __restrict__ h, const float* __restrict__ g_a_src, const float* __restrict__ g_a_dst,
    const T* __restrict__ g_out, float* __restrict__ slab_as, float* __restrict__ slab_ad,
    float* __restrict__ slab_b, long long stride, int N, int H, int C, int nps) {
  conv_param_grads_body<T>(blockIdx.x, h, g_a_src, g_a_dst, g_out, slab_as, slab_ad, slab_b, stride, N, H, C, nps);
}

// bf16 tables: k_conv_grads.h
__global__ __launch_bounds__(256) void conv_param_grads_bf16_kernel(
    const gatres_bf16* __restrict__ h, const float* __restrict__ g_a_src, const float* __restrict__ g_a_dst,
    const gatres_bf16* __restrict__ g_out, float* __restrict__ slab_as, float* __restrict__ slab_ad,
    float* __restrict__ slab_b, long long stride, int N, int H, int C, int nps) {
  conv_param_grads_bf16_body(blockIdx.x, h, g_a_src, g_a_dst, g_out, slab_as, slab_ad, slab_b, stride, N, H, C, nps);
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int num_slabs,
                                                           long long stride, long long count,
                                                           float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= num_slabs; s += 8) {                  // 8 loads in flight, summed in slab order
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = slabs[(size_t)(s + u) * stride + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v8[u];
  }
  for (; s < num_slabs; ++s) acc += slabs[(size_t)s * stride + idx];
  out[idx] = acc;
}

// The same for a range [lo_abs, lo_abs + count) of the flat parameter vector whose GATConv weight matrices only occupy the
// first w_slabs rows (the per-op driver asks the bf16 dW kernel for fewer, two-dimensional partials: k_proj.hip
// dw2d_bf16_kernel); every other parameter is summed over all num_slabs rows.  slabs / out point at the range's start.
__global__ __launch_bounds__(256) void reduce_slabs_regions_kernel(const float* __restrict__ slabs, int num_slabs, int w_slabs,
                                                                   long long stride, long long lo_abs, long long count,
                                                                   int nc, int nb, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  int rows = num_slabs;
  {
    const long long per = 2LL * nc * nc, bstride = 9LL * nc + 2 * per, off = lo_abs + idx - 2LL * nc;
    if (off >= 0 && off < (long long)nb * bstride) {
      const long long o = off % bstride;
      if ((o >= 6LL * nc && o < 6LL * nc + per) || o >= 9LL * nc + per) rows = w_slabs;
    }
  }
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= rows; s += 8) {                       // 8 loads in flight, summed in slab order
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = slabs[(size_t)(s + u) * stride + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v8[u];
  }
  for (; s < rows; ++s) acc += slabs[(size_t)s * stride + idx];
  out[idx] = acc;
}

// wt block layout: [W1^T : nc x 2nc][W2^T : 2nc x nc]
__global__ __launch_bounds__(256) void transpose_conv_weights_kernel(const float* __restrict__ params,
                                                                     float* __restrict__ wt, int num_blocks,
                                                                     int nc) {
  const int per = 2 * nc * nc;                       // elements of one conv weight
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)num_blocks * 2 * per) return;
  const int b = (int)(idx / (2 * per));
  const int r = (int)(idx % (2 * per));
  const int conv = r / per, e = r % per;
  const long long blk = 2LL * nc + (long long)b * (9LL * nc + 4LL * nc * nc);   // lin0.w + lin0.b, then blocks
  // conv1: att_src[2nc] att_dst[2nc] bias[2nc] W[2nc, nc];  conv2: att_src[nc] att_dst[nc] bias[nc] W[nc, 2nc]
  const float* W = params + blk + (conv == 0 ? 6LL * nc : 6LL * nc + per + 3LL * nc);
  const int rows = conv == 0 ? 2 * nc : nc, cols = conv == 0 ? nc : 2 * nc;
  const int orow = e / rows, ocol = e % rows;        // output is [cols, rows]
  wt[(size_t)b * 2 * per + (size_t)conv * per + e] = W[(size_t)ocol * cols + orow];
}

// bf16 operand copies of the GATConv weights for the bf16 projections: per block [W1 | W2 | W1^T | W2^T], each 2nc^2 bf16
__global__ __launch_bounds__(256) void convert_conv_weights_bf16_kernel(const float* __restrict__ params,
                                                                        gatres_bf16* __restrict__ wb, int num_blocks,
                                                                        int nc) {
  const int per = 2 * nc * nc;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)num_blocks * 4 * per) return;
  const int b = (int)(idx / (4 * per));
  const int r = (int)(idx % (4 * per));
  const int which = r / per, e = r % per;                // 0: W1, 1: W2, 2: W1^T, 3: W2^T
  const int conv = which & 1;
  const long long blk = 2LL * nc + (long long)b * (9LL * nc + 4LL * nc * nc);
  const float* W = params + blk + (conv == 0 ? 6LL * nc : 6LL * nc + per + 3LL * nc);
  const int rows = conv == 0 ? 2 * nc : nc, cols = conv == 0 ? nc : 2 * nc;     // W is [rows, cols]
  float v;
  if (which < 2) v = W[e];
  else { const int orow = e / rows, ocol = e % rows; v = W[(size_t)ocol * cols + orow]; }      // output [cols, rows]
  wb[(size_t)b * 4 * per + (size_t)which * per + e] = (gatres_bf16)v;
}

// ------------------------------------------------------------------------------------- node relabelling
// Per-op path of a relabelled plan (gatres_graph_t.perm): caller-order vectors are gathered into plan order before
// lin0 / the backward, and the results scattered back.  (The fused kernels index through perm directly.)
__global__ __launch_bounds__(256) void permute_f32_kernel(const float* __restrict__ src, const int* __restrict__ perm,
                                                          float* __restrict__ dst, int N, int scatter) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (scatter) dst[perm[i]] = src[i];
  else         dst[i] = src[perm[i]];
}
__global__ __launch_bounds__(256) void gather_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ perm,
                                                        uint8_t* __restrict__ dst, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) dst[i] = src[perm[i]];
}

// ------------------------------------------------------------------------------------------ edge_index hash


__global__ __launch_bounds__(256) void edge_hash_kernel(const int64_t* __restrict__ ei, long long E,
                                                        unsigned long long* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = 0;
  if (e < E) {
    const uint64_t a = mix64((uint64_t)ei[e] + 0x9E3779B97F4A7C15ULL * (uint64_t)(e + 1));
    v = mix64(a ^ ((uint64_t)ei[E + e] + 0xD1B54A32D192ED03ULL));
  }
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);      // integer add: order-independent
}

// ---------------------------------------------------------------------------------- device mask sampler (k_mask.h)
// One launch: MASK_WGS workgroups per graph; the batch copy -- or its collation from the snapshot matrix -- rides on it
// (gatres_stage_batch_mask / gatres_stage_rows_mask: grid-stride, before the sampling: its stores drain while the keys are ranked)
__global__ __launch_bounds__(1024) void mask_generate_kernel(const int* __restrict__ node_ptr, double rate,
                                                             uint64_t seed, const uint64_t* __restrict__ step_counter,
                                                             uint8_t* __restrict__ mask,
                                                             const float* __restrict__ x_src, const float* __restrict__ y_src,
                                                             float* __restrict__ x_dst, float* __restrict__ y_dst,
                                                             long long num_nodes, const long long* __restrict__ rows,
                                                             int rows_npg) {
  if (x_src && x_src != x_dst) {
    for (long long v = (long long)blockIdx.x * 1024 + threadIdx.x; v < num_nodes; v += (long long)gridDim.x * 1024) {
      const long long u = rows ? rows[v / rows_npg] * rows_npg + v % rows_npg : v;
      x_dst[v] = x_src[u];
      if (y_src && y_dst) y_dst[v] = y_src[u];
    }
  }
  mask_sample_graph<1024, MASK_WGS>(node_ptr, rate, seed, step_counter ? step_counter[0] : 0, mask, blockIdx.x / MASK_WGS,
                          blockIdx.x % MASK_WGS);
}

// ------------------------------------------------------------------------------------------- masked MSE
__global__ __launch_bounds__(1024) void masked_mse_kernel(const float* __restrict__ out, const float* __restrict__ y,
                                                          const uint8_t* __restrict__ mask, float* __restrict__ loss,
                                                          float* __restrict__ g_out, int N) {
  __shared__ float s_sum[1024];
  __shared__ int s_cnt[1024];
  const int tid = threadIdx.x;
  float acc = 0.f;
  int cnt = 0;
  int n = tid;
  for (; n + 3 * 1024 < N; n += 4 * 1024) {   // (four strides' loads in flight; accumulated in stride order: same bits)
    uint8_t mk[4];
    float o[4], t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { mk[u] = mask[n + u * 1024]; o[u] = out[n + u * 1024]; t[u] = y[n + u * 1024]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (mk[u]) {
        const float d = o[u] - t[u];
        acc = fmaf(d, d, acc);
        ++cnt;
      }
  }
  for (; n < N; n += 1024) {
    if (mask[n]) {
      const float d = out[n] - y[n];
      acc = fmaf(d, d, acc);
      ++cnt;
    }
  }
  s_sum[tid] = acc; s_cnt[tid] = cnt;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) { s_sum[tid] += s_sum[tid + off]; s_cnt[tid] += s_cnt[tid + off]; }
    __syncthreads();
  }
  const int M = s_cnt[0];
  if (tid == 0) loss[0] = s_sum[0] / (float)M;            // M == 0 -> NaN, like torch's mean of an empty tensor
  const float scale = M > 0 ? 2.f / (float)M : 0.f;
  _Pragma("unroll 4") for (int n = tid; n < N; n += 1024) g_out[n] = mask[n] ? (out[n] - y[n]) * scale : 0.f;
}

// ------------------------------------------------------------------------------------------------- Adam
// torch.optim.Adam (single-tensor path): g += wd*p; m.lerp_(g, 1-b1); v = b2*v + (1-b2) g*g;
// p += (-(lr/bc1) * m) / (sqrt(v)/sqrt(bc2) + eps).   step_counter[0] = step, [1] = block ticket.
constexpr int ADAM_MAX_BLOCKS = 512;
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   unsigned long long* __restrict__ step_counter, long long count,
                                                   double lr, double b1, double b2, double eps, double wd,
                                                   float grad_scale, float* __restrict__ wt, int nb, int nc,
                                                   unsigned* __restrict__ drop_count, const double* __restrict__ hp,
                                                   const int* __restrict__ mask_node_ptr, double mask_rate,
                                                   unsigned long long mask_seed, uint8_t* __restrict__ mask_next,
                                                   const unsigned long long* __restrict__ mask_snap, int update_blocks) {
  __shared__ float s_step_size, s_bc2_sqrt;
  __shared__ int s_drop;
  unsigned long long done = 0ULL;
  // The device mask of the NEXT step, sampled by the workgroups behind the update's (the data-parallel step's Adam phase; the
  // single-GPU step does the same in reduce_adam_kernel, k_fused_host.hip).  Its key uses the step count this update leaves
  // behind: the count as the parameter-gradient launch of this step snapshotted it (mask_snap[0]: nothing has moved it since;
  // the update blocks increment it at their very end), + 1 unless the step is dropped (the all-reduced gradient's fault mark).
  if (mask_next && (int)blockIdx.x >= update_blocks) {
    const int mb = (int)blockIdx.x - update_blocks;
    const bool dropped = drop_count && g[0] != g[0];
    mask_sample_graph<256, MASK_WGS_UPDATE>(mask_node_ptr, mask_rate, mask_seed, mask_snap[0] + (dropped ? 0ULL : 1ULL),
                                            mask_next, mb / MASK_WGS_UPDATE, mb % MASK_WGS_UPDATE);
    return;
  }
  if (hp) { lr = hp[0]; b1 = hp[1]; b2 = hp[2]; eps = hp[3]; wd = hp[4]; }      // (gatres_train_step_t.hparams)
  // drop_count != null (the data-parallel Adam phase of the fused path only): a fused launch that faulted marks EVERY
  // gradient entry NaN (gatres_fused_finish) and the all-reduce spreads the mark to all ranks; the step is then dropped on
  // every replica alike -- no update, no step count -- and COUNTED in *drop_count (status word 3: a NaN that a diverging run
  // puts into g[0] drops steps too, and must not do so silently; GATResTrainer.dropped_steps).  Everywhere else
  // (gatres_adam_step, FusedAdam) a NaN gradient propagates exactly as in torch.optim.Adam, which the reference uses
  // (train.py:348).
  if (threadIdx.x == 0) s_drop = (drop_count && g[0] != g[0]) ? 1 : 0;
  __syncthreads();
  if (s_drop) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(drop_count, 1u);
    return;
  }
  // No tickets where the step count was snapshotted for this launch (mask_next != null: the data-parallel step's Adam phase
  // behind a parameter-gradient launch that left mask_snap, exactly the condition under which the sampling blocks above use
  // it): every block reads the SNAPSHOT and block 0 alone stores the new count -- up to 258 read-modify-writes on one address
  // serialise in the L2 at ~20 ns each, measured 7.5 -> 6.4 us for this launch (reduce_adam_kernel lost its tickets the same way).
  const bool snap = mask_next != nullptr;
  if (threadIdx.x == 0) {
    const unsigned long long t = (snap ? mask_snap[0]
                                       : __hip_atomic_load(step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1ULL;
    const double bc1 = 1.0 - gatres_powi(b1, t), bc2 = 1.0 - gatres_powi(b2, t);
    s_step_size = (float)(lr / bc1);
    s_bc2_sqrt = (float)sqrt(bc2);
    if (snap) {
      if (blockIdx.x == 0) step_counter[0] = t;         // (nobody reads the counter in this launch: the blocks read the snapshot)
    } else {
      // the step is counted once every block has READ the counter: the ticket follows this block's read (no fence: nothing
      // orders the count behind the parameter stores but the next launch -- see reduce_adam_kernel, k_fused.hip)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      done = atomicAdd(&step_counter[1], 1ULL);         // (its result is only looked at after this block's elements)
    }
  }
  __syncthreads();
  // Grid-stride: the launch has at most ADAM_MAX_BLOCKS blocks.  One block per 256 parameters meant 6.6 k tickets on ONE
  // address for gatres_large -- they serialise in the L2 at ~20 ns each and were the whole 146 us of the launch.
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (long long)update_blocks * 256) {
  {
    const float pv = p[idx];
    float gv = g[idx] * grad_scale;
    gv = gv + (float)wd * pv;
    float mv = m[idx];
    mv = mv + (float)(1.0 - b1) * (gv - mv);
    const float vv = (float)b2 * v[idx] + (float)(1.0 - b2) * gv * gv;
    const float denom = sqrtf(vv) / s_bc2_sqrt + (float)eps;
    const float pn = pv + (-s_step_size * mv) / denom;
    p[idx] = pn;
    m[idx] = mv;
    v[idx] = vv;
    if (wt) {                     // keep the fused kernels' transposed conv weights current (transpose_conv_weights_kernel's layout)
      const long long per = 2LL * nc * nc, stride = 9LL * nc + 2 * per, off = idx - 2LL * nc;
      if (off >= 0 && off < (long long)nb * stride) {
        const long long b = off / stride, o = off % stride;
        if (o >= 6LL * nc && o < 6LL * nc + per) {                        // W1 [2nc][nc] -> [nc][2nc]
          const long long e = o - 6LL * nc, row = e / nc, col = e % nc;
          wt[b * 2 * per + col * 2 * nc + row] = pn;
        } else if (o >= 9LL * nc + per) {                                 // W2 [nc][2nc] -> [2nc][nc]
          const long long e = o - 9LL * nc - per, row = e / (2 * nc), col = e % (2 * nc);
          wt[b * 2 * per + per + col * nc + row] = pn;
        }
      }
    }
  }
  }
  if (!snap && threadIdx.x == 0 && done == (unsigned long long)update_blocks - 1ULL) {
    step_counter[1] = 0ULL;
    atomicAdd(&step_counter[0], 1ULL);
  }
}

}  // namespace

#define GATRES_DISPATCH_T(dtype_, CALL_)                                  \
  switch (dtype_) {                                                       \
    case GATRES_DTYPE_F32: { using T = float; CALL_; break; }             \
    case GATRES_DTYPE_BF16: { using T = gatres_bf16; CALL_; break; }      \
    default: return GATRES_E_UNSUPPORTED;                                 \
  }

extern "C" int gatres_t_lin0_fwd(const float* x, const uint8_t* mask, const float* w, const float* b, void* out, int num_nodes, int nc,
                      int dtype, void* stream) {
  if (!x || !w || !b || !out || num_nodes <= 0) return GATRES_E_BADARG;
  if (nc < 4 || nc % 4) return GATRES_E_UNSUPPORTED;
  if (!gatres_aligned16(w) || !gatres_aligned16(b) || !gatres_aligned16(out)) return GATRES_E_BADARG;
  const long long total = (long long)num_nodes * (nc / 4);
  GATRES_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((lin0_fwd_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, gatres_stream(stream), x,
                       mask, w, b, (T*)out, num_nodes, nc / 4);
  })
  return gatres_launch_status();
}
extern "C" int gatres_lin0_fwd(const float* x, const uint8_t* mask, const float* w, const float* b, float* out,
                               int32_t num_nodes, int32_t nc, void* stream) {
  return gatres_t_lin0_fwd(x, mask, w, b, out, num_nodes, nc, GATRES_DTYPE_F32, stream);
}

// the row-wise lin0 / lin1 backward kernels take nc = 4, 8, ..., 256 (a power of two) and 16-byte aligned activations
static inline int ilog2_(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool lin_rows_form(int nc, const void* act) {
  return nc >= 4 && nc <= 256 && gatres_is_pow2(nc) && gatres_aligned16(act) && !gatres_knobs()->lin_bwd_wave;
}

extern "C" int gatres_t_lin0_bwd(const void* g, const float* x, const uint8_t* mask, float* slab_w, float* slab_b, int num_slabs,
                      int64_t slab_stride, int num_nodes, int nc, int dtype, void* stream) {
  if (!g || !x || !slab_w || !slab_b || num_nodes <= 0 || num_slabs <= 0) return GATRES_E_BADARG;
  if (nc < 1 || nc > 256) return GATRES_E_UNSUPPORTED;
  if (lin_rows_form(nc, g)) {
    GATRES_DISPATCH_T(dtype, {
      hipLaunchKernelGGL((lin0_bwd_rows_kernel<T>), dim3(num_slabs), dim3(256), 0, gatres_stream(stream), (const T*)g, x,
                         mask, slab_w, slab_b, (long long)slab_stride, num_nodes, nc,
                         nodes_per_slab(num_nodes, num_slabs), ilog2_(nc / 4));
    })
    return gatres_launch_status();
  }
  GATRES_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((lin0_bwd_kernel<T>), dim3(num_slabs), dim3(64), 0, gatres_stream(stream), (const T*)g, x, mask,
                       slab_w, slab_b, (long long)slab_stride, num_nodes, nc, nodes_per_slab(num_nodes, num_slabs));
  })
  return gatres_launch_status();
}
extern "C" int gatres_lin0_bwd(const float* g, const float* x, const uint8_t* mask, float* slab_w, float* slab_b,
                               int32_t num_slabs, int64_t slab_stride, int32_t num_nodes, int32_t nc, void* stream) {
  return gatres_t_lin0_bwd(g, x, mask, slab_w, slab_b, num_slabs, slab_stride, num_nodes, nc, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_t_lin1_fwd(const void* x, const float* w, const float* b, float* out, int num_nodes, int nc, int dtype,
                      void* stream) {
  if (!x || !w || !out || num_nodes <= 0) return GATRES_E_BADARG;
  if (nc < 4 || !gatres_is_pow2(nc) || nc > 256) return GATRES_E_UNSUPPORTED;
  if (!gatres_aligned16(x) || !gatres_aligned16(w)) return GATRES_E_BADARG;
  const int G = nc / 4;
  int lgG = 0;
  while ((1 << lgG) < G) ++lgG;
  const long long threads = (long long)num_nodes * G;
  GATRES_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((lin1_fwd_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                       (const T*)x, w, b, out, num_nodes, nc, G, lgG);
  })
  return gatres_launch_status();
}
extern "C" int gatres_lin1_fwd(const float* x, const float* w, const float* b, float* out, int32_t num_nodes,
                               int32_t nc, void* stream) {
  return gatres_t_lin1_fwd(x, w, b, out, num_nodes, nc, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_t_lin1_bwd(const float* g_out, const void* x, const float* w, void* g_x, float* slab_w, float* slab_b,
                      int num_slabs, int64_t slab_stride, int num_nodes, int nc, int relu_mask, int dtype, void* stream) {
  if (!g_out || !x || !w || !g_x || !slab_w || !slab_b || num_nodes <= 0 || num_slabs <= 0) return GATRES_E_BADARG;
  if (nc < 1 || nc > 256) return GATRES_E_UNSUPPORTED;
  if (lin_rows_form(nc, x) && gatres_aligned16(g_x)) {
    GATRES_DISPATCH_T(dtype, {
      hipLaunchKernelGGL((lin1_bwd_rows_kernel<T>), dim3(num_slabs), dim3(256), 0, gatres_stream(stream), g_out,
                         (const T*)x, w, (T*)g_x, slab_w, slab_b, (long long)slab_stride, num_nodes, nc,
                         nodes_per_slab(num_nodes, num_slabs), relu_mask, ilog2_(nc / 4));
    })
    return gatres_launch_status();
  }
  GATRES_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((lin1_bwd_kernel<T>), dim3(num_slabs), dim3(64), 0, gatres_stream(stream), g_out, (const T*)x, w,
                       (T*)g_x, slab_w, slab_b, (long long)slab_stride, num_nodes, nc,
                       nodes_per_slab(num_nodes, num_slabs), relu_mask);
  })
  return gatres_launch_status();
}
extern "C" int gatres_lin1_bwd(const float* g_out, const float* x, const float* w, float* g_x, float* slab_w,
                               float* slab_b, int32_t num_slabs, int64_t slab_stride, int32_t num_nodes, int32_t nc,
                               int32_t relu_mask, void* stream) {
  return gatres_t_lin1_bwd(g_out, x, w, g_x, slab_w, slab_b, num_slabs, slab_stride, num_nodes, nc, relu_mask,
                           GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_t_conv_param_grads(const void* h, const float* g_a_src, const float* g_a_dst, const void* g_out,
                              float* slab_att_src, float* slab_att_dst, float* slab_bias, int num_slabs,
                              int64_t slab_stride, int num_nodes, int H, int C, int dtype, void* stream) {
  if (!h || !g_a_src || !g_a_dst || !g_out || !slab_att_src || !slab_att_dst || !slab_bias || num_nodes <= 0 ||
      num_slabs <= 0)
    return GATRES_E_BADARG;
  if (H < 1 || C < 1 || H * C > 256) return GATRES_E_UNSUPPORTED;
  if (dtype == GATRES_DTYPE_BF16) {
    if ((H * C) % 2 || C % 2) return GATRES_E_UNSUPPORTED;
    hipLaunchKernelGGL(conv_param_grads_bf16_kernel, dim3(num_slabs), dim3(256), 0, gatres_stream(stream),
                       (const gatres_bf16*)h, g_a_src, g_a_dst, (const gatres_bf16*)g_out, slab_att_src, slab_att_dst,
                       slab_bias, (long long)slab_stride, num_nodes, H, C, nodes_per_slab(num_nodes, num_slabs));
    return gatres_launch_status();
  }
  if (dtype != GATRES_DTYPE_F32) return GATRES_E_UNSUPPORTED;
  hipLaunchKernelGGL((conv_param_grads_kernel<float>), dim3(num_slabs), dim3(256), 0, gatres_stream(stream),
                     (const float*)h, g_a_src, g_a_dst, (const float*)g_out, slab_att_src, slab_att_dst, slab_bias,
                     (long long)slab_stride, num_nodes, H, C, nodes_per_slab(num_nodes, num_slabs));
  return gatres_launch_status();
}
extern "C" int gatres_conv_param_grads(const float* h, const float* g_a_src, const float* g_a_dst,
                                       const float* g_out, float* slab_att_src, float* slab_att_dst,
                                       float* slab_bias, int32_t num_slabs, int64_t slab_stride, int32_t num_nodes,
                                       int32_t H, int32_t C, void* stream) {
  return gatres_t_conv_param_grads(h, g_a_src, g_a_dst, g_out, slab_att_src, slab_att_dst, slab_bias, num_slabs,
                                   slab_stride, num_nodes, H, C, GATRES_DTYPE_F32, stream);
}

extern "C" int gatres_convert_conv_weights_bf16(const float* params, void* wb, int num_blocks, int nc, void* stream) {
  if (!params || !wb || num_blocks < 0 || nc < 1) return GATRES_E_BADARG;
  if (num_blocks == 0) return 0;
  const long long total = (long long)num_blocks * 8 * nc * nc;
  hipLaunchKernelGGL(convert_conv_weights_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     gatres_stream(stream), params, (gatres_bf16*)wb, num_blocks, nc);
  return gatres_launch_status();
}

extern "C" int gatres_reduce_slabs(const float* slabs, int32_t num_slabs, int64_t slab_stride, int64_t count,
                                   float* out, void* stream) {
  if (!slabs || !out || num_slabs <= 0 || count <= 0 || slab_stride < count) return GATRES_E_BADARG;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     slabs, num_slabs, (long long)slab_stride, (long long)count, out);
  return gatres_launch_status();
}

// (internal: model_driver.hip)
extern "C" __attribute__((visibility("hidden"))) int gatres_reduce_slabs_regions(const float* slabs, int32_t num_slabs,
                                                                                 int32_t w_slabs, int64_t slab_stride,
                                                                                 int64_t lo_abs, int64_t count, int32_t nc,
                                                                                 int32_t nb, float* out, void* stream) {
  if (!slabs || !out || num_slabs <= 0 || w_slabs <= 0 || w_slabs > num_slabs || count <= 0) return GATRES_E_BADARG;
  hipLaunchKernelGGL(reduce_slabs_regions_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     slabs, num_slabs, w_slabs, (long long)slab_stride, (long long)lo_abs, (long long)count, nc, nb, out);
  return gatres_launch_status();
}

extern "C" int gatres_transpose_conv_weights(const float* params, float* wt, int32_t num_blocks, int32_t nc,
                                             void* stream) {
  if (!params || !wt || num_blocks < 0 || nc < 1) return GATRES_E_BADARG;
  if (num_blocks == 0) return 0;
  const long long total = (long long)num_blocks * 4 * nc * nc;
  hipLaunchKernelGGL(transpose_conv_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     gatres_stream(stream), params, wt, num_blocks, nc);
  return gatres_launch_status();
}

extern "C" int gatres_edge_index_hash(const int64_t* edge_index, int64_t num_edges, uint64_t* hash_out,
                                      void* stream) {
  if (!hash_out || num_edges < 0 || (!edge_index && num_edges > 0)) return GATRES_E_BADARG;
  if (num_edges == 0) return 0;
  hipLaunchKernelGGL(edge_hash_kernel, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     edge_index, (long long)num_edges, reinterpret_cast<unsigned long long*>(hash_out));
  return gatres_launch_status();
}

extern "C" int gatres_mask_generate(const int32_t* node_ptr, int32_t num_graphs, double mask_rate, uint64_t seed,
                                    const uint64_t* step_counter, uint8_t* mask, void* stream) {
  if (!node_ptr || !mask || num_graphs <= 0 || !(mask_rate >= 0.0 && mask_rate <= 1.0)) return GATRES_E_BADARG;
  hipLaunchKernelGGL(mask_generate_kernel, dim3(num_graphs * MASK_WGS), dim3(1024), 0, gatres_stream(stream), node_ptr, mask_rate,
                     seed, step_counter, mask, nullptr, nullptr, nullptr, nullptr, 0LL, nullptr, 1);
  return gatres_launch_status();
}

extern "C" int gatres_stage_batch_mask(const float* x_src, const float* y_src, float* x_dst, float* y_dst, int64_t num_nodes,
                                       const int32_t* node_ptr, int32_t num_graphs, double mask_rate, uint64_t seed,
                                       const uint64_t* step_counter, uint8_t* mask, void* stream) {
  if (!node_ptr || !mask || num_graphs <= 0 || !(mask_rate >= 0.0 && mask_rate <= 1.0)) return GATRES_E_BADARG;
  if (!x_src || !x_dst || num_nodes <= 0 || ((y_src == nullptr) != (y_dst == nullptr))) return GATRES_E_BADARG;
  hipLaunchKernelGGL(mask_generate_kernel, dim3(num_graphs * MASK_WGS), dim3(1024), 0, gatres_stream(stream), node_ptr, mask_rate,
                     seed, step_counter, mask, x_src, y_src, x_dst, y_dst, (long long)num_nodes, nullptr, 1);
  return gatres_launch_status();
}

extern "C" int gatres_stage_rows_mask(const float* data, const int64_t* rows, int32_t nodes_per_graph, float* x_dst,
                                      float* y_dst, const int32_t* node_ptr, int32_t num_graphs, double mask_rate,
                                      uint64_t seed, const uint64_t* step_counter, uint8_t* mask, void* stream) {
  if (!node_ptr || !mask || num_graphs <= 0 || !(mask_rate >= 0.0 && mask_rate <= 1.0)) return GATRES_E_BADARG;
  if (!data || !rows || !x_dst || nodes_per_graph <= 0 || data == x_dst) return GATRES_E_BADARG;
  static_assert(sizeof(long long) == sizeof(int64_t), "row indices");
  hipLaunchKernelGGL(mask_generate_kernel, dim3(num_graphs * MASK_WGS), dim3(1024), 0, gatres_stream(stream), node_ptr, mask_rate,
                     seed, step_counter, mask, data, y_dst ? data : nullptr, x_dst, y_dst,
                     (long long)num_graphs * nodes_per_graph, reinterpret_cast<const long long*>(rows), (int)nodes_per_graph);
  return gatres_launch_status();
}

extern "C" int gatres_masked_mse(const float* out, const float* y, const uint8_t* mask, float* loss, float* g_out,
                                 int32_t num_nodes, void* stream) {
  if (!out || !y || !mask || !loss || !g_out || num_nodes <= 0) return GATRES_E_BADARG;
  hipLaunchKernelGGL(masked_mse_kernel, dim3(1), dim3(1024), 0, gatres_stream(stream), out, y, mask, loss, g_out,
                     num_nodes);
  return gatres_launch_status();
}

static inline unsigned adam_blocks(long long count) {
  const long long need = (count + 255) / 256;
  return (unsigned)(need < ADAM_MAX_BLOCKS ? need : ADAM_MAX_BLOCKS);
}

extern "C" int gatres_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                uint64_t* step_counter, int64_t count, double lr, double beta1, double beta2,
                                double eps, double weight_decay, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step_counter || count <= 0) return GATRES_E_BADARG;
  hipLaunchKernelGGL(adam_kernel, dim3(adam_blocks(count)), dim3(256), 0, gatres_stream(stream), params,
                     grads, exp_avg, exp_avg_sq, reinterpret_cast<unsigned long long*>(step_counter),
                     (long long)count, lr, beta1, beta2, eps, weight_decay, grad_scale, (float*)nullptr, 0, 0,
                     (unsigned*)nullptr, (const double*)nullptr, (const int*)nullptr, 0., 0ULL, (uint8_t*)nullptr,
                     (const unsigned long long*)nullptr, (int)adam_blocks(count));
  return gatres_launch_status();
}

// (not part of include/gatres.h) Adam with everything gatres_train_step may ask for: hyper-parameters from a device buffer
// (hp: double[5] {lr, beta1, beta2, eps, weight_decay}, or null), the fused path's transposed conv weights kept current
// (wt, or null), fault-marked steps dropped and counted (drop_count, or null), the next step's device mask sampled by extra
// workgroups (mask_next, or null; mask_snap: {step count before this update, -}, stable during the launch).
extern "C" __attribute__((visibility("hidden"))) int gatres_adam_step_ex(float* params, const float* grads, float* exp_avg,
                                   float* exp_avg_sq, uint64_t* step_counter, int64_t count, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, const double* hp, float grad_scale,
                                   float* wt, int32_t num_blocks, int32_t nc, uint32_t* drop_count,
                                   const int32_t* mask_node_ptr, int32_t mask_graphs, double mask_rate, uint64_t mask_seed,
                                   uint8_t* mask_next, const uint64_t* mask_snap, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step_counter || count <= 0) return GATRES_E_BADARG;
  const bool sample = mask_next && mask_node_ptr && mask_snap && mask_graphs > 0;
  const unsigned ub = adam_blocks(count);
  hipLaunchKernelGGL(adam_kernel, dim3(ub + (sample ? (unsigned)mask_graphs * MASK_WGS_UPDATE : 0u)), dim3(256), 0,
                     gatres_stream(stream), params,
                     grads, exp_avg, exp_avg_sq, reinterpret_cast<unsigned long long*>(step_counter),
                     (long long)count, lr, beta1, beta2, eps, weight_decay, grad_scale, wt, (int)num_blocks, (int)nc,
                     drop_count, hp, sample ? mask_node_ptr : nullptr, mask_rate, (unsigned long long)mask_seed,
                     sample ? mask_next : nullptr, reinterpret_cast<const unsigned long long*>(mask_snap), (int)ub);
  return gatres_launch_status();
}

extern "C" int gatres_permute_f32(const float* src, const int32_t* perm, float* dst, int32_t num_nodes, int32_t scatter,
                                  void* stream) {
  if (!src || !perm || !dst || num_nodes <= 0) return GATRES_E_BADARG;
  hipLaunchKernelGGL(permute_f32_kernel, dim3((unsigned)((num_nodes + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     src, perm, dst, num_nodes, scatter);
  return gatres_launch_status();
}

extern "C" int gatres_gather_u8(const uint8_t* src, const int32_t* perm, uint8_t* dst, int32_t num_nodes, void* stream) {
  if (!src || !perm || !dst || num_nodes <= 0) return GATRES_E_BADARG;
  hipLaunchKernelGGL(gather_u8_kernel, dim3((unsigned)((num_nodes + 255) / 256)), dim3(256), 0, gatres_stream(stream),
                     src, perm, dst, num_nodes);
  return gatres_launch_status();
}
