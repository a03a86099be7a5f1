// Shared device/host helpers for the gfx950 GATRes kernels.  Compiled with -ffp-contract=off: every
// multiply-add below rounds twice unless written as fmaf(), which keeps the sparse sums in the same
// association as the reference's index_add_/scatter path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gatres.h"

#define GATRES_NEG_SLOPE 0.2f        // GATConv default negative_slope
#define GATRES_SOFTMAX_EPS 1e-16f    // torch_geometric.utils.softmax: out_sum + 1e-16
#define GATRES_WAVE 64

// The DIAGNOSTIC build of the library (-DGATRES_DIAG_BUILD: _build.build_native(diag=True) -> lib/libgatres_hip_diag.so,
// loaded when GATRES_DIAG_LIB=1) carries stage stamps, the switches that produce WRONG results and every switch that selects
// a measured-and-lost alternative; the product build carries none of them.
#ifdef GATRES_DIAG_BUILD
#define GATRES_DIAG 1
#else
#define GATRES_DIAG 0
#endif

// Switches of the library, read from the environment ONCE (first use) -- never on a launch path.  A process that changes
// the environment afterwards (the tests do) calls gatres_knobs_reload().
// PRODUCT switches (eleven; each one selects a path that some plan / model takes on its own, so each has a test that forces
// it -- tests named in DESIGN.md section 8):
struct gatres_knobs_t {
  int fused_split;            // GATRES_FUSED_SPLIT = 1 .. 8: workgroups per snapshot (0: automatic)
  int fused_safe_sync;        // GATRES_FUSED_SAFE_SYNC: always agent-scope hand-offs (what parts on different XCDs get)
  int fused_no_halo;          // GATRES_FUSED_NO_HALO: whole-segment kernel, bulk pulls (what an overflowing halo list gets)
  int fused_no_consumers;     // 1 unless GATRES_FUSED_WITH_CONSUMERS=1: parameter gradients as a launch of their own (the default)
  int fused_no_rounds;        // GATRES_FUSED_NO_ROUNDS: 49 .. 96 segments resident at fewer parts instead of two rounds
  int agg_lane_features;      // GATRES_AGG_LANE_FEATURES = 4 | 8 (0: automatic -- 8 for rows of 64 features and more)
  int lin_bwd_wave;           // GATRES_LIN_BWD_WAVE: lin0 / lin1 backward, one wave per slab (what odd widths get)
  int no_proj_lds;            // GATRES_NO_PROJ_LDS: fp32 projections without the LDS copy of W (what small batches get)
  int dw_1d;                  // GATRES_DW_1D: bf16 weight gradients, one matrix per workgroup (what narrow models get)
  int side_stream;            // GATRES_SIDE_STREAM = 0 | 1: the per-op backward's parameter-gradient launches never / always on
                              // the library's side stream (-1, unset: where it was measured faster -- fp32, nc >= 128)
  int blocked;                // GATRES_BLOCKED: wide bf16 models (nc = 128) take the blocked launches of k_blocked.hip (a sparse
                              // stage + the projection behind it as one kernel) where the plan allows; default off: bit-identical
                              // to the per-op pairs, measured 5 - 20 % slower than them (profiles/r05_blocked_probe.txt)
  int window_runtime_phases;  // GATRES_WINDOW_RUNTIME_PHASES: the window kernel reads its phases at run time (one register allocation
                              // for forward and backward) instead of taking the per-phase instantiations
  int window_sync_start;      // GATRES_WINDOW_SYNC_START: a split launch starts with its cross-CU barrier even where the dispatch was probed
  int window_ph_mask;         // GATRES_WINDOW_PH_MASK (default all ones): which compile-time facts a launch may use (k_window.hip)
  // DIAGNOSTIC build only (fixed at the defaults in the product build): measured-and-lost alternatives, tuning sweeps and the
  // switches that give WRONG results
  int agg_wide_offsets;       // GATRES_AGG_WIDE_OFFSETS
  int fused_threads;          // GATRES_FUSED_THREADS = 512 (default 1024)
  int fused_no_window;        // GATRES_FUSED_NO_WINDOW
  int fused_prefer_consumers; // GATRES_FUSED_PREFER_CONSUMERS
  int fused_consumers_cap;    // GATRES_FUSED_CONSUMERS (default 2, at most 4)
  int fused_nocache;          // GATRES_FUSED_NOCACHE
  int fused_wide;             // GATRES_FUSED_WIDE (nc > 32 on the per-snapshot kernels)
  int fused_no_keep;          // GATRES_FUSED_NO_KEEP
  int fused_heartbeat;        // GATRES_FUSED_HEARTBEAT: pace the hand-offs by heartbeat granules even on symmetric plans
  int param_grads_no_stream;  // GATRES_PARAM_GRADS_NO_STREAM
  int proj_rows;              // GATRES_PROJ_ROWS (0: default)
  int proj_stream;            // GATRES_PROJ_STREAM=1: bf16 projections of gatres_large by proj_bf16_stream_kernel
  int dw_fp32;                // GATRES_DW_FP32
  int no_co_launch;           // GATRES_NO_CO_LAUNCH
  int co_launch_always;       // GATRES_CO_LAUNCH_ALWAYS
  int dw_slab_rows;           // GATRES_DW_SLAB_ROWS (0: default)
  int xch_nowait;             // WRONG results: GATRES_XCH_NOWAIT
  int diag_nomask;            // WRONG results: GATRES_DIAG_NOMASK
};
extern "C" __attribute__((visibility("hidden"))) const gatres_knobs_t* gatres_knobs();

// The library's side stream of the current device (created at first use, never destroyed) and the events of the per-op
// backward's fork / join: the parameter-gradient launches of a convolution feed nothing but the optimizer, so they may run
// beside the chain of launches that carries the gradient down the network (model_driver.hip).  nullptr: the runtime refused
// a stream / event.
struct gatres_side_t {
  hipStream_t stream;
  hipEvent_t fork_a, fork_b, done_a, done_b;
  void* mu;              // std::mutex*: one caller at a time enqueues a fork / join sequence on these events
};
extern "C" __attribute__((visibility("hidden"))) gatres_side_t* gatres_side();
extern "C" __attribute__((visibility("hidden"))) gatres_side_t* gatres_side_peek();      // never creates one

static inline int gatres_launch_status() { return (int)hipGetLastError(); }
static inline hipStream_t gatres_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool gatres_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool gatres_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// lanes that cooperate on one feature row: one float4 per lane, rounded up to a power of two
static inline int gatres_row_lanes(int width) {
  int need = (width + 3) / 4, g = 1;
  while (g < need) g <<= 1;
  return g;
}

// beta^t for the Adam bias corrections: square-and-multiply in double (<= 2 * 64 multiplications, a few ulp from
// pow(); every workgroup of the update kernels needs it before it can start, and pow() costs ~1 us there)
__device__ __forceinline__ double gatres_powi(double b, unsigned long long t) {
  double r = 1.0;
  while (t) {
    if (t & 1ULL) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

__device__ __forceinline__ float gatres_leaky(float v) { return v > 0.f ? v : v * GATRES_NEG_SLOPE; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// acc += a * v with ONE rounding per element (fma); used by every neighbour accumulation, fused and per-op alike,
// so both paths produce bit-identical sums.
// Written on two-element vectors so that hipcc emits v_pk_fma_f32 (two IEEE fmas per instruction at the issue rate of one: the
// sparse stages are VALU-issue-bound); element for element the same fmaf as before -- the same bits.
typedef float gatres_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gatres_axpy4(float4& acc, float a, const float4 v) {
#ifdef GATRES_NO_PK_FMA
  acc.x = fmaf(a, v.x, acc.x); acc.y = fmaf(a, v.y, acc.y); acc.z = fmaf(a, v.z, acc.z); acc.w = fmaf(a, v.w, acc.w);
#else
  const gatres_f2 aa = {a, a};
  const gatres_f2 lo = __builtin_elementwise_fma(aa, (gatres_f2){v.x, v.y}, (gatres_f2){acc.x, acc.y});
  const gatres_f2 hi = __builtin_elementwise_fma(aa, (gatres_f2){v.z, v.w}, (gatres_f2){acc.z, acc.w});
  acc.x = lo.x; acc.y = lo.y; acc.z = hi.x; acc.w = hi.y;
#endif
}

// Sum of `d` over the LH adjacent lanes that own one attention head of a row (LH = C/4, a power of two).
// DPP row operations (quad_perm xor-1, xor-2, row_half_mirror, row_mirror) are register-to-register VALU modifiers:
// no LDS round trip, unlike __shfl_xor (ds_bpermute).  Every lane of the group ends with the same value.  Used by
// the fused and the per-op kernels alike so that both associate the sum identically.
template <int CTRL>
__device__ __forceinline__ float gatres_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LH>
__device__ __forceinline__ float gatres_head_reduce(float d) {
  if constexpr (LH >= 2) d += gatres_dpp<0xB1>(d);     // quad_perm [1,0,3,2]
  if constexpr (LH >= 4) d += gatres_dpp<0x4E>(d);     // quad_perm [2,3,0,1]
  if constexpr (LH >= 8) d += gatres_dpp<0x141>(d);    // row_half_mirror: lane i <-> 7 - i
  if constexpr (LH >= 16) d += gatres_dpp<0x140>(d);   // row_mirror:      lane i <-> 15 - i
  if constexpr (LH >= 32) d += __shfl_xor(d, 16);      // beyond one 16-lane DPP row
  return d;
}
__device__ __forceinline__ float gatres_head_dot4(const float4 a, const float4 b) {
  float d = a.x * b.x;
  d = fmaf(a.y, b.y, d);
  d = fmaf(a.z, b.z, d);
  return fmaf(a.w, b.w, d);
}

// ---------------------------------------------------------------------------------------------------- storage types
// Activation-sized tensors are stored as fp32 (GATRES_DTYPE_F32: every kernel's arithmetic is then bit-for-bit the fp32
// statement the parity tests pin) or as bf16 (GATRES_DTYPE_BF16, BASELINE config 3): loads widen to fp32, every sum /
// softmax / accumulator stays fp32, stores round to nearest-even (v_cvt_pk_bf16_f32).  A row segment of four features is
// one 16-byte (fp32) or one 8-byte (bf16) access per lane either way.
typedef __bf16 gatres_bf16;
template <typename T> struct gatres_store;
template <> struct gatres_store<float> { static constexpr int dtype = GATRES_DTYPE_F32; };
template <> struct gatres_store<gatres_bf16> { static constexpr int dtype = GATRES_DTYPE_BF16; };

__device__ __forceinline__ float4 ldrow4(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldrow4(const gatres_bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void strow4(float* p, float4 v) { st4(p, v); }
__device__ __forceinline__ void strow4(gatres_bf16* p, float4 v) {
  typedef gatres_bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 b;
  b[0] = (gatres_bf16)v.x; b[1] = (gatres_bf16)v.y; b[2] = (gatres_bf16)v.z; b[3] = (gatres_bf16)v.w;
  *reinterpret_cast<bf16x4*>(p) = b;
}
// W consecutive features of a row per lane (W = 4, 8 or 16: W / 4 float4s; bf16 rows in 16-byte accesses from W = 8): the
// sparse kernels give a row HC / W lanes, so a larger W means fewer lanes repeating a row's scalar work (k_aggregate.hip).
template <int W> struct gatres_rowv { float4 v[W / 4]; };
template <int W> __device__ __forceinline__ gatres_rowv<W> rowv_zero() {
  gatres_rowv<W> r;
#pragma unroll
  for (int i = 0; i < W / 4; ++i) r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  return r;
}
template <int W> __device__ __forceinline__ gatres_rowv<W> ldrowv(const float* p) {
  gatres_rowv<W> r;
#pragma unroll
  for (int i = 0; i < W / 4; ++i) r.v[i] = ld4(p + 4 * i);
  return r;
}
template <int W> __device__ __forceinline__ gatres_rowv<W> ldrowv(const gatres_bf16* p) {
  gatres_rowv<W> r;
  if constexpr (W % 8 == 0) {                    // 16-byte accesses: eight features each
#pragma unroll
    for (int k = 0; k < W / 8; ++k) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + 8 * k);
      r.v[2 * k] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                               __uint_as_float(u.y & 0xffff0000u));
      r.v[2 * k + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                   __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
    }
  } else {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) r.v[i] = ldrow4(p + 4 * i);
  }
  return r;
}
template <int W> __device__ __forceinline__ void strowv(float* p, const gatres_rowv<W>& r) {
#pragma unroll
  for (int i = 0; i < W / 4; ++i) st4(p + 4 * i, r.v[i]);
}
template <int W> __device__ __forceinline__ void strowv(gatres_bf16* p, const gatres_rowv<W>& r) {
  if constexpr (W % 8 == 0) {
    typedef gatres_bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int k = 0; k < W / 8; ++k) {
      const float4 lo = r.v[2 * k], hi = r.v[2 * k + 1];
      bf16x8 b;
      b[0] = (gatres_bf16)lo.x; b[1] = (gatres_bf16)lo.y; b[2] = (gatres_bf16)lo.z; b[3] = (gatres_bf16)lo.w;
      b[4] = (gatres_bf16)hi.x; b[5] = (gatres_bf16)hi.y; b[6] = (gatres_bf16)hi.z; b[7] = (gatres_bf16)hi.w;
      *reinterpret_cast<bf16x8*>(p + 8 * k) = b;
    }
  } else {
#pragma unroll
    for (int i = 0; i < W / 4; ++i) strow4(p + 4 * i, r.v[i]);
  }
}
__device__ __forceinline__ float ldval(const float* p) { return *p; }
__device__ __forceinline__ float ldval(const gatres_bf16* p) { return (float)*p; }
__device__ __forceinline__ void stval(float* p, float v) { *p = v; }
__device__ __forceinline__ void stval(gatres_bf16* p, float v) { *p = (gatres_bf16)v; }

// Workgroup b of an n-workgroup grid -> the block of work it takes, such that the workgroups of ONE XCD (ids b, b + 8, ...
// under round-robin dispatch: MI355X_MICROARCH.md, Workgroup dispatch) take a CONTIGUOUS range of blocks.  Row-sorted
// sparse kernels gather neighbour rows that lie near their own: with the identity map every XCD's L2 sees the whole table,
// with this one an eighth of it plus a halo.  A bijection for any n; speed only, never correctness.
__device__ __forceinline__ int gatres_xcd_block(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, j = b >> 3;
  return x * q + (x < r ? x : r) + j;
}
