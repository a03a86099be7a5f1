// Sparse stages of the window kernel (k_window.hip), hand-scheduled around the LDS.
//
// Why this file exists (round 3, read off the gfx950 ISA of round 2's kernel): the stage functions of k_fused_dev.h say
// "all neighbour loads of a row are issued together", but behind the wave-uniform `k < dmax` branches and the per-lane
// `k < deg` selects hipcc emitted, per neighbour slot, ds_read_u16 (index) -> s_waitcnt lgkmcnt(0) -> ds_read_b128 (row)
// -> s_waitcnt lgkmcnt(0): twelve dependent LDS round trips per row where the data dependences need three (row
// descriptor -> neighbour ids -> neighbour rows).  Here
//   * every own row has a PADDED NEIGHBOUR DESCRIPTOR in LDS ({first edge, degree, six neighbour ids} = 16 bytes, one
//     ds_read_b128, the lanes of a row read the same address: a broadcast), built once per launch phase, so the chain is
//     descriptor -> rows: TWO round trips;
//   * the loads of one round are one asm statement: issued back to back, one s_waitcnt (cdna_hip_programming.md 5.7,
//     form (i): loads and their wait in ONE statement, early-clobber outputs), which hipcc neither splits nor reorders;
//   * no branch between a stage's loads: slots beyond a row's degree read a valid row and weigh 0 (as before), rows
//     with more than MAXD edges send their wave through the edge-at-a-time functions of k_fused_dev.h.
// Arithmetic, operand order and rounding are exactly those of k_fused_dev.h / the per-op kernels: results stay
// bit-identical (tests/test_gpu_model.py compares the three).
#pragma once
#include "k_fused_dev.h"

namespace {

// The thread id of a stage, opaque to the optimiser: everything a stage derives from it (column group, row of the trip,
// table addresses) is then recomputed inside the stage -- a handful of VALU instructions -- instead of being hoisted out of
// the block loop and kept alive across all thirteen stages of it, which is what ran the 128-VGPR kernel into scratch.
__device__ __forceinline__ int stage_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// ---------------------------------------------------------------------------------------------- LDS primitives
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }

// one 16-byte read
__device__ __forceinline__ uint4 lds_rd128(unsigned a) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
// one 16-byte read + one dword, one wait
__device__ __forceinline__ void lds_rd128_32(unsigned a, unsigned b, uint4& v, float& s) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v), "=&v"(s) : "v"(a), "v"(b) : "memory");
}
// two 16-byte reads, one wait
__device__ __forceinline__ void lds_rd128x2(unsigned a, unsigned b, uint4& v, uint4& w) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v), "=&v"(w) : "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ void lds_rd128x2_32(unsigned a, unsigned b, unsigned c, uint4& v, uint4& w, float& s) {
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v), "=&v"(w), "=&v"(s) : "v"(a), "v"(b), "v"(c) : "memory");
}
// six 16-byte reads, one wait
__device__ __forceinline__ void lds_rd128x6(const unsigned (&a)[MAXD], f32x4 (&v)[MAXD]) {
  static_assert(MAXD == 6, "six slots");
  asm volatile(
      "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\t"
      "ds_read_b128 %3, %9\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5])
      : "memory");
}
// six dwords, one wait
__device__ __forceinline__ void lds_rd32x6(const unsigned (&a)[MAXD], float (&v)[MAXD]) {
  asm volatile(
      "ds_read_b32 %0, %6\n\tds_read_b32 %1, %7\n\tds_read_b32 %2, %8\n\t"
      "ds_read_b32 %3, %9\n\tds_read_b32 %4, %10\n\tds_read_b32 %5, %11\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5])
      : "memory");
}
// six 16-byte reads + six dwords, one wait
__device__ __forceinline__ void lds_rd128x6_32x6(const unsigned (&a)[MAXD], const unsigned (&b)[MAXD], f32x4 (&v)[MAXD],
                                                 float (&s)[MAXD]) {
  asm volatile(
      "ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %14\n\t"
      "ds_read_b128 %3, %15\n\tds_read_b128 %4, %16\n\tds_read_b128 %5, %17\n\t"
      "ds_read_b32 %6, %18\n\tds_read_b32 %7, %19\n\tds_read_b32 %8, %20\n\t"
      "ds_read_b32 %9, %21\n\tds_read_b32 %10, %22\n\tds_read_b32 %11, %23\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(s[0]), "=&v"(s[1]),
        "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
        "v"(b[4]), "v"(b[5])
      : "memory");
}
// six 16-byte reads + twelve dwords, one wait
__device__ __forceinline__ void lds_rd128x6_32x12(const unsigned (&a)[MAXD], const unsigned (&b)[MAXD],
                                                  const unsigned (&c)[MAXD], f32x4 (&v)[MAXD], float (&s)[MAXD],
                                                  float (&t)[MAXD]) {
  asm volatile(
      "ds_read_b128 %0, %18\n\tds_read_b128 %1, %19\n\tds_read_b128 %2, %20\n\t"
      "ds_read_b128 %3, %21\n\tds_read_b128 %4, %22\n\tds_read_b128 %5, %23\n\t"
      "ds_read_b32 %6, %24\n\tds_read_b32 %7, %25\n\tds_read_b32 %8, %26\n\t"
      "ds_read_b32 %9, %27\n\tds_read_b32 %10, %28\n\tds_read_b32 %11, %29\n\t"
      "ds_read_b32 %12, %30\n\tds_read_b32 %13, %31\n\tds_read_b32 %14, %32\n\t"
      "ds_read_b32 %15, %33\n\tds_read_b32 %16, %34\n\tds_read_b32 %17, %35\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(s[0]), "=&v"(s[1]),
        "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]),
        "=&v"(t[4]), "=&v"(t[5])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
        "v"(b[4]), "v"(b[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5])
      : "memory");
}
// eighteen dwords, one wait
__device__ __forceinline__ void lds_rd32x18(const unsigned (&a)[MAXD], const unsigned (&b)[MAXD], const unsigned (&c)[MAXD],
                                            float (&u)[MAXD], float (&s)[MAXD], float (&t)[MAXD]) {
  asm volatile(
      "ds_read_b32 %0, %18\n\tds_read_b32 %1, %19\n\tds_read_b32 %2, %20\n\t"
      "ds_read_b32 %3, %21\n\tds_read_b32 %4, %22\n\tds_read_b32 %5, %23\n\t"
      "ds_read_b32 %6, %24\n\tds_read_b32 %7, %25\n\tds_read_b32 %8, %26\n\t"
      "ds_read_b32 %9, %27\n\tds_read_b32 %10, %28\n\tds_read_b32 %11, %29\n\t"
      "ds_read_b32 %12, %30\n\tds_read_b32 %13, %31\n\tds_read_b32 %14, %32\n\t"
      "ds_read_b32 %15, %33\n\tds_read_b32 %16, %34\n\tds_read_b32 %17, %35\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(s[0]), "=&v"(s[1]),
        "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]),
        "=&v"(t[4]), "=&v"(t[5])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
        "v"(b[4]), "v"(b[5]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5])
      : "memory");
}

__device__ __forceinline__ float4 as_f4(const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 as_f4(const uint4 v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// N 16-byte reads, one wait (N = 4, 6, 8, 12: the operand fragments of one MFMA work unit)
__device__ __forceinline__ void lds_rd128x4(const unsigned (&a)[4], f32x4 (&v)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
}
__device__ __forceinline__ void lds_rd128x8(const unsigned (&a)[8], f32x4 (&v)[8]) {
  asm volatile(
      "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
      "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
      : "memory");
}
__device__ __forceinline__ void lds_rd128x12(const unsigned (&a)[12], f32x4 (&v)[12]) {
  asm volatile(
      "ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %14\n\tds_read_b128 %3, %15\n\t"
      "ds_read_b128 %4, %16\n\tds_read_b128 %5, %17\n\tds_read_b128 %6, %18\n\tds_read_b128 %7, %19\n\t"
      "ds_read_b128 %8, %20\n\tds_read_b128 %9, %21\n\tds_read_b128 %10, %22\n\tds_read_b128 %11, %23\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
        "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
        "v"(a[10]), "v"(a[11])
      : "memory");
}
template <int N> __device__ __forceinline__ void lds_rd128xN(const unsigned (&a)[N], f32x4 (&v)[N]) {
  static_assert(N == 4 || N == 6 || N == 8 || N == 12, "batch sizes of win_proj");
  if constexpr (N == 4) lds_rd128x4(a, v);
  else if constexpr (N == 6) lds_rd128x6(a, v);
  else if constexpr (N == 8) lds_rd128x8(a, v);
  else lds_rd128x12(a, v);
}

// ---------------------------------------------------------------------------------------------- granule exchange, 16-byte form
// The hand-offs of k_fused_dev.h (xch_export / xch_import: one 8-byte {value, epoch} granule per memory instruction, one
// table per call) restated for the window kernel:
//   * TWO granules per access -- {v0, ep, v1, ep} is one 16-byte store / load; each 8-byte half still carries its own
//     tag, so a 16-byte access that the memory system splits is harmless (MI355X_MICROARCH.md: R2's granule, "also for
//     16-B sc1 halves") -- half the memory instructions per exchange;
//   * the row table and the small per-row / per-edge table of an exchange point go through ONE sweep: their polls are in
//     flight together (two calls were two dependent L2 round trips);
//   * accesses are raw-buffer builtins on one descriptor of the segment's exchange region (the compiler counts them:
//     no hand-written waits), loads `sc1` (never served from this CU's L1), stores plain when every part of the segment
//     sits on one XCD (the line stays in that L2) and `sc1` (write-through) otherwise -- as gran_store / gran_load.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct XchBuf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ XchBuf xch_buffer(u64* base, long long granules) {
  XchBuf b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(granules * 8), 0x00020000);
  return b;
}
// one listed row / edge of a table W wide = P accesses of E elements
template <int W> struct XchGeom {
  static constexpr int E = W >= 2 ? 2 : 1, P = W >= 2 ? W / 2 : (W == 1 ? 1 : 0);
  static_assert(W <= 1 || W % 2 == 0, "row width");
};

// tables 1 and 2 (either list may be the same): src: LDS table by the same index; t1 / t2: the tables' first granule
template <int W1, int W2, int THREADS>
__device__ __forceinline__ void xch_export2(const Xch& x, const XchBuf& xb, const u16* l1, int c1, const float* s1, unsigned t1,
                                            const u16* l2, int c2, const float* s2, unsigned t2) {
  constexpr int P1 = XchGeom<W1>::P, P2 = XchGeom<W2>::P, E1 = XchGeom<W1>::E, E2 = XchGeom<W2>::E;
  const int tid = stage_tid();
  const int n1 = c1 * P1, total = n1 + c2 * P2;
  for (int k = tid; k < total; k += THREADS) {
    const bool first = k < n1;
    const int kk = first ? k : k - n1;
    int o, two;
    const float* src;
    unsigned tb;
    if (first) { o = (int)l1[kk / P1] * W1 + E1 * (kk % P1); two = E1 == 2; src = s1; tb = t1; }
    else       { o = (int)l2[kk / (P2 ? P2 : 1)] * W2 + E2 * (kk % (P2 ? P2 : 1)); two = E2 == 2; src = s2; tb = t2; }
    const float v0 = src[o], v1 = two ? src[o + 1] : 0.f;
    const unsigned off = (tb + (unsigned)o) * 8u;
    if (two) {
      const v4u g = {__float_as_uint(v0), x.ep, __float_as_uint(v1), x.ep};
      if (x.local) __builtin_amdgcn_raw_buffer_store_b128(g, xb.r, off, 0, 0);
      else         __builtin_amdgcn_raw_buffer_store_b128(g, xb.r, off, 0, 16);
    } else {
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      const v2u g = {__float_as_uint(v0), x.ep};
      if (x.local) __builtin_amdgcn_raw_buffer_store_b64(g, xb.r, off, 0, 0);
      else         __builtin_amdgcn_raw_buffer_store_b64(g, xb.r, off, 0, 16);
    }
  }
}

// granules of every listed row / edge of both tables -> the LDS tables, re-read until their tags carry this exchange's epoch
template <int W1, int W2, int THREADS>
__device__ __forceinline__ void xch_import2(Xch& x, const XchBuf& xb, const u16* l1, int c1, unsigned t1, float* d1,
                                            const u16* l2, int c2, unsigned t2, float* d2) {
  constexpr int P1 = XchGeom<W1>::P, P2 = XchGeom<W2>::P, E1 = XchGeom<W1>::E, E2 = XchGeom<W2>::E;
  // ONE granule access per lane and trip: the hand-off lists of a part fit one trip of the workgroup's threads (C-Town, 8 parts:
  // 320 - 700 accesses), and a second, always-empty access per lane cost 13 us per step in index arithmetic and predication
  // (0.3555 - 0.3565 vs 0.3678 - 0.3709 ms/step); longer lists take more trips.
  constexpr int U = 1;
  const int tid = stage_tid();
  const int n1 = c1 * P1, total = n1 + c2 * P2;
  for (int base = 0; base < total; base += U * THREADS) {
    if (base + (tid & ~63) >= total) continue;                             // nothing left for this wave
    int o[U];
    bool valid[U], two[U], first[U];
    unsigned off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = base + u * THREADS + tid;
      valid[u] = k < total;
      first[u] = k < n1;
      const int kk = first[u] ? k : k - n1;
      if (!valid[u])     { o[u] = 0; two[u] = false; off[u] = 0; }
      else if (first[u]) { o[u] = (int)l1[kk / P1] * W1 + E1 * (kk % P1); two[u] = E1 == 2; off[u] = (t1 + (unsigned)o[u]) * 8u; }
      else               { o[u] = (int)l2[kk / (P2 ? P2 : 1)] * W2 + E2 * (kk % (P2 ? P2 : 1)); two[u] = E2 == 2; off[u] = (t2 + (unsigned)o[u]) * 8u; }
    }
    v4u v[U];
    int spin = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (valid[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xb.r, off[u], 0, 16);
#pragma unroll
      for (int u = 0; u < U; ++u)
        ok = ok && (!valid[u] || (v[u].y == x.ep && (!two[u] || v[u].w == x.ep)));
      if (__all(ok || x.dead)) break;
      __builtin_amdgcn_s_sleep(1);                 // (a failed poll: 64 cycles before the next one -- 0.8 us per step less L2 pressure)
      if (++spin > SPIN_LIMIT) { *x.err = 1; x.dead = true; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (valid[u]) {
        float* d = first[u] ? d1 : d2;
        d[o[u]] = __uint_as_float(v[u].x);
        if (two[u]) d[o[u] + 1] = __uint_as_float(v[u].z);
      }
  }
}

// The same sweep run by the waves [T0 / 64, (T0 + NT) / 64) only, for the epoch `ep` -- DURING the stage that precedes the
// hand-off, by waves that hold no rows in it (early import): what the partners store early in their own stage is then in LDS
// when the stage's closing barrier falls, and the hand-off behind it has no sweep of its own.
template <int W1, int W2, int T0, int NT>
__device__ __forceinline__ void xch_import2_by(Xch& x, unsigned ep, const XchBuf& xb, const u16* l1, int c1, unsigned t1, float* d1,
                                               const u16* l2, int c2, unsigned t2, float* d2) {
  constexpr int P1 = XchGeom<W1>::P, P2 = XchGeom<W2>::P, E1 = XchGeom<W1>::E, E2 = XchGeom<W2>::E;
  constexpr int U = 1;                         // (see xch_import2)
  const int tid = stage_tid() - T0;
  if (tid < 0 || tid >= NT) return;
  const int n1 = c1 * P1, total = n1 + c2 * P2;
  for (int base = 0; base < total; base += U * NT) {
    if (base + (tid & ~63) >= total) continue;                             // nothing left for this wave
    int o[U];
    bool valid[U], two[U], first[U];
    unsigned off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = base + u * NT + tid;
      valid[u] = k < total;
      first[u] = k < n1;
      const int kk = first[u] ? k : k - n1;
      if (!valid[u])     { o[u] = 0; two[u] = false; off[u] = 0; }
      else if (first[u]) { o[u] = (int)l1[kk / P1] * W1 + E1 * (kk % P1); two[u] = E1 == 2; off[u] = (t1 + (unsigned)o[u]) * 8u; }
      else               { o[u] = (int)l2[kk / (P2 ? P2 : 1)] * W2 + E2 * (kk % (P2 ? P2 : 1)); two[u] = E2 == 2; off[u] = (t2 + (unsigned)o[u]) * 8u; }
    }
    v4u v[U];
    int spin = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (valid[u]) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xb.r, off[u], 0, 16);
#pragma unroll
      for (int u = 0; u < U; ++u)
        ok = ok && (!valid[u] || (v[u].y == ep && (!two[u] || v[u].w == ep)));
      if (__all(ok || x.dead)) break;
      __builtin_amdgcn_s_sleep(1);                 // (a failed poll: 64 cycles before the next one -- 0.8 us per step less L2 pressure)
      if (++spin > SPIN_LIMIT) { *x.err = 1; x.dead = true; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (valid[u]) {
        float* d = first[u] ? d1 : d2;
        d[o[u]] = __uint_as_float(v[u].x);
        if (two[u]) d[o[u] + 1] = __uint_as_float(v[u].z);
      }
  }
}

// After a part's sweep.  pace: announce "exchange ep is behind me" and wait until every other part has announced exchange
// ep - 1 (k_fused_dev.h: xch_heartbeat) -- needed only when the plan cannot promise that partners owe each other rows in
// both directions (GATRES_GRAPH_SYMMETRIC: then a part cannot reach the exchange that rewrites a cell before the cell's
// readers have passed the exchange in between, because that one needs THEIR granules, which they store after their own
// sweep).  drain: every wave waits for its outstanding global stores, so that the barrier that follows publishes them to
// the consumer workgroups (publish_items) -- needed only in launches that have consumers.
template <int THREADS>
__device__ __forceinline__ void xch_after(Xch& x, bool pace, bool drain) {
  if (pace) {
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
  }
  if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// EXPORT FROM THE PRODUCING STAGE.  xch_export2 runs behind the producer's closing barrier and re-reads the rows it
// publishes from LDS: a barrier, an LDS round trip and a list walk in front of every hand-off (~0.3 us, ninety times per
// launch).  The stages that PRODUCE exported values hold them in registers: they store the granules themselves, for the rows
// / edges a per-row flag marks (rows: `flag`, own-row byte table; edges: the source row lies outside the part), tagged with
// the epoch of the hand-off that FOLLOWS (ep).  The partner's sweep then finds them on its first pass more often, and the
// hand-off itself is barrier -> sweep -> barrier.  on == false: the stage exports nothing (xch_export2 does).
struct XOut {
  XchBuf xb;
  unsigned ep;          // epoch of the hand-off these values belong to
  bool local, on;
  const unsigned char* flag;    // own rows (shifted view: flag[row]): some partner imports this row
  unsigned t_rows, t_small;     // first granule of the row table / of the per-row or per-edge table
};
// four consecutive granules (32 bytes) = two 16-byte stores
__device__ __forceinline__ void xout_store4(const XOut& x, unsigned granule, const float4 v) {
  const v4u g0 = {__float_as_uint(v.x), x.ep, __float_as_uint(v.y), x.ep};
  const v4u g1 = {__float_as_uint(v.z), x.ep, __float_as_uint(v.w), x.ep};
  const unsigned off = granule * 8u;
  if (x.local) {
    __builtin_amdgcn_raw_buffer_store_b128(g0, x.xb.r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(g1, x.xb.r, off + 16u, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b128(g0, x.xb.r, off, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(g1, x.xb.r, off + 16u, 0, 16);
  }
}
__device__ __forceinline__ void xout_store1(const XOut& x, unsigned granule, const float v) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u g = {__float_as_uint(v), x.ep};
  if (x.local) __builtin_amdgcn_raw_buffer_store_b64(g, x.xb.r, granule * 8u, 0, 0);
  else         __builtin_amdgcn_raw_buffer_store_b64(g, x.xb.r, granule * 8u, 0, 16);
}

// ---------------------------------------------------------------------------------------------- MFMA stages
// seg_proj (k_fused_dev.h) restated for the window kernel's widths (K, M multiples of 16, W and x operands in LDS):
//   * a WORK UNIT is (16-row tile, group of NTG 16-column tiles): a part has three or four row tiles, so one tile per wave
//     left twelve waves idle behind a chain of 32 dependent-issue MFMAs (~0.45 us); with the column tiles of a row tile on
//     different waves the chain is 8 or 16 (NTG = 1 for the dX stages: no reduction across columns; NTG = M / H / 16 for the
//     forward projections: a wave owns whole heads, so the attention logits reduce inside it, in seg_proj's order);
//   * the unit's operand fragments -- x: K / 16 reads, W: NTG * K / 16 reads of 16 bytes -- are ONE batch (one wait) in
//     front of the MFMA chain; seg_proj's loop had a wait after every 16-byte read of W.
// Same lane map, k order and epilogue arithmetic as seg_proj: bit-identical results.
//   X: [row][K] own rows (LDS, shifted view); wl: W slot [M][K + 4] (+ att_src[M] | att_dst[M] for EPI_ATT);
//   OUT (global, row ob + r) / OUT2 (LDS, shifted view) / OUT3 (LDS [row][M], shifted view); resid_l: LDS [row][M];
//   m64 / m32: relu_bits words per row (fields 16 / 8 bits apart).
//   XS: row stride of X in floats (K + 4 where the table is padded: 16 rows K floats apart share their banks -- the x fragment
//   reads were 8- / 16-way conflicted, profiles/r05_lds_conflicts.txt).
//   MW: the units go round the first MW waves only -- the waves behind them issue the stage's LDS-DMA, and a wave that
//   streams from HBM sits in its issue loop for up to a microsecond (every CU streams at the same moment; a work unit on
//   such a wave would start that much later: measured, 4.0 -> 4.9 us for dX1 + the hand-off behind it).
template <int K, int M, int H, int EPI, int NTG, int MW, int THREADS, int XS = K>
__device__ __forceinline__ void win_proj(Rows rw, const float* X, const float* wl, float* OUT, int ob, float* OUT2,
                                         float* as_g, float* ad_g, float* as_l, float* ad_l, const float* resid_l,
                                         float* OUT3, const unsigned long long* m64, const unsigned* m32, const XOut& xo,
                                         const u16* cnt_rp = nullptr) {
  // cnt_rp (dX1 only): SimpleConv's in-edge rowptr (LDS, shifted view).  OUT2 and the exported granules then carry
  // o / max(indeg(r), 1) -- the form K3 backward gathers (win_bwd_dst<MEAN, PRE>): ONE division per element where it is
  // produced instead of one per out-edge where it is consumed (the same operands: the same bits).
  static_assert(K % 16 == 0 && M % 16 == 0, "whole tiles");
  constexpr int KQ = K / 4, NT = M / 16, GROUPS = NT / NTG, KP = K + 4;
  static_assert(MW >= 1 && MW <= THREADS / 64, "waves that carry work units");
  constexpr int XR = KQ / 4, WR = NTG * KQ / 4;                    // 16-byte reads per unit: x fragment, W fragments
  static_assert(NT % NTG == 0 && (EPI != EPI_ATT || NTG * 16 * H == M || H == 1), "a wave owns whole heads");
  const int tid = stage_tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  if (rw.hi <= rw.lo) return;                                  // (workgroup-uniform) an empty part of a split segment
  const int tlo = rw.lo >> 4, ntiles = (rw.hi + 15) >> 4;      // rw.lo is 16-aligned
  const int units = (ntiles - tlo) * GROUPS;
  const unsigned a_x = lds_addr(X) + (unsigned)(q * KQ) * 4u, a_w = lds_addr(wl) + (unsigned)(i * KP + q * KQ) * 4u;
  if (wave >= MW) return;
  for (int u = wave; u < units; u += MW) {                     // (wave-uniform bounds; `wave` sits in a VGPR: see stage_tid)
    const int t0 = tlo + u / GROUPS, g = u % GROUPS;
    const int r = t0 * 16 + i;
    const bool rok = r < rw.hi;
    const bool xrow = xo.on && rok && xo.flag[min(r, rw.hi - 1)] != 0;      // (read now: the wait falls behind the MFMA chain)
    float rcnt = 1.f;
    if (cnt_rp) { const int rr = min(r, rw.hi - 1); rcnt = (float)max((int)cnt_rp[rr + 1] - (int)cnt_rp[rr], 1); }
    unsigned ad[XR + WR];
    f32x4 fr[XR + WR];
#pragma unroll
#ifdef GATRES_PROBE_XPAD      // (timing probe, WRONG results: the x operand read as if its rows were K + 4 floats apart -- no bank conflicts)
    for (int s = 0; s < XR; ++s) ad[s] = a_x + (unsigned)((min(r, rw.hi - 1) - rw.lo) * (K + 4) + rw.lo * K + 4 * s) * 4u;
#else
    for (int s = 0; s < XR; ++s) ad[s] = a_x + (unsigned)(min(r, rw.hi - 1) * XS + 4 * s) * 4u;      // (XS: the x table's row stride)
#endif
#pragma unroll
    for (int tt = 0; tt < NTG; ++tt)
#pragma unroll
      for (int s = 0; s < XR; ++s) ad[XR + tt * XR + s] = a_w + (unsigned)(((g * NTG + tt) * 16) * KP + 4 * s) * 4u;
    lds_rd128xN<XR + WR>(ad, fr);
    f32x4 acc[NTG];
#pragma unroll
    for (int tt = 0; tt < NTG; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // seg_proj's order: k in steps of SC = 4 (one 16-byte fragment), all column tiles per step
#pragma unroll
    for (int s = 0; s < XR; ++s)
#pragma unroll
      for (int tt = 0; tt < NTG; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[XR + tt * XR + s][e], fr[s][e], acc[tt], 0, 0, 0);
    if constexpr (EPI == EPI_ATT) {
      // attention logits of the head(s) this unit owns: sum over the head's column tiles in order, then over the 4 lane
      // groups (q) by the same two shuffles as seg_proj
      constexpr int C = M / H, TPH = C / 16;                   // column tiles per head
      const float* attS = wl + M * KP;
      const float* attD = attS + M;
      constexpr int HG = (NTG + TPH - 1) / TPH;                // heads per unit (1, or H when the unit spans all columns)
      float ps[HG], pd[HG];
#pragma unroll
      for (int hh = 0; hh < HG; ++hh) { ps[hh] = 0.f; pd[hh] = 0.f; }
#pragma unroll
      for (int tt = 0; tt < NTG; ++tt) {
        const int mb = (g * NTG + tt) * 16 + q * 4;
        const float4 as = ld4(attS + mb), adv = ld4(attD + mb);
        const float ds = fmaf(acc[tt][3], as.w, fmaf(acc[tt][2], as.z, fmaf(acc[tt][1], as.y, acc[tt][0] * as.x)));
        const float dd = fmaf(acc[tt][3], adv.w, fmaf(acc[tt][2], adv.z, fmaf(acc[tt][1], adv.y, acc[tt][0] * adv.x)));
        ps[tt / TPH] += ds; pd[tt / TPH] += dd;
      }
#pragma unroll
      for (int hh = 0; hh < HG; ++hh) {
        ps[hh] += __shfl_xor(ps[hh], 16); ps[hh] += __shfl_xor(ps[hh], 32);
        pd[hh] += __shfl_xor(pd[hh], 16); pd[hh] += __shfl_xor(pd[hh], 32);
      }
      if (q == 0 && rok) {
#pragma unroll
        for (int hh = 0; hh < HG; ++hh) {
          const int hd = (g * NTG) / TPH + hh;
          if (as_g) {
            as_g[(unsigned)(r * H + hd)] = ps[hh];
            ad_g[(unsigned)(r * H + hd)] = pd[hh];
          }
          as_l[r * H + hd] = ps[hh]; ad_l[r * H + hd] = pd[hh];
          if (xrow) xout_store1(xo, xo.t_small + (unsigned)(r * H + hd), ps[hh]);       // a_src goes to the partners
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
      for (int tt = 0; tt < NTG; ++tt) {
        const int mb = (g * NTG + tt) * 16 + q * 4;
        float4 o = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
        if constexpr (EPI == EPI_RESID_MASK) {
          if (resid_l) add4(o, ld4(resid_l + (unsigned)(r * M + mb)));
          if (m64 || m32) {
            const unsigned long long b = mw >> (mb >> 2);
            o.x = (b & 1) ? o.x : 0.f;                 o.y = ((b >> fs) & 1) ? o.y : 0.f;
            o.z = ((b >> (2 * fs)) & 1) ? o.z : 0.f;   o.w = ((b >> (3 * fs)) & 1) ? o.w : 0.f;
          }
        }
        if (OUT) st4(OUT + (unsigned)((ob + r) * M + mb), o);      // (null: an inference launch keeps no saved activations)
        float4 os = o;
        if (cnt_rp) { os.x = o.x / rcnt; os.y = o.y / rcnt; os.z = o.z / rcnt; os.w = o.w / rcnt; }
        if (OUT2) st4(OUT2 + (unsigned)(r * M + mb), os);
        if (OUT3) st4(OUT3 + (unsigned)(r * M + mb), o);
        if (xrow) xout_store4(xo, xo.t_rows + (unsigned)(r * M + mb), os);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- neighbour descriptors
// In-edge descriptor of an own row (GATConv graph or SimpleConv graph), 8 x u16 = one ds_read_b128:
//   {beg, deg, n0 .. n5}: beg = the row's first edge (own-edge-relative index), deg = its in-degree, n_k = local id of the
//   source of edge beg + min(k, deg - 1) (a row of degree 0 -- SimpleConv only -- lists itself).
// Out-edge descriptor (source-major stages), 16 x u16 = two ds_read_b128:
//   {deg, 0, d0 .. d5 | x0 .. x5, 0, 0}: d_k = local id of the destination of out-edge min(k, deg - 1); x_k = the edge's
//   id in the destination-sorted list (GATConv, window-edge-relative) or max(in-degree of d_k, 1) (SimpleConv).
struct NbrIn {
  int beg, deg;
  int n[MAXD];
};
__device__ __forceinline__ NbrIn unpack_in(const uint4 w) {
  NbrIn d;
  d.beg = (int)(w.x & 0xffffu); d.deg = (int)(w.x >> 16);
  d.n[0] = (int)(w.y & 0xffffu); d.n[1] = (int)(w.y >> 16);
  d.n[2] = (int)(w.z & 0xffffu); d.n[3] = (int)(w.z >> 16);
  d.n[4] = (int)(w.w & 0xffffu); d.n[5] = (int)(w.w >> 16);
  return d;
}
struct NbrOut {
  int deg;
  int d[MAXD], x[MAXD];
};
__device__ __forceinline__ NbrOut unpack_out(const uint4 a, const uint4 b) {
  NbrOut o;
  o.deg = (int)(a.x & 0xffffu);
  o.d[0] = (int)(a.y & 0xffffu); o.d[1] = (int)(a.y >> 16);
  o.d[2] = (int)(a.z & 0xffffu); o.d[3] = (int)(a.z >> 16);
  o.d[4] = (int)(a.w & 0xffffu); o.d[5] = (int)(a.w >> 16);
  o.x[0] = (int)(b.x & 0xffffu); o.x[1] = (int)(b.x >> 16);
  o.x[2] = (int)(b.y & 0xffffu); o.x[3] = (int)(b.y >> 16);
  o.x[4] = (int)(b.z & 0xffffu); o.x[5] = (int)(b.z >> 16);
  return o;
}

// Build the in-edge descriptors of the own rows from the 16-bit CSR copies (rp: own rows, shifted view; col: own edges).
// self_pad: a slot beyond the degree names the row itself (SimpleConv) instead of the row's last neighbour (GATConv).
template <int THREADS>
__device__ __forceinline__ void build_nbr_in(Rows rw, const u16* rp, const u16* col, int ne, bool self_pad, u16* tab) {
  const int elast = max(ne - 1, 0);
  for (int r = rw.lo + (int)threadIdx.x; r < rw.hi; r += THREADS) {
    const int beg = rp[r], deg = (int)rp[r + 1] - beg;
    u16* t = tab + (r - rw.lo) * 8;
    t[0] = (u16)beg; t[1] = (u16)deg;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      int j;
      if (self_pad) j = k < deg ? (int)col[min(beg + k, elast)] : r;
      else          j = col[min(beg + min(k, max(deg - 1, 0)), elast)];
      t[2 + k] = (u16)j;
    }
  }
}
// Out-edge descriptors.  trp: own rows (shifted view), dst / eid: own out-edges.  cnt_rp: SimpleConv's in-edge rowptr over
// the WINDOW (shifted view) when x_k is the destination's in-degree, nullptr when x_k is eid[.] - eid_sub.
template <int THREADS>
__device__ __forceinline__ void build_nbr_out(Rows rw, const u16* trp, const u16* dst, const u16* eid, int eid_sub,
                                              const u16* cnt_rp, int ne, u16* tab) {
  const int elast = max(ne - 1, 0);
  for (int r = rw.lo + (int)threadIdx.x; r < rw.hi; r += THREADS) {
    const int beg = trp[r], deg = (int)trp[r + 1] - beg;
    u16* t = tab + (r - rw.lo) * 16;
    t[0] = (u16)deg; t[1] = 0; t[14] = 0; t[15] = 0;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const int kk = min(beg + min(k, max(deg - 1, 0)), elast);
      const int ii = (cnt_rp && !(k < deg)) ? r : (int)dst[kk];
      t[2 + k] = (u16)ii;
      t[8 + k] = cnt_rp ? (u16)max((int)cnt_rp[ii + 1] - (int)cnt_rp[ii], 1) : (u16)((int)eid[kk] - eid_sub);
    }
  }
}

// does any lane of this wave hold a row with more than MAXD edges?  (wave-uniform)
__device__ __forceinline__ bool wave_has_hub(int deg) { return __ballot(deg > MAXD) != 0ull; }
// NH: the plan promises that no row of any CSR has more than MAXD entries (GATRES_GRAPH_DEG_LE6) -- the edge-at-a-time
// paths of the stages are not compiled
template <bool NH> __device__ __forceinline__ bool wave_has_hub_t(int deg) {
  if constexpr (NH) return false;
  else return wave_has_hub(deg);
}

// ---------------------------------------------------------------------------------------------- forward stages
// K2 forward, sub-stage A (seg_softmax): one thread per (row, head).
//   nb: in-edge descriptors; asrc: [row][H] over the window (shifted view); adst: [row][H] own rows (shifted view);
//   alpha_g: saved table (global, [elo + e][H]); alpha_l: LDS table [e][H] over the own edges.
template <int H, int THREADS>
__device__ __forceinline__ void win_softmax(Rows rw, const u16* nb, const u16* rp, const u16* col, const float* asrc,
                                            const float* adst_t, float* __restrict__ alpha_g, int eb, float* alpha_l) {
  const int tid = stage_tid();
  const unsigned a_nb = lds_addr(nb), a_as = lds_addr(asrc), a_ad = lds_addr(adst_t);
  for (int idx0 = rw.lo * H; idx0 < rw.hi * H; idx0 += THREADS) {
    if (idx0 + (int)(tid & ~63u) >= rw.hi * H) continue;             // (wave-uniform) nothing left for this wave
    const int idx = min(idx0 + tid, rw.hi * H - 1);
    const bool valid = idx0 + tid < rw.hi * H;
    const int r = idx / H, hd = idx % H;
    uint4 w;
    float adst;
    lds_rd128_32(a_nb + (unsigned)(r - rw.lo) * 16u, a_ad + (unsigned)(r * H + hd) * 4u, w, adst);
    const NbrIn d = unpack_in(w);
    if (__builtin_expect(wave_has_hub(d.deg), 0)) {
      if (valid) {
        Rows one; one.lo = r; one.hi = r + 1;      // (per lane: the edge-at-a-time loops of seg_softmax for this row / head)
        const int beg = rp[r], end = rp[r + 1];
        auto at = [&](int e) -> float { return asrc[(unsigned)((int)col[e] * H + hd)]; };
        float m = -INFINITY;
        for (int e = beg; e < end; ++e) m = fmaxf(m, gatres_leaky(at(e) + adst));
        float Z = 0.f;
        for (int e = beg; e < end; ++e) Z = Z + expf(gatres_leaky(at(e) + adst) - m);
        Z = Z + GATRES_SOFTMAX_EPS;
        for (int e = beg; e < end; ++e) {
          const float al = expf(gatres_leaky(at(e) + adst) - m) / Z;
          alpha_g[(unsigned)((eb + e) * H + hd)] = al;
          alpha_l[e * H + hd] = al;
        }
        (void)one;
      }
      continue;
    }
    unsigned a[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) a[k] = a_as + (unsigned)(d.n[k] * H + hd) * 4u;
    float so[MAXD];
    lds_rd32x6(a, so);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const float sv = gatres_leaky(so[k] + adst);
      so[k] = k < d.deg ? sv : -INFINITY;
      m = fmaxf(m, so[k]);
    }
    float Z = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      so[k] = expf(so[k] - m);                                   // exp(-inf) = 0 on padding slots
      Z = Z + so[k];
    }
    Z = Z + GATRES_SOFTMAX_EPS;
    if (valid) {
#pragma unroll
      for (int k = 0; k < MAXD; ++k)
        if (k < d.deg) {
          const float al = so[k] / Z;
          alpha_g[(unsigned)((eb + d.beg + k) * H + hd)] = al;
          alpha_l[(d.beg + k) * H + hd] = al;
        }
    }
  }
}

// K2 forward, sub-stage B (seg_gather): out[r] = sum_e alpha_e h[src(e)] + bias (+ReLU); HC/4 lanes per row.
//   hsrc: [row][HC] over the window (shifted view); alpha: LDS table [e][H] over the own edges.
template <bool RELU, int H, int C, int THREADS>
__device__ __forceinline__ void win_gather(Rows rw, const u16* nb, const u16* rp, const u16* col, const float* hsrc,
                                           const float* alpha, const float* bias, float* out, int ob, float* out_pub,
                                           unsigned long long* mask64) {
  const int tid = stage_tid();
  constexpr int HC = H * C, G = HC / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const int hd = c0 / C;
  const unsigned a_nb = lds_addr(nb), a_h = lds_addr(hsrc) + (unsigned)c0 * 4u, a_al = lds_addr(alpha) + (unsigned)hd * 4u;
  const float4 b = ld4(bias + c0);
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;      // (wave-uniform) no row of this trip for this wave
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;
    if (!valid) r = rw.hi - 1;
    const NbrIn d = unpack_in(lds_rd128(a_nb + (unsigned)(r - rw.lo) * 16u));
    float4 acc = f4zero();
    if (__builtin_expect(wave_has_hub(d.deg), 0)) {
      const int beg = rp[r], end = rp[r + 1];
      for (int e = beg; e < end; ++e)
        gatres_axpy4(acc, alpha[(unsigned)(e * H + hd)], ld4(hsrc + (unsigned)((int)col[e] * HC + c0)));
    } else {
      unsigned av[MAXD], aa[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        av[k] = a_h + (unsigned)(d.n[k] * HC) * 4u;
        aa[k] = a_al + (unsigned)((d.beg + min(k, d.deg - 1)) * H) * 4u;
      }
      f32x4 v[MAXD];
      float al[MAXD];
      lds_rd128x6_32x6(av, aa, v, al);
#pragma unroll
      for (int k = 0; k < MAXD; ++k) gatres_axpy4(acc, k < d.deg ? al[k] : 0.f, as_f4(v[k]));     // 0 on padding
    }
    add4(acc, b);
    if (RELU) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
      acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    if (valid) {
      st4(out + (unsigned)((ob + r) * HC + c0), acc);
      if (out_pub) st4(out_pub + (unsigned)(r * HC + c0), acc);
    }
    if constexpr (RELU && G <= 16) {
      if (mask64) {                                             // (workgroup-uniform)
        const unsigned long long w = relu_bits<G, 16>(acc);
        if (valid && tid % G == 0) mask64[r] = w;
      }
    }
  }
}

// K2 forward, BOTH sub-stages in one (win_softmax + win_gather without the barrier, the stage entry and the alpha table
// between them), for heads of 32 channels = 8 lanes: lane j of a head's lane group owns edge slot j (slots 6 and 7 idle).
// It forms the slot's logit and ONE exp; the maximum runs over the group by DPP (exact in any order); the six exponentials
// go round the group (ds_swizzle: the LDS crossbar, no memory) so that every lane adds them up in CSR order -- the same
// ((((e0 + e1) + e2) + e3) + e4) + e5 as the thread-per-head form, bit for bit -- ; ONE divide per lane, the coefficients
// go round once more, and every lane weighs its four columns of the six neighbour rows it loaded at the start.
// (All lanes of a head recomputing all six coefficients was measured in rounds 1 and 2: slower than two stages.)
//   alpha_l: LDS table [e][H], used only by waves that hold a row with more than MAXD edges.
template <int K> __device__ __forceinline__ float group8_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x18 | (K << 5)));      // lane (l & 0x18) | K
}
template <bool RELU, int H, int C, int THREADS, bool NH = false, int OPS = H * C>
__device__ __forceinline__ void win_fwd_agg(Rows rw, const u16* nb, const u16* rp, const u16* col, const float* hsrc,
                                            const float* asrc, const float* adst_t, float* __restrict__ alpha_g, int eb,
                                            float* alpha_l, const float* bias, float* out, int ob, float* out_pub,
                                            unsigned long long* mask64, const XOut& xo) {
  static_assert(C == 32 && MAXD <= 8, "one edge slot per lane of a head's eight lanes");
  const int tid = stage_tid();
  constexpr int HC = H * C, G = HC / 4, LH = C / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const int hd = c0 / C;
  const int j = tid % LH;                                            // this lane's edge slot
  const unsigned a_nb = lds_addr(nb), a_h = lds_addr(hsrc) + (unsigned)c0 * 4u, a_as = lds_addr(asrc) + (unsigned)hd * 4u,
                 a_ad = lds_addr(adst_t) + (unsigned)hd * 4u;
  const float4 b = ld4(bias + c0);
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;      // no row of this trip for this wave
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;
    if (!valid) r = rw.hi - 1;
    uint4 w;
    float adst;
    lds_rd128_32(a_nb + (unsigned)(r - rw.lo) * 16u, a_ad + (unsigned)(r * H) * 4u, w, adst);
    const NbrIn d = unpack_in(w);
    float4 acc = f4zero();
#ifdef GATRES_VALU_PROBE      // experiment builds only (profiles/r04_valu_probe.txt): N extra VALU instructions per lane and call
#ifdef GATRES_VALU_PROBE_H
    if constexpr (H == GATRES_VALU_PROBE_H)
#endif
#ifdef GATRES_VALU_PROBE_ILP
    asm volatile(".rept " GATRES_VALU_PROBE "\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\t.endr" : "+v"(acc.x), "+v"(acc.y));
#else
    asm volatile(".rept " GATRES_VALU_PROBE "\n\tv_mov_b32 %0, %0\n\t.endr" : "+v"(acc.x));
#endif
#endif
    if (__builtin_expect(wave_has_hub_t<NH>(d.deg), 0)) {
      // edge at a time: the head's first lane forms the coefficients (seg_softmax's loops), through the LDS table
      const int beg = rp[r], end = rp[r + 1];
      if (c0 % C == 0) {
        auto at = [&](int e) -> float { return asrc[(unsigned)((int)col[e] * H + hd)]; };
        float m = -INFINITY;
        for (int e = beg; e < end; ++e) m = fmaxf(m, gatres_leaky(at(e) + adst));
        float Z = 0.f;
        for (int e = beg; e < end; ++e) Z = Z + expf(gatres_leaky(at(e) + adst) - m);
        Z = Z + GATRES_SOFTMAX_EPS;
        for (int e = beg; e < end; ++e) {
          const float al = expf(gatres_leaky(at(e) + adst) - m) / Z;
          if (alpha_g && valid) alpha_g[(unsigned)((eb + e) * H + hd)] = al;
          alpha_l[e * H + hd] = al;
        }
      }
      for (int e = beg; e < end; ++e)
        gatres_axpy4(acc, alpha_l[(unsigned)(e * H + hd)], ld4(hsrc + (unsigned)((int)col[e] * HC + c0)));
    } else {
      // this lane's slot: source row n_j (slots 6, 7: n_5, never used)
      const int jj = min(j, MAXD - 1);
      const unsigned wsel = jj < 2 ? w.y : (jj < 4 ? w.z : w.w);
      const int nj = (int)((jj & 1) ? (wsel >> 16) : (wsel & 0xffffu));
      unsigned av[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) av[k] = a_h + (unsigned)(d.n[k] * HC) * 4u;
      f32x4 v[MAXD];
      float asj;
      asm volatile(
          "ds_read_b128 %0, %7\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %9\n\t"
          "ds_read_b128 %3, %10\n\tds_read_b128 %4, %11\n\tds_read_b128 %5, %12\n\tds_read_b32 %6, %13\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(asj)
          : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(a_as + (unsigned)(nj * H) * 4u)
          : "memory");
      const float sv = gatres_leaky(asj + adst);
      const float so = j < d.deg ? sv : -INFINITY;
      float m = so;
      m = fmaxf(m, gatres_dpp<0xB1>(m));       // quad_perm [1,0,3,2]
      m = fmaxf(m, gatres_dpp<0x4E>(m));       // quad_perm [2,3,0,1]
      m = fmaxf(m, gatres_dpp<0x141>(m));      // row_half_mirror
      const float ex = expf(so - m);           // exp(-inf) = 0 on padding slots
      float e[MAXD];
      e[0] = group8_bcast<0>(ex); e[1] = group8_bcast<1>(ex); e[2] = group8_bcast<2>(ex);
      e[3] = group8_bcast<3>(ex); e[4] = group8_bcast<4>(ex); e[5] = group8_bcast<5>(ex);
      float Z = 0.f;
#pragma unroll
      for (int k = 0; k < MAXD; ++k) Z = Z + e[k];
      Z = Z + GATRES_SOFTMAX_EPS;
      const float alj = ex / Z;
      if (alpha_g && valid && j < d.deg) alpha_g[(unsigned)((eb + d.beg + j) * H + hd)] = alj;
      float al[MAXD];
      al[0] = group8_bcast<0>(alj); al[1] = group8_bcast<1>(alj); al[2] = group8_bcast<2>(alj);
      al[3] = group8_bcast<3>(alj); al[4] = group8_bcast<4>(alj); al[5] = group8_bcast<5>(alj);
#pragma unroll
      for (int k = 0; k < MAXD; ++k) gatres_axpy4(acc, k < d.deg ? al[k] : 0.f, as_f4(v[k]));     // 0 on padding
    }
    add4(acc, b);
    if (RELU) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
      acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    if (valid) {
      if (out) st4(out + (unsigned)((ob + r) * HC + c0), acc);
      if (out_pub) st4(out_pub + (unsigned)(r * OPS + c0), acc);      // (OPS: row stride of the x operand table it fills)
      if (xo.on && xo.flag[r] != 0) xout_store4(xo, xo.t_rows + (unsigned)(r * HC + c0), acc);
    }
    if constexpr (RELU && G <= 16) {
      if (mask64) {                                             // (workgroup-uniform)
        const unsigned long long mw = relu_bits<G, 16>(acc);
        if (valid && tid % G == 0) mask64[r] = mw;
      }
    }
  }
}

// K3 forward (seg_mean_fwd): out = relu(mean_{j->r} y[j] + x0[r]).
template <int C, int THREADS, bool NH = false, int XS = C>
__device__ __forceinline__ void win_mean_fwd(Rows rw, const u16* mb, const u16* mrp, const u16* mcol, const float* y,
                                             const float* x0, float* out, float* out2, unsigned* mask32) {
  const int tid = stage_tid();
  constexpr int G = C / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const unsigned a_mb = lds_addr(mb), a_y = lds_addr(y) + (unsigned)c0 * 4u, a_x = lds_addr(x0) + (unsigned)c0 * 4u;
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;
    if (!valid) r = rw.hi - 1;
    uint4 w, xr;
    lds_rd128x2(a_mb + (unsigned)(r - rw.lo) * 16u, a_x + (unsigned)(r * XS) * 4u, w, xr);      // (x0 / out2: the x table, row stride XS)
    const NbrIn d = unpack_in(w);
    const float4 rr = as_f4(xr);
    float4 acc = f4zero();
    if (__builtin_expect(wave_has_hub_t<NH>(d.deg), 0)) {
      const int beg = mrp[r], end = mrp[r + 1];
      for (int e = beg; e < end; ++e) add4(acc, ld4(y + (unsigned)((int)mcol[e] * C + c0)));
    } else {
      unsigned av[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) av[k] = a_y + (unsigned)(d.n[k] * C) * 4u;
      f32x4 v[MAXD];
      lds_rd128x6(av, v);
#pragma unroll
      for (int k = 0; k < MAXD; ++k) gatres_axpy4(acc, k < d.deg ? 1.f : 0.f, as_f4(v[k]));   // fma(1,v,acc) = acc+v
    }
    const float cnt = (float)max(d.deg, 1);
    float4 o;
    o.x = fmaxf(acc.x / cnt + rr.x, 0.f); o.y = fmaxf(acc.y / cnt + rr.y, 0.f);
    o.z = fmaxf(acc.z / cnt + rr.z, 0.f); o.w = fmaxf(acc.w / cnt + rr.w, 0.f);
    if (valid) {
      if (out) st4(out + (unsigned)(r * C + c0), o);
      st4(out2 + (unsigned)(r * XS + c0), o);
    }
    if constexpr (G <= 8) {
      if (mask32) {                                             // (workgroup-uniform)
        const unsigned mw = (unsigned)relu_bits<G, 8>(o);
        if (valid && tid % G == 0) mask32[r] = mw;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward stages
// K3 backward (seg_mean_bwd): g_y[r] = sum over out-edges (r -> i) of g_pre[i] / max(indeg(i), 1)
template <int C, int THREADS>
__device__ __forceinline__ void win_mean_bwd(Rows rw, const u16* mo, const u16* mrp, const u16* mtrp, const u16* mtdst,
                                             const float* g_pre, float* g_y) {
  const int tid = stage_tid();
  constexpr int G = C / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const unsigned a_mo = lds_addr(mo), a_g = lds_addr(g_pre) + (unsigned)c0 * 4u;
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;
    if (!valid) r = rw.hi - 1;
    uint4 wa, wb;
    lds_rd128x2(a_mo + (unsigned)(r - rw.lo) * 32u, a_mo + (unsigned)(r - rw.lo) * 32u + 16u, wa, wb);
    const NbrOut d = unpack_out(wa, wb);
    float4 acc = f4zero();
    if (__builtin_expect(wave_has_hub(d.deg), 0)) {
      const int beg = mtrp[r], end = mtrp[r + 1];
      for (int t = beg; t < end; ++t) {
        const int ii = mtdst[t];
        const float cnt = (float)max((int)mrp[ii + 1] - (int)mrp[ii], 1);
        const float4 v = ld4(g_pre + (unsigned)(ii * C + c0));
        acc.x = acc.x + v.x / cnt; acc.y = acc.y + v.y / cnt;
        acc.z = acc.z + v.z / cnt; acc.w = acc.w + v.w / cnt;
      }
    } else {
      unsigned av[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) av[k] = a_g + (unsigned)(d.d[k] * C) * 4u;
      f32x4 v[MAXD];
      lds_rd128x6(av, v);
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        const float w = k < d.deg ? 1.f : 0.f;                      // x + 0 * q == x exactly
        const float cnt = (float)d.x[k];
        acc.x = fmaf(w, v[k][0] / cnt, acc.x); acc.y = fmaf(w, v[k][1] / cnt, acc.y);
        acc.z = fmaf(w, v[k][2] / cnt, acc.z); acc.w = fmaf(w, v[k][3] / cnt, acc.w);
      }
    }
    if (valid) st4(g_y + (unsigned)(r * C + c0), acc);
  }
}

// K2 backward, destination-major sub-stage A (seg_edge_dots): ga_e = <g_out[i,h,:], h[j,h,:]> for every in-edge.
//   g_out: [row][HC] own rows; h: [row][HC] over the window; g_e: LDS [e][H] over the own edges.
template <int H, int C, int THREADS>
__device__ __forceinline__ void win_edge_dots(Rows rw, const u16* nb, const u16* rp, const u16* col, const float* g_out,
                                              const float* h, float* g_e) {
  const int tid = stage_tid();
  constexpr int HC = H * C, G = HC / 4, LH = C / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const int hd = c0 / C;
  const unsigned a_nb = lds_addr(nb), a_go = lds_addr(g_out) + (unsigned)c0 * 4u, a_h = lds_addr(h) + (unsigned)c0 * 4u;
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;          // every lane stays in the loop: the head reduction spans the head's lanes
    if (!valid) r = rw.hi - 1;
    const bool leader = valid && (c0 % C) == 0;
    uint4 w, gw;
    lds_rd128x2(a_nb + (unsigned)(r - rw.lo) * 16u, a_go + (unsigned)(r * HC) * 4u, w, gw);
    const NbrIn d = unpack_in(w);
    const float4 go = as_f4(gw);
    if (__builtin_expect(wave_has_hub(d.deg), 0)) {
      const int beg = rp[r], end = rp[r + 1];
      for (int e = beg; e < end; ++e) {
        const float ga = gatres_head_reduce<LH>(gatres_head_dot4(go, ld4(h + (unsigned)((int)col[e] * HC + c0))));
        if (leader) g_e[(unsigned)(e * H + hd)] = ga;
      }
      continue;
    }
    unsigned av[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) av[k] = a_h + (unsigned)(d.n[k] * HC) * 4u;
    f32x4 hv[MAXD];
    lds_rd128x6(av, hv);
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const float ga = gatres_head_reduce<LH>(gatres_head_dot4(go, as_f4(hv[k])));
      if (leader && k < d.deg) g_e[(unsigned)((d.beg + k) * H + hd)] = ga;
    }
  }
}

// K2 backward, destination-major sub-stage B (seg_softmax_bwd): one thread per (row, head).
//   alpha / g_e: LDS [e][H] over the own edges; a_src: [row][H] over the window; a_dst / g_a_dst: [row][H] own rows.
template <int H, int THREADS>
__device__ __forceinline__ void win_softmax_bwd(Rows rw, const u16* nb, const u16* rp, const u16* col, const float* alpha,
                                                const float* a_src, const float* a_dst, float* g_e, float* g_a_dst) {
  const int tid = stage_tid();
  const unsigned a_nb = lds_addr(nb), a_al = lds_addr(alpha), a_as = lds_addr(a_src), a_ad = lds_addr(a_dst),
                 a_ge = lds_addr(g_e);
  for (int idx0 = rw.lo * H; idx0 < rw.hi * H; idx0 += THREADS) {
    if (idx0 + (int)(tid & ~63u) >= rw.hi * H) continue;
    const int idx = min(idx0 + tid, rw.hi * H - 1);
    const bool valid = idx0 + tid < rw.hi * H;
    const int r = idx / H, hd = idx % H;
    uint4 w;
    float adst;
    lds_rd128_32(a_nb + (unsigned)(r - rw.lo) * 16u, a_ad + (unsigned)(r * H + hd) * 4u, w, adst);
    const NbrIn d = unpack_in(w);
    float S = 0.f, gad = 0.f;
    if (__builtin_expect(wave_has_hub(d.deg), 0)) {
      if (valid) {
        const int beg = rp[r], end = rp[r + 1];
        for (int e = beg; e < end; ++e) S = fmaf(alpha[(unsigned)(e * H + hd)], g_e[(unsigned)(e * H + hd)], S);
        for (int e = beg; e < end; ++e) {
          const float gs = alpha[(unsigned)(e * H + hd)] * (g_e[(unsigned)(e * H + hd)] - S);
          const float raw = a_src[(unsigned)((int)col[e] * H + hd)] + adst;
          const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
          g_e[(unsigned)(e * H + hd)] = ge;
          gad = gad + ge;
        }
        g_a_dst[(unsigned)(r * H + hd)] = gad;
      }
      continue;
    }
    unsigned aa[MAXD], ag[MAXD], as[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const unsigned e = (unsigned)((d.beg + min(k, d.deg - 1)) * H + hd) * 4u;
      aa[k] = a_al + e;
      ag[k] = a_ge + e;
      as[k] = a_as + (unsigned)(d.n[k] * H + hd) * 4u;
    }
    float al[MAXD], ga[MAXD], raw[MAXD];
    lds_rd32x18(aa, ag, as, al, ga, raw);
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      raw[k] = raw[k] + adst;
      al[k] = k < d.deg ? al[k] : 0.f;                                   // padding slots weigh nothing
      S = fmaf(al[k], ga[k], S);
    }
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const float gs = al[k] * (ga[k] - S);
      const float ge = raw[k] > 0.f ? gs : gs * GATRES_NEG_SLOPE;
      if (valid && k < d.deg) g_e[(unsigned)((d.beg + k) * H + hd)] = ge;
      gad = gad + ge;                                                  // ge == 0 on padding slots
    }
    if (valid) g_a_dst[(unsigned)(r * H + hd)] = gad;
  }
}

// K2 backward, destination-major, BOTH sub-stages in one (win_edge_dots + win_softmax_bwd without the barrier and the LDS
// hop between them): after the head reduction every lane of a head holds all of the row's edge dots, so the lanes compute
// S, g_e and g_a_dst redundantly (no exp, no divide: ~40 VALU instructions) and the head's first lane stores them.
// MEAN (conv2 only, H == 1): the stage starts with K3 backward (win_mean_bwd) for the same row -- its result g_y2[r] is
// the g_out operand of the edge dots, held by the same lanes -- so  B1 exchange -> [K3 bwd, edge dots, softmax bwd] -> B2
// exchange  is ONE stage.  g_out_rw: [row][HC] over the window (shifted view): read (MEAN = false) or written (MEAN).
template <bool MEAN, int H, int C, int THREADS, bool PRE = false, bool NH = false>
__device__ __forceinline__ void win_bwd_dst(Rows rw, const u16* nb, const u16* rp, const u16* col, float* g_out_rw,
                                            const float* h, const float* alpha, const float* a_src, const float* a_dst,
                                            float* g_e, float* g_a_dst,
                                            const u16* mo, const u16* mrp, const u16* mtrp, const u16* mtdst,
                                            const float* g_pre, const XOut& xo, int eabs) {
  // (eabs: absolute local id of the part's first own edge -- the edge tables of the hand-offs are indexed by it)
  static_assert(!MEAN || H == 1, "K3 runs at conv2's width");
  const int tid = stage_tid();
  constexpr int HC = H * C, G = HC / 4, LH = C / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const int hd = c0 / C;
  const unsigned a_nb = lds_addr(nb), a_go = lds_addr(g_out_rw) + (unsigned)c0 * 4u, a_h = lds_addr(h) + (unsigned)c0 * 4u,
                 a_al = lds_addr(alpha) + (unsigned)hd * 4u, a_as = lds_addr(a_src) + (unsigned)hd * 4u,
                 a_ad = lds_addr(a_dst) + (unsigned)hd * 4u;
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;          // every lane stays in the loop: the head reduction spans the head's lanes
    if (!valid) r = rw.hi - 1;
    const bool leader = valid && (c0 % C) == 0;
    float4 go;
    NbrIn d;
    float adst;
    if constexpr (MEAN) {
      const unsigned a_mo = lds_addr(mo) + (unsigned)(r - rw.lo) * 32u;
      uint4 wa, wb, w;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %5\n\tds_read_b32 %3, %6\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(wa), "=&v"(wb), "=&v"(w), "=&v"(adst)
                   : "v"(a_mo), "v"(a_nb + (unsigned)(r - rw.lo) * 16u), "v"(a_ad + (unsigned)(r * H) * 4u)
                   : "memory");
      const NbrOut od = unpack_out(wa, wb);
      d = unpack_in(w);
      float4 acc = f4zero();
      if (__builtin_expect(wave_has_hub_t<NH>(od.deg), 0)) {
        const int beg = mtrp[r], end = mtrp[r + 1];
        for (int t = beg; t < end; ++t) {
          const int ii = mtdst[t];
          [[maybe_unused]] const float cnt = (float)max((int)mrp[ii + 1] - (int)mrp[ii], 1);
          const float4 v = ld4(g_pre + (unsigned)(ii * C + c0));
          if constexpr (PRE) { add4(acc, v); }
          else {
            acc.x = acc.x + v.x / cnt; acc.y = acc.y + v.y / cnt;
            acc.z = acc.z + v.z / cnt; acc.w = acc.w + v.w / cnt;
          }
        }
      } else {
        const unsigned a_g = lds_addr(g_pre) + (unsigned)c0 * 4u;
        unsigned av[MAXD];
#pragma unroll
        for (int k = 0; k < MAXD; ++k) av[k] = a_g + (unsigned)(od.d[k] * C) * 4u;
        f32x4 v[MAXD];
        lds_rd128x6(av, v);
#pragma unroll
        for (int k = 0; k < MAXD; ++k) {
          const float w1 = k < od.deg ? 1.f : 0.f;                      // x + 0 * q == x exactly
          if constexpr (PRE) {                                          // g_pre holds g / max(indeg, 1): win_proj's cnt_rp
            gatres_axpy4(acc, w1, as_f4(v[k]));
          } else {
            const float cnt = (float)od.x[k];
            acc.x = fmaf(w1, v[k][0] / cnt, acc.x); acc.y = fmaf(w1, v[k][1] / cnt, acc.y);
            acc.z = fmaf(w1, v[k][2] / cnt, acc.z); acc.w = fmaf(w1, v[k][3] / cnt, acc.w);
          }
        }
      }
      go = acc;
      if (valid) {
        st4(g_out_rw + (unsigned)(r * HC + c0), acc);
        if (xo.on && xo.flag[r] != 0) xout_store4(xo, xo.t_rows + (unsigned)(r * HC + c0), acc);
      }
    } else {
      uint4 w, gw;
      lds_rd128x2_32(a_nb + (unsigned)(r - rw.lo) * 16u, a_go + (unsigned)(r * HC) * 4u, a_ad + (unsigned)(r * H) * 4u, w,
                     gw, adst);
      d = unpack_in(w);
      go = as_f4(gw);
    }
    if (__builtin_expect(wave_has_hub_t<NH>(d.deg), 0)) {
      // edge at a time: the dots of the row through LDS (written and read by lanes of this wave, in program order)
      const int beg = rp[r], end = rp[r + 1];
      for (int e = beg; e < end; ++e) {
        const float ga = gatres_head_reduce<LH>(gatres_head_dot4(go, ld4(h + (unsigned)((int)col[e] * HC + c0))));
        if (leader) g_e[(unsigned)(e * H + hd)] = ga;
      }
      if (leader) {
        float S = 0.f, gad = 0.f;
        for (int e = beg; e < end; ++e) S = fmaf(alpha[(unsigned)(e * H + hd)], g_e[(unsigned)(e * H + hd)], S);
        for (int e = beg; e < end; ++e) {
          const float gs = alpha[(unsigned)(e * H + hd)] * (g_e[(unsigned)(e * H + hd)] - S);
          const float raw = a_src[(unsigned)((int)col[e] * H + hd)] + adst;
          const float ge = raw > 0.f ? gs : gs * GATRES_NEG_SLOPE;
          g_e[(unsigned)(e * H + hd)] = ge;
          const int j = col[e];
          if (xo.on && (j < rw.lo || j >= rw.hi)) xout_store1(xo, xo.t_small + (unsigned)((eabs + e) * H + hd), ge);
          gad = gad + ge;
        }
        g_a_dst[(unsigned)(r * H + hd)] = gad;
      }
      continue;
    }
    if constexpr (LH == 8) {
      // One edge slot per lane of the head's eight lanes for everything that is per EDGE (win_fwd_agg's scheme): lane j reads
      // alpha_j and a_src of its slot's source (2 LDS reads instead of 12), the coefficients go round the group for S, lane j
      // forms g_s / g_e of its slot and stores it (table + exported granule), the g_e go round once more for g_a_dst in CSR
      // order.  Same operands, same order: the same bits; ~45 VALU instructions per lane less.
      const int j = tid % LH;
      const int jj = min(j, MAXD - 1);
      const int nj = d.n[0] * (jj == 0) + d.n[1] * (jj == 1) + d.n[2] * (jj == 2) + d.n[3] * (jj == 3) + d.n[4] * (jj == 4) +
                     d.n[5] * (jj == 5);
      unsigned av[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) av[k] = a_h + (unsigned)(d.n[k] * HC) * 4u;
      const unsigned aaj = a_al + (unsigned)((d.beg + min(jj, d.deg - 1)) * H) * 4u, asj = a_as + (unsigned)(nj * H) * 4u;
      f32x4 hv[MAXD];
      float alj, rawj;
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\t"
          "ds_read_b128 %3, %11\n\tds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\t"
          "ds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(hv[0]), "=&v"(hv[1]), "=&v"(hv[2]), "=&v"(hv[3]), "=&v"(hv[4]), "=&v"(hv[5]), "=&v"(alj), "=&v"(rawj)
          : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(aaj), "v"(asj)
          : "memory");
      const float al_m = j < d.deg ? alj : 0.f;                            // padding slots weigh nothing
      float al[MAXD], ga[MAXD];
      al[0] = group8_bcast<0>(al_m); al[1] = group8_bcast<1>(al_m); al[2] = group8_bcast<2>(al_m);
      al[3] = group8_bcast<3>(al_m); al[4] = group8_bcast<4>(al_m); al[5] = group8_bcast<5>(al_m);
      float S = 0.f, gad = 0.f, ga_m = 0.f;
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        ga[k] = gatres_head_reduce<LH>(gatres_head_dot4(go, as_f4(hv[k])));
        S = fmaf(al[k], ga[k], S);
        ga_m = j == k ? ga[k] : ga_m;
      }
      const float gs = al_m * (ga_m - S);
      const float ge_m = (rawj + adst) > 0.f ? gs : gs * GATRES_NEG_SLOPE;
      if (valid && j < d.deg) {
        g_e[(unsigned)((d.beg + j) * H + hd)] = ge_m;
        // the owner of the edge's SOURCE row needs g_e in its source-major stage
        if (xo.on && (nj < rw.lo || nj >= rw.hi)) xout_store1(xo, xo.t_small + (unsigned)((eabs + d.beg + j) * H + hd), ge_m);
      }
      gad = gad + group8_bcast<0>(ge_m); gad = gad + group8_bcast<1>(ge_m); gad = gad + group8_bcast<2>(ge_m);
      gad = gad + group8_bcast<3>(ge_m); gad = gad + group8_bcast<4>(ge_m); gad = gad + group8_bcast<5>(ge_m);
      if (leader) g_a_dst[(unsigned)(r * H + hd)] = gad;
      continue;
    }
    unsigned av[MAXD], aa[MAXD], as[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      av[k] = a_h + (unsigned)(d.n[k] * HC) * 4u;
      aa[k] = a_al + (unsigned)((d.beg + min(k, d.deg - 1)) * H) * 4u;
      as[k] = a_as + (unsigned)(d.n[k] * H) * 4u;
    }
    f32x4 hv[MAXD];
    float al[MAXD], raw[MAXD], ga[MAXD];
    lds_rd128x6_32x12(av, aa, as, hv, al, raw);
    float S = 0.f, gad = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      ga[k] = gatres_head_reduce<LH>(gatres_head_dot4(go, as_f4(hv[k])));
      raw[k] = raw[k] + adst;
      al[k] = k < d.deg ? al[k] : 0.f;                                   // padding slots weigh nothing
      S = fmaf(al[k], ga[k], S);
    }
#pragma unroll
    for (int k = 0; k < MAXD; ++k) {
      const float gs = al[k] * (ga[k] - S);
      const float ge = raw[k] > 0.f ? gs : gs * GATRES_NEG_SLOPE;
      if (leader && k < d.deg) {
        g_e[(unsigned)((d.beg + k) * H + hd)] = ge;
        // the owner of the edge's SOURCE row needs g_e in its source-major stage
        if (xo.on && (d.n[k] < rw.lo || d.n[k] >= rw.hi)) xout_store1(xo, xo.t_small + (unsigned)((eabs + d.beg + k) * H + hd), ge);
      }
      gad = gad + ge;                                                  // ge == 0 on padding slots
    }
    if (leader) g_a_dst[(unsigned)(r * H + hd)] = gad;
  }
}

// K2 backward, source-major over CSR^T (seg_agg_bwd_src).
//   to: out-edge descriptors (x_k = window-relative edge id); g_out: [row][HC] over the window; alpha / g_e: LDS [e][H]
//   over the WINDOW's edges; g_a_dst: [row][H] own rows; g_h: global (kept for the deferred parameter gradients, row
//   hb + r); g_h2: LDS x operand of the dX stage.
template <int H, int C, int THREADS, bool NH = false, int G2S = H * C>
__device__ __forceinline__ void win_agg_bwd_src(Rows rw, const u16* to, const u16* trp, const u16* teid, const u16* tdst,
                                                int eid_sub, const float* g_out, const float* alpha, const float* g_e,
                                                const float* g_a_dst, const float* att_src, const float* att_dst,
                                                float* g_h, int hb,
                                                float* keep_gas, float* keep_gad, float* g_h2) {
  const int tid = stage_tid();
  constexpr int HC = H * C, G = HC / 4, RPP = THREADS / G;
  const int c0 = (tid % G) * 4;
  const int hd = c0 / C;
  const unsigned a_to = lds_addr(to), a_go = lds_addr(g_out) + (unsigned)c0 * 4u,
                 a_al = lds_addr(alpha) + (unsigned)hd * 4u, a_ge = lds_addr(g_e) + (unsigned)hd * 4u,
                 a_gd = lds_addr(g_a_dst) + (unsigned)hd * 4u;
  const float4 as = ld4(att_src + c0), ad = ld4(att_dst + c0);
  for (int r0 = rw.lo; r0 < rw.hi; r0 += RPP) {
    if (r0 + (int)((tid & ~63u) / G) >= rw.hi) continue;
    int r = r0 + tid / G;
    const bool valid = r < rw.hi;
    if (!valid) r = rw.hi - 1;
    uint4 wa, wb;
    float gad;
    lds_rd128x2_32(a_to + (unsigned)(r - rw.lo) * 32u, a_to + (unsigned)(r - rw.lo) * 32u + 16u,
                   a_gd + (unsigned)(r * H) * 4u, wa, wb, gad);
    const NbrOut d = unpack_out(wa, wb);
    float4 acc = f4zero();
    float gas = 0.f;
    if (__builtin_expect(wave_has_hub_t<NH>(d.deg), 0)) {
      const int beg = trp[r], end = trp[r + 1];
      for (int t = beg; t < end; ++t) {
        const int e = (int)teid[t] - eid_sub, ii = tdst[t];
        gas = gas + g_e[(unsigned)(e * H + hd)];
        gatres_axpy4(acc, alpha[(unsigned)(e * H + hd)], ld4(g_out + (unsigned)(ii * HC + c0)));
      }
    } else {
      if constexpr (C == 32) {
        // one out-edge slot per lane of the head's eight lanes for the per-EDGE operands (win_bwd_dst's scheme): lane j reads
        // alpha and g_e of its slot's edge (2 LDS reads instead of 12) and the two go round the group, masked at the source
        const int j = tid % (C / 4);
        const int jj = min(j, MAXD - 1);
        const int xj = d.x[0] * (jj == 0) + d.x[1] * (jj == 1) + d.x[2] * (jj == 2) + d.x[3] * (jj == 3) + d.x[4] * (jj == 4) +
                       d.x[5] * (jj == 5);
        unsigned av[MAXD];
#pragma unroll
        for (int k = 0; k < MAXD; ++k) av[k] = a_go + (unsigned)(d.d[k] * HC) * 4u;
        f32x4 v[MAXD];
        float alj, gej;
        asm volatile(
            "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\t"
            "ds_read_b128 %3, %11\n\tds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\t"
            "ds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(alj), "=&v"(gej)
            : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(a_al + (unsigned)(xj * H) * 4u),
              "v"(a_ge + (unsigned)(xj * H) * 4u)
            : "memory");
        const bool okj = j < d.deg;
        const float al_m = okj ? alj : 0.f, ge_m = okj ? gej : 0.f;
        gas = gas + group8_bcast<0>(ge_m); gas = gas + group8_bcast<1>(ge_m); gas = gas + group8_bcast<2>(ge_m);
        gas = gas + group8_bcast<3>(ge_m); gas = gas + group8_bcast<4>(ge_m); gas = gas + group8_bcast<5>(ge_m);
        gatres_axpy4(acc, group8_bcast<0>(al_m), as_f4(v[0])); gatres_axpy4(acc, group8_bcast<1>(al_m), as_f4(v[1]));
        gatres_axpy4(acc, group8_bcast<2>(al_m), as_f4(v[2])); gatres_axpy4(acc, group8_bcast<3>(al_m), as_f4(v[3]));
        gatres_axpy4(acc, group8_bcast<4>(al_m), as_f4(v[4])); gatres_axpy4(acc, group8_bcast<5>(al_m), as_f4(v[5]));
      } else {
      unsigned av[MAXD], aa[MAXD], ag[MAXD];
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        av[k] = a_go + (unsigned)(d.d[k] * HC) * 4u;
        aa[k] = a_al + (unsigned)(d.x[k] * H) * 4u;
        ag[k] = a_ge + (unsigned)(d.x[k] * H) * 4u;
      }
      f32x4 v[MAXD];
      float al[MAXD], ge[MAXD];
      lds_rd128x6_32x12(av, aa, ag, v, al, ge);
#pragma unroll
      for (int k = 0; k < MAXD; ++k) {
        const bool ok = k < d.deg;
        gas = gas + (ok ? ge[k] : 0.f);
        gatres_axpy4(acc, ok ? al[k] : 0.f, as_f4(v[k]));
      }
      }
    }
    const bool leader = valid && (c0 % C) == 0;
    if (leader) {                      // kept (global row hb + r) for the deferred att_src / att_dst gradients
      keep_gas[(unsigned)((hb + r) * H + hd)] = gas;
      keep_gad[(unsigned)((hb + r) * H + hd)] = gad;
    }
    gatres_axpy4(acc, gas, as);
    gatres_axpy4(acc, gad, ad);
    if (valid) {
      st4(g_h + (unsigned)((hb + r) * HC + c0), acc);
      st4(g_h2 + (unsigned)(r * G2S + c0), acc);      // (g_h2: the dX stage's x table, row stride G2S)
    }
  }
}

}  // namespace
