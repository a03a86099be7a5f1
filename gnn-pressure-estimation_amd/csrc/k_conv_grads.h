// Attention-vector / bias gradient partials of one GATConv (GATConv's att_src / att_dst / bias, GraphModels.py:464-466
// backward): shared by k_misc.hip (the stand-alone launch) and k_proj.hip (co-launched with the weight-gradient partials).
#pragma once
#include "gatres_common.h"

// bf16 tables: a lane owns TWO adjacent columns (one 4-byte load per row and table instead of two 2-byte ones: the
// 2-byte form ran 3x slower than the fp32 kernel on gatres_large).  Column sums do not depend on which lane forms them.
// (the body of conv_param_grads_bf16_kernel, k_misc.hip, for slab s: also run by the extra workgroups of the bf16 weight-
//  gradient launch, k_proj.hip -- the two partial-sum launches of a convolution's backward as ONE launch)
__device__ __forceinline__ void conv_param_grads_bf16_body(
    int s, const gatres_bf16* __restrict__ h, const float* __restrict__ g_a_src, const float* __restrict__ g_a_dst,
    const gatres_bf16* __restrict__ g_out, float* __restrict__ slab_as, float* __restrict__ slab_ad,
    float* __restrict__ slab_b, long long stride, int N, int H, int C, int nps) {
  __shared__ float part[3][3][256];                 // [as|ad|ab][waves 1..3][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HC = H * C;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  float as[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, ad[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, ab[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int n0 = nbeg + wave; n0 < nend; n0 += 32) {
    unsigned hv[8][2], go[8][2];
    float gs[8][2], gd[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n0 + 4 * u;
      const bool ok = n < nend;
      const int nn = ok ? n : nbeg;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = 2 * lane + 128 * cc;
        const bool cok = ok && c < HC;
        const int ci = c < HC ? c : 0;
        hv[u][cc] = cok ? *reinterpret_cast<const unsigned*>(h + (size_t)nn * HC + ci) : 0u;
        go[u][cc] = cok ? *reinterpret_cast<const unsigned*>(g_out + (size_t)nn * HC + ci) : 0u;
        gs[u][cc] = cok ? g_a_src[nn * H + ci / C] : 0.f;
        gd[u][cc] = cok ? g_a_dst[nn * H + ci / C] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const float h0 = __uint_as_float(hv[u][cc] << 16), h1 = __uint_as_float(hv[u][cc] & 0xffff0000u);
        as[cc][0] = fmaf(gs[u][cc], h0, as[cc][0]); as[cc][1] = fmaf(gs[u][cc], h1, as[cc][1]);
        ad[cc][0] = fmaf(gd[u][cc], h0, ad[cc][0]); ad[cc][1] = fmaf(gd[u][cc], h1, ad[cc][1]);
        ab[cc][0] += __uint_as_float(go[u][cc] << 16); ab[cc][1] += __uint_as_float(go[u][cc] & 0xffff0000u);
      }
  }
  if (wave > 0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = 2 * lane + 128 * cc + e;
        part[0][wave - 1][c] = as[cc][e]; part[1][wave - 1][c] = ad[cc][e]; part[2][wave - 1][c] = ab[cc][e];
      }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = 2 * lane + 128 * cc + e;
        if (c < HC) {
          slab_as[(size_t)s * stride + c] = ((as[cc][e] + part[0][0][c]) + part[0][1][c]) + part[0][2][c];
          slab_ad[(size_t)s * stride + c] = ((ad[cc][e] + part[1][0][c]) + part[1][1][c]) + part[1][2][c];
          slab_b[(size_t)s * stride + c] = ((ab[cc][e] + part[2][0][c]) + part[2][1][c]) + part[2][2][c];
        }
      }
  }
}


// One workgroup of four waves per slab: wave w takes rows nbeg + w, nbeg + w + 4, ... with four rows in flight (a
// single wave walking its rows one dependent load at a time ran at 0.4 TB/s on 100k-row graphs); the four partial
// sums meet in LDS in wave order, so the result is deterministic.
template <typename T>
__device__ __forceinline__ void conv_param_grads_body(
    int s, const T* __restrict__ h, const float* __restrict__ g_a_src, const float* __restrict__ g_a_dst,
    const T* __restrict__ g_out, float* __restrict__ slab_as, float* __restrict__ slab_ad,
    float* __restrict__ slab_b, long long stride, int N, int H, int C, int nps) {
  __shared__ float part[3][3][256];                 // [as|ad|ab][waves 1..3][column]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HC = H * C;
  const int nbeg = s * nps, nend = min(N, nbeg + nps);
  float as[4] = {0.f, 0.f, 0.f, 0.f}, ad[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = nbeg + wave; n0 < nend; n0 += 16) {
    float hv[4][4], go[4][4], gs[4][4], gd[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + 4 * u;
      const bool ok = n < nend;
      const int nn = ok ? n : nbeg;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = lane + 64 * cc;
        const bool cok = ok && c < HC;
        const int ci = c < HC ? c : 0;
        hv[u][cc] = cok ? ldval(h + (size_t)nn * HC + ci) : 0.f;
        go[u][cc] = cok ? ldval(g_out + (size_t)nn * HC + ci) : 0.f;
        gs[u][cc] = cok ? g_a_src[nn * H + ci / C] : 0.f;
        gd[u][cc] = cok ? g_a_dst[nn * H + ci / C] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        as[cc] = fmaf(gs[u][cc], hv[u][cc], as[cc]);
        ad[cc] = fmaf(gd[u][cc], hv[u][cc], ad[cc]);
        ab[cc] += go[u][cc];
      }
  }
  if (wave > 0) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      part[0][wave - 1][lane + 64 * cc] = as[cc];
      part[1][wave - 1][lane + 64 * cc] = ad[cc];
      part[2][wave - 1][lane + 64 * cc] = ab[cc];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = lane + 64 * cc;
      if (c < HC) {
        slab_as[(size_t)s * stride + c] = ((as[cc] + part[0][0][c]) + part[0][1][c]) + part[0][2][c];
        slab_ad[(size_t)s * stride + c] = ((ad[cc] + part[1][0][c]) + part[1][1][c]) + part[1][2][c];
        slab_b[(size_t)s * stride + c] = ((ab[cc] + part[2][0][c]) + part[2][1][c]) + part[2][2][c];
      }
    }
  }
}


// what a weight-gradient launch needs to run the column sums in its extra workgroups (h == nullptr: no co-launch)
template <typename T>
struct gatres_conv_grads_co {
  const T* h;
  const float* g_a_src;
  const float* g_a_dst;
  const T* g_out;
  float* slab_as;
  float* slab_ad;
  float* slab_b;
  long long stride;
  int H, C, nps;
};
__device__ __forceinline__ void conv_param_grads_co_run(int s, const gatres_conv_grads_co<float>& cg, int N) {
  conv_param_grads_body<float>(s, cg.h, cg.g_a_src, cg.g_a_dst, cg.g_out, cg.slab_as, cg.slab_ad, cg.slab_b, cg.stride, N,
                               cg.H, cg.C, cg.nps);
}
__device__ __forceinline__ void conv_param_grads_co_run(int s, const gatres_conv_grads_co<gatres_bf16>& cg, int N) {
  conv_param_grads_bf16_body(s, cg.h, cg.g_a_src, cg.g_a_dst, cg.g_out, cg.slab_as, cg.slab_ad, cg.slab_b, cg.stride, N, cg.H,
                             cg.C, cg.nps);
}
