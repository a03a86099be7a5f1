// Device mask sampler (utils/auxil.py:143-182 on the GPU), shared by mask_generate_kernel (k_misc.hip: the sampler's own
// launch, which also stages / collates the batch) and reduce_adam_kernel (k_fused_host.hip: the mask of the NEXT step is
// sampled by extra workgroups of the update launch).
#pragma once
#include "gatres_common.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Every node gets a unique 64-bit key (hash32(seed, step, node) << 32 | local id); the k = int(n*rate) smallest keys are
// masked: an exactly-k uniform subset without replacement, the same distribution as np.random.choice(n, k, replace=False)
// in utils/auxil.py:154-157.  The k-th key is found by an 8-pass byte-wise radix select in LDS (graphs beyond 2048 nodes)
// or by an all-pairs rank.
__device__ __forceinline__ uint64_t mask_key(uint64_t seed, uint64_t step, int gnode, int local) {
  const uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (step + 1) + 0xBF58476D1CE4E5B9ULL * (uint64_t)(gnode + 1));
  return ((z >> 32) << 32) | (uint32_t)local;
}

constexpr int MASK_WGS = 4;          // workgroups per graph of the sampler's own launch: each ranks one slice of the graph's nodes
constexpr int MASK_WGS_UPDATE = 16;  // ... of the update launch's sampling tail (256 threads each: smaller slices, 3 us beside the reduction)
// Workgroup `wq` of WGS for graph g (T threads): mask[n0 .. n0 + n) for its slice (the whole graph beyond 2048 nodes, wq == 0).
template <int T, int WGS>
__device__ __forceinline__ void mask_sample_graph(const int* __restrict__ node_ptr, double rate, uint64_t seed, uint64_t step,
                                                  uint8_t* __restrict__ mask, int g, int wq) {
  constexpr int SMALL = 2048;                      // graphs up to SMALL nodes: all-pairs rank in LDS (~2 us for C-Town)
  __shared__ uint64_t s_keys[SMALL];
  __shared__ int s_rank[SMALL];
  __shared__ int hist[256];
  __shared__ uint64_t s_prefix;
  __shared__ int s_k;
  const int tid = threadIdx.x;
  const int n0 = node_ptr[g], n = node_ptr[g + 1] - n0;
  const int k = (int)((double)n * rate);          // Python: int(num_nodes * masking_rate)
  if (k <= 0) {
    if (wq == 0)
      for (int v = tid; v < n; v += T) mask[n0 + v] = 0;
    return;
  }
  if (n > SMALL && wq != 0) return;               // the radix-select path runs in one workgroup
  if (n <= SMALL) {
    // keys are unique, so "masked" == "fewer than k keys are smaller than mine".  The n x n comparisons are spread
    // over all threads: Q threads per node, each ranks the node against one slice of the keys.
    for (int v = tid; v < n; v += T) { s_keys[v] = mask_key(seed, step, n0 + v, v); s_rank[v] = 0; }
    __syncthreads();
    const int Q = min(16, max(1, (n + 63) / 64));   // slices of ~64 keys: 7 for C-Town's 388 nodes
    const int slice = (n + Q - 1) / Q;
    const int per = (n + WGS - 1) / WGS, vlo = min(n, wq * per), vhi = min(n, vlo + per);
    for (int w = tid; w < (vhi - vlo) * Q; w += T) {
      const int v = vlo + w / Q, q = w % Q;
      const uint64_t mine = s_keys[v];
      const int ub = q * slice, ue = min(n, ub + slice);
      int rank = 0;
      for (int u = ub; u < ue; ++u) rank += s_keys[u] < mine ? 1 : 0;
      if (Q == 1) s_rank[v] = rank; else atomicAdd(&s_rank[v], rank);
    }
    __syncthreads();
    for (int v = vlo + tid; v < vhi; v += T) mask[n0 + v] = s_rank[v] < k ? 1 : 0;
    return;
  }
  if (tid == 0) { s_prefix = 0; s_k = k; }
  for (int pass = 7; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint64_t prefix = s_prefix;
    for (int v = tid; v < n; v += T) {
      const uint64_t key = mask_key(seed, step, n0 + v, v);
      const bool match = (pass == 7) || ((key >> (8 * (pass + 1))) == prefix);
      if (match) atomicAdd(&hist[(int)((key >> (8 * pass)) & 255)], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int kk = s_k, b = 0;
      while (b < 255 && kk > hist[b]) { kk -= hist[b]; ++b; }
      s_k = kk;
      s_prefix = (prefix << 8) | (uint64_t)b;
    }
    __syncthreads();
  }
  const uint64_t kth = s_prefix;
  for (int v = tid; v < n; v += T) mask[n0 + v] = mask_key(seed, step, n0 + v, v) <= kth ? 1 : 0;
}

}  // namespace
