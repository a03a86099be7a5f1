// Microbenchmark (diagnostic, not a test): what do the primitives of a one-workgroup-per-CU kernel cost on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each test: 1024 threads, loops ITER times, thread 0 writes elapsed wall-clock ticks (100 MHz)
__global__ __launch_bounds__(1024) void k_barrier_only(unsigned long long* t, int iters) {
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) __syncthreads();
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
}
__global__ __launch_bounds__(1024) void k_store_barrier(float* buf, unsigned long long* t, int iters) {
  float4* p = reinterpret_cast<float4*>(buf) + (size_t)blockIdx.x * 1024 * 8 + threadIdx.x;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) { p[(i & 7) * 1024] = make_float4(i, i, i, i); __syncthreads(); }
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
}
// store, barrier, then read what ANOTHER wave of the same workgroup wrote, barrier
__global__ __launch_bounds__(1024) void k_store_barrier_load(float* buf, unsigned long long* t, float* sink, int iters) {
  float4* base = reinterpret_cast<float4*>(buf) + (size_t)blockIdx.x * 1024 * 8;
  float acc = 0.f;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    base[(i & 7) * 1024 + threadIdx.x] = make_float4(i, i, i, i);
    __syncthreads();
    float4 v = base[(i & 7) * 1024 + ((threadIdx.x + 517) & 1023)];
    acc += v.x;
    __syncthreads();
  }
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
  if (acc == 12345.f) sink[0] = acc;
}
// dependent global loads (pointer chase inside a 64 KB region: L2 resident after first pass)
__global__ __launch_bounds__(1024) void k_chase_global(const int* nxt, unsigned long long* t, int* sink, int iters) {
  int j = threadIdx.x;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) j = nxt[blockIdx.x * 16384 + j];
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
  if (j == -1) sink[0] = j;
}
__global__ __launch_bounds__(1024) void k_chase_lds(const int* nxt, unsigned long long* t, int* sink, int iters) {
  __shared__ int l[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) l[i] = nxt[i];
  __syncthreads();
  int j = threadIdx.x;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) j = l[j];
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
  if (j == -1) sink[0] = j;
}
// 64 MFMAs per wave per iteration, 16 waves
__global__ __launch_bounds__(1024) void k_mfma(unsigned long long* t, float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, w = 1.0001f;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a3, 0, 0, 0);
    }
  }
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
  if (a0[0] + a1[0] + a2[0] + a3[0] == 1.f) sink[0] = 1.f;
}
// VALU issue: 256 dependent-free fmas per thread per iteration
__global__ __launch_bounds__(1024) void k_valu(unsigned long long* t, float* sink, int iters) {
  float a[8];
  for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 32; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = fmaf(a[k], 1.0001f, 0.5f);
  }
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64() - t0;
  float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
  if (s == 1.f) sink[0] = s;
}

int main() {
  const int WG = 32, IT = 200;
  unsigned long long* t; float* buf; float* sink; int* nxt; int* isink;
  CK(hipMalloc(&t, WG * 8)); CK(hipMalloc(&buf, (size_t)WG * 1024 * 8 * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&isink, 64));
  std::vector<int> h(WG * 16384);
  for (int b = 0; b < WG; ++b) for (int i = 0; i < 16384; ++i) h[b * 16384 + i] = (i * 1031 + 77) & 16383;
  CK(hipMalloc(&nxt, h.size() * 4)); CK(hipMemcpy(nxt, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<unsigned long long> ht(WG);
  auto report = [&](const char* name, double per) {
    CK(hipDeviceSynchronize()); CK(hipMemcpy(ht.data(), t, WG * 8, hipMemcpyDeviceToHost));
    double mx = 0; for (auto v : ht) mx = v > mx ? v : mx;
    printf("%-34s %8.3f us per iteration (%s)\n", name, mx / 100.0 / IT, per > 0 ? "" : ""); return 0; };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_barrier_only, dim3(WG), dim3(1024), 0, 0, t, IT); report("barrier only", 0);
    hipLaunchKernelGGL(k_store_barrier, dim3(WG), dim3(1024), 0, 0, buf, t, IT); report("16B store + barrier", 0);
    hipLaunchKernelGGL(k_store_barrier_load, dim3(WG), dim3(1024), 0, 0, buf, t, sink, IT); report("store+bar+load(other wave)+bar", 0);
    hipLaunchKernelGGL(k_chase_global, dim3(WG), dim3(1024), 0, 0, nxt, t, isink, IT); report("dependent global load (L2)", 0);
    hipLaunchKernelGGL(k_chase_lds, dim3(WG), dim3(1024), 0, 0, nxt, t, isink, IT); report("dependent LDS load", 0);
    hipLaunchKernelGGL(k_mfma, dim3(WG), dim3(1024), 0, 0, t, sink, IT); report("64 MFMA16x16x4f32/wave x16 waves", 0);
    hipLaunchKernelGGL(k_valu, dim3(WG), dim3(1024), 0, 0, t, sink, IT); report("256 fma/thread x16 waves", 0);
  }
  return 0;
}
