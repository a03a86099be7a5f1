// Micro-probe (not a test): what does a wave that has issued LDS-DMA (global_load_lds) wait for?
//   t_issue : cycles to issue N DMA instructions (no wait)
//   t_lgkm  : cycles until `s_waitcnt lgkmcnt(0)` returns after the issue
//   t_vm    : cycles until `s_waitcnt vmcnt(0)` returns (the transfers have landed)
// One workgroup of 256 threads per CU, every CU streaming a different 1-MiB-strided range (HBM-resident: 1 GiB buffer).
// Build: hipcc -O3 --offload-arch=gfx950 tests/micro/lds_dma_probe.hip -o tests/micro/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, long long* __restrict__ out, int stride_floats) {
  __shared__ __attribute__((aligned(16))) float lds[4 * N * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* p = src + (size_t)blockIdx.x * stride_floats + (size_t)wave * N * 256 + lane * 4;
  float* d = lds + wave * N * 256;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll
  for (int k = 0; k < N; ++k) __builtin_amdgcn_global_load_lds(p + k * 256, d + k * 256, 16, 0, 0);
  const long long t1 = clock64();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t2 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t3 = clock64();
  if (lane == 0) {
    long long* o = out + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = t1 - t0; o[1] = t2 - t0; o[2] = t3 - t0; o[3] = (long long)lds[lane];
  }
}

template <int N>
static void run(const float* src, long long* out, int blocks, const char* what) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(256), 0, 0, src, out, 1 << 18);
  hipDeviceSynchronize();
  std::vector<long long> h((size_t)blocks * 16);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double s[3] = {0, 0, 0};
  for (int b = 0; b < blocks * 4; ++b)
    for (int k = 0; k < 3; ++k) s[k] += (double)h[(size_t)b * 4 + k];
  printf("%-28s N=%2d (%3d KB per CU)  issue %7.0f  lgkmcnt(0) %7.0f  vmcnt(0) %7.0f  cycles (mean over waves)\n", what, N,
         N * 4, s[0] / (blocks * 4), s[1] / (blocks * 4), s[2] / (blocks * 4));
}

int main() {
  float* src;
  long long* out;
  hipMalloc(&src, (size_t)1 << 30);
  hipMemset(src, 0, (size_t)1 << 30);
  hipMalloc(&out, 1 << 20);
  run<2>(src, out, 256, "all 256 CUs");
  run<8>(src, out, 256, "all 256 CUs");
  run<16>(src, out, 256, "all 256 CUs");
  run<8>(src, out, 8, "8 CUs");
  run<16>(src, out, 8, "8 CUs");
  return 0;
}
