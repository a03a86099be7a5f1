// Microbenchmark (diagnostic, not a test): what does a hand-off between two workgroups on DIFFERENT CUs cost on
// MI355X?  Decides whether one snapshot can be spread over several CUs with flag synchronisation between stages.
//   build: hipcc -O3 --offload-arch=gfx950 tests/micro/xcu_sync.hip -o tests/micro/xcu_sync
//   1. prints the XCD every workgroup of a 64 x 1024-thread, 160-KB-LDS grid lands on (dispatch order);
//   2. ping-pong between workgroup pairs (same XCD / different XCD): flag only, and flag + a 48-KB payload written
//      by all 1024 threads of the producer and read by all 1024 threads of the consumer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

__global__ __launch_bounds__(1024) void k_where(unsigned* out) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc_id(); out[2 * blockIdx.x + 1] = hw_id(); lds[0] = 1.f; }
}

constexpr long long SPIN_LIMIT = 20000000;   // ~ tens of ms: a lost partner ends the test instead of hanging the GPU

// Workgroups pa and pb ping-pong `iters` times.  flags[0]: a -> b, flags[1]: b -> a.  PAYLOAD floats per hand-off.
// MODE 0: agent-scope release / acquire (what the memory model asks for: L2 write-back + invalidate on a multi-XCD part)
// MODE 1: relaxed agent-scope flag accesses + "buffer_inv sc0" (L1 only) on the consumer: enough when both CUs share an L2?
// MODE 2: relaxed + "buffer_inv sc1"      MODE 3: relaxed, no invalidate (expected to read stale L1 lines)
template <int MODE>
__device__ __forceinline__ void flag_store(unsigned* f, unsigned v) {
  if (MODE == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  else           __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ bool flag_wait(unsigned* f, unsigned v) {
  long long spin = 0;
  bool ok = true;
  if (MODE == 0) {
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != v) if (++spin > 20000000) { ok = false; break; }
  } else {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != v) if (++spin > 20000000) { ok = false; break; }
    if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
    if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
  }
  return ok;
}

// background traffic: every other workgroup keeps writing its own 256-KB region (dirty L2 lines) until told to stop
__device__ __forceinline__ void background(float* big, const unsigned* stop) {
  float4* mine = reinterpret_cast<float4*>(big) + (size_t)blockIdx.x * 16384;
  for (int it = 0; it < 200000; ++it) {
    for (int k = threadIdx.x; k < 16384; k += 1024) mine[k] = make_float4(it, k, 0.f, 0.f);
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
  }
}

template <int PAYLOAD, int MODE, bool BG>
__global__ __launch_bounds__(1024) void k_pingpong(int pa, int pb, unsigned* flags, float* buf, unsigned long long* t,
                                                   float* sink, int iters, int* err, float* big) {
  extern __shared__ float lds[];
  const int me = blockIdx.x == pa ? 0 : (blockIdx.x == pb ? 1 : -1);
  if (me < 0) {
    if (BG) background(big, flags + 64);
    return;
  }
  float acc = 0.f;
  float4* mine = reinterpret_cast<float4*>(buf) + (size_t)me * (PAYLOAD / 4 + 1024);
  const float4* theirs = reinterpret_cast<const float4*>(buf) + (size_t)(1 - me) * (PAYLOAD / 4 + 1024);
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    if (me == 0) {
      for (int k = threadIdx.x; k < PAYLOAD / 4; k += 1024) mine[k] = make_float4(it, it, it, k);
      __syncthreads();
      if (threadIdx.x == 0) {
        flag_store<MODE>(&flags[0], (unsigned)it);
        if (!flag_wait<MODE>(&flags[32], (unsigned)it)) *err = 1;
      }
      __syncthreads();
      for (int k = threadIdx.x; k < PAYLOAD / 4; k += 1024) { const float4 v = theirs[k]; acc += v.x - (float)it; }
    } else {
      if (threadIdx.x == 0 && !flag_wait<MODE>(&flags[0], (unsigned)it)) *err = 1;
      __syncthreads();
      for (int k = threadIdx.x; k < PAYLOAD / 4; k += 1024) { const float4 v = theirs[k]; acc += v.x - (float)it; }
      for (int k = threadIdx.x; k < PAYLOAD / 4; k += 1024) mine[k] = make_float4(it, it, it, k);
      __syncthreads();
      if (threadIdx.x == 0) flag_store<MODE>(&flags[32], (unsigned)it);
    }
  }
  if (threadIdx.x == 0) {
    t[me] = wall_clock64() - t0;
    if (BG) __hip_atomic_store(flags + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // acc must be exactly 0 if every payload read saw the value of its own iteration
  atomicAdd(sink, fabsf(acc));
}

// M workgroups (ids base, base + stride, ...) run an all-to-all flag barrier `iters` times: every member publishes
// its epoch, then waits for every other member's.  This is the stage barrier a split snapshot would use.
__global__ __launch_bounds__(1024) void k_group_barrier(int base, int stride, int M, unsigned* flags,
                                                        unsigned long long* t, int iters, int* err) {
  extern __shared__ float lds[];
  int me = -1;
  for (int k = 0; k < M; ++k) if ((int)blockIdx.x == base + k * stride) me = k;
  if (me < 0) return;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[me * 32], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < M && threadIdx.x != me) {
      long long spin = 0;
      while (__hip_atomic_load(&flags[threadIdx.x * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it)
        if (++spin > SPIN_LIMIT) { *err = 1; break; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) t[me] = wall_clock64() - t0;
}

int main() {
  const int G = 64, LDS = 160 * 1024;
  unsigned* where; unsigned* flags; float* buf; unsigned long long* t; float* sink; int* err; float* big;
  CK(hipMalloc(&big, (size_t)64 * 16384 * 16));
  CK(hipMalloc(&where, 2 * G * 4)); CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&buf, 4 << 20));
  CK(hipMalloc(&t, 64 * 8)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&err, 4));
  CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_group_barrier, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL(k_where, dim3(G), dim3(1024), LDS, 0, where);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> hw(2 * G);
  CK(hipMemcpy(hw.data(), where, 2 * G * 4, hipMemcpyDeviceToHost));
  printf("workgroup -> XCD (dispatch order, 64 x 1024 threads x 160 KB LDS):\n");
  for (int i = 0; i < G; ++i) printf("%u%s", hw[2 * i], (i % 16 == 15) ? "\n" : " ");
  printf("HW_ID of wg 0, 1, 8, 9: %08x %08x %08x %08x\n", hw[1], hw[3], hw[17], hw[19]);

  const int iters = 2000;
#define PP(PAY, MODE, BG) do { CK(hipFuncSetAttribute((const void*)k_pingpong<PAY, MODE, BG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
    hipLaunchKernelGGL((k_pingpong<PAY, MODE, BG>), dim3(G), dim3(1024), LDS, 0, pa, pb, flags, buf, t, sink, iters, err, big); } while (0)
  auto pingpong = [&](const char* name, int pa, int pb, int variant) -> int {
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(sink, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(t, 0, 16));
    switch (variant) {
      case 0: PP(0, 0, false); break;
      case 1: PP(12288, 0, false); break;
      case 10: PP(12288, 0, true); break;
      case 11: PP(12288, 1, true); break;
      case 12: PP(12288, 2, true); break;
      case 13: PP(12288, 3, true); break;
      case 21: PP(12288, 1, false); break;
      case 30: PP(512, 0, false); break;
      case 31: PP(512, 1, false); break;
      case 32: PP(512, 2, false); break;
      case 33: PP(512, 3, false); break;
      case 23: PP(12288, 3, false); break;
    }
    CK(hipDeviceSynchronize());
    unsigned long long ht[2]; float hs; int he;
    CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, sink, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    printf("%-52s wg %2d <-> %2d (XCD %u / %u): %.3f us per ONE-WAY hand-off   stale-read sum %.1f  timeout %d\n", name, pa,
           pb, hw[2 * pa], hw[2 * pb], ht[0] / 100.0 / iters / 2.0, hs, he);
    return 0;
  };
  // find a same-XCD partner and a different-XCD partner of workgroup 0
  int same = -1, diff = -1;
  for (int i = 1; i < G; ++i) {
    if (same < 0 && hw[2 * i] == hw[0]) same = i;
    if (diff < 0 && hw[2 * i] != hw[0]) diff = i;
  }
  if (same > 0) {
    pingpong("flag only, same XCD", 0, same, 0);
    pingpong("flag + 48 KB payload, same XCD", 0, same, 1);
    pingpong("48 KB, relaxed + buffer_inv sc0, quiet", 0, same, 21);
    pingpong("48 KB, relaxed, NO invalidate, quiet", 0, same, 23);
    pingpong("2 KB (L1-resident), agent acq/rel", 0, same, 30);
    pingpong("2 KB (L1-resident), relaxed + buffer_inv sc0", 0, same, 31);
    pingpong("2 KB (L1-resident), relaxed + buffer_inv sc1", 0, same, 32);
    pingpong("2 KB (L1-resident), relaxed, NO invalidate", 0, same, 33);
    pingpong("48 KB, agent acq/rel, 62 WGs writing", 0, same, 10);
    pingpong("48 KB, relaxed + buffer_inv sc0, 62 WGs writing", 0, same, 11);
    pingpong("48 KB, relaxed + buffer_inv sc1, 62 WGs writing", 0, same, 12);
    pingpong("48 KB, relaxed, NO invalidate, 62 WGs writing", 0, same, 13);
  }
  if (diff > 0) { pingpong("flag only, different XCD", 0, diff, 0); pingpong("flag + 48 KB payload, different XCD", 0, diff, 1); }

  auto group = [&](const char* name, int base, int stride, int M) -> int {
    CK(hipMemset(flags, 0, 4096)); CK(hipMemset(err, 0, 4)); CK(hipMemset(t, 0, 64 * 8));
    hipLaunchKernelGGL(k_group_barrier, dim3(G), dim3(1024), LDS, 0, base, stride, M, flags, t, iters, err);
    CK(hipDeviceSynchronize());
    unsigned long long ht[8]; int he;
    CK(hipMemcpy(ht, t, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    printf("%-30s M=%d base %d stride %d (XCDs", name, M, base, stride);
    for (int k = 0; k < M; ++k) printf(" %u", hw[2 * (base + k * stride)]);
    printf("): %.3f us per barrier  timeout %d\n", ht[0] / 100.0 / iters, he);
    return 0;
  };
  group("group barrier", 0, 8, 2);
  group("group barrier", 0, 8, 4);
  group("group barrier", 0, 8, 8);
  group("group barrier", 0, 1, 2);
  group("group barrier", 0, 1, 4);
  group("group barrier", 0, 1, 8);
  return 0;
}
