// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, qq = j >> 2, pp = j & 3;
  // group g, lane 4*qq + pp: row 4g + qq of the image, columns 4pp .. 4pp + 3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + (4 * g + qq) * 64 + 4 * pp));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short h[64 * 64], o[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (short)(r * 100 + c);
  short *d, *dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int g = lane >> 4, j = lane & 15;
    for (int e = 0; e < 4; ++e) bad += o[lane * 4 + e] != (short)((4 * g + e) * 100 + j);     // expected: v[e] = img[4g + e][j]
  }
  printf("expected v[e] = img[4g+e][j]: mismatches %d\n", bad);
  for (int lane : {0, 1, 5, 16, 37}) printf("lane %2d: %d %d %d %d\n", lane, o[lane*4], o[lane*4+1], o[lane*4+2], o[lane*4+3]);
  return bad != 0;
}
