// Tuning aid: what limits the fp32 MFMA rate?  Pure MFMA stream with LDS fragment reads and one block barrier
// per 32 MFMAs (the GEMM's skeleton), plus NV independent VALU ops, NS LDS stores and NL global float4 loads per
// 32 MFMAs, spread behind the MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 -w mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NS, int NL>
__global__ __launch_bounds__(256) void k(float* out, const float4* src, int iters) {
  __shared__ float lds[2 * 16 * 132 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 16 * 132 * 2; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[4] = {};
  float a0 = lane, a1 = lane + 1, b0 = 0.5f, b1 = 0.25f;
  const float* Ab = lds + (wave & 1) * 64 + (lane & 31);
  const float* Bb = lds + 16 * 132 * 2 + (wave >> 1) * 64 + (lane & 31);
  const int kh = lane >> 5;
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float4 g[4] = {};
  const float4* sp = src + (size_t)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    float* w = lds + ((it & 1) ? 0 : 16 * 132) + threadIdx.x;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float na0, na1, nb0, nb1;
      const int kr = ((kk + 1) & 7) * 2 + kh;
      na0 = Ab[kr * 132]; na1 = Ab[kr * 132 + 32];
      nb0 = Bb[kr * 132]; nb1 = Bb[kr * 132 + 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(m < 2 ? a0 : a1, (m & 1) ? b1 : b0, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = kk * 4 + m;
#pragma unroll
        for (int q = 0; q < (NV + 31 - slot) / 32; ++q) v[(slot + q) & 7] = v[(slot + q) & 7] * 1.0001f + 0.5f;
        if (slot < NS) w[slot * 264] = v[slot & 7];
        if (slot >= 16 && slot - 16 < NL) g[slot - 16] = sp[(size_t)(it & 63) * 262144 + (slot - 16) * 65536];
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (NL > 0) { v[0] += g[0].x + g[1].y + g[2].z + g[3].w; }
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NS, int NL>
void run(int blocks, int iters, float* out, const float4* src) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NS, NL>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NS, NL>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("valu=%3d stores=%2d loads=%d blocks=%5d %9.3f ms %7.1f TF\n", NV, NS, NL, blocks, ms, fl / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  float4* src; hipMalloc(&src, (size_t)64 * 262144 * 16 + (size_t)4 * 65536 * 16 + 1024 * 256 * 16);
  hipMemset(src, 0, (size_t)64 * 262144 * 16 + (size_t)4 * 65536 * 16 + 1024 * 256 * 16);
  for (int blocks : {256, 512, 768, 1024}) {
    run<0, 0, 0>(blocks, 2000, out, src);
    run<32, 0, 0>(blocks, 2000, out, src);
    run<128, 0, 0>(blocks, 2000, out, src);
    run<256, 0, 0>(blocks, 2000, out, src);
    run<0, 8, 0>(blocks, 2000, out, src);
    run<0, 16, 0>(blocks, 2000, out, src);
    run<0, 0, 4>(blocks, 2000, out, src);
    run<128, 16, 4>(blocks, 2000, out, src);
  }
  return 0;
}
