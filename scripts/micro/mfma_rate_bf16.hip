// Tuning aid: what limits the three-plane split GEMM's K step?  Skeleton of that step on the bf16 MFMA:
// 24 v_mfma_f32_32x32x16_bf16 per wave, NR ds_read_b128 fragment reads, NV VALU ops (the plane split), NW
// ds_write_b64 tile stores, NL global float4 loads and one block barrier, 48 KB of LDS per block (3 blocks / CU,
// as the library kernel).  PHASED = 0 spreads the VALU / stores / loads behind the MFMAs one slot at a time,
// PHASED = 1 issues them after the 24 MFMAs (what in-order code after compute() gives).
// Build: hipcc --offload-arch=gfx950 -O3 -w mfma_rate_bf16.hip -o mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 32;
constexpr int kPlaneBytes = 256 * kRowBytes;       // A rows then B rows
constexpr int kBufBytes = 3 * kPlaneBytes;         // 24 KB
template <int NV, int NR, int NW, int NL, int PHASED, int SWZ>
__global__ __launch_bounds__(256, 3) void k(float* out, const float4* src, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufBytes];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * kBufBytes / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.f;
  __syncthreads();
  f32x16 acc[4] = {};
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh = lane >> 5;
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float4 g[4] = {};
  const float4* sp = src + (size_t)blockIdx.x * 256 + tid;
  for (int it = 0; it < iters; ++it) {
    const unsigned char* rd = lds + (it & 1) * kBufBytes;
    unsigned char* wr = lds + ((it & 1) ^ 1) * kBufBytes;
    bf16x8 fa[3][2], fb[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = 128 + wn * 64 + i * 32 + l31;
        const int ha = SWZ ? (kh ^ ((ra >> 3) & 1)) : kh, hb = SWZ ? (kh ^ ((rb >> 3) & 1)) : kh;
        if (s * 4 + i * 2 < NR) fa[s][i] = *reinterpret_cast<const bf16x8*>(rd + s * kPlaneBytes + ra * kRowBytes + ha * 16);
        else fa[s][i] = fa[0][0];
        if (s * 4 + i * 2 + 1 < NR) fb[s][i] = *reinterpret_cast<const bf16x8*>(rd + s * kPlaneBytes + rb * kRowBytes + hb * 16);
        else fb[s][i] = fb[0][0];
      }
    constexpr int pa_[6] = {0, 0, 1, 1, 0, 2}, pb_[6] = {0, 1, 0, 1, 2, 0};
    auto side = [&](int slot) {
#pragma unroll
      for (int q = 0; q < (NV + 23 - slot) / 24; ++q) v[(slot + q) & 7] = v[(slot + q) & 7] * 1.0001f + 0.5f;
      if (slot < NW) {
        bf16x4 w4;
        w4[0] = (__bf16)v[slot & 7]; w4[1] = w4[0]; w4[2] = w4[0]; w4[3] = w4[0];
        const int idx = tid + (slot & 3) * 256, row = (idx >> 2) & 255, kq = idx & 3;
        *reinterpret_cast<bf16x4*>(wr + (slot >> 2) * kPlaneBytes + row * kRowBytes + kq * 8) = w4;
      }
      if (slot >= 12 && slot - 12 < NL) {
        // loads 0, 1: the A tile (unique per block, streams from HBM); loads 2, 3: the W tile (every block reads the
        // same 16 KB per step out of a 1 MB panel: L2 hits).  NL = 2 loads A only.
        const int q = slot - 12;
        g[q] = q < 2 ? sp[(size_t)(it & 63) * 262144 + q * 65536] : src[(size_t)(it & 63) * 1024 + (q - 2) * 256 + tid];
      }
    };
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa_[q]][m >> 1], fb[pb_[q]][m & 1], acc[m], 0, 0, 0);
        if (!PHASED) {
          __builtin_amdgcn_sched_barrier(0);
          side(q * 4 + m);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    if (PHASED) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int slot = 0; slot < 24; ++slot) side(slot);
    }
    if (NL > 0) v[0] += g[0].x + g[1].y + g[2].z + g[3].w;
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + tid] = s;
}
template <int NV, int NR, int NW, int NL, int PHASED, int SWZ>
void run(int blocks, int iters, float* out, const float4* src) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NR, NW, NL, PHASED, SWZ>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NR, NW, NL, PHASED, SWZ>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * 4 * iters * 24 * 32768.0;
  printf("valu=%3d reads=%2d stores=%2d loads=%d phased=%d swz=%d blocks=%5d %8.3f ms %7.1f TF bf16 = %5.1f TF fp32-equivalent\n", NV, NR, NW,
         NL, PHASED, SWZ, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 6);
  fflush(stdout);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const size_t nsrc = (size_t)64 * 262144 * 16 + (size_t)4 * 65536 * 16 + 4096 * 256 * 16;
  float4* src; hipMalloc(&src, nsrc);
  hipMemset(src, 0, nsrc);
  for (int blocks : {256, 768}) {
    run<0, 0, 0, 0, 0, 0>(blocks, 1000, out, src);
    run<0, 12, 0, 0, 0, 0>(blocks, 1000, out, src);
    run<0, 12, 0, 0, 0, 1>(blocks, 1000, out, src);
    run<72, 12, 0, 0, 0, 1>(blocks, 1000, out, src);
    run<144, 12, 0, 0, 0, 1>(blocks, 1000, out, src);
    run<144, 12, 0, 0, 1, 1>(blocks, 1000, out, src);
    run<0, 12, 12, 0, 0, 1>(blocks, 1000, out, src);
    run<0, 12, 0, 4, 0, 1>(blocks, 1000, out, src);
    run<0, 12, 0, 2, 0, 1>(blocks, 1000, out, src);
    run<144, 12, 12, 4, 0, 0>(blocks, 1000, out, src);
    run<72, 12, 12, 4, 0, 1>(blocks, 1000, out, src);
    run<72, 12, 9, 4, 0, 1>(blocks, 1000, out, src);
    run<48, 12, 9, 4, 0, 1>(blocks, 1000, out, src);
    run<144, 12, 12, 4, 0, 1>(blocks, 1000, out, src);
    run<144, 12, 12, 4, 1, 1>(blocks, 1000, out, src);
  }
  return 0;
}
