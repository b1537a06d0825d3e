// Prototype: fp32 GEMM C[M,N] = A[M,K] * W[N,K]^T on the bf16 MFMA by splitting every fp32 operand into
// NS bf16 planes (hi, mid, lo) at LDS-store time and multiplying NP plane pairs (NS=3, NP=6: all pairs with
// i + j <= 2, error ~2^-24 like an fp32 FMA chain; NS=2, NP=3: ~2^-16).  Question answered here: what does
// it buy on the S64 layer shapes against the exact-fp32 MFMA (1/16 of the bf16 rate), and at what error.
// Build: hipcc --offload-arch=gfx950 -O3 -w gemm_split_bf16.hip -o gemm_split_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_ROW = 48;  // bytes per tile row per plane: 16 bf16 (32 B) + 16 B pad (keeps ds_read_b128 aligned)

template <int NS>
__device__ __forceinline__ void split4(const float4 v, bf16x4 (&p)[NS]) {
  float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 h = (__bf16)r[e];  // round to nearest even
      p[s][e] = h;
      r[e] -= (float)h;               // exact
    }
  }
}

template <int NS, int NP>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                              int M, int N, int K, int lda, int ldw, int ldc) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NS * (BM + BN) * LDS_ROW];
  auto Ap = [&](int buf, int s) { return smem + ((buf * NS + s) * (BM + BN)) * LDS_ROW; };
  auto Bp = [&](int buf, int s) { return Ap(buf, s) + BM * LDS_ROW; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (K + BK - 1) / BK;
  // each thread stages 2 float4 of A and 2 of W per K step: idx = tid + 256 p -> row idx >> 2, k quarter idx & 3
  float4 ra[2], rb[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256, row = idx >> 2, kq = idx & 3;
      const int gk = k0 + kq * 4;
      const int gi = min(m0 + row, M - 1), gj = min(n0 + row, N - 1);
      ra[p] = gk < K ? *reinterpret_cast<const float4*>(A + (size_t)gi * lda + gk) : make_float4(0, 0, 0, 0);
      rb[p] = gk < K ? *reinterpret_cast<const float4*>(W + (size_t)gj * ldw + gk) : make_float4(0, 0, 0, 0);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256, row = idx >> 2, kq = idx & 3;
      bf16x4 pa[NS], pb[NS];
      split4<NS>(ra[p], pa);
      split4<NS>(rb[p], pb);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        *reinterpret_cast<bf16x4*>(Ap(buf, s) + row * LDS_ROW + kq * 8) = pa[s];
        *reinterpret_cast<bf16x4*>(Bp(buf, s) + row * LDS_ROW + kq * 8) = pb[s];
      }
    }
  };
  f32x16 acc[2][2] = {};
  const int r31 = lane & 31, h = lane >> 5;
  load(0);
  store(0);
  load(BK);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[s][i] = *reinterpret_cast<const bf16x8*>(Ap(buf, s) + (wm * 64 + i * 32 + r31) * LDS_ROW + h * 16);
        fb[s][i] = *reinterpret_cast<const bf16x8*>(Bp(buf, s) + (wn * 64 + i * 32 + r31) * LDS_ROW + h * 16);
      }
    // smallest products first
#pragma unroll
    for (int sum = NS == 3 ? 2 : 1; sum >= 0; --sum)
#pragma unroll
      for (int sa = 0; sa <= sum; ++sa) {
        const int sb = sum - sa;
        if (sa >= NS || sb >= NS) continue;
        if (NP == 3 && NS == 2 && sum > 1) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][i], fb[sb][j], acc[i][j], 0, 0, 0);
      }
    if (t + 1 < nk) store(buf ^ 1);
    if (t + 2 < nk) load((t + 2) * BK);
    __syncthreads();
  }
  // plain epilogue: row = (reg & 3) + 8 (reg >> 2) + 4 h, col = lane & 31
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        const int col = n0 + wn * 64 + j * 32 + r31;
        if (row < M && col < N) C[(size_t)row * ldc + col] = acc[i][j][reg];
      }
}

template <int NS, int NP>
void run(const char* name, int M, int N, int K) {
  const int lda = (K + 3) / 4 * 4, ldw = lda, ldc = (N + 3) / 4 * 4;
  std::vector<float> hA((size_t)M * lda), hW((size_t)N * ldw);
  srand(1);
  for (auto& v : hA) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hW) v = (float)rand() / RAND_MAX * 2 - 1;
  float *A, *W, *C;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C, (size_t)M * ldc * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm<NS, NP>), dim3(grid), dim3(256), 0, 0, A, W, C, M, N, K, lda, ldw, ldc);
  hipDeviceSynchronize();
  const int n = 20;
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_gemm<NS, NP>), dim3(grid), dim3(256), 0, 0, A, W, C, M, N, K, lda, ldw, ldc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / n * 1e3;
  // error against fp64 on a sample of rows
  std::vector<float> hC((size_t)M * ldc);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double max_err = 0, max_ref = 0;
  for (int r = 0; r < M; r += M / 16 + 1)
    for (int c = 0; c < N; ++c) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * lda + k] * hW[(size_t)c * ldw + k];
      max_err = fmax(max_err, fabs(ref - hC[(size_t)r * ldc + c]));
      max_ref = fmax(max_ref, fabs(ref));
    }
  printf("%-18s NS=%d NP=%d M=%6d N=%4d K=%4d %8.1f us %7.1f TF(algorithmic)  max|err|/max|ref| = %.2e\n", name, NS, NP, M, N, K, us,
         2.0 * M * N * K / us / 1e6, max_err / max_ref);
  hipFree(A); hipFree(W); hipFree(C);
}

int main() {
  run<3, 6>("big square", 4096, 4096, 4096);
  run<2, 3>("big square", 4096, 4096, 4096);
  run<3, 6>("enc L1 fwd", 65536, 256, 205);
  run<2, 3>("enc L1 fwd", 65536, 256, 205);
  run<3, 6>("enc L2 fwd", 65536, 128, 256);
  run<2, 3>("enc L2 fwd", 65536, 128, 256);
  run<3, 6>("dec L1 fwd", 65536, 256, 32);
  run<3, 6>("yproj fwd", 65536, 141, 128);
  run<2, 3>("yproj fwd", 65536, 141, 128);
  return 0;
}
