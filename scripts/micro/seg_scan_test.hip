// Stand-alone check of the half-wavefront component sort + segmented DPP scan used by k_loss (debugging aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_get_or_minus1(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, ROW_MASK, 0xF, false); }

__global__ void k_test(const int* kin, const float* gin, int K, float* acc_out, int* dbg) {
  __shared__ int order_s[64];
  __shared__ float acc[32];
  const int lane = threadIdx.x;
  if (lane < 32) acc[lane] = 0.f;
  const int k_nat = kin[lane];
  int rank = 0, base_lo = 0, base_hi = 0;
  for (int j = 0; j < K; ++j) {
    const unsigned long long m = __ballot(k_nat == j);
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    const int below = lane >= 32 ? 32 + base_hi + (int)__builtin_amdgcn_mbcnt_hi(hi, 0u) : base_lo + (int)__builtin_amdgcn_mbcnt_lo(lo, 0u);
    if (k_nat == j) rank = below;
    base_lo += __popc(lo); base_hi += __popc(hi);
  }
  order_s[rank] = lane;
  const int my_row = order_s[lane];
  const int k = __builtin_amdgcn_ds_bpermute(my_row << 2, k_nat);
  const float g = gin[my_row];
  const bool sg1 = dpp_get_or_minus1<0x111>(k) == k, sg2 = dpp_get_or_minus1<0x112>(k) == k;
  const bool sg4 = dpp_get_or_minus1<0x114>(k) == k, sg8 = dpp_get_or_minus1<0x118>(k) == k;
  const bool sgb = dpp_get_or_minus1<0x142, 0xA>(k) == k;
  const int k_next = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, k);
  const bool sg_end = ((lane & 31) == 31) | (k_next != k);
  float v = g, t;
  t = v + dpp_get<0x111>(v); v = sg1 ? t : v;
  t = v + dpp_get<0x112>(v); v = sg2 ? t : v;
  t = v + dpp_get<0x114>(v); v = sg4 ? t : v;
  t = v + dpp_get<0x118>(v); v = sg8 ? t : v;
  t = v + dpp_get<0x142, 0xA>(v); v = sgb ? t : v;
  if (sg_end) (void)__hip_atomic_fetch_add(acc + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  if (lane < 32) acc_out[lane] = acc[lane];
  dbg[lane] = my_row | (k << 8) | (sg1 << 16) | (sg2 << 17) | (sg4 << 18) | (sg8 << 19) | (sgb << 20) | (sg_end << 21);
  ((float*)dbg)[64 + lane] = v;
}
int main() {
  int bad = 0;
  srand(1);
  int *dk, *ddbg; float *dg, *dacc;
  (void)hipMalloc(&dk, 256); (void)hipMalloc(&dg, 256); (void)hipMalloc(&dacc, 128); (void)hipMalloc(&ddbg, 512);
  for (int trial = 0; trial < 200; ++trial) {
    const int K = 1 + rand() % 12;
    std::vector<int> k(64); std::vector<float> g(64);
    for (int i = 0; i < 64; ++i) { k[i] = rand() % K; g[i] = (rand() % 2001 - 1000) / 1000.f; }
    if (trial == 0) { const int kk[64] = {1,2,0,2,2,2,0,2,2,0,2,0,2,1,1,2,0,2,0,1,2,2,0,0,2,0,0,0,2,2,2,0,1,1,0,1,2,1,2,2,0,2,1,2,0,0,1,0,2,2,2,0,2,0,0,0,0,1,0,1,1,2,0,1}; for (int i = 0; i < 64; ++i) k[i] = kk[i]; }
    (void)hipMemcpy(dk, k.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dg, g.data(), 256, hipMemcpyHostToDevice);
    const int Kk = trial == 0 ? 3 : K;
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dk, dg, Kk, dacc, ddbg);
    float acc[32]; int dbg[128];
    (void)hipMemcpy(acc, dacc, 128, hipMemcpyDeviceToHost); (void)hipMemcpy(dbg, ddbg, 512, hipMemcpyDeviceToHost);
    for (int j = 0; j < Kk; ++j) {
      double ref = 0; for (int i = 0; i < 64; ++i) if (k[i] == j) ref += g[i];
      if (fabs(ref - acc[j]) > 1e-4) {
        if (bad < 3) {
          printf("trial %d K %d comp %d got %f ref %f\n", trial, Kk, j, acc[j], ref);
          for (int i = 0; i < 64; ++i) printf("  lane %2d row %2d k %d flags %02x v %f\n", i, dbg[i] & 255, (dbg[i] >> 8) & 255, dbg[i] >> 16, ((float*)dbg)[64 + i]);
        }
        ++bad;
      }
    }
  }
  printf("bad %d\n", bad);
  return bad != 0;
}
