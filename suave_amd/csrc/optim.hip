// Fixed-order slab reduction (gradients) and the fused flat-buffer Adam step.
//
// Replaces the accumulation autograd performs into every parameter's .grad and
// torch.optim.Adam.step (suave/model.py:1443, :1504-1506; learning-rate groups
// model.py:1894-1909).
#include "kernels.h"

namespace suave {


__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  const int seg_id = blockIdx.y;
  if (seg_id == a.n_segs) {  // scalar sums (double), one block
    if (blockIdx.x == 0 && threadIdx.x < 3 && a.sums_slab && a.sums_out) {
      double t = 0.0;
      for (int s = 0; s < a.n_sum_slabs; ++s) t += a.sums_slab[(size_t)s * 4 + threadIdx.x];
      a.sums_out[threadIdx.x] += t;
      if (threadIdx.x == 0) a.sums_out[3] += a.n_rows;
    }
    return;
  }
  const ReduceSeg sg = a.segs[seg_id];
  const int64_t total = (int64_t)sg.rows * sg.cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / sg.cols), j = (int)(e % sg.cols);
    const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
    // fixed summation order; 16 independent loads in flight per thread
    float t = 0.f;
    int s = 0;
    for (; s + 16 <= sg.n_slabs; s += 16) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = src[(int64_t)(s + q) * sg.slab_stride];
#pragma unroll
      for (int q = 0; q < 16; ++q) t += v[q];
    }
    for (; s < sg.n_slabs; ++s) t += src[(int64_t)s * sg.slab_stride];
    a.dst[sg.dst_off + (int64_t)i * sg.dst_si + (int64_t)j * sg.dst_sj] = t;
  }
}

int launch_reduce(const ReduceArgs& a, int max_elems, hipStream_t s) {
  int gx = std::max(1, std::min((max_elems + 255) / 256, 256));
  ProfScope prof_scope(kProfReduce, s);
  hipLaunchKernelGGL(k_reduce, dim3(gx, a.n_segs + 1), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

constexpr int kMaxGroups = 8;
struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  int32_t begin[kMaxGroups + 1];
  float step_size[kMaxGroups];  // lr_g / (1 - beta1^t)
  int n_groups;
  float beta1, beta2, eps, inv_sqrt_bc2;
};

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  const int n = a.begin[a.n_groups];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (i < a.begin[0]) continue;
    int gidx = 0;
#pragma unroll
    for (int q = 1; q < kMaxGroups; ++q)
      if (q < a.n_groups && i >= a.begin[q]) gidx = q;
    if (a.step_size[gidx] == 0.f) continue;  // frozen range: no moment update either
    const float g = a.g[i];
    const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
    a.m[i] = m;
    a.v[i] = v;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    a.p[i] -= a.step_size[gidx] * (m / denom);
  }
}

}  // namespace suave

extern "C" int suave_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               const int32_t* group_begin_host, const float* lr_host, int32_t n_groups,
                               int64_t step_t, float beta1, float beta2, float eps, void* stream) {
  using namespace suave;
  SUAVE_REQUIRE(n_groups >= 1 && n_groups <= kMaxGroups, "adam: 1..8 learning-rate groups");
  SUAVE_REQUIRE(step_t >= 1, "adam: step count is 1-based");
  AdamArgs a;
  a.p = params;
  a.g = grads;
  a.m = exp_avg;
  a.v = exp_avg_sq;
  a.n_groups = n_groups;
  const double bc1 = 1.0 - pow((double)beta1, (double)step_t);
  const double bc2 = 1.0 - pow((double)beta2, (double)step_t);
  for (int i = 0; i <= n_groups; ++i) a.begin[i] = group_begin_host[i];
  for (int i = 0; i < n_groups; ++i) {
    SUAVE_REQUIRE(a.begin[i] <= a.begin[i + 1], "adam: group ranges must be ascending");
    a.step_size[i] = (float)((double)lr_host[i] / bc1);
  }
  a.beta1 = beta1;
  a.beta2 = beta2;
  a.eps = eps;
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int n = a.begin[n_groups];
  if (n <= 0) return 0;
  const int grid = std::min((n + 255) / 256, 2048);
  ProfScope prof_scope(kProfAdam, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}
