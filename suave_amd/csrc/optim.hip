// Fixed-order slab reduction (gradients) and the fused flat-buffer Adam step.
//
// Replaces the accumulation autograd performs into every parameter's .grad and
// torch.optim.Adam.step (suave/model.py:1443, :1504-1506; learning-rate groups
// model.py:1894-1909).
#include "kernels.h"

namespace suave {


__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  const int seg_id = blockIdx.y;
  if (seg_id == a.n_segs) {  // scalar sums (double), one block: strided partials, then a fixed-order tree
    if (blockIdx.x == 0 && a.sums_slab && a.sums_out) {
      __shared__ double part[3][256];
      const int t = threadIdx.x;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
      for (int s = t; s < a.n_sum_slabs; s += 256) {
        s0 += a.sums_slab[(size_t)s * 4 + 0];
        s1 += a.sums_slab[(size_t)s * 4 + 1];
        s2 += a.sums_slab[(size_t)s * 4 + 2];
      }
      part[0][t] = s0; part[1][t] = s1; part[2][t] = s2;
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { part[0][t] += part[0][t + w]; part[1][t] += part[1][t + w]; part[2][t] += part[2][t + w]; }
        __syncthreads();
      }
      if (t < 3) a.sums_out[t] += part[t][0];
      if (t == 0) a.sums_out[3] += a.n_rows;
    }
    return;
  }
  // wave = quarter of the slab range, lane = element: 4 x 16 loads in flight per element, the four partial
  // sums are combined in a fixed order (deterministic).
  __shared__ float part[4][64];
  const ReduceSeg sg = a.segs[seg_id];
  const int64_t total = (int64_t)sg.rows * sg.cols;
  const int sub = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (sg.n_slabs + 3) / 4;
  const int s_begin = min(sub * per, sg.n_slabs), s_end = min(sg.n_slabs, s_begin + per);
  // Slabs whose rows are 16-byte aligned (every GEMM slab: ldc and the slab offsets are multiples of 4 floats) are
  // read as float4: lane = 4 consecutive columns of one row, 1 KB per wave load instead of 256 B.
  const bool vec = ((sg.src_off | sg.src_ld | sg.slab_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.slab) & 15) == 0;
  if (vec) {
    __shared__ float4 part4[4][64];
    const int qpr = (sg.cols + 3) >> 2;  // float4 per row (the padding columns up to src_ld are readable)
    const int64_t total4 = (int64_t)sg.rows * qpr;
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total4; e0 += (int64_t)gridDim.x * 64) {
      const int64_t e = e0 + lane;
      const bool valid = e < total4;
      const int i = valid ? (int)(e / qpr) : 0, j = valid ? (int)(e % qpr) * 4 : 0;
      const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        int s = s_begin;
        for (; s + 16 <= s_end; s += 16) {
          float4 v[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(src + (int64_t)(s + q) * sg.slab_stride);
#pragma unroll
          for (int q = 0; q < 16; ++q) { t.x += v[q].x; t.y += v[q].y; t.z += v[q].z; t.w += v[q].w; }
        }
        for (; s < s_end; ++s) {
          const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)s * sg.slab_stride);
          t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
      }
      part4[sub][lane] = t;
      __syncthreads();
      if (sub == 0 && valid) {
        const float4 p0 = part4[0][lane], p1 = part4[1][lane], p2 = part4[2][lane], p3 = part4[3][lane];
        const float r[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                            ((p0.w + p1.w) + p2.w) + p3.w};
        float* d = a.dst + sg.dst_off + (int64_t)i * sg.dst_si + (int64_t)j * sg.dst_sj;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (j + c < sg.cols) d[(int64_t)c * sg.dst_sj] = r[c];
      }
      __syncthreads();
    }
    return;
  }
  for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < total; e0 += (int64_t)gridDim.x * 64) {
    const int64_t e = e0 + lane;
    const bool valid = e < total;
    const int i = valid ? (int)(e / sg.cols) : 0, j = valid ? (int)(e % sg.cols) : 0;
    const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
    float t = 0.f;
    if (valid) {
      int s = s_begin;
      for (; s + 16 <= s_end; s += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(int64_t)(s + q) * sg.slab_stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) t += v[q];
      }
      for (; s < s_end; ++s) t += src[(int64_t)s * sg.slab_stride];
    }
    part[sub][lane] = t;
    __syncthreads();
    if (sub == 0 && valid)
      a.dst[sg.dst_off + (int64_t)i * sg.dst_si + (int64_t)j * sg.dst_sj] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    __syncthreads();
  }
}

int launch_reduce(const ReduceArgs& a, int max_elems, hipStream_t s) {
  int gx = std::max(1, std::min((max_elems + 63) / 64, 1024));
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
  suave::drop_stale_hip_error();
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
