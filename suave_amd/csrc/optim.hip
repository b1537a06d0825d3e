// Fixed-order slab reduction (gradients) and the fused flat-buffer Adam step.
//
// Replaces the accumulation autograd performs into every parameter's .grad and
// torch.optim.Adam.step (suave/model.py:1443, :1504-1506; learning-rate groups
// model.py:1894-1909).
#include "kernels.h"

namespace suave {


// Work list of one launch: block 0 sums the scalar (double) partials; blocks blk_begin[g] .. blk_begin[g + 1] own
// reduce_epb(segment g) consecutive (vector) elements each.  A flat list instead of a (max elements, segments) grid:
// most blocks of that grid had nothing to do, and the few that had ran two to a CU.
struct ReduceMap {
  int32_t blk_begin[kMaxReduceSegs + 1];
};
constexpr int kReduceWaves = 16;  // waves of a block

// Elements per block.  The slab range of an element is cut over the block's waves, and for segments with many
// slabs and few elements (per-block column sums: 1024 slabs of 74 x 12 floats) also over lane groups of a wave, so
// that a lane's whole share (<= 16 loads) is in flight at once: the kernel is a latency chain, not a stream.
// With at most 16 slabs (small batches) a thread sums its element's whole range itself, 1024 elements per block:
// sixteen waves cutting a range of four slabs left most of them idle and the launch was mostly block dispatch.
constexpr int kReduceFewSlabs = 16;
__host__ __device__ inline int reduce_epb(int n_slabs) {
  return n_slabs > 1024 ? 4 : n_slabs > 256 ? 16 : n_slabs > kReduceFewSlabs ? 64 : kReduceWaves * 64;
}

__host__ __device__ inline bool reduce_seg_vec(const ReduceSeg& sg, const float* slab) {
  // Slabs whose rows are 16-byte aligned (every GEMM slab: ldc and the slab offsets are multiples of 4 floats) are
  // read as float4: lane = 4 consecutive columns of one row, 1 KB per wave load instead of 256 B.
  return ((sg.src_off | sg.src_ld | sg.slab_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
}

// dW_head[row(q, c)][j] = sum over the chunks of component c (all chunks for the logits rows), fixed order.
// Both shapes are latency chains of strided loads, so they are cut short rather than wide:
//   * blocks [0, logit_blocks): the K logits rows, whose sums run over EVERY chunk (266 at B = 65 536): 64 elements
//     per block, the four waves take a quarter of the chunk range each, 16 loads in flight, quarters added in order;
//   * the rest: the mu / logvar rows of component c sum only c's chunks (~27): one thread per element, 16 loads in
//     flight, the tail of the range as a guarded batch instead of one load at a time.
// It rides in k_reduce's launch (as a launch of its own between the head's backward pair and the encoder trunk it
// was 5 - 7 us on the step's chain): a k_reduce block of 16 waves runs four of its 256-thread blocks side by side.
__device__ __forceinline__ void head_reduce_blocks(const HeadReduceArgs& a, int logit_blocks, int n_vblocks, int first_vb) {
  __shared__ float part[4][4][64];  // [virtual block][quarter][element]
  const int v = threadIdx.x >> 8, vtid = threadIdx.x & 255;
  const int vb = first_vb + v;
  const int n1 = a.n_in + 1;
  const size_t stride = (size_t)a.ldS * a.ldc;
  const bool logit_vb = vb < logit_blocks;
  const int sub = vtid >> 6, lane = vtid & 63;
  const int e_l = vb * 64 + lane;
  const bool valid_l = logit_vb && e_l < a.K * n1;
  const int R_l = valid_l ? e_l / n1 : 0, j_l = valid_l ? e_l - R_l * n1 : 0;
  if (logit_vb) {
    const int S = a.chunk_begin[a.K];
    const int per = (S + 3) / 4;
    const int s0 = min(sub * per, S), s1 = min(S, s0 + per);
    const float* src = a.slab + (size_t)R_l * a.ldc + j_l;
    float t = 0.f;
    if (valid_l) {
      for (int s = s0; s < s1; s += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (s + u < s1) ? src[(size_t)(s + u) * stride] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += x[u];
      }
    }
    part[v][sub][lane] = t;
  }
  __syncthreads();  // (every thread of the block: the virtual blocks of one block may be of either kind)
  if (logit_vb) {
    if (sub == 0 && valid_l) {
      const float r = ((part[v][0][lane] + part[v][1][lane]) + part[v][2][lane]) + part[v][3][lane];
      if (j_l < a.n_in) a.grads[a.w_off + (size_t)R_l * a.ld + j_l] = r;
      else a.grads[a.b_off + R_l] = r;
    }
    return;
  }
  if (vb >= n_vblocks) return;
  const int64_t total = (int64_t)2 * a.K * a.D * n1;
  for (int64_t e = (int64_t)(vb - logit_blocks) * 256 + vtid; e < total; e += (int64_t)(n_vblocks - logit_blocks) * 256) {
    const int t_row = (int)(e / n1), j = (int)(e % n1);
    const int R = a.K + t_row;
    const bool is_lv = t_row >= a.K * a.D;
    const int u_row = is_lv ? t_row - a.K * a.D : t_row;
    const int c = u_row / a.D, d = u_row % a.D;
    const int q = a.ldK + (is_lv ? a.ldD : 0) + d;
    const int s0 = a.chunk_begin[c], s1 = a.chunk_begin[c + 1];
    const float* src = a.slab + (size_t)q * a.ldc + j;
    float t = 0.f;
    for (int s = s0; s < s1; s += 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = (s + u < s1) ? src[(size_t)(s + u) * stride] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += x[u];
    }
    if (j < a.n_in) a.grads[a.w_off + (size_t)R * a.ld + j] = t;
    else a.grads[a.b_off + R] = t;
  }
}

__global__ __launch_bounds__(kReduceWaves * 64) void k_reduce(ReduceArgs a, ReduceMap map, HeadReduceArgs head, int head_logit_blocks,
                                                              int head_vblocks) {
  const int t = threadIdx.x;
  if ((int)blockIdx.x > map.blk_begin[kMaxReduceSegs]) {  // behind the segment blocks: the encoder head's chunk slabs
    head_reduce_blocks(head, head_logit_blocks, head_vblocks, ((int)blockIdx.x - 1 - map.blk_begin[kMaxReduceSegs]) * 4);
    return;
  }
  if (blockIdx.x == 0) {  // scalar sums (double), one block -- first in dispatch order: it is a latency chain
    if (a.sums_slab && a.sums_out) {
      __shared__ double part[3][kReduceWaves];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
      for (int s = t; s < a.n_sum_slabs; s += kReduceWaves * 64 * 4) {  // 12 loads in flight per thread
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = s + kReduceWaves * 64 * u;
#pragma unroll
          for (int c = 0; c < 3; ++c) v[u][c] = e < a.n_sum_slabs ? a.sums_slab[(size_t)e * 4 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }
      }
      // lanes of a wave in a fixed butterfly order, then the sixteen wave sums in order: one barrier instead of ten
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
      }
      if ((t & 63) == 0) { part[0][t >> 6] = s0; part[1][t >> 6] = s1; part[2][t >> 6] = s2; }
      __syncthreads();
      if (t < 3) {
        double r = 0.0;
        for (int w = 0; w < kReduceWaves; ++w) r += part[t][w];
        part[t][0] = r;
      }
      if (t < 3) a.sums_out[t] += part[t][0];
      if (t == 0) a.sums_out[3] += a.n_rows;
    }
    return;
  }
  int seg_id = 0;
  const int blk = blockIdx.x - 1;
#pragma unroll
  for (int g = 1; g < kMaxReduceSegs; ++g)  // (uniform; the whole ascending table is fetched at once, entries past
    seg_id = blk >= map.blk_begin[g] ? g : seg_id;  //  the last segment hold the block count: never selected)
  const ReduceSeg sg = a.segs[seg_id];
  if (sg.n_slabs <= kReduceFewSlabs) {  // (block-uniform) one thread per (vector) element, every slab load in flight at once
    const int cols_f = sg.cols + (sg.bias_off1 ? 1 : 0);
    const bool vec = reduce_seg_vec(sg, a.slab);
    const int qpr = vec ? (cols_f + 3) >> 2 : cols_f;
    const int64_t e = (int64_t)(blk - map.blk_begin[seg_id]) * (kReduceWaves * 64) + t;
    if (e >= (int64_t)sg.rows * qpr) return;
    const int i = (int)(e / qpr), j = (int)(e % qpr) * (vec ? 4 : 1);
    const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      float4 v[kReduceFewSlabs];
#pragma unroll
      for (int q = 0; q < kReduceFewSlabs; ++q)
        v[q] = *reinterpret_cast<const float4*>(src + (int64_t)min(q, max(sg.n_slabs - 1, 0)) * sg.slab_stride);
#pragma unroll
      for (int q = 0; q < kReduceFewSlabs; ++q)
        if (q < sg.n_slabs) { r[0] += v[q].x; r[1] += v[q].y; r[2] += v[q].z; r[3] += v[q].w; }
    } else {
      float v[kReduceFewSlabs];
#pragma unroll
      for (int q = 0; q < kReduceFewSlabs; ++q) v[q] = src[(int64_t)min(q, max(sg.n_slabs - 1, 0)) * sg.slab_stride];
#pragma unroll
      for (int q = 0; q < kReduceFewSlabs; ++q)
        if (q < sg.n_slabs) r[0] += v[q];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!vec && c > 0) break;
      const int jj = j + c;
      if (jj < sg.cols) a.dst[sg.dst_off + (int64_t)i * sg.dst_si + (int64_t)jj * sg.dst_sj] = r[c];
      else if (jj < cols_f) a.dst[sg.bias_off1 - 1 + i] = r[c];
    }
    return;
  }
  const int sub = t >> 6, lane = t & 63;
  const int epb = reduce_epb(sg.n_slabs), cpw = 64 / epb;  // elements per block, slab cuts per wave
  const int n_cuts = kReduceWaves * cpw;
  const int el = lane % epb, cut = sub * cpw + lane / epb;
  const int per = (sg.n_slabs + n_cuts - 1) / n_cuts;
  const int s_begin = min(cut * per, sg.n_slabs), s_end = min(sg.n_slabs, s_begin + per);
  const int64_t e = (int64_t)(blk - map.blk_begin[seg_id]) * epb + el;
  // The layer's bias gradient is column `cols` of the slab rows (the GEMM's ones column): it rides along as one more
  // column and is stored to its own place (a segment of its own read one float per 1 KB row: 64 cache lines a load).
  const int cols_x = sg.cols + (sg.bias_off1 ? 1 : 0);
  auto store = [&](int i, int j, float v) {
    if (j < sg.cols) a.dst[sg.dst_off + (int64_t)i * sg.dst_si + (int64_t)j * sg.dst_sj] = v;
    else if (j < cols_x) a.dst[sg.bias_off1 - 1 + i] = v;
  };
  // the cuts' partial sums are combined in a fixed order (deterministic)
  if (reduce_seg_vec(sg, a.slab)) {
    __shared__ float4 part4[kReduceWaves][64];
    const int qpr = (cols_x + 3) >> 2;  // float4 per row (the padding columns up to src_ld are readable)
    const bool valid = e < (int64_t)sg.rows * qpr;
    const int i = valid ? (int)(e / qpr) : 0, j = valid ? (int)(e % qpr) * 4 : 0;
    const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      for (int s = s_begin; s < s_end; s += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)  // (clamped: a branch-free batch of eight loads in flight)
          v[q] = *reinterpret_cast<const float4*>(src + (int64_t)min(s + q, s_end - 1) * sg.slab_stride);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (s + q < s_end) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
      }
    }
    part4[sub][lane] = acc;
    __syncthreads();
    if (cut == 0 && valid) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int c = 0; c < n_cuts; ++c) {
        const float4 pw = part4[c / cpw][(c % cpw) * epb + el];
        r.x += pw.x; r.y += pw.y; r.z += pw.z; r.w += pw.w;
      }
      store(i, j, r.x); store(i, j + 1, r.y); store(i, j + 2, r.z); store(i, j + 3, r.w);
    }
    return;
  }
  __shared__ float part[kReduceWaves][64];
  const bool valid = e < (int64_t)sg.rows * cols_x;
  const int i = valid ? (int)(e / cols_x) : 0, j = valid ? (int)(e % cols_x) : 0;
  const float* src = a.slab + sg.src_off + (int64_t)i * sg.src_ld + j;
  float acc = 0.f;
  if (valid) {
    for (int s = s_begin; s < s_end; s += 16) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = src[(int64_t)min(s + q, s_end - 1) * sg.slab_stride];  // (clamped: branch-free batch)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += s + q < s_end ? v[q] : 0.f;
    }
  }
  part[sub][lane] = acc;
  __syncthreads();
  if (cut == 0 && valid) {
    float r = 0.f;
    for (int c = 0; c < n_cuts; ++c) r += part[c / cpw][(c % cpw) * epb + el];
    store(i, j, r);
  }
}

int launch_reduce(const ReduceArgs& a, int max_elems, hipStream_t s, const HeadReduceArgs* head) {
  (void)max_elems;
  ReduceMap map;
  int blocks = 0;
  for (int g = 0; g < a.n_segs; ++g) {
    const ReduceSeg& sg = a.segs[g];
    const int cols_x = sg.cols + (sg.bias_off1 ? 1 : 0);
    const int64_t elems = reduce_seg_vec(sg, a.slab) ? (int64_t)sg.rows * ((cols_x + 3) >> 2) : (int64_t)sg.rows * cols_x;
    const int epb = reduce_epb(sg.n_slabs);
    map.blk_begin[g] = blocks;
    blocks += (int)((elems + epb - 1) / epb);
  }
  for (int g = a.n_segs; g <= kMaxReduceSegs; ++g) map.blk_begin[g] = blocks;
  ProfScope prof_scope(kProfReduce, s);
  HeadReduceArgs h{};
  int logit_blocks = 0, vblocks = 0;
  if (head != nullptr) {
    h = *head;
    const int64_t total = (int64_t)2 * h.K * h.D * (h.n_in + 1);
    logit_blocks = (h.K * (h.n_in + 1) + 63) / 64;
    vblocks = logit_blocks + (int)std::min<int64_t>(std::max<int64_t>((total + 255) / 256, 1), 2048);
  }
  hipLaunchKernelGGL(k_reduce, dim3(1 + blocks + (vblocks + 3) / 4), dim3(kReduceWaves * 64), 0, s, a, map, h, logit_blocks, vblocks);
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
