// The classifier head of behaviour="supervised" on the device: ClassificationHead without hidden layers
// (suave/modules/heads.py:62-91: dropout on the latent inputs, `linear`, class-weighted cross-entropy) as used by
//   * _train_head_phase (suave/model.py:1743-1864): Adam on the head over cached posterior means, and
//   * _run_joint_finetune (suave/model.py:1973-1999): weight * CE(classifier(outputs["latent"])) added to the ELBO
//     loss, whose gradient reaches the encoder through `latent`.
// One launch scores a batch: logits, weighted CE, d CE / d latent (added to the ELBO step's dz) and per-block partial
// sums of the weight / bias gradients; a one-block launch sums the partials in block order (deterministic), scales
// them, accumulates the epoch's loss and, on request, applies torch.optim.Adam's update to the head's parameters.
//
// Layout of a tile of R rows in LDS: x [D][R + 1] (the dropped-out inputs, column = row: phase B reads down a column,
// phase C along two rows), g [C][R] (logits, then the weighted softmax residuals), W [C][D], b [C], and the block's
// gradient accumulators [C D + C].
#include <algorithm>
#include <cmath>

#include "kernels.h"

namespace suave {

namespace {

struct HeadArgs {
  const float* latent;      // [*, ld]
  int ld, D, C;
  int latent_by_row_id;     // 1: row i of the batch reads latent[row id] (a cached table); 0: latent[i]
  const int32_t* row_index; // [n] row ids, or null: row_begin + i
  int64_t row_begin, n;
  const int32_t* targets;   // class of every TABLE row (indexed by row id)
  const float* class_weight;  // [C] or null
  const float* params;      // linear.weight [C, D] row-major, then linear.bias [C]
  float drop_p;
  uint32_t drop_key;
  const float* weight_sum;  // device scalar: sum of class weights over the GLOBAL batch (null: not needed / not known)
  float loss_scale;
  float* dz;                // [n, ldz] += loss_scale * d CE / d latent, or null
  int ldz;
  float* logits_out;        // [n, C] or null
  float* slab;              // [grid][C D + C + 2]
  int R;                    // rows per tile == blockDim.x
};

// supervised.RowKeyedDropout's hash (lowbias32): keep decision of (table row, feature) under (key, layer 0)
__device__ __forceinline__ uint32_t h32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool head_keep(uint32_t row, uint32_t key, int d, float p) {
  const uint32_t r = h32(row * 0x9E3779B1u + key);
  const uint32_t c = (uint32_t)d * 0x632BE5ABu + 0x165667B1u;
  const uint32_t h = h32(r ^ c);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

__global__ __launch_bounds__(256) void k_head_loss(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = a.D, C = a.C, R = a.R, ldx = R + 1, CD = C * D;
  float* W_s = lds;                 // [C][D] then b [C]
  float* acc_s = W_s + CD + C;      // [C D + C]
  float* x_s = acc_s + CD + C;      // [D][ldx]
  float* g_s = x_s + D * ldx;       // [C][R]
  float* red_s = g_s + C * R;       // [2][R / 64]
  const int tid = threadIdx.x;
  for (int i = tid; i < CD + C; i += R) { W_s[i] = a.params[i]; acc_s[i] = 0.f; }
  const bool drop = a.drop_p > 0.f;
  const float keep_scale = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const float inv_wsum = a.weight_sum ? a.loss_scale / *a.weight_sum : 0.f;
  float lsum = 0.f, wsum = 0.f;
  const int64_t n_tiles = (a.n + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * R;
    const int nrows = (int)min<int64_t>(R, a.n - r0);
    __syncthreads();  // the previous tile's readers are done (and W_s / acc_s are initialised)
    // A: the tile's inputs, dropped out, transposed into LDS
    for (int e = tid; e < R * D; e += R) {
      const int i = e / D, d = e - i * D;
      float v = 0.f;
      if (i < nrows) {
        const int64_t rid = a.row_index ? (int64_t)a.row_index[r0 + i] : a.row_begin + r0 + i;
        const int64_t src = a.latent_by_row_id ? rid : r0 + i;
        v = a.latent[src * a.ld + d];
        if (drop) v = head_keep((uint32_t)rid, a.drop_key, d, a.drop_p) ? v * keep_scale : 0.f;
      }
      x_s[d * ldx + i] = v;
    }
    __syncthreads();
    // B: one thread per row -- logits, softmax, weighted CE, residuals g = w_y (p - onehot)
    {
      const int i = tid;
      int64_t rid = 0;
      int y = 0;
      float wy = 0.f;
      if (i < nrows) {
        rid = a.row_index ? (int64_t)a.row_index[r0 + i] : a.row_begin + r0 + i;
        y = a.targets[rid];
        wy = a.class_weight ? a.class_weight[y] : 1.0f;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) {
          float l = W_s[CD + c];
          const float* wr = W_s + c * D;
          for (int d = 0; d < D; ++d) l = fmaf(wr[d], x_s[d * ldx + i], l);
          g_s[c * R + i] = l;
          mx = fmaxf(mx, l);
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(g_s[c * R + i] - mx);
        const float lse = mx + __logf(se), inv = 1.0f / se;
        lsum += wy * (lse - g_s[y * R + i]);
        wsum += wy;
        for (int c = 0; c < C; ++c) {
          const float l = g_s[c * R + i];
          if (a.logits_out) a.logits_out[(r0 + i) * C + c] = l;
          g_s[c * R + i] = wy * (__expf(l - mx) * inv - (c == y ? 1.0f : 0.f));
        }
      } else {
        for (int c = 0; c < C; ++c) g_s[c * R + i] = 0.f;
      }
      // D (same thread, its own column of g): loss_scale / sum_w * W^T g through the dropout mask, added to dz
      if (a.dz != nullptr && i < nrows) {
        for (int d = 0; d < D; ++d) {
          float dl = 0.f;
          for (int c = 0; c < C; ++c) dl = fmaf(g_s[c * R + i], W_s[c * D + d], dl);
          if (drop) dl = head_keep((uint32_t)rid, a.drop_key, d, a.drop_p) ? dl * keep_scale : 0.f;
          a.dz[(r0 + i) * a.ldz + d] += dl * inv_wsum;
        }
      }
    }
    __syncthreads();
    // C: the tile's share of g^T x (weight) and of the column sums of g (bias); entry q is owned by one thread
    for (int q = tid; q < CD + C; q += R) {
      float s = 0.f;
      if (q < CD) {
        const int c = q / D, d = q - c * D;
        const float* gr = g_s + c * R;
        const float* xr = x_s + d * ldx;
        for (int i = 0; i < R; ++i) s = fmaf(gr[i], xr[i], s);
      } else {
        const float* gr = g_s + (q - CD) * R;
        for (int i = 0; i < R; ++i) s += gr[i];
      }
      acc_s[q] += s;
    }
  }
  __syncthreads();
  // the block's partial sums, in a fixed order: waves by shuffles, then thread 0 over the waves
  for (int off = 32; off > 0; off >>= 1) {
    lsum += __shfl_down(lsum, off, 64);
    wsum += __shfl_down(wsum, off, 64);
  }
  const int nw = R / 64;
  if ((tid & 63) == 0) { red_s[tid >> 6] = lsum; red_s[nw + (tid >> 6)] = wsum; }
  __syncthreads();
  float* out = a.slab + (size_t)blockIdx.x * (CD + C + 2);
  for (int q = tid; q < CD + C; q += R) out[q] = acc_s[q];
  if (tid == 0) {
    float l = 0.f, w = 0.f;
    for (int k = 0; k < nw; ++k) { l += red_s[k]; w += red_s[nw + k]; }
    out[CD + C] = l;
    out[CD + C + 1] = w;
  }
}

struct HeadFinishArgs {
  const float* slab;
  int n_slabs, len;          // len = C D + C
  const float* weight_sum;   // device scalar or null (then: the sum over this call's rows)
  float loss_scale, loss_rows;
  float* grads;              // [len] or null
  double* loss_sum;          // += CE * loss_rows, or null
  // torch.optim.Adam (null params: no update)
  float* params; float* m; float* v;
  float step_size, beta1, beta2, eps, inv_sqrt_bc2;
};

__global__ __launch_bounds__(256) void k_head_finish(HeadFinishArgs a) {
  __shared__ float tot[2];
  const int stride = a.len + 2;
  if (threadIdx.x == 0) {
    float l = 0.f, w = 0.f;
    for (int s = 0; s < a.n_slabs; ++s) { l += a.slab[(size_t)s * stride + a.len]; w += a.slab[(size_t)s * stride + a.len + 1]; }
    tot[0] = l;
    tot[1] = a.weight_sum ? *a.weight_sum : w;
    if (a.loss_sum) *a.loss_sum += (double)(tot[1] > 0.f ? l / tot[1] : 0.f) * (double)a.loss_rows;
  }
  __syncthreads();
  const float scale = tot[1] > 0.f ? a.loss_scale / tot[1] : 0.f;
  for (int q = threadIdx.x; q < a.len; q += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < a.n_slabs; ++k) s += a.slab[(size_t)k * stride + q];
    const float g = s * scale;
    if (a.grads) a.grads[q] = g;
    if (a.params) {
      const float m = a.beta1 * a.m[q] + (1.0f - a.beta1) * g;
      const float v = a.beta2 * a.v[q] + (1.0f - a.beta2) * g * g;
      a.m[q] = m;
      a.v[q] = v;
      a.params[q] -= a.step_size * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
    }
  }
}

__global__ __launch_bounds__(256) void k_head_weight_sum(const int32_t* targets, const int32_t* row_index, int64_t row_begin,
                                                         int64_t n, const float* class_weight, float* out) {
  __shared__ float part[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int64_t rid = row_index ? (int64_t)row_index[i] : row_begin + i;
    s += class_weight ? class_weight[targets[rid]] : 1.0f;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = ((part[0] + part[1]) + part[2]) + part[3];
}

}  // namespace

// rows per tile / LDS bytes of k_head_loss for a head of C classes on D inputs; 0 rows: outside the kernel's shapes
// (small batches take 64-row tiles whatever fits: a 256-row batch is then four blocks of short loops instead of one)
static int head_tile_rows(int C, int D, size_t* lds_bytes, int64_t n = int64_t(1) << 40) {
  for (int R : {256, 128, 64}) {
    if (n <= 8192 && R > 64) continue;
    const size_t b = sizeof(float) * (2 * (size_t)(C * D + C) + (size_t)D * (R + 1) + (size_t)C * R + 2 * (R / 64));
    if (b <= 144 * 1024) { *lds_bytes = b; return R; }
  }
  return 0;
}

constexpr int kHeadMaxBlocks = 64;

bool head_shape_ok(int C, int D) {
  size_t b;
  return C >= 2 && C <= 64 && D >= 1 && D <= 256 && head_tile_rows(C, D, &b) > 0;
}

size_t head_workspace_floats(int C, int D) { return (size_t)kHeadMaxBlocks * (C * D + C + 2) + 4; }

int launch_head_loss(const suave_head_t& h, int D, const float* latent, int ld, int latent_by_row_id, const int32_t* row_index,
                     int64_t row_begin, int64_t n, float* dz, int ldz, float* logits_out, hipStream_t s) {
  SUAVE_REQUIRE(head_shape_ok(h.n_classes, D), "classifier head: n_classes must be 2..64 and latent_dim <= 256 (C * D within LDS)");
  SUAVE_REQUIRE(h.params && h.targets && h.workspace, "classifier head: params / targets / workspace is null");
  SUAVE_REQUIRE(h.workspace_bytes >= sizeof(float) * head_workspace_floats(h.n_classes, D), "classifier head: workspace too small (suave_head_workspace_bytes)");
  SUAVE_REQUIRE(h.dropout >= 0.f && h.dropout < 1.f, "classifier head: dropout must be in [0, 1)");
  const int C = h.n_classes;
  size_t lds = 0;
  const int R = head_tile_rows(C, D, &lds, n);
  float* wsp = static_cast<float*>(h.workspace);
  float* wsum_cell = wsp + (size_t)kHeadMaxBlocks * (C * D + C + 2);
  const float* wsum = h.weight_sum;
  if (n > 0 && dz != nullptr && wsum == nullptr) {  // d latent needs the batch's weight sum before the rows are scored
    hipLaunchKernelGGL(k_head_weight_sum, dim3(1), dim3(256), 0, s, h.targets, row_index, row_begin, n, h.class_weight, wsum_cell);
    SUAVE_CHECK_HIP(hipGetLastError());
    wsum = wsum_cell;
  }
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + R - 1) / R, kHeadMaxBlocks));
  HeadArgs a{};
  a.latent = latent; a.ld = ld; a.D = D; a.C = C; a.latent_by_row_id = latent_by_row_id; a.row_index = row_index;
  a.row_begin = row_begin; a.n = n; a.targets = h.targets; a.class_weight = h.class_weight; a.params = h.params;
  a.drop_p = h.training ? h.dropout : 0.f; a.drop_key = (uint32_t)h.dropout_key; a.weight_sum = wsum;
  a.loss_scale = h.loss_scale; a.dz = dz; a.ldz = ldz; a.logits_out = logits_out; a.slab = wsp; a.R = R;
  if (lds > 64 * 1024)  // (beyond the default dynamic-LDS limit: wide heads only)
    SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_loss), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  hipLaunchKernelGGL(k_head_loss, dim3(grid), dim3(R), lds, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  HeadFinishArgs f{};
  f.slab = wsp; f.n_slabs = grid; f.len = C * D + C; f.weight_sum = wsum; f.loss_scale = h.loss_scale;
  f.loss_rows = h.loss_rows; f.grads = h.grads; f.loss_sum = h.loss_sum;
  if (h.adam_lr > 0.f) {
    SUAVE_REQUIRE(h.adam_m && h.adam_v && h.adam_step >= 1, "classifier head: Adam needs its moment buffers and a 1-based step");
    const double b1 = 0.9, b2 = 0.999;  // torch.optim.Adam defaults (model.py:1771)
    const double bc1 = 1.0 - std::pow(b1, (double)h.adam_step), bc2 = 1.0 - std::pow(b2, (double)h.adam_step);
    f.params = h.params; f.m = h.adam_m; f.v = h.adam_v;
    f.step_size = (float)((double)h.adam_lr / bc1); f.beta1 = (float)b1; f.beta2 = (float)b2; f.eps = 1e-8f;
    f.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  }
  hipLaunchKernelGGL(k_head_finish, dim3(1), dim3(256), 0, s, f);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
