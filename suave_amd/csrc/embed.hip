// First encoder layer, wide categorical columns: embedding gather (forward) and deterministic scatter
// (weight gradient).
//
// The reference feeds every categorical column to the encoder as a one-hot block
// (suave/model.py:2866-2870, :2905) and multiplies it by encoder.backbone.0.weight
// (suave/modules/encoder.py:49-53): W1[:, block] @ onehot(code) is column `code` of the block, or zero for the
// all-zero one-hot of a missing cell.  For BASELINE config #4 (224 categoricals of up to 256 classes) those
// blocks are 14 280 of the 16 629 input columns, so the dense product spends 86 % of its multiplies on zeros.
// Here the blocks of the wide columns (C >= SUAVE_WIDE_CAT_MIN_CLASSES) are stored transposed, [C, ld] per
// column (suave_col_t::enc_embed), and
//   forward   pre[b, :]  = sum_j E_j[code[b, j], :]              (k_embed_fwd; the dense GEMM then adds
//                                                                 x_dense @ W1_dense^T + b and applies ReLU/dropout)
//   backward  dE_j[c, :] = sum_{b : code[b, j] = c} dpre[b, :]   (k_embed_bwd)
// The scatter is deterministic: one wavefront owns one row of one block and adds the batch rows that selected
// it in row order, so there are no float atomics and no slabs.
#include "kernels.h"

namespace suave {

namespace {

// One wave = 4 batch rows at a time.  The rows' code vectors are staged in LDS (coalesced row reads of the
// packed table); then, per 128-output chunk, the wave walks the embedded columns and every lane accumulates
// one float2 of each of the 4 rows: 4 independent 512-byte row gathers in flight per column.
constexpr int kEmbRows = 4;

__global__ __launch_bounds__(256) void k_embed_fwd(EmbedFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) int16_t codes_s[];  // [4 waves][kEmbRows][n_disc]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int16_t* cs = codes_s + (size_t)wave * kEmbRows * a.n_disc;
  const int n_groups = (a.n_rows + kEmbRows - 1) / kEmbRows;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
    const int r0 = g * kEmbRows;
    // ---- stage the code vectors of the 4 rows ----
#pragma unroll
    for (int rr = 0; rr < kEmbRows; ++rr) {
      const int r = min(r0 + rr, a.n_rows - 1);
      const int64_t tr = a.batch.row_index ? (int64_t)a.batch.row_index[r] : a.batch.row_begin + r;
      const int16_t* src = a.batch.codes + tr * a.batch.ld_disc;
      for (int i = lane; i < a.n_disc; i += 64) cs[rr * a.n_disc + i] = src[i];
    }
    // (LDS operations of one wave execute in order: no barrier needed, the slice is wave-private)
    if (a.ecodes) {  // column-major copy of the embedded columns' codes for the weight-gradient kernel
      for (int j = lane; j < a.n_embed; j += 64) {
        const int xi = a.cols[j].x_index;
#pragma unroll
        for (int rr = 0; rr < kEmbRows; ++rr)
          if (r0 + rr < a.n_rows) a.ecodes[(size_t)j * a.n_rows + r0 + rr] = cs[rr * a.n_disc + xi];
      }
    }
    for (int o0 = 0; o0 < a.ld; o0 += 128) {
      const int o = o0 + 2 * lane;
      const bool in = o < a.ld;  // ld is a multiple of 4: the float2 never straddles the row end
      float2 acc[kEmbRows];
#pragma unroll
      for (int rr = 0; rr < kEmbRows; ++rr) acc[rr] = make_float2(0.f, 0.f);
      for (int j = 0; j < a.n_embed; ++j) {
        const EmbedCol ec = a.cols[j];  // wave-uniform (scalar loads)
        const float* blk = a.params + ec.e_off + (in ? o : 0);
#pragma unroll
        for (int rr = 0; rr < kEmbRows; ++rr) {
          const int code = cs[rr * a.n_disc + ec.x_index];  // LDS broadcast
          if (code >= 0 && code < ec.C) {                   // all-zero one-hot otherwise (model.py:2866-2870)
            const float2 v = *reinterpret_cast<const float2*>(blk + (size_t)code * a.ld);
            acc[rr].x += v.x;
            acc[rr].y += v.y;
          }
        }
      }
      if (in) {
#pragma unroll
        for (int rr = 0; rr < kEmbRows; ++rr)
          if (r0 + rr < a.n_rows) *reinterpret_cast<float2*>(a.pre + (size_t)(r0 + rr) * a.ld + o) = acc[rr];
      }
    }
  }
}

// One wave = one (embedded column j, class c) pair = one row of one embedding block.  The wave scans the
// column's codes (coalesced, 4 x 64 rows per step), compacts the rows that carry class c and adds their
// dpre rows in row order into registers (lane = 2 outputs per 128): no LDS accumulators, no atomics, no
// slabs, and the sum order is the row order, so the gradient is reproducible bit for bit.
template <int TT>
__global__ __launch_bounds__(64) void k_embed_bwd(EmbedBwdArgs a) {
  __shared__ int rows_s[4 * 64];
  const int lane = threadIdx.x;
  const int g = blockIdx.x;  // class index over all embedded columns
  int lo = 0, hi = a.n_embed;
  while (hi - lo > 1) {  // cls_begin[lo] <= g < cls_begin[lo + 1]   (wave-uniform)
    const int mid = (lo + hi) >> 1;
    if (a.cols[mid].cls_begin <= g) lo = mid; else hi = mid;
  }
  const EmbedCol ec = a.cols[lo];
  const int c = g - ec.cls_begin;
  const int16_t* codes = a.ecodes + (size_t)lo * a.n_rows;
  float2 acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) acc[t] = make_float2(0.f, 0.f);
  for (int r0 = 0; r0 < a.n_rows; r0 += 256) {
    int cv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q * 64 + lane;
      cv[q] = r < a.n_rows ? (int)codes[r] : -1;
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool hit = cv[q] == c;
      const unsigned long long bal = __ballot(hit);
      if (hit) rows_s[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = r0 + q * 64 + lane;
      cnt += __popcll(bal);
    }
    for (int i0 = 0; i0 < cnt; i0 += 4) {
      float2 v[4][TT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = rows_s[min(i0 + u, cnt - 1)];  // LDS broadcast; same wave wrote it (in-order DS)
        const float* src = a.dpre + (size_t)row * a.ld;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const int o = 2 * lane + 128 * t;
          v[u][t] = o < a.ld ? *reinterpret_cast<const float2*>(src + o) : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u < cnt) {
#pragma unroll
          for (int t = 0; t < TT; ++t) { acc[t].x += v[u][t].x; acc[t].y += v[u][t].y; }
        }
    }
  }
  float* dst = a.grads + ec.e_off + (size_t)c * a.ld;
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int o = 2 * lane + 128 * t;
    if (o < a.ld) *reinterpret_cast<float2*>(dst + o) = acc[t];
  }
}

}  // namespace

int launch_embed_fwd(const EmbedFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0 || a.n_embed <= 0) return 0;
  SUAVE_REQUIRE(a.ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.pre) & 15) == 0, "embed: output must be 16-byte aligned");
  const size_t lds = (size_t)4 * kEmbRows * a.n_disc * sizeof(int16_t);
  SUAVE_REQUIRE(lds <= 64 * 1024, "embed: too many discrete columns for the code staging buffer");
  ProfScope prof_scope(kProfGemm, s);
  const int n_groups = (a.n_rows + kEmbRows - 1) / kEmbRows;
  const int grid = std::max(1, std::min((n_groups + 3) / 4, 8192));
  hipLaunchKernelGGL(k_embed_fwd, dim3(grid), dim3(256), lds, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_embed_bwd(const EmbedBwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0 || a.n_embed <= 0 || a.n_classes_total <= 0) return 0;
  SUAVE_REQUIRE(a.ld % 4 == 0 && a.ld <= 1024, "embed: first hidden layer wider than 1024 is not supported with embedded columns");
  ProfScope prof_scope(kProfGemm, s);
  const dim3 grid(a.n_classes_total), block(64);
  const int tt = (a.ld + 127) / 128;
  if (tt <= 1) hipLaunchKernelGGL(k_embed_bwd<1>, grid, block, 0, s, a);
  else if (tt <= 2) hipLaunchKernelGGL(k_embed_bwd<2>, grid, block, 0, s, a);
  else if (tt <= 4) hipLaunchKernelGGL(k_embed_bwd<4>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_embed_bwd<8>, grid, block, 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
