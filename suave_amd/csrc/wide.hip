// Row-wise kernels of the wide categorical columns (SUAVE_WIDE_CAT_MIN_CLASSES and more classes).
//
// The reference's CatHead (suave/modules/decoder.py:194-222) applies one bias-free linear to the feature
// vector [y_col ; s] and scores the one-hot target with log_softmax (distributions.py:57-64).  For a wide
// column that linear is an MFMA GEMM (gemm.hip, gemm_wide_kernel) on the column's slot of the internal y
// buffer; the kernels here write the assignment part of the slots, turn the logits into log-likelihoods and
// (GRAD) into d loss / d logits in place, and collect the assignment gradient out of the slots afterwards.
#include <algorithm>

#include "kernels.h"

namespace suave {

// slot[r, yd .. yd+K) = s[r, :]: one-hot of the row's component (ELBO step, component-indexed decode) or the
// dense assignment vector (soft imputation, model.py:3635-3644).  Lane = (wide column, j) of one row, so a store
// instruction writes runs of K consecutive floats (a thread per (row, column) wrote 64 scattered floats per
// instruction); the slots' offsets come from an LDS table.
__global__ __launch_bounds__(256) void k_wide_fill(WideRowArgs a) {
  extern __shared__ int soff_s[];  // [n_wide] offset of the assignment part of the column's slot
  for (int i = threadIdx.x; i < a.n_wide; i += 256) soff_s[i] = a.wide[i].y_off + a.wide[i].yd;
  __syncthreads();
  const int per_row = a.n_wide * a.K;
  for (int r = blockIdx.x; r < a.n_rows; r += gridDim.x) {
    float* yrow = a.y + (size_t)r * a.ldY;
    const int k = a.s_dense ? 0 : (a.kidx ? a.kidx[r] : (r % a.vdiv));
    for (int i = threadIdx.x; i < per_row; i += 256) {
      const int c = i / a.K, j = i - c * a.K;
      yrow[soff_s[c] + j] = a.s_dense ? a.s_dense[(size_t)r * a.K + j] : (j == k ? 1.f : 0.f);
    }
  }
}

// One wavefront per row walks the wide columns: lanes stride over the classes, two wave reductions give the
// log-sum-exp (logit 0 is pinned to 0, decoder.py:207-210).  recon_out[r] = sum over the wide columns of
// log_px; with GRAD the logits are overwritten by g_c = up * (onehot_c - softmax_c) (exactly 0 for masked
// cells and all-zero one-hot rows, as `_apply_observed_linear` and the x mask give, decoder.py:41-53).
template <bool GRAD>
__global__ __launch_bounds__(256) void k_wide_softmax(WideRowArgs a) {
  __shared__ double blk_sum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double wsum = 0.0;
  for (int r = blockIdx.x * 4 + wave; r < a.n_rows; r += gridDim.x * 4) {
    const int drow = a.vdiv == 1 ? r : r / a.vdiv;
    const int32_t* cell_ptr = a.cells + ((size_t)(drow >> 6) * a.n_cols) * 64 + (drow & 63);
    const float up = GRAD ? a.c_up * (a.row_weight ? a.row_weight[r] : 1.0f) : 0.f;
    float* Lrow = a.L + (size_t)r * a.ldL;
    float recon = 0.f;
    for (int c = 0; c < a.n_wide; ++c) {
      const WideCol wc = a.wide[c];
      const int cell = cell_ptr[(size_t)wc.col * 64];
      const bool obs = cell != kCellMissing;
      const int code = obs ? cell : -1;
      const bool live = code >= 0 && code < wc.C;
      float* Lc = Lrow + wc.l_off;
      const int n = wc.C - 1;
      float mx = 0.f;
      for (int i = lane; i < n; i += 64) mx = fmaxf(mx, Lc[i]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float se = 0.f;
      for (int i = lane; i < n; i += 64) se += __expf(Lc[i] - mx);
      se = wave_sum(se) + __expf(-mx);
      const float lse = mx + __logf(se);
      const float lp = live ? (code >= 1 ? Lc[code - 1] : 0.f) - lse : 0.f;  // (masked cell: all-zero one-hot, 0)
      recon += lp;
      if (lane == 0 && a.log_px) a.log_px[(size_t)r * a.n_cols + wc.col] = lp;
      if (lane == 0 && a.log_px_missing) a.log_px_missing[(size_t)r * a.n_cols + wc.col] = 0.f;
      if (GRAD) {
        for (int i = lane; i < n; i += 64) {
          const float p = __expf(Lc[i] - lse);
          Lc[i] = live ? up * ((i + 1 == code ? 1.0f : 0.f) - p) : 0.f;
        }
      }
    }
    if (lane == 0) {
      a.recon_out[r] = recon;
      wsum += (double)recon;
    }
  }
  if (a.sums_slab) {
    if (lane == 0) blk_sum[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double* s = a.sums_slab + (size_t)blockIdx.x * 4;
      s[0] = ((blk_sum[0] + blk_sum[1]) + blk_sum[2]) + blk_sum[3];
      s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    }
  }
}

// The same, register-resident and software-pipelined (every wide column of the schema has at most 1024 * NS / 4
// classes): a lane owns classes 4 lane .. 4 lane + 3 of each 256-class slot, so a column's logits are ONE 16-byte
// load per lane and slot, read once, kept in registers for the maximum, the exponential sum and the gradient, and
// written once (the first kernel read them three times, each a dependent round trip: 4.2 us per row and column).
// The next column's descriptor, cell, logits and observed-class logit are loaded before the current column's
// reductions start.  Pad classes of a column's 4-aligned slot are written as zeros.
template <bool GRAD, int NS>
__global__ __launch_bounds__(256) void k_wide_softmax_reg(WideRowArgs a) {
  __shared__ double blk_sum[4];
  extern __shared__ int wtab_s[];  // [n_wide][3]: schema column, classes, logits offset
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < a.n_wide; i += 256) {
    wtab_s[3 * i] = a.wide[i].col;
    wtab_s[3 * i + 1] = a.wide[i].C;
    wtab_s[3 * i + 2] = a.wide[i].l_off;
  }
  __syncthreads();
  const float4 ninf4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  double wsum = 0.0;
  for (int r = blockIdx.x * 4 + wave; r < a.n_rows; r += gridDim.x * 4) {
    const int drow = a.vdiv == 1 ? r : r / a.vdiv;
    const int32_t* cell_ptr = a.cells + ((size_t)(drow >> 6) * a.n_cols) * 64 + (drow & 63);
    const float up = GRAD ? a.c_up * (a.row_weight ? a.row_weight[r] : 1.0f) : 0.f;
    float* Lrow = a.L + (size_t)r * a.ldL;
    float recon = 0.f;
    // column 0 in flight; every trip sends the next column's loads before its own reductions (the last trip re-reads
    // its own column).  Two columns ahead was slower (1.65 vs 1.42 ms at W1024): the kernel is then on the memory
    // system, 4.4 GB at 3.1 TB/s of mixed reads and writes in 600-byte pieces.
    int col, C, loff, code, ncol, nC, nloff, ncode;
    float xs, nxs;
    float4 cur[NS], nxt[NS];
#define SUAVE_WIDE_FETCH(cc, col_, C_, loff_, code_, xs_, buf_)                                          \
    {                                                                                                    \
      const int ci_ = min((cc), a.n_wide - 1);                                                           \
      col_ = wtab_s[3 * ci_]; C_ = wtab_s[3 * ci_ + 1]; loff_ = wtab_s[3 * ci_ + 2];                      \
      const int cell_ = cell_ptr[(size_t)col_ * 64];                                                     \
      code_ = cell_ != kCellMissing ? cell_ : -1;                                                        \
      xs_ = (code_ >= 1 && code_ < C_) ? Lrow[loff_ + code_ - 1] : 0.f;                                  \
      _Pragma("unroll") for (int sl = 0; sl < NS; ++sl) {                                                \
        const int i4 = 256 * sl + 4 * lane;                                                              \
        buf_[sl] = i4 < C_ - 1 ? *reinterpret_cast<const float4*>(Lrow + loff_ + i4) : ninf4;            \
      }                                                                                                  \
    }
    SUAVE_WIDE_FETCH(0, col, C, loff, code, xs, cur)
    for (int c = 0; c < a.n_wide; ++c) {
      SUAVE_WIDE_FETCH(c + 1, ncol, nC, nloff, ncode, nxs, nxt)
      const bool live = code >= 0 && code < C;
      const int n = C - 1;
      float v[NS][4];
      float mx = 0.f;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const int i4 = 256 * sl + 4 * lane;
        v[sl][0] = cur[sl].x; v[sl][1] = cur[sl].y; v[sl][2] = cur[sl].z; v[sl][3] = cur[sl].w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (i4 + e >= n) v[sl][e] = -INFINITY;  // pad classes of the last quad
          mx = fmaxf(mx, v[sl][e]);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      float se = 0.f;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += __expf(v[sl][e] - mx);
      se = wave_sum(se) + __expf(-mx);
      const float lse = mx + __logf(se);
      const float lp = live ? xs - lse : 0.f;  // (class 0: logit 0; masked cell: all-zero one-hot, 0)
      recon += lp;
      if (lane == 0 && a.log_px) a.log_px[(size_t)r * a.n_cols + col] = lp;
      if (lane == 0 && a.log_px_missing) a.log_px_missing[(size_t)r * a.n_cols + col] = 0.f;
      if (GRAD) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          const int i4 = 256 * sl + 4 * lane;
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __expf(v[sl][e] - lse);
            g[e] = (live && i4 + e < n) ? up * ((i4 + e + 1 == code ? 1.0f : 0.f) - p) : 0.f;
          }
          if (i4 < n) *reinterpret_cast<float4*>(Lrow + loff + i4) = make_float4(g[0], g[1], g[2], g[3]);
        }
      }
      col = ncol; C = nC; loff = nloff; code = ncode; xs = nxs;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) cur[sl] = nxt[sl];
    }
#undef SUAVE_WIDE_FETCH
    if (lane == 0) {
      a.recon_out[r] = recon;
      wsum += (double)recon;
    }
  }
  if (a.sums_slab) {
    if (lane == 0) blk_sum[wave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double* s = a.sums_slab + (size_t)blockIdx.x * 4;
      s[0] = ((blk_sum[0] + blk_sum[1]) + blk_sum[2]) + blk_sum[3];
      s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    }
  }
}

// After the data-gradient GEMM the slots hold d loss / d [y_col ; s].  The assignment part is summed over
// the wide columns into ds_out[r, 0..K) (the straight-through path needs the full K-vector, SURVEY.md
// Appendix A) and cleared, so that the slots' pad rows of y_projection see a zero gradient.  A block owns a row:
// thread (column group g, j) adds its columns in order, eight loads in flight before the eight clearing stores
// (one load -> store round trip per column was a 112-deep latency chain), and the groups are combined in a fixed
// order.
__global__ __launch_bounds__(256) void k_wide_ds(WideRowArgs a) {
  extern __shared__ int soff_s[];  // [n_wide] slot offsets | [G][K] partial sums (as floats)
  float* part_s = reinterpret_cast<float*>(soff_s + a.n_wide);
  for (int i = threadIdx.x; i < a.n_wide; i += 256) soff_s[i] = a.wide[i].y_off + a.wide[i].yd;
  __syncthreads();
  const int G = 256 / a.K;                       // column groups (K <= 32)
  const int g = threadIdx.x / a.K, j = threadIdx.x - g * a.K;
  for (int r = blockIdx.x; r < a.n_rows; r += gridDim.x) {
    float* yrow = a.y + (size_t)r * a.ldY;
    float t = 0.f;
    if (g < G) {
      for (int c0 = g; c0 < a.n_wide; c0 += 8 * G) {
        float v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + u * G;
          off[u] = c < a.n_wide ? soff_s[c] + j : -1;
          v[u] = off[u] >= 0 ? yrow[off[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          t += v[u];
          if (off[u] >= 0) yrow[off[u]] = 0.f;
        }
      }
      part_s[g * a.K + j] = t;
    }
    __syncthreads();
    if (threadIdx.x < a.K && a.ds_out) {
      float tot = 0.f;
      for (int q = 0; q < G; ++q) tot += part_s[q * a.K + threadIdx.x];
      a.ds_out[(size_t)threadIdx.x * a.n_rows + r] = tot;  // plane-major [j][row], as the likelihood kernel's chunks
    }
    __syncthreads();
  }
}

// Row-major transposed copy of every wide column's logits_layer weight (the split data-gradient GEMM multiplies
// dL [rows, C-1] by W^T stored [ldw, round_up(C-1, 4)]).  blockIdx.y = wide column; reads are coalesced.
__global__ __launch_bounds__(256) void k_wide_transpose(const WideCol* __restrict__ cols, const float* __restrict__ params,
                                                        float* __restrict__ wT) {
  const WideCol wc = cols[blockIdx.y];
  const int kc = wc.C - 1, ldt = (kc + 3) & ~3;
  float* dst = wT + wc.t_off;
  const int total = ldt * wc.ldw;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = e / wc.ldw, n = e - k * wc.ldw;
    dst[(size_t)n * ldt + k] = k < kc ? params[wc.w_off + (size_t)k * wc.ldw + n] : 0.f;
  }
}

static int wide_grid(int64_t work, int per_block, int cap) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((work + per_block - 1) / per_block, cap));
}

int launch_wide_fill(const WideRowArgs& a, hipStream_t s) {
  if (a.n_rows <= 0 || a.n_wide <= 0) return 0;
  ProfScope prof_scope(kProfLoss, s);
  SUAVE_REQUIRE(sizeof(int) * (size_t)a.n_wide <= 48 * 1024, "wide columns: offset table does not fit in LDS");
  hipLaunchKernelGGL(k_wide_fill, dim3(std::min(a.n_rows, 8192)), dim3(256), sizeof(int) * a.n_wide, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int wide_softmax_grid(int n_rows) { return wide_grid(n_rows, 4, 4096); }

int launch_wide_softmax(const WideRowArgs& a, bool grad, int max_classes, hipStream_t s) {
  if (a.n_rows <= 0 || a.n_wide <= 0) return 0;
  ProfScope prof_scope(kProfLoss, s);
  const dim3 grid(wide_softmax_grid(a.n_rows)), block(256);
  const int nmax = a.n_wide <= 4000 ? max_classes - 1 : 1 << 30;  // logits per column (column table: 12 bytes each in LDS)
  const size_t tab = sizeof(int) * 3 * a.n_wide;
#define SUAVE_WIDE_SOFTMAX(NS)                                                              \
  do {                                                                                      \
    if (grad) hipLaunchKernelGGL((k_wide_softmax_reg<true, NS>), grid, block, tab, s, a);   \
    else hipLaunchKernelGGL((k_wide_softmax_reg<false, NS>), grid, block, tab, s, a);       \
  } while (0)
  if (nmax <= 256) SUAVE_WIDE_SOFTMAX(1);
  else if (nmax <= 512) SUAVE_WIDE_SOFTMAX(2);
  else if (nmax <= 1024) SUAVE_WIDE_SOFTMAX(4);
  else if (grad) hipLaunchKernelGGL(k_wide_softmax<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_wide_softmax<false>, grid, block, 0, s, a);
#undef SUAVE_WIDE_SOFTMAX
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_wide_transpose(const WideCol* cols, int n_wide, const float* params, float* wT, hipStream_t s) {
  if (n_wide <= 0) return 0;
  ProfScope prof_scope(kProfGemm, s);
  hipLaunchKernelGGL(k_wide_transpose, dim3(32, n_wide), dim3(256), 0, s, cols, params, wT);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_wide_ds(const WideRowArgs& a, hipStream_t s) {
  if (a.n_rows <= 0 || a.n_wide <= 0) return 0;
  ProfScope prof_scope(kProfLoss, s);
  const size_t lds = sizeof(int) * a.n_wide + sizeof(float) * (256 / a.K) * a.K;
  SUAVE_REQUIRE(lds <= 48 * 1024, "wide columns: offset table does not fit in LDS");
  hipLaunchKernelGGL(k_wide_ds, dim3(std::min(a.n_rows, 8192)), dim3(256), lds, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
