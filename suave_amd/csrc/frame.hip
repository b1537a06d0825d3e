// Data formats either side of the hot path.
//
// k_pack_columns: raw table columns (one device array per column, as a DataFrame holds them) -> the packed table
//   (float32 numerics, int16 codes, uint8 mask) with the training normalisation fused in.  Replaces the host path of
//   suave/model.py:2784-2922 (_prepare_training_tensors), :3116-3213 (_apply_training_normalization) and
//   suave/data.py:133-266 (standardize) -- no one-hot / thermometer tensors, no per-row loop (model.py:2890-2891).
//   The arithmetic is the reference's: float64 (x - mean) / std, log1p, log, then one rounding to float32.
// k_unpack_columns: decode_sample's row-major outputs -> columnar blocks a DataFrame can wrap without copies
//   (float32 for real / pos, int64 for counts after rint / clamp, int8 / int16 category codes):
//   sampling.decoder_outputs_to_frame (suave/sampling.py:138-179).
#include <algorithm>

#include "kernels.h"

namespace suave {

namespace {

__device__ __forceinline__ double load_raw(const void* src, int dtype, int64_t i) {
  switch (dtype) {
    case SUAVE_RAW_F64: return static_cast<const double*>(src)[i];
    case SUAVE_RAW_F32: return static_cast<double>(static_cast<const float*>(src)[i]);
    case SUAVE_RAW_I64: return static_cast<double>(static_cast<const int64_t*>(src)[i]);
    case SUAVE_RAW_I32: return static_cast<double>(static_cast<const int32_t*>(src)[i]);
    case SUAVE_RAW_I16: return static_cast<double>(static_cast<const int16_t*>(src)[i]);
    default: return static_cast<double>(static_cast<const int8_t*>(src)[i]);
  }
}

constexpr double kEps = 1e-6;  // data.py:21 (_EPS)

// A block owns `rows_per_block` rows.  Wave w walks the columns w, w + 4, ...; lane = row, so the raw reads are
// coalesced along the column; the cells go to an LDS image of the packed rows and leave as contiguous row segments.
__global__ __launch_bounds__(256) void k_pack_columns(const suave_pack_col_t* __restrict__ cols, int n_cols, int n_num,
                                                      int n_disc, int64_t n, int rows_per_block, float* __restrict__ xnum,
                                                      int ld_num, int16_t* __restrict__ codes, int ld_disc,
                                                      uint8_t* __restrict__ mask, int ld_mask,
                                                      unsigned long long* __restrict__ rejected) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
  float* xn_s = reinterpret_cast<float*>(stage);                                      // [R][n_num]
  int16_t* cd_s = reinterpret_cast<int16_t*>(xn_s + (size_t)rows_per_block * n_num);  // [R][n_disc]
  uint8_t* mk_s = reinterpret_cast<uint8_t*>(cd_s + (size_t)rows_per_block * ((n_disc + 1) & ~1));  // [R][n_cols]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = (n_disc + 1) & ~1;
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block; r0 < n; r0 += (int64_t)gridDim.x * rows_per_block) {
    const int nrows = (int)min<int64_t>(rows_per_block, n - r0);
    __syncthreads();
    for (int c = wave; c < n_cols; c += 4) {
      const suave_pack_col_t col = cols[c];
      unsigned long long bad_here = 0;
      for (int lr = lane; lr < nrows; lr += 64) {
        const double x = load_raw(col.src, col.src_dtype, r0 + lr);
        bool obs = false;
        if (col.kind <= SUAVE_TYPE_COUNT) {
          double v;
          if (col.kind == SUAVE_TYPE_REAL) {
            v = (x - col.a) / fmax(col.b, kEps);
            if (fabs(v) > 60000.0 && isfinite(v)) ++bad_here;  // not rejected: counted (beyond the fp16 planes' range)
          } else if (col.kind == SUAVE_TYPE_POS) {
            if (x < -1.0 + kEps) ++bad_here;  // the reference raises: must be >= -1 to apply log1p
            v = (log1p(x) - col.a) / fmax(col.b, kEps);
          } else {
            if (x < 0.0) ++bad_here;  // the reference raises: counts are non-negative
            v = log(x + col.a);
          }
          obs = isfinite(v);
          xn_s[lr * n_num + col.slot] = obs ? static_cast<float>(v) : 0.f;
        } else if (col.kind == SUAVE_TYPE_CAT) {  // src holds category codes (-1: missing / unseen)
          const int code = (x >= 0.0 && x < static_cast<double>(col.n_classes)) ? static_cast<int>(x) : -1;
          obs = code >= 0;
          cd_s[lr * ldd + col.slot] = static_cast<int16_t>(code);
        } else {  // ordinal levels: out of range -> missing (the reference warns), truncation as astype(int64)
          const bool is_nan = x != x;
          const bool bad = !is_nan && (x < 0.0 || x > static_cast<double>(col.n_classes - 1));
          if (bad) ++bad_here;
          obs = !is_nan && !bad;
          const int lvl = obs ? min(max(static_cast<int>(x), 0), col.n_classes - 1) : -1;
          cd_s[lr * ldd + col.slot] = static_cast<int16_t>(lvl);
        }
        mk_s[lr * n_cols + c] = obs ? 1 : 0;
      }
      if (bad_here) atomicAdd(&rejected[c], bad_here);
    }
    __syncthreads();
    for (int i = tid; i < nrows * n_num; i += 256) {
      const int lr = i / n_num, j = i - lr * n_num;
      xnum[(size_t)(r0 + lr) * ld_num + j] = xn_s[i];
    }
    for (int i = tid; i < nrows * n_disc; i += 256) {
      const int lr = i / n_disc, j = i - lr * n_disc;
      codes[(size_t)(r0 + lr) * ld_disc + j] = cd_s[lr * ldd + j];
    }
    for (int i = tid; i < nrows * n_cols; i += 256) {
      const int lr = i / n_cols, j = i - lr * n_cols;
      mask[(size_t)(r0 + lr) * ld_mask + j] = mk_s[i];
    }
  }
}

// One thread per (output column, row); lane = row, so the columnar stores are coalesced (the row-major reads hit
// each row's cache line once per column group; the table of a chunk sits in L2).
__global__ __launch_bounds__(256) void k_unpack_columns(const suave_unpack_col_t* __restrict__ cols, int n_cols, int64_t n,
                                                        const float* __restrict__ out_num, int ld_num,
                                                        const int16_t* __restrict__ out_codes, int ld_disc, float* f32_block,
                                                        int64_t* i64_block, int16_t* i16_block, int8_t* i8_block,
                                                        int64_t ld_rows) {
  const suave_unpack_col_t col = cols[blockIdx.y];
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
    if (col.block == 0) {
      f32_block[(size_t)col.dst * ld_rows + r] = out_num[(size_t)r * ld_num + col.src];
    } else if (col.block == 1) {  // counts: max(rint(v), 0) as int64 (sampling.py:159-160)
      i64_block[(size_t)col.dst * ld_rows + r] = static_cast<int64_t>(fmaxf(rintf(out_num[(size_t)r * ld_num + col.src]), 0.f));
    } else if (col.block == 2) {
      i16_block[(size_t)col.dst * ld_rows + r] = out_codes[(size_t)r * ld_disc + col.src];
    } else {
      i8_block[(size_t)col.dst * ld_rows + r] = static_cast<int8_t>(out_codes[(size_t)r * ld_disc + col.src]);
    }
  }
}

}  // namespace

int launch_pack_columns(const suave_pack_col_t* cols_dev, int n_cols, int n_num, int n_disc, int64_t n, float* xnum,
                        int ld_num, int16_t* codes, int ld_disc, uint8_t* mask, int ld_mask, unsigned long long* rejected,
                        hipStream_t s) {
  if (n <= 0) return 0;
  const size_t row_bytes = (size_t)n_num * 4 + (size_t)((n_disc + 1) & ~1) * 2 + (size_t)n_cols;
  int rows = 64;
  while (rows > 4 && rows * row_bytes > 48 * 1024) rows >>= 1;
  SUAVE_REQUIRE(rows * row_bytes <= 48 * 1024, "pack_columns: schema too wide for the staging buffer");
  const int grid = (int)std::min<int64_t>((n + rows - 1) / rows, 8192);
  hipLaunchKernelGGL(k_pack_columns, dim3(grid), dim3(256), rows * row_bytes, s, cols_dev, n_cols, n_num, n_disc, n, rows, xnum,
                     ld_num, codes, ld_disc, mask, ld_mask, rejected);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_unpack_columns(const suave_unpack_col_t* cols_dev, int n_cols, int64_t n, const float* out_num, int ld_num,
                          const int16_t* out_codes, int ld_disc, float* f32_block, int64_t* i64_block, int16_t* i16_block,
                          int8_t* i8_block, int64_t ld_rows, hipStream_t s) {
  if (n <= 0 || n_cols <= 0) return 0;
  const int gx = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_unpack_columns, dim3(gx, n_cols), dim3(256), 0, s, cols_dev, n_cols, n, out_num, ld_num, out_codes,
                     ld_disc, f32_block, i64_block, i16_block, i8_block, ld_rows);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
