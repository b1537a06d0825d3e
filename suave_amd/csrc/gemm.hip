// fp32 MFMA GEMM for the encoder / decoder layers (forward, data-gradient and
// weight-gradient forms) with fused epilogues.
//
// Replaces the nn.Linear / ReLU / Dropout chains of suave/modules/encoder.py:46-58,
// :60-72 and suave/modules/decoder.py:286-300, :343-344 and their autograd
// backward.  All three GEMM forms of a layer are one kernel:
//   forward      C[B,N]   = A[B,K]   * W[N,K]^T   (A row-major, B operand k-contiguous)
//   data grad    dA[B,K]  = dC[B,N]  * W[N,K]     (B operand k-major)
//   weight grad  dW[N,K]  = dC[B,N]^T * A[B,K]    (both operands k-major, split over B)
// Arithmetic is v_mfma_f32_32x32x2_f32: exact binary32 products accumulated in
// binary32 (the parity bar is 1e-4 on gradients, which bf16 inputs do not hold).
// Block tile 128x128 (2x2 waves) or 256x64 (4x1 waves), wave tile 64x64 = 2x2
// MFMA tiles, K step 16, LDS tiles stored k-major ([k][i]) so that the MFMA
// operand fetch (lane l: i = l&31, k = l>>5) is a conflict-free ds_read_b32,
// double-buffered with the next tile's global loads in flight during the MFMAs.
#include "common.h"

namespace suave {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int PAD = 4;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// Stateless dropout decision keyed by (seed, step, stream, table row, column).
__device__ __forceinline__ bool drop_keep(uint32_t key, uint32_t row_id, uint32_t col, uint32_t thr) {
  uint32_t h = hash32(row_id * 0x9E3779B1u + key);
  h = hash32(h ^ (col * 0x85EBCA77u));
  return h >= thr;
}

template <int WAVES_M, int WAVES_N, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int LDAS = BM + PAD, LDBS = BN + PAD;
  constexpr int NA = BM * BK / 4 / 256;  // float4 loads per thread for the A tile
  constexpr int NB = BN * BK / 4 / 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDAS + LDBS)];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware tile order: blocks id, id+8, ... run on one XCD and share its L2,
  // so give them the column tiles of the same row tile (they re-read one A tile).
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  int id = blockIdx.x, tile_m, tile_n;
  {
    const int full = (tiles_m / 8) * 8;
    const int n_full = full * tiles_n;
    if (id < n_full) {
      const int xcd = id & 7, w = id >> 3;
      tile_m = (w / tiles_n) * 8 + xcd;
      tile_n = w % tiles_n;
    } else {
      const int r = id - n_full;
      tile_m = full + r / tiles_n;
      tile_n = r % tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int k_begin = 0, k_end = a.K;
  if (a.splits > 1) {
    int chunk = (a.K + a.splits - 1) / a.splits;
    chunk = (chunk + BK - 1) / BK * BK;
    k_begin = blockIdx.z * chunk;
    k_end = min(a.K, k_begin + chunk);
  }
  const int nk = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;

  // Loads only issue the (predicated) global reads; the ragged-edge masking is done by
  // fix_tiles() right before the LDS store so the reads stay in flight during the MFMAs.
  auto load_tiles = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB]) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + p * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!A_KM) {
        const int row = idx >> 2, kq = idx & 3;
        const int gi = m0 + row, gk = k0 + kq * 4;
        if (gi < a.M && gk < k_end) v = *reinterpret_cast<const float4*>(a.A + (size_t)gi * a.lda + gk);
      } else {
        const int krow = idx / (BM / 4), iq = idx % (BM / 4);
        const int gk = k0 + krow, gi = m0 + iq * 4;
        if (gk < k_end && gi < a.M) v = *reinterpret_cast<const float4*>(a.A + (size_t)gk * a.lda + gi);
      }
      ra[p] = v;
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = tid + p * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!B_KM) {
        const int row = idx >> 2, kq = idx & 3;
        const int gj = n0 + row, gk = k0 + kq * 4;
        if (gj < a.N && gk < k_end) v = *reinterpret_cast<const float4*>(a.B + (size_t)gj * a.ldb + gk);
      } else {
        const int krow = idx / (BN / 4), jq = idx % (BN / 4);
        const int gk = k0 + krow, gj = n0 + jq * 4;
        if (gk < k_end && gj < a.N) v = *reinterpret_cast<const float4*>(a.B + (size_t)gk * a.ldb + gj);
      }
      rb[p] = v;
    }
  };
  auto store_tiles = [&](int k0, int buf, float4 (&ra)[NA], float4 (&rb)[NB]) {
    float* Ab = As + buf * BK * LDAS;
    float* Bb = Bs + buf * BK * LDBS;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + p * 256;
      const int lim = !A_KM ? k_end - (k0 + (idx & 3) * 4) : a.M - (m0 + (idx % (BM / 4)) * 4);
      float4 v = ra[p];
      if (lim < 4) {
        if (lim < 2) v.y = 0.f;
        if (lim < 3) v.z = 0.f;
        v.w = 0.f;
      }
      if (!A_KM) {
        const int row = idx >> 2, kq = idx & 3;
        Ab[(kq * 4 + 0) * LDAS + row] = v.x;
        Ab[(kq * 4 + 1) * LDAS + row] = v.y;
        Ab[(kq * 4 + 2) * LDAS + row] = v.z;
        Ab[(kq * 4 + 3) * LDAS + row] = v.w;
      } else {
        const int krow = idx / (BM / 4), iq = idx % (BM / 4);
        *reinterpret_cast<float4*>(Ab + krow * LDAS + iq * 4) = v;
      }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = tid + p * 256;
      const int lim = !B_KM ? k_end - (k0 + (idx & 3) * 4) : a.N - (n0 + (idx % (BN / 4)) * 4);
      float4 v = rb[p];
      if (lim < 4) {
        if (lim < 2) v.y = 0.f;
        if (lim < 3) v.z = 0.f;
        v.w = 0.f;
      }
      if (!B_KM) {
        const int row = idx >> 2, kq = idx & 3;
        Bb[(kq * 4 + 0) * LDBS + row] = v.x;
        Bb[(kq * 4 + 1) * LDBS + row] = v.y;
        Bb[(kq * 4 + 2) * LDBS + row] = v.z;
        Bb[(kq * 4 + 3) * LDBS + row] = v.w;
      } else {
        const int krow = idx / (BN / 4), jq = idx % (BN / 4);
        *reinterpret_cast<float4*>(Bb + krow * LDBS + jq * 4) = v;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // 32-row / 32-column sub-tiles entirely outside the matrix are skipped (wave-uniform).
  const bool act_m0 = m0 + wm * 64 < a.M, act_m1 = m0 + wm * 64 + 32 < a.M;
  const bool act_n0 = n0 + wn * 64 < a.N, act_n1 = n0 + wn * 64 + 32 < a.N;
  const int kh = lane >> 5, l31 = lane & 31;
  // bias gradient (weight-gradient form): column sums of the A operand, taken from the LDS tile
  float bias_acc = 0.f;
  const bool do_bias = a.epi == kEpiSlab && a.ones_col >= 0 && n0 == 0 && tid < BM;

  // One K step on LDS buffer `buf`; operand fragments of step kk+1 are fetched before the MFMAs of kk.
  auto compute = [&](int buf) {
    const float* Ab = As + buf * BK * LDAS + wm * 64 + l31;
    const float* Bb = Bs + buf * BK * LDBS + wn * 64 + l31;
    if (act_m0 && act_n0) {
      float a0 = Ab[kh * LDAS], a1 = Ab[kh * LDAS + 32];
      float b0 = Bb[kh * LDBS], b1 = Bb[kh * LDBS + 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        if (kk + 1 < BK / 2) {
          const int kr = (kk + 1) * 2 + kh;
          na0 = Ab[kr * LDAS]; na1 = Ab[kr * LDAS + 32];
          nb0 = Bb[kr * LDBS]; nb1 = Bb[kr * LDBS + 32];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        if (act_n1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        if (act_m1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        if (act_m1 && act_n1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      }
    }
    if (do_bias) {
      const float* Ac = As + buf * BK * LDAS + tid;
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) bias_acc += Ac[kk * LDAS];
    }
  };

  // Two register sets: while tile t is multiplied out of LDS, the global loads of tiles t+1 and t+2
  // are in flight (the chip needs ~2 us of loads in flight per CU; one K step is < 1 us of MFMA).
  float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
  if (nk > 0) {
    load_tiles(k_begin, ra0, rb0);
    store_tiles(k_begin, 0, ra0, rb0);
    if (nk > 1) load_tiles(k_begin + BK, ra0, rb0);
    if (nk > 2) load_tiles(k_begin + 2 * BK, ra1, rb1);
  }
  __syncthreads();
  for (int t = 0; t < nk; t += 2) {
    compute(0);
    if (t + 1 < nk) store_tiles(k_begin + (t + 1) * BK, 1, ra0, rb0);
    if (t + 3 < nk) load_tiles(k_begin + (t + 3) * BK, ra0, rb0);
    __syncthreads();
    if (t + 1 < nk) {
      compute(1);
      if (t + 2 < nk) store_tiles(k_begin + (t + 2) * BK, 0, ra1, rb1);
      if (t + 4 < nk) load_tiles(k_begin + (t + 4) * BK, ra1, rb1);
      __syncthreads();
    }
  }

  // ---- epilogue -------------------------------------------------------------
  float* Cout = a.C + (a.epi == kEpiSlab ? (size_t)blockIdx.z * a.slab_stride : 0);
  if (do_bias && m0 + tid < a.M) Cout[(size_t)(m0 + tid) * a.ldc + a.ones_col] = bias_acc;
  const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t thr = a.drop_p > 0.f ? static_cast<uint32_t>(fminf(a.drop_p * 4294967296.0f, 4294967040.0f)) : 0u;
  const uint32_t key = hash32(static_cast<uint32_t>(a.seed) ^ hash32(static_cast<uint32_t>(a.seed >> 32) + a.stream_id)) ^
                       hash32(static_cast<uint32_t>(a.step) * 0x9E3779B1u + 0x632BE5ABu);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + l31;
      const int row_base = m0 + wm * 64 + mi * 32 + 4 * kh;
      if (row_base - 4 * kh >= a.M || n0 + wn * 64 + ni * 32 >= a.N) continue;
      float bias = 0.f;
      if ((a.epi == kEpiBias || a.epi == kEpiReluDrop) && a.bias != nullptr && col < a.N) bias = a.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        if (row >= a.M || col >= a.N) continue;
        float v = acc[mi][ni][r];
        const size_t off = (size_t)row * a.ldc + col;
        if (a.epi == kEpiBias) {
          v += bias;
        } else if (a.epi == kEpiReluDrop) {
          v = fmaxf(v + bias, 0.f);
          if (a.drop_p > 0.f) {
            bool keep;
            if (a.keep != nullptr) {
              keep = a.keep[(size_t)row * a.N + col] != 0;
            } else {
              const uint32_t rid = a.row_ids ? static_cast<uint32_t>(a.row_ids[row])
                                             : static_cast<uint32_t>(a.row_begin + row);
              keep = drop_keep(key, rid, static_cast<uint32_t>(col), thr);
            }
            v = keep ? v * inv_keep : 0.f;
          }
        } else if (a.epi == kEpiMaskBwd) {
          v = a.H[off] > 0.f ? v * inv_keep : 0.f;
        }
        Cout[off] = v;
      }
    }
  }
}

}  // namespace

int launch_gemm(const GemmArgs& a, int n_cu, hipStream_t stream) {
  (void)n_cu;
  if (a.M <= 0 || a.N <= 0) return 0;
  SUAVE_REQUIRE(a.lda % 4 == 0 && a.ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4 floats");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0,
                "gemm: operands must be 16-byte aligned");
  SUAVE_REQUIRE(a.a_kmajor ? a.lda >= a.M : a.lda >= a.K, "gemm: lda too small");
  SUAVE_REQUIRE(a.b_kmajor ? a.ldb >= a.N : a.ldb >= a.K, "gemm: ldb too small");
  SUAVE_REQUIRE(a.ldc >= a.N + (a.ones_col >= 0 ? 1 : 0), "gemm: ldc too small");
  SUAVE_REQUIRE(a.ones_col < 0 || a.ones_col == a.N, "gemm: the bias column sits right after the N real columns");
  SUAVE_REQUIRE(a.epi != kEpiSlab || a.splits >= 1, "gemm: bad split count");
  const int splits = a.epi == kEpiSlab ? a.splits : 1;
  GemmArgs args = a;
  args.splits = splits;
  ProfScope prof_scope(kProfGemm, stream);
  const bool narrow = a.N <= 64;
  const int tiles = narrow ? ((a.M + 255) / 256) * ((a.N + 63) / 64) : ((a.M + 127) / 128) * ((a.N + 127) / 128);
  const dim3 grid(tiles, 1, splits), block(256);
#define SUAVE_GEMM_LAUNCH(AKM, BKM)                                                              \
  do {                                                                                           \
    if (narrow) hipLaunchKernelGGL((gemm_kernel<4, 1, AKM, BKM>), grid, block, 0, stream, args); \
    else hipLaunchKernelGGL((gemm_kernel<2, 2, AKM, BKM>), grid, block, 0, stream, args);        \
  } while (0)
  if (!a.a_kmajor && !a.b_kmajor) SUAVE_GEMM_LAUNCH(false, false);
  else if (!a.a_kmajor && a.b_kmajor) SUAVE_GEMM_LAUNCH(false, true);
  else if (a.a_kmajor && a.b_kmajor) SUAVE_GEMM_LAUNCH(true, true);
  else SUAVE_REQUIRE(false, "gemm: unsupported operand layout (A k-major with B k-contiguous)");
#undef SUAVE_GEMM_LAUNCH
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
