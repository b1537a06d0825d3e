// fp32 MFMA GEMM for the encoder / decoder layers (forward, data-gradient and
// weight-gradient forms) with fused epilogues.
//
// Replaces the nn.Linear / ReLU / Dropout chains of suave/modules/encoder.py:46-58,
// :60-72 and suave/modules/decoder.py:286-300, :343-344 and their autograd
// backward.  All three GEMM forms of a layer are one kernel:
//   forward      C[B,N]   = A[B,K]   * W[N,K]^T   (A row-major, B operand k-contiguous)
//   data grad    dA[B,K]  = dC[B,N]  * W[N,K]     (B operand k-major)
//   weight grad  dW[N,K]  = dC[B,N]^T * A[B,K]    (both operands k-major, split over B)
// Arithmetic (suave_gemm_arithmetic_set), fp32 in / fp32 accumulate / fp32 out either way:
//   mode 4 (default): two fp16 planes, x = hi + lo' * 2^-11, three plane products on v_mfma_f32_32x32x16_f16, the two
//           cross products in a second accumulator set added with weight 2^-11 (see split_planes() below): half the
//           matrix-pipe time of mode 3 at the same measured accuracy; operands within fp16's range;
//   mode 3: every fp32 operand element is cut into three bf16 planes when its tile is stored to LDS and
//           the six plane products with i + j <= 2 run on v_mfma_f32_32x32x16_bf16 -- the error of an fp32 FMA
//           chain (the parity bar is 1e-4 on gradients, which plain bf16 inputs do not hold) at 6/16 of the fp32
//           MFMA's matrix-pipe time; row-major forms read [row][16 k] plane images with one ds_read_b128 per
//           fragment, k-major forms (weight gradients) keep [16 k][128 features] images and transpose on the way
//           out with ds_read_b64_tr_b16;
//   mode 0: v_mfma_f32_32x32x2_f32 everywhere (exact binary32 products), LDS tiles k-major ([k][i]) so that the
//           operand fetch (lane l: i = l&31, k = l>>5) is a conflict-free ds_read_b32.
// Block tile 128x128 (2x2 waves) or 256x64 (4x1 waves, N <= 64), wave tile 64x64 = 2x2 MFMA tiles, K step 16,
// double-buffered LDS with the next two tiles' global loads in flight during the MFMAs.
// ELBO steps of at most 8192 rows (suave_small_batch_rows_set) take the split-K kernels instead (gemm_small_*,
// mode 4 only): a block per 32 x 64 output tile, the K range divided among its four waves, fragments loaded
// straight from global memory -- a short latency chain where the tile kernel leaves the chip empty.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace suave {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

int split_mode();  // defined with the launchers below

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

// Stateless dropout decision keyed by (seed, step, stream, table row, column): this is the row part; the
// epilogue mixes in the column pair and spends 16 hash bits per element.
__device__ __forceinline__ uint32_t drop_row_hash(uint32_t key, uint32_t row_id) { return hash32(row_id * 0x9E3779B1u + key); }

// Split mode (SPLIT = NS > 0, row-major operands only): every fp32 operand element is cut into NS bf16 planes
// (hi, mid, lo: x = hi + mid + lo up to 2^-24 |x| for NS = 3) when its tile is stored to LDS, and the tile
// product is the sum of the plane products on v_mfma_f32_32x32x16_bf16, accumulated in fp32:
//   NS = 3: the six pairs with i + j <= 2  -> the error of an fp32 FMA chain (measured 4e-7 of max|C| at
//           K = 256, 1.6e-6 at K = 4096; the exact-fp32 MFMA kernel: 3e-6 at K = 4096), at 6/16 of its
//           matrix-pipe time;
//   NS = 2: three pairs                    -> ~5e-6 on well-scaled operands, at 3/16 (not dispatched: see
//           split_mode()).
// The bf16 MFMA has its own pipe: the split arithmetic (VALU) runs beside it instead of taking its cycles, which
// the f32 MFMA cannot offer (it executes on the SIMD's FMA lanes).  LDS image per plane: [row][16 k] bf16, 32 B
// per row: a lane's MFMA fragment (row l & 31, k = 8 (l >> 5) ... + 7) is one ds_read_b128.
//
// k-major operands (weight gradients: element (i, k) at X[k * ld + i], 128 x 128 tile only) stay k-major in LDS:
// plane image [16 k][128 features] bf16 with 320-byte k rows (store_tiles), transposed on the way out by
// ds_read_b64_tr_b16 (compute()).  split_row_bytes sizes that image: (128 + 128) x 40 = 2 x 16 x 320 bytes a plane.
// (Tried and dropped: an unpadded, XOR-swizzled image -- 48 KB, three blocks per CU -- needs the 168-register
// budget, where these kernels spill 108 bytes per lane.)
//
// SPLIT = 4 (arithmetic mode 4): TWO fp16 planes, x = hi + lo' * 2^-11 with hi = fp16(x) and lo' = fp16((x - hi) * 2^11)
// (22 significand bits; the residual is scaled so that it does not fall into the fp16 subnormals), and THREE products
// on v_mfma_f32_32x32x16_f16: hi*hi into the main accumulators, hi*lo' + lo'*hi into a second set that is added with
// weight 2^-11 after the K loop -- half the matrix-pipe time of the three-plane bf16 split, two thirds of its LDS
// traffic, at the same measured accuracy (scripts/split_emulate.py: worst gradient error 4e-7 .. 1e-6 of the tensor
// scale on the S64 step, where the two-plane bf16 split is 5e-2 off).  fp16's range is the price: |x| <= 65504 (a larger
// operand gives inf / NaN, loudly), and gradient operands are pre-scaled by a power of two (GemmArgs::a_scale).
constexpr int split_planes(int split) { return split == 4 ? 2 : split; }
constexpr float kF16LoScale = 2048.0f, kF16LoInv = 1.0f / 2048.0f;
constexpr int split_row_bytes(bool any_km) { return any_km ? 2 * BK + 8 : 2 * BK; }
template <int WAVES_M, int WAVES_N, int SPLIT = 0, bool KM = false>
constexpr int gemm_smem_floats() {
  constexpr int f32 = 2 * BK * ((WAVES_M * 64 + PAD) + (WAVES_N * 64 + PAD));
  constexpr int spl = 2 * split_planes(SPLIT) * (WAVES_M * 64 + WAVES_N * 64) * split_row_bytes(KM) / 4;
  return f32 > spl ? f32 : spl;
}

// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements comes back column-major:
// lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q.
// (Needs EXEC all ones and an 8-byte aligned address.)
__device__ __forceinline__ bf16x4 lds_read_tr16(const unsigned char* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  // the low 32 bits of a generic pointer into LDS are the LDS offset
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(uint32_t)(uintptr_t)p));
}

template <int NS>
__device__ __forceinline__ void split4(const float4 v, float scale, bf16x4 (&p)[split_planes(NS) > 0 ? split_planes(NS) : 1]) {
  float r[4] = {v.x, v.y, v.z, v.w};
  if constexpr (NS == 4) {  // fp16 hi + scaled fp16 residual (the planes travel as 16-bit patterns in the bf16 types)
    // (v_fma_mixlo/mixhi_f16 + v_fma_mix_f32 in inline asm take 13 % of the K loop's VALU instructions away and
    // leave the step time where it was: the loop is not VALU-bound.  Plain conversions kept.)
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = r[e] * scale;
      const _Float16 h = (_Float16)x;  // round to nearest even
      hi[e] = h;
      lo[e] = (_Float16)((x - (float)h) * kF16LoScale);  // x - h is exact in fp32
    }
    p[0] = __builtin_bit_cast(bf16x4, hi);
    p[1] = __builtin_bit_cast(bf16x4, lo);
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 h = (__bf16)r[e];  // round to nearest even (v_cvt_pk_bf16_f32)
      p[s][e] = h;
      r[e] -= (float)h;               // exact in fp32
    }
  }
}

// kEpiReluDrop on one quad of a row: (+ addend) + bias, ReLU, dropout (hashed keep decisions or the injected mask)
__device__ __forceinline__ void relu_drop_quad(const GemmArgs& a, float (&v)[4], const float4& addend, const float (&bias4)[4],
                                               bool hashed_drop, uint32_t keep4, int col, int ncol, float inv_keep, int row) {
  if (a.H != nullptr) {  // first encoder layer of wide schemas: + the embedding rows (embed.hip)
    v[0] += addend.x; v[1] += addend.y; v[2] += addend.z; v[3] += addend.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias4[e], 0.f);
  if (a.drop_p > 0.f) {
    if (hashed_drop) {  // keep4: the quad's four keep decisions, hashed before the K loop (gemm_body)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ((keep4 >> e) & 1u) ? v[e] * inv_keep : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < ncol) v[e] = a.keep[(size_t)row * a.N + col + e] != 0 ? v[e] * inv_keep : 0.f;
    }
  }
}

// One block's work: tile `bid` of split `bz`.
template <int WAVES_M, int WAVES_N, bool A_KM, bool B_KM, int GRP, int EPI, int SPLIT = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& a, const int bid, const int bz, float* smem, int* perm_s) {
  static_assert(SPLIT == 0 || (A_KM == B_KM && (GRP != 2 || A_KM) && !(A_KM && (WAVES_M != 2 || WAVES_N != 2))),
                "split mode: both operands row-major, or both k-major on the 128x128 tile");
  constexpr int kSplitRowBytes = split_row_bytes(A_KM);
  constexpr bool KM_SPLIT = SPLIT > 0 && A_KM;
  constexpr int NP = split_planes(SPLIT);  // LDS plane images per operand tile
  const float a_scale = (SPLIT == 4 && a.a_scale > 0.f) ? a.a_scale : 1.0f;
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  constexpr int kKmRowBytes = 2 * BM + 64;  // k-major split tiles: one k row of BM bf16 features + 64 bytes
  static_assert(!KM_SPLIT || 2 * BK * kKmRowBytes == (BM + BN) * kSplitRowBytes, "plane size of the k-major image");
  constexpr int LDAS = BM + PAD, LDBS = BN + PAD;
  constexpr int NA = BM * BK / 4 / 256;  // float4 loads per thread for the A tile
  constexpr int NB = BN * BK / 4 / 256;
  static_assert(kGroupRows % BM == 0, "a row tile never straddles two component chunks");
  float* As = smem;
  float* Bs = smem + 2 * BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware tile order: blocks id, id+8, ... run on one XCD and share its L2,
  // so give them the column tiles of the same row tile (they re-read one A tile).
  // Many column tiles (tiles_n >= 16: the 1 024-column schema's y_projection, 133 of them): an XCD walks the column
  // tiles of TWO of its row tiles together, row tile fastest, so each B (weight) tile comes out of HBM once per two
  // row tiles (W1024 y_projection forward: 1.39 -> 1.32 ms; four or eight row tiles together were slower, 1.40 /
  // 1.58 ms: the kernel is at its compute rate, 218 TFLOP/s, and the wider write footprint costs more than the
  // weight re-reads save).
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  int id = bid, tile_m, tile_n;
  {
    const int G = tiles_n >= 16 ? 2 : 1;                 // row tiles of one XCD walked together
    const int full = (tiles_m / (8 * G)) * (8 * G);
    const int n_full = full * tiles_n;
    if (id < n_full) {
      const int xcd = id & 7, w = id >> 3;
      const int grp = w / (G * tiles_n), wi = w - grp * (G * tiles_n);
      tile_m = (grp * G + wi % G) * 8 + xcd;
      tile_n = wi / G;
    } else {
      const int r = id - n_full;
      tile_m = full + r / tiles_n;
      tile_n = r % tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int k_begin = 0, k_end = a.K;
  if (a.splits > 1 || GRP == 2) {
    int chunk = (a.K + a.splits - 1) / a.splits;
    chunk = (chunk + BK - 1) / BK * BK;
    k_begin = bz * chunk;
    k_end = min(a.K, k_begin + chunk);
  }
  const int nk = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;

  // Grouped forms: the component of this tile / chunk; unused chunks have no work (block-uniform exit).
  int comp = 0;
  if (GRP == 1) comp = a.chunk_comp[m0 / kGroupRows];
  if (GRP == 2) comp = a.chunk_comp[bz];
  if (GRP != 0 && comp < 0) return;
  if (GRP == 2) {
    perm_s[tid] = a.perm[k_begin + tid];  // kGroupRows == blockDim.x
    __syncthreads();
  }
  // Row gathers that do not depend on the K step.
  int arow[NA], brow[NB];
#pragma unroll
  for (int p = 0; p < NA; ++p) arow[p] = (GRP == 1 && !A_KM) ? a.perm[m0 + ((tid + p * 256) >> 2)] : 0;
#pragma unroll
  for (int p = 0; p < NB; ++p)
    brow[p] = (GRP == 1 && !B_KM) ? head_row(a.q0 + n0 + ((tid + p * 256) >> 2), comp, a.hK, a.hD, a.hldK, a.hldD) : 0;

  // Loads are unconditional (an out-of-range element reads the operand's first float4 instead) so that the
  // main loop has no control flow around them and the waits on them are exact counts; `okm` remembers which
  // were real and the LDS store zeroes the others and the ragged edges.
  auto load_tiles = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB], uint32_t& okm) {
    okm = 0;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + p * 256;
      bool ok;
      size_t off;
      if (!A_KM) {
        const int row = idx >> 2, kq = idx & 3;
        const int gi = m0 + row, gk = k0 + kq * 4;
        if (GRP == 1) { ok = arow[p] >= 0 && gk < k_end; off = (size_t)arow[p] * a.lda + gk; }
        else { ok = gi < a.M && gk < k_end; off = (size_t)gi * a.lda + gk; }
      } else {
        // split mode: a thread takes k rows 2g, 2g + 1 of one feature group (its 4 x 2 register block)
        const int krow = idx / (BM / 4), iq = idx % (BM / 4);
        const int gk = k0 + krow, gi = m0 + iq * 4;
        if (GRP == 2) {
          const int sr = perm_s[min(gk - k_begin, kGroupRows - 1)];
          ok = sr >= 0 && gk < k_end && gi < a.M;
          off = (size_t)sr * a.lda + gi;
        } else { ok = gk < k_end && gi < a.M; off = (size_t)gk * a.lda + gi; }
      }
      ra[p] = *reinterpret_cast<const float4*>(ok ? a.A + off : a.A);
      okm |= ok ? (1u << p) : 0u;
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = tid + p * 256;
      bool ok;
      size_t off;
      if (!B_KM) {
        const int row = idx >> 2, kq = idx & 3;
        const int gj = n0 + row, gk = k0 + kq * 4;
        if (GRP == 1) { ok = brow[p] >= 0 && gj < a.N && gk < k_end; off = (size_t)brow[p] * a.ldb + gk; }
        else { ok = gj < a.N && gk < k_end; off = (size_t)gj * a.ldb + gk; }
      } else {
        const int krow = idx / (BN / 4), jq = idx % (BN / 4);
        const int gk = k0 + krow, gj = n0 + jq * 4;
        if (GRP == 1) {
          const int br = head_row(a.q0 + gk, comp, a.hK, a.hD, a.hldK, a.hldD);
          ok = br >= 0 && gk < k_end && gj < a.N;
          off = (size_t)br * a.ldb + gj;
        } else if (GRP == 2) {
          const int sr = perm_s[min(gk - k_begin, kGroupRows - 1)];
          ok = sr >= 0 && gk < k_end && gj < a.N;
          off = (size_t)sr * a.ldb + gj;
        } else { ok = gk < k_end && gj < a.N; off = (size_t)gk * a.ldb + gj; }
      }
      rb[p] = *reinterpret_cast<const float4*>(ok ? a.B + off : a.B);
      okm |= ok ? (1u << (8 + p)) : 0u;
    }
  };
  // split mode: plane s of buffer `buf`, A rows then B rows
  auto plane = [&](int buf, int s) {
    return reinterpret_cast<unsigned char*>(smem) + (size_t)((buf * (NP > 0 ? NP : 1) + s) * (BM + BN)) * kSplitRowBytes;
  };
  // `masked` (std::true_type / std::false_type, split paths only): false skips the zero-fill selects.  Zeros are only
  // owed past k_end (both operands: garbage there would reach every output) and, in the grouped form, in the rows of
  // absent heads; a tile row or feature outside the matrix only feeds outputs the epilogue never writes.  The main
  // loop stores its full K steps unmasked (40 of 146 VALU per step less).
  auto store_tiles = [&](int k0, int buf, float4 (&ra)[NA], float4 (&rb)[NB], uint32_t okm, auto masked) {
    constexpr bool kMasked = decltype(masked)::value;
    if constexpr (SPLIT > 0) {
      // Keep every loaded float4 one whole register tuple until this point: without it the compiler copies single
      // elements out right behind the loads (before the barrier and the MFMAs) and waits for the loads there.
      auto pin = [](float4& v) {
        typedef float f32x4n __attribute__((ext_vector_type(4)));
        f32x4n t = {v.x, v.y, v.z, v.w};
        asm volatile("" : "+v"(t));
        v = make_float4(t[0], t[1], t[2], t[3]);
      };
#pragma unroll
      for (int p = 0; p < NA; ++p) pin(ra[p]);
#pragma unroll
      for (int p = 0; p < NB; ++p) pin(rb[p]);
    }
    if constexpr (KM_SPLIT) {
      // k-major tiles stay k-major in LDS: plane image [16 k][128 features] bf16, rows 320 bytes apart (256 + 64:
      // the four k rows one transposed read touches land on disjoint 16-bank windows); a thread stores the 4
      // features of its float4 with one ds_write_b64 per plane, and compute() transposes on the way out with
      // ds_read_b64_tr_b16.
      static_assert(!KM_SPLIT || (NA == 2 && NB == 2 && BM == 128 && BN == 128), "128 x 128 tile");
      const int kr = tid >> 5, fq = tid & 31;  // load_tiles: slot p holds k row kr + 8 p, features 4 fq .. 4 fq + 3
#pragma unroll
      for (int op = 0; op < 2; ++op) {  // 0: A tile, 1: B tile
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float4 v = op == 0 ? ra[p] : rb[p];
          if constexpr (kMasked) {
            const bool okp = (okm >> (op == 0 ? p : 8 + p)) & 1u;                              // k row inside [k0, k_end)
            const int lim = okp ? (op == 0 ? a.M - (m0 + fq * 4) : a.N - (n0 + fq * 4)) : 0;  // features inside the matrix
            v.x = lim > 0 ? v.x : 0.f;
            v.y = lim > 1 ? v.y : 0.f;
            v.z = lim > 2 ? v.z : 0.f;
            v.w = lim > 3 ? v.w : 0.f;
          }
          bf16x4 pl[NP > 0 ? NP : 1];
          split4<SPLIT>(v, op == 0 ? a_scale : 1.0f, pl);
#pragma unroll
          for (int s = 0; s < NP; ++s)
            *reinterpret_cast<bf16x4*>(plane(buf, s) + (size_t)(op * BK + kr + 8 * p) * kKmRowBytes + fq * 8) = pl[s];
        }
      }
      return;
    } else if constexpr (SPLIT > 0) {
      {
#pragma unroll
        for (int p = 0; p < NA + NB; ++p) {
          const bool isA = p < NA;
          const int q = isA ? p : p - NA;
          const int idx = tid + q * 256;
          const int row = idx >> 2, kq = idx & 3;
          float4 v = isA ? ra[q] : rb[q];
          if constexpr (kMasked) {
            int lim = k_end - (k0 + kq * 4);
            if (!((okm >> (isA ? q : 8 + q)) & 1u)) lim = 0;
            v.x = lim > 0 ? v.x : 0.f;
            v.y = lim > 1 ? v.y : 0.f;
            v.z = lim > 2 ? v.z : 0.f;
            v.w = lim > 3 ? v.w : 0.f;
          }
          bf16x4 pl[NP > 0 ? NP : 1];
          split4<SPLIT>(v, isA ? a_scale : 1.0f, pl);
          // the 16-byte halves of a row swap places in every second group of 8 rows: the 16 lanes one ds_read_b128
          // cycle serves (rows 0-3, 12-15, 20-27 / 4-11, 16-19, 28-31) then cover all 64 banks
          const int kbyte = (((kq >> 1) ^ ((row >> 3) & 1)) << 4) | ((kq & 1) << 3);
#pragma unroll
          for (int s = 0; s < NP; ++s)
            *reinterpret_cast<bf16x4*>(plane(buf, s) + (size_t)((isA ? 0 : BM) + row) * kSplitRowBytes + kbyte) = pl[s];
        }
      }
      return;
    }
    float* Ab = As + buf * BK * LDAS;
    float* Bb = Bs + buf * BK * LDBS;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + p * 256;
      int lim = !A_KM ? k_end - (k0 + (idx & 3) * 4) : a.M - (m0 + (idx % (BM / 4)) * 4);
      if (!((okm >> p) & 1u)) lim = 0;
      float4 v = ra[p];
      v.x = lim > 0 ? v.x : 0.f;
      v.y = lim > 1 ? v.y : 0.f;
      v.z = lim > 2 ? v.z : 0.f;
      v.w = lim > 3 ? v.w : 0.f;
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
      int lim = !B_KM ? k_end - (k0 + (idx & 3) * 4) : a.N - (n0 + (idx % (BN / 4)) * 4);
      if (!((okm >> (8 + p)) & 1u)) lim = 0;
      float4 v = rb[p];
      v.x = lim > 0 ? v.x : 0.f;
      v.y = lim > 1 ? v.y : 0.f;
      v.z = lim > 2 ? v.z : 0.f;
      v.w = lim > 3 ? v.w : 0.f;
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
  f32x16 accx[SPLIT == 4 ? 2 : 1][SPLIT == 4 ? 2 : 1];  // mode 4: the two cross products (weight 2^-11)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (SPLIT == 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.f;
  }

  // A wave whose 64x64 tile lies entirely outside the matrix skips the MFMAs (wave-uniform); partly
  // outside sub-tiles multiply the zero rows the LDS store wrote (one straight-line MFMA stream).
  const bool act = m0 + wm * 64 < a.M && n0 + wn * 64 < a.N;
  const int kh = lane >> 5, l31 = lane & 31;
  // bias gradient (weight-gradient form): column sums of the A operand, taken from the LDS tile
  float bias_acc = 0.f;
  float4 kbias4 = make_float4(0.f, 0.f, 0.f, 0.f);                                   // k-major split path
  const bool bias_blk = EPI == kEpiSlab && a.ones_col >= 0 && n0 == 0;  // block-uniform
  const bool do_bias = EPI == kEpiSlab && a.ones_col >= 0 && n0 == 0 && tid < BM;

  // One K step on LDS buffer `buf`; operand fragments of step kk+1 are fetched before the MFMAs of kk.
  auto compute = [&](int buf) {
    if constexpr (SPLIT > 0) {
      {  // (no `act` test here: a wave outside the matrix multiplies tiles nobody stores; its block waits at the
         // barrier for the other waves anyway, and the K step stays one basic block the scheduler can interleave)
        bf16x8 fa[NP > 0 ? NP : 1][2], fb[NP > 0 ? NP : 1][2];
        // fragments are fetched plane by plane and the plane products run in the order hi*hi, hi*mid, mid*hi,
        // mid*mid, hi*lo, lo*hi: the first MFMAs wait for four ds_read_b128 only, the rest of the fetch lands
        // behind them
#pragma unroll
        for (int s = 0; s < NP; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if constexpr (KM_SPLIT) {
              // lane 16 g + 4 q + p of a 32-lane half addresses k row 8 kh + q (+4 for the second read), features
              // 16 g + 4 p ..; it receives feature l31 with k = 8 kh + 0..3 (+4): the MFMA fragment order
              const int q4 = (lane >> 2) & 3, fcol = (lane & 16) + 4 * (lane & 3);
              const unsigned char* pa = plane(buf, s) + (size_t)(8 * kh + q4) * kKmRowBytes + (wm * 64 + i * 32 + fcol) * 2;
              const unsigned char* pb = plane(buf, s) + (size_t)(BK + 8 * kh + q4) * kKmRowBytes + (wn * 64 + i * 32 + fcol) * 2;
              const bf16x4 a0 = lds_read_tr16(pa), a1 = lds_read_tr16(pa + 4 * kKmRowBytes);
              const bf16x4 b0 = lds_read_tr16(pb), b1 = lds_read_tr16(pb + 4 * kKmRowBytes);
              fa[s][i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
              fb[s][i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
              // row-major tiles: halves swapped per 8-row group (store_tiles); l31 >> 3 is the row's group parity
              const int hoff = (kh ^ ((l31 >> 3) & 1)) * 16;
              const unsigned char* pa = plane(buf, s) + (size_t)(wm * 64 + i * 32 + l31) * kSplitRowBytes + hoff;
              const unsigned char* pb = plane(buf, s) + (size_t)(BM + wn * 64 + i * 32 + l31) * kSplitRowBytes + hoff;
              fa[s][i] = *reinterpret_cast<const bf16x8*>(pa);
              fb[s][i] = *reinterpret_cast<const bf16x8*>(pb);
            }
          }
        if constexpr (SPLIT == 4) {
          auto h8 = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(fa[0][i]), h8(fb[0][j]), acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(fa[0][i]), h8(fb[1][j]), accx[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(fa[1][i]), h8(fb[0][j]), accx[i][j], 0, 0, 0);
        }
        constexpr int kPairs = SPLIT == 4 ? 0 : SPLIT == 3 ? 6 : 3;
        constexpr int pa_[6] = {0, 0, 1, 1, 0, 2}, pb_[6] = {0, 1, 0, 1, 2, 0};
        constexpr int qa_[3] = {0, 0, 1}, qb_[3] = {0, 1, 0};
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
          const int sa = SPLIT == 3 ? pa_[q] : qa_[q], sb = SPLIT == 3 ? pb_[q] : qb_[q];
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][0], fb[sb][0], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][0], fb[sb][1], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][1], fb[sb][0], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][1], fb[sb][1], acc[1][1], 0, 0, 0);
        }
      }
      if constexpr (KM_SPLIT) {
        // bias gradient = sums over k of the A tile (dC^T): thread (k group kg, feature quad fq) adds its 4 k rows of
        // the three planes (they add back to the fp32 values); the four k groups are combined after the K loop
        if (bias_blk && tid < 128) {
          const int fq = tid & 31, kg = tid >> 5;
#pragma unroll
          for (int s = 0; s < NP; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bf16x4 v = *reinterpret_cast<const bf16x4*>(plane(buf, s) + (size_t)(kg * 4 + r) * kKmRowBytes + fq * 8);
              if constexpr (SPLIT == 4) {
                const f16x4 h = __builtin_bit_cast(f16x4, v);
                const float wgt = s == 0 ? 1.0f : kF16LoInv;
                kbias4.x += (float)h[0] * wgt; kbias4.y += (float)h[1] * wgt; kbias4.z += (float)h[2] * wgt; kbias4.w += (float)h[3] * wgt;
              } else {
                kbias4.x += (float)v[0]; kbias4.y += (float)v[1]; kbias4.z += (float)v[2]; kbias4.w += (float)v[3];
              }
            }
        }
      }
      return;
    }
    if (act) {
      const float* Ab = As + buf * BK * LDAS + wm * 64 + l31;
      const float* Bb = Bs + buf * BK * LDBS + wn * 64 + l31;
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
        __builtin_amdgcn_sched_barrier(0);  // keep the fetch of step kk+1 ahead of the MFMAs of step kk
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
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
  // Tiles past the end load nothing real and store zeros into a buffer nobody multiplies.
  // Dropout keep decisions of this lane's 16 output quads (16 bits of a per-(row, column pair) hash against
  // round(p * 65536)), packed as 64 bits.  They depend on nothing the K loop computes, so the row ids are fetched first
  // and the hashes run while the first operand tiles are in flight -- issue slots the wave would spend waiting --
  // instead of in the epilogue.
  const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t thr16 = a.drop_p > 0.f ? min(65535u, static_cast<uint32_t>(a.drop_p * 65536.0f + 0.5f)) : 0u;
  const bool hashed_drop = EPI == kEpiReluDrop && a.drop_p > 0.f && a.keep == nullptr;
  uint32_t keep_lo = 0xFFFFFFFFu, keep_hi = 0xFFFFFFFFu;
  uint32_t drop_rid[2][4][2];
  if (EPI == kEpiReluDrop && hashed_drop && act) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = m0 + wm * 64 + mi * 32 + 8 * g + (lane >> 4) + 4 * h;
          uint32_t rid = 0u;
          if (row < a.M) {
            if (a.row_vdiv > 1) {
              const int drow = row / a.row_vdiv;
              const uint32_t trow = a.row_ids ? static_cast<uint32_t>(a.row_ids[drow]) : static_cast<uint32_t>(a.row_begin + drow);
              rid = trow * static_cast<uint32_t>(a.row_vdiv) + static_cast<uint32_t>(row - drow * a.row_vdiv);
            } else {
              rid = a.row_ids ? static_cast<uint32_t>(a.row_ids[row]) : static_cast<uint32_t>(a.row_begin + row);
            }
          }
          drop_rid[mi][g][h] = rid;
        }
  }
  auto drop_bits = [&]() {
    if (!(EPI == kEpiReluDrop && hashed_drop && act)) return;
    const uint32_t key = hash32(static_cast<uint32_t>(a.seed) ^ hash32(static_cast<uint32_t>(a.seed >> 32) + a.stream_id)) ^
                         hash32(static_cast<uint32_t>(a.step) * 0x9E3779B1u + 0x632BE5ABu);
    const int col = n0 + wn * 64 + (lane & 15) * 4;
    const uint32_t c0 = static_cast<uint32_t>(col >> 1) * 0x85EBCA77u, c1 = static_cast<uint32_t>((col >> 1) + 1) * 0x85EBCA77u;
    keep_lo = 0u;
    keep_hi = 0u;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t hrow = drop_row_hash(key, drop_rid[mi][g][h]);
          const uint32_t h0 = hash32(hrow ^ c0), h1 = hash32(hrow ^ c1);
          const uint32_t four = ((h0 & 0xFFFFu) >= thr16 ? 1u : 0u) | ((h0 >> 16) >= thr16 ? 2u : 0u) |
                                ((h1 & 0xFFFFu) >= thr16 ? 4u : 0u) | ((h1 >> 16) >= thr16 ? 8u : 0u);
          const int wi = ((mi * 4 + g) * 2 + h) * 4;  // compile-time after unrolling
          if (wi < 32) keep_lo |= four << wi; else keep_hi |= four << (wi - 32);
        }
  };
  float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
  uint32_t ok0 = 0, ok1 = 0;
  if constexpr (SPLIT > 0) {
    // Split path: LDS ping-pong, two register sets (while tile t is multiplied out of LDS, the loads of tiles t+1
    // and t+2 are in flight).  The main loop only stores full K steps, unmasked; each half of its body is one
    // basic block (ds_read, 24 MFMAs, the plane split of the next tile, ds_write, global loads, barrier).  The
    // tail (partial last step, grouped form) runs the same pipeline with masked stores.
    constexpr std::true_type kMask{};
    constexpr std::false_type kNoMask{};
    const int nfull = GRP != 0 ? 0 : (k_end - k_begin) / BK;  // tiles 0 .. nfull-1 lie entirely inside [k_begin, k_end)
    load_tiles(k_begin, ra0, rb0, ok0);
    drop_bits();
    store_tiles(k_begin, 0, ra0, rb0, ok0, kMask);
    load_tiles(k_begin + BK, ra0, rb0, ok0);
    load_tiles(k_begin + 2 * BK, ra1, rb1, ok1);
    __syncthreads();
    int t = 0;
    for (; t + 2 < nfull; t += 2) {
      compute(0);
      store_tiles(k_begin + (t + 1) * BK, 1, ra0, rb0, ok0, kNoMask);
      load_tiles(k_begin + (t + 3) * BK, ra0, rb0, ok0);
      __syncthreads();
      compute(1);
      store_tiles(k_begin + (t + 2) * BK, 0, ra1, rb1, ok1, kNoMask);
      load_tiles(k_begin + (t + 4) * BK, ra1, rb1, ok1);
      __syncthreads();
    }
    for (; t + 1 < nk; t += 2) {
      compute(0);
      store_tiles(k_begin + (t + 1) * BK, 1, ra0, rb0, ok0, kMask);
      load_tiles(k_begin + (t + 3) * BK, ra0, rb0, ok0);
      __syncthreads();
      compute(1);
      store_tiles(k_begin + (t + 2) * BK, 0, ra1, rb1, ok1, kMask);
      load_tiles(k_begin + (t + 4) * BK, ra1, rb1, ok1);
      __syncthreads();
    }
    if (t < nk) compute(0);
  } else {
  load_tiles(k_begin, ra0, rb0, ok0);
  drop_bits();
  store_tiles(k_begin, 0, ra0, rb0, ok0, std::true_type{});
  load_tiles(k_begin + BK, ra0, rb0, ok0);
  load_tiles(k_begin + 2 * BK, ra1, rb1, ok1);
  __syncthreads();
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    compute(0);
    store_tiles(k_begin + (t + 1) * BK, 1, ra0, rb0, ok0, std::true_type{});
    load_tiles(k_begin + (t + 3) * BK, ra0, rb0, ok0);
    __syncthreads();
    compute(1);
    store_tiles(k_begin + (t + 2) * BK, 0, ra1, rb1, ok1, std::true_type{});
    load_tiles(k_begin + (t + 4) * BK, ra1, rb1, ok1);
    __syncthreads();
  }
  if (t < nk) compute(0);
  }

  if constexpr (SPLIT == 4) {  // main + cross * 2^-11, and the A operand's pre-scale taken out again (powers of two: exact)
    const float inv_scale = 1.0f / a_scale;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + accx[i][j][r] * kF16LoInv) * inv_scale;
    kbias4.x *= inv_scale; kbias4.y *= inv_scale; kbias4.z *= inv_scale; kbias4.w *= inv_scale;
  }

  // ---- epilogue -------------------------------------------------------------
  // The accumulators (lane = column, 4 consecutive rows per register group) go through a per-wave LDS
  // patch of 8 rows x 64 columns and come back as float4 along the row, so that C is written (and the
  // activation mask read) with 16-byte accesses: a wave store covers 4 rows x 256 contiguous bytes.
  float* Cout = a.C + (EPI == kEpiSlab ? (size_t)bz * a.slab_stride : 0);
  if constexpr (KM_SPLIT) {
    if (bias_blk) {  // combine the four k groups in a fixed order, through the tile memory
      float4* comb = reinterpret_cast<float4*>(smem);
      __syncthreads();  // the last K step's fragment reads are done
      if (tid < 128) comb[tid] = kbias4;
      __syncthreads();
      if (tid < 32) {
        const float4 c0 = comb[tid], c1 = comb[32 + tid], c2 = comb[64 + tid], c3 = comb[96 + tid];
        const float r[4] = {((c0.x + c1.x) + c2.x) + c3.x, ((c0.y + c1.y) + c2.y) + c3.y, ((c0.z + c1.z) + c2.z) + c3.z,
                            ((c0.w + c1.w) + c2.w) + c3.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (m0 + tid * 4 + e < a.M) Cout[(size_t)(m0 + tid * 4 + e) * a.ldc + a.ones_col] = r[e];
      }
    }
  } else {
    if (do_bias && m0 + tid < a.M) Cout[(size_t)(m0 + tid) * a.ldc + a.ones_col] = bias_acc;
  }
  constexpr int LDP = 68;
  __syncthreads();  // every wave is done with the operand tiles
  float* patch = smem + wave * (8 * LDP);
  const int c4 = (lane & 15) * 4;
  const int col = n0 + wn * 64 + c4;  // first of this lane's 4 columns
  const int ncol = a.N - col;         // valid columns among them (may be <= 0 or >= 4)
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if ((EPI == kEpiBias || EPI == kEpiReluDrop) && a.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < ncol) {
        const int bi = GRP == 1 ? head_row(a.q0 + col + e, comp, a.hK, a.hD, a.hldK, a.hldD) : col + e;
        if (bi >= 0) bias4[e] = a.bias[bi];
      }
    }
  }
  // Every global load of a 32-row half (permuted output rows, dropout row ids, the activation / addend quads) is
  // issued before the half's first store: vmcnt counts stores as well, so a load behind a store waits for that
  // store's acknowledgement -- row by row that was a chain of sixteen round trips per wave (a third of the kernel
  // on the K <= 256 layers).
  const bool need_h = (EPI == kEpiMaskBwd && a.mask_in == nullptr) || (EPI == kEpiReluDrop && a.H != nullptr);
  const size_t mbase = ((size_t)(tile_m * tiles_n + tile_n) * 4 + wave) * 64;  // this wave's 64 sign words
  uint32_t sign_lo = 0u, sign_hi = 0u;  // this lane's word: bit ((mi*4 + g)*2 + h)*4 + e of its 16 quads
  if (EPI == kEpiMaskBwd && a.mask_in != nullptr && act) {
    const uint64_t wbits = a.mask_in[mbase + lane];
    sign_lo = static_cast<uint32_t>(wbits);
    sign_hi = static_cast<uint32_t>(wbits >> 32);
  }
  if (act) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row_tile = m0 + wm * 64 + mi * 32;
      if (row_tile >= a.M) continue;
      int orow_[4][2];
      float4 hq_[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = row_tile + 8 * g + (lane >> 4) + 4 * h;
          const bool ok = row < a.M && ncol > 0;
          int orow = ok ? row : -1;
          if (GRP == 1 && ok) orow = a.perm[row];
          orow_[g][h] = orow;
        }
      if (need_h) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            hq_[g][h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (orow_[g][h] >= 0) {
              const float* hp = a.H + (size_t)orow_[g][h] * a.ldc + col;
              if (ncol >= 4) hq_[g][h] = *reinterpret_cast<const float4*>(hp);
              else {  // ragged right edge
                hq_[g][h].x = hp[0];
                if (ncol > 1) hq_[g][h].y = hp[1];
                if (ncol > 2) hq_[g][h].z = hp[2];
              }
            }
          }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          patch[(4 * kh + j) * LDP + l31] = acc[mi][0][g * 4 + j];
          patch[(4 * kh + j) * LDP + 32 + l31] = acc[mi][1][g * 4 + j];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int lrow = (lane >> 4) + 4 * h;
          const float4 v4 = *reinterpret_cast<const float4*>(patch + lrow * LDP + c4);
          const int orow = orow_[g][h];
          if (orow < 0) continue;
          float v[4] = {v4.x, v4.y, v4.z, v4.w};
          const float hv[4] = {hq_[g][h].x, hq_[g][h].y, hq_[g][h].z, hq_[g][h].w};
          const size_t off = (size_t)orow * a.ldc + col;
          if (EPI == kEpiBias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias4[e];
          } else if (EPI == kEpiReluDrop) {
            {
              const int wi = ((mi * 4 + g) * 2 + h) * 4;  // compile-time after unrolling
              relu_drop_quad(a, v, hq_[g][h], bias4, hashed_drop, ((wi < 32 ? keep_lo : keep_hi) >> (wi & 31)) & 15u, col, ncol,
                             inv_keep, row_tile + 8 * g + lrow);
            }
            if (a.mask_out != nullptr) {  // v >= 0 here: positive <=> any bit set
              const int wi = ((mi * 4 + g) * 2 + h) * 4;  // compile-time after unrolling
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t one = min(__float_as_uint(v[e]), 1u) << ((wi + e) & 31);
                if (wi < 32) sign_lo |= one; else sign_hi |= one;
              }
            }
          } else if (EPI == kEpiMaskBwd) {
            if (a.mask_in != nullptr) {  // the forward's sign bits of this lane's quads (compile-time bit index)
              const int wi = ((mi * 4 + g) * 2 + h) * 4;  // compile-time after unrolling
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = (((wi < 32 ? sign_lo : sign_hi) >> ((wi + e) & 31)) & 1u) ? v[e] * inv_keep : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = hv[e] > 0.f ? v[e] * inv_keep : 0.f;
            }
          }
          if (ncol >= 4) {
            *reinterpret_cast<float4*>(Cout + off) = make_float4(v[0], v[1], v[2], v[3]);
          } else {  // ragged right edge: never touch the columns past N (the slab's bias column lives there)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < ncol) Cout[off + e] = v[e];
          }
        }
      }
    }
    if (EPI == kEpiReluDrop && a.mask_out != nullptr)
      a.mask_out[mbase + lane] = (static_cast<uint64_t>(sign_hi) << 32) | sign_lo;  // quads never stored: bits 0
  }
}

// Small batches (the reference's own defaults are 128 - 256 rows, suave/defaults.py:116-130): a 128 x 128 block tile
// leaves a handful of blocks on 256 CUs, each the latency chain of 13 - 16 K steps of ds_write -> barrier -> ds_read ->
// 12 MFMAs.  Here a BLOCK owns a 32 x 64 output tile and its four waves split the K range (wave w takes K steps
// w, w + 4, ...), so a layer of K <= 256 is one round: with row-major operands (forward layers: A [M, K], W [N, K])
// the fragment of v_mfma_f32_32x32x16_f16 -- lane l: row l & 31, k = 8 (l >> 5) .. + 7 -- is eight contiguous floats
// of the operand's row, so every lane loads its own fragments of up to four K steps straight from global memory in
// one go (L2 resident at these sizes; no LDS staging, no barrier in the loop), cuts them into the two fp16 planes in
// registers and issues the three plane products per 32 x 32 tile.  The four partial tiles meet in LDS; wave w then
// finishes register groups 4 w .. 4 w + 3 of both 32-column halves: bias, ReLU, dropout (the same per-(row, column
// pair) hash as gemm_body, or the injected mask).  The accumulator layout (lane = column, 4 consecutive rows per
// register group) stores 128 contiguous bytes per half wave and row.  Arithmetic mode 4 only.
// GRP == 1: the selected-component encoder head (GemmArgs::grouped == 1) -- rows of A and C are sorted positions read /
// written through perm (a 32-row tile lies in one component's chunk, its valid rows are a prefix), rows of B are
// head_row(q0 + ., component).  B_KM (with GRP == 1, the head's data gradient): element (k, j) of B is
// W[head_row(q0 + k, component)][j] -- eight dword loads per fragment, coalesced across the lanes' columns.
template <int EPI, int GRP = 0, bool B_KM = false>
__device__ __forceinline__ void gemm_small_body(const GemmArgs& a, int block, float (&part_s)[4][32][64]) {  // part_s: [wave][j * 16 + r][lane]
  static_assert(GRP == 0 || EPI != kEpiReluDrop, "the grouped forms have no dropout epilogue");
  static_assert(!B_KM || GRP == 1, "k-major B: only the grouped data gradient");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (a.N + 63) / 64;
  const int tile_m = block / tiles_n, tile_n = block - tile_m * tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 64;
  const int l31 = lane & 31, kh = lane >> 5;
  int comp = 0;
  if (GRP == 1) {  // block-uniform exits, before any barrier
    comp = a.chunk_comp[m0 / kGroupRows];
    if (comp < 0 || a.perm[m0] < 0) return;  // unused chunk / a tile of padding positions only
  }
  // out-of-range rows / columns read a valid row and are never stored (GRP: absent rows read as zeros)
  const int arow = GRP == 1 ? a.perm[m0 + l31] : min(m0 + l31, a.M - 1);
  const bool a_ok = GRP == 1 ? arow >= 0 : true;
  int brow0 = min(n0 + l31, a.N - 1), brow1 = min(n0 + 32 + l31, a.N - 1);
  bool b_ok0 = true, b_ok1 = true;
  if (GRP == 1 && !B_KM) {
    brow0 = head_row(a.q0 + brow0, comp, a.hK, a.hD, a.hldK, a.hldD);
    brow1 = head_row(a.q0 + brow1, comp, a.hK, a.hD, a.hldK, a.hldD);
    b_ok0 = brow0 >= 0; b_ok1 = brow1 >= 0;
  }
  const float* pa = a.A + (size_t)max(arow, 0) * a.lda + 8 * kh;
  const float* pb0 = B_KM ? a.B + brow0 : a.B + (size_t)max(brow0, 0) * a.ldb + 8 * kh;
  const float* pb1 = B_KM ? a.B + brow1 : a.B + (size_t)max(brow1, 0) * a.ldb + 8 * kh;
  const bool two = n0 + 32 < a.N;  // block-uniform: the second 32 columns exist
  f32x16 acc[2], accx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accx[j][r] = 0.f; }
  auto load8 = [&](const float* p, bool ok, int k0, float (&v)[8]) {
    // lda / ldb are multiples of 4 floats: a quad that starts inside the K range lies inside the row's storage
    const int k = k0 + 8 * kh;
    const float4 q0 = (ok && k < a.K) ? *reinterpret_cast<const float4*>(p + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 q1 = (ok && k + 4 < a.K) ? *reinterpret_cast<const float4*>(p + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (k + e >= a.K) v[e] = 0.f;  // ragged K inside a quad
  };
  auto load8_km = [&](const float* pcol, int k0, float (&v)[8]) {  // B_KM: rows head_row(q0 + k), this lane's column
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + 8 * kh + e;
      const int br = k < a.K ? head_row(a.q0 + k, comp, a.hK, a.hD, a.hldK, a.hldD) : -1;
      v[e] = br >= 0 ? pcol[(size_t)br * a.ldb] : 0.f;
    }
  };
  const float a_scale = a.a_scale > 0.f ? a.a_scale : 1.0f;  // gradient operands: pre-scaled into the fp16 range
  auto planes = [&](const float (&v)[8], float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = v[e] * scale;
      const _Float16 h = (_Float16)x;
      hi[e] = h;
      lo[e] = (_Float16)((x - (float)h) * kF16LoScale);
    }
  };
  constexpr int G = 4;  // K steps whose loads are in flight together
  const int n_steps = (a.K + BK - 1) / BK;
  for (int s0 = wave; s0 < n_steps; s0 += 4 * G) {
    float va[G][8], vb0[G][8], vb1[G][8];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int k0 = (s0 + 4 * u) * BK;  // (past the end: load8 returns zeros without touching memory)
      load8(pa, a_ok, k0, va[u]);
      if (B_KM) {
        load8_km(pb0, k0, vb0[u]);
        if (two) load8_km(pb1, k0, vb1[u]);
      } else {
        load8(pb0, b_ok0, k0, vb0[u]);
        if (two) load8(pb1, b_ok1, k0, vb1[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (s0 + 4 * u >= n_steps) break;  // wave-uniform
      f16x8 ah, al, bh, bl;
      planes(va[u], a_scale, ah, al);
      planes(vb0[u], 1.0f, bh, bl);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
      accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[0], 0, 0, 0);
      accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[0], 0, 0, 0);
      if (two) {
        planes(vb1[u], 1.0f, bh, bl);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
        accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[1], 0, 0, 0);
        accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[1], 0, 0, 0);
      }
    }
  }
  const float inv_scale = 1.0f / a_scale;  // (a power of two: exact)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) part_s[wave][j * 16 + r][lane] = (acc[j][r] + accx[j][r] * kF16LoInv) * inv_scale;
  // this wave's share of the epilogue: register groups 4 wave .. 4 wave + 3 of both halves (rows 8 wave + 4 kh + 0..3)
  const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const bool hashed_drop = EPI == kEpiReluDrop && a.drop_p > 0.f && a.keep == nullptr;
  uint32_t keep_bits = 0xFFu;
  if (hashed_drop) {  // (before the barrier: the hashes fill the wait for the other waves)
    const uint32_t thr16 = min(65535u, static_cast<uint32_t>(a.drop_p * 65536.0f + 0.5f));
    const uint32_t key = hash32(static_cast<uint32_t>(a.seed) ^ hash32(static_cast<uint32_t>(a.seed >> 32) + a.stream_id)) ^
                         hash32(static_cast<uint32_t>(a.step) * 0x9E3779B1u + 0x632BE5ABu);
    keep_bits = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 8 * wave + 4 * kh + i;
      uint32_t rid = 0u;
      if (row < a.M) {
        if (a.row_vdiv > 1) {
          const int drow = row / a.row_vdiv;
          const uint32_t trow = a.row_ids ? static_cast<uint32_t>(a.row_ids[drow]) : static_cast<uint32_t>(a.row_begin + drow);
          rid = trow * static_cast<uint32_t>(a.row_vdiv) + static_cast<uint32_t>(row - drow * a.row_vdiv);
        } else {
          rid = a.row_ids ? static_cast<uint32_t>(a.row_ids[row]) : static_cast<uint32_t>(a.row_begin + row);
        }
      }
      const uint32_t hrow = drop_row_hash(key, rid);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + 32 * j + l31;
        const uint32_t h = hash32(hrow ^ (static_cast<uint32_t>(col >> 1) * 0x85EBCA77u));
        const uint32_t bits16 = (col & 1) ? (h >> 16) : (h & 0xFFFFu);
        keep_bits |= (bits16 >= thr16 ? 1u : 0u) << (4 * j + i);
      }
    }
  }
  int orow[4];  // output rows of this lane (GRP: through perm; -1 = nothing to store)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + 8 * wave + 4 * kh + i;
    orow[i] = row < a.M ? (GRP == 1 ? a.perm[row] : row) : -1;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 32 * j + l31;
    if (col >= a.N) continue;
    float bias = 0.f;
    if (EPI != kEpiMaskBwd && a.bias != nullptr) {
      const int bi = GRP == 1 ? head_row(a.q0 + col, comp, a.hK, a.hD, a.hldK, a.hldD) : col;
      if (bi >= 0) bias = a.bias[bi];
    }
    float hv[4] = {1.f, 1.f, 1.f, 1.f};
    if (EPI == kEpiMaskBwd) {  // the layer's output activation (post ReLU / dropout): all four loads before the stores
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (orow[i] >= 0) hv[i] = a.H[(size_t)orow[i] * a.ldc + col];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = orow[i];
      if (row < 0) continue;
      const int slot = j * 16 + 4 * wave + i;
      float v = ((part_s[0][slot][lane] + part_s[1][slot][lane]) + part_s[2][slot][lane]) + part_s[3][slot][lane] + bias;
      if (EPI == kEpiReluDrop) {
        v = fmaxf(v, 0.f);
        if (a.drop_p > 0.f) {
          const bool keep = hashed_drop ? ((keep_bits >> (4 * j + i)) & 1u) != 0u : a.keep[(size_t)row * a.N + col] != 0;
          v = keep ? v * inv_keep : 0.f;
        }
      } else if (EPI == kEpiMaskBwd) {
        v = hv[i] > 0.f ? v * inv_keep : 0.f;
      }
      a.C[(size_t)row * a.ldc + col] = v;
    }
  }
}

// The weight gradient of a small batch: C[split][m][n] = sum over the split's batch rows r of A[r][m] * B[r][n]
// (A = dC [rows, lda], B = the layer input [rows, ldb]; both "k-major": the reduction index is the row), + the bias
// gradient (plain fp32 column sums of dC) in column ones_col.  Same shape as gemm_small_kernel: a block owns a 32 x 64
// tile of one split, its four waves take the split's 16-row steps round-robin with the loads of four steps in flight
// (a fragment is eight rows of one column: eight dword loads, coalesced across the lanes), partial tiles meet in LDS.
// GRP == 2 (the selected-component head, GemmArgs::grouped == 2): split = chunk of kGroupRows sorted positions, the
// batch row of a position read through perm (-1: padding, contributes zeros).
template <int GRP = 0>
__device__ __forceinline__ void gemm_small_wgrad_body(const GemmArgs& a, int rows_per_split, int tile, int split,
                                                      float (&part_s)[4][32][64], float (&bias_s)[4][32]) {
  if (GRP == 2 && a.chunk_comp[split] < 0) return;  // unused chunk (block-uniform, before any barrier)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (a.N + 63) / 64;
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 64;
  const int l31 = lane & 31, kh = lane >> 5;
  const int r_begin = split * rows_per_split, r_end = min(a.K, r_begin + rows_per_split);
  const float* pa = a.A + min(m0 + l31, a.M - 1);  // out-of-range columns read a valid one and are never stored
  const float* pb0 = a.B + min(n0 + l31, a.N - 1);
  const float* pb1 = a.B + min(n0 + 32 + l31, a.N - 1);
  const bool two = n0 + 32 < a.N;
  const float a_scale = a.a_scale > 0.f ? a.a_scale : 1.0f;
  f32x16 acc[2], accx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accx[j][r] = 0.f; }
  auto rows8 = [&](int r0, int (&src)[8]) {  // source rows of a step's fragment (-1: none)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = r0 + 8 * kh + e;
      src[e] = r < r_end ? (GRP == 2 ? a.perm[r] : r) : -1;
    }
  };
  auto load8 = [&](const float* p, int ld, const int (&src)[8], float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e] >= 0 ? p[(size_t)src[e] * ld] : 0.f;
  };
  auto planes = [&](const float (&v)[8], float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = v[e] * scale;
      const _Float16 h = (_Float16)x;
      hi[e] = h;
      lo[e] = (_Float16)((x - (float)h) * kF16LoScale);
    }
  };
  constexpr int G = 4;
  const int n_steps = (max(r_end - r_begin, 0) + BK - 1) / BK;
  float bsum = 0.f;
  for (int s0 = wave; s0 < n_steps; s0 += 4 * G) {
    float va[G][8], vb0[G][8], vb1[G][8];
    int src[G][8];
#pragma unroll
    for (int u = 0; u < G; ++u) rows8(r_begin + (s0 + 4 * u) * BK, src[u]);  // (past the split's end: -1, no memory access)
#pragma unroll
    for (int u = 0; u < G; ++u) {
      load8(pa, a.lda, src[u], va[u]);
      load8(pb0, a.ldb, src[u], vb0[u]);
      if (two) load8(pb1, a.ldb, src[u], vb1[u]);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (s0 + 4 * u >= n_steps) break;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum += va[u][e];
      f16x8 ah, al, bh, bl;
      planes(va[u], a_scale, ah, al);
      planes(vb0[u], 1.0f, bh, bl);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
      accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[0], 0, 0, 0);
      accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[0], 0, 0, 0);
      if (two) {
        planes(vb1[u], 1.0f, bh, bl);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
        accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[1], 0, 0, 0);
        accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[1], 0, 0, 0);
      }
    }
  }
  const float inv_scale = 1.0f / a_scale;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) part_s[wave][j * 16 + r][lane] = (acc[j][r] + accx[j][r] * kF16LoInv) * inv_scale;
  bsum += __shfl_xor(bsum, 32);
  if (lane < 32) bias_s[wave][lane] = bsum;
  __syncthreads();
  float* Cout = a.C + (size_t)split * a.slab_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 32 * j + l31;
    if (col >= a.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 8 * wave + 4 * kh + i;
      if (row >= a.M) continue;
      const int slot = j * 16 + 4 * wave + i;
      Cout[(size_t)row * a.ldc + col] =
          ((part_s[0][slot][lane] + part_s[1][slot][lane]) + part_s[2][slot][lane]) + part_s[3][slot][lane];
    }
  }
  if (a.ones_col >= 0 && tile_n == 0 && wave == 0 && lane < 32 && m0 + lane < a.M) {
    const float bsum_all = ((bias_s[0][lane] + bias_s[1][lane]) + bias_s[2][lane]) + bias_s[3][lane];
    if (a.bias_out != nullptr) a.bias_out[m0 + lane] = bsum_all;  // direct form (one split): straight into the gradient
    else Cout[(size_t)(m0 + lane) * a.ldc + a.ones_col] = bsum_all;
  }
}

// XCD-aware order of the small-batch launches: workgroups go to the eight XCDs round-robin, so dispatch positions
// p, p + 8, ... share an L2; they get CONSECUTIVE work items (the column tiles of one row tile read the same A rows,
// the tiles of one weight-gradient split the same batch rows).  0.2 - 0.7 % of a 1 024 - 8 192-row step.
__device__ __forceinline__ int xcd_consecutive(int p, int n) {
  const int per = n >> 3;
  return p < per * 8 ? (p & 7) * per + (p >> 3) : p;
}

template <int EPI, int GRP = 0>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs a) {
  __shared__ float part_s[4][32][64];
  gemm_small_body<EPI, GRP, false>(a, xcd_consecutive(blockIdx.x, gridDim.x), part_s);
}

__global__ __launch_bounds__(256) void gemm_small_wgrad_kernel(GemmArgs a, int rows_per_split) {
  __shared__ float part_s[4][32][64];
  __shared__ float bias_s[4][32];
  const int lin = xcd_consecutive(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);  // (tile fastest)
  gemm_small_wgrad_body(a, rows_per_split, lin % (int)gridDim.x, lin / (int)gridDim.x, part_s, bias_s);
}

// Both backward GEMMs of a layer over a small batch in one launch (they are independent and neither fills the chip):
// blocks [0, w_blocks) are the weight gradient's (tile, split) pairs, the rest the data gradient's tiles.
template <int DEPI, int GRP = 0>
__global__ __launch_bounds__(256) void gemm_small_pair_kernel(GemmArgs d, GemmArgs w, int rows_per_split, int w_tiles, int w_blocks) {
  __shared__ float part_s[4][32][64];
  __shared__ float bias_s[4][32];
  const int b = blockIdx.x;
  if (b < w_blocks) {
    const int wb = xcd_consecutive(b, w_blocks);
    gemm_small_wgrad_body<GRP == 1 ? 2 : 0>(w, rows_per_split, wb % w_tiles, wb / w_tiles, part_s, bias_s);
  } else {
    gemm_small_body<DEPI, GRP, GRP == 1>(d, xcd_consecutive(b - w_blocks, (int)gridDim.x - w_blocks), part_s);
  }
}

// Weight-gradient launches are (tiles, K splits) grids of a few tiles: the tiles of ONE split read the same rows of
// both operands (dC and the layer input), so they should meet in one L2.  Workgroups go to the XCDs round-robin by
// dispatch order; this maps dispatch position `lin` of a (T tiles, S splits) launch to (tile, split) so that the T
// tiles of a split share an XCD and are dispatched within 8 T positions of each other.  (Dispatched tile-fastest,
// the tiles of a split sat on T different XCDs and each operand came out of HBM once per tile: 260 MB read for
// the 121 MB of the first encoder layer's weight gradient.)
__device__ __forceinline__ void xcd_split_order(int lin, int T, int S, int* tile, int* split) {
  const int full = S / 8 * 8;  // splits in whole groups of eight
  if (lin < full * T) {
    const int g = lin / (8 * T), j = lin - g * 8 * T;
    *split = g * 8 + (j & 7);
    *tile = j >> 3;
  } else {  // the last, partial group: plain order
    const int r = lin - full * T;
    *split = full + r / T;
    *tile = r % T;
  }
}

template <int WAVES_M, int WAVES_N, bool A_KM, bool B_KM, int GRP, int EPI, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT == 4 ? 2 : SPLIT > 0 ? 3 : 1) void gemm_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<WAVES_M, WAVES_N, SPLIT, A_KM>()];
  __shared__ int perm_s[GRP == 2 ? kGroupRows : 1];
  int tile = blockIdx.x, split = blockIdx.z;
  if (EPI == kEpiSlab && GRP == 0 && gridDim.z > 1) xcd_split_order(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x, gridDim.z, &tile, &split);
  gemm_body<WAVES_M, WAVES_N, A_KM, B_KM, GRP, EPI, SPLIT>(a, tile, split, smem, perm_s);
}

// The two backward GEMMs of one layer in one launch: they read the same dC and do not depend on each
// other.  The weight-gradient blocks (few and long: one per CU) are dispatched first, the data-gradient
// blocks fill the rest of each CU.
template <int DM, int DN, int D_GRP, int D_EPI, int WM, int WN, int W_GRP, int D_SPLIT = 0, int WG_SPLIT = 0>
__global__ __launch_bounds__(256, (D_SPLIT > 0 || WG_SPLIT > 0) ? 2 : 1) void gemm_pair_kernel(GemmArgs d, GemmArgs w, int w_blocks, int w_tiles) {
  // the weight-gradient half joins the split path on the 128x128 tile (D_SPLIT: ungrouped, WG_SPLIT: grouped by
  // component -- its data-gradient partner gathers W rows and stays exact); otherwise exact fp32
  constexpr int W_SPLIT = (WM == 2 && WN == 2) ? (W_GRP == 0 ? D_SPLIT : WG_SPLIT) : 0;
  constexpr int kSmemW = gemm_smem_floats<WM, WN, W_SPLIT, true>();
  constexpr int kSmem = gemm_smem_floats<DM, DN, D_SPLIT>() > kSmemW ? gemm_smem_floats<DM, DN, D_SPLIT>() : kSmemW;
  __shared__ __attribute__((aligned(16))) float smem[kSmem];
  __shared__ int perm_s[W_GRP == 2 ? kGroupRows : 1];
  const int b = blockIdx.x;
  if (b < w_blocks) {
    int tile = b % w_tiles, split = b / w_tiles;
    if (W_GRP == 0) xcd_split_order(b, w_tiles, w_blocks / w_tiles, &tile, &split);
    gemm_body<WM, WN, true, true, W_GRP, kEpiSlab, W_SPLIT>(w, tile, split, smem, perm_s);
  }
  else gemm_body<DM, DN, false, D_SPLIT == 0, D_GRP, D_EPI, D_SPLIT>(d, b - w_blocks, 0, smem, perm_s);  // split: row-major W^T
}

// All wide categorical columns in one launch: a block looks up its column in the tile prefix table and runs
// the ordinary tile body on that column's operands.
template <int MODE, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT == 4 ? 2 : SPLIT > 0 ? 3 : 1) void gemm_wide_kernel(WideGemmArgs w) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<2, 2, SPLIT, MODE == 2>()];
  __shared__ int perm_s[1];
  const int tiles_m = MODE == 2 ? 1 : (w.rows + 127) / 128;
  const int q = blockIdx.x / tiles_m;
  // binary search: cum[c] <= q < cum[c + 1]   (uniform across the block)
  int lo = 0, hi = w.n_wide;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (w.cum[mid] <= q) lo = mid; else hi = mid;
  }
  const WideCol wc = w.cols[lo];
  const int local = blockIdx.x - w.cum[lo] * tiles_m;
  GemmArgs g;
  g.bias = nullptr; g.H = nullptr; g.keep = nullptr; g.row_ids = nullptr; g.row_begin = 0;
  g.splits = 1; g.slab_stride = 0; g.ones_col = -1; g.drop_p = 0.f; g.seed = 0; g.step = 0; g.stream_id = 0;
  g.grouped = 0; g.perm = nullptr; g.chunk_comp = nullptr; g.hK = g.hD = g.hldK = g.hldD = g.q0 = 0;
  g.row_vdiv = 0; g.a_scale = MODE == 0 ? 1.0f : w.grad_scale; g.mask_out = nullptr; g.mask_in = nullptr;
  float* slot = w.y + wc.y_off;
  float* logit = w.L + wc.l_off;
  const float* wt = w.params + wc.w_off;
  if (MODE == 0) {
    g.A = slot; g.B = wt; g.C = logit; g.M = w.rows; g.N = wc.C - 1; g.K = wc.yd + w.K;
    g.lda = w.ldY; g.ldb = wc.ldw; g.ldc = w.ldL; g.a_kmajor = 0; g.b_kmajor = 0; g.epi = kEpiBias;
    gemm_body<2, 2, false, false, 0, kEpiBias, SPLIT>(g, local, 0, smem, perm_s);
  } else if (MODE == 1) {
    g.A = logit; g.B = wt; g.C = slot; g.M = w.rows; g.N = wc.ldw; g.K = wc.C - 1;
    g.lda = w.ldL; g.ldc = w.ldY; g.a_kmajor = 0; g.epi = kEpiBias;
    if constexpr (SPLIT > 0) {  // row-major W^T from the workspace (k_wide_transpose)
      g.B = w.wT + wc.t_off; g.ldb = (wc.C - 1 + 3) & ~3; g.b_kmajor = 0;
      gemm_body<2, 2, false, false, 0, kEpiBias, SPLIT>(g, local, 0, smem, perm_s);
    } else {
      g.ldb = wc.ldw; g.b_kmajor = 1;
      gemm_body<2, 2, false, true, 0, kEpiBias>(g, local, 0, smem, perm_s);
    }
  } else {
    g.A = logit; g.B = slot; g.C = w.slab + (wc.w_off - w.wide_w0); g.M = wc.C - 1; g.N = wc.ldw; g.K = w.rows;
    g.lda = w.ldL; g.ldb = w.ldY; g.ldc = wc.ldw; g.a_kmajor = 1; g.b_kmajor = 1; g.epi = kEpiSlab;
    g.splits = w.splits; g.slab_stride = w.slab_stride;
    gemm_body<2, 2, true, true, 0, kEpiSlab, SPLIT>(g, local, blockIdx.z, smem, perm_s);
  }
}

}  // namespace

int launch_gemm_wide(const WideGemmArgs& a, int mode, hipStream_t stream) {
  if (a.n_wide <= 0 || a.rows <= 0) return 0;
  SUAVE_REQUIRE(a.ldY % 4 == 0 && a.ldL % 4 == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.L) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.params) & 15) == 0,
                "wide gemm: operands must be 16-byte aligned");
  SUAVE_REQUIRE(mode != 2 || (a.slab != nullptr && a.splits >= 1 && a.slab_stride % 4 == 0 &&
                              (reinterpret_cast<uintptr_t>(a.slab) & 15) == 0), "wide gemm: bad slab");
  ProfScope prof_scope(kProfGemm, stream);
  const int tiles_m = mode == 2 ? 1 : (a.rows + 127) / 128;
  const int64_t blocks = (int64_t)a.cum_total * tiles_m;
  SUAVE_REQUIRE(blocks > 0 && blocks < (int64_t(1) << 31), "wide gemm: grid out of range");
  const dim3 grid((unsigned)blocks, 1, mode == 2 ? a.splits : 1), block(256);
  const int sp = split_mode();
  if (mode == 0 && sp == 4) hipLaunchKernelGGL((gemm_wide_kernel<0, 4>), grid, block, 0, stream, a);
  else if (mode == 0 && sp == 3) hipLaunchKernelGGL((gemm_wide_kernel<0, 3>), grid, block, 0, stream, a);
  else if (mode == 0) hipLaunchKernelGGL(gemm_wide_kernel<0>, grid, block, 0, stream, a);
  else if (mode == 1 && sp == 4 && a.wT != nullptr) hipLaunchKernelGGL((gemm_wide_kernel<1, 4>), grid, block, 0, stream, a);
  else if (mode == 1 && sp == 3 && a.wT != nullptr) hipLaunchKernelGGL((gemm_wide_kernel<1, 3>), grid, block, 0, stream, a);
  else if (mode == 1) hipLaunchKernelGGL(gemm_wide_kernel<1>, grid, block, 0, stream, a);
  else if (sp == 4) hipLaunchKernelGGL((gemm_wide_kernel<2, 4>), grid, block, 0, stream, a);
  else if (sp == 3) hipLaunchKernelGGL((gemm_wide_kernel<2, 3>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL(gemm_wide_kernel<2>, grid, block, 0, stream, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {
int check_gemm(const GemmArgs& a) {
  SUAVE_REQUIRE(a.lda % 4 == 0 && a.ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4 floats");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && a.ldc % 4 == 0 && a.slab_stride % 4 == 0 &&
                    ((a.epi != kEpiMaskBwd && a.epi != kEpiReluDrop) || (reinterpret_cast<uintptr_t>(a.H) & 15) == 0),
                "gemm: operands must be 16-byte aligned");
  if (a.grouped == 0) {
    SUAVE_REQUIRE(a.a_kmajor ? a.lda >= a.M : a.lda >= a.K, "gemm: lda too small");
    SUAVE_REQUIRE(a.b_kmajor ? a.ldb >= a.N : a.ldb >= a.K, "gemm: ldb too small");
  } else {
    SUAVE_REQUIRE(a.perm != nullptr && a.chunk_comp != nullptr, "gemm: grouped form needs perm and chunk_comp");
    SUAVE_REQUIRE((a.grouped == 1 ? a.M : a.K) % kGroupRows == 0, "gemm: grouped extent must be whole chunks");
    SUAVE_REQUIRE(a.grouped == 1 ? !a.a_kmajor : (a.a_kmajor && a.b_kmajor && a.epi == kEpiSlab && a.splits == a.K / kGroupRows),
                  "gemm: unsupported grouped operand layout");
  }
  SUAVE_REQUIRE(a.ldc >= a.N + ((a.ones_col >= 0 && a.bias_out == nullptr) ? 1 : 0), "gemm: ldc too small");
  SUAVE_REQUIRE(a.ones_col < 0 || a.ones_col == a.N, "gemm: the bias column sits right after the N real columns");
  SUAVE_REQUIRE(a.epi != kEpiSlab || a.splits >= 1, "gemm: bad split count");
  return 0;
}
inline bool gemm_narrow(const GemmArgs& a) { return a.N <= 64; }
// SUAVE_GEMM_SPLIT = 4 (default: two fp16 planes, scaled residual, three products), 3 (three bf16 planes, six
// products) or 0 (exact-fp32 MFMA everywhere).  Only the row-major forms and the 128x128 k-major form have split
// kernels.  A two-plane BF16 split (three products) was measured in round 1 and dropped: 5e-6 per GEMM on
// well-scaled data, but 5e-2 on the decoder weight gradients of the std = 12 real columns, whose dY spans three
// decades within a row (16 significand bits); the fp16 pair keeps 22 bits and measures like mode 3
// (scripts/split_accuracy.py on the GPU, scripts/split_emulate.py on the CPU).
std::atomic<int>& split_mode_cell() {
  static std::atomic<int> mode{[] {
    const char* e = getenv("SUAVE_GEMM_SPLIT");
    return (e && atoi(e) == 0) ? 0 : (e && atoi(e) == 3) ? 3 : 4;
  }()};
  return mode;
}
int split_mode() { return split_mode_cell().load(std::memory_order_relaxed); }
inline int gemm_tiles(const GemmArgs& a) {
  return gemm_narrow(a) ? ((a.M + 255) / 256) * ((a.N + 63) / 64) : ((a.M + 127) / 128) * ((a.N + 127) / 128);
}
}  // namespace

int gemm_split_mode() { return split_mode(); }
// Rows up to which the row-major forward forms run on gemm_small_kernel (arithmetic mode 4 only), and only inside an
// ELBO step (GemmSmallScope): the split-K summation order differs from gemm_body's in the last bits, and the
// inference entry points promise results that do not depend on how the caller chunks the rows.
namespace { thread_local int t_small_scope = 0; }
GemmSmallScope::GemmSmallScope() { ++t_small_scope; }
GemmSmallScope::~GemmSmallScope() { --t_small_scope; }
namespace {
std::atomic<int>& small_rows_cell() {
  static std::atomic<int> rows{[] { const char* e = getenv("SUAVE_GEMM_SMALL_M"); return e ? std::max(0, atoi(e)) : 8192; }()};
  return rows;
}
}  // namespace
int64_t gemm_small_rows_limit() { return small_rows_cell().load(std::memory_order_relaxed); }
int gemm_small_rows_limit_set(int rows) {
  SUAVE_REQUIRE(rows >= 0 && rows <= 8192, "gemm: the small-batch path takes ELBO steps of 0 (off) .. 8192 rows");
  small_rows_cell().store(rows, std::memory_order_relaxed);
  return 0;
}
bool gemm_small_rows(int64_t M) { return t_small_scope > 0 && M <= gemm_small_rows_limit() && split_mode() == 4; }
int64_t gemm_mask_words(int64_t M, int N) {
  const int64_t tiles = N <= 64 ? ((M + 255) / 256) * ((N + 63) / 64) : ((M + 127) / 128) * ((N + 127) / 128);
  return tiles * 256;  // 4 waves x 64 words per tile
}
int gemm_split_mode_set(int mode) {
  SUAVE_REQUIRE(mode == 0 || mode == 3 || mode == 4,
                "gemm: the arithmetic mode is 0 (exact-fp32 MFMA), 3 (three bf16 planes) or 4 (two fp16 planes)");
  split_mode_cell().store(mode, std::memory_order_relaxed);
  return 0;
}

namespace {
// forward layers and data gradients on row-major operands (the data gradient reads the transposed weight copy)
bool small_rowmajor_ok(const GemmArgs& a) {
  if (!gemm_small_rows(a.M) || a.a_kmajor || a.b_kmajor || a.grouped != 0 || a.mask_out != nullptr || a.mask_in != nullptr) return false;
  if (a.epi == kEpiMaskBwd) return a.H != nullptr;
  return (a.epi == kEpiBias || a.epi == kEpiReluDrop) && a.H == nullptr;
}
// the selected-component head over a small batch: forward (grouped == 1, row-major W) / data gradient (k-major W) /
// weight gradient (grouped == 2).  The grouped extent is ceil(rows / kGroupRows) + K chunks of kGroupRows positions.
bool small_grouped_rows(const GemmArgs& a) {
  static const bool off = getenv("SUAVE_NO_SMALL_GROUPED") != nullptr;  // tuning knob
  const int64_t ext = a.grouped == 1 ? a.M : a.K;
  return !off && a.perm != nullptr && a.chunk_comp != nullptr && gemm_small_rows(ext - (int64_t)a.hK * kGroupRows);
}
bool small_grouped_fwd_ok(const GemmArgs& a) {
  static const bool off = getenv("SUAVE_NO_SMALL_GFWD") != nullptr;  // tuning knob
  return !off && a.grouped == 1 && small_grouped_rows(a) && !a.a_kmajor && !a.b_kmajor && a.epi == kEpiBias && a.H == nullptr;
}
bool small_grouped_dgrad_ok(const GemmArgs& a) {
  return a.grouped == 1 && small_grouped_rows(a) && !a.a_kmajor && a.b_kmajor && a.epi == kEpiMaskBwd && a.H != nullptr &&
         a.mask_in == nullptr;
}
bool small_grouped_wgrad_ok(const GemmArgs& a) {
  return a.grouped == 2 && small_grouped_rows(a) && a.a_kmajor && a.b_kmajor && a.epi == kEpiSlab &&
         a.splits == a.K / kGroupRows;
}
// weight gradients over a small batch (K = batch rows; at most 64 steps per block)
bool small_wgrad_ok(const GemmArgs& a) {
  static const bool off = getenv("SUAVE_NO_SMALL_WGRAD") != nullptr;  // tuning knob
  return !off && gemm_small_rows(a.K) && a.a_kmajor && a.b_kmajor && a.grouped == 0 && a.epi == kEpiSlab && a.splits >= 1 &&
         (a.K + a.splits - 1) / a.splits <= 1024;
}
}  // namespace

// A weight gradient over `rows` batch rows in ONE split on the small-batch kernel can write the gradient itself
// (C = the layer's weight gradient, GemmArgs::bias_out = its bias gradient): no slab, no reduction segment.
bool gemm_wgrad_direct_ok(int64_t rows, int splits) {
  static const bool off = getenv("SUAVE_NO_SMALL_WGRAD") != nullptr || getenv("SUAVE_NO_SMALL_BWD") != nullptr ||
                          getenv("SUAVE_NO_DIRECT_WGRAD") != nullptr;  // tuning knobs
  return !off && splits == 1 && gemm_small_rows(rows) && rows <= 1024;
}

int launch_gemm(const GemmArgs& a, int n_cu, hipStream_t stream) {
  (void)n_cu;
  if (a.M <= 0 || a.N <= 0) return 0;
  if (int rc = check_gemm(a)) return rc;
  SUAVE_REQUIRE(a.bias_out == nullptr || small_wgrad_ok(a), "gemm: the direct weight-gradient form needs the small-batch kernel");
  const int splits = a.epi == kEpiSlab ? a.splits : 1;
  GemmArgs args = a;
  args.splits = splits;
  ProfScope prof_scope(kProfGemm, stream);
  // small batches (inside an ELBO step, arithmetic mode 4): the split-K kernels
  if (small_rowmajor_ok(a)) {
    const dim3 sgrid(((a.M + 31) / 32) * ((a.N + 63) / 64)), sblock(256);
    if (a.epi == kEpiBias) hipLaunchKernelGGL(gemm_small_kernel<kEpiBias>, sgrid, sblock, 0, stream, args);
    else if (a.epi == kEpiReluDrop) hipLaunchKernelGGL(gemm_small_kernel<kEpiReluDrop>, sgrid, sblock, 0, stream, args);
    else hipLaunchKernelGGL(gemm_small_kernel<kEpiMaskBwd>, sgrid, sblock, 0, stream, args);
    SUAVE_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (small_grouped_fwd_ok(a)) {
    const dim3 sgrid(((a.M + 31) / 32) * ((a.N + 63) / 64)), sblock(256);
    hipLaunchKernelGGL((gemm_small_kernel<kEpiBias, 1>), sgrid, sblock, 0, stream, args);
    SUAVE_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (small_wgrad_ok(a)) {
    const int per = ((a.K + splits - 1) / splits + BK - 1) / BK * BK;
    const dim3 sgrid(((a.M + 31) / 32) * ((a.N + 63) / 64), splits), sblock(256);
    hipLaunchKernelGGL(gemm_small_wgrad_kernel, sgrid, sblock, 0, stream, args, per);
    SUAVE_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const bool narrow = gemm_narrow(a);
  const int tiles = gemm_tiles(a);
  const dim3 grid(tiles, 1, splits), block(256);
#define SUAVE_GEMM_LAUNCH(AKM, BKM, GRP, EPI)                                                              \
  do {                                                                                                     \
    if (narrow) hipLaunchKernelGGL((gemm_kernel<4, 1, AKM, BKM, GRP, EPI>), grid, block, 0, stream, args); \
    else hipLaunchKernelGGL((gemm_kernel<2, 2, AKM, BKM, GRP, EPI>), grid, block, 0, stream, args);        \
  } while (0)
#define SUAVE_GEMM_LAUNCH_RM(GRP, EPI) /* row-major operands: exact fp32 or split bf16 */                       \
  do {                                                                                                          \
    if (sp == 4) {                                                                                              \
      if (narrow) hipLaunchKernelGGL((gemm_kernel<4, 1, false, false, GRP, EPI, 4>), grid, block, 0, stream, args); \
      else hipLaunchKernelGGL((gemm_kernel<2, 2, false, false, GRP, EPI, 4>), grid, block, 0, stream, args);    \
    } else if (sp == 3) {                                                                                       \
      if (narrow) hipLaunchKernelGGL((gemm_kernel<4, 1, false, false, GRP, EPI, 3>), grid, block, 0, stream, args); \
      else hipLaunchKernelGGL((gemm_kernel<2, 2, false, false, GRP, EPI, 3>), grid, block, 0, stream, args);    \
    } else SUAVE_GEMM_LAUNCH(false, false, GRP, EPI);                                                           \
  } while (0)
  const int e = a.epi;
  const int sp = split_mode();
  if (a.grouped == 1 && !a.b_kmajor && e == kEpiBias) SUAVE_GEMM_LAUNCH_RM(1, kEpiBias);
  else if (a.grouped == 1 && a.b_kmajor && e == kEpiMaskBwd) SUAVE_GEMM_LAUNCH(false, true, 1, kEpiMaskBwd);
  else if (a.grouped == 1 && a.b_kmajor && e == kEpiBias) SUAVE_GEMM_LAUNCH(false, true, 1, kEpiBias);
  else if (a.grouped == 2) SUAVE_GEMM_LAUNCH(true, true, 2, kEpiSlab);
  else if (a.grouped != 0) SUAVE_REQUIRE(false, "gemm: unsupported grouped form");
  else if (!a.a_kmajor && !a.b_kmajor && e == kEpiBias) SUAVE_GEMM_LAUNCH_RM(0, kEpiBias);
  else if (!a.a_kmajor && !a.b_kmajor && e == kEpiReluDrop) SUAVE_GEMM_LAUNCH_RM(0, kEpiReluDrop);
  else if (!a.a_kmajor && !a.b_kmajor && e == kEpiMaskBwd) SUAVE_GEMM_LAUNCH_RM(0, kEpiMaskBwd);
  else if (!a.a_kmajor && a.b_kmajor && e == kEpiBias) SUAVE_GEMM_LAUNCH(false, true, 0, kEpiBias);
  else if (!a.a_kmajor && a.b_kmajor && e == kEpiMaskBwd) SUAVE_GEMM_LAUNCH(false, true, 0, kEpiMaskBwd);
  else if (a.a_kmajor && a.b_kmajor && e == kEpiSlab) {
    if (sp == 4 && !narrow) hipLaunchKernelGGL((gemm_kernel<2, 2, true, true, 0, kEpiSlab, 4>), grid, block, 0, stream, args);
    else if (sp == 3 && !narrow) hipLaunchKernelGGL((gemm_kernel<2, 2, true, true, 0, kEpiSlab, 3>), grid, block, 0, stream, args);
    else SUAVE_GEMM_LAUNCH(true, true, 0, kEpiSlab);
  } else SUAVE_REQUIRE(false, "gemm: unsupported operand layout / epilogue combination");
#undef SUAVE_GEMM_LAUNCH_RM
#undef SUAVE_GEMM_LAUNCH
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

// d: data-gradient GEMM (A k-contiguous, B k-major, kEpiMaskBwd or kEpiBias); w: weight-gradient GEMM of the
// same layer (both k-major, kEpiSlab).  Combinations without a fused instantiation run as two launches.
int launch_gemm_pair(const GemmArgs& d, const GemmArgs& w, int n_cu, hipStream_t stream) {
  const bool d_rm = !d.a_kmajor && !d.b_kmajor && d.grouped == 0;  // data gradient on a transposed (row-major) weight
  const bool fusable = d.M > 0 && d.N > 0 && w.M > 0 && w.N > 0 && !d.a_kmajor && (d.b_kmajor || d_rm) && w.a_kmajor && w.b_kmajor &&
                       w.epi == kEpiSlab && (d.epi == kEpiMaskBwd || d.epi == kEpiBias) &&
                       ((d.grouped == 0 && w.grouped == 0) || (d.grouped == 1 && w.grouped == 2 && d.epi == kEpiMaskBwd));
  static const bool no_pair = getenv("SUAVE_NO_PAIR") != nullptr;  // tuning knob: run the two GEMMs as two launches
  static const bool no_small_pair = getenv("SUAVE_NO_SMALL_BWD") != nullptr;  // tuning knob
  const bool small_pair = !no_small_pair && small_rowmajor_ok(d) && small_wgrad_ok(w);  // small batch: two split-K launches
  if (!no_small_pair && small_grouped_dgrad_ok(d) && small_grouped_wgrad_ok(w)) {
    if (int rc = check_gemm(d)) return rc;
    if (int rc = check_gemm(w)) return rc;
    ProfScope prof_scope(kProfGemm, stream);
    const int w_tiles = ((w.M + 31) / 32) * ((w.N + 63) / 64), w_blocks = w_tiles * w.splits;
    const int d_blocks = ((d.M + 31) / 32) * ((d.N + 63) / 64);
    hipLaunchKernelGGL((gemm_small_pair_kernel<kEpiMaskBwd, 1>), dim3(w_blocks + d_blocks), dim3(256), 0, stream, d, w,
                       kGroupRows, w_tiles, w_blocks);
    SUAVE_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (small_pair && d.M > 0 && d.N > 0 && w.M > 0 && w.N > 0) {
    if (int rc = check_gemm(d)) return rc;
    if (int rc = check_gemm(w)) return rc;
    ProfScope prof_scope(kProfGemm, stream);
    const int per = ((w.K + w.splits - 1) / w.splits + BK - 1) / BK * BK;
    const int w_tiles = ((w.M + 31) / 32) * ((w.N + 63) / 64), w_blocks = w_tiles * w.splits;
    const int d_blocks = ((d.M + 31) / 32) * ((d.N + 63) / 64);
    const dim3 sgrid(w_blocks + d_blocks), sblock(256);
    if (d.epi == kEpiMaskBwd) hipLaunchKernelGGL(gemm_small_pair_kernel<kEpiMaskBwd>, sgrid, sblock, 0, stream, d, w, per, w_tiles, w_blocks);
    else hipLaunchKernelGGL(gemm_small_pair_kernel<kEpiBias>, sgrid, sblock, 0, stream, d, w, per, w_tiles, w_blocks);
    SUAVE_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (!fusable || w.bias_out != nullptr || (no_pair && d.grouped == 0)) {  // (bias_out: only launch_gemm's small-batch kernel)
    if (int rc = launch_gemm(w, n_cu, stream)) return rc;
    return launch_gemm(d, n_cu, stream);
  }
  if (int rc = check_gemm(d)) return rc;
  if (int rc = check_gemm(w)) return rc;
  GemmArgs da = d, wa = w;
  da.splits = 1;
  const int w_tiles = gemm_tiles(w), w_blocks = w_tiles * w.splits, d_tiles = gemm_tiles(d);
  const bool dn = gemm_narrow(d), wn = gemm_narrow(w);
  const int sp = d_rm ? split_mode() : 0;
  SUAVE_REQUIRE(!d_rm || sp != 0, "gemm: a row-major data-gradient operand needs the split kernels (SUAVE_GEMM_SPLIT)");
  ProfScope prof_scope(kProfGemm, stream);
  const dim3 grid(w_blocks + d_tiles), block(256);
#define SUAVE_PAIR(DM, DN, DG, DE, WM, WN, WG, DS) \
  hipLaunchKernelGGL((gemm_pair_kernel<DM, DN, DG, DE, WM, WN, WG, (WG == 0 ? DS : 0), (WG == 0 ? 0 : DS)>), grid, block, 0, stream, da, wa, \
                     w_blocks, w_tiles)
#define SUAVE_PAIR_SHAPES(DG, DE, WG, DS)                  \
  do {                                                     \
    if (!dn && !wn) SUAVE_PAIR(2, 2, DG, DE, 2, 2, WG, DS); \
    else if (!dn && wn) SUAVE_PAIR(2, 2, DG, DE, 4, 1, WG, DS); \
    else if (dn && !wn) SUAVE_PAIR(4, 1, DG, DE, 2, 2, WG, DS); \
    else SUAVE_PAIR(4, 1, DG, DE, 4, 1, WG, DS);            \
  } while (0)
  if (d.grouped == 1) {
    if (split_mode() == 4) SUAVE_PAIR_SHAPES(1, kEpiMaskBwd, 2, 4);
    else if (split_mode() == 3) SUAVE_PAIR_SHAPES(1, kEpiMaskBwd, 2, 3);
    else SUAVE_PAIR_SHAPES(1, kEpiMaskBwd, 2, 0);
  }
  else if (d.epi == kEpiMaskBwd) {
    if (sp == 4) SUAVE_PAIR_SHAPES(0, kEpiMaskBwd, 0, 4);
    else if (sp == 3) SUAVE_PAIR_SHAPES(0, kEpiMaskBwd, 0, 3);
    else SUAVE_PAIR_SHAPES(0, kEpiMaskBwd, 0, 0);
  } else {
    if (sp == 4) SUAVE_PAIR_SHAPES(0, kEpiBias, 0, 4);
    else if (sp == 3) SUAVE_PAIR_SHAPES(0, kEpiBias, 0, 3);
    else SUAVE_PAIR_SHAPES(0, kEpiBias, 0, 0);
  }
#undef SUAVE_PAIR_SHAPES
#undef SUAVE_PAIR
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave

// Debugging / tuning aid (not part of include/suave_hip.h): one plain GEMM launch on device pointers.
extern "C" int suave_debug_gemm(const float* A, const float* B, float* C, const float* bias, const float* H, int M, int N,
                                int K, int lda, int ldb, int ldc, int a_kmajor, int b_kmajor, int epi, int splits,
                                float drop_p, void* stream) {
  suave::drop_stale_hip_error();
  suave::GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.H = H;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor; g.epi = epi; g.splits = splits > 0 ? splits : 1;
  g.slab_stride = (int64_t)M * ldc;
  g.ones_col = -1;
  g.drop_p = drop_p;
  g.seed = 1234; g.step = 1;
  return suave::launch_gemm(g, 256, static_cast<hipStream_t>(stream));
}
