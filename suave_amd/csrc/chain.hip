// Fused MLP chains: a whole stack of Linear (+ReLU +Dropout | x activation mask) layers in ONE launch, with the
// hidden activations handed from layer to layer in registers.
//
// Replaces, for layer widths up to 256, the per-layer launches of gemm.hip on
//   * the encoder trunk  Linear->ReLU->Dropout x n  (+ the component_logits rows of the head)
//                                                         suave/modules/encoder.py:46-58, :60-66
//   * the decoder trunk  Linear->ReLU->Dropout x n -> y_projection
//                                                         suave/modules/decoder.py:286-300, :343-344
//   * the decoder's data-gradient chain dY -> d h_n -> ... -> d z of their autograd backward.
//
// Shape of the computation (the GEMM-class review of round 1: every layer round-tripped its activations through
// HBM 2-3 times and both operands were cut into bf16 planes again by every tile that touched them):
//   * A wavefront owns 32 batch rows for the WHOLE chain.  A layer is computed transposed, C^T[feature, row] =
//     W[feature, k] * A^T[k, row] on v_mfma_f32_32x32x16_bf16, so a lane ends up holding, for its own row, four
//     consecutive features per accumulator register group -- which is exactly the B-operand fragment of the next
//     layer (k = this layer's feature) up to a fixed permutation of the 16 k of a tile.  That permutation is baked
//     into the weight images, so the next layer's operand is built from the lane's own accumulators: bias, ReLU,
//     dropout and the three-plane bf16 split are done once per activation element, in registers, between the
//     MFMAs of the next layer.  No activation goes through LDS, waves never exchange data.
//   * Arithmetic is the three-plane split of gemm.hip (fp32 = hi + mid + lo in bf16, the six plane products with
//     i + j <= 2, fp32 accumulate): fp32-grade results.  The weights are cut into planes ONCE per call by
//     k_weight_images (0.9 MB for the 64-column model) instead of by every row tile.
//   * The weight image of a k pair (32 k) is the same for the four waves of a block: it streams from L2 into a
//     two-stage LDS ring by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), one barrier per k pair;
//     fragments are single ds_read_b128 (same 32-byte row + half swap swizzle as gemm.hip's row-major images).
//   * Hidden activations are written to HBM once (fp32, for the backward's weight gradients) or not at all
//     (generation), and never read back by the forward.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace suave {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kMaxF = kChainMaxWidth / 32;       // feature tiles of a layer
constexpr int kStageBytes = 2 * kMaxF * 3 * 1024;  // one k PAIR (two k tiles) of the widest layer: 48 KB

__device__ __forceinline__ uint32_t hash32c(uint32_t x) {  // == gemm.hip hash32: the two paths drop the same cells
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// fp32 -> three bf16 planes, 8 elements: x = hi + mid + lo up to 2^-24 |x| (residuals exact in fp32)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&p)[3]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 h = (__bf16)r[e];
      p[s][e] = h;
      r[e] -= (float)h;
    }
  }
}

// Weight image of one layer: for k tile kt, feature tile F, plane s a 1 KB block [32 features][32 bytes]; the
// 16 bf16 of a row are the tile's k in the fragment order of the accumulator layout
//   element i of half h  <->  k = 16 kt + 8 (i >> 2) + 4 h + (i & 3),
// and the two halves of a row swap places in every second group of 8 rows (bank swizzle, see gemm.hip).  The
// image holds an even number of k tiles (zeros past K): the kernel streams it in pairs.
struct WimgArgs {
  const float* W[kChainMaxLayers];
  const float* bias[kChainMaxLayers];   // [N] or null
  float* biasp[kChainMaxLayers];        // [32 * ftiles]: the bias, zeros past N
  unsigned char* img[kChainMaxLayers];
  int N[kChainMaxLayers], K[kChainMaxLayers], ld[kChainMaxLayers], transposed[kChainMaxLayers];
  int ktiles[kChainMaxLayers], ftiles[kChainMaxLayers];
};

__global__ __launch_bounds__(256) void k_weight_images(WimgArgs a) {
  const int m = blockIdx.y;
  const int N = a.N[m], K = a.K[m], ld = a.ld[m], ftiles = a.ftiles[m];
  const int chunks = a.ktiles[m] * ftiles * 64;  // (kt, F, row, half)
  const float* W = a.W[m];
  if (blockIdx.x == 0)
    for (int f = threadIdx.x; f < 32 * ftiles; f += 256) a.biasp[m][f] = (a.bias[m] != nullptr && f < N) ? a.bias[m][f] : 0.f;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < chunks; c += gridDim.x * 256) {
    const int h = c & 1, r = (c >> 1) & 31, t = c >> 6;
    const int F = t % ftiles, kt = t / ftiles;
    const int n = 32 * F + r;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * kt + 8 * (i >> 2) + 4 * h + (i & 3);
      float x = 0.f;
      if (n < N && k < K) x = a.transposed[m] ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k];
      v[i] = x;
    }
    bf16x8 p[3];
    split8(v, p);
    unsigned char* base = a.img[m] + (size_t)(kt * ftiles + F) * 3072 + r * 32 + ((h ^ ((r >> 3) & 1)) << 4);
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<bf16x8*>(base + s * 1024) = p[s];
  }
}

// Everything a wave carries through the chain.
struct ChainState {
  const unsigned char* wsrc;     // where the weight loader reads next (null: the stream is exhausted)
  int w_blocks;                  // 1 KB blocks of the stage it reads (a k pair of the layer it is in)
  int w_left;                    // stages left in that layer
  int w_layer;                   // the loader's layer
  int stage;                     // LDS stage the next pair is multiplied from
};

// Start the LDS-DMA of the loader's next stage (a k pair of its layer) into `dst` and advance the loader.  The
// image is copied verbatim (its bank swizzle is baked in): each wave-instruction moves one 1 KB block, lane i
// bytes 16 i .. 16 i + 15 -- global_load_lds_dwordx4, no VGPR and no ds_write on the way.
__device__ __forceinline__ void ring_fill(const ChainArgs& a, ChainState& st, unsigned char* dst, int wave, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  if (st.wsrc == nullptr) return;  // (uniform)
  for (int b = wave; b < st.w_blocks; b += 4)
    __builtin_amdgcn_global_load_lds(static_cast<const void*>(st.wsrc + (size_t)b * 1024 + lane * 16),
                                     (lds_ptr)(dst + b * 1024), 16, 0, 0);
  st.wsrc += (size_t)st.w_blocks * 1024;
  if (--st.w_left == 0) {  // next layer's image (uniform)
    if (++st.w_layer < a.n_layers) {
      const ChainLayer& L = a.L[st.w_layer];
      st.wsrc = L.wimg;
      st.w_blocks = 2 * L.ftiles * 3;
      st.w_left = L.ktiles / 2;
    } else {
      st.wsrc = nullptr;
    }
  }
}

// One layer with NF feature tiles of output: K loop over k pairs + epilogue.  The operand of layer 0 comes from
// the input rows in HBM, that of the others from `prev`, and the layer's activated outputs are left in `prev`.
// The k-pair loop is unrolled with uniform guards so that prev[] is only ever indexed by constants (a dynamic
// index would move the whole array to scratch memory).
template <int NF>
__device__ __forceinline__ void chain_layer(const ChainArgs& a, const int l, ChainState& st, f32x16 (&prev)[kMaxF],
                                            unsigned char (*ring)[kStageBytes], const int tid, const int row,
                                            const int rowc, const bool row_ok, const uint32_t rid) {
  const ChainLayer& L = a.L[l];
  const int lane = tid & 63;
  // The lane coordinates are laundered per layer: everything derived from them (input addresses of every k pair,
  // the per-feature-group dropout constants, ...) is loop-invariant across the layer loop, and hoisted out of it
  // by the compiler those few hundred values no longer fit the register file (they were spilled to scratch).
  int l31 = lane & 31, kh = lane >> 5, rowv = rowc;
  asm volatile("" : "+v"(l31), "+v"(kh), "+v"(rowv));
  const bool first = l == 0;
  f32x16 cur[NF];
#pragma unroll
  for (int F = 0; F < NF; ++F)
#pragma unroll
    for (int r = 0; r < 16; ++r) cur[F][r] = 0.f;

  const float* xrow = a.in + (size_t)rowv * a.ld_in;
  float4 xin[4];
  auto load_input = [&](int kp) {  // the lane's 16 inputs of k pair kp, in fragment order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = 32 * kp + 8 * q + 4 * kh;
      xin[q] = *reinterpret_cast<const float4*>(k < a.ld_in ? xrow + k : xrow);
    }
  };
  if (first) load_input(0);

  const int npairs = L.ktiles / 2;
  const int lds_lane = l31 * 32 + ((kh ^ ((l31 >> 3) & 1)) << 4);
#pragma unroll
  for (int kp = 0; kp < kMaxF; ++kp) {
    if (kp < npairs) {  // (uniform)
      // the next stage streams into the other LDS buffer (every wave left it at the previous barrier) while this
      // one is multiplied; the barrier at the end of the step waits for it
      ring_fill(a, st, &ring[st.stage ^ 1][0], tid >> 6, lane);
      // ---- the 16 activations of this k pair, this lane's row ----
      float t[16];
      if (first) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 32 * kp + 8 * q + 4 * kh;
          t[4 * q + 0] = k + 0 < L.K ? xin[q].x : 0.f;  // pad columns of the input may hold anything
          t[4 * q + 1] = k + 1 < L.K ? xin[q].y : 0.f;
          t[4 * q + 2] = k + 2 < L.K ? xin[q].z : 0.f;
          t[4 * q + 3] = k + 3 < L.K ? xin[q].w : 0.f;
        }
        if (kp + 1 < npairs) load_input(kp + 1);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = prev[kp][i];
      }
      const unsigned char* wbase = &ring[st.stage][0] + lds_lane;
      // Fragments are fetched one feature tile ahead and the schedule is pinned tile by tile: left alone the
      // compiler hoists all 6 NF fragment reads of the pair (up to 192 VGPRs) and spills.  The split of the
      // second k tile's activations sits in the first tile's region, under its MFMAs.
      bf16x8 fb[2][3];
      {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = t[i];
        split8(v, fb[0]);
      }
      bf16x8 fa[2][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) fa[0][s] = *reinterpret_cast<const bf16x8*>(wbase + s * 1024);
#pragma unroll
      for (int idx = 0; idx < 2 * NF; ++idx) {  // idx = half * NF + F: the order of the blocks in the stage
        const int half = idx / NF, F = idx % NF;
        if (idx + 1 < 2 * NF) {
#pragma unroll
          for (int s = 0; s < 3; ++s)
            fa[(idx + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(wbase + ((idx + 1) * 3 + s) * 1024);
        }
        if (idx == 0) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = t[8 + i];
          split8(v, fb[1]);
        }
        const bf16x8(&A)[3] = fa[idx & 1];
        const bf16x8(&B)[3] = fb[half];
        // plane products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi (the order of gemm.hip)
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], cur[F], 0, 0, 0);
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], cur[F], 0, 0, 0);
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], cur[F], 0, 0, 0);
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], cur[F], 0, 0, 0);
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], cur[F], 0, 0, 0);
        cur[F] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], cur[F], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      st.stage ^= 1;
    }
  }

  // ---- epilogue, in registers: lane = row, register group g of tile F = features 32F + 8g + 4kh .. +3 ----
  // Every load below is unconditional on a clamped address (no per-element branches): the padded bias holds
  // zeros past N, and values of features past N are zeroed at the end.
  const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const uint32_t thr16 = a.drop_p > 0.f ? min(65535u, static_cast<uint32_t>(a.drop_p * 65536.0f + 0.5f)) : 0u;
  const uint32_t key = hash32c(static_cast<uint32_t>(a.seed) ^ hash32c(static_cast<uint32_t>(a.seed >> 32) + L.stream_id)) ^
                       hash32c(static_cast<uint32_t>(a.step) * 0x9E3779B1u + 0x632BE5ABu);
  const uint32_t hrow = hash32c(rid * 0x9E3779B1u + key);
  const int n_store = (L.N + 3) & ~3;  // columns the output row owns (ldo >= n_store)
  const bool relu = L.epi == kChainReluDrop;
  const bool hashed = relu && a.drop_p > 0.f && L.keep == nullptr;
  const bool injected = relu && a.drop_p > 0.f && L.keep != nullptr;
  const bool masked = L.epi == kChainMask;
  // (the schedule is pinned per feature tile, with the next tile's bias in flight: left alone the compiler hoists
  // every load of the unrolled epilogue to its top -- 128+ VGPRs -- and spills, which a lone wave per SIMD cannot hide)
  float4 bnext[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bnext[g] = *reinterpret_cast<const float4*>(L.biasp + 8 * g + 4 * kh);
#pragma unroll
  for (int F = 0; F < NF; ++F) {
    float4 bcur[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bcur[g] = bnext[g];
    if (F + 1 < NF) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bnext[g] = *reinterpret_cast<const float4*>(L.biasp + 32 * (F + 1) + 8 * g + 4 * kh);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f0 = 32 * F + 8 * g + 4 * kh;
      const float4 b4 = bcur[g];
      float v[4] = {cur[F][4 * g + 0] + b4.x, cur[F][4 * g + 1] + b4.y, cur[F][4 * g + 2] + b4.z, cur[F][4 * g + 3] + b4.w};
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (hashed) {
        const uint32_t h0 = hash32c(hrow ^ (static_cast<uint32_t>(f0 >> 1) * 0x85EBCA77u));
        const uint32_t h1 = hash32c(hrow ^ (static_cast<uint32_t>((f0 >> 1) + 1) * 0x85EBCA77u));
        v[0] = (h0 & 0xFFFFu) >= thr16 ? v[0] * inv_keep : 0.f;
        v[1] = (h0 >> 16) >= thr16 ? v[1] * inv_keep : 0.f;
        v[2] = (h1 & 0xFFFFu) >= thr16 ? v[2] * inv_keep : 0.f;
        v[3] = (h1 >> 16) >= thr16 ? v[3] * inv_keep : 0.f;
      }
      if (injected) {  // parity runs: keep mask [rows, N] bytes
        const uint8_t* kr = L.keep + (size_t)rowv * L.N;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = kr[min(f0 + j, L.N - 1)] != 0 ? v[j] * inv_keep : 0.f;
      }
      if (masked) {  // backward through ReLU + dropout: the forward activation tells both
        const float4 hv = *reinterpret_cast<const float4*>(L.mask_src + (size_t)rowv * L.ld_mask + (f0 < n_store ? f0 : 0));
        v[0] = hv.x > 0.f ? v[0] * inv_keep : 0.f;
        v[1] = hv.y > 0.f ? v[1] * inv_keep : 0.f;
        v[2] = hv.z > 0.f ? v[2] * inv_keep : 0.f;
        v[3] = hv.w > 0.f ? v[3] * inv_keep : 0.f;
      }
      // (features past N are exactly 0 here without a guard: zero weight rows, zero padded bias, relu / dropout /
      // mask of 0 is 0 -- so the next layer's pad columns are finite)
#pragma unroll
      for (int j = 0; j < 4; ++j) prev[F][4 * g + j] = v[j];
      if (L.out != nullptr && row_ok && f0 < n_store)
        *reinterpret_cast<float4*>(L.out + (size_t)row * L.ldo + f0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256, 1) void k_chain(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[2][kStageBytes];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = blockIdx.x * kChainRows + wave * 32 + (lane & 31);
  const bool row_ok = row < a.rows;
  const int rowc = row_ok ? row : a.rows - 1;  // loads of rows past the end read the last row; nothing is stored
  uint32_t rid;  // dropout key of the row: its table row (x components + component for virtual rows), see GemmArgs
  if (a.row_vdiv > 1) {
    const int drow = rowc / a.row_vdiv;
    const uint32_t trow = a.row_ids ? static_cast<uint32_t>(a.row_ids[drow]) : static_cast<uint32_t>(a.row_begin + drow);
    rid = trow * static_cast<uint32_t>(a.row_vdiv) + static_cast<uint32_t>(rowc - drow * a.row_vdiv);
  } else {
    rid = a.row_ids ? static_cast<uint32_t>(a.row_ids[rowc]) : static_cast<uint32_t>(a.row_begin + rowc);
  }

  ChainState st;
  f32x16 prev[kMaxF];  // activated outputs of the previous layer (fp32), this lane's row
#pragma unroll
  for (int F = 0; F < kMaxF; ++F)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[F][r] = 0.f;
  st.w_layer = 0;
  st.wsrc = a.L[0].wimg;
  st.w_blocks = 2 * a.L[0].ftiles * 3;
  st.w_left = a.L[0].ktiles / 2;
  st.stage = 0;
  ring_fill(a, st, &ring[0][0], wave, lane);  // prologue: the first stage
  __syncthreads();

  for (int l = 0; l < a.n_layers; ++l) {
    switch (a.L[l].ftiles) {  // uniform
      case 1: chain_layer<1>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 2: chain_layer<2>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 3: chain_layer<3>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 4: chain_layer<4>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 5: chain_layer<5>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 6: chain_layer<6>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      case 7: chain_layer<7>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
      default: chain_layer<8>(a, l, st, prev, ring, tid, row, rowc, row_ok, rid); break;
    }
  }
}

}  // namespace

bool chain_enabled() {
  // opt-in while the kernel is being tuned (SUAVE_CHAIN=1): correct on every parity test, not yet faster
  static const bool on = [] { const char* e = getenv("SUAVE_CHAIN"); return e && atoi(e) != 0; }();
  return on && gemm_split_mode() == 3;
}

bool chain_layer_ok(int n_out, int n_in) { return n_out >= 1 && n_in >= 1 && n_out <= kChainMaxWidth && n_in <= kChainMaxWidth; }

size_t chain_image_bytes(int n_out, int n_in) {  // an even number of k tiles + the padded bias
  const size_t ftiles = (n_out + 31) / 32;
  return (size_t)(((n_in + 31) / 32) * 2) * ftiles * 3072 + ftiles * 32 * sizeof(float);
}

int launch_chain(ChainArgs& a, const ChainWeights* w, unsigned char* images, hipStream_t stream) {
  if (a.rows <= 0 || a.n_layers <= 0) return 0;
  SUAVE_REQUIRE(a.n_layers <= kChainMaxLayers, "chain: too many layers");
  SUAVE_REQUIRE(a.ld_in % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0, "chain: input rows must be 16-byte aligned");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(images) & 15) == 0, "chain: weight images must be 16-byte aligned");
  WimgArgs wa{};
  size_t off = 0;
  int max_chunks = 1;
  int k_in = a.in_width;
  for (int l = 0; l < a.n_layers; ++l) {
    ChainLayer& L = a.L[l];
    SUAVE_REQUIRE(chain_layer_ok(w[l].n_out, w[l].n_in), "chain: layer wider than the register-chained kernel supports");
    SUAVE_REQUIRE(w[l].n_in == k_in, "chain: layer widths do not chain");
    L.N = w[l].n_out; L.K = w[l].n_in;
    L.ktiles = ((L.K + 31) / 32) * 2; L.ftiles = (L.N + 31) / 32;
    SUAVE_REQUIRE(L.out == nullptr || (L.ldo % 4 == 0 && L.ldo >= ((L.N + 3) & ~3) && (reinterpret_cast<uintptr_t>(L.out) & 15) == 0),
                  "chain: output rows must be 16-byte aligned and hold N rounded up to 4 columns");
    SUAVE_REQUIRE(L.epi != kChainMask || (L.mask_src != nullptr && L.ld_mask % 4 == 0 && L.ld_mask >= ((L.N + 3) & ~3) &&
                                          (reinterpret_cast<uintptr_t>(L.mask_src) & 15) == 0),
                  "chain: activation mask rows must be 16-byte aligned");
    L.wimg = images + off;
    L.biasp = reinterpret_cast<const float*>(images + off + (size_t)L.ktiles * L.ftiles * 3072);
    wa.bias[l] = L.bias; wa.biasp[l] = const_cast<float*>(L.biasp);
    wa.W[l] = w[l].W; wa.img[l] = images + off; wa.N[l] = L.N; wa.K[l] = L.K; wa.ld[l] = w[l].ld;
    wa.transposed[l] = w[l].transposed; wa.ktiles[l] = L.ktiles; wa.ftiles[l] = L.ftiles;
    max_chunks = std::max(max_chunks, L.ktiles * L.ftiles * 64);
    off += chain_image_bytes(L.N, L.K);
    k_in = L.N;
  }
  {
    ProfScope prof_scope(kProfOther, stream);
    hipLaunchKernelGGL(k_weight_images, dim3((max_chunks + 255) / 256, a.n_layers), dim3(256), 0, stream, wa);
    SUAVE_CHECK_HIP(hipGetLastError());
  }
  ProfScope prof_scope(kProfGemm, stream);
  const int grid = (a.rows + kChainRows - 1) / kChainRows;
  hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, stream, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
