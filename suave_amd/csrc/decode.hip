// Fused generative decode: decoder trunk -> y_projection -> per-column sample draw in ONE launch.
//
// Replaces, for SUAVE.sample / SUAVE.impute (hard assignment) / the Monte-Carlo posterior predictive, the chain
//   Decoder.backbone (Linear -> ReLU [-> Dropout, identity in eval]) x 2, Decoder.y_projection
//                                                   suave/modules/decoder.py:286-300, :343-344
//   the column loop of Decoder.forward with the head samplers      decoder.py:348-379, distributions.py:67-174
//   sampling.decoder_outputs_to_frame's value conversion            suave/sampling.py:138-179
// which the per-layer path (api.hip decode_sample_impl) runs as three GEMM launches + k_sample_heads with the hidden
// activations and y_full written to HBM and read back (1.35 GB per 262 144-row chunk of the 64-column model).  Here
// nothing but z comes in and nothing but the sampled cells goes out (about 0.3 KB a row).
//
// Shape of the kernel
//   * A block is four wavefronts on 128 rows; a WAVE owns 32 rows for the whole MLP.  Every layer is computed
//     transposed, C^T[feature, row] = W[feature, k] * A^T[k, row] on v_mfma_f32_32x32x16_f16, so that a lane ends up
//     with, for ITS row, four consecutive features per accumulator register group -- which is the B-operand fragment of
//     the next layer up to a fixed permutation of the 16 k of a step; the permutation is baked into the weight images.
//     Hidden activations therefore never leave the registers and waves never exchange them.
//   * Arithmetic is gemm.hip's mode 4: every fp32 value x = hi + lo' * 2^-11 (two fp16 planes), three plane products,
//     hi*hi into the main fp32 accumulators and the two cross products into a second set that is added with weight
//     2^-11 -- fp32-grade results (same scheme, same accuracy as the per-layer kernels).
//   * The first hidden layer is produced one 32-feature tile at a time and each tile is consumed at once as two k
//     steps of the second layer, whose accumulators (<= 128 features) stay live: h1 is never held whole.  The second
//     layer's output (<= 128 features) becomes 8 k steps of fp16 planes in registers, and y_projection is produced one
//     32-feature tile at a time from them.
//   * The weights are cut into plane images ONCE per call (k_decode_images) in the exact order the block consumes
//     them.  Those of the two hidden layers stream from L2 through a two-stage LDS ring by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no ds_write), one barrier per stage, a fragment is one ds_read_b128; those of
//     the output tiles are loaded straight into registers (the four waves of a block read the same 16 KB between two
//     barriers: L1 hits), one tile ahead, under the sampling of the tile before.
//   * THE HEADS' LINEAR MAPS ARE ONE MORE MFMA LAYER.  Every head starts with dot products of its y slice with a few
//     weight rows (mean_layer, logits_layer rows, the ordinal mean; decoder.py:64-67,113-116,201-206,244-247).  With
//     W_h those rows as a block-sparse matrix over y, u = W_h y = (W_h W_y) h2 + W_h b_y: k_decode_images multiplies the
//     rows through y_projection once per call, and the kernel produces `u` tiles exactly like y tiles.  Per cell the
//     VALU is left with the component's weight (a gather), the random draws and the transcendental functions; the
//     per-component constants (sampling standard deviations, softplus of the ordinal partition weights) are tables
//     written once per block over the staged head weights.
//   * Finished y and u tiles go into two 64-feature LDS windows [128 rows][2 tiles] (the u window lies in the ring's
//     memory, idle by then).  The host merges the two tile sequences -- always the kind that covers fewer columns so
//     far -- and after every tile the columns that are now covered by both are drawn with lane = row: waves 2g, 2g + 1
//     share rows 64 g .. 64 g + 63 and take those columns in turn (sample_col.h: the two-stage decisions of
//     k_sample_heads -- a race that is too close is run again from the y slice and the raw weights with detmath -- so
//     outcomes are bit-identical to the per-layer path's).  A narrow column's slices are at most 32 wide, so they lie
//     inside the windows when the column is drawn, and the tile they start in is only overwritten after that.
// Two blocks per CU (256 registers a lane, <= 80 KB LDS): while one block's waves are in their MFMA phase the other
// block's waves sample, so the matrix pipe works in the shadow of the VALU-bound sampling.
#include <algorithm>
#include <cstdio>
#include <cstring>

#include "sample_col.h"

#ifndef DEC_ABL
#define DEC_ABL 0
#endif

#ifndef DEC_TIMING
#define DEC_TIMING 0
#endif
#if DEC_TIMING
#define DEC_T(i) do { const long long t_now = clock64(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define DEC_T(i) do { } while (0)
#endif

namespace suave {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kDecRows = 128;                 // rows per block: 4 waves x 32
constexpr int kDecMaxKT1 = 4;                 // latent_dim <= 64
constexpr int kDecMaxNF1 = 8;                 // first hidden width <= 256
constexpr int kDecMaxNF2 = 4;                 // second hidden width <= 128
constexpr int kDecMaxNF3 = 16;                // y rows <= 512, head rows (u) <= 512
constexpr int kDecMaxSteps = 2 * kDecMaxNF3;  // output tiles of a row tile: y tiles + u tiles
constexpr int kDecMaxCols = 512;              // schema columns (the schedule travels in the kernel arguments)
constexpr int kDecWin = 64, kDecWinLd = 65;   // y window: two 32-feature tiles, odd row stride (lane = row reads)
#ifndef DEC_PREFETCH_ALL
#define DEC_PREFETCH_ALL 1
#endif
constexpr bool kPrefetchAll = DEC_PREFETCH_ALL != 0;  // both halves of the next output tile's fragments under the sampling
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

struct DecodeFusedArgs {
  SampleArgs s;
  const float* z;  // [n, D]
  int D, z_vec4;
  int KT1, NF1, NF2, n_steps;
  const unsigned char* img;   // ring stages A_0 .. A_{NF1-1}
  const unsigned char* imgB;  // output tiles in step order, 2 NF2 pairs each
  const float* bias[3];       // layer 1, layer 2 (whole tiles, zeros past N), output tiles in step order [n_steps][32]
  int stageA_bytes, stageB_bytes, ring_bytes;  // ring_bytes: both ring buffers == the u window's room
  float* y_out;  // [n, ld_yout] or null
  int ld_yout, y_width;
  const int32_t* u_off;                 // [n_cols] first u row of the column
  unsigned long long* dbg;              // DEC_TIMING builds: per-phase cycle sums of wave 0 of every block
  // (32-bit entries: a dynamically indexed 16-bit kernel argument is fetched with a VECTOR load, and the s_waitcnt
  // vmcnt(0) behind it also waits for every output store in flight -- once per column)
  int32_t step_tile[kDecMaxSteps];      // (kind << 8) | tile: kind 0 = y tile, 1 = u tile
  // after step t the two waves of a row group draw col_list[step_begin[t] .. step_mid[t]) and [step_mid[t] ..
  // step_begin[t + 1]): the columns that tile completes, dealt out by estimated cost
  int32_t step_begin[kDecMaxSteps + 1];
  int32_t step_mid[kDecMaxSteps];
  int32_t col_list[kDecMaxCols];
};

struct DecImgArgs {
  const float* params;
  const float* W[3];
  const float* b[3];
  int N[3], K[3], ld[3];
  int KT1, NF1, NF2, n_steps, n_u;
  const int32_t* u_map;  // [n_u][3]: flat-parameter offset of the weight row, y_off, y_dim
  int32_t step_tile[kDecMaxSteps];
  unsigned char* img;
  unsigned char* imgB;
  float* bias[3];
  int stageA_bytes, stageB_bytes;
  int chunksA, chunksB;
};

// Stage images.  A unit is 1 KB: [32 features][2 halves x 16 B], one fp16 plane of one (feature tile, k step); the two
// 16-byte halves of a row swap places in every second group of 8 rows (bank swizzle of gemm.hip's row-major images).
// Element i of half h is k = 16 kt + 8 h + i for the first layer (its operand comes from z in that order) and
// k = 16 kt + 8 (i >> 2) + 4 h + (i & 3) for the chained layers (the accumulator layout of the producing tile).
//   ring stage A_F (F < NF1):  KT1 pairs (hi, lo) of layer 1, tile F; then for s = 0, 1 and F2 < NF2 the pair of
//                              layer 2, k step 2 F + s, tile F2
//   output step t:             2 NF2 pairs, k step kt, of y_projection's tile (kind 0) or of the head rows' tile (kind 1):
//                              row o of W_u = sum_i w_o[i] * W_y[y_off + i, :] over the row's y slice
__global__ __launch_bounds__(256) void k_decode_images(DecImgArgs a) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  for (int l = 0; l < 2; ++l) {
    const int nt = l == 0 ? a.NF1 : a.NF2;
    for (int f = tid; f < 32 * nt; f += gridDim.x * 256) a.bias[l][f] = (a.b[l] != nullptr && f < a.N[l]) ? a.b[l][f] : 0.f;
  }
  for (int f = tid; f < 32 * a.n_steps; f += gridDim.x * 256) {
    const int st = a.step_tile[f >> 5], n = 32 * (st & 255) + (f & 31);
    float v = 0.f;
    if ((st >> 8) == 0) {
      if (n < a.N[2]) v = a.b[2][n];
    } else if (n < a.n_u) {  // W_h b_y
      const int woff = a.u_map[3 * n], yo = a.u_map[3 * n + 1], yd = a.u_map[3 * n + 2];
      for (int i = 0; i < yd; ++i) v = fmaf(a.params[woff + i], a.b[2][yo + i], v);
    }
    a.bias[2][f] = v;
  }
  for (int c = tid; c < a.chunksA + a.chunksB; c += gridDim.x * 256) {
    const bool isB = c >= a.chunksA;
    const int off = (isB ? c - a.chunksA : c) * 16;
    const int sb = isB ? a.stageB_bytes : a.stageA_bytes;
    const int t = off / sb, rem = off - t * sb;
    const int u = rem >> 10, o = rem & 1023;
    const int r = o >> 5, h = ((o >> 4) & 1) ^ ((r >> 3) & 1);
    int layer, kt, tile, plane, kind = 0;
    if (!isB) {
      if (u < 2 * a.KT1) { layer = 0; kt = u >> 1; tile = t; plane = u & 1; }
      else {
        const int q = u - 2 * a.KT1, pair = q >> 1;
        layer = 1; plane = q & 1; kt = 2 * t + pair / a.NF2; tile = pair % a.NF2;
      }
    } else {
      layer = 2; kt = u >> 1; plane = u & 1;
      const int st = a.step_tile[t];
      kind = st >> 8; tile = st & 255;
    }
    const int n = 32 * tile + r;
    int woff = 0, yo = 0, yd = 0;
    if (kind == 1 && n < a.n_u) { woff = a.u_map[3 * n]; yo = a.u_map[3 * n + 1]; yd = a.u_map[3 * n + 2]; }
    f16x8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = layer == 0 ? 16 * kt + 8 * h + i : 16 * kt + 8 * (i >> 2) + 4 * h + (i & 3);
      float x = 0.f;
      if (k < a.K[layer]) {
        if (kind == 0) {
          if (n < a.N[layer]) x = a.W[layer][(size_t)n * a.ld[layer] + k];
        } else {
          for (int j = 0; j < yd; ++j) x = fmaf(a.params[woff + j], a.W[2][(size_t)(yo + j) * a.ld[2] + k], x);
        }
      }
      const _Float16 hi = (_Float16)x;
      out[i] = plane == 0 ? hi : (_Float16)((x - (float)hi) * kLoScale);
    }
    *reinterpret_cast<f16x8*>((isB ? a.imgB : a.img) + off) = out;
  }
}

// 16 fp32 values of a lane (register r = 4 g + j <-> feature 8 g + 4 kh + j of the tile) -> the two fp16 planes of the
// two k steps the tile is for the next layer: step s is registers 8 s .. 8 s + 7.
__device__ __forceinline__ void planes16(const float (&v)[16], f16x8 (&hi)[2], f16x8 (&lo)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = v[8 * s + i];
      const _Float16 h = (_Float16)x;
      hi[s][i] = h;
      lo[s][i] = (_Float16)((x - (float)h) * kLoScale);
    }
}

template <int KT1, int NF2, bool INJ>
__global__ __launch_bounds__(256, 2) void k_decode_fused(DecodeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dec_lds[];
  unsigned char* ring = dec_lds;                                      // [2][stageA_bytes] during layers 1 + 2 ...
  float* uwin = reinterpret_cast<float*>(dec_lds);                    // ... the u window [128][kDecWinLd] afterwards
  float* ywin = reinterpret_cast<float*>(dec_lds + a.ring_bytes);     // [128][kDecWinLd]
  float* w_s = ywin + kDecRows * kDecWinLd;                           // [head_len]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const SampleArgs& sa = a.s;
  const Philox rng(sa.seed);
  const int NF1 = a.NF1;
  const int64_t n_tiles = (sa.n + kDecRows - 1) / kDecRows;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int r0_dummy = sa.n < 0 ? 1 : 0;
  auto fill = [&](int F, unsigned char* dst) {  // start the LDS-DMA of ring stage F (block-uniform)
    const unsigned char* src = a.img + (size_t)F * a.stageA_bytes;
#if DEC_ABL == 3 || DEC_ABL == 4  // timing only: no ring fills (+ no sampling)
    const int units = r0_dummy;
#else
    const int units = a.stageA_bytes >> 10;
#endif
    for (int b = wave; b < units; b += 4)
      __builtin_amdgcn_global_load_lds(static_cast<const void*>(src + (size_t)b * 1024 + lane * 16), (lds_ptr)(dst + b * 1024), 16, 0, 0);
  };
  // ---- head weights, once per block; per-component tables written over them (sample_col.h FAST_U) ----
  for (int i = tid; i < sa.head_len; i += 256) w_s[i] = sa.params[sa.head_off + i];
  __syncthreads();
  for (int c = tid; c < sa.n_cols; c += 256) {
    const suave_col_t col = sa.cols[c];
    if (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) {  // sampling standard deviation per component
      float* t = w_s + (col.w2_off - sa.head_off);
      const float s2 = col.type == SUAVE_TYPE_POS ? col.scale * col.scale : 1.0f;
      for (int k = 0; k < sa.K; ++k) t[k] = sqrtf(fmaxf((softplus_f(t[k]) + kEps) * s2, kEps));
    } else if (col.type == SUAVE_TYPE_ORDINAL) {  // threshold increments per (level, component)
      float* t = w_s + (col.w_off - sa.head_off);
      for (int i = 0; i < (col.n_classes - 1) * sa.K; ++i) t[i] = softplus_f(t[i]) + kEps;
    }
  }
  // (the first barrier of the tile loop orders these writes before any read)

#if DEC_TIMING
  long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
#endif
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kDecRows;
    // The lane coordinates are laundered per row tile: everything derived from them (LDS fragment addresses, bias and
    // image offsets of every phase) is invariant across the tile loop, and hoisted out of it those few dozen values
    // stay live through every phase and push the weight fragments of the output tiles into scratch.
    int l31 = lane & 31, kh = lane >> 5;
    asm volatile("" : "+v"(l31), "+v"(kh));
    const int lane_off = l31 * 32 + ((kh ^ ((l31 >> 3) & 1)) << 4);
    __syncthreads();  // the previous tile's last columns are drawn: the u window becomes the ring again
    DEC_T(0);
    fill(0, ring);
    // ---- this lane's row of z, as the first layer's B fragments (k = 16 kt + 8 kh + i) ----
    f16x8 zh[KT1], zl[KT1];
    {
      const int64_t zr = min(r0 + 32 * wave + l31, sa.n - 1);
      const float* zrow = a.z + (size_t)zr * a.D;
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt) {
        float v[8];
        const int k0 = 16 * kt + 8 * kh;
        if (a.z_vec4) {  // (uniform) D % 4 == 0: a quad is inside the row or outside
          const float4 q0 = *reinterpret_cast<const float4*>(zrow + min(k0, a.D - 4));
          const float4 q1 = *reinterpret_cast<const float4*>(zrow + min(k0 + 4, a.D - 4));
          v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = zrow[min(k0 + i, a.D - 1)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = k0 + i < a.D ? v[i] : 0.f;
          const _Float16 h = (_Float16)x;
          zh[kt][i] = h;
          zl[kt][i] = (_Float16)((x - (float)h) * kLoScale);
        }
      }
    }
    f32x16 acc2[NF2], acc2x[NF2];
#pragma unroll
    for (int F2 = 0; F2 < NF2; ++F2)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc2[F2][r] = 0.f; acc2x[F2][r] = 0.f; }

    float4 b4n[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b4n[g] = *reinterpret_cast<const float4*>(a.bias[0] + 8 * g + 4 * kh);
    DEC_T(1);
    // ---- layers 1 + 2: one first-layer tile per stage, consumed at once as two k steps of the second layer ----
    for (int F = 0; F < NF1; ++F) {
      __syncthreads();  // stage F has landed (vmcnt(0) before the barrier); every wave has left the other buffer
      DEC_T(2);
      if (F + 1 < NF1) fill(F + 1, ring + ((F + 1) & 1) * (size_t)a.stageA_bytes);
      const unsigned char* st = ring + (F & 1) * (size_t)a.stageA_bytes + lane_off;
      float4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] = b4n[g];
      if (F + 1 < NF1) {  // the next tile's bias: in flight under this iteration's MFMAs
#pragma unroll
        for (int g = 0; g < 4; ++g) b4n[g] = *reinterpret_cast<const float4*>(a.bias[0] + 32 * (F + 1) + 8 * g + 4 * kh);
      }
      f32x16 c, cx;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c[r] = 0.f; cx[r] = 0.f; }
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt) {
        const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * kt) * 1024);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (2 * kt + 1) * 1024);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zh[kt], c, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zl[kt], cx, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, zh[kt], cx, 0, 0, 0);
      }
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[4 * g + 0] = fmaxf(c[4 * g + 0] + cx[4 * g + 0] * kLoInv + b4[g].x, 0.f);
        v[4 * g + 1] = fmaxf(c[4 * g + 1] + cx[4 * g + 1] * kLoInv + b4[g].y, 0.f);
        v[4 * g + 2] = fmaxf(c[4 * g + 2] + cx[4 * g + 2] * kLoInv + b4[g].z, 0.f);
        v[4 * g + 3] = fmaxf(c[4 * g + 3] + cx[4 * g + 3] * kLoInv + b4[g].w, 0.f);
      }
      f16x8 bh[2], bl[2];
      planes16(v, bh, bl);
      const unsigned char* st2 = st + (size_t)(2 * KT1) * 1024;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int F2 = 0; F2 < NF2; ++F2) {
          const int pair = s * NF2 + F2;
          const f16x8 wh = *reinterpret_cast<const f16x8*>(st2 + (2 * pair) * 1024);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(st2 + (2 * pair + 1) * 1024);
          acc2[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[s], acc2[F2], 0, 0, 0);
          acc2x[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[s], acc2x[F2], 0, 0, 0);
          acc2x[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[s], acc2x[F2], 0, 0, 0);
        }
    }

    // ---- second layer's epilogue: h2 as the fp16 planes of 2 NF2 k steps ----
    f16x8 h2h[2 * NF2], h2l[2 * NF2];
    int kh2 = kh;  // (defined behind the loop above: the bias loads below are otherwise hoisted over it and spilled)
    asm volatile("" : "+v"(kh2) : "v"(acc2[0][0]));
#pragma unroll
    for (int F2 = 0; F2 < NF2; ++F2) {
      float v[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias[1] + 32 * F2 + 8 * g + 4 * kh2);
        v[4 * g + 0] = fmaxf(acc2[F2][4 * g + 0] + acc2x[F2][4 * g + 0] * kLoInv + b.x, 0.f);
        v[4 * g + 1] = fmaxf(acc2[F2][4 * g + 1] + acc2x[F2][4 * g + 1] * kLoInv + b.y, 0.f);
        v[4 * g + 2] = fmaxf(acc2[F2][4 * g + 2] + acc2x[F2][4 * g + 2] * kLoInv + b.z, 0.f);
        v[4 * g + 3] = fmaxf(acc2[F2][4 * g + 3] + acc2x[F2][4 * g + 3] * kLoInv + b.w, 0.f);
      }
      f16x8 ph[2], pl[2];
      planes16(v, ph, pl);
      h2h[2 * F2] = ph[0]; h2h[2 * F2 + 1] = ph[1];
      h2l[2 * F2] = pl[0]; h2l[2 * F2 + 1] = pl[1];
    }

    DEC_T(4);
    // ---- rows of the sampling phase: waves 2 g, 2 g + 1 share rows 64 g .. 64 g + 63, lane = row ----
    const int rgrp = wave >> 1;
    const int64_t rs = r0 + 64 * rgrp + lane;
    const bool valid = rs < sa.n;
    const int64_t r = valid ? rs : sa.n - 1;
    const uint64_t gid = (uint64_t)(sa.row_offset + r);
    const uint32_t rid = (uint32_t)gid, rid_hi = (uint32_t)(gid >> 32);
    int kc = sa.comp[r];
    asm volatile("" : "+v"(kc));  // (wait for it HERE: a first use inside the column loop puts an s_waitcnt vmcnt(0) -- which
                                  // also waits for the output stores in flight -- into every iteration)
    const float* yrow = ywin + (64 * rgrp + lane) * kDecWinLd;
    const float* urow = uwin + (64 * rgrp + lane) * kDecWinLd;
    NumDraws nd;

    // ---- output tiles (y_projection's and the head rows') in the host's merged order; their weight fragments come
    // straight from the image (L1 / L2), one tile ahead ----
    // The first half of a tile's fragments (k steps 0 .. NF2 - 1) is loaded one tile ahead, under the sampling of the
    // tile before; the second half at the start of the step, behind the first half's MFMAs.
    f16x8 wfh[2 * NF2], wfl[2 * NF2];
    // (the offset is defined HERE, behind the second layer: without it the loads of the first tile's fragments are
    // hoisted above the first two layers and their 32 registers spilled across them)
    uint32_t lane_off_g = (uint32_t)lane_off;
    asm volatile("" : "+v"(lane_off_g) : "v"(h2h[0]));
    auto load_frags = [&](int t, int half) {  // (uniform base + one 32-bit lane offset per access: the saddr form of global_load)
      const unsigned char* src = a.imgB + (size_t)t * a.stageB_bytes;
#pragma unroll
      for (int kt = 0; kt < 2 * NF2; ++kt) {
        if ((kt < NF2) == (half == 0)) {  // (compile-time after inlining: `half` is a literal at every call)
          wfh[kt] = *reinterpret_cast<const f16x8*>(src + (2 * kt) * 1024 + lane_off_g);
          wfl[kt] = *reinterpret_cast<const f16x8*>(src + (2 * kt + 1) * 1024 + lane_off_g);
        }
      }
    };
    load_frags(0, 0);
    if (!kPrefetchAll) { } else load_frags(0, 1);
    for (int t = 0; t < a.n_steps; ++t) {
      const int st = a.step_tile[t];
      const int kind = st >> 8, T = st & 255;
      if (!kPrefetchAll) load_frags(t, 1);
      float4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const float4*>(a.bias[2] + 32 * t + 8 * g + 4 * kh);
      f32x16 c, cx;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) { c[rr] = 0.f; cx[rr] = 0.f; }
#pragma unroll
      for (int kt = 0; kt < 2 * NF2; ++kt) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[kt], h2h[kt], c, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfh[kt], h2l[kt], cx, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfl[kt], h2h[kt], cx, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the next tile's fragments reuse these registers: keep the loads behind the MFMAs)
      if (t + 1 < a.n_steps) {  // in flight under this tile's columns
        load_frags(t + 1, 0);
        if (kPrefetchAll) load_frags(t + 1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      DEC_T(5);
      __syncthreads();  // the previous step's columns are drawn: its window half (and, at t == 0, the ring) is free
      DEC_T(6);
      {
        float* wrow = (kind == 0 ? ywin : uwin) + (32 * wave + l31) * kDecWinLd + ((32 * T) & (kDecWin - 1)) + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wrow[8 * g + 0] = c[4 * g + 0] + cx[4 * g + 0] * kLoInv + b4[g].x;
          wrow[8 * g + 1] = c[4 * g + 1] + cx[4 * g + 1] * kLoInv + b4[g].y;
          wrow[8 * g + 2] = c[4 * g + 2] + cx[4 * g + 2] * kLoInv + b4[g].z;
          wrow[8 * g + 3] = c[4 * g + 3] + cx[4 * g + 3] * kLoInv + b4[g].w;
        }
      }
      __syncthreads();  // the tile is in its window
      DEC_T(7);
      if (kind == 0 && a.y_out != nullptr) {
        for (int idx = tid; idx < kDecRows * 32; idx += 256) {
          const int rl = idx >> 5, f = 32 * T + (idx & 31);
          if (r0 + rl < sa.n && f < a.y_width) a.y_out[(size_t)(r0 + rl) * a.ld_yout + f] = ywin[rl * kDecWinLd + (f & (kDecWin - 1))];
        }
      }
#if DEC_ABL == 2 || DEC_ABL == 3  // timing only: no sampling
      const int li_end = tile < 0 ? a.step_mid[t] : 0;
      const int li_begin = 0;
#else
      const int li_begin = (wave & 1) ? a.step_mid[t] : a.step_begin[t];
      const int li_end = (wave & 1) ? a.step_begin[t + 1] : a.step_mid[t];
#endif
      for (int li = li_begin; li < li_end; ++li) {
        // The plan's tables are read through the constant address space: with the output stores in this loop the
        // compiler cannot prove a plain global load unclobbered and issues it as a VECTOR load -- an L2 round trip on
        // the critical path of every column (the descriptor decides everything that follows).
        const int cidx = __builtin_amdgcn_readfirstlane(a.col_list[li]);
        suave_col_t col;
        {
          typedef const __attribute__((address_space(4))) int32_t* cint_ptr;
          const cint_ptr cp = (cint_ptr)(uintptr_t)(sa.cols + cidx);
          int32_t raw[sizeof(suave_col_t) / 4];
#pragma unroll
          for (int q = 0; q < (int)(sizeof(suave_col_t) / 4); ++q) raw[q] = cp[q];
          __builtin_memcpy(&col, raw, sizeof(col));
        }
#if DEC_ABL >= 5 && DEC_ABL <= 8  // timing only: one column type skipped (5 real + pos, 6 count, 7 categorical, 8 ordinal)
        if ((DEC_ABL == 5 && col.type <= SUAVE_TYPE_POS) || (DEC_ABL == 6 && col.type == SUAVE_TYPE_COUNT) ||
            (DEC_ABL == 7 && col.type == SUAVE_TYPE_CAT) || (DEC_ABL == 8 && col.type == SUAVE_TYPE_ORDINAL))
          continue;
#endif
        const int y_off = col.y_off;
        const int u_off = ((const __attribute__((address_space(4))) int32_t*)(uintptr_t)a.u_off)[cidx];
        const float* w = w_s + (col.w_off - sa.head_off);
        const float* w2 = w_s + (max(col.w2_off, sa.head_off) - sa.head_off);
        float val = 0.f;
        int code = 0;
        sample_cell<true, INJ>(sa, col, cidx, [&](int i) { return yrow[(y_off + i) & (kDecWin - 1)]; },
                          [&](int j) { return urow[(u_off + j) & (kDecWin - 1)]; }, w, w2, sa.params + col.w_off, nullptr, kc, r,
                          rid, rid_hi, rng, nd, val, code);
#if DEC_ABL == 1  // timing only: no output stores
        if (valid && val == 12345.678f && code == 77) sa.out_num[(size_t)r * sa.ld_num + col.x_index] = val;
#else
        if (valid) {
          if (col.type <= SUAVE_TYPE_COUNT) sa.out_num[(size_t)r * sa.ld_num + col.x_index] = val;
          else sa.out_codes[(size_t)r * sa.ld_disc + col.x_index] = (int16_t)code;
        }
#endif
      }
      DEC_T(8);
    }
  }
#if DEC_TIMING
  if (a.dbg != nullptr && lane == 0)
    for (int i = 0; i < 10; ++i) atomicAdd(a.dbg + 10 * wave + i, (unsigned long long)t_acc[i]);
#endif
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------

// rows of the heads' linear maps over y (the `u` layer): real / pos mean_layer, cat logits_layer rows 1 .. C-1, the
// ordinal mean_layer; count heads draw from y itself
static int head_u_rows(const suave_col_t& c) {
  switch (c.type) {
    case SUAVE_TYPE_REAL: case SUAVE_TYPE_POS: case SUAVE_TYPE_ORDINAL: return 1;
    case SUAVE_TYPE_CAT: return c.n_classes - 1;
    default: return 0;
  }
}

int decode_plan_init(suave_plan* p) {  // suave_plan_create: the device tables of the u layer
  const suave_model_t& m = p->model;
  p->u_off_host.assign(m.n_cols, 0);
  std::vector<int32_t> map;
  int n_u = 0;
  for (int c = 0; c < m.n_cols; ++c) {
    const suave_col_t& col = p->cols_host[c];
    p->u_off_host[c] = n_u;
    const int rows = col_is_wide(col) ? 0 : head_u_rows(col);
    for (int j = 0; j < rows; ++j) {
      int woff = col.w_off;
      if (col.type == SUAVE_TYPE_CAT) woff += j * cat_ldw(col, m.n_components);
      if (col.type == SUAVE_TYPE_ORDINAL) woff = col.w2_off;
      map.push_back(woff); map.push_back(col.y_off); map.push_back(col.y_dim);
    }
    n_u += rows;
  }
  p->n_u = n_u;
  SUAVE_CHECK_HIP(hipMalloc(&p->u_off_dev, sizeof(int32_t) * std::max(m.n_cols, 1)));
  SUAVE_CHECK_HIP(hipMalloc(&p->u_map_dev, sizeof(int32_t) * std::max<size_t>(map.size(), 3)));
  SUAVE_CHECK_HIP(hipMemcpy(p->u_off_dev, p->u_off_host.data(), sizeof(int32_t) * m.n_cols, hipMemcpyHostToDevice));
  if (!map.empty()) SUAVE_CHECK_HIP(hipMemcpy(p->u_map_dev, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice));
  // ---- schedule of the output tiles (models the fused kernel takes; see decode_fused_ok) ----
  p->dec_step_tile.clear(); p->dec_step_begin.clear(); p->dec_step_mid.clear(); p->dec_col_list.clear();
  if (m.n_hidden != 2 || !p->wide_host.empty() || m.n_cols > kDecMaxCols || m.y_rows > 32 * kDecMaxNF3 || n_u > 32 * kDecMaxNF3) return 0;
  for (const suave_col_t& c : p->cols_host)
    if (c.y_dim > 32 || c.y_dim < 1 || head_u_rows(c) > 32) return 0;
  // The merged tile order: always the kind (y tiles / head-row tiles) that covers fewer columns so far.  Writing a
  // kind's tile t + 2 reuses the window half of its tile t; at that point this kind covers no more columns than the
  // other, so every column it covers is drawn, and the columns it does not cover end in tile >= t + 2, i.e. (slices
  // are at most 32 wide) do not touch tile t.
  const int nc = m.n_cols, NFy = (m.y_rows + 31) / 32, NFu = (n_u + 31) / 32;
  auto y_last = [&](int c) { return p->cols_host[c].y_off + p->cols_host[c].y_dim - 1; };
  auto u_last = [&](int c) { return p->u_off_host[c] + head_u_rows(p->cols_host[c]) - 1; };  // count: the row before it
  auto cost = [&](int c) {  // relative VALU cost of a cell (instruction counts of sample_cell's first stage)
    const suave_col_t& col = p->cols_host[c];
    switch (col.type) {
      case SUAVE_TYPE_REAL: case SUAVE_TYPE_POS: return 1.0;
      case SUAVE_TYPE_COUNT: return 1.2;
      case SUAVE_TYPE_CAT: return 0.5 + 0.3 * col.n_classes + 0.7 * ((col.n_classes + 3) / 4);
      default: return 0.6 + 0.45 * col.n_classes + 0.7 * ((col.n_classes + 3) / 4);
    }
  };
  int yf = 0, uf = 0, cy = 0, cu = 0, done = 0;
  auto advance = [&]() {
    while (cy < nc && y_last(cy) < 32 * yf) ++cy;
    while (cu < nc && u_last(cu) < 32 * uf) ++cu;
  };
  advance();
  p->dec_step_begin.push_back(0);
  while (yf < NFy || uf < NFu) {
    const bool take_y = uf >= NFu || (yf < NFy && cy <= cu);
    p->dec_step_tile.push_back((int16_t)(take_y ? yf : (256 | uf)));
    if (take_y) ++yf; else ++uf;
    advance();
    const int ready = std::max(done, std::min(cy, cu));
    // deal the newly covered columns [done, ready) to the two waves of a row group: units in descending cost to the
    // lighter wave; the two numeric columns of a Philox pair (xnum columns 2 j, 2 j + 1) stay together, back to back
    struct Unit { int c0, n; double cost; };
    std::vector<Unit> units;
    for (int c = done; c < ready; ++c) {
      const suave_col_t& col = p->cols_host[c];
      bool joined = false;
      if (col.type <= SUAVE_TYPE_COUNT && (col.x_index & 1))
        for (Unit& u : units)
          if (u.n == 1 && p->cols_host[u.c0].type <= SUAVE_TYPE_COUNT && p->cols_host[u.c0].x_index == col.x_index - 1) {
            u.n = -c - 2;  // second column of the pair (encoded: columns need not be adjacent in the schema)
            u.cost += cost(c) - 0.6;
            joined = true;
            break;
          }
      if (!joined) units.push_back(Unit{c, 1, cost(c)});
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& x, const Unit& y) { return x.cost > y.cost; });
    std::vector<int16_t> lst[2];
    double load[2] = {0.0, 0.0};
    for (const Unit& u : units) {
      const int w = load[1] < load[0] ? 1 : 0;
      lst[w].push_back((int16_t)u.c0);
      if (u.n < 0) lst[w].push_back((int16_t)(-u.n - 2));
      load[w] += u.cost;
    }
    for (int16_t c : lst[0]) p->dec_col_list.push_back(c);
    p->dec_step_mid.push_back((int16_t)p->dec_col_list.size());
    for (int16_t c : lst[1]) p->dec_col_list.push_back(c);
    p->dec_step_begin.push_back((int16_t)p->dec_col_list.size());
    done = ready;
  }
  if (done != nc) {  // (a column outside y_projection's rows: leave the model to the per-layer path)
    p->dec_step_tile.clear(); p->dec_col_list.clear();
  }
  return 0;
}

void decode_plan_free(suave_plan* p) {
  if (p->u_off_dev) (void)hipFree(p->u_off_dev);
  if (p->u_map_dev) (void)hipFree(p->u_map_dev);
  p->u_off_dev = nullptr; p->u_map_dev = nullptr;
}

namespace {
struct DecGeom {
  int KT1, NF1, NF2, NFy, NFu, n_steps;
  int stageA, stageB;
  size_t bytesA, bytesB, bias_floats, lds;
};
DecGeom dec_geom(const suave_plan* p) {
  const suave_model_t& m = p->model;
  DecGeom g;
  g.KT1 = (m.latent_dim + 15) / 16; g.NF1 = (m.dec[0].n_out + 31) / 32; g.NF2 = (m.dec[1].n_out + 31) / 32;
  g.NFy = (m.y_rows + 31) / 32; g.NFu = (p->n_u + 31) / 32; g.n_steps = g.NFy + g.NFu;
  g.stageA = (g.KT1 + 2 * g.NF2) * 2048; g.stageB = 2 * g.NF2 * 2048;
  g.bytesA = (size_t)g.NF1 * g.stageA; g.bytesB = (size_t)g.n_steps * g.stageB;
  g.bias_floats = 32 * (size_t)(g.NF1 + g.NF2 + g.n_steps);
  const size_t ring = std::max<size_t>(2 * (size_t)g.stageA, sizeof(float) * kDecRows * kDecWinLd);
  g.lds = ((ring + 15) & ~size_t(15)) + sizeof(float) * ((size_t)kDecRows * kDecWinLd + m.head_len);
  return g;
}
}  // namespace

bool decode_fused_ok(const suave_plan* p) {
  const suave_model_t& m = p->model;
  if (gemm_split_mode() != 4 || m.n_hidden != 2 || !p->wide_host.empty() || p->u_off_dev == nullptr) return false;
  if (m.latent_dim > 16 * kDecMaxKT1 || m.dec[0].n_out > 32 * kDecMaxNF1 || m.dec[1].n_out > 32 * kDecMaxNF2) return false;
  if (m.y_rows > 32 * kDecMaxNF3 || p->n_u > 32 * kDecMaxNF3 || m.n_cols > kDecMaxCols) return false;
  if (p->dec_step_tile.empty() || (int)p->dec_col_list.size() != m.n_cols) return false;
  if (m.head_len <= 0) return false;
  for (const suave_col_t& c : p->cols_host) {
    if (c.y_dim > 32 || c.y_dim < 1 || head_u_rows(c) > 32) return false;
    if (c.w_off < m.head_off || c.w_off >= m.head_off + m.head_len) return false;
  }
  // y slices in schema order (a column is drawn when the tiles that hold its last y feature and head row have landed)
  for (size_t i = 1; i < p->cols_host.size(); ++i)
    if (p->cols_host[i].y_off < p->cols_host[i - 1].y_off + p->cols_host[i - 1].y_dim) return false;
  return dec_geom(p).lds <= 80 * 1024;
}

size_t decode_fused_image_bytes(const suave_plan* p) {
  const DecGeom g = dec_geom(p);
  return ((g.bytesA + 255) & ~size_t(255)) + ((g.bytesB + 255) & ~size_t(255)) + sizeof(float) * g.bias_floats + 256;
}

int launch_decode_fused(const suave_plan* p, const float* params, const float* z, const int32_t* comp, int64_t n,
                        const float* nn, const float* nu, const float* ne, uint64_t seed, int64_t row_offset,
                        unsigned char* images, float* out_num, int ld_num, int16_t* out_codes, int ld_disc, float* y_out,
                        hipStream_t st) {
  if (n <= 0) return 0;
  const suave_model_t& m = p->model;
  SUAVE_REQUIRE(decode_fused_ok(p), "decode_fused: model outside the fused kernel's shapes");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(images) & 255) == 0, "decode_fused: weight images must be 256-byte aligned");
  const DecGeom g = dec_geom(p);
  DecodeFusedArgs a{};
  SUAVE_REQUIRE((int)p->dec_step_tile.size() == g.n_steps && (int)p->dec_col_list.size() == m.n_cols, "decode_fused: no schedule in the plan");
  for (int t = 0; t < g.n_steps; ++t) { a.step_tile[t] = p->dec_step_tile[t]; a.step_mid[t] = p->dec_step_mid[t]; }
  for (int t = 0; t <= g.n_steps; ++t) a.step_begin[t] = p->dec_step_begin[t];
  for (int c = 0; c < m.n_cols; ++c) a.col_list[c] = p->dec_col_list[c];
  DecImgArgs ia{};
  const suave_layer_t* L[3] = {&m.dec[0], &m.dec[1], &m.y_proj};
  ia.params = params;
  for (int l = 0; l < 3; ++l) {
    ia.W[l] = params + L[l]->w_off;
    ia.b[l] = params + L[l]->b_off;
    ia.N[l] = L[l]->n_out; ia.K[l] = L[l]->n_in; ia.ld[l] = L[l]->ld;
  }
  ia.KT1 = g.KT1; ia.NF1 = g.NF1; ia.NF2 = g.NF2; ia.n_steps = g.n_steps; ia.n_u = p->n_u; ia.u_map = p->u_map_dev;
  std::memcpy(ia.step_tile, a.step_tile, sizeof(ia.step_tile));
  ia.stageA_bytes = g.stageA; ia.stageB_bytes = g.stageB;
  ia.img = images;
  ia.imgB = images + ((g.bytesA + 255) & ~size_t(255));
  float* bias0 = reinterpret_cast<float*>(ia.imgB + ((g.bytesB + 255) & ~size_t(255)));
  ia.bias[0] = bias0; ia.bias[1] = bias0 + 32 * g.NF1; ia.bias[2] = bias0 + 32 * (g.NF1 + g.NF2);
  ia.chunksA = (int)(g.bytesA / 16); ia.chunksB = (int)(g.bytesB / 16);
  {
    ProfScope prof_scope(kProfOther, st);
    hipLaunchKernelGGL(k_decode_images, dim3(std::min((ia.chunksA + ia.chunksB + 255) / 256, 2048)), dim3(256), 0, st, ia);
    SUAVE_CHECK_HIP(hipGetLastError());
  }
  SampleArgs& s = a.s;
  s.cols = p->cols_dev; s.cls_off = p->cls_off_dev; s.params = params; s.y = nullptr; s.s = nullptr; s.comp = comp; s.n = n;
  s.n_cols = m.n_cols; s.n_num = m.n_num; s.n_disc = m.n_disc; s.K = m.n_components; s.ldY = p->ldY; s.Y = m.y_rows;
  s.total_classes = p->total_classes;
  s.noise_normal = nn; s.noise_uniform = nu; s.noise_exp = ne; s.seed = seed; s.row_offset = row_offset;
  s.out_num = out_num; s.ld_num = ld_num; s.out_codes = out_codes; s.ld_disc = ld_disc;
  s.head_off = m.head_off; s.head_len = m.head_len;
  a.z = z; a.D = m.latent_dim;
  a.z_vec4 = (m.latent_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0) ? 1 : 0;
  a.KT1 = g.KT1; a.NF1 = g.NF1; a.NF2 = g.NF2; a.n_steps = g.n_steps;
  a.img = ia.img; a.imgB = ia.imgB;
  for (int l = 0; l < 3; ++l) a.bias[l] = ia.bias[l];
  a.stageA_bytes = g.stageA; a.stageB_bytes = g.stageB;
  a.ring_bytes = (int)((std::max<size_t>(2 * (size_t)g.stageA, sizeof(float) * kDecRows * kDecWinLd) + 15) & ~size_t(15));
  a.y_out = y_out; a.ld_yout = m.y_rows; a.y_width = m.y_rows;
  a.u_off = p->u_off_dev;
#if DEC_TIMING
  {
    static unsigned long long* dbg = nullptr;
    if (dbg == nullptr) { (void)hipMalloc(&dbg, 40 * sizeof(unsigned long long)); (void)hipMemset(dbg, 0, 40 * sizeof(unsigned long long)); }
    a.dbg = dbg;
    static int calls = 0;
    if (++calls % 5 == 0) {
      unsigned long long h[40];
      (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
      (void)hipMemset(dbg, 0, sizeof(h));
      const char* names[10] = {"tile-top barrier", "z + setup", "F: barrier", "F: compute", "L2 epilogue", "step: mfma+prefetch", "step: barrier1", "step: write+barrier2", "step: sample", ""};
      for (int w = 0; w < 4; ++w) {
        unsigned long long tot = 0;
        for (int i = 0; i < 9; ++i) tot += h[10 * w + i];
        fprintf(stderr, "[dec timing] wave %d:", w);
        for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.1f%%;", names[i], 100.0 * h[10 * w + i] / (double)tot);
        fprintf(stderr, " total %.0f cycles/block-tile\n", (double)tot / (double)(((n + kDecRows - 1) / kDecRows) * 5));
      }
    }
  }
#endif
  typedef void (*kern_t)(DecodeFusedArgs);
#define DEC_ROW(I, K) {k_decode_fused<K, 1, I>, k_decode_fused<K, 2, I>, k_decode_fused<K, 3, I>, k_decode_fused<K, 4, I>}
  static const kern_t kerns[2][kDecMaxKT1][kDecMaxNF2] = {{DEC_ROW(false, 1), DEC_ROW(false, 2), DEC_ROW(false, 3), DEC_ROW(false, 4)},
                                                          {DEC_ROW(true, 1), DEC_ROW(true, 2), DEC_ROW(true, 3), DEC_ROW(true, 4)}};
#undef DEC_ROW
  const int inj = (nn != nullptr || nu != nullptr || ne != nullptr) ? 1 : 0;  // injected noise: the variant that reads it
  const kern_t kern = kerns[inj][g.KT1 - 1][g.NF2 - 1];
  static thread_local size_t lds_set[2][kDecMaxKT1][kDecMaxNF2] = {};
  if (g.lds > 48 * 1024 && g.lds > lds_set[inj][g.KT1 - 1][g.NF2 - 1]) {
    SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
    lds_set[inj][g.KT1 - 1][g.NF2 - 1] = g.lds;
  }
  const int64_t tiles = (n + kDecRows - 1) / kDecRows;
  const int grid = (int)std::min<int64_t>(tiles, 2 * (int64_t)p->n_cu);
  ProfScope prof_scope(kProfSample, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), g.lds, st, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
