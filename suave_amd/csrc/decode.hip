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
//   * The weights are cut into plane images ONCE per call (k_decode_images: 240 KB for the 64-column model) in the
//     exact order the block consumes them; they stream from L2 through a two-stage LDS ring by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no ds_write), one barrier per stage; a fragment is one ds_read_b128.
//   * A finished y tile goes into a 64-feature LDS window [128 rows][2 tiles]; the columns whose y slice ENDS in that
//     tile are then drawn with lane = row: waves 2g, 2g + 1 share rows 64 g .. 64 g + 63 and take those columns in turn
//     (sample_col.h: the same two-stage decisions as k_sample_heads, bit-identical outcomes).  A narrow column's slice
//     is at most 32 wide, so it lies inside the window when its last tile lands.
// Two blocks per CU (256 registers a lane, <= 80 KB LDS): while one block's waves are in their MFMA phase the other
// block's waves sample, so the matrix pipe works in the shadow of the VALU-bound sampling.
#include <algorithm>
#include <cstring>

#include "sample_col.h"

namespace suave {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kDecRows = 128;                 // rows per block: 4 waves x 32
constexpr int kDecMaxKT1 = 4;                 // latent_dim <= 64
constexpr int kDecMaxNF1 = 8;                 // first hidden width <= 256
constexpr int kDecMaxNF2 = 4;                 // second hidden width <= 128
constexpr int kDecMaxNF3 = 16;                // y rows <= 512
constexpr int kDecWin = 64, kDecWinLd = 65;   // y window: two 32-feature tiles, odd row stride (lane = row reads)
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

struct DecodeFusedArgs {
  SampleArgs s;
  const float* z;  // [n, D]
  int D, z_vec4;
  int KT1, NF1, NF2, NF3;
  const unsigned char* img;
  const float* bias[3];  // padded to whole tiles, zeros past N
  int stageA_bytes, stageB_bytes, ring_stride;
  float* y_out;  // [n, ld_yout] or null
  int ld_yout, y_width;
  int16_t tile_col[kDecMaxNF3 + 1];  // columns [tile_col[t], tile_col[t + 1]) end in y tile t
};

struct DecImgArgs {
  const float* W[3];
  const float* b[3];
  int N[3], K[3], ld[3];
  int KT1, NF1, NF2, NF3;
  unsigned char* img;
  float* bias[3];
  int stageA_bytes, stageB_bytes;
  int total_chunks;
};

// Stage images.  A unit is 1 KB: [32 features][2 halves x 16 B], one fp16 plane of one (feature tile, k step); the two
// 16-byte halves of a row swap places in every second group of 8 rows (bank swizzle of gemm.hip's row-major images).
// Element i of half h is k = 16 kt + 8 h + i for the first layer (its operand comes from z in that order) and
// k = 16 kt + 8 (i >> 2) + 4 h + (i & 3) for the chained layers (the accumulator layout of the producing tile).
//   stage A_F (F < NF1):  KT1 pairs (hi, lo) of layer 1, tile F; then for s = 0, 1 and F2 < NF2 the pair of layer 2,
//                         k step 2 F + s, tile F2
//   stage B_F3 (F3 < NF3): 2 NF2 pairs of layer 3, k step kt, tile F3
__global__ __launch_bounds__(256) void k_decode_images(DecImgArgs a) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  for (int l = 0; l < 3; ++l) {
    const int nt = l == 0 ? a.NF1 : l == 1 ? a.NF2 : a.NF3;
    for (int f = tid; f < 32 * nt; f += gridDim.x * 256) a.bias[l][f] = (a.b[l] != nullptr && f < a.N[l]) ? a.b[l][f] : 0.f;
  }
  for (int c = tid; c < a.total_chunks; c += gridDim.x * 256) {
    const int off = c * 16;
    const int a_total = a.NF1 * a.stageA_bytes;
    int t, rem;
    if (off < a_total) { t = off / a.stageA_bytes; rem = off - t * a.stageA_bytes; }
    else { t = (off - a_total) / a.stageB_bytes; rem = (off - a_total) - t * a.stageB_bytes; t += a.NF1; }
    const int u = rem >> 10, o = rem & 1023;
    const int r = o >> 5, h = ((o >> 4) & 1) ^ ((r >> 3) & 1);
    int layer, kt, tile, plane;
    if (t < a.NF1) {
      if (u < 2 * a.KT1) { layer = 0; kt = u >> 1; tile = t; plane = u & 1; }
      else {
        const int q = u - 2 * a.KT1, pair = q >> 1;
        layer = 1; plane = q & 1; kt = 2 * t + pair / a.NF2; tile = pair % a.NF2;
      }
    } else { layer = 2; kt = u >> 1; tile = t - a.NF1; plane = u & 1; }
    const int n = 32 * tile + r;
    f16x8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = layer == 0 ? 16 * kt + 8 * h + i : 16 * kt + 8 * (i >> 2) + 4 * h + (i & 3);
      float x = 0.f;
      if (n < a.N[layer] && k < a.K[layer]) x = a.W[layer][(size_t)n * a.ld[layer] + k];
      const _Float16 hi = (_Float16)x;
      out[i] = plane == 0 ? hi : (_Float16)((x - (float)hi) * kLoScale);
    }
    *reinterpret_cast<f16x8*>(a.img + off) = out;
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

__global__ __launch_bounds__(256, 2) void k_decode_fused(DecodeFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dec_lds[];
  unsigned char* ring = dec_lds;                                                 // [2][ring_stride]
  float* ywin = reinterpret_cast<float*>(dec_lds + 2 * (size_t)a.ring_stride);   // [128][kDecWinLd]
  float* w_s = ywin + kDecRows * kDecWinLd;                                      // [head_len]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int lane_off = l31 * 32 + ((kh ^ ((l31 >> 3) & 1)) << 4);
  const SampleArgs& sa = a.s;
  const Philox rng(sa.seed);
  const int KT1 = a.KT1, NF1 = a.NF1, NF2 = a.NF2, NF3 = a.NF3;
  const int n_stage = NF1 + NF3;
  const int64_t n_tiles = (sa.n + kDecRows - 1) / kDecRows;
  const int64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  int64_t fills_left = my_tiles * n_stage;  // the ring is only filled for stages this block will consume
  int fill_t = 0;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto fill = [&](unsigned char* dst) {  // start the LDS-DMA of the next stage (block-uniform)
    if (fills_left <= 0) return;
    --fills_left;
    const unsigned char* src;
    int units;
    if (fill_t < NF1) { src = a.img + (size_t)fill_t * a.stageA_bytes; units = a.stageA_bytes >> 10; }
    else { src = a.img + (size_t)NF1 * a.stageA_bytes + (size_t)(fill_t - NF1) * a.stageB_bytes; units = a.stageB_bytes >> 10; }
    for (int b = wave; b < units; b += 4)
      __builtin_amdgcn_global_load_lds(static_cast<const void*>(src + (size_t)b * 1024 + lane * 16), (lds_ptr)(dst + b * 1024), 16, 0, 0);
    if (++fill_t == n_stage) fill_t = 0;
  };
  if (sa.head_len > 0)
    for (int i = tid; i < sa.head_len; i += 256) w_s[i] = sa.params[sa.head_off + i];
  int buf = 0;
  fill(ring);  // prologue: the first stage

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kDecRows;
    // ---- this lane's row of z, as the first layer's B fragments (k = 16 kt + 8 kh + i) ----
    f16x8 zh[kDecMaxKT1], zl[kDecMaxKT1];
    {
      const int64_t zr = min(r0 + 32 * wave + l31, sa.n - 1);
      const float* zrow = a.z + (size_t)zr * a.D;
#pragma unroll
      for (int kt = 0; kt < kDecMaxKT1; ++kt) {
        float v[8];
        if (kt < KT1) {
          const int k0 = 16 * kt + 8 * kh;
          if (a.z_vec4) {
            const float4 q0 = k0 < a.D ? *reinterpret_cast<const float4*>(zrow + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 q1 = k0 + 4 < a.D ? *reinterpret_cast<const float4*>(zrow + k0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = k0 + i < a.D ? zrow[k0 + i] : 0.f;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const _Float16 h = (_Float16)v[i];
          zh[kt][i] = h;
          zl[kt][i] = (_Float16)((v[i] - (float)h) * kLoScale);
        }
      }
    }
    f32x16 acc2[kDecMaxNF2], acc2x[kDecMaxNF2];
#pragma unroll
    for (int F2 = 0; F2 < kDecMaxNF2; ++F2)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc2[F2][r] = 0.f; acc2x[F2][r] = 0.f; }

    // ---- layers 1 + 2: one first-layer tile per stage, consumed at once as two k steps of the second layer ----
    for (int F = 0; F < NF1; ++F) {
      __syncthreads();  // stage `buf` has landed (vmcnt(0) before the barrier); every wave has left the other buffer
      fill(ring + (buf ^ 1) * (size_t)a.ring_stride);
      const unsigned char* st = ring + buf * (size_t)a.ring_stride + lane_off;
      float4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const float4*>(a.bias[0] + 32 * F + 8 * g + 4 * kh);
      f32x16 c, cx;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c[r] = 0.f; cx[r] = 0.f; }
#pragma unroll
      for (int kt = 0; kt < kDecMaxKT1; ++kt) {
        if (kt < KT1) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * kt) * 1024);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (2 * kt + 1) * 1024);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zh[kt], c, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, zl[kt], cx, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, zh[kt], cx, 0, 0, 0);
        }
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
        for (int F2 = 0; F2 < kDecMaxNF2; ++F2) {
          if (F2 < NF2) {
            const int pair = s * NF2 + F2;
            const f16x8 wh = *reinterpret_cast<const f16x8*>(st2 + (2 * pair) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(st2 + (2 * pair + 1) * 1024);
            acc2[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[s], acc2[F2], 0, 0, 0);
            acc2x[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[s], acc2x[F2], 0, 0, 0);
            acc2x[F2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[s], acc2x[F2], 0, 0, 0);
          }
        }
      buf ^= 1;
    }

    // ---- second layer's epilogue: h2 as the fp16 planes of 2 NF2 k steps ----
    f16x8 h2h[2 * kDecMaxNF2], h2l[2 * kDecMaxNF2];
#pragma unroll
    for (int F2 = 0; F2 < kDecMaxNF2; ++F2) {
      float v[16];
      if (F2 < NF2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias[1] + 32 * F2 + 8 * g + 4 * kh);
          v[4 * g + 0] = fmaxf(acc2[F2][4 * g + 0] + acc2x[F2][4 * g + 0] * kLoInv + b.x, 0.f);
          v[4 * g + 1] = fmaxf(acc2[F2][4 * g + 1] + acc2x[F2][4 * g + 1] * kLoInv + b.y, 0.f);
          v[4 * g + 2] = fmaxf(acc2[F2][4 * g + 2] + acc2x[F2][4 * g + 2] * kLoInv + b.z, 0.f);
          v[4 * g + 3] = fmaxf(acc2[F2][4 * g + 3] + acc2x[F2][4 * g + 3] * kLoInv + b.w, 0.f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
      }
      f16x8 ph[2], pl[2];
      planes16(v, ph, pl);
      h2h[2 * F2] = ph[0]; h2h[2 * F2 + 1] = ph[1];
      h2l[2 * F2] = pl[0]; h2l[2 * F2 + 1] = pl[1];
    }

    // ---- rows of the sampling phase: waves 2 g, 2 g + 1 share rows 64 g .. 64 g + 63, lane = row ----
    const int rgrp = wave >> 1;
    const int64_t rs = r0 + 64 * rgrp + lane;
    const bool valid = rs < sa.n;
    const int64_t r = valid ? rs : sa.n - 1;
    const uint64_t gid = (uint64_t)(sa.row_offset + r);
    const uint32_t rid = (uint32_t)gid, rid_hi = (uint32_t)(gid >> 32);
    const int kc = sa.comp[r];
    const float* yrow = ywin + (64 * rgrp + lane) * kDecWinLd;

    // ---- y_projection, one 32-feature tile per stage, and the columns that end in it ----
    for (int F3 = 0; F3 < NF3; ++F3) {
      __syncthreads();  // stage landed; the previous tile's columns are drawn (their window half may be overwritten)
      fill(ring + (buf ^ 1) * (size_t)a.ring_stride);
      const unsigned char* st = ring + buf * (size_t)a.ring_stride + lane_off;
      float4 b4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const float4*>(a.bias[2] + 32 * F3 + 8 * g + 4 * kh);
      f32x16 c, cx;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) { c[rr] = 0.f; cx[rr] = 0.f; }
#pragma unroll
      for (int kt = 0; kt < 2 * kDecMaxNF2; ++kt) {
        if (kt < 2 * NF2) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(st + (2 * kt) * 1024);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(st + (2 * kt + 1) * 1024);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, h2h[kt], c, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, h2l[kt], cx, 0, 0, 0);
          cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, h2h[kt], cx, 0, 0, 0);
        }
      }
      {
        float* wrow = ywin + (32 * wave + l31) * kDecWinLd + ((32 * F3) & (kDecWin - 1)) + 4 * kh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wrow[8 * g + 0] = c[4 * g + 0] + cx[4 * g + 0] * kLoInv + b4[g].x;
          wrow[8 * g + 1] = c[4 * g + 1] + cx[4 * g + 1] * kLoInv + b4[g].y;
          wrow[8 * g + 2] = c[4 * g + 2] + cx[4 * g + 2] * kLoInv + b4[g].z;
          wrow[8 * g + 3] = c[4 * g + 3] + cx[4 * g + 3] * kLoInv + b4[g].w;
        }
      }
      buf ^= 1;
      __syncthreads();  // the tile is in the window
      if (a.y_out != nullptr) {
        for (int idx = tid; idx < kDecRows * 32; idx += 256) {
          const int rl = idx >> 5, f = 32 * F3 + (idx & 31);
          if (r0 + rl < sa.n && f < a.y_width) a.y_out[(size_t)(r0 + rl) * a.ld_yout + f] = ywin[rl * kDecWinLd + (f & (kDecWin - 1))];
        }
      }
      const int c_end = a.tile_col[F3 + 1];
      for (int cidx = a.tile_col[F3] + (wave & 1); cidx < c_end; cidx += 2) {
        const suave_col_t col = sa.cols[cidx];
        const int y_off = col.y_off;
        const float* w = w_s + (col.w_off - sa.head_off);
        const float* w2 = w_s + (max(col.w2_off, sa.head_off) - sa.head_off);
        float val = 0.f;
        int code = 0;
        sample_cell(sa, col, cidx, [&](int i) { return yrow[(y_off + i) & (kDecWin - 1)]; }, w, w2, nullptr, kc, r, rid, rid_hi, rng,
                    val, code);
        if (valid) {
          if (col.type <= SUAVE_TYPE_COUNT) sa.out_num[(size_t)r * sa.ld_num + col.x_index] = val;
          else sa.out_codes[(size_t)r * sa.ld_disc + col.x_index] = (int16_t)code;
        }
      }
    }
  }
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------

bool decode_fused_ok(const suave_plan* p) {
  const suave_model_t& m = p->model;
  if (gemm_split_mode() != 4 || m.n_hidden != 2 || !p->wide_host.empty()) return false;
  if (m.latent_dim > 16 * kDecMaxKT1 || m.dec[0].n_out > 32 * kDecMaxNF1 || m.dec[1].n_out > 32 * kDecMaxNF2) return false;
  if (m.y_rows > 32 * kDecMaxNF3 || m.n_cols > 32000) return false;
  if (m.head_len <= 0) return false;
  for (const suave_col_t& c : p->cols_host) {
    if (c.y_dim > 32 || c.y_dim < 1) return false;
    if (c.w_off < m.head_off || c.w_off >= m.head_off + m.head_len) return false;
  }
  // y slices in schema order (a column is drawn when the tile that holds its last feature lands)
  for (size_t i = 1; i < p->cols_host.size(); ++i)
    if (p->cols_host[i].y_off < p->cols_host[i - 1].y_off + p->cols_host[i - 1].y_dim) return false;
  const size_t KT1 = (m.latent_dim + 15) / 16, NF2 = (m.dec[1].n_out + 31) / 32;
  const size_t stage = std::max((KT1 + 2 * NF2) * 2048, 2 * NF2 * 2048);
  const size_t lds = 2 * stage + sizeof(float) * ((size_t)kDecRows * kDecWinLd + m.head_len);
  return lds <= 80 * 1024;
}

size_t decode_fused_image_bytes(const suave_plan* p) {
  const suave_model_t& m = p->model;
  const size_t KT1 = (m.latent_dim + 15) / 16, NF1 = (m.dec[0].n_out + 31) / 32, NF2 = (m.dec[1].n_out + 31) / 32,
               NF3 = (m.y_rows + 31) / 32;
  const size_t img = NF1 * (KT1 + 2 * NF2) * 2048 + NF3 * (2 * NF2) * 2048;
  return img + sizeof(float) * 32 * (NF1 + NF2 + NF3) + 256;
}

int launch_decode_fused(const suave_plan* p, const float* params, const float* z, const int32_t* comp, int64_t n,
                        const float* nn, const float* nu, const float* ne, uint64_t seed, int64_t row_offset,
                        unsigned char* images, float* out_num, int ld_num, int16_t* out_codes, int ld_disc, float* y_out,
                        hipStream_t st) {
  if (n <= 0) return 0;
  const suave_model_t& m = p->model;
  SUAVE_REQUIRE(decode_fused_ok(p), "decode_fused: model outside the fused kernel's shapes");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(images) & 255) == 0, "decode_fused: weight images must be 256-byte aligned");
  const int KT1 = (m.latent_dim + 15) / 16, NF1 = (m.dec[0].n_out + 31) / 32, NF2 = (m.dec[1].n_out + 31) / 32,
            NF3 = (m.y_rows + 31) / 32;
  DecImgArgs ia{};
  const suave_layer_t* L[3] = {&m.dec[0], &m.dec[1], &m.y_proj};
  for (int l = 0; l < 3; ++l) {
    ia.W[l] = params + L[l]->w_off;
    ia.b[l] = params + L[l]->b_off;
    ia.N[l] = L[l]->n_out; ia.K[l] = L[l]->n_in; ia.ld[l] = L[l]->ld;
  }
  ia.KT1 = KT1; ia.NF1 = NF1; ia.NF2 = NF2; ia.NF3 = NF3;
  ia.stageA_bytes = (KT1 + 2 * NF2) * 2048; ia.stageB_bytes = 2 * NF2 * 2048;
  const size_t img_bytes = (size_t)NF1 * ia.stageA_bytes + (size_t)NF3 * ia.stageB_bytes;
  ia.img = images;
  float* bias0 = reinterpret_cast<float*>(images + ((img_bytes + 255) & ~size_t(255)));
  ia.bias[0] = bias0; ia.bias[1] = bias0 + 32 * NF1; ia.bias[2] = bias0 + 32 * (NF1 + NF2);
  ia.total_chunks = (int)(img_bytes / 16);
  {
    ProfScope prof_scope(kProfOther, st);
    hipLaunchKernelGGL(k_decode_images, dim3(std::min((ia.total_chunks + 255) / 256, 1024)), dim3(256), 0, st, ia);
    SUAVE_CHECK_HIP(hipGetLastError());
  }
  DecodeFusedArgs a{};
  SampleArgs& s = a.s;
  s.cols = p->cols_dev; s.cls_off = p->cls_off_dev; s.params = params; s.y = nullptr; s.s = nullptr; s.comp = comp; s.n = n;
  s.n_cols = m.n_cols; s.n_num = m.n_num; s.n_disc = m.n_disc; s.K = m.n_components; s.ldY = p->ldY; s.Y = m.y_rows;
  s.total_classes = p->total_classes;
  s.noise_normal = nn; s.noise_uniform = nu; s.noise_exp = ne; s.seed = seed; s.row_offset = row_offset;
  s.out_num = out_num; s.ld_num = ld_num; s.out_codes = out_codes; s.ld_disc = ld_disc;
  s.head_off = m.head_off; s.head_len = m.head_len;
  a.z = z; a.D = m.latent_dim;
  a.z_vec4 = (m.latent_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0) ? 1 : 0;
  a.KT1 = KT1; a.NF1 = NF1; a.NF2 = NF2; a.NF3 = NF3;
  a.img = images;
  for (int l = 0; l < 3; ++l) a.bias[l] = ia.bias[l];
  a.stageA_bytes = ia.stageA_bytes; a.stageB_bytes = ia.stageB_bytes;
  a.ring_stride = std::max(ia.stageA_bytes, ia.stageB_bytes);
  a.y_out = y_out; a.ld_yout = m.y_rows; a.y_width = m.y_rows;
  {  // columns by the y tile their slice ends in
    int c = 0;
    for (int t = 0; t < NF3; ++t) {
      a.tile_col[t] = (int16_t)c;
      while (c < m.n_cols && (p->cols_host[c].y_off + p->cols_host[c].y_dim - 1) / 32 <= t) ++c;
    }
    a.tile_col[NF3] = (int16_t)c;
    SUAVE_REQUIRE(c == m.n_cols, "decode_fused: a column's y slice lies outside y_projection's rows");
  }
  const size_t lds = 2 * (size_t)a.ring_stride + sizeof(float) * ((size_t)kDecRows * kDecWinLd + m.head_len);
  static thread_local size_t lds_set = 0;
  if (lds > 48 * 1024 && lds > lds_set) {
    SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  const int64_t tiles = (n + kDecRows - 1) / kDecRows;
  const int grid = (int)std::min<int64_t>(tiles, 2 * (int64_t)p->n_cu);
  ProfScope prof_scope(kProfSample, st);
  hipLaunchKernelGGL(k_decode_fused, dim3(grid), dim3(256), lds, st, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
