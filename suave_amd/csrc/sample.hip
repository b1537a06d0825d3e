// Generative decode: latent draws (prior / conditional) and the fused per-column
// sample kernel.
//
// Replaces sampling.sample_prepared_mixture_latents (suave/sampling.py:38-56),
// SUAVE._draw_conditional_latents (suave/model.py:3042-3114, a Python per-row
// loop with two host syncs per row), the head samplers
// (suave/modules/distributions.py:67-80,112-117,130-134,167-174 as called from
// suave/modules/decoder.py:90-91,139,172,214-216,258) and the value conversion of
// sampling.decoder_outputs_to_frame (sampling.py:138-179).  The log-likelihood
// the reference computes on placeholder batches and throws away
// (sampling.py:79-100) is not computed at all.
//
// Every discrete decision (argmax / threshold) comes out as the individually
// rounded binary32 operations of detmath.h, in the order of
// oracle/sampling_oracle.py, would take it, so sampled categories / levels /
// counts / components are bit-identical to the oracle for identical inputs and
// noise.  Categorical and ordinal races are first run with the hardware
// transcendentals and fused dots; only a row whose winner leads by less than a
// bound on the difference of the two evaluations (kCatMargin / kOrdDp below) is
// decided again with detmath (about one race in 10^4).  Real and positive VALUES
// owe the reference 1e-4 and take the hardware path outright; counts stay on
// detmath (their inverse-CDF walk compares against every partial sum).
#include "sample_col.h"

namespace suave {

namespace {

__device__ __forceinline__ float philox_u(const Philox& rng, uint32_t a, uint32_t b, uint32_t stream, uint32_t d) {
  return u01(rng(a, b, stream, d).x);
}

// STAGED: the 64-row y tile and the outputs go through LDS (coalesced global accesses).  Schemas whose y row
// does not fit (wide categoricals, BASELINE config #4) run unstaged: lane = row reads y from global memory.
// A block is kSampleWaves wavefronts on one 64-row tile (lane = row, the waves take the columns in turn): eight
// rather than four, because the LDS tile admits two blocks per CU and two waves per SIMD left the VALU idle half
// of the time (54 % of the wave cycles were waits; PMC pass in profiles/).
constexpr int kSampleWaves = 8;
constexpr int kSampleThreads = kSampleWaves * 64;
template <bool STAGED>
__global__ __launch_bounds__(kSampleThreads, STAGED ? 2 : 1) void k_sample_heads(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ytile = lds;                                      // [64][ldy]
  float* onum = ytile + 64 * a.ldy;                        // [64][ldon]
  int* ocode = reinterpret_cast<int*>(onum + 64 * a.ldon); // [64][ldoc]
  float* s_s = reinterpret_cast<float*>(ocode + 64 * a.ldoc);  // [64][K] (dense s only)
  float* w_s = s_s + 64 * a.K;                                 // [head_len] the heads' weights
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K;
  const Philox rng(a.seed);
  const int64_t n_tiles = (a.n + 63) / 64;
  // Head weights: every dot below walks them with a wave-uniform index.  Out of global memory that is a dependent
  // load (hundreds of cycles) per multiply; the block keeps them in LDS for all its tiles instead.
  if (STAGED && a.head_len > 0)
    for (int i = tid; i < a.head_len; i += kSampleThreads) w_s[i] = a.params[a.head_off + i];
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * 64;
    const int nrows = (int)min((int64_t)64, a.n - r0);
    if (STAGED) {
      __syncthreads();
      const int nq = a.ldY / 4;
      for (int idx = tid; idx < nrows * nq; idx += kSampleThreads) {
        const int lr = idx / nq, q = idx % nq;
        const float4 v = *reinterpret_cast<const float4*>(a.y + (size_t)(r0 + lr) * a.ldY + q * 4);
        float* dst = ytile + lr * a.ldy + q * 4;
        if (q * 4 + 0 < a.Y) dst[0] = v.x;
        if (q * 4 + 1 < a.Y) dst[1] = v.y;
        if (q * 4 + 2 < a.Y) dst[2] = v.z;
        if (q * 4 + 3 < a.Y) dst[3] = v.w;
      }
      if (a.s)
        for (int idx = tid; idx < nrows * K; idx += kSampleThreads) s_s[idx] = a.s[(size_t)r0 * K + idx];
      __syncthreads();
    }
    const bool valid = lane < nrows;
    const int64_t r = r0 + (valid ? lane : 0);
    const uint32_t rid = (uint32_t)(a.row_offset + r), rid_hi = (uint32_t)((uint64_t)(a.row_offset + r) >> 32);
    const int kc = a.s ? 0 : a.comp[r];
    const float* yrow = STAGED ? ytile + lane * a.ldy : a.y + (size_t)r * a.ldY;
    const float* srow = STAGED ? s_s + lane * K : (a.s ? a.s + (size_t)r * K : nullptr);

    NumDraws nd;
    for (int c = wave; c < a.n_cols; c += kSampleWaves) {
      const suave_col_t col = a.cols[c];
      if (a.skip_wide && col_is_wide(col)) continue;
      const float* yb = yrow + col.y_off;
      // (wide categorical columns keep their weights behind the staged range: they are read from global memory)
      // (never form w_s - head_off: an LDS address below the window's start does not survive the cast to a flat pointer)
      const bool in_lds = STAGED && col.w_off >= a.head_off && col.w_off < a.head_off + a.head_len;
      const float* w = in_lds ? w_s + (col.w_off - a.head_off) : a.params + col.w_off;
      const float* w2 = in_lds ? w_s + (max(col.w2_off, a.head_off) - a.head_off) : a.params + max(col.w2_off, 0);
      float val = 0.f;
      int code = 0;
      sample_cell<false, true>(a, col, c, [&](int i) { return yb[i]; }, NoHeadU{}, w, w2, w, srow, kc, r, rid, rid_hi, rng, nd, val, code);
      if (col.type <= SUAVE_TYPE_COUNT) {
        if (STAGED) onum[lane * a.ldon + col.x_index] = val;
        else if (valid) a.out_num[(size_t)r * a.ld_num + col.x_index] = val;
      } else {
        if (STAGED) ocode[lane * a.ldoc + col.x_index] = code;
        else if (valid) a.out_codes[(size_t)r * a.ld_disc + col.x_index] = (int16_t)code;
      }
    }
    if (STAGED) {
      __syncthreads();
      for (int idx = tid; idx < nrows * a.n_num; idx += kSampleThreads) {
        const int lr = idx / a.n_num, j = idx % a.n_num;
        a.out_num[(size_t)(r0 + lr) * a.ld_num + j] = onum[lr * a.ldon + j];
      }
      for (int idx = tid; idx < nrows * a.n_disc; idx += kSampleThreads) {
        const int lr = idx / a.n_disc, j = idx % a.n_disc;
        a.out_codes[(size_t)(r0 + lr) * a.ld_disc + j] = (int16_t)ocode[lr * a.ldoc + j];
      }
    }
  }
}

// Wide categorical columns (SUAVE_WIDE_CAT_MIN_CLASSES .. 32 767 classes) of the decode: the race over a column's classes with the LOGITS
// ALREADY COMPUTED by the MFMA GEMM of the wide heads (gemm_wide_kernel, the training step's forward).  One wavefront
// per row walks the wide columns; lane q owns classes 4q .. 4q + 3 (+ 256 ...): the Philox block of those four
// classes, their scores L - log(e), then two wave reductions give the winner (first maximum, as the sequential scan)
// and the runner-up.  The GEMM's logits differ from the reference's individually rounded dot by summation order only
// (measured 4e-7 of the largest |logit| at 266 terms), so a lead above kWideMargin (1 + |score|) is decided; below it
// the wave evaluates the column again with detmath dots straight from [y ; s] and the weights -- the arithmetic of
// the narrow-column kernel and of oracle/sampling_oracle.py -- about one (row, column) in 10^4.
// (The lane = row kernel it replaces for these columns spent 2.5 M scalar multiply-adds per row of BASELINE config
// #4 on uncoalesced operands: 18 us a row.)
constexpr float kWideMargin = 5e-5f;
struct WideSampleArgs {
  const WideCol* wide;
  int n_wide;
  const suave_col_t* cols;
  const int32_t* cls_off;
  const float* params;
  const float* y;   // [n, ldY] slots [y ; s] filled (k_wide_fill)
  int ldY;
  const float* L;   // [n, ldL] logits of classes 1 .. C-1
  int ldL;
  const float* s;        // dense assignments or null
  const int32_t* comp;   // component per row (s == null)
  int K;
  const float* noise_exp;
  int total_classes;
  uint64_t seed;
  int64_t row_offset, n;
  int16_t* out_codes;
  int ld_disc;
};

__global__ __launch_bounds__(256) void k_wide_sample(WideSampleArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Philox rng(a.seed);
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < a.n; r += (int64_t)gridDim.x * 4) {
    const uint64_t gid = (uint64_t)(a.row_offset + r);
    const uint32_t rid = (uint32_t)gid, rid_hi = (uint32_t)(gid >> 32);
    const int kc = a.s ? 0 : a.comp[r];
    for (int c = 0; c < a.n_wide; ++c) {
      const WideCol wc = a.wide[c];
      const int C = wc.C;
      const int coff = a.cls_off[wc.col];
      const float* Lc = a.L + (size_t)r * a.ldL + wc.l_off;
      auto draw4 = [&](int c0, float (&e)[4]) {  // the exponential draws of classes c0 .. c0 + 3 (generator keys of k_sample_heads)
        uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
        if (!a.noise_exp) q4 = rng(rid, ((uint32_t)wc.col << 13) | (uint32_t)(c0 >> 2), kStreamSampleExp, rid_hi);
        const uint32_t word[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
          e[u] = a.noise_exp ? (c0 + u < C ? a.noise_exp[(size_t)r * a.total_classes + coff + c0 + u] : 1.0f) : exp_draw(word[u]);
      };
      auto wave_argmax = [&](float& b, int& i) {  // first maximum wins
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float ob = __shfl_xor(b, off, 64);
          const int oi = __shfl_xor(i, off, 64);
          if (ob > b || (ob == b && oi < i)) { b = ob; i = oi; }
        }
      };
      float best = -INFINITY, second = -INFINITY;
      int bi = 0x7FFFFFFF;
      for (int c0 = 4 * lane; c0 < C; c0 += 256) {
        float e[4];
        draw4(c0, e);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cl = c0 + u;
          if (cl < C) {
            const float sc = (cl == 0 ? 0.f : Lc[cl - 1]) - __logf(e[u]);
            if (sc > best) { second = best; best = sc; bi = cl; }
            else second = fmaxf(second, sc);
          }
        }
      }
      float wb = best;
      int wi = bi;
      wave_argmax(wb, wi);
      float runner = bi == wi ? second : best;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) runner = fmaxf(runner, __shfl_xor(runner, off, 64));
      if (!((wb - runner) - kWideMargin * (1.0f + fabsf(wb)) > 0.f)) {  // (uniform) too close: the reference arithmetic decides
        const float* yb = a.y + (size_t)r * a.ldY + wc.y_off;
        const float* srow = a.s ? a.s + (size_t)r * a.K : nullptr;
        best = -INFINITY;
        bi = 0x7FFFFFFF;
        for (int c0 = 4 * lane; c0 < C; c0 += 256) {
          float e[4];
          draw4(c0, e);
#pragma unroll 1
          for (int u = 0; u < 4; ++u) {
            const int cl = c0 + u;
            if (cl >= C) break;
            float acc = 0.f;
            if (cl > 0) {  // sequential, individually rounded dot of [y ; s] with the class's weight row
              const float* w = a.params + wc.w_off + (size_t)(cl - 1) * wc.ldw;
              for (int i = 0; i < wc.yd; ++i) acc = detm::add(acc, detm::mul(w[i], yb[i]));
              if (srow) {
                for (int j = 0; j < a.K; ++j) acc = detm::add(acc, detm::mul(w[wc.yd + j], srow[j]));
              } else {
                acc = detm::add(acc, w[wc.yd + kc]);
              }
            }
            const float sc = detm::sub(acc, detm::log(e[u]));
            if (sc > best) { best = sc; bi = cl; }
          }
        }
        wb = best;
        wi = bi;
        wave_argmax(wb, wi);
      }
      if (lane == 0) a.out_codes[(size_t)r * a.ld_disc + a.cols[wc.col].x_index] = (int16_t)wi;
    }
  }
}

struct PriorArgs {
  const float* params;
  int prior_mu_off, prior_logvar_off, pm_w_off, pm_b_off;
  const float* cdf;
  const float* u;
  const float* eps;
  uint64_t seed;
  int64_t row_offset, n;
  int K, D;
  float* z;
  int32_t* comp;
};

// rows per block iteration of the latent-draw kernels: a thread per row draws the component (all 256 threads busy),
// then the block writes the rows' z quads
constexpr int kLatTile = 256;
__global__ __launch_bounds__(256) void k_prior_latents(PriorArgs a) {
  __shared__ int k_s[kLatTile];
  const int tid = threadIdx.x;
  const Philox rng(a.seed);
  const int64_t n_tiles = (a.n + kLatTile - 1) / kLatTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kLatTile;
    const int nrows = (int)min((int64_t)kLatTile, a.n - r0);
    __syncthreads();
    if (tid < nrows) {
      const int64_t r = r0 + tid;
      const uint64_t gid = (uint64_t)(a.row_offset + r);
      const float u = a.u ? a.u[r] : philox_u(rng, (uint32_t)gid, 0, kStreamPriorU, (uint32_t)(gid >> 32));
      int k = 0;
      for (int j = 0; j < a.K - 1; ++j) k += (u >= a.cdf[j]) ? 1 : 0;
      k_s[tid] = k;
      a.comp[r] = k;
    }
    __syncthreads();
    const int dq = (a.D + 3) / 4;
    for (int idx = tid; idx < nrows * dq; idx += 256) {
      const int lr = idx / dq, q = idx % dq;
      const int64_t r = r0 + lr;
      const int k = k_s[lr];
      const uint64_t gid = (uint64_t)(a.row_offset + r);
      float ev[4] = {0.f, 0.f, 0.f, 0.f};
      if (!a.eps) {
        const uint4 w = rng((uint32_t)gid, (uint32_t)q, kStreamPriorEps, (uint32_t)(gid >> 32));
        const float2 n0 = normal2(w.x, w.y), n1 = normal2(w.z, w.w);
        ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = q * 4 + e;
        if (d >= a.D) break;
        float mu, lv;
        if (a.prior_mu_off >= 0) {
          mu = a.params[a.prior_mu_off + k * a.D + d];
          lv = a.params[a.prior_logvar_off + k * a.D + d];
        } else {  // PriorMean: mu[k,d] = W[d,k] + b[d]; logvar == 0 (prior.py:31-39, model.py:538-542)
          mu = a.params[a.pm_w_off + d * a.K + k] + a.params[a.pm_b_off + d];
          lv = 0.f;
        }
        const float eps = a.eps ? a.eps[(size_t)r * a.D + d] : ev[e];
        a.z[(size_t)r * a.D + d] = mu + eps * __expf(0.5f * lv);
      }
    }
  }
}

struct CondArgs {
  const int64_t* class_offsets;
  const int64_t* class_rows;
  const int32_t* labels;
  const float* train_probs;
  const float* train_mu;
  const float* train_logvar;
  const float* u_row;
  const float* e_comp;
  const float* eps;
  uint64_t seed;
  int64_t row_offset, n;
  int K, D;
  float* z;
  int32_t* comp;
};

__global__ __launch_bounds__(256) void k_cond_latents(CondArgs a) {
  __shared__ int k_s[kLatTile];
  __shared__ long long row_s[kLatTile];
  const int tid = threadIdx.x;
  const Philox rng(a.seed);
  const int64_t n_tiles = (a.n + kLatTile - 1) / kLatTile;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * kLatTile;
    const int nrows = (int)min((int64_t)kLatTile, a.n - r0);
    __syncthreads();
    if (tid < nrows) {
      const int64_t r = r0 + tid;
      const uint64_t gid = (uint64_t)(a.row_offset + r);
      const int lab = a.labels[r];
      const int64_t start = a.class_offsets[lab];
      const int64_t count = a.class_offsets[lab + 1] - start;
      const float u = a.u_row ? a.u_row[r] : philox_u(rng, (uint32_t)gid, 0, kStreamCondU, (uint32_t)(gid >> 32));
      int64_t pick = (int64_t)detm::mul(u, (float)count);  // torch.randint analogue, model.py:3088
      pick = min(pick, count - 1);
      const int64_t row = a.class_rows[start + pick];
      const float* pr = a.train_probs + (size_t)row * a.K;
      float best = -1.0f;
      int kb = 0;
      uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
      for (int j = 0; j < a.K; ++j) {  // Categorical(probs).sample(): argmax(p / e), model.py:3092-3094
        float e;
        if (a.e_comp) e = a.e_comp[(size_t)r * a.K + j];
        else {  // one Philox block = the draws of four components
          if ((j & 3) == 0) q4 = rng((uint32_t)gid, (uint32_t)(j >> 2), kStreamCondExp, (uint32_t)(gid >> 32));
          e = exp_draw((j & 3) == 0 ? q4.x : (j & 3) == 1 ? q4.y : (j & 3) == 2 ? q4.z : q4.w);
        }
        const float sc = detm::dvd(pr[j], e);
        if (sc > best) { best = sc; kb = j; }
      }
      k_s[tid] = kb;
      row_s[tid] = row;
      a.comp[r] = kb;
    }
    __syncthreads();
    const int dq = (a.D + 3) / 4;
    for (int idx = tid; idx < nrows * dq; idx += 256) {
      const int lr = idx / dq, q = idx % dq;
      const int64_t r = r0 + lr;
      const int k = k_s[lr];
      const int64_t row = row_s[lr];
      const uint64_t gid = (uint64_t)(a.row_offset + r);
      float ev[4] = {0.f, 0.f, 0.f, 0.f};
      if (!a.eps) {
        const uint4 w = rng((uint32_t)gid, (uint32_t)q, kStreamPriorEps, (uint32_t)(gid >> 32));
        const float2 n0 = normal2(w.x, w.y), n1 = normal2(w.z, w.w);
        ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
      }
      const float* mu = a.train_mu + ((size_t)row * a.K + k) * a.D;
      const float* lv = a.train_logvar + ((size_t)row * a.K + k) * a.D;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = q * 4 + e;
        if (d >= a.D) break;
        const float eps = a.eps ? a.eps[(size_t)r * a.D + d] : ev[e];
        a.z[(size_t)r * a.D + d] = mu[d] + eps * __expf(0.5f * lv[d]);
      }
    }
  }
}


// Posterior draws for the Monte-Carlo predictive (SUAVE._posterior_draws, model.py:3304-3336):
// L component draws per row (Gumbel-max on the logits == Categorical(logits).sample) and the
// reparameterised latent of the drawn component.  Output row l*B + b.
struct PostDrawArgs {
  const float* enc_out;  // [B, K + 2KD] logits | mu | clamped logvar (suave_encode layout)
  const float* e;        // [L,B,K] Exp(1) or null
  const float* eps;      // [L,B,D] or null
  uint64_t seed;
  int64_t row_offset;
  int B, L, K, D;
  float* z;              // [L*B, D]
  int32_t* comp;         // [L*B]
  int row_major;         // output row of (row b, draw l): l * B + b (the reference's order), or b * L + l
};

__global__ __launch_bounds__(256) void k_posterior_draws(PostDrawArgs a) {
  const Philox rng(a.seed);
  const int64_t total = (int64_t)a.L * a.B;
  const int E = a.K + 2 * a.K * a.D;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (int64_t)gridDim.x * 256) {
    const int b = (int)(v % a.B);
    const int64_t vo = a.row_major ? (int64_t)b * a.L + v / a.B : v;  // (injected draws are always draw-major: v)
    // generator key of (table row row_offset + b, draw l): independent of how the rows are chunked or sharded
    const uint64_t gid = (uint64_t)(a.row_offset + b) * (uint64_t)a.L + (uint64_t)(v / a.B);
    const float* row = a.enc_out + (size_t)b * E;
    float best = -INFINITY;
    int kb = 0;
    uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
    for (int j = 0; j < a.K; ++j) {
      float e;
      if (a.e) e = a.e[(size_t)v * a.K + j];
      else {
        if ((j & 3) == 0) q4 = rng((uint32_t)gid, (uint32_t)(j >> 2), kStreamCondExp, (uint32_t)(gid >> 32));
        e = exp_draw((j & 3) == 0 ? q4.x : (j & 3) == 1 ? q4.y : (j & 3) == 2 ? q4.z : q4.w);
      }
      const float sc = detm::sub(row[j], detm::log(e));
      if (sc > best) { best = sc; kb = j; }
    }
    a.comp[vo] = kb;
    const float* mu = row + a.K + kb * a.D;
    const float* lv = mu + a.K * a.D;
    for (int d = 0; d < a.D; d += 2) {
      float e0, e1 = 0.f;
      if (a.eps) {
        e0 = a.eps[(size_t)v * a.D + d];
        if (d + 1 < a.D) e1 = a.eps[(size_t)v * a.D + d + 1];
      } else {
        const uint4 w = rng((uint32_t)gid, (uint32_t)(d >> 1), kStreamPriorEps, (uint32_t)(gid >> 32));
        const float2 n2 = normal2(w.x, w.y);
        e0 = n2.x; e1 = n2.y;
      }
      a.z[(size_t)vo * a.D + d] = mu[d] + e0 * __expf(0.5f * lv[d]);
      if (d + 1 < a.D) a.z[(size_t)vo * a.D + d + 1] = mu[d + 1] + e1 * __expf(0.5f * lv[d + 1]);
    }
  }
}

// The same draws (same generator keys, same arithmetic: bit-identical outputs) with eight lanes per table row, for
// D % 4 == 0 and K <= 32.  The one-thread-per-draw kernel above read its row of enc_out at a 2.6 KB lane stride and
// wrote z as 64 scattered 4-byte stores per instruction (460 us per 524 288 draws, 20x its HBM time).  Here a group
// of eight lanes owns ROW b for all L draws: its logits stay in registers, lane q of the group draws the Philox
// block of components 4q .. 4q + 3 and the group's argmax is three shuffles; every lane then writes one float4 of z,
// so a wave stores 1 KB of eight consecutive rows per draw, and the row's mu / logvar are re-read from L1 / L2 only.
__global__ __launch_bounds__(256) void k_posterior_draws8(PostDrawArgs a) {
  const Philox rng(a.seed);
  const int K = a.K, D = a.D, dq = a.D >> 2;
  const int E = K + 2 * K * D;
  const int l8 = threadIdx.x & 7;
  for (int64_t b = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); b < a.B; b += (int64_t)gridDim.x * 32) {
    const float* row = a.enc_out + (size_t)b * E;
    const bool has_k = 4 * l8 < K;
    float lg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) lg[u] = 4 * l8 + u < K ? row[4 * l8 + u] : 0.f;
    for (int l = 0; l < a.L; ++l) {
      const int64_t v = (int64_t)l * a.B + b;
      const int64_t vo = a.row_major ? b * a.L + l : v;
      const uint64_t gid = (uint64_t)(a.row_offset + b) * (uint64_t)a.L + (uint64_t)l;
      const uint32_t glo = (uint32_t)gid, ghi = (uint32_t)(gid >> 32);
      float best = -INFINITY;
      int kb = has_k ? 4 * l8 : 0x7FFFFFFF;
      if (has_k) {
        uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
        if (!a.e) q4 = rng(glo, (uint32_t)l8, kStreamCondExp, ghi);
        const uint32_t word[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * l8 + u;
          if (j < K) {
            const float e = a.e ? a.e[(size_t)v * K + j] : exp_draw(word[u]);
            const float sc = detm::sub(lg[u], detm::log(e));
            if (sc > best) { best = sc; kb = j; }
          }
        }
      }
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {  // first maximum wins, as in the sequential scan
        const float ob = __shfl_xor(best, off, 64);
        const int ok = __shfl_xor(kb, off, 64);
        if (ob > best || (ob == best && ok < kb)) { best = ob; kb = ok; }
      }
      if (l8 == 0) a.comp[vo] = kb;
      const float* mu = row + K + kb * D;
      const float* lv = mu + K * D;
      for (int q = l8; q < dq; q += 8) {
        float ev[4];
        if (a.eps) {
          const float4 e4 = *reinterpret_cast<const float4*>(a.eps + (size_t)v * D + 4 * q);
          ev[0] = e4.x; ev[1] = e4.y; ev[2] = e4.z; ev[3] = e4.w;
        } else {
          const uint4 w0 = rng(glo, (uint32_t)(2 * q), kStreamPriorEps, ghi);
          const uint4 w1 = rng(glo, (uint32_t)(2 * q + 1), kStreamPriorEps, ghi);
          const float2 n0 = normal2(w0.x, w0.y), n1 = normal2(w1.x, w1.y);
          ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
        }
        float zz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) zz[u] = mu[4 * q + u] + ev[u] * __expf(0.5f * lv[4 * q + u]);
        *reinterpret_cast<float4*>(a.z + (size_t)vo * D + 4 * q) = make_float4(zz[0], zz[1], zz[2], zz[3]);
      }
    }
  }
}

// Parameters of ONE likelihood head (the ``params`` dict of the reference heads, decoder.py:89,
// 134-138,173,215,259) for every row: real -> mean, var; pos -> mean_log, var_log, mean;
// count -> rate; cat -> probs[C]; ordinal -> probs[C].
struct HeadParamArgs {
  suave_col_t col;
  const float* params;
  const float* y;    // [n, ldY]
  const float* s;    // [n, K] or null
  const int32_t* comp;
  int64_t n;
  int K, ldY, width;
  float* out;        // [n, width]
};

__global__ __launch_bounds__(256) void k_head_params(HeadParamArgs a) {
  const suave_col_t col = a.col;
  const int yd = col.y_dim, K = a.K;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n; r += (int64_t)gridDim.x * 256) {
    const float* yb = a.y + (size_t)r * a.ldY + col.y_off;
    const float* srow = a.s ? a.s + (size_t)r * K : nullptr;
    const int kc = a.s ? 0 : a.comp[r];
    auto dot = [&](const float* w, int ydim) {
      float acc = 0.f;
      for (int i = 0; i < ydim; ++i) acc += w[i] * yb[i];
      if (srow) { for (int j = 0; j < K; ++j) acc += w[ydim + j] * srow[j]; }
      else acc += w[ydim + kc];
      return acc;
    };
    float* o = a.out + (size_t)r * a.width;
    const float* w = a.params + col.w_off;
    if (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) {
      const float am = dot(w, yd);
      const float v = softplus_f(dot(a.params + col.w2_off, 0)) + kEps;
      const float mean = am * col.scale + col.shift, vs = v * (col.scale * col.scale);
      o[0] = mean;
      o[1] = vs;
      if (col.type == SUAVE_TYPE_POS) o[2] = __expf(mean + 0.5f * fmaxf(vs, 0.f)) - 1.0f;
    } else if (col.type == SUAVE_TYPE_COUNT) {
      o[0] = softplus_f(dot(w, yd)) + kEps;
    } else if (col.type == SUAVE_TYPE_CAT) {
      const int C = col.n_classes, ldw = cat_ldw(col, K);
      float mx = 0.f;
      o[0] = 0.f;
      for (int c = 1; c < C; ++c) { o[c] = dot(w + (c - 1) * ldw, yd); mx = fmaxf(mx, o[c]); }
      float sum = 0.f;
      for (int c = 0; c < C; ++c) { o[c] = __expf(o[c] - mx); sum += o[c]; }
      for (int c = 0; c < C; ++c) o[c] /= sum;
    } else {
      const int C = col.n_classes;
      const float mo = dot(a.params + col.w2_off, yd);
      float thr = 0.f, prev = 0.f;
      for (int c = 0; c < C - 1; ++c) {
        thr += softplus_f(dot(w + c * K, 0)) + kEps;
        const float sg = sigmoid_f(thr - mo);
        o[c] = fminf(fmaxf(sg - prev, kEps), 1.0f);
        prev = sg;
      }
      o[C - 1] = fminf(fmaxf(1.0f - prev, kEps), 1.0f);
    }
  }
}

}  // namespace

int launch_head_params(const suave_plan* p, int col_index, const float* params, const float* y, const float* s,
                       const int32_t* comp, int64_t n, float* out, int width, hipStream_t st) {
  if (n <= 0) return 0;
  HeadParamArgs a{};
  a.col = p->cols_host[col_index];
  a.params = params; a.y = y; a.s = s; a.comp = comp; a.n = n; a.K = p->model.n_components; a.ldY = p->ldY;
  a.width = width; a.out = out;
  ProfScope prof_scope(kProfSample, st);
  hipLaunchKernelGGL(k_head_params, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_sample_heads(const suave_plan* p, const int32_t* cls_off_dev, const float* params, const float* y,
                        const float* s, const int32_t* comp, int64_t n, const float* nn, const float* nu,
                        const float* ne, uint64_t seed, int64_t row_offset, float* out_num, int ld_num,
                        int16_t* out_codes, int ld_disc, hipStream_t st, const float* wide_logits) {
  if (n <= 0) return 0;
  const suave_model_t& m = p->model;
  SampleArgs a{};
  a.cols = p->cols_dev; a.cls_off = cls_off_dev; a.params = params; a.y = y; a.s = s; a.comp = comp; a.n = n;
  a.n_cols = m.n_cols; a.n_num = m.n_num; a.n_disc = m.n_disc; a.K = m.n_components; a.ldY = p->ldY; a.Y = m.y_rows;
  a.total_classes = p->total_classes;
  a.noise_normal = nn; a.noise_uniform = nu; a.noise_exp = ne; a.seed = seed; a.row_offset = row_offset;
  a.out_num = out_num; a.ld_num = ld_num; a.out_codes = out_codes; a.ld_disc = ld_disc;
  a.ldy = p->ldY | 1; a.ldon = std::max(m.n_num, 1) | 1; a.ldoc = std::max(m.n_disc, 1) | 1;
  size_t lds = sizeof(float) * ((size_t)64 * a.ldy + 64 * a.ldon + 64 * a.ldoc + 64 * m.n_components);
  const bool staged = lds <= 64 * 1024;
  if (staged && lds + sizeof(float) * m.head_len <= 78 * 1024) {  // two blocks per CU keep their 160 KB
    a.head_off = m.head_off; a.head_len = m.head_len;
    lds += sizeof(float) * m.head_len;
  }
  static thread_local size_t lds_set = 0;
  if (staged && lds > 48 * 1024 && lds > lds_set) {
    SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sample_heads<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  const int64_t tiles = (n + 63) / 64;
  ProfScope prof_scope(kProfSample, st);
  const int n_wide = (int)p->wide_host.size();
  const bool wide_race = wide_logits != nullptr && n_wide > 0;
  if (wide_race) a.skip_wide = 1;
  if (staged) {
    const int grid = (int)std::min<int64_t>(tiles, 2 * (int64_t)p->n_cu);
    hipLaunchKernelGGL(k_sample_heads<true>, dim3(grid), dim3(kSampleThreads), lds, st, a);
  } else {
    const int grid = (int)std::min<int64_t>(tiles, 8 * (int64_t)p->n_cu);
    hipLaunchKernelGGL(k_sample_heads<false>, dim3(grid), dim3(kSampleThreads), 0, st, a);
  }
  SUAVE_CHECK_HIP(hipGetLastError());
  if (wide_race) {  // the wide categorical columns: race on the GEMM's logits.  AFTER the row kernel: its staged variant
    // writes every code column of a tile from LDS, the skipped ones included
    WideSampleArgs wa{};
    wa.wide = p->wide_dev; wa.n_wide = n_wide; wa.cols = p->cols_dev; wa.cls_off = cls_off_dev; wa.params = params;
    wa.y = y; wa.ldY = p->ldY; wa.L = wide_logits; wa.ldL = p->ldL; wa.s = s; wa.comp = comp; wa.K = m.n_components;
    wa.noise_exp = ne; wa.total_classes = p->total_classes; wa.seed = seed; wa.row_offset = row_offset; wa.n = n;
    wa.out_codes = out_codes; wa.ld_disc = ld_disc;
    hipLaunchKernelGGL(k_wide_sample, dim3((int)std::min<int64_t>((n + 3) / 4, 16 * (int64_t)p->n_cu)), dim3(256), 0, st, wa);
    SUAVE_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace suave

using namespace suave;

extern "C" int suave_prior_latents(const suave_plan* p, const float* params, const float* cdf, const float* u,
                                   const float* eps, uint64_t seed, int64_t row_offset, int64_t n, void* stream,
                                   float* z, int32_t* component) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p && params, "prior_latents: null plan/params");
  if (n <= 0) return 0;  // empty request: nothing to read or write, buffers may be null
  SUAVE_REQUIRE(cdf && z && component, "prior_latents: null argument");
  PriorArgs a{};
  a.params = params;
  a.prior_mu_off = p->model.prior_mu_off; a.prior_logvar_off = p->model.prior_logvar_off;
  a.pm_w_off = p->model.pm_w_off; a.pm_b_off = p->model.pm_b_off;
  a.cdf = cdf; a.u = u; a.eps = eps; a.seed = seed; a.row_offset = row_offset; a.n = n;
  a.K = p->model.n_components; a.D = p->model.latent_dim; a.z = z; a.comp = component;
  const int grid = (int)std::min<int64_t>((n + kLatTile - 1) / kLatTile, 4096);
  hipLaunchKernelGGL(k_prior_latents, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int suave_conditional_latents(const suave_plan* p, const int64_t* class_offsets, const int64_t* class_rows,
                                         const int32_t* labels, const float* train_probs, const float* train_mu,
                                         const float* train_logvar, const float* u_row, const float* e_comp,
                                         const float* eps, uint64_t seed, int64_t row_offset, int64_t n, void* stream,
                                         float* z, int32_t* component) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p, "conditional_latents: null plan");
  if (n <= 0) return 0;  // empty request: buffers may be null
  SUAVE_REQUIRE(class_offsets && class_rows && labels && train_probs && train_mu && train_logvar && z && component,
                "conditional_latents: null argument");
  CondArgs a{};
  a.class_offsets = class_offsets; a.class_rows = class_rows; a.labels = labels; a.train_probs = train_probs;
  a.train_mu = train_mu; a.train_logvar = train_logvar; a.u_row = u_row; a.e_comp = e_comp; a.eps = eps;
  a.seed = seed; a.row_offset = row_offset; a.n = n; a.K = p->model.n_components; a.D = p->model.latent_dim;
  a.z = z; a.comp = component;
  const int grid = (int)std::min<int64_t>((n + kLatTile - 1) / kLatTile, 4096);
  hipLaunchKernelGGL(k_cond_latents, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

static int posterior_draws_impl(const suave_plan* p, const float* enc_out, int64_t n_rows, int32_t n_draws, const float* e,
                                const float* eps, uint64_t seed, int64_t row_offset, void* stream, float* z,
                                int32_t* component, int row_major) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p, "posterior_draws: null plan");
  SUAVE_REQUIRE(n_draws >= 1, "posterior_draws: L must be a positive integer");  // model.py:3313-3314
  SUAVE_REQUIRE(n_rows >= 0 && n_rows * (int64_t)n_draws < (int64_t(1) << 31), "posterior_draws: too many rows");
  if (n_rows == 0) return 0;  // empty request: buffers may be null
  SUAVE_REQUIRE(enc_out && z && component, "posterior_draws: null argument");
  PostDrawArgs a{};
  a.enc_out = enc_out; a.e = e; a.eps = eps; a.seed = seed; a.row_offset = row_offset; a.B = (int)n_rows; a.L = n_draws;
  a.K = p->model.n_components; a.D = p->model.latent_dim; a.z = z; a.comp = component; a.row_major = row_major;
  const int64_t total = n_rows * n_draws;
  const bool aligned = (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (!eps || (reinterpret_cast<uintptr_t>(eps) & 15) == 0);
  if (a.D % 4 == 0 && a.K <= 32 && aligned)
    hipLaunchKernelGGL(k_posterior_draws8, dim3((int)std::min<int64_t>((n_rows + 31) / 32, 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(k_posterior_draws, dim3((int)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int suave_posterior_draws(const suave_plan* p, const float* enc_out, int64_t n_rows, int32_t n_draws,
                                     const float* e, const float* eps, uint64_t seed, int64_t row_offset, void* stream,
                                     float* z, int32_t* component) {
  return posterior_draws_impl(p, enc_out, n_rows, n_draws, e, eps, seed, row_offset, stream, z, component, 0);
}

extern "C" int suave_posterior_draws_by_row(const suave_plan* p, const float* enc_out, int64_t n_rows, int32_t n_draws,
                                            const float* e, const float* eps, uint64_t seed, int64_t row_offset,
                                            void* stream, float* z, int32_t* component) {
  return posterior_draws_impl(p, enc_out, n_rows, n_draws, e, eps, seed, row_offset, stream, z, component, 1);
}
