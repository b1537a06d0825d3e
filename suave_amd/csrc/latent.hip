// Row-wise kernels around the latent layer: packed batch -> encoder input,
// Gumbel-softmax component assignment + reparameterisation (forward), their
// backward scatter into the encoder-head gradient, and posterior moment matching.
//
// Reference: suave/model.py:2784-2922 (_prepare_training_tensors layout),
// :1573-1598 (assignment + reparam), :2924-2942, :2944-2964
// (_mixture_posterior_statistics); torch F.gumbel_softmax.
#include "kernels.h"
#include "detmath.h"

namespace suave {

// enc_src entry: bits 0..15 class index, bits 16..29 column, bit 30 = mask entry; -1 = pad
__global__ __launch_bounds__(256) void k_expand(const int32_t* __restrict__ enc_src, const suave_col_t* __restrict__ cols,
                                                suave_batch_t b, int ldI, float* __restrict__ x) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + wave; r < b.n_rows; r += gridDim.x * 4) {
    const int64_t tr = b.row_index ? (int64_t)b.row_index[r] : b.row_begin + r;
    const float* xn = b.xnum + tr * b.ld_num;
    const int16_t* cd = b.codes + tr * b.ld_disc;
    const uint8_t* mk = b.mask + tr * b.ld_mask;
    for (int i = lane; i < ldI; i += 64) {
      const int32_t e = enc_src[i];
      float v = 0.f;
      if (e >= 0) {
        const int c = (e >> 16) & 0x3FFF;
        if (e & (1 << 30)) {
          v = mk[c] ? 1.f : 0.f;
        } else {
          const suave_col_t col = cols[c];
          const int cls = e & 0xFFFF;
          if (col.type <= SUAVE_TYPE_COUNT) {
            v = xn[col.x_index];
          } else {
            const int code = cd[col.x_index];
            if (col.type == SUAVE_TYPE_CAT) v = (code == cls) ? 1.f : 0.f;
            else v = (code >= 0 && cls <= code) ? 1.f : 0.f;  // thermometer, model.py:2890-2891
          }
        }
      }
      x[(size_t)r * ldI + i] = v;
    }
  }
}


// supervised: y_soft = softmax((logits + g)/tau), k = argmax, z = mu_k + eps * exp(lv_k / 2)
__global__ __launch_bounds__(256) void k_latent_fwd(LatentFwdArgs a) {
  __shared__ int k_s[64];
  const int tid = threadIdx.x;
  const Philox rng(a.seed);
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, a.n_rows - r0);
    if (tid < nrows) {
      const int r = r0 + tid;
      const uint32_t rid = a.row_index ? (uint32_t)a.row_index[r] : (uint32_t)(a.row_begin + r);
      const float* lg = a.encout + (size_t)r * a.ldE;
      float t[SUAVE_MAX_COMPONENTS];
      float best = -INFINITY;
      int kb = 0;
#pragma unroll 1
      for (int j = 0; j < a.K; ++j) {
        float e;
        if (a.exp_noise) {
          e = a.exp_noise[(size_t)r * a.K + j];
        } else {
          const uint4 w = rng(rid, (uint32_t)(j >> 2), kStreamGumbel, (uint32_t)a.step);
          const uint32_t word = (j & 3) == 0 ? w.x : (j & 3) == 1 ? w.y : (j & 3) == 2 ? w.z : w.w;
          e = -__logf(u01(word));
        }
        // Gumbel noise g = -log(e); the assignment path is deterministic binary32
        const float tj = detm::dvd(detm::sub(lg[j], detm::log(e)), a.tau);
        t[j] = tj;
        if (tj > best) { best = tj; kb = j; }
      }
      float sum = 0.f;
#pragma unroll 1
      for (int j = 0; j < a.K; ++j) { t[j] = __expf(t[j] - best); sum += t[j]; }
      const float inv = 1.0f / sum;
#pragma unroll 1
      for (int j = 0; j < a.K; ++j) a.ysoft[(size_t)r * a.ldK + j] = t[j] * inv;
      a.kidx[r] = kb;
      if (a.comp_out) a.comp_out[r] = kb;
      k_s[tid] = kb;
    }
    __syncthreads();
    const int dq = (a.D + 3) / 4;
    for (int idx = tid; idx < nrows * dq; idx += 256) {
      const int lr = idx / dq, q = idx % dq;
      const int r = r0 + lr, k = k_s[lr];
      const uint32_t rid = a.row_index ? (uint32_t)a.row_index[r] : (uint32_t)(a.row_begin + r);
      float ev[4];
      if (a.eps_in == nullptr) {
        const uint4 w = rng(rid, (uint32_t)q, kStreamEps, (uint32_t)a.step);
        const float2 n0 = normal2(w.x, w.y), n1 = normal2(w.z, w.w);
        ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
      }
      const float* mu = a.encout + (size_t)r * a.ldE + a.K + k * a.D;
      const float* lv = mu + a.K * a.D;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = q * 4 + e;
        if (d >= a.D) break;
        const float eps = a.eps_in ? a.eps_in[(size_t)r * a.D + d] : ev[e];
        const float lvc = fminf(fmaxf(lv[d], kLogvarMin), kLogvarMax);
        const float zz = mu[d] + eps * __expf(0.5f * lvc);
        a.z[(size_t)r * a.ldD + d] = zz;
        a.eps_buf[(size_t)r * a.ldD + d] = eps;
        if (a.latent_out) a.latent_out[(size_t)r * a.D + d] = zz;
      }
    }
    __syncthreads();
  }
}


// Straight-through Gumbel path + scatter of the selected component's gradients
// (SURVEY Appendix A): all other entries of d(enc_out) are zero.
__global__ __launch_bounds__(256) void k_latent_bwd(LatentBwdArgs a) {
  __shared__ int k_s[64];
  const int tid = threadIdx.x;
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, a.n_rows - r0);
    if (tid < nrows) {
      const int r = r0 + tid;
      k_s[tid] = a.kidx[r];
      const float* ys = a.ysoft + (size_t)r * a.ldK;
      float ds[SUAVE_MAX_COMPONENTS];
      float inner = 0.f;
#pragma unroll 1
      for (int j = 0; j < a.K; ++j) {
        float v = 0.f;
        for (int c = 0; c < a.n_chunks; ++c) v += a.ds_part[((size_t)c * a.n_rows + r) * a.ldK + j];
        ds[j] = v;
        inner += ys[j] * v;
      }
      const float inv_tau = 1.0f / a.tau;
#pragma unroll 1
      for (int j = 0; j < a.K; ++j)
        a.dencout[(size_t)r * a.ldE + j] = ys[j] * (ds[j] - inner) * inv_tau + a.gkl_logit[(size_t)r * a.ldK + j];
    }
    __syncthreads();
    const int kd = a.K * a.D;
    const int width = a.ldE - a.K;  // mu | logvar | pad
    for (int idx = tid; idx < nrows * width; idx += 256) {
      const int lr = idx / width, c = idx % width;
      const int r = r0 + lr, k = k_s[lr];
      float v = 0.f;
      if (c < 2 * kd) {
        const bool is_lv = c >= kd;
        const int cc = is_lv ? c - kd : c;
        const int j = cc / a.D, d = cc % a.D;
        if (j == k) {
          const float g = a.dz[(size_t)r * a.ldD + d];
          if (!is_lv) {
            v = g + a.gkl_mu[(size_t)r * a.ldD + d];
          } else {
            const float raw = a.encout[(size_t)r * a.ldE + a.K + c];
            if (raw >= kLogvarMin && raw <= kLogvarMax) {  // clamp passes gradient inside its range only
              const float eps = a.eps_buf[(size_t)r * a.ldD + d];
              v = g * eps * 0.5f * __expf(0.5f * raw) + a.gkl_lv[(size_t)r * a.ldD + d];
            }
          }
        }
      }
      a.dencout[(size_t)r * a.ldE + a.K + c] = v;
    }
    __syncthreads();
  }
}

// encode: clamp logvar, softmax(logits * inv_temperature), mixture moment matching
__global__ __launch_bounds__(256) void k_posterior_stats(const float* __restrict__ encout, int ldE, int K, int D, int n_rows,
                                                         float inv_temp, float* __restrict__ enc_out_user,
                                                         float* __restrict__ post_mean, float* __restrict__ post_logvar,
                                                         float* __restrict__ probs) {
  __shared__ float p_s[64 * SUAVE_MAX_COMPONENTS];
  const int tid = threadIdx.x;
  const int E = K + 2 * K * D;
  for (int r0 = blockIdx.x * 64; r0 < n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, n_rows - r0);
    if (tid < nrows) {
      const float* lg = encout + (size_t)(r0 + tid) * ldE;
      float mx = -INFINITY;
      for (int j = 0; j < K; ++j) mx = fmaxf(mx, lg[j] * inv_temp);
      float sum = 0.f;
      for (int j = 0; j < K; ++j) { const float e = __expf(lg[j] * inv_temp - mx); p_s[tid * SUAVE_MAX_COMPONENTS + j] = e; sum += e; }
      const float inv = 1.0f / sum;
      for (int j = 0; j < K; ++j) {
        const float p = p_s[tid * SUAVE_MAX_COMPONENTS + j] * inv;
        p_s[tid * SUAVE_MAX_COMPONENTS + j] = p;
        if (probs) probs[(size_t)(r0 + tid) * K + j] = p;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < nrows * D; idx += 256) {
      const int lr = idx / D, d = idx % D;
      const float* row = encout + (size_t)(r0 + lr) * ldE + K;
      float m = 0.f, s2 = 0.f;
      for (int j = 0; j < K; ++j) {
        const float p = p_s[lr * SUAVE_MAX_COMPONENTS + j];
        const float mu = row[j * D + d];
        const float lv = fminf(fmaxf(row[K * D + j * D + d], kLogvarMin), kLogvarMax);
        m += p * mu;
        s2 += p * (__expf(lv) + mu * mu);
      }
      const float var = fmaxf(s2 - m * m, 1e-6f);
      if (post_mean) post_mean[(size_t)(r0 + lr) * D + d] = m;
      if (post_logvar) post_logvar[(size_t)(r0 + lr) * D + d] = __logf(var);
    }
    if (enc_out_user) {
      for (int idx = tid; idx < nrows * E; idx += 256) {
        const int lr = idx / E, c = idx % E;
        float v = encout[(size_t)(r0 + lr) * ldE + c];
        if (c >= K + K * D) v = fminf(fmaxf(v, kLogvarMin), kLogvarMax);
        enc_out_user[(size_t)(r0 + lr) * E + c] = v;
      }
    }
    __syncthreads();
  }
}

// ---- unsupervised branch -------------------------------------------------------
// Relaxed (hard=False) Gumbel-softmax weights over the components and a reparameterised latent for
// EVERY component (model.py:1576-1580, :1620): virtual row v = b*K + k feeds the decoder.
__global__ __launch_bounds__(256) void k_mix_fwd(MixFwdArgs a) {
  __shared__ float w_s[64 * SUAVE_MAX_COMPONENTS];
  const int tid = threadIdx.x;
  const Philox rng(a.seed);
  const int K = a.K, D = a.D;
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, a.n_rows - r0);
    __syncthreads();
    if (tid < nrows) {
      const int r = r0 + tid;
      const uint32_t rid = a.row_index ? (uint32_t)a.row_index[r] : (uint32_t)(a.row_begin + r);
      const float* lg = a.encout + (size_t)r * a.ldE;
      float* ws = w_s + tid * SUAVE_MAX_COMPONENTS;
      float best = -INFINITY;
#pragma unroll 1
      for (int j = 0; j < K; ++j) {
        float tj = lg[j];
        if (a.tau > 0.f) {
          float e;
          if (a.exp_noise) e = a.exp_noise[(size_t)r * K + j];
          else {
            const uint4 q = rng(rid, (uint32_t)(j >> 2), kStreamGumbel, (uint32_t)a.step);
            const uint32_t word = (j & 3) == 0 ? q.x : (j & 3) == 1 ? q.y : (j & 3) == 2 ? q.z : q.w;
            e = -__logf(u01(word));
          }
          tj = (tj - __logf(e)) / a.tau;
        }
        ws[j] = tj;
        best = fmaxf(best, tj);
      }
      float sum = 0.f;
#pragma unroll 1
      for (int j = 0; j < K; ++j) { ws[j] = __expf(ws[j] - best); sum += ws[j]; }
      const float inv = 1.0f / sum;
#pragma unroll 1
      for (int j = 0; j < K; ++j) {
        const float wj = ws[j] * inv;
        ws[j] = wj;
        a.w[(size_t)r * a.ldK + j] = wj;
        a.wflat[(size_t)r * K + j] = wj;
      }
    }
    __syncthreads();
    const int dq = (D + 3) / 4;
    for (int idx = tid; idx < nrows * K * dq; idx += 256) {
      const int lr = idx / (K * dq), rem = idx % (K * dq);
      const int k = rem / dq, q = rem % dq;
      const int r = r0 + lr;
      const uint32_t rid = a.row_index ? (uint32_t)a.row_index[r] : (uint32_t)(a.row_begin + r);
      float ev[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.eps_in == nullptr) {
        const uint4 wd = rng(rid, (uint32_t)(k * dq + q), kStreamEps, (uint32_t)a.step);
        const float2 n0 = normal2(wd.x, wd.y), n1 = normal2(wd.z, wd.w);
        ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
      }
      const float* mu = a.encout + (size_t)r * a.ldE + K + k * D;
      const float* lv = mu + K * D;
      const size_t v = (size_t)r * K + k;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int d = q * 4 + e;
        if (d >= D) break;
        const float eps = a.eps_in ? a.eps_in[v * D + d] : ev[e];
        const float lvc = fminf(fmaxf(lv[d], kLogvarMin), kLogvarMax);
        a.z_all[v * a.ldD + d] = mu[d] + eps * __expf(0.5f * lvc);
        a.eps_buf[v * a.ldD + d] = eps;
      }
    }
    if (a.latent_out) {
      __syncthreads();
      for (int idx = tid; idx < nrows * D; idx += 256) {
        const int lr = idx / D, d = idx % D;
        const int r = r0 + lr;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += w_s[lr * SUAVE_MAX_COMPONENTS + k] * a.z_all[((size_t)r * K + k) * a.ldD + d];
        a.latent_out[(size_t)r * D + d] = acc;
      }
    }
  }
}

// Expectation over the components: reconstruction = sum_k w_k log p_k, mixture KL (losses.py:60-84),
// categorical KL, and (GRAD) the gradients w.r.t. the encoder head and the PriorMean layer.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_mix_finalize(MixFinArgs a) {
  __shared__ float w_s[64 * SUAVE_MAX_COMPONENTS];
  __shared__ float kl_s[64 * SUAVE_MAX_COMPONENTS];
  __shared__ float red_s[3 * 64];
  const int tid = threadIdx.x;
  const int K = a.K, D = a.D;
  double s_rec = 0.0, s_cat = 0.0, s_gau = 0.0;
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, a.n_rows - r0);
    __syncthreads();
    for (int idx = tid; idx < nrows * K; idx += 256) {
      const int lr = idx / K, k = idx % K;
      w_s[lr * SUAVE_MAX_COMPONENTS + k] = a.w[(size_t)(r0 + lr) * a.ldK + k];
      kl_s[lr * SUAVE_MAX_COMPONENTS + k] = 0.f;
    }
    __syncthreads();
    // per (row, component, latent dim): KL term and gradients of mu / logvar; KL summed over d per (row, k)
    for (int idx = tid; idx < nrows * K; idx += 256) {
      const int lr = idx / K, k = idx % K;
      const int r = r0 + lr;
      const float* mu = a.encout + (size_t)r * a.ldE + K + k * D;
      const float* lv = mu + K * D;
      const float wk = w_s[lr * SUAVE_MAX_COMPONENTS + k];
      const size_t v = (size_t)r * K + k;
      float kl = 0.f;
      for (int d = 0; d < D; ++d) {
        const float pm = a.params[a.pm_w_off + d * K + k] + a.params[a.pm_b_off + d];  // prior.py:31-39
        const float raw = lv[d];
        const float lvc = fminf(fmaxf(raw, kLogvarMin), kLogvarMax);
        const float diff = mu[d] - pm;
        const float elv = __expf(lvc);
        kl += 0.5f * (-lvc + elv + diff * diff - 1.0f);  // prior logvar == 0 (model.py:541)
        if (GRAD) {
          const float dz = a.dz_all[v * a.ldD + d];
          const float eps = a.eps_buf[v * a.ldD + d];
          const float gm = a.bk * wk * diff;
          a.dencout[(size_t)r * a.ldE + K + k * D + d] = dz + gm;
          const bool in_range = raw >= kLogvarMin && raw <= kLogvarMax;
          a.dencout[(size_t)r * a.ldE + K + K * D + k * D + d] =
              in_range ? dz * eps * 0.5f * __expf(0.5f * lvc) + a.bk * wk * 0.5f * (elv - 1.0f) : 0.f;
          a.prior_terms[(size_t)r * a.ldT + k * D + d] = -gm;
        }
      }
      kl_s[lr * SUAVE_MAX_COMPONENTS + k] = kl;
    }
    __syncthreads();
    if (tid < nrows) {
      const int r = r0 + tid;
      const float* lg = a.encout + (size_t)r * a.ldE;
      const float* ws = w_s + tid * SUAVE_MAX_COMPONENTS;
      float lp[SUAVE_MAX_COMPONENTS];
      float rec = 0.f, gk = 0.f;
#pragma unroll 1
      for (int k = 0; k < K; ++k) {
        float t = 0.f;
        for (int c = 0; c < a.n_chunks; ++c) t += a.recon_part[(size_t)c * a.n_rows * K + (size_t)r * K + k];
        lp[k] = t;
        rec += ws[k] * t;
        gk += ws[k] * kl_s[tid * SUAVE_MAX_COMPONENTS + k];
      }
      float mq = -INFINITY;
      for (int k = 0; k < K; ++k) mq = fmaxf(mq, lg[k]);
      float sq = 0.f;
      for (int k = 0; k < K; ++k) sq += __expf(lg[k] - mq);
      const float lse = mq + __logf(sq);
      const float logp = -__logf((float)K);  // prior logits == 0
      float ck = 0.f;
      for (int k = 0; k < K; ++k) { const float lq = lg[k] - lse; ck += __expf(lq) * (lq - logp); }
      a.recon[r] = rec;
      a.catkl[r] = ck;
      a.gausskl[r] = gk;
      red_s[tid] = rec; red_s[64 + tid] = ck; red_s[128 + tid] = gk;
      if (GRAD) {
        // d loss / d w_k = c * lp_k + bk * KL_k, through the (relaxed Gumbel-)softmax Jacobian
        float inner = 0.f;
        float dw[SUAVE_MAX_COMPONENTS];
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
          dw[k] = a.c_up * lp[k] + a.bk * kl_s[tid * SUAVE_MAX_COMPONENTS + k];
          inner += ws[k] * dw[k];
        }
        const float inv_tau = a.tau > 0.f ? 1.0f / a.tau : 1.0f;
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
          const float lq = lg[k] - lse;
          const float q = __expf(lq);
          a.dencout[(size_t)r * a.ldE + k] = ws[k] * (dw[k] - inner) * inv_tau + a.bk * q * ((lq - logp) - ck);
        }
      }
    } else if (tid < 64) {
      red_s[tid] = 0.f; red_s[64 + tid] = 0.f; red_s[128 + tid] = 0.f;
    }
    if (GRAD) {
      // pad columns of d(enc_out) and the bias part of the prior-mean gradient (sum over k)
      const int E = K + 2 * K * D;
      for (int idx = tid; idx < nrows * (a.ldE - E); idx += 256)
        a.dencout[(size_t)(r0 + idx / (a.ldE - E)) * a.ldE + E + idx % (a.ldE - E)] = 0.f;
      __syncthreads();
      for (int idx = tid; idx < nrows * D; idx += 256) {
        const int lr = idx / D, d = idx % D;
        float t = 0.f;
        for (int k = 0; k < K; ++k) t += a.prior_terms[(size_t)(r0 + lr) * a.ldT + k * D + d];
        a.prior_terms[(size_t)(r0 + lr) * a.ldT + K * D + d] = t;
      }
    }
    __syncthreads();
    if (tid == 0)
      for (int i = 0; i < 64; ++i) { s_rec += red_s[i]; s_cat += red_s[64 + i]; s_gau += red_s[128 + i]; }
  }
  if (tid == 0 && a.sums_slab) {
    double* s = a.sums_slab + (size_t)blockIdx.x * 4;
    s[0] = s_rec; s[1] = s_cat; s[2] = s_gau; s[3] = 0.0;
  }
}

// column sums of a [n_rows, n_cols] matrix into per-block slabs (fixed-order reduce follows)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ src, int ld, int n_rows, int n_cols,
                                                float* __restrict__ slab) {
  const int rows_per = (n_rows + gridDim.x - 1) / gridDim.x;
  const int rb = blockIdx.x * rows_per, re = min(n_rows, rb + rows_per);
  for (int c = threadIdx.x; c < n_cols; c += 256) {
    float t = 0.f;
    for (int r = rb; r < re; ++r) t += src[(size_t)r * ld + c];
    slab[(size_t)blockIdx.x * n_cols + c] = t;
  }
}

// ---- host launchers -----------------------------------------------------------
int launch_mix_fwd(const MixFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  hipLaunchKernelGGL(k_mix_fwd, dim3(std::min((a.n_rows + 63) / 64, 2048)), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_mix_finalize(const MixFinArgs& a, bool grad, int grid, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  if (grad) hipLaunchKernelGGL((k_mix_finalize<true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_mix_finalize<false>), dim3(grid), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_colsum(const float* src, int ld, int n_rows, int n_cols, float* slab, int n_blocks, hipStream_t s) {
  ProfScope prof_scope(kProfReduce, s);
  hipLaunchKernelGGL(k_colsum, dim3(n_blocks), dim3(256), 0, s, src, ld, n_rows, n_cols, slab);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_expand(const suave_plan* p, const int32_t* enc_src_dev, const suave_batch_t& b, float* x, hipStream_t s) {
  if (b.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfExpand, s);
  const int grid = std::min((b.n_rows + 3) / 4, 4096);
  hipLaunchKernelGGL(k_expand, dim3(grid), dim3(256), 0, s, enc_src_dev, p->cols_dev, b, p->ldI, x);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_latent_fwd(const LatentFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  const int grid = std::min((a.n_rows + 63) / 64, 2048);
  ProfScope prof_scope(kProfLatent, s);
  hipLaunchKernelGGL(k_latent_fwd, dim3(grid), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_latent_bwd(const LatentBwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  const int grid = std::min((a.n_rows + 63) / 64, 2048);
  ProfScope prof_scope(kProfLatent, s);
  hipLaunchKernelGGL(k_latent_bwd, dim3(grid), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_posterior_stats(const float* encout, int ldE, int K, int D, int n_rows, float inv_temp, float* enc_out_user,
                           float* post_mean, float* post_logvar, float* probs, hipStream_t s) {
  if (n_rows <= 0) return 0;
  const int grid = std::min((n_rows + 63) / 64, 2048);
  ProfScope prof_scope(kProfLatent, s);
  hipLaunchKernelGGL(k_posterior_stats, dim3(grid), dim3(256), 0, s, encout, ldE, K, D, n_rows, inv_temp, enc_out_user,
                     post_mean, post_logvar, probs);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
