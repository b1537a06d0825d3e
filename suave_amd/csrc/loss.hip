// Fused, schema-segmented, mask-aware likelihood + KL kernel (forward and backward).
//
// One launch replaces the reference's per-column Python loop
// (suave/modules/decoder.py:348-379), every head module (decoder.py:56-265),
// the distribution math (suave/modules/distributions.py:42-164), the masked
// `_apply_observed_linear` scatter (decoder.py:41-53), the reconstruction sum
// (suave/modules/losses.py:29-32), both KL terms (losses.py:51-57, :87-97) and,
// with GRAD, the autograd backward of all of them (formulas: SURVEY.md
// Appendix A, each verified against the reference's autograd).
//
// Geometry: a block owns a tile of 64 rows x one chunk of consecutive schema
// columns.  The tile's slice of y_full is staged in LDS with coalesced float4
// loads; then each of the 4 wavefronts walks its share of the chunk's columns
// with LANE = ROW, so the column's type, class count, normalisation statistics
// and head weights are wave-uniform (scalar loads, no divergence between
// lanes).  dY overwrites y in LDS and is written back coalesced.  Per-row
// terms are reduced across the 4 waves through LDS; head-weight gradients are
// reduced across the 64 rows with DPP wavefront reductions, accumulated per
// block in LDS over the block's tiles, and flushed once to a slab that a
// fixed-order reduction sums (deterministic, no float atomics).
#include "kernels.h"

namespace suave {


namespace {

// Sum over the 64 lanes with DPP row operations; result is wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ void acc_add(float* hw, int idx, float v, int lane) {
  if (lane == 0) hw[idx] += v;
}

}  // namespace

template <bool GRAD, int KMAX>
__global__ __launch_bounds__(256, 2) void k_loss(LossArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = a.K;
  float* ytile = lds;                         // [64][ldy]
  float* ds_w = ytile + 64 * a.ldy;           // [4][64][K]
  float* recon_w = ds_w + 4 * 64 * K;         // [4][64]
  float* hw_acc = recon_w + 4 * 64;           // [hw_max]
  float* gs = hw_acc + a.hw_max;              // [4][max_classes][64]
  int* k_s = reinterpret_cast<int*>(gs + 4 * a.max_classes * 64);  // [64]
  float* up_s = reinterpret_cast<float*>(k_s + 64);                // [64]
  float* red_s = up_s + 64;                   // [16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk_id = blockIdx.y;
  const ColChunk ch = a.chunks[chunk_id];
  const int hw_begin = a.cols[ch.col_begin].w_off;
  const int n_tiles = (a.n_rows + 63) / 64;
  const int yq = (ch.y_len + 3) / 4;  // float4 per row of the chunk (ldY % 4 == 0, y_begin % 4 handled below)
  double sum_recon = 0.0, sum_cat = 0.0, sum_gauss = 0.0;

  if (GRAD)
    for (int i = tid; i < a.hw_max; i += 256) hw_acc[i] = 0.f;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int r0 = tile * 64;
    const int nrows = min(64, a.n_rows - r0);
    __syncthreads();
    // ---- stage the y tile (coalesced) ----
    {
      const int ybase = ch.y_begin & ~3;              // aligned start column
      const int lead = ch.y_begin - ybase;
      const int nq = (lead + ch.y_len + 3) / 4;
      for (int idx = tid; idx < nrows * nq; idx += 256) {
        const int lr = idx / nq, q = idx % nq;
        const float4 v = *reinterpret_cast<const float4*>(a.y + (size_t)(r0 + lr) * a.ldY + ybase + q * 4);
        const int c0 = q * 4 - lead;
        float* dst = ytile + lr * a.ldy;
        if (c0 >= 0 && c0 < ch.y_len) dst[c0] = v.x;
        if (c0 + 1 >= 0 && c0 + 1 < ch.y_len) dst[c0 + 1] = v.y;
        if (c0 + 2 >= 0 && c0 + 2 < ch.y_len) dst[c0 + 2] = v.z;
        if (c0 + 3 >= 0 && c0 + 3 < ch.y_len) dst[c0 + 3] = v.w;
      }
      (void)yq;
      for (int idx = tid; idx < (64 - nrows) * ch.y_len; idx += 256)  // rows past the table: finite zeros
        ytile[(nrows + idx / ch.y_len) * a.ldy + idx % ch.y_len] = 0.f;
      if (tid < 64) {
        const int r = r0 + tid;
        int k = 0;
        float up = 0.f;
        if (tid < nrows) {
          k = a.kidx ? a.kidx[r] : (r % a.vdiv);
          up = a.c_up * (a.row_weight ? a.row_weight[r] : 1.0f);
        }
        k_s[tid] = k;
        up_s[tid] = up;
      }
    }
    __syncthreads();

    // ---- column phase: lane = row, wave-uniform column ----
    const bool valid = lane < nrows;
    const int r = r0 + (valid ? lane : 0);
    const int64_t drow_b = (int64_t)(r / a.vdiv);
    const int64_t tr = a.batch.row_index ? (int64_t)a.batch.row_index[drow_b] : a.batch.row_begin + drow_b;
    const float* xn = a.batch.xnum + tr * a.batch.ld_num;
    const int16_t* cd = a.batch.codes + tr * a.batch.ld_disc;
    const uint8_t* mk = a.batch.mask + tr * a.batch.ld_mask;
    const int k = k_s[lane];
    const float up = up_s[lane];
    float recon = 0.f;
    float ds[KMAX];
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) ds[j] = 0.f;
    }
    float* yrow = ytile + lane * a.ldy;
    float* gsw = gs + wave * a.max_classes * 64 + lane;

    for (int c = ch.col_begin + wave; c < ch.col_end; c += 4) {
      const suave_col_t col = a.cols[c];
      const bool obs = valid && mk[c] != 0;
      const float u = obs ? up : 0.f;
      float* yb = yrow + (col.y_off - ch.y_begin);
      const int yd = col.y_dim;
      const float* w = a.params + col.w_off;
      float* hw = hw_acc + (col.w_off - hw_begin);
      if (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) {
        const float* wv = a.params + col.w2_off;
        float* hwv = hw_acc + (col.w2_off - hw_begin);
        float am = w[yd + k];
        for (int i = 0; i < yd; ++i) am += w[i] * yb[i];
        const float bv = wv[k];
        const float x = xn[col.x_index];
        const float v = softplus_f(bv) + kEps;
        const float mean = am * col.scale + col.shift;
        const float vs = v * (col.scale * col.scale);
        const float xr = x * col.scale + col.shift;
        const float cc = fmaxf(vs, kEps);
        const float rr = xr - mean;
        float nll = 0.5f * (kLog2Pi + __logf(cc) + rr * rr / cc);
        if (col.type == SUAVE_TYPE_POS) nll += xr;
        recon += obs ? -nll : 0.f;
        if (GRAD) {
          // masked / out-of-tile lanes get an exact zero (select, not 0 * x: x may be inf or NaN there)
          const float ga = obs ? u * (rr / cc) * col.scale : 0.f;
          const float gb = (obs && vs >= kEps) ? u * (-0.5f) * (1.0f / cc - rr * rr / (cc * cc)) *
                                                      (col.scale * col.scale) * softplus_grad_f(bv)
                                                : 0.f;
          for (int i = 0; i < yd; ++i) {
            const float t = wave_sum_dpp(ga * yb[i]);
            acc_add(hw, i, t, lane);
            yb[i] = ga * w[i];
          }
#pragma unroll
          for (int j = 0; j < KMAX; ++j) {
            if (j < K) {
              ds[j] += ga * w[yd + j] + gb * wv[j];
              const float ta = wave_sum_dpp(k == j ? ga : 0.f);
              const float tb = wave_sum_dpp(k == j ? gb : 0.f);
              acc_add(hw, yd + j, ta, lane);
              acc_add(hwv, j, tb, lane);
            }
          }
        }
      } else if (col.type == SUAVE_TYPE_COUNT) {
        float rho = w[yd + k];
        for (int i = 0; i < yd; ++i) rho += w[i] * yb[i];
        const float x = xn[col.x_index];
        const float rate = softplus_f(rho) + kEps;
        const float cnt = fmaxf(__expf(x) - col.shift, 0.f);
        const float rc = fmaxf(rate, kEps);
        const float lp = (cnt == 0.f ? 0.f : cnt * __logf(rc)) - rc - lgammaf(cnt + 1.0f);
        recon += obs ? lp : 0.f;
        if (GRAD) {
          const float gr = obs ? u * (cnt / rc - 1.0f) * softplus_grad_f(rho) : 0.f;
          for (int i = 0; i < yd; ++i) {
            const float t = wave_sum_dpp(gr * yb[i]);
            acc_add(hw, i, t, lane);
            yb[i] = gr * w[i];
          }
#pragma unroll
          for (int j = 0; j < KMAX; ++j) {
            if (j < K) {
              ds[j] += gr * w[yd + j];
              const float t = wave_sum_dpp(k == j ? gr : 0.f);
              acc_add(hw, yd + j, t, lane);
            }
          }
        }
      } else if (col.type == SUAVE_TYPE_CAT) {
        const int C = col.n_classes;
        const int ldw = yd + K;
        const int code = valid ? (int)cd[col.x_index] : -1;
        // sweep 1: logits, online log-sum-exp (logit 0 is pinned to 0)
        float mx = 0.f, se = 1.f, lcode = 0.f;
        for (int cl = 1; cl < C; ++cl) {
          const float* wr = w + (cl - 1) * ldw;
          float L = wr[yd + k];
          for (int i = 0; i < yd; ++i) L += wr[i] * yb[i];
          gsw[cl * 64] = L;
          if (L > mx) { se = se * __expf(mx - L) + 1.0f; mx = L; }
          else se += __expf(L - mx);
          if (cl == code) lcode = L;
        }
        const float lse = mx + __logf(se);
        const bool has = code >= 0 && code < C;
        recon += (obs && has) ? (lcode - lse) : 0.f;
        if (GRAD) {
          const bool live = obs && has;     // all-zero one-hot: sum(x) = 0 -> zero gradient
          // sweep 2: g_c = u * (x_c - softmax_c); weight gradients while y is intact
          for (int cl = 1; cl < C; ++cl) {
            const float p = __expf(gsw[cl * 64] - lse);
            const float g = live ? u * ((cl == code ? 1.0f : 0.f) - p) : 0.f;
            gsw[cl * 64] = g;
            float* hr = hw + (cl - 1) * ldw;
            for (int i = 0; i < yd; ++i) {
              const float t = wave_sum_dpp(g * yb[i]);
              acc_add(hr, i, t, lane);
            }
            const float* wr = w + (cl - 1) * ldw;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              if (j < K) {
                ds[j] += g * wr[yd + j];
                const float t = wave_sum_dpp(k == j ? g : 0.f);
                acc_add(hr, yd + j, t, lane);
              }
            }
          }
          // sweep 3: dY
          for (int i = 0; i < yd; ++i) {
            float dy = 0.f;
            for (int cl = 1; cl < C; ++cl) dy += gsw[cl * 64] * w[(cl - 1) * ldw + i];
            yb[i] = dy;
          }
        }
      } else {  // ordinal: cumulative link
        const int C = col.n_classes;
        const float* wm = a.params + col.w2_off;  // mean_layer [yd + K]
        float* hwm = hw_acc + (col.w2_off - hw_begin);
        const int code = valid ? (int)cd[col.x_index] : -1;
        const int t = max(code, 0);  // target = max(round(sum thermo) - 1, 0)
        float mo = wm[yd + k];
        for (int i = 0; i < yd; ++i) mo += wm[i] * yb[i];
        float thr = 0.f, sg_t = 1.0f, sg_tm1 = 0.f;
        for (int j = 0; j < C - 1; ++j) {
          thr += softplus_f(w[j * K + k]) + kEps;
          const float sgj = sigmoid_f(thr - mo);
          if (j == t) sg_t = sgj;
          if (j == t - 1) sg_tm1 = sgj;
        }
        const float p_raw = sg_t - sg_tm1;
        const float p = fminf(fmaxf(p_raw, kEps), 1.0f);
        recon += obs ? __logf(p) : 0.f;
        if (GRAD) {
          const float gp = (obs && p_raw >= kEps && p_raw <= 1.0f) ? u / p : 0.f;
          const float du_t = (gp != 0.f && t < C - 1) ? gp * sg_t * (1.0f - sg_t) : 0.f;
          const float du_tm1 = (gp != 0.f && t > 0) ? -gp * sg_tm1 * (1.0f - sg_tm1) : 0.f;
          const float gmean = -(du_t + du_tm1);
          for (int i = 0; i < yd; ++i) {
            const float s = wave_sum_dpp(gmean * yb[i]);
            acc_add(hwm, i, s, lane);
            yb[i] = gmean * wm[i];
          }
#pragma unroll
          for (int j = 0; j < KMAX; ++j) {
            if (j < K) {
              ds[j] += gmean * wm[yd + j];
              const float s = wave_sum_dpp(k == j ? gmean : 0.f);
              acc_add(hwm, yd + j, s, lane);
            }
          }
          for (int jj = 0; jj < C - 1; ++jj) {
            const float dsp = (jj <= t - 1 ? du_tm1 : 0.f) + (jj <= t ? du_t : 0.f);
            const float gpart = dsp * softplus_grad_f(w[jj * K + k]);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              if (j < K) {
                ds[j] += gpart * w[jj * K + j];
                const float s = wave_sum_dpp(k == j ? gpart : 0.f);
                acc_add(hw, jj * K + j, s, lane);
              }
            }
          }
        }
      }
    }

    // ---- cross-wave reduction of the per-row terms ----
    recon_w[wave * 64 + lane] = recon;
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) ds_w[(wave * 64 + lane) * K + j] = ds[j];
    }
    __syncthreads();
    if (tid < 64) {
      const float rsum = recon_w[tid] + recon_w[64 + tid] + recon_w[128 + tid] + recon_w[192 + tid];
      if (tid < nrows) a.recon_part[(size_t)chunk_id * a.n_rows + r0 + tid] = rsum;
      const float tot = wave_sum_dpp(tid < nrows ? rsum : 0.f);
      if (tid == 0) sum_recon += (double)tot;
    }
    if (GRAD && a.ds_part) {
      for (int idx = tid; idx < nrows * K; idx += 256) {
        const int lr = idx / K, j = idx % K;
        const float v = ds_w[(0 * 64 + lr) * K + j] + ds_w[(1 * 64 + lr) * K + j] + ds_w[(2 * 64 + lr) * K + j] +
                        ds_w[(3 * 64 + lr) * K + j];
        a.ds_part[((size_t)chunk_id * a.n_rows + r0 + lr) * a.ldK + j] = v;
      }
    }
    // ---- write dY back (coalesced) ----
    if (GRAD) {
      for (int idx = tid; idx < nrows * ch.y_len; idx += 256) {
        const int lr = idx / ch.y_len, cc = idx % ch.y_len;
        a.y[(size_t)(r0 + lr) * a.ldY + ch.y_begin + cc] = ytile[lr * a.ldy + cc];
      }
    }

    // ---- KL phase (chunk 0 only): 4 lanes per row ----
    if (a.do_kl && chunk_id == 0) {
      const int lr = tid >> 2, qd = tid & 3;
      const bool v4 = lr < nrows;
      const int rr = r0 + (v4 ? lr : 0);
      const int kk = k_s[lr];
      const int D = a.D;
      const float* mu = a.encout + (size_t)rr * a.ldE + K + kk * D;
      const float* lv = mu + K * D;
      const float* pm = a.params + a.prior_mu_off + kk * D;
      const float* plv = a.params + a.prior_logvar_off + kk * D;
      float gk = 0.f;
      if (v4) {
        for (int d = qd; d < D; d += 4) {
          const float lvc = fminf(fmaxf(lv[d], kLogvarMin), kLogvarMax);
          const float vp_raw = __expf(plv[d]);
          const float vp = fmaxf(vp_raw, 1e-6f);
          const float diff = mu[d] - pm[d];
          const float elv = __expf(lvc);
          gk += 0.5f * (plv[d] - lvc + (elv + diff * diff) / vp - 1.0f);
          if (GRAD) {
            a.gkl_mu[(size_t)rr * a.ldD + d] = a.bk * diff / vp;
            a.gkl_lv[(size_t)rr * a.ldD + d] = a.bk * 0.5f * (elv / vp - 1.0f);
            a.prior_terms[(size_t)rr * a.ldT + d] = -a.bk * diff / vp;
            a.prior_terms[(size_t)rr * a.ldT + D + d] =
                a.bk * 0.5f * (1.0f - (vp_raw >= 1e-6f ? (elv + diff * diff) / vp : 0.f));
          }
        }
      }
      gk += __shfl_xor(gk, 1, 64);
      gk += __shfl_xor(gk, 2, 64);
      float ck = 0.f;
      if (v4 && qd == 0) {
        const float* lg = a.encout + (size_t)rr * a.ldE;
        const float* pl = a.params + a.prior_logits_off;
        float mq = -INFINITY, mp = -INFINITY;
        for (int j = 0; j < K; ++j) { mq = fmaxf(mq, lg[j]); mp = fmaxf(mp, pl[j]); }
        float sq = 0.f, sp = 0.f;
        for (int j = 0; j < K; ++j) { sq += __expf(lg[j] - mq); sp += __expf(pl[j] - mp); }
        const float lq = mq + __logf(sq), lp = mp + __logf(sp);
        for (int j = 0; j < K; ++j) {
          const float logq = lg[j] - lq, logp = pl[j] - lp;
          ck += __expf(logq) * (logq - logp);
        }
        if (GRAD) {
          for (int j = 0; j < K; ++j) {
            const float logq = lg[j] - lq, logp = pl[j] - lp;
            const float q = __expf(logq);
            a.gkl_logit[(size_t)rr * a.ldK + j] = a.bk * q * ((logq - logp) - ck);
            a.prior_terms[(size_t)rr * a.ldT + 2 * D + j] = a.bk * (__expf(logp) - q);
          }
        }
        a.catkl[rr] = ck;
        a.gausskl[rr] = gk;
      }
      const float t_c = wave_sum_dpp((v4 && qd == 0) ? ck : 0.f);
      const float t_g = wave_sum_dpp((v4 && qd == 0) ? gk : 0.f);
      if (lane == 0) { red_s[wave * 2] = t_c; red_s[wave * 2 + 1] = t_g; }
      __syncthreads();
      if (tid == 0) {
        sum_cat += (double)red_s[0] + (double)red_s[2] + (double)red_s[4] + (double)red_s[6];
        sum_gauss += (double)red_s[1] + (double)red_s[3] + (double)red_s[5] + (double)red_s[7];
      }
    }
  }

  __syncthreads();
  if (GRAD && a.head_slab) {
    const int len = a.hw_max;  // this chunk's head range is [hw_begin, hw_begin + chunk length)
    const int next = (chunk_id + 1 < (int)gridDim.y) ? a.cols[a.chunks[chunk_id + 1].col_begin].w_off
                                                      : a.head_off + a.head_len;
    const int n = min(len, next - hw_begin);
    float* dst = a.head_slab + (size_t)blockIdx.x * a.head_len + (hw_begin - a.head_off);
    for (int i = tid; i < n; i += 256) dst[i] = hw_acc[i];
  }
  if (tid == 0 && a.sums_slab) {
    double* s = a.sums_slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    s[0] = sum_recon;
    s[1] = sum_cat;
    s[2] = sum_gauss;
    s[3] = 0.0;
  }
}

size_t loss_lds_bytes(int ldy, int K, int hw_max, int max_classes) {
  size_t floats = (size_t)64 * ldy + (size_t)4 * 64 * K + 4 * 64 + hw_max + (size_t)4 * max_classes * 64 + 64 + 64 + 16;
  return floats * sizeof(float);
}

template <int KMAX>
static int launch_loss_k(const LossArgs& a, bool grad, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  if (grad)
    hipLaunchKernelGGL((k_loss<true, KMAX>), dim3(grid_x, n_chunks), dim3(256), lds_bytes, s, a);
  else
    hipLaunchKernelGGL((k_loss<false, KMAX>), dim3(grid_x, n_chunks), dim3(256), lds_bytes, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_loss(const LossArgs& a, bool grad, int kmax, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLoss, s);
  if (kmax <= 8) return launch_loss_k<8>(a, grad, grid_x, n_chunks, lds_bytes, s);
  if (kmax <= 16) return launch_loss_k<16>(a, grad, grid_x, n_chunks, lds_bytes, s);
  return launch_loss_k<32>(a, grad, grid_x, n_chunks, lds_bytes, s);
}

template <bool GRAD, int KMAX>
static int set_lds_one(size_t lds_bytes) {
  SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loss<GRAD, KMAX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  return 0;
}

int loss_set_lds_limit(size_t lds_bytes) {
  int rc = 0;
  rc |= set_lds_one<true, 8>(lds_bytes);
  rc |= set_lds_one<false, 8>(lds_bytes);
  rc |= set_lds_one<true, 16>(lds_bytes);
  rc |= set_lds_one<false, 16>(lds_bytes);
  rc |= set_lds_one<true, 32>(lds_bytes);
  rc |= set_lds_one<false, 32>(lds_bytes);
  return rc;
}

}  // namespace suave
