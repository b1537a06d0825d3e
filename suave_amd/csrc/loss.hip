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

constexpr int kMissing = 0x7FC0DEAD;  // cell sentinel: masked / unobserved
// 8 wavefronts share one 64-row tile (each walks every 8th column): two blocks per CU put 4 waves on every
// SIMD, which is what hides the LDS / transcendental latency of the per-column chains.
constexpr int kLossWaves = 8;
constexpr int kLossThreads = kLossWaves * 64;

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

// Fire-and-forget LDS add (ds_add_f32, no return value to wait for).  Every index is only ever
// touched by one wavefront, in program order, so the sum order is fixed (deterministic).
__device__ __forceinline__ void acc_add(float* hw, int idx, float v, int lane) {
  if (lane == 0) (void)__hip_atomic_fetch_add(hw + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Per-lane LDS add: lane r adds its value at the slot of its own component.  Lanes that share a
// component collide on one address and are serialised by the LDS unit in a fixed order, and no other
// wavefront ever touches these slots, so the result is reproducible run to run.
__device__ __forceinline__ void acc_add_lanes(float* slot, float v) {
  (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

#ifdef SUAVE_LOSS_TIMING
__device__ unsigned long long g_loss_dbg[8];
#define LOSS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define LOSS_ACC(i, v) dbg_acc[i] += (v)
#else
#define LOSS_T(var)
#define LOSS_ACC(i, v)
#endif

template <bool GRAD, int KMAX>
__global__ __launch_bounds__(kLossThreads, 4) void k_loss(LossArgs a) {
#ifdef SUAVE_LOSS_TIMING
  unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = a.K;
  float* ytile = lds;                         // [64][ldy]
  float* hw_acc = ytile + 64 * a.ldy;         // [hw_max]
  float* wsm = hw_acc + a.hw_max;             // [hw_max] this chunk's head weights
  int* k_s = reinterpret_cast<int*>(wsm + a.hw_max);               // [64]
  float* up_s = reinterpret_cast<float*>(k_s + 64);                // [64]
  float* red_s = up_s + 64;                   // [16]
  long long* trow_s = reinterpret_cast<long long*>(red_s + 16);    // [64] table row of each tile row
  int* xsrc_s = reinterpret_cast<int*>(trow_s + 64);               // [ldx] cell source of each chunk column
  float* gs = reinterpret_cast<float*>(xsrc_s + a.ldx);            // [waves][max_classes][64]
  int* xs = reinterpret_cast<int*>(gs + kLossWaves * a.max_classes * 64);  // [64][ldx] cell bits, kMissing when masked
  float* ds_w = gs;                           // [waves][64][K]  (aliases gs + xs: used after the column phase)
  float* recon_w = ds_w + kLossWaves * 64 * K;  // [waves][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk_id = blockIdx.y;
  const ColChunk ch = a.chunks[chunk_id];
  const int hw_begin = a.cols[ch.col_begin].w_off;
  const int n_tiles = (a.n_rows + 63) / 64;
  const int yq = (ch.y_len + 3) / 4;  // float4 per row of the chunk (ldY % 4 == 0, y_begin % 4 handled below)
  double sum_recon = 0.0, sum_cat = 0.0, sum_gauss = 0.0;

  const int ncc = ch.col_end - ch.col_begin;
  {
    const int hw_next = (chunk_id + 1 < (int)gridDim.y) ? a.cols[a.chunks[chunk_id + 1].col_begin].w_off
                                                        : a.head_off + a.head_len;
    const int hw_len = min(a.hw_max, hw_next - hw_begin);
    for (int i = tid; i < hw_len; i += kLossThreads) wsm[i] = a.params[hw_begin + i];
    for (int i = tid; i < ncc; i += kLossThreads) xsrc_s[i] = a.xsrc[ch.col_begin + i];
    if (GRAD)
      for (int i = tid; i < a.hw_max; i += kLossThreads) hw_acc[i] = 0.f;
  }

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int r0 = tile * 64;
    const int nrows = min(64, a.n_rows - r0);
    __syncthreads();
    LOSS_T(t0);
    // ---- stage the y tile (coalesced) ----
    {
      const int ybase = ch.y_begin & ~3;              // aligned start column
      const int lead = ch.y_begin - ybase;
      const int nq = (lead + ch.y_len + 3) / 4;
      // wave = row, lane = float4 of the row; 4 rows of loads are issued before the first LDS store
      for (int lr0 = wave; lr0 < 64; lr0 += 4 * kLossWaves) {
        for (int q0 = 0; q0 < nq; q0 += 64) {
          const int q = q0 + lane;
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int lr = lr0 + kLossWaves * u;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the table: finite zeros
            if (q < nq && lr < nrows) v[u] = *reinterpret_cast<const float4*>(a.y + (size_t)(r0 + lr) * a.ldY + ybase + q * 4);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int lr = lr0 + kLossWaves * u;
            if (q < nq && lr < 64) {
              float* dst = ytile + lr * a.ldy;
              const int c0 = q * 4 - lead;
              if (c0 >= 0 && c0 < ch.y_len) dst[c0] = v[u].x;
              if (c0 + 1 >= 0 && c0 + 1 < ch.y_len) dst[c0 + 1] = v[u].y;
              if (c0 + 2 >= 0 && c0 + 2 < ch.y_len) dst[c0 + 2] = v[u].z;
              if (c0 + 3 >= 0 && c0 + 3 < ch.y_len) dst[c0 + 3] = v[u].w;
            }
          }
        }
      }
      (void)yq;
      if (tid < 64) {
        const int r = r0 + tid;
        int k = 0;
        float up = 0.f;
        long long trow = 0;
        if (tid < nrows) {
          k = a.kidx ? a.kidx[r] : (r % a.vdiv);
          up = a.c_up * (a.row_weight ? a.row_weight[r] : 1.0f);
          const int64_t drow = (int64_t)(r / a.vdiv);
          trow = a.batch.row_index ? (long long)a.batch.row_index[drow] : (long long)(a.batch.row_begin + drow);
        }
        k_s[tid] = k;
        up_s[tid] = up;
        trow_s[tid] = trow;
      }
    }
    __syncthreads();
    LOSS_T(t1);
    // stage the tile's cells (mask folded in): wave = row, lane = column, 4 rows of loads in flight
    for (int j0 = 0; j0 < ncc; j0 += 64) {
      const int cc = j0 + lane;
      const bool cok = cc < ncc;
      const int src = cok ? xsrc_s[cc] : 0;
      for (int lr0 = wave; lr0 < nrows; lr0 += 4 * kLossWaves) {
        int v[4];
        unsigned char mk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int lr = lr0 + kLossWaves * u;
          v[u] = 0;
          mk[u] = 0;
          if (cok && lr < nrows) {
            const long long trow = trow_s[lr];
            if (src & (1 << 30)) v[u] = (int)a.batch.codes[trow * a.batch.ld_disc + (src & 0xFFFFFF)];
            else v[u] = __float_as_int(a.batch.xnum[trow * a.batch.ld_num + src]);
            mk[u] = a.batch.mask[trow * a.batch.ld_mask + ch.col_begin + cc];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int lr = lr0 + kLossWaves * u;
          if (cok && lr < nrows) xs[lr * a.ldx + cc] = mk[u] ? v[u] : kMissing;
        }
      }
    }
    __syncthreads();
    LOSS_T(t2);

    // ---- column phase: lane = row, wave-uniform column ----
    const bool valid = lane < nrows;
    const int r = r0 + (valid ? lane : 0);
    (void)r;
    const int* xrow = xs + lane * a.ldx - ch.col_begin;
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

    for (int c = ch.col_begin + wave; c < ch.col_end; c += kLossWaves) {
      const suave_col_t col = a.cols[c];
      const int cell = valid ? xrow[c] : kMissing;
      const bool obs = cell != kMissing;
      const float u = obs ? up : 0.f;
      float* yb = yrow + (col.y_off - ch.y_begin);
      const int yd = col.y_dim;
      // head weights come from the block's LDS copy: a dependent scalar load from global per weight
      // (s_load + s_waitcnt) costs hundreds of cycles each and serialises the column loop
      const float* wl = wsm + (col.w_off - hw_begin);
      const float* w = wl;
      float* hw = hw_acc + (col.w_off - hw_begin);
      if (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) {
        const float* wvl = wsm + (col.w2_off - hw_begin);
        const float* wv = wvl;
        float* hwv = hw_acc + (col.w2_off - hw_begin);
        float am = wl[yd + k];
        for (int i = 0; i < yd; ++i) am += w[i] * yb[i];
        const float bv = wvl[k];
        const float x = obs ? __int_as_float(cell) : 0.f;
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
            }
          }
          acc_add_lanes(hw + yd + k, ga);
          acc_add_lanes(hwv + k, gb);
        }
      } else if (col.type == SUAVE_TYPE_COUNT) {
        float rho = wl[yd + k];
        for (int i = 0; i < yd; ++i) rho += w[i] * yb[i];
        const float x = obs ? __int_as_float(cell) : 0.f;
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
            }
          }
          acc_add_lanes(hw + yd + k, gr);
        }
      } else if (col.type == SUAVE_TYPE_CAT) {
        const int C = col.n_classes;
        const int ldw = yd + K;
        const int code = obs ? cell : -1;
        // sweep 1: logits, online log-sum-exp (logit 0 is pinned to 0)
        float mx = 0.f, se = 1.f, lcode = 0.f;
        for (int cl = 1; cl < C; ++cl) {
          const float* wr = w + (cl - 1) * ldw;
          float L = wl[(cl - 1) * ldw + yd + k];
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
              }
            }
            acc_add_lanes(hr + yd + k, g);
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
        const float* wml = wsm + (col.w2_off - hw_begin);  // mean_layer [yd + K]
        const float* wm = wml;
        float* hwm = hw_acc + (col.w2_off - hw_begin);
        const int code = obs ? cell : -1;
        const int t = max(code, 0);  // target = max(round(sum thermo) - 1, 0)
        float mo = wml[yd + k];
        for (int i = 0; i < yd; ++i) mo += wm[i] * yb[i];
        float thr = 0.f, sg_t = 1.0f, sg_tm1 = 0.f;
        for (int j = 0; j < C - 1; ++j) {
          thr += softplus_f(wl[j * K + k]) + kEps;
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
            }
          }
          acc_add_lanes(hwm + yd + k, gmean);
          for (int jj = 0; jj < C - 1; ++jj) {
            const float dsp = (jj <= t - 1 ? du_tm1 : 0.f) + (jj <= t ? du_t : 0.f);
            const float gpart = dsp * softplus_grad_f(wl[jj * K + k]);
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
              if (j < K) {
                ds[j] += gpart * w[jj * K + j];
              }
            }
            acc_add_lanes(hw + jj * K + k, gpart);
          }
        }
      }
    }

    // ---- cross-wave reduction of the per-row terms (ds_w / recon_w alias the cell tile) ----
    LOSS_T(t3);
    __syncthreads();
    LOSS_T(t3b);
    recon_w[wave * 64 + lane] = recon;
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) ds_w[(wave * 64 + lane) * K + j] = ds[j];
    }
    __syncthreads();
    if (tid < 64) {
      float rsum = 0.f;
#pragma unroll
      for (int w = 0; w < kLossWaves; ++w) rsum += recon_w[w * 64 + tid];
      if (tid < nrows) a.recon_part[(size_t)chunk_id * a.n_rows + r0 + tid] = rsum;
      const float tot = wave_sum_dpp(tid < nrows ? rsum : 0.f);
      if (tid == 0) sum_recon += (double)tot;
    }
    if (GRAD && a.ds_part) {
      if (tid < nrows) {
        float* dst = a.ds_part + ((size_t)chunk_id * a.n_rows + r0 + tid) * a.ldK;
        for (int j = 0; j < K; ++j) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < kLossWaves; ++w) t += ds_w[(w * 64 + tid) * K + j];
          dst[j] = t;
        }
      }
    }
    // ---- write dY back (coalesced) ----
    if (GRAD) {
      for (int lr = wave; lr < nrows; lr += kLossWaves) {
        float* dst = a.y + (size_t)(r0 + lr) * a.ldY + ch.y_begin;
        const float* src = ytile + lr * a.ldy;
        for (int cc = lane; cc < ch.y_len; cc += 64) dst[cc] = src[cc];
      }
    }

    // ---- KL phase (chunk 0 only): 4 lanes per row ----
    if (a.do_kl && chunk_id == 0) {
     if (tid < 256) {  // waves 0-3: 4 lanes per row
      const int lr = tid >> 2, qd = tid & 3;
      const bool v4 = lr < nrows;
      const int rr = r0 + (v4 ? lr : 0);
      const int kk = k_s[lr];
      const int D = a.D;
      // compact head buffer: [logits (ldK) | mu of the selected component (ldD) | raw logvar (ldD)]
      const float* mu = a.encout + (size_t)rr * a.ldE + a.ldK;
      const float* lv = mu + a.ldD;
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
     }
      __syncthreads();
      if (tid == 0) {
        sum_cat += (double)red_s[0] + (double)red_s[2] + (double)red_s[4] + (double)red_s[6];
        sum_gauss += (double)red_s[1] + (double)red_s[3] + (double)red_s[5] + (double)red_s[7];
      }
    }
    LOSS_T(t5);
    LOSS_ACC(0, t1 - t0); LOSS_ACC(1, t2 - t1); LOSS_ACC(2, t3 - t2); LOSS_ACC(3, t3b - t3); LOSS_ACC(4, t5 - t3b);
    LOSS_ACC(5, 1);
  }
#ifdef SUAVE_LOSS_TIMING
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_loss_dbg[i], dbg_acc[i]);
#endif

  __syncthreads();
  if (GRAD && a.head_slab) {
    const int len = a.hw_max;  // this chunk's head range is [hw_begin, hw_begin + chunk length)
    const int next = (chunk_id + 1 < (int)gridDim.y) ? a.cols[a.chunks[chunk_id + 1].col_begin].w_off
                                                      : a.head_off + a.head_len;
    const int n = min(len, next - hw_begin);
    float* dst = a.head_slab + (size_t)blockIdx.x * a.head_len + (hw_begin - a.head_off);
    for (int i = tid; i < n; i += kLossThreads) dst[i] = hw_acc[i];
  }
  if (tid == 0 && a.sums_slab) {
    double* s = a.sums_slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    s[0] = sum_recon;
    s[1] = sum_cat;
    s[2] = sum_gauss;
    s[3] = 0.0;
  }
}

size_t loss_lds_bytes(int ldy, int K, int hw_max, int max_classes, int ldx, int ldm) {
  (void)ldm;
  const size_t cells = std::max((size_t)kLossWaves * max_classes * 64 + (size_t)64 * ldx, (size_t)kLossWaves * 64 * K + kLossWaves * 64);
  size_t floats = (size_t)64 * ldy + 2 * (size_t)hw_max + 64 + 64 + 16 + 2 * 64 + ldx + cells;
  return floats * sizeof(float);
}

template <int KMAX>
static int launch_loss_k(const LossArgs& a, bool grad, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  if (grad)
    hipLaunchKernelGGL((k_loss<true, KMAX>), dim3(grid_x, n_chunks), dim3(kLossThreads), lds_bytes, s, a);
  else
    hipLaunchKernelGGL((k_loss<false, KMAX>), dim3(grid_x, n_chunks), dim3(kLossThreads), lds_bytes, s, a);
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

#ifdef SUAVE_LOSS_TIMING
extern "C" int suave_debug_loss_timing(unsigned long long* out, int reset) {
  unsigned long long z[8] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loss_dbg), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_loss_dbg), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

}  // namespace suave
