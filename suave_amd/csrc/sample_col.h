// The per-column sample draw shared by the decode kernels (sample.hip: k_sample_heads on a y_full tile from HBM;
// decode.hip: the fused decoder chain, y tiles handed over through LDS).
//
// Head samplers of suave/modules/distributions.py:67-80,112-117,130-134,167-174 as called from
// suave/modules/decoder.py:90-91,139,172,214-216,258, and the value conversion of sampling.decoder_outputs_to_frame
// (suave/sampling.py:138-179).  Every discrete decision comes out as the individually rounded binary32 operations of
// detmath.h would take it (oracle/sampling_oracle.py); see sample.hip for the two-stage scheme.
// Translation units that include this header are built with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt.
#pragma once
#include <type_traits>

#include "kernels.h"
#include "detmath.h"

namespace suave {

// Decision margins of the two-stage races (see sample.hip).
constexpr float kCatMargin = 2e-5f;
constexpr float kOrdDp = 4e-5f;
constexpr float kCountMargin = 1e-4f;

// Exact Poisson draw from one uniform for 64 <= rate < 65536: inversion over the values in the order k0, k0+1, k0-1,
// k0+2, ... from the mode k0 = floor(rate) (value k owns the next p(k) of [0,1) in that order, so P(k) = p(k)); about
// 1.6 sqrt(rate) values on average.  p(k0) = exp(k0 log1p(d/k0) - d - c) / sqrt(2 pi k0) with d = rate - k0 (exact),
// the log1p series written so that no large terms cancel, and Stirling's c = 1/(12 k0) - 1/(360 k0^3); neighbours by
// p(k+1) = p(k) rate/(k+1), p(k-1) = p(k) k/rate.  Mirrors sampling_oracle.poisson_mode_walk operation by operation.
__device__ inline float poisson_mode_walk(float rate, float u) {
  const float k0 = floorf(rate);
  const float d = detm::sub(rate, k0);
  const float x = detm::dvd(d, k0);
  float poly = detm::mul(0.2f, x);
  poly = detm::add(poly, -0.25f);
  poly = detm::mul(poly, x);
  poly = detm::add(poly, 0.33333334f);
  poly = detm::mul(poly, x);
  poly = detm::add(poly, -0.5f);
  const float e = detm::mul(detm::mul(d, x), poly);
  const float ik = detm::dvd(1.0f, k0);
  const float c = detm::mul(ik, detm::sub(0.083333336f, detm::mul(detm::mul(ik, ik), 0.0027777778f)));
  const float p0 = detm::dvd(detm::exp(detm::sub(e, c)), detm::sqrt_rn(detm::mul(6.2831855f, k0)));
  float acc = p0, out = k0, pu = p0, pd = p0;
  bool go = u > acc;
#pragma unroll 1
  for (int i = 1; i <= 4096 && go; ++i) {
    const float fi = (float)i;
    const float ku = detm::add(k0, fi);
    pu = detm::dvd(detm::mul(pu, rate), ku);
    acc = detm::add(acc, pu);
    if (!(u > acc)) { out = ku; break; }
    const float kd = detm::sub(k0, detm::sub(fi, 1.0f));  // the value whose probability pd holds
    pd = detm::dvd(detm::mul(pd, kd), rate);
    acc = detm::add(acc, pd);
    if (!(u > acc)) { out = detm::sub(kd, 1.0f); break; }
    go = (pu > 1e-9f) || (pd > 1e-9f);  // (smaller terms no longer move a binary32 sum near 1; then: the mode)
  }
  return out;
}

// (rate, uniform, normal) -> Poisson count; mirrors sampling_oracle.poisson_from_noise: sequential CDF inversion from
// 0 below rate 64, inversion from the mode up to 65536 (both exact up to binary32 CDF resolution), the normal draw with
// its first Cornish-Fisher term beyond
__device__ inline float poisson_from_noise(float rate, float u, float n) {
  const bool small = rate < 64.0f;
  if (!small && rate < 65536.0f) return poisson_mode_walk(rate, u);
  const float r = small ? rate : 1.0f;
  float p = detm::exp(-r);
  float cdf = p;
  float k = 0.f;
  bool go = small && (u > cdf);
  for (int i = 1; i <= 512 && go; ++i) {
    const float fi = (float)i;
    p = detm::dvd(detm::mul(p, r), fi);
    cdf = detm::add(cdf, p);
    k = fi;
    go = (u > cdf) && ((p > 1e-30f) || (fi <= r));
  }
  const float skew = detm::mul(detm::sub(detm::mul(n, n), 1.0f), 0.16666667f);
  float big = detm::add(rate, detm::mul(detm::sqrt_rn(rate), n));
  big = floorf(detm::add(detm::add(big, skew), 0.5f));
  big = fmaxf(big, 0.f);
  return small ? k : big;
}

// 1 / i, correctly rounded, for the first-stage Poisson walk: the loop index is wave-uniform, so the entry is a scalar
// load (same value as __frcp_rn(i), which is the ten-instruction IEEE division in these translation units)
struct InvIntTable { float v[513]; };
constexpr InvIntTable make_inv_int_table() {
  InvIntTable t{};
  for (int i = 1; i <= 512; ++i) t.v[i] = 1.0f / (float)i;
  return t;
}
static __constant__ InvIntTable kInvInt = make_inv_int_table();

struct SampleArgs {
  const suave_col_t* cols;
  const int32_t* cls_off;  // [n_cols] offset of the column's classes in noise_exp
  const float* params;
  const float* y;          // [n, ldY]
  const float* s;          // [n, K] dense assignments or null
  const int32_t* comp;     // [n] component index (used when s == null)
  int64_t n;
  int n_cols, n_num, n_disc, K, ldY, Y, total_classes;
  const float* noise_normal;   // [n, n_cols] or null
  const float* noise_uniform;  // [n, n_cols] or null
  const float* noise_exp;      // [n, total_classes] or null
  uint64_t seed;
  int64_t row_offset;
  float* out_num;
  int ld_num;
  int16_t* out_codes;
  int ld_disc;
  int ldy, ldon, ldoc;  // LDS row strides
  int head_off, head_len;  // staged weights: the likelihood heads' range of the flat parameter vector (0: not staged)
  int skip_wide;           // the wide categorical columns are drawn by k_wide_sample from GEMM logits
};

// the numeric draws of one (row, pair of numeric columns): see sample_cell
struct NumDraws {
  int pair = -1;  // (wave-uniform) the pair these belong to
  float n0, n1;
  uint32_t u0, u1;
};

struct NoHeadU {
  __device__ __forceinline__ float operator()(int) const { return 0.f; }
};
constexpr float kCatMarginU = 5e-5f;  // FAST_U: the logits come from an MFMA product (summation order, fp16 planes)

// One cell: column `c` (descriptor `col`) of table row `r` (generator key rid / rid_hi).  `y(i)` is element i of the
// column's y slice, `w` / `w2` its two weight blocks (LDS or global), `srow` the row's dense assignments (a.s != null)
// or `kc` its component.  Numeric columns return the value in `val`, discrete ones the code in `code`.
// FAST_U (decode.hip, component indices only): the y part of every head's linear map was already multiplied on the
// matrix cores -- u(j) = w_row_j . y_slice for the column's j-th output row (mean / logits rows / ordinal mean) -- and
// the staged weights carry per-component tables in place: w2[k] = the sampling standard deviation (real, pos) and
// w[cl * K + k] = softplus(partition weight) + eps (ordinal).  Values and the FIRST run of a race use them; a race
// that is too close is decided again from y and the raw weights (`w_part_raw`: the ordinal partition block as stored in
// the parameter vector) exactly as without FAST_U, so every outcome is the detmath one either way.
// INJ = false compiles the injected-noise loads out (the caller guarantees the three noise pointers are null): a global
// load on one side of a branch costs the other side an s_waitcnt vmcnt(0) at the join, and in a loop that also stores
// its results that wait is the round trip of the previous column's store.
template <bool FAST_U, bool INJ, class YF, class UF>
__device__ __forceinline__ void sample_cell(const SampleArgs& a, const suave_col_t& col, const int c, const YF& y, const UF& u,
                                            const float* w, const float* w2, const float* w_part_raw, const float* srow,
                                            const int kc, const int64_t r, const uint32_t rid, const uint32_t rid_hi,
                                            const Philox& rng, NumDraws& nd, float& val, int& code) {
  const int K = a.K;
  const int yd = col.y_dim;
  // the dot of [y ; s] with [w_y ; w_s] with fused multiply-adds (values, and the first run of a race)
  auto dotf = [&](const float* wv, int ydim) {
    float acc = 0.f;
    for (int i = 0; i < ydim; ++i) acc = fmaf(wv[i], y(i), acc);
    if (a.s) {
      for (int j = 0; j < K; ++j) acc = fmaf(wv[ydim + j], srow[j], acc);
    } else {
      acc += wv[ydim + kc];
    }
    return acc;
  };
  // sequential, individually rounded
  auto dot = [&](const float* wv, int ydim) {
    float acc = 0.f;
    for (int i = 0; i < ydim; ++i) acc = detm::add(acc, detm::mul(wv[i], y(i)));
    if (a.s) {
      for (int j = 0; j < K; ++j) acc = detm::add(acc, detm::mul(wv[ydim + j], srow[j]));
    } else {
      acc = detm::add(acc, wv[ydim + kc]);
    }
    return acc;
  };
  if (col.type <= SUAVE_TYPE_COUNT) {
    // One Philox block per (row, PAIR of numeric columns 2 j, 2 j + 1 in xnum order): words x, y -> one Box-Muller
    // transform, whose two normals go to the two columns; words z, w -> their uniform draws.  `nd` keeps the pair's
    // draws of this lane, so a caller that visits both columns of a pair back to back generates them once.
    float nrm, uni = 0.f;
    const int pair = col.x_index >> 1, odd = col.x_index & 1;
    const bool inj_n = INJ && a.noise_normal != nullptr, inj_u = INJ && a.noise_uniform != nullptr;
    if ((!inj_n || (col.type == SUAVE_TYPE_COUNT && !inj_u)) && nd.pair != pair) {
      const uint4 q = rng(rid, (uint32_t)pair, kStreamSampleNormal, rid_hi);
      const float2 n2 = normal2(q.x, q.y);
      nd.pair = pair; nd.n0 = n2.x; nd.n1 = n2.y; nd.u0 = q.z; nd.u1 = q.w;
    }
    if (inj_n) nrm = a.noise_normal[(size_t)r * a.n_cols + c];
    else nrm = odd ? nd.n1 : nd.n0;
    if (col.type == SUAVE_TYPE_COUNT) {
      if (inj_u) uni = a.noise_uniform[(size_t)r * a.n_cols + c];
      else uni = u01(odd ? nd.u1 : nd.u0);
      // Two stages, like the races below.  First the CDF walk on hardware exp / log / reciprocal; its partial sums differ
      // from the detmath ones by the rate's relative error (<= 4e-6: the dot's roundings + exp and log1p of softplus)
      // times |d cdf / d rate| * rate <= sqrt(rate / 2 pi) <= 3.2 below rate 64, plus the roundings of the walk
      // itself: under 2e-5.  The walk takes the same decisions as the reference arithmetic as long as u stays further
      // than kCountMargin from every partial sum it is compared with; otherwise (and for rates near or beyond the
      // branch point 64) the cell is drawn again with detmath -- the outcome is always the detmath one.
      const float rate_f = softplus_f(dotf(w, yd)) + kEps;
      bool sure = rate_f < 63.5f;
      {
        float p = __expf(-rate_f), cdf = p, k = 0.f, gap = fabsf(uni - cdf);
        bool go = sure && (uni > cdf);
        for (int i = 1; i <= 512 && go; ++i) {
          const float fi = (float)i;
          p = p * rate_f * kInvInt.v[i];
          cdf += p;
          k = fi;
          gap = fminf(gap, fabsf(uni - cdf));
          go = (uni > cdf) && ((p > 1e-30f) || (fi <= rate_f));
        }
        sure = sure && gap > kCountMargin;
        val = k;
      }
      if (!sure) {
        const float rho = dot(w, yd);
        const float rate = detm::add(detm::softplus(rho), kEps);
        val = poisson_from_noise(rate, uni, nrm);
      }
    } else {
      // values, not decisions: hardware exp / log / sqrt (within 1e-5 of the oracle; the tests ask for 1e-4)
      if constexpr (FAST_U) {
        const float am = u(0) + w[yd + kc];
        const float sd = w2[kc];
        if (col.type == SUAVE_TYPE_REAL) val = fmaf(fmaf(nrm, sd, am), col.scale, col.shift);
        else val = __expf(fmaf(nrm, sd, fmaf(am, col.scale, col.shift))) - 1.0f;
      } else {
        const float am = dotf(w, yd);
        const float bv = dotf(w2, 0);
        const float v = softplus_f(bv) + kEps;
        if (col.type == SUAVE_TYPE_REAL) {  // decoder.py:90-91
          const float sd = sqrtf(fmaxf(v, kEps));
          val = fmaf(fmaf(nrm, sd, am), col.scale, col.shift);
        } else {  // decoder.py:139
          const float mean = fmaf(am, col.scale, col.shift);
          const float vs = v * (col.scale * col.scale);
          const float sd = sqrtf(fmaxf(vs, kEps));
          val = __expf(fmaf(nrm, sd, mean)) - 1.0f;
        }
      }
    }
    return;
  }
  const int C = col.n_classes;
  const int coff = ((const __attribute__((address_space(4))) int32_t*)(uintptr_t)a.cls_off)[c];  // (plan table: scalar load)
  // classes are visited in order: one Philox block serves four consecutive classes
  uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
  auto draw_e = [&](int cl) {
    if (INJ && a.noise_exp) return a.noise_exp[(size_t)r * a.total_classes + coff + cl];
    if ((cl & 3) == 0) q4 = rng(rid, ((uint32_t)c << 13) | (uint32_t)(cl >> 2), kStreamSampleExp, rid_hi);
    const uint32_t word = (cl & 3) == 0 ? q4.x : (cl & 3) == 1 ? q4.y : (cl & 3) == 2 ? q4.z : q4.w;
    return exp_draw(word);
  };
  int best_i = 0;
  if (col.type == SUAVE_TYPE_CAT) {  // Gumbel-max == exponential race on softmax(logits)
    const int ldw = cat_ldw(col, K);
    // DET = false: fused dots and v_log_f32.  Against the detmath evaluation a score L - log(e) moves by at most
    // (y_dim + K + 1) roundings of the dot's partial sums plus 2 ulp of the logarithm: far inside
    // kCatMargin * (1 + |score|) for the few-term, O(1)..O(100) logits of a head.
    auto race = [&](auto det, float& lead) {
      constexpr bool DET = decltype(det)::value;
      q4 = make_uint4(0u, 0u, 0u, 0u);
      const float e0 = draw_e(0);
      float best = DET ? detm::sub(0.f, detm::log(e0)) : -ln_normal(e0);
      float second = -INFINITY;
      int bi = 0;
      for (int cl = 1; cl < C; ++cl) {
        const float ecl = draw_e(cl);
        float sc;
        if (DET) sc = detm::sub(dot(w + (cl - 1) * ldw, yd), detm::log(ecl));
        else if constexpr (FAST_U) sc = (u(cl - 1) + w[(cl - 1) * ldw + yd + kc]) - ln_normal(ecl);
        else sc = dotf(w + (cl - 1) * ldw, yd) - ln_normal(ecl);
        if (sc > best) { second = best; best = sc; bi = cl; }
        else second = fmaxf(second, sc);
      }
      lead = (best - second) - (FAST_U ? kCatMarginU : kCatMargin) * (1.0f + fabsf(best));
      return bi;
    };
    float lead;
    best_i = race(std::false_type{}, lead);
    if (!(lead > 0.f)) best_i = race(std::true_type{}, lead);  // too close (or NaN): the reference arithmetic decides
  } else {  // ordinal: race on the clipped cumulative-link probabilities
    // DET = false: hardware softplus / sigmoid / reciprocal.  A clipped probability then differs from detmath's by
    // at most kOrdDp (C <= 32 accumulated softplus errors through sigmoid' <= 1/4, plus the sigmoid's own), so two
    // scores p / e compare the same way whenever they are further apart than kOrdDp (1/e_a + 1/e_b) plus the
    // rounding of the quotients.
    auto race = [&](auto det, float& lead) {
      constexpr bool DET = decltype(det)::value;
      q4 = make_uint4(0u, 0u, 0u, 0u);
      float mo;
      if (DET) mo = dot(w2, yd);
      else if constexpr (FAST_U) mo = u(0) + w2[yd + kc];
      else mo = dotf(w2, yd);
      float thr = 0.f, prev = 0.f, best = -1.0f, second = -1.0f, ie_max = 0.f;
      int bi = 0;
      for (int cl = 0; cl < C; ++cl) {
        float pcl;
        if (cl < C - 1) {
          float sg;
          if (DET) {
            thr = detm::add(thr, detm::add(detm::softplus(dot(w_part_raw + cl * K, 0)), kEps));
            sg = detm::sigmoid(detm::sub(thr, mo));
            pcl = detm::sub(sg, prev);
          } else {
            if constexpr (FAST_U) thr += w[cl * K + kc];
            else thr += softplus_f(dotf(w + cl * K, 0)) + kEps;
            sg = sigmoid_f(thr - mo);
            pcl = sg - prev;
          }
          prev = sg;
        } else {
          pcl = DET ? detm::sub(1.0f, prev) : 1.0f - prev;
        }
        pcl = fminf(fmaxf(pcl, kEps), 1.0f);
        const float ecl = draw_e(cl);
        float sc;
        if (DET) sc = detm::dvd(pcl, ecl);
        else { const float ie = __frcp_rn(ecl); sc = pcl * ie; ie_max = fmaxf(ie_max, ie); }
        if (sc > best) { second = best; best = sc; bi = cl; }
        else second = fmaxf(second, sc);
      }
      lead = (best - second) - (2.0f * kOrdDp * ie_max + 4e-6f * best);
      return bi;
    };
    float lead;
    best_i = race(std::false_type{}, lead);
    if (!(lead > 0.f)) best_i = race(std::true_type{}, lead);
  }
  code = best_i;
}

}  // namespace suave
