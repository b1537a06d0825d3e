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
// lanes).  dY overwrites y in LDS and is written back coalesced.  Head-weight
// gradients are reduced across the 64 rows inside the wavefront -- the y part
// with DPP wavefront reductions, the per-component part with a segmented DPP
// scan over lanes ordered by component (see seg_add) -- accumulated per wave
// in LDS over its tiles, and flushed once per block to a slab that a
// fixed-order reduction sums (deterministic, no float atomics in global
// memory; the lanes of an LDS add instruction always hit distinct addresses,
// so the order of the additions is program order).
#include <type_traits>

#include "kernels.h"

namespace suave {


namespace {

// Timing-only ablation bits (scripts/loss_ablate.py builds one library per value; results are wrong by construction;
// 0 in every product build, where the guarded code folds away).  1: no dY write-back; 2 / 4: no per-component / no
// y-part accumulator adds; 8: no ds_part stores; 16: no wavefront sums; 32: no column loop; 64: y-part adds as plain
// stores; 512: y-part adds one lane at a time; 1024 / 2048 / 4096 / 8192: skip real+pos / count / categorical /
// ordinal columns; 16384: non-temporal dY and ds_part stores; 32768: dY read from LDS but not stored; 65536: no
// ragged-edge dY stores; 131072: dY stored fully coalesced (wrong place).
#ifndef SUAVE_LOSS_ABL
#define SUAVE_LOSS_ABL 0
#endif
constexpr int kAbl = SUAVE_LOSS_ABL;

constexpr int kMissing = kCellMissing;
// Every wavefront is an independent worker on (64-row tile, column chunk) tasks with private LDS: no block
// barrier anywhere in the kernel, so a slow column type in one wave never stalls another.
constexpr int kLossWaves = 4;
constexpr int kLossThreads = kLossWaves * 64;
constexpr int kWeightPad = 32;  // readable floats past a chunk's weight image: the numeric heads read KMAX weights at once

// Sum over the 64 lanes with DPP row operations; result is wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  if (kAbl & 16) return v;
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// v of the lane a DPP control selects (lanes the control leaves out, and rows outside row_mask, read 0)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_get_or_minus1(int v) {
  return __builtin_amdgcn_update_dpp(-1, v, CTRL, ROW_MASK, 0xF, false);
}

// v_rcp_f32 (1 ulp) in place of the IEEE division sequence (ten instructions): the likelihood terms and their
// gradients are checked to 1e-4, every divisor here is >= kEps or >= 1
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_fast(float x) { return rcp_fast(1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_grad_fast(float x) { return x > 20.0f ? 1.0f : sigmoid_fast(x); }

// Fire-and-forget LDS add (ds_add_f32, no return value to wait for).  Every index is only ever
// touched by one wavefront, in program order, so the sum order is fixed (deterministic).
__device__ __forceinline__ void acc_add(float* hw, int idx, float v, int lane) {
  if (kAbl & 4) return;
  if (kAbl & 64) { if (lane == 0) hw[idx] = v; return; }
  if (lane == 0) (void)__hip_atomic_fetch_add(hw + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The same for up to YB consecutive accumulators at once: lane i takes the wave-uniform t[i] (v_writelane) and lanes
// 0 .. n-1 add with ONE DS instruction at distinct addresses, instead of n single-lane adds.
template <int YB>
__device__ __forceinline__ void acc_add_vec(float* hw, const float (&t)[YB], int n, int lane) {
  if (kAbl & 4) return;
  if (kAbl & 512) {
#pragma unroll
    for (int i = 0; i < YB; ++i)
      if (i < n) acc_add(hw, i, t[i], lane);
    return;
  }
  int packed = 0;
#pragma unroll
  for (int i = 0; i < YB; ++i)  // (no clang builtin for v_writelane_b32 in this toolchain; immediate lane select: no hazard)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(packed) : "s"(__builtin_bit_cast(int, t[i])), "n"(i));
  if (lane < n)
    (void)__hip_atomic_fetch_add(hw + lane, __builtin_bit_cast(float, packed), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace

#ifdef SUAVE_LOSS_TIMING
__device__ unsigned long long g_loss_dbg[16];
#define LOSS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define LOSS_ACC(i, v) dbg_acc[i] += (v)
#else
#define LOSS_T(var)
#define LOSS_ACC(i, v)
#endif

// WIDE16: the plan has narrow discrete heads of 9 .. 16 classes, which take the 16-wide register blocks (168 registers,
// three waves per SIMD).  Without them the kernel fits 128 registers with no prefetch slots and no scratch: four waves
// per SIMD, 97 -> 89 us on S64 at 65 536 rows (same box) now that the per-wave LDS admits four blocks per CU.
template <bool GRAD, int KMAX, bool WIDE16>
__global__ __launch_bounds__(kLossThreads, WIDE16 ? 3 : 4) void k_loss(LossArgs a) {
  constexpr bool kWide16 = WIDE16;
#ifdef SUAVE_LOSS_TIMING
  unsigned long long dbg_acc[16] = {0};
  LOSS_T(tk0);
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int K = a.K;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: [hw_max + pad] the chunk's head weights, shared by the block's waves (they work on the same chunk) | per wave:
  // [64][ldy] y slice, overwritten with dY | [hw_max] head-weight gradients | [max_classes][64] scratch (the lane order
  // of a tile; the logits of the operand-at-a-time categorical path)
  const int wave_floats = 64 * a.ldy + a.hw_max + a.max_classes * 64;
  float* wsm = lds;
  float* ytile = lds + a.hw_max + kWeightPad + (size_t)wave * wave_floats;
  float* hw_acc = ytile + 64 * a.ldy;
  float* gsw = hw_acc + a.hw_max + lane;
  const int chunk_id = blockIdx.y;
  const ColChunk ch = a.chunks[chunk_id];
  const int hw_begin = a.cols[ch.col_begin].w_off;
  const int n_tiles = (a.n_rows + 63) / 64;
  const int worker = blockIdx.x * kLossWaves + wave, n_workers = gridDim.x * kLossWaves;
  double sum_recon = 0.0;
  const int hw_next = (chunk_id + 1 < (int)gridDim.y) ? a.cols[a.chunks[chunk_id + 1].col_begin].w_off
                                                      : a.head_off + a.head_len;
  const int hw_len = min(a.hw_max, hw_next - hw_begin);
  for (int i = tid; i < a.hw_max + kWeightPad; i += kLossThreads) wsm[i] = i < hw_len ? a.params[hw_begin + i] : 0.f;
  if (GRAD)
    for (int i = lane; i < a.hw_max; i += 64) hw_acc[i] = 0.f;
  __syncthreads();  // the shared weight image is complete (the kernel's only other barrier is the one before the flush)

  // Tile staging is software-pipelined: the y slice and the per-row scalars of the NEXT tile are loaded into
  // registers before the column phase of the current one and stored to LDS after it, so the HBM round trip
  // hides behind the arithmetic.  Each lane owns up to kPre float4 of the 64 x y_len slice (fixed (row,
  // float4) coordinates for the whole kernel); wider chunks stage the rest directly.
  // (Two slots: six -- the whole 24-column slice in registers -- cost 20 registers, 40 bytes of scratch per lane and
  // 5 us at 65 536 rows, where a wave has one tile and nothing to prefetch; at 262 144 rows two beat none by 2 %.)
  // The four-wave variant has none (they are what it gives up for its 128 registers).
#ifdef SUAVE_LOSS_KPRE  // tuning knob
  constexpr int kPre = SUAVE_LOSS_KPRE;
#else
  constexpr int kPre = WIDE16 ? 2 : 0;
#endif
  const int ybase = ch.y_begin & ~3;  // aligned start column
  const int lead = ch.y_begin - ybase;
  const int nq = (lead + ch.y_len + 3) / 4;
  // slot u of a lane is flat index lane + 64 u of the [64][nq] float4 grid; only slot 0's coordinates are
  // kept, the others follow by adding (64 / nq, 64 % nq) with carry (wave-uniform increments)
  const int y_lr0 = lane / nq, y_q0 = lane - y_lr0 * nq;
  const int step_lr = 64 / nq, step_q = 64 - step_lr * nq;
  float4 y_pre[kPre > 0 ? kPre : 1];
  int k_pre = 0;
  float up_pre = 0.f;
  auto prefetch = [&](int tile) {
    const int r0 = tile * 64;
    const int nrows = min(64, a.n_rows - r0);
    int lr = y_lr0, q = y_q0;
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      y_pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the table: finite zeros
      if (lr < nrows) y_pre[u] = *reinterpret_cast<const float4*>(a.y + (size_t)(r0 + lr) * a.ldY + ybase + q * 4);
      lr += step_lr; q += step_q;
      if (q >= nq) { q -= nq; ++lr; }
    }
    k_pre = 0;
    up_pre = 0.f;
    if (lane < nrows) {
      const int r = r0 + lane;
      k_pre = a.kidx ? a.kidx[r] : (r % a.vdiv);
      up_pre = a.c_up * (a.row_weight ? a.row_weight[r] : 1.0f);
    }
  };
  auto put_y = [&](int lr, int q, const float4& v) {
    float* dst = ytile + lr * a.ldy;
    const int c0 = q * 4 - lead;
    if (c0 >= 0 && c0 < ch.y_len) dst[c0] = v.x;
    if (c0 + 1 >= 0 && c0 + 1 < ch.y_len) dst[c0 + 1] = v.y;
    if (c0 + 2 >= 0 && c0 + 2 < ch.y_len) dst[c0 + 2] = v.z;
    if (c0 + 3 >= 0 && c0 + 3 < ch.y_len) dst[c0 + 3] = v.w;
  };
  // dY of one float4 slot back to global: whole float4 when it lies inside the chunk, else its valid lanes
  auto get_dy = [&](int r0, int nrows, int lr, int q) {
    if (lr >= nrows) return;
    const float* src = ytile + lr * a.ldy;
    float* dst = a.y + (size_t)(r0 + lr) * a.ldY + ybase + q * 4;
    const int c0 = q * 4 - lead;
    if (c0 >= 0 && c0 + 3 < ch.y_len) {
      const float4 v4 = make_float4(src[c0], src[c0 + 1], src[c0 + 2], src[c0 + 3]);
      if (kAbl & 32768) { asm volatile("" ::"v"(v4.x), "v"(v4.y), "v"(v4.z), "v"(v4.w)); return; }
      if (kAbl & 131072) {  // same bytes, fully coalesced (timing only)
        float* d2 = a.y + ((size_t)(r0 / 64) * gridDim.y + blockIdx.y) * (64 * ch.y_len) + (size_t)(lr * nq + q) * 4;
        *reinterpret_cast<float4*>(d2 - ((size_t)d2 & 15) / 4) = v4;
        return;
      }
      if (kAbl & 16384) {
        __builtin_nontemporal_store(v4.x, dst); __builtin_nontemporal_store(v4.y, dst + 1);
        __builtin_nontemporal_store(v4.z, dst + 2); __builtin_nontemporal_store(v4.w, dst + 3);
        return;
      }
      *reinterpret_cast<float4*>(dst) = v4;
    } else if (!(kAbl & 65536)) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + e >= 0 && c0 + e < ch.y_len) dst[e] = src[c0 + e];
    }
  };
  // Component part of a head-weight gradient: slot_base[k] += sum of g over the tile's rows assigned to component k.
  // The lanes of each half wavefront work on their 32 tile rows in component order (set up per tile below), so a
  // component's rows are a run of neighbouring lanes: a segmented scan over DPP row shifts (four steps inside a row of
  // 16 lanes, one row_bcast:15 step across the two rows of a half) leaves the run's sum in its last lane, and those
  // lanes -- at most one per component and half -- add it to the accumulator.  (One ds_add_f32 from all 64 lanes at
  // slot_base[k] was applied lane after lane by the LDS unit where lanes shared a component, 6.4 of them on average
  // at K = 10: a fifth of the kernel's time.)  Fixed order: every slot is touched by this wavefront only, DS
  // operations of a wavefront execute in program order, and the two halves add in two instructions.
  bool sg1 = false, sg2 = false, sg4 = false, sg8 = false, sgb = false, sg_end = false;
  int seg_k = 0;
  auto seg_add = [&](float* slot_base, float g) {
    if (kAbl & 2) return;
    float v = g, t;
    t = v + dpp_get<0x111>(v); v = sg1 ? t : v;        // row_shr:1
    t = v + dpp_get<0x112>(v); v = sg2 ? t : v;        // row_shr:2
    t = v + dpp_get<0x114>(v); v = sg4 ? t : v;        // row_shr:4
    t = v + dpp_get<0x118>(v); v = sg8 ? t : v;        // row_shr:8
    t = v + dpp_get<0x142, 0xA>(v); v = sgb ? t : v;   // row_bcast:15 into rows 1 and 3
#ifdef SUAVE_LOSS_ONE_ADD  // timing knob: both halves in one instruction (two lanes may then share a slot)
    if (sg_end) (void)__hip_atomic_fetch_add(slot_base + seg_k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    // one instruction per half: no two lanes of a DS instruction ever add at the same address, so the order of the
    // additions is program order and nothing rests on how the LDS unit serialises same-address lanes
    if (sg_end && lane < 32) (void)__hip_atomic_fetch_add(slot_base + seg_k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (sg_end && lane >= 32) (void)__hip_atomic_fetch_add(slot_base + seg_k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
  };
  // Register-blocked discrete heads (y_dim = classes <= YB): the column's y slice and, class by class, its weight
  // row are read from LDS in one batch (clamped, branch-free), and the YB wavefront reductions of a class's weight
  // gradient are independent DPP chains the scheduler interleaves.  (The operand-at-a-time loops they replace
  // waited for an LDS round trip per multiply and ran the reductions one after the other.)
  auto cat_col = [&](auto tag, const suave_col_t& col, int cell, bool obs, float u, float* yb, const float* wl, float* hw,
                     int k, float& recon, float (&ds)[KMAX], float& lp_out) {
    constexpr int YB = decltype(tag)::value;
    const int yd = col.y_dim, C = col.n_classes, ldw = yd + K;
    const int code = obs ? cell : -1;
    float yv[YB];
#pragma unroll
    for (int i = 0; i < YB; ++i) {
      const float t = yb[min(i, yd - 1)];
      yv[i] = i < yd ? t : 0.f;
    }
    // sweep 1: logits, online log-sum-exp (logit 0 is pinned to 0)
    float mx = 0.f, se = 1.f, lcode = 0.f;
    for (int cl = 1; cl < C; ++cl) {
      const float* wr = wl + (cl - 1) * ldw;
      float wv[YB];
#pragma unroll
      for (int i = 0; i < YB; ++i) wv[i] = wr[min(i, yd - 1)];
      float L = wr[yd + k];
#pragma unroll
      for (int i = 0; i < YB; ++i) L += wv[i] * yv[i];
      if (L > mx) { se = se * __expf(mx - L) + 1.0f; mx = L; }
      else se += __expf(L - mx);
      if (cl == code) lcode = L;
    }
    const float lse = mx + __logf(se);
    const bool has = code >= 0 && code < C;
    lp_out = has ? (lcode - lse) : 0.f;  // (a masked cell is an all-zero one-hot: 0)
    recon += lp_out;
    if (GRAD) {
      const bool live = obs && has;  // all-zero one-hot: sum(x) = 0 -> zero gradient
      float dy[YB];
#pragma unroll
      for (int i = 0; i < YB; ++i) dy[i] = 0.f;
      // sweep 2: g_c = u * (x_c - softmax_c); weight, assignment and y gradients of the class in one pass
      for (int cl = 1; cl < C; ++cl) {
        const float* wr = wl + (cl - 1) * ldw;
        float* hr = hw + (cl - 1) * ldw;
        float wv[YB], wk[KMAX];
#pragma unroll
        for (int i = 0; i < YB; ++i) wv[i] = wr[min(i, yd - 1)];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) wk[j] = wr[yd + j];  // (reads past K stay inside the padded weight image)
        float L = wr[yd + k];  // the class's logit again (YB multiply-adds) instead of 256 bytes of LDS per class and wave
#pragma unroll
        for (int i = 0; i < YB; ++i) L += wv[i] * yv[i];
        const float p = __expf(L - lse);
        const float g = live ? u * ((cl == code ? 1.0f : 0.f) - p) : 0.f;
        float t[YB];
#pragma unroll
        for (int i = 0; i < YB; ++i) t[i] = wave_sum_dpp(g * yv[i]);
        acc_add_vec<YB>(hr, t, yd, lane);
#pragma unroll
        for (int i = 0; i < YB; ++i) dy[i] += g * wv[i];
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < K) ds[j] += g * wk[j];
        seg_add(hr + yd, g);
      }
#pragma unroll
      for (int i = 0; i < YB; ++i)
        if (i < yd) yb[i] = dy[i];
    }
  };
  auto ord_col = [&](auto tag, const suave_col_t& col, int cell, bool obs, float u, float* yb, const float* wl, float* hw,
                     int k, float& recon, float (&ds)[KMAX], float& lp_out) {
    constexpr int YB = decltype(tag)::value;
    const int yd = col.y_dim, C = col.n_classes;
    const float* wml = wsm + (col.w2_off - hw_begin);  // mean_layer [yd + K]
    float* hwm = hw_acc + (col.w2_off - hw_begin);
    const int code = obs ? cell : -1;
    const int t = max(code, 0);  // target = max(round(sum thermo) - 1, 0)
    float yv[YB], wmv[YB], wmk[KMAX], raw[YB];
#pragma unroll
    for (int i = 0; i < YB; ++i) {
      const float y = yb[min(i, yd - 1)];
      yv[i] = i < yd ? y : 0.f;
      wmv[i] = wml[min(i, yd - 1)];
      raw[i] = wl[min(i, max(C - 2, 0)) * K + k];  // threshold parameters of this row's component
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) wmk[j] = wml[yd + j];
    float mo = wml[yd + k];
#pragma unroll
    for (int i = 0; i < YB; ++i) mo += wmv[i] * yv[i];
    float thr = 0.f, sg_t = 1.0f, sg_tm1 = 0.f;
#pragma unroll
    for (int j = 0; j < YB; ++j) {
      if (j < C - 1) {
        thr += softplus_f(raw[j]) + kEps;
        const float sgj = sigmoid_fast(thr - mo);
        if (j == t) sg_t = sgj;
        if (j == t - 1) sg_tm1 = sgj;
      }
    }
    const float p_raw = sg_t - sg_tm1;
    const float p = fminf(fmaxf(p_raw, kEps), 1.0f);
    lp_out = __logf(p);
    recon += obs ? lp_out : 0.f;
    if (GRAD) {
      const float gp = (obs && p_raw >= kEps && p_raw <= 1.0f) ? u * rcp_fast(p) : 0.f;
      const float du_t = (gp != 0.f && t < C - 1) ? gp * sg_t * (1.0f - sg_t) : 0.f;
      const float du_tm1 = (gp != 0.f && t > 0) ? -gp * sg_tm1 * (1.0f - sg_tm1) : 0.f;
      const float gmean = -(du_t + du_tm1);
      float s[YB];
#pragma unroll
      for (int i = 0; i < YB; ++i) s[i] = wave_sum_dpp(gmean * yv[i]);
      acc_add_vec<YB>(hwm, s, yd, lane);
#pragma unroll
      for (int i = 0; i < YB; ++i)
        if (i < yd) yb[i] = gmean * wmv[i];
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) ds[j] += gmean * wmk[j];
      seg_add(hwm + yd, gmean);
#pragma unroll
      for (int jj = 0; jj < YB; ++jj) {
        if (jj < C - 1) {
          float wk[KMAX];
#pragma unroll
          for (int j = 0; j < KMAX; ++j) wk[j] = wl[jj * K + j];
          const float dsp = (jj <= t - 1 ? du_tm1 : 0.f) + (jj <= t ? du_t : 0.f);
          const float gpart = dsp * softplus_grad_fast(raw[jj]);
#pragma unroll
          for (int j = 0; j < KMAX; ++j)
            if (j < K) ds[j] += gpart * wk[j];
          seg_add(hw + jj * K, gpart);
        }
      }
    }
  };
  if (worker < n_tiles) prefetch(worker);

  for (int tile = worker; tile < n_tiles; tile += n_workers) {
    const int r0 = tile * 64;
    const int nrows = min(64, a.n_rows - r0);
    LOSS_T(t0);
    // ---- registers -> LDS (DS operations of one wave execute in order: no barrier needed) ----
    {
      int lr = y_lr0, q = y_q0;
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        if (lr < 64) put_y(lr, q, y_pre[u]);
        lr += step_lr; q += step_q;
        if (q >= nq) { q -= nq; ++lr; }
      }
    }
    for (int idx = lane + kPre * 64; idx < 64 * nq; idx += 64) {  // wide chunks
      const int lr = idx / nq, q = idx - lr * nq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lr < nrows) v = *reinterpret_cast<const float4*>(a.y + (size_t)(r0 + lr) * a.ldY + ybase + q * 4);
      put_y(lr, q, v);
    }
    const int k_nat = k_pre;  // (lane = tile row order)
    const float up_nat = up_pre;
    if (tile + n_workers < n_tiles) prefetch(tile + n_workers);

    // ---- lane order of the column phase: each half wavefront takes its 32 tile rows sorted by (component, row) ----
    // (rows 0..31 stay in lanes 0..31, so the y reads at row * ldy keep their conflict-free bank pattern; the cell
    // loads and per-row stores move inside one 256-byte segment)
    int rank = 0;
    {
      int base_lo = 0, base_hi = 0;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        if (j < K) {
          const unsigned long long m = __ballot(k_nat == j);
          const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
          const int below = lane >= 32 ? 32 + base_hi + (int)__builtin_amdgcn_mbcnt_hi(hi, 0u)
                                       : base_lo + (int)__builtin_amdgcn_mbcnt_lo(lo, 0u);
          if (k_nat == j) rank = below;
          base_lo += __popc(lo);
          base_hi += __popc(hi);
        }
      }
    }
    int* order_s = reinterpret_cast<int*>(gsw - lane);  // 64 words of the categorical scratch (free between columns)
    order_s[rank] = lane;
    const int my_row = order_s[lane];
    const int k = __builtin_amdgcn_ds_bpermute(my_row << 2, k_nat);
    const float up = __int_as_float(__builtin_amdgcn_ds_bpermute(my_row << 2, __float_as_int(up_nat)));
    seg_k = k;
    sg1 = dpp_get_or_minus1<0x111>(k) == k;
    sg2 = dpp_get_or_minus1<0x112>(k) == k;
    sg4 = dpp_get_or_minus1<0x114>(k) == k;
    sg8 = dpp_get_or_minus1<0x118>(k) == k;
    sgb = dpp_get_or_minus1<0x142, 0xA>(k) == k;
    // (read by every lane before the test: a short-circuited || would switch lanes 31 and 63 off, and lanes 30
    // and 62 would read an inactive lane's zero)
    const int k_next = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, k);
    sg_end = ((lane & 31) == 31) | (k_next != k);

    // ---- column phase: lane = one tile row (my_row), wave-uniform column ----
    const bool valid = my_row < nrows;
    // packed cells in tile-transposed layout [row / 64][column][row % 64]: one coalesced load per column
    const int drow = a.vdiv == 1 ? r0 + my_row : (r0 + my_row) / a.vdiv;
    const int32_t* cell_ptr = a.cells + ((size_t)(drow >> 6) * a.n_cols) * 64 + (drow & 63);
    float recon = 0.f;
    float ds[KMAX];
    if (GRAD) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) ds[j] = 0.f;
    }
    float* yrow = ytile + my_row * a.ldy;
#ifndef SUAVE_LOSS_CELL_AHEAD  // tuning knob: columns whose cell is in flight ahead of the one being worked on (2 .. 4: within 2 % on S64 and W1024)
#define SUAVE_LOSS_CELL_AHEAD 1
#endif
    constexpr int kAhead = SUAVE_LOSS_CELL_AHEAD;
    int cell_q[kAhead];  // at the top of the column loop: the cells of columns c .. c + kAhead - 1
#pragma unroll
    for (int u = 0; u < kAhead; ++u)
      cell_q[u] = (valid && ch.col_begin + u < ch.col_end) ? cell_ptr[(size_t)(ch.col_begin + u) * 64] : kMissing;
    suave_col_t col_next = a.cols[ch.col_begin];  // descriptor and cell of the next column are fetched a column ahead

    LOSS_T(t1);
    LOSS_ACC(0, t1 - t0);
    for (int c = ch.col_begin; c < ((kAbl & 32) ? ch.col_begin : ch.col_end); ++c) {
      LOSS_T(tc0);
      const suave_col_t col = col_next;
      const int cell = cell_q[0];
      if (c + 1 < ch.col_end) {
#pragma unroll
        for (int u = 0; u + 1 < kAhead; ++u) cell_q[u] = cell_q[u + 1];
        cell_q[kAhead - 1] = (valid && c + kAhead < ch.col_end) ? cell_ptr[(size_t)(c + kAhead) * 64] : kMissing;
        col_next = a.cols[c + 1];
      }
      const bool obs = cell != kMissing;
      const float u = obs ? up : 0.f;
      float* yb = yrow + (col.y_off - ch.y_begin);
      const int yd = col.y_dim;
      // head weights come from the block's LDS copy: a dependent scalar load from global per weight
      // (s_load + s_waitcnt) costs hundreds of cycles each and serialises the column loop
      const float* wl = wsm + (col.w_off - hw_begin);
      const float* w = wl;
      float* hw = hw_acc + (col.w_off - hw_begin);
      float lp_cell = 0.f;  // the column's log-likelihood of this row's cell value (placeholder value when masked)
      if (((kAbl & 1024) && (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS)) ||
          ((kAbl & 2048) && col.type == SUAVE_TYPE_COUNT) || ((kAbl & 4096) && col.type == SUAVE_TYPE_CAT) ||
          ((kAbl & 8192) && col.type == SUAVE_TYPE_ORDINAL)) {
      } else if ((col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) && yd == 1) {
        // y_dim == 1 (the default; decoder.py:87-99): every LDS operand of the column is read before the arithmetic starts, the
        // K assignment weights unconditionally up to KMAX (the weight image is padded), so one wait covers them all
        const float* wvl = wsm + (col.w2_off - hw_begin);
        float* hwv = hw_acc + (col.w2_off - hw_begin);
        const float y0 = yb[0], w0 = wl[0];
        const float bm = wl[1 + k], bv = wvl[k];
        float wk[KMAX], wvk[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) { wk[j] = wl[1 + j]; wvk[j] = wvl[j]; }
        const float am = bm + w0 * y0;
        const float x = obs ? __int_as_float(cell) : 0.f;
        const float v = softplus_f(bv) + kEps;
        const float mean = am * col.scale + col.shift;
        const float vs = v * (col.scale * col.scale);
        const float xr = x * col.scale + col.shift;
        const float cc = fmaxf(vs, kEps);
        const float rr = xr - mean;
        const float icc = rcp_fast(cc), rri = rr * icc;
        float nll = 0.5f * (kLog2Pi + __logf(cc) + rr * rri);
        if (col.type == SUAVE_TYPE_POS) nll += xr;
        lp_cell = -nll;
        recon += obs ? lp_cell : 0.f;
        if (GRAD) {
          // masked / out-of-tile lanes get an exact zero (select, not 0 * x: x may be inf or NaN there)
          const float ga = obs ? u * rri * col.scale : 0.f;
          const float gb = (obs && vs >= kEps) ? u * (-0.5f) * (icc - rri * rri) *
                                                      (col.scale * col.scale) * softplus_grad_fast(bv)
                                                : 0.f;
          const float t = wave_sum_dpp(ga * y0);
          yb[0] = ga * w0;
#pragma unroll
          for (int j = 0; j < KMAX; ++j)
            if (j < K) ds[j] += ga * wk[j] + gb * wvk[j];
          acc_add(hw, 0, t, lane);
          seg_add(hw + 1, ga);
          seg_add(hwv, gb);
        }
      } else if (col.type == SUAVE_TYPE_COUNT && yd == 1) {
        const float y0 = yb[0], w0 = wl[0];
        float wk[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) wk[j] = wl[1 + j];
        const float rho = wl[1 + k] + w0 * y0;
        const float x = obs ? __int_as_float(cell) : 0.f;
        const float rate = softplus_f(rho) + kEps;
        const float cnt = fmaxf(__expf(x) - col.shift, 0.f);
        const float rc = fmaxf(rate, kEps);
        const float lp = (cnt == 0.f ? 0.f : cnt * __logf(rc)) - rc - lgammaf(cnt + 1.0f);
        lp_cell = lp;
        recon += obs ? lp : 0.f;
        if (GRAD) {
          const float gr = obs ? u * (cnt * rcp_fast(rc) - 1.0f) * softplus_grad_fast(rho) : 0.f;
          const float t = wave_sum_dpp(gr * y0);
          yb[0] = gr * w0;
#pragma unroll
          for (int j = 0; j < KMAX; ++j)
            if (j < K) ds[j] += gr * wk[j];
          acc_add(hw, 0, t, lane);
          seg_add(hw + 1, gr);
        }
      } else if (col.type == SUAVE_TYPE_REAL || col.type == SUAVE_TYPE_POS) {  // y_dim > 1 (Schema y_dimensions override)
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
        const float icc = rcp_fast(cc), rri = rr * icc;
        float nll = 0.5f * (kLog2Pi + __logf(cc) + rr * rri);
        if (col.type == SUAVE_TYPE_POS) nll += xr;
        lp_cell = -nll;
        recon += obs ? lp_cell : 0.f;
        if (GRAD) {
          // masked / out-of-tile lanes get an exact zero (select, not 0 * x: x may be inf or NaN there)
          const float ga = obs ? u * rri * col.scale : 0.f;
          const float gb = (obs && vs >= kEps) ? u * (-0.5f) * (icc - rri * rri) *
                                                      (col.scale * col.scale) * softplus_grad_fast(bv)
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
          seg_add(hw + yd, ga);
          seg_add(hwv, gb);
        }
      } else if (col.type == SUAVE_TYPE_COUNT) {
        float rho = wl[yd + k];
        for (int i = 0; i < yd; ++i) rho += w[i] * yb[i];
        const float x = obs ? __int_as_float(cell) : 0.f;
        const float rate = softplus_f(rho) + kEps;
        const float cnt = fmaxf(__expf(x) - col.shift, 0.f);
        const float rc = fmaxf(rate, kEps);
        const float lp = (cnt == 0.f ? 0.f : cnt * __logf(rc)) - rc - lgammaf(cnt + 1.0f);
        lp_cell = lp;
        recon += obs ? lp : 0.f;
        if (GRAD) {
          const float gr = obs ? u * (cnt * rcp_fast(rc) - 1.0f) * softplus_grad_fast(rho) : 0.f;
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
          seg_add(hw + yd, gr);
        }
      } else if (col.type == SUAVE_TYPE_CAT) {
        if (yd <= 2) cat_col(std::integral_constant<int, 2>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else if (yd <= 4) cat_col(std::integral_constant<int, 4>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else if (yd <= 8) cat_col(std::integral_constant<int, 8>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else if (yd <= 16 && kWide16) cat_col(std::integral_constant<int, 16>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else {  // more classes than the register-blocked paths take: one LDS round trip per operand
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
          lp_cell = has ? (lcode - lse) : 0.f;
          recon += lp_cell;
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
              seg_add(hr + yd, g);
            }
            // sweep 3: dY
            for (int i = 0; i < yd; ++i) {
              float dy = 0.f;
              for (int cl = 1; cl < C; ++cl) dy += gsw[cl * 64] * w[(cl - 1) * ldw + i];
              yb[i] = dy;
            }
          }
        }
      } else {  // ordinal: cumulative link
        const int need = max(yd, col.n_classes - 1);  // y features and thresholds both live in YB registers
        if (need <= 4) ord_col(std::integral_constant<int, 4>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else if (need <= 8) ord_col(std::integral_constant<int, 8>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else if (need <= 16 && kWide16) ord_col(std::integral_constant<int, 16>{}, col, cell, obs, u, yb, wl, hw, k, recon, ds, lp_cell);
        else {
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
            const float sgj = sigmoid_fast(thr - mo);
            if (j == t) sg_t = sgj;
            if (j == t - 1) sg_tm1 = sgj;
          }
          const float p_raw = sg_t - sg_tm1;
          const float p = fminf(fmaxf(p_raw, kEps), 1.0f);
          lp_cell = __logf(p);
          recon += obs ? lp_cell : 0.f;
          if (GRAD) {
            const float gp = (obs && p_raw >= kEps && p_raw <= 1.0f) ? u * rcp_fast(p) : 0.f;
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
            seg_add(hwm + yd, gmean);
            for (int jj = 0; jj < C - 1; ++jj) {
              const float dsp = (jj <= t - 1 ? du_tm1 : 0.f) + (jj <= t ? du_t : 0.f);
              const float gpart = dsp * softplus_grad_fast(wl[jj * K + k]);
#pragma unroll
              for (int j = 0; j < KMAX; ++j) {
                if (j < K) {
                  ds[j] += gpart * w[jj * K + j];
                }
              }
              seg_add(hw + jj * K, gpart);
            }
          }
        }
      }
      if (a.log_px && valid) a.log_px[(size_t)(r0 + my_row) * a.n_cols + c] = obs ? lp_cell : 0.f;
      if (a.log_px_missing && valid) a.log_px_missing[(size_t)(r0 + my_row) * a.n_cols + c] = obs ? 0.f : lp_cell;
      LOSS_T(tc1);
      LOSS_ACC(1 + col.type, tc1 - tc0);
      LOSS_ACC(8 + col.type, 1);
    }
    LOSS_T(t2);

    // ---- per-row outputs straight from the registers (lane = row) ----
    if (valid) a.recon_part[(size_t)chunk_id * a.n_rows + r0 + my_row] = recon;
    {
      const float tot = wave_sum_dpp(valid ? recon : 0.f);
      sum_recon += (double)tot;
    }
#ifdef SUAVE_LOSS_TIMING
    LOSS_T(t2a);
    LOSS_ACC(15, t2a - t2);
#endif
    if (GRAD && a.ds_part && valid && !(kAbl & 8)) {
      // plane-major [chunk][j][row]: lane = row, so every store instruction writes 256 contiguous bytes (the
      // row-major layout scattered 64 x 4 bytes at a 48-byte stride, 15 % of the kernel's time)
      float* dst = a.ds_part + (size_t)chunk_id * a.ldK * a.n_rows + r0 + my_row;
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < K) {
          if (kAbl & 16384) __builtin_nontemporal_store(ds[j], dst + (size_t)j * a.n_rows);
          else dst[(size_t)j * a.n_rows] = ds[j];
        }
    }
    // ---- write dY back ----
    if (GRAD && !(kAbl & 1)) {
      int lr = y_lr0, q = y_q0;
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        if (lr < 64) get_dy(r0, nrows, lr, q);
        lr += step_lr; q += step_q;
        if (q >= nq) { q -= nq; ++lr; }
      }
      for (int idx = lane + kPre * 64; idx < 64 * nq; idx += 64) {
        const int lr = idx / nq;
        get_dy(r0, nrows, lr, idx - lr * nq);
      }
    }
    LOSS_T(t3);
    LOSS_ACC(6, t3 - t2);
    LOSS_ACC(13, 1);
  }
#ifdef SUAVE_LOSS_TIMING
  {
    LOSS_T(tk1);
    dbg_acc[7] = tk1 - tk0;
    dbg_acc[14] = 1;
    if (lane == 0)
      for (int i = 0; i < 16; ++i) atomicAdd(&g_loss_dbg[i], dbg_acc[i]);
  }
#endif
  if (GRAD && a.head_slab) {
    // One slab row per BLOCK: the four wave workers' accumulators are added in a fixed order (the only block
    // barrier of the kernel, after every wave has left its tile loop).  A row per worker made the slab reduction
    // four times deeper -- it was that launch's critical path.
    __syncthreads();
    const float* h0 = lds + a.hw_max + kWeightPad + 64 * a.ldy;
    float* dst = a.head_slab + (size_t)blockIdx.x * a.head_len + (hw_begin - a.head_off);
    for (int i = tid; i < hw_len; i += kLossThreads)
      dst[i] = ((h0[i] + h0[(size_t)wave_floats + i]) + h0[2 * (size_t)wave_floats + i]) + h0[3 * (size_t)wave_floats + i];
  }
  if (lane == 0 && a.sums_slab) {
    double* s = a.sums_slab + ((size_t)blockIdx.y * n_workers + worker) * 4;
    s[0] = sum_recon;
    s[1] = 0.0;
    s[2] = 0.0;
    s[3] = 0.0;
  }
}

size_t loss_lds_bytes(int ldy, int K, int hw_max, int max_classes, int ldx, int ldm) {
  (void)K; (void)ldx; (void)ldm;
  return ((size_t)hw_max + kWeightPad + (size_t)kLossWaves * ((size_t)64 * ldy + (size_t)hw_max + (size_t)max_classes * 64)) * sizeof(float);
}

int loss_workers_per_block() { return kLossWaves; }

template <int KMAX, bool WIDE16>
static int launch_loss_k(const LossArgs& a, bool grad, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  if (grad)
    hipLaunchKernelGGL((k_loss<true, KMAX, WIDE16>), dim3(grid_x, n_chunks), dim3(kLossThreads), lds_bytes, s, a);
  else
    hipLaunchKernelGGL((k_loss<false, KMAX, WIDE16>), dim3(grid_x, n_chunks), dim3(kLossThreads), lds_bytes, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

template <bool WIDE16>
static int launch_loss_w(const LossArgs& a, bool grad, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  if (a.K <= 8) return launch_loss_k<8, WIDE16>(a, grad, grid_x, n_chunks, lds_bytes, s);
  if (a.K <= 12) return launch_loss_k<12, WIDE16>(a, grad, grid_x, n_chunks, lds_bytes, s);
  if (a.K <= 16) return launch_loss_k<16, WIDE16>(a, grad, grid_x, n_chunks, lds_bytes, s);
  return launch_loss_k<32, WIDE16>(a, grad, grid_x, n_chunks, lds_bytes, s);
}

// wide16: the plan's loss_wide16 (a narrow discrete head of 9 .. 16 classes exists: the three-wave variant)
int launch_loss(const LossArgs& a, bool grad, int kmax, bool wide16, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s) {
  (void)kmax;
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLoss, s);
  return wide16 ? launch_loss_w<true>(a, grad, grid_x, n_chunks, lds_bytes, s)
                : launch_loss_w<false>(a, grad, grid_x, n_chunks, lds_bytes, s);
}

template <bool GRAD, int KMAX>
static int set_lds_one(size_t lds_bytes) {
  SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loss<GRAD, KMAX, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  SUAVE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loss<GRAD, KMAX, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  return 0;
}

int loss_set_lds_limit(size_t lds_bytes) {
  int rc = 0;
  rc |= set_lds_one<true, 8>(lds_bytes);
  rc |= set_lds_one<false, 8>(lds_bytes);
  rc |= set_lds_one<true, 12>(lds_bytes);
  rc |= set_lds_one<false, 12>(lds_bytes);
  rc |= set_lds_one<true, 16>(lds_bytes);
  rc |= set_lds_one<false, 16>(lds_bytes);
  rc |= set_lds_one<true, 32>(lds_bytes);
  rc |= set_lds_one<false, 32>(lds_bytes);
  return rc;
}


#ifdef SUAVE_LOSS_TIMING
extern "C" int suave_debug_loss_timing(unsigned long long* out, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_loss_dbg), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_loss_dbg), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif

}  // namespace suave
