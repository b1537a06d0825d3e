// Row-wise kernels around the latent layer: packed batch -> encoder input,
// Gumbel-softmax component assignment + reparameterisation (forward), their
// backward scatter into the encoder-head gradient, and posterior moment matching.
//
// Reference: suave/model.py:2784-2922 (_prepare_training_tensors layout),
// :1573-1598 (assignment + reparam), :2924-2942, :2944-2964
// (_mixture_posterior_statistics); torch F.gumbel_softmax.
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

#include "detmath.h"

namespace suave {

// enc_map entry (built in suave_plan_create): bits 31..29 kind, 28..14 class index, 13..0 source index
//   kind 0 pad, 1 mask of column idx, 2 numeric cell xnum[idx], 3 one-hot of codes[idx], 4 thermometer of codes[idx]
// A block stages the packed cells of its rows in LDS with coalesced row reads, then writes
//   x     [B, ldI]    the encoder input (model.py:2784-2922 layout, quirks Q6/Q7), and
//   cells [ceil(B/64), n_cols, 64] one int per cell for the likelihood kernel (tile-transposed): float bits /
//         class code, kCellMissing when masked.
// 32 x 32 tiles `first`, `first + stride`, ... of matrix m of a TransArgs
__device__ __forceinline__ void transpose_tiles(const TransArgs& a, int m, int first, int stride, float (&tile)[32][33]) {
  const int n_out = a.n_out[m], n_in = a.n_in[m], ld = a.ld[m], ldT = a.ldT[m];
  const int tiles_i = (n_in + 31) / 32, tiles_o = (n_out + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = first; t < tiles_i * tiles_o; t += stride) {
    const int o0 = (t / tiles_i) * 32, i0 = (t % tiles_i) * 32;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int o = o0 + r, i = i0 + tx;
      tile[r][tx] = (o < n_out && i < n_in) ? a.src[m][(size_t)o * ld + i] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = i0 + r, o = o0 + tx;
      if (i < n_in && o < ldT) a.dst[m][(size_t)i * ldT + o] = tile[tx][r];  // pad columns o >= n_out get zeros
    }
  }
}

// Blocks [0, n_expand) expand the batch; blocks behind them write the transposed weight copies of `trans`
// (independent work riding in the step's first launch).
#ifndef SUAVE_EXPAND_MINBLOCKS  // tuning knob
#define SUAVE_EXPAND_MINBLOCKS 1
#endif
#ifndef SUAVE_EXPAND_STAGE  // tuning knob
#define SUAVE_EXPAND_STAGE 4
#endif
__global__ __launch_bounds__(256, SUAVE_EXPAND_MINBLOCKS) void k_expand(const int32_t* __restrict__ enc_map, const int32_t* __restrict__ xsrc,
                                                suave_batch_t b, int n_num, int n_disc, int n_cols, int ldI,
                                                int rows_per_block, float* __restrict__ x, int32_t* __restrict__ cells,
                                                int n_expand, TransArgs trans) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
  __shared__ float trans_tile[32][33];
  if ((int)blockIdx.x >= n_expand) {  // block-uniform
    const int idx = blockIdx.x - n_expand;
    transpose_tiles(trans, idx / trans.blocks_per_mat, idx % trans.blocks_per_mat, trans.blocks_per_mat, trans_tile);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // row strides: odd word counts for the lane = row reads of the cell phase (a 64-byte mask row put all 64 lanes on
  // two banks)
  const int ldn = n_num | 1, ldd = (n_disc + 1) & ~1, ldm = ((n_cols + 3) & ~3) + 4;
  float* xn_s = reinterpret_cast<float*>(stage);                                   // [R][ldn]
  int16_t* cd_s = reinterpret_cast<int16_t*>(xn_s + (size_t)rows_per_block * ldn);  // [R][ldd]
  uint8_t* mk_s = reinterpret_cast<uint8_t*>(cd_s + (size_t)rows_per_block * ldd);  // [R][ldm]
  for (int r0 = blockIdx.x * rows_per_block; r0 < b.n_rows; r0 += n_expand * rows_per_block) {
    const int nrows = min(rows_per_block, b.n_rows - r0);
    __syncthreads();
    // each wave stages a contiguous quarter of the block's rows: one load of their table row ids, then
    // kStage rows of cell loads in flight at a time (lane = element)
    const int rows_w = (rows_per_block + 3) / 4;
    const int lw0 = wave * rows_w;
    int64_t my_tr = -1;
    if (lane < rows_w && lw0 + lane < nrows) my_tr = b.row_index ? (int64_t)b.row_index[r0 + lw0 + lane] : b.row_begin + r0 + lw0 + lane;
    const int n_max = max(n_num, max(n_disc, n_cols));
    // Rows of cell loads in flight per wave.  The row ids (64-bit) and the staged values of kStage rows are what sets the
    // kernel's register count: 8 rows -> 168 registers, three blocks per CU and 2 552 blocks in 3.3 rounds; 4 rows -> 96
    // registers, five blocks per CU: 47 -> 38 us at 65 536 rows, W1024 0.57 -> 0.47 ms (2 rows: 60 registers, 40 us).
    constexpr int kStage = SUAVE_EXPAND_STAGE;
    for (int j0 = 0; j0 < rows_w; j0 += kStage) {
      int64_t tr[kStage];
#pragma unroll
      for (int u = 0; u < kStage; ++u) tr[u] = __shfl(my_tr, (j0 + u) & 63, 64);
      for (int i0 = 0; i0 < n_max; i0 += 64) {
        const int i = i0 + lane;
        float xv[kStage];
        int16_t cv[kStage];
        uint8_t mv[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const bool ok = j0 + u < rows_w && tr[u] >= 0;
          xv[u] = (ok && i < n_num) ? b.xnum[tr[u] * b.ld_num + i] : 0.f;
          cv[u] = (ok && i < n_disc) ? b.codes[tr[u] * b.ld_disc + i] : (int16_t)0;
          mv[u] = (ok && i < n_cols) ? b.mask[tr[u] * b.ld_mask + i] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int lr = lw0 + j0 + u;
          if (j0 + u < rows_w && tr[u] >= 0) {
            if (i < n_num) xn_s[lr * ldn + i] = xv[u];
            if (i < n_disc) cd_s[lr * ldd + i] = cv[u];
            if (i < n_cols) mk_s[lr * ldm + i] = mv[u];
          }
        }
      }
    }
    __syncthreads();
    // lane = input feature (its map entry stays in a register), wave walks the rows: no index division.
    // Four 64-feature chunks per round so that four map loads are in flight (wide schemas: thousands of
    // features, few rows per block).
    for (int i0 = 0; i0 < ldI; i0 += 256) {
      uint32_t e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * 64 + lane;
        e[q] = i < ldI ? (uint32_t)enc_map[i] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * 64 + lane;
        const int kind = e[q] >> 29, cls = (e[q] >> 14) & 0x7FFF, src = e[q] & 0x3FFF;
        if (i < ldI) {
          // Branch-free: the three staged arrays are read unconditionally (index 0 when the feature is of another
          // kind -- always inside the block's LDS) and the value is selected afterwards, four rows per round, so
          // twelve independent LDS reads are in flight instead of one read per divergent branch.
          const int sx = kind == 2 ? src : 0, sm = kind == 1 ? src : 0, sc = kind >= 3 ? src : 0;
          for (int lr0 = wave; lr0 < nrows; lr0 += 16) {
            float xv[4];
            int cv[4], mv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int lr = min(lr0 + 4 * u, nrows - 1);
              xv[u] = xn_s[lr * ldn + sx];
              mv[u] = mk_s[lr * ldm + sm];
              cv[u] = cd_s[lr * ldd + sc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int lr = lr0 + 4 * u;
              const float oh = kind == 3 ? (cv[u] == cls ? 1.f : 0.f) : ((cv[u] >= 0 && cls <= cv[u]) ? 1.f : 0.f);  // model.py:2890-2891
              const float v = kind == 2 ? xv[u] : kind == 1 ? (mv[u] ? 1.f : 0.f) : kind >= 3 ? oh : 0.f;
              if (lr < nrows) x[(size_t)(r0 + lr) * ldI + i] = v;
            }
          }
        }
      }
    }
    if (cells) {
      // tile-transposed layout [row / 64][column][row % 64]: lane = row, so the likelihood kernel's
      // lane = row reads of one column are one coalesced 256-byte load
      for (int lr = lane; lr < nrows; lr += 64) {
        const int r = r0 + lr;
        int32_t* dst = cells + ((size_t)(r >> 6) * n_cols) * 64 + (r & 63);
        for (int c0 = wave; c0 < n_cols; c0 += 16) {  // four columns per round: their table entries and LDS reads overlap
          int sx[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) sx[u] = xsrc[min(c0 + 4 * u, n_cols - 1)];
          int v[4], mk[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + 4 * u, n_cols - 1);
            const int src = sx[u] & 0xFFFFFF;
            v[u] = (sx[u] & (1 << 30)) ? (int)cd_s[lr * ldd + src] : __float_as_int(xn_s[lr * ldn + src]);
            mk[u] = mk_s[lr * ldm + c];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + 4 * u;
            if (c < n_cols) dst[(size_t)c * 64] = mk[u] ? v[u] : kCellMissing;
          }
        }
      }
    }
  }
}


// supervised, step 1: y_soft = softmax((logits + g)/tau), k = argmax.  Also the first pass of the stable counting
// sort by component: the row's rank among the rows of its 256-row block that chose the same component, and the
// block's per-component counts.
// A block is 256 rows x FOUR LANES PER ROW: the assignment path is deterministic binary32 (detmath log and divide,
// ~300 dependent instructions per component), so one thread per row was a chain of K of them (12 us at one wave per
// SIMD); lane q of a row takes the components of Philox blocks q, q + 4, ... (four components per block: one Philox
// call per lane for K <= 16), the row's four lanes combine (best score, first index) and the exponential sums by
// shuffles in a fixed order.  NB = Philox blocks per lane (compile-time: the scores stay in registers).
template <int NB>
__global__ __launch_bounds__(1024) void k_assign(LatentFwdArgs a) {
  __shared__ int wcnt[4][SUAVE_MAX_COMPONENTS];
  __shared__ int kb_s[256];
  const int t1024 = threadIdx.x;
  const Philox rng(a.seed);
  // (the -1 fill of the sorted-position table rides along: as a memset of its own it was a 4.6 us dispatch on the chain)
  if (a.perm_fill != nullptr)
    for (int i = blockIdx.x * 1024 + t1024; i < a.perm_fill_n; i += gridDim.x * 1024) a.perm_fill[i] = -1;
  {
    const int q = t1024 & 3;
    const int r = blockIdx.x * 256 + (t1024 >> 2);
    const bool valid = r < a.n_rows;
    const int rr = valid ? r : 0;  // (all four lanes of a row take part in the shuffles)
    const uint32_t rid = a.row_index ? (uint32_t)a.row_index[rr] : (uint32_t)(a.row_begin + rr);
    const float* lg = a.sel + (size_t)rr * a.ldS;
    float lgv[NB][4], ev[NB][4], t[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // every load first
        const int j = 4 * (q + 4 * b) + u;
        lgv[b][u] = j < a.K ? lg[j] : 0.f;
        ev[b][u] = (a.exp_noise && j < a.K) ? a.exp_noise[(size_t)rr * a.K + j] : 1.0f;
      }
    if (!a.exp_noise) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j4 = q + 4 * b;
        if (j4 * 4 < a.K) {
          const uint4 w4 = rng(rid, (uint32_t)j4, kStreamGumbel, (uint32_t)a.step);  // one block = 4 draws
          const uint32_t word[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) ev[b][u] = -__logf(u01(word[u]));
        }
      }
    }
    float best = -INFINITY;
    int kb = 0x7FFFFFFF;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * (q + 4 * b) + u;
        // Gumbel noise g = -log(e); the assignment path is deterministic binary32
        const float tj = detm::dvd(detm::sub(lgv[b][u], detm::log(ev[b][u])), a.tau);
        t[b][u] = tj;
        if (j < a.K && (tj > best || (tj == best && j < kb))) { best = tj; kb = j; }  // first maximum (ascending j within b; blocks interleave over lanes)
      }
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {  // the row's four lanes: greater score, on a tie the smaller index
      const float bo = __shfl_xor(best, off, 64);
      const int ko = __shfl_xor(kb, off, 64);
      if (bo > best || (bo == best && ko < kb)) { best = bo; kb = ko; }
    }
    if (kb == 0x7FFFFFFF) kb = 0;  // (every score NaN: component 0, as the sequential scan)
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * (q + 4 * b) + u;
        t[b][u] = j < a.K ? __expf(t[b][u] - best) : 0.f;
        sum += t[b][u];
      }
    const int l0 = (t1024 & 63) & ~3;
    const float s0 = __shfl(sum, l0, 64), s1 = __shfl(sum, l0 + 1, 64), s2 = __shfl(sum, l0 + 2, 64), s3 = __shfl(sum, l0 + 3, 64);
    const float inv = 1.0f / (((s0 + s1) + s2) + s3);
    if (valid) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * (q + 4 * b) + u;
          if (j < a.K) a.ysoft[(size_t)r * a.ldK + j] = t[b][u] * inv;
        }
      if (q == 0) {
        a.kidx[r] = kb;
        if (a.comp_out) a.comp_out[r] = kb;
      }
    }
    if (q == 0) kb_s[t1024 >> 2] = valid ? kb : -1;
  }
  __syncthreads();
  // the counting part: one thread per row (the first four waves)
  const int tid = t1024, lane = tid & 63, wave = tid >> 6;
  const bool counting = tid < 256;
  const int r = blockIdx.x * 256 + tid;
  const bool valid = counting && r < a.n_rows;
  const int kb = counting ? kb_s[tid] : -1;
  int rank = 0;
  if (counting) {
#pragma unroll 1
    for (int c = 0; c < a.K; ++c) {
      const unsigned long long m = __ballot(kb == c);
      if (kb == c) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][c] = __popcll(m);
    }
  }
  __syncthreads();
  if (valid) {
    for (int w = 0; w < wave; ++w) rank += wcnt[w][kb];
    a.lrank[r] = rank;
  }
  if (tid < a.K) a.blk_cnt[(size_t)blockIdx.x * a.K + tid] = wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
  if (a.one_chunk_comp != nullptr) {  // the whole batch is this block (launch_assign): the chunk table and the scatter
    __shared__ int cb[SUAVE_MAX_COMPONENTS + 1];
    if (tid == 0) {
      int ch = 0;
      for (int c = 0; c < a.K; ++c) {
        cb[c] = ch;
        ch = min(ch + (wcnt[0][c] + wcnt[1][c] + wcnt[2][c] + wcnt[3][c] + kGroupRows - 1) / kGroupRows, a.one_chunks_max);
      }
      cb[a.K] = ch;
    }
    __syncthreads();  // (also orders the -1 fill above before the scatter below: one block wrote both)
    if (tid <= a.K) a.one_chunk_begin[tid] = cb[tid];
    for (int ch = tid; ch < a.one_chunks_max; ch += 1024) {
      int comp = -1;
      for (int c = 0; c < a.K; ++c)
        if (ch >= cb[c] && ch < cb[c + 1]) comp = c;
      a.one_chunk_comp[ch] = comp;
    }
    if (valid) a.perm_fill[cb[kb] * kGroupRows + rank] = r;
  }
}

// Counting sort, passes 2 + 3 in one launch.  Every block works out for itself where its runs start: wave w takes
// components w, w + 4, ..., its lanes the blocks' counts (one strided load round per component, all hits in L2:
// the table is n_blocks x K ints), two shuffles trees give the component's rows below this block and its total.  From
// the totals every block derives the same chunk table (each component's rows start on a chunk boundary); block 0
// writes it out for the grouped GEMMs.  The sorted position of a row = first position of its component's chunks +
// its block's start within the component + its rank within the block's run.  (This replaced a one-block scan
// kernel, an 11 us latency chain, and its [n_blocks, K] table of starts.)
__global__ __launch_bounds__(256) void k_group_scatter(GroupArgs a) {
  __shared__ int below_s[SUAVE_MAX_COMPONENTS], tot_s[SUAVE_MAX_COMPONENTS], cb[SUAVE_MAX_COMPONENTS + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = wave; c < a.K; c += 4) {
    int below = 0, tot = 0;
    for (int b0 = 0; b0 < a.n_blocks; b0 += 256) {
      int v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + lane + 64 * u;
        v[u] = b < a.n_blocks ? a.blk_cnt[(size_t)b * a.K + c] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tot += v[u];
        below += (b0 + lane + 64 * u) < (int)blockIdx.x ? v[u] : 0;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      below += __shfl_xor(below, off, 64);
      tot += __shfl_xor(tot, off, 64);
    }
    if (lane == 0) { below_s[c] = below; tot_s[c] = tot; }
  }
  __syncthreads();
  if (tid == 0) {
    int ch = 0;
    for (int c = 0; c < a.K; ++c) {
      cb[c] = ch;
      ch = min(ch + (tot_s[c] + kGroupRows - 1) / kGroupRows, a.n_chunks_max);
    }
    cb[a.K] = ch;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid <= a.K) a.chunk_begin[tid] = cb[tid];
    for (int ch = tid; ch < a.n_chunks_max; ch += 256) {  // chunk -> component table
      int comp = -1;
      for (int c = 0; c < a.K; ++c)
        if (ch >= cb[c] && ch < cb[c + 1]) comp = c;
      a.chunk_comp[ch] = comp;
    }
  }
  const int r = blockIdx.x * 256 + tid;
  if (r < a.n_rows) {
    const int k = a.kidx[r];
    a.perm[cb[k] * kGroupRows + below_s[k] + a.lrank[r]] = r;
  }
}

// supervised, step 2 (4 lanes per row): z = mu_k + eps * exp(lv_k / 2) from the selected component's head
// outputs, and both KL terms with their gradients -- everything that only needs the compact head row.
#ifndef SUAVE_REPARAM_ABL
#define SUAVE_REPARAM_ABL 0  // timing-only ablation bits (1: no prior column sums, 2: no categorical KL, 4: no generator); 0 in product builds
#endif
__global__ __launch_bounds__(256) void k_reparam(LatentFwdArgs a) {
  constexpr int kAblR = SUAVE_REPARAM_ABL;
  extern __shared__ float prior_s[];  // [K*D] prior mu | [K*D] prior logvar | [K*D] +-1 / max(exp(logvar), 1e-6)
  __shared__ float red_s[8];
  // Prior gradients without the [B, 2D + K] round trip through HBM: the 64 rows' terms go to an LDS tile; thread
  // (row group g, term column c) walks its rows in order and adds each term to the LDS accumulator of the row's
  // component (a thread owns its accumulators: no atomics, fixed order); the row groups are combined in order into
  // the block's slab, and the per-block slabs are summed by the fixed-order reduction, as k_prior_colsum's were.
  __shared__ int kk_s[64];
  __shared__ float lp_s[SUAVE_MAX_COMPONENTS], ep_s[SUAVE_MAX_COMPONENTS];  // log-softmax of the prior logits, and its exp
  const bool colsum = a.grad && a.prior_slab != nullptr;
  const int n_terms = 2 * a.D + a.K;
  const int kTS = n_terms | 1;                     // odd row stride of the tile
  const int KP = a.K | 1;                          // odd accumulator stride
  const int RG = max(1, 256 / n_terms);            // row groups
  float* terms_s = prior_s + 3 * a.K * a.D;        // [64][kTS], present when colsum (launch_reparam sizes the LDS)
  float* acc_s = terms_s + 64 * kTS;               // [RG][n_terms][KP]
  float* slab = colsum ? a.prior_slab + (size_t)blockIdx.x * n_terms * a.prior_ldc : nullptr;
  if (colsum)
    for (int i = threadIdx.x; i < RG * n_terms * KP; i += 256) acc_s[i] = 0.f;  // (the combine re-zeroes what it reads)
  const Philox rng(a.seed);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, D = a.D;
  const int dq = (D + 3) / 4;
  float* pm_s = prior_s;
  float* plv_s = prior_s + K * D;
  float* ivp_s = prior_s + 2 * K * D;
  for (int i = tid; i < K * D; i += 256) {
    pm_s[i] = a.params[a.prior_mu_off + i];
    const float l = a.params[a.prior_logvar_off + i];
    plv_s[i] = l;
    const float vp_raw = __expf(l);  // sign of the stored reciprocal: negative when the 1e-6 floor is active
    ivp_s[i] = vp_raw >= 1e-6f ? 1.0f / vp_raw : -1e6f;
  }
  if (tid < K) {
    const float* pl = a.params + a.prior_logits_off;
    float mp = -INFINITY, sp = 0.f;
    for (int j = 0; j < K; ++j) mp = fmaxf(mp, pl[j]);
    for (int j = 0; j < K; ++j) sp += __expf(pl[j] - mp);
    lp_s[tid] = pl[tid] - (mp + __logf(sp));
    ep_s[tid] = __expf(lp_s[tid]);
  }
  __syncthreads();
  const bool vec = (D & 3) == 0;
  double sum_cat = 0.0, sum_gauss = 0.0;
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int lr = tid >> 2, qd = tid & 3;
    const bool v4 = r0 + lr < a.n_rows;
    const int r = v4 ? r0 + lr : r0;
    const int kk = a.kidx[r];
    const float* mu = a.sel + (size_t)r * a.ldS + a.ldK;
    const float* lv = mu + a.ldD;
    const uint32_t rid = a.row_index ? (uint32_t)a.row_index[r] : (uint32_t)(a.row_begin + r);
    float gk = 0.f;
    if (v4) {
      // the head row's next float4 pair is loaded before this one's arithmetic and stores (the stores would
      // otherwise fence the loads of the next trip: the compiler cannot prove they do not alias)
      float4 mu_nx = make_float4(0.f, 0.f, 0.f, 0.f), lv_nx = mu_nx;
      if (qd < dq) {
        mu_nx = *reinterpret_cast<const float4*>(mu + qd * 4);
        lv_nx = *reinterpret_cast<const float4*>(lv + qd * 4);
      }
      for (int q = qd; q < dq; q += 4) {
        const float4 mu4 = mu_nx, lv4 = lv_nx;
        if (q + 4 < dq) {
          mu_nx = *reinterpret_cast<const float4*>(mu + (q + 4) * 4);
          lv_nx = *reinterpret_cast<const float4*>(lv + (q + 4) * 4);
        }
        float ev[4] = {0.f, 0.f, 0.f, 0.f};
        if (kAblR & 4) {
          ev[0] = ev[1] = ev[2] = ev[3] = 0.5f;
        } else if (a.eps_in == nullptr) {
          const uint4 w = rng(rid, (uint32_t)q, kStreamEps, (uint32_t)a.step);
          const float2 n0 = normal2(w.x, w.y), n1 = normal2(w.z, w.w);
          ev[0] = n0.x; ev[1] = n0.y; ev[2] = n1.x; ev[3] = n1.y;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < D) ev[e] = a.eps_in[(size_t)r * D + q * 4 + e];
        }
        const float mv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, lvv[4] = {lv4.x, lv4.y, lv4.z, lv4.w};
        float zz[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, pl2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = q * 4 + e;
          if (d >= D) break;
          const float lvc = fminf(fmaxf(lvv[e], kLogvarMin), kLogvarMax);
          zz[e] = mv[e] + ev[e] * __expf(0.5f * lvc);
          const float plvd = plv_s[kk * D + d], ivp_sg = ivp_s[kk * D + d];
          const float ivp = fabsf(ivp_sg);
          const float diff = mv[e] - pm_s[kk * D + d];
          const float elv = __expf(lvc);
          const float quad = (elv + diff * diff) * ivp;
          gk += 0.5f * (plvd - lvc + quad - 1.0f);
          gm[e] = a.bk * diff * ivp;
          pl2[e] = a.bk * 0.5f * (1.0f - (ivp_sg > 0.f ? quad : 0.f));  // max(exp(lv_p), 1e-6) passes gradient above its floor only
        }
        if (vec) {
          *reinterpret_cast<float4*>(a.z + (size_t)r * a.ldD + q * 4) = make_float4(zz[0], zz[1], zz[2], zz[3]);
          if (a.grad) {
            if (colsum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                terms_s[lr * kTS + q * 4 + e] = -gm[e];
                terms_s[lr * kTS + D + q * 4 + e] = pl2[e];
              }
            } else {
              *reinterpret_cast<float4*>(a.prior_terms + (size_t)r * a.ldT + q * 4) = make_float4(-gm[0], -gm[1], -gm[2], -gm[3]);
              *reinterpret_cast<float4*>(a.prior_terms + (size_t)r * a.ldT + D + q * 4) = make_float4(pl2[0], pl2[1], pl2[2], pl2[3]);
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int d = q * 4 + e;
            if (d >= D) break;
            a.z[(size_t)r * a.ldD + d] = zz[e];
            if (a.grad) {
              if (colsum) {
                terms_s[lr * kTS + d] = -gm[e];
                terms_s[lr * kTS + D + d] = pl2[e];
              } else {
                a.prior_terms[(size_t)r * a.ldT + d] = -gm[e];
                a.prior_terms[(size_t)r * a.ldT + D + d] = pl2[e];
              }
            }
          }
        }
        if (a.latent_out) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < D) a.latent_out[(size_t)r * D + q * 4 + e] = zz[e];
        }
      }
    }
    gk += __shfl_xor(gk, 1, 64);
    gk += __shfl_xor(gk, 2, 64);
    // categorical KL of the row and its gradients: the row's four lanes take every fourth component each (the prior's
    // log-softmax comes from LDS, once per block)
    float ck = 0.f;
    if (!(kAblR & 2)) {
      const float* lg = a.sel + (size_t)r * a.ldS;
      constexpr int kPer = SUAVE_MAX_COMPONENTS / 4;
      float lgv[kPer];
      float mq = -INFINITY;
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int j = qd + 4 * i;
        lgv[i] = j < K ? lg[j] : -INFINITY;
        mq = fmaxf(mq, lgv[i]);
      }
      mq = fmaxf(mq, __shfl_xor(mq, 1, 64));
      mq = fmaxf(mq, __shfl_xor(mq, 2, 64));
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < kPer; ++i)
        if (qd + 4 * i < K) sq += __expf(lgv[i] - mq);
      sq += __shfl_xor(sq, 1, 64);
      sq += __shfl_xor(sq, 2, 64);
      const float lq = mq + __logf(sq);
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int j = qd + 4 * i;
        if (j < K) {
          const float logq = lgv[i] - lq;
          ck += __expf(logq) * (logq - lp_s[j]);
        }
      }
      ck += __shfl_xor(ck, 1, 64);
      ck += __shfl_xor(ck, 2, 64);
      if (a.grad && v4) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
          const int j = qd + 4 * i;
          if (j < K) {
            const float logq = lgv[i] - lq, logp = lp_s[j];
            const float q = __expf(logq);
            a.gkl_logit[(size_t)r * a.ldK + j] = a.bk * q * ((logq - logp) - ck);
            if (colsum) terms_s[lr * kTS + 2 * D + j] = a.bk * (ep_s[j] - q);
            else a.prior_terms[(size_t)r * a.ldT + 2 * D + j] = a.bk * (ep_s[j] - q);
          }
        }
      }
    }
    if (v4 && qd == 0) {
      a.catkl[r] = ck;
      a.gausskl[r] = gk;
    }
    const float t_c = wave_sum((v4 && qd == 0) ? ck : 0.f);
    const float t_g = wave_sum((v4 && qd == 0) ? gk : 0.f);
    if (colsum && qd == 0) kk_s[lr] = v4 ? kk : -1;
    __syncthreads();
    if (lane == 0) { red_s[wave * 2] = t_c; red_s[wave * 2 + 1] = t_g; }
    if (colsum && !(kAblR & 1)) {
      const int g = tid / n_terms;
      if (g < RG) {
        for (int c = tid - g * n_terms; c < n_terms; c += 256) {  // (more than 256 term columns: RG = 1, several per thread)
          float* my = acc_s + ((size_t)g * n_terms + c) * KP;
          for (int rr = g; rr < 64; rr += RG) {
            const int kr = kk_s[rr];
            if (kr >= 0) my[kr] += terms_s[rr * kTS + c];
          }
        }
      }
    }
    __syncthreads();
    if (colsum && !(kAblR & 1)) {  // row groups in order -> slab[c][k] (c < 2 D: per component; the K logits terms: all rows, column K)
      const bool first = r0 == (int)blockIdx.x * 64;  // later tiles of the block add to its slab (same thread, same address)
      for (int o = tid; o < 2 * D * K + K; o += 256) {
        const bool per_comp = o < 2 * D * K;
        const int c = per_comp ? o / K : 2 * D + (o - 2 * D * K);
        float t = 0.f;
        if (per_comp) {
          const int kq = o - c * K;
          for (int q = 0; q < RG; ++q) {
            float* src = acc_s + ((size_t)q * n_terms + c) * KP + kq;
            t += *src;
            *src = 0.f;
          }
        } else {
          for (int kq = 0; kq < K; ++kq)
            for (int q = 0; q < RG; ++q) {
              float* src = acc_s + ((size_t)q * n_terms + c) * KP + kq;
              t += *src;
              *src = 0.f;
            }
        }
        float* dst = slab + (size_t)c * a.prior_ldc + (per_comp ? o - c * K : K);
        *dst = first ? t : *dst + t;
      }
    }
    if (tid == 0) {
      sum_cat += (double)red_s[0] + (double)red_s[2] + (double)red_s[4] + (double)red_s[6];
      sum_gauss += (double)red_s[1] + (double)red_s[3] + (double)red_s[5] + (double)red_s[7];
    }
  }
  if (tid == 0 && a.sums_slab) {
    double* s = a.sums_slab + (size_t)blockIdx.x * 4;
    s[0] = 0.0; s[1] = sum_cat; s[2] = sum_gauss; s[3] = 0.0;
  }
}


// Straight-through Gumbel path + the selected component's mu / logvar gradients in the compact head
// layout: every other entry of d(enc_out) is zero and is never materialised.
// The kernel is a latency chain (one 64-row tile per block at the bench batch), so every global load of a tile is
// issued before anything waits: the assignment-gradient partials (four items x four chunks in flight per thread
// instead of one load at a time), y_soft, the KL logits gradient, and the dz / mu / logvar quads of the two
// Gaussian passes, which run as one pass (same quad, one dz load) ahead of the barriers of the softmax part.
#ifndef SUAVE_LATBWD_MINBLOCKS  // tuning knob
#define SUAVE_LATBWD_MINBLOCKS 4  // 128 registers (145 unconstrained, three blocks per CU: 1 024 blocks in 1.33 rounds); latent class 72.0 -> 68.9 us
#endif
__global__ __launch_bounds__(256, SUAVE_LATBWD_MINBLOCKS) void k_latent_bwd(LatentBwdArgs a) {
  constexpr int kDsStride = SUAVE_MAX_COMPONENTS + 1;  // odd stride: lane = row accesses hit distinct banks
  __shared__ float ds_s[64 * kDsStride];
  __shared__ float ys_s[64 * kDsStride];
  __shared__ float inner_s[64];
  extern __shared__ float lprior_s[];  // [K*D] prior mu | [K*D] 1 / max(exp(prior logvar), 1e-6)
  const int tid = threadIdx.x;
  const int K = a.K, D = a.D;
  // eps and the two Gaussian-KL gradients are recomputed here (same generator call, same expressions as k_reparam)
  // instead of being stored by the forward and read back: 6 x [B, ldD] floats of traffic for one more read of mu.
  const float* pm_s = lprior_s;
  const float* ivp_s = lprior_s + K * D;
  const Philox rng(a.seed);
  for (int i = tid; i < K * D; i += 256) {
    lprior_s[i] = a.params[a.prior_mu_off + i];
    const float vp_raw = __expf(a.params[a.prior_logvar_off + i]);
    lprior_s[K * D + i] = vp_raw >= 1e-6f ? 1.0f / vp_raw : 1e6f;
  }
  const float inv_tau = 1.0f / a.tau;
  const int nqk = a.ldK >> 2, nqd = a.ldD >> 2;  // ldK and ldD are multiples of 4: a float4 never straddles regions
  for (int r0 = blockIdx.x * 64; r0 < a.n_rows; r0 += gridDim.x * 64) {
    const int nrows = min(64, a.n_rows - r0);
    __syncthreads();  // prior tables staged; previous tile's LDS reads done
    // ---- assignment-gradient partials: ds_part is plane-major [chunk][j][row], lane = row within a plane
    // (coalesced), chunks summed in fixed order
    for (int i0 = 0; i0 < 64 * K; i0 += 1024) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = 0; c0 < a.n_chunks; c0 += 4) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = i0 + tid + 256 * u;
          const int j = idx >> 6, lr = idx & 63;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const bool ok = idx < 64 * K && lr < nrows && c0 + cc < a.n_chunks;
            v[u][cc] = ok ? a.ds_part[((size_t)(c0 + cc) * a.ldK + j) * a.n_rows + r0 + lr] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) acc[u] += v[u][cc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = i0 + tid + 256 * u;
        if (idx < 64 * K) ds_s[(idx & 63) * kDsStride + (idx >> 6)] = acc[u];
      }
    }
    for (int idx = tid; idx < nrows * K; idx += 256) {
      const int lr = idx / K, j = idx - lr * K;
      ys_s[lr * kDsStride + j] = a.ysoft[(size_t)(r0 + lr) * a.ldK + j];
    }
    // ---- selected mu: dz + KL;  selected logvar: dz * eps * exp(lv / 2) / 2 + KL -- two quads per trip, loads first
    for (int i0 = 0; i0 < nrows * nqd; i0 += 512) {
      float4 g4[2], mu4[2], raw4[2];
      int kk[2], rr[2], dd[2];
      uint32_t rid[2];
      bool ok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = i0 + tid + 256 * u;
        ok[u] = idx < nrows * nqd;
        const int lr = ok[u] ? idx / nqd : 0;
        dd[u] = ok[u] ? (idx - lr * nqd) * 4 : 0;
        rr[u] = r0 + lr;
        g4[u] = *reinterpret_cast<const float4*>(a.dz + (size_t)rr[u] * a.ldD + dd[u]);
        mu4[u] = *reinterpret_cast<const float4*>(a.sel + (size_t)rr[u] * a.ldS + a.ldK + dd[u]);
        raw4[u] = *reinterpret_cast<const float4*>(a.sel + (size_t)rr[u] * a.ldS + a.ldK + a.ldD + dd[u]);
        kk[u] = a.kidx[rr[u]];
        rid[u] = a.row_index ? (uint32_t)a.row_index[rr[u]] : (uint32_t)(a.row_begin + rr[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = rr[u], d = dd[u];
        const float g[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w}, mu[4] = {mu4[u].x, mu4[u].y, mu4[u].z, mu4[u].w};
        const float raw[4] = {raw4[u].x, raw4[u].y, raw4[u].z, raw4[u].w};
        float eps[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.eps_in == nullptr) {
          const uint4 w = rng(rid[u], (uint32_t)(d >> 2), kStreamEps, (uint32_t)a.step);
          const float2 n0 = normal2(w.x, w.y), n1 = normal2(w.z, w.w);
          eps[0] = n0.x; eps[1] = n0.y; eps[2] = n1.x; eps[3] = n1.y;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (d + e < D) eps[e] = a.eps_in[(size_t)r * D + d + e];
        }
        const int pk = kk[u] * D + d;
        float vm[4] = {0.f, 0.f, 0.f, 0.f}, vl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (d + e < D) {
            const float ivp = ivp_s[pk + e];
            vm[e] = g[e] + a.bk * (mu[e] - pm_s[pk + e]) * ivp;
            // clamp passes gradient inside its range only (where the clamped logvar is the raw one)
            if (raw[e] >= kLogvarMin && raw[e] <= kLogvarMax)
              vl[e] = g[e] * eps[e] * 0.5f * __expf(0.5f * raw[e]) + a.bk * 0.5f * (__expf(raw[e]) * ivp - 1.0f);
          }
        }
        if (ok[u]) {
          float* dst = a.dsel + (size_t)r * a.ldS + a.ldK + d;
          *reinterpret_cast<float4*>(dst) = make_float4(vm[0], vm[1], vm[2], vm[3]);
          *reinterpret_cast<float4*>(dst + a.ldD) = make_float4(vl[0], vl[1], vl[2], vl[3]);
        }
      }
    }
    __syncthreads();
    if (tid < nrows) {
      float inner = 0.f;
      for (int j = 0; j < K; ++j) inner += ys_s[tid * kDsStride + j] * ds_s[tid * kDsStride + j];
      inner_s[tid] = inner;
    }
    __syncthreads();
    for (int idx = tid; idx < nrows * nqk; idx += 256) {  // ---- component logits: straight-through softmax + KL
      const int lr = idx / nqk, q = (idx - lr * nqk) * 4;
      const int r = r0 + lr;
      const float4 gl = *reinterpret_cast<const float4*>(a.gkl_logit + (size_t)r * a.ldK + q);
      const float g4[4] = {gl.x, gl.y, gl.z, gl.w};
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q + e < K) v[e] = ys_s[lr * kDsStride + q + e] * (ds_s[lr * kDsStride + q + e] - inner_s[lr]) * inv_tau + g4[e];
      *reinterpret_cast<float4*>(a.dsel + (size_t)r * a.ldS + q) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// encode: clamp logvar, softmax(logits * inv_temperature), mixture moment matching
__global__ __launch_bounds__(256) void k_posterior_stats(const float* __restrict__ encout, int ldE, int K, int D, int n_rows,
                                                         float inv_temp, float* __restrict__ enc_out_user,
                                                         float* __restrict__ post_mean, float* __restrict__ post_logvar,
                                                         float* __restrict__ probs, int32_t* __restrict__ hard_comp,
                                                         float* __restrict__ hard_mu) {
  __shared__ float p_s[64 * SUAVE_MAX_COMPONENTS];
  __shared__ int kmax_s[64];
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
      float pbest = -1.0f;
      int kbest = 0;
      for (int j = 0; j < K; ++j) {
        const float p = p_s[tid * SUAVE_MAX_COMPONENTS + j] * inv;
        p_s[tid * SUAVE_MAX_COMPONENTS + j] = p;
        if (probs) probs[(size_t)(r0 + tid) * K + j] = p;
        if (p > pbest) { pbest = p; kbest = j; }  // posterior_probs.argmax: the first maximum (model.py:3627-3634)
      }
      kmax_s[tid] = kbest;
      if (hard_comp) hard_comp[r0 + tid] = kbest;
    }
    __syncthreads();
    if (hard_mu) {
      for (int idx = tid; idx < nrows * D; idx += 256) {
        const int lr = idx / D, d = idx % D;
        hard_mu[(size_t)(r0 + lr) * D + d] = encout[(size_t)(r0 + lr) * ldE + K + kmax_s[lr] * D + d];
      }
    }
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
  const int K = a.K, D = a.D, RB = a.rb;
  for (int r0 = blockIdx.x * RB; r0 < a.n_rows; r0 += gridDim.x * RB) {
    const int nrows = min(RB, a.n_rows - r0);
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
  const int K = a.K, D = a.D, RB = a.rb;
  double s_rec = 0.0, s_cat = 0.0, s_gau = 0.0;
  for (int r0 = blockIdx.x * RB; r0 < a.n_rows; r0 += gridDim.x * RB) {
    const int nrows = min(RB, a.n_rows - r0);
    __syncthreads();
    for (int idx = tid; idx < nrows * K; idx += 256) {
      const int lr = idx / K, k = idx % K;
      w_s[lr * SUAVE_MAX_COMPONENTS + k] = a.w[(size_t)(r0 + lr) * a.ldK + k];
      kl_s[lr * SUAVE_MAX_COMPONENTS + k] = 0.f;
    }
    __syncthreads();
    // per (row, component, latent dim): KL term and gradients of mu / logvar; KL summed over d per (row, k).
    // Eight lanes per (row, component): lane g takes d = g, g + 8, ... (32-byte runs of every array per group and
    // instruction) and the eight partial KL sums meet by a fixed butterfly of shuffles -- the same order every run.
    for (int idx = tid; idx < nrows * K * 8; idx += 256) {
      const int pk = idx >> 3, g = idx & 7;
      const int lr = pk / K, k = pk % K;
      const int r = r0 + lr;
      const float* mu = a.encout + (size_t)r * a.ldE + K + k * D;
      const float* lv = mu + K * D;
      const float wk = w_s[lr * SUAVE_MAX_COMPONENTS + k];
      const size_t v = (size_t)r * K + k;
      float kl = 0.f;
      for (int d = g; d < D; d += 8) {
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
      kl += __shfl_xor(kl, 1, 64);
      kl += __shfl_xor(kl, 2, 64);
      kl += __shfl_xor(kl, 4, 64);
      if (g == 0) kl_s[lr * SUAVE_MAX_COMPONENTS + k] = kl;
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

// Supervised prior gradients: out[c][k] = sum over the block's rows with component k of terms[row][c], and
// out[c][K] = the plain column sum (the prior-logits rows use it).  A thread owns one column and K private LDS
// register accumulators; it adds its rows in row order (eight loads in flight), so the result is reproducible; the
// per-block slabs [n_cols][ldc] are summed by the fixed-order reduction.  (This replaced a one-hot matrix and a
// K = batch-rows GEMM with ten output columns: 34 us -> 8 us per S64 step.)
template <int KMAX>
__global__ __launch_bounds__(256) void k_prior_colsum(const float* __restrict__ terms, int ldT, const int32_t* __restrict__ kidx,
                                                      int n_rows, int n_cols, int K, int ldc, float* __restrict__ slab) {
  extern __shared__ float acc_s[];  // [row groups][n_cols][K | 1]
  const int KP = K | 1;
  const int RG = max(1, 256 / n_cols);  // row groups: thread (g, c) takes rows rb + g, rb + g + RG, ... of column c
  const int rows_per = (n_rows + gridDim.x - 1) / gridDim.x;
  const int rb = blockIdx.x * rows_per, re = min(n_rows, rb + rows_per);
  const int g = threadIdx.x / n_cols;
  if (g < RG) {
    for (int c = threadIdx.x - g * n_cols; c < n_cols; c += 256) {  // (more than 256 columns: RG = 1, a thread takes several)
      float acc[KMAX];  // per-component sums in registers (compare-select: no dependent LDS round trip per row)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
      for (int r = rb + g; r < re; r += 8 * RG) {
        float v[8];
        int kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r + u * RG;
          v[u] = rr < re ? terms[(size_t)rr * ldT + c] : 0.f;
          kk[u] = rr < re ? kidx[rr] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < KMAX; ++k) acc[k] += kk[u] == k ? v[u] : 0.f;
      }
      float* my = acc_s + ((size_t)g * n_cols + c) * KP;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) my[k] = acc[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_cols; c += 256) {  // row groups combined in a fixed order
    float* dst = slab + ((size_t)blockIdx.x * n_cols + c) * ldc;
    float tot = 0.f;
    for (int k = 0; k < K; ++k) {
      float t = 0.f;
      for (int q = 0; q < RG; ++q) t += acc_s[((size_t)q * n_cols + c) * KP + k];
      dst[k] = t;
      tot += t;
    }
    dst[K] = tot;
  }
}

// ---- host launchers -----------------------------------------------------------
int prior_colsum_blocks(int n_rows) {
  static const int cap = [] { const char* e = getenv("SUAVE_PRIOR_BLOCKS"); return e ? std::max(1, atoi(e)) : 1024; }();  // tuning knob
  return std::max(1, std::min((n_rows + 63) / 64, cap));
}  // ~21 rows per thread: three load rounds (512 blocks: no better)

int launch_prior_colsum(const float* terms, int ldT, const int32_t* kidx, int n_rows, int n_cols, int K, int ldc, float* slab,
                        hipStream_t s) {
  ProfScope prof_scope(kProfReduce, s);
  const size_t lds = (size_t)std::max(1, 256 / n_cols) * n_cols * (K | 1) * sizeof(float);
  SUAVE_REQUIRE(lds <= 48 * 1024 && ldc >= K + 1, "prior colsum: too many columns for the LDS accumulators");
  const dim3 grid(prior_colsum_blocks(n_rows)), block(256);
  if (K <= 8) hipLaunchKernelGGL(k_prior_colsum<8>, grid, block, lds, s, terms, ldT, kidx, n_rows, n_cols, K, ldc, slab);
  else if (K <= 12) hipLaunchKernelGGL(k_prior_colsum<12>, grid, block, lds, s, terms, ldT, kidx, n_rows, n_cols, K, ldc, slab);
  else if (K <= 16) hipLaunchKernelGGL(k_prior_colsum<16>, grid, block, lds, s, terms, ldT, kidx, n_rows, n_cols, K, ldc, slab);
  else hipLaunchKernelGGL(k_prior_colsum<32>, grid, block, lds, s, terms, ldT, kidx, n_rows, n_cols, K, ldc, slab);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_mix_fwd(const MixFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  SUAVE_REQUIRE(a.rb >= 1 && a.rb <= 64, "mix_fwd: rows per block must be 1..64");
  hipLaunchKernelGGL(k_mix_fwd, dim3(std::min((a.n_rows + a.rb - 1) / a.rb, 2048)), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

// table rows per block of the two mixture kernels: 64 at training batch sizes; 8 below 16 384 rows, where 64-row blocks
// leave the chip empty and every thread walks 80+ dependent elements (a 256-row M20 step: 130 us -> see DESIGN.md)
int mix_rows_per_block(int n_rows) { return n_rows >= 16384 ? 64 : 8; }

int launch_mix_finalize(const MixFinArgs& a, bool grad, int grid, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  SUAVE_REQUIRE(a.rb >= 1 && a.rb <= 64 && grid >= 1, "mix_finalize: rows per block must be 1..64");
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

int launch_expand(const suave_plan* p, const suave_batch_t& b, float* x, int32_t* cells, hipStream_t s, const TransArgs* trans) {
  if (b.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfExpand, s);
  const suave_model_t& m = p->model;
  const int row_bytes = (m.n_num | 1) * 4 + ((m.n_disc + 1) & ~1) * 2 + ((m.n_cols + 3) & ~3) + 4;  // k_expand's ldn, ldd, ldm
  // Rows per block: 64 from 256 rows per CU on, 32 from 64 rows per CU on, else 16 (measured on S64 with five blocks per CU resident:
  // 65 536 rows 38.7 us at 32 rows, 33.2 at 64, 46 at 128; 16 384 rows 18.5 at 32, 25.2 at 64; 4 096 rows 16.7 at 32,
  // 13.3 at 16; 256 rows 15.8 / 12.5 / 12.2 at 32 / 16 / 8; 32 768 rows: 32 ahead of 64 by 2 % of the step), capped by the staging buffer (wide schemas: 8 rows).
  static const int rows_env = [] { const char* e = getenv("SUAVE_EXPAND_ROWS"); return e ? std::max(4, atoi(e) & ~3) : 0; }();  // tuning knob
  int rows_cap = rows_env;
  if (rows_cap == 0) rows_cap = b.n_rows >= (int64_t)256 * p->n_cu ? 64 : b.n_rows >= (int64_t)64 * p->n_cu ? 32 : 16;
  int rows_per_block = std::max(4, std::min(rows_cap, (48 * 1024) / std::max(row_bytes, 1)) & ~3);
  SUAVE_REQUIRE((size_t)rows_per_block * row_bytes <= 48 * 1024, "expand: schema too wide for the staging buffer");
  const int grid = std::min((b.n_rows + rows_per_block - 1) / rows_per_block, 4096);
  TransArgs t{};
  if (trans != nullptr) t = *trans;
  const int t_blocks = t.n_mats * t.blocks_per_mat;
  hipLaunchKernelGGL(k_expand, dim3(grid + t_blocks), dim3(256), (size_t)rows_per_block * row_bytes, s, p->enc_map_dev, p->xsrc_dev, b,
                     m.n_num, m.n_disc, m.n_cols, p->ldI, rows_per_block, x, cells, grid, t);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_assign(const LatentFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  const dim3 grid((a.n_rows + 255) / 256), block(1024);
  static_assert(SUAVE_MAX_COMPONENTS <= 32, "k_assign: two Philox blocks per lane cover 32 components");
  if (a.K <= 16) hipLaunchKernelGGL(k_assign<1>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(k_assign<2>, grid, block, 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_group(const GroupArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  // (a.perm was filled with -1 by k_assign: LatentFwdArgs::perm_fill)
  hipLaunchKernelGGL(k_group_scatter, dim3(a.n_blocks), dim3(256), 0, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int reparam_grid(int n_rows) { return std::max(1, std::min((n_rows + 63) / 64, 1024)); }

// LDS budgets of the fused paths (48 KB of dynamic LDS without a function attribute); beyond them the step stores
// prior_terms / eps / the KL gradients and runs k_prior_colsum, as before
static size_t reparam_colsum_floats(int K, int D) {
  const int n_terms = 2 * D + K;
  return 64 * (size_t)(n_terms | 1) + (size_t)std::max(1, 256 / n_terms) * n_terms * (K | 1);
}
bool reparam_colsum_ok(int K, int D) { return sizeof(float) * (3 * (size_t)K * D + reparam_colsum_floats(K, D)) <= 48 * 1024; }

int launch_reparam(const LatentFwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  ProfScope prof_scope(kProfLatent, s);
  size_t lds = sizeof(float) * 3 * a.K * a.D;
  if (a.grad && a.prior_slab) {
    SUAVE_REQUIRE(reparam_colsum_ok(a.K, a.D) && a.prior_ldc >= a.K + 1, "reparam: prior column sums do not fit the LDS tile");
    lds += sizeof(float) * reparam_colsum_floats(a.K, a.D);
  }
  hipLaunchKernelGGL(k_reparam, dim3(reparam_grid(a.n_rows)), dim3(256), lds, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_latent_bwd(const LatentBwdArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return 0;
  const int grid = std::min((a.n_rows + 63) / 64, 2048);
  ProfScope prof_scope(kProfLatent, s);
  const size_t lds = sizeof(float) * 2 * a.K * a.D;  // prior mean and reciprocal variance tables
  SUAVE_REQUIRE(lds <= 48 * 1024, "latent backward: the prior tables (2 K D floats) do not fit in LDS");
  hipLaunchKernelGGL(k_latent_bwd, dim3(grid), dim3(256), lds, s, a);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_posterior_stats(const float* encout, int ldE, int K, int D, int n_rows, float inv_temp, float* enc_out_user,
                           float* post_mean, float* post_logvar, float* probs, hipStream_t s, int32_t* hard_comp, float* hard_mu) {
  if (n_rows <= 0) return 0;
  const int grid = std::min((n_rows + 63) / 64, 2048);
  ProfScope prof_scope(kProfLatent, s);
  hipLaunchKernelGGL(k_posterior_stats, dim3(grid), dim3(256), 0, s, encout, ldE, K, D, n_rows, inv_temp, enc_out_user,
                     post_mean, post_logvar, probs, hard_comp, hard_mu);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace suave
