// Kernel argument structs and host launchers shared between translation units.
#pragma once
#include "common.h"

namespace suave {

constexpr int kCellMissing = 0x7FC0DEAD;  // cell sentinel: masked / unobserved

struct LossArgs {
  const suave_col_t* cols;
  const ColChunk* chunks;
  const float* params;
  float* y;            // [n_rows, ldY]; overwritten with dY when GRAD
  suave_batch_t batch; // packed table (data rows)
  int n_rows;          // rows of y (virtual rows when vdiv > 1)
  int vdiv;            // virtual rows per data row (K in the unsupervised branch, else 1)
  const int32_t* kidx; // [n_rows] component of each row, or null -> row % vdiv
  const float* row_weight;  // [n_rows] multiplies the upstream gradient, or null
  float c_up;          // d loss / d recon_row
  float bk;            // d loss / d KL_row
  int K, D, ldY, ldy, ldK, ldD, ldE, ldT;
  int max_classes;
  float* log_px;          // [n_rows, n_cols] per-column terms on request (vdiv == 1), else null
  float* log_px_missing;  // [n_rows, n_cols] or null
  float* recon_part;   // [n_chunks, n_rows]
  float* ds_part;      // [n_chunks, ldK, n_rows]  (plane-major)
  int do_kl;
  const float* encout;  // KL phase: compact head buffer sel[n_rows, ldE] (see LatentFwdArgs), ldE = ldK + 2*ldD
  int prior_logits_off, prior_mu_off, prior_logvar_off;
  float* catkl;
  float* gausskl;
  float* gkl_logit;    // [n_rows, ldK]
  float* gkl_mu;       // [n_rows, ldD]
  float* gkl_lv;       // [n_rows, ldD]
  float* prior_terms;  // [n_rows, ldT]: d/d prior_mu_k | d/d prior_logvar_k | d/d prior_logits
  float* head_slab;    // [workers per chunk, head_len]  (worker = wavefront)
  double* sums_slab;   // [chunks * workers per chunk, 4]
  int head_off, head_len, hw_max;
  const int32_t* xsrc;  // [n_cols] x_index | (1 << 30 for cat / ordinal)
  const int32_t* cells; // [ceil(data rows / 64), n_cols, 64] packed cells from k_expand (tile-transposed)
  int n_cols;
  int ldx, ldm;         // LDS row strides of the staged cells (ints) and masks (bytes)
};

// Supervised branch: the encoder head is evaluated for the selected component only, so the per-row
// head buffer is compact: sel[B, ldS] = [logits (ldK) | mu_k (ldD) | raw logvar_k (ldD)], ldS = ldK + 2*ldD.
struct LatentFwdArgs {
  const float* sel;  // [B, ldS]
  int ldS, K, D, ldK, ldD;
  int n_rows;
  const int32_t* row_index;
  int64_t row_begin;
  const float* exp_noise;  // [B,K] or null
  const float* eps_in;     // [B,D] or null
  uint64_t seed, step;
  float tau;
  float* ysoft;   // [B, ldK]
  int32_t* kidx;  // [B]
  float* z;       // [B, ldD]
  float* latent_out;    // [B, D] or null
  int32_t* comp_out;    // [B] or null
  // rows sorted by component (stable): rank of the row inside its 256-row block, per-block counts
  int32_t* lrank;       // [B]
  int32_t* blk_cnt;     // [ceil(B/256), K]
  int32_t* perm_fill;   // k_assign pre-fills the sorted-position table of k_group_scatter with -1 (null: no fill)
  int perm_fill_n;
  // batches of one 256-row block: k_assign also does k_group_scatter's work (one launch less on the step's chain)
  int32_t* one_chunk_comp;   // [n_chunks_max] or null
  int32_t* one_chunk_begin;  // [K + 1]
  int one_chunks_max;
  // KL terms (k_reparam): losses.py:51-57, :87-97 with the prior of the selected component (model.py:1607-1613)
  const float* params;
  int prior_logits_off, prior_mu_off, prior_logvar_off;
  float bk;             // d loss / d KL_row = beta / global rows
  int grad;             // write the gradient buffers below
  int ldT;
  float* catkl;         // [B]
  float* gausskl;       // [B]
  float* gkl_logit;     // [B, ldK]
  float* prior_terms;   // [B, ldT]: d/d prior_mu_k | d/d prior_logvar_k | d/d prior_logits
  float* prior_slab;    // [gridDim.x, 2D + K, prior_ldc]: the block's per-component column sums of the prior terms
  int prior_ldc;        //   (column k: rows of component k, column K: all rows), replacing prior_terms + k_prior_colsum
  double* sums_slab;    // [gridDim.x, 4]: (0, sum cat KL, sum gauss KL, 0) per block
};

// Stable counting sort of the batch rows by selected component, in chunks of kGroupRows positions.
struct GroupArgs {
  int K, n_rows, n_blocks, n_chunks_max;
  const int32_t* kidx;
  const int32_t* lrank;
  const int32_t* blk_cnt;
  int32_t* chunk_comp;   // [n_chunks_max] component of each chunk, -1 past the last one
  int32_t* chunk_begin;  // [K + 1] first chunk of each component
  int32_t* perm;         // [n_chunks_max * kGroupRows] sorted position -> row (pre-filled with -1)
};

// Reduction of the per-chunk encoder-head weight-gradient slabs into the stacked head gradient.
struct HeadReduceArgs {
  const float* slab;          // [n_chunks, ldS, ldc]
  const int32_t* chunk_begin; // [K + 1]
  int K, D, ldK, ldD, ldS, n_in, ldc;
  float* grads;
  int w_off, b_off, ld;
};

struct LatentBwdArgs {
  const float* sel;
  int ldS, K, D, ldK, ldD, n_rows, n_chunks;
  const float* ysoft;
  const int32_t* kidx;
  const float* dz;         // [B, ldD]
  const float* ds_part;    // [n_chunks, ldK, B]  (plane-major)
  const float* gkl_logit;  // [B, ldK]  d loss/d logits from the categorical KL
  float tau;
  float* dsel;             // [B, ldS] gradient in the compact head layout (pads zero)
  // eps and both Gaussian-KL gradients are recomputed (k_reparam's generator call and expressions), not stored
  const float* params;
  int prior_mu_off, prior_logvar_off;
  float bk;
  const int32_t* row_index;
  int64_t row_begin;
  const float* eps_in;     // [B, D] injected eps or null -> the counter generator of k_reparam
  uint64_t seed, step;
};

// dst[dst_off + i*dst_si + j*dst_sj] = sum_s src[src_off + s*slab_stride + i*src_ld + j]
struct ReduceSeg {
  int64_t src_off, slab_stride;
  int32_t n_slabs, rows, cols, src_ld;
  int32_t dst_off, dst_si, dst_sj;
  int32_t bias_off1;  // != 0: column `cols` of every slab row is summed too, into dst[bias_off1 - 1 + i]
};

constexpr int kMaxReduceSegs = 48;
struct ReduceArgs {
  ReduceSeg segs[kMaxReduceSegs];
  int n_segs;
  const float* slab;
  float* dst;
  const double* sums_slab;  // [n_sum_slabs, 4] or null
  int n_sum_slabs;
  double* sums_out;         // [4], accumulated (+=); entry 3 += n_rows
  double n_rows;
};

// ---- unsupervised (expectation-form) ELBO: model.py:1576-1580, :1619-1654 ----
struct MixFwdArgs {
  const float* encout;     // [B, ldE]
  int ldE, K, D, ldK, ldD, n_rows;
  const int32_t* row_index;
  int64_t row_begin;
  const float* exp_noise;  // [B,K] or null
  const float* eps_in;     // [B,K,D] or null
  uint64_t seed, step;
  float tau;               // <= 0: no Gumbel relaxation, weights = softmax(logits)
  float* w;                // [B, ldK] posterior weights
  float* wflat;            // [B*K] the same weights per virtual row
  float* z_all;            // [B*K, ldD]
  float* eps_buf;          // [B*K, ldD]
  float* latent_out;       // [B, D] or null: sum_k w_k z_k
  int rb;                  // table rows per block (<= 64): 8 for small batches, so that a few hundred rows still fill CUs
};

struct MixFinArgs {
  const float* encout;
  int ldE, K, D, ldK, ldD, n_rows, n_chunks, ldT;
  int rb;                  // table rows per block (<= 64), as in MixFwdArgs
  const float* params;
  int pm_w_off, pm_b_off;
  const float* w;          // [B, ldK]
  const float* eps_buf;    // [B*K, ldD]
  const float* recon_part; // [n_chunks, B*K] per-virtual-row log-likelihood
  const float* dz_all;     // [B*K, ldD] (grad only)
  float tau, c_up, bk;
  float* recon;            // [B]
  float* catkl;
  float* gausskl;
  float* dencout;          // [B, ldE] (grad only)
  float* prior_terms;      // [B, ldT]: d loss / d prior-mean[k,d] | summed over k  (grad only)
  double* sums_slab;       // [gridDim.x, 4]
};

// Row-wise kernels of the wide categorical columns (wide.hip).
struct WideRowArgs {
  const WideCol* wide;
  int n_wide, n_rows, vdiv, K, ldK;
  float* y;             // [n_rows, ldY] internal y buffer
  int ldY;
  float* L;             // [n_rows, ldL] logits, overwritten with d loss / d logits when GRAD
  int ldL;
  const int32_t* kidx;  // [n_rows] component or null -> row % vdiv
  const float* s_dense; // [n_rows, K] dense assignments (decode) or null
  const float* row_weight;
  float c_up;
  const int32_t* cells; // tile-transposed packed cells (k_expand)
  int n_cols;
  float* recon_out;     // [n_rows]
  float* ds_out;        // [ldK, n_rows] (plane-major) or null
  double* sums_slab;    // [grid, 4] or null
  float* log_px;          // [n_rows, n_cols] per-column terms on request, else null
  float* log_px_missing;
};
int launch_wide_fill(const WideRowArgs& a, hipStream_t s);
int wide_softmax_grid(int n_rows);
int launch_wide_softmax(const WideRowArgs& a, bool grad, int max_classes, hipStream_t s);  // max_classes: over the wide columns
int launch_wide_ds(const WideRowArgs& a, hipStream_t s);
// wT[t_off + n * round_up(C-1, 4) + k] = W[k][n] for every wide column (pad columns zero)
int launch_wide_transpose(const WideCol* cols, int n_wide, const float* params, float* wT, hipStream_t s);

int launch_pack_columns(const suave_pack_col_t* cols_dev, int n_cols, int n_num, int n_disc, int64_t n, float* xnum,
                        int ld_num, int16_t* codes, int ld_disc, uint8_t* mask, int ld_mask, unsigned long long* rejected,
                        hipStream_t s);
int launch_unpack_columns(const suave_unpack_col_t* cols_dev, int n_cols, int64_t n, const float* out_num, int ld_num,
                          const int16_t* out_codes, int ld_disc, float* f32_block, int64_t* i64_block, int16_t* i16_block,
                          int8_t* i8_block, int64_t ld_rows, hipStream_t s);

int mix_rows_per_block(int n_rows);
int launch_mix_fwd(const MixFwdArgs& a, hipStream_t s);
int launch_mix_finalize(const MixFinArgs& a, bool grad, int grid, hipStream_t s);
int launch_colsum(const float* src, int ld, int n_rows, int n_cols, float* slab, int n_blocks, hipStream_t s);
int prior_colsum_blocks(int n_rows);
bool reparam_colsum_ok(int K, int D);    // prior column sums inside k_reparam (LDS tile fits)

int launch_prior_colsum(const float* terms, int ldT, const int32_t* kidx, int n_rows, int n_cols, int K, int ldc, float* slab,
                        hipStream_t s);  // per-component column sums, slabs [blocks][n_cols][ldc]

size_t loss_lds_bytes(int ldy, int K, int hw_max, int max_classes, int ldx, int ldm);
int loss_set_lds_limit(size_t lds_bytes);
int loss_workers_per_block();
int launch_loss(const LossArgs& a, bool grad, int kmax, bool wide16, int grid_x, int n_chunks, size_t lds_bytes, hipStream_t s);
// Row-major copies W^T [n_in, ldT] of the layer weights W [n_out, ld]: with them the data-gradient GEMM
// dA = dC * W reads both operands k-contiguous, i.e. it is a forward-type product the split kernels handle.
// The copies are written by extra blocks of the step's first launch (k_expand): as a launch of its own the
// transposition sat on the step's dependency chain for 5 us.
struct TransArgs {
  const float* src[2 * SUAVE_MAX_HIDDEN + 2];
  float* dst[2 * SUAVE_MAX_HIDDEN + 2];
  int n_out[2 * SUAVE_MAX_HIDDEN + 2], n_in[2 * SUAVE_MAX_HIDDEN + 2], ld[2 * SUAVE_MAX_HIDDEN + 2], ldT[2 * SUAVE_MAX_HIDDEN + 2];
  int n_mats, blocks_per_mat;  // n_mats == 0: nothing to transpose
};
int launch_expand(const suave_plan* p, const suave_batch_t& b, float* x, int32_t* cells, hipStream_t s,
                  const TransArgs* trans = nullptr);
int launch_assign(const LatentFwdArgs& a, hipStream_t s);
int launch_group(const GroupArgs& a, hipStream_t s);
int reparam_grid(int n_rows);
int launch_reparam(const LatentFwdArgs& a, hipStream_t s);
int launch_latent_bwd(const LatentBwdArgs& a, hipStream_t s);
int launch_posterior_stats(const float* encout, int ldE, int K, int D, int n_rows, float inv_temp, float* enc_out_user,
                           float* post_mean, float* post_logvar, float* probs, hipStream_t s, int32_t* hard_comp = nullptr,
                           float* hard_mu = nullptr);
int launch_sample_heads(const suave_plan* p, const int32_t* cls_off_dev, const float* params, const float* y,
                        const float* s, const int32_t* comp, int64_t n, const float* nn, const float* nu,
                        const float* ne, uint64_t seed, int64_t row_offset, float* out_num, int ld_num,
                        int16_t* out_codes, int ld_disc, hipStream_t st, const float* wide_logits = nullptr);
int launch_head_params(const suave_plan* p, int col_index, const float* params, const float* y, const float* s,
                       const int32_t* comp, int64_t n, float* out, int width, hipStream_t st);
int launch_reduce(const ReduceArgs& a, int max_elems, hipStream_t s, const HeadReduceArgs* head = nullptr);  // head: the encoder head's chunk slabs, summed by extra blocks of the same launch

// ---- fused generative decode (decode.hip): decoder trunk -> y_projection -> sample draw in one launch -------------
int decode_plan_init(suave_plan* p);                  // suave_plan_create / _destroy: device tables of the head-row layer
void decode_plan_free(suave_plan* p);
bool decode_fused_ok(const suave_plan* p);            // the model is inside the kernel's shapes (and arithmetic mode 4)
size_t decode_fused_image_bytes(const suave_plan* p); // scratch of a call: the fp16 plane images of the decoder's weights
int launch_decode_fused(const suave_plan* p, const float* params, const float* z, const int32_t* comp, int64_t n,
                        const float* nn, const float* nu, const float* ne, uint64_t seed, int64_t row_offset,
                        unsigned char* images, float* out_num, int ld_num, int16_t* out_codes, int ld_disc, float* y_out,
                        hipStream_t st);

// First encoder layer, embedded (wide categorical) columns: see embed.hip
struct EmbedFwdArgs {
  const EmbedCol* cols;
  int n_embed, n_disc;
  suave_batch_t batch;
  int n_rows;
  const float* params;
  int ld;            // row stride of the embedding blocks and of pre (= ldH[0])
  float* pre;        // [n_rows, ld] sum of the selected embedding rows
  int16_t* ecodes;   // [n_embed, n_rows] codes of the embedded columns (for the weight gradient) or null
};
struct EmbedBwdArgs {
  const EmbedCol* cols;
  int n_embed;
  int n_rows;
  const int16_t* ecodes;
  const float* dpre;  // [n_rows, ld] gradient of the first layer's pre-activation
  int ld;
  float* grads;       // flat gradient buffer
  int n_classes_total;  // sum of the embedded columns' class counts = rows of all embedding blocks
};
int launch_embed_fwd(const EmbedFwdArgs& a, hipStream_t s);
int launch_embed_bwd(const EmbedBwdArgs& a, hipStream_t s);

// head.hip: the classifier head (heads.py:62-91) -- loss / gradients / optional Adam of one batch
bool head_shape_ok(int C, int D);
size_t head_workspace_floats(int C, int D);
int launch_head_loss(const suave_head_t& h, int D, const float* latent, int ld, int latent_by_row_id, const int32_t* row_index,
                     int64_t row_begin, int64_t n, float* dz, int ldz, float* logits_out, hipStream_t s);

}  // namespace suave
