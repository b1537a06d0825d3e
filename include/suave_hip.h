/*
 * suave_hip.h -- C-ABI of libsuave_hip.so: the MI355X (gfx950) implementation of
 * SUAVE's ELBO training step and generative decode.
 *
 * The reference (xuxu-wei/SUAVE @ 2025-10-03) is pure Python/PyTorch and has no
 * FFI; its boundary for this path is the object protocol between
 * suave/model.py and suave/modules/{encoder,decoder,distributions,losses,prior}.py
 * + suave/sampling.py.  Each entry point below names the
 * reference code it replaces (file:line under the reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller owns every buffer; the library allocates no device memory after
 *    suave_plan_create (query sizes with suave_workspace_bytes);
 *  - all launches go to the caller-supplied hipStream_t (passed as void*);
 *    calls are asynchronous with respect to the host;
 *  - return value: 0 on success, non-zero error code otherwise; the message of
 *    the last error on the calling thread is returned by suave_last_error();
 *    no C++ exception crosses the ABI;
 *  - one plan per device; a plan is immutable after creation and may be shared
 *    by calls that use distinct workspaces.
 *
 * Data layout in HBM
 *  - packed table: xnum  float32 [n_rows, ld_num]   real/pos/count cells, schema order
 *                  codes int16   [n_rows, ld_disc]  cat code / ordinal level, -1 = all-zero encoding
 *                  mask  uint8   [n_rows, ld_mask]  1 = observed, schema order
 *    (the reference's dense one-hot / thermometer tensors, model.py:2784-2922,
 *    are never materialised in HBM except as the encoder input tile);
 *  - parameters: one flat float32 buffer, layout given by suave_layer_t /
 *    suave_col_t offsets; weight matrices are [n_out, ld] row-major with ld a
 *    multiple of 4 floats, pad entries zero;
 *  - gradients, Adam moments: flat buffers with the parameter layout.
 */
#ifndef SUAVE_HIP_H
#define SUAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUAVE_TYPE_REAL 0
#define SUAVE_TYPE_POS 1
#define SUAVE_TYPE_COUNT 2
#define SUAVE_TYPE_CAT 3
#define SUAVE_TYPE_ORDINAL 4

#define SUAVE_MAX_HIDDEN 8
#define SUAVE_MAX_COMPONENTS 32
/* Categorical columns with at least this many classes are "wide": their logits_layer
 * (decoder.py:201-206) runs as MFMA GEMMs instead of inside the fused likelihood kernel.
 * Layout consequences: the y slices of the narrow columns are
 * contiguous in schema order from 0; behind them (16-byte aligned) every wide column owns a
 * slot [y (y_dim) | assignment s (K) | pad to 4] of the internal y buffer, and its weight
 * matrix is stored [C-1, ld] with ld = y_dim + K rounded up to 4, 16-byte aligned, behind
 * the narrow head weights. */
#define SUAVE_WIDE_CAT_MIN_CLASSES 17

/* One column of the schema, in schema order (suave/types.py:9-33; head
 * modules suave/modules/decoder.py:56-265). */
typedef struct suave_col_t {
  int32_t type;      /* SUAVE_TYPE_* */
  int32_t n_classes; /* cat / ordinal, else 0 */
  int32_t y_off;     /* first column of this feature's slice of the internal y buffer (decoder.py:306-314) */
  int32_t y_dim;     /* width of that slice (types.py:267-287) */
  int32_t x_index;   /* column in xnum (numeric types) or in codes (cat / ordinal) */
  int32_t enc_off;   /* first column of the feature's values in the encoder input (model.py:2813-2905) */
  int32_t enc_mask;  /* column of the feature's mask in the encoder input */
  int32_t w_off;     /* flat-parameter offset: mean_layer / rate_layer / logits_layer / partition_layer weight */
  int32_t w2_off;    /* flat-parameter offset: var_layer (real,pos) / mean_layer (ordinal); -1 otherwise */
  float scale;       /* std (real) / std_log (pos); 1 otherwise */
  float shift;       /* mean (real) / mean_log (pos) / offset (count); 0 otherwise */
  int32_t enc_embed; /* -1, or (wide categorical column, n_hidden > 0) the flat-parameter offset of the column's
                      * encoder embedding block [n_classes, ld] (ld = enc[0].n_out rounded up to 4): the n_classes
                      * columns of encoder.backbone.0.weight that multiply the column's one-hot (model.py:2866-2870),
                      * stored transposed.  Such a column has no value entries in the dense encoder input (enc_off = -1):
                      * the first layer adds row `code` of the block instead of multiplying a one-hot. */
} suave_col_t;

/* One dense layer: weight [n_out, ld] row-major at w_off, bias [n_out] at b_off. */
typedef struct suave_layer_t {
  int32_t w_off, b_off, n_out, n_in, ld;
} suave_layer_t;

/* Model description.  Layers: encoder hidden layers, encoder head (rows
 * [0,K) = component_logits, [K,K+K*D) = mu_layer, then logvar_layer;
 * encoder.py:56-58), decoder hidden layers, y_projection (decoder.py:300). */
typedef struct suave_model_t {
  int32_t n_cols, n_num, n_disc;
  int32_t n_components; /* K */
  int32_t latent_dim;   /* D */
  int32_t n_hidden;     /* number of hidden layers in encoder and decoder */
  int32_t enc_in;       /* width of the dense encoder input: I of the reference minus the one-hot blocks of the
                         * columns with enc_embed >= 0 */
  int32_t total_y;      /* Y */
  int32_t behaviour;    /* 0 supervised (trainable prior tensors), 1 unsupervised (PriorMean) */
  int32_t n_params;     /* length of the flat parameter buffer */
  float dropout;        /* p of nn.Dropout after every hidden ReLU */
  suave_layer_t enc[SUAVE_MAX_HIDDEN];
  suave_layer_t enc_head;
  suave_layer_t dec[SUAVE_MAX_HIDDEN];
  suave_layer_t y_proj;
  int32_t head_off, head_len; /* contiguous range of all likelihood-head weights */
  /* prior: supervised -> logits[K], mu[K,D], logvar[K,D] (model.py:543-551);
   * unsupervised -> pm_w = PriorMean.linear.weight [D,K], pm_b = bias [D]
   * (prior.py:20-39), prior logits / logvar fixed at zero (model.py:538-542). */
  int32_t prior_logits_off, prior_mu_off, prior_logvar_off;
  int32_t pm_w_off, pm_b_off;
  /* rows of y_projection / columns of the internal y buffer: total_y plus the alignment pads and
   * assignment slots of the wide categorical columns (== total_y when the schema has none).  Pad
   * rows of y_projection hold zeros. */
  int32_t y_rows;
} suave_model_t;

/* A batch of rows of the packed table.  row_index == NULL selects rows
 * [row_begin, row_begin + n_rows); otherwise row_index[i] is the table row of
 * batch row i (the per-epoch permutation slice, model.py:1465-1491). */
typedef struct suave_batch_t {
  const float* xnum;
  const int16_t* codes;
  const uint8_t* mask;
  const int32_t* row_index;
  int64_t row_begin;
  int32_t n_rows;
  int32_t ld_num, ld_disc, ld_mask;
} suave_batch_t;

/* Noise of one ELBO step (SURVEY.md section 8a "noise inventory").  A NULL
 * pointer makes the kernels draw that noise themselves from the counter-based
 * generator keyed by (seed, step, table row); non-NULL pointers inject
 * explicit tensors (parity runs). */
typedef struct suave_noise_t {
  const float* exp_noise;  /* [B,K]   Exp(1) draws behind the Gumbel noise (F.gumbel_softmax) */
  const float* eps;        /* [B,D] supervised / [B,K,D] unsupervised: N(0,1) of model.py:2929 */
  const uint8_t* enc_keep[SUAVE_MAX_HIDDEN]; /* [B,H_i] dropout keep masks, encoder.py:52 */
  const uint8_t* dec_keep[SUAVE_MAX_HIDDEN]; /* [B(*K),H_i] dropout keep masks, decoder.py:292 */
  uint64_t seed;
  uint64_t step;
} suave_noise_t;

/* Per-row outputs of the ELBO step: the return contract of
 * SUAVE._forward_elbo_batch (model.py:1656-1667).  Any pointer may be NULL. */
typedef struct suave_elbo_out_t {
  float* reconstruction; /* [B] */
  float* categorical_kl; /* [B] */
  float* gaussian_kl;    /* [B] */
  float* latent;         /* [B,D] */
  int32_t* component;    /* [B] selected component (supervised) */
  float* enc_out;        /* [B, K + 2*K*D]: component_logits | component_mu | component_logvar (clamped) */
  double* sums;          /* [4]: sum recon, sum cat_kl, sum gauss_kl, n_rows -- accumulated (+=) */
  /* Per-column terms of the decoder output dict (decoder.py:348-379: "log_px" / "log_px_missing" of every head):
   * [B, n_cols] row-major, schema column order.  log_px[b,c] is the column's log-likelihood where the cell is
   * observed and 0 where it is masked; log_px_missing the complement (the likelihood of the masked cell's
   * placeholder value, 0 where observed).  reconstruction[b] == sum_c log_px[b,c].  Supervised branch only. */
  float* log_px;
  float* log_px_missing;
} suave_elbo_out_t;

typedef struct suave_plan suave_plan; /* opaque */

const char* suave_last_error(void);

/* Compile a schema into a device-resident plan (column table, chunking,
 * launch geometry).  Replaces the Python-side construction of EncoderMLP /
 * Decoder / heads (encoder.py:33-58, decoder.py:271-331). */
int suave_plan_create(const suave_model_t* model, const suave_col_t* cols, suave_plan** out);
int suave_plan_destroy(suave_plan* plan);

/* Bytes of scratch the step functions need for batches of up to max_rows rows
 * (max_rows counts virtual rows B*K for the unsupervised branch). */
int suave_workspace_bytes(const suave_plan* plan, int64_t max_rows, size_t* bytes);

/* ELBO forward only (validation / eval mode): SUAVE._forward_elbo_batch under
 * torch.no_grad() as driven by _compute_elbo_on_dataset (model.py:2497-2559).
 * training != 0 applies dropout. */
int suave_elbo_forward(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                       const suave_noise_t* noise, float beta_scale, float tau, int training,
                       void* workspace, size_t workspace_bytes, void* stream,
                       const suave_elbo_out_t* out);

/* ELBO forward + backward: SUAVE._forward_elbo_batch followed by
 * loss.backward() (model.py:1496-1505).  loss = -mean_b(recon - beta*(KLs)) over
 * global_rows rows (== batch rows on one GPU; the global batch under data
 * parallelism, so that summing rank gradients gives the single-batch gradient).
 * grads [n_params] is overwritten with d loss / d params (all zeros for a batch of zero rows: the share of a
 * rank whose shard of a small ragged batch is empty).
 * dlatent (may be NULL; supervised branch only) is [B,D] = d(extra loss)/d latent, added to the
 * gradient that reaches outputs["latent"]: the classification loss of joint fine-tuning
 * (`logits = classifier(outputs["latent"])`, model.py:1987-1991), whose head lives outside this library. */
int suave_elbo_backward(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                        const suave_noise_t* noise, float beta_scale, float tau,
                        int64_t global_rows, void* workspace, size_t workspace_bytes, void* stream,
                        const suave_elbo_out_t* out, float* grads, const float* dlatent);

/* suave_elbo_backward for data-parallel callers -- no counterpart in the reference (it has no distributed code;
 * the insertion point is between loss.backward() and optimizer.step(), model.py:1504-1506).  Same arguments and
 * results.  In addition the library reduces the decoder-side bucket of the flat gradient (decoder trunk,
 * y_projection, likelihood heads, prior: one contiguous range) as soon as the decoder backward has produced it
 * and records `decoder_ready_event` (a hipEvent_t) on the stream at that point; the encoder backward and its
 * reduction follow.  A caller makes its communication stream wait for the event and all-reduces that bucket while
 * the encoder backward runs, then all-reduces the encoder bucket once the call's stream work is done.  Both
 * branches: in the unsupervised one the bucket is decoder + heads + PriorMean, whose gradient (column sums of the
 * mixture finalisation) only needs the decoder backward.  An empty batch writes
 * a zero gradient and records the event. */
int suave_elbo_backward_overlap(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                                const suave_noise_t* noise, float beta_scale, float tau,
                                int64_t global_rows, void* workspace, size_t workspace_bytes, void* stream,
                                const suave_elbo_out_t* out, float* grads, const float* dlatent,
                                void* decoder_ready_event);

/* ---- the classifier head of behaviour="supervised" (row f4) -------------------------------------------------
 * ClassificationHead without hidden layers (suave/modules/heads.py:62-91: dropout on the latent inputs, one
 * nn.Linear, class-weighted F.cross_entropy) evaluated by the library, so that neither the head phase
 * (_train_head_phase, model.py:1743-1864) nor the joint step (model.py:1973-1999) leaves the stream for an
 * autograd framework.  Heads with hidden layers stay with the caller (suave_elbo_backward's dlatent).
 * C = n_classes in 2..64, D = latent_dim <= 256.  params = linear.weight [C, D] row-major followed by
 * linear.bias [C].  targets holds the class index of every TABLE row and is addressed by row id (batch row_index /
 * row_begin + i), like the packed table.  The loss of a call is CE = sum_i w[y_i] ce_i / weight_sum with
 * weight_sum = sum_i w[y_i] over the call's rows, or *weight_sum when given (data parallelism: the sum over the
 * GLOBAL batch, so that rank gradients add up to the one-GPU gradient).  Outputs: grads (if not NULL) =
 * loss_scale * d CE / d params; *loss_sum (if not NULL) += CE * loss_rows (the reference's
 * `epoch_loss += loss.item() * batch_count`, model.py:1802).  adam_lr > 0 applies torch.optim.Adam's update
 * (defaults betas (0.9, 0.999), eps 1e-8; adam_step 1-based) to params in the same launch that sums the gradient.
 * Dropout (training != 0) keeps input d of table row r when hash(r, dropout_key, d) >= dropout -- a stateless
 * lowbias32 hash, the same on every rank and for every sharding of a batch (suave_amd/supervised.py
 * RowKeyedDropout is its torch statement).  workspace: suave_head_workspace_bytes(C, D) bytes of device memory. */
typedef struct {
  int32_t n_classes;
  int32_t training;
  float* params;
  float* grads;
  const float* class_weight; /* [C] or NULL (unweighted) */
  const int32_t* targets;
  const float* weight_sum;   /* device scalar or NULL */
  float loss_scale;
  float loss_rows;
  double* loss_sum;          /* device scalar or NULL */
  float dropout;
  uint64_t dropout_key;
  float adam_lr;
  int64_t adam_step;
  float* adam_m;
  float* adam_v;
  void* workspace;
  size_t workspace_bytes;
} suave_head_t;
int suave_head_workspace_bytes(int32_t n_classes, int32_t latent_dim, size_t* bytes);

/* One step of the head phase on cached latents: row i of the batch is table row rid = row_index[i] (or
 * row_begin + i) and reads latent[rid * ld .. + D) and targets[rid].  logits_out [n, C] optional. */
int suave_head_step(const suave_head_t* head, int32_t latent_dim, const float* latent, int32_t ld,
                    const int32_t* row_index, int64_t row_begin, int64_t n, void* stream, float* logits_out);

/* The joint fine-tuning step in one pass (model.py:1973-1999: loss = ELBO loss + weight * CE(classifier(latent)),
 * one backward): suave_elbo_backward_overlap with the head's cross-entropy evaluated on the step's own `latent`
 * between the decoder and the encoder backward -- head->loss_scale * d CE / d latent joins the latent gradient,
 * head->grads receives the head's gradient.  Supervised branch only; decoder_ready_event may be NULL. */
int suave_elbo_joint_backward(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                              const suave_noise_t* noise, float beta_scale, float tau, int64_t global_rows,
                              void* workspace, size_t workspace_bytes, void* stream, const suave_elbo_out_t* out,
                              float* grads, const suave_head_t* head, void* decoder_ready_event);

/* Decoder.forward's per-column log-likelihoods at a GIVEN latent (decoder.py:333-379 as called by the reference's
 * decode-side code and tests): y_full = trunk(z), then every head scores the batch's cells with the row's
 * component as the assignment one-hot.  log_px / log_px_missing: [n, n_cols], laid out as the fields of that name in the ELBO output struct (either may
 * be NULL).  batch->n_rows rows; z [n, D]; component [n]. */
int suave_decode_log_px(const suave_plan* plan, const float* params, const float* z, const int32_t* component,
                        const suave_batch_t* batch, void* workspace, size_t workspace_bytes, void* stream,
                        float* log_px, float* log_px_missing);

/* torch.optim.Adam step (defaults betas=(0.9,0.999), eps=1e-8, no weight decay;
 * model.py:1443) on the flat buffers.  group_begin[g]..group_begin[g+1] is the
 * parameter range of learning-rate group g (model.py:1894-1909); lr_host has
 * n_groups entries.  step_t is the 1-based step count. */
int suave_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    const int32_t* group_begin_host, const float* lr_host, int32_t n_groups,
                    int64_t step_t, float beta1, float beta2, float eps, void* stream);

/* Encoder pass + mixture moment matching: SUAVE.encode /
 * _collect_posterior_statistics (encoder.py:60-72, model.py:2944-2964).
 * enc_out [B, K+2KD] (logits | mu | clamped logvar); post_mean, post_logvar
 * [B,D]; probs [B,K].  inv_temperature scales the logits (1 = none). */
int suave_encode(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                 float inv_temperature, void* workspace, size_t workspace_bytes, void* stream,
                 float* enc_out, float* post_mean, float* post_logvar, float* probs);

/* Encoder pass + the hard assignment of SUAVE.impute / SUAVE.encode(return_components=True)
 * (model.py:3627-3634, :4254-4258): component [B] = argmax_k softmax(logits * inv_temperature)
 * (first maximum) and z [B,D] = mu of that component -- the (latent, assignment) pair the
 * decoder receives under assignment_strategy="hard", without writing the [B, K+2KD] head
 * output to HBM for the caller to gather from. */
int suave_encode_hard(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                      float inv_temperature, void* workspace, size_t workspace_bytes, void* stream,
                      int32_t* component, float* z);

/* The word -> variate maps every in-kernel generator of this library uses, applied to caller-given
 * 32-bit words: uniform in (0,1) (never 0 or 1), Exp(1), N(0,1) (Box-Muller over words 2j, 2j+1;
 * normal[2j], normal[2j+1] are its two outputs).  No counterpart in the reference (torch's
 * generators draw these): it lets a test pin the maps' range at the extreme words, and a caller
 * that replays a Philox stream reproduce the library's draws.  Any output may be NULL. */
int suave_rng_draws(const uint32_t* words, int64_t n, void* stream, float* uniform, float* exponential,
                    float* normal);

/* Prior latent draw: sampling.sample_prepared_mixture_latents (sampling.py:38-56).
 * cdf [K] cumulative softmax(prior_logits); u [n] uniform and eps [n,D] normal
 * draws, or NULL to generate from (seed, row_offset + i). */
int suave_prior_latents(const suave_plan* plan, const float* params, const float* cdf,
                        const float* u, const float* eps, uint64_t seed, int64_t row_offset, int64_t n,
                        void* stream, float* z, int32_t* component);

/* Conditional latent draw: SUAVE._draw_conditional_latents (model.py:3042-3114)
 * without the per-row Python loop.  class_offsets [n_classes+1] / class_rows
 * index the cached training rows per class; labels [n]; train_probs [N,K],
 * train_mu / train_logvar [N,K,D]. */
int suave_conditional_latents(const suave_plan* plan, const int64_t* class_offsets,
                              const int64_t* class_rows, const int32_t* labels, const float* train_probs,
                              const float* train_mu, const float* train_logvar, const float* u_row,
                              const float* e_comp, const float* eps, uint64_t seed, int64_t row_offset,
                              int64_t n, void* stream, float* z, int32_t* component);

/* Decoder pass + per-column sample draw: Decoder.forward as used by
 * SUAVE.sample / SUAVE.impute (decoder.py:333-384, head samplers
 * distributions.py:67-80,112-117,130-134,167-174) followed by the value
 * conversion of sampling.decoder_outputs_to_frame (sampling.py:138-179).
 * z [n,D]; s [n,K] dense assignments (may be soft, model.py:3635-3644) or NULL
 * with component [n].  Output, columnar in the packed-table layout:
 * out_num float32 [n, ld_num] (real, pos, count as float), out_codes int16
 * [n, ld_disc].  noise_normal [n, n_cols], noise_uniform [n, n_cols],
 * noise_exp [n, total classes] inject draws (any may be NULL).  y_full_out
 * [n, y_rows] optionally receives decoder.y_projection's output (the input of
 * the heads, decoder.py:344) in the internal column layout -- NULL to skip. */
int suave_decode_sample(const suave_plan* plan, const float* params, const float* z, const float* s,
                        const int32_t* component, int64_t n, const float* noise_normal,
                        const float* noise_uniform, const float* noise_exp, uint64_t seed,
                        int64_t row_offset, void* workspace, size_t workspace_bytes, void* stream,
                        float* out_num, int32_t ld_num, int16_t* out_codes, int32_t ld_disc,
                        float* y_full_out);

/* suave_decode_sample for callers that cannot allocate -- no counterpart in the reference.  It is for code under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc: same arguments and results as
 * suave_decode_sample minus the workspace.  The call's scratch (device memory beyond the inputs and outputs) lives
 * in a file-scope device array of SUAVE_STATIC_SCRATCH_BYTES inside the library, one copy per GPU, resolved once per
 * device by suave_plan_create (hipGetSymbolAddress; if that fails the plan still works and only this entry point
 * reports the missing address) and held until the process ends.  The call makes no hipMalloc / hipFree and calls no
 * allocating routine (with suave_profile_enable(1) the event pool may grow: leave profiling off).
 * What the scratch holds, and so the row limit, is what the library carves for the call (carve_decode, api.hip):
 *   - component-index calls on a model inside the fused decode kernel's shapes (two hidden layers, no categorical
 *     column above 16 classes, default arithmetic): the decoder's fp16 weight images only -- no per-row scratch, any n;
 *   - every other call (dense assignments s, wide categorical columns, other depths): per row 4 bytes x (ldD + the
 *     hidden widths + ldY, + ldL logits of the wide categorical columns when the schema has any), every array rounded
 *     up to 256 bytes -- 14 979 rows for the 64-column benchmark model with dense assignments.
 * suave_decode_static_max_rows returns the limit for a plan and a kind of call (dense_assignments: s will be
 * non-NULL); a larger n fails with an error that names it.  Every call on one GPU shares that GPU's array, so calls
 * that overlap on one GPU (two streams, two threads) race: serialise them.  The array keeps the previous call's
 * contents; no kernel of this path needs it cleared. */
#define SUAVE_STATIC_SCRATCH_BYTES (32u << 20)
int suave_decode_sample_static(const suave_plan* plan, const float* params, const float* z, const float* s,
                               const int32_t* component, int64_t n, const float* noise_normal,
                               const float* noise_uniform, const float* noise_exp, uint64_t seed,
                               int64_t row_offset, void* stream, float* out_num, int32_t ld_num,
                               int16_t* out_codes, int32_t ld_disc, float* y_full_out);
int suave_decode_static_max_rows(const suave_plan* plan, int32_t dense_assignments, int64_t* max_rows);

/* Posterior draws for the Monte-Carlo predictive: SUAVE._posterior_draws
 * (model.py:3304-3336).  enc_out [n_rows, K+2KD] as written by suave_encode;
 * n_draws = L; e [L, n_rows, K] Exp(1) and eps [L, n_rows, D] N(0,1) inject the
 * draws (NULL: generated from (seed, (row_offset + b) * L + l) -- row_offset is the
 * global index of the first TABLE row, so draw l of a row does not depend on how a
 * caller chunks or shards the rows).  Outputs
 * z [L*n_rows, D] and component [L*n_rows], draw-major like the reference's
 * `.view(L * batch, ...)` (model.py:3368-3369). */
int suave_posterior_draws(const suave_plan* plan, const float* enc_out, int64_t n_rows, int32_t n_draws,
                          const float* e, const float* eps, uint64_t seed, int64_t row_offset, void* stream,
                          float* z, int32_t* component);
/* The same draws (same keys, same bits) laid out by TABLE ROW: draw l of row b is output row b * L + l, so a
 * row's draws are contiguous and a decode of them with row_offset * L keys every decoded row by (global table
 * row, draw) whatever the chunking or the sharding (impute_mc / predict_* of the Python host).  Injected e / eps
 * keep the reference's draw-major layout [L, n_rows, .]. */
int suave_posterior_draws_by_row(const suave_plan* plan, const float* enc_out, int64_t n_rows, int32_t n_draws,
                                 const float* e, const float* eps, uint64_t seed, int64_t row_offset, void* stream,
                                 float* z, int32_t* component);

/* Decoder pass + parameters of ONE head: the `params` of
 * SUAVE._posterior_predictive_params (model.py:3349-3386; head outputs
 * decoder.py:89,134-138,173,215,259).  out_params [n, width] with width from
 * suave_head_params_width: real {mean, var}; pos {mean_log, var_log, mean};
 * count {rate}; cat / ordinal {probs[C]}. */
int suave_head_params_width(const suave_plan* plan, int32_t col_index, int32_t* width);
int suave_decode_params(const suave_plan* plan, const float* params, const float* z, const float* s,
                        const int32_t* component, int64_t n, int32_t col_index, void* workspace,
                        size_t workspace_bytes, void* stream, float* out_params);

/* ---- data formats either side of the path ----------------------------
 *
 * suave_pack_columns replaces the host tensorisation of the reference --
 * SUAVE._prepare_training_tensors (suave/model.py:2784-2922, its per-row ordinal loop :2890-2891),
 * _apply_training_normalization (:3116-3213) and data.standardize (suave/data.py:133-266): the raw columns of
 * a table, one device array per column in whatever dtype the caller's frame holds them, become the packed table
 * (float32 numerics, int16 codes, uint8 mask) with the normalisation fused in.  Arithmetic as the reference's
 * numpy path: float64 (x - mean) / max(std, 1e-6), (log1p(x) - mean_log) / max(std_log, 1e-6), log(x + offset),
 * rounded once to float32; a non-finite result is a missing cell (value 0, mask 0).  Categorical columns arrive
 * as category codes (the mapping of labels to codes is pandas' and stays with the caller); ordinal columns as
 * numeric levels, out-of-range levels become missing.  rejected[c] (device, zeroed by the caller) counts the
 * values of column c that the reference rejects: pos < -1 and count < 0 (it raises ValueError), ordinal levels
 * outside [0, n_classes) (it warns and treats them as missing).  For a REAL column (which the reference never
 * rejects) rejected[c] counts the finite cells whose normalised value exceeds 60 000 in magnitude: such a table is
 * beyond the default arithmetic's fp16 planes and wants suave_gemm_arithmetic_set(3) (the Python host does that by
 * itself and logs it).  cols: DEVICE array of the plan's n_cols descriptors in schema order. */
#define SUAVE_RAW_F64 0
#define SUAVE_RAW_F32 1
#define SUAVE_RAW_I64 2
#define SUAVE_RAW_I32 3
#define SUAVE_RAW_I16 4
#define SUAVE_RAW_I8 5
typedef struct {
  const void* src;    /* device array of n raw values of this column */
  int32_t src_dtype;  /* SUAVE_RAW_* */
  int32_t kind;       /* SUAVE_TYPE_* of the column */
  int32_t n_classes;  /* cat / ordinal */
  int32_t slot;       /* x_index of the column: its position in xnum (real / pos / count) or codes (cat / ordinal) */
  double a, b;        /* real: mean, std;  pos: mean_log, std_log;  count: offset, unused */
} suave_pack_col_t;
int suave_pack_columns(const suave_plan* plan, const suave_pack_col_t* cols, int64_t n, float* xnum, int32_t ld_num,
                       int16_t* codes, int32_t ld_disc, uint8_t* mask, int32_t ld_mask, uint64_t* rejected, void* stream);

/* suave_unpack_columns replaces the per-column .cpu() / astype / concat of sampling.decoder_outputs_to_frame
 * (suave/sampling.py:138-179, model.py:4420-4431): the row-major outputs of suave_decode_sample become columnar
 * blocks [columns][ld_rows] that a DataFrame wraps without copying -- float32 (real, pos), int64 (count:
 * max(rint(v), 0)), int16 / int8 category codes.  Descriptor i moves source column `src` (of out_num for blocks
 * 0 and 1, of out_codes for 2 and 3) to row `dst` of its block.  cols: DEVICE array of n_out descriptors.  The
 * blocks may be rows [r0, r0 + n) of larger arrays: ld_rows is the row stride of a block column. */
typedef struct {
  int32_t block;  /* 0 float32, 1 int64 (counts), 2 int16 codes, 3 int8 codes */
  int32_t src;
  int32_t dst;
  int32_t reserved;
} suave_unpack_col_t;
int suave_unpack_columns(const suave_plan* plan, const suave_unpack_col_t* cols, int32_t n_out, int64_t n,
                         const float* out_num, int32_t ld_num, const int16_t* out_codes, int32_t ld_disc, float* f32_block,
                         int64_t* i64_block, int16_t* i16_block, int8_t* i8_block, int64_t ld_rows, void* stream);

/* Arithmetic of the matrix products (no counterpart in the reference, which
 * multiplies in fp32 through torch: suave/modules/encoder.py:49-53,
 * decoder.py:300-318).  All modes take fp32 in, accumulate in fp32 and give
 * fp32-grade results.  Mode 4 (default): every fp32
 * operand is cut into two fp16 planes, x = hi + lo * 2^-11, and three plane
 * products run on the fp16 MFMA; operands must stay within fp16's range,
 * |x| <= 65504 (a larger activation gives inf / NaN outputs, never a silently
 * wrong number).  Mode 3: three bf16 planes, six products on the bf16 MFMA
 * (fp32's full exponent range, twice the matrix-pipe time).  Mode 0: the
 * exact-fp32 MFMA everywhere.  Process-wide; the environment variable
 * SUAVE_GEMM_SPLIT gives the initial value.  Do not switch while a call of this
 * library is running on another thread. */
int suave_gemm_arithmetic_set(int mode);
int suave_gemm_arithmetic_get(int* mode);

/* Small-batch path of the ELBO entry points, suave_elbo_forward / _backward /
 * _backward_overlap (no counterpart in the reference, whose
 * default batches are 128 - 256 rows: suave/defaults.py:116-130).  An ELBO step
 * over at most `rows` (virtual) rows, in arithmetic mode 4, runs its layer GEMMs
 * on split-K kernels (a block per 32 x 64 output tile, the K range divided
 * among its four waves) and its likelihood kernel on a finer column chunk set;
 * larger steps, every other arithmetic mode and every inference entry point
 * (whose results must not depend on how the caller chunks the rows) take the
 * 128 x 128 tile kernels.  Results agree to fp32 summation order.  rows = 0
 * switches the path off; the default (and largest) value is 8192, or the
 * environment variable SUAVE_GEMM_SMALL_M.  Process-wide; size workspaces
 * (suave_workspace_bytes) after changing it, and do not switch while a call of
 * this library is running on another thread. */
int suave_small_batch_rows_set(int rows);
int suave_small_batch_rows_get(int* rows);

/* Measurement aid (no counterpart in the reference): when enabled, every kernel
 * launch of the library is bracketed by HIP events on its stream;
 * suave_profile_read synchronises and returns the summed device time (ms) and
 * launch count per kernel class since the last read.  Classes: 0 GEMM,
 * 1 likelihood+KL, 2 latent, 3 expand, 4 reduce, 5 Adam, 6 sample, 7 other. */
#define SUAVE_PROFILE_CLASSES 8
int suave_profile_enable(int on);
int suave_profile_read(double* ms_by_class, int64_t* launches_by_class, int32_t n_classes);

#ifdef __cplusplus
}
#endif
#endif /* SUAVE_HIP_H */
