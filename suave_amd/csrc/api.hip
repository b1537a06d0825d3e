// C-ABI entry points: plan construction, workspace carving and the launch
// sequences of the ELBO step / encode paths.  See include/suave_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "kernels.h"

namespace suave {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int resolve_static_scratch(suave_plan* p);  // defined beside the static scratch array below

// ---------------------------------------------------------------------------
// per-kernel-class timing (off by default; bench.py turns it on)
// ---------------------------------------------------------------------------
namespace {
struct ProfState {
  bool enabled = false;
  std::vector<hipEvent_t> start[kProfCount], stop[kProfCount];
  std::vector<hipEvent_t> pool;
} g_prof;
hipEvent_t prof_event() {
  if (!g_prof.pool.empty()) {
    hipEvent_t e = g_prof.pool.back();
    g_prof.pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

ProfScope::ProfScope(int c, hipStream_t s) : cls(c), stream(s), on(g_prof.enabled) {
  if (!on) return;
  hipEvent_t e = prof_event();
  if (!e) { on = false; return; }
  (void)hipEventRecord(e, stream);
  g_prof.start[cls].push_back(e);
}
ProfScope::~ProfScope() {
  if (!on) return;
  hipEvent_t e = prof_event();
  if (!e) { g_prof.pool.push_back(g_prof.start[cls].back()); g_prof.start[cls].pop_back(); return; }
  (void)hipEventRecord(e, stream);
  g_prof.stop[cls].push_back(e);
}

__global__ void k_onehot(const int32_t* kidx, int n_rows, int K, int ld, float* out) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)n_rows * K; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / K), j = (int)(e % K);
    out[(size_t)r * ld + j] = kidx[r] == j ? 1.f : 0.f;
  }
}

int launch_onehot(const int32_t* kidx, int n_rows, int K, int ld, float* out, hipStream_t s) {
  const int64_t n = (int64_t)n_rows * K;
  hipLaunchKernelGGL(k_onehot, dim3((int)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, kidx, n_rows, K, ld, out);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}


namespace {

// ---------------------------------------------------------------------------
// workspace carving: all scratch of one step, sized by the number of rows
// ---------------------------------------------------------------------------
struct Workspace {
  float *x, *encout, *ysoft, *z, *eps, *y, *dz, *dencout;
  float* he[SUAVE_MAX_HIDDEN];
  float* hd[SUAVE_MAX_HIDDEN];
  float* dhe[SUAVE_MAX_HIDDEN];
  float* dhd[SUAVE_MAX_HIDDEN];
  uint64_t* mask_e[SUAVE_MAX_HIDDEN];  // sign bits of he[i] / hd[i] (GemmArgs::mask_out), read by the data gradients
  uint64_t* mask_d[SUAVE_MAX_HIDDEN];
  int32_t* kidx;
  float *recon_part, *ds_part, *catkl, *gausskl, *gkl_logit, *gkl_mu, *gkl_lv, *prior_terms, *recon;
  float* onehot;
  int32_t* cells;  // [rows, n_cols] packed cells of the batch rows, mask folded in
  float *log_px = nullptr, *log_px_missing = nullptr;  // caller's per-column outputs of this call (not carved)
  // selected-component encoder head (supervised branch)
  float *sel, *dsel, *gslab;
  int32_t *perm, *lrank, *blk_cnt, *chunk_comp, *chunk_begin;
  int n_blocks, n_chunks_max;
  float* wideL;  // [rows, ldL] logits / logit gradients of the wide categorical columns
  float* wideT;  // row-major transposed copies of the wide heads' weights (WideCol::t_off), rewritten every backward
  int16_t* ecodes;  // [n_embed, rows] codes of the embedded columns (first encoder layer)
  float* wT[2 * SUAVE_MAX_HIDDEN + 2];  // row-major transposes of the layer weights, in all_layers() order
  float* slab;
  double* sums_slab;
  size_t slab_floats;
  size_t total_bytes;
};

struct Carver {
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// split count of a weight-gradient GEMM over `rows` batch rows
int tn_splits(const suave_plan* p, int n_out, int n_in1, int64_t rows) {
  const bool narrow = n_in1 <= 64;
  const int tiles = narrow ? ((n_out + 255) / 256) * ((n_in1 + 63) / 64) : ((n_out + 127) / 128) * ((n_in1 + 127) / 128);
  static const int factor = [] { const char* e = getenv("SUAVE_TN_FACTOR"); return e ? std::max(1, atoi(e)) : 1; }();
  static const int divisor = [] { const char* e = getenv("SUAVE_TN_DIV"); return e ? std::max(1, atoi(e)) : 1; }();
  // the first encoder layer's weight gradient has no data-gradient partner filling the CUs beside it
  static const int first = [] { const char* e = getenv("SUAVE_TN_FIRST"); return e ? std::max(1, atoi(e)) : 2; }();  // (1: +0.8 % on the S64 step)
  const suave_model_t& mm = p->model;
  const bool unpaired = mm.n_hidden > 0 && n_out == mm.enc[0].n_out && n_in1 == mm.enc[0].n_in;
  int64_t s = ((unpaired ? first : 1) * factor * (int64_t)p->n_cu / divisor + tiles - 1) / tiles;
  s = std::min<int64_t>(s, std::max<int64_t>(1, rows / 256));
  return (int)std::max<int64_t>(1, std::min<int64_t>(s, 512));
}

// blocks per column chunk of the likelihood kernel: about 28 independent wave workers per CU over all chunks
int loss_grid(const suave_plan* p, int64_t rows, int n_chunks = -1) {
  const int64_t tiles = (rows + 63) / 64;
  const int64_t nch = n_chunks > 0 ? n_chunks : std::max<size_t>(1, p->chunks_host.size());
  const int wpb = loss_workers_per_block();
  static const int wpc = [] { const char* e = getenv("SUAVE_LOSS_WPC"); return e ? std::max(1, atoi(e)) : 28; }();  // tuning knob (swept with SUAVE_LOSS_CHUNK_Y: scripts/sweep_loss.sh)
  const int64_t workers = std::min<int64_t>(tiles, (wpc * (int64_t)p->n_cu + nch - 1) / nch);
  return (int)std::max<int64_t>(1, (workers + wpb - 1) / wpb);
}
int loss_workers(const suave_plan* p, int64_t rows, int n_chunks = -1) { return loss_grid(p, rows, n_chunks) * loss_workers_per_block(); }

// Column chunks of the likelihood kernel for a step over `rows` (virtual) rows: the plan's chunks of ~24 y floats, or,
// for a small batch inside an ELBO step (gemm_small_rows), the finer set of ~8 -- a chunk is one wave's serial chain
// per 64 rows, and a few hundred rows leave the chip empty either way.
struct LossChunks { const suave::ColChunk* dev; int n; };
LossChunks loss_chunks(const suave_plan* p, int64_t rows) {
  if (!p->chunks_small_host.empty() && gemm_small_rows(rows))
    return LossChunks{p->chunks_dev + p->chunks_host.size(), (int)p->chunks_small_host.size()};
  return LossChunks{p->chunks_dev, (int)p->chunks_host.size()};
}

// split count of the grouped weight-gradient GEMM of the wide categorical columns
int wide_splits(const suave_plan* p, int64_t rows) {
  const int64_t tiles = std::max(1, p->wide_cum_total[2]);
  int64_t s = (2 * (int64_t)p->n_cu + tiles - 1) / tiles;
  s = std::min<int64_t>(s, std::max<int64_t>(1, rows / 256));
  return (int)std::max<int64_t>(1, std::min<int64_t>(s, 64));
}

struct LayerRef {
  const suave_layer_t* l;
};

std::vector<const suave_layer_t*> all_layers(const suave_plan* p) {
  std::vector<const suave_layer_t*> v;
  for (int i = 0; i < p->model.n_hidden; ++i) v.push_back(&p->model.enc[i]);
  v.push_back(&p->model.enc_head);
  for (int i = 0; i < p->model.n_hidden; ++i) v.push_back(&p->model.dec[i]);
  v.push_back(&p->model.y_proj);
  return v;
}

void carve(const suave_plan* p, int64_t rows, void* base, Workspace* w) {
  const suave_model_t& m = p->model;
  Carver c{static_cast<char*>(base)};
  const size_t R = (size_t)std::max<int64_t>(rows, 1);
  w->x = c.take<float>(R * p->ldI);
  w->encout = c.take<float>(R * p->ldE);
  w->ysoft = c.take<float>(R * p->ldK);
  w->z = c.take<float>(R * p->ldD);
  w->eps = c.take<float>(R * p->ldD * (m.behaviour == 1 ? 1 : 1));
  w->y = c.take<float>(R * p->ldY);
  w->dz = c.take<float>(R * p->ldD);
  w->dencout = c.take<float>(R * p->ldE);
  for (int i = 0; i < m.n_hidden; ++i) {
    w->he[i] = c.take<float>(R * p->ldH[i]);
    w->hd[i] = c.take<float>(R * p->ldH[i]);
    w->dhe[i] = c.take<float>(R * p->ldH[i]);
    w->dhd[i] = c.take<float>(R * p->ldH[i]);
    w->mask_e[i] = c.take<uint64_t>((size_t)gemm_mask_words((int64_t)R, m.enc[i].n_out));
    w->mask_d[i] = c.take<uint64_t>((size_t)gemm_mask_words((int64_t)R, m.dec[i].n_out));
  }
  w->kidx = c.take<int32_t>(R);
  // (sized for either chunk set when the batch is small enough for the fine one: no dependence on the call's scope)
  const bool fine_fits = rows <= gemm_small_rows_limit() && !p->chunks_small_host.empty();
  const size_t nch = fine_fits ? std::max(p->chunks_host.size(), p->chunks_small_host.size()) : p->chunks_host.size();
  const bool any_wide = !p->wide_host.empty();
  w->recon_part = c.take<float>((nch + 1) * R);
  w->ds_part = c.take<float>((nch + 1) * R * p->ldK);
  w->wideL = c.take<float>(any_wide ? R * p->ldL : 4);
  w->wideT = c.take<float>(any_wide ? (size_t)p->wide_t_len + 4 : 4);
  w->ecodes = c.take<int16_t>(std::max<size_t>(p->embed_host.size() * R, 8));
  {
    int li = 0;
    for (const suave_layer_t* l : all_layers(p)) w->wT[li++] = c.take<float>((size_t)l->n_in * round_up(l->n_out, 4) + 4);
  }
  w->catkl = c.take<float>(R);
  w->gausskl = c.take<float>(R);
  w->recon = c.take<float>(R);
  w->gkl_logit = c.take<float>(R * p->ldK);
  w->gkl_mu = c.take<float>(R * p->ldD);
  w->gkl_lv = c.take<float>(R * p->ldD);
  const int ldT = m.behaviour == 0 ? round_up(2 * m.latent_dim + m.n_components, 4)
                                   : round_up(m.n_components * m.latent_dim + m.latent_dim, 4);
  w->prior_terms = c.take<float>(R * ldT);
  w->onehot = c.take<float>(R * p->ldK);
  w->cells = c.take<int32_t>(((R + 63) / 64) * 64 * m.n_cols);
  {
    const int ldS = p->ldK + 2 * p->ldD;
    const bool sup = m.behaviour == 0;
    const size_t Rg = sup ? R : 1;
    w->n_blocks = (int)((Rg + 255) / 256);
    w->n_chunks_max = (int)((Rg + kGroupRows - 1) / kGroupRows) + m.n_components;
    w->sel = c.take<float>(Rg * ldS);
    w->dsel = c.take<float>(Rg * ldS);
    w->perm = c.take<int32_t>((size_t)w->n_chunks_max * kGroupRows);
    w->lrank = c.take<int32_t>(Rg);
    w->blk_cnt = c.take<int32_t>((size_t)w->n_blocks * m.n_components);
    w->chunk_comp = c.take<int32_t>(w->n_chunks_max);
    w->chunk_begin = c.take<int32_t>(m.n_components + 1);
    w->gslab = c.take<float>(sup ? (size_t)w->n_chunks_max * ldS * round_up(m.enc_head.n_in + 1, 4) : 1);
  }
  // slabs: every weight-gradient GEMM + the head slab + the prior slab
  size_t slab = 0;
  for (const suave_layer_t* l : all_layers(p)) {
    const int ldc = round_up(l->n_in + 1, 4);
    slab += (size_t)tn_splits(p, l->n_out, l->n_in, rows) * l->n_out * ldc;
  }
  const int lg = fine_fits ? std::max(loss_workers(p, rows), loss_workers(p, rows, (int)p->chunks_small_host.size())) : loss_workers(p, rows);
  slab += (size_t)lg * m.head_len + 4;
  {
    const int Mt = 2 * m.latent_dim + m.n_components, Nt = m.n_components;
    slab += (size_t)std::max(prior_colsum_blocks((int)rows), reparam_grid((int)rows)) * Mt * round_up(Nt + 1, 4);
  }
  if (m.behaviour == 1) slab += (size_t)lg * (m.n_components * m.latent_dim + m.latent_dim) + 4;
  if (any_wide) slab += (size_t)wide_splits(p, rows) * p->wide_len + 4;
  w->slab_floats = slab;
  w->slab = c.take<float>(slab);
  w->sums_slab = c.take<double>(((size_t)lg * nch + 1024 + 4096) * 4);
  w->total_bytes = (c.off + 255) & ~size_t(255);
  // (a decode call uses the head of the same workspace; its fused path needs the weight images whatever the row count)
  if (m.n_hidden == 2) w->total_bytes = std::max(w->total_bytes, (decode_fused_image_bytes(p) + 511) & ~size_t(255));
}

int check_batch(const suave_plan* p, const suave_batch_t* b) {
  SUAVE_REQUIRE(b != nullptr, "batch is null");
  SUAVE_REQUIRE(b->n_rows >= 0, "batch: negative row count");
  SUAVE_REQUIRE(b->mask != nullptr, "batch: mask is null");
  SUAVE_REQUIRE(p->model.n_num == 0 || b->xnum != nullptr, "batch: xnum is null");
  SUAVE_REQUIRE(p->model.n_disc == 0 || b->codes != nullptr, "batch: codes is null");
  SUAVE_REQUIRE(b->ld_num >= p->model.n_num && b->ld_disc >= p->model.n_disc && b->ld_mask >= p->model.n_cols,
                "batch: leading dimensions smaller than the schema");
  return 0;
}

GemmArgs gemm_base() {
  GemmArgs g;
  std::memset(&g, 0, sizeof(g));
  g.ones_col = -1;
  g.splits = 1;
  g.a_scale = 1.0f;
  return g;
}

// Arithmetic mode 4 (fp16 planes): gradient operands are ~1 / batch rows; the backward GEMMs multiply them by this
// power of two before the split (GemmArgs::a_scale) so that they stay clear of the fp16 subnormals.  Set per step.
thread_local float t_grad_scale = 1.0f;
float grad_scale_for(int64_t global_rows) {
  int e = 0;
  while (e < 16 && (int64_t(2) << e) <= global_rows) ++e;
  // 2^(floor(log2 rows) - 4): a per-row gradient d logp / dy of magnitude g arrives as ~g / 16, so the fp16 range is
  // only left beyond |g| ~ 1e6, while |g| down to 1e-3 still sits in fp16's normal range (and the scaled residual
  // plane keeps smaller ones to an absolute 1e-11)
  return std::ldexp(1.0f, std::max(e - 4, 0));
}

// forward layer: C = epi(A * W^T + b)
int fwd_layer(const suave_plan* p, const float* params, const suave_layer_t& l, const float* A, int lda, float* C,
              int ldc, int rows, bool relu_drop, float drop_p, const uint8_t* keep, const suave_batch_t* b,
              const suave_noise_t* nz, uint32_t stream_id, int vdiv, hipStream_t s, uint64_t* mask_out = nullptr) {
  GemmArgs g = gemm_base();
  g.mask_out = relu_drop ? mask_out : nullptr;
  g.A = A; g.B = params + l.w_off; g.C = C; g.bias = params + l.b_off;
  g.M = rows; g.N = l.n_out; g.K = l.n_in;
  g.lda = lda; g.ldb = l.ld; g.ldc = ldc;
  g.epi = relu_drop ? kEpiReluDrop : kEpiBias;
  g.drop_p = relu_drop ? drop_p : 0.f;
  g.keep = keep;
  g.row_ids = b->row_index; g.row_begin = b->row_begin; g.row_vdiv = vdiv;
  g.seed = nz ? nz->seed : 0;
  g.step = nz ? nz->step : 0;
  g.stream_id = stream_id;
  return launch_gemm(g, p->n_cu, s);
}

// data gradient: dA = (dC * W) [* relu/dropout mask of H]
GemmArgs bwd_data_args(const float* params, const suave_layer_t& l, const float* dC, int lddc, float* dA, int ldda, int rows,
                       const float* H, float drop_p, const float* wT = nullptr, const uint64_t* mask_in = nullptr) {
  GemmArgs g = gemm_base();
  g.mask_in = H ? mask_in : nullptr;
  g.A = dC; g.B = params + l.w_off; g.C = dA;
  g.M = rows; g.N = l.n_in; g.K = l.n_out;
  g.lda = lddc; g.ldb = l.ld; g.ldc = ldda;
  g.b_kmajor = 1;
  if (wT != nullptr && gemm_split_mode() != 0) {  // row-major W^T [n_in, ldT]: the split kernels' operand form
    g.B = wT; g.ldb = round_up(l.n_out, 4); g.b_kmajor = 0;
  }
  g.epi = H ? kEpiMaskBwd : kEpiBias;
  g.H = H;
  g.drop_p = drop_p;
  g.a_scale = t_grad_scale;
  return g;
}

// weight gradient into a slab + its reduction segments (weight, bias)
int bwd_weight_args(const suave_plan* p, const suave_layer_t& l, const float* dC, int lddc, const float* A, int lda,
                    int rows, float* slab_base, size_t* slab_off, ReduceArgs* red, int* max_elems, GemmArgs* out) {
  const int ldc = round_up(l.n_in + 1, 4);
  const int splits = tn_splits(p, l.n_out, l.n_in, rows);
  if (gemm_wgrad_direct_ok(rows, splits) && red->dst != nullptr && l.ld % 4 == 0 && l.w_off % 4 == 0) {
    // a few hundred rows, one split: the small-batch kernel writes the layer's gradient itself (no slab, no segment)
    GemmArgs g = gemm_base();
    g.a_scale = t_grad_scale;
    g.A = dC; g.B = A; g.C = red->dst + l.w_off; g.bias_out = red->dst + l.b_off;
    g.M = l.n_out; g.N = l.n_in; g.K = rows;
    g.lda = lddc; g.ldb = lda; g.ldc = l.ld;
    g.a_kmajor = 1; g.b_kmajor = 1;
    g.epi = kEpiSlab; g.splits = 1; g.slab_stride = 0; g.ones_col = l.n_in;
    *out = g;
    return 0;
  }
  GemmArgs g = gemm_base();
  g.a_scale = t_grad_scale;
  g.A = dC; g.B = A; g.C = slab_base + *slab_off;
  g.M = l.n_out; g.N = l.n_in; g.K = rows;
  g.lda = lddc; g.ldb = lda; g.ldc = ldc;
  g.a_kmajor = 1; g.b_kmajor = 1;
  g.epi = kEpiSlab;
  g.splits = splits;
  g.slab_stride = (int64_t)l.n_out * ldc;
  g.ones_col = l.n_in;
  SUAVE_REQUIRE(red->n_segs + 1 <= kMaxReduceSegs, "too many reduction segments");
  ReduceSeg w{};  // weight columns + the ones column (bias gradient) of every slab row
  w.src_off = (int64_t)*slab_off; w.slab_stride = g.slab_stride; w.n_slabs = splits;
  w.rows = l.n_out; w.cols = l.n_in; w.src_ld = ldc;
  w.dst_off = l.w_off; w.dst_si = l.ld; w.dst_sj = 1;
  w.bias_off1 = l.b_off + 1;
  red->segs[red->n_segs++] = w;
  *max_elems = std::max(*max_elems, l.n_out * l.n_in);
  *slab_off += (size_t)splits * l.n_out * ldc;
  *out = g;
  return 0;
}

int bwd_data(const suave_plan* p, const float* params, const suave_layer_t& l, const float* dC, int lddc, float* dA,
             int ldda, int rows, const float* H, float drop_p, hipStream_t s, const float* wT = nullptr) {
  return launch_gemm(bwd_data_args(params, l, dC, lddc, dA, ldda, rows, H, drop_p, wT), p->n_cu, s);
}

// index of a layer in all_layers() order (= index of its transposed copy in Workspace::wT)
int layer_index(const suave_plan* p, const suave_layer_t* l) {
  int i = 0;
  for (const suave_layer_t* q : all_layers(p)) {
    if (q == l) return i;
    ++i;
  }
  return -1;
}

TransArgs transpose_args(const suave_plan* p, const float* params, const Workspace& w) {
  TransArgs t{};
  if (gemm_split_mode() == 0) return t;
  int n = 0, max_tiles = 1;
  for (const suave_layer_t* l : all_layers(p)) {
    t.src[n] = params + l->w_off; t.dst[n] = w.wT[n];
    t.n_out[n] = l->n_out; t.n_in[n] = l->n_in; t.ld[n] = l->ld; t.ldT[n] = round_up(l->n_out, 4);
    max_tiles = std::max(max_tiles, ((l->n_in + 31) / 32) * ((l->n_out + 31) / 32));
    ++n;
  }
  t.n_mats = n;
  t.blocks_per_mat = std::min(max_tiles, 512);  // (a 256 x 256 layer is 64 tiles; blocks beyond a matrix's tiles exit at once)
  return t;
}

int bwd_weight(const suave_plan* p, const suave_layer_t& l, const float* dC, int lddc, const float* A, int lda,
               int rows, float* slab_base, size_t* slab_off, ReduceArgs* red, int* max_elems, hipStream_t s) {
  GemmArgs g;
  if (int rc = bwd_weight_args(p, l, dC, lddc, A, lda, rows, slab_base, slab_off, red, max_elems, &g)) return rc;
  return launch_gemm(g, p->n_cu, s);
}

// both backward GEMMs of layer l in one launch: dW (slab) from (dC, A_in) and dA_in from (dC, W)
int bwd_layer(const suave_plan* p, const float* params, const suave_layer_t& l, const float* dC, int lddc,
              const float* A_in, int lda_in, float* dA_in, int ldda, int rows, const float* H_in, float drop_p,
              float* slab_base, size_t* slab_off, ReduceArgs* red, int* max_elems, hipStream_t s,
              const float* wT = nullptr, const uint64_t* mask_in = nullptr) {
  GemmArgs w;
  if (int rc = bwd_weight_args(p, l, dC, lddc, A_in, lda_in, rows, slab_base, slab_off, red, max_elems, &w)) return rc;
  return launch_gemm_pair(bwd_data_args(params, l, dC, lddc, dA_in, ldda, rows, H_in, drop_p, wT, mask_in), w, p->n_cu, s);
}

WideRowArgs wide_row_args(const suave_plan* p, const Workspace& w, int rows) {
  WideRowArgs a{};
  a.wide = p->wide_dev; a.n_wide = (int)p->wide_host.size(); a.n_rows = rows; a.vdiv = 1;
  a.K = p->model.n_components; a.ldK = p->ldK; a.y = w.y; a.ldY = p->ldY; a.L = w.wideL; a.ldL = p->ldL;
  a.cells = w.cells; a.n_cols = p->model.n_cols;
  a.log_px = w.log_px; a.log_px_missing = w.log_px_missing;
  return a;
}

WideGemmArgs wide_gemm_args(const suave_plan* p, const float* params, const Workspace& w, int rows, int mode) {
  WideGemmArgs g{};
  g.cols = p->wide_dev; g.cum = p->wide_cum_dev[mode]; g.n_wide = (int)p->wide_host.size();
  g.cum_total = p->wide_cum_total[mode]; g.params = params; g.y = w.y; g.ldY = p->ldY; g.L = w.wideL; g.ldL = p->ldL;
  g.rows = rows; g.K = p->model.n_components; g.splits = 1; g.slab_stride = p->wide_len; g.wide_w0 = p->wide_w0;
  g.grad_scale = t_grad_scale;
  return g;
}

// Forward part of the wide categorical columns: assignment slots, logits GEMM, log-softmax (+ d logits).
int wide_forward(const suave_plan* p, const float* params, const Workspace& w, int rows, int vdiv, const int32_t* kidx,
                 const float* s_dense, const float* row_weight, float c_up, bool grad, float* recon_out,
                 double* sums_slab, hipStream_t s) {
  WideRowArgs a = wide_row_args(p, w, rows);
  a.vdiv = vdiv; a.kidx = kidx; a.s_dense = s_dense; a.row_weight = row_weight; a.c_up = c_up;
  a.recon_out = recon_out; a.sums_slab = sums_slab;
  if (int rc = launch_wide_fill(a, s)) return rc;
  if (int rc = launch_gemm_wide(wide_gemm_args(p, params, w, rows, 0), 0, s)) return rc;
  if (recon_out == nullptr) return 0;
  int wide_max_classes = 2;
  for (const suave::WideCol& wc : p->wide_host) wide_max_classes = std::max(wide_max_classes, (int)wc.C);
  return launch_wide_softmax(a, grad, wide_max_classes, s);
}

// Backward part: weight-gradient slabs (one reduction segment mirroring the weight region), data gradient into
// the y slots, assignment gradient out of the slots.
int wide_backward(const suave_plan* p, const float* params, const Workspace& w, int rows, float* ds_out, size_t* slab_off,
                  ReduceArgs* red, int* max_elems, hipStream_t s) {
  *slab_off = (*slab_off + 3) & ~size_t(3);
  WideGemmArgs g = wide_gemm_args(p, params, w, rows, 2);
  g.splits = wide_splits(p, rows);
  g.slab = w.slab + *slab_off;
  if (int rc = launch_gemm_wide(g, 2, s)) return rc;
  SUAVE_REQUIRE(red->n_segs + 1 <= kMaxReduceSegs, "too many reduction segments");
  ReduceSeg sg{};
  sg.src_off = (int64_t)*slab_off; sg.slab_stride = p->wide_len; sg.n_slabs = g.splits; sg.rows = 1; sg.cols = p->wide_len;
  sg.src_ld = p->wide_len; sg.dst_off = p->wide_w0; sg.dst_si = 0; sg.dst_sj = 1;
  red->segs[red->n_segs++] = sg;
  *max_elems = std::max(*max_elems, p->wide_len);
  *slab_off += (size_t)g.splits * p->wide_len;
  WideGemmArgs gd = wide_gemm_args(p, params, w, rows, 1);
  if (gemm_split_mode() != 0) {  // row-major W^T for the split kernels
    if (int rc = launch_wide_transpose(p->wide_dev, (int)p->wide_host.size(), params, w.wideT, s)) return rc;
    gd.wT = w.wideT;
  }
  if (int rc = launch_gemm_wide(gd, 1, s)) return rc;
  WideRowArgs a = wide_row_args(p, w, rows);
  a.ds_out = ds_out;
  return launch_wide_ds(a, s);
}

__global__ void k_sum_chunks(const float* part, int n_chunks, int n_rows, float* out) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n_rows; r += gridDim.x * 256) {
    float t = 0.f;
    for (int c = 0; c < n_chunks; ++c) t += part[(size_t)c * n_rows + r];
    out[r] = t;
  }
}


// Encoder hidden layers (encoder.py:46-53, :63).  With embedded columns the first layer is
//   relu/dropout(x_dense @ W1_dense^T + b + sum_j E_j[code_j])   (embed.hip)
int encoder_trunk_forward(const suave_plan* p, const float* params, const suave_batch_t* batch, const suave_noise_t* nz,
                          float drop_p, const Workspace& w, bool want_ecodes, const float** act_out, int* ld_out,
                          hipStream_t s) {
  const suave_model_t& m = p->model;
  const float* act = w.x;
  int ld_act = p->ldI;
  for (int i = 0; i < m.n_hidden; ++i) {
    const bool embed = i == 0 && !p->embed_host.empty();
    if (embed) {
      EmbedFwdArgs e{};
      e.cols = p->embed_dev; e.n_embed = (int)p->embed_host.size(); e.n_disc = m.n_disc; e.batch = *batch;
      e.n_rows = batch->n_rows; e.params = params; e.ld = p->ldH[0]; e.pre = w.he[0];
      e.ecodes = want_ecodes ? w.ecodes : nullptr;
      if (int rc = launch_embed_fwd(e, s)) return rc;
    }
    GemmArgs g = gemm_base();
    const suave_layer_t& l = m.enc[i];
    g.A = act; g.B = params + l.w_off; g.C = w.he[i]; g.bias = params + l.b_off;
    g.M = batch->n_rows; g.N = l.n_out; g.K = l.n_in;
    g.lda = ld_act; g.ldb = l.ld; g.ldc = p->ldH[i];
    g.epi = kEpiReluDrop;
    g.H = embed ? w.he[0] : nullptr;
    g.drop_p = drop_p;
    g.keep = nz ? nz->enc_keep[i] : nullptr;
    g.row_ids = batch->row_index;
    g.row_begin = batch->row_begin;
    g.seed = nz ? nz->seed : 0;
    g.step = nz ? nz->step : 0;
    g.stream_id = kStreamEncDrop + i;
    g.mask_out = gemm_small_rows(batch->n_rows) ? nullptr : w.mask_e[i];
    if (int rc = launch_gemm(g, p->n_cu, s)) return rc;
    act = w.he[i];
    ld_act = p->ldH[i];
  }
  *act_out = act;
  *ld_out = ld_act;
  return 0;
}

int encoder_embed_backward(const suave_plan* p, const Workspace& w, int rows, float* grads, hipStream_t s) {
  if (p->embed_host.empty()) return 0;
  EmbedBwdArgs e{};
  e.cols = p->embed_dev; e.n_embed = (int)p->embed_host.size(); e.n_rows = rows; e.ecodes = w.ecodes;
  e.dpre = w.dhe[0]; e.ld = p->ldH[0]; e.grads = grads; e.n_classes_total = p->embed_classes_total;
  return launch_embed_bwd(e, s);
}


// Unsupervised (expectation-form) ELBO: SUAVE._forward_elbo_batch with behaviour="unsupervised"
// (model.py:1576-1580, :1619-1654).  The decoder runs on B*K virtual rows (row v = b*K + k uses the
// latent of component k and the one-hot assignment k); tau <= 0 means "temperature is None".
__global__ void k_add_rows(float* dst, int ld_dst, const float* src, int ld_src, int n_rows, int n_cols) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)n_rows * n_cols; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n_cols), c = (int)(e % n_cols);
    dst[(size_t)r * ld_dst + c] += src[(size_t)r * ld_src + c];
  }
}

int elbo_unsup_impl(const suave_plan* p, const float* params, const suave_batch_t* batch, const suave_noise_t* noise,
                    float beta_scale, float tau, int training, int64_t global_rows, void* workspace,
                    size_t workspace_bytes, hipStream_t s, const suave_elbo_out_t* out, float* grads,
                    hipEvent_t ready_event = nullptr) {
  const suave_model_t& m = p->model;
  const bool grad = grads != nullptr;
  const int B = batch->n_rows, K = m.n_components, D = m.latent_dim;
  if (B == 0) return 0;
  const int64_t V64 = (int64_t)B * K;
  SUAVE_REQUIRE(V64 < (int64_t(1) << 31), "unsupervised step: B*K too large");
  const int V = (int)V64;
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0, "params must be 16-byte aligned");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  Workspace w;
  carve(p, V, workspace, &w);
  SUAVE_REQUIRE(workspace_bytes >= w.total_bytes, "workspace too small (size it for B*K rows in the unsupervised branch)");
  const float drop_p = training ? m.dropout : 0.f;
  const int nh = m.n_hidden;
  suave_noise_t nz_default;
  std::memset(&nz_default, 0, sizeof(nz_default));
  const suave_noise_t* nz = noise ? noise : &nz_default;
  const float tau_c = tau > 0.f ? std::max(tau, 1e-6f) : 0.f;

  {  // (+ the transposed weight copies of the backward, in the same launch)
    const TransArgs trans = grad ? transpose_args(p, params, w) : TransArgs{};
    if (int rc = launch_expand(p, *batch, w.x, w.cells, s, grad ? &trans : nullptr)) return rc;
  }
  const float* act = w.x;
  int ld_act = p->ldI;
  if (int rc = encoder_trunk_forward(p, params, batch, nz, drop_p, w, grad, &act, &ld_act, s)) return rc;
  if (int rc = fwd_layer(p, params, m.enc_head, act, ld_act, w.encout, p->ldE, B, false, 0.f, nullptr, batch, nz, 0, 1, s))
    return rc;
  float* wflat = w.gkl_mu;  // [V]
  {
    MixFwdArgs a{};
    a.encout = w.encout; a.ldE = p->ldE; a.K = K; a.D = D; a.ldK = p->ldK; a.ldD = p->ldD; a.n_rows = B;
    a.row_index = batch->row_index; a.row_begin = batch->row_begin; a.exp_noise = nz->exp_noise; a.eps_in = nz->eps;
    a.seed = nz->seed; a.step = nz->step; a.tau = tau_c; a.w = w.ysoft; a.wflat = wflat; a.z_all = w.z;
    a.eps_buf = w.eps; a.latent_out = out ? out->latent : nullptr; a.rb = mix_rows_per_block(B);
    if (int rc = launch_mix_fwd(a, s)) return rc;
  }
  act = w.z;
  ld_act = p->ldD;
  for (int i = 0; i < nh; ++i) {
    if (int rc = fwd_layer(p, params, m.dec[i], act, ld_act, w.hd[i], p->ldH[i], V, true, drop_p, nz->dec_keep[i], batch,
                           nz, kStreamDecDrop + i, K, s))
      return rc;
    act = w.hd[i];
    ld_act = p->ldH[i];
  }
  if (int rc = fwd_layer(p, params, m.y_proj, act, ld_act, w.y, p->ldY, V, false, 0.f, nullptr, batch, nz, 0, K, s))
    return rc;
  const LossChunks lch = loss_chunks(p, V);
  const int n_chunks = lch.n;
  const int lgrid = loss_grid(p, V, n_chunks);
  const int lworkers = loss_workers(p, V, n_chunks);
  const int ldT = round_up(K * D + D, 4);
  size_t slab_off = 0;
  float* head_slab = w.slab;
  slab_off += ((size_t)lworkers * m.head_len + 3) & ~size_t(3);  // GEMM slabs start 16-byte aligned
  {
    LossArgs a{};
    a.cols = p->cols_dev; a.chunks = lch.dev; a.params = params; a.y = w.y; a.batch = *batch;
    a.n_rows = V; a.vdiv = K; a.kidx = nullptr; a.row_weight = wflat;
    a.c_up = -1.0f / (float)global_rows; a.bk = beta_scale / (float)global_rows;
    a.K = K; a.D = D; a.ldY = p->ldY; a.ldy = p->max_chunk_y | 1; a.ldK = p->ldK; a.ldD = p->ldD; a.ldE = p->ldE;
    a.ldT = ldT; a.max_classes = p->max_classes; a.recon_part = w.recon_part; a.ds_part = nullptr; a.do_kl = 0;
    a.head_slab = grad ? head_slab : nullptr; a.sums_slab = nullptr;
    a.head_off = m.head_off; a.head_len = m.head_len; a.hw_max = p->hw_max;
    a.xsrc = p->xsrc_dev; a.ldx = p->loss_ldx; a.ldm = p->loss_ldm; a.cells = w.cells; a.n_cols = m.n_cols;
    if (n_chunks > 0)
      if (int rc = launch_loss(a, grad, p->kmax, p->loss_wide16 != 0, lgrid, n_chunks, p->loss_lds_bytes, s)) return rc;
  }
  const int n_wide = (int)p->wide_host.size();
  const int n_parts = n_chunks + (n_wide ? 1 : 0);
  if (n_wide)
    if (int rc = wide_forward(p, params, w, V, K, nullptr, nullptr, wflat, -1.0f / (float)global_rows, grad,
                              w.recon_part + (size_t)n_chunks * V, nullptr, s))
      return rc;
  ReduceArgs red{};
  red.n_segs = 0; red.slab = w.slab; red.dst = grads;
  int max_elems = 256;
  if (grad) {
    if (n_wide)
      if (int rc = wide_backward(p, params, w, V, nullptr, &slab_off, &red, &max_elems, s)) return rc;
    const float* hlast = nh ? w.hd[nh - 1] : w.z;
    const int ldh = nh ? p->ldH[nh - 1] : p->ldD;
    if (int rc = bwd_weight(p, m.y_proj, w.y, p->ldY, hlast, ldh, V, w.slab, &slab_off, &red, &max_elems, s)) return rc;
    const float* dcur = w.y;
    int lddcur = p->ldY;
    const suave_layer_t* lcur = &m.y_proj;
    for (int i = nh - 1; i >= 0; --i) {
      if (int rc = bwd_data(p, params, *lcur, dcur, lddcur, w.dhd[i], p->ldH[i], V, w.hd[i], drop_p, s, w.wT[layer_index(p, lcur)])) return rc;
      const float* ain = i > 0 ? w.hd[i - 1] : w.z;
      const int ldain = i > 0 ? p->ldH[i - 1] : p->ldD;
      if (int rc = bwd_weight(p, m.dec[i], w.dhd[i], p->ldH[i], ain, ldain, V, w.slab, &slab_off, &red, &max_elems, s))
        return rc;
      dcur = w.dhd[i];
      lddcur = p->ldH[i];
      lcur = &m.dec[i];
    }
    if (int rc = bwd_data(p, params, *lcur, dcur, lddcur, w.dz, p->ldD, V, nullptr, 0.f, s, w.wT[layer_index(p, lcur)])) return rc;
  }
  const int mix_rb = mix_rows_per_block(B);
  const int fgrid = std::max(1, std::min((B + mix_rb - 1) / mix_rb, 2 * p->n_cu));
  {
    MixFinArgs a{};
    a.encout = w.encout; a.ldE = p->ldE; a.K = K; a.D = D; a.ldK = p->ldK; a.ldD = p->ldD; a.n_rows = B;
    a.n_chunks = n_parts; a.ldT = ldT; a.params = params; a.pm_w_off = m.pm_w_off; a.pm_b_off = m.pm_b_off;
    a.w = w.ysoft; a.eps_buf = w.eps; a.recon_part = w.recon_part; a.dz_all = w.dz; a.tau = tau_c;
    a.c_up = -1.0f / (float)global_rows; a.bk = beta_scale / (float)global_rows;
    a.recon = w.recon; a.catkl = w.catkl; a.gausskl = w.gausskl; a.dencout = w.dencout; a.prior_terms = w.prior_terms;
    a.sums_slab = w.sums_slab; a.rb = mix_rb;
    if (int rc = launch_mix_finalize(a, grad, fgrid, s)) return rc;
  }
  red.sums_slab = w.sums_slab;
  red.n_sum_slabs = fgrid;
  red.sums_out = out ? out->sums : nullptr;
  red.n_rows = (double)B;
  if (grad) {
    // PriorMean gradients: column sums of prior_terms -> linear.weight[d,k] and linear.bias[d]
    {
      const int ncol = K * D + D;
      const int nb = std::max(1, std::min(lgrid, (B + 31) / 32));  // (the slab holds loss_workers >= lgrid of them)
      if (int rc = launch_colsum(w.prior_terms, ldT, B, ncol, w.slab + slab_off, nb, s)) return rc;
      SUAVE_REQUIRE(red.n_segs + 3 <= kMaxReduceSegs, "too many reduction segments");
      ReduceSeg sw{};
      sw.src_off = (int64_t)slab_off; sw.slab_stride = ncol; sw.n_slabs = nb; sw.rows = K; sw.cols = D; sw.src_ld = D;
      sw.dst_off = m.pm_w_off; sw.dst_si = 1; sw.dst_sj = K;
      red.segs[red.n_segs++] = sw;
      ReduceSeg sb = sw;
      sb.src_off += (int64_t)K * D; sb.rows = 1; sb.cols = D; sb.dst_off = m.pm_b_off; sb.dst_si = 0; sb.dst_sj = 1;
      red.segs[red.n_segs++] = sb;
      slab_off += ((size_t)nb * ncol + 3) & ~size_t(3);  // the encoder's GEMM slabs follow: 16-byte aligned
    }
    ReduceSeg sh{};
    sh.src_off = 0; sh.slab_stride = m.head_len; sh.n_slabs = lworkers / loss_workers_per_block(); sh.rows = 1; sh.cols = m.head_len;  // one row per block
    sh.src_ld = m.head_len; sh.dst_off = m.head_off; sh.dst_si = 0; sh.dst_sj = 1;
    red.segs[red.n_segs++] = sh;
    max_elems = std::max(max_elems, m.head_len);
    // The decoder-side bucket (decoder, likelihood heads, PriorMean: one contiguous range of the flat gradient) is
    // complete here: with an event to signal, reduce it now and record the event, so that a data-parallel caller
    // all-reduces it while the encoder backward below is still running (as the supervised branch does).
    if (ready_event) {
      if (int rc = launch_reduce(red, max_elems, s)) return rc;
      SUAVE_CHECK_HIP(hipEventRecord(ready_event, s));
      red.n_segs = 0;
      red.sums_slab = nullptr; red.n_sum_slabs = 0; red.sums_out = nullptr;  // the epoch sums went with the first launch
      max_elems = 256;
    }
    const float* helast = nh ? w.he[nh - 1] : w.x;
    const int ldhe = nh ? p->ldH[nh - 1] : p->ldI;
    if (int rc = bwd_weight(p, m.enc_head, w.dencout, p->ldE, helast, ldhe, B, w.slab, &slab_off, &red, &max_elems, s))
      return rc;
    const float* dcur = w.dencout;
    int lddcur = p->ldE;
    const suave_layer_t* lcur = &m.enc_head;
    for (int i = nh - 1; i >= 0; --i) {
      if (int rc = bwd_data(p, params, *lcur, dcur, lddcur, w.dhe[i], p->ldH[i], B, w.he[i], drop_p, s, w.wT[layer_index(p, lcur)])) return rc;
      const float* ain = i > 0 ? w.he[i - 1] : w.x;
      const int ldain = i > 0 ? p->ldH[i - 1] : p->ldI;
      if (int rc = bwd_weight(p, m.enc[i], w.dhe[i], p->ldH[i], ain, ldain, B, w.slab, &slab_off, &red, &max_elems, s))
        return rc;
      dcur = w.dhe[i];
      lddcur = p->ldH[i];
      lcur = &m.enc[i];
    }
    if (nh > 0)
      if (int rc = encoder_embed_backward(p, w, B, grads, s)) return rc;
    SUAVE_REQUIRE(slab_off <= w.slab_floats, "internal: slab overflow");
  }
  if (grad || red.sums_out) {
    if (int rc = launch_reduce(red, max_elems, s)) return rc;
  }
  if (out) {
    if (out->reconstruction)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->reconstruction, w.recon, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->categorical_kl)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->categorical_kl, w.catkl, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->gaussian_kl)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->gaussian_kl, w.gausskl, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->enc_out)
      if (int rc = launch_posterior_stats(w.encout, p->ldE, K, D, B, 1.0f, out->enc_out, nullptr, nullptr, nullptr, s))
        return rc;
  }
  return 0;
}

int elbo_impl(const suave_plan* p, const float* params, const suave_batch_t* batch, const suave_noise_t* noise,
              float beta_scale, float tau, int training, int64_t global_rows, void* workspace,
              size_t workspace_bytes, hipStream_t s, const suave_elbo_out_t* out, float* grads, const float* dlatent,
              hipEvent_t ready_event = nullptr, const suave_head_t* head = nullptr) {
  SUAVE_REQUIRE(p != nullptr && params != nullptr, "plan/params is null");
  if (int rc = check_batch(p, batch)) return rc;
  const suave_model_t& m = p->model;
  SUAVE_REQUIRE(dlatent == nullptr || m.behaviour == 0, "dlatent is only defined for the supervised branch (model.py:1987-1991)");
  SUAVE_REQUIRE(global_rows >= 1, "global_rows must be positive");
  t_grad_scale = grad_scale_for(global_rows);
  GemmSmallScope small_scope;
  SUAVE_REQUIRE(head == nullptr || (m.behaviour == 0 && grads != nullptr), "the classifier head joins the supervised backward only");
  if (batch->n_rows == 0) {  // an empty shard still owes its gradient (zero) and its event to the all-reduce
    if (grads) SUAVE_CHECK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * m.n_params, s));
    if (head)
      if (int rc = launch_head_loss(*head, m.latent_dim, nullptr, 0, 0, batch->row_index, batch->row_begin, 0, nullptr, 0, nullptr, s))
        return rc;
    if (ready_event) SUAVE_CHECK_HIP(hipEventRecord(ready_event, s));
    return 0;
  }
  SUAVE_REQUIRE(m.behaviour == 0 || out == nullptr || (out->log_px == nullptr && out->log_px_missing == nullptr),
                "per-column log_px / log_px_missing are outputs of the supervised branch (the unsupervised one scores "
                "every column once per component)");
  if (m.behaviour == 1) {
    // (the PriorMean gradient comes out of the mixture finalisation, which only needs the decoder backward: the
    // decoder-side bucket is reduced and the event recorded before the encoder backward, as in the supervised branch)
    if (int rc = elbo_unsup_impl(p, params, batch, noise, beta_scale, tau, training, global_rows, workspace, workspace_bytes,
                                 s, out, grads, grads ? ready_event : nullptr))
      return rc;
    if (!grads && ready_event) SUAVE_CHECK_HIP(hipEventRecord(ready_event, s));
    return 0;
  }
  const bool grad = grads != nullptr;
  const int B = batch->n_rows;
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0, "params must be 16-byte aligned");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  Workspace w;
  carve(p, B, workspace, &w);
  SUAVE_REQUIRE(workspace_bytes >= w.total_bytes, "workspace too small (see suave_workspace_bytes)");
  if (out) { w.log_px = out->log_px; w.log_px_missing = out->log_px_missing; }
  const float tau_c = std::max(tau, 1e-6f);  // model.py:1587
  const float drop_p = training ? m.dropout : 0.f;
  const int nh = m.n_hidden;
  suave_noise_t nz_default;
  std::memset(&nz_default, 0, sizeof(nz_default));
  const suave_noise_t* nz = noise ? noise : &nz_default;

  // ---- encoder ----
  {  // (+ the transposed weight copies of the backward, in the same launch)
    const TransArgs trans = grad ? transpose_args(p, params, w) : TransArgs{};
    if (int rc = launch_expand(p, *batch, w.x, w.cells, s, grad ? &trans : nullptr)) return rc;
  }
  const float* act = w.x;
  int ld_act = p->ldI;
  if (int rc = encoder_trunk_forward(p, params, batch, nz, drop_p, w, grad, &act, &ld_act, s)) return rc;
  // ---- encoder head, selected component only ----
  // Only the logits and the selected component's mu / logvar enter the supervised loss (model.py:1587-1598,
  // losses.py:60-84), so the head is: logits GEMM -> assignment -> stable sort of the rows by component ->
  // one grouped GEMM whose row tiles each use the 2*D weight rows of their component.
  const int K = m.n_components, D = m.latent_dim;
  const int ldS = p->ldK + 2 * p->ldD;
  const float* hlast_e = act;
  const int ldhe_last = ld_act;
  const int Vg = w.n_chunks_max * kGroupRows;
  LatentFwdArgs la{};
  la.sel = w.sel; la.ldS = ldS; la.K = K; la.D = D; la.ldK = p->ldK; la.ldD = p->ldD;
  la.n_rows = B; la.row_index = batch->row_index; la.row_begin = batch->row_begin;
  la.exp_noise = nz->exp_noise; la.eps_in = nz->eps; la.seed = nz->seed; la.step = nz->step; la.tau = tau_c;
  la.ysoft = w.ysoft; la.kidx = w.kidx; la.z = w.z;
  la.latent_out = out ? out->latent : nullptr;
  la.comp_out = out ? out->component : nullptr;
  la.lrank = w.lrank; la.blk_cnt = w.blk_cnt;
  la.perm_fill = w.perm; la.perm_fill_n = w.n_chunks_max * kGroupRows;
  const bool one_block = w.n_blocks == 1;  // <= 256 rows: k_assign sorts as well
  if (one_block) { la.one_chunk_comp = w.chunk_comp; la.one_chunk_begin = w.chunk_begin; la.one_chunks_max = w.n_chunks_max; }
  GroupArgs ga{};
  ga.K = K; ga.n_rows = B; ga.n_blocks = w.n_blocks; ga.n_chunks_max = w.n_chunks_max; ga.kidx = w.kidx;
  ga.lrank = w.lrank; ga.blk_cnt = w.blk_cnt; ga.chunk_comp = w.chunk_comp;
  ga.chunk_begin = w.chunk_begin; ga.perm = w.perm;
  auto head_geometry = [&](GemmArgs& g) {
    g.perm = w.perm; g.chunk_comp = w.chunk_comp; g.hK = K; g.hD = D; g.hldK = p->ldK; g.hldD = p->ldD;
  };
  {
    GemmArgs g = gemm_base();  // logits = h * W[0:K]^T + b[0:K]  -> sel[:, 0:K]
    g.A = hlast_e; g.B = params + m.enc_head.w_off; g.C = w.sel; g.bias = params + m.enc_head.b_off;
    g.M = B; g.N = K; g.K = m.enc_head.n_in; g.lda = ldhe_last; g.ldb = m.enc_head.ld; g.ldc = ldS;
    g.epi = kEpiBias;
    if (int rc = launch_gemm(g, p->n_cu, s)) return rc;
  }
  if (int rc = launch_assign(la, s)) return rc;
  if (!one_block)
    if (int rc = launch_group(ga, s)) return rc;
  {
    GemmArgs g = gemm_base();  // [mu_k | logvar_k] of each row's own component -> sel[:, ldK:]
    g.A = hlast_e; g.B = params + m.enc_head.w_off; g.C = w.sel + p->ldK; g.bias = params + m.enc_head.b_off;
    g.M = Vg; g.N = 2 * p->ldD; g.K = m.enc_head.n_in; g.lda = ldhe_last; g.ldb = m.enc_head.ld; g.ldc = ldS;
    g.epi = kEpiBias; g.grouped = 1; g.q0 = p->ldK;
    head_geometry(g);
    if (int rc = launch_gemm(g, p->n_cu, s)) return rc;
  }
  const LossChunks lch = loss_chunks(p, B);
  const int n_chunks = lch.n;
  const int lgrid = loss_grid(p, B, n_chunks);
  const int lworkers = loss_workers(p, B, n_chunks);
  const int ldT = round_up(2 * m.latent_dim + m.n_components, 4);
  la.params = params; la.prior_logits_off = m.prior_logits_off; la.prior_mu_off = m.prior_mu_off;
  la.prior_logvar_off = m.prior_logvar_off; la.bk = beta_scale / (float)global_rows; la.grad = grad ? 1 : 0; la.ldT = ldT;
  la.catkl = w.catkl; la.gausskl = w.gausskl; la.gkl_logit = w.gkl_logit;
  la.prior_terms = w.prior_terms; la.sums_slab = w.sums_slab + (size_t)lworkers * n_chunks * 4;
  // The prior-gradient column sums are taken inside k_reparam (slab below: no prior_terms round trip, no
  // k_prior_colsum launch) unless K x D is beyond its LDS tile; k_latent_bwd recomputes eps and the Gaussian-KL
  // gradients instead of reading three [B, ldD] buffers back.
  const bool fuse_colsum = grad && reparam_colsum_ok(K, D);
  const int prior_Mt = 2 * m.latent_dim + m.n_components, prior_ldc = round_up(m.n_components + 1, 4);
  size_t slab_off = 0;
  float* head_slab = w.slab;
  slab_off += ((size_t)lworkers * m.head_len + 3) & ~size_t(3);  // GEMM slabs start 16-byte aligned
  const size_t prior_slab_off = slab_off;  // [reparam_grid(B)][2D + K][prior_ldc], right behind the head slab
  if (fuse_colsum) {
    la.prior_slab = w.slab + prior_slab_off; la.prior_ldc = prior_ldc; la.prior_terms = nullptr;
    slab_off += (size_t)reparam_grid(B) * prior_Mt * prior_ldc;
  }
  if (int rc = launch_reparam(la, s)) return rc;
  // ---- decoder trunk ----
  act = w.z;
  ld_act = p->ldD;
  for (int i = 0; i < nh; ++i) {
    if (int rc = fwd_layer(p, params, m.dec[i], act, ld_act, w.hd[i], p->ldH[i], B, true, drop_p, nz->dec_keep[i], batch,
                           nz, kStreamDecDrop + i, 1, s, (grad && !gemm_small_rows(B)) ? w.mask_d[i] : nullptr))
      return rc;
    act = w.hd[i];
    ld_act = p->ldH[i];
  }
  if (int rc = fwd_layer(p, params, m.y_proj, act, ld_act, w.y, p->ldY, B, false, 0.f, nullptr, batch, nz, 0, 1, s))
    return rc;
  // the forward launches leave sign bits of the hidden activations for the data gradients
  static const bool sign_bits = getenv("SUAVE_NO_SIGN_BITS") == nullptr;  // measurement knob: read the activations instead
  const bool dec_masks = sign_bits && !gemm_small_rows(B);
  const bool enc_masks = sign_bits && !gemm_small_rows(B);
  // ---- fused likelihood (+ gradients) ----
  {
    LossArgs a{};
    a.cols = p->cols_dev; a.chunks = lch.dev; a.params = params; a.y = w.y; a.batch = *batch;
    a.n_rows = B; a.vdiv = 1; a.kidx = w.kidx; a.row_weight = nullptr;
    a.c_up = -1.0f / (float)global_rows;
    a.bk = beta_scale / (float)global_rows;
    a.K = m.n_components; a.D = m.latent_dim; a.ldY = p->ldY; a.ldy = p->max_chunk_y | 1; a.ldK = p->ldK;
    a.ldD = p->ldD; a.ldE = p->ldE; a.ldT = ldT; a.max_classes = p->max_classes;
    a.recon_part = w.recon_part; a.ds_part = w.ds_part; a.do_kl = 0;
    a.prior_logits_off = m.prior_logits_off; a.prior_mu_off = m.prior_mu_off; a.prior_logvar_off = m.prior_logvar_off;
    a.catkl = w.catkl; a.gausskl = w.gausskl; a.gkl_logit = w.gkl_logit; a.gkl_mu = w.gkl_mu; a.gkl_lv = w.gkl_lv;
    a.prior_terms = w.prior_terms; a.head_slab = grad ? head_slab : nullptr; a.sums_slab = w.sums_slab;
    a.head_off = m.head_off; a.head_len = m.head_len; a.hw_max = p->hw_max;
    a.xsrc = p->xsrc_dev; a.ldx = p->loss_ldx; a.ldm = p->loss_ldm; a.cells = w.cells; a.n_cols = m.n_cols;
    a.log_px = w.log_px; a.log_px_missing = w.log_px_missing;
    if (n_chunks > 0)
      if (int rc = launch_loss(a, grad, p->kmax, p->loss_wide16 != 0, lgrid, n_chunks, p->loss_lds_bytes, s)) return rc;
  }
  const int n_wide = (int)p->wide_host.size();
  const int n_parts = n_chunks + (n_wide ? 1 : 0);
  if (n_wide)
    if (int rc = wide_forward(p, params, w, B, 1, w.kidx, nullptr, nullptr, -1.0f / (float)global_rows, grad,
                              w.recon_part + (size_t)n_chunks * B,
                              w.sums_slab + ((size_t)lworkers * n_chunks + reparam_grid(B)) * 4, s))
      return rc;
  const float* recon_final = w.recon_part;
  if (n_parts > 1 && out && out->reconstruction) {  // per-row reconstruction only on request (the sums have their own slabs)
    hipLaunchKernelGGL(k_sum_chunks, dim3(std::min((B + 255) / 256, 1024)), dim3(256), 0, s, w.recon_part, n_parts, B,
                       w.recon);
    SUAVE_CHECK_HIP(hipGetLastError());
    recon_final = w.recon;
  }

  ReduceArgs red{};
  red.n_segs = 0;
  red.slab = w.slab;
  red.dst = grads;
  red.sums_slab = w.sums_slab;
  red.n_sum_slabs = lworkers * n_chunks + reparam_grid(B) + (n_wide ? wide_softmax_grid(B) : 0);
  red.sums_out = out ? out->sums : nullptr;
  red.n_rows = (double)B;
  int max_elems = 256;
  HeadReduceArgs head_red{};  // the encoder head's chunk slabs: summed by extra blocks of the final reduction launch
  bool have_head = false;

  if (grad) {
    if (n_wide)
      if (int rc = wide_backward(p, params, w, B, w.ds_part + (size_t)n_chunks * B * p->ldK, &slab_off, &red, &max_elems, s))
        return rc;
    // ---- decoder backward: per layer, dW and d(input) in one launch ----
    {
      const float* dcur = w.y;
      int lddcur = p->ldY;
      const suave_layer_t* lcur = &m.y_proj;
      for (int i = nh - 1; i >= -1; --i) {
        const float* ain = i >= 0 ? w.hd[i] : w.z;          // input activation of layer lcur
        const int ldain = i >= 0 ? p->ldH[i] : p->ldD;
        float* dain = i >= 0 ? w.dhd[i] : w.dz;
        if (int rc = bwd_layer(p, params, *lcur, dcur, lddcur, ain, ldain, dain, ldain, B, i >= 0 ? w.hd[i] : nullptr,
                               i >= 0 ? drop_p : 0.f, w.slab, &slab_off, &red, &max_elems, s, w.wT[layer_index(p, lcur)],
                               (i >= 0 && dec_masks) ? w.mask_d[i] : nullptr))
          return rc;
        dcur = dain;
        lddcur = ldain;
        if (i >= 0) lcur = &m.dec[i];
      }
    }
    // ---- prior gradients: per-component column sums of prior_terms, out[c][k] (k_prior_colsum) ----
    {
      const int Mt = prior_Mt, Nt = m.n_components;
      const int ldc = prior_ldc;
      const int splits = fuse_colsum ? reparam_grid(B) : prior_colsum_blocks(B);
      const size_t src_off = fuse_colsum ? prior_slab_off : slab_off;  // fused: k_reparam wrote the slabs already
      if (!fuse_colsum)
        if (int rc = launch_prior_colsum(w.prior_terms, ldT, w.kidx, B, Mt, Nt, ldc, w.slab + slab_off, s)) return rc;
      SUAVE_REQUIRE(red.n_segs + 4 <= kMaxReduceSegs, "too many reduction segments");
      ReduceSeg sm{};  // prior.mu[k][d] = out[d][k]
      sm.src_off = (int64_t)src_off; sm.slab_stride = (int64_t)Mt * ldc; sm.n_slabs = splits;
      sm.rows = D; sm.cols = K; sm.src_ld = ldc; sm.dst_off = m.prior_mu_off; sm.dst_si = 1; sm.dst_sj = D;
      red.segs[red.n_segs++] = sm;
      ReduceSeg sv = sm;  // prior.logvar[k][d] = out[D + d][k]
      sv.src_off += (int64_t)D * ldc; sv.dst_off = m.prior_logvar_off;
      red.segs[red.n_segs++] = sv;
      ReduceSeg sl = sm;  // prior.logits[j] = out[2D + j][K]  (ones column)
      sl.src_off += (int64_t)2 * D * ldc + K; sl.rows = K; sl.cols = 1; sl.dst_off = m.prior_logits_off;
      sl.dst_si = 1; sl.dst_sj = 1;
      red.segs[red.n_segs++] = sl;
      if (!fuse_colsum) slab_off += (size_t)splits * Mt * ldc;
    }
    // ---- likelihood-head weights ----
    {
      ReduceSeg sh{};
      sh.src_off = 0; sh.slab_stride = m.head_len; sh.n_slabs = lworkers / loss_workers_per_block(); sh.rows = 1; sh.cols = m.head_len;  // one row per block
      sh.src_ld = m.head_len; sh.dst_off = m.head_off; sh.dst_si = 0; sh.dst_sj = 1;
      red.segs[red.n_segs++] = sh;
      max_elems = std::max(max_elems, m.head_len);
    }
    // ---- the decoder-side bucket (decoder, likelihood heads, prior: one contiguous range of the flat gradient) is
    // complete here.  With an event to signal, reduce it now and record the event, so that a data-parallel caller
    // can all-reduce this bucket while the encoder backward below is still running (suave_elbo_backward_overlap).
    if (ready_event) {
      if (int rc = launch_reduce(red, max_elems, s)) return rc;
      SUAVE_CHECK_HIP(hipEventRecord(ready_event, s));
      red.n_segs = 0;
      red.sums_slab = nullptr; red.n_sum_slabs = 0; red.sums_out = nullptr;  // the epoch sums went with the first launch
      max_elems = 256;
    }
    // ---- latent backward ----
    if (head) {  // joint fine-tuning in one pass: CE(classifier(latent)) on the step's own z, its gradient into dz
      if (int rc = launch_head_loss(*head, D, w.z, p->ldD, 0, batch->row_index, batch->row_begin, B, w.dz, p->ldD, nullptr, s))
        return rc;
    }
    if (dlatent) {  // classification loss of joint fine-tuning reaches the encoder through outputs["latent"]
      hipLaunchKernelGGL(k_add_rows, dim3(std::min((B * D + 255) / 256, 2048)), dim3(256), 0, s, w.dz, p->ldD, dlatent, D, B, D);
      SUAVE_CHECK_HIP(hipGetLastError());
    }
    {
      LatentBwdArgs a{};
      a.sel = w.sel; a.ldS = ldS; a.K = K; a.D = D; a.ldK = p->ldK; a.ldD = p->ldD;
      a.n_rows = B; a.n_chunks = n_parts; a.ysoft = w.ysoft; a.kidx = w.kidx; a.dz = w.dz;
      a.ds_part = w.ds_part; a.gkl_logit = w.gkl_logit; a.tau = tau_c;
      a.dsel = w.dsel;
      a.params = params; a.prior_mu_off = m.prior_mu_off; a.prior_logvar_off = m.prior_logvar_off;
      a.bk = beta_scale / (float)global_rows; a.row_index = batch->row_index; a.row_begin = batch->row_begin;
      a.eps_in = nz->eps; a.seed = nz->seed; a.step = nz->step;
      if (int rc = launch_latent_bwd(a, s)) return rc;
    }
    // ---- encoder head backward (grouped by component) ----
    {
      const suave_layer_t& l = m.enc_head;
      const int ldc = round_up(l.n_in + 1, 4);
      GemmArgs g = gemm_base();  // per-chunk slabs of dsel^T * h (+ column sums = bias gradient)
      g.A = w.dsel; g.B = hlast_e; g.C = w.gslab;
      g.M = ldS; g.N = l.n_in; g.K = Vg; g.lda = ldS; g.ldb = ldhe_last; g.ldc = ldc;
      g.a_kmajor = 1; g.b_kmajor = 1; g.epi = kEpiSlab; g.splits = w.n_chunks_max;
      g.slab_stride = (int64_t)ldS * ldc; g.ones_col = l.n_in; g.grouped = 2; g.a_scale = t_grad_scale;
      head_geometry(g);
      GemmArgs d = gemm_base();  // d h = dsel * W[rows of the row's component], relu/dropout mask of h
      if (nh > 0) {
        d.A = w.dsel; d.B = params + m.enc_head.w_off; d.C = w.dhe[nh - 1];
        d.M = Vg; d.N = m.enc_head.n_in; d.K = ldS; d.lda = ldS; d.ldb = m.enc_head.ld; d.ldc = p->ldH[nh - 1];
        d.b_kmajor = 1; d.epi = kEpiMaskBwd; d.H = w.he[nh - 1]; d.drop_p = drop_p; d.grouped = 1; d.q0 = 0;
        d.a_scale = t_grad_scale;  // (read by the fp16-plane small-batch kernel; the exact-fp32 tile kernel ignores it)
        head_geometry(d);
        if (int rc = launch_gemm_pair(d, g, p->n_cu, s)) return rc;
      } else {
        if (int rc = launch_gemm(g, p->n_cu, s)) return rc;
      }
      HeadReduceArgs& h = head_red;  // (summed by extra blocks of the final reduction launch)
      have_head = true;
      h.slab = w.gslab; h.chunk_begin = w.chunk_begin; h.K = K; h.D = D; h.ldK = p->ldK; h.ldD = p->ldD; h.ldS = ldS;
      h.n_in = l.n_in; h.ldc = ldc; h.grads = grads; h.w_off = l.w_off; h.b_off = l.b_off; h.ld = l.ld;
    }
    // ---- encoder trunk backward ----
    for (int i = nh - 1; i >= 0; --i) {
      const float* ain = i > 0 ? w.he[i - 1] : w.x;
      const int ldain = i > 0 ? p->ldH[i - 1] : p->ldI;
      if (i > 0) {
        if (int rc = bwd_layer(p, params, m.enc[i], w.dhe[i], p->ldH[i], ain, ldain, w.dhe[i - 1], ldain, B, w.he[i - 1],
                               drop_p, w.slab, &slab_off, &red, &max_elems, s, w.wT[layer_index(p, &m.enc[i])],
                               enc_masks ? w.mask_e[i - 1] : nullptr))
          return rc;
      } else {
        if (int rc = bwd_weight(p, m.enc[i], w.dhe[i], p->ldH[i], ain, ldain, B, w.slab, &slab_off, &red, &max_elems, s))
          return rc;
        if (int rc = encoder_embed_backward(p, w, B, grads, s)) return rc;
      }
    }
    SUAVE_REQUIRE(slab_off <= w.slab_floats, "internal: slab overflow");
  }
  if ((grad && red.n_segs > 0) || red.sums_out || have_head) {
    if (int rc = launch_reduce(red, max_elems, s, have_head ? &head_red : nullptr)) return rc;
  }
  if (!grad && ready_event) SUAVE_CHECK_HIP(hipEventRecord(ready_event, s));
  // ---- per-row outputs ----
  if (out) {
    if (out->reconstruction)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->reconstruction, recon_final, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->categorical_kl)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->categorical_kl, w.catkl, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->gaussian_kl)
      SUAVE_CHECK_HIP(hipMemcpyAsync(out->gaussian_kl, w.gausskl, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    if (out->enc_out) {  // the full head is only evaluated when the caller asks for it
      if (int rc = fwd_layer(p, params, m.enc_head, hlast_e, ldhe_last, w.encout, p->ldE, B, false, 0.f, nullptr, batch, nz, 0, 1, s))
        return rc;
      if (int rc = launch_posterior_stats(w.encout, p->ldE, m.n_components, m.latent_dim, B, 1.0f, out->enc_out, nullptr,
                                          nullptr, nullptr, s))
        return rc;
    }
  }
  return 0;
}

}  // namespace

}  // namespace suave

using namespace suave;

extern "C" const char* suave_last_error(void) { return g_last_error.c_str(); }

extern "C" int suave_plan_create(const suave_model_t* model, const suave_col_t* cols, suave_plan** out) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(model && cols && out, "plan_create: null argument");
  const suave_model_t& m = *model;
  SUAVE_REQUIRE(m.n_cols > 0 && m.n_cols < 16384, "plan_create: 1..16383 columns");
  SUAVE_REQUIRE(m.n_components >= 1 && m.n_components <= SUAVE_MAX_COMPONENTS, "plan_create: 1..32 mixture components");
  SUAVE_REQUIRE(m.latent_dim >= 1, "plan_create: latent_dim must be positive");
  SUAVE_REQUIRE(m.n_hidden >= 0 && m.n_hidden <= SUAVE_MAX_HIDDEN, "plan_create: at most 8 hidden layers");
  SUAVE_REQUIRE(m.dropout >= 0.f && m.dropout < 1.f, "plan_create: dropout must lie in [0,1)");
  suave_plan* p = new suave_plan();
  p->model = m;
  p->cols_host.assign(cols, cols + m.n_cols);
  const int K = m.n_components, D = m.latent_dim;
  p->ldI = round_up(m.enc_in, 4);
  p->enc_out_width = K + 2 * K * D;
  p->ldE = round_up(p->enc_out_width, 4);
  p->ldK = round_up(K, 4);
  p->ldD = round_up(D, 4);
  p->ldY = round_up(m.y_rows, 4);
  for (int i = 0; i < m.n_hidden; ++i) p->ldH[i] = round_up(m.enc[i].n_out, 4);
  p->kmax = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
  auto fail = [&](const char* msg) {
    set_error(msg);
    suave_plan_destroy(p);
    return 1;
  };
  // layer table consistency: every ld is the padded input width, offsets 16-byte aligned
  {
    int prev = p->ldI, prev_n = m.enc_in;
    for (int i = 0; i < m.n_hidden; ++i) {
      if (m.enc[i].ld != prev || m.enc[i].n_in != prev_n || m.enc[i].w_off % 4) return fail("plan_create: encoder layer table inconsistent");
      if (m.dec[i].n_out != m.enc[i].n_out) return fail("plan_create: encoder/decoder hidden sizes differ");
      prev = p->ldH[i];
      prev_n = m.enc[i].n_out;
    }
    if (m.enc_head.ld != prev || m.enc_head.n_in != prev_n || m.enc_head.n_out != p->enc_out_width || m.enc_head.w_off % 4)
      return fail("plan_create: encoder head table inconsistent");
    prev = p->ldD;
    prev_n = D;
    for (int i = 0; i < m.n_hidden; ++i) {
      if (m.dec[i].ld != prev || m.dec[i].n_in != prev_n || m.dec[i].w_off % 4) return fail("plan_create: decoder layer table inconsistent");
      prev = p->ldH[i];
      prev_n = m.dec[i].n_out;
    }
    if (m.y_proj.ld != prev || m.y_proj.n_in != prev_n || m.y_proj.n_out != m.y_rows || m.y_proj.w_off % 4)
      return fail("plan_create: y_projection table inconsistent");
  }
  // column table: narrow columns have contiguous y slices and head weights in schema order; the wide categorical
  // columns follow behind them with 16-byte aligned y slots and weight matrices (include/suave_hip.h)
  int y = 0, hw = m.head_off, total_classes = 0, max_classes = 1, sum_y = 0;
  for (int c = 0; c < m.n_cols; ++c) {
    const suave_col_t& col = cols[c];
    if (col.type < 0 || col.type > SUAVE_TYPE_ORDINAL) return fail("plan_create: bad column type");
    if (col.y_dim < 1) return fail("plan_create: y_dim must be positive");
    const bool disc = col.type >= SUAVE_TYPE_CAT;
    if (disc && (col.n_classes < 2 || col.n_classes > 32767)) return fail("plan_create: cat/ordinal need 2..32767 classes");
    if (col.x_index < 0 || col.x_index >= (disc ? m.n_disc : m.n_num)) return fail("plan_create: x_index out of range");
    sum_y += col.y_dim;
    if (disc) total_classes += col.n_classes;
    if (col_is_wide(col)) continue;
    if (col.y_off != y) return fail("plan_create: y slices of the narrow columns must be contiguous in schema order");
    y += col.y_dim;
    if (col.w_off != hw) return fail("plan_create: head weights of the narrow columns must be contiguous in schema order");
    int len1 = 0, len2 = 0;
    switch (col.type) {
      case SUAVE_TYPE_REAL: case SUAVE_TYPE_POS: len1 = col.y_dim + K; len2 = K; break;
      case SUAVE_TYPE_COUNT: len1 = col.y_dim + K; break;
      case SUAVE_TYPE_CAT: len1 = (col.n_classes - 1) * (col.y_dim + K); break;
      case SUAVE_TYPE_ORDINAL: len1 = (col.n_classes - 1) * K; len2 = col.y_dim + K; break;
    }
    if (len2 ? col.w2_off != hw + len1 : col.w2_off != -1) return fail("plan_create: second head tensor misplaced");
    hw += len1 + len2;
    // per-class LDS scratch of the likelihood kernel: only its operand-at-a-time categorical path (y_dim > 16: a
    // Schema y_dimensions override) parks logits there; the register-blocked heads recompute them
    if (col.type == SUAVE_TYPE_CAT && col.y_dim > 16) max_classes = std::max(max_classes, col.n_classes);
    // (loss.hip: y_dim for a categorical head, max(y_dim, classes - 1) for an ordinal one, 9 .. 16)
    const int blk = col.type == SUAVE_TYPE_CAT ? col.y_dim : col.type == SUAVE_TYPE_ORDINAL ? std::max(col.y_dim, col.n_classes - 1) : 0;
    if (blk > 8 && blk <= 16) p->loss_wide16 = 1;
    ++p->n_narrow;
  }
  if (sum_y != m.total_y) return fail("plan_create: total_y mismatch");
  if (hw != m.head_off + m.head_len) return fail("plan_create: head_len mismatch");
  {
    int yw = round_up(y, 4), ww = round_up(hw, 4), l_off = 0;
    int cum[3] = {0, 0, 0};
    std::vector<int32_t> cum_host[3];
    p->wide_w0 = ww;
    for (int c = 0; c < m.n_cols; ++c) {
      const suave_col_t& col = cols[c];
      if (!col_is_wide(col)) continue;
      const int ldw = cat_ldw(col, K);
      if (col.y_off != yw) return fail("plan_create: wide categorical y slots must follow the narrow slices, 16-byte aligned");
      if (col.w_off != ww || col.w2_off != -1) return fail("plan_create: wide categorical weights must follow the narrow heads, 16-byte aligned");
      suave::WideCol wc{};
      wc.col = c; wc.C = col.n_classes; wc.yd = col.y_dim; wc.y_off = yw; wc.l_off = l_off; wc.w_off = ww; wc.ldw = ldw;
      wc.t_off = p->wide_t_len;
      p->wide_t_len += ldw * round_up(col.n_classes - 1, 4);
      p->wide_host.push_back(wc);
      for (int q = 0; q < 3; ++q) cum_host[q].push_back(cum[q]);
      cum[0] += (col.n_classes - 1 + 127) / 128;
      cum[1] += (ldw + 127) / 128;
      cum[2] += ((col.n_classes - 1 + 127) / 128) * ((ldw + 127) / 128);
      yw += ldw;
      ww += (col.n_classes - 1) * ldw;
      l_off += round_up(col.n_classes - 1, 4);
    }
    const bool any_wide = !p->wide_host.empty();
    if (m.y_rows != (any_wide ? yw : m.total_y)) return fail("plan_create: y_rows mismatch");
    if (m.y_proj.n_out != m.y_rows) return fail("plan_create: y_projection must have y_rows rows");
    p->ldL = std::max(l_off, 4);
    p->wide_len = ww - p->wide_w0;
    if (ww > m.n_params) return fail("plan_create: wide categorical weights exceed the parameter buffer");
    for (int q = 0; q < 3; ++q) {
      cum_host[q].push_back(cum[q]);
      p->wide_cum_total[q] = cum[q];
    }
    if (any_wide) {
      const size_t nw = p->wide_host.size();
      if (hipMalloc(&p->wide_dev, sizeof(suave::WideCol) * nw) != hipSuccess ||
          hipMemcpy(p->wide_dev, p->wide_host.data(), sizeof(suave::WideCol) * nw, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("plan_create: hipMalloc/hipMemcpy of the wide column table failed");
        suave_plan_destroy(p);
        return 2;
      }
      for (int q = 0; q < 3; ++q) {
        if (hipMalloc(&p->wide_cum_dev[q], sizeof(int32_t) * (nw + 1)) != hipSuccess ||
            hipMemcpy(p->wide_cum_dev[q], cum_host[q].data(), sizeof(int32_t) * (nw + 1), hipMemcpyHostToDevice) != hipSuccess) {
          set_error("plan_create: hipMalloc/hipMemcpy of the wide tile table failed");
          suave_plan_destroy(p);
          return 2;
        }
      }
    }
  }
  p->total_classes = total_classes;
  p->max_classes = max_classes;
  // chunking of the narrow columns for the likelihood kernel (LDS budget); a chunk never spans a wide column
  {
    // chunks of about equal y width, ~24 floats of y per tile row (swept together with the workers per CU: S64
    // 20/16 -> 24/28 takes 5 % off the kernel, W1024 2 %); its y tile fits the float4 slots of the register prefetch
    static const int kChunkY = [] { const char* e = getenv("SUAVE_LOSS_CHUNK_Y"); return e ? std::max(4, atoi(e)) : 24; }();  // tuning knob
    static const int kChunkYSmall = [] { const char* e = getenv("SUAVE_LOSS_CHUNK_Y_SMALL"); return e ? atoi(e) : 8; }();  // tuning knob (0: off)
    int hw_max = 0, max_y = 0, max_cc = 1;
    for (int pass = 0; pass < 2; ++pass) {
    const int chunk_y = pass == 0 ? kChunkY : kChunkYSmall;
    if (chunk_y <= 0 || (pass == 1 && chunk_y >= kChunkY)) continue;
    std::vector<ColChunk>& chunk_out = pass == 0 ? p->chunks_host : p->chunks_small_host;
    const int n_target = std::max(1, (y + chunk_y - 1) / chunk_y);
    const int y_target = (y + n_target - 1) / n_target;
    ColChunk cur{0, 0, 0, 0};
    auto w_end = [&](int c) {
      for (int n = c + 1; n < m.n_cols; ++n)
        if (!col_is_wide(cols[n])) return cols[n].w_off;
      return m.head_off + m.head_len;
    };
    auto close = [&]() {
      if (cur.col_end <= cur.col_begin) return;
      chunk_out.push_back(cur);
      hw_max = std::max(hw_max, w_end(cur.col_end - 1) - cols[cur.col_begin].w_off);
      max_y = std::max(max_y, cur.y_len);
      max_cc = std::max(max_cc, cur.col_end - cur.col_begin);
    };
    for (int c = 0; c < m.n_cols; ++c) {
      if (col_is_wide(cols[c])) {
        close();
        cur = ColChunk{c + 1, c + 1, 0, 0};
        continue;
      }
      if (cur.col_end > cur.col_begin && cur.y_len + cols[c].y_dim > y_target) {
        close();
        cur = ColChunk{c, c, cols[c].y_off, 0};
      }
      if (cur.col_end == cur.col_begin) { cur.col_begin = c; cur.y_begin = cols[c].y_off; }
      cur.col_end = c + 1;
      cur.y_len += cols[c].y_dim;
    }
    close();
    }
    if (p->chunks_small_host.size() <= p->chunks_host.size()) p->chunks_small_host.clear();
    p->max_chunk_y = std::max(max_y, 1);
    hw_max = round_up(std::max(hw_max, 4), 4);
    p->hw_max = hw_max;
    p->loss_ldx = max_cc | 1;
    p->loss_ldm = (round_up(max_cc, 4) / 4 | 1) * 4;
    p->loss_lds_bytes = loss_lds_bytes(p->max_chunk_y | 1, K, hw_max, max_classes, p->loss_ldx, p->loss_ldm);
    if (p->loss_lds_bytes > 150 * 1024)
      return fail("plan_create: a narrow column needs more than 150 KB of LDS in the likelihood kernel");
  }
  // device tables
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("plan_create: no HIP device");
  p->device = dev;
  p->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  std::vector<int32_t> enc_src(p->ldI, 0);  // kind 0 = pad
  for (int c = 0; c < m.n_cols; ++c) {
    const suave_col_t& col = cols[c];
    const int width = col.type >= SUAVE_TYPE_CAT ? col.n_classes : 1;
    if (col.enc_mask < 0 || col.enc_mask >= m.enc_in) return fail("plan_create: encoder mask offset out of range");
    enc_src[col.enc_mask] = (int32_t)((1u << 29) | (uint32_t)c);
    if (col.enc_embed >= 0) {  // embedded column: no entries in the dense input (include/suave_hip.h)
      const int ldh = m.n_hidden > 0 ? round_up(m.enc[0].n_out, 4) : 0;
      if (!col_is_wide(col) || m.n_hidden == 0 || col.enc_embed % 4 ||
          (int64_t)col.enc_embed + (int64_t)col.n_classes * ldh > m.n_params)
        return fail("plan_create: bad encoder embedding block (wide categorical columns, n_hidden > 0)");
      suave::EmbedCol ec{};
      ec.x_index = col.x_index; ec.C = col.n_classes; ec.e_off = col.enc_embed; ec.cls_begin = p->embed_classes_total;
      p->embed_host.push_back(ec);
      p->embed_classes_total += col.n_classes;
      continue;
    }
    if (col.enc_off < 0 || col.enc_off + width > m.enc_in) return fail("plan_create: encoder input offsets out of range");
    const uint32_t kind = col.type <= SUAVE_TYPE_COUNT ? 2u : (col.type == SUAVE_TYPE_CAT ? 3u : 4u);
    for (int k = 0; k < width; ++k)
      enc_src[col.enc_off + k] = (int32_t)((kind << 29) | ((uint32_t)k << 14) | (uint32_t)col.x_index);
  }
  auto dfail = [&](const char* msg) {
    set_error(msg);
    suave_plan_destroy(p);
    return 2;
  };
  if (hipMalloc(&p->cols_dev, sizeof(suave_col_t) * m.n_cols) != hipSuccess) return dfail("plan_create: hipMalloc failed");
  if (hipMalloc(&p->chunks_dev, sizeof(ColChunk) * std::max<size_t>(p->chunks_host.size() + p->chunks_small_host.size(), 1)) != hipSuccess) return dfail("plan_create: hipMalloc failed");
  if (hipMalloc(&p->enc_map_dev, sizeof(int32_t) * p->ldI) != hipSuccess) return dfail("plan_create: hipMalloc failed");
  if (hipMalloc(&p->cls_off_dev, sizeof(int32_t) * m.n_cols) != hipSuccess) return dfail("plan_create: hipMalloc failed");
  if (hipMalloc(&p->xsrc_dev, sizeof(int32_t) * m.n_cols) != hipSuccess) return dfail("plan_create: hipMalloc failed");
  {
    std::vector<int32_t> xsrc(m.n_cols);
    for (int c = 0; c < m.n_cols; ++c) xsrc[c] = cols[c].x_index | (cols[c].type >= SUAVE_TYPE_CAT ? (1 << 30) : 0);
    if (hipMemcpy(p->xsrc_dev, xsrc.data(), sizeof(int32_t) * m.n_cols, hipMemcpyHostToDevice) != hipSuccess)
      return dfail("plan_create: hipMemcpy failed");
  }
  {
    std::vector<int32_t> cls_off(m.n_cols, 0);
    int acc = 0;
    for (int c = 0; c < m.n_cols; ++c) {
      cls_off[c] = acc;
      if (cols[c].type >= SUAVE_TYPE_CAT) acc += cols[c].n_classes;
    }
    if (hipMemcpy(p->cls_off_dev, cls_off.data(), sizeof(int32_t) * m.n_cols, hipMemcpyHostToDevice) != hipSuccess)
      return dfail("plan_create: hipMemcpy failed");
  }
  if (decode_plan_init(p) != 0) { suave_plan_destroy(p); return 2; }  // (error text set)
  if (hipMemcpy(p->cols_dev, cols, sizeof(suave_col_t) * m.n_cols, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->chunks_dev, p->chunks_host.data(), sizeof(ColChunk) * p->chunks_host.size(), hipMemcpyHostToDevice) != hipSuccess ||
      (!p->chunks_small_host.empty() &&
       hipMemcpy(p->chunks_dev + p->chunks_host.size(), p->chunks_small_host.data(), sizeof(ColChunk) * p->chunks_small_host.size(),
                 hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(p->enc_map_dev, enc_src.data(), sizeof(int32_t) * p->ldI, hipMemcpyHostToDevice) != hipSuccess)
    return dfail("plan_create: hipMemcpy failed");
  if (p->loss_lds_bytes > 48 * 1024)
    if (loss_set_lds_limit(p->loss_lds_bytes)) { suave_plan_destroy(p); return 2; }
  if (!p->embed_host.empty()) {
    const size_t ne = p->embed_host.size();
    if (hipMalloc(&p->embed_dev, sizeof(suave::EmbedCol) * ne) != hipSuccess ||
        hipMemcpy(p->embed_dev, p->embed_host.data(), sizeof(suave::EmbedCol) * ne, hipMemcpyHostToDevice) != hipSuccess)
      return dfail("plan_create: hipMalloc/hipMemcpy of the embedding table failed");
    if (round_up(m.enc[0].n_out, 4) > 1024) return fail("plan_create: first hidden layer wider than 1024 with embedded columns");
  }
  resolve_static_scratch(p);  // (a failure leaves static_scratch null: only suave_decode_sample_static needs it, and says so)
  *out = p;
  return 0;
}

extern "C" int suave_plan_destroy(suave_plan* p) {
  if (!p) return 0;
  if (p->cols_dev) (void)hipFree(p->cols_dev);
  if (p->chunks_dev) (void)hipFree(p->chunks_dev);
  if (p->enc_map_dev) (void)hipFree(p->enc_map_dev);
  if (p->cls_off_dev) (void)hipFree(p->cls_off_dev);
  decode_plan_free(p);
  if (p->xsrc_dev) (void)hipFree(p->xsrc_dev);
  if (p->wide_dev) (void)hipFree(p->wide_dev);
  if (p->embed_dev) (void)hipFree(p->embed_dev);
  for (int q = 0; q < 3; ++q)
    if (p->wide_cum_dev[q]) (void)hipFree(p->wide_cum_dev[q]);
  delete p;
  return 0;
}

extern "C" int suave_workspace_bytes(const suave_plan* plan, int64_t max_rows, size_t* bytes) {
  SUAVE_REQUIRE(plan && bytes, "workspace_bytes: null argument");
  SUAVE_REQUIRE(max_rows >= 0 && max_rows < (int64_t(1) << 31), "workspace_bytes: row count out of range");
  Workspace w;
  carve(plan, max_rows, nullptr, &w);
  *bytes = w.total_bytes;
  return 0;
}

extern "C" int suave_elbo_forward(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                                  const suave_noise_t* noise, float beta_scale, float tau, int training,
                                  void* workspace, size_t workspace_bytes, void* stream,
                                  const suave_elbo_out_t* out) {
  suave::drop_stale_hip_error();
  return elbo_impl(plan, params, batch, noise, beta_scale, tau, training, batch ? std::max(batch->n_rows, 1) : 1,
                   workspace, workspace_bytes, static_cast<hipStream_t>(stream), out, nullptr, nullptr);
}

extern "C" int suave_elbo_backward(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                                   const suave_noise_t* noise, float beta_scale, float tau, int64_t global_rows,
                                   void* workspace, size_t workspace_bytes, void* stream,
                                   const suave_elbo_out_t* out, float* grads, const float* dlatent) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(grads != nullptr, "elbo_backward: grads is null");
  return elbo_impl(plan, params, batch, noise, beta_scale, tau, 1, global_rows, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream), out, grads, dlatent);
}

extern "C" int suave_elbo_backward_overlap(const suave_plan* plan, const float* params, const suave_batch_t* batch,
                                           const suave_noise_t* noise, float beta_scale, float tau, int64_t global_rows,
                                           void* workspace, size_t workspace_bytes, void* stream,
                                           const suave_elbo_out_t* out, float* grads, const float* dlatent,
                                           void* decoder_ready_event) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(grads != nullptr, "elbo_backward_overlap: grads is null");
  SUAVE_REQUIRE(decoder_ready_event != nullptr, "elbo_backward_overlap: the event is null (use suave_elbo_backward)");
  return elbo_impl(plan, params, batch, noise, beta_scale, tau, 1, global_rows, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream), out, grads, dlatent, static_cast<hipEvent_t>(decoder_ready_event));
}

namespace {
int encode_impl(const suave_plan* p, const float* params, const suave_batch_t* batch, float inv_temperature, void* workspace,
                size_t workspace_bytes, void* stream, float* enc_out, float* post_mean, float* post_logvar, float* probs,
                int32_t* hard_comp, float* hard_mu) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p != nullptr && params != nullptr, "plan/params is null");
  if (int rc = check_batch(p, batch)) return rc;
  const suave_model_t& m = p->model;
  const int B = batch->n_rows;
  if (B == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Workspace w;
  carve(p, B, workspace, &w);
  SUAVE_REQUIRE(workspace_bytes >= w.total_bytes, "workspace too small (see suave_workspace_bytes)");
  if (int rc = launch_expand(p, *batch, w.x, nullptr, s)) return rc;
  const float* act = w.x;
  int ld_act = p->ldI;
  if (int rc = encoder_trunk_forward(p, params, batch, nullptr, 0.f, w, false, &act, &ld_act, s)) return rc;
  if (int rc = fwd_layer(p, params, m.enc_head, act, ld_act, w.encout, p->ldE, B, false, 0.f, nullptr, batch, nullptr, 0, 1, s))
    return rc;
  return launch_posterior_stats(w.encout, p->ldE, m.n_components, m.latent_dim, B, inv_temperature, enc_out, post_mean,
                                post_logvar, probs, s, hard_comp, hard_mu);
}
}  // namespace

extern "C" int suave_head_workspace_bytes(int32_t n_classes, int32_t latent_dim, size_t* bytes) {
  SUAVE_REQUIRE(bytes != nullptr, "head_workspace_bytes: null argument");
  SUAVE_REQUIRE(suave::head_shape_ok(n_classes, latent_dim), "classifier head: n_classes must be 2..64 and latent_dim <= 256 (C * D within LDS)");
  *bytes = sizeof(float) * suave::head_workspace_floats(n_classes, latent_dim);
  return 0;
}

extern "C" int suave_head_step(const suave_head_t* head, int32_t latent_dim, const float* latent, int32_t ld,
                               const int32_t* row_index, int64_t row_begin, int64_t n, void* stream, float* logits_out) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(head != nullptr && n >= 0, "head_step: null head");
  SUAVE_REQUIRE(n == 0 || latent != nullptr, "head_step: latent is null");
  SUAVE_REQUIRE(ld >= latent_dim, "head_step: ld < latent_dim");
  return suave::launch_head_loss(*head, latent_dim, latent, ld, 1, row_index, row_begin, n, nullptr, 0, logits_out,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int suave_elbo_joint_backward(const suave_plan* p, const float* params, const suave_batch_t* batch,
                                         const suave_noise_t* noise, float beta_scale, float tau, int64_t global_rows,
                                         void* workspace, size_t workspace_bytes, void* stream, const suave_elbo_out_t* out,
                                         float* grads, const suave_head_t* head, void* decoder_ready_event) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(grads != nullptr && head != nullptr, "elbo_joint_backward: grads / head is null");
  return elbo_impl(p, params, batch, noise, beta_scale, tau, 1, global_rows, workspace, workspace_bytes,
                   static_cast<hipStream_t>(stream), out, grads, nullptr, static_cast<hipEvent_t>(decoder_ready_event), head);
}

extern "C" int suave_encode(const suave_plan* p, const float* params, const suave_batch_t* batch,
                            float inv_temperature, void* workspace, size_t workspace_bytes, void* stream,
                            float* enc_out, float* post_mean, float* post_logvar, float* probs) {
  return encode_impl(p, params, batch, inv_temperature, workspace, workspace_bytes, stream, enc_out, post_mean, post_logvar,
                     probs, nullptr, nullptr);
}

extern "C" int suave_encode_hard(const suave_plan* p, const float* params, const suave_batch_t* batch,
                                 float inv_temperature, void* workspace, size_t workspace_bytes, void* stream,
                                 int32_t* component, float* z) {
  return encode_impl(p, params, batch, inv_temperature, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr,
                     component, z);
}

namespace {
// the library's word -> variate maps (common.h) on caller-given words: normal[i] pairs words 2*(i/2), 2*(i/2)+1
__global__ void k_rng_draws(const uint32_t* words, int64_t n, float* uniform, float* exponential, float* normal) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (uniform) uniform[i] = suave::u01(words[i]);
  if (exponential) exponential[i] = suave::exp_draw(words[i]);
  if (normal) {
    const int64_t j = i & ~int64_t(1);
    const float2 z = suave::normal2(words[j], words[j + 1 < n ? j + 1 : j]);
    normal[i] = (i & 1) ? z.y : z.x;
  }
}
}  // namespace

extern "C" int suave_rng_draws(const uint32_t* words, int64_t n, void* stream, float* uniform, float* exponential,
                               float* normal) {
  SUAVE_REQUIRE(words != nullptr && n >= 0, "words is null");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_rng_draws, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), words, n,
                     uniform, exponential, normal);
  SUAVE_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace {

// Scratch of the decode path: z with padded rows, the decoder's hidden activations, y_full.
struct DecodeScratch {
  float* z;
  float* hd[SUAVE_MAX_HIDDEN];
  float* y;
  float* wideL;              // [rows, ldL] logits of the wide categorical columns (null: the schema has none)
  unsigned char* fused_img;  // weight images of the fused decode kernel (decode.hip); null: the model is outside its shapes
};

// A decode call takes decode.hip's one-launch kernel when the rows come with component indices (hard assignments:
// sample, impute's default strategy, the Monte-Carlo draws) and the model is inside the kernel's shapes; dense (soft)
// assignments, wide categorical columns, other depths / widths and arithmetic modes 0 / 3 keep the per-layer path.
bool decode_takes_fused(const suave_plan* p, const float* s_dense) {
  static const bool off = getenv("SUAVE_NO_FUSED_DECODE") != nullptr;  // measurement knob: the per-layer path
  return !off && s_dense == nullptr && decode_fused_ok(p);
}

// fused: the call takes decode.hip's one-launch path (decode_takes_fused) and needs the weight images only
size_t carve_decode(const suave_plan* p, int64_t rows, void* base, DecodeScratch* d, bool fused) {
  Carver c{static_cast<char*>(base)};
  const size_t R = (size_t)std::max<int64_t>(rows, 1);
  d->fused_img = nullptr;
  if (fused) {  // nothing but the weight images: activations and y_full stay on the chip
    d->z = nullptr; d->y = nullptr; d->wideL = nullptr;
    for (int i = 0; i < p->model.n_hidden; ++i) d->hd[i] = nullptr;
    d->fused_img = c.take<unsigned char>(decode_fused_image_bytes(p));
    return (c.off + 255) & ~size_t(255);
  }
  d->z = c.take<float>(R * p->ldD);
  for (int i = 0; i < p->model.n_hidden; ++i) d->hd[i] = c.take<float>(R * p->ldH[i]);
  d->y = c.take<float>(R * p->ldY);
  d->wideL = p->wide_host.empty() ? nullptr : c.take<float>(R * p->ldL);
  return (c.off + 255) & ~size_t(255);
}

// Decoder trunk + y_projection + per-column sample draw on caller-validated arguments.  None of the kernels
// below needs its scratch to start at any value: pad columns are masked when the tiles are staged.
// eval-mode decoder trunk + y_projection on w.z -> w.y: no dropout (model.py:3613-3617, 4397-4419)
int decoder_trunk_eval(const suave_plan* p, const float* params, const DecodeScratch& w, int B, hipStream_t st) {
  const suave_model_t& m = p->model;
  suave_batch_t dummy;
  std::memset(&dummy, 0, sizeof(dummy));
  const float* act = w.z;
  int ld_act = p->ldD;
  for (int i = 0; i < m.n_hidden; ++i) {
    if (int rc = fwd_layer(p, params, m.dec[i], act, ld_act, w.hd[i], p->ldH[i], B, true, 0.f, nullptr, &dummy, nullptr, 0, 1, st))
      return rc;
    act = w.hd[i];
    ld_act = p->ldH[i];
  }
  return fwd_layer(p, params, m.y_proj, act, ld_act, w.y, p->ldY, B, false, 0.f, nullptr, &dummy, nullptr, 0, 1, st);
}

int decode_sample_impl(const suave_plan* p, const float* params, const float* z, const float* s_dense,
                       const int32_t* component, int B, const float* noise_normal, const float* noise_uniform,
                       const float* noise_exp, uint64_t seed, int64_t row_offset, const DecodeScratch& w, hipStream_t st,
                       float* out_num, int32_t ld_num, int16_t* out_codes, int32_t ld_disc, float* y_full_out) {
  const suave_model_t& m = p->model;
  if (w.fused_img != nullptr)  // decoder trunk -> y_projection -> sample draw in one launch (decode.hip)
    return launch_decode_fused(p, params, z, component, B, noise_normal, noise_uniform, noise_exp, seed, row_offset,
                               w.fused_img, out_num, ld_num, out_codes, ld_disc, y_full_out, st);
  SUAVE_CHECK_HIP(hipMemcpy2DAsync(w.z, sizeof(float) * p->ldD, z, sizeof(float) * m.latent_dim,
                                   sizeof(float) * m.latent_dim, B, hipMemcpyDeviceToDevice, st));
  if (int rc = decoder_trunk_eval(p, params, w, B, st)) return rc;
  if (y_full_out)
    SUAVE_CHECK_HIP(hipMemcpy2DAsync(y_full_out, sizeof(float) * m.y_rows, w.y, sizeof(float) * p->ldY,
                                     sizeof(float) * m.y_rows, B, hipMemcpyDeviceToDevice, st));
  if (w.wideL != nullptr) {  // wide categorical heads: assignment slots + logits GEMM, as in the training forward
    WideRowArgs a{};
    a.wide = p->wide_dev; a.n_wide = (int)p->wide_host.size(); a.n_rows = B; a.vdiv = 1; a.K = m.n_components; a.ldK = p->ldK;
    a.y = w.y; a.ldY = p->ldY; a.L = w.wideL; a.ldL = p->ldL; a.kidx = s_dense ? nullptr : component; a.s_dense = s_dense;
    if (int rc = launch_wide_fill(a, st)) return rc;
    WideGemmArgs g{};
    g.cols = p->wide_dev; g.cum = p->wide_cum_dev[0]; g.n_wide = a.n_wide; g.cum_total = p->wide_cum_total[0];
    g.params = params; g.y = w.y; g.ldY = p->ldY; g.L = w.wideL; g.ldL = p->ldL; g.rows = B; g.K = m.n_components;
    g.splits = 1; g.slab_stride = p->wide_len; g.wide_w0 = p->wide_w0;
    if (int rc = launch_gemm_wide(g, 0, st)) return rc;
  }
  return launch_sample_heads(p, p->cls_off_dev, params, w.y, s_dense, component, B, noise_normal, noise_uniform,
                             noise_exp, seed, row_offset, out_num, ld_num, out_codes, ld_disc, st, w.wideL);
}

int check_decode_args(const suave_plan* p, const float* params, const float* z, const float* s_dense,
                      const int32_t* component, int64_t n, const float* out_num, int32_t ld_num, const int16_t* out_codes,
                      int32_t ld_disc) {
  SUAVE_REQUIRE(p && params, "decode_sample: null plan/params");
  const suave_model_t& m = p->model;
  SUAVE_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "decode_sample: row count out of range");
  if (n == 0) return 0;  // empty request: nothing to read or write, buffers may be null
  SUAVE_REQUIRE(z != nullptr, "decode_sample: z is null");
  SUAVE_REQUIRE(s_dense != nullptr || component != nullptr, "decode_sample: need dense assignments or component indices");
  SUAVE_REQUIRE((m.n_num == 0 || (out_num && ld_num >= m.n_num)) && (m.n_disc == 0 || (out_codes && ld_disc >= m.n_disc)),
                "decode_sample: output buffers missing or too narrow");
  return 0;
}

}  // namespace

extern "C" int suave_decode_sample(const suave_plan* p, const float* params, const float* z, const float* s_dense,
                                   const int32_t* component, int64_t n, const float* noise_normal,
                                   const float* noise_uniform, const float* noise_exp, uint64_t seed,
                                   int64_t row_offset, void* workspace, size_t workspace_bytes, void* stream,
                                   float* out_num, int32_t ld_num, int16_t* out_codes, int32_t ld_disc,
                                   float* y_full_out) {
  suave::drop_stale_hip_error();
  if (int rc = check_decode_args(p, params, z, s_dense, component, n, out_num, ld_num, out_codes, ld_disc)) return rc;
  if (n == 0) return 0;
  // the workspace is sized by suave_workspace_bytes (the training step's carving); the decode uses its first part
  Workspace w;
  carve(p, n, nullptr, &w);
  SUAVE_REQUIRE(workspace_bytes >= w.total_bytes, "workspace too small (see suave_workspace_bytes)");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  DecodeScratch d;
  SUAVE_REQUIRE(carve_decode(p, n, workspace, &d, decode_takes_fused(p, s_dense)) <= workspace_bytes,
                "workspace too small (see suave_workspace_bytes)");
  return decode_sample_impl(p, params, z, s_dense, component, (int)n, noise_normal, noise_uniform, noise_exp, seed,
                            row_offset, d, static_cast<hipStream_t>(stream), out_num, ld_num, out_codes, ld_disc, y_full_out);
}

// ---- the entry point that allocates nothing (include/suave_hip.h) -------------------------------------------
// The decode's scratch lives in this file-scope device array: one copy per GPU the library runs on, held until
// the process ends.  Host code never uses the array's name as an address: suave_plan_create resolves the
// device's copy with hipGetSymbolAddress (plan->static_scratch) and every launch takes that pointer.
static __device__ __align__(256) unsigned char scratch[SUAVE_STATIC_SCRATCH_BYTES];

namespace suave {
int resolve_static_scratch(suave_plan* p) {
  void* ptr = nullptr;
  p->static_scratch = nullptr;
  if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess) {
    (void)hipGetLastError();  // the plan works without it; suave_decode_sample_static reports the null address
    return 1;
  }
  if (ptr == nullptr || (reinterpret_cast<uintptr_t>(ptr) & 255) != 0) return 1;
  p->static_scratch = ptr;
  return 0;
}
}  // namespace suave

extern "C" int suave_decode_static_max_rows(const suave_plan* p, int32_t dense_assignments, int64_t* max_rows) {
  SUAVE_REQUIRE(p != nullptr && max_rows != nullptr, "decode_static_max_rows: null argument");
  DecodeScratch d;
  const float* dense = dense_assignments ? reinterpret_cast<const float*>(p) : nullptr;  // (only tested against null)
  const bool fused = decode_takes_fused(p, dense);
  const int64_t cap = (int64_t(1) << 31) - 1;  // (n is an int inside the kernels)
  if (carve_decode(p, 1, nullptr, &d, fused) > (size_t)SUAVE_STATIC_SCRATCH_BYTES) { *max_rows = 0; return 0; }
  int64_t lo = 1, hi = cap;  // carve_decode is non-decreasing in the row count
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo + 1) / 2;
    if (carve_decode(p, mid, nullptr, &d, fused) <= (size_t)SUAVE_STATIC_SCRATCH_BYTES) lo = mid;
    else hi = mid - 1;
  }
  *max_rows = lo;
  return 0;
}

extern "C" int suave_decode_static_max_rows(const suave_plan* p, int32_t dense_assignments, int64_t* max_rows);

extern "C" int suave_decode_sample_static(const suave_plan* p, const float* params, const float* z, const float* s_dense,
                                          const int32_t* component, int64_t n, const float* noise_normal,
                                          const float* noise_uniform, const float* noise_exp, uint64_t seed,
                                          int64_t row_offset, void* stream, float* out_num, int32_t ld_num,
                                          int16_t* out_codes, int32_t ld_disc, float* y_full_out) {
  suave::drop_stale_hip_error();
  if (int rc = check_decode_args(p, params, z, s_dense, component, n, out_num, ld_num, out_codes, ld_disc)) return rc;
  if (n == 0) return 0;
  SUAVE_REQUIRE(p->static_scratch != nullptr, "decode_sample_static: the plan holds no static scratch address");
  int dev = -1;
  SUAVE_CHECK_HIP(hipGetDevice(&dev));
  SUAVE_REQUIRE(dev == p->device, "decode_sample_static: the plan belongs to another device (each GPU has its own scratch)");
  DecodeScratch d;
  const bool fused = decode_takes_fused(p, s_dense);
  const size_t need = carve_decode(p, n, p->static_scratch, &d, fused);
  if (need > (size_t)SUAVE_STATIC_SCRATCH_BYTES) {
    int64_t fit = 0;
    suave_decode_static_max_rows(p, s_dense != nullptr, &fit);
    set_error("decode_sample_static: " + std::to_string((long long)n) + " rows need " + std::to_string(need) +
              " bytes of scratch, the static buffer holds " + std::to_string((size_t)SUAVE_STATIC_SCRATCH_BYTES) +
              " (" + std::to_string((long long)fit) + " rows of this model and this kind of call: suave_decode_static_max_rows); "
              "split the request");
    return 1;
  }
  return decode_sample_impl(p, params, z, s_dense, component, (int)n, noise_normal, noise_uniform, noise_exp, seed,
                            row_offset, d, static_cast<hipStream_t>(stream), out_num, ld_num, out_codes, ld_disc, y_full_out);
}

extern "C" int suave_decode_log_px(const suave_plan* p, const float* params, const float* z, const int32_t* component,
                                   const suave_batch_t* batch, void* workspace, size_t workspace_bytes, void* stream,
                                   float* log_px, float* log_px_missing) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p && params, "decode_log_px: null plan/params");
  if (int rc = check_batch(p, batch)) return rc;
  const suave_model_t& m = p->model;
  const int B = batch->n_rows;
  if (B == 0) return 0;
  SUAVE_REQUIRE(z && component, "decode_log_px: z / component is null");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0, "params must be 16-byte aligned");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace w;
  carve(p, B, workspace, &w);
  SUAVE_REQUIRE(workspace_bytes >= w.total_bytes, "workspace too small (see suave_workspace_bytes)");
  w.log_px = log_px; w.log_px_missing = log_px_missing;
  // y_full = trunk(z), eval mode (per-layer path: the training carving has the hidden buffers)
  SUAVE_CHECK_HIP(hipMemcpy2DAsync(w.z, sizeof(float) * p->ldD, z, sizeof(float) * m.latent_dim,
                                   sizeof(float) * m.latent_dim, B, hipMemcpyDeviceToDevice, st));
  DecodeScratch d{};
  d.z = w.z; d.y = w.y; d.wideL = nullptr; d.fused_img = nullptr;
  for (int i = 0; i < m.n_hidden; ++i) d.hd[i] = w.hd[i];
  if (int rc = decoder_trunk_eval(p, params, d, B, st)) return rc;
  if (int rc = launch_expand(p, *batch, w.x, w.cells, st)) return rc;  // the packed cells of the batch rows
  const int n_chunks = (int)p->chunks_host.size();
  const int lgrid = loss_grid(p, B), lworkers = loss_workers(p, B);
  LossArgs a{};
  a.cols = p->cols_dev; a.chunks = p->chunks_dev; a.params = params; a.y = w.y; a.batch = *batch;
  a.n_rows = B; a.vdiv = 1; a.kidx = component; a.c_up = 0.f; a.bk = 0.f;
  a.K = m.n_components; a.D = m.latent_dim; a.ldY = p->ldY; a.ldy = p->max_chunk_y | 1; a.ldK = p->ldK;
  a.ldD = p->ldD; a.ldE = p->ldE; a.max_classes = p->max_classes;
  a.recon_part = w.recon_part; a.sums_slab = w.sums_slab;
  a.head_off = m.head_off; a.head_len = m.head_len; a.hw_max = p->hw_max;
  a.xsrc = p->xsrc_dev; a.ldx = p->loss_ldx; a.ldm = p->loss_ldm; a.cells = w.cells; a.n_cols = m.n_cols;
  a.log_px = log_px; a.log_px_missing = log_px_missing;
  if (n_chunks > 0)
    if (int rc = launch_loss(a, false, p->kmax, p->loss_wide16 != 0, lgrid, n_chunks, p->loss_lds_bytes, st)) return rc;
  if (!p->wide_host.empty())
    return wide_forward(p, params, w, B, 1, component, nullptr, nullptr, 0.f, false, w.recon_part + (size_t)n_chunks * B,
                        w.sums_slab + (size_t)lworkers * n_chunks * 4, st);
  return 0;
}

extern "C" int suave_pack_columns(const suave_plan* p, const suave_pack_col_t* cols, int64_t n, float* xnum, int32_t ld_num,
                                  int16_t* codes, int32_t ld_disc, uint8_t* mask, int32_t ld_mask, uint64_t* rejected,
                                  void* stream) {
  drop_stale_hip_error();
  SUAVE_REQUIRE(p != nullptr && cols != nullptr && rejected != nullptr, "pack_columns: plan / cols / rejected is null");
  SUAVE_REQUIRE(n >= 0, "pack_columns: negative row count");
  const suave_model_t& m = p->model;
  SUAVE_REQUIRE(mask != nullptr && (m.n_num == 0 || xnum != nullptr) && (m.n_disc == 0 || codes != nullptr),
                "pack_columns: output buffer is null");
  SUAVE_REQUIRE(ld_num >= m.n_num && ld_disc >= m.n_disc && ld_mask >= m.n_cols,
                "pack_columns: leading dimensions smaller than the schema");
  return launch_pack_columns(cols, m.n_cols, m.n_num, m.n_disc, n, xnum, ld_num, codes, ld_disc, mask, ld_mask,
                             reinterpret_cast<unsigned long long*>(rejected), static_cast<hipStream_t>(stream));
}

extern "C" int suave_unpack_columns(const suave_plan* p, const suave_unpack_col_t* cols, int32_t n_out, int64_t n,
                                    const float* out_num, int32_t ld_num, const int16_t* out_codes, int32_t ld_disc,
                                    float* f32_block, int64_t* i64_block, int16_t* i16_block, int8_t* i8_block,
                                    int64_t ld_rows, void* stream) {
  drop_stale_hip_error();
  SUAVE_REQUIRE(p != nullptr && (cols != nullptr || n_out == 0), "unpack_columns: plan / cols is null");
  SUAVE_REQUIRE(n >= 0 && n_out >= 0 && ld_rows >= n, "unpack_columns: bad sizes");
  SUAVE_REQUIRE(ld_num >= p->model.n_num && ld_disc >= p->model.n_disc, "unpack_columns: leading dimensions smaller than the schema");
  return launch_unpack_columns(cols, n_out, n, out_num, ld_num, out_codes, ld_disc, f32_block, i64_block, i16_block, i8_block,
                               ld_rows, static_cast<hipStream_t>(stream));
}

extern "C" int suave_gemm_arithmetic_set(int mode) { return gemm_split_mode_set(mode); }

extern "C" int suave_small_batch_rows_set(int rows) { return gemm_small_rows_limit_set(rows); }

extern "C" int suave_small_batch_rows_get(int* rows) {
  SUAVE_REQUIRE(rows != nullptr, "small_batch_rows_get: null argument");
  *rows = (int)gemm_small_rows_limit();
  return 0;
}

extern "C" int suave_gemm_arithmetic_get(int* mode) {
  SUAVE_REQUIRE(mode != nullptr, "gemm_arithmetic_get: null argument");
  *mode = gemm_split_mode();
  return 0;
}

extern "C" int suave_profile_enable(int on) {
  g_prof.enabled = on != 0;
  return 0;
}

extern "C" int suave_profile_read(double* ms_by_class, int64_t* launches_by_class, int32_t n_classes) {
  SUAVE_REQUIRE(ms_by_class && launches_by_class && n_classes >= kProfCount, "profile_read: need room for all classes");
  for (int c = 0; c < kProfCount; ++c) {
    double total = 0.0;
    const size_t n = std::min(g_prof.start[c].size(), g_prof.stop[c].size());
    for (size_t i = 0; i < n; ++i) {
      SUAVE_CHECK_HIP(hipEventSynchronize(g_prof.stop[c][i]));
      float ms = 0.f;
      SUAVE_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.start[c][i], g_prof.stop[c][i]));
      total += ms;
    }
    ms_by_class[c] = total;
    launches_by_class[c] = (int64_t)n;
    for (hipEvent_t e : g_prof.start[c]) g_prof.pool.push_back(e);
    for (hipEvent_t e : g_prof.stop[c]) g_prof.pool.push_back(e);
    g_prof.start[c].clear();
    g_prof.stop[c].clear();
  }
  return 0;
}

extern "C" int suave_head_params_width(const suave_plan* p, int32_t col_index, int32_t* width) {
  SUAVE_REQUIRE(p && width, "head_params_width: null argument");
  SUAVE_REQUIRE(col_index >= 0 && col_index < p->model.n_cols, "head_params_width: column out of range");
  const suave_col_t& c = p->cols_host[col_index];
  *width = c.type == SUAVE_TYPE_REAL ? 2 : c.type == SUAVE_TYPE_POS ? 3 : c.type == SUAVE_TYPE_COUNT ? 1 : c.n_classes;
  return 0;
}

extern "C" int suave_decode_params(const suave_plan* p, const float* params, const float* z, const float* s_dense,
                                   const int32_t* component, int64_t n, int32_t col_index, void* workspace,
                                   size_t workspace_bytes, void* stream, float* out_params) {
  suave::drop_stale_hip_error();
  SUAVE_REQUIRE(p && params, "decode_params: null plan/params");
  SUAVE_REQUIRE(col_index >= 0 && col_index < p->model.n_cols, "decode_params: column out of range");
  SUAVE_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "decode_params: row count out of range");
  if (n == 0) return 0;  // empty request: buffers may be null
  SUAVE_REQUIRE(z && out_params, "decode_params: null argument");
  SUAVE_REQUIRE(s_dense != nullptr || component != nullptr, "decode_params: need dense assignments or component indices");
  const suave_model_t& m = p->model;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Workspace wfull;
  carve(p, n, nullptr, &wfull);
  SUAVE_REQUIRE(workspace_bytes >= wfull.total_bytes, "workspace too small (see suave_workspace_bytes)");
  SUAVE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "workspace must be 256-byte aligned");
  DecodeScratch w;
  SUAVE_REQUIRE(carve_decode(p, n, workspace, &w, false) <= workspace_bytes, "workspace too small (see suave_workspace_bytes)");
  const int B = (int)n;
  SUAVE_CHECK_HIP(hipMemcpy2DAsync(w.z, sizeof(float) * p->ldD, z, sizeof(float) * m.latent_dim,
                                   sizeof(float) * m.latent_dim, B, hipMemcpyDeviceToDevice, st));
  if (int rc = decoder_trunk_eval(p, params, w, B, st)) return rc;
  int32_t width = 0;
  if (int rc = suave_head_params_width(p, col_index, &width)) return rc;
  return launch_head_params(p, col_index, params, w.y, s_dense, component, n, out_params, width, st);
}
