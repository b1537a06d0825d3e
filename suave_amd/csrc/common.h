// Internal declarations shared by the translation units of libsuave_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/suave_hip.h"

namespace suave {

void set_error(const std::string& msg);

// hipGetLastError() is sticky per thread: an error left behind by the caller's own earlier HIP calls (seen with a
// freshly initialised torch: hipErrorNoDevice from a probe) would otherwise fail this library's first launch check.
// Every compute entry point drops it on entry.
inline void drop_stale_hip_error() { (void)hipGetLastError(); }

#define SUAVE_CHECK_HIP(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::suave::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));               \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)

#define SUAVE_REQUIRE(cond, msg)                                                           \
  do {                                                                                     \
    if (!(cond)) {                                                                         \
      ::suave::set_error(std::string(msg) + " [" #cond "]");                               \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

constexpr float kEps = 1e-6f;            // suave/modules/distributions.py:18
constexpr float kLog2Pi = 1.8378770664f; // distributions.py:19
constexpr float kLogvarMin = -15.0f;     // suave/modules/encoder.py:31
constexpr float kLogvarMax = 15.0f;

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Wide categorical columns (SUAVE_WIDE_CAT_MIN_CLASSES, include/suave_hip.h) run their logits_layer as GEMMs.
__host__ __device__ inline bool col_is_wide(const suave_col_t& c) {
  return c.type == SUAVE_TYPE_CAT && c.n_classes >= SUAVE_WIDE_CAT_MIN_CLASSES;
}
// row stride of a categorical column's logits_layer weight [C-1, y_dim + K]
__host__ __device__ inline int cat_ldw(const suave_col_t& c, int K) {
  return col_is_wide(c) ? (c.y_dim + K + 3) / 4 * 4 : c.y_dim + K;
}

// One wide categorical column (device table of the plan).
struct WideCol {
  int32_t col;     // schema column
  int32_t C;       // classes
  int32_t yd;      // y_dim
  int32_t y_off;   // slot [y (yd) | s (K) | pad] in the internal y buffer, multiple of 4
  int32_t l_off;   // logits of classes 1..C-1 at columns l_off .. l_off + C - 2 of the wide logits buffer (multiple of 4)
  int32_t w_off;   // logits_layer weight [C-1, ldw]
  int32_t ldw;     // round_up(yd + K, 4)
  int32_t t_off;   // transposed copy [ldw, round_up(C-1, 4)] in the workspace's wideT buffer (row-major W^T: the split
                   // data-gradient GEMM's operand form)
};

// Encoder embedding block of one wide categorical column: row `code` of the block [C, ld] at e_off is the
// column's contribution to the first encoder layer (see suave_col_t::enc_embed).
struct EmbedCol {
  int32_t x_index;  // column in codes
  int32_t C;
  int32_t e_off;    // flat-parameter offset of the block
  int32_t cls_begin;  // classes of the embedded columns before this one
};


// A chunk of consecutive schema columns handled by one loss-kernel block column.
struct ColChunk {
  int32_t col_begin, col_end;
  int32_t y_begin, y_len;
};

}  // namespace suave

// Opaque plan (C-ABI name).
struct suave_plan {
  suave_model_t model;
  std::vector<suave_col_t> cols_host;
  std::vector<suave::ColChunk> chunks_host;
  std::vector<suave::ColChunk> chunks_small_host;  // finer set for small batches; on the device right behind chunks_host
  suave_col_t* cols_dev = nullptr;
  suave::ColChunk* chunks_dev = nullptr;
  int device = 0;
  int n_cu = 256;
  int32_t* enc_map_dev = nullptr;  // encoder input feature -> packed-cell source (k_expand)
  int32_t* cls_off_dev = nullptr;
  int32_t* xsrc_dev = nullptr;
  int loss_ldx = 1, loss_ldm = 4;
  // padded leading dimensions (floats)
  int ldI, ldE, ldK, ldD, ldY;
  int ldH[SUAVE_MAX_HIDDEN];
  int enc_out_width;  // K + 2*K*D
  int total_classes;  // sum of n_classes over cat/ordinal columns
  int max_chunk_y;    // widest y chunk
  int max_classes;
  int kmax;           // template bucket for K (8, 16 or 32)
  int hw_max;         // widest per-chunk range of likelihood-head weights
  int loss_blocks;    // grid.x of the loss kernel
  size_t loss_lds_bytes;
  int loss_wide16;    // a narrow categorical / ordinal column needs the 16-wide register blocks of the likelihood kernel
  // wide categorical columns
  std::vector<suave::WideCol> wide_host;
  suave::WideCol* wide_dev = nullptr;
  int32_t* wide_cum_dev[3] = {nullptr, nullptr, nullptr};  // tile prefixes: forward N tiles, data-grad N tiles, weight-grad tiles
  int wide_cum_total[3] = {0, 0, 0};
  int ldL = 0;        // columns of the wide logits buffer
  int wide_w0 = 0;    // first float of the wide weight region
  int wide_len = 0;   // floats of the wide weight region
  int wide_t_len = 0; // floats of its transposed copy (WideCol::t_off)
  int n_narrow = 0;   // columns handled by the fused likelihood kernel
  // encoder embedding blocks of the wide categorical columns (first encoder layer)
  std::vector<suave::EmbedCol> embed_host;
  suave::EmbedCol* embed_dev = nullptr;
  int embed_classes_total = 0;
  // fused decode (decode.hip): the heads' linear maps as one more MFMA layer -- u index of each column's first output
  // row, and per u index {flat-parameter offset of the weight row, y_off, y_dim}
  std::vector<int32_t> u_off_host;
  int32_t* u_off_dev = nullptr;
  int32_t* u_map_dev = nullptr;
  int n_u = 0;
  // ... and the kernel's schedule: output tiles in merged order, the columns each completes, dealt to two waves
  std::vector<int16_t> dec_step_tile, dec_step_begin, dec_step_mid, dec_col_list;
  void* static_scratch = nullptr;  // this device's copy of api.hip's static scratch array (suave_decode_sample_static)
};

namespace suave {

// ----- counter-based RNG (Philox-4x32-10) -----------------------------------
struct Philox {
  uint32_t key0, key1;
  __device__ __forceinline__ Philox(uint64_t seed)
      : key0(static_cast<uint32_t>(seed)), key1(static_cast<uint32_t>(seed >> 32)) {}
  __device__ __forceinline__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
    uint32_t k0 = key0, k1 = key1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: both are
      // quarter-rate instructions, and the generators are where the sampling kernels spend their VALU time
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
      uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
  }
};

// uniform in (0,1): 23 random bits, (m + 1/2) / 2^23 -- exact in binary32, so never 0 and never 1 (with 24 bits the
// top value m + 1/2 = 2^24 - 1/2 rounds to 2^24: u = 1, an Exp(1) draw of 0 and a Gumbel draw of +inf once in 2^24)
__device__ __forceinline__ float u01(uint32_t r) { return (static_cast<float>(r >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// ln(x) for NORMAL positive x (no denormal scaling, unlike __logf): v_log_f32 and one multiply
__device__ __forceinline__ float ln_normal(float x) { return 0.6931471806f * __builtin_amdgcn_logf(x); }
// Exp(1) draw from a 32-bit word: u01 is in [2^-24, 1 - 2^-24], the draw in [6e-8, 16.7)
__device__ __forceinline__ float exp_draw(uint32_t r) { return -ln_normal(u01(r)); }

// two N(0,1) from two 32-bit words (Box-Muller)
__device__ __forceinline__ float2 normal2(uint32_t r0, uint32_t r1) {
  float u = u01(r0), v = u01(r1);
  float rad = sqrtf(-2.0f * __logf(u));
  float s, c;
  __sincosf(6.28318530718f * v, &s, &c);
  return make_float2(rad * c, rad * s);
}

// RNG stream ids (counter word c2)
enum : uint32_t {
  kStreamGumbel = 1,
  kStreamEps = 2,
  kStreamEncDrop = 16,  // + layer
  kStreamDecDrop = 32,  // + layer
  kStreamSampleNormal = 64,
  kStreamSampleUniform = 65,
  kStreamSampleExp = 66,
  kStreamPriorU = 67,
  kStreamPriorEps = 68,
  kStreamCondU = 69,
  kStreamCondExp = 70,
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// softplus with torch's threshold; log1p(t) by its series below t = 0.02 (relative error < 2e-6)
__device__ __forceinline__ float softplus_f(float x) {
  if (x > 20.0f) return x;
  const float t = __expf(x);
  return t < 0.02f ? t * (1.0f - t * (0.5f - t * (1.0f / 3.0f))) : __logf(1.0f + t);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_grad_f(float x) { return x > 20.0f ? 1.0f : sigmoid_f(x); }

// ----- GEMM launcher (gemm.hip) ----------------------------------------------
enum GemmEpi : int {
  kEpiBias = 0,       // C = acc + bias
  kEpiReluDrop = 1,   // C = dropout(relu(acc + bias))
  kEpiMaskBwd = 2,    // C = acc * (H > 0 ? inv_keep : 0)     (dropout+relu backward)
  kEpiSlab = 3,       // split-K partial slab, no bias
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // kEpiBias / kEpiReluDrop (may be null)
  const float* H;         // kEpiMaskBwd: forward activation, same shape/ld as C;
                          // kEpiReluDrop: optional addend of the pre-activation (same shape/ld as C, may be C itself)
  const uint8_t* keep;    // kEpiReluDrop: injected keep mask [M, N] (ld = N) or null
  const int32_t* row_ids; // table row id per data row (RNG key) or null -> row_begin + data row
  int64_t row_begin;
  int row_vdiv;           // > 1: output row i is virtual row (data row i / row_vdiv, component i % row_vdiv) and its
                          // RNG key is table row * row_vdiv + component, so the draw follows the table row
  int M, N, K;
  int lda, ldb, ldc;
  int a_kmajor, b_kmajor; // element (i,k) at A[k*lda+i] if a_kmajor else A[i*lda+k]; same for B(k,j)
  int epi;
  int splits;             // kEpiSlab: K is split into `splits` ranges, slab s at C + s*slab_stride
  int64_t slab_stride;
  int ones_col;           // kEpiSlab: virtual B column index holding 1.0 (bias gradient) or -1
  float drop_p;
  uint64_t seed, step;
  uint32_t stream_id;
  // Selected-component encoder head (rows sorted by component, see latent.hip k_assign / k_group_scatter):
  //   grouped == 1: rows of A and C are sorted positions, read / written through perm; row r of B is
  //                 weight row head_row(q0 + r, component of the tile)        (forward and data gradient)
  //   grouped == 2: the k rows of A and B are sorted positions read through perm; split s is chunk s
  //                 of kGroupRows positions                                    (weight gradient)
  int grouped;
  const int32_t* perm;        // sorted position -> batch row, -1 = padding
  const int32_t* chunk_comp;  // chunk of kGroupRows sorted positions -> component, -1 = unused chunk
  int hK, hD, hldK, hldD, q0;
  // Activation sign bits instead of the activation itself (backward of a ReLU / dropout layer only needs h > 0): the
  // forward epilogue (kEpiReluDrop) writes, per 64 x 64 wave tile, one 64-bit word per LANE -- bit
  // ((mi*4 + g)*2 + h)*4 + e says that element e of the lane's quad in epilogue round (mi, g, h) is positive -- and the
  // data-gradient epilogue (kEpiMaskBwd) of the SAME matrix shape (same tiles, same lane -> element map) reads its
  // lane's word back: 1 bit instead of 4 bytes per activation, one coalesced 512-byte access per wave either way.
  // Indexed by (tile_m, tile_n, wave, lane); gemm_mask_words() sizes the buffer.  Null: not written / H is read.
  uint64_t* mask_out;
  const uint64_t* mask_in;
  float* bias_out;            // kEpiSlab on the small-batch kernel, one split: the ones column's sums go here (the bias
                              // gradient) instead of column ones_col of C, and C is the weight gradient itself
  float a_scale;              // arithmetic mode 4 (fp16 planes): power of two the A operand is multiplied by before it is
                              // cut into planes (gradient operands are ~1 / batch rows: keeps them clear of the fp16
                              // subnormals); the product is scaled back in the epilogue.  <= 0 means 1.
};

// Sorted positions are handed out in chunks of kGroupRows rows of one component (multiple of every GEMM row tile).
constexpr int kGroupRows = 256;

// Compact per-row head layout [logits (ldK) | mu_k (ldD) | logvar_k (ldD)] -> row of the stacked encoder head
// [logits (K) ; mu (K*D) ; logvar (K*D)] for component c; -1 for pad columns.
__host__ __device__ __forceinline__ int head_row(int q, int c, int K, int D, int ldK, int ldD) {
  if (q < ldK) return q < K ? q : -1;
  q -= ldK;
  if (q < ldD) return q < D ? K + c * D + q : -1;
  q -= ldD;
  return q < D ? K + K * D + c * D + q : -1;
}

int launch_gemm(const GemmArgs& a, int n_cu, hipStream_t stream);

// Grouped GEMMs of all wide categorical columns in one launch (gemm.hip).
//   mode 0: logits  L_j[rows, C-1]   = slot_j[rows, yd+K] * W_j^T                  (decoder.py:201-206)
//   mode 1: d slot_j[rows, ldw]      = g_j[rows, C-1] * W_j          (g overwrote L in place)
//   mode 2: d W_j[C-1, ldw] slabs    = g_j^T * slot_j, split over the rows
struct WideGemmArgs {
  const WideCol* cols;
  const int32_t* cum;  // [n_wide + 1] tile prefix of this mode
  int n_wide, cum_total;
  const float* params;
  float* y;
  int ldY;
  float* L;
  int ldL;
  int rows, K;
  float* slab;          // mode 2: [splits][wide_len], mirrors the wide weight region
  int64_t slab_stride;  // wide_len
  int splits;
  int wide_w0;
  const float* wT;      // mode 1, split arithmetic: transposed weights (WideCol::t_off)
  float grad_scale;     // modes 1 and 2: GemmArgs::a_scale of the logit gradients
};
int launch_gemm_wide(const WideGemmArgs& a, int mode, hipStream_t stream);
struct GemmSmallScope {  // alive inside an ELBO step: small batches may take gemm_small_kernel (see gemm_small_rows)
  GemmSmallScope();
  ~GemmSmallScope();
  GemmSmallScope(const GemmSmallScope&) = delete;
  GemmSmallScope& operator=(const GemmSmallScope&) = delete;
};
bool gemm_wgrad_direct_ok(int64_t rows, int splits);  // see gemm.hip
int gemm_small_rows_limit_set(int rows);
int64_t gemm_small_rows_limit();  // the row count up to which gemm_small_rows can hold (whatever the scope / mode)
bool gemm_small_rows(int64_t M);  // the forward forms of an M-row batch take the split-K small-batch kernel (no sign words)
int64_t gemm_mask_words(int64_t M, int N);  // 64-bit words of GemmArgs::mask_out for an M x N activation
int launch_gemm_pair(const GemmArgs& d, const GemmArgs& w, int n_cu, hipStream_t stream);
int gemm_split_mode_set(int mode);  // 0 on success (suave_gemm_arithmetic)
int gemm_split_mode();  // 0 exact-fp32 MFMA, 3 bf16 planes per operand on the row-major forms (gemm.hip)

// ----- optional per-kernel-class timing with HIP events (suave_profile_*) -----
enum ProfClass : int { kProfGemm = 0, kProfLoss, kProfLatent, kProfExpand, kProfReduce, kProfAdam, kProfSample, kProfOther, kProfCount };
struct ProfScope {
  int cls;
  hipStream_t stream;
  bool on;
  ProfScope(int cls, hipStream_t stream);
  ~ProfScope();
};

}  // namespace suave
