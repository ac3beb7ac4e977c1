// Internal helpers shared by the HIP translation units of libcmh.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/cmh.h"

namespace cmh {

// thread-local error text behind cmh_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define CMH_CHECK_ARG(cond, ...) \
  do {                           \
    if (!(cond)) return ::cmh::fail(CMH_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define CMH_CHECK_LAUNCH(what)                                                            \
  do {                                                                                    \
    hipError_t e__ = hipGetLastError();                                                   \
    if (e__ != hipSuccess)                                                                \
      return ::cmh::fail(CMH_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e__));          \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- bf16 (stored as uint16_t) ---------------------------------------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(static_cast<uint32_t>(v) << 16);
}
// round-to-nearest-even; NaN stays NaN (quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<bf16_t>((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<bf16_t>(u >> 16);
}

// two f32 -> packed bf16x2 (lo = a) in ONE instruction (v_cvt_pk_bf16_f32, RNE, NaN-preserving on gfx950)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  const f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}

// two f32 -> packed f16x2 (lo = a), RNE
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  const f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_t));
}
__device__ __forceinline__ float f16lo_to_f32(uint32_t w) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  return static_cast<float>(__builtin_bit_cast(h2_t, w)[0]);
}
__device__ __forceinline__ float f16hi_to_f32(uint32_t w) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  return static_cast<float>(__builtin_bit_cast(h2_t, w)[1]);
}

// ---- element kinds of the backward building blocks (CMH_KIND_*) ----------------------------------------------
constexpr int kF32 = 0, kBF16 = 1, kF16 = 2;

__device__ __forceinline__ float load_as_f32(const void* p, size_t i, int kind) {
  if (kind == kF32) return static_cast<const float*>(p)[i];
  const uint16_t h = static_cast<const uint16_t*>(p)[i];
  if (kind == kBF16) return bf16_to_f32(h);
  return static_cast<float>(__builtin_bit_cast(_Float16, h));
}
__device__ __forceinline__ void store_from_f32(void* p, size_t i, int kind, float v) {
  if (kind == kF32) static_cast<float*>(p)[i] = v;
  else if (kind == kBF16) static_cast<uint16_t*>(p)[i] = f32_to_bf16(v);
  else static_cast<uint16_t*>(p)[i] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
}

// 4 consecutive elements (i % 4 == 0, 8/16-byte aligned)
__device__ __forceinline__ float4 load4_as_f32(const void* p, size_t i, int kind) {
  if (kind == kF32) return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
  const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p) + i);
  if (kind == kBF16) return float4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                   __uint_as_float(u.y & 0xffff0000u)};
  return float4{f16lo_to_f32(u.x), f16hi_to_f32(u.x), f16lo_to_f32(u.y), f16hi_to_f32(u.y)};
}
__device__ __forceinline__ void store4_from_f32(void* p, size_t i, int kind, float4 v) {
  if (kind == kF32) *reinterpret_cast<float4*>(static_cast<float*>(p) + i) = v;
  else if (kind == kBF16) *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p) + i) = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  else *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p) + i) = uint2{pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w)};
}

// ---- epilogue flags of the GEMM ---------------------------------------------------------------
enum : int {
  EPI_BIAS = 1,       // + bias[n]
  EPI_QUICKGELU = 2,  // x * sigmoid(1.702 x)      (model/base/model.py:162-164)
  EPI_RESIDUAL = 4,   // + residual[m,n] (f32)
  EPI_OUT_BF16 = 8,   // store bf16 instead of f32
  EPI_GELU = 16,      // exact GELU 0.5 x (1 + erf(x/sqrt 2))  (nn.GELU(), MITH ResidualMLPs model/MITH.py:224-233)
  EPI_RELU = 32,      // max(x, 0)
  EPI_RES_F16 = 64,   // residual is IEEE fp16 (the bf16 mode's residual stream, see encoders.hip)
  EPI_OUT_F16 = 128,  // store fp16 instead of f32
  EPI_MUL_DQGELU = 1024, // acc *= QuickGELU'(aux[m,n]); aux (in the residual slot) has the OUTPUT's type (bf16 with EPI_OUT_BF16,
                         // else f32): the dgrad through c_fc's activation.  N % 256 == 0 (wide kernel only); 256/512 are ablation bits
  EPI_SCALE = 2048,      // acc *= alpha * colscale[n] before the bias: dequantisation of fp8 operands (launch_gemm_fp8)
  EPI_OUT_FP8 = 4096,    // store OCP e4m3 of clamp(v * oscale, +-448) (the next fp8 GEMM's operand)
  EPI_SAVE_PRE = 8192    // (bf16 out, wide kernel) also store the value BEFORE the activation, as bf16 [M, N], through the residual
                         // slot's pointer: the training forward of c_fc keeps the pre-activation for the backward
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// C[M,N] = epilogue(A[M,K] . W[N,K]^T).  A/W dtype = dt (f32 or bf16); residual f32; out f32|bf16.
// m_dev (optional, device int32): the real row count when M is only an upper bound (packed text rows); m_hint: a likely value of
// it for the tile-height choice (never for correctness)
bool gemm_wide_enabled();          // false under the CMH_GEMM_WIDE=0 diagnostic (gemm.hip)
// One GEMM of a grouped launch (gemm_wide.hip, template parameter GRP): two problems with the same arithmetic (dt, output kind,
// epilogue flags) in ONE persistent grid - the same layer of the image and the text tower.  launch_gemm_grouped runs them as two plain
// launches when the wide kernel cannot take both (N % 256, few rows, CMH_GEMM_WIDE=0): results never depend on the grouping.
struct GemmProblem {
  const void* A; const void* W; const float* bias; const float* residual; void* out;
  int M, N, K;
  const int32_t* m_dev; int m_hint;                           // device-side row count (M = upper bound) and its likely value
  const float* colscale; float alpha, oscale;                 // fp8 operands (launch_gemm_fp8's arguments); unused otherwise
};
int launch_gemm_grouped(int dt, const GemmProblem& a, const GemmProblem& b, int epi, hipStream_t st);
bool gemm_rows_takes(int M, int N, int K, int epi);   // gemm_rows.hip: this launch would run on the few-row kernel
int launch_gemm(int dt, const void* A, const void* W, const float* bias, const float* residual,
                void* out, int M, int N, int K, int epi, hipStream_t st, const int32_t* m_dev = nullptr, int m_hint = -1);
// C = epilogue(alpha * colscale[n] * (A8 . W8^T)): OCP e4m3 operands [M,K] / [N,K] (K % 128 == 0, N % 256 == 0), f32 accumulate;
// out f32 | bf16 | fp16 | (EPI_OUT_FP8) e4m3 of v * oscale.  EPI_SCALE is implied.
int launch_gemm_fp8(const void* A8, const void* W8, const float* colscale, float alpha, const float* bias, const float* residual,
                    void* out, float oscale, int M, int N, int K, int epi, hipStream_t st, const int32_t* m_dev = nullptr, int m_hint = -1);

// gemm_wide.hip, TN form (wgrad): out[Mm, Nn] f32 = sum_k Xk[k, m] * Wk[k, n], bf16 operands [Kd, Mm] / [Kd, Nn] row-major
bool gemm_wide_tn_supported(int Mm, int Nn, int Kd);
// colsum_partial (optional): [*colsum_slices, Mm] partial column sums of Xk (wgrad's bias gradient, first stage), <= 64 slices
int launch_gemm_wide_tn(const void* Xk, const void* Wk, float* out, float* partials, size_t part_bytes, int Mm, int Nn, int Kd,
                        hipStream_t st, float* colsum_partial = nullptr, int* colsum_slices = nullptr);
// several wgrad GEMMs in one launch (gemm_wide.hip: gemm_wide_tn_multi_kernel); out_i[Mm, Nn] = Xk_i^T Wk_i over Kd rows
struct TnMultiJob { const void* Xk; const void* Wk; float* out; float* colsum; int Mm, Nn, Kd; };
bool gemm_wide_tn_multi_enabled();
bool gemm_wide_tn_multi_fits(const TnMultiJob* jobs, int n);
int launch_gemm_wide_tn_multi(const TnMultiJob* jobs, int n, float* partials, size_t part_bytes, hipStream_t st, int* slices);

// LayerNorm over rows of x[M,d] (f32) -> out (f32 or bf16 per out_bf16). rows optionally gathered:
// row r reads x[row_index[r]] when row_index != null.
int launch_layernorm(const float* x, const int32_t* row_index, const float* w, const float* b,
                     void* out, int out_bf16, int M, int d, hipStream_t st);
// same, x either f32 or (x_f16) the fp16 residual stream of the bf16 mode
// m_dev (optional, device int32): the real row count when M is only an upper bound (packed text rows: no host round trip)
int launch_layernorm_x(const void* x, int x_f16, const int32_t* row_index, const float* w, const float* b,
                       void* out, int out_bf16, int M, int d, hipStream_t st, const int32_t* m_dev = nullptr);

// the bf16 mode's LayerNorm (fp16 stream -> bf16 rows) of two row sets in ONE launch (the lock-step pair path); CMH_OK = launched,
// 1 = no kernel for this pair of widths (or CMH_PAIR_KERNELS=0): launch the two singly; < 0 = the launch itself failed
int launch_layernorm_h2b_pair(const void* x0, const float* w0, const float* b0, void* out0, int M0, int d0, const int32_t* md0,
                               const void* x1, const float* w1, const float* b1, void* out1, int M1, int d1, const int32_t* md1,
                               hipStream_t st);
// image [B,3,R,R] f32 -> patches [B*g*g, 3*p*p] (dt)
int launch_patchify(const float* image, void* patches, int dt, int B, int R, int p, hipStream_t st);
// tokens: x[b,0]=cls+pos[0]; x[b,1+i]=patch_out[b*g2+i]+pos[1+i]; then ln_pre -> x f32|f16 [B*(g2+1), d]
int launch_vit_assemble_lnpre(const float* patch_out, const float* cls, const float* pos,
                              const float* lnw, const float* lnb, void* x, int x_f16, int B, int g2, int d,
                              hipStream_t st);
// x[b,t] = tok_emb[tokens[b,t]] + pos[t]; eot_row[b] = b*L + argmax_t tokens[b,t]
int launch_text_embed(const int64_t* tokens, const float* tok_emb, const float* pos, void* x, int x_f16,
                      int32_t* eot_row, int B, int L, int d, int vocab, hipStream_t st);
// packed text rows (tokens up to the EOT only): offsets plan and the embedding into packed rows
// (kpm + eot_pos: the all-token form - rows up to the last unpadded position, the EOT's position into eot_pos; launch_text_embed_packed
// with eot_is_pos then turns eot_row[b] from that position into the packed row)
int launch_text_pack_plan(const int64_t* tokens, int B, int L, int32_t* seq_off, hipStream_t st, const uint8_t* kpm = nullptr,
                          int32_t* eot_pos = nullptr);
int launch_text_embed_packed(const int64_t* tokens, const float* tok_emb, const float* pos, void* x, int x_f16, int32_t* eot_row,
                             int B, int L, int d, int vocab, const int32_t* seq_off, hipStream_t st, bool eot_is_pos = false);
bool text_token_packing();      // CMH_TEXT_PACK_TOKENS / cmh_set_text_token_packing (encoders.hip)
// packed f32 rows <-> the dense [B, L, E] layout (zeros behind each caption's kept rows); EOT rows as dense indices
int launch_unpack_token_rows(const float* packed, const int32_t* seq_off, float* dense, int B, int L, int E, const int32_t* eot_packed,
                             int32_t* eot_dense, hipStream_t st);
int launch_pack_token_rows(const float* dense, const int32_t* seq_off, float* packed, int B, int L, int E, hipStream_t st);
// cls_row[b] = b*T
int launch_iota_rows(int32_t* rows, int B, int T, hipStream_t st);
int launch_scatter_rows(const void* src, const int32_t* rows, void* dst, int B, int row_bytes, hipStream_t st);
bool pooled_tail_enabled();     // encoders.hip: cmh_set_pooled_tail / CMH_POOLED_TAIL (default on)
int launch_gather_rows2(const void* srcA, void* dstA, int bytesA, const void* srcB, void* dstB, int bytesB, const int32_t* rows, int B,
                        hipStream_t st);
// gemm_wide.hip: split-K for few-tile / long-K products (wgrad); plan returns S (1 = do not split), partials = S*M*N floats
int gemm_wide_splitk_plan(int dt, int M, int N, int K);
int launch_gemm_wide_splitk(int dt, const void* A, const void* W, float* out, float* partials, int S, int M, int N, int K,
                            hipStream_t st);
// norm_embed.hip (training forward): LayerNorm between any stream kinds; token assembly without ln_pre
int launch_layernorm_any(const void* x, int x_kind, const int32_t* row_index, const float* w, const float* b, void* out,
                         int out_kind, int M, int d, hipStream_t st);
int launch_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int B, int g2, int d, hipStream_t st);
// attention_bwd.hip: backward of launch_attention_varlen (seq_off may be NULL)
int launch_attention_backward(int dtype, const void* qkv, const void* o, const void* dout, void* dqkv, int B, int T, int d, int causal,
                              const uint8_t* key_padding_mask, const int32_t* seq_off, hipStream_t st);
// backward.hip: dst[c*dst_ld + r] = cast(src[r*cols + c]); LayerNorm backward with optional gathered rows (x / dx rows = row_index[r])
int launch_transpose(const void* src, int skind, void* dst, int dkind, int rows, int cols, int dst_ld, hipStream_t st,
                     float* colsum_partial = nullptr);   // + [ceil(rows/64), cols] partial column sums (vectorised path only)
bool transpose_is_vectorised(const void* src, const void* dst, int rows, int cols, int dst_ld);
// n <= 4 transposes of [R[j], C[j]] matrices of one element kind in one launch (R, C multiples of 4; dst [C, R])
int launch_transpose_multi(const void* const* src, void* const* dst, const int* R, const int* C, int n, int kind, hipStream_t st);
int launch_colsum_final(const float* partial, int slices, int cols, float* out, hipStream_t st);
// deferred final stage of column reductions: up to kMax (partial [slices, cols] -> out [cols]) jobs for one launch
struct FinalJobs {
  static constexpr int kMax = 8;
  int n = 0;
  const float* partial[kMax];
  int slices[kMax];
  int cols[kMax];
  float* out[kMax];
  void add(const float* p, int s, int c, float* o) { partial[n] = p; slices[n] = s; cols[n] = c; out[n] = o; ++n; }
};
int launch_final_jobs(FinalJobs& jobs, hipStream_t st);
int launch_colsum_partial(const void* x, int kind, int rows, int cols, float* partial, hipStream_t st);   // [ceil(rows/64), cols]
int launch_layernorm_backward(const void* x, int x_kind, const void* dy, int dy_kind, const float* gamma, const int32_t* row_index,
                              int M, int d, float* dx, int accumulate, float* dgamma, float* dbeta, void* workspace,
                              size_t workspace_bytes, hipStream_t st, void* dx_bf16 = nullptr,   // + a bf16 copy of the new dx
                              FinalJobs* defer = nullptr,                                         // queue dgamma / dbeta's final stage
                              const void* acc16 = nullptr,      // the new dx is added to THIS bf16 buffer's values (accumulate = 0)
                              bool write_f32 = true);           // false: only the bf16 copy leaves (the 16-bit gradient stream)

// attention over qkv [B*T, 3d] (dt) -> o [B*T, d] (dt); heads = d/64; causal adds the -inf triu mask
int launch_attention(const void* qkv, void* o, int dt, int B, int T, int d, int causal,
                     const uint8_t* key_padding_mask, hipStream_t st);
int launch_attention_varlen(const void* qkv, void* o, int dt, int B, int T, int d, int causal,
                            const uint8_t* key_padding_mask, const int32_t* seq_off, hipStream_t st,
                            float o8_inv_scale = 0.f);   // > 0 (bf16 qkv, T <= 128): o is e4m3 of o * o8_inv_scale (fp8 mode)
// fp8.hip: LayerNorm of the fp16 residual stream straight to e4m3 (y * inv_scale); max |x| into a device scalar (running maximum)
int launch_layernorm_q(const void* x_f16, const float* w, const float* b, void* out_fp8, float inv_scale, int M, int d, hipStream_t st,
                       const int32_t* m_dev = nullptr);
int launch_amax(const void* x, int kind, size_t n, float* out, hipStream_t st);

// y[M,N] f32 = act((x[M,K] . w[N,K]^T + bias) * mask*keep_scale); x,w dtype dt; any N, K%4==0, K<=4096
int launch_small_linear(int dt, const void* x, const void* w, const float* bias, const float* mask,
                        float keep_scale, int act, float* y, int M, int N, int K, hipStream_t st);

}  // namespace cmh
