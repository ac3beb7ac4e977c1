// C-ABI entry points of the two CLIP towers: kernel sequencing only (every kernel lives in gemm.hip,
// norm_embed.hip, attention.hip, heads.hip).  Nothing here allocates or synchronises: all launches go
// to the caller's stream and all scratch comes from the caller's workspace, so a whole encode is
// graph-capturable.
//
// Data layout in HBM (per call, M = B*T rows, d = width, e = GEMM element size 4|2):
//   x    f32 [M, d]    residual stream.  f32 mode: fp32 (the parity mode for the trainers' model.float()).
//                      bf16 mode (throughput): IEEE fp16, as a raw build_model CLIP keeps it on a GPU (convert_weights,
//                      model/base/model.py:391-412): halves the bytes of the 4 read-modify-write passes per layer
//                      (2 LayerNorms, 2 residual GEMMs).
//                      CMH_RESID_F16=0, widths that are not a multiple of 256 or a taps request keep it fp32.
//   h    e   [M, d]    LayerNorm output / attention output (GEMM A operand)
//   qkv  e   [M, 3d]   packed in_proj output            (vision: patch_out f32 [B*g2, d] aliases it)
//   mlp  e   [M, 4d]   c_fc output after QuickGELU      (vision: patches e [B*g2, 3p^2] aliases it)
//   rows i32 [B]       pooled row per sample (class token / EOT token)
//   pool e   [B, d]    ln_post / ln_final of the pooled rows
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "cmh_common.h"

namespace cmh {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* p) : base(static_cast<char*>(p)) {}
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
};

struct TowerBufs {
  float* x;
  int xh = 0;   // x holds fp16 (bf16 mode, see header)
  void* h;
  void* qkv;
  void* mlp;
  int32_t* rows;
  void* pool;
  int32_t* seq;     // [B + 2] packed text: row offsets of the captions, seq[B] = the packed row count (read by the kernels themselves)
  size_t total;
};

static TowerBufs carve(void* ws, size_t M, size_t B, size_t d, size_t e, size_t extra_qkv, size_t extra_mlp) {
  Arena a(ws);
  TowerBufs t;
  t.x = static_cast<float*>(a.take(M * d * 4));
  t.h = a.take(M * d * e);
  size_t qkv_b = M * 3 * d * e, mlp_b = M * 4 * d * e;
  t.qkv = a.take(qkv_b > extra_qkv ? qkv_b : extra_qkv);
  t.mlp = a.take(mlp_b > extra_mlp ? mlp_b : extra_mlp);
  t.rows = static_cast<int32_t*>(a.take(B * 4));
  t.pool = a.take(B * d * e);
  t.seq = static_cast<int32_t*>(a.take((B + 2) * 4));
  t.total = a.off;
  return t;
}

static int tap(const cmh_taps* taps, int idx, const float* x, size_t bytes, hipStream_t st) {
  if (!taps || idx >= taps->count || !taps->ptrs[idx]) return CMH_OK;
  if (hipMemcpyAsync(taps->ptrs[idx], x, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "tap copy failed");
  return CMH_OK;
}

// One ResidualAttentionBlock (model/base/model.py:191-196):
//   x += out_proj(attn(in_proj(ln_1(x))));  x += c_proj(QuickGELU(c_fc(ln_2(x))))
// fp8 mode (CMH_FP8): the four GEMMs on e4m3 operands.  Activations are quantised by their producers with the per-tensor
// scales of cmh_block_weights.act_scale (LayerNorm -> launch_layernorm_q, attention -> its fp8 store, QuickGELU -> the c_fc
// epilogue); the GEMM epilogues undo act_scale * colscale[n].  Residual stream fp16, qkv bf16 (attention is the bf16 kernel).
static int run_block_fp8(const cmh_block_weights& w, const TowerBufs& t, int B, int T, int d, int causal, const uint8_t* kpm,
                         hipStream_t st, int M, const int32_t* seq_off, const int32_t* md = nullptr, int mh = -1) {
  const float* a = w.act_scale;
  CMH_CHECK_ARG(t.xh, "fp8 mode runs on the fp16 residual stream (width %% 256 == 0, no taps)");
  CMH_CHECK_ARG(w.in_proj_cs && w.out_proj_cs && w.fc_cs && w.proj_cs, "fp8 mode: weight scales missing");
  CMH_CHECK_ARG(a[0] > 0.f && a[1] > 0.f && a[2] > 0.f && a[3] > 0.f, "fp8 mode: activation scales missing (run the calibration pass)");
  const int rx = EPI_BIAS | EPI_RESIDUAL | EPI_RES_F16 | EPI_OUT_F16;
  int rc;
  if ((rc = launch_layernorm_q(t.x, w.ln1_w, w.ln1_b, t.h, 1.0f / a[0], M, d, st, md))) return rc;
  if ((rc = launch_gemm_fp8(t.h, w.in_proj_w, w.in_proj_cs, a[0], w.in_proj_b, nullptr, t.qkv, 1.f, M, 3 * d, d, EPI_BIAS | EPI_OUT_BF16, st, md, mh))) return rc;
  if ((rc = launch_attention_varlen(t.qkv, t.h, CMH_BF16, B, T, d, causal, kpm, seq_off, st, 1.0f / a[1]))) return rc;
  if ((rc = launch_gemm_fp8(t.h, w.out_proj_w, w.out_proj_cs, a[1], w.out_proj_b, t.x, t.x, 1.f, M, d, d, rx, st, md, mh))) return rc;
  if ((rc = launch_layernorm_q(t.x, w.ln2_w, w.ln2_b, t.h, 1.0f / a[2], M, d, st, md))) return rc;
  if ((rc = launch_gemm_fp8(t.h, w.fc_w, w.fc_cs, a[2], w.fc_b, nullptr, t.mlp, 1.0f / a[3], M, 4 * d, d,
                            EPI_BIAS | EPI_QUICKGELU | EPI_OUT_FP8, st, md, mh))) return rc;
  if ((rc = launch_gemm_fp8(t.mlp, w.proj_w, w.proj_cs, a[3], w.proj_b, t.x, t.x, 1.f, M, d, 4 * d, rx, st, md, mh))) return rc;
  return CMH_OK;
}

static int run_block(const cmh_block_weights& w, int dt, const TowerBufs& t, int B, int T, int d, int causal,
                     const uint8_t* kpm, hipStream_t st, int rows = -1, const int32_t* seq_off = nullptr,
                     float* amax = nullptr,     // amax [4] (bf16 mode): running maxima of the four GEMM inputs (fp8 calibration)
                     const int32_t* md = nullptr, int mh = -1) {   // md: the packed row count on the device (rows = upper bound)
  const int M = rows >= 0 ? rows : B * T;      // packed variable-length text: `rows` real rows, T = the longest sequence
  if (dt == CMH_FP8) return run_block_fp8(w, t, B, T, d, causal, kpm, st, M, seq_off, md, mh);
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  const int rx = EPI_BIAS | EPI_RESIDUAL | (t.xh ? EPI_RES_F16 | EPI_OUT_F16 : 0);
  int rc;
  if (amax) {   // the calibration pass: the same kernels, with a reduction after each producer
    CMH_CHECK_ARG(dt == CMH_BF16, "fp8 calibration runs in bf16 mode");
    const size_t n = static_cast<size_t>(M) * d;
    if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w.ln1_w, w.ln1_b, t.h, 1, M, d, st))) return rc;
    if ((rc = launch_amax(t.h, kBF16, n, amax + 0, st))) return rc;
    if ((rc = launch_gemm(dt, t.h, w.in_proj_w, w.in_proj_b, nullptr, t.qkv, M, 3 * d, d, EPI_BIAS | obf, st))) return rc;
    if ((rc = launch_attention_varlen(t.qkv, t.h, dt, B, T, d, causal, kpm, seq_off, st))) return rc;
    if ((rc = launch_amax(t.h, kBF16, n, amax + 1, st))) return rc;
    if ((rc = launch_gemm(dt, t.h, w.out_proj_w, w.out_proj_b, t.x, t.x, M, d, d, rx, st))) return rc;
    if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w.ln2_w, w.ln2_b, t.h, 1, M, d, st))) return rc;
    if ((rc = launch_amax(t.h, kBF16, n, amax + 2, st))) return rc;
    if ((rc = launch_gemm(dt, t.h, w.fc_w, w.fc_b, nullptr, t.mlp, M, 4 * d, d, EPI_BIAS | EPI_QUICKGELU | obf, st))) return rc;
    if ((rc = launch_amax(t.mlp, kBF16, n * 4, amax + 3, st))) return rc;
    return launch_gemm(dt, t.mlp, w.proj_w, w.proj_b, t.x, t.x, M, d, 4 * d, rx, st);
  }
  if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w.ln1_w, w.ln1_b, t.h, dt == CMH_BF16, M, d, st, md))) return rc;
  if ((rc = launch_gemm(dt, t.h, w.in_proj_w, w.in_proj_b, nullptr, t.qkv, M, 3 * d, d, EPI_BIAS | obf, st, md, mh))) return rc;
  if ((rc = launch_attention_varlen(t.qkv, t.h, dt, B, T, d, causal, kpm, seq_off, st))) return rc;
  if ((rc = launch_gemm(dt, t.h, w.out_proj_w, w.out_proj_b, t.x, t.x, M, d, d, rx, st, md, mh))) return rc;
  if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w.ln2_w, w.ln2_b, t.h, dt == CMH_BF16, M, d, st, md))) return rc;
  if ((rc = launch_gemm(dt, t.h, w.fc_w, w.fc_b, nullptr, t.mlp, M, 4 * d, d, EPI_BIAS | EPI_QUICKGELU | obf, st, md, mh))) return rc;
  return launch_gemm(dt, t.mlp, w.proj_w, w.proj_b, t.x, t.x, M, d, 4 * d, rx, st, md, mh);
}

// The LAST block when only the pooled feature is wanted (encode_image / encode_text, model/base/model.py:247-250, 366-370): after its
// attention nothing mixes rows any more - out_proj, ln_2, the MLP, ln_post / ln_final and the projection are all row-wise - so only the
// B pooled rows (class token / EOT) are carried through them: three GEMMs of M = B instead of M = B*T.  Every kept row sees the
// arithmetic of the full-size path (same kernels, same K order), so the features are bit-identical.  The compact rows live in the qkv
// scratch, which is dead once the attention has run; *x_pooled [B, d] is the block's output (residual-stream type).
static int g_pooled_tail = -1;   // -1: from the environment (default on)
bool pooled_tail_enabled() {
  static const bool env_on = []() { const char* e = getenv("CMH_POOLED_TAIL"); return !(e && !strcmp(e, "0")); }();
  return g_pooled_tail < 0 ? env_on : g_pooled_tail != 0;
}

// CMH_TEXT_PACK_TOKENS=0 / cmh_set_text_token_packing(0): the all-token text trunk computes every position, as in rounds 1-4
static int g_pack_tokens = -1;
bool text_token_packing() {
  static const bool env_on = []() { const char* e = getenv("CMH_TEXT_PACK_TOKENS"); return !(e && !strcmp(e, "0")); }();
  return g_pack_tokens < 0 ? env_on : g_pack_tokens != 0;
}

static int run_block_pooled(const cmh_block_weights& w, int dtb, const TowerBufs& t, int B, int T, int d, int causal,
                            const uint8_t* kpm, hipStream_t st, int M, const int32_t* seq_off, void** x_pooled,
                            const int32_t* md = nullptr, int mh = -1,
                            bool head_done = false) {   // ln_1, QKV and the attention have run already (run_block_pair)
  const int dt = dtb == CMH_FP8 ? CMH_BF16 : dtb;
  const size_t e = dt == CMH_BF16 ? 2 : 4, xe = t.xh ? 2 : 4;
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  const int rx = EPI_BIAS | EPI_RESIDUAL | (t.xh ? EPI_RES_F16 | EPI_OUT_F16 : 0);
  const float* a = w.act_scale;
  char* scratch = static_cast<char*>(t.qkv);
  float* xp = reinterpret_cast<float*>(scratch);
  void* hp = scratch + align_up(static_cast<size_t>(B) * d * 4, 256);
  int rc;
  if (dtb == CMH_FP8) {
    CMH_CHECK_ARG(t.xh && w.in_proj_cs && w.out_proj_cs && w.fc_cs && w.proj_cs && a[0] > 0.f && a[1] > 0.f && a[2] > 0.f && a[3] > 0.f,
                  "fp8 mode: scales missing (run the calibration pass)");
    if (!head_done) {
      if ((rc = launch_layernorm_q(t.x, w.ln1_w, w.ln1_b, t.h, 1.0f / a[0], M, d, st, md))) return rc;
      if ((rc = launch_gemm_fp8(t.h, w.in_proj_w, w.in_proj_cs, a[0], w.in_proj_b, nullptr, t.qkv, 1.f, M, 3 * d, d, EPI_BIAS | EPI_OUT_BF16, st, md, mh))) return rc;
      if ((rc = launch_attention_varlen(t.qkv, t.h, CMH_BF16, B, T, d, causal, kpm, seq_off, st, 1.0f / a[1]))) return rc;
    }
    if ((rc = launch_gather_rows2(t.x, xp, static_cast<int>(d * xe), t.h, hp, d, t.rows, B, st))) return rc;
    if ((rc = launch_gemm_fp8(hp, w.out_proj_w, w.out_proj_cs, a[1], w.out_proj_b, xp, xp, 1.f, B, d, d, rx, st))) return rc;
    if ((rc = launch_layernorm_q(xp, w.ln2_w, w.ln2_b, hp, 1.0f / a[2], B, d, st))) return rc;
    if ((rc = launch_gemm_fp8(hp, w.fc_w, w.fc_cs, a[2], w.fc_b, nullptr, t.mlp, 1.0f / a[3], B, 4 * d, d,
                              EPI_BIAS | EPI_QUICKGELU | EPI_OUT_FP8, st))) return rc;
    if ((rc = launch_gemm_fp8(t.mlp, w.proj_w, w.proj_cs, a[3], w.proj_b, xp, xp, 1.f, B, d, 4 * d, rx, st))) return rc;
  } else {
    if (!head_done) {
      if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w.ln1_w, w.ln1_b, t.h, dt == CMH_BF16, M, d, st, md))) return rc;
      if ((rc = launch_gemm(dt, t.h, w.in_proj_w, w.in_proj_b, nullptr, t.qkv, M, 3 * d, d, EPI_BIAS | obf, st, md, mh))) return rc;
      if ((rc = launch_attention_varlen(t.qkv, t.h, dt, B, T, d, causal, kpm, seq_off, st))) return rc;
    }
    if ((rc = launch_gather_rows2(t.x, xp, static_cast<int>(d * xe), t.h, hp, static_cast<int>(d * e), t.rows, B, st))) return rc;
    if ((rc = launch_gemm(dt, hp, w.out_proj_w, w.out_proj_b, xp, xp, B, d, d, rx, st))) return rc;
    if ((rc = launch_layernorm_x(xp, t.xh, nullptr, w.ln2_w, w.ln2_b, hp, dt == CMH_BF16, B, d, st))) return rc;
    if ((rc = launch_gemm(dt, hp, w.fc_w, w.fc_b, nullptr, t.mlp, B, 4 * d, d, EPI_BIAS | EPI_QUICKGELU | obf, st))) return rc;
    if ((rc = launch_gemm(dt, t.mlp, w.proj_w, w.proj_b, xp, xp, B, d, 4 * d, rx, st))) return rc;
  }
  *x_pooled = xp;
  return CMH_OK;
}

// ---- both towers in lock-step (round 4: grouped launches) ---------------------------------------------------------------------------
// The image and the text tower are 12 blocks of the same four GEMMs (model/base/model.py:167-207, built twice by CLIP.__init__:
// :254-306); run one after the other - or on two streams - every GEMM is a launch of its own whose fixed third (first stage landing
// on 256 CUs at once, last tile's epilogue and store drain) the short-K text launches cannot amortise, and whose last round leaves
// CUs idle.  Here layer i of BOTH towers is one grouped launch of the wide kernel (gemm_wide.hip, GRP): ~50 GEMM launches per encoded
// batch instead of ~100, text tiles filling the image launches' last round.  Every output element sees the arithmetic of the
// single-tower path: the features are bit-identical (tests/test_gpu_grouped.py).
struct TowerRun {
  TowerBufs t;
  int dtb = 0, d = 0, B = 0, T = 0, M = 0, causal = 0;   // M: rows (an upper bound when md is set)
  const int32_t* seq_off = nullptr;                      // packed text: per-caption row offsets
  const int32_t* md = nullptr;                           // packed text: the row count on the device
  int mh = -1;                                           // ... and its likely value (tile heights only)
  bool packed_tokens = false;                            // packed text, every kept row is an output (the MITH trunk)
};

static GemmProblem problem_of(const TowerRun& r, const void* A, const void* W, const float* bias, const void* residual, void* out, int N,
                              int K, const float* colscale = nullptr, float alpha = 1.f, float oscale = 1.f) {
  return GemmProblem{A, W, bias, static_cast<const float*>(residual), out, r.M, N, K, r.md, r.mh, colscale, alpha, oscale};
}

// the full-size part of a block for both towers: ln_1, QKV, attention (then, unless `upto_attention`, out_proj, ln_2, c_fc, c_proj)
static int run_block_pair(const cmh_block_weights& wa, const cmh_block_weights& wb, const TowerRun& a, const TowerRun& b, hipStream_t st,
                          bool upto_attention) {
  const int dtb = a.dtb;
  int rc;
  if (dtb == CMH_FP8) {
    for (const auto* pr : {&a, &b}) {
      const cmh_block_weights& w = pr == &a ? wa : wb;
      const float* s = w.act_scale;
      CMH_CHECK_ARG(pr->t.xh, "fp8 mode runs on the fp16 residual stream (width %% 256 == 0, no taps)");
      CMH_CHECK_ARG(w.in_proj_cs && w.out_proj_cs && w.fc_cs && w.proj_cs, "fp8 mode: weight scales missing");
      CMH_CHECK_ARG(s[0] > 0.f && s[1] > 0.f && s[2] > 0.f && s[3] > 0.f, "fp8 mode: activation scales missing (run the calibration pass)");
    }
    const float *sa = wa.act_scale, *sb = wb.act_scale;
    const int rx = EPI_BIAS | EPI_RESIDUAL | EPI_RES_F16 | EPI_OUT_F16;
    if ((rc = launch_layernorm_q(a.t.x, wa.ln1_w, wa.ln1_b, a.t.h, 1.0f / sa[0], a.M, a.d, st, a.md))) return rc;
    if ((rc = launch_layernorm_q(b.t.x, wb.ln1_w, wb.ln1_b, b.t.h, 1.0f / sb[0], b.M, b.d, st, b.md))) return rc;
    if ((rc = launch_gemm_grouped(CMH_FP8, problem_of(a, a.t.h, wa.in_proj_w, wa.in_proj_b, nullptr, a.t.qkv, 3 * a.d, a.d, wa.in_proj_cs, sa[0]),
                                  problem_of(b, b.t.h, wb.in_proj_w, wb.in_proj_b, nullptr, b.t.qkv, 3 * b.d, b.d, wb.in_proj_cs, sb[0]),
                                  EPI_BIAS | EPI_OUT_BF16, st))) return rc;
    // (the two attentions stay two launches: one launch for both - each side's body compiled for its own key-tile count - runs every
    // wave at the wider side's register budget, 2 waves per SIMD instead of 3 for the image side: 36.3 us against 16.6 + 16.1,
    // profiles/r04_j_bench_kernel_stats.csv; removed again)
    if ((rc = launch_attention_varlen(a.t.qkv, a.t.h, CMH_BF16, a.B, a.T, a.d, a.causal, nullptr, a.seq_off, st, 1.0f / sa[1]))) return rc;
    if ((rc = launch_attention_varlen(b.t.qkv, b.t.h, CMH_BF16, b.B, b.T, b.d, b.causal, nullptr, b.seq_off, st, 1.0f / sb[1]))) return rc;
    if (upto_attention) return CMH_OK;
    if ((rc = launch_gemm_grouped(CMH_FP8, problem_of(a, a.t.h, wa.out_proj_w, wa.out_proj_b, a.t.x, a.t.x, a.d, a.d, wa.out_proj_cs, sa[1]),
                                  problem_of(b, b.t.h, wb.out_proj_w, wb.out_proj_b, b.t.x, b.t.x, b.d, b.d, wb.out_proj_cs, sb[1]), rx, st))) return rc;
    if ((rc = launch_layernorm_q(a.t.x, wa.ln2_w, wa.ln2_b, a.t.h, 1.0f / sa[2], a.M, a.d, st, a.md))) return rc;
    if ((rc = launch_layernorm_q(b.t.x, wb.ln2_w, wb.ln2_b, b.t.h, 1.0f / sb[2], b.M, b.d, st, b.md))) return rc;
    if ((rc = launch_gemm_grouped(CMH_FP8, problem_of(a, a.t.h, wa.fc_w, wa.fc_b, nullptr, a.t.mlp, 4 * a.d, a.d, wa.fc_cs, sa[2], 1.0f / sa[3]),
                                  problem_of(b, b.t.h, wb.fc_w, wb.fc_b, nullptr, b.t.mlp, 4 * b.d, b.d, wb.fc_cs, sb[2], 1.0f / sb[3]),
                                  EPI_BIAS | EPI_QUICKGELU | EPI_OUT_FP8, st))) return rc;
    return launch_gemm_grouped(CMH_FP8, problem_of(a, a.t.mlp, wa.proj_w, wa.proj_b, a.t.x, a.t.x, a.d, 4 * a.d, wa.proj_cs, sa[3]),
                               problem_of(b, b.t.mlp, wb.proj_w, wb.proj_b, b.t.x, b.t.x, b.d, 4 * b.d, wb.proj_cs, sb[3]), rx, st);
  }
  const int dt = dtb;
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  CMH_CHECK_ARG(a.t.xh == b.t.xh, "run_block_pair: the towers' residual streams differ in kind (the caller runs such towers one by one)");
  const int rx = EPI_BIAS | EPI_RESIDUAL | (a.t.xh ? EPI_RES_F16 | EPI_OUT_F16 : 0);
  const bool ln_pair = dt == CMH_BF16 && a.t.xh && b.t.xh;     // fp16 stream -> bf16 rows: both towers' rows in one launch
  rc = ln_pair ? launch_layernorm_h2b_pair(a.t.x, wa.ln1_w, wa.ln1_b, a.t.h, a.M, a.d, a.md, b.t.x, wb.ln1_w, wb.ln1_b, b.t.h, b.M, b.d, b.md, st) : 1;
  if (rc < 0) return rc;
  if (rc > 0) {
    if ((rc = launch_layernorm_x(a.t.x, a.t.xh, nullptr, wa.ln1_w, wa.ln1_b, a.t.h, dt == CMH_BF16, a.M, a.d, st, a.md))) return rc;
    if ((rc = launch_layernorm_x(b.t.x, b.t.xh, nullptr, wb.ln1_w, wb.ln1_b, b.t.h, dt == CMH_BF16, b.M, b.d, st, b.md))) return rc;
  }
  if ((rc = launch_gemm_grouped(dt, problem_of(a, a.t.h, wa.in_proj_w, wa.in_proj_b, nullptr, a.t.qkv, 3 * a.d, a.d),
                                problem_of(b, b.t.h, wb.in_proj_w, wb.in_proj_b, nullptr, b.t.qkv, 3 * b.d, b.d), EPI_BIAS | obf, st))) return rc;
  if ((rc = launch_attention_varlen(a.t.qkv, a.t.h, dt, a.B, a.T, a.d, a.causal, nullptr, a.seq_off, st))) return rc;
  if ((rc = launch_attention_varlen(b.t.qkv, b.t.h, dt, b.B, b.T, b.d, b.causal, nullptr, b.seq_off, st))) return rc;
  if (upto_attention) return CMH_OK;
  if ((rc = launch_gemm_grouped(dt, problem_of(a, a.t.h, wa.out_proj_w, wa.out_proj_b, a.t.x, a.t.x, a.d, a.d),
                                problem_of(b, b.t.h, wb.out_proj_w, wb.out_proj_b, b.t.x, b.t.x, b.d, b.d), rx, st))) return rc;
  rc = ln_pair ? launch_layernorm_h2b_pair(a.t.x, wa.ln2_w, wa.ln2_b, a.t.h, a.M, a.d, a.md, b.t.x, wb.ln2_w, wb.ln2_b, b.t.h, b.M, b.d, b.md, st) : 1;
  if (rc < 0) return rc;
  if (rc > 0) {
    if ((rc = launch_layernorm_x(a.t.x, a.t.xh, nullptr, wa.ln2_w, wa.ln2_b, a.t.h, dt == CMH_BF16, a.M, a.d, st, a.md))) return rc;
    if ((rc = launch_layernorm_x(b.t.x, b.t.xh, nullptr, wb.ln2_w, wb.ln2_b, b.t.h, dt == CMH_BF16, b.M, b.d, st, b.md))) return rc;
  }
  if ((rc = launch_gemm_grouped(dt, problem_of(a, a.t.h, wa.fc_w, wa.fc_b, nullptr, a.t.mlp, 4 * a.d, a.d),
                                problem_of(b, b.t.h, wb.fc_w, wb.fc_b, nullptr, b.t.mlp, 4 * b.d, b.d), EPI_BIAS | EPI_QUICKGELU | obf, st))) return rc;
  return launch_gemm_grouped(dt, problem_of(a, a.t.mlp, wa.proj_w, wa.proj_b, a.t.x, a.t.x, a.d, 4 * a.d),
                             problem_of(b, b.t.mlp, wb.proj_w, wb.proj_b, b.t.x, b.t.x, b.d, 4 * b.d), rx, st);
}

// The packed row count of the LAST finished call for a (batch, seq_len), as a hint for the next call's tile heights: every call
// queues an asynchronous copy of its count into a pinned host word and records an event; the next call takes the value if that
// event has completed (hipEventQuery never blocks) and keeps its previous hint otherwise.  Results never depend on the hint.
namespace {
struct RowsHint { int B = 0, L = 0, hint = -1; int32_t* pinned = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
std::mutex g_hint_mu;
RowsHint g_hint;
}  // namespace
static int rows_hint_exchange(const int32_t* count_dev, int B, int L, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_hint_mu);
  RowsHint& h = g_hint;
  if (!h.pinned) {
    if (hipHostMalloc(reinterpret_cast<void**>(&h.pinned), 64, hipHostMallocDefault) != hipSuccess) { h.pinned = nullptr; return -1; }
    if (hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) { h.ev = nullptr; return -1; }
  }
  if (!h.ev) return -1;
  if (h.pending && hipEventQuery(h.ev) == hipSuccess) {
    h.pending = false;
    if (h.B == B && h.L == L && *h.pinned > 0 && *h.pinned <= B * L) h.hint = *h.pinned;
  }
  const int out = (h.B == B && h.L == L) ? h.hint : -1;
  if (!h.pending) {          // one copy in flight at a time: the pinned word is not rewritten under a reader
    if (h.B != B || h.L != L) { h.B = B; h.L = L; h.hint = -1; }
    if (hipMemcpyAsync(h.pinned, count_dev, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(h.ev, st) == hipSuccess) h.pending = true;
  }
  return out;
}

static int final_projection(int dt, const void* pool, const void* w_t, float* feat, int B, int embed, int d,
                            hipStream_t st) {
  const int bk = dt == CMH_F32 ? 32 : 64;
  if (embed % 128 == 0 && d % bk == 0) return launch_gemm(dt, pool, w_t, nullptr, nullptr, feat, B, embed, d, 0, st);
  return launch_small_linear(dt, pool, w_t, nullptr, nullptr, 1.f, CMH_ACT_NONE, feat, B, embed, d, st);
}

// fp16 residual stream?  (bf16 mode only; the residual GEMMs must take the wide kernel; taps are defined on an f32 stream)
static int resid_f16(int dt, int d, const cmh_taps* taps) {
  static const bool off = []() { const char* e = getenv("CMH_RESID_F16"); return e && !strcmp(e, "0"); }();
  if (dt == CMH_FP8) return d % 256 == 0 && !taps;
  return dt == CMH_BF16 && d % 256 == 0 && !taps && !off;
}

static int check_tower(int dt, int width, int layers, int embed, const cmh_block_weights* blocks) {
  CMH_CHECK_ARG(dt == CMH_F32 || dt == CMH_BF16 || dt == CMH_FP8, "bad gemm_dtype %d", dt);
  CMH_CHECK_ARG(width > 0 && width % 128 == 0 && width <= 1024, "width %d must be a multiple of 128, <= 1024", width);
  CMH_CHECK_ARG(dt != CMH_FP8 || width % 256 == 0, "fp8 mode needs width %% 256 == 0 (width %d)", width);
  CMH_CHECK_ARG(layers >= 0 && embed > 0 && embed % 4 == 0, "bad layers/embed_dim");
  CMH_CHECK_ARG(layers == 0 || blocks, "blocks is null");
  return CMH_OK;
}

}  // namespace cmh

using namespace cmh;

extern "C" const char* cmh_last_error(void) { return err_buf(); }
extern "C" int cmh_set_pooled_tail(int32_t on) { g_pooled_tail = on ? 1 : 0; return CMH_OK; }
extern "C" int cmh_set_text_token_packing(int32_t on) {
  CMH_CHECK_ARG(on >= -1 && on <= 1, "set_text_token_packing: %d (-1 environment, 0 off, 1 on)", on);
  g_pack_tokens = on;
  return CMH_OK;
}
extern "C" int cmh_version(void) { return CMH_VERSION; }

extern "C" size_t cmh_vit_workspace_bytes(const cmh_vit_weights* w, int32_t batch) {
  if (!w || batch <= 0 || w->patch <= 0) return 0;
  const size_t g = w->resolution / w->patch, g2 = g * g, T = g2 + 1, d = w->width;
  const size_t e = w->gemm_dtype == CMH_F32 ? 4 : 2;   // fp8 mode: sized like bf16 (conv1 and the stream are the bf16 mode's)
  const size_t B = batch, pk = 3ull * w->patch * w->patch;
  return carve(nullptr, B * T, B, d, e, B * g2 * d * 4, B * g2 * pk * e).total;
}

// validation + everything before the first block: conv1 as a patch-matrix GEMM, [class ; patches] + positional, ln_pre
// image_b / batch_a (cmh_clip_encode_pair2): the batch's images come as TWO tensors - rows [0, batch_a) from `image`, the rest from
// image_b - patchified into consecutive rows of the one patch matrix; nothing behind that knows.
static int vit_begin(const cmh_vit_weights* w, const float* image, int32_t batch, bool want_out, void* workspace, size_t workspace_bytes,
                     const cmh_taps* taps, hipStream_t st, float* amax, TowerRun& r, const float* image_b = nullptr, int32_t batch_a = 0) {
  CMH_CHECK_ARG(w && image && want_out && workspace, "vit_encode: null pointer");
  CMH_CHECK_ARG(batch > 0, "vit_encode: batch %d", batch);
  int rc = check_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(w->patch > 0 && w->resolution % w->patch == 0 && w->patch % 4 == 0, "vit_encode: resolution %d / patch %d",
                w->resolution, w->patch);
  const int dtb = w->gemm_dtype, d = w->width, B = batch;   // dtb: arithmetic of the blocks' GEMMs
  const int dt = dtb == CMH_FP8 ? CMH_BF16 : dtb;             // everything outside the blocks (conv1, LayerNorms, projections)
  const int g = w->resolution / w->patch, g2 = g * g, T = g2 + 1, M = B * T;
  const int pk = 3 * w->patch * w->patch;
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  CMH_CHECK_ARG(!amax || dtb == CMH_BF16, "vit_calibrate_fp8: weights must be the bf16 mode's");
  CMH_CHECK_ARG(pk % (dt == CMH_F32 ? 32 : 64) == 0, "vit_encode: 3*patch^2 = %d not a multiple of the GEMM K-step", pk);
  const size_t need = cmh_vit_workspace_bytes(w, batch);
  if (workspace_bytes < need) return fail(CMH_ERR_WORKSPACE, "vit_encode: workspace %zu < %zu bytes", workspace_bytes, need);
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "vit_encode: workspace must be 256-byte aligned");
  TowerBufs& t = r.t;
  t = carve(workspace, static_cast<size_t>(M), B, d, e, static_cast<size_t>(B) * g2 * d * 4, static_cast<size_t>(B) * g2 * pk * e);
  t.xh = resid_f16(dtb, d, taps);
  r.dtb = dtb; r.d = d; r.B = B; r.T = T; r.M = M; r.causal = 0;
  void* patches = t.mlp;
  float* patch_out = static_cast<float*>(t.qkv);

  // conv1 (kernel = stride = patch, no bias) as patch-matrix GEMM  (model.py:215,231-235)
  if (image_b) {
    CMH_CHECK_ARG(batch_a > 0 && batch_a < B, "vit_encode: split batch %d of %d", batch_a, B);
    if ((rc = launch_patchify(image, patches, dt, batch_a, w->resolution, w->patch, st))) return rc;
    if ((rc = launch_patchify(image_b, static_cast<char*>(patches) + static_cast<size_t>(batch_a) * g2 * pk * e, dt, B - batch_a, w->resolution,
                              w->patch, st))) return rc;
  } else if ((rc = launch_patchify(image, patches, dt, B, w->resolution, w->patch, st))) return rc;
  if ((rc = launch_gemm(dt, patches, w->conv1_w, nullptr, nullptr, patch_out, B * g2, d, pk, 0, st))) return rc;
  // [class ; patches] + positional, ln_pre  (:237-239)
  if ((rc = launch_vit_assemble_lnpre(patch_out, w->class_embedding, w->positional_embedding, w->ln_pre_w,
                                      w->ln_pre_b, t.x, t.xh, B, g2, d, st))) return rc;
  return tap(taps, 0, t.x, static_cast<size_t>(M) * d * 4, st);
}

// ln_post (+ proj) after the last block: on every token (MITH trunk) and / or on the class token
static int vit_finish(const cmh_vit_weights* w, const TowerRun& r, void* x_pooled, float* feat, float* tokens_out, hipStream_t st) {
  const TowerBufs& t = r.t;
  const int dt = r.dtb == CMH_FP8 ? CMH_BF16 : r.dtb, d = r.d, B = r.B, T = r.T, M = r.M;
  int rc;
  if (tokens_out) {
    // MITH trunk (model/MITH.py:70-80): ln_post and proj on EVERY token
    if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w->ln_post_w, w->ln_post_b, t.h, dt == CMH_BF16, M, d, st))) return rc;
    if ((rc = final_projection(dt, t.h, w->proj_t, tokens_out, M, w->embed_dim, d, st))) return rc;
  }
  if (feat) {
    // ln_post on the class token, @ proj  (:247-250)
    if (x_pooled) {
      if ((rc = launch_layernorm_x(x_pooled, t.xh, nullptr, w->ln_post_w, w->ln_post_b, t.pool, dt == CMH_BF16, B, d, st))) return rc;
    } else {
      if ((rc = launch_iota_rows(t.rows, B, T, st))) return rc;
      if ((rc = launch_layernorm_x(t.x, t.xh, t.rows, w->ln_post_w, w->ln_post_b, t.pool, dt == CMH_BF16, B, d, st))) return rc;
    }
    if ((rc = final_projection(dt, t.pool, w->proj_t, feat, B, w->embed_dim, d, st))) return rc;
  }
  return CMH_OK;
}

static int vit_encode_impl(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, float* tokens_out,
                           void* workspace, size_t workspace_bytes, const cmh_taps* taps, void* stream, float* amax = nullptr) {
  hipStream_t st = as_stream(stream);
  TowerRun r;
  int rc = vit_begin(w, image, batch, feat || tokens_out, workspace, workspace_bytes, taps, st, amax, r);
  if (rc) return rc;
  const TowerBufs& t = r.t;
  const int dtb = r.dtb, d = r.d, B = r.B, T = r.T, M = r.M;
  const bool tail = feat && !tokens_out && !taps && !amax && w->layers > 0 && pooled_tail_enabled();
  void* x_pooled = nullptr;
  if (tail && (rc = launch_iota_rows(t.rows, B, T, st))) return rc;
  for (int i = 0; i < w->layers; ++i) {
    if (tail && i == w->layers - 1) {
      if ((rc = run_block_pooled(w->blocks[i], dtb, t, B, T, d, /*causal=*/0, nullptr, st, M, nullptr, &x_pooled))) return rc;
      break;
    }
    if ((rc = run_block(w->blocks[i], dtb, t, B, T, d, /*causal=*/0, nullptr, st, -1, nullptr, amax ? amax + 4 * i : nullptr))) return rc;
    if ((rc = tap(taps, 1 + i, t.x, static_cast<size_t>(M) * d * 4, st))) return rc;
  }
  return vit_finish(w, r, x_pooled, feat, tokens_out, st);
}

extern "C" int cmh_vit_encode(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat,
                              void* workspace, size_t workspace_bytes, const cmh_taps* taps, void* stream) {
  CMH_CHECK_ARG(feat, "vit_encode: null pointer");
  return vit_encode_impl(w, image, batch, feat, nullptr, workspace, workspace_bytes, taps, stream);
}

extern "C" int cmh_vit_encode_tokens(const cmh_vit_weights* w, const float* image, int32_t batch, float* tokens_out,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "vit_encode_tokens: null pointer");
  return vit_encode_impl(w, image, batch, nullptr, tokens_out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" size_t cmh_text_workspace_bytes(const cmh_text_weights* w, int32_t batch, int32_t seq_len) {
  if (!w || batch <= 0 || seq_len <= 0) return 0;
  const size_t e = w->gemm_dtype == CMH_F32 ? 4 : 2;
  return carve(nullptr, static_cast<size_t>(batch) * seq_len, batch, w->width, e, 0, 0).total;
}

// validation + everything before the first block: the pack plan (packed mode), token + positional embedding, the EOT rows
static int text_begin(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len, const uint8_t* key_padding_mask,
                      bool want_out, bool tokens_wanted, void* workspace, size_t workspace_bytes, const cmh_taps* taps, hipStream_t st,
                      int32_t* packed_rows_out, float* amax, TowerRun& r, bool pack_tokens_req = false) {
  CMH_CHECK_ARG(w && tokens && want_out && workspace, "text_encode: null pointer");
  CMH_CHECK_ARG(batch > 0 && seq_len > 0, "text_encode: batch %d seq_len %d", batch, seq_len);
  int rc = check_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(seq_len <= w->context_length, "text_encode: seq_len %d > context_length %d", seq_len, w->context_length);
  const int dtb = w->gemm_dtype, d = w->width, B = batch, L = seq_len, M = B * L;
  const int dt = dtb == CMH_FP8 ? CMH_BF16 : dtb;
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  CMH_CHECK_ARG(!amax || dtb == CMH_BF16, "text_calibrate_fp8: weights must be the bf16 mode's");
  const size_t need = cmh_text_workspace_bytes(w, batch, seq_len);
  if (workspace_bytes < need) return fail(CMH_ERR_WORKSPACE, "text_encode: workspace %zu < %zu bytes", workspace_bytes, need);
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "text_encode: workspace must be 256-byte aligned");
  TowerBufs& t = r.t;
  t = carve(workspace, static_cast<size_t>(M), B, d, e, 0, 0);
  t.xh = resid_f16(dtb, d, taps);

  // Packed mode (pooled output only): under the causal mask nothing after a caption's EOT can reach the EOT row that
  // encode_text returns (model.py:366-370), so only the tokens 0..EOT of every caption are embedded and run through the
  // blocks - rows [seq_off[b], seq_off[b+1]) of one packed matrix; every kept row goes through exactly the arithmetic of
  // the dense path (row-wise kernels, per-row dot products, attention over the same key tiles), so the features are
  // bit-identical.  The row count is read back once (the GEMM grids need it on the host).
  const int32_t* seq_off = nullptr;
  const int32_t* md = nullptr;      // device-side row count of the packed matrix (the kernels read it themselves)
  int rows = M, mh = -1;
  // Round 5: the all-token trunk of MITH (model/MITH.py:120-144 returns every position; HashingModel gives the padded ones weight 0
  // in LocalizedTokenAggregation, :349-376, and reads them nowhere else) is packed too: a caption's rows run to its last unpadded
  // position, the projected tokens go back to their dense [B, L, E] places with zeros behind (text_finish).  Kept rows see the dense
  // path's arithmetic (the mask is still applied to the keys inside the kept prefix): same bits there.
  // (asked for per call - cmh_text_encode_tokens_packed: the caller promises not to read the padded positions - and only then)
  const bool pack_tokens = pack_tokens_req && tokens_wanted && key_padding_mask && !packed_rows_out && !taps && !amax && text_token_packing() &&
                           w->embed_dim % 128 == 0 && d % (dt == CMH_F32 ? 32 : 64) == 0 &&      // the packed projection is a GEMM launch
                           static_cast<size_t>(w->embed_dim) * 4 <= static_cast<size_t>(4) * d * e;   // ... into the MLP scratch
  if (pack_tokens) packed_rows_out = reinterpret_cast<int32_t*>(1);
  r.packed_tokens = pack_tokens;
  if (packed_rows_out) {
    CMH_CHECK_ARG(pack_tokens || (!key_padding_mask && !tokens_wanted && !taps), "text_encode_packed: pooled features only, no mask / taps");
    if ((rc = launch_text_pack_plan(tokens, B, L, t.seq, st, pack_tokens ? key_padding_mask : nullptr, pack_tokens ? t.rows : nullptr))) return rc;
    seq_off = t.seq;
    if (amax || d % 256 != 0 || !gemm_wide_enabled()) {
      // the calibration pass reduces over whole buffers on the host's row count, and widths that are not a multiple of 256 (the
      // test-sized towers) - or any width under the CMH_GEMM_WIDE=0 diagnostic - run on the 128 x 128 fallback GEMMs, which take
      // their row count from the host: these alone read the count back (one synchronisation)
      int32_t total = 0;
      if (hipMemcpyAsync(&total, t.seq + B, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(CMH_ERR_LAUNCH, "text_encode_packed: reading the packed row count failed");
      CMH_CHECK_ARG(total > 0 && total <= M, "text_encode_packed: bad packed row count %d", total);
      rows = total;
    } else {
      // LayerNorm and the GEMMs take M = B*L as an upper bound and read the real count from seq[B]; the tile height is chosen for
      // the count of an earlier call (rows_hint: captions of one dataset are alike), never waited for
      md = t.seq + B;
      mh = rows_hint_exchange(t.seq + B, B, L, st);
    }
    if (packed_rows_out != reinterpret_cast<int32_t*>(1) &&
        hipMemcpyAsync(packed_rows_out, t.seq + B, 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "text_encode_packed: copying the row count failed");
  }

  // token_embedding gather + positional_embedding[:L]; EOT row = argmax(tokens)  (model.py:360-362,370)
  r.dtb = dtb; r.d = d; r.B = B; r.T = L; r.M = rows; r.causal = 1; r.seq_off = seq_off; r.md = md; r.mh = mh;
  return launch_text_embed_packed(tokens, w->token_embedding, w->positional_embedding, t.x, t.xh, t.rows, B, L, d, w->vocab_size, seq_off, st,
                                  pack_tokens);
}

// ln_final (+ text_projection) after the last block: on every token (MITH trunk) and / or on the EOT rows
static int text_finish(const cmh_text_weights* w, const TowerRun& r, void* x_pooled, float* feat, float* tokens_out, int32_t* eot_rows_out,
                       hipStream_t st) {
  const TowerBufs& t = r.t;
  const int dt = r.dtb == CMH_FP8 ? CMH_BF16 : r.dtb, d = r.d, B = r.B, M = r.B * r.T;
  int rc;
  if (tokens_out && r.packed_tokens) {
    // the kept rows only (device row count), then back to their dense places; the EOT rows leave as dense indices with them
    float* tmp = static_cast<float*>(t.mlp);      // [rows, E] f32: the MLP scratch is free behind the last block (text_begin checked the sizes)
    if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w->ln_final_w, w->ln_final_b, t.h, dt == CMH_BF16, r.M, d, st, r.md))) return rc;
    if ((rc = launch_gemm(dt, t.h, w->text_projection_t, nullptr, nullptr, tmp, r.M, w->embed_dim, d, 0, st, r.md, r.mh))) return rc;
    if ((rc = launch_unpack_token_rows(tmp, r.seq_off, tokens_out, B, r.T, w->embed_dim, t.rows, eot_rows_out, st))) return rc;
  } else if (tokens_out) {
    // MITH trunk (model/MITH.py:136-139): ln_final and text_projection on EVERY token
    if ((rc = launch_layernorm_x(t.x, t.xh, nullptr, w->ln_final_w, w->ln_final_b, t.h, dt == CMH_BF16, M, d, st))) return rc;
    if ((rc = final_projection(dt, t.h, w->text_projection_t, tokens_out, M, w->embed_dim, d, st))) return rc;
  }
  if (eot_rows_out && !(tokens_out && r.packed_tokens) &&
      hipMemcpyAsync(eot_rows_out, t.rows, static_cast<size_t>(B) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "text_encode: eot row copy failed");
  if (feat) {
    // ln_final (row-wise, so only the pooled rows are normalised), @ text_projection  (:366-370)
    if ((rc = launch_layernorm_x(x_pooled ? x_pooled : t.x, t.xh, x_pooled ? nullptr : t.rows, w->ln_final_w, w->ln_final_b, t.pool,
                                 dt == CMH_BF16, B, d, st))) return rc;
    if ((rc = final_projection(dt, t.pool, w->text_projection_t, feat, B, w->embed_dim, d, st))) return rc;
  }
  return CMH_OK;
}

static int text_encode_impl(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                            const uint8_t* key_padding_mask, float* feat, float* tokens_out, int32_t* eot_rows_out,
                            void* workspace, size_t workspace_bytes, const cmh_taps* taps, void* stream,
                            int32_t* packed_rows_out = nullptr, float* amax = nullptr, bool pack_tokens_req = false) {
  hipStream_t st = as_stream(stream);
  TowerRun r;
  int rc = text_begin(w, tokens, batch, seq_len, key_padding_mask, feat || tokens_out, tokens_out != nullptr, workspace, workspace_bytes,
                      taps, st, packed_rows_out, amax, r, pack_tokens_req);
  if (rc) return rc;
  const TowerBufs& t = r.t;
  const int dtb = r.dtb, d = r.d, B = r.B, L = r.T, M = B * L, rows = r.M, mh = r.mh;
  const int32_t* seq_off = r.seq_off;
  const int32_t* md = r.md;
  const bool tail = feat && !tokens_out && !taps && !amax && !eot_rows_out && w->layers > 0 && pooled_tail_enabled();
  void* x_pooled = nullptr;
  for (int i = 0; i < w->layers; ++i) {
    if (tail && i == w->layers - 1) {
      if ((rc = run_block_pooled(w->blocks[i], dtb, t, B, L, d, /*causal=*/1, key_padding_mask, st, rows, seq_off, &x_pooled, md, mh))) return rc;
      break;
    }
    if ((rc = run_block(w->blocks[i], dtb, t, B, L, d, /*causal=*/1, key_padding_mask, st, rows, seq_off, amax ? amax + 4 * i : nullptr, md, mh))) return rc;
    if ((rc = tap(taps, 1 + i, t.x, static_cast<size_t>(M) * d * 4, st))) return rc;
  }
  return text_finish(w, r, x_pooled, feat, tokens_out, eot_rows_out, st);
}

extern "C" int cmh_text_encode(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                               const uint8_t* key_padding_mask, float* feat, void* workspace,
                               size_t workspace_bytes, const cmh_taps* taps, void* stream) {
  CMH_CHECK_ARG(feat, "text_encode: null pointer");
  return text_encode_impl(w, tokens, batch, seq_len, key_padding_mask, feat, nullptr, nullptr, workspace,
                          workspace_bytes, taps, stream);
}

extern "C" int cmh_text_encode_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len, float* feat,
                                      int32_t* rows_computed_dev, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(feat, "text_encode_packed: null pointer");
  return text_encode_impl(w, tokens, batch, seq_len, nullptr, feat, nullptr, nullptr, workspace, workspace_bytes, nullptr, stream,
                          rows_computed_dev ? rows_computed_dev : reinterpret_cast<int32_t*>(1));   // 1: packed, count not wanted
}

// encode_image + encode_text of one batch with the two towers in lock-step (reference model/modelbase.py:105-108 runs them back to
// back; model/base/model.py:340-372): layer i of both towers shares its launches (run_block_pair).  Same features, bit for bit, as
// cmh_vit_encode + cmh_text_encode[_packed].
static int clip_encode_pair_impl(const cmh_vit_weights* vw, const float* image, const float* image_b, int32_t batch_a,
                                 const cmh_text_weights* tw, const int64_t* tokens,
                                 int32_t batch, int32_t seq_len, int32_t packed, float* feat_image, float* feat_text,
                                 int32_t* rows_computed_dev, void* ws_image, size_t ws_image_bytes, void* ws_text,
                                 size_t ws_text_bytes, void* stream) {
  CMH_CHECK_ARG(vw && tw && feat_image && feat_text, "clip_encode_pair: null pointer");
  CMH_CHECK_ARG(vw->gemm_dtype == tw->gemm_dtype, "clip_encode_pair: both towers must run in one arithmetic mode (%d / %d)", vw->gemm_dtype,
                tw->gemm_dtype);
  hipStream_t st = as_stream(stream);
  TowerRun a, b;
  int rc;
  if ((rc = vit_begin(vw, image, batch, true, ws_image, ws_image_bytes, nullptr, st, nullptr, a, image_b, batch_a))) return rc;
  if ((rc = text_begin(tw, tokens, batch, seq_len, nullptr, true, false, ws_text, ws_text_bytes, nullptr, st,
                       packed ? (rows_computed_dev ? rows_computed_dev : reinterpret_cast<int32_t*>(1)) : nullptr, nullptr, b))) return rc;
  const bool tail = pooled_tail_enabled();
  void *xa = nullptr, *xb = nullptr;
  if (tail && vw->layers > 0 && (rc = launch_iota_rows(a.t.rows, a.B, a.T, st))) return rc;
  const int deepest = vw->layers > tw->layers ? vw->layers : tw->layers;
  for (int i = 0; i < deepest; ++i) {
    const bool has_a = i < vw->layers, has_b = i < tw->layers;
    const bool last_a = tail && i == vw->layers - 1, last_b = tail && i == tw->layers - 1;
    // lock-step only when both towers carry the same kind of residual stream (fp16 for widths that are multiples of 256 in the bf16
    // mode, else f32: resid_f16 decides per tower): a grouped launch has ONE set of epilogue flags
    if (has_a && has_b && last_a == last_b && a.t.xh == b.t.xh) {
      if ((rc = run_block_pair(vw->blocks[i], tw->blocks[i], a, b, st, last_a))) return rc;
      if (last_a) {       // the rest of the last block on the pooled rows of each tower (few-row kernels)
        if ((rc = run_block_pooled(vw->blocks[i], a.dtb, a.t, a.B, a.T, a.d, 0, nullptr, st, a.M, nullptr, &xa, nullptr, -1, true))) return rc;
        if ((rc = run_block_pooled(tw->blocks[i], b.dtb, b.t, b.B, b.T, b.d, 1, nullptr, st, b.M, b.seq_off, &xb, b.md, b.mh, true))) return rc;
      }
      continue;
    }
    if (has_a) {
      if (last_a) { if ((rc = run_block_pooled(vw->blocks[i], a.dtb, a.t, a.B, a.T, a.d, 0, nullptr, st, a.M, nullptr, &xa))) return rc; }
      else if ((rc = run_block(vw->blocks[i], a.dtb, a.t, a.B, a.T, a.d, 0, nullptr, st))) return rc;
    }
    if (has_b) {
      if (last_b) { if ((rc = run_block_pooled(tw->blocks[i], b.dtb, b.t, b.B, b.T, b.d, 1, nullptr, st, b.M, b.seq_off, &xb, b.md, b.mh))) return rc; }
      else if ((rc = run_block(tw->blocks[i], b.dtb, b.t, b.B, b.T, b.d, 1, nullptr, st, b.M, b.seq_off, nullptr, b.md, b.mh))) return rc;
    }
  }
  if ((rc = vit_finish(vw, a, xa, feat_image, nullptr, st))) return rc;
  return text_finish(tw, b, xb, feat_text, nullptr, nullptr, st);
}

extern "C" int cmh_clip_encode_pair(const cmh_vit_weights* vw, const float* image, const cmh_text_weights* tw, const int64_t* tokens,
                                    int32_t batch, int32_t seq_len, int32_t packed, float* feat_image, float* feat_text,
                                    int32_t* rows_computed_dev, void* ws_image, size_t ws_image_bytes, void* ws_text,
                                    size_t ws_text_bytes, void* stream) {
  return clip_encode_pair_impl(vw, image, nullptr, 0, tw, tokens, batch, seq_len, packed, feat_image, feat_text, rows_computed_dev, ws_image,
                               ws_image_bytes, ws_text, ws_text_bytes, stream);
}

// TWO loader batches as one: images as two tensors (batch_a + batch_b rows), the captions of both as one [batch_a + batch_b, seq_len]
// matrix.  Every row sees the arithmetic of cmh_clip_encode_pair on its own batch: the same features, half as many launches per pair.
extern "C" int cmh_clip_encode_pair2(const cmh_vit_weights* vw, const float* image_a, int32_t batch_a, const float* image_b, int32_t batch_b,
                                     const cmh_text_weights* tw, const int64_t* tokens, int32_t seq_len, int32_t packed,
                                     float* feat_image, float* feat_text, int32_t* rows_computed_dev, void* ws_image,
                                     size_t ws_image_bytes, void* ws_text, size_t ws_text_bytes, void* stream) {
  CMH_CHECK_ARG(image_a && image_b && batch_a > 0 && batch_b > 0, "clip_encode_pair2: two non-empty batches");
  return clip_encode_pair_impl(vw, image_a, image_b, batch_a, tw, tokens, batch_a + batch_b, seq_len, packed, feat_image, feat_text,
                               rows_computed_dev, ws_image, ws_image_bytes, ws_text, ws_text_bytes, stream);
}

extern "C" int cmh_linear_gemm_grouped(int32_t dtype, const cmh_gemm_problem* pa, const cmh_gemm_problem* pb, int32_t epilogue, void* stream) {
  CMH_CHECK_ARG(pa && pb, "linear_gemm_grouped: null pointer");
  CMH_CHECK_ARG(dtype == CMH_F32 || dtype == CMH_BF16 || dtype == CMH_FP8, "linear_gemm_grouped: bad dtype %d", dtype);
  auto conv = [](const cmh_gemm_problem* g) {
    return GemmProblem{g->x, g->w, g->bias, static_cast<const float*>(g->residual), g->out, g->M, g->N, g->K, g->m_dev, -1, g->colscale,
                       g->alpha, g->out_scale > 0.f ? 1.0f / g->out_scale : 1.0f};      // (out = e4m3(v / out_scale), as cmh_linear_gemm_fp8)
  };
  for (const cmh_gemm_problem* g : {pa, pb})
    CMH_CHECK_ARG(g->x && g->w && g->out && g->M > 0 && g->N > 0 && g->K > 0, "linear_gemm_grouped: null pointer / empty problem");
  return launch_gemm_grouped(dtype, conv(pa), conv(pb), epilogue, as_stream(stream));
}

extern "C" int cmh_vit_calibrate_fp8(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, float* amax,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(feat && amax, "vit_calibrate_fp8: null pointer");
  return vit_encode_impl(w, image, batch, feat, nullptr, workspace, workspace_bytes, nullptr, stream, amax);
}

extern "C" int cmh_text_calibrate_fp8(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len, float* feat,
                                      float* amax, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(feat && amax, "text_calibrate_fp8: null pointer");
  // the packed path: calibrate on the rows the fp8 mode will compute
  return text_encode_impl(w, tokens, batch, seq_len, nullptr, feat, nullptr, nullptr, workspace, workspace_bytes, nullptr, stream,
                          reinterpret_cast<int32_t*>(1), amax);
}

extern "C" int cmh_text_encode_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                      const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "text_encode_tokens: null pointer");
  return text_encode_impl(w, tokens, batch, seq_len, key_padding_mask, nullptr, tokens_out, eot_rows_out, workspace,
                          workspace_bytes, nullptr, stream);
}

// The same with the promise that nothing reads the padded positions of tokens_out (MITH: HashingModel masks them, model/MITH.py:349-376):
// positions behind a caption's last unpadded token are not computed and come back as zeros (cmh_set_text_token_packing)
extern "C" int cmh_text_encode_tokens_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                             const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "text_encode_tokens_packed: null pointer");
  return text_encode_impl(w, tokens, batch, seq_len, key_padding_mask, nullptr, tokens_out, eot_rows_out, workspace,
                          workspace_bytes, nullptr, stream, nullptr, nullptr, /*pack_tokens_req=*/true);
}

// A stack of ResidualAttentionBlocks on a caller-owned f32 residual stream x [B*T, d] (in place): the 2-layer
// concept transformer of MITH's LocalConceptTransforming (model/MITH.py:379-396 over model/MITH.py:11-46 blocks).
extern "C" size_t cmh_blocks_workspace_bytes(int32_t dtype, int32_t B, int32_t T, int32_t d) {
  if (B <= 0 || T <= 0 || d <= 0) return 0;
  return carve(nullptr, static_cast<size_t>(B) * T, B, d, dtype == CMH_BF16 ? 2 : 4, 0, 0).total;
}

extern "C" int cmh_transformer_blocks(const cmh_block_weights* blocks, int32_t layers, int32_t dtype, float* x, int32_t B,
                                      int32_t T, int32_t d, int32_t causal, const uint8_t* key_padding_mask,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(blocks && x && workspace && layers > 0 && B > 0 && T > 0, "transformer_blocks: bad arguments");
  CMH_CHECK_ARG(dtype != CMH_FP8, "transformer_blocks: f32 / bf16 only (the fp8 mode runs on the towers' fp16 residual stream)");
  int rc = check_tower(dtype, d, layers, 4, blocks);
  if (rc) return rc;
  if (workspace_bytes < cmh_blocks_workspace_bytes(dtype, B, T, d)) return fail(CMH_ERR_WORKSPACE, "transformer_blocks: workspace too small");
  hipStream_t st = as_stream(stream);
  const size_t M = static_cast<size_t>(B) * T;
  TowerBufs t = carve(workspace, M, B, d, dtype == CMH_BF16 ? 2 : 4, 0, 0);
  if (hipMemcpyAsync(t.x, x, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "transformer_blocks: copy failed");
  for (int i = 0; i < layers; ++i)
    if ((rc = run_block(blocks[i], dtype, t, B, T, d, causal, key_padding_mask, st))) return rc;
  if (hipMemcpyAsync(x, t.x, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "transformer_blocks: copy failed");
  return CMH_OK;
}

extern "C" int cmh_linear_gemm(int32_t dtype, const void* x, const void* w, const float* bias, const float* residual,
                               void* out, int32_t M, int32_t N, int32_t K, int32_t epilogue, void* stream) {
  CMH_CHECK_ARG(x && w && out, "linear_gemm: null pointer");
  return launch_gemm(dtype, x, w, bias, residual, out, M, N, K, epilogue, as_stream(stream));
}

extern "C" int cmh_layernorm(const float* x, const float* w, const float* b, void* out, int32_t out_dtype, int32_t M,
                             int32_t d, void* stream) {
  CMH_CHECK_ARG(x && w && b && out && M > 0, "layernorm: bad arguments");
  return launch_layernorm(x, nullptr, w, b, out, out_dtype == CMH_BF16, M, d, as_stream(stream));
}

extern "C" int cmh_attention(int32_t dtype, const void* qkv, void* o, int32_t B, int32_t T, int32_t d, int32_t causal,
                             const uint8_t* key_padding_mask, void* stream) {
  CMH_CHECK_ARG(qkv && o, "attention: null pointer");
  CMH_CHECK_ARG(dtype == CMH_F32 || dtype == CMH_BF16, "attention: bad dtype");
  return launch_attention(qkv, o, dtype, B, T, d, causal, key_padding_mask, as_stream(stream));
}
