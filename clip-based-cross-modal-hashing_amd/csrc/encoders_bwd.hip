// Training forward (keeps a tape) and backward of the two CLIP towers (SURVEY §8f "next" #2): kernel sequencing only.
//
// Forward with a tape = the forward of encoders.hip with three differences: every block reads its input x_in and writes
// x_mid / the next block's x_in to fresh tape buffers instead of updating one stream in place (the residual GEMMs already
// take residual and output separately); c_fc stores the PRE-activation and QuickGELU runs as its own pass; LayerNorm and
// attention outputs are kept.  Per block the tape holds x_in, h1 = ln_1(x_in), qkv, attn, x_mid, h2 = ln_2(x_mid), pre:
// M*(2*xs*d + 10*e*d) bytes (xs = residual element size, e = GEMM element size): 236 MB per vision block at batch 256, bf16.
//
// Backward of one block, given dx = dL/dx_out (f32 gradient stream, updated in place to dL/dx_in).  With Y = X W^T + b:
//   dX = dY W          -> the forward GEMM kernel on (dY, W^T): W is transposed once per use (<= 4.7 MB);
//   dW = dY^T X        -> the same kernel on (dY^T, X^T): both activations are transposed (and cast to the GEMM dtype) first,
//                         zero-padded along M to the K-step; f32 output straight into the gradient buffer;
//   db = column sums of dY.
//   1. dpre = (dx W_proj) o QuickGELU'(pre)     [epilogue EPI_MUL_DQGELU]     dW_proj = dx^T gelu(pre),  db_proj = sum dx
//   2. dh2  = dpre W_fc                                                       dW_fc = dpre^T h2,        db_fc = sum dpre
//   3. dx  += LayerNorm_bwd(x_mid, dh2) (+ dln_2)
//   4. dattn = dx W_out                                                       dW_out = dx^T attn,       db_out = sum dx
//   5. dqkv = attention_bwd(qkv, attn, dattn)
//   6. dh1  = dqkv W_in                                                       dW_in = dqkv^T h1,        db_in = sum dqkv
//   7. dx  += LayerNorm_bwd(x_in, dh1) (+ dln_1)
// Gradients are WRITTEN (not accumulated) into caller-owned f32 buffers shaped like the reference's parameters.
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cmh_common.h"

namespace cmh {

namespace {

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <typename T = void> T* take(size_t bytes) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += align_up(bytes, 256);
    return p;
  }
};

struct LayerTape { void *x_in, *h1, *qkv, *attn, *x_mid, *h2, *pre, *act; };   // act = QuickGELU(pre): c_proj's operand, kept for its wgrad

struct TrainBufs {
  std::vector<LayerTape> L;
  void* x_last;        // [M,d] xs   output of the last block
  void* patches;       // vision: [B*g2, pk] e
  float* x_pre;        // vision: [M,d] f32 tokens + positional before ln_pre;   text: unused
  float* patch_out;    // vision: [B*g2, d] f32
  int32_t* rows;       // [B] pooled row of every sample
  int32_t* seq_off;    // [B+1] packed text: row offsets of the captions (tokens 0..EOT only)
  void* pool;          // [B,d] e    ln_post / ln_final of the pooled rows
  // scratch
  float* dx;           // [M,d] f32 gradient stream
  float* dx2;          // [M,d] f32
  void* dxe;           // [M,d] e
  void* dpre;          // [M,4d] e
  float* part;         // split-K partial planes of the wgrad GEMMs
  size_t part_bytes;
  void* dh;            // [M,d] e
  void* dqkv;          // [M,3d] e
  void* tA;            // [rmax, Mpad] e   dY^T
  void* tB;            // [rmax, Mpad] e   X^T
  void* wT;            // [max(12 d, rmax) * d] e   W^T of the block's four weight matrices (one launch), or of one head matrix
  void* tokE;          // [M, embed] e
  float* tokP;         // [M, embed] f32   packed all-token head: projected rows before they go to their dense places / packed dtokens
  float* projT;        // [embed, d] f32
  float* small;        // [2*B*max(d,E)] f32
  void* red;           // reduction workspace
  size_t red_bytes;
  size_t tAB_bytes;
  size_t total;
};

size_t pad64(size_t m) { return (m + 63) / 64 * 64; }

TrainBufs carve_train(void* ws, size_t M, size_t B, size_t d, size_t e, size_t xs, int layers, size_t g2rows, size_t pk,
                      size_t embed) {
  Carver a(ws);
  TrainBufs t;
  t.L.resize(layers);
  for (int i = 0; i < layers; ++i) {
    t.L[i].x_in = a.take(M * d * xs);
    t.L[i].h1 = a.take(M * d * e);
    t.L[i].qkv = a.take(M * 3 * d * e);
    t.L[i].attn = a.take(M * d * e);
    t.L[i].x_mid = a.take(M * d * xs);
    t.L[i].h2 = a.take(M * d * e);
    t.L[i].pre = a.take(M * 4 * d * e);
    t.L[i].act = a.take(M * 4 * d * e);
  }
  t.x_last = a.take(M * d * xs);
  t.patches = a.take(g2rows * pk * e);
  t.x_pre = a.take<float>(g2rows ? M * d * 4 : 0);
  t.patch_out = a.take<float>(g2rows * d * 4);
  t.rows = a.take<int32_t>(B * 4);
  t.seq_off = a.take<int32_t>((B + 1) * 4);
  t.pool = a.take(B * d * e);
  t.dx = a.take<float>(M * d * 4);
  t.dx2 = a.take<float>(M * d * 4);
  t.dxe = a.take(M * d * e);
  t.dpre = a.take(M * 4 * d * e);
  t.part_bytes = static_cast<size_t>(256 + 64) * 160 * 256 * 4;      // <= one round of 160x256 f32 tiles (kPartBytes)
  t.part = a.take<float>(t.part_bytes);
  t.dh = a.take(M * d * e);
  t.dqkv = a.take(M * 3 * d * e);
  size_t rmax = 4 * d > pk ? 4 * d : pk;
  rmax = rmax > embed ? rmax : embed;                       // the all-token head's wgrad / dgrad have `embed` rows
  const size_t mp = pad64(M > g2rows ? M : g2rows);
  t.tAB_bytes = rmax * mp * e;
  t.tA = a.take(t.tAB_bytes);
  t.tB = a.take(t.tAB_bytes);
  t.wT = a.take((12 * d > rmax ? 12 * d : rmax) * d * e);   // the four W^T of a block side by side (12 d^2), or one head matrix
  t.tokE = a.take(M * embed * e);                           // all-token head: dtokens as a GEMM operand
  t.tokP = a.take<float>(g2rows ? 0 : M * embed * 4);       // text only
  t.projT = a.take<float>(embed * d * 4);                   //                 d(proj^T) [embed, d] before its transpose
  const size_t wide = d > embed ? d : embed;
  t.small = a.take<float>(2 * B * wide * 4);
  size_t rb = cmh_layernorm_backward_workspace_bytes(static_cast<int>(M), static_cast<int>(d));
  const size_t cands[] = {cmh_colsum_workspace_bytes(static_cast<int>(M), static_cast<int>(4 * d)),
                          cmh_colsum_workspace_bytes(static_cast<int>(B), static_cast<int>((M / B) * d))};
  for (size_t c : cands) rb = c > rb ? c : rb;
  // a block's six deferred reductions keep their partials side by side: 9 [M, d]-bias slices (proj 1, fc 4, out 1, in 3) + 2 LayerNorms
  const size_t batched = 9 * align_up(((M + 63) / 64) * d * 4 + 256, 256) + 2 * align_up(cmh_layernorm_backward_workspace_bytes(static_cast<int>(M), static_cast<int>(d)), 256);
  rb = batched > rb ? batched : rb;
  t.red_bytes = rb;
  t.red = a.take(rb);
  t.total = a.off;
  return t;
}

// the last block's row-wise tail on the pooled rows only (block_forward_train / block_backward): the rule both calls of a step apply,
// so cmh_set_pooled_tail must not change between a tape's forward and its backward; CMH_TRAIN_POOLED_TAIL=0 keeps the training
// towers on the full-size path
bool train_pooled_tail() {
  static const bool off = []() { const char* e = getenv("CMH_TRAIN_POOLED_TAIL"); return e && e[0] == '0'; }();
  return !off && pooled_tail_enabled();
}
int xkind(int xh) { return xh ? kF16 : kF32; }
int ekind(int dt) { return dt == CMH_BF16 ? kBF16 : kF32; }

// training runs keep the fp16 residual stream rule of the inference path (bf16 mode, width % 256 == 0, CMH_RESID_F16 != 0)
int train_xh(int dt, int d) {
  static const bool off = []() { const char* e = getenv("CMH_RESID_F16"); return e && !strcmp(e, "0"); }();
  return dt == CMH_BF16 && d % 256 == 0 && !off;
}

// x [M,d] f32 -> GEMM dtype copy (bf16 mode) or the same pointer (f32 mode)
int as_gemm_operand(int dt, const float* x, void* scratch, size_t n, hipStream_t st, const void** out) {
  if (dt == CMH_F32) { *out = x; return CMH_OK; }
  *out = scratch;
  return cmh_cast_f32_to_bf16(x, scratch, static_cast<int64_t>(n), st);
}

struct BlockGradPtrs { float *in_w, *in_b, *out_w, *out_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b, *fc_w, *fc_b, *proj_w, *proj_b; };

// pooled_rows (the LAST block of a tower whose only output is the pooled feature, like encoders.hip::run_block_pooled): after the
// attention the block's row-wise tail - out_proj, ln_2, the MLP - runs on the B pooled rows only.  The tape slots then hold B-row
// matrices: x_mid, h2, pre, act rows [0, B); the pooled attention rows (out_proj's wgrad operand) sit in h2 rows [B, 2B); x_next
// receives B rows.  block_backward(..., pooled_rows) reads them back the same way.
int block_forward_train(const cmh_block_weights& w, int dt, int xh, const LayerTape& L, void* x_next, int B, int T,
                        int d, int causal, const uint8_t* kpm, hipStream_t st, int rows = -1, const int32_t* seq_off = nullptr,
                        const int32_t* pooled_rows = nullptr) {
  const int M = rows >= 0 ? rows : B * T;
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  const int rx = EPI_BIAS | EPI_RESIDUAL | (xh ? EPI_RES_F16 | EPI_OUT_F16 : 0);
  const size_t esz = dt == CMH_BF16 ? 2 : 4, xsz = xh ? 2 : 4;
  int rc;
  if ((rc = launch_layernorm_x(L.x_in, xh, nullptr, w.ln1_w, w.ln1_b, L.h1, dt == CMH_BF16, M, d, st))) return rc;
  if ((rc = launch_gemm(dt, L.h1, w.in_proj_w, w.in_proj_b, nullptr, L.qkv, M, 3 * d, d, EPI_BIAS | obf, st))) return rc;
  if ((rc = launch_attention_varlen(L.qkv, L.attn, dt, B, T, d, causal, kpm, seq_off, st))) return rc;
  const void* attn = L.attn;
  if (pooled_rows) {
    void* attn_p = static_cast<char*>(L.h2) + static_cast<size_t>(B) * d * esz;
    if ((rc = launch_gather_rows2(L.x_in, L.x_mid, static_cast<int>(d * xsz), L.attn, attn_p, static_cast<int>(d * esz), pooled_rows, B, st))) return rc;
    attn = attn_p;
  }
  const int Mt = pooled_rows ? B : M;
  const void* resid = pooled_rows ? L.x_mid : L.x_in;          // pooled: the gathered residual rows are updated in place
  if ((rc = launch_gemm(dt, attn, w.out_proj_w, w.out_proj_b, static_cast<const float*>(resid), L.x_mid, Mt, d, d, rx, st))) return rc;
  if ((rc = launch_layernorm_x(L.x_mid, xh, nullptr, w.ln2_w, w.ln2_b, L.h2, dt == CMH_BF16, Mt, d, st))) return rc;
  static const bool fuse_act = []() { const char* e = getenv("CMH_FUSE_PRE"); return !(e && e[0] == '0'); }();
  if (dt == CMH_BF16 && (4 * d) % 256 == 0 && fuse_act) {   // the N % 256 == 0 GEMM kernel has the epilogue
    // one launch: the activation from the f32 accumulator into L.act, the bf16 pre-activation into L.pre (EPI_SAVE_PRE)
    if ((rc = launch_gemm(dt, L.h2, w.fc_w, w.fc_b, static_cast<const float*>(L.pre), L.act, Mt, 4 * d, d,
                          EPI_BIAS | EPI_QUICKGELU | EPI_SAVE_PRE | obf, st))) return rc;
  } else {
    if ((rc = launch_gemm(dt, L.h2, w.fc_w, w.fc_b, nullptr, L.pre, Mt, 4 * d, d, EPI_BIAS | obf, st))) return rc;
    if ((rc = cmh_quick_gelu(L.pre, L.act, static_cast<int64_t>(Mt) * 4 * d, ekind(dt), st))) return rc;
  }
  if ((rc = launch_gemm(dt, L.act, w.proj_w, w.proj_b, static_cast<const float*>(L.x_mid), x_next, Mt, d, 4 * d, rx, st))) return rc;
  return CMH_OK;
}

// dW[O, I] = dY^T X with dY [M, O] (kind ky) and X [M, I] (kind kx); db[O] = column sums of dY
struct WgradScratch { void* tA; void* tB; float* part; size_t part_bytes; void* red; size_t red_bytes; FinalJobs* defer = nullptr; };

// dYe (optional): dY as a bf16 GEMM operand when dY itself is f32 (the residual-stream gradient keeps both)
int wgrad_core(int dt, const void* dY, int ky, int O, const void* X, int kx, int I, int M, float* dW, float* db,
               const WgradScratch& w, hipStream_t st, const void* dYe = nullptr) {
  const int mp = static_cast<int>(pad64(M));
  int rc;
  // bf16 mode: dW = dY^T X straight from the row-major operands (TN GEMM, transposing LDS reads): no transposed copies
  const void* dYb = ky == kBF16 ? dY : dYe;
  if (dt == CMH_BF16 && kx == kBF16 && dYb && gemm_wide_tn_supported(O, I, M)) {
    // the bias gradient's first stage (column sums of dY per K split) is a by-product of the same launch
    float* dbp = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(w.red) + 255) & ~static_cast<uintptr_t>(255));
    const bool cs = db && w.red_bytes >= static_cast<size_t>(64) * O * 4 + 256;
    int slices = 0;
    if ((rc = launch_gemm_wide_tn(dYb, X, dW, w.part, w.part_bytes, O, I, M, st, cs ? dbp : nullptr, &slices))) return rc;
    if (cs) {
      if (w.defer && w.defer->n < FinalJobs::kMax) w.defer->add(dbp, slices, O, db);
      else if ((rc = launch_colsum_final(dbp, slices, O, db, st))) return rc;
    } else if (db && (rc = cmh_colsum(dY, ky, M, O, db, w.red, w.red_bytes, st))) return rc;
    return CMH_OK;
  }
  // db = column sums of dY: their first stage rides on the transpose of dY when that takes the tiled path
  const bool fuse_db = db && transpose_is_vectorised(dY, w.tA, M, O, mp) && w.red_bytes >= static_cast<size_t>((M + 63) / 64) * O * 4 + 256;
  float* dbp = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(w.red) + 255) & ~static_cast<uintptr_t>(255));
  if ((rc = launch_transpose(dY, ky, w.tA, ekind(dt), M, O, mp, st, fuse_db ? dbp : nullptr))) return rc;
  if (fuse_db && w.defer && w.defer->n < FinalJobs::kMax) w.defer->add(dbp, (M + 63) / 64, O, db);     // final stage queued
  else if (fuse_db && (rc = launch_colsum_final(dbp, (M + 63) / 64, O, db, st))) return rc;
  if ((rc = launch_transpose(X, kx, w.tB, ekind(dt), M, I, mp, st))) return rc;
  const int S = gemm_wide_splitk_plan(dt, O, I, mp);
  if (S > 1 && static_cast<size_t>(S) * O * I * 4 <= w.part_bytes) {
    // few output tiles, K = M rows: split K over S groups of workgroups, partial planes summed afterwards
    if ((rc = launch_gemm_wide_splitk(dt, w.tA, w.tB, dW, w.part, S, O, I, mp, st))) return rc;
  } else if ((rc = launch_gemm(dt, w.tA, w.tB, nullptr, nullptr, dW, O, I, mp, 0, st))) return rc;
  if (db && !fuse_db && (rc = cmh_colsum(dY, ky, M, O, db, w.red, w.red_bytes, st))) return rc;
  return CMH_OK;
}

int wgrad(int dt, const void* dY, int ky, int O, const void* X, int kx, int I, int M, float* dW, float* db, TrainBufs& t,
          hipStream_t st, FinalJobs* defer = nullptr, size_t red_off = 0, const void* dYe = nullptr) {
  return wgrad_core(dt, dY, ky, O, X, kx, I, M, dW, db,
                    WgradScratch{t.tA, t.tB, t.part, t.part_bytes, static_cast<char*>(t.red) + red_off, t.red_bytes - red_off, defer}, st,
                    dYe);
}

// dX[M, I] = dY[M, O] . W[O, I]  (W in the GEMM dtype, row-major [O, I]); out typed by `epi`
int dgrad(int dt, const void* dYe, const void* W, int O, int I, int M, const void* aux, void* dX, int epi, TrainBufs& t,
          hipStream_t st, const void* Wt = nullptr) {      // Wt: W^T [I, O] when the caller has it already
  int rc;
  if (!Wt) {
    if ((rc = launch_transpose(W, ekind(dt), t.wT, ekind(dt), O, I, O, st))) return rc;      // W^T [I, O]
    Wt = t.wT;
  }
  return launch_gemm(dt, dYe, Wt, nullptr, static_cast<const float*>(aux), dX, M, I, O, epi, st);
}

int zero_pad_buffers(TrainBufs& t, size_t M, hipStream_t st);

static int g_grad_stream16 = -1;      // cmh_set_grad_stream16: -1 = from the environment (CMH_GRAD_STREAM16=0 switches it off)
bool grad_stream16_enabled() {
  static const bool env_on = []() { const char* e = getenv("CMH_GRAD_STREAM16"); return !(e && e[0] == '0'); }();
  return g_grad_stream16 < 0 ? env_on : g_grad_stream16 != 0;
}

// pooled_rows / dxp (the last block after block_forward_train(..., pooled_rows)): the incoming gradient is dxp [B, d] f32 on the
// pooled rows; steps 1-4 run on those B rows, then the two gradients that go on - d(attention output) and the residual stream -
// are scattered into zeroed full-size buffers for the attention backward and ln_1.
// keep_f32: the caller reads the f32 gradient stream (t.dx) behind this block (the embeddings' backward after block 0; the
// caller-visible dx of cmh_blocks_backward) - see the 16-bit stream below.
int block_backward(const cmh_block_weights& w, const BlockGradPtrs& g, int dt, int xh, const LayerTape& L, TrainBufs& t, int B,
                   int T, int d, int causal, const uint8_t* kpm, hipStream_t st, bool dxe_ready, int rows = -1,
                   const int32_t* seq_off = nullptr, const int32_t* pooled_rows = nullptr, float* dxp = nullptr,
                   bool keep_f32 = true) {
  const int M = rows >= 0 ? rows : B * T;
  const int Mt = pooled_rows ? B : M;                  // rows of the row-wise tail (steps 1-4)
  float* dxt = pooled_rows ? dxp : t.dx;               // its gradient stream
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  const int ek = ekind(dt), xk = xkind(xh);
  const size_t md = static_cast<size_t>(M) * d;
  const void* dxe = nullptr;
  // bf16 mode: the LayerNorm backward kernels also leave the new dx as the bf16 GEMM operand (t.dxe)
  void* dx_copy = dt == CMH_BF16 ? t.dxe : nullptr;
  int rc;
  // The final stage of the block's six small reductions (four bias gradients, two LayerNorm parameter pairs) is queued and
  // launched once at the end of the block; their partial sums live side by side in t.red (sized for that by carve_train).
  FinalJobs jobs;
  const size_t db_slice = align_up(static_cast<size_t>((M + 63) / 64) * d * 4 + 256, 256);       // partials of a [M, d] bias gradient
  const size_t ln_ws = align_up(cmh_layernorm_backward_workspace_bytes(M, d), 256);
  const size_t off_proj = 0, off_fc = db_slice, off_out = 5 * db_slice, off_in = 6 * db_slice, off_ln2 = 9 * db_slice,
               off_ln1 = off_ln2 + ln_ws;
  const bool batched = off_ln1 + ln_ws <= t.red_bytes;
  FinalJobs* dj = batched ? &jobs : nullptr;
  char* red = static_cast<char*>(t.red);
  // W^T of the four weight matrices, one launch: proj [d, 4d], fc [4d, d], out_proj [d, d], in_proj [3d, d]
  const size_t esz = dt == CMH_BF16 ? 2 : 4, dd = static_cast<size_t>(d) * d;
  char* wt = static_cast<char*>(t.wT);
  const void* wsrc[4] = {w.proj_w, w.fc_w, w.out_proj_w, w.in_proj_w};
  void* wdst[4] = {wt, wt + 4 * dd * esz, wt + 8 * dd * esz, wt + 9 * dd * esz};
  const int wR[4] = {d, 4 * d, d, 3 * d}, wC[4] = {4 * d, d, d, d};
  if ((rc = launch_transpose_multi(wsrc, wdst, wR, wC, 4, ek, st))) return rc;
  if (pooled_rows && (rc = zero_pad_buffers(t, static_cast<size_t>(B), st))) return rc;      // the tail's transposes are B rows wide
  // Round 4: the block's four weight gradients as ONE launch (gemm_wide.hip: gemm_wide_tn_multi_kernel), issued after the last dgrad
  // and before ln_1's backward: together they are one round of the chip with little or no K split, no partial planes, one launch's
  // fixed cost instead of four.  Their operands must all still be there at that point: the residual gradient as a GEMM operand exists
  // in two versions per block (at the block's entry: c_proj's dY; after ln_2: out_proj's), so ln_2's backward writes its copy to a
  // second buffer (t.dx2, otherwise only the pooled tail's) and ln_1's, as before, to t.dxe - the next block's entry.
  const TnMultiJob probe[4] = {{nullptr, nullptr, nullptr, nullptr, d, 4 * d, M}, {nullptr, nullptr, nullptr, nullptr, 4 * d, d, M},
                               {nullptr, nullptr, nullptr, nullptr, d, d, M}, {nullptr, nullptr, nullptr, nullptr, 3 * d, d, M}};
  const bool multi = dt == CMH_BF16 && !pooled_rows && batched && dx_copy && gemm_wide_tn_multi_enabled() && gemm_wide_tn_multi_fits(probe, 4) &&
                     db_slice >= static_cast<size_t>(64) * d * 4 + 256;
  void* dx_copy2 = multi ? static_cast<void*>(t.dx2) : dx_copy;      // where ln_2's backward leaves the bf16 copy of the new dx
  // Round 5: the 16-bit gradient stream.  The residual gradient is needed twice per block as a bf16 GEMM operand (c_proj's and
  // out_proj's dY) and twice as the sum a LayerNorm backward adds to; kept in f32 the two LayerNorm launches read AND write it
  // beside the bf16 copy they leave anyway (14 bytes per element: x 2, dy 2, dx 4 + 4, copy 2).  In the bf16 mode's multi-launch
  // blocks the bf16 copy IS the stream: ln_2 adds to the entry copy (t.dxe) and writes t.dx2, ln_1 adds to t.dx2 and writes t.dxe
  // - 8 bytes per element; the sum is formed in f32 and rounded to bf16 once per LayerNorm (24 roundings along a 12-block tower,
  // each 2^-9 relative: below what the bf16 GEMM operands already cost, tests/test_gpu_backward.py).  t.dx is written only where
  // somebody reads it (keep_f32).  The f32 mode, the pooled last block and CMH_GRAD_STREAM16=0 keep the f32 stream.
  const bool s16 = multi && grad_stream16_enabled();
  // 1. MLP projection (dxe_ready: the previous block's ln_1 backward already wrote t.dxe)
  const size_t mtd = static_cast<size_t>(Mt) * d;
  if (dxe_ready && dx_copy) dxe = t.dxe;
  else if ((rc = as_gemm_operand(dt, dxt, t.dxe, mtd, st, &dxe))) return rc;
  const void* dxe_entry = dxe;
  if ((rc = dgrad(dt, dxe, w.proj_w, d, 4 * d, Mt, L.pre, t.dpre, EPI_MUL_DQGELU | obf, t, st, wdst[0]))) return rc;
  if (!multi && (rc = wgrad(dt, dxt, kF32, d, L.act, ek, 4 * d, Mt, g.proj_w, g.proj_b, t, st, dj, batched ? off_proj : 0,
                            dt == CMH_BF16 ? dxe : nullptr))) return rc;
  // 2. c_fc
  if ((rc = dgrad(dt, t.dpre, w.fc_w, 4 * d, d, Mt, nullptr, t.dh, obf, t, st, wdst[1]))) return rc;
  if (!multi && (rc = wgrad(dt, t.dpre, ek, 4 * d, L.h2, ek, d, Mt, g.fc_w, g.fc_b, t, st, dj, batched ? off_fc : 0))) return rc;
  // 3. ln_2
  if ((rc = launch_layernorm_backward(L.x_mid, xk, t.dh, ek, w.ln2_w, nullptr, Mt, d, dxt, s16 ? 0 : 1, g.ln2_w, g.ln2_b,
                                      batched ? red + off_ln2 : red, batched ? ln_ws : t.red_bytes, st, dx_copy2, dj,
                                      s16 ? dxe_entry : nullptr, !s16))) return rc;
  // 4. out_proj
  if (dx_copy2) dxe = dx_copy2;
  else if ((rc = as_gemm_operand(dt, dxt, t.dxe, mtd, st, &dxe))) return rc;
  const void* dxe_mid = dxe;
  const void* attn_rows = pooled_rows ? static_cast<const char*>(L.h2) + static_cast<size_t>(B) * d * esz : L.attn;
  if ((rc = dgrad(dt, dxe, w.out_proj_w, d, d, Mt, nullptr, t.dh, obf, t, st, wdst[2]))) return rc;
  if (!multi && (rc = wgrad(dt, dxt, kF32, d, attn_rows, ek, d, Mt, g.out_w, g.out_b, t, st, dj, batched ? off_out : 0,
                            dt == CMH_BF16 ? dxe : nullptr))) return rc;
  const void* dattn = t.dh;
  if (pooled_rows) {
    // back to full size: d(attention output) is zero off the pooled rows (t.dxe is free now), the residual gradient likewise
    if (hipMemsetAsync(t.dxe, 0, md * esz, st) != hipSuccess || hipMemsetAsync(t.dx, 0, md * 4, st) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "backward: memset failed");
    if ((rc = launch_scatter_rows(t.dh, pooled_rows, t.dxe, B, static_cast<int>(d * esz), st))) return rc;
    if ((rc = launch_scatter_rows(dxp, pooled_rows, t.dx, B, d * 4, st))) return rc;
    dattn = t.dxe;
    if ((rc = zero_pad_buffers(t, static_cast<size_t>(M), st))) return rc;       // the B-row transposes left their rows in the padding
  }
  // 5. attention
  if ((rc = launch_attention_backward(dt, L.qkv, L.attn, dattn, t.dqkv, B, T, d, causal, kpm, seq_off, st))) return rc;
  // 6. in_proj
  if ((rc = dgrad(dt, t.dqkv, w.in_proj_w, 3 * d, d, M, nullptr, t.dh, obf, t, st, wdst[3]))) return rc;
  if (!multi && (rc = wgrad(dt, t.dqkv, ek, 3 * d, L.h1, ek, d, M, g.in_w, g.in_b, t, st, dj, batched ? off_in : 0))) return rc;
  if (multi) {
    auto dbp = [&](size_t off) { return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(red + off) + 255) & ~static_cast<uintptr_t>(255)); };
    const TnMultiJob jobs4[4] = {{dxe_entry, L.act, g.proj_w, dbp(off_proj), d, 4 * d, M},
                                 {t.dpre, L.h2, g.fc_w, dbp(off_fc), 4 * d, d, M},
                                 {dxe_mid, L.attn, g.out_w, dbp(off_out), d, d, M},
                                 {t.dqkv, L.h1, g.in_w, dbp(off_in), 3 * d, d, M}};
    float* const dbs[4] = {g.proj_b, g.fc_b, g.out_b, g.in_b};
    int slices[4] = {0, 0, 0, 0};
    if ((rc = launch_gemm_wide_tn_multi(jobs4, 4, t.part, t.part_bytes, st, slices))) return rc;
    for (int i = 0; i < 4; ++i) jobs.add(jobs4[i].colsum, slices[i], jobs4[i].Mm, dbs[i]);
  }
  // 7. ln_1
  if ((rc = launch_layernorm_backward(L.x_in, xk, t.dh, ek, w.ln1_w, nullptr, M, d, t.dx, s16 ? 0 : 1, g.ln1_w, g.ln1_b,
                                      batched ? red + off_ln1 : red, batched ? ln_ws : t.red_bytes, st, dx_copy, dj,
                                      s16 ? dxe_mid : nullptr, !s16 || keep_f32))) return rc;
  return launch_final_jobs(jobs, st);
}

BlockGradPtrs grads_of(const cmh_block_grads& g) {
  return BlockGradPtrs{g.in_proj_w, g.in_proj_b, g.out_proj_w, g.out_proj_b, g.ln1_w, g.ln1_b, g.ln2_w, g.ln2_b,
                       g.fc_w, g.fc_b, g.proj_w, g.proj_b};
}
int check_block_grads(const cmh_block_grads* g, int layers) {
  for (int i = 0; i < layers; ++i) {
    const BlockGradPtrs p = grads_of(g[i]);
    CMH_CHECK_ARG(p.in_w && p.in_b && p.out_w && p.out_b && p.ln1_w && p.ln1_b && p.ln2_w && p.ln2_b && p.fc_w && p.fc_b &&
                  p.proj_w && p.proj_b, "backward: block %d has a null gradient pointer", i);
  }
  return CMH_OK;
}

// ---- pooled rows: feat = LN(x_last[rows]) . proj ------------------------------------------------------------------------------
// dpool[b, i] = sum_j dfeat[b, j] * proj_t[j, i]   (dfeat row in LDS, 8 independent partial sums per thread; 32 chains or 8 rows per workgroup: 217 / 68-109 us against 72)
template <typename T>
__global__ __launch_bounds__(256) void pool_dgrad_kernel(const float* __restrict__ dfeat, const T* __restrict__ proj_t,
                                                         float* __restrict__ dpool, int B, int d, int E) {
  __shared__ float row[2048];
  const int b = blockIdx.x;
  constexpr int kind = sizeof(T) == 4 ? kF32 : kBF16;
  for (int j = threadIdx.x; j < E; j += 256) row[j] = dfeat[static_cast<size_t>(b) * E + j];
  __syncthreads();
  for (int i = threadIdx.x; i < d; i += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 8 <= E; j += 8)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = fmaf(row[j + u], load_as_f32(proj_t, static_cast<size_t>(j + u) * d + i, kind), acc[u]);
    for (; j < E; ++j) acc[0] = fmaf(row[j], load_as_f32(proj_t, static_cast<size_t>(j) * d + i, kind), acc[0]);
    dpool[static_cast<size_t>(b) * d + i] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
}
// bf16 weights, d % 8 == 0: a thread owns 8 consecutive outputs and reads them as ONE 16-byte load per j (the kernel above issues a
// 2-byte load per output and j: 3 x 512 of them per thread in chains of 8 - 72 us at batch 256, and more of them in flight was slower,
// not faster).  Every output keeps the kernel above's arithmetic: chain u = j mod 8, the same final tree - same bits.
__global__ __launch_bounds__(128) void pool_dgrad_v8_kernel(const float* __restrict__ dfeat, const bf16_t* __restrict__ proj_t,
                                                            float* __restrict__ dpool, int B, int d, int E) {
  __shared__ float row[2048];
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < E; j += 128) row[j] = dfeat[static_cast<size_t>(b) * E + j];
  __syncthreads();
  for (int i8 = threadIdx.x; i8 < d / 8; i8 += 128) {
    float acc[8][8];                                   // [column][chain]
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[c][u] = 0.f;
    auto fma8 = [&](const uint4& q, float x, int u) __attribute__((always_inline)) {
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        acc[2 * h][u] = fmaf(x, __uint_as_float(w[h] << 16), acc[2 * h][u]);
        acc[2 * h + 1][u] = fmaf(x, __uint_as_float(w[h] & 0xffff0000u), acc[2 * h + 1][u]);
      }
    };
    int j = 0;
    for (; j + 8 <= E; j += 8) {
      uint4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const uint4*>(proj_t + static_cast<size_t>(j + u) * d + i8 * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u) fma8(q[u], row[j + u], u);
    }
    for (; j < E; ++j) fma8(*reinterpret_cast<const uint4*>(proj_t + static_cast<size_t>(j) * d + i8 * 8), row[j], 0);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      dpool[static_cast<size_t>(b) * d + i8 * 8 + c] =
          ((acc[c][0] + acc[c][1]) + (acc[c][2] + acc[c][3])) + ((acc[c][4] + acc[c][5]) + (acc[c][6] + acc[c][7]));
  }
}
// dproj[i, j] = sum_b pool[b, i] * dfeat[b, j]      (the reference's [width, embed_dim] parameter layout)
template <typename T>
__global__ __launch_bounds__(256) void pool_wgrad_kernel(const T* __restrict__ pool, const float* __restrict__ dfeat,
                                                         float* __restrict__ dproj, int B, int d, int E) {
  const int i = blockIdx.x;
  constexpr int kind = sizeof(T) == 4 ? kF32 : kBF16;
  for (int j = threadIdx.x; j < E; j += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // eight independent chains over the batch (one chain: 137 us)
    int b = 0;
    for (; b + 8 <= B; b += 8)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc[u] = fmaf(load_as_f32(pool, static_cast<size_t>(b + u) * d + i, kind), dfeat[static_cast<size_t>(b + u) * E + j], acc[u]);
    for (; b < B; ++b) acc[0] = fmaf(load_as_f32(pool, static_cast<size_t>(b) * d + i, kind), dfeat[static_cast<size_t>(b) * E + j], acc[0]);
    dproj[static_cast<size_t>(i) * E + j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
}

// compact = the forward carried only the pooled rows through the last block (x_last holds B rows): the gradient stays compact too,
// B rows of t.dx2, for block_backward(..., pooled_rows, dxp)
int pooled_backward(int dt, int xh, const void* x_last, const int32_t* rows, const void* pool, const void* proj_t,
                    const float* ln_w, const float* dfeat, float* dproj, float* dln_w, float* dln_b, TrainBufs& t, int B, int M,
                    int d, int E, hipStream_t st, bool compact = false) {
  float* dpool = t.small;
  if (dt == CMH_F32) {
    hipLaunchKernelGGL(pool_dgrad_kernel<float>, dim3(B), dim3(256), 0, st, dfeat, static_cast<const float*>(proj_t), dpool, B, d, E);
    hipLaunchKernelGGL(pool_wgrad_kernel<float>, dim3(d), dim3(256), 0, st, static_cast<const float*>(pool), dfeat, dproj, B, d, E);
  } else {
    if (d % 8 == 0 && E <= 2048 && (reinterpret_cast<uintptr_t>(proj_t) & 15) == 0)
      hipLaunchKernelGGL(pool_dgrad_v8_kernel, dim3(B), dim3(128), 0, st, dfeat, static_cast<const bf16_t*>(proj_t), dpool, B, d, E);
    else
      hipLaunchKernelGGL(pool_dgrad_kernel<bf16_t>, dim3(B), dim3(256), 0, st, dfeat, static_cast<const bf16_t*>(proj_t), dpool, B, d, E);
    hipLaunchKernelGGL(pool_wgrad_kernel<bf16_t>, dim3(d), dim3(256), 0, st, static_cast<const bf16_t*>(pool), dfeat, dproj, B, d, E);
  }
  CMH_CHECK_LAUNCH("pooled projection backward");
  if (compact)
    return launch_layernorm_backward(x_last, xkind(xh), dpool, kF32, ln_w, nullptr, B, d, t.dx2, 0, dln_w, dln_b, t.red, t.red_bytes, st);
  if (hipMemsetAsync(t.dx, 0, static_cast<size_t>(M) * d * 4, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "backward: memset failed");
  return launch_layernorm_backward(x_last, xkind(xh), dpool, kF32, ln_w, rows, B, d, t.dx, 0, dln_w, dln_b, t.red, t.red_bytes, st);
}

// token embedding: dE[token[r], :] += dx[r, :]
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ dx,
                                                            float* __restrict__ dE, int rows, int d, int vocab) {
  const int r = blockIdx.x;
  int64_t id = tokens[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dE + static_cast<size_t>(id) * d + c, dx[static_cast<size_t>(r) * d + c]);
}

// packed text: the training tower follows encode_text's packing rule (no mask, CMH_TEXT_PACK != 0)
bool text_packing(const uint8_t* kpm) {
  static const bool off = []() { const char* e = getenv("CMH_TEXT_PACK"); return e && !strcmp(e, "0"); }();
  return !kpm && !off;
}

// dpos[t, :] = sum over the captions that have a token t of dx[seq_off[b] + t, :]   (deterministic: fixed caption order)
__global__ __launch_bounds__(256) void packed_dpos_kernel(const float* __restrict__ dx, const int32_t* __restrict__ seq_off, int B, int d,
                                                          float* __restrict__ dpos) {
  const int t = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= d) return;
  float acc = 0.f;
  // eight captions' rows in flight, added in caption order (one dependent load per caption made this 75 us of pure latency)
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = seq_off[b + u];
      const bool has = t < seq_off[b + u + 1] - o;
      v[u] = has ? dx[static_cast<size_t>(o + t) * d + c] : 0.f;
      v[u] = has ? v[u] : -0.f;                                // (x + -0 == x for every x: a caption without token t adds nothing, bit for bit)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; b < B; ++b) {
    const int o = seq_off[b];
    if (t < seq_off[b + 1] - o) acc += dx[static_cast<size_t>(o + t) * d + c];
  }
  dpos[static_cast<size_t>(t) * d + c] = acc;
}

// token embedding from packed rows: dE[tokens[b, t], :] += dx[seq_off[b] + t, :]
__global__ __launch_bounds__(256) void packed_embed_scatter_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ dx,
                                                                   const int32_t* __restrict__ seq_off, float* __restrict__ dE, int B,
                                                                   int L, int d, int vocab) {
  const int b = blockIdx.x / L, t = blockIdx.x - b * L;
  if (t >= seq_off[b + 1] - seq_off[b]) return;
  int64_t id = tokens[blockIdx.x];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const size_t r = static_cast<size_t>(seq_off[b] + t);
  for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dE + static_cast<size_t>(id) * d + c, dx[r * d + c]);
}

int zero_pad_buffers(TrainBufs& t, size_t M, hipStream_t st) {
  if (pad64(M) == M) return CMH_OK;     // no padding columns: nothing to clear
  if (hipMemsetAsync(t.tA, 0, t.tAB_bytes, st) != hipSuccess || hipMemsetAsync(t.tB, 0, t.tAB_bytes, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "backward: memset failed");
  return CMH_OK;
}

int check_train_tower(int dt, int width, int layers, int embed, const cmh_block_weights* blocks) {
  CMH_CHECK_ARG(dt == CMH_F32 || dt == CMH_BF16, "bad gemm_dtype %d", dt);
  CMH_CHECK_ARG(width > 0 && width % 128 == 0 && width <= 1024, "width %d must be a multiple of 128, <= 1024", width);
  CMH_CHECK_ARG(layers >= 0 && embed > 0 && embed % 4 == 0, "bad layers/embed_dim");
  CMH_CHECK_ARG(layers == 0 || blocks, "blocks is null");
  return CMH_OK;
}

}  // namespace
}  // namespace cmh

using namespace cmh;

// ================================================================================================================ wgrad
static constexpr size_t kPartBytes = static_cast<size_t>(256 + 64) * 160 * 256 * 4;      // <= one round of 160x256 f32 tiles

extern "C" size_t cmh_linear_wgrad_workspace_bytes(int32_t dtype, int32_t M, int32_t O, int32_t I) {
  if (M <= 0 || O <= 0 || I <= 0) return 0;
  const size_t e = dtype == CMH_BF16 ? 2 : 4, mp = pad64(static_cast<size_t>(M));
  return align_up(static_cast<size_t>(O) * mp * e, 256) + align_up(static_cast<size_t>(I) * mp * e, 256) + kPartBytes +
         align_up(cmh_colsum_workspace_bytes(M, O), 256) + 256;
}

extern "C" int cmh_linear_wgrad(int32_t dtype, const void* dy, int32_t dy_kind, const void* x, int32_t x_kind, int32_t M, int32_t O,
                                int32_t I, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(dy && x && dw && workspace && M > 0 && O > 0 && I > 0, "linear_wgrad: bad arguments");
  CMH_CHECK_ARG(dtype == CMH_F32 || dtype == CMH_BF16, "linear_wgrad: bad dtype");
  if (workspace_bytes < cmh_linear_wgrad_workspace_bytes(dtype, M, O, I)) return fail(CMH_ERR_WORKSPACE, "linear_wgrad: workspace too small");
  const size_t e = dtype == CMH_BF16 ? 2 : 4, mp = pad64(static_cast<size_t>(M));
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  WgradScratch w;
  w.tA = ws; ws += align_up(static_cast<size_t>(O) * mp * e, 256);
  w.tB = ws; ws += align_up(static_cast<size_t>(I) * mp * e, 256);
  w.part = reinterpret_cast<float*>(ws); w.part_bytes = kPartBytes; ws += kPartBytes;
  w.red = ws; w.red_bytes = cmh_colsum_workspace_bytes(M, O);
  hipStream_t st = as_stream(stream);
  if (mp != static_cast<size_t>(M) &&
      (hipMemsetAsync(w.tA, 0, static_cast<size_t>(O) * mp * e, st) != hipSuccess || hipMemsetAsync(w.tB, 0, static_cast<size_t>(I) * mp * e, st) != hipSuccess))
    return fail(CMH_ERR_LAUNCH, "linear_wgrad: memset failed");
  return wgrad_core(dtype, dy, dy_kind, O, x, x_kind, I, M, dw, db, w, st);
}

// ---- all-token head (MITH trunk, model/MITH.py:70-80,136-139): tokens = LN(x_last, every row) . proj ---------------------------
// forward: the normalised rows go through t.dxe (free during the forward pass)
int tokens_head_forward(int dt, int xh, const TrainBufs& t, const float* ln_w, const float* ln_b, const void* proj_t, float* tokens_out,
                        int M, int d, int E, hipStream_t st) {
  int rc;
  if ((rc = launch_layernorm_x(t.x_last, xh, nullptr, ln_w, ln_b, t.dxe, dt == CMH_BF16, M, d, st))) return rc;
  const int bk = dt == CMH_F32 ? 32 : 64;
  if (E % 128 == 0 && d % bk == 0) return launch_gemm(dt, t.dxe, proj_t, nullptr, nullptr, tokens_out, M, E, d, 0, st);
  return launch_small_linear(dt, t.dxe, proj_t, nullptr, nullptr, 1.f, CMH_ACT_NONE, tokens_out, M, E, d, st);
}
// backward: dproj (reference layout [d, E]) = (dtok^T h)^T, dh = dtok . proj_t, then LayerNorm backward over every row into t.dx
// (+ its bf16 operand copy in t.dxe).  Scratch: h in t.dxe, dtok operand in t.tokE, dproj^T in t.projT.
int tokens_head_backward(int dt, int xh, TrainBufs& t, const float* ln_w, const float* ln_b, const void* proj_t, const float* dtok,
                         float* dproj, float* dln_w, float* dln_b, int M, int d, int E, hipStream_t st) {
  int rc;
  if ((rc = launch_layernorm_x(t.x_last, xh, nullptr, ln_w, ln_b, t.dxe, dt == CMH_BF16, M, d, st))) return rc;     // h again
  float* dprojT = t.projT;
  if ((rc = wgrad(dt, dtok, kF32, E, t.dxe, ekind(dt), d, M, dprojT, nullptr, t, st))) return rc;                     // [E, d]
  if ((rc = launch_transpose(dprojT, kF32, dproj, kF32, E, d, E, st))) return rc;                                      // -> [d, E]
  const void* dtok_e = nullptr;
  if ((rc = as_gemm_operand(dt, dtok, t.tokE, static_cast<size_t>(M) * E, st, &dtok_e))) return rc;
  const int obf = dt == CMH_BF16 ? EPI_OUT_BF16 : 0;
  if ((rc = dgrad(dt, dtok_e, proj_t, E, d, M, nullptr, t.dh, obf, t, st))) return rc;
  return launch_layernorm_backward(t.x_last, xkind(xh), t.dh, ekind(dt), ln_w, nullptr, M, d, t.dx, 0, dln_w, dln_b, t.red, t.red_bytes,
                                   st, dt == CMH_BF16 ? t.dxe : nullptr);
}

// ================================================================================================================ vision
extern "C" size_t cmh_vit_train_bytes(const cmh_vit_weights* w, int32_t batch) {
  if (!w || batch <= 0 || w->patch <= 0) return 0;
  const size_t g = w->resolution / w->patch, g2 = g * g, T = g2 + 1, d = w->width, B = batch;
  const size_t e = w->gemm_dtype == CMH_BF16 ? 2 : 4, xs = train_xh(w->gemm_dtype, w->width) ? 2 : 4;
  return carve_train(nullptr, B * T, B, d, e, xs, w->layers, B * g2, 3ull * w->patch * w->patch, w->embed_dim).total;
}

static int vit_forward_train_impl(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, float* tokens_out,
                                  void* tape, size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(w && image && (feat || tokens_out) && tape && batch > 0, "vit_forward_train: bad arguments");
  int rc = check_train_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(w->layers > 0, "vit_forward_train: no layers");
  CMH_CHECK_ARG(w->patch > 0 && w->resolution % w->patch == 0 && w->patch % 4 == 0, "vit_forward_train: resolution / patch");
  const int dt = w->gemm_dtype, d = w->width, B = batch;
  const int g = w->resolution / w->patch, g2 = g * g, T = g2 + 1, M = B * T, pk = 3 * w->patch * w->patch;
  CMH_CHECK_ARG(pk % (dt == CMH_F32 ? 32 : 64) == 0, "vit_forward_train: 3*patch^2 = %d not a multiple of the GEMM K-step", pk);
  if (tape_bytes < cmh_vit_train_bytes(w, batch)) return fail(CMH_ERR_WORKSPACE, "vit_forward_train: tape too small");
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(tape) & 255) == 0, "vit_forward_train: tape must be 256-byte aligned");
  const int xh = train_xh(dt, d);
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, static_cast<size_t>(M), B, d, e, xh ? 2 : 4, w->layers, static_cast<size_t>(B) * g2, pk, w->embed_dim);
  if ((rc = launch_patchify(image, t.patches, dt, B, w->resolution, w->patch, st))) return rc;
  if ((rc = launch_gemm(dt, t.patches, w->conv1_w, nullptr, nullptr, t.patch_out, B * g2, d, pk, 0, st))) return rc;
  // x_pre = [cls ; patches] + positional (kept for ln_pre's backward), x_0 = ln_pre(x_pre)
  if ((rc = launch_vit_assemble(t.patch_out, w->class_embedding, w->positional_embedding, t.x_pre, B, g2, d, st))) return rc;
  if ((rc = launch_layernorm_any(t.x_pre, kF32, nullptr, w->ln_pre_w, w->ln_pre_b, t.L[0].x_in, xkind(xh), M, d, st))) return rc;
  const bool tail = !tokens_out && train_pooled_tail();      // the same rule in vit_backward_impl
  if ((rc = launch_iota_rows(t.rows, B, T, st))) return rc;
  for (int i = 0; i < w->layers; ++i) {
    void* nxt = i + 1 < w->layers ? t.L[i + 1].x_in : t.x_last;
    if ((rc = block_forward_train(w->blocks[i], dt, xh, t.L[i], nxt, B, T, d, 0, nullptr, st, -1, nullptr,
                                  tail && i == w->layers - 1 ? t.rows : nullptr))) return rc;
  }
  if (tokens_out) return tokens_head_forward(dt, xh, t, w->ln_post_w, w->ln_post_b, w->proj_t, tokens_out, M, d, w->embed_dim, st);
  if ((rc = launch_layernorm_x(t.x_last, xh, tail ? nullptr : t.rows, w->ln_post_w, w->ln_post_b, t.pool, dt == CMH_BF16, B, d, st))) return rc;
  const int bk = dt == CMH_F32 ? 32 : 64;
  if (w->embed_dim % 128 == 0 && d % bk == 0) return launch_gemm(dt, t.pool, w->proj_t, nullptr, nullptr, feat, B, w->embed_dim, d, 0, st);
  return launch_small_linear(dt, t.pool, w->proj_t, nullptr, nullptr, 1.f, CMH_ACT_NONE, feat, B, w->embed_dim, d, st);
}

extern "C" int cmh_vit_forward_train(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, void* tape,
                                     size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(feat, "vit_forward_train: null feature pointer");
  return vit_forward_train_impl(w, image, batch, feat, nullptr, tape, tape_bytes, stream);
}

extern "C" int cmh_vit_forward_train_tokens(const cmh_vit_weights* w, const float* image, int32_t batch, float* tokens_out,
                                            void* tape, size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "vit_forward_train_tokens: null output pointer");
  return vit_forward_train_impl(w, image, batch, nullptr, tokens_out, tape, tape_bytes, stream);
}

// layer_hi / layer_lo: the part of the backward pass this call runs - the head (ln_post, proj) iff layer_hi == layers, then blocks
// layer_hi-1 .. layer_lo, then the embeddings (ln_pre, positional, class, conv1) iff layer_lo == 0.  The gradient stream between
// parts lives in the tape, so consecutive parts (layers .. a, a .. b, b .. 0) equal the one-call form bit for bit; each part's
// parameter gradients are final when it returns (a data-parallel trainer sends that bucket while the next part runs).
static int vit_backward_impl(const cmh_vit_weights* w, int32_t batch, const float* dfeat, const float* dtokens, const cmh_vit_grads* gr,
                             void* tape, size_t tape_bytes, void* stream, int layer_hi = -1, int layer_lo = 0) {
  CMH_CHECK_ARG(w && (dfeat || dtokens) && gr && tape && batch > 0, "vit_backward: bad arguments");
  if (layer_hi < 0) layer_hi = w->layers;
  CMH_CHECK_ARG(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= w->layers, "vit_backward: layers [%d, %d) of %d", layer_lo, layer_hi, w->layers);
  int rc = check_train_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(gr->conv1_w && gr->class_embedding && gr->positional_embedding && gr->ln_pre_w && gr->ln_pre_b && gr->ln_post_w &&
                gr->ln_post_b && gr->proj && gr->blocks, "vit_backward: null gradient pointer");
  if ((rc = check_block_grads(gr->blocks, w->layers))) return rc;
  if (tape_bytes < cmh_vit_train_bytes(w, batch)) return fail(CMH_ERR_WORKSPACE, "vit_backward: tape too small");
  const int dt = w->gemm_dtype, d = w->width, B = batch, E = w->embed_dim;
  const int g = w->resolution / w->patch, g2 = g * g, T = g2 + 1, M = B * T, pk = 3 * w->patch * w->patch;
  const int xh = train_xh(dt, d);
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, static_cast<size_t>(M), B, d, e, xh ? 2 : 4, w->layers, static_cast<size_t>(B) * g2, pk, E);
  const bool tail = !dtokens && train_pooled_tail();
  if (layer_hi == w->layers) {
    if ((rc = zero_pad_buffers(t, static_cast<size_t>(M), st))) return rc;
    if (dtokens) {
      if ((rc = tokens_head_backward(dt, xh, t, w->ln_post_w, w->ln_post_b, w->proj_t, dtokens, gr->proj, gr->ln_post_w, gr->ln_post_b, M, d,
                                     E, st))) return rc;
    } else if ((rc = pooled_backward(dt, xh, t.x_last, t.rows, t.pool, w->proj_t, w->ln_post_w, dfeat, gr->proj, gr->ln_post_w,
                                     gr->ln_post_b, t, B, M, d, E, st, tail))) return rc;
  }
  for (int i = layer_hi - 1; i >= layer_lo; --i) {
    const bool last = i == w->layers - 1;
    if ((rc = block_backward(w->blocks[i], grads_of(gr->blocks[i]), dt, xh, t.L[i], t, B, T, d, 0, nullptr, st,
                             !last || dtokens != nullptr, -1, nullptr, tail && last ? t.rows : nullptr, tail && last ? t.dx2 : nullptr,
                             /*keep_f32=*/i == 0))) return rc;
  }
  if (layer_lo > 0) return CMH_OK;
  // ln_pre, then the embeddings: x_pre[b,0] = cls + pos[0], x_pre[b,1+i] = patch_out[b*g2+i] + pos[1+i]
  if ((rc = launch_layernorm_backward(t.x_pre, kF32, t.dx, kF32, w->ln_pre_w, nullptr, M, d, t.dx2, 0, gr->ln_pre_w, gr->ln_pre_b,
                                      t.red, t.red_bytes, st))) return rc;
  if ((rc = cmh_colsum(t.dx2, kF32, B, T * d, gr->positional_embedding, t.red, t.red_bytes, st))) return rc;
  if (hipMemcpyAsync(gr->class_embedding, gr->positional_embedding, static_cast<size_t>(d) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "vit_backward: copy failed");
  // conv1: dW[d, pk] = dpatch_out^T patches, dpatch_out = the non-class rows of dx2 (compacted into dx)
  if (hipMemcpy2DAsync(t.dx, static_cast<size_t>(g2) * d * 4, t.dx2 + d, static_cast<size_t>(T) * d * 4, static_cast<size_t>(g2) * d * 4,
                       B, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "vit_backward: compaction failed");
  if ((rc = zero_pad_buffers(t, static_cast<size_t>(B) * g2, st))) return rc;
  return wgrad(dt, t.dx, kF32, d, t.patches, ekind(dt), pk, B * g2, gr->conv1_w, nullptr, t, st);
}

extern "C" int cmh_set_grad_stream16(int32_t on) {
  CMH_CHECK_ARG(on >= -1 && on <= 1, "set_grad_stream16: %d (-1 environment, 0 off, 1 on)", on);
  g_grad_stream16 = on;
  return CMH_OK;
}

extern "C" int cmh_vit_backward(const cmh_vit_weights* w, int32_t batch, const float* dfeat, const cmh_vit_grads* gr, void* tape,
                                size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(dfeat, "vit_backward: null gradient");
  return vit_backward_impl(w, batch, dfeat, nullptr, gr, tape, tape_bytes, stream);
}

extern "C" int cmh_vit_backward_part(const cmh_vit_weights* w, int32_t batch, const float* dfeat, const cmh_vit_grads* gr, void* tape,
                                     size_t tape_bytes, int32_t layer_hi, int32_t layer_lo, void* stream) {
  CMH_CHECK_ARG(dfeat && layer_hi >= 0, "vit_backward_part: bad arguments");
  return vit_backward_impl(w, batch, dfeat, nullptr, gr, tape, tape_bytes, stream, layer_hi, layer_lo);
}

extern "C" int cmh_vit_backward_tokens(const cmh_vit_weights* w, int32_t batch, const float* dtokens, const cmh_vit_grads* gr,
                                       void* tape, size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(dtokens, "vit_backward_tokens: null gradient");
  return vit_backward_impl(w, batch, nullptr, dtokens, gr, tape, tape_bytes, stream);
}

// ================================================================================================================ text
// The packed row count of a text tape (the GEMM grids need it on the host): text_forward_train reads it back once and remembers
// it here, keyed by the tape's address, so that the backward call(s) on the same tape need no stream synchronisation of their own.
namespace {
std::mutex g_tape_rows_mu;
std::unordered_map<const void*, int> g_tape_rows;
std::vector<const void*> g_tape_order;                    // insertion order: the OLDEST entry goes when the table is full
void remember_tape_rows(const void* tape, int rows) {
  std::lock_guard<std::mutex> lk(g_tape_rows_mu);
  if (g_tape_rows.find(tape) == g_tape_rows.end()) {
    if (g_tape_order.size() >= 64) {                      // tapes of long-gone steps; a forward's entry outlives the 63 forwards after it
      g_tape_rows.erase(g_tape_order.front());
      g_tape_order.erase(g_tape_order.begin());
    }
    g_tape_order.push_back(tape);
  }
  g_tape_rows[tape] = rows;
}
int recall_tape_rows(const void* tape) {
  std::lock_guard<std::mutex> lk(g_tape_rows_mu);
  const auto it = g_tape_rows.find(tape);
  return it == g_tape_rows.end() ? -1 : it->second;
}
}  // namespace

extern "C" size_t cmh_text_train_bytes(const cmh_text_weights* w, int32_t batch, int32_t seq_len) {
  if (!w || batch <= 0 || seq_len <= 0) return 0;
  const size_t e = w->gemm_dtype == CMH_BF16 ? 2 : 4, xs = train_xh(w->gemm_dtype, w->width) ? 2 : 4;
  return carve_train(nullptr, static_cast<size_t>(batch) * seq_len, batch, w->width, e, xs, w->layers, 0, 0, w->embed_dim).total;
}

static int text_forward_train_impl(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                   const uint8_t* key_padding_mask, float* feat, float* tokens_out, int32_t* eot_rows_out, void* tape,
                                   size_t tape_bytes, void* stream, bool pack_tokens_req = false) {
  CMH_CHECK_ARG(w && tokens && (feat || tokens_out) && tape && batch > 0 && seq_len > 0, "text_forward_train: bad arguments");
  int rc = check_train_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(w->layers > 0 && seq_len <= w->context_length, "text_forward_train: layers / seq_len");
  if (tape_bytes < cmh_text_train_bytes(w, batch, seq_len)) return fail(CMH_ERR_WORKSPACE, "text_forward_train: tape too small");
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(tape) & 255) == 0, "text_forward_train: tape must be 256-byte aligned");
  const int dt = w->gemm_dtype, d = w->width, B = batch, L = seq_len, M = B * L;
  const int xh = train_xh(dt, d);
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, static_cast<size_t>(M), B, d, e, xh ? 2 : 4, w->layers, 0, 0, w->embed_dim);
  // packed like encode_text (encoders.hip): only the tokens 0..EOT of every caption are run through the blocks
  const int32_t* seq_off = nullptr;
  int rows = M;
  // the all-token trunk (MITH): packed only on the caller's word that the padded positions of tokens_out are read by nobody
  // (cmh_text_forward_train_tokens_packed; encoders.hip text_begin has the reasoning) - their gradient is then exactly zero as well
  const int bk0 = dt == CMH_F32 ? 32 : 64;
  const bool pack_tokens = tokens_out && pack_tokens_req && key_padding_mask && text_token_packing() && w->embed_dim % 128 == 0 && d % bk0 == 0;
  if (tokens_out && !pack_tokens) remember_tape_rows(tape, 0);      // 0: this tape's all-token head is dense (text_backward_impl)
  if ((!tokens_out && text_packing(key_padding_mask)) || pack_tokens) {
    if ((rc = launch_text_pack_plan(tokens, B, L, t.seq_off, st, pack_tokens ? key_padding_mask : nullptr, pack_tokens ? t.rows : nullptr))) return rc;
    int32_t total = 0;
    if (hipMemcpyAsync(&total, t.seq_off + B, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "text_forward_train: reading the packed row count failed");
    CMH_CHECK_ARG(total > 0 && total <= M, "text_forward_train: bad packed row count %d", total);
    seq_off = t.seq_off;
    rows = total;
    remember_tape_rows(tape, rows);
  }
  if ((rc = launch_text_embed_packed(tokens, w->token_embedding, w->positional_embedding, t.L[0].x_in, xh, t.rows, B, L, d,
                                     w->vocab_size, seq_off, st, pack_tokens))) return rc;
  // (the gathered attention rows are parked behind the first B rows of the h2 slot: the pooled tail needs 2 B rows of tape there)
  const bool tail = !tokens_out && train_pooled_tail() && rows >= 2 * B;      // the same rule in text_backward_impl
  for (int i = 0; i < w->layers; ++i) {
    void* nxt = i + 1 < w->layers ? t.L[i + 1].x_in : t.x_last;
    if ((rc = block_forward_train(w->blocks[i], dt, xh, t.L[i], nxt, B, L, d, 1, key_padding_mask, st, rows, seq_off,
                                  tail && i == w->layers - 1 ? t.rows : nullptr))) return rc;
  }
  if (tokens_out && pack_tokens) {
    if ((rc = tokens_head_forward(dt, xh, t, w->ln_final_w, w->ln_final_b, w->text_projection_t, t.tokP, rows, d, w->embed_dim, st))) return rc;
    return launch_unpack_token_rows(t.tokP, t.seq_off, tokens_out, B, L, w->embed_dim, t.rows, eot_rows_out, st);
  }
  if (tokens_out) {
    if (eot_rows_out && hipMemcpyAsync(eot_rows_out, t.rows, static_cast<size_t>(B) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(CMH_ERR_LAUNCH, "text_forward_train_tokens: eot row copy failed");
    if (hipMemsetAsync(t.seq_off, 0, static_cast<size_t>(B + 1) * 4, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "text_forward_train_tokens: memset failed");
    return tokens_head_forward(dt, xh, t, w->ln_final_w, w->ln_final_b, w->text_projection_t, tokens_out, M, d, w->embed_dim, st);
  }
  if ((rc = launch_layernorm_x(t.x_last, xh, tail ? nullptr : t.rows, w->ln_final_w, w->ln_final_b, t.pool, dt == CMH_BF16, B, d, st))) return rc;
  const int bk = dt == CMH_F32 ? 32 : 64;
  if (w->embed_dim % 128 == 0 && d % bk == 0)
    return launch_gemm(dt, t.pool, w->text_projection_t, nullptr, nullptr, feat, B, w->embed_dim, d, 0, st);
  return launch_small_linear(dt, t.pool, w->text_projection_t, nullptr, nullptr, 1.f, CMH_ACT_NONE, feat, B, w->embed_dim, d, st);
}

extern "C" int cmh_text_forward_train(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                      const uint8_t* key_padding_mask, float* feat, void* tape, size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(feat, "text_forward_train: null feature pointer");
  return text_forward_train_impl(w, tokens, batch, seq_len, key_padding_mask, feat, nullptr, nullptr, tape, tape_bytes, stream);
}

extern "C" int cmh_text_forward_train_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                             const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out, void* tape,
                                             size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "text_forward_train_tokens: null output pointer");
  return text_forward_train_impl(w, tokens, batch, seq_len, key_padding_mask, nullptr, tokens_out, eot_rows_out, tape, tape_bytes, stream);
}

extern "C" int cmh_text_forward_train_tokens_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                                    const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out, void* tape,
                                                    size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(tokens_out, "text_forward_train_tokens_packed: null output pointer");
  return text_forward_train_impl(w, tokens, batch, seq_len, key_padding_mask, nullptr, tokens_out, eot_rows_out, tape, tape_bytes, stream,
                                 /*pack_tokens_req=*/true);
}

static int text_backward_impl(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                              const uint8_t* key_padding_mask, const float* dfeat, const float* dtokens, const cmh_text_grads* gr,
                              void* tape, size_t tape_bytes, void* stream, int layer_hi = -1, int layer_lo = 0) {   // parts: see vit_backward_impl
  CMH_CHECK_ARG(w && tokens && (dfeat || dtokens) && gr && tape && batch > 0 && seq_len > 0, "text_backward: bad arguments");
  if (layer_hi < 0) layer_hi = w->layers;
  CMH_CHECK_ARG(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= w->layers, "text_backward: layers [%d, %d) of %d", layer_lo, layer_hi, w->layers);
  int rc = check_train_tower(w->gemm_dtype, w->width, w->layers, w->embed_dim, w->blocks);
  if (rc) return rc;
  CMH_CHECK_ARG(gr->token_embedding && gr->positional_embedding && gr->ln_final_w && gr->ln_final_b && gr->text_projection &&
                gr->blocks, "text_backward: null gradient pointer");
  if ((rc = check_block_grads(gr->blocks, w->layers))) return rc;
  if (tape_bytes < cmh_text_train_bytes(w, batch, seq_len)) return fail(CMH_ERR_WORKSPACE, "text_backward: tape too small");
  const int dt = w->gemm_dtype, d = w->width, B = batch, L = seq_len, M = B * L, E = w->embed_dim;
  const int xh = train_xh(dt, d);
  const size_t e = dt == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, static_cast<size_t>(M), B, d, e, xh ? 2 : 4, w->layers, 0, 0, E);
  const int32_t* seq_off = nullptr;
  int rows = M;
  if (!dtokens && text_packing(key_padding_mask)) {       // same rule as the forward call that filled this tape
    int32_t total = recall_tape_rows(tape);                // remembered by the forward call (no synchronisation here)
    if (total < 0) {                                       // a tape this process did not fill: read the count back
      if (hipMemcpyAsync(&total, t.seq_off + B, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(CMH_ERR_LAUNCH, "text_backward: reading the packed row count failed");
    }
    CMH_CHECK_ARG(total > 0 && total <= M, "text_backward: the tape holds no packed plan (row count %d)", total);
    seq_off = t.seq_off;
    rows = total;
  }
  if (dtokens) {       // a tape that cmh_text_forward_train_tokens_packed filled holds a packed plan: the forward call remembered its row
                       // count (0 for a dense tape, whose forward also zeroed seq_off); a tape this process does not remember is asked
    int32_t total = recall_tape_rows(tape);
    if (total < 0 && (hipMemcpyAsync(&total, t.seq_off + B, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
      return fail(CMH_ERR_LAUNCH, "text_backward_tokens: reading the packed row count failed");
    CMH_CHECK_ARG(total >= 0 && total <= M, "text_backward_tokens: the tape holds a bad packed row count %d", total);
    if (total > 0) { seq_off = t.seq_off; rows = total; }
  }
  const bool tail = !dtokens && train_pooled_tail() && rows >= 2 * B;
  if (layer_hi == w->layers) {
    if ((rc = zero_pad_buffers(t, static_cast<size_t>(rows), st))) return rc;
    if (dtokens) {
      const float* dtok = dtokens;
      if (seq_off) {   // the kept rows of the dense [B, L, E] gradient (the padded rows' gradient is zero by the caller's promise)
        if ((rc = launch_pack_token_rows(dtokens, seq_off, t.tokP, B, L, E, st))) return rc;
        dtok = t.tokP;
      }
      if ((rc = tokens_head_backward(dt, xh, t, w->ln_final_w, w->ln_final_b, w->text_projection_t, dtok, gr->text_projection,
                                     gr->ln_final_w, gr->ln_final_b, rows, d, E, st))) return rc;
    } else if ((rc = pooled_backward(dt, xh, t.x_last, t.rows, t.pool, w->text_projection_t, w->ln_final_w, dfeat, gr->text_projection,
                                     gr->ln_final_w, gr->ln_final_b, t, B, rows, d, E, st, tail))) return rc;
  }
  for (int i = layer_hi - 1; i >= layer_lo; --i) {
    const bool last = i == w->layers - 1;
    if ((rc = block_backward(w->blocks[i], grads_of(gr->blocks[i]), dt, xh, t.L[i], t, B, L, d, 1, key_padding_mask, st,
                             !last || dtokens != nullptr, rows, seq_off, tail && last ? t.rows : nullptr, tail && last ? t.dx2 : nullptr,
                             /*keep_f32=*/i == 0))) return rc;
  }
  if (layer_lo > 0) return CMH_OK;
  // x_0[b, t] = token_embedding[tokens[b, t]] + positional_embedding[t]
  if (hipMemsetAsync(gr->positional_embedding, 0, static_cast<size_t>(w->context_length) * d * 4, st) != hipSuccess ||
      hipMemsetAsync(gr->token_embedding, 0, static_cast<size_t>(w->vocab_size) * d * 4, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "text_backward: memset failed");
  if (seq_off) {
    hipLaunchKernelGGL(packed_dpos_kernel, dim3(L, (d + 255) / 256), dim3(256), 0, st, t.dx, seq_off, B, d, gr->positional_embedding);
    hipLaunchKernelGGL(packed_embed_scatter_kernel, dim3(M), dim3(256), 0, st, tokens, t.dx, seq_off, gr->token_embedding, B, L, d,
                       w->vocab_size);
  } else {
    if ((rc = cmh_colsum(t.dx, kF32, B, L * d, gr->positional_embedding, t.red, t.red_bytes, st))) return rc;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(M), dim3(256), 0, st, tokens, t.dx, gr->token_embedding, M, d, w->vocab_size);
  }
  CMH_CHECK_LAUNCH("embedding scatter");
  return CMH_OK;
}

extern "C" int cmh_text_backward(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                 const uint8_t* key_padding_mask, const float* dfeat, const cmh_text_grads* gr, void* tape,
                                 size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(dfeat, "text_backward: null gradient");
  return text_backward_impl(w, tokens, batch, seq_len, key_padding_mask, dfeat, nullptr, gr, tape, tape_bytes, stream);
}

extern "C" int cmh_text_backward_part(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                      const uint8_t* key_padding_mask, const float* dfeat, const cmh_text_grads* gr, void* tape,
                                      size_t tape_bytes, int32_t layer_hi, int32_t layer_lo, void* stream) {
  CMH_CHECK_ARG(dfeat && layer_hi >= 0, "text_backward_part: bad arguments");
  return text_backward_impl(w, tokens, batch, seq_len, key_padding_mask, dfeat, nullptr, gr, tape, tape_bytes, stream, layer_hi, layer_lo);
}

extern "C" int cmh_text_backward_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                        const uint8_t* key_padding_mask, const float* dtokens, const cmh_text_grads* gr, void* tape,
                                        size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(dtokens, "text_backward_tokens: null gradient");
  return text_backward_impl(w, tokens, batch, seq_len, key_padding_mask, nullptr, dtokens, gr, tape, tape_bytes, stream);
}


// ================================================================================================ a bare stack of blocks
// MITH's concept transformer (model/MITH.py:379-396: 2 ResidualAttentionBlocks over the K "concept tokens" of every sample, no
// mask) under training: f32 residual stream in and out, GEMM operands in `dtype`.
extern "C" size_t cmh_blocks_train_bytes(int32_t dtype, int32_t B, int32_t T, int32_t d, int32_t layers) {
  if (B <= 0 || T <= 0 || d <= 0 || layers <= 0) return 0;
  const size_t e = dtype == CMH_BF16 ? 2 : 4;
  return carve_train(nullptr, static_cast<size_t>(B) * T, B, d, e, 4, layers, 0, 0, 4).total;
}

extern "C" int cmh_blocks_forward_train(const cmh_block_weights* blocks, int32_t layers, int32_t dtype, const float* x, float* y,
                                        int32_t B, int32_t T, int32_t d, void* tape, size_t tape_bytes, void* stream) {
  CMH_CHECK_ARG(blocks && x && y && tape && layers > 0 && B > 0 && T > 0, "blocks_forward_train: bad arguments");
  int rc = check_train_tower(dtype, d, layers, 4, blocks);
  if (rc) return rc;
  if (tape_bytes < cmh_blocks_train_bytes(dtype, B, T, d, layers)) return fail(CMH_ERR_WORKSPACE, "blocks_forward_train: tape too small");
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(tape) & 255) == 0, "blocks_forward_train: tape must be 256-byte aligned");
  const size_t M = static_cast<size_t>(B) * T, e = dtype == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, M, B, d, e, 4, layers, 0, 0, 4);
  if (hipMemcpyAsync(t.L[0].x_in, x, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "blocks_forward_train: copy failed");
  for (int i = 0; i < layers; ++i) {
    void* nxt = i + 1 < layers ? t.L[i + 1].x_in : t.x_last;
    if ((rc = block_forward_train(blocks[i], dtype, /*xh=*/0, t.L[i], nxt, B, T, d, 0, nullptr, st))) return rc;
  }
  if (hipMemcpyAsync(y, t.x_last, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "blocks_forward_train: copy failed");
  return CMH_OK;
}

extern "C" int cmh_blocks_backward(const cmh_block_weights* blocks, const cmh_block_grads* grads, int32_t layers, int32_t dtype,
                                   const float* dy, float* dx, int32_t B, int32_t T, int32_t d, void* tape, size_t tape_bytes,
                                   void* stream) {
  CMH_CHECK_ARG(blocks && grads && dy && dx && tape && layers > 0 && B > 0 && T > 0, "blocks_backward: bad arguments");
  int rc = check_train_tower(dtype, d, layers, 4, blocks);
  if (rc) return rc;
  if ((rc = check_block_grads(grads, layers))) return rc;
  if (tape_bytes < cmh_blocks_train_bytes(dtype, B, T, d, layers)) return fail(CMH_ERR_WORKSPACE, "blocks_backward: tape too small");
  const size_t M = static_cast<size_t>(B) * T, e = dtype == CMH_BF16 ? 2 : 4;
  hipStream_t st = as_stream(stream);
  TrainBufs t = carve_train(tape, M, B, d, e, 4, layers, 0, 0, 4);
  if ((rc = zero_pad_buffers(t, M, st))) return rc;
  if (hipMemcpyAsync(t.dx, dy, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "blocks_backward: copy failed");
  for (int i = layers - 1; i >= 0; --i)
    if ((rc = block_backward(blocks[i], grads_of(grads[i]), dtype, /*xh=*/0, t.L[i], t, B, T, d, 0, nullptr, st, i != layers - 1, -1, nullptr,
                             nullptr, nullptr, /*keep_f32=*/i == 0))) return rc;
  if (hipMemcpyAsync(dx, t.dx, M * d * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "blocks_backward: copy failed");
  return CMH_OK;
}
