/*
 * cmh.h — C ABI of libcmh.so, the MI355X (gfx950) native hot path of CLIP-based cross-modal hashing.
 *
 * Drop-in boundary.  The reference (QinLab-WFU/CLIP-based-Cross-Modal-Hashing) is pure Python on
 * PyTorch and has no FFI of its own; these entry points are what a ctypes binding
 * for its hot path binds.  Each one names the reference interface it replaces (paths relative to
 * the reference root).  The Python mirror of the reference API that calls them lives in
 * clip-based-cross-modal-hashing_amd/ (model/modelbase.py, model/base/model.py, utils/calc_utils.py ...);
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch types.  All data pointers are DEVICE pointers unless
 *    a parameter says "host".  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - the library never allocates or frees DEVICE memory: scratch comes from the caller
 *    (`workspace`, size from the matching *_workspace_bytes()).  It owns two small page-locked HOST
 *    buffers, created on first use and kept for the life of the process: the staging ring of
 *    cmh_bert_adam_step's tensor tables and one 64-byte word through which the packed text
 *    encoders learn, without waiting, how many rows earlier calls packed.
 *  - the inference entry points (cmh_vit_encode*, cmh_text_encode*, heads, losses, codes, ranking)
 *    only enqueue kernels on `stream` and never wait for the device (graph-capturable).  The
 *    calls that DO wait for `stream` say so at their declaration: the fp8 calibration pass
 *    (cmh_*_calibrate_fp8), cmh_prof_gemm_end, the packed TRAINING forward of the text tower
 *    (cmh_text_forward_train reads the packed row count once: it sizes the tape's grids and the
 *    backward's split-K plans), and towers whose width is not a multiple of 256 when they take
 *    packed captions (test-sized configurations).
 *  - return value: 0 on success, negative cmh_status on error; cmh_last_error() gives the text
 *    (thread-local).  No exceptions cross the ABI.
 *  - matrices are dense row-major.  Linear weights are [out_features, in_features] exactly as
 *    torch.nn.Linear / nn.MultiheadAttention store them.
 *  - activations are token-major [B*T, d] (batch folded into rows), not the reference's [T,B,d];
 *    values are identical.
 */
#ifndef CMH_H_
#define CMH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMH_VERSION 6   /* = the round that last changed a struct layout or a signature; cmh_native.lib() refuses any other */

typedef enum cmh_status {
  CMH_OK = 0,
  CMH_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  CMH_ERR_WORKSPACE = -2,   /* workspace too small */
  CMH_ERR_LAUNCH = -3,      /* HIP launch / runtime error */
  CMH_ERR_DATA = -4         /* input data outside the domain (e.g. codes not in {-1,0,+1}) */
} cmh_status;

/* GEMM arithmetic of the encoder.  CMH_F32: f32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 FMA
 * chain; the parity mode against the fp32 reference, train/.../hash_train.py `self.model.float()`).
 * CMH_BF16: bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16; LayerNorm/softmax/residual
 * stream stay fp32. */
typedef enum cmh_dtype { CMH_F32 = 0, CMH_BF16 = 1, CMH_FP8 = 2 } cmh_dtype;
/* CMH_FP8 (BASELINE configs[4] "fp8 MFMA CLIP encoders"): the four GEMMs of every block (QKV, out_proj, c_fc, c_proj) take OCP
 * e4m3 operands on v_mfma_scale_f32_16x16x128_f8f6f4 (f32 accumulate): weights quantised per output channel, activations per
 * tensor with scales from a calibration pass; everything else as CMH_BF16 (fp16 residual stream, f32 LayerNorm statistics and
 * softmax; conv1 and the final projections stay bf16).  Needs width % 256 == 0.  See "fp8 encoder mode" below. */

const char* cmh_last_error(void);
int cmh_version(void);

/* ---------------------------------------------------------------------------------------------
 * Encoder weights.  Pointers borrow caller-owned device memory.  `*_w` GEMM weights are in the
 * dtype named by cmh_*_weights.gemm_dtype (f32 or bf16 copies made with cmh_cast_f32_to_bf16);
 * biases, LayerNorm parameters and embeddings are always f32.
 * One transformer block = reference model/base/model.py:167-196 ResidualAttentionBlock.
 * ------------------------------------------------------------------------------------------- */
typedef struct cmh_block_weights {
  const void* in_proj_w;    /* [3d, d]  attn.in_proj_weight */
  const float* in_proj_b;   /* [3d]     attn.in_proj_bias   */
  const void* out_proj_w;   /* [d, d]   attn.out_proj.weight */
  const float* out_proj_b;  /* [d] */
  const float* ln1_w;       /* [d] ln_1 */
  const float* ln1_b;
  const float* ln2_w;       /* [d] ln_2 */
  const float* ln2_b;
  const void* fc_w;         /* [4d, d]  mlp.c_fc.weight */
  const float* fc_b;        /* [4d] */
  const void* proj_w;       /* [d, 4d]  mlp.c_proj.weight */
  const float* proj_b;      /* [d] */
  /* CMH_FP8 only (ignored otherwise): the four `*_w` above are e4m3 bytes from cmh_fp8_quantize_weight */
  const float* in_proj_cs;  /* [3d] per-output-channel scales of in_proj_w */
  const float* out_proj_cs; /* [d] */
  const float* fc_cs;       /* [4d] */
  const float* proj_cs;     /* [d] */
  float act_scale[4];       /* per-tensor scales of the four GEMM inputs: ln_1 output, attention output, ln_2 output,
                             * QuickGELU(c_fc) output; x_fp8 = e4m3(x / act_scale).  From cmh_*_calibrate_fp8: headroom * amax / 448 */
} cmh_block_weights;

/* Image tower = reference model/base/model.py:210-252 VisionTransformer. */
typedef struct cmh_vit_weights {
  int32_t gemm_dtype;       /* cmh_dtype of every `const void*` weight below and in blocks[] */
  int32_t resolution;       /* input H = W (224) */
  int32_t patch;            /* 32 */
  int32_t width;            /* 768; heads = width/64 (model.py:284) */
  int32_t layers;           /* 12 */
  int32_t embed_dim;        /* 512 */
  const void* conv1_w;      /* [width, 3*patch*patch]  visual.conv1.weight flattened (c,py,px) */
  const float* class_embedding;      /* [width] */
  const float* positional_embedding; /* [grid*grid+1, width] */
  const float* ln_pre_w;  const float* ln_pre_b;
  const float* ln_post_w; const float* ln_post_b;
  const void* proj_t;       /* [embed_dim, width] = visual.proj TRANSPOSED (Linear layout) */
  const cmh_block_weights* blocks;   /* host array [layers] */
} cmh_vit_weights;

/* Text tower = reference model/base/model.py:359-372 CLIP.encode_text (+ :340-346 causal mask). */
typedef struct cmh_text_weights {
  int32_t gemm_dtype;
  int32_t context_length;   /* 77 */
  int32_t vocab_size;       /* 49408 */
  int32_t width;            /* 512; heads = width/64 (model.py:437) */
  int32_t layers;           /* 12 */
  int32_t embed_dim;        /* 512 */
  const float* token_embedding;      /* [vocab, width] */
  const float* positional_embedding; /* [context_length, width] */
  const float* ln_final_w; const float* ln_final_b;
  const void* text_projection_t;     /* [embed_dim, width] = text_projection TRANSPOSED */
  const cmh_block_weights* blocks;   /* host array [layers] */
} cmh_text_weights;

/* Optional taps for parity tests: when non-NULL the f32 residual stream [B*T, width] after
 * ln_pre (vision only, index 0) / after block i (index 1+i) is copied to taps[index]. */
typedef struct cmh_taps {
  float* const* ptrs;       /* host array of device pointers (entries may be NULL) */
  int32_t count;
} cmh_taps;

size_t cmh_vit_workspace_bytes(const cmh_vit_weights* w, int32_t batch);
size_t cmh_text_workspace_bytes(const cmh_text_weights* w, int32_t batch, int32_t seq_len);

/* encode_image: replaces CLIP.encode_image / VisionTransformer.forward (model/base/model.py:228-252,
 * :356-357).  image f32 [B,3,R,R] NCHW -> feat f32 [B, embed_dim]. */
int cmh_vit_encode(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat,
                   void* workspace, size_t workspace_bytes, const cmh_taps* taps, void* stream);

/* encode_text: replaces CLIP.encode_text (model/base/model.py:359-372).
 * tokens i64 [B, L] (L <= context_length) -> feat f32 [B, embed_dim]; the pooled row is
 * argmax(tokens[b]) (first maximum), as the reference takes it.
 * key_padding_mask (optional, u8 [B,L], 1 = ignore key) serves the MITH trunk (model/MITH.py:120-144). */
int cmh_text_encode(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                    const uint8_t* key_padding_mask, float* feat, void* workspace,
                    size_t workspace_bytes, const cmh_taps* taps, void* stream);

/* Calibration pass of the fp8 mode: the CMH_BF16 encode of this batch (`w` holds bf16 weights; feat as cmh_vit_encode /
 * cmh_text_encode_packed) which also records, per block i, the largest magnitude of the four GEMM inputs into
 * amax[4*i + {0: ln_1 output, 1: attention output, 2: ln_2 output, 3: QuickGELU(c_fc) output}] (device f32 [4*layers], running
 * maximum: zero it before the first batch).  The host turns it into cmh_block_weights.act_scale. */
int cmh_vit_calibrate_fp8(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, float* amax, void* workspace,
                          size_t workspace_bytes, void* stream);
int cmh_text_calibrate_fp8(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len, float* feat,
                           float* amax, void* workspace, size_t workspace_bytes, void* stream);

/* encode_text without the padding: under the causal mask (model/base/model.py:340-346) no token after a caption's EOT can
 * influence the EOT row that encode_text returns (:366-370), so only the tokens 0..EOT of every caption are computed, packed
 * into one matrix of <= batch*seq_len rows.  `feat` is bit-identical to cmh_text_encode's.  Nothing is synchronised: the packed row
 * count stays on the device - LayerNorm and the GEMM kernels read it themselves (their grids are sized for batch*seq_len rows) - and
 * rows_computed_dev (DEVICE int32 [1], may be NULL) receives a copy for the caller to read whenever it likes.  The GEMMs' tile
 * height is chosen for the count of an earlier call with the same (batch, seq_len), fetched without ever waiting; the first call
 * assumes batch*seq_len.  (Widths that are not a multiple of 256 - test-sized towers on the 128 x 128 fallback GEMMs - read the count
 * back once instead.)  Same workspace as cmh_text_encode. */
int cmh_text_encode_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len, float* feat,
                           int32_t* rows_computed_dev, void* workspace, size_t workspace_bytes, void* stream);
/* encode_image + encode_text of one batch with the two towers in lock-step (reference model/modelbase.py:105-108 calls them back to
 * back): layer i of both towers shares its GEMM launches (cmh_linear_gemm_grouped's kernel).  image f32 [batch,3,R,R], tokens i64
 * [batch, seq_len]; packed != 0: the text tower skips the positions after each caption's EOT like cmh_text_encode_packed
 * (rows_computed_dev as there, may be NULL).  Workspaces as for cmh_vit_encode / cmh_text_encode.  feat_image / feat_text
 * [batch, embed_dim] f32: bit-identical to the two single-tower calls.  Both towers must share gemm_dtype. */
int cmh_clip_encode_pair(const cmh_vit_weights* vw, const float* image, const cmh_text_weights* tw, const int64_t* tokens,
                         int32_t batch, int32_t seq_len, int32_t packed, float* feat_image, float* feat_text,
                         int32_t* rows_computed_dev, void* ws_image, size_t ws_image_bytes, void* ws_text, size_t ws_text_bytes,
                         void* stream);
/* The same for TWO loader batches in one call (the evaluation loop train/base.py:130-148 walks independent batches): the images of
 * the two batches stay two tensors (batch_a, batch_b rows), their captions come as one [batch_a + batch_b, seq_len] matrix; feat_image /
 * feat_text [batch_a + batch_b, embed_dim], batch a's rows first.  Same bits per row as cmh_clip_encode_pair on each batch; the
 * workspaces are sized for batch_a + batch_b. */
int cmh_clip_encode_pair2(const cmh_vit_weights* vw, const float* image_a, int32_t batch_a, const float* image_b, int32_t batch_b,
                          const cmh_text_weights* tw, const int64_t* tokens, int32_t seq_len, int32_t packed,
                          float* feat_image, float* feat_text, int32_t* rows_computed_dev, void* ws_image,
                          size_t ws_image_bytes, void* ws_text, size_t ws_text_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Building blocks of the towers, exported for unit-level parity tests and for heads that want them.
 * ------------------------------------------------------------------------------------------- */
#define CMH_EPI_BIAS 1        /* + bias[n] */
#define CMH_EPI_QUICKGELU 2   /* x*sigmoid(1.702x), model/base/model.py:162-164 */
#define CMH_EPI_RESIDUAL 4    /* + residual[m,n] (f32; may alias out) */
#define CMH_EPI_OUT_BF16 8    /* out is bf16 instead of f32 */
/* out[M,N] = epi(x[M,K] . w[N,K]^T): nn.Linear with the epilogue fused.  x,w in `dtype`.
 * Needs N % 128 == 0 and K % 32 (f32) / 64 (bf16) == 0. */
int cmh_linear_gemm(int32_t dtype, const void* x, const void* w, const float* bias, const float* residual,
                    void* out, int32_t M, int32_t N, int32_t K, int32_t epilogue, void* stream);
/* LayerNorm over the last dim of x f32 [M,d] (eps 1e-5, fp32 statistics; model/base/model.py:153-159). */
int cmh_layernorm(const float* x, const float* w, const float* b, void* out, int32_t out_dtype, int32_t M,
                  int32_t d, void* stream);
/* Two GEMMs of the same kind as ONE launch (round 4: grouped launches; csrc/gemm_wide.hip, template parameter GRP).  The image and
 * the text tower are 12 blocks of the same four Linear layers (reference model/base/model.py:167-207, instantiated twice by CLIP.__init__,
 * :254-306): layer i of both runs as one persistent grid that walks the tiles of the longer-K problem first and continues into the
 * other's.  Both problems share `dtype`, the output kind and `epilogue` (the flags of cmh_linear_gemm / cmh_linear_gemm_fp8); each
 * brings its own pointers, shape and - CMH_FP8 - scales.  m_dev (optional, device int32): the real row count when M is an upper bound
 * (packed captions).  Every output element is computed exactly as by the single launch: identical bits.  Shapes the wide kernel cannot
 * take together (N % 256, M <= 2048 rows) run as two plain launches - the result never depends on the grouping. */
typedef struct cmh_gemm_problem {
  const void* x;          /* [M, K] dtype */
  const void* w;          /* [N, K] dtype */
  const float* bias;      /* [N] or NULL */
  const void* residual;   /* [M, N] f32 / fp16 (CMH_EPI_RES_F16) or NULL */
  void* out;              /* [M, N] */
  int32_t M, N, K;
  const int32_t* m_dev;   /* NULL, or the device word holding the real row count (<= M) */
  const float* colscale;  /* CMH_FP8: [N] weight scales */
  float alpha;            /* CMH_FP8: activation scale of x */
  float out_scale;        /* CMH_FP8 with CMH_EPI_OUT_FP8: out = e4m3(clamp(v / out_scale, +-448)), as cmh_linear_gemm_fp8; > 0 */
} cmh_gemm_problem;
int cmh_linear_gemm_grouped(int32_t dtype, const cmh_gemm_problem* a, const cmh_gemm_problem* b, int32_t epilogue, void* stream);
/* on = 0: grouped requests run as two plain launches (A/B measurements, the equality tests); 1 = grouped; -1 = environment
 * (CMH_GEMM_GROUPED=0 is off).  Process-wide, not thread-safe. */
int cmh_set_gemm_grouped(int32_t on);
/* softmax(q k^T / 8 [+causal] [+key padding]) v per head (head dim 64) on packed qkv [B*T, 3d] -> o [B*T, d]. */
int cmh_attention(int32_t dtype, const void* qkv, void* o, int32_t B, int32_t T, int32_t d, int32_t causal,
                  const uint8_t* key_padding_mask, void* stream);

/* Measurement hook (bench.py `roofline`): while enabled, every GEMM launch of the library is timed with a pair of HIP events
 * on the launch stream.  The two hand-written GEMM kernels launch through hipExtLaunchKernelGGL, which stamps the pair with the
 * dispatch's own begin / end - the duration rocprofv3's kernel trace reports; the 128 x 128 fallback kernels are bracketed by two
 * recorded events (CMH_GEMM_PROF_BRACKET=1: every launch, round 1-2's method, which adds the marker packets' gaps to each launch).
 * _end() synchronises on those events and returns the summed launch durations, the summed algorithmic FLOPs (2*M*N*K, real rows
 * only) and the launch count over ALL GEMM launches; _by_kernel() (after _end) splits the same sums by kernel:
 * [0] gemm_wide_kernel (N % 256 == 0: every large GEMM), [1] gemm_rows_kernel (M <= 512: the pooled-row tail), [2] the fallbacks.
 * Not thread-safe; single-stream benchmarking only. */
int cmh_prof_gemm_begin(int32_t max_launches);
int cmh_prof_gemm_end(double* total_ms, double* total_flops, int64_t* launches);
int cmh_prof_gemm_by_kernel(double* ms3, double* flops3, int64_t* launches3);

/* Tuning overrides of the N % 256 == 0 GEMM kernel, for A/B measurements and for tests that must reach every variant:
 * tile_rows in {96, 128, 160} pins the tile height (-1: chosen per launch from M, N, K and the CU count); order_group picks the
 * tile order: -1 chosen per launch (the n-blocked order for N >= 1024: panel blocks outermost, so an XCD keeps its block of W in
 * its L2), 0 plain n-fastest, 1..64 n-panels per group inside each XCD's band (round 2), -2..-8 the n-blocked order with that many
 * blocks.  Results do not depend on either.
 * Process-wide, not thread-safe.  (No reference counterpart: upstream's GEMMs are ATen's, model/base/model.py:167-196.) */
int cmh_gemm_tuning(int32_t tile_rows, int32_t order_group);
/* GEMMs of few rows (M <= 512: the pooled-row tail of the towers, small heads) run on 64 x 64 tiles (csrc/gemm_rows.hip) instead of
 * the wide kernel's 96..160 x 256 ones: same bits per output element (same MFMA chain over K, same epilogue order), more workgroups.
 * on = 0 sends them to the wide kernel again (A/B, tests), 1 forces the default, -1 = environment (CMH_GEMM_ROWS=0 is off). */
int cmh_set_gemm_rows(int32_t on);
/* Round 5: the loader / consumer form of the N % 256 == 0 GEMM (csrc/gemm_lc.hip: four waves of a workgroup only stage operands by
 * LDS-DMA, the other four only multiply; 128 x 256 tiles; bf16 operands with 16-bit outputs and the forward epilogues of a transformer
 * block: bias, + QuickGELU, + fp16 residual; plain and grouped launches).  Same bits per output element as the wide kernel (same MFMA
 * chain over K, same epilogue order).  mode 0: never; 1: every launch it can take; 2: every such launch without QuickGELU;
 * 3: where the host's cost model expects it to be faster; 4 / 5 / 6: the 12-wave form (4 staging + 8 MFMA waves, stores deferred) for
 * every block launch with K >= 512 / for those without QuickGELU / for QKV and out_proj only, the wide kernel elsewhere; 7: the form on e4m3 operands (fp8 QKV launches); -1: the environment's
 * CMH_GEMM_LC (default).  Process-wide, not thread-safe. */
int cmh_set_gemm_lc(int32_t mode);

/* encode_image / encode_text return one pooled row per sample (model/base/model.py:247-250, 366-370), and past the last block's
 * attention every operation is row-wise, so cmh_vit_encode / cmh_text_encode[_packed] carry only those B rows through the last
 * block's out_proj, ln_2 and MLP (bit-identical features, three GEMMs of M = B instead of M = B*T).  on = 0 switches that off
 * (A/B measurements, the equality test); also CMH_POOLED_TAIL=0 in the environment.  Process-wide, not thread-safe. */
int cmh_set_pooled_tail(int32_t on);

/* bf16 training mode: the residual-gradient stream between the blocks of a tower's backward pass (cmh_vit_backward[_part],
 * cmh_text_backward[_part], cmh_blocks_backward; reference: autograd through model/base/model.py:167-207) is carried as bf16 - the
 * copy the LayerNorm backward kernels leave as the next GEMMs' operand anyway - instead of f32 beside that copy: each LayerNorm
 * backward forms (new dx + stream) in f32 and rounds once.  on = 0 keeps the f32 stream (also CMH_GRAD_STREAM16=0); -1: the
 * environment (default on).  The f32 mode never uses it.  Process-wide, not thread-safe. */
int cmh_set_grad_stream16(int32_t on);

/* cmh_text_encode_tokens / cmh_text_forward_train_tokens with a key_padding_mask (the MITH trunk, model/MITH.py:120-144): positions
 * behind a caption's last unpadded token are never read by the reference's HashingModel (LocalizedTokenAggregation gives them weight
 * 0, model/MITH.py:349-376), so only the rows up to that position run through the tower; the projected tokens are written back to
 * their dense [B, L, E] places with ZEROS in the padded positions (the reference computes finite values there that nothing reads), the
 * EOT rows are dense indices as before.  Kept positions: the same bits as the dense path.  on = 0: every position is computed
 * (also CMH_TEXT_PACK_TOKENS=0); -1: the environment (default on).  Process-wide, not thread-safe. */
int cmh_set_text_token_packing(int32_t on);

/* ---------------------------------------------------------------------------------------------
 * fp8 encoder mode (CMH_FP8).  Mirrors the reference's precision hook convert_weights (model/base/model.py:391-412): the same
 * tensors it lowers to fp16 - Linear / MultiheadAttention weights - go to e4m3 here, with the activations that feed them.
 * ------------------------------------------------------------------------------------------- */
#define CMH_EPI_OUT_FP8 4096  /* cmh_linear_gemm_fp8: out is e4m3 of clamp(v / out_scale, +-448) */
#define CMH_FP8_MAX 448.0f

/* w f32 [N,K] -> w_fp8 e4m3 [N,K] and colscale f32 [N] = amax_k|w[n,k]| / 448 (1 for an all-zero row): w ~ w_fp8 * colscale[n]. */
int cmh_fp8_quantize_weight(const float* w, void* w_fp8, float* colscale, int32_t N, int32_t K, void* stream);
/* x (kind 0 f32 / 1 bf16 / 2 fp16, n % 4 == 0 elements) -> x_fp8 = e4m3(clamp(x / scale)); and back: out = f32(x_fp8) * scale. */
int cmh_fp8_quantize(const void* x, int32_t kind, void* x_fp8, int64_t n, float scale, void* stream);
int cmh_fp8_dequantize(const void* x_fp8, float* out, int64_t n, float scale, void* stream);
/* amax_inout[0] = max(amax_inout[0], max|x|) (device scalar, zero it first; a NaN in x reports +inf): calibration of the
 * per-tensor activation scales. */
int cmh_amax(const void* x, int32_t kind, int64_t n, float* amax_inout, void* stream);
/* out[M,N] = epi( alpha * colscale[n] * (x_fp8[M,K] . w_fp8[N,K]^T) ), N % 256 == 0, K % 128 == 0.  `epilogue` takes CMH_EPI_BIAS,
 * _QUICKGELU, _GELU, _RELU, _RESIDUAL (+ _RES_F16) and ONE of _OUT_BF16 / _OUT_F16 / _OUT_FP8 (default f32).  alpha = the
 * activation tensor's scale, colscale = cmh_fp8_quantize_weight's. */
int cmh_linear_gemm_fp8(const void* x_fp8, const void* w_fp8, const float* colscale, float alpha, const float* bias,
                        const float* residual, void* out, float out_scale, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                        void* stream);

/* DNpH quadratic spherical mutual information loss, reference train/DNpH_TMM/loss.py:5-72 `qmi_loss(images, texts, targets)` in
 * its default form (use_cosine, use_square_clamp, M = B^2 / sum(D)): img, txt f32 [B,K] hash outputs, labels bit-packed as by
 * cmh_pack_labels ([B, ceil(C/32)] u32) -> loss f32 [1] and sum_d f32 [1] (the number of label-sharing pairs, kept for the
 * backward).  Backward: dimg, dtxt f32 [B,K] = dloss[0] * d loss / d img, txt.  K <= 1024, C <= 512. */
size_t cmh_qmi_workspace_bytes(int32_t B);
int cmh_qmi_loss(const float* img, const float* txt, const uint32_t* labels_packed, int32_t B, int32_t K, int32_t C, float eps,
                 float* loss, float* sum_d, void* workspace, size_t workspace_bytes, void* stream);
int cmh_qmi_loss_backward(const float* img, const float* txt, const uint32_t* labels_packed, int32_t B, int32_t K, int32_t C,
                          float eps, const float* sum_d, const float* dloss, float* dimg, float* dtxt, void* workspace,
                          size_t workspace_bytes, void* stream);

/* DMsH-LN multi-similarity loss, reference train/DMsH_LN/MSLOSS.py:13-55 `MultiSimilarityLoss.forward(feats, labels, feat2)` in the
 * branch its trainer takes (train/DMsH_LN/hash_train.py:58-60; not the "cifar10-1" branch): feats (and feat2, NULL = feats) f32
 * [B,K] hash outputs, labels f32 [B,Kl] = the LabelNet codes (two samples are similar when their codes' dot product is > 0)
 * -> loss f32 [1]; thresh 0.5, margin 0.1, scales 2 / 40 as the reference fixes them.  Backward: dfeats (and dfeat2 when feat2 is
 * given; with feat2 = NULL both roles' gradients are summed into dfeats) = dloss[0] (NULL: 1) * d loss / d feats; it recomputes
 * the forward's statistics, so the workspace need not be kept between the two calls.  B <= 8192 (the global batch under data parallelism: B * 4 bytes of
 * dynamic LDS per backward launch, two [B, B] f32 matrices of workspace), K, Kl <= 1024. */
size_t cmh_msl_workspace_bytes(int32_t B);
int cmh_msl_loss(const float* feats, const float* feat2, const float* labels, int32_t B, int32_t K, int32_t Kl, float* loss,
                 void* workspace, size_t workspace_bytes, void* stream);
int cmh_msl_loss_backward(const float* feats, const float* feat2, const float* labels, int32_t B, int32_t K, int32_t Kl,
                          const float* dloss, float* dfeats, float* dfeat2, void* workspace, size_t workspace_bytes, void* stream);

/* DHaPH self-paced contrastive loss, reference train/DHaPH/MSLoss.py:13-33 `MSLoss.forward(image_feature, text_feature, labels,
 * epoch)`: a (and b, NULL = a: the image-image / text-text calls of train/DHaPH/hash_train.py:68-69) f32 [B,K] hash outputs, labels
 * f32 [B,C] multi-hot -> loss f32 [1] = mean_i -log(P_i / (P_i + N_i)) over exp(cos / temperature) of the label-sharing / other
 * pairs, with the detached self-paced weights exp(-1 - cos)^(delta/4) / exp(-1 + cos)^delta; delta in [0, 1] is what :23-27 derive
 * from (epoch, totalepoch) (0 = self_paced off).  Backward: da (and db; with b = NULL both roles' gradients are summed into da) =
 * dloss[0] (NULL: 1) * d loss / d a; it recomputes the forward's statistics.  B <= 8192 (as above), K, C <= 1024. */
size_t cmh_spl_workspace_bytes(int32_t B);
int cmh_spl_loss(const float* a, const float* b, const float* labels, int32_t B, int32_t K, int32_t C, float temperature, float delta,
                 float* loss, void* workspace, size_t workspace_bytes, void* stream);
int cmh_spl_loss_backward(const float* a, const float* b, const float* labels, int32_t B, int32_t K, int32_t C, float temperature,
                          float delta, const float* dloss, float* da, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* f32 -> bf16 (round-to-nearest-even) copy used to prepare CMH_BF16 GEMM weights. */
int cmh_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hash heads and code generation.
 * ------------------------------------------------------------------------------------------- */
typedef enum cmh_act { CMH_ACT_NONE = 0, CMH_ACT_TANH = 1, CMH_ACT_RELU = 2 } cmh_act;

/* y[M,N] = act( scale_mask ? (x W^T + b) * drop_mask/(1-p) : x W^T + b ), all f32.
 * Replaces LinearHash.forward (model/modelbase.py:25-35: fc -> dropout(0.2) -> tanh; drop_mask
 * NULL = eval), HashLayer.fc (model/DCHMT.py:13,20-21), Pre_Layer (model/DNPH_TOMM.py:7-14).
 * drop_mask: optional f32 [M,N] of {0,1}; keep_scale = 1/(1-p). */
int cmh_linear_act(const float* x, const float* w, const float* b, const float* drop_mask,
                   float keep_scale, int32_t act, float* y, int32_t M, int32_t N, int32_t K,
                   void* stream);

/* DCHMT select head: softmax over each adjacent pair of z[M, 2K] -> p[M, 2K]
 * (model/DCHMT.py:24 torch.softmax(item(embed), -1) for the K two-way Linears, concatenated as the
 * trainer does at train/DCHMT/hash_train.py:55-57). */
int cmh_pair_softmax(const float* z, float* p, int32_t M, int32_t K, void* stream);

/* codes = sign(h) in {-1,0,+1} f32 (train/base.py:141-143 torch.sign). */
int cmh_sign_codes(const float* h, float* codes, int64_t n, void* stream);

/* codes[M,K] from pair probabilities p[M,2K]: argmax of the pair, index 0 -> -1, 1 -> +1, tie -> -1
 * (train/base.py:150-158 make_hash_code_DCHMT). */
int cmh_pair_argmax_codes(const float* p, float* codes, int32_t M, int32_t K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hamming ranking / mAP  (reference utils/calc_utils.py).
 * Codes are packed to two bit-planes of W = ceil(K/32) u32 words per code:
 *   sign plane bit t = (code[t] > 0), nz plane bit t = (code[t] != 0).
 * Labels are packed to ceil(C/32) u32 words, bit c = (label[c] != 0).
 * ------------------------------------------------------------------------------------------- */
/* bad_flag (device i32, caller zeroes it) is set to 1 if any code is not exactly -1, 0 or +1. */
int cmh_pack_codes(const float* codes, int64_t n, int32_t bits, uint32_t* sign_plane,
                   uint32_t* nz_plane, int32_t* bad_flag, void* stream);
/* The inverse of cmh_pack_codes: codes f32 [n, bits] in {-1, 0, +1} from the two bit planes.  Data-parallel evaluation exchanges the
 * PLANES (16 bytes per 64-bit code pair instead of 512 bytes of floats: SURVEY 8e "all-gather packed DB codes") and restores the
 * reference's float code matrices (train/base.py:130-148 img_buffer / text_buffer) on every rank with this call. */
int cmh_unpack_codes(const uint32_t* sign_plane, const uint32_t* nz_plane, int64_t n, int32_t bits, float* codes, void* stream);
/* bad_flag set if any label is negative or NaN (the reference's `L.L^T > 0` test then stops
 * being an intersection test). */
int cmh_pack_labels(const float* labels, int64_t n, int32_t classes, uint32_t* packed,
                    int32_t* bad_flag, void* stream);

/* calc_hammingDist (utils/calc_utils.py:8-13): dist[Q,N] f32 = 0.5*(K - q.r), exact for codes in
 * {-1,0,+1}, computed by AND/XOR + popcount on the packed planes. */
int cmh_hamming_dist(const uint32_t* q_sign, const uint32_t* q_nz, const uint32_t* r_sign,
                     const uint32_t* r_nz, int32_t Q, int64_t N, int32_t bits, float* dist,
                     void* stream);

/* calc_neighbor (utils/calc_utils.py:42-45): sim[A,B] f32 = (la . lb^T > 0). */
int cmh_calc_neighbor(const uint32_t* la, const uint32_t* lb, int32_t A, int32_t B, int32_t classes,
                      float* sim, void* stream);

typedef enum cmh_tie_order {
  CMH_TIE_REFERENCE = 0,    /* torch.sort(stable=False) on CPU == libstdc++ introsort order */
  CMH_TIE_STABLE = 1        /* ties by ascending database index (torch.sort(stable=True)): LSD radix sort, ~4x faster */
} cmh_tie_order;

size_t cmh_map_workspace_bytes(int32_t Q, int64_t N, int32_t bits, int32_t tie_order);

/* calc_map_k_matrix (utils/calc_utils.py:16-39).  topk <= 0 means k = N (the reference's k=None).
 *   ap[Q]   f32  per-query AP (0 for queries with no relevant item; they still count in the mean)
 *   map[1]  f32  sum(ap)/Q accumulated in f32 in query order like the reference's `map += ...`
 *   perm    optional i32 [Q,N]: the full ranking `ind` (utils/calc_utils.py:31) per query
 * depth_limit_override < 0 : libstdc++'s 2*floor(log2 N); >= 0 forces the introsort depth limit
 * (test hook for the heapsort fallback). */
int cmh_hamming_map(const uint32_t* q_sign, const uint32_t* q_nz, const uint32_t* q_label,
                    const uint32_t* r_sign, const uint32_t* r_nz, const uint32_t* r_label,
                    int32_t Q, int64_t N, int32_t bits, int32_t classes, int64_t topk,
                    int32_t tie_order, int32_t depth_limit_override, float* ap, float* map,
                    int32_t* perm, void* workspace, size_t workspace_bytes, void* stream);
/* map[1] = (((ap[0] + ap[1]) + ...) / Q), f32, in query order: the mean cmh_hamming_map itself appends, exposed for per-query APs
 * that were computed on query shards by several ranks and gathered (utils/calc_utils.py:37-38 `map += AP`; `map / num_query`). */
int cmh_map_mean(const float* ap, int32_t Q, float* map, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pairwise similarity / quantisation losses (forward).  All f32; `loss` is a device scalar.
 * ------------------------------------------------------------------------------------------- */
size_t cmh_loss_workspace_bytes(int32_t B, int32_t K, int32_t C);

/* DSPH HyP.forward (train/DSPH/loss.py:22-72).  x,y [B,K] hash outputs, label [B,C] in {0,1},
 * proxies [C,K]. */
int cmh_dsph_hyp_loss(const float* x, const float* y, const float* label, const float* proxies,
                      int32_t B, int32_t K, int32_t C, float threshold, float alpha, float* loss,
                      void* workspace, size_t workspace_bytes, void* stream);

/* DCHMT our_loss with hash_layer == "select" (train/DCHMT/hash_train.py:82-150).
 * img,txt [B,D] (D = 2*output_dim), label [B,C].  similarity: 0 = euclidean, 1 = cosine;
 * loss_type: 1 = l1, 2 = l2. */
int cmh_dchmt_loss(const float* img, const float* txt, const float* label, int32_t B, int32_t D,
                   int32_t C, int32_t output_dim, int32_t similarity, int32_t loss_type,
                   float vartheta, float sim_threshold, float* loss, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DNPH (TOMM) and TwDH heads / losses (BASELINE.json configs 4 and 5).
 * ------------------------------------------------------------------------------------------- */
/* nn.BatchNorm1d in TRAINING mode (batch mean, biased batch variance): the TwDH image head's `norm`
 * (model/TwDH.py:61,78), which the reference never switches to eval.  x,y f32 [B,d]. */
int cmh_batchnorm1d_train(const float* x, const float* w, const float* b, float eps, float* y, int32_t B,
                          int32_t d, void* stream);

/* The module side effect of that forward: running_mean / running_var (momentum 0.1, unbiased variance), part of the checkpoint. */
int cmh_batchnorm1d_update_running(const float* x, float momentum, float* running_mean, float* running_var, int32_t B,
                                   int32_t d, void* stream);
/* Backward of cmh_batchnorm1d_train: dx [B,d], dw [d], db [d] from dy [B,d] (statistics recomputed from x). */
int cmh_batchnorm1d_backward(const float* x, const float* w, float eps, const float* dy, float* dx, float* dw, float* db,
                             int32_t B, int32_t d, void* stream);
/* Backward of cmh_twdh_loss: dp_img / dp_txt f32 [B,2K] from the upstream gradients of the two scalars (device pointers to one
 * float each; NULL = 0).  BCELoss as ATen differentiates it: (p - t) / max((1 - p) p, 1e-12) / numel. */
int cmh_twdh_loss_backward(const float* p_img, const float* p_txt, const float* target, int32_t B, int32_t K,
                           const float* d_nce, const float* d_quan, float* dp_img, float* dp_txt, void* stream);

/* hash_center_multilables (train/TwDH/hash_train.py:93-115): code[b,k] = sign(mean of the centers of b's
 * classes), exact zeros replaced by random_center[k] (a +-1 vector drawn once per call by the caller).
 * label f32 [B,C], center f32 [C,K] (+-1), code f32 [B,K]. */
int cmh_twdh_targets(const float* label, const float* center, const float* random_center, float* code,
                     int32_t B, int32_t C, int32_t K, void* stream);

/* TwDH loss terms for one code length (train/TwDH/hash_train.py:117-139): p_img/p_txt f32 [B,2K] pair
 * probabilities, target f32 [B,K] in {-1,+1}.  out2[0] = (BCE_img + BCE_txt)/2 with hash_convert one-hot
 * targets, out2[1] = ((1-mean((2p_img-1)^2)) + (1-mean((2p_txt-1)^2)))/2.  workspace >= 256 bytes. */
int cmh_twdh_loss(const float* p_img, const float* p_txt, const float* target, int32_t B, int32_t K,
                  float* out2, void* workspace, size_t workspace_bytes, void* stream);

/* DNPH_out.forward + the noise term of the DNPH step (train/DNPH_TOMM/loss.py:14-32,
 * train/DNPH_TOMM/hash_train.py:70-81).  hash_* f32 [B,K], pre_* f32 [B,C], label f32 [B,C], proxies [C,K],
 * noise_* f32 [B,K] (the Hungarian-assigned +-1 rows, host-side as upstream) or both NULL.
 * out3[0] = p_loss + d_loss - noise_weight*noise, out3[1] = p_loss + d_loss, out3[2] = noise. */
int cmh_dnph_loss(const float* hash_img, const float* hash_txt, const float* pre_img, const float* pre_txt,
                  const float* label, const float* proxies, const float* noise_img, const float* noise_txt,
                  int32_t B, int32_t K, int32_t C, float margin, float noise_weight, float* out3,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MITH (BASELINE.json config 3): token-returning trunk, HashingModel pieces, losses.
 * Activations are batch-major: [N, K, D] where the reference holds [K, N, D].
 * ------------------------------------------------------------------------------------------- */
#define CMH_EPI_GELU 16       /* exact GELU (nn.GELU()) in cmh_linear_gemm's epilogue, model/MITH.py:224-233 */
#define CMH_EPI_RELU 32
#define CMH_EPI_RES_F16 64    /* residual is IEEE fp16 (the bf16 mode's residual stream); needs N % 256 == 0 */
#define CMH_EPI_OUT_F16 128   /* out is fp16 instead of f32; needs N % 256 == 0, excludes CMH_EPI_OUT_BF16 */
#define CMH_EPI_MUL_DQGELU 1024  /* out *= QuickGELU'(aux[m,n]), aux = the saved pre-activation passed as `residual`, typed like
                                  * the output (bf16 with CMH_EPI_OUT_BF16, else f32); N % 256 == 0; excludes CMH_EPI_RESIDUAL */

/* ViT.forward of the MITH trunk (model/MITH.py:56-82): ln_post + proj on EVERY token.
 * tokens_out f32 [B*(g*g+1), embed_dim]; row b*(g*g+1) is the class token, the rest the patch tokens. */
int cmh_vit_encode_tokens(const cmh_vit_weights* w, const float* image, int32_t batch, float* tokens_out,
                          void* workspace, size_t workspace_bytes, void* stream);
/* CLIP1.encode_text (model/MITH.py:120-144): causal mask + key_padding_mask in every block, ln_final +
 * text_projection on every token.  tokens_out f32 [B*L, embed_dim]; eot_rows_out i32 [B] = b*L + argmax(tokens[b]). */
int cmh_text_encode_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                           const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same, with the caller's promise that nothing reads the padded positions of tokens_out - MITH's HashingModel gives them weight 0
 * (model/MITH.py:349-376): rows behind a caption's last unpadded token are not computed and come back as ZEROS; every other position
 * carries the bits of cmh_text_encode_tokens.  key_padding_mask NULL, or cmh_set_text_token_packing(0): exactly cmh_text_encode_tokens. */
int cmh_text_encode_tokens_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                  const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* `layers` ResidualAttentionBlocks applied in place to a caller-owned residual stream x f32 [B*T, d]
 * (MITH's 2-layer concept transformer, model/MITH.py:385-395). */
size_t cmh_blocks_workspace_bytes(int32_t dtype, int32_t B, int32_t T, int32_t d);
int cmh_transformer_blocks(const cmh_block_weights* blocks, int32_t layers, int32_t dtype, float* x, int32_t B,
                           int32_t T, int32_t d, int32_t causal, const uint8_t* key_padding_mask,
                           void* workspace, size_t workspace_bytes, void* stream);

/* LocalizedTokenAggregation.forward (model/MITH.py:317-376).  tokens f32 [B*Ltot, D], sim f32 [B*Ltot, K] (tanh
 * concept scores of every token); the L tokens l0..l0+L-1 of each sample take part (image: Ltot=50, l0=1, L=49).
 * key_padding_mask u8 [B, L] optional.  out f32 [B, K, D].  L <= 80, K <= 128, top_k <= 8. */
int cmh_mith_lta(const float* tokens, const float* sim, const uint8_t* key_padding_mask, float* out, int32_t B,
                 int32_t Ltot, int32_t l0, int32_t L, int32_t K, int32_t D, int32_t top_k, void* stream);
/* x[b,t,:] += pe[t,:]   (PositionalEncoding.forward, model/MITH.py:270-273; pe is the registered buffer). */
int cmh_add_positional(float* x, const float* pe, int32_t B, int32_t T, int32_t D, void* stream);
/* BitwiseHashing (model/MITH.py:276-293): out[b,k] = tanh(x[b,k,:] . w[k,:] + bias[k]). */
int cmh_bitwise_hash(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t K,
                     int32_t D, void* stream);
/* F.normalize(x, dim=-1) (eps 1e-12) over rows of x f32 [R, D]. */
int cmh_l2_normalize_rows(const float* x, float* y, int32_t R, int32_t D, void* stream);
/* B = sign(lambda*(img_cls+txt_cls) + (1-lambda)*(img_tok+txt_tok)); H_img = .5 img_cls + .5 img_tok; H_txt likewise
 * (train/MITH/hash_train.py:80-83,179-180). */
int cmh_mith_mix(const float* img_cls, const float* img_tok, const float* txt_cls, const float* txt_tok,
                 float hyper_lambda, float* B_codes, float* H_img, float* H_txt, int64_t n, void* stream);
/* out[0] = sum (a-b)^2   (F.mse_loss(reduction='sum'): quantisation and distillation terms, :146-147,193-200). */
int cmh_sq_diff_sum(const float* a, const float* b, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                    void* stream);
/* bayesian_loss (train/MITH/hash_train.py:138-144) of a memory bank [Mb,K] against the batch codes [B,K] with
 * label_sim = (bank_label . label^T > 0) computed on the fly (bank_label f32 [Mb,C], label f32 [B,C]). */
int cmh_mith_bayesian_loss(const float* bank, const float* batch, const float* bank_label, const float* label,
                           int32_t Mb, int32_t B, int32_t K, int32_t C, float* out, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Symmetric InfoNCE with diagonal targets inside groups of G consecutive rows: G = R reproduces info_nce_loss
 * (:103-114), a/b = [N*L, D] with G = L reproduces info_nce_loss_bmm (:116-136). */
int cmh_info_nce(const float* a, const float* b, int32_t R, int32_t G, int32_t D, float temperature, float* out,
                 void* workspace, size_t workspace_bytes, void* stream);
/* Workspace that lets cmh_info_nce / cmh_info_nce_backward form the R x G score matrix once, as 16 x 16 tiles on the f32 matrix
 * pipe (round 3; needs D % 64 == 0).  With a smaller workspace (>= 256 bytes forward, >= 8 R + 256 backward) both calls keep the
 * per-row kernels of round 1: same values up to f32 summation order. */
size_t cmh_info_nce_workspace_bytes(int32_t R, int32_t G);

/* ---------------------------------------------------------------------------------------------
 * BertAdam, one fused multi-tensor step (model/base/optimization.py:103-168; SURVEY 8f "next" #1).
 * `tensors` is a HOST array; p/g/m/v are device pointers to f32 tensors of n elements (m = state['next_m'],
 * v = state['next_v']).  Per tensor: g *= min(1, max_grad_norm/(||g||+1e-6)) in place when max_grad_norm > 0 (:135-136);
 * m = m*b1 + (1-b1)*g (:141); v = v*b2 + (1-b2)*g*g (:143); p -= lr * (m/(sqrt(v)+eps) + weight_decay*p) (:144-164),
 * with lr = the caller's already scheduled rate (lr * schedule(step/t_total, warmup), :157-161).
 * Synchronises `stream` once on entry (it re-uses a host staging table).
 * ------------------------------------------------------------------------------------------- */
typedef struct cmh_adam_tensor {
  float* p; float* g; float* m; float* v;
  int64_t n;
  float lr, weight_decay, max_grad_norm;
  void* p_bf16;             /* optional (NULL): bf16 [n] copy of p - the encoder's CMH_BF16 GEMM operand - rewritten with the updated
                             * values (round-to-nearest-even, as cmh_cast_f32_to_bf16), so that no cast pass follows the step */
} cmh_adam_tensor;
size_t cmh_bert_adam_workspace_bytes(int32_t count, int64_t total_elems);
/* b1 / b2 / eps are doubles because the reference forms 1 - b in python doubles before ATen rounds it to f32. */
int cmh_bert_adam_step(const cmh_adam_tensor* tensors, int32_t count, double b1, double b2, double eps,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward building blocks of the towers (SURVEY 8f "next" #2).  Element kinds: 0 = f32, 1 = bf16, 2 = fp16.
 * ------------------------------------------------------------------------------------------- */
#define CMH_KIND_F32 0
#define CMH_KIND_BF16 1
#define CMH_KIND_F16 2
/* dst[c, r] = cast(src[r, c]): wgrad operands (dW = dY^T . X contracts over rows). */
int cmh_transpose(const void* src, int32_t src_kind, void* dst, int32_t dst_kind, int32_t rows, int32_t cols, void* stream);
/* out[c] = sum_r x[r, c] (bias gradients), f32 result, deterministic two-pass reduction. */
size_t cmh_colsum_workspace_bytes(int32_t rows, int32_t cols);
int cmh_colsum(const void* x, int32_t kind, int32_t rows, int32_t cols, float* out, void* workspace, size_t workspace_bytes,
               void* stream);
/* nn.LayerNorm backward (model/base/model.py:153-159, eps 1e-5): x f32|fp16 [M,d] (the forward's input), dy f32|bf16 [M,d],
 * dx f32 [M,d] (accumulate != 0: dx += ...), dgamma / dbeta f32 [d]. */
size_t cmh_layernorm_backward_workspace_bytes(int32_t M, int32_t d);
int cmh_layernorm_backward(const void* x, int32_t x_kind, const void* dy, int32_t dy_kind, const float* gamma, int32_t M,
                           int32_t d, float* dx, int32_t accumulate, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Backward of cmh_attention (same layouts and masks): qkv [B*T,3d], o = the forward output [B*T,d], dout [B*T,d] ->
 * dqkv [B*T,3d]; softmax probabilities are recomputed.  T <= 128. */
int cmh_attention_backward(int32_t dtype, const void* qkv, const void* o, const void* dout, void* dqkv, int32_t B, int32_t T,
                           int32_t d, int32_t causal, const uint8_t* key_padding_mask, void* stream);
/* out = pre * sigmoid(1.702 pre) element-wise from a saved pre-activation (model/base/model.py:162-164). */
int cmh_quick_gelu(const void* pre, void* out, int64_t n, int32_t kind, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training forward (keeps a tape of activations) and backward of the two towers.  Gradients are WRITTEN (not accumulated) into
 * caller-owned f32 buffers shaped like the reference's parameters (model/base/model.py; note `proj` / `text_projection` are in
 * the parameter's own [width, embed_dim] layout, not transposed).  `tape` (cmh_*_train_bytes) holds the activations between
 * the forward and the backward call plus all scratch, so one tape serves one (forward, backward) pair at a time.
 * ------------------------------------------------------------------------------------------- */
typedef struct cmh_block_grads {
  float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b, *fc_w, *fc_b, *proj_w, *proj_b;
} cmh_block_grads;
typedef struct cmh_vit_grads {
  float* conv1_w;               /* [width, 3*patch*patch] */
  float* class_embedding;       /* [width] */
  float* positional_embedding;  /* [grid*grid+1, width] */
  float *ln_pre_w, *ln_pre_b, *ln_post_w, *ln_post_b;
  float* proj;                  /* [width, embed_dim] */
  const cmh_block_grads* blocks;   /* host array [layers] */
} cmh_vit_grads;
typedef struct cmh_text_grads {
  float* token_embedding;       /* [vocab, width]  (zeroed, then scatter-added) */
  float* positional_embedding;  /* [context_length, width] */
  float *ln_final_w, *ln_final_b;
  float* text_projection;       /* [width, embed_dim] */
  const cmh_block_grads* blocks;
} cmh_text_grads;
/* Weight / bias gradient of y = x W^T + b: dw[O,I] f32 = dy^T x, db[O] f32 = column sums of dy (db may be NULL).  dy [M,O]
 * and x [M,I] are given in CMH_KIND_* element types and are transposed + cast to `dtype` (the GEMM arithmetic) internally;
 * the product runs on the encoder's GEMM kernel, split over K = M when the output has few tiles. */
size_t cmh_linear_wgrad_workspace_bytes(int32_t dtype, int32_t M, int32_t O, int32_t I);
int cmh_linear_wgrad(int32_t dtype, const void* dy, int32_t dy_kind, const void* x, int32_t x_kind, int32_t M, int32_t O,
                     int32_t I, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of cmh_linear_act (LinearHash, model/modelbase.py:25-35): y = the forward output, dy its gradient, drop_mask /
 * keep_scale / act as in the forward call -> dx [M,K], dw [N,K], db [N].  workspace >= M*N*4 + 256 bytes. */
int cmh_linear_act_backward(const float* x, const float* w, const float* y, const float* dy, const float* drop_mask,
                            float keep_scale, int32_t act, float* dx, float* dw, float* db, int32_t M, int32_t N, int32_t K,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Backward of cmh_dsph_hyp_loss (train/DSPH/loss.py:22-72): dloss = optional device scalar (NULL = 1) ->
 * dx, dy [B,K], dproxies [C,K].  K <= 512. */
size_t cmh_head_backward_workspace_bytes(int32_t B, int32_t K, int32_t C);
int cmh_dsph_hyp_loss_backward(const float* x, const float* y, const float* label, const float* proxies, int32_t B, int32_t K,
                               int32_t C, float threshold, float alpha, const float* dloss, float* dx, float* dy,
                               float* dproxies, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of cmh_pair_softmax (DCHMT select head): p = the forward output [M,2K], dp its gradient -> dz [M,2K]. */
int cmh_pair_softmax_backward(const float* p, const float* dp, float* dz, int32_t M, int32_t K, void* stream);
/* Backward of cmh_dchmt_loss (train/DCHMT/hash_train.py:82-150): same arguments, dloss = optional device scalar (NULL = 1)
 * -> dimg, dtxt [B,D].  D <= 512; workspace >= cmh_head_backward_workspace_bytes(B, D, C). */
int cmh_dchmt_loss_backward(const float* img, const float* txt, const float* label, int32_t B, int32_t D, int32_t C,
                            int32_t output_dim, int32_t similarity, int32_t loss_type, float vartheta, float sim_threshold,
                            const float* dloss, float* dimg, float* dtxt, void* workspace, size_t workspace_bytes, void* stream);
/* Backward of cmh_dnph_loss (train/DNPH_TOMM/loss.py:14-32 + the noise term of hash_train.py:65-81; pass NULL noises for
 * DNPH_out alone): gradients w.r.t. both hash outputs [B,K], both classifier outputs [B,C] and the proxies [C,K]. */
size_t cmh_dnph_backward_workspace_bytes(int32_t B, int32_t K, int32_t C);
int cmh_dnph_loss_backward(const float* hash_img, const float* hash_txt, const float* pre_img, const float* pre_txt,
                           const float* label, const float* proxies, const float* noise_img, const float* noise_txt, int32_t B,
                           int32_t K, int32_t C, float margin, float noise_weight, const float* dloss, float* dhash_img,
                           float* dhash_txt, float* dpre_img, float* dpre_txt, float* dproxies, void* workspace,
                           size_t workspace_bytes, void* stream);
/* ---- input pipeline, image side ---------------------------------------------------------------------
 * dataset/base.py:35-44 (the two transform chains) and :55-64 (_load_image) for a whole batch of DECODED images:
 * train != 0: Resize(R, BICUBIC) -> CenterCrop(R) -> ToTensor -> Normalize(mean, std); train == 0: Resize((R, R), BICUBIC) -> ...
 * Bit-identical to Pillow's Image.resize + torchvision's rules (tests/golden/preprocess.npz).
 *   pixels  device, uint8 RGB HWC images back to back;  offsets device int64 [batch] byte offset of each image;
 *   hw      device int32 [batch, 2] = (height, width);  max_h / max_w >= every image's size (sizes the launch and the workspace);
 *   mean / stdv  HOST float[3];  out device f32 [batch, 3, R, R] and/or out_u8 device uint8 [batch, R, R, 3] (the image that
 *   reaches ToTensor; either may be NULL). */
size_t cmh_image_preprocess_workspace_bytes(int32_t batch, int32_t max_h, int32_t max_w, int32_t R);
int cmh_image_preprocess(const uint8_t* pixels, const int64_t* offsets, const int32_t* hw, int32_t batch, int32_t max_h,
                         int32_t max_w, int32_t R, int32_t train, const float* mean, const float* stdv, float* out,
                         uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream);

/* ToTensor + Normalize of resized images that already live on the device (dataset/base.py:38-39 / :43-44 alone): u8 uint8
 * [N, R, R, 3]; rows (optional, device int64 [batch]) picks the images, else the first `batch`; out f32 [batch, 3, R, R].
 * Serves a device-resident cache of the resized dataset: identical floats to cmh_image_preprocess's, no decode / resize / PCIe. */
int cmh_image_normalize(const uint8_t* u8, const int64_t* rows, int32_t batch, int32_t R, const float* mean, const float* stdv,
                        float* out, void* stream);

/* ---- input pipeline, text side: host code, no GPU involved ------------------------------------------------
 * model/base/simple_tokenizer.py:62-79 (tables from the merges text: pass the gunzipped bpe_simple_vocab_16e6.txt) and
 * dataset/base.py:66-83 (_load_text) for n captions at once: clean -> lower -> split -> BPE -> [SOT] ids [EOT] cut to
 * max_words, zero padded, int64 [n, max_words].  texts = the captions' UTF-8 bytes back to back, offsets int64 [n + 1].
 * status[i] = 0: row i is filled; 1: caption i holds bytes outside printable ASCII / tab / CR / LF, or '&' — ftfy / html.unescape
 * territory — and must go through the Python path (row i is zeroed).  threads <= 0: one per hardware thread (at most 64). */
typedef struct cmh_bpe cmh_bpe;
int cmh_bpe_create(const char* merges_utf8, size_t bytes, cmh_bpe** out);
void cmh_bpe_destroy(cmh_bpe* tokenizer);
int32_t cmh_bpe_vocab_size(const cmh_bpe* tokenizer);
int cmh_bpe_encode_captions(const cmh_bpe* tokenizer, const char* texts, const int64_t* offsets, int32_t n, int32_t max_words,
                            int64_t* out, uint8_t* status, int32_t threads);

size_t cmh_vit_train_bytes(const cmh_vit_weights* w, int32_t batch);
/* same `feat` as cmh_vit_encode (c_fc's QuickGELU runs as a separate pass over the stored pre-activation) */
int cmh_vit_forward_train(const cmh_vit_weights* w, const float* image, int32_t batch, float* feat, void* tape, size_t tape_bytes,
                          void* stream);
int cmh_vit_backward(const cmh_vit_weights* w, int32_t batch, const float* dfeat, const cmh_vit_grads* grads, void* tape,
                     size_t tape_bytes, void* stream);
/* cmh_vit_backward in parts, for a data-parallel trainer that sends each bucket of gradients while the next part runs: one call
 * runs the head (ln_post, proj) iff layer_hi == layers, then blocks layer_hi-1 .. layer_lo, then the embeddings (ln_pre, positional,
 * class, conv1) iff layer_lo == 0; the gradient stream between parts stays in the tape.  Consecutive parts (layers..a, a..b, b..0)
 * on one tape write exactly the gradients of the one-call form; the parameters of a part are final when its call returns.
 * (Upstream: one loss.backward(), train/DSPH/hash_train.py:66.) */
int cmh_vit_backward_part(const cmh_vit_weights* w, int32_t batch, const float* dfeat, const cmh_vit_grads* grads, void* tape,
                          size_t tape_bytes, int32_t layer_hi, int32_t layer_lo, void* stream);
/* The same two calls for the MITH trunk (model/MITH.py:56-82): ln_post + proj on EVERY token.  tokens_out / dtokens f32
 * [batch * (g*g + 1), embed_dim] (row b*(g*g+1) is the class token); tape as for cmh_vit_forward_train. */
int cmh_vit_forward_train_tokens(const cmh_vit_weights* w, const float* image, int32_t batch, float* tokens_out, void* tape,
                                 size_t tape_bytes, void* stream);
int cmh_vit_backward_tokens(const cmh_vit_weights* w, int32_t batch, const float* dtokens, const cmh_vit_grads* grads, void* tape,
                            size_t tape_bytes, void* stream);
size_t cmh_text_train_bytes(const cmh_text_weights* w, int32_t batch, int32_t seq_len);
int cmh_text_forward_train(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                           const uint8_t* key_padding_mask, float* feat, void* tape, size_t tape_bytes, void* stream);
int cmh_text_backward(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                      const uint8_t* key_padding_mask, const float* dfeat, const cmh_text_grads* grads, void* tape,
                      size_t tape_bytes, void* stream);
/* cmh_text_backward in parts (see cmh_vit_backward_part): head = ln_final + text_projection, embeddings = token + positional. */
int cmh_text_backward_part(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                           const uint8_t* key_padding_mask, const float* dfeat, const cmh_text_grads* grads, void* tape,
                           size_t tape_bytes, int32_t layer_hi, int32_t layer_lo, void* stream);

/* CLIP1.encode_text of the MITH trunk under training (model/MITH.py:120-144): causal mask + key_padding_mask, every position is
 * run (no packing), ln_final + text_projection on every token.  tokens_out / dtokens f32 [batch * seq_len, embed_dim];
 * eot_rows_out i32 [batch] = b * seq_len + argmax(tokens[b]) (may be NULL). */
int cmh_text_forward_train_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                  const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out, void* tape,
                                  size_t tape_bytes, void* stream);
/* cmh_text_forward_train_tokens with the promise of cmh_text_encode_tokens_packed (then the padded positions' gradient is zero too):
 * tape and backward run on the kept rows; cmh_text_backward_tokens on this tape takes the dense [B*L, embed_dim] gradient as before
 * and ignores its padded rows. */
int cmh_text_forward_train_tokens_packed(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                                         const uint8_t* key_padding_mask, float* tokens_out, int32_t* eot_rows_out, void* tape,
                                         size_t tape_bytes, void* stream);
int cmh_text_backward_tokens(const cmh_text_weights* w, const int64_t* tokens, int32_t batch, int32_t seq_len,
                             const uint8_t* key_padding_mask, const float* dtokens, const cmh_text_grads* grads, void* tape,
                             size_t tape_bytes, void* stream);

/* ---- MITH HashingModel under training (model/MITH.py:217-453) -------------------------------------------------------------
 * A bare stack of ResidualAttentionBlocks (the concept transformer, :379-396) with a tape: x / y / dy / dx f32 [B*T, d]. */
size_t cmh_blocks_train_bytes(int32_t dtype, int32_t B, int32_t T, int32_t d, int32_t layers);
int cmh_blocks_forward_train(const cmh_block_weights* blocks, int32_t layers, int32_t dtype, const float* x, float* y, int32_t B,
                             int32_t T, int32_t d, void* tape, size_t tape_bytes, void* stream);
int cmh_blocks_backward(const cmh_block_weights* blocks, const cmh_block_grads* grads, int32_t layers, int32_t dtype,
                        const float* dy, float* dx, int32_t B, int32_t T, int32_t d, void* tape, size_t tape_bytes, void* stream);
/* Gradient of cmh_mith_lta w.r.t. the tokens (the similarities are detached upstream, :345): dtokens f32 [B, Ltot, D]. */
int cmh_mith_lta_backward(const float* sim, const uint8_t* key_padding_mask, const float* dmerge, float* dtokens, int32_t B,
                          int32_t Ltot, int32_t l0, int32_t L, int32_t K, int32_t D, int32_t top_k, void* stream);
/* nn.GELU() (exact, erf) and its backward, element-wise f32 (the ResidualMLPs' activation, :224-233). */
int cmh_gelu(const float* x, float* y, int64_t n, void* stream);
int cmh_gelu_backward(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* Backward of cmh_l2_normalize_rows (F.normalize, eps 1e-12) and of cmh_bitwise_hash (x [B,K,D], w [K,D], y = the forward output). */
int cmh_l2_normalize_backward(const float* x, const float* dy, float* dx, int32_t R, int32_t D, void* stream);
int cmh_bitwise_hash_backward(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                              int32_t B, int32_t K, int32_t D, void* stream);

/* Backward of the three MITH loss primitives (train/MITH/hash_train.py:103-147); dloss = device pointer to the upstream scalar
 * gradient (NULL = 1).  Bayesian: gradient w.r.t. the batch codes only (the bank holds detached codes).  InfoNCE: da, db f32
 * [R, D].  Squared difference: da / db f32 [n], either may be NULL. */
size_t cmh_mith_bayesian_backward_workspace_bytes(int32_t B, int32_t K);
int cmh_mith_bayesian_loss_backward(const float* bank, const float* batch, const float* bank_label, const float* label, int32_t Mb,
                                    int32_t B, int32_t K, int32_t C, const float* dloss, float* dbatch, void* workspace,
                                    size_t workspace_bytes, void* stream);
int cmh_info_nce_backward(const float* a, const float* b, int32_t R, int32_t G, int32_t D, float temperature, const float* dloss,
                          float* da, float* db, void* workspace, size_t workspace_bytes, void* stream);
int cmh_sq_diff_sum_backward(const float* a, const float* b, int64_t n, const float* dloss, float* da, float* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* CMH_H_ */
