// Heads and losses of the DNPH (TOMM) and TwDH methods (configs 4 and 5 of BASELINE.json):
//   TwDH ModalityHash     model/TwDH.py:54-85     (length-1 MHA == out_proj(v_proj(x)); BatchNorm1d in TRAIN mode for
//                                                  the image head - Baseclip.eval() never reaches it, SURVEY §7 - /
//                                                  LayerNorm for text; fc2 + ReLU + pair softmax)
//   TwDH targets + loss   train/TwDH/hash_train.py:77-163 (hash_center_multilables, hash_convert, BCELoss, soft-argmax term)
//   DNPH_out + noise term train/DNPH_TOMM/loss.py:14-32, train/DNPH_TOMM/hash_train.py:65-81
// All operands are [batch, <= 4096] matrices: launch/latency-bound, so each op is ONE launch (+ a 1-thread finalize).
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ float h2_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double h2_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float h2_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// BatchNorm1d, training mode: per-feature batch mean and BIASED variance (what nn.BatchNorm1d normalises with).
// One wave per feature column, lanes stride the batch.
__global__ __launch_bounds__(256) void batchnorm_train_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps,
                                                              float* __restrict__ y, int B, int d) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= d) return;
  float s = 0.f;
  for (int r = lane; r < B; r += 64) s += x[static_cast<size_t>(r) * d + col];
  const float mean = h2_wave_sum(s) / static_cast<float>(B);
  float ss = 0.f;
  for (int r = lane; r < B; r += 64) {
    const float c = x[static_cast<size_t>(r) * d + col] - mean;
    ss = fmaf(c, c, ss);
  }
  const float rstd = 1.0f / sqrtf(h2_wave_sum(ss) / static_cast<float>(B) + eps);
  const float g = w[col], bb = b[col];
  for (int r = lane; r < B; r += 64)
    y[static_cast<size_t>(r) * d + col] = (x[static_cast<size_t>(r) * d + col] - mean) * rstd * g + bb;
}

// The same with the module's side effect: running_mean / running_var <- (1 - momentum) * old + momentum * batch statistic, the
// variance UNBIASED (B / (B - 1)) as nn.BatchNorm1d stores it.  The reference's image head is never put in eval mode, so every
// forward — validation included — moves these buffers, and they are part of the checkpoint.
__global__ __launch_bounds__(256) void batchnorm_running_kernel(const float* __restrict__ x, float momentum,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                int B, int d) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= d) return;
  float s = 0.f;
  for (int r = lane; r < B; r += 64) s += x[static_cast<size_t>(r) * d + col];
  const float mean = h2_wave_sum(s) / static_cast<float>(B);
  float ss = 0.f;
  for (int r = lane; r < B; r += 64) {
    const float c = x[static_cast<size_t>(r) * d + col] - mean;
    ss = fmaf(c, c, ss);
  }
  ss = h2_wave_sum(ss);
  if (lane == 0) {
    const float unbiased = B > 1 ? ss / static_cast<float>(B - 1) : ss;
    running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
    running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
  }
}

// backward of batchnorm_train_kernel: xh = (x - mean) rstd;  dg = sum dy xh, db = sum dy,  dx = g rstd (dy - db / B - xh dg / B)
__global__ __launch_bounds__(256) void batchnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int d) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= d) return;
  float s = 0.f;
  for (int r = lane; r < B; r += 64) s += x[static_cast<size_t>(r) * d + col];
  const float mean = h2_wave_sum(s) / static_cast<float>(B);
  float ss = 0.f;
  for (int r = lane; r < B; r += 64) {
    const float c = x[static_cast<size_t>(r) * d + col] - mean;
    ss = fmaf(c, c, ss);
  }
  const float rstd = 1.0f / sqrtf(h2_wave_sum(ss) / static_cast<float>(B) + eps);
  float sg = 0.f, sb = 0.f;
  for (int r = lane; r < B; r += 64) {
    const float g = dy[static_cast<size_t>(r) * d + col];
    sg = fmaf(g, (x[static_cast<size_t>(r) * d + col] - mean) * rstd, sg);
    sb += g;
  }
  sg = h2_wave_sum(sg); sb = h2_wave_sum(sb);
  if (lane == 0) { dw[col] = sg; db[col] = sb; }
  const float k = w[col] * rstd, inv = 1.0f / static_cast<float>(B);
  for (int r = lane; r < B; r += 64) {
    const float xh = (x[static_cast<size_t>(r) * d + col] - mean) * rstd;
    dx[static_cast<size_t>(r) * d + col] = k * (dy[static_cast<size_t>(r) * d + col] - sb * inv - xh * sg * inv);
  }
}

// backward of twdh_loss_kernel's two terms for one element of p (train/TwDH/hash_train.py:117-139):
//   nce  = (BCE(p_img, t) + BCE(p_txt, t)) / 2, BCELoss mean over the n2 = B * 2K elements; ATen's backward is
//          (p - t) / max((1 - p) p, 1e-12) / n2;      quan = ((1 - mean((2p - 1)^2))_img + (...)_txt) / 2  ->  -4 (2p - 1) / n2
__global__ __launch_bounds__(256) void twdh_loss_bwd_kernel(const float* __restrict__ p_img, const float* __restrict__ p_txt,
                                                            const float* __restrict__ target, int64_t n2,
                                                            const float* __restrict__ d_nce, const float* __restrict__ d_quan,
                                                            float* __restrict__ dp_img, float* __restrict__ dp_txt) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  const float gn = (d_nce ? d_nce[0] : 0.f) * 0.5f / static_cast<float>(n2);
  const float gq = (d_quan ? d_quan[0] : 0.f) * 0.5f / static_cast<float>(n2);
  const float code = target[i >> 1];
  const float t = ((i & 1) != 0) == (code > 0.f) ? 1.f : 0.f;         // hash_convert: code > 0 -> pair (0, 1), else (1, 0)
  const float a = p_img[i], b = p_txt[i];
  dp_img[i] = gn * (a - t) / fmaxf((1.f - a) * a, 1e-12f) - gq * 4.f * (2.f * a - 1.f);
  dp_txt[i] = gn * (b - t) / fmaxf((1.f - b) * b, 1e-12f) - gq * 4.f * (2.f * b - 1.f);
}

// hash_center_multilables: code[b,k] = sign(mean_{c: label[b,c]==1} center[c,k]), exact zeros replaced by
// random_center[k] (a +-1 vector the caller draws once per call, like the reference's torch.randint_like).
// Rows without any label give NaN means upstream, which hash_convert maps to "bit 0" == -1.
__global__ __launch_bounds__(256) void twdh_targets_kernel(const float* __restrict__ label, const float* __restrict__ center,
                                                           const float* __restrict__ random_center,
                                                           float* __restrict__ code, int B, int C, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  float s = 0.f;
  int n = 0;
  for (int c = 0; c < C; ++c)
    if (label[static_cast<size_t>(b) * C + c] == 1.f) { s += center[static_cast<size_t>(c) * K + k]; ++n; }
  float v;
  if (n == 0) v = -1.f;
  else if (s > 0.f) v = 1.f;
  else if (s < 0.f) v = -1.f;
  else v = random_center[k];
  code[i] = v;
}

// acc[0] += sum BCE(p_img, onehot(target)) ; acc[1] += sum BCE(p_txt, ..) ; acc[2] += sum (2p_img-1)^2 ; acc[3] += .. txt
// p: [B, 2K] pair probabilities; target [B, K] in {-1,+1}: pair (1,0) for -1, (0,1) for +1 (hash_convert).
// nn.BCELoss clamps log at -100.
__global__ __launch_bounds__(256) void twdh_loss_kernel(const float* __restrict__ pi, const float* __restrict__ pt,
                                                        const float* __restrict__ target, int64_t n2,
                                                        double* __restrict__ acc) {
  __shared__ double sh[4][4];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  double v[4] = {0, 0, 0, 0};
  if (i < n2) {
    const float y = ((target[i >> 1] > 0.f) == ((i & 1) == 1)) ? 1.f : 0.f;
    const float a = pi[i], t = pt[i];
    v[0] = -(y * fmaxf(logf(a), -100.f) + (1.f - y) * fmaxf(logf(1.f - a), -100.f));
    v[1] = -(y * fmaxf(logf(t), -100.f) + (1.f - y) * fmaxf(logf(1.f - t), -100.f));
    v[2] = (2.f * a - 1.f) * (2.f * a - 1.f);
    v[3] = (2.f * t - 1.f) * (2.f * t - 1.f);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double s = h2_wave_sum(v[j]);
    if (lane == 0) sh[j][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double s = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    if (s != 0.0) atomicAdd(&acc[threadIdx.x], s);
  }
}

// out[0] = nce = (mean BCE_img + mean BCE_txt)/2 ; out[1] = quan = ((1-mean sq_img) + (1-mean sq_txt))/2
__global__ void twdh_finalize_kernel(const double* __restrict__ acc, double n2, float* __restrict__ out) {
  out[0] = static_cast<float>((acc[0] / n2 + acc[1] / n2) * 0.5);
  out[1] = static_cast<float>(((1.0 - acc[2] / n2) + (1.0 - acc[3] / n2)) * 0.5);
}

// DNPH_out: one wave per row r of cat(img, txt) (2B rows).  acc[0] += sum_c -label*log_softmax(-D)_c,
// acc[1] += CE(pre_img), acc[2] += CE(pre_txt), acc[3] += <hash_img, noise_img>, acc[4] += <hash_txt, noise_txt>.
__global__ __launch_bounds__(256) void dnph_row_kernel(const float* __restrict__ fi, const float* __restrict__ ft,
                                                       const float* __restrict__ prei, const float* __restrict__ pret,
                                                       const float* __restrict__ label, const float* __restrict__ prox,
                                                       const float* __restrict__ ni, const float* __restrict__ nt,
                                                       float mrg, int B, int K, int C, double* __restrict__ acc) {
  extern __shared__ float sf[];   // per wave: normalised feature row [K]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wid;
  if (r >= 2 * B) return;
  const int b = r < B ? r : r - B;
  const float* f = (r < B ? fi : ft) + static_cast<size_t>(b) * K;
  const float* pre = (r < B ? prei : pret) + static_cast<size_t>(b) * C;
  const float* nz = r < B ? ni : nt;
  float* row = sf + wid * K;
  float ss = 0.f, dotn = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float v = f[k];
    ss = fmaf(v, v, ss);
    if (nz) dotn = fmaf(v, nz[static_cast<size_t>(b) * K + k], dotn);
  }
  const float inv = 1.0f / fmaxf(sqrtf(h2_wave_sum(ss)), 1e-12f);
  for (int k = lane; k < K; k += 64) row[k] = f[k] * inv;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // logits_c = -(||f - p_c||^2 + mrg*[label==1]); lanes stride the classes
  float mx = -1e30f;
  for (int c = lane; c < C; c += 64) {
    const float* p = prox + static_cast<size_t>(c) * K;
    float pn = 0.f;
    for (int k = 0; k < K; ++k) pn = fmaf(p[k], p[k], pn);
    const float pinv = 1.0f / fmaxf(sqrtf(pn), 1e-12f);
    float d2 = 0.f;
    for (int k = 0; k < K; ++k) {
      const float df = row[k] - p[k] * pinv;
      d2 = fmaf(df, df, d2);
    }
    const float lg = -(d2 + (label[static_cast<size_t>(b) * C + c] == 1.f ? mrg : 0.f));
    mx = fmaxf(mx, lg);
  }
  mx = h2_wave_max(mx);
  float se = 0.f, lab_lg = 0.f, lab_n = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float* p = prox + static_cast<size_t>(c) * K;
    float pn = 0.f;
    for (int k = 0; k < K; ++k) pn = fmaf(p[k], p[k], pn);
    const float pinv = 1.0f / fmaxf(sqrtf(pn), 1e-12f);
    float d2 = 0.f;
    for (int k = 0; k < K; ++k) {
      const float df = row[k] - p[k] * pinv;
      d2 = fmaf(df, df, d2);
    }
    const float l = label[static_cast<size_t>(b) * C + c];
    const float lg = -(d2 + (l == 1.f ? mrg : 0.f));
    se += expf(lg - mx);
    lab_lg = fmaf(l, lg, lab_lg);
    lab_n += l;
  }
  se = h2_wave_sum(se);
  lab_lg = h2_wave_sum(lab_lg);
  lab_n = h2_wave_sum(lab_n);
  const float lse = mx + logf(se);
  const double ploss = -(static_cast<double>(lab_lg) - static_cast<double>(lab_n) * lse);   // sum_c -l*(lg - lse)
  // CrossEntropy(pre, argmax(label)) : first maximum, like torch.argmax
  float best = -1e30f;
  int besti = 0x7fffffff;
  float pmx = -1e30f;
  for (int c = lane; c < C; c += 64) {
    const float l = label[static_cast<size_t>(b) * C + c];
    if (l > best) { best = l; besti = c; }
    pmx = fmaxf(pmx, pre[c]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  pmx = h2_wave_max(pmx);
  float pse = 0.f;
  for (int c = lane; c < C; c += 64) pse += expf(pre[c] - pmx);
  pse = h2_wave_sum(pse);
  const double ce = -(static_cast<double>(pre[besti]) - (pmx + logf(pse)));
  const float dn = h2_wave_sum(dotn);
  if (lane == 0) {
    atomicAdd(&acc[0], ploss);
    atomicAdd(&acc[r < B ? 1 : 2], ce);
    if (nz) atomicAdd(&acc[r < B ? 3 : 4], static_cast<double>(dn));
  }
}

__global__ void dnph_finalize_kernel(const double* __restrict__ acc, int B, float noise_weight, float* __restrict__ out) {
  const double p_loss = acc[0] / (2.0 * B);
  const double d_loss = acc[1] / B + acc[2] / B;
  const double noise = acc[3] / B + acc[4] / B;
  out[0] = static_cast<float>(p_loss + d_loss - static_cast<double>(noise_weight) * noise);
  out[1] = static_cast<float>(p_loss + d_loss);
  out[2] = static_cast<float>(noise);
}

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_batchnorm1d_train(const float* x, const float* w, const float* b, float eps, float* y, int32_t B,
                                     int32_t d, void* stream) {
  CMH_CHECK_ARG(x && w && b && y && B > 0 && d > 0, "batchnorm1d_train: bad arguments");
  hipLaunchKernelGGL(batchnorm_train_kernel, dim3((d + 3) / 4), dim3(256), 0, as_stream(stream), x, w, b, eps, y, B, d);
  CMH_CHECK_LAUNCH("batchnorm1d_train");
  return CMH_OK;
}

extern "C" int cmh_batchnorm1d_update_running(const float* x, float momentum, float* running_mean, float* running_var,
                                              int32_t B, int32_t d, void* stream) {
  CMH_CHECK_ARG(x && running_mean && running_var && B > 0 && d > 0, "batchnorm1d_update_running: bad arguments");
  hipLaunchKernelGGL(batchnorm_running_kernel, dim3((d + 3) / 4), dim3(256), 0, as_stream(stream), x, momentum, running_mean,
                     running_var, B, d);
  CMH_CHECK_LAUNCH("batchnorm1d_update_running");
  return CMH_OK;
}

extern "C" int cmh_batchnorm1d_backward(const float* x, const float* w, float eps, const float* dy, float* dx, float* dw,
                                        float* db, int32_t B, int32_t d, void* stream) {
  CMH_CHECK_ARG(x && w && dy && dx && dw && db && B > 0 && d > 0, "batchnorm1d_backward: bad arguments");
  hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3((d + 3) / 4), dim3(256), 0, as_stream(stream), x, w, eps, dy, dx, dw, db, B, d);
  CMH_CHECK_LAUNCH("batchnorm1d_backward");
  return CMH_OK;
}

extern "C" int cmh_twdh_loss_backward(const float* p_img, const float* p_txt, const float* target, int32_t B, int32_t K,
                                      const float* d_nce, const float* d_quan, float* dp_img, float* dp_txt, void* stream) {
  CMH_CHECK_ARG(p_img && p_txt && target && dp_img && dp_txt && B > 0 && K > 0, "twdh_loss_backward: bad arguments");
  const int64_t n2 = static_cast<int64_t>(B) * K * 2;
  hipLaunchKernelGGL(twdh_loss_bwd_kernel, dim3(static_cast<unsigned>((n2 + 255) / 256)), dim3(256), 0, as_stream(stream), p_img,
                     p_txt, target, n2, d_nce, d_quan, dp_img, dp_txt);
  CMH_CHECK_LAUNCH("twdh_loss_backward");
  return CMH_OK;
}

extern "C" int cmh_twdh_targets(const float* label, const float* center, const float* random_center, float* code,
                                int32_t B, int32_t C, int32_t K, void* stream) {
  CMH_CHECK_ARG(label && center && random_center && code && B > 0 && C > 0 && K > 0, "twdh_targets: bad arguments");
  hipLaunchKernelGGL(twdh_targets_kernel, dim3((B * K + 255) / 256), dim3(256), 0, as_stream(stream), label, center,
                     random_center, code, B, C, K);
  CMH_CHECK_LAUNCH("twdh_targets");
  return CMH_OK;
}

extern "C" int cmh_twdh_loss(const float* p_img, const float* p_txt, const float* target, int32_t B, int32_t K,
                             float* out2, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(p_img && p_txt && target && out2 && workspace && B > 0 && K > 0, "twdh_loss: bad arguments");
  if (workspace_bytes < 256) return fail(CMH_ERR_WORKSPACE, "twdh_loss: workspace must be >= 256 bytes");
  hipStream_t st = as_stream(stream);
  double* acc = static_cast<double*>(workspace);
  if (hipMemsetAsync(acc, 0, 64, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "twdh_loss: memset failed");
  const int64_t n2 = static_cast<int64_t>(B) * K * 2;
  hipLaunchKernelGGL(twdh_loss_kernel, dim3(static_cast<unsigned>((n2 + 255) / 256)), dim3(256), 0, st, p_img, p_txt,
                     target, n2, acc);
  hipLaunchKernelGGL(twdh_finalize_kernel, dim3(1), dim3(1), 0, st, acc, static_cast<double>(n2), out2);
  CMH_CHECK_LAUNCH("twdh_loss");
  return CMH_OK;
}

extern "C" int cmh_dnph_loss(const float* hash_img, const float* hash_txt, const float* pre_img, const float* pre_txt,
                             const float* label, const float* proxies, const float* noise_img, const float* noise_txt,
                             int32_t B, int32_t K, int32_t C, float margin, float noise_weight, float* out3,
                             void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(hash_img && hash_txt && pre_img && pre_txt && label && proxies && out3 && workspace, "dnph_loss: null pointer");
  CMH_CHECK_ARG(B > 0 && K > 0 && K <= 4096 && C > 0, "dnph_loss: bad shape");
  CMH_CHECK_ARG((noise_img == nullptr) == (noise_txt == nullptr), "dnph_loss: give both noise matrices or neither");
  if (workspace_bytes < 256) return fail(CMH_ERR_WORKSPACE, "dnph_loss: workspace must be >= 256 bytes");
  hipStream_t st = as_stream(stream);
  double* acc = static_cast<double*>(workspace);
  if (hipMemsetAsync(acc, 0, 64, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "dnph_loss: memset failed");
  hipLaunchKernelGGL(dnph_row_kernel, dim3((2 * B + 3) / 4), dim3(256), static_cast<size_t>(4) * K * sizeof(float), st,
                     hash_img, hash_txt, pre_img, pre_txt, label, proxies, noise_img, noise_txt, margin, B, K, C, acc);
  hipLaunchKernelGGL(dnph_finalize_kernel, dim3(1), dim3(1), 0, st, acc, B, noise_weight, out3);
  CMH_CHECK_LAUNCH("dnph_loss");
  return CMH_OK;
}
