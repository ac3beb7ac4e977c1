// DMsH-LN multi-similarity loss: reference train/DMsH_LN/MSLOSS.py:13-55 `MultiSimilarityLoss.forward(feats, labels, feat2)` in the
// branch its trainer takes (dataset != "cifar10-1"), forward and backward.
//
//   sim   = F.normalize(feats . feat2^T)          rows of the B x B product scaled to unit L2 norm (eps 1e-12)       (:18-21)
//   same  = labels . labels^T > 0                  labels = LabelNet codes [B, Kl]                                     (:25-26)
//   row i : pos_ = sim[i][same[i]] with sim < 1 - 1e-5;  neg_ = sim[i][!same[i]];  skipped when either is empty       (:31-38)
//           neg  = neg_[neg_ + 0.1 > min(pos_)];  pos = pos_[pos_ - 0.1 < max(neg_)];  skipped when either is empty   (:40-44)
//           loss_i = log(1 + sum exp(-2 (pos - 0.5))) / 2 + log(1 + sum exp(40 (neg - 0.5))) / 40                     (:47-51)
//   loss  = sum_i loss_i / B                                                                                            (:56-57)
//
// Workgroup i owns row i: its 256 threads stride over j.  The raw products S, the similarity bits and per-row statistics stay in
// the caller's workspace; the row losses are added in row order by one wave (f64), so the value does not depend on scheduling.
// Backward: the mining thresholds min(pos_) / max(neg_) only select (comparisons, no gradient), so
//   d loss / d sim_ij = (1/B) * [ -exp(-2 (s - 0.5)) / (1 + sum_pos) | +exp(40 (s - 0.5)) / (1 + sum_neg) ]  on the mined entries,
// then through the row normalisation, dS_i = (g_i - sim_i (sim_i . g_i)) / |S_i|, and dfeats = dS . feat2, dfeat2 = dS^T . feats
// (feat2 = feats: the two are added).  B = 256, K = 64: 4 MFLOP per call - launch-latency-bound like every loss kernel of the path.
#include "cmh_common.h"

namespace cmh {

constexpr int kMslMaxK = 1024;
constexpr float kMslThresh = 0.5f, kMslMargin = 0.1f, kMslScalePos = 2.0f, kMslScaleNeg = 40.0f, kMslEps = 1e-5f;

struct MslWs {
  float* S;         // [B, B]  raw products
  float* G;         // [B, B]  d loss / d S (backward)
  uint8_t* same;    // [B, B]
  float* row;       // [B, 8]  |S_i| clamped, min(pos_), max(neg_), sum_pos, sum_neg, valid, loss_i, unused
  size_t total;
};
static MslWs msl_carve(void* ws, size_t B) {
  char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~static_cast<uintptr_t>(255));
  MslWs w;
  size_t off = 0;
  w.S = reinterpret_cast<float*>(p + off); off += align_up(B * B * 4, 256);
  w.G = reinterpret_cast<float*>(p + off); off += align_up(B * B * 4, 256);
  w.same = reinterpret_cast<uint8_t*>(p + off); off += align_up(B * B, 256);
  w.row = reinterpret_cast<float*>(p + off); off += align_up(B * 8 * 4, 256);
  w.total = off + 256;
  return w;
}

template <typename T, typename Op>
__device__ __forceinline__ T msl_block_reduce(T v, T* red, Op op) {      // over the 256 threads of a workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return op(op(red[0], red[1]), op(red[2], red[3]));
}

// row statistics of the forward (also the first half of the backward)
__global__ __launch_bounds__(256) void msl_rows_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ lab, int B, int K, int Kl, MslWs w) {
  __shared__ float sx[kMslMaxK], sl[kMslMaxK];
  __shared__ float redf[4];
  __shared__ int redi[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < K; k += 256) sx[k] = x[static_cast<size_t>(i) * K + k];
  for (int k = tid; k < Kl; k += 256) sl[k] = lab[static_cast<size_t>(i) * Kl + k];
  __syncthreads();
  float* Si = w.S + static_cast<size_t>(i) * B;
  uint8_t* Li = w.same + static_cast<size_t>(i) * B;
  float sumsq = 0.f;
  for (int j = tid; j < B; j += 256) {
    const float* yj = y + static_cast<size_t>(j) * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(sx[k], yj[k], s);
    const float* lj = lab + static_cast<size_t>(j) * Kl;
    float l = 0.f;
    for (int k = 0; k < Kl; ++k) l = fmaf(sl[k], lj[k], l);
    Si[j] = s;
    Li[j] = l > 0.f ? 1 : 0;
    sumsq = fmaf(s, s, sumsq);
  }
  auto addf = [](float a, float b) { return a + b; };
  auto addi = [](int a, int b) { return a + b; };
  const float norm = sqrtf(msl_block_reduce(sumsq, redf, addf));
  const float denom = fmaxf(norm, 1e-12f);
  float mn = 3.0e38f, mx = -3.0e38f;
  int cp = 0, cn = 0;
  for (int j = tid; j < B; j += 256) {
    const float s = Si[j] / denom;
    if (Li[j]) { if (s < 1.0f - kMslEps) { mn = fminf(mn, s); ++cp; } }
    else { mx = fmaxf(mx, s); ++cn; }
  }
  mn = msl_block_reduce(mn, redf, [](float a, float b) { return fminf(a, b); });
  mx = msl_block_reduce(mx, redf, [](float a, float b) { return fmaxf(a, b); });
  cp = msl_block_reduce(cp, redi, addi);
  cn = msl_block_reduce(cn, redi, addi);
  float sp = 0.f, sn = 0.f;
  int np_ = 0, nn = 0;
  if (cp > 0 && cn > 0) {
    for (int j = tid; j < B; j += 256) {
      const float s = Si[j] / denom;
      if (Li[j]) {
        if (s < 1.0f - kMslEps && s - kMslMargin < mx) { sp += expf(-kMslScalePos * (s - kMslThresh)); ++np_; }
      } else if (s + kMslMargin > mn) { sn += expf(kMslScaleNeg * (s - kMslThresh)); ++nn; }
    }
  }
  sp = msl_block_reduce(sp, redf, addf);
  sn = msl_block_reduce(sn, redf, addf);
  np_ = msl_block_reduce(np_, redi, addi);
  nn = msl_block_reduce(nn, redi, addi);
  if (tid == 0) {
    const bool valid = cp > 0 && cn > 0 && np_ > 0 && nn > 0;
    float* r = w.row + static_cast<size_t>(i) * 8;
    r[0] = denom; r[1] = mn; r[2] = mx; r[3] = sp; r[4] = sn; r[5] = valid ? 1.f : 0.f;
    r[6] = valid ? log1pf(sp) / kMslScalePos + log1pf(sn) / kMslScaleNeg : 0.f;
    r[7] = norm;
  }
}

__global__ __launch_bounds__(64) void msl_sum_kernel(const float* __restrict__ row, int B, float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < B; ++i) t += static_cast<double>(row[static_cast<size_t>(i) * 8 + 6]);    // `sum(loss)` in row order (:56)
  loss[0] = static_cast<float>(t / static_cast<double>(B));
}

// G[i, :] = d loss / d S[i, :];  dx[i, :] (+)= G[i, :] . y
__global__ __launch_bounds__(256) void msl_bwd_rows_kernel(const float* __restrict__ y, int B, int K, MslWs w,
                                                           const float* __restrict__ dloss, float* __restrict__ dx) {
  extern __shared__ float grow[];                    // [B] this row of G
  __shared__ float redf[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* r = w.row + static_cast<size_t>(i) * 8;
  const float denom = r[0], mn = r[1], mx = r[2], sp = r[3], sn = r[4];
  const bool valid = r[5] != 0.f;
  const float up = (dloss ? dloss[0] : 1.f) / static_cast<float>(B);
  const float* Si = w.S + static_cast<size_t>(i) * B;
  const uint8_t* Li = w.same + static_cast<size_t>(i) * B;
  float dot = 0.f;
  for (int j = tid; j < B; j += 256) {
    const float s = Si[j] / denom;
    float g = 0.f;
    if (valid) {
      if (Li[j]) {
        if (s < 1.0f - kMslEps && s - kMslMargin < mx) g = -up * expf(-kMslScalePos * (s - kMslThresh)) / (1.0f + sp);
      } else if (s + kMslMargin > mn) g = up * expf(kMslScaleNeg * (s - kMslThresh)) / (1.0f + sn);
    }
    grow[j] = g;
    dot = fmaf(s, g, dot);
  }
  dot = msl_block_reduce(dot, redf, [](float a, float b) { return a + b; });
  const bool clamped = r[7] < 1e-12f;                 // F.normalize divided by eps: no projection term
  float* Gi = w.G + static_cast<size_t>(i) * B;
  for (int j = tid; j < B; j += 256) {
    const float s = Si[j] / denom;
    const float v = clamped ? grow[j] / denom : (grow[j] - s * dot) / denom;
    grow[j] = v;
    Gi[j] = v;
  }
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 4 <= B; j += 4) {
      a0 = fmaf(grow[j], y[static_cast<size_t>(j) * K + k], a0);
      a1 = fmaf(grow[j + 1], y[static_cast<size_t>(j + 1) * K + k], a1);
      a2 = fmaf(grow[j + 2], y[static_cast<size_t>(j + 2) * K + k], a2);
      a3 = fmaf(grow[j + 3], y[static_cast<size_t>(j + 3) * K + k], a3);
    }
    for (; j < B; ++j) a0 = fmaf(grow[j], y[static_cast<size_t>(j) * K + k], a0);
    dx[static_cast<size_t>(i) * K + k] = (a0 + a1) + (a2 + a3);
  }
}

// dy[j, :] (accumulate ? += : =) sum_i G[i, j] x[i, :]
__global__ __launch_bounds__(256) void msl_bwd_cols_kernel(const float* __restrict__ x, int B, int K, MslWs w, int accumulate,
                                                           float* __restrict__ dy) {
  extern __shared__ float gcol[];                    // [B] column j of G
  const int j = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < B; i += 256) gcol[i] = w.G[static_cast<size_t>(i) * B + j];
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 4 <= B; i += 4) {
      a0 = fmaf(gcol[i], x[static_cast<size_t>(i) * K + k], a0);
      a1 = fmaf(gcol[i + 1], x[static_cast<size_t>(i + 1) * K + k], a1);
      a2 = fmaf(gcol[i + 2], x[static_cast<size_t>(i + 2) * K + k], a2);
      a3 = fmaf(gcol[i + 3], x[static_cast<size_t>(i + 3) * K + k], a3);
    }
    for (; i < B; ++i) a0 = fmaf(gcol[i], x[static_cast<size_t>(i) * K + k], a0);
    const float v = (a0 + a1) + (a2 + a3);
    float* o = dy + static_cast<size_t>(j) * K + k;
    *o = accumulate ? *o + v : v;
  }
}

// ---- DHaPH self-paced contrastive loss: reference train/DHaPH/MSLoss.py:13-33 -------------------------------------------------------
//   same = labels . labels^T > 0;  s = cos(a_i, b_j) (F.normalize, eps 1e-12);  e = exp(s / tau)
//   self-paced weights (DETACHED, :28-29): w+ = exp(-1 - s)^(delta / 4) on the similar pairs, w- = exp(-1 + s)^delta on the others,
//   delta = epoch / int(total / 3) up to a third of the run, then 1 (:23-27; delta = 0 <=> self_paced = False)
//   loss = mean_i -log(P_i / (P_i + N_i)),  P_i = sum_same e w+,  N_i = sum_other e w-
// Backward (weights are constants): d loss / d s_ij = (1 / (B tau)) [ e w (1 / (P_i + N_i)) - same_ij e w+ / P_i ], then through both
// normalisations; a = b (the image-image / text-text calls): the two roles' gradients are added.
// Workspace use: S = the cosines, same bits, row[i] = {P_i, N_i, loss_i}, G = d loss / d s, norms after row.
__global__ __launch_bounds__(256) void spl_norms_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int K,
                                                        float* __restrict__ norms) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 2 * B) return;
  const float* src = row < B ? a + static_cast<size_t>(row) * K : b + static_cast<size_t>(row - B) * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(src[k], src[k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) norms[row] = fmaxf(sqrtf(s), 1e-12f);
}

__global__ __launch_bounds__(256) void spl_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ lab, const float* __restrict__ norms, int B, int K,
                                                       int C, float inv_tau, float delta, MslWs w) {
  __shared__ float sa[kMslMaxK], sl[kMslMaxK];
  __shared__ float redf[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float na = norms[i];
  for (int k = tid; k < K; k += 256) sa[k] = a[static_cast<size_t>(i) * K + k] / na;
  for (int k = tid; k < C; k += 256) sl[k] = lab[static_cast<size_t>(i) * C + k];
  __syncthreads();
  float* Si = w.S + static_cast<size_t>(i) * B;
  uint8_t* Li = w.same + static_cast<size_t>(i) * B;
  float P = 0.f, Nn = 0.f;
  for (int j = tid; j < B; j += 256) {
    const float* bj = b + static_cast<size_t>(j) * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(sa[k], bj[k], s);
    s /= norms[B + j];
    const float* lj = lab + static_cast<size_t>(j) * C;
    float l = 0.f;
    for (int k = 0; k < C; ++k) l = fmaf(sl[k], lj[k], l);
    const bool same = l > 0.f;
    Si[j] = s;
    Li[j] = same ? 1 : 0;
    const float e = expf(s * inv_tau);
    if (same) P += e * expf((-1.f - s) * (delta * 0.25f));
    else Nn += e * expf((-1.f + s) * delta);
  }
  auto addf = [](float x, float y) { return x + y; };
  P = msl_block_reduce(P, redf, addf);
  Nn = msl_block_reduce(Nn, redf, addf);
  if (tid == 0) {
    float* r = w.row + static_cast<size_t>(i) * 8;
    r[0] = P; r[1] = Nn; r[6] = -logf(P / (Nn + P));
  }
}

// G[i, :] = d loss / d s[i, :];  da[i, :] = through the normalisation of a_i of sum_j G[i, j] bn_j
__global__ __launch_bounds__(256) void spl_bwd_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ norms, int B, int K, float inv_tau, float delta,
                                                           MslWs w, const float* __restrict__ dloss, float* __restrict__ da) {
  extern __shared__ float grow[];                    // [B] g_ij / |b_j|
  __shared__ float redf[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* r = w.row + static_cast<size_t>(i) * 8;
  const float P = r[0], Nn = r[1];
  const float up = (dloss ? dloss[0] : 1.f) * inv_tau / static_cast<float>(B);
  const float* Si = w.S + static_cast<size_t>(i) * B;
  const uint8_t* Li = w.same + static_cast<size_t>(i) * B;
  float* Gi = w.G + static_cast<size_t>(i) * B;
  for (int j = tid; j < B; j += 256) {
    const float s = Si[j];
    const float e = expf(s * inv_tau);
    float g;
    if (Li[j]) { const float ew = e * expf((-1.f - s) * (delta * 0.25f)); g = up * ew * (1.f / (P + Nn) - 1.f / P); }
    else g = up * e * expf((-1.f + s) * delta) / (P + Nn);
    Gi[j] = g;
    grow[j] = g / norms[B + j];
  }
  __syncthreads();
  const float na = norms[i];
  float dot = 0.f;
  float dv[4];                                       // K <= 1024: four columns per thread
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = tid + 256 * q;
    dv[q] = 0.f;
    if (k < K) {
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 2 <= B; j += 2) {
        a0 = fmaf(grow[j], b[static_cast<size_t>(j) * K + k], a0);
        a1 = fmaf(grow[j + 1], b[static_cast<size_t>(j + 1) * K + k], a1);
      }
      for (; j < B; ++j) a0 = fmaf(grow[j], b[static_cast<size_t>(j) * K + k], a0);
      dv[q] = a0 + a1;                               // d loss / d an_i[k]
      dot = fmaf(dv[q], a[static_cast<size_t>(i) * K + k] / na, dot);
    }
  }
  dot = msl_block_reduce(dot, redf, [](float x, float y) { return x + y; });
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = tid + 256 * q;
    if (k < K) da[static_cast<size_t>(i) * K + k] = (dv[q] - a[static_cast<size_t>(i) * K + k] / na * dot) / na;
  }
}

// db[j, :] (accumulate ? += : =) through the normalisation of b_j of sum_i G[i, j] an_i
__global__ __launch_bounds__(256) void spl_bwd_cols_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ norms, int B, int K, MslWs w, int accumulate,
                                                           float* __restrict__ db) {
  extern __shared__ float gcol[];                    // [B] G[i, j] / |a_i|
  __shared__ float redf[4];
  const int j = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < B; i += 256) gcol[i] = w.G[static_cast<size_t>(i) * B + j] / norms[i];
  __syncthreads();
  const float nb = norms[B + j];
  float dot = 0.f;
  float dv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = tid + 256 * q;
    dv[q] = 0.f;
    if (k < K) {
      float a0 = 0.f, a1 = 0.f;
      int i = 0;
      for (; i + 2 <= B; i += 2) {
        a0 = fmaf(gcol[i], a[static_cast<size_t>(i) * K + k], a0);
        a1 = fmaf(gcol[i + 1], a[static_cast<size_t>(i + 1) * K + k], a1);
      }
      for (; i < B; ++i) a0 = fmaf(gcol[i], a[static_cast<size_t>(i) * K + k], a0);
      dv[q] = a0 + a1;
      dot = fmaf(dv[q], b[static_cast<size_t>(j) * K + k] / nb, dot);
    }
  }
  dot = msl_block_reduce(dot, redf, [](float x, float y) { return x + y; });
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = tid + 256 * q;
    if (k < K) {
      const float v = (dv[q] - b[static_cast<size_t>(j) * K + k] / nb * dot) / nb;
      float* o = db + static_cast<size_t>(j) * K + k;
      *o = accumulate ? *o + v : v;
    }
  }
}

}  // namespace cmh

using namespace cmh;

// Largest (global) batch of the multi-similarity / self-paced losses: the backward kernels keep one f32 per batch row in dynamic
// LDS (B * 4 bytes next to ~1 KB of static LDS, inside the 64 KB a launch gets without raising its limit) and the workspace holds
// two [B, B] f32 matrices (512 MB at the limit).  Under data parallelism B is the GLOBAL batch (dist_utils.gather_loss_inputs).
static constexpr int kMslMaxB = 8192;

extern "C" size_t cmh_spl_workspace_bytes(int32_t B) {
  if (B <= 0) return 0;
  return msl_carve(nullptr, static_cast<size_t>(B)).total + align_up(static_cast<size_t>(B) * 8, 256);
}

static int spl_check(const float* a, const float* labels, int B, int K, int C, float tau, float delta, void* ws, size_t ws_bytes, const char* what) {
  CMH_CHECK_ARG(a && labels && ws, "%s: null pointer", what);
  CMH_CHECK_ARG(B > 0 && B <= kMslMaxB && K > 0 && K <= kMslMaxK && C > 0 && C <= kMslMaxK, "%s: bad shape B=%d K=%d C=%d", what, B, K, C);
  CMH_CHECK_ARG(tau > 0.f && delta >= 0.f && delta <= 1.f, "%s: temperature %g, delta %g", what, tau, delta);
  if (ws_bytes < cmh_spl_workspace_bytes(B)) return fail(CMH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, cmh_spl_workspace_bytes(B));
  return CMH_OK;
}

extern "C" int cmh_spl_loss(const float* a, const float* b, const float* labels, int32_t B, int32_t K, int32_t C, float temperature,
                            float delta, float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = spl_check(a, labels, B, K, C, temperature, delta, workspace, workspace_bytes, "spl_loss");
  if (rc) return rc;
  CMH_CHECK_ARG(loss, "spl_loss: null pointer");
  const MslWs w = msl_carve(workspace, static_cast<size_t>(B));
  float* norms = reinterpret_cast<float*>(reinterpret_cast<char*>(w.row) + align_up(static_cast<size_t>(B) * 32, 256));
  hipStream_t st = as_stream(stream);
  const float* bb = b ? b : a;
  hipLaunchKernelGGL(spl_norms_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, a, bb, B, K, norms);
  hipLaunchKernelGGL(spl_rows_kernel, dim3(B), dim3(256), 0, st, a, bb, labels, norms, B, K, C, 1.0f / temperature, delta, w);
  hipLaunchKernelGGL(msl_sum_kernel, dim3(1), dim3(64), 0, st, w.row, B, loss);
  CMH_CHECK_LAUNCH("spl_loss");
  return CMH_OK;
}

extern "C" int cmh_spl_loss_backward(const float* a, const float* b, const float* labels, int32_t B, int32_t K, int32_t C,
                                     float temperature, float delta, const float* dloss, float* da, float* db, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  int rc = spl_check(a, labels, B, K, C, temperature, delta, workspace, workspace_bytes, "spl_loss_backward");
  if (rc) return rc;
  CMH_CHECK_ARG(da && (!b || db), "spl_loss_backward: null gradient pointer");
  const MslWs w = msl_carve(workspace, static_cast<size_t>(B));
  float* norms = reinterpret_cast<float*>(reinterpret_cast<char*>(w.row) + align_up(static_cast<size_t>(B) * 32, 256));
  hipStream_t st = as_stream(stream);
  const float* bb = b ? b : a;
  const size_t lds = static_cast<size_t>(B) * 4;
  hipLaunchKernelGGL(spl_norms_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, a, bb, B, K, norms);
  hipLaunchKernelGGL(spl_rows_kernel, dim3(B), dim3(256), 0, st, a, bb, labels, norms, B, K, C, 1.0f / temperature, delta, w);
  hipLaunchKernelGGL(spl_bwd_rows_kernel, dim3(B), dim3(256), lds, st, a, bb, norms, B, K, 1.0f / temperature, delta, w, dloss, da);
  hipLaunchKernelGGL(spl_bwd_cols_kernel, dim3(B), dim3(256), lds, st, a, bb, norms, B, K, w, b ? 0 : 1, b ? db : da);
  CMH_CHECK_LAUNCH("spl_loss_backward");
  return CMH_OK;
}

extern "C" size_t cmh_msl_workspace_bytes(int32_t B) {
  if (B <= 0) return 0;
  return msl_carve(nullptr, static_cast<size_t>(B)).total;
}

static int msl_check(const float* x, const float* labels, int B, int K, int Kl, void* workspace, size_t workspace_bytes, const char* what) {
  CMH_CHECK_ARG(x && labels && workspace, "%s: null pointer", what);
  CMH_CHECK_ARG(B > 0 && B <= kMslMaxB && K > 0 && K <= kMslMaxK && Kl > 0 && Kl <= kMslMaxK, "%s: bad shape B=%d K=%d Kl=%d", what, B, K, Kl);
  if (workspace_bytes < cmh_msl_workspace_bytes(B)) return fail(CMH_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, workspace_bytes, cmh_msl_workspace_bytes(B));
  return CMH_OK;
}

extern "C" int cmh_msl_loss(const float* feats, const float* feat2, const float* labels, int32_t B, int32_t K, int32_t Kl, float* loss,
                            void* workspace, size_t workspace_bytes, void* stream) {
  int rc = msl_check(feats, labels, B, K, Kl, workspace, workspace_bytes, "msl_loss");
  if (rc) return rc;
  CMH_CHECK_ARG(loss, "msl_loss: null pointer");
  const MslWs w = msl_carve(workspace, static_cast<size_t>(B));
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(msl_rows_kernel, dim3(B), dim3(256), 0, st, feats, feat2 ? feat2 : feats, labels, B, K, Kl, w);
  hipLaunchKernelGGL(msl_sum_kernel, dim3(1), dim3(64), 0, st, w.row, B, loss);
  CMH_CHECK_LAUNCH("msl_loss");
  return CMH_OK;
}

extern "C" int cmh_msl_loss_backward(const float* feats, const float* feat2, const float* labels, int32_t B, int32_t K, int32_t Kl,
                                     const float* dloss, float* dfeats, float* dfeat2, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  int rc = msl_check(feats, labels, B, K, Kl, workspace, workspace_bytes, "msl_loss_backward");
  if (rc) return rc;
  CMH_CHECK_ARG(dfeats && (!feat2 || dfeat2), "msl_loss_backward: null gradient pointer");
  const MslWs w = msl_carve(workspace, static_cast<size_t>(B));
  hipStream_t st = as_stream(stream);
  const float* y = feat2 ? feat2 : feats;
  const size_t lds = static_cast<size_t>(B) * 4;
  hipLaunchKernelGGL(msl_rows_kernel, dim3(B), dim3(256), 0, st, feats, y, labels, B, K, Kl, w);       // the forward's statistics
  hipLaunchKernelGGL(msl_bwd_rows_kernel, dim3(B), dim3(256), lds, st, y, B, K, w, dloss, dfeats);
  hipLaunchKernelGGL(msl_bwd_cols_kernel, dim3(B), dim3(256), lds, st, feats, B, K, w, feat2 ? 0 : 1, feat2 ? dfeat2 : dfeats);
  CMH_CHECK_LAUNCH("msl_loss_backward");
  return CMH_OK;
}
