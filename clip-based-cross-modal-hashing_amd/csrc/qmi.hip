// DNpH (TMM 2024) quadratic spherical mutual information loss: reference train/DNpH_TMM/loss.py:5-72 `qmi_loss` in its default
// configuration (use_cosine=True, use_square_clamp=True), forward and backward.
//
//   xh = x / (|x| + eps), th = t / (|t| + eps)                                   (:18-19, :22-23)
//   Y = (xh xh^T + 1)/2, T = (th th^T + 1)/2, YT = (xh th^T + 1)/2               (:19-28)   three B x B matrices
//   D_ij = [label_i . label_j > 0];  M = B^2 / sum(D)                            (:38-44)
//   loss = sum_ij sum_{S in {Y,T,YT}} (D_ij S_ij - 1)^2 + S_ij^2 / M             (:46-57)
//
// Nothing B x B is materialised: workgroup i owns row i of the three matrices (and, for the gradient of the texts through YT,
// column i: YT_ji), its 256 threads stride over j, every pair costs four K-long dot products out of LDS / L2.  The forward leaves
// per-row partial sums (in f64) that a one-wave kernel adds in row order, so the value does not depend on scheduling; the backward
// reuses sum(D) from the forward.  B = 256, K = 64: 17 MFLOP - launch-latency-bound, like every loss kernel of the path.
#include "cmh_common.h"

namespace cmh {

constexpr int kQmiMaxK = 1024;

__device__ __forceinline__ float qmi_block_sum(float v, float* red) {   // sum over the 256 threads of a workgroup
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// norms[0][i] = |x_i|, norms[1][i] = |t_i|
__global__ __launch_bounds__(256) void qmi_norms_kernel(const float* __restrict__ x, const float* __restrict__ t, int B, int K,
                                                        float* __restrict__ norms) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 2 * B) return;
  const float* src = row < B ? x + static_cast<size_t>(row) * K : t + static_cast<size_t>(row - B) * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(src[k], src[k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) norms[row] = sqrtf(s);
}

// BWD = false: part[i] = {sum_j in-terms, sum_j S^2 terms, sum_j D_ij}  (f64 x 3 per row)
// BWD = true : dx[i], dt[i] = gradient rows (scaled by gscale[0]); inv_m = sum(D) / B^2 from the forward
template <bool BWD>
__global__ __launch_bounds__(256) void qmi_rows_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       const uint32_t* __restrict__ lab, const float* __restrict__ norms, int B,
                                                       int K, int LW, float eps, double* __restrict__ part,
                                                       const float* __restrict__ sum_d, const float* __restrict__ gscale,
                                                       float* __restrict__ dx, float* __restrict__ dt) {
  __shared__ float sx[kQmiMaxK], stx[kQmiMaxK];    // xh_i, th_i
  __shared__ float gx[BWD ? 4 * kQmiMaxK : 1], gt[BWD ? 4 * kQmiMaxK : 1];   // BWD: gradient w.r.t. xh_i / th_i, one slice per wave
  __shared__ float red[4];
  __shared__ uint32_t slab[16];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float nx = norms[i], nt = norms[B + i];
  for (int k = tid; k < K; k += 256) {
    sx[k] = x[static_cast<size_t>(i) * K + k] / (nx + eps);
    stx[k] = t[static_cast<size_t>(i) * K + k] / (nt + eps);
    if (BWD) {
#pragma unroll
      for (int w = 0; w < 4; ++w) { gx[w * kQmiMaxK + k] = 0.f; gt[w * kQmiMaxK + k] = 0.f; }
    }
  }
  if (tid < LW) slab[tid] = lab[static_cast<size_t>(i) * LW + tid];
  __syncthreads();
  const float inv_m = BWD ? sum_d[0] / (static_cast<float>(B) * static_cast<float>(B)) : 0.f;
  float in_sum = 0.f, sq_sum = 0.f, d_sum = 0.f;
  // one (i, j) pair per thread and iteration; in the backward a wave handles 64 different j at once, so the accumulation over j
  // is a wave reduction per k into the wave's own LDS slice (no atomics: the four slices are added in a fixed order afterwards)
  for (int j0 = 0; j0 < B; j0 += 256) {
    const int j = j0 + tid;
    float y = 0.f, tt = 0.f, yt = 0.f, ty = 0.f, d = 0.f, rxj = 0.f, rtj = 0.f;
    if (j < B) {
      rxj = 1.0f / (norms[j] + eps);
      rtj = 1.0f / (norms[B + j] + eps);
      const float* xj = x + static_cast<size_t>(j) * K;
      const float* tj = t + static_cast<size_t>(j) * K;
      float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
      for (int k = 0; k < K; ++k) {
        const float xv = xj[k], tv = tj[k];
        a = fmaf(sx[k], xv, a);      // xh_i . x_j
        b = fmaf(stx[k], tv, b);     // th_i . t_j
        c = fmaf(sx[k], tv, c);      // xh_i . t_j
        e = fmaf(stx[k], xv, e);     // th_i . x_j
      }
      y = 0.5f * (a * rxj + 1.0f);
      tt = 0.5f * (b * rtj + 1.0f);
      yt = 0.5f * (c * rtj + 1.0f);  // YT_ij
      ty = 0.5f * (e * rxj + 1.0f);  // YT_ji
      bool hit = false;
      for (int w = 0; w < LW; ++w) hit |= (slab[w] & lab[static_cast<size_t>(j) * LW + w]) != 0;
      d = hit ? 1.f : 0.f;
    }
    if (!BWD) {
      if (j < B) {
        const float a1 = d * y - 1.f, a2 = d * tt - 1.f, a3 = d * yt - 1.f;
        in_sum += a1 * a1 + a2 * a2 + a3 * a3;
        sq_sum += y * y + tt * tt + yt * yt;
        d_sum += d;
      }
    } else {
      // dL/dS = 2 (D S - 1) D + 2 S / M; S = (cos + 1)/2 -> dL/dcos = half of it.  Y and T are symmetric and appear twice
      // (S_ij and S_ji) in the sum, YT once as row i (for xh_i) and once as column i (for th_i).
      const float gy = j < B ? (2.f * (d * y - 1.f) * d + 2.f * y * inv_m) : 0.f;
      const float gtt = j < B ? (2.f * (d * tt - 1.f) * d + 2.f * tt * inv_m) : 0.f;
      const float gyt = j < B ? 0.5f * (2.f * (d * yt - 1.f) * d + 2.f * yt * inv_m) : 0.f;
      const float gty = j < B ? 0.5f * (2.f * (d * ty - 1.f) * d + 2.f * ty * inv_m) : 0.f;
      // gx += gy * xh_j + gyt * th_j ; gt += gtt * th_j + gty * xh_j   (xh_j = x_j * rxj)
      const float* xj = x + static_cast<size_t>(j < B ? j : 0) * K;
      const float* tj = t + static_cast<size_t>(j < B ? j : 0) * K;
      for (int k = 0; k < K; ++k) {
        const float xv = xj[k] * rxj, tv = tj[k] * rtj;
        float vx = gy * xv + gyt * tv, vt = gtt * tv + gty * xv;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { vx += __shfl_xor(vx, o, 64); vt += __shfl_xor(vt, o, 64); }
        if ((tid & 63) == 0) { gx[(tid >> 6) * kQmiMaxK + k] += vx; gt[(tid >> 6) * kQmiMaxK + k] += vt; }
      }
    }
  }
  if (!BWD) {
    const float a = qmi_block_sum(in_sum, red), b = qmi_block_sum(sq_sum, red), c = qmi_block_sum(d_sum, red);
    if (tid == 0) { part[3 * i] = a; part[3 * i + 1] = b; part[3 * i + 2] = c; }
    return;
  }
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    gx[k] = ((gx[k] + gx[kQmiMaxK + k]) + gx[2 * kQmiMaxK + k]) + gx[3 * kQmiMaxK + k];
    gt[k] = ((gt[k] + gt[kQmiMaxK + k]) + gt[2 * kQmiMaxK + k]) + gt[3 * kQmiMaxK + k];
  }
  __syncthreads();
  // through the normalisation xh = x / (n + eps):  dx = g / (n + eps) - x (x . g) / (n (n + eps)^2)
  float px = 0.f, pt = 0.f;
  for (int k = tid; k < K; k += 256) {
    px = fmaf(x[static_cast<size_t>(i) * K + k], gx[k], px);
    pt = fmaf(t[static_cast<size_t>(i) * K + k], gt[k], pt);
  }
  const float dotx = qmi_block_sum(px, red), dott = qmi_block_sum(pt, red);
  const float gs = gscale[0];
  const float cx = nx > 0.f ? dotx / (nx * (nx + eps) * (nx + eps)) : 0.f;
  const float ct = nt > 0.f ? dott / (nt * (nt + eps) * (nt + eps)) : 0.f;
  for (int k = tid; k < K; k += 256) {
    dx[static_cast<size_t>(i) * K + k] = gs * (gx[k] / (nx + eps) - x[static_cast<size_t>(i) * K + k] * cx);
    dt[static_cast<size_t>(i) * K + k] = gs * (gt[k] / (nt + eps) - t[static_cast<size_t>(i) * K + k] * ct);
  }
}

// loss = sum_i in_i + (sum_i d_i / B^2) * sum_i sq_i, rows added in order (f64); sum_d kept for the backward
__global__ __launch_bounds__(64) void qmi_finalize_kernel(const double* __restrict__ part, int B, float* __restrict__ loss,
                                                          float* __restrict__ sum_d) {
  if (threadIdx.x != 0) return;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = 0; i < B; ++i) { a += part[3 * i]; b += part[3 * i + 1]; c += part[3 * i + 2]; }
  const double inv_m = c / (static_cast<double>(B) * static_cast<double>(B));     // 1/M; sum(D) = 0 -> M = inf -> 0, as torch gives
  loss[0] = static_cast<float>(a + inv_m * b);
  sum_d[0] = static_cast<float>(c);
}

}  // namespace cmh

using namespace cmh;

extern "C" size_t cmh_qmi_workspace_bytes(int32_t B) {
  return B > 0 ? align_up(static_cast<size_t>(B) * 3 * sizeof(double), 256) + align_up(static_cast<size_t>(B) * 2 * sizeof(float), 256) : 0;
}

static int qmi_check(const void* a, const void* b, const void* c, int B, int K, int C) {
  CMH_CHECK_ARG(a && b && c, "qmi_loss: null pointer");
  CMH_CHECK_ARG(B > 0 && K > 0 && K <= kQmiMaxK && C > 0 && C <= 512, "qmi_loss: B=%d K=%d (<= %d) C=%d (<= 512)", B, K, kQmiMaxK, C);
  return CMH_OK;
}

extern "C" int cmh_qmi_loss(const float* img, const float* txt, const uint32_t* labels_packed, int32_t B, int32_t K, int32_t C,
                            float eps, float* loss, float* sum_d, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = qmi_check(img, txt, labels_packed, B, K, C);
  if (rc) return rc;
  CMH_CHECK_ARG(loss && sum_d && workspace && workspace_bytes >= cmh_qmi_workspace_bytes(B), "qmi_loss: outputs / workspace");
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  float* norms = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(static_cast<size_t>(B) * 3 * sizeof(double), 256));
  hipLaunchKernelGGL(qmi_norms_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, img, txt, B, K, norms);
  hipLaunchKernelGGL(qmi_rows_kernel<false>, dim3(B), dim3(256), 0, st, img, txt, labels_packed, norms, B, K, (C + 31) / 32, eps, part,
                     nullptr, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(qmi_finalize_kernel, dim3(1), dim3(64), 0, st, part, B, loss, sum_d);
  CMH_CHECK_LAUNCH("qmi_loss");
  return CMH_OK;
}

extern "C" int cmh_qmi_loss_backward(const float* img, const float* txt, const uint32_t* labels_packed, int32_t B, int32_t K, int32_t C,
                                     float eps, const float* sum_d, const float* dloss, float* dimg, float* dtxt, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  int rc = qmi_check(img, txt, labels_packed, B, K, C);
  if (rc) return rc;
  CMH_CHECK_ARG(sum_d && dloss && dimg && dtxt && workspace && workspace_bytes >= cmh_qmi_workspace_bytes(B), "qmi_loss_backward: arguments");
  hipStream_t st = as_stream(stream);
  float* norms = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(static_cast<size_t>(B) * 3 * sizeof(double), 256));
  hipLaunchKernelGGL(qmi_norms_kernel, dim3((2 * B + 3) / 4), dim3(256), 0, st, img, txt, B, K, norms);
  hipLaunchKernelGGL(qmi_rows_kernel<true>, dim3(B), dim3(256), 0, st, img, txt, labels_packed, norms, B, K, (C + 31) / 32, eps, nullptr,
                     sum_d, dloss, dimg, dtxt);
  CMH_CHECK_LAUNCH("qmi_loss_backward");
  return CMH_OK;
}
