// Pairwise similarity / quantisation losses (forward) behind the hashing heads.
//   DSPH  HyP.forward                 train/DSPH/loss.py:22-72
//   DCHMT similarity_loss / our_loss  train/DCHMT/hash_train.py:82-150 (+ utils/utils.py:26-69)
// The matrices are at most B x B with B = batch (<= a few hundred) and K <= 256, i.e. a few hundred KB
// that live in L2: these kernels are launch/latency-bound, not HBM- or MFMA-bound, so the
// design goal is FEW launches (the reference issues ~20-40 tiny ATen kernels per loss): one thread per
// (i,j) pair, block reduction, one double atomic per block per accumulator, then a 1-thread finalize.
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// rows of a [R,K] -> unit L2 norm (F.normalize: x / max(||x||, 1e-12)) or plain division (eps = 0,
// utils/utils.py:61-62); also flags[0] |= any element non-zero.  One wave per row.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* __restrict__ a, float* __restrict__ out, int R,
                                                            int K, float eps, int* __restrict__ anynz) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float ss = 0.f;
  bool nz = false;
  for (int k = lane; k < K; k += 64) {
    const float v = a[static_cast<size_t>(row) * K + k];
    ss = fmaf(v, v, ss);
    nz |= v != 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  float nrm = sqrtf(ss);
  if (eps > 0.f) nrm = fmaxf(nrm, eps);
  for (int k = lane; k < K; k += 64) out[static_cast<size_t>(row) * K + k] = a[static_cast<size_t>(row) * K + k] / nrm;
  if (anynz && __ballot(nz) && lane == 0) atomicOr(anynz, 1);
}

// ---- DSPH ------------------------------------------------------------------------------------------------
// acc: [0] pos_x [1] neg_x [2] pos_t [3] neg_t [4] P_num [5] N_num [6] reg_x [7] reg_t [8] reg_xt [9] zero_pairs
__global__ __launch_bounds__(256) void dsph_proxy_kernel(const float* __restrict__ xn, const float* __restrict__ yn,
                                                         const float* __restrict__ pn, const float* __restrict__ label,
                                                         int B, int K, int C, float thr, double* __restrict__ acc) {
  __shared__ double sh[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v[6] = {0, 0, 0, 0, 0, 0};
  if (i < B * C) {
    const int b = i / C, c = i - b * C;
    float cx = 0.f, ct = 0.f;
    for (int k = 0; k < K; ++k) {
      const float p = pn[static_cast<size_t>(c) * K + k];
      cx = fmaf(xn[static_cast<size_t>(b) * K + k], p, cx);
      ct = fmaf(yn[static_cast<size_t>(b) * K + k], p, ct);
    }
    const float l = label[i];
    if (l == 1.f) { v[0] = 1.f - cx; v[2] = 1.f - ct; }
    if (l == 0.f) { v[1] = fmaxf(cx - thr, 0.f); v[3] = fmaxf(ct - thr, 0.f); v[5] = 1.0; }
    else v[4] = 1.0;
  }
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const double s = block_sum(v[t], sh);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&acc[t], s);
  }
}

// multi[b] = label[b].sum() > 1   (train/DSPH/loss.py:43)
__global__ __launch_bounds__(256) void dsph_multi_kernel(const float* __restrict__ label, int B, int C,
                                                         int* __restrict__ multi) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += label[static_cast<size_t>(b) * C + c];
  multi[b] = s > 1.f;
}

// Pair term of HyP (train/DSPH/loss.py:43-63) over all ordered pairs (i, j) of multi-label rows with disjoint labels.
// One workgroup per 64 x 64 tile of pairs, a thread owns 4 x 4 of them: rows and labels are staged through LDS in chunks of 32
// columns (every global element is read once per tile instead of once per pair), and each pair keeps the same ascending-k fma
// chain as a straightforward loop.  O(B^2 K) work: this is the part of a data-parallel step that grows with the GLOBAL batch.
template <int PT>   // pairs per thread and side: tile = 16 * PT rows (64, or 32 while the grid would not fill the chip)
__global__ __launch_bounds__(256) void dsph_pair_kernel(const float* __restrict__ xn, const float* __restrict__ yn,
                                                        const float* __restrict__ label, const int* __restrict__ multi,
                                                        int B, int K, int C, float thr, double* __restrict__ acc) {
  constexpr int TS = 16 * PT, CH = 32;
  __shared__ float sXi[TS][CH + 1], sTi[TS][CH + 1], sXj[TS][CH + 1], sTj[TS][CH + 1];
  __shared__ double sh[4];
  __shared__ int any_i, any_j;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if (threadIdx.x == 0) { any_i = 0; any_j = 0; }
  __syncthreads();
  if (threadIdx.x < TS) {
    if (i0 + threadIdx.x < B && multi[i0 + threadIdx.x]) any_i = 1;
    if (j0 + threadIdx.x < B && multi[j0 + threadIdx.x]) any_j = 1;
  }
  __syncthreads();
  double v[4] = {0, 0, 0, 0};
  if (any_i && any_j) {                      // uniform: a tile without multi-label rows on one side has no pair
    auto stage = [&](const float* src, int r0, int c0, int ncol, float (*dst)[CH + 1]) {
      for (int e = threadIdx.x; e < TS * CH; e += 256) {
        const int r = e / CH, c = e - r * CH;
        dst[r][c] = (r0 + r < B && c0 + c < ncol) ? src[static_cast<size_t>(r0 + r) * ncol + c0 + c] : 0.f;
      }
    };
    float ll[PT][PT], sx[PT][PT], st[PT][PT], sxt[PT][PT];
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
      for (int b = 0; b < PT; ++b) { ll[a][b] = 0.f; sx[a][b] = 0.f; st[a][b] = 0.f; sxt[a][b] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += CH) {
      __syncthreads();
      stage(label, i0, c0, C, sXi);
      stage(label, j0, c0, C, sXj);
      __syncthreads();
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int a = 0; a < PT; ++a)
#pragma unroll
          for (int b = 0; b < PT; ++b) ll[a][b] = fmaf(sXi[ty * PT + a][c], sXj[tx * PT + b][c], ll[a][b]);
    }
    for (int k0 = 0; k0 < K; k0 += CH) {
      __syncthreads();
      stage(xn, i0, k0, K, sXi); stage(yn, i0, k0, K, sTi);
      stage(xn, j0, k0, K, sXj); stage(yn, j0, k0, K, sTj);
      __syncthreads();
      const int kc = K - k0 < CH ? K - k0 : CH;
      for (int k = 0; k < kc; ++k) {
        float xi[PT], ti[PT], xj[PT], tj[PT];
#pragma unroll
        for (int a = 0; a < PT; ++a) { xi[a] = sXi[ty * PT + a][k]; ti[a] = sTi[ty * PT + a][k]; xj[a] = sXj[tx * PT + a][k]; tj[a] = sTj[tx * PT + a][k]; }
#pragma unroll
        for (int a = 0; a < PT; ++a)
#pragma unroll
          for (int b = 0; b < PT; ++b) {
            sx[a][b] = fmaf(xi[a], xj[b], sx[a][b]);
            st[a][b] = fmaf(ti[a], tj[b], st[a][b]);
            sxt[a][b] = fmaf(xi[a], tj[b], sxt[a][b]);
          }
      }
    }
#pragma unroll
    for (int a = 0; a < PT; ++a)
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        const int i = i0 + ty * PT + a, j = j0 + tx * PT + b;
        if (i < B && j < B && multi[i] && multi[j] && ll[a][b] == 0.f) {
          v[0] += fmaxf(sx[a][b] - thr, 0.f);
          v[1] += fmaxf(st[a][b] - thr, 0.f);
          v[2] += fmaxf(sxt[a][b] - thr, 0.f);
          v[3] += 1.0;
        }
      }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double s = block_sum(v[t], sh);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&acc[6 + t], s);
  }
}

__global__ void dsph_finalize_kernel(const double* __restrict__ acc, float alpha, float* __restrict__ loss) {
  double t = acc[0] / acc[4] + acc[1] / acc[5] + acc[2] / acc[4] + acc[3] / acc[5];
  if (alpha > 0.f && acc[9] > 0.0) t += static_cast<double>(alpha) * (acc[6] + acc[7] + acc[8]) / acc[9];
  loss[0] = static_cast<float>(t);
}

// ---- DCHMT -----------------------------------------------------------------------------------------------
// one (a,b) similarity matrix: acc[0] += sum pos-term, acc[1] += sum neg-term  (means taken in finalize)
__global__ __launch_bounds__(256) void dchmt_pair_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ label, int B, int D, int C,
                                                         int cosine, int l2, float thr, float maxv,
                                                         double* __restrict__ acc) {
  __shared__ double sh[4];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  double v[2] = {0, 0};
  if (idx < B * B) {
    const int i = idx / B, j = idx - i * B;
    float ll = 0.f;
    for (int c = 0; c < C; ++c) ll = fmaf(label[static_cast<size_t>(i) * C + c], label[static_cast<size_t>(j) * C + c], ll);
    const bool same = ll > 0.f;                              // calc_neighbor
    float s = 0.f;
    if (cosine) {
      for (int k = 0; k < D; ++k) s = fmaf(a[static_cast<size_t>(i) * D + k], b[static_cast<size_t>(j) * D + k], s);
      s = 1.f - s;
    } else {
      for (int k = 0; k < D; ++k) {
        const float d = a[static_cast<size_t>(i) * D + k] - b[static_cast<size_t>(j) * D + k];
        s = fmaf(d, d, s);
      }
      s = sqrtf(s);
    }
    float pos, neg;
    if (cosine) {
      pos = fmaxf(same ? s : 0.f, thr) - thr;                // clip(min=thr) - thr
      neg = same ? 0.f : 1.f - fminf(s, 1.f);                // 1*(1-ls) - clip(max=1)
    } else {
      pos = same ? s : 0.f;
      neg = same ? 0.f : maxv - fminf(s, maxv);
    }
    v[0] = l2 ? static_cast<double>(pos) * pos : pos;
    v[1] = l2 ? static_cast<double>(neg) * neg : neg;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const double s = block_sum(v[t], sh);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&acc[t], s);
  }
}

__global__ void dchmt_finalize_kernel(const double* __restrict__ acc, int B, float* __restrict__ loss) {
  loss[0] = static_cast<float>((acc[0] + acc[1]) / (static_cast<double>(B) * B));
}

// conditional copy used by the cosine branch: utils/utils.py:61-62 normalises only if the matrix has a non-zero
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ raw, const float* __restrict__ nrm,
                                                          const int* __restrict__ anynz, float* __restrict__ out,
                                                          int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *anynz ? nrm[i] : raw[i];
}

}  // namespace cmh

using namespace cmh;

extern "C" size_t cmh_loss_workspace_bytes(int32_t B, int32_t K, int32_t C) {
  if (B <= 0 || K <= 0 || C <= 0) return 0;
  const size_t bk = align_up(static_cast<size_t>(B) * K * 4, 256);
  return 256 /*acc*/ + 256 /*flags*/ + 4 * bk + align_up(static_cast<size_t>(C) * K * 4, 256) +
         align_up(static_cast<size_t>(B) * 4, 256);
}

extern "C" int cmh_dsph_hyp_loss(const float* x, const float* y, const float* label, const float* proxies, int32_t B,
                                 int32_t K, int32_t C, float threshold, float alpha, float* loss, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(x && y && label && proxies && loss && workspace, "dsph_hyp_loss: null pointer");
  CMH_CHECK_ARG(B > 0 && K > 0 && C > 0 && B <= 32768, "dsph_hyp_loss: bad shape B=%d K=%d C=%d", B, K, C);
  if (workspace_bytes < cmh_loss_workspace_bytes(B, K, C)) return fail(CMH_ERR_WORKSPACE, "dsph_hyp_loss: workspace too small");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  double* acc = reinterpret_cast<double*>(ws);
  const size_t bk = align_up(static_cast<size_t>(B) * K * 4, 256);
  float* xn = reinterpret_cast<float*>(ws + 512);
  float* yn = reinterpret_cast<float*>(ws + 512 + bk);
  float* pn = reinterpret_cast<float*>(ws + 512 + 4 * bk);
  int* multi = reinterpret_cast<int*>(ws + 512 + 4 * bk + align_up(static_cast<size_t>(C) * K * 4, 256));
  if (hipMemsetAsync(acc, 0, 256, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "dsph_hyp_loss: memset failed");
  hipLaunchKernelGGL(row_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, x, xn, B, K, 1e-12f, nullptr);
  hipLaunchKernelGGL(row_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, y, yn, B, K, 1e-12f, nullptr);
  hipLaunchKernelGGL(row_normalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, proxies, pn, C, K, 1e-12f, nullptr);
  hipLaunchKernelGGL(dsph_proxy_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, xn, yn, pn, label, B, K, C, threshold, acc);
  if (alpha > 0.f) {
    hipLaunchKernelGGL(dsph_multi_kernel, dim3((B + 255) / 256), dim3(256), 0, st, label, B, C, multi);
    if (B > 1024)
      hipLaunchKernelGGL(dsph_pair_kernel<4>, dim3((B + 63) / 64, (B + 63) / 64), dim3(256), 0, st, xn, yn, label, multi, B, K, C, threshold, acc);
    else
      hipLaunchKernelGGL(dsph_pair_kernel<2>, dim3((B + 31) / 32, (B + 31) / 32), dim3(256), 0, st, xn, yn, label, multi, B, K, C, threshold, acc);
  }
  hipLaunchKernelGGL(dsph_finalize_kernel, dim3(1), dim3(1), 0, st, acc, alpha, loss);
  CMH_CHECK_LAUNCH("dsph_hyp_loss");
  return CMH_OK;
}

extern "C" int cmh_dchmt_loss(const float* img, const float* txt, const float* label, int32_t B, int32_t D, int32_t C,
                              int32_t output_dim, int32_t similarity, int32_t loss_type, float vartheta,
                              float sim_threshold, float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(img && txt && label && loss && workspace, "dchmt_loss: null pointer");
  CMH_CHECK_ARG(B > 0 && D > 0 && C > 0 && B <= 32768, "dchmt_loss: bad shape");
  CMH_CHECK_ARG(similarity == 0 || similarity == 1, "dchmt_loss: similarity must be 0 (euclidean) or 1 (cosine)");
  CMH_CHECK_ARG(loss_type == 1 || loss_type == 2, "dchmt_loss: loss_type must be 1 (l1) or 2 (l2)");
  if (workspace_bytes < cmh_loss_workspace_bytes(B, D, C)) return fail(CMH_ERR_WORKSPACE, "dchmt_loss: workspace too small");
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  double* acc = reinterpret_cast<double*>(ws);
  int* flags = reinterpret_cast<int*>(ws + 256);
  const size_t bk = align_up(static_cast<size_t>(B) * D * 4, 256);
  float* in_ = reinterpret_cast<float*>(ws + 512);
  float* tn_ = reinterpret_cast<float*>(ws + 512 + bk);
  float* ia = reinterpret_cast<float*>(ws + 512 + 2 * bk);
  float* ta = reinterpret_cast<float*>(ws + 512 + 3 * bk);
  if (hipMemsetAsync(ws, 0, 512, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "dchmt_loss: memset failed");
  const float* A = img;
  const float* T = txt;
  const float thr = sim_threshold != 0.f ? sim_threshold : 0.05f;   // hash_train.py:82,86-87
  if (similarity == 1) {
    const int64_t n = static_cast<int64_t>(B) * D;
    hipLaunchKernelGGL(row_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, img, in_, B, D, 0.f, flags);
    hipLaunchKernelGGL(row_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, txt, tn_, B, D, 0.f, flags + 1);
    hipLaunchKernelGGL(select_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, img, in_, flags, ia, n);
    hipLaunchKernelGGL(select_rows_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, txt, tn_, flags + 1, ta, n);
    A = ia;
    T = ta;
  }
  const float maxv = sqrtf(static_cast<float>(output_dim) * 2.f * vartheta);   // :101
  const dim3 grid((B * B + 255) / 256);
  const int l2 = loss_type == 2;
  hipLaunchKernelGGL(dchmt_pair_kernel, grid, dim3(256), 0, st, A, T, label, B, D, C, similarity, l2, thr, maxv, acc);
  hipLaunchKernelGGL(dchmt_pair_kernel, grid, dim3(256), 0, st, A, A, label, B, D, C, similarity, l2, thr, maxv, acc);
  hipLaunchKernelGGL(dchmt_pair_kernel, grid, dim3(256), 0, st, T, T, label, B, D, C, similarity, l2, thr, maxv, acc);
  hipLaunchKernelGGL(dchmt_finalize_kernel, dim3(1), dim3(1), 0, st, acc, B, loss);
  CMH_CHECK_LAUNCH("dchmt_loss");
  return CMH_OK;
}
