// Backward of the DSPH training step's small pieces (SURVEY §8f "next" #2): the hash head and the HyP proxy loss.
//   cmh_linear_act_backward    LinearHash: y = act((x W^T + b) * mask * keep_scale)            (model/modelbase.py:25-35)
//   cmh_dsph_hyp_loss_backward HyP.forward (train/DSPH/loss.py:22-72): d loss / d(x, y, proxies)
// All f32, launch-latency bound (B = 256, K = 64, C <= ~100): a handful of launches, one wave per row.
#include "cmh_common.h"

namespace cmh {
namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- LinearHash ----------------------------------------------------------------------------------------------------------
// dz[m,n] = dy[m,n] * act'(.) * (mask ? mask*keep_scale : 1);   act' from the OUTPUT y: tanh -> 1 - y^2, relu -> y > 0
__global__ __launch_bounds__(256) void la_dz_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                    const float* __restrict__ mask, float keep_scale, int act, float* __restrict__ dz,
                                                    int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float g = dy[i];
  if (act == CMH_ACT_TANH) g *= 1.0f - y[i] * y[i];
  else if (act == CMH_ACT_RELU) g = y[i] > 0.f ? g : 0.f;
  if (mask) g *= mask[i] * keep_scale;
  dz[i] = g;
}
// dx[m,k] = sum_n dz[m,n] W[n,k]
__global__ __launch_bounds__(256) void la_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                                                    int M, int N, int K) {
  const int m = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc = fmaf(dz[static_cast<size_t>(m) * N + n], w[static_cast<size_t>(n) * K + k], acc);
    dx[static_cast<size_t>(m) * K + k] = acc;
  }
}
// dW[n,k] = sum_m dz[m,n] x[m,k];  db[n] = sum_m dz[m,n]
__global__ __launch_bounds__(256) void la_dw_kernel(const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ dw,
                                                    float* __restrict__ db, int M, int N, int K) {
  // grid (N, ceil(K / 256)); eight independent chains over m (rows m = u mod 8), added in a fixed order: a single serial chain
  // over the 256 batch rows on 64 workgroups took 160 us
  const int n = blockIdx.x;
  const int k = blockIdx.y * 256 + threadIdx.x;
  if (k < K) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int m = 0;
    for (; m + 8 <= M; m += 8)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc[u] = fmaf(dz[static_cast<size_t>(m + u) * N + n], x[static_cast<size_t>(m + u) * K + k], acc[u]);
    for (; m < M; ++m) acc[0] = fmaf(dz[static_cast<size_t>(m) * N + n], x[static_cast<size_t>(m) * K + k], acc[0]);
    dw[static_cast<size_t>(n) * K + k] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  if (blockIdx.y == 0 && threadIdx.x < 64) {     // db[n]: one wave, lanes stride over m, butterfly
    float a = 0.f;
    for (int m = threadIdx.x; m < M; m += 64) a += dz[static_cast<size_t>(m) * N + n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) db[n] = a;
  }
}

// ---- HyP ---------------------------------------------------------------------------------------------------------------------
// rows -> unit vectors (F.normalize, eps 1e-12) + the divisor used
__global__ __launch_bounds__(256) void hyp_normalize_kernel(const float* __restrict__ a, float* __restrict__ an, float* __restrict__ nrm,
                                                            int R, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float ss = 0.f;
  for (int k = lane; k < K; k += 64) { const float v = a[static_cast<size_t>(row) * K + k]; ss = fmaf(v, v, ss); }
  const float n = fmaxf(sqrtf(wsum(ss)), 1e-12f);
  for (int k = lane; k < K; k += 64) an[static_cast<size_t>(row) * K + k] = a[static_cast<size_t>(row) * K + k] / n;
  if (lane == 0) nrm[row] = n;
}

// cnt[0] = #(label != 0) (P_num), cnt[1] = #(label == 0) (N_num); multi[b] = label[b].sum() > 1.  One workgroup.
__global__ __launch_bounds__(256) void hyp_counts_kernel(const float* __restrict__ label, int B, int C, int* __restrict__ multi,
                                                         float* __restrict__ cnt) {
  __shared__ float sh[2][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float p = 0.f, n = 0.f;
  for (int i = threadIdx.x; i < B * C; i += 256) { if (label[i] != 0.f) p += 1.f; else n += 1.f; }
  for (int b = threadIdx.x; b < B; b += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += label[static_cast<size_t>(b) * C + c];
    multi[b] = s > 1.f;
  }
  p = wsum(p); n = wsum(n);
  if (lane == 0) { sh[0][wid] = p; sh[1][wid] = n; }
  __syncthreads();
  if (threadIdx.x < 2) cnt[threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
  if (threadIdx.x == 2) cnt[2] = 0.f;
}
// cnt[2] += #{j : multi[i] && multi[j] && label_i . label_j == 0}   (integers below 2^24: the float atomics are exact)
__global__ __launch_bounds__(64) void hyp_pairs_kernel(const float* __restrict__ label, const int* __restrict__ multi, int B, int C,
                                                       float* __restrict__ cnt) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (!multi[i]) return;
  float z = 0.f;
  for (int j = lane; j < B; j += 64) {
    if (!multi[j]) continue;
    float ll = 0.f;
    for (int c = 0; c < C; ++c) ll = fmaf(label[static_cast<size_t>(i) * C + c], label[static_cast<size_t>(j) * C + c], ll);
    if (ll == 0.f) z += 1.f;
  }
  z = wsum(z);
  if (lane == 0 && z != 0.f) atomicAdd(cnt + 2, z);
}

// One wave per sample b: gradients w.r.t. the NORMALISED rows xn[b], yn[b], then through the normalisation.
//   proxy terms : G[b,c] = -1/P (label == 1) | (cos > thr)/N (label == 0)          (loss.py:25-40; `label == 1` / `== 0` literally)
//   pair terms  : rows i, j both multi-label with label_i . label_j == 0, weight alpha / Z         (loss.py:42-64)
//     x_sim[i,j] (counted at (i,j) and (j,i)) -> dxn_i += 2 w [x_sim > thr] xn_j;   t_sim likewise for yn;
//     xt_sim[i,j] = xn_i . yn_j -> dxn_i += w [.] yn_j;   xt_sim[j,i] = xn_j . yn_i -> dyn_i += w [.] xn_j
template <int KV>   // 64-column slices a lane owns: K <= 64 KV (sized to K: a 64-bit head pays for one slice, not for the 512-bit maximum)
__global__ __launch_bounds__(64) void hyp_rows_kernel(const float* __restrict__ xn, const float* __restrict__ yn,
                                                      const float* __restrict__ pn, const float* __restrict__ nx,
                                                      const float* __restrict__ ny, const float* __restrict__ label,
                                                      const int* __restrict__ multi, const float* __restrict__ cnt, int B, int K,
                                                      int C, float thr, float alpha, const float* __restrict__ dloss,
                                                      float* __restrict__ dx, float* __restrict__ dy) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float up = dloss ? dloss[0] : 1.f;
  const float invP = 1.f / cnt[0], invN = 1.f / cnt[1];
  float xi[KV], yi[KV], gx[KV], gy[KV];
#pragma unroll
  for (int v = 0; v < KV; ++v) {
    const int k = lane + 64 * v;
    xi[v] = k < K ? xn[static_cast<size_t>(b) * K + k] : 0.f;
    yi[v] = k < K ? yn[static_cast<size_t>(b) * K + k] : 0.f;
    gx[v] = 0.f; gy[v] = 0.f;
  }
  for (int c = 0; c < C; ++c) {
    float px[KV], cx = 0.f, ct = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const int k = lane + 64 * v;
      px[v] = k < K ? pn[static_cast<size_t>(c) * K + k] : 0.f;
      cx = fmaf(xi[v], px[v], cx); ct = fmaf(yi[v], px[v], ct);
    }
    cx = wsum(cx); ct = wsum(ct);
    const float l = label[static_cast<size_t>(b) * C + c];
    float wx = 0.f, wt = 0.f;
    if (l == 1.f) { wx = -invP; wt = -invP; }
    else if (l == 0.f) { wx = cx > thr ? invN : 0.f; wt = ct > thr ? invN : 0.f; }
#pragma unroll
    for (int v = 0; v < KV; ++v) { gx[v] = fmaf(wx, px[v], gx[v]); gy[v] = fmaf(wt, px[v], gy[v]); }
  }
  if (alpha > 0.f && cnt[2] > 0.f && multi[b]) {
    const float w = alpha / cnt[2];
    // which j pair with b: 64 candidates at a time, one per lane (a butterfly per candidate made this the longest part of the kernel:
    // 161 us at batch 256); the qualifying ones are then visited in ascending j, as before - same sums, same order
    for (int j0 = 0; j0 < B; j0 += 64) {
      bool pairs = false;
      if (j0 + lane < B && multi[j0 + lane]) {
        float ll = 0.f;
        for (int c = 0; c < C; ++c) ll = fmaf(label[static_cast<size_t>(b) * C + c], label[static_cast<size_t>(j0 + lane) * C + c], ll);
        pairs = ll == 0.f;
      }
      unsigned long long todo = __ballot(pairs);
    while (todo) {
      const int j = j0 + __builtin_ctzll(todo);
      todo &= todo - 1;
      float xj[KV], yj[KV], sx = 0.f, st = 0.f, sxt = 0.f, stx = 0.f;
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        const int k = lane + 64 * v;
        xj[v] = k < K ? xn[static_cast<size_t>(j) * K + k] : 0.f;
        yj[v] = k < K ? yn[static_cast<size_t>(j) * K + k] : 0.f;
        sx = fmaf(xi[v], xj[v], sx); st = fmaf(yi[v], yj[v], st); sxt = fmaf(xi[v], yj[v], sxt); stx = fmaf(xj[v], yi[v], stx);
      }
      sx = wsum(sx); st = wsum(st); sxt = wsum(sxt); stx = wsum(stx);
      const float a = sx > thr ? 2.f * w : 0.f, bq = st > thr ? 2.f * w : 0.f, cq = sxt > thr ? w : 0.f, dq = stx > thr ? w : 0.f;
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        gx[v] = fmaf(a, xj[v], fmaf(cq, yj[v], gx[v]));
        gy[v] = fmaf(bq, yj[v], fmaf(dq, xj[v], gy[v]));
      }
    }
    }
  }
  // through F.normalize: d a = (g - an (an . g)) / max(||a||, eps)
  float dxg = 0.f, dyg = 0.f;
#pragma unroll
  for (int v = 0; v < KV; ++v) { dxg = fmaf(xi[v], gx[v], dxg); dyg = fmaf(yi[v], gy[v], dyg); }
  dxg = wsum(dxg); dyg = wsum(dyg);
  const float rx = up / nx[b], ry = up / ny[b];
#pragma unroll
  for (int v = 0; v < KV; ++v) {
    const int k = lane + 64 * v;
    if (k < K) {
      dx[static_cast<size_t>(b) * K + k] = (gx[v] - xi[v] * dxg) * rx;
      dy[static_cast<size_t>(b) * K + k] = (gy[v] - yi[v] * dyg) * ry;
    }
  }
}

// One wave per proxy c: dpn[c] = sum_b G_x[b,c] xn[b] + G_t[b,c] yn[b], then through the normalisation
// (four waves per proxy, a quarter of the batch each, partial rows added in wave order: one wave per proxy left the chip with C = 24
// waves for a 256-long chain of butterflies, 103 us)
template <int KV>
__global__ __launch_bounds__(256) void hyp_proxy_kernel(const float* __restrict__ xn, const float* __restrict__ yn,
                                                       const float* __restrict__ pn, const float* __restrict__ np_,
                                                       const float* __restrict__ label, const float* __restrict__ cnt, int B, int K,
                                                       int C, float thr, const float* __restrict__ dloss, float* __restrict__ dp) {
  __shared__ float part[4][64 * KV];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float up = dloss ? dloss[0] : 1.f;
  const float invP = 1.f / cnt[0], invN = 1.f / cnt[1];
  float pc[KV], g[KV];
#pragma unroll
  for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; pc[v] = k < K ? pn[static_cast<size_t>(c) * K + k] : 0.f; g[v] = 0.f; }
  const int per = (B + 3) / 4, b0 = wid * per, b1 = b0 + per < B ? b0 + per : B;
  for (int b = b0; b < b1; ++b) {
    float xb[KV], yb[KV], cx = 0.f, ct = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const int k = lane + 64 * v;
      xb[v] = k < K ? xn[static_cast<size_t>(b) * K + k] : 0.f;
      yb[v] = k < K ? yn[static_cast<size_t>(b) * K + k] : 0.f;
      cx = fmaf(xb[v], pc[v], cx); ct = fmaf(yb[v], pc[v], ct);
    }
    cx = wsum(cx); ct = wsum(ct);
    const float l = label[static_cast<size_t>(b) * C + c];
    float wx = 0.f, wt = 0.f;
    if (l == 1.f) { wx = -invP; wt = -invP; }
    else if (l == 0.f) { wx = cx > thr ? invN : 0.f; wt = ct > thr ? invN : 0.f; }
#pragma unroll
    for (int v = 0; v < KV; ++v) g[v] = fmaf(wx, xb[v], fmaf(wt, yb[v], g[v]));
  }
#pragma unroll
  for (int v = 0; v < KV; ++v) part[wid][lane + 64 * v] = g[v];
  __syncthreads();
  if (wid != 0) return;
#pragma unroll
  for (int v = 0; v < KV; ++v) g[v] = (part[0][lane + 64 * v] + part[1][lane + 64 * v]) + (part[2][lane + 64 * v] + part[3][lane + 64 * v]);
  float pg = 0.f;
#pragma unroll
  for (int v = 0; v < KV; ++v) pg = fmaf(pc[v], g[v], pg);
  pg = wsum(pg);
  const float r = up / np_[c];
#pragma unroll
  for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; if (k < K) dp[static_cast<size_t>(c) * K + k] = (g[v] - pc[v] * pg) * r; }
}

// ---- DCHMT ------------------------------------------------------------------------------------------------------------------
// pair softmax backward: dz = p o (dp - (p0 dp0 + p1 dp1)) per adjacent pair
__global__ __launch_bounds__(256) void pair_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               float* __restrict__ dz, int64_t npairs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= npairs) return;
  const float2 pv = *reinterpret_cast<const float2*>(p + 2 * i), dv = *reinterpret_cast<const float2*>(dp + 2 * i);
  const float s = pv.x * dv.x + pv.y * dv.y;
  *reinterpret_cast<float2*>(dz + 2 * i) = float2{pv.x * (dv.x - s), pv.y * (dv.y - s)};
}

// d loss / d similarity of ONE similarity_loss term (train/DCHMT/hash_train.py:82-114), n = B*B elements per mean.
// torch.clamp's backward passes the gradient where min <= x <= max.
__device__ __forceinline__ float dchmt_dsim(float s, bool same, int cosine, int l2, float thr, float maxv, float inv_n) {
  if (cosine) {
    if (same) { if (s < thr) return 0.f; return (l2 ? 2.f * (s - thr) : 1.f) * inv_n; }       // clip(min=thr) - thr
    if (s > 1.f) return 0.f;                                                                 // 1 - clip(max=1)
    return -(l2 ? 2.f * (1.f - s) : 1.f) * inv_n;
  }
  if (same) return (l2 ? 2.f * s : 1.f) * inv_n;
  if (s > maxv) return 0.f;
  return -(l2 ? 2.f * (maxv - s) : 1.f) * inv_n;
}

// One wave per sample i: d loss / d img_i and d txt_i of  L = sl(img,txt) + sl(img,img) + sl(txt,txt)   (our_loss, :116-125).
// A / T are the rows the similarities were taken on (raw for euclidean, unit rows for cosine); na / nt their norms (cosine).
//   euclidean: sim = |a - b| (torch.cdist; zero distance -> zero gradient), d sim / d a = (a - b) / sim
//   cosine   : sim = 1 - an . bn,  d sim / d an = -bn, then through an = a / |a|
// img_i appears as `a` of (img_i, txt_j), as `a` of (img_i, img_j) and as `b` of (img_j, img_i) (equal by symmetry -> factor 2).
__global__ __launch_bounds__(64) void dchmt_rows_kernel(const float* __restrict__ A, const float* __restrict__ T,
                                                        const float* __restrict__ na, const float* __restrict__ nt,
                                                        const float* __restrict__ label, int B, int D, int C, int cosine, int l2,
                                                        float thr, float maxv, const float* __restrict__ dloss,
                                                        float* __restrict__ dimg, float* __restrict__ dtxt) {
  constexpr int DV = 8;                                 // D <= 512
  const int i = blockIdx.x, lane = threadIdx.x;
  const float up = dloss ? dloss[0] : 1.f;
  const float inv_n = 1.f / (static_cast<float>(B) * static_cast<float>(B));
  float ai[DV], ti[DV], ga[DV], gt[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) {
    const int k = lane + 64 * v;
    ai[v] = k < D ? A[static_cast<size_t>(i) * D + k] : 0.f;
    ti[v] = k < D ? T[static_cast<size_t>(i) * D + k] : 0.f;
    ga[v] = 0.f; gt[v] = 0.f;
  }
  for (int j = 0; j < B; ++j) {
    float ll = 0.f;
    for (int c = lane; c < C; c += 64) ll = fmaf(label[static_cast<size_t>(i) * C + c], label[static_cast<size_t>(j) * C + c], ll);
    const bool same = wsum(ll) > 0.f;                    // calc_neighbor
    float aj[DV], tj[DV], s_it = 0.f, s_ti = 0.f, s_ii = 0.f, s_tt = 0.f;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const int k = lane + 64 * v;
      aj[v] = k < D ? A[static_cast<size_t>(j) * D + k] : 0.f;
      tj[v] = k < D ? T[static_cast<size_t>(j) * D + k] : 0.f;
      if (cosine) {
        s_it = fmaf(ai[v], tj[v], s_it); s_ti = fmaf(aj[v], ti[v], s_ti); s_ii = fmaf(ai[v], aj[v], s_ii); s_tt = fmaf(ti[v], tj[v], s_tt);
      } else {
        const float d1 = ai[v] - tj[v], d2 = aj[v] - ti[v], d3 = ai[v] - aj[v], d4 = ti[v] - tj[v];
        s_it = fmaf(d1, d1, s_it); s_ti = fmaf(d2, d2, s_ti); s_ii = fmaf(d3, d3, s_ii); s_tt = fmaf(d4, d4, s_tt);
      }
    }
    s_it = wsum(s_it); s_ti = wsum(s_ti); s_ii = wsum(s_ii); s_tt = wsum(s_tt);
    if (cosine) { s_it = 1.f - s_it; s_ti = 1.f - s_ti; s_ii = 1.f - s_ii; s_tt = 1.f - s_tt; }
    else { s_it = sqrtf(s_it); s_ti = sqrtf(s_ti); s_ii = sqrtf(s_ii); s_tt = sqrtf(s_tt); }
    const float w_it = dchmt_dsim(s_it, same, cosine, l2, thr, maxv, inv_n), w_ti = dchmt_dsim(s_ti, same, cosine, l2, thr, maxv, inv_n);
    const float w_ii = 2.f * dchmt_dsim(s_ii, same, cosine, l2, thr, maxv, inv_n), w_tt = 2.f * dchmt_dsim(s_tt, same, cosine, l2, thr, maxv, inv_n);
    if (cosine) {
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        ga[v] -= w_it * tj[v] + w_ii * aj[v];
        gt[v] -= w_ti * aj[v] + w_tt * tj[v];
      }
    } else {
      const float r_it = s_it > 0.f ? w_it / s_it : 0.f, r_ti = s_ti > 0.f ? w_ti / s_ti : 0.f;
      const float r_ii = s_ii > 0.f ? w_ii / s_ii : 0.f, r_tt = s_tt > 0.f ? w_tt / s_tt : 0.f;
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        ga[v] += r_it * (ai[v] - tj[v]) + r_ii * (ai[v] - aj[v]);
        gt[v] += r_ti * (ti[v] - aj[v]) + r_tt * (ti[v] - tj[v]);
      }
    }
  }
  float pa = 0.f, pt = 0.f;
  if (cosine) {                                          // through a / |a| (no eps: utils/utils.py:61-62)
#pragma unroll
    for (int v = 0; v < DV; ++v) { pa = fmaf(ai[v], ga[v], pa); pt = fmaf(ti[v], gt[v], pt); }
    pa = wsum(pa); pt = wsum(pt);
  }
#pragma unroll
  for (int v = 0; v < DV; ++v) {
    const int k = lane + 64 * v;
    if (k < D) {
      dimg[static_cast<size_t>(i) * D + k] = cosine ? (ga[v] - ai[v] * pa) * up / na[i] : ga[v] * up;
      dtxt[static_cast<size_t>(i) * D + k] = cosine ? (gt[v] - ti[v] * pt) * up / nt[i] : gt[v] * up;
    }
  }
}

// ---- DNPH (TOMM) ---------------------------------------------------------------------------------------------------------------
// DNPH_out.forward (train/DNPH_TOMM/loss.py:14-32) + the noise term of the step (hash_train.py:65-81):
//   f = normalize(cat(img, txt)), p = normalize(proxies), D = |f - p|^2 + mrg [label == 1]
//   p_loss = mean_r sum_c -label[r,c] log_softmax(-D)[r,c];   d_loss = CE(pre_img, argmax label) + CE(pre_txt, argmax label)
//   loss = p_loss + d_loss - noise_weight * (mean_r img_r . noise_img_r + mean_r txt_r . noise_txt_r)
// One wave per row r of cat(img, txt): dz = (softmax(-D) * sum_c label - label) / (2B), dD = -dz (kept in G for the proxy
// pass), df_n = sum_c dD 2 (f_n - p_c), then through the normalisation, plus the noise term.
__global__ __launch_bounds__(64) void dnph_rows_bwd_kernel(const float* __restrict__ fn, const float* __restrict__ nf,
                                                           const float* __restrict__ pn, const float* __restrict__ label,
                                                           const float* __restrict__ noise_img, const float* __restrict__ noise_txt,
                                                           int B, int K, int C, float mrg, float noise_weight,
                                                           const float* __restrict__ dloss, float* __restrict__ G,
                                                           float* __restrict__ dimg, float* __restrict__ dtxt) {
  constexpr int KV = 8;                                  // K <= 512
  extern __shared__ float zs[];                          // [C] -D of this row
  const int r = blockIdx.x, lane = threadIdx.x;
  const int lb = r < B ? r : r - B;                      // label row (label_all = cat(label, label))
  const float up = dloss ? dloss[0] : 1.f;
  float fi[KV], g[KV];
#pragma unroll
  for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; fi[v] = k < K ? fn[static_cast<size_t>(r) * K + k] : 0.f; g[v] = 0.f; }
  float lsum = 0.f;
  for (int c = lane; c < C; c += 64) lsum += label[static_cast<size_t>(lb) * C + c];
  lsum = wsum(lsum);
  float m = -1e30f;
  for (int c = 0; c < C; ++c) {
    float dd = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; const float d = k < K ? fi[v] - pn[static_cast<size_t>(c) * K + k] : 0.f; dd = fmaf(d, d, dd); }
    dd = wsum(dd);
    const float z = -(dd + (label[static_cast<size_t>(lb) * C + c] == 1.f ? mrg : 0.f));
    if (lane == 0) zs[c] = z;
    m = fmaxf(m, z);
  }
  __syncthreads();
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(zs[c] - m);
  se = wsum(se);
  const float inv2b = 1.f / (2.f * static_cast<float>(B));
  for (int c = 0; c < C; ++c) {
    const float sm = expf(zs[c] - m) / se;
    const float dz = (sm * lsum - label[static_cast<size_t>(lb) * C + c]) * inv2b;
    const float dD = -dz;
    if (lane == 0) G[static_cast<size_t>(r) * C + c] = dD;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const int k = lane + 64 * v;
      if (k < K) g[v] = fmaf(2.f * dD, fi[v] - pn[static_cast<size_t>(c) * K + k], g[v]);
    }
  }
  float pg = 0.f;
#pragma unroll
  for (int v = 0; v < KV; ++v) pg = fmaf(fi[v], g[v], pg);
  pg = wsum(pg);
  const float rn = 1.f / nf[r];
  const float* noise = r < B ? noise_img : noise_txt;
  float* out = r < B ? dimg : dtxt;
#pragma unroll
  for (int v = 0; v < KV; ++v) {
    const int k = lane + 64 * v;
    if (k < K) {
      float val = (g[v] - fi[v] * pg) * rn;
      if (noise) val -= noise_weight * noise[static_cast<size_t>(lb) * K + k] / static_cast<float>(B);
      out[static_cast<size_t>(lb) * K + k] = val * up;
    }
  }
}

// One wave per proxy c: dp_n = sum_r G[r,c] 2 (p_c - f_r), then through the normalisation
__global__ __launch_bounds__(64) void dnph_proxy_bwd_kernel(const float* __restrict__ fn, const float* __restrict__ pn,
                                                            const float* __restrict__ np_, const float* __restrict__ G, int R, int K,
                                                            int C, const float* __restrict__ dloss, float* __restrict__ dp) {
  constexpr int KV = 8;
  const int c = blockIdx.x, lane = threadIdx.x;
  const float up = dloss ? dloss[0] : 1.f;
  float pc[KV], g[KV];
#pragma unroll
  for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; pc[v] = k < K ? pn[static_cast<size_t>(c) * K + k] : 0.f; g[v] = 0.f; }
  for (int r = 0; r < R; ++r) {
    const float w = 2.f * G[static_cast<size_t>(r) * C + c];
#pragma unroll
    for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; if (k < K) g[v] = fmaf(w, pc[v] - fn[static_cast<size_t>(r) * K + k], g[v]); }
  }
  float pg = 0.f;
#pragma unroll
  for (int v = 0; v < KV; ++v) pg = fmaf(pc[v], g[v], pg);
  pg = wsum(pg);
  const float rn = up / np_[c];
#pragma unroll
  for (int v = 0; v < KV; ++v) { const int k = lane + 64 * v; if (k < K) dp[static_cast<size_t>(c) * K + k] = (g[v] - pc[v] * pg) * rn; }
}

// CrossEntropyLoss(pre, argmax(label)) mean over the batch: dpre = (softmax(pre) - onehot) / B   (first maximum, like torch.argmax)
__global__ __launch_bounds__(64) void ce_argmax_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ label, int B, int C,
                                                           const float* __restrict__ dloss, float* __restrict__ dpre) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float up = dloss ? dloss[0] : 1.f;
  float m = -1e30f, lbest = -1e30f;
  int li = 0;
  for (int c = lane; c < C; c += 64) {
    m = fmaxf(m, pre[static_cast<size_t>(r) * C + c]);
    const float l = label[static_cast<size_t>(r) * C + c];
    if (l > lbest) { lbest = l; li = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    const float ol = __shfl_xor(lbest, o, 64);
    const int oi = __shfl_xor(li, o, 64);
    if (ol > lbest || (ol == lbest && oi < li)) { lbest = ol; li = oi; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(pre[static_cast<size_t>(r) * C + c] - m);
  se = wsum(se);
  for (int c = lane; c < C; c += 64)
    dpre[static_cast<size_t>(r) * C + c] = (expf(pre[static_cast<size_t>(r) * C + c] - m) / se - (c == li ? 1.f : 0.f)) * up / static_cast<float>(B);
}

// rows / |row| (no eps) and the norms
__global__ __launch_bounds__(256) void plain_normalize_kernel(const float* __restrict__ a, float* __restrict__ an, float* __restrict__ nrm,
                                                              int R, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float ss = 0.f;
  for (int k = lane; k < K; k += 64) { const float v = a[static_cast<size_t>(row) * K + k]; ss = fmaf(v, v, ss); }
  const float n = sqrtf(wsum(ss));
  for (int k = lane; k < K; k += 64) an[static_cast<size_t>(row) * K + k] = a[static_cast<size_t>(row) * K + k] / n;
  if (lane == 0) nrm[row] = n;
}

}  // namespace
}  // namespace cmh

using namespace cmh;

extern "C" size_t cmh_dnph_backward_workspace_bytes(int32_t B, int32_t K, int32_t C) {
  if (B <= 0 || K <= 0 || C <= 0) return 0;
  return align_up(static_cast<size_t>(2) * B * K * 4, 256) + align_up(static_cast<size_t>(C) * K * 4, 256) +
         align_up(static_cast<size_t>(2) * B * 4, 256) + align_up(static_cast<size_t>(C) * 4, 256) +
         align_up(static_cast<size_t>(2) * B * C * 4, 256) + 512;
}

extern "C" int cmh_dnph_loss_backward(const float* hash_img, const float* hash_txt, const float* pre_img, const float* pre_txt,
                                      const float* label, const float* proxies, const float* noise_img, const float* noise_txt,
                                      int32_t B, int32_t K, int32_t C, float margin, float noise_weight, const float* dloss,
                                      float* dhash_img, float* dhash_txt, float* dpre_img, float* dpre_txt, float* dproxies,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(hash_img && hash_txt && pre_img && pre_txt && label && proxies && dhash_img && dhash_txt && dpre_img && dpre_txt &&
                dproxies && workspace, "dnph_loss_backward: null pointer");
  CMH_CHECK_ARG((noise_img == nullptr) == (noise_txt == nullptr), "dnph_loss_backward: give both noise matrices or none");
  CMH_CHECK_ARG(B > 0 && K > 0 && K <= 512 && C > 0 && C <= 4096, "dnph_loss_backward: bad shape B=%d K=%d C=%d", B, K, C);
  if (workspace_bytes < cmh_dnph_backward_workspace_bytes(B, K, C)) return fail(CMH_ERR_WORKSPACE, "dnph_loss_backward: workspace too small");
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  float* fn = reinterpret_cast<float*>(ws); ws += align_up(static_cast<size_t>(2) * B * K * 4, 256);
  float* pn = reinterpret_cast<float*>(ws); ws += align_up(static_cast<size_t>(C) * K * 4, 256);
  float* nf = reinterpret_cast<float*>(ws); ws += align_up(static_cast<size_t>(2) * B * 4, 256);
  float* np_ = reinterpret_cast<float*>(ws); ws += align_up(static_cast<size_t>(C) * 4, 256);
  float* G = reinterpret_cast<float*>(ws);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, hash_img, fn, nf, B, K);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, hash_txt, fn + static_cast<size_t>(B) * K, nf + B, B, K);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, proxies, pn, np_, C, K);
  hipLaunchKernelGGL(dnph_rows_bwd_kernel, dim3(2 * B), dim3(64), static_cast<size_t>(C) * 4, st, fn, nf, pn, label, noise_img, noise_txt, B, K,
                     C, margin, noise_img ? noise_weight : 0.f, dloss, G, dhash_img, dhash_txt);
  hipLaunchKernelGGL(dnph_proxy_bwd_kernel, dim3(C), dim3(64), 0, st, fn, pn, np_, G, 2 * B, K, C, dloss, dproxies);
  hipLaunchKernelGGL(ce_argmax_bwd_kernel, dim3(B), dim3(64), 0, st, pre_img, label, B, C, dloss, dpre_img);
  hipLaunchKernelGGL(ce_argmax_bwd_kernel, dim3(B), dim3(64), 0, st, pre_txt, label, B, C, dloss, dpre_txt);
  CMH_CHECK_LAUNCH("dnph_loss_backward");
  return CMH_OK;
}

extern "C" int cmh_pair_softmax_backward(const float* p, const float* dp, float* dz, int32_t M, int32_t K, void* stream) {
  CMH_CHECK_ARG(p && dp && dz && M > 0 && K > 0, "pair_softmax_backward: bad arguments");
  const int64_t np = static_cast<int64_t>(M) * K;
  hipLaunchKernelGGL(pair_softmax_bwd_kernel, dim3(static_cast<unsigned>((np + 255) / 256)), dim3(256), 0, as_stream(stream), p, dp, dz, np);
  CMH_CHECK_LAUNCH("pair_softmax_backward");
  return CMH_OK;
}

extern "C" int cmh_dchmt_loss_backward(const float* img, const float* txt, const float* label, int32_t B, int32_t D, int32_t C,
                                       int32_t output_dim, int32_t similarity, int32_t loss_type, float vartheta,
                                       float sim_threshold, const float* dloss, float* dimg, float* dtxt, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(img && txt && label && dimg && dtxt && workspace, "dchmt_loss_backward: null pointer");
  CMH_CHECK_ARG(B > 0 && D > 0 && D <= 512 && C > 0 && B <= 32768, "dchmt_loss_backward: bad shape");
  CMH_CHECK_ARG((similarity == 0 || similarity == 1) && (loss_type == 1 || loss_type == 2), "dchmt_loss_backward: bad similarity / loss_type");
  if (workspace_bytes < cmh_head_backward_workspace_bytes(B, D, C)) return fail(CMH_ERR_WORKSPACE, "dchmt_loss_backward: workspace too small");
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t bk = align_up(static_cast<size_t>(B) * D * 4, 256), v = align_up(static_cast<size_t>(B > C ? B : C) * 4, 256);
  float* an = reinterpret_cast<float*>(ws);
  float* tn = reinterpret_cast<float*>(ws + bk);
  float* na = reinterpret_cast<float*>(ws + 2 * bk);
  float* nt = reinterpret_cast<float*>(ws + 2 * bk + v);
  hipStream_t st = as_stream(stream);
  const float* A = img;
  const float* T = txt;
  if (similarity == 1) {   // the forward normalises unless a matrix is identically zero (pair probabilities never are)
    hipLaunchKernelGGL(plain_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, img, an, na, B, D);
    hipLaunchKernelGGL(plain_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, txt, tn, nt, B, D);
    A = an; T = tn;
  }
  const float thr = sim_threshold != 0.f ? sim_threshold : 0.05f;
  const float maxv = sqrtf(static_cast<float>(output_dim) * 2.f * vartheta);
  hipLaunchKernelGGL(dchmt_rows_kernel, dim3(B), dim3(64), 0, st, A, T, na, nt, label, B, D, C, similarity, loss_type == 2, thr, maxv,
                     dloss, dimg, dtxt);
  CMH_CHECK_LAUNCH("dchmt_loss_backward");
  return CMH_OK;
}

extern "C" size_t cmh_head_backward_workspace_bytes(int32_t B, int32_t K, int32_t C) {
  if (B <= 0 || K <= 0 || C <= 0) return 0;
  const size_t bk = align_up(static_cast<size_t>(B) * K * 4, 256), ck = align_up(static_cast<size_t>(C) * K * 4, 256);
  return 2 * bk + ck + 3 * align_up(static_cast<size_t>(B > C ? B : C) * 4, 256) + align_up(static_cast<size_t>(B) * 4, 256) + 512;
}

extern "C" int cmh_linear_act_backward(const float* x, const float* w, const float* y, const float* dy, const float* drop_mask,
                                       float keep_scale, int32_t act, float* dx, float* dw, float* db, int32_t M, int32_t N,
                                       int32_t K, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(x && w && y && dy && dx && dw && db && workspace && M > 0 && N > 0 && K > 0, "linear_act_backward: bad arguments");
  CMH_CHECK_ARG(act == CMH_ACT_NONE || act == CMH_ACT_TANH || act == CMH_ACT_RELU, "linear_act_backward: bad act %d", act);
  if (workspace_bytes < static_cast<size_t>(M) * N * 4 + 256) return fail(CMH_ERR_WORKSPACE, "linear_act_backward: workspace too small");
  float* dz = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(la_dz_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, y, dy, drop_mask, keep_scale, act, dz, M * N);
  hipLaunchKernelGGL(la_dx_kernel, dim3(M), dim3(256), 0, st, dz, w, dx, M, N, K);
  hipLaunchKernelGGL(la_dw_kernel, dim3(N, (K + 255) / 256), dim3(256), 0, st, dz, x, dw, db, M, N, K);
  CMH_CHECK_LAUNCH("linear_act_backward");
  return CMH_OK;
}

extern "C" int cmh_dsph_hyp_loss_backward(const float* x, const float* y, const float* label, const float* proxies, int32_t B,
                                          int32_t K, int32_t C, float threshold, float alpha, const float* dloss, float* dx,
                                          float* dy, float* dproxies, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(x && y && label && proxies && dx && dy && dproxies && workspace, "dsph_hyp_loss_backward: null pointer");
  CMH_CHECK_ARG(B > 0 && K > 0 && K <= 512 && C > 0 && B <= 32768, "dsph_hyp_loss_backward: bad shape B=%d K=%d C=%d", B, K, C);
  if (workspace_bytes < cmh_head_backward_workspace_bytes(B, K, C)) return fail(CMH_ERR_WORKSPACE, "dsph_hyp_loss_backward: workspace too small");
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t bk = align_up(static_cast<size_t>(B) * K * 4, 256), ck = align_up(static_cast<size_t>(C) * K * 4, 256);
  const size_t v = align_up(static_cast<size_t>(B > C ? B : C) * 4, 256);
  float* xn = reinterpret_cast<float*>(ws);
  float* yn = reinterpret_cast<float*>(ws + bk);
  float* pn = reinterpret_cast<float*>(ws + 2 * bk);
  float* nx = reinterpret_cast<float*>(ws + 2 * bk + ck);
  float* ny = reinterpret_cast<float*>(ws + 2 * bk + ck + v);
  float* np_ = reinterpret_cast<float*>(ws + 2 * bk + ck + 2 * v);
  int* multi = reinterpret_cast<int*>(ws + 2 * bk + ck + 3 * v);
  float* cnt = reinterpret_cast<float*>(ws + 2 * bk + ck + 3 * v + align_up(static_cast<size_t>(B) * 4, 256));
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, x, xn, nx, B, K);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((B + 3) / 4), dim3(256), 0, st, y, yn, ny, B, K);
  hipLaunchKernelGGL(hyp_normalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, proxies, pn, np_, C, K);
  hipLaunchKernelGGL(hyp_counts_kernel, dim3(1), dim3(256), 0, st, label, B, C, multi, cnt);
  if (alpha > 0.f) hipLaunchKernelGGL(hyp_pairs_kernel, dim3(B), dim3(64), 0, st, label, multi, B, C, cnt);
#define HYP_BWD(KV)                                                                                                               \
  do {                                                                                                                          \
    hipLaunchKernelGGL(hyp_rows_kernel<KV>, dim3(B), dim3(64), 0, st, xn, yn, pn, nx, ny, label, multi, cnt, B, K, C, threshold, alpha, \
                       dloss, dx, dy);                                                                                          \
    hipLaunchKernelGGL(hyp_proxy_kernel<KV>, dim3(C), dim3(256), 0, st, xn, yn, pn, np_, label, cnt, B, K, C, threshold, dloss,    \
                       dproxies);                                                                                               \
  } while (0)
  if (K <= 64) HYP_BWD(1);
  else if (K <= 128) HYP_BWD(2);
  else if (K <= 256) HYP_BWD(4);
  else HYP_BWD(8);
#undef HYP_BWD
  CMH_CHECK_LAUNCH("dsph_hyp_loss_backward");
  return CMH_OK;
}
