// MITH (config 3): the pieces of HashingModel and of the MITH losses that are not plain GEMM / LayerNorm / attention.
//   LocalizedTokenAggregation   model/MITH.py:317-376   (positive-only, per-token top-k over concepts, softmax over tokens, bmm)
//   PositionalEncoding add      model/MITH.py:249-273
//   BitwiseHashing              model/MITH.py:276-293   (K x Linear(512,1) + tanh)
//   F.normalize                 model/MITH.py:442-443,450-451
//   losses                      train/MITH/hash_train.py:80-83,103-147,168-201 (bayesian vs memory bank, InfoNCE,
//                               token-level InfoNCE, quantisation / distillation squared errors, B = sign(mix))
// Activations are batch-major ([N, K, D] where the reference holds [K, N, D]); values are identical.
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ float m_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double m_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float m_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double m_block_sum(double v, double* sh) {
  v = m_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

constexpr int kLtaMaxL = 80;     // tokens per sample (49 patches / <= 77 words)
constexpr int kLtaMaxK = 128;    // concepts (= hash bits)
constexpr int kTop = 8;          // compile-time bound of top_k_label (reference default 8)
constexpr float kNegInf = -__builtin_huge_valf();

// grid (B, ceil(D / 256)): a workgroup owns one sample and 256 columns of D (steps 1-3 are repeated per column block: they are
// cheap next to the L x K x D aggregation, and two blocks per sample fill the chip at batch 128).  sim rows [row0 + l] (l < L)
// of a [B*Ltot, K] matrix, tokens rows likewise of [B*Ltot, D].  The weights sit in LDS with a 16-byte aligned row pitch: the
// aggregation reads them four concepts at a time (one ds_read_b128 per 4 FMAs instead of one ds_read_b32 per FMA).
// steps 1-3 of LocalizedTokenAggregation for one sample: w[l][k] = the aggregation weights (softmax over the tokens of the
// masked, positive, per-token top-k similarities); shared by the forward kernel and the gradient w.r.t. the tokens
__device__ __forceinline__ void lta_build_weights(float (*w)[kLtaMaxK + 4], const float* __restrict__ sim,
                                                  const uint8_t* __restrict__ kpm, int b, size_t row0, int L, int K, int top_k) {
  const int tid = threadIdx.x;
  // 1. load, key-padding -> -inf, non-positive -> -inf   (:350-361)
  for (int i = tid; i < L * K; i += 256) {
    const int l = i / K, k = i - l * K;
    float v = sim[(row0 + l) * K + k];
    if (kpm && kpm[static_cast<size_t>(b) * L + l]) v = kNegInf;
    w[l][k] = v > 0.f ? v : kNegInf;
  }
  __syncthreads();
  // 2. per token: keep entries >= the top_k-th largest value (ties keep more, like torch.ge against min(topk))  (:322-333)
  if (tid < L) {
    float top[kTop];
#pragma unroll
    for (int j = 0; j < kTop; ++j) top[j] = kNegInf;
    for (int k = 0; k < K; ++k) {
      float v = w[tid][k];
      if (v > top[kTop - 1]) {
        top[kTop - 1] = v;
#pragma unroll
        for (int j = kTop - 1; j > 0; --j)
          if (top[j] > top[j - 1]) { const float t = top[j]; top[j] = top[j - 1]; top[j - 1] = t; }
      }
    }
    // the top_k-th largest with multiplicity sits at index top_k-1 once at least top_k finite values were seen;
    // with fewer positives it is -inf and everything is kept
    float vmin = kNegInf;
#pragma unroll
    for (int j = 0; j < kTop; ++j)
      if (j == top_k - 1) vmin = top[j];
    for (int k = 0; k < K; ++k)
      if (!(w[tid][k] >= vmin)) w[tid][k] = kNegInf;
  }
  __syncthreads();
  // 3. softmax over the tokens for every concept; an all -inf column gives NaN upstream, replaced by 0  (:364-366)
  if (tid < K) {
    float m = kNegInf;
    for (int l = 0; l < L; ++l) m = fmaxf(m, w[l][tid]);
    if (m == kNegInf) {
      for (int l = 0; l < L; ++l) w[l][tid] = 0.f;
    } else {
      float s = 0.f;
      for (int l = 0; l < L; ++l) {
        const float e = expf(w[l][tid] - m);
        w[l][tid] = e;
        s += e;
      }
      const float inv = 1.0f / s;
      for (int l = 0; l < L; ++l) w[l][tid] *= inv;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void lta_kernel(const float* __restrict__ tokens, const float* __restrict__ sim,
                                                  const uint8_t* __restrict__ kpm, float* __restrict__ out, int Ltot,
                                                  int l0, int L, int K, int D, int top_k) {
  __shared__ __attribute__((aligned(16))) float w[kLtaMaxL][kLtaMaxK + 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t row0 = static_cast<size_t>(b) * Ltot + l0;
  lta_build_weights(w, sim, kpm, b, row0, L, K, top_k);
  // 4. merge[b,k,:] = sum_l w[l,k] * tokens[b,l,:]   (:370-375)
  const int d = blockIdx.y * 256 + tid;
  if (d < D) {
    float acc[kLtaMaxK];
#pragma unroll
    for (int k = 0; k < kLtaMaxK; ++k) acc[k] = 0.f;
    for (int l = 0; l < L; ++l) {
      const float xv = tokens[(row0 + l) * D + d];
#pragma unroll
      for (int k4 = 0; k4 < kLtaMaxK / 4; ++k4)
        if (k4 * 4 < K) {                       // K % 4 == 0 (host-checked): whole quads; columns >= K hold zeros
          const float4 wv = *reinterpret_cast<const float4*>(&w[l][k4 * 4]);
          acc[k4 * 4 + 0] = fmaf(wv.x, xv, acc[k4 * 4 + 0]); acc[k4 * 4 + 1] = fmaf(wv.y, xv, acc[k4 * 4 + 1]);
          acc[k4 * 4 + 2] = fmaf(wv.z, xv, acc[k4 * 4 + 2]); acc[k4 * 4 + 3] = fmaf(wv.w, xv, acc[k4 * 4 + 3]);
        }
    }
#pragma unroll
    for (int k = 0; k < kLtaMaxK; ++k)
      if (k < K) out[(static_cast<size_t>(b) * K + k) * D + d] = acc[k];
  }
}

__global__ __launch_bounds__(256) void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pe, int64_t n, int T,
                                                      int D) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = static_cast<int>(i % D);
  const int t = static_cast<int>((i / D) % T);
  x[i] += pe[static_cast<size_t>(t) * D + d];
}

// out[b,k] = tanh(x[b,k,:] . w[k,:] + bias[k]); one wave per (b,k)
__global__ __launch_bounds__(256) void bitwise_hash_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int B, int K, int D) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B * K) return;
  const int k = i % K;
  const float* xr = x + static_cast<size_t>(i) * D;
  const float* wr = w + static_cast<size_t>(k) * D;
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 a = *reinterpret_cast<const float4*>(xr + d);
    const float4 c = *reinterpret_cast<const float4*>(wr + d);
    s = fmaf(a.x, c.x, s); s = fmaf(a.y, c.y, s); s = fmaf(a.z, c.z, s); s = fmaf(a.w, c.w, s);
  }
  s = m_wave_sum(s);
  if (lane == 0) out[i] = tanhf(s + bias[k]);
}

__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int D) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[static_cast<size_t>(r) * D + d]; ss = fmaf(v, v, ss); }
  const float inv = 1.0f / fmaxf(sqrtf(m_wave_sum(ss)), 1e-12f);
  for (int d = lane; d < D; d += 64) y[static_cast<size_t>(r) * D + d] = x[static_cast<size_t>(r) * D + d] * inv;
}

// B = sign(lambda*(ic + tc) + (1-lambda)*(it + tt)); Hi = .5 ic + .5 it; Ht = .5 tc + .5 tt   (hash_train.py:80-83,179-180)
__global__ __launch_bounds__(256) void mith_mix_kernel(const float* __restrict__ ic, const float* __restrict__ it,
                                                       const float* __restrict__ tc, const float* __restrict__ tt,
                                                       float lam, float* __restrict__ Bc, float* __restrict__ Hi,
                                                       float* __restrict__ Ht, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (ic[i] * lam + it[i] * (1.f - lam)) + (tc[i] * lam + tt[i] * (1.f - lam));
  Bc[i] = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);
  Hi[i] = ic[i] * 0.5f + it[i] * 0.5f;
  Ht[i] = tc[i] * 0.5f + tt[i] * 0.5f;
}

// acc[slot] += sum (a-b)^2
__global__ __launch_bounds__(256) void sq_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      double* __restrict__ acc) {
  __shared__ double sh[4];
  double v = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const double d = static_cast<double>(a[i]) - static_cast<double>(b[i]);
    v += d * d;
  }
  const double s = m_block_sum(v, sh);
  if (threadIdx.x == 0 && s != 0.0) atomicAdd(acc, s);
}

// bayesian_loss(bank, batch, label_sim) partial: acc += sum_{m,b} (ls*s - log(1+e^s)), s = .5*clamp(bank_m.batch_b, +-64),
// ls = (bank_label_m . label_b > 0).  grid (B, 16 bank slices): a wave takes four bank rows at a time, lanes across the bits /
// the classes (coalesced rows, wave reductions); the value of every (m, b) pair is formed exactly as a per-pair loop would.
__global__ __launch_bounds__(256) void bayes_kernel(const float* __restrict__ bank, const float* __restrict__ batch,
                                                    const float* __restrict__ bank_label, const float* __restrict__ label,
                                                    int Mb, int B, int K, int C, double* __restrict__ acc) {
  __shared__ double sh[4];
  const int b = blockIdx.x, sl = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int per_slice = (Mb + gridDim.y - 1) / gridDim.y;
  const int s0 = sl * per_slice, s1 = s0 + per_slice < Mb ? s0 + per_slice : Mb;
  const int n = s1 > s0 ? s1 - s0 : 0;
  double v = 0.0;
  for (int mb = wid * 4; mb < n; mb += 16) {
    float dot[4], ll[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = s0 + (mb + u < n ? mb + u : n - 1);
      float d = 0.f, l = 0.f;
      for (int k = lane; k < K; k += 64) d = fmaf(bank[static_cast<size_t>(m) * K + k], batch[static_cast<size_t>(b) * K + k], d);
      for (int c = lane; c < C; c += 64) l = fmaf(bank_label[static_cast<size_t>(m) * C + c], label[static_cast<size_t>(b) * C + c], l);
      dot[u] = d; ll[u] = l;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) { dot[u] += __shfl_xor(dot[u], o, 64); ll[u] += __shfl_xor(ll[u], o, 64); }
    if (lane < 4 && mb + lane < n) {
      const float d = lane == 0 ? dot[0] : (lane == 1 ? dot[1] : (lane == 2 ? dot[2] : dot[3]));
      const float l = lane == 0 ? ll[0] : (lane == 1 ? ll[1] : (lane == 2 ? ll[2] : ll[3]));
      const float s = 0.5f * fminf(fmaxf(d, -64.f), 64.f);
      v += static_cast<double>((l > 0.f ? s : 0.f) - logf(1.f + expf(s)));
    }
  }
  const double t = m_block_sum(v, sh);
  if (threadIdx.x == 0 && t != 0.0) atomicAdd(acc, t);
}

// ---- the same loss on the f32 matrix pipe (K and C <= 128) ----------------------------------------------------------------------------
// The loop above spends its time in butterflies: every (bank row, batch row) pair costs a 6-step wave reduction for 64 + 80
// products (384 us for 256 x 10 000 x 64 bit x 80 classes).  v_mfma_f32_16x16x4_f32 forms 16 x 16 such dot products as k-ordered f32
// FMA chains with one operand element per lane and no cross-lane step: a wave keeps 16 batch rows as B operands (bq / lq: element
// j = k 4j + lane/16 of row lane%16) and walks over 16-row tiles of the bank; lane (row%16 = lane&15, quad = lane>>4) then holds the
// pairs (bank row 4 quad + r, batch row lane&15), r = 0..3.  Values differ from the loop above by f32 summation order only.
typedef float m_f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBayesQ = 32;   // operand quads per lane: K, C <= 128

struct BayesOperands { float bq[kBayesQ], lq[kBayesQ]; };

__device__ __forceinline__ void bayes_load_batch(BayesOperands& o, const float* __restrict__ batch, const float* __restrict__ label, int b,
                                                 int K, int C, int quad) {
#pragma unroll
  for (int j = 0; j < kBayesQ; ++j) {
    const int k = 4 * j + quad;
    o.bq[j] = k < K ? batch[static_cast<size_t>(b) * K + k] : 0.f;
    o.lq[j] = k < C ? label[static_cast<size_t>(b) * C + k] : 0.f;
  }
}

// d[r], l[r]: bank row (tile row 4 quad + r) . batch row, bank label row . batch label row; `m` = this lane's (clamped) bank row
__device__ __forceinline__ void bayes_tile(const BayesOperands& o, const float* __restrict__ bank, const float* __restrict__ bank_label,
                                           int m, int K, int C, int quad, m_f32x4_t& d, m_f32x4_t& l) {
  const float* br = bank + static_cast<size_t>(m) * K;
  const float* lr = bank_label + static_cast<size_t>(m) * C;
  d = m_f32x4_t{0.f, 0.f, 0.f, 0.f};
  l = m_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < kBayesQ; ++j) {
    const int k = 4 * j + quad;
    if (4 * j < K) d = __builtin_amdgcn_mfma_f32_16x16x4f32(k < K ? br[k] : 0.f, o.bq[j], d, 0, 0, 0);
    if (4 * j < C) l = __builtin_amdgcn_mfma_f32_16x16x4f32(k < C ? lr[k] : 0.f, o.lq[j], l, 0, 0, 0);
  }
}

// grid (ceil(B / 16), slices); a wave takes every fourth 16-row tile of its bank slice
__global__ __launch_bounds__(256) void bayes_mfma_kernel(const float* __restrict__ bank, const float* __restrict__ batch,
                                                         const float* __restrict__ bank_label, const float* __restrict__ label,
                                                         int Mb, int B, int K, int C, double* __restrict__ acc) {
  __shared__ double sh[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, quad = lane >> 4;
  const int b0 = blockIdx.x * 16, sl = blockIdx.y;
  const int per_slice = (Mb + gridDim.y - 1) / gridDim.y;
  const int s0 = sl * per_slice, s1 = s0 + per_slice < Mb ? s0 + per_slice : Mb;
  BayesOperands o;
  bayes_load_batch(o, batch, label, b0 + r16 < B ? b0 + r16 : B - 1, K, C, quad);
  double v = 0.0;
  for (int mt = s0 + wid * 16; mt < s1; mt += 64) {
    m_f32x4_t d, l;
    bayes_tile(o, bank, bank_label, mt + r16 < s1 ? mt + r16 : s1 - 1, K, C, quad, d, l);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (mt + quad * 4 + r < s1 && b0 + r16 < B) {
        const float sv = 0.5f * fminf(fmaxf(d[r], -64.f), 64.f);
        v += static_cast<double>((l[r] > 0.f ? sv : 0.f) - logf(1.f + expf(sv)));
      }
  }
  const double t = m_block_sum(v, sh);
  if (threadIdx.x == 0 && t != 0.0) atomicAdd(acc, t);
}

// Grouped row cross-entropy with diagonal targets: row i (group g = i / G, index t = i % G) scores a_i . b_j / temp against
// the G rows j of its group; acc += logsumexp_j - score_{i,t}.  G = N: info_nce_loss; G = L: info_nce_loss_bmm.
__global__ __launch_bounds__(256) void row_ce_kernel(const float* __restrict__ a, const float* __restrict__ b, int R, int G,
                                                     int D, float inv_temp, double* __restrict__ acc) {
  extern __shared__ float sa[];   // 4 waves x D
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wid;
  if (i >= R) return;
  const int g0 = (i / G) * G;
  float* ar = sa + wid * D;
  for (int d = lane; d < D; d += 64) ar[d] = a[static_cast<size_t>(i) * D + d];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float mx = -1e30f, diag = 0.f;
  // pass 1: scores for this lane's candidates (kept in registers for up to 4 candidates per lane: G <= 256) else recomputed
  for (int j = lane; j < G; j += 64) {
    const float* br = b + static_cast<size_t>(g0 + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(ar[d], br[d], s);
    s *= inv_temp;
    mx = fmaxf(mx, s);
    if (g0 + j == i) diag = s;
  }
  mx = m_wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < G; j += 64) {
    const float* br = b + static_cast<size_t>(g0 + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(ar[d], br[d], s);
    se += expf(s * inv_temp - mx);
  }
  se = m_wave_sum(se);
  diag = m_wave_sum(diag);
  if (lane == 0) atomicAdd(acc, static_cast<double>(mx + logf(se) - diag));
}

__global__ void read_acc_kernel(const double* __restrict__ acc, double scale, float* __restrict__ out) {
  out[0] = static_cast<float>(acc[0] * scale);
}

// ---- backward of the loss terms (train/MITH/hash_train.py:103-201) ---------------------------------------------------------------
// bayesian_loss = -mean_{m,b}(ls s - log(1 + e^s)), s = .5 clamp(bank_m . batch_b, +-64):  d/d batch_b = -(1/(Mb B)) sum_m (ls - sigmoid(s))
// * .5 bank_m where the clamp is inactive.  One workgroup per batch row: coefficients of all bank rows first, then one thread per bit.
constexpr int kBayesSlices = 16;      // bank slices per batch row: 16 x B workgroups keep the chip busy at B = 128
constexpr int kBayesMaxRows = 4096;   // bank rows per slice held in LDS

__global__ __launch_bounds__(256) void bayes_bwd_kernel(const float* __restrict__ bank, const float* __restrict__ batch,
                                                        const float* __restrict__ bank_label, const float* __restrict__ label, int Mb,
                                                        int B, int K, int C, const float* __restrict__ dloss, float* __restrict__ partial) {
  __shared__ float cf[kBayesMaxRows];
  __shared__ float red[4][128];
  const int b = blockIdx.x, sl = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int per_slice = (Mb + kBayesSlices - 1) / kBayesSlices;
  const int s0 = sl * per_slice, s1 = s0 + per_slice < Mb ? s0 + per_slice : Mb;
  const int n = s1 > s0 ? s1 - s0 : 0;
  const float g = -(dloss ? dloss[0] : 1.f) * 0.5f / (static_cast<float>(Mb) * static_cast<float>(B));
  // phase A: a wave takes four bank rows at a time (independent load / reduction chains), lanes across the bits / the classes
  const float bk0 = lane < K ? batch[static_cast<size_t>(b) * K + lane] : 0.f;
  const float bk1 = lane + 64 < K ? batch[static_cast<size_t>(b) * K + lane + 64] : 0.f;
  const float lb0 = lane < C ? label[static_cast<size_t>(b) * C + lane] : 0.f;
  for (int mb = wid * 4; mb < n; mb += 16) {
    float dot[4], ll[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = s0 + (mb + u < n ? mb + u : n - 1);
      const float* br = bank + static_cast<size_t>(m) * K;
      const float* lr = bank_label + static_cast<size_t>(m) * C;
      float d = lane < K ? br[lane] * bk0 : 0.f;
      if (lane + 64 < K) d = fmaf(br[lane + 64], bk1, d);
      float l = lane < C ? lr[lane] * lb0 : 0.f;
      for (int c = lane + 64; c < C; c += 64) l = fmaf(lr[c], label[static_cast<size_t>(b) * C + c], l);
      dot[u] = d; ll[u] = l;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) { dot[u] += __shfl_xor(dot[u], o, 64); ll[u] += __shfl_xor(ll[u], o, 64); }
    if (lane < 4 && mb + lane < n) {
      const float d = lane == 0 ? dot[0] : (lane == 1 ? dot[1] : (lane == 2 ? dot[2] : dot[3]));
      const float l = lane == 0 ? ll[0] : (lane == 1 ? ll[1] : (lane == 2 ? ll[2] : ll[3]));
      const float sv = 0.5f * fminf(fmaxf(d, -64.f), 64.f);
      const float sig = 1.0f / (1.0f + expf(-sv));
      cf[mb + lane] = (d >= -64.f && d <= 64.f) ? g * ((l > 0.f ? 1.f : 0.f) - sig) : 0.f;
    }
  }
  __syncthreads();
  // phase B: wave w sums its quarter of the slice for every bit (four rows in flight), the partial rows meet in LDS (K <= 128)
  const int per = (n + 3) / 4, m0 = wid * per, m1 = m0 + per < n ? m0 + per : n;
  for (int k = lane; k < K; k += 64) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = m0;
    for (; m + 4 <= m1; m += 4) {
      const float v0 = bank[static_cast<size_t>(s0 + m) * K + k], v1 = bank[static_cast<size_t>(s0 + m + 1) * K + k];
      const float v2 = bank[static_cast<size_t>(s0 + m + 2) * K + k], v3 = bank[static_cast<size_t>(s0 + m + 3) * K + k];
      a0 = fmaf(cf[m], v0, a0); a1 = fmaf(cf[m + 1], v1, a1); a2 = fmaf(cf[m + 2], v2, a2); a3 = fmaf(cf[m + 3], v3, a3);
    }
    for (; m < m1; ++m) a0 = fmaf(cf[m], bank[static_cast<size_t>(s0 + m) * K + k], a0);
    red[wid][k] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    partial[(static_cast<size_t>(sl) * B + b) * K + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}
// The backward on the matrix pipe (K % 16 == 0, K, C <= 128): grid (ceil(B / 16), kBayesSlices).  Phase A = the forward's tiles, the
// coefficient of every (bank row, batch row) pair goes to LDS as cf[bank row][batch row]; phase B = dbatch tile [16 batch rows, K]
// += cf^T . bank, again 16x16x4 MFMAs: A operand = cf (lane: batch row lane&15, bank row 4 i + lane>>4), B operand = 4 bank rows x 16
// bits (a 64-byte segment per row).  A wave takes every fourth quad of bank rows; the four partial tiles meet in LDS.
constexpr int kBayesChunk = 640;      // bank rows per LDS pass (a 10 000-row bank in 16 slices: one pass)
constexpr int kBayesCfLd = 20;        // floats per cf row: 16 + 4 (lanes of one store hit 64 different banks)

__global__ __launch_bounds__(256) void bayes_bwd_mfma_kernel(const float* __restrict__ bank, const float* __restrict__ batch,
                                                             const float* __restrict__ bank_label, const float* __restrict__ label,
                                                             int Mb, int B, int K, int C, const float* __restrict__ dloss,
                                                             float* __restrict__ partial) {
  __shared__ float cf[kBayesChunk * kBayesCfLd];
  __shared__ float red[4][16][128 + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, quad = lane >> 4;
  const int b0 = blockIdx.x * 16, sl = blockIdx.y;
  const int per_slice = (Mb + kBayesSlices - 1) / kBayesSlices;
  const int s0 = sl * per_slice, s1 = s0 + per_slice < Mb ? s0 + per_slice : Mb;
  const float g = -(dloss ? dloss[0] : 1.f) * 0.5f / (static_cast<float>(Mb) * static_cast<float>(B));
  BayesOperands o;
  bayes_load_batch(o, batch, label, b0 + r16 < B ? b0 + r16 : B - 1, K, C, quad);
  const int KT = K / 16;
  m_f32x4_t dacc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) dacc[t] = m_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int c0 = s0; c0 < s1; c0 += kBayesChunk) {
    const int c1 = c0 + kBayesChunk < s1 ? c0 + kBayesChunk : s1;
    const int rows16 = (c1 - c0 + 15) & ~15;                 // phase B reads whole quads: rows past c1 hold zeros
    __syncthreads();                                         // (the previous chunk's phase B is done with cf)
    for (int mt = c0 + wid * 16; mt < c0 + rows16; mt += 64) {
      m_f32x4_t d, l;
      bayes_tile(o, bank, bank_label, mt + r16 < c1 ? mt + r16 : c1 - 1, K, C, quad, d, l);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = mt + quad * 4 + r;
        float c = 0.f;
        if (mm < c1 && b0 + r16 < B && d[r] >= -64.f && d[r] <= 64.f) {
          const float sig = 1.0f / (1.0f + expf(-0.5f * d[r]));
          c = g * ((l[r] > 0.f ? 1.f : 0.f) - sig);
        }
        cf[(mm - c0) * kBayesCfLd + r16] = c;
      }
    }
    __syncthreads();
    for (int mq = wid * 4; mq < rows16; mq += 16) {
      const int ml = mq + quad;                              // this lane's bank row of the quad
      const float a = cf[ml * kBayesCfLd + r16];
      const int m = c0 + ml < c1 ? c0 + ml : c1 - 1;         // (its coefficient is zero past c1)
      const float* br = bank + static_cast<size_t>(m) * K + r16;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < KT) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, br[t * 16], dacc[t], 0, 0, 0);
    }
  }
  // dacc[t][r] = (batch row 4 quad + r, bit 16 t + lane&15) of this wave's share
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (t < KT)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wid][quad * 4 + r][t * 16 + r16] = dacc[t][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * K; i += 256) {
    const int bb = i / K, k = i - bb * K;
    if (b0 + bb < B)
      partial[(static_cast<size_t>(sl) * B + b0 + bb) * K + k] = (red[0][bb][k] + red[1][bb][k]) + (red[2][bb][k] + red[3][bb][k]);
  }
}
// dbatch[b,k] = sum over the slices, in slice order
__global__ __launch_bounds__(256) void bayes_bwd_reduce_kernel(const float* __restrict__ partial, int n, float* __restrict__ dbatch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int sl = 0; sl < kBayesSlices; ++sl) acc += partial[static_cast<size_t>(sl) * n + i];
  dbatch[i] = acc;
}

// InfoNCE: logsumexp of every row of a against its group of b rows (the forward's row_ce_kernel, keeping the value per row)
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ a, const float* __restrict__ b, int R, int G, int D,
                                                      float inv_temp, float* __restrict__ lse) {
  extern __shared__ float sa[];   // 4 waves x D
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wid;
  if (i >= R) return;
  const int g0 = (i / G) * G;
  float* ar = sa + wid * D;
  for (int d = lane; d < D; d += 64) ar[d] = a[static_cast<size_t>(i) * D + d];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  float mx = -1e30f;
  for (int j = lane; j < G; j += 64) {
    const float* br = b + static_cast<size_t>(g0 + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(ar[d], br[d], s);
    mx = fmaxf(mx, s * inv_temp);
  }
  mx = m_wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < G; j += 64) {
    const float* br = b + static_cast<size_t>(g0 + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(ar[d], br[d], s);
    se += expf(s * inv_temp - mx);
  }
  se = m_wave_sum(se);
  if (lane == 0) lse[i] = mx + logf(se);
}
// d/dx_i of 0.5/R [sum_i CE(x_i . y / T) + sum_i CE(y_i . x / T)] (diagonal targets inside the group):
//   dx_i = scale [ sum_j (exp(s_ij - lse_x[i]) + exp(s_ij - lse_y[j])) y_j - 2 y_i ],  s_ij = x_i . y_j / T,  scale = g * 0.5 / (R T)
// lse_x[i] = logsumexp_j s_ij, lse_y[j] = logsumexp_i s_ij (rows of y against x).  One wave per row, weights through LDS.
__global__ __launch_bounds__(256) void info_nce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ lse_x, const float* __restrict__ lse_y, int R,
                                                           int G, int D, float inv_temp, const float* __restrict__ dloss,
                                                           float* __restrict__ dx) {
  extern __shared__ float sm[];   // 4 waves x (D + G)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wid;
  if (i >= R) return;
  const int g0 = (i / G) * G;
  float* xr = sm + wid * (D + G);
  float* wj = xr + D;
  for (int d = lane; d < D; d += 64) xr[d] = x[static_cast<size_t>(i) * D + d];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const float li = lse_x[i];
  for (int j = lane; j < G; j += 64) {
    const float* yr = y + static_cast<size_t>(g0 + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(xr[d], yr[d], s);
    s *= inv_temp;
    wj[j] = expf(s - li) + expf(s - lse_y[g0 + j]) - (g0 + j == i ? 2.f : 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const float scale = (dloss ? dloss[0] : 1.f) * 0.5f * inv_temp / static_cast<float>(R);
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < G; ++j) acc = fmaf(wj[j], y[static_cast<size_t>(g0 + j) * D + d], acc);
    dx[static_cast<size_t>(i) * D + d] = acc * scale;
  }
}

// ---- InfoNCE on the f32 matrix pipe (D % 64 == 0 and a workspace for the score matrix) ---------------------------------------------------
// The kernels above give every candidate row to ONE lane that walks its D products alone, recompute the scores in a second pass and once
// more per direction (token-level term of configs[2]: 16 384 rows in groups of 64, D = 512: ~0.5 ms per launch, six launches per step).
// Here the scores S = a_i . b_j / T of a group are formed once, as 16 x 16 tiles of k-ordered f32 FMA chains (v_mfma_f32_16x16x4_f32; a
// lane loads 4 consecutive k of its row and feeds them to four MFMAs, so the k order inside a block of 16 is permuted - both operands
// alike), both directions' log-sum-exps are row / column reductions of S, and the gradients are S-shaped weights times the other operand,
// tiled the same way.
__device__ __forceinline__ m_f32x4_t m_ld4(const float* p) { return *reinterpret_cast<const m_f32x4_t*>(p); }

// one wave per 16 x 16 tile of a group's G x G scores
__global__ __launch_bounds__(256) void nce_scores_kernel(const float* __restrict__ a, const float* __restrict__ b, int R, int G, int D,
                                                         float inv_temp, float* __restrict__ S) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, quad = lane >> 4;
  const int tg = (G + 15) / 16;
  const long tile = static_cast<long>(blockIdx.x) * 4 + wid, total = static_cast<long>(R / G) * tg * tg;
  if (tile >= total) return;
  const int grp = static_cast<int>(tile / (tg * tg)), rem = static_cast<int>(tile - static_cast<long>(grp) * tg * tg);
  const int it = rem / tg, jt = rem - it * tg, g0 = grp * G;
  const int ia = it * 16 + r16 < G ? it * 16 + r16 : G - 1, jb = jt * 16 + r16 < G ? jt * 16 + r16 : G - 1;
  const float* ar = a + static_cast<size_t>(g0 + ia) * D + 4 * quad;
  const float* br = b + static_cast<size_t>(g0 + jb) * D + 4 * quad;
  m_f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = 0; k0 < D; k0 += 16) {
    const m_f32x4_t av = m_ld4(ar + k0), bv = m_ld4(br + k0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
  }
  const int j = jt * 16 + r16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = it * 16 + quad * 4 + r;
    if (i < G && j < G) S[static_cast<size_t>(g0 + i) * G + j] = acc[r] * inv_temp;
  }
}

// grid (groups, ceil(G / 64)): block (grp, sb) takes rows and columns [64 sb, 64 sb + 64) of the group: lse_row[i] = logsumexp_j S_ij,
// lse_col[j] = logsumexp_i S_ij (either may be NULL), acc += (lse_row[i] - S_ii) + (lse_col[i] - S_ii)
__global__ __launch_bounds__(256) void nce_lse_kernel(const float* __restrict__ S, int G, float* __restrict__ lse_row,
                                                      float* __restrict__ lse_col, double* __restrict__ acc) {
  __shared__ double sh[4];
  __shared__ float cmx[4][64], cse[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = blockIdx.x, sb = blockIdx.y;
  const float* Sg = S + static_cast<size_t>(grp) * G * G;
  double v = 0.0;
  for (int i = sb * 64 + wid; i < sb * 64 + 64 && i < G; i += 4) {
    const float* row = Sg + static_cast<size_t>(i) * G;
    float mx = -1e30f;
    for (int j = lane; j < G; j += 64) mx = fmaxf(mx, row[j]);
    mx = m_wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < G; j += 64) se += expf(row[j] - mx);
    se = m_wave_sum(se);
    const float l = mx + logf(se);
    if (lane == 0) {
      if (lse_row) lse_row[static_cast<size_t>(grp) * G + i] = l;
      v += static_cast<double>(l - row[i]);
    }
  }
  // columns: lane = column, the four waves take a quarter of the rows each, the partial (max, sum) pairs meet in LDS
  const int j = sb * 64 + lane;
  const int per = (G + 3) / 4, i0 = wid * per, i1 = i0 + per < G ? i0 + per : G;
  float mx = -1e30f, se = 0.f;
  if (j < G) {
    for (int i = i0; i < i1; ++i) mx = fmaxf(mx, Sg[static_cast<size_t>(i) * G + j]);
    for (int i = i0; i < i1; ++i) se += expf(Sg[static_cast<size_t>(i) * G + j] - mx);
  }
  cmx[wid][lane] = mx;
  cse[wid][lane] = se;
  __syncthreads();
  if (wid == 0 && j < G) {
    const float m4 = fmaxf(fmaxf(cmx[0][lane], cmx[1][lane]), fmaxf(cmx[2][lane], cmx[3][lane]));
    float s4 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s4 += cse[w][lane] * expf(cmx[w][lane] - m4);   // (an empty quarter: 0 * exp(-1e30 - m4) = 0)
    const float l = m4 + logf(s4);
    if (lse_col) lse_col[static_cast<size_t>(grp) * G + j] = l;
    v += static_cast<double>(l - Sg[static_cast<size_t>(j) * G + j]);
  }
  const double t = m_block_sum(v, sh);
  if (acc && threadIdx.x == 0 && t != 0.0) atomicAdd(acc, t);
}

// S -> W in place: W_ij = exp(S_ij - lse_row[i]) + exp(S_ij - lse_col[j]) - 2 [i == j]
__global__ __launch_bounds__(256) void nce_weights_kernel(float* __restrict__ S, const float* __restrict__ lse_row,
                                                          const float* __restrict__ lse_col, int G, size_t n) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= n) return;
  const size_t i = idx / G;                 // global row
  const int j = static_cast<int>(idx - i * G);
  const size_t g0 = (i / G) * G;
  const float sv = S[idx];
  S[idx] = expf(sv - lse_row[i]) + expf(sv - lse_col[g0 + j]) - (i - g0 == static_cast<size_t>(j) ? 2.f : 0.f);
}

// out[g0 + i, :] = scale * sum_j w(i, j) y[g0 + j, :], w(i, j) = W[g0 + i][j] (TR = false) or W[g0 + j][i] (TR = true);
// one wave per 16 rows x 64 columns
template <bool TR>
__global__ __launch_bounds__(256) void nce_grad_kernel(const float* __restrict__ W, const float* __restrict__ y, int R, int G, int D,
                                                       const float* __restrict__ dloss, float scale0, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r16 = lane & 15, quad = lane >> 4;
  const int tg = (G + 15) / 16, dq = D / 64;
  const long wv = static_cast<long>(blockIdx.x) * 4 + wid, total = static_cast<long>(R / G) * tg * dq;
  if (wv >= total) return;
  const int grp = static_cast<int>(wv / (tg * dq)), rem = static_cast<int>(wv - static_cast<long>(grp) * tg * dq);
  const int it = rem / dq, d0 = (rem - it * dq) * 64, g0 = grp * G;
  const int ii = it * 16 + r16 < G ? it * 16 + r16 : G - 1;
  m_f32x4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = m_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < G; j0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int jj = j0 + 4 * quad + e;
      const int jc = jj < G ? jj : G - 1;
      float wv_ = TR ? W[static_cast<size_t>(g0 + jc) * G + ii] : W[static_cast<size_t>(g0 + ii) * G + jc];
      if (jj >= G) wv_ = 0.f;
      const float* yr = y + static_cast<size_t>(g0 + jc) * D + d0 + r16;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv_, yr[16 * t], acc[t], 0, 0, 0);
    }
  }
  const float scale = (dloss ? dloss[0] : 1.f) * scale0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = it * 16 + quad * 4 + r;
    if (i < G)
#pragma unroll
      for (int t = 0; t < 4; ++t) out[static_cast<size_t>(g0 + i) * D + d0 + 16 * t + r16] = acc[t][r] * scale;
  }
}

// sum (a - b)^2: da = 2 g (a - b), db = -da (either may be NULL)
__global__ __launch_bounds__(256) void sq_diff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                          const float* __restrict__ dloss, float* __restrict__ da, float* __restrict__ db) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = 2.f * (dloss ? dloss[0] : 1.f) * (a[i] - b[i]);
  if (da) da[i] = v;
  if (db) db[i] = -v;
}

// ---- backward pieces of the HashingModel (the similarities that steer the aggregation are detached upstream, model/MITH.py:345,
// so the aggregation is linear in the tokens) ------------------------------------------------------------------------------------
// dtokens[b, l0 + l, :] = sum_k w[l, k] * dmerge[b, k, :]; grid (B, ceil(D / 256))
__global__ __launch_bounds__(256) void lta_bwd_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ kpm,
                                                      const float* __restrict__ dmerge, float* __restrict__ dtokens, int Ltot, int l0,
                                                      int L, int K, int D, int top_k) {
  __shared__ __attribute__((aligned(16))) float w[kLtaMaxL][kLtaMaxK + 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t row0 = static_cast<size_t>(b) * Ltot + l0;
  lta_build_weights(w, sim, kpm, b, row0, L, K, top_k);
  const int d = blockIdx.y * 256 + tid;
  if (d >= D) return;
  float g[kLtaMaxK];
#pragma unroll
  for (int k = 0; k < kLtaMaxK; ++k) g[k] = k < K ? dmerge[(static_cast<size_t>(b) * K + k) * D + d] : 0.f;
  for (int l = 0; l < L; ++l) {
    float acc = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < kLtaMaxK / 4; ++k4)
      if (k4 * 4 < K) {
        const float4 wv = *reinterpret_cast<const float4*>(&w[l][k4 * 4]);
        acc = fmaf(wv.x, g[k4 * 4], acc); acc = fmaf(wv.y, g[k4 * 4 + 1], acc);
        acc = fmaf(wv.z, g[k4 * 4 + 2], acc); acc = fmaf(wv.w, g[k4 * 4 + 3], acc);
      }
    dtokens[(row0 + l) * D + d] = acc;
  }
}

// exact GELU (nn.GELU(): 0.5 x (1 + erf(x / sqrt 2))) and its derivative 0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = gelu_erf(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                       int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  dx[i] = dy[i] * (0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * expf(-0.5f * v * v) * 0.3989422804014327f);
}

// F.normalize(x, dim=-1) backward: y = x / max(|x|, 1e-12);  dx = (dy - y (y . dy)) / max(|x|, 1e-12).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int R, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xr = x + static_cast<size_t>(row) * D;
  const float* gr = dy + static_cast<size_t>(row) * D;
  float ss = 0.f, sg = 0.f;
  for (int d = lane; d < D; d += 64) { ss = fmaf(xr[d], xr[d], ss); sg = fmaf(xr[d], gr[d], sg); }
  ss = m_wave_sum(ss); sg = m_wave_sum(sg);
  const float nrm = sqrtf(ss);
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);
  // below the eps clamp y = x / eps is linear: dx = dy / eps
  const float proj = nrm > 1e-12f ? sg * inv * inv : 0.f;          // (y . dy) / |x| = (x . dy) / |x|^2 ... times x / |x| below
  for (int d = lane; d < D; d += 64) dx[static_cast<size_t>(row) * D + d] = (gr[d] - xr[d] * proj) * inv;
}

// BitwiseHashing backward: y[n,k] = tanh(x[n,k,:] . w[k,:] + b[k]).  dz = dy (1 - y^2);
//   dx[n,k,:] = dz[n,k] w[k,:]  (grid (N*K) waves), dw[k,:] = sum_n dz[n,k] x[n,k,:], db[k] = sum_n dz[n,k]  (grid K workgroups)
__global__ __launch_bounds__(256) void bithash_dx_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                         const float* __restrict__ dy, float* __restrict__ dx, int NK, int K, int D) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= NK) return;
  const int k = i % K;
  const float dz = dy[i] * (1.0f - y[i] * y[i]);
  for (int d = lane; d < D; d += 64) dx[static_cast<size_t>(i) * D + d] = dz * w[static_cast<size_t>(k) * D + d];
}
// grid (K, ceil(D / 64)): 64 feature columns per workgroup, the four waves take a quarter of the samples each (four chains per thread),
// the quarter sums meet in LDS in wave order (round 1: K workgroups of one 256-long chain per thread, 204 us at 256 x 64 x 512)
__global__ __launch_bounds__(256) void bithash_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                                                         int Nb, int K, int D) {
  __shared__ float red[4][64];
  __shared__ float redb[4];
  const int k = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int d = blockIdx.y * 64 + lane;
  const int per = (Nb + 3) / 4, n0 = wid * per, n1 = n0 + per < Nb ? n0 + per : Nb;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (d < D) {
    int n = n0;
    for (; n + 4 <= n1; n += 4) {
      const size_t i0 = static_cast<size_t>(n) * K + k, i1 = i0 + K, i2 = i1 + K, i3 = i2 + K;
      a0 = fmaf(dy[i0] * (1.0f - y[i0] * y[i0]), x[i0 * D + d], a0);
      a1 = fmaf(dy[i1] * (1.0f - y[i1] * y[i1]), x[i1 * D + d], a1);
      a2 = fmaf(dy[i2] * (1.0f - y[i2] * y[i2]), x[i2 * D + d], a2);
      a3 = fmaf(dy[i3] * (1.0f - y[i3] * y[i3]), x[i3 * D + d], a3);
    }
    for (; n < n1; ++n) {
      const size_t i = static_cast<size_t>(n) * K + k;
      a0 = fmaf(dy[i] * (1.0f - y[i] * y[i]), x[i * D + d], a0);
    }
  }
  red[wid][lane] = (a0 + a1) + (a2 + a3);
  if (blockIdx.y == 0) {                       // db[k]: lanes across the wave's samples, wave sums in wave order
    float sb = 0.f;
    for (int n = n0 + lane; n < n1; n += 64) { const size_t i = static_cast<size_t>(n) * K + k; sb += dy[i] * (1.0f - y[i] * y[i]); }
    sb = m_wave_sum(sb);
    if (lane == 0) redb[wid] = sb;
  }
  __syncthreads();
  if (wid == 0 && d < D) dw[static_cast<size_t>(k) * D + d] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (blockIdx.y == 0 && threadIdx.x == 0) db[k] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_mith_lta(const float* tokens, const float* sim, const uint8_t* key_padding_mask, float* out, int32_t B,
                            int32_t Ltot, int32_t l0, int32_t L, int32_t K, int32_t D, int32_t top_k, void* stream) {
  CMH_CHECK_ARG(tokens && sim && out, "mith_lta: null pointer");
  CMH_CHECK_ARG(B > 0 && L > 0 && L <= kLtaMaxL && l0 >= 0 && l0 + L <= Ltot, "mith_lta: token range (L <= %d)", kLtaMaxL);
  CMH_CHECK_ARG(K > 0 && K <= kLtaMaxK && D > 0, "mith_lta: K must be <= %d", kLtaMaxK);
  CMH_CHECK_ARG(top_k >= 1 && top_k <= kTop, "mith_lta: top_k must be in 1..%d", kTop);
  CMH_CHECK_ARG(K % 4 == 0, "mith_lta: K=%d must be a multiple of 4", K);
  hipLaunchKernelGGL(lta_kernel, dim3(B, (D + 255) / 256), dim3(256), 0, as_stream(stream), tokens, sim, key_padding_mask, out, Ltot,
                     l0, L, K, D, top_k);
  CMH_CHECK_LAUNCH("mith_lta");
  return CMH_OK;
}

extern "C" int cmh_add_positional(float* x, const float* pe, int32_t B, int32_t T, int32_t D, void* stream) {
  CMH_CHECK_ARG(x && pe && B > 0 && T > 0 && D > 0, "add_positional: bad arguments");
  const int64_t n = static_cast<int64_t>(B) * T * D;
  hipLaunchKernelGGL(add_pos_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, pe, n, T, D);
  CMH_CHECK_LAUNCH("add_positional");
  return CMH_OK;
}

extern "C" int cmh_bitwise_hash(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t K,
                                int32_t D, void* stream) {
  CMH_CHECK_ARG(x && w && bias && out && B > 0 && K > 0 && D > 0 && D % 4 == 0, "bitwise_hash: bad arguments");
  hipLaunchKernelGGL(bitwise_hash_kernel, dim3((B * K + 3) / 4), dim3(256), 0, as_stream(stream), x, w, bias, out, B, K, D);
  CMH_CHECK_LAUNCH("bitwise_hash");
  return CMH_OK;
}

extern "C" int cmh_l2_normalize_rows(const float* x, float* y, int32_t R, int32_t D, void* stream) {
  CMH_CHECK_ARG(x && y && R > 0 && D > 0, "l2_normalize_rows: bad arguments");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, as_stream(stream), x, y, R, D);
  CMH_CHECK_LAUNCH("l2_normalize_rows");
  return CMH_OK;
}

extern "C" int cmh_mith_mix(const float* img_cls, const float* img_tok, const float* txt_cls, const float* txt_tok,
                            float hyper_lambda, float* B_codes, float* H_img, float* H_txt, int64_t n, void* stream) {
  CMH_CHECK_ARG(img_cls && img_tok && txt_cls && txt_tok && B_codes && H_img && H_txt && n > 0, "mith_mix: bad arguments");
  hipLaunchKernelGGL(mith_mix_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), img_cls,
                     img_tok, txt_cls, txt_tok, hyper_lambda, B_codes, H_img, H_txt, n);
  CMH_CHECK_LAUNCH("mith_mix");
  return CMH_OK;
}

static bool bayes_mfma_off() {   // CMH_BAYES_MFMA=0: the butterfly kernels (A/B, and the shapes the tiles do not cover)
  static const bool off = []() { const char* e = getenv("CMH_BAYES_MFMA"); return e && e[0] == '0'; }();
  return off;
}

extern "C" int cmh_sq_diff_sum(const float* a, const float* b, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                               void* stream) {
  CMH_CHECK_ARG(a && b && out && workspace && n > 0 && workspace_bytes >= 256, "sq_diff_sum: bad arguments");
  hipStream_t st = as_stream(stream);
  double* acc = static_cast<double*>(workspace);
  if (hipMemsetAsync(acc, 0, 8, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "sq_diff_sum: memset failed");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(sq_diff_kernel, dim3(static_cast<unsigned>(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, a, b, n, acc);
  hipLaunchKernelGGL(read_acc_kernel, dim3(1), dim3(1), 0, st, acc, 1.0, out);
  CMH_CHECK_LAUNCH("sq_diff_sum");
  return CMH_OK;
}

extern "C" int cmh_mith_bayesian_loss(const float* bank, const float* batch, const float* bank_label, const float* label,
                                      int32_t Mb, int32_t B, int32_t K, int32_t C, float* out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(bank && batch && bank_label && label && out && workspace && workspace_bytes >= 256, "mith_bayesian_loss: bad arguments");
  CMH_CHECK_ARG(Mb > 0 && B > 0 && K > 0 && C > 0, "mith_bayesian_loss: bad shape");
  hipStream_t st = as_stream(stream);
  double* acc = static_cast<double*>(workspace);
  if (hipMemsetAsync(acc, 0, 8, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "mith_bayesian_loss: memset failed");
  const int64_t n = static_cast<int64_t>(Mb) * B;
  if (K <= 128 && C <= 128 && !bayes_mfma_off())
    hipLaunchKernelGGL(bayes_mfma_kernel, dim3((B + 15) / 16, Mb >= 1024 ? 16 : 1), dim3(256), 0, st, bank, batch, bank_label, label, Mb, B, K, C, acc);
  else
    hipLaunchKernelGGL(bayes_kernel, dim3(B, Mb >= 1024 ? 16 : 1), dim3(256), 0, st, bank, batch, bank_label, label, Mb, B, K, C, acc);
  hipLaunchKernelGGL(read_acc_kernel, dim3(1), dim3(1), 0, st, acc, -1.0 / static_cast<double>(n), out);
  CMH_CHECK_LAUNCH("mith_bayesian_loss");
  return CMH_OK;
}

// workspace of the matrix-pipe path of cmh_info_nce / cmh_info_nce_backward: 256 bytes of accumulators, the R x G scores, two R-vectors
extern "C" size_t cmh_info_nce_workspace_bytes(int32_t R, int32_t G) {
  return R > 0 && G > 0 ? 256 + (static_cast<size_t>(R) * G + 2 * static_cast<size_t>(R)) * 4 + 256 : 0;
}

extern "C" int cmh_info_nce(const float* a, const float* b, int32_t R, int32_t G, int32_t D, float temperature, float* out,
                            void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(a && b && out && workspace && workspace_bytes >= 256, "info_nce: bad arguments");
  CMH_CHECK_ARG(R > 0 && G > 0 && R % G == 0 && D > 0 && D <= 4096 && temperature > 0.f, "info_nce: bad shape");
  hipStream_t st = as_stream(stream);
  double* acc = static_cast<double*>(workspace);
  if (hipMemsetAsync(acc, 0, 8, st) != hipSuccess) return fail(CMH_ERR_LAUNCH, "info_nce: memset failed");
  if (D % 64 == 0 && workspace_bytes >= cmh_info_nce_workspace_bytes(R, G) && !bayes_mfma_off()) {
    float* S = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    const long tiles = static_cast<long>(R / G) * ((G + 15) / 16) * ((G + 15) / 16);
    hipLaunchKernelGGL(nce_scores_kernel, dim3(static_cast<unsigned>((tiles + 3) / 4)), dim3(256), 0, st, a, b, R, G, D, 1.0f / temperature, S);
    hipLaunchKernelGGL(nce_lse_kernel, dim3(R / G, (G + 63) / 64), dim3(256), 0, st, S, G, static_cast<float*>(nullptr),
                       static_cast<float*>(nullptr), acc);
  } else {
    const size_t smem = static_cast<size_t>(4) * D * sizeof(float);
    hipLaunchKernelGGL(row_ce_kernel, dim3((R + 3) / 4), dim3(256), smem, st, a, b, R, G, D, 1.0f / temperature, acc);
    hipLaunchKernelGGL(row_ce_kernel, dim3((R + 3) / 4), dim3(256), smem, st, b, a, R, G, D, 1.0f / temperature, acc);
  }
  hipLaunchKernelGGL(read_acc_kernel, dim3(1), dim3(1), 0, st, acc, 0.5 / static_cast<double>(R), out);
  CMH_CHECK_LAUNCH("info_nce");
  return CMH_OK;
}

extern "C" int cmh_mith_lta_backward(const float* sim, const uint8_t* key_padding_mask, const float* dmerge, float* dtokens, int32_t B,
                                     int32_t Ltot, int32_t l0, int32_t L, int32_t K, int32_t D, int32_t top_k, void* stream) {
  CMH_CHECK_ARG(sim && dmerge && dtokens, "mith_lta_backward: null pointer");
  CMH_CHECK_ARG(B > 0 && L > 0 && L <= kLtaMaxL && l0 >= 0 && l0 + L <= Ltot, "mith_lta_backward: token range (L <= %d)", kLtaMaxL);
  CMH_CHECK_ARG(K > 0 && K <= kLtaMaxK && K % 4 == 0 && D > 0, "mith_lta_backward: K must be a multiple of 4, <= %d", kLtaMaxK);
  CMH_CHECK_ARG(top_k >= 1 && top_k <= kTop, "mith_lta_backward: top_k must be in 1..%d", kTop);
  hipStream_t st = as_stream(stream);
  if (L != Ltot && hipMemsetAsync(dtokens, 0, static_cast<size_t>(B) * Ltot * D * 4, st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "mith_lta_backward: memset failed");
  hipLaunchKernelGGL(lta_bwd_kernel, dim3(B, (D + 255) / 256), dim3(256), 0, st, sim, key_padding_mask, dmerge, dtokens, Ltot, l0, L, K,
                     D, top_k);
  CMH_CHECK_LAUNCH("mith_lta_backward");
  return CMH_OK;
}

extern "C" int cmh_gelu(const float* x, float* y, int64_t n, void* stream) {
  CMH_CHECK_ARG(x && y && n > 0, "gelu: bad arguments");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, n);
  CMH_CHECK_LAUNCH("gelu");
  return CMH_OK;
}

extern "C" int cmh_gelu_backward(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  CMH_CHECK_ARG(x && dy && dx && n > 0, "gelu_backward: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, dy, dx, n);
  CMH_CHECK_LAUNCH("gelu_backward");
  return CMH_OK;
}

extern "C" int cmh_l2_normalize_backward(const float* x, const float* dy, float* dx, int32_t R, int32_t D, void* stream) {
  CMH_CHECK_ARG(x && dy && dx && R > 0 && D > 0, "l2_normalize_backward: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, as_stream(stream), x, dy, dx, R, D);
  CMH_CHECK_LAUNCH("l2_normalize_backward");
  return CMH_OK;
}

extern "C" int cmh_bitwise_hash_backward(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                                         float* db, int32_t B, int32_t K, int32_t D, void* stream) {
  CMH_CHECK_ARG(x && w && y && dy && dx && dw && db && B > 0 && K > 0 && D > 0, "bitwise_hash_backward: bad arguments");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(bithash_dx_kernel, dim3((B * K + 3) / 4), dim3(256), 0, st, w, y, dy, dx, B * K, K, D);
  hipLaunchKernelGGL(bithash_dw_kernel, dim3(K, (D + 63) / 64), dim3(256), 0, st, x, y, dy, dw, db, B, K, D);
  CMH_CHECK_LAUNCH("bitwise_hash_backward");
  return CMH_OK;
}

extern "C" size_t cmh_mith_bayesian_backward_workspace_bytes(int32_t B, int32_t K) {
  return B > 0 && K > 0 ? static_cast<size_t>(kBayesSlices) * B * K * 4 + 256 : 0;
}

extern "C" int cmh_mith_bayesian_loss_backward(const float* bank, const float* batch, const float* bank_label, const float* label,
                                               int32_t Mb, int32_t B, int32_t K, int32_t C, const float* dloss, float* dbatch,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(bank && batch && bank_label && label && dbatch && workspace, "mith_bayesian_loss_backward: null pointer");
  CMH_CHECK_ARG(Mb > 0 && B > 0 && K > 0 && K <= 128 && C > 0, "mith_bayesian_loss_backward: bad shape (K <= 128)");
  CMH_CHECK_ARG((Mb + kBayesSlices - 1) / kBayesSlices <= kBayesMaxRows, "mith_bayesian_loss_backward: bank of %d rows is too large", Mb);
  if (workspace_bytes < cmh_mith_bayesian_backward_workspace_bytes(B, K)) return fail(CMH_ERR_WORKSPACE, "mith_bayesian_loss_backward: workspace too small");
  float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  hipStream_t st = as_stream(stream);
  if (K % 16 == 0 && C <= 128 && !bayes_mfma_off())
    hipLaunchKernelGGL(bayes_bwd_mfma_kernel, dim3((B + 15) / 16, kBayesSlices), dim3(256), 0, st, bank, batch, bank_label, label, Mb, B, K, C, dloss, partial);
  else
    hipLaunchKernelGGL(bayes_bwd_kernel, dim3(B, kBayesSlices), dim3(256), 0, st, bank, batch, bank_label, label, Mb, B, K, C, dloss, partial);
  hipLaunchKernelGGL(bayes_bwd_reduce_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, partial, B * K, dbatch);
  CMH_CHECK_LAUNCH("mith_bayesian_loss_backward");
  return CMH_OK;
}

extern "C" int cmh_info_nce_backward(const float* a, const float* b, int32_t R, int32_t G, int32_t D, float temperature,
                                     const float* dloss, float* da, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(a && b && da && db && workspace, "info_nce_backward: null pointer");
  CMH_CHECK_ARG(R > 0 && G > 0 && R % G == 0 && D > 0 && temperature > 0.f, "info_nce_backward: bad shape R=%d G=%d D=%d", R, G, D);
  CMH_CHECK_ARG(static_cast<size_t>(4) * (D + G) * 4 <= 64 * 1024, "info_nce_backward: D + G = %d too large for the LDS rows", D + G);
  if (workspace_bytes < static_cast<size_t>(2) * R * 4 + 256) return fail(CMH_ERR_WORKSPACE, "info_nce_backward: workspace must hold 2 R floats");
  float* lse_a = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  float* lse_b = lse_a + R;
  hipStream_t st = as_stream(stream);
  const float it = 1.0f / temperature;
  if (D % 64 == 0 && workspace_bytes >= cmh_info_nce_workspace_bytes(R, G) && !bayes_mfma_off()) {
    float* S = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    float* lr = S + static_cast<size_t>(R) * G;
    float* lc = lr + R;
    const int tg = (G + 15) / 16;
    const long tiles = static_cast<long>(R / G) * tg * tg, gw = static_cast<long>(R / G) * tg * (D / 64);
    const size_t n = static_cast<size_t>(R) * G;
    hipLaunchKernelGGL(nce_scores_kernel, dim3(static_cast<unsigned>((tiles + 3) / 4)), dim3(256), 0, st, a, b, R, G, D, it, S);
    hipLaunchKernelGGL(nce_lse_kernel, dim3(R / G, (G + 63) / 64), dim3(256), 0, st, S, G, lr, lc, static_cast<double*>(nullptr));
    hipLaunchKernelGGL(nce_weights_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, S, lr, lc, G, n);
    const float scale0 = 0.5f * it / static_cast<float>(R);
    hipLaunchKernelGGL(nce_grad_kernel<false>, dim3(static_cast<unsigned>((gw + 3) / 4)), dim3(256), 0, st, S, b, R, G, D, dloss, scale0, da);
    hipLaunchKernelGGL(nce_grad_kernel<true>, dim3(static_cast<unsigned>((gw + 3) / 4)), dim3(256), 0, st, S, a, R, G, D, dloss, scale0, db);
    CMH_CHECK_LAUNCH("info_nce_backward");
    return CMH_OK;
  }
  const dim3 grid((R + 3) / 4);
  hipLaunchKernelGGL(row_lse_kernel, grid, dim3(256), static_cast<size_t>(4) * D * 4, st, a, b, R, G, D, it, lse_a);
  hipLaunchKernelGGL(row_lse_kernel, grid, dim3(256), static_cast<size_t>(4) * D * 4, st, b, a, R, G, D, it, lse_b);
  hipLaunchKernelGGL(info_nce_bwd_kernel, grid, dim3(256), static_cast<size_t>(4) * (D + G) * 4, st, a, b, lse_a, lse_b, R, G, D, it, dloss, da);
  hipLaunchKernelGGL(info_nce_bwd_kernel, grid, dim3(256), static_cast<size_t>(4) * (D + G) * 4, st, b, a, lse_b, lse_a, R, G, D, it, dloss, db);
  CMH_CHECK_LAUNCH("info_nce_backward");
  return CMH_OK;
}

extern "C" int cmh_sq_diff_sum_backward(const float* a, const float* b, int64_t n, const float* dloss, float* da, float* db, void* stream) {
  CMH_CHECK_ARG(a && b && n > 0 && (da || db), "sq_diff_sum_backward: bad arguments");
  hipLaunchKernelGGL(sq_diff_bwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, n, dloss, da, db);
  CMH_CHECK_LAUNCH("sq_diff_sum_backward");
  return CMH_OK;
}
