// Multi-head attention backward for the towers' packed qkv layout (nn.MultiheadAttention inside ResidualAttentionBlock,
// model/base/model.py:171-189): given qkv [B*T, 3d], the forward output o [B*T, d] and do [B*T, d], produce dqkv [B*T, 3d].
// Per (batch, head), hd = 64, scale = 1/8:
//   P = softmax(scale * Q K^T + mask)        (recomputed, nothing but qkv and o is saved)
//   dV = P^T dO;   D_i = dO_i . O_i (= sum_j P_ij dP_ij);   dS = scale * P o (dO V^T - D);   dQ = dS K;   dK = dS^T Q
// First version: fp32 VALU, one 256-thread workgroup per (batch, head), T <= 128, everything staged in LDS
// (two [T][65] operand buffers + one [T][T+1] score buffer, <= 132 KB).  Inputs f32 or bf16, accumulation f32, outputs in the
// input type.  An MFMA version for the bf16 mode is the next step (this one costs ~0.2-0.3 ms per layer at batch 256).
#include "cmh_common.h"

namespace cmh {

constexpr int HDB = 64;

template <typename T> __device__ __forceinline__ float ab_ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ab_ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ab_ld<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void ab_st(T* p, size_t i, float v);
template <> __device__ __forceinline__ void ab_st<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void ab_st<bf16_t>(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                            const T* __restrict__ dout, T* __restrict__ dqkv, int B, int Tn,
                                                            int d, int causal, const uint8_t* __restrict__ kpm) {
  extern __shared__ float smem[];
  float* bufA = smem;                       // [Tn][65]
  float* bufB = bufA + Tn * 65;             // [Tn][65]
  float* S = bufB + Tn * 65;                // [Tn][Tn+1]
  float* Dv = S + Tn * (Tn + 1);            // [Tn]
  const int ST = Tn + 1;
  const int heads = d / HDB;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const size_t ld = static_cast<size_t>(3) * d;
  const size_t rowq = static_cast<size_t>(b) * Tn;
  const T* qb = qkv + rowq * ld + h * HDB;            // q: +0, k: +d, v: +2d
  const T* ob = o + rowq * d + h * HDB;
  const T* dob = dout + rowq * d + h * HDB;
  T* dqb = dqkv + rowq * ld + h * HDB;
  const int tid = threadIdx.x;

  auto load_tile = [&](float* buf, const T* base, size_t stride) {
    for (int idx = tid; idx < Tn * HDB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      buf[r * 65 + c] = ab_ld<T>(base, static_cast<size_t>(r) * stride + c);
    }
  };
  // ---- 1. S = scale * Q K^T, masked ------------------------------------------------------------------------------------
  load_tile(bufA, qb, ld);
  load_tile(bufB, qb + d, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * Tn; idx += 256) {
    const int i = idx / Tn, j = idx - i * Tn;
    float acc = 0.f;
#pragma unroll 16
    for (int c = 0; c < HDB; ++c) acc += bufA[i * 65 + c] * bufB[j * 65 + c];
    bool ok = !(causal && j > i);
    if (ok && kpm) ok = kpm[static_cast<size_t>(b) * Tn + j] == 0;
    S[i * ST + j] = ok ? acc * 0.125f : -1e30f;
  }
  __syncthreads();
  // ---- 2. row softmax (one wave per row) -------------------------------------------------------------------------------
  {
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < Tn; i += 4) {
      float m = -1e30f;
      for (int j = lane; j < Tn; j += 64) m = fmaxf(m, S[i * ST + j]);
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) m = fmaxf(m, __shfl_xor(m, of, 64));
      float l = 0.f;
      for (int j = lane; j < Tn; j += 64) {
        const float s = S[i * ST + j];
        const float p = s > -1e29f ? expf(s - m) : 0.f;
        S[i * ST + j] = p;
        l += p;
      }
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) l += __shfl_xor(l, of, 64);
      const float inv = 1.0f / l;
      for (int j = lane; j < Tn; j += 64) S[i * ST + j] *= inv;
    }
  }
  __syncthreads();
  // ---- 3. A <- dO, B <- V;  dV = P^T dO;  D_i = dO_i . O_i;  dS = scale * P o (dO V^T - D) ------------------------------
  load_tile(bufA, dob, d);
  load_tile(bufB, qb + 2 * d, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * HDB; idx += 256) {           // dV[j][c]
    const int j = idx >> 6, c = idx & 63;
    float acc = 0.f;
    for (int i = 0; i < Tn; ++i) acc += S[i * ST + j] * bufA[i * 65 + c];
    ab_st<T>(dqb + 2 * d, static_cast<size_t>(j) * ld + c, acc);
  }
  {
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < Tn; i += 4) {
      float v = bufA[i * 65 + lane] * ab_ld<T>(ob, static_cast<size_t>(i) * d + lane);
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) v += __shfl_xor(v, of, 64);
      if (lane == 0) Dv[i] = v;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < Tn * Tn; idx += 256) {
    const int i = idx / Tn, j = idx - i * Tn;
    float acc = 0.f;
#pragma unroll 16
    for (int c = 0; c < HDB; ++c) acc += bufA[i * 65 + c] * bufB[j * 65 + c];
    S[i * ST + j] = S[i * ST + j] * (acc - Dv[i]) * 0.125f;
  }
  __syncthreads();
  // ---- 4. B <- K: dQ = dS K;  A <- Q: dK = dS^T Q -------------------------------------------------------------------------
  load_tile(bufB, qb + d, ld);
  load_tile(bufA, qb, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * HDB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float aq = 0.f, ak = 0.f;
    for (int t = 0; t < Tn; ++t) {
      aq += S[r * ST + t] * bufB[t * 65 + c];      // dQ[r][c] = sum_j dS[r][j] K[j][c]
      ak += S[t * ST + r] * bufA[t * 65 + c];      // dK[r][c] = sum_i dS[i][r] Q[i][c]
    }
    ab_st<T>(dqb, static_cast<size_t>(r) * ld + c, aq);
    ab_st<T>(dqb + d, static_cast<size_t>(r) * ld + c, ak);
  }
}

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_attention_backward(int32_t dtype, const void* qkv, const void* o, const void* dout, void* dqkv, int32_t B,
                                      int32_t T, int32_t d, int32_t causal, const uint8_t* key_padding_mask, void* stream) {
  CMH_CHECK_ARG(qkv && o && dout && dqkv && B > 0 && T > 0, "attention_backward: bad arguments");
  CMH_CHECK_ARG(dtype == CMH_F32 || dtype == CMH_BF16, "attention_backward: bad dtype");
  CMH_CHECK_ARG(d % HDB == 0, "attention_backward: width %d is not a multiple of 64", d);
  CMH_CHECK_ARG(T <= 128, "attention_backward: T=%d > 128 is not built (both CLIP towers have T <= 77)", T);
  const size_t lds = (static_cast<size_t>(2) * T * 65 + static_cast<size_t>(T) * (T + 1) + T) * 4;
  const dim3 grid(B * (d / HDB));
  hipStream_t st = as_stream(stream);
  if (dtype == CMH_F32) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess) return fail(CMH_ERR_LAUNCH, "attention_backward: LDS");
    hipLaunchKernelGGL(attention_bwd_kernel<float>, grid, dim3(256), lds, st, static_cast<const float*>(qkv),
                       static_cast<const float*>(o), static_cast<const float*>(dout), static_cast<float*>(dqkv), B, T, d, causal,
                       key_padding_mask);
  } else {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess) return fail(CMH_ERR_LAUNCH, "attention_backward: LDS");
    hipLaunchKernelGGL(attention_bwd_kernel<bf16_t>, grid, dim3(256), lds, st, static_cast<const bf16_t*>(qkv),
                       static_cast<const bf16_t*>(o), static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dqkv), B, T, d,
                       causal, key_padding_mask);
  }
  CMH_CHECK_LAUNCH("attention_backward");
  return CMH_OK;
}
