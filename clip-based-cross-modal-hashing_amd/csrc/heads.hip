// Hash heads and code generation (all tiny next to the towers; one launch each):
//   LinearHash      model/modelbase.py:25-35   tanh(dropout(fc(x)))
//   HashLayer       model/DCHMT.py:8-26        relu(fc) -> K x softmax(Linear(128,2))
//   Pre_Layer       model/DNPH_TOMM.py:7-14    fc
//   sign / argmax   train/base.py:141-158      codes in {-1,0,+1}
// small_linear also serves the towers' final projection when embed_dim is not a multiple of the GEMM
// tile (tiny test configs).
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  return float4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u), __uint_as_float(a.y << 16),
                __uint_as_float(a.y & 0xffff0000u)};
}

constexpr int kSLMaxVec = 16;   // K <= 16*256 = 4096

// y[m,n] = act((sum_k x[m,k] w[n,k] + b[n]) * (mask ? mask[m,n]*keep_scale : 1)); block = one row m,
// 4 waves take groups of 4 consecutive n, lanes split k (float4 each, stride 256).
template <typename T>
__global__ __launch_bounds__(256) void small_linear_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ mask, float keep_scale, int act,
                                                           float* __restrict__ y, int M, int N, int K) {
  const int m = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const T* xr = x + static_cast<size_t>(m) * K;
  float4 xv[kSLMaxVec];
#pragma unroll
  for (int j = 0; j < kSLMaxVec; ++j) {
    const int k0 = lane * 4 + 256 * j;
    xv[j] = k0 < K ? ld4(xr + k0) : float4{0.f, 0.f, 0.f, 0.f};
  }
  // four outputs per iteration: their weight loads are independent, so their latencies overlap (one output at a time spent
  // ~1 us per output waiting on L2: 24 us for a 256 x 512 -> 64 head).  Each output's sum is formed exactly as before
  // (same lane partition of k, same butterfly), so results are unchanged bit for bit.
  for (int n0 = wid * 4; n0 < N; n0 += 16) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kSLMaxVec; ++j) {
      const int k0 = lane * 4 + 256 * j;
      if (k0 < K) {
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + u < N ? n0 + u : N - 1;
          wv[u] = ld4(w + static_cast<size_t>(n) * K + k0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s[u] = fmaf(xv[j].x, wv[u].x, s[u]);
          s[u] = fmaf(xv[j].y, wv[u].y, s[u]);
          s[u] = fmaf(xv[j].z, wv[u].z, s[u]);
          s[u] = fmaf(xv[j].w, wv[u].w, s[u]);
        }
      }
    }
    // the butterfly leaves every sum in every lane: lane u keeps output u, then ONE pass of bias / mask / activation / store for the
    // four (a lane-0-only epilogue per output made the wave walk through tanhf once per output)
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = wsum(s[u]);
      if (lane == u) t = v;
    }
    const int n = n0 + lane;
    if (lane < 4 && n < N) {
      if (bias) t += bias[n];
      if (mask) t *= mask[static_cast<size_t>(m) * N + n] * keep_scale;
      if (act == CMH_ACT_TANH) t = tanhf(t);
      else if (act == CMH_ACT_RELU) t = fmaxf(t, 0.f);
      y[static_cast<size_t>(m) * N + n] = t;
    }
  }
}

// The same arithmetic for MANY rows (MITH's token-level concept similarities: 12 544 x 512 -> 64): one block per row re-reads the
// whole weight matrix from L2 for every row (1.6 GB for that shape, 267 us).  Here a block owns R consecutive rows, keeps their
// K <= 1024 inputs in registers and applies every weight vector it loads to all of them.  Each output is still formed by the same
// lane partition of k, the same FMA chain per lane and the same butterfly: bit-identical to small_linear_kernel.
template <typename T, int R, int KV>   // KV float4 per lane and row: K <= 256 KV
__global__ __launch_bounds__(256, 2) void small_linear_rows_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                                const float* __restrict__ bias, const float* __restrict__ mask,
                                                                float keep_scale, int act, float* __restrict__ y, int M, int N, int K) {
  const int m0 = blockIdx.x * R;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4 xv[R][KV];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m = m0 + r < M ? m0 + r : M - 1;
    const T* xr = x + static_cast<size_t>(m) * K;
#pragma unroll
    for (int j = 0; j < KV; ++j) {
      const int k0 = lane * 4 + 256 * j;
      xv[r][j] = k0 < K ? ld4(xr + k0) : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int n0 = wid * 4; n0 < N; n0 += 16) {
    float s[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[r][u] = 0.f;
#pragma unroll
    for (int j = 0; j < KV; ++j) {
      const int k0 = lane * 4 + 256 * j;
      if (k0 < K) {
        float4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + u < N ? n0 + u : N - 1;
          wv[u] = ld4(w + static_cast<size_t>(n) * K + k0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s[r][u] = fmaf(xv[r][j].x, wv[u].x, s[r][u]);
            s[r][u] = fmaf(xv[r][j].y, wv[u].y, s[r][u]);
            s[r][u] = fmaf(xv[r][j].z, wv[u].z, s[r][u]);
            s[r][u] = fmaf(xv[r][j].w, wv[u].w, s[r][u]);
          }
      }
    }
    float t = 0.f;          // lane 4 r + u keeps output (row r, column u): one epilogue pass for the 4 R outputs
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = wsum(s[r][u]);
        if (lane == r * 4 + u) t = v;
      }
    const int n = n0 + (lane & 3), m = m0 + (lane >> 2);
    if (lane < 4 * R && n < N && m < M) {
      if (bias) t += bias[n];
      if (mask) t *= mask[static_cast<size_t>(m) * N + n] * keep_scale;
      if (act == CMH_ACT_TANH) t = tanhf(t);
      else if (act == CMH_ACT_RELU) t = fmaxf(t, 0.f);
      y[static_cast<size_t>(m) * N + n] = t;
    }
  }
}

int launch_small_linear(int dt, const void* x, const void* w, const float* bias, const float* mask,
                        float keep_scale, int act, float* y, int M, int N, int K, hipStream_t st) {
  CMH_CHECK_ARG(M > 0 && N > 0 && K > 0, "linear: empty problem M=%d N=%d K=%d", M, N, K);
  CMH_CHECK_ARG(K % 4 == 0 && K <= 256 * kSLMaxVec, "linear: K=%d must be a multiple of 4 and <= 4096", K);
  static const bool rows_off = []() { const char* e = getenv("CMH_SMALL_LINEAR_ROWS"); return e && e[0] == '0'; }();
  if (dt == CMH_F32 && K <= 1024 && M >= 2048 && !rows_off) {   // >= 256 blocks of 8 rows: the chip stays full
    constexpr int R = 8;
    if (K <= 512)
      hipLaunchKernelGGL((small_linear_rows_kernel<float, R, 2>), dim3((M + R - 1) / R), dim3(256), 0, st, static_cast<const float*>(x),
                         static_cast<const float*>(w), bias, mask, keep_scale, act, y, M, N, K);
    else   // (four rows: eight rows of 1024 inputs do not fit two waves per SIMD)
      hipLaunchKernelGGL((small_linear_rows_kernel<float, 4, 4>), dim3((M + 3) / 4), dim3(256), 0, st, static_cast<const float*>(x),
                         static_cast<const float*>(w), bias, mask, keep_scale, act, y, M, N, K);
    CMH_CHECK_LAUNCH("small_linear_rows");
    return CMH_OK;
  }
  if (dt == CMH_F32)
    hipLaunchKernelGGL(small_linear_kernel<float>, dim3(M), dim3(256), 0, st, static_cast<const float*>(x),
                       static_cast<const float*>(w), bias, mask, keep_scale, act, y, M, N, K);
  else
    hipLaunchKernelGGL(small_linear_kernel<bf16_t>, dim3(M), dim3(256), 0, st, static_cast<const bf16_t*>(x),
                       static_cast<const bf16_t*>(w), bias, mask, keep_scale, act, y, M, N, K);
  CMH_CHECK_LAUNCH("small_linear");
  return CMH_OK;
}

__global__ __launch_bounds__(256) void pair_softmax_kernel(const float* __restrict__ z, float* __restrict__ p,
                                                           int64_t pairs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const float2 v = *reinterpret_cast<const float2*>(z + 2 * i);
  const float mx = fmaxf(v.x, v.y);
  const float e0 = expf(v.x - mx), e1 = expf(v.y - mx);
  const float inv = 1.0f / (e0 + e1);
  *reinterpret_cast<float2*>(p + 2 * i) = float2{e0 * inv, e1 * inv};
}

__global__ __launch_bounds__(256) void sign_kernel(const float* __restrict__ h, float* __restrict__ c, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = h[i];
  c[i] = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);   // sign(0)=0, NaN stays NaN (torch.sign)
}

__global__ __launch_bounds__(256) void pair_argmax_kernel(const float* __restrict__ p, float* __restrict__ c,
                                                          int64_t pairs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const float2 v = *reinterpret_cast<const float2*>(p + 2 * i);
  c[i] = v.y > v.x ? 1.f : -1.f;                  // argmax==1 only if strictly larger; ties -> index 0 -> -1
}

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_linear_act(const float* x, const float* w, const float* b, const float* drop_mask,
                              float keep_scale, int32_t act, float* y, int32_t M, int32_t N, int32_t K,
                              void* stream) {
  CMH_CHECK_ARG(x && w && y, "linear_act: null pointer");
  CMH_CHECK_ARG(act >= CMH_ACT_NONE && act <= CMH_ACT_RELU, "linear_act: bad activation %d", act);
  return launch_small_linear(CMH_F32, x, w, b, drop_mask, keep_scale, act, y, M, N, K, as_stream(stream));
}

extern "C" int cmh_pair_softmax(const float* z, float* p, int32_t M, int32_t K, void* stream) {
  CMH_CHECK_ARG(z && p && M > 0 && K > 0, "pair_softmax: bad arguments");
  const int64_t pairs = static_cast<int64_t>(M) * K;
  hipLaunchKernelGGL(pair_softmax_kernel, dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0,
                     as_stream(stream), z, p, pairs);
  CMH_CHECK_LAUNCH("pair_softmax");
  return CMH_OK;
}

extern "C" int cmh_sign_codes(const float* h, float* codes, int64_t n, void* stream) {
  CMH_CHECK_ARG(h && codes && n > 0, "sign_codes: bad arguments");
  hipLaunchKernelGGL(sign_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), h,
                     codes, n);
  CMH_CHECK_LAUNCH("sign_codes");
  return CMH_OK;
}

extern "C" int cmh_pair_argmax_codes(const float* p, float* codes, int32_t M, int32_t K, void* stream) {
  CMH_CHECK_ARG(p && codes && M > 0 && K > 0, "pair_argmax_codes: bad arguments");
  const int64_t pairs = static_cast<int64_t>(M) * K;
  hipLaunchKernelGGL(pair_argmax_kernel, dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0,
                     as_stream(stream), p, codes, pairs);
  CMH_CHECK_LAUNCH("pair_argmax_codes");
  return CMH_OK;
}
