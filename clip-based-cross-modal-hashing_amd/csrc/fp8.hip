// fp8 encoder mode (BASELINE configs[4] "fp8 MFMA CLIP encoders"): quantisation side of the e4m3 GEMMs of gemm_wide.hip.
//
// What goes to fp8 (OCP e4m3fn, the gfx950 format): the operands of the four GEMMs of every ResidualAttentionBlock - QKV,
// out_proj, c_fc, c_proj (reference model/base/model.py:167-196) - i.e. the tensors the reference's own precision hook
// `convert_weights` (model/base/model.py:391-412) lowers: Linear / MultiheadAttention weights, plus the activations that feed
// them.  LayerNorm statistics, softmax, the residual stream (fp16), biases, accumulation (f32) stay as in the bf16 mode;
// conv1 (patch embedding) and the two final projections stay bf16.
//   weights      per-OUTPUT-CHANNEL scale  colscale[n] = amax_k |W[n,k]| / 448          (cmh_fp8_quantize_weight, once per weight)
//   activations  per-TENSOR scale per GEMM input site and layer: a = headroom * amax / 448 from a calibration pass over a batch in
//                bf16 mode (cmh_amax after the producer of each site), quantised by the producer itself: LayerNorm (here),
//                attention (attention.hip), the c_fc epilogue (gemm_wide.hip, EPI_OUT_FP8).
// The GEMM epilogue multiplies the f32 accumulator by a * colscale[n] (one FMA together with the bias).
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };   // e4m3fn has no infinity: saturate (NaN stays NaN)
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), w, true);
  return static_cast<uint32_t>(w);
}

// one wave per row of W [N,K] f32: amax -> scale -> e4m3 row
__global__ __launch_bounds__(256) void fp8_quantize_rows_kernel(const float* __restrict__ w, uint8_t* __restrict__ q,
                                                                float* __restrict__ colscale, int N, int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  const float* src = w + static_cast<size_t>(row) * K;
  float m = 0.f;
  bool nan = false;      // fmaxf drops NaN operands: a NaN weight makes the row's scale NaN (and with it every output of that channel)
  for (int k = lane * 4; k < K; k += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + k);
    nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  nan = __any(nan);
  const float scale = nan ? __uint_as_float(0x7fc00000u) : (m > 0.f ? m / 448.f : 1.f);
  const float inv = 1.f / scale;
  if (lane == 0) colscale[row] = scale;
  uint32_t* dst = reinterpret_cast<uint32_t*>(q + static_cast<size_t>(row) * K);
  for (int k = lane * 4; k < K; k += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + k);
    dst[k >> 2] = pack_fp8x4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  }
}

// x (f32 / bf16 / f16) -> e4m3(x * inv_scale), 4 elements per thread
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const void* __restrict__ x, int kind, uint32_t* __restrict__ q, size_t n4,
                                                           float inv_scale) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    const float4 v = load4_as_f32(x, i * 4, kind);
    q[i] = pack_fp8x4(v.x * inv_scale, v.y * inv_scale, v.z * inv_scale, v.w * inv_scale);
  }
}

// e4m3 -> f32 (tests and the host-side dequantised reference)
__global__ __launch_bounds__(256) void fp8_dequantize_kernel(const uint32_t* __restrict__ q, float* __restrict__ out, size_t n4, float scale) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    const int w = static_cast<int>(q[i]);
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    *reinterpret_cast<float4*>(out + i * 4) = float4{lo[0] * scale, lo[1] * scale, hi[0] * scale, hi[1] * scale};
  }
}

// out[0] = max(out[0], max |x|): non-negative floats order like their bit patterns, so one atomicMax per workgroup
__global__ __launch_bounds__(256) void amax_kernel(const void* __restrict__ x, int kind, size_t n4, float* __restrict__ out) {
  float m = 0.f;
  bool nan = false;      // fmaxf drops NaN operands, so a NaN is tracked on its own and reported as +inf (scales then fail loudly)
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    const float4 v = load4_as_f32(x, i * 4, kind);
    nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (nan) m = __uint_as_float(0x7f800000u);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}

int launch_amax(const void* x, int kind, size_t n, float* out, hipStream_t st) {
  CMH_CHECK_ARG(n % 4 == 0, "amax: %zu elements (multiple of 4 expected)", n);
  const size_t n4 = n / 4, blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(amax_kernel, dim3(static_cast<unsigned>(blocks < 2048 ? (blocks ? blocks : 1) : 2048)), dim3(256), 0, st, x, kind, n4, out);
  CMH_CHECK_LAUNCH("amax");
  return CMH_OK;
}

// The fp8 mode's LayerNorm: fp16 residual-stream row in, e4m3 row out (x * inv_scale), d a multiple of 256; the layout of
// layernorm_h2b_kernel (norm_embed.hip): half a wave per row, a lane owns 8 consecutive elements of every 256-element block.
template <int NB>
__global__ __launch_bounds__(256) void layernorm_h2q_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, uint8_t* __restrict__ out, int M,
                                                            float inv_scale, const int32_t* __restrict__ m_dev) {
  constexpr int d = NB * 256;
  if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }
  const int hl = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  const uint16_t* xr = x + static_cast<size_t>(row) * d;
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  float v[NB][8];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const u4 u = *reinterpret_cast<const u4*>(xr + j * 256 + hl * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[j][2 * k] = f16lo_to_f32(u[k]); v[j][2 * k + 1] = f16hi_to_f32(u[k]); }
  }
  auto half_sum = [](float t) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[j][k];
  const float mean = half_sum(s) / static_cast<float>(d);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float c = v[j][k] - mean; ss += c * c; }
  const float rstd = 1.0f / sqrtf(half_sum(ss) / static_cast<float>(d) + 1e-5f);
  uint8_t* orow = out + static_cast<size_t>(row) * d;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int e0 = j * 256 + hl * 8;
    const float4 w0 = *reinterpret_cast<const float4*>(w + e0), w1 = *reinterpret_cast<const float4*>(w + e0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + e0), b1 = *reinterpret_cast<const float4*>(b + e0 + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float y[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = ((v[j][k] - mean) * rstd * wv[k] + bv[k]) * inv_scale;
    *reinterpret_cast<uint2*>(orow + e0) = uint2{pack_fp8x4(y[0], y[1], y[2], y[3]), pack_fp8x4(y[4], y[5], y[6], y[7])};
  }
}

int launch_layernorm_q(const void* x_f16, const float* w, const float* b, void* out_fp8, float inv_scale, int M, int d, hipStream_t st,
                       const int32_t* m_dev) {
  CMH_CHECK_ARG(d % 256 == 0 && d <= 1024, "layernorm (fp8): d=%d must be a multiple of 256, <= 1024", d);
  const uint16_t* xh = static_cast<const uint16_t*>(x_f16);
  uint8_t* o = static_cast<uint8_t*>(out_fp8);
  const dim3 grid((M + 7) / 8), block(256);
  switch (d / 256) {
    case 1: hipLaunchKernelGGL(layernorm_h2q_kernel<1>, grid, block, 0, st, xh, w, b, o, M, inv_scale, m_dev); break;
    case 2: hipLaunchKernelGGL(layernorm_h2q_kernel<2>, grid, block, 0, st, xh, w, b, o, M, inv_scale, m_dev); break;
    case 3: hipLaunchKernelGGL(layernorm_h2q_kernel<3>, grid, block, 0, st, xh, w, b, o, M, inv_scale, m_dev); break;
    default: hipLaunchKernelGGL(layernorm_h2q_kernel<4>, grid, block, 0, st, xh, w, b, o, M, inv_scale, m_dev); break;
  }
  CMH_CHECK_LAUNCH("layernorm_fp8");
  return CMH_OK;
}

}  // namespace cmh

using namespace cmh;

extern "C" int cmh_fp8_quantize_weight(const float* w, void* w_fp8, float* colscale, int32_t N, int32_t K, void* stream) {
  CMH_CHECK_ARG(w && w_fp8 && colscale && N > 0 && K > 0 && K % 4 == 0, "fp8_quantize_weight: bad arguments (N=%d K=%d)", N, K);
  hipLaunchKernelGGL(fp8_quantize_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), w, static_cast<uint8_t*>(w_fp8),
                     colscale, N, K);
  CMH_CHECK_LAUNCH("fp8_quantize_weight");
  return CMH_OK;
}

extern "C" int cmh_fp8_quantize(const void* x, int32_t kind, void* x_fp8, int64_t n, float scale, void* stream) {
  CMH_CHECK_ARG(x && x_fp8 && n > 0 && n % 4 == 0 && kind >= 0 && kind <= 2 && scale > 0.f, "fp8_quantize: bad arguments");
  const size_t n4 = static_cast<size_t>(n) / 4, blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(fp8_quantize_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream), x,
                     kind, static_cast<uint32_t*>(x_fp8), n4, 1.0f / scale);
  CMH_CHECK_LAUNCH("fp8_quantize");
  return CMH_OK;
}

extern "C" int cmh_fp8_dequantize(const void* x_fp8, float* out, int64_t n, float scale, void* stream) {
  CMH_CHECK_ARG(x_fp8 && out && n > 0 && n % 4 == 0, "fp8_dequantize: bad arguments");
  const size_t n4 = static_cast<size_t>(n) / 4, blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(fp8_dequantize_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0, as_stream(stream),
                     static_cast<const uint32_t*>(x_fp8), out, n4, scale);
  CMH_CHECK_LAUNCH("fp8_dequantize");
  return CMH_OK;
}

extern "C" int cmh_amax(const void* x, int32_t kind, int64_t n, float* amax_inout, void* stream) {
  CMH_CHECK_ARG(x && amax_inout && n > 0 && kind >= 0 && kind <= 2, "amax: bad arguments");
  return launch_amax(x, kind, static_cast<size_t>(n), amax_inout, as_stream(stream));
}

extern "C" int cmh_linear_gemm_fp8(const void* x_fp8, const void* w_fp8, const float* colscale, float alpha, const float* bias,
                                   const float* residual, void* out, float out_scale, int32_t M, int32_t N, int32_t K,
                                   int32_t epilogue, void* stream) {
  CMH_CHECK_ARG(out_scale > 0.f, "linear_gemm_fp8: out_scale must be positive");
  return launch_gemm_fp8(x_fp8, w_fp8, colscale, alpha, bias, residual, out, 1.0f / out_scale, M, N, K, epilogue, as_stream(stream));
}
