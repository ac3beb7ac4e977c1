// Row-wise kernels of the CLIP towers: LayerNorm (fp32 statistics, model/base/model.py:153-159),
// conv1 patch extraction (:215,:231-235), class/positional embedding + ln_pre (:237-239),
// token embedding gather + positional (:360-362), EOT row selection (:370), f32->bf16 casts.
// All HBM-bound; one wave per row, 16-byte accesses.
#include "cmh_common.h"

namespace cmh {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int kMaxVec = 4;  // up to 4 float4 per lane -> d <= 1024
constexpr int kOutF32 = 0, kOutBf16 = 1, kOutF16 = 2;

// Normalise one row held as float4 fragments v[0..nv) (lane owns elements lane*4 + 256*j).
__device__ __forceinline__ void ln_row(float4 (&v)[kMaxVec], int nv, int d, int lane, const float* __restrict__ w,
                                       const float* __restrict__ b, void* __restrict__ out_row, int out_kind) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j)
    if (j < nv && lane * 4 + 256 * j < d) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = wave_sum(s) / static_cast<float>(d);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j)
    if (j < nv && lane * 4 + 256 * j < d) {
      const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
      ss += (a * a + bb * bb) + (c * c + e * e);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / static_cast<float>(d) + 1e-5f);
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j) {
    const int e0 = lane * 4 + 256 * j;
    if (j < nv && e0 < d) {
      const float4 wv = *reinterpret_cast<const float4*>(w + e0);
      const float4 bv = *reinterpret_cast<const float4*>(b + e0);
      float4 y;
      y.x = (v[j].x - mean) * rstd * wv.x + bv.x;
      y.y = (v[j].y - mean) * rstd * wv.y + bv.y;
      y.z = (v[j].z - mean) * rstd * wv.z + bv.z;
      y.w = (v[j].w - mean) * rstd * wv.w + bv.w;
      if (out_kind == kOutBf16) {
        uint2 pk;
        pk.x = static_cast<uint32_t>(f32_to_bf16(y.x)) | (static_cast<uint32_t>(f32_to_bf16(y.y)) << 16);
        pk.y = static_cast<uint32_t>(f32_to_bf16(y.z)) | (static_cast<uint32_t>(f32_to_bf16(y.w)) << 16);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(out_row) + e0) = pk;
      } else if (out_kind == kOutF16) {
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out_row) + e0) = uint2{pack_f16x2(y.x, y.y), pack_f16x2(y.z, y.w)};
      } else {
        *reinterpret_cast<float4*>(static_cast<float*>(out_row) + e0) = y;
      }
    }
  }
}

// 4 consecutive elements of a residual-stream row: f32 (16-byte load) or fp16 (8-byte load), see encoders.hip
template <bool XH>
__device__ __forceinline__ float4 load_x4(const void* row, int e0) {
  if constexpr (XH) {
    const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(row) + e0);
    return float4{f16lo_to_f32(u.x), f16hi_to_f32(u.x), f16lo_to_f32(u.y), f16hi_to_f32(u.y)};
  } else {
    return *reinterpret_cast<const float4*>(static_cast<const float*>(row) + e0);
  }
}

template <bool XH>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, const int32_t* __restrict__ row_index,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        void* __restrict__ out, int out_kind, int M, int d,
                                                        const int32_t* __restrict__ m_dev) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }      // packed text: the real row count lives on the device
  if (row >= M) return;
  const size_t src = row_index ? static_cast<size_t>(row_index[row]) : static_cast<size_t>(row);
  const char* xr = static_cast<const char*>(x) + src * d * (XH ? 2 : 4);
  const int nv = (d + 255) / 256;
  float4 v[kMaxVec];
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j) {
    const int e0 = lane * 4 + 256 * j;
    v[j] = (j < nv && e0 < d) ? load_x4<XH>(xr, e0) : float4{0.f, 0.f, 0.f, 0.f};
  }
  char* orow = static_cast<char*>(out) + static_cast<size_t>(row) * d * (out_kind == kOutF32 ? 4 : 2);
  ln_row(v, nv, d, lane, w, b, orow, out_kind);     // 0 f32, 1 bf16, 2 fp16 (the values of the old out_bf16 flag still hold)
}

// The bf16 mode's LayerNorm: fp16 row in, bf16 row out, d a multiple of 256.  HALF a wave per row: a lane owns 8 consecutive
// elements of every 256-element block, so a row is d/256 16-byte loads per lane (the one-wave-per-row kernel above moves
// 8 bytes per lane per load and has too few bytes in flight: 39 MB in 12.4 us = 3.2 TB/s at width 768).
template <int NB>   // d / 256; `block`: this workgroup's index among the blocks of its row set
__device__ __forceinline__ void layernorm_h2b_rows(const uint16_t* __restrict__ x, const int32_t* __restrict__ row_index,
                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                   uint16_t* __restrict__ out, int M, const int32_t* __restrict__ m_dev, int block) {
  constexpr int d = NB * 256;
  if (m_dev) { const int md = *m_dev; M = md < M ? md : M; }      // packed text: the real row count lives on the device
  constexpr int RPH = NB <= 2 ? 2 : 1;     // rows per half-wave: narrow rows (512 elements = two loads per lane) have too few bytes in
                                           // flight one at a time (10 499 x 512: 7.7 us = 2.8 TB/s); both rows' loads are issued first
  const int hl = threadIdx.x & 31;
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  const int row0 = (block * 8 + (threadIdx.x >> 5)) * RPH;
  if (row0 >= M) return;
  float v[RPH][NB][8];
#pragma unroll
  for (int r = 0; r < RPH; ++r) {
    const int row = row0 + r < M ? row0 + r : M - 1;
    const size_t src = row_index ? static_cast<size_t>(row_index[row]) : static_cast<size_t>(row);
    const uint16_t* xr = x + src * d;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const u4 u = *reinterpret_cast<const u4*>(xr + j * 256 + hl * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[r][j][2 * k] = f16lo_to_f32(u[k]); v[r][j][2 * k + 1] = f16hi_to_f32(u[k]); }
    }
  }
  auto half_sum = [](float t) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
#pragma unroll
  for (int r = 0; r < RPH; ++r) {
    const int row = row0 + r;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[r][j][k];
    const float mean = half_sum(s) / static_cast<float>(d);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float c = v[r][j][k] - mean; ss += c * c; }
    const float rstd = 1.0f / sqrtf(half_sum(ss) / static_cast<float>(d) + 1e-5f);
    if (row >= M) continue;                 // (the butterflies above run for every lane of the wave)
    uint16_t* orow = out + static_cast<size_t>(row) * d;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int e0 = j * 256 + hl * 8;
      const float4 w0 = *reinterpret_cast<const float4*>(w + e0), w1 = *reinterpret_cast<const float4*>(w + e0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b + e0), b1 = *reinterpret_cast<const float4*>(b + e0 + 4);
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      u4 pk;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        pk[k] = pack_bf16x2((v[r][j][2 * k] - mean) * rstd * wv[2 * k] + bv[2 * k],
                            (v[r][j][2 * k + 1] - mean) * rstd * wv[2 * k + 1] + bv[2 * k + 1]);
      *reinterpret_cast<u4*>(orow + e0) = pk;
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256) void layernorm_h2b_kernel(const uint16_t* __restrict__ x, const int32_t* __restrict__ row_index,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            uint16_t* __restrict__ out, int M, const int32_t* __restrict__ m_dev) {
  layernorm_h2b_rows<NB>(x, row_index, w, b, out, M, m_dev, blockIdx.x);
}

// The same LayerNorm of BOTH towers' rows in one launch (round 4, the lock-step pair path of encoders.hip): blocks [0, blocks0) take
// the first row set (width 256 * NB0), the rest the second (256 * NB1).  Every row sees layernorm_h2b_rows' arithmetic: identical bits.
struct LnRows { const uint16_t* x; const float* w; const float* b; uint16_t* out; int M; const int32_t* m_dev; };
template <int NB0, int NB1>
__global__ __launch_bounds__(256) void layernorm_h2b_pair_kernel(LnRows r0, LnRows r1, int blocks0) {
  if (static_cast<int>(blockIdx.x) < blocks0) layernorm_h2b_rows<NB0>(r0.x, nullptr, r0.w, r0.b, r0.out, r0.M, r0.m_dev, blockIdx.x);
  else layernorm_h2b_rows<NB1>(r1.x, nullptr, r1.w, r1.b, r1.out, r1.M, r1.m_dev, blockIdx.x - blocks0);
}

// fp16 stream -> bf16 rows of two row sets; false when the pair is not one this file has a kernel for (the caller launches twice)
int launch_layernorm_h2b_pair(const void* x0, const float* w0, const float* b0, void* out0, int M0, int d0, const int32_t* md0,
                               const void* x1, const float* w1, const float* b1, void* out1, int M1, int d1, const int32_t* md1,
                               hipStream_t st) {
  static const bool off = []() { const char* e = getenv("CMH_PAIR_KERNELS"); return e && e[0] == '0'; }();
  if (off || M0 <= 0 || M1 <= 0) return 1;
  const LnRows r0{static_cast<const uint16_t*>(x0), w0, b0, static_cast<uint16_t*>(out0), M0, md0};
  const LnRows r1{static_cast<const uint16_t*>(x1), w1, b1, static_cast<uint16_t*>(out1), M1, md1};
  auto blocks = [](int M, int d) { return d <= 512 ? (M + 15) / 16 : (M + 7) / 8; };     // two rows per half-wave for d <= 512
  const int bl0 = blocks(M0, d0), bl1 = blocks(M1, d1);
  const dim3 grid(bl0 + bl1), block(256);
  if (d0 == 768 && d1 == 512) hipLaunchKernelGGL((layernorm_h2b_pair_kernel<3, 2>), grid, block, 0, st, r0, r1, bl0);
  else if (d0 == 768 && d1 == 768) hipLaunchKernelGGL((layernorm_h2b_pair_kernel<3, 3>), grid, block, 0, st, r0, r1, bl0);
  else if (d0 == 1024 && d1 == 768) hipLaunchKernelGGL((layernorm_h2b_pair_kernel<4, 3>), grid, block, 0, st, r0, r1, bl0);
  else if (d0 == 1024 && d1 == 512) hipLaunchKernelGGL((layernorm_h2b_pair_kernel<4, 2>), grid, block, 0, st, r0, r1, bl0);
  else if (d0 == 512 && d1 == 512) hipLaunchKernelGGL((layernorm_h2b_pair_kernel<2, 2>), grid, block, 0, st, r0, r1, bl0);
  else return 1;
  CMH_CHECK_LAUNCH("layernorm (pair)");
  return CMH_OK;
}

int launch_layernorm_x(const void* x, int x_f16, const int32_t* row_index, const float* w, const float* b, void* out,
                       int out_bf16, int M, int d, hipStream_t st, const int32_t* m_dev) {
  CMH_CHECK_ARG(d % 4 == 0 && d <= 256 * kMaxVec, "layernorm: d=%d must be a multiple of 4 and <= 1024", d);
  if (x_f16 && out_bf16 && d % 256 == 0) {
    const uint16_t* xh = static_cast<const uint16_t*>(x);
    uint16_t* oh = static_cast<uint16_t*>(out);
    const dim3 grid((M + 7) / 8), grid2((M + 15) / 16), block(256);     // two rows per half-wave for d <= 512
    switch (d / 256) {
      case 1: hipLaunchKernelGGL(layernorm_h2b_kernel<1>, grid2, block, 0, st, xh, row_index, w, b, oh, M, m_dev); break;
      case 2: hipLaunchKernelGGL(layernorm_h2b_kernel<2>, grid2, block, 0, st, xh, row_index, w, b, oh, M, m_dev); break;
      case 3: hipLaunchKernelGGL(layernorm_h2b_kernel<3>, grid, block, 0, st, xh, row_index, w, b, oh, M, m_dev); break;
      default: hipLaunchKernelGGL(layernorm_h2b_kernel<4>, grid, block, 0, st, xh, row_index, w, b, oh, M, m_dev); break;
    }
    CMH_CHECK_LAUNCH("layernorm");
    return CMH_OK;
  }
  if (x_f16)
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, x, row_index, w, b, out, out_bf16, M, d, m_dev);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, x, row_index, w, b, out, out_bf16, M, d, m_dev);
  CMH_CHECK_LAUNCH("layernorm");
  return CMH_OK;
}

int launch_layernorm(const float* x, const int32_t* row_index, const float* w, const float* b, void* out,
                     int out_bf16, int M, int d, hipStream_t st) {
  return launch_layernorm_x(x, 0, row_index, w, b, out, out_bf16, M, d, st, nullptr);
}

// any input kind (f32 / fp16) to any output kind (f32 / bf16 / fp16): the training forward's ln_pre writes the stream's type
int launch_layernorm_any(const void* x, int x_kind, const int32_t* row_index, const float* w, const float* b, void* out,
                         int out_kind, int M, int d, hipStream_t st) {
  CMH_CHECK_ARG(d % 4 == 0 && d <= 256 * kMaxVec, "layernorm: d=%d must be a multiple of 4 and <= 1024", d);
  CMH_CHECK_ARG((x_kind == 0 || x_kind == 2) && out_kind >= 0 && out_kind <= 2, "layernorm: bad kinds %d -> %d", x_kind, out_kind);
  if (x_kind == 2)
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, x, row_index, w, b, out, out_kind, M, d, nullptr);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, x, row_index, w, b, out, out_kind, M, d, nullptr);
  CMH_CHECK_LAUNCH("layernorm");
  return CMH_OK;
}

// x_pre[b,0] = cls + pos[0]; x_pre[b,1+i] = patch_out[b*g2+i] + pos[1+i]   (f32; the training forward keeps it for ln_pre's backward)
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int B, int g2,
                                                           int d) {
  const int T = g2 + 1;
  const size_t total4 = static_cast<size_t>(B) * T * d / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total4; i += static_cast<size_t>(gridDim.x) * 256) {
    const size_t e = i * 4;
    const int c = static_cast<int>(e % d);
    const size_t row = e / d;
    const int b = static_cast<int>(row / T), t = static_cast<int>(row - static_cast<size_t>(b) * T);
    const float4 a = *reinterpret_cast<const float4*>(t == 0 ? cls + c : patch_out + (static_cast<size_t>(b) * g2 + (t - 1)) * d + c);
    const float4 q = *reinterpret_cast<const float4*>(pos + static_cast<size_t>(t) * d + c);
    *reinterpret_cast<float4*>(x + e) = float4{a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w};
  }
}

int launch_vit_assemble(const float* patch_out, const float* cls, const float* pos, float* x, int B, int g2, int d, hipStream_t st) {
  CMH_CHECK_ARG(d % 4 == 0, "vit_assemble: width %d unsupported", d);
  const size_t total4 = static_cast<size_t>(B) * (g2 + 1) * d / 4;
  const size_t blocks = (total4 + 255) / 256;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(static_cast<unsigned>(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, patch_out, cls,
                     pos, x, B, g2, d);
  CMH_CHECK_LAUNCH("vit_assemble");
  return CMH_OK;
}

// ---- conv1 as GEMM: patch extraction -----------------------------------------------------------
// patches[(b*g + gy)*g + gx][c*p*p + py*p + px] = image[b][c][gy*p+py][gx*p+px]
// one thread moves 4 consecutive px (16-B read; 16-B f32 / 8-B bf16 write).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ image, void* __restrict__ patches,
                                                       int out_bf16, int B, int R, int p) {
  const int g = R / p;
  const size_t total4 = static_cast<size_t>(B) * 3 * R * R / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total4;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const size_t e = i * 4;                       // flat index into image
    const int x = static_cast<int>(e % R);
    const int y = static_cast<int>((e / R) % R);
    const int c = static_cast<int>((e / (static_cast<size_t>(R) * R)) % 3);
    const int b = static_cast<int>(e / (static_cast<size_t>(R) * R * 3));
    const int gx = x / p, px = x % p, gy = y / p, py = y % p;
    const float4 v = *reinterpret_cast<const float4*>(image + e);
    const size_t row = (static_cast<size_t>(b) * g + gy) * g + gx;
    const size_t col = (static_cast<size_t>(c) * p + py) * p + px;
    const size_t o = row * (3 * p * p) + col;
    if (out_bf16) {
      uint2 pk;
      pk.x = static_cast<uint32_t>(f32_to_bf16(v.x)) | (static_cast<uint32_t>(f32_to_bf16(v.y)) << 16);
      pk.y = static_cast<uint32_t>(f32_to_bf16(v.z)) | (static_cast<uint32_t>(f32_to_bf16(v.w)) << 16);
      *reinterpret_cast<uint2*>(static_cast<bf16_t*>(patches) + o) = pk;
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(patches) + o) = v;
    }
  }
}

int launch_patchify(const float* image, void* patches, int dt, int B, int R, int p, hipStream_t st) {
  CMH_CHECK_ARG(p % 4 == 0 && R % p == 0, "patchify: resolution %d / patch %d unsupported", R, p);
  const size_t total4 = static_cast<size_t>(B) * 3 * R * R / 4;
  const int blocks = static_cast<int>(total4 / 256 + 1 < 4096 ? total4 / 256 + 1 : 4096);
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, st, image, patches, dt == CMH_BF16, B, R, p);
  CMH_CHECK_LAUNCH("patchify");
  return CMH_OK;
}

// ---- [cls ; patches] + positional -> ln_pre -------------------------------------------------------
__global__ __launch_bounds__(256) void vit_assemble_lnpre_kernel(const float* __restrict__ patch_out,
                                                                 const float* __restrict__ cls,
                                                                 const float* __restrict__ pos,
                                                                 const float* __restrict__ lnw,
                                                                 const float* __restrict__ lnb, void* __restrict__ x,
                                                                 int x_f16, int B, int g2, int d) {
  const int lane = threadIdx.x & 63;
  const int T = g2 + 1;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int b = row / T, t = row - b * T;
  const float* src = t == 0 ? cls : patch_out + (static_cast<size_t>(b) * g2 + (t - 1)) * d;
  const float* pr = pos + static_cast<size_t>(t) * d;
  const int nv = (d + 255) / 256;
  float4 v[kMaxVec];
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j) {
    const int e0 = lane * 4 + 256 * j;
    if (j < nv && e0 < d) {
      const float4 a = *reinterpret_cast<const float4*>(src + e0);
      const float4 q = *reinterpret_cast<const float4*>(pr + e0);
      v[j] = float4{a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w};
    } else {
      v[j] = float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  ln_row(v, nv, d, lane, lnw, lnb, static_cast<char*>(x) + static_cast<size_t>(row) * d * (x_f16 ? 2 : 4),
         x_f16 ? kOutF16 : kOutF32);
}

int launch_vit_assemble_lnpre(const float* patch_out, const float* cls, const float* pos, const float* lnw,
                              const float* lnb, void* x, int x_f16, int B, int g2, int d, hipStream_t st) {
  CMH_CHECK_ARG(d % 4 == 0 && d <= 256 * kMaxVec, "vit_assemble: width %d unsupported", d);
  const int rows = B * (g2 + 1);
  hipLaunchKernelGGL(vit_assemble_lnpre_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, patch_out, cls, pos, lnw,
                     lnb, x, x_f16, B, g2, d);
  CMH_CHECK_LAUNCH("vit_assemble_lnpre");
  return CMH_OK;
}

// ---- token embedding + positional; EOT row = first argmax of the token ids ------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos, void* __restrict__ x,
                                                         int x_f16, int32_t* __restrict__ eot_row, int B, int L, int d,
                                                         int vocab, const int32_t* __restrict__ seq_off, int eot_is_pos) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * L) return;
  const int b = row / L, t = row - b * L;
  if (seq_off) {                           // packed: only the tokens up to the EOT exist, sequence b = rows [seq_off[b], seq_off[b+1])
    if (t >= seq_off[b + 1] - seq_off[b]) return;
    // (eot_is_pos: the plan kept the rows up to the last unpadded position and left the EOT's POSITION in eot_row[b]; one thread per
    // caption turns it into the packed row)
    if (t == 0 && lane == 0) eot_row[b] = eot_is_pos ? seq_off[b] + eot_row[b] : seq_off[b + 1] - 1;
  }
  const int orow = seq_off ? seq_off[b] + t : row;
  int64_t id = tokens[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // out-of-vocabulary ids would fault; clamp
  const float* er = tok_emb + static_cast<size_t>(id) * d;
  const float* pr = pos + static_cast<size_t>(t) * d;
  char* xr = static_cast<char*>(x) + static_cast<size_t>(orow) * d * (x_f16 ? 2 : 4);
  for (int e0 = lane * 4; e0 < d; e0 += 256) {
    const float4 a = *reinterpret_cast<const float4*>(er + e0);
    const float4 q = *reinterpret_cast<const float4*>(pr + e0);
    const float4 y = float4{a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w};
    if (x_f16)
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(xr) + e0) = uint2{pack_f16x2(y.x, y.y), pack_f16x2(y.z, y.w)};
    else
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(xr) + e0) = y;
  }
  if (t == 0 && !seq_off) {   // this wave also finds argmax over the caption (first maximum, like torch.argmax)
    int64_t best = INT64_MIN;
    int besti = 0;
    for (int i = lane; i < L; i += 64) {
      const int64_t v = tokens[static_cast<size_t>(b) * L + i];
      if (v > best) { best = v; besti = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int64_t ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) eot_row[b] = b * L + besti;
  }
}

// seq_off[b] = sum_{b' < b} (argmax(tokens[b']) + 1): the packed row offsets of encode_text without padding.  One workgroup of
// 16 waves: a wave takes whole captions (coalesced row reads, first-maximum argmax by a butterfly on (value, index)), then wave 0
// turns the lengths into offsets with 64-wide prefix scans.  (Round 1's one-thread-per-caption loop took 37 us at B = 256.)
// kpm / eot_pos (the all-token trunk of MITH, model/MITH.py:120-144): the kept rows of caption b run to its LAST UNPADDED position
// (or its EOT, whichever is later) - everything behind is masked as a key, never read as a query's output (LocalizedTokenAggregation
// gives padded positions weight 0, model/MITH.py:349-376) - and the EOT's position goes to eot_pos[b].
__global__ __launch_bounds__(256) void text_pack_plan_kernel(const int64_t* __restrict__ tokens, int B, int L,
                                                             int32_t* __restrict__ seq_off, const uint8_t* __restrict__ kpm,
                                                             int32_t* __restrict__ eot_pos) {
  // One THREAD per caption: a running (max, first index) over its tokens - no cross-lane traffic at all.
  // Then an inclusive scan of the lengths over the workgroup, 256 captions per pass.
  __shared__ int wsum[4];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 256) {
    const int b = base + tid;
    int len = 0;
    if (b < B) {
      const int64_t* row = tokens + static_cast<size_t>(b) * L;
      long long best = INT64_MIN;
      int besti = 0;
      for (int i0 = 0; i0 < L; i0 += 16) {            // 16 loads in flight: one at a time the 77 dependent round trips took 35 us
        long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = i0 + u < L ? row[i0 + u] : INT64_MIN;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (v[u] > best) { best = v[u]; besti = i0 + u; }      // strict >: the first maximum, like torch.argmax
      }
      len = besti + 1;
      if (kpm) {
        const uint8_t* mrow = kpm + static_cast<size_t>(b) * L;
        int last = -1;
        for (int i0 = 0; i0 < L; i0 += 16) {
          uint8_t m[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) m[u] = i0 + u < L ? mrow[i0 + u] : 1;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (!m[u]) last = i0 + u;
        }
        len = last + 1 > len ? last + 1 : len;
      }
      if (eot_pos) eot_pos[b] = besti;
    }
    int v = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (b < B) seq_off[b + 1] = before + v;
    __syncthreads();
    if (tid == 255) carry = before + v;
    __syncthreads();
  }
  if (tid == 0) seq_off[0] = 0;
}

int launch_text_pack_plan(const int64_t* tokens, int B, int L, int32_t* seq_off, hipStream_t st, const uint8_t* kpm, int32_t* eot_pos) {
  hipLaunchKernelGGL(text_pack_plan_kernel, dim3(1), dim3(256), 0, st, tokens, B, L, seq_off, kpm, eot_pos);
  CMH_CHECK_LAUNCH("text_pack_plan");
  return CMH_OK;
}

int launch_text_embed(const int64_t* tokens, const float* tok_emb, const float* pos, void* x, int x_f16, int32_t* eot_row,
                      int B, int L, int d, int vocab, hipStream_t st) {
  return launch_text_embed_packed(tokens, tok_emb, pos, x, x_f16, eot_row, B, L, d, vocab, nullptr, st);
}

int launch_text_embed_packed(const int64_t* tokens, const float* tok_emb, const float* pos, void* x, int x_f16, int32_t* eot_row,
                             int B, int L, int d, int vocab, const int32_t* seq_off, hipStream_t st, bool eot_is_pos) {
  CMH_CHECK_ARG(d % 4 == 0, "text_embed: width %d unsupported", d);
  const int rows = B * L;
  hipLaunchKernelGGL(text_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, tokens, tok_emb, pos, x, x_f16,
                     eot_row, B, L, d, vocab, seq_off, eot_is_pos ? 1 : 0);
  CMH_CHECK_LAUNCH("text_embed");
  return CMH_OK;
}

// dense[b, t, :] = t < len_b ? packed[seq_off[b] + t, :] : 0  (f32 rows of E floats, E % 4 == 0); eot_dense[b] = b L + (eot_packed[b] - seq_off[b]).
// One pass writes every byte of `dense`: no memset in front.
__global__ __launch_bounds__(256) void unpack_token_rows_kernel(const float* __restrict__ packed, const int32_t* __restrict__ seq_off,
                                                                float* __restrict__ dense, int B, int L, int E,
                                                                const int32_t* __restrict__ eot_packed, int32_t* __restrict__ eot_dense) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const int b = row / L, t = row - b * L;
  const int s0 = seq_off[b], len = seq_off[b + 1] - s0;
  if (t == 0 && lane == 0 && eot_dense) eot_dense[b] = b * L + (eot_packed[b] - s0);
  float4* o = reinterpret_cast<float4*>(dense + static_cast<size_t>(row) * E);
  if (t < len) {
    const float4* p = reinterpret_cast<const float4*>(packed + static_cast<size_t>(s0 + t) * E);
    for (int i = lane; i < E / 4; i += 64) o[i] = p[i];
  } else {
    for (int i = lane; i < E / 4; i += 64) o[i] = float4{0.f, 0.f, 0.f, 0.f};
  }
}

int launch_unpack_token_rows(const float* packed, const int32_t* seq_off, float* dense, int B, int L, int E, const int32_t* eot_packed,
                             int32_t* eot_dense, hipStream_t st) {
  CMH_CHECK_ARG(E % 4 == 0, "unpack_token_rows: E %d", E);
  hipLaunchKernelGGL(unpack_token_rows_kernel, dim3((B * L + 3) / 4), dim3(256), 0, st, packed, seq_off, dense, B, L, E, eot_packed, eot_dense);
  CMH_CHECK_LAUNCH("unpack_token_rows");
  return CMH_OK;
}

// packed[seq_off[b] + t, :] = dense[b, t, :] for the kept rows (the gradient of the all-token head on its way into the packed backward)
__global__ __launch_bounds__(256) void pack_token_rows_kernel(const float* __restrict__ dense, const int32_t* __restrict__ seq_off,
                                                              float* __restrict__ packed, int B, int L, int E) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const int b = row / L, t = row - b * L;
  const int s0 = seq_off[b], len = seq_off[b + 1] - s0;
  if (t >= len) return;
  const float4* p = reinterpret_cast<const float4*>(dense + static_cast<size_t>(row) * E);
  float4* o = reinterpret_cast<float4*>(packed + static_cast<size_t>(s0 + t) * E);
  for (int i = lane; i < E / 4; i += 64) o[i] = p[i];
}

int launch_pack_token_rows(const float* dense, const int32_t* seq_off, float* packed, int B, int L, int E, hipStream_t st) {
  CMH_CHECK_ARG(E % 4 == 0, "pack_token_rows: E %d", E);
  hipLaunchKernelGGL(pack_token_rows_kernel, dim3((B * L + 3) / 4), dim3(256), 0, st, dense, seq_off, packed, B, L, E);
  CMH_CHECK_LAUNCH("pack_token_rows");
  return CMH_OK;
}

__global__ void iota_rows_kernel(int32_t* rows, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) rows[i] = i * T;
}

int launch_iota_rows(int32_t* rows, int B, int T, hipStream_t st) {
  hipLaunchKernelGGL(iota_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, st, rows, B, T);
  CMH_CHECK_LAUNCH("iota_rows");
  return CMH_OK;
}

// dstA[b] = srcA[rows[b]], dstB[b] = srcB[rows[b]] for rows of bytesA / bytesB (multiples of 16) bytes, one launch: the pooled rows
// of the residual stream and of the attention output before the last block's row-wise tail (encoders.hip)
__global__ __launch_bounds__(256) void gather_rows2_kernel(const uint4* __restrict__ srcA, uint4* __restrict__ dstA, int chunksA,
                                                           const uint4* __restrict__ srcB, uint4* __restrict__ dstB, int chunksB,
                                                           const int32_t* __restrict__ rows, int B) {
  const int per = chunksA + chunksB;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * per) return;
  const int b = i / per, ch = i - b * per;
  const size_t r = static_cast<size_t>(rows[b]);
  if (ch < chunksA) dstA[static_cast<size_t>(b) * chunksA + ch] = srcA[r * chunksA + ch];
  else dstB[static_cast<size_t>(b) * chunksB + ch - chunksA] = srcB[r * chunksB + ch - chunksA];
}

int launch_gather_rows2(const void* srcA, void* dstA, int bytesA, const void* srcB, void* dstB, int bytesB, const int32_t* rows, int B,
                        hipStream_t st) {
  CMH_CHECK_ARG(bytesA > 0 && bytesA % 16 == 0 && bytesB > 0 && bytesB % 16 == 0, "gather_rows: rows of %d / %d bytes", bytesA, bytesB);
  const int per = bytesA / 16 + bytesB / 16;
  hipLaunchKernelGGL(gather_rows2_kernel, dim3((B * per + 255) / 256), dim3(256), 0, st, static_cast<const uint4*>(srcA),
                     static_cast<uint4*>(dstA), bytesA / 16, static_cast<const uint4*>(srcB), static_cast<uint4*>(dstB), bytesB / 16, rows, B);
  CMH_CHECK_LAUNCH("gather_rows");
  return CMH_OK;
}

// dst[rows[b]] = src[b] for rows of row_bytes (a multiple of 16) bytes (dst zeroed by the caller): the pooled rows' gradients back
// into the full-size streams before the last block's attention backward (encoders_bwd.hip)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ rows,
                                                           uint4* __restrict__ dst, int B, int chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * chunks) return;
  const int b = i / chunks, ch = i - b * chunks;
  dst[static_cast<size_t>(rows[b]) * chunks + ch] = src[i];
}

int launch_scatter_rows(const void* src, const int32_t* rows, void* dst, int B, int row_bytes, hipStream_t st) {
  CMH_CHECK_ARG(row_bytes > 0 && row_bytes % 16 == 0, "scatter_rows: row of %d bytes", row_bytes);
  const int chunks = row_bytes / 16;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((B * chunks + 255) / 256), dim3(256), 0, st, static_cast<const uint4*>(src), rows,
                     static_cast<uint4*>(dst), B, chunks);
  CMH_CHECK_LAUNCH("scatter_rows");
  return CMH_OK;
}

// ---- f32 -> bf16 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                        int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
    uint2 pk;
    pk.x = static_cast<uint32_t>(f32_to_bf16(v.x)) | (static_cast<uint32_t>(f32_to_bf16(v.y)) << 16);
    pk.y = static_cast<uint32_t>(f32_to_bf16(v.z)) | (static_cast<uint32_t>(f32_to_bf16(v.w)) << 16);
    *reinterpret_cast<uint2*>(dst + i * 4) = pk;
  }
  for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f32_to_bf16(src[i]);
}

}  // namespace cmh

extern "C" int cmh_cast_f32_to_bf16(const float* src, void* dst_bf16, int64_t n, void* stream) {
  using namespace cmh;
  CMH_CHECK_ARG(n >= 0, "cast: negative n");
  if (n == 0) return CMH_OK;
  CMH_CHECK_ARG((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst_bf16) & 7) == 0,
                "cast: pointers must be 16-byte (src) / 8-byte (dst) aligned");
  const int64_t blocks = (n / 4 + 255) / 256 + 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     as_stream(stream), src, static_cast<bf16_t*>(dst_bf16), n);
  CMH_CHECK_LAUNCH("cast_f32_to_bf16");
  return CMH_OK;
}
