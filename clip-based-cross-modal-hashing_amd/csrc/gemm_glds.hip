// GEMM v2: same contract and tile shape as gemm.hip (out[M,N] = epi(X[M,K].W[N,K]^T), 128x128 tile per
// 256-thread workgroup, D = W_tile.X_tile^T so a lane owns 4 consecutive n), but the operand tiles travel
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) instead of through VGPRs:
//   * no staging registers and no ds_write pass: per K-step a wave issues 8 x 1-KiB DMA pieces;
//   * the LDS image must be lane-linear per piece (dest = wave base + lane*16), so the XOR swizzle that
//     keeps the fragment ds_read_b128 conflict-free is applied to the per-lane SOURCE address: lane i of a
//     piece fills (row = 8*piece + i/8, physical chunk = i%8) and therefore fetches logical chunk
//     (i%8) ^ (row&7) of that row (cdna guide §5.4 rule 21: linear dest + swizzled source + swizzled read);
//   * one barrier per K-step: tile k+1 is issued right after the barrier that publishes tile k and lands
//     while the 32 MFMAs of tile k run (2 LDS buffers).
#include <cstdlib>

#include "cmh_common.h"

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 g_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float g_f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t g_u32x4_t;

constexpr int gTile = 128;
constexpr int gRowBytes = 128;
constexpr int gTileBytes = gTile * gRowBytes;

__device__ __forceinline__ int g_swz(int row, int chunk) { return row * gRowBytes + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ float g_quick_gelu(float v) { return v / (1.0f + __expf(-1.702f * v)); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool F32>
__global__ __launch_bounds__(256) void gemm_glds_kernel(const char* __restrict__ X, const char* __restrict__ W,
                                                        const float* __restrict__ bias, const float* residual,
                                                        void* out, int M, int N, int K, int epi, int order) {
  __shared__ __attribute__((aligned(1024))) char lds[2][2][gTileBytes];   // [buf][0=W,1=X]

  constexpr int ELT = F32 ? 4 : 2;
  constexpr int BK = gRowBytes / ELT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wn = wid >> 1, wm = wid & 1;

  const int tiles_n = N / gTile;
  const int tiles_m = (M + gTile - 1) / gTile;
  const int total = tiles_n * tiles_m;
  const int bid = blockIdx.x;
  int logical = bid;
  if (order == 0 || order == 3) {   // blocks with equal bid%8 share an XCD: give each a contiguous tile range
    const int xcd = bid & 7, local = bid >> 3;
    const int q = total >> 3, r = total & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int tile_m, tile_n;
  if (order == 2) {                 // m fastest
    tile_n = logical / tiles_m;
    tile_m = logical - tile_n * tiles_m;
  } else if (order == 3) {          // bands of 8 m-tiles, m fastest inside a band: 8x8 super-tiles in flight
    const int band = logical / (8 * tiles_n);
    const int in_band = logical - band * 8 * tiles_n;
    const int band_rows = (tiles_m - band * 8) < 8 ? (tiles_m - band * 8) : 8;
    tile_n = in_band / band_rows;
    tile_m = band * 8 + (in_band - tile_n * band_rows);
  } else {                          // n fastest
    tile_m = logical / tiles_n;
    tile_n = logical - tile_m * tiles_n;
  }
  const int m0 = tile_m * gTile, n0 = tile_n * gTile;

  // DMA pieces: a tile is 16 pieces of 1 KiB (8 rows); wave `wid` moves pieces wid*4 .. wid*4+3 of each tile.
  const size_t row_stride = static_cast<size_t>(K) * ELT;
  const char* gW[4];
  const char* gX[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wid * 4 + i;
    const int row = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    gW[i] = W + static_cast<size_t>(n0 + row) * row_stride + chunk * 16;
    int xr = m0 + row;
    xr = xr < M ? xr : M - 1;
    gX[i] = X + static_cast<size_t>(xr) * row_stride + chunk * 16;
  }

  auto stage = [&](int buf, size_t koff) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wid * 4 + i;
      __builtin_amdgcn_global_load_lds((gptr_t)(gW[i] + koff), (lptr_t)(&lds[buf][0][piece * 1024]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gX[i] + koff), (lptr_t)(&lds[buf][1][piece * 1024]), 16, 0, 0);
    }
  };

  g_f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  const int frow = lane & 15;
  const int fq = lane >> 4;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    __syncthreads();   // tile kt has landed for every wave (vmcnt(0) + barrier); buffer cur^1 is free again
    if (kt + 1 < nk) stage(cur ^ 1, static_cast<size_t>(kt + 1) * gRowBytes);
    const char* tW = &lds[cur][0][0];
    const char* tX = &lds[cur][1][0];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 4 + fq;
      g_u32x4_t fw[4], fx[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fw[t] = *reinterpret_cast<const g_u32x4_t*>(tW + g_swz(wn * 64 + t * 16 + frow, chunk));
        fx[t] = *reinterpret_cast<const g_u32x4_t*>(tX + g_swz(wm * 64 + t * 16 + frow, chunk));
      }
      if constexpr (F32) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[a][s]), __uint_as_float(fx[b][s]),
                                                               acc[a][b], 0, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g_bf16x8_t, fw[a]),
                                                                __builtin_bit_cast(g_bf16x8_t, fx[b]), acc[a][b], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int m = m0 + wm * 64 + b * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = n0 + wn * 64 + a * 16 + fq * 4;
      g_f32x4_t v = acc[a][b];
      if (epi & EPI_BIAS) v += *reinterpret_cast<const g_f32x4_t*>(bias + n);
      if (epi & EPI_QUICKGELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = g_quick_gelu(v[j]);
      }
      if (epi & (EPI_GELU | EPI_RELU)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (epi & EPI_GELU) ? gelu_erf(v[j]) : fmaxf(v[j], 0.f);
      }
      const size_t o = static_cast<size_t>(m) * N + n;
      if (epi & EPI_RESIDUAL) v += *reinterpret_cast<const g_f32x4_t*>(residual + o);
      if (epi & EPI_OUT_BF16) {
        uint2 pk;
        pk.x = static_cast<uint32_t>(f32_to_bf16(v[0])) | (static_cast<uint32_t>(f32_to_bf16(v[1])) << 16);
        pk.y = static_cast<uint32_t>(f32_to_bf16(v[2])) | (static_cast<uint32_t>(f32_to_bf16(v[3])) << 16);
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(out) + o) = pk;
      } else {
        *reinterpret_cast<g_f32x4_t*>(static_cast<float*>(out) + o) = v;
      }
    }
  }
}

void launch_gemm_glds(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                      int M, int N, int K, int epi, hipStream_t st) {
  const int total = (N / gTile) * ((M + gTile - 1) / gTile);
  static const int order = []() { const char* e = getenv("CMH_GEMM_ORDER"); return e ? atoi(e) : 0; }();
  if (dt == CMH_F32)
    hipLaunchKernelGGL(gemm_glds_kernel<true>, dim3(total), dim3(256), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), bias, residual, out, M, N, K, epi, order);
  else
    hipLaunchKernelGGL(gemm_glds_kernel<false>, dim3(total), dim3(256), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), bias, residual, out, M, N, K, epi, order);
}

}  // namespace cmh
