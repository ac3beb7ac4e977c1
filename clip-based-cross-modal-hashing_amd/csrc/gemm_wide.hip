// GEMM v3 ("wide", persistent): out[M,N] = epi(X[M,K].W[N,K]^T), 160(m) x 256(n) tile per 512-thread workgroup,
// one workgroup per CU walking its share of the tiles.
//
// Why this shape.  On the encoder's GEMMs (M = 12 800 / 19 712 rows, N in {512,768,1536,2048,2304,3072}) the
// 128x128 kernels are limited by two things measured in round 1: LDS traffic (per K-step a 128x128 tile moves
// as many LDS bytes as it has MFMA cycles: <= ~50 % MFMA duty) and tile quantisation (N = 768 -> 600 tiles on
// 512 resident slots = 59 %).  A 160x256 tile, 8 waves as 2(m) x 4(n), 80x64 per wave:
//   * 320 MFMAs per K-step against 52 KB of LDS-DMA writes + 147 KB of fragment reads -> MFMA-bound in principle;
//   * tile counts of 237..248 (N = 512/768) or ~3-4 full rounds (N >= 1536) on 256 CUs: >= 93 % quantisation
//     efficiency on every encoder shape;
//   * 3 LDS stages x 52 KB = 156 KB of the CU's 160 KB: two K-steps of operands are in flight while a third is
//     being multiplied; waits are COUNTED (s_waitcnt vmcnt(pieces of one stage)), the barrier is a raw s_barrier,
//     so LDS-DMA stays in flight across barriers (cdna guide §5 "Pipelining across barriers").
// Why persistent.  With K = 512/768 a tile has only 8-12 K-steps and the non-overlapped per-tile cost (workgroup
// launch, cold prologue DMA, epilogue store drain; ~12 us measured vs ~1.2 us per K-step) was ~45 % of the time.
// Here the K-steps of ALL tiles of a workgroup form one flat pipeline: the first two stages of the next tile are
// issued during the last two K-steps of the current one and land while its epilogue runs.
// Everything else as in gemm_glds.hip: W rows feed the MFMA A operand (lane owns 4 consecutive n), lane-linear LDS
// image with the XOR swizzle on the DMA source chunk and on the ds_read_b128, fused epilogue, XCD-aware tile order
// (workgroups with equal blockIdx%8 share an XCD and take neighbouring tiles of one contiguous range, n fastest).
#include "cmh_common.h"

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 w_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float w_f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t w_u32x4_t;

constexpr int wBM = 160, wBN = 256;
constexpr int wRowBytes = 128;
constexpr int wWBytes = wBN * wRowBytes;            // 32 KB
constexpr int wXBytes = wBM * wRowBytes;            // 20 KB
constexpr int wStageBytes = wWBytes + wXBytes;      // 52 KB
constexpr int wXPieces = wBM / 8;                   // 20 pieces of 1 KiB

__device__ __forceinline__ int w_swz(int row, int chunk) { return row * wRowBytes + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ float w_quick_gelu(float v) { return v / (1.0f + __expf(-1.702f * v)); }

typedef const __attribute__((address_space(1))) void* w_gptr_t;
typedef __attribute__((address_space(3))) void* w_lptr_t;

template <bool F32>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const char* __restrict__ X, const char* __restrict__ W,
                                                        const float* __restrict__ bias, const float* residual,
                                                        void* out, int M, int N, int K, int epi) {
  __shared__ __attribute__((aligned(1024))) char lds[3 * wStageBytes];

  constexpr int ELT = F32 ? 4 : 2;
  constexpr int BK = wRowBytes / ELT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, wm = wid >> 2;
  const int sub = lane >> 3;
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const bool three = wid < wXPieces - 16;   // waves 0..3 move a third X piece per stage

  // ---- this workgroup's tiles: XCD x = blockIdx%8 owns a contiguous range of the n-fastest tile order ----
  const int tiles_n = N / wBN;
  const int tiles_m = (M + wBM - 1) / wBM;
  const int total = tiles_n * tiles_m;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const int q = total >> 3, r = total & 7;
  const int range_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int range_len = xcd < r ? q + 1 : q;
  const int my_tiles = slot < range_len ? (range_len - slot + per_xcd_blocks - 1) / per_xcd_blocks : 0;
  if (my_tiles == 0) return;

  const int nk = K / BK;
  const size_t row_stride = static_cast<size_t>(K) * ELT;
  const int total_steps = my_tiles * nk;

  // ---- issue side: DMA source pointers of the tile currently being staged ----------------------------------
  const char* gW[4];
  const char* gX[3];
  auto set_issue_tile = [&](int ti) {
    const int logical = range_lo + slot + ti * per_xcd_blocks;
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    const int m0 = tm * wBM, n0 = tn * wBN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wid * 4 + i) * 8 + sub;
      gW[i] = W + static_cast<size_t>(n0 + row) * row_stride + (((lane & 7) ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = (wid + 8 * i) * 8 + sub;
      int xr = m0 + row;
      xr = xr < M ? xr : M - 1;   // rows past M are computed on duplicated data and never stored
      gX[i] = X + static_cast<size_t>(xr) * row_stride + (((lane & 7) ^ (row & 7)) << 4);
    }
  };
  int issued = 0, issue_kt = 0, issue_tile = 0, issue_buf = 0;
  auto issue_next = [&]() {
    if (issued >= total_steps) return;
    char* base = lds + issue_buf * wStageBytes;
    const size_t koff = static_cast<size_t>(issue_kt) * wRowBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(gW[i] + koff), (w_lptr_t)(base + (wid * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(gX[i] + koff), (w_lptr_t)(base + wWBytes + (wid + 8 * i) * 1024), 16, 0, 0);
    if (three)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(gX[2] + koff), (w_lptr_t)(base + wWBytes + (wid + 16) * 1024), 16, 0, 0);
    ++issued;
    issue_buf = issue_buf == 2 ? 0 : issue_buf + 1;
    if (++issue_kt == nk) {
      issue_kt = 0;
      if (++issue_tile < my_tiles) set_issue_tile(issue_tile);
    }
  };

  set_issue_tile(0);
  issue_next();
  issue_next();

  int consumed = 0, cur = 0;
  for (int ti = 0; ti < my_tiles; ++ti) {
    w_f32x4_t acc[4][5];   // [n-tile][m-tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) acc[a][b] = w_f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      // stage `consumed` must have landed; at most ONE younger stage may stay in flight.  (Epilogue stores of the
      // previous tile are younger than both stages: the counted wait then also retires the DMAs - safe.)
      if (issued - consumed > 1) {
        if (three) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // stage visible to all waves; every wave is done with stage consumed-1
      issue_next();                   // refills the buffer that stage consumed-1 occupied
      const char* tW = lds + cur * wStageBytes;
      const char* tX = tW + wWBytes;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + fq;
        w_u32x4_t fw[4], fx[5];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          fw[t] = *reinterpret_cast<const w_u32x4_t*>(tW + w_swz(wn * 64 + t * 16 + frow, chunk));
#pragma unroll
        for (int t = 0; t < 5; ++t)
          fx[t] = *reinterpret_cast<const w_u32x4_t*>(tX + w_swz(wm * 80 + t * 16 + frow, chunk));
        if constexpr (F32) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 5; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[a][s]), __uint_as_float(fx[b][s]),
                                                                 acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w_bf16x8_t, fw[a]),
                                                                  __builtin_bit_cast(w_bf16x8_t, fx[b]), acc[a][b], 0, 0, 0);
        }
      }
      ++consumed;
      cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- epilogue of tile ti (the next tile's first two stages are already in flight) -----------------------
    const int logical = range_lo + slot + ti * per_xcd_blocks;
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    const int m0 = tm * wBM, n0 = tn * wBN;
    // All loads first, then all stores: vmcnt counts loads, stores and LDS-DMA in one in-order queue, so a load
    // issued behind a store (or waited for with DMA in flight) would wait for every older store to be acknowledged.
    w_f32x4_t bv[4];
    if (epi & EPI_BIAS) {
#pragma unroll
      for (int a = 0; a < 4; ++a) bv[a] = *reinterpret_cast<const w_f32x4_t*>(bias + n0 + wn * 64 + a * 16 + fq * 4);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] += bv[a];
    }
    if (epi & EPI_QUICKGELU) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][j] = w_quick_gelu(acc[a][b][j]);
    }
    if (epi & EPI_RESIDUAL) {
      w_f32x4_t rv[4][5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        int m = m0 + wm * 80 + b * 16 + frow;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          rv[a][b] = *reinterpret_cast<const w_f32x4_t*>(residual + static_cast<size_t>(m) * N + n0 + wn * 64 + a * 16 + fq * 4);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] += rv[a][b];
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int m = m0 + wm * 80 + b * 16 + frow;
      if (m >= M) continue;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int n = n0 + wn * 64 + a * 16 + fq * 4;
        const w_f32x4_t v = acc[a][b];
        const size_t o = static_cast<size_t>(m) * N + n;
        if (epi & EPI_OUT_BF16) {
          uint2 pk;
          pk.x = static_cast<uint32_t>(f32_to_bf16(v[0])) | (static_cast<uint32_t>(f32_to_bf16(v[1])) << 16);
          pk.y = static_cast<uint32_t>(f32_to_bf16(v[2])) | (static_cast<uint32_t>(f32_to_bf16(v[3])) << 16);
          *reinterpret_cast<uint2*>(static_cast<bf16_t*>(out) + o) = pk;
        } else {
          *reinterpret_cast<w_f32x4_t*>(static_cast<float*>(out) + o) = v;
        }
      }
    }
  }
}

bool gemm_wide_supported(int N) { return N % wBN == 0; }

int launch_gemm_wide(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                     int M, int N, int K, int epi, hipStream_t st) {
  const int total = (N / wBN) * ((M + wBM - 1) / wBM);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 8) cus = 256;
    cus &= ~7;   // whole groups of 8: blockIdx % 8 names the XCD share
  }
  int grid = total < cus ? ((total + 7) & ~7) : cus;
  if (dt == CMH_F32)
    hipLaunchKernelGGL(gemm_wide_kernel<true>, dim3(grid), dim3(512), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), bias, residual, out, M, N, K, epi);
  else
    hipLaunchKernelGGL(gemm_wide_kernel<false>, dim3(grid), dim3(512), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), bias, residual, out, M, N, K, epi);
  return 0;
}

}  // namespace cmh
