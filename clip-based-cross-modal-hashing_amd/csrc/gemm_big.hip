// GEMM v4 ("big", persistent): out[M,N] bf16 = epi(X[M,K].W[N,K]^T + bias), 256(m) x 256(n) tile per 512-thread workgroup,
// eight waves as 4(n) x 2(m) of 64 x 128 (two per SIMD), one workgroup per CU walking its share of the tiles.  OPT-IN (see the end).
//
// Why a second tile shape.  gemm_wide.hip's K loop runs at the rate at which a CU can be FED: 53 KB per K-step into
// LDS at the ~56-62 GB/s per CU that L2 + Infinity Cache deliver, against 1280 cycles of MFMA work - 98 FLOP per fetched byte caps
// the loop near 0.6 of the matrix pipe.  A 256 x 256 tile has 128 FLOP per byte (64 KB per K-step against 2048 MFMA cycles per SIMD)
// and 0.375 fragment reads per MFMA instead of 0.45.  It can only pay where the output has enough such tiles - in_proj / c_fc
// (N = 2304 / 3072 / 1536 / 2048: 246..600 tiles); the N = 512 / 768 launches (100..150 tiles for 256 CUs) stay on the 160 x 256
// kernel.  The vendor library's kernels for these shapes are of this form and are ahead on exactly these launches
// (profiles/r03_h_gemm_microbench_vs_vendor.txt).
//
// Same bits as gemm_wide.hip: every output element is the same chain of v_mfma_f32_16x16x32_bf16 (W rows as the A operand,
// k = 64 kt + 32 ks + 8 fq + j ascending in kt, ks), then + bias, QuickGELU (same formula), one rounding to bf16.
//
// Structure.  160 KB of LDS hold only TWO stages of 64 KB (256 W rows + 256 X rows of 128 B, lane-linear pieces with the XOR swizzle
// on the DMA source chunk and on the ds_read_b128), so a stage has one K-step of flight, not two.  Per K-step s:
//   first half : 32 MFMAs per wave on (s, k 0..31); the W fragments of the second half are fetched up front, the eight X tiles live in
//                ONE set of registers refilled tile by tile (right after a tile's four MFMAs its registers take the same rows' k 32..63);
//   middle     : lgkmcnt(0), counted vmcnt (stage s+1 landed), s_barrier -> every wave has finished reading buffer s%2;
//   second half: 32 MFMAs on (s, k 32..63); the X registers refill from stage s+1, its first W fragments are fetched, and the 8 LDS-DMA
//                pieces per wave of stage s+2 go into buffer s%2 - waves 0..3 among the first MFMAs, waves 4..7 among the last, so
//                that the two waves of a SIMD do not sit in the DMA issue together.
// The K-steps of all tiles of a workgroup form one flat pipeline (the next tile's first two stages are issued during the last
// K-step of the current one, so the epilogue's 16 stores per lane are YOUNGER than them and a counted wait lets them drain under
// the next tile's first K-step).  248 registers, no scratch.
#include <cstdlib>

#include "cmh_common.h"

#include <hip/hip_ext.h>

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 g_b16x8_t;
typedef __attribute__((ext_vector_type(4))) float g_f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t g_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned g_u2_t;
typedef const __attribute__((address_space(1))) void* g_gptr_t;
typedef __attribute__((address_space(3))) void* g_lptr_t;

constexpr int gT = 256;                  // tile rows and columns
constexpr int gRowB = 128;               // bytes of one K-step of a row (64 bf16)
constexpr int gHalf = gT * gRowB;        // 32 KB: one operand's part of a stage
constexpr int gStage = 2 * gHalf;        // 64 KB

__device__ __forceinline__ int g_swz(int row, int chunk) { return row * gRowB + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ float g_quick_gelu(float v) {   // gemm_wide.hip: w_quick_gelu, operation for operation
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
}

__global__ __launch_bounds__(512) void gemm_big_kernel(const char* __restrict__ X, const char* __restrict__ W,
                                                       const float* __restrict__ bias, uint16_t* __restrict__ out, int Mub, int N,
                                                       int K, int epi, const int* __restrict__ m_dev) {
  int M = Mub;
  if (m_dev) { const int md = *m_dev; M = md < Mub ? md : Mub; }
  __shared__ __attribute__((aligned(1024))) char lds[5 * gHalf];   // W stage s in buffer s % 2 at [0, 64 KB), X stage s in buffer s % 3 behind them: all 160 KB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, wm = wid >> 2;          // 4(n) x 2(m) waves of 64(n) x 128(m)
  const int frow = lane & 15, fq = lane >> 4;

  // ---- this workgroup's tiles: XCD x = blockIdx % 8 owns a contiguous range of the n-fastest tile order (gemm_wide.hip) ----
  const int tiles_n = N / gT;
  const int tiles_m = (M + gT - 1) / gT;
  const int total = tiles_n * tiles_m;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const int q = total >> 3, r = total & 7;
  const int range_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int range_len = xcd < r ? q + 1 : q;
  const int my_tiles = slot < range_len ? (range_len - slot + per_xcd_blocks - 1) / per_xcd_blocks : 0;
  if (my_tiles == 0) return;
  const int nk = K / 64;
  const uint32_t row_stride = static_cast<uint32_t>(K) * 2;

  // ---- issue side: a stage = 64 pieces of 1 KiB (8 rows); wave w moves pieces 8w .. 8w+7: waves 0..3 (group A) the W rows, waves
  // 4..7 (group B) the X rows.  Lane i of a piece fills LDS (row 8p + i/8, physical chunk i%8) and fetches logical chunk (i%8) ^ (row & 7).
  const bool group_b = wid >= 4;
  uint32_t off[8];
  const char* src_tile = W;       // W + n0 * row_stride for the W waves, X for the X waves
  auto set_issue_tile = [&](int ti) {
    const int logical = range_lo + slot + ti * per_xcd_blocks;
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    int sub = lane >> 3;
    asm volatile("" : "+v"(sub));   // (recomputed here: hoisted row constants would be spilled across the K loop)
    if (!group_b) {
      src_tile = W + static_cast<size_t>(tn) * gT * row_stride;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (wid * 8 + i) * 8 + sub;
        off[i] = static_cast<uint32_t>(row) * row_stride + (((lane & 7) ^ (row & 7)) << 4);
      }
    } else {
      src_tile = X;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = ((wid - 4) * 8 + i) * 8 + sub;
        int xr = tm * gT + row;
        xr = xr < M ? xr : M - 1;   // rows past M are computed on duplicated data and never stored
        off[i] = static_cast<uint32_t>(xr) * row_stride + (((lane & 7) ^ (row & 7)) << 4);   // < 4 GiB: checked on the host
      }
    }
  };
  int issue_kt = 0, issue_tile = 0, issue_buf = 0;
  auto issue_piece = [&](int i) {   // i is a compile-time constant at every call site
    char* dst = group_b ? lds + 2 * gHalf + issue_buf * gHalf + ((wid - 4) * 8 + i) * 1024 : lds + issue_buf * gHalf + (wid * 8 + i) * 1024;
    __builtin_amdgcn_global_load_lds((g_gptr_t)(src_tile + static_cast<size_t>(issue_kt) * gRowB + off[i]), (g_lptr_t)dst, 16, 0, 0);
  };
  // The K loop is one straight-line steady state: it issues a stage in every K-step.  The two stages issued past the workgroup's
  // last one re-stage its last tile into buffers nobody reads any more (drained before the kernel ends).
  auto issue_done = [&]() {
    issue_buf = group_b ? (issue_buf == 2 ? 0 : issue_buf + 1) : (issue_buf ^ 1);
    if (++issue_kt == nk) {
      issue_kt = 0;
      if (++issue_tile < my_tiles) set_issue_tile(issue_tile);
    }
  };

  // ---- fragment reads (inline asm: their waits are ours): the 16-row fragment tiles of an operand sit 2048 bytes apart, the second
  // 32-deep half is the first one's address XOR 64.  The W fragments of both halves of a K-step are held (2 x 4 tiles); the eight X
  // tiles live in ONE set of registers that is refilled tile by tile: right after a tile's four MFMAs its registers take the same
  // rows of the next half-step (the other k half of this stage in the first half, the next stage's first half after the barrier).
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((g_lptr_t)lds));
  const uint32_t aW = lds_base + g_swz(wn * 64 + frow, fq);
  const uint32_t aX = lds_base + 2 * gHalf + g_swz(wm * 128 + frow, fq);
#define G_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define G_WAIT_ALL(cnt)                                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                                \
               : "+v"(f0w[0]), "+v"(f0w[1]), "+v"(f0w[2]), "+v"(f0w[3]), "+v"(f1w[0]), "+v"(f1w[1]), "+v"(f1w[2]), "+v"(f1w[3]), \
                 "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7])::"memory")
  g_u32x4_t f0w[4], f1w[4], xf[8];
  g_f32x4_t acc[4][8];   // [n-tile][m-tile]; lane (frow, fq) holds out[m = .. + frow][n = .. + 4 fq + j]
  auto mfma = [&](const g_u32x4_t& fw, const g_u32x4_t& fx, g_f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g_b16x8_t, fw), __builtin_bit_cast(g_b16x8_t, fx), c, 0, 0, 0);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
#define G_XREAD(b, addr)                                            \
  do {                                                              \
    if (b == 0) G_READ(xf[0], addr, 0);                             \
    if (b == 1) G_READ(xf[1], addr, 2048);                          \
    if (b == 2) G_READ(xf[2], addr, 4096);                          \
    if (b == 3) G_READ(xf[3], addr, 6144);                          \
    if (b == 4) G_READ(xf[4], addr, 8192);                          \
    if (b == 5) G_READ(xf[5], addr, 10240);                         \
    if (b == 6) G_READ(xf[6], addr, 12288);                         \
    if (b == 7) G_READ(xf[7], addr, 14336);                         \
  } while (0)

  // ---- prologue: two stages, the first one's fragments ----
  zero_acc();
  set_issue_tile(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece(i);
  issue_done();
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_piece(i);
  issue_done();
  if (group_b) {     // the X operand runs one stage further ahead (three buffers)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(i);
    issue_done();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // X stage 0 landed (two younger stages may fly)
  } else {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // W stage 0 landed (stage 1 may fly)
  }
  __builtin_amdgcn_s_barrier();
  G_READ(f0w[0], aW, 0); G_READ(f0w[1], aW, 2048); G_READ(f0w[2], aW, 4096); G_READ(f0w[3], aW, 6144);
#pragma unroll
  for (int b = 0; b < 8; ++b) G_XREAD(b, aX);

  int cw = 0, cx = 0;            // W buffer (s % 2) and X buffer (s % 3) of the K-step being multiplied
  int sp = 0;                    // K-steps for which the previous tile's 16 stores per lane are younger than the stage a counted wait needs
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int kt = 0; kt < nk; ++kt) {
      const uint32_t w1 = (aW + static_cast<uint32_t>(cw) * gHalf) ^ 64u, x1 = (aX + static_cast<uint32_t>(cx) * gHalf) ^ 64u;
      // ---- first half: the W fragments of the second half go out, then everything older (this half's fragments) has landed
      G_READ(f1w[0], w1, 0); G_READ(f1w[1], w1, 2048); G_READ(f1w[2], w1, 4096); G_READ(f1w[3], w1, 6144);
      G_WAIT_ALL(4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int a = 0; a < 4; ++a) mfma(f0w[a], xf[b], acc[a][b]);
#ifndef G_ABL_NOXREAD
        G_XREAD(b, x1);                       // this tile's rows, k 32..63 of the same stage
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- middle: own reads done; stage s+1 landed (older than the previous tile's stores, when those are in flight); barrier
      G_WAIT_ALL(0);
      // vmcnt is one in-order queue.  Waves 0..3 (W pieces): stage s+1 is the youngest stage issued - unless the previous tile's 16
      // stores followed it (first K-step of a tile).  Waves 4..7 (X pieces): one more stage of 8 pieces is younger, and the stores sit
      // between stages for the first two K-steps of a tile.
      if (group_b) {
        if (sp > 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        if (sp == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      sp = sp > 0 ? sp - 1 : 0;
#ifndef G_ABL_NOBARRIER
      __builtin_amdgcn_s_barrier();
#endif
      // ---- second half: MFMAs on the second k half; stage s+2 into the buffer just released (group A's pieces among the first
      // MFMAs, group B's among the last: the two waves of a SIMD do not sit in the DMA issue together), fragments of stage s+1
      const int nw = cw ^ 1, nx = cx == 2 ? 0 : cx + 1;
      const uint32_t w0 = aW + static_cast<uint32_t>(nw) * gHalf, x0 = aX + static_cast<uint32_t>(nx) * gHalf;
      G_READ(f0w[0], w0, 0); G_READ(f0w[1], w0, 2048); G_READ(f0w[2], w0, 4096); G_READ(f0w[3], w0, 6144);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
#ifndef G_ABL_NODMA
        if (!group_b) { if (b < 4) issue_piece(2 * b); }
        else { if (b >= 4) issue_piece(2 * (b - 4)); }
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef G_ABL_NODMA
          if (a == 2) {
            if (!group_b) { if (b < 4) issue_piece(2 * b + 1); }
            else { if (b >= 4) issue_piece(2 * (b - 4) + 1); }
          }
#endif
          mfma(f1w[a], xf[b], acc[a][b]);
        }
#ifndef G_ABL_NOXREAD
        G_XREAD(b, x0);                       // the next stage's first k half
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      issue_done();
      __builtin_amdgcn_sched_barrier(0);
      cw = nw;
      cx = nx;
    }

    // ---- epilogue of tile ti (the next tile's first two stages are already in flight) ----
    const int logical = range_lo + slot + ti * per_xcd_blocks;
    const int tm = logical / tiles_n, tn = logical - tm * tiles_n;
    const int m0 = tm * gT, n0 = tn * gT;
    const bool full = m0 + gT <= M;
    // all loads first, then all stores: a load issued behind a store waits for every older store to be acknowledged
    g_f32x4_t bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
      bv[a] = (epi & EPI_BIAS) ? *reinterpret_cast<const g_f32x4_t*>(bias + n0 + wn * 64 + a * 16 + fq * 4) : g_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const g_f32x4_t bv0 = bv[2 * pr], bv1 = bv[2 * pr + 1];
      // v_permlane16_swap exchanges, between the lane rows (fq, fq+1), the packed words of two neighbouring n-tiles: an even lane row
      // then owns 8 consecutive n of tile 2pr and an odd one 8 consecutive n of tile 2pr+1 -> 16-byte stores (gemm_wide.hip)
      const int col = n0 + wn * 64 + pr * 32 + (fq & 1) * 16 + (fq & 2) * 4;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        g_f32x4_t v0 = acc[2 * pr][b] + bv0, v1 = acc[2 * pr + 1][b] + bv1;
        if (epi & EPI_QUICKGELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { v0[j] = g_quick_gelu(v0[j]); v1[j] = g_quick_gelu(v1[j]); }
        }
        const uint32_t lo0 = pack_bf16x2(v0[0], v0[1]), lo1 = pack_bf16x2(v0[2], v0[3]);
        const uint32_t hi0 = pack_bf16x2(v1[0], v1[1]), hi1 = pack_bf16x2(v1[2], v1[3]);
        const g_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo0, hi0, false, false);
        const g_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo1, hi1, false, false);
        const int m = m0 + wm * 128 + b * 16 + frow;
        if (m < M) *reinterpret_cast<g_u32x4_t*>(out + static_cast<size_t>(m) * N + col) = g_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
      }
    }
    if (full) {
      sp = 2;
    } else {   // a partial tile may have skipped store instructions: no counting on them
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sp = 0;
    }
    zero_acc();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-issued stages must land before the LDS is released
#undef G_READ
#undef G_WAIT_ALL
#undef G_XREAD
}

// Which launches could come here: bf16 operands and output, bias / QuickGELU only, N % 256 == 0, and enough 256 x 256 tiles that a round
// of them keeps the chip busy (the N = 512 / 768 residual GEMMs never do).
// OFF by default (cmh_set_gemm_big(1) / CMH_GEMM_BIG=1 switch it on).  Measured (profiles/r03_i_gemm_big.txt): ahead of
// the wide kernel where K is long - 12 800 x 2304 x 3072: 1050 TFLOP/s against 977 (a first version with four waves of 128 x 128 and
// the accumulators in AGPRs: 1161-1188), 12 800 x 3072 x 6144: 1152 against 1065 - level on v_fc1 / t_qkv / the 7000-row probe, behind on
// v_qkv (57.8 us against 52.3) and t_fc1 (37.6 against 35.1): with 8-12 K-steps per tile the one-K-step prefetch distance and the
// 128-value-per-lane epilogue per tile cost what the feed rate gains.
static int g_big_on = -1;   // cmh_set_gemm_big: -1 = from the environment
bool gemm_big_takes(int dt, int M, int N, int K, int epi, const void* residual) {
  static const bool env_on = []() { const char* e = getenv("CMH_GEMM_BIG"); return e && e[0] == '1'; }();
  const bool off = g_big_on < 0 ? !env_on : g_big_on == 0;
  if (off || dt != CMH_BF16 || residual) return false;
  if (N % gT != 0 || K % 64 != 0 || K < 128) return false;
  if ((epi & ~(EPI_BIAS | EPI_QUICKGELU | EPI_OUT_BF16)) || !(epi & EPI_OUT_BF16)) return false;
  if (static_cast<size_t>(M) * K * 2 >= (1ull << 32) || static_cast<size_t>(gT) * K * 2 >= (1ull << 32)) return false;
  const int tiles = (N / gT) * ((M + gT - 1) / gT);
  return tiles >= 240;
}

int launch_gemm_big(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int epi, hipStream_t st,
                    const int32_t* m_dev, hipEvent_t ev0, hipEvent_t ev1) {
  static const int cus = []() {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n < 8) n = 256;
    return n & ~7;
  }();
  const int total = (N / gT) * ((M + gT - 1) / gT);
  const int grid = total < cus ? ((total + 7) & ~7) : cus;
  if (ev0)
    hipExtLaunchKernelGGL(gemm_big_kernel, dim3(grid), dim3(512), 0, st, ev0, ev1, 0, static_cast<const char*>(A),
                          static_cast<const char*>(W), bias, static_cast<uint16_t*>(out), M, N, K, epi, m_dev);
  else
    hipLaunchKernelGGL(gemm_big_kernel, dim3(grid), dim3(512), 0, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,
                       static_cast<uint16_t*>(out), M, N, K, epi, m_dev);
  return 0;
}

}  // namespace cmh

extern "C" int cmh_set_gemm_big(int32_t on) {
  CMH_CHECK_ARG(on >= -1 && on <= 1, "set_gemm_big: %d (-1 default, 0 off, 1 on)", on);
  cmh::g_big_on = on;
  return CMH_OK;
}
