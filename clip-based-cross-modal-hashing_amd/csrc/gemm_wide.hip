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
// x * sigmoid(1.702 x) with v_exp + v_rcp (1 ulp) instead of an IEEE division (~10 VALU ops): the epilogue applies it to
// 80 accumulators per lane while the matrix pipe idles.
__device__ __forceinline__ float w_quick_gelu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
}

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

  // ---- rotated, software-pipelined K loop --------------------------------------------------------------------
  // Per K-step s (stage s in LDS buffer s%3), with F0 = fragments of (s, k 0..31) already in registers:
  //   1. issue the ds_reads of F1 = (s, k 32..63)
  //   2. 20 MFMAs on F0                                    <- cover the F1 reads
  //   3. lgkmcnt(0) (own F1 reads done), vmcnt: stage s+1 landed (stage s+2 may stay in flight), s_barrier
  //      -> every wave has finished reading buffer s%3 and sees stage s+1
  //   4. LDS-DMA of stage s+3 into buffer s%3              <- two full K-steps of flight
  //   5. issue the ds_reads of F0' = (s+1, k 0..31)
  //   6. 20 MFMAs on F1                                    <- cover the F0' reads and the DMA issue
  // Fragment reads are inline asm so that THEIR waits are ours: hipcc's own bookkeeping turns any ds_read that is
  // still pending at a loop back-edge into s_waitcnt lgkmcnt(0) in front of the next MFMA block, which would serialise
  // reads and MFMAs.  The waits below are counted (LDS returns in order) and carry the fragment registers as "+v"
  // operands, so no MFMA that consumes them can be scheduled above the wait (cdna guide §5.4 rule 18).
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((w_lptr_t)lds));
  auto load_frags = [&](w_u32x4_t (&fw)[4], w_u32x4_t (&fx)[5], int buf, int ks) {
    const uint32_t tW = lds_base + buf * wStageBytes;
    const uint32_t tX = tW + wWBytes;
    const int chunk = ks * 4 + fq;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      asm volatile("ds_read_b128 %0, %1" : "=v"(fw[t]) : "v"(tW + w_swz(wn * 64 + t * 16 + frow, chunk)));
#pragma unroll
    for (int t = 0; t < 5; ++t)
      asm volatile("ds_read_b128 %0, %1" : "=v"(fx[t]) : "v"(tX + w_swz(wm * 80 + t * 16 + frow, chunk)));
  };
#define W_WAIT_FRAGS(cnt, fw, fx)                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                   \
               : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]),    \
                 "+v"(fx[3]), "+v"(fx[4])::"memory")

  set_issue_tile(0);
  issue_next();
  issue_next();
  issue_next();
  // stage 0 landed (two younger stages may fly), visible to everyone
  if (issued >= 3) {
    if (three) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  int consumed = 0, cur = 0;
  int since_epi = 2, epi_stores = 0;   // K-steps since the last epilogue / store instructions it issued per wave
  for (int ti = 0; ti < my_tiles; ++ti) {
    w_f32x4_t acc[4][5];   // [n-tile][m-tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b) acc[a][b] = w_f32x4_t{0.f, 0.f, 0.f, 0.f};

    w_u32x4_t f0w[4], f0x[5], f1w[4], f1x[5];
    load_frags(f0w, f0x, cur, 0);

    auto mma = [&](const w_u32x4_t (&fw)[4], const w_u32x4_t (&fx)[5]) {
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
    };

    for (int kt = 0; kt < nk; ++kt) {
      load_frags(f1w, f1x, cur, 1);                                   // 1
      W_WAIT_FRAGS(9, f0w, f0x);                                      //    F0 (older than the 9 F1 reads) is in registers
      __builtin_amdgcn_sched_barrier(0);
      mma(f0w, f0x);                                                  // 2
      __builtin_amdgcn_sched_barrier(0);
      // 3: own LDS reads done; stage consumed+1 landed, at most one younger stage still in flight.  (Epilogue stores of
      //    a previous tile are younger than the DMAs: the counted wait then retires the DMAs as well - safe.)
      W_WAIT_FRAGS(0, f1w, f1x);
      {
        // vmcnt is ONE in-order queue of loads, stores and LDS-DMA.  Younger than the stage we need are: one more
        // stage (P pieces) if the pipeline is still being fed, and - during the first two K-steps after an epilogue -
        // that epilogue's S stores (the stage we need was issued before them).  Allowing exactly P+S outstanding
        // lets the stores drain behind the MFMAs instead of stalling every wave at the barrier.
        const int allow = (issued - consumed > 2 ? (three ? 7 : 6) : 0) + (since_epi < 2 ? epi_stores : 0);
        ++since_epi;
        switch (allow) {
          case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
          case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
          case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
          case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
          case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
          case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      }
      __builtin_amdgcn_s_barrier();
      // 4-6 interleaved: one DMA piece / one fragment read between groups of 3 MFMAs.  Issued as one burst, the 52
      // pieces of a stage queue up in the CU's address path (~16 clk each) and every wave sits in a VMEM issue stall
      // while the matrix pipe idles; trickled in, they ride under the MFMAs.
      const int nxt = cur == 2 ? 0 : cur + 1;
      {
        const bool do_issue = issued < total_steps;
        char* ibase = lds + issue_buf * wStageBytes;
        const size_t koff = static_cast<size_t>(issue_kt) * wRowBytes;
        const bool pref = kt + 1 < nk;
        const uint32_t nW = lds_base + nxt * wStageBytes, nX = nW + wWBytes;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
          if (i % 3 == 0 && do_issue) {
            const int p = i / 3;
            if (p < 4)
              __builtin_amdgcn_global_load_lds((w_gptr_t)(gW[p] + koff), (w_lptr_t)(ibase + (wid * 4 + p) * 1024), 16, 0, 0);
            else if (p < 6)
              __builtin_amdgcn_global_load_lds((w_gptr_t)(gX[p - 4] + koff),
                                               (w_lptr_t)(ibase + wWBytes + (wid + 8 * (p - 4)) * 1024), 16, 0, 0);
            else if (three)
              __builtin_amdgcn_global_load_lds((w_gptr_t)(gX[2] + koff), (w_lptr_t)(ibase + wWBytes + (wid + 16) * 1024), 16, 0, 0);
          }
          if (pref && i >= 1 && i <= 9) {
            const int t = i - 1;
            if (t < 4)
              asm volatile("ds_read_b128 %0, %1" : "=v"(f0w[t]) : "v"(nW + w_swz(wn * 64 + t * 16 + frow, fq)));
            else
              asm volatile("ds_read_b128 %0, %1" : "=v"(f0x[t - 4]) : "v"(nX + w_swz(wm * 80 + (t - 4) * 16 + frow, fq)));
          }
          const int a = i / 5, b = i % 5;
          if constexpr (F32) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(f1w[a][s]), __uint_as_float(f1x[b][s]),
                                                               acc[a][b], 0, 0, 0);
          } else {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w_bf16x8_t, f1w[a]),
                                                                __builtin_bit_cast(w_bf16x8_t, f1x[b]), acc[a][b], 0, 0, 0);
          }
          if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (do_issue) {
          ++issued;
          issue_buf = issue_buf == 2 ? 0 : issue_buf + 1;
          if (++issue_kt == nk) {
            issue_kt = 0;
            if (++issue_tile < my_tiles) set_issue_tile(issue_tile);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      ++consumed;
      cur = nxt;
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
    if (epi & (EPI_GELU | EPI_RELU)) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[a][b][j] = (epi & EPI_GELU) ? gelu_erf(acc[a][b][j]) : fmaxf(acc[a][b][j], 0.f);
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
    // Stores.  Row-scattered 8-byte stores are issue-bound (measured: ~5.5 us per tile, i.e. as much as 5 K-steps),
    // so the bf16 path first widens them: v_permlane16_swap exchanges, between the lane pairs (l, l+16), the packed
    // words of two neighbouring n-tiles, after which an even lane-row owns 8 consecutive n of tile a and an odd
    // lane-row 8 consecutive n of tile a+1 -> one 16-byte store per lane, half the store instructions.
    if (epi & EPI_OUT_BF16) {
      const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;    // + 32*pair
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int m = m0 + wm * 80 + b * 16 + frow;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint32_t lo[2], hi[2];   // packed words of tiles a = 2pr (lo) and 2pr+1 (hi)
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
            hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
          }
          typedef __attribute__((ext_vector_type(2))) unsigned w_u2_t;
          const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
          const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
          w_u32x4_t pk;
          pk[0] = s0[0]; pk[1] = s1[0]; pk[2] = s0[1]; pk[3] = s1[1];
          if (m < M && !(epi & 256))   // 256 = timing-only ablation: skip stores
            *reinterpret_cast<w_u32x4_t*>(static_cast<bf16_t*>(out) + static_cast<size_t>(m) * N + col + 32 * pr) = pk;
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int m = m0 + wm * 80 + b * 16 + frow;
        if (m >= M || (epi & 256)) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = n0 + wn * 64 + a * 16 + fq * 4;
          *reinterpret_cast<w_f32x4_t*>(static_cast<float*>(out) + static_cast<size_t>(m) * N + n) = acc[a][b];
        }
      }
    }
    // bookkeeping for the counted waits of the next tile's first two K-steps
    if (m0 + wBM > M || (epi & 256)) {   // partial tile: some store instructions were skipped -> drain, count nothing
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      epi_stores = 0;
    } else {
      epi_stores = (epi & EPI_OUT_BF16) ? 10 : 20;
    }
    since_epi = 0;
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
