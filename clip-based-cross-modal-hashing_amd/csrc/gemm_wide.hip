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
// launch, cold prologue DMA, epilogue) was ~45 % of the time.  Here the K-steps of ALL tiles of a workgroup form one
// flat pipeline: the first stages of the next tile are issued during the last K-steps of the current one.
// Why deferred stores.  A CU drains stores at only ~10 B/clk: the 80 KB of a bf16 tile take ~4 us if the waves sit in
// the epilogue until their stores are accepted (measured by ablation: 727 -> 1000 TF/s without stores).  So a tile's
// packed outputs stay in 40 VGPRs and leave one 16-byte store per wave per K-step UNDER the next tile's MFMAs; only a
// workgroup's last tile (and f32 / partial tiles) stores from the epilogue.
// Everything else as in gemm_glds.hip: W rows feed the MFMA A operand (lane owns 4 consecutive n), lane-linear LDS
// image with the XOR swizzle on the DMA source chunk and on the ds_read_b128, fused epilogue, XCD-aware tile order
// (workgroups with equal blockIdx%8 share an XCD and take neighbouring tiles of one contiguous range, n fastest).
#include <cstdlib>
#include <cstring>

#include "cmh_common.h"

#include <hip/hip_ext.h>

#include <type_traits>

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 w_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 w_f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 w_f16x2_t;
typedef __attribute__((ext_vector_type(4))) float w_f32x4_t;
typedef __attribute__((ext_vector_type(2))) float w_f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t w_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned w_u2_t;

constexpr int wBM = 160, wBN = 256;
constexpr int wRowBytes = 128;
constexpr int wWBytes = wBN * wRowBytes;            // 32 KB

__device__ __forceinline__ int w_swz(int row, int chunk) { return row * wRowBytes + ((chunk ^ (row & 7)) << 4); }
// x * sigmoid(1.702 x) with v_exp + v_rcp (1 ulp) instead of an IEEE division (~10 VALU ops): the epilogue applies it to
// 80 accumulators per lane while the matrix pipe idles.
__device__ __forceinline__ float w_quick_gelu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));   // exp(-1.702 v) = 2^(-1.702 log2(e) v): one multiply
}

// 16-byte output store.  Diagnostic builds choose a cache policy with -DW_STORE_POLICY=n: 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 sc0,
// 6 nt through the compiler's builtin (counted by hipcc's own vmcnt bookkeeping); 7 no store at all, decided at run time (the
// address is never odd) so that the epilogue's arithmetic stays: what the output stores cost a launch (timing only).
#ifndef W_STORE_POLICY
#define W_STORE_POLICY 0
#endif
__device__ __forceinline__ void w_store16(void* p, const w_u32x4_t& v) {
#if W_STORE_POLICY == 1
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif W_STORE_POLICY == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif W_STORE_POLICY == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif W_STORE_POLICY == 4
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif W_STORE_POLICY == 5
  asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif W_STORE_POLICY == 6
  __builtin_nontemporal_store(v, reinterpret_cast<w_u32x4_t*>(p));
#elif W_STORE_POLICY == 7
  if (reinterpret_cast<uintptr_t>(p) & 1) *reinterpret_cast<w_u32x4_t*>(p) = v;
#else
  *reinterpret_cast<w_u32x4_t*>(p) = v;
#endif
}

typedef const __attribute__((address_space(1))) void* w_gptr_t;
typedef __attribute__((address_space(3))) void* w_lptr_t;

// cache policy of the LDS-DMA loads (the builtin's aux operand: 1 sc0, 2 nt, 16 sc1), per operand; diagnostic builds: -DW_X_CPOL=2 ...
#ifndef W_X_CPOL
#define W_X_CPOL 0
#endif
#ifndef W_W_CPOL
#define W_W_CPOL 0
#endif
#ifdef W_STAMPS
__device__ unsigned g_wide_stamps[256 * 8 * 4];
#endif
#ifdef W_TIMELINE   // diagnostic build only (tools/wide_timeline.py): wall-clock (100 MHz) stamps of each workgroup's phases
__device__ unsigned long long g_wide_tl[256 * 8];
#define W_TL(k) do { if (threadIdx.x == 0) g_wide_tl[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define W_TL(k) do { } while (0)
#endif

// fp8 operands (DT = 2, the throughput mode of BASELINE configs[4]): OCP e4m3 rows of 128 bytes = 128 k per K-step, one
// v_mfma_scale_f32_16x16x128_f8f6f4 per (n-tile, m-tile) and K-step (block scales fixed at 2^0; twice the bf16 MFMA's flops per
// cycle), the dequantisation `acc * alpha * colscale[n]` (per-tensor activation scale x per-output-channel weight scale) fused
// with the bias into one FMA of the epilogue, and optionally an e4m3 OUTPUT (`* oscale`, clamped to +-448) for the next GEMM.
struct WideScales {
  const float* colscale;   // [N] or null (EPI_SCALE)
  float alpha;             // EPI_SCALE: acc *= alpha * colscale[n]
  float oscale;            // EPI_OUT_FP8: out = e4m3(clamp(v * oscale))
  int kreal;               // TN: rows of the K-major operands that exist (K is padded to whole K-steps)
  float* colsum;           // TN, optional: [ksplit, M] partial column sums of the Xk operand (wgrad: the bias gradient's first stage)
  const int* m_dev;        // optional: the real row count on the device (M is then an upper bound)
};

// GROUPED launches (template parameter GRP; round 4).  The two CLIP towers are 12 blocks of the same four GEMMs, and one tower's
// launch leaves the chip idle where the other has work: the text GEMMs (K = 512, 8 K-steps per tile) cannot amortise a launch's
// fixed third - first stage landing on 256 CUs at once, last tile's epilogue and store drain - and the image GEMMs end with
// 0.2-0.4 of a round of CUs idle.  A grouped launch gives ONE persistent grid two problems: the kernel's own arguments describe the
// first (the longer K: the image tower's), WideProblem the second; a workgroup walks its share of the first problem's tiles and then
// continues - same XCD, same stride - into the second problem's, so the workgroups that got one image tile fewer take the first
// text tiles, the flat K-step pipeline runs across the problem switch like across any tile boundary (no second prologue, no
// drain in between), and per block of the towers there is one launch instead of two.  Both problems share the epilogue flags, the
// operand / output types and the tile height; per-problem are the pointers, M (or its device-side count), N, K and the fp8 scales.
// Every output element sees the arithmetic of an ungrouped launch (same K order, same epilogue): bit-identical results.
struct WideProblem {
  const char* X; const char* W; const float* bias; const float* residual; void* out;
  const float* colscale; const int* m_dev;
  int Mub, N, K;
  float alpha, oscale;
};

// TN operands (wgrad: dW[O,I] = dY^T X with dY [M,O] and X [M,I] as the backward pass has them, the reduction index m being the
// SLOW axis of both): out[Mm,Nn] = sum_k Xk[k,m] Wk[k,n].  A stage holds 64 k-rows of both operands as they lie in memory (rows
// of 512 B = this tile's 256 n, rows of 256 B = its 128 m; LDS-DMA pieces are whole rows, 16-byte chunks XOR-swizzled on the
// source side), and the MFMA fragments - 8 consecutive k of one n or m per lane - come out of LDS through the transposing read
// `ds_read_b64_tr_b16` (4 k-rows x 16 columns per 16-lane group, two reads per fragment).  No transposed copy of any operand is
// ever written: round 1 spent 3.2 ms of the 23.6 ms training step in transpose kernels (11.6 GB of traffic per step).
__device__ const uint4 g_wide_zero[16] = {};   // 256 zero bytes: the source of k-rows past the end of the Xk operand
typedef __attribute__((ext_vector_type(2))) uint32_t w_u32x2_t;

// The kernel's body.  ONE: the workgroup computes exactly the (virtual) tile `one_tile` instead of its share of the persistent tile
// walk - what the multi-problem wgrad launch below needs (gemm_wide_tn_multi_kernel: every workgroup looks its problem up and runs one
// long-K tile of it); all other launches pass ONE = false and nothing of it remains in their code.
template <int DT, int OK, int MF, bool TN = false, bool GRP = false, bool DGE = false, bool ONE = false>   // DT: 0 f32, 1 bf16, 2 fp8 operands; OK: 0 f32, 1 16-bit (bf16 /
                                    // fp16), 2 fp8 outputs; MF = 16-row m-fragments per wave: tile rows = 32*MF (160, 128 or 96); TN, GRP: above;
                                    // DGE: the QuickGELU of a tile runs under the NEXT tile's MFMAs (see "deferred QuickGELU" below)
__device__ __forceinline__ void gemm_wide_body(const char* __restrict__ X, const char* __restrict__ W,
                                               const float* __restrict__ bias_a, const float* residual_a,
                                               void* out_a, int Mub, int N_a, int K, int epi, int ksplit, int ordG,
                                               WideScales sc_a, WideProblem p1, int one_tile) {
  static_assert(!ONE || (TN && !GRP && !DGE), "single-tile workgroups: the multi-problem wgrad launch only");
  // The names the compute side (K loop, epilogue) reads.  Plain launches never reassign them - they ARE the arguments; a grouped
  // launch moves them to the second problem once per workgroup (to_problem1 below).  The issue side, which runs up to three stages
  // ahead of the compute side, keeps its own view (i_* in set_issue_tile).
  const float* bias = bias_a;
  const float* residual = residual_a;
  void* out = out_a;
  int N = N_a;
  WideScales sc = sc_a;
  // M: the row count; sc.m_dev (packed text rows) holds the real one on the device, Mub is then only an upper bound
  int M = Mub;
  if (sc.m_dev) { const int md = *sc.m_dev; M = md < Mub ? md : Mub; }
  int M1 = 0;          // GRP: the second problem's row count
  if constexpr (GRP) {
    M1 = p1.Mub;
    if (p1.m_dev) { const int md = *p1.m_dev; M1 = md < M1 ? md : M1; }
  }
  constexpr bool F32 = DT == 0, FP8 = DT == 2, OUTBF = OK == 1, OUT8 = OK == 2;
  constexpr int BMt = 32 * MF;                       // tile rows
  constexpr int WR = 16 * MF;                        // rows per wave
  constexpr int STG = wWBytes + BMt * wRowBytes;     // bytes per stage: 52 KB (MF = 5) or 48 KB (MF = 4)
  constexpr int XP = BMt / 8;                        // X pieces of 1 KiB per stage: 20 or 16
  constexpr int NPEND = OUT8 ? MF : 2 * MF;          // 16-byte stores per lane per tile
  // how many of them wait under the next tile's MFMAs (fp8, 160 rows, 16-bit output: only 4 - with all 10 pending next to 72 fragment
  // registers and 80 accumulators the allocator spills inside the K loop; the other 6 leave from the epilogue)
  constexpr int NDEFER = (FP8 && MF == 5 && OK == 1) ? 4 : NPEND;
  constexpr int NM = 4 * MF;                         // MFMAs per 32-deep half-step
  static_assert(MF >= 3 && MF <= 5, "wave layout: 2(m) x 4(n) waves of MF x 4 fragments");
  static_assert(!TN || (DT == 1 && OK == 0 && MF == 4), "TN: bf16 operands, f32 (split-K) output, 128-row tile");
  static_assert(!GRP || !TN, "grouped launches: the NT forward GEMMs only (no split-K, plain n-fastest tile order)");
  static_assert(!DGE || (DT == 1 && OK == 1 && MF == 4 && !TN), "deferred QuickGELU: bf16 operands, 16-bit output, 128-row tile");
  constexpr int PSTEP = MF == 3 ? 2 : 3;             // one LDS-DMA piece per PSTEP MFMAs: 6 (MF = 3, 4) or 7 pieces in 4*MF MFMAs
  __shared__ __attribute__((aligned(1024))) char lds[3 * STG];

  constexpr int ELT = F32 ? 4 : (FP8 ? 1 : 2);
  constexpr int BK = wRowBytes / ELT;

  const int tid = threadIdx.x;
  W_TL(0);   // entry
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid & 3, wm = wid >> 2;
  const int sub = lane >> 3;
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const bool three = wid < XP - 16;   // waves 0..3 move a third X piece per stage (MF = 5)
  // every wave issues >= 6 pieces per stage (the counted vmcnt waits rely on it): with 12 X pieces (MF = 3) waves 4..7 have no
  // second X piece of their own and fetch their first one again (same bytes to the same LDS address)
  const int xpiece1 = wid + 8 < XP ? wid + 8 : wid;

  // ---- this workgroup's tiles: XCD x = blockIdx%8 owns a contiguous range of the n-fastest tile order ----
  int tiles_n = N / wBN;
  const int tiles_m = (M + BMt - 1) / BMt;
  // split-K (wgrad: few output tiles, very long K): virtual tile v = split * base_total + tile computes K-steps
  // [split * nk, (split + 1) * nk) into the f32 partial plane out + split * M * N (summed by splitk_reduce_kernel)
  const int base_total = tiles_n * tiles_m;
  const int total = base_total * ksplit;
  const int xcd = blockIdx.x & 7, slot = ONE ? 0 : blockIdx.x >> 3, per_xcd_blocks = ONE ? 1 : gridDim.x >> 3;
  const int q = total >> 3, r = total & 7;
  const int range_lo = ONE ? one_tile : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
  const int range_len = ONE ? 1 : (xcd < r ? q + 1 : q);
  // GRP: this XCD's contiguous range of the SECOND problem's tiles follows its range of the first in the workgroups' stride:
  // position j = slot + ti * per_xcd_blocks of the concatenation, so a workgroup's first n_first tiles are the first problem's
  int tiles_n1 = 0, range_lo1 = 0, range_len1 = 0, n_first = 0;
  if constexpr (GRP) {
    tiles_n1 = p1.N / wBN;
    const int total1 = tiles_n1 * ((M1 + BMt - 1) / BMt);
    const int q1 = total1 >> 3, r1 = total1 & 7;
    range_lo1 = xcd < r1 ? xcd * (q1 + 1) : r1 * (q1 + 1) + (xcd - r1) * q1;
    range_len1 = xcd < r1 ? q1 + 1 : q1;
    n_first = slot < range_len ? (range_len - slot + per_xcd_blocks - 1) / per_xcd_blocks : 0;
  }
  const int span = GRP ? range_len + range_len1 : range_len;
  const int my_tiles = slot < span ? (span - slot + per_xcd_blocks - 1) / per_xcd_blocks : 0;
  if (my_tiles == 0) return;

  int nk = K / BK / ksplit;   // K-steps per (virtual) tile
  const int nk1 = GRP ? p1.K / BK : 0;
  // Tile order inside the n-fastest ranges above would sweep ALL of W (3.5 MB at N = 2304, K = 768) with every round of an XCD's
  // 32 workgroups: more than its 4 MiB L2 holds next to the X tiles and the outputs passing through, so W came back from the
  // Infinity Cache once per round (round 1: 4.4x read amplification on the QKV / c_fc shapes).  Order: bands of `band` m-tiles
  // (one band ~ one XCD's share), inside a band groups of G n-panels, inside a group m-tile-major: an XCD's concurrent tiles share
  // G W panels (<= 1.6 MB) and its band's X tiles (~2.5 MB), and the next group re-reads only the X tiles, still in L2.
  // (ordG = n-panels per group, chosen by the host: wide_order_group(); 0 = the plain n-fastest order, for A/B runs)
  const int band = (tiles_m + 7) >> 3;
  auto tile_coords = [&](int logical, int& tm, int& tn) {
    if (ordG == 0) { tm = logical / tiles_n; tn = logical - tm * tiles_n; return; }
    if (ordG >= 100) {
      // n-BLOCKED order (round 4): blocks of ordG - 100 n-panels outermost, inside a block n-fastest over ALL m-tiles.  The XCDs'
      // contiguous eighths of this order are (m-range x panel-block) rectangles: an XCD streams only its block of W (<= ~2 MB: it
      // stays in the 4 MiB L2 across the XCD's rounds) and its m-range of X, which one or two other XCDs read as well - instead of
      // ALL of W once per round (profiles/r04_c_gemm_read_traffic_split.txt: 27 W-sized fetches per QKV launch, 35 per c_fc launch).
      const int nbw = ordG - 100;
      const int bt = tiles_m * nbw;
      const int nb = logical / bt;
      const int rem = logical - nb * bt;
      const int wdt = tiles_n - nb * nbw < nbw ? tiles_n - nb * nbw : nbw;   // the last block may be narrower
      const int mi = rem / wdt;
      tm = mi;
      tn = nb * nbw + (rem - mi * wdt);
      return;
    }
    const int b = logical / (band * tiles_n);
    const int blen = tiles_m - b * band < band ? tiles_m - b * band : band;   // the last band may be shorter
    const int rem = logical - b * band * tiles_n;
    const int g = rem / (blen * ordG);
    const int glen = tiles_n - g * ordG < ordG ? tiles_n - g * ordG : ordG;   // the last group may be narrower
    const int rem2 = rem - g * blen * ordG;
    const int mi = rem2 / glen;
    tm = b * band + mi;
    tn = g * ordG + (rem2 - mi * glen);
  };
  const uint32_t row_stride = static_cast<uint32_t>(K) * ELT;
  // (virtual) tile ti of this workgroup -> its split and its logical index in the n-fastest order of its problem
  auto tile_logical = [&](int ti, int& split) {
    if constexpr (GRP) {
      split = 0;
      const int j = slot + ti * per_xcd_blocks;
      return ti < n_first ? range_lo + j : range_lo1 + (j - range_len);
    } else {
      const int virt = range_lo + slot + ti * per_xcd_blocks;
      split = virt / base_total;
      return virt - split * base_total;
    }
  };

  // ---- issue side.  DMA source = uniform tile base (SGPRs) + 32-bit per-lane offset (one VGPR per piece):
  // lane i of a 1-KiB piece fills LDS (row 8*piece + i/8, physical chunk i%8) and fetches logical chunk (i%8)^(row&7).
  uint32_t offW[4], offX[3];
  // TN: k-row of the lane inside a stage and its (swizzled) 16-byte chunk offset inside the tile's row, per piece
  uint32_t krW[4], krX[2];
  const uint32_t ldw_b = static_cast<uint32_t>(N) * 2, ldx_b = static_cast<uint32_t>(M) * 2;   // TN: bytes per k-row of Wk / Xk
  auto tn_f = [](int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); };                       // chunk XOR of k-row kr (cdna guide T10 (b))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (TN) {
      const int kr = (wid * 4 + i) * 2 + (lane >> 5), pch = lane & 31;       // a 1-KiB piece = two 512-byte rows
      krW[i] = kr;
      offW[i] = static_cast<uint32_t>((pch & 16) | ((pch & 15) ^ tn_f(kr))) << 4;
    } else {
      const int row = (wid * 4 + i) * 8 + sub;
      offW[i] = static_cast<uint32_t>(row) * row_stride + (((lane & 7) ^ (row & 7)) << 4);
    }
  }
  if constexpr (TN) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = (wid + 8 * i) * 4 + (lane >> 4), pch = lane & 15;       // a piece = four 256-byte rows
      krX[i] = kr;
      offX[i] = static_cast<uint32_t>(pch ^ tn_f(kr)) << 4;
    }
  }
  int issue_krow0 = 0;   // TN: first k-row of the (virtual) tile being staged
  const char* Wt = W;   // W + n0*row_stride (+ the split's K offset) of the tile being staged
  const char* Xt = X;   // X (+ the split's K offset)
  // GRP: the issue side's view of the problem it is staging (the kernel arguments until its first tile of the second problem)
  bool i_second = false;
  int i_nk = nk, i_M = M;
  uint32_t i_row_stride = row_stride;
  auto set_issue_tile = [&](int ti) {
    if constexpr (GRP) {
      int split;
      const int logical = tile_logical(ti, split);
      if (ti >= n_first && !i_second) {      // once per workgroup: the W-piece offsets follow the second problem's row stride
        i_second = true;
        i_nk = nk1;
        i_M = M1;
        i_row_stride = static_cast<uint32_t>(p1.K) * ELT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (wid * 4 + i) * 8 + sub;
          offW[i] = static_cast<uint32_t>(row) * i_row_stride + (((lane & 7) ^ (row & 7)) << 4);
        }
      }
      const int tn_i = i_second ? tiles_n1 : N_a / wBN;
      const int tm = logical / tn_i, tn = logical - tm * tn_i;
      const int m0 = tm * BMt, n0 = tn * wBN;
      Wt = (i_second ? p1.W : W) + static_cast<size_t>(n0) * i_row_stride;
      Xt = i_second ? p1.X : X;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int row = (i == 1 ? xpiece1 : wid + 8 * i) * 8 + sub;
        int xr = m0 + row;
        xr = xr < i_M ? xr : i_M - 1;
        offX[i] = static_cast<uint32_t>(xr) * i_row_stride + (((lane & 7) ^ (row & 7)) << 4);
      }
      return;
    }
    const int virt = range_lo + slot + ti * per_xcd_blocks;
    const int split = virt / base_total, logical = virt - split * base_total;
    int tm, tn;
    tile_coords(logical, tm, tn);
    const int m0 = tm * BMt, n0 = tn * wBN;
    if constexpr (TN) {
      Wt = W + static_cast<size_t>(n0) * 2;      // column origins; the k-rows are added per piece
      Xt = X + static_cast<size_t>(m0) * 2;
      issue_krow0 = split * nk * 64;
      return;
    }
    const size_t kbase = static_cast<size_t>(split) * nk * wRowBytes;
    Wt = W + static_cast<size_t>(n0) * row_stride + kbase;
    Xt = X + kbase;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = (i == 1 ? xpiece1 : wid + 8 * i) * 8 + sub;
      int xr = m0 + row;
      xr = xr < M ? xr : M - 1;   // rows past M are computed on duplicated data and never stored
      offX[i] = static_cast<uint32_t>(xr) * row_stride + (((lane & 7) ^ (row & 7)) << 4);   // < 4 GiB: M*K*ELT checked on host
    }
  };
  int issue_kt = 0, issue_tile = 0, issue_buf = 0;
  auto issue_piece = [&](int p) {   // p is a compile-time constant at every call site
    char* base = lds + issue_buf * STG;
    if constexpr (TN) {
      const int kr0 = issue_krow0 + issue_kt * 64;
      if (p < 4) {          // Wk rows: past the end they repeat the last row (finite, and multiplied by the zeros below)
        int kr = kr0 + static_cast<int>(krW[p]);
        kr = kr < sc.kreal ? kr : sc.kreal - 1;
        __builtin_amdgcn_global_load_lds((w_gptr_t)(Wt + (static_cast<uint32_t>(kr) * ldw_b + offW[p])),
                                         (w_lptr_t)(base + (wid * 4 + p) * 1024), 16, 0, 0);
      } else if (p < 6) {   // Xk rows: past the end they are zeros
        const int kr = kr0 + static_cast<int>(krX[p - 4]);
        const char* src = kr < sc.kreal ? Xt + (static_cast<uint32_t>(kr) * ldx_b + offX[p - 4])
                                        : reinterpret_cast<const char*>(g_wide_zero) + offX[p - 4];
        __builtin_amdgcn_global_load_lds((w_gptr_t)src, (w_lptr_t)(base + wWBytes + (wid + 8 * (p - 4)) * 1024), 16, 0, 0);
      }
      return;
    }
    const size_t koff = static_cast<size_t>(issue_kt) * wRowBytes;
    if (p < 4)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(Wt + koff + offW[p]), (w_lptr_t)(base + (wid * 4 + p) * 1024), 16, 0, W_W_CPOL);
    else if (p < 6)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(Xt + koff + offX[p - 4]),
                                       (w_lptr_t)(base + wWBytes + (p == 4 ? wid : xpiece1) * 1024), 16, 0, W_X_CPOL);
    else if (three)
      __builtin_amdgcn_global_load_lds((w_gptr_t)(Xt + koff + offX[2]), (w_lptr_t)(base + wWBytes + (wid + 16) * 1024), 16, 0, W_X_CPOL);
  };
  // The K loop below is ONE straight-line steady state: it issues a stage in every K-step.  The three stages issued past
  // the workgroup's last one re-stage its last tile into buffers nobody reads any more (drained before the kernel ends).
  auto issue_done = [&]() {
    issue_buf = issue_buf == 2 ? 0 : issue_buf + 1;
    if (++issue_kt == (GRP ? i_nk : nk)) {
      issue_kt = 0;
      if (++issue_tile < my_tiles) set_issue_tile(issue_tile);
    }
  };
  auto issue_stage = [&]() {
#pragma unroll
    for (int p = 0; p < 7; ++p) issue_piece(p);
    issue_done();
  };

  // ---- fragment reads: inline asm (their waits are ours, see W_WAIT_FRAGS) with immediate offsets: the 16-row fragment
  // tiles of one operand sit 2048 bytes apart, the second 32-deep half is the first one's address XOR 64.
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((w_lptr_t)lds));
  const uint32_t aW = lds_base + w_swz(wn * 64 + frow, fq);
  const uint32_t aX = lds_base + wWBytes + w_swz(wm * WR + frow, fq);
  // TN: transposing reads.  In its 16-lane group (fq) lane 4q+p hands in the address of k-row q, columns 4p..4p+3 of a 4 x 16 block
  // and receives the four k of column `frow`: block (tile t, half jb) = k-rows 8 fq + 4 jb .. +3 (+32 for the second half-step)
  // x the tile's 16 columns.  [tile][jb] byte addresses inside a stage; the n-tiles of a wave are 32 bytes apart in a row.
  uint32_t tW[4][2], tX[MF][2];
  if constexpr (TN) {
    const int q = frow >> 2, pp = frow & 3;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const int kr = 8 * fq + 4 * jb + q;
      const int f = (q << 2) | ((2 * fq + jb) & 3);      // = tn_f(kr)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int chunk = wn * 8 + a * 2 + (pp >> 1);
        tW[a][jb] = lds_base + kr * 512 + (((chunk & 16) | ((chunk & 15) ^ f)) << 4) + 8 * (pp & 1);
      }
#pragma unroll
      for (int b = 0; b < MF; ++b) {
        const int chunk = wm * (WR / 8) + b * 2 + (pp >> 1);
        tX[b][jb] = lds_base + wWBytes + kr * 256 + ((chunk ^ f) << 4) + 8 * (pp & 1);
      }
    }
  }
#ifdef W_STAMPS   // timing build only (tools/wide_stamps.py): per-wave cycle sums of the K loop's segments
#define W_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_sum[k] += static_cast<unsigned>(t_ - st_last); st_last = t_; } while (0)
#else
#define W_STAMP(k) do { } while (0)
#endif
#define W_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  auto load_frags = [&](w_u32x4_t (&fw)[4], w_u32x4_t (&fx)[MF], int buf, int ks) {
    const uint32_t bo = static_cast<uint32_t>(buf) * STG;
    const uint32_t w = (aW + bo) ^ (ks ? 64u : 0u), x = (aX + bo) ^ (ks ? 64u : 0u);
    W_READ(fw[0], w, 0); W_READ(fw[1], w, 2048); W_READ(fw[2], w, 4096); W_READ(fw[3], w, 6144);
    W_READ(fx[0], x, 0); W_READ(fx[1], x, 2048); W_READ(fx[2], x, 4096);
    if constexpr (MF >= 4) W_READ(fx[3], x, 6144);
    if constexpr (MF == 5) W_READ(fx[4], x, 8192);
  };
#define W_WAIT_FRAGS(cnt, fw, fx)                                                                                \
  do {                                                                                                           \
    if constexpr (MF == 5)                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                 \
                   : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]),  \
                     "+v"(fx[3]), "+v"(fx[MF - 1])::"memory");                                                   \
    else if constexpr (MF == 4)                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                 \
                   : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]),  \
                     "+v"(fx[3])::"memory");                                                                     \
    else                                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                 \
                   : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2])   \
                   ::"memory");                                                                                  \
  } while (0)

  // ---- deferred stores of the previous tile (bf16 outputs only) --------------------------------------------
  constexpr bool DEFER = true;
  w_u32x4_t pend[DGE ? 1 : NPEND];
  bool pend_valid = false;
  // Deferred QuickGELU (DGE, round 4).  Measured from outside (tools/gemm_tile_cost.py, profiles/r04_o_gemm_tile_cost.txt): a K-step of
  // the 160-row tile costs 0.96 us and a tile switch 2.2 us - but 5.0 us when the epilogue carries QuickGELU: 80 accumulators per lane
  // through v_exp_f32 + v_rcp_f32 (quarter rate) while the matrix pipe idles, a third of a c_fc tile at K = 768.  Here the tile's
  // pre-activations (accumulator + bias, f32) move to 64 registers of their own instead of 32 packed ones, the next tile's K loop
  // starts at once, and each of its first 8 K-steps activates, packs and stores one 16-byte piece per lane in the issue slots the MFMAs
  // leave free.  Same arithmetic per element, same bits.  Only the 128-row tile has the registers (64 + 64 + 64 fragments).
  w_f32x4_t pendf[DGE ? 4 : 1][DGE ? MF : 1];
  char* pend_ptr = nullptr;            // &out[(m0 + wm*WR + frow) * N + col] of the pending tile
  constexpr int OEL = OUT8 ? 1 : 2;    // bytes per output element of the packed store paths
  size_t row16 = static_cast<size_t>(16) * N * OEL;   // bytes between the 16-row fragments of a wave
  size_t pend_row16 = row16;                          // GRP: the same of the PENDING tile (it may belong to the first problem still)
  int sps = (NDEFER + nk - 1) / nk;   // stores per K-step so that all of them leave within one tile's K loop
  auto store_pending = [&](int idx) {
    if constexpr (DGE) {
      // piece idx = (m-fragment b, n-tile pair pr): its 8 pre-activations -> QuickGELU -> two packed words per n-tile -> the lane-pair
      // exchange of the epilogue below -> one 16-byte store
      auto gelu2 = [](float x, float y) {
        const w_f32x2_t v = {x, y};
        const w_f32x2_t t = v * w_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
        const w_f32x2_t d = w_f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + w_f32x2_t{1.0f, 1.0f};
        return v * w_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
      };
      // (the pre-activations are read where they lie - a copy per piece would cost 8 more registers next to 192 that cannot move)
      auto piece = [&](const w_f32x4_t& va, const w_f32x4_t& vb) __attribute__((always_inline)) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const w_f32x2_t ga = gelu2(va[2 * w], va[2 * w + 1]);
          lo[w] = (epi & EPI_OUT_F16) ? pack_f16x2(ga[0], ga[1]) : pack_bf16x2(ga[0], ga[1]);
          const w_f32x2_t gb = gelu2(vb[2 * w], vb[2 * w + 1]);
          hi[w] = (epi & EPI_OUT_F16) ? pack_f16x2(gb[0], gb[1]) : pack_bf16x2(gb[0], gb[1]);
        }
        const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
        const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
        w_store16(pend_ptr + (idx / 2) * (GRP ? pend_row16 : row16) + (idx % 2) * 64, w_u32x4_t{s0[0], s1[0], s0[1], s1[1]});
      };
      switch (idx) {   // compile-time register choice per case
#define W_PF(j) case j: piece(pendf[2 * (j % 2)][(j / 2) < MF ? (j / 2) : 0], pendf[2 * (j % 2) + 1][(j / 2) < MF ? (j / 2) : 0]); break;
        W_PF(0) W_PF(1) W_PF(2) W_PF(3) W_PF(4) W_PF(5) W_PF(6)
#undef W_PF
        default: piece(pendf[2][MF - 1], pendf[3][MF - 1]); break;
      }
    } else {
    switch (idx) {   // compile-time register choice per case: no dynamically indexed vector arrays (they would go to scratch)
#define W_ST(j) case j: if constexpr (j < NPEND) w_store16(pend_ptr + (OUT8 ? j : j / 2) * (GRP ? pend_row16 : row16) + (OUT8 ? 0 : (j % 2) * 64), pend[j < NPEND ? j : 0]); break;
      W_ST(0) W_ST(1) W_ST(2) W_ST(3) W_ST(4) W_ST(5) W_ST(6) W_ST(7) W_ST(8) W_ST(9)
#undef W_ST
      default: break;
    }
    }
  };

  // ---- rotated, software-pipelined K loop --------------------------------------------------------------------
  // Per K-step s (stage s in LDS buffer s%3), with F0 = fragments of (s, k 0..31) already in registers:
  //   1. issue the ds_reads of F1 = (s, k 32..63)
  //   2. 20 MFMAs on F0 (+ up to `sps` deferred stores of the previous tile)
  //   3. lgkmcnt(0) (own F1 reads done), vmcnt: stage s+1 landed (stage s+2 and this step's stores may stay in flight),
  //      s_barrier -> every wave has finished reading buffer s%3 and sees stage s+1
  //   4.-6. LDS-DMA of stage s+3 into buffer s%3 (two full K-steps of flight), the ds_reads of F0' = (s+1, k 0..31) and
  //      the 20 MFMAs on F1, interleaved: one DMA piece / one read per 3 MFMAs.  Issued as one burst, the 52 pieces of a
  //      stage queue up in the CU's address path and every wave sits in a VMEM issue stall while the matrix pipe idles.
  // Anti-phase issue.  Waves w and w+4 share a SIMD and, released by the same barrier, would run the same instructions
  // at the same time: both stuck in the ~100-cycle issue of an LDS-DMA piece (the CU's address path moves 64 B/clk) with
  // the matrix pipe idle.  So group A (waves 0..3) issues its pieces of stage s+3 among the MFMAs of the SECOND half of
  // K-step s and group B (waves 4..7) among those of the FIRST half of K-step s+1 (and its deferred stores in the other
  // half): while one wave of a SIMD waits on the address path its partner streams MFMAs.
  const bool group_b = wid >= 4;
  // Short K: the first tile's accumulators start as the residual tile (bias commutes, activations would not): its 20 loads
  // per lane are older than every LDS-DMA, so the prologue's counted wait covers them and their HBM latency hides behind
  // the prologue instead of sitting exposed in the epilogue (residual GEMMs have <= 1 tile per workgroup: N = 512 / 768).
  // Measured inside the encoder: K = 512 / 768: -2.9 / -1.6 us per launch; K = 2048 / 3072: +0.3 / +2.3 us (the 40 MB burst
  // delays the first stage and a long K loop has no trouble hiding the epilogue's loads behind other workgroups) -> nk <= 16.
  // (Round 3 tried the other order - stage 0's DMA first, the residual rows behind it as inline-asm loads under the same counted
  // wait, added after the other prologue stages were issued: 0.5 % slower on the bench line, A/B/A/B on one box; not kept.)
  // (fp8 launches always carry EPI_SCALE: never there - said at compile time, so that the prologue holds no residual code)
  // GRP: the compute side moves on to the second problem (once per workgroup: before its first tile if it has none of the first
  // problem's, else at the top of tile n_first)
  [[maybe_unused]] auto to_problem1 = [&]() {
    M = M1; N = p1.N; bias = p1.bias; residual = p1.residual; out = p1.out;
    sc.colscale = p1.colscale; sc.alpha = p1.alpha; sc.oscale = p1.oscale;
    nk = nk1; tiles_n = tiles_n1;
    row16 = static_cast<size_t>(16) * N * OEL;
    sps = (NDEFER + nk - 1) / nk;
  };
  if constexpr (GRP) { if (n_first == 0) to_problem1(); }
  // Round 4: the rule holds for EVERY tile of such a GEMM, not only a workgroup's first one - (residual + sum) + bias and (sum + bias)
  // + residual round differently, and which tiles come first in a workgroup depends on the tile height, the CU count and, in a grouped
  // launch, on the other problem; results must depend on none of them.  A later tile's residual loads are issued at the top of the
  // tile: younger than the stages in flight, they wait for those, which after an epilogue have long landed.
  const bool res_cond = !FP8 && (epi & EPI_RESIDUAL) && !(epi & (EPI_QUICKGELU | EPI_GELU | EPI_RELU | EPI_SCALE));
  auto res_first = [&]() { return res_cond && nk <= 16; };   // (nk: of the problem the compute side is in)
  auto mfma = [&](const w_u32x4_t& fw, const w_u32x4_t& fx, w_f32x4_t& c) {
    if constexpr (FP8) {
      (void)fw; (void)fx; (void)c;   // the fp8 K-step multiplies whole 128-k rows: mfma8 below
    } else if constexpr (F32) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[s]), __uint_as_float(fx[s]), c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w_bf16x8_t, fw), __builtin_bit_cast(w_bf16x8_t, fx), c,
                                                  0, 0, 0);
    }
  };

  // fp8: the two 16-byte halves of a lane's 32 k (chunks fq and fq+4 of the 128-byte row, the same for both operands, so the
  // pairing inside the dot product is consistent whatever order the instruction gives its k) form one 8-register operand
  typedef __attribute__((ext_vector_type(8))) int w_i32x8_t;
  auto mfma8 = [&](const w_u32x4_t& w0, const w_u32x4_t& w1, const w_u32x4_t& x0, const w_u32x4_t& x1, w_f32x4_t& c) {
    w_i32x8_t a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = static_cast<int>(w0[j]); a[4 + j] = static_cast<int>(w1[j]);
      b[j] = static_cast<int>(x0[j]); b[4 + j] = static_cast<int>(x1[j]);
    }
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // e4m3 x e4m3, block scales 2^0
  };
#ifdef W_STAMPS
  unsigned st_sum[4] = {0, 0, 0, 0};
  unsigned long long st_last = __builtin_readcyclecounter();
#endif
  auto run = [&](auto gb) {
  constexpr bool GB = decltype(gb)::value;
  w_f32x4_t acc[4][MF];   // [n-tile][m-tile]
  // acc += residual tile at (m0, n0): f32 rows, or (EPI_RES_F16) fp16 rows read in the 16-byte layout of the packed output
  // (lane = one row x 8 consecutive n) and brought back to the accumulator layout by the same v_permlane16_swap.
  // fp16 residual rows in the 16-byte layout of the packed output (lane = one row x 8 consecutive n), brought back to the accumulator
  // layout by the same v_permlane16_swap
  auto load_res16 = [&](int m0, int n0, w_u32x4_t (&r)[NPEND]) __attribute__((always_inline)) {
    const uint16_t* res16 = reinterpret_cast<const uint16_t*>(residual);
    const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;    // + 32*pair
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      int m = m0 + wm * WR + b * 16 + frow;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
        if constexpr (!OUT8) r[b * 2 + pr] = *reinterpret_cast<const w_u32x4_t*>(res16 + static_cast<size_t>(m) * N + col + 32 * pr);
    }
  };
  auto apply_res16 = [&](const w_u32x4_t (&r)[NPEND]) __attribute__((always_inline)) {
    if constexpr (!OUT8) {
#pragma unroll
      for (int b = 0; b < MF; ++b) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const w_u32x4_t q = r[b * 2 + pr];
          const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);   // -> words 0 of tiles 2pr, 2pr+1
          const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);   // -> words 1
          acc[2 * pr][b][0] += f16lo_to_f32(s0[0]); acc[2 * pr][b][1] += f16hi_to_f32(s0[0]);
          acc[2 * pr][b][2] += f16lo_to_f32(s1[0]); acc[2 * pr][b][3] += f16hi_to_f32(s1[0]);
          acc[2 * pr + 1][b][0] += f16lo_to_f32(s0[1]); acc[2 * pr + 1][b][1] += f16hi_to_f32(s0[1]);
          acc[2 * pr + 1][b][2] += f16lo_to_f32(s1[1]); acc[2 * pr + 1][b][3] += f16hi_to_f32(s1[1]);
        }
      }
    }
  };
  auto add_residual = [&](int m0, int n0) __attribute__((always_inline)) {
    if constexpr (FP8 && MF == 5) {
      // fp8, 160 rows (fp16 residual stream only): two groups of row fragments, 24 + 16 registers in flight instead of 40 - still
      // every load before any store.  With all 40 (and the f32 path's 80) the allocator spills long-lived values whose reloads land
      // inside the K loop (tools/asm_loop_scratch.py)
      const uint16_t* res16 = reinterpret_cast<const uint16_t*>(residual);
      const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;    // + 32*pair
      constexpr int BG = 3;
#pragma unroll
      for (int bg = 0; bg < MF; bg += BG) {
        w_u32x4_t r[2 * BG];
#pragma unroll
        for (int b = bg; b < bg + BG && b < MF; ++b) {
          int m = m0 + wm * WR + b * 16 + frow;
          m = m < M ? m : M - 1;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            r[(b - bg) * 2 + pr] = *reinterpret_cast<const w_u32x4_t*>(res16 + static_cast<size_t>(m) * N + col + 32 * pr);
        }
#pragma unroll
        for (int b = bg; b < bg + BG && b < MF; ++b) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const w_u32x4_t q = r[(b - bg) * 2 + pr];
            const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
            const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
            acc[2 * pr][b][0] += f16lo_to_f32(s0[0]); acc[2 * pr][b][1] += f16hi_to_f32(s0[0]);
            acc[2 * pr][b][2] += f16lo_to_f32(s1[0]); acc[2 * pr][b][3] += f16hi_to_f32(s1[0]);
            acc[2 * pr + 1][b][0] += f16lo_to_f32(s0[1]); acc[2 * pr + 1][b][1] += f16hi_to_f32(s0[1]);
            acc[2 * pr + 1][b][2] += f16lo_to_f32(s1[1]); acc[2 * pr + 1][b][3] += f16hi_to_f32(s1[1]);
          }
        }
      }
    } else
    if (epi & EPI_RES_F16) {
      w_u32x4_t r[NPEND];
      load_res16(m0, n0, r);
      apply_res16(r);
    } else if constexpr (!FP8) {   // (the fp8 mode's residual stream is fp16)
      w_f32x4_t rv[4][MF];
#pragma unroll
      for (int b = 0; b < MF; ++b) {
        int m = m0 + wm * WR + b * 16 + frow;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          rv[a][b] = *reinterpret_cast<const w_f32x4_t*>(residual + static_cast<size_t>(m) * N + n0 + wn * 64 + a * 16 + fq * 4);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] += rv[a][b];
    }
  };
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = w_f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (res_first()) {
    int tm, tn;
    if constexpr (GRP) { int sp_; tile_coords(tile_logical(0, sp_), tm, tn); }
    else tile_coords(range_lo + slot, tm, tn);
    add_residual(tm * BMt, tn * wBN);
  }
  set_issue_tile(0);
  issue_stage();
  issue_stage();
  if constexpr (!GB) {
    issue_stage();
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // stage 0 landed (two younger stages of >= 6 pieces may fly)
  } else {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // group B issues stage 2 in the first half of K-step 0
  }
  __builtin_amdgcn_s_barrier();
  W_TL(1);   // first stage landed

  int cur = 0;
  int ns = 0;                   // deferred stores issued since the last counted wait
  w_u32x4_t f0w[4], f0x[MF], f1w[4], f1x[MF];
  // every fragment register of the wave as an operand of a counted wait (fp8 K-step): nothing that reads them moves above it
#define W_WAIT_ALL(cnt)                                                                                                   \
  do {                                                                                                                    \
    if constexpr (MF == 5)                                                                                                \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                          \
                   : "+v"(f0w[0]), "+v"(f0w[1]), "+v"(f0w[2]), "+v"(f0w[3]), "+v"(f1w[0]), "+v"(f1w[1]), "+v"(f1w[2]),    \
                     "+v"(f1w[3]), "+v"(f0x[0]), "+v"(f0x[1]), "+v"(f0x[2]), "+v"(f0x[MF > 3 ? 3 : 0]),                   \
                     "+v"(f0x[MF > 4 ? 4 : 0]), "+v"(f1x[0]), "+v"(f1x[1]), "+v"(f1x[2]), "+v"(f1x[MF > 3 ? 3 : 0]),      \
                     "+v"(f1x[MF > 4 ? 4 : 0])::"memory");                                                                \
    else if constexpr (MF == 4)                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                          \
                   : "+v"(f0w[0]), "+v"(f0w[1]), "+v"(f0w[2]), "+v"(f0w[3]), "+v"(f1w[0]), "+v"(f1w[1]), "+v"(f1w[2]),    \
                     "+v"(f1w[3]), "+v"(f0x[0]), "+v"(f0x[1]), "+v"(f0x[2]), "+v"(f0x[MF > 3 ? 3 : 0]), "+v"(f1x[0]),     \
                     "+v"(f1x[1]), "+v"(f1x[2]), "+v"(f1x[MF > 3 ? 3 : 0])::"memory");                                    \
    else                                                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                          \
                   : "+v"(f0w[0]), "+v"(f0w[1]), "+v"(f0w[2]), "+v"(f0w[3]), "+v"(f1w[0]), "+v"(f1w[1]), "+v"(f1w[2]),    \
                     "+v"(f1w[3]), "+v"(f0x[0]), "+v"(f0x[1]), "+v"(f0x[2]), "+v"(f1x[0]), "+v"(f1x[1]), "+v"(f1x[2])     \
                   ::"memory");                                                                                           \
  } while (0)
#define W_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  // TN fragments as pairs of 8-byte halves (jb = 0: k 0..3, jb = 1: k 4..7 of the lane's eight)
  w_u32x2_t h0w[TN ? 4 : 1][2], h0x[TN ? MF : 1][2], h1w[TN ? 4 : 1][2], h1x[TN ? MF : 1][2];
#define W_WAIT_H(cnt, hw, hx)                                                                                            \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                             \
               : "+v"(hw[0][0]), "+v"(hw[0][1]), "+v"(hw[TN ? 1 : 0][0]), "+v"(hw[TN ? 1 : 0][1]), "+v"(hw[TN ? 2 : 0][0]), \
                 "+v"(hw[TN ? 2 : 0][1]), "+v"(hw[TN ? 3 : 0][0]), "+v"(hw[TN ? 3 : 0][1]), "+v"(hx[0][0]), "+v"(hx[0][1]), \
                 "+v"(hx[TN ? 1 : 0][0]), "+v"(hx[TN ? 1 : 0][1]), "+v"(hx[TN ? 2 : 0][0]), "+v"(hx[TN ? 2 : 0][1]),        \
                 "+v"(hx[TN ? 3 : 0][0]), "+v"(hx[TN ? 3 : 0][1])::"memory")
  auto mfma_h = [&](const w_u32x2_t (&w2)[2], const w_u32x2_t (&x2)[2], w_f32x4_t& c) {
    const w_u32x4_t fw = {w2[0][0], w2[0][1], w2[1][0], w2[1][1]}, fx = {x2[0][0], x2[0][1], x2[1][0], x2[1][1]};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(w_bf16x8_t, fw), __builtin_bit_cast(w_bf16x8_t, fx), c, 0, 0, 0);
  };
  // TN: column sums of the Xk operand (= wgrad's bias gradient) ride along as one more MFMA per half-step against an all-ones
  // operand: the four waves that share Xk fragments take one m-tile each (b = wn), in tiles of the first n-panel only
  w_f32x4_t csacc = {0.f, 0.f, 0.f, 0.f};
  const w_u32x2_t ones2[2] = {{0x3f803f80u, 0x3f803f80u}, {0x3f803f80u, 0x3f803f80u}};
  bool do_cs = false;
  auto cs_mfma = [&](const w_u32x2_t (&hx)[TN ? MF : 1][2]) __attribute__((always_inline)) {
    if constexpr (TN) {
      switch (wn) {     // wave-uniform; a runtime index would push the fragments to scratch
        case 0: mfma_h(ones2, hx[0], csacc); break;
        case 1: mfma_h(ones2, hx[1], csacc); break;
        case 2: mfma_h(ones2, hx[2], csacc); break;
        default: mfma_h(ones2, hx[3], csacc); break;
      }
    }
  };
  if constexpr (TN) {
#pragma unroll
    for (int a = 0; a < 4; ++a) { W_TR(h0w[a][0], tW[a][0], 0); W_TR(h0w[a][1], tW[a][1], 0); }
#pragma unroll
    for (int b = 0; b < MF; ++b) { W_TR(h0x[b][0], tX[b][0], 0); W_TR(h0x[b][1], tX[b][1], 0); }
  } else if constexpr (FP8) {
    // fp8 K-step s starts with BOTH halves of W tiles 0,1 and of every X tile of stage s in (or on their way to) registers
    const uint32_t w0 = aW, w1 = aW ^ 64u, x0 = aX, x1 = aX ^ 64u;
    W_READ(f0w[0], w0, 0); W_READ(f1w[0], w1, 0); W_READ(f0w[1], w0, 2048); W_READ(f1w[1], w1, 2048);
    W_READ(f0x[0], x0, 0); W_READ(f1x[0], x1, 0); W_READ(f0x[1], x0, 2048); W_READ(f1x[1], x1, 2048);
    W_READ(f0x[2], x0, 4096); W_READ(f1x[2], x1, 4096);
    if constexpr (MF >= 4) { W_READ(f0x[MF >= 4 ? 3 : 0], x0, 6144); W_READ(f1x[MF >= 4 ? 3 : 0], x1, 6144); }
    if constexpr (MF == 5) { W_READ(f0x[MF - 1], x0, 8192); W_READ(f1x[MF - 1], x1, 8192); }
  } else {
    load_frags(f0w, f0x, 0, 0);   // from here on F0 of K-step s+1 (also across tiles) is fetched in the second half of s
  }
  for (int ti = 0; ti < my_tiles; ++ti) {
    if (ti > 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] = w_f32x4_t{0.f, 0.f, 0.f, 0.f};
      if constexpr (GRP) { if (ti == n_first) to_problem1(); }
      if (res_first()) {
        int sp_, tm_r, tn_r;
        tile_coords(tile_logical(ti, sp_), tm_r, tn_r);
        add_residual(tm_r * BMt, tn_r * wBN);
      }
    }
    if constexpr (TN) {
      if (sc.colsum) {
        const int virt_c = range_lo + slot + ti * per_xcd_blocks;
        int tm_c, tn_c;
        tile_coords(virt_c - (virt_c / base_total) * base_total, tm_c, tn_c);
        do_cs = tn_c == 0;
        csacc = w_f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }

    // DGE: the bf16 K-step below with piece J of the previous tile's deferred QuickGELU BETWEEN this wave's own MFMAs (J < 0: none).
    // In-order issue is what bounds a K-step (a wave issues in ~30 % of its cycles): the same 36 VALU instructions placed in front of
    // the MFMA burst cost the K-step what they cost the epilogue (tile switch 3.4 us either way, tools/gemm_tile_cost.py), while one
    // v_exp / v_mul behind every MFMA of EVERY K-step costs 5 % of a K-step (diagnostic build) - so: one 2-value step of
    // the activation behind each MFMA of the half in which this wave issues no LDS-DMA, four MFMAs per pair of values, the lane-pair
    // exchange and the 16-byte store after the sixteenth.  The piece is a compile-time index: its registers are named, not indexed.
    [[maybe_unused]] auto kstep_dge = [&](auto jc) __attribute__((always_inline)) {
      if constexpr (DGE) {
        constexpr int J = decltype(jc)::value;
        constexpr int PB = J >= 0 ? J / 2 : 0, PR = J >= 0 ? J % 2 : 0;
        w_f32x2_t gx[4];
        uint32_t gword[4];
        auto micro = [&](int i) __attribute__((always_inline)) {
          if constexpr (J >= 0) {
            const int pq = i >> 2, ph = i & 3;                      // pair 0: tile 2PR values 0,1; 1: tile 2PR+1 values 0,1; 2, 3: values 2,3
            const w_f32x4_t& src = (pq & 1) ? pendf[2 * PR + 1][PB] : pendf[2 * PR][PB];
            const w_f32x2_t v = {src[2 * (pq >> 1)], src[2 * (pq >> 1) + 1]};
            if (ph == 0) {
              gx[pq] = v * w_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
            } else if (ph == 1) {
              gx[pq] = w_f32x2_t{__builtin_amdgcn_exp2f(gx[pq][0]), __builtin_amdgcn_exp2f(gx[pq][1])};
            } else if (ph == 2) {
              const w_f32x2_t d = gx[pq] + w_f32x2_t{1.0f, 1.0f};
              gx[pq] = w_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            } else {
              const w_f32x2_t o = v * gx[pq];
              gword[pq] = (epi & EPI_OUT_F16) ? pack_f16x2(o[0], o[1]) : pack_bf16x2(o[0], o[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        auto piece_out = [&]() __attribute__((always_inline)) {
          if constexpr (J >= 0) {
            const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(gword[0], gword[1], false, false);
            const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(gword[2], gword[3], false, false);
            w_store16(pend_ptr + PB * (GRP ? pend_row16 : row16) + PR * 64, w_u32x4_t{s0[0], s1[0], s0[1], s1[1]});
            ++ns;
          }
        };
        load_frags(f1w, f1x, cur, 1);
        W_WAIT_FRAGS(8, f0w, f0x);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(GB ? 0 : 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if constexpr (GB) { if (i % PSTEP == 0) issue_piece(i / PSTEP); }
          mfma(f0w[i / MF], f0x[i % MF], acc[i / MF][i % MF]);
          if constexpr (!GB) micro(i);
          if constexpr (GB) { if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0); }
        }
        if constexpr (GB) issue_done();
        if constexpr (!GB) piece_out();
        __builtin_amdgcn_sched_barrier(0);
        W_WAIT_FRAGS(0, f1w, f1x);
        if (ns == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ns == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        ns = 0;
        __builtin_amdgcn_s_barrier();
        const int nxt = cur == 2 ? 0 : cur + 1;
        {
          const uint32_t bo = static_cast<uint32_t>(nxt) * STG;
          const uint32_t nW = aW + bo, nX = aX + bo;
          __builtin_amdgcn_s_setprio(GB ? 1 : 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            if constexpr (!GB) { if (i % PSTEP == 0) issue_piece(i / PSTEP); }
            if (i == 1) W_READ(f0w[0], nW, 0);
            if (i == 2) W_READ(f0w[1], nW, 2048);
            if (i == 3) W_READ(f0w[2], nW, 4096);
            if (i == 4) W_READ(f0w[3], nW, 6144);
            if (i == 5) W_READ(f0x[0], nX, 0);
            if (i == 6) W_READ(f0x[1], nX, 2048);
            if (i == 7) W_READ(f0x[2], nX, 4096);
            if (i == 8) W_READ(f0x[MF >= 4 ? 3 : 0], nX, 6144);
            mfma(f1w[i / MF], f1x[i % MF], acc[i / MF][i % MF]);
            if constexpr (GB) micro(i);
            if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (!GB) issue_done();
          if constexpr (GB) piece_out();
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    };
    int kt_first = 0;
    if constexpr (DGE) {
      // straight-line: the first 8 K-steps of a tile that follows a deferred one carry its pieces (nk >= 8: wide_dge_applies)
      if (pend_valid) {
        kstep_dge(std::integral_constant<int, 0>{}); kstep_dge(std::integral_constant<int, 1>{});
        kstep_dge(std::integral_constant<int, 2>{}); kstep_dge(std::integral_constant<int, 3>{});
        kstep_dge(std::integral_constant<int, 4>{}); kstep_dge(std::integral_constant<int, 5>{});
        kstep_dge(std::integral_constant<int, 6>{}); kstep_dge(std::integral_constant<int, 7>{});
        kt_first = NPEND;
      }
    }
    for (int kt = kt_first; kt < nk; ++kt) {
      if constexpr (DGE) {
        kstep_dge(std::integral_constant<int, -1>{});
      } else
      if constexpr (TN) {
        // ---- TN K-step: the bf16 step below with every ds_read_b128 replaced by two transposing 8-byte reads ----------------
        {
          const uint32_t bo = static_cast<uint32_t>(cur) * STG;
#pragma unroll
          for (int a = 0; a < 4; ++a) { W_TR(h1w[a][0], tW[a][0] + bo, 16384); W_TR(h1w[a][1], tW[a][1] + bo, 16384); }
#pragma unroll
          for (int b = 0; b < MF; ++b) { W_TR(h1x[b][0], tX[b][0] + bo, 8192); W_TR(h1x[b][1], tX[b][1] + bo, 8192); }
        }
        W_WAIT_H(15, h0w, h0x);          // lgkmcnt has 4 bits: the 16 older reads (first half-step) and one of the new ones have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(GB ? 0 : 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if constexpr (GB) { if (i % 3 == 0) issue_piece(i / 3); }
          mfma_h(h0w[i / MF], h0x[i % MF], acc[i / MF][i % MF]);
          if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (do_cs) cs_mfma(h0x);
        if constexpr (GB) issue_done();
        __builtin_amdgcn_sched_barrier(0);
        W_WAIT_H(0, h1w, h1x);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nxt = cur == 2 ? 0 : cur + 1;
        {
          const uint32_t bo = static_cast<uint32_t>(nxt) * STG;
          __builtin_amdgcn_s_setprio(GB ? 1 : 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NM; ++i) {
            if constexpr (!GB) { if (i % 3 == 0) issue_piece(i / 3); }
            if (i < 8) W_TR(h0w[i >> 1][i & 1], tW[i >> 1][i & 1] + bo, 0);
            else W_TR(h0x[(i - 8) >> 1][i & 1], tX[(i - 8) >> 1][i & 1] + bo, 0);
            mfma_h(h1w[i / MF], h1x[i % MF], acc[i / MF][i % MF]);
            if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
          }
          if (do_cs) cs_mfma(h1x);
          if constexpr (!GB) issue_done();
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      } else if constexpr (FP8) {
        // ---- fp8 K-step: 4*MF scaled MFMAs of 128 k, 2*MF per half; same barrier / DMA / store protocol as below ---------
        constexpr int NM2 = 2 * MF;                    // MFMAs per half
        constexpr int NP = MF == 5 ? 7 : 6;            // LDS-DMA pieces per wave and stage
        auto stores8 = [&]() {
          if constexpr (OK != 0 && DEFER) {
            if (pend_valid) {
              for (int j = 0; j < sps; ++j) {
                const int idx = kt * sps + j;
                if (idx < NDEFER) { store_pending(idx); ++ns; }
              }
            }
          }
        };
        {   // first half: W tiles 2,3 of this stage arrive while tiles 0,1 multiply
          const uint32_t bo = static_cast<uint32_t>(cur) * STG;
          const uint32_t w0 = aW + bo, w1 = (aW + bo) ^ 64u;
          W_READ(f0w[2], w0, 4096); W_READ(f1w[2], w1, 4096); W_READ(f0w[3], w0, 6144); W_READ(f1w[3], w1, 6144);
        }
        W_WAIT_ALL(6);   // all but these four and the last X tile's two reads (issued last in the previous half) have landed
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!GB) stores8();
        __builtin_amdgcn_s_setprio(GB ? 0 : 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM2; ++i) {
          if constexpr (GB) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) if (i == pc * NM2 / NP) issue_piece(pc);
          }
          if (i == 2 * (MF - 1)) { W_WAIT_ALL(4); __builtin_amdgcn_sched_barrier(0); }
          mfma8(f0w[i & 1], f1w[i & 1], f0x[i >> 1], f1x[i >> 1], acc[i & 1][i >> 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (GB) issue_done();
        __builtin_amdgcn_sched_barrier(0);
        W_WAIT_ALL(0);
        if (ns == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ns == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        ns = 0;
        __builtin_amdgcn_s_barrier();
        const int nxt = cur == 2 ? 0 : cur + 1;
        {   // second half: W tiles 2,3; the next stage's W tiles 0,1 and X tiles replace registers as they fall free
          const uint32_t bo = static_cast<uint32_t>(nxt) * STG;
          const uint32_t nW0 = aW + bo, nW1 = (aW + bo) ^ 64u, nX0 = aX + bo, nX1 = (aX + bo) ^ 64u;
          if constexpr (GB) stores8();
          __builtin_amdgcn_s_setprio(GB ? 1 : 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NM2; ++i) {
            if constexpr (!GB) {
#pragma unroll
              for (int pc = 0; pc < NP; ++pc) if (i == pc * NM2 / NP) issue_piece(pc);
            }
            if (i == 0) W_READ(f0w[0], nW0, 0);
            if (i == 1) W_READ(f1w[0], nW1, 0);
            if (i == 2) { W_READ(f0w[1], nW0, 2048); W_READ(f0x[0], nX0, 0); }
            if (i == 3) { W_READ(f1w[1], nW1, 2048); W_READ(f1x[0], nX1, 0); }
            if (i == 4) W_READ(f0x[1], nX0, 2048);
            if (i == 5) W_READ(f1x[1], nX1, 2048);
            if constexpr (MF >= 4) { if (i == 6) W_READ(f0x[2], nX0, 4096); if (i == 7) W_READ(f1x[2], nX1, 4096); }
            if constexpr (MF == 5) { if (i == 8) W_READ(f0x[MF > 3 ? 3 : 0], nX0, 6144); if (i == 9) W_READ(f1x[MF > 3 ? 3 : 0], nX1, 6144); }
            mfma8(f0w[2 + (i & 1)], f1w[2 + (i & 1)], f0x[i >> 1], f1x[i >> 1], acc[2 + (i & 1)][i >> 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          // the last X tile's registers are free only now
          if constexpr (MF == 3) { W_READ(f0x[2], nX0, 4096); W_READ(f1x[2], nX1, 4096); }
          if constexpr (MF == 4) { W_READ(f0x[MF > 3 ? 3 : 0], nX0, 6144); W_READ(f1x[MF > 3 ? 3 : 0], nX1, 6144); }
          if constexpr (MF == 5) { W_READ(f0x[MF - 1], nX0, 8192); W_READ(f1x[MF - 1], nX1, 8192); }
          if constexpr (!GB) issue_done();
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      } else {
      W_STAMP(0);   // second half of the previous K-step (+ epilogue at kt = 0)
#ifndef W_ABL_NOREAD
      load_frags(f1w, f1x, cur, 1);                                   // 1
#endif
      if constexpr (MF == 5) W_WAIT_FRAGS(9, f0w, f0x);               //    F0 (older than the 4 + MF F1 reads) is in registers
      else if constexpr (MF == 4) W_WAIT_FRAGS(8, f0w, f0x);
      else W_WAIT_FRAGS(7, f0w, f0x);
      __builtin_amdgcn_sched_barrier(0);
      auto deferred_stores = [&]() {
        if constexpr (OUTBF) {
          if (pend_valid) {
            for (int j = 0; j < sps; ++j) {
              const int idx = kt * sps + j;
              if (idx < NDEFER) { store_pending(idx); ++ns; }
            }
          }
        }
      };
      if constexpr (!GB) deferred_stores();
      __builtin_amdgcn_s_setprio(GB ? 0 : 1);   // the wave of a SIMD that is NOT issuing LDS-DMA in this half goes first (+1 % on the mix)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM; ++i) {                                  // 2
#ifndef W_ABL_NODMA
        if constexpr (GB) { if (i % PSTEP == 0) issue_piece(i / PSTEP); }
#endif
        mfma(f0w[i / MF], f0x[i % MF], acc[i / MF][i % MF]);
        if constexpr (GB) { if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0); }
      }
      if constexpr (GB) issue_done();
      __builtin_amdgcn_sched_barrier(0);
      // 3: vmcnt is ONE in-order queue of loads, stores and LDS-DMA.  Younger than the stage we need: one more stage
      //    (>= 6 pieces per wave) and the `ns` deferred stores issued since the last wait.  Every taken branch costs the
      //    wave ~16 issue cycles, hence the short decision tree instead of a switch over all counts; a smaller count is
      //    always safe.
      W_WAIT_FRAGS(0, f1w, f1x);
      W_STAMP(1);   // first half
      if (ns == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (ns == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      ns = 0;
      W_STAMP(2);   // counted vmcnt wait
      __builtin_amdgcn_s_barrier();
      W_STAMP(3);   // barrier
      const int nxt = cur == 2 ? 0 : cur + 1;
      {                                                               // 4-6: no branches between the MFMAs
        const uint32_t bo = static_cast<uint32_t>(nxt) * STG;
        const uint32_t nW = aW + bo, nX = aX + bo;
        if constexpr (GB) deferred_stores();
        __builtin_amdgcn_s_setprio(GB ? 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
#ifndef W_ABL_NODMA
          if constexpr (!GB) { if (i % PSTEP == 0) issue_piece(i / PSTEP); }
#endif
#ifndef W_ABL_NOREAD
          if (i == 1) W_READ(f0w[0], nW, 0);
          if (i == 2) W_READ(f0w[1], nW, 2048);
          if (i == 3) W_READ(f0w[2], nW, 4096);
          if (i == 4) W_READ(f0w[3], nW, 6144);
          if (i == 5) W_READ(f0x[0], nX, 0);
          if (i == 6) W_READ(f0x[1], nX, 2048);
          if (i == 7) W_READ(f0x[2], nX, 4096);
          if constexpr (MF >= 4) { if (i == 8) W_READ(f0x[MF >= 4 ? 3 : 0], nX, 6144); }
          if constexpr (MF == 5) { if (i == 9) W_READ(f0x[MF - 1], nX, 8192); }
#endif
          mfma(f1w[i / MF], f1x[i % MF], acc[i / MF][i % MF]);
          if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!GB) issue_done();
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
      }
    }
    pend_valid = false;   // sps * nk >= 10: every deferred store of the previous tile has been issued
    if (ti == 0) W_TL(2);              // first tile's K loop done
    if (ti == my_tiles - 1) W_TL(3);   // last tile's K loop done

    // ---- epilogue of tile ti (the next tile's first stages are already in flight) ----------------------------
    int split;
    const int logical = tile_logical(ti, split);
    int tm, tn;
    tile_coords(logical, tm, tn);
    const int m0 = tm * BMt, n0 = tn * wBN;
    if constexpr (TN) {
      // every row of the ones-product holds the column sums: lane (fq = 0, frow) stores column m0 + wm*WR + wn*16 + frow of split `split`
      if (do_cs && fq == 0) {
        const int mc = m0 + wm * WR + wn * 16 + frow;
        if (mc < M) sc.colsum[static_cast<size_t>(split) * M + mc] = csacc[0];
      }
    }
    if constexpr (DGE) {
      // ---- deferred QuickGELU: epi = [bias] + QuickGELU [+ the saved pre-activation] + 16-bit output, nothing else (wide_dge_applies).  The pre-activations move to
      // pendf - the accumulators are free for the next tile at once; a workgroup's last tile and partial tiles finish here.
      {
        w_f32x4_t bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
          bv[a] = (epi & EPI_BIAS) ? *reinterpret_cast<const w_f32x4_t*>(bias + n0 + wn * 64 + a * 16 + fq * 4) : w_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < MF; ++b) pendf[a][b] = acc[a][b] + bv[a];
      }
      if (epi & EPI_SAVE_PRE) {
        // training forward of c_fc: the pre-activation leaves from here as bf16 (the generic epilogue's layout), only the activation waits
        char* p2 = reinterpret_cast<char*>(const_cast<float*>(residual)) +
                   (static_cast<size_t>(m0 + wm * WR + frow) * N + n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
#pragma unroll
        for (int b = 0; b < MF; ++b) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            uint32_t lo[2], hi[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              lo[w] = pack_bf16x2(pendf[2 * pr][b][2 * w], pendf[2 * pr][b][2 * w + 1]);
              hi[w] = pack_bf16x2(pendf[2 * pr + 1][b][2 * w], pendf[2 * pr + 1][b][2 * w + 1]);
            }
            const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
            const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
            if (m0 + wm * WR + b * 16 + frow < M)
              *reinterpret_cast<w_u32x4_t*>(p2 + b * row16 + pr * 64) = w_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
          }
        }
      }
      pend_ptr = static_cast<char*>(out) + (static_cast<size_t>(m0 + wm * WR + frow) * N + n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
      if constexpr (GRP) pend_row16 = row16;
      const bool full = m0 + BMt <= M;
      if (full && ti + 1 < my_tiles) {
        pend_valid = true;
      } else {
#pragma unroll
        for (int j = 0; j < NPEND; ++j)
          if (m0 + wm * WR + (j / 2) * 16 + frow < M) store_pending(j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ns = 0;
      }
    } else {
    // All loads first, then all stores: a load issued behind a store (or waited for with DMA in flight) would wait for
    // every older store to be acknowledged.
    if (FP8 && (epi & EPI_SCALE)) {   // dequantisation of fp8 operands, fused with the bias: acc * (alpha * colscale[n]) + bias[n]
      // (fp8 instantiations only, one n-tile at a time: the 160-row variants have no registers to spare)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const w_f32x4_t cs = *reinterpret_cast<const w_f32x4_t*>(sc.colscale + n0 + wn * 64 + a * 16 + fq * 4) * sc.alpha;
        const w_f32x4_t bv = (epi & EPI_BIAS) ? *reinterpret_cast<const w_f32x4_t*>(bias + n0 + wn * 64 + a * 16 + fq * 4)
                                              : w_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] = acc[a][b] * cs + bv;
      }
    } else if (epi & EPI_BIAS) {
      w_f32x4_t bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) bv[a] = *reinterpret_cast<const w_f32x4_t*>(bias + n0 + wn * 64 + a * 16 + fq * 4);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) acc[a][b] += bv[a];
    }
#ifndef W_NO_SAVE_PRE
    if constexpr (OUTBF && !FP8) {
      if (epi & EPI_SAVE_PRE) {
        // training forward of c_fc: the pre-activation (what QuickGELU' needs in the backward) leaves as bf16 through the residual
        // slot's pointer, straight from the epilogue (same 16-byte layout as the main output), and the activation below is taken
        // from the f32 accumulator: no separate QuickGELU pass over [M, 4d]
        char* p2 = reinterpret_cast<char*>(const_cast<float*>(residual)) +
                   (static_cast<size_t>(m0 + wm * WR + frow) * N + n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
#pragma unroll
        for (int b = 0; b < MF; ++b) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            uint32_t lo[2], hi[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            }
            const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
            const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
            if (m0 + wm * WR + b * 16 + frow < M)
              *reinterpret_cast<w_u32x4_t*>(p2 + b * row16 + pr * 64) = w_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
          }
        }
      }
    }
#endif
    if (epi & EPI_QUICKGELU) {
      // two values at a time: the multiply by -1.702 log2(e), the +1 and the final product are packed-f32 instructions (v_pk_mul_f32 /
      // v_pk_add_f32: two IEEE results per issue slot, same bits as w_quick_gelu); v_exp_f32 / v_rcp_f32 stay one value each
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const w_f32x2_t v = {acc[a][b][j], acc[a][b][j + 1]};
            const w_f32x2_t t = v * w_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
            const w_f32x2_t d = w_f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + w_f32x2_t{1.0f, 1.0f};
            const w_f32x2_t o = v * w_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            acc[a][b][j] = o[0];
            acc[a][b][j + 1] = o[1];
          }
    }
    if constexpr (!FP8)
    if (epi & (EPI_GELU | EPI_RELU)) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[a][b][j] = (epi & EPI_GELU) ? gelu_erf(acc[a][b][j]) : fmaxf(acc[a][b][j], 0.f);
    }
    if ((epi & EPI_RESIDUAL) && !res_first()) add_residual(m0, n0);
    if constexpr (!FP8)
    if (epi & EPI_MUL_DQGELU) {
      // backward of c_fc's QuickGELU fused into the dgrad GEMM: acc *= d/dv [v sigmoid(1.702 v)] at the saved pre-activation
      // `aux` (passed in the residual slot; bf16 in the packed 16-byte layout for 16-bit outputs, else f32 rows)
      auto dq = [](float v) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
        return sg * (1.0f + 1.702f * v * (1.0f - sg));
      };
      if constexpr (OUTBF) {
        const uint16_t* aux = reinterpret_cast<const uint16_t*>(residual);
        const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          int m = m0 + wm * WR + b * 16 + frow;
          m = m < M ? m : M - 1;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const w_u32x4_t q = *reinterpret_cast<const w_u32x4_t*>(aux + static_cast<size_t>(m) * N + col + 32 * pr);
            const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
            const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
            auto lo16 = [](uint32_t w) { return __uint_as_float(w << 16); };
            auto hi16 = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
            acc[2 * pr][b][0] *= dq(lo16(s0[0])); acc[2 * pr][b][1] *= dq(hi16(s0[0]));
            acc[2 * pr][b][2] *= dq(lo16(s1[0])); acc[2 * pr][b][3] *= dq(hi16(s1[0]));
            acc[2 * pr + 1][b][0] *= dq(lo16(s0[1])); acc[2 * pr + 1][b][1] *= dq(hi16(s0[1]));
            acc[2 * pr + 1][b][2] *= dq(lo16(s1[1])); acc[2 * pr + 1][b][3] *= dq(hi16(s1[1]));
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          int m = m0 + wm * WR + b * 16 + frow;
          m = m < M ? m : M - 1;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const w_f32x4_t pv = *reinterpret_cast<const w_f32x4_t*>(residual + static_cast<size_t>(m) * N + n0 + wn * 64 + a * 16 + fq * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][b][j] *= dq(pv[j]);
          }
        }
      }
    }
    const bool full = m0 + BMt <= M;
    if constexpr (OUT8) {
      // e4m3 outputs: a lane's 4 consecutive n of n-tile a become one dword; a 4x4 transpose between the lane quarters (fq) and
      // the n-tile index (v_permlane32_swap on tiles (0,2), (1,3), then v_permlane16_swap on (0,1), (2,3)) leaves quarter fq with
      // 16 consecutive n of one row: n0 + wn*64 + fq*16 .. +15 -> one 16-byte store per 16-row fragment
#pragma unroll
      for (int b = 0; b < MF; ++b) {
        uint32_t t[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(acc[a][b][j] * sc.oscale, -448.f), 448.f);
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
          t[a] = static_cast<uint32_t>(w);
        }
        const w_u2_t p02 = __builtin_amdgcn_permlane32_swap(t[0], t[2], false, false);   // -> [t0.lo, t2.lo], [t0.hi, t2.hi]
        const w_u2_t p13 = __builtin_amdgcn_permlane32_swap(t[1], t[3], false, false);
        const w_u2_t q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
        const w_u2_t q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
        pend[b][0] = q01[0]; pend[b][1] = q01[1]; pend[b][2] = q23[0]; pend[b][3] = q23[1];
      }
      pend_ptr = static_cast<char*>(out) + static_cast<size_t>(m0 + wm * WR + frow) * N + n0 + wn * 64 + fq * 16;
      if constexpr (GRP) pend_row16 = row16;
      if (full && ti + 1 < my_tiles && !(epi & (256 | 512))) {
        pend_valid = true;
      } else if (!(epi & 256)) {
#pragma unroll
        for (int j = 0; j < NPEND; ++j)
          if (m0 + wm * WR + j * 16 + frow < M) w_store16(pend_ptr + j * row16, pend[j]);
      }
    } else if constexpr (OUTBF) {
      // v_permlane16_swap exchanges, between the lane pairs (l, l+16), the packed words of two neighbouring n-tiles: an even
      // lane-row then owns 8 consecutive n of tile a and an odd lane-row 8 consecutive n of tile a+1 -> 16-byte stores.
      const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;    // + 32*pair
#pragma unroll
      for (int b = 0; b < MF; ++b) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint32_t lo[2], hi[2];   // packed words of tiles a = 2pr (lo) and 2pr+1 (hi)
          if (epi & EPI_OUT_F16) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              lo[w] = pack_f16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_f16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            }
          } else {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            }
          }
          const w_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
          const w_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
          pend[b * 2 + pr][0] = s0[0]; pend[b * 2 + pr][1] = s1[0]; pend[b * 2 + pr][2] = s0[1]; pend[b * 2 + pr][3] = s1[1];
        }
      }
      pend_ptr = static_cast<char*>(out) + (static_cast<size_t>(m0 + wm * WR + frow) * N + col) * 2;
      if constexpr (GRP) pend_row16 = row16;
      if (DEFER && full && ti + 1 < my_tiles && !(epi & (256 | 512))) {   // 512 = ablation: store from the epilogue
        pend_valid = true;                       // leave under the next tile's MFMAs
#pragma unroll
        for (int j = NDEFER; j < NPEND; ++j)     // (the part that does not wait; a full tile: every row exists)
          w_store16(pend_ptr + (j / 2) * row16 + (j % 2) * 64, pend[j]);
      } else if (!(epi & 256)) {                 // 256 = timing-only ablation: skip stores
#pragma unroll
        for (int j = 0; j < NPEND; ++j)
          if (m0 + wm * WR + (j / 2) * 16 + frow < M)
            w_store16(pend_ptr + (j / 2) * row16 + (j % 2) * 64, pend[j]);
      }
    } else if (!(epi & 256)) {
#pragma unroll
      for (int b = 0; b < MF; ++b) {
        const int m = m0 + wm * WR + b * 16 + frow;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = n0 + wn * 64 + a * 16 + fq * 4;
          *reinterpret_cast<w_f32x4_t*>(static_cast<float*>(out) + (static_cast<size_t>(split) * M + m) * N + n) = acc[a][b];
        }
      }
    }
    // Epilogue-issued stores are younger than every DMA in flight, so the next counted waits (P + ns outstanding) would
    // simply also retire the DMAs: safe, slightly conservative.  A partial tile may have skipped store instructions.
    if (!full) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ns = 0; }
    }
    if (ti == my_tiles - 1) W_TL(4);   // last epilogue's stores issued
  }
  };
  if (group_b) run(std::true_type{}); else run(std::false_type{});
#ifdef W_STAMPS
  if (lane == 0)
    for (int k = 0; k < 4; ++k) g_wide_stamps[(blockIdx.x * 8 + wid) * 4 + k] = st_sum[k];
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-issued stages must land before the LDS is released
  W_TL(5);   // everything this workgroup issued has completed
}

template <int DT, int OK, int MF, bool TN = false, bool GRP = false, bool DGE = false>
__global__ __launch_bounds__(512) void gemm_wide_kernel(const char* __restrict__ X, const char* __restrict__ W,
                                                        const float* __restrict__ bias_a, const float* residual_a,
                                                        void* out_a, int Mub, int N_a, int K, int epi, int ksplit, int ordG,
                                                        WideScales sc_a, WideProblem p1) {
  gemm_wide_body<DT, OK, MF, TN, GRP, DGE, false>(X, W, bias_a, residual_a, out_a, Mub, N_a, K, epi, ksplit, ordG, sc_a, p1, 0);
}

// Several wgrad GEMMs (TN, see above) in ONE launch: the four weight gradients of a transformer block.  Launched one by one, each of
// them splits its K (= the batch's rows) over S groups of workgroups to fill the chip (S = 4 / 14 / 3 / 3 on the image tower's
// [2304|768|3072|768 x 768|3072] gradients), pays a launch's fixed ~10 us, writes S f32 planes and needs a reduce launch; together they
// have 216 output tiles - one round of the 256 CUs with NO split at all (text tower: 96 tiles, S = 2).  Every workgroup looks up the
// job its blockIdx falls in and runs one (virtual) tile of it through the kernel body.
struct TnJob {
  const char* Xk; const char* Wk; float* out; float* colsum;
  int Mm, Nn, Kpad, kreal, S, first;      // first: this job's first workgroup
};
struct TnMulti { int n; TnJob j[4]; };
__global__ __launch_bounds__(512) void gemm_wide_tn_multi_kernel(TnMulti tab) {
  int p = 0;
  for (int i = 1; i < tab.n; ++i) p = static_cast<int>(blockIdx.x) >= tab.j[i].first ? i : p;
  const TnJob J = tab.j[p];
  const WideScales sc{nullptr, 1.f, 1.f, J.kreal, J.colsum, nullptr};
  gemm_wide_body<1, 0, 4, true, false, false, true>(J.Xk, J.Wk, nullptr, nullptr, J.out, J.Mm, J.Nn, J.Kpad, 0, J.S, 0, sc, WideProblem{},
                                                    static_cast<int>(blockIdx.x) - J.first);
}

bool gemm_wide_supported(int N) { return N % wBN == 0; }

// bench.py's roofline leg (cmh_prof_gemm_*): when launch_gemm hands over an event pair, the forward launch goes through
// hipExtLaunchKernelGGL, which stamps the events with the DISPATCH's own begin / end (what rocprofv3's kernel trace reports),
// instead of bracketing the launch with two hipEventRecord marker packets (those add the inter-packet gaps to every launch).
static hipEvent_t g_wide_ev0 = nullptr, g_wide_ev1 = nullptr;
void gemm_wide_time_next(hipEvent_t start, hipEvent_t stop) { g_wide_ev0 = start; g_wide_ev1 = stop; }
// gemm_lc.hip: the loader / consumer form of this GEMM (round 5): same bits, other wave roles
int gemm_lc_mode();
bool gemm_lc_takes(int dt, int N, int K, int epi);
bool gemm_lc_res_first(int epi, int K);
int launch_gemm_lc(const GemmProblem& a, const GemmProblem* b, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
bool g_force_rows_set();
int launch_gemm_lc2q(const GemmProblem& g, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
bool gemm_lc2q_takes(int N, int K, int epi);
static bool wide_goes_lc(int dt, int epi) {
  const int mode = gemm_lc_mode();
  if (mode == 0 || mode == 7 || g_force_rows_set()) return false;
  if (mode == 2 && (epi & EPI_QUICKGELU)) return false;
  return dt == CMH_BF16;
}
#define W_GO(KERNEL, GRID, ST, ...)                                                                              \
  do {                                                                                                          \
    if (g_wide_ev0) {                                                                                           \
      hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(512), 0, ST, g_wide_ev0, g_wide_ev1, 0, __VA_ARGS__);      \
      g_wide_ev0 = g_wide_ev1 = nullptr;                                                                        \
    } else {                                                                                                    \
      hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(512), 0, ST, __VA_ARGS__);                                    \
    }                                                                                                           \
  } while (0)

// out[i] = sum_s partial[s * n + i]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, size_t n, float* __restrict__ out) {
  const size_t n4 = n / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    w_f32x4_t a = *reinterpret_cast<const w_f32x4_t*>(partial + i * 4);
    for (int s = 1; s < S; ++s) a += *reinterpret_cast<const w_f32x4_t*>(partial + static_cast<size_t>(s) * n + i * 4);
    *reinterpret_cast<w_f32x4_t*>(out + i * 4) = a;
  }
}

// the same for up to four (planes, out) pairs in one launch: blockIdx.y names the job (the multi-problem wgrad launch's sums)
struct ReduceJobs { const float* partial[4]; float* out[4]; size_t n[4]; int S; };
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(ReduceJobs J) {
  const int job = blockIdx.y;
  const float* __restrict__ partial = J.partial[job];
  float* __restrict__ out = J.out[job];
  const size_t n = J.n[job], n4 = n / 4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    w_f32x4_t a = *reinterpret_cast<const w_f32x4_t*>(partial + i * 4);
    for (int s = 1; s < J.S; ++s) a += *reinterpret_cast<const w_f32x4_t*>(partial + static_cast<size_t>(s) * n + i * 4);
    *reinterpret_cast<w_f32x4_t*>(out + i * 4) = a;
  }
}

// Tuning overrides (cmh_gemm_tuning; initial values from CMH_GEMM_BM / CMH_GEMM_ORDER): -1 = decided per launch
static int g_force_rows = []() { const char* e = getenv("CMH_GEMM_BM"); return e ? atoi(e) : -1; }();
bool g_force_rows_set() { return g_force_rows > 0; }      // a pinned tile height names the wide kernel's own variants
static int g_force_order = []() { const char* e = getenv("CMH_GEMM_ORDER"); return e ? atoi(e) : -1; }();

// Deferred QuickGELU (template parameter DGE): which launches take it, and what the activation costs a tile switch in the cost
// models below, in K-steps (tools/gemm_tile_cost.py: switch 1.70 / 1.77 / 2.19 us with a bias epilogue, 3.02 / 3.40 / 5.04 us with
// QuickGELU at 96 / 128 / 160 rows, K-steps of 0.68 / 0.80 / 0.96 us).  CMH_GEMM_DGE=0 turns the variant off (A/B runs).
static bool wide_dge_enabled() {
  static const bool off = []() { const char* e = getenv("CMH_GEMM_DGE"); return e && e[0] == '0'; }();
  return !off;
}
static bool wide_dge_applies(int dt, int epi, int mf, int kmin) {   // kmin: the shortest K of the launch - a tile's 8 pieces need 8 K-steps of the next
  return wide_dge_enabled() && dt == CMH_BF16 && mf == 4 && kmin >= 8 * 64 && (epi & EPI_QUICKGELU) && (epi & (EPI_OUT_BF16 | EPI_OUT_F16)) &&
         !(epi & ~(EPI_BIAS | EPI_QUICKGELU | EPI_OUT_BF16 | EPI_OUT_F16 | EPI_SAVE_PRE)) && (!(epi & EPI_SAVE_PRE) || (epi & EPI_OUT_BF16));
}
static int wide_gelu_ksteps(int dt, int epi, int mf, int kmin) {
  if (!(epi & EPI_QUICKGELU) || wide_dge_applies(dt, epi, mf, kmin)) return 0;
  return mf == 5 ? 3 : 2;
}

// n-panels per group of the tile order (see tile_coords in the kernel); 0 = the n-fastest order
static int wide_order_group(int N) {
  // Round 2 (tools/gemm_bench2.py + bench.py A/B on one box): GROUPS of 3-4 panels inside each XCD's band of m-tiles (ordG 1..64) are
  // +6 % on a back-to-back chain of a block's four GEMMs but -6 ... -10 % on the QKV / c_fc launches INSIDE the encoder (an X tile
  // is re-read once per group, after the outputs have passed through the L2): never the default.
  // Round 4: panel BLOCKS outermost (ordG >= 100, see tile_coords): 2 blocks for 4..11 panels, 3 from 12 on.  Fabric reads per launch
  // 116 -> 90 MB (QKV, N = 2304) and 185 -> 114 MB (c_fc, N = 3072), L2 hit rate 68-72 -> 76 % - and the SAME launch duration to
  // +-1 % (profiles/r04_d_gemm_nblocked_order_ab.txt, r04_d_gemm_nblocked_order_traffic.txt): the K loop does not wait on L2 misses.
  // It is the default for the traffic it saves the other tower's kernels (+0.4 % on the two-stream step, A/B/A/B on one box).
  const int tn = N / wBN;
  auto blocked = [&](int nb) { return tn >= 2 * nb ? 100 + (tn + nb - 1) / nb : 0; };
  if (g_force_order <= -2) return blocked(-g_force_order);     // -NB: the n-blocked order with NB panel blocks
  if (g_force_order >= 0) return g_force_order;                 // 0: plain n-fastest; 1..64: round 2's panel groups
  return tn >= 12 ? blocked(3) : (tn >= 4 ? blocked(2) : 0);
}

static int wide_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 8) cus = 256;
    cus &= ~7;   // whole groups of 8: blockIdx % 8 names the XCD share
  }
  return cus;
}

// Split-K plan for out[M,N] f32 = A.W^T with few tiles and a long K (the wgrad GEMMs): the largest S that divides the K-steps
// and keeps tiles * S within one round of workgroups.  Returns 1 when splitting does not pay.
int gemm_wide_splitk_plan(int dt, int M, int N, int K) {
  if (N % wBN != 0) return 1;
  const int nk = K / (dt == CMH_F32 ? 32 : 64);
  const int tiles = (N / wBN) * ((M + wBM - 1) / wBM);
  int best = 1;
  for (int s = 2; s <= 64 && tiles * s <= wide_cus(); ++s)
    if (nk % s == 0 && nk / s >= 8) best = s;
  return best;
}

// partial planes [S, M, N] f32 in `partials` (S*M*N*4 bytes), reduced into out
int launch_gemm_wide_splitk(int dt, const void* A, const void* W, float* out, float* partials, int S, int M, int N, int K,
                            hipStream_t st) {
  const int total = (N / wBN) * ((M + wBM - 1) / wBM) * S;
  const int cus = wide_cus();
  const int grid = total < cus ? ((total + 7) & ~7) : cus;
  if (dt == CMH_F32)
    hipLaunchKernelGGL((gemm_wide_kernel<0, 0, 5>), dim3(grid), dim3(512), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), nullptr, nullptr, partials, M, N, K, 0, S, wide_order_group(N), WideScales{nullptr, 1.f, 1.f, 0, nullptr, nullptr}, WideProblem{});
  else
    hipLaunchKernelGGL((gemm_wide_kernel<1, 0, 5>), dim3(grid), dim3(512), 0, st, static_cast<const char*>(A),
                       static_cast<const char*>(W), nullptr, nullptr, partials, M, N, K, 0, S, wide_order_group(N), WideScales{nullptr, 1.f, 1.f, 0, nullptr, nullptr}, WideProblem{});
  const size_t n = static_cast<size_t>(M) * N;
  const size_t blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, partials, S, n, out);
  return 0;
}

// out[Mm, Nn] f32 = sum_k Xk[k, m] * Wk[k, n]: bf16 operands as [Kd, Mm] / [Kd, Nn] row-major matrices (the reduction index is the
// slow axis of both: wgrad's dY and X), split over S groups of workgroups whose f32 partial planes are summed afterwards.
// The K range of a split need not divide Kd: rows past Kd read as zeros.  Returns CMH_ERR_INVALID when the shape does not fit
// (Nn % 256, Mm % 128, 32-bit offsets) - the caller then takes the transposing path.
bool gemm_wide_tn_supported(int Mm, int Nn, int Kd) {
  static const bool off = []() { const char* e = getenv("CMH_WGRAD_TN"); return e && e[0] == '0'; }();
  return !off && Nn % wBN == 0 && Mm % 128 == 0 && Kd > 0 &&
         static_cast<size_t>(Kd) * (Mm > Nn ? Mm : Nn) * 2 < (1ull << 32);
}

int launch_gemm_wide_tn(const void* Xk, const void* Wk, float* out, float* partials, size_t part_bytes, int Mm, int Nn, int Kd,
                        hipStream_t st, float* colsum_partial, int* colsum_slices) {
  if (!gemm_wide_tn_supported(Mm, Nn, Kd)) return fail(CMH_ERR_INVALID, "gemm_tn: unsupported shape Mm=%d Nn=%d Kd=%d", Mm, Nn, Kd);
  const int cus = wide_cus();
  const int tiles = (Nn / wBN) * (Mm / 128);
  const int nkt = (Kd + 63) / 64;
  int S = cus / tiles;
  if (S > nkt / 8) S = nkt / 8;
  if (S > 64) S = 64;                                    // (the column-sum partials are sized for 64 slices)
  while (S > 1 && static_cast<size_t>(S) * Mm * Nn * 4 > part_bytes) --S;
  if (S < 1) S = 1;
  const int nk_per = (nkt + S - 1) / S;
  const int Kpad = S * nk_per * 64;                      // the kernel's K: whole K-steps per split; rows >= Kd are zero-filled
  const int total = tiles * S;
  const int grid = total < cus ? ((total + 7) & ~7) : cus;
  const WideScales sc{nullptr, 1.f, 1.f, Kd, colsum_partial, nullptr};   // colsum_partial: [S, Mm] (>= 64 * Mm floats are always enough)
  if (colsum_slices) *colsum_slices = S;
  hipLaunchKernelGGL((gemm_wide_kernel<1, 0, 4, true>), dim3(grid), dim3(512), 0, st, static_cast<const char*>(Xk),
                     static_cast<const char*>(Wk), nullptr, nullptr, S > 1 ? static_cast<void*>(partials) : static_cast<void*>(out),
                     Mm, Nn, Kpad, 0, S, 0, sc, WideProblem{});
  if (S > 1) {
    const size_t n = static_cast<size_t>(Mm) * Nn;
    const size_t blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, partials, S,
                       n, out);
  }
  return 0;
}

// n <= 4 wgrad GEMMs (out_i[Mm_i, Nn_i] = Xk_i^T Wk_i over Kd_i rows) as ONE launch (gemm_wide_tn_multi_kernel).  All jobs take the
// same split S: the largest that still fits the chip with every (tile, split) on its own workgroup; S > 1 puts job i's planes at
// partials + sum_{j<i} S Mm_j Nn_j and sums them afterwards.  colsum_slices[i] = S.  Returns CMH_ERR_INVALID when a shape does not fit
// (the caller then launches them one by one).
bool gemm_wide_tn_multi_enabled() {      // CMH_WGRAD_MULTI=0: the four launches of rounds 1-3 (read per call: the tests flip it)
  const char* e = getenv("CMH_WGRAD_MULTI");
  return !(e && e[0] == '0');
}
bool gemm_wide_tn_multi_fits(const TnMultiJob* jobs, int n) {
  if (n < 1 || n > 4) return false;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (!gemm_wide_tn_supported(jobs[i].Mm, jobs[i].Nn, jobs[i].Kd) || jobs[i].Kd < 8 * 64) return false;
    tiles += (jobs[i].Nn / wBN) * (jobs[i].Mm / 128);
  }
  return tiles <= wide_cus();
}
int launch_gemm_wide_tn_multi(const TnMultiJob* jobs, int n, float* partials, size_t part_bytes, hipStream_t st, int* slices) {
  if (n < 1 || n > 4) return fail(CMH_ERR_INVALID, "gemm_tn_multi: %d jobs", n);
  const int cus = wide_cus();
  int tiles = 0, min_nkt = 1 << 30;
  for (int i = 0; i < n; ++i) {
    if (!gemm_wide_tn_supported(jobs[i].Mm, jobs[i].Nn, jobs[i].Kd))
      return fail(CMH_ERR_INVALID, "gemm_tn_multi: unsupported shape Mm=%d Nn=%d Kd=%d", jobs[i].Mm, jobs[i].Nn, jobs[i].Kd);
    tiles += (jobs[i].Nn / wBN) * (jobs[i].Mm / 128);
    const int nkt = (jobs[i].Kd + 63) / 64;
    min_nkt = nkt < min_nkt ? nkt : min_nkt;
  }
  if (tiles > cus) return fail(CMH_ERR_INVALID, "gemm_tn_multi: %d tiles for %d CUs", tiles, cus);
  int S = cus / tiles;
  if (S > min_nkt / 8) S = min_nkt / 8;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  size_t need = 0;
  for (int i = 0; i < n; ++i) need += static_cast<size_t>(S) * jobs[i].Mm * jobs[i].Nn * 4;
  while (S > 1 && need > part_bytes) {
    --S;
    need = 0;
    for (int i = 0; i < n; ++i) need += static_cast<size_t>(S) * jobs[i].Mm * jobs[i].Nn * 4;
  }
  TnMulti tab;
  tab.n = n;
  int first = 0;
  float* plane = partials;
  for (int i = 0; i < n; ++i) {
    const TnMultiJob& g = jobs[i];
    const int nkt = (g.Kd + 63) / 64, nk_per = (nkt + S - 1) / S;
    TnJob& J = tab.j[i];
    J.Xk = static_cast<const char*>(g.Xk); J.Wk = static_cast<const char*>(g.Wk);
    J.out = S > 1 ? plane : g.out;
    J.colsum = g.colsum;
    J.Mm = g.Mm; J.Nn = g.Nn; J.Kpad = S * nk_per * 64; J.kreal = g.Kd; J.S = S; J.first = first;
    first += (g.Nn / wBN) * (g.Mm / 128) * S;
    plane += static_cast<size_t>(S) * g.Mm * g.Nn;
    if (slices) slices[i] = S;
  }
  for (int i = n; i < 4; ++i) tab.j[i] = TnJob{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 1, 1 << 30};
  hipLaunchKernelGGL(gemm_wide_tn_multi_kernel, dim3(first), dim3(512), 0, st, tab);
  CMH_CHECK_LAUNCH("gemm_tn_multi");
  if (S > 1) {
    ReduceJobs R;
    R.S = S;
    plane = partials;
    size_t most = 0;
    for (int i = 0; i < 4; ++i) {
      const size_t cnt = i < n ? static_cast<size_t>(jobs[i].Mm) * jobs[i].Nn : 0;
      R.partial[i] = plane; R.out[i] = i < n ? jobs[i].out : nullptr; R.n[i] = cnt;
      plane += static_cast<size_t>(S) * cnt;
      most = cnt > most ? cnt : most;
    }
    const size_t blocks = (most / 4 + 255) / 256;
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(static_cast<unsigned>(blocks < 1024 ? blocks : 1024), n), dim3(256), 0, st, R);
    CMH_CHECK_LAUNCH("gemm_tn_multi (reduce)");
  }
  return 0;
}

int launch_gemm_wide(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out,
                     int M, int N, int K, int epi, hipStream_t st, const float* colscale, float alpha, float oscale,
                     const int32_t* m_dev, int m_hint) {
  if (wide_goes_lc(dt, epi) && gemm_lc_takes(dt, N, K, epi)) {
    const GemmProblem g{A, W, bias, residual, out, M, N, K, m_dev, m_hint, nullptr, 1.f, 1.f};
    const int rc = launch_gemm_lc(g, nullptr, epi, st, g_wide_ev0, g_wide_ev1);
    g_wide_ev0 = g_wide_ev1 = nullptr;
    return rc;
  }
  if (dt == CMH_FP8 && !g_force_rows_set() && gemm_lc2q_takes(N, K, epi)) {      // the 12-wave form on e4m3 operands (experiment, CMH_GEMM_LC=7)
    const GemmProblem g{A, W, bias, residual, out, M, N, K, m_dev, m_hint, colscale, alpha, oscale};
    const int rc = launch_gemm_lc2q(g, epi, st, g_wide_ev0, g_wide_ev1);
    g_wide_ev0 = g_wide_ev1 = nullptr;
    return rc;
  }
  const WideScales sc{colscale, alpha, oscale, 0, nullptr, m_dev};
  const size_t esz = dt == CMH_F32 ? 4 : (dt == CMH_FP8 ? 1 : 2);
  if (static_cast<size_t>(M) * K * esz >= (1ull << 32) || static_cast<size_t>(wBN) * K * esz >= (1ull << 32))
    return fail(CMH_ERR_INVALID, "gemm: operand of %zu bytes exceeds the 32-bit offset range of the wide kernel",
                static_cast<size_t>(M) * K * esz);
  // Tile rows: 160 (MF = 5), 128 (MF = 4) or 96 (MF = 3), whichever needs less time for this M on the chip's CUs: rounds of
  // workgroups x (K-steps + ~4 K-steps of per-tile overhead) x (rows + a share that does not shrink with the tile).  M = 12 800 /
  // 19 712 (the dense towers) take 160; the packed text rows (M ~ 10 k) take 128 at N = 1536 / 2048 and 96 at N = 512, where one
  // round of 220 tiles keeps 86 % of the CUs busy instead of 65 % (166 tiles of 128 rows).  CMH_GEMM_BM=96|128|160 forces one.
  const int cus = wide_cus();
  const int nk = K / (dt == CMH_F32 ? 32 : (dt == CMH_FP8 ? 128 : 64));
  const int Mc = m_dev && m_hint > 0 && m_hint <= M ? m_hint : M;   // the tile height is chosen for the likely row count
  auto cost = [&](int mf) {   // rounds x (rows + the per-K-step cost that does not shrink with the tile: W fragment reads, barrier) x K-steps
    const int tiles = (N / wBN) * ((Mc + 32 * mf - 1) / (32 * mf));
    return static_cast<long long>((tiles + cus - 1) / cus) * (10 * mf + 6) * (nk + 4 + wide_gelu_ksteps(dt, epi, mf, K));
  };
  const int forced = g_force_rows;
  // fp8: 160 rows by default only with e4m3 OUTPUT (the c_fc launches).  The 160-row variant needs 56 fragment registers live across the epilogue
  // (both 16-byte halves of the next K-step's operands, as aligned 8-register MFMA operands) next to 80 accumulators and the pending
  // stores; left alone hipcc spills inside the K loop (scratch reloads with vmcnt(0) drain the LDS-DMA pipeline: measured 64 us
  // against bf16's 46 on QKV).  With the epilogue paths fp8 never takes compiled out, the residual loads in two groups and - for the
  // 16-bit outputs - only 4 of the 10 stores deferred, the K loops of the e4m3- and 16-bit-output variants are free of scratch
  // (tools/asm_loop_scratch.py: every variant a launch can select must show no scratch at loop depth 2).
  // (the f32-output fp8 variant still spills in its K loop; the 16-bit-output one is loop-clean but pays 28 scratch instructions per
  // tile in its epilogue: serialized GEMM time of an fp8 step -1.3 %, the overlapped step -1.5 % in pairs/s, A/B/A/B on one box - so
  // it is taken only on request, CMH_GEMM_BM=160 / cmh_gemm_tuning)
  const bool fp8_160 = epi & (EPI_OUT_FP8 | EPI_OUT_BF16 | EPI_OUT_F16);
  int mf = dt == CMH_FP8 && !(epi & EPI_OUT_FP8) ? 4 : 5;
  if (mf == 5 && cost(4) < cost(mf)) mf = 4;
  if (cost(3) < cost(mf)) mf = 3;
  if (forced == 96 || forced == 128 || (forced == 160 && (dt != CMH_FP8 || fp8_160))) mf = forced / 32;
  const int total = (N / wBN) * ((M + 32 * mf - 1) / (32 * mf));
  int grid = total < cus ? ((total + 7) & ~7) : cus;
  const int ordg = wide_order_group(N);
#define W_LAUNCH(DT, OK)                                                                                                  \
  do {                                                                                                                    \
    if (mf == 3)                                                                                                          \
      W_GO((gemm_wide_kernel<DT, OK, 3>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,       \
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});                                                                     \
    else if (mf == 4)                                                                                                     \
      W_GO((gemm_wide_kernel<DT, OK, 4>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,       \
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});                                                                     \
    else                                                                                                                  \
      W_GO((gemm_wide_kernel<DT, OK, 5>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,       \
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});                                                                     \
  } while (0)
  const bool obf = epi & (EPI_OUT_BF16 | EPI_OUT_F16);   // 16-bit outputs share the packed store path
  const bool o8 = epi & EPI_OUT_FP8;
  if (dt == CMH_F32) { if (obf) W_LAUNCH(0, 1); else W_LAUNCH(0, 0); }
  else if (dt == CMH_BF16) {
    if (obf && wide_dge_applies(dt, epi, mf, K))
      W_GO((gemm_wide_kernel<1, 1, 4, false, false, true>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});
    else if (obf) W_LAUNCH(1, 1);
    else W_LAUNCH(1, 0);
  }
  else {
#undef W_LAUNCH
#define W_LAUNCH(DT, OK)                                                                                                  \
  do {                                                                                                                    \
    if (mf == 3)                                                                                                          \
      W_GO((gemm_wide_kernel<DT, OK, 3>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,       \
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});                                                                     \
    else                                                                                                                  \
      W_GO((gemm_wide_kernel<DT, OK, 4>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias,       \
           residual, out, M, N, K, epi, 1, ordg, sc, WideProblem{});                                                                     \
  } while (0)
    if (o8) {
      if (mf == 5)
        W_GO((gemm_wide_kernel<2, 2, 5>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias, residual, out, M, N,
             K, epi, 1, ordg, sc, WideProblem{});
      else
        W_LAUNCH(2, 2);
    } else if (obf) {
      if (mf == 5)
        W_GO((gemm_wide_kernel<2, 1, 5>), grid, st, static_cast<const char*>(A), static_cast<const char*>(W), bias, residual, out, M, N,
             K, epi, 1, ordg, sc, WideProblem{});
      else
        W_LAUNCH(2, 1);
    } else {
      W_LAUNCH(2, 0);
    }
  }
#undef W_LAUNCH
  return 0;
}

// Two problems in one persistent grid (template parameter GRP; see WideProblem).  `a` should be the problem with the longer K (its
// tiles go first); both share dt, the output kind and the epilogue flags.  The tile height is the one whose WORST workgroup - the
// kernel's own static assignment, replayed here - finishes first.
// Does ONE grouped launch beat two plain ones?  The same cost units as the tile-height choice (K-steps + 4 per tile, x rows + 6),
// the worst workgroup of the kernel's static assignment against the two plain launches' rounds, plus a fixed ~8 K-steps per LAUNCH
// (first stage landing on every CU at once, last epilogue's store drain: profiles/r02_a_gemm_launch_timeline.txt).  Measured at batch
// 256 (profiles/r04_b_grouped_per_shape.txt): QKV 74.5 -> 68.8 us, out_proj 41.8 -> 36.9, c_fc 104.8 -> 96.4 grouped - but c_proj 92.0
// -> 98.3: one 48-K-step image tile per workgroup plus a 32-K-step text tile on every second one is a worse packing than two launches;
// the model reproduces all four.
static long long wide_plain_cost(int dt, const GemmProblem& g, int epi, int cus) {
  const int nk = g.K / (dt == CMH_F32 ? 32 : (dt == CMH_FP8 ? 128 : 64));
  const int M = g.m_dev && g.m_hint > 0 && g.m_hint <= g.M ? g.m_hint : g.M;
  long long best = -1;
  for (int mf = (dt == CMH_FP8 && !(epi & EPI_OUT_FP8)) ? 4 : 5; mf >= 3; --mf) {
    const int tiles = (g.N / wBN) * ((M + 32 * mf - 1) / (32 * mf));
    const long long c = static_cast<long long>((tiles + cus - 1) / cus) * (10 * mf + 6) * (nk + 4 + wide_gelu_ksteps(dt, epi, mf, g.K));
    if (best < 0 || c < best) best = c;
  }
  return best;
}
static long long wide_grouped_cost(int dt, const GemmProblem& a, const GemmProblem& b, int epi, int cus, int* mf_out);

bool gemm_wide_grouping_pays(int dt, const GemmProblem& a, const GemmProblem& b, int epi) {
  static const bool always = []() { const char* e = getenv("CMH_GEMM_GROUPED"); return e && !strcmp(e, "always"); }();
  if (always || g_force_rows > 0) return true;            // (a forced tile height: A/B runs and the tests that walk every variant)
  const int cus = wide_cus();
  const long long fixed = 8 * 56;
  return wide_grouped_cost(dt, a, b, epi, cus, nullptr) + fixed <= wide_plain_cost(dt, a, epi, cus) + wide_plain_cost(dt, b, epi, cus) + 2 * fixed;
}

static long long wide_grouped_cost(int dt, const GemmProblem& a, const GemmProblem& b, int epi, int cus, int* mf_out) {
  const int bk = dt == CMH_F32 ? 32 : (dt == CMH_FP8 ? 128 : 64);
  const int nk0 = a.K / bk, nk1 = b.K / bk;
  auto likely = [](const GemmProblem& g) { return g.m_dev && g.m_hint > 0 && g.m_hint <= g.M ? g.m_hint : g.M; };
  const int Ma = likely(a), Mb = likely(b);
  auto tiles_of = [](int M, int N, int mf) { return (N / wBN) * ((M + 32 * mf - 1) / (32 * mf)); };
  auto cost = [&](int mf) {
    const int total = tiles_of(a.M, a.N, mf) + tiles_of(b.M, b.N, mf);
    const int per = (total < cus ? ((total + 7) & ~7) : cus) >> 3;
    const int t0 = tiles_of(Ma, a.N, mf), t1 = tiles_of(Mb, b.N, mf);
    long long worst = 0;
    for (int x = 0; x < 8; ++x) {
      const int len0 = (t0 >> 3) + (x < (t0 & 7)), len1 = (t1 >> 3) + (x < (t1 & 7));
      for (int sl = 0; sl < per; ++sl) {
        const int n0 = sl < len0 ? (len0 - sl + per - 1) / per : 0;
        const int nall = sl < len0 + len1 ? (len0 + len1 - sl + per - 1) / per : 0;
        const int ge = wide_gelu_ksteps(dt, epi, mf, a.K < b.K ? a.K : b.K);
        const long long c = static_cast<long long>(n0) * (nk0 + 4 + ge) + static_cast<long long>(nall - n0) * (nk1 + 4 + ge);
        worst = c > worst ? c : worst;
      }
    }
    return worst * (10 * mf + 6);
  };
  int mf = dt == CMH_FP8 && !(epi & EPI_OUT_FP8) ? 4 : 5;
  if (mf == 5 && cost(4) < cost(mf)) mf = 4;
  if (cost(3) < cost(mf)) mf = 3;
  if (mf_out) *mf_out = mf;
  return cost(mf);
}

int launch_gemm_wide_grouped(int dt, const GemmProblem& a, const GemmProblem& b, int epi, hipStream_t st) {
  if (wide_goes_lc(dt, epi) && gemm_lc_takes(dt, a.N, a.K, epi) && gemm_lc_takes(dt, b.N, b.K, epi) &&
      gemm_lc_res_first(epi, a.K) == gemm_lc_res_first(epi, b.K)) {
    const int rc = launch_gemm_lc(a, &b, epi, st, g_wide_ev0, g_wide_ev1);
    g_wide_ev0 = g_wide_ev1 = nullptr;
    return rc;
  }
  const size_t esz = dt == CMH_F32 ? 4 : (dt == CMH_FP8 ? 1 : 2);
  for (const GemmProblem* g : {&a, &b})
    if (static_cast<size_t>(g->M) * g->K * esz >= (1ull << 32) || static_cast<size_t>(wBN) * g->K * esz >= (1ull << 32))
      return fail(CMH_ERR_INVALID, "gemm (grouped): operand of %zu bytes exceeds the 32-bit offset range of the wide kernel",
                  static_cast<size_t>(g->M) * g->K * esz);
  const int cus = wide_cus();
  auto tiles_of = [](int M, int N, int mf) { return (N / wBN) * ((M + 32 * mf - 1) / (32 * mf)); };
  auto grid_of = [&](int mf) {   // sized for the upper bounds: workgroups beyond the real tile count exit at once
    const int total = tiles_of(a.M, a.N, mf) + tiles_of(b.M, b.N, mf);
    return total < cus ? ((total + 7) & ~7) : cus;
  };
  const int forced = g_force_rows;
  const bool fp8_160 = epi & EPI_OUT_FP8;       // (the 16-bit-output 160-row fp8 variant is an opt-in of the plain launches only)
  int mf = 5;
  (void)wide_grouped_cost(dt, a, b, epi, cus, &mf);
  if (forced == 96 || forced == 128 || (forced == 160 && (dt != CMH_FP8 || fp8_160))) mf = forced / 32;
  const int grid = grid_of(mf);
  const WideScales sc{a.colscale, a.alpha, a.oscale, 0, nullptr, a.m_dev};
  const WideProblem p1{static_cast<const char*>(b.A), static_cast<const char*>(b.W), b.bias, b.residual, b.out, b.colscale, b.m_dev,
                       b.M, b.N, b.K, b.alpha, b.oscale};
#define W_LAUNCH_G(DT, OK, MFV)                                                                                            \
  W_GO((gemm_wide_kernel<DT, OK, MFV, false, true>), grid, st, static_cast<const char*>(a.A), static_cast<const char*>(a.W), a.bias, \
       a.residual, a.out, a.M, a.N, a.K, epi, 1, 0, sc, p1)
#define W_LAUNCH_G3(DT, OK)                                                                                                \
  do { if (mf == 3) W_LAUNCH_G(DT, OK, 3); else if (mf == 4) W_LAUNCH_G(DT, OK, 4); else W_LAUNCH_G(DT, OK, 5); } while (0)
  const bool obf = epi & (EPI_OUT_BF16 | EPI_OUT_F16);
  if (dt == CMH_F32) {
    if (obf) return fail(CMH_ERR_INVALID, "gemm (grouped): f32 operands come with f32 outputs");
    W_LAUNCH_G3(0, 0);
  } else if (dt == CMH_BF16) {
    if (!obf) return fail(CMH_ERR_INVALID, "gemm (grouped): bf16 operands come with 16-bit outputs");
    if (wide_dge_applies(dt, epi, mf, a.K < b.K ? a.K : b.K))
      W_GO((gemm_wide_kernel<1, 1, 4, false, true, true>), grid, st, static_cast<const char*>(a.A), static_cast<const char*>(a.W), a.bias,
           a.residual, a.out, a.M, a.N, a.K, epi, 1, 0, sc, p1);
    else
      W_LAUNCH_G3(1, 1);
  } else if (epi & EPI_OUT_FP8) {
    W_LAUNCH_G3(2, 2);
  } else {
    if (!obf) return fail(CMH_ERR_INVALID, "gemm (grouped): fp8 operands come with 16-bit or e4m3 outputs");
    if (mf == 3) W_LAUNCH_G(2, 1, 3); else W_LAUNCH_G(2, 1, 4);
  }
#undef W_LAUNCH_G3
#undef W_LAUNCH_G
  CMH_CHECK_LAUNCH("gemm (grouped)");
  return 0;
}

}  // namespace cmh

extern "C" int cmh_gemm_tuning(int32_t tile_rows, int32_t order_group) {
  using namespace cmh;
  CMH_CHECK_ARG(tile_rows == -1 || tile_rows == 96 || tile_rows == 128 || tile_rows == 160, "gemm_tuning: tile_rows %d (-1, 96, 128, 160)", tile_rows);
  CMH_CHECK_ARG(order_group >= -8 && order_group <= 64, "gemm_tuning: order_group %d (-8..-2: n-blocked, -1: default, 0: n-fastest, > 0: panel groups)", order_group);
  g_force_rows = tile_rows;
  g_force_order = order_group;
  return CMH_OK;
}

#ifdef W_TIMELINE
extern "C" int cmh_debug_wide_timeline(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cmh::g_wide_tl), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
extern "C" int cmh_debug_wide_timeline_clear() {
  static unsigned long long z[256 * 8];
  return hipMemcpyToSymbol(HIP_SYMBOL(cmh::g_wide_tl), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
#ifdef W_STAMPS
extern "C" int cmh_debug_wide_stamps(unsigned* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cmh::g_wide_stamps), sizeof(unsigned) * 256 * 8 * 4) == hipSuccess ? 0 : -1;
}
#endif
