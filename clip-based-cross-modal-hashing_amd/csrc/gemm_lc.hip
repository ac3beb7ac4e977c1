// GEMM "lc" (loader / consumer waves, round 5): out[M,N] = epi(X[M,K].W[N,K]^T), bf16 operands, 16-bit output, 128(m) x 256(n) tile per
// 512-thread workgroup, persistent like the wide kernel (gemm_wide.hip) - but with the two jobs of its K loop on DIFFERENT waves.
//
// Why.  The wide kernel's eight waves each issue their share of a stage's LDS-DMA pieces between their own MFMAs.  A wave issues in
// order: while it sits in the ~60-180 cycle issue of a `global_load_lds_dwordx4` (the CU's address path moves 64 B/clk) it issues
// no MFMA, and its SIMD partner has only its own 20 MFMAs of that half-step to cover the hole (anti-phase issue).  Ablation
// of round 1: the same loop with the DMA compiled out runs 19 % faster; the K-step costs 1750-2000 cycles against 1280 of MFMA work.
// Here waves 0..3 (one per SIMD) do nothing but stage operands: 12 pieces of 1 KiB per wave and K-step, a counted `s_waitcnt vmcnt`,
// the barrier.  Waves 4..7 (the other wave of each SIMD) do nothing but multiply: 64 x 128 of the tile each, 64 MFMAs per K-step
// issued back to back from ONE wave (a single wave keeps a SIMD's matrix pipe full: 16 cycles per v_mfma_f32_16x16x32_bf16,
// MI355X_MICROARCH cycle table), fragment reads in the MFMAs' issue gaps.  Their vmcnt queue holds only their own epilogue's loads and
// stores: nothing in the K loop ever waits for a store.
//
// Ring and protocol (the wide kernel's): 3 stages x 48 KB (W 256 rows x 128 B | X 128 rows x 128 B, lane-linear image, XOR swizzle on
// the DMA source chunk and on the ds_read_b128); K-step s lives in buffer s % 3; ONE s_barrier per K-step, placed after the MFMA
// waves hold every fragment of stage s in registers: behind it the loaders overwrite buffer s % 3 with stage s + 3 (two K-steps of
// flight) and the MFMA waves start reading stage s + 1, which the loaders' counted wait in front of the barrier has seen land.
// Fragments: the W operand's eight 16-row fragments are refilled IN PLACE (fragment a of the next half-step is read as soon as the
// four MFMAs that use fragment a have issued), the X operand's four are double-buffered: 64 fragment registers + 128 accumulators.
// Same k order per output element as the wide kernel (k = 64 kt + 32 ks + 8 fq + j inside v_mfma_f32_16x16x32_bf16, K-steps in
// sequence), same epilogue operation order, same residual-first rule: the SAME BITS (tests/test_gpu_lc.py compares with torch.equal).
//
// Outcome (round 5): an EXPERIMENT, off by default (CMH_GEMM_LC / cmh_set_gemm_lc), not the product path.  The K loop does
// what it was built for - 1 240-1 300 cycles per K-step against 1 024 of MFMA issue, no store or DMA instruction in the MFMA waves'
// stream - and WITHOUT its output stores the kernel beats the wide kernel by 7 % (QKV 42.9 against 46.1 us).  With them it loses:
// the epilogue of the ONE MFMA wave a SIMD has (4 850-4 930 cycles per tile: ~400 VALU instructions and 16 KB of stores through a CU
// write path that moves ~16-20 B/clk, the same with 36 busy CUs as with 256) is covered by nobody, where the wide kernel's second
// wave per SIMD multiplies meanwhile; the drained stores also slow the next K-steps' DMA (1 292 -> 1 554 cycles).  Stores cost
// this kernel 21-24 % of a launch, the wide kernel 3-11 % (profiles/r05_l, r05_m, r05_n).  Built and measured on the way, none
// kept: the epilogue handed to the staging waves (role swap: they have ~370 idle cycles per K-step, not the 1 600 a tile's epilogue
// needs), stores deferred into the next tile's K-steps (a store in the lone MFMA wave's in-order stream stalls its SIMD).  Routing
// the output through LDS to the staging waves' idle registers fails on capacity: ring 144 KB + bias 2 KB leave 14 KB of 160.
// What does work is the 12-wave form further down (gemm_lc2_kernel): two MFMA waves per SIMD again, the staging still on its own waves.
#include <cstdlib>
#include <cstring>

#include "cmh_common.h"

#include <hip/hip_ext.h>

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 lc_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float lc_f32x4_t;
typedef __attribute__((ext_vector_type(2))) float lc_f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t lc_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned lc_u2_t;
typedef const __attribute__((address_space(1))) void* lc_gptr_t;
typedef __attribute__((address_space(3))) void* lc_lptr_t;

constexpr int lcBM = 128, lcBN = 256;
constexpr int lcRowBytes = 128;                    // one K-step of one row: 64 bf16
constexpr int lcWBytes = lcBN * lcRowBytes;        // 32 KB
constexpr int lcSTG = lcWBytes + lcBM * lcRowBytes;   // 48 KB per stage

struct LcProblem {
  const char* X; const char* W; const float* bias; const void* residual; void* out;
  const int* m_dev;      // optional: the real row count on the device (Mub is then an upper bound)
  int Mub, N, K;
  const float* colscale = nullptr;      // fp8 operands (gemm_lc2q_kernel): out = epi(alpha * colscale[n] * (X8 . W8^T) + bias)
  float alpha = 1.f, oscale = 1.f;
};

__device__ __forceinline__ int lc_swz(int row, int chunk) { return row * lcRowBytes + ((chunk ^ (row & 7)) << 4); }

#define LC_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

#ifdef LC_STAMPS   // diagnostic build only (tools/lc_stamps.py; make BUILD=build_st EXTRA=-DLC_STAMPS): per workgroup, MFMA wave 0: shader-clock
                   // ticks (s_memtime) of the whole kernel, until the first stage, inside the K loops, inside the epilogues; and the 100 MHz
                   // wall clock over the same span (clock = ticks / wall)
__device__ unsigned long long g_lc_stamps[256 * 8];
#define LC_T() __builtin_amdgcn_s_memtime()
#endif

// RF: the residual-first rule applies to this launch (short K with a residual: the accumulators start as the residual tile); a template
// parameter so that a tile's first K-step is one straight path - C = 0 inside the first MFMAs, or the loaded residual
// ABL (diagnostic instantiations only, CMH_LC_ABL): 1 the MFMA waves skip their MFMAs (what the feed alone sustains), 2 the loaders stage
// nothing after the prologue (what the MFMA side alone sustains), 4 one stage in flight instead of two
// F16O: the 16-bit output is IEEE fp16 (the residual stream) instead of bf16 - compile-time, like RF: inside the epilogue every taken
// branch costs a lone wave ~16 issue cycles, and there were three per 16-byte store
// RES: 0 no residual, 1 residual first (RF), 2 residual behind the bias (long K) - compile-time as well: with the late-residual code in
// the epilogue of the launches that never take it, the allocator spilled 165 registers around it
template <bool GRP, int RES, bool F16O, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_lc_kernel(LcProblem p0, LcProblem p1, int epi) {
  constexpr bool RF = RES == 1;
  __shared__ __attribute__((aligned(1024))) char lds[3 * lcSTG + 2 * 1024];      // the ring + two bias rows (tile parity)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // every field the kernel reads, as wave-uniform locals (selecting between the two by-value structs at run time would put them on the stack)
  const char* const X0 = p0.X; const char* const W0 = p0.W; const float* const B0 = p0.bias;
  const void* const R0 = p0.residual; void* const O0 = p0.out;
  const int N0 = p0.N, K0 = p0.K;
  const char* const X1 = p1.X; const char* const W1 = p1.W; const float* const B1 = p1.bias;
  const void* const R1 = p1.residual; void* const O1 = p1.out;
  const int N1 = p1.N, K1 = p1.K;
  int M0 = p0.Mub;
  if (p0.m_dev) { const int md = *p0.m_dev; M0 = md < M0 ? md : M0; }
  M0 = __builtin_amdgcn_readfirstlane(M0);        // (a loaded value: the compiler cannot know that it is uniform)
  int M1 = 0;
  if constexpr (GRP) {
    M1 = p1.Mub;
    if (p1.m_dev) { const int md = *p1.m_dev; M1 = md < M1 ? md : M1; }
    M1 = __builtin_amdgcn_readfirstlane(M1);
  }
  // ---- this workgroup's tiles: the wide kernel's static assignment.  XCD x = blockIdx % 8 owns a contiguous eighth of each problem's
  // n-fastest tile order; workgroup `slot` of the XCD takes positions slot, slot + per, ... of the concatenation of the two eighths.
  const int tiles_n0 = N0 / lcBN, tiles_n1 = GRP ? N1 / lcBN : 1;
  const int total0 = tiles_n0 * ((M0 + lcBM - 1) / lcBM);
  const int total1 = GRP ? tiles_n1 * ((M1 + lcBM - 1) / lcBM) : 0;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q0 = total0 >> 3, r0 = total0 & 7, q1 = total1 >> 3, r1 = total1 & 7;
  const int lo0 = xcd < r0 ? xcd * (q0 + 1) : r0 * (q0 + 1) + (xcd - r0) * q0, len0 = xcd < r0 ? q0 + 1 : q0;
  const int lo1 = xcd < r1 ? xcd * (q1 + 1) : r1 * (q1 + 1) + (xcd - r1) * q1, len1 = xcd < r1 ? q1 + 1 : q1;
  const int n_first = slot < len0 ? (len0 - slot + per - 1) / per : 0;
  const int span = len0 + len1;
  const int my_tiles = slot < span ? (span - slot + per - 1) / per : 0;
  if (my_tiles == 0) return;
  const int nk0 = K0 / 64, nk1 = GRP ? K1 / 64 : 0;
  const int S = n_first * nk0 + (my_tiles - n_first) * nk1;      // K-steps of this workgroup = barriers after the prologue's
  // tile ti -> (second problem?, m0, n0)
  auto tile_of = [&](int ti, bool& second, int& m0, int& n0) {
    const int j = slot + ti * per;
    second = GRP && ti >= n_first;
    const int logical = second ? lo1 + (j - len0) : lo0 + j;
    const int tn_cnt = second ? tiles_n1 : tiles_n0;
    const int tm = logical / tn_cnt;
    m0 = tm * lcBM;
    n0 = (logical - tm * tn_cnt) * lcBN;
  };

  if (wid < 4) {
    // =================================================== loader waves ===============================================================
    // Piece = 1 KiB = 8 rows x 128 B: lane i fills (row 8 piece + i / 8, physical chunk i % 8) and fetches logical chunk (i % 8) ^ (row & 7).
    // Wave l stages W pieces 8 l .. 8 l + 7 and X pieces 4 l .. 4 l + 3 of every stage.  Source = uniform base + 32-bit lane offset.
    const int sub = lane >> 3, ch = lane & 7;
    uint32_t offW[8], offX[4];
    const char* Wt = nullptr;
    const char* Xt = nullptr;
    const char* Bt = nullptr;      // &bias[n0] of the tile being staged
    int i_nk = 0;
    int w_prob = -1;               // the problem offW was computed for
    auto set_tile = [&](int ti) {
      bool second; int m0, n0;
      tile_of(ti, second, m0, n0);
      const uint32_t rs = static_cast<uint32_t>(second ? K1 : K0) * 2;
      if (w_prob != static_cast<int>(second)) {      // once per problem: the W offsets follow its row stride
        w_prob = static_cast<int>(second);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (wid * 8 + i) * 8 + sub;
          offW[i] = static_cast<uint32_t>(row) * rs + ((ch ^ (row & 7)) << 4);
        }
      }
      i_nk = second ? nk1 : nk0;
      const int Mp = second ? M1 : M0;
      Wt = (second ? W1 : W0) + static_cast<size_t>(n0) * rs;
      Xt = second ? X1 : X0;
      Bt = reinterpret_cast<const char*>((second ? B1 : B0) + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + sub;
        int xr = m0 + row;
        xr = xr < Mp ? xr : Mp - 1;               // rows past M are computed on duplicated data and never stored
        offX[i] = static_cast<uint32_t>(xr) * rs + ((ch ^ (row & 7)) << 4);      // < 4 GiB: checked on the host
      }
    };
    int i_tile = 0, i_kt = 0, ibuf = 0;
    set_tile(0);
    const bool bias_wave = wid == 0 && (epi & EPI_BIAS);
    auto issue_stage = [&]() {
      char* base = lds + ibuf * lcSTG;
      const uint32_t koff = static_cast<uint32_t>(i_kt) * lcRowBytes;
      // The tile's 256 bias values (1 KiB) go to LDS with its first stage: OLDER than that stage's pieces in this wave's queue, so
      // every counted wait that retires the stage retires them too.  Slot = tile parity: tile t - 2's epilogue is over before the
      // loaders reach tile t (three stages ahead of the MFMA waves at most, and a tile has >= 4 K-steps).
      if (i_kt == 0 && bias_wave)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Bt + lane * 16), (lc_lptr_t)(lds + 3 * lcSTG + (i_tile & 1) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Wt + koff + offW[i]), (lc_lptr_t)(base + (wid * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Xt + koff + offX[i]), (lc_lptr_t)(base + lcWBytes + (wid * 4 + i) * 1024), 16, 0, 0);
      ibuf = ibuf == 2 ? 0 : ibuf + 1;
      if (++i_kt == i_nk) {
        i_kt = 0;
        if (++i_tile < my_tiles) set_tile(i_tile);
      }
    };
    issue_stage();
    if (S > 1) issue_stage();
    if (S > 2) issue_stage();
    if (S > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");        // stage 0 landed; two younger stages may fly
    else if (S > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < S; ++s) {
      // in front of barrier s: stage s + 1 has landed (stage s + 2 may fly)
      if (s + 2 < S && !(ABL & 4)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // behind it nobody reads buffer s % 3 any more
      if (s + 3 < S && !(ABL & 2)) issue_stage();
    }
    return;
  }

  // ===================================================== MFMA waves ===================================================================
  const int c = wid - 4;
  const int wm = c >> 1, wn = c & 1;              // 2 (m) x 2 (n) waves of 64 x 128
  const int frow = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lc_lptr_t)lds));
  const uint32_t aW = lds_base + lc_swz(wn * 128 + frow, fq);
  const uint32_t aX = lds_base + lcWBytes + lc_swz(wm * 64 + frow, fq);

  lc_f32x4_t acc[8][4];                            // [n-fragment][m-fragment]
  lc_u32x4_t fw[8], fxa[4], fxb[4];

  // the compute side's view of its problem
  const void* residual = R0;
  void* out = O0;
  int N = N0, M = M0, nk = nk0;
  [[maybe_unused]] auto to_problem1 = [&]() {
    residual = R1; out = O1; N = N1; M = M1; nk = nk1;
  };
  if constexpr (GRP) { if (n_first == 0) to_problem1(); }
  // the wide kernel's residual-first rule: short K -> the accumulators START as the residual tile ((residual + sum) + bias), long K ->
  // the residual is added behind the bias; per GEMM, never per tile position
  // (RF is decided on the host from the same condition: EPI_RESIDUAL without an activation, and nk <= 16 for every problem of the launch)

  // fp16 residual rows in the 16-byte layout of the packed output (lane = one row x 8 consecutive n), brought to the accumulator layout
  // by v_permlane16_swap; one 16-row fragment (b) at a time: 4 loads of 16 bytes in flight per lane
  auto add_residual = [&](int m0, int n0) __attribute__((always_inline)) {
    const uint16_t* res16 = reinterpret_cast<const uint16_t*>(residual);
    const int col = n0 + wn * 128 + (fq & 1) * 16 + (fq & 2) * 4;      // + 32 * pair
#pragma unroll
    for (int bg = 0; bg < 4; bg += 2) {
      lc_u32x4_t r[2][4];
#pragma unroll
      for (int b = bg; b < bg + 2; ++b) {
        int m = m0 + wm * 64 + b * 16 + frow;
        m = m < M ? m : M - 1;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) r[b - bg][pr] = *reinterpret_cast<const lc_u32x4_t*>(res16 + static_cast<size_t>(m) * N + col + 32 * pr);
      }
#pragma unroll
      for (int b = bg; b < bg + 2; ++b) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const lc_u32x4_t qv = r[b - bg][pr];
          const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(qv[0], qv[2], false, false);
          const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(qv[1], qv[3], false, false);
          acc[2 * pr][b][0] += f16lo_to_f32(s0[0]); acc[2 * pr][b][1] += f16hi_to_f32(s0[0]);
          acc[2 * pr][b][2] += f16lo_to_f32(s1[0]); acc[2 * pr][b][3] += f16hi_to_f32(s1[0]);
          acc[2 * pr + 1][b][0] += f16lo_to_f32(s0[1]); acc[2 * pr + 1][b][1] += f16hi_to_f32(s0[1]);
          acc[2 * pr + 1][b][2] += f16lo_to_f32(s1[1]); acc[2 * pr + 1][b][3] += f16hi_to_f32(s1[1]);
        }
      }
    }
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  auto mfma = [&](const lc_u32x4_t& w, const lc_u32x4_t& x, const lc_f32x4_t& cin) __attribute__((always_inline)) {
    if constexpr (ABL & 1) {      // keep the operands alive, skip the instruction
      asm volatile("" ::"v"(w), "v"(x));
      return cin;
    } else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lc_bf16x8_t, w), __builtin_bit_cast(lc_bf16x8_t, x), cin, 0, 0, 0);
  };
#define LC_WAIT5(cnt, r0, r1, r2, r3, r4) \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)::"memory")
#define LC_WAIT1(cnt, r0) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(r0)::"memory")
#define LC_WAIT_ALLW(cnt) \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fw[4]), "+v"(fw[5]), "+v"(fw[6]), "+v"(fw[7])::"memory")

  // W fragment a / X fragment b of (buffer byte offset bo, half ks): 16-row fragments sit 2048 bytes apart, the second 32-deep half is
  // the first one's address XOR 64
#define LC_READ_W(a, addr)                                                       \
  do {                                                                           \
    if constexpr ((a) == 0) LC_READ(fw[0], addr, 0);                             \
    else if constexpr ((a) == 1) LC_READ(fw[1], addr, 2048);                     \
    else if constexpr ((a) == 2) LC_READ(fw[2], addr, 4096);                     \
    else if constexpr ((a) == 3) LC_READ(fw[3], addr, 6144);                     \
    else if constexpr ((a) == 4) LC_READ(fw[4], addr, 8192);                     \
    else if constexpr ((a) == 5) LC_READ(fw[5], addr, 10240);                    \
    else if constexpr ((a) == 6) LC_READ(fw[6], addr, 12288);                    \
    else LC_READ(fw[7], addr, 14336);                                            \
  } while (0)
#define LC_READ_X(fx, b, addr)                                                   \
  do {                                                                           \
    if constexpr ((b) == 0) LC_READ(fx[0], addr, 0);                             \
    else if constexpr ((b) == 1) LC_READ(fx[1], addr, 2048);                     \
    else if constexpr ((b) == 2) LC_READ(fx[2], addr, 4096);                     \
    else LC_READ(fx[3], addr, 6144);                                             \
  } while (0)

#ifdef LC_STAMPS
  const unsigned long long st_t0 = LC_T(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_first = 0, st_k = 0, st_e = 0, st_mark = 0;
#endif
  // ---- prologue: the first tile's residual rows (if they come first) are on their way while stage 0 lands ----------------------
  int cur = 0;
  {
    bool second; int m0, n0;
    tile_of(0, second, m0, n0);
    if constexpr (RF) { zero_acc(); add_residual(m0, n0); }
  }
  __builtin_amdgcn_s_barrier();                    // stage 0 has landed
#ifdef LC_STAMPS
  st_first = LC_T() - st_t0;
  st_mark = LC_T();
#endif
  // fragments of (stage 0, k 0..31): X then W, the order every later half-step issues them in
  LC_READ_X(fxa, 0, aX); LC_READ_X(fxa, 1, aX); LC_READ_X(fxa, 2, aX); LC_READ_X(fxa, 3, aX);
  LC_READ_W(0, aW); LC_READ_W(1, aW); LC_READ_W(2, aW); LC_READ_W(3, aW);
  LC_READ_W(4, aW); LC_READ_W(5, aW); LC_READ_W(6, aW); LC_READ_W(7, aW);

  // One K-step.  Half 0 multiplies (fxa, fw) = k 0..31 and reads k 32..63 of the same stage (fxb; fw in place); half 1 multiplies those,
  // passes the barrier after its second fragment group - every read of this stage was issued at least 8 MFMAs earlier - and reads
  // k 0..31 of the next stage (fxa; fw in place).  LDS returns a wave's reads in order, so "fragment a has landed" is a count of the
  // reads issued after it: 11 in the steady state of half 0 (7 - a older W fragments still to come, the 4 X reads and the a W reads of
  // the next half-step issued since), 7 - a at the top of half 1.
  // (Deferred stores, the wide kernel's answer to a CU that drains stores at ~10-13 B/clk - a tile's 64 KB keep the lone wave of each
  // SIMD in the epilogue for ~4 900 cycles, tools/lc_stamps.py - were built here too: eight of the sixteen 16-byte pieces per lane
  // parked in registers and issued one or two per K-step among the next tile's MFMAs.  Measured WORSE on the grouped launches
  // (QKV 68.2 -> 72.5 us; profiles/r05_j): a store in the lone MFMA wave's in-order stream waits for the address path behind the
  // stage's 48 DMA pieces, and nobody else issues MFMAs on that SIMD meanwhile.  Not kept.)
  auto kstep = [&](auto zc) __attribute__((always_inline)) {
    constexpr bool ZC = decltype(zc)::value;      // first K-step of a tile whose accumulators start at zero: C = 0 instead of 128 v_mov
    const uint32_t bo = static_cast<uint32_t>(cur) * lcSTG;
    const uint32_t w1 = (aW + bo) ^ 64u, x1 = (aX + bo) ^ 64u;
    // ---------------- half 0 ----------------
    LC_WAIT5(7, fxa[0], fxa[1], fxa[2], fxa[3], fw[0]);
    __builtin_amdgcn_sched_barrier(0);
#define LC_MFMA0(a, b) acc[a][b] = mfma(fw[a], fxa[b], ZC ? lc_f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[a][b])
#define LC_GROUP0(a)                                                                                   \
  do {                                                                                                 \
    if constexpr ((a) > 0) { LC_WAIT1(11, fw[a]); __builtin_amdgcn_sched_barrier(0); }                 \
    if constexpr ((a) == 0) LC_READ_X(fxb, 0, x1);                                                     \
    LC_MFMA0(a, 0);                                                                                    \
    if constexpr ((a) == 0) LC_READ_X(fxb, 1, x1);                                                     \
    LC_MFMA0(a, 1);                                                                                    \
    if constexpr ((a) == 0) LC_READ_X(fxb, 2, x1);                                                     \
    LC_MFMA0(a, 2);                                                                                    \
    if constexpr ((a) == 0) LC_READ_X(fxb, 3, x1);                                                     \
    LC_MFMA0(a, 3);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    LC_READ_W(a, w1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
    LC_GROUP0(0); LC_GROUP0(1); LC_GROUP0(2); LC_GROUP0(3); LC_GROUP0(4); LC_GROUP0(5); LC_GROUP0(6); LC_GROUP0(7);
#undef LC_GROUP0
#undef LC_MFMA0
    // ---------------- half 1 ----------------
    const int nxt = cur == 2 ? 0 : cur + 1;
    const uint32_t bn = static_cast<uint32_t>(nxt) * lcSTG;
    const uint32_t w0 = aW + bn, x0 = aX + bn;
#define LC_MFMA1(a, b) acc[a][b] = mfma(fw[a], fxb[b], acc[a][b])
    LC_WAIT5(7, fxb[0], fxb[1], fxb[2], fxb[3], fw[0]);
    __builtin_amdgcn_sched_barrier(0);
    LC_MFMA1(0, 0); LC_MFMA1(0, 1); LC_MFMA1(0, 2); LC_MFMA1(0, 3);
    __builtin_amdgcn_sched_barrier(0);
    LC_WAIT1(6, fw[1]);
    __builtin_amdgcn_sched_barrier(0);
    LC_MFMA1(1, 0); LC_MFMA1(1, 1); LC_MFMA1(1, 2); LC_MFMA1(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    LC_WAIT_ALLW(0);                               // every fragment of this stage is in registers
    __builtin_amdgcn_s_barrier();                  // ... in every MFMA wave; the next stage has landed
    __builtin_amdgcn_sched_barrier(0);
    // group 2 carries the X reads of the next stage and the two W fragments whose MFMAs ran in front of the barrier
    LC_READ_X(fxa, 0, x0);
    LC_MFMA1(2, 0);
    LC_READ_X(fxa, 1, x0);
    LC_MFMA1(2, 1);
    LC_READ_X(fxa, 2, x0);
    LC_MFMA1(2, 2);
    LC_READ_X(fxa, 3, x0);
    LC_MFMA1(2, 3);
    __builtin_amdgcn_sched_barrier(0);
    LC_READ_W(0, w0);
    LC_READ_W(1, w0);
    LC_READ_W(2, w0);
    __builtin_amdgcn_sched_barrier(0);
#define LC_GROUP1(a)                                                                                   \
  do {                                                                                                 \
    LC_MFMA1(a, 0); LC_MFMA1(a, 1); LC_MFMA1(a, 2); LC_MFMA1(a, 3);                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    LC_READ_W(a, w0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
    LC_GROUP1(3); LC_GROUP1(4); LC_GROUP1(5); LC_GROUP1(6); LC_GROUP1(7);
#undef LC_GROUP1
#undef LC_MFMA1
    cur = nxt;
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    bool second; int m0, n0;
    tile_of(ti, second, m0, n0);
    if constexpr (GRP) { if (ti == n_first && ti > 0) to_problem1(); }
    // (zeroed with 128 v_mov here rather than by C = 0 in a peeled first K-step: a second copy of the K-step body leaves the allocator
    // two register assignments to reconcile at every tile boundary, and it does so through scratch)
    if constexpr (RF) { if (ti > 0) { zero_acc(); add_residual(m0, n0); } }
    else zero_acc();
    for (int kt = 0; kt < nk; ++kt) kstep(std::false_type{});

    // ---- epilogue (the loaders are already staging the next tile; its first fragments are on their way into fxa / fw) -------------
    // One wave per SIMD: nobody covers a memory round trip or a taken branch here.  The bias row comes from LDS (staged by the loaders
    // with the tile's first stage): eight reads, ONE wait; the output kind is a template parameter; a full tile's stores carry no
    // lane mask; a lane's store address is a uniform base + one 32-bit offset.
#ifdef LC_STAMPS
    { const unsigned long long t_ = LC_T(); st_k += t_ - st_mark; st_mark = t_; }
#endif
    if (epi & EPI_BIAS) {
      const uint32_t ab = lds_base + 3 * lcSTG + (ti & 1) * 1024 + (wn * 128 + fq * 4) * 4;
      lc_f32x4_t bv[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[a]) : "v"(ab), "n"(a * 64));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7])::"memory");
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += bv[a];
    }
    if (epi & EPI_QUICKGELU) {
      // two values at a time, the wide kernel's instruction sequence (packed-f32 multiply / add, v_exp_f32 / v_rcp_f32 per value)
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const lc_f32x2_t v = {acc[a][b][j], acc[a][b][j + 1]};
            const lc_f32x2_t t = v * lc_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
            const lc_f32x2_t d = lc_f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + lc_f32x2_t{1.0f, 1.0f};
            const lc_f32x2_t o = v * lc_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            acc[a][b][j] = o[0];
            acc[a][b][j + 1] = o[1];
          }
    }
    if constexpr (RES == 2) add_residual(m0, n0);
    __builtin_amdgcn_sched_barrier(0);
    {
      // v_permlane16_swap exchanges, between the lane pairs (l, l + 16), the packed words of two neighbouring n-fragments: an even
      // lane-row then owns 8 consecutive n of fragment 2 pr and an odd lane-row 8 consecutive n of fragment 2 pr + 1 -> 16-byte stores
      char* const obase = static_cast<char*>(out) + (static_cast<size_t>(m0) * N + n0) * 2;       // uniform
      const uint32_t ldn = static_cast<uint32_t>(N) * 2;
      const uint32_t off0 = static_cast<uint32_t>(wm * 64 + frow) * ldn + static_cast<uint32_t>(wn * 128 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
      const bool full = m0 + lcBM <= M && !(epi & 256);      // uniform (256 = timing-only ablation: skip stores)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t offb = off0 + static_cast<uint32_t>(b * 16) * ldn;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          uint32_t lo[2], hi[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            if constexpr (F16O) {
              lo[w] = pack_f16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_f16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            } else {
              lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            }
          }
          const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
          const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
          const lc_u32x4_t val = {s0[0], s1[0], s0[1], s1[1]};
          if (full) *reinterpret_cast<lc_u32x4_t*>(obase + (offb + pr * 64)) = val;
          else if (m0 + wm * 64 + b * 16 + frow < M && !(epi & 256)) *reinterpret_cast<lc_u32x4_t*>(obase + (offb + pr * 64)) = val;
        }
      }
    }
#ifdef LC_STAMPS
    { const unsigned long long t_ = LC_T(); st_e += t_ - st_mark; st_mark = t_; }
#endif
  }
#ifdef LC_STAMPS
  if (c == 0 && lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_lc_stamps + blockIdx.x * 8;
    o[0] = LC_T() - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0; o[2] = st_first; o[3] = st_k; o[4] = st_e;
    o[5] = static_cast<unsigned long long>(S); o[6] = static_cast<unsigned long long>(my_tiles); o[7] = static_cast<unsigned long long>(my_tiles);
  }
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads issued past the last stage
}

// ---- "lc2": the same split with TWO MFMA waves per SIMD ---------------------------------------------------------------------------------
// What the lc kernel lacks is somebody to multiply while a SIMD's MFMA wave is in its epilogue, and somebody to cover the issue of a
// store.  Here a workgroup has 12 waves, 3 per SIMD (168 VGPRs): waves 0..3 stage exactly as above, waves 4..11 multiply 64 x 64 each
// (2 (m) x 4 (n) of the same 128 x 256 tile; the two MFMA waves of a SIMD share their W columns): 32 MFMAs per wave and K-step, 4 W
// fragments refilled in place, X double-buffered: 64 accumulators + 48 fragment registers + the previous tile's 32 packed output
// registers, which leave one 16-byte store per K-step under the next tile's MFMAs (the wide kernel's deferred stores - there is a
// partner wave to cover their issue now).  The residual forms use the same 32 registers a second time: once the parked pieces have
// left (two per K-step: four K-steps) the fp16 residual rows are fetched INTO them - RES 1 (short K, the residual comes first):
// the NEXT tile's, and the epilogue rotates piece by piece (pack this tile's piece, start the next tile's accumulators from the
// residual piece, park the packed piece where it lay); RES 2 (long K): this tile's own, added behind the bias - so no tile ever
// waits a memory round trip for its residual.  Same k order per output element, same epilogue order: the same bits.
template <bool GRP, int RES, bool F16O>
__global__ __launch_bounds__(768) void gemm_lc2_kernel(LcProblem p0, LcProblem p1, int epi) {
  __shared__ __attribute__((aligned(1024))) char lds[3 * lcSTG + 2 * 1024];      // the ring + two bias rows (tile parity)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  const char* const X0 = p0.X; const char* const W0 = p0.W; const float* const B0 = p0.bias; void* const O0 = p0.out;
  const void* const R0 = p0.residual;
  const int N0 = p0.N, K0 = p0.K;
  const char* const X1 = p1.X; const char* const W1 = p1.W; const float* const B1 = p1.bias; void* const O1 = p1.out;
  const void* const R1 = p1.residual;
  const int N1 = p1.N, K1 = p1.K;
  int M0 = p0.Mub;
  if (p0.m_dev) { const int md = *p0.m_dev; M0 = md < M0 ? md : M0; }
  M0 = __builtin_amdgcn_readfirstlane(M0);
  int M1 = 0;
  if constexpr (GRP) {
    M1 = p1.Mub;
    if (p1.m_dev) { const int md = *p1.m_dev; M1 = md < M1 ? md : M1; }
    M1 = __builtin_amdgcn_readfirstlane(M1);
  }
  const int tiles_n0 = N0 / lcBN, tiles_n1 = GRP ? N1 / lcBN : 1;
  const int total0 = tiles_n0 * ((M0 + lcBM - 1) / lcBM);
  const int total1 = GRP ? tiles_n1 * ((M1 + lcBM - 1) / lcBM) : 0;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q0 = total0 >> 3, r0 = total0 & 7, q1 = total1 >> 3, r1 = total1 & 7;
  const int lo0 = xcd < r0 ? xcd * (q0 + 1) : r0 * (q0 + 1) + (xcd - r0) * q0, len0 = xcd < r0 ? q0 + 1 : q0;
  const int lo1 = xcd < r1 ? xcd * (q1 + 1) : r1 * (q1 + 1) + (xcd - r1) * q1, len1 = xcd < r1 ? q1 + 1 : q1;
  const int n_first = slot < len0 ? (len0 - slot + per - 1) / per : 0;
  const int span = len0 + len1;
  const int my_tiles = slot < span ? (span - slot + per - 1) / per : 0;
  if (my_tiles == 0) return;
  const int nk0 = K0 / 64, nk1 = GRP ? K1 / 64 : 0;
  const int S = n_first * nk0 + (my_tiles - n_first) * nk1;
  auto tile_of = [&](int ti, bool& second, int& m0, int& n0) {
    const int j = slot + ti * per;
    second = GRP && ti >= n_first;
    const int logical = second ? lo1 + (j - len0) : lo0 + j;
    const int tn_cnt = second ? tiles_n1 : tiles_n0;
    const int tm = logical / tn_cnt;
    m0 = tm * lcBM;
    n0 = (logical - tm * tn_cnt) * lcBN;
  };

  if (wid < 4) {
    // ---- loader waves: the lc kernel's, piece for piece ----
    const int sub = lane >> 3, ch = lane & 7;
    uint32_t offW[8], offX[4];
    const char* Wt = nullptr;
    const char* Xt = nullptr;
    const char* Bt = nullptr;
    int i_nk = 0;
    int w_prob = -1;
    auto set_tile = [&](int ti) {
      bool second; int m0, n0;
      tile_of(ti, second, m0, n0);
      const uint32_t rs = static_cast<uint32_t>(second ? K1 : K0) * 2;
      if (w_prob != static_cast<int>(second)) {
        w_prob = static_cast<int>(second);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = (wid * 8 + i) * 8 + sub;
          offW[i] = static_cast<uint32_t>(row) * rs + ((ch ^ (row & 7)) << 4);
        }
      }
      i_nk = second ? nk1 : nk0;
      const int Mp = second ? M1 : M0;
      Wt = (second ? W1 : W0) + static_cast<size_t>(n0) * rs;
      Xt = second ? X1 : X0;
      Bt = reinterpret_cast<const char*>((second ? B1 : B0) + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + sub;
        int xr = m0 + row;
        xr = xr < Mp ? xr : Mp - 1;
        offX[i] = static_cast<uint32_t>(xr) * rs + ((ch ^ (row & 7)) << 4);
      }
    };
    int i_tile = 0, i_kt = 0, ibuf = 0;
    set_tile(0);
    const bool bias_wave = wid == 0 && (epi & EPI_BIAS);
    auto issue_stage = [&]() {
      char* base = lds + ibuf * lcSTG;
      const uint32_t koff = static_cast<uint32_t>(i_kt) * lcRowBytes;
      if (i_kt == 0 && bias_wave)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Bt + lane * 16), (lc_lptr_t)(lds + 3 * lcSTG + (i_tile & 1) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Wt + koff + offW[i]), (lc_lptr_t)(base + (wid * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Xt + koff + offX[i]), (lc_lptr_t)(base + lcWBytes + (wid * 4 + i) * 1024), 16, 0, 0);
      ibuf = ibuf == 2 ? 0 : ibuf + 1;
      if (++i_kt == i_nk) {
        i_kt = 0;
        if (++i_tile < my_tiles) set_tile(i_tile);
      }
    };
    issue_stage();
    if (S > 1) issue_stage();
    if (S > 2) issue_stage();
    if (S > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < S; ++s) {
      if (s + 2 < S) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 3 < S) issue_stage();
    }
    return;
  }

  // ---- MFMA waves ----
  const int c = wid - 4;
  const int wm = c >> 2, wn = c & 3;              // 2 (m) x 4 (n) waves of 64 x 64; the SIMD of wave c is c & 3: its two waves share wn
  const int frow = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lc_lptr_t)lds));
  const uint32_t aW = lds_base + lc_swz(wn * 64 + frow, fq);
  const uint32_t aX = lds_base + lcWBytes + lc_swz(wm * 64 + frow, fq);

  lc_f32x4_t acc[4][4];                            // [n-fragment][m-fragment]
  lc_u32x4_t fw[4], fxa[4], fxb[4];
  lc_u32x4_t pend[8];                              // the previous tile's packed outputs: piece (b, pr) at 2 b + pr
  int pend_idx = 8;                                // next parked piece to store (8: none)
  char* pend_base = nullptr;
  uint32_t pend_off0 = 0, pend_ldn = 0;
  [[maybe_unused]] int res_kt = -1, res_tile = -1;      // RES != 0: at K-step res_kt (-1: never) the residual rows of tile res_tile are fetched into pend

  void* out = O0;
  int N = N0, M = M0, nk = nk0;
  [[maybe_unused]] auto to_problem1 = [&]() { out = O1; N = N1; M = M1; nk = nk1; };
  if constexpr (GRP) { if (n_first == 0) to_problem1(); }

  auto mfma = [&](const lc_u32x4_t& w, const lc_u32x4_t& x, const lc_f32x4_t& cin) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lc_bf16x8_t, w), __builtin_bit_cast(lc_bf16x8_t, x), cin, 0, 0, 0);
  };
  auto store_pending = [&](int idx) __attribute__((always_inline)) {      // idx is wave-uniform: a jump, not a select
    char* p = pend_base + (pend_off0 + static_cast<uint32_t>(idx >> 1) * 16u * pend_ldn + static_cast<uint32_t>(idx & 1) * 64u);
    switch (idx) {
      case 0: *reinterpret_cast<lc_u32x4_t*>(p) = pend[0]; break;
      case 1: *reinterpret_cast<lc_u32x4_t*>(p) = pend[1]; break;
      case 2: *reinterpret_cast<lc_u32x4_t*>(p) = pend[2]; break;
      case 3: *reinterpret_cast<lc_u32x4_t*>(p) = pend[3]; break;
      case 4: *reinterpret_cast<lc_u32x4_t*>(p) = pend[4]; break;
      case 5: *reinterpret_cast<lc_u32x4_t*>(p) = pend[5]; break;
      case 6: *reinterpret_cast<lc_u32x4_t*>(p) = pend[6]; break;
      default: *reinterpret_cast<lc_u32x4_t*>(p) = pend[7]; break;
    }
  };
  // the fp16 residual rows of tile `ti` into the (free) parked-piece registers, in the packed 16-byte layout of the output
  [[maybe_unused]] auto fetch_residual = [&](int ti) __attribute__((always_inline)) {
    bool second; int m0, n0;
    tile_of(ti, second, m0, n0);
    const uint16_t* res16 = reinterpret_cast<const uint16_t*>(second ? R1 : R0);
    const int Nn = second ? N1 : N0, Mp = second ? M1 : M0;
    const int col = n0 + wn * 64 + (fq & 1) * 16 + (fq & 2) * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int m = m0 + wm * 64 + b * 16 + frow;
      m = m < Mp ? m : Mp - 1;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) pend[2 * b + pr] = *reinterpret_cast<const lc_u32x4_t*>(res16 + static_cast<size_t>(m) * Nn + col + 32 * pr);
    }
  };
  // piece (b, pr) of a fetched residual, brought to the accumulator layout (v_permlane16_swap) and ADDED to the two n-fragments it covers
  [[maybe_unused]] auto add_piece = [&](int b, int pr, const lc_u32x4_t& qv) __attribute__((always_inline)) {
    const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(qv[0], qv[2], false, false);
    const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(qv[1], qv[3], false, false);
    acc[2 * pr][b][0] += f16lo_to_f32(s0[0]); acc[2 * pr][b][1] += f16hi_to_f32(s0[0]);
    acc[2 * pr][b][2] += f16lo_to_f32(s1[0]); acc[2 * pr][b][3] += f16hi_to_f32(s1[0]);
    acc[2 * pr + 1][b][0] += f16lo_to_f32(s0[1]); acc[2 * pr + 1][b][1] += f16hi_to_f32(s0[1]);
    acc[2 * pr + 1][b][2] += f16lo_to_f32(s1[1]); acc[2 * pr + 1][b][3] += f16hi_to_f32(s1[1]);
  };
#define L2_WAIT5(cnt, r0, r1, r2, r3, r4) \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)::"memory")
#define L2_WAIT1(cnt, r0) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(r0)::"memory")
#define L2_WAIT_ALLW(cnt) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3])::"memory")
#define L2_READ_W(a, addr)                                                       \
  do {                                                                           \
    if constexpr ((a) == 0) LC_READ(fw[0], addr, 0);                             \
    else if constexpr ((a) == 1) LC_READ(fw[1], addr, 2048);                     \
    else if constexpr ((a) == 2) LC_READ(fw[2], addr, 4096);                     \
    else LC_READ(fw[3], addr, 6144);                                             \
  } while (0)
#define L2_READ_X(fx, b, addr)                                                   \
  do {                                                                           \
    if constexpr ((b) == 0) LC_READ(fx[0], addr, 0);                             \
    else if constexpr ((b) == 1) LC_READ(fx[1], addr, 2048);                     \
    else if constexpr ((b) == 2) LC_READ(fx[2], addr, 4096);                     \
    else LC_READ(fx[3], addr, 6144);                                             \
  } while (0)

  int cur = 0;
  if constexpr (RES == 1) {                        // the first tile's accumulators start as its residual rows (0 + r: the wide kernel's bits)
    fetch_residual(0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) add_piece(b, pr, pend[2 * b + pr]);
  }
  __builtin_amdgcn_s_barrier();                    // stage 0 has landed
  L2_READ_X(fxa, 0, aX); L2_READ_X(fxa, 1, aX); L2_READ_X(fxa, 2, aX); L2_READ_X(fxa, 3, aX);
  L2_READ_W(0, aW); L2_READ_W(1, aW); L2_READ_W(2, aW); L2_READ_W(3, aW);

  // One K-step.  LDS returns a wave's reads in order; at the top of half 0 the outstanding ones are X''0..3, W''0, W''1, W''2, W''3
  // (issued in half 1 of the K-step before): fragment W''0 has landed when 3 younger reads may still fly; W''a (a >= 1) when 7 may
  // (the rest of the W'' plus the four X' and the a W' reads issued since).  Half 1: X'0..3, W'0 | W'1 | W'2 | W'3 -> 3, 2, then all.
  auto kstep = [&](int kt) __attribute__((always_inline)) {
    const uint32_t bo = static_cast<uint32_t>(cur) * lcSTG;
    const uint32_t w1 = (aW + bo) ^ 64u, x1 = (aX + bo) ^ 64u;
    L2_WAIT5(3, fxa[0], fxa[1], fxa[2], fxa[3], fw[0]);
    __builtin_amdgcn_sched_barrier(0);
#define L2_MFMA0(a, b) acc[a][b] = mfma(fw[a], fxa[b], acc[a][b])
#define L2_GROUP0(a)                                                                                   \
  do {                                                                                                 \
    if constexpr ((a) > 0) { L2_WAIT1(7, fw[a]); __builtin_amdgcn_sched_barrier(0); }                  \
    if constexpr ((a) == 0) L2_READ_X(fxb, 0, x1);                                                     \
    L2_MFMA0(a, 0);                                                                                    \
    if constexpr ((a) == 0) L2_READ_X(fxb, 1, x1);                                                     \
    L2_MFMA0(a, 1);                                                                                    \
    if constexpr ((a) == 0) L2_READ_X(fxb, 2, x1);                                                     \
    L2_MFMA0(a, 2);                                                                                    \
    if constexpr ((a) == 0) L2_READ_X(fxb, 3, x1);                                                     \
    L2_MFMA0(a, 3);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    L2_READ_W(a, w1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
    L2_GROUP0(0); L2_GROUP0(1);
    // (pend_idx / the K-step counter are kept OPAQUE to the optimizer: knowing "kt < 4" it peels four copies of this body, hoists
    // the residual addresses across them through scratch and - measured - gets the first row block's address wrong)
    if (pend_idx < 8) {                               // two 16-byte pieces of the previous tile per K-step: gone after four
      store_pending(pend_idx); store_pending(pend_idx + 1);
      pend_idx += 2;
      asm volatile("" : "+s"(pend_idx));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (RES != 0) {
      if (kt == res_kt) { fetch_residual(res_tile); __builtin_amdgcn_sched_barrier(0); }
    }
    L2_GROUP0(2); L2_GROUP0(3);
#undef L2_GROUP0
#undef L2_MFMA0
    const int nxt = cur == 2 ? 0 : cur + 1;
    const uint32_t bn = static_cast<uint32_t>(nxt) * lcSTG;
    const uint32_t w0 = aW + bn, x0 = aX + bn;
#define L2_MFMA1(a, b) acc[a][b] = mfma(fw[a], fxb[b], acc[a][b])
    L2_WAIT5(3, fxb[0], fxb[1], fxb[2], fxb[3], fw[0]);
    __builtin_amdgcn_sched_barrier(0);
    L2_MFMA1(0, 0); L2_MFMA1(0, 1); L2_MFMA1(0, 2); L2_MFMA1(0, 3);
    __builtin_amdgcn_sched_barrier(0);
    L2_WAIT1(2, fw[1]);
    __builtin_amdgcn_sched_barrier(0);
    L2_MFMA1(1, 0); L2_MFMA1(1, 1); L2_MFMA1(1, 2); L2_MFMA1(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    L2_WAIT_ALLW(0);                               // every fragment of this stage is in registers
    __builtin_amdgcn_s_barrier();                  // ... in every MFMA wave; the next stage has landed
    __builtin_amdgcn_sched_barrier(0);
    L2_READ_X(fxa, 0, x0);
    L2_MFMA1(2, 0);
    L2_READ_X(fxa, 1, x0);
    L2_MFMA1(2, 1);
    L2_READ_X(fxa, 2, x0);
    L2_MFMA1(2, 2);
    L2_READ_X(fxa, 3, x0);
    L2_MFMA1(2, 3);
    __builtin_amdgcn_sched_barrier(0);
    L2_READ_W(0, w0);
    L2_READ_W(1, w0);
    L2_READ_W(2, w0);
    __builtin_amdgcn_sched_barrier(0);
    L2_MFMA1(3, 0); L2_MFMA1(3, 1); L2_MFMA1(3, 2); L2_MFMA1(3, 3);
    __builtin_amdgcn_sched_barrier(0);
    L2_READ_W(3, w0);
    __builtin_amdgcn_sched_barrier(0);
#undef L2_MFMA1
    cur = nxt;
  };

  for (int ti = 0; ti < my_tiles; ++ti) {
    bool second; int m0, n0;
    tile_of(ti, second, m0, n0);
    if constexpr (GRP) { if (ti == n_first && ti > 0) to_problem1(); }
    if constexpr (RES != 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (RES == 1) { res_tile = ti + 1; res_kt = ti + 1 < my_tiles ? 4 : -1; }      // the NEXT tile's rows, once the parked pieces are out
    // this tile's own rows, at the same K-step: the registers are free from there on.  (Fetched at K-step nk - 3 instead, ONE build of
    // this kernel returned garbage in the first row block of some lanes, run to run; the same source rebuilt with a diagnostic macro
    // beside it - and nk - 6 / - 2 / - 1 - did not.  Not reproduced since, not understood: tools/lc2_stress.py hammers the kept form.)
    if constexpr (RES == 2) { res_tile = ti; res_kt = 4; }
    if constexpr (RES != 0) asm volatile("" : "+s"(res_kt));
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("" : "+s"(kt));
      kstep(kt);
    }
    // nk >= 8 (host): every parked piece has left after four K-steps

    if (epi & EPI_BIAS) {
      const uint32_t ab = lds_base + 3 * lcSTG + (ti & 1) * 1024 + (wn * 64 + fq * 4) * 4;
      lc_f32x4_t bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[a]) : "v"(ab), "n"(a * 64));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3])::"memory");
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] += bv[a];
    }
    if (epi & EPI_QUICKGELU) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const lc_f32x2_t v = {acc[a][b][j], acc[a][b][j + 1]};
            const lc_f32x2_t t = v * lc_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
            const lc_f32x2_t d = lc_f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + lc_f32x2_t{1.0f, 1.0f};
            const lc_f32x2_t o = v * lc_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            acc[a][b][j] = o[0];
            acc[a][b][j + 1] = o[1];
          }
    }
    if constexpr (RES == 2) {                        // long K: the residual behind the bias (fetched three K-steps ago)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3]), "+v"(pend[4]), "+v"(pend[5]), "+v"(pend[6]), "+v"(pend[7])::"memory");
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) add_piece(b, pr, pend[2 * b + pr]);
    }
    if constexpr (RES == 1) {
      if (res_kt >= 0)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3]), "+v"(pend[4]), "+v"(pend[5]), "+v"(pend[6]), "+v"(pend[7])::"memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      char* const obase = static_cast<char*>(out) + (static_cast<size_t>(m0) * N + n0) * 2;       // uniform
      const uint32_t ldn = static_cast<uint32_t>(N) * 2;
      const uint32_t off0 = static_cast<uint32_t>(wm * 64 + frow) * ldn + static_cast<uint32_t>(wn * 64 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint32_t lo[2], hi[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            if constexpr (F16O) {
              lo[w] = pack_f16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_f16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            } else {
              lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
              hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
            }
          }
          const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
          const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
          if constexpr (RES == 1) {                  // the rotation: this piece's accumulators restart as the next tile's residual piece
            const lc_u32x4_t nextres = pend[2 * b + pr];
            acc[2 * pr][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[2 * pr + 1][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (res_kt >= 0) add_piece(b, pr, nextres);
          }
          pend[2 * b + pr] = lc_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
        }
      pend_base = obase; pend_off0 = off0; pend_ldn = ldn;
      const bool full = m0 + lcBM <= M;
      if (full && ti + 1 < my_tiles && !(epi & 256)) {
        pend_idx = 0;                              // the next tile's first four K-steps carry the pieces out
        asm volatile("" : "+s"(pend_idx));
      } else if (!(epi & 256)) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
            if (m0 + wm * 64 + b * 16 + frow < M)
              *reinterpret_cast<lc_u32x4_t*>(obase + (off0 + static_cast<uint32_t>(b * 16) * ldn + pr * 64)) = pend[2 * b + pr];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads issued past the last stage
#undef L2_WAIT5
#undef L2_WAIT1
#undef L2_WAIT_ALLW
#undef L2_READ_W
#undef L2_READ_X
}

int gemm_lc_mode();
// ---- "lc2q": the 12-wave form on e4m3 operands (the fp8 mode, BASELINE configs[4]) -------------------------------------------------------
// A K-step of fp8 operands moves the same 48 KB as a bf16 one for half as many MFMA cycles per k (v_mfma_scale_f32_16x16x128_f8f6f4:
// one instruction per 16 x 16 x 128), so a K = 768 tile has only SIX K-steps and the wide kernel's fp8 launches are dominated by what
// is not the loop: DMA issue inside the MFMA streams (half as many MFMAs to hide it under) and the tile switch.  Here: the 4 staging
// waves of lc2, 8 MFMA waves of 64 x 64 whose fragments (8 registers each: both 16-byte halves of a lane's 32 k) are refilled in place.
// First form: bias + dequantisation, bf16 output, no residual, direct stores.  Same operand registers per MFMA as the wide kernel's
// fp8 K-step (chunks fq and fq + 4 of the 128-byte row), same epilogue expression: the same bits.
template <bool GRP>
__global__ __launch_bounds__(768) void gemm_lc2q_kernel(LcProblem p0, LcProblem p1, int epi) {
  static_assert(!GRP, "first form: plain launches");
  __shared__ __attribute__((aligned(1024))) char lds[3 * lcSTG];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const X0 = p0.X; const char* const W0 = p0.W; const float* const B0 = p0.bias; void* const O0 = p0.out;
  const float* const CS0 = p0.colscale; const float alpha0 = p0.alpha;
  const int N0 = p0.N, K0 = p0.K;
  int M0 = p0.Mub;
  if (p0.m_dev) { const int md = *p0.m_dev; M0 = md < M0 ? md : M0; }
  M0 = __builtin_amdgcn_readfirstlane(M0);
  const int tiles_n0 = N0 / lcBN;
  const int total0 = tiles_n0 * ((M0 + lcBM - 1) / lcBM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q0 = total0 >> 3, r0 = total0 & 7;
  const int lo0 = xcd < r0 ? xcd * (q0 + 1) : r0 * (q0 + 1) + (xcd - r0) * q0, len0 = xcd < r0 ? q0 + 1 : q0;
  const int my_tiles = slot < len0 ? (len0 - slot + per - 1) / per : 0;
  if (my_tiles == 0) return;
  const int nk0 = K0 / 128;                        // 128 k per K-step (rows of 128 bytes)
  const int S = my_tiles * nk0;
  auto tile_of = [&](int ti, int& m0, int& n0) {
    const int logical = lo0 + slot + ti * per;
    const int tm = logical / tiles_n0;
    m0 = tm * lcBM;
    n0 = (logical - tm * tiles_n0) * lcBN;
  };
  if (wid < 4) {
    // ---- loader waves: the lc kernel's (a row of either operand is 128 bytes per K-step, whatever the element type) ----
    const int sub = lane >> 3, ch = lane & 7;
    uint32_t offW[8], offX[4];
    const char* Wt = nullptr;
    const uint32_t rs = static_cast<uint32_t>(K0);      // bytes per row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wid * 8 + i) * 8 + sub;
      offW[i] = static_cast<uint32_t>(row) * rs + ((ch ^ (row & 7)) << 4);
    }
    auto set_tile = [&](int ti) {
      int m0, n0;
      tile_of(ti, m0, n0);
      Wt = W0 + static_cast<size_t>(n0) * rs;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (wid * 4 + i) * 8 + sub;
        int xr = m0 + row;
        xr = xr < M0 ? xr : M0 - 1;
        offX[i] = static_cast<uint32_t>(xr) * rs + ((ch ^ (row & 7)) << 4);
      }
    };
    int i_tile = 0, i_kt = 0, ibuf = 0;
    set_tile(0);
    auto issue_stage = [&]() {
      char* base = lds + ibuf * lcSTG;
      const uint32_t koff = static_cast<uint32_t>(i_kt) * lcRowBytes;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(Wt + koff + offW[i]), (lc_lptr_t)(base + (wid * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((lc_gptr_t)(X0 + koff + offX[i]), (lc_lptr_t)(base + lcWBytes + (wid * 4 + i) * 1024), 16, 0, 0);
      ibuf = ibuf == 2 ? 0 : ibuf + 1;
      if (++i_kt == nk0) {
        i_kt = 0;
        if (++i_tile < my_tiles) set_tile(i_tile);
      }
    };
    issue_stage();
    if (S > 1) issue_stage();
    if (S > 2) issue_stage();
    if (S > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < S; ++s) {
      if (s + 2 < S) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 3 < S) issue_stage();
    }
    return;
  }
  // ---- MFMA waves ----
  const int c = wid - 4;
  const int wm = c >> 2, wn = c & 3;
  const int frow = lane & 15, fq = lane >> 4;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lc_lptr_t)lds));
  const uint32_t aW = lds_base + lc_swz(wn * 64 + frow, fq);
  const uint32_t aX = lds_base + lcWBytes + lc_swz(wm * 64 + frow, fq);
  typedef __attribute__((ext_vector_type(8))) int lc_i32x8_t;
  lc_f32x4_t acc[4][4];
  lc_u32x4_t fw[4][2], fx[4][2];                   // [fragment][16-byte half: chunk fq, chunk fq + 4]
  // 64 accumulators + 64 fragment registers leave room for HALF of a tile's packed outputs (16 registers: row blocks 2 and 3); they
  // leave two per K-step under the next tile's first two K-steps, row blocks 0 and 1 are stored by the epilogue itself
  lc_u32x4_t pend[4];
  int pend_idx = 4;
  char* pend_base = nullptr;
  uint32_t pend_off0 = 0, pend_ldn = 0;
  auto store_pending = [&](int idx) __attribute__((always_inline)) {
    char* p = pend_base + (pend_off0 + static_cast<uint32_t>(2 + (idx >> 1)) * 16u * pend_ldn + static_cast<uint32_t>(idx & 1) * 64u);
    switch (idx) {
      case 0: *reinterpret_cast<lc_u32x4_t*>(p) = pend[0]; break;
      case 1: *reinterpret_cast<lc_u32x4_t*>(p) = pend[1]; break;
      case 2: *reinterpret_cast<lc_u32x4_t*>(p) = pend[2]; break;
      default: *reinterpret_cast<lc_u32x4_t*>(p) = pend[3]; break;
    }
  };
  auto mfma8 = [&](int a, int b) __attribute__((always_inline)) {
    lc_i32x8_t va, vb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      va[j] = static_cast<int>(fw[a][0][j]); va[4 + j] = static_cast<int>(fw[a][1][j]);
      vb[j] = static_cast<int>(fx[b][0][j]); vb[4 + j] = static_cast<int>(fx[b][1][j]);
    }
    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  };
#define Q_READ2(dst, addr, off)                                                                                   \
  do {                                                                                                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(addr), "n"(off));                          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"((addr) ^ 64u), "n"(off));                  \
  } while (0)
#define Q_WAIT_ALL()                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                             \
               : "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[1][0]), "+v"(fw[1][1]), "+v"(fw[2][0]), "+v"(fw[2][1]), "+v"(fw[3][0]), "+v"(fw[3][1]), \
                 "+v"(fx[0][0]), "+v"(fx[0][1]), "+v"(fx[1][0]), "+v"(fx[1][1]), "+v"(fx[2][0]), "+v"(fx[2][1]), "+v"(fx[3][0]), "+v"(fx[3][1])::"memory")
  int cur = 0;
  __builtin_amdgcn_s_barrier();                    // stage 0 has landed
  Q_READ2(fx[0], aX, 0); Q_READ2(fx[1], aX, 2048); Q_READ2(fx[2], aX, 4096); Q_READ2(fx[3], aX, 6144);
  Q_READ2(fw[0], aW, 0); Q_READ2(fw[1], aW, 2048); Q_READ2(fw[2], aW, 4096); Q_READ2(fw[3], aW, 6144);
  // One K-step: every fragment of the stage is in registers (read during the K-step before), the barrier says so for every MFMA wave
  // and that the next stage has landed; then 16 MFMAs, each fragment refilled from the next stage as soon as its last MFMA has issued.
  auto kstep = [&]() __attribute__((always_inline)) {
    Q_WAIT_ALL();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = cur == 2 ? 0 : cur + 1;
    const uint32_t bn = static_cast<uint32_t>(nxt) * lcSTG;
    const uint32_t w0 = aW + bn, x0 = aX + bn;
    mfma8(0, 0); mfma8(0, 1); mfma8(0, 2); mfma8(0, 3);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fw[0], w0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(1, 0); mfma8(1, 1); mfma8(1, 2); mfma8(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    if (pend_idx < 4) {
      store_pending(pend_idx); store_pending(pend_idx + 1);
      pend_idx += 2;
      asm volatile("" : "+s"(pend_idx));
      __builtin_amdgcn_sched_barrier(0);
    }
    Q_READ2(fw[1], w0, 2048);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(2, 0); mfma8(2, 1); mfma8(2, 2); mfma8(2, 3);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fw[2], w0, 4096);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(3, 0);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fx[0], x0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fx[1], x0, 2048);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(3, 2);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fx[2], x0, 4096);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(3, 3);
    __builtin_amdgcn_sched_barrier(0);
    Q_READ2(fx[3], x0, 6144);
    Q_READ2(fw[3], w0, 6144);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  };
  for (int ti = 0; ti < my_tiles; ++ti) {
    int m0, n0;
    tile_of(ti, m0, n0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk0; ++kt) kstep();
    // dequantisation fused with the bias: acc * (alpha * colscale[n]) + bias[n]  (the wide kernel's expression)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const lc_f32x4_t cs = *reinterpret_cast<const lc_f32x4_t*>(CS0 + n0 + wn * 64 + a * 16 + fq * 4) * alpha0;
      const lc_f32x4_t bv = (epi & EPI_BIAS) ? *reinterpret_cast<const lc_f32x4_t*>(B0 + n0 + wn * 64 + a * 16 + fq * 4) : lc_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = acc[a][b] * cs + bv;
    }
    char* const obase = static_cast<char*>(O0) + (static_cast<size_t>(m0) * N0 + n0) * 2;
    const uint32_t ldn = static_cast<uint32_t>(N0) * 2;
    const uint32_t off0 = static_cast<uint32_t>(wm * 64 + frow) * ldn + static_cast<uint32_t>(wn * 64 + (fq & 1) * 16 + (fq & 2) * 4) * 2;
    const bool park = m0 + lcBM <= M0 && ti + 1 < my_tiles && !(epi & 256);      // nk >= 4 (host): both parked K-steps exist
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          lo[w] = pack_bf16x2(acc[2 * pr][b][2 * w], acc[2 * pr][b][2 * w + 1]);
          hi[w] = pack_bf16x2(acc[2 * pr + 1][b][2 * w], acc[2 * pr + 1][b][2 * w + 1]);
        }
        const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
        const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
        const lc_u32x4_t val = {s0[0], s1[0], s0[1], s1[1]};
        if (b >= 2) pend[(b - 2) * 2 + pr] = val;
        if ((b < 2 || !park) && m0 + wm * 64 + b * 16 + frow < M0 && !(epi & 256))
          *reinterpret_cast<lc_u32x4_t*>(obase + (off0 + static_cast<uint32_t>(b * 16) * ldn + pr * 64)) = val;
      }
    if (park) {
      pend_base = obase; pend_off0 = off0; pend_ldn = ldn;
      pend_idx = 0;
      asm volatile("" : "+s"(pend_idx));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef Q_READ2
#undef Q_WAIT_ALL
}

int launch_gemm_lc2q(const GemmProblem& g, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  if (static_cast<size_t>(g.M) * g.K >= (1ull << 32)) return fail(CMH_ERR_INVALID, "gemm (lc2q): operand exceeds the 32-bit offset range");
  LcProblem P0{static_cast<const char*>(g.A), static_cast<const char*>(g.W), g.bias, g.residual, g.out, g.m_dev, g.M, g.N, g.K};
  P0.colscale = g.colscale; P0.alpha = g.alpha; P0.oscale = g.oscale;
  const LcProblem P1{};
  const int total = (g.N / lcBN) * ((g.M + lcBM - 1) / lcBM);
  int cus = 256;
  { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) cus = prop.multiProcessorCount & ~7; }
  const int grid = total < cus ? ((total + 7) & ~7) : cus;
  if (ev0) hipExtLaunchKernelGGL((gemm_lc2q_kernel<false>), dim3(grid), dim3(768), 0, st, ev0, ev1, 0, P0, P1, epi);
  else hipLaunchKernelGGL((gemm_lc2q_kernel<false>), dim3(grid), dim3(768), 0, st, P0, P1, epi);
  CMH_CHECK_LAUNCH("gemm (lc2q)");
  return 0;
}
// first form: bias (+ the implied scale), bf16 output, no residual, no activation
bool gemm_lc2q_takes(int N, int K, int epi) {
  return gemm_lc_mode() == 7 && N % lcBN == 0 && K % 128 == 0 && K >= 512 &&
         (epi & ~(EPI_BIAS | EPI_SCALE | 256)) == EPI_OUT_BF16;
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int lc_env_mode() { static const int m = []() { const char* e = getenv("CMH_GEMM_LC"); return e ? atoi(e) : 0; }(); return m; }
static int g_lc_mode = -1;         // cmh_set_gemm_lc: -1 = environment (CMH_GEMM_LC, default 0 = off)
int gemm_lc_mode() { return g_lc_mode < 0 ? lc_env_mode() : g_lc_mode; }
void gemm_lc_set_mode(int m) { g_lc_mode = m; }

// bf16 operands, 16-bit output, the forward epilogues of a transformer block (bias, + QuickGELU, + fp16 residual), N % 256 == 0
bool gemm_lc_takes(int dt, int N, int K, int epi) {
  if (dt != CMH_BF16 || N % lcBN != 0 || K % 64 != 0 || K < 256) return false;      // >= 4 K-steps per tile: the bias slots' reuse distance
  if (gemm_lc_mode() >= 4) {      // the 12-wave form (lc2): its parked stores need 8 K-steps per tile
    if (K < 512) return false;
    if (gemm_lc_mode() >= 5 && (epi & EPI_QUICKGELU)) return false;
    if (gemm_lc_mode() == 6 && (epi & EPI_RESIDUAL) && K / 64 > 16) return false;      // 6: QKV and out_proj only (where it measured level or better)
  }
  if (!(epi & (EPI_OUT_BF16 | EPI_OUT_F16)) || ((epi & EPI_OUT_BF16) && (epi & EPI_OUT_F16))) return false;
  if (epi & ~(EPI_BIAS | EPI_QUICKGELU | EPI_RESIDUAL | EPI_RES_F16 | EPI_OUT_BF16 | EPI_OUT_F16 | 256)) return false;
  if ((epi & EPI_RESIDUAL) && !((epi & EPI_RES_F16) && (epi & EPI_OUT_F16))) return false;      // a residual = the fp16 stream, in and out
  if ((epi & EPI_RESIDUAL) && (epi & EPI_QUICKGELU)) return false;
  return true;
}

// the wide kernel's residual-first rule (gemm_wide.hip, res_first): per GEMM
bool gemm_lc_res_first(int epi, int K) {
  return (epi & EPI_RESIDUAL) && !(epi & (EPI_QUICKGELU | EPI_GELU | EPI_RELU)) && K / 64 <= 16;
}

static int lc_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus < 8) cus = 256;
    cus &= ~7;
  }
  return cus;
}

// b == nullptr: one problem.  The problem with the longer K goes first (its tiles are the long jobs of the static schedule).
int launch_gemm_lc(const GemmProblem& a, const GemmProblem* b, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  for (const GemmProblem* g : {&a, b}) {
    if (!g) continue;
    if (static_cast<size_t>(g->M) * g->K * 2 >= (1ull << 32) || static_cast<size_t>(lcBN) * g->K * 2 >= (1ull << 32))
      return fail(CMH_ERR_INVALID, "gemm (lc): operand of %zu bytes exceeds the 32-bit offset range", static_cast<size_t>(g->M) * g->K * 2);
  }
  auto prob = [](const GemmProblem& g) {
    return LcProblem{static_cast<const char*>(g.A), static_cast<const char*>(g.W), g.bias, g.residual, g.out, g.m_dev, g.M, g.N, g.K};
  };
  auto tiles_of = [](const GemmProblem& g) { return (g.N / lcBN) * ((g.M + lcBM - 1) / lcBM); };
  const int cus = lc_cus();
  const int total = tiles_of(a) + (b ? tiles_of(*b) : 0);
  const int grid = total < cus ? ((total + 7) & ~7) : cus;     // sized for the upper bounds: workgroups without a tile exit at once
  const LcProblem P0 = prob(a), P1 = b ? prob(*b) : LcProblem{};
  if (gemm_lc_mode() >= 4) {
    const int res2 = !(epi & EPI_RESIDUAL) ? 0 : (gemm_lc_res_first(epi, a.K) ? 1 : 2);
    const bool f16 = (epi & EPI_OUT_F16) != 0;
#define LC2_GO(G, R, F)                                                                                                       \
  do {                                                                                                                       \
    if (ev0) hipExtLaunchKernelGGL((gemm_lc2_kernel<G, R, F>), dim3(grid), dim3(768), 0, st, ev0, ev1, 0, P0, P1, epi);      \
    else hipLaunchKernelGGL((gemm_lc2_kernel<G, R, F>), dim3(grid), dim3(768), 0, st, P0, P1, epi);                          \
  } while (0)
#define LC2_GO_G(G)                                                                                                           \
  do {                                                                                                                       \
    if (res2 == 1) LC2_GO(G, 1, true);                                                                                       \
    else if (res2 == 2) LC2_GO(G, 2, true);                                                                                  \
    else if (f16) LC2_GO(G, 0, true);                                                                                        \
    else LC2_GO(G, 0, false);                                                                                                \
  } while (0)
    if (b) LC2_GO_G(true); else LC2_GO_G(false);
#undef LC2_GO_G
#undef LC2_GO
    CMH_CHECK_LAUNCH("gemm (lc2)");
    return 0;
  }
  const bool rf = gemm_lc_res_first(epi, a.K);      // (the caller has checked that both problems agree)
  const bool f16o = (epi & EPI_OUT_F16) != 0;
#define LC_GO2(G, R, F)                                                                                                      \
  do {                                                                                                                       \
    if (ev0) hipExtLaunchKernelGGL((gemm_lc_kernel<G, R, F>), dim3(grid), dim3(512), 0, st, ev0, ev1, 0, P0, P1, epi);       \
    else hipLaunchKernelGGL((gemm_lc_kernel<G, R, F>), dim3(grid), dim3(512), 0, st, P0, P1, epi);                           \
  } while (0)
  // five forms: no residual -> bf16 or fp16 output; a residual (the fp16 stream, fp16 output) first or behind the bias
  const int res = !(epi & EPI_RESIDUAL) ? 0 : (rf ? 1 : 2);
#define LC_GO(G, R_unused)                                                                                                   \
  do {                                                                                                                       \
    if (res == 1) LC_GO2(G, 1, true);                                                                                        \
    else if (res == 2) LC_GO2(G, 2, true);                                                                                   \
    else if (f16o) LC_GO2(G, 0, true);                                                                                       \
    else LC_GO2(G, 0, false);                                                                                                \
  } while (0)
  static const int abl = []() { const char* e = getenv("CMH_LC_ABL"); return e ? atoi(e) : 0; }();
  if (abl && !b && !rf && !(epi & EPI_OUT_F16)) {      // diagnostic builds of the plain, residual-free, bf16-output form only
#define LC_GO_A(A)                                                                                                            \
  do {                                                                                                                       \
    if (ev0) hipExtLaunchKernelGGL((gemm_lc_kernel<false, 0, false, A>), dim3(grid), dim3(512), 0, st, ev0, ev1, 0, P0, P1, epi); \
    else hipLaunchKernelGGL((gemm_lc_kernel<false, 0, false, A>), dim3(grid), dim3(512), 0, st, P0, P1, epi);                    \
  } while (0)
    if (abl == 1) LC_GO_A(1); else if (abl == 2) LC_GO_A(2); else if (abl == 3) LC_GO_A(3); else LC_GO_A(4);
#undef LC_GO_A
    CMH_CHECK_LAUNCH("gemm (lc, diagnostic)");
    return 0;
  }
  if (b) { if (rf) LC_GO(true, true); else LC_GO(true, false); }
  else { if (rf) LC_GO(false, true); else LC_GO(false, false); }
#undef LC_GO
#undef LC_GO2
  CMH_CHECK_LAUNCH("gemm (lc)");
  return 0;
}

}  // namespace cmh

#ifdef LC_STAMPS
extern "C" int cmh_debug_lc_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cmh::g_lc_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int cmh_set_gemm_lc(int32_t mode) {
  CMH_CHECK_ARG(mode >= -1 && mode <= 7, "set_gemm_lc: mode %d (-1 environment, 0 off, 1 every eligible launch, 2 all but QuickGELU launches, 3 by cost model, 4 / 5 / 6 the 12-wave form for every block launch / without the QuickGELU ones / for QKV and out_proj only)", mode);
  cmh::gemm_lc_set_mode(mode);
  return CMH_OK;
}
