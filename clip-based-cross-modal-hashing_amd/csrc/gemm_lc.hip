// GEMM "lc" (loader / consumer wave groups that swap roles every tile, round 5): out[M,N] = epi(X[M,K].W[N,K]^T), bf16 operands, 16-bit
// output, 128(m) x 256(n) tile per 512-thread workgroup, persistent like the wide kernel (gemm_wide.hip) - but with the jobs of its
// K loop and of its tile switch on DIFFERENT waves.
//
// Why.  (1) The wide kernel's eight waves each issue their share of a stage's LDS-DMA pieces between their own MFMAs.  A wave issues
// in order: while it sits in the ~60-180 cycle issue of a `global_load_lds_dwordx4` it issues no MFMA (1750-2000 cycles per
// K-step against 1280 of MFMA work).  (2) At a tile switch every wave runs the epilogue - bias, activation, packing, stores - while
// the matrix pipes idle: 2.2 us of a 13.7 us QKV tile, 5.0 us with QuickGELU (tools/gemm_tile_cost.py).
// Here the workgroup is two groups of four waves, one wave of each group per SIMD.  For tile t group t % 2 MULTIPLIES - 64 x 128 of
// the tile per wave, 64 MFMAs per K-step issued back to back from one wave (one wave keeps a SIMD's matrix pipe full: 16 cycles per
// v_mfma_f32_16x16x32_bf16), fragment reads in the MFMAs' issue gaps, nothing else - and the other group STAGES: 12 LDS-DMA pieces of
// 1 KiB per wave and K-step, a counted `s_waitcnt vmcnt`, the barrier.  At the end of the tile the groups swap: the group that holds
// tile t's accumulators becomes the staging group of tile t + 1 and works its epilogue off in eight slices, one per K-step, between
// its DMA issues - under the OTHER group's MFMAs, which started tile t + 1 without waiting for anything.  Measured with stamps
// (tools/lc_stamps.py, profiles/r05_e_lc_stamps.txt, before the swap existed): 1240 cycles per K-step (the matrix pipe 83 % busy) -
// and 5050 cycles of epilogue per tile on the critical path, a quarter of a QKV tile; the swap takes them off it.
//
// Ring and protocol (the wide kernel's): 3 stages x 48 KB (W 256 rows x 128 B | X 128 rows x 128 B, lane-linear image, XOR swizzle on
// the DMA source chunk and on the ds_read_b128); K-step s lives in buffer s % 3; ONE s_barrier per K-step, placed after the MFMA
// waves hold every fragment of stage s in registers: behind it the staging group overwrites buffer s % 3 with stage s + 3 (two
// K-steps of flight) and the MFMA waves start reading stage s + 1.  Whoever ISSUED a stage waits for it (its own counted vmcnt) in
// front of the barrier its readers pass: for the first two K-steps of a tile that is the group which now multiplies.
// Fragments: the W operand's eight 16-row fragments are refilled IN PLACE (fragment a of the next half-step is read as soon as the
// four MFMAs that use fragment a have issued), the X operand's four are double-buffered: 64 fragment registers + 128 accumulators.
// Same k order per output element as the wide kernel (k = 64 kt + 32 ks + 8 fq + j inside v_mfma_f32_16x16x32_bf16, K-steps in
// sequence), same epilogue operation order, same residual-first rule: the SAME BITS (tests/test_gpu_lc.py compares with torch.equal).
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
constexpr int lcSlices = 8;                        // a tile's deferred epilogue: this many slices, one per K-step of the next tile
constexpr int lcMinNk = lcSlices;                  // ... so a tile has at least that many K-steps (K >= 512)

struct LcProblem {
  const char* X; const char* W; const float* bias; const void* residual; void* out;
  const int* m_dev;      // optional: the real row count on the device (Mub is then an upper bound)
  int Mub, N, K;
};

__device__ __forceinline__ int lc_swz(int row, int chunk) { return row * lcRowBytes + ((chunk ^ (row & 7)) << 4); }

#define LC_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

#ifdef LC_STAMPS   // diagnostic build only (tools/lc_stamps.py): per workgroup, wave 0: shader-clock ticks (s_memtime) of the whole kernel,
                   // until the first stage, inside its K loops, inside its direct epilogues; and the 100 MHz wall clock over the same span
__device__ unsigned long long g_lc_stamps[256 * 8];
__device__ unsigned long long g_lc_stamps2[256 * 8];      // the staging side (wave 4): ticks in the counted wait + barrier / DMA issue / slices, and the steps of each kind
#define LC_T() __builtin_amdgcn_s_memtime()
#endif

// the lane id, computed where it is needed: `volatile`, so that neither it nor what is derived from it is hoisted out of the tile loop as
// a loop invariant (such values live - spilled - through both roles, and a scratch reload in the staging role comes with an
// `s_waitcnt vmcnt(0)` that drains the LDS-DMA queue)
__device__ __forceinline__ int lc_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// counted wait on the vector-memory queue (LDS-DMA pieces, loads and stores retire in issue order): a smaller count is always safe
__device__ __forceinline__ void lc_wait_vm(int n) {
  if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// RF: the residual-first rule applies to this launch (short K with a residual: the accumulators start as the residual tile); a template
// parameter so that a tile's first K-step is one straight path - C = 0 inside the first MFMAs, or the loaded residual
template <bool GRP, bool RF>
__global__ __launch_bounds__(512) void gemm_lc_kernel(LcProblem p0, LcProblem p1, int epi) {
  __shared__ __attribute__((aligned(1024))) char lds[3 * lcSTG + 3 * 1024];      // the ring + three bias rows (tile index mod 3)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;        // role group: multiplies the tiles of its parity, stages the others
  const int wv = wid & 3;          // wave inside the group

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
  auto nk_of = [&](int ti) __attribute__((always_inline)) { return (GRP && ti >= n_first) ? nk1 : nk0; };
  // tile ti -> (second problem?, m0, n0)
  auto tile_of = [&](int ti, bool& second, int& m0, int& n0) __attribute__((always_inline)) {
    const int j = slot + ti * per;
    second = GRP && ti >= n_first;
    const int logical = second ? lo1 + (j - len0) : lo0 + j;
    const int tn_cnt = second ? tiles_n1 : tiles_n0;
    const int tm = logical / tn_cnt;
    m0 = tm * lcBM;
    n0 = (logical - tm * tn_cnt) * lcBN;
  };
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lc_lptr_t)lds));

  // ================================================= the staging side ==================================================================
  // Piece = 1 KiB = 8 rows x 128 B: lane i fills (row 8 piece + i / 8, physical chunk i % 8) and fetches logical chunk (i % 8) ^ (row & 7).
  // Wave v of the staging group moves W pieces 8 v .. 8 v + 7 and X pieces 4 v .. 4 v + 3 of every stage.
  // The issue STATE (which stage comes next: tile, K-step, ring buffer) is plain scalar bookkeeping that BOTH groups advance once per
  // K-step, so that whichever group stages next finds it current; the per-lane offsets are recomputed when a group takes the role up
  // (they are dead while it multiplies) and when the tile being staged changes.
  uint32_t offW[8], offX[4];
  const char* Wt = nullptr;
  const char* Xt = nullptr;
  const char* Bt = nullptr;        // &bias[n0] of the tile being staged
  int i_tile = 0, i_kt = 0, ibuf = 0, i_nk = nk_of(0);
  int issued = 0;                  // stages issued so far by the workgroup = index of the stage to issue next
  int off_tile = -1;               // the tile offX / Wt / Xt / Bt are valid for (this group's view)
  auto set_offsets = [&](int ti) __attribute__((always_inline)) {
    // (the lane's row / chunk are derived HERE from an opaque copy of the lane id: hoisted out of the tile loop as loop invariants,
    // the twelve row constants would live - spilled - through the multiplying role, and their reloads, `s_waitcnt vmcnt(0)` in front,
    // would drain the LDS-DMA queue)
    const int l_ = lc_lane_now();
    const int sub = l_ >> 3, ch = l_ & 7;
    bool second; int m0, n0;
    tile_of(ti, second, m0, n0);
    const uint32_t rs = static_cast<uint32_t>(second ? K1 : K0) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wv * 8 + i) * 8 + sub;
      offW[i] = static_cast<uint32_t>(row) * rs + ((ch ^ (row & 7)) << 4);
    }
    const int Mp = second ? M1 : M0;
    Wt = (second ? W1 : W0) + static_cast<size_t>(n0) * rs;
    Xt = second ? X1 : X0;
    Bt = reinterpret_cast<const char*>((second ? B1 : B0) + n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wv * 4 + i) * 8 + sub;
      int xr = m0 + row;
      xr = xr < Mp ? xr : Mp - 1;               // rows past M are computed on duplicated data and never stored
      offX[i] = static_cast<uint32_t>(xr) * rs + ((ch ^ (row & 7)) << 4);      // < 4 GiB: checked on the host
    }
    off_tile = ti;
  };
  const bool bias_wave = wv == 0 && (epi & EPI_BIAS);
  // the pieces of stage `issued` (the staging group only).  The tile's 256 bias values (1 KiB) go to LDS with its first stage, OLDER
  // than that stage's pieces in this wave's queue: every counted wait that retires the stage retires them too.  Slot = tile mod 3:
  // tile t's epilogue is read during tile t + 1 at the latest, tile t + 3's row arrives during tile t + 2.
  auto issue_pieces = [&]() __attribute__((always_inline)) {
    if (off_tile != i_tile) set_offsets(i_tile);
    char* base = lds + ibuf * lcSTG;
    const uint32_t koff = static_cast<uint32_t>(i_kt) * lcRowBytes;
    if (i_kt == 0 && bias_wave) {
      const uint32_t l16 = static_cast<uint32_t>(lc_lane_now()) * 16u;
      __builtin_amdgcn_global_load_lds((lc_gptr_t)(Bt + l16), (lc_lptr_t)(lds + 3 * lcSTG + (i_tile % 3) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((lc_gptr_t)(Wt + koff + offW[i]), (lc_lptr_t)(base + (wv * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((lc_gptr_t)(Xt + koff + offX[i]), (lc_lptr_t)(base + lcWBytes + (wv * 4 + i) * 1024), 16, 0, 0);
  };
  auto advance_issue = [&]() __attribute__((always_inline)) {     // scalars only: both groups, once per issued stage
    ++issued;
    ibuf = ibuf == 2 ? 0 : ibuf + 1;
    if (++i_kt == i_nk) { i_kt = 0; ++i_tile; i_nk = nk_of(i_tile); }
  };

  // ================================================ the multiplying side ==============================================================
  const int wm = wv >> 1, wn = wv & 1;            // 2 (m) x 2 (n) waves of 64 x 128
  const int frow = lane & 15, fq = lane >> 4;
  const uint32_t aW = lds_base + lc_swz(wn * 128 + frow, fq);
  const uint32_t aX = lds_base + lcWBytes + lc_swz(wm * 64 + frow, fq);

  lc_f32x4_t acc[8][4];                            // [n-fragment][m-fragment]
  lc_u32x4_t fw[8], fxa[4], fxb[4];

  // fp16 residual rows in the 16-byte layout of the packed output (lane = one row x 8 consecutive n), brought to the accumulator layout
  // by v_permlane16_swap; two 16-row fragments (b) at a time: 8 loads of 16 bytes in flight per lane
  auto add_residual = [&](const void* residual, int M, int N, int m0, int n0) __attribute__((always_inline)) {
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
  // the twelve fragments of (ring buffer `buf`, k 0..31): X then W, the order every half-step issues them in
  auto read_first_frags = [&](int buf) __attribute__((always_inline)) {
    const uint32_t bo = static_cast<uint32_t>(buf) * lcSTG;
    const uint32_t w0 = aW + bo, x0 = aX + bo;
    LC_READ_X(fxa, 0, x0); LC_READ_X(fxa, 1, x0); LC_READ_X(fxa, 2, x0); LC_READ_X(fxa, 3, x0);
    LC_READ_W(0, w0); LC_READ_W(1, w0); LC_READ_W(2, w0); LC_READ_W(3, w0);
    LC_READ_W(4, w0); LC_READ_W(5, w0); LC_READ_W(6, w0); LC_READ_W(7, w0);
  };

  int cur = 0;            // ring buffer of the current K-step (both groups keep it)
  int ns1 = 0;            // stores this wave issued in the previous K-step (younger entries of its vmcnt queue)

  // One K-step of the multiplying group.  Half 0 multiplies (fxa, fw) = k 0..31 and reads k 32..63 of the same stage (fxb; fw in place);
  // half 1 multiplies those, passes the barrier after its second fragment group - every read of this stage was issued at least 8 MFMAs
  // earlier - and (unless LAST: the next tile belongs to the other group) reads k 0..31 of the next stage (fxa; fw in place).  LDS returns
  // a wave's reads in order, so "fragment a has landed" is a count of the reads issued after it: 11 in the steady state of half 0 (7 - a
  // older W fragments still to come, the 4 X reads and the a W reads of the next half-step issued since), 7 - a at the top of half 1.
  // vm_wait >= 0: this group issued the stage that must have landed at this barrier (the first two K-steps after it stopped staging).
  // ONE instantiation, one loop: a tile's first K-step finds its accumulators zeroed or holding the residual tile; its last K-step reads the next stage's fragments like any other (they belong to the
  // other group's tile: twelve wasted LDS reads per tile instead of a second copy of the loop body for the allocator to reconcile).
  auto kstep = [&](int vm_wait) __attribute__((always_inline)) {
    constexpr bool ZC = false, LAST = false;
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
    if (vm_wait >= 0) lc_wait_vm(vm_wait);         // ... and the stage this group staged for the next K-step has landed
    __builtin_amdgcn_s_barrier();                  // every wave; the next stage has landed
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) {
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
    } else {
      LC_MFMA1(2, 0); LC_MFMA1(2, 1); LC_MFMA1(2, 2); LC_MFMA1(2, 3);
      __builtin_amdgcn_sched_barrier(0);
    }
#define LC_GROUP1(a)                                                                                   \
  do {                                                                                                 \
    LC_MFMA1(a, 0); LC_MFMA1(a, 1); LC_MFMA1(a, 2); LC_MFMA1(a, 3);                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if constexpr (!LAST) { LC_READ_W(a, w0); __builtin_amdgcn_sched_barrier(0); }                      \
  } while (0)
    LC_GROUP1(3); LC_GROUP1(4); LC_GROUP1(5); LC_GROUP1(6); LC_GROUP1(7);
#undef LC_GROUP1
#undef LC_MFMA1
    cur = nxt;
    if (issued < S) advance_issue();               // (the staging group issued a stage behind this K-step's barrier)
  };

  // ---- a tile's epilogue: bias -> [QuickGELU] -> [residual, long K] -> 16-bit pack -> 16-byte stores -------------------------------
  // the tile whose accumulators this wave holds
  // (all wave-uniform: what a lane needs on top - its row, its column, its LDS bias address - it derives from its lane id in every slice;
  // four per-lane values kept across the staging role's K-steps were four values the allocator spilled there)
  char* ep_obase = nullptr;        // &out[m0 * N + n0] of that tile
  uint32_t ep_ldn = 0;             // bytes per output row
  int ep_m0 = 0, ep_M = 0;         // the tile's first row / the problem's row count (rows >= M are never stored)
  uint32_t ep_bslot = 0;           // LDS address of the tile's bias row
  bool ep_full = false;            // every row of the tile exists (and the store-skipping ablation is off)
  bool pending = false;            // the epilogue waits to be worked off in slices while this group stages the next tile
  // slice J (compile-time): n-fragment pair J / 2, m-fragments 2 (J % 2) and 2 (J % 2) + 1: two 16-byte stores
  lc_f32x4_t ep_bv0 = {0.f, 0.f, 0.f, 0.f}, ep_bv1 = {0.f, 0.f, 0.f, 0.f};      // deferred slices: the pair's bias values, read one K-step ahead
  // BIAS: 0 none (already added), 1 read here, 2 the values are in ep_bv0 / ep_bv1 (on their way since the previous slice)
  auto ep_pair = [&](auto prc, auto b0c, auto biasc) __attribute__((always_inline)) {
    constexpr int PR = decltype(prc)::value, B0 = decltype(b0c)::value;
    constexpr int BIAS = decltype(biasc)::value;
    lc_f32x4_t bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
    const int l_ = lc_lane_now();
    const int frow_ = l_ & 15, fq_ = l_ >> 4;
    const uint32_t ep_bias = ep_bslot + (wn * 128 + fq_ * 4) * 4;
    const int ep_mrow = ep_m0 + wm * 64 + frow_;
    // the lane's first output byte inside the tile: row (wm*64 + frow), column wn*128 + (fq&1)*16 + (fq&2)*4  (+ 32 per pair)
    const uint32_t ep_off = static_cast<uint32_t>(wm * 64 + frow_) * ep_ldn + static_cast<uint32_t>(wn * 128 + (fq_ & 1) * 16 + (fq_ & 2) * 4) * 2;
    if (BIAS == 1 && (epi & EPI_BIAS)) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv0) : "v"(ep_bias), "n"(2 * PR * 64));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv1) : "v"(ep_bias), "n"((2 * PR + 1) * 64));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv0), "+v"(bv1)::"memory");
    }
    if (BIAS == 2 && (epi & EPI_BIAS)) {
      if constexpr (B0 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ep_bv0), "+v"(ep_bv1)::"memory");      // issued a K-step ago
      bv0 = ep_bv0; bv1 = ep_bv1;
    }
#pragma unroll
    for (int b = B0; b < B0 + 2; ++b) {
      lc_f32x4_t v0 = acc[2 * PR][b], v1 = acc[2 * PR + 1][b];
      if (BIAS != 0 && (epi & EPI_BIAS)) { v0 += bv0; v1 += bv1; }
      if (epi & EPI_QUICKGELU) {
        // two values at a time, the wide kernel's instruction sequence (packed-f32 multiply / add, v_exp_f32 / v_rcp_f32 per value)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          lc_f32x4_t& vv = h ? v1 : v0;
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const lc_f32x2_t v = {vv[j], vv[j + 1]};
            const lc_f32x2_t t = v * lc_f32x2_t{-2.4554669595930157f, -2.4554669595930157f};
            const lc_f32x2_t d = lc_f32x2_t{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + lc_f32x2_t{1.0f, 1.0f};
            const lc_f32x2_t o = v * lc_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            vv[j] = o[0];
            vv[j + 1] = o[1];
          }
        }
      }
      uint32_t lo[2], hi[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        if (epi & EPI_OUT_F16) {
          lo[w] = pack_f16x2(v0[2 * w], v0[2 * w + 1]);
          hi[w] = pack_f16x2(v1[2 * w], v1[2 * w + 1]);
        } else {
          lo[w] = pack_bf16x2(v0[2 * w], v0[2 * w + 1]);
          hi[w] = pack_bf16x2(v1[2 * w], v1[2 * w + 1]);
        }
      }
      // v_permlane16_swap exchanges, between the lane pairs (l, l + 16), the packed words of two neighbouring n-fragments: an even
      // lane-row then owns 8 consecutive n of fragment 2 pr and an odd lane-row 8 consecutive n of fragment 2 pr + 1 -> 16-byte stores
      const lc_u2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
      const lc_u2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
      if (ep_full) {                                          // (uniform: a full tile's stores need no lane mask)
        *reinterpret_cast<lc_u32x4_t*>(ep_obase + (ep_off + static_cast<uint32_t>(b * 16) * ep_ldn + PR * 64)) = lc_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
      } else if (ep_mrow + b * 16 < ep_M && !(epi & 256)) {   // 256 = timing-only ablation: skip stores (with M made ragged)
        *reinterpret_cast<lc_u32x4_t*>(ep_obase + (ep_off + static_cast<uint32_t>(b * 16) * ep_ldn + PR * 64)) = lc_u32x4_t{s0[0], s1[0], s0[1], s1[1]};
      }
    }
    if constexpr (BIAS == 2 && B0 == 2 && PR < 3) {           // the next pair's bias values: on their way while the next K-step's DMA is issued
      if (epi & EPI_BIAS) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ep_bv0) : "v"(ep_bias), "n"(2 * (PR + 1) * 64));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ep_bv1) : "v"(ep_bias), "n"((2 * (PR + 1) + 1) * 64));
      }
    }
  };
  // the first pair's bias values, before the first deferred slice
  auto ep_bias_first = [&]() __attribute__((always_inline)) {
    if (epi & EPI_BIAS) {
      const int fq_ = lc_lane_now() >> 4;
      const uint32_t ep_bias = ep_bslot + (wn * 128 + fq_ * 4) * 4;
      asm volatile("ds_read_b128 %0, %1" : "=v"(ep_bv0) : "v"(ep_bias));
      asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(ep_bv1) : "v"(ep_bias));
    }
  };
  auto ep_slice = [&](auto jc, auto biasc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    ep_pair(std::integral_constant<int, J / 2>{}, std::integral_constant<int, 2 * (J % 2)>{}, biasc);
  };

#ifdef LC_STAMPS
  unsigned long long sl_wait = 0, sl_dma = 0, sl_slice = 0, sl_steps = 0, sl_slices = 0;
#endif
  int sg = 0;             // the workgroup's K-step counter (both groups keep it)
  // One K-step of the staging group: [its counted wait][barrier][first fragments of its next tile][the next stage's pieces][slice J of
  // its pending epilogue, J >= 0].  must_wait: this group issued the stage that must have landed now - always, but in the first two
  // K-steps after it stopped multiplying.  Its queue, oldest first: ... [stage s+1][stores of step s-2][stage s+2][stores of step s-1].
  auto lstep = [&](auto jc, bool must_wait, bool prefetch) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
#ifdef LC_STAMPS
    const unsigned long long l0_ = LC_T();
#endif
    // The slice FIRST: its two stores meet an idle address path here; behind this step's DMA burst they would wait until all 48 pieces
    // of the stage have been accepted (measured: 1 200 ticks per slice that way).  The queue of this wave, oldest first:
    // ... [stage s+1][stores of step s-1][stage s+2][stores of this step].
    int ns0 = 0;
    if constexpr (J >= 0) { ep_slice(jc, std::integral_constant<int, 2>{}); ns0 = 2; }
#ifdef LC_STAMPS
    const unsigned long long l1_ = LC_T();
#endif
    if (must_wait && sg + 1 < S) lc_wait_vm((sg + 2 < S ? 12 : 0) + ns0 + ns1);
    __builtin_amdgcn_s_barrier();
#ifdef LC_STAMPS
    const unsigned long long l2_ = LC_T();
#endif
    if (prefetch) read_first_frags(cur == 2 ? 0 : cur + 1);      // this group multiplies the next tile
    if (issued < S) { issue_pieces(); advance_issue(); }
    ns1 = ns0;
#ifdef LC_STAMPS
    { const unsigned long long l3_ = LC_T(); sl_slice += l1_ - l0_; sl_wait += l2_ - l1_; sl_dma += l3_ - l2_; sl_steps += 1; if (J >= 0) sl_slices += 1; }
#endif
    cur = cur == 2 ? 0 : cur + 1;
    ++sg;
  };

#ifdef LC_STAMPS
  const unsigned long long st_t0 = LC_T(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_first = 0, st_k = 0, st_e = 0, st_mark = 0, st_ks = 0, st_nt = 0;
#endif
  // ---- prologue: group 1 stages (it is tile 0's staging group), group 0 gets tile 0's residual rows on their way (if they come first) ----
  {
    const int n_pro = S < 3 ? S : 3;
    if (grp == 1) {
      set_offsets(0);
      for (int j = 0; j < n_pro; ++j) { issue_pieces(); advance_issue(); }
      if (n_pro > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");        // stage 0 landed; two younger stages may fly
      else if (n_pro > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int j = 0; j < n_pro; ++j) advance_issue();
      if constexpr (RF) {
        bool second; int m0, n0;
        tile_of(0, second, m0, n0);
        zero_acc();
        add_residual(second ? R1 : R0, second ? M1 : M0, second ? N1 : N0, m0, n0);
      }
    }
  }
  __builtin_amdgcn_s_barrier();                    // stage 0 has landed
#ifdef LC_STAMPS
  st_first = LC_T() - st_t0;
#endif
  if (grp == 0) read_first_frags(0);

  for (int ti = 0; ti < my_tiles; ++ti) {
    const int nk = nk_of(ti);
    if ((ti & 1) == grp) {
      // ------------------------------------------------ this group multiplies tile ti ----------------------------------------------
      bool second; int m0, n0;
      tile_of(ti, second, m0, n0);
      const int M = second ? M1 : M0, N = second ? N1 : N0;
      const void* residual = second ? R1 : R0;
      // (zeroed HERE: in the idle time of the staging role it would keep all 128 registers live across that role's loop, and the
      // allocator then spills its way through the staging steps - whose reloads drain the LDS-DMA queue)
      if constexpr (RF) { if (ti > 0) { zero_acc(); add_residual(residual, M, N, m0, n0); } }
      else zero_acc();
#ifdef LC_STAMPS
      st_mark = LC_T();
#endif
      // the two stages in flight when this group stopped staging are its own: it waits for them (K-steps 0 and 1)
      __builtin_amdgcn_s_setprio(3);      // the multiplying wave of a SIMD goes first; its partner's slices and DMA issues take the slots it leaves
      {
        // its queue when it stopped staging: ... [stage s0+1][stores of the last staging step][stage s0+2]
        const int vm0 = ti > 0 ? 12 + ns1 : -1, vm1 = ti > 0 ? 0 : -1;
        for (int kt = 0; kt < nk; ++kt) kstep(kt == 0 ? vm0 : (kt == 1 ? vm1 : -1));
        ns1 = 0;
      }
      sg += nk;
      __builtin_amdgcn_s_setprio(0);
#ifdef LC_STAMPS
      { const unsigned long long t_ = LC_T(); st_k += t_ - st_mark; st_mark = t_; st_ks += nk; ++st_nt; }
#endif
      // ---- the tile's epilogue: deferred (worked off in slices while this group stages the next tile) or here ----------------------
      ep_obase = static_cast<char*>(second ? O1 : O0) + (static_cast<size_t>(m0) * N + n0) * 2;
      ep_ldn = static_cast<uint32_t>(N) * 2;
      ep_m0 = m0;
      ep_M = M;
      ep_bslot = lds_base + 3 * lcSTG + (ti % 3) * 1024;
      ep_full = m0 + lcBM <= M && !(epi & 256);
      const bool late_residual = (epi & EPI_RESIDUAL) && !RF;      // long K: the residual goes behind the bias
      if (ti + 1 < my_tiles && !late_residual) {
        pending = true;
      } else {
        if (late_residual) {
          if (epi & EPI_BIAS) {
            const uint32_t ep_bias = ep_bslot + (wn * 128 + fq * 4) * 4;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
              lc_f32x4_t bv0, bv1;
              asm volatile("ds_read_b128 %0, %1" : "=v"(bv0) : "v"(ep_bias + 2 * pr * 64));
              asm volatile("ds_read_b128 %0, %1" : "=v"(bv1) : "v"(ep_bias + (2 * pr + 1) * 64));
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv0), "+v"(bv1)::"memory");
#pragma unroll
              for (int b = 0; b < 4; ++b) { acc[2 * pr][b] += bv0; acc[2 * pr + 1][b] += bv1; }
            }
          }
          add_residual(residual, M, N, m0, n0);
#define LC_EP_ALL(BT) \
  ep_slice(std::integral_constant<int, 0>{}, BT); ep_slice(std::integral_constant<int, 1>{}, BT); ep_slice(std::integral_constant<int, 2>{}, BT); \
  ep_slice(std::integral_constant<int, 3>{}, BT); ep_slice(std::integral_constant<int, 4>{}, BT); ep_slice(std::integral_constant<int, 5>{}, BT); \
  ep_slice(std::integral_constant<int, 6>{}, BT); ep_slice(std::integral_constant<int, 7>{}, BT)
          LC_EP_ALL((std::integral_constant<int, 0>{}));
        } else {
          LC_EP_ALL((std::integral_constant<int, 1>{}));
        }
#undef LC_EP_ALL
      }
#ifdef LC_STAMPS
      st_e += LC_T() - st_mark;
#endif
    } else {
      // ------------------------------------------- this group stages tile ti (and finishes tile ti - 1's epilogue) -----------------
      set_offsets(i_tile);                            // the lane offsets were dead while this group multiplied
      const bool own = ti == 0;                       // tile 0: the prologue's stages are this group's own
      const bool nxt = ti + 1 < my_tiles;             // this group multiplies the next tile: it fetches its first fragments itself
      int kt = 0;
      if (pending) {                                  // straight-line: the slice is a compile-time choice of registers
        ep_bias_first();
        lstep(std::integral_constant<int, 0>{}, own, false);
        lstep(std::integral_constant<int, 1>{}, own, false);
        lstep(std::integral_constant<int, 2>{}, true, false);
        lstep(std::integral_constant<int, 3>{}, true, false);
        lstep(std::integral_constant<int, 4>{}, true, false);
        lstep(std::integral_constant<int, 5>{}, true, false);
        lstep(std::integral_constant<int, 6>{}, true, false);
        lstep(std::integral_constant<int, 7>{}, true, nxt && nk == lcSlices);
        pending = false;
        kt = lcSlices;
      }
      for (; kt < nk; ++kt) lstep(std::integral_constant<int, -1>{}, own || kt >= 2, nxt && kt == nk - 1);
    }
  }
#ifdef LC_STAMPS
  if (wid == 0 && lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_lc_stamps + blockIdx.x * 8;
    o[0] = LC_T() - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0; o[2] = st_first; o[3] = st_k; o[4] = st_e;
    o[5] = st_ks; o[6] = static_cast<unsigned long long>(my_tiles); o[7] = st_nt;
  }
  if (wid == 4 && lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_lc_stamps2 + blockIdx.x * 8;
    o[0] = sl_wait; o[1] = sl_dma; o[2] = sl_slice; o[3] = sl_steps; o[4] = sl_slices; o[5] = static_cast<unsigned long long>(my_tiles);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (nothing of this wave's is in flight into LDS when it ends)
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static int lc_env_mode() { static const int m = []() { const char* e = getenv("CMH_GEMM_LC"); return e ? atoi(e) : 0; }(); return m; }
static int g_lc_mode = -1;         // cmh_set_gemm_lc: -1 = environment (CMH_GEMM_LC, default 0 = off)
int gemm_lc_mode() { return g_lc_mode < 0 ? lc_env_mode() : g_lc_mode; }
void gemm_lc_set_mode(int m) { g_lc_mode = m; }

// bf16 operands, 16-bit output, the forward epilogues of a transformer block (bias, + QuickGELU, + fp16 residual), N % 256 == 0, K >= 512
bool gemm_lc_takes(int dt, int N, int K, int epi) {
  if (dt != CMH_BF16 || N % lcBN != 0 || K % 64 != 0 || K < 64 * lcMinNk) return false;      // a tile's K-steps carry the previous tile's epilogue slices
  if (!(epi & (EPI_OUT_BF16 | EPI_OUT_F16)) || ((epi & EPI_OUT_BF16) && (epi & EPI_OUT_F16))) return false;
  if (epi & ~(EPI_BIAS | EPI_QUICKGELU | EPI_RESIDUAL | EPI_RES_F16 | EPI_OUT_BF16 | EPI_OUT_F16 | 256)) return false;
  if ((epi & EPI_RESIDUAL) && !(epi & EPI_RES_F16)) return false;      // the 16-bit-output launches carry the fp16 stream
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
  const bool rf = gemm_lc_res_first(epi, a.K);      // (the caller has checked that both problems agree)
#define LC_GO(G, R)                                                                                                          \
  do {                                                                                                                       \
    if (ev0) hipExtLaunchKernelGGL((gemm_lc_kernel<G, R>), dim3(grid), dim3(512), 0, st, ev0, ev1, 0, P0, P1, epi);          \
    else hipLaunchKernelGGL((gemm_lc_kernel<G, R>), dim3(grid), dim3(512), 0, st, P0, P1, epi);                              \
  } while (0)
  if (b) { if (rf) LC_GO(true, true); else LC_GO(true, false); }
  else { if (rf) LC_GO(false, true); else LC_GO(false, false); }
#undef LC_GO
  CMH_CHECK_LAUNCH("gemm (lc)");
  return 0;
}

}  // namespace cmh

#ifdef LC_STAMPS
extern "C" int cmh_debug_lc_stamps(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cmh::g_lc_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
extern "C" int cmh_debug_lc_stamps2(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cmh::g_lc_stamps2), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int cmh_set_gemm_lc(int32_t mode) {
  CMH_CHECK_ARG(mode >= -1 && mode <= 3, "set_gemm_lc: mode %d (-1 environment, 0 off, 1 every eligible launch, 2 all but QuickGELU launches, 3 by cost model)", mode);
  cmh::gemm_lc_set_mode(mode);
  return CMH_OK;
}
