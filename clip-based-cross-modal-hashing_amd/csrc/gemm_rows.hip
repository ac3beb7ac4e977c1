// GEMM for FEW ROWS (M <= 2048; round 3 started with 512): out[M,N] = epi(X[M,K].W[N,K]^T) on 64(m) x 64(n) tiles, one 256-thread workgroup per tile.
//
// Why it exists.  The pooled-row tail of the towers (encoders.hip: run_block_pooled) multiplies B = 256 rows by the block's
// weights: three GEMMs of M = 256 per tower and step.  On the wide kernel (gemm_wide.hip: 96..160 x 256 tiles, one persistent
// workgroup per CU) such a launch has 6..36 tiles for 256 CUs and walks the whole K range at ~1 us per K-step: 14..46 us per
// launch (profiles/r02_c_small_gemm_dispatches.txt), 4.6 % of a step's GEMM time for 0.2 % of its FLOPs.  Here the same product is
// cut into 64 x 64 tiles (M = 256, N = 768..3072: 48..192 workgroups, two per CU), each streaming its 64 W rows and 64 X rows through
// a 4-stage LDS-DMA ring of 16 KB stages.
//
// Same bits as the wide kernel.  Every output element is the same chain of v_mfma_f32_16x16x32_bf16 (v_mfma_f32_16x16x4_f32 in f32
// mode) accumulations - W rows as the A operand, k = 64 kt + 32 ks + 8 fq + j ascending in kt, ks - followed by the same epilogue
// operations in the same order (residual pre-loaded into the accumulator when the wide kernel would: no activation, <= 16 K-steps),
// so the pooled-row tail stays bit-identical to the full-size path (tests/test_gpu_clip.py::
// test_pooled_rows_through_the_last_block_are_bit_identical).  Epilogues: bias, QuickGELU / GELU / ReLU, f32 or fp16 residual,
// f32 / bf16 / fp16 output, and the fp8 mode's e4m3 operands with scale + bias, fp16 / bf16 / e4m3 output; EPI_MUL_DQGELU and
// EPI_SAVE_PRE stay on the wide kernel.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "cmh_common.h"

namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 r_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float r_f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t r_u32x4_t;

constexpr int rT = 64;                 // tile rows and columns
constexpr int rRowBytes = 128;         // one K-step of a row
constexpr int rHalf = rT * rRowBytes;  // 8 KB: one operand's part of a stage
constexpr int rStages = 4;

__device__ __forceinline__ int r_swz(int row, int chunk) { return row * rRowBytes + ((chunk ^ (row & 7)) << 4); }
// the wide kernel's QuickGELU, operation for operation (gemm_wide.hip: w_quick_gelu)
__device__ __forceinline__ float r_quick_gelu(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
}

typedef const __attribute__((address_space(1))) void* r_gptr_t;
typedef __attribute__((address_space(3))) void* r_lptr_t;

struct RowsScales { const float* colscale; float alpha, oscale; };   // fp8: acc * (alpha * colscale[n]) + bias; e4m3 output of v * oscale

template <int DT>   // operands: 0 f32, 1 bf16, 2 OCP e4m3 (the wide kernel's fp8 K-step: one scaled MFMA of 128 k per fragment pair)
__global__ __launch_bounds__(256) void gemm_rows_kernel(const char* __restrict__ X, const char* __restrict__ W,
                                                        const float* __restrict__ bias, const float* residual, void* out, int M, int N,
                                                        int K, int epi, RowsScales sc) {
  constexpr bool F32 = DT == 0, FP8 = DT == 2;
  __shared__ __attribute__((aligned(1024))) char lds[rStages * 2 * rHalf];   // [stage][0 = W, 1 = X][64 rows x 128 B]

  constexpr int ELT = F32 ? 4 : (FP8 ? 1 : 2);
  constexpr int BK = rRowBytes / ELT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 1, wm = wid & 1;       // 2(n) x 2(m) waves of 32 x 32
  const int frow = lane & 15, fq = lane >> 4;

  const int tiles_n = N / rT;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;   // n fastest: neighbours share the X rows
  const int m0 = tile_m * rT, n0 = tile_n * rT;
  const int nk = K / BK;

  // accumulators [n-fragment][m-fragment]; lane (frow, fq) holds out[m = .. + frow][n = .. + 4 fq + j]
  r_f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = r_f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto add_residual = [&]() {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int m = m0 + wm * 32 + b * 16 + frow;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const size_t o = static_cast<size_t>(m) * N + n0 + wn * 32 + a * 16 + fq * 4;
        if (epi & EPI_RES_F16) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(residual) + o);
          acc[a][b][0] += f16lo_to_f32(u.x); acc[a][b][1] += f16hi_to_f32(u.x);
          acc[a][b][2] += f16lo_to_f32(u.y); acc[a][b][3] += f16hi_to_f32(u.y);
        } else {
          acc[a][b] += *reinterpret_cast<const r_f32x4_t*>(residual + o);
        }
      }
    }
  };
  // the wide kernel's rule (gemm_wide.hip: res_first): short K without an activation starts from the residual tile
  const bool res_first = (epi & EPI_RESIDUAL) && !(epi & (EPI_QUICKGELU | EPI_GELU | EPI_RELU | EPI_SCALE)) && nk <= 16;
  if (res_first) add_residual();

  // LDS-DMA: a stage = 8 W pieces + 8 X pieces of 1 KiB (8 rows each); wave w moves pieces 2w, 2w+1 of both operands.
  // Lane i of a piece fills (row 8 p + i/8, physical chunk i%8) and fetches logical chunk (i%8) ^ (row & 7).
  const size_t row_stride = static_cast<size_t>(K) * ELT;
  const char* gW[2];
  const char* gX[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wid * 2 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    gW[i] = W + static_cast<size_t>(n0 + row) * row_stride + chunk * 16;
    int xr = m0 + row;
    xr = xr < M ? xr : M - 1;           // rows past M are computed on duplicated data and never stored
    gX[i] = X + static_cast<size_t>(xr) * row_stride + chunk * 16;
  }
  auto stage = [&](int kt) {
    char* base = lds + (kt & (rStages - 1)) * 2 * rHalf;
    const size_t koff = static_cast<size_t>(kt) * rRowBytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((r_gptr_t)(gW[i] + koff), (r_lptr_t)(base + (wid * 2 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((r_gptr_t)(gX[i] + koff), (r_lptr_t)(base + rHalf + (wid * 2 + i) * 1024), 16, 0, 0);
    }
  };

  // three stages in flight; per K-step: counted wait for stage kt (4 pieces per wave and younger stage), raw barrier, refill the
  // buffer that step kt-1 has finished reading, fragments, 8 MFMAs
  stage(0);
  if (nk > 1) stage(1);
  if (nk > 2) stage(2);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < nk) stage(kt + 3);
    const char* tW = lds + (kt & (rStages - 1)) * 2 * rHalf;
    const char* tX = tW + rHalf;
    if constexpr (FP8) {
      // both 16-byte halves of a lane's 32 k (chunks fq and fq + 4, as in the wide kernel) form one 8-register operand
      typedef __attribute__((ext_vector_type(8))) int r_i32x8_t;
      r_i32x8_t fw8[2], fx8[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const r_u32x4_t w0 = *reinterpret_cast<const r_u32x4_t*>(tW + r_swz(wn * 32 + t * 16 + frow, fq));
        const r_u32x4_t w1 = *reinterpret_cast<const r_u32x4_t*>(tW + r_swz(wn * 32 + t * 16 + frow, fq + 4));
        const r_u32x4_t x0 = *reinterpret_cast<const r_u32x4_t*>(tX + r_swz(wm * 32 + t * 16 + frow, fq));
        const r_u32x4_t x1 = *reinterpret_cast<const r_u32x4_t*>(tX + r_swz(wm * 32 + t * 16 + frow, fq + 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fw8[t][j] = static_cast<int>(w0[j]); fw8[t][4 + j] = static_cast<int>(w1[j]);
          fx8[t][j] = static_cast<int>(x0[j]); fx8[t][4 + j] = static_cast<int>(x1[j]);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw8[a], fx8[b], acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 4 + fq;
      r_u32x4_t fw[2], fx[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fw[t] = *reinterpret_cast<const r_u32x4_t*>(tW + r_swz(wn * 32 + t * 16 + frow, chunk));
        fx[t] = *reinterpret_cast<const r_u32x4_t*>(tX + r_swz(wm * 32 + t * 16 + frow, chunk));
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (F32) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(fw[a][s]), __uint_as_float(fx[b][s]), acc[a][b], 0, 0, 0);
          } else {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(r_bf16x8_t, fw[a]),
                                                                __builtin_bit_cast(r_bf16x8_t, fx[b]), acc[a][b], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue: the wide kernel's operations in the wide kernel's order ----
  if (FP8 && (epi & EPI_SCALE)) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const r_f32x4_t cs = *reinterpret_cast<const r_f32x4_t*>(sc.colscale + n0 + wn * 32 + a * 16 + fq * 4) * sc.alpha;
      const r_f32x4_t bv = (epi & EPI_BIAS) ? *reinterpret_cast<const r_f32x4_t*>(bias + n0 + wn * 32 + a * 16 + fq * 4)
                                            : r_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = acc[a][b] * cs + bv;
    }
  } else if (epi & EPI_BIAS) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const r_f32x4_t bv = *reinterpret_cast<const r_f32x4_t*>(bias + n0 + wn * 32 + a * 16 + fq * 4);
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] += bv;
    }
  }
  if (epi & EPI_QUICKGELU) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][b][j] = r_quick_gelu(acc[a][b][j]);
  }
  if (epi & (EPI_GELU | EPI_RELU)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][b][j] = (epi & EPI_GELU) ? gelu_erf(acc[a][b][j]) : fmaxf(acc[a][b][j], 0.f);
  }
  if ((epi & EPI_RESIDUAL) && !res_first) add_residual();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int m = m0 + wm * 32 + b * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const size_t o = static_cast<size_t>(m) * N + n0 + wn * 32 + a * 16 + fq * 4;
      const r_f32x4_t v = acc[a][b];
      if (FP8 && (epi & EPI_OUT_FP8)) {           // the wide kernel's conversion; a lane's 4 consecutive n are one dword
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = fminf(fmaxf(v[j] * sc.oscale, -448.f), 448.f);
        int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
        w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w8, true);
        *reinterpret_cast<int*>(static_cast<char*>(out) + o) = w8;
      } else
      if (epi & EPI_OUT_F16)
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out) + o) = uint2{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])};
      else if (epi & EPI_OUT_BF16)
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out) + o) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      else
        *reinterpret_cast<r_f32x4_t*>(static_cast<float*>(out) + o) = v;
    }
  }
}

// Which launches come here: few rows (the wide kernel would leave most CUs without a tile), an epilogue this kernel has, and
// CMH_GEMM_ROWS != 0 (A/B switch).
static int g_rows_on = -1;   // cmh_set_gemm_rows: -1 = from the environment (CMH_GEMM_ROWS=0 switches it off)
bool gemm_rows_takes(int M, int N, int K, int epi) {
  static const bool env_off = []() { const char* e = getenv("CMH_GEMM_ROWS"); return e && e[0] == '0'; }();
  const bool off = g_rows_on < 0 ? env_off : g_rows_on == 0;
  // up to 2048 rows: a 1 600-row operand (batch 32 of the image tower, configs[0]) has 51..204 wide tiles for 256 CUs; on 64 x 64
  // tiles the same launch fills the chip (configs[0]: validation -10 %, training step -1..3 %; CMH_GEMM_ROWS_MAX_M to compare)
  static const int max_m = []() { const char* e = getenv("CMH_GEMM_ROWS_MAX_M"); return e ? atoi(e) : 2048; }();
  if (off || M > max_m || N % rT != 0) return false;
  if (epi & (EPI_MUL_DQGELU | EPI_SAVE_PRE | 256 | 512)) return false;   // (EPI_SCALE / EPI_OUT_FP8: the e4m3 instantiation, launch_gemm_rows_fp8)
  if ((epi & EPI_OUT_F16) && (epi & EPI_OUT_BF16)) return false;
  (void)K;
  return true;
}

int launch_gemm_rows(int dt, const void* A, const void* W, const float* bias, const float* residual, void* out, int M, int N, int K,
                     int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const int grid = (N / rT) * ((M + rT - 1) / rT);
#define R_GO(KERNEL)                                                                                                         \
  do {                                                                                                                        \
    if (ev0)                                                                                                                  \
      hipExtLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, st, ev0, ev1, 0, static_cast<const char*>(A),                    \
                            static_cast<const char*>(W), bias, residual, out, M, N, K, epi, sc);                              \
    else                                                                                                                      \
      hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, st, static_cast<const char*>(A), static_cast<const char*>(W), bias, \
                         residual, out, M, N, K, epi, sc);                                                                    \
  } while (0)
  const RowsScales sc{nullptr, 1.f, 1.f};
  if (dt == CMH_F32) R_GO(gemm_rows_kernel<0>); else R_GO(gemm_rows_kernel<1>);
  return 0;
}

int launch_gemm_rows_fp8(const void* A, const void* W, const float* colscale, float alpha, const float* bias, const float* residual,
                         void* out, float oscale, int M, int N, int K, int epi, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const int grid = (N / rT) * ((M + rT - 1) / rT);
  const RowsScales sc{colscale, alpha, oscale};
  R_GO(gemm_rows_kernel<2>);
#undef R_GO
  return 0;
}

}  // namespace cmh

extern "C" int cmh_set_gemm_rows(int32_t on) {
  CMH_CHECK_ARG(on >= -1 && on <= 1, "set_gemm_rows: %d (-1 default, 0 off, 1 on)", on);
  cmh::g_rows_on = on;
  return CMH_OK;
}
