// Multi-head attention backward for the towers' packed qkv layout (nn.MultiheadAttention inside ResidualAttentionBlock,
// model/base/model.py:171-189): given qkv [B*T, 3d], the forward output o [B*T, d] and do [B*T, d], produce dqkv [B*T, 3d].
// Per (batch, head), hd = 64, scale = 1/8:
//   P = softmax(scale * Q K^T + mask)        (recomputed, nothing but qkv and o is saved)
//   dV = P^T dO;   D_i = dO_i . O_i (= sum_j P_ij dP_ij);   dS = scale * P o (dO V^T - D);   dQ = dS K;   dK = dS^T Q
// First version: fp32 VALU, one 256-thread workgroup per (batch, head), T <= 128, everything staged in LDS
// (two [T][65] operand buffers + one [T][T+1] score buffer, <= 132 KB).  Inputs f32 or bf16, accumulation f32, outputs in the
// input type.  An MFMA version for the bf16 mode is the next step (this one costs ~0.2-0.3 ms per layer at batch 256).
#include <cstdlib>

#include "cmh_common.h"

namespace cmh {

constexpr int HDB = 64;

template <typename T> __device__ __forceinline__ float ab_ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ab_ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ab_ld<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void ab_st(T* p, size_t i, float v);
template <> __device__ __forceinline__ void ab_st<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void ab_st<bf16_t>(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }

template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                            const T* __restrict__ dout, T* __restrict__ dqkv, int B, int Tmax,
                                                            int d, int causal, const uint8_t* __restrict__ kpm,
                                                            const int32_t* __restrict__ seq_off) {
  extern __shared__ float smem[];
  // packed variable-length sequences: sequence b = rows [seq_off[b], seq_off[b+1]) (LDS is sized for Tmax)
  const int heads0 = d / HDB, b0 = blockIdx.x / heads0;
  const int Tn = seq_off ? seq_off[b0 + 1] - seq_off[b0] : Tmax;
  float* bufA = smem;                       // [Tn][65]
  float* bufB = bufA + Tn * 65;             // [Tn][65]
  float* S = bufB + Tn * 65;                // [Tn][Tn+1]
  float* Dv = S + Tn * (Tn + 1);            // [Tn]
  const int ST = Tn + 1;
  const int heads = d / HDB;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const size_t ld = static_cast<size_t>(3) * d;
  const size_t rowq = seq_off ? static_cast<size_t>(seq_off[b]) : static_cast<size_t>(b) * Tmax;
  const T* qb = qkv + rowq * ld + h * HDB;            // q: +0, k: +d, v: +2d
  const T* ob = o + rowq * d + h * HDB;
  const T* dob = dout + rowq * d + h * HDB;
  T* dqb = dqkv + rowq * ld + h * HDB;
  const int tid = threadIdx.x;

  auto load_tile = [&](float* buf, const T* base, size_t stride) {
    for (int idx = tid; idx < Tn * HDB; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      buf[r * 65 + c] = ab_ld<T>(base, static_cast<size_t>(r) * stride + c);
    }
  };
  // ---- 1. S = scale * Q K^T, masked ------------------------------------------------------------------------------------
  load_tile(bufA, qb, ld);
  load_tile(bufB, qb + d, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * Tn; idx += 256) {
    const int i = idx / Tn, j = idx - i * Tn;
    float acc = 0.f;
#pragma unroll 16
    for (int c = 0; c < HDB; ++c) acc += bufA[i * 65 + c] * bufB[j * 65 + c];
    bool ok = !(causal && j > i);
    if (ok && kpm) ok = kpm[static_cast<size_t>(b) * Tmax + j] == 0;
    S[i * ST + j] = ok ? acc * 0.125f : -1e30f;
  }
  __syncthreads();
  // ---- 2. row softmax (one wave per row) -------------------------------------------------------------------------------
  {
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < Tn; i += 4) {
      float m = -1e30f;
      for (int j = lane; j < Tn; j += 64) m = fmaxf(m, S[i * ST + j]);
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) m = fmaxf(m, __shfl_xor(m, of, 64));
      float l = 0.f;
      for (int j = lane; j < Tn; j += 64) {
        const float s = S[i * ST + j];
        const float p = s > -1e29f ? expf(s - m) : 0.f;
        S[i * ST + j] = p;
        l += p;
      }
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) l += __shfl_xor(l, of, 64);
      const float inv = 1.0f / l;
      for (int j = lane; j < Tn; j += 64) S[i * ST + j] *= inv;
    }
  }
  __syncthreads();
  // ---- 3. A <- dO, B <- V;  dV = P^T dO;  D_i = dO_i . O_i;  dS = scale * P o (dO V^T - D) ------------------------------
  load_tile(bufA, dob, d);
  load_tile(bufB, qb + 2 * d, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * HDB; idx += 256) {           // dV[j][c]
    const int j = idx >> 6, c = idx & 63;
    float acc = 0.f;
    for (int i = 0; i < Tn; ++i) acc += S[i * ST + j] * bufA[i * 65 + c];
    ab_st<T>(dqb + 2 * d, static_cast<size_t>(j) * ld + c, acc);
  }
  {
    const int lane = tid & 63, wid = tid >> 6;
    for (int i = wid; i < Tn; i += 4) {
      float v = bufA[i * 65 + lane] * ab_ld<T>(ob, static_cast<size_t>(i) * d + lane);
#pragma unroll
      for (int of = 32; of > 0; of >>= 1) v += __shfl_xor(v, of, 64);
      if (lane == 0) Dv[i] = v;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < Tn * Tn; idx += 256) {
    const int i = idx / Tn, j = idx - i * Tn;
    float acc = 0.f;
#pragma unroll 16
    for (int c = 0; c < HDB; ++c) acc += bufA[i * 65 + c] * bufB[j * 65 + c];
    S[i * ST + j] = S[i * ST + j] * (acc - Dv[i]) * 0.125f;
  }
  __syncthreads();
  // ---- 4. B <- K: dQ = dS K;  A <- Q: dK = dS^T Q -------------------------------------------------------------------------
  load_tile(bufB, qb + d, ld);
  load_tile(bufA, qb, ld);
  __syncthreads();
  for (int idx = tid; idx < Tn * HDB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float aq = 0.f, ak = 0.f;
    for (int t = 0; t < Tn; ++t) {
      aq += S[r * ST + t] * bufB[t * 65 + c];      // dQ[r][c] = sum_j dS[r][j] K[j][c]
      ak += S[t * ST + r] * bufA[t * 65 + c];      // dK[r][c] = sum_i dS[i][r] Q[i][c]
    }
    ab_st<T>(dqb, static_cast<size_t>(r) * ld + c, aq);
    ab_st<T>(dqb + d, static_cast<size_t>(r) * ld + c, ak);
  }
}

}  // namespace cmh

// ============================================================================================================================
// bf16 MFMA version (T <= 96): one workgroup per (batch, head).  Q, K, V, dO are staged ONCE, row-major [rows][64] in LDS; the five
// products  out[i][j] = sum_k A[i][k] B[j][k]  run on v_mfma_f32_16x16x32_bf16:
//   S  [q][key] = Q . K^T      (A = Q,  B = K,  k = head dim)           dP [q][key] = dO . V^T   (A = dO, B = V)
//   dV^T[hd][key] = sum_q   dO[q][hd]  P^T[key][q]                       dK^T[hd][key] = sum_q Q[q][hd] dS^T[key][q]
//   dQ^T[hd][q]   = sum_key K[key][hd] dS^T[key][q]
// The three output products contract over ROWS of the staged matrices: their A fragments (and dQ's B fragment) come out of the
// row-major images through ds_read_b64_tr_b16 (lane c of a 16-lane group gets column c of a 4-row x 16-column block; two reads per
// fragment, k order 4g..4g+3, 16+4g..16+4g+3) - no transposed copies (round 1 wrote Q^T, K^T, dO^T with 24 two-byte LDS stores per
// 16-byte slot: a third of the kernel), and the operand read the plain way takes the same k order as two 8-byte reads.
// A wave owns query tiles: it keeps the S and dP accumulators of its 16 queries against all keys in registers, does the row softmax,
// D_q = sum_key P dP and dS = P (dP - D) / 8 there, and writes P^T and dS^T ([key][q], bf16, 8-byte stores) for the second phase.
namespace cmh {

typedef __attribute__((ext_vector_type(8))) __bf16 ab_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ab_f32x4_t;

__device__ __forceinline__ ab_f32x4_t ab_mma(const bf16_t* A, int lda, int arow0, const bf16_t* Bm, int ldb, int brow0, int ksteps,
                                             int g, int c) {
  ab_f32x4_t acc = ab_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < ksteps; ++ks) {
    const ab_bf16x8_t a = *reinterpret_cast<const ab_bf16x8_t*>(A + static_cast<size_t>(arow0 + c) * lda + ks * 32 + g * 8);
    const ab_bf16x8_t b = *reinterpret_cast<const ab_bf16x8_t*>(Bm + static_cast<size_t>(brow0 + c) * ldb + ks * 32 + g * 8);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// fragment of the TRANSPOSE of a row-major image M[k][col]: "row" col0 + c, k = k0 + {4g..4g+3, 16+4g..16+4g+3}
// (every lane of the wave must be active: the read gathers across the 16 lanes of a group)
// rows >= `rows` do not exist in the image (the contraction is padded to a multiple of 32): those lanes point at `zero_addr`, 8 zero bytes
__device__ __forceinline__ ab_bf16x8_t ab_frag_tr(uint32_t lds_base, int ld, int k0, int col0, int g, int c, int rows, uint32_t zero_addr) {
  const int r0 = k0 + 4 * g + (c >> 2);
  const uint32_t off = static_cast<uint32_t>((r0 * ld + col0 + 4 * (c & 3)) * 2);
  const uint32_t a0 = r0 < rows ? lds_base + off : zero_addr;
  const uint32_t a1 = r0 + 16 < rows ? lds_base + off + static_cast<uint32_t>(16 * ld * 2) : zero_addr;
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
  return __builtin_bit_cast(ab_bf16x8_t, uint4{lo.x, lo.y, hi.x, hi.y});
}
// the same, reads only (the caller waits once for several fragments); by value - a fragment handed out through references went to scratch
struct AbHalves { uint2 lo, hi; };
__device__ __forceinline__ AbHalves ab_frag_tr_issue(uint32_t lds_base, int ld, int k0, int col0, int g, int c, int rows, uint32_t zero_addr) {
  const int r0 = k0 + 4 * g + (c >> 2);
  const uint32_t off = static_cast<uint32_t>((r0 * ld + col0 + 4 * (c & 3)) * 2);
  const uint32_t a0 = r0 < rows ? lds_base + off : zero_addr;
  const uint32_t a1 = r0 + 16 < rows ? lds_base + off + static_cast<uint32_t>(16 * ld * 2) : zero_addr;
  AbHalves h;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h.lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h.hi) : "v"(a1));
  return h;
}
// fragment of a row-major image M[row][k] in the same k order: two 8-byte reads
__device__ __forceinline__ ab_bf16x8_t ab_frag_perm(const bf16_t* M, int ld, int row, int k0, int g) {
  const uint2 lo = *reinterpret_cast<const uint2*>(M + static_cast<size_t>(row) * ld + k0 + 4 * g);
  const uint2 hi = *reinterpret_cast<const uint2*>(M + static_cast<size_t>(row) * ld + k0 + 16 + 4 * g);
  return __builtin_bit_cast(ab_bf16x8_t, uint4{lo.x, lo.y, hi.x, hi.y});
}

__device__ __forceinline__ float ab_row_max(float v) {      // over the 16 lanes c of a lane group g
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float ab_row_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}

template <int NT> struct AbwdShape {
  static constexpr int TR = NT * 16, KP = ((NT + 1) / 2) * 32;
  static constexpr int LR = NT <= 4 ? 80 : 72;   // leading dimension of the [rows][64] images: 160-byte rows keep the transposing reads of a
                                                 // 32-lane half on distinct banks (T <= 64); 144-byte rows (partly 2-way) let two workgroups
                                                 // of the longer captions share a CU
  static constexpr int LT = KP + 8;              // leading dimension of P^T / dS^T [key][q]
  static constexpr bool ALIAS = NT <= 4;         // every wave owns at most one query tile: P^T takes over V's slot after phase 1
  static constexpr int kTotal = 4 * TR * LR + (ALIAS ? 1 : 2) * TR * LT;      // bf16 elements
  static_assert(!ALIAS || LT <= LR, "P^T must fit V's slot");
};

template <int NT>    // row tiles: TR = 16 * NT rows, contraction over rows padded to KP = 32 * ceil(NT / 2)
__global__ __launch_bounds__(NT <= 4 ? 256 : 512) void attention_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                 bf16_t* __restrict__ dqkv, int B, int Tmax, int d, int causal,
                                                                 const uint8_t* __restrict__ kpm, const int32_t* __restrict__ seq_off) {
  using SH = AbwdShape<NT>;
  constexpr int TR = SH::TR, KP = SH::KP, LR = SH::LR, LT = SH::LT, kTotal = SH::kTotal;
  constexpr bool ALIAS = SH::ALIAS;
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  bf16_t* sQ = sm;                   // [TR][LR]
  bf16_t* sK = sQ + TR * LR;
  bf16_t* sV = sK + TR * LR;
  bf16_t* sDO = sV + TR * LR;
  bf16_t* sDST = sDO + TR * LR;      // [TR][LT]   dS^T [key][q]
  bf16_t* sPT = ALIAS ? sV : sDST + TR * LT;       // [TR][LT]   P^T  [key][q]

  // NT > 4 (captions longer than 64 tokens): 8 waves share the query tiles
  constexpr int NW = NT <= 4 ? 4 : 8, NTH = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, c = lane & 15;
  const int heads = d / HDB;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const size_t ld = static_cast<size_t>(3) * d;
  const int Tn = seq_off ? seq_off[b + 1] - seq_off[b] : Tmax;
  const size_t row0 = seq_off ? static_cast<size_t>(seq_off[b]) : static_cast<size_t>(b) * Tmax;
  const bf16_t* qb = qkv + row0 * ld + h * HDB;
  const bf16_t* dob = dout + row0 * d + h * HDB;
  bf16_t* dqb = dqkv + row0 * ld + h * HDB;

  // ---- stage: zero everything (padding rows / columns must be exact zeros), then the real rows.  Round 4: the rows' global loads are
  // issued FIRST (every 16-byte slot of Q, K, V, dO a thread owns: 8 loads in flight), the zero fill runs under their latency, and the
  // rows are stored after the barrier - the first form zeroed, waited, and only then loaded, slot by slot.
  constexpr int NIT = (TR * 8 + NTH - 1) / NTH;              // slots per thread and image (1 or 2)
  // (named registers, not arrays: the arrays of the first form of this went through scratch)
  static_assert(NIT == 1 || NIT == 2, "one or two slots per thread");
  const int slot0 = tid, r0s = slot0 >> 3, ch0 = slot0 & 7, rc0 = r0s < Tn ? r0s : Tn - 1;      // clamped: the loads are unconditional
  const int slot1 = tid + NTH, r1s = slot1 >> 3, ch1 = slot1 & 7, rc1 = r1s < Tn ? r1s : Tn - 1;
  const uint4 q0 = *reinterpret_cast<const uint4*>(qb + static_cast<size_t>(rc0) * ld + ch0 * 8);
  const uint4 k0 = *reinterpret_cast<const uint4*>(qb + d + static_cast<size_t>(rc0) * ld + ch0 * 8);
  const uint4 v0 = *reinterpret_cast<const uint4*>(qb + 2 * d + static_cast<size_t>(rc0) * ld + ch0 * 8);
  const uint4 o0 = *reinterpret_cast<const uint4*>(dob + static_cast<size_t>(rc0) * d + ch0 * 8);
  uint4 q1 = q0, k1 = k0, v1 = v0, o1 = o0;
  if constexpr (NIT == 2) {
    q1 = *reinterpret_cast<const uint4*>(qb + static_cast<size_t>(rc1) * ld + ch1 * 8);
    k1 = *reinterpret_cast<const uint4*>(qb + d + static_cast<size_t>(rc1) * ld + ch1 * 8);
    v1 = *reinterpret_cast<const uint4*>(qb + 2 * d + static_cast<size_t>(rc1) * ld + ch1 * 8);
    o1 = *reinterpret_cast<const uint4*>(dob + static_cast<size_t>(rc1) * d + ch1 * 8);
  }
  for (int i = tid; i < kTotal / 8; i += NTH) reinterpret_cast<uint4*>(sm)[i] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();
  if (r0s < Tn) {
    *reinterpret_cast<uint4*>(sQ + r0s * LR + ch0 * 8) = q0;
    *reinterpret_cast<uint4*>(sK + r0s * LR + ch0 * 8) = k0;
    *reinterpret_cast<uint4*>(sV + r0s * LR + ch0 * 8) = v0;
    *reinterpret_cast<uint4*>(sDO + r0s * LR + ch0 * 8) = o0;
  }
  if (NIT == 2 && r1s < Tn) {
    *reinterpret_cast<uint4*>(sQ + r1s * LR + ch1 * 8) = q1;
    *reinterpret_cast<uint4*>(sK + r1s * LR + ch1 * 8) = k1;
    *reinterpret_cast<uint4*>(sV + r1s * LR + ch1 * 8) = v1;
    *reinterpret_cast<uint4*>(sDO + r1s * LR + ch1 * 8) = o1;
  }
  __syncthreads();

  // ---- phase 1: per query tile: S, dP, softmax, D, dS -> P^T, dS^T ------------------------------------------------------------
  const int nrt = (Tn + 15) >> 4;        // row tiles that hold real rows
  auto phase1_compute = [&](int ti, ab_f32x4_t (&s)[NT], ab_f32x4_t (&dp)[NT], float (&dsum)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      if (tj < nrt) {
        s[tj] = ab_mma(sQ, LR, ti * 16, sK, LR, tj * 16, 2, g, c);       // [q = 16ti + 4g + r][key = 16tj + c]
        dp[tj] = ab_mma(sDO, LR, ti * 16, sV, LR, tj * 16, 2, g, c);
      } else {
        s[tj] = ab_f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[tj] = s[tj];
      }
    }
    float m[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      const int key = tj * 16 + c;
      bool kok = key < Tn;
      if (kok && kpm) kok = kpm[static_cast<size_t>(b) * Tmax + key] == 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = ti * 16 + 4 * g + r;
        const bool ok = kok && !(causal && key > q);
        s[tj][r] = ok ? s[tj][r] * 0.125f : -1e30f;
        m[r] = fmaxf(m[r], s[tj][r]);
      }
    }
    float l[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = ab_row_max(m[r]);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = s[tj][r] > -1e29f ? __expf(s[tj][r] - m[r]) : 0.f;
        s[tj][r] = p;
        l[r] += p;
      }
    dsum[0] = dsum[1] = dsum[2] = dsum[3] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) l[r] = 1.0f / ab_row_sum(l[r]);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[tj][r] *= l[r]; dsum[r] += s[tj][r] * dp[tj][r]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) dsum[r] = ab_row_sum(dsum[r]);
  };
  auto phase1_write = [&](int ti, ab_f32x4_t (&s)[NT], ab_f32x4_t (&dp)[NT], float (&dsum)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      if (tj >= nrt) break;
      const int key = tj * 16 + c, q0 = ti * 16 + 4 * g;
      float dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dsv[r] = s[tj][r] * (dp[tj][r] - dsum[r]) * 0.125f;
      *reinterpret_cast<uint2*>(sPT + key * LT + q0) = uint2{pack_bf16x2(s[tj][0], s[tj][1]), pack_bf16x2(s[tj][2], s[tj][3])};
      *reinterpret_cast<uint2*>(sDST + key * LT + q0) = uint2{pack_bf16x2(dsv[0], dsv[1]), pack_bf16x2(dsv[2], dsv[3])};
    }
  };
  if constexpr (ALIAS) {
    ab_f32x4_t s[NT], dp[NT];
    float dsum[4];
    if (wid < nrt) phase1_compute(wid, s, dp, dsum);
    __syncthreads();                      // every wave is done with V: its slot becomes P^T
    // the padding of P^T must be exact zeros where phase 2 reads it: rows >= Tn and columns >= Tn
    for (int i = tid; i < TR * LR / 8; i += NTH) reinterpret_cast<uint4*>(sV)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (wid < nrt) phase1_write(wid, s, dp, dsum);
  } else {
    for (int ti = wid; ti < nrt; ti += NW) {
      ab_f32x4_t s[NT], dp[NT];
      float dsum[4];
      phase1_compute(ti, s, dp, dsum);
      phase1_write(ti, s, dp, dsum);
    }
  }
  __syncthreads();

  // ---- phase 2: dV, dK (contraction over queries), dQ (over keys); out[hd = 16th + 4g + r][row = 16tr + c] --------------------
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t aDO = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)sDO)), aQ = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)sQ)),
                 aK = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)sK)), aDST = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)sDST));
  const uint32_t aZero = aQ + 64 * 2;      // columns 64.. of Q's row 0: padding, zero since the first pass, never written
  constexpr int KS = KP / 32;
  for (int job = wid; job < 3 * nrt * 4; job += NW) {          // wave-uniform trip count: the transposing reads need every lane
    const int prod = job / (nrt * 4), rem = job - prod * nrt * 4, tr = rem >> 2, th = rem & 3;
    const uint32_t aA = prod == 0 ? aDO : (prod == 1 ? aQ : aK);
    const bf16_t* Bn = prod == 0 ? sPT : sDST;
    ab_f32x4_t acc = ab_f32x4_t{0.f, 0.f, 0.f, 0.f};
    // every fragment read of the job is issued before the one wait (a wait per fragment put 2 KS LDS round trips in front of each MFMA)
    uint2 alo[KS], ahi[KS], blo[KS], bhi[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const AbHalves ha = ab_frag_tr_issue(aA, LR, ks * 32, th * 16, g, c, TR, aZero);          // dO^T / Q^T / K^T [hd][row k]
      alo[ks] = ha.lo; ahi[ks] = ha.hi;
      if (prod == 2) {
        const AbHalves hb = ab_frag_tr_issue(aDST, LT, ks * 32, tr * 16, g, c, TR, aZero);      // dS [q][key k] = (dS^T)^T
        blo[ks] = hb.lo; bhi[ks] = hb.hi;
      } else {                                                                                  // P^T / dS^T [key][q k]
        blo[ks] = *reinterpret_cast<const uint2*>(Bn + static_cast<size_t>(tr * 16 + c) * LT + ks * 32 + 4 * g);
        bhi[ks] = *reinterpret_cast<const uint2*>(Bn + static_cast<size_t>(tr * 16 + c) * LT + ks * 32 + 16 + 4 * g);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(alo[ks]), "+v"(ahi[ks]), "+v"(blo[ks]), "+v"(bhi[ks]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const ab_bf16x8_t fa = __builtin_bit_cast(ab_bf16x8_t, uint4{alo[ks].x, alo[ks].y, ahi[ks].x, ahi[ks].y});
      const ab_bf16x8_t fb = __builtin_bit_cast(ab_bf16x8_t, uint4{blo[ks].x, blo[ks].y, bhi[ks].x, bhi[ks].y});
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    }
    const int row = tr * 16 + c;
    if (row < Tn) {
      bf16_t* op = dqb + (prod == 0 ? 2 * d : (prod == 1 ? d : 0)) + static_cast<size_t>(row) * ld + th * 16 + 4 * g;
      *reinterpret_cast<uint2*>(op) = uint2{pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])};
    }
  }
}

template <int NT>
static int launch_attention_bwd_mfma(const void* qkv, const void* dout, void* dqkv, int B, int T, int d, int causal,
                                     const uint8_t* kpm, const int32_t* seq_off, hipStream_t st) {
  const size_t lds = static_cast<size_t>(AbwdShape<NT>::kTotal) * 2;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_mfma_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          static_cast<int>(lds)) != hipSuccess) return fail(CMH_ERR_LAUNCH, "attention_backward: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(attention_bwd_mfma_kernel<NT>, dim3(B * (d / HDB)), dim3(NT <= 4 ? 256 : 512), lds, st, static_cast<const bf16_t*>(qkv),
                     static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dqkv), B, T, d, causal, kpm, seq_off);
  return CMH_OK;
}

int launch_attention_bwd_bf16(const void* qkv, const void* dout, void* dqkv, int B, int T, int d, int causal, const uint8_t* kpm,
                              const int32_t* seq_off, hipStream_t st) {
  const int nt = (T + 15) / 16;
  switch (nt) {
    case 1: return launch_attention_bwd_mfma<1>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    case 2: return launch_attention_bwd_mfma<2>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    case 3: return launch_attention_bwd_mfma<3>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    case 4: return launch_attention_bwd_mfma<4>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    case 5: return launch_attention_bwd_mfma<5>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    case 6: return launch_attention_bwd_mfma<6>(qkv, dout, dqkv, B, T, d, causal, kpm, seq_off, st);
    default: return 1;       // T > 96: the LDS image (208 / 230 KB) does not fit -> caller uses the fp32 VALU kernel
  }
}

}  // namespace cmh

using namespace cmh;

namespace cmh {
int launch_attention_backward(int dtype, const void* qkv, const void* o, const void* dout, void* dqkv, int B, int T, int d, int causal,
                              const uint8_t* key_padding_mask, const int32_t* seq_off, hipStream_t st) {
  CMH_CHECK_ARG(qkv && o && dout && dqkv && B > 0 && T > 0, "attention_backward: bad arguments");
  CMH_CHECK_ARG(dtype == CMH_F32 || dtype == CMH_BF16, "attention_backward: bad dtype");
  CMH_CHECK_ARG(d % HDB == 0, "attention_backward: width %d is not a multiple of 64", d);
  CMH_CHECK_ARG(T <= 128, "attention_backward: T=%d > 128 is not built (both CLIP towers have T <= 77)", T);
  const size_t lds = (static_cast<size_t>(2) * T * 65 + static_cast<size_t>(T) * (T + 1) + T) * 4;
  const dim3 grid(B * (d / HDB));
  if (dtype == CMH_F32) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess) return fail(CMH_ERR_LAUNCH, "attention_backward: LDS");
    hipLaunchKernelGGL(attention_bwd_kernel<float>, grid, dim3(256), lds, st, static_cast<const float*>(qkv),
                       static_cast<const float*>(o), static_cast<const float*>(dout), static_cast<float*>(dqkv), B, T, d, causal,
                       key_padding_mask, seq_off);
  } else if (T <= 96 && !getenv("CMH_ATTN_BWD_VALU")) {
    const int rc = launch_attention_bwd_bf16(qkv, dout, dqkv, B, T, d, causal, key_padding_mask, seq_off, st);
    if (rc) return rc;
  } else {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds)) != hipSuccess) return fail(CMH_ERR_LAUNCH, "attention_backward: LDS");
    hipLaunchKernelGGL(attention_bwd_kernel<bf16_t>, grid, dim3(256), lds, st, static_cast<const bf16_t*>(qkv),
                       static_cast<const bf16_t*>(o), static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dqkv), B, T, d,
                       causal, key_padding_mask, seq_off);
  }
  CMH_CHECK_LAUNCH("attention_backward");
  return CMH_OK;
}
}  // namespace cmh

extern "C" int cmh_attention_backward(int32_t dtype, const void* qkv, const void* o, const void* dout, void* dqkv, int32_t B,
                                      int32_t T, int32_t d, int32_t causal, const uint8_t* key_padding_mask, void* stream) {
  return cmh::launch_attention_backward(dtype, qkv, o, dout, dqkv, B, T, d, causal, key_padding_mask, nullptr, as_stream(stream));
}
