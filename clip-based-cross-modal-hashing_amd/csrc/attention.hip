// Multi-head self-attention core of nn.MultiheadAttention as the reference uses it
// (model/base/model.py:171,184-189; head dim 64 because heads = width/64, :284/:437):
//   q scaled by 1/sqrt(64), scores q.k^T, additive -inf causal mask for the text tower (:340-346),
//   optional bool key_padding_mask (MITH trunk, model/MITH.py:25,134), softmax, P.V.
// in_proj and out_proj are GEMMs (gemm.hip); this kernel reads the packed qkv [B*T, 3d] and writes the
// concatenated heads o [B*T, d].
//
// v1: fp32 VALU, flash-style.  One wave per (batch, head, 64-query block); lane = query row with q and
// the output accumulator in registers; K/V tiles of 32 keys staged in LDS as f32 and read as
// wave-broadcast ds_read_b128 (all lanes read the same key row -> conflict-free); one online-softmax
// rescale per key tile.  Sequence lengths here are 50 / <=77, so the whole kernel is ~1 % of the
// encoder FLOPs; exactness (plain fp32 FMA order, accurate expf) matters more.
#include "cmh_common.h"

namespace cmh {

constexpr int HD = 64;   // head dim
constexpr int KT = 32;   // keys per LDS tile

template <typename T>
__device__ __forceinline__ void load_row64(const T* __restrict__ p, float (&dst)[HD], float scale);

template <>
__device__ __forceinline__ void load_row64<float>(const float* __restrict__ p, float (&dst)[HD], float scale) {
#pragma unroll
  for (int i = 0; i < HD / 4; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
    dst[4 * i + 0] = v.x * scale; dst[4 * i + 1] = v.y * scale;
    dst[4 * i + 2] = v.z * scale; dst[4 * i + 3] = v.w * scale;
  }
}
template <>
__device__ __forceinline__ void load_row64<bf16_t>(const bf16_t* __restrict__ p, float (&dst)[HD], float scale) {
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + 8 * i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dst[8 * i + 2 * j + 0] = __uint_as_float(w[j] << 16) * scale;
      dst[8 * i + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u) * scale;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ o, int B, int Tmax,
                                                       int d, int causal, const uint8_t* __restrict__ kpm,
                                                       const int32_t* __restrict__ seq_off) {
  __shared__ __attribute__((aligned(16))) float sK[KT][HD];
  __shared__ __attribute__((aligned(16))) float sV[KT][HD];

  const int lane = threadIdx.x;
  const int heads = d / HD;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  // packed variable-length sequences (encode_text without its padding): sequence b = rows [seq_off[b], seq_off[b+1])
  const int Tn = seq_off ? seq_off[b + 1] - seq_off[b] : Tmax;
  const size_t srow = seq_off ? static_cast<size_t>(seq_off[b]) : static_cast<size_t>(b) * Tmax;
  const int q0 = blockIdx.y * 64;
  const int row = q0 + lane;
  const bool active = row < Tn;
  const size_t ld = static_cast<size_t>(3) * d;
  const T* base = qkv + srow * ld + h * HD;

  float q[HD], acc[HD];
#pragma unroll
  for (int i = 0; i < HD; ++i) acc[i] = 0.f;
  if (active) {
    load_row64<T>(base + static_cast<size_t>(row) * ld, q, 0.125f);   // 1/sqrt(64), exact
  } else {
#pragma unroll
    for (int i = 0; i < HD; ++i) q[i] = 0.f;
  }
  float m = -1e30f, l = 0.f;

  const int last_row = (q0 + 63 < Tn ? q0 + 63 : Tn - 1);
  const int k_end = causal ? last_row + 1 : Tn;   // keys beyond the block's last query are all masked

  for (int k0 = 0; k0 < k_end; k0 += KT) {
    __syncthreads();
    // stage K and V rows k0..k0+KT-1: KT*HD elements each = KT*16 float4 slots, 64 lanes
#pragma unroll
    for (int it = 0; it < KT * HD / 4 / 64; ++it) {
      const int slot = it * 64 + lane;
      const int kr = slot >> 4, c4 = slot & 15;
      const int kg = k0 + kr;
      float4 kv = float4{0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (kg < Tn) {
        const T* kp = base + static_cast<size_t>(kg) * ld + d + c4 * 4;
        const T* vp = kp + d;
        if constexpr (sizeof(T) == 4) {
          kv = *reinterpret_cast<const float4*>(kp);
          vv = *reinterpret_cast<const float4*>(vp);
        } else {
          const uint2 a = *reinterpret_cast<const uint2*>(kp);
          const uint2 c = *reinterpret_cast<const uint2*>(vp);
          kv = float4{__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xffff0000u),
                      __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xffff0000u)};
          vv = float4{__uint_as_float(c.x << 16), __uint_as_float(c.x & 0xffff0000u),
                      __uint_as_float(c.y << 16), __uint_as_float(c.y & 0xffff0000u)};
        }
      }
      *reinterpret_cast<float4*>(&sK[kr][c4 * 4]) = kv;
      *reinterpret_cast<float4*>(&sV[kr][c4 * 4]) = vv;
    }
    __syncthreads();

    float s[KT];
    float tmax = -1e30f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        const float4 kv = *reinterpret_cast<const float4*>(&sK[j][4 * i]);
        a0 = fmaf(q[4 * i + 0], kv.x, a0);
        a1 = fmaf(q[4 * i + 1], kv.y, a1);
        a2 = fmaf(q[4 * i + 2], kv.z, a2);
        a3 = fmaf(q[4 * i + 3], kv.w, a3);
      }
      const int kg = k0 + j;
      bool ok = kg < Tn && (!causal || kg <= row);
      if (kpm && kg < Tn) ok = ok && (kpm[static_cast<size_t>(b) * Tmax + kg] == 0);
      s[j] = ok ? (a0 + a1) + (a2 + a3) : -1e30f;
      tmax = fmaxf(tmax, s[j]);
    }
    const float m_new = fmaxf(m, tmax);
    const float alpha = expf(m - m_new);
    l *= alpha;
#pragma unroll
    for (int i = 0; i < HD; ++i) acc[i] *= alpha;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const float p = s[j] > -1e29f ? expf(s[j] - m_new) : 0.f;
      l += p;
#pragma unroll
      for (int i = 0; i < HD / 4; ++i) {
        const float4 vv = *reinterpret_cast<const float4*>(&sV[j][4 * i]);
        acc[4 * i + 0] = fmaf(p, vv.x, acc[4 * i + 0]);
        acc[4 * i + 1] = fmaf(p, vv.y, acc[4 * i + 1]);
        acc[4 * i + 2] = fmaf(p, vv.z, acc[4 * i + 2]);
        acc[4 * i + 3] = fmaf(p, vv.w, acc[4 * i + 3]);
      }
    }
    m = m_new;
  }

  if (!active) return;
  const float inv = 1.0f / l;
  T* op = o + (srow + row) * d + h * HD;
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i)
      *reinterpret_cast<float4*>(op + 4 * i) =
          float4{acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv};
  } else {
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      uint4 pk;
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        w[j] = static_cast<uint32_t>(f32_to_bf16(acc[8 * i + 2 * j] * inv)) |
               (static_cast<uint32_t>(f32_to_bf16(acc[8 * i + 2 * j + 1] * inv)) << 16);
      pk.x = w[0]; pk.y = w[1]; pk.z = w[2]; pk.w = w[3];
      *reinterpret_cast<uint4*>(op + 8 * i) = pk;
    }
  }
}

// ---- bf16 MFMA path (T <= 128) ---------------------------------------------------------------------------
// One wave per (batch, head).  All K rows sit in registers as MFMA A-fragments (16-byte global loads, no LDS);
// V goes through LDS once to be re-read key-major (the PV product needs V^T fragments) and then also stays in
// registers.  Scores are computed TRANSPOSED, S^T = K.Q^T, so a lane owns 4 keys x 1 query per tile: the row
// softmax is in-lane + two xor-shuffles, and the S^T accumulators are already laid out as the B operand of
// O^T = V^T.P^T (k order permuted identically on both operands: key(g,j) = 32s + 16(j>>2) + 4g + (j&3)).
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8_t;
typedef __attribute__((ext_vector_type(4))) float af32x4_t;

constexpr int kAttnVST = 80;   // LDS row stride in bf16 elements (160 B = 40 dwords): 16-byte rows for the b128 stores, and the eight
                               // key rows one 32-lane half touches in a transposed read start on banks 0,40,16,56,32,8,48,24

// one wave = one (batch, head): `block` is its index among the (batch, head) pairs of its tower; sV: NKT * 16 * kAttnVST bf16 of LDS
template <int NKT>
__device__ __forceinline__ void attention_mfma_body(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, int Tmax, int d, int causal,
                                                    const uint8_t* __restrict__ kpm, const int32_t* __restrict__ seq_off,
                                                    float o8_inv_scale, int block, bf16_t* sV) {
  constexpr int NKS = NKT / 2;
  constexpr int TP = NKT * 16;
  constexpr int VST = kAttnVST;

  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int heads = d / HD;
  const int b = block / heads, h = block - b * heads;
  // (the packed offsets are the same in every lane; said so, the sequence length and the row base stay in scalar registers)
  const int Tn = seq_off ? __builtin_amdgcn_readfirstlane(seq_off[b + 1] - seq_off[b]) : Tmax;
  const size_t srow = seq_off ? static_cast<size_t>(__builtin_amdgcn_readfirstlane(seq_off[b])) : static_cast<size_t>(b) * Tmax;
  const size_t ld = static_cast<size_t>(3) * d;
  const bf16_t* base = qkv + srow * ld + h * HD;
  if (Tn <= 0) return;      // an empty sequence has no rows to read or write (the clamps below would reach row -1); one wave per block

  // Round 4: every load of the prologue is issued before anything waits.  The first form staged V with a rolled loop - one 16-byte
  // load in flight per lane, a wait and an LDS store per iteration: eight (NKT = 4) to sixteen serial memory round trips in front of
  // the first MFMA, ~40 % of the launch - and the key-padding bytes one at a time (sixteen more round trips with a mask).
  // K fragments: A operand rows = keys (they stay in registers)
  abf16x8_t kf[NKT][2];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    int row = kt * 16 + c;
    row = row < Tn ? row : Tn - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      kf[kt][s] = *reinterpret_cast<const abf16x8_t*>(base + static_cast<size_t>(row) * ld + d + s * 32 + g * 8);
  }
  // V -> LDS (rows >= T zeroed so that 0 * garbage can never be NaN): unconditional loads from clamped rows, then the stores
  constexpr int VIT = TP * 8 / 64;      // 16-byte slots per lane
  constexpr int VB = NKT == 8 ? 8 : VIT;   // ... in flight at once (NKT = 8: two batches, or the kernel drops to one wave per SIMD)
  uint4 vst[VB];
  auto load_v = [&](int i0) {
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      const int slot = lane + 64 * (i0 + i), row = slot >> 3, ch = slot & 7;
      const int rc = row < Tn ? row : Tn - 1;
      vst[i] = *reinterpret_cast<const uint4*>(base + static_cast<size_t>(rc) * ld + 2 * d + ch * 8);
    }
  };
  auto store_v = [&](int i0) {
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      const int slot = lane + 64 * (i0 + i), row = slot >> 3, ch = slot & 7;
      *reinterpret_cast<uint4*>(&sV[row * VST + ch * 8]) = row < Tn ? vst[i] : uint4{0u, 0u, 0u, 0u};
    }
  };
  load_v(0);
  // which of this lane's keys (kt, r) -> key = 16kt + 4g + r are usable at all (inside the sequence, not padded), and per key tile
  // whether ALL / NONE of its 16 keys are (wave-uniform): only a mixed tile pays for per-score selects.  The causal mask only ever
  // cuts the diagonal tile, where it is the same for every query tile: key <= query <=> 4g + r <= c.
  uint32_t padded = 0;                  // bit (kt, r): the key is masked by key_padding_mask
  if (kpm) {
    uint8_t kb[NKT][4];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        kb[kt][r] = kpm[static_cast<size_t>(b) * Tmax + (key < Tn ? key : Tn - 1)];      // the mask is [B, Tmax] whatever the row layout (packed rows: seq_off)
      }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) padded |= (kb[kt][r] != 0 ? 1u : 0u) << (kt * 4 + r);
  }
  store_v(0);
  if constexpr (VB < VIT) { load_v(VB); store_v(VB); }
  uint32_t keyok = 0;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const bool ok = key < Tn && !((padded >> (kt * 4 + r)) & 1u);
      keyok |= (ok ? 1u : 0u) << (kt * 4 + r);
    }
  uint32_t tile_all = 0, tile_none = 0;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const uint32_t bits = (keyok >> (kt * 4)) & 15u;
    if (__ballot(bits != 15u) == 0) tile_all |= 1u << kt;
    if (__ballot(bits != 0u) == 0) tile_none |= 1u << kt;
  }
  uint32_t diagok = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) diagok |= (4 * g + r <= c ? 1u : 0u) << r;
  __syncthreads();
  // V^T fragments: A operand rows = head-dim, k = keys in the permuted order.  ds_read_b64_tr_b16 hands lane c of a 16-lane group
  // column c of a 4-key x 16-column block: lane 4q+p supplies the address of key row r0+q, columns 4p..4p+3; two reads (keys
  // 32s+4g.. and 32s+16+4g..) make one fragment - 2 reads instead of 8 ds_read_u16 + packing.
  abf16x8_t vf[NKS][4];
  {
    typedef __attribute__((address_space(3))) void* lptr_t;
    const uint32_t a0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lptr_t)sV)) +
                        static_cast<uint32_t>(((4 * g + (c >> 2)) * VST + 4 * (c & 3)) * 2);
    uint2 raw[NKS][4][2];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                       : "=v"(raw[s][dt][hh])
                       : "v"(a0), "i"((32 * s + 16 * hh) * VST * 2 + 32 * dt));
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(raw[s][0][0]), "+v"(raw[s][0][1]), "+v"(raw[s][1][0]), "+v"(raw[s][1][1]), "+v"(raw[s][2][0]),
                     "+v"(raw[s][2][1]), "+v"(raw[s][3][0]), "+v"(raw[s][3][1]));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        vf[s][dt] = __builtin_bit_cast(abf16x8_t, uint4{raw[s][dt][0].x, raw[s][dt][0].y, raw[s][dt][1].x, raw[s][dt][1].y});
    }
  }

#if defined(ATT_ABL) && ATT_ABL == 3
  {
    float keep = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) keep += static_cast<float>(vf[s][dt][0]);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) keep += static_cast<float>(kf[kt][0][0]) + static_cast<float>(kf[kt][1][0]);
    if (keep == 123.456f) o[0] = 1;
    return;
  }
#endif
  const int nqt = (Tn + 15) >> 4;
  // the NEXT query tile's fragments are fetched while the current tile computes (8 more registers): loaded at the top of its
  // own iteration, each tile's first MFMA waited a full L2 / Infinity-Cache round trip for them
  auto load_q = [&](int qt, abf16x8_t (&q)[2]) {
    const int qrow = qt * 16 + c;
    const int qr = qrow < Tn ? qrow : Tn - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) q[s] = *reinterpret_cast<const abf16x8_t*>(base + static_cast<size_t>(qr) * ld + s * 32 + g * 8);
  };
  abf16x8_t qn[2];
  load_q(0, qn);
  for (int qt = 0; qt < nqt; ++qt) {
    const int qrow = qt * 16 + c;
    abf16x8_t qf[2] = {qn[0], qn[1]};
    load_q(qt + 1, qn);        // rows are clamped to the sequence: the read past the last tile stays in bounds and is never used
                               // (round 4: two tiles ahead - 8 more registers - measured level: 17.5 us either way)
    // raw scores (the 1/sqrt(64) scale is folded into the exponent below): the softmax costs one add + one max, then sub, mul,
    // v_exp, add per score - it is VALU issue, not the matrix pipe, that three waves per SIMD queue for in this kernel
    af32x4_t sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if ((causal && kt > qt) || ((tile_none >> kt) & 1u)) {
        sc[kt] = af32x4_t{-1e30f, -1e30f, -1e30f, -1e30f};
      } else {
        sc[kt] = af32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][s], qf[s], sc[kt], 0, 0, 0);
        uint32_t okb = (keyok >> (kt * 4)) & 15u;
        if (causal && kt == qt) okb &= diagok;
        if (!((tile_all >> kt) & 1u) || (causal && kt == qt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[kt][r] = ((okb >> r) & 1u) ? sc[kt][r] : -1e30f;
        }
      }
    }
    float m = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    // exp((s - m) / 8) = 2^(s c - m c), c = log2(e) / 8: ONE packed FMA per two scores (v_pk_fma_f32) ahead of the v_exp_f32, and the
    // row sums as packed adds - this kernel is VALU-issue bound (three waves per SIMD queue for it), not matrix-pipe bound
    typedef __attribute__((ext_vector_type(2))) float af32x2_t;
    const float kC = 0.18033688011112042f;
    const af32x2_t mc = {-m * kC, -m * kC}, cc = {kC, kC};
    af32x2_t l2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const af32x2_t e = __builtin_elementwise_fma(af32x2_t{sc[kt][r], sc[kt][r + 1]}, cc, mc);   // masked scores (-1e30) give 2^-huge = 0
        const af32x2_t p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        sc[kt][r] = p[0];
        sc[kt][r + 1] = p[1];
        l2 += p;
      }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    l = m > -1e29f ? l : 0.f;                                   // every key masked: no row to normalise (as before: 0 / 0)
    af32x4_t oc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oc[dt] = af32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      abf16x8_t pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = static_cast<__bf16>(sc[2 * s + (j >> 2)][j & 3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s][dt], pb, oc[dt], 0, 0, 0);
    }
#if defined(ATT_ABL) && (ATT_ABL == 1 || ATT_ABL == 2)
    {
      float keep = l;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) keep += oc[dt][0] + oc[dt][1] + oc[dt][2] + oc[dt][3];
      if (keep == 123.456f) o[0] = 1;
      continue;
    }
#endif
    if (qrow < Tn && o8_inv_scale > 0.f) {
      // fp8 mode: the out_proj GEMM's operand leaves as e4m3(o / act_scale), 4 bytes per lane and head-dim tile
      const float inv = o8_inv_scale / l;
      uint8_t* op8 = reinterpret_cast<uint8_t*>(o) + (srow + qrow) * d + h * HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(oc[dt][0] * inv), cl(oc[dt][1] * inv), 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(oc[dt][2] * inv), cl(oc[dt][3] * inv), w, true);
        *reinterpret_cast<uint32_t*>(op8 + 16 * dt) = static_cast<uint32_t>(w);
      }
    } else {
      // bf16 output, 16 bytes per lane and store: v_permlane16_swap exchanges, between the lane rows (g, g+1), the packed words of
      // two neighbouring head-dim tiles, so that an even lane row owns 8 consecutive dims of tile 2p and an odd one 8 consecutive
      // dims of tile 2p+1 - two 16-byte stores per lane and query tile (64-byte runs per row) instead of four 8-byte ones: the
      // kernel's store tail was issue-bound (profiles/r03_attention_ablation.txt: 6 of 20 us; staging the tile through LDS to write
      // whole 128-byte rows measured no better: 18.1 against 17.8 us)
      const float inv = 1.0f / l;
      bf16_t* op = o + (srow + (qrow < Tn ? qrow : Tn - 1)) * d + h * HD + (g & 1) * 16 + (g & 2) * 4;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          lo[w] = pack_bf16x2(oc[2 * pr][2 * w] * inv, oc[2 * pr][2 * w + 1] * inv);
          hi[w] = pack_bf16x2(oc[2 * pr + 1][2 * w] * inv, oc[2 * pr + 1][2 * w + 1] * inv);
        }
        typedef __attribute__((ext_vector_type(2))) unsigned au2_t;
        const au2_t s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
        const au2_t s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
        if (qrow < Tn) *reinterpret_cast<uint4*>(op + 32 * pr) = uint4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}

template <int NKT>
__global__ __launch_bounds__(64) void attention_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            int B, int Tmax, int d, int causal,
                                                            const uint8_t* __restrict__ kpm,
                                                            const int32_t* __restrict__ seq_off, float o8_inv_scale) {
  __shared__ __attribute__((aligned(16))) bf16_t sV[NKT * 16 * kAttnVST];
  (void)B;
  attention_mfma_body<NKT>(qkv, o, Tmax, d, causal, kpm, seq_off, o8_inv_scale, blockIdx.x, sV);
}

template <int NKT>
static void launch_attention_mfma(const void* qkv, void* o, int B, int T, int d, int causal, const uint8_t* kpm,
                                  const int32_t* seq_off, hipStream_t st, float o8_inv_scale) {
  hipLaunchKernelGGL(attention_mfma_kernel<NKT>, dim3(B * (d / HD)), dim3(64), 0, st, static_cast<const bf16_t*>(qkv),
                     static_cast<bf16_t*>(o), B, T, d, causal, kpm, seq_off, o8_inv_scale);
}

int launch_attention(const void* qkv, void* o, int dt, int B, int T, int d, int causal,
                     const uint8_t* key_padding_mask, hipStream_t st) {
  return launch_attention_varlen(qkv, o, dt, B, T, d, causal, key_padding_mask, nullptr, st, 0.f);
}

// seq_off (device int32 [B+1], may be NULL): packed variable-length sequences, T = the longest
int launch_attention_varlen(const void* qkv, void* o, int dt, int B, int T, int d, int causal,
                            const uint8_t* key_padding_mask, const int32_t* seq_off, hipStream_t st, float o8_inv_scale) {
  CMH_CHECK_ARG(d % HD == 0, "attention: width %d is not a multiple of 64", d);
  CMH_CHECK_ARG(B > 0 && T > 0, "attention: empty batch");
  CMH_CHECK_ARG(o8_inv_scale == 0.f || (dt == CMH_BF16 && T <= 128), "attention: e4m3 output needs bf16 qkv and T <= 128 (T=%d)", T);
  const dim3 grid(B * (d / HD), (T + 63) / 64);
  if (dt == CMH_BF16 && T <= 128) {
    if (T <= 32) launch_attention_mfma<2>(qkv, o, B, T, d, causal, key_padding_mask, seq_off, st, o8_inv_scale);
    else if (T <= 64) launch_attention_mfma<4>(qkv, o, B, T, d, causal, key_padding_mask, seq_off, st, o8_inv_scale);
    else if (T <= 96) launch_attention_mfma<6>(qkv, o, B, T, d, causal, key_padding_mask, seq_off, st, o8_inv_scale);
    else launch_attention_mfma<8>(qkv, o, B, T, d, causal, key_padding_mask, seq_off, st, o8_inv_scale);
    CMH_CHECK_LAUNCH("attention_mfma");
    return CMH_OK;
  }
  if (dt == CMH_F32)
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(64), 0, st, static_cast<const float*>(qkv),
                       static_cast<float*>(o), B, T, d, causal, key_padding_mask, seq_off);
  else
    hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(64), 0, st, static_cast<const bf16_t*>(qkv),
                       static_cast<bf16_t*>(o), B, T, d, causal, key_padding_mask, seq_off);
  CMH_CHECK_LAUNCH("attention");
  return CMH_OK;
}

}  // namespace cmh
