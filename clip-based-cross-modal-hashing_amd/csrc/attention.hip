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
__global__ __launch_bounds__(64) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ o, int B, int Tn,
                                                       int d, int causal, const uint8_t* __restrict__ kpm) {
  __shared__ __attribute__((aligned(16))) float sK[KT][HD];
  __shared__ __attribute__((aligned(16))) float sV[KT][HD];

  const int lane = threadIdx.x;
  const int heads = d / HD;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int q0 = blockIdx.y * 64;
  const int row = q0 + lane;
  const bool active = row < Tn;
  const size_t ld = static_cast<size_t>(3) * d;
  const T* base = qkv + static_cast<size_t>(b) * Tn * ld + h * HD;

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
      if (kpm && kg < Tn) ok = ok && (kpm[static_cast<size_t>(b) * Tn + kg] == 0);
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
  T* op = o + (static_cast<size_t>(b) * Tn + row) * d + h * HD;
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

int launch_attention(const void* qkv, void* o, int dt, int B, int T, int d, int causal,
                     const uint8_t* key_padding_mask, hipStream_t st) {
  CMH_CHECK_ARG(d % HD == 0, "attention: width %d is not a multiple of 64", d);
  CMH_CHECK_ARG(B > 0 && T > 0, "attention: empty batch");
  const dim3 grid(B * (d / HD), (T + 63) / 64);
  if (dt == CMH_F32)
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(64), 0, st, static_cast<const float*>(qkv),
                       static_cast<float*>(o), B, T, d, causal, key_padding_mask);
  else
    hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(64), 0, st, static_cast<const bf16_t*>(qkv),
                       static_cast<bf16_t*>(o), B, T, d, causal, key_padding_mask);
  CMH_CHECK_LAUNCH("attention");
  return CMH_OK;
}

}  // namespace cmh
