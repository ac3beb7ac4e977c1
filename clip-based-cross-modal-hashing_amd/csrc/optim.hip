// BertAdam as ONE fused multi-tensor step (SURVEY §8f "next" #1) — replaces the per-parameter loop of the reference's
// model/base/optimization.py:103-168, which issues ~12 small kernels for each of CLIP ViT-B/32's 302 parameter tensors
// (clip_grad_norm_, mul_, add_, mul_, addcmul_, sqrt, add, div, mul, add_, ...) every step.
//
// Per tensor t with gradient g (optimization.py lines in brackets):
//   c      = min(1, max_grad_norm / (||g||_2 + 1e-6))   and g *= c in place      [135-136: clip_grad_norm_(p, max_norm)]
//   m      = m*b1 + (1-b1)*g                                                      [141]
//   v      = v*b2 + (1-b2)*g*g                                                    [143]
//   update = m / (sqrt(v) + e) + weight_decay * p                                 [144, 153-154]
//   p     -= lr_scheduled * update                                                [163-164]; lr_scheduled is computed by the
// host mirror (python double, like the reference) and arrives per tensor.
//
// HBM-bound: 4 streams in (p, g, m, v), 4 out, 32 B per element — 151 M parameters = 4.8 GB per step.  Three launches:
//   adam_sumsq_kernel : one 256-thread workgroup per 16 Ki-element chunk, 16-byte loads, f32 tree inside the workgroup;
//   adam_coef_kernel  : one workgroup per tensor adds its chunk partials in f64 by a fixed tree (deterministic: no float
//                       atomics) -> clip coefficient;
//   adam_update_kernel: same chunking; every f32 operation is an explicit *_rn intrinsic in the reference's operation order
//                       (mul then fused-multiply-add for m as ATen's add(alpha) does, mul/mul/add for v, IEEE sqrt and
//                       divide), so the step is bit-comparable to the CPU reference up to the norm's summation order.
#include <vector>

#include "cmh_common.h"

namespace cmh {

constexpr int kAdamChunk = 16384;   // elements per workgroup

struct AdamTensorDev {
  float* p; float* g; float* m; float* v;
  long long n;
  float lr, wd, max_norm;
  int first_chunk;                  // index of this tensor's first chunk
  bf16_t* p_bf16;                   // optional: the encoder's bf16 GEMM copy of p, rewritten here (no cast pass per step)
};

struct AdamChunk { int tensor; int index; };   // chunk `index` of tensor `tensor`

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const AdamTensorDev* __restrict__ T, const AdamChunk* __restrict__ C,
                                                         float* __restrict__ partial) {
  __shared__ float red[4];
  const AdamChunk c = C[blockIdx.x];
  const AdamTensorDev t = T[c.tensor];
  if (!(t.max_norm > 0.f)) return;
  const long long lo = static_cast<long long>(c.index) * kAdamChunk;
  const long long hi = lo + kAdamChunk < t.n ? lo + kAdamChunk : t.n;
  const float* g = t.g;
  float s = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const long long v4 = (hi - lo) / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g + lo);
    for (long long i = threadIdx.x; i < v4; i += 256) {
      const float4 x = g4[i];
      s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    for (long long i = lo + v4 * 4 + threadIdx.x; i < hi; i += 256) s += g[i] * g[i];
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += 256) s += g[i] * g[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__device__ __forceinline__ void adam_elem(float& p, float& g, float& m, float& v, float coef, bool clip, float b1, float omb1,
                                          float b2, float omb2, float eps, float wd, float lr) {
  if (clip) g = __fmul_rn(g, coef);
  m = __fmaf_rn(g, omb1, __fmul_rn(m, b1));
  v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(omb2, g), g));
  float upd = __fdiv_rn(m, __fadd_rn(__fsqrt_rn(v), eps));
  if (wd > 0.f) upd = __fadd_rn(upd, __fmul_rn(wd, p));
  p = __fadd_rn(p, -__fmul_rn(lr, upd));
}

// clip coefficient of every tensor: its chunk partials summed in f64 by a fixed tree (thread i takes partials i, i+256, ...)
__global__ __launch_bounds__(256) void adam_coef_kernel(const AdamTensorDev* __restrict__ T, const float* __restrict__ partial,
                                                        float* __restrict__ coef) {
  __shared__ double red[256];
  const AdamTensorDev t = T[blockIdx.x];
  if (!(t.max_norm > 0.f)) { if (threadIdx.x == 0) coef[blockIdx.x] = 1.f; return; }
  const int nch = static_cast<int>((t.n + kAdamChunk - 1) / kAdamChunk);
  double ss = 0.0;
  for (int i = threadIdx.x; i < nch; i += 256) ss += static_cast<double>(partial[t.first_chunk + i]);
  red[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = static_cast<float>(sqrt(red[0]));
    const float cc = __fdiv_rn(t.max_norm, __fadd_rn(norm, 1e-6f));     // torch: max_norm / (total_norm + 1e-6), clamp(max=1)
    coef[blockIdx.x] = cc < 1.f ? cc : 1.f;
  }
}

__global__ __launch_bounds__(256) void adam_update_kernel(const AdamTensorDev* __restrict__ T, const AdamChunk* __restrict__ C,
                                                          const float* __restrict__ coefs, float b1, float omb1, float b2,
                                                          float omb2, float eps) {
  const AdamChunk c = C[blockIdx.x];
  const AdamTensorDev t = T[c.tensor];
  const float coef = t.max_norm > 0.f ? coefs[c.tensor] : 1.f;
  // clip_grad_norm_ scales p.grad in place only when the norm exceeds max_norm (coefficient clamped to 1): a tensor inside its norm
  // keeps its gradient bits - g * 1.0f is g - so neither the multiply nor the 4-byte write-back per element happens for it
  // (per-parameter norms of 1.0: most tensors of most steps; 0.6 GB of 4.8 GB per step)
  const bool clip = coef < 1.f;
  const long long lo = static_cast<long long>(c.index) * kAdamChunk;
  const long long hi = lo + kAdamChunk < t.n ? lo + kAdamChunk : t.n;
  const bool al = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                    reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
  long long tail = lo;
  if (al) {
    const long long v4 = (hi - lo) / 4;
    float4* p4 = reinterpret_cast<float4*>(t.p + lo); float4* g4 = reinterpret_cast<float4*>(t.g + lo);
    float4* m4 = reinterpret_cast<float4*>(t.m + lo); float4* w4 = reinterpret_cast<float4*>(t.v + lo);
    for (long long i = threadIdx.x; i < v4; i += 256) {
      float4 p = p4[i], g = g4[i], m = m4[i], v = w4[i];
      adam_elem(p.x, g.x, m.x, v.x, coef, clip, b1, omb1, b2, omb2, eps, t.wd, t.lr);
      adam_elem(p.y, g.y, m.y, v.y, coef, clip, b1, omb1, b2, omb2, eps, t.wd, t.lr);
      adam_elem(p.z, g.z, m.z, v.z, coef, clip, b1, omb1, b2, omb2, eps, t.wd, t.lr);
      adam_elem(p.w, g.w, m.w, v.w, coef, clip, b1, omb1, b2, omb2, eps, t.wd, t.lr);
      p4[i] = p; m4[i] = m; w4[i] = v;
      if (clip) g4[i] = g;
      if (t.p_bf16) *reinterpret_cast<uint2*>(t.p_bf16 + lo + i * 4) = uint2{pack_bf16x2(p.x, p.y), pack_bf16x2(p.z, p.w)};
    }
    tail = lo + v4 * 4;
  }
  for (long long i = tail + threadIdx.x; i < hi; i += 256) {
    float p = t.p[i], g = t.g[i], m = t.m[i], v = t.v[i];
    adam_elem(p, g, m, v, coef, clip, b1, omb1, b2, omb2, eps, t.wd, t.lr);
    t.p[i] = p; t.m[i] = m; t.v[i] = v;
    if (clip) t.g[i] = g;
    if (t.p_bf16) t.p_bf16[i] = f32_to_bf16(p);
  }
}

static size_t adam_chunks(const cmh_adam_tensor* t, int count) {
  size_t c = 0;
  for (int i = 0; i < count; ++i) c += static_cast<size_t>((t[i].n + kAdamChunk - 1) / kAdamChunk);
  return c;
}

}  // namespace cmh

using namespace cmh;

extern "C" size_t cmh_bert_adam_workspace_bytes(int32_t count, int64_t total_elems) {
  if (count <= 0 || total_elems <= 0) return 0;
  const size_t chunks = static_cast<size_t>(total_elems / kAdamChunk) + static_cast<size_t>(count);
  return align_up(sizeof(AdamTensorDev) * count, 256) + align_up(sizeof(AdamChunk) * chunks, 256) + align_up(4 * chunks, 256) +
         align_up(4 * static_cast<size_t>(count), 256) + 256;
}

extern "C" int cmh_bert_adam_step(const cmh_adam_tensor* tensors, int32_t count, double b1, double b2, double eps,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(tensors && count > 0 && workspace, "bert_adam_step: null pointer / empty list");
  CMH_CHECK_ARG(b1 >= 0.0 && b1 < 1.0 && b2 >= 0.0 && b2 < 1.0 && eps >= 0.0, "bert_adam_step: bad b1/b2/eps");
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    CMH_CHECK_ARG(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n > 0,
                  "bert_adam_step: tensor %d has a null pointer or no elements", i);
    total += tensors[i].n;
  }
  const size_t need = cmh_bert_adam_workspace_bytes(count, total);
  if (workspace_bytes < need) return fail(CMH_ERR_WORKSPACE, "bert_adam_step: workspace %zu < %zu bytes", workspace_bytes, need);
  const size_t chunks = adam_chunks(tensors, count);
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  AdamTensorDev* dT = reinterpret_cast<AdamTensorDev*>(ws);
  AdamChunk* dC = reinterpret_cast<AdamChunk*>(ws + align_up(sizeof(AdamTensorDev) * count, 256));
  float* dP = reinterpret_cast<float*>(reinterpret_cast<char*>(dC) + align_up(sizeof(AdamChunk) * chunks, 256));
  float* dK = reinterpret_cast<float*>(reinterpret_cast<char*>(dP) + align_up(4 * chunks, 256));
  // Host tables go through two pinned staging buffers used in turn: a buffer is rewritten only after the event recorded behind
  // its previous upload (two calls ago) has passed, so the call never waits for the stream — the tables are built and queued
  // while the GPU is still busy with the backward pass.
  struct Staging {
    void* buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int cur = 0;
  };
  static thread_local Staging sg;
  hipStream_t st = as_stream(stream);
  sg.cur ^= 1;
  const int k = sg.cur;
  if (sg.ev[k] && hipEventSynchronize(sg.ev[k]) != hipSuccess) return fail(CMH_ERR_LAUNCH, "bert_adam_step: staging event wait failed");
  const size_t tb = align_up(sizeof(AdamTensorDev) * count, 256), cb = sizeof(AdamChunk) * chunks;
  if (sg.cap[k] < tb + cb) {
    if (sg.buf[k]) (void)hipHostFree(sg.buf[k]);
    sg.cap[k] = (tb + cb) * 5 / 4;
    if (hipHostMalloc(&sg.buf[k], sg.cap[k], hipHostMallocDefault) != hipSuccess) {
      sg.buf[k] = nullptr; sg.cap[k] = 0;
      return fail(CMH_ERR_LAUNCH, "bert_adam_step: pinned staging allocation failed");
    }
  }
  if (!sg.ev[k] && hipEventCreateWithFlags(&sg.ev[k], hipEventDisableTiming) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "bert_adam_step: event creation failed");
  AdamTensorDev* hT = reinterpret_cast<AdamTensorDev*>(sg.buf[k]);
  AdamChunk* hC = reinterpret_cast<AdamChunk*>(static_cast<char*>(sg.buf[k]) + tb);
  size_t ci = 0;
  for (int i = 0; i < count; ++i) {
    hT[i] = AdamTensorDev{tensors[i].p, tensors[i].g, tensors[i].m, tensors[i].v, tensors[i].n, tensors[i].lr,
                          tensors[i].weight_decay, tensors[i].max_grad_norm, static_cast<int>(ci),
                          static_cast<bf16_t*>(tensors[i].p_bf16)};
    const int nch = static_cast<int>((tensors[i].n + kAdamChunk - 1) / kAdamChunk);
    for (int c = 0; c < nch; ++c) hC[ci++] = AdamChunk{i, c};
  }
  if (hipMemcpyAsync(dT, hT, sizeof(AdamTensorDev) * count, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(dC, hC, cb, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(sg.ev[k], st) != hipSuccess)
    return fail(CMH_ERR_LAUNCH, "bert_adam_step: table upload failed");
  // the reference forms 1 - beta in python doubles and ATen rounds each scalar to f32 once (optimization.py:141,143)
  const float omb1 = static_cast<float>(1.0 - b1), omb2 = static_cast<float>(1.0 - b2);
  hipLaunchKernelGGL(adam_sumsq_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, st, dT, dC, dP);
  CMH_CHECK_LAUNCH("adam_sumsq");
  hipLaunchKernelGGL(adam_coef_kernel, dim3(static_cast<unsigned>(count)), dim3(256), 0, st, dT, dP, dK);
  CMH_CHECK_LAUNCH("adam_coef");
  hipLaunchKernelGGL(adam_update_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, st, dT, dC, dK, static_cast<float>(b1), omb1, static_cast<float>(b2),
                     omb2, static_cast<float>(eps));
  CMH_CHECK_LAUNCH("adam_update");
  return CMH_OK;
}
