// What the matrix pipe sustains on this chip with nothing else in the way: N back-to-back MFMAs per wave on registers,
// 1 or 2 waves per SIMD, every CU busy.  Prints TFLOP/s and the shader clock seen by s_memtime (ticks / wall time).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* sink, unsigned long long* ticks, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  unsigned long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) r += c[j][0] + c[j][3];
  } else {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) r += c[j][0] + c[j][7];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  if (r == 12345.678f) sink[0] = r;
}

template <int SHAPE>
void run(int threads, int iters, float* sink, unsigned long long* ticks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256, reps = 20;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(threads), 0, 0, sink, ticks, iters);
  hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(threads), 0, 0, sink, ticks, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double per = SHAPE == 16 ? 32.0 * 2 * 16 * 16 * 32 : 16.0 * 2 * 32 * 32 * 16;   // flops per loop iteration per wave
  const double flops = per * iters * (threads / 64) * grid * reps;
  const double us = ms * 1e3 / reps;
  printf("mfma %s  %d waves/SIMD  %7.1f us/launch  %7.1f TFLOP/s   s_memtime %.0f ticks/launch -> %.0f MHz tick rate\n",
         SHAPE == 16 ? "16x16x32" : "32x32x16", threads / 256, us, flops / (ms * 1e-3) / 1e12, (double)t, t / us);
}

int main() {
  float* sink; unsigned long long* ticks;
  hipMalloc(&sink, 4); hipMalloc(&ticks, 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int threads : {256, 512}) {
      run<16>(threads, 20000, sink, ticks);     // ~0.64M MFMAs per wave: a few ms per launch
      run<32>(threads, 10000, sink, ticks);
    }
  return 0;
}
