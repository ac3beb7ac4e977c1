// Backward building blocks of the towers (SURVEY §8f "next" #2) — the row-wise / element-wise HBM-bound kernels:
//   transpose (+ cast)        wgrad operands: dW = dY^T.X contracts over the M rows, and the GEMM kernels want both
//                             operands contraction-contiguous, so dY and X are transposed (and cast to the GEMM dtype) once
//   column sums               bias gradients  db = sum_m dY[m, :]
//   LayerNorm backward        dx (+= into the gradient stream), dgamma, dbeta       (nn.LayerNorm, model/base/model.py:153-159)
//   QuickGELU forward         mlp = pre * sigmoid(1.702 pre) from the saved pre-activation (model/base/model.py:162-164);
//                             its derivative is fused into the dgrad GEMM's epilogue (EPI_MUL_DQGELU, gemm_wide.hip)
// Data types: the gradient stream is f32; GEMM operands are f32 (f32 mode) or bf16; the forward residual stream x is f32 or
// fp16 (bf16 mode).  "kind": 0 = f32, 1 = bf16, 2 = fp16.
#include "cmh_common.h"

namespace cmh {

// ---- transpose: dst[c, r] = cast(src[r, c]) -------------------------------------------------------------------------
// 64x64 tile per 256-thread workgroup through LDS (row stride 65 words: conflict-free both ways); reads are coalesced
// along c, writes along r.
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ src, void* __restrict__ dst, int R, int C,
                                                        int dst_ld, int skind, int dkind) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? load_as_f32(src, static_cast<size_t>(r) * C + c, skind) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) store_from_f32(dst, static_cast<size_t>(c) * dst_ld + r, dkind, tile[tx][i]);
  }
}

// fast path (C, dst_ld multiples of 4, dst_ld >= R rounded up to 4): 8/16-byte reads along c and writes along r (the last quad
// of a row writes zeros into the padding).  With `colsum` the workgroup also leaves the column sums of its 64 source rows in
// colsum[blockIdx.y][c] (the bias gradient's first stage, for free: the tile is in LDS anyway).
__global__ __launch_bounds__(256) void transpose4_kernel(const void* __restrict__ src, void* __restrict__ dst, int R, int C,
                                                         int dst_ld, int skind, int dkind, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int q = threadIdx.x & 15, p = threadIdx.x >> 4;      // 16 quads per 64-wide row, 16 rows per pass
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = pass * 16 + p, r = r0 + i, c = c0 + q * 4;
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (r < R && c < C) v = load4_as_f32(src, static_cast<size_t>(r) * C + c, skind);
    tile[i][q * 4 + 0] = v.x; tile[i][q * 4 + 1] = v.y; tile[i][q * 4 + 2] = v.z; tile[i][q * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = pass * 16 + p, c = c0 + i, r = r0 + q * 4;
    if (c < C && r < R)
      store4_from_f32(dst, static_cast<size_t>(c) * dst_ld + r, dkind,
                      float4{tile[q * 4 + 0][i], tile[q * 4 + 1][i], tile[q * 4 + 2][i], tile[q * 4 + 3][i]});
  }
  if (colsum && threadIdx.x < 64 && c0 + static_cast<int>(threadIdx.x) < C) {
    float s = 0.f;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) s += tile[i][threadIdx.x];
    colsum[static_cast<size_t>(blockIdx.y) * C + c0 + threadIdx.x] = s;
  }
}

// Up to 4 same-dtype transposes in one launch (the four weight matrices a block's dgrad GEMMs take as W^T: each is 1-5 MB, and as four
// launches the 96 transposes of a training step cost 0.48 ms of mostly launch latency).  R, C multiples of 4; dst_ld = R.
struct TransposeJobs {
  const void* src[4];
  void* dst[4];
  int R[4], C[4], first_tile[5];   // job j owns tiles [first_tile[j], first_tile[j+1]), 64 x 64 each, column-tile fastest
  int n, kind;
};
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeJobs J) {
  __shared__ float tile[64][65];
  int j = 0;
  while (j + 1 < J.n && static_cast<int>(blockIdx.x) >= J.first_tile[j + 1]) ++j;
  const int R = J.R[j], C = J.C[j], tcols = (C + 63) / 64, local = blockIdx.x - J.first_tile[j];
  const int c0 = (local % tcols) * 64, r0 = (local / tcols) * 64;
  const void* src = J.src[j];
  void* dst = J.dst[j];
  const int q = threadIdx.x & 15, p = threadIdx.x >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = pass * 16 + p, r = r0 + i, c = c0 + q * 4;
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (r < R && c < C) v = load4_as_f32(src, static_cast<size_t>(r) * C + c, J.kind);
    tile[i][q * 4 + 0] = v.x; tile[i][q * 4 + 1] = v.y; tile[i][q * 4 + 2] = v.z; tile[i][q * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int i = pass * 16 + p, c = c0 + i, r = r0 + q * 4;
    if (c < C && r < R)
      store4_from_f32(dst, static_cast<size_t>(c) * R + r, J.kind,
                      float4{tile[q * 4 + 0][i], tile[q * 4 + 1][i], tile[q * 4 + 2][i], tile[q * 4 + 3][i]});
  }
}

int launch_transpose_multi(const void* const* src, void* const* dst, const int* R, const int* C, int n, int kind, hipStream_t st) {
  CMH_CHECK_ARG(n >= 1 && n <= 4 && (kind == kF32 || kind == kBF16 || kind == kF16), "transpose_multi: %d jobs", n);
  TransposeJobs J;
  J.n = n; J.kind = kind; J.first_tile[0] = 0;
  for (int j = 0; j < n; ++j) {
    CMH_CHECK_ARG(src[j] && dst[j] && R[j] > 0 && C[j] > 0 && R[j] % 4 == 0 && C[j] % 4 == 0, "transpose_multi: job %d is %d x %d", j, R[j], C[j]);
    J.src[j] = src[j]; J.dst[j] = dst[j]; J.R[j] = R[j]; J.C[j] = C[j];
    J.first_tile[j + 1] = J.first_tile[j] + ((R[j] + 63) / 64) * ((C[j] + 63) / 64);
  }
  for (int j = n; j < 4; ++j) { J.src[j] = nullptr; J.dst[j] = nullptr; J.R[j] = 0; J.C[j] = 0; J.first_tile[j + 1] = J.first_tile[n]; }
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(J.first_tile[n]), dim3(256), 0, st, J);
  CMH_CHECK_LAUNCH("transpose_multi");
  return CMH_OK;
}

// ---- column sums: partial[b, c] = sum over the block's rows; then a second pass over the partials ------------------------
constexpr int kColRows = 64;    // rows per workgroup (= the transpose tile, whose fused column sums share the layout)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ x, int kind, int R, int C,
                                                             float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * kColRows;
  const int r1 = r0 + kColRows < R ? r0 + kColRows : R;
  if (c >= C) return;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += load_as_f32(x, static_cast<size_t>(r) * C + c, kind);
  partial[static_cast<size_t>(blockIdx.y) * C + c] = s;
}
// fast path (C % 4 == 0): 64 column quads x 4 row lanes per workgroup
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const void* __restrict__ x, int kind, int R, int C,
                                                              float* __restrict__ partial) {
  __shared__ float4 red[4][64];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + cg * 4;
  const int r0 = blockIdx.y * kColRows;
  const int r1 = r0 + kColRows < R ? r0 + kColRows : R;
  float4 s = float4{0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float4 v = load4_as_f32(x, static_cast<size_t>(r) * C + c, kind);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[rl][cg] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float4 t = red[0][cg];
#pragma unroll
    for (int k = 1; k < 4; ++k) { t.x += red[k][cg].x; t.y += red[k][cg].y; t.z += red[k][cg].z; t.w += red[k][cg].w; }
    *reinterpret_cast<float4*>(partial + static_cast<size_t>(blockIdx.y) * C + c) = t;
  }
}

// 32 columns x kFinSlices row-slices per workgroup: slice s adds partial rows s, s + kFinSlices, ... in f64 (four independent
// chains so that the loads overlap), the slices are combined in a fixed order -> deterministic.
constexpr int kFinSlices = 32;
__device__ __forceinline__ double colsum_slice(const float* __restrict__ partial, int nb, int C, int c, int sl) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = sl;
  for (; b + 3 * kFinSlices < nb; b += 4 * kFinSlices) {
    s0 += static_cast<double>(partial[static_cast<size_t>(b) * C + c]);
    s1 += static_cast<double>(partial[static_cast<size_t>(b + kFinSlices) * C + c]);
    s2 += static_cast<double>(partial[static_cast<size_t>(b + 2 * kFinSlices) * C + c]);
    s3 += static_cast<double>(partial[static_cast<size_t>(b + 3 * kFinSlices) * C + c]);
  }
  for (; b < nb; b += kFinSlices) s0 += static_cast<double>(partial[static_cast<size_t>(b) * C + c]);
  return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int nb, int C,
                                                            float* __restrict__ out, const float* __restrict__ partial2 = nullptr,
                                                            float* __restrict__ out2 = nullptr) {
  __shared__ double red[kFinSlices][33];
  if (blockIdx.y == 1) { partial = partial2; out = out2; }      // second (partial, out) pair in the same launch
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  red[sl][cl] = c < C ? colsum_slice(partial, nb, C, c, sl) : 0.0;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kFinSlices; ++k) t += red[k][cl];
    out[c] = static_cast<float>(t);
  }
}

// Several (partial, out) pairs in one launch: blockIdx.y names the job (a block of the backward pass queues the final stage of
// its bias / LayerNorm-parameter gradients — six tiny reductions — and launches them once).
__global__ __launch_bounds__(1024) void colsum_final_multi_kernel(FinalJobs jobs) {
  __shared__ double red[kFinSlices][33];
  const int job = blockIdx.y;
  const float* __restrict__ partial = jobs.partial[job];
  const int nb = jobs.slices[job], C = jobs.cols[job];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  if (blockIdx.x * 32 >= C) return;
  red[sl][cl] = c < C ? colsum_slice(partial, nb, C, c, sl) : 0.0;
  __syncthreads();
  if (sl == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kFinSlices; ++k) t += red[k][cl];
    jobs.out[job][c] = static_cast<float>(t);
  }
}

// ---- LayerNorm backward ------------------------------------------------------------------------------------------------------
// y = (x - mean) * rstd * g + b.  With xh = (x - mean) * rstd and a = dy * g:
//   dx = rstd * (a - mean(a) - xh * mean(a * xh)),  dg = sum_rows dy * xh,  db = sum_rows dy.
// One wave per row (lane owns elements lane*4 + 256*j, d <= 1024); a workgroup of 4 waves walks kLnRows rows, every lane keeps
// its dg/db columns in registers, the 4 waves are combined through LDS and one partial row per workgroup goes to HBM
// (column-summed by colsum_final_kernel).
constexpr int kLnRows = 16;   // rows per workgroup: 800 workgroups for a 12 800-row layer (32 rows left the CUs at 1.5 waves per SIMD; 8 rows
                              // double the partial rows the final reduction has to add: measured slower in total)
template <int NV>   // 256-column slices a lane owns: d <= 256 NV (sized to d: a 768-wide row keeps 3 slices of dg / db / gamma, not 4)
__global__ __launch_bounds__(256, (NV <= 3 ? 4 : 1)) void layernorm_bwd_kernel(const void* __restrict__ x, int xkind, const void* __restrict__ dy,
                                                            int dykind, const float* __restrict__ g, float* __restrict__ dx,
                                                            int accumulate, int M, int d, float* __restrict__ pg,
                                                            float* __restrict__ pb, const int32_t* __restrict__ row_index,
                                                            bf16_t* __restrict__ dx_bf16, const bf16_t* __restrict__ acc16,
                                                            int write_f32) {
  __shared__ float red[2][4][1024];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float gam[NV][4], dg[NV][4], db[NV][4];
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = lane * 4 + 256 * j + k;
      gam[j][k] = (e < d) ? g[e] : 0.f;
      dg[j][k] = 0.f; db[j][k] = 0.f;
    }
  const int rbase = blockIdx.x * kLnRows;
  // (round 5, ablation: WITHOUT the two statistics butterflies - as if the forward had saved mean / rstd - the launch takes the same
  // time, 36.1 against 34.7 us at 768 columns: the reductions are not what it waits for)
  // two rows per wave and iteration: both rows' loads are issued before either is reduced, so the second row's memory latency hides
  // under the first row's three butterflies (one row at a time the kernel ran at 2.8 TB/s: 49 us for the 137 MB of a vision layer)
  // acc16 (the 16-bit gradient stream of the bf16 training mode): the sum the new dx is added to comes as bf16 from ANOTHER buffer - its
  // 8 bytes per lane and slice are fetched with the row, not behind the three butterflies
  auto load_row = [&](int row, float (&xv)[NV][4], float (&dv)[NV][4], uint2 (&av)[NV]) -> size_t {
    const size_t xrow = row_index ? static_cast<size_t>(row_index[row]) : static_cast<size_t>(row);   // x / dx row (pooled rows)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e0 = lane * 4 + 256 * j;
      const bool ok = e0 < d;
      const float4 xa = ok ? load4_as_f32(x, xrow * d + e0, xkind) : float4{0.f, 0.f, 0.f, 0.f};
      const float4 da = ok ? load4_as_f32(dy, static_cast<size_t>(row) * d + e0, dykind) : float4{0.f, 0.f, 0.f, 0.f};
      av[j] = (ok && acc16) ? *reinterpret_cast<const uint2*>(acc16 + xrow * d + e0) : uint2{0u, 0u};
      xv[j][0] = xa.x; xv[j][1] = xa.y; xv[j][2] = xa.z; xv[j][3] = xa.w;
      dv[j][0] = da.x; dv[j][1] = da.y; dv[j][2] = da.z; dv[j][3] = da.w;
    }
    return xrow;
  };
  auto do_row = [&](size_t xrow, float (&xv)[NV][4], float (&dv)[NV][4], const uint2 (&av)[NV]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / static_cast<float>(d);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = lane * 4 + 256 * j + k;
        if (e < d) { const float c = xv[j][k] - mean; ss += c * c; }
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss / static_cast<float>(d) + 1e-5f);
    float sa = 0.f, sax = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = lane * 4 + 256 * j + k;
        if (e < d) {
          const float xh = (xv[j][k] - mean) * rstd;
          const float a = dv[j][k] * gam[j][k];
          sa += a; sax += a * xh;
          dg[j][k] += dv[j][k] * xh;
          db[j][k] += dv[j][k];
          xv[j][k] = xh;            // keep xh, a for the dx pass
          dv[j][k] = a;
        }
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sax += __shfl_xor(sax, o, 64); }
    const float ma = sa / static_cast<float>(d), max_ = sax / static_cast<float>(d);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int e0 = lane * 4 + 256 * j;
      if (e0 < d) {
        float4* o = reinterpret_cast<float4*>(dx + xrow * d + e0);
        float4 v = float4{rstd * (dv[j][0] - ma - xv[j][0] * max_), rstd * (dv[j][1] - ma - xv[j][1] * max_),
                          rstd * (dv[j][2] - ma - xv[j][2] * max_), rstd * (dv[j][3] - ma - xv[j][3] * max_)};
        if (accumulate) { const float4 c = *o; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
        else if (acc16) {
          v.x += __uint_as_float(av[j].x << 16); v.y += __uint_as_float(av[j].x & 0xffff0000u);
          v.z += __uint_as_float(av[j].y << 16); v.w += __uint_as_float(av[j].y & 0xffff0000u);
        }
        if (write_f32) *o = v;
        // the next GEMMs take this gradient as a bf16 operand: write it here instead of a cast pass over the f32 stream
        if (dx_bf16) *reinterpret_cast<uint2*>(dx_bf16 + xrow * d + e0) = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
      }
    }
  };
  for (int rr = wid * 2; rr < kLnRows; rr += 8) {
    const int row = rbase + rr;
    if (row >= M) break;
    const bool two = row + 1 < M;
    float xa[NV][4], da[NV][4], xb[NV][4], dbv[NV][4];
    uint2 aa[NV], ab[NV];
    const size_t ra = load_row(row, xa, da, aa);
    const size_t rb = load_row(two ? row + 1 : row, xb, dbv, ab);
    do_row(ra, xa, da, aa);
    if (two) do_row(rb, xb, dbv, ab);
  }
  // combine the 4 waves' dg / db columns, one partial row per workgroup
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = lane * 4 + 256 * j + k;
      if (e < 1024) { red[0][wid][e] = dg[j][k]; red[1][wid][e] = db[j][k]; }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < d; e += 256) {
    pg[static_cast<size_t>(blockIdx.x) * d + e] = (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
    pb[static_cast<size_t>(blockIdx.x) * d + e] = (red[1][0][e] + red[1][1][e]) + (red[1][2][e] + red[1][3][e]);
  }
}

// ---- QuickGELU forward from the saved pre-activation -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void quick_gelu_kernel(const void* __restrict__ pre, void* __restrict__ out, int64_t n, int kind) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const bool vec = (reinterpret_cast<uintptr_t>(pre) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = load4_as_f32(pre, static_cast<size_t>(i) * 4, kind);
    v.x *= __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v.x)); v.y *= __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v.y));
    v.z *= __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v.z)); v.w *= __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v.w));
    store4_from_f32(out, static_cast<size_t>(i) * 4, kind, v);
  }
  for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = load_as_f32(pre, i, kind);
    store_from_f32(out, i, kind, v / (1.0f + __expf(-1.702f * v)));
  }
}

static int kind_ok(int k) { return k == kF32 || k == kBF16 || k == kF16; }

}  // namespace cmh

using namespace cmh;

namespace cmh {
bool transpose_is_vectorised(const void* src, const void* dst, int rows, int cols, int dst_ld) {
  return cols % 4 == 0 && dst_ld % 4 == 0 && dst_ld >= (rows + 3) / 4 * 4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
}

// colsum_partial (optional, vectorised path only — check transpose_is_vectorised first): [ceil(rows / 64), cols] partial column sums
int launch_transpose(const void* src, int skind, void* dst, int dkind, int rows, int cols, int dst_ld, hipStream_t st,
                     float* colsum_partial) {
  CMH_CHECK_ARG(src && dst && rows > 0 && cols > 0 && dst_ld >= rows, "transpose: bad arguments");
  CMH_CHECK_ARG(kind_ok(skind) && kind_ok(dkind), "transpose: bad element kind %d / %d", skind, dkind);
  const bool vec = transpose_is_vectorised(src, dst, rows, cols, dst_ld);
  CMH_CHECK_ARG(vec || !colsum_partial, "transpose: fused column sums need the vectorised path");
  if (vec)
    hipLaunchKernelGGL(transpose4_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, dst, rows, cols, dst_ld,
                       skind, dkind, colsum_partial);
  else
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, st, src, dst, rows, cols, dst_ld,
                       skind, dkind);
  CMH_CHECK_LAUNCH("transpose");
  return CMH_OK;
}
}  // namespace cmh

extern "C" int cmh_transpose(const void* src, int32_t src_kind, void* dst, int32_t dst_kind, int32_t rows, int32_t cols,
                             void* stream) {
  return launch_transpose(src, src_kind, dst, dst_kind, rows, cols, rows, as_stream(stream), nullptr);
}

namespace cmh {
// first stage only: partial [ceil(rows / 64), cols] column sums of 64-row slices (the layout launch_colsum_final / FinalJobs expect)
int launch_colsum_partial(const void* x, int kind, int rows, int cols, float* partial, hipStream_t st) {
  const int nb = (rows + kColRows - 1) / kColRows;
  if (cols % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3((cols + 255) / 256, nb), dim3(256), 0, st, x, kind, rows, cols, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, nb), dim3(256), 0, st, x, kind, rows, cols, partial);
  CMH_CHECK_LAUNCH("colsum_partial");
  return CMH_OK;
}
int launch_colsum_final(const float* partial, int slices, int cols, float* out, hipStream_t st) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 31) / 32), dim3(1024), 0, st, partial, slices, cols, out, nullptr, nullptr);
  CMH_CHECK_LAUNCH("colsum_final");
  return CMH_OK;
}
}  // namespace cmh

extern "C" size_t cmh_colsum_workspace_bytes(int32_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return static_cast<size_t>((rows + kColRows - 1) / kColRows) * cols * 4 + 256;
}

extern "C" int cmh_colsum(const void* x, int32_t kind, int32_t rows, int32_t cols, float* out, void* workspace,
                          size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(x && out && workspace && rows > 0 && cols > 0 && kind_ok(kind), "colsum: bad arguments");
  if (workspace_bytes < cmh_colsum_workspace_bytes(rows, cols)) return fail(CMH_ERR_WORKSPACE, "colsum: workspace too small");
  float* partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const int nb = (rows + kColRows - 1) / kColRows;
  hipStream_t st = as_stream(stream);
  if (cols % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3((cols + 255) / 256, nb), dim3(256), 0, st, x, kind, rows, cols, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, nb), dim3(256), 0, st, x, kind, rows, cols, partial);
  CMH_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 31) / 32), dim3(1024), 0, st, partial, nb, cols, out, nullptr, nullptr);
  CMH_CHECK_LAUNCH("colsum_final");
  return CMH_OK;
}

extern "C" size_t cmh_layernorm_backward_workspace_bytes(int32_t M, int32_t d) {
  if (M <= 0 || d <= 0) return 0;
  return static_cast<size_t>((M + kLnRows - 1) / kLnRows) * d * 8 + 512;
}

namespace cmh {
int launch_layernorm_backward(const void* x, int x_kind, const void* dy, int dy_kind, const float* gamma, const int32_t* row_index,
                              int M, int d, float* dx, int accumulate, float* dgamma, float* dbeta, void* workspace,
                              size_t workspace_bytes, hipStream_t st, void* dx_bf16, FinalJobs* defer, const void* acc16,
                              bool write_f32) {
  CMH_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && workspace && M > 0, "layernorm_backward: bad arguments");
  CMH_CHECK_ARG(!acc16 || (!accumulate && dx_bf16 && acc16 != dx_bf16 && !row_index), "layernorm_backward: the 16-bit sum comes from another buffer, instead of the f32 one");
  CMH_CHECK_ARG(write_f32 || dx_bf16, "layernorm_backward: no output");
  CMH_CHECK_ARG(d % 4 == 0 && d <= 1024, "layernorm_backward: d=%d must be a multiple of 4 and <= 1024", d);
  CMH_CHECK_ARG((x_kind == kF32 || x_kind == kF16) && (dy_kind == kF32 || dy_kind == kBF16), "layernorm_backward: bad kinds");
  if (workspace_bytes < cmh_layernorm_backward_workspace_bytes(M, d)) return fail(CMH_ERR_WORKSPACE, "layernorm_backward: workspace too small");
  const int nb = (M + kLnRows - 1) / kLnRows;
  float* pg = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  float* pb = pg + static_cast<size_t>(nb) * d;
#define LN_BWD(NV)                                                                                                          \
  hipLaunchKernelGGL(layernorm_bwd_kernel<NV>, dim3(nb), dim3(256), 0, st, x, x_kind, dy, dy_kind, gamma, dx, accumulate, M, d, pg, \
                     pb, row_index, static_cast<bf16_t*>(dx_bf16), static_cast<const bf16_t*>(acc16), write_f32 ? 1 : 0)
  if (d <= 256) LN_BWD(1);
  else if (d <= 512) LN_BWD(2);
  else if (d <= 768) LN_BWD(3);
  else LN_BWD(4);
#undef LN_BWD
  CMH_CHECK_LAUNCH("layernorm_backward");
  if (defer && defer->n + 2 <= FinalJobs::kMax) {      // the caller launches the final stage later (the partials stay in `workspace`)
    defer->add(pg, nb, d, dgamma);
    defer->add(pb, nb, d, dbeta);
    return CMH_OK;
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 31) / 32, 2), dim3(1024), 0, st, pg, nb, d, dgamma, pb, dbeta);
  CMH_CHECK_LAUNCH("layernorm_backward dgamma/dbeta");
  return CMH_OK;
}

int launch_final_jobs(FinalJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return CMH_OK;
  int maxc = 0;
  for (int i = 0; i < jobs.n; ++i) maxc = jobs.cols[i] > maxc ? jobs.cols[i] : maxc;
  hipLaunchKernelGGL(colsum_final_multi_kernel, dim3((maxc + 31) / 32, jobs.n), dim3(1024), 0, st, jobs);
  CMH_CHECK_LAUNCH("final reductions");
  jobs.n = 0;
  return CMH_OK;
}
}  // namespace cmh

extern "C" int cmh_layernorm_backward(const void* x, int32_t x_kind, const void* dy, int32_t dy_kind, const float* gamma,
                                      int32_t M, int32_t d, float* dx, int32_t accumulate, float* dgamma, float* dbeta,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return launch_layernorm_backward(x, x_kind, dy, dy_kind, gamma, nullptr, M, d, dx, accumulate, dgamma, dbeta, workspace,
                                   workspace_bytes, as_stream(stream));
}

extern "C" int cmh_quick_gelu(const void* pre, void* out, int64_t n, int32_t kind, void* stream) {
  CMH_CHECK_ARG(pre && out && n > 0 && kind_ok(kind), "quick_gelu: bad arguments");
  const int64_t blocks = (n / 4 + 255) / 256 + 1;
  hipLaunchKernelGGL(quick_gelu_kernel, dim3(static_cast<unsigned>(blocks < 8192 ? blocks : 8192)), dim3(256), 0, as_stream(stream),
                     pre, out, n, kind);
  CMH_CHECK_LAUNCH("quick_gelu");
  return CMH_OK;
}
