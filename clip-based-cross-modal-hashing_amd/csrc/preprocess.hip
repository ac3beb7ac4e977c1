// Image side of the input pipeline on the GPU (reference dataset/base.py:35-44, :55-64):
//   train: Resize(R, BICUBIC) -> CenterCrop(R) -> ToTensor -> Normalize      eval: Resize((R, R), BICUBIC) -> ToTensor -> Normalize
// on a ragged batch of decoded RGB images (uint8 HWC, back to back in one buffer).  The resampler is Pillow's
// ImagingResample for 8-bit channels, bit for bit (libImaging/Resample.c; the parity tests pin it to Pillow 12.2 itself):
// per axis double-precision bicubic (a = -0.5) weights over a support scaled by the shrink factor, normalised, rounded to
// 22-bit fixed point; horizontal pass into a uint8 intermediate, vertical pass from it.  Only the pixels of the centre crop
// (and the rows the vertical pass will read) are produced.  Three HBM-bound launches per batch:
//   1. prep_coeffs_kernel   one thread per output index per axis: window + integer weights   (double, every operation rounded once)
//   2. prep_horizontal_kernel   one workgroup per 8 source rows: rows -> LDS (a word per pixel), R x 3 weighted sums -> uint8 [H, R, 3]
//   3. prep_vertical_kernel     one workgroup per output row: weighted sums down the columns -> uint8 -> (x / 255 - mean) / std, CHW f32
#include "cmh_common.h"

#pragma clang fp contract(off)

namespace cmh {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct PrepImage {   // per image, written by the coefficient kernel
  int32_t row_lo, row_hi;   // source rows the vertical pass reads
};

__device__ __forceinline__ double prep_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return __dadd_rn(__dmul_rn(__dmul_rn(__dsub_rn(__dmul_rn(a + 2.0, x), a + 3.0), x), x), 1.0);
  if (x < 2.0) return __dmul_rn(__dsub_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dsub_rn(x, 5.0), x), 8.0), x), 4.0), a);
  return 0.0;
}

// resized size of one axis and the crop origin on it
__device__ __forceinline__ void prep_axis_geometry(int h, int w, int R, int train, int axis, int* in_size, int* out_size, int* origin) {
  const int in = axis == 0 ? h : w;
  int out = R;
  if (train) {   // torchvision _compute_resized_output_size: short edge -> R, long edge -> int(R * long / short); w <= h: width is short
    const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
    const int new_long = static_cast<int>(static_cast<double>(static_cast<long long>(R) * lng) / static_cast<double>(shrt));
    out = ((axis == 1) == (w <= h)) ? R : new_long;
  }
  const int d = out - R;                      // int(round(d / 2.0)): half to even
  *in_size = in;
  *out_size = out;
  *origin = (d >> 1) + ((d & 1) & ((d >> 1) & 1));
}

// grid (B, 2): axis 0 = vertical (source rows), axis 1 = horizontal.  coef [B][2][KS][R] (tap-major: neighbouring outputs read
// neighbouring words), bounds [B][2][R][2] = (first source index, taps).
__global__ __launch_bounds__(256) void prep_coeffs_kernel(const int32_t* __restrict__ hw, int R, int train, int KS,
                                                          int32_t* __restrict__ coef, int32_t* __restrict__ bounds,
                                                          PrepImage* __restrict__ info) {
  const int b = blockIdx.x, axis = blockIdx.y;
  const int h = hw[2 * b], w = hw[2 * b + 1];
  int in_size, out_size, origin;
  prep_axis_geometry(h, w, R, train, axis, &in_size, &out_size, &origin);
  const double scale = __ddiv_rn(static_cast<double>(in_size), static_cast<double>(out_size));
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = __dmul_rn(2.0, filterscale);
  const double ss = __ddiv_rn(1.0, filterscale);
  int32_t* kc = coef + (static_cast<size_t>(b) * 2 + axis) * KS * R;
  int32_t* bd = bounds + (static_cast<size_t>(b) * 2 + axis) * R * 2;
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    const int xx = j + origin;
    const double center = __dmul_rn(static_cast<double>(xx) + 0.5, scale);
    int xmin = static_cast<int>(__dadd_rn(__dsub_rn(center, support), 0.5));
    if (xmin < 0) xmin = 0;
    int xmax = static_cast<int>(__dadd_rn(__dadd_rn(center, support), 0.5));
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > KS) xmax = KS;   // cannot happen: KS = ceil(support) * 2 + 1 of the largest image (host-checked)
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x)
      ww = __dadd_rn(ww, prep_bicubic(__dmul_rn(__dadd_rn(__dsub_rn(static_cast<double>(x + xmin), center), 0.5), ss)));
    for (int x = 0; x < KS; ++x) {
      int32_t q = 0;
      if (x < xmax) {
        double v = prep_bicubic(__dmul_rn(__dadd_rn(__dsub_rn(static_cast<double>(x + xmin), center), 0.5), ss));
        if (ww != 0.0) v = __ddiv_rn(v, ww);
        const double s = __dmul_rn(v, static_cast<double>(1 << kPrecisionBits));
        q = v < 0.0 ? static_cast<int32_t>(__dadd_rn(-0.5, s)) : static_cast<int32_t>(__dadd_rn(0.5, s));
      }
      kc[static_cast<size_t>(x) * R + j] = q;
    }
    bd[2 * j] = xmin;
    bd[2 * j + 1] = xmax;
    if (axis == 0) {
      if (j == 0) info[b].row_lo = xmin;
      if (j == R - 1) info[b].row_hi = xmin + xmax;
    }
  }
}

__device__ __forceinline__ int prep_clip8(int v) {
  v >>= kPrecisionBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

constexpr int kPrepRows = 8;   // source rows per workgroup of the horizontal pass

// grid (ceil(max_h / 8), B), block 256: source rows 8*blockIdx.x .. +7 of image b -> tmp[b][y][j][c], j < R.
// The rows are staged in LDS as one 32-bit word per pixel (R | G << 8 | B << 16): a tap is one ds_read_b32 instead of three
// byte reads, and every weight (loaded once per tap) serves 8 rows x 3 channels = 24 independent accumulators.
__global__ __launch_bounds__(256) void prep_horizontal_kernel(const uint8_t* __restrict__ pixels, const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ hw, int R, int KS,
                                                              const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                              const PrepImage* __restrict__ info, uint8_t* __restrict__ tmp,
                                                              size_t tmp_stride, int pitch) {
  extern __shared__ uint32_t rows[];   // [8][pitch]
  const int b = blockIdx.y, y0 = blockIdx.x * kPrepRows;
  const int h = hw[2 * b], w = hw[2 * b + 1];
  const int lo = info[b].row_lo, hi = info[b].row_hi < h ? info[b].row_hi : h;
  if (y0 >= hi || y0 + kPrepRows <= lo) return;
  const uint8_t* img = pixels + offsets[b];
  for (int r = 0; r < kPrepRows; ++r) {
    int y = y0 + r;
    y = y < h ? y : h - 1;                       // rows past the image repeat the last one (never stored)
    const uint8_t* src = img + static_cast<size_t>(y) * w * 3;
    for (int x = threadIdx.x; x < w; x += blockDim.x)
      rows[r * pitch + x] = static_cast<uint32_t>(src[3 * x]) | (static_cast<uint32_t>(src[3 * x + 1]) << 8) |
                            (static_cast<uint32_t>(src[3 * x + 2]) << 16);
  }
  __syncthreads();
  const int32_t* kc = coef + (static_cast<size_t>(b) * 2 + 1) * KS * R;
  const int32_t* bd = bounds + (static_cast<size_t>(b) * 2 + 1) * R * 2;
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    const int xmin = bd[2 * j], n = bd[2 * j + 1];
    int acc[kPrepRows][3];
#pragma unroll
    for (int r = 0; r < kPrepRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < n; ++x) {
      const int k = kc[static_cast<size_t>(x) * R + j];
#pragma unroll
      for (int r = 0; r < kPrepRows; ++r) {
        const uint32_t v = rows[r * pitch + xmin + x];
        acc[r][0] += static_cast<int>(v & 255u) * k;
        acc[r][1] += static_cast<int>((v >> 8) & 255u) * k;
        acc[r][2] += static_cast<int>(v >> 16) * k;
      }
    }
#pragma unroll
    for (int r = 0; r < kPrepRows; ++r) {
      const int y = y0 + r;
      if (y >= lo && y < hi) {
        uint8_t* dst = tmp + static_cast<size_t>(b) * tmp_stride + (static_cast<size_t>(y) * R + j) * 3;
        dst[0] = static_cast<uint8_t>(prep_clip8(acc[r][0]));
        dst[1] = static_cast<uint8_t>(prep_clip8(acc[r][1]));
        dst[2] = static_cast<uint8_t>(prep_clip8(acc[r][2]));
      }
    }
  }
}

// grid (R, B), block 256: output row i of image b.  The pass is element-wise along the row, so a thread takes 4 consecutive
// BYTES of the interleaved row (one 32-bit load per tap; R*3 is padded to words by the row pitch), the finished uint8 row goes
// through LDS and leaves as three coalesced float planes.
__global__ __launch_bounds__(256) void prep_vertical_kernel(const uint8_t* __restrict__ tmp, size_t tmp_stride, int R, int KS,
                                                            const int32_t* __restrict__ coef, const int32_t* __restrict__ bounds,
                                                            float m0, float m1, float m2, float d0, float d1, float d2,
                                                            float* __restrict__ out, uint8_t* __restrict__ out_u8) {
  __shared__ uint32_t line[(1024 * 3 + 3) / 4];
  const int b = blockIdx.y, i = blockIdx.x;
  const int32_t* kc = coef + (static_cast<size_t>(b) * 2 + 0) * KS * R;
  const int32_t* bd = bounds + (static_cast<size_t>(b) * 2 + 0) * R * 2;
  const int ymin = bd[2 * i], n = bd[2 * i + 1];
  const int row_bytes = R * 3, words = (row_bytes + 3) >> 2;
  const uint8_t* src = tmp + static_cast<size_t>(b) * tmp_stride + static_cast<size_t>(ymin) * row_bytes;
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | static_cast<uintptr_t>(row_bytes)) & 3) == 0;
  for (int q = threadIdx.x; q < words; q += blockDim.x) {
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int y = 0; y < n; ++y) {
      const int k = kc[static_cast<size_t>(y) * R + i];
      const uint8_t* p = src + static_cast<size_t>(y) * row_bytes + 4 * q;
      uint32_t v;
      if (aligned) v = *reinterpret_cast<const uint32_t*>(p);
      else {                                   // bytes past the row's end belong to the next row / the padding: never stored
        const int left = row_bytes - 4 * q;
        v = p[0] | (left > 1 ? static_cast<uint32_t>(p[1]) << 8 : 0u) | (left > 2 ? static_cast<uint32_t>(p[2]) << 16 : 0u) |
            (left > 3 ? static_cast<uint32_t>(p[3]) << 24 : 0u);
      }
      s0 += static_cast<int>(v & 255u) * k; s1 += static_cast<int>((v >> 8) & 255u) * k;
      s2 += static_cast<int>((v >> 16) & 255u) * k; s3 += static_cast<int>(v >> 24) * k;
    }
    // hipcc (ROCm 7.2) fuses shift + clamp + pack of two values into v_ashr_pk_u8_i32, which writes only the low 16 bits of
    // its destination, and then ORs the other bytes into the stale upper half.  The explicit masks keep each pair clean.
    uint32_t lo = static_cast<uint32_t>(prep_clip8(s0)) | (static_cast<uint32_t>(prep_clip8(s1)) << 8);
    uint32_t hi = static_cast<uint32_t>(prep_clip8(s2)) | (static_cast<uint32_t>(prep_clip8(s3)) << 8);
    asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(lo));
    asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(hi));
    line[q] = lo | (hi << 16);
  }
  __syncthreads();
  const uint8_t* lb = reinterpret_cast<const uint8_t*>(line);
  const size_t plane = static_cast<size_t>(R) * R;
  if (out_u8) {
    uint8_t* q8 = out_u8 + (static_cast<size_t>(b) * R + i) * row_bytes;
    for (int q = threadIdx.x; q < row_bytes; q += blockDim.x) q8[q] = lb[q];
  }
  if (out) {
    for (int j = threadIdx.x; j < R; j += blockDim.x) {
      float* o = out + static_cast<size_t>(b) * 3 * plane + static_cast<size_t>(i) * R + j;
      o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j]), 255.0f), m0), d0);
      o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j + 1]), 255.0f), m1), d1);
      o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j + 2]), 255.0f), m2), d2);
    }
  }
}

// ToTensor + Normalize of already resized images: out[b] = (u8[rows ? rows[b] : b] / 255 - mean) / std, HWC uint8 -> CHW f32.
// The tail of the vertical pass on its own: with the resized uint8 images of a whole dataset resident in HBM (a 25 k-image
// set is 3.8 GB at 224 x 224) every epoch after the first is a gather + this kernel — no decode, no resize, no PCIe traffic.
__global__ __launch_bounds__(256) void prep_normalize_kernel(const uint8_t* __restrict__ u8, const int64_t* __restrict__ rows, int R,
                                                             float m0, float m1, float m2, float d0, float d1, float d2,
                                                             float* __restrict__ out) {
  __shared__ uint32_t line[(1024 * 3 + 3) / 4];
  const int b = blockIdx.y, i = blockIdx.x;
  const int row_bytes = R * 3, words = (row_bytes + 3) >> 2;
  const size_t img = rows ? static_cast<size_t>(rows[b]) : static_cast<size_t>(b);
  const uint8_t* src = u8 + (img * R + i) * row_bytes;
  if (((reinterpret_cast<uintptr_t>(src) | static_cast<uintptr_t>(row_bytes)) & 3) == 0) {
    for (int q = threadIdx.x; q < words; q += blockDim.x) line[q] = reinterpret_cast<const uint32_t*>(src)[q];
  } else {
    uint8_t* lb8 = reinterpret_cast<uint8_t*>(line);
    for (int q = threadIdx.x; q < row_bytes; q += blockDim.x) lb8[q] = src[q];
  }
  __syncthreads();
  const uint8_t* lb = reinterpret_cast<const uint8_t*>(line);
  const size_t plane = static_cast<size_t>(R) * R;
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    float* o = out + static_cast<size_t>(b) * 3 * plane + static_cast<size_t>(i) * R + j;
    o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j]), 255.0f), m0), d0);
    o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j + 1]), 255.0f), m1), d1);
    o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn(static_cast<float>(lb[3 * j + 2]), 255.0f), m2), d2);
  }
}

static int prep_ksize(int max_h, int max_w, int R) {
  const int in = max_h > max_w ? max_h : max_w;
  // the largest shrink of any axis is bounded by in / R (eval) or in / min-edge-scale (train: both axes shrink by short / R <= in / R)
  const double fs = static_cast<double>(in) / R < 1.0 ? 1.0 : static_cast<double>(in) / R;
  return static_cast<int>(__builtin_ceil(2.0 * fs)) * 2 + 1;
}

}  // namespace cmh

using namespace cmh;

extern "C" size_t cmh_image_preprocess_workspace_bytes(int32_t batch, int32_t max_h, int32_t max_w, int32_t R) {
  if (batch <= 0 || max_h <= 0 || max_w <= 0 || R <= 0) return 0;
  const size_t KS = prep_ksize(max_h, max_w, R);
  return align_up(static_cast<size_t>(batch) * 2 * KS * R * 4, 256) + align_up(static_cast<size_t>(batch) * 2 * R * 2 * 4, 256) +
         align_up(static_cast<size_t>(batch) * sizeof(PrepImage), 256) + align_up(static_cast<size_t>(batch) * max_h * R * 3, 256) + 256;
}

extern "C" int cmh_image_preprocess(const uint8_t* pixels, const int64_t* offsets, const int32_t* hw, int32_t batch, int32_t max_h,
                                    int32_t max_w, int32_t R, int32_t train, const float* mean, const float* stdv, float* out,
                                    uint8_t* out_u8, void* workspace, size_t workspace_bytes, void* stream) {
  CMH_CHECK_ARG(pixels && offsets && hw && mean && stdv && workspace && (out || out_u8), "image_preprocess: null pointer");
  CMH_CHECK_ARG(batch > 0 && batch <= 65535 && R > 0 && R <= 1024, "image_preprocess: bad batch %d / resolution %d", batch, R);
  CMH_CHECK_ARG(max_h > 0 && max_w > 0 && max_h <= 16384 && max_w <= 4096, "image_preprocess: bad max size %dx%d (width <= 4096)", max_h, max_w);
  const size_t need = cmh_image_preprocess_workspace_bytes(batch, max_h, max_w, R);
  if (workspace_bytes < need) return fail(CMH_ERR_WORKSPACE, "image_preprocess: workspace %zu < %zu bytes", workspace_bytes, need);
  const int KS = prep_ksize(max_h, max_w, R);
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  int32_t* coef = reinterpret_cast<int32_t*>(ws); ws += align_up(static_cast<size_t>(batch) * 2 * KS * R * 4, 256);
  int32_t* bounds = reinterpret_cast<int32_t*>(ws); ws += align_up(static_cast<size_t>(batch) * 2 * R * 2 * 4, 256);
  PrepImage* info = reinterpret_cast<PrepImage*>(ws); ws += align_up(static_cast<size_t>(batch) * sizeof(PrepImage), 256);
  uint8_t* tmp = reinterpret_cast<uint8_t*>(ws);
  const size_t tmp_stride = static_cast<size_t>(max_h) * R * 3;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(prep_coeffs_kernel, dim3(batch, 2), dim3(256), 0, st, hw, R, train ? 1 : 0, KS, coef, bounds, info);
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(prep_horizontal_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) == hipSuccess;
  if (!lds_ok) return fail(CMH_ERR_LAUNCH, "image_preprocess: cannot raise the dynamic LDS limit");
  const int pitch = max_w + 1;   // odd-ish pitch: the 8 rows of a tap land on different banks
  hipLaunchKernelGGL(prep_horizontal_kernel, dim3((max_h + kPrepRows - 1) / kPrepRows, batch), dim3(256),
                     static_cast<size_t>(kPrepRows) * pitch * 4, st, pixels, offsets, hw, R, KS, coef, bounds, info, tmp, tmp_stride, pitch);
  hipLaunchKernelGGL(prep_vertical_kernel, dim3(R, batch), dim3(256), 0, st, tmp, tmp_stride, R, KS, coef, bounds, mean[0], mean[1], mean[2],
                     stdv[0], stdv[1], stdv[2], out, out_u8);
  CMH_CHECK_LAUNCH("image_preprocess");
  return CMH_OK;
}

extern "C" int cmh_image_normalize(const uint8_t* u8, const int64_t* rows, int32_t batch, int32_t R, const float* mean,
                                   const float* stdv, float* out, void* stream) {
  CMH_CHECK_ARG(u8 && mean && stdv && out, "image_normalize: null pointer");
  CMH_CHECK_ARG(batch > 0 && batch <= 65535 && R > 0 && R <= 1024, "image_normalize: bad batch %d / resolution %d", batch, R);
  hipLaunchKernelGGL(prep_normalize_kernel, dim3(R, batch), dim3(256), 0, as_stream(stream), u8, rows, R, mean[0], mean[1], mean[2],
                     stdv[0], stdv[1], stdv[2], out);
  CMH_CHECK_LAUNCH("image_normalize");
  return CMH_OK;
}
