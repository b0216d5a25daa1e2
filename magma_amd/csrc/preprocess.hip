// preprocess.hip -- CLIP image preprocessing on the device (the step in front of the hot path).
//
// Reference: magma/transforms.py:121-134 = torchvision Resize(n, BICUBIC) on a PIL image -> CenterCrop ->
// RGB -> ToTensor -> Normalize.  Resize on a PIL image IS Pillow's ImagingResample (third-party, Pillow
// src/libImaging/Resample.c): separable, antialiased bicubic (a = -0.5, support 2 * max(scale, 1)), 8-bit
// fixed point: coefficients rounded to 22 fractional bits, int32 accumulation from 1 << 21, arithmetic shift,
// clamp to [0, 255]; horizontal pass first, uint8 intermediate.  The coefficient tables are built on the host
// (magma_amd/transforms.py, doubles, same expression order as Pillow); the two integer passes and the final
// fp32 normalisation run here.  Integer work: bit-exact against PIL (tests/test_preprocess_gpu.py).
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// one thread per output pixel (3 interleaved channels).  AXIS 1: out[y][xx] from src[y][xmin .. xmin+n);
// AXIS 0: out[yy][x] from src[ymin .. ymin+n)[x].
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                          uint8_t* __restrict__ dst, int out_size,
                                                          const int32_t* __restrict__ kk, const int32_t* __restrict__ bounds,
                                                          int ksize) {
  const int Ho = AXIS == 1 ? H : out_size, Wo = AXIS == 1 ? out_size : W;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)Ho * Wo) return;
  const int y = (int)(idx / Wo), x = (int)(idx - (int64_t)y * Wo);
  const int o = AXIS == 1 ? x : y;
  const int lo = bounds[2 * o], n = bounds[2 * o + 1];
  const int32_t* k = kk + (int64_t)o * ksize;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  if (AXIS == 1) {
    const uint8_t* p = src + ((int64_t)y * W + lo) * 3;
    for (int i = 0; i < n; ++i) { const int c = k[i]; s0 += p[3 * i] * c; s1 += p[3 * i + 1] * c; s2 += p[3 * i + 2] * c; }
  } else {
    const uint8_t* p = src + ((int64_t)lo * W + x) * 3;
    for (int i = 0; i < n; ++i) { const int c = k[i]; const uint8_t* q = p + (int64_t)i * W * 3; s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c; }
  }
  uint8_t* d = dst + idx * 3;
  d[0] = (uint8_t)min(255, max(0, s0 >> PRECISION_BITS));
  d[1] = (uint8_t)min(255, max(0, s1 >> PRECISION_BITS));
  d[2] = (uint8_t)min(255, max(0, s2 >> PRECISION_BITS));
}

// crop [top, top+n) x [left, left+n) of an HWC uint8 image -> CHW fp32, (v / 255 - mean[c]) / std[c]
// (IEEE fp32 division and subtraction in torch's order: bit-identical to ToTensor + Normalize on the host)
__global__ __launch_bounds__(256) void crop_normalize_kernel(const uint8_t* __restrict__ src, int W, int top, int left, int n,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * n) return;
  const int y = idx / n, x = idx - y * n;
  const uint8_t* p = src + ((int64_t)(top + y) * W + left + x) * 3;
  out[idx] = ((float)p[0] / 255.0f - m0) / s0;
  out[n * n + idx] = ((float)p[1] / 255.0f - m1) / s1;
  out[2 * n * n + idx] = ((float)p[2] / 255.0f - m2) / s2;
}

}  // namespace

extern "C" int mg_resample_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t out_size, int32_t axis,
                              const int32_t* coeffs, const int32_t* bounds, int32_t ksize, void* stream) {
  if (!src || !dst || !coeffs || !bounds) MG_FAIL(MG_ERR_SHAPE, "mg_resample_u8: null pointer");
  if (H <= 0 || W <= 0 || out_size <= 0 || ksize <= 0 || (axis != 0 && axis != 1)) MG_FAIL(MG_ERR_SHAPE, "mg_resample_u8: bad geometry");
  const int64_t n = axis == 1 ? (int64_t)H * out_size : (int64_t)out_size * W;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (axis == 1) hipLaunchKernelGGL(resample_u8_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, H, W, dst, out_size, coeffs, bounds, ksize);
  else hipLaunchKernelGGL(resample_u8_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, src, H, W, dst, out_size, coeffs, bounds, ksize);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_crop_normalize_f32(const uint8_t* src, int32_t H, int32_t W, int32_t top, int32_t left, int32_t n,
                                     const float* mean3, const float* std3, float* out, void* stream) {
  if (!src || !out || !mean3 || !std3) MG_FAIL(MG_ERR_SHAPE, "mg_crop_normalize_f32: null pointer");
  if (n <= 0 || top < 0 || left < 0 || top + n > H || left + n > W) MG_FAIL(MG_ERR_SHAPE, "mg_crop_normalize_f32: crop window outside the image");
  hipLaunchKernelGGL(crop_normalize_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, W, top, left, n,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
