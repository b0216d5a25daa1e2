// nfnet.hip -- the kernels the NF-ResNet-50 image encoder needs on top of the conv GEMMs (encoder_name "nfresnet50":
// reference magma/image_encoders.py:31-45 = timm nf_resnet50 minus its classifier + AdaptiveAvgPool2d((1,1))).
//
//   weight_standardize   ScaledStdConv2d's weight transform  W_hat = (W - mean_o) * rsqrt(var_o + eps) * gain_o * scale
//                        (biased variance over the fan-in of output channel o), written as the bf16 GEMM operand in the
//                        column order the conv's A loader produces: (cin,ky,kx) kept for 1x1 / explicit-im2col convs,
//                        (ky,kx,cin) for the implicit-im2col 3x3 over NHWC.  One workgroup per output channel.
//   im2col_nchw          small-Cin strided conv (the 7x7 / stride-2 stem, Cin = 3) as an explicit im2col of the NCHW image:
//                        rows = output pixels, col = (c*k + ky)*k + kx, zero padded to ldo columns.
//   maxpool3x3s2         MaxPool2d(3, stride 2, padding 1), NHWC, 16-byte channel chunks.
//   subsample2           rows/columns 0, 2, 4, ... of an NHWC map: a stride-2 3x3 conv (padding 1) is its stride-1 output
//                        sampled at the even positions.
//   relu_mean_rows       ReLU then mean over the H*W positions of every image: final_act + AdaptiveAvgPool2d((1,1)).
// All HBM-bound element-wise / reduction work; the matrix products stay in gemm.hip.
#include "common.h"

namespace {

unsigned host_grid(int64_t total, int per_block) {
  const int64_t g = (total + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > 1048560 ? 1048560 : g));
}

__global__ __launch_bounds__(256) void weight_standardize_kernel(const mg_bf16* __restrict__ w, const mg_bf16* __restrict__ gain,
                                                                 mg_bf16* __restrict__ out, int fan_in, int64_t ldo, int cin,
                                                                 int kk, int to_khwc, float scale, float eps) {
  __shared__ float red[2][4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const mg_bf16* row = w + (int64_t)o * fan_in;
  float s = 0.f, ss = 0.f;
  for (int i = tid; i < fan_in; i += 256) { const float v = bf2f(row[i]); s += v; ss += v * v; }
  s = wave_sum(s); ss = wave_sum(ss);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const float mean = s / (float)fan_in;
  // two-pass variance (the mean is known now): sum (w - mean)^2, immune to cancellation for rows with a large mean
  float d2 = 0.f;
  for (int i = tid; i < fan_in; i += 256) { const float v = bf2f(row[i]) - mean; d2 += v * v; }
  d2 = wave_sum(d2);
  __syncthreads();
  if ((tid & 63) == 0) red[0][tid >> 6] = d2;
  __syncthreads();
  const float var = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)fan_in;
  const float g = bf2f(gain[o]) * scale * rsqrtf(var + eps);
  mg_bf16* orow = out + (int64_t)o * ldo;
  for (int i = tid; i < ldo; i += 256) {
    float v = 0.f;
    if (i < fan_in) {
      int src = i;
      if (to_khwc) { const int c = i % cin, t = i / cin; src = c * kk + t; }     // out column (ky,kx,c) <- weight (c,ky,kx)
      v = (bf2f(row[src]) - mean) * g;
    }
    orow[i] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void im2col_nchw_kernel(const mg_bf16* __restrict__ img, mg_bf16* __restrict__ out, int B,
                                                          int C, int H, int W, int k, int stride, int pad, int Ho, int Wo,
                                                          int ldo) {
  const int chunks = ldo >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * chunks;
  const int K = C * k * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    int64_t m = i / chunks;
    const int xo = (int)(m % Wo);
    const int64_t t = m / Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    uint16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = ch * 8 + j;
      uint16_t x = 0;
      if (col < K) {
        const int kx = col % k, r = col / k, ky = r % k, c = r / k;
        const int yy = yo * stride + ky - pad, xx = xo * stride + kx - pad;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) x = img[(((int64_t)b * C + c) * H + yy) * W + xx];
      }
      v[j] = x;
    }
    u32x4 wv;
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = (uint32_t)v[2 * j] | ((uint32_t)v[2 * j + 1] << 16);
    *(u32x4*)(out + m * ldo + ch * 8) = wv;
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const mg_bf16* __restrict__ x, mg_bf16* __restrict__ y, int B, int H,
                                                           int W, int C, int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = 2 * yo + ky - 1;
      if (yy < 0 || yy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = 2 * xo + kx - 1;
        if (xx < 0 || xx >= W) continue;
        const u32x4 a = *(const u32x4*)(x + (((int64_t)b * H + yy) * W + xx) * C + c * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[2 * j] = fmaxf(m[2 * j], bflo(a[j])); m[2 * j + 1] = fmaxf(m[2 * j + 1], bfhi(a[j])); }
      }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2bf(m[2 * j], m[2 * j + 1]);
    *(u32x4*)(y + i * 8) = o;
  }
}

__global__ __launch_bounds__(256) void subsample2_kernel(const mg_bf16* __restrict__ x, mg_bf16* __restrict__ y, int B, int H,
                                                         int W, int C, int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho), b = (int)(t / Ho);
    *(u32x4*)(y + i * 8) = *(const u32x4*)(x + (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + c * 8);
  }
}

// one workgroup per (image, 64-channel slab): thread = (position slice ps of 32, channel pair of 32); fp32 partial sums
// reduced through LDS
__global__ __launch_bounds__(1024) void relu_mean_rows_kernel(const mg_bf16* __restrict__ x, mg_bf16* __restrict__ y, int HW, int C) {
  __shared__ float part[32][64];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int cp = threadIdx.x & 31, ps = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f;
  if (c0 + cp * 2 < C) {
    for (int p = ps; p < HW; p += 32) {
      const uint32_t w = *(const uint32_t*)(x + ((int64_t)b * HW + p) * C + c0 + cp * 2);
      s0 += fmaxf(bflo(w), 0.f); s1 += fmaxf(bfhi(w), 0.f);
    }
  }
  part[ps][cp * 2] = s0; part[ps][cp * 2 + 1] = s1;
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += part[i][threadIdx.x];
    y[(int64_t)b * C + c0 + threadIdx.x] = f2bf(s / (float)HW);
  }
}

// ---- backward pieces (training the NF-ResNet encoder) --------------------------------------------------------------------
// ScaledStdConv2d weight transform, backward: with n = (w - mean) * r, r = rsqrt(var + eps), W_hat = n * gain * scale and
// dn = dW_hat * gain * scale:   dgain += scale * sum(dW_hat * n),   dw += r * (dn - mean(dn) - n * mean(dn * n))
// (the LayerNorm backward over the fan-in of one output channel).  One workgroup per output channel; fp32 accumulation
// into the caller's gradient buffers (grad accumulation across micro-steps).
__global__ __launch_bounds__(256) void weight_standardize_bwd_kernel(const mg_bf16* __restrict__ w, const mg_bf16* __restrict__ gain,
                                                                     const float* __restrict__ dwhat, int64_t ldd,
                                                                     float* __restrict__ dw, float* __restrict__ dgain,
                                                                     int fan_in, float scale, float eps, float dmult) {
  __shared__ float red[3][4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const mg_bf16* row = w + (int64_t)o * fan_in;
  const float* drow = dwhat + (int64_t)o * ldd;
  float s = 0.f;
  for (int i = tid; i < fan_in; i += 256) s += bf2f(row[i]);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[0][tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)fan_in;
  __syncthreads();
  float d2 = 0.f;
  for (int i = tid; i < fan_in; i += 256) { const float v = bf2f(row[i]) - mean; d2 += v * v; }
  d2 = wave_sum(d2);
  if ((tid & 63) == 0) red[0][tid >> 6] = d2;
  __syncthreads();
  const float r = rsqrtf((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)fan_in + eps);
  __syncthreads();
  const float gs = bf2f(gain[o]) * scale;
  float sdn = 0.f, sdnn = 0.f, sdwn = 0.f;
  for (int i = tid; i < fan_in; i += 256) {
    const float n = (bf2f(row[i]) - mean) * r, dh = drow[i] * dmult, dn = dh * gs;
    sdn += dn; sdnn += dn * n; sdwn += dh * n;
  }
  sdn = wave_sum(sdn); sdnn = wave_sum(sdnn); sdwn = wave_sum(sdwn);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sdn; red[1][tid >> 6] = sdnn; red[2][tid >> 6] = sdwn; }
  __syncthreads();
  const float m1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)fan_in;
  const float m2 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)fan_in;
  if (tid == 0) dgain[o] += scale * (red[2][0] + red[2][1] + red[2][2] + red[2][3]);
  float* orow = dw + (int64_t)o * fan_in;
  for (int i = tid; i < fan_in; i += 256) {
    const float n = (bf2f(row[i]) - mean) * r;
    orow[i] += r * (drow[i] * dmult * gs - m1 - n * m2);
  }
}

// MaxPool2d(3, stride 2, padding 1) backward as a gather (deterministic, no atomics): an input pixel receives dy of every
// window whose FIRST maximum (row-major scan of the window, as PyTorch's max_pool2d picks it) it is.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const mg_bf16* __restrict__ x, const mg_bf16* __restrict__ dy,
                                                               mg_bf16* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H), b = (int)(t / H);
    float acc = 0.f;
    for (int yo = max(0, yi / 2); yo <= min(Ho - 1, (yi + 1) / 2); ++yo)
      for (int xo = max(0, xi / 2); xo <= min(Wo - 1, (xi + 1) / 2); ++xo) {
        float best = -INFINITY; int by = -1, bx = -1;
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = 2 * yo + ky - 1;
          if (yy < 0 || yy >= H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = 2 * xo + kx - 1;
            if (xx < 0 || xx >= W) continue;
            const float v = bf2f(x[(((int64_t)b * H + yy) * W + xx) * C + c]);
            if (v > best) { best = v; by = yy; bx = xx; }
          }
        }
        if (by == yi && bx == xi) acc += bf2f(dy[(((int64_t)b * Ho + yo) * Wo + xo) * C + c]);
      }
    dx[i] = f2bf(acc);
  }
}

// dx[b, 2i, 2j, :] = dy[b, i, j, :], zero elsewhere (backward of subsample2)
__global__ __launch_bounds__(256) void subsample2_bwd_kernel(const mg_bf16* __restrict__ dy, mg_bf16* __restrict__ dx, int B, int H,
                                                             int W, int C, int Ho, int Wo) {
  const int cv = C >> 3;
  const int64_t total = (int64_t)B * H * W * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H), b = (int)(t / H);
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (!(yi & 1) && !(xi & 1)) v = *(const u32x4*)(dy + (((int64_t)b * Ho + (yi >> 1)) * Wo + (xi >> 1)) * C + c * 8);
    *(u32x4*)(dx + i * 8) = v;
  }
}

// dx[b, p, c] = x[b, p, c] > 0 ? g[b, c] / HW : 0  (backward of relu_mean_rows)
__global__ __launch_bounds__(256) void relu_mean_rows_bwd_kernel(const mg_bf16* __restrict__ x, const mg_bf16* __restrict__ g,
                                                                 mg_bf16* __restrict__ dx, int B, int HW, int C) {
  const int64_t total = (int64_t)B * HW * C;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((int64_t)HW * C));
    dx[i] = f2bf(bf2f(x[i]) > 0.f ? bf2f(g[(int64_t)b * C + c]) * inv : 0.f);
  }
}

}  // namespace

extern "C" int mg_weight_standardize_bf16(const mg_bf16* w, const mg_bf16* gain, mg_bf16* out, int32_t cout, int32_t cin,
                                          int32_t kh, int32_t kw, int64_t ldo, int32_t to_khwc, float scale, float eps,
                                          void* stream) {
  if (!w || !gain || !out || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_weight_standardize_bf16: bad arguments");
  const int64_t fan_in = (int64_t)cin * kh * kw;
  if (ldo < fan_in || fan_in > (1 << 24)) MG_FAIL(MG_ERR_SHAPE, "mg_weight_standardize_bf16: ldo (%lld) must be >= cin*kh*kw (%lld)", (long long)ldo, (long long)fan_in);
  if (!(eps >= 0.f)) MG_FAIL(MG_ERR_SHAPE, "mg_weight_standardize_bf16: eps must be >= 0");
  hipLaunchKernelGGL(weight_standardize_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream, w, gain, out, (int)fan_in, ldo,
                     cin, kh * kw, to_khwc ? 1 : 0, scale, eps);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_im2col_nchw_bf16(const mg_bf16* img, mg_bf16* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k,
                                   int32_t stride, int32_t pad, int32_t ldo, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0) MG_FAIL(MG_ERR_SHAPE, "mg_im2col_nchw_bf16: bad arguments");
  if ((ldo & 7) || ldo < C * k * k) MG_FAIL(MG_ERR_SHAPE, "mg_im2col_nchw_bf16: ldo must be a multiple of 8 and >= C*k*k");
  if (!MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_im2col_nchw_bf16: out must be 16-byte aligned");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_im2col_nchw_bf16: empty output");
  const int64_t total = (int64_t)B * Ho * Wo * (ldo >> 3);
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(host_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, C, H, W, k,
                     stride, pad, Ho, Wo, ldo);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_maxpool3x3s2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_maxpool3x3s2_nhwc_bf16: need C %% 8 == 0");
  if (!MG_ALIGNED16(x) || !MG_ALIGNED16(y)) MG_FAIL(MG_ERR_ALIGN, "mg_maxpool3x3s2_nhwc_bf16: pointers must be 16-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(host_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_subsample2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_subsample2_nhwc_bf16: need C %% 8 == 0");
  if (!MG_ALIGNED16(x) || !MG_ALIGNED16(y)) MG_FAIL(MG_ERR_ALIGN, "mg_subsample2_nhwc_bf16: pointers must be 16-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(subsample2_kernel, dim3(host_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_relu_mean_rows_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t HW, int32_t C, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || (C & 1)) MG_FAIL(MG_ERR_SHAPE, "mg_relu_mean_rows_bf16: need C even");
  hipLaunchKernelGGL(relu_mean_rows_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, (hipStream_t)stream, x, y, HW, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_weight_standardize_bwd_f32(const mg_bf16* w, const mg_bf16* gain, const float* dwhat, int64_t ldd, float* dw,
                                             float* dgain, int32_t cout, int32_t fan_in, float scale, float eps, float dmult, void* stream) {
  if (!w || !gain || !dwhat || !dw || !dgain || cout <= 0 || fan_in <= 0 || ldd < fan_in) MG_FAIL(MG_ERR_SHAPE, "mg_weight_standardize_bwd_f32: bad arguments");
  hipLaunchKernelGGL(weight_standardize_bwd_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream, w, gain, dwhat, ldd, dw, dgain, fan_in, scale, eps, dmult);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_maxpool3x3s2_bwd_nhwc_bf16(const mg_bf16* x, const mg_bf16* dy, mg_bf16* dx, int32_t B, int32_t H, int32_t W,
                                             int32_t C, void* stream) {
  if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_maxpool3x3s2_bwd_nhwc_bf16: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(host_grid((int64_t)B * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B, H, W, C, Ho, Wo);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_subsample2_bwd_nhwc_bf16(const mg_bf16* dy, mg_bf16* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_subsample2_bwd_nhwc_bf16: need C %% 8 == 0");
  if (!MG_ALIGNED16(dy) || !MG_ALIGNED16(dx)) MG_FAIL(MG_ERR_ALIGN, "mg_subsample2_bwd_nhwc_bf16: pointers must be 16-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(host_grid((int64_t)B * H * W * (C >> 3), 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C, Ho, Wo);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_relu_mean_rows_bwd_bf16(const mg_bf16* x, const mg_bf16* g, mg_bf16* dx, int32_t B, int32_t HW, int32_t C, void* stream) {
  if (!x || !g || !dx || B <= 0 || HW <= 0 || C <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_relu_mean_rows_bwd_bf16: bad arguments");
  hipLaunchKernelGGL(relu_mean_rows_bwd_kernel, dim3(host_grid((int64_t)B * HW * C, 256)), dim3(256), 0, (hipStream_t)stream, x, g, dx, B, HW, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
