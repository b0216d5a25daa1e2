// elementwise.hip -- the HBM-bound kernels of the MAGMA hot path (gfx950):
// LayerNorm, embedding gather, NHWC avg-pool, stem im2col, greedy argmax,
// build_labels (integer, exact) and the shifted cross-entropy pieces.
// All bf16 traffic is 16 B per lane (guide G13).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// LayerNorm: one 256-thread workgroup per row, two-pass statistics in fp32 on
// register-resident data (d <= 256*8*MAXV).
// ---------------------------------------------------------------------------
constexpr int LN_MAXV = 8;  // 16-B vectors per thread -> d <= 16384

__global__ __launch_bounds__(256) void layernorm_kernel(const mg_bf16* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        mg_bf16* __restrict__ y, int64_t ldy, int d, float eps) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nvec = d >> 3;
  const mg_bf16* xr = x + (int64_t)row * ldx;
  float v[LN_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
      const u32x4 w = *(const u32x4*)(xr + vi * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][2 * j] = bflo(w[j]); v[i][2 * j + 1] = bfhi(w[j]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float t = v[i][j] - mean; q += t * t; }
    }
  }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d + eps);
  mg_bf16* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
      const float4 g0 = *(const float4*)(gamma + vi * 8), g1 = *(const float4*)(gamma + vi * 8 + 4);
      const float4 b0 = *(const float4*)(beta + vi * 8), b1 = *(const float4*)(beta + vi * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      u32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        w[j] = pack2bf((v[i][2 * j] - mean) * rstd * g[2 * j] + bb[2 * j],
                       (v[i][2 * j + 1] - mean) * rstd * g[2 * j + 1] + bb[2 * j + 1]);
      *(u32x4*)(yr + vi * 8) = w;
    }
  }
}

// The same arithmetic, LN_ROWS consecutive rows per workgroup, for the big activations of the forward / training step (M = 32768,
// d = 4096): gamma / beta (fp32: 8 x the bytes of a bf16 row slice) are loaded ONCE per workgroup instead of once per row -- per row the
// one-row kernel pulls 8 KB of x from HBM and 32 KB of gamma / beta from L2 -- and row r+1 is in flight while row r is reduced.
// NV = 16-byte vectors per thread (d <= 2048 * NV).  Bit-identical to layernorm_kernel (same order of every sum).
constexpr int LN_ROWS = 4;
template <int NV>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const mg_bf16* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, mg_bf16* __restrict__ y, int64_t ldy,
                                                             int d, float eps, int rows) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = d >> 3;
  float g[NV][8], bb[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
      const float4 g0 = *(const float4*)(gamma + vi * 8), g1 = *(const float4*)(gamma + vi * 8 + 4);
      const float4 b0 = *(const float4*)(beta + vi * 8), b1 = *(const float4*)(beta + vi * 8 + 4);
      g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
      bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
    }
  }
  const int row0 = blockIdx.x * LN_ROWS, row1 = min(rows, row0 + LN_ROWS);
  u32x4 cur[NV], nxt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = tid + i * 256;
    cur[i] = (u32x4){0u, 0u, 0u, 0u};
    if (vi < nvec) cur[i] = *(const u32x4*)(x + (int64_t)row0 * ldx + vi * 8);
  }
  for (int row = row0; row < row1; ++row) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * 256;
      nxt[i] = (u32x4){0u, 0u, 0u, 0u};
      if (row + 1 < row1 && vi < nvec) nxt[i] = *(const u32x4*)(x + (int64_t)(row + 1) * ldx + vi * 8);
    }
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (tid + i * 256 < nvec) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][2 * j] = bflo(cur[i][j]); v[i][2 * j + 1] = bfhi(cur[i][j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      }
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (tid + i * 256 < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[i][j] - mean; q += t * t; }
      }
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();      // also: everybody has read red[0..3] of this row before the next row overwrites it
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d + eps);
    mg_bf16* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = tid + i * 256;
      if (vi < nvec) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[j] = pack2bf((v[i][2 * j] - mean) * rstd * g[i][2 * j] + bb[i][2 * j],
                         (v[i][2 * j + 1] - mean) * rstd * g[i][2 * j + 1] + bb[i][2 * j + 1]);
        *(u32x4*)(yr + vi * 8) = w;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
  }
}

// ---------------------------------------------------------------------------
// embedding gather: one workgroup per token.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, int T,
                                                        const mg_bf16* __restrict__ wte, int vocab, int d,
                                                        mg_bf16* __restrict__ out, int64_t out_bstride,
                                                        int row_off) {
  const int tok = blockIdx.x;
  const int b = tok / T, t = tok - b * T;
  int64_t id = ids[tok];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4* src = (const u32x4*)(wte + id * (int64_t)d);
  u32x4* dst = (u32x4*)(out + b * out_bstride + (int64_t)(row_off + t) * d);
  for (int i = threadIdx.x; i < (d >> 3); i += 256) dst[i] = src[i];
}

// ---------------------------------------------------------------------------
// 2x2 average pool, NHWC.  thread = one 16-B channel chunk of one output pixel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_kernel(const mg_bf16* __restrict__ x, mg_bf16* __restrict__ y,
                                                       int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cv = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const mg_bf16* p00 = x + (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + c * 8;
    const u32x4 a = *(const u32x4*)p00, bq = *(const u32x4*)(p00 + C);
    const u32x4 cq = *(const u32x4*)(p00 + (int64_t)W * C), dq = *(const u32x4*)(p00 + (int64_t)W * C + C);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(0.25f * (bflo(a[j]) + bflo(bq[j]) + bflo(cq[j]) + bflo(dq[j])),
                     0.25f * (bfhi(a[j]) + bfhi(bq[j]) + bfhi(cq[j]) + bfhi(dq[j])));
    *(u32x4*)(y + i * 8) = o;
  }
}

// ---------------------------------------------------------------------------
// stem conv1 im2col: NCHW bf16 image -> [B*(H/2)*(W/2), 32], col = c*9 + ky*3 + kx
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_im2col_kernel(const mg_bf16* __restrict__ img,
                                                          mg_bf16* __restrict__ out, int B, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t total = (int64_t)B * Ho * Wo;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < total; m += (int64_t)gridDim.x * 256) {
    const int xo = (int)(m % Wo);
    const int64_t t = m / Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    uint16_t col[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) col[i] = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
#pragma unroll
          for (int c = 0; c < 3; ++c) col[c * 9 + ky * 3 + kx] = img[(((int64_t)b * 3 + c) * H + yy) * W + xx];
        }
      }
    u32x4* dst = (u32x4*)(out + m * 32);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (uint32_t)col[g * 8 + 2 * j] | ((uint32_t)col[g * 8 + 2 * j + 1] << 16);
      dst[g] = w;
    }
  }
}

// ---------------------------------------------------------------------------
// greedy argmax over fp32 logits, first maximum wins (torch.argmax on CPU).
// one 1024-thread workgroup per row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                      int64_t* __restrict__ token) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)blockIdx.x * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  int i0 = 0;
  if ((((uintptr_t)row) & 15u) == 0) {          // 16-byte loads, four in flight per thread (the scalar loop was latency-bound:
    const int nq = V >> 2;                      // 20 us for a 200 KB row)
    for (int qb = tid; qb < nq; qb += 4096) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = qb + u * 1024;
        v[u] = q < nq ? ((const f32x4*)row)[q] : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (qb + u * 1024 >= nq) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = (qb + u * 1024) * 4 + j;
          if (v[u][j] > best || (v[u][j] == best && i < idx)) { best = v[u][j]; idx = i; }
        }
      }
    }
    i0 = nq << 2;
  }
  for (int i = i0 + tid; i < V; i += 1024) {
    const float v = row[i];
    if (v > best || (v == best && i < idx)) { best = v; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    token[blockIdx.x] = (idx == 0x7fffffff) ? 0 : idx;
  }
}

__global__ void advance_pos_kernel(int* d_pos, int delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *d_pos += delta;
}

// ---------------------------------------------------------------------------
// build_labels (reference magma/utils.py:334-364), integer, exact.
// one workgroup per row: find the first eos among labels[P..S) with a block
// min-reduction instead of the reference's per-token host loop.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_labels_kernel(const int64_t* __restrict__ cap,
                                                           int64_t* __restrict__ lab, int S, int P,
                                                           int64_t eos) {
  __shared__ int wmin[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* c = cap + (int64_t)blockIdx.x * S;
  int64_t* l = lab + (int64_t)blockIdx.x * S;
  const int T = S - P;  // caption tokens that survive the truncation
  int first = 0x7fffffff;
  for (int t = tid; t < T; t += 256)
    if (c[t] == eos) first = min(first, t);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
  if (lane == 0) wmin[wave] = first;
  __syncthreads();
  first = min(min(wmin[0], wmin[1]), min(wmin[2], wmin[3]));
  for (int k = tid; k < S; k += 256) {
    int64_t v = -100;
    if (k >= P) {
      const int t = k - P;
      if (t <= first) v = c[t];   // eos itself is kept, everything after is masked
    }
    l[k] = v;
  }
}

// ---------------------------------------------------------------------------
// cross entropy rows: loss_row[r] = logsumexp(logits[r,:]) - logits[r,tgt]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int64_t ld,
                                                      const int64_t* __restrict__ tgt,
                                                      float* __restrict__ loss_row, int V) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int64_t tg = tgt[r];
  if (tg < 0 || tg >= V) { if (tid == 0) loss_row[r] = 0.f; return; }
  const float* row = logits + (int64_t)r * ld;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int i = tid; i < V; i += 256) s += expf(row[i] - mx);
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  if (tid == 0) loss_row[r] = logf(red[4] + red[5] + red[6] + red[7]) + mx - row[tg];
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_row,
                                                        const int64_t* __restrict__ tgt, int R,
                                                        float* __restrict__ out) {
  __shared__ float rs[4], rc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f, c = 0.f;
  for (int i = tid; i < R; i += 256)
    if (tgt[i] >= 0) { s += loss_row[i]; c += 1.f; }
  s = wave_sum(s); c = wave_sum(c);
  if (lane == 0) { rs[wave] = s; rc[wave] = c; }
  __syncthreads();
  if (tid == 0) {
    const float ts = rs[0] + rs[1] + rs[2] + rs[3], tc = rc[0] + rc[1] + rc[2] + rc[3];
    out[0] = ts / tc;   // NaN when no valid target, like F.cross_entropy
    out[1] = tc;
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int mg_layernorm_bf16(const mg_bf16* x, int64_t ldx, const float* gamma, const float* beta, mg_bf16* y,
                                 int64_t ldy, int32_t rows, int32_t d, float eps, void* stream) {
  if (rows <= 0 || d <= 0 || (d & 7) || d > 256 * 8 * LN_MAXV) MG_FAIL(MG_ERR_SHAPE, "mg_layernorm_bf16: need rows>0, d%%8==0, d<=%d (d=%d)", 256 * 8 * LN_MAXV, d);
  if (!x || !y || !gamma || !beta) MG_FAIL(MG_ERR_SHAPE, "mg_layernorm_bf16: null pointer");
  if (!MG_ALIGNED16(x) || !MG_ALIGNED16(y) || !MG_ALIGNED16(gamma) || !MG_ALIGNED16(beta) || (ldx & 7) || (ldy & 7))
    MG_FAIL(MG_ERR_ALIGN, "mg_layernorm_bf16: 16-byte alignment required");
  // big activations: LN_ROWS rows per workgroup (gamma / beta once per workgroup); small ones keep one row per workgroup to fill the chip
  static const bool multi = [] { const char* e = getenv("MAGMA_LN_ROWS"); return !e || atoi(e) != 1; }();   // A/B knob
  if (multi && rows >= 8192 && d <= 4096) {
    const dim3 grid((unsigned)((rows + LN_ROWS - 1) / LN_ROWS));
    if (d <= 2048) hipLaunchKernelGGL(layernorm_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, d, eps, rows);
    else hipLaunchKernelGGL(layernorm_rows_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, d, eps, rows);
  } else {
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, y, ldy, d, eps);
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_embedding_bf16(const int64_t* ids, int32_t B, int32_t T, const mg_bf16* wte, int32_t vocab,
                                 int32_t d, mg_bf16* out, int64_t out_bstride, int32_t row_off, void* stream) {
  if (B <= 0 || T <= 0 || vocab <= 0 || d <= 0 || (d & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_embedding_bf16: bad shape");
  if (!ids || !wte || !out) MG_FAIL(MG_ERR_SHAPE, "mg_embedding_bf16: null pointer");
  if (!MG_ALIGNED16(wte) || !MG_ALIGNED16(out) || (out_bstride & 7)) MG_FAIL(MG_ERR_ALIGN, "mg_embedding_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(embedding_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, ids, T, wte, vocab, d, out, out_bstride, row_off);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// ---------------------------------------------------------------------------
// CLIP ViT front end + short-sequence attention (see include/magma_hip.h)
// ---------------------------------------------------------------------------
namespace {
// one workgroup per patch; thread -> (c, py) row of P pixels
__global__ __launch_bounds__(256) void patchify_kernel(const mg_bf16* __restrict__ img, mg_bf16* __restrict__ out, int H, int W, int P) {
  const int gw = W / P, gh = H / P;
  const int patch = blockIdx.x, b = patch / (gh * gw), g = patch - b * gh * gw, gy = g / gw, gx = g - gy * gw;
  mg_bf16* dst = out + (int64_t)patch * (3 * P * P);
  for (int r = threadIdx.x; r < 3 * P; r += 256) {
    const int c = r / P, py = r - c * P;
    const mg_bf16* src = img + (((int64_t)b * 3 + c) * H + gy * P + py) * W + gx * P;
    for (int px = 0; px < P; ++px) dst[r * P + px] = src[px];
  }
}

__global__ __launch_bounds__(256) void vit_embed_kernel(const mg_bf16* __restrict__ patches, const mg_bf16* __restrict__ cls,
                                                        const mg_bf16* __restrict__ pos, mg_bf16* __restrict__ out, int G, int width) {
  const int row = blockIdx.x, b = row / (G + 1), t = row - b * (G + 1);
  const mg_bf16* src = t == 0 ? cls : patches + ((int64_t)b * G + (t - 1)) * width;
  for (int i = threadIdx.x; i < width; i += 256)
    out[(int64_t)row * width + i] = f2bf(bf2f(src[i]) + bf2f(pos[(int64_t)t * width + i]));
}

// one workgroup per (b, h); K and V rows of the head in LDS as fp32; thread = one query row
constexpr int AS_DH = 64, AS_MAXS = 256;
__global__ __launch_bounds__(256) void attn_small_kernel(const mg_bf16* __restrict__ qkv, mg_bf16* __restrict__ out, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ks = (float*)smem;                 // [S][65] (+1: bank skew)
  float* vs = ks + AS_MAXS * (AS_DH + 1);
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int w3 = 3 * H * AS_DH;
  for (int i = threadIdx.x; i < S * AS_DH; i += 256) {
    const int s = i / AS_DH, d = i - s * AS_DH;
    const mg_bf16* row = qkv + (int64_t)(b * S + s) * w3 + h * AS_DH + d;
    ks[s * (AS_DH + 1) + d] = bf2f(row[H * AS_DH]);
    vs[s * (AS_DH + 1) + d] = bf2f(row[2 * H * AS_DH]);
  }
  __syncthreads();
  const int qi = threadIdx.x;
  if (qi >= S) return;
  float q[AS_DH];
  const mg_bf16* qrow = qkv + (int64_t)(b * S + qi) * w3 + h * AS_DH;
#pragma unroll
  for (int d = 0; d < AS_DH; ++d) q[d] = bf2f(qrow[d]) * 0.125f;        // 1 / sqrt(64)
  float m = -1e30f, l = 0.f, acc[AS_DH];
#pragma unroll
  for (int d = 0; d < AS_DH; ++d) acc[d] = 0.f;
  for (int j = 0; j < S; ++j) {                                         // online softmax, fp32
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < AS_DH; ++d) sc += q[d] * ks[j * (AS_DH + 1) + d];
    const float mn = fmaxf(m, sc), a = __expf(m - mn), pj = __expf(sc - mn);
    l = l * a + pj;
#pragma unroll
    for (int d = 0; d < AS_DH; ++d) acc[d] = acc[d] * a + pj * vs[j * (AS_DH + 1) + d];
    m = mn;
  }
  const float inv = 1.0f / l;
  mg_bf16* orow = out + (int64_t)(b * S + qi) * (H * AS_DH) + h * AS_DH;
#pragma unroll
  for (int d = 0; d < AS_DH; ++d) orow[d] = f2bf(acc[d] * inv);
}
// Backward of attn_small_kernel.  One workgroup per (b, h), everything of the head in LDS as fp32 (S <= 64):
//   P = softmax(q k^T / 8)            dV = P^T dO            dP = dO V^T
//   dS = P o (dP - rowsum(P o dP)) / 8        dQ = dS K        dK = dS^T Q
// The sequences are tens of tokens (CLIP ViT-B/32: 50): plain fp32 FMAs, no MFMA -- 12 heads x B workgroups of ~1 MFLOP.
constexpr int ASB_MAXS = 64;
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const mg_bf16* __restrict__ qkv, const mg_bf16* __restrict__ dout,
                                                             mg_bf16* __restrict__ dqkv, int S, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LD = AS_DH + 1;
  float* qs = (float*)smem;            // [S][65]
  float* ks = qs + ASB_MAXS * LD;
  float* vs = ks + ASB_MAXS * LD;
  float* gs = vs + ASB_MAXS * LD;      // dO
  float* ps = gs + ASB_MAXS * LD;      // [S][S+1]  P, then dS
  float* ds = ps + ASB_MAXS * (ASB_MAXS + 1);   // [S][S+1]  dP
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int w = H * AS_DH, w3 = 3 * w, tid = threadIdx.x, LP = S + 1;
  for (int i = tid; i < S * AS_DH; i += 256) {
    const int s = i / AS_DH, d = i - s * AS_DH;
    const mg_bf16* row = qkv + (int64_t)(b * S + s) * w3 + h * AS_DH + d;
    qs[s * LD + d] = bf2f(row[0]);
    ks[s * LD + d] = bf2f(row[w]);
    vs[s * LD + d] = bf2f(row[2 * w]);
    gs[s * LD + d] = bf2f(dout[(int64_t)(b * S + s) * w + h * AS_DH + d]);
  }
  __syncthreads();
  for (int idx = tid; idx < S * S; idx += 256) {           // scores and dP
    const int i = idx / S, j = idx - i * S;
    float sc = 0.f, dp = 0.f;
#pragma unroll 8
    for (int d = 0; d < AS_DH; ++d) { sc += qs[i * LD + d] * ks[j * LD + d]; dp += gs[i * LD + d] * vs[j * LD + d]; }
    ps[i * LP + j] = sc * 0.125f;
    ds[i * LP + j] = dp;
  }
  __syncthreads();
  if (tid < S) {                                           // row softmax, then dS = P (dP - D) / 8 in place
    float* pr = ps + tid * LP;
    const float* dr = ds + tid * LP;
    float m = -1e30f;
    for (int j = 0; j < S; ++j) m = fmaxf(m, pr[j]);
    float l = 0.f;
    for (int j = 0; j < S; ++j) { const float e = __expf(pr[j] - m); pr[j] = e; l += e; }
    const float inv = 1.0f / l;
    float D = 0.f;
    for (int j = 0; j < S; ++j) { pr[j] *= inv; D += pr[j] * dr[j]; }
    float* dsr = ds + tid * LP;
    for (int j = 0; j < S; ++j) dsr[j] = pr[j] * (dr[j] - D) * 0.125f;     // ds <- dS; ps keeps P for dV
  }
  __syncthreads();
  for (int idx = tid; idx < S * AS_DH; idx += 256) {
    const int s = idx / AS_DH, d = idx - s * AS_DH;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < S; ++j) {
      dq += ds[s * LP + j] * ks[j * LD + d];               // dQ[s] = sum_j dS[s][j] K[j]
      dk += ds[j * LP + s] * qs[j * LD + d];               // dK[s] = sum_i dS[i][s] Q[i]
      dv += ps[j * LP + s] * gs[j * LD + d];               // dV[s] = sum_i P[i][s] dO[i]
    }
    mg_bf16* row = dqkv + (int64_t)(b * S + s) * w3 + h * AS_DH + d;
    row[0] = f2bf(dq);
    row[w] = f2bf(dk);
    row[2 * w] = f2bf(dv);
  }
}
}  // namespace

extern "C" int mg_attn_small_bwd_bf16(const mg_bf16* qkv, const mg_bf16* d_out, mg_bf16* d_qkv, int32_t B, int32_t S, int32_t H,
                                      void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || S > ASB_MAXS || !qkv || !d_out || !d_qkv) MG_FAIL(MG_ERR_SHAPE, "mg_attn_small_bwd_bf16: need 0 < S <= %d (head dim 64)", ASB_MAXS);
  const int lds = (4 * ASB_MAXS * (AS_DH + 1) + 2 * ASB_MAXS * (ASB_MAXS + 1)) * 4;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_small_bwd_kernel, lds, "mg_attn_small_bwd_bf16")) return rc;
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, d_out, d_qkv, S, H);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_patchify_bf16(const mg_bf16* img, mg_bf16* out, int32_t B, int32_t H, int32_t W, int32_t P, void* stream) {
  if (B <= 0 || P <= 0 || H <= 0 || W <= 0 || H % P || W % P || !img || !out) MG_FAIL(MG_ERR_SHAPE, "mg_patchify_bf16: need H, W multiples of the patch size");
  hipLaunchKernelGGL(patchify_kernel, dim3(B * (H / P) * (W / P)), dim3(256), 0, (hipStream_t)stream, img, out, H, W, P);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_vit_embed_bf16(const mg_bf16* patches, const mg_bf16* class_embedding, const mg_bf16* pos, mg_bf16* out,
                                 int32_t B, int32_t G, int32_t width, void* stream) {
  if (B <= 0 || G <= 0 || width <= 0 || !patches || !class_embedding || !pos || !out) MG_FAIL(MG_ERR_SHAPE, "mg_vit_embed_bf16: bad arguments");
  hipLaunchKernelGGL(vit_embed_kernel, dim3(B * (G + 1)), dim3(256), 0, (hipStream_t)stream, patches, class_embedding, pos, out, G, width);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_small_bf16(const mg_bf16* qkv, mg_bf16* out, int32_t B, int32_t S, int32_t H, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || S > AS_MAXS || !qkv || !out) MG_FAIL(MG_ERR_SHAPE, "mg_attn_small_bf16: need 0 < S <= %d (head dim 64)", AS_MAXS);
  const int lds = 2 * AS_MAXS * (AS_DH + 1) * 4;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_small_kernel, lds, "mg_attn_small_bf16")) return rc;
  hipLaunchKernelGGL(attn_small_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, qkv, out, S, H);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_avgpool2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_avgpool2_nhwc_bf16: need even H,W and C%%8==0");
  if (!x || !y || !MG_ALIGNED16(x) || !MG_ALIGNED16(y)) MG_FAIL(MG_ERR_ALIGN, "mg_avgpool2_nhwc_bf16: null/unaligned pointer");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_stem_im2col_bf16(const mg_bf16* img, mg_bf16* out, int32_t B, int32_t H, int32_t W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) MG_FAIL(MG_ERR_SHAPE, "mg_stem_im2col_bf16: need even H,W");
  if (!img || !out || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_stem_im2col_bf16: null/unaligned pointer");
  const int64_t total = (int64_t)B * (H / 2) * (W / 2);
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_argmax_f32(const float* logits, int64_t ld, int32_t B, int32_t V, int64_t* token, void* stream) {
  if (B <= 0 || V <= 0 || !logits || !token) MG_FAIL(MG_ERR_SHAPE, "mg_argmax_f32: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, ld, V, token);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_advance_pos(int32_t* d_pos, int32_t delta, void* stream) {
  if (!d_pos) MG_FAIL(MG_ERR_SHAPE, "mg_advance_pos: null pointer");
  hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_pos, delta);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_build_labels_i64(const int64_t* captions, int64_t* labels, int32_t B, int32_t S, int32_t P,
                                   int64_t eos, void* stream) {
  if (B <= 0 || S <= 0 || P < 0 || P > S) MG_FAIL(MG_ERR_SHAPE, "mg_build_labels_i64: captions.shape[1] (%d) must be >= prefix length (%d)", S, P);
  if (!captions || !labels) MG_FAIL(MG_ERR_SHAPE, "mg_build_labels_i64: null pointer");
  hipLaunchKernelGGL(build_labels_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, captions, labels, S, P, eos);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_ce_rows_f32(const float* logits, int64_t ld, const int64_t* tgt, float* loss_row, int32_t R,
                              int32_t V, void* stream) {
  if (R <= 0 || V <= 0 || !logits || !tgt || !loss_row) MG_FAIL(MG_ERR_SHAPE, "mg_ce_rows_f32: bad arguments");
  hipLaunchKernelGGL(ce_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, tgt, loss_row, V);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_ce_reduce_f32(const float* loss_row, const int64_t* tgt, int32_t R, float* out, void* stream) {
  if (R <= 0 || !loss_row || !tgt || !out) MG_FAIL(MG_ERR_SHAPE, "mg_ce_reduce_f32: bad arguments");
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_row, tgt, R, out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}


// ---------------------------------------------------------------------------
// bf16 rows -> OCP e4m3 with one fp32 scale per row (x ~= q * scale, scale = amax / 448).  One workgroup
// per row; the row is read twice (the second pass hits L2).  Columns [K, ldq) are zero-filled so the byte
// matrix can be fed to the GEMM with K rounded up.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const mg_bf16* __restrict__ x, int64_t ldx, int K,
                                                                uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const mg_bf16* xr = x + (int64_t)row * ldx;
  float amax = 0.f;
  for (int c = tid * 8; c < K; c += 256 * 8) {          // K % 8 == 0
    const u32x4 w = *(const u32x4*)(xr + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bflo(w[i])), fabsf(bfhi(w[i]))));
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / sc;
  if (tid == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * ldq;
  for (int c = tid * 8; c < (int)ldq; c += 256 * 8) {   // ldq % 8 == 0
    u32x2 o = {0u, 0u};
    if (c < K) {
      const u32x4 w = *(const u32x4*)(xr + c);
      float f[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(w[i]) * inv; f[2 * i + 1] = bfhi(w[i]) * inv; }
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i], -448.f), 448.f);
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      o[0] = (uint32_t)lo; o[1] = (uint32_t)hi;
    }
    *(u32x2*)(qr + c) = o;
  }
}
}  // namespace

// ---------------------------------------------------------------------------
// OCP MX (microscaling) quantiser: bf16 rows -> e4m3 elements with ONE E8M0 scale per 32 consecutive elements
// (OCP MX v1.0: shared exponent = floor(log2(max|x|)) - emax(e4m3 = 8), element = saturate_e4m3(x * 2^-shared)),
// the operand format of v_mfma_scale_f32_16x16x128_f8f6f4.  Elements: the plain layout, K order.  That IS what the instruction consumes
// (measured, tests/test_fp8_gpu.py::test_mx_mfma_lane_and_scale_semantics): lane l supplies row l & 15; its registers 0-3
// are k = 16 (l >> 4) .. + 15 and its registers 4-7 k = 64 + 16 (l >> 4) .. + 15 of the 128-wide chunk -- exactly the two
// 16-byte pieces the GEMM kernels' loaders already hand it -- and the scale of block b of a row is read from lane
// row + 16 b (so lane l supplies the scale byte of block l >> 4, although its own elements belong to two other blocks).
// Scales: bytes arranged so that ONE dword load gives a lane what it supplies for four 16-row MFMA fragments at once -- the
// dword index is ((chunk * 4 + block) * ceil(R / 64) + row / 64) * 16 + row % 16 and byte (row % 64) / 16 inside it is the
// E8M0 of (row, 4 * chunk + block): a wave's 64-row slab of a K-tile costs one load per lane, the MFMA picks the fragment's
// byte with its op_sel field (mx_scale_index below; ops.mx_scales_rowmajor undoes it for the tests).
// One workgroup per row, 4 lanes per block.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void quantize_mx_fp8_kernel(const mg_bf16* __restrict__ x, int64_t ldx, int K,
                                                              uint8_t* __restrict__ q, int64_t ldq,
                                                              uint8_t* __restrict__ scales, int rgroups) {
  const int row = blockIdx.x, tid = threadIdx.x;
  const mg_bf16* xr = x + (int64_t)row * ldx;
  uint8_t* qr = q + (int64_t)row * ldq;
  for (int c = tid * 8; c < (int)ldq; c += 256 * 8) {          // ldq % 128 == 0: whole blocks, whole quads of lanes
    u32x4 w = {0u, 0u, 0u, 0u};
    if (c < K) w = *(const u32x4*)(xr + c);                     // K % 8 == 0
    mx_emit8(w, qr, scales, rgroups, row, c);                   // common.h: the ONE statement of the MX rule (producer epilogues use it too)
  }
}

// one wave, one v_mfma_scale_f32_16x16x128_f8f6f4: lane l supplies 32 operand bytes + one scale dword per operand.
// Test-only probe of the instruction's lane / block / scale-byte semantics (tests/test_fp8_gpu.py).
__global__ __launch_bounds__(64) void mx_mfma_probe_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ sa,
                                                           const uint32_t* __restrict__ b, const uint32_t* __restrict__ sb,
                                                           float* __restrict__ out) {
  typedef __attribute__((ext_vector_type(8))) int i32x8_;
  const int l = threadIdx.x;
  i32x8_ av, bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = (int)a[l * 8 + i]; bv[i] = (int)b[l * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
}  // namespace

extern "C" int64_t mg_mx_scale_bytes(int32_t rows, int32_t K) {
  if (rows <= 0 || K <= 0) return 0;
  return (int64_t)((K + 127) / 128) * 4 * ((rows + 63) / 64) * 64;
}

extern "C" int mg_quantize_mx_fp8(const mg_bf16* x, int64_t ldx, int32_t M, int32_t K, uint8_t* q, int64_t ldq,
                                  uint8_t* scales, void* stream) {
  if (!x || !q || !scales) MG_FAIL(MG_ERR_SHAPE, "mg_quantize_mx_fp8: null pointer");
  if (M <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldq & 127) || ldq < K || ldx < K || ldq != ((K + 127) / 128) * 128)
    MG_FAIL(MG_ERR_SHAPE, "mg_quantize_mx_fp8: need K, ldx multiples of 8 and ldq == ceil(K / 128) * 128");
  if (!MG_ALIGNED16(x) || ((uintptr_t)q & 7) || ((uintptr_t)scales & 3)) MG_FAIL(MG_ERR_ALIGN, "mg_quantize_mx_fp8: x 16-byte, q 8-byte, scales 4-byte aligned");
  hipLaunchKernelGGL(quantize_mx_fp8_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, K, q, ldq, scales, (M + 63) / 64);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_debug_mx_mfma(const uint32_t* a, const uint32_t* scale_a, const uint32_t* b, const uint32_t* scale_b, float* out,
                                void* stream) {
  if (!a || !scale_a || !b || !scale_b || !out) MG_FAIL(MG_ERR_SHAPE, "mg_debug_mx_mfma: null pointer");
  hipLaunchKernelGGL(mx_mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, scale_a, b, scale_b, out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_quantize_rows_fp8(const mg_bf16* x, int64_t ldx, int32_t M, int32_t K, uint8_t* q, int64_t ldq,
                                    float* scale, void* stream) {
  if (!x || !q || !scale) MG_FAIL(MG_ERR_SHAPE, "mg_quantize_rows_fp8: null pointer");
  if (M <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldq & 7) || ldq < K || ldx < K) MG_FAIL(MG_ERR_SHAPE, "mg_quantize_rows_fp8: need K, ldx, ldq multiples of 8, ldq >= K");
  if (!MG_ALIGNED16(x) || ((uintptr_t)q & 7)) MG_FAIL(MG_ERR_ALIGN, "mg_quantize_rows_fp8: x must be 16-byte and q 8-byte aligned");
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, K, q, ldq, scale);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
