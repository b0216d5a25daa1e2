// backward.hip -- HBM-bound kernels of the training path (gfx950): transposes
// feeding the wgrad GEMMs, column sums (bias / affine gradients), LayerNorm
// backward, cross-entropy backward, inverse rotary merge, avg-pool backward,
// BatchNorm (frozen statistics) affine gradients, and the fused clip + AdamW.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// transpose: in [R, C] (ld_in) -> out [C, R] (ld_out); 64x64 tiles through LDS.
// R, C multiples of 8.  batched through blockIdx.z.
// ---------------------------------------------------------------------------
// COLSUM: also colsum[c] += sum_r in[r][c] (fp32 atomics, 64 per workgroup) from the tile already in LDS -- the bias gradient of a
// Linear whose weight gradient needs this very transpose (adapters: one pass over g instead of colsum_kernel + transpose_kernel).
template <bool COLSUM>
__global__ __launch_bounds__(256) void transpose_kernel(const mg_bf16* __restrict__ in, int64_t ld_in,
                                                        int64_t bs_in, mg_bf16* __restrict__ out,
                                                        int64_t ld_out, int64_t bs_out, int R, int C, float* __restrict__ colsum) {
  __shared__ mg_bf16 tile[64][64 + 2];
  const int tid = threadIdx.x;
  in += (int64_t)blockIdx.z * bs_in;
  out += (int64_t)blockIdx.z * bs_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int r = ci >> 3, cc = (ci & 7) * 8;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (r0 + r < R && c0 + cc < C) v = *(const u32x4*)(in + (int64_t)(r0 + r) * ld_in + c0 + cc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[r][cc + 2 * j] = (mg_bf16)(v[j] & 0xffffu);
      tile[r][cc + 2 * j + 1] = (mg_bf16)(v[j] >> 16);
    }
  }
  __syncthreads();
  if constexpr (COLSUM) {       // rows beyond R were stored as zeros: no bounds in the sum
    if (tid < 64 && c0 + tid < C) {
      float acc = 0.f;
#pragma unroll 8
      for (int r = 0; r < 64; ++r) acc += bf2f(tile[r][tid]);
      atomicAdd(colsum + c0 + tid, acc);
    }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int c = ci >> 3, rr = (ci & 7) * 8;   // output row c, 8 consecutive input rows
    if (c0 + c < C && r0 + rr < R) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (uint32_t)tile[rr + 2 * j][c] | ((uint32_t)tile[rr + 2 * j + 1][c] << 16);
      *(u32x4*)(out + (int64_t)(c0 + c) * ld_out + r0 + rr) = o;
    }
  }
}

// transpose + the parameter gradients of a frozen-statistics BatchNorm in ONE pass over g (the CLIP trunk's backward: g^T is the
// operand of the convolution's weight gradient, and bn_param_grad_kernel read the same g a second time from a grid of a few
// workgroups): out = g^T, dbeta[c] += sum_r g[r][c], dgamma[c] += sum_r g[r][c] * (y[r][c] - sub[r][c] - beta[c]) / gamma[c]
// -- element arithmetic of bn_param_grad_kernel; the sums leave the workgroup as 2 x 64 fp32 atomics.
__global__ __launch_bounds__(256) void transpose_bn_grad_kernel(const mg_bf16* __restrict__ in, int64_t ld_in, mg_bf16* __restrict__ out,
                                                                int64_t ld_out, int R, int C, const mg_bf16* __restrict__ y,
                                                                const mg_bf16* __restrict__ sub, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
  __shared__ mg_bf16 tile[64][64 + 2];
  __shared__ float prod[64][64 + 1];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int r = ci >> 3, cc = (ci & 7) * 8;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u}, yv = v, sv = v;
    float bt[8], ig[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bt[j] = 0.f; ig[j] = 0.f; }
    if (r0 + r < R && c0 + cc < C) {
      const int64_t off = (int64_t)(r0 + r) * ld_in + c0 + cc;
      v = *(const u32x4*)(in + off);
      yv = *(const u32x4*)(y + off);
      if (sub) sv = *(const u32x4*)(sub + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) { bt[j] = beta[c0 + cc + j]; const float gm = gamma[c0 + cc + j]; ig[j] = gm != 0.f ? 1.f / gm : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[r][cc + 2 * j] = (mg_bf16)(v[j] & 0xffffu);
      tile[r][cc + 2 * j + 1] = (mg_bf16)(v[j] >> 16);
      prod[r][cc + 2 * j] = bflo(v[j]) * (bflo(yv[j]) - bflo(sv[j]) - bt[2 * j]) * ig[2 * j];
      prod[r][cc + 2 * j + 1] = bfhi(v[j]) * (bfhi(yv[j]) - bfhi(sv[j]) - bt[2 * j + 1]) * ig[2 * j + 1];
    }
  }
  __syncthreads();
  if (tid < 128) {       // rows beyond R were stored as zeros: no bounds in the sums.  Threads 0-63: dbeta, 64-127: dgamma
    const int c = tid & 63;
    if (c0 + c < C) {
      float acc = 0.f;
      if (tid < 64) {
#pragma unroll 8
        for (int r = 0; r < 64; ++r) acc += bf2f(tile[r][c]);
        atomicAdd(dbeta + c0 + c, acc);
      } else {
#pragma unroll 8
        for (int r = 0; r < 64; ++r) acc += prod[r][c];
        atomicAdd(dgamma + c0 + c, acc);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int c = ci >> 3, rr = (ci & 7) * 8;
    if (c0 + c < C && r0 + rr < R) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (uint32_t)tile[rr + 2 * j][c] | ((uint32_t)tile[rr + 2 * j + 1][c] << 16);
      *(u32x4*)(out + (int64_t)(c0 + c) * ld_out + r0 + rr) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// head transpose: element (b, s, h, d) at src + b*sb + s*ss + h*sh + d  ->
// dst[(((b*H + h)*(ld/32) + s/32)*256 + d)*32 + s%32]  (column-tiled transposed layout); positions in
// [S, round_up(S,32)) zero filled.
// grid (ceil(S/32), B*H), 256 threads.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_transpose_kernel(const mg_bf16* __restrict__ src, int64_t sb,
                                                             int64_t ss, int64_t sh, mg_bf16* __restrict__ dst,
                                                             int ld, int H, int S) {
  __shared__ __attribute__((aligned(16))) mg_bf16 tile[32 * 256];
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int s0 = blockIdx.x * 32;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (s0 + row < S) v = *(const u32x4*)(src + b * sb + (int64_t)(s0 + row) * ss + h * sh + c * 8);
    *(u32x4*)(tile + row * 256 + c * 8) = v;
  }
  __syncthreads();
  // column-tiled transposed layout [b,h][ld/32 tiles][256][32]: every 32-position tile is 16 KiB contiguous, so
  // the attention kernels' LDS-DMA reads whole cache lines (64-byte row pieces cost 30 % of the fill rate)
  mg_bf16* d = dst + (((int64_t)bh * (ld >> 5) + blockIdx.x) * 256 + tid) * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      o[w] = (uint32_t)tile[(g * 8 + w * 2) * 256 + tid] | ((uint32_t)tile[(g * 8 + w * 2 + 1) * 256 + tid] << 16);
    *(u32x4*)(d + g * 8) = o;
  }
}

// ---------------------------------------------------------------------------
// column sums: out[n] += sum_m x[m][n] * (y ? y[m][n] : 1)   (fp32 atomics)
// grid (ceil(N/512), ceil(M/rows_per_block))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const mg_bf16* __restrict__ x, int64_t ldx,
                                                     const mg_bf16* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ out, int M, int N, int rows_per_block) {
  __shared__ float red[4][64][8];
  const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int n = blockIdx.x * 512 + cl * 8;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int m = m0 + rl; m < m1; m += 4) {
      const u32x4 a = *(const u32x4*)(x + (int64_t)m * ldx + n);
      if (y) {
        const u32x4 b = *(const u32x4*)(y + (int64_t)m * ldy + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += bflo(a[j]) * bflo(b[j]); acc[2 * j + 1] += bfhi(a[j]) * bfhi(b[j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[2 * j] += bflo(a[j]); acc[2 * j + 1] += bfhi(a[j]); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cl][j] = acc[j];
  __syncthreads();
  if (rl == 0 && n < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (n + j < N) atomicAdd(out + n + j, red[0][cl][j] + red[1][cl][j] + red[2][cl][j] + red[3][cl][j]);
  }
}

// ---------------------------------------------------------------------------
// LayerNorm backward (input gradient), one workgroup per row:
//   xh = (x-mu)*rstd ; g = dy*gamma ; dx = rstd*(g - mean(g) - xh*mean(g*xh)) (+ res)
// optionally writes xh (bf16) for the affine-gradient column sums.
// ---------------------------------------------------------------------------
constexpr int LNB_MAXV = 4;  // d <= 8192

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const mg_bf16* __restrict__ dy, int64_t lddy,
                                                            const mg_bf16* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const mg_bf16* __restrict__ res, int64_t ldr,
                                                            mg_bf16* __restrict__ dx, int64_t lddx,
                                                            mg_bf16* __restrict__ xhat, int64_t ldxh, int d,
                                                            float eps) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x, nvec = d >> 3;
  float xv[LNB_MAXV][8], gv[LNB_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
      const u32x4 w = *(const u32x4*)(x + (int64_t)row * ldx + vi * 8);
      const u32x4 g = *(const u32x4*)(dy + (int64_t)row * lddy + vi * 8);
      const float4 g0 = *(const float4*)(gamma + vi * 8), g1 = *(const float4*)(gamma + vi * 8 + 4);
      const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[i][2 * j] = bflo(w[j]); xv[i][2 * j + 1] = bfhi(w[j]);
        gv[i][2 * j] = bflo(g[j]) * gm[2 * j]; gv[i][2 * j + 1] = bfhi(g[j]) * gm[2 * j + 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[i][j];
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i)
    if (tid + i * 256 < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float t = xv[i][j] - mean; q += t * t; }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i)
    if (tid + i * 256 < nvec)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xv[i][j] = (xv[i][j] - mean) * rstd;   // xhat
        sg += gv[i][j];
        sgx += gv[i][j] * xv[i][j];
      }
  sg = wave_sum(sg); sgx = wave_sum(sgx);
  if (lane == 0) { red[8 + wave] = sg; red[12 + wave] = sgx; }
  __syncthreads();
  const float mg_ = (red[8] + red[9] + red[10] + red[11]) / (float)d;
  const float mgx = (red[12] + red[13] + red[14] + red[15]) / (float)d;
#pragma unroll
  for (int i = 0; i < LNB_MAXV; ++i) {
    const int vi = tid + i * 256;
    if (vi < nvec) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd * (gv[i][j] - mg_ - xv[i][j] * mgx);
      if (res) {
        const u32x4 r = *(const u32x4*)(res + (int64_t)row * ldr + vi * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] += bflo(r[j]); o[2 * j + 1] += bfhi(r[j]); }
      }
      u32x4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = pack2bf(o[2 * j], o[2 * j + 1]);
      *(u32x4*)(dx + (int64_t)row * lddx + vi * 8) = w;
      if (xhat) {
        u32x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = pack2bf(xv[i][2 * j], xv[i][2 * j + 1]);
        *(u32x4*)(xhat + (int64_t)row * ldxh + vi * 8) = h;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// cross-entropy backward: dlogits[r][v] = (softmax(logits[r])[v] - [v==tgt]) * inv_n
// inv_n = 1 / stats[1] (valid-target count written by ce_reduce).  bf16 out,
// columns [V, ldo) zero filled (they are the K padding of the dgrad GEMM).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                     const int64_t* __restrict__ tgt,
                                                     const float* __restrict__ stats,
                                                     mg_bf16* __restrict__ out, int64_t ldo, int V) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int64_t tg = tgt[r];
  const float* row = logits + (int64_t)r * ld;
  mg_bf16* orow = out + (int64_t)r * ldo;
  if (tg < 0 || tg >= V) {
    for (int i = tid; i < ldo; i += 256) orow[i] = 0;
    return;
  }
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
  const float inv_sum = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  const float inv_n = 1.0f / stats[1];
  for (int i = tid; i < ldo; i += 256) {
    float v = 0.f;
    if (i < V) v = (expf(row[i] - mx) * inv_sum - (i == tg ? 1.f : 0.f)) * inv_n;
    orow[i] = f2bf(v);
  }
}

// ---------------------------------------------------------------------------
// inverse rotary + merge: dq,dk,dv [B,H,S,256] -> dqkv [B*S, 3*H*256]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotary_merge_bwd_kernel(const mg_bf16* __restrict__ dq,
                                                               const mg_bf16* __restrict__ dk,
                                                               const mg_bf16* __restrict__ dv, int B, int S, int H,
                                                               int rot_dim, const float* __restrict__ sin_t,
                                                               const float* __restrict__ cos_t,
                                                               mg_bf16* __restrict__ dqkv) {
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int s0 = blockIdx.x * 32;
  const int dmodel = H * 256, half_rot = rot_dim >> 1;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31, s = s0 + row, d0 = c * 8;
    if (s >= S) continue;
    const int64_t src = ((int64_t)bh * S + s) * 256 + d0;
    u32x4 qv = *(const u32x4*)(dq + src), kv = *(const u32x4*)(dk + src);
    const u32x4 vv = *(const u32x4*)(dv + src);
    if (d0 < rot_dim) {
      const float* sp = sin_t + (int64_t)s * half_rot + (d0 >> 1);
      const float* cp = cos_t + (int64_t)s * half_rot + (d0 >> 1);
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        const float sn = sp[pi], cs = cp[pi];
        const float q0 = bflo(qv[pi]), q1 = bfhi(qv[pi]), k0 = bflo(kv[pi]), k1 = bfhi(kv[pi]);
        qv[pi] = pack2bf(q0 * cs + q1 * sn, q1 * cs - q0 * sn);   // R(-theta)
        kv[pi] = pack2bf(k0 * cs + k1 * sn, k1 * cs - k0 * sn);
      }
    }
    mg_bf16* base = dqkv + (int64_t)(b * S + s) * (3 * dmodel) + h * 256 + d0;
    *(u32x4*)base = qv;
    *(u32x4*)(base + dmodel) = kv;
    *(u32x4*)(base + 2 * dmodel) = vv;
  }
}

// ---------------------------------------------------------------------------
// avg-pool 2x2 backward (NHWC): dx[b,2y+i,2x+j,c] = 0.25*dy[b,y,x,c] (optionally
// gated by aux > 0 and accumulated with add)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const mg_bf16* __restrict__ dy,
                                                           const mg_bf16* __restrict__ gate,
                                                           mg_bf16* __restrict__ dx, int B, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, cv = C >> 3;
  const int64_t total = (int64_t)B * H * W * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cv);
    int64_t t = i / cv;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const u32x4 g = *(const u32x4*)(dy + ((((int64_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c * 8));
    u32x4 gt = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (gate) gt = *(const u32x4*)(gate + i * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(bflo(gt[j]) > 0.f ? 0.25f * bflo(g[j]) : 0.f, bfhi(gt[j]) > 0.f ? 0.25f * bfhi(g[j]) : 0.f);
    *(u32x4*)(dx + i * 8) = o;
  }
}

// elementwise helpers ------------------------------------------------------------
// out = a (+ b) gated by (gate > 0) when gate != null   (bf16, 16-B vectors)
__global__ __launch_bounds__(256) void add_gate_kernel(const mg_bf16* __restrict__ a, const mg_bf16* __restrict__ b,
                                                       const mg_bf16* __restrict__ gate, mg_bf16* __restrict__ out,
                                                       int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4 x = ((const u32x4*)a)[i];
    u32x4 y = (u32x4){0u, 0u, 0u, 0u}, g = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    if (b) y = ((const u32x4*)b)[i];
    if (gate) g = ((const u32x4*)gate)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = (bflo(g[j]) > 0.f) ? bflo(x[j]) + bflo(y[j]) : 0.f;
      const float hi = (bfhi(g[j]) > 0.f) ? bfhi(x[j]) + bfhi(y[j]) : 0.f;
      o[j] = pack2bf(lo, hi);
    }
    ((u32x4*)out)[i] = o;
  }
}

// BatchNorm with frozen statistics: y = conv*scale + shift (scale = gamma*rstd).
// dgamma[c] += sum_m g[m][c] * (ybn[m][c] - beta[c]) / gamma[c] ; dbeta[c] += sum_m g[m][c]
// where ybn = y - (sub ? sub : 0) (the pre-residual BN output) and g is already
// gated by the ReLU.  grid (ceil(C/512), row blocks)
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const mg_bf16* __restrict__ g, const mg_bf16* __restrict__ y,
                                                            const mg_bf16* __restrict__ sub,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int M, int C, int rows_per_block) {
  __shared__ float red[2][4][64][8];
  const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int n = blockIdx.x * 512 + cl * 8;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float ag[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ab[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < C) {
    float bt[8], ig[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bt[j] = beta[n + j]; const float gm = gamma[n + j]; ig[j] = gm != 0.f ? 1.f / gm : 0.f; }
    // four rows in flight per wave: a workgroup walks 256 rows with one wave per row -- 64 dependent round trips to HBM when
    // every iteration waits for its own loads (the launch has fewer waves than the chip has SIMDs)
    for (int m = m0 + rl; m < m1; m += 16) {
      u32x4 gv[4], yv[4], sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int mm = m + 4 * u;
        gv[u] = yv[u] = sv[u] = (u32x4){0u, 0u, 0u, 0u};      // a missing row adds g = 0 to both sums
        if (mm < m1) {
          gv[u] = *(const u32x4*)(g + (int64_t)mm * C + n);
          yv[u] = *(const u32x4*)(y + (int64_t)mm * C + n);
          if (sub) sv[u] = *(const u32x4*)(sub + (int64_t)mm * C + n);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g0 = bflo(gv[u][j]), g1 = bfhi(gv[u][j]);
          ab[2 * j] += g0; ab[2 * j + 1] += g1;
          ag[2 * j] += g0 * (bflo(yv[u][j]) - bflo(sv[u][j]) - bt[2 * j]) * ig[2 * j];
          ag[2 * j + 1] += g1 * (bfhi(yv[u][j]) - bfhi(sv[u][j]) - bt[2 * j + 1]) * ig[2 * j + 1];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][rl][cl][j] = ag[j]; red[1][rl][cl][j] = ab[j]; }
  __syncthreads();
  if (rl == 0 && n < C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(dgamma + n + j, red[0][0][cl][j] + red[0][1][cl][j] + red[0][2][cl][j] + red[0][3][cl][j]);
      atomicAdd(dbeta + n + j, red[1][0][cl][j] + red[1][1][cl][j] + red[1][2][cl][j] + red[1][3][cl][j]);
    }
  }
}

// im2col^T for the 3x3 wgrad: out[(ci*9 + tap)][m] = x[(b,y+ky-1,x+kx-1)][ci]  (row order =
// the conv weight's own [cin,3,3] flattening, so dW needs no re-layout)
// (zero outside), out row stride ldo >= M.  grid (ceil(M/64), Cin/64.. , 9)
__global__ __launch_bounds__(256) void im2col_t_kernel(const mg_bf16* __restrict__ x, mg_bf16* __restrict__ out,
                                                       int64_t ldo, int B, int H, int W, int Cin) {
  __shared__ mg_bf16 tile[64][64 + 2];
  const int tid = threadIdx.x;
  const int tap = blockIdx.z, ky = tap / 3, kx = tap - ky * 3;
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int M = B * H * W;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int r = ci >> 3, cc = (ci & 7) * 8;
    const int m = m0 + r;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (m < M && c0 + cc < Cin) {
      const int xx = m % W, t = m / W, yy = t % H;
      const int y2 = yy + ky - 1, x2 = xx + kx - 1;
      if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)
        v = *(const u32x4*)(x + ((int64_t)m + (int64_t)(ky - 1) * W + (kx - 1)) * Cin + c0 + cc);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[r][cc + 2 * j] = (mg_bf16)(v[j] & 0xffffu);
      tile[r][cc + 2 * j + 1] = (mg_bf16)(v[j] >> 16);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = tid + it * 256;
    const int c = ci >> 3, rr = (ci & 7) * 8;
    if (c0 + c < Cin && m0 + rr < M) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (uint32_t)tile[rr + 2 * j][c] | ((uint32_t)tile[rr + 2 * j + 1][c] << 16);
      *(u32x4*)(out + ((int64_t)(c0 + c) * 9 + tap) * ldo + m0 + rr) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// batch-statistics BatchNorm (what the reference's tower runs once train.py:182 has flipped it to train mode).
// The conv GEMM writes the raw output z; per-channel sums come from colsum_kernel; then:
//   bn_batch_fold   mean, biased var -> scale = gamma * rstd, shift = beta - mean * scale; running statistics updated as
//                   nn.BatchNorm2d does (momentum, unbiased variance)
//   bn_apply        y = [relu](z * scale + shift [+ residual])
//   bn_bwd_dz       dz = gamma * rstd * (g - dbeta / M - xhat * dgamma / M),  xhat = (z - mean) * rstd
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_batch_fold_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float inv_m, float unbias, float eps, float momentum,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mean = sum[c] * inv_m;
  const float var = fmaxf(sumsq[c] * inv_m - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc; shift[c] = beta[c] - mean * sc; mean_out[c] = mean; rstd_out[c] = rstd;
  if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
  if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * unbias;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const mg_bf16* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const mg_bf16* __restrict__ res,
                                                       int relu, mg_bf16* __restrict__ y, int64_t nvec, int cvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cvec) * 8;
    const u32x4 zv = ((const u32x4*)z)[i];
    u32x4 rv = (u32x4){0u, 0u, 0u, 0u};
    if (res) rv = ((const u32x4*)res)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = bflo(zv[j]) * scale[c + 2 * j] + shift[c + 2 * j] + bflo(rv[j]);
      float b = bfhi(zv[j]) * scale[c + 2 * j + 1] + shift[c + 2 * j + 1] + bfhi(rv[j]);
      if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
      o[j] = pack2bf(a, b);
    }
    ((u32x4*)y)[i] = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const mg_bf16* __restrict__ g, const mg_bf16* __restrict__ z,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta, float inv_m,
                                                        mg_bf16* __restrict__ dz, int64_t nvec, int cvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % cvec) * 8;
    const u32x4 gv = ((const u32x4*)g)[i], zv = ((const u32x4*)z)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cc = c + 2 * j + h;
        const float gg = h ? bfhi(gv[j]) : bflo(gv[j]), zz = h ? bfhi(zv[j]) : bflo(zv[j]);
        const float xh = (zz - mean[cc]) * rstd[cc];
        r[h] = gamma[cc] * rstd[cc] * (gg - dbeta[cc] * inv_m - xh * dgamma[cc] * inv_m);
      }
      o[j] = pack2bf(r[0], r[1]);
    }
    ((u32x4*)dz)[i] = o;
  }
}

// ---------------------------------------------------------------------------
// optimizer: sum of squares (global grad norm) and fused clip + AdamW
// ---------------------------------------------------------------------------
// gradient element as fp32: fp32 flat gradients (single GPU) or the bf16 buckets that went through RCCL (DP)
MG_DEV float grad_ld(const float* g, int64_t i) { return g[i]; }
MG_DEV float grad_ld(const mg_bf16* g, int64_t i) { return bf2f(g[i]); }

// (VEC: 16 bytes per lane and load -- 4 fp32 / 8 bf16 gradients --, two loads in flight per thread; the 4-byte form streamed at 2 TB/s)
MG_DEV float sumsq16(const float* g, int64_t q) { const f32x4 v = ((const f32x4*)g)[q]; return v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
MG_DEV float sumsq16(const mg_bf16* g, int64_t q) {
  const u32x4 v = ((const u32x4*)g)[q];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float a = bflo(v[j]), b = bfhi(v[j]); s += a * a + b * b; }
  return s;
}
template <typename G, bool VEC = false>
__global__ __launch_bounds__(256) void sumsq_kernel(const G* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  if constexpr (VEC) {
    constexpr int E = 16 / (int)sizeof(G);
    const int64_t nq = n / E, stride = (int64_t)gridDim.x * 256;
    float s1 = 0.f;
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; q + stride < nq; q += 2 * stride) { s += sumsq16(g, q); s1 += sumsq16(g, q + stride); }
    if (q < nq) s += sumsq16(g, q);
    s += s1;
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nq * E)) { const float v = grad_ld(g, nq * E + threadIdx.x); s += v * v; }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float v = grad_ld(g, i); s += v * v; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// p (fp32 master), m, v, g (gradient SUM over micro-steps and ranks; grad_scale = 1 / (gas * world) turns it into the
// mean); writes the bf16 model copy.  clip = min(1, max_norm / (sqrt(*norm_sq) + 1e-6)).
template <typename G, bool VEC = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    const G* __restrict__ g, mg_bf16* __restrict__ p_bf16,
                                                    int64_t n, float lr, float beta1, float beta2, float eps,
                                                    float wd, float bc1, float bc2, float max_norm,
                                                    const float* __restrict__ norm_sq, float grad_scale) {
  float clip = grad_scale;
  if (norm_sq && max_norm > 0.f) {
    const float nrm = sqrtf(*norm_sq) * grad_scale;
    clip = grad_scale * fminf(1.0f, max_norm / (nrm + 1e-6f));
  }
  auto update = [&](float graw, float& pi, float& mi, float& vi) {      // ONE element: the arithmetic of both forms below
    const float gi = graw * clip;
    pi -= lr * wd * pi;                                  // decoupled weight decay (torch.optim.AdamW)
    mi = beta1 * mi + (1.f - beta1) * gi;
    vi = beta2 * vi + (1.f - beta2) * gi * gi;
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  };
  int64_t i0 = 0;
  if constexpr (VEC) {
    // four elements per thread and iteration as 16-byte accesses (alone both forms stream at 4.7 TB/s: profiles/r06_adamw_forms.jsonl)
    const int64_t nq = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
      f32x4 pv = ((const f32x4*)p)[q], mv = ((const f32x4*)m)[q], vv = ((const f32x4*)v)[q];
      float gr[4];
      if constexpr (sizeof(G) == 4) { const f32x4 gv = ((const f32x4*)g)[q]; gr[0] = gv[0]; gr[1] = gv[1]; gr[2] = gv[2]; gr[3] = gv[3]; }
      else { const u32x2 gv = ((const u32x2*)g)[q]; gr[0] = bflo(gv[0]); gr[1] = bfhi(gv[0]); gr[2] = bflo(gv[1]); gr[3] = bfhi(gv[1]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { float pj = pv[j], mj = mv[j], vj = vv[j]; update(gr[j], pj, mj, vj); pv[j] = pj; mv[j] = mj; vv[j] = vj; }
      ((f32x4*)m)[q] = mv; ((f32x4*)v)[q] = vv; ((f32x4*)p)[q] = pv;
      if (p_bf16) { u32x2 o; o[0] = pack2bf(pv[0], pv[1]); o[1] = pack2bf(pv[2], pv[3]); ((u32x2*)p_bf16)[q] = o; }
    }
    i0 = nq << 2;
    if (blockIdx.x != 0) return;                          // the (< 4 element) tail: first workgroup
  }
  for (int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    update(grad_ld(g, i), pi, mi, vi);
    m[i] = mi; v[i] = vi;
    p[i] = pi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

// fp32 flat gradients -> bf16 exchange bucket (n % 4 == 0: one 16-byte load, one 8-byte store per thread)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, mg_bf16* __restrict__ dst, int64_t nq) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)src)[i];
    u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
    ((u32x2*)dst)[i] = o;
  }
}

// out = a * b (bf16) -- dropout backward
__global__ __launch_bounds__(256) void mul_kernel(const mg_bf16* __restrict__ a, const mg_bf16* __restrict__ b,
                                                  mg_bf16* __restrict__ out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4 x = ((const u32x4*)a)[i], y = ((const u32x4*)b)[i];
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2bf(bflo(x[j]) * bflo(y[j]), bfhi(x[j]) * bfhi(y[j]));
    ((u32x4*)out)[i] = o;
  }
}

// torch.nn.GELU() (erf form) and its derivative as stand-alone element-wise passes over [rows, cols] bf16 matrices with row
// strides (cols % 8 == 0): y = gelu(x), and g *= gelu'(pre).  Only adapters built with activation=nn.GELU (reference
// adapters.py:11,20 -- an option no shipped config sets) come here; the erf polynomial stays out of the GEMM epilogues, where
// it cost the 256x256 kernels registers (40-byte spills in the shipped bf16 kernel when it was tried there).
MG_DEV float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
MG_DEV float gelu_erf_grad_f(float x) {      // Phi(x) + x phi(x)
  return 0.5f * (1.0f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
template <bool GRAD>
__global__ __launch_bounds__(256) void gelu_erf_kernel(const mg_bf16* __restrict__ x, int64_t ldx, const mg_bf16* __restrict__ g,
                                                       int64_t ldg, mg_bf16* __restrict__ y, int64_t ldy, int rows, int cv) {
  const int64_t total = (int64_t)rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cv), c = (int)(i - (int64_t)r * cv) * 8;
    const u32x4 v = *(const u32x4*)(x + (int64_t)r * ldx + c);
    u32x4 o;
    if constexpr (GRAD) {
      const u32x4 gv = *(const u32x4*)(g + (int64_t)r * ldg + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack2bf(bflo(gv[j]) * gelu_erf_grad_f(bflo(v[j])), bfhi(gv[j]) * gelu_erf_grad_f(bfhi(v[j])));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack2bf(gelu_erf_f(bflo(v[j])), gelu_erf_f(bfhi(v[j])));
    }
    *(u32x4*)(y + (int64_t)r * ldy + c) = o;
  }
}

// dst[r*cols + c] += src[r*lds + c] * (row_scale ? row_scale[r] : 1)   (fp32 gradient accumulation)
__global__ __launch_bounds__(256) void scale_rows_acc_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                             int64_t lds_, const float* __restrict__ row_scale,
                                                             int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    dst[i] += src[(int64_t)r * lds_ + c] * (row_scale ? row_scale[r] : 1.0f);
  }
}

inline int grid_for(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int mg_transpose_bf16(const mg_bf16* in, int64_t ld_in, int64_t bs_in, mg_bf16* out, int64_t ld_out,
                                 int64_t bs_out, int32_t R, int32_t C, int32_t batch, void* stream) {
  if (R <= 0 || C <= 0 || batch <= 0 || (C & 7) || ld_out < ((R + 7) & ~7)) MG_FAIL(MG_ERR_SHAPE, "mg_transpose_bf16: C%%8==0 and ld_out >= round_up(R,8) required (padding columns are zero filled)");
  if (!in || !out || !MG_ALIGNED16(in) || !MG_ALIGNED16(out) || (ld_in & 7) || (ld_out & 7) || (bs_in & 7) || (bs_out & 7))
    MG_FAIL(MG_ERR_ALIGN, "mg_transpose_bf16: 16-byte alignment required");
  dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
  hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, bs_in, out, ld_out, bs_out, R, C, (float*)nullptr);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// g [R, C] -> out = g^T [C, ld_out] (as mg_transpose_bf16) while accumulating the frozen-statistics BatchNorm parameter gradients of
// mg_bn_param_grad_f32 (y, sub: the layout of g; sub may be NULL) -- one pass over g instead of two.
extern "C" int mg_transpose_bn_param_grad_bf16(const mg_bf16* g, int64_t ld_in, mg_bf16* out, int64_t ld_out, int32_t R, int32_t C,
                                               const mg_bf16* y, const mg_bf16* sub, const float* gamma, const float* beta,
                                               float* dgamma, float* dbeta, void* stream) {
  if (R <= 0 || C <= 0 || (C & 7) || ld_out < ((R + 7) & ~7)) MG_FAIL(MG_ERR_SHAPE, "mg_transpose_bn_param_grad_bf16: C%%8==0 and ld_out >= round_up(R,8) required");
  if (!g || !out || !y || !gamma || !beta || !dgamma || !dbeta || !MG_ALIGNED16(g) || !MG_ALIGNED16(out) || !MG_ALIGNED16(y) || !MG_ALIGNED16(sub) ||
      (ld_in & 7) || (ld_out & 7))
    MG_FAIL(MG_ERR_ALIGN, "mg_transpose_bn_param_grad_bf16: null pointer or 16-byte alignment violated");
  dim3 grid((C + 63) / 64, (R + 63) / 64, 1);
  hipLaunchKernelGGL(transpose_bn_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, ld_in, out, ld_out, R, C, y, sub, gamma, beta, dgamma, dbeta);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// The same transpose (one matrix) that also ACCUMULATES the column sums of `in` into colsum[C] (fp32): d bias and the operand of d weight
// of a Linear from one pass over its output gradient.
extern "C" int mg_transpose_colsum_bf16(const mg_bf16* in, int64_t ld_in, mg_bf16* out, int64_t ld_out, int32_t R, int32_t C,
                                        float* colsum, void* stream) {
  if (R <= 0 || C <= 0 || (C & 7) || ld_out < ((R + 7) & ~7)) MG_FAIL(MG_ERR_SHAPE, "mg_transpose_colsum_bf16: C%%8==0 and ld_out >= round_up(R,8) required");
  if (!in || !out || !colsum || !MG_ALIGNED16(in) || !MG_ALIGNED16(out) || (ld_in & 7) || (ld_out & 7))
    MG_FAIL(MG_ERR_ALIGN, "mg_transpose_colsum_bf16: 16-byte alignment required");
  dim3 grid((C + 63) / 64, (R + 63) / 64, 1);
  hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, (int64_t)0, out, ld_out, (int64_t)0, R, C, colsum);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_head_transpose_bf16(const mg_bf16* src, int64_t sb, int64_t ss, int64_t sh, mg_bf16* dst, int32_t ld,
                                      int32_t B, int32_t H, int32_t S, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || (ld & 31) || ld < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_head_transpose_bf16: ld must be a multiple of 32 and >= S");
  if (!src || !dst || !MG_ALIGNED16(src) || !MG_ALIGNED16(dst) || (sb & 7) || (ss & 7) || (sh & 7)) MG_FAIL(MG_ERR_ALIGN, "mg_head_transpose_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(head_transpose_kernel, dim3((S + 31) / 32, B * H), dim3(256), 0, (hipStream_t)stream, src, sb, ss, sh, dst, ld, H, S);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_colsum_f32(const mg_bf16* x, int64_t ldx, const mg_bf16* y, int64_t ldy, float* out, int32_t M,
                             int32_t N, void* stream) {
  if (M <= 0 || N <= 0 || (N & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_colsum_f32: N must be a positive multiple of 8");
  if (!x || !out || !MG_ALIGNED16(x) || !MG_ALIGNED16(y) || (ldx & 7) || (y && (ldy & 7))) MG_FAIL(MG_ERR_ALIGN, "mg_colsum_f32: 16-byte alignment required");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, out, M, N, rpb);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_layernorm_bwd_bf16(const mg_bf16* dy, int64_t lddy, const mg_bf16* x, int64_t ldx, const float* gamma,
                                     const mg_bf16* res, int64_t ldr, mg_bf16* dx, int64_t lddx, mg_bf16* xhat,
                                     int64_t ldxh, int32_t rows, int32_t d, float eps, void* stream) {
  if (rows <= 0 || d <= 0 || (d & 7) || d > 256 * 8 * LNB_MAXV) MG_FAIL(MG_ERR_SHAPE, "mg_layernorm_bwd_bf16: need d%%8==0 and d<=%d", 256 * 8 * LNB_MAXV);
  if (!dy || !x || !gamma || !dx) MG_FAIL(MG_ERR_SHAPE, "mg_layernorm_bwd_bf16: null pointer");
  if (!MG_ALIGNED16(dy) || !MG_ALIGNED16(x) || !MG_ALIGNED16(gamma) || !MG_ALIGNED16(res) || !MG_ALIGNED16(dx) || !MG_ALIGNED16(xhat) ||
      (lddy & 7) || (ldx & 7) || (lddx & 7) || (res && (ldr & 7)) || (xhat && (ldxh & 7)))
    MG_FAIL(MG_ERR_ALIGN, "mg_layernorm_bwd_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx, gamma, res, ldr, dx, lddx, xhat, ldxh, d, eps);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_ce_bwd_bf16(const float* logits, int64_t ld, const int64_t* tgt, const float* stats, mg_bf16* out,
                              int64_t ldo, int32_t R, int32_t V, void* stream) {
  if (R <= 0 || V <= 0 || ldo < V || !logits || !tgt || !stats || !out) MG_FAIL(MG_ERR_SHAPE, "mg_ce_bwd_bf16: bad arguments");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, tgt, stats, out, ldo, V);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_rotary_merge_bwd_bf16(const mg_bf16* dq, const mg_bf16* dk, const mg_bf16* dv, int32_t B, int32_t S,
                                        int32_t H, int32_t rot_dim, const float* sin_t, const float* cos_t,
                                        mg_bf16* dqkv, void* stream) {
  if (B <= 0 || S <= 0 || H <= 0 || rot_dim < 0 || rot_dim > 256 || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_merge_bwd_bf16: bad shape");
  if (!dq || !dk || !dv || !dqkv || !MG_ALIGNED16(dq) || !MG_ALIGNED16(dk) || !MG_ALIGNED16(dv) || !MG_ALIGNED16(dqkv)) MG_FAIL(MG_ERR_ALIGN, "mg_rotary_merge_bwd_bf16: null/unaligned pointer");
  hipLaunchKernelGGL(rotary_merge_bwd_kernel, dim3((S + 31) / 32, B * H), dim3(256), 0, (hipStream_t)stream, dq, dk, dv, B, S, H, rot_dim, sin_t, cos_t, dqkv);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_avgpool2_bwd_nhwc_bf16(const mg_bf16* dy, const mg_bf16* gate, mg_bf16* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_avgpool2_bwd_nhwc_bf16: need even H,W (input size) and C%%8==0");
  if (!dy || !dx || !MG_ALIGNED16(dy) || !MG_ALIGNED16(dx) || !MG_ALIGNED16(gate)) MG_FAIL(MG_ERR_ALIGN, "mg_avgpool2_bwd_nhwc_bf16: null/unaligned pointer");
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, dy, gate, dx, B, H, W, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_add_gate_bf16(const mg_bf16* a, const mg_bf16* b, const mg_bf16* gate, mg_bf16* out, int64_t n, void* stream) {
  if (n <= 0 || (n & 7) || !a || !out) MG_FAIL(MG_ERR_SHAPE, "mg_add_gate_bf16: n must be a positive multiple of 8");
  if (!MG_ALIGNED16(a) || !MG_ALIGNED16(b) || !MG_ALIGNED16(gate) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_add_gate_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(add_gate_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, gate, out, n / 8);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bn_param_grad_f32(const mg_bf16* g, const mg_bf16* y, const mg_bf16* sub, const float* gamma,
                                    const float* beta, float* dgamma, float* dbeta, int32_t M, int32_t C, void* stream) {
  if (M <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_bn_param_grad_f32: C must be a positive multiple of 8");
  if (!g || !y || !gamma || !beta || !dgamma || !dbeta || !MG_ALIGNED16(g) || !MG_ALIGNED16(y) || !MG_ALIGNED16(sub)) MG_FAIL(MG_ERR_ALIGN, "mg_bn_param_grad_f32: null/unaligned pointer");
  const int rpb = 256;
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 511) / 512, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, g, y, sub, gamma, beta, dgamma, dbeta, M, C, rpb);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bn_batch_fold_f32(const float* sum, const float* sumsq, const float* gamma, const float* beta, int64_t M,
                                    float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                    float* shift, float* mean, float* rstd, int32_t C, void* stream) {
  if (!sum || !sumsq || !gamma || !beta || !scale || !shift || !mean || !rstd || C <= 0 || M <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_bn_batch_fold_f32: bad arguments");
  const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  hipLaunchKernelGGL(bn_batch_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum, sumsq, gamma, beta,
                     (float)(1.0 / (double)M), unbias, eps, momentum, running_mean, running_var, scale, shift, mean, rstd, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bn_apply_bf16(const mg_bf16* z, const float* scale, const float* shift, const mg_bf16* res, int32_t relu,
                                mg_bf16* y, int64_t M, int32_t C, void* stream) {
  if (!z || !scale || !shift || !y || M <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_bn_apply_bf16: need C %% 8 == 0");
  if (!MG_ALIGNED16(z) || !MG_ALIGNED16(y) || !MG_ALIGNED16(res)) MG_FAIL(MG_ERR_ALIGN, "mg_bn_apply_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, z, scale, shift, res, relu, y, M * (C / 8), C / 8);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bn_bwd_dz_bf16(const mg_bf16* g, const mg_bf16* z, const float* mean, const float* rstd, const float* gamma,
                                 const float* dgamma, const float* dbeta, mg_bf16* dz, int64_t M, int32_t C, void* stream) {
  if (!g || !z || !mean || !rstd || !gamma || !dgamma || !dbeta || !dz || M <= 0 || C <= 0 || (C & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_bn_bwd_dz_bf16: need C %% 8 == 0");
  if (!MG_ALIGNED16(g) || !MG_ALIGNED16(z) || !MG_ALIGNED16(dz)) MG_FAIL(MG_ERR_ALIGN, "mg_bn_bwd_dz_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(bn_bwd_dz_kernel, dim3(grid_for(M * (C / 8))), dim3(256), 0, (hipStream_t)stream, g, z, mean, rstd, gamma, dgamma, dbeta,
                     (float)(1.0 / (double)M), dz, M * (C / 8), C / 8);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_im2col_t_bf16(const mg_bf16* x, mg_bf16* out, int64_t ldo, int32_t B, int32_t H, int32_t W, int32_t Cin, void* stream) {
  const int64_t M = (int64_t)B * H * W;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 7) || ldo < ((M + 7) & ~7) || (ldo & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_im2col_t_bf16: need Cin%%8==0, ldo>=round_up(M,8), ldo%%8==0");
  if (!x || !out || !MG_ALIGNED16(x) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_im2col_t_bf16: null/unaligned pointer");
  hipLaunchKernelGGL(im2col_t_kernel, dim3((unsigned)((M + 63) / 64), (Cin + 63) / 64, 9), dim3(256), 0, (hipStream_t)stream, x, out, ldo, B, H, W, Cin);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_sumsq_f32(const float* g, int64_t n, float* out, void* stream) {
  if (n <= 0 || !g || !out) MG_FAIL(MG_ERR_SHAPE, "mg_sumsq_f32: bad arguments");
  if (MG_ALIGNED16(g) && n >= 4096)
    hipLaunchKernelGGL((sumsq_kernel<float, true>), dim3(grid_for(n / 4) > 2048 ? 2048 : grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  else
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_sumsq_bf16(const mg_bf16* g, int64_t n, float* out, void* stream) {
  if (n <= 0 || !g || !out) MG_FAIL(MG_ERR_SHAPE, "mg_sumsq_bf16: bad arguments");
  if (MG_ALIGNED16(g) && n >= 4096)
    hipLaunchKernelGGL((sumsq_kernel<mg_bf16, true>), dim3(grid_for(n / 8) > 2048 ? 2048 : grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  else
    hipLaunchKernelGGL(sumsq_kernel<mg_bf16>, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_cast_f32_bf16(const float* src, mg_bf16* dst, int64_t n, void* stream) {
  if (n <= 0 || (n & 3) || !src || !dst) MG_FAIL(MG_ERR_SHAPE, "mg_cast_f32_bf16: n must be a positive multiple of 4");
  if (!MG_ALIGNED16(src) || (((uintptr_t)dst) & 7u)) MG_FAIL(MG_ERR_ALIGN, "mg_cast_f32_bf16: src 16-byte, dst 8-byte alignment required");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

namespace {
template <typename G>
int adamw_launch(float* p, float* m, float* v, const G* g, mg_bf16* p_bf16, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int32_t step, float max_norm, const float* norm_sq, float grad_scale,
                 void* stream, const char* who) {
  if (n <= 0 || step <= 0 || !p || !m || !v || !g) MG_FAIL(MG_ERR_SHAPE, "%s: bad arguments", who);
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  const bool vec = MG_ALIGNED16(p) && MG_ALIGNED16(m) && MG_ALIGNED16(v) && !((uintptr_t)g & (4 * sizeof(G) - 1)) && !((uintptr_t)p_bf16 & 7u) && n >= 1024;
  if (vec)
    hipLaunchKernelGGL((adamw_kernel<G, true>), dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, m, v, g, p_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, max_norm, norm_sq, grad_scale);
  else
    hipLaunchKernelGGL(adamw_kernel<G>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, m, v, g, p_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, max_norm, norm_sq, grad_scale);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
}  // namespace

extern "C" int mg_adamw_f32(float* p, float* m, float* v, const float* g, mg_bf16* p_bf16, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                            const float* norm_sq, float grad_scale, void* stream) {
  return adamw_launch<float>(p, m, v, g, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, norm_sq, grad_scale, stream, "mg_adamw_f32");
}

extern "C" int mg_adamw_gbf16_f32(float* p, float* m, float* v, const mg_bf16* g, mg_bf16* p_bf16, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                                  const float* norm_sq, float grad_scale, void* stream) {
  return adamw_launch<mg_bf16>(p, m, v, g, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, norm_sq, grad_scale, stream, "mg_adamw_gbf16_f32");
}

extern "C" int mg_mul_bf16(const mg_bf16* a, const mg_bf16* b, mg_bf16* out, int64_t n, void* stream) {
  if (n <= 0 || (n & 7) || !a || !b || !out) MG_FAIL(MG_ERR_SHAPE, "mg_mul_bf16: n must be a positive multiple of 8");
  if (!MG_ALIGNED16(a) || !MG_ALIGNED16(b) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_mul_bf16: 16-byte alignment required");
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 8);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// y = gelu_erf(x) (g == NULL) or y = g * gelu_erf'(x) over [rows, cols] bf16 with row strides; in place allowed (y == x or y == g)
extern "C" int mg_gelu_erf_bf16(const mg_bf16* x, int64_t ldx, const mg_bf16* g, int64_t ldg, mg_bf16* y, int64_t ldy, int32_t rows,
                                int32_t cols, void* stream) {
  if (rows <= 0 || cols <= 0 || (cols & 7) || !x || !y) MG_FAIL(MG_ERR_SHAPE, "mg_gelu_erf_bf16: cols must be a positive multiple of 8");
  if (!MG_ALIGNED16(x) || !MG_ALIGNED16(y) || !MG_ALIGNED16(g) || (ldx & 7) || (ldy & 7) || (g && (ldg & 7))) MG_FAIL(MG_ERR_ALIGN, "mg_gelu_erf_bf16: 16-byte aligned rows required");
  const int cv = cols / 8;
  if (g) hipLaunchKernelGGL(gelu_erf_kernel<true>, dim3(grid_for((int64_t)rows * cv)), dim3(256), 0, (hipStream_t)stream, x, ldx, g, ldg, y, ldy, rows, cv);
  else hipLaunchKernelGGL(gelu_erf_kernel<false>, dim3(grid_for((int64_t)rows * cv)), dim3(256), 0, (hipStream_t)stream, x, ldx, g, ldg, y, ldy, rows, cv);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_scale_rows_acc_f32(float* dst, const float* src, int64_t ld_src, const float* row_scale, int32_t rows,
                                     int32_t cols, void* stream) {
  if (rows <= 0 || cols <= 0 || !dst || !src || ld_src < cols) MG_FAIL(MG_ERR_SHAPE, "mg_scale_rows_acc_f32: bad arguments");
  hipLaunchKernelGGL(scale_rows_acc_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, dst, src, ld_src, row_scale, rows, cols);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// ---------------------------------------------------------------------------
// Conv weight [Cout][Cin][k][k] (k = 1 or 3, bf16, trainable: re-laid out every step) -> row-major GEMM operand.
//   mode 0 (forward / implicit-im2col order):  out[co][tap*Cin + ci]      = w[co][ci][ky][kx]
//   mode 1 (dgrad: dX = conv(dY, W') with the taps flipped, folded BN scale applied in fp32):
//                                              out[ci][tap*Cout + co]     = bf16( w[co][ci][k-1-ky][k-1-kx] * scale[co] )
// tap = ky*k + kx; rows zero padded to ldo.  One launch replaces the permute / flip / multiply / cast / pad chain of
// PyTorch copies (about 10 launches per conv and step on the 127 convs of the trunk).
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void conv_weight_relayout_kernel(const mg_bf16* __restrict__ w, const float* __restrict__ scale,
                                                                   mg_bf16* __restrict__ out, int64_t ldo, int Cout, int Cin, int k, int mode) {
  const int taps = k * k;
  const int rows = mode == 0 ? Cout : Cin, inner = mode == 0 ? Cin : Cout;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * ldo) return;
  const int r = (int)(idx / ldo), c = (int)(idx - (int64_t)r * ldo);
  mg_bf16 v = 0;
  if (c < taps * inner) {
    const int tap = c / inner, i = c - tap * inner;
    const int ky = tap / k, kx = tap - ky * k;
    if (mode == 0) v = w[(((int64_t)r * Cin + i) * k + ky) * k + kx];
    else v = f2bf(bf2f(w[(((int64_t)i * Cin + r) * k + (k - 1 - ky)) * k + (k - 1 - kx)]) * (scale ? scale[i] : 1.0f));
  }
  out[idx] = v;
}
}  // namespace

extern "C" int mg_conv_weight_relayout_bf16(const mg_bf16* w, const float* scale, mg_bf16* out, int64_t ldo, int32_t Cout,
                                            int32_t Cin, int32_t k, int32_t mode, void* stream) {
  if (!w || !out) MG_FAIL(MG_ERR_SHAPE, "mg_conv_weight_relayout_bf16: null pointer");
  if (Cout <= 0 || Cin <= 0 || (k != 1 && k != 3) || (mode != 0 && mode != 1)) MG_FAIL(MG_ERR_SHAPE, "mg_conv_weight_relayout_bf16: bad geometry");
  const int64_t need = (int64_t)k * k * (mode == 0 ? Cin : Cout);
  if (ldo < need) MG_FAIL(MG_ERR_SHAPE, "mg_conv_weight_relayout_bf16: ldo too small");
  const int64_t n = (int64_t)(mode == 0 ? Cout : Cin) * ldo;
  hipLaunchKernelGGL(conv_weight_relayout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale, out, ldo,
                     Cout, Cin, k, mode);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// All the weight re-layouts of one training step in ONE launch (the 127 convolutions of the CLIP trunk x {forward operand, dgrad
// operand}: 252 launches of ~7 us on 1-40 workgroups each before).  `jobs` is a device array sorted by first_block; a
// workgroup finds its job by bisection over the (wave-uniform) block index, then does what conv_weight_relayout_kernel does.
namespace {
__global__ __launch_bounds__(256) void conv_weight_relayout_batch_kernel(const mg_relayout_job* __restrict__ jobs, int njobs) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                      // last job with first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const mg_relayout_job j = jobs[lo];
  const int k = j.k, taps = k * k;
  const int rows = j.mode == 0 ? j.Cout : j.Cin, inner = j.mode == 0 ? j.Cin : j.Cout;
  const int64_t idx = (b - j.first_block) * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * j.ldo) return;
  const int r = (int)(idx / j.ldo), c = (int)(idx - (int64_t)r * j.ldo);
  mg_bf16 v = 0;
  if (c < taps * inner) {
    const int tap = c / inner, i = c - tap * inner;
    const int ky = tap / k, kx = tap - ky * k;
    if (j.mode == 0) v = j.w[(((int64_t)r * j.Cin + i) * k + ky) * k + kx];
    else v = f2bf(bf2f(j.w[(((int64_t)i * j.Cin + r) * k + (k - 1 - ky)) * k + (k - 1 - kx)]) * (j.scale ? j.scale[i] : 1.0f));
  }
  j.out[idx] = v;
}

__global__ __launch_bounds__(256) void bn_fold_batch_kernel(const mg_bn_fold_job* __restrict__ jobs, int njobs) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const mg_bn_fold_job j = jobs[lo];
  const int c = (int)(b - j.first_block) * 256 + threadIdx.x;
  if (c >= j.C) return;
  const float s = j.gamma[c] / sqrtf(j.var[c] + j.eps);      // the arithmetic of bn_fold_kernel
  j.scale[c] = s;
  j.shift[c] = j.beta[c] - j.mean[c] * s;
}
}  // namespace

extern "C" int mg_conv_weight_relayout_batch(const mg_relayout_job* jobs, int32_t njobs, int64_t total_blocks, void* stream) {
  if (!jobs || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffff) MG_FAIL(MG_ERR_SHAPE, "mg_conv_weight_relayout_batch: bad arguments");
  hipLaunchKernelGGL(conv_weight_relayout_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, njobs);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bn_fold_batch(const mg_bn_fold_job* jobs, int32_t njobs, int64_t total_blocks, void* stream) {
  if (!jobs || njobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffff) MG_FAIL(MG_ERR_SHAPE, "mg_bn_fold_batch: bad arguments");
  hipLaunchKernelGGL(bn_fold_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, njobs);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// frozen-statistics BatchNorm folded to a per-channel affine: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
namespace {
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = gamma[c] / sqrtf(var[c] + eps);
  scale[c] = s;
  shift[c] = beta[c] - mean[c] * s;
}
}  // namespace

extern "C" int mg_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                              float* scale, float* shift, int32_t C, void* stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift || C <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_bn_fold_f32: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, scale, shift, C);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
