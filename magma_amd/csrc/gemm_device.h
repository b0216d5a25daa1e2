// gemm_device.h -- device code shared by the GEMM translation units: the fused epilogue and the
// body of the decode weight-streaming GEMV (so that it can be co-launched with other workgroup kinds).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// shared epilogue: 4 consecutive n of row m.  Split in two so that row-walking callers load the
// per-column vectors (BatchNorm scale, bias) once.
// ---------------------------------------------------------------------------
struct EpiCols { float sc[4], bi[4]; };

MG_DEV void epilogue_cols(const mg_epilogue& ep, int n, int N, EpiCols& c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { c.sc[r] = 1.f; c.bi[r] = 0.f; }
  if (n + 3 < N) {
    if (ep.scale) { const float4 t = *(const float4*)(ep.scale + n); c.sc[0] = t.x; c.sc[1] = t.y; c.sc[2] = t.z; c.sc[3] = t.w; }
    if (ep.bias)  { const float4 t = *(const float4*)(ep.bias + n);  c.bi[0] = t.x; c.bi[1] = t.y; c.bi[2] = t.z; c.bi[3] = t.w; }
  } else {
    for (int r = 0; r < 4; ++r) if (n + r < N) {
      if (ep.scale) c.sc[r] = ep.scale[n + r];
      if (ep.bias) c.bi[r] = ep.bias[n + r];
    }
  }
}

MG_DEV void epilogue_apply4(const mg_epilogue& ep, const EpiCols& c, int m, int n, f32x4 v, int N) {
  const bool full = (n + 3 < N);
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = v[r] * c.sc[r] + c.bi[r];
  if (ep.C2) {  // pre-activation copy for the backward pass
    mg_bf16* cp = ep.C2 + (int64_t)m * ep.ldc2 + n;
    if (full) { u32x2 w; w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]); *(u32x2*)cp = w; }
    else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = apply_act(o[r], ep.act);
  float ax[4] = {1.f, 1.f, 1.f, 1.f};
  if (ep.aux_mode != MG_AUX_NONE) {
    const mg_bf16* ap = ep.aux + (int64_t)m * ep.ldaux + n;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) { const u32x2 w = *(const u32x2*)ap; a[0] = bflo(w[0]); a[1] = bfhi(w[0]); a[2] = bflo(w[1]); a[3] = bfhi(w[1]); }
    else for (int r = 0; r < 4; ++r) if (n + r < N) a[r] = bf2f(ap[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      ax[r] = ep.aux_mode == MG_AUX_RELU_GATE ? (a[r] > 0.f ? 1.f : 0.f)
            : ep.aux_mode == MG_AUX_GELU_GRAD ? gelu_new_grad_f(a[r]) : a[r];
    if (!ep.aux_after) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] *= ax[r];
    }
  }
  const mg_bf16* rs[3] = {ep.res0, ep.res1, ep.res2};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (rs[t]) {
      const mg_bf16* rp = rs[t] + (int64_t)m * ep.ldr + n;
      if (full) {
        const u32x2 w = *(const u32x2*)rp;
        o[0] += bflo(w[0]); o[1] += bfhi(w[0]); o[2] += bflo(w[1]); o[3] += bfhi(w[1]);
      } else {
        for (int r = 0; r < 4; ++r) if (n + r < N) o[r] += bf2f(rp[r]);
      }
    }
  }
  if (ep.aux_mode != MG_AUX_NONE && ep.aux_after) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= ax[r];
  }
  if (ep.act_after == MG_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
  }
  if (ep.out_f32) {
    float* cp = (float*)ep.C + (int64_t)m * ep.ldc + n;
    if (full) *(float4*)cp = make_float4(o[0], o[1], o[2], o[3]);
    else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = o[r];
  } else {
    mg_bf16* cp = (mg_bf16*)ep.C + (int64_t)m * ep.ldc + n;
    if (full) { u32x2 w; w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]); *(u32x2*)cp = w; }
    else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
  }
}

MG_DEV void epilogue_store4(const mg_epilogue& ep, int m, int n, f32x4 v, int N) {
  if (n >= N) return;
  EpiCols c;
  epilogue_cols(ep, n, N, c);
  epilogue_apply4(ep, c, m, n, v, N);
}

// Tile epilogue through LDS.  The MFMA accumulators of a workgroup tile were parked in LDS as fp32
// rows of NCOLS columns (row stride ROWB bytes; ROWB % 128 == 16 keeps the 8-lane groups of the
// fragment-shaped ds_write_b128 on distinct bank slots).  Here every wave walks whole rows: NCOLS/4
// lanes cover one row with 4 consecutive columns each, so residual / aux reads and the output stores
// are full contiguous lines, and the code is one small rolled loop instead of one inlined epilogue
// per accumulator (which made the GEMM kernels > 20k instructions, mostly instruction-cache misses).
// Tile row r is global row  m_base + (r >> 6) * hi_stride + (r & 63).
template <int NCOLS, int ROWB>
MG_DEV void epilogue_rows(const mg_epilogue& ep, const char* lds, int rows, int nwaves, int wave, int lane,
                          int m_base, int hi_stride, int n0, int M, int N) {
  constexpr int LPR = NCOLS / 4, RPI = 64 / LPR;
  const int cl = lane % LPR;
  const int n = n0 + cl * 4;
  if (n >= N) return;
  EpiCols c;
  epilogue_cols(ep, n, N, c);
#pragma unroll 2
  for (int r = wave * RPI + lane / LPR; r < rows; r += nwaves * RPI) {
    const int m = m_base + (r >> 6) * hi_stride + (r & 63);
    if (m < M) epilogue_apply4(ep, c, m, n, *(const f32x4*)(lds + r * ROWB + cl * 16), N);
  }
}


// ---------------------------------------------------------------------------
// skinny (decode) kernel
// ---------------------------------------------------------------------------
struct SkinnyParams {
  const mg_bf16* X; int64_t ldx;
  const mg_bf16* W;
  int M, N, ntiles, ksteps;
  // LayerNorm folded into the GEMV (decode): W' = W*gamma, bias' = b + W.beta are baked
  // into the operands; the kernel gets the row statistics from the x fragments it already
  // streams and applies  y = rstd*(acc - mean*colsum[n]) + bias'[n]  in the epilogue.
  const float* ln_colsum; float ln_inv_d, ln_eps;
  // two output segments (fused qkv | fc_in): columns >= split_n go to ep_b
  int split_n;
  mg_epilogue ep;
  mg_epilogue ep_b;
};

// Device body: `block` is the workgroup's index inside THIS problem's grid, `lds` a caller-provided
// scratch of skinny_lds_bytes<WAVES,NT>() bytes -- so several problems (and other workgroup kinds,
// see decode_fused.hip) can share one launch.
template <int WAVES, int NT>
constexpr int skinny_lds_bytes() { return WAVES * NT * 256 * 4 + WAVES * 16 * 2 * 4; }

template <int WAVES, int KC, int NT>
MG_DEV void skinny_body(const SkinnyParams& p, int block, char* lds) {
  float* red = (float*)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_wave = p.ksteps / WAVES;
  const int ks0 = wave * per_wave;
  const int nt0 = block * NT;
  const int li = lane & 15, lq = lane >> 4;
  const bool xok = li < p.M;
  const mg_bf16* xrow = p.X + (int64_t)(xok ? li : 0) * p.ldx + lq * 8;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float xs = 0.f, xss = 0.f;   // row statistics of x (LayerNorm fold)

  for (int kc = 0; kc < per_wave; kc += KC) {
    // Issue the whole chunk's loads before the first MFMA (GEMV recipe: loads
    // straight to VGPRs, deep queue, late wait): weights first (HBM, non-temporal
    // -- each byte is read exactly once per step), then the x fragments (L2 hits).
    u32x4 wf[NT][KC];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt = min(nt0 + t, p.ntiles - 1);
      const u32x4* wp = (const u32x4*)p.W + ((int64_t)nt * p.ksteps + ks0 + kc) * 64 + lane;
#pragma unroll
      for (int i = 0; i < KC; ++i) wf[t][i] = __builtin_nontemporal_load(wp + i * 64);
    }
    bf16x8 xf[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      u32x4 raw = *(const u32x4*)(xrow + (int64_t)(ks0 + kc + i) * 32);
      if (!xok) raw = (u32x4){0u, 0u, 0u, 0u};
      xf[i] = __builtin_bit_cast(bf16x8, raw);
    }
    if (p.ln_colsum) {
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        const u32x4 raw = __builtin_bit_cast(u32x4, xf[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = bflo(raw[j]), b = bfhi(raw[j]);
          xs += a + b;
          xss += a * a + b * b;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < KC; ++i)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[t][i]), xf[i], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // cross-wave (split-K) reduction through LDS, then epilogue by wave t
  float (*rstat)[16][2] = (float (*)[16][2])(lds + WAVES * NT * 256 * 4);
#pragma unroll
  for (int t = 0; t < NT; ++t) *(f32x4*)(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
  if (p.ln_colsum) {   // lanes li, li+16, li+32, li+48 hold the same row: fold the 4 k-slots
    xs += __shfl_xor(xs, 16, 64); xs += __shfl_xor(xs, 32, 64);
    xss += __shfl_xor(xss, 16, 64); xss += __shfl_xor(xss, 32, 64);
    if (lq == 0) { rstat[wave][li][0] = xs; rstat[wave][li][1] = xss; }
  }
  __syncthreads();
  float mean = 0.f, rstd = 1.f;
  if (p.ln_colsum) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { a += rstat[w][li][0]; b += rstat[w][li][1]; }
    mean = a * p.ln_inv_d;
    rstd = rsqrtf(fmaxf(b * p.ln_inv_d - mean * mean, 0.f) + p.ln_eps);
  }
  for (int t = wave; t < NT; t += WAVES) {
    if (nt0 + t >= p.ntiles) break;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += *(const f32x4*)(red + ((w * NT + t) * 64 + lane) * 4);
    if (!xok) continue;
    const int n = (nt0 + t) * 16 + lq * 4;
    if (p.ln_colsum && n < p.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = rstd * (s[r] - mean * (n + r < p.N ? p.ln_colsum[n + r] : 0.f));
    }
    if (p.split_n > 0 && n >= p.split_n) epilogue_store4(p.ep_b, li, n - p.split_n, s, p.N - p.split_n);
    else epilogue_store4(p.ep, li, n, s, p.split_n > 0 ? p.split_n : p.N);
  }
}

