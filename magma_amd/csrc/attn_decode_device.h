// attn_decode_device.h -- body of the single-query decode attention (rotary + KV append + attention),
// shared by attention.hip (stand-alone launch) and decode_fused.hip (co-launched with a GEMV).
#pragma once
#include "common.h"
#include "gemm_device.h"

// ---------------------------------------------------------------------------
// decode attention: one query per (b,h); ctx = *d_pos + 1 keys.  grid B*H, 256 thr.
//   FUSED = false: q [B,H,256] already rotated, K/V already appended.
//   FUSED = true : reads the fused qkv row [B, 3*H*256] of the new token, applies
//                  the rotary to q and k, appends k,v to the cache at *d_pos and
//                  attends over [0, *d_pos] -- one launch instead of two.
// Scores: S^T[key][.] = K . q^T on the MFMA (16 keys per wave step, all 8 K
// fragment loads of a step in flight at once, no cross-lane reductions); the 16
// "query columns" of the B operand all carry the same q.  PV: VALU, lane = 4
// dims, 8 keys in flight per wave step (V rows are coalesced 512-B reads).
// ---------------------------------------------------------------------------
constexpr int DEC_MAX_CTX = 4096;

constexpr int ATTN_DEC_LDS = DEC_MAX_CTX * 4 + 4 * 256 * 4 + 8 * 4 + 256 * 2 + 32;

struct AttnDecodeParams {
  const mg_bf16* qin; mg_bf16* kcache; mg_bf16* vcache; mg_bf16* out;
  int H, Smax; const int* d_pos; int rot_dim; const float* sin_t; const float* cos_t;
  int64_t ld_out = 0;      // row stride of `out` in elements (0 = H * 256)
};

// Device body: `bh` = (batch, head) index, `lds` = ATTN_DEC_LDS bytes of scratch (16-byte aligned).
// COH (persistent decode step): the fused qkv row comes from other workgroups of the same launch and the context row
// goes to others -- both through the coherent accessors of gemm_device.h.
template <bool FUSED, bool COH = false>
MG_DEV void attn_decode_body(const AttnDecodeParams& P, int bh, char* lds) {
  constexpr int DH = 256;
  const mg_bf16* __restrict__ qin = P.qin;
  mg_bf16* __restrict__ kcache = P.kcache;
  mg_bf16* __restrict__ vcache = P.vcache;
  mg_bf16* __restrict__ out = P.out;
  const int H = P.H, Smax = P.Smax, rot_dim = P.rot_dim;
  const int* __restrict__ d_pos = P.d_pos;
  const float* __restrict__ sin_t = P.sin_t;
  const float* __restrict__ cos_t = P.cos_t;
  mg_bf16* qs = (mg_bf16*)lds;                                   // 512 B (first: 16-byte aligned)
  float* sc = (float*)(lds + 512);
  float* red = sc + DEC_MAX_CTX;
  float* wred = red + 4 * DH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int b = bh / H, h = bh - b * H;
  const int64_t out_off = P.ld_out ? (int64_t)b * P.ld_out + (int64_t)h * DH : (int64_t)bh * DH;
  const int pos = *d_pos;
  const int ctx = min(pos + 1, min(Smax, DEC_MAX_CTX));
  mg_bf16* kb = kcache + (int64_t)bh * Smax * DH;
  mg_bf16* vb = vcache + (int64_t)bh * Smax * DH;
  if (FUSED) {
    // threads 0..31: q chunks, 32..63: k chunks, 64..95: v chunks (8 dims each)
    const int dmodel = H * DH;
    if (tid < 96) {
      const int which = tid >> 5, c = tid & 31, d0 = c * 8;
      u32x4 v;
      if constexpr (COH) v = ld16_coh(qin + (int64_t)b * 3 * dmodel + which * dmodel + h * DH + d0);
      else v = *(const u32x4*)(qin + (int64_t)b * 3 * dmodel + which * dmodel + h * DH + d0);
      if (which < 2 && d0 < rot_dim) {
        const int half_rot = rot_dim >> 1;
        const float* sp = sin_t + (int64_t)pos * half_rot + (d0 >> 1);
        const float* cp = cos_t + (int64_t)pos * half_rot + (d0 >> 1);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          const float sn = sp[pi], cs = cp[pi], x0 = bflo(v[pi]), x1 = bfhi(v[pi]);
          v[pi] = pack2bf(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
        }
      }
      if (which == 0) *(u32x4*)(qs + d0) = v;
      else if (which == 1) *(u32x4*)(kb + (int64_t)pos * DH + d0) = v;
      else *(u32x4*)(vb + (int64_t)pos * DH + d0) = v;
    }
    __threadfence_block();
  } else {
    if (tid < 32) *(u32x4*)(qs + tid * 8) = *(const u32x4*)(qin + (int64_t)bh * DH + tid * 8);
  }
  __syncthreads();
  // ---- phase 1: scores on the MFMA, 16 keys per wave step ----
  bf16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qs + ks * 32 + lq * 8);
  for (int t0 = wave * 16; t0 < ctx; t0 += 64) {
    const int key = min(t0 + li, ctx - 1);
    const mg_bf16* kp = kb + (int64_t)key * DH + lq * 8;
    bf16x8 kf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const bf16x8*)(kp + ks * 32);
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[ks], s, 0, 0, 0);
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + lq * 4 + r;
        if (t < ctx) sc[t] = s[r] * 0.0625f;
      }
    }
  }
  __syncthreads();
  // ---- phase 2: softmax statistics ----
  float mx = -1e30f;
  for (int t = tid; t < ctx; t += 256) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) wred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  float sm = 0.f;
  for (int t = tid; t < ctx; t += 256) {
    const float e = __expf(sc[t] - mx);
    sc[t] = e;
    sm += e;
  }
  sm = wave_sum(sm);
  if (lane == 0) wred[4 + wave] = sm;
  __syncthreads();
  const float inv = 1.0f / (wred[4] + wred[5] + wred[6] + wred[7]);
  // ---- phase 3: o = sum_t p[t] V[t]; wave w takes keys == w mod 4, 8 keys in flight ----
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t0 = wave; t0 < ctx; t0 += 32) {
    u32x2 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = min(t0 + u * 4, ctx - 1);
      w[u] = *(const u32x2*)(vb + (int64_t)t * DH + lane * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * 4;
      const float pt = t < ctx ? sc[t] : 0.f;
      acc[0] += pt * bflo(w[u][0]); acc[1] += pt * bfhi(w[u][0]);
      acc[2] += pt * bflo(w[u][1]); acc[3] += pt * bfhi(w[u][1]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * DH + lane * 4 + r] = acc[r];
  __syncthreads();
  const float v = (red[tid] + red[DH + tid] + red[2 * DH + tid] + red[3 * DH + tid]) * inv;
  if constexpr (COH) {      // four dims per lane -> one 8-byte coherent store (lanes 0..63 of wave 0)
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    if (tid < 64) {
      u32x2 w; w[0] = pack2bf(red[tid * 4], red[tid * 4 + 1]); w[1] = pack2bf(red[tid * 4 + 2], red[tid * 4 + 3]);
      st8_coh(out + out_off + tid * 4, w);
    }
  } else {
    out[out_off + tid] = f2bf(v);
  }
}

