// attention_bwd.hip -- causal flash-attention backward, head dim 256 (gfx950).
//
// Two kernels that recompute P from (q, k, lse) -- no S x S tensor in HBM:
//
//   attn_bwd_dq_kernel    one workgroup per 64 queries (4 waves x 16), loops over
//                         KV tiles of 32.  Works in the transposed frame of the
//                         forward kernel (lane&15 = query):
//                           S^T  = K Q^T        dP^T = V dO^T
//                           dS^T = P^T o (dP^T - D) / 16
//                           dQ^T += K^T dS^T
//   attn_bwd_dkdv_kernel  one workgroup per 64 keys (4 waves x 16, lane&15 = key),
//                         loops over query tiles of 32 from the diagonal down:
//                           S  = Q K^T          dP = dO V^T
//                           dV^T += dO^T P      dK^T += Q^T dS
//                         (two instantiations: <false> = dV, <true> = dK)
//
// MFMA contracts over 8 consecutive elements per lane, so every operand is
// needed with its contraction index contiguous: K,V,Q,dO row-major [s][256] for
// the d-contractions and K^T,Q^T,dO^T [256][s] for the s-contractions (made by
// mg_head_transpose_bf16).  The same row permutation as in the forward kernel
// turns accumulator registers directly into the next product's operand.
// D[b,h,q] = sum_d dO*O comes from attn_bwd_prep_kernel.
#include "common.h"

namespace {

constexpr int DH = 256;
// LDS images chosen conflict-free for the ds_read_b128 lane groups of gfx950 (searched
// offline over the fragment access patterns below):
//   row tiles [32][256]: unpadded 512-B rows, 16-B chunk index XORed with
//                        f(row) = (row&3) | ((row>>3)<<2)
//   T tiles  [256][32]:  96-B row stride (6 x 16 B)
constexpr int ROW_STRIDE = DH * 2;       // 512 B
constexpr int T_STRIDE = 32 * 2 + 32;    // 96 B
MG_DEV int row_swz(int row) { return (row & 3) | ((row >> 3) << 2); }
constexpr int ROW_TILE = 32 * ROW_STRIDE;  // 16896
constexpr int T_TILE = DH * T_STRIDE;      // 20480

// D = rowsum(dO o O); one wave per (b, s, h) row of 256
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const mg_bf16* __restrict__ dO,
                                                            const mg_bf16* __restrict__ O, float* __restrict__ D,
                                                            int B, int H, int S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;   // over B*S*H, (b,s,h) order = memory order of [M, H*256]
  if (row >= (int64_t)B * S * H) return;
  const u32x2 a = *(const u32x2*)(dO + row * DH + lane * 4);
  const u32x2 o = *(const u32x2*)(O + row * DH + lane * 4);
  float s = bflo(a[0]) * bflo(o[0]) + bfhi(a[0]) * bfhi(o[0]) + bflo(a[1]) * bflo(o[1]) + bfhi(a[1]) * bfhi(o[1]);
  s = wave_sum(s);
  if (lane == 0) {
    const int h = (int)(row % H);
    const int64_t bs = row / H;
    const int sidx = (int)(bs % S), b = (int)(bs / S);
    D[((int64_t)b * H + h) * S + sidx] = s;
  }
}

// Staging is split into "issue the global loads" and "write the registers to LDS" so the
// loads of tile t+1 are in flight while tile t is being multiplied (both kernels run at one
// or two waves per SIMD: there is no other latency hiding).
MG_DEV void load_rows(u32x4 (&r)[4], const mg_bf16* base, int64_t row_stride_elems, int r0, int rmax, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int rr = min(r0 + (ci >> 5), rmax - 1);            // 32 rows x 32 chunks, rows clamped
    r[it] = *(const u32x4*)(base + (int64_t)rr * row_stride_elems + (ci & 31) * 8);
  }
}
MG_DEV void store_rows(char* lds, const u32x4 (&r)[4], int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5;
    *(u32x4*)(lds + row * ROW_STRIDE + (((ci & 31) ^ row_swz(row)) << 4)) = r[it];
  }
}
MG_DEV void load_cols(u32x4 (&r)[4], const mg_bf16* base_t, int ld, int c0, int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;                            // [256][32] slice at column c0
    r[it] = *(const u32x4*)(base_t + (int64_t)(ci >> 2) * ld + c0 + (ci & 3) * 8);
  }
}
MG_DEV void store_cols(char* lds, const u32x4 (&r)[4], int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    *(u32x4*)(lds + (ci >> 2) * T_STRIDE + (ci & 3) * 16) = r[it];
  }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ kt, const mg_bf16* __restrict__ dO, const float* __restrict__ lse,
    const float* __restrict__ Dv, mg_bf16* __restrict__ dq, int B, int H, int S, int ld_t) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ROW_TILE + T_TILE];
  char* k_lds = smem;
  char* v_lds = smem + ROW_TILE;
  char* kt_lds = smem + 2 * ROW_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int qt0 = blockIdx.x * 64;
  const int qrow = qt0 + wave * 16 + li, qrow_c = min(qrow, S - 1);
  const mg_bf16* kb = k + (int64_t)bh * S * DH;
  const mg_bf16* vb = v + (int64_t)bh * S * DH;
  const mg_bf16* ktb = kt + (int64_t)bh * DH * ld_t;
  const int dmodel = H * DH;

  bf16x8 qf[8], dof[8];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + lq * 8;
    const mg_bf16* dp = dO + (int64_t)(b * S + qrow_c) * dmodel + h * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 32); dof[ks] = *(const bf16x8*)(dp + ks * 32); }
  }
  const float L2E = 1.4426950408889634f;
  const float sc2 = 0.0625f * L2E;
  const float lse2 = lse[(int64_t)bh * S + qrow_c] * L2E;
  const float Dq = Dv[(int64_t)bh * S + qrow_c];
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int kv_end = min(S, qt0 + 64);
  const int ntiles = (kv_end + 31) >> 5;
  u32x4 rk[4], rv[4], rkt[4];
  load_rows(rk, kb, DH, 0, S, tid);
  load_rows(rv, vb, DH, 0, S, tid);
  load_cols(rkt, ktb, ld_t, 0, tid);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 32;
    __syncthreads();
    store_rows(k_lds, rk, tid);
    store_rows(v_lds, rv, tid);
    store_cols(kt_lds, rkt, tid);
    __syncthreads();
    if (t + 1 < ntiles) {
      load_rows(rk, kb, DH, kv0 + 32, S, tid);
      load_rows(rv, vb, DH, kv0 + 32, S, tid);
      load_cols(rkt, ktb, ld_t, kv0 + 32, tid);
    }
    f32x4 st[2], dp[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      st[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = (li >> 2) * 8 + tt * 4 + (li & 3);
      const int sw = row_swz(krow);
      const char* kp = k_lds + krow * ROW_STRIDE;
      const char* vp = v_lds + krow * ROW_STRIDE;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int off = ((ks * 4 + lq) ^ sw) << 4;
        st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kp + off), qf[ks], st[tt], 0, 0, 0);
        dp[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(vp + off), dof[ks], dp[tt], 0, 0, 0);
      }
    }
    float ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kv0 + lq * 8 + j;
      const float p = (key > qrow || key >= S) ? 0.f : exp2f(st[j >> 2][j & 3] * sc2 - lse2);
      ds[j] = p * (dp[j >> 2][j & 3] - Dq) * 0.0625f;
    }
    u32x4 dw;
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[j] = pack2bf(ds[2 * j], ds[2 * j + 1]);
    const bf16x8 dsf = __builtin_bit_cast(bf16x8, dw);
    const char* tp = kt_lds + li * T_STRIDE + lq * 16;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tp + dt * 16 * T_STRIDE), dsf, acc[dt], 0, 0, 0);
  }
  if (qrow < S) {
    mg_bf16* op = dq + ((int64_t)bh * S + qrow) * DH + lq * 4;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 w;
      w[0] = pack2bf(acc[dt][0], acc[dt][1]);
      w[1] = pack2bf(acc[dt][2], acc[dt][3]);
      *(u32x2*)(op + dt * 16) = w;
    }
  }
}

// ---------------------------------------------------------------------------
// DK = false: dV only (needs S -> P and dO^T);  DK = true: dK only (needs S, dP, dS and Q^T).
// One accumulator set (64 VGPRs) per instantiation instead of two keeps the kernel at two
// waves per SIMD / two workgroups per CU, which is what hides the LDS and HBM latency here;
// the price is recomputing S once more (80 instead of 64 MFMAs per key-tile x query-tile).
template <bool DK>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ k, const mg_bf16* __restrict__ v,
    const mg_bf16* __restrict__ qt, const mg_bf16* __restrict__ dO, const mg_bf16* __restrict__ dOt,
    const float* __restrict__ lse, const float* __restrict__ Dv, mg_bf16* __restrict__ dout,
    int B, int H, int S, int ld_t) {
  // LDS: Q rows (+ dO rows for dK) + one transposed tile (Q^T for dK, dO^T for dV) + lse/D
  __shared__ __attribute__((aligned(16))) char smem[(DK ? 2 : 1) * ROW_TILE + T_TILE + 256];
  char* q_lds = smem;
  char* do_lds = smem + ROW_TILE;                         // DK only
  char* t_lds = smem + (DK ? 2 : 1) * ROW_TILE;
  float* ls_lds = (float*)(t_lds + T_TILE);               // 32 lse2 + 32 D
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int k0 = blockIdx.x * 64;
  const int key = k0 + wave * 16 + li, key_c = min(key, S - 1);
  const int dmodel = H * DH;
  const mg_bf16* qb = q + (int64_t)bh * S * DH;
  const mg_bf16* tb = (DK ? qt : dOt) + (int64_t)bh * DH * ld_t;
  const mg_bf16* dob = dO + (int64_t)b * S * dmodel + h * DH;   // row stride dmodel

  bf16x8 kf[8], vf[DK ? 8 : 1];
  {
    const mg_bf16* kp = k + ((int64_t)bh * S + key_c) * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const bf16x8*)(kp + ks * 32);
    if (DK) {
      const mg_bf16* vp = v + ((int64_t)bh * S + key_c) * DH + lq * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) vf[ks] = *(const bf16x8*)(vp + ks * 32);
    }
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float L2E = 1.4426950408889634f;
  const float sc2 = 0.0625f * L2E;

  const int q_begin = k0 & ~31;               // first query tile that can see key k0
  const int q_tiles_end = (S + 31) >> 5;
  u32x4 rq[4], rdo[DK ? 4 : 1], rt[4];
  float r_ls = 0.f;
  auto load_all = [&](int q0) {
    load_rows(rq, qb, DH, q0, S, tid);
    if constexpr (DK) load_rows(rdo, dob, dmodel, q0, S, tid);
    load_cols(rt, tb, ld_t, q0, tid);
    if (tid < 64) {
      const int qq = min(q0 + (tid & 31), S - 1);
      r_ls = tid < 32 ? lse[(int64_t)bh * S + qq] * L2E : Dv[(int64_t)bh * S + qq];
    }
  };
  load_all(q_begin);
  for (int t = q_begin >> 5; t < q_tiles_end; ++t) {
    const int q0 = t * 32;
    __syncthreads();
    store_rows(q_lds, rq, tid);
    if constexpr (DK) store_rows(do_lds, rdo, tid);
    store_cols(t_lds, rt, tid);
    if (tid < 64) ls_lds[tid] = r_ls;
    __syncthreads();
    if (t + 1 < q_tiles_end) load_all(q0 + 32);
    f32x4 s[2], dp[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      s[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int qr = (li >> 2) * 8 + tt * 4 + (li & 3);   // row permutation: acc regs -> 8 consecutive queries
      const int sw = row_swz(qr);
      const char* qp = q_lds + qr * ROW_STRIDE;
      const char* dop = do_lds + qr * ROW_STRIDE;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int off = ((ks * 4 + lq) ^ sw) << 4;
        s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(qp + off), kf[ks], s[tt], 0, 0, 0);
        if constexpr (DK)
          dp[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dop + off), vf[ks], dp[tt], 0, 0, 0);
      }
    }
    // lane holds queries q0 + lq*8 + j (j = tt*4 + r) for its key
    float val[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ql = lq * 8 + j, qg = q0 + ql;
      const float p = (key > qg || qg >= S || key >= S) ? 0.f : exp2f(s[j >> 2][j & 3] * sc2 - ls_lds[ql]);
      val[j] = DK ? p * (dp[j >> 2][j & 3] - ls_lds[32 + ql]) * 0.0625f : p;
    }
    u32x4 pw;
#pragma unroll
    for (int j = 0; j < 4; ++j) pw[j] = pack2bf(val[2 * j], val[2 * j + 1]);
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
    const char* tp = t_lds + li * T_STRIDE + lq * 16;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt)
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(tp + dt * 16 * T_STRIDE), pf, acc[dt], 0, 0, 0);
  }
  if (key < S) {
    mg_bf16* op = dout + ((int64_t)bh * S + key) * DH + lq * 4;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 w;
      w[0] = pack2bf(acc[dt][0], acc[dt][1]); w[1] = pack2bf(acc[dt][2], acc[dt][3]);
      *(u32x2*)(op + dt * 16) = w;
    }
  }
}

}  // namespace

// q,k,v [B,H,S,256]; kt,qt,dOt [B,H,256,ld_t] (ld_t >= round_up(S,32), zero padded);
// dO, O [B*S, H*256]; lse [B,H,S]; D [B,H,S] workspace; dq,dk,dv [B,H,S,256]
extern "C" int mg_attn_bwd_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                                const mg_bf16* kt, const mg_bf16* dO, const mg_bf16* dOt, const mg_bf16* O,
                                const float* lse, float* D, mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, int32_t B,
                                int32_t H, int32_t S, int32_t ld_t, void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || (ld_t & 7) || ld_t < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_bf16: ld_t must be a multiple of 8 and >= round_up(S,32)");
  const void* ptrs[] = {q, k, v, qt, kt, dO, dOt, O, lse, D, dq, dk, dv};
  for (const void* p : ptrs) {
    if (!p) MG_FAIL(MG_ERR_SHAPE, "mg_attn_bwd_bf16: null pointer");
    if (!MG_ALIGNED16(p)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_bwd_bf16: pointers must be 16-byte aligned");
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * S * H;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dO, O, D, B, H, S);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((S + 63) / 64, B * H), dim3(256), 0, s, q, k, v, kt, dO, lse, D, dq, B, H, S, ld_t);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<false>, dim3((S + 63) / 64, B * H), dim3(256), 0, s, q, k, v, qt, dO, dOt, lse, D, dv, B, H, S, ld_t);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<true>, dim3((S + 63) / 64, B * H), dim3(256), 0, s, q, k, v, qt, dO, dOt, lse, D, dk, B, H, S, ld_t);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
