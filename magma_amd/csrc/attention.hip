// attention.hip -- GPT-J attention pieces for gfx950: rotary + KV scatter,
// causal flash attention (head dim 256, fp32 online softmax) and the
// single-query decode attention.
//
// Layouts (ours to choose: the KV cache is opaque to reference
// magma/sampling.py:81-93, it is only handed back):
//   q      [B, H, S, 256]            rotated
//   kcache [B, H, Smax, 256]         rotated keys, row-major (decode + prefill)
//   vcache [B, H, Smax, 256]         values, row-major (decode: coalesced rows)
//   vt     [B, H, 256, vt_ld]        values transposed (prefill/training PV
//                                    operand: MFMA contracts over 8 consecutive
//                                    keys per lane, so V must be key-contiguous)
#include "common.h"
#include "attn_decode_device.h"

namespace {

constexpr int DH = 256;

// ---------------------------------------------------------------------------
// rotary + split.  grid (ceil(S/32), B*H), 256 threads.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotary_split_kernel(
    const mg_bf16* __restrict__ qkv, int B, int S, int H, int rot_dim,
    const float* __restrict__ sin_t, const float* __restrict__ cos_t, int pos0_host,
    const int* __restrict__ d_pos, mg_bf16* __restrict__ q_out, mg_bf16* __restrict__ kcache,
    mg_bf16* __restrict__ vcache, int Smax, mg_bf16* __restrict__ vt, int vt_ld) {
  __shared__ __attribute__((aligned(16))) mg_bf16 vtile[32 * DH];
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int s0 = blockIdx.x * 32;
  const int pos0 = d_pos ? *d_pos : pos0_host;
  const int dmodel = H * DH;
  const int half_rot = rot_dim >> 1;

#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31;
    const int s = s0 + row;
    const int d0 = c * 8;
    u32x4 vv = (u32x4){0u, 0u, 0u, 0u};
    if (s < S) {
      const mg_bf16* base = qkv + (int64_t)(b * S + s) * (3 * dmodel) + h * DH + d0;
      u32x4 qv = *(const u32x4*)base;
      u32x4 kv = *(const u32x4*)(base + dmodel);
      vv = *(const u32x4*)(base + 2 * dmodel);
      const int pos = pos0 + s;
      if (d0 < rot_dim) {
        // GPT-J interleaved pairs: (x[2i], x[2i+1]) rotated by pos*theta_i
        const float* sp = sin_t + (int64_t)pos * half_rot + (d0 >> 1);
        const float* cp = cos_t + (int64_t)pos * half_rot + (d0 >> 1);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          const float sn = sp[pi], cs = cp[pi];
          const float q0 = bflo(qv[pi]), q1 = bfhi(qv[pi]);
          const float k0 = bflo(kv[pi]), k1 = bfhi(kv[pi]);
          qv[pi] = pack2bf(q0 * cs - q1 * sn, q1 * cs + q0 * sn);
          kv[pi] = pack2bf(k0 * cs - k1 * sn, k1 * cs + k0 * sn);
        }
      }
      *(u32x4*)(q_out + ((int64_t)bh * S + s) * DH + d0) = qv;
      *(u32x4*)(kcache + ((int64_t)bh * Smax + pos) * DH + d0) = kv;
      *(u32x4*)(vcache + ((int64_t)bh * Smax + pos) * DH + d0) = vv;
    }
    if (vt) *(u32x4*)(vtile + row * DH + d0) = vv;   // zero rows beyond S
  }
  if (!vt) return;
  __syncthreads();
  // thread = one d; gather its 32 keys and write 64 contiguous bytes of V^T
  const int dd = tid;
  mg_bf16* dst = vt + ((int64_t)bh * DH + dd) * vt_ld + s0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t lo = vtile[(g * 8 + w * 2) * DH + dd];
      const uint32_t hi = vtile[(g * 8 + w * 2 + 1) * DH + dd];
      o[w] = lo | (hi << 16);
    }
    *(u32x4*)(dst + g * 8) = o;
  }
}

// ---------------------------------------------------------------------------
// flash attention forward, causal, dh = 256.
// grid (ceil(S/64), B*H), 256 threads = 4 waves x 16 query rows.
// Per KV tile of 32 keys and per wave:
//   S^T[key][q] = K . Q^T   (2 key-subtiles x 8 k-steps  = 16 MFMA)
//   O^T[d][q]  += V^T . P^T (16 d-subtiles x 1 k-step    = 16 MFMA)
// Both products are computed transposed so that every per-query quantity
// (running max, sum, rescale factor) is lane-local: lane&15 = query.
// The key permutation keymap(i,t) = (i>>2)*8 + t*4 + (i&3) makes the S^T
// accumulator registers of a lane exactly the 8 consecutive keys it must
// supply as the P^T operand of the PV product -- no cross-lane movement.
// ---------------------------------------------------------------------------
// conflict-free LDS images for the ds_read_b128 lane groups (searched offline): K rows
// unpadded with the 16-B chunk index XORed by f(row) = (row&3)|((row>>3)<<2); V^T rows 96 B.
constexpr int K_STRIDE = DH * 2;        // bytes per K row in LDS (512)
constexpr int VT_STRIDE = 32 * 2 + 32;  // bytes per V^T row in LDS (96)
MG_DEV int krow_swz(int row) { return (row & 3) | ((row >> 3) << 2); }
constexpr int FA_LDS = 32 * K_STRIDE + DH * VT_STRIDE;  // 16896 + 20480

__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ kcache,
    const mg_bf16* __restrict__ vt, mg_bf16* __restrict__ out, float* __restrict__ lse,
    int B, int H, int S, int Smax, int vt_ld) {
  __shared__ __attribute__((aligned(16))) char smem[FA_LDS];
  char* k_lds = smem;
  char* v_lds = smem + 32 * K_STRIDE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  // all query blocks of one (b,h) on ONE XCD: its K / V^T stream is re-read from that XCD's L2
  const int nblk = (S + 63) >> 6;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 64;   // longest blocks first
  const int qrow = qt0 + wave * 16 + li;       // this lane's query
  const int qrow_c = min(qrow, S - 1);
  const mg_bf16* kbase = kcache + (int64_t)bh * Smax * DH;
  const mg_bf16* vbase = vt + (int64_t)bh * DH * vt_ld;

  // Q fragments (B operand of S^T): Q[q][ks*32 + lq*8 .. +7]
  bf16x8 qf[8];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  }
  f32x4 o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m2 = -1e30f;   // running max in log2 domain
  float lsum = 0.f;    // this lane's partial row sum (its 8 keys per tile)
  const float sc2 = 0.0625f * 1.4426950408889634f;  // 1/sqrt(256) * log2(e)

  const int kv_end = min(S, qt0 + 64);          // causal: keys <= last query of the tile
  const int ntiles = (kv_end + 31) >> 5;

  // staging registers: 4 K chunks + 4 V^T chunks per thread
  u32x4 kreg[4], vreg[4];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ci = tid + it * 256;
      const int row = ci >> 5, c = ci & 31;                 // K: 32 rows x 32 chunks
      const int key = min(kv0 + row, S - 1);
      kreg[it] = *(const u32x4*)(kbase + (int64_t)key * DH + c * 8);
      const int dr = ci >> 2, vc = ci & 3;                  // V^T: 256 rows x 4 chunks
      vreg[it] = *(const u32x4*)(vbase + (int64_t)dr * vt_ld + kv0 + vc * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ci = tid + it * 256;
      *(u32x4*)(k_lds + (ci >> 5) * K_STRIDE + (((ci & 31) ^ krow_swz(ci >> 5)) << 4)) = kreg[it];
      *(u32x4*)(v_lds + (ci >> 2) * VT_STRIDE + (ci & 3) * 16) = vreg[it];
    }
  };

  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * 32;
    __syncthreads();          // all waves finished reading the previous tile
    store_tile();
    __syncthreads();
    if (t + 1 < ntiles) load_tile(kv0 + 32);   // in flight during the MFMAs

    // ---- S^T = K Q^T ----
    f32x4 st[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      st[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = (li >> 2) * 8 + tt * 4 + (li & 3);
      const int sw = krow_swz(krow);
      const char* kp = k_lds + krow * K_STRIDE;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kp + (((ks * 4 + lq) ^ sw) << 4));
        st[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[tt], 0, 0, 0);
      }
    }
    // lane holds keys kv0 + lq*8 + j, j = tt*4 + r, for query li
    float p[8];
    float tmax = -1e30f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = kv0 + lq * 8 + j;
      float v = st[j >> 2][j & 3] * sc2;
      if (key > qrow || key >= S) v = -1e30f;
      p[j] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m2, tmax);
    const float alpha = exp2f(m2 - mnew);
    m2 = mnew;
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = (p[j] <= -1e29f) ? 0.f : exp2f(p[j] - mnew);
      psum += p[j];
    }
    lsum = lsum * alpha + psum;
    u32x4 pw;
#pragma unroll
    for (int j = 0; j < 4; ++j) pw[j] = pack2bf(p[2 * j], p[2 * j + 1]);
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
    // ---- O^T = O^T * alpha + V^T P^T ----
    const char* vp = v_lds + li * VT_STRIDE + lq * 16;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      const bf16x8 vf = *(const bf16x8*)(vp + dt * 16 * VT_STRIDE);
      o[dt] *= alpha;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
    }
  }
  // row sum across the 4 key-slot lanes of this query
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (qrow < S) {
    const float inv = 1.0f / lsum;
    mg_bf16* op = out + (int64_t)(b * S + qrow) * (H * DH) + h * DH + lq * 4;
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) {
      u32x2 w;
      w[0] = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
      w[1] = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
      *(u32x2*)(op + dt * 16) = w;
    }
    if (lse && lq == 0) lse[(int64_t)bh * S + qrow] = (m2 + log2f(lsum)) * 0.6931471805599453f;
  }
}

template <bool FUSED>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnDecodeParams P) {
  __shared__ __attribute__((aligned(16))) char lds[ATTN_DEC_LDS];
  attn_decode_body<FUSED>(P, blockIdx.x, lds);
}

}  // namespace

extern "C" int mg_rotary_split_bf16(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                                    const float* sin_t, const float* cos_t, int32_t pos0_host,
                                    const int32_t* d_pos, mg_bf16* q_out, mg_bf16* kcache,
                                    mg_bf16* vcache, int32_t Smax, mg_bf16* vt, int32_t vt_ld,
                                    void* stream) {
  if (B <= 0 || S <= 0 || H <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: B,S,H must be positive");
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !q_out || !kcache || !vcache || (rot_dim && (!sin_t || !cos_t))) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: null pointer");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(q_out) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vcache) || !MG_ALIGNED16(vt))
    MG_FAIL(MG_ERR_ALIGN, "mg_rotary_split_bf16: pointers must be 16-byte aligned");
  if (vt) {
    if (d_pos || pos0_host != 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: V^T output requires pos0 == 0 (prefill)");
    if ((vt_ld & 7) || vt_ld < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: vt_ld must be a multiple of 8 and >= round_up(S,32)");
  }
  if (!d_pos && pos0_host + S > Smax) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: pos0+S exceeds Smax");
  dim3 grid((S + 31) / 32, B * H);
  hipLaunchKernelGGL(rotary_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, B, S, H, rot_dim, sin_t,
                     cos_t, pos0_host, d_pos, q_out, kcache, vcache, Smax, vt, vt_ld);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_prefill_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out,
                                    float* lse, int32_t B, int32_t H, int32_t S, int32_t Smax, int32_t vt_ld,
                                    void* stream) {
  if (B <= 0 || H <= 0 || S <= 0 || S > Smax) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: bad B/H/S/Smax");
  if ((vt_ld & 7) || vt_ld < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: vt_ld must be a multiple of 8 and >= round_up(S,32)");
  if (!q || !kcache || !vt || !out) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: null pointer");
  if (!MG_ALIGNED16(q) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vt) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_prefill_bf16: pointers must be 16-byte aligned");
  dim3 grid((unsigned)(((S + 63) / 64) * B * H));
  hipLaunchKernelGGL(attn_prefill_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, kcache, vt, out, lse, B, H, S,
                     Smax, vt_ld);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_decode_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vcache, mg_bf16* out,
                                   int32_t B, int32_t H, int32_t Smax, const int32_t* d_pos, void* stream) {
  if (B <= 0 || H <= 0 || Smax <= 0 || Smax > DEC_MAX_CTX) MG_FAIL(MG_ERR_SHAPE, "mg_attn_decode_bf16: need 0 < Smax <= %d", DEC_MAX_CTX);
  if (!q || !kcache || !vcache || !out || !d_pos) MG_FAIL(MG_ERR_SHAPE, "mg_attn_decode_bf16: null pointer");
  if (!MG_ALIGNED16(q) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vcache) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_decode_bf16: pointers must be 16-byte aligned");
  AttnDecodeParams P{q, (mg_bf16*)kcache, (mg_bf16*)vcache, out, H, Smax, d_pos, 0, nullptr, nullptr};
  hipLaunchKernelGGL(attn_decode_kernel<false>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, P);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_decode_fused_bf16(const mg_bf16* qkv, mg_bf16* kcache, mg_bf16* vcache, mg_bf16* out, int32_t B,
                                         int32_t H, int32_t Smax, const int32_t* d_pos, int32_t rot_dim,
                                         const float* sin_t, const float* cos_t, void* stream) {
  if (B <= 0 || H <= 0 || Smax <= 0 || Smax > DEC_MAX_CTX) MG_FAIL(MG_ERR_SHAPE, "mg_attn_decode_fused_bf16: need 0 < Smax <= %d", DEC_MAX_CTX);
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_decode_fused_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !kcache || !vcache || !out || !d_pos || (rot_dim && (!sin_t || !cos_t))) MG_FAIL(MG_ERR_SHAPE, "mg_attn_decode_fused_bf16: null pointer");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vcache) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_decode_fused_bf16: pointers must be 16-byte aligned");
  AttnDecodeParams P{qkv, kcache, vcache, out, H, Smax, d_pos, rot_dim, sin_t, cos_t};
  hipLaunchKernelGGL(attn_decode_kernel<true>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, P);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
