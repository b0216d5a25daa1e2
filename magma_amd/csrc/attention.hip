// attention.hip -- GPT-J attention pieces for gfx950: rotary + KV scatter,
// causal flash attention (head dim 256, fp32 online softmax) and the
// single-query decode attention.
//
// Layouts (ours to choose: the KV cache is opaque to reference
// magma/sampling.py:81-93, it is only handed back):
//   q      [B, H, S, 256]            rotated
//   kcache [B, H, Smax, 256]         rotated keys, row-major (decode + prefill)
//   vcache [B, H, Smax, 256]         values, row-major (decode: coalesced rows)
//   vt     [B, H, vt_ld/32, 256, 32] values transposed, in 32-key tiles (prefill/training PV
//                                    operand: MFMA contracts over 8 consecutive
//                                    keys per lane, so V must be key-contiguous)
#include "common.h"
#include <stdlib.h>
#include "attn_tile_device.h"
#include "attn_decode_device.h"
#include "attn32_device.h"

namespace {

// ---------------------------------------------------------------------------
// rotary + split.  grid (ceil(S/32), B*H), 256 threads.
// ---------------------------------------------------------------------------
// QKT: also write the rotated q and k transposed (qt, kt, same column-tiled layout as vt) -- the s-contraction
// operands of the backward kernels, which otherwise cost two more passes (mg_head_transpose_bf16) over q and k.
template <bool QKT>
__global__ __launch_bounds__(256) void rotary_split_kernel(
    const mg_bf16* __restrict__ qkv, int64_t ld_qkv, int B, int S, int H, int rot_dim,
    const float* __restrict__ sin_t, const float* __restrict__ cos_t, int pos0_host,
    const int* __restrict__ d_pos, mg_bf16* __restrict__ q_out, mg_bf16* __restrict__ kcache,
    mg_bf16* __restrict__ vcache, int Smax, mg_bf16* __restrict__ vt, int vt_ld, mg_bf16* __restrict__ qt,
    mg_bf16* __restrict__ kt) {
  __shared__ __attribute__((aligned(16))) mg_bf16 vtile[(QKT ? 3 : 1) * 32 * DH];
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
    u32x4 vv = (u32x4){0u, 0u, 0u, 0u}, qz = vv, kz = vv;
    if (s < S) {
      const mg_bf16* base = qkv + (int64_t)(b * S + s) * ld_qkv + h * DH + d0;
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
      qz = qv; kz = kv;
    }
    if (vt) *(u32x4*)(vtile + row * DH + d0) = vv;   // zero rows beyond S
    if (QKT) {
      *(u32x4*)(vtile + 32 * DH + row * DH + d0) = qz;
      *(u32x4*)(vtile + 64 * DH + row * DH + d0) = kz;
    }
  }
  if (!vt) return;
  __syncthreads();
  // thread = one d; gather its 32 positions and write 64 contiguous bytes of the transposed tile
  const int dd = tid;
  const int64_t toff = (((int64_t)bh * (vt_ld >> 5) + blockIdx.x) * DH + dd) * 32;   // column-tiled: [b,h][tile][256][32]
#pragma unroll
  for (int which = 0; which < (QKT ? 3 : 1); ++which) {
    mg_bf16* dst = (which == 0 ? vt : which == 1 ? qt : kt) + toff;
    const mg_bf16* tile = vtile + which * 32 * DH;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t lo = tile[(g * 8 + w * 2) * DH + dd];
        const uint32_t hi = tile[(g * 8 + w * 2 + 1) * DH + dd];
        o[w] = lo | (hi << 16);
      }
      *(u32x4*)(dst + g * 8) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// rotary + split with OCP MX e4m3 copies for the fp8 attention forward (BASELINE config[4]: "fp8 MFMA path for GPT-J attention";
// attention_fwd32_fp8.hip).  Same tile as rotary_split_kernel<true>; besides the rotated bf16 q / k / v [B,H,S,256] and q^T / k^T
// (the backward stays bf16) it writes what v_mfma_scale_f32_32x32x64_f8f6f4 multiplies:
//   q8, k8 [B,H,S,256]   e4m3 elements, ONE E8M0 scale per token (eq, ek [B,H,Sp] bytes, Sp = 64-key tiles + 256): the power of
//                        two 2^e >= amax / 448 (no element saturates).  Every 32-element block of a row carries the same scale, so
//                        nothing depends on which operand bytes the hardware calls a block.
//   v8t [B,H,ceil(S/64),256,64]  e4m3 V^T in 64-key tiles with OCP MX scales along the KEYS: one E8M0 per (d, 32 keys), sv8
//                        [B,H,tile,2,32,8] bytes = [key block][d % 32][d / 32] (a lane's eight scale bytes are one 8-byte read).
//                        Inside a tile the keys are stored in the order the S^T accumulators of a lane hold them, so that a lane's
//                        32 P values ARE its 32 operand bytes of the PV product: byte 32 hi + 16 b + r of row d <-> key
//                        32 b + (r & 3) + 8 (r >> 2) + 4 hi.  Measured (tools/probes/mx32_probe.hip): the instruction's block b =
//                        bytes 16 b .. 16 b + 15 of both half-wave lanes of a row = key block b here, its scale comes from lane
//                        row + 32 b.
// Blocks past the sequence (the second half of a ragged last tile) write zero elements / unit scales.
// ---------------------------------------------------------------------------
MG_DEV u32x2 quant8_e4m3(const float (&x)[8], float inv) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * inv, x[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * inv, x[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * inv, x[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * inv, x[7] * inv, hi, true);
  return (u32x2){(uint32_t)lo, (uint32_t)hi};
}
// smallest e with 2^e >= amax / 448 (amax > 0), as the biased E8M0 byte; 127 (2^0) for an all-zero block
MG_DEV int e8m0_cover(float amax) {
  if (!(amax > 0.f)) return 127;
  const float t = amax * (1.0f / 448.0f);
  int e = (int)((__float_as_uint(t) >> 23) & 0xff) - 127;              // floor(log2 t) for a normal t
  if (__uint_as_float((uint32_t)(e + 127) << 23) < t) ++e;
  return min(max(e + 127, 1), 254);
}
MG_DEV float e8m0_inv(int byte) { return __uint_as_float((uint32_t)(254 - byte) << 23); }    // 2^-(byte - 127)
MG_DEV float row_amax32(const float (&x)[8]) {       // maximum |x| over the 32 lanes (8 values each) that hold one 256-wide row
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a = fmaxf(a, fabsf(x[i]));
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 32));
  return a;
}
__global__ __launch_bounds__(256) void rotary_split_fp8_kernel(
    const mg_bf16* __restrict__ qkv, int B, int S, int H, int rot_dim, const float* __restrict__ sin_t, const float* __restrict__ cos_t,
    mg_bf16* __restrict__ q_out, mg_bf16* __restrict__ k_out, mg_bf16* __restrict__ v_out, mg_bf16* __restrict__ qt, mg_bf16* __restrict__ kt,
    int ld_t, uint8_t* __restrict__ q8, uint8_t* __restrict__ k8, uint8_t* __restrict__ v8t, uint8_t* __restrict__ eq, uint8_t* __restrict__ ek,
    uint8_t* __restrict__ sv8, int Sp, int inplace) {
  __shared__ __attribute__((aligned(16))) mg_bf16 tile[3 * 32 * DH];        // rotated q, k rows (transposes) and v rows
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int s0 = blockIdx.x * 32;
  const int dmodel = H * DH, half_rot = rot_dim >> 1;
  const int nt64 = (S + 63) >> 6;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int ci = tid + it * 256;
    const int row = ci >> 5, c = ci & 31;
    const int s = s0 + row, d0 = c * 8;
    u32x4 qz = (u32x4){0u, 0u, 0u, 0u}, kz = qz, vv = qz;
    float qf[8], kf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { qf[i] = 0.f; kf[i] = 0.f; }
    if (s < S) {
      const mg_bf16* base = qkv + (int64_t)(b * S + s) * (3 * dmodel) + h * DH + d0;
      u32x4 qv = *(const u32x4*)base;
      u32x4 kv = *(const u32x4*)(base + dmodel);
      vv = *(const u32x4*)(base + 2 * dmodel);
      if (d0 < rot_dim) {
        const float* sp = sin_t + (int64_t)s * half_rot + (d0 >> 1);
        const float* cp = cos_t + (int64_t)s * half_rot + (d0 >> 1);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
          const float sn = sp[pi], cs = cp[pi];
          const float q0 = bflo(qv[pi]), q1 = bfhi(qv[pi]);
          const float k0 = bflo(kv[pi]), k1 = bfhi(kv[pi]);
          qv[pi] = pack2bf(q0 * cs - q1 * sn, q1 * cs + q0 * sn);
          kv[pi] = pack2bf(k0 * cs - k1 * sn, k1 * cs + k0 * sn);
        }
      }
      if (inplace && d0 < rot_dim) {      // the rotated q / k sections back into the fused activation (each lane rewrites the 16 bytes it read):
        *(u32x4*)const_cast<mg_bf16*>(base) = qv;                 // the bf16 backward reads q, k, v as rows of that buffer
        *(u32x4*)const_cast<mg_bf16*>(base + dmodel) = kv;
      }
      if (q_out) {                        // (NULL: forward only -- nobody reads the bf16 copies)
        *(u32x4*)(q_out + ((int64_t)bh * S + s) * DH + d0) = qv;
        *(u32x4*)(k_out + ((int64_t)bh * S + s) * DH + d0) = kv;
        *(u32x4*)(v_out + ((int64_t)bh * S + s) * DH + d0) = vv;
      }
      qz = qv; kz = kv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {       // the e4m3 copies are taken from the bf16-ROUNDED rotated values (what the bf16 path multiplies)
        qf[2 * i] = bflo(qv[i]); qf[2 * i + 1] = bfhi(qv[i]);
        kf[2 * i] = bflo(kv[i]); kf[2 * i + 1] = bfhi(kv[i]);
      }
    }
    // per-token power-of-two scales: the 32 lanes of a row agree on them (rows past S: all zero, 2^0)
    const int bq = e8m0_cover(row_amax32(qf)), bk = e8m0_cover(row_amax32(kf));
    if (s < S) {
      *(u32x2*)(q8 + ((int64_t)bh * S + s) * DH + d0) = quant8_e4m3(qf, e8m0_inv(bq));
      *(u32x2*)(k8 + ((int64_t)bh * S + s) * DH + d0) = quant8_e4m3(kf, e8m0_inv(bk));
    }
    if (c == 0) { eq[(int64_t)bh * Sp + s] = (uint8_t)bq; ek[(int64_t)bh * Sp + s] = (uint8_t)bk; }    // s < nt64 * 64 <= Sp
    *(u32x4*)(tile + 2 * 32 * DH + row * DH + d0) = vv;                  // zero rows beyond S
    if (qt) {
      *(u32x4*)(tile + row * DH + d0) = qz;
      *(u32x4*)(tile + 32 * DH + row * DH + d0) = kz;
    }
  }
  __syncthreads();
  const int dd = tid;               // thread = one d
  if (qt && s0 < ((S + 31) & ~31)) {
    const int64_t toff = (((int64_t)bh * (ld_t >> 5) + blockIdx.x) * DH + dd) * 32;   // column-tiled: [b,h][tile][256][32]
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      mg_bf16* dst = (which == 0 ? qt : kt) + toff;
      const mg_bf16* tl = tile + which * 32 * DH;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = (uint32_t)tl[(g * 8 + w * 2) * DH + dd] | ((uint32_t)tl[(g * 8 + w * 2 + 1) * DH + dd] << 16);
        *(u32x4*)(dst + g * 8) = o;
      }
    }
  }
  {
    // V^T of this 32-key block for column d: one MX block (32 keys of one d), its E8M0, its 32 elements in the kernel's key order
    const mg_bf16* vt_ = tile + 2 * 32 * DH;
    float x[32], amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { x[i] = bf2f(vt_[i * DH + dd]); amax = fmaxf(amax, fabsf(x[i])); }
    const int bv = e8m0_cover(amax);
    const float inv = e8m0_inv(bv);
    const int t64 = s0 >> 6, bb = (s0 >> 5) & 1;
    sv8[(((int64_t)bh * nt64 + t64) * 2 + bb) * 256 + (dd & 31) * 8 + (dd >> 5)] = (uint8_t)bv;
    uint8_t* dst = v8t + (((int64_t)bh * nt64 + t64) * DH + dd) * 64 + 16 * bb;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      int w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int k0 = (r & 3) + 8 * (r >> 2) + 4 * hi;          // keys of registers r, r + 1 (consecutive: r even)
        if (r & 2) w[r >> 2] = __builtin_amdgcn_cvt_pk_fp8_f32(x[k0] * inv, x[k0 + 1] * inv, w[r >> 2], true);
        else w[r >> 2] = __builtin_amdgcn_cvt_pk_fp8_f32(x[k0] * inv, x[k0 + 1] * inv, w[r >> 2], false);
      }
      *(u32x4*)(dst + 32 * hi) = (u32x4){(uint32_t)w[0], (uint32_t)w[1], (uint32_t)w[2], (uint32_t)w[3]};
    }
  }
}

// ---------------------------------------------------------------------------
// flash attention forward, causal, dh = 256.
// One workgroup = 128 queries = 8 waves x 16 query rows.  Per KV tile of 32 keys and per wave:
//   S^T[key][q] = K . Q^T   (2 key-subtiles x 8 k-steps  = 16 MFMA)
//   O^T[d][q]  += V^T . P^T (16 d-subtiles x 1 k-step    = 16 MFMA)
// Both products are computed transposed so that every per-query quantity
// (running max, sum, rescale factor) is lane-local: lane&15 = query.
// The key permutation keymap(i,t) = (i>>2)*8 + t*4 + (i&3) makes the S^T
// accumulator registers of a lane exactly the 8 consecutive keys it must
// supply as the P^T operand of the PV product -- no cross-lane movement.
// K rows and V^T tiles stream through a 4-stage LDS ring filled by LDS-DMA (three tiles in
// flight, counted vmcnt + one barrier per tile) -- same pipeline as attention_bwd.hip.
// ---------------------------------------------------------------------------
constexpr int FA_STAGE = ROW_TILE + T_TILE;   // K rows | V^T
constexpr int FA_STAGES = 4;

constexpr float FA2_DEFER = 8.0f;                   // log2 units: rescale only when the running max grows by more than 256x (guide T13)

// ---------------------------------------------------------------------------
// The kernel: software-pipelined across KV tiles (round 4).
//
// Rounds 1-3 ran a tile's four parts as ONE dependent chain per wave -- K reads -> S^T MFMAs -> softmax VALU (~100
// instructions, several dependent reductions) -> PV MFMAs (git history: attn_prefill_kernel; 0.926-0.953 ms per layer at
// B = 16, S = 2048 against 0.901-0.949 for this one in the same runs, profiles/r04_attention_fwd_variants.txt).  Here the
// softmax of a tile is split in two halves that each sit beside an MFMA burst of ANOTHER tile (the "att[2]" pipeline of the
// guide, T15):
//     iteration t:   [ K(t+1) fragment reads  ||  part 2 of softmax(t): exp2, row sums, bf16 pack ]
//                    [ S^T(t+1) = K(t+1) Q^T  : 16 MFMAs        ||  V^T(t) fragment reads          ]
//                    [ O^T += V^T(t) P^T(t)   : 16 MFMAs        ||  part 1 of softmax(t+1): mask, row max, new running
//                                                                    max, rescale factor (VALU placed between the MFMAs) ]
// so a wave's chain per tile is  K reads -> 16 MFMAs -> 16 MFMAs  and the exponentials are off it.  Same arithmetic in the
// same order per query as the chain form (same tiles, same deferred running max, same accumulation order): bit-identical
// results (every attention test of rounds 1-3 passes unchanged).  The K tile is read one iteration earlier than before, so the ring is waited one tile deeper
// (tile t+1 complete at the top of iteration t; tiles t+2, t+3 in flight afterwards).
// T13 (deferred max) order: P(t) is exponentiated against the maximum decided for tile t, O / l are rescaled by alpha(t)
// BEFORE P(t) V(t) is added, and alpha(t+1) -- decided while P(t) V(t) is still in the pipe -- is applied in iteration t+1,
// after that product has completed: nothing is ever scaled twice or not at all.
// ---------------------------------------------------------------------------
// ABL (timing ablations, WRONG results; `make ABL=1` library only, MAGMA_ATTN_ABL=n): 1 = no exponentials / row sums (P = the raw
// scores), 2 = no LDS fragment reads after the first tile (stale registers), 3 = no MFMAs, 4 = no LDS-DMA after the prologue,
// 5 = no barrier (and no DMA): each part's price is the time it removes.
template <int ABL = 0>
__global__ __launch_bounds__(512) void attn_prefill_sp_kernel(
    const mg_bf16* __restrict__ q, const mg_bf16* __restrict__ kcache,
    const mg_bf16* __restrict__ vt, mg_bf16* __restrict__ out, int64_t ld_out, float* __restrict__ lse,
    int B, int H, int S, int Smax, int vt_ld, float defer) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int nblk = (S + 127) >> 7;
  const int wg = xcd_contiguous_index(blockIdx.x, gridDim.x);
  const int bh = wg / nblk, b = bh / H, h = bh - b * H;
  const int qt0 = (nblk - 1 - (wg - bh * nblk)) * 128;   // longest blocks first
  const int qrow = qt0 + wave * 16 + li;                 // this lane's query
  const int qrow_c = min(qrow, S - 1);
  const mg_bf16* kbase = kcache + (int64_t)bh * Smax * DH;
  const mg_bf16* vbase = vt + (int64_t)bh * DH * vt_ld;

  const int kv_end = min(S, qt0 + 128);
  const int ntiles = (kv_end + 31) >> 5;
  const uint32_t smem_u = lds_u32(smem);
  auto issue = [&](int t, int buf) {
    const int c0 = min(t, ntiles - 1) * 32;
    const uint32_t st = smem_u + (uint32_t)(buf * FA_STAGE);
    dma_rows(st, kbase, DH, c0, S, wave, lane);
    dma_cols(st + ROW_TILE, vbase, vt_ld, c0, wave, lane);
  };
#pragma unroll
  for (int i = 0; i < FA_STAGES - 1; ++i) issue(i, i);

  bf16x8 qf[8];
  {
    const mg_bf16* qp = q + ((int64_t)bh * S + qrow_c) * DH + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  }
  f32x4 o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m2 = -1e30f, lsum = 0.f;
  const float sc2 = 0.0625f * 1.4426950408889634f;  // 1/sqrt(256) * log2(e)
  const int my_first = qt0 + wave * 16;
  const int my_last = my_first + 15;
  const int n_act = min(ntiles, (my_last >> 5) + 1);   // this wave's tiles: 0 .. n_act-1 (later ones are fully masked for it)
  const int tsw = t_swz(li);
  const int krow0 = (li >> 2) * 8 + (li & 3);
  const int sw0 = row_swz(krow0);
  MG_USE8(qf);

  // part 1 of the softmax of the tile whose scores are in st[]: mask, row maximum, running maximum, rescale factor.
  // Leaves the masked scores in sv[], returns alpha; m2 becomes the reference point of this tile's exponentials.
  // Branch-free (one basic block with the PV MFMAs it is scheduled between): key j of this lane's 8 is visible iff
  // j <= min(query, S - 1) - kv0 - 8 lq.  A tile past this wave's last query comes out fully masked: row maximum -1e30,
  // running maximum unchanged, alpha = 1 -- computed once at the end of a wave's tiles and never used.
  float sv[8];
  const int lim0 = min(qrow, S - 1) - lq * 8;
  auto part1 = [&](const f32x4 (&st)[2], int kv0) -> float {
    const int lim = lim0 - kv0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sv[j] = j > lim ? -1e30f : st[j >> 2][j & 3];
    float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    tmax = quad_rows_max(tmax);
    const float cand = tmax * sc2;
    const float mnew = (cand > m2 + defer) ? cand : m2;
    const float a = __builtin_amdgcn_exp2f(m2 - mnew);
    m2 = mnew;
    return a;
  };

  bf16x8 fa[8], fb[8];
  f32x4 sn[2];
  float alpha;
  // ---- prologue: tiles 0 and 1 landed; S^T(0) and part 1 of its softmax ----
  MG_WAIT_VMCNT(4);
  MG_BARRIER_KEEP_DMA();
  {
    const char* kp = smem + krow0 * 512;
    rd_row8(fa, kp, lq, sw0);
    rd_row8(fb, kp + 4 * 512, lq, sw0);
    sn[0] = mma8(fa, qf);
    sn[1] = mma8(fb, qf);
    alpha = part1(sn, 0);
  }
  int sc = 0;
  for (int t = 0; t < ntiles; ++t) {
    if constexpr (ABL != 4 && ABL != 5) MG_WAIT_VMCNT(4);   // this wave's pieces of tile t+1 landed (tile t+2 may be in flight)
    if constexpr (ABL != 5) MG_BARRIER_KEEP_DMA();          // tile t+1 complete; everyone is done with iteration t-1 (K(t), V^T(t-1))
    if constexpr (ABL != 4 && ABL != 5) issue(t + FA_STAGES - 1, sc == 0 ? FA_STAGES - 1 : sc - 1);
    const int scn = sc == FA_STAGES - 1 ? 0 : sc + 1;
    if (t < n_act) {
      // ---- block A: K(t+1) fragment reads | part 2 of softmax(t) | S^T(t+1) MFMAs | V^T(t) fragment reads ----
      // (tile t+1 exists in the ring even past the last tile: issue() clamps to it; its scores are then fully masked)
      if (ABL != 2 || t == 0) {
        const char* kp = smem + scn * FA_STAGE + krow0 * 512;
        rd_row8(fa, kp, lq, sw0);
        rd_row8(fb, kp + 4 * 512, lq, sw0);
      }
      float psum = 0.f;
      float p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = ABL == 1 ? sv[j] : __builtin_amdgcn_exp2f(fmaf(sv[j], sc2, -m2));
        if (ABL != 1) psum += p[j];
      }
      lsum = lsum * alpha + psum;
      u32x4 pw;
#pragma unroll
      for (int j = 0; j < 4; ++j) pw[j] = pack2bf(p[2 * j], p[2 * j + 1]);
      // (pins part 2 HERE: its results are first used in block B, and LLVM would sink the whole computation there,
      //  back onto the chain in front of the PV MFMAs)
      asm volatile("" : "+v"(pw[0]), "+v"(pw[1]), "+v"(pw[2]), "+v"(pw[3]), "+v"(lsum));
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
      MG_SCHED_FENCE();
      if constexpr (ABL != 3) {
        sn[0] = mma8(fa, qf);
        sn[1] = mma8(fb, qf);
      } else {
        sn[0] = __builtin_bit_cast(f32x4, fa[0]); sn[1] = __builtin_bit_cast(f32x4, fb[0]);
      }
      MG_SCHED_FENCE();
      const char* tp = smem + sc * FA_STAGE + ROW_TILE + li * 64 + ((lq ^ tsw) << 4);
      if (ABL != 2 || t == 0) {
        rd_t8(fa, tp);
        rd_t8(fb, tp + 8 * 1024);
      }
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {      // rare under the deferred running max
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) o[dt] *= alpha;
      }
      // ---- block B: O^T += V^T(t) P^T(t), 16 MFMAs, with part 1 of softmax(t+1) placed between them ----
      if constexpr (ABL != 3) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[dt], pf, o[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[8 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[dt], pf, o[8 + dt], 0, 0, 0);
      } else {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) { asm volatile("" ::"v"(fa[dt]), "v"(fb[dt]), "v"(pf)); }
      }
      alpha = part1(sn, (t + 1) * 32);
#pragma unroll
      for (int g = 0; g < 16; ++g) {       // 1 MFMA : 3 VALU for as long as part 1 lasts
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    }
    sc = scn;
  }
  MG_WAIT_VMCNT(0);
  lsum = quad_rows_sum(lsum);
  if (qrow < S) {
    const float inv = 1.0f / lsum;
    mg_bf16* op = out + (int64_t)(b * S + qrow) * ld_out + h * DH + lq * 4;
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

extern "C" int mg_rotary_split_bf16(const mg_bf16* qkv, int64_t ld_qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                                    const float* sin_t, const float* cos_t, int32_t pos0_host,
                                    const int32_t* d_pos, mg_bf16* q_out, mg_bf16* kcache,
                                    mg_bf16* vcache, int32_t Smax, mg_bf16* vt, int32_t vt_ld,
                                    void* stream) {
  if (B <= 0 || S <= 0 || H <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: B,S,H must be positive");
  if (ld_qkv == 0) ld_qkv = (int64_t)3 * H * DH;
  if (ld_qkv < (int64_t)3 * H * DH || (ld_qkv & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: ld_qkv must be a multiple of 8 and >= 3*H*256");
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !q_out || !kcache || !vcache || (rot_dim && (!sin_t || !cos_t))) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: null pointer");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(q_out) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vcache) || !MG_ALIGNED16(vt))
    MG_FAIL(MG_ERR_ALIGN, "mg_rotary_split_bf16: pointers must be 16-byte aligned");
  if (vt) {
    if (d_pos || pos0_host != 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: V^T output requires pos0 == 0 (prefill)");
    if ((vt_ld & 31) || vt_ld < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: vt_ld must be a multiple of 32 and >= S");
  }
  if (!d_pos && pos0_host + S > Smax) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_bf16: pos0+S exceeds Smax");
  dim3 grid((S + 31) / 32, B * H);
  hipLaunchKernelGGL(rotary_split_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, qkv, ld_qkv, B, S, H, rot_dim, sin_t,
                     cos_t, pos0_host, d_pos, q_out, kcache, vcache, Smax, vt, vt_ld, nullptr, nullptr);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// training form: positions 0..S-1, q / k / v [B,H,S,256] and all three transposes vt, qt, kt [B,H,ld_t/32,256,32] in one pass
extern "C" int mg_rotary_split_train_bf16(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                                          const float* sin_t, const float* cos_t, mg_bf16* q, mg_bf16* k, mg_bf16* v,
                                          mg_bf16* vt, mg_bf16* qt, mg_bf16* kt, int32_t ld_t, void* stream) {
  if (B <= 0 || S <= 0 || H <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_train_bf16: B,S,H must be positive");
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_train_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !q || !k || !v || !vt || !qt || !kt || (rot_dim && (!sin_t || !cos_t))) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_train_bf16: null pointer");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(q) || !MG_ALIGNED16(k) || !MG_ALIGNED16(v) || !MG_ALIGNED16(vt) || !MG_ALIGNED16(qt) || !MG_ALIGNED16(kt))
    MG_FAIL(MG_ERR_ALIGN, "mg_rotary_split_train_bf16: pointers must be 16-byte aligned");
  if ((ld_t & 31) || ld_t < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_train_bf16: ld_t must be a multiple of 32 and >= S");
  dim3 grid((S + 31) / 32, B * H);
  hipLaunchKernelGGL(rotary_split_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, qkv, (int64_t)3 * H * DH, B, S, H, rot_dim, sin_t,
                     cos_t, 0, nullptr, q, k, v, S, vt, ld_t, qt, kt);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// training form for the fp8 attention forward: everything mg_rotary_split_train_bf16 writes except V^T, plus the OCP MX e4m3 copies
// (rotary_split_fp8_kernel): q8 / k8 [B,H,S,256] with one E8M0 per token in eq / ek [B,H,Sp] (bytes; Sp = mg_attn_fp8_scale_stride(S)),
// v8t [B,H,ceil(S/64),256,64] with one E8M0 per (d, 32 keys) in sv8 [B,H,ceil(S/64),512].  qt / kt may be NULL (no backward), and
// so may q / k / v together (forward only: just the e4m3 operands).  inplace != 0: the rotated q / k are also written back into
// qkv (what mg_rotary_qk_inplace_bf16 does, without its pass): the bf16 attention backward reads rows of that buffer.
extern "C" int32_t mg_attn_fp8_scale_stride(int32_t S) { return ((S + 63) / 64) * 64 + 256; }
extern "C" int mg_rotary_split_fp8(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim, const float* sin_t,
                                   const float* cos_t, mg_bf16* q, mg_bf16* k, mg_bf16* v, mg_bf16* qt, mg_bf16* kt, int32_t ld_t,
                                   uint8_t* q8, uint8_t* k8, uint8_t* v8t, uint8_t* eq, uint8_t* ek, uint8_t* sv8, int32_t inplace, void* stream) {
  if (B <= 0 || S <= 0 || H <= 0) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_fp8: B,S,H must be positive");
  if (rot_dim < 0 || rot_dim > DH || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_fp8: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !q8 || !k8 || !v8t || !eq || !ek || !sv8 || (rot_dim && (!sin_t || !cos_t)) || (!qt != !kt) || (!q != !k) || (!q != !v) || (qt && !q))
    MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_fp8: null pointer (q / k / v together or not at all; qt / kt only with them)");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(q) || !MG_ALIGNED16(k) || !MG_ALIGNED16(v) || !MG_ALIGNED16(qt) || !MG_ALIGNED16(kt) ||
      !MG_ALIGNED16(q8) || !MG_ALIGNED16(k8) || !MG_ALIGNED16(v8t) || !MG_ALIGNED16(sv8) || ((uintptr_t)eq & 3) || ((uintptr_t)ek & 3))
    MG_FAIL(MG_ERR_ALIGN, "mg_rotary_split_fp8: pointers must be 16-byte aligned (eq / ek: 4)");
  if (qt && ((ld_t & 31) || ld_t < ((S + 31) & ~31))) MG_FAIL(MG_ERR_SHAPE, "mg_rotary_split_fp8: ld_t must be a multiple of 32 and >= S");
  dim3 grid(((S + 63) / 64) * 2, B * H);      // both 32-key halves of every 64-key V^T tile
  hipLaunchKernelGGL(rotary_split_fp8_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, B, S, H, rot_dim, sin_t, cos_t, q, k, v, qt, kt,
                     ld_t, q8, k8, v8t, eq, ek, sv8, mg_attn_fp8_scale_stride(S), inplace);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_attn_prefill_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out, int64_t ld_out,
                                    float* lse, int32_t B, int32_t H, int32_t S, int32_t Smax, int32_t vt_ld,
                                    void* stream) {
  if (ld_out == 0) ld_out = (int64_t)H * DH;
  if (ld_out < (int64_t)H * DH || (ld_out & 3)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: ld_out must be 0 or a multiple of 4 >= H*256");
  if (B <= 0 || H <= 0 || S <= 0 || S > Smax) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: bad B/H/S/Smax");
  if ((vt_ld & 31) || vt_ld < ((S + 31) & ~31)) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: vt_ld must be a multiple of 32 and >= S");
  if (!q || !kcache || !vt || !out) MG_FAIL(MG_ERR_SHAPE, "mg_attn_prefill_bf16: null pointer");
  if (!MG_ALIGNED16(q) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vt) || !MG_ALIGNED16(out)) MG_FAIL(MG_ERR_ALIGN, "mg_attn_prefill_bf16: pointers must be 16-byte aligned");
  // MAGMA_ATTN_FWD=4: 16-query waves x 8 (two per SIMD), deferred running max: 0.90-0.97 ms per layer at B = 16, S = 2048 (=0 keeps
  // the plain running max: 1.00 ms).  Default since round 5: the 32-query-wave kernel with O^T and Q pinned to AGPRs
  // (attention_fwd32.hip: 0.86-0.88 ms same-box, faster at every shape tried).  The round-2 attempt at 32-query waves left
  // the register allocation to hipcc and measured 1.10 ms (profiles/r02_attention_variants.txt).
  // MAGMA_ATTN_FWD=0: plain running max instead of the deferred one (guide T13; 1.00 vs 0.95 ms in round 2)
  // MAGMA_ATTN_FWD=5: 32-query waves on the 32x32x16 MFMA, one wave per SIMD, O^T pinned to AGPRs (attention_fwd32.hip, round 5)
  const char* env_v = getenv("MAGMA_ATTN_FWD");          // read per call: tests and A/B scripts switch it in-process
  const int variant = env_v ? atoi(env_v) : 5;
  const float defer = variant == 0 ? 0.0f : FA2_DEFER;
  if (variant == 5 && !(ld_out & 7))
    return attn_prefill32_launch(q, kcache, vt, out, ld_out, lse, B, H, S, Smax, vt_ld, defer, (hipStream_t)stream, "mg_attn_prefill_bf16");
  const int lds = FA_STAGES * FA_STAGE;
  if (int rc = mg_allow_dynamic_lds((const void*)attn_prefill_sp_kernel<0>, lds, "mg_attn_prefill_bf16")) return rc;
  dim3 grid((unsigned)(((S + 127) / 128) * B * H));
#ifdef MG_GEMM_ABLATIONS      // `make ABL=1`: timing ablations of the pipelined kernel (WRONG results), MAGMA_ATTN_ABL=1..5
  {
    const char* e = getenv("MAGMA_ATTN_ABL");
    const int abl = e ? atoi(e) : 0;
#define MG_ABL(N_)                                                                                                    \
    if (abl == N_) {                                                                                                  \
      if (int rc = mg_allow_dynamic_lds((const void*)attn_prefill_sp_kernel<N_>, lds, "mg_attn_prefill_bf16")) return rc; \
      hipLaunchKernelGGL(attn_prefill_sp_kernel<N_>, grid, dim3(512), lds, (hipStream_t)stream, q, kcache, vt, out, ld_out, lse, B, H, S, Smax, vt_ld, defer); \
      MG_CHECK_LAUNCH();                                                                                              \
      return MG_OK;                                                                                                   \
    }
    MG_ABL(1) MG_ABL(2) MG_ABL(3) MG_ABL(4) MG_ABL(5)
#undef MG_ABL
  }
#endif
  hipLaunchKernelGGL(attn_prefill_sp_kernel<0>, grid, dim3(512), lds, (hipStream_t)stream, q, kcache, vt, out, ld_out, lse, B, H, S, Smax, vt_ld, defer);
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
