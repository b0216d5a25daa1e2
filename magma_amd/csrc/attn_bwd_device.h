// attn_bwd_device.h -- what the two translation units of the flash-attention backward share
// (attention_bwd.hip: statistics pass, 16-row-wave kernels, C entry points; attention_bwd32.hip: the 32-row-wave kernels).
#pragma once
#include "common.h"
#include "attn_tile_device.h"

constexpr int LD_TILE = 256;              // the statistics of 32 queries, {-16 lse, -D} (fp32): 8 B each, interleaved or as two arrays of 32

// Where a gradient row goes.  merged == nullptr: out[(bh*S + s)*256 + d] (dq / dk / dv as [B,H,S,256]).  Otherwise the
// row lands in the gradient of the fused qkv projection, merged[(b*S + s)*3*H*256 + which*H*256 + h*256 + d], with the
// inverse GPT-J rotary R(-theta_s) applied to the first rot_dim columns of dq and dk (what mg_rotary_merge_bwd_bf16 did
// in a separate pass over 3 x [B,H,S,256]).
struct GradOut {
  mg_bf16* out;
  mg_bf16* merged;
  const float* sin_t;
  const float* cos_t;
  int which, rot_dim;
  // round 6 (attention_tr.hip, merged form only): the OCP MX e4m3 copy of the merged gradient, q8 [B*S, 3 H 256] + E8M0 scales in the
  // layout of mg_quantize_mx_fp8 -- the operand of the qkv dgrad's MX GEMM, written from the epilogue's row pieces (no quantisation
  // pass over dqkv).  `merged` may then be null (no bf16 copy at all).
  uint8_t* q8 = nullptr;
  uint8_t* q8_scales = nullptr;
  int mx_rows = 0;          // B * S: the row count the scale layout is built for
};
MG_DEV mg_bf16* grad_row_ptr(const GradOut& g, int b, int h, int H, int S, int s) {
  return (g.merged || !g.out) ? g.merged + ((int64_t)b * S + s) * (3 * H * DH) + (int64_t)g.which * H * DH + h * DH
                  : g.out + (((int64_t)b * H + h) * S + s) * DH;
}

// attention_bwd32.hip: dK and dV of one (b, h, 128 keys) in ONE kernel (S and dP computed once), 4 waves x 32 keys on the
// 32x32x16 MFMA, one wave per SIMD.  Same operands as the 16-row kernels.
int attn_bwd_dkdv32_launch(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt, const mg_bf16* dO,
                           const mg_bf16* dOt, const float* ld2, const GradOut& gk, const GradOut& gv, int B, int H, int S,
                           int ld_t, int variant, hipStream_t s, const char* who);
// dQ of one (b, h, 128 queries): 4 waves x 32 queries, same structure
int attn_bwd_dq32_launch(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* kt, const mg_bf16* dO,
                         const float* ld2, const GradOut& gq, int B, int H, int S, int ld_t, int variant, hipStream_t s,
                         const char* who);
// ---- attention_tr.hip: the attention kernels WITHOUT transposed operand images (ds_read_b64_tr_b16 from the row images) ----------
// q / k / v of head (b, h) as rows of 256 at an arbitrary row stride: position s of the head is at ptr + b stride_b + h stride_h + s ld
// (elements).  [B,H,S,256]: ld 256, stride_h S 256, stride_b H S 256.  The fused qkv activation [B*S, 3 H 256] itself: q = qkv,
// k = qkv + H 256, v = qkv + 2 H 256, ld 3 H 256, stride_h 256, stride_b S 3 H 256 -- no split pass, no copies.
struct AttnRows {
  const mg_bf16 *q, *k, *v;
  int64_t stride_b, stride_h;
  int ld;
};
int attn_bwd_dkdv32_tr_launch(const AttnRows& x, const mg_bf16* dO, const float* ld2,
                              const GradOut& gk, const GradOut& gv, int B, int H, int S, int stages, hipStream_t s, const char* who, int nrun = 0);
int attn_bwd_dq32_tr_launch(const AttnRows& x, const mg_bf16* dO, const float* ld2,
                            const GradOut& gq, int B, int H, int S, int stages, hipStream_t s, const char* who, int nrun = 0);
int attn_fwd32_tr_launch(const AttnRows& x, mg_bf16* out, int64_t ld_out, float* lse, int B, int H, int S, float defer, hipStream_t s,
                         const char* who);
