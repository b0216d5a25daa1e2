/*
 * magma_hip.h -- C ABI of libmagma_hip.so (gfx950 / MI355X only).
 *
 * The reference (Aleph-Alpha/magma) has no FFI of its own: its hot path is
 * PyTorch eager ops inside three Python seams:
 *     model.image_prefix(x)                 reference magma/magma.py:208,254
 *     model.word_embedding(ids)             reference magma/magma.py:205,258
 *     model.lm(inputs_embeds|input_ids, use_cache, past_key_values, labels)
 *                                           reference magma/magma.py:270-274,
 *                                           magma/sampling.py:81-93
 * Every entry point below is one fused device computation those seams launch
 * (SURVEY.md 2.4, K1..K24).  The host side (magma_amd/, mirror of the
 * reference's Python surface) binds them with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch-ROCm
 *     allocations); the library never allocates, frees or synchronises;
 *   - every call only enqueues on `stream` (a hipStream_t passed as void*);
 *     graph-capture safe;
 *   - bf16 = raw uint16 storage; fp32 where stated; ids int64 (torch.long);
 *   - return 0 on success, negative MG_ERR_* otherwise; message via
 *     mg_last_error() (thread local).  Shapes are validated before any launch.
 */
#ifndef MAGMA_HIP_H
#define MAGMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_SHAPE (-1)
#define MG_ERR_ALIGN (-2)
#define MG_ERR_HIP (-3)
#define MG_ERR_UNSUPPORTED (-4)

#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_GELU_NEW 2

/* weight layouts for the B operand of the GEMMs */
#define MG_W_ROWMAJOR 0 /* W[n*ldw + k], k zero-padded to a multiple of 64   */
#define MG_W_FRAGTILED 1 /* [N/16][Kp/32][64 lanes][8]: lane=(kq*16+n%16) holds \
                            W[n][ks*32+kq*8 .. +7]; N padded to 16, Kp to 64 */

#define MG_A_DENSE 0    /* A[m*lda + k]                                       */
#define MG_A_CONV3X3 1  /* A = NHWC image, implicit im2col, pad 1 stride 1    */

typedef uint16_t mg_bf16;

const char* mg_version(void);
const char* mg_last_error(void);

/* Fused epilogue shared by both GEMM kernels:
 *   v = acc * scale[n] + bias[n];  v = act(v);  v += res0 + res1 + res2;
 *   v = act_after(v);  C[m*ldc + n] = v  (bf16, or fp32 when out_f32)
 * Covers: Linear(+bias) / gelu_new / adapter ReLU / 3-way GPT-J residual /
 * folded BatchNorm + ReLU / bottleneck "relu(bn3(conv3)+identity)".          */
typedef struct mg_epilogue {
  const float* scale; /* [N] or NULL */
  const float* bias;  /* [N] or NULL */
  int32_t act;        /* MG_ACT_* applied before the residual adds */
  int32_t act_after;  /* MG_ACT_NONE or MG_ACT_RELU, after the residual adds */
  const mg_bf16* res0;
  const mg_bf16* res1;
  const mg_bf16* res2;
  int64_t ldr;        /* row stride of res* (elements) */
  void* C;
  int64_t ldc;
  int32_t out_f32;
  int32_t _pad;
} mg_epilogue;

/* K9/K11/K12/K13/K14/K18 (GPT-J + adapter GEMMs, prefill/training shapes),
 * K6 (ImagePrefix proj), K2/K3 (CLIP 1x1 / 3x3 convs as implicit GEMM).
 * C[M,N] = A[M,K] * W[N,K]^T, bf16 in, fp32 accumulate, MFMA 16x16x32.
 * Replaces F.linear / F.conv2d calls reached from reference
 * magma/image_prefix.py:83,93 and the LM call at magma/magma.py:270-274.    */
typedef struct mg_gemm_desc {
  const mg_bf16* A;
  int64_t lda;
  const mg_bf16* W;
  int64_t ldw;       /* rowmajor: row stride; fragtiled: padded K (Kp)        */
  int32_t M, N, K;   /* K multiple of 8                                       */
  int32_t a_mode;    /* MG_A_*                                                */
  int32_t w_layout;  /* MG_W_*                                                */
  int32_t H, Wd, Cin; /* conv3x3 only: A is [B,H,Wd,Cin], M = B*H*Wd, K=9*Cin */
  const mg_bf16* zero_page; /* >= 16 zero bytes, 16-B aligned (K/halo padding) */
  mg_epilogue ep;
} mg_gemm_desc;

int mg_gemm_bf16(const mg_gemm_desc* d, void* stream);

/* Decode-shape (M <= 16) weight-streaming GEMM: HBM-bound, W must be
 * MG_W_FRAGTILED.  Same epilogue.  Used for every projection of a decode
 * step (reference magma/sampling.py:86-90 -> model.lm(input_ids=...)).       */
typedef struct mg_skinny_desc {
  const mg_bf16* X;
  int64_t ldx;
  const mg_bf16* W;
  int32_t M, N, Kp;  /* Kp = padded K of the tiled weight (multiple of 64)    */
  int32_t nt_hint;   /* 0 = auto; else n-tiles (of 16 rows) per workgroup     */
  mg_epilogue ep;
} mg_skinny_desc;

int mg_gemm_skinny_bf16(const mg_skinny_desc* d, void* stream);

/* K8/K17 + ImagePrefix LN: y = (x-mean)/sqrt(var+eps)*gamma+beta, fp32 stats */
int mg_layernorm_bf16(const mg_bf16* x, int64_t ldx, const float* gamma, const float* beta,
                      mg_bf16* y, int64_t ldy, int32_t rows, int32_t d, float eps, void* stream);

/* K7: out[b*out_bstride + (row_off+t)*d .. ] = wte[ids[b*T+t]]
 * (reference magma/magma.py:205,258; writes straight into the concatenated
 * [prefix|text] buffer, replacing torch.cat at magma.py:212,261-267).        */
int mg_embedding_bf16(const int64_t* ids, int32_t B, int32_t T, const mg_bf16* wte, int32_t vocab,
                      int32_t d, mg_bf16* out, int64_t out_bstride, int32_t row_off, void* stream);

/* K9 epilogue: split fused qkv rows, GPT-J interleaved rotary on the first
 * rot_dim dims of q,k, scatter K,V into the cache, optional V^T for prefill.
 *   qkv [B*S, 3*H*256];  q_out [B,H,S,256];  kcache/vcache [B,H,Smax,256];
 *   vt (nullable) [B,H,256,vt_ld];  position of row s = pos0 + s where
 *   pos0 = d_pos ? *d_pos : pos0_host.  sin/cos tables [n_pos, rot_dim/2] f32 */
int mg_rotary_split_bf16(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                         const float* sin_t, const float* cos_t, int32_t pos0_host,
                         const int32_t* d_pos, mg_bf16* q_out, mg_bf16* kcache, mg_bf16* vcache,
                         int32_t Smax, mg_bf16* vt, int32_t vt_ld, void* stream);

/* K10 prefill/training forward: causal flash attention, head dim 256, fp32
 * online softmax, scale 1/16.  q [B,H,S,256]; k rows from kcache [B,H,Smax,256];
 * vt [B,H,256,vt_ld]; out [B*S, H*256].  lse (nullable) [B,H,S] fp32.         */
int mg_attn_prefill_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out,
                         float* lse, int32_t B, int32_t H, int32_t S, int32_t Smax, int32_t vt_ld,
                         void* stream);

/* K10 decode: one query row per (b,h) against ctx = *d_pos + 1 cached keys.   */
int mg_attn_decode_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vcache,
                        mg_bf16* out, int32_t B, int32_t H, int32_t Smax, const int32_t* d_pos,
                        void* stream);

/* K24 greedy: token[b] = argmax_v logits[b, v] (first maximum), int64 out;
 * optionally appends to out_tokens[b*out_ld + *d_pos_out] and bumps *d_pos.  */
int mg_argmax_f32(const float* logits, int64_t ld, int32_t B, int32_t V, int64_t* token,
                  void* stream);
int mg_advance_pos(int32_t* d_pos, int32_t delta, void* stream);

/* K4: 2x2 average pool, NHWC bf16.  x [B,H,W,C] -> y [B,H/2,W/2,C]           */
int mg_avgpool2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C,
                          void* stream);

/* K1 stem conv1 (3->C, 3x3, stride 2, pad 1): explicit im2col of the NCHW
 * bf16 image into [B*(H/2)*(W/2), 32] (27 taps*channels + 5 zero columns,
 * column = (ky*3+kx)*3 + c) for mg_gemm_bf16.                                */
int mg_stem_im2col_bf16(const mg_bf16* img_nchw, mg_bf16* out, int32_t B, int32_t H, int32_t W,
                        void* stream);

/* K20 (integer, exact): reference magma/utils.py:334-364 build_labels.
 * labels[b, :P] = -100; labels[b, P+t] = captions[b, t] up to and including
 * the first eos of the row; -100 after.  captions/labels [B, S] int64.       */
int mg_build_labels_i64(const int64_t* captions, int64_t* labels, int32_t B, int32_t S, int32_t P,
                        int64_t eos, void* stream);

/* K19: shifted CE pieces.  For each row r of `logits` [R, V] fp32 with target
 * tgt[r] (int64, -100 = ignore): loss_row[r] = logsumexp - logit[tgt] (0 if
 * ignored).  The caller passes already-shifted rows.  mg_ce_reduce produces
 * mean over valid rows into out[0] and the valid count into out[1].          */
int mg_ce_rows_f32(const float* logits, int64_t ld, const int64_t* tgt, float* loss_row,
                   int32_t R, int32_t V, void* stream);
int mg_ce_reduce_f32(const float* loss_row, const int64_t* tgt, int32_t R, float* out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGMA_HIP_H */
