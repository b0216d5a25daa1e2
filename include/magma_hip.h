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
#define MG_ERR_COMM (-5)          /* RCCL missing or an RCCL call failed (mg_comm_*) */

#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_GELU_NEW 2
#define MG_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x): CLIP ViT MLP */

/* auxiliary-operand modes of the epilogue (backward passes, dropout) */
#define MG_AUX_NONE 0
#define MG_AUX_RELU_GATE 1 /* v *= (aux > 0)            -- ReLU backward            */
#define MG_AUX_GELU_GRAD 2 /* v *= gelu_new'(aux)       -- aux = saved pre-activation */
#define MG_AUX_MUL 3       /* v *= aux                  -- dropout mask (pre-scaled)  */
#define MG_AUX_QUICK_GELU_GRAD 4 /* v *= d/dx [x sigmoid(1.702 x)](aux) -- CLIP ViT MLP backward, aux = saved pre-activation */

/* weight layouts for the B operand of the GEMMs */
#define MG_W_ROWMAJOR 0 /* W[n*ldw + k], k zero-padded to a multiple of 64   */
#define MG_W_FRAGTILED 1 /* [N/16][Kp/32][64 lanes][8]: lane=(kq*16+n%16) holds \
                            W[n][ks*32+kq*8 .. +7]; N padded to 16, Kp to 64 */

#define MG_A_DENSE 0    /* A[m*lda + k]                                       */
#define MG_A_CONV3X3 1  /* A = NHWC image, implicit im2col, pad 1 stride 1    */

typedef uint16_t mg_bf16;

const char* mg_version(void);
/* Binary interface revision.  A binder built against another revision must not call into the library: arguments moved.
 *   1  rounds 1-3.  Within it (before this counter existed) three signatures changed: mg_epilogue._pad became act_n0;
 *      mg_rotary_split_bf16 gained ld_qkv as its 2nd argument; mg_sample_f32's top_p went from float to double.
 *   2  round 4: mg_decode_attn_gemv_bf16 gained ld_attn_out (5th argument), mg_attn_prefill_bf16 gained ld_out (5th), mg_attn_bwd_bf16 /
 *      mg_attn_bwd_merged_bf16 gained ld_o (before the stream); removed: mg_decode_attn_2gemv_bf16,
 *      mg_decode_ctx_counter_ints, the persistent decode step's four entry points mg_decode_plan_* / mg_decode_step_* (in-launch hand-off
 *      experiments, measured slower than the launch chain: DESIGN.md 8).
 *   3  round 5: added mg_stream_create_cu_mask / mg_stream_destroy, mg_rotary_split_fp8 / mg_attn_prefill_fp8 /
 *      mg_attn_fp8_scale_stride (nothing moved; a revision-2 binder keeps working, the loader of this repo asks for 3 because it
 *      binds the new ones).
 *   4  round 5: mg_epilogue grew by C8 / ldc8 / c8_scales / c8_rgroups (the MX e4m3 copy of a tile GEMM's output): every
 *      descriptor that embeds an epilogue changed size.
 *   5  round 6: added mg_rotary_qk_inplace_bf16 / mg_attn_fwd_rows_bf16 / mg_attn_bwd_rows_bf16 (attention without transposed operand
 *      images, q / k / v taken as strided rows -- straight from the fused qkv activation); mg_attn_prefill_fp8 gained
 *      out8 / ld_out8 / out8_scales (the MX e4m3 copy of its output, before the stream).
 *   6  round 6: mg_epilogue.reserved0 became `accumulate` and the struct grew by `row_scale` (weight-gradient GEMMs add
 *      row_scale[m] * (A W^T) straight into the fp32 gradient: no temporary, no second pass): every descriptor that embeds an
 *      epilogue changed size.  Added mg_conv_weight_relayout_batch / mg_bn_fold_batch (one launch per step for all
 *      convolutions of the image encoder) and mg_transpose_bn_param_grad_bf16; mg_rotary_split_fp8 gained `inplace` (before the stream):
 *      the rotated q / k also written back into the fused qkv activation, replacing a mg_rotary_qk_inplace_bf16 pass; mg_attn_bwd_rows_bf16
 *      gained `first_rows` (before the stream): only the gradients of the first positions (the bottom block of a frozen LM).                                                                                            */
#define MG_ABI_VERSION 6
int32_t mg_abi_version(void);
const char* mg_last_error(void);

/* Fused epilogue shared by both GEMM kernels:
 *   v = acc * scale[n] + bias[n];  C2[m*ldc2+n] = v (optional, bf16: the
 *   pre-activation a later backward pass needs);  v = act(v);
 *   v = aux_op(v, aux[m*ldaux+n]) (unless aux_after);  v += res0 + res1 + res2;
 *   v = aux_op(v, aux) (if aux_after);  v = act_after(v);
 *   C[m*ldc + n] = v  (bf16, or fp32 when out_f32)
 * Covers: Linear(+bias) / gelu_new / adapter ReLU / 3-way GPT-J residual /
 * folded BatchNorm + ReLU / bottleneck "relu(bn3(conv3)+identity)" / dropout /
 * and, in backward, ReLU and GELU gradients fused into the dgrad GEMMs.      */
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
  int32_t aux_mode;   /* MG_AUX_* */
  const mg_bf16* aux;
  int64_t ldaux;
  int32_t aux_after;  /* apply the aux op after the residual adds */
  int32_t act_n0;     /* `act` applies to columns n >= act_n0 only (0: all) -- two projections sharing one A in one launch:
                       * [q|k|v | fc_in] with gelu_new on the fc_in columns; multiple of 8 */
  mg_bf16* C2;        /* optional second output (value before `act`), bf16 */
  int64_t ldc2;
  /* optional OCP MX copy of the final value (what C receives, rounded to bf16 first: the same bytes mg_quantize_mx_fp8 makes
   * of the bf16 output): e4m3 elements C8[m * ldc8 + n] with one E8M0 scale per 32 consecutive n in that quantiser's layout
   * (c8_rgroups = ceil(M / 64)) -- the A operand of a following mg_gemm_mx_fp8 without a quantisation pass in between.
   * Tile GEMMs only (mg_gemm_bf16 / mg_gemm_fp8 / mg_gemm_mx_fp8), un-split, N % 32 == 0, ldc8 == ceil(N / 128) * 128,
   * bf16 C (or C == NULL: only the fp8 copy is written).  ABI revision 4.                                              */
  uint8_t* C8;
  int64_t ldc8;
  uint8_t* c8_scales;
  int32_t c8_rgroups;
  /* fp32 outputs only (out_f32): C[m*ldc+n] += v instead of = v -- the weight-gradient GEMMs accumulate into the gradient
   * buffer in place (reference semantics: .grad += over micro-batches, magma/train_loop.py:22-33 through DeepSpeed).  ABI 6.  */
  int32_t accumulate;
  /* optional per-ROW factor of the accumulator, fp32 [M]: v = acc * row_scale[m] * scale[n] + bias[n] (the folded-BatchNorm scale
   * of a convolution's weight gradient: dW[co][:] = scale[co] * g^T a).  Tile GEMMs only; mg_gemm_fp8 takes its activation row
   * scale as an argument instead and refuses both.  ABI 6.                                                                     */
  const float* row_scale;
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
  int32_t tile_hint; /* 0 = auto; 128 / 256 force the workgroup-tile kernel; 258 / 259: the 256x256 bf16 kernel on the
                      * 32x32x16 / 16x16x32 MFMA (bit-identical, A/B runs); 261..272: timing ablations that exist only in the
                      * ablation build of the library (magma_amd/csrc/Makefile, ABL=1; MG_ERR_UNSUPPORTED otherwise)          */
  int32_t split_k;   /* 0 = auto (only when a workspace is given); 1 = never; n = n-way */
  /* Split-K scratch (optional): small-M GEMMs with a long K (prefill at a few hundred rows,
   * the late CLIP stages) launch fewer tiles than the chip has CUs; with a workspace the
   * library cuts K across several workgroups per tile, each writing an fp32 slab
   * [M][ceil(N/128)*128] here, and a second launch adds the slabs in a fixed order and runs
   * the epilogue (deterministic).  Needs split_k * M * ceil(N/128)*128 * 4 bytes; the
   * library uses fewer splits if it is smaller.  Contents are scratch: do not share one
   * workspace between streams.                                                          */
  float* workspace;
  int64_t workspace_bytes;
} mg_gemm_desc;

int mg_gemm_bf16(const mg_gemm_desc* d, void* stream);
/* Bytes of split-K scratch mg_gemm_bf16 / mg_gemm_fp8 can use for an M x N x K problem at the largest split the automatic
 * policy picks (0 when the shape is never split): what a caller that owns all memory should allocate once.  */
int64_t mg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K);

/* fp8 operands (BASELINE config 5): A and W hold OCP e4m3 bytes (mg_quantize_rows_fp8), the product
 * runs on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) at twice the bf16 MFMA rate with fp32
 * accumulation, and the epilogue applies  acc * row_scale[m] * ep.scale[n]  (per-row activation scale, per-column
 * weight scale) before bias / activation / residuals.  Same descriptor with every K / lda / ldw counting fp8
 * ELEMENTS: K and lda multiples of 16; row-major W: ldw multiple of 16; fragment-tiled W: the bf16 tiling applied
 * to byte pairs, Kp (ldw) multiple of 128; dense A only; 128x128 tile kernel (split-K as for bf16).  Output
 * bf16 / fp32.                                                                                               */
int mg_gemm_fp8(const mg_gemm_desc* d, const float* row_scale, void* stream);
/* OCP MX (microscaling) form -- BASELINE config 5 as SURVEY 8d states it: e4m3 elements with ONE E8M0 scale per 32 consecutive
 * K-elements of every operand row, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 with those block scales (no per-row / per-column
 * scale, fp32 accumulate).
 *   mg_quantize_mx_fp8  x [M, K] bf16 -> q [M, ldq] e4m3 in K order (ldq == ceil(K / 128) * 128, zero padded) + the E8M0 block
 *                       scales (shared exponent floor(log2 max|x|) - 8, biased by 127, and ONE HIGHER when the block maximum
 *                       would otherwise exceed 448 -- the v1.0 formula alone saturates maxima in (448, 512) 2^e by up to 12.5 %,
 *                       which the gradient operands of the training path do not tolerate; revision 4) as
 *                       mg_mx_scale_bytes(M, K) bytes: with R = ceil(M / 64), byte ((((k / 128) * 4 + (k / 32) % 4) * R + m / 64) * 16
 *                       + m % 16) * 4 + (m % 64) / 16 is the scale of (row m, block k / 32) -- one dword load hands an MFMA lane
 *                       what it supplies for four 16-row fragments of a 64-row slab (rows >= M of the last slab: unwritten).
 *   mg_gemm_mx_fp8      descriptor as for mg_gemm_fp8 (K / lda / ldw count fp8 elements; rows padded to whole 128-element chunks;
 *                       row-major or fragment-tiled W: the bf16 tiling applied to the byte pairs of the row-major image), the scale
 *                       arrays of the two operands from the quantiser (4-byte aligned).  tile_hint 0 / 128 / 256: the 128x128
 *                       kernel (scales in registers a K-tile ahead; split-K as for bf16) or the 256x256 one (scales staged
 *                       through LDS with their K-tile; bit-identical un-split); the usual epilogue.
 *   mg_debug_mx_mfma    test probe: ONE wave-level MFMA on caller-supplied operand registers (a, b: [64 lanes][8] dwords) and
 *                       per-lane scale dwords -> out [64][4]; pins the instruction's lane / block / scale-byte semantics.            */
int64_t mg_mx_scale_bytes(int32_t rows, int32_t K);
int mg_quantize_mx_fp8(const mg_bf16* x, int64_t ldx, int32_t M, int32_t K, uint8_t* q, int64_t ldq, uint8_t* scales, void* stream);
int mg_gemm_mx_fp8(const mg_gemm_desc* d, const uint8_t* a_scales, const uint8_t* w_scales, void* stream);
int mg_debug_mx_mfma(const uint32_t* a, const uint32_t* scale_a, const uint32_t* b, const uint32_t* scale_b, float* out, void* stream);
/* bf16 rows -> e4m3 rows + one fp32 scale per row (x ~= q * scale[m], scale = rowmax|x| / 448, round to
 * nearest even, saturating); q columns [K, ldq) are zero-filled.  K, ldx, ldq multiples of 8.                */
int mg_quantize_rows_fp8(const mg_bf16* x, int64_t ldx, int32_t M, int32_t K, uint8_t* q, int64_t ldq,
                         float* scale, void* stream);

/* Decode-shape (M <= 16) weight-streaming GEMM: HBM-bound, W must be
 * MG_W_FRAGTILED.  Same epilogue.  Used for every projection of a decode
 * step (reference magma/sampling.py:86-90 -> model.lm(input_ids=...)).       */
typedef struct mg_skinny_desc {
  const mg_bf16* X;
  int64_t ldx;
  const mg_bf16* W;
  int32_t M, N, Kp;  /* Kp = padded K of the tiled weight (multiple of 64)    */
  int32_t nt_hint;   /* 0 = auto; else variant nt | waves<<4 | kc<<8          */
  mg_epilogue ep;
  /* LayerNorm folded into the GEMV: W must already be W*gamma and ep.bias must be
   * b + W.beta; ln_colsum[n] = sum_k W'[n][k].  The kernel derives mean/rstd of each
   * x row from the fragments it streams:  y = rstd*(acc - mean*ln_colsum[n]) + bias[n].
   * NULL = plain GEMV.  (Saves the separate LayerNorm launch of every decode layer.)   */
  const float* ln_colsum;
  float ln_inv_d, ln_eps;
  /* two output segments in one launch (fused qkv | fc_in of a GPT-J block, which share
   * their input): columns [split_n, N) use ep_b (pointers/vectors indexed from column
   * split_n = 0).  0 = single segment.                                                 */
  int32_t split_n;
  int32_t _pad;
  mg_epilogue ep_b;
  /* fp8 weights, bf16 activations (W8A16; halves the weight stream of a decode step).  NULL = bf16 weights.
   * Otherwise W holds e4m3 bytes tiled as [ceil(N/16)][Kp/64][64 lanes][16 B] (lane = kq*16 + n; bytes 0-7 =
   * W[n][64j + 8kq ..+7], bytes 8-15 = W[n][64j + 32 + 8kq ..+7]), w_scale[n] is the per-output-channel scale
   * (W ~= q * w_scale[n]); the kernel widens the bytes to bf16 in registers (exact) and multiplies the
   * accumulators by w_scale before the LayerNorm fold / epilogue.  Needs Kp % 1024 == 0.                     */
  const float* w_scale;
} mg_skinny_desc;

int mg_gemm_skinny_bf16(const mg_skinny_desc* d, void* stream);

/* Two independent decode GEMVs in ONE launch (out_proj || adapter-down of a GPT-J block:
 * the 64-workgroup adapter GEMV hides under the other one's weight stream).           */
int mg_gemm_skinny2_bf16(const mg_skinny_desc* a, const mg_skinny_desc* b, void* stream);

/* mg_attn_decode_fused_bf16 co-launched with one GEMV that does not depend on it (fc_out of
 * the parallel block): attention workgroups first, GEMV workgroups behind them, one grid.  */
int mg_decode_attn_gemv_bf16(const mg_bf16* qkv, mg_bf16* kcache, mg_bf16* vcache, mg_bf16* attn_out,
                             int64_t ld_attn_out /* row stride of attn_out in elements; 0 = H*256.  ABI 2: a wider row lets the
                                                  * context land beside another activation, [ctx | t], the input of one K-concatenated
                                                  * GEMV [W_out | W_up] */,
                             int32_t B, int32_t H, int32_t Smax, const int32_t* d_pos, int32_t rot_dim,
                             const float* sin_t, const float* cos_t, const mg_skinny_desc* gemv,
                             void* stream);

/* K8/K17 + ImagePrefix LN: y = (x-mean)/sqrt(var+eps)*gamma+beta, fp32 stats.  Replaces nn.LayerNorm at reference
 * magma/image_prefix.py:58-60,106-107 and ln_1 / ln_f of the GPT-J blocks built at magma/language_model.py:12-45
 * (arithmetic in the un-vendored transformers fork).                                                        */
int mg_layernorm_bf16(const mg_bf16* x, int64_t ldx, const float* gamma, const float* beta,
                      mg_bf16* y, int64_t ldy, int32_t rows, int32_t d, float eps, void* stream);

/* K7: out[b*out_bstride + (row_off+t)*d .. ] = wte[ids[b*T+t]]
 * (reference magma/magma.py:205,258; writes straight into the concatenated
 * [prefix|text] buffer, replacing torch.cat at magma.py:212,261-267).        */
int mg_embedding_bf16(const int64_t* ids, int32_t B, int32_t T, const mg_bf16* wte, int32_t vocab,
                      int32_t d, mg_bf16* out, int64_t out_bstride, int32_t row_off, void* stream);

/* K9 epilogue (the fork's GPT-Neo attention with rotary=True, jax=True, configured at reference
 * magma/language_model.py:17-24; called through magma/magma.py:270-274): split fused qkv rows, GPT-J interleaved rotary on the first
 * rot_dim dims of q,k, scatter K,V into the cache, optional V^T for prefill.
 *   qkv [B*S, 3*H*256];  q_out [B,H,S,256];  kcache/vcache [B,H,Smax,256];
 *   vt (nullable) [B,H,vt_ld/32,256,32] (values transposed in 32-key tiles, vt_ld % 32 == 0);  position of row s = pos0 + s where
 *   pos0 = d_pos ? *d_pos : pos0_host.  sin/cos tables [n_pos, rot_dim/2] f32 */
int mg_rotary_split_bf16(const mg_bf16* qkv, int64_t ld_qkv /* row stride in elements, 0 = 3*H*256 */, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                         const float* sin_t, const float* cos_t, int32_t pos0_host,
                         const int32_t* d_pos, mg_bf16* q_out, mg_bf16* kcache, mg_bf16* vcache,
                         int32_t Smax, mg_bf16* vt, int32_t vt_ld, void* stream);

/* Training form of K9 (positions 0..S-1): q, k, v [B,H,S,256] plus ALL THREE column-tiled transposes vt, qt, kt
 * [B,H,ld_t/32,256,32] in one pass over qkv -- vt feeds mg_attn_prefill_bf16, qt / kt are the s-contraction operands
 * of mg_attn_bwd_bf16 (otherwise two more mg_head_transpose_bf16 passes in the backward).                          */
int mg_rotary_split_train_bf16(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                               const float* sin_t, const float* cos_t, mg_bf16* q, mg_bf16* k, mg_bf16* v,
                               mg_bf16* vt, mg_bf16* qt, mg_bf16* kt, int32_t ld_t, void* stream);

/* K10 prefill/training forward (same attention module; reference call sites magma/magma.py:270-274 and the
 * prefill step magma/sampling.py:81-85): causal flash attention, head dim 256, fp32
 * online softmax, scale 1/16.  q [B,H,S,256]; k rows from kcache [B,H,Smax,256];
 * vt [B,H,vt_ld/32,256,32]; out [B*S, H*256].  lse (nullable) [B,H,S] fp32.         */
int mg_attn_prefill_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out,
                         int64_t ld_out /* row stride of out in elements; 0 = H*256 (ABI 2: see mg_decode_attn_gemv_bf16) */,
                         float* lse, int32_t B, int32_t H, int32_t S, int32_t Smax, int32_t vt_ld,
                         void* stream);

/* K10 decode (reference magma/sampling.py:86-90, past_key_values path): one query row per (b,h) against
 * ctx = *d_pos + 1 cached keys.                                                                            */
int mg_attn_decode_bf16(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vcache,
                        mg_bf16* out, int32_t B, int32_t H, int32_t Smax, const int32_t* d_pos,
                        void* stream);

/* K9 epilogue + K10 decode in ONE launch: takes the fused qkv row of the new token
 * [B, 3*H*256], rotates q,k, appends k,v at *d_pos, attends over [0, *d_pos].      */
int mg_attn_decode_fused_bf16(const mg_bf16* qkv, mg_bf16* kcache, mg_bf16* vcache, mg_bf16* out, int32_t B,
                              int32_t H, int32_t Smax, const int32_t* d_pos, int32_t rot_dim,
                              const float* sin_t, const float* cos_t, void* stream);

/* K24 greedy (reference magma/sampling.py:96-97, temperature == 0.0): token[b] = argmax_v logits[b, v] (first maximum), int64 out;
 * optionally appends to out_tokens[b*out_ld + *d_pos_out] and bumps *d_pos.  */
int mg_argmax_f32(const float* logits, int64_t ld, int32_t B, int32_t V, int64_t* token,
                  void* stream);
int mg_advance_pos(int32_t* d_pos, int32_t delta, void* stream);

/* K24 sampled (reference magma/sampling.py:99-107: top_k_filter :22-30, top_p_filter :7-19 -- the reference's own rule, SURVEY Q6 --
 * softmax(logits / temperature), multinomial) as one launch per token step, graph-capturable (csrc/sampling.hip).
 *   logits [B, V] fp32 (row stride ld); top_k == 0 / top_p == 0 disable the respective filter; top_p is a DOUBLE: the reference
 *   compares fp32 probabilities with the Python scalar (1 - threshold), i.e. with fp32(double(1) - double(threshold));
 *   seed  device uint64 (read at run time: a new seed needs no re-capture), state device int32[2] = {step, first step at which
 *   every row produced eos (-1 until then)}: the random stream is Philox4x32-10 keyed by the seed at counter (step, row);
 *   token [B] int64 sampled ids (NULL: filter only); filtered [B, V] optional copy of the filtered logits (-inf where the
 *   reference's filters put -inf).
 * mg_sample_finish is the loop bookkeeping of a token step in one small launch: records the first step with
 * (token == eos).all() (reference sampling.py:109, read by the host every few steps instead of a sync per token), advances
 * the step counter, bumps the KV-cache write position *d_pos by delta (NULL: untouched) and appends the tokens to
 * history[b * ld_history + step] (NULL: off; the host copies the history once when generate() ends); `clear` (NULL: off) =
 * n_clear int32 at clear[i * clear_stride] set to zero for the next step (device-side counters a caller wants re-armed
 * inside the captured step).                                                                                             */
int mg_sample_f32(const float* logits, int64_t ld, int32_t B, int32_t V, float temperature, int32_t top_k, double top_p,
                  const uint64_t* seed, const int32_t* state, int64_t* token, float* filtered, int64_t ld_filtered,
                  void* stream);
int mg_sample_finish(const int64_t* token, int32_t B, int64_t eos, int32_t* state, int32_t* d_pos, int32_t delta,
                     int64_t* history, int64_t ld_history, int32_t history_cols, int32_t* clear, int32_t n_clear,
                     int32_t clear_stride, void* stream);

/* CLIP VisionTransformer front end and attention (encoder_name "clip" = ViT-B/32; reference magma/image_encoders.py:56-63):
 *   patchify   img [B,3,H,W] bf16 NCHW -> rows [B*(H/P)*(W/P), 3*P*P] in (c, py, px) order = the im2col of the stride-P patch
 *              conv, which then is mg_gemm_bf16 against conv1.weight.reshape(width, 3*P*P);
 *   vit_embed  out[b,0,:] = class_embedding + pos[0];  out[b,1+g,:] = patches[b*G+g,:] + pos[1+g]   (fp32 add, bf16 out);
 *   attn_small non-causal multi-head attention for short sequences (S <= 256, head dim 64): qkv [B*S, 3*H*64] = [q | k | v]
 *              with the in_proj bias already added, out [B*S, H*64]; fp32 scores / softmax, scale 1/8.              */
int mg_patchify_bf16(const mg_bf16* img, mg_bf16* out, int32_t B, int32_t H, int32_t W, int32_t P, void* stream);
int mg_vit_embed_bf16(const mg_bf16* patches, const mg_bf16* class_embedding, const mg_bf16* pos, mg_bf16* out, int32_t B,
                      int32_t G, int32_t width, void* stream);
int mg_attn_small_bf16(const mg_bf16* qkv, mg_bf16* out, int32_t B, int32_t S, int32_t H, void* stream);
/* Backward of mg_attn_small_bf16 (training the CLIP ViT encoder; autograd through nn.MultiheadAttention in the reference's
 * CLIP dependency): qkv [B*S, 3*H*64] as in the forward, d_out [B*S, H*64] -> d_qkv [B*S, 3*H*64].  Recomputes the
 * probabilities (fp32) per (b, h); S <= 64.                                                                         */
int mg_attn_small_bwd_bf16(const mg_bf16* qkv, const mg_bf16* d_out, mg_bf16* d_qkv, int32_t B, int32_t S, int32_t H, void* stream);

/* K4: 2x2 average pool, NHWC bf16.  x [B,H,W,C] -> y [B,H/2,W/2,C]  (stem pool and the anti-aliased stride of
 * CLIP's ModifiedResNet bottlenecks; trunk selected at reference magma/image_encoders.py:65-74).              */
int mg_avgpool2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C,
                          void* stream);

/* K1 stem conv1 of the same trunk (3->C, 3x3, stride 2, pad 1): explicit im2col of the NCHW
 * bf16 image into [B*(H/2)*(W/2), 32] (27 taps*channels + 5 zero columns,
 * column = c*9 + ky*3 + kx, i.e. the conv weight's own
 * [cout, 3, 3, 3] flattening) for mg_gemm_bf16.                                */
int mg_stem_im2col_bf16(const mg_bf16* img_nchw, mg_bf16* out, int32_t B, int32_t H, int32_t W,
                        void* stream);

/* NF-ResNet-50 image encoder (encoder_name "nfresnet50"; reference magma/image_encoders.py:31-45 = timm nf_resnet50 minus its
 * classifier + AdaptiveAvgPool2d((1,1)), feeding the pooled ImagePrefix branch magma/image_prefix.py:17,67-72,96-101).  The
 * convolutions are mg_gemm_bf16; these are the pieces around them (csrc/nfnet.hip):
 *   weight_standardize  timm ScaledStdConv2d's weight transform: out[o, :] = (w[o, :] - mean_o) * rsqrt(var_o + eps) * gain[o] * scale
 *                       (biased variance over the cin*kh*kw fan-in; scale = gamma * fan_in^-0.5 from the caller), w [cout, cin, kh, kw]
 *                       bf16 -> out [cout, ldo] bf16 zero padded; to_khwc = 1 permutes the columns to (ky, kx, cin), the order of the
 *                       implicit-im2col 3x3 A loader (MG_A_CONV3X3), 0 keeps (cin, ky, kx) (1x1 convs, mg_im2col_nchw_bf16).
 *   im2col_nchw         img [B, C, H, W] bf16 -> rows [B*Ho*Wo, ldo], column (c*k + ky)*k + kx, zero padded (the 7x7 / stride-2 stem).
 *   maxpool3x3s2        MaxPool2d(3, stride 2, padding 1) on NHWC: x [B,H,W,C] -> y [B,(H-1)/2+1,(W-1)/2+1,C].
 *   subsample2          y[b,i,j,:] = x[b,2i,2j,:] (a stride-2, padding-1 3x3 conv = its stride-1 output at the even positions).
 *   relu_mean_rows      y[b, c] = mean_p relu(x[b, p, c]), x [B, HW, C]: final_act + AdaptiveAvgPool2d((1,1)).                          */
int mg_weight_standardize_bf16(const mg_bf16* w, const mg_bf16* gain, mg_bf16* out, int32_t cout, int32_t cin, int32_t kh,
                               int32_t kw, int64_t ldo, int32_t to_khwc, float scale, float eps, void* stream);
int mg_im2col_nchw_bf16(const mg_bf16* img, mg_bf16* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride,
                        int32_t pad, int32_t ldo, void* stream);
int mg_maxpool3x3s2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mg_subsample2_nhwc_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mg_relu_mean_rows_bf16(const mg_bf16* x, mg_bf16* y, int32_t B, int32_t HW, int32_t C, void* stream);
/* ... and their backward forms (training the nfresnet50 encoder, reference default freeze_img_encoder: false):
 *   weight_standardize_bwd  dwhat [cout, ldd] fp32 = gradient wrt the standardised weights in the weight's own (cin,ky,kx) order;
 *                           dw [cout, fan_in] += r (dn - mean(dn) - n mean(dn n)), dgain [cout] += scale sum(dwhat n),
 *                           n = (w - mean) r, r = rsqrt(var + eps), dn = dwhat * gain * scale   (fp32 accumulate into the caller's buffers);
 *   maxpool3x3s2_bwd        dx [B,H,W,C] from dy [B,Ho,Wo,C]: routed to the first maximum of each window (PyTorch's choice), as a gather;
 *   subsample2_bwd          dx[b,2i,2j,:] = dy[b,i,j,:], zero elsewhere;
 *   relu_mean_rows_bwd      dx[b,p,c] = x[b,p,c] > 0 ? g[b,c] / HW : 0.                                                                   */
int mg_weight_standardize_bwd_f32(const mg_bf16* w, const mg_bf16* gain, const float* dwhat, int64_t ldd, float* dw, float* dgain,
                                  int32_t cout, int32_t fan_in, float scale, float eps, float dmult /* dwhat is read times dmult */, void* stream);
int mg_maxpool3x3s2_bwd_nhwc_bf16(const mg_bf16* x, const mg_bf16* dy, mg_bf16* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mg_subsample2_bwd_nhwc_bf16(const mg_bf16* dy, mg_bf16* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mg_relu_mean_rows_bwd_bf16(const mg_bf16* x, const mg_bf16* g, mg_bf16* dx, int32_t B, int32_t HW, int32_t C, void* stream);

/* Data-parallel exchange step over RCCL / xGMI (csrc/comm.hip): the seam that replaces DeepSpeed's gradient reduction
 * (reference magma/train_loop.py:18-19 model_engine.backward / .step, train.py:103-111 deepspeed.initialize, magma/utils.py:26-34
 * reduce_losses).  One communicator per process (= per GPU; mg_comm_init binds the CURRENT HIP device):
 *   mg_comm_unique_id   rank 0 draws the 128-byte id, the caller hands it to every rank over any side channel;
 *   mg_comm_init        collective over all ranks; *comm_out = opaque handle;
 *   mg_comm_allreduce_sum  in-place SUM over the ranks, count elements of dtype MG_COMM_F32 / MG_COMM_BF16, enqueued on
 *                       `stream` (the mean of the reference's ZeRO-2 reduction is the 1/world folded into mg_adamw_*);
 *   mg_comm_broadcast   in-place from `root` (whole-model broadcast at engine construction; MG_COMM_BYTES = raw bytes);
 *   mg_comm_destroy.
 * RCCL is loaded with dlopen at the first call (the copy already in the process if there is one): MG_ERR_COMM if absent. */
#define MG_COMM_F32 0
#define MG_COMM_BF16 1
#define MG_COMM_BYTES 2
int mg_comm_unique_id(uint8_t* id128);
int mg_comm_init(void** comm_out, const uint8_t* id128, int32_t rank, int32_t world);
int mg_comm_allreduce_sum(void* comm, void* buf, int64_t count, int32_t dtype, void* stream);
int mg_comm_broadcast(void* comm, void* buf, int64_t count, int32_t dtype, int32_t root, void* stream);
int mg_comm_destroy(void* comm);

/* K20 (integer, exact): reference magma/utils.py:334-364 build_labels.
 * labels[b, :P] = -100; labels[b, P+t] = captions[b, t] up to and including
 * the first eos of the row; -100 after.  captions/labels [B, S] int64.       */
int mg_build_labels_i64(const int64_t* captions, int64_t* labels, int32_t B, int32_t S, int32_t P,
                        int64_t eos, void* stream);

/* K19: shifted cross-entropy of the LM call with labels (reference magma/magma.py:270-274; loss computed inside
 * the fork's GPTNeoForCausalLM.forward in fp32).  For each row r of `logits` [R, V] fp32 with target
 * tgt[r] (int64, -100 = ignore): loss_row[r] = logsumexp - logit[tgt] (0 if
 * ignored).  The caller passes already-shifted rows.  mg_ce_reduce produces
 * mean over valid rows into out[0] and the valid count into out[1].          */
int mg_ce_rows_f32(const float* logits, int64_t ld, const int64_t* tgt, float* loss_row,
                   int32_t R, int32_t V, void* stream);
int mg_ce_reduce_f32(const float* loss_row, const int64_t* tgt, int32_t R, float* out,
                     void* stream);

/* ===================== training path (backward + optimizer) =====================
 * Replaces model_engine.backward(loss) / model_engine.step() of reference magma/train_loop.py:15-19 (torch.autograd +
 * the DeepSpeed engine configured at magma/config.py:113-134).                                                  */

/* out[C,R] = in[R,C]^T (batched).  Feeds the wgrad GEMMs, which contract over the
 * row index M of activations: dW[N,K] = dY^T[N,M] * (X^T[K,M])^T.                 */
int mg_transpose_bf16(const mg_bf16* in, int64_t ld_in, int64_t bs_in, mg_bf16* out, int64_t ld_out,
                      int64_t bs_out, int32_t R, int32_t C, int32_t batch, void* stream);
/* The same transpose of ONE matrix that also accumulates colsum[c] += sum_r in[r][c] (fp32): the bias gradient and the weight-gradient
 * operand of a Linear (reference adapters.py:18-25 Linear layers; torch.autograd in the reference) from one pass over its output gradient. */
int mg_transpose_colsum_bf16(const mg_bf16* in, int64_t ld_in, mg_bf16* out, int64_t ld_out, int32_t R, int32_t C,
                             float* colsum, void* stream);

/* per-head transposes for the attention kernels, column-tiled: dst[(((b*H+h)*(ld/32) + s/32)*256 + d)*32 + s%32]
 * = src[b*sb + s*ss + h*sh + d]; ld = round_up(S,32), zero filled.                                              */
int mg_head_transpose_bf16(const mg_bf16* src, int64_t sb, int64_t ss, int64_t sh, mg_bf16* dst, int32_t ld,
                           int32_t B, int32_t H, int32_t S, void* stream);

/* out[n] += sum_m x[m][n] * (y ? y[m][n] : 1)  (bias / affine gradients, fp32 atomics). */
int mg_colsum_f32(const mg_bf16* x, int64_t ldx, const mg_bf16* y, int64_t ldy, float* out, int32_t M,
                  int32_t N, void* stream);

/* LayerNorm input gradient (+ optional residual add, optional xhat output).         */
int mg_layernorm_bwd_bf16(const mg_bf16* dy, int64_t lddy, const mg_bf16* x, int64_t ldx, const float* gamma,
                          const mg_bf16* res, int64_t ldr, mg_bf16* dx, int64_t lddx, mg_bf16* xhat,
                          int64_t ldxh, int32_t rows, int32_t d, float eps, void* stream);

/* dlogits = (softmax(logits) - onehot(tgt)) / n_valid, bf16, zero padded to ldo.
 * stats = output of mg_ce_reduce_f32 (stats[1] = number of valid targets).          */
int mg_ce_bwd_bf16(const float* logits, int64_t ld, const int64_t* tgt, const float* stats, mg_bf16* out,
                   int64_t ldo, int32_t R, int32_t V, void* stream);

/* inverse rotary on dq,dk and merge dq|dk|dv [B,H,S,256] -> dqkv [B*S, 3*H*256].    */
int mg_rotary_merge_bwd_bf16(const mg_bf16* dq, const mg_bf16* dk, const mg_bf16* dv, int32_t B, int32_t S,
                             int32_t H, int32_t rot_dim, const float* sin_t, const float* cos_t,
                             mg_bf16* dqkv, void* stream);

/* causal flash-attention backward, head dim 256 (recomputes P from q,k,lse).
 * q,k,v [B,H,S,256]; qt,kt,dOt [B,H,ld_t/32,256,32] (column-tiled transposes from mg_head_transpose_bf16,
 * ld_t = round_up(S,32), zero padded);
 * dO [B*S,H*256]; O [B*S, >= H*256] with row stride ld_o elements (ABI 2: the attention output may sit in a wider
 * [ctx | t] buffer, the operand of the [W_out | W_up] GEMM); lse [B,H,S]; D = fp32 workspace of 2*B*H*S floats ({-16 lse, -rowsum(dO o O)}
 * per query, written by the first launch).                                           */
int mg_attn_bwd_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                     const mg_bf16* kt, const mg_bf16* dO, const mg_bf16* dOt, const mg_bf16* O,
                     const float* lse, float* D, mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, int32_t B,
                     int32_t H, int32_t S, int32_t ld_t, int64_t ld_o, void* stream);

/* The same backward written straight into the gradient of the fused qkv projection (autograd through
 * GPTJAttention's rotary + head split, /root/reference call site magma/magma.py:263-276 -> HF modeling_gptj):
 * dqkv [B*S, 3*H*256] = [dq | dk | dv] per token, the inverse GPT-J rotary R(-theta_s) applied to the first
 * rot_dim columns of every dq and dk head (sin_t, cos_t fp32 [>= S, rot_dim/2]).  Equals mg_attn_bwd_bf16
 * followed by mg_rotary_merge_bwd_bf16 with one bf16 rounding instead of two.  dOt is a WORKSPACE
 * [B,H,ld_t/32,256,32] here: the first launch transposes dO into it while it computes D.               */
int mg_attn_bwd_merged_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, const mg_bf16* qt,
                            const mg_bf16* kt, const mg_bf16* dO, mg_bf16* dOt, const mg_bf16* O,
                            const float* lse, float* D, mg_bf16* dqkv, int32_t rot_dim, const float* sin_t,
                            const float* cos_t, int32_t B, int32_t H, int32_t S, int32_t ld_t, int64_t ld_o, void* stream);

/* ---- attention without transposed operand images (round 6, csrc/attention_tr.hip) -------------------------------------------------
 * The path replaced: reference magma/magma.py:263-276 -> HF GPTJAttention (rotary on q / k, causal softmax(q k^T / 16) v) and its
 * autograd.  The s-contraction operands (V^T in the forward; Q^T, dO^T, K^T in the backward) are read from the ROW images in LDS with
 * ds_read_b64_tr_b16, so no transposed tensor exists in HBM, and q / k / v are taken as rows of 256 at arbitrary strides:
 * position s of head (b, h) at  ptr + b stride_b + h stride_h + s ld_row  (elements; ld_row >= 256, all multiples of 8,
 * S ld_row 2 < 2^31).  [B,H,S,256]: ld_row 256, stride_h S 256, stride_b H S 256.  The fused qkv activation [B*S, 3 H 256] ITSELF:
 * q = qkv, k = qkv + H 256, v = qkv + 2 H 256, ld_row 3 H 256, stride_h 256, stride_b S 3 H 256 -- after mg_rotary_qk_inplace_bf16
 * there is no split pass and no copy of q, k or v.                                                                                   */
/* GPT-J rotary (interleaved pairs) applied in place to the first rot_dim columns of every q and k head of qkv [B*S, ld_qkv >= 3 H 256],
 * angles of position s = row % S (tables sin_t / cos_t fp32 [>= S, rot_dim / 2]); v is not touched.                                  */
int mg_rotary_qk_inplace_bf16(mg_bf16* qkv, int64_t ld_qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim,
                              const float* sin_t, const float* cos_t, void* stream);
/* causal flash-attention forward: out [B*S, >= H*256] bf16 at row stride ld_out (0 = H*256; % 8 == 0), lse [B,H,S] fp32 or NULL.
 * Same arithmetic as mg_attn_prefill_bf16 (fp32 online softmax, 1/16 scale).                                                          */
int mg_attn_fwd_rows_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, int64_t ld_row, int64_t stride_b,
                          int64_t stride_h, mg_bf16* out, int64_t ld_out, float* lse, int32_t B, int32_t H, int32_t S, void* stream);
/* backward: dO [B*S, H*256]; O [B*S, >= H*256] at row stride ld_o; lse [B,H,S]; D fp32 workspace of 2 B H S floats.  Output EITHER
 * dq, dk, dv [B,H,S,256] (dqkv NULL) OR dqkv [B*S, 3 H 256] = the gradient of the fused qkv projection with the inverse rotary applied
 * to the first rot_dim columns of every dq and dk head (dq, dk, dv NULL) -- as mg_attn_bwd_bf16 / mg_attn_bwd_merged_bf16.
 * dqkv8 / dqkv8_scales (or NULL): also the OCP MX e4m3 copy of dqkv, [B*S, 3 H 256] bytes + E8M0 scales of mg_mx_scale_bytes(B*S, 3 H 256)
 * bytes, bit for bit what mg_quantize_mx_fp8 makes of the bf16 dqkv -- the operand of the qkv dgrad's mg_gemm_mx_fp8 (BASELINE config[4]),
 * written from the epilogue's row pieces; dqkv itself may then be NULL (no bf16 copy).                                               */
int mg_attn_bwd_rows_bf16(const mg_bf16* q, const mg_bf16* k, const mg_bf16* v, int64_t ld_row, int64_t stride_b,
                          int64_t stride_h, const mg_bf16* dO, const mg_bf16* O, int64_t ld_o, const float* lse, float* D,
                          mg_bf16* dq, mg_bf16* dk, mg_bf16* dv, mg_bf16* dqkv, int32_t rot_dim, const float* sin_t,
                          const float* cos_t, int32_t B, int32_t H, int32_t S, uint8_t* dqkv8, uint8_t* dqkv8_scales, int32_t first_rows, void* stream);

/* CLIP trunk backward helpers */
int mg_avgpool2_bwd_nhwc_bf16(const mg_bf16* dy, const mg_bf16* gate, mg_bf16* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mg_mul_bf16(const mg_bf16* a, const mg_bf16* b, mg_bf16* out, int64_t n, void* stream);
/* torch.nn.GELU() (erf form) for adapters built with activation=nn.GELU (reference magma/adapters.py:11,20): y = gelu(x) when g is NULL,
 * y = g * gelu'(x) otherwise, over [rows, cols] bf16 with row strides (elements; cols % 8 == 0); in place allowed.  A pass of its own:
 * the erf polynomial is kept out of the GEMM epilogues (it cost the 256x256 kernels registers).                                      */
int mg_gelu_erf_bf16(const mg_bf16* x, int64_t ldx, const mg_bf16* g, int64_t ldg, mg_bf16* y, int64_t ldy, int32_t rows,
                     int32_t cols, void* stream);
int mg_scale_rows_acc_f32(float* dst, const float* src, int64_t ld_src, const float* row_scale, int32_t rows,
                          int32_t cols, void* stream);
int mg_add_gate_bf16(const mg_bf16* a, const mg_bf16* b, const mg_bf16* gate, mg_bf16* out, int64_t n, void* stream);
int mg_bn_param_grad_f32(const mg_bf16* g, const mg_bf16* y, const mg_bf16* sub, const float* gamma,
                         const float* beta, float* dgamma, float* dbeta, int32_t M, int32_t C, void* stream);
/* The same sums taken while g is transposed for the convolution's weight-gradient GEMM (out = g^T [C, ld_out], as mg_transpose_bf16;
 * y / sub in g's layout): one pass over g instead of two, from a grid of 64x64 tiles (ABI 6).                                   */
int mg_transpose_bn_param_grad_bf16(const mg_bf16* g, int64_t ld_in, mg_bf16* out, int64_t ld_out, int32_t R, int32_t C,
                                    const mg_bf16* y, const mg_bf16* sub, const float* gamma, const float* beta,
                                    float* dgamma, float* dbeta, void* stream);
/* trainable conv weight [Cout][Cin][k][k] (k = 1, 3) -> row-major GEMM operand, rows zero padded to ldo:
 *   mode 0: out[co][tap*Cin + ci] = w[co][ci][ky][kx]                       (forward, implicit-im2col order)
 *   mode 1: out[ci][tap*Cout + co] = bf16(w[co][ci][k-1-ky][k-1-kx] * scale[co])   (dgrad: flipped taps, BN scale folded)
 * (one launch instead of PyTorch's permute / flip / multiply / cast / pad copies, every step, for each of the 127 convs). */
int mg_conv_weight_relayout_bf16(const mg_bf16* w, const float* scale, mg_bf16* out, int64_t ldo, int32_t Cout, int32_t Cin,
                                 int32_t k, int32_t mode, void* stream);
/* frozen-statistics BatchNorm as a per-channel affine for the conv epilogues (one launch per BN and step):
 * scale = gamma / sqrt(var + eps), shift = beta - mean * scale; all fp32 [C].                                   */
int mg_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   float* scale, float* shift, int32_t C, void* stream);
/* The two calls above for MANY convolutions / BatchNorms in one launch each (ABI 6): the training engine re-derives the GEMM
 * operands of all 127 trunk convolutions (forward + dgrad layout) and the folded affine of their BatchNorms once per step --
 * 2 launches instead of ~380.  `jobs`: DEVICE array (caller-owned, 8-byte aligned), sorted by first_block; job i covers the
 * workgroups [first_block, first_block + ceil(rows * ldo / 256)) resp. [first_block, first_block + ceil(C / 256));
 * total_blocks = the end of the last job.  Same arithmetic, element for element, as the single-job entry points.
 * (reference: the convolutions / BatchNorms of magma/image_encoders.py:48-76's CLIP trunk, trained as in magma/magma.py:98-100) */
typedef struct mg_relayout_job {
  const mg_bf16* w;      /* [Cout][Cin][k][k] */
  const float* scale;    /* mode 1: [Cout] or NULL */
  mg_bf16* out;          /* rows x ldo */
  int64_t ldo;
  int32_t Cout, Cin, k, mode;
  int64_t first_block;
} mg_relayout_job;
typedef struct mg_bn_fold_job {
  const float* gamma; const float* beta; const float* mean; const float* var;
  float* scale; float* shift;
  float eps; int32_t C;
  int64_t first_block;
} mg_bn_fold_job;
int mg_conv_weight_relayout_batch(const mg_relayout_job* jobs, int32_t njobs, int64_t total_blocks, void* stream);
int mg_bn_fold_batch(const mg_bn_fold_job* jobs, int32_t njobs, int64_t total_blocks, void* stream);
int mg_im2col_t_bf16(const mg_bf16* x, mg_bf16* out, int64_t ldo, int32_t B, int32_t H, int32_t W, int32_t Cin, void* stream);

/* batch-statistics BatchNorm of the CLIP trunk in training (the reference's tower runs in train mode after its
 * first eval phase, reference train.py:164,182).  z = raw conv output [M, C] bf16 (M = B*H*W); sum / sumsq = per-channel sums
 * of z and z*z (mg_colsum_f32).  fold: scale = gamma * rstd, shift = beta - mean * scale, and running_mean / running_var
 * (may be NULL) updated like nn.BatchNorm2d (momentum, unbiased variance).  apply: y = [relu](z*scale + shift [+ res]).
 * bwd_dz: dz = gamma * rstd * (g - dbeta/M - xhat * dgamma/M) with xhat = (z - mean) * rstd, dgamma / dbeta = this call's
 * per-channel sums of g * xhat and g.                                                                                    */
int mg_bn_batch_fold_f32(const float* sum, const float* sumsq, const float* gamma, const float* beta, int64_t M, float eps,
                         float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                         float* rstd, int32_t C, void* stream);
int mg_bn_apply_bf16(const mg_bf16* z, const float* scale, const float* shift, const mg_bf16* res, int32_t relu, mg_bf16* y,
                     int64_t M, int32_t C, void* stream);
int mg_bn_bwd_dz_bf16(const mg_bf16* g, const mg_bf16* z, const float* mean, const float* rstd, const float* gamma,
                      const float* dgamma, const float* dbeta, mg_bf16* dz, int64_t M, int32_t C, void* stream);

/* optimizer (replaces DeepSpeed's fp16 ZeRO-2 step: global-norm clip + AdamW on fp32
 * master copies, reference train.py:96-101, config.py:124-134).                     */
int mg_sumsq_f32(const float* g, int64_t n, float* out, void* stream);
int mg_adamw_f32(float* p, float* m, float* v, const float* g, mg_bf16* p_bf16, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                 const float* norm_sq, float grad_scale, void* stream);
/* data-parallel exchange (reference train_loop.py:18-19 -> DeepSpeed ZeRO-2 reduces fp16 gradients; here bf16, 0.77 GB
 * per step for MAGMA_v1): the fp32 flat gradients are cast into bf16 buckets, summed over the ranks by RCCL
 * (torch.distributed, host side) and consumed as bf16 by the same fused clip + AdamW (g holds the SUM; grad_scale the
 * 1 / (gas * world) of the mean).  n % 4 == 0 for the cast.                                                          */
int mg_cast_f32_bf16(const float* src, mg_bf16* dst, int64_t n, void* stream);
int mg_sumsq_bf16(const mg_bf16* g, int64_t n, float* out, void* stream);
int mg_adamw_gbf16_f32(float* p, float* m, float* v, const mg_bf16* g, mg_bf16* p_bf16, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                       const float* norm_sq, float grad_scale, void* stream);

/* ---- fp8 attention forward (BASELINE config[4]: "fp8 MFMA path for GPT-J attention"; reference call site magma/magma.py:270-274) ----
 * mg_rotary_split_fp8: the training form of the rotary split (everything mg_rotary_split_train_bf16 writes except V^T: the backward
 * stays bf16) plus OCP MX e4m3 copies for the forward: q8 / k8 [B,H,S,256] with ONE power-of-two (E8M0) scale per token in
 * eq / ek [B,H,Sp] bytes, Sp = mg_attn_fp8_scale_stride(S); v8t [B,H,ceil(S/64),256,64] = V^T in 64-key tiles with one E8M0 per
 * (d, 32 keys) in sv8 [B,H,ceil(S/64),512] (layout [key block][d % 32][d / 32]); inside a tile the keys are stored in the order
 * the attention kernel's accumulators hold them (attention.hip: rotary_split_fp8_kernel).  qt / kt may be NULL (no backward);
 * q / k / v may be NULL together (forward only).
 * mg_attn_prefill_fp8: causal flash attention on v_mfma_scale_f32_32x32x64_f8f6f4 with those operands, fp32 softmax, P as
 * e4m3(16 p) with scale 2^-4; outputs as mg_attn_prefill_bf16 (out bf16 [B*S, >= H*256] at row stride ld_out % 8 == 0, lse fp32).
 * ABI 5: out8 / out8_scales (or NULL) = also the OCP MX e4m3 copy of out ([B*S, ld_out8 = ceil(H*256 / 128) * 128] bytes + E8M0 scales
 * of mg_mx_scale_bytes(B*S, H*256) bytes), bit for bit what mg_quantize_mx_fp8 makes of it: the operand of out_proj's MX GEMM.      */
int32_t mg_attn_fp8_scale_stride(int32_t S);
int mg_rotary_split_fp8(const mg_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t rot_dim, const float* sin_t,
                        const float* cos_t, mg_bf16* q, mg_bf16* k, mg_bf16* v, mg_bf16* qt, mg_bf16* kt, int32_t ld_t,
                        uint8_t* q8, uint8_t* k8, uint8_t* v8t, uint8_t* eq, uint8_t* ek, uint8_t* sv8, int32_t inplace, void* stream);
int mg_attn_prefill_fp8(const uint8_t* q8, const uint8_t* k8, const uint8_t* v8t, const uint8_t* eq, const uint8_t* ek,
                        const uint8_t* sv8, mg_bf16* out, int64_t ld_out, float* lse, int32_t B, int32_t H, int32_t S,
                        uint8_t* out8, int64_t ld_out8, uint8_t* out8_scales, void* stream);

/* A HIP stream whose kernels never run on `reserve` of the device's CUs (spread evenly; hipExtStreamCreateWithCUMask): the
 * training engine can run its compute on it so that the RCCL kernels of the gradient exchange (reference train_loop.py:18-19 /
 * DeepSpeed's overlapped reduce) always find free CUs instead of waiting for tile-GEMM workgroup boundaries.  Off by default
 * (MAGMA_DP_RESERVE_CUS).  The caller destroys the stream.                                                             */
int mg_stream_create_cu_mask(void** stream_out, int32_t reserve);
int mg_stream_destroy(void* stream);

/* ---- image preprocessing (reference magma/transforms.py:121-134) ----------------
 * One pass of Pillow's 8-bit antialiased resampling (ImagingResample, the arithmetic behind torchvision's
 * Resize(n, BICUBIC) on a PIL image): src is HWC uint8 RGB [H][W][3]; axis 1 resamples every row to
 * out_size pixels (dst [H][out_size][3]), axis 0 every column (dst [out_size][W][3]).  coeffs is
 * [out_size][ksize] int32 (22 fractional bits), bounds [out_size][2] = {first source index, count}; both
 * are built on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.  Bit-exact.   */
int mg_resample_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t out_size, int32_t axis,
                   const int32_t* coeffs, const int32_t* bounds, int32_t ksize, void* stream);
/* CenterCrop + ToTensor + Normalize: out[c][y][x] = (src[top+y][left+x][c] / 255 - mean[c]) / std[c] in
 * IEEE fp32 (CHW, n x n); mean3 / std3 are HOST pointers to 3 floats.                                   */
int mg_crop_normalize_f32(const uint8_t* src, int32_t H, int32_t W, int32_t top, int32_t left, int32_t n,
                          const float* mean3, const float* std3, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGMA_HIP_H */
