// attn32_device.h -- what the 32-row-wave attention kernels share (attention_fwd32.hip, attention_bwd32.hip): the 32x32x16 MFMA
// as inline-assembly statements with the accumulator pinned to one half of the register file, the row permutation, and the
// fragment bursts over the LDS images of attn_tile_device.h.
#pragma once
#include "common.h"
#include "attn_tile_device.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator

MG_DEV f32x16 mfma32(const bf16x8 a, const bf16x8 b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// the same with the accumulator pinned to the AGPR half of the register file: left to itself hipcc (ROCm 7.2) mixes the 256
// accumulator registers of the two gradient tiles with the operand fragments across both halves and spills ~750 registers per
// lane.  As an asm statement the MFMA is opaque to the hazard recogniser: an accumulate chain on the same registers needs no
// wait states, the operands are never written by the instruction in front (ds_read results arrive behind hipcc's own lgkmcnt
// wait, the packed P / dS operands are produced a phase earlier), and the epilogue reads the accumulators behind s_nop pads.
MG_DEV void mfma32a(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// ... and with the accumulator pinned to the VGPR half (S, dP: read by the softmax arithmetic).  hipcc gives a builtin MFMA's
// result AGPRs of its own choice under this register pressure -- on top of the 256 pinned ones, which it then shuffles through
// VGPRs every step.
MG_DEV void mfma32v(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// ... between VALU writes of a packed operand and the asm MFMA that reads it
MG_DEV void mfma_operand_ready(bf16x8& a, bf16x8& b) { asm volatile("s_nop 3" : "+v"(a), "+v"(b)); }
// The LAST MFMA of an accumulate chain carries its wait states itself (8-pass XDL op -> any other reader or writer of the
// result: 12+): whatever hipcc schedules behind the statement -- the softmax arithmetic, but also register copies of its own
// around a loop exit (seen: one accumulator register read right behind the loop, two gradient columns wrong) -- finds the
// result written.  The pad is issue time of THIS wave only; the matrix pipe is busy with the MFMA meanwhile.
MG_DEV void mfma32v_last(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(c) : "v"(a), "v"(b));
}
MG_DEV void mfma32a_last(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+a"(c) : "v"(a), "v"(b));
}
// B operand resident in the AGPR half (MFMA A / B operands may be AGPRs on gfx950): the per-wave constant fragments (Q in the
// forward) then cost no VGPRs at all -- given "v" operands, hipcc parks them in AGPRs anyway and copies four registers back in
// front of every MFMA.
MG_DEV void mfma32v_ba(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
MG_DEV void mfma32v_ba_last(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(c) : "v"(a), "a"(b));
}
MG_DEV void mfma32v0_ba(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
// ONE wait for a whole fragment burst (the four oldest of eight outstanding reads) in front of its four MFMAs: left alone hipcc
// emits a ladder lgkmcnt(7) .. (4), one s_waitcnt per MFMA -- 48 instructions per tile step of a wave that is bound by its
// instruction stream.  The builtin (not asm) so that hipcc's scoreboard sees it and drops its own.  gfx9 encoding: vmcnt 63,
// expcnt 7, lgkmcnt 4.
#define MG_LGKM4() __builtin_amdgcn_s_waitcnt(0xC47F)
#define MG_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
MG_DEV int perm32(int i) { return (i & 19) | ((i & 4) << 1) | ((i & 8) >> 1); }

constexpr int EP_ROW = 528;   // epilogue staging image: 512-B rows padded to 132 dwords (a lane group's 16 rows hit 16 bank quads)

// The same burst addressed as base ^ (ks << 5): chunk ((ks << 1) | hi) ^ sw of row R sits at byte R 512 + (((ks << 1) | hi) ^ sw) 16
// = (R 512 + (sw ^ hi) 16) ^ (ks << 5), so ONE lane constant (row_base32) + the stage offset + one v_xor per read replace the 16
// per-lane offsets hipcc otherwise hoists out of the tile loop (16 VGPRs these kernels do not have).  `off` = stage / image
// offset (a multiple of 512) + row_base32(...).
MG_DEV uint32_t row_base32(int R, int sw, int hi) { return (uint32_t)(R * 512 + ((sw ^ hi) << 4)); }
MG_DEV void rd_row4x(bf16x8 (&f)[4], const char* lds, uint32_t off, int g) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = *(const bf16x8*)(lds + (off ^ (uint32_t)((g * 4 + i) << 5)));
}
// fragment bursts of four: chunk ((ks << 1) | hi) of tile row R (swizzled), ks = g*4 .. g*4+3
MG_DEV void rd_row4(bf16x8 (&f)[4], const char* row, int g, int hi, int sw) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = *(const bf16x8*)(row + (((((g * 4 + i) << 1) | hi) ^ sw) << 4));
}
// T image: rows d = db*32 + l31 (64-byte rows), chunk ((ks2 << 1) | hi) ^ tsw; batch g = d-blocks 2g, 2g+1 x both k-steps
MG_DEV void rd_t4(bf16x8 (&f)[4], const char* tp, int g, int x) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = *(const bf16x8*)(tp + (g * 2 + (i >> 1)) * 2048 + ((((i & 1) << 1) ^ x) << 4));
}

// first MFMA of a chain: C = 0 (no zero-fill of the accumulator registers)
MG_DEV void mfma32v0(f32x16& c, const bf16x8 a, const bf16x8 b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}


// attention_fwd32.hip: the forward on 32-query waves (same operands as mg_attn_prefill_bf16)
int attn_prefill32_launch(const mg_bf16* q, const mg_bf16* kcache, const mg_bf16* vt, mg_bf16* out, int64_t ld_out, float* lse,
                          int B, int H, int S, int Smax, int vt_ld, float defer, hipStream_t s, const char* who);
