// common.h -- device helpers shared by the gfx950 kernels of libmagma_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/magma_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MG_DEV __device__ __forceinline__

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) --------------------
MG_DEV float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
MG_DEV uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// two floats -> packed bf16 pair: ONE v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, same values as f2bf)
typedef __attribute__((ext_vector_type(2))) float mg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 mg_bf16x2;
MG_DEV uint32_t pack2bf(float lo, float hi) {
  const mg_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_bf16x2));
}
MG_DEV float bflo(uint32_t w) { return __uint_as_float(w << 16); }
MG_DEV float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

MG_DEV float gelu_new_f(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))   (HF NewGELUActivation)
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  // 0.5 (1 + tanh u) = 1 / (1 + e^-2u): one v_exp_f32 + one v_rcp_f32 instead of the libm tanh
  const float u = k0 * (x + k1 * x * x * x);
  return x * __frcp_rn(1.0f + __expf(-2.0f * u));
}
MG_DEV float quick_gelu_grad_f(float x) {      // d/dx [x * sigmoid(1.702 x)]
  const float sg = 1.0f / (1.0f + __expf(-1.702f * x));
  return sg * (1.0f + 1.702f * x * (1.0f - sg));
}
MG_DEV float gelu_new_grad_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float sg = __frcp_rn(1.0f + __expf(-2.0f * k0 * (x + k1 * x * x * x)));   // 0.5 (1 + tanh u)
  const float t = 2.0f * sg - 1.0f;
  return sg + 0.5f * x * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * x * x);
}
MG_DEV float apply_act(float v, int act) {
  if (act == MG_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MG_ACT_GELU_NEW) return gelu_new_f(v);
  if (act == MG_ACT_QUICK_GELU) return v * __frcp_rn(1.0f + __expf(-1.702f * v));   // CLIP's QuickGELU: x * sigmoid(1.702 x)
  return v;
}

// ---- OCP MX e4m3 copy of 8 consecutive bf16 values held by one lane ---------------------------------------------------------
// The rule of mg_quantize_mx_fp8 (include/magma_hip.h): one E8M0 scale per 32 consecutive columns of a row = the smallest power of
// two that keeps the block maximum <= 448, elements rounded to e4m3 by v_cvt_pk_fp8_f32.  `w` = columns c .. c+7 of row `row`
// (c % 8 == 0); the FOUR lanes that hold one 32-column block must be consecutive lanes l, l^1, l^2, l^3 of the wave, all active.
// Writes the 8 bytes at qrow + c (qrow = q + row * ldq) and -- the lane with c % 32 == 0 -- the block's scale byte into the
// [chunk][block][row / 64][row % 16][(row % 64) / 16] layout the MX GEMM reads.  Used by the quantiser itself and by producer
// epilogues that hold whole rows (attention forward / backward), so their copies are the quantiser's bit for bit.
MG_DEV void mx_emit8(const u32x4 w, uint8_t* __restrict__ qrow, uint8_t* __restrict__ scales, int rgroups, int row, int c) {
  float f[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(w[i]); f[2 * i + 1] = bfhi(w[i]); }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  // floor(log2(amax)) from the exponent field (amax is a bf16 value: normal or zero; bf16 subnormals -> exponent field 0)
  const int ef = (int)((__float_as_uint(amax) >> 23) & 0xff);
  int e8 = amax > 0.f ? ef - 8 : 127;                          // E8M0 byte = floor(log2 amax) - 8 + 127
  e8 = max(0, min(254, e8));
  float inv = __uint_as_float((uint32_t)(254 - e8) << 23);   // 2^-(e8 - 127)
  if (amax * inv > 448.f) {    // the block maximum lies in (448, 512) 2^e: one exponent up instead of saturating it
    e8 = min(254, e8 + 1);
    inv = __uint_as_float((uint32_t)(254 - e8) << 23);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_fmed3f(f[i] * inv, -448.f, 448.f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  const u32x2 o = {(uint32_t)lo, (uint32_t)hi};
  *(u32x2*)(qrow + c) = o;
  if ((c & 31) == 0) {
    const int chunk = c >> 7, b = (c & 127) >> 5;
    scales[((((int64_t)chunk * 4 + b) * rgroups + (row >> 6)) * 16 + (row & 15)) * 4 + ((row & 63) >> 4)] = (uint8_t)e8;
  }
}

// ---- LDS-DMA: global -> LDS without a register round trip -----------------------
typedef const __attribute__((address_space(1))) void* mg_gptr_t;
typedef __attribute__((address_space(3))) void* mg_lptr_t;
MG_DEV void glds16(const void* g, char* lds_wave_base) {
  // 64 lanes x 16 B -> 1 KiB at lds_wave_base (wave-uniform) + lane*16; any swizzle goes into the SOURCE address
  __builtin_amdgcn_global_load_lds((mg_gptr_t)g, (mg_lptr_t)lds_wave_base, 16, 0, 0);
}
MG_DEV void glds4(const void* g, char* lds_wave_base) {   // 64 lanes x 4 B -> 256 B
  __builtin_amdgcn_global_load_lds((mg_gptr_t)g, (mg_lptr_t)lds_wave_base, 4, 0, 0);
}
// The same two as inline assembly.  hipcc models the builtin as a FLAT access that may touch LDS; while one is in flight
// (always, in a DMA ring that is only ever waited on with a counted vmcnt) its scoreboard calls both counters out of
// order and turns EVERY wait for an ordinary ds_read into s_waitcnt lgkmcnt(0) -- a full LDS round trip in front of each
// MFMA burst that only needed its oldest fragment.  Issued from assembly the DMA is invisible to that pass, the ds_read
// waits are counted again, and the ring is ordered by hand (MG_WAIT_VMCNT + barrier), as it already was.  A kernel that
// uses these must not also use the builtin forms (M0 is written behind the compiler's back).
MG_DEV void glds16a(const void* g, char* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(g), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mg_lptr_t)lds_wave_base)) : "memory");
}
// SGPR base + 32-bit per-lane BYTE offset: no 64-bit address VGPRs (which hipcc otherwise hoists out of the K loop, one
// pair per DMA piece, and spills when the kernel is at its register limit)
MG_DEV void glds16s(const void* base_uniform, uint32_t byte_off, char* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(byte_off), "s"(base_uniform),
                  "s"(__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mg_lptr_t)lds_wave_base)) : "memory");
}
// the same with the LDS destination as a plain 32-bit LDS address (lds_u32 of the dynamic-LDS base, once per kernel, plus
// integer offsets): a generic char* destination costs a 64-bit add and a null check (s_cmp_lg_u64 / s_cselect) per piece
MG_DEV uint32_t lds_u32(const void* p) { return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mg_lptr_t)p); }
MG_DEV void glds16au(const void* g, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_addr) : "memory");
}
// SGPR base + 32-bit per-lane byte offset + 32-bit LDS address
MG_DEV void glds16su(const void* base_uniform, uint32_t byte_off, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(byte_off), "s"(base_uniform), "s"(lds_addr) : "memory");
}
MG_DEV void glds4s(const void* base_uniform, uint32_t byte_off, char* lds_wave_base) {   // 64 lanes x 4 B, SGPR base + lane offset
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1"
               :: "v"(byte_off), "s"(base_uniform),
                  "s"(__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mg_lptr_t)lds_wave_base)) : "memory");
}
MG_DEV void glds4su(const void* base_uniform, uint32_t byte_off, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" :: "v"(byte_off), "s"(base_uniform), "s"(lds_addr) : "memory");
}
MG_DEV void glds4au(const void* g, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_addr) : "memory");
}
MG_DEV void glds4a(const void* g, char* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off"
               :: "v"(g), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(mg_lptr_t)lds_wave_base)) : "memory");
}
// an ordinary dword load issued from assembly (SGPR base + 32-bit per-lane byte offset): like the DMA forms above it is
// invisible to hipcc's wait-count pass, so a loop that retires its VMEM traffic with hand-counted vmcnt waits can carry it
// without the compiler draining the ring in front of the first use.  The CALLER guarantees that one of its own waits
// retires the load before dst is read (dst must not be touched in between).
MG_DEV void gld32s_async(uint32_t& dst, const void* base_uniform, uint32_t byte_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base_uniform) : "memory");
}
#define MG_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// vmcnt(0) through the BUILTIN (gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt untouched): unlike the inline-asm
// form this one updates hipcc's own scoreboard, so it does not re-wait (vmcnt(0), draining the DMA ring)
// at the first use of an ordinary load's result inside the pipelined loop.  Use it once, before the loop.
#define MG_WAIT_VMCNT0_TRACKED()            \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_waitcnt(0x0F70);    \
    asm volatile("" ::: "memory");         \
  } while (0)
// "these registers are needed now": hipcc waits (in its own scoreboard) for the ordinary loads that produce them.  Loads
// through const __restrict__ pointers are invariant loads and float across memory clobbers -- and across the builtin
// wait above -- so the only way to retire them BEFORE a pipelined loop is to consume their results there.
#define MG_USE8(f) asm volatile("" ::"v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"((f)[6]), "v"((f)[7]))
// workgroup barrier that does NOT drain in-flight LDS-DMA (a __syncthreads() would emit vmcnt(0)):
// retire this wave's LDS reads, barrier, and keep the compiler from moving LDS accesses across it
#define MG_BARRIER_KEEP_DMA()                          \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  } while (0)

// ---- XCD-aware workgroup order ---------------------------------------------------
// The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs (each with a private
// 4 MiB L2).  This bijection turns the hardware id into a logical index such that every XCD owns
// ONE contiguous run of logical indices: neighbours in logical order share an L2.
MG_DEV int xcd_contiguous_index(int bid, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// ---- reductions over the four lanes {l, l^16, l^32, l^48} that share an MFMA row / column -----------------
// v_permlane32_swap / v_permlane16_swap (new on gfx950) exchange half-waves / odd-even rows between two VGPRs in
// one VALU op each: no LDS round trip (ds_bpermute costs two in the attention kernel's serial softmax chain).
MG_DEV float quad_rows_max(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t v = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
MG_DEV float quad_rows_sum(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t v = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- wave / block reductions (wave = 64 lanes) -------------------------------
MG_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
MG_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host-side error plumbing -------------------------------------------------
void mg_set_error(const char* fmt, ...);
#define MG_FAIL(code, ...)      \
  do {                          \
    mg_set_error(__VA_ARGS__);  \
    return (code);              \
  } while (0)
#define MG_CHECK_LAUNCH()                                               \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) MG_FAIL(MG_ERR_HIP, "%s: %s", __func__, hipGetErrorString(e__)); \
  } while (0)
#define MG_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)
// Raise a kernel's dynamic-LDS limit once per (kernel, device): thread-safe, and per device because
// hipFuncSetAttribute acts on the current device's copy of the function (capi.hip).
int mg_allow_dynamic_lds(const void* fn, int bytes, const char* who);
