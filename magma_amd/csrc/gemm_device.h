// gemm_device.h -- device code shared by the GEMM translation units: the fused epilogue and the
// body of the decode weight-streaming GEMV (so that it can be co-launched with other workgroup kinds).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// fp8 (OCP e4m3) MFMA: v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales -- the only fp8 form that runs
// at twice the bf16 rate on gfx950 (the non-scaled 16x16x32 fp8 MFMA runs at the bf16 rate).  A lane supplies 32
// consecutive-in-its-own-order k bytes: here the two 16-byte fragments a bf16 kernel would feed to two 16x16x32
// steps.  Both operands use the same k order, so the dot product is unaffected.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
MG_DEV f32x4 mfma_fp8_k128(i32x8 a, i32x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A: e4m3*/, 0 /*B: e4m3*/, 0, 0x7F7F7F7F /*2^0*/, 0, 0x7F7F7F7F);
}
// the same with the MX block scales of the two operands.  sa / sb: one dword per lane holding the E8M0 bytes this lane supplies
// for up to four fragments (byte t = fragment t: mg_quantize_mx_fp8's scale layout); OA / OB select the byte (the MFMA's
// op_sel field, an immediate).  The hardware reads the scale of block b of row r from lane r + 16 b; a lane's own 32 operand
// bytes are k = 16 q .. + 15 and 64 + 16 q .. + 15 (q = lane >> 4): the chunk's plain byte order under the kernels' loaders.
template <int OA, int OB>
MG_DEV f32x4 mfma_mx_k128_sel(i32x8 a, uint32_t sa, i32x8 b, uint32_t sb, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)sa, OB, (int)sb);
}
MG_DEV f32x4 mfma_mx_k128(i32x8 a, int oa, uint32_t sa, i32x8 b, int ob, uint32_t sb, f32x4 c) {
  // oa / ob are compile-time constants at every call site (unrolled fragment loops): the switch folds away
  switch (oa * 4 + ob) {
#define MG_MX_CASE(A_, B_) case A_ * 4 + B_: return mfma_mx_k128_sel<A_, B_>(a, sa, b, sb, c);
    MG_MX_CASE(0, 0) MG_MX_CASE(0, 1) MG_MX_CASE(0, 2) MG_MX_CASE(0, 3)
    MG_MX_CASE(1, 0) MG_MX_CASE(1, 1) MG_MX_CASE(1, 2) MG_MX_CASE(1, 3)
    MG_MX_CASE(2, 0) MG_MX_CASE(2, 1) MG_MX_CASE(2, 2) MG_MX_CASE(2, 3)
    MG_MX_CASE(3, 0) MG_MX_CASE(3, 1) MG_MX_CASE(3, 2) MG_MX_CASE(3, 3)
#undef MG_MX_CASE
  }
  return c;
}

// ---------------------------------------------------------------------------
// shared epilogue: 4 consecutive n of row m.  Split in two so that row-walking callers load the
// per-column vectors (BatchNorm scale, bias) once.
// ---------------------------------------------------------------------------
template <int W> struct EpiColsW { float sc[W], bi[W]; };
typedef EpiColsW<4> EpiCols;

template <int W>
MG_DEV void epilogue_cols(const mg_epilogue& ep, int n, int N, EpiColsW<W>& c) {
#pragma unroll
  for (int r = 0; r < W; ++r) { c.sc[r] = 1.f; c.bi[r] = 0.f; }
  if (n + W - 1 < N) {
#pragma unroll
    for (int g = 0; g < W; g += 4) {
      if (ep.scale) { const float4 t = *(const float4*)(ep.scale + n + g); c.sc[g] = t.x; c.sc[g + 1] = t.y; c.sc[g + 2] = t.z; c.sc[g + 3] = t.w; }
      if (ep.bias)  { const float4 t = *(const float4*)(ep.bias + n + g);  c.bi[g] = t.x; c.bi[g + 1] = t.y; c.bi[g + 2] = t.z; c.bi[g + 3] = t.w; }
    }
  } else {
    for (int r = 0; r < W; ++r) if (n + r < N) {
      if (ep.scale) c.sc[r] = ep.scale[n + r];
      if (ep.bias) c.bi[r] = ep.bias[n + r];
    }
  }
}

// ---- cross-workgroup hand-offs inside ONE launch (persistent decode step, decode_mega_kernel) ----
// A CU's vector L1 is never refreshed by another CU's stores and the per-XCD L2s are not coherent with each other
// (MI355X_MICROARCH.md, inter-workgroup visibility).  Data produced and consumed by different workgroups of the same
// launch therefore moves through 8-byte agent-scope relaxed atomics on BOTH sides (global_load/store_dwordx2 sc1: L1
// bypassed, written through), ordered by a completion counter: the valid form "{8-B agent atomics both sides}".
// (hipcc follows every __hip_atomic_load with a full s_waitcnt vmcnt(0), which would serialise the weight stream around each
// activation fragment, so the loads are written as asm: `sc1` = the agent-scope form, and ONE wait per batch of loads.)
MG_DEV u32x2 ld8_coh(const void* p) {
  u32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
MG_DEV void st8_coh(void* p, u32x2 w) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)w[0] | ((unsigned long long)w[1] << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
MG_DEV u32x4 ld16_coh(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// up to three 8-byte rows (the residual operands of an epilogue) with one wait
MG_DEV void ld8x3_coh(const void* p0, const void* p1, const void* p2, u32x2& a, u32x2& b, u32x2& c) {
  asm volatile("global_load_dwordx2 %0, %3, off sc1\n\tglobal_load_dwordx2 %1, %4, off sc1\n\t"
               "global_load_dwordx2 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
// eight consecutive 64-byte-spaced 16-byte fragments (one chunk of activation k-steps) with one wait; the wait also
// retires the weight loads issued before it -- the MFMAs that follow need both
MG_DEV void ld16x8_coh(const void* p, u32x4 (&v)[8]) {
  asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:64 sc1\n\t"
               "global_load_dwordx4 %2, %8, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:192 sc1\n\t"
               "global_load_dwordx4 %4, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:320 sc1\n\t"
               "global_load_dwordx4 %6, %8, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:448 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
               : "v"(p) : "memory");
}

// W consecutive bf16 of a row <-> floats (W = 4: one 8-B access, W = 8: one 16-B access; p must be W*2-byte aligned)
template <int W, bool COH = false>
MG_DEV void load_bf16_row(const mg_bf16* p, float* a) {
  if constexpr (COH) {
    static_assert(W == 4, "coherent rows are 8-byte accesses");
    const u32x2 w = ld8_coh(p);
    a[0] = bflo(w[0]); a[1] = bfhi(w[0]); a[2] = bflo(w[1]); a[3] = bfhi(w[1]);
  } else if constexpr (W == 8) {
    const u32x4 w = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[2 * i] = bflo(w[i]); a[2 * i + 1] = bfhi(w[i]); }
  } else {
    const u32x2 w = *(const u32x2*)p;
    a[0] = bflo(w[0]); a[1] = bfhi(w[0]); a[2] = bflo(w[1]); a[3] = bfhi(w[1]);
  }
}
template <int W, bool NT, bool COH = false>
MG_DEV void store_bf16_row(mg_bf16* p, const float* o) {
  if constexpr (COH) {
    static_assert(W == 4, "coherent rows are 8-byte accesses");
    u32x2 w; w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]);
    st8_coh(p, w);
  } else if constexpr (W == 8) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack2bf(o[2 * i], o[2 * i + 1]);
    if (NT) __builtin_nontemporal_store(w, (u32x4*)p); else *(u32x4*)p = w;
  } else {
    u32x2 w; w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]);
    if (NT) __builtin_nontemporal_store(w, (u32x2*)p); else *(u32x2*)p = w;
  }
}

// Residual / aux operands of one full row piece, loaded ahead of the arithmetic: epilogue_rows fetches them for several
// rows before it stores anything (output and residual pointers may alias as far as the compiler knows, so a load written
// after a store is never hoisted above it -- one row in flight, 2 TB/s on a three-residual epilogue).
// (kept as raw bf16 words: W/2 registers per operand and row)
template <int W> struct EpiPre { uint32_t aux[W / 2]; uint32_t res[3][W / 2]; };
template <int W>
MG_DEV void epi_raw_load(const mg_bf16* p, uint32_t (&w)[W / 2]) {
  if constexpr (W == 8) { const u32x4 t = *(const u32x4*)p; w[0] = t[0]; w[1] = t[1]; w[2] = t[2]; w[3] = t[3]; }
  else { const u32x2 t = *(const u32x2*)p; w[0] = t[0]; w[1] = t[1]; }
}
template <int W>
MG_DEV void epilogue_prefetch(const mg_epilogue& ep, int m, int n, EpiPre<W>& p) {
#pragma unroll
  for (int r = 0; r < W / 2; ++r) { p.aux[r] = 0u; p.res[0][r] = 0u; p.res[1][r] = 0u; p.res[2][r] = 0u; }
  if (ep.aux_mode != MG_AUX_NONE) epi_raw_load<W>(ep.aux + (int64_t)m * ep.ldaux + n, p.aux);
  if (ep.res0) epi_raw_load<W>(ep.res0 + (int64_t)m * ep.ldr + n, p.res[0]);
  if (ep.res1) epi_raw_load<W>(ep.res1 + (int64_t)m * ep.ldr + n, p.res[1]);
  if (ep.res2) epi_raw_load<W>(ep.res2 + (int64_t)m * ep.ldr + n, p.res[2]);
}

// v[W] = accumulators of columns n .. n+W-1 of row m.  NT: non-temporal output stores (large outputs
// that nobody re-reads soon: keeps the L2 for the operand panels and streams the tile out).
// FULL: the caller guarantees n + W <= N (interior column tile): the per-element tail paths are not even compiled -- they are
// most of the instructions of an epilogue, and a tile's epilogue runs once per ~100 us, from a cold instruction cache.
template <int W, bool NT, bool COH, bool PRE, bool FULL = false>
MG_DEV void epilogue_apply_impl(const mg_epilogue& ep, const EpiColsW<W>& c, int m, int n, const float* v, int N,
                                const EpiPre<W>& pre) {
  const bool full = FULL || (n + W - 1 < N);
  float o[W];
#pragma unroll
  for (int r = 0; r < W; ++r) o[r] = v[r] * c.sc[r] + c.bi[r];
  if (ep.C2) {  // pre-activation copy for the backward pass
    mg_bf16* cp = ep.C2 + (int64_t)m * ep.ldc2 + n;
    if (full) store_bf16_row<W, NT>(cp, o);
    else for (int r = 0; r < W; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
  }
  const int act = n >= ep.act_n0 ? ep.act : MG_ACT_NONE;      // act_n0 % 8 == 0: a lane's W columns are on one side
#pragma unroll
  for (int r = 0; r < W; ++r) o[r] = apply_act(o[r], act);
  float ax[W];
#pragma unroll
  for (int r = 0; r < W; ++r) ax[r] = 1.f;
  if (ep.aux_mode != MG_AUX_NONE) {
    const mg_bf16* ap = ep.aux + (int64_t)m * ep.ldaux + n;
    float a[W];
#pragma unroll
    for (int r = 0; r < W; ++r) a[r] = 0.f;
    if constexpr (PRE) {
#pragma unroll
      for (int r = 0; r < W / 2; ++r) { a[2 * r] = bflo(pre.aux[r]); a[2 * r + 1] = bfhi(pre.aux[r]); }
    } else if (full) load_bf16_row<W>(ap, a);
    else for (int r = 0; r < W; ++r) if (n + r < N) a[r] = bf2f(ap[r]);
#pragma unroll
    for (int r = 0; r < W; ++r)
      ax[r] = ep.aux_mode == MG_AUX_RELU_GATE ? (a[r] > 0.f ? 1.f : 0.f)
            : ep.aux_mode == MG_AUX_GELU_GRAD ? gelu_new_grad_f(a[r])
            : ep.aux_mode == MG_AUX_QUICK_GELU_GRAD ? quick_gelu_grad_f(a[r]) : a[r];
    if (!ep.aux_after) {
#pragma unroll
      for (int r = 0; r < W; ++r) o[r] *= ax[r];
    }
  }
  const mg_bf16* rs[3] = {ep.res0, ep.res1, ep.res2};
  if constexpr (COH) {       // all residual rows in flight together, one wait (W == 4: 8-byte rows)
    if (full && (rs[0] || rs[1] || rs[2])) {
      const mg_bf16* any = rs[0] ? rs[0] : (rs[1] ? rs[1] : rs[2]);
      u32x2 w[3];
      ld8x3_coh((rs[0] ? rs[0] : any) + (int64_t)m * ep.ldr + n, (rs[1] ? rs[1] : any) + (int64_t)m * ep.ldr + n,
                (rs[2] ? rs[2] : any) + (int64_t)m * ep.ldr + n, w[0], w[1], w[2]);
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (rs[t]) { o[0] += bflo(w[t][0]); o[1] += bfhi(w[t][0]); o[2] += bflo(w[t][1]); o[3] += bfhi(w[t][1]); }
      rs[0] = rs[1] = rs[2] = nullptr;
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (rs[t]) {
      const mg_bf16* rp = rs[t] + (int64_t)m * ep.ldr + n;
      if constexpr (PRE) {
#pragma unroll
        for (int r = 0; r < W / 2; ++r) { o[2 * r] += bflo(pre.res[t][r]); o[2 * r + 1] += bfhi(pre.res[t][r]); }
      } else if (full) {
        float a[W];
        load_bf16_row<W, COH>(rp, a);
#pragma unroll
        for (int r = 0; r < W; ++r) o[r] += a[r];
      } else {
        for (int r = 0; r < W; ++r) if (n + r < N) o[r] += bf2f(rp[r]);
      }
    }
  }
  if (ep.aux_mode != MG_AUX_NONE && ep.aux_after) {
#pragma unroll
    for (int r = 0; r < W; ++r) o[r] *= ax[r];
  }
  if (ep.act_after == MG_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < W; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
  }
  if (ep.out_f32) {
    float* cp = (float*)ep.C + (int64_t)m * ep.ldc + n;
    if (full) {
#pragma unroll
      for (int g = 0; g < W; g += 4) {
        f32x4 w = {o[g], o[g + 1], o[g + 2], o[g + 3]};
        if (ep.accumulate) w += *(const f32x4*)(cp + g);       // weight gradients: += into the fp32 gradient buffer
        if (NT) __builtin_nontemporal_store(w, (f32x4*)(cp + g)); else *(f32x4*)(cp + g) = w;
      }
    } else for (int r = 0; r < W; ++r) if (n + r < N) cp[r] = ep.accumulate ? cp[r] + o[r] : o[r];
  } else {
    mg_bf16* cp = (mg_bf16*)ep.C + (int64_t)m * ep.ldc + n;
    if (full) store_bf16_row<W, NT, COH>(cp, o);
    else for (int r = 0; r < W; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
  }
}

template <int W, bool NT, bool COH = false, bool FULL = false>
MG_DEV void epilogue_apply(const mg_epilogue& ep, const EpiColsW<W>& c, int m, int n, const float* v, int N) {
  const EpiPre<W> none{};
  epilogue_apply_impl<W, NT, COH, false, FULL>(ep, c, m, n, v, N, none);
}

template <bool COH = false>
MG_DEV void epilogue_store4(const mg_epilogue& ep, int m, int n, f32x4 v, int N) {
  if (n >= N) return;
  if constexpr (!COH) {
    // The common decode / split-K fix-up case -- four columns inside the matrix, no aux operand, no pre-activation copy -- as
    // one short straight-line block (same arithmetic and order as epilogue_apply_impl).  A weight-streaming GEMV runs its
    // epilogue ONCE per workgroup, from a cold instruction cache: what counts is how few cache lines the taken path touches.
    if (n + 3 < N && ep.aux_mode == MG_AUX_NONE && !ep.C2) {
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (ep.scale) sc = *(const f32x4*)(ep.scale + n);
      if (ep.bias) bi = *(const f32x4*)(ep.bias + n);
      u32x2 r0 = {0u, 0u}, r1 = {0u, 0u}, r2 = {0u, 0u};
      if (ep.res0) r0 = *(const u32x2*)(ep.res0 + (int64_t)m * ep.ldr + n);
      if (ep.res1) r1 = *(const u32x2*)(ep.res1 + (int64_t)m * ep.ldr + n);
      if (ep.res2) r2 = *(const u32x2*)(ep.res2 + (int64_t)m * ep.ldr + n);
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = v[r] * sc[r] + bi[r];
      if (n >= ep.act_n0 && ep.act != MG_ACT_NONE) {
        if (ep.act == MG_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
        } else if (ep.act == MG_ACT_GELU_NEW) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = gelu_new_f(o[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = apply_act(o[r], ep.act);
        }
      }
      if (ep.res0) { o[0] += bflo(r0[0]); o[1] += bfhi(r0[0]); o[2] += bflo(r0[1]); o[3] += bfhi(r0[1]); }
      if (ep.res1) { o[0] += bflo(r1[0]); o[1] += bfhi(r1[0]); o[2] += bflo(r1[1]); o[3] += bfhi(r1[1]); }
      if (ep.res2) { o[0] += bflo(r2[0]); o[1] += bfhi(r2[0]); o[2] += bflo(r2[1]); o[3] += bfhi(r2[1]); }
      if (ep.act_after == MG_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
      }
      if (ep.out_f32) {
        f32x4* cp = (f32x4*)((float*)ep.C + (int64_t)m * ep.ldc + n);
        f32x4 w = {o[0], o[1], o[2], o[3]};
        if (ep.accumulate) w += *cp;
        *cp = w;
      } else {
        u32x2 w; w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]);
        *(u32x2*)((mg_bf16*)ep.C + (int64_t)m * ep.ldc + n) = w;
      }
      return;
    }
  }
  EpiCols c;
  epilogue_cols<4>(ep, n, N, c);
  const float vv[4] = {v[0], v[1], v[2], v[3]};
  epilogue_apply<4, false, COH>(ep, c, m, n, vv, N);
}

// 16-byte accesses need every row start 16-byte aligned
MG_DEV bool epilogue_wide_ok(const mg_epilogue& ep) {
  return !((ep.ldc & 7) | (ep.C2 ? (ep.ldc2 & 7) : 0) | (ep.aux_mode != MG_AUX_NONE ? (ep.ldaux & 7) : 0) |
           ((ep.res0 || ep.res1 || ep.res2) ? (ep.ldr & 7) : 0));
}

// Tile epilogue through LDS.  The MFMA accumulators of a workgroup tile were parked in LDS as fp32
// rows of NCOLS columns (row stride ROWB bytes; ROWB % 128 == 16 keeps the 8-lane groups of the
// fragment-shaped ds_write_b128 on distinct bank slots).  Here every wave walks whole rows: NCOLS/4
// lanes cover one row with W = 4 or 8 consecutive columns each, so residual / aux reads and the output
// stores are full contiguous lines (8 or 16 bytes per lane), and the code is one small rolled loop instead of one inlined epilogue
// per accumulator (which made the GEMM kernels > 20k instructions, mostly instruction-cache misses).
// Tile row r is global row  m_base + (r >> 6) * hi_stride + (r & 63).
// epilogue_rows_c: the per-column vectors `c` (epilogue_cols of this lane's W columns) come from the caller, so that a tile
// walked in several passes loads them ONCE -- a global load inside a later pass makes hipcc wait for vmcnt(0), and on this
// chip that also waits for the acknowledgement of every store of the pass before.
// LOADS = false (caller's promise: no aux operand, no residual, and rows <= 8 * step): the walk is compiled without a single
// global load -- see `iteration` below.
// Q8: the build that can also write the MX e4m3 copy (mg_epilogue.C8) -- its own instantiation, so that the kernels that never
// write one keep their register allocation (the extra live values cost the bf16 256x256 kernel four spilled registers).
template <int NCOLS, int ROWB, int W, bool NT, bool FULL = false, bool LOADS = true, bool Q8 = false>
MG_DEV void epilogue_rows_c(const mg_epilogue& ep, const EpiColsW<W>& c, const char* lds, int rows, int nwaves, int wave, int lane,
                            int m_base, int hi_stride, int n0, int M, int N, const float* row_scale = nullptr) {
  constexpr int LPR = NCOLS / W, RPI = 64 / LPR;     // lanes per row, rows per wave-iteration
  const int cl = lane % LPR;
  const int n = n0 + cl * W;
  if (n >= N) return;
  const int step = nwaves * RPI;
  int r = wave * RPI + lane / LPR;
  auto row_m = [&](int rr) { return m_base + (rr >> 6) * hi_stride + (rr & 63); };
  auto lds_row = [&](int rr, float (&v)[W]) {
#pragma unroll
    for (int g = 0; g < W; g += 4) {
      const f32x4 t = *(const f32x4*)(lds + rr * ROWB + (cl * W + g) * 4);
      v[g] = t[0]; v[g + 1] = t[1]; v[g + 2] = t[2]; v[g + 3] = t[3];
    }
  };
  if (FULL || n + W - 1 < N) {
    // Lanes whose W columns are all inside the matrix (every lane of an interior tile): TWO rows per iteration, and the
    // epilogue's run-time options are branched on once per iteration -- uniform branches around straight-line code for all
    // two rows -- not once per row and element.  (The per-row form of this loop was ~2 300 instructions of divergent
    // control flow per iteration and took 5 us per 128-row pass of a 256x256 tile; in-kernel stamps, tools/kbench.py stamps.)
    // Order of the arithmetic as in epilogue_apply_impl: (acc * row_scale) * scale + bias -> [C2] -> act -> aux (before) ->
    // residuals 0, 1, 2 -> aux (after) -> trailing ReLU -> store.
    constexpr int R = 2, H = W / 2;      // (four rows: 195 registers in the loop alone -- spills next to the live accumulator half)
    const bool act_on = n >= ep.act_n0;              // act_n0 % 8 == 0: a lane's W columns are on one side
    // LOADS = false: the iteration contains no global load at all (no aux, no residual; the row scales come in as values).
    // That matters beyond the loads themselves: with a load anywhere in the loop hipcc guards the first use with
    // s_waitcnt vmcnt(0), which on this chip also waits for the acknowledgement of the PREVIOUS iteration's stores.
    auto iteration = [&](int r, const float* rsc_in) {
      int m[R], mc[R], rr[R];
      bool ok[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int q = r + k * step;
        rr[k] = min(q, rows - 1);
        m[k] = row_m(rr[k]);
        ok[k] = q < rows && m[k] < M;
        mc[k] = min(m[k], M - 1);
      }
      uint32_t ax[R][H], rs0[R][H], rs1[R][H], rs2[R][H];
      float rsc[R];
      const bool has_aux = LOADS && ep.aux_mode != MG_AUX_NONE;
      const bool has_r0 = LOADS && ep.res0, has_r1 = LOADS && ep.res1, has_r2 = LOADS && ep.res2;
      if (has_aux) {
#pragma unroll
        for (int k = 0; k < R; ++k) epi_raw_load<W>(ep.aux + (int64_t)mc[k] * ep.ldaux + n, ax[k]);
      }
      if (has_r0) {
#pragma unroll
        for (int k = 0; k < R; ++k) epi_raw_load<W>(ep.res0 + (int64_t)mc[k] * ep.ldr + n, rs0[k]);
      }
      if (has_r1) {
#pragma unroll
        for (int k = 0; k < R; ++k) epi_raw_load<W>(ep.res1 + (int64_t)mc[k] * ep.ldr + n, rs1[k]);
      }
      if (has_r2) {
#pragma unroll
        for (int k = 0; k < R; ++k) epi_raw_load<W>(ep.res2 + (int64_t)mc[k] * ep.ldr + n, rs2[k]);
      }
      if (row_scale) {
#pragma unroll
        for (int k = 0; k < R; ++k) rsc[k] = LOADS ? row_scale[mc[k]] : rsc_in[k];
      }
      float o[R][W];
#pragma unroll
      for (int k = 0; k < R; ++k) lds_row(rr[k], o[k]);
      if (row_scale) {   // fp8 operands: per-row activation scale (the per-column weight scale is ep.scale)
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int g = 0; g < W; ++g) o[k][g] *= rsc[k];
      }
#pragma unroll
      for (int k = 0; k < R; ++k)
#pragma unroll
        for (int g = 0; g < W; ++g) o[k][g] = o[k][g] * c.sc[g] + c.bi[g];
      if (ep.C2) {       // pre-activation copy for the backward pass
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (ok[k]) store_bf16_row<W, NT>(ep.C2 + (int64_t)m[k] * ep.ldc2 + n, o[k]);
      }
      if (ep.act == MG_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int g = 0; g < W; ++g) o[k][g] = act_on ? (o[k][g] > 0.f ? o[k][g] : 0.f) : o[k][g];
      } else if (ep.act == MG_ACT_GELU_NEW) {
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int g = 0; g < W; ++g) { const float t = gelu_new_f(o[k][g]); o[k][g] = act_on ? t : o[k][g]; }
      } else if (ep.act == MG_ACT_QUICK_GELU) {
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int g = 0; g < W; ++g) { const float t = apply_act(o[k][g], MG_ACT_QUICK_GELU); o[k][g] = act_on ? t : o[k][g]; }
      }
      auto aux_mul = [&]() {
        // o *= f(aux): the mode is uniform, one straight-line block per mode
#define MG_AUX_BLOCK(F_)                                                                  \
        _Pragma("unroll") for (int k = 0; k < R; ++k)                                     \
          _Pragma("unroll") for (int h = 0; h < H; ++h) {                                 \
            const float a0 = bflo(ax[k][h]), a1 = bfhi(ax[k][h]);                         \
            o[k][2 * h] *= F_(a0); o[k][2 * h + 1] *= F_(a1);                             \
          }
#define MG_F_GATE(a_) ((a_) > 0.f ? 1.f : 0.f)
#define MG_F_ID(a_) (a_)
        if (ep.aux_mode == MG_AUX_RELU_GATE) { MG_AUX_BLOCK(MG_F_GATE) }
        else if (ep.aux_mode == MG_AUX_GELU_GRAD) { MG_AUX_BLOCK(gelu_new_grad_f) }
        else if (ep.aux_mode == MG_AUX_QUICK_GELU_GRAD) { MG_AUX_BLOCK(quick_gelu_grad_f) }
        else { MG_AUX_BLOCK(MG_F_ID) }
#undef MG_F_ID
#undef MG_F_GATE
#undef MG_AUX_BLOCK
      };
      if (has_aux && !ep.aux_after) aux_mul();
#define MG_RES_ADD(RS_)                                                                   \
      _Pragma("unroll") for (int k = 0; k < R; ++k)                                       \
        _Pragma("unroll") for (int h = 0; h < H; ++h) { o[k][2 * h] += bflo(RS_[k][h]); o[k][2 * h + 1] += bfhi(RS_[k][h]); }
      if (has_r0) { MG_RES_ADD(rs0) }
      if (has_r1) { MG_RES_ADD(rs1) }
      if (has_r2) { MG_RES_ADD(rs2) }
#undef MG_RES_ADD
      if (has_aux && ep.aux_after) aux_mul();
      if (ep.act_after == MG_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < R; ++k)
#pragma unroll
          for (int g = 0; g < W; ++g) o[k][g] = o[k][g] > 0.f ? o[k][g] : 0.f;
      }
      if constexpr (W == 8 && Q8) {
        if (ep.C8) {   // OCP MX copy (mg_epilogue.C8): the arithmetic of quantize_mx_fp8_kernel on the bf16-rounded values
#pragma unroll
          for (int k = 0; k < R; ++k) {
            float f[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              const uint32_t pk = pack2bf(o[k][2 * h], o[k][2 * h + 1]);
              f[2 * h] = bflo(pk); f[2 * h + 1] = bfhi(pk);
            }
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
            // four consecutive lanes = 32 consecutive columns of one row: the maximum over a quad with two DPP moves
            // (quad_perm [1,0,3,2] and [2,3,0,1]); __shfl_xor would go through ds_bpermute, an LDS round trip per step
            amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));
            amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x4E, 0xf, 0xf, true)));
            const int ef = (int)((__float_as_uint(amax) >> 23) & 0xff);
            int e8 = amax > 0.f ? ef - 8 : 127;
            e8 = max(0, min(254, e8));
            float inv = __uint_as_float((uint32_t)(254 - e8) << 23);   // 2^-(e8 - 127)
            if (amax * inv > 448.f) {    // the block maximum lies in (448, 512) 2^e: one exponent up instead of saturating it (header: MX scale rule)
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
            if (ok[k]) {
              const u32x2 w8 = {(uint32_t)lo, (uint32_t)hi};
              *(u32x2*)(ep.C8 + (int64_t)m[k] * ep.ldc8 + n) = w8;
              // (collecting a pass's scale bytes in LDS and storing them as whole dwords after its barrier was measured: no gain)
              if ((cl & 3) == 0)
                ep.c8_scales[((((int64_t)(n >> 7) * 4 + ((n & 127) >> 5)) * ep.c8_rgroups + (m[k] >> 6)) * 16 + (m[k] & 15)) * 4 +
                             ((m[k] & 63) >> 4)] = (uint8_t)e8;
            }
          }
        }
      }
      if (Q8 && !ep.C) {
      } else if (ep.out_f32) {
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (ok[k]) {
            float* cp = (float*)ep.C + (int64_t)m[k] * ep.ldc + n;
#pragma unroll
            for (int g = 0; g < W; g += 4) {
              f32x4 w = {o[k][g], o[k][g + 1], o[k][g + 2], o[k][g + 3]};
              if (LOADS && ep.accumulate) w += *(const f32x4*)(cp + g);     // (the callers count `accumulate` among the loads)
              if (NT) __builtin_nontemporal_store(w, (f32x4*)(cp + g)); else *(f32x4*)(cp + g) = w;
            }
          }
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (ok[k]) store_bf16_row<W, NT>((mg_bf16*)ep.C + (int64_t)m[k] * ep.ldc + n, o[k]);
      }
    };
    constexpr int MAXIT = 4;
    if constexpr (!LOADS) {
      // up to four iterations, unrolled; the row scales of all of them are loaded up front (one wait, before any store)
      float rsc_all[MAXIT][R];
      if (row_scale) {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
          for (int k = 0; k < R; ++k) rsc_all[it][k] = row_scale[min(row_m(min(r + (it * R + k) * step, rows - 1)), M - 1)];
      }
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (r + it * R * step < rows) iteration(r + it * R * step, rsc_all[it]);
    } else {
      for (; r < rows; r += R * step) iteration(r, nullptr);
    }
    return;
  }
  // lanes on a partial group of columns (last column tile of a matrix whose N is not a multiple of W): one row at a time
  for (; r < rows; r += step) {
    const int m = row_m(r);
    if (m < M) {
      float v[W];
      lds_row(r, v);
      if (row_scale) {
        const float rs = row_scale[m];
#pragma unroll
        for (int g = 0; g < W; ++g) v[g] *= rs;
      }
      epilogue_apply<W, NT>(ep, c, m, n, v, N);
    }
  }
}

template <int NCOLS, int ROWB, int W, bool NT>
MG_DEV void epilogue_rows(const mg_epilogue& ep, const char* lds, int rows, int nwaves, int wave, int lane,
                          int m_base, int hi_stride, int n0, int M, int N, const float* row_scale = nullptr) {
  const int n = n0 + (lane % (NCOLS / W)) * W;
  if (n >= N) return;
  EpiColsW<W> c;
  epilogue_cols<W>(ep, n, N, c);
  // no aux / residual operand and at most four iterations per lane: the build of the walk without global loads
  const bool loads = ep.aux_mode != MG_AUX_NONE || ep.res0 || ep.res1 || ep.res2 || ep.accumulate;
  if (!loads && rows <= 8 * nwaves * (64 / (NCOLS / W)))
    epilogue_rows_c<NCOLS, ROWB, W, NT, false, false>(ep, c, lds, rows, nwaves, wave, lane, m_base, hi_stride, n0, M, N, row_scale);
  else
    epilogue_rows_c<NCOLS, ROWB, W, NT>(ep, c, lds, rows, nwaves, wave, lane, m_base, hi_stride, n0, M, N, row_scale);
}


// ---------------------------------------------------------------------------
// skinny (decode) kernel
// ---------------------------------------------------------------------------
#ifdef MG_GEMM_ABLATIONS
#define MG_SKINNY_DBG_NOSTATS(p_) (((p_).dbg & 1) != 0)
#else
#define MG_SKINNY_DBG_NOSTATS(p_) false      // the product library has no switch that produces wrong results
#endif
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
  // fp8 weights, bf16 activations (W8A16): W holds e4m3 bytes in the layout [n-tile][k-step pair][64 lanes][16 B]
  // (lane = kq*16 + n: bytes 0-7 = W[n][32*(2j) + 8kq ..], bytes 8-15 = the same columns of k-step 2j+1) and
  // w_scale[n] the per-output-channel scale; the weights are widened to bf16 in registers, so the stream is half as long.
  const float* w_scale;
#ifdef MG_GEMM_ABLATIONS
  int dbg;   // ablation library only (`make ABL=1`, MAGMA_SKINNY_DBG): bit 0 = skip the LayerNorm-fold row statistics (WRONG results)
#endif
};

// Device body: `block` is the workgroup's index inside THIS problem's grid, `lds` a caller-provided
// scratch of skinny_lds_bytes<WAVES,NT>() bytes -- so several problems (and other workgroup kinds,
// see decode_fused.hip) can share one launch.
template <int WAVES, int NT>
constexpr int skinny_lds_bytes() { return WAVES * NT * 256 * 4 + WAVES * 16 * 2 * 4; }

// 8 e4m3 bytes -> 8 bf16 (exact: every e4m3 value is a bf16 value)
MG_DEV bf16x8 fp8x8_to_bf16(uint32_t w0, uint32_t w1) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
  const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
  const u32x4 o = {pack2bf(a[0], a[1]), pack2bf(b[0], b[1]), pack2bf(c[0], c[1]), pack2bf(d[0], d[1])};
  return __builtin_bit_cast(bf16x8, o);
}

// gemv_dma.hip: the same GEMV with an LDS-DMA loader wave and one persistent workgroup per CU (nt_hint bit 17)
int skinny_dma_launch(const SkinnyParams& sp, int variant, hipStream_t s);

struct NoWait { MG_DEV void operator()() const {} };

// COH: activations in / out go through the coherent 8-byte accessors (persistent decode step); `wait` is called once,
// AFTER the first chunk's weight loads have been issued and BEFORE the first activation load: inside the persistent
// launch it blocks on the producer's completion counter while the weights are already streaming in.
// PIPE: the weight bursts are double-buffered in registers -- burst c+1 is issued BEFORE the MFMAs of burst c, so a wave
// always has KC..2*KC weight loads in flight instead of draining its queue at every burst boundary.  That matters where
// occupancy cannot hide the drain: fc_out (N = 4096 -> 256 workgroups of 4 waves, ONE wave per SIMD, 8 bursts of 16 loads
// each).  Same MFMA order per accumulator, hence bit-identical results.
template <int WAVES, int KC, int NT, bool W8 = false, bool COH = false, class Wait = NoWait, bool PIPE = false>
MG_DEV void skinny_body(const SkinnyParams& p, int block, char* lds, Wait wait = Wait()) {
  static_assert(!W8 || KC % 2 == 0, "fp8 weights are stored in k-step pairs");
  static_assert(!PIPE || (!COH && __is_same(Wait, NoWait)), "the pipelined stream has no dependency wait / coherent loads");
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

  constexpr int WL = W8 ? KC / 2 : KC;      // 16-byte loads per n-tile and chunk
  u32x4 wf[NT][WL];
  auto load_w = [&](int kc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int nt = min(nt0 + t, p.ntiles - 1);
      const u32x4* wp = W8 ? (const u32x4*)p.W + ((int64_t)nt * (p.ksteps >> 1) + ((ks0 + kc) >> 1)) * 64 + lane
                           : (const u32x4*)p.W + ((int64_t)nt * p.ksteps + ks0 + kc) * 64 + lane;
#pragma unroll
      for (int i = 0; i < WL; ++i) wf[t][i] = __builtin_nontemporal_load(wp + i * 64);
    }
  };
  constexpr bool AHEAD = !__is_same(Wait, NoWait);     // persistent step: first weight burst before the dependency wait
  if constexpr (AHEAD) { load_w(0); wait(); }
  if constexpr (PIPE) {
    u32x4 wg[NT][WL];                                    // second register buffer (wf is the first)
    auto load_into = [&](u32x4 (&dst)[NT][WL], int kc) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int nt = min(nt0 + t, p.ntiles - 1);
        const u32x4* wp = W8 ? (const u32x4*)p.W + ((int64_t)nt * (p.ksteps >> 1) + ((ks0 + kc) >> 1)) * 64 + lane
                             : (const u32x4*)p.W + ((int64_t)nt * p.ksteps + ks0 + kc) * 64 + lane;
#pragma unroll
        for (int i = 0; i < WL; ++i) dst[t][i] = __builtin_nontemporal_load(wp + i * 64);
      }
    };
    // one burst: x fragments of burst kc (L2 hits), THEN the weights of the following burst, then the MFMAs of burst kc --
    // vmcnt counts in order, so the wait for x leaves exactly the following burst's weight loads outstanding
    // (has_next is a compile-time fact of the call site: behind a run-time branch hipcc has to assume at the join that
    //  the following burst was NOT issued and waits for vmcnt(0) -- which drains the very loads this is about)
    auto burst = [&](u32x4 (&cur)[NT][WL], u32x4 (&nxt)[NT][WL], int kc, auto has_next) {
      bf16x8 xf[KC];
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        u32x4 raw = *(const u32x4*)(xrow + (int64_t)(ks0 + kc + i) * 32);
        if (!xok) raw = (u32x4){0u, 0u, 0u, 0u};
        xf[i] = __builtin_bit_cast(bf16x8, raw);
      }
      if constexpr (decltype(has_next)::value) load_into(nxt, kc + KC);
      __builtin_amdgcn_sched_barrier(0);
      if (p.ln_colsum && !MG_SKINNY_DBG_NOSTATS(p)) {
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
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < KC; ++i) {
          bf16x8 w;
          if constexpr (W8) w = fp8x8_to_bf16(cur[t][i >> 1][(i & 1) * 2], cur[t][i >> 1][(i & 1) * 2 + 1]);
          else w = __builtin_bit_cast(bf16x8, cur[t][i]);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf[i], acc[t], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr std::integral_constant<bool, true> more{};
    constexpr std::integral_constant<bool, false> last{};
    load_into(wf, 0);
    int kc = 0;
    for (; kc + 2 * KC < per_wave; kc += 2 * KC) {       // both bursts of the pair have a successor
      burst(wf, wg, kc, more);
      burst(wg, wf, kc + KC, more);
    }
    if (kc + KC < per_wave) { burst(wf, wg, kc, more); burst(wg, wf, kc + KC, last); }
    else burst(wf, wg, kc, last);
  } else
  for (int kc = 0; kc < per_wave; kc += KC) {
    // Issue the whole chunk's loads before the first MFMA (GEMV recipe: loads
    // straight to VGPRs, deep queue, late wait): weights first (HBM, non-temporal
    // -- each byte is read exactly once per step), then the x fragments (L2 hits).
    if (!AHEAD || kc > 0) load_w(kc);
    bf16x8 xf[KC];
    if constexpr (COH) {
      static_assert(KC == 8, "the coherent fragment batch is 8 k-steps");
      u32x4 raw[8];
      ld16x8_coh(xrow + (int64_t)(ks0 + kc) * 32, raw);
#pragma unroll
      for (int i = 0; i < KC; ++i) xf[i] = __builtin_bit_cast(bf16x8, xok ? raw[i] : (u32x4){0u, 0u, 0u, 0u});
    } else {
#pragma unroll
      for (int i = 0; i < KC; ++i) {
        u32x4 raw = *(const u32x4*)(xrow + (int64_t)(ks0 + kc + i) * 32);
        if (!xok) raw = (u32x4){0u, 0u, 0u, 0u};
        xf[i] = __builtin_bit_cast(bf16x8, raw);
      }
    }
    if (p.ln_colsum && !MG_SKINNY_DBG_NOSTATS(p)) {
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
      for (int i = 0; i < KC; ++i) {
        bf16x8 w;
        if constexpr (W8) w = fp8x8_to_bf16(wf[t][i >> 1][(i & 1) * 2], wf[t][i >> 1][(i & 1) * 2 + 1]);
        else w = __builtin_bit_cast(bf16x8, wf[t][i]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf[i], acc[t], 0, 0, 0);
      }
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
    if (W8 && n < p.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] *= (n + r < p.N ? p.w_scale[n + r] : 0.f);
    }
    if (p.ln_colsum && n < p.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = rstd * (s[r] - mean * (n + r < p.N ? p.ln_colsum[n + r] : 0.f));
    }
    if (p.split_n > 0 && n >= p.split_n) epilogue_store4<COH>(p.ep_b, li, n - p.split_n, s, p.N - p.split_n);
    else epilogue_store4<COH>(p.ep, li, n, s, p.split_n > 0 ? p.split_n : p.N);
  }
}

