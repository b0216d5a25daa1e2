// gemv_dma.hip -- the decode weight-streaming GEMV with LDS-DMA loader waves.  ABLATION LIBRARY ONLY (`make ABL=1`, `nt_hint`
// bit 17 of mg_gemm_skinny_bf16, `tools/kbench.py dma`): a round-5 experiment that lost -- every form below streams at
// 5.4-5.9 TB/s where skinny_body (gemm_device.h: weights straight into registers) runs at 6.0-6.2 on the same shapes
// (profiles/r05_gemv_lds_dma_experiment.txt).
//
// One persistent workgroup per CU.  Wave 0 only moves bytes: the workgroup's share of W -- whole n-tiles, contiguous in the
// fragment-tiled image, [n-tile][k-step][64 lanes][16 B] -- goes through a ring of RING 16-KiB slots (16 k-steps of one
// n-tile each) with `global_load_lds_dwordx4`, RING - 1 slots in flight behind a counted vmcnt.  Waves 1..NC multiply: an
// n-tile belongs to ONE consumer (tile t -> consumer t % NC), which walks its slots in order, keeps the 16 x 16 accumulator
// in registers over the whole contraction and runs the epilogue itself -- no cross-wave reduction, no barrier after the
// prologue.  x ([M <= 8][K]) is staged once per workgroup as the MFMA's B-operand image ([k-step][k-quarter][row] x 16 B;
// the default kernel re-reads it from L2 in every one of its ntiles workgroups), and the LayerNorm-fold row statistics come
// from the fragments of a consumer's first tile.
// Hand-shake through LDS words: landed = slots the loader has seen complete; done[c] = slots consumer c has released.
#include "gemm_device.h"

namespace {

constexpr int DMA_SLOT = 16384;        // 16 k-steps x 1 KiB
constexpr int DMA_NC = 3;              // consumer waves

// flag words by 32-bit LDS address, as assembly: a volatile access through a generic pointer becomes a FLAT instruction,
// which counts on vmcnt as well -- the counter the loader's ring is ordered by
MG_DEV uint32_t lds_peek(uint32_t a) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}
MG_DEV void lds_poke(uint32_t a, uint32_t v) { asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory"); }
MG_DEV void lds_inc(uint32_t a) { asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(1u) : "memory"); }

template <bool NTL>
MG_DEV void dma16(const void* base_uniform, uint32_t byte_off, uint32_t lds_addr) {
  if constexpr (NTL)
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(byte_off), "s"(base_uniform), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(byte_off), "s"(base_uniform), "s"(lds_addr) : "memory");
}

// NL loader waves (slot j belongs to loader j % NL; each keeps DMA_W slots in flight behind its own vmcnt), XLDS: x staged in
// LDS once per workgroup (ring of 4 slots beside it) or read from L2 per slot as the default kernel does (ring of 8)
// R ring slots; DMA_W = slots a loader leaves in flight behind its wait: 16 * (W + 1) <= 64 DMA instructions at the peak
template <bool NTL, int NL, bool XLDS, int R, int DMA_W>
__global__ __launch_bounds__(64 * (NL + DMA_NC)) void skinny_dma_kernel(const SkinnyParams p) {
  extern __shared__ __attribute__((aligned(1024))) char dlds[];
  static_assert(DMA_W == 2 || DMA_W == 3, "wait immediates below");
  // [ring R x 16 KiB][x image ksteps x 512 B (XLDS)][flags]
  char* ring = dlds;
  char* ximg = dlds + R * DMA_SLOT;
  uint32_t* flagw = (uint32_t*)(ximg + (XLDS ? p.ksteps * 512 : 0));   // landed[NL], done[NC], consumers staged
  const uint32_t flags = lds_u32(flagw);
  const uint32_t f_done = flags + 4 * NL, f_arrived = f_done + 4 * DMA_NC;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((int64_t)p.ntiles * b / G), t1 = (int)((int64_t)p.ntiles * (b + 1) / G);
  const int spt = p.ksteps >> 4;                               // slots per n-tile
  const int nslots = (t1 - t0) * spt;
  if (threadIdx.x < NL + DMA_NC + 1) flagw[threadIdx.x] = 0;
  __syncthreads();

  if (wave < NL) {
    // ---- loaders ----
    const char* src = (const char*)p.W + (int64_t)t0 * p.ksteps * 1024;
    const uint32_t ring_a = lds_u32(ring);
    const uint32_t lane_off = lane * 16;
    const uint32_t f_landed = flags + 4 * wave;
    // slot order: the NC tiles of a group round-robin (tile 0 slot 0, tile 1 slot 0, tile 2 slot 0, tile 0 slot 1, ...), so
    // that all consumers work at the same time, each on its own tile
    const int ntl = t1 - t0, gslots = DMA_NC * spt;
    auto where = [&](int j, int& tile, int& sl, int& own) {
      const int g = j / gslots, r = j - g * gslots;
      const int tig = min(DMA_NC, ntl - g * DMA_NC);
      own = r % tig; sl = r / tig; tile = g * DMA_NC + own;
    };
    auto owner_done = [&](int j) {        // has slot j been released by its consumer?
      int tile, sl, own;
      where(j, tile, sl, own);
      return lds_peek(f_done + 4 * own) > (uint32_t)((tile / DMA_NC) * spt + sl);
    };
    const int nmine = (nslots - wave + NL - 1) / NL;
    for (int k = 0; k < nmine; ++k) {
      const int j = k * NL + wave;
      if (j >= R) while (!owner_done(j - R)) __builtin_amdgcn_s_sleep(1);
      const uint32_t dst = ring_a + (j % R) * DMA_SLOT;
      int tile, sl, own;
      where(j, tile, sl, own);
      const uint32_t off = (uint32_t)(tile * spt + sl) * DMA_SLOT + lane_off;
#pragma unroll
      for (int q = 0; q < 16; ++q) dma16<NTL>(src, off + q * 1024, dst + q * 1024);
      if (k >= DMA_W) {
        if constexpr (DMA_W == 3) MG_WAIT_VMCNT(48); else MG_WAIT_VMCNT(32);   // W slots may still be in flight: my slot k - W has landed
        lds_poke(f_landed, (uint32_t)(k - DMA_W + 1));
      }
    }
    if (DMA_W > 2 && nmine > 2) { MG_WAIT_VMCNT(32); lds_poke(f_landed, (uint32_t)(nmine - 2)); }
    if (nmine > 1) { MG_WAIT_VMCNT(16); lds_poke(f_landed, (uint32_t)(nmine - 1)); }
    MG_WAIT_VMCNT(0);
    lds_poke(f_landed, (uint32_t)nmine);
    return;
  }

  // ---- consumers ----
  const int c = wave - NL;
  const int li = lane & 15, lq = lane >> 4;
  const bool xok = li < p.M;
  if constexpr (XLDS) {  // stage x: 8 rows x ksteps x 4 pieces of 16 B, rows >= M as zeros
    const int pieces = p.ksteps * 4;
    for (int idx = threadIdx.x - 64 * NL; idx < 8 * pieces; idx += 64 * DMA_NC) {
      const int row = idx / pieces, pc = idx - row * pieces;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < p.M) v = *(const u32x4*)(p.X + (int64_t)row * p.ldx + pc * 8);
      *(u32x4*)(ximg + ((pc * 8) + row) * 16) = v;            // pc = ks * 4 + kq
    }
    // consumers only: a named barrier does not exist on gfx950, so count arrivals in LDS
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's x pieces are in LDS
    if (lane == 0) lds_inc(f_arrived);
    while (lds_peek(f_arrived) < (uint32_t)DMA_NC) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  }
  const char* xl = ximg + (lq * 8 + (li & 7)) * 16;
  const mg_bf16* xrow = p.X + (int64_t)(xok ? li : 0) * p.ldx + lq * 8;
  float mean = 0.f, rstd = 1.f;
  bool have_stats = false;
  uint32_t released = 0;
  for (int t = t0 + c; t < t1; t += DMA_NC) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};     // two chains: even / odd k-steps
    float xs = 0.f, xss = 0.f;
    const int g = (t - t0) / DMA_NC, tig = min(DMA_NC, (t1 - t0) - g * DMA_NC);
    for (int s = 0; s < spt; ++s) {
      const int slot = g * DMA_NC * spt + s * tig + c;
      bf16x8 wf[16], xf[16];
      if constexpr (!XLDS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          u32x4 raw = *(const u32x4*)(xrow + (int64_t)(s * 16 + i) * 32);
          if (!xok) raw = (u32x4){0u, 0u, 0u, 0u};
          xf[i] = __builtin_bit_cast(bf16x8, raw);
        }
      }
      while (lds_peek(flags + 4 * (slot % NL)) <= (uint32_t)(slot / NL)) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      const char* wl = ring + (slot % R) * DMA_SLOT + lane * 16;
      const char* xk = xl + s * 16 * 512;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        wf[i] = *(const bf16x8*)(wl + i * 1024);
        if constexpr (XLDS) xf[i] = *(const bf16x8*)(xk + i * 512);
      }
      if (p.ln_colsum && !have_stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const u32x4 raw = __builtin_bit_cast(u32x4, xf[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = bflo(raw[j]), bb = bfhi(raw[j]);
            xs += a + bb;
            xss += a * a + bb * bb;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[i], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i + 1], xf[i + 1], acc1, 0, 0, 0);
      }
      asm volatile("" ::: "memory");
      ++released;
      lds_poke(f_done + 4 * c, released);     // the fragments are in registers (the MFMAs above consumed them)
    }
    if (p.ln_colsum && !have_stats) {
      xs += __shfl_xor(xs, 16, 64); xs += __shfl_xor(xs, 32, 64);
      xss += __shfl_xor(xss, 16, 64); xss += __shfl_xor(xss, 32, 64);
      mean = xs * p.ln_inv_d;
      rstd = rsqrtf(fmaxf(xss * p.ln_inv_d - mean * mean, 0.f) + p.ln_eps);
      have_stats = true;
    }
    if (!xok) continue;
    const int n = t * 16 + lq * 4;
    f32x4 sres = acc + acc1;
    if (p.ln_colsum && n < p.N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sres[r] = rstd * (sres[r] - mean * (n + r < p.N ? p.ln_colsum[n + r] : 0.f));
    }
    if (p.split_n > 0 && n >= p.split_n) epilogue_store4<false>(p.ep_b, li, n - p.split_n, sres, p.N - p.split_n);
    else epilogue_store4<false>(p.ep, li, n, sres, p.split_n > 0 ? p.split_n : p.N);
  }
}

template <bool NTL, int NL, bool XLDS, int R, int W>
void launch_dma(const SkinnyParams& sp, int grid, hipStream_t s) {
  const int lds = R * DMA_SLOT + (XLDS ? sp.ksteps * 512 : 0) + 64;
  static const bool attr = [] {
    (void)hipFuncSetAttribute((const void*)skinny_dma_kernel<NTL, NL, XLDS, R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL((skinny_dma_kernel<NTL, NL, XLDS, R, W>), dim3(grid), dim3(64 * (NL + DMA_NC)), lds, s, sp);
}

}  // namespace

// variant bit 0: non-temporal DMA; bits 1..3: form; bits 4..12: workgroups (0 = 256)
int skinny_dma_launch(const SkinnyParams& sp, int variant, hipStream_t s) {
  if (sp.w_scale) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: the LDS-DMA GEMV streams bf16 weights");
  if (sp.M > 8 || (sp.ksteps & 15) || sp.ksteps > 128) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: the LDS-DMA GEMV needs M <= 8 and K %% 512 == 0, K <= 4096 (M=%d K=%d)", sp.M, sp.ksteps * 32);
  int grid = (variant >> 4) & 0x1FF;
  if (grid == 0) grid = 256;
  if (grid > sp.ntiles) grid = sp.ntiles;
  const bool nt = variant & 1;
  const int form = (variant >> 1) & 7;     // 0: one loader, ring 4, W 3 (x in LDS); 1: two loaders, x from L2, ring 8; 2: one loader, ring 5; 3: two loaders, ring 5, W 2
  if (!nt) {
    if (form == 1) launch_dma<false, 2, false, 8, 3>(sp, grid, s); else launch_dma<false, 1, true, 4, 3>(sp, grid, s);
  } else switch (form) {
    case 1: launch_dma<true, 2, false, 8, 3>(sp, grid, s); break;
    case 2: launch_dma<true, 1, true, 5, 3>(sp, grid, s); break;
    case 3: launch_dma<true, 2, true, 5, 2>(sp, grid, s); break;
    default: launch_dma<true, 1, true, 4, 3>(sp, grid, s); break;
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}
