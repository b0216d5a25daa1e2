// gemm.hip -- bf16 MFMA GEMM family for gfx950 (MI355X).
//
//   C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), fp32 accumulate.
//
// Kernels:
//   gemm128_kernel   128x128x64 workgroup tile, 4 waves (2x2), each wave a 64x64
//                    sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators.
//                    Operand tiles go HBM -> LDS with global_load_lds_dwordx4
//                    (LDS-DMA, no VGPR round trip), double buffered, one barrier
//                    per K-tile.  Row-major operands use an XOR chunk swizzle
//                    applied on the *source* address (the DMA writes LDS
//                    lane-linearly) and again on the ds_read_b128; fragment-tiled
//                    weights need no swizzle at all (LDS image == fragment order).
//                    A-operand loaders: dense rows, or implicit-im2col 3x3 conv
//                    over an NHWC image (CLIP trunk).  Split-K for grids that
//                    would leave most CUs idle (+ splitk_fixup_kernel).
//   gemm256_kernel   256x256x64 tile, 8 waves, deep LDS-DMA pipeline with counted
//                    waits and two staggered wave groups (see its own header).
//   skinny_kernel    M <= 16 (decode): pure weight streaming from fragment-tiled
//                    weights straight into VGPRs (1 KiB contiguous per wave
//                    instruction), K split across the waves of a workgroup,
//                    fp32 cross-wave reduction through LDS.  HBM-bound.
//                    skinny2 / decode_attn_gemv co-launch two workgroup kinds.
//
// MFMA operand roles are swapped (weights as the "A" operand, activations as
// "B") so that each lane ends up with 4 *consecutive n* of one output row:
//   acc[r] = C[m0 + (lane&15)][n0 + (lane>>4)*4 + r]
// so consecutive lanes own consecutive columns.  The tile kernels park the accumulators in LDS
// and run the epilogue row-wise with 16-byte accesses (gemm_device.h: epilogue_rows).
#include "common.h"
#include <type_traits>

#include <algorithm>
#include <string.h>
#include <vector>
#include <stdlib.h>
#include "gemm_device.h"
#include "attn_decode_device.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + B
constexpr int EPI128_ROWB = BN * 4 + 16;        // fp32 row of the LDS-staged epilogue (+16 B: bank skew)
constexpr int GEMM_LDS = BM * EPI128_ROWB;     // max(double-buffered stages = 64 KiB, epilogue image = 66 KiB)
static_assert(GEMM_LDS >= 2 * STAGE_BYTES, "LDS must hold both K-tile stages");


// ---------------------------------------------------------------------------
// workgroup id -> output tile.  (1) undo the dispatcher's round-robin over the
// 8 XCDs so each XCD (private 4 MiB L2) owns a contiguous run of tiles
// (bijective form, guide 5.5 T1); (2) inside the run walk groups of GROUP_M
// row-tiles x all column-tiles so the A panels of a group stay L2 resident.
// ---------------------------------------------------------------------------
MG_DEV void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn, int group_m = 8) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GROUP_M = group_m;
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(GROUP_M, tiles_m - first_m);
  const int in_group = wg - group * per_group;
  tm = first_m + in_group % gm;
  tn = in_group / gm;
}

struct GemmParams {
  const mg_bf16* A; int64_t lda;
  const mg_bf16* W; int64_t ldw;
  int M, N, K;
  int H, Wd, Cin;
  const mg_bf16* zero;
  int tiles_m, tiles_n;
  int group_m;       // 256x256 kernel: row-tiles per group of the tile walk (tile_coords)
  int64_t a_kt;      // 256x256 kernel: elements between consecutive K-tiles of A (64 = plain row-major rows)
  // split-K (gemm128 only): `splits` workgroups per output tile, each reducing kt_per K-tiles into
  // its own fp32 slab ws[split][M][ldws]; splitk_fixup_kernel adds the slabs in a fixed order and
  // applies the epilogue (deterministic, no atomics).  splits == 1: the epilogue runs in place.
  int splits, kt_per;
  int nt;            // non-temporal output stores (large outputs)
  const float* row_scale;   // fp8 path: per-row scale of the A operand (nullptr otherwise)
  // MX fp8 path (mg_gemm_mx_fp8): E8M0 block scales in mg_quantize_mx_fp8's dword layout
  // [(chunk * 4 + block) * rgroups + row / 64][row % 16] with byte (row % 64) / 16; nullptr: unit scales
  const uint32_t* mx_a; const uint32_t* mx_w; int rg_a, rg_w;
  float* ws; int64_t ldws;
  mg_epilogue ep;
};

// FP8: A and W hold OCP e4m3 bytes and every quantity below counts PAIRS of them (the host passes K/2, lda/2, ldw/2:
// a 64-"element" K-tile is 128 fp8 values, the byte images in LDS are the same); only the MFMA differs.
template <int AMODE, int WLAYOUT, bool FP8 = false, bool MX = false>
__global__ __launch_bounds__(256) void gemm128_kernel(const GemmParams p) {
  static_assert(!MX || FP8, "block scales belong to the fp8 path");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn, sp = 0;
  tile_coords(blockIdx.x, p.tiles_m, p.tiles_n * p.splits, tm, tn);
  if (p.splits > 1) { const int t = tn / p.splits; sp = tn - t * p.splits; tn = t; }  // the splits of a tile are neighbours
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = sp * p.kt_per, kt1 = min(nkt, kt0 + p.kt_per);

  // ---- per-lane staging state: 4 DMA pieces of A and 4 of B per K-tile ----
  // piece j of wave w fills LDS rows [32w+8j, 32w+8j+8) x 128 B; lane -> (row
  // r = 32w+8j+(lane>>3), LDS chunk c = lane&7), reads global chunk g = c ^ f(r)
  // with f(r) = (r>>1)&7  (conflict-free ds_read_b128 for the 16-lane groups).
  const mg_bf16* a_src[4];
  int a_k[4];
  // conv state
  int cv_y[4], cv_x[4], cv_tap[4], cv_cc[4];
  bool cv_ok[4];
  const int cpt = (AMODE == MG_A_CONV3X3) ? (p.Cin >> 3) : 1;  // 16-B chunks per tap
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + j * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    a_k[j] = g * 8;
    const int m = m0 + r;
    if (AMODE == MG_A_DENSE) {
      const int mc = min(m, p.M - 1);
      a_src[j] = p.A + (int64_t)mc * p.lda + g * 8;
    } else {
      const int mc = min(m, p.M - 1);
      const int x = mc % p.Wd;
      const int t = mc / p.Wd;
      const int y = t % p.H;
      cv_y[j] = y; cv_x[j] = x;
      cv_ok[j] = (m < p.M);
      a_src[j] = p.A + (int64_t)mc * p.Cin;   // centre pixel, channel 0
      const int g_abs = g + kt0 * 8;          // first chunk this workgroup stages
      cv_tap[j] = g_abs / cpt;
      cv_cc[j] = g_abs - cv_tap[j] * cpt;
    }
  }
  const mg_bf16* b_src[4];
  int b_k[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (WLAYOUT == MG_W_ROWMAJOR) {
      const int r = wave * 32 + j * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((r >> 1) & 7);
      b_k[j] = g * 8;
      const int nc = min(n0 + r, p.N - 1);
      b_src[j] = p.W + (int64_t)nc * p.ldw + g * 8;
    } else {
      // fragment-tiled: piece = (n-tile nt_l = (4w+j)>>1, k-step ks_l = (4w+j)&1)
      const int bi = wave * 4 + j;
      const int ntiles = (p.N + 15) >> 4;
      const int nt = min((n0 >> 4) + (bi >> 1), ntiles - 1);
      const int64_t ksteps = p.ldw >> 5;  // Kp / 32
      b_src[j] = p.W + ((int64_t)nt * ksteps + (bi & 1)) * 512 + lane * 8;
      b_k[j] = 0;
    }
  }

  // LDS-DMA from inline asm with 32-bit LDS destinations (common.h glds16au): hipcc then counts the waits of the fragment
  // reads instead of draining them (lgkmcnt(0)) in front of every MFMA burst; the DMA itself is retired by hand
  // (MG_WAIT_VM(0) in front of the barrier that publishes a stage).
  const uint32_t smem_u = lds_u32(smem);
  auto stage = [&](int kt, int buf) {
    const uint32_t abase = smem_u + (uint32_t)(buf * STAGE_BYTES);
    const uint32_t bbase = abase + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const mg_bf16* src;
      if (AMODE == MG_A_DENSE) {
        src = (kt * BK + a_k[j] < p.K) ? a_src[j] + kt * BK : p.zero;
      } else {
        // implicit im2col: k = tap*Cin + ci, tap = ky*3+kx, pad 1
        const int tap = cv_tap[j];
        const int ky = tap / 3, kx = tap - ky * 3;
        const int yy = cv_y[j] + ky - 1, xx = cv_x[j] + kx - 1;
        const bool ok = cv_ok[j] && tap < 9 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
        src = ok ? a_src[j] + ((int64_t)(ky - 1) * p.Wd + (kx - 1)) * p.Cin + cv_cc[j] * 8 : p.zero;
        // advance by one K-tile = 8 chunks
        int cc = cv_cc[j] + 8, tp = tap;
        while (cc >= cpt) { cc -= cpt; ++tp; }
        cv_cc[j] = cc; cv_tap[j] = tp;
      }
      glds16au(src, abase + (wave * 32 + j * 8) * 128);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const mg_bf16* src;
      if (WLAYOUT == MG_W_ROWMAJOR) {
        src = (kt * BK + b_k[j] < p.K) ? b_src[j] + kt * BK : p.zero;
        glds16au(src, bbase + (wave * 32 + j * 8) * 128);
      } else {
        src = b_src[j] + (int64_t)kt * 1024;   // 2 k-steps * 512 elements
        glds16au(src, bbase + (wave * 4 + j) * 1024);
      }
    }
  };

  // ---- reader offsets -----------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lq = lane >> 4;
  const int fsw = (li >> 1) & 7;  // f(r) only depends on lane (tile rows are 16-aligned)
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_rd[s] = (wm * 64 + li) * 128 + (((s * 4 + lq) ^ fsw) << 4);
    if (WLAYOUT == MG_W_ROWMAJOR) b_rd[s] = TILE_BYTES + (wn * 64 + li) * 128 + (((s * 4 + lq) ^ fsw) << 4);
    else b_rd[s] = TILE_BYTES + ((wn * 4) * 2 + s) * 1024 + lane * 16;
  }
  constexpr int A_MT_STRIDE = 16 * 128;
  constexpr int B_NT_STRIDE = (WLAYOUT == MG_W_ROWMAJOR) ? 16 * 128 : 2 * 1024;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // MX block scales (mg_gemm_mx_fp8): per K-tile ONE dword per operand and lane -- byte t = what this lane supplies for fragment
  // t of the wave's 64-row slab (mg_quantize_mx_fp8's layout; the MFMA's op_sel picks the byte).  Tile kt + 1's pair is loaded
  // from assembly while tile kt is multiplied (invisible to hipcc's wait counting, retired by the loop's vmcnt(0)).
  uint32_t mx_sa = 0x7f7f7f7fu, mx_sw = 0x7f7f7f7fu, mx_sa_n = 0x7f7f7f7fu, mx_sw_n = 0x7f7f7f7fu;
  constexpr bool mx = MX;
  const uint32_t mx_oa = (uint32_t)(((lq * p.rg_a + min((m0 >> 6) + wm, p.rg_a - 1)) * 16 + li) * 4);      // slabs past M / N: rows
  const uint32_t mx_ow = (uint32_t)(((lq * p.rg_w + min((n0 >> 6) + wn, p.rg_w - 1)) * 16 + li) * 4);      // that are never stored
  auto mx_load = [&](int kt, uint32_t& a, uint32_t& w) {
    gld32s_async(a, (const char*)p.mx_a + (int64_t)kt * p.rg_a * 256, mx_oa);
    gld32s_async(w, (const char*)p.mx_w + (int64_t)kt * p.rg_w * 256, mx_ow);
  };
  if (mx) mx_load(kt0, mx_sa, mx_sw);
  stage(kt0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    if (kt + 1 < kt1) { stage(kt + 1, cur ^ 1); if (mx) mx_load(kt + 1, mx_sa_n, mx_sw_n); }
    const char* sb = smem + cur * STAGE_BYTES;
    if constexpr (FP8) {
      // MX block scales of this K-tile: loaded one tile ahead (below), retired by the loop's own vmcnt(0) before the barrier
      i32x8 af[4], bfr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        af[t] = __builtin_shufflevector(*(const i32x4*)(sb + a_rd[0] + t * A_MT_STRIDE), *(const i32x4*)(sb + a_rd[1] + t * A_MT_STRIDE), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        bfr[t] = __builtin_shufflevector(*(const i32x4*)(sb + b_rd[0] + t * B_NT_STRIDE), *(const i32x4*)(sb + b_rd[1] + t * B_NT_STRIDE), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (MX) acc[i][j] = mfma_mx_k128(bfr[j], j, mx_sw, af[i], i, mx_sa, acc[i][j]);
          else acc[i][j] = mfma_fp8_k128(bfr[j], af[i], acc[i][j]);
        }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = *(const bf16x8*)(sb + a_rd[s] + t * A_MT_STRIDE);
#pragma unroll
        for (int t = 0; t < 4; ++t) bfr[t] = *(const bf16x8*)(sb + b_rd[s] + t * B_NT_STRIDE);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed
    __syncthreads();                                   // ... for everyone, and everyone is done reading `cur`
    if (mx) { asm volatile("" : "+v"(mx_sa_n), "+v"(mx_sw_n)); mx_sa = mx_sa_n; mx_sw = mx_sw_n; }
    cur ^= 1;
  }

  // ---- epilogue -------------------------------------------------------------
  if (p.splits > 1) {   // raw partial sums -> this split's slab (ldws covers whole tiles)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + li;
      if (m >= p.M) continue;
      float* row = p.ws + ((int64_t)sp * p.M + m) * p.ldws + n0 + wn * 64 + lq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(row + j * 16) = acc[i][j];
    }
    return;
  }
  // park the accumulators in LDS (the K loop ended on a barrier: nobody reads the stages any more),
  // then walk rows -- see epilogue_rows
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(f32x4*)(smem + (wm * 64 + i * 16 + li) * EPI128_ROWB + (wn * 64 + j * 16 + lq * 4) * 4) = acc[i][j];
  __syncthreads();
  if (epilogue_wide_ok(p.ep)) epilogue_rows<BN, EPI128_ROWB, 8, false>(p.ep, smem, BM, 4, wave, lane, m0, 64, n0, p.M, p.N, p.row_scale);
  else epilogue_rows<BN, EPI128_ROWB, 4, false>(p.ep, smem, BM, 4, wave, lane, m0, 64, n0, p.M, p.N, p.row_scale);
}

// second half of a split-K GEMM: add the slabs (fixed order) and run the fused epilogue
__global__ __launch_bounds__(256) void splitk_fixup_kernel(const float* __restrict__ ws, int splits, int M, int N, int64_t ldws,
                                                           const mg_epilogue ep, const float* __restrict__ row_scale) {
  const int nq = (N + 3) >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * nq) return;
  const int m = (int)(idx / nq), n = (int)(idx - (int64_t)m * nq) * 4;
  const float* src = ws + (int64_t)m * ldws + n;
  f32x4 v = *(const f32x4*)src;
  for (int s = 1; s < splits; ++s) {
    const f32x4 t = *(const f32x4*)(src + (int64_t)s * M * ldws);
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  if (row_scale) { const float rs = row_scale[m]; v[0] *= rs; v[1] *= rs; v[2] *= rs; v[3] *= rs; }
  epilogue_store4(ep, m, n, v, N);
}

// ---------------------------------------------------------------------------
// gemm256_kernel: 256x256x64 workgroup tile, 8 waves (2 x 4), each wave a 128x64
// sub-tile as 8x4 MFMA accumulators (128 VGPRs).  Deep-pipelined K loop in the
// style of the 8-phase structure of the CDNA4 guide:
//   * LDS holds two K-tiles (2 x 64 KiB); each is staged as four 16-KiB half-tiles
//     (W rows 0-127 / 128-255, A rows 0-127 / 128-255) by LDS-DMA, ONE half-tile per
//     phase, started right after the last read of the buffer it overwrites (5 to 2
//     phases before its first read).  The only wait is a COUNTED s_waitcnt vmcnt(2)
//     once per K-tile, so loads stay in flight across the workgroup barriers.
//   * a K-tile is multiplied in four phases of 16 MFMAs (one 64x32 quadrant of the
//     wave's tile x K=64):  [ds_read the quadrant's new fragments | issue one DMA
//     half-tile] - barrier - [16 MFMA under s_setprio 1] - barrier.
//   * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same
//     program shifted by ONE barrier, so that while one group is in its MFMA segment
//     the other is in its read/DMA segment: the matrix pipe of every SIMD always has
//     a wave to issue from.
// Phase q of K-tile t (buffer t&1); registers: A 4 m-tiles x 2, W both n-halves.
// bf16 16x16x32 path (EARLY, round 3): the DMA units are cut by the phase of their LAST READ -- A_m = the mh-th 64 rows of both
// groups, W_n = the nh-th 32 rows of the four wave columns -- and restaged as soon as that read is over, into the buffer that is
// still being multiplied:
//   q0: read A(mh0), W(nh0) | DMA A_1(t+1)  (last read: q2 of t-1)   | MFMA (mh0,nh0)
//   q1: read W(nh1)         | DMA A_0(t+2)  (last read: q0 of t)     | MFMA (mh0,nh1)
//   q2: read A(mh1)         | DMA W_0(t+2)  (last read: q0 of t)     | MFMA (mh1,nh1)
//   q3:                     | DMA W_1(t+2)  (last read: q1 of t), vmcnt(6): tile t+1 landed | MFMA (mh1,nh0)
// Every unit is in flight for 3 to 6 phases before the wait that retires it (the first version, below, cut the units by wave
// group and issued them at q0..q3 of the tile BEFORE their use: A_hi had ONE phase -- ~350 ns, less than an L2 hit -- so every
// K-tile waited for memory; an L2-hot ablation ran 3-19 % faster, the K = 16384 shapes most).
// (32x32x16 path: the same units and schedule; fp8 path: its own unit order, see MG_PHASEQ.)
// First version of the schedule, kept as tile_hint 266 for A/B runs (units cut by group: W_lo / W_hi = W rows 0-127 / 128-255,
// A_lo / A_hi = the rows of group 0 / 1):
//   q0: read A(mh0), W(nh0) | DMA W_hi(t+1) | MFMA (mh0,nh0)
//   q1: read W(nh1)         | DMA A_lo(t+1) | MFMA (mh0,nh1)
//   q2: read A(mh1)         | DMA A_hi(t+1) | MFMA (mh1,nh1)
//   q3:                     | DMA W_lo(t+2) , vmcnt(2): tile t+1 landed | MFMA (mh1,nh0)
// Hazards (slots = intervals between consecutive barriers; group 1 is one slot late: group 0's read segment of phase q is
// slot 2q, group 1's is slot 2q+1):
//   WAR  (EARLY) a unit is restaged in the phase after its last read: the last reader is group 1 (slot 2q+1, its reads retired
//        by lgkmcnt(0) before the barrier that ends the slot), the first writer group 0 in slot 2q+2.
//   WAR  (by group) buffer t is last read (lgkmcnt(0) before the barrier) in q2 of group 1; the
//        first DMA into it is W_lo(t+2) at q3 of group 0, one barrier later.
//   RAW  every wave retires its own pieces of tile t+1 (vmcnt(2)) in q3 before that
//        phase's first barrier; the first reader (group 0, q0 of t+1) has passed the
//        barrier that group 1 reaches after ITS q3 wait.
// Requirements: dense A, K % 128 == 0.  Operand layouts, swizzle, swapped MFMA roles
// and epilogue as in gemm128_kernel.
// ---------------------------------------------------------------------------
constexpr int MG_GEMM256_MFMA_DEFAULT = 16;      // MFMA shape of the bf16 256x256 kernel unless MAGMA_GEMM256_MFMA says 32
constexpr int G256_TILE = 256 * 64 * 2;          // 32 KiB per operand per K-tile
constexpr int G256_BUF = 2 * G256_TILE;          // A + W
constexpr int EPI256_ROWB = 256 * 4 + 16;        // fp32 row of the LDS-staged epilogue (+16 B: bank skew)
constexpr int G256_LDS = 128 * EPI256_ROWB;      // max(two K-tiles = 128 KiB, epilogue image = 130 KiB)
static_assert(G256_LDS >= 2 * G256_BUF, "LDS must hold two K-tiles");

#define MG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define MG_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// FP8: as in gemm128_kernel the operands are e4m3 bytes counted in pairs; a K-tile is 128 fp8 values per row (the same 128
// bytes), its two 16-byte k-substep fragments form ONE 8-register operand of v_mfma_scale_f32_16x16x128_f8f6f4 -- half the
// MFMA instructions per phase, each twice as long, twice the flops per byte staged.
// MFMA32 (bf16 only): the same tile, pipeline and LDS images on v_mfma_f32_32x32x16_bf16 -- a wave's 128 x 64 sub-tile as
// 4 x 2 accumulators of 32 x 32 (16 registers each).  The 16x16x32 form tops out at ~0.83 of the matrix peak on this chip
// (2 075 vs 2 382 TF in the guide's micro-benchmarks: ~19 vs 32 cycles per SIMD for half the flops), so an MFMA segment of a
// phase -- 16 instructions there, 8 here -- is shorter for the same work.  Operand lane map: lane l supplies row / column
// l & 31 and the 8 consecutive k of half (l >> 5) of a 16-wide k-substep (4 substeps per K-tile); with the swizzle
// chunk ^ ((r >> 1) & 7) the 16-lane groups of ds_read_b128 still cover all 16 slots of a 256-byte bank row (row-major
// images), and in the fragment-tiled W image lane l reads slot (l & 15) of 1-KiB block (l & 31) >> 4 -- conflict-free too.
// Accumulator map: column m = l & 31, rows n = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5): a register quad is 4 consecutive n,
// exactly what the LDS-staged epilogue stores.
typedef __attribute__((ext_vector_type(16))) float f32x16;
// ABL (timing ablations, WRONG results unless noted; tile_hint 261.., tools/kbench.py abl / ksweep; compiled with `make ABL=1`):
// 1 = every K-tile's DMA reads K-tile (t & 1) (operands always L2-hot: isolates memory latency), 2 = no fragment reads after
// the first two K-tiles (isolates the LDS read segments), 3 = no MFMA, 4 = no DMA after the prologue, 6 = the first DMA
// schedule (correct results), 7 = no epilogue, 8 = epilogue staging only, 10 = every tile stores to tile (0, 0).
template <int WLAYOUT, bool LATE_LGKM, bool FP8 = false, bool MFMA32 = false, int ABL = 0, bool SPLITK = false, bool MX = false, bool Q8 = false>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  static_assert(!Q8 || (FP8 && !SPLITK && ABL == 0), "the MX output copy: fp8 path, un-split");
  static_assert(!(FP8 && MFMA32), "the 32x32 form is the bf16 path");
  static_assert(!MX || (FP8 && !SPLITK && ABL == 0), "block scales: fp8 path, un-split");
  bool abl_on = true;           // ABL 2 / 4: false once the pipeline is primed
  const unsigned long long t_start = ABL == 11 ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const unsigned long long c_start = ABL == 11 ? __builtin_amdgcn_s_memtime() : 0ull;     // shader-clock cycles (the 100-MHz counter above is wall time)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  // split-K (long contractions with few output tiles -- the adapters' weight gradients over B*S rows): blocks
  // [sp * tiles, (sp + 1) * tiles) reduce K-tiles [sp * kt_per, (sp + 1) * kt_per) into fp32 slab sp of the workspace through a
  // plain epilogue (set up by the host); splitk_fixup_kernel adds the slabs in a fixed order and runs the real epilogue.
  // (SPLITK is its own instantiation: the extra scalars cost the fp8 form, which lives at the register limit, 320 bytes of scratch)
  int bid = blockIdx.x, sp = 0;
  if constexpr (SPLITK) { const int tiles = p.tiles_m * p.tiles_n; sp = bid / tiles; bid -= sp * tiles; }
  tile_coords(bid, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nkt = SPLITK ? p.kt_per : p.K >> 6;     // even (K % 128 == 0; the host picks an even kt_per)
  const int kt0 = SPLITK ? sp * p.kt_per : 0;
#define KT_SRC(kt) (ABL == 1 ? ((kt) & 1) : (kt0 + (kt)))
  const int wr = wave >> 2, wc = wave & 3;        // wr is also the wave group
  const int li = lane & 15, lq = lane >> 4;

  // ---- DMA sources: per half-tile this wave moves 2 pieces of 1 KiB ----
  int a_off[2][2], b_off[2][2];                   // element offsets (< 2^31 on this path)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = h * 128 + wave * 16 + j * 8 + (lane >> 3);
      const int gch = (lane & 7) ^ ((r >> 1) & 7);
      a_off[h][j] = (int)(min(m0 + r, p.M - 1) * p.lda + gch * 8);
      if (WLAYOUT == MG_W_ROWMAJOR) {
        b_off[h][j] = (int)(min(n0 + r, p.N - 1) * p.ldw + gch * 8);
      } else {
        const int ntiles = (p.N + 15) >> 4;
        const int nt = min((n0 >> 4) + h * 8 + wave, ntiles - 1);
        b_off[h][j] = (int)(((int64_t)nt * (p.ldw >> 5) + j) * 512 + lane * 8);   // j = k-step inside the tile
      }
    }
  // the K-tile advance goes into the (scalar) base, the per-lane part stays a 32-bit byte offset: see glds16s
#define MG_DMA_A(kt, h)                                                                                  \
  {                                                                                                      \
    char* dst_ = smem + ((kt) & 1) * G256_BUF + ((h) * 128 + wave * 16) * 128;                           \
    const mg_bf16* src_ = p.A + (int64_t)KT_SRC(kt) * p.a_kt;                            \
    if (ABL != 4 || abl_on) {                                                                            \
      glds16s(src_, (uint32_t)a_off[h][0] * 2u, dst_);                                                   \
      glds16s(src_, (uint32_t)a_off[h][1] * 2u, dst_ + 1024);                                            \
    }                                                                                                    \
  }
#define MG_DMA_B(kt, h)                                                                                  \
  {                                                                                                      \
    char* base_ = smem + ((kt) & 1) * G256_BUF + G256_TILE;                                              \
    if (ABL == 4 && !abl_on) {                                                                           \
    } else if (WLAYOUT == MG_W_ROWMAJOR) {                                                               \
      const mg_bf16* src_ = p.W + (int64_t)KT_SRC(kt) * 64;                          \
      glds16s(src_, (uint32_t)b_off[h][0] * 2u, base_ + ((h) * 128 + wave * 16) * 128);                  \
      glds16s(src_, (uint32_t)b_off[h][1] * 2u, base_ + ((h) * 128 + wave * 16 + 8) * 128);              \
    } else {                                                                                             \
      const mg_bf16* src_ = p.W + (int64_t)KT_SRC(kt) * 1024;                        \
      glds16s(src_, (uint32_t)b_off[h][0] * 2u, base_ + (((h) * 8 + wave) * 2) * 1024);                  \
      glds16s(src_, (uint32_t)b_off[h][1] * 2u, base_ + (((h) * 8 + wave) * 2 + 1) * 1024);              \
    }                                                                                                    \
  }

  // ---- DMA units of the EARLY schedule (bf16 16x16x32 path): the 16-KiB units are cut by the PHASE in which their rows are
  // last read instead of by wave group -- A_m = the mh-th 64 rows of BOTH groups (read in q0 / q2), W_n = the nh-th 32 rows of all
  // four wave columns (read in q0 / q1) -- so that a unit of the buffer being multiplied can be restaged while the tile is still
  // in progress and every unit gets at least 3 phases of flight (see the schedule below).  Same LDS image.
  constexpr bool EARLY = ABL != 6;
  constexpr bool BALANCED = EARLY && !FP8 && !MFMA32 && !LATE_LGKM && ABL == 12;     // ABL 12 only: measured neutral (+-1 %), see the schedule
  int a2_off[2][2], b2_off[2][2];
  if constexpr (EARLY) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ra = (wave >> 2) * 128 + u * 64 + (wave & 3) * 16 + j * 8 + (lane >> 3);
        a2_off[u][j] = (int)(min(m0 + ra, p.M - 1) * p.lda + ((lane & 7) ^ ((ra >> 1) & 7)) * 8);
        if (WLAYOUT == MG_W_ROWMAJOR) {
          const int rb = (wave >> 1) * 64 + u * 32 + (wave & 1) * 16 + j * 8 + (lane >> 3);
          b2_off[u][j] = (int)(min(n0 + rb, p.N - 1) * p.ldw + ((lane & 7) ^ ((rb >> 1) & 7)) * 8);
        } else {
          const int ntiles = (p.N + 15) >> 4;
          const int nt = min((n0 >> 4) + (wave >> 1) * 4 + u * 2 + (wave & 1), ntiles - 1);
          b2_off[u][j] = (int)(((int64_t)nt * (p.ldw >> 5) + j) * 512 + lane * 8);   // j = k-step inside the tile
        }
      }
  }
#define MG_DMA_A2(kt, u)                                                                                 \
  {                                                                                                      \
    char* dst_ = smem + ((kt) & 1) * G256_BUF + ((wave >> 2) * 128 + (u) * 64 + (wave & 3) * 16) * 128;  \
    const mg_bf16* src_ = p.A + (int64_t)KT_SRC(kt) * p.a_kt;                                            \
    if (ABL != 4 || abl_on) {                                                                            \
      glds16s(src_, (uint32_t)a2_off[u][0] * 2u, dst_);                                                  \
      glds16s(src_, (uint32_t)a2_off[u][1] * 2u, dst_ + 1024);                                           \
    }                                                                                                    \
  }
#define MG_DMA_B2(kt, u)                                                                                 \
  {                                                                                                      \
    char* base_ = smem + ((kt) & 1) * G256_BUF + G256_TILE;                                              \
    if (ABL == 4 && !abl_on) {                                                                           \
    } else if (WLAYOUT == MG_W_ROWMAJOR) {                                                               \
      const mg_bf16* src_ = p.W + (int64_t)KT_SRC(kt) * 64;                                              \
      char* d_ = base_ + ((wave >> 1) * 64 + (u) * 32 + (wave & 1) * 16) * 128;                          \
      glds16s(src_, (uint32_t)b2_off[u][0] * 2u, d_);                                                    \
      glds16s(src_, (uint32_t)b2_off[u][1] * 2u, d_ + 1024);                                             \
    } else {                                                                                             \
      const mg_bf16* src_ = p.W + (int64_t)KT_SRC(kt) * 1024;                                            \
      char* d_ = base_ + (((wave >> 1) * 4 + (u) * 2 + (wave & 1)) * 2) * 1024;                          \
      glds16s(src_, (uint32_t)b2_off[u][0] * 2u, d_);                                                    \
      glds16s(src_, (uint32_t)b2_off[u][1] * 2u, d_ + 1024);                                             \
    }                                                                                                    \
  }

  // ---- MX block scales (mg_gemm_mx_fp8) ----
  // The K loop of the fp8 form lives in exactly 256 VGPRs: scale registers loaded a tile ahead (as in gemm128_kernel) made hipcc
  // spill accumulators inside the loop (round 3).  Here the scales of a K-tile travel with the tile instead: 2 KiB per tile --
  // [operand][k-quarter lq][slab of 64 rows][row % 16] dwords, byte (row % 64) / 16, i.e. mg_quantize_mx_fp8's layout cut to
  // the workgroup's four A slabs and four W slabs -- staged behind the tile buffers by ONE 256-byte LDS-DMA per wave and
  // K-tile (wave = operand * 4 + lq; part of DMA unit A_0, whose waits count one more), read with a ds_read_b32 in the
  // phase that needs them (W and A slab 0 in q0, A slab 1 in q2): two short-lived registers.
  constexpr int MX_OFF = G256_LDS;          // behind the epilogue image
  uint32_t mx_sw = 0x7f7f7f7fu, mx_sa = 0x7f7f7f7fu;
  const int mx_op = wave >> 2, mx_lq = wave & 3;
  const int mx_rg = mx_op ? p.rg_w : p.rg_a;
  const uint32_t mx_src_off = (uint32_t)((mx_lq * mx_rg + min(((mx_op ? n0 : m0) >> 6) + (lane >> 4), mx_rg - 1)) * 64 + (lane & 15) * 4);
  const int mx_rd_w = MX_OFF + 1024 + lq * 256 + wc * 64 + li * 4;
  const int mx_rd_a = MX_OFF + lq * 256 + wr * 128 + li * 4;              // slab 1 of the wave's rows: + 64
#define MG_DMA_S(kt)                                                                                     \
  if constexpr (MX) {                                                                                    \
    const char* src_ = (const char*)(mx_op ? p.mx_w : p.mx_a) + (int64_t)(kt0 + (kt)) * mx_rg * 256;     \
    glds4s(src_, mx_src_off, smem + MX_OFF + ((kt) & 1) * 2048 + wave * 256);                           \
  }
#define MG_READ_SW(par) if constexpr (MX) mx_sw = *(const uint32_t*)(smem + mx_rd_w + (par) * 2048)
#define MG_READ_SA(par, s) if constexpr (MX) mx_sa = *(const uint32_t*)(smem + mx_rd_a + (par) * 2048 + (s) * 64)

  // ---- fragment readers (single register set) ----
  const int fsw = (li >> 1) & 7;
  const int a_rd0 = (wr * 128 + li) * 128 + ((lq ^ fsw) << 4);          // k-substep 0; substep 1 = ^ 64
  const int b_rd0 = (WLAYOUT == MG_W_ROWMAJOR) ? G256_TILE + (wc * 64 + li) * 128 + ((lq ^ fsw) << 4)
                                                : G256_TILE + (wc * 8) * 1024 + lane * 16;
  constexpr int B_NT = (WLAYOUT == MG_W_ROWMAJOR) ? 16 * 128 : 2 * 1024;
  constexpr int B_KS = (WLAYOUT == MG_W_ROWMAJOR) ? 0 : 1024;           // row-major: substep via XOR 64
  // fragments as 8-register tuples: halves [0..3] / [4..7] = k-substeps 0 / 1 (two bf16 MFMA operands, or one fp8 operand)
  i32x8 af[4];                     // current m-half: [m-tile]
  i32x8 bw[2][2];                  // both n-halves:  [n-half][n-tile]
#define MG_READ_A(par, mh)                                                                               \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    if (ABL != 2 || abl_on)                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                     \
      af[i_] = __builtin_shufflevector(*(const i32x4*)(sb_ + a_rd0 + ((mh) * 4 + i_) * 2048),            \
                                       *(const i32x4*)(sb_ + (a_rd0 ^ 64) + ((mh) * 4 + i_) * 2048), 0, 1, 2, 3, 4, 5, 6, 7); \
  }
#define MG_READ_B(par, nh)                                                                               \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    if (ABL != 2 || abl_on)                                                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                     \
      bw[nh][j_] = __builtin_shufflevector(                                                              \
          *(const i32x4*)(sb_ + b_rd0 + ((nh) * 2 + j_) * B_NT),                                         \
          *(const i32x4*)(sb_ + ((WLAYOUT == MG_W_ROWMAJOR) ? (b_rd0 ^ 64) : (b_rd0 + B_KS)) + ((nh) * 2 + j_) * B_NT), 0, 1, 2, 3, 4, 5, 6, 7); \
  }
  // the same read into an explicit register slot: the balanced bf16 schedule keeps n-half 0 of EVEN tiles in bw[0] and of ODD
  // tiles in bw[1] (it is read one phase before the tile starts, while the other slot is still in use)
#define MG_READ_BS(par, nh, slot)                                                                        \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    if (ABL != 2 || abl_on)                                                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                     \
      bw[slot][j_] = __builtin_shufflevector(                                                            \
          *(const i32x4*)(sb_ + b_rd0 + ((nh) * 2 + j_) * B_NT),                                         \
          *(const i32x4*)(sb_ + ((WLAYOUT == MG_W_ROWMAJOR) ? (b_rd0 ^ 64) : (b_rd0 + B_KS)) + ((nh) * 2 + j_) * B_NT), 0, 1, 2, 3, 4, 5, 6, 7); \
  }
#define MG_HALF(v, s_) __builtin_bit_cast(bf16x8, (s_) ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3))

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define MG_MMA(mh, nh)                                                                                   \
  {                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    if (ABL == 3) {                                                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("" ::"v"(af[i_]));                   \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) asm volatile("" ::"v"(bw[nh][j_]));               \
    } else                                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                 \
          acc[(mh) * 4 + i_][(nh) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
              MG_HALF(bw[nh][j_], s_), MG_HALF(af[i_], s_), acc[(mh) * 4 + i_][(nh) * 2 + j_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  }
#define MG_MMAS(mh, nh, slot)                                                                            \
  {                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    if (ABL == 3) {                                                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("" ::"v"(af[i_]));                   \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) asm volatile("" ::"v"(bw[slot][j_]));             \
    } else                                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                 \
          acc[(mh) * 4 + i_][(nh) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
              MG_HALF(bw[slot][j_], s_), MG_HALF(af[i_], s_), acc[(mh) * 4 + i_][(nh) * 2 + j_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  }
  // ---- 32x32x16 variant: fragments [tile][k-substep], same regions per phase as the 16x16 readers above ----
  const int l31 = lane & 31, l5 = lane >> 5;
  const int fsw32 = (l31 >> 1) & 7;
  const int a32_rd0 = (wr * 128 + l31) * 128 + ((l5 ^ fsw32) << 4);     // k-substep s: ^ (s << 5); m32-tile t: + t * 4096
  const int b32_rd0 = (WLAYOUT == MG_W_ROWMAJOR) ? G256_TILE + (wc * 64 + l31) * 128 + ((l5 ^ fsw32) << 4)
                                                  : G256_TILE + ((wc * 4 + (l31 >> 4)) * 2) * 1024 + (l5 * 16 + (lane & 15)) * 16;
  bf16x8 af32[2][4];               // current m-half: [m32-tile][k-substep]
  bf16x8 bw32[2][4];               // both n-halves (one n32-tile each): [n-half][k-substep]
  f32x16 acc32[4][2];
  if constexpr (MFMA32) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  }
#define MG_READ_A32(par, mh)                                                                             \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                     \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                   \
        af32[i_][s_] = *(const bf16x8*)(sb_ + (a32_rd0 ^ (s_ << 5)) + ((mh) * 2 + i_) * 4096);           \
  }
#define MG_READ_B32(par, nh)                                                                             \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                     \
      bw32[nh][s_] = *(const bf16x8*)(sb_ + ((WLAYOUT == MG_W_ROWMAJOR) ? ((b32_rd0 ^ (s_ << 5)) + (nh) * 4096) \
                                                                        : (b32_rd0 + (nh) * 4096 + (s_ >> 1) * 1024 + (s_ & 1) * 512))); \
  }
#define MG_MMA32(mh, nh)                                                                                 \
  {                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                   \
        acc32[(mh) * 2 + i_][nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw32[nh][s_], af32[i_][s_],   \
                                                                          acc32[(mh) * 2 + i_][nh], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  }
#define MG_PHASE32(READS, DMA, MH, NH)                                                                   \
  {                                                                                                      \
    READS;                                                                                               \
    DMA;                                                                                                 \
    MG_WAIT_LGKM0();                                                                                     \
    MG_BAR();                                                                                            \
    MG_MMA32(MH, NH);                                                                                    \
    MG_BAR();                                                                                            \
  }
#define MG_BAR() __builtin_amdgcn_s_barrier()
  // one phase: read segment (with the DMA issue / wait given as statements), barrier, MFMA segment, barrier
#define MG_PHASE(READS, DMA, MH, NH)                                                                     \
  {                                                                                                      \
    READS;                                                                                               \
    DMA;                                                                                                 \
    if (!LATE_LGKM) MG_WAIT_LGKM0();                                                                     \
    MG_BAR();                                                                                            \
    if (LATE_LGKM) MG_WAIT_LGKM0();                                                                      \
    MG_MMA(MH, NH);                                                                                      \
    MG_BAR();                                                                                            \
  }
#define MG_PHASES(READS, DMA, MH, NH, SLOT)                                                              \
  {                                                                                                      \
    READS;                                                                                               \
    DMA;                                                                                                 \
    MG_WAIT_LGKM0();                                                                                     \
    MG_BAR();                                                                                            \
    MG_MMAS(MH, NH, SLOT);                                                                               \
    MG_BAR();                                                                                            \
  }
  // FP8 phases: the 8-register operands leave no room for four A and four W fragments next to 128 accumulators, so a
  // phase is one m-QUARTER (two m-tiles) against all four n-tiles: W is read once per K-tile (q0), A two tiles per phase.
  // DMA units as on the bf16 path (A_0 / A_1 = quarters 0,1 / 2,3 of both groups, W_0 / W_1), restaged the phase after their
  // last read: W (all of it read in q0) at q1 / q2, A_0 (last read q1) at q3, A_1 (last read q3) at q0 of the next tile;
  // vmcnt(6) at q3 retires tile t+1 (A_1(t+1), issued in q0 of t, is the youngest of it: 3 phases of flight).
#define MG_READ_AQ(par, mq)                                                                              \
  {                                                                                                      \
    const char* sb_ = smem + (par) * G256_BUF;                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                     \
      af[i_] = __builtin_shufflevector(*(const i32x4*)(sb_ + a_rd0 + ((mq) * 2 + i_) * 2048),            \
                                       *(const i32x4*)(sb_ + (a_rd0 ^ 64) + ((mq) * 2 + i_) * 2048), 0, 1, 2, 3, 4, 5, 6, 7); \
  }
#define MG_MMAQ(mq)                                                                                      \
  {                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                     \
      _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_)                                                   \
        if constexpr (MX) acc[(mq) * 2 + i_][n_] = mfma_mx_k128(bw[n_ >> 1][n_ & 1], n_, mx_sw, af[i_], ((mq) * 2 + i_) & 3, mx_sa, acc[(mq) * 2 + i_][n_]); \
        else acc[(mq) * 2 + i_][n_] = mfma_fp8_k128(bw[n_ >> 1][n_ & 1], af[i_], acc[(mq) * 2 + i_][n_]); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
  }
#define MG_PHASEQ(READS, DMA, MQ)                                                                        \
  {                                                                                                      \
    READS;                                                                                               \
    DMA;                                                                                                 \
    MG_WAIT_LGKM0();                                                                                     \
    MG_BAR();                                                                                            \
    MG_MMAQ(MQ);                                                                                         \
    MG_BAR();                                                                                            \
  }
  // K-tile t with buffer parity PAR; NXT: tile t+1 exists, NXT2: tile t+2 exists
#define MG_G256_TILE(t, PAR, NXT, NXT2)                                                                  \
  if constexpr (FP8) {                                                                                   \
    /* all of W is read in q0, the A quarters in q0..q3: A_0 (quarters 0, 1) is free from q2, A_1 from q0 of the next tile */ \
    MG_PHASEQ({ MG_READ_AQ(PAR, 0); MG_READ_B(PAR, 0); MG_READ_B(PAR, 1); MG_READ_SW(PAR); MG_READ_SA(PAR, 0); }, { if (NXT) MG_DMA_A2((t) + 1, 1); }, 0); \
    MG_PHASEQ({ MG_READ_AQ(PAR, 1); }, { if (NXT2) MG_DMA_B2((t) + 2, 0); }, 1);                          \
    MG_PHASEQ({ MG_READ_AQ(PAR, 2); MG_READ_SA(PAR, 1); }, { if (NXT2) MG_DMA_B2((t) + 2, 1); }, 2);      \
    MG_PHASEQ({ MG_READ_AQ(PAR, 3); }, { if (NXT2) { MG_DMA_A2((t) + 2, 0); MG_DMA_S((t) + 2); if constexpr (MX) MG_WAIT_VM(7); else MG_WAIT_VM(6); } else { MG_WAIT_VM(0); } }, 3); \
  } else if constexpr (MFMA32) {                                                                         \
    MG_PHASE32({ MG_READ_A32(PAR, 0); MG_READ_B32(PAR, 0); }, { if (NXT) MG_DMA_A2((t) + 1, 1); }, 0, 0); \
    MG_PHASE32({ MG_READ_B32(PAR, 1); }, { if (NXT2) MG_DMA_A2((t) + 2, 0); }, 0, 1);                     \
    MG_PHASE32({ MG_READ_A32(PAR, 1); }, { if (NXT2) MG_DMA_B2((t) + 2, 0); }, 1, 1);                     \
    MG_PHASE32({}, { if (NXT2) { MG_DMA_B2((t) + 2, 1); MG_WAIT_VM(6); } else { MG_WAIT_VM(0); } }, 1, 0); \
  } else if constexpr (BALANCED) {                                                                       \
    /* fragment reads 8 / 4 / 8 / 4 per phase instead of 12 / 4 / 8 / 0: W(nh0) of tile t+1 is read in q3 of tile t, into   \
       the register slot W(nh1) of tile t has just left (slots swap roles every tile: PAR = slot of nh0).  Its DMA unit     \
       W_0(t+1), issued in q2 of t-1, is retired one phase earlier for that: vmcnt(8) in q2 leaves the four younger units    \
       in flight (W_1(t+1), A_1(t+1), A_0(t+2), W_0(t+2)).                                                                    */ \
    MG_PHASES({ MG_READ_A(PAR, 0); }, { if (NXT) MG_DMA_A2((t) + 1, 1); }, 0, 0, PAR);                    \
    MG_PHASES({ MG_READ_BS(PAR, 1, 1 - (PAR)); }, { if (NXT2) MG_DMA_A2((t) + 2, 0); }, 0, 1, 1 - (PAR)); \
    MG_PHASES({ MG_READ_A(PAR, 1); }, { if (NXT2) { MG_DMA_B2((t) + 2, 0); MG_WAIT_VM(8); } else if (NXT) { MG_WAIT_VM(4); } }, 1, 1, 1 - (PAR)); \
    MG_PHASES({ if (NXT) MG_READ_BS(1 - (PAR), 0, 1 - (PAR)); }, { if (NXT2) { MG_DMA_B2((t) + 2, 1); MG_WAIT_VM(6); } else { MG_WAIT_VM(0); } }, 1, 0, PAR); \
  } else if constexpr (EARLY) {                                                                          \
    MG_PHASE({ MG_READ_A(PAR, 0); MG_READ_B(PAR, 0); }, { if (NXT) MG_DMA_A2((t) + 1, 1); }, 0, 0);      \
    MG_PHASE({ MG_READ_B(PAR, 1); }, { if (NXT2) MG_DMA_A2((t) + 2, 0); }, 0, 1);                         \
    MG_PHASE({ MG_READ_A(PAR, 1); }, { if (NXT2) MG_DMA_B2((t) + 2, 0); }, 1, 1);                         \
    MG_PHASE({}, { if (NXT2) { MG_DMA_B2((t) + 2, 1); MG_WAIT_VM(6); } else { MG_WAIT_VM(0); } }, 1, 0);  \
  } else {                                                                                               \
    MG_PHASE({ MG_READ_A(PAR, 0); MG_READ_B(PAR, 0); }, { if (NXT) MG_DMA_B((t) + 1, 1); }, 0, 0);       \
    MG_PHASE({ MG_READ_B(PAR, 1); }, { if (NXT) MG_DMA_A((t) + 1, 0); }, 0, 1);                           \
    MG_PHASE({ MG_READ_A(PAR, 1); }, { if (NXT) MG_DMA_A((t) + 1, 1); }, 1, 1);                           \
    MG_PHASE({}, { if (NXT2) { MG_DMA_B((t) + 2, 0); MG_WAIT_VM(2); } else { MG_WAIT_VM(0); } }, 1, 0);   \
  }

  // ---- prologue: tile 0 complete, W_lo(1) in flight; group 1 starts one barrier late ----
  if constexpr (EARLY) {     // tile 0 complete; A_0, W_0, W_1 of tile 1 in flight (A_1(1) follows in q0 of tile 0; every path)
    MG_DMA_B2(0, 0); MG_DMA_B2(0, 1); MG_DMA_A2(0, 0); MG_DMA_S(0); MG_DMA_A2(0, 1);
    MG_DMA_A2(1, 0); MG_DMA_S(1); MG_DMA_B2(1, 0); MG_DMA_B2(1, 1);
    if constexpr (MX) MG_WAIT_VM(7); else MG_WAIT_VM(6);
  } else {
    MG_DMA_B(0, 0); MG_DMA_B(0, 1); MG_DMA_A(0, 0); MG_DMA_A(0, 1);
    MG_DMA_B(1, 0);
    MG_WAIT_VM(2);
  }
  MG_BAR();
  if (wr == 1) MG_BAR();
  if constexpr (BALANCED) { MG_READ_BS(0, 0, 0); }      // W(nh0) of tile 0: later tiles get theirs in q3 of the tile before

  int t = 0;
  for (; t + 2 < nkt; t += 2) {
    MG_G256_TILE(t, 0, true, true)
    MG_G256_TILE(t + 1, 1, true, true)
    if (ABL == 2 || ABL == 4) abl_on = false;
  }
  MG_G256_TILE(t, 0, true, false)
  MG_G256_TILE(t + 1, 1, false, false)
  if (wr == 0) MG_BAR();
#undef MG_G256_TILE
#undef MG_DMA_S
#undef MG_READ_SW
#undef MG_READ_SA
#undef MG_PHASE32
#undef MG_PHASES
#undef MG_MMAS
#undef MG_READ_BS
#undef MG_MMA32
#undef MG_READ_A32
#undef MG_READ_B32
#undef MG_PHASEQ
#undef MG_MMAQ
#undef MG_READ_AQ
#undef MG_PHASE
#undef MG_MMA
#undef MG_READ_A
#undef MG_READ_B
#undef MG_HALF
#undef MG_DMA_A
#undef MG_DMA_B
#undef MG_DMA_A2
#undef MG_DMA_B2
#undef KT_SRC

  if constexpr (ABL == 7) {      // timing ablation: no epilogue (nothing is stored)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // ---- epilogue: two passes of 128 tile rows (each wave's upper / lower 64) through LDS ----
  // Barriers here are RAW s_barrier + lgkmcnt(0): the hazards are LDS-only (staging writes vs row reads), and __syncthreads()
  // would add s_waitcnt vmcnt(0) -- with the stores of pass 0 outstanding that is a full store drain in the middle of the
  // epilogue (vmcnt counts stores on this chip).  For the same reason the per-column vectors are loaded once, before pass 0.
  mg_epilogue ep = p.ep;                      // split-K: the host's slab epilogue, moved to this split's slab
  if constexpr (SPLITK) ep.C = (float*)ep.C + (int64_t)sp * p.M * p.ldws;
  const bool wide = epilogue_wide_ok(ep);   // 16-byte accesses when every row start allows it
  // ABL 11: 100-MHz time stamps of wave 0 into p.ws[blockIdx.x * 8 ..] (start, K loop done, staged 0, walked 0, staged 1, walked 1, drained)
  unsigned long long* stamps = (unsigned long long*)p.ws + (size_t)blockIdx.x * 8;
#define MG_STAMP(i) if (ABL == 11 && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime()
  if (ABL == 11 && tid == 0) stamps[0] = t_start;
  MG_STAMP(1);
  MG_WAIT_LGKM0();
  MG_BAR();                                   // every wave has read its last fragments: the K-tile buffers are free
  // (a generic lambda called with two compile-time halves: a `for (h)` loop around this much code is not unrolled by
  //  hipcc any more, and acc[h * 4 + i] with a run-time h puts the 128 accumulators in scratch)
  auto stage = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
    if constexpr (MFMA32) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16& a = acc32[h * 2 + i][j];
            const f32x4 v = {a[g * 4], a[g * 4 + 1], a[g * 4 + 2], a[g * 4 + 3]};
            *(f32x4*)(smem + (wr * 64 + i * 32 + l31) * EPI256_ROWB + (wc * 64 + j * 32 + g * 8 + l5 * 4) * 4) = v;
          }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(f32x4*)(smem + (wr * 64 + i * 16 + li) * EPI256_ROWB + (wc * 64 + j * 16 + lq * 4) * 4) = acc[h * 4 + i][j];
    }
    MG_WAIT_LGKM0();
    MG_BAR();
  };
  // (timing ablations: 8 = LDS staging only, no row walk; 10 = every tile stores to the rows / columns of tile (0, 0))
  const int mb0 = (ABL == 10 ? 0 : m0), nb = ABL == 10 ? 0 : n0, mlim = ABL == 8 ? 0 : p.M;
  auto both = [&](auto wc_, auto ntc_, auto fullc_) {
    constexpr int W = decltype(wc_)::value;
    constexpr bool NT = decltype(ntc_)::value;
    constexpr bool FULL = decltype(fullc_)::value;      // interior column tile: no per-element tail code
    EpiColsW<W> c;
    const int n = nb + (lane % (256 / W)) * W;
    if (n < p.N) epilogue_cols<W>(ep, n, p.N, c);
    // an epilogue without aux / residual operands walks its rows in a build of the loop that contains no global load
    // (interior bf16-output tiles only: W == 8 and FULL): nothing in it waits for the stores of the rows before
    const bool loads = ep.aux_mode != MG_AUX_NONE || ep.res0 || ep.res1 || ep.res2 || ep.accumulate;
    stage(std::integral_constant<int, 0>{});
    MG_STAMP(2);
    if (FULL && W == 8 && !loads) epilogue_rows_c<256, EPI256_ROWB, W, NT, FULL, false, Q8>(ep, c, smem, 128, 8, wave, lane, mb0, 128, nb, mlim, p.N, p.row_scale);
    else epilogue_rows_c<256, EPI256_ROWB, W, NT, FULL, true, Q8>(ep, c, smem, 128, 8, wave, lane, mb0, 128, nb, mlim, p.N, p.row_scale);
    MG_WAIT_LGKM0();
    MG_BAR();                                 // pass 0 has been read
    MG_STAMP(3);
    stage(std::integral_constant<int, 1>{});
    MG_STAMP(4);
    if (FULL && W == 8 && !loads) epilogue_rows_c<256, EPI256_ROWB, W, NT, FULL, false, Q8>(ep, c, smem, 128, 8, wave, lane, mb0 + 64, 128, nb, mlim, p.N, p.row_scale);
    else epilogue_rows_c<256, EPI256_ROWB, W, NT, FULL, true, Q8>(ep, c, smem, 128, 8, wave, lane, mb0 + 64, 128, nb, mlim, p.N, p.row_scale);
    MG_STAMP(5);
    if (ABL == 11) {
      MG_WAIT_VM(0);
      MG_STAMP(6);
      // shader cycles of this workgroup's life: / its wall time (stamps[6] - stamps[0], 100 MHz) = the clock the kernel ran at
      if (tid == 0) stamps[7] = __builtin_amdgcn_s_memtime() - c_start;
    }
  };
  const bool interior = nb + 256 <= p.N;
  if (!wide) both(std::integral_constant<int, 4>{}, std::false_type{}, std::false_type{});
  else if (p.nt && interior) both(std::integral_constant<int, 8>{}, std::true_type{}, std::true_type{});
  else if (interior) both(std::integral_constant<int, 8>{}, std::false_type{}, std::true_type{});
  else if (p.nt) both(std::integral_constant<int, 8>{}, std::true_type{}, std::false_type{});
  else both(std::integral_constant<int, 8>{}, std::false_type{}, std::false_type{});
#undef MG_STAMP
}

template <int WAVES, int KC, int NT, bool W8 = false, bool PIPE = false>
__global__ __launch_bounds__(WAVES * 64) void skinny_kernel(const SkinnyParams p) {
  __shared__ __attribute__((aligned(16))) char lds[skinny_lds_bytes<WAVES, NT>()];
  skinny_body<WAVES, KC, NT, W8, false, NoWait, PIPE>(p, blockIdx.x, lds);
}

// two independent GEMVs (same variant) in ONE launch: blocks [0, g0) work on p0, the rest on p1.
// Decode uses it for out_proj || adapter-down: the 64-workgroup adapter GEMV hides under the other.
template <int WAVES, int KC, int NT, bool W8 = false, bool PIPE = false>
__global__ __launch_bounds__(WAVES * 64) void skinny2_kernel(const SkinnyParams p0, const SkinnyParams p1, int g0) {
  __shared__ __attribute__((aligned(16))) char lds[skinny_lds_bytes<WAVES, NT>()];
  if ((int)blockIdx.x < g0) skinny_body<WAVES, KC, NT, W8, false, NoWait, PIPE>(p0, blockIdx.x, lds);
  else skinny_body<WAVES, KC, NT, W8, false, NoWait, PIPE>(p1, blockIdx.x - g0, lds);
}

int check_epilogue(const mg_epilogue& ep, const char* who, bool tile_gemm = false, int N = 0) {
  if (!ep.C && !(tile_gemm && ep.C8)) MG_FAIL(MG_ERR_SHAPE, "%s: null output", who);
  if (ep.C8) {
    if (!tile_gemm) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the MX e4m3 output copy (mg_epilogue.C8) exists for the tile GEMMs only", who);
    if (!ep.c8_scales || ep.c8_rgroups <= 0 || (N & 31) || ep.ldc8 != ((N + 127) / 128) * 128 || ((uintptr_t)ep.C8 & 7) || ep.out_f32)
      MG_FAIL(MG_ERR_SHAPE, "%s: C8 needs c8_scales, c8_rgroups = ceil(M / 64), N %% 32 == 0, ldc8 == ceil(N / 128) * 128, an 8-byte aligned C8 and a bf16 (or no) C", who);
    if ((ep.ldc & 7) || (ep.C2 && (ep.ldc2 & 7)) || (ep.aux_mode != MG_AUX_NONE && (ep.ldaux & 7)) || ((ep.res0 || ep.res1 || ep.res2) && (ep.ldr & 7)))
      MG_FAIL(MG_ERR_ALIGN, "%s: C8 needs the 8-column epilogue walk: every leading dimension a multiple of 8", who);
  }
  if ((ep.ldc & 3) != 0) MG_FAIL(MG_ERR_ALIGN, "%s: ldc must be a multiple of 4", who);
  if ((ep.res0 || ep.res1 || ep.res2) && (ep.ldr & 3) != 0) MG_FAIL(MG_ERR_ALIGN, "%s: ldr must be a multiple of 4", who);
  if (!MG_ALIGNED16(ep.C) || !MG_ALIGNED16(ep.scale) || !MG_ALIGNED16(ep.bias) || !MG_ALIGNED16(ep.res0) ||
      !MG_ALIGNED16(ep.res1) || !MG_ALIGNED16(ep.res2))
    MG_FAIL(MG_ERR_ALIGN, "%s: epilogue pointers must be 16-byte aligned", who);
  if (ep.act < 0 || ep.act > 3 || ep.act_after < 0 || ep.act_after > 1) MG_FAIL(MG_ERR_SHAPE, "%s: bad activation code", who);
  if (ep.act_n0 < 0 || (ep.act_n0 & 7)) MG_FAIL(MG_ERR_SHAPE, "%s: act_n0 must be a non-negative multiple of 8", who);
  if (ep.aux_mode < 0 || ep.aux_mode > 4) MG_FAIL(MG_ERR_SHAPE, "%s: bad aux_mode", who);
  if (ep.aux_mode != MG_AUX_NONE && (!ep.aux || (ep.ldaux & 3) || !MG_ALIGNED16(ep.aux))) MG_FAIL(MG_ERR_ALIGN, "%s: aux operand must be 16-byte aligned with ldaux %% 4 == 0", who);
  if (ep.C2 && ((ep.ldc2 & 3) || !MG_ALIGNED16(ep.C2))) MG_FAIL(MG_ERR_ALIGN, "%s: C2 must be 16-byte aligned with ldc2 %% 4 == 0", who);
  if (ep.accumulate && (!ep.out_f32 || ep.C8 || !ep.C)) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: accumulate adds into an fp32 output (out_f32, no MX copy)", who);
  if (ep.row_scale && (!tile_gemm || ((uintptr_t)ep.row_scale & 3))) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: mg_epilogue.row_scale is an fp32 [M] vector of the tile GEMMs", who);
  return MG_OK;
}

template <int AMODE, int WLAYOUT, bool FP8 = false, bool MX = false>
int launch_gemm(const GemmParams& gp, hipStream_t s) {
  if (int rc = mg_allow_dynamic_lds((const void*)gemm128_kernel<AMODE, WLAYOUT, FP8, MX>, GEMM_LDS, "mg_gemm")) return rc;
  hipLaunchKernelGGL((gemm128_kernel<AMODE, WLAYOUT, FP8, MX>), dim3(gp.tiles_m * gp.tiles_n * gp.splits), dim3(256), GEMM_LDS, s, gp);
  MG_CHECK_LAUNCH();
  if (gp.splits > 1) {
    const int64_t quads = (int64_t)gp.M * ((gp.N + 3) >> 2);
    hipLaunchKernelGGL(splitk_fixup_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, gp.ws, gp.splits, gp.M, gp.N, gp.ldws, gp.ep, gp.row_scale);
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

// Row-tiles per group of the 256x256 kernel's tile walk.  An XCD runs 32 consecutive tiles of the walk at a time, i.e. a block
// of group_m row-tiles x 32 / group_m column-tiles whose A and W panels its L2 shares.  MAGMA_G256_GROUP_M overrides (tuning).
int group_m_256(int tiles_m, int tiles_n, int K) {
  const char* e = getenv("MAGMA_G256_GROUP_M");
  if (e && atoi(e) > 0) return atoi(e);
  // measured (tools/kbench.py group, profiles/r03_kbench_gemm256_*): 4 x 8 blocks are at or above every other shape for
  // K <= 8192; with a long K and few column tiles, 2 row-tiles x ALL column tiles stream A exactly once (1 GB at the fc_out
  // shape, more than the 256 MiB Infinity Cache can hand from one column block to the next): +8 % there
  if (K >= 8192 && tiles_n <= 16) return 32 / tiles_n > 0 ? 32 / tiles_n : 1;
  return 4;
}

// the automatic rule for the un-split 256x256 kernel (enough tiles to fill the chip)
inline bool want256_noforce(int tile_hint, int64_t wgs256, int M, int N) { return tile_hint == 0 && wgs256 >= 192 && M >= 1024 && N >= 512; }

template <int WLAYOUT, bool LATE_LGKM, bool FP8 = false, bool MFMA32 = false, int ABL = 0, bool SPLITK = false, bool MX = false, bool Q8 = false>
int launch_gemm256(GemmParams gp, hipStream_t s) {
  constexpr int LDS = G256_LDS + (MX ? 4096 : 0);        // + the block scales of two K-tiles
  if (int rc = mg_allow_dynamic_lds((const void*)gemm256_kernel<WLAYOUT, LATE_LGKM, FP8, MFMA32, ABL, SPLITK, MX, Q8>, LDS, "mg_gemm")) return rc;
  if (SPLITK != (gp.splits > 1)) MG_FAIL(MG_ERR_SHAPE, "mg_gemm: internal: split-K form of the 256x256 kernel selected inconsistently");
  gp.tiles_m = (gp.M + 255) / 256; gp.tiles_n = (gp.N + 255) / 256;
  gp.group_m = group_m_256(gp.tiles_m, gp.tiles_n, gp.K);
  gp.a_kt = 64;
  const mg_epilogue real_ep = gp.ep;
  const float* const real_row_scale = gp.row_scale;
  if (gp.splits > 1) {      // the kernel writes raw fp32 partial sums: slab `sp` = ws + sp * M * ldws (rows of ldws floats)
    mg_epilogue slab;
    memset(&slab, 0, sizeof(slab));
    slab.C = gp.ws; slab.ldc = gp.ldws; slab.out_f32 = 1;
    gp.ep = slab;
    gp.row_scale = nullptr;   // the fix-up applies it, once
    gp.nt = 0;
  }
  hipLaunchKernelGGL((gemm256_kernel<WLAYOUT, LATE_LGKM, FP8, MFMA32, ABL, SPLITK, MX, Q8>), dim3(gp.tiles_m * gp.tiles_n * gp.splits), dim3(512), LDS, s, gp);
  MG_CHECK_LAUNCH();
  if (gp.splits > 1) {
    const int64_t quads = (int64_t)gp.M * ((gp.N + 3) >> 2);
    hipLaunchKernelGGL(splitk_fixup_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, gp.ws, gp.splits, gp.M, gp.N, gp.ldws, real_ep, real_row_scale);
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

template <int WAVES, int KC, int NT, bool W8 = false, bool PIPE = false>
int launch_skinny(const SkinnyParams& sp, hipStream_t s) {
  const int grid = (sp.ntiles + NT - 1) / NT;
  hipLaunchKernelGGL((skinny_kernel<WAVES, KC, NT, W8, PIPE>), dim3(grid), dim3(WAVES * 64), 0, s, sp);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

}  // namespace

namespace {
// Shared by mg_gemm_bf16 and mg_gemm_fp8: validation, tile / split-K policy, launch.  For fp8 the descriptor counts
// e4m3 elements; the kernels see pairs of them (K/2, lda/2, ldw/2) -- same byte images, see gemm128_kernel.
struct MxScales { const uint32_t* a; const uint32_t* w; };
int gemm_dispatch(const mg_gemm_desc* d, bool fp8, const float* row_scale, hipStream_t s, const char* who, const MxScales* mx = nullptr) {
  if (!d) MG_FAIL(MG_ERR_SHAPE, "%s: null descriptor", who);
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) MG_FAIL(MG_ERR_SHAPE, "%s: M,N,K must be positive (%d,%d,%d)", who, d->M, d->N, d->K);
  const int epb = fp8 ? 2 : 1;               // descriptor elements per kernel element
  if (d->K & (8 * epb - 1)) MG_FAIL(MG_ERR_SHAPE, "%s: K=%d must be a multiple of %d", who, d->K, 8 * epb);
  if (!d->A || !d->W || !d->zero_page) MG_FAIL(MG_ERR_SHAPE, "%s: null A/W/zero_page", who);
  if (!MG_ALIGNED16(d->A) || !MG_ALIGNED16(d->W) || !MG_ALIGNED16(d->zero_page)) MG_FAIL(MG_ERR_ALIGN, "%s: A/W/zero_page must be 16-byte aligned", who);
  if (int rc = check_epilogue(d->ep, who, true, d->N)) return rc;
  if (d->ep.C8 && d->ep.c8_rgroups != (d->M + 63) / 64) MG_FAIL(MG_ERR_SHAPE, "%s: c8_rgroups must be ceil(M / 64)", who);
  if (d->ep.C8 && d->split_k > 1) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the MX output copy is written by the un-split kernels", who);
  mg_gemm_desc unsplit;
  if (d->ep.C8 && d->split_k != 1) { unsplit = *d; unsplit.split_k = 1; d = &unsplit; }      // no automatic split-K either
  if (fp8 && d->a_mode != MG_A_DENSE) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: fp8 operands need a dense A", who);
  GemmParams gp;
  gp.A = d->A; gp.lda = d->lda / epb; gp.W = d->W; gp.ldw = d->ldw / epb;
  gp.M = d->M; gp.N = d->N; gp.K = d->K / epb;
  gp.H = d->H; gp.Wd = d->Wd; gp.Cin = d->Cin;
  gp.zero = d->zero_page;
  gp.tiles_m = (d->M + BM - 1) / BM; gp.tiles_n = (d->N + BN - 1) / BN;
  gp.ep = d->ep;
  if (row_scale && d->ep.row_scale) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the fp8 GEMMs take the row scale as their argument, not through mg_epilogue.row_scale", who);
  if (mx && d->ep.row_scale) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: MX operands carry their scales; mg_epilogue.row_scale is not applied", who);
  gp.row_scale = row_scale ? row_scale : d->ep.row_scale;
  gp.mx_a = mx ? mx->a : nullptr; gp.mx_w = mx ? mx->w : nullptr; gp.rg_a = (d->M + 63) / 64; gp.rg_w = (d->N + 63) / 64;
  // outputs far larger than the caches are streamed out with non-temporal stores (measured -9 % on the
  // 256x256 kernel at K = 4096: the tile no longer evicts the operand panels from L2)
  gp.nt = (int64_t)d->M * d->N * (d->ep.out_f32 ? 4 : 2) >= (int64_t)64 << 20;
  gp.splits = 1; gp.kt_per = (gp.K + BK - 1) / BK; gp.ws = nullptr; gp.ldws = 0;
  if (d->split_k < 0 || d->split_k > 64) MG_FAIL(MG_ERR_SHAPE, "%s: split_k must be in [0, 64]", who);
  if (d->split_k > 1 && !d->workspace) MG_FAIL(MG_ERR_SHAPE, "%s: split_k > 1 needs a workspace", who);
  if (d->workspace && !MG_ALIGNED16(d->workspace)) MG_FAIL(MG_ERR_ALIGN, "%s: workspace must be 16-byte aligned", who);
  if (d->a_mode == MG_A_DENSE) {
    if (d->lda & (8 * epb - 1)) MG_FAIL(MG_ERR_ALIGN, "%s: lda must be a multiple of %d", who, 8 * epb);
    if (d->lda < d->K) MG_FAIL(MG_ERR_SHAPE, "%s: lda < K", who);
  } else if (d->a_mode == MG_A_CONV3X3) {
    if (d->Cin <= 0 || (d->Cin & 7) || d->K != 9 * d->Cin) MG_FAIL(MG_ERR_SHAPE, "%s: conv3x3 needs Cin%%8==0 and K==9*Cin", who);
    if (d->H <= 0 || d->Wd <= 0 || d->M % (d->H * d->Wd)) MG_FAIL(MG_ERR_SHAPE, "%s: conv3x3 needs M == B*H*W", who);
  } else MG_FAIL(MG_ERR_SHAPE, "%s: bad a_mode %d", who, d->a_mode);
  if (d->w_layout == MG_W_ROWMAJOR) {
    if ((d->ldw & (8 * epb - 1)) || d->ldw < d->K) MG_FAIL(MG_ERR_ALIGN, "%s: ldw must be >= K and a multiple of %d", who, 8 * epb);
  } else if (d->w_layout == MG_W_FRAGTILED) {
    if ((d->ldw & (64 * epb - 1)) || d->ldw < d->K) MG_FAIL(MG_ERR_ALIGN, "%s: tiled weights need Kp (ldw) %%%d==0 and >= K", who, 64 * epb);
  } else MG_FAIL(MG_ERR_SHAPE, "%s: bad w_layout %d", who, d->w_layout);
  const bool rm = d->w_layout == MG_W_ROWMAJOR;
  // large dense shapes go to the deep-pipelined 256x256 kernel (tile_hint: 0 auto, 128 / 256 force)
  const int64_t wgs256 = (int64_t)((d->M + 255) / 256) * ((d->N + 255) / 256);
  const bool can256 = d->a_mode == MG_A_DENSE && (gp.K % 128) == 0;           // gp.K counts PAIRS of fp8 values on the fp8 path
  const bool want256 = (d->tile_hint >= 256 && d->tile_hint <= 272) ||
                       (d->tile_hint == 0 && wgs256 >= 192 && d->M >= 1024 && d->N >= 512);
  // bf16: 32x32x16 MFMA (tile_hint 258) or 16x16x32 (259); 0 / 256 follow MAGMA_GEMM256_MFMA (default below)
  static const int mfma_env = [] { const char* e = getenv("MAGMA_GEMM256_MFMA"); return e ? atoi(e) : MG_GEMM256_MFMA_DEFAULT; }();
  const bool mfma32 = !fp8 && (d->tile_hint == 258 || (d->tile_hint != 259 && d->tile_hint != 257 && mfma_env == 32));
  // 256x256 kernel with split-K: too few 256x256 tiles to fill the chip but a LONG contraction (the adapters' weight gradients:
  // 4096 x 1024 outputs over K = B*S = 32768 -> 64 tiles x 4 splits = 256 workgroups of 128 K-tiles each).  The 128x128 kernel
  // with two splits ran these at 0.92 PF (298 us, profiles/r04_train_trace_by_grid.txt).  tile_hint 256 + split_k n forces it.
  int split256 = 0;
  static const bool split256_on = [] { const char* e = getenv("MAGMA_G256_SPLITK"); return !e || atoi(e) != 0; }();   // A/B knob
  if ((split256_on || d->tile_hint == 256) && can256 && !fp8 && d->workspace && d->split_k != 1 && d->a_mode == MG_A_DENSE && (d->tile_hint == 0 || d->tile_hint == 256) &&
      !want256_noforce(d->tile_hint, wgs256, d->M, d->N)) {
    const int nkt256 = gp.K >> 6;
    const int64_t slab = (int64_t)d->M * (((d->N + 255) / 256) * 256) * 4;
    int want = d->split_k;
    if (want == 0 && d->tile_hint == 0 && wgs256 >= 32 && wgs256 < 192 && nkt256 >= 256 && d->M > 256 && d->N >= 512) {     // M > 256: the prefill's fc_out at M = 456 (2 x 16 tiles x 8 splits)
      want = 1;
      while (wgs256 * want * 2 <= 320 && want < 8) want *= 2;      // 192 .. 320 workgroups
    }
    if (want > 1 && nkt256 % (2 * want) == 0 && (int64_t)want * slab <= d->workspace_bytes && (d->tile_hint == 256 || want * wgs256 >= 192))
      split256 = want;
    else if (d->tile_hint == 256 && d->split_k > 1)
      MG_FAIL(MG_ERR_SHAPE, "%s: the 256x256 kernel cannot split K=%d %d ways (needs K %% (128 * splits) == 0 and %lld bytes of workspace)", who, d->K, d->split_k, (long long)(want * slab));
  }
  if (split256 > 1) {
    gp.splits = split256; gp.kt_per = (gp.K >> 6) / split256;
    gp.ws = d->workspace; gp.ldws = (int64_t)((d->N + 255) / 256) * 256;
    return rm ? launch_gemm256<MG_W_ROWMAJOR, false, false, false, 0, true>(gp, s) : launch_gemm256<MG_W_FRAGTILED, false, false, false, 0, true>(gp, s);
  }
  if (can256 && want256) {
    if (d->tile_hint >= 261) {      // timing ablations / A-B variants of the bf16 kernel (tools/kbench.py abl, ksweep)
#ifdef MG_GEMM_ABLATIONS            // `make ABL=1`: each one is another copy of the kernel and its epilogue (minutes of compile time)
      if (fp8 || rm) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: ablation builds exist for bf16 fragment-tiled weights only", who);
      switch (d->tile_hint) {
        case 261: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 1>(gp, s);    // DMA always reads K-tiles 0 / 1 (L2-hot)
        case 262: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 2>(gp, s);    // no fragment reads
        case 263: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 3>(gp, s);    // no MFMA
        case 264: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 4>(gp, s);    // no DMA after the prologue
        case 266: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 6>(gp, s);    // first DMA schedule (correct results)
        case 267: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 7>(gp, s);    // no epilogue
        case 268: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 8>(gp, s);    // epilogue: LDS staging only
        case 270: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 10>(gp, s);   // epilogue: all tiles store to tile (0,0)
        case 272: return launch_gemm256<MG_W_FRAGTILED, false, false, false, 12>(gp, s);   // fragment reads 8 / 4 / 8 / 4 per phase instead of 12 / 4 / 8 / 0 (correct results)
        case 271:                                                                          // time stamps into the workspace (8 x uint64 per workgroup)
          if (!d->workspace || d->workspace_bytes < (int64_t)wgs256 * 64) MG_FAIL(MG_ERR_SHAPE, "%s: the stamp build needs 64 bytes of workspace per tile", who);
          gp.ws = d->workspace;
          return launch_gemm256<MG_W_FRAGTILED, false, false, false, 11>(gp, s);
        default: MG_FAIL(MG_ERR_UNSUPPORTED, "%s: no such ablation (tile_hint %d)", who, d->tile_hint);
      }
#else
      MG_FAIL(MG_ERR_UNSUPPORTED, "%s: tile_hint %d is a timing ablation; build the library with `make ABL=1`", who, d->tile_hint);
#endif
    }
    // A/B variants that measured slower (32x32x16 MFMA: -5..10 %, profiles/r03_kbench_gemm256_mfma32_vs_16.jsonl; LDS-read wait
    // after the barrier): ablation library only (`make ABL=1`), not in the product .so
#ifdef MG_GEMM_ABLATIONS
    if (mfma32) return rm ? launch_gemm256<MG_W_ROWMAJOR, false, false, true>(gp, s) : launch_gemm256<MG_W_FRAGTILED, false, false, true>(gp, s);
    if (d->tile_hint == 257 && !fp8)   // experiment: LDS-read wait after the barrier
      return rm ? launch_gemm256<MG_W_ROWMAJOR, true>(gp, s) : launch_gemm256<MG_W_FRAGTILED, true>(gp, s);
#else
    if (mfma32 || (d->tile_hint == 257 && !fp8))
      MG_FAIL(MG_ERR_UNSUPPORTED, "%s: tile_hint %d / MAGMA_GEMM256_MFMA=32 select an A/B variant of the 256x256 kernel that exists only in the ablation library (`make ABL=1`)", who, d->tile_hint);
#endif
    if (d->ep.C8) {      // the build of the fp8 kernels whose epilogue also writes the MX e4m3 copy (fragment-tiled weights)
      if (!fp8 || rm) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the MX output copy is built for the fp8 GEMMs on fragment-tiled weights", who);
      return gp.mx_a ? launch_gemm256<MG_W_FRAGTILED, false, true, false, 0, false, true, true>(gp, s)
                     : launch_gemm256<MG_W_FRAGTILED, false, true, false, 0, false, false, true>(gp, s);
    }
    if (fp8 && gp.mx_a) return rm ? launch_gemm256<MG_W_ROWMAJOR, false, true, false, 0, false, true>(gp, s)
                                  : launch_gemm256<MG_W_FRAGTILED, false, true, false, 0, false, true>(gp, s);
    if (fp8) return rm ? launch_gemm256<MG_W_ROWMAJOR, false, true>(gp, s) : launch_gemm256<MG_W_FRAGTILED, false, true>(gp, s);
    return rm ? launch_gemm256<MG_W_ROWMAJOR, false>(gp, s) : launch_gemm256<MG_W_FRAGTILED, false>(gp, s);
  }
  if (d->tile_hint == 256 && !can256) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the 256x256 kernel needs dense A and K %% %d == 0", who, 128 * epb);
  if (d->ep.C8) MG_FAIL(MG_ERR_UNSUPPORTED, "%s: the MX output copy is written by the 256x256 fp8 kernel (K %% 256 == 0; tile_hint 0 with >= 192 tiles, or 256)", who);
  // split-K: a grid that leaves most of the 256 CUs idle and has a long K loop is cut along K
  // until ~2 workgroups per CU exist (>= 4 K-tiles each); needs the caller's fp32 workspace.
  if (d->workspace && d->split_k != 1) {
    const int nkt = gp.kt_per;
    const int tiles = gp.tiles_m * gp.tiles_n;
    int want = d->split_k;
    // >= 192 tiles fill the chip, but with one 4-wave workgroup per CU a LONG K loop (weight gradients contract over
    // B*S = 32768 rows) still runs at half rate: two splits put two workgroups on every CU (0.51 -> 0.32 ms at 4096x1024x32768)
    if (want == 0) want = tiles >= 192 ? ((tiles < 512 && nkt >= 128) ? 2 : 1) : std::min(std::min(16, nkt / 4), (512 + tiles - 1) / tiles);
    want = std::max(1, std::min(want, nkt));
    const int64_t slab = (int64_t)d->M * gp.tiles_n * BN * 4;
    if ((int64_t)want * slab > d->workspace_bytes) {
      if (d->split_k > 1) MG_FAIL(MG_ERR_SHAPE, "%s: workspace too small for split_k=%d (%lld bytes needed)", who, want, (long long)(want * slab));
      want = (int)std::max<int64_t>(1, d->workspace_bytes / slab);
    }
    if (want > 1) {
      gp.kt_per = (nkt + want - 1) / want;
      gp.splits = (nkt + gp.kt_per - 1) / gp.kt_per;
      gp.ws = d->workspace; gp.ldws = (int64_t)gp.tiles_n * BN;
      if (gp.splits == 1) { gp.ws = nullptr; gp.ldws = 0; }
    }
  }
  if (fp8 && gp.mx_a) return rm ? launch_gemm<MG_A_DENSE, MG_W_ROWMAJOR, true, true>(gp, s) : launch_gemm<MG_A_DENSE, MG_W_FRAGTILED, true, true>(gp, s);
  if (fp8) return rm ? launch_gemm<MG_A_DENSE, MG_W_ROWMAJOR, true>(gp, s) : launch_gemm<MG_A_DENSE, MG_W_FRAGTILED, true>(gp, s);
  if (d->a_mode == MG_A_DENSE) return rm ? launch_gemm<MG_A_DENSE, MG_W_ROWMAJOR>(gp, s) : launch_gemm<MG_A_DENSE, MG_W_FRAGTILED>(gp, s);
  return rm ? launch_gemm<MG_A_CONV3X3, MG_W_ROWMAJOR>(gp, s) : launch_gemm<MG_A_CONV3X3, MG_W_FRAGTILED>(gp, s);
}
}  // namespace

extern "C" int64_t mg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nkt = (K + BK - 1) / BK;
  const int tiles = tiles_m * tiles_n;
  if (tiles >= 192 && !(tiles < 512 && nkt >= 128)) return 0;   // same policy as gemm_dispatch: enough tiles, no split
  const int want = tiles >= 192 ? 2 : std::max(1, std::min(std::min(16, nkt / 4), (512 + tiles - 1) / tiles));
  const int64_t bytes128 = want > 1 ? (int64_t)want * M * tiles_n * BN * 4 : 0;
  // the split form of the 256x256 kernel (gemm_dispatch: few 256x256 tiles over a long contraction) takes more slabs of wider rows;
  // with less than this the dispatch falls back to the 128x128 split above
  int64_t bytes256 = 0;
  const int64_t wgs256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
  const int nkt256 = K >> 6;
  if (K % 128 == 0 && !want256_noforce(0, wgs256, M, N) && wgs256 >= 32 && wgs256 < 192 && nkt256 >= 256 && M > 256 && N >= 512) {
    int w = 1;
    while (wgs256 * w * 2 <= 320 && w < 8) w *= 2;
    if (w > 1 && nkt256 % (2 * w) == 0 && w * wgs256 >= 192) bytes256 = (int64_t)w * M * (((N + 255) / 256) * 256) * 4;
  }
  return std::max(bytes128, bytes256);
}

extern "C" int mg_gemm_bf16(const mg_gemm_desc* d, void* stream) {
  return gemm_dispatch(d, false, nullptr, (hipStream_t)stream, "mg_gemm_bf16");
}

// MX (OCP microscaling) form of mg_gemm_fp8: operands and E8M0 block scales from mg_quantize_mx_fp8; no row / column scales.
extern "C" int mg_gemm_mx_fp8(const mg_gemm_desc* d, const uint8_t* a_scales, const uint8_t* w_scales, void* stream) {
  if (!d) MG_FAIL(MG_ERR_SHAPE, "mg_gemm_mx_fp8: null descriptor");
  const int64_t chunks = ((int64_t)d->K + 127) / 128;
  if (!a_scales || !w_scales || ((uintptr_t)a_scales & 3) || ((uintptr_t)w_scales & 3)) MG_FAIL(MG_ERR_SHAPE, "mg_gemm_mx_fp8: block scales missing or not 4-byte aligned");
  if (d->lda < chunks * 128 || d->ldw < chunks * 128) MG_FAIL(MG_ERR_SHAPE, "mg_gemm_mx_fp8: operand rows must be padded to whole 128-element chunks (mg_quantize_mx_fp8)");
  // Both tile kernels take block scales: the 128x128 one loads them into registers a tile ahead, the 256x256 one (whose K loop
  // has no register to spare) stages them through LDS with the tile they belong to.  tile_hint 0 = the usual automatic choice.
  if (d->tile_hint != 0 && d->tile_hint != 128 && d->tile_hint != 256) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_mx_fp8: tile_hint must be 0, 128 or 256");
  const MxScales mx{(const uint32_t*)a_scales, (const uint32_t*)w_scales};
  mg_gemm_desc dd = *d;
  return gemm_dispatch(&dd, true, nullptr, (hipStream_t)stream, "mg_gemm_mx_fp8", &mx);
}

extern "C" int mg_gemm_fp8(const mg_gemm_desc* d, const float* row_scale, void* stream) {
  if (!row_scale) MG_FAIL(MG_ERR_SHAPE, "mg_gemm_fp8: null row_scale");
  return gemm_dispatch(d, true, row_scale, (hipStream_t)stream, "mg_gemm_fp8");
}

namespace {
// validate a skinny descriptor and turn it into kernel parameters
int fill_skinny(const mg_skinny_desc* d, SkinnyParams& sp, const char* who) {
  if (!d) MG_FAIL(MG_ERR_SHAPE, "%s: null descriptor", who);
  if (d->M <= 0 || d->M > 16) MG_FAIL(MG_ERR_SHAPE, "%s: M=%d must be in [1,16]", who, d->M);
  if (d->N <= 0 || d->Kp <= 0 || (d->Kp & 63)) MG_FAIL(MG_ERR_SHAPE, "%s: need N>0 and Kp%%64==0 (N=%d Kp=%d)", who, d->N, d->Kp);
  if (!d->X || !d->W) MG_FAIL(MG_ERR_SHAPE, "%s: null X/W", who);
  if (!MG_ALIGNED16(d->X) || !MG_ALIGNED16(d->W) || (d->ldx & 7) || d->ldx < d->Kp) MG_FAIL(MG_ERR_ALIGN, "%s: X/W 16-byte aligned, ldx%%8==0, ldx>=Kp required", who);
  if (int rc = check_epilogue(d->ep, who)) return rc;
  sp.X = d->X; sp.ldx = d->ldx; sp.W = d->W; sp.M = d->M; sp.N = d->N;
  sp.ntiles = (d->N + 15) / 16; sp.ksteps = d->Kp / 32; sp.ep = d->ep;
  sp.ln_colsum = d->ln_colsum; sp.ln_inv_d = d->ln_inv_d; sp.ln_eps = d->ln_eps;
  sp.split_n = d->split_n; sp.ep_b = d->ep_b;
  sp.w_scale = d->w_scale;
#ifdef MG_GEMM_ABLATIONS
  static const int dbg = [] { const char* e = getenv("MAGMA_SKINNY_DBG"); return e ? atoi(e) : 0; }();
  sp.dbg = dbg;
#endif
  if (d->w_scale && (!MG_ALIGNED16(d->w_scale) || (d->Kp & 1023))) MG_FAIL(MG_ERR_SHAPE, "%s: fp8 weights need a 16-byte aligned w_scale and Kp %% 1024 == 0", who);
  if (d->split_n != 0) {
    if (d->split_n < 0 || d->split_n >= d->N || (d->split_n & 15)) MG_FAIL(MG_ERR_SHAPE, "%s: split_n must be a multiple of 16 inside (0, N)", who);
    if (int rc = check_epilogue(d->ep_b, who)) return rc;
  }
  if (d->ln_colsum && (!MG_ALIGNED16(d->ln_colsum) || d->ln_inv_d <= 0.f)) MG_FAIL(MG_ERR_ALIGN, "%s: bad LayerNorm-fold arguments", who);
  return MG_OK;
}

// decode attention workgroups and the workgroups of one weight-streaming GEMV in ONE launch: the
// attention part (B*H workgroups, latency-bound, a few hundred KB of KV) runs underneath the GEMV's
// HBM stream instead of leaving most of the chip idle for ~10 us per layer.
template <int KC, bool W8 = false, bool PIPE = false>
__global__ __launch_bounds__(256) void decode_attn_gemv_kernel(const AttnDecodeParams ap, int n_attn, const SkinnyParams sp) {
  constexpr int LDS = ATTN_DEC_LDS > skinny_lds_bytes<4, 1>() ? ATTN_DEC_LDS : skinny_lds_bytes<4, 1>();
  __shared__ __attribute__((aligned(16))) char lds[LDS];
  if ((int)blockIdx.x < n_attn) attn_decode_body<true>(ap, blockIdx.x, lds);
  else skinny_body<4, KC, 1, W8, false, NoWait, PIPE>(sp, blockIdx.x - n_attn, lds);
}

constexpr int MG_DECODE_PIPE_DEFAULT = 0;
}  // namespace

extern "C" int mg_gemm_skinny_bf16(const mg_skinny_desc* d, void* stream) {
  SkinnyParams sp;
  if (int rc = fill_skinny(d, sp, "mg_gemm_skinny_bf16")) return rc;
  hipStream_t s = (hipStream_t)stream;
  // Variant = (waves per workgroup, k-steps per load burst, n-tiles per
  // workgroup).  nt_hint == 0 -> tuned default for the shape; otherwise
  // nt_hint = nt | waves<<4 | kc<<8 (bench/tuning sweeps use this).
  int nt = d->nt_hint & 15, waves = (d->nt_hint >> 4) & 15, kc = (d->nt_hint >> 8) & 255;
  const bool pipe = (d->nt_hint >> 16) & 1;        // double-buffered weight bursts (skinny_body<..., PIPE>)
#ifdef MG_GEMM_ABLATIONS
  // round-5 experiment, ablation library only (4-11 % slower than the register-direct stream in every form:
  // profiles/r05_gemv_lds_dma_experiment.txt): bit 18 = non-temporal DMA, 19..21 = form, 22..30 = workgroups
  if ((d->nt_hint >> 17) & 1) return skinny_dma_launch(sp, d->nt_hint >> 18, s);
#else
  if ((d->nt_hint >> 17) & 1) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: the LDS-DMA GEMV exists only in the ablation library (make ABL=1)");
#endif
  if (pipe) {
    if (sp.w_scale || nt != 1 || sp.ksteps % (waves * kc) != 0) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: pipelined variants are bf16, one n-tile");
#define MG_SKP(W_, K_) if (waves == W_ && kc == K_) return launch_skinny<W_, K_, 1, false, true>(sp, s)
    MG_SKP(4, 16); MG_SKP(4, 8); MG_SKP(8, 8); MG_SKP(8, 4); MG_SKP(4, 4);
#undef MG_SKP
    MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: pipelined variant (waves=%d,kc=%d) not instantiated", waves, kc);
  }
  if (d->nt_hint == 0) {
    // measured on MI355X (tools/kbench.py, profiles/r01_kbench_*.jsonl): many
    // waves with short load bursts beat few waves with deep ones.
    if (sp.ksteps % (4 * 16) == 0 && sp.ksteps >= 512) { waves = 4; kc = 16; nt = 1; }   // K=16384 (fc_out): 5.57 TB/s
    else if (sp.ksteps % (8 * 4) == 0) { waves = 8; kc = 4; nt = 1; }
    else if (sp.ksteps % 4 == 0) { waves = 4; kc = 1; nt = 1; }
    else { waves = 1; kc = 1; nt = 1; }
  }
  if (waves <= 0 || kc <= 0 || nt <= 0 || sp.ksteps % (waves * kc) != 0)
    MG_FAIL(MG_ERR_SHAPE, "mg_gemm_skinny_bf16: variant (waves=%d,kc=%d,nt=%d) does not divide ksteps=%d", waves, kc, nt, sp.ksteps);
  if (sp.w_scale) {   // fp8 weights: a 16-byte load covers TWO k-steps, so the bursts are twice as deep as for bf16 to keep
                      // the same bytes in flight (Kp % 1024 == 0 guarantees that every variant below divides)
    if (d->nt_hint != 0) {
      if (waves == 8 && kc == 4) return launch_skinny<8, 4, 1, true>(sp, s);
      if (waves == 8 && kc == 8 && sp.ksteps % 64 == 0) return launch_skinny<8, 8, 1, true>(sp, s);
      if (waves == 4 && kc == 16 && sp.ksteps % 64 == 0) return launch_skinny<4, 16, 1, true>(sp, s);
      if (waves == 4 && kc == 8) return launch_skinny<4, 8, 1, true>(sp, s);
      MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: fp8-weight variant (waves=%d,kc=%d) not instantiated", waves, kc);
    }
    if (sp.ksteps >= 512) return launch_skinny<4, 16, 1, true>(sp, s);
    if (sp.ksteps % 64 == 0) return launch_skinny<8, 8, 1, true>(sp, s);
    return launch_skinny<4, 8, 1, true>(sp, s);
  }
#define MG_SK(W_, K_, N_) if (waves == W_ && kc == K_ && nt == N_) return launch_skinny<W_, K_, N_>(sp, s)
  MG_SK(8, 16, 1); MG_SK(8, 16, 2); MG_SK(4, 16, 1); MG_SK(4, 16, 2);
  MG_SK(8, 8, 1);  MG_SK(8, 8, 2);  MG_SK(8, 8, 4);  MG_SK(4, 8, 2); MG_SK(4, 8, 4);
  MG_SK(8, 4, 1);  MG_SK(8, 4, 2);  MG_SK(8, 4, 4);
  MG_SK(16, 8, 1); MG_SK(16, 4, 1); MG_SK(16, 2, 1); MG_SK(16, 4, 2);
  MG_SK(4, 1, 1);  MG_SK(1, 1, 1);
#undef MG_SK
  MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny_bf16: variant (waves=%d,kc=%d,nt=%d) not instantiated", waves, kc, nt);
}

// Two independent decode GEMVs in one launch (out_proj || adapter-down).  Both K must be
// multiples of 1024 (8 waves x 4 k-steps) or of 128 (4 waves x 1).
extern "C" int mg_gemm_skinny2_bf16(const mg_skinny_desc* a, const mg_skinny_desc* b, void* stream) {
  SkinnyParams pa, pb;
  if (int rc = fill_skinny(a, pa, "mg_gemm_skinny2_bf16(a)")) return rc;
  if (int rc = fill_skinny(b, pb, "mg_gemm_skinny2_bf16(b)")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int grid = pa.ntiles + pb.ntiles;
  // (burst shapes other than 8 waves x 4 k-steps -- one burst of 16, two of 8 issued up front, double-buffered 4 x 16 and
  //  8 x 4 -- were measured inside the decode graph in rounds 2 and 3: all slower, 2.55-2.60 ms per token against 2.54)
  if ((pa.w_scale != nullptr) != (pb.w_scale != nullptr)) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny2_bf16: both problems must use the same weight type");
  if (pa.w_scale && pa.ksteps % 64 == 0 && pb.ksteps % 64 == 0) {
    hipLaunchKernelGGL((skinny2_kernel<8, 8, 1, true>), dim3(grid), dim3(512), 0, s, pa, pb, pa.ntiles);
  } else if (pa.w_scale) {
    hipLaunchKernelGGL((skinny2_kernel<8, 4, 1, true>), dim3(grid), dim3(512), 0, s, pa, pb, pa.ntiles);
  } else if (pa.ksteps % 32 == 0 && pb.ksteps % 32 == 0) {
    hipLaunchKernelGGL((skinny2_kernel<8, 4, 1>), dim3(grid), dim3(512), 0, s, pa, pb, pa.ntiles);
  } else if (pa.ksteps % 4 == 0 && pb.ksteps % 4 == 0) {
    hipLaunchKernelGGL((skinny2_kernel<4, 1, 1>), dim3(grid), dim3(256), 0, s, pa, pb, pa.ntiles);
  } else {
    MG_FAIL(MG_ERR_UNSUPPORTED, "mg_gemm_skinny2_bf16: K of both problems must be a multiple of 128");
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

// Decode attention (rotary + KV append + attention, see mg_attn_decode_fused_bf16) co-launched with
// one weight-streaming GEMV that does not depend on it (fc_out of the parallel GPT-J block).
extern "C" int mg_decode_attn_gemv_bf16(const mg_bf16* qkv, mg_bf16* kcache, mg_bf16* vcache, mg_bf16* attn_out, int64_t ld_attn_out,
                                        int32_t B, int32_t H, int32_t Smax, const int32_t* d_pos, int32_t rot_dim,
                                        const float* sin_t, const float* cos_t, const mg_skinny_desc* gemv,
                                        void* stream) {
  if (B <= 0 || H <= 0 || Smax <= 0 || Smax > DEC_MAX_CTX) MG_FAIL(MG_ERR_SHAPE, "mg_decode_attn_gemv_bf16: need 0 < Smax <= %d", DEC_MAX_CTX);
  if (rot_dim < 0 || rot_dim > 256 || (rot_dim & 7)) MG_FAIL(MG_ERR_SHAPE, "mg_decode_attn_gemv_bf16: rot_dim must be a multiple of 8 in [0,256]");
  if (!qkv || !kcache || !vcache || !attn_out || !d_pos || (rot_dim && (!sin_t || !cos_t))) MG_FAIL(MG_ERR_SHAPE, "mg_decode_attn_gemv_bf16: null pointer");
  if (!MG_ALIGNED16(qkv) || !MG_ALIGNED16(kcache) || !MG_ALIGNED16(vcache) || !MG_ALIGNED16(attn_out)) MG_FAIL(MG_ERR_ALIGN, "mg_decode_attn_gemv_bf16: pointers must be 16-byte aligned");
  SkinnyParams sp;
  if (int rc = fill_skinny(gemv, sp, "mg_decode_attn_gemv_bf16(gemv)")) return rc;
  if (ld_attn_out != 0 && (ld_attn_out < (int64_t)H * 256 || (ld_attn_out & 7))) MG_FAIL(MG_ERR_SHAPE, "mg_decode_attn_gemv_bf16: ld_attn_out must be 0 or a multiple of 8 >= H*256");
  AttnDecodeParams ap{qkv, kcache, vcache, attn_out, H, Smax, d_pos, rot_dim, sin_t, cos_t, ld_attn_out};
  hipStream_t s = (hipStream_t)stream;
  const int n_attn = B * H, grid = n_attn + sp.ntiles;
  if (sp.w_scale && sp.ksteps % 64 != 0) MG_FAIL(MG_ERR_UNSUPPORTED, "mg_decode_attn_gemv_bf16: fp8 weights need K %% 2048 == 0 here");
  // MAGMA_DECODE_PIPE: 0 = burst-and-drain, 16 / 8 = double-buffered bursts of that many k-steps (bf16 weights)
  const int pipe = [] { const char* e = getenv("MAGMA_DECODE_PIPE"); return e ? atoi(e) : MG_DECODE_PIPE_DEFAULT; }();   // read per call (graph capture)
  if (sp.w_scale) hipLaunchKernelGGL((decode_attn_gemv_kernel<16, true>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else if (pipe == 16 && sp.ksteps % 64 == 0) hipLaunchKernelGGL((decode_attn_gemv_kernel<16, false, true>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else if (pipe == 8 && sp.ksteps % 32 == 0) hipLaunchKernelGGL((decode_attn_gemv_kernel<8, false, true>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else if (sp.ksteps % 64 == 0) hipLaunchKernelGGL((decode_attn_gemv_kernel<16>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else if (sp.ksteps % 16 == 0) hipLaunchKernelGGL((decode_attn_gemv_kernel<4>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else if (sp.ksteps % 4 == 0) hipLaunchKernelGGL((decode_attn_gemv_kernel<1>), dim3(grid), dim3(256), 0, s, ap, n_attn, sp);
  else MG_FAIL(MG_ERR_UNSUPPORTED, "mg_decode_attn_gemv_bf16: K of the GEMV must be a multiple of 128");
  MG_CHECK_LAUNCH();
  return MG_OK;
}
