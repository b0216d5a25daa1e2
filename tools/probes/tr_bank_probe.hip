// LDS bank behaviour of ds_read_b64_tr_b16 / ds_read_b128 on candidate attention row-tile images (gfx950, round 6).
//
// Question: can ONE [32 rows][256 d] bf16 image (512-byte rows, 16-byte chunks permuted inside a row by an XOR of the row index)
// serve BOTH fragment reads of the attention backward without bank conflicts --
//   (R) the d-contraction operand (S = Q K^T, dP = dO V^T): ds_read_b128, lane (l31, hi) reads chunk 2 ks + hi of row perm32(l31);
//   (T) the s-contraction operand (dV^T += dO^T P, dK^T += Q^T dS): ds_read_b64_tr_b16, two per 32x32x16 A fragment, lane
//       (i = lane & 15, g16 = (lane >> 4) & 1, hi = lane >> 5) reads 8 bytes of row ks 16 + hi 8 + 4 j + (i >> 2) at column
//       db 32 + g16 16 + (i & 3) 4 and receives column db 32 + (lane & 31), rows .. + 0..3
// -- so that the transposed HBM images (Q^T, dO^T, K^T, V^T) and their LDS-DMA pieces can go.
//
// The host builds, per pattern, the byte address of every lane for 16 reads; the kernel checks WHAT each read returned against the
// element ids stored in the image, then times 256 x 16 reads with s_memtime (1 wave and 4 waves per CU).
//   build: hipcc --offload-arch=gfx950 -O3 tr_bank_probe.hip -o tr_bank_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int NREAD = 16, ITERS = 2048;

template <int KIND>   // 0 = ds_read_b64_tr_b16, 1 = ds_read_b128, 2 = ds_read_b64
__global__ __launch_bounds__(256) void probe(const unsigned short* image, const unsigned* addr, unsigned* out, unsigned long long* cycles) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];   // 64 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768; i += blockDim.x) lds[i] = image[i];
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
  unsigned a[NREAD];
#pragma unroll
  for (int r = 0; r < NREAD; ++r) a[r] = base + addr[r * 64 + lane];
  // ---- what did each read return? (first wave only)
  if (tid < 64) {
#pragma unroll
    for (int r = 0; r < NREAD; ++r) {
      if constexpr (KIND == 1) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[r]) : "memory");
        for (int j = 0; j < 4; ++j) out[(r * 64 + lane) * 4 + j] = v[j];
      } else {
        u32x2 v;
        if constexpr (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[r]) : "memory");
        else asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[r]) : "memory");
        out[(r * 64 + lane) * 4] = v[0]; out[(r * 64 + lane) * 4 + 1] = v[1];
      }
    }
  }
  __syncthreads();
  // ---- timing: ITERS x NREAD reads, one wait per NREAD
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (KIND == 1) {
      u32x4 v[NREAD];
#pragma unroll
      for (int r = 0; r < NREAD; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"(a[r]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < NREAD; ++r) asm volatile("" : "+v"(v[r]));
#pragma unroll
      for (int r = 0; r < NREAD; ++r) acc ^= v[r][0];
    } else {
      u32x2 v[NREAD];
#pragma unroll
      for (int r = 0; r < NREAD; ++r) {
        if constexpr (KIND == 0) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[r]) : "v"(a[r]) : "memory");
        else asm volatile("ds_read_b64 %0, %1" : "=v"(v[r]) : "v"(a[r]) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < NREAD; ++r) asm volatile("" : "+v"(v[r]));
#pragma unroll
      for (int r = 0; r < NREAD; ++r) acc ^= v[r][0];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
  if (acc == 0x12345678u) out[0] = acc;
}

// ---- candidate images -------------------------------------------------------------------------------------------
static int perm32(int i) { return (i & 19) | ((i & 4) << 1) | ((i & 8) >> 1); }
static int swz_old(int r) { return (r & 3) | ((r >> 3) << 2); }          // attn_tile_device.h row_swz
static int swz_new(int r) { return ((r & 3) << 2) | ((r >> 3) & 3); }    // candidate: rows r..r+3 land in four different 64-byte bank spans
static int swz_new2(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }   // ... distinct on contiguous 16-lane groups as well
static int swz_none(int) { return 0; }
typedef int (*swz_fn)(int);

// element id of (row, col): row 5 bits | col 8 bits
static unsigned short eid(int row, int col) { return (unsigned short)((row << 8) | col); }
static void build_image(std::vector<unsigned short>& img, swz_fn swz) {
  img.assign(32768, 0xffff);
  for (int row = 0; row < 32; ++row)
    for (int c = 0; c < 32; ++c)
      for (int e = 0; e < 8; ++e) img[(row * 512 + ((c ^ swz(row)) << 4)) / 2 + e] = eid(row, c * 8 + e);
}
// (T) read r = (db = r >> 2, ks = (r >> 1) & 1, j = r & 1) of d-blocks 0..3
static void addr_T(std::vector<unsigned>& a, swz_fn swz, int db0) {
  a.resize(NREAD * 64);
  for (int r = 0; r < NREAD; ++r)
    for (int l = 0; l < 64; ++l) {
      const int db = db0 + (r >> 2), ks = (r >> 1) & 1, j = r & 1;
      const int i = l & 15, g16 = (l >> 4) & 1, hi = l >> 5;
      const int row = ks * 16 + hi * 8 + 4 * j + (i >> 2);
      const int col = db * 32 + g16 * 16 + (i & 3) * 4;
      a[r * 64 + l] = (unsigned)(row * 512 + (((col >> 3) ^ swz(row)) << 4) + (col & 7) * 2);
    }
}
// (R) read r = ks (0..15)
static void addr_R(std::vector<unsigned>& a, swz_fn swz) {
  a.resize(NREAD * 64);
  for (int r = 0; r < NREAD; ++r)
    for (int l = 0; l < 64; ++l) {
      const int row = perm32(l & 31), c = 2 * r + (l >> 5);
      a[r * 64 + l] = (unsigned)(row * 512 + ((c ^ swz(row)) << 4));
    }
}
static void addr_linear(std::vector<unsigned>& a, int bytes) {
  a.resize(NREAD * 64);
  for (int r = 0; r < NREAD; ++r)
    for (int l = 0; l < 64; ++l) a[r * 64 + l] = (unsigned)(r * 64 * bytes + l * bytes);
}

template <int KIND>
static void run(const char* name, const std::vector<unsigned short>& img, const std::vector<unsigned>& addr, int check /*0 none, 1 T, 2 R*/,
                int db0 = 0) {
  unsigned short* d_img; unsigned* d_addr; unsigned* d_out; unsigned long long* d_cyc;
  hipMalloc(&d_img, 65536); hipMalloc(&d_addr, NREAD * 64 * 4); hipMalloc(&d_out, NREAD * 64 * 16); hipMalloc(&d_cyc, 8 * 4096);
  hipMemcpy(d_img, img.data(), 65536, hipMemcpyHostToDevice);
  hipMemcpy(d_addr, addr.data(), NREAD * 64 * 4, hipMemcpyHostToDevice);
  double cyc[2] = {0, 0};
  std::vector<unsigned> out(NREAD * 64 * 4);
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int threads = cfg == 0 ? 64 : 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(cfg == 0 ? 1 : 1024), dim3(threads), 0, 0, d_img, d_addr, d_out, d_cyc);
    hipDeviceSynchronize();
    unsigned long long c[4];
    hipMemcpy(c, d_cyc, 32, hipMemcpyDeviceToHost);
    unsigned long long m = 0;
    for (int w = 0; w < threads / 64; ++w) m = c[w] > m ? c[w] : m;
    cyc[cfg] = (double)m / (ITERS * NREAD);
  }
  hipMemcpy(out.data(), d_out, NREAD * 64 * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  if (check == 1) {
    for (int r = 0; r < NREAD; ++r)
      for (int l = 0; l < 64; ++l) {
        const int db = db0 + (r >> 2), ks = (r >> 1) & 1, j = r & 1, hi = l >> 5;
        for (int e = 0; e < 4; ++e) {
          const unsigned got = (out[(r * 64 + l) * 4 + (e >> 1)] >> ((e & 1) * 16)) & 0xffff;
          const unsigned want = eid(ks * 16 + hi * 8 + 4 * j + e, db * 32 + (l & 31));
          bad += got != want;
        }
      }
  } else if (check == 2) {
    for (int r = 0; r < NREAD; ++r)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const unsigned got = (out[(r * 64 + l) * 4 + (e >> 1)] >> ((e & 1) * 16)) & 0xffff;
          bad += got != eid(perm32(l & 31), (2 * r + (l >> 5)) * 8 + e);
        }
  }
  // s_memtime ticks at 100 MHz on gfx950: report ticks x (shader clock / 100 MHz) is left to the reader; the RATIO between
  // patterns is what matters
  printf("{\"pattern\": \"%s\", \"ticks_per_read_1wave\": %.4f, \"ticks_per_read_4waves\": %.4f, \"wrong_elements\": %d}\n", name, cyc[0], cyc[1],
         check ? bad : -1);
  hipFree(d_img); hipFree(d_addr); hipFree(d_out); hipFree(d_cyc);
}

int main() {
  {   // bring the clocks up before the first measurement
    std::vector<unsigned short> w; std::vector<unsigned> wa;
    build_image(w, swz_none); addr_linear(wa, 16);
    for (int i = 0; i < 3; ++i) run<1>("warm-up (ignore)", w, wa, 0);
  }
  std::vector<unsigned short> img;
  std::vector<unsigned> a;
  build_image(img, swz_none);
  addr_linear(a, 8);  run<0>("tr_b64 linear 8 B/lane (conflict-free reference)", img, a, 0);
  addr_linear(a, 8);  run<2>("b64 linear 8 B/lane", img, a, 0);
  addr_linear(a, 16); run<1>("b128 linear 16 B/lane", img, a, 0);
  addr_T(a, swz_none, 0); run<0>("tr_b64 T-fragment, no swizzle (4-way by the bank model)", img, a, 1);
  build_image(img, swz_old);
  addr_T(a, swz_old, 0); run<0>("tr_b64 T-fragment, round-5 row_swz", img, a, 1);
  addr_R(a, swz_old);    run<1>("b128 R-fragment, round-5 row_swz", img, a, 2);
  build_image(img, swz_new);
  addr_T(a, swz_new, 0); run<0>("tr_b64 T-fragment, candidate swz (r&3)<<2|(r>>3) (d-blocks 0-3)", img, a, 1, 0);
  addr_T(a, swz_new, 4); run<0>("tr_b64 T-fragment, candidate swz (r&3)<<2|(r>>3) (d-blocks 4-7)", img, a, 1, 4);
  addr_R(a, swz_new);    run<1>("b128 R-fragment, candidate swz (r&3)<<2|(r>>3)", img, a, 2);
  build_image(img, swz_new2);
  addr_T(a, swz_new2, 0); run<0>("tr_b64 T-fragment, candidate swz (r&3)<<2|((r>>2)&3) (d-blocks 0-3)", img, a, 1, 0);
  addr_T(a, swz_new2, 4); run<0>("tr_b64 T-fragment, candidate swz (r&3)<<2|((r>>2)&3) (d-blocks 4-7)", img, a, 1, 4);
  addr_R(a, swz_new2);    run<1>("b128 R-fragment, candidate swz (r&3)<<2|((r>>2)&3)", img, a, 2);
  return 0;
}
