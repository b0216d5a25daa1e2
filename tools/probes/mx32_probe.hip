// v_mfma_scale_f32_32x32x64_f8f6f4 operand semantics probe (gfx950): which lane supplies the E8M0 scale of (row, 32-k block),
// and which of a lane's 32 operand bytes belong to which block.  A[row][*]: a lane's bytes 0-15 = 1.0, bytes 16-31 = 2.0 (e4m3),
// B = 1.0, unit scales except ONE lane's scale byte (A side or B side) = 2^1.  Baseline D = 32*1 + 32*2 = 96 per element.
//   build: hipcc --offload-arch=gfx950 -O3 mx32_probe.hip -o mx32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out, int lane_sel, int side, int opsel) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = i < 4 ? 0x38383838 : 0x40404040; b[i] = 0x38383838; }
  int sa = 0x7F7F7F7F, sb = 0x7F7F7F7F;
  if (l == lane_sel) { if (side == 0) sa = 0x80808080; else sb = 0x80808080; }
  f32x16 c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];   // D[row][col]
}
int main() {
  float* d; hipMalloc(&d, 4096); float h[1024];
  for (int side = 0; side < 2; ++side)
    for (int L = 0; L < 64; L += 1) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, L, side, 0);
      hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
      // summarise: which rows / cols deviate from 96 and by how much
      int nr = 0, nc = 0, r0 = -1, c0 = -1; float val = 0;
      for (int r = 0; r < 32; ++r) { bool any = false; for (int c = 0; c < 32; ++c) if (h[r * 32 + c] != 96.f) { any = true; val = h[r * 32 + c]; } if (any) { ++nr; r0 = r; } }
      for (int c = 0; c < 32; ++c) { bool any = false; for (int r = 0; r < 32; ++r) if (h[r * 32 + c] != 96.f) any = true; if (any) { ++nc; c0 = c; } }
      if (L < 4 || (L >= 30 && L < 36) || L >= 62) printf("side %c lane %2d: rows changed %d (last %d) cols changed %d (last %d) value %.0f\n", side ? 'B' : 'A', L, nr, r0, nc, c0, val);
    }
  return 0;
}
