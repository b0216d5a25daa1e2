// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds u16 element i at byte 2 i; every lane reads 8 bytes at its own
// address; prints which LDS elements each lane received.  Pattern A: lane l reads bytes [8 l, 8 l + 8).  Pattern B: the
// intended use on a row-major [m][n] tile with 128-byte rows (64 bf16): lane l (g = l >> 4, j = l & 15) reads row
// 4 g + (j >> 2), columns 4 (j & 3) .. +3.   build: hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__global__ void probe(unsigned short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr;
  if (pattern == 0) addr = 8u * l;
  else addr = (unsigned)((4 * (l >> 4) + ((l & 15) >> 2)) * 128 + (l & 3) * 8);
  addr += (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}

int main() {
  unsigned short* d; hipMalloc(&d, 512); unsigned short h[256];
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("pattern %c\n", 'A' + p);
    for (int l = 0; l < 64; ++l) {
      if (p == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      else printf("lane %2d: (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d) (r%2d,c%2d)\n", l, h[l*4]/64, h[l*4]%64, h[l*4+1]/64, h[l*4+1]%64, h[l*4+2]/64, h[l*4+2]%64, h[l*4+3]/64, h[l*4+3]%64);
    }
  }
  return 0;
}
