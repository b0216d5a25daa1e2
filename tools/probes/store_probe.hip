// Store-path probe (MI355X): how fast can ONE workgroup of 512 threads write a 256 x 256 bf16 tile (128 KiB) the way the GEMM
// epilogue does -- 16 x global_store_dwordx4 per wave, each covering 2 rows x 512 B -- alone and with 255 others?
// In-kernel 100-MHz stamps; build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT, bool CONTIG>
__global__ __launch_bounds__(512) void probe(unsigned short* out, long ldc, int tiles_n, unsigned long long* stamps, int reps) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int r = it * 16 + wave * 2 + (lane >> 5);           // tile row
      u32x4 v = {(unsigned)tid, (unsigned)it, (unsigned)rep, 7u};
      unsigned short* p = CONTIG ? out + ((long)blockIdx.x * 65536 + (long)r * 256 + (lane & 31) * 8)
                                 : out + ((long)(tm * 256 + r) * ldc + tn * 256 + (lane & 31) * 8);
      if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
}

int main() {
  const long M = 32768, N = 4096;
  unsigned short* out; unsigned long long* st;
  hipMalloc(&out, M * N * 2); hipMalloc(&st, 4096 * 4 * 8);
  std::vector<unsigned long long> h(4096 * 4);
  auto run = [&](const char* name, auto kern, int grid, int reps) {
    for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, N, 16, st, reps); hipDeviceSynchronize(); }
    hipMemcpy(h.data(), st, grid * 32, hipMemcpyDeviceToHost);
    double issue = 0, drain = 0; unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < grid; ++i) { issue += (h[i * 4 + 1] - h[i * 4]) / 100.0; drain += (h[i * 4 + 2] - h[i * 4 + 1]) / 100.0; lo = std::min(lo, h[i * 4]); hi = std::max(hi, h[i * 4 + 2]); }
    printf("{\"probe\": \"%s\", \"workgroups\": %d, \"tiles_per_wg\": %d, \"issue_us_per_tile\": %.2f, \"drain_us\": %.2f, \"span_us\": %.1f, \"GBps_aggregate\": %.0f}\n",
           name, grid, reps, issue / grid / reps, drain / grid, (hi - lo) / 100.0, grid * (double)reps * 131072 / ((hi - lo) / 100.0) / 1e3);
  };
  for (int grid : {1, 8, 32, 64, 128, 256, 2048}) {
    run("rows_512B_ldc8K_nt", probe<true, false>, grid, 1);
    run("rows_512B_ldc8K", probe<false, false>, grid, 1);
    run("contig_nt", probe<true, true>, grid, 1);
  }
  run("rows_512B_ldc8K_nt_4tiles", probe<true, false>, 256, 4);
  return 0;
}
