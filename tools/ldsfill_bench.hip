// L2 -> LDS fill rate per CU: LDS-DMA (global_load_lds, 16 B / lane) against register staging
// (global_load_dwordx4 + ds_write_b128) and a half / half mix.  256 workgroups x 512 threads, one
// per CU; every workgroup streams the GEMM's fill pattern (8 rows x 128 B per wave-instruction,
// rows 4096 B apart) from a 16 MiB operand that stays in L2 / MALL, into a 128 KiB LDS ring.
// build: hipcc --offload-arch=gfx950 -O3 tools/ldsfill_bench.hip -o tools/ldsfill_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>   // 0 = LDS-DMA, 1 = registers, 2 = waves 0-3 DMA + waves 4-7 registers
__global__ __launch_bounds__(512) void fill(const unsigned char* __restrict__ src, int ld, int nblocks, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane >> 3, slot = lane & 7;
  // this workgroup's 512 rows (two operands of 256): rows [512 wg % 3584 ...)
  const unsigned char* base = src + (size_t)((blockIdx.x * 512) % 3584) * ld;
  unsigned voff[8];
  for (int r = 0; r < 8; ++r) voff[r] = (unsigned)(64 * r + 8 * wave + row) * (unsigned)ld + (slot << 4);
  unsigned acc = 0;
  for (int kb = 0; kb < nblocks; ++kb) {
    const unsigned char* g = base + (kb % 32) * 128;
    unsigned char* d = lds + (kb & 1) * 65536 + 8 * wave * 128;
    const bool dma = MODE == 0 || (MODE == 2 && wave < 4);
    if (dma) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + voff[r]),
                                         (__attribute__((address_space(3))) void*)(d + 64 * r * 128), 16, 0, 0);
    } else {
      u32x4 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = *(const u32x4*)(g + voff[r]);
#pragma unroll
      for (int r = 0; r < 8; ++r) *(u32x4*)(d + 64 * r * 128 + lane * 16) = v[r];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *(unsigned*)(lds + (kb & 1) * 65536 + threadIdx.x * 4);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int ld = 4096, rows = 4096;
  unsigned char* src; unsigned* sink;
  CK(hipMalloc(&src, (size_t)rows * ld)); CK(hipMemset(src, 3, (size_t)rows * ld)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nblocks = 2048;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      auto go = [&]() {
        if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(256), dim3(512), 0, 0, src, ld, nblocks, sink);
        if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(256), dim3(512), 0, 0, src, ld, nblocks, sink);
        if (mode == 2) hipLaunchKernelGGL(fill<2>, dim3(256), dim3(512), 0, 0, src, ld, nblocks, sink);
      };
      go(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us_per_block = ms * 1e3 / nblocks;
      if (rep) printf("mode %d (%s): %.3f us per 64 KiB block per CU = %.1f GB/s per CU, %.1f TB/s chip\n", mode,
                      mode == 0 ? "LDS-DMA" : mode == 1 ? "registers + ds_write" : "half DMA, half registers",
                      us_per_block, 65536 / us_per_block / 1e3, 65536.0 * 256 / us_per_block / 1e6);
    }
  return 0;
}
