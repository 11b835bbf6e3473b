// How fast does the chip get through workgroups that exit at once?  (Partitioned decode launches
// of ragged batches hold many: grid = sequences x max partitions.)
// hipcc --offload-arch=gfx950 -O3 tools/empty_wg_bench.hip -o tools/empty_wg_bench
// Measured (MI355X): 16384 workgroups 4.6 us per launch = 0.3 ns each, 65536: 15 us = 0.2 ns each
// (plain or with 32 KB of LDS and 2 workgroups per CU): the ~6000 empty workgroups of a ragged
// 256 x U[2k, 16k] v2 launch cost about 1.5 us, not the 60 us that launch lacks against a uniform batch.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void empty_kernel(const int* flag, int* out) {
  if (flag[0] == 12345) out[blockIdx.x] = 1;   // never true: one scalar load, then exit
}
// same register / LDS footprint as the decode kernel would not matter for an exiting workgroup:
// dispatch needs the resources free, though, so also a fat variant
__global__ __launch_bounds__(256, 2) void fat_empty_kernel(const int* flag, int* out) {
  __shared__ int lds[8192];
  if (flag[0] == 12345) { lds[threadIdx.x] = blockIdx.x; __syncthreads(); out[blockIdx.x] = lds[255 - threadIdx.x]; }
}
int main() {
  int *flag, *out;
  hipMalloc(&flag, 4); hipMalloc(&out, 1 << 22);
  hipMemset(flag, 0, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int fat = 0; fat < 2; ++fat)
    for (int n : {512, 2048, 8192, 16384, 65536}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) {
          if (fat) hipLaunchKernelGGL(fat_empty_kernel, dim3(n), dim3(256), 0, 0, flag, out);
          else hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(256), 0, 0, flag, out);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("%s %6d workgroups: %.2f us per launch = %.1f ns per workgroup\n", fat ? "fat  " : "plain", n, ms * 1000 / 20, ms * 1e6 / 20 / n);
      }
    }
  return 0;
}
