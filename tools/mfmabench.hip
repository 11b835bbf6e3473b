// MFMA issue-rate ceiling for v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 operands, unit scales):
//   hipcc --offload-arch=gfx950 -O3 tools/mfmabench.hip -o tools/mfmabench && tools/mfmabench
// 256*k workgroups of W waves, each wave loops over NACC independent accumulators; operands
// are random bytes (zero operands read high: the chip clocks up).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(const int* __restrict__ src, float* __restrict__ dst, int iters) {
  const int lane = threadIdx.x & 63;
  i32x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(i * 8 + j) * 64 + lane];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(32 + i * 8 + j) * 64 + lane];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[(i >> 2) & 1], acc[i], 0, 0, 0,
                                                                 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  dst[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  int* src; float* dst;
  std::vector<int> h(48 * 64);
  srand(1);
  for (auto& v : h) {
    // random fp8 e4m3 bytes with exponent <= 8 (|x| <= 3.75): no NaN, no overflow in fp32
    unsigned w = 0;
    for (int k = 0; k < 4; ++k) w |= (unsigned)((rand() & 0x80) | (rand() % 0x48)) << (8 * k);
    v = (int)w;
  }
  CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&dst, 4096 * 512 * 4));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) CK(hipMemset(src, 0, h.size() * 4));
    for (int waves : {4, 8}) for (int wgs : {256, 512}) {
      mfma_loop<32><<<wgs, waves * 64>>>(src, dst, 10);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      mfma_loop<32><<<wgs, waves * 64>>>(src, dst, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = (double)wgs * waves * iters * 32 * 2.0 * 16 * 16 * 128;
      printf("%s operands  %d WGs x %d waves: %8.3f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", wgs, waves, ms,
             flop / ms / 1e9);
    }
  }
  return 0;
}
