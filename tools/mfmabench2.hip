// MFMA shape A/B under the power limit: v_mfma_scale_f32_16x16x128_f8f6f4 against v_mfma_scale_f32_32x32x64_f8f6f4
// (fp8 operands, unit scales; same FLOP per cycle on paper), random operands, 128 accumulator registers per lane,
// one or two waves per SIMD, ~10 ms per launch, in-kernel clock from s_memtime / s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 tools/mfmabench2.hip -o tools/mfmabench2 && tools/mfmabench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const int* __restrict__ src, float* __restrict__ dst, int iters,
                                                 unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  i32x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(i * 8 + j) * 64 + lane];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(32 + i * 8 + j) * 64 + lane];
  float out = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 16) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[(i >> 2) & 1], acc[i], 0, 0, 0,
                                                                   0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 32; ++i) s += acc[i];
    out = s[0] + s[1] + s[2] + s[3];
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // same FLOP per trip as the other shape: 32 x (16x16x128) = 16 x (32x32x64)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 3], b[(i >> 2) & 1], acc[i & 7], 0, 0, 0,
                                                                      0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    f32x16 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    for (int j = 0; j < 16; ++j) out += s[j];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  dst[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = out;
  if (threadIdx.x == 0 && blockIdx.x < 256) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  int* src; float* dst; unsigned long long* clk;
  std::vector<int> h(48 * 64);
  srand(1);
  for (auto& v : h) {
    unsigned w = 0;
    for (int k = 0; k < 4; ++k) w |= (unsigned)((rand() & 0x80) | (rand() % 0x48)) << (8 * k);
    v = (int)w;
  }
  CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&dst, 4096 * 512 * 4)); CK(hipMalloc(&clk, 512 * 8));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int waves : {4, 8}) for (int shape : {16, 32}) {
      auto go = [&](int n) {
        if (shape == 16) mfma_loop<16><<<256, waves * 64>>>(src, dst, n, clk);
        else mfma_loop<32><<<256, waves * 64>>>(src, dst, n, clk);
      };
      go(iters); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); go(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long hc[512];
      CK(hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost));
      double cyc = 0, rt = 0;
      for (int i = 0; i < 256; ++i) { cyc += (double)hc[2 * i]; rt += (double)hc[2 * i + 1]; }
      const double flop = 256.0 * waves * iters * 32 * 2.0 * 16 * 16 * 128;
      printf("random operands, %d waves per CU, %s: %8.3f ms  %7.1f TFLOP/s  in-kernel %.2f GHz, %.1f cycles per %s\n",
             waves, shape == 16 ? "16x16x128" : "32x32x64 ", ms, flop / ms / 1e9, cyc / rt * 0.1,
             cyc / 256 / iters / (shape == 16 ? 32 : 16) / (waves / 4), shape == 16 ? "16x16x128" : "32x32x64");
    }
  return 0;
}
