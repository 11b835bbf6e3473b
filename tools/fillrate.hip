// L2 -> CU fill THROUGHPUT (loads kept in flight, no barrier, no drain per block) for the GEMM's
// operand pattern: what one CU can pull per clock by LDS-DMA, by plain 16-byte loads into registers,
// by loads + ds_write_b128, and by fragment-shaped loads.  tools/ldsfill_bench.hip waits vmcnt(0) +
// a barrier per 64 KiB block and therefore measures issue + latency, not the rate of the path.
// 256 workgroups (one per CU: 128 KiB of LDS each), NW waves; every workgroup walks the K blocks of
// "its" 256 A rows and 256 W rows of a 4096 x K fp8 operand pair, XCD-grouped like the GEMM's tiles.
// build: hipcc --offload-arch=gfx950 -O3 tools/fillrate.hip -o tools/fillrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// MODE 0 LDS-DMA (8 rows x 128 B per instruction), 1 registers (same pattern), 2 registers + ds_write_b128,
// 3 registers, fragment-shaped (16 rows x 64 B per instruction), 4 LDS-DMA of 1 KiB contiguous (no row pattern)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void fill(const unsigned char* __restrict__ a, const unsigned char* __restrict__ w,
                                                int ld, int nkb, int reps, unsigned* sink, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int wg = blockIdx.x;
  { const int q = 256 / 8, xcd = wg % 8, idx = wg / 8; wg = xcd * q + idx; }
  const int tm = wg % 16, tn = wg / 16;
  constexpr int L = 64 / NW;          // 1 KiB loads per wave and K block (A: L/2, W: L/2)
  static_assert(L >= 2, "");
  unsigned voff[L];
  for (int r = 0; r < L; ++r) {
    int row, col;
    if (MODE == 3) { row = 16 * (NW * r + wave) + (lane & 15); col = (lane >> 4) << 4; }       // + 64 for the second half, see below
    else if (MODE == 4) { row = 8 * (NW * r + wave); col = lane << 4; }                     // 1 KiB contiguous (spans rows of ld)
    else { row = 8 * (NW * r + wave) + (lane >> 3); col = (lane & 7) << 4; }
    if (MODE == 3) {
      // 16 rows x 64 B: instruction r covers rows of half (r & 1)
      const int rr = 16 * (NW * (r >> 1) + wave) + (lane & 15);
      voff[r] = (unsigned)(rr % 256) * (unsigned)ld + (unsigned)(((lane >> 4) << 4) + 64 * (r & 1));
      (void)row; (void)col;
    } else if (MODE == 4) {
      voff[r] = (unsigned)(row % 256) * 128u + (unsigned)col;   // rows of 128 B: a packed operand
    } else {
      voff[r] = (unsigned)(row % 256) * (unsigned)ld + (unsigned)col;
    }
  }
  // the first L/2 loads read A rows, the others W rows (MODE 3: loads alternate halves, same split)
  const unsigned char* abase = a + (size_t)tm * 256 * (MODE == 4 ? 128 : ld);
  const unsigned char* wbase = w + (size_t)tn * 256 * (MODE == 4 ? 128 : ld);
  unsigned acc = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  u32x4 v[2][L];
  auto issue = [&](int kb, int buf) {
    const size_t koff = MODE == 4 ? (size_t)kb * 4096 * 128 : (size_t)kb * 128;
#pragma unroll
    for (int r = 0; r < L; ++r) {
      const unsigned char* g = (r < L / 2 ? abase : wbase) + koff + voff[r];
      if (MODE == 0 || MODE == 4) {
        unsigned char* d = lds + buf * 65536 + (NW * r + wave) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      } else {
        u32x4 t;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(g) : "memory");
        v[buf][r] = t;
      }
    }
  };
  const int total = nkb * reps;
  issue(0, 0);
  for (int i = 0; i < total; i += 2) {
    // two K blocks per trip so that the buffer index is a compile-time constant
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nx = i + h + 1;
      if (nx < total) issue(nx % nkb, (h + 1) & 1);
      if (nx < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE == 0 || MODE == 4) {
        acc += *(volatile unsigned*)(lds + h * 65536 + wave * 1024 + lane * 4);
      } else {
#pragma unroll
        for (int r = 0; r < L; ++r) {
          u32x4 t = v[h][r];
          asm volatile("" : "+v"(t));
          if (MODE == 2) *(u32x4*)(lds + h * 65536 + (NW * r + wave) * 1024 + lane * 16) = t;
          else acc ^= t.x ^ t.w;
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 2) { __syncthreads(); acc += *(unsigned*)(lds + threadIdx.x * 4); }
  if (acc == 0x12345678u) sink[0] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int NW>
static void run(const char* name, const unsigned char* a, const unsigned char* w, int ld, int nkb, int reps, unsigned* sink,
                unsigned long long* clk) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto go = [&]() { hipLaunchKernelGGL((fill<MODE, NW>), dim3(256), dim3(64 * NW), 0, 0, a, w, ld, nkb, reps, sink, clk); };
  go(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int t = 0; t < 3; ++t) {
    CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double ghz = cyc / rt * 0.1;                       // s_memrealtime ticks at 100 MHz
  const double blocks = (double)nkb * reps;
  const double us_blk = best * 1e3 / blocks;
  const double cyc_blk = cyc / 256 / blocks;
  printf("%-44s NW %2d: %.3f us / 64 KiB block = %6.1f GB/s per CU, %5.1f TB/s chip; in-kernel %.2f GHz, %6.0f cycles per block = %.1f B/clk\n",
         name, NW, us_blk, 65536 / us_blk / 1e3, 65536.0 * 256 / us_blk / 1e6, ghz, cyc_blk, 65536 / cyc_blk);
}

int main() {
  const int ld = 16384, rows = 4096, nkb = ld / 128, reps = 4;
  unsigned char *a, *w; unsigned* sink; unsigned long long* clk;
  CK(hipMalloc(&a, (size_t)rows * ld)); CK(hipMalloc(&w, (size_t)rows * ld));
  CK(hipMemset(a, 3, (size_t)rows * ld)); CK(hipMemset(w, 5, (size_t)rows * ld));
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&clk, 512 * 8));
#define RUNALL(MODE, NAME) run<MODE, 4>(NAME, a, w, ld, nkb, reps, sink, clk); run<MODE, 8>(NAME, a, w, ld, nkb, reps, sink, clk); \
  run<MODE, 16>(NAME, a, w, ld, nkb, reps, sink, clk);
  RUNALL(0, "LDS-DMA, 8 rows x 128 B");
  RUNALL(1, "global_load_dwordx4 -> VGPR, 8 rows x 128 B");
  RUNALL(2, "global_load_dwordx4 + ds_write_b128");
  RUNALL(3, "global_load_dwordx4 -> VGPR, 16 rows x 64 B");
  RUNALL(4, "LDS-DMA, 1 KiB contiguous (packed operand)");
  return 0;
}
