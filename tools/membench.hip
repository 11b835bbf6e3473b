// HBM read ceiling for the paged-attention access pattern (tools only, not in the library).
//   mode 0: linear streaming read, 16 B/lane
//   mode 1: random 2 KiB pages (fp8 KV page of one (block, head)), two streams (K and V)
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT>
__global__ __launch_bounds__(256) void rd_linear(const u32x4* __restrict__ p, size_t n_per_wg, unsigned* sink) {
  const u32x4* base = p + (size_t)blockIdx.x * n_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i < n_per_wg; i += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      size_t idx = i + (size_t)j * 256;
      if (idx < n_per_wg) v[j] = NT ? __builtin_nontemporal_load(base + idx) : base[idx];
      else v[j] = (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// each wave walks a list of pages; per page 2 KiB = 2 x (64 lanes x 16 B); 4 pages of K + 4 of V per step
// like rd_paged<.,2048> but V is read the way the attention kernel reads it: lane group g4 reads
// page g4 of the step, 16 lanes x 16 B = 256 B contiguous, 8 instructions cover the 2 KiB page
template <bool NT>
__global__ __launch_bounds__(256) void rd_paged_vseg(const unsigned char* __restrict__ k, const unsigned char* __restrict__ v,
                                                     const int* __restrict__ table, int pages_per_wg, size_t block_stride,
                                                     int heads, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int seq = blockIdx.x / heads, head = blockIdx.x % heads;
  const int* bt = table + (size_t)seq * pages_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (int pg = wave * 4; pg < pages_per_wg; pg += 16) {
    u32x4 r[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t off = (size_t)bt[pg + t] * block_stride + (size_t)head * 2048;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4* pk = (const u32x4*)(k + off + i * 1024 + lane * 16);
        r[t * 2 + i] = NT ? __builtin_nontemporal_load(pk) : *pk;
      }
    }
    const size_t voff = (size_t)bt[pg + g4] * block_stride + (size_t)head * 2048;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const u32x4* pv = (const u32x4*)(v + voff + (16 * m + c16) * 16);
      r[8 + m] = NT ? __builtin_nontemporal_load(pv) : *pv;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= r[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <bool NT, int PAGE_BYTES>
__global__ __launch_bounds__(256) void rd_paged(const unsigned char* __restrict__ k, const unsigned char* __restrict__ v,
                                                const int* __restrict__ table, int pages_per_wg, size_t block_stride,
                                                int heads, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seq = blockIdx.x / heads, head = blockIdx.x % heads;
  const int* bt = table + (size_t)seq * pages_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  constexpr int LPP = PAGE_BYTES / 1024;  // loads per page per lane
  for (int pg = wave * 4; pg < pages_per_wg; pg += 16) {
    u32x4 r[8 * LPP];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t off = (size_t)bt[pg + t] * block_stride + (size_t)head * PAGE_BYTES;
#pragma unroll
      for (int i = 0; i < LPP; ++i) {
        const u32x4* pk = (const u32x4*)(k + off + i * 1024 + lane * 16);
        const u32x4* pv = (const u32x4*)(v + off + i * 1024 + lane * 16);
        r[(t * LPP + i) * 2] = NT ? __builtin_nontemporal_load(pk) : *pk;
        r[(t * LPP + i) * 2 + 1] = NT ? __builtin_nontemporal_load(pv) : *pv;
      }
    }
#pragma unroll
    for (int j = 0; j < 8 * LPP; ++j) acc ^= r[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

// skinny-GEMM weight stream candidates: W[N][K] bytes, a workgroup of 8 waves owns 16 rows,
// wave w the K range [w*K/8, (w+1)*K/8).
//  mode 0: MFMA-operand order straight to VGPRs: lane (r = lane&15, g = lane>>4) reads 16 B at
//          row r, k = 64*i + 16*g  (16 rows x 64 B contiguous per instruction)
//  mode 1: coalesced order: lanes 0-15 read 256 contiguous bytes of one row, 4 rows per instruction
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void rd_rows16(const unsigned char* __restrict__ w, int K, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t row0 = (size_t)blockIdx.x * 16;
  const int kw = K / 8;  // bytes per wave per row
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    const int r = lane & 15, g = lane >> 4;
    const unsigned char* base = w + (row0 + r) * K + wave * kw + 16 * g;
    for (int k = 0; k < kw; k += 512) {
      u32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x4* p = (const u32x4*)(base + k + 64 * i);
        v[i] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i];
    }
  } else {
    const int c = lane & 15, rr = lane >> 4;
    for (int k = 0; k < kw; k += 256) {
      u32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4* p = (const u32x4*)(w + (row0 + 4 * i + rr) * K + wave * kw + k + 16 * c);
        v[i] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[i];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> float timeit(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f(i);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters * 1e3f;
}

int main() {
  const size_t bytes = 512ull << 20;  // per buffer (like configs[1]: 256 MiB K + 256 MiB V per cache)
  unsigned char *bufs[4]; unsigned* sink;
  for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], bytes / 2)); CK(hipMemset(bufs[i], i + 1, bytes / 2)); }
  CK(hipMalloc(&sink, 4));
  // linear: 512 MiB as 2 x 256 MiB buffers alternating sets
  for (int wgs : {512, 1024, 2048}) {
    size_t n_per_wg = (bytes / 2) / 16 / wgs;
    for (int nt = 0; nt < 2; ++nt) {
      float us = timeit([&](int it) {
        for (int h = 0; h < 2; ++h) {
          const u32x4* p = (const u32x4*)bufs[(it & 1) * 2 + h];
          if (nt) hipLaunchKernelGGL(rd_linear<true>, dim3(wgs), dim3(256), 0, 0, p, n_per_wg, sink);
          else hipLaunchKernelGGL(rd_linear<false>, dim3(wgs), dim3(256), 0, 0, p, n_per_wg, sink);
        }
      }, 20);
      printf("linear wgs=%d nt=%d: %.1f us (2 launches) -> %.0f GB/s\n", wgs, nt, us, bytes / us / 1e3);
    }
  }
  {
    // skinny GEMM weight stream: two 256 MiB weight matrices alternated, K = 4096 and 14336
    for (int K : {4096, 14336}) {
      const int N = (int)((bytes / 2) / K / 16 * 16);
      for (int mode = 0; mode < 2; ++mode) for (int nt = 0; nt < 2; ++nt) {
        float us = timeit([&](int it) {
          const unsigned char* wp = bufs[(it & 1) * 2];
          if (mode == 0 && nt == 0) hipLaunchKernelGGL((rd_rows16<0, false>), dim3(N / 16), dim3(512), 0, 0, wp, K, sink);
          if (mode == 0 && nt == 1) hipLaunchKernelGGL((rd_rows16<0, true>), dim3(N / 16), dim3(512), 0, 0, wp, K, sink);
          if (mode == 1 && nt == 0) hipLaunchKernelGGL((rd_rows16<1, false>), dim3(N / 16), dim3(512), 0, 0, wp, K, sink);
          if (mode == 1 && nt == 1) hipLaunchKernelGGL((rd_rows16<1, true>), dim3(N / 16), dim3(512), 0, 0, wp, K, sink);
        }, 20);
        printf("rows16 K=%d N=%d mode=%d nt=%d: %.1f us -> %.0f GB/s\n", K, N, mode, nt, us, (double)N * K / us / 1e3);
      }
    }
  }
  // paged: S=64 seqs x 256 blocks, 8 heads, fp8 page 2 KiB (block = 16 KiB), bf16 page 4 KiB
  for (int page_bytes : {2048, 4096}) {
    const int S = 64, heads = 8;
    const size_t block_stride = (size_t)heads * page_bytes;
    const int nblocks = (int)((bytes / 2) / block_stride);
    const int pages_per_wg = nblocks / S;
    std::vector<int> tab(nblocks);
    for (int i = 0; i < nblocks; ++i) tab[i] = i;
    std::mt19937 rng(0); std::shuffle(tab.begin(), tab.end(), rng);
    int* dtab; CK(hipMalloc(&dtab, nblocks * 4)); CK(hipMemcpy(dtab, tab.data(), nblocks * 4, hipMemcpyHostToDevice));
    for (int nt = 0; nt < 2; ++nt) {
      float us = timeit([&](int it) {
        unsigned char* k = bufs[(it & 1) * 2], *v = bufs[(it & 1) * 2 + 1];
        if (page_bytes == 2048) {
          if (nt) hipLaunchKernelGGL((rd_paged<true, 2048>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
          else hipLaunchKernelGGL((rd_paged<false, 2048>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
        } else {
          if (nt) hipLaunchKernelGGL((rd_paged<true, 4096>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
          else hipLaunchKernelGGL((rd_paged<false, 4096>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
        }
      }, 20);
      printf("paged page=%d nt=%d (512 WGs x 256 thr): %.1f us -> %.0f GB/s\n", page_bytes, nt, us, bytes / us / 1e3);
      if (page_bytes == 2048) {
        us = timeit([&](int it) {
          unsigned char* k = bufs[(it & 1) * 2], *v = bufs[(it & 1) * 2 + 1];
          if (nt) hipLaunchKernelGGL((rd_paged_vseg<true>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
          else hipLaunchKernelGGL((rd_paged_vseg<false>), dim3(S * heads), dim3(256), 0, 0, k, v, dtab, pages_per_wg, block_stride, heads, sink);
        }, 20);
        printf("paged vseg page=2048 nt=%d: %.1f us -> %.0f GB/s\n", nt, us, bytes / us / 1e3);
      }
    }
  }
  return 0;
}
