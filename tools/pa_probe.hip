// Where do the microseconds of the decode kernel go?  (tools only, not in the library)
// Compiles the library's own kernel source with -DVA_PROBES: per-workgroup s_memrealtime stamps
// (entry, first K data consumed, loop end, exit) and two ablations (loads only; no merge), on
// the configs[1] shape (64 seqs x 4096 tokens, 32q/8kv, fp8, block 16) or any S / ctx / KVH.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVA_PROBES -DNDEBUG tools/pa_probe.hip \
//        vllm_amd/csrc/common.hip -o tools/pa_probe
#include "../vllm_amd/csrc/paged_attention_mfma.hip"

#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>

namespace va {
int paged_attention_reduce(const PagedAttnParams&) { return 0; }
}
using namespace va;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VARIANT>
static void launch(bf16_t* out, const bf16_t* q, const uint8_t* kc, const uint8_t* vc, const int* bt,
                   const int* sl, const float* one, int S, int H, int KVH, int max_blocks) {
  PaK<bf16_t> A{};
  A.out = out; A.q = q; A.k_cache = kc; A.v_cache = vc; A.num_heads = H; A.num_kv_heads = KVH;
  A.scale = 0.088f; A.block_tables = bt; A.seq_lens = sl; A.max_num_blocks_per_seq = max_blocks;
  A.q_stride = (int64_t)H * 128; A.kv_block_bytes = (int64_t)KVH * 2048; A.kv_head_bytes = 2048;
  A.k_scale_p = one; A.v_scale_p = one; A.partition_size = 0; A.max_parts = 1; A.gqa_chunks = 1;
  A.slot_stride = 1; A.dr_arg = 128; A.num_seqs = S;
  hipLaunchKernelGGL((pa_mfma_kernel<bf16_t, true, 128, 16, 4, VARIANT, false, false>), dim3(S * KVH), dim3(256),
                     0, 0, PA_ARGS(A));
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 64, ctx = argc > 2 ? atoi(argv[2]) : 4096;
  const int KVH = argc > 3 ? atoi(argv[3]) : 8, H = KVH * 4;
  const int maxcode = argc > 4 ? atoi(argv[4]) : 31;   // fp8 codes 0 .. maxcode-1 (bench data: |x| < 0.088)
  const int nb_seq = ctx / 16, NB = S * nb_seq;
  const size_t cache_bytes = (size_t)NB * KVH * 2048;
  uint8_t *kc[2], *vc[2];
  std::vector<uint8_t> hb(cache_bytes);
  std::mt19937 rng(1);
  for (size_t i = 0; i < cache_bytes; ++i) hb[i] = (uint8_t)((rng() % maxcode) | ((rng() & 1) << 7));
  // argv[5]: extra bytes between the K and the V cache of a set (one big allocation, carved), to
  // see what the distance between the two streams does to the HBM channels / banks
  const size_t gap = argc > 5 ? (size_t)atoll(argv[5]) : 0;
  uint8_t* pool;
  CK(hipMalloc(&pool, 4 * cache_bytes + 4 * gap + (64 << 20)));
  for (int c = 0; c < 2; ++c) {
    kc[c] = pool + (size_t)c * (2 * cache_bytes + 2 * gap);
    vc[c] = kc[c] + cache_bytes + gap;
    CK(hipMemcpy(kc[c], hb.data(), cache_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(vc[c], hb.data() + 7, cache_bytes - 7, hipMemcpyHostToDevice));
  }
  printf("S %d ctx %d KVH %d gap %zu: K-V distance %zu bytes\n", S, ctx, KVH, gap, (size_t)(vc[0] - kc[0]));
  std::vector<int> tab(NB);
  for (int i = 0; i < NB; ++i) tab[i] = i;
  std::shuffle(tab.begin(), tab.end(), rng);
  int *bt, *sl; float* one; bf16_t *q, *out;
  CK(hipMalloc(&bt, NB * 4)); CK(hipMemcpy(bt, tab.data(), NB * 4, hipMemcpyHostToDevice));
  std::vector<int> lens(S, ctx);
  CK(hipMalloc(&sl, S * 4)); CK(hipMemcpy(sl, lens.data(), S * 4, hipMemcpyHostToDevice));
  float h1 = 1.f; CK(hipMalloc(&one, 4)); CK(hipMemcpy(one, &h1, 4, hipMemcpyHostToDevice));
  std::vector<uint16_t> hq((size_t)S * H * 128);
  for (auto& x : hq) x = (uint16_t)(0x3c00 + (rng() % 512)) ^ (uint16_t)((rng() & 1) << 15);   // ~ +-0.01
  CK(hipMalloc(&q, hq.size() * 2)); CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, hq.size() * 2));
  const double bytes = 2.0 * cache_bytes;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // clocks: ~1.5 s of back-to-back launches before anything is timed
  for (int i = 0; i < 15000; ++i) launch<1>(out, q, kc[i & 1], vc[i & 1], bt, sl, one, S, H, KVH, nb_seq);
  CK(hipDeviceSynchronize());
  const char* names[4] = {"full (variant 1)", "loads only (5)", "no merge (9)", "loads only + no merge (13)"};
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 2; ++v) {
      auto go = [&](int it) {
        const int c = it & 1;
        if (v == 0) launch<1>(out, q, kc[c], vc[c], bt, sl, one, S, H, KVH, nb_seq);
        if (v == 1) launch<5>(out, q, kc[c], vc[c], bt, sl, one, S, H, KVH, nb_seq);
        if (v == 2) launch<9>(out, q, kc[c], vc[c], bt, sl, one, S, H, KVH, nb_seq);
        if (v == 3) launch<13>(out, q, kc[c], vc[c], bt, sl, one, S, H, KVH, nb_seq);
      };
      for (int i = 0; i < 3; ++i) go(i);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int iters = 20;
      for (int i = 0; i < iters; ++i) go(i);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms / iters * 1e3;
      if (round == 2) printf("%-28s %7.2f us  %6.0f GB/s\n", names[v], us, bytes / us / 1e3);
      if (round == 2 && v == 0 && !getenv("PA_PROBE_BRIEF")) {
        // stamps of the LAST launch: per workgroup entry / first data / loop end / exit, in us
        // after the earliest entry
        const int nwg = S * KVH;
        std::vector<unsigned long long> st((size_t)nwg * 4);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_pa_stamps), st.size() * 8));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < nwg; ++w) t0 = std::min(t0, st[4 * w]);
        for (int k = 0; k < 4; ++k) {
          std::vector<double> v4(nwg);
          for (int w = 0; w < nwg; ++w) v4[w] = (double)(st[4 * w + k] - t0) / 100.0;
          std::sort(v4.begin(), v4.end());
          const char* lab[4] = {"entry", "first K consumed", "loop end", "exit"};
          printf("    %-18s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us\n", lab[k], v4[0],
                 v4[nwg / 10], v4[nwg / 2], v4[nwg * 9 / 10], v4[nwg - 1]);
        }
      }
    }
  return 0;
}
