// fp8 scaled GEMM for gfx950, M > 32: BM x BN x 128B tiles, direct-to-LDS staging
// (global_load_lds, 16 B per lane) with the NEXT K block in flight under the MFMAs of the
// current one.  One kernel template, several tile shapes (launch_glds_shape at the bottom):
//   256x256 for MFMA-bound problems, 128x128 / 128x64 / 64x64 / 64x32 for the mid-M range
//   where the point is to spread the weight stream over every CU; the small tiles also come
//   in a variant that takes KB = 2 or 4 K blocks per barrier.
//
//   out[M,N] = ((sa (.) A[M,K]) . (sb (.) W[N,K]^T)).to(out_t) (+ bias)      K % 128 == 0
//
// Structure (cdna_hip_programming.md section 5, "glds, 2 LDS buffers"):
//   * K block = 128 bytes = one v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 scales: the
//     only fp8 MFMA that runs at the 2x rate); waves as WM(M) x WN(N), a wave owns
//     (BM/WM) x (BN/WN) outputs = MT x NT MFMA tiles;
//   * the product is computed transposed (A operand = weight rows, B operand = activation
//     rows) so a lane holds 4 consecutive n of one output row;
//   * LDS image per operand and stage: [rows][128 B], 16-byte chunks XOR-swizzled by
//     f(row) (same f as scaled_mm.hip: conflict-free ds_read_b128).  global_load_lds writes
//     lane-linearly (wave base + 16*lane), so the swizzle is applied to the per-lane SOURCE
//     address and to the read address -- never to the destination;
//   * ONE barrier per K block (or per group of KB blocks): wait own DMA of the current
//     buffer -> s_barrier -> issue the DMA that refills the buffer consumed last time
//     (everyone is past its reads of it) -> fragment reads + MFMAs of the current buffer.
//     A ring of NST buffers keeps NST-1 block groups in flight;
//   * fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt: with C++ LDS
//     loads hipcc orders every ds_read behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)
//     right after the barrier), which serialises the prefetch with the math.  The waits
//     carry the fragment registers as in/out operands, so the MFMAs depend on them;
//   * stores go through LDS (the tile overlays the staging buffers) so that every store
//     instruction writes whole rows of the tile: direct stores from the MFMA layout write
//     32-byte pieces of 16 rows per instruction and cost 25 us on a 32 MB output.
#include "common.cuh"

#include <mutex>

namespace va {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int LBK = 128;

// measurement builds only (tools/): -DVA_MM_ABLATE=1 no LDS-DMA refill inside the K loop,
// 2 no MFMA (fragments kept alive), 3 no fragment reads (MFMA on whatever the registers hold).
// Results are garbage; the time says what each part of the loop costs.
#ifndef VA_MM_ABLATE
#define VA_MM_ABLATE 0
#endif
// -DVA_PROBES: shader-clock (s_memtime) and 100 MHz (s_memrealtime) stamps around the K loop of
// every workgroup -> in-kernel clock and cycles per K block (tools/mm_stamps.py)
#ifdef VA_PROBES
__device__ unsigned long long g_mm_stamps[4096 * 4];
__device__ unsigned long long g_mm_stamps_end[4096];
// per wave of the first 512 workgroups, summed over the K blocks: cycles waiting for the block's
// LDS-DMA + barrier, from the barrier to the first MFMA's fragments, and for the rest of the block
__device__ unsigned long long g_mm_phase[512 * 8 * 3];
#define VA_MM_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define VA_MM_T(x) do { } while (0)
#endif

__device__ __forceinline__ int lswz(int row) {
  return ((row >> 1) & 1) | (((row >> 2) & 1) << 2);
}

template <int OFF>
__device__ __forceinline__ i32x4 lds_read16(uint32_t addr) {
  i32x4 v;
#if VA_MM_ABLATE == 3
  asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
#endif
  return v;
}

// fragment T (16 rows) of an operand, BASE bytes into the ring: two 16-byte halves at the
// per-lane addresses a0/a1
template <int BASE, int T>
__device__ __forceinline__ void lds_read_frag(uint32_t a0, uint32_t a1, i32x4& lo, i32x4& hi) {
  static_assert(BASE + T * 16 * LBK < 65536, "ds_read offset is 16 bits");
  lo = lds_read16<BASE + T * 16 * LBK>(a0);
  hi = lds_read16<BASE + T * 16 * LBK>(a1);
}

// first reads of a K block, in the order its MFMAs consume them: W0, A0, W1 .. W(WA-1), where
// WA = min(NT, 4) weight fragments are requested up front (lgkmcnt has 4 bits: at most 15 reads
// can be counted); the first m-step requests W(nt + WA) in front of its MFMA nt.
template <int NT> constexpr int w_ahead() { return NT < 4 ? NT : 4; }
template <int BASE, int NT>
__device__ __forceinline__ void block_start(uint32_t w0, uint32_t w1, uint32_t a0, uint32_t a1,
                                            i32x4 (&wl)[NT], i32x4 (&wh)[NT], i32x4& al, i32x4& ah) {
  lds_read_frag<BASE, 0>(w0, w1, wl[0], wh[0]);
  lds_read_frag<BASE, 0>(a0, a1, al, ah);
  if constexpr (NT > 1) lds_read_frag<BASE, 1>(w0, w1, wl[1], wh[1]);
  if constexpr (NT > 2) {
    lds_read_frag<BASE, 2>(w0, w1, wl[2], wh[2]);
    lds_read_frag<BASE, 3>(w0, w1, wl[3], wh[3]);
  }
}
// reads of W fragments younger than W(nt) at the moment MFMA nt of the first m-step waits
template <int NT> constexpr int w_younger(int nt) {
  return (nt + w_ahead<NT>() < NT - 1 ? nt + w_ahead<NT>() : NT - 1) - nt;
}

// s_waitcnt lgkmcnt(LATER) = "all LDS reads but the LATER youngest have returned".  The
// fragments pass through the asm as in/out operands, so their consumers stay behind it.
template <int LATER>
__device__ __forceinline__ void wait_frag(i32x4& a0, i32x4& a1) {
  static_assert(LATER <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a0), "+v"(a1) : [n] "n"(LATER));
}
template <int LATER>
__device__ __forceinline__ void wait_wa(i32x4& w0, i32x4& w1, i32x4& a0, i32x4& a1) {
  static_assert(LATER <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(w0), "+v"(w1), "+v"(a0), "+v"(a1) : [n] "n"(LATER));
}

__device__ __forceinline__ f32x4 mm_mfma(i32x8 w, i32x8 a, f32x4 acc) {
#if VA_MM_ABLATE == 2
  asm volatile("" ::"v"(w), "v"(a));
  return acc;
#else
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
}

__device__ __forceinline__ i32x8 cat8(i32x4 lo, i32x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename out_t, int BM, int BN, int WM, int WN, int NST, int KB, bool VEC_STORE,
          bool SPLITK, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN) void scaled_mm_glds_kernel(
    out_t* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, const float* __restrict__ b_scales,
    const out_t* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
    int a_scale_numel, int b_scale_numel, float* ws, unsigned* ctr, int S) {
  const bool nt_store = (int64_t)M * N * (int64_t)sizeof(out_t) >= ((int64_t)128 << 20);
  constexpr int NW = WM * WN;
#ifdef VA_MM_LATE_FILL   // measurement builds: see the K loop
  constexpr bool LATE_FILL = NW == 8 && NST == 2 && !SPLITK;
#else
  constexpr bool LATE_FILL = false;
#endif

  constexpr int A_BYTES = BM * LBK, W_BYTES = BN * LBK, STAGE = A_BYTES + W_BYTES;
  constexpr int MT = BM / WM / 16, NT = BN / WN / 16;   // MFMA tiles per wave
#ifndef VA_MM_PF
#define VA_MM_PF 1
#endif
  // fragment prefetch distance of the one-block-per-barrier kernels, in m-steps
  constexpr int PF = (KB == 1 && MT >= 4) ? VA_MM_PF : 1;
  constexpr int RPR = 8 * NW;                            // rows staged per glds round
  constexpr int A_ROUNDS = BM / RPR, W_ROUNDS = BN / RPR;
  static_assert(BM % RPR == 0 && BN % RPR == 0, "tile rows must fill whole glds rounds");
  static_assert(NT == 1 || NT == 2 || NT == 4 || NT == 8, "");
  constexpr int WA = w_ahead<NT>();
  static_assert(MT >= 1 && MT <= 8, "");
  // NST == 5 (square tiles, KB = 1): not five stages but a ring of five HALF stages -- the A and
  // the W image of a K block are staged separately, in the order A0 W0 A1 W1 .., so that one half
  // stage is always in flight across a barrier (two whole stages leave the fill path idle for a
  // load latency after every barrier; three do not fit the 160 KB).  See the R5 loop below.
  constexpr bool R5 = NST == 5;
  static_assert(!R5 || (KB == 1 && A_BYTES == W_BYTES && !SPLITK), "");
  constexpr int LDS_TOTAL = R5 ? 5 * A_BYTES : NST * KB * STAGE;
  static_assert(NST >= 2 && LDS_TOTAL <= 160 * 1024, "");
  constexpr int LOADS = A_ROUNDS + W_ROUNDS;           // glds per lane and stage
  static_assert(R5 || (NST - 2) * KB * LOADS <= 63, "vmcnt is 6 bits");
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TOTAL];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    // hardware workgroup i runs on XCD i % 8: give every XCD a contiguous run of virtual ids
    // (split-K: of one K slice), so the tiles that share a weight panel share an L2
    const int tot = SPLITK ? nwg * S : nwg;
    const int q = tot / 8, r = tot % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int z = 0;   // K slice of this workgroup
  if constexpr (SPLITK) {
    z = wg / nwg;
    wg -= z * nwg;
  }
  const int tm = wg % tiles_m, tn = wg / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: in round r, wave w fills rows [RPR r + 8 w, +8): lane -> (row = lane>>3, slot = lane&7)
  // and loads logical chunk slot ^ f(row) so that the linear LDS image is the swizzled tile.
  // Sources are 32-bit offsets from the (uniform) operand base; the host checks they fit.
  const int srow = lane >> 3, sslot = lane & 7;
  uint32_t a_voff[A_ROUNDS], w_voff[W_ROUNDS];
#pragma unroll
  for (int r = 0; r < A_ROUNDS; ++r) {
    const int row = RPR * r + 8 * wave + srow;
    int m = m0 + row;
    m = m < M ? m : M - 1;   // rows past M: any valid row (their outputs are never stored)
    a_voff[r] = (uint32_t)m * (uint32_t)lda + ((sslot ^ lswz(row)) << 4);
  }
#pragma unroll
  for (int r = 0; r < W_ROUNDS; ++r) {
    const int row = RPR * r + 8 * wave + srow;
    int n = n0 + row;
    n = n < N ? n : N - 1;
    w_voff[r] = (uint32_t)n * (uint32_t)ldb + ((sslot ^ lswz(row)) << 4);
  }
  auto stage_load = [&](int stage, int k0) {
    uint8_t* base = lds + stage * STAGE + 8 * wave * LBK;
    const uint8_t* ag = a + k0;
    const uint8_t* bg = b + k0;
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(ag + a_voff[r]),
          (__attribute__((address_space(3))) void*)(base + RPR * r * LBK), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(bg + w_voff[r]),
          (__attribute__((address_space(3))) void*)(base + A_BYTES + RPR * r * LBK), 16, 0, 0);
  };

  // load i of a stage (0 .. LOADS-1: the A rounds, then the W rounds), for refills that are
  // spread over the m-steps of a block instead of being issued back to back
  auto stage_load_one = [&](auto ic, int stage, int k0) {
    constexpr int i = decltype(ic)::value;
    uint8_t* base = lds + stage * STAGE + 8 * wave * LBK;
    if constexpr (i < A_ROUNDS)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a + k0 + a_voff[i]),
          (__attribute__((address_space(3))) void*)(base + RPR * i * LBK), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(b + k0 + w_voff[i - A_ROUNDS]),
          (__attribute__((address_space(3))) void*)(base + A_BYTES + RPR * (i - A_ROUNDS) * LBK), 16, 0, 0);
  };
  // Measured with in-kernel stamps (profiles/r02_gemm_ablation.txt, section 6): a wave that issues
  // its 8 LDS-DMA loads back to back behind the barrier sits ~600 cycles (waves 4-7: ~1500) in the
  // vector-memory issue queue -- 8 waves x 8 loads share one address unit, ~35 cycles per load --
  // before it reaches its first MFMA.  Spreading them, one load behind the MFMAs of each m-step
  // (-DVA_MM_SPREAD=1), brings the first MFMA forward (238 cycles after the barrier) but every
  // load then blocks its wave inside the MFMA section (1072 -> 1690 / 2540 cycles): 3166 instead
  // of 2964 cycles per K block.  The address unit's ~2300 cycles per block are the limit either
  // way; back-to-back issue stays the default.
#ifndef VA_MM_SPREAD
#define VA_MM_SPREAD 0
#endif
  constexpr bool SPREAD = VA_MM_SPREAD && KB == 1 && NST == 2 && LOADS <= MT && !R5;
  // MODE 1 (8 waves, one block per barrier, two stages): the SECOND wave of every SIMD (waves
  // 4-7) stages the whole tile, two loads behind the MFMAs of each of its m-steps; waves 0-3 issue
  // no vector-memory instruction at all, so one wave per SIMD always reaches the matrix pipe while
  // its partner waits at the address unit.  Loader lw = wave - 4 fills rows [32 r + 8 lw, +8) of
  // round r = 0 .. BM/32-1 of either operand.
  constexpr bool ASYM = MODE == 1;
  static_assert(!ASYM || (NW == 8 && KB == 1 && NST == 2 && BM == BN && BM / 32 == MT && !SPLITK), "");
  // (the source offsets are recomputed per load -- the swizzle does not depend on the round -- so
  // that the loaders need no second set of offset registers)
  const int arow0 = 8 * (wave & 3) + srow;
  const uint32_t asw = (uint32_t)((sslot ^ lswz(arow0)) << 4);
  auto asym_load = [&](auto ic, int stage, int k0) {   // round r of both operands
    constexpr int r = decltype(ic)::value;
    int row = 32 * r + arow0;
    asm volatile("" : "+v"(row));   // keep hipcc from hoisting the 16 offsets out of the K loop (spills)
    const int m = min(m0 + row, M - 1), n = min(n0 + row, N - 1);
    uint8_t* base = lds + stage * STAGE + (32 * r + 8 * (wave & 3)) * LBK;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(a + k0 + ((uint32_t)m * (uint32_t)lda + asw)),
        (__attribute__((address_space(3))) void*)base, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(b + k0 + ((uint32_t)n * (uint32_t)ldb + asw)),
        (__attribute__((address_space(3))) void*)(base + A_BYTES), 16, 0, 0);
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef VA_PROBES
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  // fragment read addresses: the swizzle depends on c16 only, so tile t is an immediate
  // offset t * 16 * LBK from one per-lane base per 16-byte half (j) and operand
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  uint32_t w_addr[2], a_addr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = ((2 * g4 + j) ^ lswz(c16)) << 4;
    w_addr[j] = lds_base + A_BYTES + (wn * (BN / WN) + c16) * LBK + ch;
    a_addr[j] = lds_base + (wm * (BM / WM) + c16) * LBK + ch;
  }

  // Ring of NST groups of KB blocks.  One wait + barrier per GROUP: groups g+1 .. g+NST-2 stay
  // in flight while group g is consumed, and group g+NST-1 is issued into the buffers group
  // g-1 just left.  (KB > 1 is for small tiles, where a barrier per block would leave the
  // latency chain barrier -> ds_read -> MFMA exposed once per 2..4 MFMAs.)
  int ng = K / (LBK * KB);   // host guarantees K % (128 KB) == 0
  if constexpr (SPLITK) {
    // slice z takes groups [ng z / S, ng (z+1) / S): bump the (uniform) operand bases
    const int g0 = (int)((int64_t)ng * z / S);
    ng = (int)((int64_t)ng * (z + 1) / S) - g0;
    a += (int64_t)g0 * (KB * LBK);
    b += (int64_t)g0 * (KB * LBK);
  }
  auto group_load = [&](int buf, int g) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) stage_load(buf * KB + kb, (g * KB + kb) * LBK);
  };
  if constexpr (R5) {
    auto load_a = [&](int slot, int k0) {
      uint8_t* base = lds + slot * A_BYTES + 8 * wave * LBK;
#pragma unroll
      for (int r = 0; r < A_ROUNDS; ++r)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a + k0 + a_voff[r]),
            (__attribute__((address_space(3))) void*)(base + RPR * r * LBK), 16, 0, 0);
    };
    auto load_w = [&](int slot, int k0) {
      uint8_t* base = lds + slot * W_BYTES + 8 * wave * LBK;
#pragma unroll
      for (int r = 0; r < W_ROUNDS; ++r)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(b + k0 + w_voff[r]),
            (__attribute__((address_space(3))) void*)(base + RPR * r * LBK), 16, 0, 0);
    };
    // issue order A0 W0 A1 | W1 A2 | W2 A3 | ..: block g sits in slots 2g, 2g+1 (mod 5); behind
    // barrier g the slots of block g-1 take W(g+1) and A(g+2); A(g+1) is the one half stage
    // that is still in flight when barrier g is reached
    load_a(0, 0);
    load_w(1, 0);
    if (ng > 1) load_a(2, LBK);
    int sa = 0;   // slot of A(g)
    for (int g = 0; g < ng; ++g) {
      if (g + 1 < ng) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ROUNDS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int sw = sa + 1 >= 5 ? sa - 4 : sa + 1;
      const int sw1 = sa + 3 >= 5 ? sa - 2 : sa + 3, sa2 = sa + 4 >= 5 ? sa - 1 : sa + 4;
      const uint32_t wa0 = w_addr[0] - A_BYTES + sw * W_BYTES, wa1 = w_addr[1] - A_BYTES + sw * W_BYTES;
      const uint32_t aa0 = a_addr[0] + sa * A_BYTES, aa1 = a_addr[1] + sa * A_BYTES;
      i32x4 wl[NT], wh[NT], al[2], ah[2];
      i32x8 wf[NT];
      block_start<0>(wa0, wa1, aa0, aa1, wl, wh, al[0], ah[0]);
      if (g + 1 < ng) load_w(sw1, (g + 1) * LBK);
      if (g + 2 < ng) load_a(sa2, (g + 2) * LBK);
      sa = sa + 2 >= 5 ? sa - 3 : sa + 2;
      static_for<MT>([&](auto ic) {
        constexpr int mt = decltype(ic)::value;
        constexpr bool more = mt + 1 < MT;
        if constexpr (more) lds_read_frag<0, mt + 1>(aa0, aa1, al[(mt + 1) & 1], ah[(mt + 1) & 1]);
        constexpr int later = more ? 2 : 0;
        if constexpr (mt == 0) {
          if constexpr (WA < NT) lds_read_frag<0, WA>(wa0, wa1, wl[WA], wh[WA]);
          wait_wa<later + 2 * w_younger<NT>(0)>(wl[0], wh[0], al[0], ah[0]);
          const i32x8 af = cat8(al[0], ah[0]);
          wf[0] = cat8(wl[0], wh[0]);
          acc[0][0] = mm_mfma(wf[0], af, acc[0][0]);
          __builtin_amdgcn_sched_barrier(0);
          static_for<NT - 1>([&](auto jc) {
            constexpr int nt = decltype(jc)::value + 1;
            if constexpr (nt + WA < NT) lds_read_frag<0, nt + WA>(wa0, wa1, wl[nt + WA], wh[nt + WA]);
            wait_frag<(nt < WA ? later : 0) + 2 * w_younger<NT>(nt)>(wl[nt], wh[nt]);
            wf[nt] = cat8(wl[nt], wh[nt]);
            acc[nt][0] = mm_mfma(wf[nt], af, acc[nt][0]);
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          wait_frag<later>(al[mt & 1], ah[mt & 1]);
          const i32x8 af = cat8(al[mt & 1], ah[mt & 1]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mm_mfma(wf[nt], af, acc[nt][mt]);
        }
      });
    }
  }
  if constexpr (ASYM) {
    if (wave >= 4) static_for<MT>([&](auto ic) { asym_load(ic, 0, 0); });
  }
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (!R5 && !ASYM && s < ng) group_load(s, s);
  int cur = 0, nxt = NST - 1;   // buffer of group g / of group g + NST - 1
#ifdef VA_PROBES
  unsigned long long ph_wait = 0, ph_first = 0, ph_rest = 0;
#endif
  for (int g = 0; !R5 && g < ng; ++g) {
    VA_MM_T(ph_t0);
    // own DMA of group g has landed (in the steady state everything but the NST-2 younger
    // groups; near the end fewer are in flight, so drain); then everybody's has, and
    // everybody is done reading the buffer that is restaged next
#if VA_MM_ABLATE == 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (g + NST - 1 <= ng) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * KB * LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    VA_MM_T(ph_t1);

    const uint32_t so = cur * (KB * STAGE);
    const uint32_t wa0 = w_addr[0] + so, wa1 = w_addr[1] + so;
    const uint32_t aa0 = a_addr[0] + so, aa1 = a_addr[1] + so;

    // Software pipeline over the KB * MT steps (kb, mt) of the group, fully unrolled: the
    // fragments of step s+1 are requested before the MFMAs of step s, and every wait counts
    // exactly the reads issued after the ones it needs.  A block starts with the reads
    // W0, A0, W1 .. W(NT-1): its first MFMA waits for 4 reads, not for all of them.
    static_assert(MT >= 2, "the wait counts below assume a block has >= 2 m-steps");
    if constexpr (PF == 2) {
      // one block per barrier, activation fragments requested TWO m-steps ahead (three register
      // sets): under the LDS-DMA refill an LDS read takes longer than one m-step of MFMAs
      i32x4 wl[NT], wh[NT], al[3], ah[3];
      i32x8 wf[NT];
      block_start<0>(wa0, wa1, aa0, aa1, wl, wh, al[0], ah[0]);
      lds_read_frag<0, 1>(aa0, aa1, al[1], ah[1]);
#if VA_MM_ABLATE != 1
      if (g + NST - 1 < ng) group_load(nxt, g + NST - 1);
#endif
      nxt = cur;
      cur = cur + 1 == NST ? 0 : cur + 1;
      static_for<MT>([&](auto ic) {
        constexpr int mt = decltype(ic)::value;
        if constexpr (mt + 2 < MT) lds_read_frag<0, mt + 2>(aa0, aa1, al[(mt + 2) % 3], ah[(mt + 2) % 3]);
        constexpr int pend = 2 * (MT - 1 - mt < 2 ? MT - 1 - mt : 2);   // younger A reads
        if constexpr (mt == 0) {
          wait_wa<pend + 2 * (NT - 1)>(wl[0], wh[0], al[0], ah[0]);
          const i32x8 af = cat8(al[0], ah[0]);
          wf[0] = cat8(wl[0], wh[0]);
          acc[0][0] = mm_mfma(wf[0], af, acc[0][0]);
          __builtin_amdgcn_sched_barrier(0);
          static_for<NT - 1>([&](auto jc) {
            constexpr int nt = decltype(jc)::value + 1;
            wait_frag<pend + 2 * (NT - 1 - nt)>(wl[nt], wh[nt]);
            wf[nt] = cat8(wl[nt], wh[nt]);
            acc[nt][0] = mm_mfma(wf[nt], af, acc[nt][0]);
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          wait_frag<pend>(al[mt % 3], ah[mt % 3]);
          const i32x8 af = cat8(al[mt % 3], ah[mt % 3]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mm_mfma(wf[nt], af, acc[nt][mt]);
        }
      });
      continue;
    }
    i32x4 wl[2][NT], wh[2][NT];   // weight fragments of block kb in [kb & 1]
    i32x4 al[2], ah[2];           // activation fragment of step s in [s & 1]
    i32x8 wf[NT];
    block_start<0>(wa0, wa1, aa0, aa1, wl[0], wh[0], al[0], ah[0]);
    // The refill is issued behind the first fragment reads (they are what the MFMAs wait for).
    // (-DVA_MM_LATE_FILL, measurement builds: the second half of an 8-wave workgroup refills
    // behind its MFMAs instead, so that one wave of every SIMD computes while the other feeds the
    // DMA queue -- no gain, profiles/r02_gemm_ablation.txt.)
    const int fill = nxt;
    const bool late_fill = LATE_FILL && wave >= NW / 2;
    const bool have_next = g + NST - 1 < ng;
#if VA_MM_ABLATE != 1
    if (!SPREAD && !ASYM && !late_fill && have_next) group_load(fill, g + NST - 1);
#endif
    nxt = cur;
    cur = cur + 1 == NST ? 0 : cur + 1;
#ifdef VA_PROBES
    unsigned long long ph_t2 = 0;
#endif
    static_for<KB * MT>([&](auto ic) {
      constexpr int s = decltype(ic)::value;
      constexpr int kb = s / MT, mt = s % MT;
      constexpr int s1 = s + 1, kb1 = s1 / MT, mt1 = s1 % MT;
      constexpr bool more = s1 < KB * MT;
      constexpr bool next_block = more && mt1 == 0;   // only at mt == MT-1 >= 1
      if constexpr (next_block)
        block_start<kb1 * STAGE>(wa0, wa1, aa0, aa1, wl[kb1 & 1], wh[kb1 & 1], al[s1 & 1], ah[s1 & 1]);
      else if constexpr (more)
        lds_read_frag<kb1 * STAGE, mt1>(aa0, aa1, al[s1 & 1], ah[s1 & 1]);
      constexpr int later = next_block ? 2 + 2 * WA : (more ? 2 : 0);
      if constexpr (mt == 0) {
        // reads of this block in issue order: W0 A0 W1 .. W(WA-1), the `later` reads of the next
        // step, then W(WA) .. W(NT-1), one in front of each MFMA
        if constexpr (WA < NT) lds_read_frag<kb * STAGE, WA>(wa0, wa1, wl[kb & 1][WA], wh[kb & 1][WA]);
        wait_wa<later + 2 * w_younger<NT>(0)>(wl[kb & 1][0], wh[kb & 1][0], al[s & 1], ah[s & 1]);
#ifdef VA_PROBES
        if constexpr (s == 0) ph_t2 = __builtin_amdgcn_s_memtime();
#endif
        const i32x8 af = cat8(al[s & 1], ah[s & 1]);
        wf[0] = cat8(wl[kb & 1][0], wh[kb & 1][0]);
        acc[0][mt] = mm_mfma(wf[0], af, acc[0][mt]);
        // keep each MFMA above the next wait (the scheduler would sink all of them below
        // the last one and the early start would be lost)
        __builtin_amdgcn_sched_barrier(0);
        static_for<NT - 1>([&](auto jc) {
          constexpr int nt = decltype(jc)::value + 1;
          if constexpr (nt + WA < NT)
            lds_read_frag<kb * STAGE, nt + WA>(wa0, wa1, wl[kb & 1][nt + WA], wh[kb & 1][nt + WA]);
          // W(nt) is older than the next step's reads only while it belongs to the up-front ones
          wait_frag<(nt < WA ? later : 0) + 2 * w_younger<NT>(nt)>(wl[kb & 1][nt], wh[kb & 1][nt]);
          wf[nt] = cat8(wl[kb & 1][nt], wh[kb & 1][nt]);
          acc[nt][mt] = mm_mfma(wf[nt], af, acc[nt][mt]);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        wait_frag<later>(al[s & 1], ah[s & 1]);
        const i32x8 af = cat8(al[s & 1], ah[s & 1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt][mt] = mm_mfma(wf[nt], af, acc[nt][mt]);
      }
#if VA_MM_ABLATE != 1
      if constexpr (SPREAD && s < LOADS) {
        __builtin_amdgcn_sched_barrier(0);
        if (have_next) stage_load_one(ic, fill, (g + NST - 1) * LBK);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (ASYM) {
        __builtin_amdgcn_sched_barrier(0);
        if (have_next && wave >= 4) asym_load(ic, fill, (g + NST - 1) * LBK);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    });
#if VA_MM_ABLATE != 1
    if constexpr (LATE_FILL) {
      __builtin_amdgcn_sched_barrier(0);   // keep the refill behind the MFMAs
      if (late_fill && g + NST - 1 < ng) group_load(fill, g + NST - 1);
    }
#endif
#ifdef VA_PROBES
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long ph_t3 = __builtin_amdgcn_s_memtime();
      ph_wait += ph_t1 - ph_t0; ph_first += ph_t2 - ph_t1; ph_rest += ph_t3 - ph_t2;
    }
#endif
  }
#ifdef VA_PROBES
  if (!R5 && lane == 0 && blockIdx.x < 512 && wave < 8) {
    unsigned long long* d = g_mm_phase + ((size_t)blockIdx.x * 8 + wave) * 3;
    d[0] = ph_wait; d[1] = ph_first; d[2] = ph_rest;
  }
#endif
#ifdef VA_PROBES
  {
    const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x < 4096) {
      g_mm_stamps[blockIdx.x * 4 + 0] = st_c0; g_mm_stamps[blockIdx.x * 4 + 1] = st_r0;
      g_mm_stamps[blockIdx.x * 4 + 2] = st_c1; g_mm_stamps[blockIdx.x * 4 + 3] = st_r1;
    }
  }
#endif

  if constexpr (SPLITK) {
    // Split-K fix-up without a second kernel: every slice parks its fp32 accumulators in
    // the workspace (one float per lane and store, consecutive lanes consecutive), then
    // takes a ticket from the tile's counter; the workgroup that draws the last ticket adds
    // the S partials IN SLICE ORDER (bit-identical whoever comes last), resets the counter for
    // the next launch and runs the ordinary epilogue.  Nobody waits for anybody.
    // The partials travel as device-scope relaxed atomics (sc1 stores / loads: coherent
    // across the 8 XCD L2s by themselves).  A __threadfence() here instead writes back and
    // invalidates the whole L2 per wave and made the kernel 10-50x slower.
    constexpr int NTHR = 64 * NW;
    constexpr int NACC = NT * MT * 4;   // floats per lane
    float* part = ws + (size_t)wg * S * (NACC * NTHR) + tid;
    {
      float* mine = part + (size_t)z * (NACC * NTHR);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(mine + ((nt * MT + mt) * 4 + r) * NTHR, acc[nt][mt][r],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own partial has been written through
    __syncthreads();   // everybody's has (and every wave is past its last read of `lds`)
    volatile int* flag = reinterpret_cast<volatile int*>(lds);
    if (tid == 0) {
      const unsigned old =
          __hip_atomic_fetch_add(ctr + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = old == (unsigned)(S - 1);
    }
    __syncthreads();
    if (*flag == 0) return;
    if (tid == 0) __hip_atomic_store(ctr + wg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* p = part + ((nt * MT + mt) * 4 + r) * NTHR;
          float v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int zz = 1; zz < S; ++zz)
            v += __hip_atomic_load(p + (size_t)zz * (NACC * NTHR), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          acc[nt][mt][r] = v;
        }
  }

  // ---- epilogue: lane holds C^T[n = 16 nt + 4 g4 + r][m = 16 mt + c16] ---------------
  // column scales / bias: loaded once, unconditionally, from clamped addresses
  float sbv[NT][4], biasv[NT][4];
  const bool has_bias = bias != nullptr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * (BN / WN) + nt * 16 + 4 * g4 + r;
      const int nn = n < N ? n : N - 1;
      sbv[nt][r] = b_scales[b_scale_numel == 1 ? 0 : nn];
      biasv[nt][r] = 0.f;
    }
  if (has_bias) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * (BN / WN) + nt * 16 + 4 * g4 + r;
        biasv[nt][r] = to_f32(bias[n < N ? n : N - 1]);
      }
  }
  float sav[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (BM / WM) + mt * 16 + c16;
    sav[mt] = a_scales[a_scale_numel == 1 ? 0 : (m < M ? m : M - 1)];
  }
  if constexpr (VEC_STORE) {
    // tile through LDS as [BM rows][BN * 2 B], 16-byte chunks XOR-swizzled by the row;
    // then each store instruction writes 512 / BN whole rows
    static_assert(sizeof(out_t) == 2, "");
    constexpr int ROWB = BN * 2;                 // bytes per tile row
    constexpr int CPR = BN / 8;                  // 16-byte chunks per row
    constexpr int SWZ = (CPR < 8 ? CPR : 8) - 1; // chunk ^= row & SWZ stays inside the row
    constexpr int RPI = 64 / CPR;                // rows per store instruction
    static_assert(BM % (RPI * NW) == 0, "");
    static_assert(BM * ROWB <= LDS_TOTAL, "");
    __syncthreads();   // every wave is past its last fragment read
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wm * (BM / WM) + mt * 16 + c16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        struct alignas(8) V4 { out_t e[4]; } pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[nt][mt][r] * sav[mt];
          x = x * sbv[nt][r];
          out_t o = from_f32<out_t>(x);
          if (has_bias) o = from_f32<out_t>(to_f32(o) + biasv[nt][r]);
          pk.e[r] = o;
        }
        const int c8 = wn * (BN / WN / 4) + nt * 4 + g4;   // 8-byte chunk within the row
        *reinterpret_cast<V4*>(lds + row * ROWB + ((c8 ^ (2 * (row & SWZ))) << 3)) = pk;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM / (RPI * NW); ++i) {
      const int row = RPI * (wave * (BM / (RPI * NW)) + i) + lane / CPR;
      const int ch = lane % CPR;                 // 16-byte chunk = 8 outputs
      const uint4 v = *reinterpret_cast<const uint4*>(lds + row * ROWB + ((ch ^ (row & SWZ)) << 4));
      const int m = m0 + row, n = n0 + 8 * ch;
      if (m < M && n < N) {
        // an output beyond ~half of the 256 MiB Infinity Cache would push the operand panels the
        // other tiles still need out of it: streamed past it instead (4096 x 28672: 415 -> 395 us;
        // smaller outputs measured 0 .. 3 % slower that way)
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        u32x4* dst = reinterpret_cast<u32x4*>(out + (int64_t)m * ldc + n);
        if (nt_store) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, dst);
        else *dst = (u32x4){v.x, v.y, v.z, v.w};
      }
    }
#ifdef VA_PROBES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have left
    if (tid == 0 && blockIdx.x < 4096) g_mm_stamps_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (BM / WM) + mt * 16 + c16;
    if (m >= M) continue;
    const float sa = sav[mt];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * (BN / WN) + nt * 16 + 4 * g4;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= N) break;
        float x = acc[nt][mt][r] * sa;
        x = x * sbv[nt][r];
        out_t o = from_f32<out_t>(x);
        if (has_bias) o = from_f32<out_t>(to_f32(o) + biasv[nt][r]);
        out[(int64_t)m * ldc + n + r] = o;
      }
    }
  }
}

struct GldsArgs {
  void* out; const void* a; const void* b; const float* a_scales; const float* b_scales;
  const void* bias; int M, N, K; int64_t lda, ldb, ldc; int asn, bsn; bool vec; hipStream_t st;
  int splits;
  int ws_mode;   // 0 = registered workspace (or unsplit), 1 = the static scratch below (or an error)
};

// Split-K workspace: one stream slot of the caller-registered workspace (common.cuh,
// vllm_amd_set_workspace); no slot (none registered, stream capture, a fifth stream) or a launch
// that needs more than the slot holds -> unsplit.  Counters reset themselves (kernel above).
static std::mutex g_split_mu;   // guards g_mm_scratch_ptr below

static bool split_workspace(hipStream_t st, size_t bytes, int tiles, float** ws, unsigned** ctr) {
  WsSlot slot;
  if (tiles > (int)WS_CTR_MM_WORDS || !workspace_slot(st, &slot) || bytes > slot.data_bytes) return false;
  *ctr = slot.ctr_mm;
  *ws = (float*)slot.data;
  return true;
}

// Scratch of vllm_amd_scaled_mm_static (the entry point for callers that cannot allocate or
// register anything): one array per GPU the library runs on, held until the process ends.  In
// host code the array's NAME is only a stand-in: the device address comes from
// hipGetSymbolAddress, once per device, and that pointer is what kernels and memsets get.
constexpr size_t STATIC_PART_BYTES = (size_t)8 << 20;
constexpr int STATIC_CTR_TILES = 4096;
static __device__ __align__(256) unsigned char g_mm_scratch[STATIC_PART_BYTES + STATIC_CTR_TILES * sizeof(unsigned)];
static unsigned char* g_mm_scratch_ptr[64];

static int static_workspace(hipStream_t st, size_t bytes, int tiles, float** ws, unsigned** ctr) {
  VA_REQUIRE(bytes <= STATIC_PART_BYTES && tiles <= STATIC_CTR_TILES, VLLM_AMD_EUNSUPPORTED,
             "scaled_mm_static: this split needs %zu bytes of partials for %d tiles; the static "
             "scratch holds %zu bytes and %d tiles (use fewer K slices, or vllm_amd_scaled_mm with "
             "a registered workspace)", bytes, tiles, STATIC_PART_BYTES, STATIC_CTR_TILES);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "scaled_mm_static");
  VA_REQUIRE(dev >= 0 && dev < 64, VLLM_AMD_EUNSUPPORTED, "scaled_mm_static: device %d", dev);
  unsigned char* base;
  {
    std::lock_guard<std::mutex> lk(g_split_mu);
    if (!g_mm_scratch_ptr[dev]) {
      void* p = nullptr;
      e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_mm_scratch));
      if (e != hipSuccess || !p) return hip_fail(e, "scaled_mm_static: hipGetSymbolAddress");
      // Tickets start at zero ONCE per device (a __device__ array is zero when the code object loads; this
      // memset only matters if an earlier process state left them otherwise).  Every split launch leaves its
      // tickets at zero again (the last arriver resets them, kernel above), so no per-call memset -- it cost
      // about what split-K saves on the shapes that split.
      e = hipMemsetAsync((unsigned char*)p + STATIC_PART_BYTES, 0, STATIC_CTR_TILES * sizeof(unsigned), st);
      if (e != hipSuccess) return hip_fail(e, "scaled_mm_static: hipMemsetAsync");
      g_mm_scratch_ptr[dev] = (unsigned char*)p;
    }
    base = g_mm_scratch_ptr[dev];
  }
  *ws = (float*)base;
  *ctr = (unsigned*)(base + STATIC_PART_BYTES);
  return VLLM_AMD_OK;
}

#define GLDS_LAUNCH(VEC, SPLIT, GRID, WS, CTR, S)                                                  \
  hipLaunchKernelGGL((scaled_mm_glds_kernel<out_t, BM, BN, WM, WN, NST, KB, VEC, SPLIT, MODE>),    \
                     dim3(GRID), dim3(64 * WM * WN), 0, g.st, (out_t*)g.out, (const uint8_t*)g.a,  \
                     (const uint8_t*)g.b, g.a_scales, g.b_scales, (const out_t*)g.bias, g.M, g.N,  \
                     g.K, g.lda, g.ldb, g.ldc, g.asn, g.bsn, WS, CTR, S)

template <typename out_t, int BM, int BN, int WM, int WN, int NST, int KB, bool CAN_SPLIT, int MODE = 0>
static int launch_glds_t(const GldsArgs& g) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if constexpr (CAN_SPLIT) {
    int S = g.splits;
    const int ng = g.K / (LBK * KB);
    if (S > ng) S = ng;
    float* ws = nullptr;
    unsigned* ctr = nullptr;
    if (S > 1 && g.vec) {
      const size_t need = (size_t)tiles * S * BM * BN * sizeof(float);
      if (g.ws_mode == 1) {
        const int rc = static_workspace(g.st, need, tiles, &ws, &ctr);
        if (rc != VLLM_AMD_OK) return rc;
        GLDS_LAUNCH(true, true, tiles * S, ws, ctr, S);
        return VLLM_AMD_OK;
      }
      if (split_workspace(g.st, need, tiles, &ws, &ctr)) {
        GLDS_LAUNCH(true, true, tiles * S, ws, ctr, S);
        return VLLM_AMD_OK;
      }
    }
  }
  if (g.vec) GLDS_LAUNCH(true, false, tiles, nullptr, nullptr, 1);
  else GLDS_LAUNCH(false, false, tiles, nullptr, nullptr, 1);
  return VLLM_AMD_OK;
}
#undef GLDS_LAUNCH

template <typename out_t>
static int launch_glds_shape(int shape, const GldsArgs& g) {
  switch (shape) {
    case 2: return launch_glds_t<out_t, 256, 256, 2, 4, 2, 1, false>(g);
    case 3: return launch_glds_t<out_t, 128, 128, 2, 4, 2, 1, true>(g);
    case 4: return launch_glds_t<out_t, 128, 128, 2, 4, 2, 2, true>(g);
    case 5: return launch_glds_t<out_t, 128, 64, 4, 2, 3, 1, true>(g);
    case 6: return launch_glds_t<out_t, 128, 64, 4, 2, 3, 2, true>(g);
    case 7: return launch_glds_t<out_t, 64, 64, 2, 2, 4, 1, true>(g);
    case 8: return launch_glds_t<out_t, 64, 64, 2, 2, 2, 4, true>(g);
    case 9: return launch_glds_t<out_t, 64, 32, 2, 2, 4, 1, true>(g);
    case 10: return launch_glds_t<out_t, 64, 32, 2, 2, 3, 4, true>(g);
    case 12: return launch_glds_t<out_t, 32, 128, 1, 4, 4, 1, true>(g);
    case 14: return launch_glds_t<out_t, 256, 256, 2, 2, 2, 1, false>(g);   // 4 waves, 128 x 128 each
    case 15: return launch_glds_t<out_t, 256, 256, 2, 4, 5, 1, false>(g);   // ring of five half stages
    case 16: return launch_glds_t<out_t, 256, 256, 2, 4, 2, 1, false, 1>(g);   // waves 4-7 stage everything
    case 17: return launch_glds_t<out_t, 256, 256, 4, 4, 2, 1, false>(g);      // 16 waves, 64 x 64 each
    case 26: return launch_glds_t<out_t, 64, 64, 2, 2, 8, 1, true>(g);         // deeper rings: weights from HBM
    case 27: return launch_glds_t<out_t, 64, 64, 2, 2, 4, 2, true>(g);
    case 28: return launch_glds_t<out_t, 128, 64, 4, 2, 6, 1, true>(g);
    default: return launch_glds_t<out_t, 32, 128, 1, 4, 3, 2, true>(g);   // 13
  }
}

// The kernels address their operands with 32-bit byte offsets.
bool scaled_mm_large_fits(int M, int N, int K, int64_t lda, int64_t ldb) {
  const int64_t lim = (int64_t)1 << 32;
  return (int64_t)M * lda + K < lim && (int64_t)N * ldb + K < lim;
}

// Caller guarantees K % 128 == 0, 16-byte aligned operands and scaled_mm_large_fits();
// vec = 16-byte aligned output rows (ldc % 8 == 0, N % 8 == 0, out 16-byte aligned).
int launch_scaled_mm_large(int shape, void* out, const void* a, const void* b,
                           const float* a_scales, const float* b_scales, const void* bias, int M,
                           int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int asn, int bsn,
                           int out_dtype, bool vec, int splits, int ws_mode, hipStream_t st) {
  // shape = a GldsTile::id of the table in scaled_mm.hip.  Shapes that consume KB blocks per
  // barrier need K % (128 KB) == 0; otherwise the same tile with KB = 1.
  static const int kb_of[29] = {1, 1, 1, 1, 2, 1, 2, 1, 4, 1, 4, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1};
  if (shape < 2 || shape > 28 || (shape > 17 && shape < 26) || shape == 11) shape = 3;
  if (K % (LBK * kb_of[shape]) != 0) shape -= 1;
  const GldsArgs g{out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc, asn, bsn, vec, st,
                   splits, ws_mode};
  if (out_dtype == VLLM_AMD_BF16) return launch_glds_shape<bf16_t>(shape, g);
  return launch_glds_shape<f16_t>(shape, g);
}

}  // namespace va

#ifdef VA_PROBES
extern "C" int vllm_amd_probe_mm_stamps(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::g_mm_stamps), (size_t)nwg * 4 * 8) == hipSuccess ? 0 : -3;
}
extern "C" int vllm_amd_probe_mm_phase(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::g_mm_phase), (size_t)nwg * 8 * 3 * 8) == hipSuccess ? 0 : -3;
}
extern "C" int vllm_amd_probe_mm_stamps_end(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::g_mm_stamps_end), (size_t)nwg * 8) == hipSuccess ? 0 : -3;
}
#endif
