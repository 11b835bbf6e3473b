// Large-M fp8 scaled GEMM for gfx950: BM x 256 x 128B tiles, 8 waves, direct-to-LDS
// staging (global_load_lds, 16 B per lane) with the NEXT K block in flight under the
// MFMAs of the current one.
//
//   out[M,N] = ((sa (.) A[M,K]) . (sb (.) W[N,K]^T)).to(out_t) (+ bias)      K % 128 == 0
//
// Structure (cdna_hip_programming.md section 5, "glds, 2 LDS buffers"):
//   * tile BM(M) x 256(N), K block = 128 bytes = one v_mfma_scale_f32_16x16x128_f8f6f4;
//     8 waves as 2(M) x 4(N); a wave owns (BM/2) x 64 outputs = (BM/32) x 4 MFMA tiles;
//   * the product is computed transposed (A operand = weight rows, B operand = activation
//     rows) so a lane holds 4 consecutive n of one output row;
//   * LDS image per operand and stage: [rows][128 B], 16-byte chunks XOR-swizzled by
//     f(row) (same f as scaled_mm.hip: conflict-free ds_read_b128).  global_load_lds writes
//     lane-linearly (wave base + 16*lane), so the swizzle is applied to the per-lane SOURCE
//     address and to the read address -- never to the destination;
//   * ONE barrier per K block:   wait own DMA of stage cur -> s_barrier -> issue DMA of
//     block k+1 into stage cur^1 (everyone is past its reads of that stage) -> fragment
//     reads + MFMAs of stage cur.  The DMA has a whole MFMA phase to land;
//   * fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt: with C++ LDS
//     loads hipcc orders every ds_read behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)
//     right after the barrier), which serialises the prefetch with the math.  The waits
//     carry the fragment registers as in/out operands, so the MFMAs depend on them.
#include "common.cuh"

namespace va {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int LBN = 256, LBK = 128, L_THREADS = 512;

__device__ __forceinline__ int lswz(int row) {
  return ((row >> 1) & 1) | (((row >> 2) & 1) << 2);
}

template <int OFF>
__device__ __forceinline__ i32x4 lds_read16(uint32_t addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ i32x8 cat8(i32x4 lo, i32x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename out_t, int BM, bool VEC_STORE>
__global__ __launch_bounds__(L_THREADS) void scaled_mm_large_kernel(
    out_t* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, const float* __restrict__ b_scales,
    const out_t* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
    int a_scale_numel, int b_scale_numel) {
  constexpr int A_BYTES = BM * LBK, W_BYTES = LBN * LBK, STAGE = A_BYTES + W_BYTES;
  constexpr int MT = BM / 32;        // m tiles (16 rows) per wave
  constexpr int A_ROUNDS = BM / 64;  // glds rounds: 8 waves x 8 rows per round
  constexpr int W_ROUNDS = LBN / 64;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + LBN - 1) / LBN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = wg % tiles_m, tn = wg / tiles_m;
  const int m0 = tm * BM, n0 = tn * LBN;

  // staging: in round r, wave w fills rows [64 r + 8 w, +8): lane -> (row = lane>>3, slot = lane&7)
  // and loads logical chunk slot ^ f(row) so that the linear LDS image is the swizzled tile.
  // Sources are 32-bit offsets from the (uniform) operand base; the host checks they fit.
  const int srow = lane >> 3, sslot = lane & 7;
  uint32_t a_voff[A_ROUNDS], w_voff[W_ROUNDS];
#pragma unroll
  for (int r = 0; r < A_ROUNDS; ++r) {
    const int row = 64 * r + 8 * wave + srow;
    int m = m0 + row;
    m = m < M ? m : M - 1;   // rows past M: any valid row (their outputs are never stored)
    a_voff[r] = (uint32_t)m * (uint32_t)lda + ((sslot ^ lswz(row)) << 4);
  }
#pragma unroll
  for (int r = 0; r < W_ROUNDS; ++r) {
    const int row = 64 * r + 8 * wave + srow;
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
          (__attribute__((address_space(3))) void*)(base + 64 * r * LBK), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(bg + w_voff[r]),
          (__attribute__((address_space(3))) void*)(base + A_BYTES + 64 * r * LBK), 16, 0, 0);
  };

  f32x4 acc[4][MT];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: the swizzle depends on c16 only, so tile t is an immediate
  // offset t * 16 * LBK from one per-lane base per 16-byte half (j) and operand
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  uint32_t w_addr[2], a_addr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = ((2 * g4 + j) ^ lswz(c16)) << 4;
    w_addr[j] = lds_base + A_BYTES + (wn * 64 + c16) * LBK + ch;
    a_addr[j] = lds_base + (wm * (BM / 2) + c16) * LBK + ch;
  }
  constexpr int TOFF = 16 * LBK;

  const int nk = K / LBK;
  stage_load(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // own DMA of this stage (issued one block ago) has landed; then everybody's has, and
    // everybody is done reading the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage_load((kt + 1) & 1, (kt + 1) * LBK);

    const uint32_t so = (kt & 1) * STAGE;
    const uint32_t wa0 = w_addr[0] + so, wa1 = w_addr[1] + so;
    const uint32_t aa0 = a_addr[0] + so, aa1 = a_addr[1] + so;
    i32x4 w0 = lds_read16<0 * TOFF>(wa0), w1 = lds_read16<0 * TOFF>(wa1);
    i32x4 w2 = lds_read16<1 * TOFF>(wa0), w3 = lds_read16<1 * TOFF>(wa1);
    i32x4 w4 = lds_read16<2 * TOFF>(wa0), w5 = lds_read16<2 * TOFF>(wa1);
    i32x4 w6 = lds_read16<3 * TOFF>(wa0), w7 = lds_read16<3 * TOFF>(wa1);
    i32x4 a0 = lds_read16<0>(aa0), a1 = lds_read16<0>(aa1);
    i32x8 wf[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      i32x4 na0 = a0, na1 = a1;
      if (mt + 1 < MT) {
        // next A fragment: constant offsets need a compile-time mt -> unrolled switch
        switch (mt + 1) {
#define VA_CASE(T) case T: na0 = lds_read16<(T < MT ? T : 0) * TOFF>(aa0); \
                           na1 = lds_read16<(T < MT ? T : 0) * TOFF>(aa1); break;
          VA_CASE(1) VA_CASE(2) VA_CASE(3) VA_CASE(4) VA_CASE(5) VA_CASE(6) VA_CASE(7)
#undef VA_CASE
        }
      }
      if (mt == 0) {
        if (MT > 1)
          asm volatile("s_waitcnt lgkmcnt(2)"
                       : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6),
                         "+v"(w7), "+v"(a0), "+v"(a1));
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6),
                         "+v"(w7), "+v"(a0), "+v"(a1));
        wf[0] = cat8(w0, w1); wf[1] = cat8(w2, w3); wf[2] = cat8(w4, w5); wf[3] = cat8(w6, w7);
      } else if (mt + 1 < MT) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0), "+v"(a1));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1));
      }
      const i32x8 af = cat8(a0, a1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
            wf[nt], af, acc[nt][mt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      a0 = na0; a1 = na1;
    }
  }

  // ---- epilogue: lane holds C^T[n = 16 nt + 4 g4 + r][m = 16 mt + c16] ---------------
  // column scales / bias: loaded once, unconditionally, from clamped addresses
  float sbv[4][4], biasv[4][4];
  const bool has_bias = bias != nullptr;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * g4 + r;
      const int nn = n < N ? n : N - 1;
      sbv[nt][r] = b_scales[b_scale_numel == 1 ? 0 : nn];
      biasv[nt][r] = 0.f;
    }
  if (has_bias) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * g4 + r;
        biasv[nt][r] = to_f32(bias[n < N ? n : N - 1]);
      }
  }
  float sav[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (BM / 2) + mt * 16 + c16;
    sav[mt] = a_scales[a_scale_numel == 1 ? 0 : (m < M ? m : M - 1)];
  }
  if constexpr (VEC_STORE) {
    // Coalesced store: the scaled tile goes through LDS (it overlays the staging buffers)
    // as [BM rows][512 B], 16-byte chunks XOR-swizzled by (row & 7); then every store
    // instruction writes two full 512-byte rows.  Direct stores from the MFMA layout write
    // 32-byte pieces of 16 different rows per instruction and cost 25 us on a 32 MB output.
    static_assert(sizeof(out_t) == 2, "");
    __syncthreads();   // every wave is past its last fragment read
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wm * (BM / 2) + mt * 16 + c16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        struct alignas(8) V4 { out_t e[4]; } pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[nt][mt][r] * sav[mt];
          x = x * sbv[nt][r];
          out_t o = from_f32<out_t>(x);
          if (has_bias) o = from_f32<out_t>(to_f32(o) + biasv[nt][r]);
          pk.e[r] = o;
        }
        const int c8 = wn * 16 + nt * 4 + g4;   // 8-byte chunk within the row
        *reinterpret_cast<V4*>(lds + row * 512 + ((c8 ^ (2 * (row & 7))) << 3)) = pk;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int row = 2 * (wave * (BM / 16) + i) + (lane >> 5);
      const int c16b = lane & 31;               // 16-byte chunk = 8 outputs
      const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 512 + ((c16b ^ (row & 7)) << 4));
      const int m = m0 + row, n = n0 + 8 * c16b;
      if (m < M && n < N) *reinterpret_cast<uint4*>(out + (int64_t)m * ldc + n) = v;
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (BM / 2) + mt * 16 + c16;
    if (m >= M) continue;
    const float sa = sav[mt];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * g4;
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

template <typename out_t, int BM>
static void launch_large_t(void* out, const void* a, const void* b, const float* a_scales,
                           const float* b_scales, const void* bias, int M, int N, int K,
                           int64_t lda, int64_t ldb, int64_t ldc, int asn, int bsn, bool vec,
                           hipStream_t st) {
  const int tiles = ((M + BM - 1) / BM) * ((N + LBN - 1) / LBN);
  if (vec)
    hipLaunchKernelGGL((scaled_mm_large_kernel<out_t, BM, true>), dim3(tiles), dim3(L_THREADS), 0,
                       st, (out_t*)out, (const uint8_t*)a, (const uint8_t*)b, a_scales, b_scales,
                       (const out_t*)bias, M, N, K, lda, ldb, ldc, asn, bsn);
  else
    hipLaunchKernelGGL((scaled_mm_large_kernel<out_t, BM, false>), dim3(tiles), dim3(L_THREADS), 0,
                       st, (out_t*)out, (const uint8_t*)a, (const uint8_t*)b, a_scales, b_scales,
                       (const out_t*)bias, M, N, K, lda, ldb, ldc, asn, bsn);
}

// The kernel addresses its operands with 32-bit byte offsets.
bool scaled_mm_large_fits(int M, int N, int K, int64_t lda, int64_t ldb) {
  const int64_t lim = (int64_t)1 << 32;
  return (int64_t)M * lda + K < lim && (int64_t)N * ldb + K < lim;
}

// bm = 256 or 128; caller guarantees K % 128 == 0, 16-byte aligned operands and
// scaled_mm_large_fits().
void launch_scaled_mm_large(int bm, void* out, const void* a, const void* b,
                            const float* a_scales, const float* b_scales, const void* bias, int M,
                            int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int asn, int bsn,
                            int out_dtype, bool vec, hipStream_t st) {
  if (out_dtype == VLLM_AMD_BF16) {
    if (bm == 256) launch_large_t<bf16_t, 256>(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc, asn, bsn, vec, st);
    else launch_large_t<bf16_t, 128>(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc, asn, bsn, vec, st);
  } else {
    if (bm == 256) launch_large_t<f16_t, 256>(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc, asn, bsn, vec, st);
    else launch_large_t<f16_t, 128>(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc, asn, bsn, vec, st);
  }
}

}  // namespace va
