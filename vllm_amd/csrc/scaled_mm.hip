// w8a8 fp8 (OCP e4m3fn) scaled GEMM for gfx950:
//   out[M,N] = ((a_scales (.) A[M,K]) . (b_scales (.) B[K,N])).to(out_dtype) (+ bias)
// A row-major, B column-major (= the [N,K] weight matrix): both operands are
// K-contiguous, which is the natural MFMA operand order.
//
// Matrix core path: v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 block
// scales -- on CDNA4 the block-scaled form is the only fp8 MFMA that runs at
// twice the bf16 rate (MI355X_MICROARCH.md, Matrix cores); products of e4m3
// values are exact in fp32 and accumulation is fp32, as in the reference
// (triton_scaled_mm.py:87-110).  Per-token / per-channel scales, the cast and
// the bias are fused in the epilogue.
//
// Tile: 128(M) x 128(N) x 128(K bytes) per 256-thread workgroup, 4 waves as
// 2(M) x 2(N), each wave 64 x 64 = 4 x 4 MFMA tiles.  The MFMA computes the
// TRANSPOSED tile (A operand = weight rows, B operand = activation rows) so that
// a lane ends up with 4 consecutive n of one output row: 8-byte stores.
// LDS: [128 rows][128 B] per operand and buffer, 16-byte chunks XOR-swizzled by
// f(row) so both the ds_write_b128 of staging and the ds_read_b128 of fragment
// loads are bank-conflict free; two buffers, one barrier per K tile.
#include "common.cuh"

namespace va {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int BM = 128, BN = 128, BK = 128;
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int swz(int row) {
  return ((row >> 1) & 1) | (((row >> 2) & 1) << 2);
}
// byte offset of 16-byte chunk `chunk` (0..7) of row `row` in a [rows][128 B] tile
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * BK + ((chunk ^ swz(row)) << 4);
}

template <typename out_t>
__device__ __forceinline__ out_t finish(float acc, float sa, float sb) {
  float v = acc * sa;
  v = v * sb;
  return from_f32<out_t>(v);
}

template <typename out_t, bool VEC_STORE>
__global__ __launch_bounds__(GEMM_THREADS) void scaled_mm_kernel(
    out_t* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, const float* __restrict__ b_scales,
    const out_t* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
    int a_scale_numel, int b_scale_numel) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][BM * BK];  // [buf][A|W][tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive workgroup ids (round-robin over the 8
  // XCDs) are remapped so that each XCD walks a contiguous run of tiles and
  // neighbouring tiles share their weight panel in that XCD's L2.
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // m fastest: the tiles of one weight panel (fixed n) are adjacent
  const int tm = wg % tiles_m, tn = wg / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging map: thread -> (row = p*32 + tid/8, chunk = tid%8), p = 0..3
  const int srow = tid >> 3, schunk = tid & 7;
  uint4 ra[4], rw[4];
  auto gload = [&](int k0) {
    const int k = k0 + schunk * 16;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 32 + srow;
      const int m = m0 + row, n = n0 + row;
      ra[p] = (m < M && k < K) ? *reinterpret_cast<const uint4*>(a + (int64_t)m * lda + k)
                               : make_uint4(0, 0, 0, 0);
      rw[p] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(b + (int64_t)n * ldb + k)
                               : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 32 + srow;
      *reinterpret_cast<uint4*>(&lds[buf][0][lds_off(row, schunk)]) = ra[p];
      *reinterpret_cast<uint4*>(&lds[buf][1][lds_off(row, schunk)]) = rw[p];
    }
  };

  f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);

    i32x8 wf[4], af[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rw_ = wn * 64 + t * 16 + c16;
      const int ra_ = wm * 64 + t * 16 + c16;
      uint4 w0 = *reinterpret_cast<const uint4*>(&lds[buf][1][lds_off(rw_, 2 * g4)]);
      uint4 w1 = *reinterpret_cast<const uint4*>(&lds[buf][1][lds_off(rw_, 2 * g4 + 1)]);
      uint4 a0 = *reinterpret_cast<const uint4*>(&lds[buf][0][lds_off(ra_, 2 * g4)]);
      uint4 a1 = *reinterpret_cast<const uint4*>(&lds[buf][0][lds_off(ra_, 2 * g4 + 1)]);
      wf[t] = (i32x8){(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w,
                      (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
      af[t] = (i32x8){(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w,
                      (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
            wf[nt], af[mt], acc[nt][mt], 0 /*A fmt: fp8 e4m3*/, 0 /*B fmt*/, 0, 0x7f7f7f7f, 0,
            0x7f7f7f7f);

    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C^T[n = 16 nt + 4 g4 + r][m = 16 mt + c16] ---------------
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + c16;
    if (m >= M) continue;
    const float sa = a_scales[a_scale_numel == 1 ? 0 : m];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * g4;
      if (n >= N) continue;
      out_t v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r < N ? n + r : N - 1;
        const float sb = b_scales[b_scale_numel == 1 ? 0 : nn];
        out_t o = finish<out_t>(acc[nt][mt][r], sa, sb);
        if (bias) o = from_f32<out_t>(to_f32(o) + to_f32(bias[nn]));
        v[r] = o;
      }
      out_t* op = out + (int64_t)m * ldc + n;
      if (VEC_STORE) {
        if (n + 3 < N) {
          struct alignas(8) V4 { out_t e[4]; } pk = {{v[0], v[1], v[2], v[3]}};
          *reinterpret_cast<V4*>(op) = pk;
          continue;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) op[r] = v[r];
    }
  }
}

}  // namespace va

using namespace va;

static int launch_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                            const float* b_scales, const void* bias, int M, int N, int K,
                            int64_t lda, int64_t ldb, int64_t ldc, int a_scale_numel,
                            int b_scale_numel, int out_dtype, hipStream_t st, const char* op) {
  VA_REQUIRE(M >= 0 && N >= 0 && K >= 0, VLLM_AMD_EINVAL, "%s: negative dimension", op);
  if (M == 0 || N == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && a && b && a_scales && b_scales, VLLM_AMD_EINVAL, "%s: null pointer", op);
  VA_REQUIRE(out_dtype == VLLM_AMD_BF16 || out_dtype == VLLM_AMD_F16, VLLM_AMD_EUNSUPPORTED,
             "%s: out dtype must be bfloat16 or float16", op);
  VA_REQUIRE(K % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0 && (((uintptr_t)a) & 15) == 0 &&
                 (((uintptr_t)b) & 15) == 0,
             VLLM_AMD_EUNSUPPORTED, "%s: K, lda, ldb and operand pointers must be multiples of 16",
             op);
  VA_REQUIRE((a_scale_numel == 1 || a_scale_numel == M) && (b_scale_numel == 1 || b_scale_numel == N),
             VLLM_AMD_EINVAL, "%s: scales must be per-tensor or per-row/per-column", op);
  const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
  const bool vec = (ldc % 4 == 0) && ((((uintptr_t)out) & 7) == 0);
#define MM(T, V)                                                                              \
  hipLaunchKernelGGL((scaled_mm_kernel<T, V>), dim3(tiles), dim3(GEMM_THREADS), 0, st, (T*)out, \
                     (const uint8_t*)a, (const uint8_t*)b, a_scales, b_scales, (const T*)bias, \
                     M, N, K, lda, ldb, ldc, a_scale_numel, b_scale_numel)
  if (out_dtype == VLLM_AMD_BF16) { if (vec) MM(bf16_t, true); else MM(bf16_t, false); }
  else { if (vec) MM(f16_t, true); else MM(f16_t, false); }
#undef MM
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                                  const float* b_scales, const void* bias, int M, int N, int K,
                                  int64_t lda, int64_t ldb, int64_t ldc, int a_scale_numel,
                                  int b_scale_numel, int out_dtype, void* stream) {
  return launch_scaled_mm(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc,
                          a_scale_numel, b_scale_numel, out_dtype, (hipStream_t)stream,
                          "scaled_mm");
}

// out_c[N_in, M_in] = (in_b[N_in,K] . in_a[M_in,K]^T) * scale_a * scale_b
// (csrc/rocm/skinny_gemms.cu:1649-1714): the activation rows are the GEMM's M,
// the weight rows its N.
extern "C" int vllm_amd_wvSplitKQ(const void* in_a, const void* in_b, void* out_c,
                                  const float* scale_a, const float* scale_b, int M_in, int K_in,
                                  int N_in, int out_dtype, int cu_count, void* stream) {
  (void)cu_count;
  VA_REQUIRE(N_in >= 0 && N_in <= 4, VLLM_AMD_EUNSUPPORTED, "wvSplitKQ: batch %d > 4", N_in);
  VA_REQUIRE(K_in % 16 == 0, VLLM_AMD_EUNSUPPORTED, "wvSplitKQ: K %d %% 16 != 0", K_in);
  return launch_scaled_mm(out_c, in_b, in_a, scale_a, scale_b, nullptr, N_in, M_in, K_in, K_in,
                          K_in, M_in, 1, 1, out_dtype, (hipStream_t)stream, "wvSplitKQ");
}
