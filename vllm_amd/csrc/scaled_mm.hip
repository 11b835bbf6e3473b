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


// ---------------------------------------------------------------------------
// Skinny GEMM, M <= 64 (decode): weight-bandwidth bound.  A workgroup of 8 waves
// owns 16 weight rows (output columns) and ALL M rows; the waves split K.  W is
// streamed once, straight into VGPRs in MFMA operand order (no LDS round trip:
// each byte is used by one wave only); the contraction index inside a 128-byte
// K block is permuted (lane group g takes bytes 64 j + 16 g, j = 0,1) so that one
// load instruction reads 16 rows x 64 contiguous bytes -- A is loaded with the
// same permutation, so the product is unchanged.  Partial sums of the 8 waves
// are combined through LDS, then scales / cast / bias as in the tile kernel.
// ---------------------------------------------------------------------------
constexpr int SK_WAVES = 8;

template <typename out_t, int MT>
__global__ __launch_bounds__(SK_WAVES * 64) void scaled_mm_skinny_kernel(
    out_t* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, const float* __restrict__ b_scales,
    const out_t* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
    int a_scale_numel, int b_scale_numel) {
  __shared__ float red[SK_WAVES][MT][16][17];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nkb = (K + 127) / 128;                       // 128-byte K blocks
  const int per = (nkb + SK_WAVES - 1) / SK_WAVES;
  const int kb0 = wave * per, kb1 = min(kb0 + per, nkb);

  const int n = n0 + c16;
  const bool n_ok = n < N;
  const uint8_t* wrow = b + (int64_t)(n_ok ? n : 0) * ldb + 16 * g4;
  const uint8_t* arow[MT];
  bool m_ok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = t * 16 + c16;
    m_ok[t] = m < M;
    arow[t] = a + (int64_t)(m_ok[t] ? m : 0) * lda + 16 * g4;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint4 z4 = make_uint4(0, 0, 0, 0);
  constexpr int UN = 4;  // K blocks in flight per wave
  for (int kb = kb0; kb < kb1; kb += UN) {
    uint4 wv[UN][2];
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = (kb + u) * 128 + 64 * j;            // + 16 g4 is in the row pointer
        const bool ok = n_ok && (kb + u) < kb1 && (k + 16 * g4) < K;
        const uint4 v = *reinterpret_cast<const uint4*>(wrow + (ok ? k : 0));   // unconditional
        wv[u][j] = ok ? v : z4;
      }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const i32x8 wf = (i32x8){(int)wv[u][0].x, (int)wv[u][0].y, (int)wv[u][0].z, (int)wv[u][0].w,
                               (int)wv[u][1].x, (int)wv[u][1].y, (int)wv[u][1].z, (int)wv[u][1].w};
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        uint4 av[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = (kb + u) * 128 + 64 * j;
          const bool ok = m_ok[t] && (kb + u) < kb1 && (k + 16 * g4) < K;
          const uint4 v = *reinterpret_cast<const uint4*>(arow[t] + (ok ? k : 0));
          av[j] = ok ? v : z4;
        }
        const i32x8 af = (i32x8){(int)av[0].x, (int)av[0].y, (int)av[0].z, (int)av[0].w,
                                 (int)av[1].x, (int)av[1].y, (int)av[1].z, (int)av[1].w};
        acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, af, acc[t], 0, 0, 0,
                                                                  0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  }
  // lane holds C^T[n = 4 g4 + r][m = 16 t + c16]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][t][c16][4 * g4 + r] = acc[t][r];   // [m][n]
  __syncthreads();
  for (int idx = threadIdx.x; idx < MT * 256; idx += SK_WAVES * 64) {
    const int t = idx >> 8, mm = (idx >> 4) & 15, nn = idx & 15;
    const int m = t * 16 + mm, no = n0 + nn;
    if (m >= M || no >= N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) v += red[w][t][mm][nn];
    out_t o = finish<out_t>(v, a_scales[a_scale_numel == 1 ? 0 : m],
                            b_scales[b_scale_numel == 1 ? 0 : no]);
    if (bias) o = from_f32<out_t>(to_f32(o) + to_f32(bias[no]));
    out[(int64_t)m * ldc + no] = o;
  }
}

// scaled_mm_large.hip
int launch_scaled_mm_large(int shape, void* out, const void* a, const void* b,
                           const float* a_scales, const float* b_scales, const void* bias, int M,
                           int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int asn, int bsn,
                           int out_dtype, bool vec, int splits, int ws_mode, hipStream_t st);
bool scaled_mm_large_fits(int M, int N, int K, int64_t lda, int64_t ldb);
// scaled_mm_pp.hip
int launch_scaled_mm_pp(int variant, void* out, const void* a, const void* b, const float* a_scales,
                        const float* b_scales, const void* bias, int M, int N, int K, int64_t lda,
                        int64_t ldb, int64_t ldc, int asn, int bsn, int out_dtype, hipStream_t st);
int g_mm_kernel = 0;   // tuning: 0 auto, 1 = register-staged 128x128, 2..10, 12, 13 = glds tile shapes
int g_mm_splitk = 0;   // tuning: 0 auto, n >= 1 = n K slices (1 = never split)
int g_mm_tail = 0;     // tuning key kept for old scripts; the tail split went away with the persistent kernel
}  // namespace va

using namespace va;

static int launch_scaled_mm(void* out, const void* a, const void* b, const float* a_scales,
                            const float* b_scales, const void* bias, int M, int N, int K,
                            int64_t lda, int64_t ldb, int64_t ldc, int a_scale_numel,
                            int b_scale_numel, int out_dtype, hipStream_t st, const char* op,
                            int ws_mode = 0, int req_splits = 0) {
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
  const bool glds_ok = K % 128 == 0 && K >= 256 && scaled_mm_large_fits(M, N, K, lda, ldb);
  // M <= 32: the weight-streaming kernel (one workgroup per 16 output columns) wins while
  // its grid is a single wave of workgroups and M fits one MFMA tile; beyond that the
  // 64x32 direct-to-LDS tile is flat in M and faster (profiles/r01_gemm_sweep_small.csv).
  // "mm_kernel" 11 forces it.
  const bool skinny = g_mm_kernel == 11 ||
                      (g_mm_kernel == 0 &&
                       (!glds_ok || (M <= 16 && ceil_div(N, 16) <= 256)));   // (round 3, HBM-served weights: 14336 x 4096 at M 16 18.9 vs 20.3 us)
  if (M <= 32 && skinny) {
    const int mt = ceil_div(M, 16);
#define SK(T, MT)                                                                              \
  hipLaunchKernelGGL((scaled_mm_skinny_kernel<T, MT>), dim3(ceil_div(N, 16)),                  \
                     dim3(SK_WAVES * 64), 0, st, (T*)out, (const uint8_t*)a, (const uint8_t*)b, \
                     a_scales, b_scales, (const T*)bias, M, N, K, lda, ldb, ldc, a_scale_numel, \
                     b_scale_numel)
#define SKT(T)                                       \
  do {                                               \
    if (mt == 1) SK(T, 1); else if (mt == 2) SK(T, 2); \
    else if (mt == 3) SK(T, 3); else SK(T, 4);       \
  } while (0)
    if (out_dtype == VLLM_AMD_BF16) SKT(bf16_t); else SKT(f16_t);
#undef SKT
#undef SK
    VA_LAUNCH_CHECK(op);
    return VLLM_AMD_OK;
  }
  const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
  const bool vec = (ldc % 4 == 0) && ((((uintptr_t)out) & 7) == 0);
  if (glds_ok) {
    int sel = g_mm_kernel;
    const int want_splits = req_splits > 0 ? req_splits : g_mm_splitk;   // 0 = heuristic
    int splits = want_splits > 0 ? want_splits : 1;
    if (sel >= 29 && sel <= 36) sel = -(sel - 29 + 8) - 100;   // 29 .. 36: the same variants with 256 x 192 tiles
    if ((sel >= 18 && sel <= 25) || sel <= -100) {
      // the staggered / persistent 256 x 256 kernel (scaled_mm_pp.hip), forced: 18 plain, 19 stagger,
      // 20 persistent, 21 both
      const bool vec16 = ldc % 8 == 0 && N % 8 == 0 && ((((uintptr_t)out) & 15) == 0);
      if (vec16) {
        launch_scaled_mm_pp(sel <= -100 ? -(sel + 100) : sel - 18, out, a, b, a_scales, b_scales, bias, M, N, K, lda,
                            ldb, ldc, a_scale_numel, b_scale_numel, out_dtype, st);
        VA_LAUNCH_CHECK(op);
        return VLLM_AMD_OK;
      }
      sel = 2;
    }
    if (sel == 0 || sel == 11 || sel > 36) {
      // Largest tile that still yields ~3/4 of a full wave of workgroups (256 CUs); rows of
      // a tile beyond M are wasted MFMA work, so BM <= 2 M.  With more than one wave of tiles the
      // variant that fits several workgroups per CU wins, below that the variant that takes
      // KB K-blocks per barrier (measured: tools/sweep_gemm.py, profiles/r01_gemm_sweep.csv).
      struct GldsTile { int id, id_kb, bm, bn, kb; };
      static const GldsTile tiles_tab[5] = {{2, 2, 256, 256, 1}, {3, 4, 128, 128, 2},
                                            {5, 6, 128, 64, 2},  {7, 8, 64, 64, 4},
                                            {9, 10, 64, 32, 4}};
      for (int i = 0; i < 5; ++i) {
        const GldsTile& t = tiles_tab[i];
        if (i < 4 && t.bm > 64 && M <= t.bm / 2) continue;
        const int nt = ceil_div(M, t.bm) * ceil_div(N, t.bn);
        if (nt >= 190 || i == 4) {
          // (the 64 x 64 tile with 4 K blocks per barrier loses on long K loops: 14336 x 4096 at
          // M 192 / 256 31.1 vs 25.3 / 27.5 us, 8192 x 4096 18.2 / 19.0 vs 16.0 / 16.6 us; at K 4096
          // it wins, 8.5 vs 9.1 us -- profiles/r02_gemm_sweep_mid.csv)
          // (more than one round of tiles: the variant that fits two workgroups per CU -- 4096 x 6144
          // at M 768 = 288 tiles of 128 x 128: 38.6 -> 30.2 us, at M 384 = 288 tiles of 128 x 64: 24.4 ->
          // 20.6 us)
          // (round 3, weights streamed from HBM as in a model -- bench.py rotates them now,
          // profiles/r03_gemm_sweep_rotating.csv: the 64 x 64 tile with one K block per barrier wins at
          // every K: 4096 x 4096 at M 256 12.7 vs 15.2 us, 4096 x 6144 at M 128 13.4 vs 15.9 us)
          sel = (nt > 256 || K % (128 * t.kb) != 0 || i == 3) ? t.id : t.id_kb;
          // Even the smallest tile leaves half the CUs idle and the K loop is long: two K
          // slices (in-kernel fix-up, scaled_mm_large.hip).  The fix-up is a chain of three
          // device-scope round trips (~2.5 us), so it pays only here: 14336x4096, M 16..64:
          // 17.2 -> 15.2 us; everywhere else on the grid splitting loses
          // (profiles/r01_gemm_sweep_splitk.csv; at K = 8192 it is 11.5 vs 11.1 us).
          if (i == 4 && nt <= 128 && K >= 12288 && want_splits == 0) splits = 2;
          break;
        }
      }
    }
    if (sel >= 2) {
      // 16-byte row-contiguous stores through LDS
      const bool vec16 = ldc % 8 == 0 && N % 8 == 0 && ((((uintptr_t)out) & 15) == 0);
      // MFMA-bound problems (the heuristic's 256 x 256 choice): the staggered, persistent kernel of
      // scaled_mm_pp.hip -- waves 4-7 half a K block behind waves 0-3, one workgroup per CU walking its
      // XCD's share of the tiles (4096^3 55.9 -> 49.3 us, K 14336 170.7 -> 154.0, N 28672 405 -> 334,
      // N 6144 100 (92.7 with the tail split below) -> 85.1: profiles/r03_gemm_pp.txt).  It writes 16-byte
      // pieces of rows, so outputs that do not allow them keep the kernel of scaled_mm_large.hip.
      if (sel == 2 && g_mm_kernel == 0 && splits == 1 && vec16) {
        // 256 x 192 tiles where they save a round of workgroups: a 192-wide tile costs ~7/8 of a 256-wide one (it
        // stages 56 instead of 64 KiB per K block), so N 6144 (the fused qkv projection) at M 3072 -- 288 tiles = 2
        // rounds against 384 narrow ones = 2 x 7/8 -- goes 90.2 -> 78.9 us, at M 4096 / 2048 91.5 / 52.3 -> 89.2 /
        // 50.7 us; everywhere else on the benchmark grid the wide tile has fewer or equal rounds (profiles/r03_gemm_pp.txt 13)
        const int tm = ceil_div(M, 256);
        const int r256 = ceil_div(tm * ceil_div(N, 256), 256), r192 = ceil_div(tm * ceil_div(N, 192), 256);
        const int variant = (8 * r256 > 7 * r192) ? (3 | 8) : 3;
        launch_scaled_mm_pp(variant, out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc,
                            a_scale_numel, b_scale_numel, out_dtype, st);
        VA_LAUNCH_CHECK(op);
        return VLLM_AMD_OK;
      }
      const int rc = launch_scaled_mm_large(sel, out, a, b, a_scales, b_scales, bias, M, N, K, lda,
                                            ldb, ldc, a_scale_numel, b_scale_numel, out_dtype, vec16,
                                            splits, ws_mode, st);
      if (rc != VLLM_AMD_OK) return rc;
      VA_LAUNCH_CHECK(op);
      return VLLM_AMD_OK;
    }
  }
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

// The variant of vllm_amd_scaled_mm for callers that cannot allocate or register memory: split-K
// partials and tickets live in the library's static device array (scaled_mm_large.hip).
extern "C" int vllm_amd_scaled_mm_static(void* out, const void* a, const void* b,
                                         const float* a_scales, const float* b_scales,
                                         const void* bias, int M, int N, int K, int64_t lda,
                                         int64_t ldb, int64_t ldc, int a_scale_numel,
                                         int b_scale_numel, int out_dtype, int splits,
                                         void* stream) {
  VA_REQUIRE(splits >= 0, VLLM_AMD_EINVAL, "scaled_mm_static: negative number of K slices");
  return launch_scaled_mm(out, a, b, a_scales, b_scales, bias, M, N, K, lda, ldb, ldc,
                          a_scale_numel, b_scale_numel, out_dtype, (hipStream_t)stream,
                          "scaled_mm_static", 1, splits);
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
