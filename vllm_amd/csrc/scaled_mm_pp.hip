// fp8 scaled GEMM for gfx950, MFMA-bound problems (M >= ~1024): the 256 x 256 x 128 B tile of
// scaled_mm_large.hip re-built around two things its phase stamps asked for
// (profiles/r02_gemm_ablation.txt, section 6):
//
//   * STAGGER.  Eight waves that run the same program behind one barrier per K block all issue
//     their LDS-DMA refill and wait for their first fragments at the same time: the matrix pipe
//     idles ~600 cycles after every barrier and the two waves of a SIMD then run their MFMA
//     sections one after the other (2964 cycles per block against 2048 of MFMA).  Here waves 4-7
//     (the second wave of every SIMD) run HALF A BLOCK BEHIND: they take the fragments of a
//     block's last four m-steps into registers before the barrier and issue those 16 MFMAs
//     right behind it -- while waves 0-3 are queueing their refill and waiting for fragments --
//     and only then stage, read and compute the first half of the new block.
//   * PERSISTENT tiles.  A workgroup walks its XCD's share of the tile list (grid = 256); the
//     LDS-DMA of the next tile's first K block is issued before the epilogue of the tile that
//     just ended, and the epilogue is per wave (a wave's 16 x 64 sub-tiles pass through 4 KiB of
//     its own in the 32 KiB of LDS the two stages leave free: no workgroup barrier, stage
//     buffers untouched), so tiles do not pay launch ramp + first-load latency + a whole-chip
//     store burst each.
//
//   out[M,N] = ((sa (.) A[M,K]) . (sb (.) W[N,K]^T)).to(out_t) (+ bias)     K % 128 == 0, N % 8 == 0
//
// Layout, swizzle, fragment reads and MFMA as in scaled_mm_large.hip (transposed product: the
// MFMA's A operand = weight rows; a lane ends with 4 consecutive n of one output row).
#include "common.cuh"

namespace va {
namespace pp {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int LBK = 128;                 // K block in bytes = one MFMA
constexpr int BM = 256;
constexpr int NW = 8, WN = 4;            // waves as 2 (m) x 4 (n): 128 x 16 NT outputs per wave
constexpr int MT = 8;                    // MFMA tiles per wave along m; along n: the template's NT (4, or 3 = 256 x 192 tiles)
constexpr int A_BYTES = BM * LBK, EPI_WAVE = 4096;
constexpr int ROUNDS = BM / (8 * NW);    // 4 LDS-DMA rounds per wave for the activation rows (NT for the weight rows)

// -DVA_PROBES (tools/pp_stamps.py): 100 MHz time stamps per wave, stored as they are taken: 0 kernel
// entry, 1 exit of the first barrier, 2 end of the K loop, 3 epilogue stores issued, 4 stores done;
// 5 / 6: shader clock at 1 / 2
#ifdef VA_PROBES
__device__ unsigned long long g_pp_stamps[512 * 8 * 8];
#define PP_STAMP(i, expr)                                                                       \
  do {                                                                                          \
    const unsigned long long t__ = (expr);                                                      \
    if (lane == 0 && blockIdx.x < 512) g_pp_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + (i)] = t__; \
  } while (0)
#else
#define PP_STAMP(i, expr) do { } while (0)
#endif

__device__ __forceinline__ int lswz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); }

template <int OFF>
__device__ __forceinline__ i32x4 lds_read16(uint32_t addr) {
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// fragment T (16 rows) of an operand: two 16-byte halves at the per-lane addresses a0 / a1
template <int T>
__device__ __forceinline__ void read_frag(uint32_t a0, uint32_t a1, i32x4& lo, i32x4& hi) {
  lo = lds_read16<T * 16 * LBK>(a0);
  hi = lds_read16<T * 16 * LBK>(a1);
}
// s_waitcnt lgkmcnt(LATER) = all LDS reads but the LATER youngest have returned; the fragments go
// through the asm as in/out operands so that their consumers stay behind it
template <int LATER>
__device__ __forceinline__ void wait2(i32x4& a, i32x4& b) {
  static_assert(LATER <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b) : [n] "n"(LATER));
}
template <int LATER>
__device__ __forceinline__ void wait4(i32x4& a, i32x4& b, i32x4& c, i32x4& d) {
  static_assert(LATER <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : [n] "n"(LATER));
}
__device__ __forceinline__ f32x4 mfma(i32x8 w, i32x8 a, f32x4 acc) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
__device__ __forceinline__ i32x8 cat8(i32x4 lo, i32x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename out_t, bool STAGGER, int DEFER, int NT>
__global__ __launch_bounds__(64 * NW) void scaled_mm_pp_kernel(
    out_t* __restrict__ out, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
    const float* __restrict__ a_scales, const float* __restrict__ b_scales,
    const out_t* __restrict__ bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
    int a_scale_numel, int b_scale_numel, int persistent, int pw, int store_mode, int prio) {
  // tile = 256 x 64 NT: 256 x 256, or 256 x 192 for outputs whose 256-wide tiles leave the last round of workgroups
  // half empty (N 6144 at M 4096: 384 tiles = 2 rounds of 256 workgroups, 512 tiles of 192 columns = 2 rounds of 3/4 the work)
  constexpr int BN = 64 * NT, W_BYTES = BN * LBK, STAGE = A_BYTES + W_BYTES;
  constexpr int EPI_OFF = 2 * STAGE, LDS_TOTAL = EPI_OFF + NW * EPI_WAVE;   // 160 KiB (NT 4), 144 KiB (NT 3)
  constexpr int WROUNDS = BN / (8 * NW);
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TOTAL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  PP_STAMP(0, __builtin_amdgcn_s_memrealtime());
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const bool grp_b = STAGGER && wave >= NW / 2;   // wave-uniform
  // Output stores are non-temporal: measured 4096^3 51.7 (plain) / 49.6 (nt) / 50.5 us (write-through, sc0 sc1),
  // 4096 x 28672 343 / 338 / 350 us -- plain stores leave the tile dirty in L2 and the kernel boundary writes it
  // back.  store_mode (tuning "mm_store", measurement only): 3 = no stores (garbage: what the stores cost).

  // ---- this workgroup's tiles: ids first, first + stride, .. (cnt of them) -------------------
  // Hardware workgroup i runs on XCD i % 8; every XCD gets a contiguous run of the tile list
  // (tm fastest), so tiles that share a weight panel share an L2.
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int T = tiles_m * tiles_n;
  int first, stride, cnt;
  {
    const int q = T / 8, r = T % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int mine = q + (xcd < r ? 1 : 0);        // tiles of this XCD
    first = start + idx;
    if (persistent) {
      stride = gridDim.x / 8;                      // host: gridDim.x % 8 == 0
      cnt = idx < mine ? (mine - idx + stride - 1) / stride : 0;
    } else {
      stride = 0;
      cnt = 1;                                     // host: gridDim.x == T
    }
  }
  if (cnt == 0) return;
  // tile id -> (tm, tn): ids run through "super-columns" of pw tile columns, tn fastest, so that the 32
  // tiles an XCD works on at a time form a 8 x 4 patch (12 operand panels through its L2 instead of
  // the 18 of a 16 x 2 patch); pw <= 1: tm fastest
  auto tile_mn = [&](int id, int& tm, int& tn) __attribute__((always_inline)) {
    if (pw <= 1) { tm = id % tiles_m; tn = id / tiles_m; return; }
    const int per = tiles_m * pw, sc = id / per, r = id - sc * per;
    const int w = min(pw, tiles_n - sc * pw);
    tm = r / w;
    tn = sc * pw + (r - tm * w);
  };

  // ---- staging: in round r wave w fills rows [64 r + 8 w, +8): lane -> (row = lane >> 3, slot = lane & 7)
  // and loads logical chunk slot ^ f(row), so that the lane-linear LDS image is the swizzled tile
  const int srow = lane >> 3, sslot = lane & 7;
  uint32_t a_voff[ROUNDS], w_voff[WROUNDS];   // of the tile whose blocks are being LOADED
  auto set_load_tile = [&](int id) __attribute__((always_inline)) {
    int tm, tn;
    tile_mn(id, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int row = 64 * r + 8 * wave + srow;
      const uint32_t sw = (uint32_t)((sslot ^ lswz(row)) << 4);
      const int m = min(m0 + row, M - 1), n = min(n0 + row, N - 1);   // rows past the end: any valid row
      a_voff[r] = (uint32_t)m * (uint32_t)lda + sw;
      if (r < WROUNDS) w_voff[r] = (uint32_t)n * (uint32_t)ldb + sw;
    }
  };
  auto stage_load = [&](int stage, int k0) __attribute__((always_inline)) {
    uint8_t* base = lds + stage * STAGE + 8 * wave * LBK;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + k0 + a_voff[r]),
                                       (__attribute__((address_space(3))) void*)(base + 64 * r * LBK), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < WROUNDS; ++r)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + k0 + w_voff[r]),
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + 64 * r * LBK), 16, 0, 0);
  };

  // fragment read addresses: tile t of an operand is an immediate offset t * 16 * LBK
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  uint32_t w_addr[2], a_addr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = ((2 * g4 + j) ^ lswz(c16)) << 4;
    w_addr[j] = lds_base + A_BYTES + (wn * (BN / WN) + c16) * LBK + ch;
    a_addr[j] = lds_base + (wm * (BM / 2) + c16) * LBK + ch;
  }

  f32x4 acc[NT][MT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // ---- epilogue of one tile, per wave: lane holds C^T[n = 16 nt + 4 g4 + r][m = 16 mt + c16] -----
  // A 16 x 64 sub-tile (one m-step) goes through the wave's own 2 x 2 KiB of LDS as [16 rows][128 B]
  // (16-byte chunks XOR-swizzled by the row) and leaves as two store instructions of 8 rows x 128 B.
  auto epilogue = [&](int id) __attribute__((always_inline)) {
    int tm, tn;
    tile_mn(id, tm, tn);
    const int m0 = tm * BM + wm * (BM / 2), n0 = tn * BN + wn * (BN / WN);
    // every scale / bias load up front, unconditionally, from clamped addresses: one wait
    float sbv[NT][4], sav[MT];
    out_t bv[NT][4];
    const bool has_bias = bias != nullptr;
    const out_t* bp = has_bias ? bias : reinterpret_cast<const out_t*>(b_scales);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = min(n0 + nt * 16 + 4 * g4 + r, N - 1);
        sbv[nt][r] = b_scales[b_scale_numel == 1 ? 0 : n];
        bv[nt][r] = bp[has_bias ? n : 0];
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sav[mt] = a_scales[a_scale_numel == 1 ? 0 : min(m0 + mt * 16 + c16, M - 1)];
    uint8_t* mine = lds + EPI_OFF + wave * EPI_WAVE;
    const int rrow = lane >> 3, rch = lane & 7;   // read-back: row within 8, 16-byte chunk
    // store address of (row rrow, chunk rch) of sub-tile 0; a row down = ldc8 16-byte units
    const bool ncol_ok = n0 + 8 * rch < N && rch < 2 * NT && store_mode != 3;
    u32x4* orow = reinterpret_cast<u32x4*>(out + (int64_t)(m0 + rrow) * ldc + n0 + 8 * rch);
    const int64_t ldc8 = ldc / 8;   // host: ldc % 8 == 0
    // software pipeline over the m-steps: sub-tile mt is written to buffer mt & 1 while sub-tile mt - 1 is
    // read back from the other buffer and stored.  Two copies behind ONE wave-uniform branch: with the
    // bias folded in by a select per element the loop is VALU-bound (9 instead of 3 instructions per output).
    auto body = [&](auto with_bias) __attribute__((always_inline)) {
      constexpr bool BIAS = decltype(with_bias)::value;
      static_for<MT + 1>([&](auto ic) {
        constexpr int mt = decltype(ic)::value;
        u32x4 v[2];
        if constexpr (mt > 0) {
          const uint8_t* rb = mine + ((mt - 1) & 1) * 2048;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = 8 * i + rrow;
            v[i] = *reinterpret_cast<const u32x4*>(rb + row * 128 + ((rch ^ (row & 7)) << 4));
          }
        }
        if constexpr (mt < MT) {
          uint8_t* wb = mine + (mt & 1) * 2048;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            struct alignas(8) V4 { out_t e[4]; } pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[nt][mt][r] * sav[mt];
              x = x * sbv[nt][r];
              out_t o = from_f32<out_t>(x);
              if constexpr (BIAS) o = from_f32<out_t>(to_f32(o) + to_f32(bv[nt][r]));
              pk.e[r] = o;
            }
            const int ch = (2 * nt + (g4 >> 1)) ^ (c16 & 7);
            *reinterpret_cast<V4*>(wb + c16 * 128 + (ch << 4) + (g4 & 1) * 8) = pk;
          }
        }
        if constexpr (mt > 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int m = m0 + (mt - 1) * 16 + 8 * i + rrow;
            if (m < M && ncol_ok) __builtin_nontemporal_store(v[i], orow + (int64_t)(8 * i + (mt - 1) * 16) * ldc8);
          }
        }
        __builtin_amdgcn_wave_barrier();
      });
    };
    if (has_bias) body(std::true_type{});
    else body(std::false_type{});
  };

  // ---- K loop over the blocks of all this workgroup's tiles ---------------------------------
  // One extra trip at the end runs only what follows a tile's last barrier: group B's deferred half
  // block and the epilogue.
  const int ng = K / LBK;
  const int total = cnt * ng;

  // The whole loop once per wave group (a wave-uniform branch around two copies): with both roles in
  // one loop body the registers one role keeps across the barrier are live in the other role's code too.
  auto run = [&](auto role) __attribute__((always_inline)) {
    constexpr bool GRP_B = decltype(role)::value;
    i32x8 wf[NT];                // weight fragments of the current block (group B: kept across the barrier)
    i32x8 df[DEFER];             // group B: activation fragments of the last DEFER m-steps, deferred
    int pending = -1;            // tile whose epilogue is still to run
    int id = first;              // tile of the block being computed
    int g = 0;                   // its K block
    for (int gb = 0; gb <= total; ++gb) {
      // own LDS-DMA of this block has landed; then everybody's has, and everybody is done reading the
      // other stage (group B: has ITS fragments of it in registers)
      if (prio != 9) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ("mm_prio" 9, measurement: no wait for the refill -- garbage)
      __builtin_amdgcn_s_barrier();
#ifdef VA_PROBES
      if (gb == 0) { PP_STAMP(1, __builtin_amdgcn_s_memrealtime()); PP_STAMP(5, __builtin_amdgcn_s_memtime()); }
      if (gb == total) { PP_STAMP(2, __builtin_amdgcn_s_memrealtime()); PP_STAMP(6, __builtin_amdgcn_s_memtime()); }
#endif
      if constexpr (GRP_B) {
        if (gb > 0) {
          // the last DEFER m-steps of the previous block: every operand is in registers (the reads
          // were waited for in front of the barrier)
          static_for<DEFER>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[nt][MT - DEFER + j] = mfma(wf[nt], df[j], acc[nt][MT - DEFER + j]);
          });
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next block (of this tile, or the first of the next tile) into the other stage
      const bool more = gb + 1 < total;
      const bool more_k = g + 1 < ng;
      const int nk0 = more_k ? (g + 1) * LBK : 0;
      const uint32_t so = (gb & 1) * STAGE;
      if (pending >= 0) {
        // tile boundary: the refill first (it overlaps the epilogue), then the finished tile
        if (more) stage_load((gb & 1) ^ 1, nk0);
        epilogue(pending);
#ifdef VA_PROBES
        if (gb == total) {
          PP_STAMP(3, __builtin_amdgcn_s_memrealtime());
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          PP_STAMP(4, __builtin_amdgcn_s_memrealtime());
        }
#endif
        if (gb == total) break;
        zero_acc();
      }
      const uint32_t wa0 = w_addr[0] + so, wa1 = w_addr[1] + so;
      const uint32_t aa0 = a_addr[0] + so, aa1 = a_addr[1] + so;
      i32x4 al[2], ah[2], wl[NT], wh[NT];
      // first reads of a block in the order its MFMAs consume them: W0 A0 W1 W2 W3
      read_frag<0>(wa0, wa1, wl[0], wh[0]);
      read_frag<0>(aa0, aa1, al[0], ah[0]);
      read_frag<1>(wa0, wa1, wl[1], wh[1]);
      read_frag<2>(wa0, wa1, wl[2], wh[2]);
      if constexpr (NT > 3) read_frag<3>(wa0, wa1, wl[3], wh[3]);
      if (pending < 0) {
        if (!more_k && more) set_load_tile(id + stride);
        if (more) stage_load((gb & 1) ^ 1, nk0);
      }
      pending = -1;
      // m-steps 0 .. NOW-1 now; group B requests the fragments of its last DEFER m-steps along the way
      // (one per m-step, from the first on) and waits for them in front of the next barrier
      constexpr int NOW = GRP_B ? MT - DEFER : MT;
      static_assert(DEFER <= MT - DEFER, "");
      i32x4 dl[DEFER], dh[DEFER];
      static_for<NOW>([&](auto ic) {
        constexpr int mt = decltype(ic)::value;
        constexpr bool nxt = mt + 1 < NOW;
        constexpr bool dnow = GRP_B && mt < DEFER, dprev = GRP_B && mt > 0 && mt - 1 < DEFER;
        if constexpr (nxt) read_frag<mt + 1>(aa0, aa1, al[(mt + 1) & 1], ah[(mt + 1) & 1]);
        if constexpr (dnow) read_frag<NOW + mt>(aa0, aa1, dl[mt], dh[mt]);
        // reads younger than A(mt): A(mt+1) and D(mt) of this step, D(mt-1) of the one before
        constexpr int later = (nxt ? 2 : 0) + (dnow ? 2 : 0) + (dprev ? 2 : 0);
        if constexpr (mt == 0) {
          wait4<later + 2 * (NT - 1)>(wl[0], wh[0], al[0], ah[0]);
          const i32x8 af = cat8(al[0], ah[0]);
          wf[0] = cat8(wl[0], wh[0]);
          acc[0][0] = mfma(wf[0], af, acc[0][0]);
          __builtin_amdgcn_sched_barrier(0);
          static_for<NT - 1>([&](auto jc) {
            constexpr int nt = decltype(jc)::value + 1;
            wait2<later + 2 * (NT - 1 - nt)>(wl[nt], wh[nt]);
            wf[nt] = cat8(wl[nt], wh[nt]);
            acc[nt][0] = mfma(wf[nt], af, acc[nt][0]);
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          wait2<later>(al[mt & 1], ah[mt & 1]);
          const i32x8 af = cat8(al[mt & 1], ah[mt & 1]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = mfma(wf[nt], af, acc[nt][mt]);
          if constexpr (GRP_B) __builtin_amdgcn_sched_barrier(0);
        }
      });
      if constexpr (GRP_B) {
        // every read of this stage has returned before the barrier that lets it be refilled
        static_for<DEFER>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          wait2<0>(dl[j], dh[j]);
          df[j] = cat8(dl[j], dh[j]);
        });
      }
      if (++g == ng) {
        g = 0;
        pending = id;
        id += stride;
      }
    }
  };
  set_load_tile(first);
  stage_load(0, 0);
  // measurement (tuning "mm_prio"): a static priority for one half of the waves
  if (prio == 1 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if (prio == 2 && wave < NW / 2) __builtin_amdgcn_s_setprio(1);
  if (grp_b) run(std::true_type{});
  else run(std::false_type{});
}

}  // namespace pp

// The kernel addresses its operands with 32-bit byte offsets and writes 16-byte pieces of rows:
// caller guarantees K % 128 == 0, K >= 256, scaled_mm_large_fits(), 16-byte aligned operands,
// ldc % 8 == 0, N % 8 == 0, out 16-byte aligned.  variant bit 0 = stagger, bit 1 = persistent, bit 2 = stagger by a quarter block.
int g_mm_prio = 0;    // tuning "mm_prio" (measurement): 1 = waves 4-7 at priority 1, 2 = waves 0-3
int g_mm_store = 0;   // tuning "mm_store": see the kernel
int g_mm_patch = 8;   // tuning "mm_patch": tile columns per super-column of the tile order (<= 1: column-major)
int launch_scaled_mm_pp(int variant, void* out, const void* a, const void* b, const float* a_scales,
                        const float* b_scales, const void* bias, int M, int N, int K, int64_t lda,
                        int64_t ldb, int64_t ldc, int asn, int bsn, int out_dtype, hipStream_t st) {
  // variant bit 3: 256 x 192 tiles
  const bool narrow = (variant & 8) != 0;
  const int T = ceil_div(M, pp::BM) * ceil_div(N, narrow ? 192 : 256);
  const bool persistent = (variant & 2) && T > 256;
  const int grid = persistent ? 256 : T;
#define PP_LAUNCH_(T_, STG, DFR, NT_)                                                                    \
  hipLaunchKernelGGL((pp::scaled_mm_pp_kernel<T_, STG, DFR, NT_>), dim3(grid), dim3(64 * pp::NW), 0, st,  \
                     (T_*)out, (const uint8_t*)a, (const uint8_t*)b, a_scales, b_scales,            \
                     (const T_*)bias, M, N, K, lda, ldb, ldc, asn, bsn, persistent ? 1 : 0, g_mm_patch, g_mm_store, g_mm_prio)
#define PP_LAUNCH(T_, STG, DFR)                                                  \
  do {                                                                           \
    if (narrow) PP_LAUNCH_(T_, STG, DFR, 3); else PP_LAUNCH_(T_, STG, DFR, 4);   \
  } while (0)
  if (out_dtype == VLLM_AMD_BF16) {
    if ((variant & 5) == 5) PP_LAUNCH(bf16_t, true, 2); else if (variant & 1) PP_LAUNCH(bf16_t, true, 4);
    else PP_LAUNCH(bf16_t, false, 4);
  } else {
    if ((variant & 5) == 5) PP_LAUNCH(f16_t, true, 2); else if (variant & 1) PP_LAUNCH(f16_t, true, 4);
    else PP_LAUNCH(f16_t, false, 4);
  }
#undef PP_LAUNCH_
#undef PP_LAUNCH
  return VLLM_AMD_OK;
}

}  // namespace va

// Measurement helper (bench.py prints its result beside the GEMM's fraction of the 5 PF paper peak): a loop of
// nothing but v_mfma_scale_f32_16x16x128_f8f6f4 on pseudo-random fp8 operands held in registers, 32
// independent accumulators per wave, 8 waves per CU on every CU -- what the matrix pipes deliver under the
// chip's power management when no operand has to be moved (tools/mfmabench2.hip is the stand-alone form).
namespace va {
__global__ __launch_bounds__(512) void mfma_ceiling_kernel(float* __restrict__ sink, int iters) {
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  typedef __attribute__((ext_vector_type(8))) int i32x8;
  const unsigned lane = threadIdx.x;
  i32x8 a[4], b[2];
  unsigned h = lane * 2654435761u + 12345u;
  auto next = [&]() {
    unsigned w = 0;
    for (int k = 0; k < 4; ++k) {
      h = h * 1664525u + 1013904223u;
      w |= ((h >> 24 & 0x80u) | ((h >> 8) % 0x48u)) << (8 * k);   // |x| <= 3.75: no NaN, no overflow
    }
    return (int)w;
  };
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = next();
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = next();
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
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
}  // namespace va

// sink: 256 * 512 floats of device memory.  FLOP of one call = 256 * 8 * iters * 32 * 2 * 16 * 16 * 128.
extern "C" int vllm_amd_mfma_ceiling(void* sink, int iters, void* stream) {
  VA_REQUIRE(sink != nullptr && iters > 0, VLLM_AMD_EINVAL, "mfma_ceiling: null sink or iters <= 0");
  hipLaunchKernelGGL(va::mfma_ceiling_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, (float*)sink, iters);
  VA_LAUNCH_CHECK("mfma_ceiling");
  return VLLM_AMD_OK;
}

#ifdef VA_PROBES
extern "C" int vllm_amd_probe_pp_stamps(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::pp::g_pp_stamps), (size_t)nwg * 8 * 8 * 8) == hipSuccess ? 0 : -3;
}
#endif
