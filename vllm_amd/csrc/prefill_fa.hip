// unified_attention (vllm/attention/ops/triton_unified_attention.py:52-272, 588-700) and context_attention_fwd
// (vllm/attention/ops/prefix_prefill.py:36-288, 726-913; the PREFIX instantiation below) for LONG query chunks at head
// sizes 72 .. 128 with the causal + padding mask (unified: + sliding window): the matrix-core-bound case of prefill_attention.hip,
// re-built around what bounds that kernel (16 query rows per wave = one LDS fragment byte per
// 32 flops and ~300 VALU per 32 MFMAs; a 64-row tile restages every K / V tile for 64 rows).
//
//   * a workgroup = 8 waves = 256 tile rows; a wave owns 32 rows and computes with v_mfma_f32_32x32x16:
//     every K / V fragment read from LDS feeds 32 rows (half the LDS bytes per flop of the 16-row form), a K / V
//     tile is staged once per 256 rows (a quarter of the staging work per flop);
//   * rows are (token, q head of ONE kv head's GQA group) pairs, token-major, as the reference packs
//     BLOCK_M / num_queries_per_kv tokens (:612-613): K / V stream once per group;
//   * the transposed problem, as in the decode kernel:  S^T[n, m] = K[n, :] . Q[m, :]  (A = K rows from LDS,
//     B = Q^T from registers),  O^T[d, m] += V^T[d, n] . P^T[n, m]  (A = V^T by the transposing LDS read
//     ds_read_b64_tr_b16 from the row-major V tile, B = P^T).  A lane holds ONE query row (column m) of S^T:
//     row max and sum need one cross-half exchange, and the C layout of S^T (keys (r & 3) + 8 (r >> 2) + 4 hi of
//     a 32-key block) is the B layout of P^T once the contraction slots of the PV product are numbered the
//     same way -- the V^T fragment is simply read in that order.  P never leaves the registers;
//   * K / V tiles of 64 keys, double-buffered in LDS (64 KiB), one barrier per tile: registers hold tile t + 1
//     while tile t is computed; in interval t they are written to the other buffer (fp8 widened exactly on the way)
//     and the loads of tile t + 2 are issued -- by waves 0-3 behind their compute, by waves 4-7 in front of it, so
//     that one wave of every SIMD stages while its partner is in its MFMA sections;
//   * the running max is updated lazily: O and l are rescaled only when some row's tile max exceeds the
//     running max by more than 2^8 (the probabilities then reach at most 256, exact in the 16-bit formats'
//     exponent range; every row is normalised by its own sum at the end).
//
// fp8 cache: K / V are widened exactly (e4m3 is a subset of f16 / bf16), k_scale rides in the score scale,
// v_scale in the softmax exponent (one rounding fewer than the reference's (fp8 * scale).to(dtype)).
#include "mma.cuh"
#include "prefill_fa.cuh"

#include <float.h>
#include <type_traits>

namespace va {
namespace fa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename T> struct Mma32;
template <> struct Mma32<bf16_t> {
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
  }
};
template <> struct Mma32<f16_t> {
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                  0, 0, 0);
  }
};

constexpr int BN = 64;            // keys per tile
constexpr int KN = 2, VN = 2;     // K / V tiles in LDS (rings indexed by the tile number)
#ifndef FA_KRW
#define FA_KRW 8
#define FA_VRW 4
#endif
constexpr int KR128 = 8, VR128 = 4;   // fragment reads in flight: K reads, V^T fragments (two transposing reads each)
constexpr int KR256 = FA_KRW, VR256 = FA_VRW;
constexpr float THR = 8.f;        // lazy rescale threshold (log2 domain)

__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;   // (fmaxf on MFMA results makes hipcc canonicalise every input first)
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// own value and the value of lane ^ 32, in some order.  v_permlane32_swap vdst, src exchanges lanes 32-63 of vdst
// with lanes 0-31 of src; inline asm (with the two wait states a VALU write of an operand needs in front of it),
// because hipcc folds the two results of __builtin_amdgcn_permlane32_swap into one when they are combined again.
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xmax(float x) {
  float a = x, b = x;
  swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xsum(float x) {
  float a = x, b = x;
  swap32(a, b);
  return a + b;
}
__device__ __forceinline__ u32x2 lds_tr16(const uint8_t* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(u32x2, v);
}

// -DVA_PROBES (tools/fa_stamps.py): shader-clock time per wave spent in each part of the tile loop, summed over
// the wave's tiles: 0 K ring fill + staging (LDS writes of tile t + 1, loads of t + 2), 2 S^T, 4 softmax + PV,
// 5 barrier, 6 tiles computed, 7 whole kernel
#ifdef VA_PROBES
__device__ unsigned long long g_fa_stamps[2048 * 8 * 8];
#define FA_T(i)                                                   \
  do {                                                            \
    const unsigned long long now__ = __builtin_readcyclecounter(); \
    acc__[i] += now__ - last__;                                   \
    last__ = now__;                                               \
  } while (0)
#else
#define FA_T(i) do { } while (0)
#endif

// PREFIX = context_attention_fwd (vllm/attention/ops/prefix_prefill.py:36-288, 726-913): the keys are the sequence's
// cached context -- K [block][kv head][D / x][BS][x] (x = 16 bytes), V [block][kv head][D][BS], every cached token
// visible -- followed by the new tokens' dense k / v (causal).  Context V is already V^T: its tile is staged as
// [d][token] (8-byte units XOR-swizzled by d >> 1) and the PV MFMA's A operand is two plain 8-byte reads; the new
// tokens take the unified path's row-major image and transposing reads.  Dense k / v are never fp8.
//
// D_ = 128, NW_ = 8: head sizes 72 .. 128, eight waves = 256 rows, two waves per SIMD.
// D_ = 256, NW_ = 4: head sizes 136 .. 256: O^T alone is 128 registers for a wave's 32 rows and Q^T 64, so a wave takes a
// whole SIMD's register file: four waves = 128 rows, one per SIMD, 32 KiB tiles (128 KiB of LDS), 32 + 32 MFMAs per tile.
// SOFTCAP (unified_attention): scores softcap * tanh(score / softcap) (triton_unified_attention.py:219-221): a pass of
// ~9 VALU per score between S^T and the softmax, in its own instantiation.
template <typename T, bool FP8, bool PREFIX, int D_, int NW_, bool SOFTCAP>
__global__ __launch_bounds__(64 * NW_) void fa_fwd_kernel(const FaParams p) {
  constexpr int NW = NW_, BM = 32 * NW, D = D_;
  constexpr int ROWB = 2 * D;       // bytes per LDS row (one key, D 16-bit values)
  constexpr int TILE = BN * ROWB;   // 16 / 32 KiB
  constexpr int K_OFF = 0, V_OFF = KN * TILE;
  constexpr int KS = D / 16, DB = D / 32;         // k-steps of S^T, d blocks of O^T
  constexpr int KR = D > 128 ? KR256 : KR128, VR = D > 128 ? VR256 : VR128;
  constexpr int PPK = D / 16;                     // 32-byte staging pieces per key
  constexpr int NTHR = 64 * NW, IPT = BN * PPK / NTHR, KSTEP = NTHR / PPK;   // staging items per thread, keys between them
  static_assert(IPT * NTHR == BN * PPK && (IPT == 1 || KSTEP % 16 == 0) && 4 * D == IPT * NTHR, "");
  __shared__ __attribute__((aligned(1024))) uint8_t lds[(KN + VN) * TILE];   // 64 / 128 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, hi = lane >> 5;
  // ablations are compile-time (-DVA_FA_ABL=n, tools/fa_stamps.sh; bits in prefill_fa.cuh): a run-time branch
  // around every MFMA costs the schedule it is meant to measure
#ifndef VA_FA_ABL
#define VA_FA_ABL 0
#endif
  constexpr int abl = VA_FA_ABL;
#ifdef VA_PROBES
  unsigned long long acc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_entry__ = __builtin_readcyclecounter();
  unsigned long long last__ = t_entry__;
#endif

  // ---- workgroup -> (row block of some sequence, head).  Hardware ids go round-robin over the XCDs: the
  // bijective remap gives every XCD a contiguous range of logical ids (row blocks fastest, then heads), so the
  // row blocks and heads that read the same K / V share an L2; later (longer) row blocks first.
  int wg = blockIdx.x + gridDim.x * blockIdx.y;
  {
    const int nwg = gridDim.x * gridDim.y;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int g = gridDim.x - 1 - wg % gridDim.x;
  const int y = wg / gridDim.x;
  const int pack = p.pack, tpb = BM / pack;   // tokens per workgroup
  // grid.x enumerates the row blocks of ALL sequences (num_tokens / tpb + num_seqs is an upper bound); the
  // sequence is the last one whose first block cu[i] / tpb + i is <= g (find_seq_idx, :33-48)
  // (blocks_per_seq > 0: the caller did not say how many query tokens there are; every sequence gets that many blocks)
  int seq = 0;
  if (p.blocks_per_seq > 0) {
    seq = g / p.blocks_per_seq;
  } else {
    int lo = 0, hi2 = p.num_seqs;
    while (hi2 - lo > 1) {
      const int mid = (lo + hi2) >> 1;
      if (p.cu_seqlens_q[mid] / tpb + mid <= g) lo = mid; else hi2 = mid;
    }
    seq = lo;
  }
  const int tok0 = p.cu_seqlens_q[seq];
  const int qlen = p.cu_seqlens_q[seq + 1] - tok0;
  const int seq_len = p.seq_lens[seq];
  const int ctx = seq_len - qlen;
  // first query token of this block
  const int tok_base = (p.blocks_per_seq > 0 ? g % p.blocks_per_seq : g - (tok0 / tpb + seq)) * tpb;
  if (tok_base >= qlen) return;
  if (p.skip_decode && qlen == 1) return;
  const int new_end = min(tok_base + tpb, qlen);
  const int kvh = pack > 1 ? y : y / (p.num_heads / p.num_kv_heads);
  // head sizes 72 .. 128 (multiples of 8; 16 with an fp8 cache) run padded to 128: the columns d >= dr of Q, K and V are
  // staged as zeros from clamped addresses, the output columns are not stored
  const int dr = p.head_size;

  // ---- this lane's query row: tile row 32 wave + l32 -> (token, head) ----------------------------------------
  const int mloc = 32 * wave + l32;
  const int rtok = tok_base + mloc / pack;
  const int rhead = pack > 1 ? kvh * pack + mloc % pack : y;
  const bool rok = mloc < tpb * pack && rtok < qlen;
  // last visible key: among the new-domain keys (UNI: every key, by position; PREFIX: the new tokens, by index) and
  // among the context keys (PREFIX)
  const int bound_new = rok ? (PREFIX ? rtok : ctx + rtok) : -1;
  const int bound_ctx = rok ? ctx - 1 : -1;
  // Q^T fragments (B operand of S^T): column m = this row, k = d = 16 ks + 8 hi + (0..7)
  u32x4 qf[KS];
  {
    const T* qp = (const T*)p.q + (int64_t)(tok0 + (rok ? rtok : 0)) * p.q_stride_t + (int64_t)(rok ? rhead : 0) * p.q_stride_h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool dok = 16 * ks + 8 * hi < dr;   // head sizes below 128: zero columns
      const u32x4 v = *reinterpret_cast<const u32x4*>(qp + (dok ? 16 * ks + 8 * hi : 0));
      qf[ks] = (rok && dok) ? v : (u32x4){0, 0, 0, 0};
    }
  }
  // wave-uniform: the tokens of the wave's rows
  const bool w_rows = 32 * wave < tpb * pack && tok_base + (32 * wave) / pack < qlen;
  const int w_lo = (PREFIX ? 0 : ctx) + tok_base + (32 * wave) / pack;
  const int w_hi = (PREFIX ? 0 : ctx) + min(tok_base + (32 * wave + 31) / pack, new_end - 1);

  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *p.k_scale; vscale = *p.v_scale; }
  const int window = PREFIX ? 0 : p.window;          // keys (position - window, position] are visible; 0 = all
  const float sc2 = p.sm_scale * kLog2e;             // > 0 (host checks sm_scale)
  const float sc2k = sc2 * kscale;                   // fp8 keys: k_scale rides in the score scale
  const float vlog2 = FP8 ? log2f(vscale) : 0.f, vinv = FP8 ? 1.0f / vscale : 1.f;

  // tiles: ctx_tiles of the context (PREFIX), then the new domain's
  const int ctx_tiles = PREFIX ? (ctx + BN - 1) / BN : 0;
  const int new_tiles = ((PREFIX ? new_end : ctx + new_end) + BN - 1) / BN;
  const int ntiles = ctx_tiles + new_tiles;
  const int* bt = p.block_table + (int64_t)seq * p.bt_stride;
  const int last_page = PREFIX ? (ctx > 0 ? (ctx - 1) >> p.bs_shift : 0) : (seq_len - 1) >> p.bs_shift;

  // ---- staging: item = tid + NTHR i -> (key = tid / PPK + KSTEP i, 32 bytes of 16-bit values = LDS chunks 2 piece,
  // 2 piece + 1 with piece = tid % PPK) ----------------------------------------------------------------------------
  const int skey = tid / PPK, spiece = tid % PPK;
  const bool pv0 = 16 * spiece < dr, pv1 = 16 * spiece + 8 < dr;   // the thread's two 8-column chunks exist
  const int spl = pv0 ? spiece : 0, o1 = pv1 ? 8 : 0;              // clamped piece / second chunk (16-bit elements)
  constexpr int NST = (FP8 && !PREFIX) ? 1 : 2;
  u32x4 kst[IPT][NST], vst[IPT][NST];
  // PREFIX, context V^T: item -> (d = item % D, tokens 16 vq .. of the tile with vq = item / D)
  auto vd_of = [&](int i) __attribute__((always_inline)) { return (tid + NTHR * i) % D; };
  auto vq_of = [&](int i) __attribute__((always_inline)) { return (tid + NTHR * i) / D; };
  struct Blk { int k[IPT], v[IPT]; };
  auto load_blk = [&](int t) __attribute__((always_inline)) {
    Blk b;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      b.k[i] = 0; b.v[i] = 0;
      if constexpr (PREFIX) {
        if (t < ctx_tiles) {
          b.k[i] = bt[min((t * BN + skey + KSTEP * i) >> p.bs_shift, last_page)];
          b.v[i] = bt[min((t * BN + 16 * vq_of(i)) >> p.bs_shift, last_page)];
        }
      } else {
        if (t < ntiles) b.k[i] = bt[min((t * BN + skey + KSTEP * i) >> p.bs_shift, last_page)];
      }
    }
    return b;
  };
  auto ld16 = [](const void* ptr) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(ptr); };
  auto issue_loads = [&](int t, const Blk& blk) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int key = skey + KSTEP * i;
      if constexpr (PREFIX) {
        if (t < ctx_tiles) {
          const int bsm = (1 << p.bs_shift) - 1;
          const int slot = (t * BN + key) & bsm;
          const uint8_t* kb = p.k_cache + (int64_t)blk.k[i] * p.kc_block_bytes + (int64_t)kvh * p.kc_head_bytes;
          if constexpr (FP8) {
            kst[i][0] = ld16(kb + (((spl << p.bs_shift) + slot) << 4));
          } else {
            kst[i][0] = ld16(kb + ((((2 * spl) << p.bs_shift) + slot) << 4));
            kst[i][1] = ld16(kb + ((((2 * spl + (pv1 ? 1 : 0)) << p.bs_shift) + slot) << 4));
          }
          const int slot0 = (t * BN + 16 * vq_of(i)) & bsm;
          const uint8_t* vb = p.v_cache + (int64_t)blk.v[i] * p.vc_block_bytes + (int64_t)kvh * p.vc_head_bytes +
                              ((min(vd_of(i), dr - 1) << p.bs_shift) + slot0) * (FP8 ? 1 : 2);
          vst[i][0] = ld16(vb);
          if constexpr (!FP8) vst[i][1] = ld16(vb + 16);
        } else {
          const int64_t row = tok0 + min((t - ctx_tiles) * BN + key, qlen - 1);
          const T* kr = (const T*)p.k_new + row * p.k_stride_t + (int64_t)kvh * p.k_stride_h + 16 * spl;
          const T* vr = (const T*)p.v_new + row * p.v_stride_t + (int64_t)kvh * p.v_stride_h + 16 * spl;
          kst[i][0] = ld16(kr); kst[i][1] = ld16(kr + o1);
          vst[i][0] = ld16(vr); vst[i][1] = ld16(vr + o1);
        }
      } else {
        const int slot = (t * BN + key) & ((1 << p.bs_shift) - 1);
        const uint8_t* kr = p.k_cache + (int64_t)blk.k[i] * p.kc_block_bytes + (int64_t)slot * p.kc_slot_bytes +
                            (int64_t)kvh * p.kc_head_bytes + 16 * NST * spl;
        const uint8_t* vr = p.v_cache + (int64_t)blk.k[i] * p.vc_block_bytes + (int64_t)slot * p.vc_slot_bytes +
                            (int64_t)kvh * p.vc_head_bytes + 16 * NST * spl;
#pragma unroll
        for (int j = 0; j < NST; ++j) {
          kst[i][j] = ld16(kr + (j ? 2 * o1 : 0));   // (fp8: one load; 16-bit: the second chunk 16 bytes on, if it exists)
          vst[i][j] = ld16(vr + (j ? 2 * o1 : 0));
        }
      }
    }
  };
  // K image: chunk ^= key & 15 (conflict-free ds_read_b128 of 32 keys x one chunk); V image: chunk ^= (key & 3) << 2
  // (the four keys of a transposing read land in four different 64-byte bank ranges); context V^T image (PREFIX):
  // rows of 128 bytes = 64 tokens of one d, 8-byte unit ^= (d >> 1) & 15 (32 rows x one unit: every bank once).
  // (the keys of a thread's items are KSTEP apart, a multiple of 16: one pair of swizzled chunk offsets for all)
  uint8_t* k_wr = lds + K_OFF + skey * ROWB;
  uint8_t* v_wr = lds + V_OFF + skey * ROWB;
  const int kc0 = ((2 * spiece) ^ (skey & 15)) << 4, kc1 = ((2 * spiece + 1) ^ (skey & 15)) << 4;
  const int vc0 = ((2 * spiece) ^ ((skey & 3) << 2)) << 4, vc1 = ((2 * spiece + 1) ^ ((skey & 3) << 2)) << 4;
  auto u4 = [](u32x4 v) __attribute__((always_inline)) { return make_uint4(v[0], v[1], v[2], v[3]); };
  auto write_tile = [&](int t) __attribute__((always_inline)) {
    const int kb_ = (t & (KN - 1)) * TILE, vb_ = (t & (VN - 1)) * TILE;
    const bool is_ctx = PREFIX && t < ctx_tiles;
    const bool conv = FP8 && (!PREFIX || is_ctx);   // this tile's registers hold e4m3
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int key = skey + KSTEP * i;
      // slots past the sequence hold stale bytes (possibly NaN): K and V rows of such keys are zeroed (their
      // scores are 0 + the mask's -inf, their probabilities 0, and 0 * NaN = NaN)
      const bool live = PREFIX ? (is_ctx ? t * BN + key < ctx : (t - ctx_tiles) * BN + key < qlen)
                               : t * BN + key < seq_len;
      u32x4 k0, k1, v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
      if (conv) {
        uint4 lo, hi4;
        widen16<T>(u4(kst[i][0]), lo, hi4);
        k0 = (u32x4){lo.x, lo.y, lo.z, lo.w};
        k1 = (u32x4){hi4.x, hi4.y, hi4.z, hi4.w};
      } else {
        k0 = kst[i][0]; k1 = kst[i][NST - 1];
      }
      if (!live || !pv0) k0 = (u32x4){0, 0, 0, 0};
      if (!live || !pv1) k1 = (u32x4){0, 0, 0, 0};
      *reinterpret_cast<u32x4*>(k_wr + kb_ + KSTEP * i * ROWB + kc0) = k0;
      *reinterpret_cast<u32x4*>(k_wr + kb_ + KSTEP * i * ROWB + kc1) = k1;
      if (is_ctx) {
        // V^T: the item's 16 tokens of row vd = four 8-byte units
        const int vd = vd_of(i), vq = vq_of(i);
        const int valid = vd < dr ? ctx - (t * BN + 16 * vq) : 0;
        uint4 lo, hi4;
        if constexpr (FP8) {
          widen16<T>(mask_bytes(u4(vst[i][0]), valid), lo, hi4);
        } else {
          lo = mask_halfs(u4(vst[i][0]), valid);
          hi4 = mask_halfs(u4(vst[i][NST - 1]), valid - 8);
        }
        uint8_t* row = lds + V_OFF + vb_ + vd * 128;
        const int f = (vd >> 1) & 15;
        *reinterpret_cast<u32x2*>(row + (((4 * vq + 0) ^ f) << 3)) = (u32x2){lo.x, lo.y};
        *reinterpret_cast<u32x2*>(row + (((4 * vq + 1) ^ f) << 3)) = (u32x2){lo.z, lo.w};
        *reinterpret_cast<u32x2*>(row + (((4 * vq + 2) ^ f) << 3)) = (u32x2){hi4.x, hi4.y};
        *reinterpret_cast<u32x2*>(row + (((4 * vq + 3) ^ f) << 3)) = (u32x2){hi4.z, hi4.w};
      } else {
        if (conv) {
          uint4 lo, hi4;
          widen16<T>(u4(vst[i][0]), lo, hi4);
          v0 = (u32x4){lo.x, lo.y, lo.z, lo.w};
          v1 = (u32x4){hi4.x, hi4.y, hi4.z, hi4.w};
        } else {
          v0 = vst[i][0]; v1 = vst[i][NST - 1];
        }
        if (!live || !pv0) v0 = (u32x4){0, 0, 0, 0};
        if (!live || !pv1) v1 = (u32x4){0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(v_wr + vb_ + KSTEP * i * ROWB + vc0) = v0;
        *reinterpret_cast<u32x4*>(v_wr + vb_ + KSTEP * i * ROWB + vc1) = v1;
      }
    }
  };

  // ---- fragment read addresses (lane constants; buffer, key block and key group are immediates) ---------
  const uint8_t* k_rd[KS];   // K rows l32 (+ 32 kb), chunk 2 ks + hi
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = lds + K_OFF + l32 * ROWB + (((2 * ks + hi) ^ (l32 & 15)) << 4);
  // transposing read: the 16 lanes of group G = lane >> 4 read keys 4 hi + (0..3) (+ 16 g16 + 8 e) x d
  // 32 db + 16 (G & 1) + (0..15); lane 4 q + pp of the group addresses key q, d 4 pp .. 4 pp + 3 and receives
  // d = 32 db + (lane & 31) of the four keys
  const uint8_t* v_rd[DB];
  {
    const int q = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < DB; ++db)
      v_rd[db] = lds + V_OFF + (4 * hi + q) * ROWB + ((4 * (db ^ q) + 2 * g1 + (pp >> 1)) << 4) + 8 * (pp & 1);
  }

  // (D = 256: the same addresses from one offset each, XORed with the k-step / d block / unit at the read)
  const int k_off0 = K_OFF + l32 * ROWB + ((hi ^ (l32 & 15)) << 4);
  const int v_off0 = V_OFF + (4 * hi + ((lane & 15) >> 2)) * ROWB + ((4 * ((lane & 15) >> 2) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) << 4) +
                     8 * (lane & 1);
  // context V^T (PREFIX): row d = 32 db + l32, unit 4 g16 + 2 e + hi = tokens 16 g16 + 8 e + 4 hi .. + 3
  int vt_o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vt_o[i] = V_OFF + l32 * 128 + (((2 * i + hi) ^ ((l32 >> 1) & 15)) << 3);

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DB];   // O^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][this row]
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  // ---- one K / V tile --------------------------------------------------------------------------------------------
  // A wave's instruction stream is what bounds this kernel (in-kernel stamps, profiles/r03_prefill_fa.txt: with
  // every MFMA removed a tile still took 94 % of its time -- LDS fragment reads a fixed short distance ahead of
  // their use, then the softmax, then the next reads), so the three kinds of work are interleaved inside the
  // wave, and the sched_barriers keep hipcc from undoing it:
  //   * fragment reads run a ring ahead of the MFMAs that consume them (8 K reads, 4 V^T fragments); the K ring is
  //     filled right behind the barrier, IN FRONT of the staging work (LDS writes of tile t + 1, address
  //     arithmetic and loads of tile t + 2), the V^T ring in front of the serial part of the softmax;
  //   * S^T is computed key block by key block, the row maximum of block 0 is taken between the MFMAs of block 1;
  //   * the probabilities of key group g + 1 (exp2, row sum, pack: 7 VALU per MFMA) are computed between the
  //     four PV MFMAs of group g.
  // (all per-tile state is local to this lambda: declared outside the tile loop and assigned under `if (active)`
  // it would be live around the whole loop)
  auto compute = [&](int t, auto masked_c, auto vt_c) __attribute__((always_inline)) {
  constexpr bool MASKED = decltype(masked_c)::value;
  constexpr bool VT = decltype(vt_c)::value;               // a context tile of the PREFIX kernel
  constexpr bool TILE_FP8 = FP8 && (!PREFIX || VT);        // the tile's K / V were e4m3
  const float sct0 = TILE_FP8 ? sc2k : sc2;            // raw score -> log2 domain
  const float sct = SOFTCAP ? 1.f : sct0;                // (soft-capped scores are rewritten in the log2 domain)
  const int bound0 = VT ? bound_ctx - t * BN : bound_new - (t - ctx_tiles) * BN;
  f32x16 s[2];
  u32x4 pb[4];   // P^T fragments of key groups 16 g16 ..: slots 0-3 = keys 4 hi + (0..3), 4-7 = 8 + 4 hi + (0..3)
  u32x4 kf[KR];
  const int kofs = (t & (KN - 1)) * TILE;
  const int vofs = (t & (VN - 1)) * TILE;
  auto k_read = [&](auto ic, int kofs) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;   // read i -> (key block i / KS, k-step i % KS)
    if constexpr (D > 128)   // one base offset, the k-step XORed in (the tile is 1 KiB aligned): 15 registers fewer
      return *reinterpret_cast<const u32x4*>(lds + ((k_off0 ^ ((i % KS) << 5)) + kofs + (i / KS) * 32 * ROWB));
    else
      return *reinterpret_cast<const u32x4*>(k_rd[i % KS] + kofs + (i / KS) * 32 * ROWB);
  };
  static_for<KR>([&](auto ic) { kf[decltype(ic)::value] = k_read(ic, kofs); });
  __builtin_amdgcn_sched_barrier(0);
  float mx = 0.f;   // running maximum of the tile's raw scores in this lane
  {
    static_for<2 * KS>([&](auto ic) {
      constexpr int i = decltype(ic)::value, kb = i / KS, ks = i % KS;
      const u32x4 f = kf[i % KR];
      if constexpr (i + KR < 2 * KS) kf[i % KR] = k_read(std::integral_constant<int, i + KR>{}, kofs);
      if constexpr (ks == 0) {
        // the accumulator starts at 0, or at -inf where the key is invisible to this row: the mask rides in the
        // MFMA's C operand (K rows past the sequence are staged as zeros, so nothing there is NaN or inf)
        f32x16 z;
        const int lim = bound0 - 32 * kb - 4 * hi;
        const int lim_lo = window > 0 ? lim - window + 1 : -(1 << 30);   // sliding window: keys (bound - window, bound]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          z[r] = (!MASKED || (c <= lim && c >= lim_lo)) ? 0.f : -INFINITY;
        }
        s[kb] = Mma32<T>::mfma(f, qf[0], z);
      } else {
        if (!(abl & 4)) s[kb] = Mma32<T>::mfma(f, qf[ks], s[kb]);
      }
      if constexpr (kb == 1 && ks < 8 && !SOFTCAP) {   // block 0 is complete: its maximum, two scores per MFMA
        if constexpr (ks == 0) mx = max3(s[0][0], s[0][1], s[0][1]);
        else mx = max3(mx, s[0][2 * ks], s[0][2 * ks + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  FA_T(2);
  u32x4 vf[VR];
  auto v_read = [&](auto ic, int vofs) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;   // MFMA i -> (key group i / DB, d block i % DB)
    if constexpr (VT) {
      const int oa = D > 128 ? (vt_o[0] ^ ((2 * (i / DB)) << 4)) : vt_o[2 * (i / DB)];
      const int ob = D > 128 ? (vt_o[0] ^ ((2 * (i / DB) + 1) << 4)) : vt_o[2 * (i / DB) + 1];
      const u32x2 va = *reinterpret_cast<const u32x2*>(lds + oa + vofs + 4096 * (i % DB));
      const u32x2 vb = *reinterpret_cast<const u32x2*>(lds + ob + vofs + 4096 * (i % DB));
      return (u32x4){va[0], va[1], vb[0], vb[1]};
    } else {
      const uint8_t* vp = D > 128 ? lds + ((v_off0 ^ ((i % DB) << 6)) + vofs) : v_rd[i % DB] + vofs;
      const u32x2 va = lds_tr16(vp + (16 * (i / DB)) * ROWB);
      const u32x2 vb = lds_tr16(vp + (16 * (i / DB) + 8) * ROWB);
      return (u32x4){va[0], va[1], vb[0], vb[1]};
    }
  };
  {
    static_for<VR>([&](auto ic) { vf[decltype(ic)::value] = v_read(ic, vofs); });
    __builtin_amdgcn_sched_barrier(0);
    // lane holds S^T[key = 64 t + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi][this row]
    if constexpr (SOFTCAP) {
      // x = cap2 tanh(y), y = score / softcap, cap2 = softcap log2 e; tanh from one exp2 of -2 |y| and one rcp.  A
      // masked score (-inf) would come out as -cap2: kept at -inf.
      const float cap2 = p.softcap * kLog2e, c1 = sct0 / cap2;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float raw = s[kb][r];
          const float y = raw * c1;
          const float e2 = fast_exp2(-2.0f * kLog2e * fabsf(y));
          const float th = (1.0f - e2) * __builtin_amdgcn_rcpf(1.0f + e2);
          const float x = copysignf(cap2 * th, y);
          s[kb][r] = (MASKED && raw == -INFINITY) ? -INFINITY : x;
        }
      mx = max3(s[0][0], s[0][1], s[0][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = max3(mx, s[1][r], s[1][r + 1]);
    const float ms = xmax(mx) * sct;
    if (__any(ms > m_run + THR)) {
      const float m_new = fmaxf(m_run, ms);
      const float alpha = m_new == -INFINITY ? 1.f : fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    // v_scale rides in the exponent: exp2(s - (m - log2 v_scale)) = p * v_scale
    const float msub = (m_run == -INFINITY ? 0.f : m_run) - (TILE_FP8 ? vlog2 : 0.f);
    float psum = 0.f;
    auto p_word = [&](auto gc, auto jc) __attribute__((always_inline)) {   // word j2 of pb[g16]: two probabilities
      constexpr int g16 = decltype(gc)::value, j2 = decltype(jc)::value, r0 = 8 * (g16 & 1) + 2 * j2;
      float e0 = fmaf(s[g16 >> 1][r0], sct, -msub), e1 = fmaf(s[g16 >> 1][r0 + 1], sct, -msub);
      if (!(abl & 1)) { e0 = fast_exp2(e0); e1 = fast_exp2(e1); }
      psum += e0;
      psum += e1;
      pb[g16][j2] = Mma<T>::pack2(e0, e1);
    };
    static_for<4>([&](auto jc) { p_word(std::integral_constant<int, 0>{}, jc); });
    __builtin_amdgcn_sched_barrier(0);
    // O^T += V^T . P^T
    static_for<4 * DB>([&](auto ic) {
      constexpr int i = decltype(ic)::value, g16 = i / DB, db = i % DB;
      const u32x4 f = vf[i % VR];
      if constexpr (i + VR < 4 * DB) vf[i % VR] = v_read(std::integral_constant<int, i + VR>{}, vofs);
      if (!(abl & 8) || i < 4) o[db] = Mma32<T>::mfma(f, pb[g16], o[db]);
      if constexpr (g16 + 1 < 4 && db < 4) p_word(std::integral_constant<int, g16 + 1>{}, std::integral_constant<int, db>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    l_run += TILE_FP8 ? psum * vinv : psum;   // the sum in units of the unscaled probabilities
  }
  };

  // ---- pipeline ------------------------------------------------------------------------------------------------
  // Registers hold tile t + 1 (loads in flight) while tile t is computed from LDS buffer t & 1; in interval t they
  // are written to buffer (t + 1) & 1 -- last read for tile t - 1, before the barrier that ended interval t - 1 --
  // and the loads of tile t + 2 are issued.  One barrier per tile.
  // (sliding window: the tiles left of the window of the workgroup's first row are not even staged)
  const int t_start = window > 0 ? max(ctx + tok_base - window + 1, 0) / BN : 0;
  Blk blk = load_blk(t_start);
  issue_loads(t_start, blk);
  blk = load_blk(t_start + 1);
  write_tile(t_start);
  if (t_start + 1 < ntiles) issue_loads(t_start + 1, blk);
  blk = load_blk(t_start + 2);
  __syncthreads();
#ifdef VA_PROBES
  last__ = __builtin_readcyclecounter();
#endif
  // A wave's tiles come in three runs: interior tiles (every key visible to every row of the wave), masked
  // tiles (the diagonal and the sequence's last tile), tiles above its rows (staging only).  One loop per run:
  // with both compute copies behind a branch inside one loop hipcc spills 40 registers at the join.
  // STAGGER: the second wave of every SIMD (waves 4-7) does its share of the staging at the TOP of the interval,
  // the first one at the bottom: while one wave converts, writes and computes addresses, its partner is in its
  // MFMA sections (all eight waves in step leave the matrix pipe idle during the staging and collide in it
  // otherwise: in-kernel stamps showed tile time = MFMA time + everything else, no overlap at all).
  const bool stage_first = (p.stagger & 1) && NW == 8 && wave >= NW / 2;   // (one wave per SIMD: nothing to stagger)
  // ("pf_fa_stagger" bits 1 / 2, measurement: a static priority for waves 4-7 / 0-3)
  if ((p.stagger & 2) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if ((p.stagger & 4) && wave < NW / 2) __builtin_amdgcn_s_setprio(1);
  auto stage = [&](int t) __attribute__((always_inline)) {
    if (!(abl & 2)) {
      if (t + 1 < ntiles) write_tile(t + 1);
      if (t + 2 < ntiles) issue_loads(t + 2, blk);
      blk = load_blk(t + 3);
    }
  };
  auto tile_loop = [&](int t_begin, int t_end, auto mode_c, auto vt_c) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_c)::value;
    for (int t = t_begin; t < t_end; ++t) {
      if (stage_first) stage(t);
      FA_T(1);
      if constexpr (MODE < 2) {
        compute(t, std::integral_constant<bool, MODE == 1>{}, vt_c);
        FA_T(4);
#ifdef VA_PROBES
        acc__[6] += 1;
#endif
      }
      // (behind the compute the fragment rings are dead: room for the conversion temporaries)
      if (!stage_first) stage(t);
      FA_T(0);
      if (!(abl & 16)) __syncthreads();
      FA_T(5);
    }
  };
  constexpr std::integral_constant<int, 0> INTERIOR{};
  constexpr std::integral_constant<int, 1> MASKED_{};
  constexpr std::integral_constant<int, 2> SKIP{};
  // new-domain tiles u = t - ctx_tiles: interior while 64 u + 63 <= w_lo, computed while 64 u <= w_hi
  const int n2 = w_rows ? min(w_hi / BN + 1, new_tiles) : 0;
  const int n1 = w_rows ? min((w_lo + 1) / BN, n2) : 0;
  if constexpr (PREFIX) {
    // context: whole tiles are interior, the last one (ctx % 64 tokens) is masked
    const int c2 = w_rows ? ctx_tiles : 0, c1 = w_rows ? ctx / BN : 0;
    tile_loop(0, c1, INTERIOR, std::true_type{});
    tile_loop(c1, c2, MASKED_, std::true_type{});
    tile_loop(c2, ctx_tiles, SKIP, std::true_type{});
  }
  if constexpr (PREFIX) {
    tile_loop(ctx_tiles, ctx_tiles + n1, INTERIOR, std::false_type{});
    tile_loop(ctx_tiles + n1, ctx_tiles + n2, MASKED_, std::false_type{});
    tile_loop(ctx_tiles + n2, ntiles, SKIP, std::false_type{});
  } else {
    // sliding window: tiles wholly left of every row's window are skipped (64 t + 63 <= w_lo - window), tiles that
    // some row sees only in part (64 t <= w_hi - window) are masked
    const int e2 = max(n2, t_start);
    const int l0 = (window > 0 && w_rows) ? max(w_lo - window + 1, 0) / BN : 0;
    const int l1 = (window > 0 && w_rows && w_hi >= window) ? (w_hi - window) / BN + 1 : 0;
    const int b0 = min(max(l0, t_start), e2), b1 = min(max(l1, b0), e2), b2 = min(max(n1, b1), e2);
    tile_loop(t_start, b0, SKIP, std::false_type{});
    tile_loop(b0, b1, MASKED_, std::false_type{});
    tile_loop(b1, b2, INTERIOR, std::false_type{});
    tile_loop(b2, e2, MASKED_, std::false_type{});
    tile_loop(e2, ntiles, SKIP, std::false_type{});
  }

#ifdef VA_PROBES
  if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {
    acc__[7] = __builtin_readcyclecounter() - t_entry__;
    for (int i = 0; i < 8; ++i) g_fa_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = acc__[i];
  }
#endif
  // ---- O^T / l -> out ------------------------------------------------------------------------------------------
  if (!rok) return;
  const float l = xsum(l_run);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (p.out_scale == nullptr) {
    T* op = (T*)p.o + (int64_t)(tok0 + rtok) * p.o_stride_t + (int64_t)rhead * p.o_stride_h + 4 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const u32x2 w = {Mma<T>::pack2(o[db][4 * rq] * inv, o[db][4 * rq + 1] * inv),
                         Mma<T>::pack2(o[db][4 * rq + 2] * inv, o[db][4 * rq + 3] * inv)};
        if (32 * db + 8 * rq + 4 * hi < dr) *reinterpret_cast<u32x2*>(op + 32 * db + 8 * rq) = w;
      }
  } else {
    const float oinv = inv / *p.out_scale;   // acc / l / out_scale, clamped to the e4m3 range (:282-287)
    uint8_t* op = (uint8_t*)p.o + (int64_t)(tok0 + rtok) * p.o_stride_t + (int64_t)rhead * p.o_stride_h + 4 * hi;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = fmaxf(-448.0f, fminf(o[db][4 * rq + r] * oinv, 448.0f));
        if (32 * db + 8 * rq + 4 * hi < dr)
          *reinterpret_cast<uint32_t*>(op + 32 * db + 8 * rq) = f32x4_to_fp8x4(yv[0], yv[1], yv[2], yv[3]);
      }
  }
}

}  // namespace fa

int g_pf_fa_stagger = 1;   // tuning "pf_fa_stagger": 0 = all eight waves stage at the bottom of the interval
int launch_unified_fa(const FaParams& p, int dtype, bool fp8, hipStream_t st) {
  const bool wide = p.head_size > 128;                 // 128-row workgroups of four waves
  const int tpb = (wide ? 128 : 256) / p.pack;
  const unsigned gx = p.blocks_per_seq > 0 ? (unsigned)(p.blocks_per_seq * p.num_seqs) : (unsigned)(p.num_tokens / tpb + p.num_seqs);
  const dim3 grid(gx, (unsigned)(p.pack > 1 ? p.num_kv_heads : p.num_heads));
#define FA_LAUNCH___(T_, F_, P_, D_, NW_, S_) \
  hipLaunchKernelGGL((fa::fa_fwd_kernel<T_, F_, P_, D_, NW_, S_>), grid, dim3(64 * NW_), 0, st, p)
#define FA_LAUNCH__(T_, F_, P_, D_, NW_)                                                           \
  do {                                                                                             \
    if (!P_ && p.softcap > 0.f) FA_LAUNCH___(T_, F_, false, D_, NW_, true);                        \
    else FA_LAUNCH___(T_, F_, P_, D_, NW_, false);                                                 \
  } while (0)
#define FA_LAUNCH_(T_, F_, P_)                                                          \
  do {                                                                                  \
    if (wide) FA_LAUNCH__(T_, F_, P_, 256, 4); else FA_LAUNCH__(T_, F_, P_, 128, 8);    \
  } while (0)
#define FA_LAUNCH(T_, F_)                                                   \
  do {                                                                      \
    if (p.prefix) FA_LAUNCH_(T_, F_, true); else FA_LAUNCH_(T_, F_, false); \
  } while (0)
  if (dtype == VLLM_AMD_BF16) { if (fp8) FA_LAUNCH(bf16_t, true); else FA_LAUNCH(bf16_t, false); }
  else { if (fp8) FA_LAUNCH(f16_t, true); else FA_LAUNCH(f16_t, false); }
#undef FA_LAUNCH
#undef FA_LAUNCH_
#undef FA_LAUNCH__
#undef FA_LAUNCH___
  return VLLM_AMD_OK;
}

}  // namespace va

#ifdef VA_PROBES
extern "C" int vllm_amd_probe_fa_stamps(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::fa::g_fa_stamps), (size_t)nwg * 8 * 8 * 8) == hipSuccess ? 0 : -3;
}
#endif
