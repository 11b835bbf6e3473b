// unified_attention (vllm/attention/ops/triton_unified_attention.py:52-272, 588-700) for LONG query chunks at
// head size 128 with nothing but the causal + padding mask: the matrix-core-bound case of prefill_attention.hip,
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
//   * K / V tiles of 64 keys, double-buffered in LDS, one barrier per tile: the loads of tile t + 2 are issued
//     (global -> registers) before the PV product of tile t, and written to LDS (fp8 widened exactly on the
//     way) after the softmax of tile t + 1;
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

constexpr int BM = 256, BN = 64, D = 128, NW = 8;
constexpr int ROWB = 2 * D;       // bytes per LDS row (one key, 128 16-bit values)
constexpr int TILE = BN * ROWB;   // 16 KiB
constexpr int K_OFF = 0, V_OFF = 2 * TILE;
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

template <typename T, bool FP8>
__global__ __launch_bounds__(64 * NW) void fa_fwd_kernel(const FaParams p) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4 * TILE];   // K0 K1 V0 V1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, hi = lane >> 5;

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
  int seq = 0;
  {
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
  const int tok_base = (g - (tok0 / tpb + seq)) * tpb;   // first query token of this block
  if (tok_base >= qlen) return;
  if (p.skip_decode && qlen == 1) return;
  const int new_end = min(tok_base + tpb, qlen);
  const int kvh = pack > 1 ? y : y / (p.num_heads / p.num_kv_heads);

  // ---- this lane's query row: tile row 32 wave + l32 -> (token, head) ----------------------------------------
  const int mloc = 32 * wave + l32;
  const int rtok = tok_base + mloc / pack;
  const int rhead = pack > 1 ? kvh * pack + mloc % pack : y;
  const bool rok = mloc < tpb * pack && rtok < qlen;
  const int bound = rok ? ctx + rtok : -1;   // last visible key
  // Q^T fragments (B operand of S^T): column m = this row, k = d = 16 ks + 8 hi + (0..7)
  u32x4 qf[8];
  {
    const T* qp = (const T*)p.q + (int64_t)(tok0 + (rok ? rtok : 0)) * p.q_stride_t + (int64_t)(rok ? rhead : 0) * p.q_stride_h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * hi);
      qf[ks] = rok ? v : (u32x4){0, 0, 0, 0};
    }
  }
  // wave-uniform: the tokens of the wave's rows
  const bool w_rows = 32 * wave < tpb * pack && tok_base + (32 * wave) / pack < qlen;
  const int w_lo = ctx + tok_base + (32 * wave) / pack;
  const int w_hi = ctx + min(tok_base + (32 * wave + 31) / pack, new_end - 1);

  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *p.k_scale; vscale = *p.v_scale; }
  const float sct = p.sm_scale * kLog2e * kscale;   // > 0 (host checks sm_scale)
  const float vlog2 = FP8 ? log2f(vscale) : 0.f;

  const int ntiles = (ctx + new_end + BN - 1) / BN;
  const int* bt = p.block_table + (int64_t)seq * p.bt_stride;
  const int last_page = (seq_len - 1) >> p.bs_shift;

  // ---- staging: thread -> (key = tid >> 3, 32 bytes of 16-bit values = LDS chunks 2 piece, 2 piece + 1) -------
  const int skey = tid >> 3, spiece = tid & 7;
  constexpr int NST = FP8 ? 1 : 2;
  u32x4 kst[NST], vst[NST];
  auto load_blk = [&](int t) __attribute__((always_inline)) {
    return t < ntiles ? bt[min((t * BN + skey) >> p.bs_shift, last_page)] : 0;
  };
  auto issue_loads = [&](int t, int blk) __attribute__((always_inline)) {
    const int slot = (t * BN + skey) & ((1 << p.bs_shift) - 1);
    const uint8_t* kr = p.k_cache + (int64_t)blk * p.kc_block_bytes + (int64_t)slot * p.kc_slot_bytes +
                        (int64_t)kvh * p.kc_head_bytes + 16 * NST * spiece;
    const uint8_t* vr = p.v_cache + (int64_t)blk * p.vc_block_bytes + (int64_t)slot * p.vc_slot_bytes +
                        (int64_t)kvh * p.vc_head_bytes + 16 * NST * spiece;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      kst[i] = *reinterpret_cast<const u32x4*>(kr + 16 * i);
      vst[i] = *reinterpret_cast<const u32x4*>(vr + 16 * i);
    }
  };
  // K image: chunk ^= key & 15 (conflict-free ds_read_b128 of 32 keys x one chunk); V image: chunk ^= (key & 3) << 2
  // (the four keys of a transposing read land in four different 64-byte bank ranges)
  uint8_t* k_wr = lds + K_OFF + skey * ROWB;
  uint8_t* v_wr = lds + V_OFF + skey * ROWB;
  const int kc0 = ((2 * spiece) ^ (skey & 15)) << 4, kc1 = ((2 * spiece + 1) ^ (skey & 15)) << 4;
  const int vc0 = ((2 * spiece) ^ ((skey & 3) << 2)) << 4, vc1 = ((2 * spiece + 1) ^ ((skey & 3) << 2)) << 4;
  auto write_tile = [&](int t, int buf) __attribute__((always_inline)) {
    // slots past the sequence hold stale bytes (possibly NaN): V rows of such keys are zeroed (their
    // probabilities are 0, and 0 * NaN = NaN); K rows give scores that the mask replaces
    const bool live = t * BN + skey < seq_len;
    u32x4 k0, k1, v0, v1;
    if constexpr (FP8) {
      uint4 lo, hi4;
      widen16<T>(make_uint4(kst[0][0], kst[0][1], kst[0][2], kst[0][3]), lo, hi4);
      k0 = (u32x4){lo.x, lo.y, lo.z, lo.w};
      k1 = (u32x4){hi4.x, hi4.y, hi4.z, hi4.w};
      widen16<T>(make_uint4(vst[0][0], vst[0][1], vst[0][2], vst[0][3]), lo, hi4);
      v0 = (u32x4){lo.x, lo.y, lo.z, lo.w};
      v1 = (u32x4){hi4.x, hi4.y, hi4.z, hi4.w};
    } else {
      k0 = kst[0]; k1 = kst[NST - 1]; v0 = vst[0]; v1 = vst[NST - 1];
    }
    if (!live) { v0 = (u32x4){0, 0, 0, 0}; v1 = v0; }
    *reinterpret_cast<u32x4*>(k_wr + buf * TILE + kc0) = k0;
    *reinterpret_cast<u32x4*>(k_wr + buf * TILE + kc1) = k1;
    *reinterpret_cast<u32x4*>(v_wr + buf * TILE + vc0) = v0;
    *reinterpret_cast<u32x4*>(v_wr + buf * TILE + vc1) = v1;
  };

  // ---- fragment read addresses (lane constants; buffer, key block and key group are immediates) ---------
  const uint8_t* k_rd[8];   // K rows l32 (+ 32 kb), chunk 2 ks + hi
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = lds + K_OFF + l32 * ROWB + (((2 * ks + hi) ^ (l32 & 15)) << 4);
  // transposing read: the 16 lanes of group G = lane >> 4 read keys 4 hi + (0..3) (+ 16 g16 + 8 e) x d
  // 32 db + 16 (G & 1) + (0..15); lane 4 q + pp of the group addresses key q, d 4 pp .. 4 pp + 3 and receives
  // d = 32 db + (lane & 31) of the four keys
  const uint8_t* v_rd[4];
  {
    const int q = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 4; ++db)
      v_rd[db] = lds + V_OFF + (4 * hi + q) * ROWB + ((4 * (db ^ q) + 2 * g1 + (pp >> 1)) << 4) + 8 * (pp & 1);
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[4];   // O^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][this row]
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  // ---- one K / V tile --------------------------------------------------------------------------------------
  auto compute = [&](auto cur_c, int t, bool masked) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value;
    f32x16 s[2];
    static_for<8>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(k_rd[ks] + cur * TILE + kb * 32 * ROWB);
        if constexpr (ks == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          s[kb] = Mma32<T>::mfma(kf, qf[0], z);
        } else {
          s[kb] = Mma32<T>::mfma(kf, qf[ks], s[kb]);
        }
      }
    });
    // lane holds S^T[key = 64 t + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi][this row]
    if (masked) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int lim = bound - t * BN - 32 * kb - 4 * hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = ((r & 3) + 8 * (r >> 2)) <= lim ? s[kb][r] : -INFINITY;
      }
    }
    float mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
    mx = max3(mx, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, s[1][r], s[1][r + 1]);
    mx = fmaxf(mx, s[1][15]);
    const float ms = xmax(mx) * sct;
    if (__any(ms > m_run + THR)) {
      const float m_new = fmaxf(m_run, ms);
      const float alpha = m_new == -INFINITY ? 1.f : fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    // v_scale rides in the exponent: exp2(s - (m - log2 v_scale)) = p * v_scale
    const float msub = (m_run == -INFINITY ? 0.f : m_run) - vlog2;
    u32x4 pb[4];   // P^T fragments of key groups 16 g16 ..: slots 0-3 = keys 4 hi + (0..3), 4-7 = 8 + 4 hi + (0..3)
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e[j] = fast_exp2(fmaf(s[kb][8 * h2 + j], sct, -msub));
          psum += e[j];
        }
        pb[2 * kb + h2] = (u32x4){Mma<T>::pack2(e[0], e[1]), Mma<T>::pack2(e[2], e[3]), Mma<T>::pack2(e[4], e[5]),
                                  Mma<T>::pack2(e[6], e[7])};
      }
    l_run += psum;
    // O^T += V^T . P^T
#pragma unroll
    for (int g16 = 0; g16 < 4; ++g16)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const u32x2 va = lds_tr16(v_rd[db] + cur * TILE + (16 * g16) * ROWB);
        const u32x2 vb = lds_tr16(v_rd[db] + cur * TILE + (16 * g16 + 8) * ROWB);
        o[db] = Mma32<T>::mfma((u32x4){va[0], va[1], vb[0], vb[1]}, pb[g16], o[db]);
      }
  };

  // ---- pipeline ------------------------------------------------------------------------------------------------
  // registers hold tile t + 1 (loads in flight) while tile t is computed from LDS buffer t & 1; they are written
  // to buffer (t + 1) & 1 -- last read for tile t - 1, before the barrier that ended iteration t - 1 -- after
  // the softmax, then the loads of tile t + 2 are issued in front of the PV product.
  int blk = load_blk(0);
  issue_loads(0, blk);
  blk = load_blk(1);
  write_tile(0, 0);
  if (1 < ntiles) issue_loads(1, blk);
  blk = load_blk(2);
  __syncthreads();
  auto iter = [&](auto cur_c, int t) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value;
    const int n0 = t * BN;
    const bool active = w_rows && n0 <= w_hi;          // wave-uniform
    const bool masked = n0 + BN - 1 > w_lo;
    // (everything a wave does for a tile sits between two barriers; the K / V writes of tile t + 1 go in
    // front of the compute for waves that skip it)
    if (t + 1 < ntiles) write_tile(t + 1, cur ^ 1);
    if (t + 2 < ntiles) issue_loads(t + 2, blk);
    blk = load_blk(t + 3);
    if (active) compute(cur_c, t, masked);
    __syncthreads();
  };
  for (int t = 0; t < ntiles; t += 2) {
    iter(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) iter(std::integral_constant<int, 1>{}, t + 1);
  }

  // ---- O^T / l -> out ------------------------------------------------------------------------------------------
  if (!rok) return;
  const float l = xsum(l_run);
  const float inv = l > 0.f ? vscale / l : 0.f;   // the sum was taken of the v_scale-scaled probabilities
  if (p.out_scale == nullptr) {
    T* op = (T*)p.o + (int64_t)(tok0 + rtok) * p.o_stride_t + (int64_t)rhead * p.o_stride_h + 4 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const u32x2 w = {Mma<T>::pack2(o[db][4 * rq] * inv, o[db][4 * rq + 1] * inv),
                         Mma<T>::pack2(o[db][4 * rq + 2] * inv, o[db][4 * rq + 3] * inv)};
        *reinterpret_cast<u32x2*>(op + 32 * db + 8 * rq) = w;
      }
  } else {
    const float oinv = inv / *p.out_scale;   // acc / l / out_scale, clamped to the e4m3 range (:282-287)
    uint8_t* op = (uint8_t*)p.o + (int64_t)(tok0 + rtok) * p.o_stride_t + (int64_t)rhead * p.o_stride_h + 4 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = fmaxf(-448.0f, fminf(o[db][4 * rq + r] * oinv, 448.0f));
        *reinterpret_cast<uint32_t*>(op + 32 * db + 8 * rq) = f32x4_to_fp8x4(yv[0], yv[1], yv[2], yv[3]);
      }
  }
}

}  // namespace fa

int launch_unified_fa(const FaParams& p, int dtype, bool fp8, hipStream_t st) {
  const int tpb = fa::BM / p.pack;
  const dim3 grid((unsigned)(p.num_tokens / tpb + p.num_seqs), (unsigned)(p.pack > 1 ? p.num_kv_heads : p.num_heads));
#define FA_LAUNCH(T_, F_) hipLaunchKernelGGL((fa::fa_fwd_kernel<T_, F_>), grid, dim3(64 * fa::NW), 0, st, p)
  if (dtype == VLLM_AMD_BF16) { if (fp8) FA_LAUNCH(bf16_t, true); else FA_LAUNCH(bf16_t, false); }
  else { if (fp8) FA_LAUNCH(f16_t, true); else FA_LAUNCH(f16_t, false); }
#undef FA_LAUNCH
  return VLLM_AMD_OK;
}

}  // namespace va
