// unified_attention, decode-only launches (every query length <= 1): the bandwidth kernel of
// paged_attention_mfma.hip re-laid for the FLASH cache layout [block][slot][kv head][d] of the
// V1 engine (vllm/attention/ops/triton_unified_attention.py:52-272 for the maths; caller
// vllm/v1/attention/backends/triton_attn.py:297-317).
//
// One workgroup (4 waves) per (sequence, kv head[, key segment]) serves the whole GQA group (16 q
// heads of it at a time), so every K / V byte is read once.  Waves stride over steps of 16 KiB of KV (fp8: 64 tokens,
// 16-bit: 32), each with a private online softmax; one merge through LDS at the end.
//
//   K  a token's row is d-contiguous -- the A-operand order of S^T = K Q^T -- but a lane of the
//      MFMA layout is one token, and 16 lanes reading 16 bytes each of 16 different rows cost
//      the L1 sixteen tag look-ups per 256 bytes (measured 4.5 TB/s).  So K is loaded with
//      whole rows per 8 lanes (fp8, head 128: an instruction = 8 contiguous 128-byte rows),
//      written to a wave-private LDS image (XOR-swizzled on the SOURCE side, so the write is
//      the plain lane order) and read back in MFMA order, conflict-free.  The contraction
//      index is permuted to the chunk order (g4 + 4 i) and Q^T is loaded once in the same order.
//   V  rows are [token][d], the PV product O^T += V^T P^T wants 8 TOKENS of one d per lane.  The
//      contraction slots of lane group g4 are tokens 16 ta + 4 g4 + [0,4) and 16 tb + 4 g4 + [0,4)
//      of a tile pair (ta, tb) -- exactly the tokens whose probabilities the lane holds after the
//      softmax (C layout of S^T), so P goes from the softmax into the B operand WITHOUT leaving
//      the registers (no LDS tile, no shuffle).  For each such token the lane loads d = DT c16 ..
//      + DT (8 bytes fp8 / 16 bytes 16-bit at head 128; per instruction 4 whole token rows of one
//      page) and transposes in registers: v_perm_b32 interleaves the bytes of a token pair, the
//      fp8 -> 16-bit widening (exact) then yields the packed (token, token + 1) operand dwords.
//      MFMA j produces rows d = DT i + j (i = A row): a permutation of d undone in the merge.
//   Addresses are a wave-uniform 64-bit base per (tile, row) plus one 32-bit lane offset per
//   operand that never changes (SGPR base + VGPR offset loads); no workgroup barrier in the loop.
#include "unified_decode.cuh"
#include "mma.cuh"

#include <float.h>

namespace va {

extern int g_uni_decode;

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> struct DwVec;
template <> struct DwVec<1> { typedef unsigned int type; };
template <> struct DwVec<2> { typedef u32x2 type; };
template <> struct DwVec<4> { typedef u32x4 type; };
template <> struct DwVec<8> { typedef __attribute__((ext_vector_type(8))) unsigned int type; };   // head 256, 16-bit cache
template <int N>
__device__ __forceinline__ unsigned int dw_get(const typename DwVec<N>::type& v, int i) {
  if constexpr (N == 1) return v;
  else return v[i];
}
template <int N>
__device__ __forceinline__ typename DwVec<N>::type nt_load(const uint8_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const typename DwVec<N>::type*>(p));
}

template <typename T, bool FP8, int D, int NW>
__global__ __launch_bounds__(NW * 64, ((D > 128 && !FP8) ? 1 : 2)) void ua_decode_kernel(const UniDecodeParams p, int sub_shift, int xcd_group, int order) {
  constexpr int ES = FP8 ? 1 : 2;             // cache element bytes
  constexpr int TILES = (FP8 && D <= 128) ? 4 : 2;   // 16-token tiles per step (the 256-wide form: 32-token steps -- 223 registers, two workgroups per CU)
  constexpr int STEP_TOK = TILES * 16;
  constexpr int KS = TILES / 2;               // k-steps (tile pairs) of the PV product per step
  constexpr int DT = D / 16;                  // PV MFMAs per k-step = d elements a lane owns per token
  constexpr int KL = D * ES / 64;             // K loads per tile and lane (16 B each)
  constexpr int VDW = DT * ES / 4;            // dwords of one token's V row a lane owns
  constexpr int QF = D / 32;                  // k-steps of the QK product
  static_assert(VDW == 1 || VDW == 2 || VDW == 4 || VDW == 8, "");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  // Launches of more than one round of workgroups: consecutive hardware ids go to different XCDs
  // (id % 8); remapped so that the kv heads of a sequence -- adjacent 128-byte pieces of the same
  // token rows -- are fetched through ONE XCD's L2 at about the same time (256 x 4096 fp8:
  // 5.6 -> 5.8 TB/s, 512 x 2048: 5.55 -> 5.94; no gain below two rounds of workgroups).
  int wg = blockIdx.x;
  if (xcd_group && order <= 0) {
    const int nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  // GQA groups of more than 16 q heads (MQA models) take ceil(G / 16) workgroups per kv head,
  // chunk fastest: the chunks re-read the same K / V, through L2 when they run together
  // hp > 1 (round 3, token-major caches): hp ADJACENT kv heads are taken as one row of hp * head_size elements -- one
  // load instruction then fetches 256 contiguous bytes per token instead of 128 (or 64): pieces below 256 bytes cost
  // 10 (128 B) to 40 % (64 B) of the bandwidth, 256-byte pieces nothing (tools/bench_uni_decode_d.py).  The q heads of
  // the hp groups are the columns of ONE Q^T operand, each zero outside its own head's d range, so S^T, the per-column
  // softmax and O^T come out per head with no other change; a column's output is its head's slice of the d rows.
  const int hp = p.hp;
  const int Gq = p.num_heads / p.num_kv_heads;   // q heads per kv head
  const int nkv = p.num_kv_heads / hp;
  const int Gall = Gq * hp;                      // (hp > 1: at most 16)
  const int chunks = (Gall + 15) >> 4;
  // unsplit launches of 257 .. 1024 workgroups: sequences handed out longest first, the two
  // workgroups of a CU paired long + short (pa_lpt_wg, common.cuh; same output bits)
  int lpt_len = -1;
  if (order > 0) {
    const int groups = chunks * nkv;
    const bool xg = xcd_group != 0;
    wg = p.num_seqs <= 64    ? pa_lpt_wg<1>(p.seq_lens, p.num_seqs, groups, order, lane, wg, lpt_len, xg)
         : p.num_seqs <= 256 ? pa_lpt_wg<4>(p.seq_lens, p.num_seqs, groups, order, lane, wg, lpt_len, xg)
         : p.num_seqs <= 512 ? pa_lpt_wg<8>(p.seq_lens, p.num_seqs, groups, order, lane, wg, lpt_len, xg)
                             : pa_lpt_wg<16>(p.seq_lens, p.num_seqs, groups, order, lane, wg, lpt_len, xg);
  }
  const int chunk = wg % chunks, kvh = (wg / chunks) % nkv, seq = wg / (chunks * nkv);
  const int seg = blockIdx.y;
  const int G = min(16, Gall - 16 * chunk);   // q heads of this workgroup
  const int head0 = kvh * Gall + 16 * chunk;
  // D = head size padded to 64 / 128 / 256: chunks at d >= dr are never loaded (zeros: no contribution to
  // the scores, output columns that are never stored)
  const int dsub = p.head_size, dr = dsub * hp;

  // (all three loads before the first branch: one memory round trip, not two)
  const int seq_len = order > 0 ? lpt_len : p.seq_lens[seq];
  const int qtok = p.cu_seqlens_q[seq], qend = p.cu_seqlens_q[seq + 1];
  if (qend - qtok != 1) return;   // not a decode token: the tile kernel's (or nobody's) business
  const int ctx = seq_len - 1;                       // position of the query
  const bool split = p.num_segments > 1;
  const int kv_lo = p.window > 0 ? max(0, seq_len - p.window) : 0;   // first visible key
  int start_tok = 0, end_tok = seq_len;
  if (split) {
    const int ntiles = (seq_len + 63) >> 6;
    const int tps = (ntiles + p.num_segments - 1) / p.num_segments;
    if (seg * tps >= ntiles) return;                 // the merge does not read this segment
    start_tok = seg * tps * 64;
    end_tok = min(seq_len, start_tok + tps * 64);
  }
  start_tok = max(start_tok, kv_lo & ~(STEP_TOK - 1));
  const int nsteps = end_tok > start_tok ? (end_tok - start_tok + STEP_TOK - 1) / STEP_TOK : 0;
  const int last_page = seq_len > 0 ? (seq_len - 1) >> (4 + sub_shift) : 0;
  const int sub_mask = (1 << sub_shift) - 1;
  const int* bt = p.block_table + (int64_t)seq * p.bt_stride;

  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *p.k_scale; vscale = *p.v_scale; }
  const float sc2 = p.sm_scale * kscale * kLog2e;
  const bool col_ok = c16 < G;
  const bool alibi = p.alibi_slopes != nullptr;
  const float slope2 = (alibi && col_ok) ? p.alibi_slopes[head0 + c16] * kLog2e : 0.f;
  const float softcap = p.softcap;
  const float cap_inv = softcap > 0.f ? 1.0f / (kLog2e * softcap) : 0.f, cap2 = softcap * kLog2e;

  constexpr int K_LDS = TILES * 16 * D * ES;   // one step of K per wave
  constexpr int MERGE_BYTES = NW * D * 16 * 4 + NW * 16 * 8;
  constexpr int LDS_BYTES = NW * K_LDS > MERGE_BYTES ? NW * K_LDS : MERGE_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* const k_lds = lds + wave * K_LDS;

  // Q^T fragments (B operand of QK) in the K loads' permutation:
  // fp8 : K load i gives d = 16 (g4 + 4 i) + [0,16) -> two k-steps (lo / hi 8); 16 bit: d = 8 (g4 + 4 i) + [0,8)
  uint4 qf[QF];
  {
    const int dlo = hp > 1 ? (c16 / Gq) * dsub : 0;   // the column's head starts at row d = dlo
    const T* qp = (const T*)p.q + (int64_t)qtok * p.q_stride_t + (int64_t)(head0 + c16) * p.q_stride_h - dlo;
#pragma unroll
    for (int s = 0; s < QF; ++s) {
      int d;
      if constexpr (FP8) d = 16 * (g4 + 4 * (s >> 1)) + 8 * (s & 1);
      else d = 8 * (g4 + 4 * s);
      qf[s] = (col_ok && d >= dlo && d < dlo + dsub) ? *reinterpret_cast<const uint4*>(qp + d) : make_uint4(0, 0, 0, 0);
    }
  }

  float m_run = -FLT_MAX, l_run = 0.f;
  f32x4 o[DT];
#pragma unroll
  for (int j = 0; j < DT; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef typename DwVec<VDW>::type vrow_t;
  u32x4 kreg[TILES * KL];
  vrow_t vreg[TILES * 4];   // [tile][r]: V row piece of token 16 tile + 4 g4 + r
  int pb[TILES];            // block ids of the step whose loads are issued next

  // lane offsets within a tile (bytes, fixed for the whole kernel)
  // K: instruction i of a tile fetches rows i RPI .. + RPI, CPR lanes per row, each lane one
  // 16-byte chunk; the chunk a lane fetches is XOR-swizzled so that the image it writes to LDS
  // (lane order = [row][position]) is read back conflict-free in the MFMA order
  constexpr int RB = D * ES, CPR = RB / 16, RPI = 64 / CPR;   // row bytes, chunks per row, rows per instruction
  auto kswz = [](int row) { return (row * CPR / 16) & (CPR - 1); };
  uint32_t koff[KL], krd[KL];
  bool kval[KL];   // the chunk this lane fetches exists (head sizes below the padded D)
#pragma unroll
  for (int i = 0; i < KL; ++i) {
    const int row = i * RPI + lane / CPR;
    const int chunk = (lane % CPR) ^ kswz(row);
    kval[i] = chunk * 16 < dr * ES;
    koff[i] = (uint32_t)(row * p.kc_slot_bytes + (chunk << 4));
    krd[i] = (uint32_t)(c16 * RB + (((g4 + 4 * i) ^ kswz(c16)) << 4));
  }
  uint32_t voff[4];   // row r of the lane's four tokens of a tile
#pragma unroll
  for (int r = 0; r < 4; ++r) voff[r] = (uint32_t)((4 * g4 + r) * p.vc_slot_bytes + c16 * (DT * ES));
  const bool vval = c16 * DT < dr;   // this lane's d elements exist
  const uint8_t* const k_head = p.k_cache + (int64_t)kvh * hp * p.kc_head_bytes;
  const uint8_t* const v_head = p.v_cache + (int64_t)kvh * hp * p.vc_head_bytes;

  auto load_pb = [&](int step) {
    if (step >= nsteps) return;
    const int tile0 = (start_tok >> 4) + step * TILES;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
      pb[t] = __builtin_amdgcn_readfirstlane(bt[min((tile0 + t) >> sub_shift, last_page)]);
  };
  // (the empty asm keeps the 32-bit offsets from being widened to 64 bits outside the loop,
  // which would turn every load into a 64-bit VGPR address + add instead of SGPR base + offset)
  auto issue_k = [&](int step) {
    const int tile0 = (start_tok >> 4) + step * TILES;
    uint32_t ko[KL];
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      ko[i] = koff[i];
      asm volatile("" : "+v"(ko[i]));
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const uint8_t* base = k_head + (int64_t)pb[t] * p.kc_block_bytes +
                            (int64_t)(((tile0 + t) & sub_mask) * 16) * p.kc_slot_bytes;
#pragma unroll
      for (int i = 0; i < KL; ++i) kreg[t * KL + i] = kval[i] ? nt_load<4>(base + ko[i]) : (u32x4)(0);
    }
  };
  auto issue_v = [&](int step) {
    const int tile0 = (start_tok >> 4) + step * TILES;
    uint32_t vo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vo[r] = voff[r];
      asm volatile("" : "+v"(vo[r]));
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const uint8_t* base = v_head + (int64_t)pb[t] * p.vc_block_bytes +
                            (int64_t)(((tile0 + t) & sub_mask) * 16) * p.vc_slot_bytes;
#pragma unroll
      for (int r = 0; r < 4; ++r) vreg[t * 4 + r] = vval ? nt_load<VDW>(base + vo[r]) : (vrow_t)(0);
    }
  };

  auto do_step = [&](int step) {
    const int tok0 = start_tok + step * STEP_TOK;
    const int pf = step + NW;
    // ---------------- K: registers -> wave-private LDS image -> MFMA order ----------------
#pragma unroll
    for (int n = 0; n < TILES * KL; ++n)
      *reinterpret_cast<u32x4*>(k_lds + n * 1024 + lane * 16) = kreg[n];
    const bool more = pf < nsteps;
    if (more) issue_k(pf);
    // ---------------- S^T = K Q^T ----------------------------------------------------
    f32x4 s[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KL; ++i) {
        const u32x4 kr = *reinterpret_cast<const u32x4*>(k_lds + t * 16 * RB + krd[i]);
        const uint4 raw = make_uint4(kr.x, kr.y, kr.z, kr.w);
        if constexpr (FP8) {
          uint4 lo, hi;
          widen16<T>(raw, lo, hi);
          acc = Mma<T>::mfma(lo, qf[2 * i], acc);
          acc = Mma<T>::mfma(hi, qf[2 * i + 1], acc);
        } else {
          acc = Mma<T>::mfma(raw, qf[i], acc);
        }
      }
      s[t] = acc;
    }

    // ---------------- online softmax: lane holds S^T[token tok0 + 16 t + 4 g4 + r][q head c16] -------
    float mx = -FLT_MAX;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // triton_unified_attention.py:219-238: softcap, causal / padding mask, window, ALiBi
        const int tok = tok0 + 16 * t + 4 * g4 + r;
        float x = s[t][r] * sc2;
        if (softcap > 0.f) {   // softcap * tanh(S / softcap) from one exp2 (log2 domain throughout)
          const float y = x * cap_inv;
          const float e2 = fast_exp2(-2.0f * kLog2e * fabsf(y));
          x = copysignf(cap2 * (1.0f - e2) * __builtin_amdgcn_rcpf(1.0f + e2), y);
        }
        x += slope2 * (float)(tok - ctx);
        x = (tok >= kv_lo && tok < end_tok) ? x : -FLT_MAX;
        s[t][r] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);   // first step: exp2(-huge) = 0
    m_run = m_new;
    float psum = 0.f;
    uint32_t pk[TILES][2];   // P rounded to the query dtype: (r0, r1), (r2, r3) of tile t
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      float e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = tok0 + 16 * t + 4 * g4 + r;
        e[r] = fast_exp2(s[t][r] - m_new);
        e[r] = (tok >= kv_lo && tok < end_tok) ? e[r] : 0.f;
        psum += e[r];
      }
      pk[t][0] = Mma<T>::pack2(e[0], e[1]);
      pk[t][1] = Mma<T>::pack2(e[2], e[3]);
    }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int j = 0; j < DT; ++j) o[j] *= alpha;
    }

    // ---------------- O^T += V^T P^T ----------------------------------------------------
    if (tok0 + STEP_TOK > end_tok) {
      // never-written slots past the sequence may hold NaN patterns: 0 * NaN
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          vreg[t * 4 + r] &= (vrow_t)(tok0 + 16 * t + 4 * g4 + r < end_tok ? 0xffffffffu : 0u);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // contraction slots 0-3 = tile 2 ks, 4-7 = tile 2 ks + 1 (tokens 4 g4 + [0,4) of each)
      const uint4 pfrag = make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]);
#pragma unroll
      for (int j = 0; j < DT; ++j) {
        uint32_t a[4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const vrow_t& ta = vreg[8 * ks + 2 * pr];
          const vrow_t& tb = vreg[8 * ks + 2 * pr + 1];
          if constexpr (FP8) {
            // bytes (ta.b, tb.b, ta.b+1, tb.b+1) of the dword that holds element j
            const uint32_t x = __builtin_amdgcn_perm(dw_get<VDW>(tb, j >> 2), dw_get<VDW>(ta, j >> 2),
                                                     (j & 2) ? 0x07030602u : 0x05010400u);
            a[pr] = Mma<T>::cvt2(x, (j & 1) != 0);
          } else {
            a[pr] = __builtin_amdgcn_perm(dw_get<VDW>(tb, j >> 1), dw_get<VDW>(ta, j >> 1),
                                          (j & 1) ? 0x07060302u : 0x05040100u);
          }
        }
        o[j] = Mma<T>::mfma(make_uint4(a[0], a[1], a[2], a[3]), pfrag, o[j]);
      }
    }
    if (more) {
      issue_v(pf);
      load_pb(pf + NW);
    }
  };

  if (wave < nsteps) {
    load_pb(wave);
    issue_k(wave);
    issue_v(wave);
    load_pb(wave + NW);
  }
  for (int step = wave; step < nsteps; step += NW) {
    do_step(step);
  }

  // ---------------- merge the NW waves ----------------------------------------------------
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();   // the K images are dead; LDS becomes the merge buffer
  float* om = reinterpret_cast<float*>(lds);   // [NW][D][16]
  float* ml = om + NW * D * 16;                 // [NW][16][2]
  if (g4 == 0) {
    ml[(wave * 16 + c16) * 2 + 0] = m_run;
    ml[(wave * 16 + c16) * 2 + 1] = l_run;
  }
  __syncthreads();
  float m_tot = -FLT_MAX;
#pragma unroll
  for (int w = 0; w < NW; ++w) m_tot = fmaxf(m_tot, ml[(w * 16 + c16) * 2]);
  const float f_w = fast_exp2(m_run - m_tot) * vscale;   // waves with no step: m_run = -FLT_MAX, o = 0
  // lane holds O^T[row 4 g4 + r of MFMA j][q head c16] = d = DT (4 g4 + r) + j
#pragma unroll
  for (int j = 0; j < DT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) om[(wave * D + DT * (4 * g4 + r) + j) * 16 + c16] = o[j][r] * f_w;
  __syncthreads();

  // thread -> (q head qh, 8 consecutive d); head 256: 32 threads per q head, two passes over the 16 heads
  constexpr int TPH = D / 8, HPP = NW * 64 / TPH;
  const int d0 = (threadIdx.x % TPH) * 8;
  if (d0 >= dr) return;
  for (int qh = threadIdx.x / TPH; qh < G; qh += HPP) {
    const int dlo = hp > 1 ? (qh / Gq) * dsub : 0;   // this column's head: rows [dlo, dlo + dsub)
    if (d0 < dlo || d0 >= dlo + dsub) continue;
    float mt = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < NW; ++w) mt = fmaxf(mt, ml[(w * 16 + qh) * 2]);
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) lt += ml[(w * 16 + qh) * 2 + 1] * fast_exp2(ml[(w * 16 + qh) * 2] - mt);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += om[(w * D + d0 + j) * 16 + qh];
      acc[j] = a;
    }
    const int head = head0 + qh;
    if (split) {
      // unnormalised partial (v_scale applied), running max in the log2 domain, sum
      const int64_t row = ((int64_t)qtok * p.num_heads + head) * p.segm_stride + seg;
      float* so = p.segm_out + row * dsub + (d0 - dlo);
      *reinterpret_cast<f32x4*>(so) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(so + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
      if (d0 == dlo) {
        p.segm_max[row] = lt > 0.f ? mt : -INFINITY;
        p.segm_sum[row] = lt;
      }
      continue;
    }
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;   // acc / L (:276); nothing visible gives 0
    if (p.out_scale) {
      const float oi = inv * (1.0f / *p.out_scale);
      uint8_t* op = (uint8_t*)p.out + (int64_t)qtok * p.o_stride_t + (int64_t)head * p.o_stride_h + (d0 - dlo);
      *reinterpret_cast<uint2*>(op) =
          make_uint2(f32x4_to_fp8x4(acc[0] * oi, acc[1] * oi, acc[2] * oi, acc[3] * oi),
                     f32x4_to_fp8x4(acc[4] * oi, acc[5] * oi, acc[6] * oi, acc[7] * oi));
    } else {
      T* op = (T*)p.out + (int64_t)qtok * p.o_stride_t + (int64_t)head * p.o_stride_h + (d0 - dlo);
      *reinterpret_cast<uint4*>(op) =
          make_uint4(Mma<T>::pack2(acc[0] * inv, acc[1] * inv), Mma<T>::pack2(acc[2] * inv, acc[3] * inv),
                     Mma<T>::pack2(acc[4] * inv, acc[5] * inv), Mma<T>::pack2(acc[6] * inv, acc[7] * inv));
    }
  }
}

// Merge of the key segments of a split launch (reduce_segments,
// triton_unified_attention.py:499-586): one wave per (sequence, q head).  Lane s holds the max /
// sum of segment s (at most 16 segments), the weights come from two 16-lane butterflies, and
// the 16 partial rows are loaded back to back (segments past the sequence's last one re-read
// that last one with weight 0), so the kernel is two dependent memory round trips.
template <typename T, int D>
__global__ __launch_bounds__(256) void ua_decode_reduce_kernel(const UniDecodeParams p) {
  constexpr int VPL = D / 64;   // outputs per lane
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= p.num_seqs * p.num_heads) return;
  const int seq = pair / p.num_heads, head = pair % p.num_heads;
  const int qtok = p.cu_seqlens_q[seq];
  if (p.cu_seqlens_q[seq + 1] - qtok != 1) return;
  const int ntiles = (p.seq_lens[seq] + 63) >> 6;
  const int tps = (ntiles + p.num_segments - 1) / p.num_segments;
  const int nseg = tps > 0 ? (ntiles + tps - 1) / tps : 0;   // segments the main kernel wrote
  const int64_t row = ((int64_t)qtok * p.num_heads + head) * p.segm_stride;
  float acc[VPL], l = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) acc[j] = 0.f;
  const int dr = p.head_size;
  const bool dval = VPL * lane < dr;
  if (nseg > 0) {
    const float ms = lane < nseg ? p.segm_max[row + lane] : -INFINITY;
    const float ls = lane < nseg ? p.segm_sum[row + lane] : 0.f;
    typedef __attribute__((ext_vector_type(VPL))) float fv;
    fv part[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      part[s2] = dval ? *reinterpret_cast<const fv*>(p.segm_out + (row + min(s2, nseg - 1)) * dr + VPL * lane) : (fv)(0.f);
    float mx = ms;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    const float w = ms == -INFINITY ? 0.f : fast_exp2(ms - mx);
    l = ls * w;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) l += __shfl_xor(l, m, 64);
    l = __shfl(l, 0, 64);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const float ws = __shfl(w, s2, 64);
#pragma unroll
      for (int j = 0; j < VPL; ++j) acc[j] += part[s2][j] * ws;
    }
  }
  if (!dval) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int64_t oidx = (int64_t)qtok * p.o_stride_t + (int64_t)head * p.o_stride_h + VPL * lane;
  if (p.out_scale) {
    const float oi = inv * (1.0f / *p.out_scale);
#pragma unroll
    for (int j = 0; j < VPL; ++j) ((uint8_t*)p.out)[oidx + j] = f32_to_fp8(acc[j] * oi);
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) ((T*)p.out)[oidx + j] = from_f32<T>(acc[j] * inv);
  }
}

template <typename T, bool FP8, int D>
int launch(const UniDecodeParams& p) {
  constexpr int NW = 4;
  int sub_shift = 0;
  while ((16 << sub_shift) < p.block_size) ++sub_shift;
  const int chunks = (p.hp * (p.num_heads / p.num_kv_heads) + 15) / 16;
  const dim3 grid((unsigned)(p.num_seqs * (p.num_kv_heads / p.hp) * chunks), (unsigned)(p.num_segments > 1 ? p.num_segments : 1));
  const int xcd_group = g_uni_decode == 2 ? 1 : g_uni_decode == 3 ? 0 : (grid.x * grid.y >= 1024 ? 1 : 0);
  int order = 0;   // key bits of the longest-first order; tuning "pa_order" = 0 turns it off
  if (pa_order_enabled() && grid.y == 1 && grid.x > 256 && p.num_seqs <= kLptMaxSeqs &&
      (xcd_group ? p.num_seqs % 8 == 0 : grid.x <= 1024)) {
    int ml = p.max_seq_len > 0 ? p.max_seq_len : (1 << 29);
    if (p.window > 0 && p.window < ml) ml = p.window;   // a workgroup streams at most `window` keys
    order = 1;
    while (order < 30 && (ml >> order) != 0) ++order;
  }
  hipLaunchKernelGGL((ua_decode_kernel<T, FP8, D, NW>), grid, dim3(NW * 64), 0, p.stream, p, sub_shift,
                     xcd_group, order);
  VA_LAUNCH_CHECK("unified_attention decode launch");
  if (p.num_segments > 1) {
    hipLaunchKernelGGL((ua_decode_reduce_kernel<T, D>), dim3((unsigned)((p.num_seqs * p.num_heads + 3) / 4)),
                       dim3(256), 0, p.stream, p);
    VA_LAUNCH_CHECK("unified_attention decode merge launch");
  }
  return VLLM_AMD_OK;
}

}  // namespace

bool unified_decode_supported(const UniDecodeParams& p) {
  // head sizes up to 256 in whole 16-byte K chunks and 8-element output pieces (padded to 64 / 128 / 256)
  if (p.head_size < 8 || p.head_size > 256 || p.head_size % (p.fp8 ? 16 : 8) != 0) return false;
  // the 256-wide form's lanes own 16 consecutive d of a V row: whole lanes only (136, 152 .. stay with the tile kernel)
  if (p.head_size > 128 && p.head_size % 16 != 0) return false;
  if (p.block_size < 16 || (p.block_size & (p.block_size - 1)) != 0) return false;
  if (p.num_kv_heads <= 0 || p.num_heads % p.num_kv_heads != 0) return false;
  const int es = p.fp8 ? 1 : 2;
  const int vrow = (p.head_size <= 64 ? 4 : p.head_size <= 128 ? 8 : 16) * es;   // bytes of a V row one lane loads (4 .. 32)
  if (p.vc_slot_bytes % vrow || p.vc_block_bytes % vrow || p.vc_head_bytes % vrow) return false;
  if (p.kc_slot_bytes % 16 || p.kc_block_bytes % 16 || p.kc_head_bytes % 16) return false;
  if (p.q_stride_t % 8 || p.q_stride_h % 8 || (((uintptr_t)p.q) & 15)) return false;
  // one thread stores 8 consecutive outputs: 16 bytes (8 with an fp8 output)
  if (p.o_stride_t % 8 || p.o_stride_h % 8 || (((uintptr_t)p.out) & (p.out_scale ? 7 : 15))) return false;
  if (p.num_segments > 16 || (p.num_segments > 1 && (((uintptr_t)p.segm_out) & 15))) return false;
  return true;
}

int g_uni_hp = 0;   // tuning "uni_hp": 0 = heuristic, 1 = never pack kv heads, 2 / 4 = that many when possible

// kv heads taken per workgroup row (see the kernel): token-major caches whose (token, head) rows are shorter than 256
// bytes; the packed width must be one of the instantiations and the q heads of the packed groups fit the 16 columns
int unified_decode_heads_per_row(const UniDecodeParams& p) {
  if (g_uni_hp == 1 || p.num_kv_heads <= 0) return 1;
  const int es = p.fp8 ? 1 : 2, gq = p.num_heads / p.num_kv_heads;
  if (p.kc_head_bytes != (int64_t)p.head_size * es || p.vc_head_bytes != (int64_t)p.head_size * es) return 1;   // heads not adjacent
  if (p.head_size * es >= 256) return 1;                                                                         // long enough
  // rows of up to 128 elements first (largest packing), then the 256-wide form
  for (int pass = 0; pass < 2; ++pass)
  for (int hp = 4; hp >= 2; hp >>= 1) {
    const int w = p.head_size * hp;
    if ((w > 128) != (pass == 1)) continue;
    if ((w != 64 && w != 128 && w != 256) || gq * hp > 16 || p.num_kv_heads % hp != 0) continue;
    if (g_uni_hp > 1) { if (hp > g_uni_hp) continue; }
    else {
      // Measured (16q / 8kv and 8q / 4kv x 4096 tokens, fp8, tools/bench_uni_decode_d.py): rows of up to 128 elements
      // always gain -- head 64: 2 heads per row 92 -> 52 us at 64 sequences, 291 -> 180 at 256 (4 heads per row, the
      // 256-wide form: 59 / 184); head 128 x 2 in the 256-wide form (fp8: 32-token steps, two workgroups per CU):
      // 8 kv heads 102 -> 90 us at 64 sequences, 194 -> 171 at 128, 355 -> 353 at 256, 32.5 -> 32.1 at 16; 4 kv heads
      // 92.8 -> 87.1 at 128, 180 -> 164.5 at 256.  (w * es <= 256: a 16-bit cache never reaches the 256-wide form here.)
      if (w * es > 256) continue;
    }
    const int vrow = (w <= 64 ? 4 : w <= 128 ? 8 : 16) * es;
    if (p.vc_slot_bytes % vrow || p.vc_block_bytes % vrow || (p.vc_head_bytes * hp) % vrow) continue;
    return hp;
  }
  return 1;
}

int unified_decode(const UniDecodeParams& p) {
  const int w = p.head_size * p.hp;   // (hp from unified_decode_heads_per_row, or 1)
#define L(T) \
  (w > 128 ? (p.fp8 ? launch<T, true, 256>(p) : launch<T, false, 256>(p))  \
   : w > 64 ? (p.fp8 ? launch<T, true, 128>(p) : launch<T, false, 128>(p)) \
                      : (p.fp8 ? launch<T, true, 64>(p) : launch<T, false, 64>(p)))
  return p.dtype == VLLM_AMD_F16 ? L(f16_t) : L(bf16_t);
#undef L
}

}  // namespace va
