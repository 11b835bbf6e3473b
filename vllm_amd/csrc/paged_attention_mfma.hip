// Paged attention decode, MI355X bandwidth kernel (gfx950, wave64, MFMA).
//
// One workgroup per (sequence, kv head[, partition]) serves ALL q heads of the
// GQA group, so every K/V byte is read from HBM exactly once.  The work of a
// workgroup is cut into "steps" of 16 KiB of KV (fp8: 4 tiles of 16 tokens,
// 16-bit KV: 2 tiles); the NW waves of the workgroup stride over the steps,
// each keeping a private online-softmax state, and merge through LDS at the end.
//
// Per step and wave:
//   K  16-byte loads straight to VGPRs; for a fixed load index the 64 lanes read
//      one contiguous 1 KiB of the page ([d-chunk][token][16 B] layout: lane =
//      (chunk group g4 = lane>>4, token c16 = lane&15)).  fp8 bytes are widened
//      to the query dtype in registers (exact) and fed as the A operand of
//      mfma_f32_16x16x32: S^T[token][q head] += K[token][d] * Q^T[d][q head].
//      The contraction index d is permuted to match what a lane loaded; Q^T is
//      loaded once per workgroup in the same permutation (B operand).
//   softmax  S^T has the q head on the lane column, so running max / sum are
//      lane-local plus two cross-group shuffles; exp2 with log2(e) folded in.
//   P  rounded to the query dtype (as the reference does), transposed through a
//      2.25 KiB per-wave LDS tile into the B-operand order of the PV product.
//   V  16-byte loads of [d][16 tokens]; lane group g4 owns tile g4 of the step;
//      O^T[d][q head] += V^T[d][token] * P^T[token][q head], 8 d-tiles.
// The loads of step n+1 are issued as soon as step n's registers are free, so
// ~16 KiB per wave stay in flight; no LDS staging of K/V (each byte is used by
// one wave only -- cdna_hip_programming.md "GEMV / M <= 16" rule).
//
// Semantics: reference operators paged_attention_v1 / _v2 / ROCm paged_attention
// (citations in include/vllm_amd.h); fp8 scales folded as in
// csrc/rocm/attention.cu:514-518, 714-717; partials written in the reference's
// tmp_out / exp_sums / max_logits layout and merged by pa_reduce_kernel.
#include "paged_attention.cuh"
#include "mma.cuh"

#include <float.h>
#include <stdlib.h>

namespace va {

constexpr int kDefaultVariant = 1;   // set from measurements, see DESIGN.md

// tools/pa_probe.hip compiles this file with -DVA_PROBES: per-workgroup time stamps (100 MHz
// s_memrealtime) and two ablations (VARIANT bit 2 = loads only, bit 3 = no cross-wave merge).
// Nothing of it exists in the library build.
#ifdef VA_PROBES
// the stamps stay in SGPRs until the kernel's last instruction: a global store in front of the
// block-table loads would turn them from scalar into vector loads (no longer provably unclobbered)
__device__ unsigned long long g_pa_stamps[16384 * 4];
#define VA_STAMP_DECL unsigned long long va_stamp_[4] = {0, 0, 0, 0}
#define VA_STAMP(i) va_stamp_[i] = __builtin_amdgcn_s_memrealtime()
#ifdef VA_PROBE_HWID   // stamp 1 = HW_ID (wave / SIMD / CU / SH / SE ids) | XCC_ID << 32: where the workgroup ran
#define VA_STAMP_HWID() va_stamp_[1] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32)
#else
#define VA_STAMP_HWID() do { } while (0)
#endif
#define VA_STAMP_FLUSH()                                                                  \
  do {                                                                                    \
    va_stamp_[3] = __builtin_amdgcn_s_memrealtime();                                      \
    VA_STAMP_HWID();                                                                      \
    if (threadIdx.x == 0)                                                                 \
      for (int i_ = 0; i_ < 4; ++i_)                                                      \
        g_pa_stamps[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 4 + i_] = va_stamp_[i_]; \
  } while (0)
#else
#define VA_STAMP_DECL do { } while (0)
#define VA_STAMP(i) do { } while (0)
#define VA_STAMP_FLUSH() do { } while (0)
#endif
constexpr int kPRowBytes = 144;  // P tile row (one q head): 64 tok * 2 B + 16 B pad

// kernel arguments
template <typename scalar_t>
struct PaK {
  scalar_t* out; float* exp_sums; float* max_logits; const scalar_t* q;
  const uint8_t* k_cache; const uint8_t* v_cache;
  int num_heads, num_kv_heads; float scale;
  const int* block_tables; const int* seq_lens; const int* query_start_loc;
  int max_num_blocks_per_seq; const float* alibi_slopes;
  int64_t q_stride, kv_block_bytes, kv_head_bytes;
  const float* k_scale_p; const float* v_scale_p;
  int partition_size, max_parts, gqa_chunks, slot_stride, dr_arg;
  // balanced launches (BAL): partial slots, ticket counters, batch size
  float* bal_ws; unsigned* bal_ctr; int num_seqs;
  int window;   // sliding window of the decode token (0 = none): keys [seq_len - window, seq_len)
  int order;    // > 0: longest-first workgroup order (pa_lpt_wg), = bits of the largest sequence length
};

// What a workgroup does with one piece of one sequence: tokens [start_tok, end_tok) of `seq`.
//   mode 0  the piece is the whole sequence: normalised result -> out
//   mode 1  reference partition (v2 / rocm): normalised partial -> tmp_out slot, max / sum beside it
//   mode 2  balanced launch, part of a sequence: unnormalised fp32 partial -> workspace slot, then a
//           ticket; the workgroup that draws the sequence's last ticket merges all its pieces -> out
struct PaPiece {
  int seq; int64_t qrow; int start_tok, end_tok, seq_len; int mode; int part;
  int first_tok = 0;   // tokens in front of it are masked (sliding window; start_tok is its step, rounded down)
  float* slot; const float* slots; unsigned* ticket; int jf, jl, pre, n_tot, gk;
};

// (pa_lpt_wg, the longest-first workgroup order of unpartitioned launches: common.cuh)
template <int HEAD_SIZE> constexpr int pa_slot_floats() { return 16 * HEAD_SIZE + 32; }   // O[16][D], m[16], l[16]

// VARIANT bit 0: non-temporal K/V loads
// Partitioned launches (v2 / rocm): blockIdx.y = super-partition of `slot_stride` reference
// partitions (partition_size = its token count); the partial result goes to scratch slot
// blockIdx.y * slot_stride, the slots in between are not used (see launch_mfma_v).
// PAD: the real head size `dr` is below HEAD_SIZE (a multiple of 8, of 16 with an fp8 cache): d
// chunks / rows at d >= dr do not exist in the cache and are never loaded (zeros), output columns
// there are never stored.
#ifdef VA_PROBES
#define VA_STAMP_PARAM , unsigned long long (&va_stamp_)[4]
#define VA_STAMP_ARG , va_stamp_
#else
#define VA_STAMP_PARAM
#define VA_STAMP_ARG
#endif
template <typename scalar_t, bool FP8, int HEAD_SIZE, int BLOCK_SIZE, int NW, int VARIANT, bool PAD>
__device__ __forceinline__ void pa_piece(const PaK<scalar_t>& A, char* lds, const PaPiece& P, int kvh,
                                         int chunk VA_STAMP_PARAM) {
  scalar_t* __restrict__ out = A.out;
  const scalar_t* __restrict__ q = A.q;
  const uint8_t* __restrict__ k_cache = A.k_cache;
  const uint8_t* __restrict__ v_cache = A.v_cache;
  const int num_heads = A.num_heads, num_kv_heads = A.num_kv_heads;
  const int64_t q_stride = A.q_stride, kv_block_bytes = A.kv_block_bytes, kv_head_bytes = A.kv_head_bytes;
  const float* __restrict__ alibi_slopes = A.alibi_slopes;
  const int dr = PAD ? A.dr_arg : HEAD_SIZE;
  constexpr int ES = FP8 ? 1 : 2;             // cache element bytes
  constexpr int TILES = FP8 ? 4 : 2;          // 16-token tiles per step
  constexpr int STEP_TOK = TILES * 16;
  constexpr int DT = HEAD_SIZE / 16;          // d-tiles of the PV product
  constexpr int KL = HEAD_SIZE * ES / 64;     // K loads per tile and lane (16 B each)
  constexpr int SUBS = BLOCK_SIZE / 16;       // tiles per page
  constexpr int K_CHUNK_BYTES = BLOCK_SIZE * 16;   // one d-chunk of a page: [BS tok][16 B]
  constexpr int V_ROW_BYTES = BLOCK_SIZE * ES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int G = num_heads / num_kv_heads;
  const int head0 = kvh * G + chunk * 16;        // first q head of this workgroup
  const int nheads = min(16, G - chunk * 16);    // q heads served (columns in use)

  const int seq = P.seq;
  const int64_t qrow = P.qrow;
  const int seq_len = P.seq_len;
  const int start_tok = P.start_tok, end_tok = P.end_tok;
  const int nsteps = (end_tok - start_tok + STEP_TOK - 1) / STEP_TOK;
  const int last_page = seq_len > 0 ? (seq_len - 1) / BLOCK_SIZE : 0;
  const int* bt = A.block_tables + (int64_t)seq * A.max_num_blocks_per_seq;
  const float scale = A.scale;
  const float* k_scale_p = A.k_scale_p;
  const float* v_scale_p = A.v_scale_p;

  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *k_scale_p; vscale = *v_scale_p; }
  const float sc2 = scale * kscale * kLog2e;
  const bool col_ok = c16 < nheads;
  const float slope2 = (alibi_slopes && col_ok) ? alibi_slopes[head0 + c16] * kLog2e : 0.f;

  // ---- LDS: per-wave P tile, then (aliased) the cross-wave merge buffers ------
  constexpr int P_BYTES = 16 * kPRowBytes;                       // 2304
  constexpr int MERGE_BYTES = NW * HEAD_SIZE * 16 * 4 + NW * 16 * 8;
  constexpr int LDS_BYTES = (NW * P_BYTES > MERGE_BYTES) ? NW * P_BYTES : MERGE_BYTES;
  char* p_tile = lds + wave * P_BYTES;

  // ---- Q^T fragments (B operand of QK), permuted like the K loads -------------
  // fp8 : load i of a tile gives d = 16*(g4 + 4*i) + [0,16): two k-steps (lo/hi 8)
  // 16b : load i gives d = 8*(g4 + 4*i) + [0,8): one k-step
  constexpr int QF = HEAD_SIZE / 32;  // k-steps of the QK product
  uint4 qf[QF];
  {
    // Unconditional loads from clamped addresses, no select: a lane without a q head (c16 >=
    // nheads) carries the group's first head, its column of S^T / O^T is computed and never
    // stored; a padded d chunk (d >= dr) carries q[0..7], which only ever meets zero K chunks.
    // With a branch around these loads hipcc cannot count the outstanding loads and drains them
    // all -- V included -- in front of the first MFMA of every step.
    const scalar_t* qp = q + qrow * q_stride + (int64_t)(head0 + (col_ok ? c16 : 0)) * dr;
#pragma unroll
    for (int s = 0; s < QF; ++s) {
      int d;
      if constexpr (FP8) d = 16 * (g4 + 4 * (s >> 1)) + 8 * (s & 1);
      else d = 8 * (g4 + 4 * s);
      qf[s] = *reinterpret_cast<const uint4*>(qp + ((!PAD || d < dr) ? d : 0));
    }
  }

  // ---- running state ------------------------------------------------------------
  float m_run = -FLT_MAX;   // running max (log2 units), uniform over g4 for a column
  float l_run = 0.f;        // this lane's partial sum of exp
  f32x4 o[DT];
#pragma unroll
  for (int m = 0; m < DT; ++m) o[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr bool NT = (VARIANT & 1) != 0;
  constexpr bool DEEP = (VARIANT & 2) != 0;
  typedef uint4 KRegs[TILES * KL];
  typedef uint4 VRegs[DT];
  KRegs kregA;
  VRegs vregA;
  auto ld16 = [&](const uint8_t* p) -> uint4 {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4* q4 = reinterpret_cast<const u32x4*>(p);
    u32x4 r;
    if constexpr (NT) r = __builtin_nontemporal_load(q4);
    else r = *q4;
    return make_uint4(r.x, r.y, r.z, r.w);
  };

  auto page_of_tile = [&](int tile) -> int {   // tile index within the sequence
    int pg = tile / SUBS;
    return pg > last_page ? last_page : pg;
  };
  auto issue_k = [&](KRegs& kreg, int step) {
    const int tile0 = (start_tok >> 4) + step * TILES;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int tile = tile0 + t;
      const int64_t pb = bt[page_of_tile(tile)];
      const uint8_t* base = k_cache + pb * kv_block_bytes + (int64_t)kvh * kv_head_bytes +
                            (tile % SUBS) * 256;
#pragma unroll
      for (int i = 0; i < KL; ++i)
        kreg[t * KL + i] = (!PAD || (g4 + 4 * i) * 16 < dr * ES)
                               ? ld16(base + (g4 + 4 * i) * K_CHUNK_BYTES + c16 * 16)
                               : make_uint4(0, 0, 0, 0);
    }
  };
  auto issue_v = [&](VRegs& vreg, int step) {
    const int tile0 = (start_tok >> 4) + step * TILES;
    // lane group g4 owns tile (FP8: g4, 16-bit: g4>>1) of the step
    const int tsel = FP8 ? g4 : (g4 >> 1);
    int pb_sel = 0;
    int sub_sel = 0;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int tile = tile0 + t;
      const int pb = bt[page_of_tile(tile)];
      if (t == tsel) { pb_sel = pb; sub_sel = tile % SUBS; }
    }
    const uint8_t* base = v_cache + (int64_t)pb_sel * kv_block_bytes +
                          (int64_t)kvh * kv_head_bytes + sub_sel * 16 * ES +
                          (FP8 ? 0 : (g4 & 1) * 16);
#pragma unroll
    for (int m = 0; m < DT; ++m)
      vreg[m] = (!PAD || 16 * m + c16 < dr) ? ld16(base + (16 * m + c16) * V_ROW_BYTES)
                                            : make_uint4(0, 0, 0, 0);
  };

  // one step of the online-softmax recurrence on the registers of (kreg, vreg);
  // `pf` = the step whose K/V are loaded into the same registers once they are free
  auto do_step = [&](KRegs& kreg, VRegs& vreg, int step, int pf, auto more_c) {
    constexpr bool more = decltype(more_c)::value;   // this step is followed by step pf of the wave
    const int tok0 = start_tok + step * STEP_TOK;
    // visible tokens: first_tok <= tok < end_tok, as ONE unsigned compare of (tok - first_tok)
    const int tokr0 = tok0 - P.first_tok;
    const unsigned span = (unsigned)(end_tok - P.first_tok);
#ifdef VA_PROBES
    if constexpr ((VARIANT & 4) != 0) {   // ablation: the loads and their issue points only
      uint4 x = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TILES * KL; ++i) { x.x ^= kreg[i].x; x.y ^= kreg[i].y; x.z ^= kreg[i].z; x.w ^= kreg[i].w; }
      if (step == 0) VA_STAMP(1);
      if (more) issue_k(kreg, pf);
#pragma unroll
      for (int m = 0; m < DT; ++m) { x.x ^= vreg[m].x; x.y ^= vreg[m].y; x.z ^= vreg[m].z; x.w ^= vreg[m].w; }
      if (more) issue_v(vreg, pf);
      o[0][0] += __uint_as_float((x.x ^ x.y ^ x.z ^ x.w) & 0x3f000000u);
      (void)tok0;
      return;
    }
#endif
    // ---------------- S^T = K Q^T -------------------------------------------------
    f32x4 s[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KL; ++i) {
        if constexpr (FP8) {
          uint4 lo, hi;
          widen16<scalar_t>(kreg[t * KL + i], lo, hi);
          acc = Mma<scalar_t>::mfma(lo, qf[2 * i], acc);
          acc = Mma<scalar_t>::mfma(hi, qf[2 * i + 1], acc);
        } else {
          acc = Mma<scalar_t>::mfma(kreg[t * KL + i], qf[i], acc);
        }
      }
      s[t] = acc;
    }
    if constexpr (more) issue_k(kreg, pf);
#ifdef VA_PROBES
    if (step == 0) VA_STAMP(1);   // wave 0: its first K tiles have been consumed
#endif

    // ---------------- online softmax ----------------------------------------------
    // lane holds S^T[token = tok0 + 16 t + 4 g4 + r][q head c16]
    float mx = -FLT_MAX;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = tok0 + 16 * t + 4 * g4 + r;
        float v = s[t][r] * sc2 + slope2 * (float)(tok - seq_len + 1);
        v = (unsigned)(tokr0 + 16 * t + 4 * g4 + r) < span ? v : -FLT_MAX;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);   // first step: exp2(-huge) = 0
    m_run = m_new;
    float psum = 0.f;
    uint32_t pk[TILES][2];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = fast_exp2(s[t][r] - m_new);
        e = (unsigned)(tokr0 + 16 * t + 4 * g4 + r) < span ? e : 0.f;
        p[r] = e;
        psum += e;
      }
      pk[t][0] = Mma<scalar_t>::pack2(p[0], p[1]);
      pk[t][1] = Mma<scalar_t>::pack2(p[2], p[3]);
    }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int m = 0; m < DT; ++m) o[m] *= alpha;
    }

    // ---------------- P^T through LDS ------------------------------------------------
    // write: row = q head c16, tokens 16 t + 4 g4 .. +3   (8 bytes)
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      uint2 w = make_uint2(pk[t][0], pk[t][1]);
      *reinterpret_cast<uint2*>(p_tile + c16 * kPRowBytes + (16 * t + 4 * g4) * 2) = w;
    }
    uint4 plo, phi;
    if constexpr (FP8) {
      // lane group g4 consumes tile g4: tokens 16 g4 + [0,8) and + [8,16)
      plo = *reinterpret_cast<const uint4*>(p_tile + c16 * kPRowBytes + (16 * g4) * 2);
      phi = *reinterpret_cast<const uint4*>(p_tile + c16 * kPRowBytes + (16 * g4 + 8) * 2);
    } else {
      // tile g4>>1, tokens 8 (g4&1) + [0,8)  ==  step tokens 8 g4 + [0,8)
      plo = *reinterpret_cast<const uint4*>(p_tile + c16 * kPRowBytes + (8 * g4) * 2);
      phi = plo;
    }

    // ---------------- O^T += V^T P^T ----------------------------------------------------
    // tail: zero V of tokens >= end_tok (their P is 0, but stale bytes may be NaN)
    const int vt0 = FP8 ? tok0 + 16 * g4 : tok0 + 8 * g4;   // first token of this lane's V vector
    const int vvalid = end_tok - vt0;                        // may be <= 0 or >= 16
    const bool tail = tok0 + STEP_TOK > end_tok;             // wave-uniform
#pragma unroll
    for (int m = 0; m < DT; ++m) {
      uint4 raw = vreg[m];
      if constexpr (FP8) {
        if (tail) raw = mask_bytes(raw, vvalid);
        uint4 lo, hi;
        widen16<scalar_t>(raw, lo, hi);
        o[m] = Mma<scalar_t>::mfma(lo, plo, o[m]);
        o[m] = Mma<scalar_t>::mfma(hi, phi, o[m]);
      } else {
        if (tail) raw = mask_halfs(raw, vvalid);
        o[m] = Mma<scalar_t>::mfma(raw, plo, o[m]);
      }
    }
    if constexpr (more) issue_v(vreg, pf);
  };
  const std::true_type MORE;
  const std::false_type LAST;

  if (wave < nsteps) { issue_k(kregA, wave); issue_v(vregA, wave); }
  // Retire q (requested before the block-table chain, long arrived) here, behind the join, where
  // hipcc drains the counter completely -- the first step waits for its V as well as its K,
  // once.  With q still pending in its model at the loop head, or retired by a counted wait
  // (vmcnt(16), which its loop analysis does not carry over the back edge), it drains EVERY
  // outstanding load, V included, in front of the first MFMA of every step; after a full drain
  // the loop waits for exactly the load each MFMA consumes (vmcnt(15) .. vmcnt(8)).
#pragma unroll
  for (int s = 0; s < QF; ++s) asm volatile("" ::"v"(qf[s].x), "v"(qf[s].y), "v"(qf[s].z), "v"(qf[s].w));
  static_assert(!DEEP, "the two-register-set variant was measured slower (profiles/r01_ab_attention.txt) and is gone");
  {
    // every step but the wave's last issues the loads of its successor unconditionally (no
    // branch around a load: hipcc then counts the outstanding loads exactly instead of draining
    // them all before the first MFMA of a step)
    int step = wave;
    for (; step + NW < nsteps; step += NW) do_step(kregA, vregA, step, step + NW, MORE);
    if (step < nsteps) do_step(kregA, vregA, step, step + NW, LAST);
  }

  VA_STAMP(2);
#ifdef VA_PROBES
  if constexpr ((VARIANT & 8) != 0) {   // ablation: no merge, every wave stores something
    out[((int64_t)qrow * num_heads + head0 + c16) * dr + wave * 32 + g4 * 8] = (scalar_t)(o[0][0] + l_run + m_run);
    return;
  }
#endif
  // ---------------- merge the NW waves ------------------------------------------------
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();   // P tiles are dead; LDS becomes the merge buffer
  float* om = reinterpret_cast<float*>(lds);                         // [NW][HEAD_SIZE][16]
  float* ml = om + NW * HEAD_SIZE * 16;                               // [NW][16][2]
  if (g4 == 0) {
    ml[(wave * 16 + c16) * 2 + 0] = m_run;
    ml[(wave * 16 + c16) * 2 + 1] = l_run;
  }
  __syncthreads();
  float m_tot = -FLT_MAX;
#pragma unroll
  for (int w = 0; w < NW; ++w) m_tot = fmaxf(m_tot, ml[(w * 16 + c16) * 2]);
  const float f_w = fast_exp2(m_run - m_tot);   // waves with no step: m_run = -FLT_MAX -> 0
#pragma unroll
  for (int m = 0; m < DT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      om[(wave * HEAD_SIZE + 16 * m + 4 * g4 + r) * 16 + c16] = o[m][r] * f_w;
  __syncthreads();

  // thread -> (q head qh, 8 consecutive d)
  constexpr int TPH = HEAD_SIZE / 8;             // threads per head row
  constexpr int HPP = NW * 64 / TPH;             // head rows per pass (head 256: 8, two passes over 16 q heads)
  const int qh0 = threadIdx.x / TPH;
  const int d0 = (threadIdx.x % TPH) * 8;
  for (int qh = qh0; qh < nheads && qh0 < HPP; qh += HPP) if (!PAD || d0 < dr) {
    // l_tot / m_tot of column qh live in lanes with c16 == qh; recompute from LDS
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
      for (int w = 0; w < NW; ++w) a += om[(w * HEAD_SIZE + d0 + j) * 16 + qh];
      acc[j] = a;
    }
    if (P.mode == 2) {
      // unnormalised partial (relative to mt) -> this piece's workspace slot, written through
      float* so = P.slot + qh * HEAD_SIZE + d0;
#pragma unroll
      for (int j = 0; j < 8; ++j) __hip_atomic_store(so + j, acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d0 == 0) {
        __hip_atomic_store(P.slot + 16 * HEAD_SIZE + qh, mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.slot + 16 * HEAD_SIZE + 16 + qh, lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const float inv = __fdividef(1.f, lt + 1e-6f) * vscale;
      uint4 pk;
      pk.x = Mma<scalar_t>::pack2(acc[0] * inv, acc[1] * inv);
      pk.y = Mma<scalar_t>::pack2(acc[2] * inv, acc[3] * inv);
      pk.z = Mma<scalar_t>::pack2(acc[4] * inv, acc[5] * inv);
      pk.w = Mma<scalar_t>::pack2(acc[6] * inv, acc[7] * inv);
      const int head = head0 + qh;
      int64_t oidx;
      if (P.mode == 1) {
        const int64_t sidx = ((int64_t)seq * num_heads + head) * A.max_parts + (int64_t)P.part * A.slot_stride;
        oidx = sidx * dr + d0;
        if (d0 == 0) {
          A.max_logits[sidx] = mt * kLn2;
          A.exp_sums[sidx] = lt;
        }
      } else {
        oidx = ((int64_t)qrow * num_heads + head) * dr + d0;
      }
      *reinterpret_cast<uint4*>(out + oidx) = pk;
    }
  }
  if (P.mode != 2) return;

  // ---- balanced launch: ticket; the sequence's last piece merges all of them ---------------
  // Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): every byte of a partial is an
  // agent-scope (sc1, write-through) store; every storing wave drains its stores; one lane adds
  // to the sequence's counter behind a workgroup barrier; the workgroup whose add came last
  // reads every partial with agent-scope (sc1) loads after a barrier that lane has joined.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  volatile int* flag = reinterpret_cast<volatile int*>(lds + LDS_BYTES);
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(P.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == (unsigned)(P.jl - P.jf);
    if (last) __hip_atomic_store(P.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  for (int qh = qh0; qh < nheads && qh0 < HPP; qh += HPP) if (!PAD || d0 < dr) {
    float M = -FLT_MAX, L = 0.f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // four pieces per round: their loads (independent, ~2 us each through the fabric) are all
    // issued before the first is combined; a long sequence cut into 20 pieces merges in 5 rounds
    for (int j0 = P.jf; j0 <= P.jl; j0 += 4) {
      float mp[4], lp[4], op[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = j0 + u <= P.jl ? j0 + u : P.jl;   // past the end: the last piece again, weight 0
        // workgroup jj's piece of this sequence: its first piece if its range starts inside the
        // sequence, else its last one
        const int64_t a_jj = (int64_t)jj * P.n_tot / P.gk;
        const float* sl = P.slots + (size_t)(2 * jj + (a_jj >= P.pre ? 0 : 1)) * pa_slot_floats<HEAD_SIZE>();
        mp[u] = __hip_atomic_load(sl + 16 * HEAD_SIZE + qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lp[u] = __hip_atomic_load(sl + 16 * HEAD_SIZE + 16 + qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          op[u][j] = __hip_atomic_load(sl + qh * HEAD_SIZE + d0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j0 + u > P.jl) break;
        const float mn = fmaxf(M, mp[u]);
        const float fo = fast_exp2(M - mn), fp = fast_exp2(mp[u] - mn);
        L = L * fo + lp[u] * fp;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * fo + op[u][j] * fp;
        M = mn;
      }
    }
    const float inv = __fdividef(1.f, L + 1e-6f) * vscale;
    uint4 pk;
    pk.x = Mma<scalar_t>::pack2(acc[0] * inv, acc[1] * inv);
    pk.y = Mma<scalar_t>::pack2(acc[2] * inv, acc[3] * inv);
    pk.z = Mma<scalar_t>::pack2(acc[4] * inv, acc[5] * inv);
    pk.w = Mma<scalar_t>::pack2(acc[6] * inv, acc[7] * inv);
    *reinterpret_cast<uint4*>(out + ((int64_t)qrow * num_heads + head0 + qh) * dr + d0) = pk;
  }
}

template <typename scalar_t, bool FP8, int HEAD_SIZE, int BLOCK_SIZE, int NW, int VARIANT, bool PAD, bool BAL>
__global__ __launch_bounds__(NW * 64, (HEAD_SIZE > 128 ? 1 : 2)) void pa_mfma_kernel(
    scalar_t* __restrict__ out, float* __restrict__ exp_sums, float* __restrict__ max_logits,
    const scalar_t* __restrict__ q, const uint8_t* __restrict__ k_cache,
    const uint8_t* __restrict__ v_cache, int num_heads, int num_kv_heads, float scale,
    const int* __restrict__ block_tables, const int* __restrict__ seq_lens,
    const int* __restrict__ query_start_loc, int max_num_blocks_per_seq,
    const float* __restrict__ alibi_slopes, int64_t q_stride, int64_t kv_block_bytes,
    int64_t kv_head_bytes, const float* __restrict__ k_scale_p,
    const float* __restrict__ v_scale_p, int partition_size, int max_parts, int gqa_chunks,
    int slot_stride_arg, int dr_arg, float* __restrict__ bal_ws, unsigned* __restrict__ bal_ctr,
    int num_seqs, int window, int order) {
  VA_STAMP_DECL;
  VA_STAMP(0);
  // (the pointers stay __restrict__ kernel parameters: as members of a by-value struct they lose
  // the no-alias guarantee, the block-table loads turn from scalar into vector loads, and every
  // K / V issue then waits behind the loads in flight)
  const PaK<scalar_t> A{out, exp_sums, max_logits, q, k_cache, v_cache, num_heads, num_kv_heads, scale,
                        block_tables, seq_lens, query_start_loc, max_num_blocks_per_seq, alibi_slopes,
                        q_stride, kv_block_bytes, kv_head_bytes, k_scale_p, v_scale_p, partition_size,
                        max_parts, gqa_chunks, slot_stride_arg, dr_arg, bal_ws, bal_ctr, num_seqs, window, order};
  constexpr int STEP_TOK = (FP8 ? 4 : 2) * 16;
  constexpr int P_BYTES = 16 * kPRowBytes;
  constexpr int MERGE_BYTES = NW * HEAD_SIZE * 16 * 4 + NW * 16 * 8;
  constexpr int LDS_BYTES = (NW * P_BYTES > MERGE_BYTES) ? NW * P_BYTES : MERGE_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES + 16];   // + the "last ticket" flag
  const int lane = threadIdx.x & 63;

  // logical id = ((seq * num_kv_heads + kvh) * gqa_chunks + chunk), blockIdx.y = partition
  int wg = blockIdx.x;
  int slot_stride = A.slot_stride;
  // Large launches: consecutive hardware ids go to different XCDs (id % 8); remapped so that the
  // kv heads of a sequence -- adjacent pages of the same blocks -- stream through one XCD's L2
  // (256 sequences x 8 kv heads x 4096 fp8: 330.6 -> 324.6 us; nothing at 512 workgroups)
  const int groups = A.gqa_chunks * A.num_kv_heads;
  int lpt_len = -1;
  if (!BAL && A.order > 0) {
    const bool xg = slot_stride < 0;
    wg = A.num_seqs <= 64    ? pa_lpt_wg<1>(A.seq_lens, A.num_seqs, groups, A.order, lane, wg, lpt_len, xg)
         : A.num_seqs <= 256 ? pa_lpt_wg<4>(A.seq_lens, A.num_seqs, groups, A.order, lane, wg, lpt_len, xg)
         : A.num_seqs <= 512 ? pa_lpt_wg<8>(A.seq_lens, A.num_seqs, groups, A.order, lane, wg, lpt_len, xg)
                             : pa_lpt_wg<16>(A.seq_lens, A.num_seqs, groups, A.order, lane, wg, lpt_len, xg);
  } else if (slot_stride < 0) {
    const int nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int chunk = wg % A.gqa_chunks;
  const int kvh = (wg / A.gqa_chunks) % A.num_kv_heads;
  PaPiece P;
  if constexpr (!BAL) {
    const int seq = wg / groups;
    // partitioned launches, order < 0: the LAST super-partitions go first.  The hardware hands out
    // blockIdx.y in rising order; the high partitions exist only for the longest sequences of a
    // ragged batch, so in rising order the launch ends on a few workgroups streaming alone, while
    // partition 0 -- the last to go in falling order -- exists for every sequence
    const int part = A.order < 0 ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
    P.seq = seq;
    P.qrow = seq;
    if (A.query_start_loc) {
      if (A.query_start_loc[seq + 1] - A.query_start_loc[seq] != 1) return;
      P.qrow = A.query_start_loc[seq];
    }
    P.seq_len = A.order > 0 ? lpt_len : A.seq_lens[seq];
    const bool partitioned = A.partition_size > 0;
    P.start_tok = partitioned ? part * A.partition_size : 0;
    if (partitioned && P.start_tok >= P.seq_len) return;
    P.end_tok = partitioned ? min(P.start_tok + A.partition_size, P.seq_len) : P.seq_len;
    if (A.window > 0) {
      // sliding window: pieces in front of it write nothing (the reduce kernel starts at the
      // window's first partition), the piece it starts in begins at the window's step
      P.first_tok = max(0, P.seq_len - A.window);
      if (partitioned && P.end_tok <= P.first_tok) return;
      P.start_tok = max(P.start_tok, P.first_tok / STEP_TOK * STEP_TOK);
    }
    P.mode = partitioned ? 1 : 0;
    P.part = part;
    PaK<scalar_t> B = A;
    B.slot_stride = slot_stride < 0 ? -slot_stride : slot_stride;
    pa_piece<scalar_t, FP8, HEAD_SIZE, BLOCK_SIZE, NW, VARIANT, PAD>(B, lds, P, kvh, chunk VA_STAMP_ARG);
  } else {
    // ---- balanced launch (v1, no partitions): the steps of all sequences, concatenated, are
    // cut into gk equal ranges per (kv head, q-head chunk); workgroup j takes range j, which
    // covers the end of one sequence, whole sequences, and the start of another.  Every
    // workgroup derives the same cut from seq_lens by itself (one wave-wide scan): no queue, no
    // atomics unless a sequence is shared.  A sequence of length 0 counts one (empty) step so
    // that somebody writes its zero output.
    const int S = A.num_seqs;
    auto nst = [&](int len) { const int n = (len + STEP_TOK - 1) / STEP_TOK; return n > 0 ? n : 1; };
    int n_tot = 0;
    for (int i = lane; i < S; i += 64) n_tot += nst(A.seq_lens[i]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) n_tot += __shfl_xor(n_tot, m, 64);
    n_tot = __builtin_amdgcn_readfirstlane(n_tot);
    const int gk_launch = gridDim.x / groups;
    const int gk = gk_launch < n_tot ? gk_launch : n_tot;   // never an empty range between two pieces
    const int j = wg / groups;
    if (j >= gk) return;
    const int ra = (int)((int64_t)j * n_tot / gk), rb = (int)((int64_t)(j + 1) * n_tot / gk);
    // the sequence that holds step ra, and the number of steps in front of it
    int s = 0, pre = 0;
    {
      int base = 0;
      for (int i0 = 0; i0 < S; i0 += 64) {
        const int n = (i0 + lane < S) ? nst(A.seq_lens[i0 + lane]) : 0;
        int incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d, 64);
          if (lane >= d) incl += t;
        }
        const int excl = incl - n;
        const unsigned long long mask = __ballot(base + excl <= ra && ra < base + incl);
        if (mask) {
          const int l = __ffsll((long long)mask) - 1;
          s = i0 + l;
          pre = base + __shfl(excl, l, 64);
          break;
        }
        base += __shfl(incl, 63, 64);
      }
      s = __builtin_amdgcn_readfirstlane(s);
      pre = __builtin_amdgcn_readfirstlane(pre);
    }
    const float* slots = A.bal_ws + (size_t)(kvh * A.gqa_chunks + chunk) * gk_launch * 2 * pa_slot_floats<HEAD_SIZE>();
    int cur = ra;
    while (cur < rb) {
      const int len = A.seq_lens[s];
      const int n_s = nst(len);
      const int ls0 = cur - pre, ls1 = min(rb - pre, n_s);
      P.seq = s;
      P.qrow = s;
      P.seq_len = len;
      P.start_tok = ls0 * STEP_TOK;
      P.end_tok = min(ls1 * STEP_TOK, len);
      P.part = 0;
      if (ls0 == 0 && ls1 == n_s) {
        P.mode = 0;
      } else {
        P.mode = 2;
        P.jf = (int)((((int64_t)pre + 1) * gk - 1) / n_tot);
        P.jl = (int)((((int64_t)pre + n_s) * gk - 1) / n_tot);
        P.pre = pre;
        P.n_tot = n_tot;
        P.gk = gk;
        P.slots = slots;
        P.slot = const_cast<float*>(slots) + (size_t)(2 * j + (cur == ra ? 0 : 1)) * pa_slot_floats<HEAD_SIZE>();
        P.ticket = A.bal_ctr + ((size_t)s * A.num_kv_heads + kvh) * A.gqa_chunks + chunk;
      }
      pa_piece<scalar_t, FP8, HEAD_SIZE, BLOCK_SIZE, NW, VARIANT, PAD>(A, lds, P, kvh, chunk VA_STAMP_ARG);
      __syncthreads();   // the next piece's P tiles overlay the merge buffers this one still reads
      cur = pre + ls1;
      pre += n_s;
      ++s;
    }
  }
  VA_STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------
bool paged_attention_mfma_supported(const PagedAttnParams& p) {
  if (p.dtype != VLLM_AMD_F16 && p.dtype != VLLM_AMD_BF16) return false;
  // head sizes up to 256 in whole 16-byte chunks of K and of the output (padded to 64 / 128 / 256)
  if (p.head_size < 16 || p.head_size > 256 ||
      p.head_size % (p.kv_dtype == VLLM_AMD_KV_AUTO ? 8 : 16) != 0)
    return false;
  if (p.block_size != 16 && p.block_size != 32) return false;
  if (p.bs_vert_stride > 1) return false;                 // block-sparse -> generic
  if (p.partition_size > 0 && p.partition_size % 64 != 0) return false;
  const int es = p.kv_dtype == VLLM_AMD_KV_AUTO ? 2 : 1;
  // natural contiguous page layout per (block, head)
  if (p.kv_head_stride != (int64_t)p.head_size * p.block_size) return false;
  if ((p.q_stride * 2) % 16 != 0 || (((uintptr_t)p.query) & 15) != 0) return false;
  if ((((uintptr_t)p.out) & 15) != 0) return false;
  if (p.tmp_out && (((uintptr_t)p.tmp_out) & 15) != 0) return false;
  (void)es;
  return true;
}

static int g_pa_variant = -1;
static int pa_variant() {
  if (g_pa_variant < 0) {
    const char* e = getenv("VLLM_AMD_PA_VARIANT");
    g_pa_variant = e ? atoi(e) & 3 : kDefaultVariant;
  }
  return g_pa_variant;
}
#ifdef VA_PROBES
void set_pa_variant(int v) { g_pa_variant = v & 15; }
#else
void set_pa_variant(int v) { g_pa_variant = v & 3; }
#endif
static int g_pa_sp = 0;
void set_pa_sp(int v) { g_pa_sp = v; }
static int g_pa_group = 0;
void set_pa_group(int v) { g_pa_group = v; }

#define PA_ARGS(A)                                                                                 \
  (A).out, (A).exp_sums, (A).max_logits, (A).q, (A).k_cache, (A).v_cache, (A).num_heads,           \
      (A).num_kv_heads, (A).scale, (A).block_tables, (A).seq_lens, (A).query_start_loc,            \
      (A).max_num_blocks_per_seq, (A).alibi_slopes, (A).q_stride, (A).kv_block_bytes,              \
      (A).kv_head_bytes, (A).k_scale_p, (A).v_scale_p, (A).partition_size, (A).max_parts,          \
      (A).gqa_chunks, (A).slot_stride, (A).dr_arg, (A).bal_ws, (A).bal_ctr, (A).num_seqs, (A).window, \
      (A).order

static int g_pa_order = 1;   // tuning "pa_order": 0 = hardware order for unpartitioned launches
void set_pa_order(int v) { g_pa_order = v; }
bool pa_order_enabled() { return g_pa_order != 0; }
static int g_pa_balance = 0;   // tuning "pa_balance": 1 = balanced v1 launches whenever possible, else never
void set_pa_balance(int v) { g_pa_balance = v; }
int get_pa_balance() { return g_pa_balance; }

template <typename scalar_t, bool FP8, int HS, int BS, int VARIANT, bool PAD>
static int launch_mfma_v(const PagedAttnParams& p) {
  constexpr int NW = 4;
  const int G = p.num_heads / p.num_kv_heads;
  const int chunks = ceil_div(G, 16);
  const int es = FP8 ? 1 : 2;
  const bool part = p.partition_size > 0;
  const int sp = part ? p.sp_parts : 1;   // reference partitions per super-partition
  dim3 grid((unsigned)(p.num_seqs * p.num_kv_heads * chunks),
            part ? ceil_div(p.max_num_partitions, sp) : 1);
  void* out = part ? p.tmp_out : p.out;
  // "pa_group": 0 = heuristic, 1 = always, 2 = never
  const bool group = g_pa_group == 1 || (g_pa_group == 0 && p.num_kv_heads >= 2 && grid.x >= 1024);
  PaK<scalar_t> A;
  A.out = (scalar_t*)out; A.exp_sums = p.exp_sums; A.max_logits = p.max_logits;
  A.q = (const scalar_t*)p.query; A.k_cache = (const uint8_t*)p.key_cache;
  A.v_cache = (const uint8_t*)p.value_cache; A.num_heads = p.num_heads;
  A.num_kv_heads = p.num_kv_heads; A.scale = p.scale; A.block_tables = p.block_tables;
  A.seq_lens = p.seq_lens; A.query_start_loc = p.query_start_loc;
  A.max_num_blocks_per_seq = p.max_num_blocks_per_seq; A.alibi_slopes = p.alibi_slopes;
  A.q_stride = p.q_stride; A.kv_block_bytes = p.kv_block_stride * es;
  A.kv_head_bytes = p.kv_head_stride * es; A.k_scale_p = p.k_scale; A.v_scale_p = p.v_scale;
  A.partition_size = p.partition_size * sp; A.max_parts = p.max_num_partitions;
  A.gqa_chunks = chunks; A.slot_stride = group ? -sp : sp; A.dr_arg = p.head_size;
  A.bal_ws = nullptr; A.bal_ctr = nullptr; A.num_seqs = p.num_seqs; A.window = p.window;
  // longest-first order (pa_lpt_wg): launches of one piece per sequence (v1, or v2 / rocm when one
  // super-partition covers max_seq_len) over <= 1024 sequences, of 257..1024
  // workgroups (resident or nearly so: long + short paired per CU) or XCD-grouped (several rounds:
  // plain longest first); "pa_order" = 0 turns it off
  A.order = (part && g_pa_order != 0) ? -1 : 0;
  if ((!part || grid.y == 1) && g_pa_order != 0 && grid.x > 256 && p.num_seqs <= kLptMaxSeqs && p.max_seq_len > 0 &&
      (group ? p.num_seqs % 8 == 0 : grid.x <= 1024)) {
    // keys = lengths clamped to `bits` bits; with a sliding window a workgroup streams at most
    // `window` keys, so longer sequences rank (about) equal
    const int ml = (p.window > 0 && p.window < p.max_seq_len) ? p.window : p.max_seq_len;
    int bits = 1;
    while (bits < 30 && (ml >> bits) != 0) ++bits;
    A.order = bits;
  }

  // Balanced v1 launch (opt-in, "pa_balance" = 1): paged_attention_v1 has no scratch arguments,
  // so with one workgroup per (sequence, kv head) a skewed batch waits for its longest sequence.
  // With a registered workspace (vllm_amd_set_workspace) the concatenated steps of all sequences
  // are cut into equal ranges, two workgroups per CU; a sequence shared by several workgroups is
  // merged through the workspace by whoever finishes last (kernel above).  Measured
  // (profiles/r02_decode_balanced.txt): the merge of a shared sequence costs its last finisher
  // 5-7 us at the end of the launch, so it pays on skewed batches (one long sequence among many
  // short ones) and loses 2-7 us on uniform and mildly ragged ones -- the host cannot tell the
  // two apart (seq_lens live on the device), hence not the default.
  if constexpr (HS <= 128)   // (head 256: one workgroup per CU, not instantiated)
  if (!part && !p.query_start_loc && p.window == 0 && g_pa_balance == 1) {
    const int groups = p.num_kv_heads * chunks;
    const int64_t steps_ub = (int64_t)p.num_seqs * ceil_div(p.max_seq_len > 0 ? p.max_seq_len : 1, FP8 ? 64 : 32);
    int gk = 512 / groups;
    if (gk > steps_ub) gk = (int)steps_ub;
    WsSlot slot;
    const size_t need = (size_t)groups * (gk > 0 ? gk : 1) * 2 * pa_slot_floats<HS>() * sizeof(float);
    if (gk >= 1 && (int64_t)p.num_seqs * groups <= (int64_t)WS_CTR_PA_WORDS &&
        workspace_slot(p.stream, &slot) && need <= slot.data_bytes) {
      A.bal_ws = (float*)slot.data;
      A.bal_ctr = slot.ctr_pa;
      A.slot_stride = 1;
      hipLaunchKernelGGL((pa_mfma_kernel<scalar_t, FP8, HS, BS, NW, VARIANT, PAD, true>),
                         dim3((unsigned)(groups * gk)), dim3(NW * 64), 0, p.stream, PA_ARGS(A));
      VA_LAUNCH_CHECK("paged_attention mfma launch (balanced)");
      return VLLM_AMD_OK;
    }
  }
  hipLaunchKernelGGL((pa_mfma_kernel<scalar_t, FP8, HS, BS, NW, VARIANT, PAD, false>), grid, dim3(NW * 64), 0,
                     p.stream, PA_ARGS(A));
  VA_LAUNCH_CHECK("paged_attention mfma launch");
  if (part) return paged_attention_reduce(p);
  return VLLM_AMD_OK;
}

template <typename scalar_t, bool FP8, int HS, int BS>
static int launch_mfma(const PagedAttnParams& p) {
  // bit 1 (two register sets in flight) measured 1% slower than bit 0 alone on
  // configs[1] (profiles/r01_ab_attention.txt) and is not instantiated
  const bool nt = (pa_variant() & 1) != 0;
#ifdef VA_PROBES
  if constexpr (FP8 && HS == 128 && BS == 16)
    if ((pa_variant() & 4) && p.head_size == HS) return launch_mfma_v<scalar_t, FP8, HS, BS, 5, false>(p);
#endif
  if (p.head_size != HS)   // padded head sizes: non-temporal loads only
    return launch_mfma_v<scalar_t, FP8, HS, BS, 1, true>(p);
  return nt ? launch_mfma_v<scalar_t, FP8, HS, BS, 1, false>(p)
            : launch_mfma_v<scalar_t, FP8, HS, BS, 0, false>(p);
}

template <typename scalar_t>
static int dispatch_mfma(const PagedAttnParams& p) {
  const bool fp8 = p.kv_dtype == VLLM_AMD_KV_FP8_E4M3;
  const int hs = p.head_size <= 64 ? 64 : p.head_size <= 128 ? 128 : p.head_size == 192 ? 192 : 256;   // padded
#define L(HS, BS) return fp8 ? launch_mfma<scalar_t, true, HS, BS>(p) : launch_mfma<scalar_t, false, HS, BS>(p)
  if (hs == 128 && p.block_size == 16) L(128, 16);
  if (hs == 128 && p.block_size == 32) L(128, 32);
  if (hs == 64 && p.block_size == 16) L(64, 16);
  if (hs == 64 && p.block_size == 32) L(64, 32);
  // head sizes 136 .. 256 (the reference's 192 and 256): the registers of a step double, one workgroup per CU
  if (hs == 192 && p.block_size == 16) L(192, 16);   // (its own width: padded to 256 a quarter of the K / V / Q slots idle, 73 %)
  if (hs == 192 && p.block_size == 32) L(192, 32);
  if (hs == 256 && p.block_size == 16) L(256, 16);
  if (hs == 256 && p.block_size == 32) L(256, 32);
#undef L
  VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "paged_attention mfma: unsupported shape");
}

// Super-partitions: the reference cuts every sequence into 512- (v2) or 256-token (rocm)
// partitions, one workgroup each.  A 128 KiB workgroup is all prologue and merge on this
// chip (5.4 TB/s); the same kernel with ~1 MiB per workgroup streams at 6.4-6.5 TB/s
// (profiles/r01_ab_attention.txt).  So a workgroup takes `sp` consecutive reference
// partitions (heuristic below), writes ONE
// partial into the first of its scratch slots, and the reduce kernel visits every sp-th
// slot.  The scratch tensors are caller-allocated temporaries of the reference's shape;
// only the slots we read are written.
int paged_attention_mfma(const PagedAttnParams& p0) {
  PagedAttnParams p = p0;
  p.sp_parts = 1;
  if (p.partition_size > 0) {
    const int chunks = ceil_div(p.num_heads / p.num_kv_heads, 16);
    // The chip holds 512 of these workgroups (two per CU) and hands further ones out as others
    // finish; the host does not know the batch's lengths (seq_lens live on the device).
    const int64_t base = (int64_t)p.num_seqs * p.num_kv_heads * chunks;
    const int64_t P = p.max_num_partitions;
    const int64_t slots = p.head_size > 128 ? 256 : 512;   // head sizes above 128: one workgroup per CU
    int64_t sp;
    if (base < slots) {
      // Fewer (sequence, head) pairs than workgroup slots: the sequences must be cut to fill the
      // chip, and real batches are ragged, so aim at ~2048 workgroups of >= 2048 tokens -- the
      // hardware hands them out as others finish (ragged 16 x U[8k, 64k] tokens: 215 us, against
      // 239 us with the 512 longest pieces that a uniform batch would like best).
      const int64_t total = base * P;
      const int64_t sp_fill = (total + slots - 1) / slots, sp_bal = (total + 4 * slots - 1) / (4 * slots);
      const int64_t sp_min = p.partition_size >= 2048 ? 1 : 2048 / p.partition_size;
      sp = sp_fill < sp_min ? sp_fill : sp_min;
      if (sp < sp_bal) sp = sp_bal;
    } else {
      // Enough pairs to fill the chip: few long pieces stream best (256 x 16k tokens: one piece per
      // sequence 81.3 % of peak, sp 16 80.3 %, sp 8 78.5 %, sp 4 76.0 %) and a ragged batch is
      // balanced by the longest-first order of one-piece launches (pa_lpt_wg; 64 x U[2k, 16k]:
      // 2048-token pieces 218 us, one piece 209 us; 64 x U[8k, 64k]: 755 -> 725 us).  What is left is
      // the quantisation into rounds of 512 workgroups: with all lengths at max_seq_len a launch of
      // `pieces` per pair costs ceil(base pieces / 512) rounds of (sp partition_size + ~256 tokens
      // of prologue and merge); take the cheapest, the longest pieces within 3 %, none below 2048
      // tokens.  profiles/r02_decode_order.txt §7.
      sp = P;
      double best = 1e300;
      for (int64_t c = P; c >= 1 && c * p.partition_size >= 2048; --c) {
        const int64_t pieces = (P + c - 1) / c, rounds = (base * pieces + slots - 1) / slots;
        const double cost = (double)rounds * (double)(c * p.partition_size + 256);
        if (cost < best * 0.97) { best = cost; sp = c; }
      }
    }
    if (g_pa_sp > 0) sp = g_pa_sp;
    if (sp < 1) sp = 1;
    if (sp > p.max_num_partitions) sp = p.max_num_partitions;
    p.sp_parts = (int)sp;
  }
  if (p.dtype == VLLM_AMD_F16) return dispatch_mfma<f16_t>(p);
  return dispatch_mfma<bf16_t>(p);
}

}  // namespace va

#ifdef VA_PROBES
extern "C" int vllm_amd_probe_stamps(unsigned long long* dst, int nwg) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(va::g_pa_stamps), (size_t)nwg * 4 * 8) == hipSuccess ? 0 : -3;
}
#endif
