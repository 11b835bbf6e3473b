// Prefill / chunked-prefill attention with a paged prefix: the MFMA-bound
// sibling of the decode kernel.  Replaces context_attention_fwd
// (vllm/attention/ops/prefix_prefill.py:726-913; Triton kernels _fwd_kernel :36-288 and
// _fwd_kernel_alibi :477-722): every query token of a sequence attends to
//   (1) the sequence's cached context in the PAGED KV cache (no causal mask), then
//   (2) the new tokens' dense K / V (causal mask),
// with one online softmax over both.
//
// One workgroup (4 waves) per (128 query rows, head, sequence); a wave owns 32 rows = two
// 16-column tiles of the TRANSPOSED problem, exactly as in the decode kernel:
//   S^T[n, m] = K[n, :] . Q[m, :]      (A = K rows from LDS, B = Q^T from registers)
//   O^T[d, m] += V^T[d, n] . P^T[n, m] (A = V^T rows from LDS, B = P^T)
// The C layout of S^T (lane holds 4 consecutive n of one m) IS the B layout of P^T once
// the contraction index of the PV product is permuted (slots 0-3 <- tile a, 4-7 <- tile b,
// the V^T fragment is read with the same permutation): P never leaves the registers.
// KV tiles of 64 tokens go global -> registers -> LDS (converted on the way: fp8 context is
// scaled and rounded to the query dtype like the reference, new V is transposed), with the
// next tile's global loads in flight during the MFMAs.
#include "mma.cuh"
#include "unified_decode.cuh"
#include "prefill_fa.cuh"

#include <float.h>
#include <type_traits>

namespace va {

constexpr int PF_BN = 64;
int g_pf_rows = 0;   // tuning "pf_rows": query rows per workgroup (0 = default)
int g_uni_decode = 1;   // tuning "uni_decode": 0 = decode-only launches take the packed tile kernel
extern int g_pf_fa_stagger;   // prefill_fa.hip
int g_pf_fa_abl = 0;    // tuning "pf_fa_abl": measurement only, see prefill_fa.cuh
int g_pf_fa = 0;        // tuning "pf_fa": 1 = unified_attention never takes the 256-row kernel (prefill_fa.hip), 2 = whenever it can

struct PrefillParams {
  void* o; const void* q; const void* k; const void* v;
  const uint8_t* k_cache; const uint8_t* v_cache;
  const int* b_loc; const int* b_start_loc; const int* b_seq_len;
  const float* k_scale; const float* v_scale; const float* alibi_slopes;
  const float* fp8_out_scale;
  int num_heads, num_kv_heads;
  int head_size;   // real head size; the kernel template is padded to a multiple of 32
  int64_t b_loc_stride, q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h,
      o_stride_t, o_stride_h, kc_block_bytes, kc_head_bytes, vc_block_bytes, vc_head_bytes;
  float sm_scale; int sliding_window; int skip_decode;
  // unified_attention only (UNI): flash-layout cache [block][slot][head][d], softcap
  int64_t kc_page_bytes, vc_page_bytes;
  float softcap;
  // split over the keys (PACK launches with one row block): segment partials, caller-allocated
  float* segm_out; float* segm_max; float* segm_sum;
  int num_segments;     // segments of this launch (1 = no split)
  int bs_shift;         // UNI: log2(block_size) (any power of two >= 16; the BS template value is unused)
  int flat_batch;       // > 0 = number of sequences: grid.x enumerates the row blocks of all of them,
  int flat_tokens;      //   at most flat_tokens / tokens_per_block + flat_batch (flat_tokens = rows of q)
  int segm_stride;      // segments the buffers were allocated for
};

// transposing 8-byte LDS read (4 x 16 bit), see compute_tile
__device__ __forceinline__ uint2 lds_read_tr16(const uint8_t* p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// MT = 16-row query tiles per wave: 2 -> 4 waves of 32 rows (each K / V fragment feeds two
// MFMAs), 1 -> 8 waves of 16 rows (half the registers per wave: no spills at head size 128).
// NW = waves per workgroup: the workgroup owns PF_BM = 16 MT NW query rows.
//
// UNI = unified_attention (vllm/attention/ops/triton_unified_attention.py:52-272, 588-700):
// there are no separate new-token K / V -- every key comes from the cache, which is in the
// FLASH layout [block][slot][kv head][d] -- and the causal bound is position-based
// (key n visible to query m iff n <= ctx + m).  Every KV tile is staged like the new tokens of
// the prefix kernel (K and V row-major, V read with the transposing LDS load).
// PACK (with UNI): the rows of a workgroup's query tile are (token, q head of ONE kv head's GQA
// group) pairs, token-major -- PF_BM / group tokens x the group, as the reference packs
// BLOCK_M / num_queries_per_kv tokens (:612-613) -- instead of PF_BM tokens of one head;
// grid.y = kv heads.  A decode token then streams its K / V once for the whole group (in a
// batch that mixes prefill chunks and decodes, too).
template <typename T, bool FP8, int D, int BS, int MT, int NW, bool UNI = false, bool PACK = false>
__global__ __launch_bounds__(64 * NW, (D > 128 ? 1 : 2)) void prefill_kernel(const PrefillParams p) {
  static_assert(!PACK || UNI, "");
  constexpr int PF_THREADS = 64 * NW, PF_BM = 16 * MT * NW;
  static_assert(D % 32 == 0 && (D <= 128 || (D == 256 && MT == 1 && NW == 4)), "");   // head sizes 136 .. 256: one configuration
  // D = head size padded to a multiple of 32: chunks at d >= dr are staged as zeros (K, Q: no
  // contribution to the scores; V: output columns that are never stored)
  const int dr = p.head_size;
  constexpr int KCH = D / 8;                 // 16-byte chunks per K row (8 elements each)
  constexpr int KROW = D * 2;                // bytes per K row in LDS
  constexpr int KSWZ = (KCH % 16 == 0) ? 15 : (KCH % 8 == 0) ? 7 : 3;   // chunk ^= tok & KSWZ
  constexpr int K_BYTES = PF_BN * KROW;
  constexpr int VROW = PF_BN * 2;            // bytes per V^T row (64 tokens)
  constexpr int V_BYTES = D * VROW;
  constexpr int QK_STEPS = D / 32, DT = D / 16;
  // staging work per thread
  constexpr int K16_TOTAL = PF_BN * KCH, K8_TOTAL = PF_BN * (D / 16), V16_TOTAL = D * 8, V8_TOTAL = D * 4;
  constexpr int K16_ITEMS = (K16_TOTAL + PF_THREADS - 1) / PF_THREADS;   // 16-bit K: (tok, chunk)
  constexpr int K8_ITEMS = (PF_BN * (D / 16) + PF_THREADS - 1) / PF_THREADS;   // fp8 K
  constexpr int V16_ITEMS = (V16_TOTAL + PF_THREADS - 1) / PF_THREADS;   // 16-bit V: (d, 8 tok)
  constexpr int V8_ITEMS = (D * 4 + PF_THREADS - 1) / PF_THREADS;   // fp8 V: (d, 16 tok)
  // staging registers per operand (>= every count; unified_attention over an fp8 cache never
  // stages 16-bit items)
  constexpr int NSTG = (UNI && FP8) ? K8_ITEMS : K16_ITEMS;
  static_assert(K16_ITEMS >= K8_ITEMS && K16_ITEMS >= V16_ITEMS && K16_ITEMS >= V8_ITEMS, "");
  __shared__ __attribute__((aligned(16))) uint8_t lds[K_BYTES + V_BYTES];
  uint8_t* lds_k = lds;
  uint8_t* lds_v = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  // Workgroup order.  Consecutive hardware ids go to different XCDs (id % 8), each with its own
  // L2; the row blocks and GQA heads that read the SAME K/V must share one, or every XCD streams
  // the same K/V from HBM.  The bijective remap gives each XCD a contiguous range of logical
  // ids; within it row blocks vary fastest, then heads, then sequences (+54 % on pure prefill).
  // Causal: later row blocks have more KV tiles and start first.
  int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  // split launches (PACK, one row block per sequence): blockIdx.x is the key segment
  const bool split = PACK && p.num_segments > 1;
  const int seg = split ? wg % gridDim.x : 0;
  const int gqa = p.num_heads / p.num_kv_heads;
  const int tpb = PACK ? PF_BM / gqa : PF_BM;   // query tokens per workgroup
  const int head = (wg / gridDim.x) % gridDim.y;
  int start_m = split ? 0 : gridDim.x - 1 - wg % gridDim.x;
  int seq = wg / (gridDim.x * gridDim.y);
  if (!split && p.flat_batch > 0) {
    // grid.x enumerates the row blocks of ALL sequences (cdiv(num_tokens, tpb) + num_seqs is an
    // upper bound, as in the reference, :612-626), so a mixed batch -- one long prefill chunk
    // among decode tokens -- launches no empty workgroups for the short sequences; the
    // sequence is the last one whose first block cu[i] / tpb + i is <= the block id
    // (find_seq_idx, :33-48)
    const int g = start_m;
    int lo = 0, hi = p.flat_batch;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (p.b_start_loc[mid] / tpb + mid <= g) lo = mid; else hi = mid;
    }
    seq = lo;
    start_m = g - (p.b_start_loc[seq] / tpb + seq);
  }
  const int kvh = PACK ? head : head / gqa;
  // global row id m = start_m PF_BM + local row -> (query token within the sequence's new
  // tokens, q head); PACK: local row = token-major (token, head in group)
  auto row_tok = [&](int m) { return PACK ? start_m * tpb + (m - start_m * PF_BM) / gqa : m; };
  auto row_head = [&](int m) { return PACK ? kvh * gqa + (m - start_m * PF_BM) % gqa : head; };

  const int seq_len = p.b_seq_len[seq];
  const int tok0 = p.b_start_loc[seq];
  const int qlen = p.b_start_loc[seq + 1] - tok0;
  const int ctx = seq_len - qlen;
  if (p.skip_decode && qlen == 1) return;
  if (start_m * tpb >= qlen) return;   // the reference computes and stores nothing here
  auto row_valid = [&](int m) {
    return PACK ? (m - start_m * PF_BM < tpb * gqa && row_tok(m) < qlen) : m < qlen;
  };

  const bool alibi = p.alibi_slopes != nullptr;
  const float slope2 = (alibi && !PACK) ? p.alibi_slopes[head] * kLog2e : 0.f;
  // the prefix kernel's alibi variant has no window; unified_attention applies both
  const int window = (alibi && !UNI) ? 0 : p.sliding_window;
  const float softcap = UNI ? p.softcap : 0.f;
  const float sc2 = p.sm_scale * kLog2e;
  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *p.k_scale; vscale = *p.v_scale; }
  const float vscale_log2 = log2f(vscale), vscale_inv = 1.0f / vscale;

  // ---- Q^T fragments: column m = c16 of the wave's tile mt, k = d = 32 ks + 8 g4 + j ------
  const int m_wave = start_m * PF_BM + 16 * MT * wave;
  int rtok[MT];    // this lane's rows: query token ...
  bool rok[MT];    // ... and whether the row is real
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    rtok[mt] = row_tok(m_wave + 16 * mt + c16);
    rok[mt] = row_valid(m_wave + 16 * mt + c16);
  }
  uint4 qf[MT][QK_STEPS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_wave + 16 * mt + c16;
    const T* qp = (const T*)p.q + (int64_t)(tok0 + (rok[mt] ? rtok[mt] : 0)) * p.q_stride_t +
                  (int64_t)row_head(m) * p.q_stride_h;
#pragma unroll
    for (int ks = 0; ks < QK_STEPS; ++ks) {
      const int d0 = 32 * ks + 8 * g4;
      const uint4 v = *reinterpret_cast<const uint4*>(qp + (d0 < dr ? d0 : 0));
      qf[mt][ks] = (rok[mt] && d0 < dr) ? v : make_uint4(0, 0, 0, 0);
    }
  }

  float slope_row[MT];   // PACK: one ALiBi slope per tile row (= q head)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    slope_row[mt] = (PACK && alibi) ? p.alibi_slopes[row_head(m_wave + 16 * mt + c16)] * kLog2e : 0.f;
  float m_run[MT], l_run[MT];
  f32x4 o[DT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    m_run[mt] = -INFINITY;
    l_run[mt] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i][mt] = (f32x4){0, 0, 0, 0};
  }

  const int new_end = min((start_m + 1) * tpb, qlen);   // causal bound of this row block
  // UNI: no context tiles in the prefix kernel's sense; tiles cover keys [0, ctx + new_end)
  const int ctx_tiles = UNI ? 0 : (ctx + PF_BN - 1) / PF_BN;
  const int new_tiles = ((UNI ? ctx + new_end : new_end) + PF_BN - 1) / PF_BN;
  const int ntiles = ctx_tiles + new_tiles;
  const int* bl = p.b_loc + (int64_t)seq * p.b_loc_stride;

  // staging registers of the tile being loaded (native vectors: HIP's uint4 struct arrays were
  // demoted to scratch memory by the compiler, with a wait on every load right behind it)
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  u32x4 kst[NSTG], vst[NSTG];
  auto as_u4 = [](u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); };

  // Physical block numbers a thread needs for a context tile.  Which page of the tile an item
  // reads is fixed per thread (K: every item of a thread is the same token; V: one page per
  // item), so each thread fetches exactly those block-table entries -- ONE TILE EARLIER than
  // the K/V loads that use them, so no load of the steady state waits on a dependent
  // block-table read.  Pages past the context are clamped to its last page (masked tokens).
  constexpr int PPT = PF_BN / BS;   // pages per tile
  constexpr int VPG = FP8 ? V8_ITEMS : V16_ITEMS;
  struct Pages { int k; int v[VPG]; };
  const int last_page = UNI ? (seq_len > 0 ? (seq_len - 1) >> p.bs_shift : 0) : (ctx > 0 ? (ctx - 1) / BS : 0);
  auto v_page_of_item = [&](int i) {   // page (within the tile) of V item i of this thread
    constexpr int CPP = FP8 ? BS / 16 : BS / 8;
    return min((tid + PF_THREADS * i) / (CPP * D), PPT - 1);
  };
  auto load_pages = [&](int t) {
    Pages pg;
    const bool on = UNI ? t < ntiles : t < ctx_tiles;
    if constexpr (UNI) pg.k = on ? bl[min((t * PF_BN + (tid & 63)) >> p.bs_shift, last_page)] : 0;
    else pg.k = on ? bl[min(t * PPT + (tid & 63) / BS, last_page)] : 0;
    static_for<VPG>([&](auto ic_) {
      constexpr int i = decltype(ic_)::value;
      pg.v[i] = (on && !UNI) ? bl[min(t * PPT + v_page_of_item(i), last_page)] : 0;
    });
    return pg;
  };

  // ---- global -> registers: raw loads only (masks and conversions happen in store_tile, so
  // nothing here waits for the data) ---------------------------------------------------------
  auto load_tile = [&](int t, const Pages& pg) {
    if constexpr (UNI) {
      // every item of a thread is the same token (PF_THREADS is a multiple of 64): one page
      const int slot = (t * PF_BN + (tid & 63)) & ((1 << p.bs_shift) - 1);
      const uint8_t* krow = p.k_cache + (int64_t)pg.k * p.kc_block_bytes +
                            (int64_t)slot * p.kc_page_bytes + (int64_t)kvh * p.kc_head_bytes;
      const uint8_t* vrow = p.v_cache + (int64_t)pg.k * p.vc_block_bytes +
                            (int64_t)slot * p.vc_page_bytes + (int64_t)kvh * p.vc_head_bytes;
      if constexpr (!FP8) {
        static_for<K16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;            // item -> (tok, 8-element chunk)
          const int item = min(tid + PF_THREADS * i, K16_TOTAL - 1);
          const int c = min(item >> 6, dr / 8 - 1);
          kst[i] = *reinterpret_cast<const u32x4*>(krow + 16 * c);
          vst[i] = *reinterpret_cast<const u32x4*>(vrow + 16 * c);
        });
      } else {
        static_for<K8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;            // item -> (tok, 16-element chunk)
          const int cc = min((tid + PF_THREADS * i) >> 6, dr / 16 - 1);
          kst[i] = *reinterpret_cast<const u32x4*>(krow + 16 * cc);
          vst[i] = *reinterpret_cast<const u32x4*>(vrow + 16 * cc);
        });
      }
    } else if (t < ctx_tiles) {
      if constexpr (!FP8) {
        static_for<K16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;          // item -> (tok = item & 63, chunk)
          const int item = min(tid + PF_THREADS * i, K16_TOTAL - 1);
          const int tok = item & 63, c = min(item >> 6, dr / 8 - 1);
          const uint8_t* src = p.k_cache + (int64_t)pg.k * p.kc_block_bytes +
                               (int64_t)kvh * p.kc_head_bytes + ((c * BS + tok % BS) << 4);
          kst[i] = *reinterpret_cast<const u32x4*>(src);
        });
        static_for<V16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;          // item -> (page, d, 8-token chunk of the page)
          const int item = min(tid + PF_THREADS * i, V16_TOTAL - 1);
          constexpr int CPP = BS / 8;
          const int j = item % CPP, d = min((item / CPP) % D, dr - 1);
          const uint8_t* src = p.v_cache + (int64_t)pg.v[i] * p.vc_block_bytes +
                               (int64_t)kvh * p.vc_head_bytes + ((d * BS + 8 * j) << 1);
          vst[i] = *reinterpret_cast<const u32x4*>(src);
        });
      } else {
        static_for<K8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;           // item -> (tok, 16-element chunk)
          const int item = tid + PF_THREADS * i;
          const int tok = item & 63, cc = min(item >> 6, dr / 16 - 1);
          const uint8_t* src = p.k_cache + (int64_t)pg.k * p.kc_block_bytes +
                               (int64_t)kvh * p.kc_head_bytes + ((cc * BS + tok % BS) << 4);
          kst[i] = *reinterpret_cast<const u32x4*>(src);
        });
        static_for<V8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;           // item -> (page, d, 16-token chunk)
          const int item = tid + PF_THREADS * i;
          constexpr int CPP = BS / 16;
          const int j = item % CPP, d = min((item / CPP) % D, dr - 1);
          const uint8_t* src = p.v_cache + (int64_t)pg.v[i] * p.vc_block_bytes +
                               (int64_t)kvh * p.vc_head_bytes + (d * BS + 16 * j);
          vst[i] = *reinterpret_cast<const u32x4*>(src);
        });
      }
    } else {
      const int n0 = (t - ctx_tiles) * PF_BN;
      static_for<K16_ITEMS>([&](auto ic_) {
        constexpr int i = decltype(ic_)::value;            // item -> (tok, chunk) for K and for V
        const int item = min(tid + PF_THREADS * i, K16_TOTAL - 1);
        const int tok = item & 63, c = min(item >> 6, dr / 8 - 1);
        const int64_t row = tok0 + min(n0 + tok, qlen - 1);
        kst[i] = *reinterpret_cast<const u32x4*>((const T*)p.k + row * p.k_stride_t +
                                                 (int64_t)kvh * p.k_stride_h + 8 * c);
        vst[i] = *reinterpret_cast<const u32x4*>((const T*)p.v + row * p.v_stride_t +
                                                 (int64_t)kvh * p.v_stride_h + 8 * c);
      });
    }
  };

  // ---- registers -> LDS (K tile [tok][D], V^T tile [d][tok], both chunk-swizzled) ----------------
  auto k_slot = [&](int tok, int c) { return lds_k + tok * KROW + ((c ^ (tok & KSWZ)) << 4); };
  // V^T row d, tokens in the order the PV MFMA consumes them: 16-byte chunk q = 4 (tile / 2) + g
  // holds tokens 4 g .. 4 g + 3 of the 16-token tiles 2 (tile / 2) and 2 (tile / 2) + 1 (8 bytes
  // each), so that a lane's whole A fragment is ONE ds_read_b128 (two ds_read_b64 merged into a
  // ds_read2 run at half the LDS rate).  v_quad = address of the 8 bytes of tokens n .. n + 3.
  auto v_quad = [&](int d, int n) {
    const int tt = n >> 4, g = (n >> 2) & 3;
    return lds_v + d * VROW + ((((tt >> 1) * 4 + g) ^ ((d >> 1) & 7)) << 4) + (tt & 1) * 8;
  };
  // new tokens' V stays row-major [tok][D] (a plain copy like K) and is consumed with the
  // transposing LDS read; chunk pairs (one 16-wide d tile) are swizzled by the token
  constexpr int VNSWZ = (KCH % 16 == 0) ? 7 : (KCH % 8 == 0) ? 3 : 1;   // stays inside the row
  auto vn_slot = [&](int tok, int c) { return lds_v + tok * KROW + ((c ^ (2 * (tok & VNSWZ))) << 4); };
  auto store_tile = [&](int t) {
    if constexpr (UNI) {
      // slots past the sequence hold stale bytes (possibly NaN): V rows of such keys are zeroed
      const int tok = tid & 63;
      const bool live = t * PF_BN + tok < seq_len;
      if constexpr (!FP8) {
        static_for<K16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int item = tid + PF_THREADS * i;
          const int c = item >> 6;
          if (item < K16_TOTAL) {
            *reinterpret_cast<u32x4*>(k_slot(tok, c)) = c < dr / 8 ? kst[i] : (u32x4){0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(vn_slot(tok, c)) = (live && c < dr / 8) ? vst[i] : (u32x4){0, 0, 0, 0};
          }
        });
      } else {
        static_for<K8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int cc = (tid + PF_THREADS * i) >> 6;
          if (cc < D / 16) {
            uint4 lo, hi;   // exact widening; the scales ride in the score scale / the exponent
            widen16<T>(as_u4(cc < dr / 16 ? kst[i] : (u32x4){0, 0, 0, 0}), lo, hi);
            *reinterpret_cast<uint4*>(k_slot(tok, 2 * cc)) = lo;
            *reinterpret_cast<uint4*>(k_slot(tok, 2 * cc + 1)) = hi;
            widen16<T>(as_u4((live && cc < dr / 16) ? vst[i] : (u32x4){0, 0, 0, 0}), lo, hi);
            *reinterpret_cast<uint4*>(vn_slot(tok, 2 * cc)) = lo;
            *reinterpret_cast<uint4*>(vn_slot(tok, 2 * cc + 1)) = hi;
          }
        });
      }
    } else if (t < ctx_tiles) {
      if constexpr (!FP8) {
        static_for<K16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int item = tid + PF_THREADS * i;
          if (item < K16_TOTAL)
            *reinterpret_cast<u32x4*>(k_slot(item & 63, item >> 6)) = (item >> 6) < dr / 8 ? kst[i] : (u32x4){0, 0, 0, 0};
        });
        static_for<V16_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int item = tid + PF_THREADS * i;
          constexpr int CPP = BS / 8;
          const int j = item % CPP, d = (item / CPP) % D, pg = item / (CPP * D);
          // stale cache bytes of tokens >= ctx may be NaN, and 0 * NaN = NaN in the PV product
          const int n = t * PF_BN + pg * BS + 8 * j;
          if (item < V16_TOTAL) {
            const uint4 v = mask_halfs(as_u4(vst[i]), d < dr ? ctx - n : 0);
            const int nl = pg * BS + 8 * j;
            *reinterpret_cast<uint2*>(v_quad(d, nl)) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(v_quad(d, nl + 4)) = make_uint2(v.z, v.w);
          }
        });
      } else {
        static_for<K8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int item = tid + PF_THREADS * i;
          const int tok = item & 63, cc = item >> 6;
          if (cc < D / 16) {
            uint4 lo, hi;   // exact widening; k_scale is applied to the scores
            widen16<T>(as_u4(cc < dr / 16 ? kst[i] : (u32x4){0, 0, 0, 0}), lo, hi);
            *reinterpret_cast<uint4*>(k_slot(tok, 2 * cc)) = lo;
            *reinterpret_cast<uint4*>(k_slot(tok, 2 * cc + 1)) = hi;
          }
        });
        static_for<V8_ITEMS>([&](auto ic_) {
          constexpr int i = decltype(ic_)::value;
          const int item = tid + PF_THREADS * i;
          constexpr int CPP = BS / 16;
          const int j = item % CPP, d = (item / CPP) % D, pg = item / (CPP * D);
          const int nl = pg * BS + 16 * j;
          if (item < 4 * D) {
            const uint4 raw = mask_bytes(as_u4(vst[i]), d < dr ? ctx - (t * PF_BN + nl) : 0);
            uint4 lo, hi;   // exact widening; v_scale is applied to the probabilities
            widen16<T>(raw, lo, hi);
            *reinterpret_cast<uint2*>(v_quad(d, nl)) = make_uint2(lo.x, lo.y);
            *reinterpret_cast<uint2*>(v_quad(d, nl + 4)) = make_uint2(lo.z, lo.w);
            *reinterpret_cast<uint2*>(v_quad(d, nl + 8)) = make_uint2(hi.x, hi.y);
            *reinterpret_cast<uint2*>(v_quad(d, nl + 12)) = make_uint2(hi.z, hi.w);
          }
        });
      }
    } else {
      static_for<K16_ITEMS>([&](auto ic_) {
        constexpr int i = decltype(ic_)::value;
        const int item = tid + PF_THREADS * i;
        const int tok = item & 63, c = item >> 6;
        if (item < K16_TOTAL) {
          *reinterpret_cast<u32x4*>(k_slot(tok, c)) = c < dr / 8 ? kst[i] : (u32x4){0, 0, 0, 0};
          *reinterpret_cast<u32x4*>(vn_slot(tok, c)) =
              ((t - ctx_tiles) * PF_BN + tok < qlen && c < dr / 8) ? vst[i] : (u32x4){0, 0, 0, 0};
        }
      });
    }
  };

  // fragment read addresses, constant over the tiles (the swizzles depend on lane bits only;
  // the tile-row part is an immediate offset): K per k-step, V^T per 16-token tile, new V per
  // d tile -- otherwise ~150 VALU of address arithmetic per tile and wave
  const uint8_t* k_frag[QK_STEPS];
#pragma unroll
  for (int ks = 0; ks < QK_STEPS; ++ks) k_frag[ks] = k_slot(c16, 4 * ks + g4);
  const uint8_t* v_frag[2];   // per contraction step: tokens 4 g4 .. + 3 of tiles 2 ks2, 2 ks2 + 1
#pragma unroll
  for (int ks2 = 0; ks2 < 2; ++ks2) v_frag[ks2] = v_quad(c16, 32 * ks2 + 4 * g4);
  const uint8_t* vn_frag[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
    vn_frag[dt] = vn_slot(4 * g4 + (c16 >> 2), 2 * dt + ((c16 >> 1) & 1)) + (c16 & 1) * 8;

  // ---- one KV tile: S^T, online softmax, O^T ------------------------------------------------------
  // MASKED = false is the interior fast path (every key of the tile visible to every row of the
  // wave, no ALiBi / window): 5 VALU per score instead of ~14 -- the softmax, not the MFMA,
  // is what bounds this kernel (rocprof: 460 VALU per 32 MFMA before the split).
  // fp8 context: K / V are widened EXACTLY (e4m3 is a subset of f16 / bf16) and k_scale is
  // folded into the score scale, v_scale into the probabilities -- one rounding less than the
  // reference's (fp8 * scale).to(dtype), inside the stated tolerance.
  auto compute_tile = [&](int t, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const bool is_ctx = t < ctx_tiles;   // never with UNI: its tiles take the row-major V path
    const int n0 = is_ctx ? t * PF_BN : (t - ctx_tiles) * PF_BN;
    const bool scaled = FP8 && (UNI || is_ctx);                     // fp8 keys / values in this tile
    const float sct = scaled ? sc2 * kscale : sc2;                  // score scale of this tile
    // value scale of this tile, as log2 and as reciprocal (1 for new tokens)
    const float pv_log2 = scaled ? vscale_log2 : 0.f;
    const float pv_inv = scaled ? vscale_inv : 1.f;
    f32x4 s[4][MT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) s[nt][mt] = (f32x4){0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < QK_STEPS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(k_frag[ks] + 16 * nt * KROW);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s[nt][mt] = Mma<T>::mfma(kf, qf[mt][ks], s[nt][mt]);
      }
    }
    // lane holds S^T[n = n0 + 16 nt + 4 g4 + r][m = m_wave + 16 mt + c16]
    uint32_t pk[4][MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float mx = -INFINITY;
      if constexpr (MASKED) {
        const int m = m_wave + 16 * mt + c16;      // query index within the new tokens
        const int qpos = ctx + m;                   // position in the sequence
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * nt + 4 * g4 + r;
            if constexpr (UNI) {
              // triton_unified_attention.py:219-238: softcap, causal + padding mask, window,
              // then the ALiBi bias slope * (key position - context_len)
              float x = s[nt][mt][r] * sct;
              if (softcap > 0.f) {
                // softcap * tanh(S / softcap), S = scale * q.k; tanh from one exp2 of -2|y|
                const float y = x * (1.0f / kLog2e) / softcap;
                const float e2 = fast_exp2(-2.0f * kLog2e * fabsf(y));
                x = copysignf(softcap * kLog2e * (1.0f - e2) / (1.0f + e2), y);
              }
              bool valid = n <= ctx + rtok[mt] && rok[mt];
              if (window > 0 && ctx + rtok[mt] - n >= window) valid = false;
              if (alibi) x += (float)(n - ctx) * (PACK ? slope_row[mt] : slope2);
              x = valid ? x : -INFINITY;
              s[nt][mt][r] = x;
              mx = fmaxf(mx, x);
              continue;
            }
            const int kpos = is_ctx ? n : ctx + n;
            // context: every cached token is visible; new tokens: causal (prefix_prefill.py:175, 249)
            const bool valid = is_ctx ? n < ctx : n <= m;
            float x = s[nt][mt][r] * sct;
            if (alibi) {   // (:603-608): bias where it is <= 0 and the row is real, else -inf
              const float bias = (float)(kpos - qpos) * slope2;
              x = (bias <= 0.f && qpos < seq_len) ? x + bias : -INFINITY;
            } else if (window > 0 && qpos - kpos >= window) {
              x = -10000.0f * kLog2e;               // (:190-192, :251-254): not -inf
            }
            x = valid ? x : -INFINITY;
            s[nt][mt][r] = x;
            mx = fmaxf(mx, x);
          }
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[nt][mt][r] *= sct;
            mx = fmaxf(mx, s[nt][mt][r]);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[mt], mx);
      // a column with nothing visible yet keeps m = -inf; avoid (-inf) - (-inf)
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = fast_exp2(m_run[mt] - m_use);
      m_run[mt] = m_new;
      // v_scale rides in the exponent: exp2(s - (m - log2 v_scale)) = p * v_scale with no
      // multiply per score; the row sum is taken of the scaled values and unscaled once
      const float m_sub = FP8 ? m_use - pv_log2 : m_use;
      float psum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          e[r] = fast_exp2(s[nt][mt][r] - m_sub);
          psum += e[r];
        }
        pk[nt][mt][0] = Mma<T>::pack2(e[0], e[1]);   // p.to(v.dtype) (:216)
        pk[nt][mt][1] = Mma<T>::pack2(e[2], e[3]);
      }
      if constexpr (FP8) psum *= pv_inv;
      l_run[mt] = l_run[mt] * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][mt] *= alpha;
      }
    }
    // O^T += V^T . P^T, contraction slots: j < 4 -> token 16 ta + 4 g4 + j, j >= 4 -> tile tb
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const int ta = 2 * ks2, tb = ta + 1;
      uint4 pb[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        pb[mt] = make_uint4(pk[ta][mt][0], pk[ta][mt][1], pk[tb][mt][0], pk[tb][mt][1]);
      if (is_ctx) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint4 vf = *reinterpret_cast<const uint4*>(v_frag[ks2] + 16 * dt * VROW);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) o[dt][mt] = Mma<T>::mfma(vf, pb[mt], o[dt][mt]);
        }
      } else {
        // ds_read_b64_tr_b16: the 16 lanes of group g4 read the block of tokens
        // 16 t + 4 g4 + (0..3) x d 16 dt + (0..15); lane 4q+p addresses token q, d 4p..4p+3 and
        // lane c16 receives d = 16 dt + c16 of the four tokens: the V^T fragment, transposed
        // by the hardware.  All 64 lanes are active here (the instruction requires it).
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint2 va = lds_read_tr16(vn_frag[dt] + 16 * ta * KROW);
          const uint2 vb = lds_read_tr16(vn_frag[dt] + 16 * tb * KROW);
          const uint4 vf = make_uint4(va.x, va.y, vb.x, vb.y);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) o[dt][mt] = Mma<T>::mfma(vf, pb[mt], o[dt][mt]);
        }
      }
    }
  };
  // which path a tile takes for THIS wave (wave-uniform): 0 = skip (new tokens entirely above
  // the wave's rows), 1 = interior, 2 = masked
  const int m_hi = m_wave + 16 * MT - 1;       // last query row of the wave
  auto tile_kind = [&](int t) {
    if constexpr (UNI) {
      const int n0 = t * PF_BN;
      // query tokens of the wave's first and last real row
      const int p_lo = row_tok(m_wave);
      const int p_hi = min(min(row_tok(m_hi), start_m * tpb + tpb - 1), qlen - 1);
      if (PACK && (p_lo >= qlen || m_wave - start_m * PF_BM >= tpb * gqa)) return 0;   // no real row: the wave only stages
      if (n0 > ctx + p_hi) return 0;                                       // above the diagonal
      if (window > 0 && n0 + PF_BN - 1 < ctx + p_lo - window + 1) return 0;   // left of the window
      if (alibi || softcap > 0.f) return 2;
      if (n0 + PF_BN - 1 > ctx + p_lo || (!PACK && m_hi >= qlen)) return 2;
      return (window > 0 && ctx + p_hi - n0 >= window) ? 2 : 1;
    }
    if (alibi) return 2;
    if (t < ctx_tiles) {
      const int n0 = t * PF_BN;
      if (n0 + PF_BN > ctx) return 2;
      return (window > 0 && ctx + m_hi - n0 >= window) ? 2 : 1;
    }
    const int n0 = (t - ctx_tiles) * PF_BN;
    if (n0 > m_hi) return 0;
    if (n0 + PF_BN - 1 > m_wave) return 2;
    return (window > 0 && m_hi - n0 >= window) ? 2 : 1;
  };

  // ---- pipeline: the next tile's global loads fly under the MFMAs of the current one ---------------
  // (split: this workgroup takes tiles [t_begin, t_end) -- the reference's
  // tiles_per_segment = cdiv(seq_len, NUM_SEGMENTS * BLOCK_SIZE), :388-396 -- and leaves
  // unnormalised partials for unified_reduce_kernel; a segment past the sequence writes nothing)
  const int tps = split ? (ntiles + p.num_segments - 1) / p.num_segments : ntiles;
  const int t_begin = seg * tps, t_end = min(t_begin + tps, ntiles);
  if (split && t_begin >= ntiles) return;
  Pages pages = load_pages(t_begin);
  if (t_end > t_begin) load_tile(t_begin, pages);
  pages = load_pages(t_begin + 1);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();          // everyone is done reading the previous tile
    store_tile(t);
    __syncthreads();
    if (t + 1 < t_end) {
      load_tile(t + 1, pages);
      pages = load_pages(t + 2);
    }
    const int kind = tile_kind(t);
    if (kind == 1) compute_tile(t, std::false_type{});
    else if (kind == 2) compute_tile(t, std::true_type{});
  }

  // ---- O^T / l -> out: lane holds O^T[d = 16 dt + 4 g4 + r][m] ---------------------------------------
  const bool fp8_out = p.fp8_out_scale != nullptr;
  const float out_inv = fp8_out ? 1.0f / *p.fp8_out_scale : 1.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float l = l_run[mt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int m = m_wave + 16 * mt + c16;
    if (!rok[mt]) continue;
    if (split) {
      // partial of this segment: O (unnormalised, fp32), running max (log2 domain), sum
      const int64_t row = ((int64_t)(tok0 + rtok[mt]) * p.num_heads + row_head(m)) * p.segm_stride + seg;
      float* so = p.segm_out + row * dr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        if (16 * dt + 4 * g4 < dr) *reinterpret_cast<f32x4*>(so + 16 * dt + 4 * g4) = o[dt][mt];
      if (g4 == 0) {
        p.segm_max[row] = m_run[mt];
        p.segm_sum[row] = l;
      }
      continue;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;   // acc / l_i (:276); a fully masked row gives 0
    if (!fp8_out) {
      T* op = (T*)p.o + (int64_t)(tok0 + rtok[mt]) * p.o_stride_t + (int64_t)row_head(m) * p.o_stride_h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (16 * dt + 4 * g4 >= dr) continue;
        const uint2 w = make_uint2(Mma<T>::pack2(o[dt][mt][0] * inv, o[dt][mt][1] * inv),
                                   Mma<T>::pack2(o[dt][mt][2] * inv, o[dt][mt][3] * inv));
        *reinterpret_cast<uint2*>(op + 16 * dt + 4 * g4) = w;
      }
    } else {
      // acc / out_scale, clamp to the fp8 range, store e4m3 (:282-287)
      uint8_t* op = (uint8_t*)p.o + (int64_t)(tok0 + rtok[mt]) * p.o_stride_t +
                    (int64_t)row_head(m) * p.o_stride_h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (16 * dt + 4 * g4 >= dr) continue;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          y[r] = fmaxf(-448.0f, fminf(o[dt][mt][r] * inv * out_inv, 448.0f));
        *reinterpret_cast<uint32_t*>(op + 16 * dt + 4 * g4) = f32x4_to_fp8x4(y[0], y[1], y[2], y[3]);
      }
    }
  }
}

// Merge of the key segments (reduce_segments, triton_unified_attention.py:499-586): one wave per
// (token, q head); segment s of a sequence exists iff s * tiles_per_segment < its tiles.
template <typename T>
__global__ __launch_bounds__(64) void unified_reduce_kernel(const PrefillParams p, int rows_per_block,
                                                            int num_seqs) {
  const int tok = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  // sequence of this token: last seq with cu_seqlens_q[seq] <= tok (find_seq_idx, :33-48)
  int seq = 0;
  for (int lo = 0, hi = num_seqs; lo < hi;) {
    const int mid = (lo + hi) >> 1;
    if (p.b_start_loc[mid] <= tok) { seq = mid; lo = mid + 1; } else hi = mid;
  }
  const int seq_len = p.b_seq_len[seq];
  const int qlen = p.b_start_loc[seq + 1] - p.b_start_loc[seq];
  // mixed launch: the decode kernel has already written this token's output and the tile kernel
  // skipped its sequence, so its segm_* rows hold nothing
  if (p.skip_decode && qlen == 1) return;
  const int kv_end = min(seq_len - qlen + min(rows_per_block, qlen), seq_len);
  const int ntiles = (kv_end + PF_BN - 1) / PF_BN;
  const int tps = (ntiles + p.num_segments - 1) / p.num_segments;
  const int nseg = tps > 0 ? (ntiles + tps - 1) / tps : 0;
  const int64_t row = ((int64_t)tok * p.num_heads + head) * p.segm_stride;
  float mx = -INFINITY;
  for (int s2 = 0; s2 < nseg; ++s2) mx = fmaxf(mx, p.segm_max[row + s2]);
  const int dr = p.head_size;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
  for (int s2 = 0; s2 < nseg; ++s2) {
    const float ms = p.segm_max[row + s2];
    const float w = ms == -INFINITY ? 0.f : fast_exp2(ms - mx);
    l += p.segm_sum[row + s2] * w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = lane + 64 * j;
      if (d < dr) acc[j] += p.segm_out[(row + s2) * dr + d] * w;
    }
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int d = lane + 64 * j;
    if (d >= dr) continue;
    if (p.fp8_out_scale) {
      const float y = fmaxf(-448.0f, fminf(acc[j] * inv * (1.0f / *p.fp8_out_scale), 448.0f));
      ((uint8_t*)p.o)[(int64_t)tok * p.o_stride_t + (int64_t)head * p.o_stride_h + d] = f32_to_fp8(y);
    } else {
      ((T*)p.o)[(int64_t)tok * p.o_stride_t + (int64_t)head * p.o_stride_h + d] = from_f32<T>(acc[j] * inv);
    }
  }
}

template <typename T, bool FP8, int D, int MT, int NW, bool UNI = false, bool PACK = false>
static void launch_cfg(const PrefillParams& p, int block_size, int max_input_len, int num_heads,
                       int batch, hipStream_t st) {
  constexpr int BM = 16 * MT * NW;
  dim3 grid((max_input_len + BM - 1) / BM, num_heads, batch);
  if (p.flat_batch > 0 && p.num_segments <= 1) {
    const int tpb = PACK ? BM / (p.num_heads / p.num_kv_heads) : BM;
    grid = dim3(p.flat_tokens / tpb + p.flat_batch, num_heads, 1);
  }
  if (UNI || block_size == 16)   // UNI reads the block size from p.bs_shift
    hipLaunchKernelGGL((prefill_kernel<T, FP8, D, 16, MT, NW, UNI, PACK>), grid, dim3(64 * NW), 0, st, p);
  else if constexpr (!UNI)
    hipLaunchKernelGGL((prefill_kernel<T, FP8, D, 32, MT, NW, UNI, PACK>), grid, dim3(64 * NW), 0, st, p);
}
// unified_attention: padded head sizes 64 and 128 (one configuration each)
template <typename T, bool FP8>
static int launch_unified(PrefillParams p, int head_size, int block_size, int max_q, int max_k,
                           int num_tokens, int num_heads, int batch, hipStream_t st) {
  const int gqa = p.num_heads / p.num_kv_heads;
  // The flash-layout bandwidth kernel (unified_decode.hip) takes decode-only launches, and the
  // decode tokens of a mixed batch of >= 16 sequences (the V1 engine's usual step: a few prefill
  // chunks among many decodes; as the reference's ROCm backend pairs its prefill kernel with
  // the paged-attention op, chunked_prefill_paged_decode.py:318-338): it skips every sequence
  // whose query length is not 1, the tile kernel below then skips those whose length is.
  // num_tokens == batch * max_q means every query has max_q > 1 tokens: nothing to split off.
  const bool mixed = max_q > 1 && batch >= 16 && (int64_t)num_tokens < (int64_t)batch * max_q;
  if ((max_q == 1 || mixed) && g_uni_decode && g_pf_rows != 1) {
    UniDecodeParams u;
    u.out = p.o; u.q = p.q; u.k_cache = p.k_cache; u.v_cache = p.v_cache;
    u.cu_seqlens_q = p.b_start_loc; u.seq_lens = p.b_seq_len; u.block_table = p.b_loc;
    u.k_scale = p.k_scale; u.v_scale = p.v_scale; u.alibi_slopes = p.alibi_slopes;
    u.out_scale = p.fp8_out_scale;
    u.num_seqs = batch; u.num_heads = p.num_heads; u.num_kv_heads = p.num_kv_heads;
    u.head_size = head_size; u.block_size = block_size;
    u.bt_stride = p.b_loc_stride; u.q_stride_t = p.q_stride_t; u.q_stride_h = p.q_stride_h;
    u.o_stride_t = p.o_stride_t; u.o_stride_h = p.o_stride_h;
    u.kc_block_bytes = p.kc_block_bytes; u.kc_slot_bytes = p.kc_page_bytes; u.kc_head_bytes = p.kc_head_bytes;
    u.vc_block_bytes = p.vc_block_bytes; u.vc_slot_bytes = p.vc_page_bytes; u.vc_head_bytes = p.vc_head_bytes;
    u.sm_scale = p.sm_scale; u.softcap = p.softcap; u.window = p.sliding_window;
    u.max_seq_len = max_k;
    u.segm_out = p.segm_out; u.segm_max = p.segm_max; u.segm_sum = p.segm_sum;
    u.segm_stride = p.segm_stride;
    u.dtype = std::is_same<T, f16_t>::value ? VLLM_AMD_F16 : VLLM_AMD_BF16;
    u.fp8 = FP8; u.stream = st;
    // Split over the keys (when the caller brought scratch).  The chip holds 512 workgroups at a
    // time, so a launch of full-length workgroups takes ceil(wgs n / 512) / n rounds, and every
    // workgroup pays ~4 us of prologue and merge next to ~25 ns per token of streaming: take the
    // smallest segment count within 5 % of the best estimate (513..1023 workgroups: ~1.3
    // rounds instead of 2); few sequences: as many segments of >= 256 tokens as allowed.
    u.hp = unified_decode_heads_per_row(u);
    int nseg = 1;
    if (p.num_segments > 1 && !mixed) {
      const int wgs = batch * (p.num_kv_heads / u.hp) * ((gqa * u.hp + 15) / 16);
      int cap = (max_k + 4 * PF_BN - 1) / (4 * PF_BN);
      if (cap > p.num_segments) cap = p.num_segments;
      if (cap > 16) cap = 16;
      const float ovh = 160.f / (float)(max_k > 0 ? max_k : 1);
      // rows above 128 elements: 256 workgroups carry the bytes in flight of 512 narrow ones (16-bit cache: one workgroup
      // per CU; fp8: two of 32-token steps -- counting 512 slots there split 64 x 4096 in two: 93.8 against 89.9 us)
      const int slots = head_size * u.hp > 128 ? 256 : 512;
      auto cost = [&](int n) { return (float)((wgs * n + slots - 1) / slots) / n * (1.f + n * ovh); };
      float best = 1e30f;
      for (int n = 1; n <= cap; ++n) best = cost(n) < best ? cost(n) : best;
      nseg = cap;
      if (wgs * cap > slots)
        for (int n = 1; n <= cap; ++n)
          if (cost(n) <= best * 1.05f) { nseg = n; break; }
    }
    u.num_segments = nseg;
    if (unified_decode_supported(u)) {
      const int rc = unified_decode(u);
      if (!mixed || rc != VLLM_AMD_OK) return rc;
      p.skip_decode = 1;
    }
  }
  // Long query chunks at head size 128 with only the causal mask: the 256-row kernel (prefill_fa.hip; rows =
  // (token, q head of the group) pairs when the group divides into 256 rows).  "Long" = the longest chunk
  // fills at least one of its row blocks.
  {
    const int pack = (gqa > 1 && gqa <= 16) ? gqa : 1;
    const int fa_rows = head_size > 128 ? 128 : 256;   // rows of a workgroup: four waves at head sizes above 128
    const bool can = head_size > 64 && head_size <= 256 && p.alibi_slopes == nullptr &&
                     p.flat_batch > 0 && p.sm_scale > 0.f && g_pf_rows == 0;
    // ... and the launch has enough 256-row blocks for the chip: below ~128 of them the 64-row kernel's four times as
    // many workgroups finish sooner (chunk 256 over 4096 cached tokens among 255 decodes: 32 blocks, 82 vs 150 us)
    const int64_t ptok = mixed ? (int64_t)num_tokens - batch : (int64_t)num_tokens;   // ~ tokens of the chunks
    const int64_t fa_blocks = ptok / (fa_rows / pack) * (pack > 1 ? p.num_kv_heads : p.num_heads);
    if (can && g_pf_fa != 1 && (g_pf_fa == 2 || (max_q >= fa_rows / pack && fa_blocks >= 128))) {
      FaParams f;
      f.o = p.o; f.q = p.q; f.k_cache = p.k_cache; f.v_cache = p.v_cache;
      f.block_table = p.b_loc; f.cu_seqlens_q = p.b_start_loc; f.seq_lens = p.b_seq_len;
      f.k_scale = p.k_scale; f.v_scale = p.v_scale; f.out_scale = p.fp8_out_scale;
      f.head_size = head_size; f.num_heads = p.num_heads; f.num_kv_heads = p.num_kv_heads; f.num_seqs = batch; f.num_tokens = num_tokens;
      f.bs_shift = p.bs_shift; f.skip_decode = p.skip_decode; f.pack = pack;
      f.bt_stride = p.b_loc_stride; f.q_stride_t = p.q_stride_t; f.q_stride_h = p.q_stride_h;
      f.o_stride_t = p.o_stride_t; f.o_stride_h = p.o_stride_h;
      f.kc_block_bytes = p.kc_block_bytes; f.kc_slot_bytes = p.kc_page_bytes; f.kc_head_bytes = p.kc_head_bytes;
      f.vc_block_bytes = p.vc_block_bytes; f.vc_slot_bytes = p.vc_page_bytes; f.vc_head_bytes = p.vc_head_bytes;
      f.sm_scale = p.sm_scale;
      f.window = p.sliding_window; f.softcap = p.softcap;
      f.prefix = 0; f.k_new = f.v_new = nullptr; f.k_stride_t = f.k_stride_h = f.v_stride_t = f.v_stride_h = 0;
      f.blocks_per_seq = 0;
      f.abl = g_pf_fa_abl;
      f.stagger = g_pf_fa_stagger;
      return launch_unified_fa(f, std::is_same<T, f16_t>::value ? VLLM_AMD_F16 : VLLM_AMD_BF16, FP8, st);
    }
  }
  // GQA: rows = (token, q head of the group) pairs, grid.y = kv heads ("pf_rows" = 1: off)
  if (gqa > 1 && gqa <= 16 && g_pf_rows != 1) {
    const int bm = head_size <= 64 ? 128 : 64;   // rows per workgroup
    const int tpb = bm / gqa;                     // tokens per workgroup
    int rows = (max_q + tpb - 1) / tpb * bm;      // launch_cfg wants rows: grid.x = rows / bm
    // few sequences of one row block each (decode): split the keys over up to num_segments
    // workgroups so that the launch has ~512 of them, >= 4 tiles per segment
    if (p.num_segments > 1 && max_q <= tpb) {
      int nseg = (512 + batch * p.num_kv_heads - 1) / (batch * p.num_kv_heads);
      const int by_len = (max_k + 4 * PF_BN - 1) / (4 * PF_BN);
      if (nseg > by_len) nseg = by_len;
      if (nseg > p.num_segments) nseg = p.num_segments;
      p.num_segments = nseg > 1 ? nseg : 1;
      rows = p.num_segments * bm;
    } else {
      p.num_segments = 1;
    }
    if (head_size <= 64) launch_cfg<T, FP8, 64, 2, 4, true, true>(p, 16, rows, p.num_kv_heads, batch, st);
    else if (head_size <= 128) launch_cfg<T, FP8, 128, 1, 4, true, true>(p, 16, rows, p.num_kv_heads, batch, st);
    else launch_cfg<T, FP8, 256, 1, 4, true, true>(p, 16, rows, p.num_kv_heads, batch, st);
    if (p.num_segments > 1)
      hipLaunchKernelGGL((unified_reduce_kernel<T>), dim3(num_tokens, num_heads), dim3(64), 0, st, p, tpb,
                         batch);
    return VLLM_AMD_OK;
  }
  p.num_segments = 1;
  if (head_size <= 64) launch_cfg<T, FP8, 64, 2, 4, true>(p, 16, max_q, num_heads, batch, st);
  else if (head_size <= 128) launch_cfg<T, FP8, 128, 1, 4, true>(p, 16, max_q, num_heads, batch, st);
  else launch_cfg<T, FP8, 256, 1, 4, true>(p, 16, max_q, num_heads, batch, st);
  return VLLM_AMD_OK;
}
template <typename T, bool FP8, int D>
static void launch_bs(const PrefillParams& p, int block_size, int max_input_len, int num_heads,
                      int batch, hipStream_t st) {
  // head sizes >= 96 do not fit two query tiles per wave in 256 VGPRs without spilling
  if constexpr (D > 128) {
    launch_cfg<T, FP8, D, 1, 4>(p, block_size, max_input_len, num_heads, batch, st);
  } else if constexpr (D >= 96) {
    if (g_pf_rows == 128) launch_cfg<T, FP8, D, 1, 8>(p, block_size, max_input_len, num_heads, batch, st);
    else launch_cfg<T, FP8, D, 1, 4>(p, block_size, max_input_len, num_heads, batch, st);
  } else {
    if (g_pf_rows == 64) launch_cfg<T, FP8, D, 2, 2>(p, block_size, max_input_len, num_heads, batch, st);
    else launch_cfg<T, FP8, D, 2, 4>(p, block_size, max_input_len, num_heads, batch, st);
  }
}
template <typename T, bool FP8>
static void launch_d(const PrefillParams& p, int head_size, int block_size, int max_input_len,
                     int num_heads, int batch, hipStream_t st) {
  // Long chunks at head size 128 with only the causal mask: the 256-row kernel (prefill_fa.hip, PREFIX form)
  {
    const int gqa = p.num_heads / p.num_kv_heads;
    const int pack = (gqa > 1 && gqa <= 16) ? gqa : 1;
    const int fa_rows = head_size > 128 ? 128 : 256;
    const bool can = head_size > 64 && head_size <= 256 && p.alibi_slopes == nullptr && p.sliding_window == 0 && p.sm_scale > 0.f &&
                     g_pf_rows == 0 && (block_size & (block_size - 1)) == 0;
    const int64_t ptok = p.flat_batch > 0 ? (int64_t)p.flat_tokens : (int64_t)max_input_len * batch;
    const int64_t fa_blocks = ptok / (fa_rows / pack) * (pack > 1 ? p.num_kv_heads : p.num_heads);   // (see launch_unified)
    if (can && g_pf_fa != 1 && (g_pf_fa == 2 || (max_input_len >= fa_rows / pack && fa_blocks >= 128))) {
      FaParams f;
      f.o = p.o; f.q = p.q; f.k_cache = p.k_cache; f.v_cache = p.v_cache;
      f.block_table = p.b_loc; f.cu_seqlens_q = p.b_start_loc; f.seq_lens = p.b_seq_len;
      f.k_scale = p.k_scale; f.v_scale = p.v_scale; f.out_scale = p.fp8_out_scale;
      f.head_size = head_size; f.num_heads = p.num_heads; f.num_kv_heads = p.num_kv_heads; f.num_seqs = batch; f.num_tokens = p.flat_tokens;
      f.bs_shift = 0;
      while ((1 << f.bs_shift) < block_size) ++f.bs_shift;
      f.skip_decode = p.skip_decode; f.pack = pack;
      f.bt_stride = p.b_loc_stride; f.q_stride_t = p.q_stride_t; f.q_stride_h = p.q_stride_h;
      f.o_stride_t = p.o_stride_t; f.o_stride_h = p.o_stride_h;
      f.kc_block_bytes = p.kc_block_bytes; f.kc_slot_bytes = 0; f.kc_head_bytes = p.kc_head_bytes;
      f.vc_block_bytes = p.vc_block_bytes; f.vc_slot_bytes = 0; f.vc_head_bytes = p.vc_head_bytes;
      f.sm_scale = p.sm_scale;
      f.window = 0; f.softcap = 0.f;
      f.prefix = 1; f.k_new = p.k; f.v_new = p.v;
      f.k_stride_t = p.k_stride_t; f.k_stride_h = p.k_stride_h; f.v_stride_t = p.v_stride_t; f.v_stride_h = p.v_stride_h;
      const int tpb = fa_rows / pack;
      f.blocks_per_seq = p.flat_batch > 0 ? 0 : (max_input_len + tpb - 1) / tpb;
      f.abl = 0; f.stagger = g_pf_fa_stagger;
      launch_unified_fa(f, std::is_same<T, f16_t>::value ? VLLM_AMD_F16 : VLLM_AMD_BF16, FP8, st);
      return;
    }
  }
  switch ((head_size + 31) / 32 * 32) {   // padded to the MFMA K step
    case 32: launch_bs<T, FP8, 32>(p, block_size, max_input_len, num_heads, batch, st); break;
    case 64: launch_bs<T, FP8, 64>(p, block_size, max_input_len, num_heads, batch, st); break;
    case 96: launch_bs<T, FP8, 96>(p, block_size, max_input_len, num_heads, batch, st); break;
    case 128: launch_bs<T, FP8, 128>(p, block_size, max_input_len, num_heads, batch, st); break;
    default: launch_bs<T, FP8, 256>(p, block_size, max_input_len, num_heads, batch, st); break;
  }
}

}  // namespace va

using namespace va;

// ---------------------------------------------------------------------------------------
// context_attention_fwd for float32 queries (the reference's test grid has them,
// tests/kernels/attention/test_prefix_prefill.py:513-560; its Triton kernel then takes the
// IEEE dot, prefix_prefill.py:760-767): a plain VALU kernel in fp32, one workgroup (256 threads)
// per (query token, q head) -- chunks of 256 keys, one key per thread for the scores, one
// output column per thread for P V, online softmax across chunks.  Context keys come from the
// paged cache (K [blk][kvh][D/x][BS][x] with x = 16 bytes, V [blk][kvh][D][BS]; float32 or
// fp8 e4m3), the new tokens from the dense k / v.  Masks as the tile kernel's: causal over
// the new tokens, the window as a -10000 score (:190-192), the ALiBi variant's bias rule
// (:603-608).  Functional coverage, not a tuned path.
// ---------------------------------------------------------------------------------------
template <bool FP8>
__global__ __launch_bounds__(256) void prefix_f32_kernel(PrefillParams p, int batch, int head_size,
                                                          int block_size) {
  const int tok = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
  int lo = 0, hi = batch;   // sequence of this token: the last one with start <= tok
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.b_start_loc[mid] <= tok) lo = mid; else hi = mid - 1;
  }
  const int b = lo;
  if (b >= batch || tok >= p.b_start_loc[batch]) return;
  const int tok0 = p.b_start_loc[b];
  const int qlen = p.b_start_loc[b + 1] - tok0;
  const int seq_len = p.b_seq_len[b];
  const int ctx = seq_len - qlen;
  if (p.skip_decode && qlen == 1) return;
  const int m = tok - tok0, qpos = ctx + m;
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  constexpr int ES = FP8 ? 1 : 4, X = 16 / ES;
  const float kscale = FP8 ? *p.k_scale : 1.f, vscale = FP8 ? *p.v_scale : 1.f;
  const bool alibi = p.alibi_slopes != nullptr;
  const float slope = alibi ? p.alibi_slopes[head] : 0.f;
  const int window = alibi ? 0 : p.sliding_window;

  __shared__ float q_s[256];
  __shared__ float p_s[256];
  __shared__ float red[8];
  const float* qp = (const float*)p.q + (int64_t)tok * p.q_stride_t + (int64_t)head * p.q_stride_h;
  if (tid < head_size) q_s[tid] = qp[tid];
  __syncthreads();
  auto block_max = [&](float v) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
  };
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = v;
    __syncthreads();
    v = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
    return v;
  };
  const int* bl = p.b_loc + (int64_t)b * p.b_loc_stride;
  float m_run = -INFINITY, l_run = 0.f, o = 0.f;
  for (int n0 = 0; n0 <= qpos; n0 += 256) {
    const int n = n0 + tid;
    float x = -INFINITY;
    if (n <= qpos) {
      float acc = 0.f;
      if (n < ctx) {
        const uint8_t* kp = p.k_cache + (int64_t)bl[n / block_size] * p.kc_block_bytes +
                            (int64_t)kvh * p.kc_head_bytes + (int64_t)(n % block_size) * 16;
        for (int d = 0; d < head_size; d += X) {
          const uint4 raw = *reinterpret_cast<const uint4*>(kp + (int64_t)(d / X) * block_size * 16);
          if constexpr (FP8) {
            const uint8_t* e = reinterpret_cast<const uint8_t*>(&raw);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = fmaf(q_s[d + j], fp8_to_f32(e[j]) * kscale, acc);
          } else {
            const float* e = reinterpret_cast<const float*>(&raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = fmaf(q_s[d + j], e[j], acc);
          }
        }
      } else {
        const float* kp = (const float*)p.k + (int64_t)(tok0 + n - ctx) * p.k_stride_t + (int64_t)kvh * p.k_stride_h;
        for (int d = 0; d < head_size; d += 4) {
          const float4 e = *reinterpret_cast<const float4*>(kp + d);
          acc = fmaf(q_s[d], e.x, acc);
          acc = fmaf(q_s[d + 1], e.y, acc);
          acc = fmaf(q_s[d + 2], e.z, acc);
          acc = fmaf(q_s[d + 3], e.w, acc);
        }
      }
      x = acc * p.sm_scale;
      if (alibi) {
        const float bias = (float)(n - qpos) * slope;
        x = (bias <= 0.f && qpos < seq_len) ? x + bias : -INFINITY;
      } else if (window > 0 && qpos - n >= window) {
        x = -10000.0f;
      }
    }
    const float m_new = fmaxf(m_run, block_max(x));
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_use);   // first chunk: exp(-inf) = 0
    const float e = n <= qpos ? __expf(x - m_use) : 0.f;
    l_run = l_run * alpha + block_sum(e);
    m_run = m_new;
    p_s[tid] = e;
    __syncthreads();
    o *= alpha;
    if (tid < head_size) {
      const int cnt = min(256, qpos - n0 + 1);
      for (int i = 0; i < cnt; ++i) {
        const int nn = n0 + i;
        float v;
        if (nn < ctx) {
          const uint8_t* vp = p.v_cache + (int64_t)bl[nn / block_size] * p.vc_block_bytes +
                              (int64_t)kvh * p.vc_head_bytes + ((int64_t)tid * block_size + nn % block_size) * ES;
          v = FP8 ? fp8_to_f32(*vp) * vscale : *reinterpret_cast<const float*>(vp);
        } else {
          v = ((const float*)p.v)[(int64_t)(tok0 + nn - ctx) * p.v_stride_t + (int64_t)kvh * p.v_stride_h + tid];
        }
        o = fmaf(p_s[i], v, o);
      }
    }
    __syncthreads();
  }
  if (tid < head_size) {
    const float r = l_run > 0.f ? o / l_run : 0.f;
    const int64_t oi = (int64_t)tok * p.o_stride_t + (int64_t)head * p.o_stride_h + tid;
    if (p.fp8_out_scale) ((uint8_t*)p.o)[oi] = f32_to_fp8(fmaxf(-448.0f, fminf(r * (1.0f / *p.fp8_out_scale), 448.0f)));
    else ((float*)p.o)[oi] = r;
  }
}

extern "C" int vllm_amd_context_attention_fwd(
    void* o, const void* q, const void* k, const void* v, const void* k_cache, const void* v_cache,
    const int* b_loc, const int* b_start_loc, const int* b_seq_len, const float* k_scale,
    const float* v_scale, const float* alibi_slopes, const float* fp8_out_scale, int batch,
    int num_heads, int num_kv_heads, int head_size, int block_size, int max_input_len,
    int64_t b_loc_stride, int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, int64_t o_stride_t,
    int64_t o_stride_h, int64_t kc_block_stride, int64_t kc_head_stride, int64_t vc_block_stride,
    int64_t vc_head_stride, float sm_scale, int sliding_window, int skip_decode, int num_tokens,
    int dtype, int kv_cache_dtype, void* stream) {
  const char* op = "context_attention_fwd";
  VA_REQUIRE(batch >= 0 && max_input_len >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (batch == 0 || max_input_len == 0 || num_heads == 0) return VLLM_AMD_OK;
  VA_REQUIRE(o && q && k && v && k_cache && v_cache && b_loc && b_start_loc && b_seq_len,
             VLLM_AMD_EINVAL, "%s: null pointer", op);
  VA_REQUIRE(dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16 || dtype == VLLM_AMD_F32,
             VLLM_AMD_EUNSUPPORTED, "%s: query dtype must be float16, bfloat16 or float32", op);
  VA_REQUIRE(kv_cache_dtype == VLLM_AMD_KV_AUTO || kv_cache_dtype == VLLM_AMD_KV_FP8_E4M3,
             VLLM_AMD_EUNSUPPORTED, "Unsupported FP8 dtype: %d", kv_cache_dtype);
  if (dtype == VLLM_AMD_F32) {   // the VALU kernel above
    VA_REQUIRE(head_size >= 4 && head_size <= 256 &&
                   head_size % (kv_cache_dtype == VLLM_AMD_KV_AUTO ? 4 : 16) == 0,
               VLLM_AMD_EUNSUPPORTED,
               "%s: unsupported head size: %d (float32: multiples of 4, of 16 with an fp8 cache, up to 256)",
               op, head_size);
    VA_REQUIRE(block_size > 0, VLLM_AMD_EUNSUPPORTED, "%s: unsupported block size: %d", op, block_size);
    VA_REQUIRE(num_tokens > 0, VLLM_AMD_EINVAL, "%s: float32 queries need num_tokens", op);
    VA_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, VLLM_AMD_EINVAL,
               "%s: num_heads must be a multiple of num_kv_heads", op);
    VA_REQUIRE(kv_cache_dtype == VLLM_AMD_KV_AUTO || (k_scale && v_scale), VLLM_AMD_EINVAL,
               "%s: fp8 cache needs k_scale and v_scale", op);
    VA_REQUIRE(k_stride_t % 4 == 0 && k_stride_h % 4 == 0 && (((uintptr_t)k) & 15) == 0 &&
                   (((uintptr_t)k_cache) & 15) == 0 && (((uintptr_t)q) & 3) == 0 &&
                   (((uintptr_t)v) & 3) == 0 && (((uintptr_t)o) & 3) == 0 &&
                   (((uintptr_t)v_cache) & 3) == 0,
               VLLM_AMD_EUNSUPPORTED, "%s: k and k_cache must be 16-byte aligned with strides that are multiples of 4", op);
    const int es32 = kv_cache_dtype == VLLM_AMD_KV_AUTO ? 4 : 1;
    PrefillParams p;
    p.o = o; p.q = q; p.k = k; p.v = v;
    p.k_cache = (const uint8_t*)k_cache; p.v_cache = (const uint8_t*)v_cache;
    p.b_loc = b_loc; p.b_start_loc = b_start_loc; p.b_seq_len = b_seq_len;
    p.k_scale = k_scale; p.v_scale = v_scale; p.alibi_slopes = alibi_slopes;
    p.fp8_out_scale = fp8_out_scale;
    p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.head_size = head_size;
    p.b_loc_stride = b_loc_stride;
    p.q_stride_t = q_stride_t; p.q_stride_h = q_stride_h;
    p.k_stride_t = k_stride_t; p.k_stride_h = k_stride_h;
    p.v_stride_t = v_stride_t; p.v_stride_h = v_stride_h;
    p.o_stride_t = o_stride_t; p.o_stride_h = o_stride_h;
    p.kc_block_bytes = kc_block_stride * es32; p.kc_head_bytes = kc_head_stride * es32;
    p.vc_block_bytes = vc_block_stride * es32; p.vc_head_bytes = vc_head_stride * es32;
    p.kc_page_bytes = p.vc_page_bytes = 0;
    p.sm_scale = sm_scale;
    p.sliding_window = sliding_window > 0 ? sliding_window : 0;
    p.skip_decode = skip_decode;
    p.softcap = 0.f;
    p.segm_out = p.segm_max = p.segm_sum = nullptr;
    p.num_segments = p.segm_stride = 1;
    p.bs_shift = 0;
    p.flat_batch = batch; p.flat_tokens = num_tokens;
    const dim3 grid((unsigned)num_tokens, (unsigned)num_heads);
    if (es32 == 4) hipLaunchKernelGGL((prefix_f32_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p, batch, head_size, block_size);
    else hipLaunchKernelGGL((prefix_f32_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p, batch, head_size, block_size);
    VA_LAUNCH_CHECK(op);
    return VLLM_AMD_OK;
  }
  VA_REQUIRE(head_size >= 8 && head_size <= 256 &&
                 head_size % (kv_cache_dtype == VLLM_AMD_KV_AUTO ? 8 : 16) == 0,
             VLLM_AMD_EUNSUPPORTED,
             "%s: unsupported head size: %d (multiples of 8, of 16 with an fp8 cache, up to 256)",
             op, head_size);
  VA_REQUIRE(block_size == 16 || block_size == 32, VLLM_AMD_EUNSUPPORTED,
             "%s: unsupported block size: %d (16, 32)", op, block_size);
  VA_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, VLLM_AMD_EINVAL,
             "%s: num_heads must be a multiple of num_kv_heads", op);
  VA_REQUIRE(kv_cache_dtype == VLLM_AMD_KV_AUTO || (k_scale && v_scale), VLLM_AMD_EINVAL,
             "%s: fp8 cache needs k_scale and v_scale", op);
  VA_REQUIRE(q_stride_t % 8 == 0 && q_stride_h % 8 == 0 && k_stride_t % 8 == 0 &&
                 k_stride_h % 8 == 0 && v_stride_t % 8 == 0 && v_stride_h % 8 == 0 &&
                 o_stride_t % 4 == 0 && o_stride_h % 4 == 0 && (((uintptr_t)q) & 15) == 0 &&
                 (((uintptr_t)k) & 15) == 0 && (((uintptr_t)v) & 15) == 0 &&
                 (((uintptr_t)o) & 7) == 0 && (((uintptr_t)k_cache) & 15) == 0 &&
                 (((uintptr_t)v_cache) & 15) == 0,
             VLLM_AMD_EUNSUPPORTED, "%s: tensors must be 16-byte aligned with strides that are multiples of 8", op);
  const int es = kv_cache_dtype == VLLM_AMD_KV_AUTO ? 2 : 1;
  PrefillParams p;
  p.o = o; p.q = q; p.k = k; p.v = v;
  p.k_cache = (const uint8_t*)k_cache; p.v_cache = (const uint8_t*)v_cache;
  p.b_loc = b_loc; p.b_start_loc = b_start_loc; p.b_seq_len = b_seq_len;
  p.k_scale = k_scale; p.v_scale = v_scale; p.alibi_slopes = alibi_slopes;
  p.fp8_out_scale = fp8_out_scale;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.head_size = head_size;
  p.b_loc_stride = b_loc_stride;
  p.q_stride_t = q_stride_t; p.q_stride_h = q_stride_h;
  p.k_stride_t = k_stride_t; p.k_stride_h = k_stride_h;
  p.v_stride_t = v_stride_t; p.v_stride_h = v_stride_h;
  p.o_stride_t = o_stride_t; p.o_stride_h = o_stride_h;
  p.kc_block_bytes = kc_block_stride * es; p.kc_head_bytes = kc_head_stride * es;
  p.vc_block_bytes = vc_block_stride * es; p.vc_head_bytes = vc_head_stride * es;
  p.sm_scale = sm_scale;
  p.sliding_window = sliding_window > 0 ? sliding_window : 0;
  p.skip_decode = skip_decode;
  p.kc_page_bytes = p.vc_page_bytes = 0;
  p.softcap = 0.f;
  p.segm_out = p.segm_max = p.segm_sum = nullptr;
  p.num_segments = p.segm_stride = 1;
  p.bs_shift = 0;
  p.flat_batch = num_tokens > 0 ? batch : 0;
  p.flat_tokens = num_tokens;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLLM_AMD_BF16) {
    if (es == 2) launch_d<bf16_t, false>(p, head_size, block_size, max_input_len, num_heads, batch, st);
    else launch_d<bf16_t, true>(p, head_size, block_size, max_input_len, num_heads, batch, st);
  } else {
    if (es == 2) launch_d<f16_t, false>(p, head_size, block_size, max_input_len, num_heads, batch, st);
    else launch_d<f16_t, true>(p, head_size, block_size, max_input_len, num_heads, batch, st);
  }
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}

// unified_attention (vllm/attention/ops/triton_unified_attention.py:588-700): q over the new
// tokens of every sequence, all keys / values from the flash-layout cache, causal.

// ---------------------------------------------------------------------------------------
// unified_attention for head sizes the tile kernel does not take (136 .. 256; the reference lists
// 160 / 192 / 224 / 256, triton_attn.py:147-148): a plain VALU kernel, one workgroup (256 threads)
// per (query token, q head) -- chunks of 256 keys, one key per thread for the scores, one output
// column per thread for P V, online softmax across chunks.  Same masks and bias as the tile
// kernel (triton_unified_attention.py:219-238).  Functional coverage, not a tuned path.
// ---------------------------------------------------------------------------------------
template <typename T, bool FP8>
__global__ __launch_bounds__(256) void unified_wide_kernel(PrefillParams p, int num_seqs, int head_size,
                                                         int block_size) {
  const int tok = blockIdx.x, head = blockIdx.y, tid = threadIdx.x;
  // sequence of this token (find_seq_idx, :33-48)
  int lo = 0, hi = num_seqs;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.b_start_loc[mid] <= tok) lo = mid; else hi = mid - 1;
  }
  const int b = lo;
  if (b >= num_seqs || tok >= p.b_start_loc[num_seqs]) return;
  const int q_len = p.b_start_loc[b + 1] - p.b_start_loc[b];
  const int kv_len = p.b_seq_len[b];
  const int qpos = kv_len - q_len + (tok - p.b_start_loc[b]);
  const int kvh = head / (p.num_heads / p.num_kv_heads);
  constexpr int ES = FP8 ? 1 : 2;
  const float kscale = FP8 ? *p.k_scale : 1.f, vscale = FP8 ? *p.v_scale : 1.f;
  const float slope = p.alibi_slopes ? p.alibi_slopes[head] : 0.f;
  const int window = p.sliding_window;
  const int first = window > 0 ? max(0, qpos - window + 1) : 0;   // visible keys: [first, qpos]

  __shared__ float q_s[256];
  __shared__ float p_s[256];
  __shared__ float red[8];
  const T* qp = (const T*)p.q + (int64_t)tok * p.q_stride_t + (int64_t)head * p.q_stride_h;
  if (tid < head_size) q_s[tid] = to_f32(qp[tid]);
  __syncthreads();
  auto block_max = [&](float v) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
  };
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = v;
    __syncthreads();
    v = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
    return v;
  };
  const int* bt = p.b_loc + (int64_t)b * p.b_loc_stride;
  float m_run = -INFINITY, l_run = 0.f, o = 0.f;
  for (int n0 = first; n0 <= qpos; n0 += 256) {
    const int n = n0 + tid;
    float x = -INFINITY;
    if (n <= qpos) {
      const int64_t blk = bt[n / block_size];
      const uint8_t* kp = p.k_cache + blk * p.kc_block_bytes + (int64_t)(n % block_size) * p.kc_page_bytes +
                          (int64_t)kvh * p.kc_head_bytes;
      float acc = 0.f;
      for (int d = 0; d < head_size; d += 16 / ES) {
        const uint4 raw = *reinterpret_cast<const uint4*>(kp + d * ES);
        if constexpr (FP8) {
          const uint8_t* e = reinterpret_cast<const uint8_t*>(&raw);
#pragma unroll
          for (int j = 0; j < 16; ++j) acc = fmaf(q_s[d + j], fp8_to_f32(e[j]), acc);
        } else {
          const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc = fmaf(q_s[d + j], to_f32(e[j]), acc);
        }
      }
      x = acc * p.sm_scale * kscale;
      if (p.softcap > 0.f) x = p.softcap * tanhf(x / p.softcap);
      x += slope * (float)(n - qpos);
    }
    const float m_new = fmaxf(m_run, block_max(x));
    const float alpha = __expf(m_run - m_new);   // first chunk: exp(-inf) = 0
    float e = n <= qpos ? __expf(x - m_new) : 0.f;
    l_run = l_run * alpha + block_sum(e);
    m_run = m_new;
    p_s[tid] = to_f32(from_f32<T>(e));   // probabilities in the value dtype, as the reference's P.to(V.dtype)
    __syncthreads();
    o *= alpha;
    if (tid < head_size) {
      const int cnt = min(256, qpos - n0 + 1);
      for (int i = 0; i < cnt; ++i) {
        const int nn = n0 + i;
        const int64_t blk = bt[nn / block_size];
        const uint8_t* vp = p.v_cache + blk * p.vc_block_bytes + (int64_t)(nn % block_size) * p.vc_page_bytes +
                            (int64_t)kvh * p.vc_head_bytes + (int64_t)tid * ES;
        const float v = FP8 ? fp8_to_f32(*vp) : to_f32(*reinterpret_cast<const T*>(vp));
        o = fmaf(p_s[i], v, o);
      }
    }
    __syncthreads();
  }
  if (tid < head_size) {
    const float r = o * vscale / l_run;
    const int64_t oi = (int64_t)tok * p.o_stride_t + (int64_t)head * p.o_stride_h + tid;
    if (p.fp8_out_scale) ((uint8_t*)p.o)[oi] = f32_to_fp8(r * (1.0f / *p.fp8_out_scale));
    else ((T*)p.o)[oi] = from_f32<T>(r);
  }
}

extern "C" int vllm_amd_unified_attention(
    void* out, const void* q, const void* k_cache, const void* v_cache, const int* cu_seqlens_q,
    const int* seqused_k, const int* block_table, const float* k_descale, const float* v_descale,
    const float* alibi_slopes, const float* output_scale, int num_seqs, int num_heads,
    int num_kv_heads, int head_size, int block_size, int max_seqlen_q, int64_t block_table_stride,
    int64_t q_stride_t, int64_t q_stride_h, int64_t o_stride_t, int64_t o_stride_h,
    int64_t kc_stride_block, int64_t kc_stride_slot, int64_t kc_stride_head,
    int64_t vc_stride_block, int64_t vc_stride_slot, int64_t vc_stride_head, float softmax_scale,
    int sliding_window, float softcap, int num_tokens, int max_seqlen_k, float* segm_output,
    float* segm_max, float* segm_expsum, int num_segments, int dtype, int kv_cache_dtype,
    void* stream) {
  const char* op = "unified_attention";
  VA_REQUIRE(num_seqs >= 0 && max_seqlen_q >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (num_seqs == 0 || max_seqlen_q == 0 || num_heads == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && q && k_cache && v_cache && cu_seqlens_q && seqused_k && block_table,
             VLLM_AMD_EINVAL, "%s: null pointer", op);
  VA_REQUIRE(dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16, VLLM_AMD_EUNSUPPORTED,
             "%s: query dtype must be float16 or bfloat16", op);
  VA_REQUIRE(kv_cache_dtype == VLLM_AMD_KV_AUTO || kv_cache_dtype == VLLM_AMD_KV_FP8_E4M3,
             VLLM_AMD_EUNSUPPORTED, "Unsupported FP8 dtype: %d", kv_cache_dtype);
  const int es = kv_cache_dtype == VLLM_AMD_KV_AUTO ? 2 : 1;
  VA_REQUIRE(head_size >= 8 && head_size <= 256 && head_size % (es == 2 ? 8 : 16) == 0,
             VLLM_AMD_EUNSUPPORTED,
             "%s: unsupported head size: %d (multiples of 8, of 16 with an fp8 cache, up to 256)",
             op, head_size);
  VA_REQUIRE(block_size >= 16 && (block_size & (block_size - 1)) == 0, VLLM_AMD_EUNSUPPORTED,
             "%s: unsupported block size: %d (powers of two from 16)", op, block_size);
  VA_REQUIRE(head_size <= 128 || num_tokens > 0, VLLM_AMD_EINVAL,
             "%s: head sizes above 128 need num_tokens", op);
  VA_REQUIRE(num_kv_heads > 0 && num_heads % num_kv_heads == 0, VLLM_AMD_EINVAL,
             "%s: num_heads must be a multiple of num_kv_heads", op);
  VA_REQUIRE(es == 2 || (k_descale && v_descale), VLLM_AMD_EINVAL,
             "%s: fp8 cache needs k_descale and v_descale", op);
  VA_REQUIRE(softcap >= 0.f && sliding_window >= 0, VLLM_AMD_EINVAL,
             "%s: softcap and sliding_window must not be negative", op);
  VA_REQUIRE(num_segments <= 1 || (segm_output && segm_max && segm_expsum && num_tokens > 0 &&
                                   (((uintptr_t)segm_output) & 15) == 0),
             VLLM_AMD_EINVAL, "%s: num_segments > 1 needs the three segment buffers (16-byte aligned) and num_tokens", op);
  const int ca = 16 / es;   // cache elements per 16 bytes
  VA_REQUIRE(q_stride_t % 8 == 0 && q_stride_h % 8 == 0 && o_stride_t % 4 == 0 &&
                 o_stride_h % 4 == 0 && kc_stride_block % ca == 0 && kc_stride_slot % ca == 0 &&
                 kc_stride_head % ca == 0 && vc_stride_block % ca == 0 &&
                 vc_stride_slot % ca == 0 && vc_stride_head % ca == 0 &&
                 (((uintptr_t)q) & 15) == 0 && (((uintptr_t)out) & 7) == 0 &&
                 (((uintptr_t)k_cache) & 15) == 0 && (((uintptr_t)v_cache) & 15) == 0,
             VLLM_AMD_EUNSUPPORTED, "%s: tensors must be 16-byte aligned with 16-byte strides", op);
  PrefillParams p;
  p.o = out; p.q = q; p.k = q; p.v = q;   // no separate new-token K / V
  p.k_cache = (const uint8_t*)k_cache; p.v_cache = (const uint8_t*)v_cache;
  p.b_loc = block_table; p.b_start_loc = cu_seqlens_q; p.b_seq_len = seqused_k;
  p.k_scale = k_descale; p.v_scale = v_descale; p.alibi_slopes = alibi_slopes;
  p.fp8_out_scale = output_scale;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.head_size = head_size;
  p.b_loc_stride = block_table_stride;
  p.q_stride_t = q_stride_t; p.q_stride_h = q_stride_h;
  p.k_stride_t = p.k_stride_h = p.v_stride_t = p.v_stride_h = 0;
  p.o_stride_t = o_stride_t; p.o_stride_h = o_stride_h;
  p.kc_block_bytes = kc_stride_block * es; p.kc_page_bytes = kc_stride_slot * es;
  p.kc_head_bytes = kc_stride_head * es;
  p.vc_block_bytes = vc_stride_block * es; p.vc_page_bytes = vc_stride_slot * es;
  p.vc_head_bytes = vc_stride_head * es;
  p.sm_scale = softmax_scale;
  p.sliding_window = sliding_window;
  p.skip_decode = 0;
  p.softcap = softcap;
  p.segm_out = segm_output; p.segm_max = segm_max; p.segm_sum = segm_expsum;
  p.num_segments = num_segments > 1 ? num_segments : 1;
  p.segm_stride = p.num_segments;
  // one grid entry per row block of any sequence (launch_cfg) when the caller says how many
  // query tokens there are
  p.bs_shift = 0;
  while ((1 << p.bs_shift) < block_size) ++p.bs_shift;
  p.flat_batch = num_tokens > 0 ? num_seqs : 0;
  p.flat_tokens = num_tokens;
  hipStream_t st = (hipStream_t)stream;
  int rc = VLLM_AMD_OK;
  if (head_size > 128 && g_pf_rows == 2) {   // the VALU wide-head kernel above ("pf_rows" = 2; the tile kernel takes these sizes)
    const dim3 grid((unsigned)num_tokens, (unsigned)num_heads);
#define WIDE(T, F) hipLaunchKernelGGL((unified_wide_kernel<T, F>), grid, dim3(256), 0, st, p, num_seqs, head_size, block_size)
    if (dtype == VLLM_AMD_BF16) { if (es == 2) WIDE(bf16_t, false); else WIDE(bf16_t, true); }
    else { if (es == 2) WIDE(f16_t, false); else WIDE(f16_t, true); }
#undef WIDE
    VA_LAUNCH_CHECK(op);
    return VLLM_AMD_OK;
  }
  if (dtype == VLLM_AMD_BF16) {
    if (es == 2) rc = launch_unified<bf16_t, false>(p, head_size, block_size, max_seqlen_q, max_seqlen_k, num_tokens, num_heads, num_seqs, st);
    else rc = launch_unified<bf16_t, true>(p, head_size, block_size, max_seqlen_q, max_seqlen_k, num_tokens, num_heads, num_seqs, st);
  } else {
    if (es == 2) rc = launch_unified<f16_t, false>(p, head_size, block_size, max_seqlen_q, max_seqlen_k, num_tokens, num_heads, num_seqs, st);
    else rc = launch_unified<f16_t, true>(p, head_size, block_size, max_seqlen_q, max_seqlen_k, num_tokens, num_heads, num_seqs, st);
  }
  if (rc != VLLM_AMD_OK) return rc;
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}
