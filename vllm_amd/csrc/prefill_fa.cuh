// unified_attention, long query chunks at head size 128: the 256-row tile kernel of prefill_fa.hip
// (launched from prefill_attention.hip's launch_unified when its conditions hold).
#pragma once
#include "common.cuh"

namespace va {

struct FaParams {
  void* o; const void* q;
  const uint8_t* k_cache; const uint8_t* v_cache;        // flash layout [block][slot][kv head][d], any strides (prefix: below)
  const int* block_table; const int* cu_seqlens_q; const int* seq_lens;
  const float* k_scale; const float* v_scale;             // fp8 cache only
  const float* out_scale;                                  // != null: e4m3 output
  int num_heads, num_kv_heads, num_seqs, num_tokens;
  int head_size;       // 72 .. 256 (padded to 128, above 128 to 256, inside the kernel)
  int bs_shift;        // log2(block size)
  int skip_decode;     // sequences of query length 1 were taken by the decode kernel
  int pack;            // q heads per tile-row group: the GQA group (rows = (token, head) pairs) or 1
  int64_t bt_stride, q_stride_t, q_stride_h, o_stride_t, o_stride_h;
  int64_t kc_block_bytes, kc_slot_bytes, kc_head_bytes, vc_block_bytes, vc_slot_bytes, vc_head_bytes;
  float sm_scale;
  float softcap;       // unified_attention: scores softcap * tanh(score / softcap); 0 = none (prefix: 0)
  int window;          // unified_attention: 1 + window_size[0] of the reference call, 0 = none (prefix: 0)
  // context_attention_fwd (prefix = 1): the caches are the paged CONTEXT in the x layout -- K [block][kv head][D / x][BS][x]
  // (x = 16 bytes), V [block][kv head][D][BS]; kc / vc_slot_bytes unused -- and the new tokens' keys / values are dense
  int prefix;
  const void* k_new; const void* v_new;
  int64_t k_stride_t, k_stride_h, v_stride_t, v_stride_h;   // elements
  int blocks_per_seq;  // > 0: num_tokens unknown, grid.x = blocks_per_seq * num_seqs (ceil(max query length / tokens per block))
  int stagger;         // 1 = waves 4-7 stage at the top of the interval (tuning "pf_fa_stagger", default 1)
  int abl;             // (unused; the kernel's ablations are compile-time: -DVA_FA_ABL=bits, garbage results: 1 no exp2,
                       // 2 no staging after the first tiles, 4 no S^T MFMAs, 8 no PV MFMAs, 16 no barrier)
};

// dtype: VLLM_AMD_F16 / VLLM_AMD_BF16 (queries, output, and the cache unless fp8)
int launch_unified_fa(const FaParams& p, int dtype, bool fp8, hipStream_t st);

}  // namespace va
