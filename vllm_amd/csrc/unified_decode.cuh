// Internal interface: decode-only launches of unified_attention (unified_decode.hip).
#pragma once
#include "common.cuh"

namespace va {

struct UniDecodeParams {
  void* out; const void* q;
  const uint8_t* k_cache; const uint8_t* v_cache;       // flash layout [block][slot][kv head][d]
  const int* cu_seqlens_q; const int* seq_lens; const int* block_table;
  const float* k_scale; const float* v_scale; const float* alibi_slopes; const float* out_scale;
  int num_seqs, num_heads, num_kv_heads, head_size, block_size;
  int64_t bt_stride, q_stride_t, q_stride_h, o_stride_t, o_stride_h;   // elements
  int64_t kc_block_bytes, kc_slot_bytes, kc_head_bytes, vc_block_bytes, vc_slot_bytes, vc_head_bytes;
  float sm_scale, softcap;
  int window;            // 0 = none; key n visible iff seq_len - 1 - n < window
  int max_seq_len = 0;   // upper bound of seq_lens if the caller knows one (0 = unknown): sizes the keys of the longest-first order
  // split over the keys: segment s takes the 64-token tiles [s tps, (s+1) tps), tps =
  // cdiv(cdiv(seq_len, 64), num_segments) -- the partition unified_reduce_kernel assumes
  float* segm_out; float* segm_max; float* segm_sum;
  int num_segments, segm_stride;
  int dtype;             // VLLM_AMD_F16 / VLLM_AMD_BF16
  bool fp8;              // e4m3 cache
  hipStream_t stream;
  int hp = 1;            // adjacent kv heads taken as one row (unified_decode_heads_per_row: 1, 2 or 4)
};

bool pa_order_enabled();   // tuning "pa_order" (paged_attention_mfma.hip)

// true when every query of the launch may take the decode kernel (the caller has checked
// max_seqlen_q == 1); false WITHOUT touching memory otherwise
bool unified_decode_supported(const UniDecodeParams& p);
// how many adjacent kv heads a workgroup row may take (1 = none): set UniDecodeParams::hp to it before unified_decode
int unified_decode_heads_per_row(const UniDecodeParams& p);
// main kernel and, with num_segments > 1 (at most 16), the merge of the segment partials
int unified_decode(const UniDecodeParams& p);

}  // namespace va
