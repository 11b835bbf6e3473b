// Internal interface between the paged-attention translation units.
#pragma once
#include "common.cuh"

namespace va {

struct PagedAttnParams {
  void* out;              // v1: final out; v2/rocm: final out
  float* exp_sums;        // [S, H, P] or null (v1)
  float* max_logits;      // [S, H, P] or null (v1)
  void* tmp_out;          // [S, H, P, D] or null (v1)
  const void* query;
  const void* key_cache;
  const void* value_cache;
  int num_seqs, num_heads, head_size, num_kv_heads;
  float scale;
  const int* block_tables;
  const int* seq_lens;
  const int* query_start_loc;  // rocm variant only, may be null
  int max_num_blocks_per_seq, block_size, max_seq_len;
  int partition_size;      // 0 = v1 (no partitioning)
  int max_num_partitions;  // P (1 for v1)
  int sp_parts = 1;        // reference partitions merged per workgroup / scratch slot stride
  const float* alibi_slopes;
  int dtype, kv_dtype;
  const float* k_scale;
  const float* v_scale;
  const float* fp8_out_scale;  // rocm variant only, may be null
  int64_t q_stride, kv_block_stride, kv_head_stride;
  int tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step;
  bool rocm_semantics;  // reduce order / out indexing of csrc/rocm/attention.cu
  int window = 0;       // sliding window of the decode token, 0 = none (vllm_amd_paged_attention_2d)
  hipStream_t stream;
};

// Generic (any supported head/block size, block-sparse) VALU path.
int paged_attention_generic(const PagedAttnParams& p);
// MFMA bandwidth path (head 64/128, block 16/32, 16-bit query). Returns
// VLLM_AMD_EUNSUPPORTED *without touching memory* when the shape is not covered.
bool paged_attention_mfma_supported(const PagedAttnParams& p);
int paged_attention_mfma(const PagedAttnParams& p);
// tuning knob (A/B measurements): see vllm_amd_set_tuning
void set_pa_variant(int v);
void set_pa_sp(int v);
void set_pa_group(int v);
void set_pa_balance(int v);
int get_pa_balance();
void set_pa_order(int v);
bool pa_order_enabled();
// Split-K merge of partitions (v2 / rocm).
int paged_attention_reduce(const PagedAttnParams& p);

}  // namespace va
