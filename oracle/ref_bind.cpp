// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
//
// Binding shim for oracle/_ref/librefcpu.so: the reference's OWN CPU sources
// (csrc/cpu/attention.cpp, csrc/cpu/cache.cpp) are compiled where they lie
// under /root/reference by oracle/Makefile; this file only declares their
// entry points and registers them in a private torch namespace "refcpu" so the
// tests / bench cpu_baseline can call them.  Schemas follow the reference's
// csrc/cpu/torch_bindings.cpp:80-104 (attention) and :277-283 (cache).
#include <torch/library.h>
#include <torch/types.h>

#include <optional>
#include <string>

void paged_attention_v1(
    torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
    int64_t max_seq_len, const std::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, torch::Tensor& k_scale,
    torch::Tensor& v_scale, const int64_t tp_rank,
    const int64_t blocksparse_local_blocks,
    const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
    const int64_t blocksparse_head_sliding_step);

void paged_attention_v2(
    torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
    torch::Tensor& tmp_out, torch::Tensor& query, torch::Tensor& key_cache,
    torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
    torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
    int64_t max_seq_len, const std::optional<torch::Tensor>& alibi_slopes,
    const std::string& kv_cache_dtype, torch::Tensor& k_scale,
    torch::Tensor& v_scale, const int64_t tp_rank,
    const int64_t blocksparse_local_blocks,
    const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
    const int64_t blocksparse_head_sliding_step);

void reshape_and_cache(torch::Tensor& key, torch::Tensor& value,
                       torch::Tensor& key_cache, torch::Tensor& value_cache,
                       torch::Tensor& slot_mapping,
                       const std::string& kv_cache_dtype,
                       torch::Tensor& k_scale, torch::Tensor& v_scale);

TORCH_LIBRARY(refcpu, m) {
  m.def(
      "paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache,"
      " Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables,"
      " Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes,"
      " str kv_cache_dtype, Tensor k_scale, Tensor v_scale, int tp_rank,"
      " int blocksparse_local_blocks, int blocksparse_vert_stride,"
      " int blocksparse_block_size, int blocksparse_head_sliding_step) -> ()");
  m.impl("paged_attention_v1", torch::kCPU, &paged_attention_v1);
  m.def(
      "paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits,"
      " Tensor! tmp_out, Tensor query, Tensor key_cache, Tensor value_cache,"
      " int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens,"
      " int block_size, int max_seq_len, Tensor? alibi_slopes,"
      " str kv_cache_dtype, Tensor k_scale, Tensor v_scale, int tp_rank,"
      " int blocksparse_local_blocks, int blocksparse_vert_stride,"
      " int blocksparse_block_size, int blocksparse_head_sliding_step) -> ()");
  m.impl("paged_attention_v2", torch::kCPU, &paged_attention_v2);
  m.def(
      "reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache,"
      " Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype,"
      " Tensor k_scale, Tensor v_scale) -> ()");
  m.impl("reshape_and_cache", torch::kCPU, &reshape_and_cache);
}
