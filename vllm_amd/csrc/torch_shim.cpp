// PyTorch-ROCm operator shim over the C ABI (include/vllm_amd.h): the C++ side of
// torch.ops._C / _C_cache_ops / _rocm_C for the hot path, registered with the reference's
// schema strings (csrc/torch_bindings.cpp:45-69, 117-120, 188-211, 422-427, 575-593, 676-708;
// csrc/rocm/torch_bindings.cpp:34-56).  Every implementation unpacks data_ptr / sizes /
// strides, takes torch's current HIP stream and calls libvllm_amd.so; a non-zero status
// becomes the c10::Error (RuntimeError) the reference's TORCH_CHECK raises.  No tensor is
// allocated inside an op, except the one-time per-device workspace (ensure_workspace).
//
// Registration is a function (vllm_amd_torch_register), not a static TORCH_LIBRARY block:
// when the real vLLM extension was imported first its schemas already exist, and a static
// m.def() would abort the process at dlopen; here a duplicate definition is skipped and only
// the kernel for the CUDA(=HIP) key is (re)registered.
#include <torch/library.h>
#include <ATen/core/Tensor.h>
#include <ATen/ops/empty.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include "../../include/vllm_amd.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;
using DevGuard = at::hip::OptionalHIPGuardMasqueradingAsCUDA;

inline void check(int rc) { TORCH_CHECK(rc == 0, vllm_amd_last_error()); }

inline void* stream_of(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream();
}

inline void need_dev(const Tensor& t) {
  TORCH_CHECK(t.is_cuda(), "vllm_amd ops need device tensors (cuda/HIP); got ", t.device());
}
inline void need_dev(const OptTensor& t) { if (t.has_value()) need_dev(*t); }
template <class... Ts> inline void need_devs(const Ts&... ts) { (need_dev(ts), ...); }

inline int dt(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return VLLM_AMD_F32;
    case at::kHalf: return VLLM_AMD_F16;
    case at::kBFloat16: return VLLM_AMD_BF16;
    default: TORCH_CHECK(false, "Unsupported data type: ", t.scalar_type());
  }
}
inline bool is_fp8(const Tensor& t) {
  return t.scalar_type() == at::kByte || t.scalar_type() == at::kFloat8_e4m3fn;
}
inline bool is_16_32(const Tensor& t) {
  return t.scalar_type() == at::kFloat || t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16;
}
// csrc/quantization/fp8/amd/quant_utils.cuh:641-667
inline int kv(const std::string& s) {
  if (s == "auto") return VLLM_AMD_KV_AUTO;
  if (s == "fp8" || s == "fp8_e4m3") return VLLM_AMD_KV_FP8_E4M3;
  TORCH_CHECK(false, "Unsupported data type of kv cache: ", s);
}
inline const float* fptr(const Tensor& t) { return (const float*)t.data_ptr(); }
inline const float* fptr(const OptTensor& t) { return t.has_value() ? (const float*)t->data_ptr() : nullptr; }
inline void* vptr(const OptTensor& t) { return t.has_value() ? t->data_ptr() : nullptr; }

// ---- the library workspace (vllm_amd_set_workspace) --------------------------------------
// The C ABI allocates nothing, so this layer owns one buffer per device (4 stream slots x 16 MiB
// + ticket counters) and registers it once, outside stream capture, before that device's first
// scaled_mm -- and before its first paged_attention_v1 only when the balanced form ("pa_balance" = 1,
// off by default) is on: nothing else of that op uses it.  Without it the GEMM never splits over K.
// One host synchronisation, in that first call.  Launches under stream capture never take a workspace
// slot, so a captured GEMM runs unsplit: split-K (M <= 64, K >= 12288: 17.2 -> 15.2 us) is an
// eager-launch optimisation; bench.py's graph-timed GEMM legs time the unsplit form.
constexpr int64_t kWsBytes = (64ll << 20) + (512ll << 10);
std::mutex g_ws_mu;
Tensor g_ws[64];
std::atomic<bool> g_ws_set[64];

void ensure_workspace(const Tensor& t) {
  const int idx = t.get_device();
  if (idx < 0 || idx >= 64 || g_ws_set[idx].load(std::memory_order_acquire)) return;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (g_ws_set[idx].load(std::memory_order_relaxed)) return;
  hipStream_t s = c10::hip::getCurrentHIPStream(idx).stream();
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;   // not during capture; the next eager call registers it
  }
  DevGuard g(t.device());
  g_ws[idx] = at::empty({kWsBytes}, t.options().dtype(at::kByte));
  check(vllm_amd_set_workspace(g_ws[idx].data_ptr(), kWsBytes, (void*)s));
  // every stream's first launch must see the zeroed counters
  TORCH_CHECK(hipStreamSynchronize(s) == hipSuccess, "hipStreamSynchronize failed");
  g_ws_set[idx].store(true, std::memory_order_release);
}

// ---- attention ---------------------------------------------------------------------------
void paged_attention_v1(Tensor& out, const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
                        int64_t num_kv_heads, double scale, const Tensor& block_tables, const Tensor& seq_lens,
                        int64_t block_size, int64_t max_seq_len, const OptTensor& alibi_slopes,
                        const std::string& kv_cache_dtype, const Tensor& k_scale, const Tensor& v_scale,
                        int64_t tp_rank, int64_t bs_local_blocks, int64_t bs_vert_stride, int64_t bs_block_size,
                        int64_t bs_head_sliding_step) {
  need_devs(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes, k_scale, v_scale);
  TORCH_CHECK(query.dim() == 3 && out.is_contiguous(), "out.is_contiguous() && query.dim() == 3");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables / seq_lens must be int32");
  {
    int bal = 0;   // only the balanced form (opt-in) uses the workspace
    if (vllm_amd_get_tuning("pa_balance", &bal) == VLLM_AMD_OK && bal == 1) ensure_workspace(query);
  }
  DevGuard g(query.device());
  check(vllm_amd_paged_attention_v1(
      out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)query.size(0),
      (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale, (const int*)block_tables.data_ptr(),
      (const int*)seq_lens.data_ptr(), (int)block_tables.size(1), (int)block_size, (int)max_seq_len,
      fptr(alibi_slopes), dt(query), kv(kv_cache_dtype), fptr(k_scale), fptr(v_scale), query.stride(0),
      key_cache.stride(0), key_cache.stride(1), (int)tp_rank, (int)bs_local_blocks, (int)bs_vert_stride,
      (int)bs_block_size, (int)bs_head_sliding_step, stream_of(query)));
}

void paged_attention_v2(Tensor& out, Tensor& exp_sums, Tensor& max_logits, Tensor& tmp_out, const Tensor& query,
                        const Tensor& key_cache, const Tensor& value_cache, int64_t num_kv_heads, double scale,
                        const Tensor& block_tables, const Tensor& seq_lens, int64_t block_size, int64_t max_seq_len,
                        const OptTensor& alibi_slopes, const std::string& kv_cache_dtype, const Tensor& k_scale,
                        const Tensor& v_scale, int64_t tp_rank, int64_t bs_local_blocks, int64_t bs_vert_stride,
                        int64_t bs_block_size, int64_t bs_head_sliding_step) {
  need_devs(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, block_tables, seq_lens,
            alibi_slopes, k_scale, v_scale);
  TORCH_CHECK(query.dim() == 3 && out.is_contiguous() && tmp_out.is_contiguous() && exp_sums.is_contiguous() &&
                  max_logits.is_contiguous(), "out / tmp_out / exp_sums / max_logits must be contiguous");
  TORCH_CHECK(exp_sums.scalar_type() == at::kFloat && max_logits.scalar_type() == at::kFloat,
              "exp_sums / max_logits must be float32");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables / seq_lens must be int32");
  DevGuard g(query.device());
  check(vllm_amd_paged_attention_v2(
      out.data_ptr(), (float*)exp_sums.data_ptr(), (float*)max_logits.data_ptr(), tmp_out.data_ptr(),
      query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)query.size(0), (int)query.size(1),
      (int)query.size(2), (int)num_kv_heads, (float)scale, (const int*)block_tables.data_ptr(),
      (const int*)seq_lens.data_ptr(), (int)block_tables.size(1), (int)block_size, (int)max_seq_len,
      (int)exp_sums.size(-1), fptr(alibi_slopes), dt(query), kv(kv_cache_dtype), fptr(k_scale), fptr(v_scale),
      query.stride(0), key_cache.stride(0), key_cache.stride(1), (int)tp_rank, (int)bs_local_blocks,
      (int)bs_vert_stride, (int)bs_block_size, (int)bs_head_sliding_step, stream_of(query)));
}

void paged_attention_rocm(Tensor& out, const Tensor& exp_sums, const Tensor& max_logits, const Tensor& tmp_out,
                          const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
                          int64_t num_kv_heads, double scale, const Tensor& block_tables,
                          const Tensor& context_lens, const OptTensor& query_start_loc, int64_t block_size,
                          int64_t max_context_len, const OptTensor& alibi_slopes, const std::string& kv_cache_dtype,
                          const Tensor& k_scale, const Tensor& v_scale, const OptTensor& fp8_out_scale) {
  need_devs(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, block_tables, context_lens,
            query_start_loc, alibi_slopes, k_scale, v_scale, fp8_out_scale);
  TORCH_CHECK(query.dim() == 3 && out.is_contiguous() && tmp_out.is_contiguous() && exp_sums.is_contiguous(),
              "out / tmp_out / exp_sums must be contiguous");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && context_lens.scalar_type() == at::kInt,
              "block_tables / context_lens must be int32");
  TORCH_CHECK(!query_start_loc.has_value() || query_start_loc->scalar_type() == at::kInt,
              "query_start_loc must be int32");
  DevGuard g(query.device());
  check(vllm_amd_paged_attention_rocm(
      out.data_ptr(), (float*)exp_sums.data_ptr(), (float*)max_logits.data_ptr(), tmp_out.data_ptr(),
      query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)block_tables.size(0), (int)query.size(1),
      (int)query.size(2), (int)num_kv_heads, (float)scale, (const int*)block_tables.data_ptr(),
      (const int*)context_lens.data_ptr(), (const int*)vptr(query_start_loc), (int)block_tables.size(1),
      (int)block_size, (int)max_context_len, (int)exp_sums.size(-1), fptr(alibi_slopes), dt(query),
      kv(kv_cache_dtype), fptr(k_scale), fptr(v_scale), fptr(fp8_out_scale), query.stride(0), key_cache.stride(0),
      key_cache.stride(1), stream_of(query)));
}

// ---- cache ops ---------------------------------------------------------------------------
void reshape_and_cache(const Tensor& key, const Tensor& value, Tensor& key_cache, Tensor& value_cache,
                       const Tensor& slot_mapping, const std::string& kv_cache_dtype, const Tensor& k_scale,
                       const Tensor& v_scale) {
  need_devs(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale);
  const int64_t head_size = key.size(2);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && key.stride(2) == 1 && key.stride(1) == head_size,
              "slot_mapping must be int64 and key [num_tokens, num_heads, head_size] dense per token");
  DevGuard g(key.device());
  check(vllm_amd_reshape_and_cache(
      key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      (const int64_t*)slot_mapping.data_ptr(), (int)slot_mapping.size(0), (int)key.size(1), (int)head_size,
      (int)key_cache.size(3), (int)key_cache.size(4), key.stride(0), value.stride(0), dt(key), kv(kv_cache_dtype),
      fptr(k_scale), fptr(v_scale), stream_of(key)));
}

void reshape_and_cache_flash(const Tensor& key, const Tensor& value, Tensor& key_cache, Tensor& value_cache,
                             const Tensor& slot_mapping, const std::string& kv_cache_dtype, const Tensor& k_scale,
                             const Tensor& v_scale) {
  need_devs(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale);
  const int64_t head_size = key.size(2);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && key.stride(2) == 1 && key.stride(1) == head_size,
              "slot_mapping must be int64 and key [num_tokens, num_heads, head_size] dense per token");
  TORCH_CHECK(key_cache.stride(0) == value_cache.stride(0), "key_cache.stride(0) == value_cache.stride(0)");  // cache_kernels.cu:441
  DevGuard g(key.device());
  check(vllm_amd_reshape_and_cache_flash(
      key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      (const int64_t*)slot_mapping.data_ptr(), (int)slot_mapping.size(0), (int)key.size(1), (int)head_size,
      (int)key_cache.size(1), key.stride(0), value.stride(0), key_cache.stride(0), key_cache.stride(1),
      key_cache.stride(2), dt(key), kv(kv_cache_dtype), fptr(k_scale), fptr(v_scale), stream_of(key)));
}

void convert_fp8(Tensor& dst_cache, const Tensor& src_cache, double scale, const std::string& kv_cache_dtype) {
  need_devs(dst_cache, src_cache);
  (void)kv(kv_cache_dtype);   // validates the string; "auto" is treated as e4m3 (gfx950 has one fp8 KV type)
  int to_fp8, d;
  if (is_16_32(src_cache) && is_fp8(dst_cache)) { to_fp8 = 1; d = dt(src_cache); }
  else if (is_fp8(src_cache) && is_16_32(dst_cache)) { to_fp8 = 0; d = dt(dst_cache); }
  else TORCH_CHECK(false, "Unsupported data type: ", src_cache.scalar_type(), " -> ", dst_cache.scalar_type());
  TORCH_CHECK(src_cache.is_contiguous() && dst_cache.is_contiguous(), "convert_fp8: contiguous tensors required");
  DevGuard g(src_cache.device());
  check(vllm_amd_convert_fp8(dst_cache.data_ptr(), src_cache.data_ptr(), src_cache.numel(), (float)scale, d, to_fp8,
                             stream_of(src_cache)));
}

// csrc/cache_kernels.cu:123-170; one launch per layer (no pointer table upload, no sync)
void copy_blocks(const std::vector<Tensor>& key_caches, const std::vector<Tensor>& value_caches,
                 const Tensor& block_mapping) {
  TORCH_CHECK(key_caches.size() == value_caches.size(), "num_layers == value_caches.size()");
  if (key_caches.empty()) return;
  need_dev(block_mapping);
  TORCH_CHECK(block_mapping.scalar_type() == at::kLong && block_mapping.is_contiguous(),
              "block_mapping must be a contiguous int64 tensor");
  const int num_pairs = (int)block_mapping.size(0);
  for (size_t i = 0; i < key_caches.size(); i++) {
    const Tensor& kc = key_caches[i];
    const Tensor& vc = value_caches[i];
    need_devs(kc, vc);
    DevGuard g(kc.device());
    check(vllm_amd_copy_blocks(kc.data_ptr(), vc.data_ptr(), (const int64_t*)block_mapping.data_ptr(), num_pairs,
                               kc.stride(0) * (int64_t)kc.element_size(), stream_of(kc)));
  }
}

// csrc/cache_kernels.cu:25-67: per-pair async copies on the current stream of the device side;
// block_mapping lives on the CPU
void swap_blocks(const Tensor& src, Tensor& dst, const Tensor& block_mapping) {
  TORCH_CHECK(block_mapping.device().is_cpu(), "block_mapping must be on CPU");
  TORCH_CHECK(!(src.is_cuda() && dst.is_cuda()) || src.device() == dst.device(),
              "src and dst must be on the same GPU");
  TORCH_CHECK(src.is_cuda() || dst.is_cuda(), "Invalid device combination");
  Tensor bm = block_mapping.to(at::kLong).contiguous();
  const Tensor& gpu = src.is_cuda() ? src : dst;
  DevGuard g(gpu.device());
  check(vllm_amd_swap_blocks(src.data_ptr(), dst.data_ptr(), (int64_t)src.element_size() * src.stride(0),
                             (const int64_t*)bm.data_ptr(), bm.size(0), stream_of(gpu)));
}

// ---- fp8 quantisation --------------------------------------------------------------------
inline void fp8_out(const Tensor& result) {
  TORCH_CHECK(is_fp8(result), "Unsupported fp8 type: ", result.scalar_type());
  TORCH_CHECK(result.is_contiguous(), "out.is_contiguous()");
}
inline int64_t rows_of(const Tensor& input, int64_t hidden) { return hidden ? input.numel() / hidden : 0; }

inline int64_t quant_common(const Tensor& result, const Tensor& input) {
  need_devs(result, input);
  TORCH_CHECK(input.is_contiguous(), "input.is_contiguous()");
  TORCH_CHECK(result.is_contiguous(), "out.is_contiguous()");
  TORCH_CHECK(is_fp8(result), "Unsupported fp8 type: ", result.scalar_type());
  return rows_of(input, input.size(-1));
}

void static_scaled_fp8_quant(Tensor& result, const Tensor& input, const Tensor& scale) {
  const int64_t nt = quant_common(result, input);
  DevGuard g(input.device());
  check(vllm_amd_static_scaled_fp8_quant(result.data_ptr(), input.data_ptr(), fptr(scale), nt, (int)input.size(-1),
                                         dt(input), stream_of(input)));
}

void dynamic_scaled_fp8_quant(Tensor& result, const Tensor& input, Tensor& scale) {
  const int64_t nt = quant_common(result, input);
  DevGuard g(input.device());
  check(vllm_amd_dynamic_scaled_fp8_quant(result.data_ptr(), input.data_ptr(), (float*)scale.data_ptr(), nt,
                                          (int)input.size(-1), dt(input), stream_of(input)));
}

void dynamic_per_token_scaled_fp8_quant(Tensor& result, const Tensor& input, Tensor& scale,
                                        const OptTensor& scale_ub) {
  const int64_t nt = quant_common(result, input);
  DevGuard g(input.device());
  check(vllm_amd_dynamic_per_token_scaled_fp8_quant(result.data_ptr(), input.data_ptr(), (float*)scale.data_ptr(),
                                                    fptr(scale_ub), nt, (int)input.size(-1), dt(input),
                                                    stream_of(input)));
}

// ---- GEMM --------------------------------------------------------------------------------
// csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu checks, restated
void cutlass_scaled_mm(Tensor& out, const Tensor& a, const Tensor& b, const Tensor& a_scales,
                       const Tensor& b_scales, const OptTensor& bias) {
  need_devs(out, a, b, a_scales, b_scales, bias);
  ensure_workspace(a);
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && out.dim() == 2, "a.dim() == 2 && b.dim() == 2 && c.dim() == 2");
  const int64_t M = a.size(0), K = a.size(1), N = b.size(1);
  TORCH_CHECK(out.size(0) == M && b.size(0) == K && out.size(1) == N,
              "c.size(0) == a.size(0) && a.size(1) == b.size(0) && b.size(1) == c.size(1)");
  TORCH_CHECK(a.stride(1) == 1 && out.stride(1) == 1, "a.stride(1) == 1 && c.stride(1) == 1");  // row major
  TORCH_CHECK(b.stride(0) == 1, "b.stride(0) == 1");                                             // column major
  TORCH_CHECK(is_fp8(a) && is_fp8(b), "scaled_mm: only fp8 e4m3 x fp8 e4m3 is implemented on gfx950");
  TORCH_CHECK(a_scales.scalar_type() == at::kFloat && b_scales.scalar_type() == at::kFloat,
              "scales must be float32");
  TORCH_CHECK(a_scales.is_contiguous() && b_scales.is_contiguous(),
              "a_scales.is_contiguous() && b_scales.is_contiguous()");
  TORCH_CHECK(!bias.has_value() || (bias->numel() == N && bias->is_contiguous() && bias->dim() == 1 &&
                                    bias->scalar_type() == out.scalar_type()),
              "bias.numel() == b.size(1) && bias.is_contiguous() && bias.dim() == 1");
  DevGuard g(a.device());
  check(vllm_amd_scaled_mm(out.data_ptr(), a.data_ptr(), b.data_ptr(), fptr(a_scales), fptr(b_scales), vptr(bias),
                           (int)M, (int)N, (int)K, a.stride(0), b.stride(1), out.stride(0), (int)a_scales.numel(),
                           (int)b_scales.numel(), dt(out), stream_of(a)));
}

void wvSplitKQ(const Tensor& in_a, const Tensor& in_b, Tensor& out_c, const Tensor& scale_a, const Tensor& scale_b,
               int64_t cu_count) {
  need_devs(in_a, in_b, out_c, scale_a, scale_b);
  TORCH_CHECK(in_a.dim() == 2 && in_b.dim() == 2, "wvSplitKQ: 2-D operands required");
  TORCH_CHECK(in_a.is_contiguous() && in_b.is_contiguous() && out_c.is_contiguous(),
              "wvSplitKQ: contiguous tensors required");
  TORCH_CHECK(in_b.size(1) == in_a.size(1), "in_a.size(1) == in_b.size(1)");
  DevGuard g(in_a.device());
  check(vllm_amd_wvSplitKQ(in_a.data_ptr(), in_b.data_ptr(), out_c.data_ptr(), fptr(scale_a), fptr(scale_b),
                           (int)in_a.size(0), (int)in_a.size(1), (int)in_b.size(0), dt(out_c), (int)cu_count,
                           stream_of(in_a)));
}

// ---- fused neighbours -------------------------------------------------------
// csrc/layernorm_quant_kernels.cu:172-198
void rms_norm_static_fp8_quant(Tensor& result, const Tensor& input, const Tensor& weight, const Tensor& scale,
                               double epsilon) {
  need_devs(result, input, weight, scale);
  fp8_out(result);
  const int64_t hidden = input.size(-1);
  DevGuard g(input.device());
  check(vllm_amd_rms_norm_static_fp8_quant(result.data_ptr(), input.data_ptr(), weight.data_ptr(), fptr(scale),
                                           (float)epsilon, rows_of(input, hidden), (int)hidden,
                                           input.dim() >= 2 ? input.stride(-2) : hidden, dt(input),
                                           stream_of(input)));
}

// csrc/layernorm_quant_kernels.cu:214-253
void fused_add_rms_norm_static_fp8_quant(Tensor& result, const Tensor& input, Tensor& residual,
                                         const Tensor& weight, const Tensor& scale, double epsilon) {
  need_devs(result, input, residual, weight, scale);
  fp8_out(result);
  TORCH_CHECK(residual.is_contiguous(), "residual.is_contiguous()");
  const int64_t hidden = input.size(-1);
  DevGuard g(input.device());
  check(vllm_amd_fused_add_rms_norm_static_fp8_quant(
      result.data_ptr(), input.data_ptr(), residual.data_ptr(), weight.data_ptr(), fptr(scale), (float)epsilon,
      rows_of(input, hidden), (int)hidden, input.dim() >= 2 ? input.stride(-2) : hidden, dt(input),
      stream_of(input)));
}

// csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu:132-155
void rms_norm_dynamic_per_token_quant(Tensor& result, const Tensor& input, const Tensor& weight, Tensor& scale,
                                      double epsilon, const OptTensor& scale_ub, OptTensor residual) {
  need_devs(result, input, weight, scale, scale_ub, residual);
  const bool i8 = result.scalar_type() == at::kChar;
  if (i8) {
    // fused_layernorm_dynamic_per_token_quant.cu:139-146: int8 or fp8 output, scale_ub only with fp8
    TORCH_CHECK(!scale_ub.has_value(), "rms_norm_dynamic_per_token_quant: scale_ub needs an fp8 output");
    TORCH_CHECK(result.is_contiguous(), "out.is_contiguous()");
  } else {
    fp8_out(result);
  }
  TORCH_CHECK(input.is_contiguous(), "input.is_contiguous()");
  TORCH_CHECK(scale.scalar_type() == at::kFloat, "scales.dtype() == torch::kFloat32");
  TORCH_CHECK(!residual.has_value() || residual->is_contiguous(), "residual.is_contiguous()");
  const int64_t hidden = input.size(-1);
  DevGuard g(input.device());
  if (i8)
    check(vllm_amd_rms_norm_dynamic_per_token_quant_int8(result.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                                         (float*)scale.data_ptr(), (float)epsilon, vptr(residual),
                                                         rows_of(input, hidden), (int)hidden, dt(input),
                                                         stream_of(input)));
  else
    check(vllm_amd_rms_norm_dynamic_per_token_quant(result.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                                    (float*)scale.data_ptr(), (float)epsilon, fptr(scale_ub),
                                                    vptr(residual), rows_of(input, hidden), (int)hidden, dt(input),
                                                    stream_of(input)));
}

// csrc/quantization/activation_kernels.cu:112-121
void silu_and_mul_quant(Tensor& result, const Tensor& input, const Tensor& scale) {
  need_devs(result, input, scale);
  fp8_out(result);
  TORCH_CHECK(input.scalar_type() == at::kHalf || input.scalar_type() == at::kBFloat16,
              "input.dtype() == torch::kFloat16 || input.dtype() == torch::kBFloat16");
  TORCH_CHECK(input.size(-1) % 2 == 0, "input.size(-1) % 2 == 0");
  TORCH_CHECK(input.is_contiguous(), "input.is_contiguous()");
  const int64_t d = input.size(-1) / 2;
  DevGuard g(input.device());
  check(vllm_amd_silu_and_mul_quant(result.data_ptr(), input.data_ptr(), fptr(scale),
                                    d ? input.numel() / input.size(-1) : 0, (int)d, dt(input), stream_of(input)));
}

// Device-side BlockTable.compute_slot_mapping (vllm/v1/worker/block_table.py:84-99).  Not a
// reference op (the reference does this with numpy on the host): namespace _vllm_amd_C.
void compute_slot_mapping(Tensor& slot_mapping, const Tensor& block_table, const Tensor& req_indices,
                          const Tensor& positions, int64_t block_size) {
  need_devs(slot_mapping, block_table, req_indices, positions);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && positions.scalar_type() == at::kLong &&
                  block_table.scalar_type() == at::kInt && req_indices.scalar_type() == at::kInt,
              "compute_slot_mapping: slot_mapping / positions int64, block_table / req_indices int32");
  TORCH_CHECK(block_table.dim() == 2 && block_table.is_contiguous() && req_indices.is_contiguous() &&
                  positions.is_contiguous() && slot_mapping.is_contiguous(),
              "compute_slot_mapping: contiguous tensors required");
  const int64_t n = req_indices.size(0);
  TORCH_CHECK(slot_mapping.size(0) >= n && positions.size(0) >= n, "compute_slot_mapping: outputs shorter than n");
  DevGuard g(block_table.device());
  check(vllm_amd_compute_slot_mapping((int64_t*)slot_mapping.data_ptr(), (const int32_t*)block_table.data_ptr(),
                                      (const int32_t*)req_indices.data_ptr(), (const int64_t*)positions.data_ptr(),
                                      n, (int)block_table.size(1), (int)block_size, stream_of(block_table)));
}

// ---- registration ------------------------------------------------------------------------
#define PA_TAIL                                                                                               \
  " int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, int block_size,"                   \
  " int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, Tensor k_scale,"                              \
  " Tensor v_scale, int tp_rank, int blocksparse_local_blocks,"                                              \
  " int blocksparse_vert_stride, int blocksparse_block_size,"                                                \
  " int blocksparse_head_sliding_step) -> ()"

std::vector<std::unique_ptr<torch::Library>> g_libs;   // kept alive for the life of the process

struct Registrar {
  const char* ns;
  torch::Library* frag;
  torch::Library* impl;
  torch::Library* impl_any;
  template <class F>
  void op(const char* name, const char* schema, F* fn, bool any_device = false) {
    try {
      frag->def(schema);
    } catch (const c10::Error&) {
      // already defined (the real vLLM extension was imported first): override the kernel only
    }
    (any_device ? impl_any : impl)->impl(name, fn);
  }
};

Registrar open_ns(const char* ns) {
  auto mk = [&](torch::Library::Kind kind, std::optional<c10::DispatchKey> k) {
    g_libs.emplace_back(new torch::Library(kind, ns, k, __FILE__, __LINE__));
    return g_libs.back().get();
  };
  return Registrar{ns, mk(torch::Library::FRAGMENT, std::nullopt),
                   mk(torch::Library::IMPL, c10::DispatchKey::CUDA),
                   mk(torch::Library::IMPL, c10::DispatchKey::CompositeExplicitAutograd)};
}

void register_all() {
  Registrar c = open_ns("_C");
  c.op("paged_attention_v1",
       "paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache," PA_TAIL,
       &paged_attention_v1);
  c.op("paged_attention_v2",
       "paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out,"
       " Tensor query, Tensor key_cache, Tensor value_cache," PA_TAIL,
       &paged_attention_v2);
  c.op("static_scaled_fp8_quant", "static_scaled_fp8_quant(Tensor! result, Tensor input, Tensor scale) -> ()",
       &static_scaled_fp8_quant);
  c.op("dynamic_scaled_fp8_quant", "dynamic_scaled_fp8_quant(Tensor! result, Tensor input, Tensor! scale) -> ()",
       &dynamic_scaled_fp8_quant);
  c.op("dynamic_per_token_scaled_fp8_quant",
       "dynamic_per_token_scaled_fp8_quant(Tensor! result, Tensor input, Tensor! scale,"
       " Tensor? scale_ub) -> ()",
       &dynamic_per_token_scaled_fp8_quant);
  c.op("cutlass_scaled_mm",
       "cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales,"
       " Tensor? bias) -> ()",
       &cutlass_scaled_mm);
  c.op("silu_and_mul_quant", "silu_and_mul_quant(Tensor! result, Tensor input, Tensor scale) -> ()",
       &silu_and_mul_quant);
  c.op("rms_norm_static_fp8_quant",
       "rms_norm_static_fp8_quant(Tensor! result, Tensor input, Tensor weight, Tensor scale,"
       " float epsilon) -> ()",
       &rms_norm_static_fp8_quant);
  c.op("fused_add_rms_norm_static_fp8_quant",
       "fused_add_rms_norm_static_fp8_quant(Tensor! result, Tensor input, Tensor! residual,"
       " Tensor weight, Tensor scale, float epsilon) -> ()",
       &fused_add_rms_norm_static_fp8_quant);
  c.op("rms_norm_dynamic_per_token_quant",
       "rms_norm_dynamic_per_token_quant(Tensor! result, Tensor input, Tensor weight,"
       " Tensor! scale, float epsilon, Tensor? scale_ub, Tensor!? residual) -> ()",
       &rms_norm_dynamic_per_token_quant);

  Registrar k = open_ns("_C_cache_ops");
  k.op("reshape_and_cache",
       "reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache,"
       " Tensor slot_mapping, str kv_cache_dtype, Tensor k_scale, Tensor v_scale) -> ()",
       &reshape_and_cache);
  k.op("reshape_and_cache_flash",
       "reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache,"
       " Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype, Tensor k_scale,"
       " Tensor v_scale) -> ()",
       &reshape_and_cache_flash);
  k.op("convert_fp8", "convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()",
       &convert_fp8);
  k.op("copy_blocks",
       "copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches,"
       " Tensor block_mapping) -> ()",
       &copy_blocks);
  k.op("swap_blocks", "swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()", &swap_blocks,
       /*any_device=*/true);

  Registrar r = open_ns("_rocm_C");
  r.op("paged_attention",
       "paged_attention(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out,"
       " Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale,"
       " Tensor block_tables, Tensor context_lens, Tensor? query_start_loc, int block_size,"
       " int max_context_len, Tensor? alibi_slopes, str kv_cache_dtype, Tensor k_scale,"
       " Tensor v_scale, Tensor? fp8_out_scale) -> ()",
       &paged_attention_rocm);
  r.op("wvSplitKQ",
       "wvSplitKQ(Tensor in_a, Tensor in_b, Tensor! out_c, Tensor scale_a, Tensor scale_b,"
       " int CuCount) -> ()",
       &wvSplitKQ);

  Registrar x = open_ns("_vllm_amd_C");
  x.op("compute_slot_mapping",
       "compute_slot_mapping(Tensor! slot_mapping, Tensor block_table, Tensor req_indices,"
       " Tensor positions, int block_size) -> ()",
       &compute_slot_mapping);
}

std::string g_reg_error;
bool g_registered = false;

}  // namespace

// Called once by vllm_amd/_torch_ops.py after dlopen (in tree: by whatever loads the plugin).
// Returns 0, or -1 with vllm_amd_torch_register_error() holding the message.
extern "C" int vllm_amd_torch_register(void) {
  if (g_registered) return 0;
  try {
    register_all();
    g_registered = true;
    return 0;
  } catch (const std::exception& e) {
    g_reg_error = e.what();
    return -1;
  }
}

extern "C" const char* vllm_amd_torch_register_error(void) { return g_reg_error.c_str(); }
