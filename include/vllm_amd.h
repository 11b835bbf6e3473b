/*
 * vllm_amd.h -- C ABI of libvllm_amd.so: the MI355X (gfx950) quantized decode
 * hot path of vLLM, hand-written HIP.
 *
 * Every entry point replaces ONE reference operator (cited per function,
 * paths relative to the ROCm/vllm tree).  Conventions, identical to the
 * reference's op contract:
 *   - all pointers are DEVICE pointers owned by the caller (including the
 *     temporaries tmp_out/exp_sums/max_logits, the pre-zeroed dynamic scale
 *     and the optional split-K workspace of scaled_mm); nothing is allocated,
 *     freed or synchronised inside a call, so every call is HIP-graph
 *     capturable;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) of the CURRENT device;
 *   - strides are in ELEMENTS of the tensor they describe;
 *   - the return value is 0 on success or a negative VLLM_AMD_E* code; the
 *     message (the text the reference would put in its TORCH_CHECK) is
 *     returned by vllm_amd_last_error() for the calling thread;
 *   - calls are not re-entrant per stream but are thread-safe across threads.
 */
#ifndef VLLM_AMD_H_
#define VLLM_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation / query / output element types (reference: VLLM_DISPATCH_FLOATING_TYPES,
 * csrc/dispatch_utils.h) */
enum { VLLM_AMD_F32 = 0, VLLM_AMD_F16 = 1, VLLM_AMD_BF16 = 2 };
/* KV-cache storage (reference: Fp8KVCacheDataType, csrc/attention/dtype_fp8.cuh:14-18;
 * strings "auto" | "fp8" | "fp8_e4m3", csrc/quantization/fp8/amd/quant_utils.cuh:641-667).
 * gfx950 fp8 is OCP e4m3fn (max 448). */
enum { VLLM_AMD_KV_AUTO = 0, VLLM_AMD_KV_FP8_E4M3 = 1 };

enum {
  VLLM_AMD_OK = 0,
  VLLM_AMD_EINVAL = -1,      /* bad argument (null pointer, negative size ...) */
  VLLM_AMD_EUNSUPPORTED = -2,/* "Unsupported head size / block size / dtype ..." */
  VLLM_AMD_EHIP = -3         /* a HIP runtime call failed */
};

const char* vllm_amd_last_error(void);
/* 10000*major + 100*minor + patch */
int vllm_amd_version(void);

/* ------------------------------------------------------------------------- */
/* Paged attention (decode: one query token per sequence).                    */
/* key_cache  [num_blocks, num_kv_heads, head_size/x, block_size, x]          */
/* value_cache[num_blocks, num_kv_heads, head_size, block_size]               */
/* x = 16 / sizeof(cache element); fp8 caches are bytes, x = 16.              */
/* ------------------------------------------------------------------------- */

/* Replaces torch.ops._C.paged_attention_v1
 *   schema  csrc/torch_bindings.cpp:45-55, launcher csrc/attention/paged_attention_v1.cu:46-182,
 *   kernel  csrc/attention/attention_kernels.cuh:85-522, python vllm/_custom_ops.py:39-65.
 * out/query: [num_seqs, num_heads, head_size] (query row stride q_stride, out contiguous).
 * k_scale/v_scale: device float[1].  alibi_slopes: device float[num_heads] or NULL.
 * blocksparse_* as in the reference (vert_stride <= 1 disables). */
int vllm_amd_paged_attention_v1(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq,
    int block_size, int max_seq_len, const float* alibi_slopes, int dtype,
    int kv_dtype, const float* k_scale, const float* v_scale, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int tp_rank,
    int blocksparse_local_blocks, int blocksparse_vert_stride,
    int blocksparse_block_size, int blocksparse_head_sliding_step, void* stream);

/* Replaces torch.ops._C.paged_attention_v2
 *   schema csrc/torch_bindings.cpp:58-69, launcher csrc/attention/paged_attention_v2.cu:46-192,
 *   kernels csrc/attention/attention_kernels.cuh:529-664, python vllm/_custom_ops.py:68-97.
 * exp_sums/max_logits: float [num_seqs, num_heads, max_num_partitions];
 * tmp_out: dtype [num_seqs, num_heads, max_num_partitions, head_size];
 * partition = 512 tokens (vllm/attention/ops/paged_attn.py:16). */
int vllm_amd_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq,
    int block_size, int max_seq_len, int max_num_partitions,
    const float* alibi_slopes, int dtype, int kv_dtype, const float* k_scale,
    const float* v_scale, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, void* stream);

/* Replaces torch.ops._rocm_C.paged_attention
 *   schema csrc/rocm/torch_bindings.cpp:42-56, entry csrc/rocm/attention.cu:3494-3539,
 *   launcher :3073-3228, python vllm/_custom_ops.py:100-126.
 * partition = 256 tokens; num_seqs = block_tables rows; query_start_loc
 * (int[num_seqs+1] or NULL) selects the query/out row of each sequence and
 * sequences whose query length != 1 are skipped; fp8_out_scale (float[1] or
 * NULL): when given, `out` is e4m3 bytes = out / *fp8_out_scale. */
int vllm_amd_paged_attention_rocm(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* context_lens, const int* query_start_loc,
    int max_num_blocks_per_seq, int block_size, int max_context_len,
    int max_num_partitions, const float* alibi_slopes, int dtype, int kv_dtype,
    const float* k_scale, const float* v_scale, const float* fp8_out_scale,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    void* stream);

/* The decode side of chunked_prefill_paged_decode when a sliding window is set: the
 * reference's Triton kernel_paged_attention_2d (vllm/attention/ops/chunked_prefill_paged_decode.py
 * :27-212, launched at :340-381; its custom op has no window, platforms/rocm.py:139-140).
 * Arguments of vllm_amd_paged_attention_rocm plus sliding_window (<= 0: none): the token at
 * position seq_len - 1 attends to the keys [seq_len - sliding_window, seq_len) (:167-171).
 * Sequences whose query length is not 1 are skipped when query_start_loc is given (:71-78);
 * fp8_out_scale as :201-203.  Scratch as the rocm op (partition 256). */
int vllm_amd_paged_attention_2d(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* seq_lens, const int* query_start_loc,
    int max_num_blocks_per_seq, int block_size, int max_seq_len,
    int max_num_partitions, const float* alibi_slopes, int dtype, int kv_dtype,
    const float* k_scale, const float* v_scale, const float* fp8_out_scale,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int sliding_window, void* stream);

/* ------------------------------------------------------------------------- */
/* KV-cache writes                                                            */
/* ------------------------------------------------------------------------- */

/* Replaces torch.ops._C_cache_ops.reshape_and_cache
 *   schema csrc/torch_bindings.cpp:676-682, csrc/cache_kernels.cu:210-262, 369-395,
 *   python vllm/_custom_ops.py:1617-1629.  slot_mapping: int64[num_tokens], <0 = skip. */
int vllm_amd_reshape_and_cache(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
    int kv_dtype, const float* k_scale, const float* v_scale, void* stream);

/* Replaces torch.ops._C_cache_ops.reshape_and_cache_flash
 *   schema csrc/torch_bindings.cpp:685-693, csrc/cache_kernels.cu:264-306, 412-450,
 *   python vllm/_custom_ops.py:1632-1645.
 * caches [num_blocks, block_size, num_heads, head_size] addressed through
 * block_stride/page_stride/head_stride (NHD or permuted HND views). */
int vllm_amd_reshape_and_cache_flash(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t key_stride, int64_t value_stride,
    int64_t block_stride, int64_t page_stride, int64_t head_stride, int dtype,
    int kv_dtype, const float* k_scale, const float* v_scale, void* stream);

/* Replaces torch.ops._C_cache_ops.convert_fp8
 *   schema csrc/torch_bindings.cpp:705-708, csrc/cache_kernels.cu:505-576,
 *   python vllm/_custom_ops.py:1677-1681.
 * to_fp8 != 0: src (dtype) -> dst (e4m3 bytes), x / scale; else dst (dtype) =
 * float(src byte) * scale.  n = number of elements. */
int vllm_amd_convert_fp8(void* dst, const void* src, int64_t n, float scale,
                         int dtype, int to_fp8, void* stream);

/* Replaces torch.ops._C_cache_ops.copy_blocks for ONE layer
 *   csrc/cache_kernels.cu:71-170, python vllm/_custom_ops.py:1661-1664.
 * block_mapping: device int64 [num_pairs, 2] (src, dst); block_bytes = bytes of
 * one cache block. */
int vllm_amd_copy_blocks(void* key_cache, void* value_cache,
                         const int64_t* block_mapping, int num_pairs,
                         int64_t block_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* fp8 (e4m3fn) quantisation, input [num_tokens, hidden] contiguous           */
/* ------------------------------------------------------------------------- */

/* Replaces torch.ops._C.static_scaled_fp8_quant
 *   schema csrc/torch_bindings.cpp:575-578, csrc/quantization/fp8/common.cu:14-26, 87-110. */
int vllm_amd_static_scaled_fp8_quant(void* out, const void* input,
                                     const float* scale, int64_t num_tokens,
                                     int hidden_size, int dtype, void* stream);
/* Replaces torch.ops._C.dynamic_scaled_fp8_quant
 *   schema csrc/torch_bindings.cpp:581-585, common.cu:112-139, common.cuh:64-97.
 * *scale must be <= 0 on entry (the python wrapper zeroes it, _custom_ops.py:1297). */
int vllm_amd_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale,
                                      int64_t num_tokens, int hidden_size,
                                      int dtype, void* stream);
/* Replaces torch.ops._C.dynamic_per_token_scaled_fp8_quant
 *   schema csrc/torch_bindings.cpp:588-593, common.cu:28-83, 141-171.
 * scales: float[num_tokens] out; scale_ub: float[1] or NULL. */
int vllm_amd_dynamic_per_token_scaled_fp8_quant(void* out, const void* input,
                                                float* scales,
                                                const float* scale_ub,
                                                int64_t num_tokens,
                                                int hidden_size, int dtype,
                                                void* stream);

/* ------------------------------------------------------------------------- */
/* w8a8 fp8 GEMMs                                                             */
/* ------------------------------------------------------------------------- */

/* Replaces torch.ops._C.cutlass_scaled_mm (fp8 x fp8 case) and the
 * torch._scaled_mm call of torch_per_token_w8a8_scaled_mm
 *   schema csrc/torch_bindings.cpp:422-427, python vllm/_custom_ops.py:697-742,
 *   vllm/model_executor/layers/quantization/utils/w8a8_utils.py:199-226.
 * out[M,N] = ((a_scales (.) a) . (b_scales (.) b)).to(out_dtype) (+ bias)
 * a: e4m3 [M,K] row-major (row stride lda); b: e4m3 [K,N] COLUMN-major, i.e.
 * element (k,n) at b[n*ldb + k] (the reference passes weight.t());
 * a_scales: float[1] or [M]; b_scales: float[1] or [N]; bias: out_dtype[N] or NULL;
 * out row stride ldc.  K % 16 == 0. */
int vllm_amd_scaled_mm(void* out, const void* a, const void* b,
                       const float* a_scales, const float* b_scales,
                       const void* bias, int M, int N, int K, int64_t lda,
                       int64_t ldb, int64_t ldc, int a_scale_numel,
                       int b_scale_numel, int out_dtype, void* stream);

/* No counterpart in the reference (its libraries own their workspaces): registers (ws != NULL)
 * or removes (ws == NULL) the CURRENT device's workspace -- `bytes` of caller-owned device
 * memory, 256-byte aligned, valid until replaced or removed; at least 513 KiB, sized for full use
 * at 512 KiB + 4 x 16 MiB.  Its ticket counters are zeroed on `stream` (asynchronously: order
 * the first op behind it).  Nothing is ever allocated inside an op call; the kernels that want
 * device scratch beyond their arguments take it from here and run their scratch-free form
 * without it:
 *   vllm_amd_scaled_mm           the few shapes its heuristic splits over K (M <= 64, K >= 12288)
 *   vllm_amd_paged_attention_v1  the opt-in balanced work distribution for skewed batches
 *                                (tuning "pa_balance" = 1; the op has no scratch arguments)
 * The workspace is cut into 4 slots; a slot belongs to the first stream that asks for it,
 * launches of one stream share their slot (they are ordered), a fifth stream, a launch under
 * stream capture and a launch that needs more than a slot use the scratch-free form. */
int vllm_amd_set_workspace(void* ws, int64_t bytes, void* stream);

/* No counterpart in the reference.  The ONE entry point for callers that cannot allocate or
 * register device memory (code under HIP-graph capture, tight serving loops, harnesses that
 * forbid hipMalloc): vllm_amd_scaled_mm with its split-K partials and tickets in a file-scope
 * static __device__ array of the library (8 MiB + 16 KiB per GPU the library runs on, held until
 * the process ends) instead of a registered workspace.  Same arguments and results as
 * vllm_amd_scaled_mm, plus `splits`: K slices per output tile (0 = the heuristic of
 * vllm_amd_scaled_mm, 1 = never split, n = n slices, at most the number of K blocks).  A split
 * that needs more than the array holds (tiles x slices x tile bytes > 8 MiB, or > 4096 tiles)
 * returns VLLM_AMD_EUNSUPPORTED with a message; nothing is allocated, freed or synchronised (the FIRST
 * call on a device enqueues one hipMemsetAsync of the ticket words on `stream`; every split launch
 * leaves its tickets at zero, so later calls enqueue nothing but the kernel).  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (different streams, or a replaying graph beside
 * eager calls) race -- serialise them. */
int vllm_amd_scaled_mm_static(void* out, const void* a, const void* b,
                              const float* a_scales, const float* b_scales,
                              const void* bias, int M, int N, int K, int64_t lda,
                              int64_t ldb, int64_t ldc, int a_scale_numel,
                              int b_scale_numel, int out_dtype, int splits, void* stream);

/* No counterpart in the reference: a measurement helper for bench.py.  Runs a loop of nothing but the fp8 MFMA
 * the GEMM kernels use (v_mfma_scale_f32_16x16x128_f8f6f4, pseudo-random operands held in registers, 8 waves
 * on each of 256 workgroups) so that the GEMM's TFLOP/s can be read against what the matrix pipes deliver
 * under the chip's power management, not only against the 5 PF paper peak.  sink: 256 * 512 floats of device
 * memory; FLOP per call = 256 * 8 * iters * 32 * 2 * 16 * 16 * 128. */
int vllm_amd_mfma_ceiling(void* sink, int iters, void* stream);

/* Replaces torch.ops._rocm_C.wvSplitKQ
 *   schema csrc/rocm/torch_bindings.cpp:34-37, csrc/rocm/skinny_gemms.cu:1649-1714,
 *   python vllm/_custom_ops.py:1501-1508.
 * out_c[N_in, M_in] = (in_b[N_in,K] . in_a[M_in,K]^T) * scale_a * scale_b,
 * per-tensor scales (float[1] each), N_in <= 4, K % 16 == 0. */
int vllm_amd_wvSplitKQ(const void* in_a, const void* in_b, void* out_c,
                       const float* scale_a, const float* scale_b, int M_in,
                       int K_in, int N_in, int out_dtype, int cu_count,
                       void* stream);

/* ------------------------------------------------------------------------- */
/* Neighbours of the path: the ops that produce the GEMM's fp8   */
/* activations and the cache write's slot mapping.                              */
/* ------------------------------------------------------------------------- */

/* Replaces torch.ops._C.rms_norm_static_fp8_quant
 * (csrc/layernorm_quant_kernels.cu:25-58, 172-198; schema torch_bindings.cpp:190-195;
 * wrapper vllm/_custom_ops.py:283-286).
 * out[t, i] = fp8(clamp(scalar_t(x * rsqrt(mean(x^2) + eps)) * weight[i] * (1 / *scale))).
 * input rows are input_stride elements apart; out is contiguous [num_tokens, hidden]. */
int vllm_amd_rms_norm_static_fp8_quant(void* out, const void* input,
                                       const void* weight, const float* scale,
                                       float epsilon, int64_t num_tokens,
                                       int hidden_size, int64_t input_stride,
                                       int dtype, void* stream);

/* Replaces torch.ops._C.fused_add_rms_norm_static_fp8_quant
 * (csrc/layernorm_quant_kernels.cu:64-168, 214-253; wrapper _custom_ops.py:289-293).
 * residual <- scalar_t(input + residual) (in place, contiguous), then as above on it. */
int vllm_amd_fused_add_rms_norm_static_fp8_quant(void* out, const void* input,
                                                 void* residual, const void* weight,
                                                 const float* scale, float epsilon,
                                                 int64_t num_tokens, int hidden_size,
                                                 int64_t input_stride, int dtype,
                                                 void* stream);

/* Replaces torch.ops._C.rms_norm_dynamic_per_token_quant for fp8 output
 * (csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu:11-155,
 * layernorm_utils.cuh:20-119; wrapper _custom_ops.py:368-384).  scale_ub and residual
 * may be NULL; scales is float[num_tokens] (out); residual is updated in place with
 * scalar_t(input + residual).  The _int8 form is the same op with an int8 output
 * (fused_layernorm_dynamic_per_token_quant.cu:33-36, quant_conversions.cuh:15-31, 51-61:
 * scale = max(amax / 127, FLT_EPSILON), q = saturate(nearbyint(y * (1 / scale))); no scale_ub):
 * functional coverage on the strided kernel, outside the fp8 path. */
int vllm_amd_rms_norm_dynamic_per_token_quant_int8(void* out, const void* input, const void* weight,
                                                   float* scales, float epsilon, void* residual,
                                                   int64_t num_tokens, int hidden_size, int dtype,
                                                   void* stream);
int vllm_amd_rms_norm_dynamic_per_token_quant(void* out, const void* input,
                                              const void* weight, float* scales,
                                              float epsilon, const float* scale_ub,
                                              void* residual, int64_t num_tokens,
                                              int hidden_size, int dtype, void* stream);

/* Replaces torch.ops._C.silu_and_mul_quant (csrc/quantization/activation_kernels.cu:14-121).
 * input [num_tokens, 2 d] (f16 / bf16): out[t, i] = fp8(clamp(T(silu(x)) * y * (1 / *scale))). */
int vllm_amd_silu_and_mul_quant(void* out, const void* input, const float* scale,
                                int64_t num_tokens, int d, int dtype, void* stream);

/* Device-side BlockTable.compute_slot_mapping (vllm/v1/worker/block_table.py:84-99):
 * slot_mapping[i] = block_table[req_indices[i], positions[i] / block_size] * block_size
 *                   + positions[i] % block_size.  The reference computes this with numpy
 * on the host and copies it to the GPU every step (commit_slot_mapping, :105-107). */
int vllm_amd_compute_slot_mapping(int64_t* slot_mapping, const int32_t* block_table,
                                  const int32_t* req_indices, const int64_t* positions,
                                  int64_t num_tokens, int max_num_blocks_per_req,
                                  int block_size, void* stream);

/* Replaces torch.ops._C_cache_ops.swap_blocks (csrc/cache_kernels.cu:25-67):
 * block_mapping_host is a HOST array of num_pairs (src, dst) int64 pairs; src / dst may
 * each be device or host memory (block_bytes per block number). */
int vllm_amd_swap_blocks(const void* src, void* dst, int64_t block_bytes,
                         const int64_t* block_mapping_host, int64_t num_pairs,
                         void* stream);

/* Replaces context_attention_fwd (vllm/attention/ops/prefix_prefill.py:726-913; Triton
 * kernels _fwd_kernel :36-288, _fwd_kernel_alibi :477-722): prefill / chunked-prefill
 * attention.  For sequence b the query tokens q[b_start_loc[b] : b_start_loc[b+1]] (positions
 * ctx .. seq_len-1, ctx = b_seq_len[b] - query_len) attend to the ctx cached tokens in the
 * paged cache (through b_loc[b, :], no causal mask) and to the new tokens' dense k / v
 * (causal), one softmax over both; out = acc / l.
 *   q, o    [num_tokens, num_heads, head_size]; k, v [num_tokens, num_kv_heads, head_size]
 *           (element strides *_stride_t per token, *_stride_h per head; head dim contiguous)
 *   caches  as paged_attention_v1 (element strides of dim 0 and 1)
 *   b_loc   int32 [batch, *] block table (row stride b_loc_stride), b_start_loc int32
 *           [batch+1], b_seq_len int32 [batch]
 *   k_scale, v_scale  float[1], fp8 cache only: k = (fp8 -> f32 * k_scale).to(q dtype)
 *   alibi_slopes      float[num_heads] or NULL (with ALiBi the sliding window is ignored,
 *                     as in the reference's alibi kernel)
 *   fp8_out_scale     float[1] or NULL: o is e4m3 bytes = clamp(acc / l / scale)
 *   sliding_window    <= 0 disables; skip_decode: sequences with one query token are skipped
 *   num_tokens        rows of q (total new tokens), or <= 0 if unknown: when given, the grid has
 *                     one entry per row block of any sequence instead of max_input_len-many
 *                     blocks for every sequence (mixed batches: few long chunks among many
 *                     one-token sequences launch no empty workgroups)
 * Supported here: f16 / bf16 queries, head_size a multiple of 8 (16 with an fp8 cache) up to
 * 256 (padded to a multiple of 32, above 128 to 256, inside the kernel), block_size 16/32,
 * kv_cache_dtype auto / fp8_e4m3.  float32 queries (the reference test's second grid,
 * tests/kernels/attention/test_prefix_prefill.py:513-560; cache float32 with x = 4, or fp8):
 * head_size a multiple of 4 (16 with an fp8 cache) up to 256, any block_size, num_tokens
 * required; computed in fp32 on a plain kernel. */
int vllm_amd_context_attention_fwd(
    void* o, const void* q, const void* k, const void* v, const void* k_cache,
    const void* v_cache, const int* b_loc, const int* b_start_loc,
    const int* b_seq_len, const float* k_scale, const float* v_scale,
    const float* alibi_slopes, const float* fp8_out_scale, int batch, int num_heads,
    int num_kv_heads, int head_size, int block_size, int max_input_len,
    int64_t b_loc_stride, int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t,
    int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h, int64_t o_stride_t,
    int64_t o_stride_h, int64_t kc_block_stride, int64_t kc_head_stride,
    int64_t vc_block_stride, int64_t vc_head_stride, float sm_scale, int sliding_window,
    int skip_decode, int num_tokens, int dtype, int kv_cache_dtype, void* stream);

/* Replaces unified_attention (vllm/attention/ops/triton_unified_attention.py:588-700; Triton
 * kernels kernel_unified_attention_2d :52-272 / _3d :276-586): the V1 engine's attention over
 * the FLASH-layout cache.  For sequence b the query tokens q[cu_seqlens_q[b] :
 * cu_seqlens_q[b+1]] sit at positions ctx .. seqused_k[b]-1 (ctx = seqused_k[b] - query_len);
 * query m attends to the cached keys n <= ctx + m (causal; the new tokens are already in the
 * cache), optionally inside a sliding window, with logit soft-capping and ALiBi:
 *   S = softmax_scale * q.k;  S = softcap * tanh(S / softcap) if softcap > 0;
 *   masked (-inf) where n > ctx + m or ctx + m - n >= sliding_window (if > 0);
 *   S += alibi_slope[head] * (n - ctx);  out = softmax(S) . v  (:219-262)
 *   q, out   [num_tokens, num_heads, head_size] (element strides per token / head)
 *   k_cache, v_cache  [num_blocks, block_size, num_kv_heads, head_size] through element
 *            strides of dims 0, 1, 2 (head dim contiguous), dtype of q or e4m3 bytes
 *   block_table int32 [num_seqs, *] (row stride), cu_seqlens_q int32 [num_seqs+1],
 *            seqused_k int32 [num_seqs]
 *   k_descale, v_descale  float, element 0 is used (the kernel's tl.load(k_scale), :177, :191);
 *            fp8 cache only: k = (fp8 -> f32 * k_descale).to(q dtype)
 *   output_scale  float[1] or NULL: out is e4m3 bytes = clamp(acc / l / scale) (:254-256)
 *   sliding_window = 1 + window_size[0] of the reference call (0 = none) (:655)
 * Supported here: f16 / bf16 queries (no fp8 queries: q_descale is "not supported" in the
 * reference either, :609), head_size a multiple of 8 (16 with an fp8 cache) up to 256,
 * block_size any power of two from 16.  With GQA the tile rows are (token, q head of the group) pairs, so a
 * decode token streams its K / V once per group.  Sequences of query length 1 take a dedicated
 * bandwidth kernel (unified_decode.hip) when the whole launch
 * is decode-only (max_seqlen_q == 1) or has >= 16 sequences; the grid of the tile kernel has
 * one entry per row block of any sequence when num_tokens is given (> 0).
 *   num_tokens = rows of q (total query tokens of the launch); max_seqlen_k >= every seqused_k
 *   segm_output float [num_tokens, num_heads, num_segments, head_size], segm_max / segm_expsum
 *            float [num_tokens, num_heads, num_segments]: caller-allocated scratch as in the
 *            reference's 3-D path (:659-683; NUM_SEGMENTS = 16 there), or NULL / num_segments
 *            <= 1.  When given and every sequence fits one row block (decode), the keys of a
 *            sequence are split over up to min(num_segments, 16) workgroups and merged by a
 *            second kernel (reduce_segments, :499-586); contents afterwards are unspecified. */
int vllm_amd_unified_attention(
    void* out, const void* q, const void* k_cache, const void* v_cache,
    const int* cu_seqlens_q, const int* seqused_k, const int* block_table,
    const float* k_descale, const float* v_descale, const float* alibi_slopes,
    const float* output_scale, int num_seqs, int num_heads, int num_kv_heads, int head_size,
    int block_size, int max_seqlen_q, int64_t block_table_stride, int64_t q_stride_t,
    int64_t q_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t kc_stride_block,
    int64_t kc_stride_slot, int64_t kc_stride_head, int64_t vc_stride_block,
    int64_t vc_stride_slot, int64_t vc_stride_head, float softmax_scale, int sliding_window,
    float softcap, int num_tokens, int max_seqlen_k, float* segm_output, float* segm_max,
    float* segm_expsum, int num_segments, int dtype, int kv_cache_dtype, void* stream);

/* ------------------------------------------------------------------------- */
/* Peer-to-peer all-reduce over xGMI                         */
/* ------------------------------------------------------------------------- */
/* Replaces torch.ops._C_custom_ar.{meta_size, init_custom_ar, register_buffer, all_reduce,
 * dispose, allocate_shared_buffer_and_handle, open_mem_handle, free_shared_buffer}
 *   schemas csrc/torch_bindings.cpp:731-754, csrc/custom_all_reduce.cu:13-189,
 *   kernels csrc/custom_all_reduce.cuh:296-370, one / two stage choice :561-571,
 *   python vllm/distributed/device_communicators/custom_all_reduce.py:49-305.
 * meta_ptrs[r] / ptrs[r]: rank r's signal block (vllm_amd_ar_meta_size() bytes, zeroed, followed
 * by scratch_bytes of scratch) / rank r's copy of a registered buffer, as mapped into THIS
 * process.  rank_data: caller-owned device memory for the tuples of input pointers the kernels read
 * (64 bytes per registered or graph-captured input; the reference passes an 8 MiB tensor,
 * custom_all_reduce.py:168-176).  all_reduce: out = sum over ranks of their `inp` (f32 / f16 / bf16, fp32
 * accumulation in rank order: bit-identical on all ranks), numel * sizeof(dtype) a multiple of 16;
 * reg_buffer != NULL: inp is first copied into that registered buffer (eager mode), else inp itself
 * must be registered -- or the stream is capturing: then inp is recorded, the captured kernel points at a
 * tuple that vllm_amd_ar_register_graph_buffers fills after the capture and before the first replay
 * (custom_all_reduce.cu:125-146: get_graph_buffer_ipc_meta on every rank, all-gather, register).
 * Messages that take the two-stage kernel need their longest slice to fit scratch_bytes (error otherwise).
 * Every wait on a peer is bounded in wall-clock time (30 s; vllm_amd_ar_set_timeout_ms): a dead peer never
 * hangs the GPU, and a time-out is loud -- the output of that launch is NaN where the block gave up, and
 * every later all_reduce of the handle returns an error (a sticky word in pinned host memory: no
 * synchronisation; vllm_amd_ar_error reads the same word).
 * Logic-tested with simulated ranks on one GPU only; off by default. */
int64_t vllm_amd_ar_meta_size(void);
int vllm_amd_ar_init(void** handle, void* const* meta_ptrs, int64_t scratch_bytes, void* rank_data,
                     int64_t rank_data_bytes, int world_size, int rank, int fully_connected);
int vllm_amd_ar_set_timeout_ms(void* handle, int64_t ms);
int vllm_amd_ar_register_buffer(void* handle, void* const* ptrs);
int vllm_amd_ar_all_reduce(void* handle, const void* inp, void* out, int64_t numel, int dtype,
                           void* reg_buffer, int64_t reg_buffer_bytes, void* stream);
/* handles_out: capacity x 64 bytes (hipIpcMemHandle_t of each recorded input's allocation), offsets_out:
 * the inputs' offsets into them; *count = number recorded since the last registration (both outputs NULL:
 * only the count is returned). */
int vllm_amd_ar_get_graph_buffer_ipc_meta(void* handle, void* handles_out, int64_t* offsets_out,
                                          int capacity, int* count);
/* handles: [world][n][64] bytes, offsets: [world][n] -- every rank's answer to the call above. */
int vllm_amd_ar_register_graph_buffers(void* handle, const void* handles, const int64_t* offsets, int n);
int vllm_amd_ar_error(void* handle, int* timed_out);
int vllm_amd_ar_dispose(void* handle);
/* the only calls of the library that allocate, map or free memory (uncached device memory
 * + its 64-byte hipIpcMemHandle_t) */
int vllm_amd_ar_alloc_shared(int64_t bytes, void** ptr, void* ipc_handle);
int vllm_amd_ar_open_handle(const void* ipc_handle, void** ptr);
int vllm_amd_ar_close_handle(void* ptr);
int vllm_amd_ar_free_shared(void* ptr);


/* ------------------------------------------------------------------------- */
/* Tuning knobs (no counterpart in the reference): used by tools/ and bench   */
/* scripts to A/B kernel variants inside one process.                          */
/*   "pa_variant": bit 0 = non-temporal K/V loads, bit 1 = two steps of K/V    */
/*                 registers in flight (not instantiated, measured slower)     */
/*   "pa_sp":      reference partitions merged per workgroup in v2 / rocm      */
/*                 launches (0 = heuristic, paged_attention_mfma.hip)          */
/*   "pa_group":   XCD-grouped workgroup order of the decode kernel (0 = heuristic:   */
/*                 >= 1024 workgroups and >= 2 kv heads, 1 = always, 2 = never)     */
/*   "pa_balance": 1 = paged_attention_v1 cuts the concatenated sequences into equal    */
/*                 ranges, two workgroups per CU, and merges shared sequences through   */
/*                 the registered workspace (pays on skewed batches only; default 0)    */
/*   "pa_order":   0 = unpartitioned decode launches keep the hardware's workgroup     */
/*                 order (default 1: sequences handed out longest first, the two       */
/*                 workgroups of a CU paired long + short; same output bits)          */
/*   "mm_kernel":  scaled_mm kernel for M > 32: 0 = heuristic, 1 = register-   */
/*                 staged 128x128 tile, 2..10 = direct-to-LDS tile shapes       */
/*                 (256x256, 128x128, 128x64, 64x64, 64x32; even ids 4..10      */
/*                 take 2 or 4 K blocks per barrier; scaled_mm_large.hip),      */
/*                 11 = weight-streaming kernel (M <= 32 only),                 */
/*                 12, 13 = 32x128 tile (1 / 2 K blocks per barrier)            */
/*                 14 = 256x256 tile on 4 waves (accumulators in AGPRs),        */
/*                 15 = ring of five half stages, 16 = waves 4-7 stage the    */
/*                 whole tile (14-16: 256x256 variants, measured, not           */
/*                 used by the heuristic: profiles/r02_gemm_ablation.txt),      */
/*                 17 = 256x256 on 16 waves; 18..25 = scaled_mm_pp.hip (value - 18:   */
/*                 bit 0 stagger, bit 1 persistent, bit 2 stagger by a quarter     */
/*                 block; 21 = what the heuristic ships for MFMA-bound shapes),     */
/*                 26..28 = 64x64 / 128x64 tiles with deeper rings (no gain),       */
/*                 29..36 = 18..25 with 256 x 192 tiles (the heuristic takes them  */
/*                 where they save a round of workgroups: N 6144)                  */
/*   "mm_patch":   tile columns per super-column of the scaled_mm_pp tile order    */
/*                 (default 8; <= 1 = column-major)                                  */
/*   "mm_store":   scaled_mm_pp measurement: 3 = no output stores (garbage)         */
/*   "mm_prio":    scaled_mm_pp measurement: 1 / 2 = s_setprio 1 for waves 4-7 / 0-3 */
/*   "pf_rows":    query rows per workgroup of the prefill kernels (0 = default,    */
/*                 64 / 128); 1 = unified_attention without GQA packing          */
/*   "uni_decode": 0 = unified_attention never uses its decode kernel (1 =     */
/*                 default; 2 / 3 = XCD-grouped workgroup order always / never) */
/*   "uni_hp":     adjacent kv heads of a token-major cache taken as one row of  */
/*                 that decode kernel (0 = heuristic, 1 = never, 2 / 4 = that    */
/*                 many where the shape allows)                                  */
/*   "pf_fa":      the 256-row kernel of unified_attention /                   */
/*                 context_attention_fwd (prefill_fa.hip; head sizes 72 .. 256, */
/*                 causal mask, unified: + sliding window, softcap): 0 = for launches whose longest query chunk fills */
/*                 a row block (default), 1 = never, 2 = whenever it applies   */
/*   "pf_fa_stagger": 0 = all eight waves of that kernel stage in step (1 =    */
/*                 default: waves 4-7 at the top of the tile interval)         */
/*   "cache_runs": 0 = reshape_and_cache always per token (no run detection)    */
/*   "mm_splitk":  K slices per output tile of the direct-to-LDS kernels        */
/*                 (0 = heuristic, 1 = never split, n = n slices; reduced       */
/*                 inside the kernel through the registered workspace,          */
/*                 vllm_amd_set_workspace; unsplit without one)                 */
/* ------------------------------------------------------------------------- */
int vllm_amd_set_tuning(const char* key, int value);
/* current value of "pa_balance" / "mm_kernel" / "mm_splitk" (the torch layer asks before it allocates) */
int vllm_amd_get_tuning(const char* key, int* value);

#ifdef __cplusplus
}
#endif
#endif /* VLLM_AMD_H_ */
