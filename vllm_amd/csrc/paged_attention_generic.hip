// Paged attention, decode: generic VALU path + split-K reduce + C ABI entry points.
//
// The generic path covers every head size / block size / dtype / block-sparse
// combination of the reference operators (csrc/attention/paged_attention_v1.cu
// :94-182) and is the fallback of the MFMA bandwidth kernel in
// paged_attention_mfma.hip, which handles the shapes BASELINE.json measures.
//
// Generic kernel layout (one workgroup of 256 threads per (q head, seq, partition)):
//   phase 1  QK   one thread per token, 16-byte K-cache vectors (a token's x
//                 contiguous elements of one d-chunk), q broadcast from LDS;
//   phase 2  softmax over the partition, fp32, block max / sum;
//   phase 3  PV   one thread per head element d, 16-byte V-cache vectors (the
//                 block's tokens of one d), groups of threads stride the blocks;
//   phase 4  cross-group reduction through LDS, write out / tmp_out.
// Numerics: fp32 logits and accumulation, 1/(sum+1e-6) normalisation and
// probabilities rounded to the query dtype before PV as in the reference
// (attention_kernels.cuh:337, :393-395); fp8 K/V are used exactly and
// k_scale / v_scale are folded into the logit scale / the accumulator (the
// ROCm reference's convention, csrc/rocm/attention.cu:514-518, 714-717).
#include "paged_attention.cuh"

#include <float.h>
#include <string.h>

namespace va {

extern int g_mm_kernel;
extern int g_mm_splitk;
extern int g_mm_tail;
extern int g_mm_patch;
extern int g_mm_store;
extern int g_mm_prio;
extern int g_cache_runs;
extern int g_cache_hpw;
extern int g_uni_hp;
extern int g_pf_rows;   // prefill_attention.hip
extern int g_uni_decode;
extern int g_pf_fa;
extern int g_pf_fa_stagger;
extern int g_pf_fa_abl;

template <typename cache_t> struct CacheTraits;
template <> struct CacheTraits<float> { static constexpr int X = 4; };
template <> struct CacheTraits<f16_t> { static constexpr int X = 8; };
template <> struct CacheTraits<bf16_t> { static constexpr int X = 8; };
template <> struct CacheTraits<uint8_t> { static constexpr int X = 16; };

template <typename cache_t>
__device__ __forceinline__ float cache_to_f32(cache_t v) { return to_f32(v); }
template <>
__device__ __forceinline__ float cache_to_f32<uint8_t>(uint8_t v) { return fp8_to_f32(v); }

template <typename cache_t, int X> struct alignas(16) CVec { cache_t v[X]; };

__device__ __forceinline__ float block_reduce_max(float v, float* red, int nw) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red, int nw) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}

constexpr int kGenThreads = 256;

template <typename scalar_t, typename cache_t, bool SPARSE>
__global__ __launch_bounds__(kGenThreads) void pa_generic_kernel(
    scalar_t* __restrict__ out,       // v1: [S,H,D]   v2: tmp_out [S,H,P,D]
    float* __restrict__ exp_sums, float* __restrict__ max_logits,
    const scalar_t* __restrict__ q, const cache_t* __restrict__ k_cache,
    const cache_t* __restrict__ v_cache, int num_kv_heads, float scale,
    const int* __restrict__ block_tables, const int* __restrict__ seq_lens,
    const int* __restrict__ query_start_loc, int max_num_blocks_per_seq,
    const float* __restrict__ alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, const float* __restrict__ k_scale_p,
    const float* __restrict__ v_scale_p, int head_size, int block_size, int partition_size,
    int tp_rank, int bs_local_blocks, int bs_vert_stride, int bs_block_size,
    int bs_head_sliding_step, int window) {
  constexpr int X = CacheTraits<cache_t>::X;
  constexpr bool FP8 = sizeof(cache_t) == 1;
  const int head = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
  const int num_heads = gridDim.x, max_parts = gridDim.z;
  const int tid = threadIdx.x;
  constexpr int NW = kGenThreads / 64;

  int64_t qrow = seq;
  if (query_start_loc) {
    if (query_start_loc[seq + 1] - query_start_loc[seq] != 1) return;
    qrow = query_start_loc[seq];
  }
  const int seq_len = seq_lens[seq];
  const bool partitioned = partition_size > 0;
  if (partitioned && part * partition_size >= seq_len) return;
  if (!partitioned && seq_len <= 0) {
    for (int d = tid; d < head_size; d += kGenThreads)
      out[((int64_t)seq * num_heads + head) * head_size + d] = from_f32<scalar_t>(0.f);
    return;
  }

  const int num_seq_blocks = (seq_len + block_size - 1) / block_size;
  const int nbpp = partitioned ? partition_size / block_size : num_seq_blocks;
  // sliding window: tokens in front of first_tok are masked; partitions entirely in front of it
  // write nothing (the reduce kernel starts at the window's first partition)
  const int first_tok = window > 0 ? max(0, seq_len - window) : 0;
  if (partitioned && min((part + 1) * partition_size, seq_len) <= first_tok) return;
  const int end_block = min((partitioned ? part * nbpp : 0) + nbpp, num_seq_blocks);
  const int start_block = max(partitioned ? part * nbpp : 0, first_tok / block_size);
  const int nblocks = end_block - start_block;
  const int start_tok = start_block * block_size;
  const int end_tok = min(start_tok + nblocks * block_size, seq_len);
  const int num_tokens = end_tok - start_tok;
  const int padded_tokens = nblocks * block_size;

  const int qpk = num_heads / num_kv_heads;
  const int kvh = head / qpk;
  const float slope = alibi_slopes ? alibi_slopes[head] : 0.f;
  const int* bt = block_tables + (int64_t)seq * max_num_blocks_per_seq;
  float kscale = 1.f, vscale = 1.f;
  if constexpr (FP8) { kscale = *k_scale_p; vscale = *v_scale_p; }
  const float qk_scale = scale * kscale;

  int bs_off = 0, q_bs_id = 0;
  if constexpr (SPARSE) {
    q_bs_id = (seq_len - 1) / bs_block_size;
    if (bs_head_sliding_step >= 0)
      bs_off = (tp_rank * num_heads + head) * bs_head_sliding_step + 1;
    else
      bs_off = (tp_rank * num_kv_heads + kvh) * (-bs_head_sliding_step) + 1;
  }
  auto block_attended = [&](int b) -> bool {
    if constexpr (!SPARSE) return true;
    const int id = b * block_size / bs_block_size;
    const bool remote = ((id + bs_off) % bs_vert_stride) == 0;
    const bool local = id > q_bs_id - bs_local_blocks;
    return remote || local;
  };

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* q_s = reinterpret_cast<float*>(smem_raw);                     // [head_size]
  float* red = q_s + ((head_size + 3) & ~3);                            // [16]
  float* logits = red + 16;                                             // [padded_tokens]
  // phase-4 scratch aliases logits (needs groups*head_size floats)

  const scalar_t* qp = q + qrow * q_stride + (int64_t)head * head_size;
  for (int d = tid; d < head_size; d += kGenThreads) q_s[d] = to_f32(qp[d]);
  __syncthreads();

  // ---- phase 1: logits -----------------------------------------------------
  const int nchunks = head_size / X;
  float qk_max = -FLT_MAX;
  for (int i = tid; i < padded_tokens; i += kGenThreads) {
    const int b = start_block + i / block_size;
    const int t = i % block_size;
    const int tok = start_tok + i;
    float lg;
    if (!block_attended(b)) {
      lg = -FLT_MAX;  // attention_kernels.cuh:229-251
    } else {
      const int64_t pb = bt[b];
      const cache_t* kp = k_cache + pb * kv_block_stride + (int64_t)kvh * kv_head_stride +
                          (int64_t)t * X;
      float acc = 0.f;
      for (int c = 0; c < nchunks; ++c) {
        CVec<cache_t, X> kv =
            *reinterpret_cast<const CVec<cache_t, X>*>(kp + (int64_t)c * block_size * X);
#pragma unroll
        for (int j = 0; j < X; ++j) acc = fmaf(q_s[c * X + j], cache_to_f32(kv.v[j]), acc);
      }
      lg = acc * qk_scale;
      if (slope != 0.f) lg += slope * (float)(tok - seq_len + 1);
      if (tok >= seq_len) lg = 0.f;          // masked tail, :296-298
      else if (tok < first_tok) lg = -FLT_MAX;
      else qk_max = fmaxf(qk_max, lg);
    }
    logits[i] = lg;
  }
  qk_max = block_reduce_max(qk_max, red, NW);

  // ---- phase 2: softmax -------------------------------------------------------
  float exp_sum = 0.f;
  for (int i = tid; i < num_tokens; i += kGenThreads) {
    float v = __expf(logits[i] - qk_max);
    logits[i] = v;
    exp_sum += v;
  }
  exp_sum = block_reduce_sum(exp_sum, red, NW);
  const float inv_sum = __fdividef(1.f, exp_sum + 1e-6f);
  for (int i = tid; i < padded_tokens; i += kGenThreads) {
    float p = i < num_tokens ? logits[i] * inv_sum : 0.f;
    logits[i] = to_f32(from_f32<scalar_t>(p));  // probabilities in the query dtype
  }
  __syncthreads();
  if (partitioned && tid == 0) {
    const int64_t sidx = ((int64_t)seq * num_heads + head) * max_parts + part;
    max_logits[sidx] = qk_max;
    exp_sums[sidx] = exp_sum;
  }

  // ---- phase 3: PV ----------------------------------------------------------------
  const int groups = max(1, kGenThreads / head_size);
  const int g = tid / head_size;
  const int d = tid % head_size;
  const bool active = tid < groups * head_size;
  // rows longer than 256 (never: max head 256) are looped
  float acc = 0.f;
  float acc2 = 0.f;  // second row for head_size > 256 is not needed (max 256)
  (void)acc2;
  if (active) {
    for (int bi = g; bi < nblocks; bi += groups) {
      const int b = start_block + bi;
      if (!block_attended(b)) continue;
      const int64_t pb = bt[b];
      const cache_t* vp = v_cache + pb * kv_block_stride + (int64_t)kvh * kv_head_stride +
                          (int64_t)d * block_size;
      const float* pr = logits + bi * block_size;
      const int valid = min(block_size, seq_len - b * block_size);
      if (block_size % X == 0) {
        for (int t0 = 0; t0 < block_size; t0 += X) {
          CVec<cache_t, X> vv = *reinterpret_cast<const CVec<cache_t, X>*>(vp + t0);
#pragma unroll
          for (int j = 0; j < X; ++j)
            if (t0 + j < valid) acc = fmaf(pr[t0 + j], cache_to_f32(vv.v[j]), acc);
        }
      } else {
        for (int t = 0; t < valid; ++t) acc = fmaf(pr[t], cache_to_f32(vp[t]), acc);
      }
    }
  }
  __syncthreads();  // logits no longer needed: reuse as cross-group scratch
  float* scratch = logits;
  if (active && g > 0) scratch[(g - 1) * head_size + d] = acc;
  __syncthreads();
  if (tid < head_size) {
    for (int gg = 1; gg < groups; ++gg) acc += scratch[(gg - 1) * head_size + d];
    acc *= vscale;
    int64_t oidx;
    if (partitioned)
      oidx = (((int64_t)seq * num_heads + head) * max_parts + part) * head_size + d;
    else
      oidx = ((int64_t)qrow * num_heads + head) * head_size + d;
    out[oidx] = from_f32<scalar_t>(acc);
  }
}

// ---------------------------------------------------------------------------
// Split-K reduce (v2: attention_kernels.cuh:562-664; rocm: rocm/attention.cu:1271-1475)
// grid (num_heads, num_seqs), 128 threads.
// ---------------------------------------------------------------------------
template <typename scalar_t, bool ROCM>
__global__ __launch_bounds__(128) void pa_reduce_kernel(
    void* __restrict__ out_, const float* __restrict__ exp_sums,
    const float* __restrict__ max_logits, const scalar_t* __restrict__ tmp_out,
    const int* __restrict__ seq_lens, const int* __restrict__ query_start_loc,
    const float* __restrict__ fp8_out_scale, int head_size, int partition_size,
    int max_num_partitions, int slot_stride, int window) {
  const int head = blockIdx.x, seq = blockIdx.y, num_heads = gridDim.x;
  const int tid = threadIdx.x;
  int64_t orow = seq;
  if (query_start_loc) {
    if (query_start_loc[seq + 1] - query_start_loc[seq] != 1) return;
    orow = query_start_loc[seq];
  }
  const int seq_len = seq_lens[seq];
  // partition_size = tokens per USED scratch slot; used slots are slot_stride apart
  // sliding window: the used slots in front of the window's first one were never written
  const int j0 = window > 0 ? max(0, seq_len - window) / partition_size : 0;
  const int np = (seq_len + partition_size - 1) / partition_size - j0;
  const int64_t sidx = ((int64_t)seq * num_heads + head) * max_num_partitions + (int64_t)j0 * slot_stride;
  const scalar_t* tp = tmp_out + sidx * head_size;
  scalar_t* out = (scalar_t*)out_ + (orow * num_heads + head) * head_size;
  uint8_t* out8 = (uint8_t*)out_ + (orow * num_heads + head) * head_size;
  if (np <= 0) {
    for (int i = tid; i < head_size; i += 128) {
      if (ROCM && fp8_out_scale) out8[i] = 0; else out[i] = from_f32<scalar_t>(0.f);
    }
    return;
  }
  if (!ROCM && np == 1) {  // :577-589
    for (int i = tid; i < head_size; i += 128) out[i] = tp[i];
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* w = reinterpret_cast<float*>(smem_raw);  // [np]
  __shared__ float red[4];
  float m = -FLT_MAX;
  for (int j = tid; j < np; j += 128) {
    float l = max_logits[sidx + (int64_t)j * slot_stride];
    w[j] = l;
    m = fmaxf(m, l);
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);
  float g = 0.f;
  for (int j = tid; j < np; j += 128) {
    float r = exp_sums[sidx + (int64_t)j * slot_stride] * expf(w[j] - m);
    w[j] = r;
    g += r;
  }
  g = wave_sum(g);
  __syncthreads();
  if ((tid & 63) == 0) red[2 + (tid >> 6)] = g;
  __syncthreads();
  g = red[2] + red[3];
  const float inv = __fdividef(1.0f, g + 1e-6f);
  for (int i = tid; i < head_size; i += 128) {
    float acc = 0.f;
    if constexpr (!ROCM) {
      for (int j = 0; j < np; ++j) acc += to_f32(tp[(int64_t)j * slot_stride * head_size + i]) * w[j] * inv;
      out[i] = from_f32<scalar_t>(acc);
    } else {
      for (int j = 0; j < np; ++j) acc += to_f32(tp[(int64_t)j * slot_stride * head_size + i]) * w[j];
      acc *= inv;
      if (fp8_out_scale) {
        acc *= 1.0f / (*fp8_out_scale);
        out8[i] = f32_to_fp8(acc);
      } else {
        out[i] = from_f32<scalar_t>(acc);
      }
    }
  }
}

template <typename scalar_t>
static int launch_reduce(const PagedAttnParams& p) {
  dim3 grid(p.num_heads, p.num_seqs);
  size_t smem = sizeof(float) * (size_t)p.max_num_partitions;
  if (p.rocm_semantics)
    hipLaunchKernelGGL((pa_reduce_kernel<scalar_t, true>), grid, dim3(128), smem, p.stream, p.out,
                       p.exp_sums, p.max_logits, (const scalar_t*)p.tmp_out, p.seq_lens,
                       p.query_start_loc, p.fp8_out_scale, p.head_size, p.partition_size * p.sp_parts,
                       p.max_num_partitions, p.sp_parts, p.window);
  else
    hipLaunchKernelGGL((pa_reduce_kernel<scalar_t, false>), grid, dim3(128), smem, p.stream, p.out,
                       p.exp_sums, p.max_logits, (const scalar_t*)p.tmp_out, p.seq_lens,
                       p.query_start_loc, p.fp8_out_scale, p.head_size, p.partition_size * p.sp_parts,
                       p.max_num_partitions, p.sp_parts, p.window);
  VA_LAUNCH_CHECK("paged_attention reduce launch");
  return VLLM_AMD_OK;
}

int paged_attention_reduce(const PagedAttnParams& p) {
  if (p.dtype == VLLM_AMD_F32) return launch_reduce<float>(p);
  if (p.dtype == VLLM_AMD_F16) return launch_reduce<f16_t>(p);
  return launch_reduce<bf16_t>(p);
}

template <typename scalar_t, typename cache_t>
static int launch_generic(const PagedAttnParams& p) {
  const bool part = p.partition_size > 0;
  const int nparts = part ? p.max_num_partitions : 1;
  const int max_tokens = part ? p.partition_size
                              : p.max_num_blocks_per_seq * p.block_size;
  const int padded = ceil_div(max_tokens, p.block_size) * p.block_size;
  const int groups = kGenThreads / p.head_size > 1 ? kGenThreads / p.head_size : 1;
  size_t tail = (size_t)(padded > groups * p.head_size ? padded : groups * p.head_size);
  size_t smem = sizeof(float) * (((p.head_size + 3) & ~3) + 16 + tail);
  VA_REQUIRE(smem <= 160 * 1024 - 64, VLLM_AMD_EUNSUPPORTED,
             "paged_attention_v1: max_seq_len %d needs %zu bytes of LDS (limit 160 KiB); use v2",
             p.max_seq_len, smem);
  void* out = part ? p.tmp_out : p.out;
  dim3 grid(p.num_heads, p.num_seqs, nparts);
#define GEN(SP)                                                                                   \
  do {                                                                                            \
    auto kfn = pa_generic_kernel<scalar_t, cache_t, SP>;                                          \
    if (smem > 64 * 1024) {                                                                       \
      hipError_t e = hipFuncSetAttribute((const void*)kfn,                                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");                             \
    }                                                                                             \
    hipLaunchKernelGGL(kfn, grid, dim3(kGenThreads), smem, p.stream, (scalar_t*)out, p.exp_sums,  \
                       p.max_logits, (const scalar_t*)p.query, (const cache_t*)p.key_cache,       \
                       (const cache_t*)p.value_cache, p.num_kv_heads, p.scale, p.block_tables,    \
                       p.seq_lens, p.query_start_loc, p.max_num_blocks_per_seq, p.alibi_slopes,   \
                       p.q_stride, p.kv_block_stride, p.kv_head_stride, p.k_scale, p.v_scale,     \
                       p.head_size, p.block_size, p.partition_size, p.tp_rank, p.bs_local_blocks, \
                       p.bs_vert_stride, p.bs_block_size, p.bs_head_sliding_step, p.window);      \
  } while (0)
  if (p.bs_vert_stride > 1) GEN(true); else GEN(false);
#undef GEN
  VA_LAUNCH_CHECK("paged_attention generic launch");
  return VLLM_AMD_OK;
}

int paged_attention_generic(const PagedAttnParams& p) {
  const bool fp8 = p.kv_dtype == VLLM_AMD_KV_FP8_E4M3;
  if (p.dtype == VLLM_AMD_F32) return fp8 ? launch_generic<float, uint8_t>(p) : launch_generic<float, float>(p);
  if (p.dtype == VLLM_AMD_F16) return fp8 ? launch_generic<f16_t, uint8_t>(p) : launch_generic<f16_t, f16_t>(p);
  return fp8 ? launch_generic<bf16_t, uint8_t>(p) : launch_generic<bf16_t, bf16_t>(p);
}

}  // namespace va

using namespace va;

static int validate(const PagedAttnParams& p, const char* op) {
  VA_REQUIRE(p.num_seqs >= 0 && p.num_heads > 0 && p.num_kv_heads > 0, VLLM_AMD_EINVAL,
             "%s: bad shape", op);
  VA_REQUIRE(p.num_heads % p.num_kv_heads == 0, VLLM_AMD_EINVAL,
             "%s: num_heads %d not divisible by num_kv_heads %d", op, p.num_heads, p.num_kv_heads);
  switch (p.head_size) {  // paged_attention_v1.cu:94-125
    case 32: case 64: case 80: case 96: case 112: case 120: case 128: case 192: case 256: break;
    default: VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported head size: %d", p.head_size);
  }
  switch (p.block_size) {  // :141-158
    case 8: case 16: case 32: break;
    default: VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported block size: %d", p.block_size);
  }
  VA_REQUIRE(p.dtype == VLLM_AMD_F32 || p.dtype == VLLM_AMD_F16 || p.dtype == VLLM_AMD_BF16,
             VLLM_AMD_EUNSUPPORTED, "Unsupported data type: %d", p.dtype);
  VA_REQUIRE(p.kv_dtype == VLLM_AMD_KV_AUTO || p.kv_dtype == VLLM_AMD_KV_FP8_E4M3,
             VLLM_AMD_EUNSUPPORTED, "Unsupported data type of kv cache: %d", p.kv_dtype);
  if (p.num_seqs == 0) return VLLM_AMD_OK;
  VA_REQUIRE(p.out && p.query && p.key_cache && p.value_cache && p.block_tables && p.seq_lens,
             VLLM_AMD_EINVAL, "%s: null pointer", op);
  if (p.kv_dtype != VLLM_AMD_KV_AUTO)
    VA_REQUIRE(p.k_scale && p.v_scale, VLLM_AMD_EINVAL, "%s: null k_scale/v_scale", op);
  const int x = p.kv_dtype == VLLM_AMD_KV_AUTO ? 16 / dtype_size(p.dtype) : 16;
  VA_REQUIRE(p.head_size % x == 0, VLLM_AMD_EUNSUPPORTED, "%s: head size %d %% x %d != 0", op,
             p.head_size, x);
  const int64_t es = p.kv_dtype == VLLM_AMD_KV_AUTO ? dtype_size(p.dtype) : 1;
  VA_REQUIRE((((uintptr_t)p.key_cache) & 15) == 0 && (((uintptr_t)p.value_cache) & 15) == 0 &&
                 (p.kv_block_stride * es) % 16 == 0 && (p.kv_head_stride * es) % 16 == 0,
             VLLM_AMD_EINVAL, "%s: KV cache must be 16-byte aligned", op);
  if (p.partition_size > 0) {
    VA_REQUIRE(p.exp_sums && p.max_logits && p.tmp_out, VLLM_AMD_EINVAL, "%s: null scratch", op);
    VA_REQUIRE(p.partition_size % p.block_size == 0, VLLM_AMD_EINVAL, "%s: partition/block", op);
    VA_REQUIRE(p.max_num_partitions >= 1, VLLM_AMD_EINVAL, "%s: max_num_partitions < 1", op);
  }
  return VLLM_AMD_OK;
}

static int run(const PagedAttnParams& p, const char* op) {
  int rc = validate(p, op);
  if (rc != VLLM_AMD_OK || p.num_seqs == 0) return rc;
  if (paged_attention_mfma_supported(p)) return paged_attention_mfma(p);
  rc = paged_attention_generic(p);
  if (rc != VLLM_AMD_OK) return rc;
  if (p.partition_size > 0) return paged_attention_reduce(p);
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_paged_attention_v1(
    void* out, const void* query, const void* key_cache, const void* value_cache, int num_seqs,
    int num_heads, int head_size, int num_kv_heads, float scale, const int* block_tables,
    const int* seq_lens, int max_num_blocks_per_seq, int block_size, int max_seq_len,
    const float* alibi_slopes, int dtype, int kv_dtype, const float* k_scale, const float* v_scale,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int tp_rank,
    int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, void* stream) {
  PagedAttnParams p = {};
  p.out = out; p.query = query; p.key_cache = key_cache; p.value_cache = value_cache;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.head_size = head_size;
  p.num_kv_heads = num_kv_heads; p.scale = scale; p.block_tables = block_tables;
  p.seq_lens = seq_lens; p.max_num_blocks_per_seq = max_num_blocks_per_seq;
  p.block_size = block_size; p.max_seq_len = max_seq_len; p.partition_size = 0;
  p.max_num_partitions = 1; p.alibi_slopes = alibi_slopes; p.dtype = dtype; p.kv_dtype = kv_dtype;
  p.k_scale = k_scale; p.v_scale = v_scale; p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride; p.tp_rank = tp_rank;
  p.bs_local_blocks = blocksparse_local_blocks; p.bs_vert_stride = blocksparse_vert_stride;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.stream = (hipStream_t)stream;
  return run(p, "paged_attention_v1");
}

extern "C" int vllm_amd_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int* block_tables, const int* seq_lens,
    int max_num_blocks_per_seq, int block_size, int max_seq_len, int max_num_partitions,
    const float* alibi_slopes, int dtype, int kv_dtype, const float* k_scale, const float* v_scale,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride, int tp_rank,
    int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, void* stream) {
  PagedAttnParams p = {};
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.query = query; p.key_cache = key_cache; p.value_cache = value_cache;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.head_size = head_size;
  p.num_kv_heads = num_kv_heads; p.scale = scale; p.block_tables = block_tables;
  p.seq_lens = seq_lens; p.max_num_blocks_per_seq = max_num_blocks_per_seq;
  p.block_size = block_size; p.max_seq_len = max_seq_len; p.partition_size = 512;
  p.max_num_partitions = max_num_partitions; p.alibi_slopes = alibi_slopes; p.dtype = dtype;
  p.kv_dtype = kv_dtype; p.k_scale = k_scale; p.v_scale = v_scale; p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride; p.tp_rank = tp_rank;
  p.bs_local_blocks = blocksparse_local_blocks; p.bs_vert_stride = blocksparse_vert_stride;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.stream = (hipStream_t)stream;
  VA_REQUIRE((int64_t)max_num_partitions * 512 >= max_seq_len, VLLM_AMD_EINVAL,
             "paged_attention_v2: max_num_partitions %d too small for max_seq_len %d",
             max_num_partitions, max_seq_len);
  return run(p, "paged_attention_v2");
}

extern "C" int vllm_amd_paged_attention_rocm(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int* block_tables, const int* context_lens,
    const int* query_start_loc, int max_num_blocks_per_seq, int block_size, int max_context_len,
    int max_num_partitions, const float* alibi_slopes, int dtype, int kv_dtype,
    const float* k_scale, const float* v_scale, const float* fp8_out_scale, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, void* stream) {
  PagedAttnParams p = {};
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.query = query; p.key_cache = key_cache; p.value_cache = value_cache;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.head_size = head_size;
  p.num_kv_heads = num_kv_heads; p.scale = scale; p.block_tables = block_tables;
  p.seq_lens = context_lens; p.query_start_loc = query_start_loc;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.block_size = block_size;
  p.max_seq_len = max_context_len; p.partition_size = 256;
  p.max_num_partitions = max_num_partitions; p.alibi_slopes = alibi_slopes; p.dtype = dtype;
  p.kv_dtype = kv_dtype; p.k_scale = k_scale; p.v_scale = v_scale;
  p.fp8_out_scale = fp8_out_scale; p.q_stride = q_stride; p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride; p.bs_block_size = 64; p.rocm_semantics = true;
  p.stream = (hipStream_t)stream;
  // csrc/rocm/attention.cu:3494-3539 accepts half/bf16 queries only
  VA_REQUIRE(dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16, VLLM_AMD_EUNSUPPORTED,
             "Unsupported data type: %d", dtype);
  VA_REQUIRE((int64_t)max_num_partitions * 256 >= max_context_len, VLLM_AMD_EINVAL,
             "paged_attention: max_num_partitions %d too small for max_context_len %d",
             max_num_partitions, max_context_len);
  return run(p, "paged_attention");
}

// The decode side of chunked_prefill_paged_decode with a sliding window
// (vllm/attention/ops/chunked_prefill_paged_decode.py:27-212 `kernel_paged_attention_2d`, launch
// :340-381): the arguments of vllm_amd_paged_attention_rocm plus `sliding_window` (<= 0: none).
// A decode token at position seq_len - 1 sees the keys [seq_len - sliding_window, seq_len).
extern "C" int vllm_amd_paged_attention_2d(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int* block_tables, const int* seq_lens,
    const int* query_start_loc, int max_num_blocks_per_seq, int block_size, int max_seq_len,
    int max_num_partitions, const float* alibi_slopes, int dtype, int kv_dtype,
    const float* k_scale, const float* v_scale, const float* fp8_out_scale, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int sliding_window, void* stream) {
  PagedAttnParams p = {};
  p.out = out; p.exp_sums = exp_sums; p.max_logits = max_logits; p.tmp_out = tmp_out;
  p.query = query; p.key_cache = key_cache; p.value_cache = value_cache;
  p.num_seqs = num_seqs; p.num_heads = num_heads; p.head_size = head_size;
  p.num_kv_heads = num_kv_heads; p.scale = scale; p.block_tables = block_tables;
  p.seq_lens = seq_lens; p.query_start_loc = query_start_loc;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.block_size = block_size;
  p.max_seq_len = max_seq_len; p.partition_size = 256;
  p.max_num_partitions = max_num_partitions; p.alibi_slopes = alibi_slopes; p.dtype = dtype;
  p.kv_dtype = kv_dtype; p.k_scale = k_scale; p.v_scale = v_scale;
  p.fp8_out_scale = fp8_out_scale; p.q_stride = q_stride; p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride; p.bs_block_size = 64; p.rocm_semantics = true;
  p.window = sliding_window > 0 ? sliding_window : 0;
  p.stream = (hipStream_t)stream;
  VA_REQUIRE(dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16, VLLM_AMD_EUNSUPPORTED,
             "Unsupported data type: %d", dtype);
  VA_REQUIRE((int64_t)max_num_partitions * 256 >= max_seq_len, VLLM_AMD_EINVAL,
             "paged_attention_2d: max_num_partitions %d too small for max_seq_len %d",
             max_num_partitions, max_seq_len);
  return run(p, "paged_attention_2d");
}

// Tuning knobs for A/B measurements in one process (not part of the reference surface).
extern "C" int vllm_amd_set_tuning(const char* key, int value) {
  VA_REQUIRE(key != nullptr, VLLM_AMD_EINVAL, "set_tuning: null key");
  if (strcmp(key, "pa_variant") == 0) { set_pa_variant(value); return VLLM_AMD_OK; }
  if (strcmp(key, "pa_sp") == 0) { set_pa_sp(value); return VLLM_AMD_OK; }
  if (strcmp(key, "pa_group") == 0) { set_pa_group(value); return VLLM_AMD_OK; }
  if (strcmp(key, "pa_balance") == 0) { set_pa_balance(value); return VLLM_AMD_OK; }
  if (strcmp(key, "pa_order") == 0) { set_pa_order(value); return VLLM_AMD_OK; }
  if (strcmp(key, "mm_kernel") == 0) { g_mm_kernel = value; return VLLM_AMD_OK; }
  if (strcmp(key, "cache_runs") == 0) { g_cache_runs = value; return VLLM_AMD_OK; }
  if (strcmp(key, "cache_hpw") == 0) { g_cache_hpw = value; return VLLM_AMD_OK; }
  if (strcmp(key, "uni_hp") == 0) { g_uni_hp = value; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_splitk") == 0) { g_mm_splitk = value; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_tail") == 0) { g_mm_tail = value; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_patch") == 0) { g_mm_patch = value; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_store") == 0) { g_mm_store = value; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_prio") == 0) { g_mm_prio = value; return VLLM_AMD_OK; }
  if (strcmp(key, "pf_rows") == 0) { g_pf_rows = value; return VLLM_AMD_OK; }
  if (strcmp(key, "uni_decode") == 0) { g_uni_decode = value; return VLLM_AMD_OK; }
  if (strcmp(key, "pf_fa") == 0) { g_pf_fa = value; return VLLM_AMD_OK; }
  if (strcmp(key, "pf_fa_abl") == 0) { g_pf_fa_abl = value; return VLLM_AMD_OK; }
  if (strcmp(key, "pf_fa_stagger") == 0) { g_pf_fa_stagger = value; return VLLM_AMD_OK; }
  VA_REQUIRE(false, VLLM_AMD_EINVAL, "set_tuning: unknown key %s", key);
}

// The current value of a tuning key (the torch shim asks for "pa_balance": it only registers the
// workspace from paged_attention_v1 when the balanced form is on).
extern "C" int vllm_amd_get_tuning(const char* key, int* value) {
  VA_REQUIRE(key != nullptr && value != nullptr, VLLM_AMD_EINVAL, "get_tuning: null pointer");
  if (strcmp(key, "pa_balance") == 0) { *value = get_pa_balance(); return VLLM_AMD_OK; }
  if (strcmp(key, "mm_kernel") == 0) { *value = g_mm_kernel; return VLLM_AMD_OK; }
  if (strcmp(key, "mm_splitk") == 0) { *value = g_mm_splitk; return VLLM_AMD_OK; }
  VA_REQUIRE(false, VLLM_AMD_EINVAL, "get_tuning: unknown key %s", key);
}
