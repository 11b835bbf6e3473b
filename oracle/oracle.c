/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's quantized decode hot path.  It is
 * the parity checker for the HIP kernels in vllm_amd/csrc and (kind "port")
 * the CPU baseline in bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - attention (bf16/fp16/fp32 KV, block 16) and reshape_and_cache (auto):
 *     checked against oracle/_ref/librefcpu.so, the reference's own
 *     csrc/cpu/attention.cpp + cache.cpp compiled from where they lie;
 *   - fp8 e4m3fn encode/decode: checked exhaustively against torch's CPU
 *     float8_e4m3fn casts (tests/test_oracle.py);
 *   - fp8 quant / scaled_mm: checked against torch restatements of the
 *     formulas in the reference's tests (tests/kernels/quant_utils.py:22-97,
 *     tests/kernels/utils.py:1212-1251), fixtures in tests/golden/.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows.
 */
#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };
enum { KV_AUTO = 0, KV_FP8_E4M3 = 1 };

/* ------------------------------------------------------------------ */
/* scalar conversions                                                  */
/* ------------------------------------------------------------------ */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
/* round-to-nearest-even, NaN stays NaN (what v_cvt_pk_bf16_f32 /
 * __float2bfloat16 do). */
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f32_to_f16(float f) {
  return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
}

static inline float load_as_f32(const void* p, int64_t i, int dt) {
  switch (dt) {
    case DT_F32: return ((const float*)p)[i];
    case DT_F16: return f16_to_f32(((const uint16_t*)p)[i]);
    default:     return bf16_to_f32(((const uint16_t*)p)[i]);
  }
}
static inline void store_from_f32(void* p, int64_t i, int dt, float v) {
  switch (dt) {
    case DT_F32: ((float*)p)[i] = v; break;
    case DT_F16: ((uint16_t*)p)[i] = f32_to_f16(v); break;
    default:     ((uint16_t*)p)[i] = f32_to_bf16(v); break;
  }
}
/* round a float through the scalar type (f32: identity) */
static inline float round_to(float v, int dt) {
  switch (dt) {
    case DT_F32: return v;
    case DT_F16: return f16_to_f32(f32_to_f16(v));
    default:     return bf16_to_f32(f32_to_bf16(v));
  }
}

/* OCP fp8 e4m3fn decode: 1-4-3, bias 7, no inf, S.1111.111 = NaN. */
float oracle_fp8_e4m3_to_f32(uint8_t b) {
  uint32_t s = b >> 7, e = (b >> 3) & 0xf, m = b & 7;
  float v;
  if (e == 0xf && m == 7) return u2f(0x7fc00000u | (s << 31));
  if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf((float)(8 + m), (int)e - 10);
  return s ? -v : v;
}

/*
 * float -> fp8 e4m3fn, round-to-nearest-even with finite saturation, as
 * __hip_cvt_float_to_fp8(x, __HIP_SATFINITE, __HIP_E4M3) does on gfx950
 * (/opt/rocm/include/hip/amd_detail/amd_hip_fp8.h:548-592: NaN/Inf bypass the
 * clamp, everything else is fmed3(x, 448, -448) then the RNE hardware cvt).
 * Used by: csrc/quantization/fp8/amd/quant_utils.cuh:29-40, 231-245, 540-591.
 * NaN and +-Inf both encode as S.1111.111.
 */
uint8_t oracle_f32_to_fp8_e4m3(float x) {
  uint32_t u = f2u(x);
  uint8_t s = (uint8_t)((u >> 31) << 7);
  if ((u & 0x7f800000u) == 0x7f800000u) return s | 0x7f; /* NaN / Inf */
  float a = fabsf(x);
  if (a >= 448.0f) return s | 0x7e;
  if (a < ldexpf(1.0f, -6)) {
    /* subnormal target: multiples of 2^-9; rintf = RNE in default mode */
    float q = rintf(a * 512.0f);
    return s | (uint8_t)q; /* q == 8 encodes 2^-6 = 0.0001.000: correct */
  }
  int ex;
  (void)frexpf(a, &ex);           /* a = f * 2^ex, f in [0.5,1) */
  int e = ex - 1;                  /* a = 1.m * 2^e */
  float q = rintf(ldexpf(a, 3 - e)); /* in [8,16] */
  if (q >= 16.0f) { q = 8.0f; e += 1; }
  if (e > 8 || (e == 8 && q > 14.0f)) return s | 0x7e;
  return s | (uint8_t)(((e + 7) << 3) | ((int)q - 8));
}

/* fp8::scaled_convert<uint8_t, scalar_t>(x, scale)
 * csrc/quantization/fp8/amd/quant_utils.cuh:486-495 (half: divide, round the
 * quotient to half, then half->float->fp8), :540-547 (bf16), :585-591 (f32). */
static inline uint8_t scaled_to_fp8(float x, float scale, int dt) {
  float q = x / scale;
  if (dt == DT_F16) q = f16_to_f32(f32_to_f16(q));
  return oracle_f32_to_fp8_e4m3(q);
}
/* fp8::scaled_convert<scalar_t, uint8_t>(b, scale)
 * quant_utils.cuh:338-345 (bf16), :385-392 (f32), :436-457 (half): the
 * product float(fp8)*scale is rounded once to the scalar type. */
static inline float scaled_from_fp8(uint8_t b, float scale, int dt) {
  return round_to(oracle_fp8_e4m3_to_f32(b) * scale, dt);
}

/* ------------------------------------------------------------------ */
/* cache writes                                                        */
/* ------------------------------------------------------------------ */
static inline int elem_size(int dt) { return dt == DT_F32 ? 4 : 2; }

static inline void put_cache(void* cache, int64_t idx, const void* src,
                             int64_t sidx, int dt, int kv_dt, float scale) {
  if (kv_dt == KV_AUTO) {
    int es = elem_size(dt);
    memcpy((char*)cache + idx * es, (const char*)src + sidx * es, es);
  } else {
    ((uint8_t*)cache)[idx] = scaled_to_fp8(load_as_f32(src, sidx, dt), scale, dt);
  }
}

/* csrc/cache_kernels.cu:210-262 (kernel), :369-395 (launcher);
 * CPU twin csrc/cpu/cache.cpp:41-88. */
void oracle_reshape_and_cache(const void* key, const void* value,
                              void* key_cache, void* value_cache,
                              const int64_t* slot_mapping, int num_tokens,
                              int num_heads, int head_size, int block_size,
                              int x, int64_t key_stride, int64_t value_stride,
                              int dt, int kv_dt, float k_scale, float v_scale) {
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    int64_t blk = slot / block_size, off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      int h = i / head_size, d = i % head_size;
      int64_t kidx = blk * num_heads * (head_size / x) * block_size * x +
                     (int64_t)h * (head_size / x) * block_size * x +
                     (int64_t)(d / x) * block_size * x + off * x + d % x;
      int64_t vidx = blk * num_heads * head_size * block_size +
                     (int64_t)h * head_size * block_size +
                     (int64_t)d * block_size + off;
      put_cache(key_cache, kidx, key, t * key_stride + i, dt, kv_dt, k_scale);
      put_cache(value_cache, vidx, value, t * value_stride + i, dt, kv_dt,
                v_scale);
    }
  }
}

/* csrc/cache_kernels.cu:264-306 (kernel), :412-450 (launcher). */
void oracle_reshape_and_cache_flash(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t key_stride, int64_t value_stride,
    int64_t block_stride, int64_t page_stride, int64_t head_stride, int dt,
    int kv_dt, float k_scale, float v_scale) {
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    int64_t blk = slot / block_size, off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      int h = i / head_size, d = i % head_size;
      int64_t idx = blk * block_stride + off * page_stride + h * head_stride + d;
      put_cache(key_cache, idx, key, t * key_stride + i, dt, kv_dt, k_scale);
      put_cache(value_cache, idx, value, t * value_stride + i, dt, kv_dt,
                v_scale);
    }
  }
}

/* csrc/cache_kernels.cu:505-576: to_fp8 != 0 quantises src(dt) -> dst(u8),
 * else dequantises src(u8) -> dst(dt). */
void oracle_convert_fp8(void* dst, const void* src, int64_t n, float scale,
                        int dt, int to_fp8) {
  for (int64_t i = 0; i < n; ++i) {
    if (to_fp8)
      ((uint8_t*)dst)[i] = scaled_to_fp8(load_as_f32(src, i, dt), scale, dt);
    else
      store_from_f32(dst, i, dt,
                     oracle_fp8_e4m3_to_f32(((const uint8_t*)src)[i]) * scale);
  }
}

/* ------------------------------------------------------------------ */
/* fp8 quantisation of activations / weights                           */
/* ------------------------------------------------------------------ */
#define FP8_MAX 448.0f

/* csrc/quantization/fp8/common.cuh:38-56 */
static inline uint8_t quant_conv(float val, float scale, int inverted) {
  float x = inverted ? val * scale : val / scale;
  float r = fmaxf(-FP8_MAX, fminf(x, FP8_MAX));
  return oracle_f32_to_fp8_e4m3(r);
}

/* csrc/quantization/fp8/common.cu:14-26, 87-110 */
void oracle_static_scaled_fp8_quant(uint8_t* out, const void* in, float scale,
                                    int64_t n, int dt) {
  float inv = 1.0f / scale;
  for (int64_t i = 0; i < n; ++i) out[i] = quant_conv(load_as_f32(in, i, dt), inv, 1);
}

/* csrc/quantization/fp8/common.cu:112-139 + common.cuh:64-97: the scale is
 * max|x| / 448 with no lower clamp; the caller pre-zeroes *scale
 * (vllm/_custom_ops.py:1297), atomicMax semantics = max with existing. */
void oracle_dynamic_scaled_fp8_quant(uint8_t* out, const void* in, float* scale,
                                     int64_t n, int dt) {
  float amax = 0.f;
  for (int64_t i = 0; i < n; ++i) amax = fmaxf(amax, fabsf(load_as_f32(in, i, dt)));
  float s = amax / FP8_MAX;
  if (s > *scale) *scale = s;
  oracle_static_scaled_fp8_quant(out, in, *scale, n, dt);
}

/* csrc/quantization/fp8/common.cu:28-83, 141-171;
 * min scale csrc/quantization/utils.cuh:44-52 */
void oracle_dynamic_per_token_scaled_fp8_quant(uint8_t* out, const void* in,
                                               float* scales,
                                               const float* scale_ub,
                                               int64_t num_tokens, int hidden,
                                               int dt) {
  const float min_scale = 1.0f / (FP8_MAX * 512.0f);
  for (int64_t t = 0; t < num_tokens; ++t) {
    int64_t off = t * hidden;
    float amax = 0.f;
    for (int i = 0; i < hidden; ++i)
      amax = fmaxf(amax, fabsf(load_as_f32(in, off + i, dt)));
    float s = scale_ub ? fminf(amax, *scale_ub) : amax;
    s = fmaxf(s / FP8_MAX, min_scale);
    scales[t] = s;
    for (int i = 0; i < hidden; ++i)
      out[off + i] = quant_conv(load_as_f32(in, off + i, dt), s, 0);
  }
}

/* ------------------------------------------------------------------ */
/* scaled_mm                                                           */
/* ------------------------------------------------------------------ */
/* out[M,N] = ((sa (.) A) . (sb (.) B)).to(out_dtype) (+ bias)
 * vllm/_custom_ops.py:697-742; Triton body
 * vllm/model_executor/layers/quantization/compressed_tensors/triton_scaled_mm.py:87-110
 * (fp32 accumulate, scale_a then scale_b applied to the accumulator, cast,
 * then bias added in the output dtype); test reference
 * tests/kernels/utils.py:1212-1251.  a: [M,K] row-major fp8, lda;
 * b: [K,N] "column-major" i.e. element (k,n) at b[n*ldb + k]. */
void oracle_scaled_mm(void* out, const uint8_t* a, const uint8_t* b,
                      const float* a_scales, const float* b_scales,
                      const void* bias, int M, int N, int K, int64_t lda,
                      int64_t ldb, int a_scale_numel, int b_scale_numel,
                      int out_dt) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = oracle_fp8_e4m3_to_f32((uint8_t)i);
#pragma omp parallel for schedule(static)
  for (int m = 0; m < M; ++m) {
    float* arow = (float*)malloc(sizeof(float) * K);
    for (int k = 0; k < K; ++k) arow[k] = lut[a[m * lda + k]];
    float sa = a_scales[a_scale_numel == 1 ? 0 : m];
    for (int n = 0; n < N; ++n) {
      const uint8_t* bcol = b + n * ldb;
      /* products of two e4m3 values are exact in fp32; accumulate in double
       * so the oracle is order-independent, then round once to fp32. */
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += (double)(arow[k] * lut[bcol[k]]);
      float v = (float)acc;
      v = v * sa;
      v = v * b_scales[b_scale_numel == 1 ? 0 : n];
      v = round_to(v, out_dt);
      if (bias) v = round_to(v + load_as_f32(bias, n, out_dt), out_dt);
      store_from_f32(out, (int64_t)m * N + n, out_dt, v);
    }
    free(arow);
  }
}

/* ------------------------------------------------------------------ */
/* paged attention                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
  const void* q;
  const void* kc;
  const void* vc;
  int num_seqs, num_heads, head_size, num_kv_heads;
  float scale;
  const int* block_tables;
  const int* seq_lens;
  int max_num_blocks_per_seq, block_size;
  const float* alibi;
  int dt, kv_dt;
  float k_scale, v_scale;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  int tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_step;
} attn_args;

static inline float load_k(const attn_args* a, int64_t blk, int kvh, int d,
                           int tok_in_blk) {
  int x = a->kv_dt == KV_AUTO ? 16 / elem_size(a->dt) : 16;
  int64_t idx = blk * a->kv_block_stride + (int64_t)kvh * a->kv_head_stride +
                (int64_t)(d / x) * a->block_size * x + (int64_t)tok_in_blk * x +
                d % x;
  if (a->kv_dt == KV_AUTO) return load_as_f32(a->kc, idx, a->dt);
  return scaled_from_fp8(((const uint8_t*)a->kc)[idx], a->k_scale, a->dt);
}
static inline float load_v(const attn_args* a, int64_t blk, int kvh, int d,
                           int tok_in_blk) {
  int64_t idx = blk * a->kv_block_stride + (int64_t)kvh * a->kv_head_stride +
                (int64_t)d * a->block_size + tok_in_blk;
  if (a->kv_dt == KV_AUTO) return load_as_f32(a->vc, idx, a->dt);
  return scaled_from_fp8(((const uint8_t*)a->vc)[idx], a->v_scale, a->dt);
}

/*
 * One (seq, head, partition) of paged_attention_kernel,
 * csrc/attention/attention_kernels.cuh:85-491.  partition_size == 0 means
 * v1 (whole sequence).  Writes head_size values through out/out_idx and, for
 * v2, *max_logit / *exp_sum.  Returns 0 if the partition is empty (:111-114).
 */
static int attn_partition(const attn_args* a, int seq, int head, int part,
                          int partition_size, float* logits, void* out,
                          int64_t out_idx, float* max_logit, float* exp_sum_out) {
  const int BS = a->block_size, D = a->head_size;
  const int seq_len = a->seq_lens[seq];
  if (partition_size > 0 && part * partition_size >= seq_len) return 0;
  const int num_seq_blocks = (seq_len + BS - 1) / BS;
  const int nbpp = partition_size > 0 ? partition_size / BS : num_seq_blocks;
  const int start_block = partition_size > 0 ? part * nbpp : 0;
  int end_block = start_block + nbpp;
  if (end_block > num_seq_blocks) end_block = num_seq_blocks;
  const int nblocks = end_block - start_block;
  const int start_tok = start_block * BS;
  int end_tok = start_tok + nblocks * BS;
  if (end_tok > seq_len) end_tok = seq_len;
  const int num_tokens = end_tok - start_tok;
  const int qpk = a->num_heads / a->num_kv_heads;
  const int kvh = head / qpk;
  const float slope = a->alibi ? a->alibi[head] : 0.f;
  const int* bt = a->block_tables + (int64_t)seq * a->max_num_blocks_per_seq;
  const int sparse = a->bs_vert_stride > 1; /* IS_BLOCK_SPARSE, v1.cu:63 */
  int bs_off = 0, q_bs_id = 0;
  if (sparse) { /* :207-220 */
    q_bs_id = (seq_len - 1) / a->bs_block_size;
    if (a->bs_head_step >= 0)
      bs_off = (a->tp_rank * a->num_heads + head) * a->bs_head_step + 1;
    else
      bs_off = (a->tp_rank * a->num_kv_heads + kvh) * (-a->bs_head_step) + 1;
  }
  const void* qp = a->q;
  const int64_t qoff = (int64_t)seq * a->q_stride + (int64_t)head * D;

  float qk_max = -FLT_MAX;
  for (int b = start_block; b < end_block; ++b) { /* :222-303 */
    if (sparse) {
      int kid = b * BS / a->bs_block_size;
      int remote = ((kid + bs_off) % a->bs_vert_stride) == 0;
      int local = kid > q_bs_id - a->bs_local_blocks;
      if (!remote && !local) {
        for (int t = 0; t < BS; ++t) logits[b * BS + t - start_tok] = -FLT_MAX;
        continue;
      }
    }
    int64_t pb = bt[b];
    for (int t = 0; t < BS; ++t) {
      int tok = b * BS + t;
      float acc = 0.f;
      for (int d = 0; d < D; ++d)
        acc += load_as_f32(qp, qoff + d, a->dt) * load_k(a, pb, kvh, d, t);
      float qk = a->scale * acc;
      qk += (slope != 0.f) ? slope * (float)(tok - seq_len + 1) : 0.f;
      int mask = tok >= seq_len;
      logits[tok - start_tok] = mask ? 0.f : qk;
      if (!mask) qk_max = fmaxf(qk_max, qk);
    }
  }
  float exp_sum = 0.f; /* :327-341 */
  for (int i = 0; i < num_tokens; ++i) {
    float v = expf(logits[i] - qk_max);
    logits[i] = v;
    exp_sum += v;
  }
  const float inv_sum = 1.f / (exp_sum + 1e-6f);
  for (int i = 0; i < num_tokens; ++i) logits[i] *= inv_sum;
  if (partition_size > 0) { *max_logit = qk_max; *exp_sum_out = exp_sum; }

  /* :354-429: probabilities cast to scalar_t, products rounded in scalar_t
   * (attention_generic.cuh:40-43 dot = sum(mul<T,T,T>), dtype_bfloat16.cuh
   * :181-188, :410-413), summed in fp32 per V_VEC_SIZE group. */
  const int VV = (16 / elem_size(a->dt)) < BS ? (16 / elem_size(a->dt)) : BS;
  for (int d = 0; d < D; ++d) {
    float acc = 0.f;
    for (int b = start_block; b < end_block; ++b) {
      if (sparse) {
        int vid = b * BS / a->bs_block_size;
        if (!((vid + bs_off) % a->bs_vert_stride == 0) &&
            !(vid > q_bs_id - a->bs_local_blocks))
          continue;
      }
      int64_t pb = bt[b];
      for (int g = 0; g < BS; g += VV) {
        float prods[8];
        for (int j = 0; j < VV; ++j) {
          int tok = b * BS + g + j;
          float p = round_to(logits[tok - start_tok], a->dt);
          float v = load_v(a, pb, kvh, d, g + j);
          if (b == num_seq_blocks - 1 && tok >= seq_len) v = 0.f;
          prods[j] = round_to(p * v, a->dt);
        }
        float s;
        if (VV == 8)
          s = ((prods[0] + prods[1]) + (prods[2] + prods[3])) +
              ((prods[4] + prods[5]) + (prods[6] + prods[7]));
        else if (VV == 4)
          s = (prods[0] + prods[1]) + (prods[2] + prods[3]);
        else { s = 0.f; for (int j = 0; j < VV; ++j) s += prods[j]; }
        acc += s;
      }
    }
    store_from_f32(out, out_idx + d, a->dt, acc);
  }
  return 1;
}

static attn_args mk_args(const void* q, const void* kc, const void* vc,
                         int num_seqs, int num_heads, int head_size,
                         int num_kv_heads, float scale, const int* block_tables,
                         const int* seq_lens, int max_num_blocks_per_seq,
                         int block_size, const float* alibi, int dt, int kv_dt,
                         float k_scale, float v_scale, int64_t q_stride,
                         int64_t kv_block_stride, int64_t kv_head_stride,
                         int tp_rank, int bl, int bv, int bb, int bh) {
  attn_args a = {q, kc, vc, num_seqs, num_heads, head_size, num_kv_heads, scale,
                 block_tables, seq_lens, max_num_blocks_per_seq, block_size,
                 alibi, dt, kv_dt, k_scale, v_scale, q_stride, kv_block_stride,
                 kv_head_stride, tp_rank, bl, bv, bb, bh};
  return a;
}

/* csrc/attention/paged_attention_v1.cu:46-182 */
void oracle_paged_attention_v1(
    void* out, const void* q, const void* kc, const void* vc, int num_seqs,
    int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq,
    int block_size, const float* alibi, int dt, int kv_dt, float k_scale,
    float v_scale, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int tp_rank, int bl, int bv, int bb, int bh) {
  attn_args a = mk_args(q, kc, vc, num_seqs, num_heads, head_size, num_kv_heads,
                        scale, block_tables, seq_lens, max_num_blocks_per_seq,
                        block_size, alibi, dt, kv_dt, k_scale, v_scale, q_stride,
                        kv_block_stride, kv_head_stride, tp_rank, bl, bv, bb, bh);
  const int maxlog = max_num_blocks_per_seq * block_size + block_size;
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)maxlog);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s)
      for (int h = 0; h < num_heads; ++h)
        attn_partition(&a, s, h, 0, 0, logits, out,
                       ((int64_t)s * num_heads + h) * head_size, NULL, NULL);
    free(logits);
  }
}

/* reduce: csrc/attention/attention_kernels.cuh:562-664.  eps = 1e-6. */
static void attn_reduce(void* out, int64_t out_idx, const float* exp_sums,
                        const float* max_logits, const void* tmp_out,
                        int64_t tmp_idx, int num_partitions, int head_size,
                        int dt, float out_mul, int mode) {
  /* mode 0: v2 reduce; 1: ROCm reduce, scalar out; 2: ROCm reduce, fp8 out */
  if (num_partitions == 1 && mode == 0) { /* :577-589 copy */
    for (int i = 0; i < head_size; ++i)
      store_from_f32(out, out_idx + i, dt, load_as_f32(tmp_out, tmp_idx + i, dt));
    return;
  }
  float m = -FLT_MAX;
  for (int j = 0; j < num_partitions; ++j) m = fmaxf(m, max_logits[j]);
  float* w = (float*)malloc(sizeof(float) * (size_t)num_partitions);
  float g = 0.f;
  for (int j = 0; j < num_partitions; ++j) {
    w[j] = exp_sums[j] * expf(max_logits[j] - m);
    g += w[j];
  }
  const float inv = 1.f / (g + 1e-6f);
  for (int i = 0; i < head_size; ++i) {
    float acc = 0.f;
    if (mode == 0) {
      for (int j = 0; j < num_partitions; ++j)
        acc += load_as_f32(tmp_out, tmp_idx + (int64_t)j * head_size + i, dt) *
               w[j] * inv;
      store_from_f32(out, out_idx + i, dt, acc);
    } else {
      /* csrc/rocm/attention.cu:1419-1474: sum first, then * inv, * 1/scale */
      for (int j = 0; j < num_partitions; ++j)
        acc += load_as_f32(tmp_out, tmp_idx + (int64_t)j * head_size + i, dt) * w[j];
      acc *= inv;
      acc *= out_mul;
      if (mode == 2) ((uint8_t*)out)[out_idx + i] = oracle_f32_to_fp8_e4m3(acc);
      else store_from_f32(out, out_idx + i, dt, acc);
    }
  }
  free(w);
}

/* csrc/attention/paged_attention_v2.cu:46-192; partition_size = 512 in the
 * reference (vllm/attention/ops/paged_attn.py:16). */
void oracle_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* q,
    const void* kc, const void* vc, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int* block_tables, const int* seq_lens,
    int max_num_blocks_per_seq, int block_size, const float* alibi, int dt,
    int kv_dt, float k_scale, float v_scale, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int tp_rank, int bl, int bv,
    int bb, int bh, int partition_size, int max_num_partitions) {
  attn_args a = mk_args(q, kc, vc, num_seqs, num_heads, head_size, num_kv_heads,
                        scale, block_tables, seq_lens, max_num_blocks_per_seq,
                        block_size, alibi, dt, kv_dt, k_scale, v_scale, q_stride,
                        kv_block_stride, kv_head_stride, tp_rank, bl, bv, bb, bh);
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)(partition_size + block_size));
#pragma omp for collapse(3) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s)
      for (int h = 0; h < num_heads; ++h)
        for (int p = 0; p < max_num_partitions; ++p) {
          int64_t sidx = ((int64_t)s * num_heads + h) * max_num_partitions + p;
          attn_partition(&a, s, h, p, partition_size, logits, tmp_out,
                         sidx * head_size, &max_logits[sidx], &exp_sums[sidx]);
        }
    free(logits);
  }
#pragma omp parallel for collapse(2) schedule(static)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) {
      int np = (seq_lens[s] + partition_size - 1) / partition_size;
      int64_t sidx = ((int64_t)s * num_heads + h) * max_num_partitions;
      if (np < 1) continue;
      attn_reduce(out, ((int64_t)s * num_heads + h) * head_size, exp_sums + sidx,
                  max_logits + sidx, tmp_out, sidx * head_size, np, head_size,
                  dt, 1.f, 0);
    }
}

/*
 * ROCm custom paged attention, csrc/rocm/attention.cu:262-765 (partition body),
 * :1271-1475 (reduce), :3073-3228 (launcher).  Partition = 256 tokens.
 * Differences from v2 that matter numerically:
 *  - fp8 K/V enter the dot products exactly (fp8 -> 16 bit is exact) and
 *    k_scale is folded into the softmax scale (:514-518), v_scale multiplies
 *    the PV accumulator (:714-717);
 *  - probabilities are normalised per partition (1/(sum+1e-6)) and rounded RNE
 *    to the scalar type (:625-644), PV products are NOT rounded (MFMA fp32);
 *  - out rows are indexed by query_start_loc when given, and sequences whose
 *    query length != 1 are skipped (:296-299, :1291-1294);
 *  - optional fp8 output: acc * 1/fp8_out_scale -> e4m3 (:1460-1474).
 */
void oracle_paged_attention_rocm(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* q,
    const void* kc, const void* vc, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int* block_tables, const int* seq_lens,
    const int* query_start_loc, int max_num_blocks_per_seq, int block_size,
    const float* alibi, int dt, int kv_dt, float k_scale, float v_scale,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int max_num_partitions, const float* fp8_out_scale) {
  const int PS = 256, BS = block_size, D = head_size;
  attn_args a = mk_args(q, kc, vc, num_seqs, num_heads, head_size, num_kv_heads,
                        scale, block_tables, seq_lens, max_num_blocks_per_seq,
                        block_size, alibi, dt, kv_dt, 1.0f, 1.0f, q_stride,
                        kv_block_stride, kv_head_stride, 0, 0, 0, 64, 0);
  const float scale2 = kv_dt == KV_AUTO ? scale : scale * k_scale;
  const int qpk = num_heads / num_kv_heads;
#pragma omp parallel for collapse(3) schedule(dynamic, 1)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h)
      for (int p = 0; p < max_num_partitions; ++p) {
        if (query_start_loc && query_start_loc[s + 1] - query_start_loc[s] != 1) continue;
        const int ctx = seq_lens[s];
        const int start = p * PS;
        if (start >= ctx) continue;
        const int end = start + PS < ctx ? start + PS : ctx;
        const int64_t qrow = query_start_loc ? query_start_loc[s] : s;
        const int64_t qoff = qrow * q_stride + (int64_t)h * D;
        const int kvh = h / qpk;
        const int* bt = block_tables + (int64_t)s * max_num_blocks_per_seq;
        const float slope = alibi ? alibi[h] : 0.f;
        float logits[256];
        float m = -FLT_MAX;
        for (int tok = start; tok < end; ++tok) {
          int64_t pb = bt[tok / BS];
          float acc = 0.f;
          for (int d = 0; d < D; ++d)
            acc += load_as_f32(q, qoff + d, dt) * load_k(&a, pb, kvh, d, tok % BS);
          float v = acc * scale2;
          if (alibi) v += slope * (float)(tok - ctx + 1);
          logits[tok - start] = v;
          m = fmaxf(m, v);
        }
        float sum = 0.f;
        for (int i = 0; i < end - start; ++i) { logits[i] = expf(logits[i] - m); sum += logits[i]; }
        const float inv = 1.f / (sum + 1e-6f);
        for (int i = 0; i < end - start; ++i) logits[i] = round_to(logits[i] * inv, dt);
        int64_t sidx = ((int64_t)s * num_heads + h) * max_num_partitions + p;
        max_logits[sidx] = m;
        exp_sums[sidx] = sum;
        for (int d = 0; d < D; ++d) {
          float acc = 0.f;
          for (int tok = start; tok < end; ++tok)
            acc += logits[tok - start] * load_v(&a, bt[tok / BS], kvh, d, tok % BS);
          if (kv_dt != KV_AUTO) acc *= v_scale;
          store_from_f32(tmp_out, sidx * D + d, dt, acc);
        }
      }
#pragma omp parallel for collapse(2) schedule(static)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) {
      if (query_start_loc && query_start_loc[s + 1] - query_start_loc[s] != 1) continue;
      int np = (seq_lens[s] + PS - 1) / PS;
      if (np < 1) continue;
      int64_t sidx = ((int64_t)s * num_heads + h) * max_num_partitions;
      const int64_t orow = query_start_loc ? query_start_loc[s] : s;
      /* the ROCm reduce always runs the weighted sum (no copy shortcut) */
      attn_reduce(out, (orow * num_heads + h) * D, exp_sums + sidx,
                  max_logits + sidx, tmp_out, sidx * D, np, D, dt,
                  fp8_out_scale ? 1.0f / *fp8_out_scale : 1.0f,
                  fp8_out_scale ? 2 : 1);
    }
}

/* fp64 "ground truth" attention (no intermediate rounding) used by tests to
 * bound BOTH the oracle's and the kernel's error.  Not a reference restatement. */
void oracle_paged_attention_f64(
    double* out, const void* q, const void* kc, const void* vc, int num_seqs,
    int num_heads, int head_size, int num_kv_heads, float scale,
    const int* block_tables, const int* seq_lens, int max_num_blocks_per_seq,
    int block_size, const float* alibi, int dt, int kv_dt, float k_scale,
    float v_scale, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride) {
  attn_args a = mk_args(q, kc, vc, num_seqs, num_heads, head_size, num_kv_heads,
                        scale, block_tables, seq_lens, max_num_blocks_per_seq,
                        block_size, alibi, dt, kv_dt, 1.0f, 1.0f, q_stride,
                        kv_block_stride, kv_head_stride, 0, 0, 0, 64, 0);
  const int qpk = num_heads / num_kv_heads, BS = block_size, D = head_size;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) {
      const int ctx = seq_lens[s], kvh = h / qpk;
      const int* bt = block_tables + (int64_t)s * max_num_blocks_per_seq;
      double* lg = (double*)malloc(sizeof(double) * (size_t)(ctx > 0 ? ctx : 1));
      double m = -1e300;
      for (int tok = 0; tok < ctx; ++tok) {
        double acc = 0;
        for (int d = 0; d < D; ++d)
          acc += (double)load_as_f32(q, (int64_t)s * q_stride + (int64_t)h * D + d, dt) *
                 (double)load_k(&a, bt[tok / BS], kvh, d, tok % BS);
        lg[tok] = acc * (double)scale * (kv_dt == KV_AUTO ? 1.0 : (double)k_scale) +
                  (alibi ? (double)alibi[h] * (tok - ctx + 1) : 0.0);
        if (lg[tok] > m) m = lg[tok];
      }
      double sum = 0;
      for (int tok = 0; tok < ctx; ++tok) { lg[tok] = exp(lg[tok] - m); sum += lg[tok]; }
      for (int d = 0; d < D; ++d) {
        double acc = 0;
        for (int tok = 0; tok < ctx; ++tok)
          acc += lg[tok] * (double)load_v(&a, bt[tok / BS], kvh, d, tok % BS);
        out[((int64_t)s * num_heads + h) * D + d] =
            ctx > 0 ? acc / sum * (kv_dt == KV_AUTO ? 1.0 : (double)v_scale) : 0.0;
      }
      free(lg);
    }
}

/* ------------------------------------------------------------------ */
/* fused neighbours of the GEMM                       */
/* ------------------------------------------------------------------ */
/* 1/rms as the kernels compute it: rsqrtf(ss / hidden + eps); the sum of squares
 * is taken in double here (the reference's order is a 1024-thread tree in fp32:
 * layernorm_utils.cuh:26-47), so results agree to fp32 rounding only. */
static inline float inv_rms(double ss, int hidden, float eps) {
  return 1.0f / sqrtf((float)ss / (float)hidden + eps);
}
/* scalar_t(x * rms) * weight, product rounded in scalar_t
 * (layernorm_utils.cuh:114, layernorm_quant_kernels.cu:54) */
static inline float norm_elem(float x, float rms, float w, int dt) {
  return round_to(round_to(x * rms, dt) * w, dt);
}

/* csrc/layernorm_quant_kernels.cu:25-58 */
void oracle_rms_norm_static_fp8_quant(uint8_t* out, const void* in, int64_t in_stride,
                                      const void* weight, float scale, float eps,
                                      int64_t num_tokens, int hidden, int dt) {
  const float inv = 1.0f / scale;
  for (int64_t t = 0; t < num_tokens; ++t) {
    double ss = 0;
    for (int i = 0; i < hidden; ++i) {
      float x = load_as_f32(in, t * in_stride + i, dt);
      ss += (double)x * x;
    }
    const float r = inv_rms(ss, hidden, eps);
    for (int i = 0; i < hidden; ++i) {
      float x = load_as_f32(in, t * in_stride + i, dt);
      out[t * hidden + i] = quant_conv(norm_elem(x, r, load_as_f32(weight, i, dt), dt), inv, 1);
    }
  }
}

/* csrc/layernorm_quant_kernels.cu:64-168: z = input + residual rounded to scalar_t
 * (written back to residual), variance over z, then as above. */
void oracle_fused_add_rms_norm_static_fp8_quant(uint8_t* out, const void* in,
                                                int64_t in_stride, void* residual,
                                                const void* weight, float scale, float eps,
                                                int64_t num_tokens, int hidden, int dt) {
  const float inv = 1.0f / scale;
  for (int64_t t = 0; t < num_tokens; ++t) {
    double ss = 0;
    for (int i = 0; i < hidden; ++i) {
      float z = round_to(load_as_f32(in, t * in_stride + i, dt) +
                         load_as_f32(residual, t * hidden + i, dt), dt);
      store_from_f32(residual, t * hidden + i, dt, z);
      ss += (double)z * z;
    }
    const float r = inv_rms(ss, hidden, eps);
    for (int i = 0; i < hidden; ++i) {
      float z = load_as_f32(residual, t * hidden + i, dt);
      out[t * hidden + i] = quant_conv(norm_elem(z, r, load_as_f32(weight, i, dt), dt), inv, 1);
    }
  }
}

/* csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu:11-85 with
 * layernorm_utils.cuh:20-119: x = float(in) (+ float(residual), NOT rounded; residual is
 * overwritten with scalar_t(x)); token scale = max(min(absmax, ub) / 448, 1/(448*512));
 * fp8 output divides by the scale ("do not invert", :38-41). */
void oracle_rms_norm_dynamic_per_token_quant(uint8_t* out, const void* in, const void* weight,
                                             float* scales, float eps, const float* scale_ub,
                                             void* residual, int64_t num_tokens, int hidden,
                                             int dt) {
  const float min_scale = 1.0f / (FP8_MAX * 512.0f);
  float* xs = (float*)malloc(sizeof(float) * (size_t)(hidden > 0 ? hidden : 1));
  for (int64_t t = 0; t < num_tokens; ++t) {
    double ss = 0;
    for (int i = 0; i < hidden; ++i) {
      float x = load_as_f32(in, t * hidden + i, dt);
      if (residual) x += load_as_f32(residual, t * hidden + i, dt);
      xs[i] = x;
      ss += (double)x * x;
    }
    const float r = inv_rms(ss, hidden, eps);
    float amax = 0.f;
    for (int i = 0; i < hidden; ++i)
      amax = fmaxf(amax, fabsf(norm_elem(xs[i], r, load_as_f32(weight, i, dt), dt)));
    float s = scale_ub ? fminf(amax, *scale_ub) : amax;
    s = fmaxf(s / FP8_MAX, min_scale);
    scales[t] = s;
    for (int i = 0; i < hidden; ++i) {
      if (residual) store_from_f32(residual, t * hidden + i, dt, xs[i]);
      out[t * hidden + i] = quant_conv(norm_elem(xs[i], r, load_as_f32(weight, i, dt), dt), s, 0);
    }
  }
  free(xs);
}

/* csrc/quantization/activation_kernels.cu:14-89: out = fp8(T(silu(x)) * y in T, times 1/scale);
 * input [..., 2, d] (x then y), f16 / bf16 only (:117-118). */
void oracle_silu_and_mul_quant(uint8_t* out, const void* in, float scale, int64_t num_tokens,
                               int d, int dt) {
  const float inv = 1.0f / scale;
  for (int64_t t = 0; t < num_tokens; ++t)
    for (int i = 0; i < d; ++i) {
      float x = load_as_f32(in, t * 2 * d + i, dt);
      float y = load_as_f32(in, t * 2 * d + d + i, dt);
      float a = round_to(x / (1.0f + expf(-x)), dt);
      out[t * d + i] = quant_conv(round_to(a * y, dt), inv, 1);
    }
}

/* vllm/v1/worker/block_table.py:84-99 */
void oracle_compute_slot_mapping(int64_t* slot_mapping, const int32_t* block_table,
                                 const int32_t* req_indices, const int64_t* positions,
                                 int64_t num_tokens, int max_blocks_per_req, int block_size) {
  for (int64_t i = 0; i < num_tokens; ++i) {
    int64_t bi = (int64_t)req_indices[i] * max_blocks_per_req + positions[i] / block_size;
    slot_mapping[i] = (int64_t)block_table[bi] * block_size + positions[i] % block_size;
  }
}

/* ------------------------------------------------------------------ */
/* prefill attention with a paged prefix              */
/* ------------------------------------------------------------------ */
/* vllm/attention/ops/prefix_prefill.py:36-288 (_fwd_kernel) and :477-722
 * (_fwd_kernel_alibi), restated per (sequence, head, query row): scores in fp32 over the
 * cached context (no causal mask; fp8 context dequantised as (fp8 * scale).to(dtype),
 * :168-169, :212-213) and over the new tokens (causal), sliding window -> -10000 (:190-192),
 * ALiBi bias where <= 0 (:603-608), p rounded to the value dtype before P.V (:216),
 * out = acc / l (:276), optional fp8 output (:282-284).  The reference rounds p against the
 * RUNNING maximum of its tile order; here the final maximum is used (the difference is
 * below the rounding of p itself). */
void oracle_context_attention_fwd(
    void* o, const void* q, const void* k, const void* v, const void* k_cache,
    const void* v_cache, const int32_t* b_loc, const int32_t* b_start_loc,
    const int32_t* b_seq_len, float k_scale, float v_scale, const float* alibi_slopes,
    const float* fp8_out_scale, int batch, int num_heads, int num_kv_heads, int head_size,
    int block_size, int x, int64_t b_loc_stride, int64_t q_stride_t, int64_t q_stride_h,
    int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
    int64_t o_stride_t, int64_t o_stride_h, int64_t kc_block_stride, int64_t kc_head_stride,
    int64_t vc_block_stride, int64_t vc_head_stride, float sm_scale, int sliding_window,
    int skip_decode, int dt, int kv_dt) {
  const int D = head_size, BS = block_size;
  const int gq = num_heads / num_kv_heads;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < batch; ++b)
    for (int h = 0; h < num_heads; ++h) {
      const int tok0 = b_start_loc[b], qlen = b_start_loc[b + 1] - tok0;
      const int seq_len = b_seq_len[b], ctx = seq_len - qlen, kvh = h / gq;
      if (skip_decode && qlen == 1) continue;
      const int total = ctx + qlen;
      float* kf = (float*)malloc(sizeof(float) * (size_t)total * D);
      float* vf = (float*)malloc(sizeof(float) * (size_t)total * D);
      float* sc = (float*)malloc(sizeof(float) * (size_t)total);
      for (int n = 0; n < ctx; ++n) {
        const int64_t blk = b_loc[b * b_loc_stride + n / BS];
        const int off = n % BS;
        for (int d = 0; d < D; ++d) {
          const int64_t ki = blk * kc_block_stride + kvh * kc_head_stride +
                             (int64_t)(d / x) * BS * x + (int64_t)off * x + d % x;
          const int64_t vi = blk * vc_block_stride + kvh * vc_head_stride + (int64_t)d * BS + off;
          if (kv_dt == KV_FP8_E4M3) {
            kf[(size_t)n * D + d] = round_to(oracle_fp8_e4m3_to_f32(((const uint8_t*)k_cache)[ki]) * k_scale, dt);
            vf[(size_t)n * D + d] = round_to(oracle_fp8_e4m3_to_f32(((const uint8_t*)v_cache)[vi]) * v_scale, dt);
          } else {
            kf[(size_t)n * D + d] = load_as_f32(k_cache, ki, dt);
            vf[(size_t)n * D + d] = load_as_f32(v_cache, vi, dt);
          }
        }
      }
      for (int n = 0; n < qlen; ++n)
        for (int d = 0; d < D; ++d) {
          kf[(size_t)(ctx + n) * D + d] = load_as_f32(k, (tok0 + n) * k_stride_t + kvh * k_stride_h + d, dt);
          vf[(size_t)(ctx + n) * D + d] = load_as_f32(v, (tok0 + n) * v_stride_t + kvh * v_stride_h + d, dt);
        }
      const int window = alibi_slopes ? 0 : sliding_window;
      for (int m = 0; m < qlen; ++m) {
        const int qpos = ctx + m;
        float mx = -INFINITY;
        for (int n = 0; n <= qpos; ++n) {   /* context: all; new: causal */
          float acc = 0.f;
          for (int d = 0; d < D; ++d)
            acc += load_as_f32(q, (tok0 + m) * q_stride_t + h * q_stride_h + d, dt) * kf[(size_t)n * D + d];
          float s = acc * sm_scale;
          if (alibi_slopes) {
            float bias = (float)(n - qpos) * alibi_slopes[h];
            s = (bias <= 0.f && qpos < seq_len) ? s + bias : -INFINITY;
          } else if (window > 0 && qpos - n >= window) {
            s = -10000.0f;
          }
          sc[n] = s;
          mx = fmaxf(mx, s);
        }
        float l = 0.f;
        double acc[256];
        for (int d = 0; d < D; ++d) acc[d] = 0;
        for (int n = 0; n <= qpos; ++n) {
          float p = expf(sc[n] - mx);
          l += p;
          float pr = round_to(p, dt);
          for (int d = 0; d < D; ++d) acc[d] += (double)pr * vf[(size_t)n * D + d];
        }
        for (int d = 0; d < D; ++d) {
          float y = (float)acc[d] / l;
          if (fp8_out_scale) {
            float z = fmaxf(-FP8_MAX, fminf(y / *fp8_out_scale, FP8_MAX));
            ((uint8_t*)o)[(tok0 + m) * o_stride_t + h * o_stride_h + d] = oracle_f32_to_fp8_e4m3(z);
          } else {
            store_from_f32(o, (tok0 + m) * o_stride_t + h * o_stride_h + d, dt, y);
          }
        }
      }
      free(kf); free(vf); free(sc);
    }
}
