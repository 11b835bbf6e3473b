// The ops either side of the fp8 GEMM and of the cache write:
//   rms_norm_static_fp8_quant, fused_add_rms_norm_static_fp8_quant   (layernorm_quant_kernels.cu)
//   rms_norm_dynamic_per_token_quant                                 (fused_layernorm_dynamic_per_token_quant.cu)
//   silu_and_mul_quant                                               (activation_kernels.cu)
//   compute_slot_mapping                                             (v1/worker/block_table.py:84-99)
//   swap_blocks                                                      (cache_kernels.cu:25-67)
// All HBM-bound.  The row kernels take one workgroup per token and keep the row in
// registers between the reduction passes (the reference re-reads it from memory for the
// variance, for the absmax and for the conversion): one read and one write of the row.
#include "common.cuh"

namespace va {

constexpr float kQMax = 448.0f;
constexpr int ROW_THREADS = 256;

template <typename T> struct alignas(sizeof(T) * 8) Vec8 { T v[8]; };
// 16 bytes with a non-temporal load (16-bit T)
template <typename T>
__device__ __forceinline__ Vec8<T> ld_stream(const Vec8<T>* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  static_assert(sizeof(Vec8<T>) == 16, "");
  const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
  Vec8<T> r;
  __builtin_memcpy(&r, &raw, 16);
  return r;
}

// block-wide reductions over ROW_THREADS threads; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += red[i];
  return r;
}
__device__ __forceinline__ float block_max2(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
  return r;
}

// scalar_t(x * rms) * weight with the product rounded in scalar_t
// (layernorm_utils.cuh:114, layernorm_quant_kernels.cu:54); the fp32 product of two
// 16-bit floats is exact, so rounding it once equals the native 16-bit multiply.
template <typename T>
__device__ __forceinline__ float norm_elem(float x, float rms, T w) {
  return to_f32(from_f32<T>(to_f32(from_f32<T>(x * rms)) * to_f32(w)));
}
__device__ __forceinline__ float clamp_q(float x) { return fmaxf(-kQMax, fminf(x, kQMax)); }

enum { MODE_STATIC = 0, MODE_ADD_STATIC = 1, MODE_DYNAMIC = 2, MODE_QUANT = 3, MODE_DYNAMIC_I8 = 4 };

// x of one element under the three residual conventions; updates the residual
template <typename T, int MODE>
__device__ __forceinline__ float take_x(T in, T* res_slot, bool has_res) {
  float x = to_f32(in);
  if constexpr (MODE == MODE_ADD_STATIC) {
    // z = scalar_t(input + residual), residual = z (layernorm_quant_kernels.cu:95-98, 143-147)
    const T z = from_f32<T>(x + to_f32(*res_slot));
    *res_slot = z;
    x = to_f32(z);
  } else if constexpr (MODE == MODE_DYNAMIC) {
    // fp32 sum is used unrounded; residual = scalar_t(sum) (layernorm_utils.cuh:108-112)
    if (has_res) {
      x += to_f32(*res_slot);
      *res_slot = from_f32<T>(x);
    }
  }
  return x;
}

// Row in registers: hidden % 8 == 0, 16-byte aligned rows.  A row belongs to W waves (4 from
// 2048 elements up; shorter rows share a workgroup, and with W = 1 the reductions never leave
// the wave); a lane
// holds CH chunks of 8 elements and issues all its loads before the first use.
// MODE_QUANT is dynamic_per_token_scaled_fp8_quant (no norm, no weight).
// NWV = waves per workgroup (4; 8 for the norm modes on rows above 4096 elements, all on one row:
// two chunks per lane at 8192 like the 4096-wide rows, instead of four at lower occupancy)
template <typename T, int MODE, int W, int CH, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void row_quant_vec_kernel(
    uint8_t* __restrict__ out, const T* __restrict__ in, int64_t in_stride, T* __restrict__ residual,
    const T* __restrict__ weight, const float* __restrict__ scale_p, float* __restrict__ scales,
    const float* __restrict__ scale_ub, float eps, int hidden, int64_t num_tokens) {
  constexpr int RPB = NWV / W;   // rows per workgroup
  __shared__ float red[2][NWV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = wave % W;
  int64_t t = (int64_t)blockIdx.x * RPB + wave / W;
  const bool live = t < num_tokens;   // spare waves redo the last row and store nothing
  if (!live) t = num_tokens - 1;
  const int nchunk = hidden >> 3;
  const Vec8<T>* vin = reinterpret_cast<const Vec8<T>*>(in + t * in_stride);
  Vec8<T>* vres = residual ? reinterpret_cast<Vec8<T>*>(residual + t * hidden) : nullptr;
  const Vec8<T>* vw = reinterpret_cast<const Vec8<T>*>(weight);
  const bool has_res = residual != nullptr;
  auto row_sum = [&](float v, int k) {
    v = wave_sum(v);
    if constexpr (W > 1) {
      red[k][wave] = v;
      __syncthreads();
      v = 0.f;
#pragma unroll
      for (int i = 0; i < W; ++i) v += red[k][(wave / W) * W + i];
    }
    return v;
  };
  auto row_max = [&](float v, int k) {
    v = wave_max(v);
    if constexpr (W > 1) {
      red[k][wave] = v;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < W; ++i) v = fmaxf(v, red[k][(wave / W) * W + i]);
    }
    return v;
  };

  Vec8<T> a[CH], r[CH], w[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = (c * W + sub) * 64 + lane;
    if (i < nchunk) {
      a[c] = vin[i];
      if (MODE != MODE_STATIC && MODE != MODE_QUANT && has_res) r[c] = vres[i];
      if constexpr (MODE != MODE_QUANT) w[c] = vw[i];
    }
  }
  float x[CH][8];
  float ss = 0.f, amax = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = (c * W + sub) * 64 + lane;
    if (i < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == MODE_QUANT) {
          x[c][j] = to_f32(a[c].v[j]);
          amax = fmaxf(amax, fabsf(x[c][j]));
        } else {
          x[c][j] = take_x<T, MODE>(a[c].v[j], &r[c].v[j], has_res);
          ss += x[c][j] * x[c][j];
        }
      }
      if (MODE != MODE_STATIC && MODE != MODE_QUANT && has_res && live) vres[i] = r[c];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[c][j] = 0.f;
    }
  }
  if constexpr (MODE != MODE_QUANT) {
    ss = row_sum(ss, 0);
    const float rms = rsqrtf(ss / (float)hidden + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int i = (c * W + sub) * 64 + lane;
      if (i < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          x[c][j] = norm_elem<T>(x[c][j], rms, w[c].v[j]);
          amax = fmaxf(amax, fabsf(x[c][j]));
        }
      }
    }
  }
  float s, y = 0.f;
  if constexpr (MODE == MODE_DYNAMIC || MODE == MODE_QUANT) {
    amax = row_max(amax, 1);
    // layernorm_utils.cuh:79-90, fp8/common.cu:60-70
    s = scale_ub ? fminf(amax, *scale_ub) : amax;
    s = fmaxf(s / kQMax, 1.0f / (kQMax * 512.0f));
    y = 1.0f / s;
    if (lane == 0 && sub == 0 && live) scales[t] = s;
  } else {
    s = 1.0f / *scale_p;   // "invert scale to avoid division" (layernorm_quant_kernels.cu:50)
  }
  if (!live) return;
  uint2* vout = reinterpret_cast<uint2*>(out + t * hidden);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = (c * W + sub) * 64 + lane;
    if (i < nchunk) {
      float q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        q[j] = (MODE == MODE_DYNAMIC || MODE == MODE_QUANT) ? div_clamp448(x[c][j], s, y)
                                                            : clamp_q(x[c][j] * s);
      vout[i] = make_uint2(f32x4_to_fp8x4(q[0], q[1], q[2], q[3]),
                           f32x4_to_fp8x4(q[4], q[5], q[6], q[7]));
    }
  }
}

// Any hidden size / alignment: strided scalar passes like the reference's generic kernels.
template <typename T, int MODE>
__global__ __launch_bounds__(ROW_THREADS) void row_quant_generic_kernel(
    uint8_t* __restrict__ out, const T* __restrict__ in, int64_t in_stride, T* __restrict__ residual,
    const T* __restrict__ weight, const float* __restrict__ scale_p, float* __restrict__ scales,
    const float* __restrict__ scale_ub, float eps, int hidden) {
  __shared__ float red[8];
  const int64_t t = blockIdx.x;
  const T* tin = in + t * in_stride;
  T* tres = residual ? residual + t * hidden : nullptr;
  const bool has_res = residual != nullptr;
  // pass 1: variance.  MODE_ADD_STATIC rewrites the residual here; MODE_DYNAMIC must
  // keep reading the ORIGINAL residual in passes 2 and 3, so it only updates it last.
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden; i += ROW_THREADS) {
    float x;
    if constexpr (MODE == MODE_ADD_STATIC) x = take_x<T, MODE>(tin[i], &tres[i], true);
    else x = to_f32(tin[i]) + (((MODE == MODE_DYNAMIC || MODE == MODE_DYNAMIC_I8) && has_res) ? to_f32(tres[i]) : 0.f);
    ss += x * x;
  }
  ss = block_sum(ss, red);
  const float rms = rsqrtf(ss / (float)hidden + eps);
  auto xval = [&](int i) -> float {
    if constexpr (MODE == MODE_ADD_STATIC) return to_f32(tres[i]);
    else return to_f32(tin[i]) + (((MODE == MODE_DYNAMIC || MODE == MODE_DYNAMIC_I8) && has_res) ? to_f32(tres[i]) : 0.f);
  };
  float s;
  if constexpr (MODE == MODE_DYNAMIC_I8) {
    // int8 output (fused_layernorm_dynamic_per_token_quant.cu:33-36, layernorm_utils.cuh:51-94,
    // quant_conversions.cuh:15-31, 51-61): scale = max(amax / 127, FLT_EPSILON), no upper bound;
    // q = saturate(nearbyint(y * (1 / scale)))
    float amax = 0.f;
    for (int i = threadIdx.x; i < hidden; i += ROW_THREADS)
      amax = fmaxf(amax, fabsf(norm_elem<T>(xval(i), rms, weight[i])));
    amax = block_max2(amax, red);
    s = fmaxf(amax / 127.0f, 1.1920928955078125e-07f);
    if (threadIdx.x == 0) scales[t] = s;
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < hidden; i += ROW_THREADS) {
      const float x = xval(i);
      float q = nearbyintf(norm_elem<T>(x, rms, weight[i]) * inv);
      q = q < -128.f ? -128.f : (q > 127.f ? 127.f : q);
      out[t * hidden + i] = (uint8_t)(int8_t)q;
      if (has_res) tres[i] = from_f32<T>(x);
    }
    return;
  } else if constexpr (MODE == MODE_DYNAMIC) {
    float amax = 0.f;
    for (int i = threadIdx.x; i < hidden; i += ROW_THREADS)
      amax = fmaxf(amax, fabsf(norm_elem<T>(xval(i), rms, weight[i])));
    amax = block_max2(amax, red);
    s = scale_ub ? fminf(amax, *scale_ub) : amax;
    s = fmaxf(s / kQMax, 1.0f / (kQMax * 512.0f));
    if (threadIdx.x == 0) scales[t] = s;
  } else {
    s = 1.0f / *scale_p;
  }
  for (int i = threadIdx.x; i < hidden; i += ROW_THREADS) {
    const float x = xval(i);
    const float y = norm_elem<T>(x, rms, weight[i]);
    out[t * hidden + i] = f32_to_fp8(clamp_q(MODE == MODE_DYNAMIC ? y / s : y * s));
    if constexpr (MODE == MODE_DYNAMIC) {
      if (has_res) tres[i] = from_f32<T>(x);
    }
  }
}

// silu(x) * y -> fp8, 8 elements per thread (activation_kernels.cu:14-89)
template <typename T>
__device__ __forceinline__ float silu_mul(T x, T y) {
  const float xf = to_f32(x);
  // the hardware exp2 / rcp (each ~1 ulp): the result is rounded to T right away, and the
  // accurate expf + IEEE division made this HBM-bound op VALU-bound
  const T a = from_f32<T>(xf * __builtin_amdgcn_rcpf(1.0f + __expf(-xf)));
  return to_f32(from_f32<T>(to_f32(a) * to_f32(y)));
}

// VEC: 8 elements per thread, no index division: workgroup = (token(s), 256 chunks); rows
// shorter than 256 chunks share a workgroup (2^tpb_log2 tokens, power-of-two chunk pitch).
template <typename T, bool VEC>
__global__ __launch_bounds__(ROW_THREADS) void silu_mul_quant_kernel(
    uint8_t* __restrict__ out, const T* __restrict__ in, const float* __restrict__ scale_p,
    int64_t num_tokens, int d, int tpb_log2, int stream_loads) {
  const float inv = 1.0f / *scale_p;
  if constexpr (VEC) {
    const int cpt = d >> 3;   // chunks per token
    const int pitch_log2 = 8 - tpb_log2;
    const int64_t t = ((int64_t)blockIdx.x << tpb_log2) + (threadIdx.x >> pitch_log2);
    const int c = blockIdx.y * ROW_THREADS + (threadIdx.x & ((1 << pitch_log2) - 1));
    if (t >= num_tokens || c >= cpt) return;
    // an input beyond the 256 MB Infinity Cache is streamed with non-temporal loads (8192 x 14336:
    // 116.9 -> 100.2 us); a smaller one is most likely still cached from the kernel that wrote it,
    // and plain loads are faster then (4096 x 14336: 42.4 vs 50.6 us; profiles/r02_row_quant_persistent.txt)
    const Vec8<T>* px = reinterpret_cast<const Vec8<T>*>(in + t * 2 * d) + c;
    const Vec8<T>* py = reinterpret_cast<const Vec8<T>*>(in + t * 2 * d + d) + c;
    const Vec8<T> x = stream_loads ? ld_stream(px) : *px;
    const Vec8<T> y = stream_loads ? ld_stream(py) : *py;
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = clamp_q(silu_mul<T>(x.v[j], y.v[j]) * inv);
    reinterpret_cast<uint2*>(out + t * d)[c] =
        make_uint2(f32x4_to_fp8x4(q[0], q[1], q[2], q[3]), f32x4_to_fp8x4(q[4], q[5], q[6], q[7]));
  } else {
    const int64_t total = num_tokens * d;
    for (int64_t idx = (int64_t)blockIdx.x * ROW_THREADS + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * ROW_THREADS) {
      const int64_t t = idx / d;
      const int i = (int)(idx - t * d);
      out[idx] = f32_to_fp8(clamp_q(silu_mul<T>(in[t * 2 * d + i], in[t * 2 * d + d + i]) * inv));
    }
  }
}

// slot = block_table[req, pos / BS] * BS + pos % BS   (block_table.py:92-99)
__global__ void slot_mapping_kernel(int64_t* __restrict__ slot_mapping,
                                    const int32_t* __restrict__ block_table,
                                    const int32_t* __restrict__ req_indices,
                                    const int64_t* __restrict__ positions, int64_t num_tokens,
                                    int max_blocks, int block_size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_tokens) return;
  const int64_t pos = positions[i];
  const int64_t blk = block_table[(int64_t)req_indices[i] * max_blocks + pos / block_size];
  slot_mapping[i] = blk * block_size + pos % block_size;
}

}  // namespace va

using namespace va;

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename T, int MODE, int W, int CH, int NWV = 4>
static void launch_row_vec(void* out, const void* in, int64_t in_stride, void* residual,
                           const void* weight, const float* scale, float* scales,
                           const float* scale_ub, float eps, int64_t num_tokens, int hidden,
                           hipStream_t st) {
  constexpr int RPB = NWV / W;
  hipLaunchKernelGGL((row_quant_vec_kernel<T, MODE, W, CH, NWV>),
                     dim3((unsigned)((num_tokens + RPB - 1) / RPB)), dim3(64 * NWV), 0, st,
                     (uint8_t*)out, (const T*)in, in_stride, (T*)residual, (const T*)weight, scale,
                     scales, scale_ub, eps, hidden, num_tokens);
}

// rows of up to 16384 elements stay in registers; false = not handled (caller's fallback)
template <typename T, int MODE>
static bool launch_row_regs(void* out, const void* in, int64_t in_stride, void* residual,
                            const void* weight, const float* scale, float* scales,
                            const float* scale_ub, float eps, int64_t num_tokens, int hidden,
                            hipStream_t st) {
  const bool vec = hidden % 8 == 0 && hidden <= 16384 && (in_stride * sizeof(T)) % 16 == 0 &&
                   aligned16(in) && (weight == nullptr || aligned16(weight)) &&
                   (residual == nullptr || aligned16(residual)) && (((uintptr_t)out) & 7) == 0;
  if (!vec) return false;
#define LR(W, CH) launch_row_vec<T, MODE, W, CH>(out, in, in_stride, residual, weight, scale, scales, \
                                                 scale_ub, eps, num_tokens, hidden, st)
#define LR8(W, CH) launch_row_vec<T, MODE, W, CH, 8>(out, in, in_stride, residual, weight, scale, scales, \
                                                     scale_ub, eps, num_tokens, hidden, st)
  // at most 4 chunks per lane up to 8192 (8 chunks cost 230 VGPRs in the dynamic mode)
  // as few chunks per lane as the row allows (registers -> occupancy: 4096-wide rows measured
  // 1 wave x 8 chunks 19.4 us, 2 x 4 16.5, 4 x 2 15.9 for the dynamic mode, M = 4096)
  if (hidden <= 512) LR(1, 1);
  else if (hidden <= 1024) LR(2, 1);
  else if (hidden <= 2048) LR(4, 1);
  else if (hidden <= 4096) LR(4, 2);
  // wider rows: 8 waves per row for the norm modes (two chunks per lane at 8192, like the 4096-wide
  // rows: 4096 x 8192 dynamic 37.6 -> 27.0 us, + residual 52.9 -> 39.2, static + add 41.2 -> 35.8;
  // 14336: 98.6 -> 45.5, 176 -> 80.4, 78.8 -> 65.7 us); plain per-token quantisation keeps 4 waves
  // (17.7 vs 19.4 us at 8192, 29.1 vs 31.3 at 14336)
  else if (hidden <= 8192) { if (MODE == MODE_QUANT) LR(4, 4); else LR8(8, 2); }
  else { if (MODE == MODE_QUANT) LR(4, 8); else LR8(8, 4); }
#undef LR
#undef LR8
  return true;
}

template <typename T, int MODE>
static void launch_row(void* out, const void* in, int64_t in_stride, void* residual,
                       const void* weight, const float* scale, float* scales,
                       const float* scale_ub, float eps, int64_t num_tokens, int hidden,
                       hipStream_t st) {
  bool done = false;
  if constexpr (MODE != MODE_DYNAMIC_I8)   // int8 output: the strided kernel only (outside the fp8 path)
    done = launch_row_regs<T, MODE>(out, in, in_stride, residual, weight, scale, scales, scale_ub, eps,
                                    num_tokens, hidden, st);
  if (!done)
    hipLaunchKernelGGL((row_quant_generic_kernel<T, MODE>), dim3((unsigned)num_tokens),
                       dim3(ROW_THREADS), 0, st, (uint8_t*)out, (const T*)in, in_stride,
                       (T*)residual, (const T*)weight, scale, scales, scale_ub, eps, hidden);
}

namespace va {
// dynamic_per_token_scaled_fp8_quant with the row in registers (fp8_quant.hip calls this
// first and falls back to its strided kernel when it returns false)
bool per_token_quant_rows(void* out, const void* in, float* scales, const float* scale_ub,
                          int64_t num_tokens, int hidden, int dtype, hipStream_t st) {
  switch (dtype) {
    case VLLM_AMD_F32:
      return launch_row_regs<float, MODE_QUANT>(out, in, hidden, nullptr, nullptr, nullptr, scales,
                                                scale_ub, 0.f, num_tokens, hidden, st);
    case VLLM_AMD_F16:
      return launch_row_regs<f16_t, MODE_QUANT>(out, in, hidden, nullptr, nullptr, nullptr, scales,
                                                scale_ub, 0.f, num_tokens, hidden, st);
    case VLLM_AMD_BF16:
      return launch_row_regs<bf16_t, MODE_QUANT>(out, in, hidden, nullptr, nullptr, nullptr, scales,
                                                 scale_ub, 0.f, num_tokens, hidden, st);
    default: return false;
  }
}
}  // namespace va

template <int MODE>
static int row_entry(const char* op, void* out, const void* in, int64_t in_stride, void* residual,
                     const void* weight, const float* scale, float* scales,
                     const float* scale_ub, float eps, int64_t num_tokens, int hidden, int dtype,
                     hipStream_t st) {
  VA_REQUIRE(num_tokens >= 0 && hidden >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (num_tokens == 0 || hidden == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && in && weight, VLLM_AMD_EINVAL, "%s: null pointer", op);
  VA_REQUIRE(num_tokens <= 0x7fffffff, VLLM_AMD_EINVAL, "%s: too many tokens", op);
  VA_REQUIRE(in_stride >= hidden, VLLM_AMD_EINVAL, "%s: input stride < hidden_size", op);
  switch (dtype) {
    case VLLM_AMD_F32:
      launch_row<float, MODE>(out, in, in_stride, residual, weight, scale, scales, scale_ub, eps, num_tokens, hidden, st);
      break;
    case VLLM_AMD_F16:
      launch_row<f16_t, MODE>(out, in, in_stride, residual, weight, scale, scales, scale_ub, eps, num_tokens, hidden, st);
      break;
    case VLLM_AMD_BF16:
      launch_row<bf16_t, MODE>(out, in, in_stride, residual, weight, scale, scales, scale_ub, eps, num_tokens, hidden, st);
      break;
    default:
      VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "%s: unsupported input dtype %d", op, dtype);
  }
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_rms_norm_static_fp8_quant(void* out, const void* input, const void* weight,
                                                  const float* scale, float epsilon,
                                                  int64_t num_tokens, int hidden_size,
                                                  int64_t input_stride, int dtype, void* stream) {
  VA_REQUIRE(scale != nullptr || num_tokens == 0, VLLM_AMD_EINVAL,
             "rms_norm_static_fp8_quant: null scale");
  return row_entry<MODE_STATIC>("rms_norm_static_fp8_quant", out, input, input_stride, nullptr,
                                weight, scale, nullptr, nullptr, epsilon, num_tokens, hidden_size,
                                dtype, (hipStream_t)stream);
}

extern "C" int vllm_amd_fused_add_rms_norm_static_fp8_quant(
    void* out, const void* input, void* residual, const void* weight, const float* scale,
    float epsilon, int64_t num_tokens, int hidden_size, int64_t input_stride, int dtype,
    void* stream) {
  VA_REQUIRE((scale != nullptr && residual != nullptr) || num_tokens == 0, VLLM_AMD_EINVAL,
             "fused_add_rms_norm_static_fp8_quant: null scale or residual");
  return row_entry<MODE_ADD_STATIC>("fused_add_rms_norm_static_fp8_quant", out, input,
                                    input_stride, residual, weight, scale, nullptr, nullptr,
                                    epsilon, num_tokens, hidden_size, dtype, (hipStream_t)stream);
}

extern "C" int vllm_amd_rms_norm_dynamic_per_token_quant(void* out, const void* input,
                                                         const void* weight, float* scales,
                                                         float epsilon, const float* scale_ub,
                                                         void* residual, int64_t num_tokens,
                                                         int hidden_size, int dtype, void* stream) {
  VA_REQUIRE(scales != nullptr || num_tokens == 0, VLLM_AMD_EINVAL,
             "rms_norm_dynamic_per_token_quant: null scales");
  return row_entry<MODE_DYNAMIC>("rms_norm_dynamic_per_token_quant", out, input, hidden_size,
                                 residual, weight, nullptr, scales, scale_ub, epsilon, num_tokens,
                                 hidden_size, dtype, (hipStream_t)stream);
}

// int8 output of the same op (scale_ub is fp8-only in the reference: TORCH_CHECK at :142-144)
extern "C" int vllm_amd_rms_norm_dynamic_per_token_quant_int8(void* out, const void* input,
                                                              const void* weight, float* scales,
                                                              float epsilon, void* residual,
                                                              int64_t num_tokens, int hidden_size,
                                                              int dtype, void* stream) {
  VA_REQUIRE(scales != nullptr || num_tokens == 0, VLLM_AMD_EINVAL,
             "rms_norm_dynamic_per_token_quant: null scales");
  return row_entry<MODE_DYNAMIC_I8>("rms_norm_dynamic_per_token_quant", out, input, hidden_size,
                                    residual, weight, nullptr, scales, nullptr, epsilon, num_tokens,
                                    hidden_size, dtype, (hipStream_t)stream);
}

extern "C" int vllm_amd_silu_and_mul_quant(void* out, const void* input, const float* scale,
                                           int64_t num_tokens, int d, int dtype, void* stream) {
  const char* op = "silu_and_mul_quant";
  VA_REQUIRE(num_tokens >= 0 && d >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (num_tokens == 0 || d == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && input && scale, VLLM_AMD_EINVAL, "%s: null pointer", op);
  VA_REQUIRE(dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16, VLLM_AMD_EUNSUPPORTED,
             "%s: input must be float16 or bfloat16", op);
  const bool vec = d % 8 == 0 && aligned16(input) && (((uintptr_t)out) & 7) == 0;
  dim3 grid;
  int tpb_log2 = 0;
  if (vec) {
    const int cpt = d >> 3;
    while (tpb_log2 < 8 && (cpt << (tpb_log2 + 1)) <= ROW_THREADS) ++tpb_log2;   // tokens per workgroup
    grid = dim3((unsigned)((num_tokens + (1 << tpb_log2) - 1) >> tpb_log2),
                (unsigned)((cpt + ROW_THREADS - 1) / ROW_THREADS));
  } else {
    int64_t blocks = (num_tokens * d + ROW_THREADS - 1) / ROW_THREADS;
    if (blocks > 256 * 16) blocks = 256 * 16;
    grid = dim3((unsigned)blocks);
  }
  hipStream_t st = (hipStream_t)stream;
  const int stream_loads = num_tokens * 2 * (int64_t)d * 2 > (256ll << 20) ? 1 : 0;
#define SM(T, V)                                                                          \
  hipLaunchKernelGGL((silu_mul_quant_kernel<T, V>), grid, dim3(ROW_THREADS),               \
                     0, st, (uint8_t*)out, (const T*)input, scale, num_tokens, d, tpb_log2, stream_loads)
  if (dtype == VLLM_AMD_F16) { if (vec) SM(f16_t, true); else SM(f16_t, false); }
  else { if (vec) SM(bf16_t, true); else SM(bf16_t, false); }
#undef SM
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_compute_slot_mapping(int64_t* slot_mapping, const int32_t* block_table,
                                             const int32_t* req_indices, const int64_t* positions,
                                             int64_t num_tokens, int max_num_blocks_per_req,
                                             int block_size, void* stream) {
  const char* op = "compute_slot_mapping";
  VA_REQUIRE(num_tokens >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (num_tokens == 0) return VLLM_AMD_OK;
  VA_REQUIRE(slot_mapping && block_table && req_indices && positions, VLLM_AMD_EINVAL,
             "%s: null pointer", op);
  VA_REQUIRE(block_size > 0 && max_num_blocks_per_req > 0, VLLM_AMD_EINVAL,
             "%s: block_size and max_num_blocks_per_req must be positive", op);
  hipLaunchKernelGGL(slot_mapping_kernel, dim3((unsigned)((num_tokens + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, slot_mapping, block_table, req_indices, positions,
                     num_tokens, max_num_blocks_per_req, block_size);
  VA_LAUNCH_CHECK(op);
  return VLLM_AMD_OK;
}

// One asynchronous copy per (src, dst) pair on the caller's stream; either side may be
// host memory (hipMemcpyDefault resolves the direction), as in cache_kernels.cu:25-67.
extern "C" int vllm_amd_swap_blocks(const void* src, void* dst, int64_t block_bytes,
                                    const int64_t* block_mapping_host, int64_t num_pairs,
                                    void* stream) {
  const char* op = "swap_blocks";
  VA_REQUIRE(num_pairs >= 0 && block_bytes >= 0, VLLM_AMD_EINVAL, "%s: negative size", op);
  if (num_pairs == 0 || block_bytes == 0) return VLLM_AMD_OK;
  VA_REQUIRE(src && dst && block_mapping_host, VLLM_AMD_EINVAL, "%s: null pointer", op);
  for (int64_t i = 0; i < num_pairs; ++i) {
    const int64_t s = block_mapping_host[2 * i], d = block_mapping_host[2 * i + 1];
    VA_REQUIRE(s >= 0 && d >= 0, VLLM_AMD_EINVAL, "%s: negative block number", op);
    hipError_t e = hipMemcpyAsync((char*)dst + d * block_bytes, (const char*)src + s * block_bytes,
                                  (size_t)block_bytes, hipMemcpyDefault, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, op);
  }
  return VLLM_AMD_OK;
}
