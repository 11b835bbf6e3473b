// fp8 (OCP e4m3fn) quantisation of activations / weights:
//   static per-tensor, dynamic per-tensor, dynamic per-token.
// HBM-bound: 16 elements (one 16-byte output vector) per thread per step.
// Arithmetic follows csrc/quantization/fp8/common.cu:14-83 and common.cuh:38-97
// of the reference exactly (multiply by 1/scale for per-tensor, DIVIDE by the
// scale for per-token, clamp to +-448, RNE convert), so results are bit-exact.
#include "common.cuh"

namespace va {

constexpr float kFp8Max = 448.0f;
constexpr int kVec = 16;

template <typename T> struct alignas(sizeof(T) * 16) In16 { T v[16]; };
struct alignas(16) Out16 { uint32_t w[4]; };

template <bool INVERTED>
__device__ __forceinline__ float qscale(float v, float s) {
  // fp8::scaled_fp8_conversion, common.cuh:38-56
  float x = INVERTED ? v * s : v / s;
  return fmaxf(-kFp8Max, fminf(x, kFp8Max));
}

template <typename scalar_t, bool INVERTED>
__device__ __forceinline__ Out16 quant16(const In16<scalar_t>& in, float s) {
  Out16 o;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o.w[j] = f32x4_to_fp8x4(qscale<INVERTED>(to_f32(in.v[4 * j + 0]), s),
                            qscale<INVERTED>(to_f32(in.v[4 * j + 1]), s),
                            qscale<INVERTED>(to_f32(in.v[4 * j + 2]), s),
                            qscale<INVERTED>(to_f32(in.v[4 * j + 3]), s));
  return o;
}

template <typename scalar_t>
__device__ __forceinline__ float absmax16(const In16<scalar_t>& in, float m) {
#pragma unroll
  for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(to_f32(in.v[j])));
  return m;
}

// block-wide max; result valid in every thread.  red: >= 16 floats of LDS.
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

// ---- static per-tensor ------------------------------------------------------
template <typename scalar_t>
__global__ void static_quant_kernel(uint8_t* __restrict__ out, const scalar_t* __restrict__ in,
                                    const float* __restrict__ scale, int64_t n, bool vec_ok) {
  const float inv = 1.0f / (*scale);  // common.cu:21-25
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t done = 0;
  if (vec_ok) {
    const int64_t nv = n >> 4;
    const In16<scalar_t>* vin = reinterpret_cast<const In16<scalar_t>*>(in);
    Out16* vout = reinterpret_cast<Out16*>(out);
    for (int64_t i = tid; i < nv; i += step) vout[i] = quant16<scalar_t, true>(vin[i], inv);
    done = nv << 4;
  }
  for (int64_t i = done + tid; i < n; i += step)
    out[i] = f32_to_fp8(qscale<true>(to_f32(in[i]), inv));
}

// ---- dynamic per-tensor: pass 1, global absmax / 448 -------------------------
template <typename scalar_t>
__global__ void absmax_kernel(float* __restrict__ scale, const scalar_t* __restrict__ in,
                              int64_t n, bool vec_ok) {
  __shared__ float red[16];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  float m = 0.f;
  int64_t done = 0;
  if (vec_ok) {
    const int64_t nv = n >> 4;
    const In16<scalar_t>* vin = reinterpret_cast<const In16<scalar_t>*>(in);
    for (int64_t i = tid; i < nv; i += step) m = absmax16<scalar_t>(vin[i], m);
    done = nv << 4;
  }
  for (int64_t i = done + tid; i < n; i += step) m = fmaxf(m, fabsf(to_f32(in[i])));
  m = block_max(m, red);
  if (threadIdx.x == 0) {
    // common.cuh:92-96: atomic max of blockmax/448; non-negative floats order
    // like their bit patterns, *scale starts at <= 0.
    float s = m / kFp8Max;
    atomicMax(reinterpret_cast<int*>(scale), __float_as_int(s));
  }
}

// ---- dynamic per-token ---------------------------------------------------------
template <typename scalar_t>
__global__ void per_token_quant_kernel(uint8_t* __restrict__ out, float* __restrict__ scales,
                                       const scalar_t* __restrict__ in,
                                       const float* __restrict__ scale_ub, int hidden,
                                       bool vec_ok) {
  __shared__ float red[16];
  const int64_t off = (int64_t)blockIdx.x * hidden;
  const scalar_t* tin = in + off;
  uint8_t* tout = out + off;
  const int tid = threadIdx.x, step = blockDim.x;
  float m = 0.f;
  const int nv = vec_ok ? (hidden >> 4) : 0;
  const In16<scalar_t>* vin = reinterpret_cast<const In16<scalar_t>*>(tin);
  for (int i = tid; i < nv; i += step) m = absmax16<scalar_t>(vin[i], m);
  for (int i = (nv << 4) + tid; i < hidden; i += step) m = fmaxf(m, fabsf(to_f32(tin[i])));
  m = block_max(m, red);
  // common.cu:60-70
  float s = scale_ub ? fminf(m, *scale_ub) : m;
  s = fmaxf(s / kFp8Max, 1.0f / (kFp8Max * 512.0f));
  if (tid == 0) scales[blockIdx.x] = s;
  Out16* vout = reinterpret_cast<Out16*>(tout);
  for (int i = tid; i < nv; i += step) vout[i] = quant16<scalar_t, false>(vin[i], s);
  for (int i = (nv << 4) + tid; i < hidden; i += step)
    tout[i] = f32_to_fp8(qscale<false>(to_f32(tin[i]), s));
}

// fused_quant.hip: whole rows in registers (hidden % 8 == 0, <= 16384, aligned)
bool per_token_quant_rows(void* out, const void* in, float* scales, const float* scale_ub,
                          int64_t num_tokens, int hidden, int dtype, hipStream_t st);

}  // namespace va

using namespace va;

static inline bool aligned(const void* p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

static int grid_for(int64_t n) {
  int64_t b = (n / kVec + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;
  return (int)b;
}

template <typename T>
static void launch_static(void* out, const void* in, const float* scale, int64_t n, hipStream_t st) {
  bool vec_ok = aligned(in, sizeof(T) * 16) && aligned(out, 16);
  hipLaunchKernelGGL((static_quant_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st,
                     (uint8_t*)out, (const T*)in, scale, n, vec_ok);
}

extern "C" int vllm_amd_static_scaled_fp8_quant(void* out, const void* input, const float* scale,
                                                int64_t num_tokens, int hidden_size, int dtype,
                                                void* stream) {
  VA_REQUIRE(num_tokens >= 0 && hidden_size >= 0, VLLM_AMD_EINVAL, "static_scaled_fp8_quant: bad shape");
  int64_t n = num_tokens * hidden_size;
  if (n == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && input && scale, VLLM_AMD_EINVAL, "static_scaled_fp8_quant: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLLM_AMD_F32) launch_static<float>(out, input, scale, n, st);
  else if (dtype == VLLM_AMD_F16) launch_static<f16_t>(out, input, scale, n, st);
  else if (dtype == VLLM_AMD_BF16) launch_static<bf16_t>(out, input, scale, n, st);
  else VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "static_scaled_fp8_quant: unsupported dtype %d", dtype);
  VA_LAUNCH_CHECK("static_scaled_fp8_quant launch");
  return VLLM_AMD_OK;
}

template <typename T>
static void launch_dynamic(void* out, const void* in, float* scale, int64_t n, hipStream_t st) {
  bool vec_ok = aligned(in, sizeof(T) * 16) && aligned(out, 16);
  hipLaunchKernelGGL((absmax_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, scale,
                     (const T*)in, n, vec_ok);
  hipLaunchKernelGGL((static_quant_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st,
                     (uint8_t*)out, (const T*)in, (const float*)scale, n, vec_ok);
}

extern "C" int vllm_amd_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale,
                                                 int64_t num_tokens, int hidden_size, int dtype,
                                                 void* stream) {
  VA_REQUIRE(num_tokens >= 0 && hidden_size >= 0, VLLM_AMD_EINVAL, "dynamic_scaled_fp8_quant: bad shape");
  int64_t n = num_tokens * hidden_size;
  if (n == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && input && scale, VLLM_AMD_EINVAL, "dynamic_scaled_fp8_quant: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VLLM_AMD_F32) launch_dynamic<float>(out, input, scale, n, st);
  else if (dtype == VLLM_AMD_F16) launch_dynamic<f16_t>(out, input, scale, n, st);
  else if (dtype == VLLM_AMD_BF16) launch_dynamic<bf16_t>(out, input, scale, n, st);
  else VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "dynamic_scaled_fp8_quant: unsupported dtype %d", dtype);
  VA_LAUNCH_CHECK("dynamic_scaled_fp8_quant launch");
  return VLLM_AMD_OK;
}

template <typename T>
static void launch_per_token(void* out, const void* in, float* scales, const float* ub,
                             int64_t num_tokens, int hidden, hipStream_t st) {
  bool vec_ok = (hidden % 16 == 0) && aligned(in, sizeof(T) * 16) && aligned(out, 16);
  int threads = hidden / 16;
  if (!vec_ok) threads = hidden;
  threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : ((threads + 63) / 64) * 64);
  if (threads > 512) threads = 512;
  hipLaunchKernelGGL((per_token_quant_kernel<T>), dim3((unsigned)num_tokens), dim3(threads), 0, st,
                     (uint8_t*)out, scales, (const T*)in, ub, hidden, vec_ok);
}

extern "C" int vllm_amd_dynamic_per_token_scaled_fp8_quant(void* out, const void* input,
                                                           float* scales, const float* scale_ub,
                                                           int64_t num_tokens, int hidden_size,
                                                           int dtype, void* stream) {
  VA_REQUIRE(num_tokens >= 0 && hidden_size >= 0, VLLM_AMD_EINVAL,
             "dynamic_per_token_scaled_fp8_quant: bad shape");
  if (num_tokens == 0) return VLLM_AMD_OK;
  VA_REQUIRE(out && input && scales, VLLM_AMD_EINVAL,
             "dynamic_per_token_scaled_fp8_quant: null pointer");
  VA_REQUIRE(num_tokens < (1ll << 31), VLLM_AMD_EINVAL, "too many tokens");
  hipStream_t st = (hipStream_t)stream;
  if (hidden_size == 0) return VLLM_AMD_OK;
  if (per_token_quant_rows(out, input, scales, scale_ub, num_tokens, hidden_size, dtype, st)) {
    VA_LAUNCH_CHECK("dynamic_per_token_scaled_fp8_quant launch");
    return VLLM_AMD_OK;
  }
  if (dtype == VLLM_AMD_F32) launch_per_token<float>(out, input, scales, scale_ub, num_tokens, hidden_size, st);
  else if (dtype == VLLM_AMD_F16) launch_per_token<f16_t>(out, input, scales, scale_ub, num_tokens, hidden_size, st);
  else if (dtype == VLLM_AMD_BF16) launch_per_token<bf16_t>(out, input, scales, scale_ub, num_tokens, hidden_size, st);
  else VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "dynamic_per_token_scaled_fp8_quant: unsupported dtype %d", dtype);
  VA_LAUNCH_CHECK("dynamic_per_token_scaled_fp8_quant launch");
  return VLLM_AMD_OK;
}
