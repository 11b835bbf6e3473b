// 16-bit MFMA helpers shared by the attention kernels (decode: paged_attention_mfma.hip,
// prefill: prefill_attention.hip): mfma_f32_16x16x32_{bf16,f16} wrappers, exact
// fp8 e4m3 -> 16-bit widening, tail masks.
#pragma once
#include "common.cuh"

namespace va {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <typename scalar_t> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // two e4m3 bytes (half `hi` of w) -> two packed bf16, exact
  static __device__ __forceinline__ uint32_t cvt2(uint32_t w, bool hi) {
    bf16x2 r = hi ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true)
                  : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, r);
  }
};
template <> struct Mma<f16_t> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t cvt2(uint32_t w, bool hi) {
    f16x2 r = hi ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true)
                 : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f16x2 r = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, r);
  }
};

// 16 e4m3 bytes -> 2 fragments of 8 16-bit values (bytes 0..7, bytes 8..15)
template <typename scalar_t>
__device__ __forceinline__ void widen16(const uint4& raw, uint4& lo, uint4& hi) {
  lo.x = Mma<scalar_t>::cvt2(raw.x, false);
  lo.y = Mma<scalar_t>::cvt2(raw.x, true);
  lo.z = Mma<scalar_t>::cvt2(raw.y, false);
  lo.w = Mma<scalar_t>::cvt2(raw.y, true);
  hi.x = Mma<scalar_t>::cvt2(raw.z, false);
  hi.y = Mma<scalar_t>::cvt2(raw.z, true);
  hi.z = Mma<scalar_t>::cvt2(raw.w, false);
  hi.w = Mma<scalar_t>::cvt2(raw.w, true);
}

// keep the first `valid` (0..16) bytes of a 16-byte vector, zero the rest
__device__ __forceinline__ uint4 mask_bytes(uint4 v, int valid) {
  auto m = [&](int w) -> uint32_t {
    int r = valid - 4 * w;
    return r >= 4 ? 0xffffffffu : (r <= 0 ? 0u : ((1u << (8 * r)) - 1u));
  };
  v.x &= m(0); v.y &= m(1); v.z &= m(2); v.w &= m(3);
  return v;
}
// keep the first `valid` (0..8) 16-bit elements
__device__ __forceinline__ uint4 mask_halfs(uint4 v, int valid) {
  auto m = [&](int w) -> uint32_t {
    int r = valid - 2 * w;
    return r >= 2 ? 0xffffffffu : (r <= 0 ? 0u : 0x0000ffffu);
  };
  v.x &= m(0); v.y &= m(1); v.z &= m(2); v.w &= m(3);
  return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

}  // namespace va
