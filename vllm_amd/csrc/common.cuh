// Shared device/host helpers for libvllm_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/vllm_amd.h"

namespace va {

// ---- host side: error reporting ------------------------------------------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char* what);

#define VA_REQUIRE(cond, code, ...)   \
  do {                                \
    if (!(cond)) {                    \
      ::va::set_error(__VA_ARGS__);   \
      return (code);                  \
    }                                 \
  } while (0)

#define VA_LAUNCH_CHECK(what)                              \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return ::va::hip_fail(e__, what); \
  } while (0)

// ---- host side: the caller-registered workspace (vllm_amd_set_workspace) -----------------
// Nothing is allocated inside an op call.  Kernels that need device scratch beyond their
// arguments (split-K partials of scaled_mm, partials of the balanced decode attention) take it
// from CALLER-OWNED memory registered per device; without one they run their scratch-free form.
// The workspace is cut into WS_SLOTS slots; a slot belongs to ONE stream (the first that asks,
// until the workspace is replaced): launches of one stream are ordered and may share a slot,
// streams never share one.  No slot for: a stream beyond WS_SLOTS, a launch under stream capture
// (the graph could replay on any stream, beside eager launches).  Ticket counters sit in their
// own area, zeroed at registration, and every kernel leaves its counters at zero.
constexpr int WS_SLOTS = 4;
constexpr size_t WS_CTR_BYTES = (size_t)128 << 10;        // per slot
constexpr size_t WS_CTR_MM_WORDS = 4096;                  // [0, 16 KiB): scaled_mm split-K tickets
constexpr size_t WS_CTR_PA_WORDS = 28672;                 // [16 KiB, 128 KiB): decode attention tickets
struct WsSlot { unsigned* ctr_mm; unsigned* ctr_pa; char* data; size_t data_bytes; };
bool workspace_slot(hipStream_t st, WsSlot* out);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int dtype_size(int dt) { return dt == VLLM_AMD_F32 ? 4 : 2; }

// ---- device side: scalar types --------------------------------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;

template <int DT> struct Scalar;
template <> struct Scalar<VLLM_AMD_F32> { typedef float type; };
template <> struct Scalar<VLLM_AMD_F16> { typedef f16_t type; };
template <> struct Scalar<VLLM_AMD_BF16> { typedef bf16_t type; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
// float -> T, round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// ---- device side: OCP fp8 e4m3fn ------------------------------------------
__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
  return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
// bytes [2*hi, 2*hi+1] of w -> two floats
__device__ __forceinline__ f32x2 fp8x2_to_f32(uint32_t w, bool hi) {
  return hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true)
            : __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
}
// Round-to-nearest-even with finite saturation: the arithmetic of
// __hip_cvt_float_to_fp8(v, __HIP_SATFINITE, __HIP_E4M3) on gfx950 (NaN/Inf
// skip the clamp), which is what the reference's fp8::scaled_convert and
// fp8::cvt_c10 reduce to (csrc/quantization/fp8/amd/quant_utils.cuh:29-40).
__device__ __forceinline__ float fp8_clamp(float v) {
  uint32_t u = __float_as_uint(v);
  return ((u & 0x7f800000u) != 0x7f800000u)
             ? __builtin_amdgcn_fmed3f(v, 448.0f, -448.0f)
             : v;
}
__device__ __forceinline__ uint8_t f32_to_fp8(float v) {
  v = fp8_clamp(v);
  return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff);
}
// four floats -> four packed e4m3 bytes (a -> byte 0)
__device__ __forceinline__ uint32_t f32x4_to_fp8x4(float a, float b, float c, float d) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(a), fp8_clamp(b), 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(c), fp8_clamp(d), r, true);
  return (uint32_t)r;
}

// fp8::scaled_convert<uint8_t, scalar_t> (quant_utils.cuh:486-495, 540-547, 585-591):
// the half path rounds the quotient to half before the fp8 conversion.
// The division is IEEE x / scale, computed with Markstein's FMA sequence from the correctly
// rounded reciprocal y (one real division per thread, then 7 VALU ops per element instead of
// ~11: q0 = x y, r = x - q0 s exactly, q = q0 + r y is the correctly rounded quotient; checked
// against x / s on 10^10 random pairs on the host).  The sign is put back by hand so that
// -0 / s stays -0.
struct Divisor {
  float s, y;
  __device__ __forceinline__ explicit Divisor(float scale) : s(scale), y(1.0f / scale) {}
};
__device__ __forceinline__ float div_rn(float x, Divisor d) {
  const float q0 = x * d.y;
  const float r = fmaf(-q0, d.s, x);
  const float q = __builtin_isfinite(q0) ? fmaf(r, d.y, q0) : q0;   // overflow: keep the inf
  const uint32_t sign = (__float_as_uint(x) ^ __float_as_uint(d.s)) & 0x80000000u;
  return __uint_as_float((__float_as_uint(q) & 0x7fffffffu) | sign);
}
template <typename T> __device__ __forceinline__ float kv_quant_prep(T x, Divisor scale) {
  float q = div_rn(to_f32(x), scale);
  if constexpr (sizeof(T) == 2 && !__is_same(T, bf16_t)) q = (float)(f16_t)q;
  return q;
}

// N elements at once (cache writes): for bf16 sources the same arithmetic with packed fp32
// instructions on |x| -- v_pk_mul / v_pk_fma for the division, one integer max over the
// quotients' bit patterns to find a non-finite one (NaN / inf inputs, overflowing quotients:
// the whole vector then takes the scalar path above, wave-uniformly), min with 448 (the quotient
// is >= 0), sign (x ^ scale) put back with one or-instruction, two elements per conversion: ~7
// VALU instructions per element instead of ~16.  Results are identical bit for bit.
template <typename T, int N> struct alignas(sizeof(T) * N) Vec { T v[N]; };
template <typename T, int N>
__device__ __forceinline__ void kv_quant_vec(const Vec<T, N>& in, Divisor d, Vec<uint8_t, N>& out) {
  if constexpr (__is_same(T, bf16_t) && N % 4 == 0) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&in);
    const float as = fabsf(d.s), ay = fabsf(d.y);
    const uint32_t sm = (__float_as_uint(d.s) >> 16) & 0x8000u;
    const uint32_t sm2 = sm | (sm << 16);
    float q[N];
    uint32_t sg[N / 2], mx = 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const uint32_t xw = w[i];
      sg[i] = (xw & 0x80008000u) ^ sm2;
      const uint32_t aw = xw & 0x7fff7fffu;
      const f2 ax = {__uint_as_float(aw << 16), __uint_as_float(aw & 0xffff0000u)};
      const f2 q0 = ax * (f2){ay, ay};
      const f2 r = __builtin_elementwise_fma(-q0, (f2){as, as}, ax);
      const f2 qq = __builtin_elementwise_fma(r, (f2){ay, ay}, q0);
      q[2 * i] = qq.x;
      q[2 * i + 1] = qq.y;
      const uint32_t m2 = __float_as_uint(qq.x) > __float_as_uint(qq.y) ? __float_as_uint(qq.x) : __float_as_uint(qq.y);
      mx = mx > m2 ? mx : m2;
    }
    if (__builtin_expect(__any(mx >= 0x7f800000u), 0)) {
#pragma unroll
      for (int j = 0; j < N; ++j) out.v[j] = f32_to_fp8(kv_quant_prep(in.v[j], d));
      return;
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      int word = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * i + h;
        const float a = __uint_as_float(__float_as_uint(fminf(q[2 * k], 448.0f)) | (sg[k] << 16));
        const float b = __uint_as_float(__float_as_uint(fminf(q[2 * k + 1], 448.0f)) | (sg[k] & 0x80000000u));
        word = h == 0 ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, word, false)
                      : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, word, true);
      }
      o[i] = (uint32_t)word;
    }
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) out.v[j] = f32_to_fp8(kv_quant_prep(in.v[j], d));
  }
}

// ---- wave64 reductions ------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Unpartitioned launches of 257 .. 1024 workgroups (paged_attention_v1: no scratch, one workgroup per
// sequence and kv head): which sequence a HARDWARE workgroup id takes.  Ids i and i + 256 run on
// the same CU (measured: tools/pa_placement.py, 100 % of 512 workgroups, two per CU), everything is
// resident at once, and a CU that drew two long sequences of a ragged batch sets the launch time
// (64 x U[512, 4096] tokens: the CU with 3566 + 3934 tokens exits at 59 us, the one with
// 1021 + 1525 at 24 us).  A hardware id keeps its (kv head, q-head chunk) = id % groups -- which
// XCD streams which head does not change -- and takes, among the ids of its group (slot j = id /
// groups), a sequence by length rank (ties by index): the nf slots whose ids are below 256 take
// ranks nf-1 .. 0, the later slots ranks nf, nf+1, ...: the CU with the shortest of the first
// round gets the longest of the rest, and ids past 512, dispatched as CUs free up, come longest
// first.  Every workgroup derives the same ranking from seq_lens by itself: the lengths
// sit one per lane (4, 8 or 16 registers: up to 1024 sequences), the rank-th largest key is found by a
// bitwise search with ballots (scalar work), ties by lane order.  A batch of equal lengths
// (one ballot) keeps the identity order.  Keys are clamped to `bits` bits, so a length above the
// caller's max_seq_len only perturbs the order, never the bijection.  Returns the logical
// workgroup id seq * groups + group and the sequence's length.
// xcd_grouped (launches of >= 1024 workgroups whose hardware ids are remapped so that XCD x = id % 8
// serves the sequences [x S/8, (x+1) S/8) with all their groups; S % 8 == 0): the workgroups are
// handed out as CUs free up, so the order is plain longest first, dealt to the XCDs snake-wise
// (position t of XCD x takes rank 8 t + x, or 8 t + 7 - x for odd t) so that every XCD gets the
// same share of long sequences; the tail of the launch is then made of the shortest sequences.
constexpr int kLptMaxSeqs = 1024;   // 16 registers of 64 lengths
template <int NR = 4>   // NR registers of 64 lengths: S <= 64 NR
__device__ __forceinline__ int pa_lpt_wg(const int* __restrict__ seq_lens, int S, int groups, int bits,
                                         int lane, int hw, int& seq_len_out, bool xcd_grouped = false) {
  int j, grp, srank;
  if (xcd_grouped) {
    const int xcd = hw & 7, idx = hw >> 3, t = idx / groups;
    grp = idx - t * groups;
    j = xcd * (S >> 3) + t;                       // the remap's own sequence (kept for equal lengths)
    srank = 8 * t + ((t & 1) ? 7 - xcd : xcd);
  } else {
    j = hw / groups;
    grp = hw - j * groups;
    int nf = (256 - grp + groups - 1) / groups;   // slots of this group with a hardware id below 256
    nf = nf < S ? nf : S;
    srank = j < nf ? nf - 1 - j : j;
  }
  const int kmax = (int)((1u << bits) - 1u);
  int len[NR], key[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int idx = 64 * r + lane;
    len[r] = idx < S ? seq_lens[idx] : 0;
    key[r] = idx < S ? min(max(len[r], 0), kmax) : -1;
  }
  const int k0 = __builtin_amdgcn_readfirstlane(key[0]);
  bool uniform = true;
#pragma unroll
  for (int r = 0; r < NR; ++r) uniform = uniform && __ballot(key[r] == k0 || key[r] < 0) == ~0ull;
  int seq = j;
  if (!uniform) {
    seq = srank;
    int cand = 0;
    for (int b = bits - 1; b >= 0; --b) {
      const int t = cand | (1 << b);
      int cnt = 0;
#pragma unroll
      for (int r = 0; r < NR; ++r) cnt += __popcll(__ballot(key[r] >= t));
      if (cnt > srank) cand = t;
    }
    int k = srank;
#pragma unroll
    for (int r = 0; r < NR; ++r) k -= __popcll(__ballot(key[r] > cand));
    int found = -1;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      unsigned long long m = __ballot(key[r] == cand);
      const int c = __popcll(m);
      if (found < 0) {
        if (k < c) {
          int pos = 0;   // the k-th set bit of m
#pragma unroll
          for (int w = 32; w >= 1; w >>= 1) {
            const unsigned long long low = m & ((1ull << w) - 1ull);
            const int cl = __popcll(low);
            if (k >= cl) { k -= cl; m >>= w; pos += w; } else { m = low; }
          }
          found = 64 * r + pos;
        } else {
          k -= c;
        }
      }
    }
    if (found >= 0) seq = found;
  }
  seq = __builtin_amdgcn_readfirstlane(seq);
  const int rr = seq >> 6;
  int v = len[0];
#pragma unroll
  for (int r = 1; r < NR; ++r) v = rr == r ? len[r] : v;
  seq_len_out = __builtin_amdgcn_readlane(v, seq & 63);
  return seq * groups + grp;
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>); indices are constants
// from the front end on, so register arrays indexed by them are never demoted to scratch
// clamp(x / s, -448, 448) for s > 0 with y = 1.0f / s (correctly rounded, computed once per
// row): Markstein's FMA sequence q0 = |x| y, r = |x| - q0 s (exact), q = q0 + r y gives the
// correctly rounded quotient (3 VALU ops instead of the ~11 of an IEEE division; compared
// against x / s on 10^10 random pairs on the host: no difference).  Done on |x| so that an
// overflowing quotient (inf -> NaN in the correction) still clamps to 448; the sign goes back
// on at the end, which also keeps -0.
__device__ __forceinline__ float div_clamp448(float x, float s, float y) {
  const float ax = fabsf(x);
  const float q0 = ax * y;
  const float r = fmaf(-q0, s, ax);
  const float q = fminf(fmaf(r, y, q0), 448.0f);
  return copysignf(q, x);
}

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace va
