// KV-cache writes for the paged cache: reshape_and_cache, reshape_and_cache_flash,
// convert_fp8, copy_blocks.  Pure HBM-bound byte movement (+ e4m3 conversion):
// every thread moves 16-byte vectors on the contiguous side of each layout.
//
// Semantics follow the reference operators cited in include/vllm_amd.h
// (csrc/cache_kernels.cu:210-306, 505-576, 71-170); the kernels themselves are
// written for wave64 / 16-byte accesses and share no structure with them.
#include "common.cuh"

namespace va {


// ---------------------------------------------------------------------------
// reshape_and_cache: K -> [blk, h, d/x, off, x]  V -> [blk, h, d, off]
// One workgroup per token; one thread per x-chunk (x elements = one 16-byte
// cache vector of K).  The K store is a single 16-byte (or x*sizeof) vector,
// the V stores are the layout's inherent element scatter (stride block_size).
// ---------------------------------------------------------------------------
template <typename scalar_t, typename cache_t, bool FP8, int X>
__global__ void reshape_and_cache_kernel(
    const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
    cache_t* __restrict__ key_cache, cache_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, int64_t key_stride,
    int64_t value_stride, int num_heads, int head_size, int block_size,
    const float* __restrict__ k_scale_p, const float* __restrict__ v_scale_p) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const int chunks_per_head = head_size / X;
  const int nchunks = num_heads * chunks_per_head;
  Divisor ks(1.f), vs(1.f);
  if constexpr (FP8) { ks = Divisor(*k_scale_p); vs = Divisor(*v_scale_p); }

  const scalar_t* krow = key + token * key_stride;
  const scalar_t* vrow = value + token * value_stride;
  const int64_t blk_elems = (int64_t)num_heads * head_size * block_size;

  for (int c = threadIdx.x; c < nchunks; c += blockDim.x) {
    const int h = c / chunks_per_head;
    const int xc = c % chunks_per_head;
    const int src = h * head_size + xc * X;
    Vec<scalar_t, X> kv = *reinterpret_cast<const Vec<scalar_t, X>*>(krow + src);
    Vec<scalar_t, X> vv = *reinterpret_cast<const Vec<scalar_t, X>*>(vrow + src);
    const int64_t kdst = blk * blk_elems + (int64_t)h * head_size * block_size +
                         (int64_t)xc * block_size * X + off * X;
    const int64_t vdst = blk * blk_elems + (int64_t)h * head_size * block_size +
                         (int64_t)(xc * X) * block_size + off;
    if constexpr (FP8) {
      Vec<uint8_t, X> kq;
      kv_quant_vec(kv, ks, kq);
      *reinterpret_cast<Vec<uint8_t, X>*>(key_cache + kdst) = kq;
#pragma unroll
      for (int j = 0; j < X; ++j)
        value_cache[vdst + (int64_t)j * block_size] = f32_to_fp8(kv_quant_prep(vv.v[j], vs));
    } else {
      *reinterpret_cast<Vec<scalar_t, X>*>(key_cache + kdst) = kv;
#pragma unroll
      for (int j = 0; j < X; ++j) value_cache[vdst + (int64_t)j * block_size] = vv.v[j];
    }
  }
}

// ---------------------------------------------------------------------------
// reshape_and_cache for many tokens: one workgroup per (16 consecutive tokens, head).
// Prefill and chunked prefill hand over runs of consecutive slots; for such a group the V
// tile is transposed through LDS and every V row [d][16 tokens] leaves as ONE 16-byte
// (fp8) store instead of 16 single-byte stores, and the K vectors of consecutive tokens
// are contiguous 256-byte pieces.  A run may start anywhere in a block (chunk boundaries
// do): the LDS columns keep the slots' alignment mod 16, partial 16-slot groups are
// written as 4-byte units and single elements.  A group whose slots are not consecutive
// (decode batches, padding slots < 0) takes the per-token scatter inside the same kernel.
// Requires block_size % 16 == 0 and head_size <= 256.
// ---------------------------------------------------------------------------
template <typename scalar_t, typename cache_t, bool FP8>
__global__ __launch_bounds__(256) void reshape_and_cache_runs_kernel(
    const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
    cache_t* __restrict__ key_cache, cache_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, int64_t key_stride, int64_t value_stride,
    int num_tokens, int num_heads, int head_size, int block_size, int bs_log2, int hpw,
    const float* __restrict__ k_scale_p, const float* __restrict__ v_scale_p) {
  constexpr int CS = sizeof(cache_t);
  constexpr int X = 16 / CS;    // elements of one K cache vector
  constexpr int G = 16;         // tokens per workgroup
  constexpr int U = 4 / CS;     // elements of a 4-byte unit
  constexpr int W = 512;        // head elements (hpw * head_size <= W) per workgroup
  constexpr int KI = W / X * G / 256, VI = W / 8 * G / 256, RI = W / 256;   // items per thread
  __shared__ int64_t s_slot[G];
  __shared__ __attribute__((aligned(16))) cache_t vt[G][W + X];   // [token][head, d], padded rows
  const int tid = threadIdx.x;
  // slot -> (block, offset): a shift for the usual power-of-two block sizes (a 64-bit
  // division per item otherwise)
  auto split = [&](int64_t slot, int64_t& blk, int64_t& off) {
    if (bs_log2 >= 0) { blk = slot >> bs_log2; off = slot & (block_size - 1); }
    else { blk = slot / block_size; off = slot % block_size; }
  };
  const int t0 = blockIdx.x * G, h0 = blockIdx.y * hpw;
  const int n = num_tokens - t0 < G ? num_tokens - t0 : G;
  const int nh = num_heads - h0 < hpw ? num_heads - h0 : hpw;
  const int width = nh * head_size;   // contiguous source elements per token
  if (tid < G) s_slot[tid] = tid < n ? slot_mapping[t0 + tid] : -1;

  // every load of the workgroup is issued before the first use (rows past n: token 0)
  const scalar_t* kbase = key + (int64_t)t0 * key_stride + h0 * head_size;
  const scalar_t* vbase = value + (int64_t)t0 * value_stride + h0 * head_size;
  const int cpt = width / X;          // K chunks per token
  Vec<scalar_t, X> kreg[KI];
#pragma unroll
  for (int u = 0; u < KI; ++u) {      // item = (chunk, token), token fastest
    const int it = tid + 256 * u;
    const int i = it % G, c = it / G;
    const bool ok = c < cpt && i < n;
    kreg[u] = *reinterpret_cast<const Vec<scalar_t, X>*>(kbase + (ok ? (int64_t)i * key_stride + c * X : 0));
  }
  const int cpv = width / 8;          // V chunks of 8 per token
  const int cpv_log2 = (cpv & (cpv - 1)) == 0 ? 31 - __clz(cpv) : -1;
  auto vitem = [&](int it, int& i, int& c) {   // item -> (token, chunk), chunk fastest
    if (cpv_log2 >= 0) { i = it >> cpv_log2; c = it & (cpv - 1); }
    else { i = it / cpv; c = it % cpv; }
  };
  Vec<scalar_t, 8> vreg[VI];
#pragma unroll
  for (int u = 0; u < VI; ++u) {      // item = (token, chunk), chunk fastest
    const int it = tid + 256 * u;
    int i, c;
    vitem(it, i, c);
    const bool ok = i < n;
    vreg[u] = *reinterpret_cast<const Vec<scalar_t, 8>*>(vbase + (ok ? (int64_t)i * value_stride + c * 8 : 0));
  }
  Divisor ks(1.f), vs(1.f);
  if constexpr (FP8) { ks = Divisor(*k_scale_p); vs = Divisor(*v_scale_p); }
  __syncthreads();

  // a "run": consecutive slots up to the next multiple of 16 (segment A), then consecutive
  // slots from a multiple of 16 (segment B: the sequence's next block, anywhere in the cache)
  const int64_t slot0 = s_slot[0];
  const int colbase = slot0 >= 0 ? (int)(slot0 % 16) : 0;
  const int nA = n < 16 - colbase ? n : 16 - colbase;
  const int64_t slotB = nA < n ? s_slot[nA] : 0;
  bool run = slot0 >= 0 && slotB >= 0 && slotB % 16 == 0;
  for (int i = 1; i < nA; ++i) run = run && s_slot[i] == slot0 + i;
  for (int i = nA + 1; i < n; ++i) run = run && s_slot[i] == slotB + (i - nA);
  auto conv = [](scalar_t v, Divisor sc) -> cache_t {
    if constexpr (FP8) return f32_to_fp8(kv_quant_prep(v, sc));
    else return v;
  };
  const int64_t blk_elems = (int64_t)num_heads * head_size * block_size;
  const int64_t head_off = (int64_t)h0 * head_size * block_size;

  // K: a wave writes 256-byte pieces of a run (16 tokens x one 16-byte vector)
  const int cph = head_size / X;
#pragma unroll
  for (int u = 0; u < KI; ++u) {
    const int it = tid + 256 * u;
    const int i = it % G, c = it / G;
    const int64_t slot = s_slot[i];
    if (c >= cpt || slot < 0) continue;   // slot < 0 also covers i >= n
    int64_t blk, off;
    split(slot, blk, off);
    Vec<cache_t, X> kq;
    if constexpr (FP8) {
      kv_quant_vec(kreg[u], ks, kq);
    } else {
#pragma unroll
      for (int j = 0; j < X; ++j) kq.v[j] = conv(kreg[u].v[j], ks);
    }
    const int hl = c / cph, xc = c % cph;
    *reinterpret_cast<Vec<cache_t, X>*>(key_cache + blk * blk_elems + head_off +
                                         ((int64_t)hl * head_size * block_size) +
                                         ((int64_t)xc * block_size + off) * X) = kq;
  }

  // V: into the LDS tile (run) or scattered (anything else)
#pragma unroll
  for (int u = 0; u < VI; ++u) {
    const int it = tid + 256 * u;
    int i, c;
    vitem(it, i, c);
    if (i >= G) continue;
    const int64_t slot = s_slot[i];
    if (slot < 0) continue;
    Vec<cache_t, 8> vq;
    if constexpr (FP8) {
      kv_quant_vec(vreg[u], vs, vq);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) vq.v[j] = conv(vreg[u].v[j], vs);
    }
    if (run) {
      *reinterpret_cast<Vec<cache_t, 8>*>(&vt[i][c * 8]) = vq;
    } else {
      int64_t blk, off;
    split(slot, blk, off);
      cache_t* dst = value_cache + blk * blk_elems + head_off + (int64_t)(c * 8) * block_size + off;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[(int64_t)j * block_size] = vq.v[j];
    }
  }
  if (!run) return;
  __syncthreads();

  // rows of the transposed tile: column q = (slot0 % 16) + token keeps the slots' alignment,
  // so columns [0,16) (segment A) and [16,32) (segment B) are 16-slot groups inside a block
  if constexpr (CS == 1) {
    // fp8, whole 16-slot groups (round 3): a thread takes FOUR consecutive d -- one 4-byte LDS read per token
    // instead of four single-byte ones (the byte reads made this kernel LDS-issue bound: 16 ds_read_u8 per
    // 16-byte row) -- transposes the 4 x 16 bytes in registers (v_perm_b32) and stores four 16-byte rows.
    // Partial groups (the two ends of a run that starts mid-block) keep the per-row path below: the same
    // scheme with 4-byte and single-byte stores measured slower there (32768 tokens 50.6 -> 79.5 us).
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int q0 = 16 * hh;
      const int lo = colbase - q0 > 0 ? colbase - q0 : 0;
      const int hi = colbase + n - q0 < 16 ? colbase + n - q0 : 16;
      if (!(lo == 0 && hi == 16)) continue;   // uniform
      const int64_t sbase = hh == 0 ? slot0 - colbase : slotB;
      int64_t blk, off;
      split(sbase, blk, off);
      for (int dg = tid; dg < width / 4; dg += 256) {
        uint32_t col[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
          col[p] = *reinterpret_cast<const uint32_t*>(&vt[q0 + p - colbase][dg * 4]);
        cache_t* dst = value_cache + blk * blk_elems + head_off + (int64_t)(dg * 4) * block_size + off;
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t sel = 0x0c0c0000u | ((4u + e) << 8) | (uint32_t)e;   // byte e of (S1, S0)
          u32x4 row;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t t0 = __builtin_amdgcn_perm(col[4 * k + 1], col[4 * k], sel);
            const uint32_t t1 = __builtin_amdgcn_perm(col[4 * k + 3], col[4 * k + 2], sel);
            row[k] = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
          }
          *reinterpret_cast<u32x4*>(dst + (int64_t)e * block_size) = row;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RI; ++u) {
    const int d = tid + 256 * u;   // (head, d) row: rows of consecutive heads are contiguous
    if (d >= width) continue;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int q0 = 16 * hh;
      const int lo = colbase - q0 > 0 ? colbase - q0 : 0;
      const int hi = colbase + n - q0 < 16 ? colbase + n - q0 : 16;
      if (hi <= lo) continue;   // uniform
      if (CS == 1 && lo == 0 && hi == 16) continue;   // done above
      const int64_t sbase = hh == 0 ? slot0 - colbase : slotB;   // first slot of the 16-slot group
      int64_t blk, off;
      split(sbase, blk, off);
      cache_t* dst = value_cache + blk * blk_elems + head_off + (int64_t)d * block_size + off;
      struct alignas(16) Row { cache_t e[16]; } row;
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int i = q0 + p - colbase;
        row.e[p] = (p >= lo && p < hi) ? vt[i][d] : cache_t{};
      }
      if (lo == 0 && hi == 16) {
        *reinterpret_cast<Row*>(dst) = row;
      } else {
#pragma unroll
        for (int uu = 0; uu < 16 / U; ++uu) {
          if (uu * U >= lo && uu * U + U <= hi) {
            *reinterpret_cast<Vec<cache_t, U>*>(dst + uu * U) =
                *reinterpret_cast<const Vec<cache_t, U>*>(&row.e[uu * U]);
          } else {
#pragma unroll
            for (int j = 0; j < U; ++j)
              if (uu * U + j >= lo && uu * U + j < hi) dst[uu * U + j] = row.e[uu * U + j];
          }
        }
      }
    }
  }
}

// Scalar fallback for layouts / alignments the vector kernel does not cover
// (odd x, misaligned rows).  Same mapping as the reference element loop.
template <typename scalar_t, typename cache_t, bool FP8>
__global__ void reshape_and_cache_scalar_kernel(
    const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
    cache_t* __restrict__ key_cache, cache_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, int64_t key_stride,
    int64_t value_stride, int num_heads, int head_size, int block_size, int x,
    const float* __restrict__ k_scale_p, const float* __restrict__ v_scale_p) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  Divisor ks(1.f), vs(1.f);
  if constexpr (FP8) { ks = Divisor(*k_scale_p); vs = Divisor(*v_scale_p); }
  const int n = num_heads * head_size;
  const int64_t blk_elems = (int64_t)n * block_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / head_size, d = i % head_size;
    const int64_t kdst = blk * blk_elems + (int64_t)h * head_size * block_size +
                         (int64_t)(d / x) * block_size * x + off * x + d % x;
    const int64_t vdst = blk * blk_elems + (int64_t)h * head_size * block_size +
                         (int64_t)d * block_size + off;
    scalar_t k = key[token * key_stride + i];
    scalar_t v = value[token * value_stride + i];
    if constexpr (FP8) {
      key_cache[kdst] = f32_to_fp8(kv_quant_prep(k, ks));
      value_cache[vdst] = f32_to_fp8(kv_quant_prep(v, vs));
    } else {
      key_cache[kdst] = k;
      value_cache[vdst] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// reshape_and_cache_flash: K,V -> [blk, off, h, d] through explicit strides.
// One thread per VEC contiguous head elements: both sides are 16-byte vectors.
// ---------------------------------------------------------------------------
template <typename scalar_t, typename cache_t, bool FP8, int VEC>
__global__ void reshape_and_cache_flash_kernel(
    const scalar_t* __restrict__ key, const scalar_t* __restrict__ value,
    cache_t* __restrict__ key_cache, cache_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, int64_t block_stride,
    int64_t page_stride, int64_t head_stride, int64_t key_stride,
    int64_t value_stride, int num_heads, int head_size, int block_size,
    const float* __restrict__ k_scale_p, const float* __restrict__ v_scale_p) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  Divisor ks(1.f), vs(1.f);
  if constexpr (FP8) { ks = Divisor(*k_scale_p); vs = Divisor(*v_scale_p); }
  const int vec_per_head = head_size / VEC;
  const int nvec = num_heads * vec_per_head;
  const scalar_t* krow = key + token * key_stride;
  const scalar_t* vrow = value + token * value_stride;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    const int h = c / vec_per_head;
    const int d = (c % vec_per_head) * VEC;
    const int src = h * head_size + d;
    const int64_t dst = blk * block_stride + off * page_stride + h * head_stride + d;
    Vec<scalar_t, VEC> kv = *reinterpret_cast<const Vec<scalar_t, VEC>*>(krow + src);
    Vec<scalar_t, VEC> vv = *reinterpret_cast<const Vec<scalar_t, VEC>*>(vrow + src);
    if constexpr (FP8) {
      Vec<uint8_t, VEC> kq, vq;
      kv_quant_vec(kv, ks, kq);
      kv_quant_vec(vv, vs, vq);
      *reinterpret_cast<Vec<uint8_t, VEC>*>(key_cache + dst) = kq;
      *reinterpret_cast<Vec<uint8_t, VEC>*>(value_cache + dst) = vq;
    } else {
      *reinterpret_cast<Vec<scalar_t, VEC>*>(key_cache + dst) = kv;
      *reinterpret_cast<Vec<scalar_t, VEC>*>(value_cache + dst) = vv;
    }
  }
}

// ---------------------------------------------------------------------------
// convert_fp8 (test helper in the reference, also used to build fp8 caches)
// ---------------------------------------------------------------------------
template <typename scalar_t, bool TO_FP8>
__global__ void convert_fp8_kernel(void* __restrict__ dst_, const void* __restrict__ src_,
                                   int64_t n, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const Divisor dv(scale);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if constexpr (TO_FP8) {
      const scalar_t* src = (const scalar_t*)src_;
      ((uint8_t*)dst_)[i] = f32_to_fp8(kv_quant_prep(src[i], dv));
    } else {
      const uint8_t* src = (const uint8_t*)src_;
      ((scalar_t*)dst_)[i] = from_f32<scalar_t>(fp8_to_f32(src[i]) * scale);
    }
  }
}

// ---------------------------------------------------------------------------
// copy_blocks: grid (pair, 2 {K,V}); 16-byte vectors
// ---------------------------------------------------------------------------
__global__ void copy_blocks_kernel(uint8_t* __restrict__ kc, uint8_t* __restrict__ vc,
                                   const int64_t* __restrict__ mapping, int64_t block_bytes) {
  const int64_t src = mapping[2 * blockIdx.x];
  const int64_t dst = mapping[2 * blockIdx.x + 1];
  uint8_t* base = blockIdx.y == 0 ? kc : vc;
  const uint8_t* s = base + src * block_bytes;
  uint8_t* d = base + dst * block_bytes;
  if ((block_bytes & 15) == 0 && (((uintptr_t)base) & 15) == 0) {
    const int64_t nv = block_bytes >> 4;
    for (int64_t i = threadIdx.x; i < nv; i += blockDim.x)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else {
    for (int64_t i = threadIdx.x; i < block_bytes; i += blockDim.x) d[i] = s[i];
  }
}

}  // namespace va

using namespace va;

namespace va { int g_cache_runs = 1; int g_cache_hpw = 0; }   // "cache_hpw": heads per workgroup of the runs kernel (0 = heuristic)   // tuning "cache_runs": 0 = always the per-token kernel

static inline bool aligned(const void* p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

template <typename scalar_t, bool FP8>
static int launch_reshape_and_cache(const void* key, const void* value, void* kc, void* vc,
                                    const int64_t* slot_mapping, int num_tokens, int num_heads,
                                    int head_size, int block_size, int x, int64_t key_stride,
                                    int64_t value_stride, const float* ks, const float* vs,
                                    hipStream_t st) {
  typedef typename std::conditional<FP8, uint8_t, scalar_t>::type cache_t;
  constexpr int XN = FP8 ? 16 : 16 / (int)sizeof(scalar_t);  // natural x
  const bool vec_ok = (x == XN) && (head_size % x == 0) &&
                      (key_stride % x == 0) && (value_stride % x == 0) &&
                      aligned(key, sizeof(scalar_t) * XN) && aligned(value, sizeof(scalar_t) * XN) &&
                      aligned(kc, 16);
  // many tokens: the run-aware kernel (16 tokens x 1 head per workgroup)
  const bool runs_ok = vec_ok && sizeof(scalar_t) == 2 && num_tokens >= 256 && block_size % 16 == 0 &&
                       head_size % 8 == 0 && head_size <= 256 && aligned(vc, 16);
  if (runs_ok && g_cache_runs) {
    int bs_log2 = -1;
    for (int i = 4; i < 16; ++i) if (block_size == (1 << i)) bs_log2 = i;
    int hpw = 512 / head_size;   // heads per workgroup: hpw * head_size <= 512 elements
    if (hpw > num_heads) hpw = num_heads;
    // fewer heads per workgroup while the launch would not fill the chip a few times over
    // (round 3, with the 4-byte V path: 8192 tokens 19.8 us with 1 head per workgroup, 14.3 with 4; 2048 tokens 6.8 / 6.2
    // with 1 / 2: tools/bench_cache_hpw.py)
    while (hpw > 1 && (int64_t)((num_tokens + 15) / 16) * ((num_heads + hpw - 1) / hpw) < 512) hpw /= 2;
    if (g_cache_hpw > 0 && g_cache_hpw * head_size <= 512 && g_cache_hpw <= num_heads) hpw = g_cache_hpw;
    hipLaunchKernelGGL((reshape_and_cache_runs_kernel<scalar_t, cache_t, FP8>),
                       dim3((num_tokens + 15) / 16, (num_heads + hpw - 1) / hpw), dim3(256), 0, st,
                       (const scalar_t*)key, (const scalar_t*)value, (cache_t*)kc, (cache_t*)vc,
                       slot_mapping, key_stride, value_stride, num_tokens, num_heads, head_size,
                       block_size, bs_log2, hpw, ks, vs);
  } else if (vec_ok) {
    const int nchunks = num_heads * head_size / x;
    int threads = nchunks < 256 ? ((nchunks + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL((reshape_and_cache_kernel<scalar_t, cache_t, FP8, XN>), dim3(num_tokens),
                       dim3(threads), 0, st, (const scalar_t*)key, (const scalar_t*)value,
                       (cache_t*)kc, (cache_t*)vc, slot_mapping, key_stride, value_stride,
                       num_heads, head_size, block_size, ks, vs);
  } else {
    int n = num_heads * head_size;
    int threads = n < 512 ? ((n + 63) / 64) * 64 : 512;
    hipLaunchKernelGGL((reshape_and_cache_scalar_kernel<scalar_t, cache_t, FP8>),
                       dim3(num_tokens), dim3(threads), 0, st, (const scalar_t*)key,
                       (const scalar_t*)value, (cache_t*)kc, (cache_t*)vc, slot_mapping,
                       key_stride, value_stride, num_heads, head_size, block_size, x, ks, vs);
  }
  VA_LAUNCH_CHECK("reshape_and_cache launch");
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_reshape_and_cache(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size, int block_size,
    int x, int64_t key_stride, int64_t value_stride, int dtype, int kv_dtype,
    const float* k_scale, const float* v_scale, void* stream) {
  VA_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0 && x > 0,
             VLLM_AMD_EINVAL, "reshape_and_cache: bad shape");
  VA_REQUIRE(head_size % x == 0, VLLM_AMD_EINVAL,
             "reshape_and_cache: head_size %d not divisible by x %d", head_size, x);
  if (num_tokens == 0) return VLLM_AMD_OK;
  VA_REQUIRE(key && value && key_cache && value_cache && slot_mapping, VLLM_AMD_EINVAL,
             "reshape_and_cache: null pointer");
  hipStream_t st = (hipStream_t)stream;
#define RC(T, F) \
  return launch_reshape_and_cache<T, F>(key, value, key_cache, value_cache, slot_mapping, \
      num_tokens, num_heads, head_size, block_size, x, key_stride, value_stride, k_scale, v_scale, st)
  if (kv_dtype == VLLM_AMD_KV_AUTO) {
    if (dtype == VLLM_AMD_F32) RC(float, false);
    if (dtype == VLLM_AMD_F16) RC(f16_t, false);
    if (dtype == VLLM_AMD_BF16) RC(bf16_t, false);
    VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported input type of kv cache: %d", dtype);
  } else if (kv_dtype == VLLM_AMD_KV_FP8_E4M3) {
    VA_REQUIRE(k_scale && v_scale, VLLM_AMD_EINVAL, "reshape_and_cache: null scale");
    if (dtype == VLLM_AMD_F32) RC(float, true);
    if (dtype == VLLM_AMD_F16) RC(f16_t, true);
    if (dtype == VLLM_AMD_BF16) RC(bf16_t, true);
    VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported input type of kv cache: %d", dtype);
  }
#undef RC
  VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported data type of kv cache: %d", kv_dtype);
}

template <typename scalar_t, bool FP8>
static int launch_flash(const void* key, const void* value, void* kc, void* vc,
                        const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                        int block_size, int64_t key_stride, int64_t value_stride,
                        int64_t block_stride, int64_t page_stride, int64_t head_stride,
                        const float* ks, const float* vs, hipStream_t st) {
  typedef typename std::conditional<FP8, uint8_t, scalar_t>::type cache_t;
  constexpr int V = 16 / (int)sizeof(scalar_t);  // 16-byte source vectors (4 or 8 elements)
  const size_t cvb = V * sizeof(cache_t);       // bytes of one cache vector
  const bool vec_ok = head_size % V == 0 && key_stride % V == 0 && value_stride % V == 0 &&
                      block_stride % V == 0 && page_stride % V == 0 && head_stride % V == 0 &&
                      aligned(key, 16) && aligned(value, 16) && aligned(kc, cvb) &&
                      aligned(vc, cvb);
#define FL(VEC)                                                                               \
  do {                                                                                        \
    const int nvec = num_heads * head_size / VEC;                                             \
    int threads = nvec < 256 ? ((nvec + 63) / 64) * 64 : 256;                                 \
    hipLaunchKernelGGL((reshape_and_cache_flash_kernel<scalar_t, cache_t, FP8, VEC>),         \
                       dim3(num_tokens), dim3(threads), 0, st, (const scalar_t*)key,          \
                       (const scalar_t*)value, (cache_t*)kc, (cache_t*)vc, slot_mapping,      \
                       block_stride, page_stride, head_stride, key_stride, value_stride,      \
                       num_heads, head_size, block_size, ks, vs);                             \
  } while (0)
  if (vec_ok) FL(V); else FL(1);
#undef FL
  VA_LAUNCH_CHECK("reshape_and_cache_flash launch");
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_reshape_and_cache_flash(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size, int block_size,
    int64_t key_stride, int64_t value_stride, int64_t block_stride, int64_t page_stride,
    int64_t head_stride, int dtype, int kv_dtype, const float* k_scale, const float* v_scale,
    void* stream) {
  VA_REQUIRE(num_tokens >= 0 && num_heads > 0 && head_size > 0 && block_size > 0,
             VLLM_AMD_EINVAL, "reshape_and_cache_flash: bad shape");
  if (num_tokens == 0) return VLLM_AMD_OK;
  VA_REQUIRE(key && value && key_cache && value_cache && slot_mapping, VLLM_AMD_EINVAL,
             "reshape_and_cache_flash: null pointer");
  hipStream_t st = (hipStream_t)stream;
#define RF(T, F) \
  return launch_flash<T, F>(key, value, key_cache, value_cache, slot_mapping, num_tokens, \
      num_heads, head_size, block_size, key_stride, value_stride, block_stride, page_stride, \
      head_stride, k_scale, v_scale, st)
  if (kv_dtype == VLLM_AMD_KV_AUTO) {
    if (dtype == VLLM_AMD_F32) RF(float, false);
    if (dtype == VLLM_AMD_F16) RF(f16_t, false);
    if (dtype == VLLM_AMD_BF16) RF(bf16_t, false);
    VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported input type of kv cache: %d", dtype);
  } else if (kv_dtype == VLLM_AMD_KV_FP8_E4M3) {
    VA_REQUIRE(k_scale && v_scale, VLLM_AMD_EINVAL, "reshape_and_cache_flash: null scale");
    if (dtype == VLLM_AMD_F32) RF(float, true);
    if (dtype == VLLM_AMD_F16) RF(f16_t, true);
    if (dtype == VLLM_AMD_BF16) RF(bf16_t, true);
    VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported input type of kv cache: %d", dtype);
  }
#undef RF
  VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported data type of kv cache: %d", kv_dtype);
}

extern "C" int vllm_amd_convert_fp8(void* dst, const void* src, int64_t n, float scale, int dtype,
                                    int to_fp8, void* stream) {
  VA_REQUIRE(n >= 0, VLLM_AMD_EINVAL, "convert_fp8: negative size");
  if (n == 0) return VLLM_AMD_OK;
  VA_REQUIRE(dst && src, VLLM_AMD_EINVAL, "convert_fp8: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
#define CV(T)                                                                                 \
  do {                                                                                        \
    if (to_fp8)                                                                               \
      hipLaunchKernelGGL((convert_fp8_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, \
                         st, dst, src, n, scale);                                             \
    else                                                                                      \
      hipLaunchKernelGGL((convert_fp8_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, \
                         st, dst, src, n, scale);                                             \
  } while (0)
  if (dtype == VLLM_AMD_F32) CV(float);
  else if (dtype == VLLM_AMD_F16) CV(f16_t);
  else if (dtype == VLLM_AMD_BF16) CV(bf16_t);
  else VA_REQUIRE(false, VLLM_AMD_EUNSUPPORTED, "Unsupported data type: %d", dtype);
#undef CV
  VA_LAUNCH_CHECK("convert_fp8 launch");
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_copy_blocks(void* key_cache, void* value_cache,
                                    const int64_t* block_mapping, int num_pairs,
                                    int64_t block_bytes, void* stream) {
  VA_REQUIRE(num_pairs >= 0 && block_bytes > 0, VLLM_AMD_EINVAL, "copy_blocks: bad shape");
  if (num_pairs == 0) return VLLM_AMD_OK;
  VA_REQUIRE(key_cache && value_cache && block_mapping, VLLM_AMD_EINVAL,
             "copy_blocks: null pointer");
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(num_pairs, 2), dim3(256), 0, (hipStream_t)stream,
                     (uint8_t*)key_cache, (uint8_t*)value_cache, block_mapping, block_bytes);
  VA_LAUNCH_CHECK("copy_blocks launch");
  return VLLM_AMD_OK;
}
