// Peer-to-peer all-reduce over xGMI for the row-parallel boundary: the sum of
// one [tokens, hidden] tensor per rank, every rank reading its peers' buffers directly --
// no ring, no staging through a collective library.  Replaces the reference's custom all-reduce
// (csrc/custom_all_reduce.cu:13-189, csrc/custom_all_reduce.cuh:296-370 kernels, :561-571 choice)
// for messages up to a few MiB, where RCCL's ring latency (7 hops of ~153 GB/s links) dominates.
//
// MI355X: the 8 GPUs of a node are fully connected point to point, so
//   * one stage  (small messages): a rank reads the whole message of every peer and adds the
//     NG inputs in rank order -- the same order on every rank, so all ranks hold bit-identical
//     results -- 1 barrier in, 1 barrier out;
//   * two stages (larger ones): rank r reduces slice r of every peer's input into its own
//     scratch (reduce-scatter by reads), barrier, then gathers the NG reduced slices from the
//     peers' scratch: 2/NG of the one-stage link traffic.
// Ranks synchronise through flag words in each other's memory: block b of rank r writes its
// launch number into peer p's `start[b][r]` and spins on its own `start[b][p]`.  Every spin is
// BOUNDED: after kSpinLimit polls the block records a time-out in its signal block and goes on
// (the result is then garbage and vllm_amd_ar_error reports it) -- a dead or late peer can never
// hang the GPU.
//
// Nothing is allocated inside all_reduce; buffers come from the caller (the helpers at the end
// wrap hipExtMallocWithFlags / hipIpc* for the Python side, as the reference's
// allocate_shared_buffer_and_handle / open_mem_handle do).
//
// STATUS: logic-tested with simulated ranks (one process, one stream per rank, one GPU) and with
// two processes sharing one GPU through IPC handles; never run across GPUs (the development box
// has one).  Off by default: tensor_model_parallel_all_reduce uses RCCL unless asked otherwise.
#include "common.cuh"

#include <mutex>
#include <unordered_map>

namespace va {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 512;
constexpr unsigned kSpinLimit = 1u << 20;   // ~1 s of polling

struct ArSignal {
  alignas(128) uint32_t start[AR_MAX_BLOCKS][AR_MAX_RANKS];
  alignas(128) uint32_t mid[AR_MAX_BLOCKS][AR_MAX_RANKS];
  alignas(128) uint32_t end[AR_MAX_BLOCKS][AR_MAX_RANKS];
  alignas(128) uint32_t epoch[AR_MAX_BLOCKS];   // launches this block has taken part in
  alignas(128) uint32_t timed_out;
};

struct ArPeers { void* p[AR_MAX_RANKS]; };

__device__ __forceinline__ void st_sys(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// All NG ranks' block b meet.  WHICH selects the flag array; FENCE: what this rank wrote before
// the barrier is visible to the peers after it (release before the flag, acquire after the spin).
template <int NG, int WHICH, bool FENCE>
__device__ __forceinline__ void ar_barrier(const ArPeers& sg, ArSignal* self, int rank, uint32_t flag) {
  if constexpr (WHICH != 0) __syncthreads();   // every thread of the block is done with the phase before
  if (threadIdx.x < NG) {
    if constexpr (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    ArSignal* peer = (ArSignal*)sg.p[threadIdx.x];
    uint32_t* theirs = WHICH == 0 ? &peer->start[blockIdx.x][rank]
                     : WHICH == 1 ? &peer->mid[blockIdx.x][rank] : &peer->end[blockIdx.x][rank];
    const uint32_t* mine = WHICH == 0 ? &self->start[blockIdx.x][threadIdx.x]
                         : WHICH == 1 ? &self->mid[blockIdx.x][threadIdx.x]
                                      : &self->end[blockIdx.x][threadIdx.x];
    st_sys(theirs, flag);
    unsigned polls = 0;
    while (ld_sys(mine) != flag) {
      if (++polls > kSpinLimit) {   // the exit every wave reaches
        st_sys(&self->timed_out, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if constexpr (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  __syncthreads();
}

template <typename T> struct ArPack { static constexpr int N = 16 / sizeof(T); };

// sum of vector `i` (16 bytes) of the NG buffers, fp32 accumulation in rank order
template <typename T, int NG>
__device__ __forceinline__ uint4 ar_sum(const ArPeers& bufs, int64_t i) {
  constexpr int N = ArPack<T>::N;
  float acc[N];
  uint4 raw = reinterpret_cast<const uint4*>(bufs.p[0])[i];
  const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
  for (int j = 0; j < N; ++j) acc[j] = to_f32(e[j]);
#pragma unroll
  for (int r = 1; r < NG; ++r) {
    raw = reinterpret_cast<const uint4*>(bufs.p[r])[i];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] += to_f32(e[j]);
  }
  uint4 o;
  T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
  for (int j = 0; j < N; ++j) oe[j] = from_f32<T>(acc[j]);
  return o;
}

// nvec = message size in 16-byte vectors
template <typename T, int NG>
__global__ __launch_bounds__(AR_THREADS) void ar_one_stage_kernel(ArPeers bufs, ArPeers sg, ArSignal* self,
                                                                  uint4* __restrict__ out, int rank,
                                                                  int64_t nvec) {
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  ar_barrier<NG, 0, false>(sg, self, rank, flag);    // every peer's input is complete (earlier kernels)
  for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * AR_THREADS)
    out[i] = ar_sum<T, NG>(bufs, i);
  ar_barrier<NG, 2, false>(sg, self, rank, flag);    // nobody overwrites an input a peer still reads
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

template <typename T, int NG>
__global__ __launch_bounds__(AR_THREADS) void ar_two_stage_kernel(ArPeers bufs, ArPeers sg, ArSignal* self,
                                                                  uint4* __restrict__ out, int rank,
                                                                  int64_t nvec) {
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  const int64_t part = nvec / NG;   // slice r = [r part, (r+1) part), the last one takes the rest
  auto slice_begin = [&](int r) { return (int64_t)r * part; };
  auto slice_end = [&](int r) { return r == NG - 1 ? nvec : (int64_t)(r + 1) * part; };
  // scratch of rank r: behind its signal block
  ArPeers tmp;
#pragma unroll
  for (int r = 0; r < NG; ++r) tmp.p[r] = (char*)sg.p[r] + sizeof(ArSignal);
  ar_barrier<NG, 0, false>(sg, self, rank, flag);
  {
    uint4* mine = (uint4*)tmp.p[rank];
    const int64_t b = slice_begin(rank), e = slice_end(rank);
    for (int64_t i = b + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < e; i += (int64_t)gridDim.x * AR_THREADS)
      mine[i - b] = ar_sum<T, NG>(bufs, i);
  }
  ar_barrier<NG, 1, true>(sg, self, rank, flag);     // the reduced slices are visible to the peers
  // gather: block b of every rank reduced the same positions of its slice, so block b only reads
  // what the peers' block b wrote -- the barrier above is per block
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const int r = (rank + k) % NG;                    // start at the own slice: spreads the links
    const uint4* src = (const uint4*)tmp.p[r];
    const int64_t b = slice_begin(r), e = slice_end(r);
    for (int64_t i = b + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < e; i += (int64_t)gridDim.x * AR_THREADS)
      out[i] = src[i - b];
  }
  ar_barrier<NG, 2, false>(sg, self, rank, flag);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
}

struct ArState {
  int world = 0, rank = 0;
  bool fully_connected = false;
  ArPeers sg;
  std::unordered_map<const void*, ArPeers> buffers;
  std::mutex mu;
};

template <typename T>
static void ar_launch(ArState* st, const ArPeers& bufs, void* out, int64_t nvec, int64_t bytes, hipStream_t s) {
  ArSignal* self = (ArSignal*)st->sg.p[st->rank];
  int64_t blocks = (nvec + AR_THREADS - 1) / AR_THREADS;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  // custom_all_reduce.cuh:561-571: two ranks always one stage; fully connected: one stage below
  // 512 KiB (<= 4 ranks) / 256 KiB (<= 8 ranks), two stages above
  const bool one = st->world == 2 || (st->world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024;
#define AR_KL(NG)                                                                                   \
  do {                                                                                              \
    if (one)                                                                                        \
      hipLaunchKernelGGL((ar_one_stage_kernel<T, NG>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, \
                         s, bufs, st->sg, self, (uint4*)out, st->rank, nvec);                       \
    else                                                                                            \
      hipLaunchKernelGGL((ar_two_stage_kernel<T, NG>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, \
                         s, bufs, st->sg, self, (uint4*)out, st->rank, nvec);                       \
  } while (0)
  switch (st->world) {
    case 2: AR_KL(2); break;
    case 4: AR_KL(4); break;
    case 6: AR_KL(6); break;
    default: AR_KL(8); break;
  }
#undef AR_KL
}

}  // namespace va

using namespace va;

extern "C" int64_t vllm_amd_ar_meta_size(void) { return (int64_t)sizeof(ArSignal); }

extern "C" int vllm_amd_ar_init(void** handle, void* const* meta_ptrs, int world_size, int rank,
                                int fully_connected) {
  VA_REQUIRE(handle && meta_ptrs, VLLM_AMD_EINVAL, "ar_init: null pointer");
  VA_REQUIRE(world_size == 2 || world_size == 4 || world_size == 6 || world_size == 8, VLLM_AMD_EUNSUPPORTED,
             "custom allreduce only supports num gpus in (2,4,6,8). Actual num gpus = %d", world_size);
  VA_REQUIRE(world_size % 2 == 0 && rank >= 0 && rank < world_size, VLLM_AMD_EINVAL, "invalid rank passed in");
  VA_REQUIRE(world_size == 2 || fully_connected, VLLM_AMD_EUNSUPPORTED,
             "custom allreduce: more than two ranks need fully connected links");
  ArState* st = new ArState();
  st->world = world_size; st->rank = rank; st->fully_connected = fully_connected != 0;
  for (int i = 0; i < AR_MAX_RANKS; ++i) st->sg.p[i] = i < world_size ? meta_ptrs[i] : nullptr;
  for (int i = 0; i < world_size; ++i) VA_REQUIRE(meta_ptrs[i] != nullptr, VLLM_AMD_EINVAL, "ar_init: null meta pointer");
  *handle = st;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_register_buffer(void* handle, void* const* ptrs) {
  VA_REQUIRE(handle && ptrs, VLLM_AMD_EINVAL, "ar_register_buffer: null pointer");
  ArState* st = (ArState*)handle;
  ArPeers p;
  for (int i = 0; i < AR_MAX_RANKS; ++i) p.p[i] = i < st->world ? ptrs[i] : nullptr;
  std::lock_guard<std::mutex> lk(st->mu);
  st->buffers[ptrs[st->rank]] = p;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_all_reduce(void* handle, const void* inp, void* out, int64_t numel, int dtype,
                                      void* reg_buffer, int64_t reg_buffer_bytes, void* stream) {
  VA_REQUIRE(handle && inp && out, VLLM_AMD_EINVAL, "ar_all_reduce: null pointer");
  ArState* st = (ArState*)handle;
  const int es = dtype_size(dtype);
  VA_REQUIRE(es > 0, VLLM_AMD_EUNSUPPORTED, "custom allreduce only supports float32, float16 and bfloat16");
  const int64_t bytes = numel * es;
  VA_REQUIRE(numel >= 0 && bytes % 16 == 0, VLLM_AMD_EINVAL,
             "custom allreduce currently requires input length to be multiple of 16 bytes");
  VA_REQUIRE((((uintptr_t)inp) & 15) == 0 && (((uintptr_t)out) & 15) == 0, VLLM_AMD_EINVAL,
             "ar_all_reduce: buffers must be 16-byte aligned");
  if (numel == 0) return VLLM_AMD_OK;
  hipStream_t s = (hipStream_t)stream;
  const void* src = inp;
  if (reg_buffer) {   // eager mode: stage the input in the pre-registered buffer (custom_all_reduce.cu:78-84)
    VA_REQUIRE(bytes <= reg_buffer_bytes, VLLM_AMD_EINVAL, "ar_all_reduce: input larger than the registered buffer");
    hipError_t e = hipMemcpyAsync(reg_buffer, inp, (size_t)bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return hip_fail(e, "ar_all_reduce: hipMemcpyAsync");
    src = reg_buffer;
  }
  ArPeers bufs;
  {
    std::lock_guard<std::mutex> lk(st->mu);
    auto it = st->buffers.find(src);
    VA_REQUIRE(it != st->buffers.end(), VLLM_AMD_EINVAL, "buffer address %p is not registered!", src);
    bufs = it->second;
  }
  const int64_t nvec = bytes / 16;
  if (dtype == VLLM_AMD_F32) ar_launch<float>(st, bufs, out, nvec, bytes, s);
  else if (dtype == VLLM_AMD_F16) ar_launch<f16_t>(st, bufs, out, nvec, bytes, s);
  else ar_launch<bf16_t>(st, bufs, out, nvec, bytes, s);
  VA_LAUNCH_CHECK("custom all_reduce launch");
  return VLLM_AMD_OK;
}

// 1 in *timed_out if any barrier of this rank gave up waiting since init (synchronises the device).
extern "C" int vllm_amd_ar_error(void* handle, int* timed_out) {
  VA_REQUIRE(handle && timed_out, VLLM_AMD_EINVAL, "ar_error: null pointer");
  ArState* st = (ArState*)handle;
  uint32_t v = 0;
  hipError_t e = hipMemcpy(&v, &((ArSignal*)st->sg.p[st->rank])->timed_out, sizeof(v), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "ar_error: hipMemcpy");
  *timed_out = (int)v;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_dispose(void* handle) {
  delete (ArState*)handle;
  return VLLM_AMD_OK;
}

// ---- shared buffers for the Python side (custom_all_reduce.cu:146-189) ----------------------
// Uncached device memory (the flag words must not sit in a cache), zeroed, plus its 64-byte IPC
// handle.  These three calls allocate / map / free by design; nothing else in the library does.
extern "C" int vllm_amd_ar_alloc_shared(int64_t bytes, void** ptr, void* ipc_handle) {
  VA_REQUIRE(bytes > 0 && ptr && ipc_handle, VLLM_AMD_EINVAL, "ar_alloc_shared: bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the IPC handle travels as 64 bytes");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return hip_fail(e, "ar_alloc_shared: hipExtMallocWithFlags");
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t*)ipc_handle, p);
  if (e != hipSuccess) { (void)hipFree(p); return hip_fail(e, "ar_alloc_shared"); }
  *ptr = p;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_open_handle(const void* ipc_handle, void** ptr) {
  VA_REQUIRE(ipc_handle && ptr, VLLM_AMD_EINVAL, "ar_open_handle: null pointer");
  hipError_t e = hipIpcOpenMemHandle(ptr, *(const hipIpcMemHandle_t*)ipc_handle, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return hip_fail(e, "ar_open_handle: hipIpcOpenMemHandle");
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_close_handle(void* ptr) {
  hipError_t e = hipIpcCloseMemHandle(ptr);
  if (e != hipSuccess) return hip_fail(e, "ar_close_handle: hipIpcCloseMemHandle");
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_free_shared(void* ptr) {
  hipError_t e = hipFree(ptr);
  if (e != hipSuccess) return hip_fail(e, "ar_free_shared: hipFree");
  return VLLM_AMD_OK;
}
