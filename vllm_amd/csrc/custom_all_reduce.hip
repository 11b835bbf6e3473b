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
// launch number into peer p's `start[b][r]` and spins on its own `start[b][p]` until that word has
// reached the launch number (wrap-safe >=: a rank that arrives late catches up instead of missing
// the word its peer has already moved on from).  Every spin is BOUNDED in wall-clock time
// (s_memrealtime; 30 s unless vllm_amd_ar_set_timeout_ms says otherwise -- generous: a peer may be
// busy with host work, a graph capture, a start-up) so that a dead peer can never hang the GPU, and a
// time-out is LOUD: the block overwrites its part of `out` with NaN, and a sticky word in pinned host
// memory makes every later vllm_amd_ar_all_reduce of the handle return an error without synchronising.
//
// Nothing is allocated inside all_reduce; buffers come from the caller (the helpers at the end
// wrap hipExtMallocWithFlags / hipIpc* for the Python side, as the reference's
// allocate_shared_buffer_and_handle / open_mem_handle do).
//
// HIP graphs: an input met under stream capture needs no copy into a pre-registered buffer -- the
// captured kernel points at a tuple of the caller's rank-data table that is filled after the capture
// (vllm_amd_ar_get_graph_buffer_ipc_meta / vllm_amd_ar_register_graph_buffers, as the reference's
// custom_all_reduce.cu:125-146).
//
// STATUS: logic-tested with simulated ranks (one process, one stream per rank, one GPU) and with
// two processes sharing one GPU through IPC handles; never run across GPUs (the development box
// has one).  Off by default: tensor_model_parallel_all_reduce uses RCCL unless asked otherwise;
// `bench.py --gpus N` runs it beside RCCL, result-checked first.
#include "common.cuh"

#include <string.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace va {

constexpr int AR_MAX_RANKS = 8;
constexpr int AR_MAX_BLOCKS = 64;
constexpr int AR_THREADS = 512;
constexpr int64_t kDefaultTimeoutMs = 30000;

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

struct ArCtl {
  unsigned long long timeout_ticks;   // 100 MHz ticks (s_memrealtime)
  uint32_t* host_err;                 // pinned host word, sticky
};

// All NG ranks' block b meet.  WHICH selects the flag array; REL: what this rank wrote before the
// barrier is visible to the peers after it (release before the flag); ACQ: what the peers wrote before
// theirs is visible here (acquire behind the spin: nothing cached of the peers' memory survives it).
// Returns nothing; a time-out sets *bad (LDS), the signal block's word and the host's.
template <int NG, int WHICH, bool REL, bool ACQ>
__device__ __forceinline__ void ar_barrier(const ArPeers& sg, ArSignal* self, int rank, uint32_t flag,
                                           const ArCtl& ctl, int* bad) {
  if constexpr (WHICH != 0) __syncthreads();   // every thread of the block is done with the phase before
  if (threadIdx.x < NG) {
    if constexpr (REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    ArSignal* peer = (ArSignal*)sg.p[threadIdx.x];
    uint32_t* theirs = WHICH == 0 ? &peer->start[blockIdx.x][rank]
                     : WHICH == 1 ? &peer->mid[blockIdx.x][rank] : &peer->end[blockIdx.x][rank];
    const uint32_t* mine = WHICH == 0 ? &self->start[blockIdx.x][threadIdx.x]
                         : WHICH == 1 ? &self->mid[blockIdx.x][threadIdx.x]
                                      : &self->end[blockIdx.x][threadIdx.x];
    st_sys(theirs, flag);
    if ((int32_t)(ld_sys(mine) - flag) < 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int32_t)(ld_sys(mine) - flag) < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > ctl.timeout_ticks) {   // the exit every wave reaches
          st_sys(&self->timed_out, 1u);
          st_sys(ctl.host_err, 1u);
          *bad = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if constexpr (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
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

// nvec = message size in 16-byte vectors; bufs_p: the tuple of the ranks' input pointers, in device
// memory (the caller's rank-data table: filled at registration, for graph-captured launches AFTER the
// capture -- vllm_amd_ar_register_graph_buffers)
template <typename T, int NG>
__global__ __launch_bounds__(AR_THREADS) void ar_one_stage_kernel(const ArPeers* __restrict__ bufs_p, ArPeers sg,
                                                                  ArSignal* self, uint4* __restrict__ out,
                                                                  int rank, int64_t nvec, ArCtl ctl) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const ArPeers bufs = *bufs_p;
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  if (ld_sys(ctl.host_err) != 0) {   // an earlier launch of this handle gave up: no further waiting (launches
    // already enqueued behind it would each sit out the time limit twice), the output says so
    for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * AR_THREADS)
      out[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    return;
  }
  // every peer's input is complete (its earlier kernels released it at their end) -- and nothing this
  // GPU may still cache of those buffers from the previous launch is served again (acquire)
  ar_barrier<NG, 0, false, true>(sg, self, rank, flag, ctl, &bad);
  for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * AR_THREADS)
    out[i] = ar_sum<T, NG>(bufs, i);
  ar_barrier<NG, 2, false, false>(sg, self, rank, flag, ctl, &bad);    // nobody overwrites an input a peer still reads
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
  if (bad) {   // a peer never arrived: this block's outputs are not a sum -- make that visible (NaN in every dtype)
    for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * AR_THREADS)
      out[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
  }
}

template <typename T, int NG>
__global__ __launch_bounds__(AR_THREADS) void ar_two_stage_kernel(const ArPeers* __restrict__ bufs_p, ArPeers sg,
                                                                  ArSignal* self, uint4* __restrict__ out,
                                                                  int rank, int64_t nvec, ArCtl ctl) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const ArPeers bufs = *bufs_p;
  const uint32_t flag = self->epoch[blockIdx.x] + 1;
  if (ld_sys(ctl.host_err) != 0) {   // see the one-stage kernel
    for (int64_t i = (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * AR_THREADS)
      out[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    return;
  }
  const int64_t part = nvec / NG;   // slice r = [r part, (r+1) part), the last one takes the rest
  auto slice_begin = [&](int r) { return (int64_t)r * part; };
  auto slice_end = [&](int r) { return r == NG - 1 ? nvec : (int64_t)(r + 1) * part; };
  // scratch of rank r: behind its signal block
  ArPeers tmp;
#pragma unroll
  for (int r = 0; r < NG; ++r) tmp.p[r] = (char*)sg.p[r] + sizeof(ArSignal);
  ar_barrier<NG, 0, false, true>(sg, self, rank, flag, ctl, &bad);
  {
    uint4* mine = (uint4*)tmp.p[rank];
    const int64_t b = slice_begin(rank), e = slice_end(rank);
    for (int64_t i = b + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < e; i += (int64_t)gridDim.x * AR_THREADS)
      mine[i - b] = ar_sum<T, NG>(bufs, i);
  }
  ar_barrier<NG, 1, true, true>(sg, self, rank, flag, ctl, &bad);     // the reduced slices are visible to the peers
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
  ar_barrier<NG, 2, false, false>(sg, self, rank, flag, ctl, &bad);
  if (threadIdx.x == 0) self->epoch[blockIdx.x] = flag;
  if (bad) {
#pragma unroll
    for (int r = 0; r < NG; ++r) {
      const int64_t b = slice_begin(r), e = slice_end(r);
      for (int64_t i = b + (int64_t)blockIdx.x * AR_THREADS + threadIdx.x; i < e; i += (int64_t)gridDim.x * AR_THREADS)
        out[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
  }
}

struct ArState {
  int world = 0, rank = 0;
  bool fully_connected = false;
  ArPeers sg;
  int64_t scratch_bytes = 0;                 // behind every signal block
  ArPeers* table = nullptr;                  // caller's rank-data table (device memory)
  int64_t table_slots = 0, table_used = 0;
  std::unordered_map<const void*, const ArPeers*> buffers;   // registered input -> its tuple in the table
  std::vector<void*> graph_unreg;            // inputs met under stream capture, tuples still to be filled
  std::unordered_map<std::string, void*> opened;             // IPC handle bytes -> mapping in this process
  uint32_t* host_err = nullptr;              // pinned, mapped: sticky time-out word
  uint32_t* host_err_dev = nullptr;
  int64_t timeout_ms = kDefaultTimeoutMs;
  std::mutex mu;
};

// IPC handles this process handed out (vllm_amd_ar_get_graph_buffer_ipc_meta) -> the allocation's base:
// ranks that live in ONE process (the single-GPU simulation of the tests) find each other's buffers
// here; hipIpcOpenMemHandle refuses a handle of the opening process itself.
static std::mutex g_ipc_mu;
static std::unordered_map<std::string, void*> g_ipc_local;

static bool ar_one_stage(const ArState* st, int64_t bytes) {
  // custom_all_reduce.cuh:561-571: two ranks always one stage; fully connected: one stage below
  // 512 KiB (<= 4 ranks) / 256 KiB (<= 8 ranks), two stages above
  return st->world == 2 || (st->world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024;
}

template <typename T>
static void ar_launch(ArState* st, const ArPeers* bufs, void* out, int64_t nvec, int64_t bytes, hipStream_t s) {
  ArSignal* self = (ArSignal*)st->sg.p[st->rank];
  int64_t blocks = (nvec + AR_THREADS - 1) / AR_THREADS;
  if (blocks > AR_MAX_BLOCKS) blocks = AR_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  const bool one = ar_one_stage(st, bytes);
  const ArCtl ctl{(unsigned long long)st->timeout_ms * 100000ull, st->host_err_dev};
#define AR_KL(NG)                                                                                   \
  do {                                                                                              \
    if (one)                                                                                        \
      hipLaunchKernelGGL((ar_one_stage_kernel<T, NG>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, \
                         s, bufs, st->sg, self, (uint4*)out, st->rank, nvec, ctl);                  \
    else                                                                                            \
      hipLaunchKernelGGL((ar_two_stage_kernel<T, NG>), dim3((unsigned)blocks), dim3(AR_THREADS), 0, \
                         s, bufs, st->sg, self, (uint4*)out, st->rank, nvec, ctl);                  \
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

extern "C" int vllm_amd_ar_init(void** handle, void* const* meta_ptrs, int64_t scratch_bytes, void* rank_data,
                                int64_t rank_data_bytes, int world_size, int rank, int fully_connected) {
  VA_REQUIRE(handle && meta_ptrs, VLLM_AMD_EINVAL, "ar_init: null pointer");
  VA_REQUIRE(world_size == 2 || world_size == 4 || world_size == 6 || world_size == 8, VLLM_AMD_EUNSUPPORTED,
             "custom allreduce only supports num gpus in (2,4,6,8). Actual num gpus = %d", world_size);
  VA_REQUIRE(world_size % 2 == 0 && rank >= 0 && rank < world_size, VLLM_AMD_EINVAL, "invalid rank passed in");
  VA_REQUIRE(world_size == 2 || fully_connected, VLLM_AMD_EUNSUPPORTED,
             "custom allreduce: more than two ranks need fully connected links");
  for (int i = 0; i < world_size; ++i) VA_REQUIRE(meta_ptrs[i] != nullptr, VLLM_AMD_EINVAL, "ar_init: null meta pointer");
  VA_REQUIRE(scratch_bytes >= 0, VLLM_AMD_EINVAL, "ar_init: negative scratch size");
  VA_REQUIRE(rank_data != nullptr && rank_data_bytes >= (int64_t)sizeof(ArPeers) && (((uintptr_t)rank_data) & 15) == 0,
             VLLM_AMD_EINVAL, "ar_init: the rank-data table needs at least %zu bytes of 16-byte aligned device memory",
             sizeof(ArPeers));
  uint32_t* herr = nullptr;
  hipError_t e = hipHostMalloc((void**)&herr, 64, hipHostMallocMapped);
  if (e != hipSuccess) return hip_fail(e, "ar_init: hipHostMalloc");
  *herr = 0;
  uint32_t* herr_dev = nullptr;
  e = hipHostGetDevicePointer((void**)&herr_dev, herr, 0);
  if (e != hipSuccess) { (void)hipHostFree(herr); return hip_fail(e, "ar_init: hipHostGetDevicePointer"); }
  ArState* st = new ArState();
  st->world = world_size; st->rank = rank; st->fully_connected = fully_connected != 0;
  for (int i = 0; i < AR_MAX_RANKS; ++i) st->sg.p[i] = i < world_size ? meta_ptrs[i] : nullptr;
  st->scratch_bytes = scratch_bytes;
  st->table = (ArPeers*)rank_data;
  st->table_slots = rank_data_bytes / (int64_t)sizeof(ArPeers);
  st->host_err = herr; st->host_err_dev = herr_dev;
  *handle = st;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_set_timeout_ms(void* handle, int64_t ms) {
  VA_REQUIRE(handle && ms > 0 && ms <= 3600000, VLLM_AMD_EINVAL, "ar_set_timeout_ms: 1 ms .. 1 h");
  ((ArState*)handle)->timeout_ms = ms;
  return VLLM_AMD_OK;
}

// one tuple into the caller's table (synchronous: registration time)
static int ar_put_tuple(ArState* st, const ArPeers& p, const ArPeers** where) {
  VA_REQUIRE(st->table_used < st->table_slots, VLLM_AMD_EUNSUPPORTED,
             "Rank data buffer is overflowed by %lld", (long long)(st->table_used + 1 - st->table_slots));
  ArPeers* slot = st->table + st->table_used;
  hipError_t e = hipMemcpy(slot, &p, sizeof(ArPeers), hipMemcpyHostToDevice);
  if (e != hipSuccess) return hip_fail(e, "custom allreduce: hipMemcpy of a pointer tuple");
  st->table_used += 1;
  *where = slot;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_register_buffer(void* handle, void* const* ptrs) {
  VA_REQUIRE(handle && ptrs, VLLM_AMD_EINVAL, "ar_register_buffer: null pointer");
  ArState* st = (ArState*)handle;
  ArPeers p;
  for (int i = 0; i < AR_MAX_RANKS; ++i) p.p[i] = i < st->world ? ptrs[i] : nullptr;
  std::lock_guard<std::mutex> lk(st->mu);
  VA_REQUIRE(st->graph_unreg.empty(), VLLM_AMD_EINVAL,
             "ar_register_buffer: graph buffers met under capture are waiting for vllm_amd_ar_register_graph_buffers");
  const ArPeers* where = nullptr;
  const int rc = ar_put_tuple(st, p, &where);
  if (rc != VLLM_AMD_OK) return rc;
  st->buffers[ptrs[st->rank]] = where;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_all_reduce(void* handle, const void* inp, void* out, int64_t numel, int dtype,
                                      void* reg_buffer, int64_t reg_buffer_bytes, void* stream) {
  VA_REQUIRE(handle && inp && out, VLLM_AMD_EINVAL, "ar_all_reduce: null pointer");
  ArState* st = (ArState*)handle;
  VA_REQUIRE(dtype == VLLM_AMD_F32 || dtype == VLLM_AMD_F16 || dtype == VLLM_AMD_BF16, VLLM_AMD_EUNSUPPORTED,
             "custom allreduce only supports float32, float16 and bfloat16");
  VA_REQUIRE(*(volatile uint32_t*)st->host_err == 0, VLLM_AMD_EHIP,
             "custom allreduce: a peer did not arrive within %lld ms in an earlier all-reduce of this handle (its "
             "output was filled with NaN); the handle is unusable", (long long)st->timeout_ms);
  const int es = dtype_size(dtype);
  const int64_t bytes = numel * es;
  VA_REQUIRE(numel >= 0 && bytes % 16 == 0, VLLM_AMD_EINVAL,
             "custom allreduce currently requires input length to be multiple of 16 bytes");
  VA_REQUIRE((((uintptr_t)inp) & 15) == 0 && (((uintptr_t)out) & 15) == 0, VLLM_AMD_EINVAL,
             "ar_all_reduce: buffers must be 16-byte aligned");
  if (numel == 0) return VLLM_AMD_OK;
  const int64_t nvec = bytes / 16;
  if (!ar_one_stage(st, bytes)) {
    // the two-stage kernel parks a rank's reduced slice (the last one is the longest) behind its signal block
    const int64_t slice = (nvec - (st->world - 1) * (nvec / st->world)) * 16;
    VA_REQUIRE(slice <= st->scratch_bytes, VLLM_AMD_EINVAL,
               "ar_all_reduce: %lld bytes need %lld bytes of scratch behind the signal block, the handle has %lld",
               (long long)bytes, (long long)slice, (long long)st->scratch_bytes);
  }
  hipStream_t s = (hipStream_t)stream;
  const void* src = inp;
  if (reg_buffer) {   // eager mode: stage the input in the pre-registered buffer (custom_all_reduce.cu:78-84)
    VA_REQUIRE(bytes <= reg_buffer_bytes, VLLM_AMD_EINVAL, "ar_all_reduce: input larger than the registered buffer");
    hipError_t e = hipMemcpyAsync(reg_buffer, inp, (size_t)bytes, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return hip_fail(e, "ar_all_reduce: hipMemcpyAsync");
    src = reg_buffer;
  }
  const ArPeers* bufs = nullptr;
  {
    std::lock_guard<std::mutex> lk(st->mu);
    auto it = st->buffers.find(src);
    if (it != st->buffers.end()) {
      bufs = it->second;
    } else {
      // custom_all_reduce.cuh:537-545: under stream capture an unregistered input gets the next free
      // tuple of the table; its contents arrive with vllm_amd_ar_register_graph_buffers, before any replay
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
      VA_REQUIRE(cs == hipStreamCaptureStatusActive, VLLM_AMD_EINVAL, "buffer address %p is not registered!", src);
      const int64_t at = st->table_used + (int64_t)st->graph_unreg.size();
      VA_REQUIRE(at < st->table_slots, VLLM_AMD_EUNSUPPORTED, "Rank data buffer is overflowed by %lld",
                 (long long)(at + 1 - st->table_slots));
      bufs = st->table + at;
      st->graph_unreg.push_back(const_cast<void*>(src));
    }
  }
  if (dtype == VLLM_AMD_F32) ar_launch<float>(st, bufs, out, nvec, bytes, s);
  else if (dtype == VLLM_AMD_F16) ar_launch<f16_t>(st, bufs, out, nvec, bytes, s);
  else ar_launch<bf16_t>(st, bufs, out, nvec, bytes, s);
  VA_LAUNCH_CHECK("custom all_reduce launch");
  return VLLM_AMD_OK;
}

// custom_all_reduce.cu:125-131, .cuh:440-458: the inputs met under capture since the last registration, as
// (IPC handle of the allocation, offset into it) pairs.  handles_out: capacity x 64 bytes.
extern "C" int vllm_amd_ar_get_graph_buffer_ipc_meta(void* handle, void* handles_out, int64_t* offsets_out,
                                                     int capacity, int* count) {
  VA_REQUIRE(handle && count, VLLM_AMD_EINVAL, "ar_get_graph_buffer_ipc_meta: null pointer");
  ArState* st = (ArState*)handle;
  std::lock_guard<std::mutex> lk(st->mu);
  const int n = (int)st->graph_unreg.size();
  *count = n;
  if (n == 0 || (!handles_out && !offsets_out)) return VLLM_AMD_OK;   // (NULL outputs: only the count)
  VA_REQUIRE(handles_out && offsets_out && capacity >= n, VLLM_AMD_EINVAL,
             "ar_get_graph_buffer_ipc_meta: %d buffers, room for %d", n, capacity);
  for (int i = 0; i < n; ++i) {
    void* ptr = st->graph_unreg[i];
    void* base = nullptr;
    // (the handle names the whole allocation: share its base, carry the offset)
    hipError_t e = hipPointerGetAttribute(&base, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)ptr);
    if (e != hipSuccess || !base) return hip_fail(e, "failed to get pointer attr");
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, base);
    if (e != hipSuccess) return hip_fail(e, "ar_get_graph_buffer_ipc_meta: hipIpcGetMemHandle");
    memcpy((char*)handles_out + (size_t)i * sizeof(h), &h, sizeof(h));
    offsets_out[i] = (char*)ptr - (char*)base;
    std::lock_guard<std::mutex> lg(g_ipc_mu);
    g_ipc_local[std::string((const char*)&h, sizeof(h))] = base;
  }
  return VLLM_AMD_OK;
}

// custom_all_reduce.cu:134-146, .cuh:489-514: handles / offsets of ALL ranks ([world][n][64] bytes,
// [world][n]) for the n inputs this rank met under capture (every rank captured the same sequence):
// opens the peers' allocations and fills the n tuples the captured kernels point at.
extern "C" int vllm_amd_ar_register_graph_buffers(void* handle, const void* handles, const int64_t* offsets, int n) {
  VA_REQUIRE(handle, VLLM_AMD_EINVAL, "ar_register_graph_buffers: null handle");
  ArState* st = (ArState*)handle;
  std::lock_guard<std::mutex> lk(st->mu);
  VA_REQUIRE(n == (int)st->graph_unreg.size(), VLLM_AMD_EINVAL,
             "ar_register_graph_buffers: %d buffers given, %zu met under capture", n, st->graph_unreg.size());
  if (n == 0) return VLLM_AMD_OK;
  VA_REQUIRE(handles && offsets, VLLM_AMD_EINVAL, "ar_register_graph_buffers: null pointer");
  VA_REQUIRE(st->table_used + n <= st->table_slots, VLLM_AMD_EUNSUPPORTED, "Rank data buffer is overflowed by %lld",
             (long long)(st->table_used + n - st->table_slots));
  std::vector<ArPeers> tuples((size_t)n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < AR_MAX_RANKS; ++j) tuples[i].p[j] = nullptr;
    for (int j = 0; j < st->world; ++j) {
      if (j == st->rank) { tuples[i].p[j] = st->graph_unreg[i]; continue; }
      const char* hb = (const char*)handles + ((size_t)j * n + i) * sizeof(hipIpcMemHandle_t);
      const std::string key(hb, sizeof(hipIpcMemHandle_t));
      void* base = nullptr;
      {
        std::lock_guard<std::mutex> lg(g_ipc_mu);
        auto it = g_ipc_local.find(key);
        if (it != g_ipc_local.end()) base = it->second;   // a rank of this very process
      }
      if (!base) {
        auto it = st->opened.find(key);
        if (it != st->opened.end()) base = it->second;
      }
      if (!base) {
        hipIpcMemHandle_t h;
        memcpy(&h, hb, sizeof(h));
        hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return hip_fail(e, "ar_register_graph_buffers: hipIpcOpenMemHandle");
        st->opened[key] = base;
      }
      tuples[i].p[j] = (char*)base + offsets[(size_t)j * n + i];
    }
  }
  hipError_t e = hipMemcpy(st->table + st->table_used, tuples.data(), sizeof(ArPeers) * (size_t)n, hipMemcpyHostToDevice);
  if (e != hipSuccess) return hip_fail(e, "ar_register_graph_buffers: hipMemcpy");
  for (int i = 0; i < n; ++i) st->buffers[st->graph_unreg[i]] = st->table + st->table_used + i;
  st->table_used += n;
  st->graph_unreg.clear();
  return VLLM_AMD_OK;
}

// 1 in *timed_out if any barrier of this rank gave up waiting since init.  Reads the pinned host word:
// no synchronisation (a time-out of a launch still in flight shows up once that launch has run).
extern "C" int vllm_amd_ar_error(void* handle, int* timed_out) {
  VA_REQUIRE(handle && timed_out, VLLM_AMD_EINVAL, "ar_error: null pointer");
  ArState* st = (ArState*)handle;
  *timed_out = (int)*(volatile uint32_t*)st->host_err;
  return VLLM_AMD_OK;
}

extern "C" int vllm_amd_ar_dispose(void* handle) {
  ArState* st = (ArState*)handle;
  if (st) {
    for (auto& kv : st->opened) (void)hipIpcCloseMemHandle(kv.second);
    if (st->host_err) (void)hipHostFree(st->host_err);
  }
  delete st;
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
