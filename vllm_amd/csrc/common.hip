// Host-side error plumbing of the C ABI (include/vllm_amd.h).
#include "common.cuh"

#include <stdarg.h>
#include <stdio.h>

#include <mutex>

namespace va {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return VLLM_AMD_EHIP;
}

struct Workspace { char* base; size_t slot_bytes; hipStream_t streams[WS_SLOTS]; int nstreams; };
static Workspace g_ws[64];
static std::mutex g_ws_mu;

bool workspace_slot(hipStream_t st, WsSlot* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  Workspace& w = g_ws[dev];
  if (!w.base) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;
  }
  int slot = -1;
  for (int i = 0; i < w.nstreams; ++i)
    if (w.streams[i] == st) slot = i;
  if (slot < 0) {
    if (w.nstreams == WS_SLOTS) return false;
    slot = w.nstreams++;
    w.streams[slot] = st;
  }
  char* ctr = w.base + (size_t)slot * WS_CTR_BYTES;
  out->ctr_mm = (unsigned*)ctr;
  out->ctr_pa = (unsigned*)ctr + WS_CTR_MM_WORDS;
  out->data = w.base + WS_SLOTS * WS_CTR_BYTES + (size_t)slot * w.slot_bytes;
  out->data_bytes = w.slot_bytes;
  return true;
}
}  // namespace va

// Registers (ws != NULL) or removes (ws == NULL) the calling device's workspace: `bytes` of
// caller-owned device memory, 256-byte aligned, valid until replaced or removed.  The ticket
// counters at its start are zeroed on `stream` (asynchronously: order the first op behind it).
// Replacing a workspace while launches that use it are in flight is the caller's race.
extern "C" int vllm_amd_set_workspace(void* ws, int64_t bytes, void* stream) {
  using namespace va;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    set_error("set_workspace: no current device");
    return VLLM_AMD_EHIP;
  }
  if (ws == nullptr) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    g_ws[dev] = Workspace{};
    return VLLM_AMD_OK;
  }
  const int64_t min_bytes = (int64_t)(WS_SLOTS * WS_CTR_BYTES + WS_SLOTS * 256);
  VA_REQUIRE((((uintptr_t)ws) & 255) == 0, VLLM_AMD_EINVAL,
             "set_workspace: the workspace must be 256-byte aligned");
  VA_REQUIRE(bytes >= min_bytes, VLLM_AMD_EINVAL,
             "set_workspace: workspace of %lld bytes is too small (minimum %lld)", (long long)bytes,
             (long long)min_bytes);
  hipError_t e = hipMemsetAsync(ws, 0, WS_SLOTS * WS_CTR_BYTES, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "set_workspace");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  Workspace w{};
  w.base = (char*)ws;
  w.slot_bytes = (((size_t)bytes - WS_SLOTS * WS_CTR_BYTES) / WS_SLOTS) & ~(size_t)255;
  g_ws[dev] = w;
  return VLLM_AMD_OK;
}

extern "C" const char* vllm_amd_last_error(void) { return va::g_err; }
extern "C" int vllm_amd_version(void) { return 100; }
