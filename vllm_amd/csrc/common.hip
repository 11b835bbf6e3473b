// Host-side error plumbing of the C ABI (include/vllm_amd.h).
#include "common.cuh"

#include <stdarg.h>
#include <stdio.h>

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
}  // namespace va

extern "C" const char* vllm_amd_last_error(void) { return va::g_err; }
extern "C" int vllm_amd_version(void) { return 100; }
