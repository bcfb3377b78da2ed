// Error reporting shared by all entry points of libdeepcam_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/deepcam_hip.h"

static thread_local char g_err[512] = "";

extern "C" const char* dc_last_error(void) { return g_err; }
extern "C" int dc_version(void) { return 1; }

extern "C" int dc_set_error(int code, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", code, hipGetErrorString((hipError_t)code), file, line);
  return code ? code : -1;
}

extern "C" int dc_fail(const char* msg, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "%s (%s:%d)", msg, file ? file : "?", line);
  return -2;
}

extern "C" int dc_check_view(const void* ptr, int ld, int c, int dtype, const char* what) {
  const int es = dtype == DC_BF16 ? 2 : 4;
  const int kpv = 16 / es;
  const char* why = nullptr;
  if (ptr == nullptr) why = "null pointer";
  else if (((uintptr_t)ptr & 15) != 0) why = "pointer not 16-byte aligned";
  else if (c <= 0 || c % kpv != 0) why = "channel count not a positive multiple of the 16-byte vector";
  else if (ld < c || ld % kpv != 0) why = "pixel stride (ld) smaller than C or not vector aligned";
  if (why == nullptr) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s (ld=%d, C=%d, dtype=%d)", what, why, ld, c, dtype);
  return -3;
}
