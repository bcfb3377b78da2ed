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

// A non-blocking HIP stream at the lowest (level < 0), default (0) or highest (level > 0) priority the device offers.  The
// engine puts the weight-gradient work on a lowest-priority stream so that it fills the CUs the dependent backward chain
// leaves idle instead of competing with it workgroup for workgroup.
extern "C" int dc_stream_create(int level, void** stream) {
  if (stream == nullptr) return dc_fail("dc_stream_create: null argument", __FILE__, __LINE__);
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  const int prio = level < 0 ? least : (level > 0 ? greatest : 0);
  hipStream_t st = nullptr;
  e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio);
  if (e != hipSuccess) return dc_set_error(e, __FILE__, __LINE__);
  *stream = st;
  return 0;
}

extern "C" int dc_stream_destroy(void* stream) {
  if (stream == nullptr) return 0;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}

extern "C" int dc_stream_priority_range(int* least, int* greatest) {
  if (least == nullptr || greatest == nullptr) return dc_fail("dc_stream_priority_range: null argument", __FILE__, __LINE__);
  hipError_t e = hipDeviceGetStreamPriorityRange(least, greatest);
  return e == hipSuccess ? 0 : dc_set_error(e, __FILE__, __LINE__);
}
