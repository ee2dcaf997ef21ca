// te_api.hip -- version / status / device probe of libte_relprop.
#include "te_common.h"

#include <string.h>

extern "C" int te_version(void) { return 600; /* 0.6.0: round 6 -- te_build_id() */ }

#ifndef TE_BUILD_ID
#define TE_BUILD_ID "unstamped"      // a build that did not go through build.py: _lib.load() refuses it
#endif
extern "C" const char* te_build_id(void) { return TE_BUILD_ID; }

extern "C" int te_x6_study_build(void) {
  int bits = 0;
#ifdef TE_X6_STUDY
  bits |= 1;      // x6 study schedules (TE_X6_STAGES_3 / TE_X6_KSPLIT) and main-loop ablations compiled in
#endif
#ifdef TE_STUDY
  bits |= 2;      // getenv switches and study variants of the attention / fp32-MFMA / GELU-plane kernels compiled in
#endif
  return bits;
}

extern "C" const char* te_status_string(int status) {
  switch (status) {
    case TE_OK: return "ok";
    case TE_ERR_INVALID_ARG: return "invalid argument";
    case TE_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
    case TE_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case TE_ERR_NO_DEVICE: return "no gfx950 device visible";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "unknown status";
}

extern "C" int te_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TE_ERR_NO_DEVICE;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) continue;
    if (strncmp(p.gcnArchName, "gfx950", 6) == 0) return TE_OK;
  }
  return TE_ERR_NO_DEVICE;
}
