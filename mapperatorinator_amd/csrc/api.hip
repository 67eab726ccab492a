// Error plumbing + version of libmapperhip's C ABI (include/mapperhip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace mh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MH_ERR_LAUNCH;
  }
  return MH_OK;
}

}  // namespace mh

extern "C" const char* mh_last_error(void) { return mh::g_err; }
extern "C" int mh_abi_version(void) { return MH_ABI_VERSION; }

extern "C" int mh_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(MhGemm);
    case 1: return (int)sizeof(MhT5Config);
    case 2: return (int)sizeof(MhT5Weights);
    case 3: return (int)sizeof(MhSampling);
    case 4: return (int)sizeof(MhDiTConfig);
    case 5: return (int)sizeof(MhDiTWeights);
  }
  return -1;
}
