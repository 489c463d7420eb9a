// build_info.hip -- what this binary is.  The Makefile recompiles this file whenever any object of the library changed, so the
// time in the string is the time of the link (bench.py prints it: a stale library is visible in the line).
#include <stdio.h>

namespace urf {
double build_guard_delta();
double build_guard_ulps();
}  // namespace urf

extern "C" const char *urf_build_info(void) {
  static char info[200];
#ifdef URF_EXPERIMENTS
  const char *kind = "; EXPERIMENTS build (URF_* environment knobs are read)";
#else
  const char *kind = "";
#endif
  snprintf(info, sizeof(info), "liburf_front built %s %s; guard SuperPoint delta %.3g ulps %.3g%s", __DATE__, __TIME__,
           urf::build_guard_delta(), urf::build_guard_ulps(), kind);
  return info;
}
