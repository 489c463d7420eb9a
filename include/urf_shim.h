// urf_shim.h -- what the drop-in headers (super_point.h, super_glue.h, point_matching.h) share.
//
// Precision mode of the handles the shims create.  The reference's config structs (include/read_configs.h:9-29 of UR-MVO)
// have no such field, so the mode is chosen, in this order, by
//   1. a "#precision=N" suffix of `engine_file` -- the knob a deployment reaches from its YAML without touching code:
//        engine_file: "superpoint_v1.urfw#precision=2"
//      (the suffix is cut off before the file is opened);
//   2. set_precision(N) on the object before build() (not in the reference API);
//   3. the compile-time default URF_SHIM_PRECISION below.
// Never by the environment: what a handle guarantees must be visible where it is configured.
//
//   3 = strict parity (the default): keypoints, their order, descriptors and the match index lists are the exact mode's /
//       the CPU oracle's; SuperPoint runs in exact fp32, the matcher on the f16 matrix core with every pair whose decisions
//       sit within its error redone in exact arithmetic inside the library;
//   0 = exact fp32 everywhere (every tensor bit-identical to the oracle, a third of the throughput);
//   2 = guarded fast (the keypoint SET of every frame is the exact mode's, near-tied pairs are flagged, not redone;
//       DESIGN.md section 11) -- after build(), call SuperPoint::calibrate_guard() on a few of the deployment's own frames
//       when the weights are not the ones the built-in error model was measured on;
//   1 = fast without any guard.
#ifndef URF_SHIM_H_
#define URF_SHIM_H_

#include <cstdlib>
#include <string>

#ifndef URF_SHIM_PRECISION
#define URF_SHIM_PRECISION 3
#endif

namespace urf_shim {
// "weights.urfw#precision=2" -> path "weights.urfw", returns 2; no (or a malformed) suffix -> the whole string, URF_SHIM_PRECISION
inline int split_engine_file(const std::string &engine_file, std::string *path) {
  static const char key[] = "#precision=";
  const size_t at = engine_file.rfind(key);
  if (at == std::string::npos) { *path = engine_file; return URF_SHIM_PRECISION; }
  const std::string v = engine_file.substr(at + sizeof(key) - 1);
  if (v.size() != 1 || v[0] < '0' || v[0] > '3') { *path = engine_file; return URF_SHIM_PRECISION; }
  *path = engine_file.substr(0, at);
  return v[0] - '0';
}
}  // namespace urf_shim

#endif  // URF_SHIM_H_
