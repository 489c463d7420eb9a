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
// What a deployment can say in `engine_file` behind the file name, any number of "#key=value" suffixes in any order:
//   #precision=N        N in 0..3 (above)
//   #outlier=opencv42   PointMatching / SuperGlue only: the outlier stage of MatchingPoints is the reference's own call,
//                       cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask) as OpenCV 4.2 publishes it
//                       (urf_sg_config.outlier_stage = 1; a restatement that no OpenCV binary has checked);
//                       #outlier=8point (the default): the in-tree 8-point search with the same 3 px / 0.99
//   #calibrate=N        PointMatching / SuperGlue only, strict or guarded mode: measure the matcher's own error on the first N
//                       pairs the handle sees (0 = never; default 8) and widen the guard's margin where these weights need it
// A suffix that is not understood is left in the path (the open then fails loudly with the whole string in the message).
struct engine_options {
  int precision = URF_SHIM_PRECISION;
  int outlier_stage = 0;
  int calibrate_pairs = 8;
};
inline engine_options parse_engine_file(const std::string &engine_file, std::string *path) {
  engine_options o;
  std::string rest = engine_file;
  for (;;) {
    const size_t at = rest.rfind('#');
    if (at == std::string::npos) break;
    const std::string kv = rest.substr(at + 1);
    bool ok = true;
    if (kv.size() == 11 && kv.compare(0, 10, "precision=") == 0 && kv[10] >= '0' && kv[10] <= '3') o.precision = kv[10] - '0';
    else if (kv == "outlier=opencv42") o.outlier_stage = 1;
    else if (kv == "outlier=8point") o.outlier_stage = 0;
    else if (kv.compare(0, 10, "calibrate=") == 0 && kv.size() > 10 && kv.size() <= 13 &&
             kv.find_first_not_of("0123456789", 10) == std::string::npos) o.calibrate_pairs = std::atoi(kv.c_str() + 10);
    else ok = false;
    if (!ok) break;
    rest.erase(at);
  }
  *path = rest;
  return o;
}
// "weights.urfw#precision=2" -> path "weights.urfw", returns 2; no (or a malformed) suffix -> the whole string, URF_SHIM_PRECISION
inline int split_engine_file(const std::string &engine_file, std::string *path) {
  return parse_engine_file(engine_file, path).precision;
}
}  // namespace urf_shim

#endif  // URF_SHIM_H_
