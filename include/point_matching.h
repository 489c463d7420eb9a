// point_matching.h -- drop-in replacement of UR-MVO include/point_matching.h:7-25.
#ifndef POINT_MATCHING_H_
#define POINT_MATCHING_H_

#include <iostream>
#include <vector>

#include "super_glue.h"

class PointMatching {
 public:
  // src/point_matching.cc:6-12: a failed SuperGlue build is only reported
  PointMatching(SuperGlueConfig &superglue_config) : superglue(superglue_config) {
    _superglue_config = superglue_config;
    // (a configuration that names neither file is how the tests say "weights follow through build(blob, n)": nothing to build yet)
    if ((!superglue_config.engine_file.empty() || !superglue_config.onnx_file.empty()) && !superglue.build())
      std::cout << "Erron in superglue building" << std::endl;
  }
  bool build(const float *blob, size_t n_floats) { return superglue.build(blob, n_floats); }
  // Not in the reference (before build(blob, n); a configuration that names its files says it with "#outlier=opencv42" behind
  // engine_file, include/urf_shim.h): which outlier stage MatchingPoints(..., true) runs -- 0 = the in-tree 8-point search with
  // the reference call's 3 px / 0.99 (default), 1 = cv::findFundamentalMat(FM_RANSAC, 3, 0.99) of OpenCV 4.2 restated
  void set_outlier_stage(int stage) { superglue.set_outlier_stage(stage); }
  int outlier_stage() const { return superglue.outlier_stage(); }
  // Not in the reference: the guard word of the pair the last MatchingPoints call handled (urf_pm_near_tie_flags): 0 = its
  // match list is the exact pipeline's; non-zero in the guarded fast mode (precision 2) = a decisive entry sat within the
  // fast pipeline's error of its alternative and the list may differ there (in the strict mode, the default, such a pair
  // was redone in exact arithmetic before the call returned: the word only says that this happened)
  int last_near_tie_flags() {
    int f = 0;
    if (!superglue.handle() || urf_pm_near_tie_flags(superglue.handle(), &f, 1) != 0) return 0;
    return f;
  }

  // src/point_matching.cc:14-61
  int MatchingPoints(const Eigen::Matrix<double, 259, Eigen::Dynamic> &features0,
                     const Eigen::Matrix<double, 259, Eigen::Dynamic> &features1, std::vector<cv::DMatch> &matches,
                     bool outlier_rejection = false) {
    matches.clear();
    if (!superglue.handle()) return 0;
    std::vector<urf_dmatch> out(URF_MAX_KEYPOINTS);
    const int n = urf_match(superglue.handle(), features0.data(), (int)features0.cols(), features1.data(),
                            (int)features1.cols(), outlier_rejection ? 1 : 0, out.data(), URF_MAX_KEYPOINTS);
    if (n < 0) { std::cout << "PointMatching: " << urf_last_error() << std::endl; return 0; }
    for (int i = 0; i < n; ++i) matches.emplace_back(out[i].queryIdx, out[i].trainIdx, out[i].distance);
    return (int)matches.size();
  }

  // src/point_matching.cc:63-76
  Eigen::Matrix<double, 259, Eigen::Dynamic> NormalizeKeypoints(
      const Eigen::Matrix<double, 259, Eigen::Dynamic> &features, int width, int height) {
    Eigen::Matrix<double, 259, Eigen::Dynamic> norm_features;
    norm_features.resize(259, features.cols());
    urf_normalize_keypoints(features.data(), (int)features.cols(), width, height, norm_features.data());
    return norm_features;
  }

 private:
  SuperGlue superglue;
  SuperGlueConfig _superglue_config;
};

typedef std::shared_ptr<PointMatching> PointMatchingPtr;

#endif  // POINT_MATCHING_H_
