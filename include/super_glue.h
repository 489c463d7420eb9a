// super_glue.h -- drop-in replacement of UR-MVO include/super_glue.h:20-33,69.
#ifndef SUPER_GLUE_H_
#define SUPER_GLUE_H_

#include <cstdio>
#include <memory>
#include <string>

#if __has_include(<Eigen/Core>) && __has_include(<opencv2/opencv.hpp>)
#include <Eigen/Core>
#include <opencv2/opencv.hpp>
#else
#include "urf_compat.h"
#endif
#if __has_include("read_configs.h")
#include "read_configs.h"
#endif
#include "urf.h"

#include "urf_shim.h"

class SuperGlue {
 public:
  explicit SuperGlue(const SuperGlueConfig &superglue_config) : superglue_config_(superglue_config) {
    const urf_shim::engine_options o = urf_shim::parse_engine_file(superglue_config.engine_file, &engine_path_);
    precision_ = o.precision;
    outlier_stage_ = o.outlier_stage;
    calibrate_pairs_ = o.calibrate_pairs;
  }
  // Not in the reference: the precision mode of the handle build() creates (urf_shim.h; default = strict parity)
  void set_precision(int precision) { precision_ = precision; }
  int precision() const { return precision_; }
  // Not in the reference: the outlier stage of PointMatching::MatchingPoints (before build(); or "#outlier=opencv42" behind
  // engine_file): 0 = the in-tree 8-point search with the reference call's 3 px / 0.99 (default), 1 = that call itself,
  // cv::findFundamentalMat(FM_RANSAC) of OpenCV 4.2 restated (urf_sg_config.outlier_stage)
  void set_outlier_stage(int stage) { outlier_stage_ = stage; }
  int outlier_stage() const { return outlier_stage_; }
  // Not in the reference: pairs the handle measures its own error on after build() (urf_sg_config.calibrate_pairs; 0 = never)
  void set_calibrate_pairs(int n) { calibrate_pairs_ = n; }
  ~SuperGlue() { urf_pm_destroy(h_); }
  SuperGlue(const SuperGlue &) = delete;
  SuperGlue &operator=(const SuperGlue &) = delete;

  bool build() {  // src/super_glue.cpp:21-147
    if (h_) return true;
    if (!create()) return false;
    // deserialize_engine() when engine_file exists, else onnx_file -> build -> save_engine(), like the reference
    if (urf_pm_build_config(h_, engine_path_.c_str(), superglue_config_.onnx_file.c_str()) != 0) {
      report("build");
      urf_pm_destroy(h_); h_ = nullptr;
      return false;
    }
    return true;
  }
  bool build(const float *blob, size_t n_floats) {
    if (!create()) return false;
    if (urf_pm_build(h_, blob, n_floats) != 0) { report("build"); urf_pm_destroy(h_); h_ = nullptr; return false; }
    return true;
  }

  // src/super_glue.cpp:166-241; features carry NORMALISED keypoints
  bool infer(const Eigen::Matrix<double, 259, Eigen::Dynamic> &features0,
             const Eigen::Matrix<double, 259, Eigen::Dynamic> &features1, Eigen::VectorXi &indices0,
             Eigen::VectorXi &indices1, Eigen::VectorXd &mscores0, Eigen::VectorXd &mscores1) {
    if (!h_) return false;
    const int n0 = (int)features0.cols(), n1 = (int)features1.cols();
    Eigen::VectorXi i0, i1;
    Eigen::VectorXd m0, m1;
    i0.resize(n0); i1.resize(n1); m0.resize(n0); m1.resize(n1);
    if (urf_sg_infer(h_, features0.data(), n0, features1.data(), n1, i0.data(), i1.data(), m0.data(), m1.data(),
                     nullptr) != 0) {
      report("infer");
      return false;  // outputs untouched, like the reference on failure
    }
    indices0 = i0; indices1 = i1; mscores0 = m0; mscores1 = m1;
    return true;
  }

  void save_engine() {}  // (build() has written the engine_file cache already when it had to start from onnx_file)
  bool deserialize_engine() {
    if (!h_) return false;
    if (urf_pm_build_file(h_, engine_path_.c_str()) != 0) { report("deserialize_engine"); return false; }
    return true;
  }
  urf_pm *handle() { return h_; }

 private:
  bool create() {
    urf_sg_config c{};
    c.image_width = superglue_config_.image_width;
    c.image_height = superglue_config_.image_height;
    c.matching_threshold = superglue_config_.matching_threshold;
    c.precision = precision_;
    c.outlier_stage = outlier_stage_;
    c.calibrate_pairs = calibrate_pairs_ > 0 ? calibrate_pairs_ : -1;
    if (urf_pm_create(&c, &h_) != 0) { report("create"); return false; }
    return true;
  }
  void report(const char *what) const { std::fprintf(stderr, "SuperGlue::%s: %s\n", what, urf_last_error()); }
  SuperGlueConfig superglue_config_;
  std::string engine_path_;      // engine_file without the "#precision=N" suffix
  int precision_ = URF_SHIM_PRECISION;
  int outlier_stage_ = 0;
  int calibrate_pairs_ = 8;
  urf_pm *h_ = nullptr;
};

typedef std::shared_ptr<SuperGlue> SuperGluePtr;

#endif  // SUPER_GLUE_H_
