// super_point.h -- drop-in replacement of UR-MVO include/super_point.h:20-33,80.
// Same class, same public signatures; the TensorRT members of the reference
// (include/super_point.h:36-40) are replaced by an opaque handle of the C ABI
// (include/urf.h).  Header-only; link with liburf_front.so.
#ifndef SUPER_POINT_H_
#define SUPER_POINT_H_

#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#if __has_include(<Eigen/Core>) && __has_include(<opencv2/opencv.hpp>)
#include <Eigen/Core>
#include <opencv2/opencv.hpp>
#define URF_HAVE_CV 1
#else
#include "urf_compat.h"
#endif
#if __has_include("read_configs.h")
#include "read_configs.h"
#endif
#include "urf.h"

#include "urf_shim.h"

class SuperPoint {
 public:
  explicit SuperPoint(const SuperPointConfig &super_point_config) : super_point_config_(super_point_config) {
    precision_ = urf_shim::split_engine_file(super_point_config.engine_file, &engine_path_);
  }
  // Not in the reference: the precision mode of the handle build() creates (urf_shim.h; default = strict parity)
  void set_precision(int precision) { precision_ = precision; }
  int precision() const { return precision_; }
  ~SuperPoint() { urf_sp_destroy(h_); }
  SuperPoint(const SuperPoint &) = delete;
  SuperPoint &operator=(const SuperPoint &) = delete;

  // build(): src/super_point.cpp:18-102, the reference's flow: deserialize_engine() when engine_file exists; otherwise the
  // initialisers of onnx_file are read (urf_sp_build_config: no ONNX library involved), the handle is built from them and the
  // engine_file cache is written -- the next start deserialises.
  bool build() {
    if (h_) return true;
    urf_sp_config c{};
    c.max_keypoints = super_point_config_.max_keypoints;
    c.keypoint_threshold = super_point_config_.keypoint_threshold;
    c.remove_borders = super_point_config_.remove_borders;
    c.max_height = 1500; c.max_width = 1500;  // TensorRT profile maximum, :55-60
    c.max_batch = 1; c.device = 0; c.precision = precision_;
    if (urf_sp_create(&c, &h_) != 0) { report("create"); return false; }
    if (urf_sp_build_config(h_, engine_path_.c_str(), super_point_config_.onnx_file.c_str()) != 0) {
      report("build");
      urf_sp_destroy(h_); h_ = nullptr;
      return false;
    }
    return true;
  }
  // build from an in-memory container (tests, synthetic weights)
  bool build(const float *blob, size_t n_floats, int max_h = 1500, int max_w = 1500) {
    urf_sp_config c{};
    c.max_keypoints = super_point_config_.max_keypoints;
    c.keypoint_threshold = super_point_config_.keypoint_threshold;
    c.remove_borders = super_point_config_.remove_borders;
    c.max_height = max_h; c.max_width = max_w; c.max_batch = 1; c.device = 0; c.precision = precision_;
    if (urf_sp_create(&c, &h_) != 0) { report("create"); return false; }
    if (urf_sp_build(h_, blob, n_floats) != 0) { report("build"); urf_sp_destroy(h_); h_ = nullptr; return false; }
    return true;
  }

  // Not in the reference: the guarded fast mode's error model checked against the exact mode on one of the
  // deployment's own frames and widened if that frame needs it (urf_sp_calibrate_guard; a no-op in the other modes).  Call it
  // after build() with a few representative frames when the weights are not the ones the built-in constants were measured on.
  bool calibrate_guard(const cv::Mat &image) {
    if (!h_) return false;
    if (precision_ != 2) return true;
    const uint8_t *p = image.data;
    if (urf_sp_calibrate_guard(h_, 1, &p, image.rows, image.cols, (size_t)image.step, nullptr) != 0) { report("calibrate_guard"); return false; }
    return true;
  }

  // infer(): src/super_point.cpp:121-156.  features is resized by the callee.
  bool infer(const cv::Mat &image, const cv::Mat &mask, Eigen::Matrix<double, 259, Eigen::Dynamic> &features) {
    if (!h_) return false;
    if ((int)buf_.size() < 259 * URF_MAX_KEYPOINTS) buf_.resize((size_t)259 * URF_MAX_KEYPOINTS);
    int K = 0;
    const int rc = urf_sp_infer(h_, image.data, image.rows, image.cols, (size_t)image.step,
                                mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (size_t)mask.step,
                                buf_.data(), URF_MAX_KEYPOINTS, &K);
    if (rc != 0) { report("infer"); return false; }
    features.resize(259, K);
    std::copy(buf_.begin(), buf_.begin() + (size_t)259 * K, features.data());
    keypoints_.resize(K);
    for (int j = 0; j < K; ++j) keypoints_[j] = {(int)buf_[(size_t)259 * j + 1], (int)buf_[(size_t)259 * j + 2]};
    return true;
  }

  // visualization(): src/super_point.cpp:388-400 (needs OpenCV drawing)
  void visualization(const std::string &image_name, const cv::Mat &image) {
#ifdef URF_HAVE_CV
    cv::Mat image_display;
    if (image.channels() == 1) cv::cvtColor(image, image_display, cv::COLOR_GRAY2BGR);
    else image_display = image.clone();
    for (auto &keypoint : keypoints_)
      cv::circle(image_display, cv::Point(keypoint[0], keypoint[1]), 1, cv::Scalar(255, 0, 0), -1, 16);
    cv::imwrite(image_name + ".jpg", image_display);
#else
    // no OpenCV in the build: the same picture as a binary PPM (image_name + ".ppm"), grey image replicated to RGB,
    // every keypoint a radius-1 disc (centre + 4 neighbours) in the reference's colour, BGR (255, 0, 0) = blue
    if (image.empty()) return;
    std::vector<unsigned char> rgb((size_t)image.rows * image.cols * 3);
    for (int y = 0; y < image.rows; ++y)
      for (int x = 0; x < image.cols; ++x) {
        const unsigned char v = image.data[(size_t)y * image.step + x];
        unsigned char *px = &rgb[((size_t)y * image.cols + x) * 3];
        px[0] = px[1] = px[2] = v;
      }
    static const int disc[5][2] = {{0, 0}, {1, 0}, {-1, 0}, {0, 1}, {0, -1}};
    for (auto &keypoint : keypoints_)
      for (auto &d : disc) {
        const int x = keypoint[0] + d[0], y = keypoint[1] + d[1];
        if (x < 0 || y < 0 || x >= image.cols || y >= image.rows) continue;
        unsigned char *px = &rgb[((size_t)y * image.cols + x) * 3];
        px[0] = 0; px[1] = 0; px[2] = 255;
      }
    if (FILE *f = std::fopen((image_name + ".ppm").c_str(), "wb")) {
      std::fprintf(f, "P6\n%d %d\n255\n", image.cols, image.rows);
      std::fwrite(rgb.data(), 1, rgb.size(), f);
      std::fclose(f);
    }
#endif
  }

  void save_engine() {}  // (build() has written the engine_file cache already when it had to start from onnx_file)
  bool deserialize_engine() {
    if (!h_) return false;
    if (urf_sp_build_file(h_, engine_path_.c_str()) != 0) { report("deserialize_engine"); return false; }
    return true;
  }

 private:
  void report(const char *what) const { std::fprintf(stderr, "SuperPoint::%s: %s\n", what, urf_last_error()); }
  SuperPointConfig super_point_config_;
  std::string engine_path_;      // engine_file without the "#precision=N" suffix
  int precision_ = URF_SHIM_PRECISION;
  urf_sp *h_ = nullptr;
  std::vector<double> buf_;
  std::vector<std::vector<int>> keypoints_;
};

typedef std::shared_ptr<SuperPoint> SuperPointPtr;

#endif  // SUPER_POINT_H_
