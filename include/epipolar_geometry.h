// epipolar_geometry.h -- drop-in replacement of UR-MVO
// include/epipolar_geometry.h:9-40: class EpipolarGeometry with the reference's
// constructor and reconstruct() signature, on top of urf_epipolar_reconstruct()
// (include/urf.h).  The GPU work runs on the device/stream of the PointMatching
// handle given to attach(); the reference's process-global srand(0)
// (src/epipolar_geometry.cc:100-112) becomes the explicit `seed` member.
#ifndef EPIPOLAR_GEOMETRY_H
#define EPIPOLAR_GEOMETRY_H

#include <vector>

#if __has_include(<Eigen/Core>) && __has_include(<opencv2/opencv.hpp>)
#include <Eigen/Core>
#include <opencv2/opencv.hpp>
#else
#include "urf_compat.h"
#endif
#include "urf.h"

class EpipolarGeometry {
 public:
  EpipolarGeometry(const Eigen::Matrix3f &k, float sigma = 1.0, int iterations = 200) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) cfg_.K[r * 3 + c] = k(r, c);
    cfg_.sigma = sigma;
    cfg_.iterations = iterations;
    cfg_.seed = 0;
  }
  void attach(urf_pm *matcher) { h_ = matcher; }   // e.g. SuperGlue::handle()
  void seed(uint32_t s) { cfg_.seed = s; }

  bool reconstruct(const std::vector<cv::KeyPoint> &vKeys1, const std::vector<cv::KeyPoint> &vKeys2,
                   const std::vector<int> vMatches12, Eigen::Matrix4f &T21, std::vector<cv::Point3f> &vP3D,
                   std::vector<bool> &vbTriangulated) {
    if (!h_) return false;
    const int n1 = (int)vKeys1.size(), n2 = (int)vKeys2.size();
    std::vector<float> k1(2 * (size_t)n1), k2(2 * (size_t)n2), P(3 * (size_t)(n1 ? n1 : 1));
    for (int i = 0; i < n1; ++i) { k1[2 * i] = vKeys1[i].pt.x; k1[2 * i + 1] = vKeys1[i].pt.y; }
    for (int i = 0; i < n2; ++i) { k2[2 * i] = vKeys2[i].pt.x; k2[2 * i + 1] = vKeys2[i].pt.y; }
    std::vector<uint8_t> tri(n1 ? n1 : 1);
    float T[16], sc[2];
    int model = -1;
    const int rc = urf_epipolar_reconstruct(h_, &cfg_, k1.data(), n1, k2.data(), n2, vMatches12.data(), T, P.data(),
                                            tri.data(), &model, sc);
    if (rc != 1) return false;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) T21(r, c) = T[r * 4 + c];
    vP3D.resize(n1);
    vbTriangulated.assign(n1, false);
    for (int i = 0; i < n1; ++i) {
      vP3D[i] = cv::Point3f(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
      vbTriangulated[i] = tri[i] != 0;
    }
    return true;
  }

 private:
  urf_epi_config cfg_{};
  urf_pm *h_ = nullptr;
};

#endif  // EPIPOLAR_GEOMETRY_H
