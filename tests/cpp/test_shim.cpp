// test_shim.cpp -- the reference's C++ API (super_point.h / point_matching.h)
// driven exactly like Tracking::ExtractFeatureAndMatch (src/tracking.cc:338-377)
// on two synthetic frames; prints K and the match count so the Python test can
// compare with the ctypes path.  Usage: test_shim sp.urfw sg.urfw f0.raw f1.raw H W [vis|-] [sp.onnx sg.onnx]
// (with the two ONNX files and engine files that do not exist yet: build() starts from onnx_file and writes the caches)
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "epipolar_geometry.h"
#include "point_matching.h"
#include "super_point.h"

static std::vector<unsigned char> read_file(const char *p, size_t n) {
  std::vector<unsigned char> v(n);
  FILE *f = fopen(p, "rb");
  if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  const int H = atoi(argv[5]), W = atoi(argv[6]);
  SuperPointConfig spc{};
  spc.max_keypoints = 400; spc.keypoint_threshold = 0.0005; spc.remove_borders = 4; spc.dla_core = -1;
  spc.engine_file = argv[1];
  SuperGlueConfig sgc{};
  sgc.image_width = 640; sgc.image_height = 512; sgc.matching_threshold = 0.5; sgc.dla_core = -1;
  sgc.engine_file = argv[2];
  if (argc > 9) { spc.onnx_file = argv[8]; sgc.onnx_file = argv[9]; }
  SuperPointPtr superpoint = SuperPointPtr(new SuperPoint(spc));
  if (!superpoint->build()) { std::cout << "Error in SuperPoint building" << std::endl; return 1; }   // tracking.cc:39-43
  PointMatchingPtr point_matching = PointMatchingPtr(new PointMatching(sgc));                          // tracking.cc:45
  auto b0 = read_file(argv[3], (size_t)H * W), b1 = read_file(argv[4], (size_t)H * W);
  cv::Mat image0(H, W, 0, b0.data()), image1(H, W, 0, b1.data()), mask;
  Eigen::Matrix<double, 259, Eigen::Dynamic> features0, features1;
  // every call of the reference comes from a FRESH std::thread (src/tracking.cc:334-335, 364-366): the
  // handles keep no thread-local state and bind their HIP device on entry
  bool ok0 = false, ok1 = false;
  std::function<void()> extract_point = [&]() { ok0 = superpoint->infer(image0, mask, features0); };   // ExtractFeatrue
  std::thread t0(extract_point);
  t0.join();
  std::vector<cv::DMatch> matches;
  int n = -1;
  std::function<void()> extract_point_and_match = [&]() {                                              // ExtractFeatureAndMatch
    ok1 = superpoint->infer(image1, mask, features1);
    if (!ok1) return;
    matches.clear();
    n = point_matching->MatchingPoints(features0, features1, matches, true);                           // tracking.cc:354
  };
  std::thread t1(extract_point_and_match);
  t1.join();
  if (!ok0 || !ok1) return 1;
  if (argc > 7 && argv[7][0] != '-') superpoint->visualization(argv[7], image1);      // SuperPoint::visualization, src/super_point.cpp:388-400
  printf("K0=%ld K1=%ld matches=%d\n", (long)features0.cols(), (long)features1.cols(), n);
  for (int i = 0; i < n; ++i) printf("%d %d %.9g\n", matches[i].queryIdx, matches[i].trainIdx, matches[i].distance);
  // EpipolarGeometry compiles against the same handle (mono init, src/tracking.cc:52-55,559)
  Eigen::Matrix3f Kc;
  Kc(0, 0) = 500; Kc(0, 1) = 0; Kc(0, 2) = 320; Kc(1, 0) = 0; Kc(1, 1) = 500; Kc(1, 2) = 240; Kc(2, 0) = 0; Kc(2, 1) = 0; Kc(2, 2) = 1;
  EpipolarGeometry eg(Kc, 1.0, 200);
  (void)eg;
  return 0;
}
