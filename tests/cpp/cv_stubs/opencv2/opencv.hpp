// TEST INFRASTRUCTURE ONLY (tests/test_abi_cpu.py::test_shim_headers_opencv_branch_is_well_formed): declarations of the
// handful of OpenCV names the shim headers use when OpenCV is installed (the URF_HAVE_CV branch of include/super_point.h),
// so that the branch can be syntax-checked with -fsyntax-only in an image that has no OpenCV.  Nothing here is ever linked,
// run, or used to build any part of the reference.
#pragma once
#include <cstddef>
#include <string>
#include <vector>
namespace cv {
enum { CV_8UC1_STUB = 0 };
enum ColorConversionCodes { COLOR_GRAY2BGR = 8 };
struct Point { int x, y; Point(int a = 0, int b = 0) : x(a), y(b) {} Point(double a, double b) : x((int)a), y((int)b) {} };
struct Scalar { double v[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {} };
struct Mat {
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char *data = nullptr;
  Mat() = default;
  Mat(int r, int c, int /*type*/, void *d, size_t s = 0) : rows(r), cols(c), step(s ? s : (size_t)c), data((unsigned char *)d) {}
  bool empty() const;
  int channels() const;
  Mat clone() const;
  template <typename T> T *ptr(int y);
};
struct Point2f { float x = 0, y = 0; };
struct Point3f { float x = 0, y = 0, z = 0; Point3f() = default; Point3f(float a, float b, float c) : x(a), y(b), z(c) {} };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 0;
  DMatch() = default;
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};
void cvtColor(const Mat &src, Mat &dst, int code);
void circle(Mat &img, Point center, int radius, const Scalar &color, int thickness = 1, int lineType = 8, int shift = 0);
bool imwrite(const std::string &filename, const Mat &img);
}  // namespace cv
#ifndef CV_8UC1
#define CV_8UC1 0
#endif
