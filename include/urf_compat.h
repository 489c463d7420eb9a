// urf_compat.h -- minimal stand-ins for the cv:: / Eigen:: types that appear in
// the kept headers, used ONLY when OpenCV / Eigen are not installed (this image
// has neither).  With the real libraries present the shim headers use them and
// this file is not included.  Same member names as the real types, so the shim
// code is identical in both modes.
#ifndef URF_COMPAT_H_
#define URF_COMPAT_H_
#include <cstddef>
#include <cstdint>
#include <vector>

namespace cv {
struct Mat {  // 8-bit single-channel view, like cv::Mat(rows, cols, CV_8UC1, data, step)
  int rows = 0, cols = 0;
  size_t step = 0;
  unsigned char *data = nullptr;
  Mat() = default;
  Mat(int r, int c, int /*type*/, void *d, size_t s = 0) : rows(r), cols(c), step(s ? s : (size_t)c), data((unsigned char *)d) {}
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
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
}  // namespace cv

namespace Eigen {
const int Dynamic = -1;
template <typename T, int R, int C>
class Matrix {  // column-major R x n (R fixed) or n x 1 vector storage
 public:
  void resize(long r, long c) { rows_ = r; cols_ = c; d_.assign((size_t)(r * c), T()); }
  void resize(long n) { rows_ = n; cols_ = 1; d_.assign((size_t)n, T()); }
  long rows() const { return rows_; }
  long cols() const { return cols_; }
  long size() const { return rows_ * cols_; }
  T *data() { return d_.data(); }
  const T *data() const { return d_.data(); }
  T &operator()(long r, long c) { return d_[(size_t)(c * rows_ + r)]; }
  const T &operator()(long r, long c) const { return d_[(size_t)(c * rows_ + r)]; }
  T &operator()(long i) { return d_[(size_t)i]; }
  const T &operator()(long i) const { return d_[(size_t)i]; }
  T &operator[](long i) { return d_[(size_t)i]; }
  const T &operator[](long i) const { return d_[(size_t)i]; }
 private:
  long rows_ = (R > 0 ? R : 0), cols_ = (C > 0 ? C : 0);
  std::vector<T> d_ = std::vector<T>((size_t)((R > 0 && C > 0) ? R * C : 0));
};
typedef Matrix<int, Dynamic, 1> VectorXi;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
}  // namespace Eigen
#endif
