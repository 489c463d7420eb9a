// epipolar_api.hip -- EpipolarGeometry::reconstruct behind the C ABI (src/epipolar_geometry.cc:18-98).
// The two RANSAC searches (_find_H, _find_F: 2 x `iterations` minimal-set solves and their scores) run on
// the GPU (ransac_kernels.hip).  What follows them happens once per sequence and stays on the host like in
// the reference: model selection (:86-97), the motion candidates of the winning model (_reconstruct_F
// :451-562 with _decompose_E :900-926, or _reconstruct_H :564-733, Faugeras' eight solutions) and their
// cheirality / reprojection / parallax tally (_check_R_T :782-898 with _triangulate :928-950).
//
// Structure of this file: small value types (M3, Pose), one `tally()` per motion candidate, one selection
// rule per model.  Arithmetic spec (DESIGN.md "RANSAC"): float where the reference computes in float, sums
// associated left to right; Eigen::JacobiSVD replaced by cyclic Jacobi (12 sweeps) on the Gram matrix in f64.
#include "../../include/urf.h"
#include "urf_common.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <vector>

namespace urf {
int launch_epipolar_search(const float *keys1, int n1, const float *keys2, int n2, const float *pts0, const float *pts1,
                           const int *d_nm, int nm, float *pn0, float *pn1, float *T, float *F, float *scoreF,
                           float *H, float *scoreH, uint32_t seed, int iters, float sigma, const int *d_sets,
                           hipStream_t st);

namespace epi {

// ------------------------------------------------------------------ glibc rand(), restated
// The reference draws its minimal sets with rand() after srand(0) (src/epipolar_geometry.cc:100-117).
// glibc's default generator (stdlib/random_r.c, TYPE_3): 31 words, x[i] = x[i-31] + x[i-3] (mod 2^32),
// output x[i] >> 1; seeded by the Lehmer sequence 16807 * x mod (2^31 - 1) (Schrage's form), seed 0 -> 1,
// first 310 outputs discarded.  tests/test_abi_cpu.py checks it against the C library's own rand().
class GlibcRand {
 public:
  explicit GlibcRand(uint32_t seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    ring_[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const int32_t hi = word / 127773, lo = word % 127773;
      word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      ring_[i] = (uint32_t)word;
    }
    front_ = 3; rear_ = 0;
    for (int i = 0; i < 310; ++i) (void)next();
  }
  uint32_t next() {
    ring_[front_] += ring_[rear_];
    const uint32_t out = ring_[front_] >> 1;
    front_ = (front_ + 1) % 31;
    rear_ = (rear_ + 1) % 31;
    return out;
  }
  // Random::RandomInt :114-117 (RAND_MAX = 2^31 - 1)
  int uniform_int(int lo, int hi) { return (int)(((double)next() / 2147483648.0) * (hi - lo + 1)) + lo; }

 private:
  uint32_t ring_[31];
  int front_, rear_;
};

static uint32_t hash_counter(uint32_t seed, uint32_t ctr) {   // the sampler of ransac_kernels.hip (rs_hash)
  uint32_t x = seed ^ (ctr * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// :52-71: per iteration 8 draws without replacement, swap-with-back on the list of all match indices
static void minimal_sets(int sampler, uint32_t seed, int n, int iterations, int *sets) {
  GlibcRand libc(seed);
  std::vector<int> avail((size_t)n);
  for (int it = 0; it < iterations; ++it) {
    for (int i = 0; i < n; ++i) avail[i] = i;
    int size = n;
    for (int j = 0; j < 8; ++j) {
      int pick;
      if (sampler == URF_SAMPLER_GLIBC) {
        pick = libc.uniform_int(0, size - 1);
      } else {
        const uint32_t r = hash_counter(seed, (uint32_t)(it * 8 + j)) >> 1;
        pick = (int)(((double)r / 2147483648.0) * size);
      }
      sets[it * 8 + j] = avail[pick];
      avail[pick] = avail[size - 1];
      --size;
    }
  }
}

// ------------------------------------------------------------------ small linear algebra
struct M3 {
  float v[9];
  float operator()(int r, int c) const { return v[3 * r + c]; }
  float &operator()(int r, int c) { return v[3 * r + c]; }
};
static M3 from(const float *p) { M3 m; memcpy(m.v, p, sizeof(m.v)); return m; }
static M3 operator*(const M3 &a, const M3 &b) {
  M3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o(r, c) = (a(r, 0) * b(0, c) + a(r, 1) * b(1, c)) + a(r, 2) * b(2, c);
  return o;
}
static M3 transposed(const M3 &a) {
  M3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o(r, c) = a(c, r);
  return o;
}
static M3 inverse(const M3 &m) {   // adjugate / determinant, float (K.inverse() of a calibration matrix)
  const float c00 = m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1);
  const float c01 = -(m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0));
  const float c02 = m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0);
  const float inv_det = 1.0f / ((m(0, 0) * c00 + m(0, 1) * c01) + m(0, 2) * c02);
  M3 o;
  o(0, 0) = c00 * inv_det;
  o(0, 1) = -(m(0, 1) * m(2, 2) - m(0, 2) * m(2, 1)) * inv_det;
  o(0, 2) = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * inv_det;
  o(1, 0) = c01 * inv_det;
  o(1, 1) = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * inv_det;
  o(1, 2) = -(m(0, 0) * m(1, 2) - m(0, 2) * m(1, 0)) * inv_det;
  o(2, 0) = c02 * inv_det;
  o(2, 1) = -(m(0, 0) * m(2, 1) - m(0, 1) * m(2, 0)) * inv_det;
  o(2, 2) = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * inv_det;
  return o;
}
template <typename T>
static double det3(const T *m) {
  return ((double)m[0] * ((double)m[4] * m[8] - (double)m[5] * m[7]) - (double)m[1] * ((double)m[3] * m[8] - (double)m[5] * m[6])) +
         (double)m[2] * ((double)m[3] * m[7] - (double)m[4] * m[6]);
}

// symmetric eigen-decomposition, cyclic Jacobi with a fixed number of sweeps: on return the diagonal of `a`
// holds the eigenvalues and the columns of `vec` the eigenvectors
template <int N>
static void jacobi(std::array<double, N * N> &a, std::array<double, N * N> &vec) {
  vec.fill(0.0);
  for (int i = 0; i < N; ++i) vec[i * N + i] = 1.0;
  auto rotate = [](double &x, double &y, double c, double s) {
    const double x0 = x, y0 = y;
    x = c * x0 - s * y0;
    y = s * x0 + c * y0;
  };
  for (int sweep = 0; sweep < 12; ++sweep)
    for (int p = 0; p + 1 < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double off = a[p * N + q];
        if (fabs(off) < 1e-300) continue;
        const double theta = (a[q * N + q] - a[p * N + p]) / (2.0 * off);
        const double tn = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tn * tn + 1.0), s = tn * c;
        for (int k = 0; k < N; ++k) rotate(a[k * N + p], a[k * N + q], c, s);   // columns p, q
        for (int k = 0; k < N; ++k) rotate(a[p * N + k], a[q * N + k], c, s);   // rows p, q
        for (int k = 0; k < N; ++k) rotate(vec[k * N + p], vec[k * N + q], c, s);
      }
}
template <int N>
static int weakest(const std::array<double, N * N> &a) {
  int m = 0;
  for (int i = 1; i < N; ++i)
    if (a[i * N + i] < a[m * N + m]) m = i;
  return m;
}

// A = U diag(w) V^T with w descending, through the eigenvectors of A^T A; the third left vector is u0 x u1,
// oriented along A v2 (exactly orthogonal also when w2 ~ 0: essential matrices)
struct Svd3 { double U[9], w[3], V[9]; };
static Svd3 svd(const M3 &Af) {
  Svd3 out;
  double A[9];
  for (int k = 0; k < 9; ++k) A[k] = (double)Af.v[k];
  std::array<double, 9> gram, evec;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc = acc + A[k * 3 + r] * A[k * 3 + c];
      gram[r * 3 + c] = acc;
    }
  jacobi<3>(gram, evec);
  int order[3] = {0, 1, 2};
  for (int x = 0; x < 2; ++x)
    for (int y = x + 1; y < 3; ++y)
      if (gram[order[y] * 4] > gram[order[x] * 4]) std::swap(order[x], order[y]);
  for (int j = 0; j < 3; ++j) {
    const double ev = gram[order[j] * 4];
    out.w[j] = ev > 0.0 ? sqrt(ev) : 0.0;
    for (int k = 0; k < 3; ++k) out.V[k * 3 + j] = evec[k * 3 + order[j]];
  }
  auto A_times_vcol = [&](int r, int j) { return (A[r * 3] * out.V[j] + A[r * 3 + 1] * out.V[3 + j]) + A[r * 3 + 2] * out.V[6 + j]; };
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 3; ++r) out.U[r * 3 + j] = A_times_vcol(r, j) / out.w[j];
  const double cx = out.U[3] * out.U[7] - out.U[6] * out.U[4];
  const double cy = out.U[6] * out.U[1] - out.U[0] * out.U[7];
  const double cz = out.U[0] * out.U[4] - out.U[3] * out.U[1];
  const double along = (A_times_vcol(0, 2) * cx + A_times_vcol(1, 2) * cy) + A_times_vcol(2, 2) * cz;
  const double sign = along < 0.0 ? -1.0 : 1.0;
  out.U[2] = sign * cx; out.U[5] = sign * cy; out.U[8] = sign * cz;
  return out;
}

// ------------------------------------------------------------------ the two-view problem
struct Pose { M3 R; float t[3]; };

struct TwoView {
  const float *keys1, *keys2;
  int n1;
  std::vector<std::array<int, 2>> pairs;     // _vMatches12: (index in image 1, index in image 2)
  std::vector<uint8_t> inlier;               // of the winning model, per pair
  M3 K;
  float sigma2;
};

// per-correspondence symmetric transfer tests: the float expressions of _check_F :372-449 / _check_H :285-370
// (the same ones the scoring kernels evaluate)
static bool fits_F(const M3 &F, const float *x1, const float *x2, float inv_sigma2) {
  const float l2a = (F(0, 0) * x1[0] + F(0, 1) * x1[1]) + F(0, 2);
  const float l2b = (F(1, 0) * x1[0] + F(1, 1) * x1[1]) + F(1, 2);
  const float l2c = (F(2, 0) * x1[0] + F(2, 1) * x1[1]) + F(2, 2);
  const float r2 = (l2a * x2[0] + l2b * x2[1]) + l2c;
  const bool ok2 = !(((r2 * r2) / (l2a * l2a + l2b * l2b)) * inv_sigma2 > 3.841f);
  const float l1a = (F(0, 0) * x2[0] + F(1, 0) * x2[1]) + F(2, 0);
  const float l1b = (F(0, 1) * x2[0] + F(1, 1) * x2[1]) + F(2, 1);
  const float l1c = (F(0, 2) * x2[0] + F(1, 2) * x2[1]) + F(2, 2);
  const float r1 = (l1a * x1[0] + l1b * x1[1]) + l1c;
  const bool ok1 = !(((r1 * r1) / (l1a * l1a + l1b * l1b)) * inv_sigma2 > 3.841f);
  return ok1 && ok2;
}
static bool fits_H(const M3 &H21, const M3 &H12, const float *x1, const float *x2, float inv_sigma2) {
  auto transfer_error = [&](const M3 &H, const float *from, const float *to) {
    const float wi = (float)(1.0 / (double)((H(2, 0) * from[0] + H(2, 1) * from[1]) + H(2, 2)));
    const float px = ((H(0, 0) * from[0] + H(0, 1) * from[1]) + H(0, 2)) * wi;
    const float py = ((H(1, 0) * from[0] + H(1, 1) * from[1]) + H(1, 2)) * wi;
    return ((to[0] - px) * (to[0] - px) + (to[1] - py) * (to[1] - py)) * inv_sigma2;
  };
  const bool ok1 = !(transfer_error(H12, x2, x1) > 5.991f);
  const bool ok2 = !(transfer_error(H21, x1, x2) > 5.991f);
  return ok1 && ok2;
}

// _triangulate :928-950: null vector of the 4x4 DLT matrix
static bool triangulate(const float *x1, const float *x2, const float (&P1)[12], const float (&P2)[12], float X[3]) {
  float D[16];
  for (int c = 0; c < 4; ++c) {
    D[c] = x1[0] * P1[8 + c] - P1[c];
    D[4 + c] = x1[1] * P1[8 + c] - P1[4 + c];
    D[8 + c] = x2[0] * P2[8 + c] - P2[c];
    D[12 + c] = x2[1] * P2[8 + c] - P2[4 + c];
  }
  std::array<double, 16> gram, evec;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc = acc + (double)D[k * 4 + r] * (double)D[k * 4 + c];
      gram[r * 4 + c] = acc;
    }
  jacobi<4>(gram, evec);
  const int m = weakest<4>(gram);
  const float hx = (float)evec[m], hy = (float)evec[4 + m], hz = (float)evec[8 + m], hw = (float)evec[12 + m];
  if (hw == 0.0f) return false;
  X[0] = hx / hw; X[1] = hy / hw; X[2] = hz / hw;
  return true;
}

// _check_R_T :782-898 for one motion candidate
struct Tally {
  int good = 0;
  float parallax = 0.0f;
  std::vector<uint8_t> triangulated;   // per keypoint of image 1
  std::vector<float> points;           // 3 per keypoint of image 1
};
static Tally tally(const TwoView &tv, const Pose &pose) {
  Tally out;
  out.triangulated.assign((size_t)std::max(tv.n1, 1), 0);
  out.points.assign(3 * (size_t)std::max(tv.n1, 1), 0.0f);
  const M3 &K = tv.K, &R = pose.R;
  const float *t = pose.t;
  const float fx = K(0, 0), fy = K(1, 1), cx = K(0, 2), cy = K(1, 2);
  const float th2 = 4.0f * tv.sigma2;
  float P1[12] = {0}, P2[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) P1[r * 4 + c] = K(r, c);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      auto Rt = [&](int rr) { return c < 3 ? R(rr, c) : t[rr]; };
      P2[r * 4 + c] = (K(r, 0) * Rt(0) + K(r, 1) * Rt(1)) + K(r, 2) * Rt(2);
    }
  float centre2[3];   // O2 = -R^T t
  for (int r = 0; r < 3; ++r) centre2[r] = -((R(0, r) * t[0] + R(1, r) * t[1]) + R(2, r) * t[2]);
  std::vector<float> cosines;
  cosines.reserve(tv.pairs.size());
  for (size_t m = 0; m < tv.pairs.size(); ++m) {
    if (!tv.inlier[m]) continue;
    const int i1 = tv.pairs[m][0], i2 = tv.pairs[m][1];
    const float *x1 = tv.keys1 + 2 * i1, *x2 = tv.keys2 + 2 * i2;
    float X[3] = {0.0f, 0.0f, 0.0f};
    triangulate(x1, x2, P1, P2, X);
    if (!std::isfinite(X[0]) || !std::isfinite(X[1]) || !std::isfinite(X[2])) { out.triangulated[i1] = 0; continue; }
    const float ray2[3] = {X[0] - centre2[0], X[1] - centre2[1], X[2] - centre2[2]};
    const float len1 = sqrtf((X[0] * X[0] + X[1] * X[1]) + X[2] * X[2]);
    const float len2 = sqrtf((ray2[0] * ray2[0] + ray2[1] * ray2[1]) + ray2[2] * ray2[2]);
    const float cos_parallax = ((X[0] * ray2[0] + X[1] * ray2[1]) + X[2] * ray2[2]) / (len1 * len2);
    const bool wide = cos_parallax < 0.99998f;
    if (X[2] <= 0 && wide) continue;                            // behind camera 1
    float Y[3];                                                 // the point in camera 2
    for (int r = 0; r < 3; ++r) Y[r] = ((R(r, 0) * X[0] + R(r, 1) * X[1]) + R(r, 2) * X[2]) + t[r];
    if (Y[2] <= 0 && wide) continue;                            // behind camera 2
    auto reprojection2 = [&](const float *Pc, const float *x) {
      const float iz = (float)(1.0 / (double)Pc[2]);
      const float ex = (fx * Pc[0] * iz + cx) - x[0], ey = (fy * Pc[1] * iz + cy) - x[1];
      return ex * ex + ey * ey;
    };
    if (reprojection2(X, x1) > th2) continue;
    if (reprojection2(Y, x2) > th2) continue;
    cosines.push_back(cos_parallax);
    memcpy(&out.points[3 * (size_t)i1], X, sizeof(X));
    ++out.good;
    if (wide) out.triangulated[i1] = 1;
  }
  if (out.good > 0) {
    std::sort(cosines.begin(), cosines.end());
    const size_t idx = std::min<size_t>(50, cosines.size() - 1);
    out.parallax = (float)(acos((double)cosines[idx]) * 180.0 / 3.1415926535897932384626433832795);
  }
  return out;
}

// _reconstruct_F :451-562 (+ _decompose_E :900-926): four candidates (R1|R2) x (+t|-t)
static bool motion_from_F(TwoView &tv, const M3 &F21, const float *p0, const float *p1, const Pose **chosen_pose,
                          std::array<Pose, 8> &cands, Tally &chosen) {
  const float inv = (float)(1.0 / (double)tv.sigma2);
  int N = 0;
  for (size_t m = 0; m < tv.pairs.size(); ++m) { tv.inlier[m] = fits_F(F21, p0 + 2 * m, p1 + 2 * m, inv); N += tv.inlier[m]; }
  const M3 E = (transposed(tv.K) * F21) * tv.K;
  const Svd3 d = svd(E);
  M3 U, Vt;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { U(r, c) = (float)d.U[r * 3 + c]; Vt(r, c) = (float)d.V[c * 3 + r]; }
  float t[3];
  {
    const float len = sqrtf((U(0, 2) * U(0, 2) + U(1, 2) * U(1, 2)) + U(2, 2) * U(2, 2));
    for (int r = 0; r < 3; ++r) t[r] = U(r, 2) / len;
  }
  const M3 W = {{0, -1, 0, 1, 0, 0, 0, 0, 1}}, Wt = {{0, 1, 0, -1, 0, 0, 0, 0, 1}};
  M3 R1 = (U * W) * Vt, R2 = (U * Wt) * Vt;
  if (det3(R1.v) < 0) for (float &x : R1.v) x = -x;
  if (det3(R2.v) < 0) for (float &x : R2.v) x = -x;
  for (int c = 0; c < 4; ++c) {
    cands[c].R = (c & 1) ? R2 : R1;
    for (int r = 0; r < 3; ++r) cands[c].t[r] = (c & 2) ? -t[r] : t[r];
  }
  Tally res[4];
  int most = 0;
  for (int c = 0; c < 4; ++c) { res[c] = tally(tv, cands[c]); most = std::max(most, res[c].good); }
  const int need = std::max((int)(0.9 * N), 50);
  int similar = 0;
  for (int c = 0; c < 4; ++c) similar += res[c].good > 0.7 * most;
  if (most < need || similar > 1) return false;
  for (int c = 0; c < 4; ++c)
    if (res[c].good == most) {
      if (!(res[c].parallax > 1.0f)) return false;
      chosen = std::move(res[c]);
      *chosen_pose = &cands[c];
      return true;
    }
  return false;
}

// _reconstruct_H :564-733: Faugeras' decomposition, eight candidates
static bool motion_from_H(TwoView &tv, const M3 &H21, const M3 &H12, const float *p0, const float *p1,
                          const Pose **chosen_pose, std::array<Pose, 8> &cands, Tally &chosen) {
  const float inv = (float)(1.0 / (double)tv.sigma2);
  int N = 0;
  for (size_t m = 0; m < tv.pairs.size(); ++m) { tv.inlier[m] = fits_H(H21, H12, p0 + 2 * m, p1 + 2 * m, inv); N += tv.inlier[m]; }
  const M3 A = (inverse(tv.K) * H21) * tv.K;
  const Svd3 d = svd(A);
  double Vt_d[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Vt_d[r * 3 + c] = d.V[c * 3 + r];
  const float s = (float)(det3(d.U) * det3(Vt_d));
  const float d1 = (float)d.w[0], d2 = (float)d.w[1], d3 = (float)d.w[2];
  if (d1 / d2 < 1.00001f || d2 / d3 < 1.00001f) return false;
  M3 U, Vt;
  for (int k = 0; k < 9; ++k) { U.v[k] = (float)d.U[k]; Vt.v[k] = (float)Vt_d[k]; }
  const float span = d1 * d1 - d3 * d3;
  const float e1 = sqrtf((d1 * d1 - d2 * d2) / span), e3 = sqrtf((d2 * d2 - d3 * d3) / span);
  const float root = sqrtf((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3));
  const float sin_theta = root / ((d1 + d3) * d2), cos_theta = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2);
  const float sin_phi = root / ((d1 - d3) * d2), cos_phi = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2);
  static const float sx1[4] = {1, 1, -1, -1}, sx3[4] = {1, -1, 1, -1}, ssin[4] = {1, -1, -1, 1};
  for (int c = 0; c < 8; ++c) {
    const int i = c & 3;
    const float x1 = sx1[i] * e1, x3 = sx3[i] * e3;
    M3 Rp = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    float tp[3] = {0.0f, 0.0f, 0.0f};
    if (c < 4) {          // d' > 0
      const float sn = ssin[i] * sin_theta;
      Rp(0, 0) = cos_theta; Rp(0, 2) = -sn; Rp(1, 1) = 1.0f; Rp(2, 0) = sn; Rp(2, 2) = cos_theta;
      tp[0] = x1 * (d1 - d3); tp[2] = -x3 * (d1 - d3);
    } else {              // d' < 0
      const float sn = ssin[i] * sin_phi;
      Rp(0, 0) = cos_phi; Rp(0, 2) = sn; Rp(1, 1) = -1.0f; Rp(2, 0) = sn; Rp(2, 2) = -cos_phi;
      tp[0] = x1 * (d1 + d3); tp[2] = x3 * (d1 + d3);
    }
    const M3 R = (U * Rp) * Vt;
    for (int k = 0; k < 9; ++k) cands[c].R.v[k] = s * R.v[k];
    float tt[3];
    for (int r = 0; r < 3; ++r) tt[r] = (U(r, 0) * tp[0] + U(r, 1) * tp[1]) + U(r, 2) * tp[2];
    const float len = sqrtf((tt[0] * tt[0] + tt[1] * tt[1]) + tt[2] * tt[2]);
    for (int r = 0; r < 3; ++r) cands[c].t[r] = tt[r] / len;
  }
  int best = 0, runner_up = 0, best_c = -1;
  Tally best_res;
  for (int c = 0; c < 8; ++c) {
    Tally r = tally(tv, cands[c]);
    if (r.good > best) { runner_up = best; best = r.good; best_c = c; best_res = std::move(r); }
    else if (r.good > runner_up) runner_up = r.good;
  }
  if (!(runner_up < 0.75 * best && best_res.parallax >= 1.0f && best > 50 && best > 0.9 * N)) return false;
  chosen = std::move(best_res);
  *chosen_pose = &cands[best_c];
  return true;
}

}  // namespace epi
}  // namespace urf
using namespace urf;

extern "C" void *urf_pm_stream_(urf_pm *h);
extern "C" int urf_pm_device_(urf_pm *h);

extern "C" int urf_minimal_sets(int sampler, uint32_t seed, int n, int iterations, int *sets) {
  URF_CHECK(sets && n >= 8 && iterations >= 1, "urf_minimal_sets: need n >= 8 matches, iterations >= 1");
  URF_CHECK(sampler == URF_SAMPLER_HASH || sampler == URF_SAMPLER_GLIBC, "urf_minimal_sets: unknown sampler %d", sampler);
  epi::minimal_sets(sampler, seed, n, iterations, sets);
  return 0;
}

// device scratch of one reconstruct call (freed on every exit path)
namespace {
struct DeviceScratch {
  float *p = nullptr;
  ~DeviceScratch() { if (p) (void)hipFree(p); }
};
}  // namespace

static int reconstruct_impl(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1, const float *keys2, int n2,
                            const int *matches12, const int *sets, float *T21, float *P3D, uint8_t *tri, int *model,
                            float *scores) {
  URF_CHECK(h && cfg && keys1 && keys2 && matches12 && T21 && P3D && tri && model && scores,
            "urf_epipolar_reconstruct: null argument");
  URF_CHECK(n1 >= 0 && n1 <= kCap && n2 >= 0 && n2 <= kCap, "keypoint counts (%d,%d) outside [0,%d]", n1, n2, kCap);
  hipStream_t st = (hipStream_t)urf_pm_stream_(h);
  URF_CHECK(st, "PointMatching handle is not built");
  for (int k = 0; k < 16; ++k) T21[k] = (k % 5 == 0) ? 1.0f : 0.0f;
  std::fill(tri, tri + n1, (uint8_t)0);
  *model = -1; scores[0] = scores[1] = 0.0f;

  epi::TwoView tv;
  tv.keys1 = keys1; tv.keys2 = keys2; tv.n1 = n1;
  for (int i = 0; i < n1; ++i)
    if (matches12[i] >= 0) {
      URF_CHECK(matches12[i] < n2, "match index out of range");
      tv.pairs.push_back({i, matches12[i]});
    }
  const int nm = (int)tv.pairs.size();
  if (nm < 8) return 0;
  const int its = cfg->iterations > 0 ? cfg->iterations : 200;
  const float sigma = cfg->sigma > 0 ? cfg->sigma : 1.0f;
  tv.K = epi::from(cfg->K);
  tv.sigma2 = sigma * sigma;
  tv.inlier.assign((size_t)nm, 0);
  std::vector<float> p0(2 * (size_t)nm), p1(2 * (size_t)nm);
  for (int m = 0; m < nm; ++m) {
    memcpy(&p0[2 * (size_t)m], keys1 + 2 * tv.pairs[m][0], 2 * sizeof(float));
    memcpy(&p1[2 * (size_t)m], keys2 + 2 * tv.pairs[m][1], 2 * sizeof(float));
  }
  // minimal sets: the caller's, the reference's rand() stream, or (nullptr) the device-side counter hash
  std::vector<int> host_sets;
  if (!sets && cfg->sampler == URF_SAMPLER_GLIBC) {
    host_sets.resize((size_t)its * 8);
    epi::minimal_sets(URF_SAMPLER_GLIBC, cfg->seed, nm, its, host_sets.data());
    sets = host_sets.data();
  }
  if (sets)
    for (int k = 0; k < its * 8; ++k) URF_CHECK(sets[k] >= 0 && sets[k] < nm, "minimal set index %d outside the %d matches", sets[k], nm);

  // ---- GPU: both RANSAC searches
  URF_HIP(hipSetDevice(urf_pm_device_(h)));
  DeviceScratch scratch;
  const size_t nf = 2 * (size_t)n1 + 2 * (size_t)n2 + 8 * (size_t)nm + 18 + (size_t)its * (9 + 1 + 18 + 1 + 8) + 8;
  URF_HIP(hipMalloc((void **)&scratch.p, nf * sizeof(float)));
  float *dk1 = scratch.p, *dk2 = dk1 + 2 * n1, *dp0 = dk2 + 2 * n2, *dp1 = dp0 + 2 * nm, *dq0 = dp1 + 2 * nm, *dq1 = dq0 + 2 * nm;
  float *dT = dq1 + 2 * nm, *dF = dT + 18, *dsF = dF + (size_t)its * 9, *dH = dsF + its, *dsH = dH + (size_t)its * 18;
  int *dnm = (int *)(dsH + its), *dsets = dnm + 8;
  std::vector<float> F((size_t)its * 9), H((size_t)its * 18), sF(its), sH(its);
  URF_HIP(hipMemcpyAsync(dk1, keys1, 8 * (size_t)n1, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(dk2, keys2, 8 * (size_t)n2, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(dp0, p0.data(), 8 * (size_t)nm, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(dp1, p1.data(), 8 * (size_t)nm, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(dnm, &nm, sizeof(int), hipMemcpyHostToDevice, st));
  if (sets) URF_HIP(hipMemcpyAsync(dsets, sets, (size_t)its * 8 * sizeof(int), hipMemcpyHostToDevice, st));
  if (launch_epipolar_search(dk1, n1, dk2, n2, dp0, dp1, dnm, nm, dq0, dq1, dT, dF, dsF, dH, dsH, cfg->seed, its, sigma,
                             sets ? dsets : nullptr, st))
    return -1;
  URF_HIP(hipMemcpyAsync(F.data(), dF, F.size() * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(H.data(), dH, H.size() * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(sF.data(), dsF, (size_t)its * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(sH.data(), dsH, (size_t)its * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipStreamSynchronize(st));

  // ---- model selection :86-97 (first best hypothesis of each search, strict '>')
  const int bF = (int)(std::max_element(sF.begin(), sF.end()) - sF.begin());
  const int bH = (int)(std::max_element(sH.begin(), sH.end()) - sH.begin());
  const float SF = sF[bF] > 0.0f ? sF[bF] : 0.0f, SH = sH[bH] > 0.0f ? sH[bH] : 0.0f;
  scores[0] = SH; scores[1] = SF;
  if (SH + SF == 0.0f) return 0;
  const float RH = SH / (SH + SF);
  std::array<epi::Pose, 8> cands;
  const epi::Pose *pose = nullptr;
  epi::Tally result;
  bool ok;
  if (RH > 0.50f && SH > 0.0f) {
    *model = 0;
    ok = epi::motion_from_H(tv, epi::from(&H[(size_t)bH * 18]), epi::from(&H[(size_t)bH * 18 + 9]), p0.data(), p1.data(), &pose,
                            cands, result);
  } else if (SF > 0.0f) {
    *model = 1;
    ok = epi::motion_from_F(tv, epi::from(&F[(size_t)bF * 9]), p0.data(), p1.data(), &pose, cands, result);
  } else {
    return 0;
  }
  if (!ok) return 0;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T21[r * 4 + c] = pose->R(r, c);
    T21[r * 4 + 3] = pose->t[r];
  }
  memcpy(tri, result.triangulated.data(), (size_t)n1);
  memcpy(P3D, result.points.data(), 12 * (size_t)n1);
  return 1;
}

extern "C" int urf_epipolar_reconstruct(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1,
                                        const float *keys2, int n2, const int *matches12, float *T21, float *P3D,
                                        uint8_t *tri, int *model, float *scores) {
  return reconstruct_impl(h, cfg, keys1, n1, keys2, n2, matches12, nullptr, T21, P3D, tri, model, scores);
}

extern "C" int urf_epipolar_reconstruct_sets(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1,
                                             const float *keys2, int n2, const int *matches12, const int *sets,
                                             float *T21, float *P3D, uint8_t *tri, int *model, float *scores) {
  URF_CHECK(sets, "urf_epipolar_reconstruct_sets: null sets");
  return reconstruct_impl(h, cfg, keys1, n1, keys2, n2, matches12, sets, T21, P3D, tri, model, scores);
}
