/* cvransac_oracle.c -- TEST INFRASTRUCTURE (CPU oracle).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may call this; the product never does.
 *
 * The outlier stage the reference actually calls: cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask)
 * (src/point_matching.cc:43-58).  OpenCV is an un-vendored dependency of the reference (Ubuntu 20.04's libopencv-dev =
 * 4.2.0, docker/Dockerfile:4,132) and is not in this image: this file RESTATES the published algorithm of OpenCV 4.2.0 --
 *   modules/calib3d/src/fundam.cpp   : cv::findFundamentalMat (dispatch on the point count), run7Point,
 *                                      FMEstimatorCallback::{checkSubset, runKernel, computeError}, haveCollinearPoints
 *   modules/calib3d/src/ptsetreg.cpp : RANSACPointSetRegistrator::{run, getSubset, findInliers}, RANSACUpdateNumIters,
 *                                      LMeDSPointSetRegistrator::run
 *   modules/core (cv::RNG)           : the multiply-with-carry generator, seeded with (uint64)-1, uniform(a, b) = next() % (b - a) + a
 * -- from memory of its source; PARITY UNPINNED: there is no OpenCV binary here to check it against, and the reference holds no
 * vector for this call.  Where OpenCV calls into its own numerical library the result depends on that library's rounding, and
 * this restatement (and the HIP kernel, ur-mvo_amd/csrc/cvransac.hip, which implements the SAME written arithmetic a second
 * time and agrees with this file bit for bit) chooses a fixed, libm-free arithmetic instead:
 *   * the two-dimensional null space of the 7 x 9 system (OpenCV: cv::SVDecomp, last two rows of V^T) by Gauss-Jordan
 *     elimination with complete pivoting; each basis vector scaled to unit length.  Any basis of the null space gives the
 *     same one to three matrices F up to rounding; their ORDER within an iteration can differ from OpenCV's;
 *   * the real roots of the cubic det(lambda f1 + (1 - lambda) f2) = 0 (OpenCV: cv::solveCubic, trigonometric form) by
 *     bracketing between the critical points and bisection to the last bit, ascending;
 *   * RANSACUpdateNumIters' cvRound(log(1 - p) / log(1 - w^7)) without log / pow: the smallest k with q^k <= 1 - p by
 *     sequential multiplication (q = 1 - w^7, w^7 by six multiplications), minus one when q^(k-1) sqrt(q) < 1 - p (the real
 *     solution lies below k - 1/2).
 * Everything else follows the OpenCV source line by line: the RNG stream and the draw-until-distinct subsets, the
 * collinearity test of the LAST drawn point only, the symmetric epipolar distance max(d1^2 s1, d2^2 s2) as float against
 * (float)(3 * 3), "strictly more inliers than before and at least 7", at most 1000 iterations, no refit; fewer than 15 points:
 * LMedS with the same kernel (outlier ratio 0.45); exactly 7: the 7-point solution itself, mask all ones; fewer than 7:
 * OpenCV returns an empty matrix and leaves the mask empty, which the reference then indexes (undefined) -- here nothing is
 * rejected. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "urf_oracle.h"

typedef struct { uint64_t state; } cv_rng;
static uint32_t rng_next(cv_rng *r) {
  r->state = (uint64_t)(uint32_t)r->state * 4164903690ull + (uint32_t)(r->state >> 32);
  return (uint32_t)r->state;
}
static int rng_uniform(cv_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (uint32_t)(b - a) + (uint32_t)a); }

/* haveCollinearPoints(m, count): only the LAST point against the lines through pairs of the earlier ones */
static int collinear_last(const float *p, const int *idx, int count) {
  const int i = count - 1;
  for (int j = 0; j < i; ++j) {
    /* Point2f differences: float arithmetic, widened afterwards */
    const double dx1 = (double)(float)(p[2 * idx[j]] - p[2 * idx[i]]), dy1 = (double)(float)(p[2 * idx[j] + 1] - p[2 * idx[i] + 1]);
    for (int k = 0; k < j; ++k) {
      const double dx2 = (double)(float)(p[2 * idx[k]] - p[2 * idx[i]]), dy2 = (double)(float)(p[2 * idx[k] + 1] - p[2 * idx[i] + 1]);
      if (fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
    }
  }
  return 0;
}

/* getSubset: 7 distinct indices, redrawn as a whole while the subset fails checkSubset -- at most 10000 times from the RANSAC
 * registrator, getSubset's default 1000 from the LMedS one */
static int get_subset(const float *m1, const float *m2, int count, cv_rng *rng, int *idx, int max_attempts) {
  for (int iters = 0; iters < max_attempts; ++iters) {
    for (int i = 0; i < 7; ++i) {
      for (;;) {
        const int c = rng_uniform(rng, 0, count);
        int j = 0;
        while (j < i && idx[j] != c) ++j;
        if (j == i) { idx[i] = c; break; }
      }
    }
    if (collinear_last(m1, idx, 7) || collinear_last(m2, idx, 7)) continue;
    return 1;
  }
  return 0;
}

/* real roots of c[0] x^3 + c[1] x^2 + c[2] x + c[3], ascending, without libm */
static double cubic_at(double b, double c, double d, double x) { return ((x + b) * x + c) * x + d; }
static double bisect(double b, double c, double d, double lo, double hi) {
  double flo = cubic_at(b, c, d, lo);
  if (flo == 0.0) return lo;
  if (cubic_at(b, c, d, hi) == 0.0) return hi;
  for (int it = 0; it < 4000; ++it) {
    const double mid = lo + (hi - lo) * 0.5;
    if (mid == lo || mid == hi) return mid;
    const double fm = cubic_at(b, c, d, mid);
    if (fm == 0.0) return mid;
    if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else hi = mid;
  }
  return lo + (hi - lo) * 0.5;
}
static int solve_cubic(const double *cf, double *roots) {
  const double a = cf[0];
  if (a == 0.0) {
    if (cf[1] == 0.0) {
      if (cf[2] == 0.0) return 0;
      roots[0] = -cf[3] / cf[2];
      return 1;
    }
    const double disc = cf[2] * cf[2] - 4.0 * cf[1] * cf[3];
    if (disc < 0.0) return 0;
    const double s = sqrt(disc);
    const double r0 = (-cf[2] - s) / (2.0 * cf[1]), r1 = (-cf[2] + s) / (2.0 * cf[1]);
    if (disc == 0.0) { roots[0] = r0; return 1; }
    roots[0] = r0 < r1 ? r0 : r1; roots[1] = r0 < r1 ? r1 : r0;
    return 2;
  }
  const double b = cf[1] / a, c = cf[2] / a, d = cf[3] / a;
  double m = fabs(b);
  if (fabs(c) > m) m = fabs(c);
  if (fabs(d) > m) m = fabs(d);
  const double R = 1.0 + m;                   /* Cauchy bound: every real root lies in [-R, R] */
  const double dd = b * b - 3.0 * c;          /* discriminant / 4 of the derivative 3 x^2 + 2 b x + c */
  int n = 0;
  if (!(dd > 0.0)) {                          /* monotonic: one real root */
    roots[n++] = bisect(b, c, d, -R, R);
    return n;
  }
  const double s = sqrt(dd);
  const double x1 = (-b - s) / 3.0, x2 = (-b + s) / 3.0;
  const double brk[4] = {-R, x1, x2, R};
  for (int k = 0; k < 3; ++k) {
    const double lo = brk[k], hi = brk[k + 1];
    const double flo = cubic_at(b, c, d, lo), fhi = cubic_at(b, c, d, hi);
    if ((flo < 0.0 && fhi < 0.0) || (flo > 0.0 && fhi > 0.0)) continue;
    const double r = bisect(b, c, d, lo, hi);
    if (n == 0 || r != roots[n - 1]) roots[n++] = r;
  }
  return n;
}

/* run7Point: up to three fundamental matrices (row-major, F[8] = 1 where possible) through seven correspondences */
static int run_7point(const float *m1, const float *m2, const int *idx, double *F /* 27 */) {
  double A[7][9];
  for (int i = 0; i < 7; ++i) {
    const double x0 = m1[2 * idx[i]], y0 = m1[2 * idx[i] + 1], x1 = m2[2 * idx[i]], y1 = m2[2 * idx[i] + 1];
    A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
    A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
    A[i][6] = x0; A[i][7] = y0; A[i][8] = 1.0;
  }
  /* Gauss-Jordan with complete pivoting: perm[c] = original column now at position c */
  int perm[9];
  for (int c = 0; c < 9; ++c) perm[c] = c;
  for (int p = 0; p < 7; ++p) {
    int br = p, bc = p;
    double best = -1.0;
    for (int r = p; r < 7; ++r)
      for (int c = p; c < 9; ++c)
        if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); br = r; bc = c; }
    if (!(best > 0.0)) return 0;              /* rank below 7: no isolated solution */
    if (br != p) for (int c = 0; c < 9; ++c) { const double t = A[p][c]; A[p][c] = A[br][c]; A[br][c] = t; }
    if (bc != p) {
      for (int r = 0; r < 7; ++r) { const double t = A[r][p]; A[r][p] = A[r][bc]; A[r][bc] = t; }
      const int t = perm[p]; perm[p] = perm[bc]; perm[bc] = t;
    }
    const double inv = 1.0 / A[p][p];
    for (int c = p; c < 9; ++c) A[p][c] = A[p][c] * inv;
    for (int r = 0; r < 7; ++r) {
      if (r == p) continue;
      const double f = A[r][p];
      if (f == 0.0) continue;
      for (int c = p; c < 9; ++c) A[r][c] = A[r][c] - f * A[p][c];
    }
  }
  double f1[9], f2[9];
  for (int t = 0; t < 2; ++t) {
    double *f = t == 0 ? f1 : f2;
    double x[9];
    for (int r = 0; r < 7; ++r) x[r] = -A[r][7 + t];
    x[7] = t == 0 ? 1.0 : 0.0;
    x[8] = t == 0 ? 0.0 : 1.0;
    double ss = 0.0;
    for (int c = 0; c < 9; ++c) ss = ss + x[c] * x[c];
    const double inv = 1.0 / sqrt(ss);
    for (int c = 0; c < 9; ++c) f[perm[c]] = x[c] * inv;
  }
  for (int i = 0; i < 9; ++i) f1[i] = f1[i] - f2[i];
  double c[4], t0, t1, t2;
  t0 = f2[4] * f2[8] - f2[5] * f2[7]; t1 = f2[3] * f2[8] - f2[5] * f2[6]; t2 = f2[3] * f2[7] - f2[4] * f2[6];
  c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
  c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
         f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
         f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
  t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
  c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
         f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
         f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
  double roots[3];
  const int n = solve_cubic(c, roots);
  for (int k = 0; k < n; ++k) {
    double *Fk = F + 9 * k;
    double lambda = roots[k], mu = 1.0;
    const double s = f1[8] * roots[k] + f2[8];
    if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda = lambda * mu; Fk[8] = 1.0; }
    else Fk[8] = 0.0;
    for (int i = 0; i < 8; ++i) Fk[i] = f1[i] * lambda + f2[i] * mu;
  }
  return n;
}

/* FMEstimatorCallback::computeError for one correspondence */
static float epi_error(const double *F, const float *m1, const float *m2, int i) {
  const double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
  double a = F[0] * x1 + F[1] * y1 + F[2];
  double b = F[3] * x1 + F[4] * y1 + F[5];
  double c = F[6] * x1 + F[7] * y1 + F[8];
  const double s2 = 1.0 / (a * a + b * b);
  const double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  const double s1 = 1.0 / (a * a + b * b);
  const double d1 = x1 * a + y1 * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (float)(e1 > e2 ? e1 : e2);
}

/* RANSACUpdateNumIters(p, ep, 7, max_iters) in the libm-free form of the header comment */
static int update_num_iters(double p, double ep, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double w = 1.0 - ep;
  double w7 = w;
  for (int k = 0; k < 6; ++k) w7 = w7 * w;
  const double q = 1.0 - w7;               /* probability that a sample of 7 holds an outlier */
  if (q < DBL_MIN) return 0;
  if (!(q < 1.0)) return max_iters;        /* no inlier at all: log(q) = 0 */
  double acc = 1.0, prev = 1.0;
  int k = 0;
  while (k < max_iters && acc > num) { prev = acc; acc = acc * q; ++k; }
  if (acc > num) return max_iters;         /* the real solution is beyond max_iters */
  /* k = ceil(x), x = log(num) / log(q); cvRound(x) = k - 1 when x < k - 1/2, i.e. q^(k - 1/2) < num */
  if (k >= 1 && prev * sqrt(q) < num) k -= 1;
  return k;
}

/* cv::findFundamentalMat(m1, m2, FM_RANSAC, thresh, confidence, mask) -> number of inliers; mask[n] */
int ocv_find_fundamental_mask(const float *m1, const float *m2, int n, double thresh, double confidence, uint8_t *mask) {
  for (int i = 0; i < n; ++i) mask[i] = 1;
  if (n < 7) return n;                      /* (OpenCV: empty result, the reference then reads an empty mask) */
  if (thresh <= 0.0) thresh = 3.0;
  if (confidence < DBL_EPSILON || confidence > 1.0 - DBL_EPSILON) confidence = 0.99;
  int idx[7];
  double F[27], best[9];
  if (n == 7) return n;                     /* runKernel on the seven points; every point is an inlier by construction */
  cv_rng rng = {0xffffffffffffffffull};
  uint8_t *cur = (uint8_t *)malloc((size_t)n);
  float *err = (float *)malloc(sizeof(float) * (size_t)n);
  int result = 0;
  if (n >= 15) {
    const float t = (float)(thresh * thresh);
    int niters = 1000, max_good = 0;
    for (int iter = 0; iter < niters; ++iter) {
      if (!get_subset(m1, m2, n, &rng, idx, 10000)) { if (iter == 0) { free(cur); free(err); return n; } break; }
      const int nm = run_7point(m1, m2, idx, F);
      for (int k = 0; k < nm; ++k) {
        int good = 0;
        for (int i = 0; i < n; ++i) { cur[i] = epi_error(F + 9 * k, m1, m2, i) <= t; good += cur[i]; }
        if (good > (max_good > 6 ? max_good : 6)) {
          memcpy(mask, cur, (size_t)n);
          max_good = good;
          niters = update_num_iters(confidence, (double)(n - good) / n, niters);
        }
      }
    }
    result = max_good;
    if (max_good == 0) for (int i = 0; i < n; ++i) mask[i] = 1;     /* no model: findFundamentalMat returns empty, nothing is rejected */
  } else {
    /* LMeDSPointSetRegistrator::run, outlier ratio 0.45 */
    int niters = update_num_iters(confidence, 0.45, 1000);
    if (niters < 3) niters = 3;
    double min_median = DBL_MAX;
    for (int iter = 0; iter < niters; ++iter) {
      if (!get_subset(m1, m2, n, &rng, idx, 1000)) { if (iter == 0) { free(cur); free(err); return n; } break; }
      const int nm = run_7point(m1, m2, idx, F);
      for (int k = 0; k < nm; ++k) {
        for (int i = 0; i < n; ++i) err[i] = epi_error(F + 9 * k, m1, m2, i);
        for (int i = 1; i < n; ++i) {           /* std::sort of at most 14 non-negative floats */
          const float v = err[i];
          int j = i - 1;
          while (j >= 0 && err[j] > v) { err[j + 1] = err[j]; --j; }
          err[j + 1] = v;
        }
        const double median = (n % 2) ? (double)err[n / 2] : (double)(err[n / 2 - 1] + err[n / 2]) * 0.5;   /* (float sum, like errf.at<float>() + errf.at<float>()) */
        if (median < min_median) { min_median = median; memcpy(best, F + 9 * k, sizeof(best)); }
      }
    }
    if (min_median < DBL_MAX) {
      double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (n - 7)) * sqrt(min_median);
      if (sigma < 0.001) sigma = 0.001;
      const float t = (float)(sigma * sigma);
      int good = 0;
      for (int i = 0; i < n; ++i) { mask[i] = epi_error(best, m1, m2, i) <= t; good += mask[i]; }
      result = good;                                                /* (good < 7: run() reports failure, the mask it copied out stands) */
    }
  }
  free(cur); free(err);
  (void)result;
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += mask[i];
  return cnt;
}
