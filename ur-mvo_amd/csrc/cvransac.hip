// cvransac.hip -- the outlier stage as the reference's OWN call (opt-in: urf_sg_config.outlier_stage = 1):
//   cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask)          src/point_matching.cc:43-58
// OpenCV (4.2.0 in the reference's image) is an un-vendored dependency and absent here; this is the published algorithm of
// modules/calib3d/src/fundam.cpp + ptsetreg.cpp + cv::RNG restated: 7-point minimal sets drawn by the multiply-with-carry
// generator seeded with (uint64)-1, the collinearity test of the last drawn point, up to three models per set, the symmetric
// epipolar distance against (float)9, "strictly more inliers and at least 7", the iteration count shrunk by
// RANSACUpdateNumIters, at most 1000 iterations, no refit; LMedS below 15 points; the mask of the best hypothesis.
// PARITY UNPINNED (no OpenCV binary to compare with).  Where OpenCV calls its numerical library (SVD, solveCubic, log / pow) a
// fixed libm-free arithmetic is used instead, written down in DESIGN.md section 13: the null space of the 7 x 9 system by
// Gauss-Jordan elimination with complete pivoting, the cubic's real roots by bracketing between the critical points and
// bisection to the last bit, RANSACUpdateNumIters by a multiplication chain.  The CPU checker of the tests implements the same
// text a second time and agrees bit for bit (tests/test_gpu_parity.py); it agrees in turn with an independent numpy / LAPACK
// restatement of the algorithm over the same generator stream (tests/golden/make_cvransac_golden.py).
//
// The loop is sequential by construction (the generator's stream, and an iteration count that depends on every earlier
// result): ONE wave per pair walks it; lane 0 draws the set and solves the 7-point system in f64, all 64 lanes count the
// inliers of a model over the pair's matches.  A good pair stops after ~10 iterations (~0.1 ms); a pair without a consensus
// runs all 1000.
#include <float.h>

#include "urf_common.h"

namespace urf {
namespace {

struct DMatchC { int queryIdx, trainIdx; float distance; };

struct CvRng {
  unsigned long long state;
  __device__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + (unsigned)a); }
};

// (the differences are taken in float -- Point2f minus Point2f -- and widened afterwards, as haveCollinearPoints does)
__device__ bool cv_collinear_last(const float *p, const int *idx) {
  const float xi = p[2 * idx[6]], yi = p[2 * idx[6] + 1];
  for (int j = 0; j < 6; ++j) {
    const double dx1 = (double)(p[2 * idx[j]] - xi), dy1 = (double)(p[2 * idx[j] + 1] - yi);
    for (int k = 0; k < j; ++k) {
      const double dx2 = (double)(p[2 * idx[k]] - xi), dy2 = (double)(p[2 * idx[k] + 1] - yi);
      if (fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}

// max_attempts: 10000 from RANSACPointSetRegistrator::run, getSubset's default 1000 from LMeDSPointSetRegistrator::run
__device__ bool cv_get_subset(const float *m1, const float *m2, int count, CvRng &rng, int *idx, int max_attempts) {
  for (int attempt = 0; attempt < max_attempts; ++attempt) {
    for (int i = 0; i < 7; ++i) {
      for (;;) {
        const int c = rng.uniform(0, count);
        bool dup = false;
        for (int j = 0; j < i; ++j) dup = dup || idx[j] == c;
        if (!dup) { idx[i] = c; break; }
      }
    }
    if (cv_collinear_last(m1, idx) || cv_collinear_last(m2, idx)) continue;
    return true;
  }
  return false;
}

__device__ double cv_poly(double b, double c, double d, double x) { return ((x + b) * x + c) * x + d; }
__device__ double cv_bisect(double b, double c, double d, double lo, double hi) {
  double flo = cv_poly(b, c, d, lo);
  if (flo == 0.0) return lo;
  if (cv_poly(b, c, d, hi) == 0.0) return hi;
  for (int it = 0; it < 4000; ++it) {
    const double mid = lo + (hi - lo) * 0.5;
    if (mid == lo || mid == hi) return mid;
    const double fm = cv_poly(b, c, d, mid);
    if (fm == 0.0) return mid;
    if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else hi = mid;
  }
  return lo + (hi - lo) * 0.5;
}
__device__ int cv_solve_cubic(const double *cf, double *roots) {
  if (cf[0] == 0.0) {
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
  const double b = cf[1] / cf[0], c = cf[2] / cf[0], d = cf[3] / cf[0];
  double m = fabs(b);
  if (fabs(c) > m) m = fabs(c);
  if (fabs(d) > m) m = fabs(d);
  const double R = 1.0 + m;
  const double dd = b * b - 3.0 * c;
  if (!(dd > 0.0)) { roots[0] = cv_bisect(b, c, d, -R, R); return 1; }
  const double s = sqrt(dd);
  const double brk[4] = {-R, (-b - s) / 3.0, (-b + s) / 3.0, R};
  int n = 0;
  for (int k = 0; k < 3; ++k) {
    const double flo = cv_poly(b, c, d, brk[k]), fhi = cv_poly(b, c, d, brk[k + 1]);
    if ((flo < 0.0 && fhi < 0.0) || (flo > 0.0 && fhi > 0.0)) continue;
    const double r = cv_bisect(b, c, d, brk[k], brk[k + 1]);
    if (n == 0 || r != roots[n - 1]) roots[n++] = r;
  }
  return n;
}

// A: 7 x 9 in LDS (runtime-indexed; one lane works on it), F: up to 3 x 9 in LDS
__device__ int cv_run_7point(const float *m1, const float *m2, const int *idx, double (*A)[9], double *F) {
  for (int i = 0; i < 7; ++i) {
    const double x0 = m1[2 * idx[i]], y0 = m1[2 * idx[i] + 1], x1 = m2[2 * idx[i]], y1 = m2[2 * idx[i] + 1];
    A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
    A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
    A[i][6] = x0; A[i][7] = y0; A[i][8] = 1.0;
  }
  int perm[9];
  for (int c = 0; c < 9; ++c) perm[c] = c;
  for (int p = 0; p < 7; ++p) {
    int br = p, bc = p;
    double best = -1.0;
    for (int r = p; r < 7; ++r)
      for (int c = p; c < 9; ++c)
        if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); br = r; bc = c; }
    if (!(best > 0.0)) return 0;
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
    double x[9];
    for (int r = 0; r < 7; ++r) x[r] = -A[r][7 + t];
    x[7] = t == 0 ? 1.0 : 0.0;
    x[8] = t == 0 ? 0.0 : 1.0;
    double ss = 0.0;
    for (int c = 0; c < 9; ++c) ss = ss + x[c] * x[c];
    const double inv = 1.0 / sqrt(ss);
    // (scatter through the column permutation; perm lives in registers, so a select chain instead of a runtime index)
    for (int c = 0; c < 9; ++c) {
      const double v = x[c] * inv;
      for (int o = 0; o < 9; ++o)
        if (perm[c] == o) { if (t == 0) f1[o] = v; else f2[o] = v; }
    }
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
  const int n = cv_solve_cubic(c, roots);
  for (int k = 0; k < n; ++k) {
    double lambda = roots[k], mu = 1.0;
    const double s = f1[8] * roots[k] + f2[8];
    double f8 = 0.0;
    if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda = lambda * mu; f8 = 1.0; }
    for (int i = 0; i < 8; ++i) F[9 * k + i] = f1[i] * lambda + f2[i] * mu;
    F[9 * k + 8] = f8;
  }
  return n;
}

__device__ float cv_epi_error(const double *F, const float *m1, const float *m2, int i) {
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

__device__ int cv_update_num_iters(double p, double ep, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double w = 1.0 - ep;
  double w7 = w;
  for (int k = 0; k < 6; ++k) w7 = w7 * w;
  const double q = 1.0 - w7;
  if (q < DBL_MIN) return 0;
  if (!(q < 1.0)) return max_iters;
  double acc = 1.0, prev = 1.0;
  int k = 0;
  while (k < max_iters && acc > num) { prev = acc; acc = acc * q; ++k; }
  if (acc > num) return max_iters;
  if (k >= 1 && prev * sqrt(q) < num) k -= 1;
  return k;
}

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// one wave per pair.  mask / cur: [P][NP] bytes of scratch; out lists in match order
__global__ void __launch_bounds__(64) cv_ransac_kernel(const int *nmatch, const float *pts0, const float *pts1, double thresh,
                                                       double confidence, int enable, const DMatchC *matches, DMatchC *out,
                                                       int *nout, uint8_t *mask_g, uint8_t *cur_g, double *F_out, int *iters_out) {
  __shared__ double sA[7][9];
  __shared__ double sF[27];
  __shared__ double sBest[9];
  __shared__ int sIdx[7];
  __shared__ int sCtl[4];          // [0] models of this iteration (-1: stop), [1] scratch
  __shared__ float sErr[16];
  const int p = blockIdx.x, lane = threadIdx.x;
  const int n = nmatch[p];
  const float *m1 = pts0 + (size_t)p * kCap * 2, *m2 = pts1 + (size_t)p * kCap * 2;
  uint8_t *mask = mask_g + (size_t)p * kCap, *cur = cur_g + (size_t)p * kCap;
  for (int i = lane; i < n; i += 64) mask[i] = 1;
  bool reject = enable && n > 7;
  if (reject) {
    CvRng rng;
    rng.state = 0xffffffffffffffffull;
    if (n >= 15) {
      const float t = (float)(thresh * thresh);
      int niters = 1000, max_good = 0, iter = 0;
      for (; iter < niters; ++iter) {
        if (lane == 0) {
          int idx[7];
          int nm = -1;
          if (cv_get_subset(m1, m2, n, rng, idx, 10000)) {
            for (int i = 0; i < 7; ++i) sIdx[i] = idx[i];
            nm = cv_run_7point(m1, m2, idx, sA, sF);
          }
          sCtl[0] = nm;
        }
        __syncthreads();
        const int nm = sCtl[0];
        if (nm < 0) { if (iter == 0) max_good = -1; break; }
        for (int k = 0; k < nm; ++k) {
          int good = 0;
          for (int i = lane; i < n; i += 64) {
            const uint8_t in = cv_epi_error(sF + 9 * k, m1, m2, i) <= t;
            cur[i] = in;
            good += in;
          }
          good = wave_sum_int(good);
          if (good > (max_good > 6 ? max_good : 6)) {
            for (int i = lane; i < n; i += 64) mask[i] = cur[i];
            max_good = good;
            niters = cv_update_num_iters(confidence, (double)(n - good) / n, niters);
            if (F_out && lane < 9) F_out[9 * p + lane] = sF[9 * k + lane];   // (diagnostic output of urf_cv_find_fundamental: the model behind the mask)
          }
        }
        __syncthreads();
      }
      if (iters_out && lane == 0) iters_out[p] = iter;
      if (max_good <= 0)                      // no model (or no admissible subset at all): nothing is rejected
        for (int i = lane; i < n; i += 64) mask[i] = 1;
    } else {
      // LMeDSPointSetRegistrator::run (fewer than 15 points), outlier ratio 0.45
      int niters = cv_update_num_iters(confidence, 0.45, 1000);
      if (niters < 3) niters = 3;
      double min_median = DBL_MAX;
      bool failed = false;
      int iter = 0;
      for (; iter < niters; ++iter) {
        if (lane == 0) {
          int idx[7];
          int nm = -1;
          if (cv_get_subset(m1, m2, n, rng, idx, 1000)) nm = cv_run_7point(m1, m2, idx, sA, sF);
          sCtl[0] = nm;
        }
        __syncthreads();
        const int nm = sCtl[0];
        if (nm < 0) { failed = iter == 0; break; }
        for (int k = 0; k < nm; ++k) {
          if (lane < n) sErr[lane] = cv_epi_error(sF + 9 * k, m1, m2, lane);
          __syncthreads();
          if (lane == 0) {
            for (int i = 1; i < n; ++i) {      // at most 14 non-negative floats
              const float v = sErr[i];
              int j = i - 1;
              while (j >= 0 && sErr[j] > v) { sErr[j + 1] = sErr[j]; --j; }
              sErr[j + 1] = v;
            }
          }
          __syncthreads();
          const double median = (n % 2) ? (double)sErr[n / 2] : (double)(sErr[n / 2 - 1] + sErr[n / 2]) * 0.5;
          if (median < min_median) {
            min_median = median;
            if (lane < 9) sBest[lane] = sF[9 * k + lane];
          }
          __syncthreads();
        }
      }
      if (iters_out && lane == 0) iters_out[p] = iter;
      if (F_out && !failed && min_median < DBL_MAX && lane < 9) F_out[9 * p + lane] = sBest[lane];
      if (!failed && min_median < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (n - 7)) * sqrt(min_median);
        if (sigma < 0.001) sigma = 0.001;
        const float t = (float)(sigma * sigma);
        // (fewer than 7 inliers: run() reports failure and findFundamentalMat returns an empty matrix, but the mask has been
        // copied out by then -- the reference filters with it as it stands)
        if (lane < n) mask[lane] = cv_epi_error(sBest, m1, m2, lane) <= t;
      }
    }
  }
  __syncthreads();
  // ordered compaction of the match list by the mask (src/point_matching.cc:53-57)
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    const int keep = (i < n) ? mask[i] : 0;
    const unsigned long long bal = __ballot(keep);
    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep) out[(size_t)p * kCap + pos] = matches[(size_t)p * kCap + i];
    base += __popcll(bal);
  }
  if (lane == 0) nout[p] = base;
}

}  // namespace

// the outlier stage of a batch of P pairs in OpenCV's form.  inliers: [P][kCap] (the mask), scratch: [P][kCap] bytes
int launch_cv_ransac(const int *nmatch, const float *pts0, const float *pts1, double thresh, double confidence, int enable,
                     const void *matches, void *out, int *nout, uint8_t *inliers, uint8_t *scratch, int P, hipStream_t st,
                     double *F_out, int *iters_out) {
  hipLaunchKernelGGL(cv_ransac_kernel, dim3(P), dim3(64), 0, st, nmatch, pts0, pts1, thresh, confidence, enable,
                     (const DMatchC *)matches, (DMatchC *)out, nout, inliers, scratch, F_out, iters_out);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf

// cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, thresh, confidence, mask) on raw point arrays (host in, host out): the
// kernel of the matcher's outlier stage over ONE correspondence list.  What src/point_matching.cc:50 calls, behind the C ABI;
// F9 (row-major, the model behind the mask; untouched when there is none) and iterations are optional.
extern "C" int urf_cv_find_fundamental(const float *pts0, const float *pts1, int n, double thresh, double confidence,
                                       uint8_t *mask, double *F9, int *iterations, int device) {
  using namespace urf;
  URF_CHECK(n >= 0 && n <= kCap && (n == 0 || (pts0 && pts1 && mask)), "urf_cv_find_fundamental: 0 <= n <= %d points, non-null arrays", kCap);
  if (n == 0) { if (iterations) *iterations = 0; return 0; }
  if (thresh <= 0.0) thresh = 3.0;
  if (!(confidence > 0.0 && confidence < 1.0)) confidence = 0.99;
  URF_HIP(hipSetDevice(device));
  char *d = nullptr;
  // [nmatch | iters | F 9 f64 | pts0 | pts1 | matches | out | nout | mask | scratch], every block 16-byte aligned
  const size_t o_n = 0, o_it = 16, o_F = 32, o_p0 = 32 + 80, o_p1 = o_p0 + sizeof(float) * 2 * kCap, o_m = o_p1 + sizeof(float) * 2 * kCap,
               o_out = o_m + sizeof(DMatchC) * kCap, o_no = o_out + sizeof(DMatchC) * kCap, o_mask = o_no + 16, o_cur = o_mask + kCap,
               total = o_cur + kCap;
  URF_HIP(hipMalloc((void **)&d, total));
  int rc = 0;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == 0) { set_error("urf_cv_find_fundamental: %s", hipGetErrorString(e)); rc = -1; } };
  fail(hipMemset(d, 0, total));
  fail(hipMemcpy(d + o_n, &n, sizeof(int), hipMemcpyHostToDevice));
  fail(hipMemcpy(d + o_p0, pts0, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
  fail(hipMemcpy(d + o_p1, pts1, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
  double F0[9];
  for (int i = 0; i < 9; ++i) F0[i] = F9 ? F9[i] : 0.0;
  fail(hipMemcpy(d + o_F, F0, sizeof(F0), hipMemcpyHostToDevice));
  if (rc == 0 && launch_cv_ransac((const int *)(d + o_n), (const float *)(d + o_p0), (const float *)(d + o_p1), thresh, confidence, 1,
                                  d + o_m, d + o_out, (int *)(d + o_no), (uint8_t *)(d + o_mask), (uint8_t *)(d + o_cur), 1, 0,
                                  (double *)(d + o_F), (int *)(d + o_it)))
    rc = -1;
  fail(hipDeviceSynchronize());
  if (rc == 0) {
    fail(hipMemcpy(mask, d + o_mask, (size_t)n, hipMemcpyDeviceToHost));
    if (F9) fail(hipMemcpy(F9, d + o_F, sizeof(double) * 9, hipMemcpyDeviceToHost));
    if (iterations) fail(hipMemcpy(iterations, d + o_it, sizeof(int), hipMemcpyDeviceToHost));
  }
  (void)hipFree(d);
  return rc;
}
