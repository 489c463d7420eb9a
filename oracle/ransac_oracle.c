/* ransac_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the epipolar
 * 8-point RANSAC, following EpipolarGeometry::_find_F / _normalize /
 * _compute_F21 / _check_F, src/epipolar_geometry.cc:161-205, 735-780, 247-283,
 * 372-449, and the minimal-set sampler of ::reconstruct :56-71.
 *
 * The reference's outlier stage is cv::findFundamentalMat
 * (src/point_matching.cc:50): OpenCV 4.2 is an un-vendored dependency (not in
 * the reference tree, not in this image) -> PARITY UNPINNED for that call; the
 * build follows the in-tree, fully specified ORB-SLAM3-derived routine instead.
 *
 * Deviations that are part of the written spec (DESIGN.md "RANSAC"):
 *  - glibc rand() after a process-global srand(0) is replaced by a counter
 *    hash rs_hash(seed, 8*it+j) so hypotheses are reproducible and parallel;
 *  - Eigen::JacobiSVD is replaced by a fully pivoted Gaussian elimination of the
 *    8x9 design matrix in double (null vector) and a cyclic Jacobi eigen-solver
 *    on F^T F (rank-2 projection F - (F v)v^T);
 *  - float sums over the matches use the canonical wave-strided order.
 */
#include "urf_oracle.h"
#include "oracle_math.h"

#include <stdlib.h>

static uint32_t rs_hash(uint32_t seed, uint32_t ctr) {
  uint32_t x = seed ^ (ctr * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
/* Random::RandomInt :114-117 with rand() -> 31-bit counter hash */
static int random_int(uint32_t seed, uint32_t ctr, int mn, int mx) {
  const int d = mx - mn + 1;
  const uint32_t r = rs_hash(seed, ctr) >> 1;
  return (int)(((double)r / 2147483648.0) * d) + mn;
}

/* minimal set :59-71: draw without replacement by swap-with-back, tracked with
   a sparse map of the (<=8) modified slots instead of the N-long array */
static void draw_set(uint32_t seed, int it, int n, int set[8]) {
  int mpos[8], mval[8], nm = 0;
  for (int j = 0; j < 8; ++j) {
    const int size = n - j;
    const int randi = random_int(seed, (uint32_t)(it * 8 + j), 0, size - 1);
    int idx = randi, back = size - 1;
    for (int t = 0; t < nm; ++t) if (mpos[t] == randi) idx = mval[t];
    for (int t = 0; t < nm; ++t) if (mpos[t] == size - 1) back = mval[t];
    set[j] = idx;
    int found = 0;
    for (int t = 0; t < nm; ++t) if (mpos[t] == randi) { mval[t] = back; found = 1; }
    if (!found) { mpos[nm] = randi; mval[nm] = back; ++nm; }
  }
}

/* cyclic Jacobi on a symmetric n x n (n<=9) matrix, double; V accumulates the
   rotations (columns = eigenvectors).  Fixed sweep count => data-independent
   control flow except the tiny-pivot skip. */
#define JAC_SWEEPS 12
static void jacobi_sym(double *a, double *v, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < JAC_SWEEPS; ++sweep) {
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { /* columns p,q */
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) { /* rows p,q */
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}
static int argmin_diag(const double *a, int n) {
  int m = 0;
  for (int i = 1; i < n; ++i) if (a[i * n + i] < a[m * n + m]) m = i;
  return m;
}

/* null vector of the 8x9 design matrix by Gaussian elimination with full
   pivoting (f64).  For 8 points in general position rank(A) = 8 and the null
   vector equals the smallest right-singular vector the reference takes from
   Eigen::JacobiSVD (:267-273), up to sign.  Pivot search order: rows s..7 outer,
   columns s..8 inner, strict '>' (first maximum wins). */
static void null_vector_8x9(double A[8][9], double f[9]) {
  int perm[9];
  for (int c = 0; c < 9; ++c) perm[c] = c;
  int rank = 8;
  for (int s = 0; s < 8; ++s) {
    double best = 0.0; int pr = -1, pc = -1;
    for (int r = s; r < 8; ++r)
      for (int c = s; c < 9; ++c) {
        const double v = fabs(A[r][c]);
        if (v > best) { best = v; pr = r; pc = c; }
      }
    if (pr < 0) { rank = s; break; }
    if (pr != s) for (int c = 0; c < 9; ++c) { const double t = A[s][c]; A[s][c] = A[pr][c]; A[pr][c] = t; }
    if (pc != s) {
      for (int r = 0; r < 8; ++r) { const double t = A[r][s]; A[r][s] = A[r][pc]; A[r][pc] = t; }
      const int t = perm[s]; perm[s] = perm[pc]; perm[pc] = t;
    }
    const double piv = A[s][s];
    for (int r = s + 1; r < 8; ++r) {
      const double m = A[r][s] / piv;
      A[r][s] = 0.0;
      for (int c = s + 1; c < 9; ++c) A[r][c] = A[r][c] - m * A[s][c];
    }
  }
  double g[9];
  for (int c = 0; c < 9; ++c) g[c] = 0.0;
  g[rank] = 1.0;
  for (int s = rank - 1; s >= 0; --s) {
    double sum = 0.0;
    for (int c = s + 1; c < 9; ++c) sum = sum + A[s][c] * g[c];
    g[s] = -sum / A[s][s];
  }
  double ss = 0.0;
  for (int c = 0; c < 9; ++c) ss = ss + g[c] * g[c];
  const double inv = 1.0 / sqrt(ss);
  for (int c = 0; c < 9; ++c) f[perm[c]] = g[c] * inv;
}

/* _compute_F21 :247-283 on 8 normalised pairs -> Fn (row-major, double) */
static void compute_F21(const float *p1, const float *p2, double Fn[9]) {
  double A[8][9];
  for (int i = 0; i < 8; ++i) {
    const float u1 = p1[2 * i], v1 = p1[2 * i + 1], u2 = p2[2 * i], v2 = p2[2 * i + 1];
    /* A entries are float products in the reference (:257-265) */
    A[i][0] = (double)(u2 * u1); A[i][1] = (double)(u2 * v1); A[i][2] = (double)u2;
    A[i][3] = (double)(v2 * u1); A[i][4] = (double)(v2 * v1); A[i][5] = (double)v2;
    A[i][6] = (double)u1;        A[i][7] = (double)v1;        A[i][8] = 1.0;
  }
  double Fpre[9];
  null_vector_8x9(A, Fpre); /* row-major 3x3 (:273) */
  /* rank-2 enforcement (:275-282): F - (F v) v^T, v = weakest right-singular
     vector of F = eigenvector of F^T F (cyclic Jacobi, 3x3) */
  double g[9], W[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s = s + Fpre[k * 3 + r] * Fpre[k * 3 + c];
      g[r * 3 + c] = s;
    }
  jacobi_sym(g, W, 3);
  const int m3 = argmin_diag(g, 3);
  double vv[3] = {W[0 * 3 + m3], W[1 * 3 + m3], W[2 * 3 + m3]}, fv[3];
  for (int r = 0; r < 3; ++r)
    fv[r] = (Fpre[r * 3 + 0] * vv[0] + Fpre[r * 3 + 1] * vv[1]) + Fpre[r * 3 + 2] * vv[2];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Fn[r * 3 + c] = Fpre[r * 3 + c] - fv[r] * vv[c];
}

static void mat3_mul_f(const float *a, const float *b, float *o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      o[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}

/* _normalize :735-780 with the canonical wave-strided float sums */
static void normalize_pts(const float *pts, int n, float *out, float T[9]) {
  float *tx = (float *)malloc(4 * (size_t)n), *ty = (float *)malloc(4 * (size_t)n);
  for (int i = 0; i < n; ++i) { tx[i] = pts[2 * i]; ty[i] = pts[2 * i + 1]; }
  const float meanX = om_wave_sum(tx, n) / (float)n, meanY = om_wave_sum(ty, n) / (float)n;
  for (int i = 0; i < n; ++i) {
    out[2 * i] = pts[2 * i] - meanX; out[2 * i + 1] = pts[2 * i + 1] - meanY;
    tx[i] = fabsf(out[2 * i]); ty[i] = fabsf(out[2 * i + 1]);
  }
  const float meanDevX = om_wave_sum(tx, n) / (float)n, meanDevY = om_wave_sum(ty, n) / (float)n;
  const float sX = (float)(1.0 / (double)meanDevX), sY = (float)(1.0 / (double)meanDevY);
  for (int i = 0; i < n; ++i) { out[2 * i] = out[2 * i] * sX; out[2 * i + 1] = out[2 * i + 1] * sY; }
  for (int k = 0; k < 9; ++k) T[k] = 0.0f;
  T[0] = sX; T[4] = sY; T[2] = -meanX * sX; T[5] = -meanY * sY; T[8] = 1.0f;
  free(tx); free(ty);
}

/* _check_F :372-449, per-match terms; score summed in canonical wave order:
   lane l = matches l, l+64, ... each adding its chi terms in source order. */
static float check_F(const float *F, const float *p0, const float *p1, int n, float sigma,
                     uint8_t *inl) {
  const float f11 = F[0], f12 = F[1], f13 = F[2], f21 = F[3], f22 = F[4], f23 = F[5],
              f31 = F[6], f32 = F[7], f33 = F[8];
  const float th = 3.841f, thScore = 5.991f;
  const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float score = 0.0f;
    for (int i = l; i < n; i += 64) {
      int bIn = 1;
      const float u1 = p0[2 * i], v1 = p0[2 * i + 1], u2 = p1[2 * i], v2 = p1[2 * i + 1];
      const float a2 = (f11 * u1 + f12 * v1) + f13;
      const float b2 = (f21 * u1 + f22 * v1) + f23;
      const float c2 = (f31 * u1 + f32 * v1) + f33;
      const float num2 = (a2 * u2 + b2 * v2) + c2;
      const float squareDist1 = (num2 * num2) / (a2 * a2 + b2 * b2);
      const float chiSquare1 = squareDist1 * invSigmaSquare;
      if (chiSquare1 > th) bIn = 0; else score = score + (thScore - chiSquare1);
      const float a1 = (f11 * u2 + f21 * v2) + f31;
      const float b1 = (f12 * u2 + f22 * v2) + f32;
      const float c1 = (f13 * u2 + f23 * v2) + f33;
      const float num1 = (a1 * u1 + b1 * v1) + c1;
      const float squareDist2 = (num1 * num1) / (a1 * a1 + b1 * b1);
      const float chiSquare2 = squareDist2 * invSigmaSquare;
      if (chiSquare2 > th) bIn = 0; else score = score + (thScore - chiSquare2);
      if (inl) inl[i] = (uint8_t)bIn;
    }
    part[l] = score;
  }
  return om_bfly64_sum(part);
}

float oransac_find_F(const float *pts0, const float *pts1, int n, const oransac_config *cfg,
                     uint8_t *inliers, float *F21) {
  for (int i = 0; i < n; ++i) inliers[i] = 0;
  for (int k = 0; k < 9; ++k) F21[k] = 0.0f;
  if (n < 8) return 0.0f;
  float *pn0 = (float *)malloc(8 * (size_t)n), *pn1 = (float *)malloc(8 * (size_t)n);
  float T1[9], T2[9], T2t[9];
  normalize_pts(pts0, n, pn0, T1);
  normalize_pts(pts1, n, pn1, T2);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2t[i * 3 + j] = T2[j * 3 + i];
  float best = 0.0f; int best_it = -1;
  float *Fall = (float *)malloc(sizeof(float) * 9 * (size_t)cfg->iterations);
  float *sc = (float *)malloc(sizeof(float) * (size_t)cfg->iterations);
#pragma omp parallel for schedule(static)
  for (int it = 0; it < cfg->iterations; ++it) {
    int set[8];
    draw_set(cfg->seed, it, n, set);
    float a[16], b[16];
    for (int j = 0; j < 8; ++j) {
      a[2 * j] = pn0[2 * set[j]]; a[2 * j + 1] = pn0[2 * set[j] + 1];
      b[2 * j] = pn1[2 * set[j]]; b[2 * j + 1] = pn1[2 * set[j] + 1];
    }
    double Fn[9];
    compute_F21(a, b, Fn);
    float Fnf[9], M[9];
    for (int k = 0; k < 9; ++k) Fnf[k] = (float)Fn[k];
    mat3_mul_f(T2t, Fnf, M);                 /* F21i = T2t * Fn * T1 :193 */
    mat3_mul_f(M, T1, Fall + 9 * (size_t)it);
    sc[it] = check_F(Fall + 9 * (size_t)it, pts0, pts1, n, cfg->sigma, NULL);
  }
  for (int it = 0; it < cfg->iterations; ++it)
    if (sc[it] > best) { best = sc[it]; best_it = it; } /* strict >, first wins :197 */
  if (best_it >= 0) {
    for (int k = 0; k < 9; ++k) F21[k] = Fall[9 * (size_t)best_it + k];
    check_F(F21, pts0, pts1, n, cfg->sigma, inliers);
  }
  free(pn0); free(pn1); free(Fall); free(sc);
  return best;
}
