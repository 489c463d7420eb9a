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

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* sampler 0: the counter hash above; sampler 1: the reference's own stream -- the C library's srand(seed) and
   rand() through Random::RandomInt (:100-117), drawn serially like the reference (:56-71).  NOT reentrant
   (rand() is process-global, exactly the reference's situation). */
void oransac_minimal_sets(int sampler, uint32_t seed, int n, int iterations, int *sets) {
  if (sampler == 0) {
    for (int it = 0; it < iterations; ++it) draw_set(seed, it, n, sets + 8 * it);
    return;
  }
  srand(seed);
  int *avail = (int *)malloc(sizeof(int) * (size_t)n);
  for (int it = 0; it < iterations; ++it) {
    for (int i = 0; i < n; ++i) avail[i] = i;
    int size = n;
    for (int j = 0; j < 8; ++j) {
      const int d = (size - 1) - 0 + 1;
      const int randi = (int)(((double)rand() / ((double)RAND_MAX + 1.0)) * d) + 0;
      sets[8 * it + j] = avail[randi];
      avail[randi] = avail[size - 1];
      --size;
    }
  }
  free(avail);
}

/* hypotheses that count under a confidence-driven stop (spec: urf_oracle.h, oransac_config.confidence) */
static int confident_prefix(const float *score, const int *ninl, int n, int iters, double confidence) {
  int niters = iters;
  float best = 0.0f;
  for (int it = 0; it < iters && it < niters; ++it) {
    if (score[it] > best) {
      best = score[it];
      const double wr = (double)ninl[it] / (double)n;
      double w8 = wr * wr; w8 = w8 * w8; w8 = w8 * w8;
      const double q = 1.0 - w8, tgt = 1.0 - confidence;
      int k = 1;
      double acc = q;
      while (acc > tgt && k < iters) { acc = acc * q; ++k; }
      if (k < niters) niters = k;
    }
  }
  return niters;
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
  float *tx = (float *)calloc((size_t)n + 1, 4), *ty = (float *)calloc((size_t)n + 1, 4);
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
static float check_F_count(const float *F, const float *p0, const float *p1, int n, float sigma, uint8_t *inl, int *count);
static float check_F(const float *F, const float *p0, const float *p1, int n, float sigma, uint8_t *inl) {
  return check_F_count(F, p0, p1, n, sigma, inl, NULL);
}
static float check_F_count(const float *F, const float *p0, const float *p1, int n, float sigma,
                           uint8_t *inl, int *count) {
  const float f11 = F[0], f12 = F[1], f13 = F[2], f21 = F[3], f22 = F[4], f23 = F[5],
              f31 = F[6], f32 = F[7], f33 = F[8];
  const float th = 3.841f, thScore = 5.991f;
  const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
  float part[64];
  int cnt = 0;
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
      cnt += bIn;
    }
    part[l] = score;
  }
  if (count) *count = cnt;
  return om_bfly64_sum(part);
}

/* Canonical order of the correspondences (written spec, DESIGN.md "RANSAC"): the sampler, the
 * normalisation sums and the score sums walk the matches sorted by (x0, y0, x1, y1) -- float values
 * compared through their order-preserving integer images, original index as the last key -- so that the
 * result does not depend on the order in which the matcher lists them (keypoints are score-sorted, and
 * near-tied scores may swap between arithmetic variants).  The inlier flags are per correspondence and
 * come out in the caller's order. */
static uint32_t okey(float f) { uint32_t u = om_f2bits(f); return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u); }
typedef struct { uint32_t k[4]; int idx; } rs_ent;
static int rs_cmp(const void *a, const void *b) {
  const rs_ent *x = (const rs_ent *)a, *y = (const rs_ent *)b;
  for (int i = 0; i < 4; ++i)
    if (x->k[i] != y->k[i]) return x->k[i] < y->k[i] ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

static float find_F_impl(const float *pts0_in, const float *pts1_in, int n, const oransac_config *cfg, const int *sets,
                         uint8_t *inliers, float *F21) {
  for (int i = 0; i < n; ++i) inliers[i] = 0;
  for (int k = 0; k < 9; ++k) F21[k] = 0.0f;
  if (n < 8) return 0.0f;
  float *pts0 = (float *)malloc(8 * (size_t)n), *pts1 = (float *)malloc(8 * (size_t)n);
  if (sets) {            /* explicit sets index the caller's order: no canonical sort */
    memcpy(pts0, pts0_in, 8 * (size_t)n); memcpy(pts1, pts1_in, 8 * (size_t)n);
  } else {
    rs_ent *ent = (rs_ent *)malloc(sizeof(rs_ent) * (size_t)n);
    for (int i = 0; i < n; ++i) {
      ent[i].k[0] = okey(pts0_in[2 * i]); ent[i].k[1] = okey(pts0_in[2 * i + 1]);
      ent[i].k[2] = okey(pts1_in[2 * i]); ent[i].k[3] = okey(pts1_in[2 * i + 1]);
      ent[i].idx = i;
    }
    qsort(ent, (size_t)n, sizeof(rs_ent), rs_cmp);
    for (int i = 0; i < n; ++i) {
      pts0[2 * i] = pts0_in[2 * ent[i].idx]; pts0[2 * i + 1] = pts0_in[2 * ent[i].idx + 1];
      pts1[2 * i] = pts1_in[2 * ent[i].idx]; pts1[2 * i + 1] = pts1_in[2 * ent[i].idx + 1];
    }
    free(ent);
  }
  float *pn0 = (float *)malloc(8 * (size_t)n), *pn1 = (float *)malloc(8 * (size_t)n);
  float T1[9], T2[9], T2t[9];
  normalize_pts(pts0, n, pn0, T1);
  normalize_pts(pts1, n, pn1, T2);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2t[i * 3 + j] = T2[j * 3 + i];
  float best = 0.0f; int best_it = -1;
  float *Fall = (float *)malloc(sizeof(float) * 9 * (size_t)cfg->iterations);
  float *sc = (float *)malloc(sizeof(float) * (size_t)cfg->iterations);
  int *cnt = (int *)malloc(sizeof(int) * (size_t)cfg->iterations);
#pragma omp parallel for schedule(static)
  for (int it = 0; it < cfg->iterations; ++it) {
    int set[8];
    if (sets) memcpy(set, sets + 8 * (size_t)it, sizeof(set));
    else draw_set(cfg->seed, it, n, set);
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
    sc[it] = check_F_count(Fall + 9 * (size_t)it, pts0, pts1, n, cfg->sigma, NULL, cnt + it);
  }
  const int counted = (!sets && cfg->confidence > 0.0f) ? confident_prefix(sc, cnt, n, cfg->iterations, (double)cfg->confidence)
                                                        : cfg->iterations;
  for (int it = 0; it < counted; ++it)
    if (sc[it] > best) { best = sc[it]; best_it = it; } /* strict >, first wins :197 */
  if (best_it >= 0) {
    for (int k = 0; k < 9; ++k) F21[k] = Fall[9 * (size_t)best_it + k];
    check_F(F21, pts0_in, pts1_in, n, cfg->sigma, inliers);   /* per-point test, caller's order */
  }
  free(pn0); free(pn1); free(Fall); free(sc); free(cnt); free(pts0); free(pts1);
  return best;
}

float oransac_find_F(const float *pts0, const float *pts1, int n, const oransac_config *cfg, uint8_t *inliers, float *F21) {
  return find_F_impl(pts0, pts1, n, cfg, NULL, inliers, F21);
}
float oransac_find_F_sets(const float *pts0, const float *pts1, int n, const oransac_config *cfg, const int *sets,
                          uint8_t *inliers, float *F21) {
  return find_F_impl(pts0, pts1, n, cfg, sets, inliers, F21);
}

/* ======================================================================
 * EpipolarGeometry::reconstruct, src/epipolar_geometry.cc:18-98, with
 * _find_H :114-158, _compute_H21 :207-245, _check_H :285-370, _reconstruct_F
 * :451-562, _reconstruct_H :564-733, _check_R_T :782-898, _decompose_E :900-926,
 * _triangulate :928-950.  Eigen::JacobiSVD is replaced by cyclic Jacobi on the
 * Gram matrix in double (written spec, DESIGN.md "RANSAC"); float arithmetic
 * where the reference computes in float.
 * ====================================================================== */
static void mat3_inv_f(const float *m, float *o) { /* cofactor inverse, float */
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = (a * A + b * B) + c * C;
  const float id = 1.0f / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

/* _compute_H21 :207-245: smallest eigenvector of A^T A (A is 16x9), double */
static void compute_H21(const float *p1, const float *p2, double Hn[9]) {
  double A[16][9];
  for (int i = 0; i < 8; ++i) {
    const float u1 = p1[2 * i], v1 = p1[2 * i + 1], u2 = p2[2 * i], v2 = p2[2 * i + 1];
    double *r0 = A[2 * i], *r1 = A[2 * i + 1];
    r0[0] = 0.0; r0[1] = 0.0; r0[2] = 0.0; r0[3] = (double)(-u1); r0[4] = (double)(-v1); r0[5] = -1.0;
    r0[6] = (double)(v2 * u1); r0[7] = (double)(v2 * v1); r0[8] = (double)v2;
    r1[0] = (double)u1; r1[1] = (double)v1; r1[2] = 1.0; r1[3] = 0.0; r1[4] = 0.0; r1[5] = 0.0;
    r1[6] = (double)(-u2 * u1); r1[7] = (double)(-u2 * v1); r1[8] = (double)(-u2);
  }
  double ata[81], V[81];
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double s = 0.0;
      for (int i = 0; i < 16; ++i) s = s + A[i][r] * A[i][c];
      ata[r * 9 + c] = s;
    }
  jacobi_sym(ata, V, 9);
  const int m = argmin_diag(ata, 9);
  for (int k = 0; k < 9; ++k) Hn[k] = V[k * 9 + m];
}

static float check_H(const float *H21, const float *H12, const float *p0, const float *p1, int n, float sigma,
                     uint8_t *inl) {
  const float th = 5.991f;
  const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float score = 0.0f;
    for (int i = l; i < n; i += 64) {
      int bIn = 1;
      const float u1 = p0[2 * i], v1 = p0[2 * i + 1], u2 = p1[2 * i], v2 = p1[2 * i + 1];
      const float w2in1inv = (float)(1.0 / (double)((H12[6] * u2 + H12[7] * v2) + H12[8]));
      const float u2in1 = ((H12[0] * u2 + H12[1] * v2) + H12[2]) * w2in1inv;
      const float v2in1 = ((H12[3] * u2 + H12[4] * v2) + H12[5]) * w2in1inv;
      const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
      const float chiSquare1 = squareDist1 * invSigmaSquare;
      if (chiSquare1 > th) bIn = 0; else score = score + (th - chiSquare1);
      const float w1in2inv = (float)(1.0 / (double)((H21[6] * u1 + H21[7] * v1) + H21[8]));
      const float u1in2 = ((H21[0] * u1 + H21[1] * v1) + H21[2]) * w1in2inv;
      const float v1in2 = ((H21[3] * u1 + H21[4] * v1) + H21[5]) * w1in2inv;
      const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
      const float chiSquare2 = squareDist2 * invSigmaSquare;
      if (chiSquare2 > th) bIn = 0; else score = score + (th - chiSquare2);
      if (inl) inl[i] = (uint8_t)bIn;
    }
    part[l] = score;
  }
  return om_bfly64_sum(part);
}

/* normalisation statistics over ALL keypoints of an image (:735-780) */
static void normalize_stats(const float *keys, int n, float T[9], float *meanX, float *meanY, float *sX, float *sY) {
  float *tx = (float *)malloc(4 * (size_t)n), *ty = (float *)malloc(4 * (size_t)n);
  for (int i = 0; i < n; ++i) { tx[i] = keys[2 * i]; ty[i] = keys[2 * i + 1]; }
  *meanX = om_wave_sum(tx, n) / (float)n; *meanY = om_wave_sum(ty, n) / (float)n;
  for (int i = 0; i < n; ++i) { tx[i] = fabsf(keys[2 * i] - *meanX); ty[i] = fabsf(keys[2 * i + 1] - *meanY); }
  const float dX = om_wave_sum(tx, n) / (float)n, dY = om_wave_sum(ty, n) / (float)n;
  *sX = (float)(1.0 / (double)dX); *sY = (float)(1.0 / (double)dY);
  for (int k = 0; k < 9; ++k) T[k] = 0.0f;
  T[0] = *sX; T[4] = *sY; T[2] = -*meanX * *sX; T[5] = -*meanY * *sY; T[8] = 1.0f;
  free(tx); free(ty);
}

/* SVD of a 3x3 float matrix through Jacobi on A^T A (double): A = U diag(w) V^T,
   w descending; U columns = A v / w (third column = u0 x u1 when w2 is tiny). */
static void svd3(const float *Af, double U[9], double w[3], double V[9]) {
  double A[9], g[9], W[9];
  for (int k = 0; k < 9; ++k) A[k] = (double)Af[k];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s = s + A[k * 3 + r] * A[k * 3 + c];
      g[r * 3 + c] = s;
    }
  jacobi_sym(g, W, 3);
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (g[ord[b] * 3 + ord[b]] > g[ord[a] * 3 + ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
  for (int j = 0; j < 3; ++j) {
    const double ev = g[ord[j] * 3 + ord[j]];
    w[j] = ev > 0.0 ? sqrt(ev) : 0.0;
    for (int k = 0; k < 3; ++k) V[k * 3 + j] = W[k * 3 + ord[j]];
  }
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 3; ++r)
      U[r * 3 + j] = ((A[r * 3 + 0] * V[0 * 3 + j] + A[r * 3 + 1] * V[1 * 3 + j]) + A[r * 3 + 2] * V[2 * 3 + j]) / w[j];
  /* third left vector: u0 x u1 (exactly orthogonal even when w2 ~ 0, as for an
     essential matrix), oriented along A v2 */
  {
    const double c0 = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
    const double c1 = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
    const double c2 = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    double av[3];
    for (int r = 0; r < 3; ++r) av[r] = (A[r * 3 + 0] * V[0 * 3 + 2] + A[r * 3 + 1] * V[1 * 3 + 2]) + A[r * 3 + 2] * V[2 * 3 + 2];
    const double sgn = ((av[0] * c0 + av[1] * c1) + av[2] * c2) < 0.0 ? -1.0 : 1.0;
    U[0 * 3 + 2] = sgn * c0; U[1 * 3 + 2] = sgn * c1; U[2 * 3 + 2] = sgn * c2;
  }
}
static double det3(const double *m) {
  return (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* _triangulate :928-950: null vector of the 4x4 DLT matrix (Jacobi on A^T A) */
static int triangulate(const float *x1, const float *x2, const float *P1, const float *P2, float X[3]) {
  float Af[16];
  for (int c = 0; c < 4; ++c) {
    Af[0 * 4 + c] = x1[0] * P1[2 * 4 + c] - P1[0 * 4 + c];
    Af[1 * 4 + c] = x1[1] * P1[2 * 4 + c] - P1[1 * 4 + c];
    Af[2 * 4 + c] = x2[0] * P2[2 * 4 + c] - P2[0 * 4 + c];
    Af[3 * 4 + c] = x2[1] * P2[2 * 4 + c] - P2[1 * 4 + c];
  }
  double g[16], V[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s = s + (double)Af[k * 4 + r] * (double)Af[k * 4 + c];
      g[r * 4 + c] = s;
    }
  jacobi_sym(g, V, 4);
  const int m = argmin_diag(g, 4);
  const float h[4] = {(float)V[0 * 4 + m], (float)V[1 * 4 + m], (float)V[2 * 4 + m], (float)V[3 * 4 + m]};
  if (h[3] == 0.0f) return 0;
  X[0] = h[0] / h[3]; X[1] = h[1] / h[3]; X[2] = h[2] / h[3];
  return 1;
}

static int cmp_float(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* _check_R_T :782-898.  mpairs: nm (i1,i2) index pairs; inl: nm flags. */
static int check_R_T(const float *R, const float *t, const float *keys1, int n1, const float *keys2,
                     const int *mpairs, int nm, const uint8_t *inl, const float *K, float *P3D, float th2,
                     uint8_t *good, float *parallax) {
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  for (int i = 0; i < n1; ++i) good[i] = 0;
  float *cosv = (float *)malloc(4 * (size_t)(nm > 0 ? nm : 1));
  int ncos = 0;
  float P1[12], P2[12], Rt[12];
  for (int k = 0; k < 12; ++k) P1[k] = 0.0f;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P1[r * 4 + c] = K[r * 3 + c];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rt[r * 4 + c] = R[r * 3 + c]; Rt[r * 4 + 3] = t[r]; }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      P2[r * 4 + c] = (K[r * 3 + 0] * Rt[0 * 4 + c] + K[r * 3 + 1] * Rt[1 * 4 + c]) + K[r * 3 + 2] * Rt[2 * 4 + c];
  float O2[3];
  for (int r = 0; r < 3; ++r) O2[r] = -((R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1]) + R[2 * 3 + r] * t[2]);
  int nGood = 0;
  for (int i = 0; i < nm; ++i) {
    if (!inl[i]) continue;
    const int i1 = mpairs[2 * i], i2 = mpairs[2 * i + 1];
    const float x1[2] = {keys1[2 * i1], keys1[2 * i1 + 1]}, x2[2] = {keys2[2 * i2], keys2[2 * i2 + 1]};
    float p[3] = {0, 0, 0};
    triangulate(x1, x2, P1, P2, p);
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) { good[i1] = 0; continue; }
    const float dist1 = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
    const float n2[3] = {p[0] - O2[0], p[1] - O2[1], p[2] - O2[2]};
    const float dist2 = sqrtf((n2[0] * n2[0] + n2[1] * n2[1]) + n2[2] * n2[2]);
    const float cosParallax = ((p[0] * n2[0] + p[1] * n2[1]) + p[2] * n2[2]) / (dist1 * dist2);
    if (p[2] <= 0 && cosParallax < 0.99998f) continue;
    float q[3];
    for (int r = 0; r < 3; ++r) q[r] = ((R[r * 3 + 0] * p[0] + R[r * 3 + 1] * p[1]) + R[r * 3 + 2] * p[2]) + t[r];
    if (q[2] <= 0 && cosParallax < 0.99998f) continue;
    const float invZ1 = (float)(1.0 / (double)p[2]);
    const float im1x = fx * p[0] * invZ1 + cx, im1y = fy * p[1] * invZ1 + cy;
    const float e1 = (im1x - x1[0]) * (im1x - x1[0]) + (im1y - x1[1]) * (im1y - x1[1]);
    if (e1 > th2) continue;
    const float invZ2 = (float)(1.0 / (double)q[2]);
    const float im2x = fx * q[0] * invZ2 + cx, im2y = fy * q[1] * invZ2 + cy;
    const float e2 = (im2x - x2[0]) * (im2x - x2[0]) + (im2y - x2[1]) * (im2y - x2[1]);
    if (e2 > th2) continue;
    cosv[ncos++] = cosParallax;
    P3D[3 * i1] = p[0]; P3D[3 * i1 + 1] = p[1]; P3D[3 * i1 + 2] = p[2];
    nGood++;
    if (cosParallax < 0.99998f) good[i1] = 1;
  }
  if (nGood > 0) {
    qsort(cosv, ncos, sizeof(float), cmp_float);
    const int idx = 50 < ncos - 1 ? 50 : ncos - 1;
    *parallax = (float)(acos((double)cosv[idx]) * 180.0 / 3.1415926535897932384626433832795);
  } else {
    *parallax = 0.0f;
  }
  free(cosv);
  return nGood;
}



/* returns 1 on success. model: 0 = homography, 1 = fundamental.  SH/SF scores out. */
int oepi_reconstruct(const oepi_config *cfg, const float *keys1, int n1, const float *keys2, int n2,
                     const int *matches12, float *T21, float *P3D, uint8_t *tri, int *model, float *scores) {
  return oepi_reconstruct_sets(cfg, keys1, n1, keys2, n2, matches12, NULL, T21, P3D, tri, model, scores);
}
int oepi_reconstruct_sets(const oepi_config *cfg, const float *keys1, int n1, const float *keys2, int n2,
                          const int *matches12, const int *sets_in, float *T21, float *P3D, uint8_t *tri, int *model,
                          float *scores) {
  for (int k = 0; k < 16; ++k) T21[k] = (k % 5 == 0) ? 1.0f : 0.0f;
  for (int i = 0; i < n1; ++i) tri[i] = 0;
  int *mp = (int *)malloc(sizeof(int) * 2 * (size_t)(n1 > 0 ? n1 : 1));
  int nm = 0;
  for (int i = 0; i < n1; ++i) if (matches12[i] >= 0) { mp[2 * nm] = i; mp[2 * nm + 1] = matches12[i]; ++nm; }
  *model = -1; scores[0] = scores[1] = 0.0f;
  if (nm < 8) { free(mp); return 0; }
  float T1[9], T2[9], mX1, mY1, sX1, sY1, mX2, mY2, sX2, sY2;
  normalize_stats(keys1, n1, T1, &mX1, &mY1, &sX1, &sY1);
  normalize_stats(keys2, n2, T2, &mX2, &mY2, &sX2, &sY2);
  float *p0 = (float *)malloc(8 * (size_t)nm), *p1 = (float *)malloc(8 * (size_t)nm);
  float *q0 = (float *)malloc(8 * (size_t)nm), *q1 = (float *)malloc(8 * (size_t)nm);
  for (int i = 0; i < nm; ++i) {
    p0[2 * i] = keys1[2 * mp[2 * i]]; p0[2 * i + 1] = keys1[2 * mp[2 * i] + 1];
    p1[2 * i] = keys2[2 * mp[2 * i + 1]]; p1[2 * i + 1] = keys2[2 * mp[2 * i + 1] + 1];
    q0[2 * i] = (p0[2 * i] - mX1) * sX1; q0[2 * i + 1] = (p0[2 * i + 1] - mY1) * sY1;
    q1[2 * i] = (p1[2 * i] - mX2) * sX2; q1[2 * i + 1] = (p1[2 * i + 1] - mY2) * sY2;
  }
  float T2t[9], T2inv[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2t[i * 3 + j] = T2[j * 3 + i];
  mat3_inv_f(T2, T2inv);
  const int its = cfg->iterations;
  int *own_sets = NULL;
  if (!sets_in && cfg->sampler == 1) {   /* the reference's stream, drawn serially before the searches (:56-71) */
    own_sets = (int *)malloc(sizeof(int) * 8 * (size_t)its);
    oransac_minimal_sets(1, cfg->seed, nm, its, own_sets);
    sets_in = own_sets;
  }
  float *Fall = (float *)malloc(36 * (size_t)its), *Hall = (float *)malloc(36 * (size_t)its), *Hinv = (float *)malloc(36 * (size_t)its);
  float *scF = (float *)malloc(4 * (size_t)its), *scH = (float *)malloc(4 * (size_t)its);
#pragma omp parallel for schedule(static)
  for (int it = 0; it < its; ++it) {
    int set[8];
    if (sets_in) memcpy(set, sets_in + 8 * (size_t)it, sizeof(set));
    else draw_set(cfg->seed, it, nm, set);
    float a[16], b[16];
    for (int j = 0; j < 8; ++j) {
      a[2 * j] = q0[2 * set[j]]; a[2 * j + 1] = q0[2 * set[j] + 1];
      b[2 * j] = q1[2 * set[j]]; b[2 * j + 1] = q1[2 * set[j] + 1];
    }
    double Fn[9], Hn[9];
    float Xf[9], M[9];
    compute_F21(a, b, Fn);
    for (int k = 0; k < 9; ++k) Xf[k] = (float)Fn[k];
    mat3_mul_f(T2t, Xf, M); mat3_mul_f(M, T1, Fall + 9 * (size_t)it);
    scF[it] = check_F(Fall + 9 * (size_t)it, p0, p1, nm, cfg->sigma, NULL);
    compute_H21(a, b, Hn);
    for (int k = 0; k < 9; ++k) Xf[k] = (float)Hn[k];
    mat3_mul_f(T2inv, Xf, M); mat3_mul_f(M, T1, Hall + 9 * (size_t)it);   /* H21i = T2inv*Hn*T1 :148 */
    mat3_inv_f(Hall + 9 * (size_t)it, Hinv + 9 * (size_t)it);
    scH[it] = check_H(Hall + 9 * (size_t)it, Hinv + 9 * (size_t)it, p0, p1, nm, cfg->sigma, NULL);
  }
  float SF = 0.0f, SH = 0.0f; int bF = -1, bH = -1;
  for (int it = 0; it < its; ++it) { if (scF[it] > SF) { SF = scF[it]; bF = it; } if (scH[it] > SH) { SH = scH[it]; bH = it; } }
  scores[0] = SH; scores[1] = SF;
  int ok = 0;
  uint8_t *inl = (uint8_t *)calloc(nm, 1);
  uint8_t *gd = (uint8_t *)malloc(n1 > 0 ? n1 : 1);
  float *P = (float *)calloc(3 * (size_t)(n1 > 0 ? n1 : 1), 4);
  const float minParallax = 1.0f; const int minTri = 50;
  const float th2 = 4.0f * (cfg->sigma * cfg->sigma);
  if (SH + SF != 0.0f) {
    const float RH = SH / (SH + SF);
    if (RH > 0.50f && bH >= 0) { /* _reconstruct_H :564-733 */
      *model = 0;
      const float *H21 = Hall + 9 * (size_t)bH;
      check_H(H21, Hinv + 9 * (size_t)bH, p0, p1, nm, cfg->sigma, inl);
      int N = 0; for (int i = 0; i < nm; ++i) N += inl[i];
      float invK[9], M[9], A[9];
      mat3_inv_f(cfg->K, invK); mat3_mul_f(invK, H21, M); mat3_mul_f(M, cfg->K, A);
      double U[9], w[3], V[9];
      svd3(A, U, w, V);
      double Vt[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
      const float s = (float)(det3(U) * det3(Vt));
      const float d1 = (float)w[0], d2 = (float)w[1], d3 = (float)w[2];
      if (!(d1 / d2 < 1.00001f || d2 / d3 < 1.00001f)) {
        float Rs[8][9], ts[8][3];
        const float aux1 = sqrtf((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3));
        const float aux3 = sqrtf((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));
        const float x1[4] = {aux1, aux1, -aux1, -aux1}, x3[4] = {aux3, -aux3, aux3, -aux3};
        const float aux_st = sqrtf((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2);
        const float ctheta = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2);
        const float stheta[4] = {aux_st, -aux_st, -aux_st, aux_st};
        const float aux_sp = sqrtf((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2);
        const float cphi = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2);
        const float sphi[4] = {aux_sp, -aux_sp, -aux_sp, aux_sp};
        float Uf[9], Vtf[9];
        for (int k = 0; k < 9; ++k) { Uf[k] = (float)U[k]; Vtf[k] = (float)Vt[k]; }
        for (int h8 = 0; h8 < 8; ++h8) {
          const int i = h8 & 3; const int second = h8 >= 4;
          float Rp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tp[3];
          if (!second) { Rp[0] = ctheta; Rp[2] = -stheta[i]; Rp[4] = 1.0f; Rp[6] = stheta[i]; Rp[8] = ctheta;
                         tp[0] = x1[i] * (d1 - d3); tp[1] = 0.0f; tp[2] = -x3[i] * (d1 - d3); }
          else { Rp[0] = cphi; Rp[2] = sphi[i]; Rp[4] = -1.0f; Rp[6] = sphi[i]; Rp[8] = -cphi;
                 tp[0] = x1[i] * (d1 + d3); tp[1] = 0.0f; tp[2] = x3[i] * (d1 + d3); }
          float M1[9], M2[9];
          mat3_mul_f(Uf, Rp, M1); mat3_mul_f(M1, Vtf, M2);
          for (int k = 0; k < 9; ++k) Rs[h8][k] = s * M2[k];
          float tt[3];
          for (int r = 0; r < 3; ++r) tt[r] = (Uf[r * 3] * tp[0] + Uf[r * 3 + 1] * tp[1]) + Uf[r * 3 + 2] * tp[2];
          const float nrm = sqrtf((tt[0] * tt[0] + tt[1] * tt[1]) + tt[2] * tt[2]);
          for (int r = 0; r < 3; ++r) ts[h8][r] = tt[r] / nrm;
        }
        int bestGood = 0, second = 0, bestIdx = -1; float bestPar = -1.0f;
        uint8_t *bg = (uint8_t *)malloc(n1 > 0 ? n1 : 1); float *bP = (float *)calloc(3 * (size_t)(n1 > 0 ? n1 : 1), 4);
        for (int h8 = 0; h8 < 8; ++h8) {
          float par; memset(P, 0, 12 * (size_t)n1);
          const int nG = check_R_T(Rs[h8], ts[h8], keys1, n1, keys2, mp, nm, inl, cfg->K, P, th2, gd, &par);
          if (nG > bestGood) { second = bestGood; bestGood = nG; bestIdx = h8; bestPar = par; memcpy(bg, gd, n1); memcpy(bP, P, 12 * (size_t)n1); }
          else if (nG > second) second = nG;
        }
        if (getenv("OEPI_DEBUG")) fprintf(stderr, "H: N=%d best=%d second=%d par=%g d=%g %g %g s=%g\n", N, bestGood, second, bestPar, d1, d2, d3, s);
        if (second < 0.75 * bestGood && bestPar >= minParallax && bestGood > minTri && bestGood > 0.9 * N) {
          for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T21[r * 4 + c] = Rs[bestIdx][r * 3 + c]; T21[r * 4 + 3] = ts[bestIdx][r]; }
          memcpy(tri, bg, n1); memcpy(P3D, bP, 12 * (size_t)n1);
          ok = 1;
        }
        free(bg); free(bP);
      }
    } else if (bF >= 0) { /* _reconstruct_F :451-562 */
      *model = 1;
      const float *F21 = Fall + 9 * (size_t)bF;
      check_F(F21, p0, p1, nm, cfg->sigma, inl);
      int N = 0; for (int i = 0; i < nm; ++i) N += inl[i];
      float Kt[9], M[9], E[9];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Kt[i * 3 + j] = cfg->K[j * 3 + i];
      mat3_mul_f(Kt, F21, M); mat3_mul_f(M, cfg->K, E);
      double U[9], w[3], V[9];
      svd3(E, U, w, V);                                           /* _decompose_E :900-926 */
      float Uf[9], Vtf[9], t[3];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Uf[i * 3 + j] = (float)U[i * 3 + j]; Vtf[i * 3 + j] = (float)V[j * 3 + i]; }
      { const float nrm = sqrtf((Uf[2] * Uf[2] + Uf[5] * Uf[5]) + Uf[8] * Uf[8]); t[0] = Uf[2] / nrm; t[1] = Uf[5] / nrm; t[2] = Uf[8] / nrm; }
      const float Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
      float R1[9], R2[9], M1[9];
      mat3_mul_f(Uf, Wm, M1); mat3_mul_f(M1, Vtf, R1);
      mat3_mul_f(Uf, Wt, M1); mat3_mul_f(M1, Vtf, R2);
      { double d[9]; for (int k = 0; k < 9; ++k) d[k] = R1[k]; if (det3(d) < 0) for (int k = 0; k < 9; ++k) R1[k] = -R1[k]; }
      { double d[9]; for (int k = 0; k < 9; ++k) d[k] = R2[k]; if (det3(d) < 0) for (int k = 0; k < 9; ++k) R2[k] = -R2[k]; }
      const float t2[3] = {-t[0], -t[1], -t[2]};
      const float *Rc[4] = {R1, R2, R1, R2}; const float *tc[4] = {t, t, t2, t2};
      int nG[4]; float par[4];
      uint8_t *gds = (uint8_t *)malloc(4 * (size_t)(n1 > 0 ? n1 : 1)); float *Ps = (float *)calloc(12 * (size_t)(n1 > 0 ? n1 : 1), 4);
      for (int c = 0; c < 4; ++c)
        nG[c] = check_R_T(Rc[c], tc[c], keys1, n1, keys2, mp, nm, inl, cfg->K, Ps + 3 * (size_t)n1 * c, th2, gds + (size_t)n1 * c, &par[c]);
      if (getenv("OEPI_DEBUG")) { fprintf(stderr, "F: N=%d nG=%d %d %d %d par=%g %g %g %g w=%g %g %g t=%g %g %g\n", N, nG[0], nG[1], nG[2], nG[3], par[0], par[1], par[2], par[3], w[0], w[1], w[2], t[0], t[1], t[2]); }
      int maxGood = nG[0]; for (int c = 1; c < 4; ++c) if (nG[c] > maxGood) maxGood = nG[c];
      const int nMinGood = (int)(0.9 * N) > minTri ? (int)(0.9 * N) : minTri;
      int nsimilar = 0; for (int c = 0; c < 4; ++c) if (nG[c] > 0.7 * maxGood) nsimilar++;
      if (!(maxGood < nMinGood || nsimilar > 1)) {
        for (int c = 0; c < 4; ++c)
          if (maxGood == nG[c]) {
            if (par[c] > minParallax) {
              for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) T21[r * 4 + cc] = Rc[c][r * 3 + cc]; T21[r * 4 + 3] = tc[c][r]; }
              memcpy(tri, gds + (size_t)n1 * c, n1); memcpy(P3D, Ps + 3 * (size_t)n1 * c, 12 * (size_t)n1);
              ok = 1;
            }
            break;
          }
      }
      free(gds); free(Ps);
    }
  }
  free(own_sets);
  free(mp); free(p0); free(p1); free(q0); free(q1); free(Fall); free(Hall); free(Hinv); free(scF); free(scH);
  free(inl); free(gd); free(P);
  return ok;
}
