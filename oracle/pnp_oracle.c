/* pnp_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the per-frame pose stage that follows the
 * front-end:
 *   SolvePnPWithCV      src/g2o_optimization.cc:323-377  (cv::solvePnPRansac(obj, img, K, dist, rvec, tvec,
 *                       false, 100, 20.0, 0.99, inliers) + the Twc assembly :356-368)
 *   FrameOptimization   src/g2o_optimization.cc:179-321  (pose-only Levenberg-Marquardt over
 *                       EdgeSE3ProjectXYZOnlyPose with a Huber kernel, 4 rounds x 10 iterations, chi-square
 *                       re-classification of the observations between rounds)
 *
 * PARITY UNPINNED: cv::solvePnPRansac (OpenCV 4.2) and g2o (HEAD of master at image build, docker/Dockerfile:145)
 * are un-vendored third-party code, absent from the reference tree and from this image, and the reference
 * holds no test vector for either.  What is restated here is therefore a WRITTEN SPECIFICATION (DESIGN.md,
 * "Pose stage") that keeps every parameter of the reference's call sites:
 *   PnP-RANSAC: <= 100 hypotheses, 20 px reprojection gate, confidence 0.99, at least 8 correspondences
 *     (:352), result as Twc (:363-367), inlier list of the best hypothesis.  Minimal solver: 6-point DLT on
 *     Hartley-normalised object points (OpenCV: 5-point EPnP), null vector by cyclic Jacobi on the 12x12
 *     Gram matrix, nearest rotation by the polar factor; counter-hash sampler; final refinement of the
 *     winner on its inliers by the same Levenberg-Marquardt core (OpenCV: SOLVEPNP_ITERATIVE).
 *   FrameOptimization: the edge of g2o's EdgeSE3ProjectXYZOnlyPose (error = obs - project(T Xw), its
 *     analytic Jacobian, update T <- exp(dx) T with dx = (omega, upsilon)), Huber kernel with
 *     delta = sqrt(chi2 gate) on rounds 0-2 and none on round 3 (:289-290), every round restarted from the
 *     INPUT pose (:266-267), observations with chi2 > gate excluded from the next round and re-tested
 *     (:272-283), rounds stop when fewer than 10 observations exist (:309-310); damping as in g2o's
 *     OptimizationAlgorithmLevenberg (lambda0 = 1e-5 max diag, gain ratio, lambda *= max(1/3, 1-(2 rho-1)^3)
 *     or lambda *= nu, nu *= 2, at most 10 trials per iteration).
 * All arithmetic is f64 with sums over the observations in the canonical wave order (lane l adds j = l, l+64, ...
 * then the 64-lane butterfly), sin/cos from the polynomial kernels below: the HIP path reproduces it bit for bit. */
#include "urf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ canonical f64 helpers */
static double d_bfly64(double p[64]) {
  double q[64];
  for (int s = 32; s >= 1; s >>= 1) {
    for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ s];
    for (int l = 0; l < 64; ++l) p[l] = q[l];
  }
  return p[0];
}

/* sin and cos of x for |x| < ~1e5: Cody-Waite reduction by pi/2 in three parts, then the fdlibm
   kernel polynomials evaluated by Horner's rule without fma */
static void sincos_c(double x, double *s, double *c) {
  const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050650619224932e-11, P3 = 2.02226624879595063154e-21;
  const double n = rint(x * 6.36619772367581382433e-01);
  const double r = ((x - n * P1) - n * P2) - n * P3;
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double sr = r + r * (z * ps);
  const double cr = (1.0 - 0.5 * z) + z * (z * pc);
  const long k = (long)n & 3;
  if (k == 0) { *s = sr; *c = cr; }
  else if (k == 1) { *s = cr; *c = -sr; }
  else if (k == 2) { *s = -sr; *c = -cr; }
  else { *s = -cr; *c = sr; }
}

/* quaternion (w, x, y, z) */
static void q_mul(const double a[4], const double b[4], double o[4]) {
  o[0] = ((a[0] * b[0] - a[1] * b[1]) - a[2] * b[2]) - a[3] * b[3];
  o[1] = ((a[0] * b[1] + a[1] * b[0]) + a[2] * b[3]) - a[3] * b[2];
  o[2] = ((a[0] * b[2] - a[1] * b[3]) + a[2] * b[0]) + a[3] * b[1];
  o[3] = ((a[0] * b[3] + a[1] * b[2]) - a[2] * b[1]) + a[3] * b[0];
}
static void q_normalize(double q[4]) {
  const double n = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] = q[k] / n;
  if (q[0] < 0.0) for (int k = 0; k < 4; ++k) q[k] = -q[k];
}
static void q_to_R(const double q[4], double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}
/* rotation matrix -> quaternion (Shepperd's branches, largest component first) */
static void R_to_q(const double R[9], double q[4]) {
  const double tr = (R[0] + R[4]) + R[8];
  if (tr > 0.0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(((1.0 + R[0]) - R[4]) - R[8]) * 2.0;
    q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(((1.0 + R[4]) - R[0]) - R[8]) * 2.0;
    q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(((1.0 + R[8]) - R[0]) - R[4]) * 2.0;
    q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
  }
  q_normalize(q);
}

/* T <- exp(dx) T, dx = (omega, upsilon): g2o's SE3Quat::exp and VertexSE3Expmap::oplusImpl.  T = (q, t) */
static void se3_apply_update(const double dx[6], double q[4], double t[3]) {
  const double wx = dx[0], wy = dx[1], wz = dx[2];
  const double th2 = (wx * wx + wy * wy) + wz * wz, th = sqrt(th2);
  double a, b, cq, sq_over;   /* a = sin(th)/th, b = (1-cos th)/th^2, c = (th - sin th)/th^3 */
  double c;
  if (th < 1e-5) {
    a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0;
    cq = 1.0 - th2 / 8.0; sq_over = 0.5 - th2 / 48.0;
  } else {
    double s, co, sh, ch;
    sincos_c(th, &s, &co);
    sincos_c(0.5 * th, &sh, &ch);
    a = s / th; b = (1.0 - co) / th2; c = (th - s) / (th2 * th);
    cq = ch; sq_over = sh / th;
  }
  const double dq[4] = {cq, sq_over * wx, sq_over * wy, sq_over * wz};
  /* R(dq) from Rodrigues with the same a, b:  R = I + a [w]x + b [w]x^2 ; V = I + b [w]x + c [w]x^2 */
  const double W[9] = {0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0};
  double W2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W2[i * 3 + j] = (W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j]) + W[i * 3 + 2] * W[6 + j];
  double Rd[9], V[9];
  for (int k = 0; k < 9; ++k) {
    const double id = (k % 4 == 0) ? 1.0 : 0.0;
    Rd[k] = (id + a * W[k]) + b * W2[k];
    V[k] = (id + b * W[k]) + c * W2[k];
  }
  double tn[3];
  for (int i = 0; i < 3; ++i) {
    const double rt = (Rd[i * 3] * t[0] + Rd[i * 3 + 1] * t[1]) + Rd[i * 3 + 2] * t[2];
    const double vu = (V[i * 3] * dx[3] + V[i * 3 + 1] * dx[4]) + V[i * 3 + 2] * dx[5];
    tn[i] = rt + vu;
  }
  double qn[4];
  q_mul(dq, q, qn);
  q_normalize(qn);
  memcpy(q, qn, sizeof(qn));
  memcpy(t, tn, sizeof(tn));
}

/* (H + lambda I) x = b for the symmetric positive definite 6x6 H: Cholesky, row by row */
static int solve6(const double H[36], double lambda, const double b[6], double x[6]) {
  double L[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = H[i * 6 + j] + (i == j ? lambda : 0.0);
      for (int k = 0; k < j; ++k) s = s - L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return 0;
        L[i * 6 + i] = sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s = s - L[i * 6 + k] * y[k];
    y[i] = s / L[i * 6 + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s = s - L[k * 6 + i] * x[k];
    x[i] = s / L[i * 6 + i];
  }
  return 1;
}

typedef struct { double fx, fy, cx, cy; } cam4;

/* error = obs - project(R X + t) and, optionally, its Jacobian (EdgeSE3ProjectXYZOnlyPose::linearizeOplus) */
static void edge_eval(const cam4 *cam, const double R[9], const double t[3], const double *X, const double *obs,
                      double e[2], double J[12]) {
  const double x = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
  const double y = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
  const double z = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
  const double iz = 1.0 / z;
  e[0] = obs[0] - (x * iz * cam->fx + cam->cx);
  e[1] = obs[1] - (y * iz * cam->fy + cam->cy);
  if (J) {
    const double iz2 = iz * iz;
    J[0] = x * y * iz2 * cam->fx;           J[1] = -(1.0 + x * x * iz2) * cam->fx; J[2] = y * iz * cam->fx;
    J[3] = -iz * cam->fx;                   J[4] = 0.0;                            J[5] = x * iz2 * cam->fx;
    J[6] = (1.0 + y * y * iz2) * cam->fy;   J[7] = -x * y * iz2 * cam->fy;         J[8] = -x * iz * cam->fy;
    J[9] = 0.0;                             J[10] = -iz * cam->fy;                 J[11] = y * iz2 * cam->fy;
  }
}

/* robust chi2 over the active observations, and optionally H = sum w J^T J, b = -sum w J^T e (w = Huber weight) */
static double build_system(const cam4 *cam, const double q[4], const double t[3], const double *Xw, const double *obs, int n,
                           const uint8_t *active, double delta, double *H, double *b) {
  double R[9];
  q_to_R(q, R);
  double part[28][64];
  for (int l = 0; l < 64; ++l) {
    double acc[28];
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for (int j = l; j < n; j += 64) {
      if (!active[j]) continue;
      double e[2], J[12];
      edge_eval(cam, R, t, Xw + 3 * j, obs + 2 * j, e, H ? J : NULL);
      const double e2 = e[0] * e[0] + e[1] * e[1];
      double rho = e2, w = 1.0;
      if (delta > 0.0) {
        const double en = sqrt(e2);
        if (en > delta) { rho = 2.0 * en * delta - delta * delta; w = delta / en; }
      }
      acc[27] = acc[27] + rho;
      if (H) {
        int k = 0;
        for (int r = 0; r < 6; ++r)
          for (int c = r; c < 6; ++c) { acc[k] = acc[k] + w * (J[r] * J[c] + J[6 + r] * J[6 + c]); ++k; }
        for (int r = 0; r < 6; ++r) acc[21 + r] = acc[21 + r] - w * (J[r] * e[0] + J[6 + r] * e[1]);
      }
    }
    for (int k = 0; k < 28; ++k) part[k][l] = acc[k];
  }
  if (H) {
    int k = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r; c < 6; ++c) { const double v = d_bfly64(part[k]); H[r * 6 + c] = v; H[c * 6 + r] = v; ++k; }
    for (int r = 0; r < 6; ++r) b[r] = d_bfly64(part[21 + r]);
  }
  return d_bfly64(part[27]);
}

/* `iterations` Levenberg-Marquardt iterations on T = (q, t) (camera-from-world), g2o's damping policy */
static void lm_pose(const cam4 *cam, const double *Xw, const double *obs, int n, const uint8_t *active, double delta,
                    int iterations, double q[4], double t[3]) {
  double lambda = 0.0, ni = 2.0;
  for (int it = 0; it < iterations; ++it) {
    double H[36], b[6];
    double current = build_system(cam, q, t, Xw, obs, n, active, delta, H, b);
    if (it == 0) {
      double md = 0.0;
      for (int k = 0; k < 6; ++k) md = fmax(md, fabs(H[k * 6 + k]));
      lambda = 1e-5 * md;
      ni = 2.0;
    }
    double rho = 0.0;
    int qmax = 0;
    do {
      double qb[4], tb[3], dx[6];
      memcpy(qb, q, sizeof(qb)); memcpy(tb, t, sizeof(tb));
      const int ok = solve6(H, lambda, b, dx);
      double temp = 1.7976931348623157e308;
      if (ok) {
        se3_apply_update(dx, q, t);
        temp = build_system(cam, q, t, Xw, obs, n, active, delta, NULL, NULL);
      }
      double scale = 1e-3;
      if (ok) for (int k = 0; k < 6; ++k) scale = scale + dx[k] * (lambda * dx[k] + b[k]);
      rho = (current - temp) / scale;
      if (ok && rho > 0.0 && isfinite(temp)) {
        double alpha = 1.0 - ((2.0 * rho - 1.0) * (2.0 * rho - 1.0)) * (2.0 * rho - 1.0);
        if (alpha > 2.0 / 3.0) alpha = 2.0 / 3.0;
        lambda = lambda * fmax(1.0 / 3.0, alpha);
        ni = 2.0;
        current = temp;
      } else {
        lambda = lambda * ni;
        ni = ni * 2.0;
        memcpy(q, qb, sizeof(qb)); memcpy(t, tb, sizeof(tb));
        if (!isfinite(lambda)) break;
      }
      ++qmax;
    } while (rho < 0.0 && qmax < 10);
    if (qmax == 10 || rho == 0.0 || !isfinite(lambda)) break;
  }
}

/* Twc (q, p) <-> Tcw (q, t) */
static void invert_pose(const double q[4], const double p[3], double qi[4], double ti[3]) {
  qi[0] = q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = -q[3];
  double R[9];
  q_to_R(qi, R);
  for (int i = 0; i < 3; ++i) ti[i] = -((R[i * 3] * p[0] + R[i * 3 + 1] * p[1]) + R[i * 3 + 2] * p[2]);
}

/* FrameOptimization, src/g2o_optimization.cc:179-321 (mono edges).  q_wc (w,x,y,z), p_wc: in = prior, out = optimised.
   inlier[n]: in = the caller's flags (MonoPointConstraint::inlier), out = re-classified.  Returns n - outliers. */
int oframe_optimization(const oposeopt_config *cfg, const double *Xw, const double *obs, int n, double *q_wc, double *p_wc,
                        uint8_t *inlier) {
  const cam4 cam = {cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  const double delta = sqrt(cfg->chi2_threshold);
  double q0[4], t0[3], q[4], t[3];
  double qn[4] = {q_wc[0], q_wc[1], q_wc[2], q_wc[3]};
  q_normalize(qn);
  invert_pose(qn, p_wc, q0, t0);
  memcpy(q, q0, sizeof(q)); memcpy(t, t0, sizeof(t));
  uint8_t *level0 = (uint8_t *)malloc(n > 0 ? n : 1);       /* edges on optimisation level 0 (all of them at first) */
  for (int j = 0; j < n; ++j) level0[j] = 1;
  int outliers = 0;
  for (int round = 0; round < 4; ++round) {
    memcpy(q, q0, sizeof(q)); memcpy(t, t0, sizeof(t));     /* :266-267 every round restarts from the prior */
    lm_pose(&cam, Xw, obs, n, level0, round < 3 ? delta : 0.0, 10, q, t);
    double R[9];
    q_to_R(q, R);
    outliers = 0;
    for (int j = 0; j < n; ++j) {
      double e[2];
      edge_eval(&cam, R, t, Xw + 3 * j, obs + 2 * j, e, NULL);
      const float chi2 = (float)(e[0] * e[0] + e[1] * e[1]);           /* :279 const float chi2 */
      if (chi2 > cfg->chi2_threshold) { inlier[j] = 0; level0[j] = 0; ++outliers; }
      else { inlier[j] = 1; level0[j] = 1; }
    }
    if (n < 10) break;
  }
  free(level0);
  double qo[4], po[3];
  invert_pose(q, t, qo, po);
  q_normalize(qo);
  memcpy(q_wc, qo, sizeof(qo)); memcpy(p_wc, po, sizeof(po));
  return n - outliers;
}

/* ------------------------------------------------------------------ PnP-RANSAC */
static uint32_t pnp_hash(uint32_t seed, uint32_t ctr) {
  uint32_t x = seed ^ (ctr * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
static void draw6(uint32_t seed, int it, int n, int set[6]) {
  int mpos[6], mval[6], nm = 0;
  for (int j = 0; j < 6; ++j) {
    const int size = n - j;
    const uint32_t r = pnp_hash(seed, (uint32_t)(it * 8 + j)) >> 1;
    const int randi = (int)(((double)r / 2147483648.0) * size);
    int idx = randi, back = size - 1;
    for (int k = 0; k < nm; ++k) if (mpos[k] == randi) idx = mval[k];
    for (int k = 0; k < nm; ++k) if (mpos[k] == size - 1) back = mval[k];
    set[j] = idx;
    int found = 0;
    for (int k = 0; k < nm; ++k) if (mpos[k] == randi) { mval[k] = back; found = 1; }
    if (!found) { mpos[nm] = randi; mval[nm] = back; ++nm; }
  }
}

static void jacobi_n(double *a, double *v, int n, int sweeps) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < sweeps; ++sweep)
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double tn = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tn * tn + 1.0), s = tn * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
        }
      }
}

/* one hypothesis: 6-point DLT.  Returns 0 when the sample is degenerate.  R (row-major), t: camera-from-world. */
static int dlt6(const double *X6 /*6x3*/, const double *xn6 /*6x2 normalised image coords*/, double R[9], double t[3]) {
  double c[3] = {0, 0, 0};
  for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) c[k] = c[k] + X6[3 * i + k];
  for (int k = 0; k < 3; ++k) c[k] = c[k] / 6.0;
  double md = 0.0;
  for (int i = 0; i < 6; ++i) {
    const double dx = X6[3 * i] - c[0], dy = X6[3 * i + 1] - c[1], dz = X6[3 * i + 2] - c[2];
    md = md + sqrt((dx * dx + dy * dy) + dz * dz);
  }
  md = md / 6.0;
  if (!(md > 0.0)) return 0;
  const double s = 1.0 / md;
  double A[12][12];
  for (int i = 0; i < 6; ++i) {
    const double X = (X6[3 * i] - c[0]) * s, Y = (X6[3 * i + 1] - c[1]) * s, Z = (X6[3 * i + 2] - c[2]) * s;
    const double u = xn6[2 * i], v = xn6[2 * i + 1];
    double *r0 = A[2 * i], *r1 = A[2 * i + 1];
    r0[0] = X; r0[1] = Y; r0[2] = Z; r0[3] = 1.0; r0[4] = 0; r0[5] = 0; r0[6] = 0; r0[7] = 0;
    r0[8] = -u * X; r0[9] = -u * Y; r0[10] = -u * Z; r0[11] = -u;
    r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = 0; r1[4] = X; r1[5] = Y; r1[6] = Z; r1[7] = 1.0;
    r1[8] = -v * X; r1[9] = -v * Y; r1[10] = -v * Z; r1[11] = -v;
  }
  double G[144], V[144];
  for (int r = 0; r < 12; ++r)
    for (int cc = 0; cc < 12; ++cc) {
      double acc = 0.0;
      for (int i = 0; i < 12; ++i) acc = acc + A[i][r] * A[i][cc];
      G[r * 12 + cc] = acc;
    }
  jacobi_n(G, V, 12, 16);
  int m = 0;
  for (int i = 1; i < 12; ++i) if (G[i * 12 + i] < G[m * 12 + m]) m = i;
  double P[12];
  for (int k = 0; k < 12; ++k) P[k] = V[k * 12 + m];
  /* undo the normalisation: P = P' [s I, -s c; 0 1] */
  double M[9], tt[3];
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) M[r * 3 + k] = P[r * 4 + k] * s;
    tt[r] = P[r * 4 + 3] - ((M[r * 3] * c[0] + M[r * 3 + 1] * c[1]) + M[r * 3 + 2] * c[2]);
  }
  double det = (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (det < 0.0) { for (int k = 0; k < 9; ++k) M[k] = -M[k]; for (int k = 0; k < 3; ++k) tt[k] = -tt[k]; det = -det; }
  if (!(det > 0.0)) return 0;
  /* nearest rotation: R = M (M^T M)^(-1/2) through the eigen-decomposition of M^T M; scale = mean singular value */
  double g[9], W[9];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) g[r * 3 + cc] = (M[r] * M[cc] + M[3 + r] * M[3 + cc]) + M[6 + r] * M[6 + cc];
  jacobi_n(g, W, 3, 12);
  double sv[3], scale = 0.0;
  for (int k = 0; k < 3; ++k) { sv[k] = sqrt(g[k * 3 + k] > 0.0 ? g[k * 3 + k] : 0.0); scale = scale + sv[k]; }
  scale = scale / 3.0;
  if (!(sv[0] > 0.0 && sv[1] > 0.0 && sv[2] > 0.0)) return 0;
  double S[9];   /* (M^T M)^(-1/2) = W diag(1/sv) W^T */
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc)
      S[r * 3 + cc] = (W[r * 3] * W[cc * 3] / sv[0] + W[r * 3 + 1] * W[cc * 3 + 1] / sv[1]) + W[r * 3 + 2] * W[cc * 3 + 2] / sv[2];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) R[r * 3 + cc] = (M[r * 3] * S[cc] + M[r * 3 + 1] * S[3 + cc]) + M[r * 3 + 2] * S[6 + cc];
  for (int k = 0; k < 3; ++k) t[k] = tt[k] / scale;
  return 1;
}

/* SolvePnPWithCV, src/g2o_optimization.cc:323-377.  obj n x 3, img n x 2 (cv::Point3f / Point2f).  pose: Twc 4x4
   row-major.  inliers: n flags.  Returns the inlier count (0: fewer than 8 points or no hypothesis). */
int opnp_solve_ransac(const opnp_config *cfg, const float *obj, const float *img, int n, double *pose, uint8_t *inliers) {
  for (int k = 0; k < 16; ++k) pose[k] = (k % 5 == 0) ? 1.0 : 0.0;
  for (int j = 0; j < n; ++j) inliers[j] = 0;
  if (n < 8) return 0;
  const cam4 cam = {cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  const int its = cfg->iterations > 0 ? cfg->iterations : 100;
  const double gate = (cfg->reprojection_error > 0 ? cfg->reprojection_error : 20.0);
  const double gate2 = gate * gate, conf = cfg->confidence > 0 ? cfg->confidence : 0.99;
  double *X = (double *)malloc(sizeof(double) * 3 * (size_t)n), *uv = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  double *xn = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < 3; ++k) X[3 * j + k] = (double)obj[3 * j + k];
    uv[2 * j] = (double)img[2 * j]; uv[2 * j + 1] = (double)img[2 * j + 1];
    xn[2 * j] = (uv[2 * j] - cam.cx) / cam.fx; xn[2 * j + 1] = (uv[2 * j + 1] - cam.cy) / cam.fy;
  }
  double *Rs = (double *)malloc(sizeof(double) * 12 * (size_t)its);
  int *cnt = (int *)malloc(sizeof(int) * (size_t)its);
#pragma omp parallel for schedule(static)
  for (int it = 0; it < its; ++it) {
    int set[6];
    draw6(cfg->seed, it, n, set);
    double X6[18], x6[12];
    for (int i = 0; i < 6; ++i) {
      memcpy(X6 + 3 * i, X + 3 * set[i], 24);
      memcpy(x6 + 2 * i, xn + 2 * set[i], 16);
    }
    double *R = Rs + 12 * (size_t)it, *t = R + 9;
    cnt[it] = -1;
    if (!dlt6(X6, x6, R, t)) continue;
    int c = 0;
    for (int j = 0; j < n; ++j) {
      double e[2];
      edge_eval(&cam, R, t, X + 3 * j, uv + 2 * j, e, NULL);
      const double z = ((R[6] * X[3 * j] + R[7] * X[3 * j + 1]) + R[8] * X[3 * j + 2]) + t[2];
      if (z > 0.0 && e[0] * e[0] + e[1] * e[1] <= gate2) ++c;
    }
    cnt[it] = c;
  }
  /* sequential RANSAC bookkeeping: first best count wins, each new best shrinks the number of hypotheses that still
     count to the smallest k with (1 - w^6)^k <= 1 - confidence */
  int best = -1, best_cnt = 0, niters = its;
  for (int it = 0; it < its && it < niters; ++it) {
    if (cnt[it] > best_cnt) {
      best_cnt = cnt[it]; best = it;
      const double wr = (double)cnt[it] / (double)n;
      double w6 = (wr * wr) * wr; w6 = w6 * w6;
      const double qf = 1.0 - w6, tgt = 1.0 - conf;
      int k = 1; double acc = qf;
      while (acc > tgt && k < its) { acc = acc * qf; ++k; }
      if (k < niters) niters = k;
    }
  }
  int ninl = 0;
  if (best >= 0 && best_cnt >= 6) {
    double *R = Rs + 12 * (size_t)best, *t = R + 9;
    for (int j = 0; j < n; ++j) {
      double e[2];
      edge_eval(&cam, R, t, X + 3 * j, uv + 2 * j, e, NULL);
      const double z = ((R[6] * X[3 * j] + R[7] * X[3 * j + 1]) + R[8] * X[3 * j + 2]) + t[2];
      inliers[j] = (z > 0.0 && e[0] * e[0] + e[1] * e[1] <= gate2) ? 1 : 0;
      ninl += inliers[j];
    }
    /* refinement on the inliers (OpenCV: SOLVEPNP_ITERATIVE on the inlier set) */
    double q[4], tt[3] = {t[0], t[1], t[2]};
    R_to_q(R, q);
    lm_pose(&cam, X, uv, n, inliers, 0.0, 10, q, tt);
    double Rr[9];
    q_to_R(q, Rr);
    /* Twc = [Rcw^T, -Rcw^T tcw] :363-367 */
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) pose[r * 4 + c] = Rr[c * 3 + r];
      pose[r * 4 + 3] = -((Rr[r] * tt[0] + Rr[3 + r] * tt[1]) + Rr[6 + r] * tt[2]);
    }
  }
  free(X); free(uv); free(xn); free(Rs); free(cnt);
  return ninl;
}
