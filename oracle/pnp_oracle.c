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

/* (H + lambda I) x = b, H symmetric positive definite 6x6.  Cholesky by columns (Crout): every entry of L is the same
   left-to-right inner product as in the row-by-row form, so the factor and the solution are the same numbers */
static int solve6(const double H[36], double lambda, const double b[6], double x[6]) {
  double L[6][6];
  for (int col = 0; col < 6; ++col) {
    double d = H[col * 6 + col] + lambda;
    for (int k = 0; k < col; ++k) d = d - L[col][k] * L[col][k];
    if (!(d > 0.0)) return 0;
    L[col][col] = sqrt(d);
    for (int row = col + 1; row < 6; ++row) {
      double v = H[row * 6 + col] + 0.0;
      for (int k = 0; k < col; ++k) v = v - L[row][k] * L[col][k];
      L[row][col] = v / L[col][col];
    }
  }
  double fwd[6];
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v = v - L[i][k] * fwd[k];
    fwd[i] = v / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double v = fwd[i];
    for (int k = i + 1; k < 6; ++k) v = v - L[k][i] * x[k];
    x[i] = v / L[i][i];
  }
  return 1;
}

typedef struct { double fx, fy, cx, cy; } cam4;

/* The edges of one frame, in the order FrameOptimization adds them (src/g2o_optimization.cc:213-258): n_mono mono edges
   (EdgeSE3ProjectXYZOnlyPose, measurement u v), then n - n_mono stereo edges (EdgeStereoSE3ProjectXYZOnlyPose, u v u_right). */
typedef struct {
  cam4 cam;
  double bf;
  const double *pts;       /* [n][3] */
  const double *meas;      /* [n][stride] */
  int n, n_mono, stride;
  double gate_mono, gate_stereo;
} edge_set;

typedef struct { int rows; double res[3]; double jac[3][6]; } edge_value;

/* residual = measurement - projection of (R X + t), and (want_jac) its derivative with respect to the left increment
   (omega, upsilon): linearizeOplus of the two g2o edges.  The stereo edge's third row is the column in the right image,
   u - bf / z. */
static void evaluate_edge(const edge_set *E, int j, const double R[9], const double t[3], int want_jac, edge_value *out) {
  const double *P = E->pts + 3 * (size_t)j, *m = E->meas + (size_t)E->stride * j;
  const double xc = ((R[0] * P[0] + R[1] * P[1]) + R[2] * P[2]) + t[0];
  const double yc = ((R[3] * P[0] + R[4] * P[1]) + R[5] * P[2]) + t[1];
  const double zc = ((R[6] * P[0] + R[7] * P[1]) + R[8] * P[2]) + t[2];
  const double inv = 1.0 / zc;
  const int stereo = j >= E->n_mono;
  const double fx = E->cam.fx, fy = E->cam.fy;
  const double u_left = xc * inv * fx + E->cam.cx;
  out->rows = stereo ? 3 : 2;
  out->res[0] = m[0] - u_left;
  out->res[1] = m[1] - (yc * inv * fy + E->cam.cy);
  out->res[2] = stereo ? m[2] - (u_left - E->bf * inv) : 0.0;
  if (!want_jac) return;
  const double inv2 = inv * inv;
  double *du = out->jac[0], *dv = out->jac[1], *dr = out->jac[2];
  du[0] = xc * yc * inv2 * fx;
  du[1] = -(1.0 + xc * xc * inv2) * fx;
  du[2] = yc * inv * fx;
  du[3] = -inv * fx;
  du[4] = 0.0;
  du[5] = xc * inv2 * fx;
  dv[0] = (1.0 + yc * yc * inv2) * fy;
  dv[1] = -xc * yc * inv2 * fy;
  dv[2] = -xc * inv * fy;
  dv[3] = 0.0;
  dv[4] = -inv * fy;
  dv[5] = yc * inv2 * fy;
  if (stereo) {
    dr[0] = du[0] - E->bf * yc * inv2;
    dr[1] = du[1] + E->bf * xc * inv2;
    dr[2] = du[2];
    dr[3] = du[3];
    dr[4] = 0.0;
    dr[5] = du[5] - E->bf * inv2;
  }
}

static double squared_norm(const edge_value *v) {
  const double s = v->res[0] * v->res[0] + v->res[1] * v->res[1];
  return v->rows == 3 ? s + v->res[2] * v->res[2] : s;
}

/* robust cost over the active edges; with H != NULL also the Gauss-Newton system H = sum w J^T J, g = -sum w J^T r.
   Lane l of the canonical wave order takes the edges l, l + 64, ...; the 28 sums are closed by the butterfly. */
static double accumulate(const edge_set *E, const double q[4], const double t[3], const uint8_t *active, int robust,
                         double *H, double *g) {
  double R[9];
  q_to_R(q, R);
  const double huber_mono = robust ? sqrt(E->gate_mono) : 0.0, huber_stereo = robust ? sqrt(E->gate_stereo) : 0.0;
  double lanes[28][64];
  for (int lane = 0; lane < 64; ++lane) {
    double sum[28];
    for (int k = 0; k < 28; ++k) sum[k] = 0.0;
    for (int j = lane; j < E->n; j += 64) {
      if (!active[j]) continue;
      edge_value v;
      evaluate_edge(E, j, R, t, H != NULL, &v);
      const double r2 = squared_norm(&v);
      const double huber = v.rows == 3 ? huber_stereo : huber_mono;
      double cost = r2, weight = 1.0;
      if (huber > 0.0) {
        const double r = sqrt(r2);
        if (r > huber) { cost = 2.0 * r * huber - huber * huber; weight = huber / r; }
      }
      sum[27] = sum[27] + cost;
      if (H == NULL) continue;
      int k = 0;
      for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c, ++k) {
          double jj = v.jac[0][a] * v.jac[0][c] + v.jac[1][a] * v.jac[1][c];
          if (v.rows == 3) jj = jj + v.jac[2][a] * v.jac[2][c];
          sum[k] = sum[k] + weight * jj;
        }
      for (int a = 0; a < 6; ++a) {
        double jr = v.jac[0][a] * v.res[0] + v.jac[1][a] * v.res[1];
        if (v.rows == 3) jr = jr + v.jac[2][a] * v.res[2];
        sum[21 + a] = sum[21 + a] - weight * jr;
      }
    }
    for (int k = 0; k < 28; ++k) lanes[k][lane] = sum[k];
  }
  if (H != NULL) {
    int k = 0;
    for (int a = 0; a < 6; ++a)
      for (int c = a; c < 6; ++c, ++k) H[a * 6 + c] = H[c * 6 + a] = d_bfly64(lanes[k]);
    for (int a = 0; a < 6; ++a) g[a] = d_bfly64(lanes[21 + a]);
  }
  return d_bfly64(lanes[27]);
}

/* `iterations` Levenberg-Marquardt iterations on the camera-from-world pose (q, t) with the damping policy of g2o's
   OptimizationAlgorithmLevenberg: lambda0 = 1e-5 max diag H; a step is kept when the gain ratio is positive, and then
   lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)); otherwise lambda *= nu, nu *= 2 and the step is retried, ten times at most */
static void lm_pose(const edge_set *E, const uint8_t *active, int robust, int iterations, double q[4], double t[3]) {
  double damping = 0.0, growth = 2.0;
  for (int it = 0; it < iterations; ++it) {
    double H[36], g[6];
    double cost = accumulate(E, q, t, active, robust, H, g);
    if (it == 0) {
      double top = 0.0;
      for (int k = 0; k < 6; ++k) top = fmax(top, fabs(H[k * 7]));
      damping = 1e-5 * top;
      growth = 2.0;
    }
    double ratio = 0.0;
    int tries = 0;
    for (;;) {
      double keep_q[4], keep_t[3], step[6];
      memcpy(keep_q, q, sizeof(keep_q));
      memcpy(keep_t, t, sizeof(keep_t));
      const int solved = solve6(H, damping, g, step);
      double cost_new = 1.7976931348623157e308, predicted = 1e-3;
      if (solved) {
        se3_apply_update(step, q, t);
        cost_new = accumulate(E, q, t, active, robust, NULL, NULL);
        for (int k = 0; k < 6; ++k) predicted = predicted + step[k] * (damping * step[k] + g[k]);
      }
      ratio = (cost - cost_new) / predicted;
      if (solved && ratio > 0.0 && isfinite(cost_new)) {
        const double d = 2.0 * ratio - 1.0;
        double shrink = 1.0 - (d * d) * d;
        if (shrink > 2.0 / 3.0) shrink = 2.0 / 3.0;
        damping = damping * fmax(1.0 / 3.0, shrink);
        growth = 2.0;
        cost = cost_new;
      } else {
        damping = damping * growth;
        growth = growth * 2.0;
        memcpy(q, keep_q, sizeof(keep_q));
        memcpy(t, keep_t, sizeof(keep_t));
        if (!isfinite(damping)) break;
      }
      ++tries;
      if (!(ratio < 0.0 && tries < 10)) break;
    }
    if (tries == 10 || ratio == 0.0 || !isfinite(damping)) break;
  }
}

/* Twc (q, p) <-> Tcw (q, t) */
static void invert_pose(const double q[4], const double p[3], double qi[4], double ti[3]) {
  qi[0] = q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = -q[3];
  double R[9];
  q_to_R(qi, R);
  for (int i = 0; i < 3; ++i) ti[i] = -((R[i * 3] * p[0] + R[i * 3 + 1] * p[1]) + R[i * 3 + 2] * p[2]);
}

/* FrameOptimization, src/g2o_optimization.cc:179-321, over an edge set.  q_wc (w,x,y,z), p_wc: in = prior, out = optimised.
   inlier[n]: out = the re-classified *PointConstraint::inlier flags.  Returns n - outliers. */
static int optimise_frame(const edge_set *E, double *q_wc, double *p_wc, uint8_t *inlier) {
  const int n = E->n;
  double q_prior[4], t_prior[3], q[4], t[3];
  double q_in[4] = {q_wc[0], q_wc[1], q_wc[2], q_wc[3]};
  q_normalize(q_in);
  invert_pose(q_in, p_wc, q_prior, t_prior);
  memcpy(q, q_prior, sizeof(q)); memcpy(t, t_prior, sizeof(t));
  uint8_t *on_level0 = (uint8_t *)malloc(n > 0 ? n : 1);    /* edges on optimisation level 0 (all of them at first) */
  memset(on_level0, 1, n > 0 ? n : 1);
  int outliers = 0;
  for (int round = 0; round < 4; ++round) {
    memcpy(q, q_prior, sizeof(q)); memcpy(t, t_prior, sizeof(t));       /* :266-267 every round restarts from the prior */
    lm_pose(E, on_level0, round < 3, 10, q, t);                           /* :289-290, :304-305 */
    double R[9];
    q_to_R(q, R);
    outliers = 0;
    for (int j = 0; j < n; ++j) {
      edge_value v;
      evaluate_edge(E, j, R, t, 0, &v);
      const float chi2 = (float)squared_norm(&v);                         /* :279, :295 const float chi2 */
      const double gate = v.rows == 3 ? E->gate_stereo : E->gate_mono;
      const int bad = chi2 > gate;
      inlier[j] = !bad; on_level0[j] = !bad;
      outliers += bad;
    }
    if (n < 10) break;                                                    /* :309-310 */
  }
  free(on_level0);
  double q_out[4], p_out[3];
  invert_pose(q, t, q_out, p_out);
  q_normalize(q_out);
  memcpy(q_wc, q_out, sizeof(q_out)); memcpy(p_wc, p_out, sizeof(p_out));
  return n - outliers;
}

int oframe_optimization(const oposeopt_config *cfg, const double *Xw, const double *obs, int n, double *q_wc, double *p_wc,
                        uint8_t *inlier) {
  edge_set E = {{cfg->fx, cfg->fy, cfg->cx, cfg->cy}, 0.0, Xw, obs, n, n, 2, cfg->chi2_threshold, 0.0};
  return optimise_frame(&E, q_wc, p_wc, inlier);
}

/* the same with stereo edges: n_mono rows (u, v, unused) followed by n_stereo rows (u, v, u_right) */
int oframe_optimization_stereo(const oposeopt_stereo_config *cfg, const double *Xw, const double *obs, int n_mono, int n_stereo,
                               double *q_wc, double *p_wc, uint8_t *inlier) {
  edge_set E = {{cfg->fx, cfg->fy, cfg->cx, cfg->cy}, cfg->bf, Xw, obs, n_mono + n_stereo, n_mono, 3, cfg->chi2_mono,
                cfg->chi2_stereo};
  return optimise_frame(&E, q_wc, p_wc, inlier);
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
  const edge_set all = {cam, 0.0, X, uv, n, n, 2, 0.0, 0.0};      /* every correspondence as a mono edge */
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
      edge_value v;
      evaluate_edge(&all, j, R, t, 0, &v);
      const double z = ((R[6] * X[3 * j] + R[7] * X[3 * j + 1]) + R[8] * X[3 * j + 2]) + t[2];
      if (z > 0.0 && squared_norm(&v) <= gate2) ++c;
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
      edge_value v;
      evaluate_edge(&all, j, R, t, 0, &v);
      const double z = ((R[6] * X[3 * j] + R[7] * X[3 * j + 1]) + R[8] * X[3 * j + 2]) + t[2];
      inliers[j] = (z > 0.0 && squared_norm(&v) <= gate2) ? 1 : 0;
      ninl += inliers[j];
    }
    /* refinement on the inliers (OpenCV: SOLVEPNP_ITERATIVE on the inlier set) */
    double q[4], tt[3] = {t[0], t[1], t[2]};
    R_to_q(R, q);
    lm_pose(&all, inliers, 0, 10, q, tt);
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
