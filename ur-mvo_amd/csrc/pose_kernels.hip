// pose_kernels.hip -- the per-frame pose stage that follows the front-end, batched
// over the frames of a step:
//   urf_solve_pnp_ransac    SolvePnPWithCV      src/g2o_optimization.cc:323-377
//   urf_frame_optimization  FrameOptimization   src/g2o_optimization.cc:179-321
// cv::solvePnPRansac (OpenCV 4.2) and g2o are un-vendored: the arithmetic is the written specification of
// DESIGN.md "Pose stage" (the test suite holds a CPU restatement of it; parity unpinned), with every parameter of the
// reference's call sites kept: <= 100 hypotheses, 20 px gate, confidence 0.99, >= 8 points; Huber kernel
// sqrt(5.991), 4 rounds x 10 Levenberg-Marquardt iterations restarted from the prior, chi-square re-classification.
//
// GPU mapping.  RANSAC: one THREAD per hypothesis solves a 6-point DLT -- the 12x12 Gram matrix and its Jacobi
// rotations live in LDS, element (i,j) of thread l at [(i*12+j)*32 + l], so data-dependent indices cost neither scratch
// nor bank conflicts; one WAVE per hypothesis counts its inliers; one wave per FRAME does the sequential RANSAC
// bookkeeping and the refinement.  Levenberg-Marquardt: one wave per frame, lane l owns observations l, l+64, ...; the 28
// sums of an iteration (21 entries of H, 6 of b, chi2) are lane-sequential sums closed by the 64-lane butterfly -- the
// canonical order, so CPU and GPU agree bit for bit; the 6x6 Cholesky solve and the SE(3) update run redundantly in
// every lane.  Everything is f64; sin/cos are polynomial kernels (no device libm).
#include "../../include/urf.h"
#include "urf_common.h"

#include <math.h>

namespace urf {
namespace pose {

struct Cam { double fx, fy, cx, cy; };
struct Rigid { double q[4]; double t[3]; };     // camera-from-world: unit quaternion (w, x, y, z) and translation

__device__ __forceinline__ double wave_total(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}

// sin / cos for |x| < ~1e5: Cody-Waite reduction by pi/2 (three parts), fdlibm kernel polynomials, Horner, no fma
__device__ void sincos_poly(double x, double &sn, double &cs) {
  const double n = rint(x * 6.36619772367581382433e-01);
  const double r = ((x - n * 1.57079632673412561417e+00) - n * 6.07710050650619224932e-11) - n * 2.02226624879595063154e-21;
  const double z = r * r;
  const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  const double sr = r + r * (z * ps);
  const double cr = (1.0 - 0.5 * z) + z * (z * pc);
  switch ((long)n & 3) {
    case 0: sn = sr; cs = cr; break;
    case 1: sn = cr; cs = -sr; break;
    case 2: sn = -sr; cs = -cr; break;
    default: sn = -cr; cs = sr; break;
  }
}

__device__ void unit(double (&q)[4]) {
  const double len = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = q[k] / len;
  if (q[0] < 0.0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = -q[k];
  }
}
__device__ void rotation_of(const double (&q)[4], double (&R)[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}
__device__ void quaternion_of(const double (&R)[9], double (&q)[4]) {   // Shepperd: largest component first
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
  unit(q);
}

// T <- exp(dx) T with dx = (omega, upsilon): VertexSE3Expmap::oplusImpl / SE3Quat::exp of g2o
__device__ void left_update(const double (&dx)[6], Rigid &T) {
  const double wx = dx[0], wy = dx[1], wz = dx[2];
  const double th2 = (wx * wx + wy * wy) + wz * wz, th = sqrt(th2);
  double a, b, c, half_cos, half_sin_over;     // sin(th)/th, (1-cos th)/th^2, (th-sin th)/th^3, cos(th/2), sin(th/2)/th
  if (th < 1e-5) {
    a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0;
    half_cos = 1.0 - th2 / 8.0; half_sin_over = 0.5 - th2 / 48.0;
  } else {
    double s, co, sh, ch;
    sincos_poly(th, s, co);
    sincos_poly(0.5 * th, sh, ch);
    a = s / th; b = (1.0 - co) / th2; c = (th - s) / (th2 * th);
    half_cos = ch; half_sin_over = sh / th;
  }
  const double W[9] = {0.0, -wz, wy, wz, 0.0, -wx, -wy, wx, 0.0};
  double W2[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) W2[i * 3 + j] = (W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j]) + W[i * 3 + 2] * W[6 + j];
  double tn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double rot = 0.0, vu = 0.0;
    double Rrow[3], Vrow[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double id = (i == j) ? 1.0 : 0.0;
      Rrow[j] = (id + a * W[i * 3 + j]) + b * W2[i * 3 + j];
      Vrow[j] = (id + b * W[i * 3 + j]) + c * W2[i * 3 + j];
    }
    rot = (Rrow[0] * T.t[0] + Rrow[1] * T.t[1]) + Rrow[2] * T.t[2];
    vu = (Vrow[0] * dx[3] + Vrow[1] * dx[4]) + Vrow[2] * dx[5];
    tn[i] = rot + vu;
  }
  const double d[4] = {half_cos, half_sin_over * wx, half_sin_over * wy, half_sin_over * wz};
  const double (&p)[4] = T.q;
  double qn[4];
  qn[0] = ((d[0] * p[0] - d[1] * p[1]) - d[2] * p[2]) - d[3] * p[3];
  qn[1] = ((d[0] * p[1] + d[1] * p[0]) + d[2] * p[3]) - d[3] * p[2];
  qn[2] = ((d[0] * p[2] - d[1] * p[3]) + d[2] * p[0]) + d[3] * p[1];
  qn[3] = ((d[0] * p[3] + d[1] * p[2]) - d[2] * p[1]) + d[3] * p[0];
  unit(qn);
#pragma unroll
  for (int k = 0; k < 4; ++k) T.q[k] = qn[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) T.t[k] = tn[k];
}

// (H + lambda I) x = b, H symmetric positive definite 6x6: Cholesky row by row
__device__ bool damped_solve(const double (&H)[36], double lambda, const double (&b)[6], double (&x)[6]) {
  double L[36];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = H[i * 6 + j] + (i == j ? lambda : 0.0);
#pragma unroll
      for (int k = 0; k < j; ++k) s = s - L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i * 6 + i] = sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = s - L[i * 6 + k] * y[k];
    y[i] = s / L[i * 6 + i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s = s - L[k * 6 + i] * x[k];
    x[i] = s / L[i * 6 + i];
  }
  return true;
}

// The observations of one frame: mono edges first (2 numbers each: u, v), then stereo edges (3 numbers: u, v, u_right) in
// the same arrays with row stride `os` -- the order FrameOptimization adds them to the graph (src/g2o_optimization.cc:213-258).
struct Obs {
  const double *X, *o;     // map points [n][3], measurements [n][os]
  int n, n_mono, os;
  double bf;               // Camera::BF() of the stereo edges
  double gate_m, gate_s;   // chi-square gates cfg.mono_point / cfg.stereo_point; the Huber deltas are their square roots
};

// EdgeStereoSE3ProjectXYZOnlyPose (src/g2o_optimization.cc:235-260): the mono edge plus a third row, the column of the point in
// the right image u - bf / z; measurement (u, v, u_right)
template <bool JAC>
__device__ __forceinline__ void reprojection_stereo(const Cam &cam, double bf, const double (&R)[9], const double (&t)[3],
                                                    const double *X, const double *obs, double (&e)[3], double (&J)[18]) {
  const double x = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
  const double y = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
  const double z = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
  const double iz = 1.0 / z;
  const double u = x * iz * cam.fx + cam.cx;
  e[0] = obs[0] - u;
  e[1] = obs[1] - (y * iz * cam.fy + cam.cy);
  e[2] = obs[2] - (u - bf * iz);
  if (JAC) {
    const double iz2 = iz * iz;
    J[0] = x * y * iz2 * cam.fx;           J[1] = -(1.0 + x * x * iz2) * cam.fx; J[2] = y * iz * cam.fx;
    J[3] = -iz * cam.fx;                   J[4] = 0.0;                           J[5] = x * iz2 * cam.fx;
    J[6] = (1.0 + y * y * iz2) * cam.fy;   J[7] = -x * y * iz2 * cam.fy;         J[8] = -x * iz * cam.fy;
    J[9] = 0.0;                            J[10] = -iz * cam.fy;                 J[11] = y * iz2 * cam.fy;
    J[12] = J[0] - bf * y * iz2;           J[13] = J[1] + bf * x * iz2;          J[14] = J[2];
    J[15] = J[3];                          J[16] = 0.0;                          J[17] = J[5] - bf * iz2;
  }
}

// EdgeSE3ProjectXYZOnlyPose: error = obs - project(R X + t); JAC: its Jacobian w.r.t. (omega, upsilon)
template <bool JAC>
__device__ __forceinline__ void reprojection(const Cam &cam, const double (&R)[9], const double (&t)[3], const double *X,
                                             const double *obs, double (&e)[2], double (&J)[12], double &depth) {
  const double x = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
  const double y = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
  const double z = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
  const double iz = 1.0 / z;
  depth = z;
  e[0] = obs[0] - (x * iz * cam.fx + cam.cx);
  e[1] = obs[1] - (y * iz * cam.fy + cam.cy);
  if (JAC) {
    const double iz2 = iz * iz;
    J[0] = x * y * iz2 * cam.fx;           J[1] = -(1.0 + x * x * iz2) * cam.fx; J[2] = y * iz * cam.fx;
    J[3] = -iz * cam.fx;                   J[4] = 0.0;                           J[5] = x * iz2 * cam.fx;
    J[6] = (1.0 + y * y * iz2) * cam.fy;   J[7] = -x * y * iz2 * cam.fy;         J[8] = -x * iz * cam.fy;
    J[9] = 0.0;                            J[10] = -iz * cam.fy;                 J[11] = y * iz2 * cam.fy;
  }
}

// robust chi2 over the active observations of this frame (whole wave); SYSTEM: also H = sum w J^T J, b = -sum w J^T e.
// robust: the Huber kernel with delta = sqrt(gate) of the observation's kind (rounds 0-2), else none.
template <bool SYSTEM>
__device__ double normal_equations(const Cam &cam, const Rigid &T, const Obs &ob, const uint8_t *active, bool robust,
                                   double (&H)[36], double (&b)[6], int lane) {
  double R[9];
  rotation_of(T.q, R);
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  const double delta_m = robust ? sqrt(ob.gate_m) : 0.0, delta_s = robust ? sqrt(ob.gate_s) : 0.0;
  for (int j = lane; j < ob.n; j += 64) {
    if (!active[j]) continue;
    if (j < ob.n_mono) {
      double e[2], J[12], depth;
      reprojection<SYSTEM>(cam, R, T.t, ob.X + 3 * j, ob.o + (size_t)ob.os * j, e, J, depth);
      const double e2 = e[0] * e[0] + e[1] * e[1];
      double rho = e2, w = 1.0;
      if (delta_m > 0.0) {
        const double en = sqrt(e2);
        if (en > delta_m) { rho = 2.0 * en * delta_m - delta_m * delta_m; w = delta_m / en; }
      }
      acc[27] = acc[27] + rho;
      if (SYSTEM) {
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) { acc[k] = acc[k] + w * (J[r] * J[c] + J[6 + r] * J[6 + c]); ++k; }
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] = acc[21 + r] - w * (J[r] * e[0] + J[6 + r] * e[1]);
      }
    } else {
      double e[3], J[18];
      reprojection_stereo<SYSTEM>(cam, ob.bf, R, T.t, ob.X + 3 * j, ob.o + (size_t)ob.os * j, e, J);
      const double e2 = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
      double rho = e2, w = 1.0;
      if (delta_s > 0.0) {
        const double en = sqrt(e2);
        if (en > delta_s) { rho = 2.0 * en * delta_s - delta_s * delta_s; w = delta_s / en; }
      }
      acc[27] = acc[27] + rho;
      if (SYSTEM) {
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int c = r; c < 6; ++c) { acc[k] = acc[k] + w * ((J[r] * J[c] + J[6 + r] * J[6 + c]) + J[12 + r] * J[12 + c]); ++k; }
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] = acc[21 + r] - w * ((J[r] * e[0] + J[6 + r] * e[1]) + J[12 + r] * e[2]);
      }
    }
  }
  if (SYSTEM) {
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c) { const double v = wave_total(acc[k]); H[r * 6 + c] = v; H[c * 6 + r] = v; ++k; }
#pragma unroll
    for (int r = 0; r < 6; ++r) b[r] = wave_total(acc[21 + r]);
  }
  return wave_total(acc[27]);
}

// `iterations` Levenberg-Marquardt iterations with the damping policy of g2o's OptimizationAlgorithmLevenberg
__device__ void levenberg(const Cam &cam, const Obs &ob, const uint8_t *active, bool robust, int iterations, Rigid &T,
                          int lane) {
  double lambda = 0.0, nu = 2.0;
  for (int it = 0; it < iterations; ++it) {
    double H[36], b[6];
    double current = normal_equations<true>(cam, T, ob, active, robust, H, b, lane);
    if (it == 0) {
      double md = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) md = fmax(md, fabs(H[k * 6 + k]));
      lambda = 1e-5 * md;
      nu = 2.0;
    }
    double gain = 0.0;
    int trials = 0;
    do {
      const Rigid backup = T;
      double dx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const bool ok = damped_solve(H, lambda, b, dx);
      double trial = 1.7976931348623157e308;
      double Hd[36], bd[6];
      if (ok) {
        left_update(dx, T);
        trial = normal_equations<false>(cam, T, ob, active, robust, Hd, bd, lane);
      }
      double scale = 1e-3;
      if (ok) {
#pragma unroll
        for (int k = 0; k < 6; ++k) scale = scale + dx[k] * (lambda * dx[k] + b[k]);
      }
      gain = (current - trial) / scale;
      if (ok && gain > 0.0 && isfinite(trial)) {
        const double g1 = 2.0 * gain - 1.0;
        double alpha = 1.0 - (g1 * g1) * g1;
        if (alpha > 2.0 / 3.0) alpha = 2.0 / 3.0;
        lambda = lambda * fmax(1.0 / 3.0, alpha);
        nu = 2.0;
        current = trial;
      } else {
        lambda = lambda * nu;
        nu = nu * 2.0;
        T = backup;
        if (!isfinite(lambda)) break;
      }
      ++trials;
    } while (gain < 0.0 && trials < 10);
    if (trials == 10 || gain == 0.0 || !isfinite(lambda)) break;
  }
}

__device__ void inverse_of(const double (&q)[4], const double (&p)[3], double (&qi)[4], double (&ti)[3]) {
  qi[0] = q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = -q[3];
  double R[9];
  rotation_of(qi, R);
#pragma unroll
  for (int i = 0; i < 3; ++i) ti[i] = -((R[i * 3] * p[0] + R[i * 3 + 1] * p[1]) + R[i * 3 + 2] * p[2]);
}

// ------------------------------------------------------------------ FrameOptimization: one wave per frame
// counts_s == nullptr: mono edges only (os = 2).  Else frame f holds counts[f] mono edges followed by counts_s[f] stereo
// edges, rows of os = 3 numbers.
__global__ void __launch_bounds__(64) frame_optimization_kernel(Cam cam, double bf, double gate_m, double gate_s, const int *counts,
                                                                const int *counts_s, const double *Xw, const double *obs,
                                                                int cap, double *q_wc, double *p_wc, uint8_t *inlier,
                                                                uint8_t *level0, int *n_inliers) {
  const int f = blockIdx.x, lane = threadIdx.x;
  Obs ob;
  ob.n_mono = counts[f];
  ob.n = ob.n_mono + (counts_s ? counts_s[f] : 0);
  ob.os = counts_s ? 3 : 2;
  ob.X = Xw + (size_t)f * cap * 3;
  ob.o = obs + (size_t)f * cap * ob.os;
  ob.bf = bf; ob.gate_m = gate_m; ob.gate_s = gate_s;
  const int n = ob.n;
  uint8_t *inl = inlier + (size_t)f * cap, *lvl = level0 + (size_t)f * cap;
  double qn[4] = {q_wc[4 * f], q_wc[4 * f + 1], q_wc[4 * f + 2], q_wc[4 * f + 3]};
  const double pn[3] = {p_wc[3 * f], p_wc[3 * f + 1], p_wc[3 * f + 2]};
  unit(qn);
  Rigid prior, T;
  inverse_of(qn, pn, prior.q, prior.t);
  T = prior;
  for (int j = lane; j < n; j += 64) lvl[j] = 1;
  int outliers = 0;
  for (int round = 0; round < 4; ++round) {
    T = prior;                                                  // :266-267 every round restarts from the prior
    levenberg(cam, ob, lvl, round < 3, 10, T, lane);            // :289-290, :304-305 no robust kernel after round 2
    double R[9];
    rotation_of(T.q, R);
    int bad = 0;
    for (int j = lane; j < n; j += 64) {
      bool out;
      if (j < ob.n_mono) {
        double e[2], J[12], depth;
        reprojection<false>(cam, R, T.t, ob.X + 3 * j, ob.o + (size_t)ob.os * j, e, J, depth);
        const float chi2 = (float)(e[0] * e[0] + e[1] * e[1]);  // :279 const float chi2
        out = chi2 > gate_m;
      } else {
        double e[3], J[18];
        reprojection_stereo<false>(cam, bf, R, T.t, ob.X + 3 * j, ob.o + (size_t)ob.os * j, e, J);
        const float chi2 = (float)((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);   // :295
        out = chi2 > gate_s;
      }
      inl[j] = out ? 0 : 1;
      lvl[j] = out ? 0 : 1;
      bad += out ? 1 : 0;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) bad += __shfl_xor(bad, s, 64);
    outliers = bad;
    if (n < 10) break;
  }
  double qo[4], po[3];
  inverse_of(T.q, T.t, qo, po);
  unit(qo);
  if (lane == 0) {
    for (int k = 0; k < 4; ++k) q_wc[4 * f + k] = qo[k];
    for (int k = 0; k < 3; ++k) p_wc[3 * f + k] = po[k];
    n_inliers[f] = n - outliers;
  }
}

// ------------------------------------------------------------------ PnP-RANSAC
__device__ __forceinline__ uint32_t counter_hash(uint32_t seed, uint32_t ctr) {
  uint32_t x = seed ^ (ctr * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ void draw_six(uint32_t seed, int it, int n, int (&set)[6]) {     // swap-with-back draw, sparse bookkeeping
  int mpos[6], mval[6], nm = 0;
  for (int j = 0; j < 6; ++j) {
    const int size = n - j;
    const uint32_t r = counter_hash(seed, (uint32_t)(it * 8 + j)) >> 1;
    const int pick = (int)(((double)r / 2147483648.0) * size);
    int idx = pick, back = size - 1;
    for (int k = 0; k < nm; ++k) if (mpos[k] == pick) idx = mval[k];
    for (int k = 0; k < nm; ++k) if (mpos[k] == size - 1) back = mval[k];
    set[j] = idx;
    bool found = false;
    for (int k = 0; k < nm; ++k) if (mpos[k] == pick) { mval[k] = back; found = true; }
    if (!found) { mpos[nm] = pick; mval[nm] = back; ++nm; }
  }
}

// cyclic Jacobi on a symmetric N x N matrix held in LDS with stride S between elements (one matrix per thread)
template <int N, int S>
__device__ void jacobi_strided(double *a, double *v, int sweeps) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[(i * N + j) * S] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < sweeps; ++sweep)
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double off = a[(p * N + q) * S];
        if (fabs(off) < 1e-300) continue;
        const double theta = (a[(q * N + q) * S] - a[(p * N + p) * S]) / (2.0 * off);
        const double tn = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tn * tn + 1.0), s = tn * c;
        for (int k = 0; k < N; ++k) {
          const double x = a[(k * N + p) * S], y = a[(k * N + q) * S];
          a[(k * N + p) * S] = c * x - s * y; a[(k * N + q) * S] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const double x = a[(p * N + k) * S], y = a[(q * N + k) * S];
          a[(p * N + k) * S] = c * x - s * y; a[(q * N + k) * S] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const double x = v[(k * N + p) * S], y = v[(k * N + q) * S];
          v[(k * N + p) * S] = c * x - s * y; v[(k * N + q) * S] = s * x + c * y;
        }
      }
}

constexpr int HT = 32;      // hypotheses (threads) per workgroup of the DLT kernel

// one thread = one 6-point DLT hypothesis -> hyp[f][it] = (R row-major, t), ok[f][it]
__global__ void __launch_bounds__(HT) pnp_hypothesis_kernel(Cam cam, uint32_t seed, int iterations, const int *counts,
                                                            const float *obj, const float *img, int cap, double *hyp,
                                                            int *ok) {
  extern __shared__ double lds[];
  double *G = lds + threadIdx.x, *V = lds + 144 * HT + threadIdx.x;      // 12x12 each, element stride HT
  double *g3 = lds + 288 * HT + threadIdx.x, *w3 = lds + 297 * HT + threadIdx.x;   // 3x3 each
  const int f = blockIdx.y, it = blockIdx.x * HT + threadIdx.x;
  const int n = counts[f];
  if (it >= iterations) return;
  int &flag = ok[(size_t)f * iterations + it];
  flag = 0;
  if (n < 8) return;
  int set[6];
  draw_six(seed, it, n, set);
  const float *ob = obj + (size_t)f * cap * 3, *im = img + (size_t)f * cap * 2;
  double X6[18], x6[12], centre[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 6; ++i) {
    for (int k = 0; k < 3; ++k) { X6[3 * i + k] = (double)ob[3 * set[i] + k]; centre[k] = centre[k] + X6[3 * i + k]; }
    x6[2 * i] = ((double)im[2 * set[i]] - cam.cx) / cam.fx;
    x6[2 * i + 1] = ((double)im[2 * set[i] + 1] - cam.cy) / cam.fy;
  }
  for (int k = 0; k < 3; ++k) centre[k] = centre[k] / 6.0;
  double spread = 0.0;
  for (int i = 0; i < 6; ++i) {
    const double dx = X6[3 * i] - centre[0], dy = X6[3 * i + 1] - centre[1], dz = X6[3 * i + 2] - centre[2];
    spread = spread + sqrt((dx * dx + dy * dy) + dz * dz);
  }
  spread = spread / 6.0;
  if (!(spread > 0.0)) return;
  const double s = 1.0 / spread;
  // Gram matrix of the 12 x 12 DLT system, accumulated row by row (i ascending), in LDS
  for (int k = 0; k < 144; ++k) G[k * HT] = 0.0;
  for (int i = 0; i < 12; ++i) {
    const int pt = i >> 1;
    const double X = (X6[3 * pt] - centre[0]) * s, Y = (X6[3 * pt + 1] - centre[1]) * s, Z = (X6[3 * pt + 2] - centre[2]) * s;
    const double m = (i & 1) ? x6[2 * pt + 1] : x6[2 * pt];
    double row[12];
    for (int k = 0; k < 12; ++k) row[k] = 0.0;
    const int o = (i & 1) ? 4 : 0;
    row[o] = X; row[o + 1] = Y; row[o + 2] = Z; row[o + 3] = 1.0;
    row[8] = -m * X; row[9] = -m * Y; row[10] = -m * Z; row[11] = -m;
    for (int r = 0; r < 12; ++r)
      for (int c = 0; c < 12; ++c) G[(r * 12 + c) * HT] = G[(r * 12 + c) * HT] + row[r] * row[c];
  }
  jacobi_strided<12, HT>(G, V, 16);
  int weakest = 0;
  for (int i = 1; i < 12; ++i)
    if (G[(i * 12 + i) * HT] < G[(weakest * 12 + weakest) * HT]) weakest = i;
  double P[12];
  for (int k = 0; k < 12; ++k) P[k] = V[(k * 12 + weakest) * HT];
  double M[9], tt[3];
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) M[r * 3 + k] = P[r * 4 + k] * s;
    tt[r] = P[r * 4 + 3] - ((M[r * 3] * centre[0] + M[r * 3 + 1] * centre[1]) + M[r * 3 + 2] * centre[2]);
  }
  double det = (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (det < 0.0) {
    for (int k = 0; k < 9; ++k) M[k] = -M[k];
    for (int k = 0; k < 3; ++k) tt[k] = -tt[k];
    det = -det;
  }
  if (!(det > 0.0)) return;
  // nearest rotation R = M (M^T M)^(-1/2); the scale of P is the mean singular value of M
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) g3[(r * 3 + c) * HT] = (M[r] * M[c] + M[3 + r] * M[3 + c]) + M[6 + r] * M[6 + c];
  jacobi_strided<3, HT>(g3, w3, 12);
  double sv[3], scale = 0.0;
  for (int k = 0; k < 3; ++k) {
    const double ev = g3[(k * 3 + k) * HT];
    sv[k] = sqrt(ev > 0.0 ? ev : 0.0);
    scale = scale + sv[k];
  }
  scale = scale / 3.0;
  if (!(sv[0] > 0.0 && sv[1] > 0.0 && sv[2] > 0.0)) return;
  double Wm[9], S[9];
  for (int k = 0; k < 9; ++k) Wm[k] = w3[k * HT];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      S[r * 3 + c] = (Wm[r * 3] * Wm[c * 3] / sv[0] + Wm[r * 3 + 1] * Wm[c * 3 + 1] / sv[1]) + Wm[r * 3 + 2] * Wm[c * 3 + 2] / sv[2];
  double *out = hyp + ((size_t)f * iterations + it) * 12;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[r * 3 + c] = (M[r * 3] * S[c] + M[r * 3 + 1] * S[3 + c]) + M[r * 3 + 2] * S[6 + c];
  for (int k = 0; k < 3; ++k) out[9 + k] = tt[k] / scale;
  flag = 1;
}

__device__ __forceinline__ bool within_gate(const Cam &cam, const double (&R)[9], const double (&t)[3], const float *X3,
                                            const float *uv, double gate2) {
  const double X[3] = {(double)X3[0], (double)X3[1], (double)X3[2]}, o[2] = {(double)uv[0], (double)uv[1]};
  double e[2], J[12], depth;
  reprojection<false>(cam, R, t, X, o, e, J, depth);
  return depth > 0.0 && e[0] * e[0] + e[1] * e[1] <= gate2;
}

// one wave = one hypothesis: its inlier count (-1 for a degenerate sample)
__global__ void __launch_bounds__(256) pnp_count_kernel(Cam cam, double gate2, int iterations, const int *counts,
                                                        const float *obj, const float *img, int cap, const double *hyp,
                                                        const int *ok, int *count) {
  const int f = blockIdx.y, it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= iterations) return;
  const int n = counts[f];
  const size_t h = (size_t)f * iterations + it;
  if (n < 8 || !ok[h]) { if (lane == 0) count[h] = -1; return; }
  double R[9], t[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = hyp[h * 12 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = hyp[h * 12 + 9 + k];
  const float *ob = obj + (size_t)f * cap * 3, *im = img + (size_t)f * cap * 2;
  int c = 0;
  for (int j = lane; j < n; j += 64) c += within_gate(cam, R, t, ob + 3 * j, im + 2 * j, gate2) ? 1 : 0;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
  if (lane == 0) count[h] = c;
}

// one wave = one frame: sequential RANSAC bookkeeping, inlier flags of the winner, refinement, Twc
__global__ void __launch_bounds__(64) pnp_finish_kernel(Cam cam, double gate2, double confidence, int iterations,
                                                        const int *counts, const float *obj, const float *img, int cap,
                                                        const double *hyp, const int *count, double *Xd, double *uvd,
                                                        double *pose, uint8_t *inliers, int *n_inliers) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int n = counts[f];
  double *P = pose + (size_t)f * 16;
  uint8_t *inl = inliers + (size_t)f * cap;
  if (lane < 16) P[lane] = (lane % 5 == 0) ? 1.0 : 0.0;
  for (int j = lane; j < n; j += 64) inl[j] = 0;
  if (lane == 0) n_inliers[f] = 0;
  if (n < 8) return;
  const int *cnt = count + (size_t)f * iterations;
  // first best count wins; each new best shrinks the number of hypotheses that still count to the smallest k
  // with (1 - w^6)^k <= 1 - confidence (products accumulated sequentially in f64: the same on CPU and GPU)
  int best = -1, best_cnt = 0, horizon = iterations;
  for (int it = 0; it < iterations && it < horizon; ++it) {
    if (cnt[it] > best_cnt) {
      best_cnt = cnt[it]; best = it;
      const double wr = (double)cnt[it] / (double)n;
      double w6 = (wr * wr) * wr; w6 = w6 * w6;
      const double qf = 1.0 - w6, tgt = 1.0 - confidence;
      int k = 1;
      double acc = qf;
      while (acc > tgt && k < iterations) { acc = acc * qf; ++k; }
      if (k < horizon) horizon = k;
    }
  }
  if (best < 0 || best_cnt < 6) return;
  double R[9];
  Rigid T;
  const double *h = hyp + ((size_t)f * iterations + best) * 12;
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = h[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) T.t[k] = h[9 + k];
  const float *ob = obj + (size_t)f * cap * 3, *im = img + (size_t)f * cap * 2;
  double *X = Xd + (size_t)f * cap * 3, *O = uvd + (size_t)f * cap * 2;
  int c = 0;
  for (int j = lane; j < n; j += 64) {
    const bool in = within_gate(cam, R, T.t, ob + 3 * j, im + 2 * j, gate2);
    inl[j] = in ? 1 : 0;
    c += in ? 1 : 0;
    for (int k = 0; k < 3; ++k) X[3 * j + k] = (double)ob[3 * j + k];
    O[2 * j] = (double)im[2 * j]; O[2 * j + 1] = (double)im[2 * j + 1];
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the wave re-reads inl / X / O written by other lanes
  __builtin_amdgcn_s_waitcnt(0);
  quaternion_of(R, T.q);
  Obs ref;                                                    // refinement on the inliers (OpenCV: SOLVEPNP_ITERATIVE)
  ref.X = X; ref.o = O; ref.n = n; ref.n_mono = n; ref.os = 2; ref.bf = 0.0; ref.gate_m = 0.0; ref.gate_s = 0.0;
  levenberg(cam, ref, inl, false, 10, T, lane);
  double Rr[9];
  rotation_of(T.q, Rr);
  if (lane == 0) {
    for (int r = 0; r < 3; ++r) {                             // Twc = [Rcw^T, -Rcw^T tcw] :363-367
      for (int cc = 0; cc < 3; ++cc) P[r * 4 + cc] = Rr[cc * 3 + r];
      P[r * 4 + 3] = -((Rr[r] * T.t[0] + Rr[3 + r] * T.t[1]) + Rr[6 + r] * T.t[2]);
    }
    n_inliers[f] = c;
  }
}

}  // namespace pose
}  // namespace urf
using namespace urf;
using namespace urf::pose;

// ------------------------------------------------------------------ C ABI
struct urf_pose {
  int device = 0, maxB = 1, cap = 1024, max_iters = 0;
  hipStream_t st = nullptr;
  int *d_counts = nullptr, *d_ok = nullptr, *d_count = nullptr, *d_ninl = nullptr;
  float *d_obj = nullptr, *d_img = nullptr;
  double *d_X = nullptr, *d_uv = nullptr, *d_uv3 = nullptr, *d_hyp = nullptr, *d_pose = nullptr, *d_q = nullptr, *d_p = nullptr;
  uint8_t *d_inl = nullptr, *d_lvl = nullptr;
};

extern "C" int urf_pose_create(int device, int max_batch, int capacity, urf_pose **out) {
  URF_CHECK(out && max_batch >= 1 && max_batch <= 4096 && capacity >= 8 && capacity <= 65536, "urf_pose_create: bad argument");
  int ndev = 0;
  URF_HIP(hipGetDeviceCount(&ndev));
  URF_CHECK(ndev > 0, "no HIP device: liburf_front needs a gfx950 GPU (there is no CPU fallback)");
  URF_CHECK(device >= 0 && device < ndev, "device %d out of range (%d devices)", device, ndev);
  URF_HIP(hipSetDevice(device));
  urf_pose *h = new urf_pose();
  h->device = device; h->maxB = max_batch; h->cap = capacity; h->max_iters = 1024;
  const size_t B = (size_t)max_batch, C = (size_t)capacity;
  URF_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  URF_HIP(hipMalloc((void **)&h->d_counts, B * 4));
  URF_HIP(hipMalloc((void **)&h->d_ninl, B * 4));
  URF_HIP(hipMalloc((void **)&h->d_ok, B * h->max_iters * 4));
  URF_HIP(hipMalloc((void **)&h->d_count, B * h->max_iters * 4));
  URF_HIP(hipMalloc((void **)&h->d_obj, B * C * 3 * 4));
  URF_HIP(hipMalloc((void **)&h->d_img, B * C * 2 * 4));
  URF_HIP(hipMalloc((void **)&h->d_X, B * C * 3 * 8));
  URF_HIP(hipMalloc((void **)&h->d_uv, B * C * 2 * 8));
  URF_HIP(hipMalloc((void **)&h->d_uv3, B * C * 3 * 8));
  URF_HIP(hipMalloc((void **)&h->d_hyp, B * h->max_iters * 12 * 8));
  URF_HIP(hipMalloc((void **)&h->d_pose, B * 16 * 8));
  URF_HIP(hipMalloc((void **)&h->d_q, B * 4 * 8));
  URF_HIP(hipMalloc((void **)&h->d_p, B * 3 * 8));
  URF_HIP(hipMalloc((void **)&h->d_inl, B * C));
  URF_HIP(hipMalloc((void **)&h->d_lvl, B * C));
  *out = h;
  return 0;
}

extern "C" void urf_pose_destroy(urf_pose *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->st) { (void)hipStreamSynchronize(h->st); (void)hipStreamDestroy(h->st); }
  void *bufs[] = {h->d_counts, h->d_ninl, h->d_ok, h->d_count, h->d_obj, h->d_img, h->d_X, h->d_uv, h->d_hyp, h->d_pose,
                  h->d_q, h->d_p, h->d_inl, h->d_lvl, h->d_uv3};
  for (void *p : bufs) (void)hipFree(p);
  delete h;
}

static int pose_check(urf_pose *h, int B, const int *n, int cap) {
  URF_CHECK(h, "pose handle is null");
  URF_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, %d]", B, h->maxB);
  URF_CHECK(n && cap >= 1 && cap <= h->cap, "row capacity %d outside [1, %d]", cap, h->cap);
  for (int f = 0; f < B; ++f) URF_CHECK(n[f] >= 0 && n[f] <= cap, "frame %d: %d observations outside [0, %d]", f, n[f], cap);
  return 0;
}

extern "C" int urf_solve_pnp_ransac(urf_pose *h, const urf_pnp_config *cfg, int B, const int *n, const float *obj,
                                    const float *img, int cap, double *pose, uint8_t *inliers, int *n_inliers) {
  URF_CHECK(cfg && obj && img && pose && inliers && n_inliers, "urf_solve_pnp_ransac: null argument");
  if (pose_check(h, B, n, cap)) return -2;
  const int its = cfg->iterations > 0 ? cfg->iterations : 100;
  URF_CHECK(its <= h->max_iters, "at most %d PnP hypotheses", h->max_iters);
  const double gate = cfg->reprojection_error > 0 ? cfg->reprojection_error : 20.0;
  const double conf = cfg->confidence > 0 ? cfg->confidence : 0.99;
  URF_CHECK(conf < 1.0, "confidence must be below 1");
  const Cam cam = {cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  URF_HIP(hipSetDevice(h->device));
  hipStream_t st = h->st;
  URF_HIP(hipMemcpyAsync(h->d_counts, n, (size_t)B * 4, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_obj, obj, (size_t)B * cap * 12, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_img, img, (size_t)B * cap * 8, hipMemcpyHostToDevice, st));
  const size_t lds = sizeof(double) * (288 + 18) * HT;
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    URF_HIP(hipFuncSetAttribute((const void *)pnp_hypothesis_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark();
  }
  hipLaunchKernelGGL(pnp_hypothesis_kernel, dim3((its + HT - 1) / HT, B), dim3(HT), lds, st, cam, cfg->seed, its, h->d_counts,
                     h->d_obj, h->d_img, cap, h->d_hyp, h->d_ok);
  hipLaunchKernelGGL(pnp_count_kernel, dim3((its + 3) / 4, B), dim3(256), 0, st, cam, gate * gate, its, h->d_counts, h->d_obj,
                     h->d_img, cap, h->d_hyp, h->d_ok, h->d_count);
  hipLaunchKernelGGL(pnp_finish_kernel, dim3(B), dim3(64), 0, st, cam, gate * gate, conf, its, h->d_counts, h->d_obj, h->d_img,
                     cap, h->d_hyp, h->d_count, h->d_X, h->d_uv, h->d_pose, h->d_inl, h->d_ninl);
  URF_HIP(hipGetLastError());
  URF_HIP(hipMemcpyAsync(pose, h->d_pose, (size_t)B * 16 * 8, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(inliers, h->d_inl, (size_t)B * cap, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(n_inliers, h->d_ninl, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipStreamSynchronize(st));
  return 0;
}

extern "C" int urf_frame_optimization(urf_pose *h, const urf_poseopt_config *cfg, int B, const int *n, const double *Xw,
                                      const double *obs, int cap, double *q_wc, double *p_wc, uint8_t *inlier,
                                      int *n_inliers) {
  URF_CHECK(cfg && Xw && obs && q_wc && p_wc && inlier && n_inliers, "urf_frame_optimization: null argument");
  if (pose_check(h, B, n, cap)) return -2;
  const double gate = cfg->chi2_threshold > 0 ? cfg->chi2_threshold : 5.991;
  const Cam cam = {cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  URF_HIP(hipSetDevice(h->device));
  hipStream_t st = h->st;
  URF_HIP(hipMemcpyAsync(h->d_counts, n, (size_t)B * 4, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_X, Xw, (size_t)B * cap * 24, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_uv, obs, (size_t)B * cap * 16, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_q, q_wc, (size_t)B * 32, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_p, p_wc, (size_t)B * 24, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(frame_optimization_kernel, dim3(B), dim3(64), 0, st, cam, 0.0, gate, 0.0, h->d_counts, (const int *)nullptr,
                     h->d_X, h->d_uv, cap, h->d_q, h->d_p, h->d_inl, h->d_lvl, h->d_ninl);
  URF_HIP(hipGetLastError());
  URF_HIP(hipMemcpyAsync(q_wc, h->d_q, (size_t)B * 32, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(p_wc, h->d_p, (size_t)B * 24, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(inlier, h->d_inl, (size_t)B * cap, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(n_inliers, h->d_ninl, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipStreamSynchronize(st));
  return 0;
}

extern "C" int urf_frame_optimization_stereo(urf_pose *h, const urf_poseopt_stereo_config *cfg, int B, const int *n_mono,
                                             const int *n_stereo, const double *Xw, const double *obs, int cap, double *q_wc,
                                             double *p_wc, uint8_t *inlier, int *n_inliers) {
  URF_CHECK(cfg && n_stereo && Xw && obs && q_wc && p_wc && inlier && n_inliers, "urf_frame_optimization_stereo: null argument");
  if (pose_check(h, B, n_mono, cap)) return -2;
  for (int f = 0; f < B; ++f)
    URF_CHECK(n_stereo[f] >= 0 && n_mono[f] + n_stereo[f] <= cap, "frame %d: %d + %d observations above the row capacity %d", f,
              n_mono[f], n_stereo[f], cap);
  const double gate_m = cfg->chi2_mono > 0 ? cfg->chi2_mono : 5.991, gate_s = cfg->chi2_stereo > 0 ? cfg->chi2_stereo : 7.815;
  const Cam cam = {cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  URF_HIP(hipSetDevice(h->device));
  hipStream_t st = h->st;
  URF_HIP(hipMemcpyAsync(h->d_counts, n_mono, (size_t)B * 4, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_count, n_stereo, (size_t)B * 4, hipMemcpyHostToDevice, st));   // (the hypothesis counters double as the stereo counts)
  URF_HIP(hipMemcpyAsync(h->d_X, Xw, (size_t)B * cap * 24, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_uv3, obs, (size_t)B * cap * 24, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_q, q_wc, (size_t)B * 32, hipMemcpyHostToDevice, st));
  URF_HIP(hipMemcpyAsync(h->d_p, p_wc, (size_t)B * 24, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(frame_optimization_kernel, dim3(B), dim3(64), 0, st, cam, cfg->bf, gate_m, gate_s, h->d_counts, h->d_count,
                     h->d_X, h->d_uv3, cap, h->d_q, h->d_p, h->d_inl, h->d_lvl, h->d_ninl);
  URF_HIP(hipGetLastError());
  URF_HIP(hipMemcpyAsync(q_wc, h->d_q, (size_t)B * 32, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(p_wc, h->d_p, (size_t)B * 24, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(inlier, h->d_inl, (size_t)B * cap, hipMemcpyDeviceToHost, st));
  URF_HIP(hipMemcpyAsync(n_inliers, h->d_ninl, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  URF_HIP(hipStreamSynchronize(st));
  return 0;
}
