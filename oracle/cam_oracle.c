/* cam_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Camera::Camera map construction and Camera::UndistortImage
 * (src/camera.cc:69-85, 116-125 of the reference).  The arithmetic lives in a
 * third-party dependency that is NOT under /root/reference: OpenCV
 * (CMakeLists.txt:15 asks for >= 4.2; the reference image installs Ubuntu
 * 20.04's libopencv-dev = 4.2.0, docker/Dockerfile:4,132):
 *   cv::initUndistortRectifyMap           imgproc/src/undistort.cpp (4.2.0)
 *   cv::fisheye::initUndistortRectifyMap  calib3d/src/fisheye.cpp   (4.2.0)
 *   cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) on CV_8UC1 with CV_32FC1 maps
 *                                         imgproc/src/imgwarp.cpp   (4.2.0)
 * This file restates their published algorithms.  The reference holds no test
 * or golden vector for them and OpenCV is not in the image: PARITY UNPINNED
 * (checked only against closed-form cases, tests/test_oracle_golden.py).
 *
 * Known, documented differences from an OpenCV binary: (1) OpenCV builds with
 * AVX2 compute a map row with a different operation order than the scalar loop
 * restated here (last-ulp differences of the f64 value before the cast to
 * float); (2) fisheye inverts P*R through an SVD, here the 3x3 cofactor
 * formula is used (same matrix to ~1e-16).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "urf_oracle.h"

/* 3x3 inverse, cofactor form (cv::invert's n == 3 fast path, DECOMP_LU) */
static int inv3(const double *s, double *t) {
  double d = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
             s[2] * (s[3] * s[7] - s[4] * s[6]);
  if (d == 0.0) return -1;
  d = 1.0 / d;
  t[0] = (s[4] * s[8] - s[5] * s[7]) * d;
  t[1] = (s[2] * s[7] - s[1] * s[8]) * d;
  t[2] = (s[1] * s[5] - s[2] * s[4]) * d;
  t[3] = (s[5] * s[6] - s[3] * s[8]) * d;
  t[4] = (s[0] * s[8] - s[2] * s[6]) * d;
  t[5] = (s[2] * s[3] - s[0] * s[5]) * d;
  t[6] = (s[3] * s[7] - s[4] * s[6]) * d;
  t[7] = (s[1] * s[6] - s[0] * s[7]) * d;
  t[8] = (s[0] * s[4] - s[1] * s[3]) * d;
  return 0;
}

static void mul3(const double *a, const double *b, double *c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

/* src/camera.cc:69-85: distortion_type 0 -> cv::initUndistortRectifyMap,
 * otherwise cv::fisheye::initUndistortRectifyMap; maps are CV_32FC1. */
int ocam_init_maps(const ocam_config *c, float *map1, float *map2) {
  double PR[9], ir[9];
  mul3(c->P, c->R, PR);
  if (inv3(PR, ir)) return -1;
  const double fx = c->K[0], fy = c->K[4], u0 = c->K[2], v0 = c->K[5];
  const int W = c->width, H = c->height;
  if (c->distortion_type == 0) {
    double k[14];
    memset(k, 0, sizeof k);
    for (int i = 0; i < c->n_dist && i < 14; ++i) k[i] = c->D[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];   /* tilt (k[12], k[13]) not supported: identity */
    for (int i = 0; i < H; ++i) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < W; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
        map1[(size_t)i * W + j] = (float)(fx * xd + u0);
        map2[(size_t)i * W + j] = (float)(fy * yd + v0);
      }
    }
  } else {
    double k[4] = {0, 0, 0, 0};
    for (int i = 0; i < c->n_dist && i < 4; ++i) k[i] = c->D[i];
    for (int i = 0; i < H; ++i) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < W; ++j) {
        const double x = _x / _w, y = _y / _w;
        const double r = sqrt(x * x + y * y);
        const double theta = atan(r);
        const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
        const double theta_d = theta * (1 + k[0] * theta2 + k[1] * theta4 + k[2] * theta6 + k[3] * theta8);
        const double scale = (r == 0) ? 1.0 : theta_d / r;
        map1[(size_t)i * W + j] = (float)(fx * x * scale + u0);
        map2[(size_t)i * W + j] = (float)(fy * y * scale + v0);
        _x += ir[0]; _y += ir[3]; _w += ir[6];
      }
    }
  }
  return 0;
}

/* cvRound(float): round half to even, INT_MIN for values an int cannot hold */
static int cv_round(float v) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000u;
  return (int)lrintf(v);
}
static int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* Camera::UndistortImage src/camera.cc:116-118 = cv::remap(image, out, map1, map2, INTER_LINEAR).
 * Fixed-point bilinear of imgwarp.cpp: coordinates are quantised to 1/32 pixel
 * (INTER_BITS 5), the four weights are (32-fx)(32-fy), fx(32-fy), (32-fx)fy,
 * fx fy scaled to a sum of 32768 (INTER_REMAP_COEF_BITS 15), the result is
 * (sum + 16384) >> 15; taps outside the image read 0 (BORDER_CONSTANT). */
void ocam_remap(const uint8_t *img, int H, int W, size_t step, const float *map1, const float *map2, int oh, int ow,
                uint8_t *out, size_t ostep) {
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      const int sxq = cv_round(map1[(size_t)y * ow + x] * 32.0f);
      const int syq = cv_round(map2[(size_t)y * ow + x] * 32.0f);
      const int sx = sat_short(sxq >> 5), sy = sat_short(syq >> 5);
      const int fx = sxq & 31, fy = syq & 31;
      int v = 0;
      if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const int in_x0 = sx >= 0 && sx < W, in_x1 = sx + 1 >= 0 && sx + 1 < W;
        const int in_y0 = sy >= 0 && sy < H, in_y1 = sy + 1 >= 0 && sy + 1 < H;
        const int p00 = (in_x0 && in_y0) ? img[(size_t)sy * step + sx] : 0;
        const int p01 = (in_x1 && in_y0) ? img[(size_t)sy * step + sx + 1] : 0;
        const int p10 = (in_x0 && in_y1) ? img[(size_t)(sy + 1) * step + sx] : 0;
        const int p11 = (in_x1 && in_y1) ? img[(size_t)(sy + 1) * step + sx + 1] : 0;
        v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
        if (v > 255) v = 255;
      }
      out[(size_t)y * ostep + x] = (uint8_t)v;
    }
}
