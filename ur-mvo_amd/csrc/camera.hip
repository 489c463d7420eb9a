// camera.hip -- Camera::UndistortImage (src/camera.cc:116-118 of the reference)
// as the first device stage of the front-end, so that only the raw u8 frame
// crosses PCIe.
//
// The reference builds two CV_32FC1 maps once in Camera::Camera
// (src/camera.cc:69-85, cv::initUndistortRectifyMap or the fisheye variant) and
// calls cv::remap(image, out, map1, map2, INTER_LINEAR) per frame.  Here the
// maps are built once on the host (urf_cam_create) or handed over as they are
// (urf_cam_create_from_maps: the maintainer keeps OpenCV's own maps), converted
// ONCE to OpenCV's fixed-point form -- integer source pixel (sx, sy) as 2 x i16
// and the 2 x 5-bit fraction -- and the per-frame work is one HBM-bound gather
// kernel: 6 map bytes + 4 source taps (L2) in, 1 byte out per pixel.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/urf.h"
#include "urf_common.h"

namespace urf {

// one thread = 4 consecutive output pixels of a row (one 4-byte store).
// xy[pixel] = (sx, sy) as i16 pairs, fr[pixel] = fy * 32 + fx.
__global__ void __launch_bounds__(256) remap_kernel(const uint8_t *src, int H, int W, size_t sstep, size_t sframe,
                                                    const short2 *xy, const uint16_t *fr, int oh, int ow,
                                                    uint8_t *dst, size_t dstep, size_t dframe) {
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (x0 >= ow) return;
  const uint8_t *s = src + (size_t)blockIdx.z * sframe;
  uint8_t *d = dst + (size_t)blockIdx.z * dframe + (size_t)y * dstep;
  uint32_t packed = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = x0 + q;
    int v = 0;
    if (x < ow) {
      const short2 c = xy[(size_t)y * ow + x];
      const int f = fr[(size_t)y * ow + x];
      const int sx = c.x, sy = c.y, fx = f & 31, fy = f >> 5;
      if (!(sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
        const bool x0in = sx >= 0 && sx < W, x1in = sx + 1 >= 0 && sx + 1 < W;
        const bool y0in = sy >= 0 && sy < H, y1in = sy + 1 >= 0 && sy + 1 < H;
        const int p00 = (x0in && y0in) ? s[(size_t)sy * sstep + sx] : 0;
        const int p01 = (x1in && y0in) ? s[(size_t)sy * sstep + sx + 1] : 0;
        const int p10 = (x0in && y1in) ? s[(size_t)(sy + 1) * sstep + sx] : 0;
        const int p11 = (x1in && y1in) ? s[(size_t)(sy + 1) * sstep + sx + 1] : 0;
        // weights sum to 1024 = 32 * 32; OpenCV scales them by 32 to 32768 and rounds with 1 << 14
        v = ((p00 * (32 - fx) * (32 - fy) + p01 * fx * (32 - fy) + p10 * (32 - fx) * fy + p11 * fx * fy) * 32 +
             (1 << 14)) >> 15;
        v = v > 255 ? 255 : v;
      }
    }
    packed |= (uint32_t)v << (8 * q);
  }
  if (x0 + 3 < ow && (dstep & 3) == 0) {
    *(uint32_t *)(d + x0) = packed;
  } else {
    for (int q = 0; q < 4 && x0 + q < ow; ++q) d[x0 + q] = (uint8_t)(packed >> (8 * q));
  }
}

}  // namespace urf

struct urf_cam {
  int width = 0, height = 0;      // output (map) size = input size in the reference
  int device = 0;
  std::vector<float> map1, map2;  // CV_32FC1 maps (host copy, urf_cam_maps)
  short2 *d_xy = nullptr;
  uint16_t *d_fr = nullptr;
  uint8_t *d_in = nullptr, *d_out = nullptr;   // staging for the host-pointer entry
  size_t cap_frames = 0;
  hipStream_t st = nullptr;
};

namespace {

int inv3(const double *s, double *t) {
  const double c0 = s[4] * s[8] - s[5] * s[7], c1 = s[3] * s[8] - s[5] * s[6], c2 = s[3] * s[7] - s[4] * s[6];
  double det = s[0] * c0 - s[1] * c1 + s[2] * c2;
  if (det == 0.0) return -1;
  det = 1.0 / det;
  t[0] = c0 * det;
  t[1] = (s[2] * s[7] - s[1] * s[8]) * det;
  t[2] = (s[1] * s[5] - s[2] * s[4]) * det;
  t[3] = (s[5] * s[6] - s[3] * s[8]) * det;
  t[4] = (s[0] * s[8] - s[2] * s[6]) * det;
  t[5] = (s[2] * s[3] - s[0] * s[5]) * det;
  t[6] = c2 * det;
  t[7] = (s[1] * s[6] - s[0] * s[7]) * det;
  t[8] = (s[0] * s[4] - s[1] * s[3]) * det;
  return 0;
}

// cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap (OpenCV 4.2, scalar loops)
int build_maps(const urf_cam_config *c, std::vector<float> &m1, std::vector<float> &m2) {
  double PR[9], ir[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      PR[3 * i + j] = c->P[3 * i] * c->R[j] + c->P[3 * i + 1] * c->R[3 + j] + c->P[3 * i + 2] * c->R[6 + j];
  if (inv3(PR, ir)) return -1;
  const int W = c->width, H = c->height;
  m1.resize((size_t)W * H);
  m2.resize((size_t)W * H);
  const double fx = c->K[0], fy = c->K[4], u0 = c->K[2], v0 = c->K[5];
  double k[14] = {0};
  for (int i = 0; i < c->n_dist && i < 14; ++i) k[i] = c->D[i];
  for (int i = 0; i < H; ++i) {
    double xw = i * ir[1] + ir[2], yw = i * ir[4] + ir[5], ww = i * ir[7] + ir[8];
    float *r1 = m1.data() + (size_t)i * W, *r2 = m2.data() + (size_t)i * W;
    for (int j = 0; j < W; ++j) {
      double u, v;
      if (c->distortion_type == 0) {
        const double w = 1. / ww, x = xw * w, y = yw * w;
        const double x2 = x * x, y2 = y * y, r2v = x2 + y2, xy2 = 2 * x * y;
        const double kr = (1 + ((k[4] * r2v + k[1]) * r2v + k[0]) * r2v) / (1 + ((k[7] * r2v + k[6]) * r2v + k[5]) * r2v);
        const double xd = (x * kr + k[2] * xy2 + k[3] * (r2v + 2 * x2) + k[8] * r2v + k[9] * r2v * r2v);
        const double yd = (y * kr + k[2] * (r2v + 2 * y2) + k[3] * xy2 + k[10] * r2v + k[11] * r2v * r2v);
        u = fx * xd + u0;
        v = fy * yd + v0;
      } else {
        const double x = xw / ww, y = yw / ww;
        const double r = std::sqrt(x * x + y * y);
        const double th = std::atan(r);
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const double thd = th * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8);
        const double scale = (r == 0) ? 1.0 : thd / r;
        u = fx * x * scale + u0;
        v = fy * y * scale + v0;
      }
      r1[j] = (float)u;
      r2[j] = (float)v;
      xw += ir[0]; yw += ir[3]; ww += ir[6];
    }
  }
  return 0;
}

inline int cv_round(float v) {   // cvRound: half to even; INT_MIN when an int cannot hold it
  if (!(v > -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000u;
  return (int)std::lrintf(v);
}
inline short sat16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

int upload_maps(urf_cam *h) {
  const size_t n = (size_t)h->width * h->height;
  std::vector<short2> xy(n);
  std::vector<uint16_t> fr(n);
  for (size_t i = 0; i < n; ++i) {   // remap()'s CV_32FC1 -> fixed-point conversion, imgwarp.cpp
    const int sx = cv_round(h->map1[i] * 32.0f), sy = cv_round(h->map2[i] * 32.0f);
    xy[i] = short2{sat16(sx >> 5), sat16(sy >> 5)};
    fr[i] = (uint16_t)((sy & 31) * 32 + (sx & 31));
  }
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipMalloc((void **)&h->d_xy, n * sizeof(short2)));
  URF_HIP(hipMalloc((void **)&h->d_fr, n * sizeof(uint16_t)));
  URF_HIP(hipMemcpy(h->d_xy, xy.data(), n * sizeof(short2), hipMemcpyHostToDevice));
  URF_HIP(hipMemcpy(h->d_fr, fr.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
  URF_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  return 0;
}

int launch_remap(urf_cam *h, const uint8_t *d_src, int rows, int cols, size_t sstep, int n, uint8_t *d_dst, hipStream_t st) {
  dim3 grid((h->width + 1023) / 1024, h->height, n);
  hipLaunchKernelGGL(urf::remap_kernel, grid, dim3(256), 0, st, d_src, rows, cols, sstep, (size_t)rows * sstep, h->d_xy,
                     h->d_fr, h->height, h->width, d_dst, (size_t)h->width, (size_t)h->width * h->height);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int urf_cam_create(const urf_cam_config *cfg, urf_cam **out) {
  URF_CHECK(cfg && out, "urf_cam_create: null argument");
  URF_CHECK(cfg->width > 0 && cfg->height > 0 && cfg->width <= 32767 && cfg->height <= 32767, "urf_cam_create: bad image size");
  URF_CHECK(cfg->n_dist >= 0 && cfg->n_dist <= 14, "urf_cam_create: 0..14 distortion coefficients");
  urf_cam *h = new urf_cam;
  h->width = cfg->width; h->height = cfg->height; h->device = cfg->device;
  if (build_maps(cfg, h->map1, h->map2)) {
    delete h;
    URF_CHECK(false, "urf_cam_create: P*R is singular");
  }
  if (upload_maps(h)) { urf_cam_destroy(h); return -1; }
  *out = h;
  return 0;
}

extern "C" int urf_cam_create_from_maps(const float *map1, const float *map2, int width, int height, int device,
                                        urf_cam **out) {
  URF_CHECK(map1 && map2 && out, "urf_cam_create_from_maps: null argument");
  URF_CHECK(width > 0 && height > 0 && width <= 32767 && height <= 32767, "urf_cam_create_from_maps: bad image size");
  urf_cam *h = new urf_cam;
  h->width = width; h->height = height; h->device = device;
  h->map1.assign(map1, map1 + (size_t)width * height);
  h->map2.assign(map2, map2 + (size_t)width * height);
  if (upload_maps(h)) { urf_cam_destroy(h); return -1; }
  *out = h;
  return 0;
}

extern "C" int urf_cam_size(const urf_cam *h, int *width, int *height) {
  URF_CHECK(h && width && height, "urf_cam_size: null argument");
  *width = h->width; *height = h->height;
  return 0;
}

extern "C" void urf_cam_destroy(urf_cam *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->st) { (void)hipStreamSynchronize(h->st); (void)hipStreamDestroy(h->st); }
  (void)hipFree(h->d_xy); (void)hipFree(h->d_fr); (void)hipFree(h->d_in); (void)hipFree(h->d_out);
  delete h;
}

extern "C" int urf_cam_maps(urf_cam *h, float *map1, float *map2) {
  URF_CHECK(h && map1 && map2, "urf_cam_maps: null argument");
  memcpy(map1, h->map1.data(), h->map1.size() * sizeof(float));
  memcpy(map2, h->map2.data(), h->map2.size() * sizeof(float));
  return 0;
}

// Camera::UndistortImage(image, image_undistorted), src/camera.cc:116-118 (host in, host out)
extern "C" int urf_cam_undistort(urf_cam *h, const uint8_t *img, int rows, int cols, size_t step, uint8_t *out,
                                 size_t ostep) {
  URF_CHECK(h && img && out, "urf_cam_undistort: null argument");
  URF_CHECK(rows > 0 && cols > 0 && step >= (size_t)cols && ostep >= (size_t)h->width, "urf_cam_undistort: bad geometry");
  URF_HIP(hipSetDevice(h->device));
  const size_t need = (size_t)rows * cols;
  if (h->cap_frames < need) {
    (void)hipFree(h->d_in); (void)hipFree(h->d_out);
    h->d_in = h->d_out = nullptr; h->cap_frames = 0;
    URF_HIP(hipMalloc((void **)&h->d_in, need));
    URF_HIP(hipMalloc((void **)&h->d_out, (size_t)h->width * h->height));
    h->cap_frames = need;
  }
  URF_HIP(hipMemcpy2DAsync(h->d_in, cols, img, step, cols, rows, hipMemcpyHostToDevice, h->st));
  if (launch_remap(h, h->d_in, rows, cols, (size_t)cols, 1, h->d_out, h->st)) return -1;
  URF_HIP(hipMemcpy2DAsync(out, ostep, h->d_out, h->width, h->width, h->height, hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipStreamSynchronize(h->st));
  return 0;
}

// Device-resident batch: n frames of rows x cols (contiguous, row stride = cols) -> n frames of
// height x width, enqueued on `stream` (a hipStream_t, e.g. urf_sp_stream(): remap then runs in
// order in front of SuperPoint) or on the handle's own stream when NULL (urf_cam_sync waits).
extern "C" int urf_cam_undistort_device(urf_cam *h, const void *d_imgs, int n, int rows, int cols, void *d_out,
                                        void *stream) {
  URF_CHECK(h && d_imgs && d_out && n > 0 && rows > 0 && cols > 0, "urf_cam_undistort_device: bad argument");
  URF_HIP(hipSetDevice(h->device));
  return launch_remap(h, (const uint8_t *)d_imgs, rows, cols, (size_t)cols, n, (uint8_t *)d_out,
                      stream ? (hipStream_t)stream : h->st);
}

extern "C" int urf_cam_sync(urf_cam *h) {
  URF_CHECK(h, "urf_cam_sync: null handle");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  return 0;
}
