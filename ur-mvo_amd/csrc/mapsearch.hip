// mapsearch.hip -- Mapping::SearchByProjection (src/mapping.cc:667-735 of the
// reference) on the GPU: SURVEY.md section 8, row f4.  For every map point:
// project into the frame (Camera::Project, include/camera.h:48-68), collect the
// keypoints inside the (2r)^2 window that do not carry a good map point yet
// (Frame::FindNeighborKeypoints, src/frame.cc:320-353), descriptor distance
// 2 (1 - f1^T f2) in f64 (src/utils.cc:14-19), first-best-wins minimum, ratio
// test against the second best.  One wave per map point; a lane owns the
// keypoints l, l + 64, ... and the wave reduces (distance, visiting order) with
// a lexicographic minimum, which reproduces the reference's sequential scan
// (strict <, candidates visited grid column, grid row, keypoint index).
// Arithmetic left to Eigen by the reference is fixed by the written
// specification in DESIGN.md section 6 (component sums ((a+b)+c), ascending fma
// chain for the dot product).
#include <cmath>
#include <cstring>

#include "../../include/urf.h"
#include "urf_common.h"

namespace urf {

constexpr int kGridRows = 48, kGridCols = 64;   // include/frame.h:16-17

struct SbpArgs {
  double fx, fy, cx, cy, width, height, r;
  double R[9], t[3];          // Rwc row-major, twc
  const double *feat;         // column-major 259 x K (f64), or NULL when `slot` is set
  const float *slot;          // device feature slot (header, meta[cap][4], desc[cap][256]), f32 widened exactly
  int K;
  const uint8_t *occupied;    // K flags or NULL
  const double *mp_pos, *mp_desc;
  const uint8_t *mp_valid;    // M flags or NULL
  int M;
  int *best_idx;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) search_by_projection_kernel(SbpArgs a) {
  __shared__ double sdesc[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  int result = -1;
  const int K = a.slot ? min(((const int *)a.slot)[0], a.K) : a.K;
  bool live = !(a.mp_valid && !a.mp_valid[m]);
  double u = 0, v = 0;
  if (live) {
    const double d0 = a.mp_pos[3 * m] - a.t[0], d1 = a.mp_pos[3 * m + 1] - a.t[1], d2 = a.mp_pos[3 * m + 2] - a.t[2];
    const double pc0 = (a.R[0] * d0 + a.R[3] * d1) + a.R[6] * d2;   // Rwc^T (pw - twc)
    const double pc1 = (a.R[1] * d0 + a.R[4] * d1) + a.R[7] * d2;
    const double pc2 = (a.R[2] * d0 + a.R[5] * d1) + a.R[8] * d2;
    live = pc2 > 0;
    if (live) {
      const double z_inv = 1.0 / pc2;
      u = (pc0 * z_inv) * a.fx + a.cx;
      v = (pc1 * z_inv) * a.fy + a.cy;
      live = !(u <= 0 || u >= a.width || v <= 0 || v >= a.height);
    }
  }
  if (live) {   // wave-uniform
    for (int c = lane; c < 256; c += 64) sdesc[wave][c] = a.mp_desc[(size_t)256 * m + c];
    __builtin_amdgcn_wave_barrier();
    const double gwi = (double)kGridCols / a.width, ghi = (double)kGridRows / a.height;
    double b1 = 4.0, b2 = 4.0;        // this lane's best and second-best distance
    int k1 = 0x7fffffff, i1 = -1;     // visiting-order key and index of its best
    for (int k = lane; k < K; k += 64) {
      double x, y;
      if (a.slot) {
        const float *meta = a.slot + kSlotHeader + (size_t)k * 4;   // {score, x, y, 0}
        x = (double)meta[1]; y = (double)meta[2];
      } else {
        x = a.feat[(size_t)259 * k + 1]; y = a.feat[(size_t)259 * k + 2];
      }
      if (a.occupied && a.occupied[k]) continue;
      const double dx = (double)(float)x - u, dy = (double)(float)y - v;   // cv::KeyPoint::pt is float
      if (!(fabs(dx) < a.r && fabs(dy) < a.r)) continue;
      double dot = 0.0;
      if (a.slot) {
        const float *dk = a.slot + kSlotHeader + (size_t)kCap * 4 + (size_t)k * 256;
        for (int ch = 0; ch < 256; ++ch) dot = fma(sdesc[wave][ch], (double)dk[ch], dot);
      } else {
        const double *dk = a.feat + (size_t)259 * k + 3;
        for (int ch = 0; ch < 256; ++ch) dot = fma(sdesc[wave][ch], dk[ch], dot);
      }
      const double dist = 2 * (1.0 - dot);
      const int gx = clampi((int)round(x * gwi), 0, kGridCols - 1), gy = clampi((int)round(y * ghi), 0, kGridRows - 1);
      const int key = (gx * kGridRows + gy) * 2048 + k;
      if (dist < b1 || (dist == b1 && key < k1)) {
        if (i1 >= 0 && b1 < b2) b2 = b1;
        b1 = dist; k1 = key; i1 = k;
      } else if (dist < b2) {
        b2 = dist;
      }
    }
    // wave: winner = lexicographic minimum of (b1, k1); second = min over lanes of (winner ? b2 : b1)
    double wb = b1;
    int wk = k1, wi = i1;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const double ob = shfl_xor_f64(wb, s);
      const int ok = __shfl_xor(wk, s, 64), oi = __shfl_xor(wi, s, 64);
      if (ob < wb || (ob == wb && ok < wk)) { wb = ob; wk = ok; wi = oi; }
    }
    double sec = (i1 == wi && i1 >= 0) ? b2 : b1;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) sec = fmin(sec, shfl_xor_f64(sec, s));
    // the reference's `best_dist` only moves on a strict improvement over 4.0
    if (wi >= 0 && wb < 4.0 && wb < 0.35 && wb < 0.6 * sec) result = wi;
  }
  if (lane == 0) a.best_idx[m] = result;
}

}  // namespace urf

namespace {

int sbp_run(const urf_sbp_config *cfg, const double *feat, const void *d_slot, int K, const uint8_t *occupied,
            const double *mp_pos, const double *mp_desc, const uint8_t *mp_valid, int M, int *best_idx) {
  URF_CHECK(cfg && mp_pos && mp_desc && best_idx && M >= 0 && (feat || d_slot), "urf_search_by_projection: null argument");
  URF_CHECK(K >= 0 && K <= URF_MAX_KEYPOINTS, "urf_search_by_projection: %d keypoints (0..%d)", K, URF_MAX_KEYPOINTS);
  if (M == 0) return 0;
  URF_HIP(hipSetDevice(cfg->device));
  urf::SbpArgs a = {};
  a.fx = cfg->fx; a.fy = cfg->fy; a.cx = cfg->cx; a.cy = cfg->cy; a.width = cfg->image_width; a.height = cfg->image_height;
  a.r = 15.0 * cfg->thr;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) a.R[3 * i + j] = cfg->pose[4 * i + j];
    a.t[i] = cfg->pose[4 * i + 3];
  }
  a.K = K; a.M = M;
  // one allocation for everything this call uploads (the mapping thread calls this a few times per keyframe)
  const size_t n_feat = feat ? (size_t)259 * K * 8 : 0, n_occ = occupied ? (size_t)((K + 7) & ~7) : 0;
  const size_t n_pos = (size_t)M * 24, n_desc = (size_t)M * 256 * 8, n_val = mp_valid ? (size_t)((M + 7) & ~7) : 0;
  const size_t n_out = (size_t)M * 4;
  uint8_t *buf = nullptr;
  URF_HIP(hipMalloc((void **)&buf, n_feat + n_pos + n_desc + n_out + n_occ + n_val + 64));
  uint8_t *p = buf;
  auto up = [&](const void *src, size_t n) -> uint8_t * {
    uint8_t *d = p;
    if (n && src) (void)hipMemcpy(d, src, n, hipMemcpyHostToDevice);
    p += (n + 7) & ~(size_t)7;
    return d;
  };
  if (feat) a.feat = (const double *)up(feat, n_feat);
  else a.slot = (const float *)d_slot;
  a.mp_pos = (const double *)up(mp_pos, n_pos);
  a.mp_desc = (const double *)up(mp_desc, n_desc);
  a.best_idx = (int *)up(nullptr, n_out);
  if (occupied) a.occupied = up(occupied, (size_t)K);
  if (mp_valid) a.mp_valid = up(mp_valid, (size_t)M);
  hipLaunchKernelGGL(urf::search_by_projection_kernel, dim3((M + 3) / 4), dim3(256), 0, 0, a);
  hipError_t rc = hipGetLastError();
  if (rc == hipSuccess) rc = hipMemcpy(best_idx, a.best_idx, n_out, hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  URF_CHECK(rc == hipSuccess, "urf_search_by_projection: %s", hipGetErrorString(rc));
  return 0;
}

}  // namespace

extern "C" int urf_search_by_projection(const urf_sbp_config *cfg, const double *feat, int K, const uint8_t *occupied,
                                        const double *mp_pos, const double *mp_desc, const uint8_t *mp_valid, int M,
                                        int *best_idx) {
  return sbp_run(cfg, feat, nullptr, K, occupied, mp_pos, mp_desc, mp_valid, M, best_idx);
}

extern "C" int urf_search_by_projection_slot(const urf_sbp_config *cfg, const void *d_slot, int K, const uint8_t *occupied,
                                             const double *mp_pos, const double *mp_desc, const uint8_t *mp_valid, int M,
                                             int *best_idx) {
  URF_CHECK(d_slot, "urf_search_by_projection_slot: null slot");
  return sbp_run(cfg, nullptr, d_slot, K, occupied, mp_pos, mp_desc, mp_valid, M, best_idx);
}
