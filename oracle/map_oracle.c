/* map_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Mapping::SearchByProjection (src/mapping.cc:667-735 of the reference) with
 * what it calls: Camera::Project (include/camera.h:48-68), Frame::FindGrid and
 * Frame::FindNeighborKeypoints (src/frame.cc:70-80, 320-353), DescriptorDistance
 * (src/utils.cc:14-19).  SURVEY.md section 8, row f4.
 *
 * Written specification where the reference leaves the arithmetic to Eigen
 * (unpinned at the last ulp; the reference holds no test for this function):
 *   pc      = Rwc^T (pw - twc), each component ((a + b) + c), no contraction
 *   f1^T f2 = fma chain over the 256 channels in ascending order, from +0
 * Everything else follows the reference lines: strict comparisons, candidate
 * order = grid column, grid row, keypoint index (the order the reference walks
 * _feature_grid), first best wins, second best starts at 4.0.
 */
#include <math.h>
#include <stdint.h>

#include "urf_oracle.h"

#define GRID_ROWS 48 /* include/frame.h:16 */
#define GRID_COLS 64 /* include/frame.h:17 */

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int osbp_search(const osbp_config *c, const double *feat, int K, const uint8_t *occupied, const double *mp_pos,
                const double *mp_desc, const uint8_t *mp_valid, int M, int *best_idx) {
  const double *P = c->pose; /* Twc row-major: Rwc = P[0..2][0..2], twc = P[.][3] */
  const double gwi = (double)GRID_COLS / c->image_width, ghi = (double)GRID_ROWS / c->image_height;
  const double r = 15.0 * c->thr;
  for (int m = 0; m < M; ++m) {
    best_idx[m] = -1;
    if (mp_valid && !mp_valid[m]) continue;
    const double d0 = mp_pos[3 * m] - P[3], d1 = mp_pos[3 * m + 1] - P[7], d2 = mp_pos[3 * m + 2] - P[11];
    const double pc0 = (P[0] * d0 + P[4] * d1) + P[8] * d2;
    const double pc1 = (P[1] * d0 + P[5] * d1) + P[9] * d2;
    const double pc2 = (P[2] * d0 + P[6] * d1) + P[10] * d2;
    if (pc2 <= 0) continue;
    const double z_inv = 1.0 / pc2;
    const double u = (pc0 * z_inv) * c->fx + c->cx, v = (pc1 * z_inv) * c->fy + c->cy;
    if (u <= 0 || u >= c->image_width || v <= 0 || v >= c->image_height) continue;
    double best = 4.0, second = 4.0;
    int bi = -1;
    /* walk the candidates in the reference's order: gx, then gy, then insertion (= index) order */
    const int gx0 = clampi((int)floor((u - r) * gwi), 0, 1 << 30), gx1 = clampi((int)ceil((u + r) * gwi), -(1 << 30), GRID_COLS - 1);
    const int gy0 = clampi((int)floor((v - r) * ghi), 0, 1 << 30), gy1 = clampi((int)ceil((v + r) * ghi), -(1 << 30), GRID_ROWS - 1);
    if (gx0 >= GRID_COLS || gx1 < 0 || gy0 >= GRID_ROWS || gy1 < 0) continue;
    for (int gx = gx0; gx <= gx1; ++gx)
      for (int gy = gy0; gy <= gy1; ++gy)
        for (int k = 0; k < K; ++k) {
          const double x = feat[(size_t)259 * k + 1], y = feat[(size_t)259 * k + 2];
          /* Frame::FindGrid: round, then clamp */
          if (clampi((int)round(x * gwi), 0, GRID_COLS - 1) != gx || clampi((int)round(y * ghi), 0, GRID_ROWS - 1) != gy) continue;
          if (occupied && occupied[k]) continue;
          const double dx = (double)(float)x - u, dy = (double)(float)y - v; /* cv::KeyPoint::pt is float */
          if (!(fabs(dx) < r && fabs(dy) < r)) continue;
          double dot = 0.0;
          for (int ch = 0; ch < 256; ++ch) dot = fma(mp_desc[(size_t)256 * m + ch], feat[(size_t)259 * k + 3 + ch], dot);
          const double dist = 2 * (1.0 - dot);
          if (dist < best) { second = best; best = dist; bi = k; }
          else if (dist < second) second = dist;
        }
    if (best < 0.35 && best < 0.6 * second) best_idx[m] = bi;
  }
  return 0;
}
