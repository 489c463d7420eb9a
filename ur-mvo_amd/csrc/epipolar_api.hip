// epipolar_api.hip -- EpipolarGeometry::reconstruct behind the C ABI
// (src/epipolar_geometry.cc:18-98).  The two 200-hypothesis RANSAC searches
// (_find_H, _find_F) run on the GPU (ransac_kernels.hip); the once-per-sequence
// tail -- model selection, _reconstruct_F/_H, _decompose_E, _check_R_T,
// _triangulate (:451-950) -- is host C++ exactly as in the reference, with
// Eigen::JacobiSVD replaced by Jacobi eigen-solvers on the Gram matrix (f64).
#include "../../include/urf.h"
#include "urf_common.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace urf {
int launch_epipolar_search(const float *keys1, int n1, const float *keys2, int n2, const float *pts0, const float *pts1,
                           const int *d_nm, int nm, float *pn0, float *pn1, float *T, float *F, float *scoreF,
                           float *H, float *scoreH, uint32_t seed, int iters, float sigma, hipStream_t st);

namespace epi {
static const int kSweeps = 12;
static void jacobi_sym(double *a, double *v, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < kSweeps; ++sweep)
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
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
static int argmin_diag(const double *a, int n) {
  int m = 0;
  for (int i = 1; i < n; ++i)
    if (a[i * n + i] < a[m * n + m]) m = i;
  return m;
}
static void mul3(const float *a, const float *b, float *o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      o[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}
static void inv3(const float *m, float *o) {
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = (a * A + b * B) + c * C;
  const float id = 1.0f / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}
static double det3(const double *m) {
  return (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
// A = U diag(w) V^T, w descending (Jacobi on A^T A, f64)
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
      if (g[ord[b] * 3 + ord[b]] > g[ord[a] * 3 + ord[a]]) std::swap(ord[a], ord[b]);
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
// _check_F / _check_H per-match tests (same float expressions as the kernels)
static bool in_F(const float *F, float u1, float v1, float u2, float v2, float inv) {
  const float th = 3.841f;
  bool bIn = true;
  const float a2 = (F[0] * u1 + F[1] * v1) + F[2], b2 = (F[3] * u1 + F[4] * v1) + F[5], c2 = (F[6] * u1 + F[7] * v1) + F[8];
  const float num2 = (a2 * u2 + b2 * v2) + c2;
  if (((num2 * num2) / (a2 * a2 + b2 * b2)) * inv > th) bIn = false;
  const float a1 = (F[0] * u2 + F[3] * v2) + F[6], b1 = (F[1] * u2 + F[4] * v2) + F[7], c1 = (F[2] * u2 + F[5] * v2) + F[8];
  const float num1 = (a1 * u1 + b1 * v1) + c1;
  if (((num1 * num1) / (a1 * a1 + b1 * b1)) * inv > th) bIn = false;
  return bIn;
}
static bool in_H(const float *H21, const float *H12, float u1, float v1, float u2, float v2, float inv) {
  const float th = 5.991f;
  bool bIn = true;
  const float w2 = (float)(1.0 / (double)((H12[6] * u2 + H12[7] * v2) + H12[8]));
  const float u2in1 = ((H12[0] * u2 + H12[1] * v2) + H12[2]) * w2, v2in1 = ((H12[3] * u2 + H12[4] * v2) + H12[5]) * w2;
  if (((u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1)) * inv > th) bIn = false;
  const float w1 = (float)(1.0 / (double)((H21[6] * u1 + H21[7] * v1) + H21[8]));
  const float u1in2 = ((H21[0] * u1 + H21[1] * v1) + H21[2]) * w1, v1in2 = ((H21[3] * u1 + H21[4] * v1) + H21[5]) * w1;
  if (((u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2)) * inv > th) bIn = false;
  return bIn;
}
// _triangulate :928-950
static bool triangulate(const float *x1, const float *x2, const float *P1, const float *P2, float X[3]) {
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
  if (h[3] == 0.0f) return false;
  X[0] = h[0] / h[3]; X[1] = h[1] / h[3]; X[2] = h[2] / h[3];
  return true;
}
// _check_R_T :782-898
static int check_R_T(const float *R, const float *t, const float *keys1, int n1, const float *keys2, const int *mp, int nm,
                     const uint8_t *inl, const float *K, float *P3D, float th2, uint8_t *good, float *parallax) {
  const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  std::fill(good, good + n1, (uint8_t)0);
  std::vector<float> cosv;
  cosv.reserve(nm);
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
    const int i1 = mp[2 * i], i2 = mp[2 * i + 1];
    const float x1[2] = {keys1[2 * i1], keys1[2 * i1 + 1]}, x2[2] = {keys2[2 * i2], keys2[2 * i2 + 1]};
    float p[3] = {0, 0, 0};
    triangulate(x1, x2, P1, P2, p);
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) { good[i1] = 0; continue; }
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
    if ((im1x - x1[0]) * (im1x - x1[0]) + (im1y - x1[1]) * (im1y - x1[1]) > th2) continue;
    const float invZ2 = (float)(1.0 / (double)q[2]);
    const float im2x = fx * q[0] * invZ2 + cx, im2y = fy * q[1] * invZ2 + cy;
    if ((im2x - x2[0]) * (im2x - x2[0]) + (im2y - x2[1]) * (im2y - x2[1]) > th2) continue;
    cosv.push_back(cosParallax);
    P3D[3 * i1] = p[0]; P3D[3 * i1 + 1] = p[1]; P3D[3 * i1 + 2] = p[2];
    nGood++;
    if (cosParallax < 0.99998f) good[i1] = 1;
  }
  if (nGood > 0) {
    std::sort(cosv.begin(), cosv.end());
    const int idx = std::min(50, (int)cosv.size() - 1);
    *parallax = (float)(acos((double)cosv[idx]) * 180.0 / 3.1415926535897932384626433832795);
  } else {
    *parallax = 0.0f;
  }
  return nGood;
}
}  // namespace epi
}  // namespace urf
using namespace urf;

extern "C" void *urf_pm_stream_(urf_pm *h);
extern "C" int urf_pm_device_(urf_pm *h);

extern "C" int urf_epipolar_reconstruct(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1,
                                        const float *keys2, int n2, const int *matches12, float *T21, float *P3D,
                                        uint8_t *tri, int *model, float *scores) {
  URF_CHECK(h && cfg && keys1 && keys2 && matches12 && T21 && P3D && tri && model && scores,
            "urf_epipolar_reconstruct: null argument");
  URF_CHECK(n1 >= 0 && n1 <= kCap && n2 >= 0 && n2 <= kCap, "keypoint counts (%d,%d) outside [0,%d]", n1, n2, kCap);
  hipStream_t st = (hipStream_t)urf_pm_stream_(h);
  URF_CHECK(st, "PointMatching handle is not built");
  for (int k = 0; k < 16; ++k) T21[k] = (k % 5 == 0) ? 1.0f : 0.0f;
  std::fill(tri, tri + n1, (uint8_t)0);
  *model = -1; scores[0] = scores[1] = 0.0f;
  std::vector<int> mp;
  for (int i = 0; i < n1; ++i)
    if (matches12[i] >= 0) { URF_CHECK(matches12[i] < n2, "match index out of range"); mp.push_back(i); mp.push_back(matches12[i]); }
  const int nm = (int)mp.size() / 2;
  if (nm < 8) return 0;
  const int its = cfg->iterations > 0 ? cfg->iterations : 200;
  const float sigma = cfg->sigma > 0 ? cfg->sigma : 1.0f;
  std::vector<float> p0(2 * nm), p1(2 * nm);
  for (int i = 0; i < nm; ++i) {
    p0[2 * i] = keys1[2 * mp[2 * i]]; p0[2 * i + 1] = keys1[2 * mp[2 * i] + 1];
    p1[2 * i] = keys2[2 * mp[2 * i + 1]]; p1[2 * i + 1] = keys2[2 * mp[2 * i + 1] + 1];
  }
  // ---- GPU: both RANSAC searches
  URF_HIP(hipSetDevice(urf_pm_device_(h)));
  float *d = nullptr;
  const size_t nf = 2 * (size_t)n1 + 2 * (size_t)n2 + 8 * (size_t)nm + 18 + (size_t)its * (9 + 1 + 18 + 1) + 8;
  URF_HIP(hipMalloc((void **)&d, nf * sizeof(float)));
  float *dk1 = d, *dk2 = dk1 + 2 * n1, *dp0 = dk2 + 2 * n2, *dp1 = dp0 + 2 * nm, *dq0 = dp1 + 2 * nm, *dq1 = dq0 + 2 * nm;
  float *dT = dq1 + 2 * nm, *dF = dT + 18, *dsF = dF + (size_t)its * 9, *dH = dsF + its, *dsH = dH + (size_t)its * 18;
  int *dnm = (int *)(dsH + its);
  int rc = 0;
  std::vector<float> F((size_t)its * 9), H((size_t)its * 18), sF(its), sH(its);
  do {
    if (hipMemcpyAsync(dk1, keys1, 8 * (size_t)n1, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(dk2, keys2, 8 * (size_t)n2, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(dp0, p0.data(), 8 * (size_t)nm, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(dp1, p1.data(), 8 * (size_t)nm, hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(dnm, &nm, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { rc = -1; break; }
    if (launch_epipolar_search(dk1, n1, dk2, n2, dp0, dp1, dnm, nm, dq0, dq1, dT, dF, dsF, dH, dsH, cfg->seed, its, sigma, st)) { rc = -1; break; }
    if (hipMemcpyAsync(F.data(), dF, F.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(H.data(), dH, H.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(sF.data(), dsF, its * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -1; break; }
    if (hipMemcpyAsync(sH.data(), dsH, its * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = -1; break; }
    if (hipStreamSynchronize(st) != hipSuccess) { rc = -1; break; }
  } while (0);
  (void)hipFree(d);
  URF_CHECK(rc == 0, "urf_epipolar_reconstruct: HIP error %s", hipGetErrorString(hipGetLastError()));

  // ---- host tail: reconstruct() :86-97
  float SF = 0.0f, SH = 0.0f;
  int bF = -1, bH = -1;
  for (int it = 0; it < its; ++it) {
    if (sF[it] > SF) { SF = sF[it]; bF = it; }
    if (sH[it] > SH) { SH = sH[it]; bH = it; }
  }
  scores[0] = SH; scores[1] = SF;
  if (SH + SF == 0.0f) return 0;
  const float *K = cfg->K;
  const float inv = (float)(1.0 / (double)(sigma * sigma));
  const float th2 = 4.0f * (sigma * sigma);
  const float minParallax = 1.0f;
  const int minTri = 50;
  std::vector<uint8_t> inl(nm), gd(n1 > 0 ? n1 : 1);
  std::vector<float> P(3 * (size_t)(n1 > 0 ? n1 : 1));
  const float RH = SH / (SH + SF);
  int ok = 0;
  if (RH > 0.50f && bH >= 0) {  // _reconstruct_H :564-733
    *model = 0;
    const float *H21 = H.data() + (size_t)bH * 18, *H12 = H21 + 9;
    int N = 0;
    for (int i = 0; i < nm; ++i) { inl[i] = epi::in_H(H21, H12, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], inv); N += inl[i]; }
    float invK[9], M[9], A[9];
    epi::inv3(K, invK); epi::mul3(invK, H21, M); epi::mul3(M, K, A);
    double U[9], w[3], V[9], Vt[9];
    epi::svd3(A, U, w, V);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
    const float s = (float)(epi::det3(U) * epi::det3(Vt));
    const float d1 = (float)w[0], d2 = (float)w[1], d3 = (float)w[2];
    if (d1 / d2 < 1.00001f || d2 / d3 < 1.00001f) return 0;
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
      const int i = h8 & 3;
      const bool second = h8 >= 4;
      float Rp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tp[3];
      if (!second) { Rp[0] = ctheta; Rp[2] = -stheta[i]; Rp[4] = 1.0f; Rp[6] = stheta[i]; Rp[8] = ctheta;
                     tp[0] = x1[i] * (d1 - d3); tp[1] = 0.0f; tp[2] = -x3[i] * (d1 - d3); }
      else { Rp[0] = cphi; Rp[2] = sphi[i]; Rp[4] = -1.0f; Rp[6] = sphi[i]; Rp[8] = -cphi;
             tp[0] = x1[i] * (d1 + d3); tp[1] = 0.0f; tp[2] = x3[i] * (d1 + d3); }
      float M1[9], M2[9];
      epi::mul3(Uf, Rp, M1); epi::mul3(M1, Vtf, M2);
      for (int k = 0; k < 9; ++k) Rs[h8][k] = s * M2[k];
      float tt[3];
      for (int r = 0; r < 3; ++r) tt[r] = (Uf[r * 3] * tp[0] + Uf[r * 3 + 1] * tp[1]) + Uf[r * 3 + 2] * tp[2];
      const float nrm = sqrtf((tt[0] * tt[0] + tt[1] * tt[1]) + tt[2] * tt[2]);
      for (int r = 0; r < 3; ++r) ts[h8][r] = tt[r] / nrm;
    }
    int bestGood = 0, second = 0, bestIdx = -1;
    float bestPar = -1.0f;
    std::vector<uint8_t> bg(gd.size());
    std::vector<float> bP(P.size());
    for (int h8 = 0; h8 < 8; ++h8) {
      float par;
      std::fill(P.begin(), P.end(), 0.0f);
      const int nG = epi::check_R_T(Rs[h8], ts[h8], keys1, n1, keys2, mp.data(), nm, inl.data(), K, P.data(), th2, gd.data(), &par);
      if (nG > bestGood) { second = bestGood; bestGood = nG; bestIdx = h8; bestPar = par; bg = gd; bP = P; }
      else if (nG > second) second = nG;
    }
    if (second < 0.75 * bestGood && bestPar >= minParallax && bestGood > minTri && bestGood > 0.9 * N) {
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T21[r * 4 + c] = Rs[bestIdx][r * 3 + c]; T21[r * 4 + 3] = ts[bestIdx][r]; }
      memcpy(tri, bg.data(), n1); memcpy(P3D, bP.data(), 12 * (size_t)n1);
      ok = 1;
    }
  } else if (bF >= 0) {  // _reconstruct_F :451-562
    *model = 1;
    const float *F21 = F.data() + (size_t)bF * 9;
    int N = 0;
    for (int i = 0; i < nm; ++i) { inl[i] = epi::in_F(F21, p0[2 * i], p0[2 * i + 1], p1[2 * i], p1[2 * i + 1], inv); N += inl[i]; }
    float Kt[9], M[9], E[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Kt[i * 3 + j] = K[j * 3 + i];
    epi::mul3(Kt, F21, M); epi::mul3(M, K, E);
    double U[9], w[3], V[9];
    epi::svd3(E, U, w, V);  // _decompose_E :900-926
    float Uf[9], Vtf[9], t[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Uf[i * 3 + j] = (float)U[i * 3 + j]; Vtf[i * 3 + j] = (float)V[j * 3 + i]; }
    { const float nrm = sqrtf((Uf[2] * Uf[2] + Uf[5] * Uf[5]) + Uf[8] * Uf[8]); t[0] = Uf[2] / nrm; t[1] = Uf[5] / nrm; t[2] = Uf[8] / nrm; }
    const float Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    float R1[9], R2[9], M1[9];
    epi::mul3(Uf, Wm, M1); epi::mul3(M1, Vtf, R1);
    epi::mul3(Uf, Wt, M1); epi::mul3(M1, Vtf, R2);
    { double dd[9]; for (int k = 0; k < 9; ++k) dd[k] = R1[k]; if (epi::det3(dd) < 0) for (int k = 0; k < 9; ++k) R1[k] = -R1[k]; }
    { double dd[9]; for (int k = 0; k < 9; ++k) dd[k] = R2[k]; if (epi::det3(dd) < 0) for (int k = 0; k < 9; ++k) R2[k] = -R2[k]; }
    const float t2[3] = {-t[0], -t[1], -t[2]};
    const float *Rc[4] = {R1, R2, R1, R2};
    const float *tc[4] = {t, t, t2, t2};
    int nG[4];
    float par[4];
    std::vector<uint8_t> gds(4 * gd.size());
    std::vector<float> Ps(4 * P.size(), 0.0f);
    for (int c = 0; c < 4; ++c)
      nG[c] = epi::check_R_T(Rc[c], tc[c], keys1, n1, keys2, mp.data(), nm, inl.data(), K, Ps.data() + P.size() * c, th2,
                             gds.data() + gd.size() * c, &par[c]);
    int maxGood = nG[0];
    for (int c = 1; c < 4; ++c) maxGood = std::max(maxGood, nG[c]);
    const int nMinGood = std::max((int)(0.9 * N), minTri);
    int nsimilar = 0;
    for (int c = 0; c < 4; ++c) if (nG[c] > 0.7 * maxGood) nsimilar++;
    if (!(maxGood < nMinGood || nsimilar > 1)) {
      for (int c = 0; c < 4; ++c)
        if (maxGood == nG[c]) {
          if (par[c] > minParallax) {
            for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) T21[r * 4 + cc] = Rc[c][r * 3 + cc]; T21[r * 4 + 3] = tc[c][r]; }
            memcpy(tri, gds.data() + gd.size() * c, n1); memcpy(P3D, Ps.data() + P.size() * c, 12 * (size_t)n1);
            ok = 1;
          }
          break;
        }
    }
  }
  return ok;
}
