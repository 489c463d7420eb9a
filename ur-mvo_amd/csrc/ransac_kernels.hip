// ransac_kernels.hip -- epipolar 8-point RANSAC on the GPU: the outlier stage of
// PointMatching::MatchingPoints (src/point_matching.cc:47-58).  The reference
// calls cv::findFundamentalMat there (OpenCV, un-vendored, unpinned); this build
// runs the in-tree routine EpipolarGeometry::_find_F / _normalize /
// _compute_F21 / _check_F (src/epipolar_geometry.cc:161-205,735-780,247-283,
// 372-449) with the deviations listed in DESIGN.md "RANSAC" (counter-hash
// sampler, pivoted elimination + 3x3 Jacobi in double, canonical wave-order
// float sums).
//
// Kernels: normalise (1 wave per point set) -> one thread per hypothesis
// (8-point solve, double) -> one wave per hypothesis (score all matches) ->
// one workgroup per pair (first-best hypothesis, inlier mask, ordered compaction).
#include "urf_common.h"
#include "urf_math.h"

namespace urf {

constexpr int RNP = kCap;

struct DMatchR { int queryIdx, trainIdx; float distance; };

__device__ __forceinline__ uint32_t rs_hash(uint32_t seed, uint32_t ctr) {
  uint32_t x = seed ^ (ctr * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// Random::RandomInt (src/epipolar_geometry.cc:114-117), rand() -> 31-bit counter hash
__device__ __forceinline__ int random_int(uint32_t seed, uint32_t ctr, int mn, int mx) {
  const int d = mx - mn + 1;
  const uint32_t r = rs_hash(seed, ctr) >> 1;
  return (int)(((double)r / 2147483648.0) * d) + mn;
}
// minimal-set draw (:59-71): swap-with-back sampling via a sparse map
__device__ void draw_set(uint32_t seed, int it, int n, int set[8]) {
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

#define JAC_SWEEPS 12
// cyclic Jacobi on a symmetric 3x3 (double); fully unrolled -> registers
__device__ __forceinline__ void jacobi_sym3(double *a, double *v) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i * 3 + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < JAC_SWEEPS; ++sweep) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k * 3 + p], akq = a[k * 3 + q];
          a[k * 3 + p] = c * akp - s * akq;
          a[k * 3 + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
          a[p * 3 + k] = c * apk - s * aqk;
          a[q * 3 + k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k * 3 + p], vkq = v[k * 3 + q];
          v[k * 3 + p] = c * vkp - s * vkq;
          v[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
}

// Null vector of the 8x9 design matrix by fully pivoted Gaussian elimination
// (f64).  The matrix lives in LDS as A(r,c) = lds[(r*9+c)*64 + lane] so the
// data-dependent row/column indices cost no scratch traffic and no bank
// conflicts (the bank depends on the lane only).
#define RA(r, c) lA[((r) * 9 + (c)) * 64]
__device__ void null_vector_8x9(double *lA, int *lperm, double f[9]) {
  for (int c = 0; c < 9; ++c) lperm[c * 64] = c;
  int rank = 8;
  for (int s = 0; s < 8; ++s) {
    double best = 0.0;
    int pr = -1, pc = -1;
    for (int r = s; r < 8; ++r)
      for (int c = s; c < 9; ++c) {
        const double v = fabs(RA(r, c));
        if (v > best) { best = v; pr = r; pc = c; }
      }
    if (pr < 0) { rank = s; break; }
    if (pr != s)
      for (int c = 0; c < 9; ++c) { const double t = RA(s, c); RA(s, c) = RA(pr, c); RA(pr, c) = t; }
    if (pc != s) {
      for (int r = 0; r < 8; ++r) { const double t = RA(r, s); RA(r, s) = RA(r, pc); RA(r, pc) = t; }
      const int t = lperm[s * 64]; lperm[s * 64] = lperm[pc * 64]; lperm[pc * 64] = t;
    }
    const double piv = RA(s, s);
    for (int r = s + 1; r < 8; ++r) {
      const double m = RA(r, s) / piv;
      RA(r, s) = 0.0;
      for (int c = s + 1; c < 9; ++c) RA(r, c) = RA(r, c) - m * RA(s, c);
    }
  }
  // back substitution; g reuses row 0.. of a second LDS strip: keep it in registers
  double g[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) g[c] = (c == rank) ? 1.0 : 0.0;
#pragma unroll
  for (int s = 7; s >= 0; --s) {
    if (s < rank) {
      double sum = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c)
        if (c > s) sum = sum + RA(s, c) * g[c];
      g[s] = -sum / RA(s, s);
    }
  }
  double ss = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) ss = ss + g[c] * g[c];
  const double inv = 1.0 / sqrt(ss);
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const int pc = lperm[c * 64];
    const double val = g[c] * inv;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (k == pc) f[k] = val;
  }
}

__device__ __forceinline__ int argmin_diag3(const double *a) {
  int m = 0;
  if (a[4] < a[0]) m = 1;
  if (a[8] < a[m * 3 + m]) m = 2;
  return m;
}

// _compute_F21 (:247-283)
__device__ void compute_F21(const float *p1, const float *p2, double *lA, int *lperm, double Fn[9]) {
  for (int i = 0; i < 8; ++i) {
    const float u1 = p1[2 * i], v1 = p1[2 * i + 1], u2 = p2[2 * i], v2 = p2[2 * i + 1];
    RA(i, 0) = (double)(u2 * u1); RA(i, 1) = (double)(u2 * v1); RA(i, 2) = (double)u2;
    RA(i, 3) = (double)(v2 * u1); RA(i, 4) = (double)(v2 * v1); RA(i, 5) = (double)v2;
    RA(i, 6) = (double)u1;        RA(i, 7) = (double)v1;        RA(i, 8) = 1.0;
  }
  double Fpre[9];
  null_vector_8x9(lA, lperm, Fpre);
  double g[9], W[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s = s + Fpre[k * 3 + r] * Fpre[k * 3 + c];
      g[r * 3 + c] = s;
    }
  jacobi_sym3(g, W);
  const int m3 = argmin_diag3(g);
  double vv[3], fv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) vv[k] = (m3 == 0) ? W[k * 3] : ((m3 == 1) ? W[k * 3 + 1] : W[k * 3 + 2]);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    fv[r] = (Fpre[r * 3 + 0] * vv[0] + Fpre[r * 3 + 1] * vv[1]) + Fpre[r * 3 + 2] * vv[2];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Fn[r * 3 + c] = Fpre[r * 3 + c] - fv[r] * vv[c];
}
#undef RA

__device__ void mat3_mul_f(const float *a, const float *b, float *o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      o[i * 3 + j] = (a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j]) + a[i * 3 + 2] * b[2 * 3 + j];
}

__device__ __forceinline__ float wave_sum_strided(const float *x, int stride, int n, int lane) {
  float a = 0.0f;
  for (int i = lane; i < n; i += 64) a = a + x[(size_t)i * stride];
  return bfly64_sum(a);
}

// Canonical order of the correspondences (DESIGN.md "RANSAC"): normalisation sums, the sampler and the score
// sums walk the matches sorted by (x0, y0, x1, y1) (floats through their order-preserving integer images,
// original index last), so the outcome does not depend on the order in which the matcher lists them.
// One workgroup per pair: bitonic sort of <= 1024 (key, index) entries in LDS, sorted copies of the points out.
__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__global__ void __launch_bounds__(1024) ransac_sort_kernel(const int *nmatch, const float *pts0, const float *pts1,
                                                           float *ps0, float *ps1) {
  __shared__ uint32_t k0[RNP], k1[RNP], k2[RNP], k3[RNP];
  __shared__ int id[RNP];
  const int p = blockIdx.x, i = threadIdx.x;
  const int n = nmatch[p];
  if (n < 8) return;
  const float *q0 = pts0 + (size_t)p * RNP * 2, *q1 = pts1 + (size_t)p * RNP * 2;
  if (i < n) {
    k0[i] = order_key(q0[2 * i]); k1[i] = order_key(q0[2 * i + 1]);
    k2[i] = order_key(q1[2 * i]); k3[i] = order_key(q1[2 * i + 1]);
  } else {
    k0[i] = k1[i] = k2[i] = k3[i] = 0xffffffffu;     // padding sorts behind every real entry (index >= n breaks ties)
  }
  id[i] = i;
  __syncthreads();
  for (int k = 2; k <= RNP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int o = i ^ j;
      if (o > i) {
        const bool up = (i & k) == 0;
        bool gt;   // entry i > entry o ?
        if (k0[i] != k0[o]) gt = k0[i] > k0[o];
        else if (k1[i] != k1[o]) gt = k1[i] > k1[o];
        else if (k2[i] != k2[o]) gt = k2[i] > k2[o];
        else if (k3[i] != k3[o]) gt = k3[i] > k3[o];
        else gt = id[i] > id[o];
        if (gt == up) {
          uint32_t t;
          t = k0[i]; k0[i] = k0[o]; k0[o] = t;
          t = k1[i]; k1[i] = k1[o]; k1[o] = t;
          t = k2[i]; k2[i] = k2[o]; k2[o] = t;
          t = k3[i]; k3[i] = k3[o]; k3[o] = t;
          const int ti = id[i]; id[i] = id[o]; id[o] = ti;
        }
      }
      __syncthreads();
    }
  if (i < n) {
    const int src = id[i];
    float *s0 = ps0 + (size_t)p * RNP * 2, *s1 = ps1 + (size_t)p * RNP * 2;
    s0[2 * i] = q0[2 * src]; s0[2 * i + 1] = q0[2 * src + 1];
    s1[2 * i] = q1[2 * src]; s1[2 * i + 1] = q1[2 * src + 1];
  }
}

// _normalize (:735-780); wave 0 = image 1 points, wave 1 = image 2 points
__global__ void __launch_bounds__(128) ransac_normalize_kernel(const int *nmatch, const float *pts0,
                                                               const float *pts1, float *pn0, float *pn1,
                                                               float *T /*[P][2][9]*/) {
  const int p = blockIdx.x, which = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = nmatch[p];
  if (n < 8) return;
  const float *pts = (which ? pts1 : pts0) + (size_t)p * RNP * 2;
  float *pn = (which ? pn1 : pn0) + (size_t)p * RNP * 2;
  const float meanX = wave_sum_strided(pts, 2, n, lane) / (float)n;
  const float meanY = wave_sum_strided(pts + 1, 2, n, lane) / (float)n;
  float ax = 0.0f, ay = 0.0f;
  for (int i = lane; i < n; i += 64) {
    const float dx = pts[2 * i] - meanX, dy = pts[2 * i + 1] - meanY;
    pn[2 * i] = dx; pn[2 * i + 1] = dy;
    ax = ax + fabsf(dx); ay = ay + fabsf(dy);
  }
  const float meanDevX = bfly64_sum(ax) / (float)n, meanDevY = bfly64_sum(ay) / (float)n;
  const float sX = (float)(1.0 / (double)meanDevX), sY = (float)(1.0 / (double)meanDevY);
  for (int i = lane; i < n; i += 64) { pn[2 * i] = pn[2 * i] * sX; pn[2 * i + 1] = pn[2 * i + 1] * sY; }
  if (lane == 0) {
    float *t = T + ((size_t)p * 2 + which) * 9;
    for (int k = 0; k < 9; ++k) t[k] = 0.0f;
    t[0] = sX; t[4] = sY; t[2] = -meanX * sX; t[5] = -meanY * sY; t[8] = 1.0f;
  }
}

// sets: NULL (draw with the counter hash) or [P][iters][8] explicit minimal sets -- the reference's _vSets
// (src/epipolar_geometry.cc:52-71), indices into the match list as the hypothesis kernels see it
__global__ void __launch_bounds__(64) ransac_hyp_kernel(const int *nmatch, const float *pn0, const float *pn1,
                                                        const float *T, uint32_t seed, int iters, const int *sets,
                                                        float *F /*[P][iters][9]*/) {
  __shared__ double lA[72 * 64];
  __shared__ int lperm[9 * 64];
  const int p = blockIdx.y, it = blockIdx.x * 64 + threadIdx.x;
  const int n = nmatch[p];
  if (it >= iters || n < 8) return;
  int set[8];
  if (sets) {
    for (int j = 0; j < 8; ++j) set[j] = sets[((size_t)p * iters + it) * 8 + j];
  } else {
    draw_set(seed, it, n, set);
  }
  float a[16], b[16];
  const float *q0 = pn0 + (size_t)p * RNP * 2, *q1 = pn1 + (size_t)p * RNP * 2;
  for (int j = 0; j < 8; ++j) {
    a[2 * j] = q0[2 * set[j]]; a[2 * j + 1] = q0[2 * set[j] + 1];
    b[2 * j] = q1[2 * set[j]]; b[2 * j + 1] = q1[2 * set[j] + 1];
  }
  double Fn[9];
  compute_F21(a, b, lA + threadIdx.x, lperm + threadIdx.x, Fn);
  float Fnf[9], M[9], T2t[9];
  const float *T1 = T + ((size_t)p * 2) * 9, *T2 = T1 + 9;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T2t[i * 3 + j] = T2[j * 3 + i];
  for (int k = 0; k < 9; ++k) Fnf[k] = (float)Fn[k];
  mat3_mul_f(T2t, Fnf, M);
  float out[9];
  mat3_mul_f(M, T1, out);
  float *fo = F + ((size_t)p * iters + it) * 9;
  for (int k = 0; k < 9; ++k) fo[k] = out[k];
}

// per-match chi-square terms of _check_F (:372-449)
__device__ __forceinline__ bool check_pair(const float *F, float u1, float v1, float u2, float v2,
                                           float invSigmaSquare, float &score) {
  const float th = 3.841f, thScore = 5.991f;
  bool bIn = true;
  const float a2 = (F[0] * u1 + F[1] * v1) + F[2];
  const float b2 = (F[3] * u1 + F[4] * v1) + F[5];
  const float c2 = (F[6] * u1 + F[7] * v1) + F[8];
  const float num2 = (a2 * u2 + b2 * v2) + c2;
  const float squareDist1 = (num2 * num2) / (a2 * a2 + b2 * b2);
  const float chiSquare1 = squareDist1 * invSigmaSquare;
  if (chiSquare1 > th) bIn = false; else score = score + (thScore - chiSquare1);
  const float a1 = (F[0] * u2 + F[3] * v2) + F[6];
  const float b1 = (F[1] * u2 + F[4] * v2) + F[7];
  const float c1 = (F[2] * u2 + F[5] * v2) + F[8];
  const float num1 = (a1 * u1 + b1 * v1) + c1;
  const float squareDist2 = (num1 * num1) / (a1 * a1 + b1 * b1);
  const float chiSquare2 = squareDist2 * invSigmaSquare;
  if (chiSquare2 > th) bIn = false; else score = score + (thScore - chiSquare2);
  return bIn;
}

__global__ void __launch_bounds__(256) ransac_score_kernel(const int *nmatch, const float *pts0, const float *pts1,
                                                           const float *F, float sigma, int iters, float *score,
                                                           int *ninl /*[P][iters] inlier counts, may be NULL*/) {
  const int p = blockIdx.y, it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = nmatch[p];
  if (it >= iters || n < 8) return;
  const float *f = F + ((size_t)p * iters + it) * 9;
  float Fl[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) Fl[k] = f[k];
  const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
  const float *q0 = pts0 + (size_t)p * RNP * 2, *q1 = pts1 + (size_t)p * RNP * 2;
  float sc = 0.0f;
  int cnt = 0;
  for (int i = lane; i < n; i += 64)
    cnt += check_pair(Fl, q0[2 * i], q0[2 * i + 1], q1[2 * i], q1[2 * i + 1], invSigmaSquare, sc) ? 1 : 0;
  sc = bfly64_sum(sc);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
  if (lane == 0) {
    score[(size_t)p * iters + it] = sc;
    if (ninl) ninl[(size_t)p * iters + it] = cnt;
  }
}

// best hypothesis (strict '>' against 0, first wins: :197-201), inlier mask,
// ordered compaction of the match list (src/point_matching.cc:52-58).
// Hypotheses the sequential RANSAC loop of cv::findFundamentalMat(..., confidence) would still evaluate
// (src/point_matching.cc:50; OpenCV's RANSACUpdateNumIters with 8 model points): walking the hypotheses in
// order, every new best shrinks the bound to the smallest k with (1 - w^8)^k <= 1 - confidence, w = its inlier
// ratio; the product is accumulated by sequential f64 multiplications so that CPU and GPU agree bit for bit.
__device__ int ransac_confident_prefix(const float *score, const int *ninl, int n, int iters, double confidence) {
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

__global__ void __launch_bounds__(1024) ransac_select_kernel(const int *nmatch, const float *pts0, const float *pts1,
                                                             const float *F, const float *score, const int *ninl,
                                                             float sigma, double confidence,
                                                             int iters_max, int enable, const DMatchR *matches,
                                                             DMatchR *out, int *nout, uint8_t *inliers, float *Fbest,
                                                             float *best_score) {
  __shared__ float s_best[16];
  __shared__ int s_bi[16];
  __shared__ int wsum[16];
  __shared__ float s_F[9];
  __shared__ int s_it, s_iters;
  const int p = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
  const int n = nmatch[p];
  const DMatchR *mi = matches + (size_t)p * RNP;
  DMatchR *mo = out + (size_t)p * RNP;
  if (!enable || n < 8) {  // no rejection: fewer than 8 matches cannot seed a hypothesis
    if (i < n) { mo[i] = mi[i]; if (inliers) inliers[(size_t)p * RNP + i] = 1; }
    if (i == 0) { nout[p] = n; if (best_score) best_score[p] = 0.0f; }
    if (Fbest && i < 9) Fbest[(size_t)p * 9 + i] = 0.0f;
    return;
  }
  if (i == 0)
    s_iters = (confidence > 0.0 && ninl) ? ransac_confident_prefix(score + (size_t)p * iters_max, ninl + (size_t)p * iters_max, n,
                                                                   iters_max, confidence)
                                         : iters_max;
  __syncthreads();
  const int iters = s_iters;      // hypotheses that count; the arrays keep their stride iters_max
  float best = 0.0f;
  int bi = 0x7fffffff;
  for (int it = i; it < iters; it += 1024) {
    const float s = score[(size_t)p * iters_max + it];
    if (s > best) { best = s; bi = it; }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const float ob = __shfl_xor(best, s, 64);
    const int oi = __shfl_xor(bi, s, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { s_best[wave] = best; s_bi[wave] = bi; }
  __syncthreads();
  if (i == 0) {
    float b = s_best[0];
    int k = s_bi[0];
    for (int w = 1; w < 16; ++w)
      if (s_best[w] > b || (s_best[w] == b && s_bi[w] < k)) { b = s_best[w]; k = s_bi[w]; }
    s_it = (b > 0.0f) ? k : -1;
    if (best_score) best_score[p] = (b > 0.0f) ? b : 0.0f;
  }
  __syncthreads();
  const int bit = s_it;
  if (i < 9) {
    const float v = bit >= 0 ? F[((size_t)p * iters_max + bit) * 9 + i] : 0.0f;
    s_F[i] = v;
    if (Fbest) Fbest[(size_t)p * 9 + i] = v;
  }
  __syncthreads();
  int flag = 0;
  if (i < n && bit >= 0) {
    float Fl[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Fl[k] = s_F[k];
    const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
    const float *q0 = pts0 + (size_t)p * RNP * 2, *q1 = pts1 + (size_t)p * RNP * 2;
    float dummy = 0.0f;
    flag = check_pair(Fl, q0[2 * i], q0[2 * i + 1], q1[2 * i], q1[2 * i + 1], invSigmaSquare, dummy) ? 1 : 0;
  }
  if (inliers && i < n) inliers[(size_t)p * RNP + i] = (uint8_t)flag;
  // ordered compaction
  int incl = flag;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    const int t = wsum[w];
    if (w < wave) base += t;
    tot += t;
  }
  if (flag) mo[base + incl - 1] = mi[i];
  if (i == 0) nout[p] = tot;
}

// ======================================================================
// EpipolarGeometry::reconstruct support (mono initialisation,
// src/epipolar_geometry.cc:18-98): normalisation over ALL keypoints, the
// homography hypotheses (_compute_H21 :207-245) and their scores (_check_H
// :285-370).  The F hypotheses/scores reuse the kernels above.
// ======================================================================
__global__ void __launch_bounds__(128) epi_normalize_kernel(const float *keys1, int n1, const float *keys2, int n2,
                                                            const float *pts0, const float *pts1, int nm, float *pn0,
                                                            float *pn1, float *T /*[2][9]*/) {
  __shared__ float st[2][4];
  const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *keys = which ? keys2 : keys1;
  const int n = which ? n2 : n1;
  const float meanX = wave_sum_strided(keys, 2, n, lane) / (float)n;
  const float meanY = wave_sum_strided(keys + 1, 2, n, lane) / (float)n;
  float ax = 0.0f, ay = 0.0f;
  for (int i = lane; i < n; i += 64) { ax = ax + fabsf(keys[2 * i] - meanX); ay = ay + fabsf(keys[2 * i + 1] - meanY); }
  const float dX = bfly64_sum(ax) / (float)n, dY = bfly64_sum(ay) / (float)n;
  const float sX = (float)(1.0 / (double)dX), sY = (float)(1.0 / (double)dY);
  if (lane == 0) {
    float *t = T + which * 9;
    for (int k = 0; k < 9; ++k) t[k] = 0.0f;
    t[0] = sX; t[4] = sY; t[2] = -meanX * sX; t[5] = -meanY * sY; t[8] = 1.0f;
    st[which][0] = meanX; st[which][1] = meanY; st[which][2] = sX; st[which][3] = sY;
  }
  __syncthreads();
  const float *pts = which ? pts1 : pts0;
  float *pn = which ? pn1 : pn0;
  for (int i = lane; i < nm; i += 64) {
    pn[2 * i] = (pts[2 * i] - st[which][0]) * st[which][2];
    pn[2 * i + 1] = (pts[2 * i + 1] - st[which][1]) * st[which][3];
  }
}

__device__ void mat3_inv_f(const float *m, float *o) {  // cofactor inverse, float
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = (a * A + b * B) + c * C;
  const float id = 1.0f / det;
  o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
  o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
  o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

// cyclic Jacobi on a symmetric 9x9 held in LDS: a(i,j) = la[(i*9+j)*32 + lane32]
#define JA(i, j) la[((i) * 9 + (j)) * 32]
#define JV(i, j) lv[((i) * 9 + (j)) * 32]
__device__ void jacobi9_lds(double *la, double *lv) {
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) JV(i, j) = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < JAC_SWEEPS; ++sweep) {
    for (int p = 0; p < 8; ++p)
      for (int q = p + 1; q < 9; ++q) {
        const double apq = JA(p, q);
        if (fabs(apq) < 1e-300) continue;
        const double theta = (JA(q, q) - JA(p, p)) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 9; ++k) {
          const double akp = JA(k, p), akq = JA(k, q);
          JA(k, p) = c * akp - s * akq;
          JA(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < 9; ++k) {
          const double apk = JA(p, k), aqk = JA(q, k);
          JA(p, k) = c * apk - s * aqk;
          JA(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < 9; ++k) {
          const double vkp = JV(k, p), vkq = JV(k, q);
          JV(k, p) = c * vkp - s * vkq;
          JV(k, q) = s * vkp + c * vkq;
        }
      }
  }
}

__global__ void __launch_bounds__(32) epi_hyp_h_kernel(const float *pn0, const float *pn1, int nm, const float *T,
                                                       uint32_t seed, int iters, const int *sets,
                                                       float *H /*[iters][18]: H21 | H12*/) {
  __shared__ double la[81 * 32];
  __shared__ double lv[81 * 32];
  const int it = blockIdx.x * 32 + threadIdx.x;
  if (it >= iters || nm < 8) return;
  double *a = la + threadIdx.x, *v = lv + threadIdx.x;
  int set[8];
  if (sets) {
    for (int j = 0; j < 8; ++j) set[j] = sets[(size_t)it * 8 + j];
  } else {
    draw_set(seed, it, nm, set);
  }
  // A^T A of the 16x9 DLT system, accumulated row pair by row pair in source order
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) a[(r * 9 + c) * 32] = 0.0;
  double rows[16][9];
  for (int i = 0; i < 8; ++i) {
    const float u1 = pn0[2 * set[i]], v1 = pn0[2 * set[i] + 1], u2 = pn1[2 * set[i]], v2 = pn1[2 * set[i] + 1];
    double *r0 = rows[2 * i], *r1 = rows[2 * i + 1];
    r0[0] = 0.0; r0[1] = 0.0; r0[2] = 0.0; r0[3] = (double)(-u1); r0[4] = (double)(-v1); r0[5] = -1.0;
    r0[6] = (double)(v2 * u1); r0[7] = (double)(v2 * v1); r0[8] = (double)v2;
    r1[0] = (double)u1; r1[1] = (double)v1; r1[2] = 1.0; r1[3] = 0.0; r1[4] = 0.0; r1[5] = 0.0;
    r1[6] = (double)(-u2 * u1); r1[7] = (double)(-u2 * v1); r1[8] = (double)(-u2);
  }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) {
      double s = 0.0;
      for (int i = 0; i < 16; ++i) s = s + rows[i][r] * rows[i][c];
      a[(r * 9 + c) * 32] = s;
    }
  jacobi9_lds(a, v);
  int m = 0;
  for (int i = 1; i < 9; ++i)
    if (a[(i * 9 + i) * 32] < a[(m * 9 + m) * 32]) m = i;
  float Hn[9], M[9], T2inv[9], H21[9], H12[9];
  for (int k = 0; k < 9; ++k) Hn[k] = (float)v[(k * 9 + m) * 32];
  mat3_inv_f(T + 9, T2inv);
  mat3_mul_f(T2inv, Hn, M);
  mat3_mul_f(M, T, H21);
  mat3_inv_f(H21, H12);
  float *ho = H + (size_t)it * 18;
  for (int k = 0; k < 9; ++k) { ho[k] = H21[k]; ho[9 + k] = H12[k]; }
}
#undef JA
#undef JV

__device__ __forceinline__ bool check_pair_h(const float *H21, const float *H12, float u1, float v1, float u2, float v2,
                                             float invSigmaSquare, float &score) {
  const float th = 5.991f;
  bool bIn = true;
  const float w2in1inv = (float)(1.0 / (double)((H12[6] * u2 + H12[7] * v2) + H12[8]));
  const float u2in1 = ((H12[0] * u2 + H12[1] * v2) + H12[2]) * w2in1inv;
  const float v2in1 = ((H12[3] * u2 + H12[4] * v2) + H12[5]) * w2in1inv;
  const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
  const float chiSquare1 = squareDist1 * invSigmaSquare;
  if (chiSquare1 > th) bIn = false; else score = score + (th - chiSquare1);
  const float w1in2inv = (float)(1.0 / (double)((H21[6] * u1 + H21[7] * v1) + H21[8]));
  const float u1in2 = ((H21[0] * u1 + H21[1] * v1) + H21[2]) * w1in2inv;
  const float v1in2 = ((H21[3] * u1 + H21[4] * v1) + H21[5]) * w1in2inv;
  const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
  const float chiSquare2 = squareDist2 * invSigmaSquare;
  if (chiSquare2 > th) bIn = false; else score = score + (th - chiSquare2);
  return bIn;
}

__global__ void __launch_bounds__(256) epi_score_h_kernel(const float *pts0, const float *pts1, int nm, const float *H,
                                                          float sigma, int iters, float *score) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= iters || nm < 8) return;
  float Hl[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) Hl[k] = H[(size_t)it * 18 + k];
  const float invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));
  float sc = 0.0f;
  for (int i = lane; i < nm; i += 64)
    check_pair_h(Hl, Hl + 9, pts0[2 * i], pts0[2 * i + 1], pts1[2 * i], pts1[2 * i + 1], invSigmaSquare, sc);
  sc = bfly64_sum(sc);
  if (lane == 0) score[it] = sc;
}

// both model searches for one image pair; everything on `st`
int launch_epipolar_search(const float *keys1, int n1, const float *keys2, int n2, const float *pts0, const float *pts1,
                           const int *d_nm, int nm, float *pn0, float *pn1, float *T, float *F, float *scoreF,
                           float *H, float *scoreH, uint32_t seed, int iters, float sigma, const int *d_sets,
                           hipStream_t st) {
  hipLaunchKernelGGL(epi_normalize_kernel, dim3(1), dim3(128), 0, st, keys1, n1, keys2, n2, pts0, pts1, nm, pn0, pn1, T);
  hipLaunchKernelGGL(ransac_hyp_kernel, dim3((iters + 63) / 64, 1), dim3(64), 0, st, d_nm, pn0, pn1, T, seed, iters, d_sets, F);
  hipLaunchKernelGGL(ransac_score_kernel, dim3((iters + 3) / 4, 1), dim3(256), 0, st, d_nm, pts0, pts1, F, sigma, iters,
                     scoreF, (int *)nullptr);
  hipLaunchKernelGGL(epi_hyp_h_kernel, dim3((iters + 31) / 32), dim3(32), 0, st, pn0, pn1, nm, T, seed, iters, d_sets, H);
  hipLaunchKernelGGL(epi_score_h_kernel, dim3((iters + 3) / 4), dim3(256), 0, st, pts0, pts1, nm, H, sigma, iters,
                     scoreH);
  URF_HIP(hipGetLastError());
  return 0;
}

// d_sets: NULL, or explicit minimal sets [P][iters][8] that index the matches in the CALLER's order (then the
// canonical sort is skipped: sampler, sums and scores walk the list as given, like the reference)
int launch_ransac(const int *nmatch, const float *pts0, const float *pts1, float *ps0, float *ps1, float *pn0,
                  float *pn1, float *T, float *F, float *score, int *ninl, uint32_t seed, int iters, float sigma,
                  double confidence, const int *d_sets, int enable, const void *matches, void *out, int *nout,
                  uint8_t *inliers, float *Fbest, float *best_score, int P, hipStream_t st) {
  if (enable) {
    // ps0 / ps1: the correspondences in canonical order; the final per-point inlier test runs on the caller's order
    const float *w0 = d_sets ? pts0 : ps0, *w1 = d_sets ? pts1 : ps1;
    if (!d_sets) hipLaunchKernelGGL(ransac_sort_kernel, dim3(P), dim3(1024), 0, st, nmatch, pts0, pts1, ps0, ps1);
    hipLaunchKernelGGL(ransac_normalize_kernel, dim3(P), dim3(128), 0, st, nmatch, w0, w1, pn0, pn1, T);
    hipLaunchKernelGGL(ransac_hyp_kernel, dim3((iters + 63) / 64, P), dim3(64), 0, st, nmatch, pn0, pn1, T, seed, iters,
                       d_sets, F);
    hipLaunchKernelGGL(ransac_score_kernel, dim3((iters + 3) / 4, P), dim3(256), 0, st, nmatch, w0, w1, F, sigma,
                       iters, score, ninl);
  }
  hipLaunchKernelGGL(ransac_select_kernel, dim3(P), dim3(1024), 0, st, nmatch, pts0, pts1, F, score, ninl, sigma,
                     confidence, iters, enable, (const DMatchR *)matches, (DMatchR *)out, nout, inliers, Fbest, best_score);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf
