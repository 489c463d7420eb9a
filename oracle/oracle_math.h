/* oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Canonical fp32 arithmetic used by the CPU oracle.  The HIP product path
 * implements the SAME written specification (DESIGN.md "Canonical arithmetic")
 * in its own source (ur-mvo_amd/csrc/urf_math.h); nothing in the product
 * includes or links this file.  Every function here is built only from
 * IEEE-754 correctly rounded +,-,*,/,sqrt,fma and integer bit operations, so a
 * CPU build (-ffp-contract=off) and a gfx950 build evaluate bit-identical
 * results.
 *
 * exp_c / log_c follow the classic Cephes single-precision kernels
 * (range reduction + degree-5 / degree-8 polynomials); they are pinned against
 * libm in tests/test_oracle_math.py (<= 2 ulp).
 */
#ifndef URF_ORACLE_MATH_H_
#define URF_ORACLE_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float om_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

static inline float om_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t om_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp_c(x): x <= ~88.  Returns 0 below -87.33654. */
static inline float om_exp(float x) {
  if (x < -87.33654f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = om_fma(n, -0.693359375f, x);
  r = om_fma(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = om_fma(p, r, 1.3981999507e-3f);
  p = om_fma(p, r, 8.3334519073e-3f);
  p = om_fma(p, r, 4.1665795894e-2f);
  p = om_fma(p, r, 1.6666665459e-1f);
  p = om_fma(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = om_fma(p, r2, r) + 1.0f;
  int32_t ni = (int32_t)n;
  /* scale by 2^n in two exact steps so n in [-126,127] never overflows the
     exponent field of the scale factor */
  int32_t n1 = ni / 2, n2 = ni - n1;
  float s1 = om_bits2f((uint32_t)(n1 + 127) << 23);
  float s2 = om_bits2f((uint32_t)(n2 + 127) << 23);
  return (y * s1) * s2;
}

/* log_c(x): x positive, finite, normal. */
static inline float om_log(float x) {
  uint32_t u = om_f2bits(x);
  int32_t e = (int32_t)((u >> 23) & 0xff) - 126;           /* x = m * 2^e, m in [0.5,1) */
  float m = om_bits2f((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
  else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = om_fma(p, m, -1.1514610310e-1f);
  p = om_fma(p, m, 1.1676998740e-1f);
  p = om_fma(p, m, -1.2420140846e-1f);
  p = om_fma(p, m, 1.4249322787e-1f);
  p = om_fma(p, m, -1.6668057665e-1f);
  p = om_fma(p, m, 2.0000714765e-1f);
  p = om_fma(p, m, -2.4999993993e-1f);
  p = om_fma(p, m, 3.3333331174e-1f);
  float y = (m * z) * p;
  float fe = (float)e;
  y = om_fma(fe, -2.12194440e-4f, y);
  y = om_fma(z, -0.5f, y);
  float r = m + y;
  r = om_fma(fe, 0.693359375f, r);
  return r;
}

/* Canonical 64-lane butterfly sum: p[l] <- p[l] + p[l^s], s = 32,16,..,1.
   All 64 slots end up holding the same value; returns it. */
static inline float om_bfly64_sum(float p[64]) {
  float q[64];
  for (int s = 32; s >= 1; s >>= 1) {
    for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ s];
    for (int l = 0; l < 64; ++l) p[l] = q[l];
  }
  return p[0];
}

/* Canonical "wave-strided" sum of n floats: lane l accumulates x[l], x[l+64],
   ... in ascending order starting from +0, then the butterfly. */
static inline float om_wave_sum(const float *x, int n) {
  float p[64];
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int j = l; j < n; j += 64) a = a + x[j];
    p[l] = a;
  }
  return om_bfly64_sum(p);
}

/* Canonical "wave-strided-by-4" sum (Sinkhorn LSE): lane l accumulates the
   elements 256t + 4l + r (t ascending, r = 0..3) -- the order of 16-byte loads
   per lane -- starting from +0, then the butterfly. */
static inline float om_wave_sum4(const float *x, int n) {
  float p[64];
  for (int l = 0; l < 64; ++l) {
    float a = 0.0f;
    for (int t = 0; 256 * t + 4 * l < n; ++t)
      for (int r = 0; r < 4; ++r) {
        const int j = 256 * t + 4 * l + r;
        if (j < n) a = a + x[j];
      }
    p[l] = a;
  }
  return om_bfly64_sum(p);
}

#endif
