// urf_math.h -- canonical fp32 arithmetic of the front-end (device side).
//
// Written specification: DESIGN.md "Canonical arithmetic".  Every function is
// built from correctly rounded IEEE-754 +,-,*,/,sqrt and fma only (the library
// is compiled with -ffp-contract=off and without fast-math), so results do not
// depend on the GPU's transcendental units and are reproducible bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace urf {

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exp_c: Cephes-style expf. x<-87.33654 -> 0, clamp at 88.
__device__ __forceinline__ float exp_c(float x) {
  if (x < -87.33654f) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = fma_rn(n, -0.693359375f, x);
  r = fma_rn(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fma_rn(p, r, 1.3981999507e-3f);
  p = fma_rn(p, r, 8.3334519073e-3f);
  p = fma_rn(p, r, 4.1665795894e-2f);
  p = fma_rn(p, r, 1.6666665459e-1f);
  p = fma_rn(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = fma_rn(p, r2, r) + 1.0f;
  int ni = (int)n;
  int n1 = ni / 2, n2 = ni - n1;
  float s1 = __uint_as_float((uint32_t)(n1 + 127) << 23);
  float s2 = __uint_as_float((uint32_t)(n2 + 127) << 23);
  return (y * s1) * s2;
}

// log_c: Cephes-style logf for positive normal x.
__device__ __forceinline__ float log_c(float x) {
  uint32_t u = __float_as_uint(x);
  int e = (int)((u >> 23) & 0xff) - 126;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
  else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fma_rn(p, m, -1.1514610310e-1f);
  p = fma_rn(p, m, 1.1676998740e-1f);
  p = fma_rn(p, m, -1.2420140846e-1f);
  p = fma_rn(p, m, 1.4249322787e-1f);
  p = fma_rn(p, m, -1.6668057665e-1f);
  p = fma_rn(p, m, 2.0000714765e-1f);
  p = fma_rn(p, m, -2.4999993993e-1f);
  p = fma_rn(p, m, 3.3333331174e-1f);
  float y = (m * z) * p;
  float fe = (float)e;
  y = fma_rn(fe, -2.12194440e-4f, y);
  y = fma_rn(z, -0.5f, y);
  float r = m + y;
  r = fma_rn(fe, 0.693359375f, r);
  return r;
}

// canonical 64-lane butterfly: v <- v + v[lane^s], s = 32,16,...,1
__device__ __forceinline__ float bfly64_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}
__device__ __forceinline__ float bfly64_max(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
  return v;
}

}  // namespace urf
