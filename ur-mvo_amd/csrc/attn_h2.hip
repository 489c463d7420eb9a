// attn_h2.hip -- "fast" multi-head attention on the f16 matrix core with split
// operands (see h2gemm.hip): S = Q K^T, softmax, O = P V with every product
// evaluated as hi*hi + hi*lo + lo*hi in fp32 accumulators.  Same workgroup shape
// as the exact attn_kernel: 512 threads = 4 query tiles x 2 key halves, the 16 x
// 512 score block of a wave lives in 128 VGPRs, the S^T accumulators are fed back
// as the B operand of O^T = V^T P^T.  V arrives transposed ([d][token], written
// by h2gemm's transposed epilogue) so that 4 consecutive keys of one d are an
// 8-byte LDS read.  Opt-in precision mode, validated against the exact kernel.
#include "h2.h"

#include <float.h>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int ANP = kCap;
constexpr int AS = 72;       // LDS row stride (halfs)
constexpr int AHALF = 512;   // keys per half

__global__ void __launch_bounds__(512, 2) attn_h2_kernel(const _Float16 *qkh, const _Float16 *qkl,
                                                         const _Float16 *vth, const _Float16 *vtl, const int *counts,
                                                         int cross, _Float16 *oh, _Float16 *ol) {
  // [buffer][half][plane][64 rows][AS]
  __shared__ __attribute__((aligned(16))) _Float16 kv[2][2][2][64 * AS];
  __shared__ float s_max[8][16];
  __shared__ float s_l[4][16];
  const int im = blockIdx.z, sm = cross ? (im ^ 1) : im;
  const int head = blockIdx.y;
  const int nq = counts[im], ns = counts[sm];
  const int q0 = blockIdx.x * 64;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = wave & 3, kh = wave >> 2;
  const int px = lane & 15, g = lane >> 4;
  const int nsA = ns < AHALF ? ns : AHALF;
  const int nrounds = (nsA + 63) >> 6;
  const int kbase_h = kh * AHALF;

  // staging: thread -> (half sh, plane sp, row sr (+32u), 16-byte piece sj)
  const int sh = tid >> 8, sp = (tid >> 7) & 1, st = tid & 127;
  const int sj = st & 7, sr = st >> 3;   // 16 rows per pass, 4 passes
  f16x8 pf[4];
  auto issue_k = [&](int ch) {
    const _Float16 *base = (sp ? qkl : qkh) + ((size_t)sm * ANP) * 512 + 256 + head * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = sh * AHALF + ch * 64 + sr + 16 * u;
      pf[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (key < ns) pf[u] = *(const f16x8 *)(base + (size_t)key * 512 + 8 * sj);
    }
  };
  auto issue_v = [&](int ch) {  // V^T rows = d, columns = keys
    const _Float16 *base = (sp ? vtl : vth) + ((size_t)sm * 256 + head * 64) * ANP;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = sr + 16 * u;
      const int key0 = sh * AHALF + ch * 64 + 8 * sj;
      f16x8 v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (key0 < ns) {
        v = *(const f16x8 *)(base + (size_t)d * ANP + key0);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (key0 + e >= ns) v[e] = (_Float16)0.0f;   // stale tokens beyond the count
      }
      pf[u] = v;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *(f16x8 *)(kv[buf][sh][sp] + (sr + 16 * u) * AS + 8 * sj) = pf[u];
  };

  issue_k(0);
  // Q fragments (B operand): Q[q][8g + j + 32ks]
  f16x8 qh[2], ql[2];
  {
    const size_t qo = ((size_t)im * ANP + q0 + qt * 16 + px) * 512 + head * 64 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qh[ks] = *(const f16x8 *)(qkh + qo + 32 * ks);
      ql[ks] = *(const f16x8 *)(qkl + qo + 32 * ks);
    }
  }
  commit(0);
  __syncthreads();

  f32x4 sreg[32];
  // ---------------- phase 1: S^T = K Q^T
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    if (ch < nrounds) {
      if (ch + 1 < nrounds) issue_k(ch + 1);
      const _Float16 *kph = kv[ch & 1][kh][0] + px * AS + 8 * g;
      const _Float16 *kpl = kv[ch & 1][kh][1] + px * AS + 8 * g;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 ah = *(const f16x8 *)(kph + kt * 16 * AS + 32 * ks);
          const f16x8 al = *(const f16x8 *)(kpl + kt * 16 * AS + 32 * ks);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[ks], acc, 0, 0, 0);
        }
        const int kb0 = kbase_h + ch * 64 + kt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (kb0 + r < ns) ? acc[r] * 0.125f : -FLT_MAX;
        sreg[ch * 4 + kt] = acc;
      }
      if (ch + 1 < nrounds) commit((ch + 1) & 1);
      __syncthreads();
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) sreg[ch * 4 + kt] = f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    }
  }
  issue_v(0);
  float m = -FLT_MAX;
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, sreg[t][r]);
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (g == 0) s_max[wave][px] = m;
  commit(0);
  __syncthreads();
  m = fmaxf(m, s_max[wave ^ 4][px]);
  // ---------------- phase 2: P = exp(S - m), O^T += V^T P^T  (32 keys per k-step)
  const int nsl = ns - kbase_h;
  float part = 0.0f;
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    if (ch < nrounds) {
      if (ch + 1 < nrounds) issue_v(ch + 1);
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {           // pairs of 16-key tiles
        const int T = ch * 4 + 2 * kp;
        if (T * 16 < nsl) {
          f16x8 ph, pl;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float s = sreg[T + (e >> 2)][e & 3];
            const float p = (s == -FLT_MAX) ? 0.0f : __expf(s - m);
            part = part + p;
            const _Float16 h = (_Float16)p;
            ph[e] = h;
            pl[e] = (_Float16)(p - (float)h);
          }
          const _Float16 *vph = kv[ch & 1][kh][0] + px * AS + (2 * kp) * 16 + 4 * g;
          const _Float16 *vpl = kv[ch & 1][kh][1] + px * AS + (2 * kp) * 16 + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            f16x8 ah, al;
            const f16x4 a0 = *(const f16x4 *)(vph + dt * 16 * AS), a1 = *(const f16x4 *)(vph + dt * 16 * AS + 16);
            const f16x4 b0 = *(const f16x4 *)(vpl + dt * 16 * AS), b1 = *(const f16x4 *)(vpl + dt * 16 * AS + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ah[e] = a0[e]; ah[4 + e] = a1[e]; al[e] = b0[e]; al[4 + e] = b1[e]; }
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ph, oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pl, oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ph, oacc[dt], 0, 0, 0);
          }
        }
      }
      if (ch + 1 < nrounds) commit((ch + 1) & 1);
      __syncthreads();
    }
  }
  float l = part + __shfl_xor(part, 16, 64);
  l = l + __shfl_xor(l, 32, 64);
  // ---------------- combine halves
  float *xo = (float *)&kv[0][0][0][0] + qt * (64 * 20);
  if (kh == 1) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4 *)(xo + lane * 20 + 4 * dt) = oacc[dt];
    if (g == 0) s_l[qt][px] = l;
  }
  __syncthreads();
  if (kh == 0) {
    const float lt = l + s_l[qt][px];
    const int q = q0 + qt * 16 + px;
    const size_t oo = ((size_t)im * ANP + q) * 256 + head * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 ob = *(const f32x4 *)(xo + lane * 20 + 4 * dt);
      f16x4 h, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (oacc[dt][r] + ob[r]) / lt;
        h[r] = (_Float16)v;
        lo[r] = (_Float16)(v - (float)h[r]);
      }
      *(f16x4 *)(oh + oo + dt * 16) = h;
      *(f16x4 *)(ol + oo + dt * 16) = lo;
    }
  }
}

int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(attn_h2_kernel, dim3(ANP / 64, 4, nimg), dim3(512), 0, st, qkh, qkl, vth, vtl, counts, cross, oh, ol);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf
