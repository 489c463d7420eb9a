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
constexpr int AS = 80;       // K rows: 160 B, conflict-free ds_read_b128
constexpr int VS = 72;       // V^T rows: 144 B, conflict-free ds_read_b64

// One pass over the keys with an online softmax (running max / running sum,
// flash-attention style): per 64-key chunk a wave holds only a 16 x 64 score block
// (16 VGPRs), so four waves per SIMD stay resident and hide the K/V load latency
// that a two-pass kernel exposes once the MFMAs are this fast.
// Workgroup = 512 threads = 8 query tiles (128 queries) x 1 head; K chunk
// [64 keys][64 d] and V^T chunk [64 d][64 keys] (hi/lo planes) double-buffered in
// LDS and shared by the 8 waves.
__global__ void __launch_bounds__(512, 4) attn_h2_kernel(const _Float16 *qkh, const _Float16 *qkl,
                                                         const _Float16 *vth, const _Float16 *vtl, const int *counts,
                                                         int cross, _Float16 *oh, _Float16 *ol) {
  // [buffer][K planes | V planes]
  __shared__ __attribute__((aligned(16))) _Float16 kbuf[2][2][64 * AS];
  __shared__ __attribute__((aligned(16))) _Float16 vbuf[2][2][64 * VS];
  int qb, grp;
  xcd_group_map(blockIdx.x, ANP / 128, (int)gridDim.x / (ANP / 128), qb, grp);   // the 8 query tiles of a head on one XCD
  const int im = grp >> 2, sm = cross ? (im ^ 1) : im;
  const int head = grp & 3;
  const int nq = counts[im], ns = counts[sm];
  const int q0 = qb * 128;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int nchunk = (ns + 63) >> 6;

  // staging: thread -> (tensor sk: 0 = K, 1 = V^T; plane sp; row sr (+32u); 16-byte piece sj)
  const int sk = tid >> 8, sp = (tid >> 7) & 1, st = tid & 127;
  const int sj = st & 7, sr = st >> 3;
  f16x8 pf[4];
  auto issue = [&](int ch) {
    if (sk == 0) {
      const _Float16 *base = (sp ? qkl : qkh) + ((size_t)sm * ANP) * 512 + 256 + head * 64;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = ch * 64 + sr + 16 * u;
        pf[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (key < ns) pf[u] = *(const f16x8 *)(base + (size_t)key * 512 + 8 * sj);
      }
    } else {
      const _Float16 *base = (sp ? vtl : vth) + ((size_t)sm * 256 + head * 64) * ANP;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = sr + 16 * u;
        const int key0 = ch * 64 + 8 * sj;
        f16x8 v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (key0 < ns) {
          v = *(const f16x8 *)(base + (size_t)d * ANP + key0);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (key0 + e >= ns) v[e] = (_Float16)0.0f;   // stale tokens beyond the count
        }
        pf[u] = v;
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (sk == 0) *(f16x8 *)(kbuf[buf][sp] + (sr + 16 * u) * AS + 8 * sj) = pf[u];
      else *(f16x8 *)(vbuf[buf][sp] + (sr + 16 * u) * VS + 8 * sj) = pf[u];
    }
  };

  issue(0);
  f16x8 qh[2], ql[2];
  {
    const size_t qo = ((size_t)im * ANP + q0 + wave * 16 + px) * 512 + head * 64 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qh[ks] = *(const f16x8 *)(qkh + qo + 32 * ks);
      ql[ks] = *(const f16x8 *)(qkl + qo + 32 * ks);
    }
  }
  commit(0);
  __syncthreads();

  float m = -FLT_MAX, part = 0.0f;
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) issue(ch + 1);
    // ---- S^T = K Q^T for the 64 keys of this chunk
    f32x4 s[4];
    const _Float16 *kph = kbuf[buf][0] + px * AS + 8 * g;
    const _Float16 *kpl = kbuf[buf][1] + px * AS + 8 * g;
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
      const int kb0 = ch * 64 + kt * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (kb0 + r < ns) ? acc[r] * 0.125f : -FLT_MAX;
      s[kt] = acc;
    }
    // ---- online softmax update (per query = per px; the 4 lanes g share it)
    float cm = -FLT_MAX;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cm = fmaxf(cm, s[kt][r]);
    cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
    cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    const float mn = fmaxf(m, cm);
    const float alpha = __expf(m - mn);
    m = mn;
    part = part * alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[dt][r] = oacc[dt][r] * alpha;
    // ---- O^T += V^T P^T, 32 keys per k-step
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      f16x8 ph, pl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sv = s[2 * kp + (e >> 2)][e & 3];
        const float p = (sv == -FLT_MAX) ? 0.0f : __expf(sv - mn);
        part = part + p;
        const _Float16 h = (_Float16)p;
        ph[e] = h;
        pl[e] = (_Float16)(p - (float)h);
      }
      const _Float16 *vph = vbuf[buf][0] + px * VS + (2 * kp) * 16 + 4 * g;
      const _Float16 *vpl = vbuf[buf][1] + px * VS + (2 * kp) * 16 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f16x8 ah, al;
        const f16x4 a0 = *(const f16x4 *)(vph + dt * 16 * VS), a1 = *(const f16x4 *)(vph + dt * 16 * VS + 16);
        const f16x4 b0 = *(const f16x4 *)(vpl + dt * 16 * VS), b1 = *(const f16x4 *)(vpl + dt * 16 * VS + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ah[e] = a0[e]; ah[4 + e] = a1[e]; al[e] = b0[e]; al[4 + e] = b1[e]; }
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ph, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pl, oacc[dt], 0, 0, 0);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ph, oacc[dt], 0, 0, 0);
      }
    }
    if (ch + 1 < nchunk) commit(buf ^ 1);
    __syncthreads();
  }
  float l = part + __shfl_xor(part, 16, 64);
  l = l + __shfl_xor(l, 32, 64);
  const int q = q0 + wave * 16 + px;
  const size_t oo = ((size_t)im * ANP + q) * 256 + head * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f16x4 h, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = oacc[dt][r] / l;
      h[r] = (_Float16)v;
      lo[r] = (_Float16)(v - (float)h[r]);
    }
    *(f16x4 *)(oh + oo + dt * 16) = h;
    *(f16x4 *)(ol + oo + dt * 16) = lo;
  }
}

int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  hipLaunchKernelGGL(attn_h2_kernel, dim3((ANP / 128) * 4 * nimg), dim3(512), 0, st, qkh, qkl, vth, vtl, counts, cross, oh, ol);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf
