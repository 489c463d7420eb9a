// sg_kernels.hip -- SuperGlue device kernels: input prep, multi-head attention,
// score matrix, log-domain Sinkhorn, decode and match assembly.
// Replaces the SuperGlue TensorRT engine (src/super_glue.cpp:227) and the host
// decode / match assembly (src/super_glue.cpp:303-430, src/point_matching.cc:
// 26-45).  The linear layers run on conv_mfma.hip (TAPS==1).
//
// All matrix products are exact-fp32 v_mfma_f32_16x16x4_f32 fma chains in the
// canonical order of DESIGN.md; exp/log are the canonical polynomials.
#include <cstdlib>
#include "urf_common.h"
#include "urf_math.h"

#include <float.h>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NP = kCap;          // rows reserved per image
constexpr int LDC = 1028;         // leading dimension of the couplings matrices (>= NP+1, multiple of 4)

// ------------------------------------------------------------------- prep
// From feature slots: counts, kenc input [x_n, y_n, score, 0] and descriptors.
// NormalizeKeypoints (src/point_matching.cc:63-76) in double, narrowed like
// SuperGlue::process_input (src/super_glue.cpp:259-275).
__global__ void __launch_bounds__(256) sg_prep_slots_kernel(const float *const *slots, int width, int height,
                                                            int *counts, float *kin /*[img][NP][4]*/,
                                                            float *kxy /*[img][NP][2]*/,
                                                            float *x /*[img][NP][256]*/) {
  const int im = blockIdx.y;
  const float *sl = slots[im];
  const int n = ((const int *)sl)[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[im] = n;
  const int mx = width > height ? width : height;
  for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < NP; j += gridDim.x * 4) {
    const int lane = threadIdx.x & 63;
    f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
    if (j < n) d = *(const f32x4 *)(sl + kSlotHeader + 4 * (size_t)kCap + (size_t)j * 256 + 4 * lane);
    *(f32x4 *)(x + ((size_t)im * NP + j) * 256 + 4 * lane) = d;
    if (lane == 0) {
      f32x4 k = {0.0f, 0.0f, 0.0f, 0.0f};
      float2 xy = make_float2(0.0f, 0.0f);
      if (j < n) {
        const f32x4 m = *(const f32x4 *)(sl + kSlotHeader + 4 * (size_t)j);
        k[0] = (float)(((double)m[1] - width / 2) / (mx * 0.7));
        k[1] = (float)(((double)m[2] - height / 2) / (mx * 0.7));
        k[2] = m[0];
        xy = make_float2(m[1], m[2]);
      }
      *(f32x4 *)(kin + ((size_t)im * NP + j) * 4) = k;
      *(float2 *)(kxy + ((size_t)im * NP + j) * 2) = xy;
    }
  }
}

// -------------------------------------------------------------- attention
// One workgroup = 64 queries x 1 head of one image; wave w = 16 queries.
// Phase 1: S^T = K Q^T on MFMA (M = key, N = query), the whole 16 x ns score
//   block of the wave stays in registers (64 x f32x4).
// softmax: row max (4-lane butterfly), p = exp_c(s*0.125 - m), row sum in the
//   canonical order P_g = seq_{t,r} p[16t+4g+r], l = (P0+P1)+(P2+P3).
// Phase 2: O^T = V^T P^T on MFMA with the S^T accumulators used directly as the
//   B operand (k-slot g <-> key 16t+4g+r), so the P.V chain visits the keys of
//   each 16-block in the order 0,4,8,12,1,5,... (canonical).
constexpr int KSTR = 66, VSTR = 68;

// exp_c restricted to x <= 0 (softmax arguments): identical results to exp_c,
// the upper clamp is dead and the 2^n scaling is one exact v_ldexp_f32.
__device__ __forceinline__ float exp_c_nonpos(float x) {
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = fma_rn(n, -0.693359375f, x);
  r = fma_rn(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fma_rn(p, r, 1.3981999507e-3f);
  p = fma_rn(p, r, 8.3334519073e-3f);
  p = fma_rn(p, r, 4.1665795894e-2f);
  p = fma_rn(p, r, 1.6666665459e-1f);
  p = fma_rn(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fma_rn(p, r2, r) + 1.0f;
  const float v = __builtin_ldexpf(y, (int)n);
  return x < -87.33654f ? 0.0f : v;
}

// Two softmax arguments at a time on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: one instruction, two IEEE
// operations per lane): the same operations on each element as exp_c_nonpos, hence the same bits -- 19 instructions per pair of
// elements instead of 32.  The exact attention kernel spends 40 % of its time in this polynomial (fp32 VALU work does not hide under
// the fp32 MFMAs on gfx950: EXPERIMENTS.md section 8, round 5).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_c_nonpos2(f32x2 x) {
  auto fma2 = [](f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); };
  auto k2 = [](float c) { return f32x2{c, c}; };
  const f32x2 n = __builtin_elementwise_rint(x * 1.44269504088896341f);
  f32x2 r = fma2(n, k2(-0.693359375f), x);
  r = fma2(n, k2(2.12194440e-4f), r);
  f32x2 p = k2(1.9875691500e-4f);
  p = fma2(p, r, k2(1.3981999507e-3f));
  p = fma2(p, r, k2(8.3334519073e-3f));
  p = fma2(p, r, k2(4.1665795894e-2f));
  p = fma2(p, r, k2(1.6666665459e-1f));
  p = fma2(p, r, k2(5.0000001201e-1f));
  const f32x2 r2 = r * r;
  const f32x2 y = fma2(p, r2, r) + 1.0f;
  f32x2 v;
  v[0] = __builtin_ldexpf(y[0], (int)n[0]);
  v[1] = __builtin_ldexpf(y[1], (int)n[1]);
  v[0] = x[0] < -87.33654f ? 0.0f : v[0];
  v[1] = x[1] < -87.33654f ? 0.0f : v[1];
  return v;
}

// 512-thread workgroup = 64 queries x 1 head: wave w handles query tile (w & 3)
// against key half (w >> 2): keys [0,512) or [512,1024).  The two waves of a
// query tile share a SIMD (2 waves/SIMD hide each other's LDS/barrier stalls);
// their partial results are combined once: l = lA + lB, O = (OA + OB) / l.
constexpr int KHALF = 512;   // keys per half (8 chunks of 64)
// NQT = query tiles (16 queries each) per workgroup, two waves (key halves) per tile: 4 = the form above (64 queries, 512 threads);
// 2 (round 6) = 32 queries on 256 threads for launches of one or two pairs, whose 128 - 256 workgroups of the wide form leave half
// the chip idle (the redo engine of a strict handle, the per-call host API).  A wave's operations do not depend on NQT: same bits.
template <int NQT>
__global__ void __launch_bounds__(128 * NQT, 2) attn_kernel(const float *qkv /*[img][NP][768]*/, const int *counts,
                                                      int cross, float *o /*[img][NP][256]*/) {
  constexpr int NT = 128 * NQT, QB = 16 * NQT;          // threads, queries per workgroup
  constexpr int RPT = NT / 32, SU = 64 / RPT;           // staging: key rows per pass and half, passes per chunk
  __shared__ __attribute__((aligned(16))) float kv[2][2][64 * VSTR];  // [double buffer][half][chunk]
  __shared__ float s_max[2 * NQT][16];
  __shared__ float s_l[NQT][16];
  int qtile, grp;
  xcd_group_map(blockIdx.x, NP / QB, (int)gridDim.x / (NP / QB), qtile, grp);   // the query tiles of a head on one XCD
  const int im = grp >> 2, sm = cross ? (im ^ 1) : im;
  const int head = grp & 3;
  const int nq = counts[im], ns = counts[sm];
  const int q0 = qtile * QB;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = wave % NQT, kh = wave / NQT;
  const int px = lane & 15, g = lane >> 4;
  const float *qb = qkv + ((size_t)im * NP) * 768 + head * 64;
  const float *kb = qkv + ((size_t)sm * NP) * 768 + 256 + head * 64;
  const float *vb = qkv + ((size_t)sm * NP) * 768 + 512 + head * 64;
  const int nsA = ns < KHALF ? ns : KHALF;              // keys in half A
  const int nrounds = (nsA + 63) >> 6;                  // chunks of half A (>= chunks of half B)
  const int kbase_h = kh * KHALF;                       // first key of this wave's half

  // staging: thread -> (half sh, key row sr + RPT u, float4 sj)
  const int sh = tid / (NT / 2), st = tid % (NT / 2);
  const int sj = st & 15, sr = st >> 4;
  f32x4 pf[SU];
  auto issue = [&](const float *base, int ch) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int key = sh * KHALF + ch * 64 + sr + RPT * u;
      pf[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (key < ns) pf[u] = *(const f32x4 *)(base + (size_t)key * 768 + 4 * sj);
    }
  };
  auto commit_k = [&](int buf) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      float *dst = kv[buf][sh] + (sr + RPT * u) * KSTR + 4 * sj;
      *(float2 *)dst = make_float2(pf[u][0], pf[u][1]);
      *(float2 *)(dst + 2) = make_float2(pf[u][2], pf[u][3]);
    }
  };
  auto commit_v = [&](int buf) {
#pragma unroll
    for (int u = 0; u < SU; ++u) *(f32x4 *)(kv[buf][sh] + (sr + RPT * u) * VSTR + 4 * sj) = pf[u];
  };

  issue(kb, 0);
  float qreg[16];
  {
    const float *qr = qb + (size_t)(q0 + qt * 16 + px) * 768 + g;
#pragma unroll
    for (int s = 0; s < 16; ++s) qreg[s] = qr[4 * s];
  }
  commit_k(0);
  __syncthreads();

  f32x4 sreg[32];
  // ---------------- phase 1: S^T = K Q^T over this wave's key half
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    if (ch < nrounds) {
      if (ch + 1 < nrounds) issue(kb, ch + 1);
      f32x4 acc[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      const float *ap = kv[ch & 1][kh] + px * KSTR + g;
      {  // software pipeline over the 16 k-steps: reads of step s+1 before the MFMAs of step s
        float ka[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) ka[0][kt] = ap[kt * 16 * KSTR];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int cur = s & 1;
          if (s + 1 < 16) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) ka[cur ^ 1][kt] = ap[kt * 16 * KSTR + 4 * (s + 1)];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[cur][kt], qreg[s], acc[kt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int kb0 = kbase_h + ch * 64 + kt * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[kt][r] = (kb0 + r < ns) ? acc[kt][r] * 0.125f : -FLT_MAX;
        sreg[ch * 4 + kt] = acc[kt];
      }
      if (ch + 1 < nrounds) commit_k((ch + 1) & 1);
      __syncthreads();
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) sreg[ch * 4 + kt] = f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    }
  }
  issue(vb, 0);
  // ---------------- row max over both halves
  float m = -FLT_MAX;
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, sreg[t][r]);
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  if (g == 0) s_max[wave][px] = m;
  commit_v(0);
  __syncthreads();
  m = fmaxf(m, s_max[wave < NQT ? wave + NQT : wave - NQT][px]);   // the wave of this query tile's other key half
  // ---------------- phase 2: p = exp_c(s - m) one tile ahead of its O^T += V^T P^T
  const int nsl = ns - kbase_h;  // keys of this half (may be <= 0)
  float part = 0.0f;
  if (nsl > 0) {
    {   // (part = ((part + p0) + p1) + p2) + p3: the canonical order, whatever computes the p)
      const f32x2 a = exp_c_nonpos2(f32x2{sreg[0][0] - m, sreg[0][1] - m}), b = exp_c_nonpos2(f32x2{sreg[0][2] - m, sreg[0][3] - m});
      sreg[0][0] = a[0]; sreg[0][1] = a[1]; sreg[0][2] = b[0]; sreg[0][3] = b[1];
      part = part + a[0]; part = part + a[1]; part = part + b[0]; part = part + b[1];
    }
  }
  f32x4 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    if (ch < nrounds) {
      if (ch + 1 < nrounds) issue(vb, ch + 1);
      if (ch * 64 + 64 <= nsl) {
        // all 64 keys of the chunk valid (every chunk but the last): the V^T fragments of step s + 1 are read from LDS
        // before the MFMAs of step s and pinned there (sched_barrier) -- left alone, the scheduler reads each fragment
        // into the registers the previous MFMAs just released and every pair of MFMAs waits out an LDS round trip.
        // Same operations in the same order as the generic path below.
        const float *vbase = kv[ch & 1][kh] + (4 * g) * VSTR + px;
        float va[2][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) va[0][dt] = vbase[dt * 16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int kt = s >> 2, r = s & 3, T = ch * 4 + kt;
          if (r == 0 && T + 1 < 32 && (T + 1) * 16 < nsl) {
            {
              f32x4 &sv = sreg[T + 1 < 32 ? T + 1 : 31];
              const f32x2 a = exp_c_nonpos2(f32x2{sv[0] - m, sv[1] - m}), b = exp_c_nonpos2(f32x2{sv[2] - m, sv[3] - m});
              sv[0] = a[0]; sv[1] = a[1]; sv[2] = b[0]; sv[3] = b[1];
              part = part + a[0]; part = part + a[1]; part = part + b[0]; part = part + b[1];
            }
          }
          if (s + 1 < 16) {
            const float *nb = vbase + (((s + 1) >> 2) * 16 + ((s + 1) & 3)) * VSTR;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) va[(s + 1) & 1][dt] = nb[dt * 16];
          }
          __builtin_amdgcn_sched_barrier(0);
          const float pb = sreg[T][r];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s & 1][dt], pb, oacc[dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int T = ch * 4 + kt;
        if (T * 16 < nsl) {
          if (T + 1 < 32 && (T + 1) * 16 < nsl) {
            {
              f32x4 &sv = sreg[T + 1 < 32 ? T + 1 : 31];
              const f32x2 a = exp_c_nonpos2(f32x2{sv[0] - m, sv[1] - m}), b = exp_c_nonpos2(f32x2{sv[2] - m, sv[3] - m});
              sv[0] = a[0]; sv[1] = a[1]; sv[2] = b[0]; sv[3] = b[1];
              part = part + a[0]; part = part + a[1]; part = part + b[0]; part = part + b[1];
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pb = sreg[T][r];
            const float *ap = kv[ch & 1][kh] + (kt * 16 + 4 * g + r) * VSTR + px;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
              oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[dt * 16], pb, oacc[dt], 0, 0, 0);
          }
        }
      }
      if (ch + 1 < nrounds) commit_v((ch + 1) & 1);
      __syncthreads();
    }
  }
  float l = part + __shfl_xor(part, 16, 64);
  l = l + __shfl_xor(l, 32, 64);
  // ---------------- combine the two halves: O = (OA + OB) / (lA + lB)
  float *xo = &kv[0][0][0] + qt * (64 * 68);   // 64 lanes x 16 floats (+pad) per query tile
  if (kh == 1) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4 *)(xo + lane * 68 + 4 * dt) = oacc[dt];
    if (g == 0) s_l[qt][px] = l;
  }
  __syncthreads();
  if (kh == 0) {
    const float lt = l + s_l[qt][px];
    const int q = q0 + qt * 16 + px;
    float *op = o + ((size_t)im * NP + q) * 256 + head * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 ob = *(const f32x4 *)(xo + lane * 68 + 4 * dt);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (oacc[dt][r] + ob[r]) / lt;
      *(f32x4 *)(op + dt * 16) = v;
    }
  }
}

// ------------------------------------------------------------ score matrix
// S_ij = (chain_c fma(m0_ic, m1_jc, 0)) / 16 written into the couplings matrix
// C[pair][i][j] (ld LDC) and its transpose Ct[pair][j][i].
__global__ void __launch_bounds__(256) score_kernel(const float *mdesc /*[img][NP][256]*/, const int *counts,
                                                    float *C, float *Ct) {
  __shared__ __attribute__((aligned(16))) float at[64 * 66];
  __shared__ __attribute__((aligned(16))) float bt[64 * 66];
  const int p = blockIdx.z;
  const int n0 = counts[2 * p], n1 = counts[2 * p + 1];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  if (i0 >= n0 || j0 >= n1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const float *m0 = mdesc + ((size_t)(2 * p) * NP) * 256;
  const float *m1 = mdesc + ((size_t)(2 * p + 1) * NP) * 256;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int c0 = 0; c0 < 256; c0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * 16; i += 256) {
      const int r = i >> 4, j = i & 15;
      const f32x4 va = *(const f32x4 *)(m0 + (size_t)(i0 + r) * 256 + c0 + 4 * j);
      const f32x4 vb = *(const f32x4 *)(m1 + (size_t)(j0 + r) * 256 + c0 + 4 * j);
      float *da = at + r * 66 + 4 * j, *db = bt + r * 66 + 4 * j;
      *(float2 *)da = make_float2(va[0], va[1]); *(float2 *)(da + 2) = make_float2(va[2], va[3]);
      *(float2 *)db = make_float2(vb[0], vb[1]); *(float2 *)(db + 2) = make_float2(vb[2], vb[3]);
    }
    __syncthreads();
    const float *ap = at + (wave * 16 + px) * 66 + g;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const float a = ap[k];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bt[(t * 16 + px) * 66 + g + k], acc[t], 0, 0, 0);
    }
  }
  float *Cp = C + (size_t)p * (NP + 1) * LDC, *Ctp = Ct + (size_t)p * (NP + 1) * LDC;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + t * 16 + px;
    const int ib = i0 + wave * 16 + 4 * g;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[t][r] * 0.0625f;
    if (j < n1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ib + r < n0) Cp[(size_t)(ib + r) * LDC + j] = v[r];
      if (ib + 3 < n0) *(f32x4 *)(Ctp + (size_t)j * LDC + ib) = v;
      else
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ib + r < n0) Ctp[(size_t)j * LDC + ib + r] = v[r];
    }
  }
}

// dustbin row/column = bin_score (src/super_glue.cpp:466-474); u = v = 0
__global__ void __launch_bounds__(256) ot_init_kernel(const int *counts, float alpha, float *C, float *Ct, float *u,
                                                      float *v) {
  const int p = blockIdx.y;
  const int n0 = counts[2 * p], n1 = counts[2 * p + 1];
  float *Cp = C + (size_t)p * (NP + 1) * LDC, *Ctp = Ct + (size_t)p * (NP + 1) * LDC;
  for (int i = blockIdx.x * 256 + threadIdx.x; i <= NP; i += gridDim.x * 256) {
    if (i <= n0) { Cp[(size_t)i * LDC + n1] = alpha; Ctp[(size_t)n1 * LDC + i] = alpha; }
    if (i <= n1) { Cp[(size_t)n0 * LDC + i] = alpha; Ctp[(size_t)i * LDC + n0] = alpha; }
    u[(size_t)p * LDC + i] = 0.0f;
    v[(size_t)p * LDC + i] = 0.0f;
  }
}

// One Sinkhorn half-iteration (src/super_glue.cpp:436-451, max-stabilised):
//   out[r] = log_marg[r] - LSE_c( M[r][c] + add[c] ),  r < R, c < Cn
// Lane l holds the columns 256t + 4l + r (16-byte loads, <= 5 per row) of the row and of `add`;
// the sum is the canonical wave-strided-by-4 sum.  ROWPASS: M=C, R=n0+1, Cn=n1+1, add=v, out=u.
// Grid-stride over rows: wave w of the launch handles rows w, w + W, w + 2W, ... (W = waves in the
// pair's grid slice) with the next row's five 16-byte loads in flight while the current row is
// reduced.  The launch is sized to about four 4-wave workgroups per CU instead of one wave per
// row: a half-iteration is latency-bound, and 32 resident waves per CU would only keep the
// MFMA-bound kernels of the other streams (GNN of the next batch, SuperPoint) off the CUs.
// The arithmetic of one row (every lane ends with the same value): x = M + add over the columns below Cn, the maximum by
// the butterfly, the canonical wave-strided-by-4 sum of exp(x - max), and max + log(sum).
template <bool FAST>
__device__ __forceinline__ float sinkhorn_row_lse(const f32x4 (&cur)[5], const f32x4 (&av)[5], int lane, int Cn) {
  f32x4 x[5];
  float m = -FLT_MAX;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int c = 256 * t + 4 * lane;
    x[t] = f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c + r < Cn) { x[t][r] = cur[t][r] + av[t][r]; m = fmaxf(m, x[t][r]); }
  }
  m = bfly64_max(m);
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int c = 256 * t + 4 * lane;
    if (FAST) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c + r < Cn) s = s + __expf(x[t][r] - m);
    } else {   // the canonical exponential, two elements per packed instruction (exp_c_nonpos2: the same bits); same order of the sum
      const f32x2 a = exp_c_nonpos2(f32x2{x[t][0] - m, x[t][1] - m}), b = exp_c_nonpos2(f32x2{x[t][2] - m, x[t][3] - m});
      if (c + 0 < Cn) s = s + a[0];
      if (c + 1 < Cn) s = s + a[1];
      if (c + 2 < Cn) s = s + b[0];
      if (c + 3 < Cn) s = s + b[1];
    }
  }
  s = bfly64_sum(s);
  return m + (FAST ? __logf(s) : log_c(s));
}

template <bool ROWPASS, bool FAST>
__global__ void __launch_bounds__(256) sinkhorn_half_kernel(const int *counts, const float *M, const float *add,
                                                            float *out) {
  // No LDS, no barrier, one memory round trip: the row's five 16-byte pieces, the lane's 20 values of the
  // other side's vector (the same columns for every row of the wave; L1/L2 hits) and the two counts are all
  // requested before anything is waited for.  Bounds are applied afterwards: every piece below LDC is inside
  // the allocation whatever the counts are.
  const int p = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float *Mp = M + (size_t)p * (NP + 1) * LDC;
  const float *ad = add + (size_t)p * LDC;
  auto load_row = [&](int r, f32x4 (&mv)[5]) {
    const float *mr = Mp + (size_t)(r <= NP ? r : 0) * LDC;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int c = 256 * t + 4 * lane;
      mv[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (c + 4 <= LDC) mv[t] = *(const f32x4 *)(mr + c);
    }
  };
  f32x4 cur[5], nxt[5], av[5];
  load_row(row, cur);
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int c = 256 * t + 4 * lane;
    av[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (c + 4 <= LDC) av[t] = *(const f32x4 *)(ad + c);
  }
  const int n0 = counts[2 * p], n1 = counts[2 * p + 1];
  const int R = (ROWPASS ? n0 : n1) + 1, Cn = (ROWPASS ? n1 : n0) + 1;
  const float norm = -log_c((float)(n0 + n1));
  for (; row < R; row += nw) {
    if (row + nw < R) load_row(row + nw, nxt);
    const float lse = sinkhorn_row_lse<FAST>(cur, av, lane, Cn);
    if (lane == 0) {
      const int last = R - 1;
      const float lm = (row < last) ? norm : (log_c((float)(ROWPASS ? n1 : n0)) + norm);
      out[(size_t)p * LDC + row] = lm - lse;
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) cur[t] = nxt[t];
  }
}

// ------------------------------------------------------------------ decode
// Z_ij = ((C_ij + u_i) + v_j) - norm ; row / column argmax over the inner
// n0 x n1 block with first-max-wins (max_matrix, src/super_glue.cpp:314-343).
// ROWS: wave per i on C; else wave per j on Ct.  Optionally writes Z rows.
// GUARD (guarded fast mode): also the runner-up of every row / column; a pair is flagged (gflags[p] != 0) when a best
// entry that can become a match -- at or above the matching threshold minus the margin gz (log domain) -- is closer than
// gz to the threshold (bit 0) or closer than 2 gz to its runner-up (bit 1): decisions the exact mode could take differently.
// RESID (every mode with a fast Sinkhorn; the COLUMN pass): the column marginal of the plan the decode is about to read --
// sum_i exp(Z_ij) over the column's n0 rows and its dustbin entry.  An iteration ends with the column update v = log nu - LSE_i(C + u),
// so this sum is 1 to rounding in every correct result, converged or not (the ROW marginals are not an invariant: after 100
// iterations they still miss by up to 0.4 on the bench streams).  resid[p] = the largest |sum - 1| over the pair's columns: the
// integrity word of the pair's Sinkhorn result (sg_api.hip, pm_check_resident) -- it catches a result whose last iteration, final
// potentials or couplings were damaged, not a transient error of an earlier iteration that the later ones have absorbed.
// (The kernel writes one word per column, resid[p][column]; decode_kernel reduces them.)
template <bool ROWS, bool GUARD, bool RESID>
__global__ void __launch_bounds__(256) argmax_kernel(const int *counts, const float *M, const float *u, const float *v,
                                                     int *midx, float *mval, float *Zout, int *gflags, float gz,
                                                     float log_thr, float *resid) {
  const int p = blockIdx.y;
  const int n0 = counts[2 * p], n1 = counts[2 * p + 1];
  const int R = ROWS ? n0 : n1, Cn = ROWS ? n1 : n0;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const float norm = -log_c((float)(n0 + n1));
  const float *mr = M + (size_t)p * (NP + 1) * LDC + (size_t)row * LDC;
  const float *up = u + (size_t)p * LDC, *vp = v + (size_t)p * LDC;
  if (ROWS && Zout && row <= n0) {  // full (n0+1) x (n1+1) tensor, dustbins included
    float *zr = Zout + (size_t)p * (NP + 1) * LDC + (size_t)row * LDC;
    const float ui = up[row];
    for (int c = lane; c <= n1; c += 64) zr[c] = ((mr[c] + ui) + vp[c]) - norm;
  }
  if (row >= R) return;
  // reference semantics (max_matrix): value starts at -FLT_MAX, index 0, strict '<'
  float best = -FLT_MAX, second = -FLT_MAX;
  int bi = 0;
  float rsum = 0.0f;
  for (int c = lane; c < Cn; c += 64) {
    const float z = ROWS ? (((mr[c] + up[row]) + vp[c]) - norm) : (((mr[c] + up[c]) + vp[row]) - norm);
    if (best < z) { if (GUARD) second = best; best = z; bi = c; }
    else if (GUARD && second < z) second = z;
    if (RESID) rsum = rsum + __expf(z);
  }
  if (RESID) {
    if (lane == 0) rsum = rsum + __expf((ROWS ? ((mr[Cn] + up[row]) + vp[Cn]) : ((mr[Cn] + up[Cn]) + vp[row])) - norm);   // the dustbin entry
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) rsum = rsum + __shfl_xor(rsum, s, 64);
    // one word per column, reduced by the pair's decode workgroup (a thousand atomic maxima on one address per pair doubled this
    // kernel's time: 55 -> 120 us per batch of eight, 2.8 % of the fast mode's step)
    if (lane == 0 && resid) resid[(size_t)p * NP + row] = fabsf(rsum - 1.0f);
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const float ob = __shfl_xor(best, s, 64);
    const int oi = __shfl_xor(bi, s, 64);
    if (GUARD) {
      const float os = __shfl_xor(second, s, 64);
      second = fmaxf(fmaxf(second, os), fminf(best, ob));   // the loser of the two bests is a runner-up candidate
    }
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) {
    midx[(size_t)p * NP + row] = bi;
    mval[(size_t)p * NP + row] = best;
    if (GUARD && best >= log_thr - gz) {
      int bits = 0;
      if (best - log_thr <= gz) bits |= 1;
      if (best - second <= 2.0f * gz) bits |= 2;
      if (bits) atomicOr(&gflags[p], bits);
    }
  }
}

// decode() tail (src/super_glue.cpp:345-430) + match assembly
// (src/point_matching.cc:26-45).  One 1024-thread workgroup per pair.
struct DMatch { int queryIdx, trainIdx; float distance; };

__device__ __forceinline__ int block_excl_scan1024(int v, int *wsum, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    const int t = wsum[w];
    if (w < wave) base += t;
    tot += t;
  }
  total = tot;
  return base + incl - v;
}

__global__ void __launch_bounds__(1024) decode_kernel(const int *counts, const int *mi0, const float *mv0,
                                                      const int *mi1, double thresh, const float *kxy,
                                                      int *idx0, int *idx1, double *ms0,
                                                      double *ms1, DMatch *matches, float *pts0, float *pts1,
                                                      int *nmatch, const float *resid_cols, float *resid, float resid_bound,
                                                      int *err) {
  __shared__ int s_valid0[NP];
  __shared__ double s_ms0[NP];
  __shared__ int wsum[16];
  __shared__ float wres[16];
  const int p = blockIdx.x, i = threadIdx.x;
  const int n0 = counts[2 * p], n1 = counts[2 * p + 1];
  const int *a0 = mi0 + (size_t)p * NP, *a1 = mi1 + (size_t)p * NP;
  int my0 = -1;
  double m0 = 0.0;
  if (i < n0) {
    const int j = a0[i];
    const bool mutual0 = (j < n1) && (a1[j] == i);
    m0 = mutual0 ? (double)exp_c(mv0[(size_t)p * NP + i]) : 0.0;
    const bool valid0 = mutual0 && (m0 > thresh);
    s_valid0[i] = valid0 ? 1 : 0;
    s_ms0[i] = m0;
    my0 = valid0 ? j : -1;
    idx0[(size_t)p * NP + i] = my0;
    ms0[(size_t)p * NP + i] = m0;
  }
  __syncthreads();
  if (i < n1) {
    const int k = a1[i];
    const bool mutual1 = (k < n0) && (a0[k] == i);
    const double m1 = mutual1 ? s_ms0[k] : 0.0;
    const bool valid1 = mutual1 && s_valid0[k];
    idx1[(size_t)p * NP + i] = valid1 ? k : -1;
    ms1[(size_t)p * NP + i] = m1;
  }
  // match list: i with 0 <= idx0[i] < n1 and idx1[idx0[i]] == i  (always true for valid0)
  const int flag = (my0 >= 0) ? 1 : 0;
  int total;
  const int pos = block_excl_scan1024(flag, wsum, total);
  if (flag) {
    // mscores1[idx0[i]] == mscores0[i] for a valid mutual match
    const double d = 1.0 - (m0 + m0) / 2.0;
    DMatch dm;
    dm.queryIdx = i; dm.trainIdx = my0; dm.distance = (float)d;
    matches[(size_t)p * NP + pos] = dm;
    const float *k0 = kxy + ((size_t)(2 * p) * NP) * 2, *k1 = kxy + ((size_t)(2 * p + 1) * NP) * 2;
    pts0[((size_t)p * NP + pos) * 2] = k0[2 * i];
    pts0[((size_t)p * NP + pos) * 2 + 1] = k0[2 * i + 1];
    pts1[((size_t)p * NP + pos) * 2] = k1[2 * my0];
    pts1[((size_t)p * NP + pos) * 2 + 1] = k1[2 * my0 + 1];
  }
  if (i == 0) nmatch[p] = total;
  // integrity of the pair's Sinkhorn result (argmax_kernel, RESID): the largest column residual of the pair (a NaN wins: the
  // comparisons below are written so); above the bound the launch is reported like a give-up -- err[0] = 2 unless a give-up (1)
  // is already there, err[2..3] = the pairs -- and the host redoes the tail
  if (resid_cols) {
    float r = (i < n1) ? resid_cols[(size_t)p * NP + i] : 0.0f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float o = __shfl_xor(r, d, 64);
      r = (o > r || o != o) ? o : r;
    }
    if ((i & 63) == 0) wres[i >> 6] = r;
    __syncthreads();
    if (i == 0) {
      float m = wres[0];
      for (int w = 1; w < 16; ++w) m = (wres[w] > m || wres[w] != wres[w]) ? wres[w] : m;
      resid[p] = m;
      if (!(m <= resid_bound)) {
        atomicCAS(err, 0, 2);
        atomicOr(err + 2 + (p >> 5), 1 << (p & 31));
      }
    }
  }
}

// ----------------------------------------------------------------- guard calibration
// Calibration of the matcher's margin: the largest |Z_fast - Z_exact| over the entries of the log-assignment matrices (keypoint
// rows and columns, no dustbins) that either pass holds above log_floor -- the entries a decision can rest on.
__global__ void __launch_bounds__(256) guard_z_calib_kernel(const int *counts, const float *zf, const float *zx, float log_floor,
                                                            int *out) {
  const int p = blockIdx.y, n0 = counts[2 * p], n1 = counts[2 * p + 1];
  float worst = 0.0f;
  for (int i = blockIdx.x; i < n0; i += gridDim.x) {
    const size_t row = ((size_t)p * (NP + 1) + i) * LDC;
    for (int j = threadIdx.x; j < n1; j += 256) {
      const float a = zf[row + j], x = zx[row + j];
      if (a > log_floor || x > log_floor) worst = fmaxf(worst, a > x ? a - x : x - a);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) worst = fmaxf(worst, __shfl_xor(worst, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_int(worst));
}
int launch_guard_z_calib(const int *counts, const float *zf, const float *zx, float log_floor, int *out, int P, hipStream_t st) {
  hipLaunchKernelGGL(guard_z_calib_kernel, dim3(128, P), dim3(256), 0, st, counts, zf, zx, log_floor, out);
  URF_HIP(hipGetLastError());
  return 0;
}

// Online check of the guard's error model (strict parity): every exact redo leaves the exact couplings and potentials of a pair
// the fast pass has also matched.  For the fast pass's own row / column best entries (mi / mv of its decode; the entries its
// decisions rested on) the exact log-assignment is recomputed from the engine's buffers; the largest |Z_fast - Z_exact| over the
// entries either pass holds above log_floor comes back as float bits, one word per redone pair.  One workgroup per pair.
__global__ void __launch_bounds__(256) guard_online_kernel(const int *counts, const int *mi0f, const float *mv0f, const int *mi1f,
                                                           const float *mv1f, const float *C, const float *u, const float *v,
                                                           float log_floor, int *out) {
  const int p = blockIdx.x, n0 = counts[2 * p], n1 = counts[2 * p + 1];
  const float norm = -log_c((float)(n0 + n1));
  const float *Cp = C + (size_t)p * (NP + 1) * LDC, *up = u + (size_t)p * LDC, *vp = v + (size_t)p * LDC;
  float worst = 0.0f;
  for (int i = threadIdx.x; i < n0; i += 256) {
    const int j = mi0f[(size_t)p * NP + i];
    const float a = mv0f[(size_t)p * NP + i];
    if (j >= 0 && j < n1) {
      const float x = ((Cp[(size_t)i * LDC + j] + up[i]) + vp[j]) - norm;
      if (a > log_floor || x > log_floor) worst = fmaxf(worst, a > x ? a - x : x - a);
    }
  }
  for (int j = threadIdx.x; j < n1; j += 256) {
    const int i = mi1f[(size_t)p * NP + j];
    const float a = mv1f[(size_t)p * NP + j];
    if (i >= 0 && i < n0) {
      const float x = ((Cp[(size_t)i * LDC + j] + up[i]) + vp[j]) - norm;
      if (a > log_floor || x > log_floor) worst = fmaxf(worst, a > x ? a - x : x - a);
    }
  }
  __shared__ float part[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) worst = fmaxf(worst, __shfl_xor(worst, d, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) out[p] = __float_as_int(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
}
int launch_guard_online(const int *counts, const int *mi0f, const float *mv0f, const int *mi1f, const float *mv1f, const float *C,
                        const float *u, const float *v, float log_floor, int *out, int n, hipStream_t st) {
  hipLaunchKernelGGL(guard_online_kernel, dim3(n), dim3(256), 0, st, counts, mi0f, mv0f, mi1f, mv1f, C, u, v, log_floor, out);
  URF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ launchers
int launch_sg_prep_slots(const float *const *slots, int nimg, int width, int height, int *counts, float *kin,
                         float *kxy, float *x, hipStream_t st) {
  hipLaunchKernelGGL(sg_prep_slots_kernel, dim3(64, nimg), dim3(256), 0, st, slots, width, height, counts, kin, kxy, x);
  URF_HIP(hipGetLastError());
  return 0;
}
int g_attn_exact_nqt = -1;   // (urf_probe_attn_exact_nqt, experiments build: 2 / 4 = that form for every launch, 0 = the policy)
int launch_attn(const float *qkv, const int *counts, int cross, float *o, int nimg, hipStream_t st) {
  // one or two pairs: 32-query workgroups (twice as many, half the size); URF_ATTN_EXACT_NQT (experiments build) forces 2 or 4
  if (g_attn_exact_nqt < 0) { const char *e = urf::exp_env("URF_ATTN_EXACT_NQT"); g_attn_exact_nqt = e ? atoi(e) : 0; }
  const int nqt = g_attn_exact_nqt == 2 || g_attn_exact_nqt == 4 ? g_attn_exact_nqt : (nimg <= 4 ? 2 : 4);
  if (nqt == 2) hipLaunchKernelGGL(attn_kernel<2>, dim3((NP / 32) * 4 * nimg), dim3(256), 0, st, qkv, counts, cross, o);
  else hipLaunchKernelGGL(attn_kernel<4>, dim3((NP / 64) * 4 * nimg), dim3(512), 0, st, qkv, counts, cross, o);
  URF_HIP(hipGetLastError());
  return 0;
}
int launch_score(const float *mdesc, const int *counts, float alpha, float *C, float *Ct, float *u, float *v, int P,
                 hipStream_t st) {
  hipLaunchKernelGGL(score_kernel, dim3(NP / 64, NP / 64, P), dim3(256), 0, st, mdesc, counts, C, Ct);
  hipLaunchKernelGGL(ot_init_kernel, dim3(5, P), dim3(256), 0, st, counts, alpha, C, Ct, u, v);
  URF_HIP(hipGetLastError());
  return 0;
}
int launch_sinkhorn(const int *counts, const float *C, const float *Ct, float *u, float *v, int iters, int P,
                    bool fast, hipStream_t st) {
  // rows per wave k so that the whole launch is about `target` workgroups (4 per CU; measured at 8 pairs:
  // 1 row/wave 1332 frames/s, 1024 workgroups 1385, 512 -> 1358, 256 -> 1319 in the 3-stream pipeline)
  static int target = -1;
  if (target < 0) {
    const char *e = urf::exp_env("URF_SINKHORN_BLOCKS");   // tuning knob for A/B runs
    target = e ? atoi(e) : 1024;
    if (target < 1) target = 1024;
  }
  const int rows = NP + 1;
  int k = (P * rows + 4 * target - 1) / (4 * target);
  if (k < 1) k = 1;
  const dim3 grid((rows + 4 * k - 1) / (4 * k), P), block(256);
  for (int it = 0; it < iters; ++it) {
    if (fast) {
      hipLaunchKernelGGL((sinkhorn_half_kernel<true, true>), grid, block, 0, st, counts, C, v, u);
      hipLaunchKernelGGL((sinkhorn_half_kernel<false, true>), grid, block, 0, st, counts, Ct, u, v);
    } else {
      hipLaunchKernelGGL((sinkhorn_half_kernel<true, false>), grid, block, 0, st, counts, C, v, u);
      hipLaunchKernelGGL((sinkhorn_half_kernel<false, false>), grid, block, 0, st, counts, Ct, u, v);
    }
  }
  URF_HIP(hipGetLastError());
  return 0;
}
int launch_decode(const int *counts, const float *C, const float *Ct, const float *u, const float *v, double thresh,
                  const float *kxy, int *mi0, float *mv0, int *mi1, float *mv1,
                  int *idx0, int *idx1, double *ms0, double *ms1, void *matches, float *pts0, float *pts1,
                  int *nmatch, float *Zout, int *gflags, float gz, float *resid_cols, float *resid, float resid_bound, int *err, int P,
                  hipStream_t st) {
  const dim3 grid((NP + 1 + 3) / 4, P), block(256);
  const float log_thr = thresh > 0.0 ? (float)log(thresh) : -FLT_MAX;
  if (gflags) {
    hipLaunchKernelGGL((argmax_kernel<true, true, false>), grid, block, 0, st, counts, C, u, v, mi0, mv0, Zout, gflags, gz, log_thr, (float *)nullptr);
    hipLaunchKernelGGL((argmax_kernel<false, true, true>), grid, block, 0, st, counts, Ct, u, v, mi1, mv1, (float *)nullptr, gflags, gz, log_thr, resid_cols);
  } else if (resid_cols) {
    hipLaunchKernelGGL((argmax_kernel<true, false, false>), grid, block, 0, st, counts, C, u, v, mi0, mv0, Zout, gflags, gz, log_thr, (float *)nullptr);
    hipLaunchKernelGGL((argmax_kernel<false, false, true>), grid, block, 0, st, counts, Ct, u, v, mi1, mv1, (float *)nullptr, gflags, gz, log_thr, resid_cols);
  } else {
    hipLaunchKernelGGL((argmax_kernel<true, false, false>), grid, block, 0, st, counts, C, u, v, mi0, mv0, Zout, gflags, gz, log_thr, (float *)nullptr);
    hipLaunchKernelGGL((argmax_kernel<false, false, false>), grid, block, 0, st, counts, Ct, u, v, mi1, mv1, (float *)nullptr, gflags, gz, log_thr, (float *)nullptr);
  }
  hipLaunchKernelGGL(decode_kernel, dim3(P), dim3(1024), 0, st, counts, mi0, mv0, mi1, thresh, kxy, idx0,
                     idx1, ms0, ms1, (DMatch *)matches, pts0, pts1, nmatch, (const float *)resid_cols, resid, resid_bound, err);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf

#ifdef URF_EXPERIMENTS
extern "C" int urf_probe_attn_exact_nqt(int v) { urf::g_attn_exact_nqt = v; return 0; }
#endif
