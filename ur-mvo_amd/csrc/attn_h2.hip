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
#include <type_traits>
#include <stdlib.h>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// (p0, p1) -> packed f16 pairs hi = f16(p), lo = f16(p - hi).  The residuals come from v_fma_mix{lo,hi}_f16, which read
// hi straight from its packed half and round p - hi to f16 in one instruction each (the difference is exact in fp32, so
// the result is the same as convert-back / subtract / convert: 3 VALU instructions per pair instead of 6)
__device__ __forceinline__ void split_pair(float p0, float p1, unsigned &hi, unsigned &lo) {
  const f16x2 h = {(_Float16)p0, (_Float16)p1};
  hi = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(hi));
  lo = l;
}

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
// QT query tiles (16 queries each) per wave, NW waves per workgroup: every K / V^T fragment read from
// LDS feeds QT MFMAs instead of one.
// At most 192 VGPRs: see h2conv_kernel (co-residency with SuperPoint's convolutions in the pipeline)
#ifdef URF_ATTN_STAMPS   // diagnostic build only (make ATTN_STAMPS=1; tools/gpu_attn_stamps.py): s_memtime at the phase boundaries
__device__ long long g_attn_stamps[2][64][8];
#define AT_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && ch < 64) g_attn_stamps[wave >> 2][ch][i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif

template <int QT, int NW>
__global__ void __launch_bounds__(64 * NW) attn_h2_kernel(const _Float16 *qkh, const _Float16 *qkl,
                                                          const _Float16 *vth, const _Float16 *vtl, const int *counts,
                                                          int cross, _Float16 *oh, _Float16 *ol) {
  constexpr int NT = 64 * NW, QB = 16 * QT * NW;   // threads, queries per workgroup
  constexpr int NV = 512 / NT;                     // staging roles per thread (512 = 2 tensors x 2 planes x 16 rows x 8 pieces)
  // [buffer][K planes | V planes]
  __shared__ __attribute__((aligned(16))) _Float16 kbuf[2][2][64 * AS];
  __shared__ __attribute__((aligned(16))) _Float16 vbuf[2][2][64 * VS];
  int qb, grp;
  xcd_group_map(blockIdx.x, ANP / QB, (int)gridDim.x / (ANP / QB), qb, grp);   // the query tiles of a head on one XCD
  const int im = grp >> 2, sm = cross ? (im ^ 1) : im;
  const int head = grp & 3;
  const int nq = counts[im], ns = counts[sm];
  const int q0 = qb * QB;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int nchunk = (ns + 63) >> 6;

  // staging: virtual thread vt -> (tensor sk: 0 = K, 1 = V^T; plane sp; row sr (+16u); 16-byte piece sj)
  f16x8 pf[NV][4];
  // BOUND = false: all 64 keys of the chunk are valid -- no per-key tests, no zeroing of stale tokens
  auto issue = [&](int ch, auto bound_tag) {
    constexpr bool BOUND = decltype(bound_tag)::value;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int vt = tid + NT * v;
      const int sk = vt >> 8, sp = (vt >> 7) & 1, st = vt & 127;
      const int sj = st & 7, sr = st >> 3;
      if (sk == 0) {
        const _Float16 *base = (sp ? qkl : qkh) + ((size_t)sm * ANP) * 512 + 256 + head * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int key = ch * 64 + sr + 16 * u;
          if (BOUND) {
            pf[v][u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (key < ns) pf[v][u] = *(const f16x8 *)(base + (size_t)key * 512 + 8 * sj);
          } else {
            pf[v][u] = *(const f16x8 *)(base + (size_t)key * 512 + 8 * sj);
          }
        }
      } else {
        const _Float16 *base = (sp ? vtl : vth) + ((size_t)sm * 256 + head * 64) * ANP;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = sr + 16 * u;
          const int key0 = ch * 64 + 8 * sj;
          if (BOUND) {
            f16x8 x = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (key0 < ns) {
              x = *(const f16x8 *)(base + (size_t)d * ANP + key0);
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (key0 + e >= ns) x[e] = (_Float16)0.0f;   // stale tokens beyond the count
            }
            pf[v][u] = x;
          } else {
            pf[v][u] = *(const f16x8 *)(base + (size_t)d * ANP + key0);
          }
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int vt = tid + NT * v;
      const int sk = vt >> 8, sp = (vt >> 7) & 1, st = vt & 127;
      const int sj = st & 7, sr = st >> 3;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sk == 0) *(f16x8 *)(kbuf[buf][sp] + (sr + 16 * u) * AS + 8 * sj) = pf[v][u];
        else *(f16x8 *)(vbuf[buf][sp] + (sr + 16 * u) * VS + 8 * sj) = pf[v][u];
      }
    }
  };

  // the Q fragments are loaded BEFORE the first staging loads: the wait that commit(0) needs then covers them too.  Loaded
  // after, they are still "pending" where the chunk loop is entered, and the counter wait the compiler has to place in
  // front of their first use inside the loop is a vmcnt(0) -- which in every later iteration waits out the staging loads
  // of the NEXT chunk right after they were issued
  f16x8 qh[QT][2], ql[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const size_t qo = ((size_t)im * ANP + q0 + (wave * QT + t) * 16 + px) * 512 + head * 64 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qh[t][ks] = *(const f16x8 *)(qkh + qo + 32 * ks);
      ql[t][ks] = *(const f16x8 *)(qkl + qo + 32 * ks);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (ns >= 64) issue(0, std::false_type{});
  else issue(0, std::true_type{});
  commit(0);
  __syncthreads();

  float m[QT], part[QT];
  f32x4 oacc[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -FLT_MAX; part[t] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  // one 64-key chunk.  MASK = false for chunks whose 64 keys are all valid: no per-element bound test and no
  // -FLT_MAX sentinel (the softmax VALU work, not the MFMAs, paces this kernel: ~250 -> ~150 instructions per
  // 48 MFMAs).  The scores arrive in the log2 domain, already divided by sqrt(64): build() folds
  // log2(e) / 8 into the Q projection, so p = 2^(s - m) is one v_exp_f32.
  auto chunk = [&](int ch, auto mask_tag, auto next_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int buf = ch & 1;
    AT_STAMP(0);
    if (ch + 1 < nchunk) issue(ch + 1, next_tag);
    // Fragment reads run AHEAD of the MFMAs that use them (K: one 16-key tile, V^T: two 16-d tiles) and are pinned there
    // with sched_barrier: left to itself the scheduler reloads into the registers the previous MFMAs just read (shortest
    // live ranges), so every group of MFMAs starts with a full LDS round trip that only the second wave of the SIMD can hide.
    // ---- S^T = K Q^T for the 64 keys of this chunk
    f32x4 s[QT][4];
    const _Float16 *kph = kbuf[buf][0] + px * AS + 8 * g;
    const _Float16 *kpl = kbuf[buf][1] + px * AS + 8 * g;
    f16x8 kf[2][2][2];   // [ring][ks][plane]
    auto load_k = [&](int kt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[kt & 1][ks][0] = *(const f16x8 *)(kph + kt * 16 * AS + 32 * ks);
        kf[kt & 1][ks][1] = *(const f16x8 *)(kpl + kt * 16 * AS + 32 * ks);
      }
    };
    load_k(0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt + 1 < 4) load_k(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[QT];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f16x8 ah = kf[kt & 1][ks][0], al = kf[kt & 1][ks][1];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          // the first product of a tile takes the constant 0 as its C operand (no zeroed registers)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[t][ks], ks == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[t][ks], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[t][ks], acc[t], 0, 0, 0);
        }
      }
      const int kb0 = ch * 64 + kt * 16 + 4 * g;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        if (MASK) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = (kb0 + r < ns) ? acc[t][r] : -FLT_MAX;
        }
        s[t][kt] = acc[t];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // V^T fragments of step i = 4 kp + dt (32 keys x 16 d): ring of three, two in flight
    const _Float16 *vph = vbuf[buf][0] + px * VS + 4 * g;
    const _Float16 *vpl = vbuf[buf][1] + px * VS + 4 * g;
    constexpr int VR = 2;   // ring depth (3 = two steps in flight costs 8 more VGPRs: 200, over the co-residency budget)
    f16x4 vf[VR][4];     // [ring][hi keys 0-3 | hi keys 16-19 | lo keys 0-3 | lo keys 16-19]
    auto load_v = [&](int i) {
      const int kp = i >> 2, dt = i & 3, o = (2 * kp) * 16 + dt * 16 * VS;
      vf[i % VR][0] = *(const f16x4 *)(vph + o); vf[i % VR][1] = *(const f16x4 *)(vph + o + 16);
      vf[i % VR][2] = *(const f16x4 *)(vpl + o); vf[i % VR][3] = *(const f16x4 *)(vpl + o + 16);
    };
#pragma unroll
    for (int i = 0; i + 1 < VR; ++i) load_v(i);
    __builtin_amdgcn_sched_barrier(0);
    AT_STAMP(1);
    // ---- online softmax update (per query = per px; the 4 lanes g share it)
    float mn[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      float cm = -FLT_MAX;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cm = fmaxf(cm, s[t][kt][r]);
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      mn[t] = fmaxf(m[t], cm);
      // the running maximum of a query rarely moves after its first chunks: when it did not for any query of this tile
      // (wave-uniform test) the rescale factor is exactly 1 and the 17 multiplications and the exponential are skipped --
      // the same numbers either way
      if (!__all(mn[t] == m[t])) {
        const float alpha = __builtin_amdgcn_exp2f(m[t] - mn[t]);
        m[t] = mn[t];
        part[t] = part[t] * alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[t][dt][r] = oacc[t][dt][r] * alpha;
      }
    }
    AT_STAMP(2);
    // ---- O^T += V^T P^T, 32 keys per k-step
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      f16x8 ph[QT], pl[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          float pv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = 2 * e2 + u;
            const float sv = s[t][2 * kp + (e >> 2)][e & 3];
            float p = __builtin_amdgcn_exp2f(sv - mn[t]);
            if (MASK) p = (sv == -FLT_MAX) ? 0.0f : p;
            part[t] = part[t] + p;
            pv[u] = p;
          }
          split_pair(pv[0], pv[1], hw[e2], lw[e2]);
        }
        ph[t] = __builtin_bit_cast(f16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
        pl[t] = __builtin_bit_cast(f16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int i = 4 * kp + dt;
        __builtin_amdgcn_sched_barrier(0);
        if (i + VR - 1 < 8) load_v(i + VR - 1);
        __builtin_amdgcn_sched_barrier(0);
        f16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ah[e] = vf[i % VR][0][e]; ah[4 + e] = vf[i % VR][1][e]; al[e] = vf[i % VR][2][e]; al[4 + e] = vf[i % VR][3][e]; }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ph[t], oacc[t][dt], 0, 0, 0);
          oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pl[t], oacc[t][dt], 0, 0, 0);
          oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ph[t], oacc[t][dt], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AT_STAMP(3);
    if (ch + 1 < nchunk) commit(buf ^ 1);
    AT_STAMP(4);
    __syncthreads();
    AT_STAMP(5);
  };
  const int nfull = ns >> 6;                          // chunks whose 64 keys are all valid
  for (int ch = 0; ch + 1 < nfull; ++ch) chunk(ch, std::false_type{}, std::false_type{});
  if (nfull > 0) chunk(nfull - 1, std::false_type{}, std::true_type{});   // its successor (if any) is the partial chunk
  if (nfull < nchunk) chunk(nfull, std::true_type{}, std::true_type{});
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = part[t] + __shfl_xor(part[t], 16, 64);
    l = l + __shfl_xor(l, 32, 64);
    const int q = q0 + (wave * QT + t) * 16 + px;
    const size_t oo = ((size_t)im * ANP + q) * 256 + head * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f16x4 h, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = oacc[t][dt][r] / l;
        h[r] = (_Float16)v;
        lo[r] = (_Float16)(v - (float)h[r]);
      }
      *(f16x4 *)(oh + oo + dt * 16) = h;
      *(f16x4 *)(ol + oo + dt * 16) = lo;
    }
  }
}

template <int QT, int NW>
static int launch_attn_h2_t(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                            const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  // (round 5 measured `s_setprio 1` around the MFMA clusters: +-0.5 %, and the run-time switch for it -- two branches inside the
  // pinned schedule of the chunk -- cost 4 us per launch even when off: not kept in any build)
  hipLaunchKernelGGL((attn_h2_kernel<QT, NW>), dim3((ANP / (16 * QT * NW)) * 4 * nimg), dim3(64 * NW), 0, st, qkh, qkl, vth,
                     vtl, counts, cross, oh, ol);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  // URF_ATTN_VARIANT (A/B runs): 0 = 1 tile x 8 waves, 1 = 2 tiles x 4 waves, 2 = 2 tiles x 8 waves.
  // Measured at 8 pairs (18 launches): 1.563 / 1.703 / 1.521 ms.  Default: 2 tiles x 8 waves once that
  // still gives a workgroup per CU (16 images x 4 heads x 4 query blocks = 256), else 1 tile x 8 waves.
  static int forced = -2;
  if (forced == -2) {
    const char *e = urf::exp_env("URF_ATTN_VARIANT");
    forced = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1;
  }
  const int variant = forced >= 0 ? forced : (nimg >= 16 ? 2 : 0);
  if (variant == 1) return launch_attn_h2_t<2, 4>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  if (variant == 2) return launch_attn_h2_t<2, 8>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  return launch_attn_h2_t<1, 8>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
}

}  // namespace urf

#ifdef URF_ATTN_STAMPS
extern "C" int urf_probe_attn_stamps(long long *out) {   // [2 wave groups][64 chunks][8 stamps]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(urf::g_attn_stamps), sizeof(long long) * 2 * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif
