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
#include <utility>
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


#ifdef URF_EXPERIMENTS
typedef float f32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) (a fold, not the unroller:
// the bodies below index register arrays with these constants)
template <typename Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same arithmetic as ONE instruction stream per wave in which every MFMA is followed by the eight cycles of
// softmax VALU work that its sixteen cycles on the matrix pipe leave free on the SIMD's vector issue port
// (MI355X_MICROARCH.md, "an MFMA holds the SIMD's vector issue for 8 of its 16").  attn_h2_kernel runs a chunk as
// S, softmax, P V: while a wave is in its softmax the matrix pipe has only the SIMD's other wave to feed it, and that one
// is in the same phase (one barrier rhythm).  Here the iteration for chunk c is software-pipelined:
//     MFMAs  0..47: S^T(c+1) = K(c+1) Q^T          | VALU  0..39: p = 2^(s(c) - m), hi/lo split, row sums: keys  0..31
//     MFMAs 48..71: O^T += V^T(c) P^T(c), keys 0..31  |      40..79: the same for keys 32..63
//     MFMAs 72..95: the same for keys 32..63       |      80..95: running maximum of s(c+1), rescale test
// with sched_barrier(0) around every MFMA + micro-operation pair, so the order below IS the instruction order.
// Per accumulator the sequence of operations -- and therefore every bit of the output -- is attn_h2_kernel's.
// LDS: K(c+1) and V^T(c) are read in iteration c; K(c+2) and V^T(c+1) are loaded at its start and committed at its end
// (K(c+2) into K(c)'s buffer, V^T(c+1) into V^T(c-1)'s): one barrier per chunk, as before.
template <int QT, int NW>
__global__ void __launch_bounds__(64 * NW) attn_h2_il_kernel(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth,
                                                             const _Float16 *vtl, const int *counts, int cross, _Float16 *oh,
                                                             _Float16 *ol) {
  static_assert(QT == 2 && (NW == 4 || NW == 8), "the schedule below is written for two query tiles per wave");
  constexpr int QB = 16 * QT * NW;
  __shared__ __attribute__((aligned(16))) _Float16 kbuf[2][2][64 * AS];
  __shared__ __attribute__((aligned(16))) _Float16 vbuf[2][2][64 * VS];
  int qb, grp;
  xcd_group_map(blockIdx.x, ANP / QB, (int)gridDim.x / (ANP / QB), qb, grp);
  const int im = grp >> 2, sm = cross ? (im ^ 1) : im;
  const int head = grp & 3;
  const int nq = counts[im], ns = counts[sm];
  const int q0 = qb * QB;
  if (q0 >= nq) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int nchunk = (ns + 63) >> 6, nfull = ns >> 6;

  // staging: half of the waves stage K, the other half V^T; a wave = one plane x 64 rows (NW = 4) or x 32
  // rows (NW = 8), a lane = a 16-byte piece of NP rows (+8u).  Buffer resources: the chunk goes into the scalar offset.
  constexpr int NP = 32 / NW;                                              // 16-byte pieces per thread and chunk
  const int st_v = wave / (NW / 2), st_sp = NW == 8 ? (wave >> 1) & 1 : wave & 1;
  const int st_r0 = (NW == 8 ? (wave & 1) * 32 : 0) + (lane >> 3), st_sj = lane & 7;
  const __amdgpu_buffer_rsrc_t st_rs = st_v
      ? __builtin_amdgcn_make_buffer_rsrc((void *)((st_sp ? vtl : vth) + ((size_t)sm * 256 + head * 64) * ANP), 0, 64u * ANP * 2u, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void *)((st_sp ? qkl : qkh) + ((size_t)sm * ANP) * 512 + 256 + head * 64), 0, (unsigned)ANP * 1024u, 0x00020000);
  const unsigned st_rowb = st_v ? ANP * 2u : 1024u;                      // source row stride in bytes
  const unsigned st_vo = (unsigned)st_r0 * st_rowb + (unsigned)st_sj * 16u;
  const unsigned st_cstep = st_v ? 128u : 64u * 1024u;                   // bytes per chunk along the source
  _Float16 *const st_dst = st_v ? &vbuf[0][st_sp][st_r0 * VS + 8 * st_sj] : &kbuf[0][st_sp][st_r0 * AS + 8 * st_sj];
  const int st_ls = st_v ? VS : AS, st_bufs = st_v ? 2 * 64 * VS : 2 * 64 * AS;
  u32x4 pf[NP];
  auto stage_issue = [&](int c) __attribute__((always_inline)) {       // chunk c of this wave's tensor
#pragma unroll
    for (int u = 0; u < NP; ++u) pf[u] = __builtin_amdgcn_raw_buffer_load_b128(st_rs, st_vo + 8u * u * st_rowb, (unsigned)c * st_cstep, 0);
  };
  auto stage_commit = [&](int c) __attribute__((always_inline)) {
    if (st_v && c >= nfull) {      // the partial chunk of V^T: stale tokens beyond the count are zeroed (0 x NaN would not be)
      const int key0 = c * 64 + 8 * st_sj;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        f16x8 x = __builtin_bit_cast(f16x8, pf[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (key0 + e >= ns) x[e] = (_Float16)0.0f;
        pf[u] = __builtin_bit_cast(u32x4, x);
      }
    }
    _Float16 *dst = st_dst + (c & 1) * st_bufs;
#pragma unroll
    for (int u = 0; u < NP; ++u) *(u32x4 *)(dst + 8 * u * st_ls) = pf[u];
  };

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

  float m[QT], part[QT], mn[QT];
  f32x4 oacc[QT][4];
  f32x4 s[QT][4];               // S^T of the current chunk
  f32x4 sn[QT][4];              // S^T of the next chunk (being produced)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -FLT_MAX; part[t] = 0.0f; mn[t] = -FLT_MAX;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  // ---- the MFMA stream's pieces
  f16x8 kf[3][2];      // K fragments [ring][plane]; stage k8 = 2 kt + ks (0..7): 16 keys x 32 d
  auto load_k = [&](int buf, int k8) __attribute__((always_inline)) {
    const _Float16 *kph = kbuf[buf][0] + px * AS + 8 * g, *kpl = kbuf[buf][1] + px * AS + 8 * g;
    const int kt = k8 >> 1, ks = k8 & 1;
    kf[k8 % 3][0] = *(const f16x8 *)(kph + kt * 16 * AS + 32 * ks);
    kf[k8 % 3][1] = *(const f16x8 *)(kpl + kt * 16 * AS + 32 * ks);
  };
  f32x4 sacc[QT];
  // MFMA i (0..11) of the 16-key tile kt of chunk cn: (ks, t, product)
  auto s_mfma = [&](int cn, int kt, int i, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int ks = i / 6, t = (i % 6) / 3, j = i % 3;
    const f16x8 ah = kf[(2 * kt + ks) % 3][0], al = kf[(2 * kt + ks) % 3][1];
    if (j == 0) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[t][ks], ks == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : sacc[t], 0, 0, 0);
    if (j == 1) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[t][ks], sacc[t], 0, 0, 0);
    if (j == 2) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[t][ks], sacc[t], 0, 0, 0);
    if (i == 11) {
      const int kb0 = cn * 64 + kt * 16 + 4 * g;
#pragma unroll
      for (int tt = 0; tt < QT; ++tt) {
        if (MASK) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[tt][r] = (kb0 + r < ns) ? sacc[tt][r] : -FLT_MAX;
        }
        sn[tt][kt] = sacc[tt];
      }
    }
  };
  f16x8 ph[QT][2], pl[QT][2];   // P(c) as split f16 B operands [tile][32-key step]
  f16x4 vf[3][4];               // V^T fragments [ring][hi keys 0-3 | hi keys 16-19 | lo keys 0-3 | lo keys 16-19]
  auto load_v = [&](int buf, int slot, int kp, int dt) __attribute__((always_inline)) {
    const _Float16 *vph = vbuf[buf][0] + px * VS + 4 * g, *vpl = vbuf[buf][1] + px * VS + 4 * g;
    const int o = (2 * kp) * 16 + dt * 16 * VS;
    vf[slot][0] = *(const f16x4 *)(vph + o); vf[slot][1] = *(const f16x4 *)(vph + o + 16);
    vf[slot][2] = *(const f16x4 *)(vpl + o); vf[slot][3] = *(const f16x4 *)(vpl + o + 16);
  };
  auto pv_mfma = [&](int slot, int kp, int dt, int t, int j) __attribute__((always_inline)) {
    f16x8 ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ah[e] = vf[slot][0][e]; ah[4 + e] = vf[slot][1][e]; al[e] = vf[slot][2][e]; al[4 + e] = vf[slot][3][e]; }
    if (j == 0) oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ph[t][kp], oacc[t][dt], 0, 0, 0);
    if (j == 1) oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, pl[t][kp], oacc[t][dt], 0, 0, 0);
    if (j == 2) oacc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ph[t][kp], oacc[t][dt], 0, 0, 0);
  };
  // ---- the VALU stream's pieces.  Pair q = (kp, t, e2) of the current chunk, micro-operation u (0..4)
  f32x2 xd[16];
  float xp0[16], xp1[16];
  unsigned xh[16], xl[16];
  auto x_pair = [&](int q, int u, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int kp = q >> 3, t = (q >> 2) & 1, e2 = q & 3, e = 2 * e2;
    const int kt = 2 * kp + (e >> 2), r = e & 3;
    if (u == 0) {
      const f32x2 sv = {s[t][kt][r], s[t][kt][r + 1]};
      xd[q] = sv - f32x2{mn[t], mn[t]};            // v_pk_add_f32: the same two differences
    }
    if (u == 1) xp0[q] = __builtin_amdgcn_exp2f(xd[q][0]);
    if (u == 2) xp1[q] = __builtin_amdgcn_exp2f(xd[q][1]);
    if (u == 3) {
      if (MASK) {
        xp0[q] = (s[t][kt][r] == -FLT_MAX) ? 0.0f : xp0[q];
        xp1[q] = (s[t][kt][r + 1] == -FLT_MAX) ? 0.0f : xp1[q];
      }
      part[t] = part[t] + xp0[q];
      part[t] = part[t] + xp1[q];
      const f16x2 h = {(_Float16)xp0[q], (_Float16)xp1[q]};
      xh[q] = __builtin_bit_cast(unsigned, h);
    }
    if (u == 4) {
      unsigned l;
      asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(xp0[q]), "v"(xh[q]));
      asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(xp1[q]), "v"(xh[q]));
      xl[q] = l;
      if (e2 == 3) {
        const int b = q - 3;
        ph[t][kp] = __builtin_bit_cast(f16x8, u32x4{xh[b], xh[b + 1], xh[b + 2], xh[b + 3]});
        pl[t][kp] = __builtin_bit_cast(f16x8, u32x4{xl[b], xl[b + 1], xl[b + 2], xl[b + 3]});
      }
    }
  };
  // running maximum of the next chunk (from sn): micro-operations 0..15 of gaps 80..95; v = 0 .. 15
  float cm[QT], cmx[QT];
  int keep[QT];                 // wave-uniform: the running maximum of no query of the tile moved
  auto x_max = [&](int v) __attribute__((always_inline)) {
    if (v < 8) {                // four gaps per tile: two v_max3 each
      const int t = v >> 2, i = v & 3;
      const float a = fmaxf(fmaxf(sn[t][i][0], sn[t][i][1]), fmaxf(sn[t][i][2], sn[t][i][3]));
      cm[t] = i == 0 ? a : fmaxf(cm[t], a);
      if (i == 3) cmx[t] = __shfl_xor(cm[t], 16, 64);
    } else if (v == 10 || v == 11) {
      const int t = v - 10;
      cm[t] = fmaxf(cm[t], cmx[t]);
      cmx[t] = __shfl_xor(cm[t], 32, 64);
    } else if (v == 14 || v == 15) {
      const int t = v - 14;
      cm[t] = fmaxf(cm[t], cmx[t]);
      mn[t] = fmaxf(m[t], cm[t]);
      keep[t] = __all(mn[t] == m[t]);
    }
  };
  // the rescale of chunk c, between the iterations (rare after a query's first chunks)
  auto rescale = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (!keep[t]) {
        const float alpha = __builtin_amdgcn_exp2f(m[t] - mn[t]);
        m[t] = mn[t];
        part[t] = part[t] * alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[t][dt][r] = oacc[t][dt][r] * alpha;
      }
    }
  };
#define IL_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef URF_ATTN_STAMPS
#define IL_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && c < 64) g_attn_stamps[wave >> 2][c][i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define IL_STAMP(i) do { } while (0)
#endif
  // one iteration: chunk c's softmax and product, chunk c + 1's scores
  // V^T fragment u (use order): u = 0..3 -> (keys 0..31, dt = u), both tiles, six MFMAs; u = 4..11 -> (keys 32..63, dt = (u - 4) & 3) of
  // tile (u - 4) >> 2, three MFMAs.  Ring of three, read two steps ahead.
  auto load_vu = [&](int vb, int u) __attribute__((always_inline)) {
    if (u < 4) load_v(vb, u % 3, 0, u);
    else load_v(vb, u % 3, 1, (u - 4) & 3);
  };
  auto iter = [&](int c, auto next_tag, auto nmask_tag, auto cmask_tag) __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const int kb = (c + 1) & 1, vb = c & 1;
    const int stc = st_v ? c + 1 : c + 2;        // this wave stages V^T(c + 1) or K(c + 2)
    rescale();
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[t][kt] = sn[t][kt];
    if (NEXT) { load_k(kb, 0); load_k(kb, 1); }
    IL_FENCE();
    static_for<96>([&](auto Gc) __attribute__((always_inline)) {
      constexpr int G = decltype(Gc)::value;
      if constexpr (G % 16 == 0) IL_STAMP(G / 16);
      // ---- fragment reads, ahead of their use; the staging loads of the next chunks late enough to keep their registers short-lived
      if constexpr (NEXT && G < 48 && G % 6 == 0 && G / 6 + 2 < 8) load_k(kb, G / 6 + 2);
      if constexpr (G == 32) { if (stc < nchunk) stage_issue(stc); }
      if constexpr (G == 36) load_vu(vb, 0);
      if constexpr (G == 42) load_vu(vb, 1);
      if constexpr (G >= 48 && G < 72 && (G - 48) % 6 == 0) load_vu(vb, (G - 48) / 6 + 2);
      if constexpr (G >= 72 && (G - 72) % 3 == 0 && (G - 72) / 3 + 6 < 12) load_vu(vb, (G - 72) / 3 + 6);
      IL_FENCE();
      // ---- the MFMA of this gap
      if constexpr (G < 48) {
        if constexpr (NEXT) s_mfma(c + 1, G / 12, G % 12, nmask_tag);
      } else if constexpr (G < 72) {
        constexpr int u = (G - 48) / 6, r6 = (G - 48) % 6;               // keys 0..31: dt = u, both tiles
        pv_mfma(u % 3, 0, u, r6 / 3, r6 % 3);
      } else {
        constexpr int u = 4 + (G - 72) / 3;                              // keys 32..63, tile-major
        pv_mfma(u % 3, 1, (u - 4) & 3, (u - 4) >> 2, (G - 72) % 3);
      }
      IL_FENCE();
      // ---- the VALU micro-operation of this gap
      if constexpr (G < 80) x_pair(G / 5, G % 5, cmask_tag);
      else if constexpr (NEXT) x_max(G - 80);
      IL_FENCE();
    });
    IL_STAMP(6);
    if (stc < nchunk) stage_commit(stc);
    __syncthreads();
    IL_STAMP(7);
  };
  const std::true_type T{};
  const std::false_type F{};
  // prologue: K(0), V^T(0), K(1); S(0) and its maximum
  stage_issue(0);
  stage_commit(0);
  if (!st_v && nchunk > 1) {
    stage_issue(1);
    stage_commit(1);
  }
  __syncthreads();
  load_k(0, 0);
  load_k(0, 1);
  if (nfull > 0) {
    static_for<48>([&](auto Gc) __attribute__((always_inline)) {
      constexpr int G = decltype(Gc)::value;
      if constexpr (G % 6 == 0 && G / 6 + 2 < 8) load_k(0, G / 6 + 2);
      IL_FENCE();
      s_mfma(0, G / 12, G % 12, F);
      IL_FENCE();
    });
  } else {
    static_for<48>([&](auto Gc) __attribute__((always_inline)) {
      constexpr int G = decltype(Gc)::value;
      if constexpr (G % 6 == 0 && G / 6 + 2 < 8) load_k(0, G / 6 + 2);
      IL_FENCE();
      s_mfma(0, G / 12, G % 12, T);
      IL_FENCE();
    });
  }
  static_for<16>([&](auto vc) __attribute__((always_inline)) { x_max(decltype(vc)::value); });
  int c = 0;
  for (; c + 1 < nfull; ++c) iter(c, T, F, F);                            // chunk c and chunk c + 1 full: the steady state
  for (; c < nchunk; ++c) {
    if (c + 1 < nchunk) iter(c, T, T, F);                                 // the next chunk is the partial one
    else if (c < nfull) iter(c, F, F, F);
    else iter(c, F, F, T);
  }
#undef IL_FENCE
#undef IL_STAMP
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

#endif  // URF_EXPERIMENTS

template <int QT, int NW>
static int launch_attn_h2_t(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                            const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  // (round 5 measured `s_setprio 1` around the MFMA clusters: +-0.5 %, and the run-time switch for it -- two branches inside the
  // pinned schedule of the chunk -- cost 4 us per launch even when off: not kept in any build)
#ifdef URF_EXPERIMENTS
  // URF_ATTN_IL=1 / 2 (experiments build): the software-pipelined form on eight waves (two per SIMD) / on four waves (one per
  // SIMD, twice the workgroups).  Bit-identical to this kernel and no faster: DESIGN.md section 8, round 5
  static int il = -1;
  if (il < 0) {
    const char *e = urf::exp_env("URF_ATTN_IL");
    il = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0;
  }
  if (NW == 8 && QT == 2 && il) {
    if (il == 1)
      hipLaunchKernelGGL((attn_h2_il_kernel<2, 8>), dim3((ANP / 256) * 4 * nimg), dim3(512), 0, st, qkh, qkl, vth, vtl, counts,
                         cross, oh, ol);
    else
      hipLaunchKernelGGL((attn_h2_il_kernel<2, 4>), dim3((ANP / 128) * 4 * nimg), dim3(256), 0, st, qkh, qkl, vth, vtl, counts,
                         cross, oh, ol);
    URF_HIP(hipGetLastError());
    return 0;
  }
#endif
  hipLaunchKernelGGL((attn_h2_kernel<QT, NW>), dim3((ANP / (16 * QT * NW)) * 4 * nimg), dim3(64 * NW), 0, st, qkh, qkl, vth,
                     vtl, counts, cross, oh, ol);
  URF_HIP(hipGetLastError());
  return 0;
}

int g_attn_small = -1;   // (experiments build, urf_probe_attn_variant: force a variant at run time; -1 = the policy below)
int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st) {
  // URF_ATTN_VARIANT (A/B runs): 0 = 1 tile x 8 waves, 1 = 2 tiles x 4 waves, 2 = 2 tiles x 8 waves.
  // Measured at 8 pairs (18 launches): 1.563 / 1.703 / 1.521 ms.  Default: 2 tiles x 8 waves once that
  // still gives a workgroup per CU (16 images x 4 heads x 4 query blocks = 256), else 1 tile x 8 waves.
  static int forced = -2;
  if (forced == -2) {
    const char *e = urf::exp_env("URF_ATTN_VARIANT");
    forced = (e && e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : -1;
  }
  // Round 6: one or two pairs (the per-call path) run 3 = 1 tile x 4 waves (64 queries per workgroup: 128 / 256 workgroups instead
  // of 64 / 128): 0.59 against 0.68 ms for a pair's 18 launches.  4 = 1 tile x 2 waves (32 queries) is measured and NOT used: every
  // workgroup stages the head's whole K / V^T through its few threads, 2.47 ms.  Same per-wave arithmetic, same bits.
  const int variant = forced >= 0 ? forced : (g_attn_small >= 0 ? g_attn_small : (nimg >= 16 ? 2 : (nimg <= 4 ? 3 : 0)));
  if (variant == 1) return launch_attn_h2_t<2, 4>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  if (variant == 2) return launch_attn_h2_t<2, 8>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  if (variant == 3) return launch_attn_h2_t<1, 4>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  if (variant == 4) return launch_attn_h2_t<1, 2>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
  return launch_attn_h2_t<1, 8>(qkh, qkl, vth, vtl, counts, cross, oh, ol, nimg, st);
}

}  // namespace urf

#ifdef URF_EXPERIMENTS
extern "C" int urf_probe_attn_variant(int v) { urf::g_attn_small = v; return 0; }
#endif
#ifdef URF_ATTN_STAMPS
extern "C" int urf_probe_attn_stamps(long long *out) {   // [2 wave groups][64 chunks][8 stamps]
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(urf::g_attn_stamps), sizeof(long long) * 2 * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif
