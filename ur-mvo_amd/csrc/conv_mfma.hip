// conv_mfma.hip -- 3x3 convolution / linear layer as implicit GEMM on the
// gfx950 fp32 matrix core (v_mfma_f32_16x16x4_f32), exact fp32.
//
// Replaces the convolution / fully-connected nodes of the two TensorRT engines
// the reference executes (src/super_point.cpp:147, src/super_glue.cpp:227); the
// layer list is superpoint/SP/model.py:35-53 and SURVEY.md App. C.
//
// Tile: one 256-thread workgroup (4 waves) = 8x16 output pixels (TAPS==9) or
// 128 rows (TAPS==1) x 64 output channels.  Wave w owns rows 2w,2w+1 of the
// tile (two 16-pixel MFMA column blocks) x 4 blocks of 16 channels = 8
// accumulators.  M = output channel (A = weights from LDS), N = pixel (B =
// activations from LDS): a lane ends up with 4 consecutive channels of one
// pixel -> one 16-byte NHWC store.
//
// LDS: input tile [(8+2)*(16+2)][66] f32 (stride 66: the 32 lanes of a half
// wave hit 32 distinct banks), weights [64][80] f32 (stride 80 likewise).
// Accumulation order: acc=bias; for 64-ch chunk; for tap; for c.
#include "urf_common.h"
#include <stdlib.h>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef URF_CONV32_STAMPS   // diagnostic build only (make BUILD=build_st OUT=../liburf_front_st.so EXTRA=-DURF_CONV32_STAMPS; tools/gpu_conv32_stamps.py)
__device__ long long g_conv32_stamps[2][10];   // [8], [9]: s_memrealtime (100 MHz) at the tile's entry and exit -> the shader clock
#define C32_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define C32_ON(POOL_, FUSE_) ((POOL_) == (URF_CONV32_STAMPS == 1) && (FUSE_) == (URF_CONV32_STAMPS == 1) && bx == 300 && by == 0 && bz == 3 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 2))
#else
#define C32_NOW() 0ll
#define C32_ON(POOL_, FUSE_) false
#endif

constexpr int TH = 8, TW = 16;
constexpr int IN_STRIDE = 66;
constexpr int W_STRIDE = 80;

// MB = 16-channel output blocks per workgroup: 4 (the tile above), or 1 for the redo pass of the guarded fast mode, whose
// launches hold a few tiles each and are bound by the length of one wave's MFMA chain -- a quarter of the channels per
// workgroup, four times the workgroups, the weights fetched two taps ahead.  The chain of every output is the same.
// WDMA (round 4; the full-frame 3x3 launches of the exact mode): a (chunk, tap)'s 64 x 64 weights arrive by LDS-DMA
// (global_load_lds_dwordx4) in two alternating 16-KiB stages instead of through registers: no weight VGPRs, no LDS writes by
// the waves, ONE barrier per tap instead of two.  The stage is unpadded (a DMA piece is 1 KiB of consecutive LDS: 4 rows of 64
// floats); the bank spread the padded layout got from its stride of 80 comes from a swizzle instead -- row k keeps its four
// 16-float blocks in the order block ^ (k & 3), applied to the SOURCE address of the DMA and to the fragment read.  Every
// output's fma chain is unchanged.
template <int TAPS, bool POOL, bool FUSE1A, int MB, bool WDMA>
__device__ __forceinline__ void conv_mfma_tile(const ConvArgs &a, float *smem, const int bx, const int by, const int bz) {
  constexpr int PH = (TAPS == 9) ? TH + 2 : TH;
  constexpr int PW = (TAPS == 9) ? TW + 2 : TW;
  float *in_tile = smem;                          // [PH*PW][IN_STRIDE]
  float *w_tile = smem + PH * PW * IN_STRIDE;     // [64][W_STRIDE]  (offset is a multiple of 4 floats); WDMA: [2][64][64]
  float *patch = w_tile + (WDMA ? 2 * 64 * 64 : 64 * W_STRIDE);   // FUSE1A: [12][20]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  [[maybe_unused]] const long long st_entry = C32_NOW();
#ifdef URF_CONV32_STAMPS
  if (WDMA && C32_ON(POOL, FUSE1A)) g_conv32_stamps[(threadIdx.x >> 6) != 0][8] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  const int b = bz;
  if (a.gate && b >= a.gate[0]) return;
  const int cout_base = by * (16 * MB);

  int y0 = 0, x0 = 0;
  if (TAPS == 9) {
    const int tiles_x = (a.W + TW - 1) / TW;
    y0 = (bx / tiles_x) * TH;
    x0 = (bx % tiles_x) * TW;
  } else {
    x0 = bx * (TH * TW);
    if (a.counts && x0 >= a.counts[b]) return;  // rows beyond this item's count
  }
  if (a.gate && a.t_scale != 0) {
    // a redo slot that only needs the scores of a few cells: is this tile within reach of one of them?  (workgroup-uniform)
    const int nt = a.gate[kGateMode + b];
    if (nt >= 0) {
      if (a.t_scale < 0) return;
      bool needed = false;
      for (int t = 0; t < nt; ++t) {
        const int cell = a.gate[kGateTargets + kAmbMax * b + t];
        if (TAPS == 9) {
          const int fy = (cell / a.t_wc) * a.t_scale, fx = (cell % a.t_wc) * a.t_scale;
          needed = needed || (y0 + TH > fy - a.t_rad && y0 < fy + a.t_scale + a.t_rad && x0 + TW > fx - a.t_rad && x0 < fx + a.t_scale + a.t_rad);
        } else {
          needed = needed || (cell >= x0 && cell < x0 + TH * TW);
        }
      }
      if (!needed || cout_base >= 256) return;   // (the score head is channels 0..255 of the fused Pa || Da layer)
    }
  }

  f32x4 acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cout_base + m * 16 + 4 * g + r;
      bv[r] = co < a.Cout ? a.bias[co] : 0.0f;
    }
    acc[m][0] = bv;
    acc[m][1] = bv;
  }

  // B-operand base addresses (pixel part) for this lane's two pixel blocks
  int bpix[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (TAPS == 9) bpix[r] = ((2 * wave + r) * PW + px) * IN_STRIDE + g;
    else bpix[r] = ((2 * wave + r) * TW + px) * IN_STRIDE + g;
  }
  const int aoff = g * W_STRIDE + px;

  const int nchunks = FUSE1A ? 1 : (a.Cin + 63) / 64;

  // ---- register prefetch (issue early, commit to LDS late): weights of the next
  // (chunk, tap) stage and, for TAPS==1, the next chunk's input rows fly while
  // the MFMAs of the current stage run.
  f32x4 wpf[MB], wpf2[1];   // (wpf2: the second request in flight, MB == 1)
  auto issue_w = [&](int ch, int tap, f32x4 *wr) {
    const int c0 = ch * 64;
    const int kc = FUSE1A ? 64 : ((a.Cin - c0) < 64 ? (a.Cin - c0) : 64);
    const float *wsrc = a.w + ((size_t)tap * a.Cin + c0) * a.Cout;
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = tid + 256 * u;
      const int k = i / (4 * MB), j = i % (4 * MB);
      const int co = cout_base + 4 * j;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (k < kc) {
        const float *sp = wsrc + (size_t)k * a.Cout + co;
        if (co + 3 < a.Cout && ((a.Cout & 3) == 0)) {
          v = *(const f32x4 *)sp;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (co + r < a.Cout) ? sp[r] : 0.0f;
        }
      }
      wr[u] = v;
    }
  };
  auto commit_w = [&](const f32x4 *wr) {
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const int i = tid + 256 * u;
      *(f32x4 *)(w_tile + (i / (4 * MB)) * W_STRIDE + 4 * (i % (4 * MB))) = wr[u];
    }
  };
  f32x4 ipf[8];  // TAPS==1 only: 128 rows x (kc/4) float4
  auto issue_in = [&](int ch) {
    const int c0 = ch * 64;
    const int kc = (a.Cin - c0) < 64 ? (a.Cin - c0) : 64;
    const int vpp = kc >> 2;                    // power of two (kc in {4,8,16,32,64})
    const int vsh = 31 - __builtin_clz(vpp);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 256 * u;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (i < TH * TW * vpp) {
        const int p = i >> vsh, j = i & (vpp - 1);
        const int cc = c0 + 4 * j, row = x0 + p;
        if (row < a.W) {
          const float *src;
          if (a.in2 && cc >= a.Cin1)
            src = a.in2 + (size_t)b * a.in2_bstride + (size_t)row * a.in2_ld + a.in2_coff + (cc - a.Cin1);
          else
            src = (const float *)a.in + (size_t)b * a.in_bstride + (size_t)row * a.in_ld + a.in_coff + cc;
          v = *(const f32x4 *)src;
        }
      }
      ipf[u] = v;
    }
  };
  auto commit_in = [&](int ch) {
    const int c0 = ch * 64;
    const int kc = (a.Cin - c0) < 64 ? (a.Cin - c0) : 64;
    const int vpp = kc >> 2;
    const int vsh = 31 - __builtin_clz(vpp);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 256 * u;
      if (i < TH * TW * vpp) {
        const int p = i >> vsh, j = i & (vpp - 1);
        float *dst = in_tile + p * IN_STRIDE + 4 * j;
        *(float2 *)dst = make_float2(ipf[u][0], ipf[u][1]);
        *(float2 *)(dst + 2) = make_float2(ipf[u][2], ipf[u][3]);
      }
    }
  };

  auto stage_input = [&](int ch) {
    const int c0 = ch * 64;
    const int kc = FUSE1A ? 64 : ((a.Cin - c0) < 64 ? (a.Cin - c0) : 64);
    (void)c0; (void)kc;
    if (FUSE1A) {
      // ---- fused conv1a: u8 patch -> f32 -> 3x3 conv (VALU fma chain) -> relu
      const uint8_t *img = (const uint8_t *)a.in + (size_t)b * a.in_bstride;
      for (int i = tid; i < 12 * 20; i += 256) {
        const int yy = y0 - 2 + i / 20, xx = x0 - 2 + i % 20;
        float v = 0.0f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) v = a.lut[img[(size_t)yy * a.W + xx]];
        patch[i] = v;
      }
      // conv1a on the fp32 matrix core (see h2conv.hip): the nine taps, padded with three zero taps, are the K dimension
      // of v_mfma_f32_16x16x4_f32, whose accumulation is the ordered chain acc = fma(a_k, b_k, acc) -- the same bits as the
      // VALU chain "bias, then the taps in raster order" of the oracle (a zero tap adds +0; relu maps -0 and +0 alike).
      // 180 pixels = 12 groups of 16 x four 16-cout tiles: 36 MFMAs per wave instead of 405 chained FMAs per thread.
      f32x4 bias4[4];
      float wa[4][3];                                  // A fragments: W[16 mt + px][4 ks + g]
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[mt][r] = a.b1a[16 * mt + 4 * g + r];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int t = 4 * ks + g;
          wa[mt][ks] = t < 9 ? a.w1a[t * 64 + 16 * mt + px] : 0.0f;
        }
      }
      __syncthreads();
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int p = (wave + 4 * gi) * 16 + px;         // this lane's pixel of the group (>= PH * PW: no pixel)
        const int pc = p < PH * PW ? p : PH * PW - 1;
        const int py = pc / PW, pxx = pc % PW;
        float xb[3];                                     // B fragments: X[4 ks + g][pixel]
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int t = 4 * ks + g;
          const int tt = t < 9 ? t : 8;
          const float x = patch[(py + tt / 3) * 20 + pxx + tt % 3];
          xb[ks] = t < 9 ? x : 0.0f;
        }
        const int yy = y0 - 1 + py, xx = x0 - 1 + pxx;
        const bool inb = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 d = bias4[mt];
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][ks], xb[ks], d, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = d[r] > 0.0f ? d[r] : 0.0f;
            d[r] = inb ? v : 0.0f;
          }
          if (p < PH * PW) {
            float *dst = in_tile + p * IN_STRIDE + 16 * mt + 4 * g;   // 8-byte aligned
            *(float2 *)dst = make_float2(d[0], d[1]);
            *(float2 *)(dst + 2) = make_float2(d[2], d[3]);
          }
        }
      }
    } else if (TAPS == 9) {
      // ---- stage the input tile chunk: each pixel's kc channels are contiguous
      // (the previous chunk's last barrier already freed in_tile)
      const int vec_per_pix = kc >> 2;
      const int vsh9 = 31 - __builtin_clz(vec_per_pix);
      const int total = PH * PW * vec_per_pix;
      if (kc == 64) {
        // every chunk of the SuperPoint convolutions: 180 pixels x 16 float4 = 2880 pieces, 12 per thread (the last one
        // partial), in two batches of six with ALL loads of a batch in flight before its first LDS write -- from clamped,
        // always valid offsets, the zero padding applied afterwards.  The generic loop below is load, full wait, write per
        // piece: twelve dependent memory round trips per chunk.
        constexpr int NPIECE = PH * PW * 16, NST = (NPIECE + 255) / 256, NB = NST % 6 == 0 ? 6 : (NST % 4 == 0 ? 4 : 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((const float *)a.in + (size_t)b * a.in_bstride), 0, (unsigned)((size_t)a.H * a.W * a.in_ld * sizeof(float)), 0x00020000);
        const unsigned coff = (unsigned)((a.in_coff + c0) * (int)sizeof(float));
#pragma unroll
        for (int u0 = 0; u0 < NST; u0 += NB) {
          f32x4 sv[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * (u0 + u) < NPIECE ? tid + 256 * (u0 + u) : NPIECE - 1;
            const int p = i >> 4, j = i & 15;
            const int yy = y0 - 1 + p / PW, xx = x0 - 1 + p % PW;
            const bool inb = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const int yc = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= a.W ? a.W - 1 : xx);
            const unsigned off = (unsigned)(((yc * a.W + xc) * a.in_ld + 4 * j) * (int)sizeof(float));
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, coff, 0));
            sv[u] = inb ? v : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * (u0 + u);
            if (i < NPIECE) {
              float *dst = in_tile + (i >> 4) * IN_STRIDE + 4 * (i & 15);  // 8-byte aligned
              *(float2 *)dst = make_float2(sv[u][0], sv[u][1]);
              *(float2 *)(dst + 2) = make_float2(sv[u][2], sv[u][3]);
            }
          }
        }
      } else
      for (int i = tid; i < total; i += 256) {
        const int p = i >> vsh9, j = i & (vec_per_pix - 1);
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        const int cc = c0 + 4 * j;
        const int yy = y0 - 1 + p / PW, xx = x0 - 1 + p % PW;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          const float *src = (const float *)a.in + (size_t)b * a.in_bstride +
                             ((size_t)yy * a.W + xx) * a.in_ld + a.in_coff + cc;
          v = *(const f32x4 *)src;
        }
        float *dst = in_tile + p * IN_STRIDE + 4 * j;  // 8-byte aligned
        *(float2 *)dst = make_float2(v[0], v[1]);
        *(float2 *)(dst + 2) = make_float2(v[2], v[3]);
      }
    } else {
      commit_in(ch);
    }

  };
  auto compute_tap_dma = [&](int tap, int stage) {
    // WDMA: fragment reads from the swizzled stage: lane (px, g) of block m reads row k + g, physical block m ^ g
    const int toff = ((tap / 3) * PW + (tap % 3)) * IN_STRIDE;
    const float *bp0 = in_tile + bpix[0] + toff;
    const float *bp1 = in_tile + bpix[1] + toff;
    const float *wb = w_tile + stage * (64 * 64) + g * 64 + px;
    const float *apm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) apm[m] = wb + 16 * (m ^ g);
    float pa[2][4], pb[2][2];
#define URF_LOADD(set, k)                                                    \
  {                                                                          \
    pb[set][0] = bp0[k]; pb[set][1] = bp1[k];                                \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) pa[set][m] = apm[m][(k) * 64]; \
  }
    URF_LOADD(0, 0)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < 16) URF_LOADD(cur ^ 1, 4 * (ks + 1))
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][r2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][m], pb[cur][r2], acc[m][r2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef URF_LOADD
  };
  auto compute_tap = [&](int tap, int kc) {
      const int toff = (TAPS == 9) ? ((tap / 3) * PW + (tap % 3)) * IN_STRIDE : 0;
      const float *bp0 = in_tile + bpix[0] + toff;
      const float *bp1 = in_tile + bpix[1] + toff;
      const float *ap = w_tile + aoff;
#define URF_KSTEP(k)                                                                              \
  {                                                                                               \
    const float b0 = bp0[k], b1 = bp1[k];                                                         \
    float am[MB];                                                                                 \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) am[m] = ap[(k) * W_STRIDE + 16 * m];           \
    _Pragma("unroll") for (int m = 0; m < MB; ++m)                                                \
      acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m], b0, acc[m][0], 0, 0, 0);            \
    _Pragma("unroll") for (int m = 0; m < MB; ++m)                                                \
      acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m], b1, acc[m][1], 0, 0, 0);            \
  }
      if (kc == 64) {
        // fast path: fully unrolled and software-pipelined by hand -- the LDS
        // operand reads of k-step k+1 are issued before the 8 MFMAs of k-step k
        // (two register sets), so their latency hides behind 256 MFMA cycles.
        float pa[2][MB], pb[2][2];
#define URF_LOAD(set, k)                                                               \
  {                                                                                    \
    pb[set][0] = bp0[k]; pb[set][1] = bp1[k];                                          \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) pa[set][m] = ap[(k) * W_STRIDE + 16 * m]; \
  }
        URF_LOAD(0, 0)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int cur = ks & 1;
          if (ks + 1 < 16) URF_LOAD(cur ^ 1, 4 * (ks + 1))
          __builtin_amdgcn_sched_barrier(0);  // next step's DS reads are issued before this step's MFMAs
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int m = 0; m < MB; ++m)
              acc[m][r2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][m], pb[cur][r2], acc[m][r2], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef URF_LOAD
      } else {
        for (int k = 0; k < kc; k += 4) URF_KSTEP(k)
      }
#undef URF_KSTEP
  };

  if constexpr (WDMA) {
    static_assert(MB == 4 && TAPS == 9, "WDMA is the full-tile 3x3 path");
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    // a stage = 16 pieces of 1 KiB (4 weight rows each); wave w issues pieces 4 w .. 4 w + 3.  Lane L of a piece: row r = L >> 4,
    // 16-byte slot s = L & 15 = physical block s >> 2 -> logical block (s >> 2) ^ r (the piece's first row is a multiple of 4)
    const int dr = lane >> 4, dsl = lane & 15;
    const int dcol = cout_base + 16 * ((dsl >> 2) ^ dr) + 4 * (dsl & 3);
    auto dma_w = [&](int ch, int tap, int stage) {
      const int cin_all = FUSE1A ? 64 : a.Cin;
      const float *wsrc = a.w + ((size_t)tap * cin_all + ch * 64) * a.Cout + dcol;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = 4 * wave + u;
        __builtin_amdgcn_global_load_lds((gbl_void *)(wsrc + (size_t)(4 * i + dr) * a.Cout),
                                         (lds_void *)(w_tile + stage * (64 * 64) + i * 256), 16, 0, 0);
      }
    };
    [[maybe_unused]] const bool st_on = C32_ON(POOL, FUSE1A);
    [[maybe_unused]] long long st_t0 = C32_NOW(), st_stage = 0, st_wait = 0, st_sync = 0, st_mma = 0;
    dma_w(0, 0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
      if (ch > 0) __syncthreads();   // slower waves may still read the input tile for the previous chunk's last tap
      const long long q0 = C32_NOW();
      stage_input(ch);
      st_stage += C32_NOW() - q0;
      for (int tap = 0; tap < TAPS; ++tap) {
        const int s2 = ch * TAPS + tap;
        const long long w0 = C32_NOW();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the stage have landed
        const long long w1 = C32_NOW();
        __syncthreads();                                    // everybody's have (and the input tile), and everybody is done with the other stage
        const long long w2 = C32_NOW();
        {
          const int nt = tap + 1 < TAPS ? tap + 1 : 0, nc = tap + 1 < TAPS ? ch : ch + 1;
          if (nc < nchunks) dma_w(nc, nt, (s2 + 1) & 1);
        }
        compute_tap_dma(tap, s2 & 1);
        st_wait += w1 - w0; st_sync += w2 - w1; st_mma += C32_NOW() - w2;
      }
    }
#ifdef URF_CONV32_STAMPS
    if (st_on) {
      long long *o = g_conv32_stamps[(threadIdx.x >> 6) != 0];
      o[0] = st_t0; o[1] = st_stage; o[2] = st_wait; o[3] = st_sync; o[4] = st_mma; o[5] = C32_NOW(); o[6] = st_entry;
    }
#endif
  } else if constexpr (MB == 4) {
    issue_w(0, 0, wpf);
    if (TAPS == 1) issue_in(0);
    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch * 64;
      const int kc = FUSE1A ? 64 : ((a.Cin - c0) < 64 ? (a.Cin - c0) : 64);
      stage_input(ch);
      for (int tap = 0; tap < TAPS; ++tap) {
        commit_w(wpf);
        __syncthreads();  // in_tile + w_tile of this stage visible
        {                 // prefetch the next stage
          const bool last_tap = (tap + 1 == TAPS);
          if (!last_tap) issue_w(ch, tap + 1, wpf);
          else if (ch + 1 < nchunks) {
            issue_w(ch + 1, 0, wpf);
            if (TAPS == 1) issue_in(ch + 1);
          }
        }
        compute_tap(tap, kc);
        __syncthreads();  // every wave is done with w_tile (and in_tile after the last tap)
      }
    }
  } else {
    // stages (chunk, tap) flattened; the weights of stage s + 2 are requested when stage s has been committed to LDS, so two
    // requests are always in flight (one float4 per thread each) beside a stage that is only 32 MFMAs per wave long
    const int S = nchunks * TAPS;
    issue_w(0, 0, wpf);
    if (S > 1) issue_w(1 / TAPS, 1 % TAPS, wpf2);
    if (TAPS == 1) issue_in(0);
    auto stage = [&](int s2, f32x4 *wr) {
      const int ch = s2 / TAPS, tap = s2 % TAPS;
      const int c0 = ch * 64;
      const int kc = FUSE1A ? 64 : ((a.Cin - c0) < 64 ? (a.Cin - c0) : 64);
      if (tap == 0) stage_input(ch);
      commit_w(wr);
      __syncthreads();
      if (s2 + 2 < S) issue_w((s2 + 2) / TAPS, (s2 + 2) % TAPS, wr);
      if (TAPS == 1 && ch + 1 < nchunks) issue_in(ch + 1);   // (linear layers: the next chunk's rows fly beside this chunk's MFMAs)
      compute_tap(tap, kc);
      __syncthreads();
    };
    for (int s2 = 0; s2 < S; s2 += 2) {
      stage(s2, wpf);
      if (s2 + 1 < S) stage(s2 + 1, wpf2);
    }
  }

  // ---------------------------------------------------------------- epilogue
  float *outb = a.out + (size_t)b * a.out_bstride;
  if (POOL) {
    // 2x2 max pool (model.py:34): rows 2w,2w+1 live in this wave, the x pair in
    // lanes px, px^1.  max is exact and order independent; relu commutes.
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int oy = (y0 >> 1) + wave, ox = (x0 + px) >> 1;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = fmaxf(acc[m][0][r], acc[m][1][r]);
        t = fmaxf(t, __shfl_xor(t, 1, 64));
        if (a.relu) t = t > 0.0f ? t : 0.0f;
        v[r] = t;
      }
      const int co = cout_base + m * 16 + 4 * g;
      if ((px & 1) == 0 && oy < Ho && ox < Wo && co < a.Cout)
        *(f32x4 *)(outb + ((size_t)oy * Wo + ox) * a.out_ld + a.out_coff + co) = v;
    }
  } else {
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      size_t pix;
      bool ok;
      if (TAPS == 9) {
        const int yy = y0 + 2 * wave + r2, xx = x0 + px;
        ok = yy < a.H && xx < a.W;
        pix = (size_t)yy * a.W + xx;
      } else {
        const int row = x0 + (2 * wave + r2) * TW + px;
        ok = row < a.W;
        pix = row;
      }
      if (!ok) continue;
      const float *resb = a.res ? a.res + (size_t)b * a.res_bstride + pix * a.res_ld + a.res_coff : nullptr;
      float *op = outb + pix * a.out_ld + a.out_coff;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int co = cout_base + m * 16 + 4 * g;
        f32x4 v = acc[m][r2];
        if (a.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
        }
        if (co + 3 < a.Cout) {
          if (resb) {
            const f32x4 rv = *(const f32x4 *)(resb + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rv[r] + v[r];
          }
          *(f32x4 *)(op + co) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < a.Cout) op[co + r] = resb ? resb[co + r] + v[r] : v[r];
        }
      }
    }
  }
#ifdef URF_CONV32_STAMPS
  if (WDMA && C32_ON(POOL, FUSE1A)) { g_conv32_stamps[(threadIdx.x >> 6) != 0][7] = C32_NOW(); g_conv32_stamps[(threadIdx.x >> 6) != 0][9] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
}

template <int TAPS, bool POOL, bool FUSE1A, int MB = 4, bool WDMA = false>
__global__ void __launch_bounds__(256, 2) conv_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (round 4, measured and not kept: a persistent form -- two workgroups per CU walking the launch's tiles -- and the same with
  // the CU's second workgroup started half a tile late, picked by HW_ID.WAVE_ID: 1548 vs 1547 us for conv1 either way.  The
  // stamps of tools/gpu_conv32_stamps.py say why there is nothing to win from slot turnover or phase: the chip holds 2.13 GHz
  // inside this kernel, not the 2.4 GHz of the nominal peak, so 119 TFLOP/s is 85 % of what the matrix pipe delivers here.)
  conv_mfma_tile<TAPS, POOL, FUSE1A, MB, WDMA>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the exact linear layer with BOTH operands by LDS-DMA (the redo engine of the strict mode and every other launch
// that cannot fill the chip: one or two pairs = 2 000 - 4 000 rows).  The register-staged tile above (TAPS == 1) moves a chunk's
// 128 rows and 64 x 64 weights global -> VGPR -> LDS with one chunk of look-ahead and two barriers per chunk; on a grid of 64 - 192
// workgroups nothing else is resident to hide that, and a launch whose fma chains need 7 - 14 us of matrix-pipe time takes 20 - 28.
// Here: 64 rows x 64 channels per workgroup (twice the workgroups), wave w = rows 16 w .. 16 w + 15 x four 16-channel blocks,
// S stages of [X 64 x 64 | W 64 x 64] floats filled by global_load_lds (eight 1-KiB pieces per wave and chunk: its own four row
// pieces and four weight pieces), ONE barrier per chunk, `s_waitcnt vmcnt(8 (S - 2))` retires exactly the chunk about to be read.
// Unpadded stages; bank spread by swizzles applied to the DMA's SOURCE address and to the fragment read: weight row k keeps its
// four 16-float blocks in the order block ^ (k & 3) (as the WDMA convolution), activation row r keeps its sixteen 4-float
// blocks in the order block ^ (r & 15) -- lane (px, g) of k-step s reads float g of block s ^ px: 64 distinct banks.
// Every output is the same chain as in conv_mfma_tile: acc = bias; for chunk; for k-step: acc = fma(w, x, acc) on the fp32 matrix
// core with A = weights, B = activations.  Bit-identical to the register-staged tile (tests/test_gpu_fullsize.py).
template <int S>
__global__ void __launch_bounds__(256, 2) linear_dma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  constexpr int ST = 2 * 64 * 64;               // floats per stage: X at 0, W at 4096
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, cout_base = blockIdx.y * 64, x0 = blockIdx.x * 64;
  if (a.counts && x0 >= a.counts[b]) return;    // rows beyond this item's count
  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = a.bias[cout_base + m * 16 + 4 * g + r];

  // DMA roles.  Activations: piece i = 4 wave + u = tile rows 4 i .. 4 i + 3; lane L: row 4 i + (L >> 4), physical block L & 15
  const float *xsrc[4], *xsrc2[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 4 * (4 * wave + u) + (lane >> 4);
    int row = x0 + r;
    if (row > a.W - 1) row = a.W - 1;           // (past the end: any valid row; its outputs are never stored)
    const int lb = (lane & 15) ^ (r & 15);
    xsrc[u] = (const float *)a.in + (size_t)b * a.in_bstride + (size_t)row * a.in_ld + a.in_coff + 4 * lb;
    xsrc2[u] = a.in2 ? a.in2 + (size_t)b * a.in2_bstride + (size_t)row * a.in2_ld + a.in2_coff + 4 * lb : xsrc[u];
  }
  // weights: piece i = rows k = 4 i .. 4 i + 3 of the chunk; lane L: row 4 i + (L >> 4), 16-byte slot L & 15 = physical block
  // (L & 15) >> 2 -> logical block ((L & 15) >> 2) ^ (L >> 4)
  const int dr = lane >> 4, dsl = lane & 15;
  const float *wbase = a.w + cout_base + 16 * ((dsl >> 2) ^ dr) + 4 * (dsl & 3);
  auto issue = [&](int ch) {
    float *st = smem + (ch % S) * ST;
    const int c0 = ch * 64;
    const bool second = a.in2 && c0 >= a.Cin1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *p = second ? xsrc2[u] + (c0 - a.Cin1) : xsrc[u] + c0;
      __builtin_amdgcn_global_load_lds((gbl_void *)p, (lds_void *)(st + (4 * wave + u) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * wave + u;
      __builtin_amdgcn_global_load_lds((gbl_void *)(wbase + (size_t)(c0 + 4 * i + dr) * a.Cout), (lds_void *)(st + 4096 + i * 256), 16, 0, 0);
    }
  };
  const int nch = a.Cin / 64;
  constexpr int LOOK = S - 1;
  for (int c = 0; c < LOOK && c < nch; ++c) issue(c);
  const int xoff = (16 * wave + px) * 64 + g;   // + 4 (ks ^ px)
  const int woff = 4096 + g * 64 + px;          // + 16 (m ^ g) + 256 ks   (row 4 ks + g)
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + LOOK <= nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (LOOK - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // everybody's pieces of chunk ch are in; everybody is done with the stage refilled next
    if (ch + LOOK < nch) issue(ch + LOOK);
    const float *st = smem + (ch % S) * ST;
    const float *xp = st + xoff;
    const float *apm[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) apm[m] = st + woff + 16 * (m ^ g);
    float pa[2][4], pb[2];
#define URF_LDLOAD(set, ks)                                                         \
  {                                                                                 \
    pb[set] = xp[4 * ((ks) ^ px)];                                                  \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) pa[set][m] = apm[m][(ks) * 256];  \
  }
    URF_LDLOAD(0, 0)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < 16) URF_LDLOAD(cur ^ 1, ks + 1)
      __builtin_amdgcn_sched_barrier(0);        // the next step's DS reads are issued before this step's MFMAs
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][m], pb[cur], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef URF_LDLOAD
  }
  // epilogue: lane = 4 consecutive channels of one row
  const int row = x0 + 16 * wave + px;
  if (row >= a.W) return;
  const float *resb = a.res ? a.res + (size_t)b * a.res_bstride + (size_t)row * a.res_ld + a.res_coff : nullptr;
  float *op = a.out + (size_t)b * a.out_bstride + (size_t)row * a.out_ld + a.out_coff;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int co = cout_base + m * 16 + 4 * g;
    f32x4 v = acc[m];
    if (a.relu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
    }
    if (resb) {
      const f32x4 rv = *(const f32x4 *)(resb + co);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rv[r] + v[r];
    }
    *(f32x4 *)(op + co) = v;
  }
}

int g_linear_dma = -1;   // -1: not read yet; 0 = never, 1 = the policy of launch_conv (default), 2 / 3 = that many stages for every eligible launch

static size_t conv_lds_bytes(int taps, bool fuse, bool wdma = false) {
  const int PH = taps == 9 ? TH + 2 : TH, PW = taps == 9 ? TW + 2 : TW;
  return sizeof(float) * ((size_t)PH * PW * IN_STRIDE + (wdma ? 2 * 64 * 64 : 64 * W_STRIDE) + (fuse ? 12 * 20 : 0));
}

// host launcher.  batch = grid.z
int launch_conv(const ConvArgs &a, int taps, bool pool, bool fuse1a, int batch, hipStream_t st) {
  dim3 grid, block(256);
  if (taps == 9) {
    grid.x = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
  } else {
    grid.x = (a.W + TH * TW - 1) / (TH * TW);
  }
  // the redo pass of the guarded fast mode (target-gated launches: a handful of live tiles): 16 output channels per workgroup
  static const bool split_env = [] { const char *e = urf::exp_env("URF_GUARD_SPLIT"); return !e || atoi(e) != 0; }();
  const bool split = split_env && taps == 9 && !fuse1a && a.gate && a.t_scale > 0;   // (conv1: its fused first layer would be redone per quarter)
  // a.narrow (the linear layers of ONE pair on a handle of its own: the per-pair host API): 64 - 192 full tiles for 256 CUs, each as
  // long as one wave's chain of 8 accumulators x Cin / 4 MFMAs -- a quarter of the channels per workgroup instead: four times the
  // workgroups, a quarter of the chain; one pair alone 4.41 -> 3.87 ms.  Same fma chain per output.  NOT for the redo engine of a
  // strict-parity handle, whose launches run beside three saturated streams: there the four-fold staging of the rows is chip time
  // the other streams lose (1044 against 1066 frames/s), and not for two pairs (3.37 against 3.25 ms alone).
  // the DMA-staged linear tile (linear_dma_kernel): every layer whose K and N are whole 64-blocks (it replaces the `narrow` form
  // of the per-pair host API as well).  URF_LINEAR_DMA (experiments build): 0 never, 1 / 2 = two stages (default), 3 = three stages
  if (g_linear_dma < 0) { const char *e = urf::exp_env("URF_LINEAR_DMA"); g_linear_dma = e ? atoi(e) : 1; }
  if (taps == 1 && !a.gate && g_linear_dma && (a.Cin % 64) == 0 && (a.Cout % 64) == 0 && (a.in_ld % 4) == 0 && (a.in_coff % 4) == 0 &&
      (!a.in2 || ((a.Cin1 % 64) == 0 && (a.in2_ld % 4) == 0 && (a.in2_coff % 4) == 0))) {
    // (two stages = 64 KB, two workgroups per CU, for every eligible launch: measured +1.2 % in the exact mode at batch 8 and in the
    // strict mode at a 40 % flag rate, a lone pair's linear layers 1.47 -> 1.13 ms; three stages lose 3.5 % at batch 8)
    const int stages = g_linear_dma >= 3 ? 3 : 2;
    if (stages) {
      static DeviceOnce attr_ld;
      if (attr_ld.need()) {
        URF_HIP(hipFuncSetAttribute((const void *)linear_dma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        URF_HIP(hipFuncSetAttribute((const void *)linear_dma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_ld.mark();
      }
      const dim3 g64((a.W + 63) / 64, a.Cout / 64, batch);
      if (stages == 3) hipLaunchKernelGGL((linear_dma_kernel<3>), g64, dim3(256), 3 * 32 * 1024, st, a);
      else hipLaunchKernelGGL((linear_dma_kernel<2>), g64, dim3(256), 2 * 32 * 1024, st, a);
      URF_HIP(hipGetLastError());
      return 0;
    }
  }
  const bool narrow = taps == 1 && a.narrow && !a.gate && (a.Cout % 16) == 0 && a.Cout >= 64;
  grid.y = (split || narrow) ? (a.Cout + 15) / 16 : (a.Cout + 63) / 64;
  grid.z = batch;
  // full-frame 3x3 launches with whole 64-channel tiles: the weights by LDS-DMA (URF_CONV_WDMA=0 in an experiments build: through registers)
  static const bool wdma_on = [] { const char *e = urf::exp_env("URF_CONV_WDMA"); return !e || atoi(e) != 0; }();
  const bool wdma = wdma_on && taps == 9 && !a.gate && (a.Cout % 64) == 0 && (fuse1a || (a.Cin % 64) == 0) && (a.Cout % 4) == 0;
  // URF_CONV_LDS_PAD (experiments build): extra dynamic LDS the kernel never touches -- an occupancy what-if (with 1 KB on top of the
  // 80 KB only ONE convolution workgroup fits on a CU: what SuperPoint costs at the occupancy it has beside a matcher workgroup)
  static const long conv_pad = [] { const char *e = urf::exp_env("URF_CONV_LDS_PAD"); return e ? atol(e) : 0L; }();
  // (round 6, measured and removed: s_setprio 3 / 1 at the top of the full-frame convolutions -- SuperPoint's stream is the critical
  // one of the strict pipeline -- 1125 / 1124 against 1131 frames/s at 640x480, 940 against 934 at 1241x376: nothing)
  const size_t lds = conv_lds_bytes(taps, fuse1a, wdma) + (size_t)(conv_pad > 0 && conv_pad < 70 * 1024 ? conv_pad : 0);
  static DeviceOnce attr_done;
  if (attr_done.need()) {  // > 64 KiB of dynamic LDS needs the opt-in
    const int mx = 72 * 1024;
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<1, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    const int mxd = 160 * 1024;
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mxd));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, true, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mxd));
    URF_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<9, false, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mxd));
    attr_done.mark();
  }
  if (wdma && fuse1a && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, true, 4, true>), grid, block, lds, st, a);
  } else if (wdma && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, false, 4, true>), grid, block, lds, st, a);
  } else if (wdma) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, false, false, 4, true>), grid, block, lds, st, a);
  } else if (split && fuse1a && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, true, 1>), grid, block, lds, st, a);
  } else if (split && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, false, 1>), grid, block, lds, st, a);
  } else if (split) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, false, false, 1>), grid, block, lds, st, a);
  } else if (taps == 9 && fuse1a && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, true>), grid, block, lds, st, a);
  } else if (taps == 9 && pool) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, true, false>), grid, block, lds, st, a);
  } else if (taps == 9) {
    hipLaunchKernelGGL((conv_mfma_kernel<9, false, false>), grid, block, lds, st, a);
  } else if (narrow) {
    hipLaunchKernelGGL((conv_mfma_kernel<1, false, false, 1>), grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL((conv_mfma_kernel<1, false, false>), grid, block, lds, st, a);
  }
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf

#ifdef URF_CONV32_STAMPS
extern "C" int urf_probe_conv32_stamps(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(urf::g_conv32_stamps), sizeof(long long) * 20) == hipSuccess ? 0 : -1;
}
#endif

#ifdef URF_EXPERIMENTS
extern "C" int urf_probe_linear_dma(int v) { urf::g_linear_dma = v; return 0; }
#endif
