// h2conv.hip -- "fast" 3x3 convolution (+bias+ReLU[+2x2 max pool]) on the f16
// matrix core with split operands: x = hi + lo (two f16), products evaluated as
// hi*hi + hi*lo + lo*hi into fp32 accumulators (3 x v_mfma_f32_16x16x32_f16 per
// 32 input channels).  Same tiling as the exact conv_mfma_kernel<9>: 256-thread
// workgroup = 8x16 output pixels x 64 output channels, wave = 2 pixel rows x 4
// channel blocks, M = output channel (A = weights), N = pixel (B = activations).
// Activations travel between layers as two NHWC f16 planes (hi, lo): the same
// bytes as fp32.  Opt-in precision mode.
#include "h2.h"

#include <type_traits>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CTH = 8, CTW = 16, CPH = 10, CPW = 18;
constexpr int CS = 80;   // halfs per LDS row (64 channels + 16 pad): conflict-free ds_read_b128

#ifdef URF_CONV_STAMPS   // diagnostic build only (make EXTRA=-DURF_CONV_STAMPS; tools/gpu_conv_stamps.py)
__device__ long long g_conv_stamps[8];
#define CV_STAMP(i) do { if (FUSE1A && blockIdx.x == 700 && blockIdx.z == 0 && tid == 0) g_conv_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CV_STAMP(i) do { } while (0)
#endif

template <bool POOL, bool FUSE1A, bool OUTF32>
// launch bound of 4 workgroups per CU = a budget of 128 VGPRs (LDS keeps the real number at 2); 128 VGPRs and 78 KB of LDS on purpose: one wave per SIMD of this kernel then fits beside the two 192-register waves per
// SIMD (and the 78 KB) of attn_h2_kernel on the same CU -- in the three-stream pipeline SuperPoint's convolutions fill the
// MFMA bubbles of the matcher's attention, worth more (5 %) than what either kernel gains alone from more registers
__global__ void __launch_bounds__(256, 4) h2conv_kernel(H2ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 csm[];
  _Float16 *in_h = csm, *in_l = csm + CPH * CPW * CS;          // [180][72] each
  _Float16 *w_h = in_l + CPH * CPW * CS, *w_l = w_h + 64 * CS;  // [64 cout][72] each
  float *patch = (float *)(w_l + 64 * CS);                      // FUSE1A: [12][20] f32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, cout_base = blockIdx.y * 64;
  const int tiles_x = (a.W + CTW - 1) / CTW;
  const int y0 = (blockIdx.x / tiles_x) * CTH, x0 = (blockIdx.x % tiles_x) * CTW;

  f32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = a.bias[cout_base + m * 16 + 4 * g + r];
    acc[m][0] = bv; acc[m][1] = bv;
  }

  // weights of one (chunk, tap): 2 planes x 64 cout x 8 pieces = 1024 pieces / 256 threads = 4
  f16x8 wpf[4];
  auto issue_w = [&](int ch, int tap) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      const int plane = i >> 9, co = (i >> 3) & 63, j = i & 7;
      const _Float16 *w = plane ? a.wl : a.wh;  // [tap][Cout][Cin]
      wpf[u] = *(const f16x8 *)(w + ((size_t)tap * a.Cout + cout_base + co) * a.Cin + ch * 64 + 8 * j);
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      const int plane = i >> 9, co = (i >> 3) & 63, j = i & 7;
      *(f16x8 *)((plane ? w_l : w_h) + co * CS + 8 * j) = wpf[u];
    }
  };

  const int nchunks = FUSE1A ? 1 : (a.Cin >> 6);
  CV_STAMP(0);
  issue_w(0, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    if (FUSE1A) {
      // conv1a on the VALU in fp32 (exact chain), split on the way into the LDS tile
      const uint8_t *img = a.img + (size_t)b * a.H * a.W;
      for (int i = tid; i < 12 * 20; i += 256) {
        const int yy = y0 - 2 + i / 20, xx = x0 - 2 + i % 20;
        float v = 0.0f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) v = a.lut[img[(size_t)yy * a.W + xx]];
        patch[i] = v;
      }
      const int c = tid & 63;
      float w1[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) w1[t] = a.w1a[t * 64 + c];
      const float b1 = a.b1a[c];
      __syncthreads();
      CV_STAMP(1);
      // Work item = half a row of the 10 x 18 tile (9 pixels) for the thread's channel: the 3 x 12 patch values it needs come
      // in with nine independent 16-byte broadcast reads, then 81 independent-per-pixel fma chains run from registers.  (As
      // a loop over single pixels -- nine dependent 4-byte reads in front of every chain, a bounds branch per pixel -- this
      // prologue took 18.5 k cycles per workgroup against 14.8 k for the 432 MFMAs per wave that follow.)  Same chain per
      // pixel as before: bias, then the taps in raster order.
      auto half_rows = [&](auto hx_tag) {
        constexpr int hx = decltype(hx_tag)::value;     // which half of the row: the same for every item of a wave (wave & 1)
#pragma unroll 1
        for (int it = tid >> 6; it < 2 * CPH; it += 4) {
          const int py = it >> 1;
          f32x4 r[3][3];                                // patch rows py .. py + 2, columns 8 hx .. 8 hx + 11
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int q = 0; q < 3; ++q) r[dy][q] = *(const f32x4 *)(patch + (py + dy) * 20 + 8 * hx + 4 * q);
          const int yy = y0 - 1 + py;
          const bool rowin = yy >= 0 && yy < a.H;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const int pxx = 9 * hx + k;                 // column in the tile; patch column pxx + dx = 8 hx + (k + hx + dx)
            float v = b1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
              const int j = k + hx + t % 3;
              v = __builtin_fmaf(r[t / 3][j >> 2][j & 3], w1[t], v);
            }
            v = v > 0.0f ? v : 0.0f;
            const int xx = x0 - 1 + pxx;
            // zero padding by a 0 / 1 factor (v >= 0 here, so the product is v or +0 exactly): as a test it is wave-uniform and
            // comes back as a branch per pixel, which keeps the nine chains of a half row from interleaving
            v = v * ((rowin && xx >= 0 && xx < a.W) ? 1.0f : 0.0f);
            const _Float16 hi = (_Float16)v;
            const int p = py * CPW + pxx;
            in_h[p * CS + c] = hi;
            in_l[p * CS + c] = (_Float16)(v - (float)hi);
          }
        }
      };
      if ((tid >> 6) & 1) half_rows(std::integral_constant<int, 1>{});
      else half_rows(std::integral_constant<int, 0>{});
    } else {
      // stage the input tile chunk: 2 planes x 180 pixels x 8 pieces = 2880 16-byte pieces, 12 per thread.  The loads of a
      // plane are ALL issued before its first LDS write, from clamped (always valid) addresses with the zero padding applied afterwards:
      // as a loop with the bounds test around the load, every piece was a basic block of its own -- load, full wait,
      // write -- i.e. twelve dependent memory round trips per chunk in front of 4 us of MFMAs
      // One batch of six per plane (twelve pieces in flight would not fit the 128-VGPR budget of this kernel, see above).
      // The planes are read through buffer resources: a 32-bit offset per piece instead of a 64-bit address.
      constexpr int PPP = CPH * CPW * 8, NB = (PPP + 255) / 256;      // pieces per plane (1440), loads per thread and plane
      const unsigned plane_bytes = (unsigned)((size_t)gridDim.z * a.H * a.W * a.Cin * sizeof(_Float16));
      const unsigned img_off = (unsigned)((size_t)b * a.H * a.W * a.Cin * sizeof(_Float16)) + (unsigned)(ch * 64 * sizeof(_Float16));
#pragma unroll
      for (int plane = 0; plane < 2; ++plane) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(plane ? a.xl : a.xh), 0, plane_bytes, 0x00020000);
        u32x4 sv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int i = tid + 256 * u < PPP ? tid + 256 * u : PPP - 1;
          const int p = i >> 3, j = i & 7;
          const int yy = y0 - 1 + p / CPW, xx = x0 - 1 + p % CPW;
          const bool inb = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
          const int yc = yy < 0 ? 0 : (yy >= a.H ? a.H - 1 : yy), xc = xx < 0 ? 0 : (xx >= a.W ? a.W - 1 : xx);
          const unsigned off = (unsigned)(((yc * a.W + xc) * a.Cin + 8 * j) * (int)sizeof(_Float16));
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, img_off, 0);
          sv[u] = inb ? v : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int i = tid + 256 * u;
          if (i < PPP) *(u32x4 *)((plane ? in_l : in_h) + (i >> 3) * CS + 8 * (i & 7)) = sv[u];
        }
      }
    }
    CV_STAMP(2);
    for (int tap = 0; tap < 9; ++tap) {
      commit_w();
      __syncthreads();
      if (tap + 1 < 9) issue_w(ch, tap + 1);
      else if (ch + 1 < nchunks) issue_w(ch + 1, 0);
      // keep the weight loads of the next tap HERE: left alone, the scheduler sinks them below most of this tap's MFMAs
      // (shorter live ranges) and the commit at the top of the next tap then waits out their whole latency
      __builtin_amdgcn_sched_barrier(0);
      const int toff = ((tap / 3) * CPW + (tap % 3)) * CS;
      const _Float16 *bp0 = in_h + ((2 * wave) * CPW + px) * CS + 8 * g + toff;
      const _Float16 *bp1 = bp0 + CPW * CS;
      const _Float16 *ap = w_h + px * CS + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          ah[m] = *(const f16x8 *)(ap + m * 16 * CS + 32 * ks);
          al[m] = *(const f16x8 *)(ap + 64 * CS + m * 16 * CS + 32 * ks);
        }
        bh[0] = *(const f16x8 *)(bp0 + 32 * ks);
        bh[1] = *(const f16x8 *)(bp1 + 32 * ks);
        bl[0] = *(const f16x8 *)(bp0 + CPH * CPW * CS + 32 * ks);
        bl[1] = *(const f16x8 *)(bp1 + CPH * CPW * CS + 32 * ks);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[r], acc[m][r], 0, 0, 0);
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[r], acc[m][r], 0, 0, 0);
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[r], acc[m][r], 0, 0, 0);
          }
      }
      __syncthreads();
    }
  }

  CV_STAMP(3);
  // ---- epilogue
  auto store = [&](size_t off, f32x4 v) {
    if (OUTF32) {
      *(f32x4 *)(a.out + off) = v;
    } else {
      f16x4 h, l;
#pragma unroll
      for (int q = 0; q < 4; ++q) { h[q] = (_Float16)v[q]; l[q] = (_Float16)(v[q] - (float)h[q]); }
      *(f16x4 *)(a.oh + off) = h;
      *(f16x4 *)(a.ol + off) = l;
    }
  };
  if (POOL) {
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int oy = (y0 >> 1) + wave, ox = (x0 + px) >> 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = fmaxf(acc[m][0][r], acc[m][1][r]);
        t = fmaxf(t, __shfl_xor(t, 1, 64));
        v[r] = t > 0.0f ? t : 0.0f;
      }
      if ((px & 1) == 0 && oy < Ho && ox < Wo)
        store((size_t)b * Ho * Wo * a.Cout + ((size_t)oy * Wo + ox) * a.Cout + cout_base + m * 16 + 4 * g, v);
    }
  } else {
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int yy = y0 + 2 * wave + r2, xx = x0 + px;
      if (yy >= a.H || xx >= a.W) continue;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x4 v = acc[m][r2];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
        store((size_t)b * a.H * a.W * a.Cout + ((size_t)yy * a.W + xx) * a.Cout + cout_base + m * 16 + 4 * g, v);
      }
    }
  }
  CV_STAMP(4);
}

int launch_h2conv(const H2ConvArgs &a, bool pool, bool fuse1a, bool outf32, int batch, hipStream_t st) {
  URF_CHECK((a.Cout % 64) == 0 && (fuse1a || (a.Cin % 64) == 0), "h2conv: unsupported shape");
  const size_t lds = sizeof(_Float16) * (2 * CPH * CPW * CS + 2 * 64 * CS) + (fuse1a ? 12 * 20 * 4 : 0);
  static bool attr_done = false;
  if (!attr_done) {
    const int mx = 80 * 1024;
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    attr_done = true;
  }
  dim3 grid(((a.W + CTW - 1) / CTW) * ((a.H + CTH - 1) / CTH), a.Cout / 64, batch), block(256);
  if (fuse1a) hipLaunchKernelGGL((h2conv_kernel<true, true, false>), grid, block, lds, st, a);
  else if (pool) hipLaunchKernelGGL((h2conv_kernel<true, false, false>), grid, block, lds, st, a);
  else if (outf32) hipLaunchKernelGGL((h2conv_kernel<false, false, true>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((h2conv_kernel<false, false, false>), grid, block, lds, st, a);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf

#ifdef URF_CONV_STAMPS
extern "C" int urf_probe_conv_stamps(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(urf::g_conv_stamps), sizeof(long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
