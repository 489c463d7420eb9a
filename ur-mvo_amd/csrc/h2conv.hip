// h2conv.hip -- "fast" 3x3 convolution (+bias+ReLU[+2x2 max pool]) on the f16
// matrix core with split operands: x = hi + lo (two f16), products evaluated as
// hi*hi + hi*lo + lo*hi into fp32 accumulators (3 x v_mfma_f32_16x16x32_f16 per
// 32 input channels).  Same tiling as the exact conv_mfma_kernel<9>: 256-thread
// workgroup = 8x16 output pixels x 64 output channels, wave = 2 pixel rows x 4
// channel blocks, M = output channel (A = weights), N = pixel (B = activations).
// Activations travel between layers as two NHWC f16 planes (hi, lo): the same
// bytes as fp32.  Opt-in precision mode.
#include "h2.h"

#include <stdlib.h>
#include <type_traits>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CTH = 8, CTW = 16, CPH = 10, CPW = 18;

#ifdef URF_CONV_STAMPS   // diagnostic build only (make EXTRA=-DURF_CONV_STAMPS; tools/gpu_conv_stamps.py)
__device__ long long g_conv_stamps[8];
#ifndef URF_CONV_STAMP_FUSED
#define URF_CONV_STAMP_FUSED 1   // 0: stamp the plain (no pool, no conv1a) variant instead, workgroup 100 of its last large launch
#endif
#define CV_STAMP(i) do { if (FUSE1A == (URF_CONV_STAMP_FUSED != 0) && !POOL == !URF_CONV_STAMP_FUSED && blockIdx.x == (URF_CONV_STAMP_FUSED ? 700 : 100) && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) g_conv_stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CV_STAMP(i) do { } while (0)
#endif

template <bool POOL, bool FUSE1A, bool OUTF32>
// launch bound of 4 workgroups per CU = a budget of 128 VGPRs (LDS keeps the real number at 2); 128 VGPRs and 78 KB of LDS on purpose: one wave per SIMD of this kernel then fits beside the two 192-register waves per
// SIMD (and the 78 KB) of attn_h2_kernel on the same CU -- in the three-stream pipeline SuperPoint's convolutions fill the
// MFMA bubbles of the matcher's attention, worth more (5 %) than what either kernel gains alone from more registers
__global__ void __launch_bounds__(256, 4) h2conv_kernel(H2ConvArgs a) {
  // LDS: unpadded 128-byte rows whose eight 16-byte slots are XOR-swizzled by (row >> 1) & 7 -- the sixteen lanes of a
  // fragment read (16 consecutive rows, one k-group) then hit 16 distinct (row parity, slot) pairs = all 64 banks once -- so
  // that a tap's weights can arrive by LDS-DMA (global_load_lds_dwordx4: lane l -> byte 16 l of a 1-KiB piece = row l >> 3,
  // slot l & 7; the permutation is applied to the SOURCE address) into the stage the previous tap is not reading: no register
  // round trip, ONE barrier per tap.  (Round 1 / early round 2: padded 160-byte rows, weights staged through registers, two
  // barriers per tap.)
  constexpr int RS = 64;                                         // halfs per LDS row
  extern __shared__ __attribute__((aligned(1024))) _Float16 csm[];
  _Float16 *w_h = csm, *w_l = csm + 64 * RS;                      // [2 stages][hi | lo][64 cout][RS]; stage stride 2 * 64 * RS
  _Float16 *in_h = csm + 2 * 2 * 64 * RS, *in_l = in_h + CPH * CPW * RS;     // [180][RS] each
  float *patch = (float *)(in_l + CPH * CPW * RS);                // FUSE1A: [12][20] f32
  // element offset of (row, k-group kg) inside a tile
  auto slot = [&](int row, int kg) { return row * RS + 8 * (kg ^ ((row >> 1) & 7)); };
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

  // weights of one (chunk, tap): 16 pieces of 1 KiB = 2 planes x 8 blocks of 8 rows; wave w issues pieces 4 w .. 4 w + 3
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  auto dma_w = [&](int ch, int tap, int stage) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * wave + u, plane = i >> 3, rb = i & 7;
      const int row = 8 * rb + (lane >> 3), kg = (lane & 7) ^ ((row >> 1) & 7);
      const _Float16 *w = plane ? a.wl : a.wh;
      const _Float16 *src = w + ((size_t)tap * a.Cout + cout_base + row) * a.Cin + ch * 64 + 8 * kg;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(csm + stage * 2 * 64 * RS + plane * 64 * RS + rb * 8 * RS), 16, 0, 0);
    }
  };

  const int nchunks = FUSE1A ? 1 : (a.Cin >> 6);
  CV_STAMP(0);
  dma_w(0, 0, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch > 0) __syncthreads();   // one barrier per tap: slower waves may still read the input tile for the previous chunk's last tap
    if (FUSE1A) {
      // conv1a on the VALU in fp32 (exact chain), split on the way into the LDS tile
      const uint8_t *img = a.img + (size_t)b * a.H * a.W;
      for (int i = tid; i < 12 * 20; i += 256) {
        const int yy = y0 - 2 + i / 20, xx = x0 - 2 + i % 20;
        float v = 0.0f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) v = a.lut[img[(size_t)yy * a.W + xx]];
        patch[i] = v;
      }
      // conv1a on the fp32 matrix core: D[cout][pixel] = bias + sum_t W[cout][t] X[t][pixel] with the nine taps (padded with
      // three zero taps) as the K dimension of v_mfma_f32_16x16x4_f32, whose accumulation IS the ordered chain
      // acc = fma(a_k, b_k, acc), k ascending (tests: test_mfma_f32_is_an_ordered_fma_chain) -- the same bits as the VALU
      // chain "bias, then the taps in raster order" this replaces (a zero tap adds +0; relu maps -0 and +0 alike).
      // 180 pixels = 12 groups of 16, four 16-cout tiles: 36 MFMAs per wave instead of 405 dependent-chain FMAs per thread
      // (the VALU form took 18.5 k cycles per workgroup, 9.3 k after register blocking, against 14.5 k for the 432 MFMAs per
      // wave of conv1b).  A lane's result is 4 consecutive couts of one pixel: 8-byte LDS stores.
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
      CV_STAMP(1);
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int p = (wave + 4 * gi) * 16 + px;         // this lane's pixel of the group (>= 180: no pixel)
        const int pc = p < CPH * CPW ? p : CPH * CPW - 1;
        const int py = pc / CPW, pxx = pc % CPW;
        float xb[3];                                     // B fragments: X[4 ks + g][pixel] = patch[(py + t / 3) * 20 + pxx + t % 3]
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
          f16x4 h, l;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = d[r] > 0.0f ? d[r] : 0.0f;
            v = inb ? v : 0.0f;
            h[r] = (_Float16)v;
            l[r] = (_Float16)(v - (float)h[r]);
          }
          if (p < CPH * CPW) {     // channels 16 mt + 4 g .. + 3 = k-group 2 mt + (g >> 1), halfs 4 (g & 1) .. + 3 of it
            const int o = slot(p, 2 * mt + (g >> 1)) + 4 * (g & 1);
            *(f16x4 *)(in_h + o) = h;
            *(f16x4 *)(in_l + o) = l;
          }
        }
      }
    } else {
      CV_STAMP(1);
      // stage the input tile chunk: 2 planes x 180 pixels x 8 pieces = 2880 16-byte pieces, 12 per thread.  The loads of a
      // plane are ALL issued before its first LDS write, from clamped (always valid) addresses with the zero padding applied afterwards:
      // as a loop with the bounds test around the load, every piece was a basic block of its own -- load, full wait,
      // write -- i.e. twelve dependent memory round trips per chunk in front of 4 us of MFMAs
      // One batch of six per plane (twelve pieces in flight would not fit the 128-VGPR budget of this kernel, see above).
      // The planes are read through buffer resources: a 32-bit offset per piece instead of a 64-bit address.
      constexpr int PPP = CPH * CPW * 8, NB = (PPP + 255) / 256;      // pieces per plane (1440), loads per thread and plane
      // one resource per image (base = the image, records = its bytes): offsets stay 32-bit whatever the batch is (a
      // resource over the whole batch wraps at 4 GiB, i.e. at ~60 frames of 1500 x 1500 with 128 channels)
      const size_t img_halfs = (size_t)a.H * a.W * a.Cin;
      const unsigned plane_bytes = (unsigned)(img_halfs * sizeof(_Float16));
      const unsigned img_off = (unsigned)(ch * 64 * sizeof(_Float16));
#pragma unroll
      for (int plane = 0; plane < 2; ++plane) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((plane ? a.xl : a.xh) + (size_t)b * img_halfs), 0,
                                                                            plane_bytes, 0x00020000);
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
        __builtin_amdgcn_sched_barrier(0);   // the sixth load too: its write is conditional, and the load would follow it into that block
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int i = tid + 256 * u;
          if (i < PPP) *(u32x4 *)((plane ? in_l : in_h) + slot(i >> 3, i & 7)) = sv[u];
        }
      }
    }
    CV_STAMP(2);
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tap's weights have landed
      __syncthreads();                                    // everybody's have, and everybody is done with the other stage
      {
        const int nt = tap + 1 < 9 ? tap + 1 : 0, nc = tap + 1 < 9 ? ch : ch + 1;
        if (nc < nchunks) dma_w(nc, nt, (ch * 9 + tap + 1) & 1);
      }
      const int wst = ((ch * 9 + tap) & 1) * 2 * 64 * RS;
      const int p0 = (2 * wave + tap / 3) * CPW + px + tap % 3, p1 = p0 + CPW;   // this lane's two pixels for this tap
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int o = wst + slot(16 * m + px, g + 4 * ks);
          ah[m] = *(const f16x8 *)(w_h + o);
          al[m] = *(const f16x8 *)(w_l + o);
        }
        const int o0 = slot(p0, g + 4 * ks), o1 = slot(p1, g + 4 * ks);
        bh[0] = *(const f16x8 *)(in_h + o0);
        bh[1] = *(const f16x8 *)(in_h + o1);
        bl[0] = *(const f16x8 *)(in_l + o0);
        bl[1] = *(const f16x8 *)(in_l + o1);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[r], acc[m][r], 0, 0, 0);
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[r], acc[m][r], 0, 0, 0);
            acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[r], acc[m][r], 0, 0, 0);
          }
      }
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
  const size_t lds = sizeof(_Float16) * (2 * 2 * 64 * 64 + 2 * CPH * CPW * 64) + (fuse1a ? 12 * 20 * 4 : 0);
  static DeviceOnce attr_done;
  if (attr_done.need()) {
    const int mx = 80 * 1024;
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    URF_HIP(hipFuncSetAttribute((const void *)h2conv_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    attr_done.mark();
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
