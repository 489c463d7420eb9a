// h2mlp.hip -- one launch for the MLP of a SuperGlue GNN layer in the fast precision mode:
//     x <- x + W2 relu(W1 [x ; o] + b1) + b2          (W1 = first MLP layer with the attention `merge` layer folded in)
// (the graph the reference runs as a TensorRT engine, src/super_glue.cpp:227).  Split-f16
// arithmetic of h2gemm.hip (x = hi + lo, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate), the same chunk
// order and the same roundings: the result is bit-identical to the two h2gemm launches it replaces.
//
// What the fusion removes: the 512-wide hidden activations never leave the CU (2 x 33.5 MB per layer through
// HBM/L2 before), one launch per layer and its ~9 us of fixed cost.
//
// Workgroup = 512 threads = 64 tokens x every channel; grid = 16 token tiles x images (256 workgroups at 8 pairs).
// The hidden layer is produced and consumed in two halves of 256 channels so that it fits LDS next to the
// weight stages:
//     for h in {0, 1}:  H_h = relu(W1[256h .. 256h+255] [x;o] + b1)   (K = 512, 16 chunks)  -> LDS, f16 hi/lo planes
//                       acc2 += W2[:, 256h .. 256h+255] H_h            (K = 256,  8 chunks)
// LDS: 2 weight/activation stages of 40 KB (LDS-DMA, global_load_lds_dwordx4, swizzled 64-byte rows as in
// h2gemm_glds_kernel) + 64 KB hidden half = 144 KB -> one workgroup per CU, two waves per SIMD.
// Wave (wc, wr) owns 64 output channels x 32 tokens in every phase: 8 accumulators, 12 ds_read_b128 and 24 MFMAs per
// 32-deep chunk -- the inner loop of h2gemm_glds_kernel.
// (measured slower than the two GEMM launches it replaces, DESIGN.md section 8: compiled into the experiments build only)
#ifdef URF_EXPERIMENTS
#include <cstdlib>

#include "h2.h"

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int MK = 32;                       // K chunk = one MFMA step
constexpr int MT = 64;                       // tokens per workgroup
constexpr int MW_ROWS = 256;                 // weight rows (output channels) per phase
constexpr int M_WPLANE = MW_ROWS * MK;       // halfs per weight plane per stage
constexpr int M_APLANE = MT * MK;            // halfs per activation plane per stage / per hidden chunk
constexpr int M_STAGE = 2 * M_WPLANE + 2 * M_APLANE;   // 20480 halfs = 40 KB
constexpr int M_HID = 8 * 2 * M_APLANE;      // hidden half: 8 chunks x 2 planes = 32768 halfs = 64 KB
constexpr size_t M_LDS_BYTES = sizeof(_Float16) * (2 * M_STAGE + M_HID);

struct MlpArgs {
  _Float16 *xh, *xl;            // residual stream planes [img][NP][256], updated in place
  const _Float16 *oh, *ol;      // attention output planes [img][NP][256]
  const _Float16 *w1h, *w1l;    // [512][512] (cout, k): k < 256 multiplies x, k >= 256 multiplies o
  const _Float16 *w2h, *w2l;    // [256][512]
  const float *b1, *b2;
  const int *counts;
  int rows;                     // tokens reserved per image (NP)
};

__global__ void __launch_bounds__(512, 2) h2mlp_kernel(MlpArgs a) {
  extern __shared__ __attribute__((aligned(1024))) _Float16 lds[];
  _Float16 *stage0 = lds, *hid = lds + 2 * M_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, row0 = blockIdx.x * MT;
  if (row0 >= a.counts[b]) return;
  const int wc = wave >> 1, wr = wave & 1;
  const size_t tok0 = (size_t)b * a.rows + row0;     // first token (row of the [img*NP][256] planes) of this tile

  // ---- LDS-DMA roles.  One wave-instruction moves 16 rows x 64 B of one plane; lane -> row (lane >> 2) of the
  // 16-row block, 16-byte slot (lane & 3) ^ sw(row), sw(r) = (-(r >> 2)) & 3 (the swizzle of h2gemm_glds_kernel).
  const int drow = lane >> 2;                        // row within a 16-row block; (block * 16 + drow) >> 2 & 3 == drow >> 2
  const int dkg = (lane & 3) ^ ((-(drow >> 2)) & 3);
  // fragment read offsets (halfs) inside a plane
  const int swz = 8 * (g ^ ((-(px >> 2)) & 3));
  const int aoff = (wc * 64 + px) * MK + swz;        // weight rows of this wave (+ m * 16 * MK)
  const int boff = (wr * 32 + px) * MK + swz;        // token rows of this wave  (+ r * 16 * MK)

  // phase-1 chunk ch (K index ch * 32 of [x ; o]) of hidden half h into stage s: 32 weight + 8 activation pieces
  auto issue1 = [&](int h, int ch, int s) {
    _Float16 *st = stage0 + s * M_STAGE;
    const int c0 = ch * MK;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int I = wave * 5 + j;                    // 0..39, uniform per wave
      if (I < 32) {
        const int plane = I >> 4, rb = I & 15;
        const _Float16 *src = (plane ? a.w1l : a.w1h) + (size_t)(256 * h + rb * 16 + drow) * 512 + c0 + 8 * dkg;
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(st + plane * M_WPLANE + rb * 16 * MK), 16, 0, 0);
      } else {
        const int plane = (I - 32) >> 2, rb = (I - 32) & 3;
        const bool second = c0 >= 256;
        const _Float16 *base = second ? (plane ? a.ol : a.oh) : (plane ? (const _Float16 *)a.xl : (const _Float16 *)a.xh);
        const _Float16 *src = base + (tok0 + rb * 16 + drow) * 256 + (c0 & 255) + 8 * dkg;
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(st + 2 * M_WPLANE + plane * M_APLANE + rb * 16 * MK), 16, 0, 0);
      }
    }
  };
  // phase-2 chunk ch (hidden channel 256 h + ch * 32) into stage s: 32 weight pieces
  auto issue2 = [&](int h, int ch, int s) {
    _Float16 *st = stage0 + s * M_STAGE;
    const int c0 = 256 * h + ch * MK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int I = wave * 4 + j;
      const int plane = I >> 4, rb = I & 15;
      const _Float16 *src = (plane ? a.w2l : a.w2h) + (size_t)(rb * 16 + drow) * 512 + c0 + 8 * dkg;
      __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(st + plane * M_WPLANE + rb * 16 * MK), 16, 0, 0);
    }
  };
  // 24 MFMAs of one chunk: weights (A operand) from `wst`, tokens (B operand) from the planes bh_p / bl_p
  auto mma = [&](f32x4 (&acc)[4][2], const _Float16 *wst, const _Float16 *bh_p, const _Float16 *bl_p) {
    f16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      ah[m] = *(const f16x8 *)(wst + aoff + m * 16 * MK);
      al[m] = *(const f16x8 *)(wst + M_WPLANE + aoff + m * 16 * MK);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      bh[r] = *(const f16x8 *)(bh_p + boff + r * 16 * MK);
      bl[r] = *(const f16x8 *)(bl_p + boff + r * 16 * MK);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) {    // D[row = channel][col = token], same product order as h2gemm_glds_body
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[r], acc[m][r], 0, 0, 0);
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[r], acc[m][r], 0, 0, 0);
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[r], acc[m][r], 0, 0, 0);
      }
  };

  f32x4 acc2[4][2];                                  // x_new tile of this wave: channels wc*64 + m*16 + 4g.., tokens wr*32 + r*16 + px
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 bv;
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = a.b2[wc * 64 + m * 16 + 4 * g + q];
    acc2[m][0] = bv; acc2[m][1] = bv;
  }

  int s = 0;                                         // stage the next chunk to consume sits in
  issue1(0, 0, 0);
  for (int h = 0; h < 2; ++h) {
    // ---------------- phase 1: hidden half h = relu(W1[256h..] [x;o] + b1)
    f32x4 acc1[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 bv;
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = a.b1[256 * h + wc * 64 + m * 16 + 4 * g + q];
      acc1[m][0] = bv; acc1[m][1] = bv;
    }
    for (int ch = 0; ch < 16; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the chunk have landed
      __syncthreads();                                   // ... and everybody's; the other stage is free again
      if (ch + 1 < 16) issue1(h, ch + 1, s ^ 1);
      else issue2(h, 0, s ^ 1);                          // the first weight chunk of phase 2 rides behind the last MFMAs
      const _Float16 *st = stage0 + s * M_STAGE;
      mma(acc1, st, st + 2 * M_WPLANE, st + 2 * M_WPLANE + M_APLANE);
      s ^= 1;
    }
    // hidden half -> LDS in the chunk layout of a DMA-staged activation plane: channel k' = wc*64 + m*16 + 4g + q of
    // this half lives in chunk k'/32, 16-byte slot ((k' & 31) >> 3) ^ sw(token), halfs 4 (g & 1) .. +3
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int T = wr * 32 + r * 16 + px;
      const int sw = (-(px >> 2)) & 3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc1[m][r][q];
          v = v > 0.0f ? v : 0.0f;
          hi[q] = (_Float16)v;
          lo[q] = (_Float16)(v - (float)hi[q]);
        }
        const int c = wc * 2 + (m >> 1), slot = ((m & 1) * 2 + (g >> 1)) ^ sw;
        _Float16 *dst = hid + (size_t)(c * 2) * M_APLANE + T * MK + slot * 8 + 4 * (g & 1);
        *(f16x4 *)dst = hi;
        *(f16x4 *)(dst + M_APLANE) = lo;
      }
    }
    // ---------------- phase 2: acc2 += W2[:, 256h ..] hidden half
    for (int ch = 0; ch < 8; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // also publishes the hidden half before its first use
      if (ch + 1 < 8) issue2(h, ch + 1, s ^ 1);
      else if (h == 0) issue1(1, 0, s ^ 1);
      const _Float16 *st = stage0 + s * M_STAGE;
      mma(acc2, st, hid + (size_t)(ch * 2) * M_APLANE, hid + (size_t)(ch * 2 + 1) * M_APLANE);
      s ^= 1;
    }
    __syncthreads();                                     // the hidden half is rewritten by the next h
  }

  // ---------------- epilogue: x <- (xh + xl) + acc2, split, store (lane: 4 consecutive channels of one token)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t ro = (tok0 + wr * 32 + r * 16 + px) * 256 + wc * 64 + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f16x4 rh = *(const f16x4 *)(a.xh + ro + m * 16), rl = *(const f16x4 *)(a.xl + ro + m * 16);
      f16x4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = ((float)rh[q] + (float)rl[q]) + acc2[m][r][q];
        hi[q] = (_Float16)v;
        lo[q] = (_Float16)(v - (float)hi[q]);
      }
      *(f16x4 *)(a.xh + ro + m * 16) = hi;
      *(f16x4 *)(a.xl + ro + m * 16) = lo;
    }
  }
}

int launch_h2mlp(_Float16 *xh, _Float16 *xl, const _Float16 *oh, const _Float16 *ol, const _Float16 *w1h,
                 const _Float16 *w1l, const _Float16 *w2h, const _Float16 *w2l, const float *b1, const float *b2,
                 const int *counts, int rows, int nimg, hipStream_t st) {
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    URF_HIP(hipFuncSetAttribute((const void *)h2mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M_LDS_BYTES));
    attr_set.mark();
  }
  MlpArgs a;
  a.xh = xh; a.xl = xl; a.oh = oh; a.ol = ol; a.w1h = w1h; a.w1l = w1l; a.w2h = w2h; a.w2l = w2l;
  a.b1 = b1; a.b2 = b2; a.counts = counts; a.rows = rows;
  hipLaunchKernelGGL(h2mlp_kernel, dim3(rows / MT, nimg), dim3(512), M_LDS_BYTES, st, a);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf

#endif  // URF_EXPERIMENTS
