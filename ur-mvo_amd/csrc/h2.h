// h2.h -- split-f16 ("fast" precision mode) shared declarations
#pragma once
#include "urf_common.h"
namespace urf {
struct H2Args {
  const _Float16 *xh, *xl;   // activations [rows][ldx] f16 planes
  const _Float16 *x2h, *x2l; // optional second K-source (channels >= Cin1)
  int ldx, ldx2, Cin1;
  long x_bstride, x2_bstride;
  int rows, Cin;             // Cin % 64 == 0
  const _Float16 *wh, *wl;   // weights [Cout][Cin] f16 planes (k contiguous)
  const float *bias;
  int Cout;                  // % 128 == 0
  float *out;                // optional fp32 output [rows][ld_out]
  _Float16 *oh, *ol;         // optional split output planes [rows][ld_out]
  int ld_out;
  long out_bstride;
  const float *res;          // optional fp32 residual, same geometry as out
  const _Float16 *resh, *resl;   // or: residual as split planes (same geometry as oh/ol), v += hi + lo
  int relu;
  const int *counts;
  // transposed split output (the V projection of attention): ohT/olT [Cout][ldT] per batch item,
  // element (cout, row); written instead of out/oh/ol when non-null
  _Float16 *ohT, *olT;
  int ldT;
  long outT_bstride;
  // dual launch (LDS-DMA kernel): cout tiles below t_from take the normal epilogue (out/oh/ol), tiles from
  // t_from on the transposed one with row index cout - t_from: Q|K and V^T of a GNN layer in ONE launch
  int t_from;
  int xflags;                // diagnostics (urf_probe_h2gemm_xflags): 1 = non-temporal stores, 2 = no stores, 4 = no K loop
};
// fast 3x3 convolution (h2conv.hip).  Activations: NHWC f16 planes [B][H][W][Cin].
struct H2ConvArgs {
  const _Float16 *xh, *xl;   // input planes (unused when fused with conv1a)
  int H, W, Cin;             // conv spatial size, Cin % 64 == 0
  const _Float16 *wh, *wl;   // weights [tap][Cout][Cin] f16 planes
  const float *bias;
  int Cout;                  // % 64 == 0
  _Float16 *oh, *ol;         // output planes (pooled size when POOL), or
  float *out;                // fp32 output (OUTF32)
  // fused conv1a: u8 image [B][H][W], fp32 weights [9][64], bias[64], u8->f32 table
  const uint8_t *img;
  const float *w1a, *b1a, *lut;
};
int launch_h2conv(const H2ConvArgs &a, bool pool, bool fuse1a, bool outf32, int batch, hipStream_t st);
int launch_h2gemm(const H2Args &a, int batch, hipStream_t st);
int launch_split(const float *x, size_t n, _Float16 *h, _Float16 *l, hipStream_t st);
// fused MLP of a GNN layer (h2mlp.hip): x <- x + W2 relu(W1 [x ; o] + b1) + b2 on the split planes, in place
int launch_h2mlp(_Float16 *xh, _Float16 *xl, const _Float16 *oh, const _Float16 *ol, const _Float16 *w1h,
                 const _Float16 *w1l, const _Float16 *w2h, const _Float16 *w2l, const float *b1, const float *b2,
                 const int *counts, int rows, int nimg, hipStream_t st);
int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st);
}  // namespace urf
