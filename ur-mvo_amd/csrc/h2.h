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
  int relu;
  const int *counts;
  // transposed split output (the V projection of attention): ohT/olT [Cout][ldT] per batch item,
  // element (cout, row); written instead of out/oh/ol when non-null
  _Float16 *ohT, *olT;
  int ldT;
  long outT_bstride;
};
int launch_h2gemm(const H2Args &a, int batch, hipStream_t st);
int launch_split(const float *x, size_t n, _Float16 *h, _Float16 *l, hipStream_t st);
int launch_attn_h2(const _Float16 *qkh, const _Float16 *qkl, const _Float16 *vth, const _Float16 *vtl,
                   const int *counts, int cross, _Float16 *oh, _Float16 *ol, int nimg, hipStream_t st);
}  // namespace urf
