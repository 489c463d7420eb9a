// gemm128.hip -- exact-fp32 linear layer for wide outputs (Cout % 128 == 0):
// Y[row][cout] = act( chain_k fma(X[row][k], W[k][cout], bias[cout]) ).
// Same arithmetic as conv_mfma_kernel<1> (v_mfma_f32_16x16x4_f32 fma chain, k
// ascending) with a 128-row x 128-cout workgroup tile: each of the 4 waves owns
// 64 rows x 64 couts = 16 accumulators, so one k-step feeds 16 MFMAs from 8 LDS
// operand reads (conv_mfma: 8 from 6) and a barrier pair covers 256 MFMAs/wave.
// Used for the fused QKV projection (256->768) and the first MLP layer
// (512->512) of SuperGlue (src/super_glue.cpp:227 replaces the TensorRT engine).
#include "urf_common.h"

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int G_IN_STRIDE = 66;    // X tile [128][66]
constexpr int G_W_STRIDE = 144;    // W tile [64][144]: 144 % 32 == 16 -> k rows g, g+1 hit disjoint bank halves

__global__ void __launch_bounds__(256, 2) gemm128_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *in_tile = smem;                         // [128][66]
  float *w_tile = smem + 128 * G_IN_STRIDE;      // [64][144]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int b = blockIdx.z;
  const int cout_base = blockIdx.y * 128;
  const int row0 = blockIdx.x * 128;
  if (a.counts && row0 >= a.counts[b]) return;
  const int wr = wave >> 1, wc = wave & 1;       // wave tile: rows wr*64.., couts wc*64..

  f32x4 acc[4][4];                               // [cout tile m][row tile r]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = a.bias[cout_base + wc * 64 + m * 16 + 4 * g + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = bv;
  }

  f32x4 ipf[8], wpf[8];
  auto issue = [&](int ch) {
    const int c0 = ch * 64;
#pragma unroll
    for (int u = 0; u < 8; ++u) {                // X: 128 rows x 16 float4
      const int i = tid + 256 * u;
      const int p = i >> 4, j = i & 15;
      const int cc = c0 + 4 * j, row = row0 + p;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row < a.W) {
        const float *src;
        if (a.in2 && cc >= a.Cin1)
          src = a.in2 + (size_t)b * a.in2_bstride + (size_t)row * a.in2_ld + a.in2_coff + (cc - a.Cin1);
        else
          src = (const float *)a.in + (size_t)b * a.in_bstride + (size_t)row * a.in_ld + a.in_coff + cc;
        v = *(const f32x4 *)src;
      }
      ipf[u] = v;
    }
    const float *wsrc = a.w + (size_t)c0 * a.Cout + cout_base;
#pragma unroll
    for (int u = 0; u < 8; ++u) {                // W: 64 k x 32 float4
      const int i = tid + 256 * u;
      const int k = i >> 5, j = i & 31;
      wpf[u] = *(const f32x4 *)(wsrc + (size_t)k * a.Cout + 4 * j);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 256 * u;
      float *dst = in_tile + (i >> 4) * G_IN_STRIDE + 4 * (i & 15);
      *(float2 *)dst = make_float2(ipf[u][0], ipf[u][1]);
      *(float2 *)(dst + 2) = make_float2(ipf[u][2], ipf[u][3]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = tid + 256 * u;
      *(f32x4 *)(w_tile + (i >> 5) * G_W_STRIDE + 4 * (i & 31)) = wpf[u];
    }
  };

  const int nchunks = a.Cin >> 6;                // Cin % 64 == 0
  issue(0);
  const float *bp = in_tile + (wr * 64 + px) * G_IN_STRIDE + g;
  const float *ap = w_tile + g * G_W_STRIDE + wc * 64 + px;
  for (int ch = 0; ch < nchunks; ++ch) {
    commit();
    __syncthreads();
    if (ch + 1 < nchunks) issue(ch + 1);
    // software pipeline: operand reads of k-step ks+1 are issued before the 16
    // MFMAs of k-step ks (two register sets, pinned with sched_barrier)
    float pa[2][4], pb[2][4];
#define G_LOAD(set, k)                                                        \
  {                                                                           \
    _Pragma("unroll") for (int m = 0; m < 4; ++m) pa[set][m] = ap[(k) * G_W_STRIDE + 16 * m];      \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) pb[set][r] = bp[r * 16 * G_IN_STRIDE + (k)];     \
  }
    G_LOAD(0, 0)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < 16) G_LOAD(cur ^ 1, 4 * (ks + 1))
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[cur][m], pb[cur][r], acc[m][r], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef G_LOAD
    __syncthreads();
  }

  float *outb = a.out + (size_t)b * a.out_bstride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + wr * 64 + r * 16 + px;
    if (row >= a.W) continue;
    float *op = outb + (size_t)row * a.out_ld + a.out_coff + cout_base + wc * 64 + 4 * g;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 v = acc[m][r];
      if (a.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.0f ? v[q] : 0.0f;
      }
      *(f32x4 *)(op + m * 16) = v;
    }
  }
}

int launch_gemm128(const ConvArgs &a, int batch, hipStream_t st) {
  URF_CHECK((a.Cout % 128) == 0 && (a.Cin % 64) == 0 && a.res == nullptr, "gemm128: unsupported shape");
  const size_t lds = sizeof(float) * (128 * G_IN_STRIDE + 64 * G_W_STRIDE);
  static DeviceOnce attr_done;
  if (attr_done.need()) {
    URF_HIP(hipFuncSetAttribute((const void *)gemm128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    attr_done.mark();
  }
  dim3 grid((a.W + 127) / 128, a.Cout / 128, batch);
  hipLaunchKernelGGL(gemm128_kernel, grid, dim3(256), lds, st, a);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf
