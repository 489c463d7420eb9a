// How much VALU work fits in the shadow of v_mfma_f32_16x16x32_f16 on gfx950?  One workgroup per CU, 1 or 2 waves per SIMD,
// a stream of [MFMA, NV VALU instructions of one kind] with sched_barrier(0) pinning the order; cycles per MFMA by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/repro/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap && gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV, int NACC, int BIG>
__global__ void __launch_bounds__(512) k(const float *in, float *out, long long *cyc, int iters) {
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)in[lane + e]; b[e] = (_Float16)in[lane + 8 + e]; }
  f32x4 acc[NACC];
  f32x16 accb[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = f32x4{0, 0, 0, 0}; for (int e = 0; e < 16; ++e) accb[i][e] = 0.0f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[lane + 16 + i];
  unsigned y[4] = {0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      if (BIG) accb[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, accb[u % NACC], 0, 0, 0);
      else acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int i = (u * NV + v) & 7;
        if (KIND == 0) x[i] = x[i] + 1.5f;                                   // v_add_f32
        if (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);                  // v_exp_f32
        if (KIND == 2) {                                                     // v_pk_add_f32
          const int j = i & 6;
          f32x2 p = {x[j], x[j + 1]};
          p = p + f32x2{1.5f, 1.5f};
          x[j] = p[0]; x[j + 1] = p[1];
        }
        if (KIND == 3) {                                                     // v_cvt_pk_f16_f32
          const f16x2 h = {(_Float16)x[i], (_Float16)x[(i + 1) & 7]};
          y[v & 3] ^= __builtin_bit_cast(unsigned, h);
        }
        if (KIND == 4) {                                                     // v_fma_mixlo_f16
          unsigned l;
          asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x[i]), "v"(y[v & 3]));
          y[(v + 1) & 3] = l;
        }
        if (KIND == 6) x[i] = x[i] + x[(i + 3) & 7];                         // v_add_f32, registers only
        if (KIND == 7) x[i] = x[i] * x[(i + 3) & 7];                         // v_mul_f32
        if (KIND == 5) x[i] = fmaxf(fmaxf(x[i], x[(i + 1) & 7]), x[(i + 2) & 7]);   // v_max3_f32
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3] + accb[i][0] + accb[i][15];
  for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)(y[0] ^ y[1] ^ y[2] ^ y[3]);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NV, int NACC, int BIG = 0>
void run(const char *name, const float *in, float *out, long long *cyc) {
  for (int threads : {256, 512}) {
    const int iters = 256, grid = 256;
    hipLaunchKernelGGL((k<KIND, NV, NACC, BIG>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((k<KIND, NV, NACC, BIG>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, sizeof(long long) * grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    const int nw = threads / 64;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nw; ++w) s += (double)h[b * 8 + w];
    printf("%s %-14s NV=%d acc=%d  waves/SIMD=%d: %6.1f cycles per MFMA per wave  (%5.1f per MFMA per SIMD)\n", BIG ? "32x32x16" : "16x16x32", name, NV, NACC, threads / 256,
           s / (grid * nw) / (iters * 16.0), s / (grid * nw) / (iters * 16.0) / (threads / 256));
  }
}

int main() {
  float *in, *out; long long *cyc;
  hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 512); hipMalloc(&cyc, 8 * 256 * 8);
  std::vector<float> h(1024, 0.001f);
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  run<0, 0, 1>("none", in, out, cyc);
  run<6, 1, 1>("v_add_f32 rr", in, out, cyc);
  run<6, 2, 1>("v_add_f32 rr", in, out, cyc);
  run<6, 4, 1>("v_add_f32 rr", in, out, cyc);
  run<7, 2, 1>("v_mul_f32 rr", in, out, cyc);
  run<5, 2, 1>("v_max3_f32", in, out, cyc);
  run<5, 4, 1>("v_max3_f32", in, out, cyc);
  run<1, 1, 1>("v_exp_f32", in, out, cyc);
  run<0, 0, 1, 1>("none", in, out, cyc);
  run<0, 0, 2, 1>("none", in, out, cyc);
  run<6, 2, 1, 1>("v_add_f32 rr", in, out, cyc);
  run<6, 4, 1, 1>("v_add_f32 rr", in, out, cyc);
  run<6, 6, 1, 1>("v_add_f32 rr", in, out, cyc);
  run<6, 8, 1, 1>("v_add_f32 rr", in, out, cyc);
  run<0, 4, 1, 1>("v_add_f32 lit", in, out, cyc);
  run<1, 2, 1, 1>("v_exp_f32", in, out, cyc);
  run<1, 4, 1, 1>("v_exp_f32", in, out, cyc);
  run<5, 4, 1, 1>("v_max3_f32", in, out, cyc);
  run<2, 4, 1, 1>("v_pk_add_f32", in, out, cyc);
  return 0;
}
