// probes.hip -- micro-probes behind the C ABI used by the GPU parity tests:
// (1) the MFMA fma-chain GEMM (is v_mfma_f32_16x16x4_f32 bit-for-bit a k-ordered
//     fmaf chain?), (2) the canonical exp/log/sqrt/div on the device.
#include "../../include/urf.h"
#include "urf_common.h"
#include "urf_math.h"

namespace urf {
int launch_conv(const ConvArgs &a, int taps, bool pool, bool fuse1a, int batch, hipStream_t st);

__global__ void probe_math_kernel(const float *x, int n, float *e, float *l) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  e[i] = exp_c(x[i]);
  const float ax = fabsf(x[i]) + 1.17549435e-38f;
  l[i] = log_c(ax);
}
// sqrt / divide of f32 and f64 (IEEE correctly rounded is assumed by DESIGN.md)
__global__ void probe_divsqrt_kernel(const float *a, const float *b, int n, float *q, float *s, double *qd, double *sd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  q[i] = a[i] / b[i];
  s[i] = __builtin_sqrtf(fabsf(a[i]));
  qd[i] = (double)a[i] / (double)b[i];
  sd[i] = sqrt(fabs((double)a[i] * (double)b[i]));
}
}  // namespace urf
using namespace urf;

// C[m][n] = chain_k fma(A[m][k], B[k][n], bias[n])   (bias may be null -> 0)
extern "C" int urf_probe_fma_gemm(const float *A, const float *B, const float *bias, int M, int N, int K, float *C,
                                  int device) {
  URF_CHECK(A && B && C && (K % 4) == 0 && (N % 4) == 0 && M > 0, "probe_fma_gemm: need K%%4==0, N%%4==0");
  URF_HIP(hipSetDevice(device));
  float *dA, *dB, *db, *dC;
  URF_HIP(hipMalloc((void **)&dA, (size_t)M * K * 4));
  URF_HIP(hipMalloc((void **)&dB, (size_t)K * N * 4));
  URF_HIP(hipMalloc((void **)&db, (size_t)N * 4));
  URF_HIP(hipMalloc((void **)&dC, (size_t)M * N * 4));
  URF_HIP(hipMemcpy(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice));
  URF_HIP(hipMemcpy(dB, B, (size_t)K * N * 4, hipMemcpyHostToDevice));
  if (bias) URF_HIP(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  else URF_HIP(hipMemset(db, 0, (size_t)N * 4));
  ConvArgs a = {};
  a.in = dA; a.in_ld = K; a.H = 1; a.W = M; a.Cin = K; a.w = dB; a.bias = db; a.Cout = N;
  a.out = dC; a.out_ld = N;
  int rc = launch_conv(a, 1, false, false, 1, 0);
  if (rc == 0) {
    URF_HIP(hipDeviceSynchronize());
    URF_HIP(hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(db); (void)hipFree(dC);
  return rc;
}

extern "C" int urf_probe_math(const float *x, int n, float *exp_out, float *log_out, int device) {
  URF_CHECK(x && exp_out && log_out && n > 0, "probe_math: bad argument");
  URF_HIP(hipSetDevice(device));
  float *dx, *de, *dl;
  URF_HIP(hipMalloc((void **)&dx, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&de, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&dl, (size_t)n * 4));
  URF_HIP(hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, de, dl);
  URF_HIP(hipDeviceSynchronize());
  URF_HIP(hipMemcpy(exp_out, de, (size_t)n * 4, hipMemcpyDeviceToHost));
  URF_HIP(hipMemcpy(log_out, dl, (size_t)n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(dx); (void)hipFree(de); (void)hipFree(dl);
  return 0;
}

extern "C" int urf_probe_divsqrt(const float *a, const float *b, int n, float *q, float *s, double *qd, double *sd,
                                 int device) {
  URF_CHECK(a && b && n > 0, "probe_divsqrt: bad argument");
  URF_HIP(hipSetDevice(device));
  float *da, *db, *dq, *ds;
  double *dqd, *dsd;
  URF_HIP(hipMalloc((void **)&da, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&db, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&dq, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&ds, (size_t)n * 4));
  URF_HIP(hipMalloc((void **)&dqd, (size_t)n * 8));
  URF_HIP(hipMalloc((void **)&dsd, (size_t)n * 8));
  URF_HIP(hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
  URF_HIP(hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_divsqrt_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, n, dq, ds, dqd, dsd);
  URF_HIP(hipDeviceSynchronize());
  URF_HIP(hipMemcpy(q, dq, (size_t)n * 4, hipMemcpyDeviceToHost));
  URF_HIP(hipMemcpy(s, ds, (size_t)n * 4, hipMemcpyDeviceToHost));
  URF_HIP(hipMemcpy(qd, dqd, (size_t)n * 8, hipMemcpyDeviceToHost));
  URF_HIP(hipMemcpy(sd, dsd, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dq); (void)hipFree(ds); (void)hipFree(dqd); (void)hipFree(dsd);
  return 0;
}

// ---- one v_mfma_f32_16x16x32_f16 per case: D = A (16x32) B (32x16) + C, raw operands from the host.
// Research probe for the numerics of the f16 matrix core (is its accumulation a reproducible model?).
namespace urf {
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe_mfma_f16_kernel(const _Float16 *A, const _Float16 *B, const float *C, float *D, int n) {
  const int cs = blockIdx.x;
  if (cs >= n) return;
  const int l = threadIdx.x, px = l & 15, g = l >> 4;
  const _Float16 *a = A + (size_t)cs * 512, *b = B + (size_t)cs * 512;
  pf16x8 av, bv;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    av[e] = a[px * 32 + 8 * g + e];          // A[row = px][k = 8g + e]
    bv[e] = b[(8 * g + e) * 16 + px];        // B[k = 8g + e][col = px]
  }
  pf32x4 c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = C[(size_t)cs * 256 + (4 * g + r) * 16 + px];   // C[row = 4g + r][col = px]
  const pf32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(size_t)cs * 256 + (4 * g + r) * 16 + px] = d[r];
}
}  // namespace urf

extern "C" int urf_probe_mfma_f16(const void *A_f16, const void *B_f16, const float *C, float *D, int ncases, int device) {
  URF_CHECK(A_f16 && B_f16 && C && D && ncases > 0, "probe_mfma_f16: bad argument");
  URF_HIP(hipSetDevice(device));
  _Float16 *dA, *dB;
  float *dC, *dD;
  const size_t nh = (size_t)ncases * 512 * 2, nf = (size_t)ncases * 256 * 4;
  URF_HIP(hipMalloc((void **)&dA, nh)); URF_HIP(hipMalloc((void **)&dB, nh));
  URF_HIP(hipMalloc((void **)&dC, nf)); URF_HIP(hipMalloc((void **)&dD, nf));
  URF_HIP(hipMemcpy(dA, A_f16, nh, hipMemcpyHostToDevice));
  URF_HIP(hipMemcpy(dB, B_f16, nh, hipMemcpyHostToDevice));
  URF_HIP(hipMemcpy(dC, C, nf, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(urf::probe_mfma_f16_kernel, dim3(ncases), dim3(64), 0, 0, dA, dB, dC, dD, ncases);
  URF_HIP(hipGetLastError());
  URF_HIP(hipMemcpy(D, dD, nf, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
  return 0;
}

#ifdef URF_EXPERIMENTS   // roof probe, LDS watcher: diagnostics of the experiments build only (include/urf.h)
// what-if timing runs of bench.py: URF_H2GEMM_XFLAGS sets the diagnostics word of the fast linear kernels at load time
// (urf_probe_h2gemm_xflags: 1 = non-temporal stores, 2 = no stores, 4 = ONE K chunk only -- 1/8 .. 1/16 of the operand staging;
// 2 and 4 give wrong results: with URF_REDO_OFF=1 and --no-exact-check only)
namespace urf { extern int g_h2gemm_xflags; }
static const int g_xflags_from_env = [] { const char *e = getenv("URF_H2GEMM_XFLAGS"); if (e) urf::g_h2gemm_xflags = atoi(e); return 0; }();
// ---------------------------------------------------------------------------------------------------
// Roof probe: the split-f16 inner loop of h2gemm / h2conv / h2mlp (24 x v_mfma_f32_16x16x32_f16 on 12 operand
// fragments) with the fragments held in registers -- no LDS, no memory, no barrier.  What the chip sustains on random
// operands at `waves_per_cu` waves per CU is the ceiling the real kernels can be compared with (the clock the chip holds
// under matrix load is well below 2.4 GHz: MI355X_MICROARCH.md, DVFS give-back).
namespace urf {
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
// mode 0: the 24 MFMAs on register-resident fragments.  1: + the GEMM's 12 ds_read_b128 per step.  2: + its barrier.
// 3: + its LDS-DMA (4 x global_load_lds_dwordx4 per wave per step, vmcnt(0) before the barrier), sources L2-resident.
// 4: as 3 with the two activation planes streamed from a buffer larger than the Infinity Cache.
typedef __attribute__((address_space(3))) void probe_lds_void;
typedef const __attribute__((address_space(1))) void probe_gbl_void;
template <int MODE>
__global__ void __launch_bounds__(512, 4) probe_mfma_roof_kernel(const _Float16 *seed, float *sink, int iters, long long *clocks,
                                                                 const _Float16 *big, size_t big_halfs) {
  extern __shared__ __attribute__((aligned(1024))) _Float16 psm[];   // [stage][Ah | Al | Bh | Bl][128][32]
  constexpr int GP = 128 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int wc = (wave >> 2) & 1, wr = wave & 3;
  pf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) { ah[m] = *(const pf16x8 *)(seed + (lane + 64 * m) * 8); al[m] = *(const pf16x8 *)(seed + (lane + 64 * (m + 4)) * 8); }
#pragma unroll
  for (int r = 0; r < 2; ++r) { bh[r] = *(const pf16x8 *)(seed + (lane + 64 * (r + 8)) * 8); bl[r] = *(const pf16x8 *)(seed + (lane + 64 * (r + 10)) * 8); }
  if (MODE >= 1) {
    for (int i = tid; i < 8 * GP / 8; i += blockDim.x) *(pf16x8 *)(psm + 8 * i) = *(const pf16x8 *)(seed + 8 * (i % 768));
    __syncthreads();
  }
  pf32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m) { acc[m][0] = pf32x4{0, 0, 0, 0}; acc[m][1] = pf32x4{0, 0, 0, 0}; }
  const int swz = 8 * (g ^ ((-(px >> 2)) & 3));
  const int aoff = (wc * 64 + px) * 32 + swz, boff = (wr * 32 + px) * 32 + swz;
  const int drow = (wave & 7) * 16 + (lane >> 2);
  const _Float16 *srcA = seed + (size_t)(drow * 4 + (lane & 3)) * 8 % 6000;
  auto issue = [&](int it, int stage) {
    _Float16 *base = psm + stage * 4 * GP + (wave & 7) * 16 * 32;
    const _Float16 *sb = srcA;
    if (MODE >= 4) sb = big + (((size_t)blockIdx.x * (size_t)iters + (size_t)it) * 8192 + (size_t)(drow * 4 + (lane & 3)) * 8) % (big_halfs - 8192);
    __builtin_amdgcn_global_load_lds((probe_gbl_void *)(srcA), (probe_lds_void *)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((probe_gbl_void *)(srcA + 8), (probe_lds_void *)(base + GP), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((probe_gbl_void *)(sb), (probe_lds_void *)(base + 2 * GP), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((probe_gbl_void *)(sb + 4096), (probe_lds_void *)(base + 3 * GP), 16, 0, 0);
  };
  const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  if (MODE >= 3) issue(0, 0);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) issue(it + 1, (it + 1) & 1);
    if (MODE >= 1) {
      asm volatile("" ::: "memory");
      const _Float16 *st = psm + (it & 1) * 4 * GP;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        ah[m] = *(const pf16x8 *)(st + aoff + m * 16 * 32);
        al[m] = *(const pf16x8 *)(st + GP + aoff + m * 16 * 32);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        bh[r] = *(const pf16x8 *)(st + 2 * GP + boff + r * 16 * 32);
        bl[r] = *(const pf16x8 *)(st + 3 * GP + boff + r * 16 * 32);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[r], acc[m][r], 0, 0, 0);
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[r], acc[m][r], 0, 0, 0);
        acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[r], acc[m][r], 0, 0, 0);
      }
  }
  if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[m][0][q] + acc[m][1][q];
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}
// mode 5: the fp32 matrix core alone (v_mfma_f32_16x16x4_f32 on register operands, 8 accumulators per wave as in conv_mfma_kernel):
// what the chip sustains on the exact mode's inner loop, and the clock it holds while doing so
__global__ void __launch_bounds__(256, 2) probe_mfma32_roof_kernel(const float *seed, float *sink, int iters, long long *clocks) {
  const int lane = threadIdx.x & 63;
  float a[4], b[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) a[m] = seed[lane + 64 * m];
  b[0] = seed[lane + 256]; b[1] = seed[lane + 320];
  pf32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m) { acc[m][0] = pf32x4{0, 0, 0, 0}; acc[m][1] = pf32x4{0, 0, 0, 0}; }
  const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[r], acc[m][r], 0, 0, 0);
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  float t = 0.0f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) t += acc[m][0][q] + acc[m][1][q];
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

}  // namespace urf

// returns PFLOP/s of MFMA issue (every MFMA counted) in *pflops and the in-kernel clock (GHz) in *ghz
extern "C" int urf_probe_mfma_roof(int device, int waves_per_cu, int iters, int mode, float *pflops, float *ghz) {
  URF_CHECK(pflops && ghz && iters > 0 && (waves_per_cu == 4 || waves_per_cu == 8 || waves_per_cu == 16) && mode >= 0 && mode <= 5,
            "probe_mfma_roof: bad argument");
  if (mode == 5) {   // fp32 MFMA: 256-thread workgroups, waves_per_cu / 4 of them per CU
    URF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop5;
    URF_HIP(hipGetDeviceProperties(&prop5, device));
    const int blocks5 = prop5.multiProcessorCount * (waves_per_cu / 4);
    float *seed5, *sink5; long long *clk5;
    URF_HIP(hipMalloc((void **)&seed5, 384 * 4));
    URF_HIP(hipMalloc((void **)&sink5, (size_t)blocks5 * 256 * 4));
    URF_HIP(hipMalloc((void **)&clk5, 16));
    float h5[384];
    uint32_t x5 = 777u;
    for (auto &v : h5) { x5 = x5 * 1664525u + 1013904223u; v = ((float)(x5 >> 8) / 16777216.0f - 0.5f) * 0.25f; }
    URF_HIP(hipMemcpy(seed5, h5, sizeof(h5), hipMemcpyHostToDevice));
    hipEvent_t a0, a1;
    URF_HIP(hipEventCreate(&a0)); URF_HIP(hipEventCreate(&a1));
    hipLaunchKernelGGL(urf::probe_mfma32_roof_kernel, dim3(blocks5), dim3(256), 0, 0, seed5, sink5, iters / 10 + 1, clk5);
    URF_HIP(hipEventRecord(a0, 0));
    hipLaunchKernelGGL(urf::probe_mfma32_roof_kernel, dim3(blocks5), dim3(256), 0, 0, seed5, sink5, iters, clk5);
    URF_HIP(hipEventRecord(a1, 0));
    URF_HIP(hipDeviceSynchronize());
    float ms5 = 0.0f;
    (void)hipEventElapsedTime(&ms5, a0, a1);
    long long c5[2] = {0, 1};
    URF_HIP(hipMemcpy(c5, clk5, 16, hipMemcpyDeviceToHost));
    *pflops = (float)((double)blocks5 * 4.0 * (double)iters * 24.0 * 2048.0 / (ms5 * 1e-3) / 1e15);
    *ghz = (float)((double)c5[0] / (double)c5[1] * 0.1);
    (void)hipFree(seed5); (void)hipFree(sink5); (void)hipFree(clk5);
    (void)hipEventDestroy(a0); (void)hipEventDestroy(a1);
    return 0;
  }
  URF_CHECK(mode == 0 || waves_per_cu >= 8, "probe_mfma_roof: modes 1-4 model the 8-wave GEMM workgroup");
  URF_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  URF_HIP(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount;
  const int threads = waves_per_cu >= 8 ? 512 : 256, blocks = cus * (waves_per_cu * 64 / threads);
  const size_t big_halfs = mode >= 4 ? ((size_t)512 << 20) / 2 : 0;
  _Float16 *seed, *big = nullptr; float *sink; long long *clk;
  URF_HIP(hipMalloc((void **)&seed, 64 * 12 * 8 * 2));
  URF_HIP(hipMalloc((void **)&sink, (size_t)blocks * threads * 4));
  URF_HIP(hipMalloc((void **)&clk, 16));
  if (big_halfs) { URF_HIP(hipMalloc((void **)&big, big_halfs * 2)); URF_HIP(hipMemset(big, 0x11, big_halfs * 2)); }
  {
    _Float16 h[64 * 12 * 8];
    uint32_t x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((float)(x >> 8) / 16777216.0f - 0.5f) * 0.25f); }
    URF_HIP(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  URF_HIP(hipEventCreate(&e0)); URF_HIP(hipEventCreate(&e1));
  const size_t lds = mode >= 1 ? 65536 : 0;
  auto launch = [&](int n) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(urf::probe_mfma_roof_kernel<0>, dim3(blocks), dim3(threads), lds, 0, seed, sink, n, clk, big, big_halfs); break;
      case 1: hipLaunchKernelGGL(urf::probe_mfma_roof_kernel<1>, dim3(blocks), dim3(threads), lds, 0, seed, sink, n, clk, big, big_halfs); break;
      case 2: hipLaunchKernelGGL(urf::probe_mfma_roof_kernel<2>, dim3(blocks), dim3(threads), lds, 0, seed, sink, n, clk, big, big_halfs); break;
      case 3: hipLaunchKernelGGL(urf::probe_mfma_roof_kernel<3>, dim3(blocks), dim3(threads), lds, 0, seed, sink, n, clk, big, big_halfs); break;
      default: hipLaunchKernelGGL(urf::probe_mfma_roof_kernel<4>, dim3(blocks), dim3(threads), lds, 0, seed, sink, n, clk, big, big_halfs); break;
    }
  };
  launch(iters / 10 + 1);   // warm-up
  URF_HIP(hipEventRecord(e0, 0));
  launch(iters);
  URF_HIP(hipEventRecord(e1, 0));
  URF_HIP(hipDeviceSynchronize());
  URF_HIP(hipGetLastError());
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[2] = {0, 1};
  URF_HIP(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
  const double flop = (double)blocks * (threads / 64) * (double)iters * 24.0 * 16384.0;
  *pflops = (float)(flop / (ms * 1e-3) / 1e15);
  *ghz = (float)((double)c[0] / (double)c[1] * 0.1);      // s_memrealtime ticks at 100 MHz
  (void)hipFree(seed); (void)hipFree(sink); (void)hipFree(clk);
  if (big) (void)hipFree(big);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

// diagnostic (experiments build, tools/gpu_lds_watch.py): does anything else on the CU write into a workgroup's LDS / registers?
// Every workgroup fills its LDS (static head + dynamic tail) and a few registers with an address pattern and re-checks them for
// `ticks` of s_memrealtime while other streams' kernels (LDS-DMA users among them) come and go on the same CUs.
__global__ void __launch_bounds__(256) lds_watch_kernel(unsigned lds_words, unsigned long long ticks, unsigned long long *bad, unsigned *first) {
  extern __shared__ unsigned wsm[];
  const unsigned salt = 0x9E3779B9u * (blockIdx.x + 1);
  for (unsigned i = threadIdx.x; i < lds_words; i += blockDim.x) wsm[i] = i ^ salt;
  float r0 = (float)threadIdx.x, r1 = r0 * 3.0f + 1.0f, r2 = r0 * 5.0f + 2.0f;   // register canaries through DPP-free arithmetic
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long nbad = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    for (unsigned i = threadIdx.x; i < lds_words; i += blockDim.x) {
      const unsigned v = wsm[i];
      if (v != (i ^ salt)) {
        if (nbad == 0 && atomicAdd(bad + 1, 1ull) == 0) { first[0] = blockIdx.x; first[1] = i; first[2] = v; first[3] = i ^ salt; }
        nbad += 1;
        wsm[i] = i ^ salt;
      }
    }
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2));
    if (r1 != r0 * 3.0f + 1.0f || r2 != r0 * 5.0f + 2.0f) nbad += 1u << 20;
    __builtin_amdgcn_s_sleep(8);
  }
  if (nbad) atomicAdd(bad, nbad);
}
// diagnostic (experiments build, tools/gpu_lds_handoff.py): is a value one wave writes to LDS and publishes with a barrier always
// the value the other waves read?  The workgroup repeats the hand-offs of the register-resident Sinkhorn's iteration (DESIGN.md
// section 12) -- per-wave partials -> 32 threads -> a broadcast vector read as b128 -> one wave that polls global memory and then
// writes 1025 words for everybody -- with values that encode (iteration, index), for `ticks` of s_memrealtime, while other
// streams' kernels (the exact convolutions with their LDS-DMA) share the CUs.  bad[0] = wrong words seen, bad[1] = of those the
// PREVIOUS iteration's value (a stale read), first[0..5] = workgroup, hand-off, index, iteration, got, want of the first one.
namespace urf {
__device__ __forceinline__ unsigned ho_enc(unsigned k, unsigned what, unsigned idx) { return (k * 2654435761u) ^ (what * 0x85EBCA6Bu) ^ (idx * 40503u + 0x1234567u); }
__global__ void __launch_bounds__(256) lds_handoff_kernel(unsigned long long ticks, unsigned long long *bad, unsigned *first, const unsigned long long *poll) {
  __shared__ __attribute__((aligned(16))) unsigned avec[32];
  __shared__ unsigned rowpart[4][32];
  __shared__ unsigned csumv[1028];
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long nbad = 0, nstale = 0;
  auto check = [&](unsigned got, unsigned what, unsigned idx, unsigned k) {
    const unsigned want = ho_enc(k, what, idx);
    if (got != want) {
      if (nbad == 0 && atomicAdd(bad + 2, 1ull) == 0) { first[0] = blockIdx.x; first[1] = what; first[2] = idx; first[3] = k; first[4] = got; first[5] = want; }
      nbad += 1;
      if (got == ho_enc(k - 1, what, idx)) nstale += 1;
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned k = 0;
  bool more = true;
  while (more) {
    ++k;
    __syncthreads();
    if ((lane & 1) == 0) rowpart[wv][lane >> 1] = ho_enc(k, 1 + wv, lane >> 1);       // per-wave partials (half the lanes write, like the row pass)
    __syncthreads();
    if (tid < 32) {
      for (unsigned q = 0; q < 4; ++q) check(rowpart[q][tid], 1 + q, tid, k);
      avec[tid] = ho_enc(k, 7, tid);
    }
    __syncthreads();
    for (unsigned q = 0; q < 8; ++q) {                                                  // the broadcast vector, 16 bytes at a time
      const uint4 x = *(const uint4 *)(avec + 4 * q);
      check(x.x, 7, 4 * q, k); check(x.y, 7, 4 * q + 1, k); check(x.z, 7, 4 * q + 2, k); check(x.w, 7, 4 * q + 3, k);
    }
    if (wv == 0) {
      unsigned long long acc = 0;                                                        // the polling wave: 17 agent-scope loads, then 17 LDS writes
#pragma unroll
      for (unsigned q = 0; q < 17; ++q) acc += __hip_atomic_load(poll + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned z = (unsigned)(acc >> 63);                                          // 0 (the buffer holds small numbers); keeps the loads alive
#pragma unroll
      for (unsigned q = 0; q < 17; ++q)
        if (lane + 64 * q < 1025) csumv[lane + 64 * q] = ho_enc(k, 9, lane + 64 * q) + z;
      more = __builtin_amdgcn_s_memrealtime() - t0 < ticks;
      if (lane == 0) csumv[1026] = more ? 1u : 0u;
    }
    __syncthreads();
    for (unsigned c = 0; c < 4; ++c) check(csumv[tid + 256 * c], 9, tid + 256 * c, k);
    check(csumv[1024], 9, 1024, k);
    more = csumv[1026] != 0u;
  }
  if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad + 1, nstale); }
  if (tid == 0) atomicAdd(bad + 3, (unsigned long long)k);
}
}  // namespace urf
extern "C" int urf_probe_lds_handoff(int device, int wgs, double ms, void *stream, unsigned long long *d_bad, unsigned *d_first, const unsigned long long *d_poll) {
  URF_HIP(hipSetDevice(device));
  hipLaunchKernelGGL(urf::lds_handoff_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(ms * 1e5), d_bad, d_first, d_poll);
  URF_HIP(hipGetLastError());
  return 0;
}
// the dropped-wait reproducer at run time (tools/repro/dropped_lds_wait.hip, tools/gpu_back_edge_repro.py)
#include "../../tools/repro/dropped_lds_wait.hip"
extern "C" int urf_probe_back_edge(int fixed, int wgs, int iters, const float *d_in, float *d_out, void *stream) {
  if (fixed) hipLaunchKernelGGL(back_edge_kernel<true>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, d_out, d_in, iters);
  else hipLaunchKernelGGL(back_edge_kernel<false>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, d_out, d_in, iters);
  URF_HIP(hipGetLastError());
  return 0;
}
extern "C" int urf_probe_lds_watch(int device, int wgs, int lds_bytes, double ms, void *stream, unsigned long long *d_bad, unsigned *d_first) {
  URF_HIP(hipSetDevice(device));
  URF_HIP(hipFuncSetAttribute((const void *)lds_watch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(lds_watch_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (unsigned)(lds_bytes / 4),
                     (unsigned long long)(ms * 1e5), d_bad, d_first);
  URF_HIP(hipGetLastError());
  return 0;
}
#endif
