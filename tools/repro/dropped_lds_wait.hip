// Reproducer: hipcc (ROCm 7.2.0, gfx950, -O3) emits the barrier at the head of this loop as a bare `s_barrier`, without the
// `s_waitcnt lgkmcnt(0)` that __syncthreads()'s workgroup release fence needs for the LDS write made at the END of the previous
// iteration (`misc[0] = ...` by thread 77, on the loop's back edge).  The barriers inside the loop body get their wait.
//
//   /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -o dropped_lds_wait.s tools/repro/dropped_lds_wait.hip
//   python tools/isa_barrier_audit.py dropped_lds_wait.s        ->  "1 reachable with an LDS write in flight" (the loop header)
//
// What it costs at run time: the wave of thread 77 may pass the barrier while its ds_write is still queued; the other waves' reads of
// misc[0] behind the barrier then return the PREVIOUS iteration's value.  On an otherwise idle CU the write always lands first; with
// other workgroups' LDS traffic on the CU (LDS-DMA tile fills, 16-byte fragment reads) it sometimes does not -- the run-to-run
// differences of the register-resident Sinkhorn kernels of round 4.  tools/gpu_back_edge_repro.py runs this
// kernel (urf_probe_back_edge, experiments build) alone and beside the exact SuperPoint, with and without the wait written out.
#include <hip/hip_runtime.h>

template <bool FIXED>
__global__ void __launch_bounds__(256) back_edge_kernel(float *out, const float *in, int iters) {
  __shared__ float misc[4];
  __shared__ float part[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float acc = in[tid];
  if (tid == 0) misc[0] = 1.0f;
  for (int k = 1; k <= iters; ++k) {
    if (FIXED) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wait the compiler leaves out
    __syncthreads();
    const float b = misc[0];                                        // every thread: the value published at the end of iteration k - 1
    float v = acc * b;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) part[wv] = v;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    acc = acc * 0.5f + tot * 1e-3f;
    if (tid == 77) misc[0] = 1.0f / (1.0f + tot * tot);            // one thread publishes for the next iteration: the back edge
  }
  out[blockIdx.x * 256 + tid] = acc;
}
template __global__ void back_edge_kernel<false>(float *, const float *, int);
template __global__ void back_edge_kernel<true>(float *, const float *, int);
