// sinkhorn_resident.hip -- the 100 Sinkhorn iterations of SuperGlue's optimal-transport layer
// (recurrence: src/super_glue.cpp:432-498 of the reference) as ONE persistent launch whose
// couplings matrix never leaves the chip (fast precision mode; DESIGN.md "Sinkhorn").
//
// The streaming form (sg_kernels.hip, sinkhorn_half_kernel) reads C or C^T from the Infinity
// Cache in every one of the 200 half-iterations: 6.4 GB of fabric traffic per 8-pair step.  Here a
// pair is spread over 32 workgroups = 32 CUs (8 pairs = the whole chip); workgroup w keeps rows
// [32w, 32w+32) of the plan in LDS (32 x 1024 f32 = 128 KiB).  The iteration itself is the scaling form of the same recurrence,
//     P_ij = exp(C_ij + u0_i + v0_j),   u = u0 + log a,   v = v0 + log b,
//     a_i = mu_i / sum_j P_ij b_j        (row pass: local to the workgroup that owns row i)
//     b_j = nu_j / sum_i a_i P_ij        (column pass: 32 partial sums per column, one exchange)
// which is algebraically the reference's  u = log_mu - LSE_j(C + v),  v = log_nu - LSE_i(C + u)
// with the exponentials taken once per re-absorption instead of once per element per pass.  (u0, v0)
// are re-absorbed from C (a_i b_j folded into P, recomputed from the couplings in HBM/L2) after
// iterations 1, 2, 4, 8, ...: the scalings stay near 1, so no entry of P under- or overflows on the way.
// The dustbin row and column (constant alpha, :466-474) are carried analytically, never stored.
//
// Exchange per iteration (one all-reduce of 1025 column sums over the 32 workgroups of a pair), two
// hops of 8-byte {tag, value} granules written with agent-scope (sc1, write-through) stores and
// polled with agent-scope loads -- the data is the flag, no fence (cdna_hip_programming.md G16/R2):
//   hop 1: workgroup w -> reducer r = column / 32;   hop 2: reducer -> every workgroup.
// Every spin is bounded (0.25 s of s_memrealtime): a launch that cannot become co-resident gives up,
// raises *err and the host redoes the batch with the streaming kernels (sg_api.hip) instead of hanging the GPU.
//
// The product carries ONE kernel, sinkhorn_wide_kernel (round 4): the plan tile in registers, 64 rows per workgroup, sixteen
// workgroups per pair, every pair of a batch in one launch, each workgroup alone on its CU.  The experiments build also has the
// forms it superseded: sinkhorn_resident_kernel (plan in LDS, 128 KiB, 1024 threads) and sinkhorn_regs_kernel (plan in registers,
// 32 rows per workgroup, 256 or 512 threads, sharing the CU with other streams' kernels -- and, beside them, not reproducible run
// to run: DESIGN.md section 12).  All share the exchange.
#include "urf_common.h"
#include "urf_math.h"

#include <float.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#ifdef URF_RS_VERIFY_LOADS
__device__ unsigned long long g_rs_verify_bad = 0;
#endif   // explicitly global: the polling loads must be global_load ... sc1, never flat

constexpr int RS_NP = kCap;          // 1024 keypoints per image at most
constexpr int RS_LDC = 1028;         // leading dimension of C in HBM (sg_kernels.hip)
constexpr int RS_WG = 32;            // workgroups (CUs) per pair
constexpr int RS_ROWS = 32;          // plan rows per workgroup
[[maybe_unused]] constexpr int RS_T = 1024;           // threads per workgroup: thread t owns column t
constexpr int RS_XIN = RS_WG * RS_WG * 32 + RS_WG;   // hop-1 granules per pair: [reducer][source][32 columns] + column 1024 [source]
constexpr int RS_XBC = 1056 + 32;            // hop-2 granules per pair (1025 used) + the 32 placement granules
constexpr u64 RS_TIMEOUT_TICKS = 25000000ull;   // s_memrealtime runs at 100 MHz: 0.25 s

struct RsArgs {
  const int *counts;
  const float *C;
  float *u, *v;
  float alpha;
  int iters, pair0, npairs;
  u64 *xin, *xbc;
  unsigned salt;
  int *err;
  unsigned long long *dbg;   // diagnostic (experiments build, URF_RS_DEBUG=1): [pair][iteration][workgroup][2] bit sums of a_i and of the reduced column sums
  int poll_rmw;        // diagnostic (experiments build, URF_SINKHORN_RMW=1): the polls of the register kernel as atomic ORs of 0
  int allow_near;      // 0: always the agent-scope granule stores (A/B runs, URF_SINKHORN_NEAR=0)
  long long *stamps;   // diagnostic runs only (urf_probe_sinkhorn_stamps): s_memtime at 8 points of every iteration
};

// Granule store.  Default: the agent-scope form (sc1: written through to the memory side, ~1 us per hop) -- what the HIP memory
// model asks for between workgroups.  `near` (opt-in, URF_SINKHORN_NEAR=1): when every workgroup of the pair runs on ONE XCD
// (verified at run time, below) a workgroup-scope store only has to reach that XCD's L2 (the L1 is write-through on gfx950),
// where the readers' L1-bypassing agent-scope loads find it ~0.1 us later.  That is below the scope the memory model requires
// between workgroups and relies on this chip's cache hierarchy: measured 0.48 vs 0.60 ms per 8-pair launch, 0.8 % of the
// pipeline's throughput -- not worth leaving the model for, hence off by default.
// diagnostic build (-DURF_RS_GID_TAG): a granule's tag also carries a hash of its own address, so a value that arrives from
// ANOTHER granule of the same iteration (all of an iteration's granules share the plain tag) is taken for "not there yet" and
// polled again instead of being consumed
__device__ __forceinline__ unsigned rs_gh(const void *p) {
#ifdef URF_RS_GID_TAG
  return (unsigned)((unsigned long long)p >> 3) * 0x9E3779B1u;
#else
  (void)p;
  return 0u;
#endif
}
__device__ __forceinline__ void rs_store(u64 *p, unsigned tag, float v, bool near) {
  tag ^= rs_gh(p);
  const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
  if (near) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The workgroup barrier of these kernels: every LDS operation of the wave has completed before it arrives.  __syncthreads() asks
// for the same (a workgroup release fence), but the `s_waitcnt lgkmcnt(0)` of that fence is the compiler's to place -- and hipcc
// (ROCm 7.2, gfx950) leaves it out at a barrier that heads a loop when the pending LDS write sits on the back edge: the other
// waves then read the previous iteration's value whenever the write is still queued behind other workgroups' LDS traffic (the
// un-root-caused "nondeterminism beside other kernels" of round 4).  Inline assembly is invisible to the
// pass that drops the wait.  tools/isa_barrier_audit.py (a CPU test) checks every s_barrier of the library for this.
__device__ __forceinline__ void rs_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// Every lane of the wave with `active` set re-reads its own granule until all their tags equal `tag`.
// Returns false when the launch gave up (time-out here or in another workgroup).
__device__ __forceinline__ bool rs_wait(const u64 *p, bool active, unsigned tag, float &val, int *err) {
  u64 t0 = 0;
  for (unsigned spins = 1;; ++spins) {
    bool ok = true;
    if (active) {
      const u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      val = __uint_as_float((unsigned)x);
      ok = (unsigned)(x >> 32) == (tag ^ rs_gh(p));
    }
    if (__all(ok)) return true;
    if ((spins & 255u) == 0) {
      const u64 now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      const bool late = now - t0 > RS_TIMEOUT_TICKS;
      if (late) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (late || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// ONE wave re-reads its granules g = lane + 64 k (k < N, g < count) every pass until all their tags equal `tag`
// (the polling stays with one wave per CU: sixteen waves polling at once cost 3-5 us per hop).  The values of a
// pass are consumed on the fly -- summed per lane in k order (SUM) or written to dst[g] in LDS -- and are
// only meaningful for the pass that returns true.
template <int N, bool SUM>
__device__ __forceinline__ bool rs_sweep(const u64 *base, int count, unsigned tag, float &sum, float *dst, int lane, int *err,
                                         unsigned *passes = nullptr, bool rmw = false) {
  u64 t0 = 0;
  const gu64 *q = (const gu64 *)base + lane;
  for (unsigned spins = 1;; ++spins) {
    bool ok = true;
    float acc = 0.0f;
    asm volatile("" : "+v"(q));   // one live address per lane: keep the compiler from parking N 64-bit addresses in registers
    // all N loads are issued before the first result is looked at: a pass costs ONE memory round trip, not N
    u64 x[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      x[k] = 0;
      if (lane + 64 * k < count) {
        if (rmw) x[k] = __hip_atomic_fetch_or((u64 *)(q + 64 * k), (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // diagnostic: a read-modify-write at the L2
        else x[k] = __hip_atomic_load(q + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int g = lane + 64 * k;
      if (g < count) {
        const float v = __uint_as_float((unsigned)x[k]);
        if (SUM) acc = acc + v; else dst[g] = v;
        ok = ok && ((unsigned)(x[k] >> 32) == (tag ^ rs_gh((const u64 *)base + g)));
      }
    }
    sum = acc;
    if (passes) *passes = spins;
#ifdef URF_RS_VERIFY_LOADS   // diagnostic build: are the values this wave has just written to LDS the ones it reads back?
    if (!SUM && __all(ok)) {
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const int g = lane + 64 * k;
        if (g < count && __float_as_uint(*(volatile float *)(dst + g)) != (unsigned)x[k]) atomicAdd(&g_rs_verify_bad, 1ull << 32);
      }
    }
#endif
    if (__all(ok)) return true;
    if ((spins & 63u) == 0) {
      const u64 now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      const bool late = now - t0 > RS_TIMEOUT_TICKS;
      if (late) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (late || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// wave-wide sum on the DPP network (VALU speed; six dependent v_add_f32 instead of six LDS-crossbar ds_bpermute
// round trips): quad swaps, half-row and row mirrors, then row_bcast15 / row_bcast31; the total lands in row 3 and
// is broadcast from lane 63.  The summation order differs from bfly64_sum: fast precision mode only.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl, auto rows) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
  };
  v = v + dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});    // quad_perm [1,0,3,2]
  v = v + dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});    // quad_perm [2,3,0,1]
  v = v + dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});   // row_half_mirror
  v = v + dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});   // row_mirror: every lane = its row's sum
  v = v + dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast15 into rows 1, 3
  v = v + dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});   // row_bcast31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes of this half of the wave
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}

#ifdef URF_EXPERIMENTS
#define RS_RMW(a) ((a).poll_rmw != 0)
#else
#define RS_RMW(a) false
#endif
#define RS_STAMP(i) do { if (stamping) a.stamps[(size_t)(k - 1) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#ifdef URF_EXPERIMENTS
// The LDS-resident form (round 2; experiments build only): 1024 threads and 144 KB of LDS per workgroup, so a CU that hosts one
// hosts nothing else; half the chip per launch of four pairs.  Superseded by the register-resident kernels below.
__global__ void __launch_bounds__(RS_T) sinkhorn_resident_kernel(RsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Pt = lds;                      // [32][1024] plan rows of this workgroup
  float *bvec = Pt + RS_ROWS * RS_NP;   // [1024] b_j (0 for j >= n1)
  float *avec = bvec + RS_NP;           // [32] a_i of the own rows (0 for i >= n0)
  float *pcvec = avec + RS_ROWS;        // [32] dustbin-column entries exp(alpha + u0_i + v0_dust)
  float *wsum = pcvec + RS_ROWS;        // [16] per-wave partials of the dustbin-row sum
  float *misc = wsum + 16;              // [0] = b of the dustbin column, [1] = "the launch gave up", [2] = one XCD
  float *csumv = misc + 16;             // [1025 (+3)] reduced column sums of this iteration
  // the absorbed potentials are kept in f64: they are sums of up to eight logarithms of magnitude ~50-100, and the
  // argument of every exponential, C + u0 + v0, cancels them against each other -- in f32 that cancellation alone
  // costs ~1e-5 relative on the plan (measured: max |Z - exact| 5.3e-4 with f32 potentials)
  double *v0vec = (double *)(csumv + 1028);   // [1024] v0_j
  double *u0vec = v0vec + RS_NP;              // [32] u0_i

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = (int)blockIdx.x % a.npairs, w = (int)blockIdx.x / a.npairs;   // pairs congruent mod 8 share an XCD at 8 pairs
  const int p = a.pair0 + pl;
  const int n0 = a.counts[2 * p], n1 = a.counts[2 * p + 1];
  const float *Cp = a.C + (size_t)p * (RS_NP + 1) * RS_LDC;
  u64 *xin = a.xin + (size_t)p * RS_XIN, *xbc = a.xbc + (size_t)p * RS_XBC;
  const float tot = (float)(n0 + n1);
  const float mu = 1.0f / tot, mu_d = (float)n1 / tot, nu = 1.0f / tot, nu_d = (float)n0 / tot;
  const float alpha = a.alpha;
  const int i0 = w * RS_ROWS;                 // first plan row of this workgroup
  const int t = tid;                          // this thread's column
  const bool col_ok = t < n1;
  const int r0 = 2 * wv, r1 = 2 * wv + 1;     // the two rows this wave sums in the row pass
  // owner of the dustbin column's slot: thread n1 (its own column is invalid) or, at n1 == 1024, thread 0's second slot
  const bool own_dust = (n1 < RS_NP) ? (t == n1) : (t == 0);

  float b_t = col_ok ? 1.0f : 0.0f, pd_t = 0.0f;
  double v0_t = 0.0, u0d = -(double)alpha, v0d = 0.0;
  float Pdd = 1.0f, bdust = 1.0f, ad = 0.0f;

  // ---- u0_i = -max_j C_ij over the valid columns and the dustbin entry alpha (local to the row's owner)
  {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int ri = r0 + rr, i = i0 + ri;
      float m = alpha;
      if (i < n0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 256 * q + 4 * lane;
          const f32x4 x = *(const f32x4 *)(Cp + (size_t)i * RS_LDC + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < n1) m = fmaxf(m, x[e]);
        }
      }
      m = bfly64_max(m);
      if (lane == 0) u0vec[ri] = -(double)m;
    }
    v0vec[t] = 0.0;
  }

  // (re)build the plan tile from the couplings: P_ij = exp(C_ij + u0_i + v0_j); b = 1
  auto absorb = [&]() {
    __syncthreads();                      // u0vec / v0vec written, nobody still reads Pt / bvec
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int ri = r0 + rr, i = i0 + ri;
      const double u0i = u0vec[ri];
#pragma unroll 2
      for (int q = 0; q < 4; ++q) {
        const int c = 256 * q + 4 * lane;
        f32x4 pv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (i < n0) {
          const f32x4 x = *(const f32x4 *)(Cp + (size_t)i * RS_LDC + c);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < n1) pv[e] = __expf((float)(((double)x[e] + u0i) + v0vec[c + e]));
        }
        *(f32x4 *)(Pt + ri * RS_NP + c) = pv;
      }
      if (lane == 0) pcvec[ri] = (i < n0) ? __expf((float)(((double)alpha + u0i) + v0d)) : 0.0f;
    }
    bvec[t] = col_ok ? 1.0f : 0.0f;
    if (tid == 0) { misc[0] = 1.0f; misc[1] = 0.0f; }
    __syncthreads();
    pd_t = col_ok ? __expf((float)(((double)alpha + u0d) + v0_t)) : 0.0f;
    Pdd = __expf((float)(((double)alpha + u0d) + v0d));
    b_t = col_ok ? 1.0f : 0.0f;
    bdust = 1.0f;
  };
  absorb();

  // ---- placement: are the 32 workgroups of this pair on one XCD (they are when 8 pairs share a launch: workgroups
  // b and b + 8 are dealt to the same XCD)?  Every workgroup publishes its XCC id with the always-valid agent-scope
  // form and reads all 32; all of them see the same 32 values, so they agree on the protocol.  Speed only.
  bool near = false;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned tag0 = (a.salt << 12) | 0xFFFu;
    if (tid == 0) rs_store(xbc + 1056 + w, tag0, (float)(xcc & 15u), false);
    if (wv == 0) {
      float ids = 0.0f;
      const bool alive = rs_sweep<1, true>(xbc + 1056, RS_WG, tag0, ids, nullptr, lane, a.err);
      const float first = __shfl(ids, 0, 64);
      const bool same = __all(lane >= RS_WG || ids == first);
      if (lane == 0) { misc[2] = same ? 1.0f : 0.0f; if (!alive) misc[1] = 1.0f; }
    }
    __syncthreads();
    if (misc[1] != 0.0f) return;
    near = misc[2] != 0.0f && a.allow_near != 0;
  }

  int next_absorb = 1;
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  for (int k = 1; k <= a.iters; ++k) {
    const unsigned tag = (a.salt << 12) | (unsigned)k;
    // ---------------- row pass: a_i = mu / (sum_j P_ij b_j + pc_i b_dust)
    __syncthreads();                      // bvec, misc[0] of the previous iteration (or of absorb) are in place
    bdust = misc[0];
    RS_STAMP(0);
    {
      f32x4 b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4 *)(bvec + 256 * q + 4 * lane);
      float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 p0 = *(const f32x4 *)(Pt + r0 * RS_NP + 256 * q + 4 * lane);
        const f32x4 p1 = *(const f32x4 *)(Pt + r1 * RS_NP + 256 * q + 4 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = fma_rn(p0[e], b4[q][e], acc0);
          acc1 = fma_rn(p1[e], b4[q][e], acc1);
        }
      }
      acc0 = wave_sum_dpp(acc0);
      acc1 = wave_sum_dpp(acc1);
      const float part = wave_sum_dpp(pd_t * b_t);        // dustbin row: sum_j pd_j b_j
      if (lane == 0) {
        const float ra = fma_rn(pcvec[r0], bdust, acc0), rb = fma_rn(pcvec[r1], bdust, acc1);
        avec[r0] = (i0 + r0 < n0) ? mu / ra : 0.0f;
        avec[r1] = (i0 + r1 < n0) ? mu / rb : 0.0f;
        wsum[wv] = part;
      }
    }
    RS_STAMP(1);
    __syncthreads();
    RS_STAMP(2);
    {
      float rd = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 x = *(const f32x4 *)(wsum + 4 * q);
        rd = (((rd + x[0]) + x[1]) + x[2]) + x[3];
      }
      rd = fma_rn(Pdd, bdust, rd);
      ad = mu_d / rd;
    }
    // ---------------- column pass: partial sums over the own 32 rows (a_i broadcast from LDS, P_it read down the column: conflict-free)
    float creg = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 x = *(const f32x4 *)(avec + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) creg = fma_rn(x[e], Pt[(4 * q + e) * RS_NP + t], creg);
    }
    if (!col_ok) creg = 0.0f;
    float cdust = 0.0f;
    if (wv == ((n1 & (RS_NP - 1)) >> 6)) {    // the wave of the dustbin slot's owner (uniform branch)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 x = *(const f32x4 *)(avec + 4 * q), y = *(const f32x4 *)(pcvec + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) cdust = fma_rn(x[e], y[e], cdust);
      }
    }
    // ---------------- all-reduce of the 1025 column sums over the 32 workgroups of the pair
    {   // hop 1, publish: column t -> reducer t / 32; 32 consecutive lanes write 256 contiguous bytes
      const float v1 = (n1 < RS_NP && t == n1) ? cdust : creg;
      rs_store(xin + ((size_t)((t >> 5) * RS_WG + w) * 32 + (t & 31)), tag, v1, near);
      if (tid == 0) rs_store(xin + (size_t)RS_WG * RS_WG * 32 + w, tag, (n1 == RS_NP) ? cdust : 0.0f, near);
    }
    RS_STAMP(3);
    unsigned np1 = 0, np2 = 0;
    if (wv == 0) {   // wave 0 alone talks to the other CUs; waves 1..15 park at the barrier below
      bool alive = true;
      {   // hop 1, reduce: this workgroup sums columns [32w, 32w+32) over the 32 sources.  Granule g = 32 src + column:
          // lane L sums column L & 31 over the sources (L >> 5) + 2k in k order, then the two half-waves are added
        float x = 0.0f;
        alive = rs_sweep<16, true>(xin + (size_t)w * RS_WG * 32, RS_WG * 32, tag, x, nullptr, lane, a.err, &np1);
        x = x + __shfl_xor(x, 32, 64);
        if (alive && lane < 32) rs_store(xbc + 32 * w + lane, tag, x, near);
        if (alive && w == 31) {           // column 1024 (the dustbin column when n1 == 1024)
          float y = 0.0f;
          alive = rs_sweep<1, true>(xin + (size_t)RS_WG * RS_WG * 32, RS_WG, tag, y, nullptr, lane, a.err);
          const float ys = half_sum32(lane < 32 ? y : 0.0f);
          if (alive && lane == 0) rs_store(xbc + 1024, tag, ys, near);
        }
      }
      RS_STAMP(4);
      if (alive) {   // hop 2: the 1025 reduced sums of every reducer -> LDS
        float unused = 0.0f;
        alive = rs_sweep<17, false>(xbc, 1025, tag, unused, csumv, lane, a.err, &np2);
      }
      if (!alive && lane == 0) misc[1] = 1.0f;
      RS_STAMP(5);
      if (stamping) a.stamps[4096 * 8 + (size_t)(k - 1) * 2] = np1, a.stamps[4096 * 8 + (size_t)(k - 1) * 2 + 1] = np2;
    }
    __syncthreads();
    RS_STAMP(6);
    if (misc[1] != 0.0f) return;
    const float csum = csumv[t], csum2 = csumv[1024];
    // ---------------- b_j = nu / (sum_i a_i P_ij + a_dust pd_j)
    b_t = col_ok ? nu / fma_rn(ad, pd_t, csum) : 0.0f;
    bvec[t] = b_t;
    if (own_dust) misc[0] = nu_d / fma_rn(ad, Pdd, (n1 < RS_NP) ? csum : csum2);
    RS_STAMP(7);
    // ---------------- re-absorb the scalings into (u0, v0) and rebuild P from the couplings
    if (k == next_absorb && k < a.iters) {
      next_absorb *= 2;
      __syncthreads();                    // misc[0] written
      bdust = misc[0];
      if (tid < RS_ROWS && i0 + tid < n0) u0vec[tid] = u0vec[tid] + (double)__logf(avec[tid]);
      u0d = u0d + (double)__logf(ad);
      if (col_ok) v0_t = v0_t + (double)__logf(b_t);
      v0vec[t] = v0_t;
      v0d = v0d + (double)__logf(bdust);
      absorb();
    }
  }
  // ---------------- u = u0 + log a, v = v0 + log b
  __syncthreads();
  bdust = misc[0];
  if (tid < RS_ROWS && i0 + tid < n0) a.u[(size_t)p * RS_LDC + i0 + tid] = (float)(u0vec[tid] + (double)__logf(avec[tid]));
  if (w == 0) {
    if (col_ok) a.v[(size_t)p * RS_LDC + t] = (float)(v0_t + (double)__logf(b_t));
    if (tid == 0) {
      a.u[(size_t)p * RS_LDC + n0] = (float)(u0d + (double)__logf(ad));
      a.v[(size_t)p * RS_LDC + n1] = (float)(v0d + (double)__logf(bdust));
    }
  }
}
#endif

// ---------------------------------------------------------------------------------------------------
// Register-resident variant (URF_SINKHORN_REGS=1).  Same recurrence, same exchange, but the plan tile lives in VGPRs:
// 512 threads, thread t owns columns t and t + 512 of the workgroup's 32 rows (64 registers).  LDS shrinks from 144 KB
// to the vectors (7 KB) and the workgroup to two waves per SIMD, so a CU that hosts one can still take an h2gemm or an
// h2conv workgroup of another stream -- the 144 KB kernel above keeps its CUs to itself.  The price: every row sum is
// a reduction across the 512 threads (32 DPP wave sums per iteration instead of 2).
[[maybe_unused]] constexpr int RG_T = 512;

// wave-wide sum whose total is only valid in lane 63 (wave_sum_dpp without the broadcast)
__device__ __forceinline__ float wave_sum_dpp_l63(float v) {
#ifdef URF_RS_NO_DPP
  for (int o = 1; o < 64; o <<= 1) v = v + __shfl_xor(v, o, 64);
  return v;
#endif
  auto dpp = [](float x, auto ctrl, auto rows) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
  };
  v = v + dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});
  v = v + dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});
  v = v + dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});
  v = v + dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});
  v = v + dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
  v = v + dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
  return v;
}

// 32 values per lane (one per row) -> the 32 wave-wide sums, one per lane pair: a butterfly that halves the number of
// live values at every level (reduce-scatter).  Levels: lane ^ 32 and lane ^ 16 by v_permlane32_swap / v_permlane16_swap
// (one swap + one add per pair of rows), then row_mirror, row_half_mirror and the quad permutations on the DPP
// network (select + DPP add).  70 VALU operations against 32 x 6 for one wave sum per row, and short dependency
// chains.  Afterwards lane l (and l ^ 1) holds the sum of row rs_row_of_lane(l).
typedef unsigned rs_u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float rs_dpp(float x) {
#ifdef URF_RS_NO_DPP   // diagnostic build: the same lane permutations through ds_bpermute
  const int l = (int)(threadIdx.x & 63);
  const int src = CTRL == 0x140 ? ((l & ~15) | (15 - (l & 15))) : CTRL == 0x141 ? ((l & ~7) | (7 - (l & 7))) : CTRL == 0x4E ? (l ^ 2) : (l ^ 1);
  return __shfl(x, src, 64);
#else
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
#endif
}
__device__ __forceinline__ int rs_row_of_lane(int l) {
  return 16 * (l >> 5) + 8 * ((l >> 4) & 1) + 4 * ((l >> 3) & 1) + 2 * ((l >> 2) & 1) + ((l >> 1) & 1);
}
template <typename VAL>
__device__ __forceinline__ float rs_rows32_sum(VAL val, int lane) {   // val(r) = this lane's term of row r, evaluated once
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {      // lanes 0-31 keep row r, lanes 32-63 row r + 16
#ifdef URF_RS_NO_PERMLANE_SWAP        // diagnostic build: the same sums (the additions commute) through ds_bpermute
    const float x0 = val(r), x1 = val(r + 16);
    const bool hi = (lane & 32) != 0;
    v[r] = (hi ? x1 : x0) + __shfl_xor(hi ? x0 : x1, 32, 64);
#else
    const rs_u32x2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(val(r)), __float_as_uint(val(r + 16)), false, false);
    v[r] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
#endif
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {       // 16-lane rows 0, 2 keep row r (+16), rows 1, 3 row r + 8 (+16)
#ifdef URF_RS_NO_PERMLANE_SWAP
    const bool hi = (lane & 16) != 0;
    v[r] = (hi ? v[r + 8] : v[r]) + __shfl_xor(hi ? v[r] : v[r + 8], 16, 64);
#else
    const rs_u32x2 x = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[r + 8]), false, false);
    v[r] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
#endif
  }
  const bool b8 = (lane & 8) != 0, b4 = (lane & 4) != 0, b2 = (lane & 2) != 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {       // l <-> 15 - l within a row of 16: lanes with bit 3 keep row r + 4
    const float keep = b8 ? v[r + 4] : v[r], send = b8 ? v[r] : v[r + 4];
    v[r] = keep + rs_dpp<0x140>(send);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {       // l <-> 7 - l within 8 lanes: bit 2 keeps row r + 2
    const float keep = b4 ? v[r + 2] : v[r], send = b4 ? v[r] : v[r + 2];
    v[r] = keep + rs_dpp<0x141>(send);
  }
  {
    const float keep = b2 ? v[1] : v[0], send = b2 ? v[0] : v[1];   // lane ^ 2: bit 1 keeps row 1
    v[0] = keep + rs_dpp<0x4E>(send);
  }
  return v[0] + rs_dpp<0xB1>(v[0]);   // lane ^ 1
}

#ifdef URF_EXPERIMENTS   // the forms that SHARE their CUs (rounds 2 and 3): not reproducible run to run; experiments build only
// diagnostic builds of the shared form: -DURF_RS_STRONG_BARRIER = every barrier with a full s_waitcnt (vmcnt too) in front and
// wait states behind; -DURF_RS_READBACK = the polling wave reads one of the column sums it has just written back before the barrier
// -DURF_RS_LGKM_BARRIER = the fix: the workgroup barrier with its LDS wait written out (rs_sync below) -- hipcc of ROCm 7.2 drops
// the `s_waitcnt lgkmcnt(0)` of __syncthreads() at the loop-header barrier of these kernels (the pending write, misc[0] = b_dust,
// sits on the back edge), so the other waves could read the previous iteration's value: the run-to-run differences of DESIGN.md
// section 12.  Without the flag these forms keep the plain __syncthreads(): they are the reproducer of that fault.
#if defined(URF_RS_STRONG_BARRIER)
#define RS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); asm volatile("s_nop 7\n\ts_nop 7"); \
                       __builtin_amdgcn_s_barrier(); asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#elif defined(URF_RS_LGKM_BARRIER)
#define RS_SYNC() rs_sync()
#else
#define RS_SYNC() __syncthreads()
#endif
// MINW: waves per SIMD the register budget is cut for (3 -> 168 VGPRs, 4 -> 128, 1 -> no cut).  NC: columns per thread, 1024 / NC
// threads: NC = 2 is the kernel described above; NC = 4 (URF_SINKHORN_REGS=3) is ONE wave per SIMD with 128 registers of plan,
// which leaves the SIMD's other half of the register file to a second kernel's waves (an h2gemm workgroup needs 2 x 104).
template <int MINW, int NC>
__global__ void __launch_bounds__(1024 / NC, MINW) sinkhorn_regs_kernel(RsArgs a) {
  constexpr int T = 1024 / NC, NWV = T / 64;
  __shared__ __attribute__((aligned(16))) float avec[RS_ROWS];      // a_i of the own rows (0 for i >= n0)
  __shared__ __attribute__((aligned(16))) float pcvec[RS_ROWS];     // dustbin-column entries exp(alpha + u0_i + v0_dust)
  __shared__ float rowpart[NWV][RS_ROWS];                            // per-wave partial row sums (or maxima)
  __shared__ float wsum[NWV];                                      // per-wave partials of the dustbin-row sum
  __shared__ float misc[4];                                         // [0] = b of the dustbin column, [1] = gave up, [2] = one XCD
  __shared__ float csumv[1028];                                     // reduced column sums of this iteration
  __shared__ double u0vec[RS_ROWS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = (int)blockIdx.x % a.npairs, w = (int)blockIdx.x / a.npairs;
  const int p = a.pair0 + pl;
  const int n0 = a.counts[2 * p], n1 = a.counts[2 * p + 1];
  const float *Cp = a.C + (size_t)p * (RS_NP + 1) * RS_LDC;
  // the couplings through a buffer resource: row = scalar offset, column = ONE VGPR (+ 2048 B for the second column).  With
  // plain pointers the compiler keeps 64 loop-invariant 64-bit addresses, spills them, and every load of a re-absorption
  // waits for its address to come back from scratch (measured: 76 k ticks per re-absorption against 11 k)
  const __amdgpu_buffer_rsrc_t Crs = __builtin_amdgcn_make_buffer_rsrc((void *)Cp, 0, (RS_NP + 1) * RS_LDC * 4, 0x00020000);
  auto Cload = [&](int row, int c) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Crs, 4 * (int)threadIdx.x + 4 * T * c, row * (RS_LDC * 4), 0));
  };
  u64 *xin = a.xin + (size_t)p * RS_XIN, *xbc = a.xbc + (size_t)p * RS_XBC;
  const float tot = (float)(n0 + n1);
  const float mu = 1.0f / tot, mu_d = (float)n1 / tot, nu = 1.0f / tot, nu_d = (float)n0 / tot;
  const float alpha = a.alpha;
  const int i0 = w * RS_ROWS;
  // this thread's columns: tid + T c.  The dustbin column's sum travels in the slot of column n1 (an invalid column) or, at
  // n1 == 1024, in the extra slot of thread 0
  int col[NC];
  bool ok[NC], dust[NC];
  bool own_dust = (n1 < RS_NP) ? false : (tid == 0);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    col[c] = tid + T * c;
    ok[c] = col[c] < n1;
    dust[c] = n1 < RS_NP && col[c] == n1;
    own_dust = own_dust || dust[c];
  }
  const int dust_wave = (n1 & (T - 1)) >> 6;

  float P[NC][RS_ROWS];
  float bc[NC], pd[NC];
  double v0c[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { bc[c] = ok[c] ? 1.0f : 0.0f; pd[c] = 0.0f; v0c[c] = 0.0; }
  double u0d = -(double)alpha, v0d = 0.0;
  float Pdd = 1.0f, bdust = 1.0f, ad = 0.0f;
  // sum over this thread's columns of x_c y_c, highest column first (NC = 2: fma(x0, y0, x1 y1))
  auto dotc = [&](auto xs, const float (&ys)[NC]) {
    float acc = xs(NC - 1) * ys[NC - 1];
#pragma unroll
    for (int c = NC - 2; c >= 0; --c) acc = fma_rn(xs(c), ys[c], acc);
    return acc;
  };

  // ---- u0_i = -max_j C_ij over the valid columns and the dustbin entry alpha (all 64 loads first, then the wave maxima)
  {
    float mv[RS_ROWS];
#pragma unroll
    for (int i = 0; i < RS_ROWS; ++i) {
      // unconditional loads (every (row, column) read here exists in the buffer); rows >= n0 too: their maxima are
      // never used (a wave-uniform test here would come back as a branch with a full wait behind every load)
      float xs[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) xs[c] = Cload(i0 + i, c);
      float m = alpha;
#pragma unroll
      for (int c = 0; c < NC; ++c) m = ok[c] ? fmaxf(m, xs[c]) : m;
      mv[i] = m;
    }
#pragma unroll
    for (int i = 0; i < RS_ROWS; ++i) {
      const float m = bfly64_max(mv[i]);
      if (lane == 0) rowpart[wv][i] = m;
    }
  }
  RS_SYNC();
  if (tid < RS_ROWS) {
    float m = rowpart[0][tid];
#pragma unroll
    for (int q = 1; q < NWV; ++q) m = fmaxf(m, rowpart[q][tid]);
    u0vec[tid] = (i0 + tid < n0) ? -(double)m : -1.0e300;   // rows past the count: exp(C + u0 + v0) = 0 without a test in absorb()
  }

  auto absorb = [&]() {
    RS_SYNC();                      // u0vec written, nobody still reads avec / pcvec
#pragma unroll
    for (int i = 0; i < RS_ROWS; ++i) {
      const double u0i = u0vec[i];
      float xs[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) xs[c] = Cload(i0 + i, c);
#ifdef URF_RS_VERIFY_LOADS   // diagnostic build: every coupling read a second time through a plain global load, differences counted
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float y = __builtin_nontemporal_load(Cp + (size_t)(i0 + i) * RS_LDC + tid + T * c);
        if (__float_as_uint(y) != __float_as_uint(xs[c])) atomicAdd(&g_rs_verify_bad, 1ull);
      }
#endif
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float q = __expf((float)(((double)xs[c] + u0i) + v0c[c]));
        P[c][i] = ok[c] ? q : 0.0f;     // rows >= n0: u0 = -1e300, the exponential is 0
      }
    }
    if (tid < RS_ROWS) pcvec[tid] = (i0 + tid < n0) ? __expf((float)(((double)alpha + u0vec[tid]) + v0d)) : 0.0f;
    if (tid == 0) { misc[0] = 1.0f; misc[1] = 0.0f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      pd[c] = ok[c] ? __expf((float)(((double)alpha + u0d) + v0c[c])) : 0.0f;
      bc[c] = ok[c] ? 1.0f : 0.0f;
    }
    Pdd = __expf((float)(((double)alpha + u0d) + v0d));
    bdust = 1.0f;
    RS_SYNC();
  };
  absorb();

  // ---- placement (see sinkhorn_resident_kernel)
  bool near = false;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned tag0 = (a.salt << 12) | 0xFFFu;
    if (tid == 0) rs_store(xbc + 1056 + w, tag0, (float)(xcc & 15u), false);
    if (wv == 0) {
      float ids = 0.0f;
      const bool alive = rs_sweep<1, true>(xbc + 1056, RS_WG, tag0, ids, nullptr, lane, a.err);
      const float first = __shfl(ids, 0, 64);
      const bool same = __all(lane >= RS_WG || ids == first);
      if (lane == 0) { misc[2] = same ? 1.0f : 0.0f; if (!alive) misc[1] = 1.0f; }
    }
    RS_SYNC();
    if (misc[1] != 0.0f) return;
    near = misc[2] != 0.0f && a.allow_near != 0;
  }

  int next_absorb = 1;
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  for (int k = 1; k <= a.iters; ++k) {
    const unsigned tag = (a.salt << 12) | (unsigned)k;
    // ---------------- row pass: a_i = mu / (sum_j P_ij b_j + pc_i b_dust); 32 wave sums, 8 partials per row through LDS
    RS_SYNC();                      // misc[0] of the previous iteration (or of absorb) is in place; rowpart / avec are free
    bdust = misc[0];
    RS_STAMP(0);
    {
      const float sv = rs_rows32_sum([&](int r) { return dotc([&](int c) { return P[c][r]; }, bc); }, lane);
      if ((lane & 1) == 0) rowpart[wv][rs_row_of_lane(lane)] = sv;
    }
    {
      const float part = wave_sum_dpp_l63(dotc([&](int c) { return pd[c]; }, bc));   // dustbin row: sum_j pd_j b_j
      if (lane == 63) wsum[wv] = part;
    }
    RS_STAMP(1);
    RS_SYNC();
    if (tid < RS_ROWS) {
      float r = rowpart[0][tid];
#pragma unroll
      for (int q = 1; q < NWV; ++q) r = r + rowpart[q][tid];
      r = fma_rn(pcvec[tid], bdust, r);
      avec[tid] = (i0 + tid < n0) ? mu / r : 0.0f;
    }
    {
      float rd = wsum[0];
#pragma unroll
      for (int q = 1; q < NWV; ++q) rd = rd + wsum[q];
      rd = fma_rn(Pdd, bdust, rd);
      ad = mu_d / rd;
    }
    RS_SYNC();                      // avec written
    RS_STAMP(2);
#ifdef URF_EXPERIMENTS
    if (a.dbg && wv == 0) {
      unsigned long long sb = lane < RS_ROWS ? (unsigned long long)__float_as_uint(avec[lane]) : 0ull;
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
      if (lane == 0) a.dbg[(((size_t)pl * 128 + (k - 1)) * RS_WG + w) * 2] = sb;
    }
#endif
    // ---------------- column pass: partial sums over the own 32 rows, all in registers
    float creg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) creg[c] = 0.0f;
#pragma unroll
    for (int q = 0; q < RS_ROWS / 4; ++q) {
      const f32x4 x = *(const f32x4 *)(avec + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int c = 0; c < NC; ++c) creg[c] = fma_rn(x[e], P[c][4 * q + e], creg[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (!ok[c]) creg[c] = 0.0f;
    float cdust = 0.0f;
    if (wv == dust_wave) {                // the wave of the dustbin slot's owner (uniform branch)
#pragma unroll
      for (int q = 0; q < RS_ROWS / 4; ++q) {
        const f32x4 x = *(const f32x4 *)(avec + 4 * q), y = *(const f32x4 *)(pcvec + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) cdust = fma_rn(x[e], y[e], cdust);
      }
    }
    // ---------------- all-reduce of the 1025 column sums over the 32 workgroups of the pair (layout as above)
#pragma unroll
    for (int c = 0; c < NC; ++c)
      rs_store(xin + ((size_t)((col[c] >> 5) * RS_WG + w) * 32 + (col[c] & 31)), tag, dust[c] ? cdust : creg[c], near);
    if (tid == 0) rs_store(xin + (size_t)RS_WG * RS_WG * 32 + w, tag, (n1 == RS_NP) ? cdust : 0.0f, near);
    RS_STAMP(3);
    if (wv == 0) {
      bool alive = true;
      {
        float x = 0.0f;
        alive = rs_sweep<16, true>(xin + (size_t)w * RS_WG * 32, RS_WG * 32, tag, x, nullptr, lane, a.err, nullptr, RS_RMW(a));
        x = x + __shfl_xor(x, 32, 64);
        if (alive && lane < 32) rs_store(xbc + 32 * w + lane, tag, x, near);
        if (alive && w == 31) {
          float y = 0.0f;
          alive = rs_sweep<1, true>(xin + (size_t)RS_WG * RS_WG * 32, RS_WG, tag, y, nullptr, lane, a.err, nullptr, RS_RMW(a));
          const float ys = half_sum32(lane < 32 ? y : 0.0f);
          if (alive && lane == 0) rs_store(xbc + 1024, tag, ys, near);
        }
      }
      RS_STAMP(4);
      if (alive) {
        float unused = 0.0f;
        alive = rs_sweep<17, false>(xbc, 1025, tag, unused, csumv, lane, a.err, nullptr, RS_RMW(a));
      }
      if (!alive && lane == 0) misc[1] = 1.0f;
#ifdef URF_RS_READBACK
      if (alive) { const float rb = *(volatile float *)(csumv + lane); asm volatile("" :: "v"(rb)); }
#endif
      RS_STAMP(5);
    }
    RS_SYNC();
    RS_STAMP(6);
    if (misc[1] != 0.0f) return;
#ifdef URF_EXPERIMENTS
    if (a.dbg && wv == 0) {
      unsigned long long sb = 0;
      for (int q = 0; q < 17; ++q)
        if (lane + 64 * q < 1025) sb += (unsigned long long)__float_as_uint(csumv[lane + 64 * q]);
      for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
      if (lane == 0) a.dbg[(((size_t)pl * 128 + (k - 1)) * RS_WG + w) * 2 + 1] = sb;
    }
#endif
    float csum_dust = csumv[1024];
    // ---------------- b_j = nu / (sum_i a_i P_ij + a_dust pd_j)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float cs = csumv[col[c]];
      bc[c] = ok[c] ? nu / fma_rn(ad, pd[c], cs) : 0.0f;
      if (dust[c]) csum_dust = cs;
    }
    if (own_dust) misc[0] = nu_d / fma_rn(ad, Pdd, csum_dust);
    RS_STAMP(7);
    // ---------------- re-absorb the scalings into (u0, v0) and rebuild P from the couplings
    if (k == next_absorb && k < a.iters) {
      next_absorb *= 2;
      RS_SYNC();                    // misc[0] written
      bdust = misc[0];
      if (tid < RS_ROWS && i0 + tid < n0) u0vec[tid] = u0vec[tid] + (double)__logf(avec[tid]);
      u0d = u0d + (double)__logf(ad);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (ok[c]) v0c[c] = v0c[c] + (double)__logf(bc[c]);
      v0d = v0d + (double)__logf(bdust);
      absorb();
    }
  }
  // ---------------- u = u0 + log a, v = v0 + log b
  RS_SYNC();
  bdust = misc[0];
  if (tid < RS_ROWS && i0 + tid < n0) a.u[(size_t)p * RS_LDC + i0 + tid] = (float)(u0vec[tid] + (double)__logf(avec[tid]));
  if (w == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (ok[c]) a.v[(size_t)p * RS_LDC + col[c]] = (float)(v0c[c] + (double)__logf(bc[c]));
    if (tid == 0) {
      a.u[(size_t)p * RS_LDC + n0] = (float)(u0d + (double)__logf(ad));
      a.v[(size_t)p * RS_LDC + n1] = (float)(v0d + (double)__logf(bdust));
    }
  }
}

#endif

// ---------------------------------------------------------------------------------------------------
// Wide register-resident form (round 4): 512 threads, thread t owns columns t and t + 512 of the workgroup's SIXTY-FOUR rows (128
// VGPRs of plan; two waves per SIMD, so 256 registers per thread are there), SIXTEEN workgroups per pair -- every pair of a batch
// of eight in ONE launch on 128 CUs.  Same recurrence and the same two-hop exchange as above (16 reducers of 64 columns: a
// reducer's 1024 hop-1 granules are [source][column], so lane l of the sweeping wave sums column l over the sources in order, no
// cross-lane step).  Round 4 launched it with 110 KB of untouched dynamic LDS so that no other kernel's workgroup would share its CU:
// the shared forms were not reproducible run to run and nobody knew why.  Round 5 found why -- a barrier the compiler emitted
// without its LDS wait (rs_sync above) -- and this kernel never had that barrier (tools/isa_barrier_audit.py); the padding is gone
// (same throughput with and without, 1104 frames/s strict, and 3000-step soaks clean either way: DESIGN.md section 12).
constexpr int RW_ROWS = 64, RW_WG = 16, RW_T = 512, RW_NWV = 8, RW_NC = 2;
constexpr size_t kWidePadBytes = 0;            // (experiments build: URF_SINKHORN_WIDE_PAD=bytes brings round 4's padding back for A/B runs)
__global__ void __launch_bounds__(RW_T, 2) sinkhorn_wide_kernel(RsArgs a) {
  constexpr int NC = RW_NC, T = RW_T;
  __shared__ __attribute__((aligned(16))) float avec[RW_ROWS];
  __shared__ __attribute__((aligned(16))) float pcvec[RW_ROWS];
  __shared__ float rowpart[RW_NWV][RW_ROWS];
  __shared__ float wsum[RW_NWV];
  __shared__ float misc[4];
  __shared__ float csumv[1028];
  __shared__ double u0vec[RW_ROWS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = (int)blockIdx.x % a.npairs, w = (int)blockIdx.x / a.npairs;
  const int p = a.pair0 + pl;
  const int n0 = a.counts[2 * p], n1 = a.counts[2 * p + 1];
  const float *Cp = a.C + (size_t)p * (RS_NP + 1) * RS_LDC;
  const __amdgpu_buffer_rsrc_t Crs = __builtin_amdgcn_make_buffer_rsrc((void *)Cp, 0, (RS_NP + 1) * RS_LDC * 4, 0x00020000);
  auto Cload = [&](int row, int c) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Crs, 4 * (int)threadIdx.x + 4 * T * c, row * (RS_LDC * 4), 0));
  };
  u64 *xin = a.xin + (size_t)p * RS_XIN, *xbc = a.xbc + (size_t)p * RS_XBC;
  const float tot = (float)(n0 + n1);
  const float mu = 1.0f / tot, mu_d = (float)n1 / tot, nu = 1.0f / tot, nu_d = (float)n0 / tot;
  const float alpha = a.alpha;
  const int i0 = w * RW_ROWS;
  int col[NC];
  bool ok[NC], dust[NC];
  bool own_dust = (n1 < RS_NP) ? false : (tid == 0);          // the dustbin column's sum travels in the slot of column n1, or (n1 == 1024) in thread 0's extra slot
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    col[c] = tid + T * c;
    ok[c] = col[c] < n1;
    dust[c] = n1 < RS_NP && col[c] == n1;
    own_dust = own_dust || dust[c];
  }
  const int dust_wave = (n1 & (T - 1)) >> 6;

  float P[NC][RW_ROWS];
  float bc[NC], pd[NC];
  double v0c[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { bc[c] = ok[c] ? 1.0f : 0.0f; pd[c] = 0.0f; v0c[c] = 0.0; }
  double u0d = -(double)alpha, v0d = 0.0;
  float Pdd = 1.0f, bdust = 1.0f, ad = 0.0f;

  // ---- u0_i = -max_j C_ij over the valid columns and the dustbin entry alpha (sixteen rows at a time)
#pragma unroll
  for (int ib = 0; ib < RW_ROWS; ib += 16) {
    float mv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float m = alpha;
#pragma unroll
      for (int c = 0; c < NC; ++c) { const float x = Cload(i0 + ib + i, c); m = ok[c] ? fmaxf(m, x) : m; }
      mv[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float m = bfly64_max(mv[i]);
      if (lane == 0) rowpart[wv][ib + i] = m;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  rs_sync();
  if (tid < RW_ROWS) {
    float m = rowpart[0][tid];
#pragma unroll
    for (int q = 1; q < RW_NWV; ++q) m = fmaxf(m, rowpart[q][tid]);
    u0vec[tid] = (i0 + tid < n0) ? -(double)m : -1.0e300;
  }

  auto absorb = [&]() {
    rs_sync();                      // u0vec written, nobody still reads avec / pcvec
#pragma unroll
    for (int ib = 0; ib < RW_ROWS; ib += 16) {   // sixteen rows of couplings in flight at a time
      float xs[16][NC];
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int c = 0; c < NC; ++c) xs[i][c] = Cload(i0 + ib + i, c);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double u0i = u0vec[ib + i];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float q = __expf((float)(((double)xs[i][c] + u0i) + v0c[c]));
          P[c][ib + i] = ok[c] ? q : 0.0f;   // rows >= n0: u0 = -1e300, the exponential is 0
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < RW_ROWS) pcvec[tid] = (i0 + tid < n0) ? __expf((float)(((double)alpha + u0vec[tid]) + v0d)) : 0.0f;
    if (tid == 0) { misc[0] = 1.0f; misc[1] = 0.0f; misc[3] = 0.0f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      pd[c] = ok[c] ? __expf((float)(((double)alpha + u0d) + v0c[c])) : 0.0f;
      bc[c] = ok[c] ? 1.0f : 0.0f;
    }
    Pdd = __expf((float)(((double)alpha + u0d) + v0d));
    bdust = 1.0f;
    rs_sync();
  };
  absorb();

  // ---- the launch's workgroups check in (a launch that cannot become co-resident gives up here, like in the loop), and agree on
  // the granule stores' scope: every workgroup publishes its XCC id with the always-valid agent-scope form and reads all 16 -- the
  // same 16 values everywhere, so the same decision everywhere (`near`: rs_store above; opt-in, speed only)
  bool near = false;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned tag0 = (a.salt << 12) | 0xFFFu;
    if (tid == 0) rs_store(xbc + 1056 + w, tag0, (float)(xcc & 15u), false);
    if (wv == 0) {
      float ids = 0.0f;
      const bool alive = rs_sweep<1, true>(xbc + 1056, RW_WG, tag0, ids, nullptr, lane, a.err);
      const float first = __shfl(ids, 0, 64);
      const bool same = __all(lane >= RW_WG || ids == first);
      if (lane == 0) { misc[2] = same ? 1.0f : 0.0f; if (!alive) misc[1] = 1.0f; }
    }
    rs_sync();
    if (misc[1] != 0.0f) return;
    near = misc[2] != 0.0f && a.allow_near != 0;
  }

  int next_absorb = 1;
  for (int k = 1; k <= a.iters; ++k) {
    const unsigned tag = (a.salt << 12) | (unsigned)k;
    // ---------------- row pass: a_i = mu / (sum_j P_ij b_j + pc_i b_dust); 2 x 32 wave sums, 8 partials per row through LDS
    rs_sync();                      // misc[0] of the previous iteration (or of absorb) is in place; rowpart / avec are free
    bdust = misc[0];
    // Re-absorption out of schedule (round 5): the scalings live in fp32 between two re-absorptions, and with couplings of a large
    // range a column scaling can run away by more than the schedule (after iterations 1, 2, 4 ... 64) allows for -- inf, then NaN
    // (seen with weights of three times the default residual gain; the integrity word of the decode catches the result).  Every
    // workgroup of a pair computes the same b from the same reduced column sums, so "some b has left [2^-48, 2^48]" is the same
    // decision everywhere: the flag was raised at the end of the previous iteration, everybody reads it here, behind the barrier.
    // The scheduled re-absorption (after iterations 1, 2, 4 ... 64) happens here as well, at the top of the NEXT iteration: the
    // same state, one barrier less, and ONE site for both.
    const bool scheduled = k - 1 == next_absorb;
    if (scheduled || ((k & 3) == 1 && misc[3] != 0.0f)) {     // (the flag is looked at every fourth iteration)
      if (scheduled) next_absorb *= 2;
      if (tid < RW_ROWS && i0 + tid < n0) u0vec[tid] = u0vec[tid] + (double)__logf(avec[tid]);
      u0d = u0d + (double)__logf(ad);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (ok[c]) v0c[c] = v0c[c] + (double)__logf(bc[c]);
      v0d = v0d + (double)__logf(bdust);
      absorb();
    }
    {
      const float s0 = rs_rows32_sum([&](int r) { return fma_rn(P[0][r], bc[0], P[1][r] * bc[1]); }, lane);
      if ((lane & 1) == 0) rowpart[wv][rs_row_of_lane(lane)] = s0;
      const float s1 = rs_rows32_sum([&](int r) { return fma_rn(P[0][32 + r], bc[0], P[1][32 + r] * bc[1]); }, lane);
      if ((lane & 1) == 0) rowpart[wv][32 + rs_row_of_lane(lane)] = s1;
    }
    {
      const float part = wave_sum_dpp_l63(fma_rn(pd[0], bc[0], pd[1] * bc[1]));   // dustbin row: sum_j pd_j b_j
      if (lane == 63) wsum[wv] = part;
    }
    rs_sync();
    if (tid < RW_ROWS) {
      float r = rowpart[0][tid];
#pragma unroll
      for (int q = 1; q < RW_NWV; ++q) r = r + rowpart[q][tid];
      r = fma_rn(pcvec[tid], bdust, r);
      avec[tid] = (i0 + tid < n0) ? mu / r : 0.0f;
    }
    {
      float rd = wsum[0];
#pragma unroll
      for (int q = 1; q < RW_NWV; ++q) rd = rd + wsum[q];
      rd = fma_rn(Pdd, bdust, rd);
      ad = mu_d / rd;
    }
    rs_sync();                      // avec written
    // ---------------- column pass: partial sums over the own 64 rows, all in registers
    float creg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) creg[c] = 0.0f;
#pragma unroll
    for (int q = 0; q < RW_ROWS / 4; ++q) {
      const f32x4 x = *(const f32x4 *)(avec + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int c = 0; c < NC; ++c) creg[c] = fma_rn(x[e], P[c][4 * q + e], creg[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (!ok[c]) creg[c] = 0.0f;
    float cdust = 0.0f;
    if (wv == dust_wave) {                // the wave of the dustbin slot's owner (uniform branch)
#pragma unroll
      for (int q = 0; q < RW_ROWS / 4; ++q) {
        const f32x4 x = *(const f32x4 *)(avec + 4 * q), y = *(const f32x4 *)(pcvec + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) cdust = fma_rn(x[e], y[e], cdust);
      }
    }
    // ---------------- all-reduce of the 1025 column sums over the 16 workgroups of the pair: reducer = column / 64,
    // hop-1 granule [reducer][source][column % 64]; the dustbin column's sums at the end of the region, reduced by workgroup 15
#pragma unroll
    for (int c = 0; c < NC; ++c)
      rs_store(xin + ((size_t)((col[c] >> 6) * RW_WG + w) * 64 + (col[c] & 63)), tag, dust[c] ? cdust : creg[c], near);
    if (tid == 0) rs_store(xin + (size_t)RS_WG * RS_WG * 32 + w, tag, (n1 == RS_NP) ? cdust : 0.0f, near);
    if (wv == 0) {
      bool alive = true;
      {
        float x = 0.0f;
        alive = rs_sweep<16, true>(xin + (size_t)w * RW_WG * 64, RW_WG * 64, tag, x, nullptr, lane, a.err);   // x = sum over the sources, in order
        if (alive) rs_store(xbc + 64 * w + lane, tag, x, near);
        if (alive && w == RW_WG - 1) {
          float y = 0.0f;
          alive = rs_sweep<1, true>(xin + (size_t)RS_WG * RS_WG * 32, RW_WG, tag, y, nullptr, lane, a.err);
          const float ys = half_sum32(lane < RW_WG ? y : 0.0f);
          if (alive && lane == 0) rs_store(xbc + 1024, tag, ys, near);
        }
      }
      if (alive) {
        float unused = 0.0f;
        alive = rs_sweep<17, false>(xbc, 1025, tag, unused, csumv, lane, a.err);
      }
      if (!alive && lane == 0) misc[1] = 1.0f;
    }
    rs_sync();
    if (misc[1] != 0.0f) return;
    float csum_dust = csumv[1024];
    // ---------------- b_j = nu / (sum_i a_i P_ij + a_dust pd_j)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float cs = csumv[col[c]];
      bc[c] = ok[c] ? nu / fma_rn(ad, pd[c], cs) : 0.0f;
      if (dust[c]) csum_dust = cs;
    }
    if (own_dust) misc[0] = nu_d / fma_rn(ad, Pdd, csum_dust);
    if ((k & 3) == 0) {
      bool far = false;             // a column scaling more than 2^48 away from 1 (exponent field of the fp32 word; 0 and inf are far)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int e = (int)((__float_as_uint(bc[c]) >> 23) & 0xFFu) - 127;
        far = far || (ok[c] && (e > 48 || e < -48));
      }
      if (__any(far) && lane == 0) misc[3] = 1.0f;
    }
    // (the re-absorption of the scalings into (u0, v0) happens at the top of the next iteration)
  }
  // ---------------- u = u0 + log a, v = v0 + log b
  rs_sync();
  bdust = misc[0];
  if (tid < RW_ROWS && i0 + tid < n0) a.u[(size_t)p * RS_LDC + i0 + tid] = (float)(u0vec[tid] + (double)__logf(avec[tid]));
  if (w == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (ok[c]) a.v[(size_t)p * RS_LDC + col[c]] = (float)(v0c[c] + (double)__logf(bc[c]));
    if (tid == 0) {
      a.u[(size_t)p * RS_LDC + n0] = (float)(u0d + (double)__logf(ad));
      a.v[(size_t)p * RS_LDC + n1] = (float)(v0d + (double)__logf(bdust));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Host side.  Two resident launches must never be in flight together (each needs every workgroup of
// its grid on a CU of its own before any of them can finish): launches of one process are chained on a
// per-device event.  Launches from another process are not seen here; the bounded spins turn that
// case into an error instead of a hang.
struct RsDevice {
  std::mutex mu;
  hipEvent_t last = nullptr;
  int cus = -1;
};
static RsDevice g_rs_dev[16];

size_t sinkhorn_resident_xin_granules(int maxP) { return (size_t)maxP * RS_XIN; }
size_t sinkhorn_resident_xbc_granules(int maxP) { return (size_t)maxP * RS_XBC; }

// 1 = use the resident kernel (fast mode default), 0 = the streaming launches; URF_SINKHORN_RESIDENT overrides
int sinkhorn_resident_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = urf::exp_env("URF_SINKHORN_RESIDENT");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v;
}

long long *g_rs_stamps = nullptr;   // set by urf_probe_sinkhorn_stamps

#ifdef URF_EXPERIMENTS
static std::atomic<int> g_rs_fault{0};   // urf_probe_sinkhorn_fault: that many launches report a give-up (tests of the recovery)
extern std::atomic<int> g_rs_corrupt;    // urf_probe_sinkhorn_corrupt: that many launches get one column potential shifted afterwards
extern float g_rs_corrupt_delta;
__global__ void rs_corrupt_kernel(float *v, float delta) { if (threadIdx.x == 0) v[3] = v[3] + delta; }
#endif

// can this device hold a pair's 32 workgroups at all?  (no: the handle uses the streaming kernels, sg_api.hip)
int sinkhorn_resident_supported(int device) {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  return cus >= RS_WG ? 1 : 0;
}

static unsigned long long *g_rs_dbg = nullptr;   // diagnostic: where the NEXT launch writes its per-iteration bit sums (sg_api.hip, URF_RS_DEBUG)
void sinkhorn_resident_set_debug(unsigned long long *p) { g_rs_dbg = p; }

int launch_sinkhorn_resident(const int *counts, const float *C, float *u, float *v, float alpha, int iters, int P,
                             void *xin, void *xbc, size_t xin_bytes, size_t xbc_bytes, unsigned *salt, int *err, int device,
                             hipStream_t st) {
  URF_CHECK(device >= 0 && device < 16, "sinkhorn_resident: device %d out of range", device);
  URF_CHECK(iters >= 1 && iters < 4096, "sinkhorn_resident: iterations %d outside [1, 4095]", iters);
  RsDevice &d = g_rs_dev[device];
  std::lock_guard<std::mutex> lock(d.mu);
  const size_t lds = sizeof(float) * (RS_ROWS * RS_NP + RS_NP + 2 * RS_ROWS + 32 + 1028) + sizeof(double) * (RS_NP + RS_ROWS);
  if (d.cus < 0) {
    hipDeviceProp_t prop;
    URF_HIP(hipGetDeviceProperties(&prop, device));
    d.cus = prop.multiProcessorCount;
#ifdef URF_EXPERIMENTS
    URF_HIP(hipFuncSetAttribute((const void *)sinkhorn_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // (the padding knob URF_SINKHORN_WIDE_PAD of the experiments build only: the product launches the kernel with no dynamic LDS)
    URF_HIP(hipFuncSetAttribute((const void *)sinkhorn_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#endif
    URF_HIP(hipEventCreateWithFlags(&d.last, hipEventDisableTiming));
    URF_HIP(hipEventRecord(d.last, st));
  }
  // pairs per launch.  A launch may take one CU per workgroup, i.e. cus / 32 pairs; by default it takes HALF the
  // chip (4 pairs on an MI355X): the iterations are latency-bound (an exchange per iteration), so a second launch for
  // the other pairs costs little, and the free half keeps the MFMA kernels of the other streams running
  // (measured, 8 pairs, 3-stream pipeline: 8 per launch 1523 frames/s, 4 per launch 1697).  URF_SINKHORN_GROUP overrides.
  // URF_SINKHORN_REGS: 1 (default) = the register-resident kernel at 168 VGPRs, every pair in ONE launch unless
  // URF_SINKHORN_GROUP says otherwise; 2 = the same at 128 VGPRs (spills in the loop: slower); 0 = the LDS-resident kernel,
  // half the chip per launch.  Measured, 8 pairs, 100 iterations: 0.49 / 0.70 / 1.24 ms serialised and 1895 / 1820 / 1780
  // frames/s in the 3-stream pipeline
  // 3 (default since round 3) = the register-resident kernel with FOUR columns per thread: 256 threads, one wave per SIMD at 270
  // VGPRs without a spill, which leaves 240 registers of every SIMD lane to another stream's waves (measured against 1: 0.61 ->
  // 0.53 ms serialised, +1 % in the pipeline at 640x480, even at 1241x376)
  // 4 (default since the soak of round 4, every mode; the only form of the product build) = the WIDE register-resident kernel: 64 rows
  // per workgroup, sixteen workgroups per pair, every pair of a batch of eight in one launch on 128 CUs that the workgroups have
  // to themselves.  The forms 1 - 3 share their CUs with other streams' kernels and are not reproducible run to run beside them
  // (DESIGN.md section 12); strict mode, 640x480: 1056 frames/s against 1043 (form 3, shared) and 1025 (form 0, LDS-resident)
  static int regs_env = -2;
  if (regs_env == -2) { const char *e = urf::exp_env("URF_SINKHORN_REGS"); regs_env = e ? atoi(e) : -1; if (regs_env < -1 || regs_env > 4) regs_env = -1; }
  const int regs = regs_env >= 0 ? regs_env : 4;
  int group = d.cus / RS_WG;
  if (regs == 4) group = d.cus / RW_WG >= 8 ? 8 : d.cus / RW_WG;        // (sixteen workgroups per pair: a batch of eight in one launch)
  if (regs) {
    const char *e = urf::exp_env("URF_SINKHORN_GROUP");
    const int want = e ? atoi(e) : 0;
    if (want >= 1 && want < group) group = want;
  } else {
    static int knob = -1;
    if (knob < 0) { const char *e = urf::exp_env("URF_SINKHORN_GROUP"); knob = e ? atoi(e) : 0; }
    const int want = knob >= 1 ? knob : (group >= 2 ? group / 2 : group);
    if (want < group) group = want;
  }
  URF_CHECK(group >= 1, "sinkhorn_resident: the device has %d CUs, a pair needs %d (sinkhorn_resident_supported)", d.cus, RS_WG);
  for (int p0 = 0; p0 < P; p0 += group) {
    RsArgs a;
    a.counts = counts; a.C = C; a.u = u; a.v = v; a.alpha = alpha; a.iters = iters;
    a.pair0 = p0; a.npairs = (P - p0 < group) ? (P - p0) : group;
    a.xin = (u64 *)xin; a.xbc = (u64 *)xbc;
    if (*salt >= 0xFFFFFu) {
      // the 20-bit salt wraps: a granule that has not been rewritten since its tag was last current (the regions of pairs the
      // handle has not used for a million launches) could pass for fresh.  Clear every tag, in stream order, and start over.
      URF_HIP(hipStreamWaitEvent(st, d.last, 0));
      URF_HIP(hipMemsetAsync(xin, 0, xin_bytes, st));
      URF_HIP(hipMemsetAsync(xbc, 0, xbc_bytes, st));
      *salt = 0;
    }
    *salt = *salt + 1;                        // tags are (salt << 12 | iteration), never 0
    a.salt = *salt; a.err = err;
    {
      static int near_knob = -1;
      if (near_knob < 0) { const char *e = urf::exp_env("URF_SINKHORN_NEAR"); near_knob = e ? (atoi(e) != 0) : 0; }
      a.allow_near = near_knob;
      static int rmw_knob = -1;
      if (rmw_knob < 0) { const char *e2 = urf::exp_env("URF_SINKHORN_RMW"); rmw_knob = e2 ? (atoi(e2) != 0) : 0; }
      a.poll_rmw = rmw_knob;
      a.dbg = (iters <= 128 && P <= 8) ? g_rs_dbg : nullptr;
      g_rs_dbg = nullptr;
    }
    a.stamps = g_rs_stamps;
    URF_HIP(hipStreamWaitEvent(st, d.last, 0));
#ifdef URF_EXPERIMENTS
    static long lpad = -1;   // what-if: dynamic LDS the kernel never touches (keeps LDS-heavy workgroups of other streams off its CU)
    if (lpad < 0) {
      const char *e = urf::exp_env("URF_SINKHORN_LDS_PAD"); lpad = e ? atol(e) : 0;
      if (lpad > 0) URF_HIP(hipFuncSetAttribute((const void *)sinkhorn_regs_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    static long wpad = -1;   // what-if: the wide form WITHOUT its CU to itself (URF_SINKHORN_WIDE_PAD=0)
    if (wpad < 0) { const char *e = urf::exp_env("URF_SINKHORN_WIDE_PAD"); wpad = e ? atol(e) : (long)kWidePadBytes; }
    if (regs == 4) hipLaunchKernelGGL(sinkhorn_wide_kernel, dim3(RW_WG * a.npairs), dim3(RW_T), (size_t)wpad, st, a);
    else if (regs == 3) hipLaunchKernelGGL((sinkhorn_regs_kernel<1, 4>), dim3(RS_WG * a.npairs), dim3(256), (size_t)lpad, st, a);
    else if (regs == 2) hipLaunchKernelGGL((sinkhorn_regs_kernel<4, 2>), dim3(RS_WG * a.npairs), dim3(RG_T), 0, st, a);
    else if (regs) hipLaunchKernelGGL((sinkhorn_regs_kernel<3, 2>), dim3(RS_WG * a.npairs), dim3(RG_T), 0, st, a);
    else hipLaunchKernelGGL(sinkhorn_resident_kernel, dim3(RS_WG * a.npairs), dim3(RS_T), lds, st, a);
#else
    (void)lds;
    hipLaunchKernelGGL(sinkhorn_wide_kernel, dim3(RW_WG * a.npairs), dim3(RW_T), kWidePadBytes, st, a);   // the product carries this form only
#endif
    URF_HIP(hipGetLastError());
    URF_HIP(hipEventRecord(d.last, st));
  }
#ifdef URF_EXPERIMENTS
  if (g_rs_fault.load() > 0) {
    g_rs_fault.fetch_sub(1);
    URF_HIP(hipMemsetD32Async((hipDeviceptr_t)err, 1, 1, st));    // err[0] = 1: "a launch gave up"
  }
  if (g_rs_corrupt.load() > 0) {
    g_rs_corrupt.fetch_sub(1);
    hipLaunchKernelGGL(rs_corrupt_kernel, dim3(1), dim3(64), 0, st, v, g_rs_corrupt_delta);
    URF_HIP(hipGetLastError());
  }
#endif
  return 0;
}

#ifdef URF_RS_VERIFY_LOADS
extern "C" int urf_probe_rs_verify(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_verify_bad), sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace urf

#ifdef URF_EXPERIMENTS   // test hooks and diagnostics: experiments build only (include/urf.h)
// diagnostic: the next resident launches record s_memtime stamps (8 per iteration, workgroup 0) into a device
// buffer; enable = 0 copies the stamps of `iters` iterations to `out` and switches recording off again
extern "C" int urf_probe_sinkhorn_stamps(int enable, int iters, long long *out) {
  if (enable) {
    if (!urf::g_rs_stamps) {
      if (hipMalloc((void **)&urf::g_rs_stamps, 4096 * 10 * sizeof(long long)) != hipSuccess) return -1;
      (void)hipMemset(urf::g_rs_stamps, 0, 4096 * 10 * sizeof(long long));
    }
    return 0;
  }
  if (!urf::g_rs_stamps || !out || iters < 1 || iters > 4096) return -1;
  (void)hipDeviceSynchronize();
  int rc = hipMemcpy(out, urf::g_rs_stamps, (size_t)iters * 8 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
  // poll passes of hop 1 / hop 2 per iteration follow the stamps in `out` (iters x 2)
  if (hipMemcpy(out + (size_t)iters * 8, urf::g_rs_stamps + 4096 * 8, (size_t)iters * 2 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
  (void)hipFree(urf::g_rs_stamps);
  urf::g_rs_stamps = nullptr;
  return rc;
}

// test hook: the next `launches` resident launches of this process report a give-up although they ran (the recovery path of
// sg_api.hip is otherwise only reachable by starving the launch of CUs)
extern "C" int urf_probe_sinkhorn_fault(int launches) {
  urf::g_rs_fault.store(launches < 0 ? 0 : launches);
  return 0;
}

// test hook: the next `launches` resident launches get `delta` added to one column potential of their first pair after the
// iterations -- a damaged last iteration; the integrity check of the decode (column marginals) must catch it
namespace urf { std::atomic<int> g_rs_corrupt{0}; float g_rs_corrupt_delta = 0.0f; }
extern "C" int urf_probe_sinkhorn_corrupt(int launches, float delta) {
  urf::g_rs_corrupt_delta = delta;
  urf::g_rs_corrupt.store(launches < 0 ? 0 : launches);
  return 0;
}
#endif
