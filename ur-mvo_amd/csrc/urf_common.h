// urf_common.h -- shared host/device declarations of liburf_front.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <string>

namespace urf {

void set_error(const char *fmt, ...);

// The A/B knobs of the kernel experiments (URF_SINKHORN_*, URF_H2GEMM_*, URF_ATTN_VARIANT, ...) are read
// from the environment only in a build made with `make EXTRA=-DURF_EXPERIMENTS`; the product build never reads the
// environment -- what a handle computes and guarantees is decided by its configuration struct alone (include/urf.h).
inline const char *exp_env(const char *name) {
#ifdef URF_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// "once per device" guard for hipFuncSetAttribute: the attribute belongs to the (kernel, device) pair, and the library may
// drive several devices from several host threads.  Doing the work twice is harmless (idempotent), skipping it is not.
struct DeviceOnce {
  std::atomic<unsigned long long> done{0};
  static int cur() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d & 63; }
  bool need() { return ((done.load(std::memory_order_acquire) >> cur()) & 1ull) == 0; }
  void mark() { done.fetch_or(1ull << cur(), std::memory_order_release); }
};

#define URF_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      urf::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

#define URF_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      urf::set_error(__VA_ARGS__);      \
      return -2;                        \
    }                                   \
  } while (0)

// ---------------------------------------------------------------- conv/GEMM
// One launch = act( W (*) in + bias ) [+ res], implicit GEMM on
// v_mfma_f32_16x16x4_f32; every output is ONE fp32 fma chain:
//   acc = bias; for chunk(64 ch); for tap; for c in chunk: acc = fma(in, w, acc)
struct ConvArgs {
  const void *in;        // f32 NHWC (or u8 image when FUSE1A)
  long in_bstride;       // elements per batch item
  int in_ld, in_coff;    // floats per pixel, first channel
  const float *in2;      // optional 2nd K-source (TAPS==1): channels [Cin1, Cin) come from here
  int in2_ld, in2_coff, Cin1;
  long in2_bstride;
  int H, W;              // TAPS==9: conv spatial size.  TAPS==1: H=1, W=rows
  int Cin;               // multiple of 4; chunks of 64
  const float *w;        // [taps][Cin][Cout]
  const float *bias;     // [Cout]
  int Cout;
  float *out;
  long out_bstride;
  int out_ld, out_coff;
  const float *res;      // optional residual, same geometry as out
  long res_bstride;
  int res_ld, res_coff;
  int relu;
  const int *counts;     // optional per-batch-item valid row count (TAPS==1)
  int narrow;            // TAPS==1: 16 output channels per workgroup (a lone pair's linear layers: latency, not throughput)
  // fused conv1a (FUSE1A): in = u8 image H x W
  const float *w1a;      // [9][64]
  const float *b1a;      // [64]
  const float *lut;      // [256] u8 -> f32 ( float(u8)/255.0 )
  const int *gate;       // optional: batch items >= gate[0] are skipped (redo pipeline of the guarded fast mode)
  // redo slots that only need the scores of a few cells (gate layout below): a tile is computed only if it lies within
  // t_rad pixels (at this layer's resolution, t_scale pixels per cell) of a target cell; t_scale 0 = no such gating,
  // -1 = the layer is skipped for those slots altogether (descriptor head); t_wc = cells per row
  int t_scale, t_rad, t_wc;
};

// Redo list of the guarded fast mode ("gate", device ints): [0] = slots; [1 + r] = frame of slot r;
// [kGateMode + r] = -1: the whole frame is redone in the exact mode, n >= 0: only the scores of n target cells are needed
// (the top-k cut is the only ambiguous decision); [kGateTargets + 8 r + t] = target cell t (cy * Wc + cx)
constexpr int kGateMax = 64, kGateMode = 1 + kGateMax, kGateTargets = 1 + 2 * kGateMax, kGateInts = kGateTargets + 8 * kGateMax;
constexpr int kAmbMax = 8;     // candidates within the error of the top-k cut that are resolved one by one; more = whole frame

// near-tie guard of the fast precision mode, SuperPoint tail (sp_kernels.hip)
struct SpGuard {
  int *flags;        // [B] bit 0: top-k cut (resolved per candidate), bit 1: threshold band, bit 2: NMS near-tie, bit 3: too many
                     // candidates at the cut (bits 1-3: the whole frame is redone); null = guard off
  int *band;         // [B] scratch: some candidate lies in the threshold band
  int *nms_hi;       // [B] scratch: float bits of the highest window maximum an NMS near-tie involved
  int *amb;          // [B][1 + kAmbMax]: count and pixel indices of the candidates within the error of the cut
  float delta, ulps; // error model of a fast-mode score: delta * s * (1 - s) + ulps * ulp(s)
};

// feature slot (device): header + meta[cap][4] + desc[cap][256], all 4-byte units
constexpr int kCap = 1024;
constexpr int kSlotHeader = 4;  // [K, 0, 0, 0]
constexpr size_t kSlotFloats = kSlotHeader + (size_t)kCap * 4 + (size_t)kCap * 256;

// XCD-aware workgroup map for kernels whose workgroups re-read a per-group operand
// (attention: the QB query tiles of one (image, head) share that head's K/V).
// Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2), so
// the linear id L runs on XCD L % 8; the QB tiles of a group are given ids that are
// congruent mod 8 and adjacent in dispatch order, so the group's operand is fetched
// into ONE L2 once.  Falls back to the plain order when the group count is not a
// multiple of 8.  (MI355X_MICROARCH.md: L2 per XCD, not cross-XCD coherent.)
__device__ __forceinline__ void xcd_group_map(int L, int QB, int G, int &qb, int &group) {
  if ((G & 7) == 0) {
    const int xcd = L & 7, slot = L >> 3;
    group = xcd * (G >> 3) + slot / QB;
    qb = slot % QB;
  } else {
    group = L / QB;
    qb = L % QB;
  }
}

}  // namespace urf
