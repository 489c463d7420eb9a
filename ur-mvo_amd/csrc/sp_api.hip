// sp_api.hip -- host side of the SuperPoint stage behind the C ABI
// (include/urf.h): persistent HIP arena sized at build() (the reference
// allocates and frees device buffers on every infer, buffers.h:227-231), weight
// repack, the kernel pipeline on one HIP stream, pinned staging buffers.
// Mirrors SuperPoint::build / infer (src/super_point.cpp:18-156).
#include "../../include/urf.h"
#include "h2.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <utility>
#include <vector>

namespace urf {

// the last error text is process-wide (not thread-local): the reference calls from a fresh std::thread every time
// (src/tracking.cc:334-335) and reports from whichever thread joins it
static char g_err[512] = "";
static std::mutex g_err_mu;
void set_error(const char *fmt, ...) {
  std::lock_guard<std::mutex> lock(g_err_mu);
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// a reader gets a snapshot taken under the lock (its own thread-local copy): never a message another thread is half-way
// through writing
const char *last_error() {
  static thread_local char snap[sizeof(g_err)];
  std::lock_guard<std::mutex> lock(g_err_mu);
  memcpy(snap, g_err, sizeof(g_err));
  snap[sizeof(snap) - 1] = 0;
  return snap;
}
int g_profiling = 0;
// guarded fast mode, SuperPoint: error model of a fast-mode score, delta * s * (1 - s) + ulps * ulp(s) (sp_kernels.hip);
// measured maxima on both bench streams times a safety factor (DESIGN.md "Guarded fast mode", tools/gpu_margins.py)
static const float kGuardSpDelta = 1.6e-4f, kGuardSpUlps = 8.0f;

int launch_conv(const ConvArgs &a, int taps, bool pool, bool fuse1a, int batch, hipStream_t st);
int launch_softmax(const float *logits, int ld, int Hc, int Wc, float *heat, int B, const int *gate, hipStream_t st);
int launch_nms(const float *heat, uint8_t *mask, uint8_t *supp, float *ss, float *out, int H, int W, int B,
               const int *gate, const SpGuard &g, float thr_lo, hipStream_t st);
int launch_select(const float *scores, int H, int W, double thr, int border, const uint8_t *mask, int *counts,
                  float *cand_score, int *cand_idx, int cand_cap, int *cand_n, int k, float *kp_score, int *kp_idx,
                  int *kp_n, int B, const int *gate, const SpGuard &g, hipStream_t st);
int select_nchunk(int H, int W);
int launch_desc_norm(float *desc, int ld, int coff, int ncell_total, float *out, const int *gate, int ncell_frame,
                     hipStream_t st);
int launch_sample(const float *desc, int Hc, int Wc, const float *kp_score, const int *kp_idx, const int *kp_n,
                  int Ws, double *feat, float *slots, int B, const int *gate, int *kp_n_out, const int *guard_flags,
                  hipStream_t st);
int launch_guard_compact(const int *flags, const int *amb, int B, int Ws, int Wc, const uint8_t *imgs, size_t img_bytes,
                         uint8_t *redo_imgs, int *gate, unsigned long long *stats, hipStream_t st);
int launch_guard_calib(const float *heat_fast, const float *heat_exact, size_t n, float delta, float ulps, float thr_lo,
                       int *out, hipStream_t st);
int launch_guard_resolve(const int *gate, const int *amb, const float *heat_x, int HsWs, float *kp_score, int *kp_idx,
                         const int *kp_n, int B, hipStream_t st);

struct ConvSpec { int cin, cout, k; };
static const ConvSpec kSpConv[12] = {{1, 64, 3},    {64, 64, 3},   {64, 64, 3},   {64, 64, 3},
                                     {64, 128, 3},  {128, 128, 3}, {128, 128, 3}, {128, 128, 3},
                                     {128, 256, 3}, {256, 65, 1},  {128, 256, 3}, {256, 256, 1}};

enum { ST_UPLOAD = 0, ST_CONV1, ST_CONV2A, ST_CONV2B, ST_CONV3A, ST_CONV3B, ST_CONV4A, ST_CONV4B, ST_PADA, ST_PB,
       ST_DB, ST_SOFTMAX, ST_NMS, ST_SELECT, ST_DNORM, ST_SAMPLE, ST_DOWNLOAD, ST_COUNT };

}  // namespace urf

using namespace urf;

struct SpArena {
  float *a1 = nullptr, *a2a = nullptr, *a2b = nullptr, *a3a = nullptr, *a3b = nullptr, *a4a = nullptr,
        *a4b = nullptr, *apd = nullptr, *logits = nullptr, *ddb = nullptr, *desc = nullptr;
  float *heat = nullptr, *scores = nullptr, *ss = nullptr;
  uint8_t *mask = nullptr, *supp = nullptr;
  int *counts = nullptr, *cand_idx = nullptr, *cand_n = nullptr, *kp_idx = nullptr, *kp_n = nullptr;
  float *cand_score = nullptr, *kp_score = nullptr;
};

struct urf_sp {
  urf_sp_config cfg;
  int device = 0;
  hipStream_t st = nullptr;
  bool built = false;
  int maxB = 1, maxH = 0, maxW = 0;
  // weights (device, each tensor 256-byte aligned)
  float *d_wts = nullptr;
  size_t w_off[12], b_off[12];
  size_t wpd_off = 0, bpd_off = 0;  // convPa||convDa concatenated [9][128][512], bias[512]
  size_t wpb_off = 0, bpb_off = 0;  // convPb padded to 68 couts
  size_t lut_off = 0;
  // fast precision mode: conv weights [tap][Cout][Cin] as (hi, lo) f16 planes
  int precision = 0;
  _Float16 *d_wh = nullptr, *d_wl = nullptr;
  size_t hw_off[8];   // conv1b, 2a, 2b, 3a, 3b, 4a, 4b, Pa||Da
  size_t hpb_off = 0, hdb_off = 0;   // fast mode 1x1 heads: convPb [128 (65 used)][256], convDb [256][256] planes
  size_t bpb128_off = 0;             // convPb bias padded to 128
  // activations and selection scratch of a batch: A = the handle's arena; R = the arena of the frames the guarded fast mode
  // (precision 2) redoes in the exact mode
  SpArena A, R;
  // guarded fast mode runs on two streams: the fast pass of batch b + 1 on `st` overlaps the exact pass / resolution / tail of
  // batch b on `stx` (a latency-bound chain of small launches), so consecutive calls alternate between two fast arenas (A is
  // the one of the call being enqueued, A2 the other) and two sets of guard buffers
  SpArena A2;
  hipStream_t stx = nullptr;         // where a call's slots become final; == st unless guarded
  hipEvent_t ev_fast = nullptr;      // fast pass + redo list of the current call enqueued on st
  hipEvent_t ev_tail[2] = {nullptr, nullptr};   // tail of the call that last used arena A / A2 (by parity)
  int parity = 0;
  uint8_t *d_img = nullptr, *d_usermask = nullptr;
  int cand_cap = 0;
  // guarded fast mode: guard words, threshold-band scratch, redo list (gate), images of the frames to redo, counters
  int *g_flags = nullptr, *g_band = nullptr, *g_gate = nullptr, *g_amb = nullptr, *g_nms = nullptr;
  uint8_t *g_img = nullptr;
  int *g2_flags = nullptr, *g2_band = nullptr, *g2_gate = nullptr, *g2_amb = nullptr, *g2_nms = nullptr;   // the other set
  uint8_t *g2_img = nullptr;
  unsigned long long *g_stats = nullptr;
  float g_delta = 0.0f, g_ulps = 0.0f;
  double *d_feat = nullptr;
  float *d_slots = nullptr;
  // pinned host staging
  uint8_t *h_img = nullptr;
  double *h_feat = nullptr;
  int *h_n = nullptr;
  // last call geometry (debug taps)
  int lastH = 0, lastW = 0, lastB = 0;
  // timing
  hipEvent_t evs[4][ST_COUNT + 1];  // ring of sets: the last 4 calls' times stay readable
  hipEvent_t *ev = nullptr;          // set used by the call being enqueued
  int ev_cur = 0;
  int ev_calls = 0;
  float stage_ms[ST_COUNT];
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" const char *urf_last_error(void) { return urf::last_error(); }
// (urf_build_info: build_info.hip, recompiled with every link)
namespace urf {
double build_guard_delta() { return (double)kGuardSpDelta; }
double build_guard_ulps() { return (double)kGuardSpUlps; }
}  // namespace urf
extern "C" int urf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int urf_set_profiling(int enable) { urf::g_profiling = enable; return 0; }
extern "C" size_t urf_slot_bytes(void) { return kSlotFloats * sizeof(float); }

extern "C" int urf_sp_create(const urf_sp_config *cfg, urf_sp **out) {
  URF_CHECK(cfg && out, "urf_sp_create: null argument");
  URF_CHECK(cfg->max_keypoints == -1 || (cfg->max_keypoints >= 0 && cfg->max_keypoints <= URF_MAX_KEYPOINTS),
            "max_keypoints %d outside [-1, %d]", cfg->max_keypoints, URF_MAX_KEYPOINTS);
  int ndev = 0;
  URF_HIP(hipGetDeviceCount(&ndev));
  URF_CHECK(ndev > 0, "no HIP device: liburf_front needs a gfx950 GPU (there is no CPU fallback)");
  URF_CHECK(cfg->device >= 0 && cfg->device < ndev, "device %d out of range (%d devices)", cfg->device, ndev);
  URF_CHECK(cfg->keypoint_threshold >= 0.0, "keypoint_threshold %g is negative", cfg->keypoint_threshold);
  urf_sp *h = new urf_sp();
  h->cfg = *cfg;
  h->device = cfg->device;
  h->maxB = cfg->max_batch > 0 ? cfg->max_batch : 1;
  h->maxH = cfg->max_height > 0 ? cfg->max_height : 1500;
  h->maxW = cfg->max_width > 0 ? cfg->max_width : 1500;
  // precision 3 (strict parity) IS the exact mode for SuperPoint: slots bit-identical to the oracle (scores, order, descriptors)
  h->precision = cfg->precision == 3 ? 0 : cfg->precision;
  if (!(cfg->precision >= 0 && cfg->precision <= 3) || cfg->guard_delta < 0.0f || cfg->guard_ulps < 0.0f) {
    delete h;
    URF_CHECK(false, "precision must be 0 (exact fp32), 1 (fast split-f16), 2 (fast, guarded) or 3 (strict parity = exact here); "
                     "guard_delta / guard_ulps must not be negative");
  }
  *out = h;
  return 0;
}

template <typename T>
static int dalloc(T **p, size_t n) {
  URF_HIP(hipMalloc((void **)p, n * sizeof(T)));
  URF_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

extern "C" int urf_sp_build(urf_sp *h, const float *blob, size_t n_floats) {
  URF_CHECK(h && blob, "urf_sp_build: null argument");
  URF_CHECK(n_floats == URF_SP_BLOB_FLOATS, "SP blob has %zu floats, expected %d", n_floats, URF_SP_BLOB_FLOATS);
  URF_CHECK(!h->built, "urf_sp_build: already built");
  URF_HIP(hipSetDevice(h->device));
  {
    // URF_SP_PRIORITY (experiments): HIP stream priority of the handle's stream (hipDeviceGetStreamPriorityRange; lower = sooner)
    const char *e = urf::exp_env("URF_SP_PRIORITY");
    if (e) URF_HIP(hipStreamCreateWithPriority(&h->st, hipStreamNonBlocking, atoi(e)));
    else URF_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  }
  // ---- repack weights: every tensor 256-B aligned; Pa||Da concatenated; Pb padded
  std::vector<float> host;
  auto put = [&](const float *src, size_t n) {
    size_t off = align_up(host.size(), 64);
    host.resize(off + n, 0.0f);
    if (src) memcpy(host.data() + off, src, n * sizeof(float));
    return off;
  };
  const float *src_w[12], *src_b[12];
  {
    const float *p = blob;
    for (int i = 0; i < 12; ++i) {
      const size_t nw = (size_t)kSpConv[i].k * kSpConv[i].k * kSpConv[i].cin * kSpConv[i].cout;
      src_w[i] = p; p += nw;
      src_b[i] = p; p += kSpConv[i].cout;
    }
  }
  for (int i = 0; i < 12; ++i) {
    const size_t nw = (size_t)kSpConv[i].k * kSpConv[i].k * kSpConv[i].cin * kSpConv[i].cout;
    h->w_off[i] = put(src_w[i], nw);
    h->b_off[i] = put(src_b[i], kSpConv[i].cout);
  }
  {  // convPa || convDa : [9][128][512]
    h->wpd_off = put(nullptr, (size_t)9 * 128 * 512);
    float *d = host.data() + h->wpd_off;
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < 128; ++c) {
        memcpy(d + ((size_t)t * 128 + c) * 512, src_w[8] + ((size_t)t * 128 + c) * 256, 256 * sizeof(float));
        memcpy(d + ((size_t)t * 128 + c) * 512 + 256, src_w[10] + ((size_t)t * 128 + c) * 256, 256 * sizeof(float));
      }
    h->bpd_off = put(nullptr, 512);
    memcpy(host.data() + h->bpd_off, src_b[8], 256 * sizeof(float));
    memcpy(host.data() + h->bpd_off + 256, src_b[10], 256 * sizeof(float));
  }
  {  // convPb padded 65 -> 68 output channels (zeros)
    h->wpb_off = put(nullptr, (size_t)256 * 68);
    float *d = host.data() + h->wpb_off;
    for (int c = 0; c < 256; ++c) memcpy(d + (size_t)c * 68, src_w[9] + (size_t)c * 65, 65 * sizeof(float));
    h->bpb_off = put(nullptr, 68);
    memcpy(host.data() + h->bpb_off, src_b[9], 65 * sizeof(float));
  }
  {  // the fast mode's convPb runs as a 128-wide split-f16 GEMM: bias padded with zeros
    h->bpb128_off = put(nullptr, 128);
    memcpy(host.data() + h->bpb128_off, src_b[9], 65 * sizeof(float));
  }
  {  // u8 -> f32 : float(u8) / 255.0 in double, narrowed (src/super_point.cpp:171-172)
    h->lut_off = put(nullptr, 256);
    for (int v = 0; v < 256; ++v) host[h->lut_off + v] = (float)((double)(float)v / 255.0);
  }
  URF_HIP(hipMalloc((void **)&h->d_wts, host.size() * sizeof(float)));
  URF_HIP(hipMemcpy(h->d_wts, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));

  if (h->precision >= 1) {
    std::vector<_Float16> wh, wl;
    auto putc = [&](const float *w, int cin, int cout, int coff, int ctot, size_t base) {
      // w: [9][cin][cout] fp32 -> planes [9][ctot][cin] at rows coff..coff+cout
      for (int t = 0; t < 9; ++t)
        for (int o = 0; o < cout; ++o)
          for (int c = 0; c < cin; ++c) {
            const float v = w[((size_t)t * cin + c) * cout + o];
            const _Float16 hi = (_Float16)v;
            const size_t idx = base + ((size_t)t * ctot + coff + o) * cin + c;
            wh[idx] = hi;
            wl[idx] = (_Float16)(v - (float)hi);
          }
    };
    for (int i = 1; i <= 7; ++i) {
      const size_t base = wh.size();
      wh.resize(base + (size_t)9 * kSpConv[i].cin * kSpConv[i].cout);
      wl.resize(wh.size());
      putc(src_w[i], kSpConv[i].cin, kSpConv[i].cout, 0, kSpConv[i].cout, base);
      h->hw_off[i - 1] = base;
    }
    {
      const size_t base = wh.size();
      wh.resize(base + (size_t)9 * 128 * 512);
      wl.resize(wh.size());
      putc(src_w[8], 128, 256, 0, 512, base);
      putc(src_w[10], 128, 256, 256, 512, base);
      h->hw_off[7] = base;
    }
    auto put1x1 = [&](const float *w, int cin, int cout, int cpad) {   // w: [cin][cout] fp32 -> planes [cpad][cin]
      const size_t base = wh.size();
      wh.resize(base + (size_t)cpad * cin, (_Float16)0.0f);
      wl.resize(wh.size(), (_Float16)0.0f);
      for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
          const float v = w[(size_t)c * cout + o];
          const _Float16 hi = (_Float16)v;
          wh[base + (size_t)o * cin + c] = hi;
          wl[base + (size_t)o * cin + c] = (_Float16)(v - (float)hi);
        }
      return base;
    };
    h->hpb_off = put1x1(src_w[9], 256, 65, 128);
    h->hdb_off = put1x1(src_w[11], 256, 256, 256);
    URF_HIP(hipMalloc((void **)&h->d_wh, wh.size() * 2));
    URF_HIP(hipMalloc((void **)&h->d_wl, wl.size() * 2));
    URF_HIP(hipMemcpy(h->d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    URF_HIP(hipMemcpy(h->d_wl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
  }
  // ---- arena for the largest frame
  const size_t B = h->maxB, H = h->maxH, W = h->maxW;
  const size_t H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
  const size_t Hs = H8 * 8, Ws = W8 * 8;
  if (dalloc(&h->d_img, B * H * W)) return -1;
  if (dalloc(&h->d_usermask, H * W)) return -1;
  auto arena = [&](SpArena &A, bool fast) -> int {
    if (dalloc(&A.a1, B * H2 * W2 * 64)) return -1;
    if (dalloc(&A.a2a, B * H2 * W2 * 64)) return -1;
    if (dalloc(&A.a2b, B * H4 * W4 * 64)) return -1;
    if (dalloc(&A.a3a, B * H4 * W4 * 128)) return -1;
    if (dalloc(&A.a3b, B * H8 * W8 * 128)) return -1;
    if (dalloc(&A.a4a, B * H8 * W8 * 128)) return -1;
    if (dalloc(&A.a4b, B * H8 * W8 * 128)) return -1;
    if (dalloc(&A.apd, B * H8 * W8 * 512)) return -1;
    if (dalloc(&A.logits, B * H8 * W8 * (fast ? 128 : 68))) return -1;
    if (dalloc(&A.ddb, B * H8 * W8 * 256)) return -1;
    if (dalloc(&A.desc, B * H8 * W8 * 256)) return -1;
    if (dalloc(&A.heat, B * Hs * Ws)) return -1;
    if (dalloc(&A.scores, B * Hs * Ws)) return -1;
    if (dalloc(&A.ss, B * Hs * Ws)) return -1;
    if (dalloc(&A.mask, B * Hs * Ws)) return -1;
    if (dalloc(&A.supp, B * Hs * Ws)) return -1;
    // every pixel can be a candidate (a tied plateau survives simple_nms whole; the reference keeps every candidate
    // before top_k_keypoints, src/super_point.cpp:196-251): 2 x 4 B per pixel and frame
    if (dalloc(&A.counts, B * (size_t)select_nchunk((int)Hs, (int)Ws))) return -1;
    if (dalloc(&A.cand_score, B * (size_t)h->cand_cap)) return -1;
    if (dalloc(&A.cand_idx, B * (size_t)h->cand_cap)) return -1;
    if (dalloc(&A.cand_n, B)) return -1;
    if (dalloc(&A.kp_score, B * (size_t)kCap)) return -1;
    if (dalloc(&A.kp_idx, B * (size_t)kCap)) return -1;
    if (dalloc(&A.kp_n, B)) return -1;
    return 0;
  };
  h->cand_cap = (int)(Hs * Ws);
  if (arena(h->A, h->precision >= 1)) return -1;
  if (h->precision == 2) {
    // guarded fast mode: a second arena for the frames redone in the exact mode (every frame of a batch can be), the guard
    // words and the redo list.  Error model of a fast-mode score (sp_kernels.hip): measured by tools/gpu_margins.py on
    // both bench streams (DESIGN.md "Guarded fast mode"), overridable for experiments.
    const char *e2s = urf::exp_env("URF_SP_TWO_STREAMS");
    const bool two_streams = e2s && atoi(e2s) != 0;
    if (arena(h->R, false) || (two_streams && arena(h->A2, true))) return -1;
    URF_CHECK(B <= (size_t)kGateMax, "guarded fast mode: max_batch %zu above %d", B, kGateMax);
    // (flags | band | nms in one block: one memset per call clears the three)
    if (dalloc(&h->g_flags, 3 * B) || dalloc(&h->g_gate, kGateInts) || dalloc(&h->g_stats, 8) || dalloc(&h->g_amb, B * (1 + kAmbMax)))
      return -1;
    h->g_band = h->g_flags + B; h->g_nms = h->g_flags + 2 * B;
    if (two_streams && (dalloc(&h->g2_flags, 3 * B) || dalloc(&h->g2_gate, kGateInts) ||
                        dalloc(&h->g2_amb, B * (1 + kAmbMax)) || dalloc(&h->g2_img, B * H * W)))
      return -1;
    if (two_streams) { h->g2_band = h->g2_flags + B; h->g2_nms = h->g2_flags + 2 * B; }
    if (dalloc(&h->g_img, B * H * W)) return -1;
    // URF_SP_TWO_STREAMS=1 (A/B runs; measured and NOT the default): the exact pass, the cut resolution and the descriptor tail on
    // a second stream beside the next call's fast pass, with two alternating fast arenas.  Same box, 640x480, frames/s: unguarded
    // 1900-1940; guarded on one stream 1757; on two streams 1508 -- and 1619 even with nothing flagged (1921 on one stream): the
    // second arena set and the cross-stream events cost more than the chain's latency
    if (const char *e2 = urf::exp_env("URF_SP_TWO_STREAMS"); e2 && atoi(e2) != 0) URF_HIP(hipStreamCreateWithFlags(&h->stx, hipStreamNonBlocking));
    URF_HIP(hipEventCreateWithFlags(&h->ev_fast, hipEventDisableTiming));
    URF_HIP(hipEventCreateWithFlags(&h->ev_tail[0], hipEventDisableTiming));
    URF_HIP(hipEventCreateWithFlags(&h->ev_tail[1], hipEventDisableTiming));
    // the error model's constants come from the configuration (urf_sp_config.guard_delta / guard_ulps), never from the environment
    h->g_delta = h->cfg.guard_delta > 0.0f ? h->cfg.guard_delta : kGuardSpDelta;
    h->g_ulps = h->cfg.guard_ulps > 0.0f ? h->cfg.guard_ulps : kGuardSpUlps;
  }
  if (h->precision == 0) {
    // The SuperPoint of a strict-parity pipeline (precision 3 in the configuration): the tail of a call (NMS, selection, descriptor
    // normalisation and sampling: seven launches of a few workgroups, 0.16 ms of latency for next to no chip time) runs on a second
    // stream, beside the NEXT call's convolutions -- which therefore work in a second arena (A, A2 alternate).  The convolutions'
    // stream is the busiest of that pipeline: 1114 against 1072 frames/s at 640x480, 961 against 944 at 1241x376.  NOT in the
    // exact mode proper (precision 0), whose exact matcher streams lose more to the extra concurrency than SuperPoint gains
    // (594 against 653).  URF_SP_TAIL_STREAM = 0 / 1 overrides in the experiments build.
    static const int tail_env = [] { const char *e = urf::exp_env("URF_SP_TAIL_STREAM"); return e ? (atoi(e) != 0) : -1; }();
    const bool tail_stream = tail_env >= 0 ? tail_env != 0 : h->cfg.precision == 3;
    if (tail_stream) {
      if (arena(h->A2, false)) return -1;
      URF_HIP(hipStreamCreateWithFlags(&h->stx, hipStreamNonBlocking));
      URF_HIP(hipEventCreateWithFlags(&h->ev_fast, hipEventDisableTiming));
      URF_HIP(hipEventCreateWithFlags(&h->ev_tail[0], hipEventDisableTiming));
      URF_HIP(hipEventCreateWithFlags(&h->ev_tail[1], hipEventDisableTiming));
    }
  }
  if (dalloc(&h->d_feat, B * (size_t)kCap * 259)) return -1;
  if (dalloc(&h->d_slots, B * kSlotFloats)) return -1;
  URF_HIP(hipHostMalloc((void **)&h->h_img, B * H * W, hipHostMallocDefault));
  URF_HIP(hipHostMalloc((void **)&h->h_feat, B * (size_t)kCap * 259 * sizeof(double), hipHostMallocDefault));
  URF_HIP(hipHostMalloc((void **)&h->h_n, B * sizeof(int), hipHostMallocDefault));
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i <= ST_COUNT; ++i) URF_HIP(hipEventCreate(&h->evs[k][i]));
  h->ev = h->evs[0];
  if (!h->stx) h->stx = h->st;
  // the arena was zeroed with hipMemset on the null stream, which the handle's non-blocking stream does not wait
  // for: without this a first call could run before (or while) its buffers are being cleared
  URF_HIP(hipDeviceSynchronize());
  h->built = true;
  return 0;
}

extern "C" int urf_weights_save(const char *path, int kind, const float *blob, size_t n) {
  FILE *f = fopen(path, "wb");
  URF_CHECK(f, "cannot open %s for writing", path);
  const char magic[4] = {'U', 'R', 'F', 'W'};
  uint32_t k = (uint32_t)kind;
  uint64_t cnt = n;
  bool ok = fwrite(magic, 1, 4, f) == 4 && fwrite(&k, 4, 1, f) == 1 && fwrite(&cnt, 8, 1, f) == 1 &&
            fwrite(blob, sizeof(float), n, f) == n;
  fclose(f);
  URF_CHECK(ok, "short write to %s", path);
  return 0;
}

namespace urf {
int weights_load(const char *path, int kind, std::vector<float> &out) {
  FILE *f = fopen(path, "rb");
  URF_CHECK(f, "cannot open weight file %s", path);
  char magic[4];
  uint32_t k = 0;
  uint64_t cnt = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && fread(&k, 4, 1, f) == 1 && fread(&cnt, 8, 1, f) == 1 &&
            memcmp(magic, "URFW", 4) == 0 && (int)k == kind && cnt < (1ull << 32);
  if (ok) {
    out.resize(cnt);
    ok = fread(out.data(), sizeof(float), cnt, f) == cnt;
  }
  fclose(f);
  URF_CHECK(ok, "%s is not a valid URFW weight file of kind %d", path, kind);
  return 0;
}
}  // namespace urf

extern "C" int urf_sp_build_file(urf_sp *h, const char *path) {
  std::vector<float> blob;
  if (urf::weights_load(path, 1, blob)) return -1;
  return urf_sp_build(h, blob.data(), blob.size());
}

extern "C" void urf_sp_destroy(urf_sp *h) {
  if (!h) return;
  if (h->built) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->st);
    if (h->stx != h->st) {
      (void)hipStreamSynchronize(h->stx);
      (void)hipStreamDestroy(h->stx);
    }
    if (h->ev_fast) { (void)hipEventDestroy(h->ev_fast); (void)hipEventDestroy(h->ev_tail[0]); (void)hipEventDestroy(h->ev_tail[1]); }
    void *bufs[] = {h->d_wh, h->d_wl, h->d_wts, h->d_img, h->d_usermask, h->A.a1, h->A.a2a, h->A.a2b, h->A.a3a, h->A.a3b, h->A.a4a, h->A.a4b, h->A.apd,
                    h->A.logits, h->A.ddb, h->A.desc, h->A.heat, h->A.scores, h->A.ss, h->A.mask, h->A.supp, h->A.counts,
                    h->A.cand_score, h->A.cand_idx, h->A.cand_n, h->A.kp_score, h->A.kp_idx, h->A.kp_n, h->d_feat, h->d_slots};
    for (void *p : bufs) (void)hipFree(p);
    void *rbufs[] = {h->R.a1, h->R.a2a, h->R.a2b, h->R.a3a, h->R.a3b, h->R.a4a, h->R.a4b, h->R.apd, h->R.logits, h->R.ddb, h->R.desc,
                     h->R.heat, h->R.scores, h->R.ss, h->R.mask, h->R.supp, h->R.counts, h->R.cand_score, h->R.cand_idx, h->R.cand_n,
                     h->R.kp_score, h->R.kp_idx, h->R.kp_n, h->g_flags, h->g_gate, h->g_img, h->g_stats, h->g_amb,
                     h->g2_flags, h->g2_gate, h->g2_img, h->g2_amb,
                     h->A2.a1, h->A2.a2a, h->A2.a2b, h->A2.a3a, h->A2.a3b, h->A2.a4a, h->A2.a4b, h->A2.apd, h->A2.logits, h->A2.ddb,
                     h->A2.desc, h->A2.heat, h->A2.scores, h->A2.ss, h->A2.mask, h->A2.supp, h->A2.counts, h->A2.cand_score, h->A2.cand_idx,
                     h->A2.cand_n, h->A2.kp_score, h->A2.kp_idx, h->A2.kp_n};
    for (void *p : rbufs) (void)hipFree(p);
    (void)hipHostFree(h->h_img);
    (void)hipHostFree(h->h_feat);
    (void)hipHostFree(h->h_n);
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i <= ST_COUNT; ++i) (void)hipEventDestroy(h->evs[k][i]);
    (void)hipStreamDestroy(h->st);
  }
  delete h;
}

// the eight 3x3 convolutions in the fast precision mode (h2conv.hip).  Each fp32
// activation buffer of N floats is reused as two f16 planes of N halfs.
static int sp_convs_fast(urf_sp *h, const SpArena &A, int B, const uint8_t *d_imgs, int H, int W, hipStream_t st) {
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
  const float *wt = h->d_wts;
  const bool prof = urf::g_profiling != 0;
  auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->ev[i], st); };
  auto planes = [&](float *buf, size_t n, _Float16 **hi, _Float16 **lo) { *hi = (_Float16 *)buf; *lo = (_Float16 *)buf + n; };
  auto conv = [&](float *in, int cin, int hh, int ww, int widx, const float *bias, int cout, float *out, bool pool,
                  bool outf32) {
    H2ConvArgs a = {};
    _Float16 *ih, *il;
    planes(in, (size_t)B * hh * ww * cin, &ih, &il);
    a.xh = ih; a.xl = il; a.H = hh; a.W = ww; a.Cin = cin;
    a.wh = h->d_wh + h->hw_off[widx]; a.wl = h->d_wl + h->hw_off[widx]; a.bias = bias; a.Cout = cout;
    if (outf32) a.out = out;
    else {
      const size_t no = pool ? (size_t)B * (hh / 2) * (ww / 2) * cout : (size_t)B * hh * ww * cout;
      planes(out, no, &a.oh, &a.ol);
    }
    return launch_h2conv(a, pool, false, outf32, B, st);
  };
  mark(ST_CONV1);
  {
    H2ConvArgs a = {};
    a.H = H; a.W = W; a.Cin = 64; a.wh = h->d_wh + h->hw_off[0]; a.wl = h->d_wl + h->hw_off[0];
    a.bias = wt + h->b_off[1]; a.Cout = 64;
    planes(A.a1, (size_t)B * H2 * W2 * 64, &a.oh, &a.ol);
    a.img = d_imgs; a.w1a = wt + h->w_off[0]; a.b1a = wt + h->b_off[0]; a.lut = wt + h->lut_off;
    if (launch_h2conv(a, true, true, false, B, st)) return -1;
  }
  mark(ST_CONV2A);
  if (conv(A.a1, 64, H2, W2, 1, wt + h->b_off[2], 64, A.a2a, false, false)) return -1;
  mark(ST_CONV2B);
  if (conv(A.a2a, 64, H2, W2, 2, wt + h->b_off[3], 64, A.a2b, true, false)) return -1;
  mark(ST_CONV3A);
  if (conv(A.a2b, 64, H4, W4, 3, wt + h->b_off[4], 128, A.a3a, false, false)) return -1;
  mark(ST_CONV3B);
  if (conv(A.a3a, 128, H4, W4, 4, wt + h->b_off[5], 128, A.a3b, true, false)) return -1;
  mark(ST_CONV4A);
  if (conv(A.a3b, 128, H8, W8, 5, wt + h->b_off[6], 128, A.a4a, false, false)) return -1;
  mark(ST_CONV4B);
  if (conv(A.a4a, 128, H8, W8, 6, wt + h->b_off[7], 128, A.a4b, false, false)) return -1;
  mark(ST_PADA);
  if (conv(A.a4b, 128, H8, W8, 7, wt + h->bpd_off, 512, A.apd, false, false)) return -1;   // planes [cells][512]
  return 0;
}

// The kernel pipeline for B frames already resident in h->d_img (or d_imgs), in arena A.  fast: split-f16 convolutions
// (else exact fp32).  gate != null: the redo pass of the guarded fast mode -- every kernel skips batch items >= gate[0] and
// the results of item r go to the caller's item gate[1 + r].  guard.flags != null: the near-tie guard of the fast pass.
// part: 1 = up to the softmax, 2 = NMS and selection, 4 = descriptor normalisation and sampling (bit mask; 7 = everything).
static int sp_pipeline_on(urf_sp *h, const SpArena &A, bool fast, int B, const uint8_t *d_imgs, int H, int W,
                          const uint8_t *d_mask, double *d_feat, float *d_slots, const int *gate, const SpGuard &guard,
                          int *kp_n_out, bool timed, int part, hipStream_t st) {
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
  const int Hs = H8 * 8, Ws = W8 * 8;
  const float *wt = h->d_wts;
  const bool prof = urf::g_profiling != 0 && timed;
  auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->ev[i], st); };
  // target gating of the redo pass (urf_common.h): pixels per cell at the layer's resolution and the reach, in that
  // layer's pixels, within which a cell's logits depend on the layer's output
  const int W8c = W8;
  const int ncell = H8 * W8;
  auto conv3 = [&](const float *in, int cin, int hh, int ww, const float *w, const float *b, int cout, float *out,
                   bool pool, int t_scale, int t_rad) {
    ConvArgs a = {};
    a.gate = gate; a.t_scale = t_scale; a.t_rad = t_rad; a.t_wc = W8c;
    a.in = in; a.in_ld = cin; a.in_coff = 0; a.in_bstride = (long)hh * ww * cin;
    a.H = hh; a.W = ww; a.Cin = cin; a.w = w; a.bias = b; a.Cout = cout;
    a.out = out; a.out_ld = cout; a.out_coff = 0;
    a.out_bstride = pool ? (long)(hh / 2) * (ww / 2) * cout : (long)hh * ww * cout;
    a.relu = 1;
    return launch_conv(a, 9, pool, false, B, st);
  };
  if (part & 1) {
  if (fast) {
    if (sp_convs_fast(h, A, B, d_imgs, H, W, st)) return -1;
  } else {
  mark(ST_CONV1);
  {  // conv1a (fused, VALU) + conv1b + relu + pool
    ConvArgs a = {};
    a.gate = gate; a.t_scale = 8; a.t_rad = 36; a.t_wc = W8c;
    a.in = d_imgs; a.in_bstride = (long)H * W; a.H = H; a.W = W; a.Cin = 64;
    a.w = wt + h->w_off[1]; a.bias = wt + h->b_off[1]; a.Cout = 64;
    a.out = A.a1; a.out_ld = 64; a.out_bstride = (long)H2 * W2 * 64; a.relu = 1;
    a.w1a = wt + h->w_off[0]; a.b1a = wt + h->b_off[0]; a.lut = wt + h->lut_off;
    if (launch_conv(a, 9, true, true, B, st)) return -1;
  }
  mark(ST_CONV2A);
  if (conv3(A.a1, 64, H2, W2, wt + h->w_off[2], wt + h->b_off[2], 64, A.a2a, false, 4, 17)) return -1;
  mark(ST_CONV2B);
  if (conv3(A.a2a, 64, H2, W2, wt + h->w_off[3], wt + h->b_off[3], 64, A.a2b, true, 4, 16)) return -1;
  mark(ST_CONV3A);
  if (conv3(A.a2b, 64, H4, W4, wt + h->w_off[4], wt + h->b_off[4], 128, A.a3a, false, 2, 7)) return -1;
  mark(ST_CONV3B);
  if (conv3(A.a3a, 128, H4, W4, wt + h->w_off[5], wt + h->b_off[5], 128, A.a3b, true, 2, 6)) return -1;
  mark(ST_CONV4A);
  if (conv3(A.a3b, 128, H8, W8, wt + h->w_off[6], wt + h->b_off[6], 128, A.a4a, false, 1, 2)) return -1;
  mark(ST_CONV4B);
  if (conv3(A.a4a, 128, H8, W8, wt + h->w_off[7], wt + h->b_off[7], 128, A.a4b, false, 1, 1)) return -1;
  mark(ST_PADA);
  if (conv3(A.a4b, 128, H8, W8, wt + h->wpd_off, wt + h->bpd_off, 512, A.apd, false, 1, 0)) return -1;   // (of a target-gated slot only the score head's 256 channels)
  }
  int logit_ld = 68;
  if (fast) {
    // the two 1x1 heads as split-f16 GEMMs on the planes of Pa || Da ([cells][512]: hi plane, then lo plane)
    const _Float16 *ph = (const _Float16 *)A.apd, *pl = ph + (size_t)B * ncell * 512;
    logit_ld = 128;
    mark(ST_PB);
    {
      urf::H2Args a = {};
      a.xh = ph; a.xl = pl; a.ldx = 512; a.x_bstride = (long)ncell * 512; a.rows = ncell; a.Cin = 256;
      a.wh = h->d_wh + h->hpb_off; a.wl = h->d_wl + h->hpb_off; a.bias = wt + h->bpb128_off; a.Cout = 128;
      a.out = A.logits; a.ld_out = 128; a.out_bstride = (long)ncell * 128;
      if (urf::launch_h2gemm(a, B, st)) return -1;
    }
    mark(ST_DB);
    {
      urf::H2Args a = {};
      a.xh = ph + 256; a.xl = pl + 256; a.ldx = 512; a.x_bstride = (long)ncell * 512; a.rows = ncell; a.Cin = 256;
      a.wh = h->d_wh + h->hdb_off; a.wl = h->d_wl + h->hdb_off; a.bias = wt + h->b_off[11]; a.Cout = 256;
      a.out = A.ddb; a.ld_out = 256; a.out_bstride = (long)ncell * 256;
      if (urf::launch_h2gemm(a, B, st)) return -1;
    }
  } else {
  mark(ST_PB);
  {  // convPb 1x1 on channels [0,256) of apd -> logits (68-wide rows)
    ConvArgs a = {};
    a.gate = gate; a.t_scale = 1; a.t_rad = 0; a.t_wc = W8c;
    a.in = A.apd; a.in_ld = 512; a.in_coff = 0; a.in_bstride = (long)ncell * 512;
    a.H = 1; a.W = ncell; a.Cin = 256; a.w = wt + h->wpb_off; a.bias = wt + h->bpb_off; a.Cout = 68;
    a.out = A.logits; a.out_ld = 68; a.out_bstride = (long)ncell * 68; a.relu = 0;
    if (launch_conv(a, 1, false, false, B, st)) return -1;
  }
  mark(ST_DB);
  {  // convDb 1x1 on channels [256,512)
    ConvArgs a = {};
    a.gate = gate; a.t_scale = -1;     // not needed for scores: target-gated slots skip it
    a.in = A.apd; a.in_ld = 512; a.in_coff = 256; a.in_bstride = (long)ncell * 512;
    a.H = 1; a.W = ncell; a.Cin = 256; a.w = wt + h->w_off[11]; a.bias = wt + h->b_off[11]; a.Cout = 256;
    a.out = A.ddb; a.out_ld = 256; a.out_bstride = (long)ncell * 256; a.relu = 0;
    if (launch_conv(a, 1, false, false, B, st)) return -1;
  }
  }
  mark(ST_SOFTMAX);
  if (launch_softmax(A.logits, logit_ld, H8, W8, A.heat, B, gate, st)) return -1;
  }   // part & 1
  if (part & 2) {
  mark(ST_NMS);
  // a maximum below thr_lo can never become a keypoint (nor can anything it suppresses): no near-tie there matters
  const float thr_lo = (float)(h->cfg.keypoint_threshold * 0.5);
  if (launch_nms(A.heat, A.mask, A.supp, A.ss, A.scores, Hs, Ws, B, gate, guard, thr_lo, st)) return -1;
  mark(ST_SELECT);
  if (launch_select(A.scores, Hs, Ws, h->cfg.keypoint_threshold, h->cfg.remove_borders, d_mask, A.counts,
                    A.cand_score, A.cand_idx, h->cand_cap, A.cand_n, h->cfg.max_keypoints, A.kp_score,
                    A.kp_idx, A.kp_n, B, gate, guard, st))
    return -1;
  }   // part & 2
  if (!(part & 4)) return 0;
  mark(ST_DNORM);
  if (launch_desc_norm(A.ddb, 256, 0, B * ncell, A.desc, gate, ncell, st)) return -1;
  mark(ST_SAMPLE);
  if (launch_sample(A.desc, H8, W8, A.kp_score, A.kp_idx, A.kp_n, Ws, d_feat, d_slots, B, gate, kp_n_out,
                    (h->precision == 2 && !gate) ? h->g_flags : nullptr, st))
    return -1;
  mark(ST_DOWNLOAD);
  return 0;
}

// precision 0 / 1: one pass.  precision 2 (guarded fast): the fast pass up to the top-k selection with the near-tie guard;
// the redo list; the exact pass over it -- enqueued unconditionally, its kernels exit at once for frames that are not on the
// list and compute only the tiles that reach a target cell for frames whose sole ambiguity is the top-k cut; the
// per-candidate resolution of those cuts; then the descriptor tail of the fast pass, and of the exact pass for the frames
// redone whole.
static int sp_pipeline(urf_sp *h, int B, const uint8_t *d_imgs, int H, int W, const uint8_t *d_mask, double *d_feat,
                       float *d_slots) {
  SpGuard g = {};
  h->lastH = H; h->lastW = W; h->lastB = B;
  if (h->precision != 2) {
    if (h->stx == h->st)
      return sp_pipeline_on(h, h->A, h->precision == 1, B, d_imgs, H, W, d_mask, d_feat, d_slots, nullptr, g, nullptr, true, 7, h->st);
    // exact mode with the tail on its own stream: this call's arena is the one the call before last used
    std::swap(h->A, h->A2);
    h->parity ^= 1;
    URF_HIP(hipStreamWaitEvent(h->st, h->ev_tail[h->parity], 0));     // (never recorded: no wait)
    if (sp_pipeline_on(h, h->A, false, B, d_imgs, H, W, d_mask, d_feat, d_slots, nullptr, g, nullptr, true, 1, h->st)) return -1;
    URF_HIP(hipEventRecord(h->ev_fast, h->st));
    URF_HIP(hipStreamWaitEvent(h->stx, h->ev_fast, 0));
    if (sp_pipeline_on(h, h->A, false, B, d_imgs, H, W, d_mask, d_feat, d_slots, nullptr, g, nullptr, true, 6, h->stx)) return -1;
    URF_HIP(hipEventRecord(h->ev_tail[h->parity], h->stx));
    return 0;
  }
  hipStream_t st = h->st, sx = h->stx;
  const bool two = sx != st;
  if (two) {
    // this call's fast arena and guard buffers: the other set is still being read by the previous call's chain on stx
    std::swap(h->A, h->A2);
    std::swap(h->g_flags, h->g2_flags); std::swap(h->g_band, h->g2_band); std::swap(h->g_gate, h->g2_gate);
    std::swap(h->g_amb, h->g2_amb); std::swap(h->g_nms, h->g2_nms); std::swap(h->g_img, h->g2_img);
    h->parity ^= 1;
    URF_HIP(hipStreamWaitEvent(st, h->ev_tail[h->parity], 0));   // the call two back has left this set (never recorded: no wait)
  }
  const int Hs = H / 8 * 8, Ws = W / 8 * 8;
  g.flags = h->g_flags; g.band = h->g_band; g.amb = h->g_amb; g.nms_hi = h->g_nms; g.delta = h->g_delta; g.ulps = h->g_ulps;
  SpGuard off = {};
  URF_HIP(hipMemsetAsync(h->g_flags, 0, 3 * (size_t)h->maxB * sizeof(int), st));   // flags | band | nms
  if (sp_pipeline_on(h, h->A, true, B, d_imgs, H, W, d_mask, d_feat, d_slots, nullptr, g, nullptr, true, 3, st)) return -1;
  if (launch_guard_compact(h->g_flags, h->g_amb, B, Ws, Ws / 8, d_imgs, (size_t)H * W, h->g_img, h->g_gate, h->g_stats, st)) return -1;
  if (two) {
    URF_HIP(hipEventRecord(h->ev_fast, st));
    URF_HIP(hipStreamWaitEvent(sx, h->ev_fast, 0));
  }
  // from here on `stx` (two-stream variant: the next call's fast pass may start on `st` right away)
  if (sp_pipeline_on(h, h->R, false, B, h->g_img, H, W, d_mask, d_feat, d_slots, h->g_gate, off, nullptr, false, 3, sx)) return -1;
  if (launch_guard_resolve(h->g_gate, h->g_amb, h->R.heat, Hs * Ws, h->A.kp_score, h->A.kp_idx, h->A.kp_n, B, sx)) return -1;
  if (sp_pipeline_on(h, h->A, true, B, d_imgs, H, W, d_mask, d_feat, d_slots, nullptr, off, nullptr, true, 4, sx)) return -1;
  if (sp_pipeline_on(h, h->R, false, B, h->g_img, H, W, d_mask, d_feat, d_slots, h->g_gate, off, h->A.kp_n, false, 4, sx)) return -1;
  if (two) URF_HIP(hipEventRecord(h->ev_tail[h->parity], sx));
  return 0;
}

static int sp_check_dims(urf_sp *h, int B, int rows, int cols) {
  URF_CHECK(h && h->built, "SuperPoint handle is not built");
  URF_CHECK(B >= 1 && B <= h->maxB, "batch %d outside [1, %d]", B, h->maxB);
  URF_CHECK(rows >= 16 && cols >= 16 && rows <= h->maxH && cols <= h->maxW,
            "image %dx%d outside [16, %dx%d] (arena sized at build)", rows, cols, h->maxH, h->maxW);
  return 0;
}

static void sp_flip_events(urf_sp *h) {
  if (!urf::g_profiling) return;
  h->ev_cur = (h->ev_cur + 1) & 3;
  h->ev = h->evs[h->ev_cur];
  h->ev_calls++;
}
static void sp_collect_times(urf_sp *) {}

extern "C" int urf_sp_infer_batch(urf_sp *h, int B, const uint8_t *const *imgs, int rows, int cols, size_t step,
                                  double *feat, int cap, int *Kout) {
  if (sp_check_dims(h, B, rows, cols)) return -2;
  URF_CHECK(imgs && feat && Kout && cap >= 1, "urf_sp_infer_batch: bad argument");
  URF_HIP(hipSetDevice(h->device));
  const size_t fsz = (size_t)rows * cols;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < rows; ++y) memcpy(h->h_img + b * fsz + (size_t)y * cols, imgs[b] + (size_t)y * step, cols);
  const bool prof = urf::g_profiling != 0;
  sp_flip_events(h);
  if (prof) (void)hipEventRecord(h->ev[ST_UPLOAD], h->st);
  URF_HIP(hipMemcpyAsync(h->d_img, h->h_img, B * fsz, hipMemcpyHostToDevice, h->st));
  if (sp_pipeline(h, B, h->d_img, rows, cols, nullptr, h->d_feat, h->d_slots)) return -1;
  URF_HIP(hipMemcpyAsync(h->h_n, h->A.kp_n, B * sizeof(int), hipMemcpyDeviceToHost, h->stx));
  URF_HIP(hipMemcpyAsync(h->h_feat, h->d_feat, (size_t)B * kCap * 259 * sizeof(double), hipMemcpyDeviceToHost, h->stx));
  if (prof) (void)hipEventRecord(h->ev[ST_COUNT], h->stx);
  URF_HIP(hipStreamSynchronize(h->stx));
  sp_collect_times(h);
  for (int b = 0; b < B; ++b) URF_CHECK(h->h_n[b] <= cap, "feature buffer too small: K=%d > cap=%d", h->h_n[b], cap);
  for (int b = 0; b < B; ++b) {
    Kout[b] = h->h_n[b];
    memcpy(feat + (size_t)b * 259 * cap, h->h_feat + (size_t)b * kCap * 259, (size_t)h->h_n[b] * 259 * sizeof(double));
  }
  return 0;
}

extern "C" int urf_sp_infer(urf_sp *h, const uint8_t *img, int rows, int cols, size_t step, const uint8_t *mask,
                            size_t mstep, double *feat, int cap, int *K) {
  if (sp_check_dims(h, 1, rows, cols)) return -2;
  URF_CHECK(img && feat && K && cap >= 1, "urf_sp_infer: bad argument");
  if (!mask) return urf_sp_infer_batch(h, 1, &img, rows, cols, step, feat, cap, K);
  URF_HIP(hipSetDevice(h->device));
  const int Hs = rows / 8 * 8, Ws = cols / 8 * 8;
  const size_t fsz = (size_t)rows * cols;
  for (int y = 0; y < rows; ++y) memcpy(h->h_img + (size_t)y * cols, img + (size_t)y * step, cols);
  // mask is indexed on the heat-map grid (Hs x Ws), tight copy
  std::vector<uint8_t> m((size_t)Hs * Ws);
  for (int y = 0; y < Hs; ++y) memcpy(m.data() + (size_t)y * Ws, mask + (size_t)y * mstep, Ws);
  URF_HIP(hipMemcpyAsync(h->d_img, h->h_img, fsz, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->d_usermask, m.data(), m.size(), hipMemcpyHostToDevice, h->st));
  URF_HIP(hipStreamSynchronize(h->st));
  sp_flip_events(h);
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_UPLOAD], h->st);
  if (sp_pipeline(h, 1, h->d_img, rows, cols, h->d_usermask, h->d_feat, h->d_slots)) return -1;
  URF_HIP(hipMemcpyAsync(h->h_n, h->A.kp_n, sizeof(int), hipMemcpyDeviceToHost, h->stx));
  URF_HIP(hipMemcpyAsync(h->h_feat, h->d_feat, (size_t)kCap * 259 * sizeof(double), hipMemcpyDeviceToHost, h->stx));
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_COUNT], h->stx);
  URF_HIP(hipStreamSynchronize(h->stx));
  sp_collect_times(h);
  URF_CHECK(h->h_n[0] <= cap, "feature buffer too small: K=%d > cap=%d", h->h_n[0], cap);
  *K = h->h_n[0];
  memcpy(feat, h->h_feat, (size_t)h->h_n[0] * 259 * sizeof(double));
  return 0;
}

extern "C" int urf_sp_infer_device(urf_sp *h, int B, const uint8_t *d_imgs, int rows, int cols, void *d_slots) {
  if (sp_check_dims(h, B, rows, cols)) return -2;
  URF_CHECK(d_imgs && d_slots, "urf_sp_infer_device: null pointer");
  URF_HIP(hipSetDevice(h->device));
  sp_flip_events(h);
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_UPLOAD], h->st);
  if (sp_pipeline(h, B, d_imgs, rows, cols, nullptr, nullptr, (float *)d_slots)) return -1;
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_COUNT], h->stx);
  return 0;
}

extern "C" int urf_sp_sync(urf_sp *h) {
  URF_CHECK(h && h->built, "SuperPoint handle is not built");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  if (h->stx != h->st) URF_HIP(hipStreamSynchronize(h->stx));
  sp_collect_times(h);
  return 0;
}

extern "C" int urf_sp_near_tie_reruns(urf_sp *h, unsigned long long *out, int n) {
  URF_CHECK(h && h->built && out && n >= 1, "urf_sp_near_tie_reruns: bad argument");
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (h->precision == 2) {
    URF_HIP(hipSetDevice(h->device));
    URF_HIP(hipStreamSynchronize(h->st));
    URF_HIP(hipMemcpy(v, h->g_stats, sizeof(v), hipMemcpyDeviceToHost));
  }
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return 0;
}

// Guard calibration (include/urf.h): the fast and the exact pass up to the heat map on the caller's frames, the error model's
// two constants widened where those frames need it.  Never narrows them.
static int sp_calibrate(urf_sp *h, int B, const uint8_t *d_imgs, int rows, int cols, double *out) {
  URF_CHECK(h->precision == 2, "urf_sp_calibrate_guard: the handle is not in the guarded fast mode (precision 2)");
  const int Hs = rows / 8 * 8, Ws = cols / 8 * 8;
  const size_t n = (size_t)B * Hs * Ws;
  const SpGuard off = {};
  hipStream_t st = h->st;
  URF_HIP(hipStreamSynchronize(h->stx));     // (two-stream variant: the redo arena may still be in use)
  if (sp_pipeline_on(h, h->A, true, B, d_imgs, rows, cols, nullptr, h->d_feat, h->d_slots, nullptr, off, nullptr, false, 1, st)) return -1;
  if (sp_pipeline_on(h, h->R, false, B, d_imgs, rows, cols, nullptr, h->d_feat, h->d_slots, nullptr, off, nullptr, false, 1, st)) return -1;
  const float margin = 1.10f;                // head-room over what the frames needed (the built-in delta has 14 % over its own measurement)
  int *acc = h->g_gate;                      // (scratch: rebuilt by every call of the pipeline)
  float need[2] = {0.0f, 0.0f}, first_c = 0.0f;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: the c that the frames need beside the present delta (saturated scores: delta's term vanishes there);
    // pass 1: the delta they need beside the c that results
    URF_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(int), st));
    if (launch_guard_calib(h->A.heat, h->R.heat, n, h->g_delta, h->g_ulps, (float)(h->cfg.keypoint_threshold * 0.5), acc, st)) return -1;
    URF_HIP(hipMemcpyAsync(need, acc, sizeof(need), hipMemcpyDeviceToHost, st));
    URF_HIP(hipStreamSynchronize(st));
    if (pass == 0) {
      first_c = need[1];
      if (need[1] * margin > h->g_ulps) h->g_ulps = need[1] * margin;
    } else if (need[0] * margin > h->g_delta) {
      h->g_delta = need[0] * margin;
    }
  }
  if (out) { out[0] = need[0]; out[1] = first_c; out[2] = h->g_delta; out[3] = h->g_ulps; }
  return 0;
}

extern "C" int urf_sp_calibrate_guard_device(urf_sp *h, int B, const uint8_t *d_imgs, int rows, int cols, double *out) {
  if (sp_check_dims(h, B, rows, cols)) return -2;
  URF_CHECK(d_imgs, "urf_sp_calibrate_guard_device: null image pointer");
  URF_HIP(hipSetDevice(h->device));
  return sp_calibrate(h, B, d_imgs, rows, cols, out);
}

extern "C" int urf_sp_calibrate_guard(urf_sp *h, int B, const uint8_t *const *imgs, int rows, int cols, size_t step, double *out) {
  if (sp_check_dims(h, B, rows, cols)) return -2;
  URF_CHECK(imgs, "urf_sp_calibrate_guard: null image pointer");
  URF_HIP(hipSetDevice(h->device));
  const size_t fsz = (size_t)rows * cols;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < rows; ++y) memcpy(h->h_img + b * fsz + (size_t)y * cols, imgs[b] + (size_t)y * step, cols);
  URF_HIP(hipMemcpyAsync(h->d_img, h->h_img, B * fsz, hipMemcpyHostToDevice, h->st));
  return sp_calibrate(h, B, h->d_img, rows, cols, out);
}

extern "C" int urf_slot_to_host(const void *d_slot, double *feat, int cap, int *K) {
  URF_CHECK(d_slot && feat && K, "urf_slot_to_host: null pointer");
  std::vector<float> s(kSlotFloats);
  URF_HIP(hipMemcpy(s.data(), d_slot, kSlotFloats * sizeof(float), hipMemcpyDeviceToHost));
  int n;
  memcpy(&n, s.data(), 4);
  URF_CHECK(n >= 0 && n <= kCap && n <= cap, "slot holds %d keypoints, cap %d", n, cap);
  *K = n;
  const float *meta = s.data() + kSlotHeader, *desc = meta + 4 * (size_t)kCap;
  for (int j = 0; j < n; ++j) {
    double *col = feat + (size_t)259 * j;
    col[0] = meta[4 * j]; col[1] = meta[4 * j + 1]; col[2] = meta[4 * j + 2];
    for (int c = 0; c < 256; ++c) col[3 + c] = desc[(size_t)j * 256 + c];
  }
  return 0;
}

extern "C" int urf_sp_debug_tensor(urf_sp *h, int which, float *out, size_t n) {
  URF_CHECK(h && h->built && out, "urf_sp_debug_tensor: bad handle");
  URF_HIP(hipSetDevice(h->device));
  const float *src = nullptr;
  switch (which) {
    case 0: src = h->A.scores; break;
    case 1: src = h->A.heat; break;
    case 2: src = h->A.desc; break;
    case 101: src = h->A.a1; break;
    case 102: src = h->A.a2a; break;
    case 103: src = h->A.a2b; break;
    case 104: src = h->A.a3a; break;
    case 105: src = h->A.a3b; break;
    case 106: src = h->A.a4a; break;
    case 107: src = h->A.a4b; break;
    case 108: src = h->A.apd; break;     /* [cells][512]: Pa | Da */
    case 109: src = h->A.logits; break;  /* [cells][68] */
    case 111: src = h->A.ddb; break;
    default: URF_CHECK(false, "unknown debug tensor %d", which);
  }
  URF_HIP(hipStreamSynchronize(h->st));
  if (h->stx != h->st) URF_HIP(hipStreamSynchronize(h->stx));
  URF_HIP(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// Stage times of the call `age` calls ago (0 = latest, up to 3); that call must
// have completed.  urf_sp_stage_ms(h, ms, n) = age 0; n < 0 = age 1 with |n| entries.
extern "C" int urf_sp_stage_ms_age(urf_sp *h, float *ms, int n, int age) {
  URF_CHECK(h && ms && h->built, "urf_sp_stage_ms: bad handle");
  URF_CHECK(age >= 0 && age <= 3 && h->ev_calls > age, "no timed call of age %d (urf_set_profiling(1) before the call)", age);
  URF_HIP(hipSetDevice(h->device));
  hipEvent_t *e = h->evs[(h->ev_cur - age) & 3];
  for (int i = 0; i < n && i < ST_COUNT; ++i) {
    float t = 0.0f;
    hipError_t rc = hipEventElapsedTime(&t, e[i], e[i + 1]);
    URF_CHECK(rc == hipSuccess, "stage %d of the requested call has not completed: %s", i, hipGetErrorString(rc));
    ms[i] = t;
  }
  return ST_COUNT;
}
extern "C" int urf_sp_stage_ms(urf_sp *h, float *ms, int n) {
  return n < 0 ? urf_sp_stage_ms_age(h, ms, -n, 1) : urf_sp_stage_ms_age(h, ms, n, 0);
}

extern "C" void *urf_sp_stream(urf_sp *h) { return h && h->built ? (void *)h->st : nullptr; }
extern "C" void *urf_sp_result_stream(urf_sp *h) { return h && h->built ? (void *)h->stx : nullptr; }
