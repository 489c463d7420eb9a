// sp_api.hip -- host side of the SuperPoint stage behind the C ABI
// (include/urf.h): persistent HIP arena sized at build() (the reference
// allocates and frees device buffers on every infer, buffers.h:227-231), weight
// repack, the kernel pipeline on one HIP stream, pinned staging buffers.
// Mirrors SuperPoint::build / infer (src/super_point.cpp:18-156).
#include "../../include/urf.h"
#include "h2.h"

#include <stdarg.h>
#include <string.h>

#include <mutex>
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
const char *last_error() { return g_err; }
int g_profiling = 0;

int launch_conv(const ConvArgs &a, int taps, bool pool, bool fuse1a, int batch, hipStream_t st);
int launch_softmax(const float *logits, int ld, int Hc, int Wc, float *heat, int B, hipStream_t st);
int launch_nms(const float *heat, uint8_t *mask, uint8_t *supp, float *ss, float *out, int H, int W, int B,
               hipStream_t st);
int launch_select(const float *scores, int H, int W, double thr, int border, const uint8_t *mask, int *counts,
                  float *cand_score, int *cand_idx, int cand_cap, int *cand_n, int k, float *kp_score, int *kp_idx,
                  int *kp_n, int B, hipStream_t st);
int select_nchunk(int H, int W);
int launch_desc_norm(float *desc, int ld, int coff, int ncell_total, float *out, hipStream_t st);
int launch_sample(const float *desc, int Hc, int Wc, const float *kp_score, const int *kp_idx, const int *kp_n,
                  int Ws, double *feat, float *slots, int B, hipStream_t st);

struct ConvSpec { int cin, cout, k; };
static const ConvSpec kSpConv[12] = {{1, 64, 3},    {64, 64, 3},   {64, 64, 3},   {64, 64, 3},
                                     {64, 128, 3},  {128, 128, 3}, {128, 128, 3}, {128, 128, 3},
                                     {128, 256, 3}, {256, 65, 1},  {128, 256, 3}, {256, 256, 1}};

enum { ST_UPLOAD = 0, ST_CONV1, ST_CONV2A, ST_CONV2B, ST_CONV3A, ST_CONV3B, ST_CONV4A, ST_CONV4B, ST_PADA, ST_PB,
       ST_DB, ST_SOFTMAX, ST_NMS, ST_SELECT, ST_DNORM, ST_SAMPLE, ST_DOWNLOAD, ST_COUNT };

}  // namespace urf

using namespace urf;

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
  // activations
  float *a1 = nullptr, *a2a = nullptr, *a2b = nullptr, *a3a = nullptr, *a3b = nullptr, *a4a = nullptr,
        *a4b = nullptr, *apd = nullptr, *logits = nullptr, *ddb = nullptr, *desc = nullptr;
  float *heat = nullptr, *scores = nullptr, *ss = nullptr;
  uint8_t *mask = nullptr, *supp = nullptr, *d_img = nullptr, *d_usermask = nullptr;
  int *counts = nullptr, *cand_idx = nullptr, *cand_n = nullptr, *kp_idx = nullptr, *kp_n = nullptr;
  float *cand_score = nullptr, *kp_score = nullptr;
  int cand_cap = 0;
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
  h->precision = cfg->precision;
  URF_CHECK(h->precision == 0 || h->precision == 1, "precision must be 0 (exact fp32) or 1 (fast split-f16)");
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
  URF_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
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

  if (h->precision == 1) {
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
  if (dalloc(&h->a1, B * H2 * W2 * 64)) return -1;
  if (dalloc(&h->a2a, B * H2 * W2 * 64)) return -1;
  if (dalloc(&h->a2b, B * H4 * W4 * 64)) return -1;
  if (dalloc(&h->a3a, B * H4 * W4 * 128)) return -1;
  if (dalloc(&h->a3b, B * H8 * W8 * 128)) return -1;
  if (dalloc(&h->a4a, B * H8 * W8 * 128)) return -1;
  if (dalloc(&h->a4b, B * H8 * W8 * 128)) return -1;
  if (dalloc(&h->apd, B * H8 * W8 * 512)) return -1;
  if (dalloc(&h->logits, B * H8 * W8 * (h->precision == 1 ? 128 : 68))) return -1;
  if (dalloc(&h->ddb, B * H8 * W8 * 256)) return -1;
  if (dalloc(&h->desc, B * H8 * W8 * 256)) return -1;
  if (dalloc(&h->heat, B * Hs * Ws)) return -1;
  if (dalloc(&h->scores, B * Hs * Ws)) return -1;
  if (dalloc(&h->ss, B * Hs * Ws)) return -1;
  if (dalloc(&h->mask, B * Hs * Ws)) return -1;
  if (dalloc(&h->supp, B * Hs * Ws)) return -1;
  // every pixel can be a candidate (a tied plateau survives simple_nms whole; the reference keeps every candidate
  // before top_k_keypoints, src/super_point.cpp:196-251): 2 x 4 B per pixel and frame
  h->cand_cap = (int)(Hs * Ws);
  if (dalloc(&h->counts, B * (size_t)select_nchunk((int)Hs, (int)Ws))) return -1;
  if (dalloc(&h->cand_score, B * (size_t)h->cand_cap)) return -1;
  if (dalloc(&h->cand_idx, B * (size_t)h->cand_cap)) return -1;
  if (dalloc(&h->cand_n, B)) return -1;
  if (dalloc(&h->kp_score, B * (size_t)kCap)) return -1;
  if (dalloc(&h->kp_idx, B * (size_t)kCap)) return -1;
  if (dalloc(&h->kp_n, B)) return -1;
  if (dalloc(&h->d_feat, B * (size_t)kCap * 259)) return -1;
  if (dalloc(&h->d_slots, B * kSlotFloats)) return -1;
  URF_HIP(hipHostMalloc((void **)&h->h_img, B * H * W, hipHostMallocDefault));
  URF_HIP(hipHostMalloc((void **)&h->h_feat, B * (size_t)kCap * 259 * sizeof(double), hipHostMallocDefault));
  URF_HIP(hipHostMalloc((void **)&h->h_n, B * sizeof(int), hipHostMallocDefault));
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i <= ST_COUNT; ++i) URF_HIP(hipEventCreate(&h->evs[k][i]));
  h->ev = h->evs[0];
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
    void *bufs[] = {h->d_wh, h->d_wl, h->d_wts, h->d_img, h->d_usermask, h->a1, h->a2a, h->a2b, h->a3a, h->a3b, h->a4a, h->a4b, h->apd,
                    h->logits, h->ddb, h->desc, h->heat, h->scores, h->ss, h->mask, h->supp, h->counts,
                    h->cand_score, h->cand_idx, h->cand_n, h->kp_score, h->kp_idx, h->kp_n, h->d_feat, h->d_slots};
    for (void *p : bufs) (void)hipFree(p);
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
static int sp_convs_fast(urf_sp *h, int B, const uint8_t *d_imgs, int H, int W) {
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
  hipStream_t st = h->st;
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
    planes(h->a1, (size_t)B * H2 * W2 * 64, &a.oh, &a.ol);
    a.img = d_imgs; a.w1a = wt + h->w_off[0]; a.b1a = wt + h->b_off[0]; a.lut = wt + h->lut_off;
    if (launch_h2conv(a, true, true, false, B, st)) return -1;
  }
  mark(ST_CONV2A);
  if (conv(h->a1, 64, H2, W2, 1, wt + h->b_off[2], 64, h->a2a, false, false)) return -1;
  mark(ST_CONV2B);
  if (conv(h->a2a, 64, H2, W2, 2, wt + h->b_off[3], 64, h->a2b, true, false)) return -1;
  mark(ST_CONV3A);
  if (conv(h->a2b, 64, H4, W4, 3, wt + h->b_off[4], 128, h->a3a, false, false)) return -1;
  mark(ST_CONV3B);
  if (conv(h->a3a, 128, H4, W4, 4, wt + h->b_off[5], 128, h->a3b, true, false)) return -1;
  mark(ST_CONV4A);
  if (conv(h->a3b, 128, H8, W8, 5, wt + h->b_off[6], 128, h->a4a, false, false)) return -1;
  mark(ST_CONV4B);
  if (conv(h->a4a, 128, H8, W8, 6, wt + h->b_off[7], 128, h->a4b, false, false)) return -1;
  mark(ST_PADA);
  if (conv(h->a4b, 128, H8, W8, 7, wt + h->bpd_off, 512, h->apd, false, false)) return -1;   // planes [cells][512]
  return 0;
}

// The kernel pipeline for B frames already resident in h->d_img (or d_imgs).
static int sp_pipeline(urf_sp *h, int B, const uint8_t *d_imgs, int H, int W, const uint8_t *d_mask, double *d_feat,
                       float *d_slots) {
  const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, H8 = H4 / 2, W8 = W4 / 2;
  const int Hs = H8 * 8, Ws = W8 * 8;
  hipStream_t st = h->st;
  const float *wt = h->d_wts;
  const bool prof = urf::g_profiling != 0;
  auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->ev[i], st); };
  auto conv3 = [&](const float *in, int cin, int hh, int ww, const float *w, const float *b, int cout, float *out,
                   bool pool) {
    ConvArgs a = {};
    a.in = in; a.in_ld = cin; a.in_coff = 0; a.in_bstride = (long)hh * ww * cin;
    a.H = hh; a.W = ww; a.Cin = cin; a.w = w; a.bias = b; a.Cout = cout;
    a.out = out; a.out_ld = cout; a.out_coff = 0;
    a.out_bstride = pool ? (long)(hh / 2) * (ww / 2) * cout : (long)hh * ww * cout;
    a.relu = 1;
    return launch_conv(a, 9, pool, false, B, st);
  };
  if (h->precision == 1) {
    if (sp_convs_fast(h, B, d_imgs, H, W)) return -1;
  } else {
  mark(ST_CONV1);
  {  // conv1a (fused, VALU) + conv1b + relu + pool
    ConvArgs a = {};
    a.in = d_imgs; a.in_bstride = (long)H * W; a.H = H; a.W = W; a.Cin = 64;
    a.w = wt + h->w_off[1]; a.bias = wt + h->b_off[1]; a.Cout = 64;
    a.out = h->a1; a.out_ld = 64; a.out_bstride = (long)H2 * W2 * 64; a.relu = 1;
    a.w1a = wt + h->w_off[0]; a.b1a = wt + h->b_off[0]; a.lut = wt + h->lut_off;
    if (launch_conv(a, 9, true, true, B, st)) return -1;
  }
  mark(ST_CONV2A);
  if (conv3(h->a1, 64, H2, W2, wt + h->w_off[2], wt + h->b_off[2], 64, h->a2a, false)) return -1;
  mark(ST_CONV2B);
  if (conv3(h->a2a, 64, H2, W2, wt + h->w_off[3], wt + h->b_off[3], 64, h->a2b, true)) return -1;
  mark(ST_CONV3A);
  if (conv3(h->a2b, 64, H4, W4, wt + h->w_off[4], wt + h->b_off[4], 128, h->a3a, false)) return -1;
  mark(ST_CONV3B);
  if (conv3(h->a3a, 128, H4, W4, wt + h->w_off[5], wt + h->b_off[5], 128, h->a3b, true)) return -1;
  mark(ST_CONV4A);
  if (conv3(h->a3b, 128, H8, W8, wt + h->w_off[6], wt + h->b_off[6], 128, h->a4a, false)) return -1;
  mark(ST_CONV4B);
  if (conv3(h->a4a, 128, H8, W8, wt + h->w_off[7], wt + h->b_off[7], 128, h->a4b, false)) return -1;
  mark(ST_PADA);
  if (conv3(h->a4b, 128, H8, W8, wt + h->wpd_off, wt + h->bpd_off, 512, h->apd, false)) return -1;
  }
  const int ncell = H8 * W8;
  int logit_ld = 68;
  if (h->precision == 1) {
    // the two 1x1 heads as split-f16 GEMMs on the planes of Pa || Da ([cells][512]: hi plane, then lo plane)
    const _Float16 *ph = (const _Float16 *)h->apd, *pl = ph + (size_t)B * ncell * 512;
    logit_ld = 128;
    mark(ST_PB);
    {
      urf::H2Args a = {};
      a.xh = ph; a.xl = pl; a.ldx = 512; a.x_bstride = (long)ncell * 512; a.rows = ncell; a.Cin = 256;
      a.wh = h->d_wh + h->hpb_off; a.wl = h->d_wl + h->hpb_off; a.bias = wt + h->bpb128_off; a.Cout = 128;
      a.out = h->logits; a.ld_out = 128; a.out_bstride = (long)ncell * 128;
      if (urf::launch_h2gemm(a, B, st)) return -1;
    }
    mark(ST_DB);
    {
      urf::H2Args a = {};
      a.xh = ph + 256; a.xl = pl + 256; a.ldx = 512; a.x_bstride = (long)ncell * 512; a.rows = ncell; a.Cin = 256;
      a.wh = h->d_wh + h->hdb_off; a.wl = h->d_wl + h->hdb_off; a.bias = wt + h->b_off[11]; a.Cout = 256;
      a.out = h->ddb; a.ld_out = 256; a.out_bstride = (long)ncell * 256;
      if (urf::launch_h2gemm(a, B, st)) return -1;
    }
  } else {
  mark(ST_PB);
  {  // convPb 1x1 on channels [0,256) of apd -> logits (68-wide rows)
    ConvArgs a = {};
    a.in = h->apd; a.in_ld = 512; a.in_coff = 0; a.in_bstride = (long)ncell * 512;
    a.H = 1; a.W = ncell; a.Cin = 256; a.w = wt + h->wpb_off; a.bias = wt + h->bpb_off; a.Cout = 68;
    a.out = h->logits; a.out_ld = 68; a.out_bstride = (long)ncell * 68; a.relu = 0;
    if (launch_conv(a, 1, false, false, B, st)) return -1;
  }
  mark(ST_DB);
  {  // convDb 1x1 on channels [256,512)
    ConvArgs a = {};
    a.in = h->apd; a.in_ld = 512; a.in_coff = 256; a.in_bstride = (long)ncell * 512;
    a.H = 1; a.W = ncell; a.Cin = 256; a.w = wt + h->w_off[11]; a.bias = wt + h->b_off[11]; a.Cout = 256;
    a.out = h->ddb; a.out_ld = 256; a.out_bstride = (long)ncell * 256; a.relu = 0;
    if (launch_conv(a, 1, false, false, B, st)) return -1;
  }
  }
  mark(ST_SOFTMAX);
  if (launch_softmax(h->logits, logit_ld, H8, W8, h->heat, B, st)) return -1;
  mark(ST_NMS);
  if (launch_nms(h->heat, h->mask, h->supp, h->ss, h->scores, Hs, Ws, B, st)) return -1;
  mark(ST_SELECT);
  if (launch_select(h->scores, Hs, Ws, h->cfg.keypoint_threshold, h->cfg.remove_borders, d_mask, h->counts,
                    h->cand_score, h->cand_idx, h->cand_cap, h->cand_n, h->cfg.max_keypoints, h->kp_score,
                    h->kp_idx, h->kp_n, B, st))
    return -1;
  mark(ST_DNORM);
  if (launch_desc_norm(h->ddb, 256, 0, B * ncell, h->desc, st)) return -1;
  mark(ST_SAMPLE);
  if (launch_sample(h->desc, H8, W8, h->kp_score, h->kp_idx, h->kp_n, Ws, d_feat, d_slots, B, st)) return -1;
  mark(ST_DOWNLOAD);
  h->lastH = H; h->lastW = W; h->lastB = B;
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
  URF_HIP(hipMemcpyAsync(h->h_n, h->kp_n, B * sizeof(int), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipMemcpyAsync(h->h_feat, h->d_feat, (size_t)B * kCap * 259 * sizeof(double), hipMemcpyDeviceToHost, h->st));
  if (prof) (void)hipEventRecord(h->ev[ST_COUNT], h->st);
  URF_HIP(hipStreamSynchronize(h->st));
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
  URF_HIP(hipMemcpyAsync(h->h_n, h->kp_n, sizeof(int), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipMemcpyAsync(h->h_feat, h->d_feat, (size_t)kCap * 259 * sizeof(double), hipMemcpyDeviceToHost, h->st));
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_COUNT], h->st);
  URF_HIP(hipStreamSynchronize(h->st));
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
  if (urf::g_profiling) (void)hipEventRecord(h->ev[ST_COUNT], h->st);
  return 0;
}

extern "C" int urf_sp_sync(urf_sp *h) {
  URF_CHECK(h && h->built, "SuperPoint handle is not built");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  sp_collect_times(h);
  return 0;
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
    case 0: src = h->scores; break;
    case 1: src = h->heat; break;
    case 2: src = h->desc; break;
    case 101: src = h->a1; break;
    case 102: src = h->a2a; break;
    case 103: src = h->a2b; break;
    case 104: src = h->a3a; break;
    case 105: src = h->a3b; break;
    case 106: src = h->a4a; break;
    case 107: src = h->a4b; break;
    case 108: src = h->apd; break;     /* [cells][512]: Pa | Da */
    case 109: src = h->logits; break;  /* [cells][68] */
    case 111: src = h->ddb; break;
    default: URF_CHECK(false, "unknown debug tensor %d", which);
  }
  URF_HIP(hipStreamSynchronize(h->st));
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
