// fe_api.hip -- the batched caller of the path.
//
// Tracking::ExtractFeatureThread / ExtractFeatureAndMatch (src/tracking.cc:123-218,
// 338-377 of the reference) take ONE frame, spawn a thread, run SuperPoint, copy
// the features to the host, copy them back for SuperGlue, and sleep between
// polls.  urf_fe is the same contract for a STREAM of frames, submitted in
// batches: features stay in device slots between SuperPoint and SuperGlue, each
// frame is matched against the previous frame or against an earlier frame the
// caller names (the reference matches against its last keyframe,
// src/tracking.cc:196-203), and three HIP streams overlap the stages of
// consecutive batches (SuperPoint | matcher A | matcher B).  Only raw u8 frames
// go up and match lists (optionally features) come down.
//
// The loop shape is the one bench.py times (ur-mvo_amd/pipeline.py), driven by the caller's submits:
//   urf_fe_submit(b):  copy + SuperPoint(b)  ->  match(b - 2)  ->  fetch_begin(b - M - 1)   (the only wait: that batch's fast pass)
// so SuperPoint -- the busiest stream of the strict mode -- always has two batches queued when the host waits, a matcher handle
// gets its next batch as soon as the previous one's fetch has BEGUN (the exact redo of its flagged pairs then runs on the handle's
// redo engine, beside the next batches), and urf_fe_collect hands out the oldest batch, waiting for a redo only when the caller
// insists on a batch whose redo is still running.  A caller that collects right after a submit (integration/tracking.patch) gets
// the synchronous behaviour: collect enqueues whatever of that batch is still missing.
//
// This file is orchestration only: it calls the same C ABI a user would
// (urf_sp_infer_device, urf_match_device_async, urf_pm_fetch_begin / _end, urf_cam_*).
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/urf.h"
#include "urf_common.h"

namespace {
constexpr int kMaxMatchers = 4;
}

struct urf_fe {
  urf_fe_config cfg{};
  bool deferred_error = false;        // a match call enqueued on behalf of a later batch failed: reported by the next collect
  int B = 0, M = 0, NB = 0;           // frames per batch, matchers, ring depth (batches)
  int rows = 0, cols = 0;             // raw frame geometry (fixed at the first submit)
  int frows = 0, fcols = 0;           // geometry fed to SuperPoint (the camera's map size when undistorting)
  urf_sp *sp = nullptr;
  urf_pm *pm[kMaxMatchers] = {nullptr, nullptr, nullptr, nullptr};
  urf_cam *cam = nullptr;             // borrowed
  bool built = false;
  size_t slot_bytes = 0;
  uint8_t *d_slots = nullptr;         // NB x B slots
  uint8_t *d_raw = nullptr;           // NB x B raw frames
  uint8_t *d_und = nullptr;           // NB x B undistorted frames (camera set)
  uint8_t *h_stage = nullptr;         // pinned, NB x B raw frames
  int *h_K = nullptr;                 // pinned, NB x B keypoint counts (copied on SuperPoint's stream right after SP(b))
  std::vector<hipEvent_t> ev_K;       // per ring entry: that copy has landed
  std::vector<hipEvent_t> ev_in;      // per ring entry: the batch's frames are on the device (recorded on the copy stream)
  hipStream_t cst = nullptr;          // non-blocking stream for collect-time copies (a null-stream copy would
                                      // wait for every blocking stream, i.e. serialise the pipeline)
  long next_batch = 0;                // index of the next batch to submit
  long frames_seen = 0;               // global index of the first frame of the next batch
  struct Pending {
    long batch; int n; int first_pair; long first_frame;
    bool matched;                       // its match call has been enqueued
    bool begun;                         // its urf_pm_fetch_begin has been made (the matcher handle is free for its next batch)
    std::vector<const void *> s0, s1;
  };
  std::deque<Pending> pending;
  std::vector<long> batch_first;      // ring: global index of the first frame held by ring entry k
  std::vector<int> batch_n;           // ring: frames held by ring entry k
  std::vector<long> batch_id;         // ring: batch index held by ring entry k
};

static uint8_t *slot_ptr(urf_fe *h, long ring_entry, int j) {
  return h->d_slots + ((size_t)ring_entry * h->B + j) * h->slot_bytes;
}

extern "C" int urf_fe_create(const urf_fe_config *cfg, urf_fe **out) {
  URF_CHECK(cfg && out, "urf_fe_create: null argument");
  URF_CHECK(cfg->batch >= 1 && cfg->batch <= 64, "urf_fe_create: batch must be 1..64");
  urf_fe *h = new urf_fe;
  h->cfg = *cfg;
  h->B = cfg->batch;
  h->M = cfg->matchers <= 0 ? 2 : (cfg->matchers > kMaxMatchers ? kMaxMatchers : cfg->matchers);
  // ring: entry k is refilled by SuperPoint(b) while every batch up to b - M - 2 has had its fetch begun (its fast pass is over)
  // and a match reads slots up to 2 + history_batches submits old (NB >= M + 4 + history_batches); up to min(M + 5, 3 M + 2) batches are in
  // flight (fe_max_in_flight) -- a flagged batch's redo waits one step in the shared engine's pool for the next batch's flagged pairs, then runs
  // for about two (the hand-out lag of DESIGN.md section 12) -- and an entry is not refilled before its batch has been
  // collected: NB = M + 6 + history_batches
  h->NB = h->M + 6 + (cfg->history_batches > 0 ? cfg->history_batches : 0);
  // a strict matcher promises the oracle's lists only on slots an exact SuperPoint made (include/urf.h): a configuration that
  // asks for precision 3 on one side alone would get a 2.2e-4 margin on noisy descriptors -- neither strict nor guarded
  if (cfg->sg.precision == 3 && (cfg->sp.precision == 1 || cfg->sp.precision == 2)) {
    delete h;
    URF_CHECK(false, "urf_fe_create: a strict-parity matcher (sg.precision 3) needs an exact SuperPoint (sp.precision 0 or 3), not %d: "
              "give both entries the same #precision suffix", cfg->sp.precision);
  }
  urf_sp_config sc = cfg->sp;
  sc.max_batch = h->B;
  if (urf_sp_create(&sc, &h->sp)) { delete h; return -1; }
  for (int m = 0; m < h->M; ++m) {
    urf_sg_config gc = cfg->sg;
    gc.max_pairs = h->B;
    gc.device = sc.device;
    if (urf_pm_create(&gc, &h->pm[m])) { urf_fe_destroy(h); return -1; }
  }
  h->batch_first.assign(h->NB, -1);
  h->batch_n.assign(h->NB, 0);
  h->batch_id.assign(h->NB, -1);
  *out = h;
  return 0;
}

extern "C" int urf_fe_build(urf_fe *h, const float *sp_blob, size_t sp_floats, const float *sg_blob, size_t sg_floats) {
  URF_CHECK(h && sp_blob && sg_blob, "urf_fe_build: null argument");
  if (urf_sp_build(h->sp, sp_blob, sp_floats)) return -1;
  for (int m = 0; m < h->M; ++m)
    if (urf_pm_build(h->pm[m], sg_blob, sg_floats)) return -1;
  h->slot_bytes = urf_slot_bytes();
  URF_HIP(hipSetDevice(h->cfg.sp.device));
  URF_HIP(hipMalloc((void **)&h->d_slots, (size_t)h->NB * h->B * h->slot_bytes));
  URF_HIP(hipMemset(h->d_slots, 0, (size_t)h->NB * h->B * h->slot_bytes));
  URF_HIP(hipHostMalloc((void **)&h->h_K, sizeof(int) * h->B * h->NB, hipHostMallocDefault));
  URF_HIP(hipStreamCreateWithFlags(&h->cst, hipStreamNonBlocking));
  h->ev_K.resize(h->NB);
  for (auto &e : h->ev_K) URF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  h->ev_in.resize(h->NB);
  for (auto &e : h->ev_in) URF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  URF_HIP(hipDeviceSynchronize());   // the null-stream memset above is not ordered against the handles' streams
  h->built = true;
  return 0;
}

extern "C" int urf_fe_build_files(urf_fe *h, const char *sp_engine_file, const char *sg_engine_file) {
  URF_CHECK(h && sp_engine_file && sg_engine_file, "urf_fe_build_files: null argument");
  if (urf_sp_build_file(h->sp, sp_engine_file)) return -1;
  for (int m = 0; m < h->M; ++m)
    if (urf_pm_build_file(h->pm[m], sg_engine_file)) return -1;
  h->slot_bytes = urf_slot_bytes();
  URF_HIP(hipSetDevice(h->cfg.sp.device));
  URF_HIP(hipMalloc((void **)&h->d_slots, (size_t)h->NB * h->B * h->slot_bytes));
  URF_HIP(hipMemset(h->d_slots, 0, (size_t)h->NB * h->B * h->slot_bytes));
  URF_HIP(hipHostMalloc((void **)&h->h_K, sizeof(int) * h->B * h->NB, hipHostMallocDefault));
  URF_HIP(hipStreamCreateWithFlags(&h->cst, hipStreamNonBlocking));
  h->ev_K.resize(h->NB);
  for (auto &e : h->ev_K) URF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  h->ev_in.resize(h->NB);
  for (auto &e : h->ev_in) URF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  URF_HIP(hipDeviceSynchronize());   // the null-stream memset above is not ordered against the handles' streams
  h->built = true;
  return 0;
}

extern "C" void urf_fe_destroy(urf_fe *h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.sp.device);
  for (int m = 0; m < kMaxMatchers; ++m)
    if (h->pm[m]) { (void)urf_pm_sync(h->pm[m]); }
  if (h->sp && h->built) (void)urf_sp_sync(h->sp);
  for (int m = 0; m < kMaxMatchers; ++m) urf_pm_destroy(h->pm[m]);
  urf_sp_destroy(h->sp);
  (void)hipFree(h->d_slots); (void)hipFree(h->d_raw); (void)hipFree(h->d_und);
  (void)hipHostFree(h->h_stage); (void)hipHostFree(h->h_K);
  if (h->cst) (void)hipStreamDestroy(h->cst);
  for (auto &e : h->ev_K) (void)hipEventDestroy(e);
  for (auto &e : h->ev_in) (void)hipEventDestroy(e);
  delete h;
}

// Camera::UndistortImage in front of SuperPoint (src/tracking.cc feeds undistorted frames); `cam` is borrowed.
extern "C" int urf_fe_set_camera(urf_fe *h, urf_cam *cam, int map_rows, int map_cols) {
  URF_CHECK(h && h->next_batch == 0, "urf_fe_set_camera: set the camera before the first submit");
  URF_CHECK(!cam || (map_rows > 0 && map_cols > 0), "urf_fe_set_camera: give the map size");
  if (cam) {   // the remap writes height x width bytes per frame: the size comes from the handle, the arguments are checked
    int cw = 0, ch = 0;
    if (urf_cam_size(cam, &cw, &ch)) return -1;
    URF_CHECK(cw == map_cols && ch == map_rows, "urf_fe_set_camera: the camera's maps are %d x %d, not %d x %d", ch, cw, map_rows, map_cols);
    const int aH = h->cfg.sp.max_height > 0 ? h->cfg.sp.max_height : 1500, aW = h->cfg.sp.max_width > 0 ? h->cfg.sp.max_width : 1500;
    URF_CHECK(ch <= aH && cw <= aW, "urf_fe_set_camera: maps %d x %d exceed the SuperPoint arena %d x %d", ch, cw, aH, aW);
  }
  h->cam = cam;
  h->frows = cam ? map_rows : 0;
  h->fcols = cam ? map_cols : 0;
  return 0;
}

// Batches that may be in flight: min(M + 5, 3 M + 2).  A matcher handle holds at most TWO begun batches and one whose match
// call is enqueued (sg_api.hip: kBegun, "a second un-fetched batch is refused"), and the two newest batches have SuperPoint
// only: 3 M + 2 -- which is below M + 5 for ONE matcher (5, not 6: the sixth submit's match(b - 2) would need a third begun batch).
static int fe_max_in_flight(const urf_fe *h) { return h->M + 5 < 3 * h->M + 2 ? h->M + 5 : 3 * h->M + 2; }
static int fe_begun_on_handle(const urf_fe *h, long batch) {
  int n = 0;
  for (const auto &q : h->pending) n += (q.batch % h->M == batch % h->M && q.begun && !q.s0.empty()) ? 1 : 0;
  return n;
}

static int fe_enqueue_match(urf_fe *h, urf_fe::Pending &p) {
  urf_pm *pm = h->pm[p.batch % h->M];
  if (!p.s0.empty()) {
    // match(b) needs SP(b) (and every earlier slot) -- and nothing younger: the event recorded right behind SP(b) on the stream
    // where its slots become final, not "everything enqueued so far" (SuperPoint runs two batches ahead of this call)
    if (urf_pm_wait_event(pm, (void *)h->ev_K[p.batch % h->NB])) return -1;
    if (urf_match_device_async(pm, (int)p.s0.size(), p.s0.data(), p.s1.data(), h->cfg.outlier_rejection)) return -1;
  }
  p.matched = true;
  return 0;
}
static int fe_begin(urf_fe *h, urf_fe::Pending &p) {
  if (!p.s0.empty() && urf_pm_fetch_begin(h->pm[p.batch % h->M], (int)p.s0.size()) < 0) return -1;
  p.begun = true;
  return 0;
}
// everything of the batches up to `upto` (index into pending, oldest first) that is still missing, in order: a batch's match
// call needs its matcher's previous batch begun (M submits older: further up the queue), its begin needs its match call
static int fe_force(urf_fe *h, size_t upto) {
  for (size_t i = 0; i <= upto && i < h->pending.size(); ++i) {
    urf_fe::Pending &q = h->pending[i];
    if (!q.matched) {
      for (size_t j = 0; j < i; ++j)
        if (h->pending[j].batch % h->M == q.batch % h->M && !h->pending[j].begun && fe_begin(h, h->pending[j])) return -1;
      if (fe_enqueue_match(h, q)) return -1;
    }
  }
  return 0;
}
// the throughput step of a submit (newest batch b): match(b - 2), then fetch_begin(b - M - 1)
static int fe_pump(urf_fe *h) {
  const long b = h->pending.back().batch;
  for (size_t i = 0; i < h->pending.size(); ++i)
    if (h->pending[i].batch <= b - 2 && !h->pending[i].matched && fe_force(h, i)) return -1;
  for (size_t i = 0; i < h->pending.size(); ++i) {
    urf_fe::Pending &q = h->pending[i];
    // (opportunistic: only while the handle has room for another begun batch -- fe_force begins it when a match call needs it)
    if (q.batch <= b - h->M - 1 && q.matched && !q.begun && fe_begun_on_handle(h, q.batch) < 2 && fe_begin(h, q)) return -1;
  }
  return 0;
}

// Would the NEXT urf_fe_submit accept global frame `frame` as a reference?  The same test submit applies to a reference
// outside its own batch: the frame sits in a ring entry that is not the one about to be refilled and is at most
// NB - M - 4 = 2 + history_batches SUBMITS old (batches are ragged -- a live queue usually holds one or two frames per
// drain --, so the window cannot be derived from a frame count; integration/tracking.patch asks here).
static const uint8_t *fe_resident_slot(urf_fe *h, long want) {
  const long b = h->next_batch;
  const int k = (int)(b % h->NB);
  for (int e = 0; e < h->NB; ++e)
    if (e != k && h->batch_id[e] >= 0 && h->batch_id[e] >= b - (h->NB - h->M - 4) && want >= h->batch_first[e] &&
        want < h->batch_first[e] + h->batch_n[e])
      return slot_ptr(h, e, (int)(want - h->batch_first[e]));
  return nullptr;
}
extern "C" int urf_fe_frame_resident(urf_fe *h, long frame) {
  URF_CHECK(h && h->built, "urf_fe_frame_resident: handle is not built");
  if (frame < 0 || frame >= h->frames_seen) return 0;
  return fe_resident_slot(h, frame) ? 1 : 0;
}

// Submit n <= batch frames (host u8, row stride `step`; frame stride `frame_stride` bytes).
// ref: NULL, or n global frame indices: frame j is matched against frame ref[j] (-1 = its predecessor).
// A referenced frame must be in this batch or in one of the 2 + history_batches batches before it.
// The very first frame of the stream has no predecessor: it gets 0 matches.
extern "C" int urf_fe_submit(urf_fe *h, const uint8_t *frames, int n, int rows, int cols, size_t step,
                             size_t frame_stride, const long *ref) {
  URF_CHECK(h && h->built, "urf_fe_submit: handle is not built");
  URF_CHECK(frames && n >= 1 && n <= h->B && rows > 0 && cols > 0 && step >= (size_t)cols, "urf_fe_submit: bad argument");
  URF_CHECK((int)h->pending.size() < fe_max_in_flight(h), "urf_fe_submit: %d batches in flight, collect one first", fe_max_in_flight(h));
  URF_HIP(hipSetDevice(h->cfg.sp.device));
  if (!h->d_raw) {
    h->rows = rows; h->cols = cols;
    if (!h->cam) { h->frows = rows; h->fcols = cols; }
    const size_t fr = (size_t)rows * cols;
    URF_HIP(hipMalloc((void **)&h->d_raw, (size_t)h->NB * h->B * fr));
    URF_HIP(hipHostMalloc((void **)&h->h_stage, (size_t)h->NB * h->B * fr, hipHostMallocDefault));
    if (h->cam) URF_HIP(hipMalloc((void **)&h->d_und, (size_t)h->NB * h->B * h->frows * h->fcols));
  }
  URF_CHECK(rows == h->rows && cols == h->cols, "urf_fe_submit: frame size changed (%dx%d -> %dx%d)", h->cols, h->rows, cols, rows);
  const long b = h->next_batch;
  const int k = (int)(b % h->NB);
  // pair list first: a bad reference must fail before anything is enqueued or the ring changes
  std::vector<const void *> s0, s1;
  int first_pair = 0;
  for (int j = 0; j < n; ++j) {
    const long g = h->frames_seen + j;
    const long want = (ref && ref[j] >= 0) ? ref[j] : g - 1;
    if (want < 0) { first_pair = 1; continue; }          // first frame of the stream
    URF_CHECK(j > 0 || first_pair == 0, "urf_fe_submit: internal pair bookkeeping");
    URF_CHECK(want < g, "urf_fe_submit: frame %ld cannot be matched against frame %ld", g, want);
    const uint8_t *src = nullptr;
    if (want >= h->frames_seen) {
      src = slot_ptr(h, k, (int)(want - h->frames_seen));   // an earlier frame of this batch
    } else {
      // window = the last NB - M - 4 batches: an older ring entry may be refilled by SuperPoint while
      // this batch's matcher (up to M + 1 submits behind) still reads it; entry k is being refilled now
      src = fe_resident_slot(h, want);
    }
    URF_CHECK(src, "urf_fe_submit: reference frame %ld has left the ring (history_batches too small)", want);
    s0.push_back(src);
    s1.push_back(slot_ptr(h, k, j));
  }
  const size_t fr = (size_t)rows * cols;
  hipStream_t st = (hipStream_t)urf_sp_stream(h->sp);
  // raw frames -> pinned staging -> device.  The staging entry k was last
  // read by the copy of batch b - NB, which finished before match(b - NB) had its fetch begun.
  uint8_t *stage = h->h_stage + (size_t)k * h->B * fr;
  if (step == (size_t)cols && frame_stride == fr) memcpy(stage, frames, (size_t)n * fr);         // tight frames: one copy
  else
    for (int j = 0; j < n; ++j) {
      if (step == (size_t)cols) { memcpy(stage + (size_t)j * fr, frames + (size_t)j * frame_stride, fr); continue; }
      for (int r = 0; r < rows; ++r) memcpy(stage + (size_t)j * fr + (size_t)r * cols, frames + (size_t)j * frame_stride + (size_t)r * step, cols);
    }
  uint8_t *d_raw = h->d_raw + (size_t)k * h->B * fr;
  // on SuperPoint's own stream.  (Round 5 measured the transfer on a copy stream of its own, with an event for SuperPoint to wait
  // for: 1013 -> 922 frames/s -- a seventh stream on the runtime's four hardware queues costs more than the 60 us of PCIe time it
  // takes off the busiest stream)
  URF_HIP(hipMemcpyAsync(d_raw, stage, (size_t)n * fr, hipMemcpyHostToDevice, st));
  const uint8_t *d_in = d_raw;
  if (h->cam) {
    uint8_t *d_und = h->d_und + (size_t)k * h->B * h->frows * h->fcols;
    if (urf_cam_undistort_device(h->cam, d_raw, n, rows, cols, d_und, st)) return -1;
    d_in = d_und;
  }
  if (urf_sp_infer_device(h->sp, n, d_in, h->frows, h->fcols, slot_ptr(h, k, 0))) return -1;
  // keypoint counts (slot headers) come down behind SP(b) -- on the stream where the slots become final --, long before the
  // batch is collected
  hipStream_t rs = (hipStream_t)urf_sp_result_stream(h->sp);
  URF_HIP(hipMemcpy2DAsync(h->h_K + (size_t)k * h->B, sizeof(int), slot_ptr(h, k, 0), h->slot_bytes, sizeof(int), n,
                           hipMemcpyDeviceToHost, rs));
  URF_HIP(hipEventRecord(h->ev_K[k], rs));
  h->batch_first[k] = h->frames_seen;
  h->batch_n[k] = n;
  h->batch_id[k] = b;
  urf_fe::Pending p{b, n, first_pair, h->frames_seen, false, false, std::move(s0), std::move(s1)};
  h->pending.push_back(std::move(p));
  h->next_batch = b + 1;
  h->frames_seen += n;
  // SuperPoint(b) is enqueued; now the step of the older batches (see the head of this file).  This batch has been accepted:
  // a failure on behalf of an older one is reported by the collect that reaches it
  if (fe_pump(h)) h->deferred_error = true;
  return 0;
}

// Results of the oldest submitted batch (blocks until its match lists are on the host).
//   nframes: frames in that batch; K[j]: keypoints of frame j; nmatch[j] / matches[j*cap ..]: its matches
//   (queryIdx indexes the reference frame's keypoints, trainIdx frame j's, like
//   PointMatching::MatchingPoints(features_ref, features_j)); feat: NULL or nframes matrices
//   of 259 x URF_MAX_KEYPOINTS f64 (column-major, the reference's Eigen storage).
extern "C" int urf_fe_collect(urf_fe *h, int *nframes, int *K, urf_dmatch *matches, int cap, int *nmatch, double *feat) {
  URF_CHECK(h && h->built && nframes && K && matches && nmatch, "urf_fe_collect: bad argument");
  if (h->deferred_error) { h->deferred_error = false; return -1; }   // urf_last_error() still holds the enqueue failure
  URF_CHECK(!h->pending.empty(), "urf_fe_collect: nothing submitted");
  URF_HIP(hipSetDevice(h->cfg.sp.device));
  if (fe_force(h, 0)) return -1;                       // (a caller that collects right after its submit: the synchronous shape)
  urf_fe::Pending &p0 = h->pending.front();
  if (!p0.begun && fe_begin(h, p0)) return -1;
  const urf_fe::Pending p = p0;
  const int k = (int)(p.batch % h->NB);
  urf_pm *pm = h->pm[p.batch % h->M];
  const int P = p.n - p.first_pair;
  for (int j = 0; j < p.n; ++j) nmatch[j] = 0;
  if (P > 0) {
    if (urf_pm_fetch_end(pm, P, matches + (size_t)p.first_pair * cap, cap, nmatch + p.first_pair)) return -1;
  } else {
    if (urf_sp_sync(h->sp)) return -1;
  }
  URF_HIP(hipEventSynchronize(h->ev_K[k]));
  for (int j = 0; j < p.n; ++j) K[j] = h->h_K[(size_t)k * h->B + j];
  if (feat)
    for (int j = 0; j < p.n; ++j) {
      int kk = 0;
      if (urf_slot_to_host(slot_ptr(h, k, j), feat + (size_t)j * URF_FEAT_ROWS * URF_MAX_KEYPOINTS, URF_MAX_KEYPOINTS, &kk)) return -1;
    }
  *nframes = p.n;
  h->pending.pop_front();
  return 0;
}

// 1 when urf_fe_collect would return without waiting for the GPU (the oldest batch's lists are final), 0 when not, < 0 on error
extern "C" int urf_fe_ready(urf_fe *h) {
  URF_CHECK(h && h->built, "urf_fe_ready: handle is not built");
  if (h->pending.empty() || h->deferred_error) return 0;
  const urf_fe::Pending &p = h->pending.front();
  if (!p.matched || !p.begun) return 0;
  if (p.s0.empty()) return 1;
  return urf_pm_fetch_ready(h->pm[p.batch % h->M]);
}

extern "C" int urf_fe_in_flight(urf_fe *h) { return h ? (int)h->pending.size() : 0; }
extern "C" int urf_fe_max_in_flight(urf_fe *h) { return h ? fe_max_in_flight(h) : 0; }
extern "C" urf_sp *urf_fe_superpoint(urf_fe *h) { return h ? h->sp : nullptr; }
extern "C" urf_pm *urf_fe_matcher(urf_fe *h, int i) { return (h && i >= 0 && i < h->M) ? h->pm[i] : nullptr; }
