// sg_api.hip -- host side of SuperGlue + PointMatching behind the C ABI
// (include/urf.h).  Mirrors SuperGlue::build / infer (src/super_glue.cpp:21-241)
// and PointMatching::MatchingPoints (src/point_matching.cc:14-61): persistent
// arena, weight repack (fused QKV), one HIP stream, batched pairs.
#include "../../include/urf.h"
#include "h2.h"

#include <float.h>
#include <math.h>
#include <chrono>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace urf {
int launch_cv_ransac(const int *nmatch, const float *pts0, const float *pts1, double thresh, double confidence, int enable,
                     const void *matches, void *out, int *nout, uint8_t *inliers, uint8_t *scratch, int P, hipStream_t st,
                     double *F_out = nullptr, int *iters_out = nullptr);
extern int g_profiling;
int weights_load(const char *path, int kind, std::vector<float> &out);
int launch_conv(const ConvArgs &a, int taps, bool pool, bool fuse1a, int batch, hipStream_t st);
int launch_gemm128(const ConvArgs &a, int batch, hipStream_t st);
int launch_sg_prep_slots(const float *const *slots, int nimg, int width, int height, int *counts, float *kin,
                         float *kxy, float *x, hipStream_t st);
int launch_attn(const float *qkv, const int *counts, int cross, float *o, int nimg, hipStream_t st);
int launch_score(const float *mdesc, const int *counts, float alpha, float *C, float *Ct, float *u, float *v, int P,
                 hipStream_t st);
int launch_sinkhorn(const int *counts, const float *C, const float *Ct, float *u, float *v, int iters, int P,
                    bool fast, hipStream_t st);
size_t sinkhorn_resident_xin_granules(int maxP);
size_t sinkhorn_resident_xbc_granules(int maxP);
int sinkhorn_resident_enabled();
int sinkhorn_resident_supported(int device);
void sinkhorn_resident_set_debug(unsigned long long *p);
int launch_sinkhorn_resident(const int *counts, const float *C, float *u, float *v, float alpha, int iters, int P,
                             void *xin, void *xbc, size_t xin_bytes, size_t xbc_bytes, unsigned *salt, int *err, int device,
                             hipStream_t st);
int launch_decode(const int *counts, const float *C, const float *Ct, const float *u, const float *v, double thresh,
                  const float *kxy, int *mi0, float *mv0, int *mi1, float *mv1, int *idx0, int *idx1, double *ms0,
                  double *ms1, void *matches, float *pts0, float *pts1, int *nmatch, float *Zout, int *gflags, float gz,
                  float *resid_cols, float *resid, float resid_bound, int *err, int P, hipStream_t st);
int launch_guard_z_calib(const int *counts, const float *zf, const float *zx, float log_floor, int *out, int P, hipStream_t st);
int launch_guard_online(const int *counts, const int *mi0f, const float *mv0f, const int *mi1f, const float *mv1f, const float *C,
                        const float *u, const float *v, float log_floor, int *out, int n, hipStream_t st);
int launch_ransac(const int *nmatch, const float *pts0, const float *pts1, float *ps0, float *ps1, float *pn0,
                  float *pn1, float *T, float *F, float *score, int *ninl, uint32_t seed, int iters, float sigma,
                  double confidence, const int *d_sets, int enable, const void *matches, void *out, int *nout,
                  uint8_t *inliers, float *Fbest, float *best_score, int P, hipStream_t st);

struct RedoPool;
constexpr int NP = kCap, LDC = 1028, SG_LAYERS = 18;
enum { PT_PREP = 0, PT_KENC, PT_GNN, PT_SCORE, PT_SINKHORN, PT_DECODE, PT_RANSAC, PT_COUNT };
// stage_ms[PT_COUNT] additionally reports the attention kernels' share of PT_GNN
}  // namespace urf
using namespace urf;
static const double kQScale = 0.125 * 1.4426950408889634;   // fast mode: log2(e) / sqrt(64), folded into the Q projection
// guarded fast mode, matcher: margin (log domain) within which the log-assignment of the fast pipeline may differ from the exact
// pipeline's on entries that can become a match: the fast matcher's own error (measured maximum 2.0e-4 on both bench streams)
// plus what the fast SuperPoint's descriptor noise induces even in the exact matcher (2.4e-4), with 10 % on top (DESIGN.md
// "Guarded fast mode", tools/gpu_margins.py)
static const float kGuardSgZ = 5e-4f;
static const float kGuardSgDescNoise = 2.4e-4f;   // the descriptor-noise share of it (not measurable inside the matcher)
// strict parity mode (precision 3): the slots come from the exact SuperPoint, so only the matcher's own error counts
// (measured maximum 2.0e-4 on the entries a decision can rest on, both bench streams) with 10 % on top
static const float kGuardSgZStrict = 2.2e-4f;
// above this measured fast-vs-exact difference the error model itself is not trusted: a strict handle then redoes EVERY pair in the
// exact mode (at calibration time and in the online check alike)
static const float kGuardSgZCap = 2.5e-3f;
// the flag-rate policy of a strict handle: decided over windows of kFlagWindow pairs of fast batches; above one half the handle runs
// kExactSpell batches in the exact mode before it looks at the fast matcher again
// head-room of the start-up calibration: margin >= kCalibFactor x the largest fast-vs-exact difference of the first pairs.  Eight
// pairs underestimate a stream's maximum: by 1.44 x on the bench streams, by 2.0 x and 2.3 x on random pairs with weights of two and
// three times the residual gain (round 6's sweeps: the online check saw 4.66e-4 after a calibration of 2.06e-4) -- hence 2.5, not
// round 5's 1.6.  The online check (pm_fold_online) then keeps the margin >= 1.6 x the largest difference ANY redone pair has shown.
static const float kCalibFactor = 2.5f;
static const unsigned kFlagWindow = 32;
static const int kExactSpell = 64;
// integrity bound of a fast Sinkhorn result (every fast mode): the largest |column marginal - 1| of the plan the decode reads
// (argmax_kernel, RESID; an invariant of every correct result: the iteration's last update is the column update).  Clean
// launches stay below 2e-5 on both bench streams (tools/gpu_determinism.py prints the largest value seen);
// a pair above the bound is redone with the streaming kernels before its lists are handed out.
static const float kSinkhornResidBound = 1e-4f;

static int g_backoff_override = 0;
// test hook: handles built after this call stay on the streaming Sinkhorn for `batches` batches after a give-up (0 = the default, 64)
#ifdef URF_EXPERIMENTS
static std::atomic<int> g_redo_fault{0};   // (urf_probe_redo_fault: the next n redo passes fail after they have taken their jobs)
extern "C" int urf_probe_sinkhorn_backoff(int batches) { g_backoff_override = batches; return 0; }
extern "C" int urf_probe_redo_fault(int passes) { g_redo_fault.store(passes < 0 ? 0 : passes); return 0; }
#endif

struct urf_pm {
  urf_sg_config cfg;
  int device = 0, maxP = 1, iters = 100, r_iters = 200;
  float r_sigma = 1.0f;
  double r_conf = 0.99;              // <= 0: every hypothesis counts
  int *ninl = nullptr, *d_sets = nullptr;
  hipStream_t st = nullptr;
  bool own_stream = true;
  hipEvent_t ev_done = nullptr;
  hipEvent_t ev_sink = nullptr;      // recorded right before the Sinkhorn iterations of every batch
  hipStream_t wait_before = nullptr; // optional: stream whose recorded work the next batch must wait for
  hipEvent_t ev_ext = nullptr;
  bool built = false;
  float *d_w = nullptr;
  size_t kw[5], kb[5];
  struct { size_t wqkv, bqkv, wm, bm, w1, b1, w2, b2, b1f, bqkv_f; } L[18];
  size_t wf, bf;
  float bin_score = 1.0f;
  // fast precision mode (split-f16 MFMA): transposed weights as hi/lo f16 planes
  int precision = 0;
  _Float16 *d_wh = nullptr, *d_wl = nullptr;
  struct { size_t qk, v, w1, w2; } H[18];   // w1: first MLP layer with the merge layer folded in
  size_t hwf = 0;
  _Float16 *xh = nullptr, *xl = nullptr, *qkh = nullptr, *qkl = nullptr, *vth = nullptr, *vtl = nullptr, *oh = nullptr,
           *ol = nullptr, *hh = nullptr, *hl = nullptr;
  // activations
  int *counts = nullptr;
  float *kin = nullptr, *kxy = nullptr, *x = nullptr, *tA = nullptr, *tB = nullptr, *qkv = nullptr, *o = nullptr,
        *msg = nullptr, *hid = nullptr, *mdesc = nullptr;
  float *C = nullptr, *Ct = nullptr, *u = nullptr, *v = nullptr, *Z = nullptr;
  // LDS-resident Sinkhorn (fast mode): exchange granules, launch salt, device error word and its pinned mirror
  unsigned long long *rs_xin = nullptr, *rs_xbc = nullptr;
  unsigned rs_salt = 0;
  int *rs_err = nullptr, *h_rs_err = nullptr;   // [0]: 1 = a launch gave up, 2 = a result failed the integrity bound; [2..3]: those pairs
  float *rs_resid_cols = nullptr;                  // per pair and column: |column marginal - 1| (argmax_kernel writes, decode_kernel reduces)
  float *rs_resid = nullptr, *h_resid = nullptr;   // per pair: largest |column marginal - 1| of the plan the decode read (fast modes)
  float resid_bound = 0.0f;                      // <= 0: no integrity check
  unsigned long long batches_seen = 0, last_integrity_batch = 0;   // pm_pipeline calls of this handle; the one of the last integrity event
  unsigned long long rs_integrity_pairs = 0;     // pairs whose Sinkhorn result failed the bound and was redone
  int rs_integrity_events = 0;                   // batches in which that happened
  float last_resid[64];                          // of the batch handed out last
  bool rs_on = false;
  bool rs_wanted = false;                     // the handle uses the resident kernel unless it is backing off after a give-up
  int rs_fallbacks = 0;                       // launches that gave up and were redone with the streaming kernels
  int rs_backoff = 0, rs_backoff_next = 64;   // streaming batches left before the resident kernel is tried again / after the next give-up
  size_t rs_xin_bytes = 0, rs_xbc_bytes = 0;
  int last_P = 0; bool last_Z = false, last_ransac = false;   // what the last pm_pipeline ran (pm_check_resident redoes its tail)
  int *mi0 = nullptr, *mi1 = nullptr, *idx0 = nullptr, *idx1 = nullptr, *nmatch = nullptr, *nfinal = nullptr;
  float *mv0 = nullptr, *mv1 = nullptr;
  double *ms0 = nullptr, *ms1 = nullptr;
  urf_dmatch *matches = nullptr, *fmatches = nullptr;
  float *ps0 = nullptr, *ps1 = nullptr;   // RANSAC: correspondences in canonical (sorted) order
  float *pts0 = nullptr, *pts1 = nullptr, *pn0 = nullptr, *pn1 = nullptr, *T = nullptr, *F = nullptr, *score = nullptr,
        *Fbest = nullptr, *best_score = nullptr;
  uint8_t *inliers = nullptr, *cv_scratch = nullptr;   // (cv_scratch: the current hypothesis' mask of the OpenCV-form outlier stage)
  const float **d_slotptrs = nullptr;
  // pinned host
  urf_dmatch *h_matches = nullptr;
  int *h_n = nullptr;
  const float **h_slotptrs = nullptr;
  hipEvent_t ev[PT_COUNT + 1];
  hipEvent_t ev_attn[18][2];
  float stage_ms[PT_COUNT + 2];      // [PT_COUNT + 1]: the exact redo of the batch's flagged pairs (host time: enqueue + wait)
  bool ev_valid = false;
  int cks_slot = 0;          // experiments build: row of the checksum table (URF_CHECKSUMS)
  bool is_engine = false;    // this handle is the redo engine of a strict-parity handle (its launches run beside saturated streams)
  bool tail_exact = false;   // the exact mode's Sinkhorn behind the fast layers (strict parity: a smaller matcher margin)
  int pending_P = 0;     // pairs of the batch enqueued by urf_match_device_async and not fetched yet (0 = none)
  // guarded modes (precision 2, 3): per-pair guard words (device + pinned mirror), counters
  bool fast = false, guarded = false, strict = false;
  int *g_flags = nullptr, *h_gflags = nullptr;
  unsigned long long pairs_seen = 0;
  float g_z = 0.0f;
  // what happens to a flagged pair (urf_sg_config.redo_flagged_pairs): 0 (precision 2's default) = it is reported
  // (urf_pm_near_tie_flags, counters) -- with slots from the FAST SuperPoint the exact matcher could still differ from the
  // oracle there, because the descriptor noise moves those entries as much as the fast matcher does; 1 (precision 3 always) = it
  // is redone in the exact mode, which on exact slots IS the oracle's result
  int redo_pairs = 0;
  // The redo engine: an exact-mode handle of its own (own arena, this handle's stream).  When the host holds the guard words
  // of a batch (urf_pm_fetch, or the end of a host call) and some pair is flagged, the encoded keypoints of just the
  // flagged pairs are copied over, the exact pipeline runs with launch grids sized for THOSE pairs, and its lists replace
  // the fast ones.  Nothing is enqueued for a batch without a flagged pair.  (Round 4 measured the alternative -- the exact
  // pass enqueued unconditionally behind every fast pass over all 8 pairs, unflagged ones masked to zero counts on the
  // device -- at 4 ms of stream time per batch for ~300 launches that do nothing and 12 ms when one pair of eight was
  // flagged, against 4.5 ms for the pair alone: DESIGN.md section 12.)
  // The engine runs on ITS OWN stream: urf_pm_fetch_begin() starts the redo of a batch and returns, the caller enqueues this
  // handle's next batch, and the two run side by side (the redo is a chain of small dependent launches that uses a few per cent
  // of the chip for milliseconds; serialised in front of the handle's next batch it cost 2.7 ms of step time per flagged pair).
  // For that the result buffers exist three times (fm_set / nf_set on the device, hm_set / hn_set pinned): a handle may hold
  // TWO batches whose fetch has begun (their redos run, in order, on the engine) while a third is being computed; `fmatches` /
  // `nfinal` / `h_matches` / `h_n` alias the set of the batch enqueued last.
  // Round 5: the engine sits in a POOL.  By default the pool is the handle's own (one engine per handle, every batch's redo
  // launched at its fetch_begin: round 4's behaviour, and the fastest).  urf_sg_config.redo_shared_engine: the strict handles
  // of a device that were built from the same weights with the same configuration (a pipeline's two matcher handles, the
  // matchers of a urf_fe) share ONE engine -- 48 MB of weights and an arena less per extra handle; with redo_merge the flagged
  // pairs of CONSECUTIVE batches, which sit on different handles, go through it in ONE pass (a pass over three pairs costs 5.5 ms
  // of kernel time alone, passes over two and over one 4.1 + 3.8): a batch's job waits in the pool until the next fetch_begin
  // of ANY sharing handle and is launched together with that batch's job, if it has one -- or as soon as somebody asks for it
  // (fetch_ready, fetch_end).  Measured in the benched loop: 1118 frames/s with private engines, 1042 shared, 1026 shared and
  // merged -- the two handles' passes must run side by side, on two streams.
  struct urf::RedoPool *pool = nullptr;
  bool redo_merge = false;           // a job waits one step in the pool for a companion (urf_sg_config.redo_merge; shared engines only)
  urf_pm *redo = nullptr;            // = pool->engine (borrowed)
  static constexpr int kSets = 3, kBegun = 2;
  urf_dmatch *fm_set[kSets] = {nullptr, nullptr, nullptr}, *hm_set[kSets] = {nullptr, nullptr, nullptr};
  int *nf_set[kSets] = {nullptr, nullptr, nullptr}, *hn_set[kSets] = {nullptr, nullptr, nullptr};
  int cur_set = 0, fetched_set = 0;
  // batches whose fetch has begun (urf_pm_fetch_begin) and not ended, oldest first
  struct Begun {
    int P = 0, set = 0, n = 0;       // pairs, result set, pairs being redone (0: the lists were final at begin)
    std::atomic<bool> launched{false};   // the entry's redo pass has been enqueued on the engine's stream (else it waits in the pool); written under the pool's mutex by whichever sharing handle flushes, read by the owner
    std::atomic<bool> failed{false};     // its pass could not be enqueued (a launch or copy failed): urf_pm_fetch_end reports an error instead of handing out un-redone lists
    bool want_Z = false, ransac = false;   // of the batch (a pass takes jobs that agree on them)
    int idx[64];                     // slot k of the engine = pair idx[k] of the batch
    float resid[64];                 // the batch's Sinkhorn residuals (urf_pm_sinkhorn_residuals)
    int flags[64];                   // the batch's guard words and stage times: what urf_pm_near_tie_flags / urf_pm_stage_ms
    float stage[PT_COUNT + 2];       //   report once the batch has been handed out (a younger batch may have begun meanwhile)
    hipEvent_t ev_in = nullptr;      // the flagged pairs' inputs are in this entry's staging buffers
    hipEvent_t ev_done = nullptr;    // the redo has delivered
    int *counts = nullptr;           // staging (strict handles): counts / pixel coordinates / encoded keypoints of the flagged pairs,
    float *kxy = nullptr, *x = nullptr;   // copied on THIS handle's stream, so that its next batch never waits for the engine
    // online check of the error model: the fast pass's row / column best entries of the pairs being redone (staged like the inputs),
    // the largest fast-vs-exact difference on them per redone pair (pinned; written behind the pass), and the audit of one
    // UNFLAGGED pair (audit_k = its position in idx[], -1 = none; its fast list kept on the host for the comparison)
    int *mi0 = nullptr, *mi1 = nullptr;
    float *mv0 = nullptr, *mv1 = nullptr;
    float *h_online = nullptr;
    float margin_at_begin = 0.0f;
    int audit_k = -1;
    std::vector<urf_dmatch> audit_list;
    std::chrono::steady_clock::time_point t0;
  } bq[kBegun];
  int bq_head = 0, bq_n = 0;
  unsigned long long pairs_redone = 0, cause_thr = 0, cause_run = 0;
  float redo_ms = 0.0f;            // time from the start of the last redo until its results were waited for, profiling
  float *h_up = nullptr;           // pinned staging of the one-pair host calls: kin | kxy | x (allocated by the first such call)
  int up_n[2] = {0, 0};            //   rows of it the last call filled, per image
  int last_flags[64];              // guard words of the batch handed out by the last fetch / host call
  int fast_flags[64];              // guard words of the batch whose fast pass was enqueued last, recorded by its first begin (a retried begin reads these: the pinned words are cleared once)
  bool flags_recorded = false;
  bool redo_queued = false;        // the redo of the batch whose fast pass was enqueued last has been queued (or it needed none)
  unsigned long long pairs_flagged = 0;
  // online margin (strict handles): folded from every redo's by-product, and from an audit of one unflagged pair per audit_period begun batches
  int *online = nullptr;           // (engine) device words of the pass's pairs
  float online_worst = 0.0f;
  unsigned long long online_pairs = 0, online_violations = 0, margin_raises = 0, audits = 0, audit_mismatches = 0, begins = 0;
  int audit_period = 256;
  // a strict handle whose guard flags most pairs (or whose error model is beyond its cap: redo_all) gains nothing from the fast pass:
  // fast pass + exact redo costs more than the exact pass alone once about half the pairs are redone (bench.py,
  // strict_parity_vs_flag_rate: 519 against 652 frames/s with every pair redone).  The handle then runs its batches in the exact
  // mode itself -- same lists, no fast pass -- and looks at the fast matcher again after kExactSpell batches.
  int exact_left = 0;              // batches still to run in the exact mode before the fast matcher is tried again
  unsigned long long exact_spells = 0;
  bool exact_policy = true;        // (urf_sg_config.redo_flagged_pairs == 2 switches the diversion off: measurements of the redo path itself)
  unsigned win_pairs = 0, win_flagged = 0;   // flag statistics of the fast batches since the last decision
  unsigned long long exact_batches = 0;
  bool last_fast = false;          // the batch enqueued last ran the fast matcher
  bool calibrating = false;        // pm_calibrate_core's own fast pass: never diverted
  float *calib_zf = nullptr;       // calibration scratch: the fast pass's log-assignments of up to maxP pairs, and the reduction word
  int *calib_acc = nullptr;
  bool calib_said = false;         // "the automatic guard calibration could not run" has been said for THIS handle (no process-global state: the reference calls from fresh threads)
  // automatic calibration of the guard's margin (urf_sg_config.calibrate_pairs): pairs still to be measured, the largest difference seen
  int calib_left = 0, calib_failures = 0;
  float calib_worst = 0.0f;
  bool redo_all = false;           // the measured error is above the cap: a strict handle redoes every pair in the exact mode
};

namespace urf {
// The shared redo engine of the strict handles of one device (see urf_pm::pool).
struct RedoPool {
  std::mutex mu;
  int refs = 0;
  int device = 0, maxP = 0;
  unsigned long long blob_hash = 0;
  urf_sg_config key{};               // the engine's configuration (what a redo computes depends on all of it)
  urf_pm *engine = nullptr;
  struct Job { urf_pm *owner; int entry; };
  std::vector<Job> queue;            // staged (inputs copied on the owner's stream, event recorded), not launched yet
  unsigned long long passes = 0, pairs = 0, merged_passes = 0;   // statistics: engine passes, pairs through them, passes that took more than one job
};
static std::mutex g_pools_mu;
static std::vector<RedoPool *> g_pools;
// what a redo computes depends on every one of these (field by field: the structs' padding bytes are nobody's)
static bool same_engine_config(const urf_sg_config &a, const urf_sg_config &b) {
  return a.image_width == b.image_width && a.image_height == b.image_height && a.matching_threshold == b.matching_threshold &&
         a.sinkhorn_iterations == b.sinkhorn_iterations && a.max_pairs == b.max_pairs && a.device == b.device &&
         a.ransac_iterations == b.ransac_iterations && a.ransac_sigma == b.ransac_sigma && a.ransac_seed == b.ransac_seed &&
         a.ransac_threshold_px == b.ransac_threshold_px && a.ransac_confidence == b.ransac_confidence && a.outlier_stage == b.outlier_stage;
}
}  // namespace urf

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
template <typename T>
static int dalloc(T **p, size_t n) {
  URF_HIP(hipMalloc((void **)p, n * sizeof(T)));
  URF_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

extern "C" int urf_pm_create(const urf_sg_config *cfg, urf_pm **out) {
  URF_CHECK(cfg && out, "urf_pm_create: null argument");
  int ndev = 0;
  URF_HIP(hipGetDeviceCount(&ndev));
  URF_CHECK(ndev > 0, "no HIP device: liburf_front needs a gfx950 GPU (there is no CPU fallback)");
  URF_CHECK(cfg->device >= 0 && cfg->device < ndev, "device %d out of range (%d devices)", cfg->device, ndev);
  urf_pm *h = new urf_pm();
  h->cfg = *cfg;
  h->device = cfg->device;
  h->maxP = cfg->max_pairs > 0 ? cfg->max_pairs : 1;
  h->iters = cfg->sinkhorn_iterations > 0 ? cfg->sinkhorn_iterations : 100;
  h->r_iters = cfg->ransac_iterations > 0 ? cfg->ransac_iterations : 200;
  // inlier gate: sigma as given, else from the pixel threshold of the reference call (3 px -> sigma 1.5307,
  // 3.841 sigma^2 = 9 px^2); confidence of the reference call unless switched off
  {
    const float px = cfg->ransac_threshold_px > 0 ? cfg->ransac_threshold_px : 3.0f;
    h->r_sigma = cfg->ransac_sigma > 0 ? cfg->ransac_sigma : (float)((double)px / sqrt(3.841));
    h->r_conf = cfg->ransac_confidence < 0 ? 0.0 : (cfg->ransac_confidence > 0 ? (double)cfg->ransac_confidence : 0.99);
    URF_CHECK(h->r_conf < 1.0, "ransac_confidence must be below 1");
  }
  h->precision = cfg->precision;
  if (!(h->precision >= 0 && h->precision <= 3)) {
    delete h;
    URF_CHECK(false, "precision must be 0 (exact fp32), 1 (fast split-f16), 2 (fast, guarded) or 3 (strict parity)");
  }
  h->fast = h->precision >= 1;
  h->guarded = h->precision >= 2;
  h->strict = h->precision == 3;
  if (h->strict && cfg->redo_flagged_pairs < 0) {
    delete h;
    URF_CHECK(false, "precision 3 (strict parity) without the exact redo of flagged pairs is not strict: redo_flagged_pairs must be 0 or 1");
  }
  if (cfg->guard_margin < 0.0f) { delete h; URF_CHECK(false, "guard_margin must not be negative"); }
  if (cfg->outlier_stage != 0 && cfg->outlier_stage != 1) { delete h; URF_CHECK(false, "outlier_stage must be 0 (in-tree 8-point search) or 1 (OpenCV 4.2's findFundamentalMat restated)"); }
  *out = h;
  return 0;
}

extern "C" int urf_pm_build(urf_pm *h, const float *blob, size_t n_floats) {
  URF_CHECK(h && blob, "urf_pm_build: null argument");
  URF_CHECK(n_floats == URF_SG_BLOB_FLOATS, "SG blob has %zu floats, expected %d", n_floats, URF_SG_BLOB_FLOATS);
  URF_CHECK(!h->built, "urf_pm_build: already built");
  URF_HIP(hipSetDevice(h->device));
  if (const char *e = urf::exp_env("URF_PM_PRIORITY")) URF_HIP(hipStreamCreateWithPriority(&h->st, hipStreamNonBlocking, atoi(e)));
  else URF_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  URF_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
  URF_HIP(hipEventCreateWithFlags(&h->ev_sink, hipEventDisableTiming));
  URF_HIP(hipEventCreateWithFlags(&h->ev_ext, hipEventDisableTiming));
  std::vector<float> host;
  auto put = [&](const float *src, size_t n) {
    size_t off = align_up(host.size(), 64);
    host.resize(off + n, 0.0f);
    if (src) memcpy(host.data() + off, src, n * sizeof(float));
    return off;
  };
  static const int kd[6] = {3, 32, 64, 128, 256, 256};
  const float *p = blob;
  for (int i = 0; i < 5; ++i) {
    if (i == 0) {  // pad cin 3 -> 4 with a zero row (fma(0,0,acc) == acc)
      h->kw[0] = put(nullptr, 4 * 32);
      memcpy(host.data() + h->kw[0], p, 3 * 32 * sizeof(float));
    } else {
      h->kw[i] = put(p, (size_t)kd[i] * kd[i + 1]);
    }
    p += (size_t)kd[i] * kd[i + 1];
    h->kb[i] = put(p, kd[i + 1]);
    p += kd[i + 1];
  }
  for (int l = 0; l < SG_LAYERS; ++l) {
    const float *wq = p, *bq = wq + 65536, *wk = bq + 256, *bk = wk + 65536, *wv = bk + 256, *bv = wv + 65536;
    const float *wm = bv + 256, *bm = wm + 65536, *w1 = bm + 256, *b1 = w1 + 262144, *w2 = b1 + 512,
                *b2 = w2 + 131072;
    p = b2 + 256;
    h->L[l].wqkv = put(nullptr, (size_t)256 * 768);
    float *d = host.data() + h->L[l].wqkv;
    for (int c = 0; c < 256; ++c) {
      memcpy(d + (size_t)c * 768, wq + (size_t)c * 256, 1024);
      memcpy(d + (size_t)c * 768 + 256, wk + (size_t)c * 256, 1024);
      memcpy(d + (size_t)c * 768 + 512, wv + (size_t)c * 256, 1024);
    }
    h->L[l].bqkv = put(nullptr, 768);
    memcpy(host.data() + h->L[l].bqkv, bq, 1024);
    memcpy(host.data() + h->L[l].bqkv + 256, bk, 1024);
    memcpy(host.data() + h->L[l].bqkv + 512, bv, 1024);
    h->L[l].wm = put(wm, 65536); h->L[l].bm = put(bm, 256);
    h->L[l].w1 = put(w1, 262144); h->L[l].b1 = put(b1, 512);
    h->L[l].w2 = put(w2, 131072); h->L[l].b2 = put(b2, 256);
    if (h->fast) {
      // fast mode folds the merge layer into the first MLP layer (both linear, nothing in between):
      //   W0 [x ; Wm o + bm] + b0 = W0x x + (W0m Wm) o + (b0 + W0m bm)
      // ... and log2(e) / sqrt(64) into the Q projection: the attention kernel then gets its scores already in
      // the log2 domain and divided by sqrt(d), and p = 2^(s - max) is a single v_exp_f32
      h->L[l].bqkv_f = put(nullptr, 768);
      {
        float *bqf = host.data() + h->L[l].bqkv_f;
        for (int o = 0; o < 256; ++o) bqf[o] = (float)((double)bq[o] * kQScale);
        memcpy(bqf + 256, bk, 1024);
        memcpy(bqf + 512, bv, 1024);
      }
      h->L[l].b1f = put(nullptr, 512);
      float *bf = host.data() + h->L[l].b1f;
      for (int o = 0; o < 512; ++o) {
        double acc = b1[o];
        for (int j = 0; j < 256; ++j) acc += (double)w1[(size_t)(256 + j) * 512 + o] * (double)bm[j];
        bf[o] = (float)acc;
      }
    }
  }
  h->wf = put(p, 65536); p += 65536;
  h->bf = put(p, 256); p += 256;
  h->bin_score = *p;
  URF_HIP(hipMalloc((void **)&h->d_w, host.size() * sizeof(float)));
  URF_HIP(hipMemcpy(h->d_w, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));

  const size_t P = h->maxP, NI = 2 * P;
  if (h->fast) {
    // W^T [cout][cin] as (hi, lo) f16 planes for h2gemm
    std::vector<_Float16> wh, wl;
    auto putT = [&](const float *w, int cin, int cout) {  // w: [cin][cout] fp32
      const size_t off = wh.size();
      wh.resize(off + (size_t)cin * cout);
      wl.resize(off + (size_t)cin * cout);
      for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
          const float v = w[(size_t)c * cout + o];
          const _Float16 hi = (_Float16)v;
          wh[off + (size_t)o * cin + c] = hi;
          wl[off + (size_t)o * cin + c] = (_Float16)(v - (float)hi);
        }
      return off;
    };
    const float *q = blob + 109376;  // first GNN layer (after the keypoint encoder)
    for (int l = 0; l < SG_LAYERS; ++l) {
      const float *wq = q, *wk = wq + 65536 + 256, *wv = wk + 65536 + 256, *wm = wv + 65536 + 256;
      const float *w1 = wm + 65536 + 256, *w2 = w1 + 262144 + 512;
      q = w2 + 131072 + 256;
      {  // Q rows scaled by log2(e) / 8 (in f64, then split)
        const size_t off = wh.size();
        wh.resize(off + (size_t)256 * 256);
        wl.resize(off + (size_t)256 * 256);
        for (int o = 0; o < 256; ++o)
          for (int c = 0; c < 256; ++c) {
            const double v = (double)wq[(size_t)c * 256 + o] * kQScale;
            const _Float16 hi = (_Float16)v;
            wh[off + (size_t)o * 256 + c] = hi;
            wl[off + (size_t)o * 256 + c] = (_Float16)(v - (double)hi);
          }
        h->H[l].qk = off;
      }
      putT(wk, 256, 256);                 // rows 256..511 of the fused [512][256] matrix
      h->H[l].v = putT(wv, 256, 256);
      {  // first MLP layer with the merge layer folded in: input = [x (256) ; attention output o (256, head-major)]
        std::vector<double> wc((size_t)512 * 512);   // [cin][cout]
        for (int c = 0; c < 256; ++c)
          for (int o = 0; o < 512; ++o) wc[(size_t)c * 512 + o] = w1[(size_t)c * 512 + o];
        for (int c = 0; c < 256; ++c) {              // W'[o][c] = sum_j W0[o][256 + j] Wm[j][c]
          double *dst = wc.data() + (size_t)(256 + c) * 512;
          for (int o = 0; o < 512; ++o) dst[o] = 0.0;
          for (int j = 0; j < 256; ++j) {
            const double m = wm[(size_t)c * 256 + j];
            const float *row = w1 + (size_t)(256 + j) * 512;
            for (int o = 0; o < 512; ++o) dst[o] += (double)row[o] * m;
          }
        }
        const size_t off = wh.size();
        wh.resize(off + (size_t)512 * 512);
        wl.resize(off + (size_t)512 * 512);
        for (int o = 0; o < 512; ++o)
          for (int c = 0; c < 512; ++c) {
            const double v = wc[(size_t)c * 512 + o];
            const _Float16 hi = (_Float16)v;
            wh[off + (size_t)o * 512 + c] = hi;
            wl[off + (size_t)o * 512 + c] = (_Float16)(v - (double)hi);
          }
        h->H[l].w1 = off;
      }
      h->H[l].w2 = putT(w2, 512, 256);
    }
    h->hwf = putT(q, 256, 256);
    URF_HIP(hipMalloc((void **)&h->d_wh, wh.size() * 2));
    URF_HIP(hipMalloc((void **)&h->d_wl, wl.size() * 2));
    URF_HIP(hipMemcpy(h->d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    URF_HIP(hipMemcpy(h->d_wl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
    if (dalloc(&h->xh, NI * NP * 256) || dalloc(&h->xl, NI * NP * 256)) return -1;
    if (dalloc(&h->qkh, NI * NP * 512) || dalloc(&h->qkl, NI * NP * 512)) return -1;
    if (dalloc(&h->vth, NI * NP * 256) || dalloc(&h->vtl, NI * NP * 256)) return -1;
    if (dalloc(&h->oh, NI * NP * 256) || dalloc(&h->ol, NI * NP * 256)) return -1;
    if (dalloc(&h->hh, NI * NP * 512) || dalloc(&h->hl, NI * NP * 512)) return -1;
    h->rs_on = sinkhorn_resident_enabled() != 0;
    if (h->rs_on && !sinkhorn_resident_supported(h->device)) {
      // fewer CUs than the 32 workgroups of a pair (a partitioned or masked device): the streaming kernels, not an error
      fprintf(stderr, "liburf_front: device %d has fewer than 32 CUs, the chip-resident Sinkhorn is off (streaming kernels)\n", h->device);
      h->rs_on = false;
    }
    h->rs_wanted = h->rs_on;
    if (g_backoff_override >= 1) h->rs_backoff_next = g_backoff_override;   // tests (urf_probe_sinkhorn_backoff)
    h->rs_xin_bytes = sinkhorn_resident_xin_granules((int)P) * sizeof(unsigned long long);
    h->rs_xbc_bytes = sinkhorn_resident_xbc_granules((int)P) * sizeof(unsigned long long);
    if (dalloc(&h->rs_xin, sinkhorn_resident_xin_granules((int)P)) || dalloc(&h->rs_xbc, sinkhorn_resident_xbc_granules((int)P))) return -1;
  }
  if (const char *e = urf::exp_env("URF_TAIL_EXACT")) h->tail_exact = atoi(e) != 0;   // experiments build
  if (h->guarded) {
    if (dalloc(&h->g_flags, P)) return -1;
    URF_HIP(hipHostMalloc((void **)&h->h_gflags, P * sizeof(int), hipHostMallocDefault));
    memset(h->h_gflags, 0, P * sizeof(int));
    // margin and redo policy come from the configuration (urf_sg_config.guard_margin / redo_flagged_pairs), never from the
    // environment: they decide what the handle guarantees
    h->g_z = h->cfg.guard_margin > 0.0f ? h->cfg.guard_margin : (h->strict ? kGuardSgZStrict : kGuardSgZ);
    h->redo_pairs = h->cfg.redo_flagged_pairs != 0 ? (h->cfg.redo_flagged_pairs > 0) : (h->strict ? 1 : 0);
    h->exact_policy = h->cfg.redo_flagged_pairs != 2;
    h->redo_merge = h->cfg.redo_merge > 0 && h->cfg.redo_shared_engine != 0;
    if (const char *e = urf::exp_env("URF_REDO_MERGE")) h->redo_merge = atoi(e) != 0;   // (experiments build: A/B)
    if (const char *e = urf::exp_env("URF_REDO_OFF"); e && atoi(e) != 0) h->redo_pairs = 0;   // what-if timing runs (experiments build): results are NOT strict
    URF_CHECK(P <= 64, "guarded fast mode: max_pairs %zu above 64", P);
    memset(h->last_flags, 0, sizeof(h->last_flags));
    // (default: a strict handle measures its first 8 pairs; the guarded mode's margin also depends on the SuperPoint side and is
    // calibrated by the caller, urf_sp_calibrate_guard / urf_pm_calibrate_guard)
    h->calib_left = h->cfg.calibrate_pairs > 0 ? h->cfg.calibrate_pairs : (h->cfg.calibrate_pairs == 0 && h->strict ? 8 : 0);
    if (h->calib_left > 0) {
      URF_HIP(hipMalloc((void **)&h->calib_zf, (size_t)P * (NP + 1) * LDC * sizeof(float)));
      URF_HIP(hipMalloc((void **)&h->calib_acc, sizeof(int)));
    }
  }
  if (dalloc(&h->rs_err, 4)) return -1;
  URF_HIP(hipHostMalloc((void **)&h->h_rs_err, 4 * sizeof(int), hipHostMallocDefault));
  memset(h->h_rs_err, 0, 4 * sizeof(int));
  memset(h->last_resid, 0, sizeof(h->last_resid));
  if (h->fast) {
    URF_CHECK(P <= 64, "fast modes: max_pairs %zu above 64", P);
    if (dalloc(&h->rs_resid, P) || dalloc(&h->rs_resid_cols, P * NP)) return -1;
    URF_HIP(hipHostMalloc((void **)&h->h_resid, P * sizeof(float), hipHostMallocDefault));
    memset(h->h_resid, 0, P * sizeof(float));
    h->resid_bound = h->cfg.sinkhorn_residual_bound != 0.0f ? h->cfg.sinkhorn_residual_bound : kSinkhornResidBound;
  }
  if (dalloc(&h->counts, NI)) return -1;
  if (dalloc(&h->kin, NI * NP * 4)) return -1;
  if (dalloc(&h->kxy, NI * NP * 2)) return -1;
  if (dalloc(&h->x, NI * NP * 256)) return -1;
  if (dalloc(&h->tA, NI * NP * 256)) return -1;
  if (dalloc(&h->tB, NI * NP * 256)) return -1;
  if (dalloc(&h->qkv, NI * NP * 768)) return -1;
  if (dalloc(&h->o, NI * NP * 256)) return -1;
  if (dalloc(&h->msg, NI * NP * 256)) return -1;
  if (dalloc(&h->hid, NI * NP * 512)) return -1;
  if (dalloc(&h->mdesc, NI * NP * 256)) return -1;
  const size_t msz = (size_t)(NP + 1) * LDC;
  if (dalloc(&h->C, P * msz)) return -1;
  if (dalloc(&h->Ct, P * msz)) return -1;
  if (dalloc(&h->Z, P * msz)) return -1;
  if (dalloc(&h->u, P * LDC)) return -1;
  if (dalloc(&h->v, P * LDC)) return -1;
  if (dalloc(&h->mi0, P * NP)) return -1;
  if (dalloc(&h->mi1, P * NP)) return -1;
  if (dalloc(&h->mv0, P * NP)) return -1;
  if (dalloc(&h->mv1, P * NP)) return -1;
  if (dalloc(&h->idx0, P * NP)) return -1;
  if (dalloc(&h->idx1, P * NP)) return -1;
  if (dalloc(&h->ms0, P * NP)) return -1;
  if (dalloc(&h->ms1, P * NP)) return -1;
  if (dalloc(&h->matches, P * NP)) return -1;
  for (int k = 0; k < urf_pm::kSets; ++k)
    if (dalloc(&h->fm_set[k], P * NP) || dalloc(&h->nf_set[k], P)) return -1;
  h->fmatches = h->fm_set[0];
  if (dalloc(&h->nmatch, P)) return -1;
  h->nfinal = h->nf_set[0];
  if (dalloc(&h->pts0, P * NP * 2)) return -1;
  if (dalloc(&h->pts1, P * NP * 2)) return -1;
  if (dalloc(&h->ps0, P * NP * 2)) return -1;
  if (dalloc(&h->ps1, P * NP * 2)) return -1;
  if (dalloc(&h->pn0, P * NP * 2)) return -1;
  if (dalloc(&h->pn1, P * NP * 2)) return -1;
  if (dalloc(&h->T, P * 18)) return -1;
  if (dalloc(&h->F, P * (size_t)h->r_iters * 9)) return -1;
  if (dalloc(&h->score, P * (size_t)h->r_iters)) return -1;
  if (dalloc(&h->ninl, P * (size_t)h->r_iters)) return -1;
  if (dalloc(&h->d_sets, (size_t)h->r_iters * 8)) return -1;
  if (dalloc(&h->Fbest, P * 9)) return -1;
  if (dalloc(&h->best_score, P)) return -1;
  if (dalloc(&h->inliers, P * NP)) return -1;
  if (h->cfg.outlier_stage == 1 && dalloc(&h->cv_scratch, P * NP)) return -1;
  if (dalloc(&h->d_slotptrs, NI)) return -1;
  for (int k = 0; k < urf_pm::kSets; ++k) {
    URF_HIP(hipHostMalloc((void **)&h->hm_set[k], P * NP * sizeof(urf_dmatch), hipHostMallocDefault));
    URF_HIP(hipHostMalloc((void **)&h->hn_set[k], P * sizeof(int), hipHostMallocDefault));
  }
  h->h_matches = h->hm_set[0]; h->h_n = h->hn_set[0];
  for (int k = 0; k < urf_pm::kBegun; ++k) {
    URF_HIP(hipEventCreateWithFlags(&h->bq[k].ev_in, hipEventDisableTiming));
    URF_HIP(hipEventCreateWithFlags(&h->bq[k].ev_done, hipEventDisableTiming));
  }
  URF_HIP(hipHostMalloc((void **)&h->h_slotptrs, NI * sizeof(float *), hipHostMallocDefault));
  for (int i = 0; i <= PT_COUNT; ++i) URF_HIP(hipEventCreate(&h->ev[i]));
  for (int i = 0; i < 18; ++i) { URF_HIP(hipEventCreate(&h->ev_attn[i][0])); URF_HIP(hipEventCreate(&h->ev_attn[i][1])); }
  // the arena was zeroed with hipMemset on the null stream, which the handle's non-blocking stream does not wait
  // for: without this a first call could run before (or while) its buffers are being cleared
  URF_HIP(hipDeviceSynchronize());
  h->built = true;
  if (h->guarded && h->redo_pairs) {
    // the redo engine (see urf_pm::redo): an exact-mode handle with the same configuration and weights, on THIS handle's stream
    urf_sg_config rc = h->cfg;
    rc.precision = 0; rc.redo_flagged_pairs = 0; rc.guard_margin = 0.0f; rc.max_pairs = h->maxP;
    rc.calibrate_pairs = 0; rc.sinkhorn_residual_bound = 0.0f; rc.redo_merge = 0; rc.redo_shared_engine = 0;
    // FNV-1a over the weights: handles built from the same blob with the same configuration share one engine
    unsigned long long hash = 1469598103934665603ull;
    for (size_t i = 0; i < n_floats; ++i) {
      unsigned w;
      memcpy(&w, blob + i, 4);
      hash = (hash ^ w) * 1099511628211ull;
    }
    bool share = h->cfg.redo_shared_engine != 0;           // default: an engine of the handle's own (measured: DESIGN.md section 12)
    if (const char *e = urf::exp_env("URF_REDO_SHARED")) share = atoi(e) != 0;   // (experiments build: A/B)
    urf::RedoPool *pool = nullptr;
    {
      std::lock_guard<std::mutex> lock(urf::g_pools_mu);
      if (share)
        for (urf::RedoPool *q : urf::g_pools)
          if (q->device == h->device && q->maxP == h->maxP && q->blob_hash == hash && urf::same_engine_config(q->key, rc)) { pool = q; break; }
      if (pool) {
        std::lock_guard<std::mutex> l2(pool->mu);
        pool->refs += 1;
      }
    }
    if (!pool) {
      pool = new urf::RedoPool();
      pool->device = h->device; pool->maxP = h->maxP; pool->blob_hash = hash; pool->key = rc; pool->refs = 1;
      if (urf_pm_create(&rc, &pool->engine)) { delete pool; return -1; }
      pool->engine->is_engine = true;
      if (urf_pm_build(pool->engine, blob, n_floats)) { urf_pm_destroy(pool->engine); delete pool; return -1; }
      if (const char *e = urf::exp_env("URF_REDO_PRIORITY")) {   // experiments build: the engine's stream at another priority
        (void)hipStreamDestroy(pool->engine->st);
        URF_HIP(hipStreamCreateWithPriority(&pool->engine->st, hipStreamNonBlocking, atoi(e)));
      }
      if (share) {
        std::lock_guard<std::mutex> lock(urf::g_pools_mu);
        urf::g_pools.push_back(pool);
      }
    }
    h->pool = pool;
    h->redo = pool->engine;
    for (int k = 0; k < urf_pm::kBegun; ++k) {
      if (dalloc(&h->bq[k].counts, NI) || dalloc(&h->bq[k].kxy, NI * NP * 2) || dalloc(&h->bq[k].x, NI * NP * 256)) return -1;
      if (dalloc(&h->bq[k].mi0, P * NP) || dalloc(&h->bq[k].mi1, P * NP) || dalloc(&h->bq[k].mv0, P * NP) || dalloc(&h->bq[k].mv1, P * NP)) return -1;
      URF_HIP(hipHostMalloc((void **)&h->bq[k].h_online, 64 * sizeof(float), hipHostMallocDefault));
      memset(h->bq[k].h_online, 0, 64 * sizeof(float));
    }
    if (!pool->engine->online) URF_HIP(hipMalloc((void **)&pool->engine->online, 64 * sizeof(int)));
    // the engine's stream does its first work NOW: a HIP stream's hardware queue is set up by its first submission, and those
    // milliseconds would otherwise land on the first flagged pair of the handle (measured: 10.4 against 5.5 ms for a per-call match)
    URF_HIP(hipMemsetAsync(pool->engine->online, 0, 64 * sizeof(int), pool->engine->st));
    URF_HIP(hipStreamSynchronize(pool->engine->st));
    h->audit_period = h->cfg.audit_period > 0 ? h->cfg.audit_period : (h->cfg.audit_period < 0 ? 0 : 256);
    URF_HIP(hipDeviceSynchronize());
  }
  return 0;
}

extern "C" int urf_pm_build_file(urf_pm *h, const char *path) {
  std::vector<float> blob;
  if (urf::weights_load(path, 2, blob)) return -1;
  return urf_pm_build(h, blob.data(), blob.size());
}

extern "C" void urf_pm_destroy(urf_pm *h) {
  if (!h) return;
  if (h->built && h->st) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->st); }
  if (h->pool) {
    urf::RedoPool *pool = h->pool;
    bool last = false;
    hipStream_t engine_st = nullptr;
    {
      std::lock_guard<std::mutex> lock(pool->mu);
      for (size_t i = 0; i < pool->queue.size();)      // this handle's jobs that were never launched go with it
        if (pool->queue[i].owner == h) pool->queue.erase(pool->queue.begin() + (long)i); else ++i;
      if (pool->engine && pool->engine->built) engine_st = pool->engine->st;
    }
    // a pass that still writes into this handle's sets: waited for OUTSIDE the registry's and the pool's locks (no job of this
    // handle can be launched any more; the engine lives until its last owner -- which this handle still is one of -- lets go), so
    // that other handles' builds, destroys and flushes do not stall behind a device wait
    if (engine_st) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(engine_st); }
    {
      std::lock_guard<std::mutex> registry(urf::g_pools_mu);   // (the order of urf_pm_build: registry, then pool)
      {
        std::lock_guard<std::mutex> lock(pool->mu);
        pool->refs -= 1;
        last = pool->refs == 0;
      }
      if (last)
        for (size_t i = 0; i < urf::g_pools.size(); ++i)
          if (urf::g_pools[i] == pool) { urf::g_pools.erase(urf::g_pools.begin() + (long)i); break; }
    }
    if (last) {
      urf_pm_destroy(pool->engine);
      delete pool;
    }
    h->pool = nullptr;
    h->redo = nullptr;
  }
  if (h->built) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->st);
    void *bufs[] = {h->d_wh, h->d_wl, h->xh, h->xl, h->qkh, h->qkl, h->vth, h->vtl, h->oh, h->ol, h->hh, h->hl,
                    h->d_w, h->counts, h->kin, h->kxy, h->x, h->tA, h->tB, h->qkv, h->o, h->msg, h->hid, h->mdesc, h->C,
                    h->Ct, h->Z, h->u, h->v, h->mi0, h->mi1, h->mv0, h->mv1, h->idx0, h->idx1, h->ms0, h->ms1,
                    h->matches, h->fm_set[0], h->fm_set[1], h->fm_set[2], h->nmatch, h->nf_set[0], h->nf_set[1], h->nf_set[2], h->pts0, h->pts1, h->ps0, h->ps1, h->pn0, h->pn1, h->T, h->F,
                    h->score, h->Fbest, h->best_score, h->inliers, h->cv_scratch, (void *)h->d_slotptrs, h->rs_xin, h->rs_xbc, h->rs_err, h->ninl, h->d_sets,
                    h->g_flags, h->rs_resid, h->rs_resid_cols};
    for (void *p : bufs) (void)hipFree(p);
    for (int k = 0; k < urf_pm::kBegun; ++k) {
      (void)hipFree(h->bq[k].counts); (void)hipFree(h->bq[k].kxy); (void)hipFree(h->bq[k].x);
      (void)hipFree(h->bq[k].mi0); (void)hipFree(h->bq[k].mi1); (void)hipFree(h->bq[k].mv0); (void)hipFree(h->bq[k].mv1);
      if (h->bq[k].h_online) (void)hipHostFree(h->bq[k].h_online);
    }
    (void)hipFree(h->online);
    (void)hipFree(h->calib_zf); (void)hipFree(h->calib_acc);
    if (h->h_gflags) (void)hipHostFree(h->h_gflags);
    for (int k = 0; k < urf_pm::kSets; ++k) { (void)hipHostFree(h->hm_set[k]); (void)hipHostFree(h->hn_set[k]); }
    for (int k = 0; k < urf_pm::kBegun; ++k) { (void)hipEventDestroy(h->bq[k].ev_in); (void)hipEventDestroy(h->bq[k].ev_done); }
    (void)hipHostFree(h->h_rs_err);
    if (h->h_up) (void)hipHostFree(h->h_up);
    if (h->h_resid) (void)hipHostFree(h->h_resid);
    (void)hipHostFree((void *)h->h_slotptrs);
    for (int i = 0; i <= PT_COUNT; ++i) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 18; ++i) { (void)hipEventDestroy(h->ev_attn[i][0]); (void)hipEventDestroy(h->ev_attn[i][1]); }
    if (h->own_stream) (void)hipStreamDestroy(h->st);
    (void)hipEventDestroy(h->ev_done);
    (void)hipEventDestroy(h->ev_sink);
    (void)hipEventDestroy(h->ev_ext);
  }
  delete h;
}

// Y = act(X W + b) [+res] over all 2P images; rows beyond counts[] are skipped per tile
static int sg_linear(urf_pm *h, int nimg, const float *in, int in_ld, int cin, const float *in2, int in2_ld, int cin1,
                     size_t w, size_t b, int cout, float *out, int out_ld, bool relu, const float *res) {
  ConvArgs a = {};
  a.in = in; a.in_ld = in_ld; a.in_bstride = (long)NP * in_ld;
  a.in2 = in2; a.in2_ld = in2_ld; a.in2_bstride = (long)NP * in2_ld; a.Cin1 = cin1;
  a.H = 1; a.W = NP; a.Cin = cin;
  a.w = h->d_w + w; a.bias = h->d_w + b; a.Cout = cout;
  a.out = out; a.out_ld = out_ld; a.out_bstride = (long)NP * out_ld;
  a.res = res; a.res_ld = out_ld; a.res_bstride = (long)NP * out_ld;
  a.relu = relu ? 1 : 0;
  a.counts = h->counts;
  a.narrow = (nimg <= 2 && !h->is_engine) ? 1 : 0;
  // (one or two pairs -- the per-pair host API, the redo engine of a strict-parity handle: the 128 x 128 tiles of gemm128 would
  // be 64 - 96 workgroups for 256 CUs; the 128 x 64 tiles below are twice as many and half as long.  Same fma chains.)
  static const int small_nimg = [] { const char *e = urf::exp_env("URF_GEMM128_MIN_IMAGES"); return e ? atoi(e) : 5; }();
  if ((cout % 128) == 0 && cout >= 512 && (cin % 64) == 0 && !res && nimg >= small_nimg) return launch_gemm128(a, nimg, h->st);
  return launch_conv(a, 1, false, false, nimg, h->st);
}

// fast mode linear layer: split-f16 GEMM over all 2P images
static int h2_linear(urf_pm *h, int nimg, const _Float16 *xh, const _Float16 *xl, int ldx, int cin, const _Float16 *x2h,
                     const _Float16 *x2l, int ldx2, int cin1, size_t w, size_t bias, int cout, float *out, _Float16 *oh,
                     _Float16 *ol, int ld_out, bool relu, const float *res, bool transposed) {
  H2Args a = {};
  a.xh = xh; a.xl = xl; a.ldx = ldx; a.x_bstride = (long)NP * ldx;
  a.x2h = x2h; a.x2l = x2l; a.ldx2 = ldx2; a.x2_bstride = (long)NP * ldx2; a.Cin1 = cin1;
  a.rows = NP; a.Cin = cin;
  a.wh = h->d_wh + w; a.wl = h->d_wl + w; a.bias = h->d_w + bias; a.Cout = cout;
  a.relu = relu ? 1 : 0; a.res = res; a.counts = h->counts;
  if (transposed) {
    a.ohT = oh; a.olT = ol; a.ldT = NP; a.outT_bstride = (long)cout * NP;
  } else {
    a.out = out; a.oh = oh; a.ol = ol; a.ld_out = ld_out; a.out_bstride = (long)NP * ld_out;
  }
  return launch_h2gemm(a, nimg, h->st);
}

// the 18 GNN layers in the fast precision mode
static int pm_gnn_fast(urf_pm *h, int NI, bool prof) {
  hipStream_t st = h->st;
  if (launch_split(h->x, (size_t)NI * NP * 256, h->xh, h->xl, st)) return -1;
  for (int l = 0; l < SG_LAYERS; ++l) {
    // Q | K (token-major) and V^T ([d][token]) projections: ONE launch over the fused [768][256] weight
    // (x is read once; a second launch costs ~9 us of fixed time)
    {
      H2Args a = {};
      a.xh = h->xh; a.xl = h->xl; a.ldx = 256; a.x_bstride = (long)NP * 256;
      a.rows = NP; a.Cin = 256;
      a.wh = h->d_wh + h->H[l].qk; a.wl = h->d_wl + h->H[l].qk; a.bias = h->d_w + h->L[l].bqkv_f; a.Cout = 768;
      a.counts = h->counts;
      a.oh = h->qkh; a.ol = h->qkl; a.ld_out = 512; a.out_bstride = (long)NP * 512;
      a.ohT = h->vth; a.olT = h->vtl; a.ldT = NP; a.outT_bstride = (long)256 * NP; a.t_from = 512;
      if (launch_h2gemm(a, NI, st)) return -1;
    }
    if (prof) (void)hipEventRecord(h->ev_attn[l][0], st);
    if (launch_attn_h2(h->qkh, h->qkl, h->vth, h->vtl, h->counts, l & 1, h->oh, h->ol, NI, st)) return -1;
    if (prof) (void)hipEventRecord(h->ev_attn[l][1], st);
#ifdef URF_EXPERIMENTS
    // URF_GNN_FUSED=1: the layer's MLP as ONE launch with the hidden activations in LDS (h2mlp.hip; bit-identical
    // results).  Measured and NOT the default: it needs 144 KB of LDS, i.e. one workgroup per CU with two waves per
    // SIMD, and loses against two h2gemm launches at four waves per SIMD -- 1.87 vs 1.77 ms per 8 pairs serialised,
    // 1590 vs 1720 frames/s in the 3-stream pipeline (it also keeps the other streams' kernels off its CUs).
    static int fused = -1;
    if (fused < 0) { const char *e = urf::exp_env("URF_GNN_FUSED"); fused = e ? (atoi(e) != 0) : 0; }
    if (fused) {
      // merge + MLP0 + ReLU + MLP1 + residual in ONE launch, the hidden activations stay in LDS (h2mlp.hip)
      if (launch_h2mlp(h->xh, h->xl, h->oh, h->ol, h->d_wh + h->H[l].w1, h->d_wl + h->H[l].w1, h->d_wh + h->H[l].w2,
                       h->d_wl + h->H[l].w2, h->d_w + h->L[l].b1f, h->d_w + h->L[l].b2, h->counts, NP, NI, st))
        return -1;
      continue;
    }
#endif
    // merge + first MLP layer in one GEMM over [x ; o] (weights folded at build())
    if (h2_linear(h, NI, h->xh, h->xl, 256, 512, h->oh, h->ol, 256, 256, h->H[l].w1, h->L[l].b1f, 512, nullptr, h->hh,
                  h->hl, 512, true, nullptr, false))
      return -1;
    {  // second MLP layer + residual.  The residual stream lives in its two f16 planes only (22 bits, the
       // precision every GEMM of the fast mode sees anyway): no fp32 copy of x is written or read per layer
      H2Args a = {};
      a.xh = h->hh; a.xl = h->hl; a.ldx = 512; a.x_bstride = (long)NP * 512;
      a.rows = NP; a.Cin = 512;
      a.wh = h->d_wh + h->H[l].w2; a.wl = h->d_wl + h->H[l].w2; a.bias = h->d_w + h->L[l].b2; a.Cout = 256;
      a.counts = h->counts;
      a.oh = h->xh; a.ol = h->xl; a.ld_out = 256; a.out_bstride = (long)NP * 256;
      a.resh = h->xh; a.resl = h->xl;
      if (launch_h2gemm(a, NI, st)) return -1;
    }
  }
  return 0;
}

static int pm_tail(urf_pm *h, int P, bool want_Z, bool ransac, bool prof, bool fast);
static int pm_calibrate_core(urf_pm *h, int P, float factor, double *out);
static int pm_auto_calibrated(urf_pm *h, int P, int rc);
static int pm_prep_slots(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1);
#ifdef URF_EXPERIMENTS
static int pm_checksum(urf_pm *h, int slot, int k, const void *x, size_t n_words);
static unsigned long long *g_rsdbg_dev[4] = {nullptr, nullptr, nullptr, nullptr};
constexpr size_t kRsDbgWords = (size_t)8 * 128 * 32 * 2;
#endif

// keypoint encoder: 4(3)->32->64->128->256->256, + descriptors; fp32 in every mode
static int pm_kenc(urf_pm *h, int NI) {
  if (sg_linear(h, NI, h->kin, 4, 4, nullptr, 0, 0, h->kw[0], h->kb[0], 32, h->tA, 256, true, nullptr)) return -1;
  if (sg_linear(h, NI, h->tA, 256, 32, nullptr, 0, 0, h->kw[1], h->kb[1], 64, h->tB, 256, true, nullptr)) return -1;
  if (sg_linear(h, NI, h->tB, 256, 64, nullptr, 0, 0, h->kw[2], h->kb[2], 128, h->tA, 256, true, nullptr)) return -1;
  if (sg_linear(h, NI, h->tA, 256, 128, nullptr, 0, 0, h->kw[3], h->kb[3], 256, h->tB, 256, true, nullptr)) return -1;
  return sg_linear(h, NI, h->tB, 256, 256, nullptr, 0, 0, h->kw[4], h->kb[4], 256, h->x, 256, false, h->x);
}

// the 18 GNN layers and the final projection in the exact mode, in place on h->x (-> h->mdesc)
static int pm_gnn_exact(urf_pm *h, int NI, bool prof) {
  hipStream_t st = h->st;
  // (experiments build, what-if timing of the redo engine: bit 0 = no attention, bit 1 = no linear layers; results are NOT exact)
  static const int skip_env = [] { const char *e = urf::exp_env("URF_REDO_SKIP"); return e ? atoi(e) : 0; }();
  const int skip = h->is_engine ? skip_env : 0;
  for (int l = 0; l < SG_LAYERS; ++l) {
    if (skip & 2) { if (!(skip & 1) && launch_attn(h->qkv, h->counts, l & 1, h->o, NI, st)) return -1; continue; }
    if (sg_linear(h, NI, h->x, 256, 256, nullptr, 0, 0, h->L[l].wqkv, h->L[l].bqkv, 768, h->qkv, 768, false, nullptr))
      return -1;
    if (prof) (void)hipEventRecord(h->ev_attn[l][0], st);
    if (!(skip & 1) && launch_attn(h->qkv, h->counts, l & 1, h->o, NI, st)) return -1;
    if (prof) (void)hipEventRecord(h->ev_attn[l][1], st);
    if (sg_linear(h, NI, h->o, 256, 256, nullptr, 0, 0, h->L[l].wm, h->L[l].bm, 256, h->msg, 256, false, nullptr))
      return -1;
    if (sg_linear(h, NI, h->x, 256, 512, h->msg, 256, 256, h->L[l].w1, h->L[l].b1, 512, h->hid, 512, true, nullptr))
      return -1;
    if (sg_linear(h, NI, h->hid, 512, 512, nullptr, 0, 0, h->L[l].w2, h->L[l].b2, 256, h->x, 256, false, h->x))
      return -1;
  }
  return 0;
}

// the whole matching pipeline for P pairs whose inputs (counts, kin, kxy, x) are in place
static int pm_pipeline(urf_pm *h, int P, bool want_Z, bool ransac) {
  hipStream_t st = h->st;
  const int NI = 2 * P;
  const bool prof = urf::g_profiling != 0;
  auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->ev[i], st); };
  mark(PT_KENC);
  if (pm_kenc(h, NI)) return -1;
  mark(PT_GNN);
  const bool divert = h->fast && h->strict && h->redo_pairs && !h->calibrating && (h->redo_all || h->exact_left > 0);
  const bool fast = h->fast && !divert;
  if (divert) { h->exact_batches += 1; if (h->exact_left > 0) h->exact_left -= 1; }
  h->last_fast = fast;
  if (fast) {
    if (pm_gnn_fast(h, NI, prof)) return -1;
  } else if (pm_gnn_exact(h, NI, prof)) return -1;
  mark(PT_SCORE);
  if (fast) {
    if (h2_linear(h, NI, h->xh, h->xl, 256, 256, nullptr, nullptr, 0, 0, h->hwf, h->bf, 256, h->mdesc, nullptr, nullptr, 256, false,
                  nullptr, false))
      return -1;
  } else if (sg_linear(h, NI, h->x, 256, 256, nullptr, 0, 0, h->wf, h->bf, 256, h->mdesc, 256, false, nullptr)) return -1;
  h->last_P = P; h->last_Z = want_Z; h->last_ransac = ransac;
  h->flags_recorded = false;
  h->redo_queued = false;
  // this batch's lists go to the other result set (the previous batch's may still be waiting for its redo / its fetch_end)
  {   // a set that no begun batch holds (at most two do)
    int s2 = (h->cur_set + 1) % urf_pm::kSets;
    for (int t = 0; t < urf_pm::kSets; ++t, s2 = (s2 + 1) % urf_pm::kSets) {
      bool used = false;
      for (int k = 0; k < h->bq_n; ++k) used = used || h->bq[(h->bq_head + k) % urf_pm::kBegun].set == s2;
      if (!used) break;
    }
    h->cur_set = s2;
  }
  h->fmatches = h->fm_set[h->cur_set]; h->nfinal = h->nf_set[h->cur_set];
  h->h_matches = h->hm_set[h->cur_set]; h->h_n = h->hn_set[h->cur_set];
  h->pairs_seen += (unsigned long long)P;
  h->batches_seen += 1;
  return pm_tail(h, P, want_Z, ransac, prof, fast);
}

// scores -> Sinkhorn -> decode -> outlier stage, from the projected descriptors h->mdesc (which stay in place until the next
// pm_pipeline of this handle: pm_check_resident can run this again).  fast: the fast mode's Sinkhorn and, when the handle is
// guarded, the near-tie guard of the decode.
static int pm_tail(urf_pm *h, int P, bool want_Z, bool ransac, bool prof, bool fast) {
  hipStream_t st = h->st;
  auto mark = [&](int i) { if (prof) (void)hipEventRecord(h->ev[i], st); };
#ifdef URF_EXPERIMENTS
  if (!h->is_engine) {
    static int next_slot = 0;
    if (h->cks_slot == 0) h->cks_slot = 1 + (next_slot++ % 3);
    if (pm_checksum(h, h->cks_slot, 0, h->x, (size_t)2 * P * NP * 256)) return -1;          // the encoded keypoints (fp32, before the layers)
    if (pm_checksum(h, h->cks_slot, 1, h->mdesc, (size_t)2 * P * NP * 256)) return -1;      // the projected descriptors (after the layers)
  }
#endif
  if (launch_score(h->mdesc, h->counts, h->bin_score, h->C, h->Ct, h->u, h->v, P, st)) return -1;
#ifdef URF_EXPERIMENTS
  if (!h->is_engine && pm_checksum(h, h->cks_slot, 2, h->C, (size_t)P * (NP + 1) * LDC)) return -1;   // the couplings
#endif
  mark(PT_SINKHORN);
  (void)hipEventRecord(h->ev_sink, st);
#ifdef URF_EXPERIMENTS
  if (fast && h->rs_on && !h->is_engine) {
    static const int dbg_on = [] { const char *e = urf::exp_env("URF_RS_DEBUG"); return e ? atoi(e) : 0; }();
    if (dbg_on && h->cks_slot > 0) {
      if (!g_rsdbg_dev[h->cks_slot]) URF_HIP(hipMalloc((void **)&g_rsdbg_dev[h->cks_slot], kRsDbgWords * 8));
      URF_HIP(hipMemsetAsync(g_rsdbg_dev[h->cks_slot], 0, kRsDbgWords * 8, st));
      sinkhorn_resident_set_debug(g_rsdbg_dev[h->cks_slot]);
    }
  }
#endif
  if (fast && h->rs_on && !h->tail_exact) {
    // fast mode: one persistent launch, the plan stays in LDS (sinkhorn_resident.hip)
    if (launch_sinkhorn_resident(h->counts, h->C, h->u, h->v, h->bin_score, h->iters, P, h->rs_xin, h->rs_xbc, h->rs_xin_bytes,
                                 h->rs_xbc_bytes, &h->rs_salt, h->rs_err, h->device, st))
      return -1;
  } else {
    static const int skip_env = [] { const char *e = urf::exp_env("URF_REDO_SKIP"); return e ? atoi(e) : 0; }();   // bit 2: one Sinkhorn iteration
    if (launch_sinkhorn(h->counts, h->C, h->Ct, h->u, h->v, (h->is_engine && (skip_env & 4)) ? 1 : h->iters, P, fast && !h->tail_exact, st)) return -1;
  }
#ifdef URF_EXPERIMENTS
  if (!h->is_engine) {
    if (pm_checksum(h, h->cks_slot, 3, h->u, (size_t)P * LDC)) return -1;
    if (pm_checksum(h, h->cks_slot, 4, h->v, (size_t)P * LDC)) return -1;
  }
#endif
  mark(PT_DECODE);
  const bool guard = fast && h->guarded;
  if (guard) URF_HIP(hipMemsetAsync(h->g_flags, 0, P * sizeof(int), st));
  // integrity of the fast Sinkhorn's result: the decode's column pass sums every column of the plan it reads (the column marginals
  // are 1 in every correct result), a pair above the bound raises the handle's error word like a give-up does (pm_check_resident)
  static const int no_resid = [] { const char *e = urf::exp_env("URF_NO_RESID"); return e ? atoi(e) : 0; }();   // (experiments: what the integrity word costs)
  const bool integ = fast && !h->tail_exact && h->rs_resid != nullptr && !no_resid;
  if (launch_decode(h->counts, h->C, h->Ct, h->u, h->v, h->cfg.matching_threshold, h->kxy, h->mi0, h->mv0, h->mi1,
                    h->mv1, h->idx0, h->idx1, h->ms0, h->ms1, h->matches, h->pts0, h->pts1, h->nmatch,
                    want_Z ? h->Z : nullptr, guard ? h->g_flags : nullptr, h->g_z, integ ? h->rs_resid_cols : nullptr, h->rs_resid,
                    (integ && h->resid_bound > 0.0f) ? h->resid_bound : FLT_MAX, h->rs_err, P, st))
    return -1;
  if (guard) URF_HIP(hipMemcpyAsync(h->h_gflags, h->g_flags, P * sizeof(int), hipMemcpyDeviceToHost, st));
  if (fast) {
    URF_HIP(hipMemcpyAsync(h->h_rs_err, h->rs_err, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (integ) URF_HIP(hipMemcpyAsync(h->h_resid, h->rs_resid, P * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  mark(PT_RANSAC);
  if (h->cfg.outlier_stage == 1) {
    // the reference's own call, cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask), restated (cvransac.hip)
    const double px = h->cfg.ransac_threshold_px > 0 ? (double)h->cfg.ransac_threshold_px : 3.0;
    if (launch_cv_ransac(h->nmatch, h->pts0, h->pts1, px, h->r_conf > 0.0 ? h->r_conf : 0.99, ransac ? 1 : 0, h->matches, h->fmatches,
                         h->nfinal, h->inliers, h->cv_scratch, P, st))
      return -1;
  } else if (launch_ransac(h->nmatch, h->pts0, h->pts1, h->ps0, h->ps1, h->pn0, h->pn1, h->T, h->F, h->score, h->ninl,
                    h->cfg.ransac_seed, h->r_iters, h->r_sigma, h->r_conf, nullptr, ransac ? 1 : 0, h->matches,
                    h->fmatches, h->nfinal, h->inliers, h->Fbest, h->best_score, P, st))
    return -1;
  mark(PT_COUNT);
  return 0;
}

// The redo of a batch's flagged pairs, once the host holds the guard words (h->h_gflags: the batch's fast pass has been waited
// for).  The flagged pairs' encoded keypoints (h->x after pm_kenc: fp32 in every mode, and untouched by the fast layers, which
// work on their own f16 planes), counts and pixel coordinates go to slots 0 .. n - 1 of the redo engine; its exact pipeline
// runs over those n pairs on its own stream; its lists replace the fast ones in the batch's result set (device and pinned
// mirror) -- and, for the one-pair host calls, its index vectors, scores and log-assignment those of this handle.
// pm_begin_batch pushes the batch onto the begun queue and returns 1 when a redo was started for it, 0 when its lists are final.
// Launch what waits in the pool (its mutex is held): passes over as many jobs as the engine has room for and that agree on the
// batch's options; slot k0 .. k0 + n - 1 of the engine = the job's flagged pairs.
static int pool_flush(urf::RedoPool *pool) {
  urf_pm *r = pool->engine;
  URF_CHECK(r && r->built, "the redo engine is not built");
  hipStream_t st = r->st;
  while (!pool->queue.empty()) {
    std::vector<urf::RedoPool::Job> pass;
    int N = 0;
    const urf_pm::Begun &first = pool->queue.front().owner->bq[pool->queue.front().entry];
    for (size_t i = 0; i < pool->queue.size();) {
      const urf::RedoPool::Job j = pool->queue[i];
      const urf_pm::Begun &e = j.owner->bq[j.entry];
      if (e.want_Z == first.want_Z && e.ransac == first.ransac && N + e.n <= pool->maxP && (!first.want_Z || pass.empty())) {
        pass.push_back(j);
        N += e.n;
        pool->queue.erase(pool->queue.begin() + (long)i);
      } else {
        ++i;
      }
    }
    // a pass that fails half-way has taken its jobs out of the queue: mark every job of it that was not launched, so that its
    // owner's urf_pm_fetch_end reports the failure instead of waiting on an event that was never recorded and handing out the
    // un-redone fast lists (the jobs of OTHER owners in a shared pool included)
    struct MarkFailed {
      std::vector<urf::RedoPool::Job> &pass; bool armed;
      ~MarkFailed() { if (armed) for (const auto &j : pass) { urf_pm::Begun &e = j.owner->bq[j.entry]; if (!e.launched) e.failed = true; } }
    } mark{pass, true};
#ifdef URF_EXPERIMENTS
    if (g_redo_fault.load() > 0) {   // fault injection: the pass fails with its jobs already out of the queue (what a failing launch or copy does)
      g_redo_fault.fetch_sub(1);
      URF_CHECK(false, "injected fault: the redo pass of %zu job(s) was not enqueued", pass.size());
    }
#endif
    int k0 = 0;
    for (const auto &j : pass) {
      urf_pm::Begun &e = j.owner->bq[j.entry];
      URF_HIP(hipStreamWaitEvent(st, e.ev_in, 0));
      URF_HIP(hipMemcpyAsync(r->counts + 2 * k0, e.counts, (size_t)2 * e.n * sizeof(int), hipMemcpyDeviceToDevice, st));
      URF_HIP(hipMemcpyAsync(r->kxy + (size_t)2 * k0 * NP * 2, e.kxy, (size_t)2 * e.n * NP * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
      URF_HIP(hipMemcpyAsync(r->x + (size_t)2 * k0 * NP * 256, e.x, (size_t)2 * e.n * NP * 256 * sizeof(float), hipMemcpyDeviceToDevice, st));
      k0 += e.n;
    }
    // the exact layers, the final projection and the tail over N pairs (grids sized for N, not for a batch)
    if (pm_gnn_exact(r, 2 * N, false)) return -1;
    if (sg_linear(r, 2 * N, r->x, 256, 256, nullptr, 0, 0, r->wf, r->bf, 256, r->mdesc, 256, false, nullptr)) return -1;
    r->last_P = N; r->last_Z = first.want_Z; r->last_ransac = first.ransac;
    if (pm_tail(r, N, first.want_Z, first.ransac, false, false)) return -1;
    k0 = 0;
    for (const auto &j : pass) {
      urf_pm *h = j.owner;
      urf_pm::Begun &e = h->bq[j.entry];
      const int set = e.set;
      // the by-product of the pass: fast-vs-exact on the entries the fast decisions of these pairs rested on (pm_end_batch folds it)
      if (launch_guard_online(r->counts + 2 * k0, e.mi0, e.mv0, e.mi1, e.mv1, r->C + (size_t)k0 * (NP + 1) * LDC, r->u + (size_t)k0 * LDC,
                              r->v + (size_t)k0 * LDC, logf(0.1f), r->online + k0, e.n, st))
        return -1;
      URF_HIP(hipMemcpyAsync(e.h_online, r->online + k0, (size_t)e.n * sizeof(float), hipMemcpyDeviceToHost, st));
      for (int k = 0; k < e.n; ++k) {
        const int p = e.idx[k], q = k0 + k;
        URF_HIP(hipMemcpyAsync(h->nf_set[set] + p, r->nfinal + q, sizeof(int), hipMemcpyDeviceToDevice, st));
        URF_HIP(hipMemcpyAsync(h->fm_set[set] + (size_t)p * NP, r->fmatches + (size_t)q * NP, (size_t)NP * sizeof(urf_dmatch), hipMemcpyDeviceToDevice, st));
        URF_HIP(hipMemcpyAsync(h->hn_set[set] + p, r->nfinal + q, sizeof(int), hipMemcpyDeviceToHost, st));
        URF_HIP(hipMemcpyAsync(h->hm_set[set] + (size_t)p * NP, r->fmatches + (size_t)q * NP, (size_t)NP * sizeof(urf_dmatch), hipMemcpyDeviceToHost, st));
      }
      if (e.P == 1) {   // the one-pair host calls read these as well (urf_sg_infer: index vectors, scores, the log-assignment)
        URF_HIP(hipMemcpyAsync(h->idx0, r->idx0 + (size_t)k0 * NP, NP * sizeof(int), hipMemcpyDeviceToDevice, st));
        URF_HIP(hipMemcpyAsync(h->idx1, r->idx1 + (size_t)k0 * NP, NP * sizeof(int), hipMemcpyDeviceToDevice, st));
        URF_HIP(hipMemcpyAsync(h->ms0, r->ms0 + (size_t)k0 * NP, NP * sizeof(double), hipMemcpyDeviceToDevice, st));
        URF_HIP(hipMemcpyAsync(h->ms1, r->ms1 + (size_t)k0 * NP, NP * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (e.want_Z) URF_HIP(hipMemcpyAsync(h->Z, r->Z + (size_t)k0 * (NP + 1) * LDC, (size_t)(NP + 1) * LDC * sizeof(float), hipMemcpyDeviceToDevice, st));
      }
      URF_HIP(hipEventRecord(e.ev_done, st));
      e.launched = true;
      k0 += e.n;
    }
    mark.armed = false;
    pool->passes += 1;
    pool->pairs += (unsigned long long)N;
    pool->merged_passes += pass.size() > 1;
  }
  return 0;
}
// the entry's redo must be on the engine's stream before anybody waits for (or asks about) its event
static int pm_ensure_launched(urf_pm *h, urf_pm::Begun &e) {
  if (e.n == 0) return 0;
  if (!e.launched && !e.failed) {
    std::lock_guard<std::mutex> lock(h->pool->mu);
    if (!e.launched && !e.failed && pool_flush(h->pool)) return -1;
  }
  URF_CHECK(!e.failed, "the exact redo of this batch's flagged pairs could not be enqueued (an earlier launch or copy on the redo "
            "engine failed): its lists are NOT the strict mode's and are withheld");
  URF_CHECK(e.launched, "the exact redo of this batch's flagged pairs is neither queued nor launched");
  return 0;
}

static int pm_begin_batch(urf_pm *h) {
  URF_CHECK(h->bq_n < urf_pm::kBegun, "two batches of this handle are waiting for their urf_pm_fetch_end already");
  const int entry = (h->bq_head + h->bq_n) % urf_pm::kBegun;
  urf_pm::Begun &e = h->bq[entry];
  const int P = h->last_P;
  e.P = P; e.set = h->cur_set; e.n = 0; e.launched = false; e.failed = false; e.want_Z = h->last_Z; e.ransac = h->last_ransac;
  e.t0 = std::chrono::steady_clock::now();
  h->bq_n += 1;       // (rolled back below when a launch or copy of the redo fails: the entry must not stay queued half-made)
  struct Rollback { urf_pm *h; bool armed; ~Rollback() { if (armed) h->bq_n -= 1; } } rollback{h, true};
  memcpy(e.stage, h->stage_ms, sizeof(e.stage));
  memset(e.flags, 0, sizeof(e.flags));
  memcpy(e.resid, h->last_resid, sizeof(e.resid));
  if (!h->guarded || P < 1) { rollback.armed = false; return 0; }
  if (!h->flags_recorded) {             // (a second call -- after a redo, or a retry after a failed begin -- finds the pinned words cleared: keep the recorded ones)
    memset(h->fast_flags, 0, sizeof(h->fast_flags));
    unsigned nf = 0;
    for (int p = 0; p < P && p < 64; ++p) { h->fast_flags[p] = h->h_gflags[p]; nf += h->h_gflags[p] != 0; }
    h->pairs_flagged += nf;
    for (int p = 0; p < P; ++p) h->h_gflags[p] = 0;
    h->flags_recorded = true;
    if (h->strict && h->redo_pairs && h->last_fast && h->exact_policy) {
      h->win_pairs += (unsigned)P; h->win_flagged += nf;
      if (h->win_pairs >= kFlagWindow) {
        if (2 * h->win_flagged > h->win_pairs) {
          if (h->exact_spells == 0 || (h->exact_spells & (h->exact_spells - 1)) == 0)
            fprintf(stderr, "liburf_front: the guard flagged %u of the last %u pairs: fast pass + exact redo costs more than the exact pass alone -- "
                    "this strict handle runs its next %d batches in the exact mode (same lists), then tries the fast matcher again (spell %llu)\n",
                    h->win_flagged, h->win_pairs, kExactSpell, h->exact_spells + 1);
          h->exact_left = kExactSpell;
          h->exact_spells += 1;
        }
        h->win_pairs = 0; h->win_flagged = 0;
      }
    }
  }
  memcpy(h->last_flags, h->fast_flags, sizeof(h->last_flags));
  h->redo_ms = 0.0f;
  memcpy(e.flags, h->fast_flags, sizeof(e.flags));
  memcpy(e.stage, h->stage_ms, sizeof(e.stage));
  int n = 0;
  e.audit_k = -1;
  e.margin_at_begin = h->g_z;
  h->begins += h->redo_queued ? 0 : 1;
  // audit: one UNFLAGGED pair of every audit_period-th begun batch goes through the exact engine as well -- the only sample of the
  // error model on pairs the guard passed (its exact list replaces the fast one; the two index lists must be equal)
  const int audit_p = (h->strict && h->redo_pairs && h->pool && h->audit_period > 0 && !h->redo_all && h->begins % (unsigned long long)h->audit_period == 0)
                          ? (int)((h->begins / (unsigned long long)h->audit_period) % (unsigned long long)P) : -1;
  // (a second begin of the SAME fast batch after its redo has been queued -- the host calls look again once the redo has rewritten
  // their results -- finds nothing left to do; a retry after a FAILED begin finds the recorded words and queues the redo again)
  // ... and a batch the handle ran in the exact mode itself (diverted: pm_pipeline) has nothing to redo or audit -- its lists ARE the
  // exact ones, and its encoded keypoints (h->x, what a redo starts from) were consumed in place by the exact layers
  const bool todo = !h->redo_queued && h->last_fast;
  for (int p = 0; p < P && p < 64 && todo; ++p) {
    const bool audit = p == audit_p && !h->fast_flags[p];
    if (h->fast_flags[p] || h->redo_all || audit) {
      if (audit) {
        e.audit_k = n;
        const int cnt = h->hn_set[h->cur_set][p];
        e.audit_list.assign(h->hm_set[h->cur_set] + (size_t)p * NP, h->hm_set[h->cur_set] + (size_t)p * NP + (cnt > 0 && cnt <= NP ? cnt : 0));
      }
      e.idx[n++] = p;
    }
  }
  urf::RedoPool *pool = h->pool;
  if (!h->redo_pairs || !pool) { rollback.armed = false; return 0; }
  if (n == 0) {
    // nothing of this batch to redo: what an earlier batch (of any sharing handle) left in the pool has waited its one step
    std::lock_guard<std::mutex> lock(pool->mu);
    if (!pool->queue.empty() && pool_flush(pool)) return -1;
    rollback.armed = false;
    h->redo_queued = true;
    return 0;
  }
  // the inputs leave this handle's buffers on its OWN stream (idle: the host has waited for the batch's fast pass), into the
  // entry's staging: the handle's next batch, in order behind these copies, never waits for the engine -- where an earlier
  // pass may still be running
  for (int k = 0; k < n; ++k) {
    const int p = e.idx[k];
    h->audits += k == e.audit_k;
    h->cause_thr += (e.flags[p] & 1) != 0;
    h->cause_run += (e.flags[p] & 2) != 0;
    URF_HIP(hipMemcpyAsync(e.counts + 2 * k, h->counts + 2 * p, 2 * sizeof(int), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.kxy + (size_t)2 * k * NP * 2, h->kxy + (size_t)2 * p * NP * 2, (size_t)2 * NP * 2 * sizeof(float), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.x + (size_t)2 * k * NP * 256, h->x + (size_t)2 * p * NP * 256, (size_t)2 * NP * 256 * sizeof(float), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.mi0 + (size_t)k * NP, h->mi0 + (size_t)p * NP, NP * sizeof(int), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.mi1 + (size_t)k * NP, h->mi1 + (size_t)p * NP, NP * sizeof(int), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.mv0 + (size_t)k * NP, h->mv0 + (size_t)p * NP, NP * sizeof(float), hipMemcpyDeviceToDevice, h->st));
    URF_HIP(hipMemcpyAsync(e.mv1 + (size_t)k * NP, h->mv1 + (size_t)p * NP, NP * sizeof(float), hipMemcpyDeviceToDevice, h->st));
  }
  URF_HIP(hipEventRecord(e.ev_in, h->st));
  e.n = n;
  {
    std::lock_guard<std::mutex> lock(pool->mu);
    const bool older = !pool->queue.empty();       // a job of the previous step is waiting: this one joins it, both go now
    pool->queue.push_back({h, entry});
    if ((older || !h->redo_merge) && pool_flush(pool)) {
      for (size_t i = 0; i < pool->queue.size();)          // (this entry is rolled back: it must not stay queued)
        if (pool->queue[i].owner == h && pool->queue[i].entry == entry) pool->queue.erase(pool->queue.begin() + (long)i); else ++i;
      e.n = 0;
      return -1;
    }
  }
  rollback.armed = false;
  h->redo_queued = true;
  return 1;
}
// Online margin.  A finished redo pass has left, per redone pair, the largest |Z_fast - Z_exact| over the entries the fast pass's
// decisions rested on (guard_online_kernel) -- a free sample of the error model the guard's margin stands for, on exactly the
// deployment's pairs, for the life of the handle (the calibration at the handle's start is eight pairs).  Folded here:
//   * d > margin / 1.6 : the head-room of the calibration is used up -- the margin is raised to 1.6 d (and said on stderr); above
//     the cap the model is not trusted any more and every pair is redone, as at calibration time;
//   * d > the margin the batch was GUARDED with : the model was violated on this pair (itself redone, hence correct); counted and
//     said loudly -- an unflagged pair of that batch may have carried the same error.
// The audit pair (one unflagged pair per audit_period begun batches) is the only sample on pairs the guard PASSED: its exact index
// list must equal the fast one it replaced.
static void pm_fold_online(urf_pm *h, urf_pm::Begun &e) {
  if (!h->strict || !e.h_online) return;
  float d = 0.0f;
  for (int k = 0; k < e.n && k < 64; ++k) { const float v = e.h_online[k]; if (!(v <= d)) d = v; }   // (a NaN counts as the worst)
  h->online_pairs += (unsigned long long)e.n;
  if (!(d <= h->online_worst)) h->online_worst = d;
  if (!(d <= e.margin_at_begin)) {
    h->online_violations += 1;
    fprintf(stderr, "liburf_front: ONLINE GUARD CHECK: a redone pair's fast-vs-exact difference %.3g on a decisive entry exceeds the margin %.3g its "
            "batch was guarded with: the error model did not hold there (the pair itself was redone; violation %llu of this handle)\n",
            (double)d, (double)e.margin_at_begin, h->online_violations);
  }
  if (!(d <= h->g_z / 1.6f)) {
    const float need = 1.6f * d;
    if (!(need <= h->g_z)) {
      h->margin_raises += 1;
      const bool cap = !(d <= kGuardSgZCap) && h->redo_pairs && !h->redo_all;
      if (h->margin_raises <= 8 || (h->margin_raises & (h->margin_raises - 1)) == 0 || cap)
        fprintf(stderr, "liburf_front: online guard check: fast-vs-exact difference %.3g on a redone pair is above margin / 1.6: margin %.3g -> %.3g%s\n",
                (double)d, (double)h->g_z, (double)need, cap ? " -- above the cap the error model is trusted to: every pair will be redone in the exact mode" : "");
      if (need == need) h->g_z = need;
      if (cap) h->redo_all = true;
    }
  }
  if (e.audit_k >= 0 && e.audit_k < e.n) {
    const int p = e.idx[e.audit_k];
    const int cnt = h->hn_set[e.set][p];
    const urf_dmatch *got = h->hm_set[e.set] + (size_t)p * NP;
    bool same = cnt == (int)e.audit_list.size();
    for (int i = 0; same && i < cnt; ++i) same = got[i].queryIdx == e.audit_list[i].queryIdx && got[i].trainIdx == e.audit_list[i].trainIdx;
    if (!same) {
      h->audit_mismatches += 1;
      fprintf(stderr, "liburf_front: AUDIT: an UNFLAGGED pair's exact index list (%d matches) differs from the fast list the guard had passed (%zu): "
              "the strict guarantee did not hold for it (mismatch %llu of %llu audits; fast-vs-exact %.3g, margin %.3g)\n",
              cnt, e.audit_list.size(), h->audit_mismatches, h->audits, (double)e.h_online[e.audit_k], (double)e.margin_at_begin);
    }
    e.audit_k = -1;
  }
}

// the oldest begun batch: wait for its redo (if one was started) and pop it; *set = its result set, *P its pair count
static int pm_end_batch(urf_pm *h, int *set, int *P) {
  URF_CHECK(h->bq_n > 0, "no fetch has begun");
  urf_pm::Begun &e = h->bq[h->bq_head];
  h->redo_ms = 0.0f;
  if (e.n > 0) {
    if (pm_ensure_launched(h, e)) {
      if (e.failed) { h->bq_head = (h->bq_head + 1) % urf_pm::kBegun; h->bq_n -= 1; }   // the batch is lost; the handle goes on
      return -1;
    }
    URF_HIP(hipEventSynchronize(e.ev_done));
    h->pairs_redone += (unsigned long long)(e.n - (e.audit_k >= 0 ? 1 : 0));
    pm_fold_online(h, e);
    h->redo_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - e.t0).count();
  }
  memcpy(h->last_flags, e.flags, sizeof(e.flags));
  memcpy(h->last_resid, e.resid, sizeof(e.resid));
  memcpy(h->stage_ms, e.stage, sizeof(e.stage));
  if (set) *set = e.set;
  if (P) *P = e.P;
  h->bq_head = (h->bq_head + 1) % urf_pm::kBegun;
  h->bq_n -= 1;
  return e.n > 0 ? 1 : 0;
}
// the synchronous form (host calls): > 0 when results were rewritten (the caller repeats its copies)
static int pm_guard_redo(urf_pm *h) {
  URF_CHECK(h->bq_n == 0, "a device batch of this handle is waiting for its urf_pm_fetch_end");
  const int rc = pm_begin_batch(h);
  if (rc < 0) { h->bq_n = 0; return rc; }
  return pm_end_batch(h, nullptr, nullptr) < 0 ? -1 : rc;
}

// After the stream (or the batch's event) has been waited for: did the resident Sinkhorn give up?  (Its workgroups spin on
// each other, so a launch whose 32 workgroups per pair do not become co-resident within 0.25 s -- another process holding
// CUs, say -- abandons the exchange.)  Then the tail of the batch is redone with the streaming kernels from the projected
// descriptors, synchronously, and the handle stays on them for the next `rs_backoff_next` batches (64, doubling with every
// further give-up up to 4096) before it tries the resident kernel again: a transient loss of CUs costs seconds, not the
// rest of the process.  Said once per give-up on stderr.  Returns 1 when the device results were rewritten that way (the
// caller repeats its copies), 0 when there was nothing to do.
static int pm_check_resident(urf_pm *h) {
  if (h->h_resid && h->last_P >= 1)
    for (int p = 0; p < h->last_P && p < 64; ++p) h->last_resid[p] = h->h_resid[p];
  if (!(h->h_rs_err && h->h_rs_err[0] != 0)) {
    if (h->rs_wanted && !h->rs_on && h->rs_backoff > 0 && --h->rs_backoff == 0) h->rs_on = true;   // try again with the next batch
    return 0;
  }
  const int what = h->h_rs_err[0];
  const unsigned long long who = ((unsigned long long)(unsigned)h->h_rs_err[3] << 32) | (unsigned)h->h_rs_err[2];
  memset(h->h_rs_err, 0, 4 * sizeof(int));
  URF_HIP(hipMemsetAsync(h->rs_err, 0, 4 * sizeof(int), h->st));
  URF_CHECK(h->last_P >= 1, "the fast Sinkhorn reported a fault and there is no batch to redo");
  const bool was_on = h->rs_on;
  bool what_backoff = false;
  if (what != 2) {   // 1 (or anything else the word may hold): a launch gave up
    h->rs_fallbacks += 1;
    h->rs_backoff = h->rs_backoff_next;
    h->rs_backoff_next = h->rs_backoff_next < 4096 ? h->rs_backoff_next * 2 : 4096;
    fprintf(stderr, "liburf_front: the chip-resident Sinkhorn launch of a %d-pair batch gave up (its workgroups did not become "
            "co-resident within 0.25 s); batch redone with the streaming kernels, which this handle keeps for the next %d batches "
            "(give-up %d of this handle)\n", h->last_P, h->rs_backoff, h->rs_fallbacks);
  } else {
    // integrity: a pair's plan misses its column marginals by more than the bound -- which no correct result does, converged or
    // not.  The batch's tail is redone with the streaming kernels.  A lone event leaves the handle on the resident kernel
    // (nothing says the next launch is affected).  Events that come in a row are not transients: the resident kernel iterates in
    // the scaling domain and its plan entries leave the fp32 range (NaN) when the couplings of a pair span more than e^80 or so
    // between two re-absorptions -- seen with weights of three times the default residual gain in 5 % of random pairs, never on
    // the bench streams (profiles/r05_sweep_strict_vs_exact_other_weights_400.txt).  The log-domain streaming kernels have no
    // such limit: a second event within 16 batches puts the handle on them for a while, like a give-up does.
    int n = 0;
    float worst = 0.0f;
    for (int p = 0; p < h->last_P && p < 64; ++p)
      if ((who >> p) & 1ull) { n += 1; if (!(h->last_resid[p] <= worst)) worst = h->last_resid[p]; }
    h->rs_integrity_pairs += (unsigned long long)n;
    h->rs_integrity_events += 1;
    const bool again = h->rs_integrity_events > 1 && h->batches_seen - h->last_integrity_batch <= 16;
    h->last_integrity_batch = h->batches_seen;
    if (again) {
      h->rs_backoff = h->rs_backoff_next;
      h->rs_backoff_next = h->rs_backoff_next < 4096 ? h->rs_backoff_next * 2 : 4096;
    }
    if (h->rs_integrity_events <= 4 || (h->rs_integrity_events & (h->rs_integrity_events - 1)) == 0)   // (said for the first few, then at powers of two)
      fprintf(stderr, "liburf_front: the Sinkhorn result of %d pair(s) of a %d-pair batch failed the integrity bound (column-marginal "
              "residual %.3g > %.3g); batch redone with the streaming kernels%s (event %d of this handle)\n", n, h->last_P,
              (double)worst, (double)h->resid_bound, again ? ", which the handle keeps for a while: the second event in 16 batches" : "",
              h->rs_integrity_events);
    if (again) what_backoff = true;
  }
  h->rs_on = false;
  const int rc = pm_tail(h, h->last_P, h->last_Z, h->last_ransac, false, h->last_fast);
  if (what == 2 && !what_backoff) h->rs_on = was_on;
  if (rc) return -1;
  URF_HIP(hipStreamSynchronize(h->st));   // (the redone tail has new guard words: pm_guard_redo reads them next)
  if (h->h_rs_err[0] == 2) {
    // the streaming kernels' result fails the bound as well: not the launch -- the pair's couplings (not finite, or sums that
    // leave the fp32 range) -- keep the result, say so once
    if (!h->calib_said) { h->calib_said = true; fprintf(stderr, "liburf_front: the streaming Sinkhorn leaves the same residual: the bound is too tight for these couplings (result kept)\n"); }
    memset(h->h_rs_err, 0, 4 * sizeof(int));
    URF_HIP(hipMemsetAsync(h->rs_err, 0, 4 * sizeof(int), h->st));
  }
  if (h->h_resid)
    for (int p = 0; p < h->last_P && p < 64; ++p) h->last_resid[p] = h->h_resid[p];
  return 1;
}

// after a batch's results have been waited for: the resident Sinkhorn's give-up, then the guarded mode's near-ties.
// > 0: device results were rewritten (the caller repeats its copies); < 0: error
static int pm_after_wait(urf_pm *h) {
  const int r1 = pm_check_resident(h);
  if (r1 < 0) return -1;
  const int r2 = pm_guard_redo(h);
  if (r2 < 0) return -1;
  return r1 + r2;
}

static void pm_collect_times(urf_pm *h) {
  if (!urf::g_profiling) return;
  for (int i = 0; i < PT_COUNT; ++i) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) != hipSuccess) ms = 0.0f;
    h->stage_ms[i] = ms;
  }
  float at = 0.0f;
  for (int l = 0; l < 18; ++l) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, h->ev_attn[l][0], h->ev_attn[l][1]) == hipSuccess) at += ms;
  }
  h->stage_ms[PT_COUNT] = at;
  h->stage_ms[PT_COUNT + 1] = 0.0f;    // (the redo of this batch, if any, happens after this point: urf_pm_stage_ms reads redo_ms)
  h->ev_valid = true;
}

extern "C" void urf_normalize_keypoints(const double *feat, int n, int width, int height, double *out) {
  if (out != feat) memcpy(out, feat, (size_t)259 * n * sizeof(double));
  const int mx = width > height ? width : height;
  for (int c = 0; c < n; ++c) {
    out[(size_t)259 * c + 1] = (feat[(size_t)259 * c + 1] - width / 2) / (mx * 0.7);
    out[(size_t)259 * c + 2] = (feat[(size_t)259 * c + 2] - height / 2) / (mx * 0.7);
  }
}

// upload one pair given as host f64 features (column-major 259 x n).  normalize: the keypoints are pixel coordinates and go
// through PointMatching::NormalizeKeypoints on the way (src/point_matching.cc:63-76: (x - width / 2) / (max(width, height) * 0.7)
// in double, narrowed like SuperGlue::process_input, src/super_glue.cpp:259-275) -- else they are normalised already
// (SuperGlue::infer).  want_xy: keep the pixel coordinates for the outlier stage.  Staging is pinned memory of the handle
// (allocated by the first host call): the conversion is the only pass over the features and the copies are true DMA.
static int pm_upload_pair(urf_pm *h, const double *f0, int n0, const double *f1, int n1, bool normalize, bool want_xy) {
  const size_t nk = (size_t)2 * NP * 4, nxy = (size_t)2 * NP * 2, nx = (size_t)2 * NP * 256;
  if (!h->h_up) {
    URF_HIP(hipHostMalloc((void **)&h->h_up, (nk + nxy + nx) * sizeof(float), hipHostMallocDefault));
    memset(h->h_up, 0, (nk + nxy + nx) * sizeof(float));
    h->up_n[0] = h->up_n[1] = 0;
  }
  float *kin = h->h_up, *kxy = kin + nk, *x = kxy + nxy;
  const double *f[2] = {f0, f1};
  const int n[2] = {n0, n1};
  const int W = h->cfg.image_width, H = h->cfg.image_height, mx = W > H ? W : H;
  for (int im = 0; im < 2; ++im) {
    for (int j = 0; j < n[im]; ++j) {
      const double *col = f[im] + (size_t)259 * j;
      float *k = kin + ((size_t)im * NP + j) * 4;
      if (normalize) {
        k[0] = (float)((col[1] - W / 2) / (mx * 0.7));
        k[1] = (float)((col[2] - H / 2) / (mx * 0.7));
      } else {
        k[0] = (float)col[1]; k[1] = (float)col[2];
      }
      k[2] = (float)col[0];
      kxy[((size_t)im * NP + j) * 2] = want_xy ? (float)col[1] : 0.0f;        // cv::Point2f, point_matching.cc:39-41
      kxy[((size_t)im * NP + j) * 2 + 1] = want_xy ? (float)col[2] : 0.0f;
      float *d = x + ((size_t)im * NP + j) * 256;
      for (int c = 0; c < 256; ++c) d[c] = (float)col[3 + c];                   // :277-283
    }
    // rows the previous call filled beyond this call's count go back to zero (the device buffers hold zeros past the counts)
    const int was = h->up_n[im];
    if (was > n[im]) {
      memset(kin + ((size_t)im * NP + n[im]) * 4, 0, (size_t)(was - n[im]) * 4 * sizeof(float));
      memset(kxy + ((size_t)im * NP + n[im]) * 2, 0, (size_t)(was - n[im]) * 2 * sizeof(float));
      memset(x + ((size_t)im * NP + n[im]) * 256, 0, (size_t)(was - n[im]) * 256 * sizeof(float));
    }
    h->up_n[im] = n[im];
  }
  URF_HIP(hipMemcpyAsync(h->counts, n, 2 * sizeof(int), hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->kin, kin, nk * 4, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->kxy, kxy, nxy * 4, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->x, x, nx * 4, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipStreamSynchronize(h->st));  // (`n` lives on this frame; the pinned staging is free for the next call)
  return 0;
}

static int pm_check(urf_pm *h, int n0, int n1) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(n0 >= 0 && n0 <= NP && n1 >= 0 && n1 <= NP, "keypoint counts (%d, %d) outside [0, %d]", n0, n1, NP);
  // the one-pair host calls use the handle's buffers and result sets: not while a device batch is enqueued or waits for its fetch_end
  URF_CHECK(h->pending_P == 0 && h->bq_n == 0, "a device batch of this handle has not been handed out yet (urf_pm_fetch / urf_pm_fetch_end first)");
  return 0;
}

extern "C" int urf_sg_infer(urf_pm *h, const double *f0, int n0, const double *f1, int n1, int *idx0, int *idx1,
                            double *ms0, double *ms1, float *Zout) {
  if (pm_check(h, n0, n1)) return -2;
  URF_CHECK(n0 >= 1 && n1 >= 1, "SuperGlue needs at least one keypoint per image (profile min, src/super_glue.cpp:63-66)");
  URF_CHECK(f0 && f1 && idx0 && idx1 && ms0 && ms1, "urf_sg_infer: null pointer");
  URF_HIP(hipSetDevice(h->device));
  if (pm_upload_pair(h, f0, n0, f1, n1, false, false)) return -1;
  if (h->calib_left > 0 && h->bq_n == 0 && h->pending_P == 0) {
    if (pm_auto_calibrated(h, 1, pm_calibrate_core(h, 1, kCalibFactor, nullptr))) return -1;
    if (pm_upload_pair(h, f0, n0, f1, n1, false, false)) return -1;
  }
  if (urf::g_profiling) (void)hipEventRecord(h->ev[PT_PREP], h->st);
  if (pm_pipeline(h, 1, Zout != nullptr, false)) return -1;
  for (int pass = 0; pass < 2; ++pass) {   // a second pass only after a redo (resident Sinkhorn give-up, near-tie guard)
    URF_HIP(hipMemcpyAsync(idx0, h->idx0, n0 * sizeof(int), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipMemcpyAsync(idx1, h->idx1, n1 * sizeof(int), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipMemcpyAsync(ms0, h->ms0, n0 * sizeof(double), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipMemcpyAsync(ms1, h->ms1, n1 * sizeof(double), hipMemcpyDeviceToHost, h->st));
    if (Zout)
      URF_HIP(hipMemcpy2DAsync(Zout, (size_t)(n1 + 1) * 4, h->Z, (size_t)LDC * 4, (size_t)(n1 + 1) * 4, n0 + 1,
                               hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipStreamSynchronize(h->st));
    if (pass == 0) pm_collect_times(h);
    const int redo = pm_after_wait(h);
    if (redo < 0) return -3;
    if (redo == 0) break;
  }
  return 0;
}

extern "C" int urf_match(urf_pm *h, const double *f0, int n0, const double *f1, int n1, int outlier_rejection,
                         urf_dmatch *out, int cap) {
  if (pm_check(h, n0, n1)) return -2;
  URF_CHECK(f0 && f1 && out, "urf_match: null pointer");
  if (n0 < 1 || n1 < 1) return 0;
  URF_HIP(hipSetDevice(h->device));
  if (pm_upload_pair(h, f0, n0, f1, n1, true, true)) return -1;      // NormalizeKeypoints on the way (src/point_matching.cc:22-23)
  if (h->calib_left > 0 && h->bq_n == 0 && h->pending_P == 0) {
    if (pm_auto_calibrated(h, 1, pm_calibrate_core(h, 1, kCalibFactor, nullptr))) return -1;
    if (pm_upload_pair(h, f0, n0, f1, n1, true, true)) return -1;
  }
  if (urf::g_profiling) (void)hipEventRecord(h->ev[PT_PREP], h->st);
  if (pm_pipeline(h, 1, false, outlier_rejection != 0)) return -1;
  for (int pass = 0; pass < 2; ++pass) {
    URF_HIP(hipMemcpyAsync(h->h_n, h->nfinal, sizeof(int), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipMemcpyAsync(h->h_matches, h->fmatches, (size_t)NP * sizeof(urf_dmatch), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipStreamSynchronize(h->st));
    if (pass == 0) pm_collect_times(h);
    const int redo = pm_after_wait(h);
    if (redo < 0) return -3;
    if (redo == 0) break;
  }
  const int n = h->h_n[0];
  URF_CHECK(n <= cap, "match buffer too small: %d > cap %d", n, cap);
  memcpy(out, h->h_matches, (size_t)n * sizeof(urf_dmatch));
  return n;
}

extern "C" int urf_match_device_async(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1,
                                      int outlier_rejection) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(P >= 1 && P <= h->maxP, "pairs %d outside [1, %d]", P, h->maxP);
  URF_CHECK(d_slots0 && d_slots1, "urf_match_device: null pointer");
  URF_HIP(hipSetDevice(h->device));
  URF_CHECK(h->pending_P == 0, "urf_match_device_async: the batch enqueued last has not been fetched (urf_pm_fetch / urf_pm_fetch_begin first): a new one would overwrite it");
  URF_CHECK(h->bq_n <= urf_pm::kBegun, "urf_match_device_async: internal: begun queue overflow");
  URF_HIP(hipEventSynchronize(h->ev_done));  // previous batch (pinned pointer table) consumed
  if (h->calib_left > 0 && h->bq_n == 0) {   // the handle's first pairs: measured before they are matched (synchronous, once)
    if (pm_prep_slots(h, P, d_slots0, d_slots1)) return -1;
    if (pm_auto_calibrated(h, P, pm_calibrate_core(h, P, kCalibFactor, nullptr))) return -1;
  }
  if (urf::g_profiling) (void)hipEventRecord(h->ev[PT_PREP], h->st);
  if (pm_prep_slots(h, P, d_slots0, d_slots1)) return -1;
  if (pm_pipeline(h, P, false, outlier_rejection != 0)) return -1;
  URF_HIP(hipMemcpyAsync(h->h_n, h->nfinal, P * sizeof(int), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipMemcpyAsync(h->h_matches, h->fmatches, (size_t)P * NP * sizeof(urf_dmatch), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipEventRecord(h->ev_done, h->st));
  h->pending_P = P;
  return 0;
}

// Guard calibration (include/urf.h): the fast and the exact matcher on the same inputs, the largest difference of their
// log-assignments on the entries a decision can rest on (probability above 0.1 in either), the margin widened where needed.
// The inputs of P pairs (counts, kin, kxy, x) are in place; the exact pass runs in the handle's own buffers and destroys them (the
// caller prepares them again for the batch proper).  margin >= factor x (measured + the descriptor-noise share of the guarded mode).
static int pm_calibrate_core(urf_pm *h, int P, float factor, double *out) {
  hipStream_t st = h->st;
  // (not a batch of the caller's stream: the bookkeeping of the last handed-out batch survives the call)
  const unsigned long long seen = h->pairs_seen;
  const int keep_P = h->last_P;
  const bool keep_Z = h->last_Z, keep_ransac = h->last_ransac, keep_rec = h->flags_recorded, keep_queued = h->redo_queued;
  struct Restore {
    urf_pm *h; int P; bool Z, ransac, rec, queued;
    ~Restore() { h->last_P = P; h->last_Z = Z; h->last_ransac = ransac; h->flags_recorded = rec; h->redo_queued = queued;
                 for (int p = 0; p < h->maxP && p < 64; ++p) h->h_gflags[p] = 0; }
  } restore{h, keep_P, keep_Z, keep_ransac, keep_rec, keep_queued};
  h->calibrating = true;
  const int rc_fast = pm_pipeline(h, P, true, false);                 // the fast pass, Z kept
  h->calibrating = false;
  if (rc_fast) return -1;
  h->pairs_seen = seen;
  URF_HIP(hipStreamSynchronize(st));
  if (h->h_rs_err && h->h_rs_err[0] != 0) {                           // the resident Sinkhorn gave up (or failed its integrity bound): nothing to measure against
    (void)pm_check_resident(h);
    URF_CHECK(false, "urf_pm_calibrate_guard: the chip-resident Sinkhorn launch gave up; call again");
  }
  const size_t zbytes = (size_t)P * (NP + 1) * LDC * sizeof(float);
  // the scratch copy of the fast log-assignments lives with the handle (allocated by build() of a handle that will calibrate itself,
  // else by the first explicit calibration): no hipMalloc / hipFree inside a call documented as asynchronous
  if (!h->calib_zf) {
    URF_HIP(hipMalloc((void **)&h->calib_zf, (size_t)h->maxP * (NP + 1) * LDC * sizeof(float)));
    if (hipMalloc((void **)&h->calib_acc, sizeof(int)) != hipSuccess) { (void)hipFree(h->calib_zf); h->calib_zf = nullptr; URF_CHECK(false, "urf_pm_calibrate_guard: out of device memory"); }
  }
  float *zf = h->calib_zf;
  int *acc = h->calib_acc;
  int rc = hipMemcpyAsync(zf, h->Z, zbytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : -1;
  // the exact pass from the same encoded keypoints (h->x: the fast layers work on their own planes), in this handle's own
  // buffers (no batch is in flight; the device lists of the last fetched batch are overwritten)
  if (!rc) rc = pm_gnn_exact(h, 2 * P, false);
  if (!rc) rc = sg_linear(h, 2 * P, h->x, 256, 256, nullptr, 0, 0, h->wf, h->bf, 256, h->mdesc, 256, false, nullptr);
  if (!rc) rc = pm_tail(h, P, true, false, false, false);
  float worst = 0.0f;
  if (!rc) rc = hipMemsetAsync(acc, 0, sizeof(int), st) == hipSuccess ? 0 : -1;
  if (!rc) rc = launch_guard_z_calib(h->counts, zf, h->Z, logf(0.1f), acc, P, st);
  if (!rc) rc = hipMemcpyAsync(&worst, acc, sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess ? 0 : -1;
  if (!rc) rc = hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
  URF_CHECK(rc == 0, "urf_pm_calibrate_guard: a launch or copy failed");
  const float need = factor * (worst + (h->strict ? 0.0f : kGuardSgDescNoise));   // (strict parity: exact slots, no descriptor noise)
  if (need > h->g_z) h->g_z = need;
  if (worst > h->calib_worst) h->calib_worst = worst;
  if (out) { out[0] = worst; out[1] = h->g_z; }
  return 0;
}

static int pm_prep_slots(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1) {
  for (int p = 0; p < P; ++p) {
    h->h_slotptrs[2 * p] = (const float *)d_slots0[p];
    h->h_slotptrs[2 * p + 1] = (const float *)d_slots1[p];
  }
  URF_HIP(hipMemcpyAsync(h->d_slotptrs, h->h_slotptrs, 2 * P * sizeof(float *), hipMemcpyHostToDevice, h->st));
  return launch_sg_prep_slots(h->d_slotptrs, 2 * P, h->cfg.image_width, h->cfg.image_height, h->counts, h->kin, h->kxy, h->x, h->st);
}

extern "C" int urf_pm_calibrate_guard(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1, double *out) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(h->guarded, "urf_pm_calibrate_guard: the handle is not in a guarded mode (precision 2 or 3)");
  URF_CHECK(P >= 1 && P <= h->maxP && d_slots0 && d_slots1, "urf_pm_calibrate_guard: bad argument");
  URF_CHECK(h->pending_P == 0 && h->bq_n == 0, "urf_pm_calibrate_guard: a batch is in flight or waiting for its urf_pm_fetch_end (hand it out first): the calibration reuses its buffers");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipEventSynchronize(h->ev_done));
  if (pm_prep_slots(h, P, d_slots0, d_slots1)) return -1;
  return pm_calibrate_core(h, P, 1.10f, out);
}

// Automatic calibration (urf_sg_config.calibrate_pairs): the first pairs a guarded handle is given are measured before they are
// matched.  A handle built from a deployment's own weights thereby carries a margin that covers THEIR split-f16 error, not the
// synthetic streams' the built-in constant was measured on.  The maximum over a handful of pairs underestimates the maximum over
// a stream (1.39e-4 on eight bench pairs against 2.0e-4 over both streams), hence the factor kCalibFactor.  Above kGuardSgZCap the error
// model itself is not trusted: a strict handle then redoes EVERY pair in the exact mode (correct, at the exact mode's speed).
static int pm_auto_calibrated(urf_pm *h, int P, int rc) {
  if (rc) {   // (a give-up of the resident Sinkhorn, no memory for the scratch copy): the caller's batch does not fail for it -- the next one is measured
    if (!h->calib_said) { h->calib_said = true; fprintf(stderr, "liburf_front: the automatic guard calibration could not run (%s); it is tried again with the next pairs\n", urf_last_error()); }
    if (++h->calib_failures >= 16) {   // ... but not for ever: every attempt costs a fast and an exact pass
      h->calib_left = 0;
      fprintf(stderr, "liburf_front: the automatic guard calibration failed %d times in a row; the handle keeps the margin %.3g (urf_pm_calibrate_guard remains)\n",
              h->calib_failures, (double)h->g_z);
    }
    return 0;
  }
  h->calib_failures = 0;
  h->calib_left -= P;
  // (the cap is looked at after EVERY measurement: a pipelined loop with batches smaller than calibrate_pairs may never get to
  // measure the rest -- the handle usually holds a begun batch when its next one arrives; the online check takes over from here)
  if (h->calib_worst > kGuardSgZCap && h->strict && h->redo_pairs) h->redo_all = true;
  if (h->calib_left <= 0) {
    h->calib_left = 0;
    fprintf(stderr, "liburf_front: matcher guard calibrated on this handle's first pairs: largest fast-vs-exact difference %.3g on a decisive "
            "entry, margin in use %.3g%s\n", (double)h->calib_worst, (double)h->g_z,
            h->redo_all ? " -- above the cap the error model is trusted to: every pair will be redone in the exact mode" : "");
  }
  return 0;
}

extern "C" int urf_pm_sinkhorn_fallbacks(const urf_pm *h) { return h ? h->rs_fallbacks : -1; }

extern "C" int urf_pm_sinkhorn_integrity(urf_pm *h, double *out, int n) {
  URF_CHECK(h && h->built && out && n >= 1, "urf_pm_sinkhorn_integrity: bad argument");
  const double v[4] = {(double)h->rs_integrity_pairs, (double)h->rs_integrity_events, (double)h->resid_bound, (double)h->pairs_seen};
  for (int i = 0; i < n && i < 4; ++i) out[i] = v[i];
  return 0;
}
extern "C" int urf_pm_sinkhorn_residuals(urf_pm *h, float *out, int P) {
  URF_CHECK(h && h->built && out && P >= 1 && P <= 64, "urf_pm_sinkhorn_residuals: bad argument");
  for (int p = 0; p < P; ++p) out[p] = h->fast ? h->last_resid[p] : 0.0f;
  return 0;
}

extern "C" int urf_pm_near_tie_reruns(urf_pm *h, unsigned long long *out, int n) {
  URF_CHECK(h && h->built && out && n >= 1, "urf_pm_near_tie_reruns: bad argument");
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  v[0] = h->pairs_redone; v[2] = h->cause_thr; v[3] = h->cause_run;     // (host counters: the redo is driven from the host)
  v[1] = h->pairs_seen;
  v[4] = h->pairs_flagged;
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return 0;
}

extern "C" int urf_pm_redo_engine_stats(urf_pm *h, double *out, int n) {
  URF_CHECK(h && h->built && out && n >= 1, "urf_pm_redo_engine_stats: bad argument");
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (h->pool) {
    std::lock_guard<std::mutex> lock(h->pool->mu);
    v[0] = (double)h->pool->passes; v[1] = (double)h->pool->pairs; v[2] = (double)h->pool->merged_passes; v[3] = (double)h->pool->refs;
  }
  for (int i = 0; i < n && i < 4; ++i) out[i] = v[i];
  return 0;
}

extern "C" int urf_pm_guard_state(urf_pm *h, double *out, int n) {
  URF_CHECK(h && h->built && out && n >= 1, "urf_pm_guard_state: bad argument");
  const double v[10] = {(double)h->g_z, (double)h->calib_worst, (double)h->calib_left, h->redo_all ? 1.0 : 0.0,
                        (double)h->online_worst, (double)h->online_pairs, (double)h->margin_raises, (double)h->online_violations,
                        (double)h->audits, (double)h->audit_mismatches};
  for (int i = 0; i < n && i < 10; ++i) out[i] = v[i];
  if (n > 10) out[10] = (double)h->exact_batches;
  return 0;
}

extern "C" int urf_pm_near_tie_flags(urf_pm *h, int *flags, int P) {
  URF_CHECK(h && h->built && flags && P >= 1 && P <= 64, "urf_pm_near_tie_flags: bad argument");
  for (int p = 0; p < P; ++p) flags[p] = h->guarded ? h->last_flags[p] : 0;
  return 0;
}

extern "C" int urf_pm_sync(urf_pm *h) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  pm_collect_times(h);
  return 0;
}

// urf_pm_fetch in two halves.  begin: waits for the batch's fast pass, handles a resident-Sinkhorn give-up, reads the guard
// words and STARTS the exact redo of the flagged pairs (strict parity) on the redo engine's stream; returns 1 when a redo is
// running, 0 when the lists are final already.  The caller may now enqueue this handle's next batch (urf_match_device_async):
// it runs beside the redo.  end: waits for the redo (if any) and hands the lists out.
#ifdef URF_EXPERIMENTS
// diagnostic (experiments build, URF_CHECKSUMS=1): order-independent 64-bit sums of the bit patterns of a stage's tensors, one set
// per batch, so that a run-to-run difference of a periodic stream can be pinned to the stage that produced it
__global__ void urf_checksum_kernel(const unsigned *x, size_t n, unsigned long long *out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
static unsigned long long *g_cks_dev = nullptr, *g_cks_host = nullptr;   // [handle slot 0..3][8]
static int pm_checksum(urf_pm *h, int slot, int k, const void *x, size_t n_words) {
  static const int on = [] { const char *e = urf::exp_env("URF_CHECKSUMS"); return e ? atoi(e) : 0; }();
  if (!on) return 0;
  if (!g_cks_dev) {
    URF_HIP(hipMalloc((void **)&g_cks_dev, 4 * 8 * sizeof(unsigned long long)));
    URF_HIP(hipHostMalloc((void **)&g_cks_host, 4 * 8 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  unsigned long long *d = g_cks_dev + 8 * slot + k;
  URF_HIP(hipMemsetAsync(d, 0, sizeof(*d), h->st));
  hipLaunchKernelGGL(urf_checksum_kernel, dim3(256), dim3(256), 0, h->st, (const unsigned *)x, n_words, d);
  URF_HIP(hipMemcpyAsync(g_cks_host + 8 * slot + k, d, sizeof(*d), hipMemcpyDeviceToHost, h->st));
  return 0;
}
extern "C" int urf_probe_pm_rs_debug(urf_pm *h, unsigned long long *out) {    // the per-iteration bit sums of the handle's last resident Sinkhorn launch
  if (!g_rsdbg_dev[h->cks_slot]) return -1;
  URF_HIP(hipMemcpyAsync(out, g_rsdbg_dev[h->cks_slot], kRsDbgWords * 8, hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipStreamSynchronize(h->st));
  return 0;
}
extern "C" int urf_probe_pm_checksums(urf_pm *h, unsigned long long *out) {   // of the handle's LAST enqueued batch (after its fetch_begin)
  if (!g_cks_host) return -1;
  for (int k = 0; k < 8; ++k) out[k] = g_cks_host[8 * h->cks_slot + k];
  return 0;
}

// what-if (experiments build): URF_DUMMY_LAUNCHES=n dependent launches of an empty / a tiny-store kernel on the redo engine's
// stream per begun batch -- what do kernel boundaries alone cost the other streams?
__global__ void urf_dummy_kernel(int *p, int touch) { if (touch && threadIdx.x == 0) p[blockIdx.x] = touch; }
#endif

extern "C" int urf_pm_fetch_begin(urf_pm *h, int P) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(h->pending_P > 0, "urf_pm_fetch: no batch in flight (urf_match_device_async first)");
  URF_CHECK(P == h->pending_P, "urf_pm_fetch: %d pairs asked, the batch in flight has %d", P, h->pending_P);
  URF_CHECK(h->bq_n < urf_pm::kBegun, "urf_pm_fetch_begin: two batches of this handle have not been handed out yet (urf_pm_fetch_end first)");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipEventSynchronize(h->ev_done));  // only this handle's last batch, not the whole stream
  pm_collect_times(h);
  const int r1 = pm_check_resident(h);
  if (r1 < 0) return -3;
  if (r1 > 0) {   // the tail was redone with the streaming Sinkhorn after a give-up: fetch the rewritten lists
    URF_HIP(hipMemcpyAsync(h->h_n, h->nfinal, P * sizeof(int), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipMemcpyAsync(h->h_matches, h->fmatches, (size_t)P * NP * sizeof(urf_dmatch), hipMemcpyDeviceToHost, h->st));
    URF_HIP(hipStreamSynchronize(h->st));
  }
  const int r2 = pm_begin_batch(h);
  if (r2 < 0) return -3;
#ifdef URF_EXPERIMENTS
  {
    static const int dummy = [] { const char *e = urf::exp_env("URF_DUMMY_LAUNCHES"); return e ? atoi(e) : 0; }();
    static const int dummy_wgs = [] { const char *e = urf::exp_env("URF_DUMMY_WGS"); return e ? atoi(e) : 1; }();
    if (dummy > 0 && h->redo)
      for (int i = 0; i < dummy; ++i) hipLaunchKernelGGL(urf_dummy_kernel, dim3(dummy_wgs), dim3(64), 0, h->redo->st, h->redo->counts, 0);
  }
#endif
  h->pending_P = 0;
  return r2;
}

// 1: urf_pm_fetch_end would not block (the oldest begun batch needed no redo, or its redo has delivered); 0: it would; < 0: error
extern "C" int urf_pm_fetch_ready(urf_pm *h) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(h->bq_n > 0, "urf_pm_fetch_ready: no fetch has begun (urf_pm_fetch_begin first)");
  urf_pm::Begun &e = h->bq[h->bq_head];
  if (e.n == 0) return 1;
  URF_HIP(hipSetDevice(h->device));
  if (pm_ensure_launched(h, e)) return -1;        // (a job that waited in the pool for a companion: nobody came, it goes now)
  const hipError_t q = hipEventQuery(e.ev_done);
  if (q == hipSuccess) return 1;
  if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  URF_HIP(q);
  return 0;
}

extern "C" int urf_pm_fetch_end(urf_pm *h, int P, urf_dmatch *out, int cap, int *nout) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(h->bq_n > 0, "urf_pm_fetch_end: no fetch has begun (urf_pm_fetch_begin first)");
  URF_CHECK(P == h->bq[h->bq_head].P, "urf_pm_fetch_end: %d pairs asked, the batch being handed out has %d", P, h->bq[h->bq_head].P);
  URF_CHECK(out && nout, "urf_pm_fetch: bad argument");
  URF_HIP(hipSetDevice(h->device));
  int set = 0;
  if (pm_end_batch(h, &set, nullptr) < 0) return -3;
  const urf_dmatch *hm = h->hm_set[set];
  const int *hn = h->hn_set[set];
  h->fetched_set = set;
  for (int p = 0; p < P; ++p) URF_CHECK(hn[p] >= 0 && hn[p] <= cap, "match buffer too small: %d > cap %d", hn[p], cap);
  for (int p = 0; p < P; ++p) {
    nout[p] = hn[p];
    memcpy(out + (size_t)p * cap, hm + (size_t)p * NP, (size_t)hn[p] * sizeof(urf_dmatch));
  }
  return 0;
}

extern "C" int urf_pm_fetch(urf_pm *h, int P, urf_dmatch *out, int cap, int *nout) {
  URF_CHECK(out && nout, "urf_pm_fetch: bad argument");
  const int rc = urf_pm_fetch_begin(h, P);
  if (rc < 0) return rc;
  return urf_pm_fetch_end(h, P, out, cap, nout);
}

extern "C" int urf_match_device(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1,
                                int outlier_rejection, urf_dmatch *out, int cap, int *nout) {
  int rc = urf_match_device_async(h, P, d_slots0, d_slots1, outlier_rejection);
  if (rc) return rc;
  return urf_pm_fetch(h, P, out, cap, nout);
}

// sets == nullptr: the handle's configured search (counter-hash sampler, canonical order, confidence stop);
// else: the explicit minimal sets, caller's order, every hypothesis counts
static int pm_find_F(urf_pm *h, const float *pts0, const float *pts1, int n, const int *sets, int iterations,
                     uint8_t *inliers, float *F21, float *score) {
  URF_CHECK(h && h->built, "PointMatching handle is not built");
  URF_CHECK(pts0 && pts1 && inliers && F21 && score && n >= 0 && n <= NP, "urf_ransac_find_F: bad argument");
  URF_CHECK(h->pending_P == 0 && h->bq_n == 0, "urf_ransac_find_F: a device batch of this handle has not been handed out yet (it shares the handle's buffers)");
  memset(inliers, 0, n);
  for (int k = 0; k < 9; ++k) F21[k] = 0.0f;
  *score = 0.0f;
  if (n < 8) return 0;
  URF_HIP(hipSetDevice(h->device));
  if (sets) {
    URF_CHECK(iterations >= 1 && iterations <= h->r_iters, "explicit sets: %d iterations outside [1, %d] (ransac_iterations of the handle)",
              iterations, h->r_iters);
    for (int k = 0; k < iterations * 8; ++k) URF_CHECK(sets[k] >= 0 && sets[k] < n, "explicit sets: index %d out of range", sets[k]);
    URF_HIP(hipMemcpyAsync(h->d_sets, sets, (size_t)iterations * 8 * sizeof(int), hipMemcpyHostToDevice, h->st));
  }
  URF_HIP(hipMemcpyAsync(h->nmatch, &n, sizeof(int), hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->pts0, pts0, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipMemcpyAsync(h->pts1, pts1, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
  URF_HIP(hipStreamSynchronize(h->st));
  if (launch_ransac(h->nmatch, h->pts0, h->pts1, h->ps0, h->ps1, h->pn0, h->pn1, h->T, h->F, h->score, h->ninl,
                    h->cfg.ransac_seed, sets ? iterations : h->r_iters, h->r_sigma, sets ? 0.0 : h->r_conf,
                    sets ? h->d_sets : nullptr, 1, h->matches, h->fmatches, h->nfinal, h->inliers, h->Fbest, h->best_score, 1,
                    h->st))
    return -1;
  URF_HIP(hipMemcpyAsync(inliers, h->inliers, n, hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipMemcpyAsync(F21, h->Fbest, 9 * sizeof(float), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipMemcpyAsync(score, h->best_score, sizeof(float), hipMemcpyDeviceToHost, h->st));
  URF_HIP(hipStreamSynchronize(h->st));
  return 0;
}

extern "C" int urf_ransac_find_F(urf_pm *h, const float *pts0, const float *pts1, int n, uint8_t *inliers, float *F21,
                                 float *score) {
  return pm_find_F(h, pts0, pts1, n, nullptr, 0, inliers, F21, score);
}

extern "C" int urf_ransac_find_F_sets(urf_pm *h, const float *pts0, const float *pts1, int n, const int *sets,
                                      int iterations, uint8_t *inliers, float *F21, float *score) {
  URF_CHECK(sets, "urf_ransac_find_F_sets: null sets");
  return pm_find_F(h, pts0, pts1, n, sets, iterations, inliers, F21, score);
}

// debug / parity tap (tests): the couplings matrix (scores + dustbins) of pair 0 of the LAST call, (n0+1) x (n1+1) f32
extern "C" int urf_sg_debug_couplings(urf_pm *h, int n0, int n1, float *out) {
  URF_CHECK(h && h->built && out && n0 >= 1 && n1 >= 1 && n0 <= NP && n1 <= NP, "urf_sg_debug_couplings: bad argument");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  URF_HIP(hipMemcpy2D(out, (size_t)(n1 + 1) * 4, h->C, (size_t)LDC * 4, (size_t)(n1 + 1) * 4, n0 + 1, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int urf_pm_device_results(urf_pm *h, const urf_dmatch **d_matches, const int **d_counts) {
  URF_CHECK(h && h->built && d_matches && d_counts, "urf_pm_device_results: bad argument");
  *d_matches = h->fm_set[h->fetched_set];     // the set of the batch handed out last (urf_pm_fetch / urf_pm_fetch_end)
  *d_counts = h->nf_set[h->fetched_set];
  return 0;
}

extern "C" int urf_pm_stage_ms(urf_pm *h, float *ms, int n) {
  URF_CHECK(h && ms, "urf_pm_stage_ms: null");
  URF_CHECK(h->ev_valid, "no timed call yet (urf_set_profiling(1) before the call)");
  h->stage_ms[PT_COUNT + 1] = h->redo_ms;
  for (int i = 0; i < n && i <= PT_COUNT + 1; ++i) ms[i] = h->stage_ms[i];
  return n < PT_COUNT + 2 ? n : PT_COUNT + 2;
}

// Put this matcher on the SuperPoint handle's stream: SP(b) -> match(b) -> SP(b+1)
// then run in order on ONE stream with no host synchronisation in between.
extern "C" void *urf_sp_stream(urf_sp *h);
extern "C" int urf_pm_share_stream(urf_pm *h, urf_sp *sp) {
  URF_CHECK(h && h->built && sp, "urf_pm_share_stream: handles must be built");
  void *st = urf_sp_result_stream(sp);     // (the stream on which SuperPoint's slots become final)
  URF_CHECK(st, "urf_pm_share_stream: SuperPoint handle is not built");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamSynchronize(h->st));
  if (h->own_stream) (void)hipStreamDestroy(h->st);
  h->st = (hipStream_t)st;
  h->own_stream = false;
  return 0;
}

extern "C" void *urf_pm_stream_(urf_pm *h) { return h && h->built ? (void *)h->st : nullptr; }
extern "C" void *urf_pm_stream(urf_pm *h) { return urf_pm_stream_(h); }
extern "C" int urf_pm_device_(urf_pm *h) { return h ? h->device : 0; }

// Two-stream overlap (bench): SuperPoint of the NEXT batch is bandwidth/MFMA
// complementary to the Sinkhorn iterations of the current one.
//   urf_pm_wait_for_sp(pm, sp): the matcher's stream waits until everything already
//     enqueued on the SuperPoint stream has finished (features ready).
//   urf_sp_wait_for_sinkhorn(sp, pm): the SuperPoint stream waits until the matcher
//     has reached the Sinkhorn stage of its last enqueued batch.
extern "C" void *urf_sp_stream(urf_sp *h);
extern "C" void *urf_sp_result_stream(urf_sp *h);
extern "C" int urf_pm_wait_for_sp(urf_pm *h, urf_sp *sp) {
  URF_CHECK(h && h->built && sp, "urf_pm_wait_for_sp: bad handle");
  hipStream_t ss = (hipStream_t)urf_sp_result_stream(sp);   // where the slots become final
  URF_CHECK(ss, "SuperPoint handle is not built");
  URF_HIP(hipSetDevice(h->device));
  if (ss == h->st) return 0;
  URF_HIP(hipEventRecord(h->ev_ext, ss));
  URF_HIP(hipStreamWaitEvent(h->st, h->ev_ext, 0));
  return 0;
}
extern "C" int urf_pm_wait_event(urf_pm *h, void *event) {
  URF_CHECK(h && h->built && event, "urf_pm_wait_event: bad argument");
  URF_HIP(hipSetDevice(h->device));
  URF_HIP(hipStreamWaitEvent(h->st, (hipEvent_t)event, 0));
  return 0;
}
extern "C" int urf_sp_wait_for_sinkhorn(urf_sp *sp, urf_pm *h) {
  URF_CHECK(h && h->built && sp, "urf_sp_wait_for_sinkhorn: bad handle");
  hipStream_t ss = (hipStream_t)urf_sp_stream(sp);
  URF_CHECK(ss, "SuperPoint handle is not built");
  URF_HIP(hipSetDevice(h->device));
  if (ss == h->st) return 0;
  URF_HIP(hipStreamWaitEvent(ss, h->ev_sink, 0));
  return 0;
}
