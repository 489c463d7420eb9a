/* urf.h -- C ABI of liburf_front.so, the MI355X-native front-end that replaces
 * UR-MVO's TensorRT path (SuperPoint -> SuperGlue -> epipolar RANSAC).
 *
 * Plain pointers and sizes only; no torch / Eigen / OpenCV types.  Every entry
 * point names the reference interface it replaces (paths relative to the
 * UR-MVO tree).  The C++ shim headers include/super_point.h, super_glue.h and
 * point_matching.h rebuild the reference's classes on top of this ABI
 * (INTEGRATION.md shows the binding a maintainer adds).
 *
 * Return value: 0 = ok, <0 = error (urf_last_error() has the text).  No
 * exceptions cross this boundary; on error outputs are left untouched
 * (reference contract: bool returns, src/tracking.cc:328-331,346-350).
 * Threading: a handle may be used from any host thread, one call at a time per
 * handle (the reference serialises with _gpu_mutex, src/tracking.cc:325,345);
 * each call binds the handle's HIP device on entry, no thread-local state.
 */
#ifndef URF_H_
#define URF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URF_FEAT_ROWS 259        /* score, x, y, 256-d descriptor (src/super_point.cpp:364-384) */
#define URF_MAX_KEYPOINTS 1024   /* TensorRT profile cap, src/super_glue.cpp:67-68,97-98 */
#define URF_SP_BLOB_FLOATS 1300865
#define URF_SG_BLOB_FLOATS 12003905

const char *urf_last_error(void);
const char *urf_build_info(void);   /* build time of this binary and the guard constants compiled into it */
int urf_device_count(void);

/* ------------------------------------------------------------ SuperPoint -- */
/* SuperPointConfig, include/read_configs.h:9-18 (tensor names, dla_core and
 * file names live in the C++ shim; they carry no meaning for this back-end). */
typedef struct {
  int max_keypoints;          /* <= URF_MAX_KEYPOINTS, or -1 (= cap) */
  double keypoint_threshold;
  int remove_borders;
  int max_height, max_width;  /* arena is sized for these at build(); 0 -> 1500 (profile max, src/super_point.cpp:55-60) */
  int max_batch;              /* frames per urf_sp_infer_batch call; 0 -> 1 */
  int device;                 /* HIP device ordinal */
  /* 0 = exact fp32 (bit-identical to the oracle, default); 1 = fast: the eight
   * 3x3 convolutions run on the f16 matrix core with split operands
   * (fp32-equivalent accuracy, not bit-reproducible);
   * 2 = guarded fast: as 1, and every discrete decision (NMS maximum, 0.0005 threshold, top-k cut) that sits closer to
   * its alternative than the fast mode's error is detected on the device.  A top-k cut with up to 8 candidates inside
   * the error band is resolved per candidate: the exact mode's convolution stack is run on just the receptive fields
   * of their cells and the band's members are ranked by their exact scores.  Any other near-tie: the frame is redone
   * in the exact mode inside the library before its slot is handed on.  Slot header word 1 says which: 1 = redone
   * whole, 2 = cut resolved per candidate, 0 = the fast pass stands.  Either way the
   * keypoint SET of every frame is the exact mode's.  urf_sp_near_tie_reruns() counts both.
   * 3 = strict parity: the value a pipeline passes to BOTH handles when keypoints and match
   * index lists must be the oracle's.  For SuperPoint it is the exact mode (slots bit-identical to the oracle: scores,
   * order, descriptors); the matcher's half is described at urf_sg_config.precision. */
  int precision;
  /* guarded fast mode (precision 2), error model of a fast-mode heat-map value s: |fast - exact| <= guard_delta s (1 - s) +
   * guard_ulps ulp(s).  0 = the constants measured on the bench streams (1.6e-4, 8; urf_build_info() prints them);
   * urf_sp_calibrate_guard() widens either where a deployment's frames need it.  (Appended fields; all-zero = defaults.) */
  float guard_delta;
  float guard_ulps;
} urf_sp_config;

typedef struct urf_sp urf_sp;

/* SuperPoint::SuperPoint + build(), src/super_point.cpp:13-102.  `blob` is the
 * SP weight container: URF_SP_BLOB_FLOATS f32, host memory. */
int urf_sp_create(const urf_sp_config *cfg, urf_sp **out);
int urf_sp_build(urf_sp *h, const float *blob, size_t n_floats);
/* engine_file analogue (save_engine/deserialize_engine, :402-438): a packed
 * weight file "URFW" + kind + count + f32 payload. */
int urf_sp_build_file(urf_sp *h, const char *path);
int urf_weights_save(const char *path, int kind /*1=SP,2=SG*/, const float *blob, size_t n_floats);
/* build() from the reference's own configuration, src/super_point.cpp:18-32,99-101 (and src/super_glue.cpp:21-33,145-146):
 * deserialize_engine() if `engine_file` exists; otherwise read the initialisers of `onnx_file` (a dependency-free protobuf
 * wire-format reader: parameter names of superpoint/SP/model.py / of the public SuperGlue module when the exporter kept them,
 * else the Conv nodes in graph order), build from them, and save_engine(): write the URFW container to `engine_file`. */
int urf_sp_build_config(urf_sp *h, const char *engine_file, const char *onnx_file);
/* the ONNX half alone: kind 1 = SuperPoint (URF_SP_BLOB_FLOATS), 2 = SuperGlue (URF_SG_BLOB_FLOATS; BatchNorm folded, attention
 * channels head-major); the packed blob is what urf_sp_build / urf_pm_build take and urf_weights_save writes */
int urf_onnx_import(const char *onnx_file, int kind, float *blob, size_t n_floats);
void urf_sp_destroy(urf_sp *h);

/* SuperPoint::infer(image, mask, features), src/super_point.cpp:121-156.
 * img: u8 rows x cols, row stride `step` bytes (cv::Mat::step).  mask: NULL
 * (cv::Mat::empty()) or u8 rows x cols, non-zero = keep, stride mstep.
 * feat: caller buffer for a column-major 259 x cap f64 matrix
 * (Eigen::Matrix<double,259,Dynamic> storage); *K = columns written. */
int urf_sp_infer(urf_sp *h, const uint8_t *img, int rows, int cols, size_t step,
                 const uint8_t *mask, size_t mstep, double *feat, int cap, int *K);

/* Batch of B same-sized frames (host pointers).  feat: B matrices of 259 x cap,
 * Kout: B counts.  One upload, one pipeline, one download. */
int urf_sp_infer_batch(urf_sp *h, int B, const uint8_t *const *imgs, int rows, int cols,
                       size_t step, double *feat, int cap, int *Kout);

/* Device-resident variant: d_imgs = device pointer to B contiguous u8 frames
 * (rows*cols each).  Results stay on the GPU as feature slots
 * (urf_slot_bytes() each, layout in DESIGN.md): the SuperGlue stage and the
 * RCCL all-gather consume them without touching the host.  Asynchronous on the
 * handle's stream; urf_sp_sync() waits. */
size_t urf_slot_bytes(void);
int urf_sp_infer_device(urf_sp *h, int B, const uint8_t *d_imgs, int rows, int cols,
                        void *d_slots);
int urf_sp_sync(urf_sp *h);
/* copy one slot (device) to a host 259 x cap f64 matrix (exact widening).  Synchronous on the NULL stream: the
 * producer of the slot must have finished (urf_sp_sync, or a fetched match batch that consumed it). */
int urf_slot_to_host(const void *d_slot, double *feat, int cap, int *K);

/* guarded fast mode (precision 2), counters since build(): out[0] = frames redone whole in the exact mode, out[1] =
 * frames processed, out[2] = frames whose top-k cut was resolved per candidate, out[3..5] = frames flagged by the
 * threshold band / an NMS near-tie / more than 8 candidates at the cut (these are the frames of out[0]), out[6] =
 * candidates resolved (n <= 8 values are written; zeros in the other modes).  Waits for the handle's stream. */
int urf_sp_near_tie_reruns(urf_sp *h, unsigned long long *out, int n);

/* guarded fast mode: check the error model the guard rests on -- |fast - exact| <= delta s (1 - s) + c eps s for every heat-map
 * value s -- on B representative frames (host pointers / device pointer), and widen delta and c where these frames need it
 * (never narrows).  The built-in constants (delta 1.6e-4, c 8) were measured on the synthetic bench streams; a deployment with
 * its own weights calls this once after build() with a batch of its own frames.  out[0] = the delta the frames needed, out[1] =
 * the c they needed, out[2], out[3] = the constants in use after the call (out may be null).  Synchronous. */
int urf_sp_calibrate_guard(urf_sp *h, int B, const uint8_t *const *imgs, int rows, int cols, size_t step, double *out);
int urf_sp_calibrate_guard_device(urf_sp *h, int B, const uint8_t *d_imgs, int rows, int cols, double *out);

/* debug / parity taps (tests): dense tensors of the LAST single-frame call.
 * which: 0 = post-NMS scores [Hs][Ws], 1 = pre-NMS heat map [Hs][Ws],
 * 2 = dense descriptors [Hc][Wc][256], 100+i = conv i output (NHWC). */
int urf_sp_debug_tensor(urf_sp *h, int which, float *out, size_t n_floats);

/* ---------------------------------------------- SuperGlue / PointMatching -- */
/* SuperGlueConfig, include/read_configs.h:20-29 */
typedef struct {
  int image_width, image_height;
  double matching_threshold;
  int sinkhorn_iterations;    /* 0 -> 100 (src/super_glue.cpp:463) */
  int max_pairs;              /* pairs per batched call; 0 -> 1 */
  int device;
  /* outlier stage (replaces cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask),
   * src/point_matching.cc:50, with the in-tree 8-point search of src/epipolar_geometry.cc:161-205) */
  int ransac_iterations;      /* hypotheses at most; 0 -> 200 (EpipolarGeometry's default, src/tracking.cc:52-55) */
  float ransac_sigma;         /* inlier gate as EpipolarGeometry states it: squared distance to the epipolar line
                               * <= 3.841 sigma^2 in both images; 0 -> derived from ransac_threshold_px */
  uint32_t ransac_seed;
  /* 0 = exact fp32 (bit-identical to the oracle, default); 1 = fast: the 18 GNN
   * layers run on the f16 matrix core with split operands (fp32-equivalent
   * accuracy, not bit-reproducible); 2 = guarded fast: as 1, and a pair in which a row's or
   * column's best assignment lies within the fast PIPELINE's error (5e-4 on the log-assignment: the fast matcher's own
   * error plus what the fast SuperPoint's descriptor noise induces) of the matching threshold or of its runner-up is
   * FLAGGED, not redone: urf_pm_near_tie_flags() names the pairs of the batch just fetched, urf_pm_near_tie_reruns()
   * counts them.  Pairs that are not flagged have the exact pipeline's match set; a flagged pair keeps the fast lists
   * unless redo_flagged_pairs = 1.
   * 3 = strict parity, for slots made by an exact-mode SuperPoint (precision 0 or 3): the fast
   * matcher with its margin shrunk to the matcher's OWN error (2.2e-4; the inputs carry no noise), and every flagged pair
   * redone by the exact matcher on the same slots inside the library, behind the fast pass, before the lists are handed
   * out.  The redo's inputs are bit-identical to the oracle's, so its lists are the oracle's: every pair's match list
   * equals the exact mode's index for index. */
  int precision;
  /* the reference call's own parameters (appended fields; all-zero = the reference's values):
   * ransac_threshold_px: distance to the epipolar line in pixels, findFundamentalMat's 3rd argument.  Used when
   *   ransac_sigma == 0: sigma = px / sqrt(3.841), i.e. the same gate "both distances <= px"; 0 -> 3.0.
   * ransac_confidence: findFundamentalMat's 4th argument.  The hypotheses are walked in order and every new
   *   best one shrinks the number still evaluated to the smallest k with (1 - w^8)^k <= 1 - confidence
   *   (w = its inlier ratio; OpenCV's RANSACUpdateNumIters for 8 model points).  0 -> 0.99; < 0 -> off: all
   *   ransac_iterations hypotheses count (EpipolarGeometry::_find_F). */
  float ransac_threshold_px;
  float ransac_confidence;
  /* guard of the fast matcher (precision 2 and 3; appended fields, all-zero = the mode's defaults):
   * redo_flagged_pairs: 0 = the mode's default (2: flagged pairs are reported only; 3: redone in the exact mode),
   *   1 = redo them, -1 = never redo (precision 2 only; precision 3 without the redo is not strict and is refused),
   *   2 = redo them and NEVER divert whole batches to the exact mode (a strict handle whose guard flags more than half of the
   *   last 32 pairs -- or whose measured error is above the cap -- otherwise runs its next 64 batches in the exact mode itself:
   *   the same lists for less than fast pass + redo; urf_pm_guard_state out[10] counts those batches).
   * guard_margin: the margin on the log-assignment within which a decisive entry counts as near-tied; 0 = the mode's
   *   default (2: 5e-4, 3: 2.2e-4).  urf_pm_calibrate_guard() widens it where a deployment's pairs need it. */
  int redo_flagged_pairs;
  float guard_margin;
  /* outlier_stage (appended; 0 = default): 0 = the in-tree 8-point search configured above; 1 = the reference's own call,
   * cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, ransac_threshold_px (3), ransac_confidence (0.99), mask), as
   * OpenCV 4.2.0 publishes it (7-point minimal sets drawn by cv::RNG((uint64)-1), up to three models per set, at most 1000
   * iterations shrunk by RANSACUpdateNumIters, the mask of the best hypothesis, LMedS below 15 matches; DESIGN.md section
   * 13).  Restated, NOT verified against an OpenCV binary (there is none in this image; the reference holds no vector);
   * ransac_iterations / ransac_sigma / ransac_seed are unused.  The matches are walked in list order, as the reference's
   * points0 / points1 are. */
  int outlier_stage;
  /* integrity bound of the fast Sinkhorn (precision 1, 2, 3; appended, 0 = default 1e-4, < 0 = off): the decode sums every
   * column of the plan it reads.  An iteration ends with the column update, so the column marginals are 1 to rounding in every
   * correct result, converged or not; a pair whose columns miss by more than this (or hold a NaN) has its batch's tail redone with
   * the streaming kernels before the lists are handed out.  It catches a damaged last iteration / final potentials / couplings,
   * not an earlier transient the later iterations absorbed.  urf_pm_sinkhorn_integrity() counts,
   * urf_pm_sinkhorn_residuals() reports the measured values. */
  float sinkhorn_residual_bound;
  /* automatic calibration of the matcher's guard (precision 2, 3; appended): the first calibrate_pairs pairs the handle is given
   * (by any entry: urf_match, urf_sg_infer, urf_match_device_async) are measured BEFORE they are matched -- the fast against the
   * exact matcher, as urf_pm_calibrate_guard does -- and the margin becomes at least 2.5 x the largest difference seen (the
   * maximum over eight pairs underestimates a stream's by up to 2.3 x, measured), so that the strict guarantee follows the deployment's own weights
   * instead of the synthetic ones the built-in 2.2e-4 was measured on.  0 = the mode's default (3: 8 pairs; 2: none), < 0 =
   * never.  Costs one exact pass per measured batch, once; reported on stderr.  Above 2.5e-3 a strict handle redoes EVERY pair
   * in the exact mode (still the oracle's lists, at the exact mode's speed) and says so. */
  int calibrate_pairs;
  /* the redo engine of a strict handle (appended; all-zero = defaults: an engine of the handle's own, every batch's redo launched
   * at its urf_pm_fetch_begin -- the fastest).  redo_shared_engine != 0: the strict handles of one device
   * that were built from the same weights with the same configuration share ONE engine (48 MB of weights and an arena less per
   * extra handle; 7 % slower in the benched loop: the handles' redo passes then queue on one stream).  redo_merge > 0 (shared
   * engines only): a batch's redo waits in the engine's pool for the next urf_pm_fetch_begin of any sharing handle and goes
   * through the engine together with that batch's flagged pairs -- fewer, larger passes; launched at once when somebody asks
   * for it (urf_pm_fetch_ready / _end).  Measured slower still (one more step of latency). */
  int redo_merge;
  int redo_shared_engine;
  /* online check of the guard's error model (strict parity; appended, 0 = default): every exact redo leaves the exact
   * log-assignment of a pair the fast pass has also matched; the largest fast-vs-exact difference on the entries the fast
   * decisions rested on is folded into the handle's margin when the redo is handed out (raised to 1.6 x the difference once it
   * exceeds margin / 1.6, said on stderr; above 2.5e-3 every pair is redone) -- urf_pm_guard_state() reports it.  audit_period:
   * one UNFLAGGED pair of every audit_period-th begun batch goes through the exact engine as well (the only sample on pairs the
   * guard passed: its exact index list must equal the fast one); 0 = 256, < 0 = no audits. */
  int audit_period;
} urf_sg_config;

typedef struct { int queryIdx, trainIdx; float distance; } urf_dmatch; /* cv::DMatch fields used at src/point_matching.cc:37 */

typedef struct urf_pm urf_pm;

/* PointMatching::PointMatching -> SuperGlue::build, src/point_matching.cc:6-12,
 * src/super_glue.cpp:21-147.  blob: SG weight container, host memory. */
int urf_pm_create(const urf_sg_config *cfg, urf_pm **out);
int urf_pm_build(urf_pm *h, const float *blob, size_t n_floats);
int urf_pm_build_file(urf_pm *h, const char *path);
int urf_pm_build_config(urf_pm *h, const char *engine_file, const char *onnx_file);   /* as urf_sp_build_config */
void urf_pm_destroy(urf_pm *h);

/* PointMatching::NormalizeKeypoints, src/point_matching.cc:63-76 (host, f64). */
void urf_normalize_keypoints(const double *feat, int n, int width, int height, double *out);

/* SuperGlue::infer(features0, features1, indices0, indices1, mscores0,
 * mscores1), src/super_glue.cpp:166-241.  f0/f1: column-major 259 x n f64 with
 * ALREADY normalised keypoints.  Zout: optional (n0+1)x(n1+1) f32
 * log-assignment (the TensorRT output tensor). */
int urf_sg_infer(urf_pm *h, const double *f0, int n0, const double *f1, int n1,
                 int *idx0, int *idx1, double *ms0, double *ms1, float *Zout);

/* debug / parity tap (tests): the couplings matrix of the last urf_sg_infer call -- scores m0^T m1 / 16 with the
 * dustbin row and column (src/super_glue.cpp:466-474), (n0+1) x (n1+1) f32: the input of the Sinkhorn iterations */
int urf_sg_debug_couplings(urf_pm *h, int n0, int n1, float *out);

/* PointMatching::MatchingPoints(features0, features1, matches,
 * outlier_rejection), src/point_matching.cc:14-61.  Returns the match count
 * (>=0) or <0 on error.  The cv::findFundamentalMat call (:50) is replaced by
 * the in-tree 8-point RANSAC of src/epipolar_geometry.cc:161-205. */
int urf_match(urf_pm *h, const double *f0, int n0, const double *f1, int n1,
              int outlier_rejection, urf_dmatch *out, int cap);

/* Device-resident batch: pair p matches slot a[p] against slot b[p] (device
 * slot pointers).  out: P x cap matches, nout: P counts (host). */
int urf_match_device(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1,
                     int outlier_rejection, urf_dmatch *out, int cap, int *nout);
/* same, asynchronous: results stay in the handle until urf_pm_fetch(h, P, ...) with the same P; a fetch without a
 * batch in flight, or for another pair count, is an error. */
int urf_match_device_async(urf_pm *h, int P, const void *const *d_slots0,
                           const void *const *d_slots1, int outlier_rejection);   /* error while the batch enqueued last is unfetched */
int urf_pm_fetch(urf_pm *h, int P, urf_dmatch *out, int cap, int *nout);
/* urf_pm_fetch in two halves, for callers that keep the GPU busy across a redo (strict parity, precision 3; in the other
 * modes begin never returns 1).  begin: waits for the batch's fast pass, reads the guard words and STARTS the exact redo of
 * the flagged pairs on the redo engine's own stream; returns 1 when a redo is running, 0 when the lists are final, <0 on
 * error.  Between the halves the caller may enqueue this handle's NEXT batches (urf_match_device_async): they run beside
 * the redo.  A handle holds at most TWO begun batches (three result sets: two begun, one being computed); they are handed
 * out in the order they were begun.  ready: 1 when end would not block (the oldest begun batch needed no redo, or its redo
 * has delivered), 0 when it would; never blocks.  end: waits for the redo of the oldest begun batch and hands its lists
 * out.  urf_pm_fetch = begin + end. */
int urf_pm_fetch_begin(urf_pm *h, int P);
int urf_pm_fetch_ready(urf_pm *h);
int urf_pm_fetch_end(urf_pm *h, int P, urf_dmatch *out, int cap, int *nout);
int urf_pm_sync(urf_pm *h);
/* Run this matcher on the SuperPoint handle's stream (same device): SP(b),
 * match(b), SP(b+1) ... then execute in order on one HIP stream, the host only
 * waits in urf_pm_fetch() for the batch it reads.  `sp` must outlive `h`. */
int urf_pm_share_stream(urf_pm *h, urf_sp *sp);
/* Two-stream pipelining instead: the matcher waits for the SuperPoint work enqueued
 * so far (features ready); SuperPoint waits until the matcher's last batch has
 * reached its Sinkhorn stage, so the next batch's convolutions (MFMA-bound) run
 * beside the Sinkhorn iterations (cache-bandwidth-bound). */
int urf_pm_wait_for_sp(urf_pm *h, urf_sp *sp);
int urf_sp_wait_for_sinkhorn(urf_sp *sp, urf_pm *h);
/* the matcher's stream waits for an event the caller has recorded (hipEvent_t) -- e.g. right behind ONE SuperPoint call on
 * urf_sp_result_stream(), when later calls have been enqueued behind it already and urf_pm_wait_for_sp would wait for those too */
int urf_pm_wait_event(urf_pm *h, void *event);
void *urf_sp_stream(urf_sp *h);
/* urf_sp_stream: the stream a call READS its frames on (order producers -- an undistortion, a copy -- against it).
 * urf_sp_result_stream: the stream on which a call's slots become FINAL (order consumers -- an all-gather, a copy of the
 * headers -- against it).  The same stream except in the guarded fast mode, whose exact pass and descriptor tail run on a
 * second stream so that the next call's fast pass overlaps them. */
void *urf_sp_result_stream(urf_sp *h);
/* the handles' HIP streams (hipStream_t), for callers that order their own work (an RCCL
 * all-gather of the slots, torch ops) against the library's with events instead of host syncs */
void *urf_pm_stream(urf_pm *h);

/* EpipolarGeometry::_find_F, src/epipolar_geometry.cc:161-205: 8-point RANSAC
 * on n pixel correspondences (host arrays of x,y pairs).  Returns 0;
 * *score = best score, inliers n bytes, F21 9 floats row-major. */
int urf_ransac_find_F(urf_pm *h, const float *pts0, const float *pts1, int n,
                      uint8_t *inliers, float *F21, float *score);
/* The same search over EXPLICIT minimal sets -- the reference's _vSets (src/epipolar_geometry.cc:52-71):
 * sets[it*8 + j] indexes the n correspondences as given; the matches are walked in the caller's order (no
 * canonical sort) and all `iterations` hypotheses count.  With the sets of urf_minimal_sets(URF_SAMPLER_GLIBC,
 * 0, ...) this evaluates exactly the hypotheses of the reference's first reconstruct() call in a process. */
int urf_ransac_find_F_sets(urf_pm *h, const float *pts0, const float *pts1, int n, const int *sets, int iterations,
                           uint8_t *inliers, float *F21, float *score);
/* Minimal-set samplers (host).  URF_SAMPLER_HASH: the counter hash of this build (order-independent, parallel).
 * URF_SAMPLER_GLIBC: the reference's own stream -- Random::RandomInt over glibc rand() after srand(seed)
 * (src/epipolar_geometry.cc:56-71,100-117; glibc's TYPE_3 additive feedback generator restated in
 * epipolar_api.hip) with the swap-with-back draw without replacement.  sets: iterations x 8 indices < n. */
#define URF_SAMPLER_HASH 0
#define URF_SAMPLER_GLIBC 1
int urf_minimal_sets(int sampler, uint32_t seed, int n, int iterations, int *sets);

/* EpipolarGeometry(K, sigma, iterations) + reconstruct(vKeys1, vKeys2,
 * vMatches12, T21, vP3D, vbTriangulated), include/epipolar_geometry.h:20-40,
 * src/epipolar_geometry.cc:18-98 (monocular initialisation, src/tracking.cc:559).
 * keys: n x (x,y) pixel coordinates; matches12[n1] = index into keys2 or -1.
 * T21: 4x4 row-major, P3D: n1 x 3, triangulated: n1 bytes; *model: 0 = the
 * homography won, 1 = the fundamental matrix; scores[2] = {SH, SF}.
 * Returns 1 if the initialisation is accepted, 0 if not, <0 on error.  The
 * explicit seed replaces the process-global srand(0) (:100-112). */
typedef struct {
  float K[9]; float sigma; int iterations; uint32_t seed;
  int sampler;   /* 0 = URF_SAMPLER_HASH, the default; 1 = URF_SAMPLER_GLIBC, the rand stream of the reference restarted from the seed */
} urf_epi_config;
int urf_epipolar_reconstruct(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1,
                             const float *keys2, int n2, const int *matches12, float *T21,
                             float *P3D, uint8_t *triangulated, int *model, float *scores);
/* same with explicit minimal sets (cfg->iterations x 8 indices into the list of valid matches, in vMatches12 order) */
int urf_epipolar_reconstruct_sets(urf_pm *h, const urf_epi_config *cfg, const float *keys1, int n1,
                                  const float *keys2, int n2, const int *matches12, const int *sets, float *T21,
                                  float *P3D, uint8_t *triangulated, int *model, float *scores);

/* ------------------------------------------------------------- exchange (N GPUs) -- */
/* The data-parallel layer around the path (SURVEY.md section 8 e): the caller of SuperPoint / PointMatching,
 * Tracking::ExtractFeatureAndMatch (src/tracking.cc:338-377), sharded over the GPUs of one node.  Rank r (a
 * process, or a device of one process) runs SuperPoint on frames [r*n, (r+1)*n) of a step, ONE all-gather of
 * the feature slots (RCCL over xGMI) makes every frame's features local everywhere, rank r matches the pairs
 * whose second frame it owns, and the match lists are gathered to the rank that runs the serial tracker.
 * world == 1 without an id never touches RCCL (librccl is dlopen'ed by the first real communicator). */
#define URF_COMM_ID_BYTES 128
typedef struct urf_comm urf_comm;
/* rank 0: ncclGetUniqueId; ship the 128 bytes to the other ranks over any host channel */
int urf_comm_unique_id(void *id);
/* one rank per process.  id == NULL is allowed for world == 1 only (plain device copies, no RCCL);
 * with an id a world of one still runs the RCCL calls (tests) */
int urf_comm_init(int world, int rank, int device, const void *id, urf_comm **out);
/* one process driving ndev devices: out[i] is the communicator of devices[i] (ncclCommInitAll) */
int urf_comm_init_all(int ndev, const int *devices, urf_comm **out);
/* a host thread that drives SEVERAL devices' communicators (urf_comm_init_all) brackets the collective calls it makes for
 * them with these (ncclGroupStart / ncclGroupEnd): issued one by one, the first rank's call would wait for peers the thread
 * has not called yet.  Not needed with one rank per process; no-ops while RCCL is not loaded. */
int urf_comm_group_start(void);
int urf_comm_group_end(void);
/* one process acting as `world` logical ranks on ONE device (single-GPU rigs and tests of the N > 1 data path): out[r] is
 * rank r's communicator.  Device copies instead of RCCL; a collective completes when every rank has made the call (any
 * rank order, from one host thread), with the stream semantics of the real thing: results are ordered behind every rank's
 * producer stream, and a rank's stream may reuse its send buffer afterwards. */
int urf_comm_init_loopback(int world, int device, urf_comm **out);
void urf_comm_destroy(urf_comm *c);
int urf_comm_world(const urf_comm *c);
int urf_comm_rank(const urf_comm *c);
/* d_all[world * nslots] <- every rank's d_local[nslots] feature slots (urf_slot_bytes() each), in rank order =
 * global frame order; enqueued on `stream` (hipStream_t): order it against urf_sp_stream / urf_pm_stream with
 * events, the host never waits */
int urf_comm_allgather_slots(urf_comm *c, const void *d_local, int nslots, void *d_all, void *stream);
/* `root` receives every rank's `bytes` at d_recv + rank * bytes (device memory; e.g. the buffers of
 * urf_pm_device_results); the other ranks may pass d_recv = NULL */
int urf_comm_gather(urf_comm *c, const void *d_send, size_t bytes, void *d_recv, int root, void *stream);
/* pairs (first[j], second[j]), j < per_rank, that `rank` matches in one step, as indices into the gathered
 * slots (global frame g = rank * per_rank + j against g - 1); first[0] == -1 on rank 0 = the last frame of the
 * previous step, carried by the caller.  Host-only. */
int urf_comm_plan_pairs(int world, int rank, int per_rank, int *first, int *second);
/* device buffers holding the batch of a matcher that was handed out last (urf_pm_fetch / urf_pm_fetch_end): matches
 * [max_pairs][URF_MAX_KEYPOINTS] urf_dmatch and counts [max_pairs] int, final (redone pairs included).  Two sets alternate
 * between consecutive batches: ask again after every fetch; a set is rewritten by the handle's batch after next. */
int urf_pm_device_results(urf_pm *h, const urf_dmatch **d_matches, const int **d_counts);
/* How often this handle's LDS-resident Sinkhorn launch (fast mode) gave up -- its 32 workgroups per pair did not become
 * co-resident within 0.25 s, e.g. another process holds CUs -- and the batch was redone with the streaming kernels before its
 * results were handed out (the handle then stays on the streaming kernels).  Normally 0.  Results are the same either way;
 * a caller that shipped the device lists elsewhere before fetching (the gather above) ships them again when this number moved. */
int urf_pm_sinkhorn_fallbacks(const urf_pm *h);
/* integrity check of the fast Sinkhorn (urf_sg_config.sinkhorn_residual_bound): out[0] = pairs whose result failed the bound and
 * was redone, out[1] = batches in which that happened, out[2] = the bound in use, out[3] = pairs processed (n <= 4 values). */
int urf_pm_sinkhorn_integrity(urf_pm *h, double *out, int n);
/* the largest |column marginal - 1| of the plan the decode read, for the P pairs of the batch handed out last (zeros in the exact mode) */
int urf_pm_sinkhorn_residuals(urf_pm *h, float *out, int P);
/* guarded fast mode (precision 2), counters since build(): out[0] = pairs redone in the exact mode, out[1] = pairs
 * processed, out[2] / out[3] = redone pairs by cause (threshold margin / runner-up margin), out[4] = pairs flagged
 * (n <= 8 values written) */
int urf_pm_near_tie_reruns(urf_pm *h, unsigned long long *out, int n);
/* guard words of the P pairs of the batch handed out by the last urf_pm_fetch / urf_match / urf_sg_infer of this handle:
 * 0 = the pair's match set is the exact pipeline's; bit 0 = a best assignment within the margin of the threshold,
 * bit 1 = within the margin of its runner-up */
int urf_pm_near_tie_flags(urf_pm *h, int *flags, int P);
/* the redo engine this handle uses (strict mode): out[0] = passes it has run (for all the handles that share it), out[1] = pairs
 * through them, out[2] = passes that served more than one batch, out[3] = handles sharing the engine (n <= 4 values) */
int urf_pm_redo_engine_stats(urf_pm *h, double *out, int n);
/* guarded fast mode: measure the fast matcher against the exact matcher on P pairs of device slots (as urf_match_device takes
 * them) -- the largest difference of the two log-assignment matrices over the entries a decision can rest on (probability above
 * 0.1 in either) -- and widen the handle's margin to 1.1 x (that + 2.4e-4, the share of the fast SuperPoint's descriptor noise)
 * if it is smaller (never narrows).  out[0] = the measured difference, out[1] = the margin in use after the call (out may be
 * null).  Synchronous; allocates and frees a scratch copy of the P matrices. */
int urf_pm_calibrate_guard(urf_pm *h, int P, const void *const *d_slots0, const void *const *d_slots1, double *out);
/* the guard of a precision-2 / -3 handle: out[0] = the margin in use (log domain), out[1] = the largest fast-vs-exact difference
 * the calibrations (automatic or explicit) have measured, out[2] = pairs the automatic calibration still wants to measure,
 * out[3] = 1 when the measured error exceeded the cap and a strict handle redoes every pair in the exact mode; the online check
 * (urf_sg_config.audit_period): out[4] = the largest fast-vs-exact difference any redone pair has shown, out[5] = redone pairs
 * sampled, out[6] = times the margin was raised for it, out[7] = times it exceeded the margin its batch had been guarded with,
 * out[8] = unflagged pairs audited, out[9] = audited pairs whose exact index list differed from the fast one, out[10] = batches
 * the handle ran in the exact mode because its guard flagged most pairs (n <= 11 values) */
int urf_pm_guard_state(urf_pm *h, double *out, int n);

/* ------------------------------------------------ kernel timing (bench) ---- */
/* HIP-event timing of the pipeline stages on the handle's own stream. */
int urf_sp_stage_ms(urf_sp *h, float *ms, int n);   /* ms[i]: last call's stage times */
int urf_sp_stage_ms_age(urf_sp *h, float *ms, int n, int age); /* call `age` calls ago (0..3) */
int urf_pm_stage_ms(urf_pm *h, float *ms, int n);
int urf_set_profiling(int enable);

/* ---------------------------------------------------------------- Camera -- */
/* Camera::UndistortImage as the first device stage.
 * The reference builds two CV_32FC1 maps once (Camera::Camera, src/camera.cc:69-85,
 * cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap) and calls
 * cv::remap(image, out, map1, map2, cv::INTER_LINEAR) per frame (:116-118). */
typedef struct {
  int width, height;       /* image_width / image_height (src/camera.cc:16-17) */
  int distortion_type;     /* 0 = radial-tangential, else fisheye (src/camera.cc:42,76-84) */
  double K[9];             /* LEFT_K, row-major */
  double D[14];            /* LEFT_D: k1 k2 p1 p2 [k3 [k4 k5 k6 [s1..s4]]] or fisheye k1..k4 */
  int n_dist;
  double R[9];             /* cv::Mat::eye for the monocular setups (src/camera.cc:78), LEFT_R for stereo */
  double P[9];             /* LEFT_P.rowRange(0,3).colRange(0,3) */
  int device;
} urf_cam_config;
typedef struct urf_cam urf_cam;

/* Camera::Camera's map construction (host, once). */
int urf_cam_create(const urf_cam_config *cfg, urf_cam **out);
/* Same handle from maps the caller already has (Camera::_map1/_map2, CV_32FC1, width x height):
 * the drop-in a maintainer uses, OpenCV's own maps stay the source of truth. */
int urf_cam_create_from_maps(const float *map1, const float *map2, int width, int height, int device, urf_cam **out);
void urf_cam_destroy(urf_cam *h);
int urf_cam_size(const urf_cam *h, int *width, int *height);   /* size of the maps = of the undistorted frames */
int urf_cam_maps(urf_cam *h, float *map1, float *map2);
/* Camera::UndistortImage(image, image_undistorted), src/camera.cc:116-118: host u8 in, host u8 out. */
int urf_cam_undistort(urf_cam *h, const uint8_t *img, int rows, int cols, size_t step, uint8_t *out, size_t ostep);
/* n device-resident frames (rows x cols, contiguous) -> n frames height x width, enqueued on
 * `stream` (hipStream_t; pass urf_sp_stream(sp) to run in order in front of
 * urf_sp_infer_device) or on the handle's own stream when NULL. */
int urf_cam_undistort_device(urf_cam *h, const void *d_imgs, int n, int rows, int cols, void *d_out, void *stream);
int urf_cam_sync(urf_cam *h);

/* ------------------------------------------------------------ frame stream -- */
/* The batched caller of the path: what
 * Tracking::ExtractFeatureThread / ExtractFeatureAndMatch (src/tracking.cc:123-218,
 * 338-377) do per frame -- SuperPoint, then PointMatching against the previous
 * (key)frame with outlier rejection -- for a stream of frames submitted in batches.
 * Features stay in device slots between the two stages; SuperPoint and
 * `matchers` PointMatching handles run on their own HIP streams so consecutive
 * batches overlap; only raw u8 frames go up and match lists come down. */
typedef struct {
  urf_sp_config sp;          /* max_batch is overridden by `batch` */
  urf_sg_config sg;          /* max_pairs / device are overridden (batch, sp.device) */
  int batch;                 /* frames per submit, 1..64 */
  int matchers;              /* 0 -> 2 alternating matcher handles */
  int history_batches;       /* extra batches kept resident for references older than 2 batches */
  int outlier_rejection;     /* MatchingPoints(..., outlier_rejection): src/tracking.cc:355 passes true */
} urf_fe_config;
typedef struct urf_fe urf_fe;

int urf_fe_create(const urf_fe_config *cfg, urf_fe **out);
int urf_fe_build(urf_fe *h, const float *sp_blob, size_t sp_floats, const float *sg_blob, size_t sg_floats);
int urf_fe_build_files(urf_fe *h, const char *sp_engine_file, const char *sg_engine_file);
void urf_fe_destroy(urf_fe *h);
/* Camera::UndistortImage in front of SuperPoint; `cam` is borrowed, map_rows x map_cols is its map size. */
int urf_fe_set_camera(urf_fe *h, urf_cam *cam, int map_rows, int map_cols);
/* n <= batch host frames (u8, row stride `step`, frame stride `frame_stride` bytes); asynchronous.
 * ref: NULL or n global frame indices (counted from the first submitted frame): frame j is matched
 * against frame ref[j] (-1 = its predecessor) -- the reference matches against the last keyframe
 * (src/tracking.cc:196-203).  A referenced frame must be in this batch or in one of the
 * 2 + history_batches batches before it.  At most urf_fe_max_in_flight() = min(matchers + 5, 3 matchers + 2) batches may be
 * in flight (a matcher handle holds two begun batches and one enqueued one; 7 with the default two matchers, 5 with one): a
 * submit enqueues its
 * own SuperPoint, the match call of the batch two submits back and begins the fetch of the batch `matchers` + 1 submits
 * back (the only wait, for that batch's fast pass; a strict handle's exact redo of flagged pairs then runs beside the
 * next batches) -- the loop bench.py times, driven by the caller:
 *     urf_fe_submit(b);  while (urf_fe_in_flight() >= urf_fe_max_in_flight() || urf_fe_ready() == 1) urf_fe_collect(...);
 * A caller that collects right after every submit gets the synchronous behaviour (collect enqueues what is missing). */
int urf_fe_submit(urf_fe *h, const uint8_t *frames, int n, int rows, int cols, size_t step, size_t frame_stride,
                  const long *ref);
/* Oldest batch in flight: K[j] keypoints, nmatch[j] matches at matches[j*cap ...] (queryIdx -> the
 * reference frame, trainIdx -> frame j); the stream's first frame has no reference and 0 matches.
 * feat: NULL or nframes matrices of 259 x URF_MAX_KEYPOINTS f64 (column-major). */
int urf_fe_collect(urf_fe *h, int *nframes, int *K, urf_dmatch *matches, int cap, int *nmatch, double *feat);
int urf_fe_in_flight(urf_fe *h);
/* the most batches urf_fe_submit accepts before one must be collected (see urf_fe_submit) */
int urf_fe_max_in_flight(urf_fe *h);
/* 1 when urf_fe_collect would hand out the oldest batch without waiting for the GPU, 0 when it would wait, <0 on error */
int urf_fe_ready(urf_fe *h);
/* 1 if the NEXT urf_fe_submit may name global frame `frame` in `ref` (its slot is still in the ring: one of the last
 * 2 + history_batches SUBMITS, whatever their sizes), 0 if not, <0 on error.  A caller that tracks keyframes asks here
 * instead of counting frames (batches are ragged), and matches the host features of a frame that has left the ring. */
int urf_fe_frame_resident(urf_fe *h, long frame);
urf_sp *urf_fe_superpoint(urf_fe *h);
urf_pm *urf_fe_matcher(urf_fe *h, int i);

/* ------------------------------------------------------------------ pose stage -- */
/* The per-frame pose computation that consumes the front-end's matches, batched over the
 * frames of a step: SolvePnPWithCV (src/g2o_optimization.cc:323-377) and FrameOptimization (:179-321).
 * cv::solvePnPRansac and g2o are un-vendored third-party code: the arithmetic is the written specification of
 * DESIGN.md "Pose stage" (parity unpinned), every parameter of the reference's call sites is kept. */
typedef struct urf_pose urf_pose;
/* buffers for up to max_batch frames of up to `capacity` observations each, on `device` */
int urf_pose_create(int device, int max_batch, int capacity, urf_pose **out);
void urf_pose_destroy(urf_pose *h);
typedef struct {
  double fx, fy, cx, cy;       /* Camera::GetCamerMatrix (:329-330); images are undistorted, dist_coeffs = 0 */
  int iterations;              /* 0 -> 100   (cv::solvePnPRansac(..., false, 100, 20.0, 0.99, inliers), :352-353) */
  double reprojection_error;   /* 0 -> 20.0 px */
  double confidence;           /* 0 -> 0.99 */
  uint32_t seed;
} urf_pnp_config;
/* B frames; frame f has n[f] <= cap correspondences obj[f][cap][3] (cv::Point3f) / img[f][cap][2] (cv::Point2f).
 * pose[f]: Twc, 4x4 row-major f64 (:363-367); inliers[f][cap]: 1 = cv_inliers lists the correspondence;
 * n_inliers[f] = the function's return value (0: fewer than 8 correspondences (:352) or no valid hypothesis). */
int urf_solve_pnp_ransac(urf_pose *h, const urf_pnp_config *cfg, int B, const int *n, const float *obj, const float *img,
                         int cap, double *pose, uint8_t *inliers, int *n_inliers);
typedef struct {
  double fx, fy, cx, cy;
  double chi2_threshold;       /* 0 -> 5.991 (OptimizationConfig::mono_point) */
} urf_poseopt_config;
/* FrameOptimization over the mono observations of B frames: Xw[f][cap][3] map points, obs[f][cap][2] keypoints;
 * q_wc[f][4] (w, x, y, z) / p_wc[f][3]: the prior pose Twc in, the optimised pose out; inlier[f][cap]: the
 * re-classified MonoPointConstraint::inlier flags; n_inliers[f] = the function's return value (n - outliers). */
int urf_frame_optimization(urf_pose *h, const urf_poseopt_config *cfg, int B, const int *n, const double *Xw,
                           const double *obs, int cap, double *q_wc, double *p_wc, uint8_t *inlier, int *n_inliers);

/* FrameOptimization with the stereo edges of a rectified pair as well (EdgeStereoSE3ProjectXYZOnlyPose,
 * src/g2o_optimization.cc:235-260, 291-306): frame f holds n_mono[f] mono observations followed by n_stereo[f] stereo ones in
 * the same rows -- Xw[f][cap][3], obs[f][cap][3] = (u, v, u_right), u_right ignored for the mono rows; inlier[f][cap] in that
 * order.  The third residual is u_right - (u - bf / z); Huber deltas sqrt(chi2_mono) / sqrt(chi2_stereo). */
typedef struct {
  double fx, fy, cx, cy;
  double bf;                   /* Camera::BF(): baseline times fx */
  double chi2_mono;            /* 0 -> 5.991 (OptimizationConfig::mono_point) */
  double chi2_stereo;          /* 0 -> 7.815 (OptimizationConfig::stereo_point) */
} urf_poseopt_stereo_config;
int urf_frame_optimization_stereo(urf_pose *h, const urf_poseopt_stereo_config *cfg, int B, const int *n_mono, const int *n_stereo,
                                  const double *Xw, const double *obs, int cap, double *q_wc, double *p_wc, uint8_t *inlier,
                                  int *n_inliers);

/* ------------------------------------------------ map-point projection search -- */
/* Mapping::SearchByProjection(frame, mappoints, thr, good_projections), src/mapping.cc:667-735
 * (SURVEY.md section 8 f4): projection (include/camera.h:48-68), window search
 * (Frame::FindNeighborKeypoints, src/frame.cc:320-353), descriptor distance (src/utils.cc:14-19),
 * first-best-wins minimum and the 0.35 / 0.6 acceptance tests.  Host arrays in, host array out. */
typedef struct {
  double fx, fy, cx, cy;
  double image_width, image_height;
  double pose[16];            /* Twc = Frame::GetPose(), row-major 4x4 */
  int thr;                    /* radius = 15 * thr */
  int device;
} urf_sbp_config;
/* feat: column-major 259 x K f64 (Frame::GetAllFeatures); occupied: K flags, non-zero = the keypoint
 * already has a good map point (NULL = none); mp_pos M x 3, mp_desc M x 256 f64, mp_valid M flags or
 * NULL; best_idx[m] = keypoint index of an accepted projection, else -1. */
int urf_search_by_projection(const urf_sbp_config *cfg, const double *feat, int K, const uint8_t *occupied,
                             const double *mp_pos, const double *mp_desc, const uint8_t *mp_valid, int M, int *best_idx);
/* same, features taken from a device feature slot (f32, widened exactly: what the host copy holds); the slot's
 * producer must have finished (urf_sp_sync) -- the search is not ordered against the SuperPoint stream */
int urf_search_by_projection_slot(const urf_sbp_config *cfg, const void *d_slot, int K, const uint8_t *occupied,
                                  const double *mp_pos, const double *mp_desc, const uint8_t *mp_valid, int M,
                                  int *best_idx);

/* cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, thresh (3), confidence (0.99), mask) on raw point arrays:
 * what /root/reference/src/point_matching.cc:50 calls, as the outlier stage 1 above runs it (cvransac.hip; the restatement
 * of OpenCV 4.2.0's published algorithm -- PARITY UNPINNED, no OpenCV binary in this image).  pts0 / pts1: n x 2 f32 (n <= 1024),
 * host arrays; mask: n bytes, 1 = inlier (all ones when no model is found, as the reference then filters nothing);
 * F9 (optional): the row-major model behind the mask, untouched when there is none; iterations (optional): hypothesis rounds
 * run.  Touches no handle. */
int urf_cv_find_fundamental(const float *pts0, const float *pts1, int n, double thresh, double confidence, uint8_t *mask,
                            double *F9, int *iterations, int device);

/* arithmetic taps of the GPU parity tests: pure functions of their arguments (the fp32 matrix core as an ordered fma chain, the
 * canonical exp / log, IEEE division and square root, the split-f16 GEMM, one f16 matrix-core instruction).  They touch no handle. */
int urf_probe_fma_gemm(const float *A, const float *B, const float *bias, int M, int N, int K, float *C, int device);
int urf_probe_math(const float *x, int n, float *exp_out, float *log_out, int device);
int urf_probe_divsqrt(const float *a, const float *b, int n, float *q, float *s, double *qd, double *sd, int device);
/* split-f16 GEMM (fast precision mode): Y = X W + bias, avg ms over reps */
int urf_probe_h2gemm(const float *X, const float *W, const float *bias, int M, int N, int K, float *Y,
                     int reps, float *ms_out, int device);
/* ncases independent v_mfma_f32_16x16x32_f16 (A 16x32 f16, B 32x16 f16, C/D 16x16 f32, row-major) */
int urf_probe_mfma_f16(const void *A_f16, const void *B_f16, const float *C, float *D, int ncases, int device);

#ifdef URF_EXPERIMENTS
/* Test hooks and diagnostics: exported ONLY by the experiments build (`make -C ur-mvo_amd/csrc experiments`,
 * liburf_front_exp.so) -- fault injection and kernel A/B switches have no place in the product library. */
/* force the split-f16 GEMM kernel (0 = register-staged 128x128, 1 = register-staged 64x128, 2 = LDS-DMA 128x128 = default) */
int urf_probe_h2gemm_variant(int v);
/* s_memtime stamps of the resident Sinkhorn (8 per iteration, workgroup 0); tools/gpu_sinkhorn_stamps.py */
int urf_probe_sinkhorn_stamps(int enable, int iters, long long *out);
/* the next `launches` resident Sinkhorn launches of this process report a give-up (exercises the recovery of sg_api.hip) */
int urf_probe_sinkhorn_fault(int launches);
/* matcher handles built after this call stay on the streaming Sinkhorn for `batches` batches after a give-up (0 = the default, 64) */
int urf_probe_sinkhorn_backoff(int batches);
/* the next `launches` resident Sinkhorn launches get `delta` added to one column potential of their first pair AFTER the
 * iterations (a damaged last iteration); exercises the integrity check */
int urf_probe_sinkhorn_corrupt(int launches, float delta);
/* roof probe: the split-f16 MFMA inner loop, `waves_per_cu` in {4, 8, 16}: PFLOP/s of MFMA issue and the in-kernel clock the
 * chip holds under that load.  mode 0 = register-resident operands, no memory; 1 = plus the linear-layer kernel's fragment reads
 * from LDS; 2 = plus its barrier per step; 3 = plus its LDS-DMA from L2-resident sources; 4 = activations streamed from HBM;
 * 5 = the fp32 matrix core (v_mfma_f32_16x16x4_f32, the exact mode's instruction) on register operands */
int urf_probe_mfma_roof(int device, int waves_per_cu, int iters, int mode, float *pflops, float *ghz);
/* diagnostics of the linear-layer kernel for tools/gpu_h2fixed.py: 1 = non-temporal stores, 2 = no stores, 4 = one K chunk only
 * (2 and 4 give wrong results: timing only); 0 restores the product behaviour */
int urf_probe_h2gemm_xflags(int flags);
/* A/B switches of round 6's small-grid kernels (bit-identical forms of the fast matcher's linear and attention kernels):
 * urf_probe_h2gemm_deep: 0 = never the deep-ring tile, 1 = the launcher's policy (default), 6 / 3 = that ring depth everywhere;
 * urf_probe_attn_variant: -1 = the launcher's policy, 0 .. 4 = 1x8, 2x4, 2x8, 1x4, 1x2 (query tiles per wave x waves) */
/* the next `passes` passes of a redo engine fail after they have taken their jobs out of the engine's queue -- what a failing
 * launch or copy does; the owners' urf_pm_fetch_begin / _end must report it instead of handing out un-redone lists */
int urf_probe_redo_fault(int passes);
int urf_probe_h2gemm_deep(int depth);
/* the exact linear layer with both operands by LDS-DMA (linear_dma_kernel): 0 = never (the register-staged tile), 1 / 2 = two
 * stages (default), 3 = three stages.  Bit-identical to the register-staged tile. */
int urf_probe_linear_dma(int v);
/* the exact attention kernel's workgroup: 4 = 64 queries on 512 threads, 2 = 32 queries on 256 threads, 0 = the launcher's
 * policy (2 for at most two pairs).  Same bits. */
int urf_probe_attn_exact_nqt(int v);
int urf_probe_attn_variant(int variant);
#endif

#ifdef __cplusplus
}
#endif
#endif
