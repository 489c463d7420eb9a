/* urf_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement (the parity oracle) of
 * the UR-MVO learned front-end: SuperPoint -> SuperGlue -> epipolar RANSAC.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (ur-mvo_amd/) never includes, links or calls
 * anything in oracle/.
 *
 * Every function cites the reference file:line (relative to the UR-MVO tree)
 * whose behaviour it restates.  Pinning: see oracle/README.md and DESIGN.md.
 */
#ifndef URF_ORACLE_H_
#define URF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- weight containers (same written layout as the product) ---- */
#define OSP_NUM_CONV 12
#define OSP_BLOB_FLOATS 1300865
#define OSG_BLOB_FLOATS 12003905
#define OSG_LAYERS 18

/* ---------------- SuperPoint ---------------- */
typedef struct {
  int max_keypoints;          /* include/read_configs.h:10 */
  double keypoint_threshold;  /* :11 */
  int remove_borders;         /* :12 */
} osp_config;

/* Dense network, superpoint/SP/model.py:55-86 (+ simple_nms :15-26) fed by
 * SuperPoint::process_input src/super_point.cpp:158-176.
 * img: u8 H x W with row stride `step`.  Outputs (any may be NULL):
 *   scores_nms : [Hs][Ws] f32 post-NMS heat map, Hs=(H/8)*8, Ws=(W/8)*8
 *   heat       : [Hs][Ws] f32 pre-NMS heat map (softmax + depth-to-space)
 *   desc       : [Hc][Wc][256] f32 L2-normalised dense descriptors (NHWC)
 *   layers[i]  : optional per-conv post-activation dumps (NHWC), i in [0,12)
 */
int osp_dense(const float *blob, const uint8_t *img, int H, int W, size_t step,
              float *scores_nms, float *heat, float *desc, float **layers);

/* SuperPoint::process_output src/super_point.cpp:338-386 (host post-process).
 * scores: [Hs][Ws]; desc: [Hc][Wc][256]; mask: NULL or u8 [Hs][Ws] stride mstep.
 * feat: column-major 259 x cap doubles.  Returns 0; *K = columns written.
 * kp_index (optional, cap ints): raster index y*Ws+x of each keypoint. */
int osp_postprocess(const float *scores, int Hs, int Ws, const float *desc,
                    int Hc, int Wc, const uint8_t *mask, size_t mstep,
                    const osp_config *cfg, double *feat, int cap, int *K,
                    int *kp_index);

/* SuperPoint::infer src/super_point.cpp:121-156 = dense + postprocess. */
int osp_infer(const float *blob, const osp_config *cfg, const uint8_t *img,
              int H, int W, size_t step, const uint8_t *mask, size_t mstep,
              double *feat, int cap, int *K);

/* simple_nms alone (superpoint/SP/model.py:15-26), radius 4. */
void osp_simple_nms(const float *heat, int Hs, int Ws, float *out);

/* ---------------- SuperGlue + matching ---------------- */
typedef struct {
  int image_width;            /* include/read_configs.h:21 */
  int image_height;           /* :22 */
  double matching_threshold;  /* :24 */
  int sinkhorn_iterations;    /* src/super_glue.cpp:463 (default 100) */
} osg_config;

/* PointMatching::NormalizeKeypoints src/point_matching.cc:63-76 */
void osg_normalize_keypoints(const double *feat, int n, int width, int height,
                             double *out);

/* SuperGlue graph (I/O contract src/super_glue.cpp:63-98,
 * 198-215, Sinkhorn recurrence :432-498).  f0/f1: column-major 259 x n with
 * ALREADY normalised keypoints (as SuperGlue::infer receives them).
 * Z: (n0+1) x (n1+1) row-major log-assignment (may be NULL).
 * final0/final1: optional [n][256] projected descriptors. */
int osg_graph(const float *blob, int iters, const double *f0, int n0,
              const double *f1, int n1, float *Z, float *final0, float *final1);

/* decode() src/super_glue.cpp:401-430 on a (h x w) log-assignment. */
void osg_decode(const float *Z, int h, int w, double thresh, int *idx0,
                int *idx1, double *ms0, double *ms1);

/* SuperGlue::infer src/super_glue.cpp:166-241 = graph + decode. */
int osg_infer(const float *blob, const osg_config *cfg, const double *f0,
              int n0, const double *f1, int n1, int *idx0, int *idx1,
              double *ms0, double *ms1, float *Z);

typedef struct { int queryIdx, trainIdx; float distance; } o_dmatch;

typedef struct {
  int iterations;   /* EpipolarGeometry(K, sigma, iterations): src/tracking.cc:52-55 -> 200 */
  float sigma;      /* 1.0; the reference call's 3 px gate (src/point_matching.cc:50) is sigma = 3 / sqrt(3.841) */
  uint32_t seed;    /* explicit, replaces process-global srand(0) (src/epipolar_geometry.cc:100-112) */
  float confidence; /* <= 0: every hypothesis counts (_find_F); else the sequential loop of cv::findFundamentalMat's
                     * 4th argument: hypotheses walked in order, each new best shrinks the count to the smallest k with
                     * (1 - w^8)^k <= 1 - confidence (OpenCV RANSACUpdateNumIters, 8 model points) */
  int stage;        /* 0 = the in-tree 8-point search above (default); 1 = the restatement of OpenCV 4.2's
                     * cv::findFundamentalMat(..., FM_RANSAC, 3, confidence, mask) (cvransac_oracle.c; sigma / iterations / seed unused) */
} oransac_config;

/* minimal sets: sampler 0 = the build's counter hash, 1 = the reference's stream, i.e. the C library's own
 * srand(seed) / rand() through Random::RandomInt (src/epipolar_geometry.cc:56-71,100-117). sets: iterations x 8 */
void oransac_minimal_sets(int sampler, uint32_t seed, int n, int iterations, int *sets);
/* _find_F over explicit minimal sets (the reference's _vSets), matches walked in the caller's order */
float oransac_find_F_sets(const float *pts0, const float *pts1, int n, const oransac_config *cfg, const int *sets,
                          uint8_t *inliers, float *F21);

/* EpipolarGeometry::_find_F (+_normalize,_compute_F21,_check_F)
 * src/epipolar_geometry.cc:161-205,247-283,372-449,735-780 on n matched pixel
 * pairs.  inliers: n bytes.  F21: 9 floats row-major.  Returns best score. */
float oransac_find_F(const float *pts0, const float *pts1, int n,
                     const oransac_config *cfg, uint8_t *inliers, float *F21);

/* cv::findFundamentalMat(m1, m2, cv::FM_RANSAC, thresh, confidence, mask) of OpenCV 4.2.0 restated (cvransac_oracle.c; parity
 * unpinned: no OpenCV binary to check against).  m1 / m2: n x (x, y) floats; mask: n bytes; returns the inlier count. */
int ocv_find_fundamental_mask(const float *m1, const float *m2, int n, double thresh, double confidence, uint8_t *mask);

/* PointMatching::MatchingPoints src/point_matching.cc:14-61; the
 * cv::findFundamentalMat call (:50) is replaced by oransac_find_F.
 * Returns number of matches written (<= cap). */
int omatch_points(const float *sg_blob, const osg_config *cfg,
                  const oransac_config *rcfg, const double *f0, int n0,
                  const double *f1, int n1, int outlier_rejection,
                  o_dmatch *out, int cap);

/* EpipolarGeometry::reconstruct src/epipolar_geometry.cc:18-98 (mono init).
 * keys: n x (x,y) pixel coords; matches12[n1]: index into keys2 or -1.
 * T21: 4x4 row-major; P3D: n1 x 3; tri: n1 flags; model: 0 = H, 1 = F;
 * scores[2] = {SH, SF}.  Returns 1 when the initialisation is accepted. */
typedef struct { float K[9]; float sigma; int iterations; uint32_t seed; int sampler; } oepi_config;
int oepi_reconstruct(const oepi_config *cfg, const float *keys1, int n1, const float *keys2, int n2,
                     const int *matches12, float *T21, float *P3D, uint8_t *tri, int *model, float *scores);
/* same over explicit minimal sets (cfg->iterations x 8 indices into the list of valid matches) */
int oepi_reconstruct_sets(const oepi_config *cfg, const float *keys1, int n1, const float *keys2, int n2,
                          const int *matches12, const int *sets, float *T21, float *P3D, uint8_t *tri, int *model,
                          float *scores);

/* ---------------- pose stage (pnp_oracle.c) ---------------- */
typedef struct {
  double fx, fy, cx, cy;       /* Camera::GetCamerMatrix, src/g2o_optimization.cc:329-330 */
  int iterations;              /* 100 (:352-353) */
  double reprojection_error;   /* 20.0 px */
  double confidence;           /* 0.99 */
  uint32_t seed;
} opnp_config;
/* SolvePnPWithCV src/g2o_optimization.cc:323-377.  pose: Twc 4x4 row-major; returns the inlier count */
int opnp_solve_ransac(const opnp_config *cfg, const float *obj, const float *img, int n, double *pose, uint8_t *inliers);
typedef struct {
  double fx, fy, cx, cy;
  double chi2_threshold;       /* cfg.mono_point (5.991) */
} oposeopt_config;
/* FrameOptimization src/g2o_optimization.cc:179-321 (mono edges).  q_wc (w,x,y,z) / p_wc in-out; returns n - outliers */
int oframe_optimization(const oposeopt_config *cfg, const double *Xw, const double *obs, int n, double *q_wc, double *p_wc,
                        uint8_t *inlier);

/* FrameOptimization with stereo edges as well (EdgeStereoSE3ProjectXYZOnlyPose, src/g2o_optimization.cc:235-260,291-306):
   n_mono rows (u, v, unused) followed by n_stereo rows (u, v, u_right); Xw and obs have 3 numbers per row */
typedef struct {
  double fx, fy, cx, cy, bf;
  double chi2_mono, chi2_stereo;   /* cfg.mono_point, cfg.stereo_point */
} oposeopt_stereo_config;
int oframe_optimization_stereo(const oposeopt_stereo_config *cfg, const double *Xw, const double *obs, int n_mono, int n_stereo,
                               double *q_wc, double *p_wc, uint8_t *inlier);

/* ---------------- camera ---------------- */
typedef struct {
  int width, height;       /* image_width / image_height, src/camera.cc:16-17 */
  int distortion_type;     /* 0 = radial-tangential, else fisheye (src/camera.cc:42,76-84) */
  double K[9];             /* LEFT_K */
  double D[14];            /* LEFT_D */
  int n_dist;
  double R[9];             /* identity for the monocular setups (src/camera.cc:78) */
  double P[9];             /* LEFT_P(0:3,0:3) */
} ocam_config;

/* Camera::Camera map construction, src/camera.cc:69-85 (OpenCV initUndistortRectifyMap). */
int ocam_init_maps(const ocam_config *c, float *map1, float *map2);
/* Camera::UndistortImage, src/camera.cc:116-118 (cv::remap INTER_LINEAR, constant border 0). */
void ocam_remap(const uint8_t *img, int H, int W, size_t step, const float *map1, const float *map2, int oh, int ow,
                uint8_t *out, size_t ostep);

/* ---------------- map-point projection search ---------------- */
typedef struct {
  double fx, fy, cx, cy;              /* Camera::Project, include/camera.h:48-68 */
  double image_width, image_height;
  double pose[16];                    /* Twc, row-major 4x4 (Frame::GetPose) */
  int thr;                            /* search radius = 15 * thr pixels (src/mapping.cc:680) */
} osbp_config;
/* Mapping::SearchByProjection src/mapping.cc:667-735.  feat: column-major 259 x K; occupied: K flags
 * (keypoint already carries a good map point -> skipped, src/frame.cc:342) or NULL; mp_*: M map points.
 * best_idx[m] = keypoint index of an accepted projection or -1. */
int osbp_search(const osbp_config *c, const double *feat, int K, const uint8_t *occupied, const double *mp_pos,
                const double *mp_desc, const uint8_t *mp_valid, int M, int *best_idx);

/* canonical math probes (tests) */
float o_exp(float x);
float o_log(float x);
float o_wave_sum(const float *x, int n);
/* fmaf-chain GEMM probe: C[m][n] = chain_k fma(A[m][k], B[k][n], C0[m][n]) */
void o_fma_gemm(const float *A, const float *B, const float *C0, int M, int N,
                int K, float *C);

#ifdef __cplusplus
}
#endif
#endif
