/* sg_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the SuperGlue path.
 *
 * The SuperGlue ONNX graph and its weights are NOT in the reference tree
 * (.MISSING_LARGE_BLOBS); the network restated here is the public SuperGlue
 * architecture implied by the reference's I/O contract
 * (src/super_glue.cpp:63-98,198-215), its dead-code Sinkhorn (:432-498) and the
 * keypoint normalisation (src/point_matching.cc:63-76) -- SURVEY.md App. C.
 * PARITY UNPINNED against the reference for the graph itself; structurally
 * cross-checked against transformers' SuperGlue modules (tests/golden).
 * Host-side pieces (repack, decode, match assembly) follow the cited lines.
 *
 * Tensors are token-major [n][C]; weights are [cin][cout] with BatchNorm folded
 * and attention channels head-major (c = h*64 + d) -- DESIGN.md "SG container".
 */
#include "urf_oracle.h"
#include "oracle_math.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

#define D 256
#define HEADS 4
#define DH 64

static const int kKencDims[6] = {3, 32, 64, 128, 256, 256};

typedef struct {
  const float *kw[5], *kb[5];
  struct {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wm, *bm, *w1, *b1, *w2, *b2;
  } L[OSG_LAYERS];
  const float *wf, *bf;
  float bin_score;
} sg_weights;

static void sg_parse(const float *blob, sg_weights *w) {
  const float *p = blob;
  for (int i = 0; i < 5; ++i) {
    w->kw[i] = p; p += (size_t)kKencDims[i] * kKencDims[i + 1];
    w->kb[i] = p; p += kKencDims[i + 1];
  }
  for (int l = 0; l < OSG_LAYERS; ++l) {
    w->L[l].wq = p; p += D * D; w->L[l].bq = p; p += D;
    w->L[l].wk = p; p += D * D; w->L[l].bk = p; p += D;
    w->L[l].wv = p; p += D * D; w->L[l].bv = p; p += D;
    w->L[l].wm = p; p += D * D; w->L[l].bm = p; p += D;
    w->L[l].w1 = p; p += 2 * D * 2 * D; w->L[l].b1 = p; p += 2 * D;
    w->L[l].w2 = p; p += 2 * D * D; w->L[l].b2 = p; p += D;
  }
  w->wf = p; p += D * D; w->bf = p; p += D;
  w->bin_score = *p;
}

/* Y[n][cout] = act( chain_c fma(X[n][c], W[c][cout], b[cout]) ) */
static void linear(const float *X, int n, int cin, const float *W, const float *b,
                   int cout, int relu, float *Y) {
  enum { OB = 32 };
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float *x = X + (size_t)i * cin;
    for (int o0 = 0; o0 < cout; o0 += OB) {
      const int no = cout - o0 < OB ? cout - o0 : OB;
      float acc[OB];
      for (int o = 0; o < OB; ++o) acc[o] = o < no ? b[o0 + o] : 0.0f;
      if (no == OB) {
        for (int c = 0; c < cin; ++c) {
          const float a = x[c];
          const float *wr = W + (size_t)c * cout + o0;
#pragma omp simd
          for (int o = 0; o < OB; ++o) acc[o] = __builtin_fmaf(a, wr[o], acc[o]);
        }
      } else {
        for (int c = 0; c < cin; ++c) {
          const float a = x[c];
          const float *wr = W + (size_t)c * cout + o0;
          for (int o = 0; o < no; ++o) acc[o] = __builtin_fmaf(a, wr[o], acc[o]);
        }
      }
      for (int o = 0; o < no; ++o) {
        float v = acc[o];
        if (relu) v = v > 0.0f ? v : 0.0f;
        Y[(size_t)i * cout + o0 + o] = v;
      }
    }
  }
}

/* multi-head attention message, canonical order:
 *   s_ij = (chain_d fma(q_id,k_jd,0)) * 0.125 ; m_i = max_j ; p_ij = exp_c(s_ij-m_i)
 *   keys are visited in 16-blocks; inside a block in the order 4j+r for r=0..3,
 *   j=0..3 (0,4,8,12,1,5,...): the order in which an MFMA accumulator tile is
 *   consumed as the next MFMA's operand.
 *   per key half H in {[0,512), [512,1024)}:
 *     l_H = (P0+P1)+(P2+P3), P_g = seq_{t,r} p[16t+4g+r]
 *     a_H[d] = chain_{t,r,j} fma(p[16t+4j+r], v[16t+4j+r][d], 0)
 *   o_id = (a_A[d] + a_B[d]) / (l_A + l_B)                                        */
static void attention(const float *q, int nq, const float *k, const float *v, int ns,
                      float *o) {
  const int nblk = (ns + 15) / 16;
  for (int h = 0; h < HEADS; ++h) {
    float *kt = (float *)malloc((size_t)DH * ns * sizeof(float));
    for (int j = 0; j < ns; ++j)
      for (int d = 0; d < DH; ++d) kt[(size_t)d * ns + j] = k[(size_t)j * D + h * DH + d];
#pragma omp parallel
    {
      float *s = (float *)malloc((size_t)nblk * 16 * sizeof(float));
#pragma omp for schedule(static)
      for (int i = 0; i < nq; ++i) {
        const float *qi = q + (size_t)i * D + h * DH;
        for (int j = 0; j < ns; ++j) s[j] = 0.0f;
        for (int d = 0; d < DH; ++d) {
          const float a = qi[d];
          const float *kr = kt + (size_t)d * ns;
#pragma omp simd
          for (int j = 0; j < ns; ++j) s[j] = __builtin_fmaf(a, kr[j], s[j]);
        }
        float m = -FLT_MAX;
        for (int j = 0; j < ns; ++j) { s[j] = s[j] * 0.125f; m = s[j] > m ? s[j] : m; }
        for (int j = 0; j < ns; ++j) s[j] = om_exp(s[j] - m);
        for (int j = ns; j < nblk * 16; ++j) s[j] = 0.0f;
        /* keys [0,512) and [512,1024) are reduced separately (two waves of the
           GPU kernel) and combined once: l = lA + lB, o = (accA + accB) / l */
        float lh[2], acch[2][DH];
        for (int half = 0; half < 2; ++half) {
          const int t0 = 32 * half, t1 = nblk < 32 * (half + 1) ? nblk : 32 * (half + 1);
          float P[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int t = t0; t < t1; ++t)
            for (int g = 0; g < 4; ++g)
              for (int r = 0; r < 4; ++r) P[g] = P[g] + s[16 * t + 4 * g + r];
          lh[half] = (P[0] + P[1]) + (P[2] + P[3]);
          float *acc = acch[half];
          for (int d = 0; d < DH; ++d) acc[d] = 0.0f;
          for (int t = t0; t < t1; ++t)
            for (int r = 0; r < 4; ++r)
              for (int j4 = 0; j4 < 4; ++j4) {
                const int key = 16 * t + 4 * j4 + r;
                if (key >= ns) continue;
                const float p = s[key];
                const float *vr = v + (size_t)key * D + h * DH;
#pragma omp simd
                for (int d = 0; d < DH; ++d) acc[d] = __builtin_fmaf(p, vr[d], acc[d]);
              }
        }
        const float l = lh[0] + lh[1];
        float acc[DH];
        for (int d = 0; d < DH; ++d) acc[d] = acch[0][d] + acch[1][d];
        for (int d = 0; d < DH; ++d) o[(size_t)i * D + h * DH + d] = acc[d] / l;
      }
      free(s);
    }
    free(kt);
  }
}

/* one AttentionalPropagation: delta = MLP([x ; Wm*attn(x,src)]) */
static void propagate(const sg_weights *w, int l, const float *x, int nx, const float *src,
                      int ns, float *delta) {
  float *q = (float *)malloc((size_t)nx * D * 4), *k = (float *)malloc((size_t)ns * D * 4);
  float *v = (float *)malloc((size_t)ns * D * 4), *o = (float *)malloc((size_t)nx * D * 4);
  float *msg = (float *)malloc((size_t)nx * D * 4);
  float *cat = (float *)malloc((size_t)nx * 2 * D * 4), *hid = (float *)malloc((size_t)nx * 2 * D * 4);
  linear(x, nx, D, w->L[l].wq, w->L[l].bq, D, 0, q);
  linear(src, ns, D, w->L[l].wk, w->L[l].bk, D, 0, k);
  linear(src, ns, D, w->L[l].wv, w->L[l].bv, D, 0, v);
  attention(q, nx, k, v, ns, o);
  linear(o, nx, D, w->L[l].wm, w->L[l].bm, D, 0, msg);
  for (int i = 0; i < nx; ++i) {
    memcpy(cat + (size_t)i * 2 * D, x + (size_t)i * D, D * 4);
    memcpy(cat + (size_t)i * 2 * D + D, msg + (size_t)i * D, D * 4);
  }
  linear(cat, nx, 2 * D, w->L[l].w1, w->L[l].b1, 2 * D, 1, hid);
  linear(hid, nx, 2 * D, w->L[l].w2, w->L[l].b2, D, 0, delta);
  free(q); free(k); free(v); free(o); free(msg); free(cat); free(hid);
}

/* SuperGlue::process_input src/super_glue.cpp:243-301: f64 -> f32 repack, then
   keypoint encoder + residual add. x: [n][256]. */
static void encode(const sg_weights *w, const double *f, int n, float *x) {
  float *a = (float *)malloc((size_t)n * 256 * 4), *b = (float *)malloc((size_t)n * 256 * 4);
  for (int i = 0; i < n; ++i) {
    a[i * 3 + 0] = (float)f[(size_t)259 * i + 1];
    a[i * 3 + 1] = (float)f[(size_t)259 * i + 2];
    a[i * 3 + 2] = (float)f[(size_t)259 * i + 0];
  }
  float *cur = a, *nxt = b;
  for (int l = 0; l < 5; ++l) {
    linear(cur, n, kKencDims[l], w->kw[l], w->kb[l], kKencDims[l + 1], l < 4, nxt);
    float *t = cur; cur = nxt; nxt = t;
  }
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < D; ++c)
      x[(size_t)i * D + c] = (float)f[(size_t)259 * i + 3 + c] + cur[(size_t)i * D + c];
  free(a); free(b);
}

/* LSE over a row of length n of (c[j] + add[j]); canonical wave-strided-by-4 sum */
static float row_lse(const float *c, const float *add, int n, float *tmp) {
  float m = -FLT_MAX;
  for (int j = 0; j < n; ++j) { tmp[j] = c[j] + add[j]; m = tmp[j] > m ? tmp[j] : m; }
  for (int j = 0; j < n; ++j) tmp[j] = om_exp(tmp[j] - m);
  return m + om_log(om_wave_sum4(tmp, n));
}

/* log_optimal_transport + log_sinkhorn_iterations, src/super_glue.cpp:432-498
   (max-stabilised logsumexp as in the public graph). S: n0 x n1. Z: (n0+1)x(n1+1) */
static void log_ot(const float *S, int n0, int n1, float alpha, int iters, float *Z) {
  const int R = n0 + 1, C = n1 + 1;
  float *cp = (float *)malloc((size_t)R * C * 4), *ct = (float *)malloc((size_t)R * C * 4);
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      const float v = (i == n0 || j == n1) ? alpha : S[(size_t)i * n1 + j];
      cp[(size_t)i * C + j] = v;
      ct[(size_t)j * R + i] = v;
    }
  const float norm = -om_log((float)(n0 + n1));
  float *log_mu = (float *)malloc(R * 4), *log_nu = (float *)malloc(C * 4);
  float *u = (float *)calloc(R, 4), *v = (float *)calloc(C, 4);
  for (int i = 0; i < n0; ++i) log_mu[i] = norm;
  log_mu[n0] = om_log((float)n1) + norm;
  for (int j = 0; j < n1; ++j) log_nu[j] = norm;
  log_nu[n1] = om_log((float)n0) + norm;
  for (int it = 0; it < iters; ++it) {
#pragma omp parallel
    {
      float *tmp = (float *)malloc((size_t)(R > C ? R : C) * 4);
#pragma omp for schedule(static)
      for (int i = 0; i < R; ++i) u[i] = log_mu[i] - row_lse(cp + (size_t)i * C, v, C, tmp);
#pragma omp for schedule(static)
      for (int j = 0; j < C; ++j) v[j] = log_nu[j] - row_lse(ct + (size_t)j * R, u, R, tmp);
      free(tmp);
    }
  }
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j)
      Z[(size_t)i * C + j] = ((cp[(size_t)i * C + j] + u[i]) + v[j]) - norm;
  free(cp); free(ct); free(log_mu); free(log_nu); free(u); free(v);
}

int osg_graph(const float *blob, int iters, const double *f0, int n0, const double *f1,
              int n1, float *Z, float *final0, float *final1) {
  if (n0 < 1 || n1 < 1) return -1;
  sg_weights w;
  sg_parse(blob, &w);
  float *x0 = (float *)malloc((size_t)n0 * D * 4), *x1 = (float *)malloc((size_t)n1 * D * 4);
  float *d0 = (float *)malloc((size_t)n0 * D * 4), *d1 = (float *)malloc((size_t)n1 * D * 4);
  encode(&w, f0, n0, x0);
  encode(&w, f1, n1, x1);
  for (int l = 0; l < OSG_LAYERS; ++l) {
    const int cross = l & 1; /* ['self','cross'] * 9 */
    propagate(&w, l, x0, n0, cross ? x1 : x0, cross ? n1 : n0, d0);
    propagate(&w, l, x1, n1, cross ? x0 : x1, cross ? n0 : n1, d1);
    for (size_t i = 0; i < (size_t)n0 * D; ++i) x0[i] = x0[i] + d0[i];
    for (size_t i = 0; i < (size_t)n1 * D; ++i) x1[i] = x1[i] + d1[i];
  }
  linear(x0, n0, D, w.wf, w.bf, D, 0, d0);
  linear(x1, n1, D, w.wf, w.bf, D, 0, d1);
  if (final0) memcpy(final0, d0, (size_t)n0 * D * 4);
  if (final1) memcpy(final1, d1, (size_t)n1 * D * 4);
  if (Z) {
    /* S_ij = (chain_c fma(m0_ic, m1_jc, 0)) * (1/16) */
    float *S = (float *)malloc((size_t)n0 * n1 * 4);
    float *m1t = (float *)malloc((size_t)D * n1 * 4);
    for (int j = 0; j < n1; ++j)
      for (int c = 0; c < D; ++c) m1t[(size_t)c * n1 + j] = d1[(size_t)j * D + c];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n0; ++i) {
      float *s = S + (size_t)i * n1;
      for (int j = 0; j < n1; ++j) s[j] = 0.0f;
      for (int c = 0; c < D; ++c) {
        const float a = d0[(size_t)i * D + c];
        const float *r = m1t + (size_t)c * n1;
#pragma omp simd
        for (int j = 0; j < n1; ++j) s[j] = __builtin_fmaf(a, r[j], s[j]);
      }
      for (int j = 0; j < n1; ++j) s[j] = s[j] * 0.0625f;
    }
    log_ot(S, n0, n1, w.bin_score, iters, Z);
    free(S); free(m1t);
  }
  free(x0); free(x1); free(d0); free(d1);
  return 0;
}

/* decode() src/super_glue.cpp:401-430 with helpers :303-399 */
void osg_decode(const float *Z, int h, int w, double thresh, int *idx0, int *idx1,
                double *ms0, double *ms1) {
  const int n0 = h - 1, n1 = w - 1;
  int *mi0 = (int *)malloc(sizeof(int) * (n0 + 1)), *mi1 = (int *)malloc(sizeof(int) * (n1 + 1));
  float *mv0 = (float *)malloc(4 * (n0 + 1));
  for (int i = 0; i < n0; ++i) { /* max_matrix dim==2 :316-327 */
    float mv = -FLT_MAX; int mi = 0;
    for (int j = 0; j < n1; ++j)
      if (mv < Z[(size_t)i * w + j]) { mv = Z[(size_t)i * w + j]; mi = j; }
    mv0[i] = mv; mi0[i] = mi;
  }
  for (int j = 0; j < n1; ++j) { /* dim==1 :328-341 */
    float mv = -FLT_MAX; int mi = 0;
    for (int i = 0; i < n0; ++i)
      if (mv < Z[(size_t)i * w + j]) { mv = Z[(size_t)i * w + j]; mi = i; }
    mi1[j] = mi;
  }
  int *valid0 = (int *)malloc(sizeof(int) * (n0 + 1));
  for (int i = 0; i < n0; ++i) {
    const int mutual0 = (mi1[mi0[i]] == i);                     /* equal_gather :345-354 */
    ms0[i] = mutual0 ? (double)om_exp(mv0[i]) : 0.0;            /* where_exp :356-365 */
    valid0[i] = (mutual0 && ms0[i] > thresh);                   /* and_threshold :379-388 */
    idx0[i] = valid0[i] ? mi0[i] : -1;                          /* where_negative_one :303-312 */
  }
  for (int j = 0; j < n1; ++j) {
    const int mutual1 = (mi0[mi1[j]] == j);
    ms1[j] = mutual1 ? ms0[mi1[j]] : 0.0;                       /* where_gather :367-377 */
    const int valid1 = (mutual1 && valid0[mi1[j]]);             /* and_gather :390-399 */
    idx1[j] = valid1 ? mi1[j] : -1;
  }
  free(mi0); free(mi1); free(mv0); free(valid0);
}

int osg_infer(const float *blob, const osg_config *cfg, const double *f0, int n0,
              const double *f1, int n1, int *idx0, int *idx1, double *ms0, double *ms1,
              float *Zout) {
  if (n0 < 1 || n1 < 1) return -1;
  float *Z = Zout ? Zout : (float *)malloc((size_t)(n0 + 1) * (n1 + 1) * 4);
  int rc = osg_graph(blob, cfg->sinkhorn_iterations, f0, n0, f1, n1, Z, NULL, NULL);
  if (rc == 0) osg_decode(Z, n0 + 1, n1 + 1, cfg->matching_threshold, idx0, idx1, ms0, ms1);
  if (!Zout) free(Z);
  return rc;
}

/* PointMatching::NormalizeKeypoints src/point_matching.cc:63-76 (integer w/2) */
void osg_normalize_keypoints(const double *feat, int n, int width, int height, double *out) {
  memcpy(out, feat, (size_t)259 * n * sizeof(double));
  const int mx = width > height ? width : height;
  for (int c = 0; c < n; ++c) {
    out[(size_t)259 * c + 1] = (feat[(size_t)259 * c + 1] - width / 2) / (mx * 0.7);
    out[(size_t)259 * c + 2] = (feat[(size_t)259 * c + 2] - height / 2) / (mx * 0.7);
  }
}

/* PointMatching::MatchingPoints src/point_matching.cc:14-61 */
int omatch_points(const float *sg_blob, const osg_config *cfg, const oransac_config *rcfg,
                  const double *f0, int n0, const double *f1, int n1, int outlier_rejection,
                  o_dmatch *out, int cap) {
  if (n0 < 1 || n1 < 1) return 0;
  double *nf0 = (double *)malloc((size_t)259 * n0 * 8), *nf1 = (double *)malloc((size_t)259 * n1 * 8);
  osg_normalize_keypoints(f0, n0, cfg->image_width, cfg->image_height, nf0);
  osg_normalize_keypoints(f1, n1, cfg->image_width, cfg->image_height, nf1);
  int *i0 = (int *)malloc(4 * n0), *i1 = (int *)malloc(4 * n1);
  double *m0 = (double *)malloc(8 * n0), *m1 = (double *)malloc(8 * n1);
  int nm = 0;
  if (osg_infer(sg_blob, cfg, nf0, n0, nf1, n1, i0, i1, m0, m1, NULL) == 0) {
    float *p0 = (float *)malloc(8 * (size_t)n0), *p1 = (float *)malloc(8 * (size_t)n0);
    for (int i = 0; i < n0 && nm < cap; ++i) {
      if (i0[i] < n1 && i0[i] >= 0 && i1[i0[i]] == i) { /* :33-35 */
        const double d = 1.0 - (m0[i] + m1[i0[i]]) / 2.0;
        out[nm].queryIdx = i; out[nm].trainIdx = i0[i]; out[nm].distance = (float)d;
        p0[2 * nm] = (float)f0[(size_t)259 * i + 1];     p0[2 * nm + 1] = (float)f0[(size_t)259 * i + 2];
        p1[2 * nm] = (float)f1[(size_t)259 * i0[i] + 1]; p1[2 * nm + 1] = (float)f1[(size_t)259 * i0[i] + 2];
        ++nm;
      }
    }
    /* :48-58, cv::findFundamentalMat replaced by the in-tree 8-point RANSAC.
       Fewer than 8 matches cannot seed a hypothesis: all are kept. */
    if (outlier_rejection && rcfg->stage == 1) {
      /* the call the reference makes (:50), restated from OpenCV 4.2 (cvransac_oracle.c) */
      uint8_t *inl = (uint8_t *)malloc((size_t)(nm > 0 ? nm : 1));
      ocv_find_fundamental_mask(p0, p1, nm, 3.0, rcfg->confidence > 0 ? (double)rcfg->confidence : 0.99, inl);
      int j = 0;
      for (int i = 0; i < nm; ++i)
        if (inl[i]) out[j++] = out[i];
      nm = j;
      free(inl);
    } else if (outlier_rejection && nm >= 8) {
      uint8_t *inl = (uint8_t *)malloc(nm);
      float F[9];
      oransac_find_F(p0, p1, nm, rcfg, inl, F);
      int j = 0;
      for (int i = 0; i < nm; ++i)
        if (inl[i]) out[j++] = out[i];
      nm = j;
      free(inl);
    }
    free(p0); free(p1);
  }
  free(nf0); free(nf1); free(i0); free(i1); free(m0); free(m1);
  return nm;
}
