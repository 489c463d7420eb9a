/* sp_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the SuperPoint path.
 * Follows superpoint/SP/model.py:15-26,55-86 (network + in-graph NMS) and
 * src/super_point.cpp:158-386 (input conversion + host post-processing) of the
 * UR-MVO reference.  Arithmetic order is the canonical order of DESIGN.md:
 * every convolution output is ONE fp32 fma chain  acc=bias; for 64-channel
 * chunk; for tap(ky,kx); for c in chunk: acc=fma(in,w,acc)  (zero padding feeds
 * exact zeros).
 */
#include "urf_oracle.h"
#include "oracle_math.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int cin, cout, k, pool, relu, src; } conv_desc;
/* src = index of the conv whose (pooled) output feeds this one; -1 = image */
static const conv_desc kConv[OSP_NUM_CONV] = {
    {1, 64, 3, 0, 1, -1},   /* conv1a model.py:38 */
    {64, 64, 3, 1, 1, 0},   /* conv1b + pool :39,60 */
    {64, 64, 3, 0, 1, 1},   /* conv2a */
    {64, 64, 3, 1, 1, 2},   /* conv2b + pool */
    {64, 128, 3, 0, 1, 3},  /* conv3a */
    {128, 128, 3, 1, 1, 4}, /* conv3b + pool */
    {128, 128, 3, 0, 1, 5}, /* conv4a */
    {128, 128, 3, 0, 1, 6}, /* conv4b */
    {128, 256, 3, 0, 1, 7}, /* convPa :48 */
    {256, 65, 1, 0, 0, 8},  /* convPb :49 */
    {128, 256, 3, 0, 1, 7}, /* convDa :52 */
    {256, 256, 1, 0, 0, 10} /* convDb :53 */
};

static size_t conv_w_floats(int i) {
  return (size_t)kConv[i].k * kConv[i].k * kConv[i].cin * kConv[i].cout;
}
static const float *conv_w(const float *blob, int i) {
  size_t off = 0;
  for (int j = 0; j < i; ++j) off += conv_w_floats(j) + kConv[j].cout;
  return blob + off;
}

/* out[y][x][o] = act( chain ), NHWC, zero padding k/2. */
static void conv_nhwc(const float *in, int H, int W, int Cin, const float *w,
                      const float *b, int Cout, int k, int relu, float *out) {
  const int pad = k / 2;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  float *pin = (float *)in;
  if (pad) {
    pin = (float *)calloc((size_t)Hp * Wp * Cin, sizeof(float));
    for (int y = 0; y < H; ++y)
      memcpy(pin + ((size_t)(y + pad) * Wp + pad) * Cin, in + (size_t)y * W * Cin,
             (size_t)W * Cin * sizeof(float));
  }
  enum { OB = 16, PB = 4 };
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    for (int x0 = 0; x0 < W; x0 += PB) {
      const int np = (W - x0) < PB ? (W - x0) : PB;
      for (int o0 = 0; o0 < Cout; o0 += OB) {
        const int no = (Cout - o0) < OB ? (Cout - o0) : OB;
        float acc[PB][OB];
        for (int p = 0; p < PB; ++p)
          for (int o = 0; o < OB; ++o) acc[p][o] = (o < no) ? b[o0 + o] : 0.0f;
        /* canonical order: 64-channel chunk -> tap -> channel */
        for (int c0 = 0; c0 < Cin; c0 += 64) {
          const int c1 = (c0 + 64 < Cin) ? c0 + 64 : Cin;
          if (no == OB && np == PB) {
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx) {
                const float *wp = w + (size_t)(ky * k + kx) * Cin * Cout + o0;
                const float *ip = pin + ((size_t)(y + ky) * Wp + (x0 + kx)) * Cin;
                for (int c = c0; c < c1; ++c) {
                  const float *wr = wp + (size_t)c * Cout;
                  const float a0 = ip[c], a1 = ip[Cin + c], a2 = ip[2 * Cin + c],
                              a3 = ip[3 * Cin + c];
#pragma omp simd
                  for (int o = 0; o < OB; ++o) {
                    const float wv = wr[o];
                    acc[0][o] = __builtin_fmaf(a0, wv, acc[0][o]);
                    acc[1][o] = __builtin_fmaf(a1, wv, acc[1][o]);
                    acc[2][o] = __builtin_fmaf(a2, wv, acc[2][o]);
                    acc[3][o] = __builtin_fmaf(a3, wv, acc[3][o]);
                  }
                }
              }
          } else {
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx) {
                const float *wp = w + (size_t)(ky * k + kx) * Cin * Cout + o0;
                for (int p = 0; p < np; ++p) {
                  const float *ip = pin + ((size_t)(y + ky) * Wp + (x0 + p + kx)) * Cin;
                  for (int c = c0; c < c1; ++c) {
                    const float a = ip[c];
                    const float *wr = wp + (size_t)c * Cout;
                    for (int o = 0; o < no; ++o)
                      acc[p][o] = __builtin_fmaf(a, wr[o], acc[p][o]);
                  }
                }
              }
          }
        }
        for (int p = 0; p < np; ++p) {
          float *op = out + ((size_t)y * W + x0 + p) * Cout + o0;
          for (int o = 0; o < no; ++o) {
            float v = acc[p][o];
            if (relu) v = v > 0.0f ? v : 0.0f;
            op[o] = v;
          }
        }
      }
    }
  }
  if (pad) free(pin);
}

/* nn.MaxPool2d(2,2) model.py:34 (floor). NHWC. */
static void pool2_nhwc(const float *in, int H, int W, int C, float *out) {
  const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < Ho; ++y)
    for (int x = 0; x < Wo; ++x)
      for (int c = 0; c < C; ++c) {
        const float a = in[((size_t)(2 * y) * W + 2 * x) * C + c];
        const float bq = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
        const float cq = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c];
        const float d = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
        const float m0 = a > bq ? a : bq, m1 = cq > d ? cq : d;
        out[((size_t)y * Wo + x) * C + c] = m0 > m1 ? m0 : m1;
      }
}

/* 9x9 stride-1 max pool with implicit -inf padding (MaxPool, model.py:5-12),
 * separable: max is order independent so this is exact. */
static void maxpool9(const float *in, int H, int W, float *out) {
  float *tmp = (float *)malloc((size_t)H * W * sizeof(float));
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int x0 = x - 4 < 0 ? 0 : x - 4, x1 = x + 4 >= W ? W - 1 : x + 4;
      float m = in[(size_t)y * W + x0];
      for (int xx = x0 + 1; xx <= x1; ++xx) {
        float v = in[(size_t)y * W + xx];
        m = v > m ? v : m;
      }
      tmp[(size_t)y * W + x] = m;
    }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    int y0 = y - 4 < 0 ? 0 : y - 4, y1 = y + 4 >= H ? H - 1 : y + 4;
    for (int x = 0; x < W; ++x) {
      float m = tmp[(size_t)y0 * W + x];
      for (int yy = y0 + 1; yy <= y1; ++yy) {
        float v = tmp[(size_t)yy * W + x];
        m = v > m ? v : m;
      }
      out[(size_t)y * W + x] = m;
    }
  }
  free(tmp);
}

/* simple_nms(scores, 4): model.py:15-26 */
void osp_simple_nms(const float *s, int H, int W, float *out) {
  const size_t n = (size_t)H * W;
  float *mp = (float *)malloc(n * 4), *mask = (float *)malloc(n * 4);
  float *supp = (float *)malloc(n * 4), *ss = (float *)malloc(n * 4);
  maxpool9(s, H, W, mp);
  for (size_t i = 0; i < n; ++i) mask[i] = (s[i] == mp[i]) ? 1.0f : 0.0f; /* :20 */
  for (int it = 0; it < 2; ++it) {                                          /* :21 */
    maxpool9(mask, H, W, mp);
    for (size_t i = 0; i < n; ++i) supp[i] = mp[i] > 0.0f ? 1.0f : 0.0f;    /* :22 */
    for (size_t i = 0; i < n; ++i) ss[i] = supp[i] != 0.0f ? 0.0f : s[i];   /* :23 */
    maxpool9(ss, H, W, mp);
    for (size_t i = 0; i < n; ++i) {                                        /* :24-25 */
      int newmax = (ss[i] == mp[i]);
      if (newmax && supp[i] == 0.0f) mask[i] = 1.0f;
    }
  }
  for (size_t i = 0; i < n; ++i) out[i] = mask[i] != 0.0f ? s[i] : 0.0f;    /* :26 */
  free(mp); free(mask); free(supp); free(ss);
}

int osp_dense(const float *blob, const uint8_t *img, int H, int W, size_t step,
              float *scores_nms, float *heat_out, float *desc_out, float **layers) {
  if (H < 8 || W < 8) return -1;
  float *act[OSP_NUM_CONV] = {0};
  int aH[OSP_NUM_CONV], aW[OSP_NUM_CONV];
  /* SuperPoint::process_input src/super_point.cpp:169-174: float(u8)/255.0
     (double division, narrowed on store). */
  float *in0 = (float *)malloc((size_t)H * W * sizeof(float));
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      in0[(size_t)y * W + x] = (float)((double)(float)img[(size_t)y * step + x] / 255.0);

  for (int i = 0; i < OSP_NUM_CONV; ++i) {
    const conv_desc *d = &kConv[i];
    const float *src = d->src < 0 ? in0 : act[d->src];
    const int sh = d->src < 0 ? H : aH[d->src], sw = d->src < 0 ? W : aW[d->src];
    const float *w = conv_w(blob, i);
    const float *b = w + conv_w_floats(i);
    float *o = (float *)malloc((size_t)sh * sw * d->cout * sizeof(float));
    conv_nhwc(src, sh, sw, d->cin, w, b, d->cout, d->k, d->relu, o);
    if (d->pool) {
      float *p = (float *)malloc((size_t)(sh / 2) * (sw / 2) * d->cout * sizeof(float));
      pool2_nhwc(o, sh, sw, d->cout, p);
      free(o);
      o = p;
      aH[i] = sh / 2; aW[i] = sw / 2;
    } else {
      aH[i] = sh; aW[i] = sw;
    }
    act[i] = o;
    if (layers && layers[i])
      memcpy(layers[i], o, (size_t)aH[i] * aW[i] * d->cout * sizeof(float));
  }
  const int Hc = aH[9], Wc = aW[9], Hs = Hc * 8, Ws = Wc * 8;

  /* softmax over 65, drop dustbin, depth-to-space: model.py:73-76.
     canonical: m=max; e_k=exp_c(l_k-m); sum sequential k=0..64; p_k=e_k/sum */
  if (scores_nms || heat_out) {
    float *heat = (float *)malloc((size_t)Hs * Ws * sizeof(float));
    const float *lg = act[9];
#pragma omp parallel for schedule(static)
    for (int cell = 0; cell < Hc * Wc; ++cell) {
      const float *l = lg + (size_t)cell * 65;
      float m = l[0];
      for (int k = 1; k < 65; ++k) m = l[k] > m ? l[k] : m;
      float e[65], sum = 0.0f;
      for (int k = 0; k < 65; ++k) { e[k] = om_exp(l[k] - m); sum = sum + e[k]; }
      const int hc = cell / Wc, wc = cell % Wc;
      for (int k = 0; k < 64; ++k)
        heat[(size_t)(hc * 8 + k / 8) * Ws + wc * 8 + (k % 8)] = e[k] / sum;
    }
    if (heat_out) memcpy(heat_out, heat, (size_t)Hs * Ws * sizeof(float));
    if (scores_nms) osp_simple_nms(heat, Hs, Ws, scores_nms);
    free(heat);
  }
  /* F.normalize(p=2, dim=1) model.py:83: x / max(||x||, 1e-12).
     canonical sum of squares: lane l owns channels 4l..4l+3 (fma chain from
     x0*x0), then 64-lane butterfly. */
  if (desc_out) {
    const float *dd = act[11];
#pragma omp parallel for schedule(static)
    for (int cell = 0; cell < Hc * Wc; ++cell) {
      const float *v = dd + (size_t)cell * 256;
      float p[64];
      for (int l = 0; l < 64; ++l) {
        float a = v[4 * l] * v[4 * l];
        a = om_fma(v[4 * l + 1], v[4 * l + 1], a);
        a = om_fma(v[4 * l + 2], v[4 * l + 2], a);
        a = om_fma(v[4 * l + 3], v[4 * l + 3], a);
        p[l] = a;
      }
      float nrm = sqrtf(om_bfly64_sum(p));
      nrm = nrm > 1e-12f ? nrm : 1e-12f;
      for (int c = 0; c < 256; ++c) desc_out[(size_t)cell * 256 + c] = v[c] / nrm;
    }
  }
  for (int i = 0; i < OSP_NUM_CONV; ++i) free(act[i]);
  free(in0);
  return 0;
}

static int clipi(int v, int mx) { return v < 0 ? 0 : (v < mx - 1 ? v : mx - 1); } /* :267-271 */

typedef struct { float s; int idx; } cand_t;
/* order defined by the build: score descending, raster index ascending on ties
   (std::sort in sort_indexes src/super_point.cpp:230-236 leaves ties open) */
static int cand_cmp(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int osp_postprocess(const float *scores, int Hs, int Ws, const float *desc,
                    int Hc, int Wc, const uint8_t *mask, size_t mstep,
                    const osp_config *cfg, double *feat, int cap, int *K,
                    int *kp_index) {
  const size_t n = (size_t)Hs * Ws;
  cand_t *cand = (cand_t *)malloc(n * sizeof(cand_t));
  int nc = 0;
  const int border = cfg->remove_borders;
  for (size_t i = 0; i < n; ++i) {
    /* find_high_score_index :196-208: float promoted to double, strict > */
    if (!((double)scores[i] > cfg->keypoint_threshold)) continue;
    const int r = (int)(i / Ws), c = (int)(i % Ws);
    if (!mask) { /* remove_borders :210-228 */
      if (!(r >= border && r < Hs - border && c >= border && c < Ws - border)) continue;
    } else {     /* filter_points :178-194 (no border test) */
      if (mask[(size_t)r * mstep + c] == 0) continue;
    }
    cand[nc].s = scores[i];
    cand[nc].idx = (int)i;
    ++nc;
  }
  /* top_k_keypoints :238-251 */
  const int k = cfg->max_keypoints;
  if (k != -1 && k < nc) {
    qsort(cand, nc, sizeof(cand_t), cand_cmp);
    nc = k;
  }
  if (nc > cap) { free(cand); return -2; }
  *K = nc;
  /* sample_descriptors :328-336 = normalize_keypoints :253-265 + grid_sample
     :273-313 + normalize_descriptors :315-326 ; all in double like the reference */
  const int s = 8, h = Hc, w = Wc;
  for (int j = 0; j < nc; ++j) {
    const int kx = cand[j].idx % Ws, ky = cand[j].idx / Ws; /* (x=col, y=row) swap :219 */
    double *col = feat + (size_t)259 * j;
    col[0] = (double)cand[j].s;
    col[1] = (double)kx;
    col[2] = (double)ky;
    if (kp_index) kp_index[j] = cand[j].idx;
    double g0 = kx - s / 2 + 0.5, g1 = ky - s / 2 + 0.5;
    g0 = g0 / (w * s - s / 2 - 0.5);
    g1 = g1 / (h * s - s / 2 - 0.5);
    g0 = g0 * 2 - 1;
    g1 = g1 * 2 - 1;
    const double ix = ((g0 + 1) / 2) * (w - 1);
    const double iy = ((g1 + 1) / 2) * (h - 1);
    const int ix_nw = clipi((int)floor(ix), w), iy_nw = clipi((int)floor(iy), h);
    const int ix_ne = clipi(ix_nw + 1, w), iy_ne = clipi(iy_nw, h);
    const int ix_sw = clipi(ix_nw, w), iy_sw = clipi(iy_nw + 1, h);
    const int ix_se = clipi(ix_nw + 1, w), iy_se = clipi(iy_nw + 1, h);
    const double nw = (ix_se - ix) * (iy_se - iy);
    const double ne = (ix - ix_sw) * (iy_sw - iy);
    const double sw = (ix_ne - ix) * (iy - iy_ne);
    const double se = (ix - ix_nw) * (iy - iy_nw);
    const float *pnw = desc + ((size_t)iy_nw * w + ix_nw) * 256;
    const float *pne = desc + ((size_t)iy_ne * w + ix_ne) * 256;
    const float *psw = desc + ((size_t)iy_sw * w + ix_sw) * 256;
    const float *pse = desc + ((size_t)iy_se * w + ix_se) * 256;
    double ssq = 0.0;
    for (int c = 0; c < 256; ++c) {
      double v = pnw[c] * nw + pne[c] * ne + psw[c] * sw + pse[c] * se;
      col[3 + c] = v;
    }
    for (int c = 0; c < 256; ++c) ssq = ssq + col[3 + c] * col[3 + c]; /* inner_product :316 */
    const double norm_inv = 1.0 / sqrt(ssq);
    for (int c = 0; c < 256; ++c) col[3 + c] = col[3 + c] * norm_inv;
  }
  free(cand);
  return 0;
}

int osp_infer(const float *blob, const osp_config *cfg, const uint8_t *img,
              int H, int W, size_t step, const uint8_t *mask, size_t mstep,
              double *feat, int cap, int *K) {
  const int Hc = H / 8, Wc = W / 8, Hs = Hc * 8, Ws = Wc * 8;
  float *sc = (float *)malloc((size_t)Hs * Ws * 4);
  float *ds = (float *)malloc((size_t)Hc * Wc * 256 * 4);
  int rc = osp_dense(blob, img, H, W, step, sc, NULL, ds, NULL);
  if (rc == 0) rc = osp_postprocess(sc, Hs, Ws, ds, Hc, Wc, mask, mstep, cfg, feat, cap, K, NULL);
  free(sc); free(ds);
  return rc;
}

float o_exp(float x) { return om_exp(x); }
float o_log(float x) { return om_log(x); }
float o_wave_sum(const float *x, int n) { return om_wave_sum(x, n); }
void o_fma_gemm(const float *A, const float *B, const float *C0, int M, int N,
                int K, float *C) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = C0 ? C0[(size_t)m * N + n] : 0.0f;
      for (int k = 0; k < K; ++k) acc = om_fma(A[(size_t)m * K + k], B[(size_t)k * N + n], acc);
      C[(size_t)m * N + n] = acc;
    }
}
