// sp_kernels.hip -- SuperPoint tail: softmax + depth-to-space, the in-graph
// simple_nms, threshold / border / top-k selection, descriptor normalisation and
// bilinear sampling.  Replaces the tail of the SuperPoint ONNX graph
// (superpoint/SP/model.py:15-26,73-84) and SuperPoint::process_output
// (src/super_point.cpp:178-386).  HBM/LDS-bound VALU kernels: coalesced loads,
// LDS tiles with halo for the 9x9 max filters, wave reductions.
#include "urf_common.h"
#include "urf_math.h"

#include <float.h>

namespace urf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------- near-tie guard
// Guarded fast mode (DESIGN.md "Guarded fast mode"): the fast precision mode computes every score within
//   err(s) = delta * s * (1 - s) + ulps * ulp(s)
// of the exact mode's (softmax of logits that carry ~1e-6 of rounding noise: the error of s = 1 / (1 + R) is s (1 - s)
// times the relative error of R; constants measured by tools/gpu_margins.py, with a safety factor).  Every discrete
// decision of the SuperPoint tail that two scores closer than err(a) + err(b) could flip sets a bit in flags[frame]; the
// frame is then redone in the exact mode by kernels that are already enqueued behind the fast ones and exit at once
// when their frame is not in the redo list (`gate`: gate[0] = frames to redo, gate[1 + r] = frame index of redo slot r).
// (struct SpGuard: urf_common.h)
__device__ __forceinline__ float guard_err(float s, float delta, float ulps) {
  const float a = s < 0.0f ? -s : s;
  const float one_m = a < 1.0f ? 1.0f - a : 0.0f;
  return delta * a * one_m + ulps * 1.1920929e-7f * a;
}
__device__ __forceinline__ bool guard_near(float a, float b, const SpGuard &g) {
  const float d = a > b ? a - b : b - a;
  return d <= guard_err(a, g.delta, g.ulps) + guard_err(b, g.delta, g.ulps);
}
#define URF_GATE(idx) if (gate && (int)(idx) >= gate[0]) return
// the exact tail (NMS, selection, descriptors) only runs for slots whose whole frame is redone
#define URF_GATE_FULL(idx) if (gate && ((int)(idx) >= gate[0] || gate[kGateMode + (int)(idx)] >= 0)) return

// ------------------------------------------------------------------ softmax
// logits [B][Hc*Wc][ld] (65 used) -> heat [B][Hs][Ws].  One lane per cell:
// m = max; e_k = exp_c(l_k - m); sum sequential k=0..64; p_k = e_k / sum.
__global__ void __launch_bounds__(256) softmax_d2s_kernel(const float *logits, int ld, int Hc, int Wc,
                                                          float *heat, const int *gate) {
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  URF_GATE(b);
  if (cell >= Hc * Wc) return;
  if (gate && gate[kGateMode + b] >= 0) {   // a slot that only needs the scores of its target cells
    bool needed = false;
    for (int t = 0; t < gate[kGateMode + b]; ++t) needed = needed || gate[kGateTargets + kAmbMax * b + t] == cell;
    if (!needed) return;
  }
  const float *l = logits + ((size_t)b * Hc * Wc + cell) * ld;
  float v[65];
  float m = l[0];
  v[0] = m;
#pragma unroll
  for (int k = 1; k < 65; ++k) { v[k] = l[k]; m = v[k] > m ? v[k] : m; }
  float sum = 0.0f;
#pragma unroll
  for (int k = 0; k < 65; ++k) { v[k] = exp_c(v[k] - m); sum = sum + v[k]; }
  const int hc = cell / Wc, wc = cell % Wc, Ws = Wc * 8;
  float *hp = heat + (size_t)b * Hc * 8 * Ws + (size_t)(hc * 8) * Ws + wc * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    f32x4 a, c;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[r * 8 + j] / sum; c[j] = v[r * 8 + 4 + j] / sum; }
    *(f32x4 *)(hp + (size_t)r * Ws) = a;
    *(f32x4 *)(hp + (size_t)r * Ws + 4) = c;
  }
}

// ---------------------------------------------------------------------- NMS
// simple_nms (model.py:15-26) as five 9x9 max-filter passes.  Block = 32x64
// pixel tile; LDS holds the tile with a 4-pixel halo (implicit -inf padding).
//  MODE 0 (A): mask = (s == mp(s))
//  MODE 1 (B): supp = mp(mask) > 0 ; ss = supp ? 0 : s
//  MODE 2 (C): mask |= (ss == mp(ss)) & ~supp ; if LAST: out = mask ? s : 0
constexpr int NR = 32, NC = 64;
template <int MODE, bool LAST>
__global__ void __launch_bounds__(256) nms_pass_kernel(const float *s, uint8_t *mask, uint8_t *supp, float *ss,
                                                       float *out, int H, int W, const int *gate, SpGuard g,
                                                       float thr_lo) {
  __shared__ float tin[(NR + 8)][(NC + 8)];
  __shared__ float hm[(NR + 8)][NC];
  const int b = blockIdx.z;
  URF_GATE_FULL(b);
  float near_tie = 0.0f;     // the highest window maximum that some pixel came within the error of
  const size_t boff = (size_t)b * H * W;
  const int y0 = blockIdx.y * NR, x0 = blockIdx.x * NC;
  const int tid = threadIdx.x;
  for (int i = tid; i < (NR + 8) * (NC + 8); i += 256) {
    const int r = i / (NC + 8), c = i % (NC + 8);
    const int y = y0 - 4 + r, x = x0 - 4 + c;
    float v = -FLT_MAX;  // stands for -inf padding; every real value is > -FLT_MAX
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const size_t p = boff + (size_t)y * W + x;
      if (MODE == 0) v = s[p];
      else if (MODE == 1) v = mask[p] ? 1.0f : 0.0f;
      else v = ss[p];
    }
    tin[r][c] = v;
  }
  __syncthreads();
  for (int i = tid; i < (NR + 8) * NC; i += 256) {
    const int r = i / NC, c = i % NC;
    float m = tin[r][c];
#pragma unroll
    for (int d = 1; d < 9; ++d) m = fmaxf(m, tin[r][c + d]);
    hm[r][c] = m;
  }
  __syncthreads();
  for (int i = tid; i < NR * NC; i += 256) {
    const int r = i / NC, c = i % NC;
    const int y = y0 + r, x = x0 + c;
    if (y >= H || x >= W) continue;
    float m = hm[r][c];
#pragma unroll
    for (int d = 1; d < 9; ++d) m = fmaxf(m, hm[r + d][c]);
    const size_t p = boff + (size_t)y * W + x;
    const float center = tin[r + 4][c + 4];
    // guard: a pixel that loses against its window's maximum by less than the two scores' error could win in the exact
    // mode (only where the maximum can become a keypoint at all)
    if (MODE != 1 && g.flags && center != m && m > thr_lo && guard_near(center, m, g)) near_tie = fmaxf(near_tie, m);
    if (MODE == 0) {
      mask[p] = (center == m) ? 1 : 0;
    } else if (MODE == 1) {
      const bool sp = m > 0.0f;
      supp[p] = sp ? 1 : 0;
      ss[p] = sp ? 0.0f : s[p];
    } else {
      const bool nm = (center == m) && (supp[p] == 0);
      const uint8_t mk = (mask[p] != 0 || nm) ? 1 : 0;
      if (LAST) out[p] = mk ? s[p] : 0.0f;
      else mask[p] = mk;
    }
  }
  // not a flag yet: a near-tie only matters if its maximum can reach the final keypoint set, which the top-k selection decides
  // once it knows the cut (topk_kernel); positive floats order like their bit patterns
  if (MODE != 1 && g.flags && near_tie > 0.0f) atomicMax(&g.nms_hi[b], __float_as_int(near_tie));
}

// guard: two surviving pixels within 4 px of each other (Chebyshev) can only be an exact tie of the fast scores (each is the
// maximum of a window that holds the other); the exact mode may separate them.  Sparse: only survivors look around.
__global__ void __launch_bounds__(256) nms_tie_kernel(const float *out, int H, int W, float thr_lo, SpGuard g) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const float *o = out + (size_t)b * H * W;
  const float c = o[i];
  if (!(c > thr_lo)) return;
  const int y = i / W, x = i % W;
  // (no short circuit: the 80 reads of a survivor are independent and go out together)
  bool tie = false;
#pragma unroll
  for (int dy = -4; dy <= 4; ++dy) {
    const int yy = y + dy, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
    for (int dx = -4; dx <= 4; ++dx) {
      const int xx = x + dx, xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
      const bool other = yy == yc && xx == xc && !(dx == 0 && dy == 0);
      tie = tie | (other & (o[(size_t)yc * W + xc] > 0.0f));
    }
  }
  if (tie) atomicMax(&g.nms_hi[b], __float_as_int(c));
}

// ---------------------------------------------------------------- selection
// find_high_score_index + remove_borders / filter_points
// (src/super_point.cpp:178-228), ordered (raster) compaction in two passes.
constexpr int SEL_PIX = 2048;  // pixels per block, 8 consecutive per thread

__device__ __forceinline__ bool sel_flag(float sc, int y, int x, int H, int W, double thr, int border,
                                          const uint8_t *mask, size_t moff) {
  if (!((double)sc > thr)) return false;
  if (mask) return mask[moff + (size_t)y * W + x] != 0;
  return y >= border && y < H - border && x >= border && x < W - border;
}

// block-wide exclusive scan of one int per thread (blockDim = 256 or 1024)
__device__ __forceinline__ int block_excl_scan(int v, int *wsum, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const int t = wsum[w];
    if (w < wave) base += t;
    tot += t;
  }
  total = tot;
  return base + incl - v;
}

__global__ void __launch_bounds__(256) sel_count_kernel(const float *scores, int H, int W, double thr, int border,
                                                        const uint8_t *mask, int *counts, int nchunk, const int *gate,
                                                        SpGuard g) {
  __shared__ int wsum[16];
  const int b = blockIdx.y, chunk = blockIdx.x;
  URF_GATE_FULL(b);
  const size_t boff = (size_t)b * H * W;
  const int base = chunk * SEL_PIX + threadIdx.x * 8;
  int c = 0;
  bool in_band = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = base + j;
    if (i < H * W) {
      const float sc = scores[boff + i];
      c += sel_flag(sc, i / W, i % W, H, W, thr, border, mask, boff) ? 1 : 0;
      // guard: a surviving pixel (scores are 0 where the NMS suppressed) whose score is within its error of the threshold,
      // on either side, inside the border / mask
      if (g.flags && sc > 0.0f && guard_near(sc, (float)thr, g) && sel_flag(sc, i / W, i % W, H, W, -1.0, border, mask, boff))
        in_band = true;
    }
  }
  if (g.flags && in_band) atomicOr(&g.band[b], 1);
  int total;
  block_excl_scan(c, wsum, total);
  if (threadIdx.x == 0) counts[b * nchunk + chunk] = total;
}

__global__ void __launch_bounds__(256) sel_scatter_kernel(const float *scores, int H, int W, double thr, int border,
                                                          const uint8_t *mask, const int *counts, int nchunk,
                                                          float *cand_score, int *cand_idx, int cand_cap,
                                                          int *cand_n, const int *gate) {
  __shared__ int wsum[16];
  __shared__ int s_off;
  const int b = blockIdx.y, chunk = blockIdx.x;
  URF_GATE_FULL(b);
  const size_t boff = (size_t)b * H * W;
  // offset of this chunk = sum of the counts of earlier chunks (integers: exact)
  int part = 0;
  for (int i = threadIdx.x; i < nchunk; i += 256)
    if (i < chunk) part += counts[b * nchunk + i];
  int tot_all = 0;
  for (int i = threadIdx.x; i < nchunk; i += 256) tot_all += counts[b * nchunk + i];
  // reduce both
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { part += __shfl_xor(part, d, 64); tot_all += __shfl_xor(tot_all, d, 64); }
  __shared__ int red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = part; red[1][threadIdx.x >> 6] = tot_all; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_off = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    if (chunk == 0) {
      const int t = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      cand_n[b] = t < cand_cap ? t : cand_cap;
    }
  }
  __syncthreads();
  const int base = chunk * SEL_PIX + threadIdx.x * 8;
  bool f[8];
  float sc[8];
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = base + j;
    f[j] = false;
    sc[j] = 0.0f;
    if (i < H * W) {
      sc[j] = scores[boff + i];
      f[j] = sel_flag(sc[j], i / W, i % W, H, W, thr, border, mask, boff);
    }
    c += f[j] ? 1 : 0;
  }
  int total;
  int pos = s_off + block_excl_scan(c, wsum, total);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (f[j]) {
      if (pos < cand_cap) {
        cand_score[(size_t)b * cand_cap + pos] = sc[j];
        cand_idx[(size_t)b * cand_cap + pos] = base + j;
      }
      ++pos;
    }
}

// top_k_keypoints (src/super_point.cpp:230-251).  n <= k: raster order kept.
// Otherwise the k best by (score desc, raster index asc): 4-pass radix select
// on the score bits, first-come tie completion, bitonic sort of <=1024 keys.
__global__ void __launch_bounds__(1024) topk_kernel(const float *cand_score, const int *cand_idx, const int *cand_n,
                                                    int cand_cap, int k, float *kp_score, int *kp_idx, int *kp_n,
                                                    const int *gate, SpGuard g, float thr) {
  __shared__ int hist[256];
  __shared__ int wsum[16];
  __shared__ unsigned long long keys[kCap];
  __shared__ unsigned s_prefix, s_mask;
  __shared__ int s_rank, s_cnt;
  __shared__ unsigned s_next[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  URF_GATE_FULL(b);
  const int n = cand_n[b];
  const float *cs = cand_score + (size_t)b * cand_cap;
  const int *ci = cand_idx + (size_t)b * cand_cap;
  float *os = kp_score + (size_t)b * kCap;
  int *oi = kp_idx + (size_t)b * kCap;
  const int kk = (k < 0 || k > kCap) ? kCap : k;
  if (n <= kk) {  // includes the k == -1 case up to the slot capacity
    for (int i = tid; i < n; i += 1024) { os[i] = cs[i]; oi[i] = ci[i]; }
    if (tid == 0) {
      kp_n[b] = n;
      // guard: every candidate is kept, so a pixel within its error of the threshold changes the result, and so does any
      // NMS near-tie whose maximum is a candidate
      if (g.flags) {
        int bits = g.band[b] ? 2 : 0;
        if (__int_as_float(g.nms_hi[b]) > (float)thr - guard_err((float)thr, g.delta, g.ulps)) bits |= 4;
        if (bits) atomicOr(&g.flags[b], bits);
        g.amb[(size_t)b * (1 + kAmbMax)] = 0;
      }
    }
    return;
  }
  // ---- radix select: find T = kk-th largest score bit pattern
  if (tid == 0) { s_prefix = 0; s_mask = 0; s_rank = kk; }
  __syncthreads();
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, msk = s_mask;
    for (int i = tid; i < n; i += 1024) {
      const unsigned u = __float_as_uint(cs[i]);
      if ((u & msk) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255], 1);
    }
    __syncthreads();
    {
      // the bin that holds the r-th largest: the one whose count of strictly higher bins is < r <= that count + its own
      // (thread t takes bin 255 - t, so an exclusive scan over the threads is the count above the bin; one thread matches)
      const int r = s_rank, bin = 255 - tid;
      const int own = tid < 256 ? hist[bin] : 0;
      int tot;
      const int above = block_excl_scan(own, wsum, tot);
      if (tid < 256 && above < r && r <= above + own) {
        s_rank = r - above;  // rank inside the chosen bin
        s_prefix = prefix | ((unsigned)bin << (8 * pass));
        s_mask = msk | (0xFFu << (8 * pass));
      }
    }
    __syncthreads();
  }
  const unsigned T = s_prefix;
  const int need_eq = s_rank;  // how many candidates with bits == T to take (first in raster order)
  if (g.flags) {
    // guard: the best candidate that does NOT make the cut (scores are positive floats: their bit patterns order like the
    // values).  More candidates with the cut's own bits than are taken = an exact tie across the cut.
    unsigned nx = 0;
    int eq_total = 0;
    for (int i = tid; i < n; i += 1024) {
      const unsigned u = __float_as_uint(cs[i]);
      if (u < T && u > nx) nx = u;
      eq_total += (u == T) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)nx, d, 64);
      nx = o > nx ? o : nx;
      eq_total += __shfl_xor(eq_total, d, 64);
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_next[tid >> 6] = nx; wsum[tid >> 6] = eq_total; }
    __syncthreads();
    const float cut = __uint_as_float(T);
    if (tid == 0) {
      int eq = 0;
      for (int w = 0; w < 16; ++w) { nx = s_next[w] > nx ? s_next[w] : nx; eq += wsum[w]; }
      const float nxt = eq > need_eq ? cut : __uint_as_float(nx);
      int bits = 0;
      if (guard_near(cut, nxt, g)) bits |= 1;
      // a pixel within its error of the threshold matters only if it could enter the top k
      if (g.band[b] && guard_near(cut, thr, g)) bits |= 2;
      // an NMS near-tie matters only if its maximum could be among the top k: at or above the cut's error band
      if (__int_as_float(g.nms_hi[b]) >= cut - 2.1f * guard_err(cut, g.delta, g.ulps)) bits |= 4;
      s_cnt = bits;
      g.amb[(size_t)b * (1 + kAmbMax)] = 0;
    }
    __syncthreads();
    if (s_cnt & 1) {
      // the candidates within the error of the cut, on either side: their exact scores decide who is in (guard_resolve_kernel).
      // One uniform error for the band (the scores in it differ from the cut by a few 1e-6, so do their errors): 2.1 x err(cut)
      const float band = 2.1f * guard_err(cut, g.delta, g.ulps);
      for (int i = tid; i < n; i += 1024) {
        const float d = cs[i] > cut ? cs[i] - cut : cut - cs[i];
        if (d <= band) {
          const int p = atomicAdd(&g.amb[(size_t)b * (1 + kAmbMax)], 1);
          if (p < kAmbMax) g.amb[(size_t)b * (1 + kAmbMax) + 1 + p] = ci[i];
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int bits = s_cnt;
      if ((bits & 1) && g.amb[(size_t)b * (1 + kAmbMax)] > kAmbMax) bits |= 8;   // too many to resolve one by one: the whole frame
      if (bits) atomicOr(&g.flags[b], bits);
    }
    __syncthreads();
  }
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < kCap; i += 1024) keys[i] = 0ull;
  __syncthreads();
  int eq_seen = 0;  // number of == T candidates in earlier rounds
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    unsigned u = 0;
    bool gt = false, eq = false;
    if (i < n) { u = __float_as_uint(cs[i]); gt = u > T; eq = u == T; }
    int tot;
    const int erank = eq_seen + block_excl_scan(eq ? 1 : 0, wsum, tot);
    if (gt || (eq && erank < need_eq)) {
      const int p = atomicAdd(&s_cnt, 1);
      keys[p] = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ci[i]);
    }
    eq_seen += tot;
    __syncthreads();
  }
  __syncthreads();
  // ---- bitonic sort, descending, 1024 keys
  for (int size = 2; size <= kCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = tid;
      const int j = i ^ stride;
      if (j > i) {
        const unsigned long long a = keys[i], c = keys[j];
        const bool desc = ((i & size) == 0);
        if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[j] = a; }
      }
      __syncthreads();
    }
  }
  if (tid < kk) {
    const unsigned long long key = keys[tid];
    os[tid] = __uint_as_float((unsigned)(key >> 32));
    oi[tid] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
  }
  if (tid == 0) kp_n[b] = kk;
}

// ------------------------------------------------------- descriptor normalise
// F.normalize(p=2, dim=1) (model.py:83).  One wave per cell; lane l owns
// channels 4l..4l+3 (fma chain from x0*x0), 64-lane butterfly, x / max(norm,1e-12)
__global__ void __launch_bounds__(256) desc_norm_kernel(float *desc, int ld, int coff, int ncell, float *out,
                                                        const int *gate, int ncell_frame) {
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (cell >= ncell) return;
  if (gate && (cell >= gate[0] * ncell_frame || gate[kGateMode + cell / ncell_frame] >= 0)) return;
  const f32x4 v = *(const f32x4 *)(desc + (size_t)cell * ld + coff + 4 * lane);
  float a = v[0] * v[0];
  a = __builtin_fmaf(v[1], v[1], a);
  a = __builtin_fmaf(v[2], v[2], a);
  a = __builtin_fmaf(v[3], v[3], a);
  float nrm = __builtin_sqrtf(bfly64_sum(a));
  nrm = nrm > 1e-12f ? nrm : 1e-12f;
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = v[r] / nrm;
  *(f32x4 *)(out + (size_t)cell * 256 + 4 * lane) = o;
}

// ------------------------------------------------------------------ sampling
// sample_descriptors (src/super_point.cpp:253-336) in double like the reference,
// plus packing of the 259-row feature column (:364-384) and of the f32 slot.
// One wave per keypoint, lane l = channels 4l..4l+3; lane 0 does the
// sequential 256-term double sum of squares (std::inner_product order).
__device__ __forceinline__ int clipi(int v, int mx) { return v < 0 ? 0 : (v < mx - 1 ? v : mx - 1); }

__global__ void __launch_bounds__(256) sample_kernel(const float *desc /*[B][Hc*Wc][256]*/, int Hc, int Wc,
                                                     const float *kp_score, const int *kp_idx, const int *kp_n,
                                                     int Ws, double *feat /*[B][kCap][259] or null*/,
                                                     float *slots /*[B][kSlotFloats] or null*/, const int *gate,
                                                     int *kp_n_out, const int *guard_flags) {
  __shared__ double vals[4][256];
  __shared__ double s_inv[4];
  const int b = blockIdx.y;
  URF_GATE_FULL(b);
  // redo of a flagged frame in the exact mode: arena item b goes to the caller's item gate[1 + b]; hdr[1] = 1 marks the slot
  const int ob = gate ? gate[1 + b] : b;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wv;
  const int n = kp_n[b];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (slots) {
      int *hdr = (int *)(slots + (size_t)ob * kSlotFloats);
      // word 1: 1 = the frame was redone whole in the exact mode, 2 = its top-k cut was resolved per candidate (guarded fast mode)
      hdr[0] = n; hdr[1] = gate ? 1 : ((guard_flags && guard_flags[b] == 1) ? 2 : 0); hdr[2] = 0; hdr[3] = 0;
    }
    if (kp_n_out) kp_n_out[ob] = n;
  }
  const bool active = j < n;
  double v[4] = {0, 0, 0, 0};
  int kx = 0, ky = 0;
  float sc = 0.0f;
  if (active) {
    const int idx = kp_idx[(size_t)b * kCap + j];
    sc = kp_score[(size_t)b * kCap + j];
    kx = idx % Ws; ky = idx / Ws;
    const int s = 8, h = Hc, w = Wc;
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
    const float *db = desc + (size_t)b * Hc * Wc * 256;
    const f32x4 a = *(const f32x4 *)(db + ((size_t)iy_nw * w + ix_nw) * 256 + 4 * lane);
    const f32x4 c = *(const f32x4 *)(db + ((size_t)iy_ne * w + ix_ne) * 256 + 4 * lane);
    const f32x4 d = *(const f32x4 *)(db + ((size_t)iy_sw * w + ix_sw) * 256 + 4 * lane);
    const f32x4 e = *(const f32x4 *)(db + ((size_t)iy_se * w + ix_se) * 256 + 4 * lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (double)a[r] * nw + (double)c[r] * ne + (double)d[r] * sw + (double)e[r] * se;
      vals[wv][4 * lane + r] = v[r];
    }
  }
  __syncthreads();
  if (active && lane == 0) {
    double ssq = 0.0;
    for (int c = 0; c < 256; ++c) ssq = ssq + vals[wv][c] * vals[wv][c];
    s_inv[wv] = 1.0 / sqrt(ssq);
  }
  __syncthreads();
  if (!active) return;
  const double inv = s_inv[wv];
  if (feat) {
    double *col = feat + ((size_t)ob * kCap + j) * 259;
    if (lane == 0) { col[0] = (double)sc; col[1] = (double)kx; col[2] = (double)ky; }
#pragma unroll
    for (int r = 0; r < 4; ++r) col[3 + 4 * lane + r] = v[r] * inv;
  }
  if (slots) {
    float *sl = slots + (size_t)ob * kSlotFloats;
    if (lane == 0) {
      f32x4 m = {sc, (float)kx, (float)ky, 0.0f};
      *(f32x4 *)(sl + kSlotHeader + 4 * (size_t)j) = m;
    }
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (float)(v[r] * inv);
    *(f32x4 *)(sl + kSlotHeader + 4 * (size_t)kCap + (size_t)j * 256 + 4 * lane) = o;
  }
}

// ----------------------------------------------------------------- guard redo
// After the fast pass of a batch (up to the top-k selection): frames whose guard word is set go to the redo list (layout:
// urf_common.h), their u8 images to the redo arena.  A frame whose only ambiguity is the top-k cut needs nothing but the exact
// scores of the cells of a few candidates (mode = their number, targets = their cells); any other bit: the whole frame.
// grid (frames, kCompactChunks): the chunks of a frame share the copy of its image.  stats: [0] frames redone whole, [1] frames seen, [2] frames with a cut resolved per candidate,
// [3] threshold band, [4] NMS near-tie, [5] too many candidates at the cut, [6] candidates resolved.
__global__ void __launch_bounds__(256) guard_compact_kernel(const int *flags, const int *amb, int B, int Ws, int Wc,
                                                            const uint8_t *imgs, size_t img_bytes, uint8_t *redo_imgs,
                                                            int *gate, unsigned long long *stats) {
  const int b = blockIdx.x;
  int r = 0, total = 0;
  for (int i = 0; i < B; ++i) {
    const int f = flags[i] != 0;
    r += (i < b) ? f : 0;
    total += f;
  }
  const int mine = flags[b];
  if (threadIdx.x == 0 && blockIdx.y == 0) {
    if (b == 0) {
      gate[0] = total;
      atomicAdd(&stats[1], (unsigned long long)B);
    }
    if (mine) {
      gate[1 + r] = b;
      const bool whole = (mine & ~1) != 0;
      const int na = amb[(size_t)b * (1 + kAmbMax)];
      gate[kGateMode + r] = whole ? -1 : na;
      if (!whole)
        for (int t = 0; t < na; ++t) {
          const int pix = amb[(size_t)b * (1 + kAmbMax) + 1 + t];
          gate[kGateTargets + kAmbMax * r + t] = ((pix / Ws) >> 3) * Wc + ((pix % Ws) >> 3);
        }
      if (whole) atomicAdd(&stats[0], 1ull);
      else { atomicAdd(&stats[2], 1ull); atomicAdd(&stats[6], (unsigned long long)na); }
      if (mine & 2) atomicAdd(&stats[3], 1ull);
      if (mine & 4) atomicAdd(&stats[4], 1ull);
      if (mine & 8) atomicAdd(&stats[5], 1ull);
    }
  }
  if (!mine) return;
  const uint8_t *src = imgs + (size_t)b * img_bytes;
  uint8_t *dst = redo_imgs + (size_t)r * img_bytes;
  const size_t first = (size_t)blockIdx.y * 256 + threadIdx.x, step = (size_t)gridDim.y * 256;
  if ((((size_t)src | (size_t)dst | img_bytes) & 15) == 0) {
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)dst;
    for (size_t i = first; i < img_bytes / 16; i += step) d4[i] = s4[i];
  } else {
    for (size_t i = first; i < img_bytes; i += step) dst[i] = src[i];
  }
}

// Per-candidate resolution of the top-k cut.  Slot r (mode >= 0) of the redo list: the candidates of frame b within the fast
// mode's error of the cut (amb) now have exact scores (heat_x = the exact mode's heat map of the slot, valid at their cells).
// Everything above the band is in the exact mode's top k, everything below it is out (the band is wider than twice the error);
// the places the band's members held in the fast top k go to the band's best by (exact score descending, raster index
// ascending) -- the exact mode's own rule.  The frame's keypoint list is rewritten in place, in score order, the band's members
// carrying their exact scores: a merge by rank (the kept entries are in order already), no sort.  One 1024-thread workgroup
// per redo slot.
__global__ void __launch_bounds__(1024) guard_resolve_kernel(const int *gate, const int *amb, const float *heat_x, int HsWs,
                                                             float *kp_score, int *kp_idx, const int *kp_n) {
  __shared__ unsigned long long akey[kAmbMax], asort[kAmbMax];
  __shared__ int above[kAmbMax];     // kept entries that rank before the band's t-th best
  __shared__ int wsum[16];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (r >= gate[0] || gate[kGateMode + r] < 0) return;
  const int b = gate[1 + r];
  const int na = amb[(size_t)b * (1 + kAmbMax)];
  const int *ai = amb + (size_t)b * (1 + kAmbMax) + 1;
  const int n = kp_n[b];
  float *os = kp_score + (size_t)b * kCap;
  int *oi = kp_idx + (size_t)b * kCap;
  if (tid < kAmbMax) {
    above[tid] = 0;
    akey[tid] = 0ull;
    if (tid < na) {
      const float sx = heat_x[(size_t)r * HsWs + ai[tid]];
      akey[tid] = ((unsigned long long)__float_as_uint(sx) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ai[tid]);
    }
  }
  __syncthreads();
  // the band by (exact score descending, raster index ascending): the keys are distinct, a member's place is the number of
  // members before it
  if (tid < na) {
    int place = 0;
    for (int u = 0; u < na; ++u) place += akey[u] > akey[tid] ? 1 : 0;
    asort[place] = akey[tid];
  }
  // the list (in key order already: topk_kernel sorted it) without the band's members, each kept entry with its place among
  // the kept ones
  unsigned long long key = 0ull;
  bool keep = false;
  if (tid < n) {
    const int idx = oi[tid];
    bool member = false;
    for (int t = 0; t < na; ++t) member = member || ai[t] == idx;
    keep = !member;
    key = ((unsigned long long)__float_as_uint(os[tid]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
  }
  int kept;
  int place = block_excl_scan(keep ? 1 : 0, wsum, kept);     // (its barriers also publish asort and end the reads of the list)
  const int freed = n - kept;                                // the places the band's members held go to the band's best
  for (int t = 0; t < freed; ++t) {
    const unsigned long long bk = asort[t];
    place += (keep && bk > key) ? 1 : 0;
    const unsigned long long before = __ballot(keep && key > bk);
    if ((tid & 63) == 0 && before) atomicAdd(&above[t], __popcll(before));
  }
  __syncthreads();
  if (keep) {
    os[place] = __uint_as_float((unsigned)(key >> 32));
    oi[place] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
  }
  if (tid < freed) {
    const unsigned long long bk = asort[tid];
    const int at = tid + above[tid];
    os[at] = __uint_as_float((unsigned)(bk >> 32));
    oi[at] = (int)(0xFFFFFFFFu - (unsigned)(bk & 0xFFFFFFFFu));
  }
}

// Calibration of the guard's error model err(s) = delta s (1 - s) + c eps s against the exact mode on caller-supplied frames:
// over all pixels, out[0] = the delta that the fast heat map's error needs given c, out[1] = the c it needs given delta
// (bit patterns of non-negative floats, atomicMax).  s is the FAST score, as in guard_err's callers.
// Only scores that can take part in a decision count (at least one of the two above thr_lo, half the keypoint threshold -- the
// guard's own relevance rule); a score of exactly 1 has no delta term, its error is c's to cover (pass 0 of the caller).
__global__ void __launch_bounds__(256) guard_calib_kernel(const float *heat_fast, const float *heat_exact, size_t n,
                                                          float delta, float ulps, float thr_lo, int *out) {
  float need_d = 0.0f, need_c = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = heat_fast[i], x = heat_exact[i];
    const float e = a > x ? a - x : x - a;
    if (!(a > thr_lo || x > thr_lo) || !(a > 0.0f)) continue;
    const float bend = a * (a < 1.0f ? 1.0f - a : 0.0f), grain = 1.1920929e-7f * a;
    if (e > ulps * grain && bend > 0.0f) need_d = fmaxf(need_d, (e - ulps * grain) / bend);
    if (e > delta * bend) need_c = fmaxf(need_c, (e - delta * bend) / grain);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    need_d = fmaxf(need_d, __shfl_xor(need_d, d, 64));
    need_c = fmaxf(need_c, __shfl_xor(need_c, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], __float_as_int(need_d));
    atomicMax(&out[1], __float_as_int(need_c));
  }
}

// ------------------------------------------------------------------ launchers
int launch_softmax(const float *logits, int ld, int Hc, int Wc, float *heat, int B, const int *gate, hipStream_t st) {
  dim3 grid((Hc * Wc + 255) / 256, B);
  hipLaunchKernelGGL(softmax_d2s_kernel, grid, dim3(256), 0, st, logits, ld, Hc, Wc, heat, gate);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_nms(const float *heat, uint8_t *mask, uint8_t *supp, float *ss, float *out, int H, int W, int B,
               const int *gate, const SpGuard &g, float thr_lo, hipStream_t st) {
  dim3 grid((W + NC - 1) / NC, (H + NR - 1) / NR, B), block(256);
  hipLaunchKernelGGL((nms_pass_kernel<0, false>), grid, block, 0, st, heat, mask, supp, ss, out, H, W, gate, g, thr_lo);
  hipLaunchKernelGGL((nms_pass_kernel<1, false>), grid, block, 0, st, heat, mask, supp, ss, out, H, W, gate, g, thr_lo);
  hipLaunchKernelGGL((nms_pass_kernel<2, false>), grid, block, 0, st, heat, mask, supp, ss, out, H, W, gate, g, thr_lo);
  hipLaunchKernelGGL((nms_pass_kernel<1, false>), grid, block, 0, st, heat, mask, supp, ss, out, H, W, gate, g, thr_lo);
  hipLaunchKernelGGL((nms_pass_kernel<2, true>), grid, block, 0, st, heat, mask, supp, ss, out, H, W, gate, g, thr_lo);
  if (g.flags) hipLaunchKernelGGL(nms_tie_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, out, H, W, thr_lo, g);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_select(const float *scores, int H, int W, double thr, int border, const uint8_t *mask, int *counts,
                  float *cand_score, int *cand_idx, int cand_cap, int *cand_n, int k, float *kp_score, int *kp_idx,
                  int *kp_n, int B, const int *gate, const SpGuard &g, hipStream_t st) {
  const int nchunk = (H * W + SEL_PIX - 1) / SEL_PIX;
  dim3 grid(nchunk, B);
  hipLaunchKernelGGL(sel_count_kernel, grid, dim3(256), 0, st, scores, H, W, thr, border, mask, counts, nchunk, gate, g);
  hipLaunchKernelGGL(sel_scatter_kernel, grid, dim3(256), 0, st, scores, H, W, thr, border, mask, counts, nchunk,
                     cand_score, cand_idx, cand_cap, cand_n, gate);
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), 0, st, cand_score, cand_idx, cand_n, cand_cap, k, kp_score,
                     kp_idx, kp_n, gate, g, (float)thr);
  URF_HIP(hipGetLastError());
  return 0;
}
int select_nchunk(int H, int W) { return (H * W + SEL_PIX - 1) / SEL_PIX; }

int launch_desc_norm(float *desc, int ld, int coff, int ncell_total, float *out, const int *gate, int ncell_frame,
                     hipStream_t st) {
  hipLaunchKernelGGL(desc_norm_kernel, dim3((ncell_total + 3) / 4), dim3(256), 0, st, desc, ld, coff, ncell_total, out,
                     gate, ncell_frame);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_sample(const float *desc, int Hc, int Wc, const float *kp_score, const int *kp_idx, const int *kp_n,
                  int Ws, double *feat, float *slots, int B, const int *gate, int *kp_n_out, const int *guard_flags,
                  hipStream_t st) {
  hipLaunchKernelGGL(sample_kernel, dim3(kCap / 4, B), dim3(256), 0, st, desc, Hc, Wc, kp_score, kp_idx, kp_n, Ws,
                     feat, slots, gate, kp_n_out, guard_flags);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_guard_compact(const int *flags, const int *amb, int B, int Ws, int Wc, const uint8_t *imgs, size_t img_bytes,
                         uint8_t *redo_imgs, int *gate, unsigned long long *stats, hipStream_t st) {
  hipLaunchKernelGGL(guard_compact_kernel, dim3(B, 16), dim3(256), 0, st, flags, amb, B, Ws, Wc, imgs, img_bytes, redo_imgs, gate,
                     stats);
  URF_HIP(hipGetLastError());
  return 0;
}
int launch_guard_resolve(const int *gate, const int *amb, const float *heat_x, int HsWs, float *kp_score, int *kp_idx,
                         const int *kp_n, int B, hipStream_t st) {
  hipLaunchKernelGGL(guard_resolve_kernel, dim3(B), dim3(1024), 0, st, gate, amb, heat_x, HsWs, kp_score, kp_idx, kp_n);
  URF_HIP(hipGetLastError());
  return 0;
}

int launch_guard_calib(const float *heat_fast, const float *heat_exact, size_t n, float delta, float ulps, float thr_lo,
                       int *out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16));
  hipLaunchKernelGGL(guard_calib_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, heat_fast, heat_exact, n, delta, ulps, thr_lo, out);
  URF_HIP(hipGetLastError());
  return 0;
}

}  // namespace urf
