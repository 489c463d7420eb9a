"""Golden vectors for the OpenCV-4.2 outlier stage (oracle/cvransac_oracle.c, ur-mvo_amd/csrc/cvransac.hip): an INDEPENDENT
numpy restatement of cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask) (src/point_matching.cc:50 of the
reference) -- the same published algorithm (fundam.cpp / ptsetreg.cpp / cv::RNG of OpenCV 4.2.0), but with numpy's own
numerics where OpenCV uses its numerical library: numpy.linalg.svd (LAPACK) for the null space of the 7 x 9 system,
numpy.roots for the cubic, math.log for RANSACUpdateNumIters.  It shares no code and no arithmetic with the oracle (Gauss-
Jordan, bisection, multiplication chains).  UNVERIFIED AGAINST AN OPENCV BINARY: none exists in this image.

    python tests/golden/make_cvransac_golden.py        ->  tests/golden/cvransac_*.npz
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class RNG:
    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def collinear_last(p, count):
    i = count - 1
    for j in range(i):
        # Point2f differences: float32 arithmetic, widened afterwards
        dx1, dy1 = float(np.float32(p[j][0]) - np.float32(p[i][0])), float(np.float32(p[j][1]) - np.float32(p[i][1]))
        for k in range(j):
            dx2, dy2 = float(np.float32(p[k][0]) - np.float32(p[i][0])), float(np.float32(p[k][1]) - np.float32(p[i][1]))
            if abs(dx2 * dy1 - dy2 * dx1) <= np.finfo(np.float32).eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def get_subset(m1, m2, rng, max_attempts=10000):
    n = len(m1)
    for _ in range(max_attempts):
        idx = []
        for i in range(7):
            while True:
                c = rng.uniform(0, n)
                if c not in idx:
                    idx.append(c)
                    break
        if collinear_last(m1[idx], 7) or collinear_last(m2[idx], 7):
            continue
        return idx
    return None


def run7(m1, m2):
    A = np.zeros((7, 9))
    for i in range(7):
        x0, y0, x1, y1 = float(m1[i][0]), float(m1[i][1]), float(m2[i][0]), float(m2[i][1])
        A[i] = [x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0]
    vt = np.linalg.svd(A, full_matrices=True)[2]
    f1, f2 = vt[7].copy(), vt[8].copy()
    f1 -= f2
    # det(lambda f1 + f2) is a cubic in lambda: its coefficients from four evaluations (OpenCV expands it symbolically)
    F1, F2 = f1.reshape(3, 3), f2.reshape(3, 3)
    xs = np.array([0.0, 1.0, -1.0, 2.0])
    ys = np.array([np.linalg.det(x * F1 + F2) for x in xs])
    c = np.linalg.solve(np.vander(xs, 4), ys)                       # c[0] x^3 + c[1] x^2 + c[2] x + c[3]
    roots = [r.real for r in np.roots(c) if abs(r.imag) < 1e-9 * max(1.0, abs(r.real))]
    out = []
    for r in sorted(roots):
        lam, mu = r, 1.0
        s = f1[8] * r + f2[8]
        if abs(s) > np.finfo(float).eps:
            mu = 1.0 / s
            lam *= mu
        Fm = f1 * lam + f2 * mu
        Fm[8] = 1.0 if abs(s) > np.finfo(float).eps else 0.0
        out.append(Fm)
    return out


def errors(F, m1, m2):
    x1, y1, x2, y2 = (m1[:, 0].astype(np.float64), m1[:, 1].astype(np.float64), m2[:, 0].astype(np.float64), m2[:, 1].astype(np.float64))
    a = F[0] * x1 + F[1] * y1 + F[2]; b = F[3] * x1 + F[4] * y1 + F[5]; c = F[6] * x1 + F[7] * y1 + F[8]
    s2 = 1.0 / (a * a + b * b); d2 = x2 * a + y2 * b + c
    a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8]
    s1 = 1.0 / (a * a + b * b); d1 = x1 * a + y1 * b + c
    return np.maximum(d1 * d1 * s1, d2 * d2 * s2).astype(np.float32)


def update_iters(p, ep, max_iters):
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(float).tiny)
    denom = 1.0 - (1.0 - ep) ** 7
    if denom < np.finfo(float).tiny:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(round(num / denom))


def lmeds_mask(m1, m2, conf=0.99):
    """LMeDSPointSetRegistrator::run of OpenCV 4.2 (8 .. 14 points): outlier ratio 0.45, getSubset's default 1000 attempts, the
    model with the least median error, sigma = 2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median), the mask as findInliers leaves
    it -- also when fewer than 7 points pass (run() then reports failure; the mask has been copied out)"""
    n = len(m1)
    rng = RNG()
    niters = max(update_iters(conf, 0.45, 1000), 3)
    best, min_median, it = None, float("inf"), 0
    for it in range(niters):
        idx = get_subset(m1, m2, rng, 1000)
        if idx is None:
            if it == 0:
                return np.ones(n, np.uint8), 0
            break
        for F in run7(m1[idx], m2[idx]):
            e = np.sort(errors(F, m1, m2))
            med = float(e[n // 2]) if n % 2 else float(np.float32(e[n // 2 - 1] + e[n // 2])) * 0.5
            if med < min_median:
                min_median, best = med, F
    if best is None:
        return np.ones(n, np.uint8), it + 1, None
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * math.sqrt(min_median), 0.001)
    return (errors(best, m1, m2) <= np.float32(sigma * sigma)).astype(np.uint8), it + 1, best


def find_fundamental_mask(m1, m2, thresh=3.0, conf=0.99):
    n = len(m1)
    if n <= 7:
        return np.ones(n, np.uint8), 0, None
    if n < 15:
        return lmeds_mask(m1, m2, conf)
    rng = RNG()
    t = np.float32(thresh * thresh)
    niters, max_good, best, bestF = 1000, 0, np.ones(n, np.uint8), None
    it = 0
    while it < niters:
        idx = get_subset(m1, m2, rng)
        if idx is None:
            break
        for F in run7(m1[idx], m2[idx]):
            cur = (errors(F, m1, m2) <= t).astype(np.uint8)
            good = int(cur.sum())
            if good > max(max_good, 6):
                best, max_good, bestF = cur, good, F
                niters = update_iters(conf, (n - good) / n, niters)
        it += 1
    return best, it, bestF


def scene(seed, n_in, n_out, noise):
    rng = np.random.default_rng(seed)
    K = np.array([[420.0, 0, 320], [0, 420, 240], [0, 0, 1]])
    ang = 0.08 * rng.standard_normal(3)
    th = np.linalg.norm(ang); k = ang / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * Kx @ Kx
    tv = np.array([0.4, 0.05, 0.1]) + 0.05 * rng.standard_normal(3)
    X = np.c_[rng.uniform(-3, 3, n_in), rng.uniform(-2, 2, n_in), rng.uniform(4, 12, n_in)]
    p0 = (K @ X.T).T; p0 = p0[:, :2] / p0[:, 2:]
    X1 = (R @ X.T).T + tv
    p1 = (K @ X1.T).T; p1 = p1[:, :2] / p1[:, 2:]
    p0 = p0 + noise * rng.standard_normal(p0.shape); p1 = p1 + noise * rng.standard_normal(p1.shape)
    o0 = np.c_[rng.uniform(0, 640, n_out), rng.uniform(0, 480, n_out)]
    o1 = np.c_[rng.uniform(0, 640, n_out), rng.uniform(0, 480, n_out)]
    m0 = np.r_[p0, o0]; m1 = np.r_[p1, o1]
    order = rng.permutation(len(m0))
    # keypoint coordinates are integers in the reference (src/super_point.cpp:374-379)
    return np.rint(m0[order]).astype(np.float32), np.rint(m1[order]).astype(np.float32), (order < n_in).astype(np.uint8)


def main():
    # e, f: the LMedS branch (8 .. 14 points) -- a clean set, and one so cluttered that fewer than 7 points pass its own threshold
    for name, args in (("a", (3, 300, 60, 0.3)), ("b", (4, 700, 20, 0.5)), ("c", (5, 40, 25, 0.2)), ("d", (6, 15, 3, 0.1)),
                       ("e", (7, 12, 2, 0.1)), ("f", (8, 6, 8, 0.1))):   # (14 points each: with 13 or fewer the median is the error
        # of one of the seven sample points of a model, i.e. rounding noise, and which model "wins" is numerics, not algorithm)
        m0, m1, truth = scene(*args)
        mask, iters, Fbest = find_fundamental_mask(m0, m1)
        # F: the model behind the mask as numpy's SVD / roots give it (scale and sign are the solver's: compare up to both)
        np.savez_compressed(os.path.join(HERE, f"cvransac_{name}.npz"), m0=m0, m1=m1, truth=truth, mask=mask, iterations=iters,
                            F=np.zeros((3, 3)) if Fbest is None else np.asarray(Fbest, np.float64).reshape(3, 3))
        print(name, len(m0), "points,", int(truth.sum()), "true inliers,", int(mask.sum()), "in the mask,", iters, "iterations")


if __name__ == "__main__":
    main()
