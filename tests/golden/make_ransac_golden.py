"""Generates tests/golden/ransac_*.npz: an INDEPENDENT restatement of the reference's two-view initialisation
(EpipolarGeometry, /root/reference/src/epipolar_geometry.cc) in numpy -- float32 arithmetic like the reference,
numpy.linalg.svd (LAPACK) where the reference calls Eigen::JacobiSVD, the reference's own minimal sets drawn from the
C library's srand(0)/rand() exactly as :52-71,100-117 do.  Neither the CPU oracle (oracle/ransac_oracle.c: pivoted
elimination + cyclic Jacobi in f64, wave-order sums) nor the HIP path shares code or summation order with this
script, so agreement within the SVD-vs-Jacobi tolerance pins both against the reference's formulas.

    python tests/golden/make_ransac_golden.py

Functions restated (file:line of the reference): _normalize :735-780, _compute_F21 :247-283, _compute_H21 :207-245,
_check_F :372-449, _check_H :285-370, _find_F :161-205, _find_H :119-159, reconstruct :18-98, _reconstruct_F :451-562,
_reconstruct_H :564-733, _decompose_E :900-926, _triangulate :928-950, _check_R_T :782-898.
Fixtures are data only: inputs (K, keypoints, matches, minimal sets) and expected outputs.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import two_view_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def reference_sets(n, iterations, seed=0):
    """:52-71 with Random::RandomInt :114-117 over the C library's rand() after srand(seed) (:100-112)"""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    sets = np.zeros((iterations, 8), np.int32)
    for it in range(iterations):
        avail = list(range(n))
        for j in range(8):
            d = (len(avail) - 1) - 0 + 1
            randi = int((libc.rand() / (2147483647 + 1.0)) * d) + 0
            sets[it, j] = avail[randi]
            avail[randi] = avail[-1]
            avail.pop()
    return sets


def seq_sum(x):
    """left-to-right float32 accumulation (the reference's `+=` loops)"""
    return np.cumsum(np.asarray(x, f32), dtype=f32)[-1] if len(x) else f32(0)


def normalize(keys):
    keys = np.asarray(keys, f32)
    n = f32(len(keys))
    mx, my = seq_sum(keys[:, 0]) / n, seq_sum(keys[:, 1]) / n
    dx, dy = keys[:, 0] - mx, keys[:, 1] - my
    sX = f32(1.0 / (seq_sum(np.abs(dx)) / n))
    sY = f32(1.0 / (seq_sum(np.abs(dy)) / n))
    T = np.zeros((3, 3), f32)
    T[0, 0], T[1, 1], T[0, 2], T[1, 2], T[2, 2] = sX, sY, -mx * sX, -my * sY, 1
    return np.stack([dx * sX, dy * sY], 1).astype(f32), T


def compute_F21(p1, p2):
    u1, v1, u2, v2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    A = np.stack([u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, np.ones_like(u1)], 1).astype(f32)
    _, _, vt = np.linalg.svd(A, full_matrices=True)
    Fpre = vt[8].reshape(3, 3).astype(f32)
    U, w, Vt = np.linalg.svd(Fpre)
    w[2] = 0
    return (U @ np.diag(w) @ Vt).astype(f32)


def compute_H21(p1, p2):
    rows = []
    for (u1, v1), (u2, v2) in zip(p1, p2):
        rows.append([0, 0, 0, -u1, -v1, -1, v2 * u1, v2 * v1, v2])
        rows.append([u1, v1, 1, 0, 0, 0, -u2 * u1, -u2 * v1, -u2])
    _, _, vt = np.linalg.svd(np.array(rows, f32), full_matrices=True)
    return vt[8].reshape(3, 3).astype(f32)


def check_F(F, x1, x2, sigma):
    th, th_score, inv = f32(3.841), f32(5.991), f32(1.0 / (sigma * sigma))
    u1, v1, u2, v2 = x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]
    a2 = F[0, 0] * u1 + F[0, 1] * v1 + F[0, 2]
    b2 = F[1, 0] * u1 + F[1, 1] * v1 + F[1, 2]
    c2 = F[2, 0] * u1 + F[2, 1] * v1 + F[2, 2]
    num2 = a2 * u2 + b2 * v2 + c2
    chi1 = (num2 * num2 / (a2 * a2 + b2 * b2) * inv).astype(f32)
    a1 = F[0, 0] * u2 + F[1, 0] * v2 + F[2, 0]
    b1 = F[0, 1] * u2 + F[1, 1] * v2 + F[2, 1]
    c1 = F[0, 2] * u2 + F[1, 2] * v2 + F[2, 2]
    num1 = a1 * u1 + b1 * v1 + c1
    chi2 = (num1 * num1 / (a1 * a1 + b1 * b1) * inv).astype(f32)
    terms = np.stack([np.where(chi1 > th, 0, th_score - chi1), np.where(chi2 > th, 0, th_score - chi2)], 1).reshape(-1)
    margin = np.minimum(np.abs(chi1 - th), np.abs(chi2 - th))
    return seq_sum(terms), (chi1 <= th) & (chi2 <= th), margin


def check_H(H21, H12, x1, x2, sigma):
    th, inv = f32(5.991), f32(1.0 / (sigma * sigma))
    u1, v1, u2, v2 = x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1]
    w2 = (1.0 / (H12[2, 0] * u2 + H12[2, 1] * v2 + H12[2, 2])).astype(f32)
    du = u1 - (H12[0, 0] * u2 + H12[0, 1] * v2 + H12[0, 2]) * w2
    dv = v1 - (H12[1, 0] * u2 + H12[1, 1] * v2 + H12[1, 2]) * w2
    chi1 = ((du * du + dv * dv) * inv).astype(f32)
    w1 = (1.0 / (H21[2, 0] * u1 + H21[2, 1] * v1 + H21[2, 2])).astype(f32)
    du = u2 - (H21[0, 0] * u1 + H21[0, 1] * v1 + H21[0, 2]) * w1
    dv = v2 - (H21[1, 0] * u1 + H21[1, 1] * v1 + H21[1, 2]) * w1
    chi2 = ((du * du + dv * dv) * inv).astype(f32)
    terms = np.stack([np.where(chi1 > th, 0, th - chi1), np.where(chi2 > th, 0, th - chi2)], 1).reshape(-1)
    margin = np.minimum(np.abs(chi1 - th), np.abs(chi2 - th))
    return seq_sum(terms), (chi1 <= th) & (chi2 <= th), margin


def triangulate(x1, x2, P1, P2):
    A = np.stack([x1[0] * P1[2] - P1[0], x1[1] * P1[2] - P1[1], x2[0] * P2[2] - P2[0], x2[1] * P2[2] - P2[1]]).astype(f32)
    _, _, vt = np.linalg.svd(A)
    h = vt[3]
    return (h[:3] / h[3]).astype(f32)


def check_R_T(R, t, keys1, keys2, pairs, inl, K, th2):
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    n1 = len(keys1)
    good = np.zeros(n1, bool)
    P3D = np.zeros((n1, 3), f32)
    P1 = np.zeros((3, 4), f32); P1[:, :3] = K
    P2 = (K @ np.c_[R, t]).astype(f32)
    O2 = (-R.T @ t).astype(f32)
    cosines, n_good, border = [], 0, []
    for (i1, i2), ok in zip(pairs, inl):
        if not ok:
            continue
        x1, x2 = keys1[i1], keys2[i2]
        p = triangulate(x1, x2, P1, P2)
        if not np.all(np.isfinite(p)):
            continue
        n2v = p - O2
        cos_par = f32(np.dot(p, n2v) / (np.linalg.norm(p) * np.linalg.norm(n2v)))
        if p[2] <= 0 and cos_par < 0.99998:
            continue
        q = (R @ p + t).astype(f32)
        if q[2] <= 0 and cos_par < 0.99998:
            continue
        e1 = (fx * p[0] / p[2] + cx - x1[0]) ** 2 + (fy * p[1] / p[2] + cy - x1[1]) ** 2
        if e1 > th2:
            continue
        e2 = (fx * q[0] / q[2] + cx - x2[0]) ** 2 + (fy * q[1] / q[2] + cy - x2[1]) ** 2
        if e2 > th2:
            continue
        border.append(min(abs(e1 - th2), abs(e2 - th2)))
        cosines.append(cos_par)
        P3D[i1] = p
        n_good += 1
        if cos_par < 0.99998:
            good[i1] = True
    parallax = 0.0
    if n_good > 0:
        cosines.sort()
        with np.errstate(invalid="ignore"):      # acos of a cosine rounded above 1 is NaN in the reference too
            parallax = float(np.degrees(np.arccos(cosines[min(50, len(cosines) - 1)])))
    return n_good, parallax, good, P3D


def reconstruct(K, keys1, keys2, matches12, sets, sigma=1.0):
    K = np.asarray(K, f32)
    keys1, keys2 = np.asarray(keys1, f32), np.asarray(keys2, f32)
    pairs = [(i, int(m)) for i, m in enumerate(matches12) if m >= 0]
    x1 = np.array([keys1[a] for a, _ in pairs], f32)
    x2 = np.array([keys2[b] for _, b in pairs], f32)
    pn1, T1 = normalize(keys1)
    pn2, T2 = normalize(keys2)
    T2inv, T2t = np.linalg.inv(T2).astype(f32), T2.T
    its = len(sets)
    sF, sH, Fs, Hs = np.zeros(its, f32), np.zeros(its, f32), [], []
    for it in range(its):
        a = np.array([pn1[pairs[j][0]] for j in sets[it]], f32)
        b = np.array([pn2[pairs[j][1]] for j in sets[it]], f32)
        F21 = (T2t @ compute_F21(a, b) @ T1).astype(f32)
        H21 = (T2inv @ compute_H21(a, b) @ T1).astype(f32)
        Fs.append(F21); Hs.append(H21)
        sF[it] = check_F(F21, x1, x2, sigma)[0]
        sH[it] = check_H(H21, np.linalg.inv(H21).astype(f32), x1, x2, sigma)[0]
    bF, bH = int(np.argmax(sF)), int(np.argmax(sH))          # first maximum = strict '>' in order (:150,:197)
    SF, SH = sF[bF], sH[bH]
    out = dict(scoresF=sF, scoresH=sH, bestF=bF, bestH=bH, F21=Fs[bF], H21=Hs[bH])
    _, inlF, marF = check_F(Fs[bF], x1, x2, sigma)
    _, inlH, marH = check_H(Hs[bH], np.linalg.inv(Hs[bH]).astype(f32), x1, x2, sigma)
    out.update(inlF=inlF, marginF=marF, inlH=inlH, marginH=marH)
    RH = SH / (SH + SF)
    th2 = f32(4.0 * sigma * sigma)
    ok, T21, tri, P3D = False, np.eye(4, dtype=f32), np.zeros(len(keys1), bool), np.zeros((len(keys1), 3), f32)
    if RH > 0.5:
        model, N = 0, int(inlH.sum())
        A = (np.linalg.inv(K) @ Hs[bH] @ K).astype(f32)
        U, w, Vt = np.linalg.svd(A)
        s = np.linalg.det(U) * np.linalg.det(Vt)
        d1, d2, d3 = w
        if not (d1 / d2 < 1.00001 or d2 / d3 < 1.00001):
            aux1 = np.sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)); aux3 = np.sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3))
            x1s, x3s = [aux1, aux1, -aux1, -aux1], [aux3, -aux3, aux3, -aux3]
            ast = np.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2)
            ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2)
            st = [ast, -ast, -ast, ast]
            asp = np.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2)
            cp = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2)
            sp = [asp, -asp, -asp, asp]
            cands = []
            for i in range(4):
                Rp = np.array([[ct, 0, -st[i]], [0, 1, 0], [st[i], 0, ct]], f32)
                tp = np.array([x1s[i], 0, -x3s[i]], f32) * (d1 - d3)
                t = U @ tp
                cands.append(((s * U @ Rp @ Vt).astype(f32), (t / np.linalg.norm(t)).astype(f32)))
            for i in range(4):
                Rp = np.array([[cp, 0, sp[i]], [0, -1, 0], [sp[i], 0, -cp]], f32)
                tp = np.array([x1s[i], 0, x3s[i]], f32) * (d1 + d3)
                t = U @ tp
                cands.append(((s * U @ Rp @ Vt).astype(f32), (t / np.linalg.norm(t)).astype(f32)))
            best, second, best_i, best_res = 0, 0, -1, None
            for i, (R, t) in enumerate(cands):
                res = check_R_T(R, t, keys1, keys2, pairs, inlH, K, th2)
                if res[0] > best:
                    second, best, best_i, best_res = best, res[0], i, res
                elif res[0] > second:
                    second = res[0]
            out.update(nGood=np.int32(best), nSecond=np.int32(second))
            if second < 0.75 * best and best_res[1] >= 1.0 and best > 50 and best > 0.9 * N:
                ok = True
                T21[:3, :3], T21[:3, 3] = cands[best_i]
                tri, P3D = best_res[2], best_res[3]
    else:
        model, N = 1, int(inlF.sum())
        E = (K.T @ Fs[bF] @ K).astype(f32)
        U, _, Vt = np.linalg.svd(E)
        t = U[:, 2] / np.linalg.norm(U[:, 2])
        W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], f32)
        R1, R2 = U @ W @ Vt, U @ W.T @ Vt
        R1 = -R1 if np.linalg.det(R1) < 0 else R1
        R2 = -R2 if np.linalg.det(R2) < 0 else R2
        cands = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
        res = [check_R_T(R.astype(f32), tt.astype(f32), keys1, keys2, pairs, inlF, K, th2) for R, tt in cands]
        goods = [r[0] for r in res]
        most = max(goods)
        out.update(nGood=np.int32(most), nGoods=np.array(goods, np.int32))
        if not (most < max(int(0.9 * N), 50) or sum(g > 0.7 * most for g in goods) > 1):
            c = goods.index(most)
            if res[c][1] > 1.0:
                ok = True
                T21[:3, :3], T21[:3, 3] = cands[c]
                tri, P3D = res[c][2], res[c][3]
    out.update(model=np.int32(model), ok=np.bool_(ok), T21=T21.astype(f32), tri=tri, P3D=P3D.astype(f32), SF=SF, SH=SH)
    return out


def main():
    cases = [
        ("ransac_general.npz", dict(seed=0, noise=0.1, outliers=40), 200),              # fundamental-matrix branch, accepted
        ("ransac_general2.npz", dict(seed=5, noise=0.1, outliers=60, motion=2.0), 200),  # larger motion, more outliers
        # a plane seen with few hypotheses: the homography wins (RH = 0.5035); Faugeras' two-fold ambiguity -> rejected
        ("ransac_planar.npz", dict(seed=25, planar=True, outliers=10, noise=0.1), 6),
        ("ransac_allmatched.npz", dict(seed=2, noise=0.1, outliers=60, unmatched=0), 200),   # keys == matches: _find_F alone
    ]
    for name, kw, its in cases:
        K, k1, k2, m, R, t = two_view_scene(**kw)
        if kw.get("unmatched", 30) == 0:          # reorder image 2 so that matches12 is the identity: _find_F over plain lists
            k2 = k2[m]
            m = np.arange(len(m), dtype=np.int32)
        nm = int((m >= 0).sum())
        sets = reference_sets(nm, its, seed=0)
        r = reconstruct(K, k1, k2, m, sets)
        # the fixture is only useful if SVD-vs-Jacobi noise cannot flip the winner
        sF, sH = np.sort(r["scoresF"])[::-1], np.sort(r["scoresH"])[::-1]
        print(name, "model", int(r["model"]), "ok", bool(r["ok"]), "SF %.2f (runner-up %.2f)" % (sF[0], sF[1]),
              "SH %.2f (runner-up %.2f)" % (sH[0], sH[1]), "nGood", int(r.get("nGood", -1)))
        np.savez_compressed(os.path.join(OUT, name), K=K, keys1=k1, keys2=k2, matches12=m, sets=sets, R_true=R, t_true=t,
                            iterations=np.int32(its), **r)


if __name__ == "__main__":
    main()
