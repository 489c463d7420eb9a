"""Which arithmetic does v_mfma_f32_16x16x32_f16 implement?  Compares the device result with candidate
models evaluated exactly (Python fractions) on adversarial operands.  python tools/gpu_mfma_model.py"""
import ctypes as C
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg()
L = U._lib.lib()


def run(A, B, Cm):
    n = A.shape[0]
    D = np.zeros((n, 16, 16), np.float32)
    A, B, Cm = np.ascontiguousarray(A, np.float16), np.ascontiguousarray(B, np.float16), np.ascontiguousarray(Cm, np.float32)
    rc = L.urf_probe_mfma_f16(A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), Cm.ctypes.data_as(C.c_void_p),
                              D.ctypes.data_as(C.c_void_p), n, 0)
    assert rc == 0, L.urf_last_error()
    return D


def rn32(fr):
    """round an exact Fraction to the nearest float32 (ties to even)"""
    if fr == 0:
        return np.float32(0.0)
    f = np.float64(fr.numerator) / np.float64(fr.denominator) if abs(fr.numerator) < 2 ** 1000 else float(fr)
    # exact: find the two neighbouring float32 values around fr
    x = np.float32(float(fr))
    cands = [x, np.nextafter(x, np.float32(np.inf)), np.nextafter(x, np.float32(-np.inf))]
    best = None
    for c in cands:
        err = abs(Fraction(float(c)) - fr)
        key = (err, int(np.float32(c).view(np.uint32)) & 1)
        if best is None or key < best[0]:
            best = (key, c)
    return np.float32(best[1])


def model_exact(a, b, c):          # one rounding of the exact sum
    return rn32(Fraction(float(c)) + sum(Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a, b)))


def model_blocks(a, b, c, blk):    # exact sum per block of `blk` k's, rounded after each block
    acc = np.float32(c)
    for k0 in range(0, 32, blk):
        acc = rn32(Fraction(float(acc)) + sum(Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a[k0:k0 + blk], b[k0:k0 + blk])))
    return acc


def model_chain(a, b, c):          # fmaf chain
    acc = np.float32(c)
    for x, y in zip(a, b):
        acc = rn32(Fraction(float(acc)) + Fraction(float(x)) * Fraction(float(y)))
    return acc


rng = np.random.default_rng(0)
n = 64
A = np.zeros((n, 16, 32), np.float16)
B = np.zeros((n, 32, 16), np.float16)
Cm = np.zeros((n, 16, 16), np.float32)
for i in range(n):
    spread = [0, 2, 6, 10, 14][i % 5]
    A[i] = (rng.standard_normal((16, 32)) * np.exp2(rng.integers(-spread, spread + 1, (16, 32)))).astype(np.float16)
    B[i] = (rng.standard_normal((32, 16)) * np.exp2(rng.integers(-spread, spread + 1, (32, 16)))).astype(np.float16)
    Cm[i] = (rng.standard_normal((16, 16)) * np.exp2(rng.integers(-spread, spread + 1, (16, 16)))).astype(np.float32)
# crafted: C = 2^24, 32 unit products
A[0] = 1; B[0] = 1; Cm[0] = 2.0 ** 24
D = run(A, B, Cm)
import math


def fl(fr, mode):
    """Fraction -> float32 with rounding mode 'n' (nearest even) or 'z' (toward zero)"""
    if mode == "n":
        return rn32(fr)
    x = np.float32(float(fr))
    if abs(Fraction(float(x))) > abs(fr):
        x = np.nextafter(x, np.float32(0.0))
    return np.float32(x)


def model_align(a, b, c, blk, guard, mode, order=None):
    """per block: every addend (accumulator and products) is truncated toward zero to a grid of
    2^(emax - 23 - guard), emax = largest exponent among them; exact sum; one rounding"""
    acc = np.float32(c)
    for k0 in range(0, 32, blk):
        terms = [Fraction(float(acc))] + [Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a[k0:k0 + blk], b[k0:k0 + blk])]
        nz = [t for t in terms if t != 0]
        if not nz:
            continue
        emax = max(math.floor(math.log2(abs(float(t)))) if float(t) != 0 else -200 for t in nz)
        q = Fraction(2) ** (emax - 23 - guard)
        tot = Fraction(0)
        for t in terms:
            n_ = t / q
            tot += Fraction(math.floor(n_) if n_ >= 0 else -math.floor(-n_)) * q
        acc = fl(tot, mode)
    return acc


models = {"exact sum, one rounding": model_exact, "fmaf chain over k": model_chain}
for blk in (4, 8, 16, 32):
    models[f"exact per {blk}-block, RN per block"] = (lambda a, b, c, blk=blk: model_blocks(a, b, c, blk))
    for guard in (0, 1, 2, 3, 4, 8):
        for mode in ("n", "z"):
            models[f"aligned-trunc blk{blk} guard{guard} R{mode}"] = (lambda a, b, c, blk=blk, guard=guard, mode=mode: model_align(a, b, c, blk, guard, mode))
hits = {k: 0 for k in models}
hits0 = {k: 0 for k in models}
tot = tot0 = 0
for i in range(n):
    spread = [0, 2, 6, 10, 14][i % 5]
    for r in range(0, 16, 5):
        for cidx in range(0, 16, 5):
            a, b, c = A[i, r, :], B[i, :, cidx], Cm[i, r, cidx]
            tot += 1
            tot0 += spread == 0
            for k, f in models.items():
                m = f(a, b, c)
                ok = m.view(np.uint32) == D[i, r, cidx].view(np.uint32)
                hits[k] += ok
                hits0[k] += ok and spread == 0
print("case 0 (C = 2^24, 32 unit products): device", float(D[0, 0, 0]))
for k in sorted(models, key=lambda k: -hits[k])[:14]:
    print(f"{k:45s}: {hits[k]}/{tot} bit-identical ({hits0[k]}/{tot0} without exponent spread)")
