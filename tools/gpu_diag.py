"""GPU diagnostic: every stage of the HIP path against the CPU oracle, with
bit-exactness counts.  Run on the GPU box:  python tools/gpu_diag.py [quick]"""
import importlib.util
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ur_mvo_amd", os.path.join(ROOT, "ur-mvo_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "ur-mvo_amd")])
U = importlib.util.module_from_spec(spec)
sys.modules["ur_mvo_amd"] = U
spec.loader.exec_module(U)
from oracle import oracle as O  # noqa: E402

F = U.frontend
synth = U.synth


def cmp(name, a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        print(f"  {name}: SHAPE MISMATCH {a.shape} vs {b.shape}")
        return False
    neq = int((a != b).sum())
    mx = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.size else 0.0
    print(f"  {name}: n={a.size} mismatching={neq} max|diff|={mx:.3e} {'EXACT' if neq == 0 else ''}")
    return neq == 0


def section(t):
    print(f"\n=== {t}", flush=True)


def run(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()


def t_probes():
    section("MFMA fma-chain probe")
    rng = np.random.default_rng(0)
    for (M, N, K) in [(16, 16, 4), (128, 64, 64), (200, 68, 256), (300, 512, 512)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        g = F.probe_fma_gemm(A, B, bias)
        o = O.fma_gemm(A, B, np.tile(bias, (M, 1)))
        cmp(f"gemm {M}x{N}x{K}", g, o)
    section("canonical exp/log/div/sqrt")
    x = np.concatenate([np.linspace(-100, 20, 100001), -np.logspace(-8, 2, 5000)]).astype(np.float32)
    e, l = F.probe_math(x)
    eo = np.array([O.lib().o_exp(float(v)) for v in x[::7]], np.float32)
    lo = np.array([O.lib().o_log(float(abs(v)) + 1.17549435e-38) for v in x[::7]], np.float32)
    cmp("exp_c", e[::7], eo)
    cmp("log_c", l[::7], lo)
    a = rng.standard_normal(200000).astype(np.float32) * 10
    b = (rng.standard_normal(200000).astype(np.float32) * 3 + 0.01)
    q, s, qd, sd = F.probe_divsqrt(a, b)
    cmp("f32 div", q, a / b)
    cmp("f32 sqrt", s, np.sqrt(np.abs(a)))
    cmp("f64 div", qd, a.astype(np.float64) / b.astype(np.float64))
    cmp("f64 sqrt", sd, np.sqrt(np.abs(a.astype(np.float64) * b.astype(np.float64))))


SP_TAPS = {1: (101, 2, 64), 2: (102, 2, 64), 3: (103, 4, 64), 4: (104, 4, 128), 5: (105, 8, 128), 6: (106, 8, 128),
           7: (107, 8, 128)}


def t_sp(H, W, blob, seed=1, topk=1000, mask=False):
    section(f"SuperPoint {H}x{W} k={topk} mask={mask}")
    img = synth.shift_stream(seed, 1, H, W)[0]
    cfg = F.SuperPointConfig(max_keypoints=topk)
    sp = F.SuperPoint(cfg, max_height=H, max_width=W)
    assert sp.build(blob), U._lib.lib().urf_last_error()
    m = None
    if mask:
        m = np.zeros((H, W), np.uint8)
        m[:, : W // 2] = 255
    F.set_profiling(True)
    t = time.time()
    feat = sp.infer(img, m)
    dt = time.time() - t
    assert feat is not None, U._lib.lib().urf_last_error()
    print(f"  gpu infer wall {dt*1e3:.2f} ms; K={feat.shape[0]}")
    ms = sp.stage_ms()
    print("  stages ms:", ", ".join(f"{n}={v:.3f}" for n, v in zip(F.SP_STAGES, ms)))
    t = time.time()
    o = O.sp_dense(blob, img, want_layers=True)
    print(f"  oracle dense {time.time()-t:.2f} s")
    for i, (which, sc, ch) in SP_TAPS.items():
        g = sp.debug_tensor(which, (H // sc, W // sc, ch))
        cmp(f"conv{i} out", g, o["layers"][i])
    Hc, Wc = H // 8, W // 8
    g = sp.debug_tensor(108, (Hc, Wc, 512))
    cmp("convPa", g[..., :256], o["layers"][8])
    cmp("convDa", g[..., 256:], o["layers"][10])
    g = sp.debug_tensor(109, (Hc, Wc, 68))
    cmp("convPb logits", g[..., :65], o["layers"][9])
    cmp("convDb", sp.debug_tensor(111, (Hc, Wc, 256)), o["layers"][11])
    cmp("heat", sp.debug_tensor(1, (Hc * 8, Wc * 8)), o["heat"])
    cmp("scores(nms)", sp.debug_tensor(0, (Hc * 8, Wc * 8)), o["scores"])
    cmp("desc dense", sp.debug_tensor(2, (Hc, Wc, 256)), o["desc"])
    ocfg = O.SPConfig(topk, 0.0005, 4)
    of = O.sp_infer(blob, ocfg, img, mask=m)
    print(f"  oracle K={of.shape[0]}")
    if of.shape == feat.shape:
        cmp("feat score/x/y", feat[:, :3], of[:, :3])
        cmp("feat desc(f64)", feat[:, 3:], of[:, 3:])
    else:
        print("  K MISMATCH")
    return sp, feat, of


def t_sg(n0, n1, sgb, seed=0):
    section(f"SuperGlue n0={n0} n1={n1}")
    rng = np.random.default_rng(seed)

    def mk(n):
        f = np.zeros((n, 259))
        f[:, 0] = rng.uniform(0.001, 1, n).astype(np.float32)
        f[:, 1] = rng.integers(4, 636, n)
        f[:, 2] = rng.integers(4, 476, n)
        d = rng.standard_normal((n, 256))
        f[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
        return f

    f0 = mk(n0)
    f1 = mk(n1)
    m = min(n0, n1) // 2
    f1[:m, 3:] = f0[:m, 3:]  # plant matches
    f1[:m, 1:3] = f0[:m, 1:3] + 3
    sg = F.SuperGlue(F.SuperGlueConfig())
    assert sg.build(sgb), U._lib.lib().urf_last_error()
    nf0 = O.sg_normalize(f0, 640, 512)
    nf1 = O.sg_normalize(f1, 640, 512)
    F.set_profiling(True)
    t = time.time()
    r = sg.infer(nf0, nf1, want_scores=True)
    dt = time.time() - t
    assert r is not None, U._lib.lib().urf_last_error()
    print(f"  gpu sg wall {dt*1e3:.2f} ms; stages:", ", ".join(f"{n}={v:.3f}" for n, v in zip(F.PM_STAGES, sg.stage_ms())))
    i0, i1, m0, m1, Z = r
    ocfg = O.SGConfig(640, 512, 0.5, 100)
    t = time.time()
    oi0, oi1, om0, om1, oZ = O.sg_infer(sgb, ocfg, nf0, nf1)
    print(f"  oracle sg {time.time()-t:.2f} s; matches gpu={int((i0>=0).sum())} oracle={int((oi0>=0).sum())}")
    cmp("Z", Z, oZ)
    cmp("indices0", i0, oi0)
    cmp("indices1", i1, oi1)
    cmp("mscores0", m0, om0)
    cmp("mscores1", m1, om1)
    del sg


def t_match(sgb, f0, f1):
    section(f"MatchingPoints + RANSAC on SP features {f0.shape[0]},{f1.shape[0]}")
    pm = F.PointMatching(F.SuperGlueConfig())
    assert pm.build(sgb)
    F.set_profiling(True)
    t = time.time()
    g = pm.MatchingPoints(f0, f1, True)
    print(f"  gpu wall {(time.time()-t)*1e3:.2f} ms; stages:", ", ".join(f"{n}={v:.3f}" for n, v in zip(F.PM_STAGES, pm.stage_ms())))
    g0 = pm.MatchingPoints(f0, f1, False)
    ocfg = O.SGConfig(640, 512, 0.5, 100)
    rc = O.RansacConfig(200, 1.0, 0)
    o = O.match_points(sgb, ocfg, rc, f0, f1, True)
    o0 = O.match_points(sgb, ocfg, rc, f0, f1, False)
    print(f"  matches: gpu {len(g0)} -> {len(g)} after RANSAC ; oracle {len(o0)} -> {len(o)}")
    print("  no-rejection lists equal:", g0 == o0, " with-rejection lists equal:", g == o)
    if len(g0) >= 8:
        q = np.array([m[0] for m in g0])
        tr = np.array([m[1] for m in g0])
        s, inl, Fm = pm.find_F(f0[q, 1:3], f1[tr, 1:3])
        so, inlo, Fo = O.ransac_find_F(f0[q, 1:3], f1[tr, 1:3], rc)
        print(f"  find_F score gpu {s} oracle {so}")
        cmp("inliers", inl, inlo)
        cmp("F21", Fm, Fo)
    del pm


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    print("devices:", U._lib.lib().urf_device_count())
    spb = synth.pack_sp(synth.sp_weights(0))
    sgb = synth.pack_sg(synth.sg_weights(0))
    run(t_probes)
    run(lambda: t_sp(120, 160, spb))
    if not quick:
        run(lambda: t_sp(104, 136, spb, topk=-1))
        run(lambda: t_sp(240, 320, spb, topk=300, mask=True))
        run(lambda: t_sp(480, 640, spb))
        run(lambda: t_sp(376, 1241, spb))
    run(lambda: t_sg(150, 200, sgb))
    if not quick:
        run(lambda: t_sg(700, 1000, sgb, seed=1))

    def pair():
        fr = synth.shift_stream(1, 2, 240, 320)
        cfg = F.SuperPointConfig(max_keypoints=500)
        sp = F.SuperPoint(cfg, max_height=240, max_width=320)
        assert sp.build(spb)
        f0, f1 = sp.infer(fr[0]), sp.infer(fr[1])
        t_match(sgb, f0, f1)

    run(pair)


if __name__ == "__main__":
    main()
