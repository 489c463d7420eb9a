"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs (bit-exact), against the committed
golden fixtures made from the reference graph, and through size-independent
properties at the full BASELINE sizes."""
import os

import numpy as np
import pytest

from conftest import golden, make_features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F(U):
    assert U._lib.lib().urf_device_count() >= 1, "GPU tests need an MI355X"
    return U.frontend


@pytest.fixture(scope="module")
def sp640(F, sp_blob):
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=480, max_width=1248, max_batch=8)
    assert sp.build(sp_blob)
    return sp


@pytest.fixture(scope="module")
def pm(F, sg_blob):
    p = F.PointMatching(F.SuperGlueConfig(), max_pairs=8)
    assert p.build(sg_blob)
    return p


@pytest.fixture(scope="module")
def pm_sigma1(F, sg_blob):
    """a matcher whose outlier stage is stated like EpipolarGeometry(K, sigma = 1, 200 iterations), every hypothesis counting"""
    p = F.PointMatching(F.SuperGlueConfig(), ransac_sigma=1.0, ransac_confidence=-1)
    assert p.build(sg_blob)
    return p


# ------------------------------------------------------------------ primitives
def test_mfma_f32_is_an_ordered_fma_chain(F, O):
    """v_mfma_f32_16x16x4_f32 accumulates k in order, one rounding per product:
    the property the whole exact-parity design rests on."""
    rng = np.random.default_rng(0)
    for (M, N, K) in [(16, 16, 4), (128, 64, 64), (200, 68, 256), (257, 512, 512)]:
        A = (rng.standard_normal((M, K)) * 3).astype(np.float32)
        B = (rng.standard_normal((K, N)) * 3).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        assert np.array_equal(F.probe_fma_gemm(A, B, bias), O.fma_gemm(A, B, np.tile(bias, (M, 1))))


def test_canonical_math_matches_oracle_bit_for_bit(F, O):
    x = np.concatenate([np.linspace(-100, 20, 20001), -np.logspace(-8, 2, 3000)]).astype(np.float32)
    e, l = F.probe_math(x)
    eo = np.array([O.lib().o_exp(float(v)) for v in x], np.float32)
    lo = np.array([O.lib().o_log(float(abs(v)) + 1.17549435e-38) for v in x], np.float32)
    assert np.array_equal(e, eo) and np.array_equal(l, lo)
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100000) * 10).astype(np.float32)
    b = (rng.standard_normal(100000) * 3 + 0.01).astype(np.float32)
    q, s, qd, sd = F.probe_divsqrt(a, b)
    assert np.array_equal(q, a / b) and np.array_equal(s, np.sqrt(np.abs(a)))          # IEEE divide / sqrt
    assert np.array_equal(qd, a.astype(np.float64) / b) and np.array_equal(sd, np.sqrt(np.abs(a.astype(np.float64) * b)))


# ------------------------------------------------------------------ SuperPoint
@pytest.mark.parametrize("H,W,k,seed", [(120, 160, 1000, 1), (104, 136, -1, 2), (250, 333, 200, 3)])
def test_superpoint_dense_and_features_bit_exact_vs_oracle(U, F, O, sp_blob, H, W, k, seed):
    img = U.synth.shift_stream(seed, 1, H, W)[0]
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=k), max_height=H, max_width=W)
    assert sp.build(sp_blob)
    feat = sp.infer(img)
    o = O.sp_dense(sp_blob, img)
    Hc, Wc = H // 8, W // 8
    assert np.array_equal(sp.debug_tensor(1, (Hc * 8, Wc * 8)), o["heat"])
    assert np.array_equal(sp.debug_tensor(0, (Hc * 8, Wc * 8)), o["scores"])
    assert np.array_equal(sp.debug_tensor(2, (Hc, Wc, 256)), o["desc"])
    of = O.sp_infer(sp_blob, O.SPConfig(k, 0.0005, 4), img)
    assert feat.shape == of.shape and np.array_equal(feat, of)      # keypoints, scores AND f64 descriptors


def test_superpoint_full_size_640x480_and_kitti(U, O, sp_blob, sp640):
    for (H, W, seed) in [(480, 640, 11), (376, 1241, 12)]:
        img = U.synth.shift_stream(seed, 1, H, W)[0]
        feat = sp640.infer(img)
        of = O.sp_infer(sp_blob, O.SPConfig(1000, 0.0005, 4), img)
        assert feat.shape == (1000, 259) and np.array_equal(feat, of)
        assert feat[:, 1].max() < (W // 8) * 8 and feat[:, 1].min() >= 4           # valid region, borders
        assert np.all(np.diff(feat[:, 0]) <= 0)                                     # score-descending
        assert np.abs(np.linalg.norm(feat[:, 3:], axis=1) - 1).max() < 1e-12


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("name", ["sp_sparse_240x320.npz", "sp_sparse_376x1241.npz", "sp_sparse_480x640.npz"])
def test_superpoint_vs_reference_graph_golden(F, sp_blob, name, prec):
    """HIP output vs the torch run of the reference's model.py (committed fixture).  Exact mode: the same
    keypoint set.  Fast mode (not bit-reproducible): at most one swapped pair of keypoints -- a near-tie at the
    top-k cut -- and every common keypoint is within tolerance."""
    g = golden(name)
    H, W = g["image"].shape
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=int(g["k"])), max_height=H, max_width=W, precision=prec)
    assert sp.build(sp_blob)
    f = sp.infer(g["image"])
    ko = {(int(r[1]), int(r[2])): j for j, r in enumerate(f)}
    want = {(int(x), int(y)): j for j, (x, y) in enumerate(zip(g["x"], g["y"]))}
    if prec == 0:
        assert set(ko) == set(want)                                                   # same keypoint set
    else:
        assert len(set(ko) ^ set(want)) <= 2          # at most one swapped pair at the top-k cut (measured: 0 on all three)
    common = sorted(set(ko) & set(want))
    pf = np.array([ko[c] for c in common]); pg = np.array([want[c] for c in common])
    np.testing.assert_allclose(f[pf, 0], g["score"][pg], rtol=1e-3, atol=1e-5)
    assert np.abs(f[pf, 3:] - g["desc"][pg].astype(np.float64)).max() < 1e-3


def test_superpoint_dense_golden(F, sp_blob):
    g = golden("sp_dense_96x128.npz")
    sp = F.SuperPoint(F.SuperPointConfig(), max_height=96, max_width=128)
    assert sp.build(sp_blob)
    sp.infer(g["image"])
    np.testing.assert_allclose(sp.debug_tensor(0, (96, 128)), g["scores"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sp.debug_tensor(2, (12, 16, 256)), g["desc"], rtol=1e-3, atol=1e-5)


def test_superpoint_mask_blank_and_strided_inputs(U, F, O, sp_blob):
    H, W = 128, 160
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W)
    assert sp.build(sp_blob)
    img = U.synth.shift_stream(5, 1, H, W)[0]
    mask = np.zeros((H, W), np.uint8)
    mask[:, :80] = 7
    f = sp.infer(img, mask)
    of = O.sp_infer(sp_blob, O.SPConfig(1000, 0.0005, 4), img, mask=mask)
    assert np.array_equal(f, of) and f[:, 1].max() < 80 and f[:, 2].min() < 4       # mask path: no border test
    blank = np.zeros((H, W), np.uint8)
    fb = sp.infer(blank)
    assert np.array_equal(fb, O.sp_infer(sp_blob, O.SPConfig(1000, 0.0005, 4), blank))
    big = np.zeros((H, W + 32), np.uint8)
    big[:, :W] = img
    view = big[:, :W]                                                                # cv::Mat with step > cols
    assert np.array_equal(sp.infer(view), sp.infer(img))
    assert sp.infer(np.zeros((8, 8), np.uint8)) is None                              # below the profile minimum


def test_superpoint_batch_and_device_slots_equal_single(U, F, sp_blob, sp640):
    import torch
    frames = U.synth.shift_stream(21, 8, 480, 640)
    single = [sp640.infer(f) for f in frames]
    batch = sp640.infer_batch(frames)
    assert all(np.array_equal(a, b) for a, b in zip(single, batch))
    d = torch.from_numpy(np.stack(frames)).cuda()
    slots = torch.zeros((8, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sp640.infer_device(d.data_ptr(), 8, 480, 640, slots.data_ptr())
    sp640.sync()
    for j in range(8):
        s = F.slot_to_host(slots[j].data_ptr())
        assert np.array_equal(s[:, :3], single[j][:, :3])
        assert np.array_equal(s[:, 3:], single[j][:, 3:].astype(np.float32).astype(np.float64))


def test_nms_idempotent_on_its_own_output(U, F, sp_blob, sp640, O):
    """size-independent property at full size: simple_nms(simple_nms(x)) keeps
    every survivor (survivors are >= 5 px apart)."""
    img = U.synth.shift_stream(31, 1, 480, 640)[0]
    sp640.infer(img)
    s = sp640.debug_tensor(0, (480, 640))
    assert np.array_equal(O.sp_nms(s) != 0, s != 0)
    ys, xs = np.nonzero(s)
    assert len(ys) > 1000
    order = np.lexsort((xs, ys))
    pts = np.c_[ys[order], xs[order]]
    for i in range(0, len(pts), 37):
        d = np.abs(pts - pts[i]).max(axis=1)
        assert (d[d > 0] > 4).all()


# ------------------------------------------------------------------- SuperGlue
@pytest.mark.parametrize("n0,n1,seed", [(1, 1, 0), (17, 130, 1), (64, 64, 2), (300, 257, 3)])
def test_superglue_bit_exact_vs_oracle(F, O, sg_blob, n0, n1, seed):
    rng = np.random.default_rng(seed)
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=min(n0, n1) // 2)
    nf0, nf1 = O.sg_normalize(f0, 640, 512), O.sg_normalize(f1, 640, 512)
    sg = F.SuperGlue(F.SuperGlueConfig())
    assert sg.build(sg_blob)
    i0, i1, m0, m1, Z = sg.infer(nf0, nf1, want_scores=True)
    oi0, oi1, om0, om1, oZ = O.sg_infer(sg_blob, O.SGConfig(640, 512, 0.5, 100), nf0, nf1)
    assert np.array_equal(Z, oZ)                                   # the whole log-assignment tensor
    assert np.array_equal(i0, oi0) and np.array_equal(i1, oi1)
    assert np.array_equal(m0, om0) and np.array_equal(m1, om1)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("name", ["sg_n96.npz", "sg_n320.npz"])
def test_superglue_vs_public_architecture_golden(F, O, sg_blob, name, prec):
    """both precision modes against the public implementation's log-assignment (north_star: scores within
    1e-3, match indices identical)"""
    from conftest import sg_golden_features
    g = golden(name)
    f0, f1 = (g["f0"], g["f1"]) if "f0" in g else sg_golden_features(int(g["n"]), int(g["planted"]), int(g["seed"]))
    nf0, nf1 = O.sg_normalize(f0, 640, 512), O.sg_normalize(f1, 640, 512)
    sg = F.SuperGlue(F.SuperGlueConfig(), precision=prec)
    assert sg.build(sg_blob)
    i0, i1, m0, m1, Z = sg.infer(nf0, nf1, want_scores=True)
    assert np.abs(Z - g["Z"]).max() < 1e-3                          # north_star tolerance on the score tensor
    j0, j1, _, _ = O.sg_decode(g["Z"], 0.5)
    assert np.array_equal(i0, j0) and np.array_equal(i1, j1)


def test_superglue_full_size_and_swap_symmetry(F, O, sg_blob, pm):
    """n = 1024 (profile maximum): swapping the two images transposes the result,
    up to borderline pairs (Sinkhorn updates u before v, so the swap is not an
    exact symmetry in floating point)."""
    rng = np.random.default_rng(9)
    f0 = make_features(rng, 1024)
    f1 = make_features(rng, 1000, planted_from=f0, m=600)
    a = pm.MatchingPoints(f0, f1, False)
    b = pm.MatchingPoints(f1, f0, False)
    assert len(a) >= 590
    sa, sb = {(q, t) for q, t, _ in a}, {(t, q) for q, t, _ in b}
    planted = {(i, i) for i in range(600)}
    assert len(planted & sa) >= 590 and len(planted & sb) >= 590
    assert len(planted & sa & sb) >= 585                      # the confident matches agree
    assert len(sa & sb) >= 0.9 * max(len(sa), len(sb))
    assert len(set(q for q, _, _ in a)) == len(a) and len(set(t for _, t, _ in a)) == len(a)   # one-to-one
    assert pm.MatchingPoints(f0[:0], f1, True) == []


# ------------------------------------------------------------ matching + RANSAC
def test_matching_points_and_ransac_bit_exact_vs_oracle(U, F, O, sp_blob, sg_blob, pm):
    fr = U.synth.shift_stream(1, 2, 240, 320)
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=500), max_height=240, max_width=320)
    assert sp.build(sp_blob)
    f0, f1 = sp.infer(fr[0]), sp.infer(fr[1])
    cfg, rc = O.SGConfig(640, 512, 0.5, 100), O.ref_ransac()
    g_raw, g_rej = pm.MatchingPoints(f0, f1, False), pm.MatchingPoints(f0, f1, True)
    assert g_raw == O.match_points(sg_blob, cfg, rc, f0, f1, False)
    assert g_rej == O.match_points(sg_blob, cfg, rc, f0, f1, True)
    # most matches are true correspondences of the 8-px shift and survive RANSAC
    true = [(q, t) for q, t, _ in g_rej if abs(f0[q, 1] - 8 - f1[t, 1]) < 0.5 and abs(f0[q, 2] - 8 - f1[t, 2]) < 0.5]
    assert len(g_rej) > 200 and len(true) > 0.85 * len(g_rej)
    q = np.array([m[0] for m in g_raw])
    t = np.array([m[1] for m in g_raw])
    s, inl, Fm = pm.find_F(f0[q, 1:3], f1[t, 1:3])
    so, inlo, Fo = O.ransac_find_F(f0[q, 1:3], f1[t, 1:3], rc)
    assert s == so and np.array_equal(inl, inlo) and np.array_equal(Fm, Fo)
    s7, inl7, _ = pm.find_F(f0[q[:7], 1:3], f1[t[:7], 1:3])
    assert s7 == 0 and inl7.sum() == 0                                   # fewer than 8 points


def test_outlier_stage_configurations_bit_exact_vs_oracle(U, F, O, pm, pm_sigma1):
    """the reference call's parameters (3 px, confidence 0.99: default), a pixel threshold of 1.5, and
    EpipolarGeometry's own statement (sigma = 1, every hypothesis): each equals the oracle configured alike,
    and the confidence stop really shortens the search"""
    from conftest import two_view_scene
    _, k1, k2, m12, _, _ = two_view_scene(seed=5, noise=0.4, outliers=60)
    sel = np.where(m12 >= 0)[0]
    p0, p1 = k1[sel], k2[m12[sel]]
    got = {}
    for name, h, oc in (("default", pm, O.ref_ransac()), ("sigma1", pm_sigma1, O.RansacConfig(200, 1.0, 0, 0.0))):
        s, inl, Fm = h.find_F(p0, p1)
        so, io, Fo = O.ransac_find_F(p0, p1, oc)
        assert s == so and np.array_equal(inl, io) and np.array_equal(Fm, Fo), name
        got[name] = inl.sum()
    assert got["default"] > got["sigma1"] > 200                # the 3 px gate keeps more matches than the 1.96 px one
    hp = F.PointMatching(F.SuperGlueConfig(), ransac_threshold_px=1.5, ransac_confidence=0.9)
    assert hp.build(U.synth.pack_sg(U.synth.sg_weights(0)))
    s, inl, Fm = hp.find_F(p0, p1)
    so, io, Fo = O.ransac_find_F(p0, p1, O.RansacConfig(200, float(np.float32(1.5 / np.sqrt(3.841))), 0, 0.9))
    assert s == so and np.array_equal(inl, io) and np.array_equal(Fm, Fo)


def test_device_resident_batch_equals_host_path(U, F, sp_blob, sg_blob, sp640, pm):
    """SP slots -> SuperGlue -> RANSAC entirely on the GPU (8 pairs) gives the
    same match lists as the host-feature API of the reference."""
    import torch
    frames = U.synth.shift_stream(41, 9, 480, 640)
    feats = [sp640.infer(f) for f in frames]
    d = torch.from_numpy(np.stack(frames)).cuda()
    slots = torch.zeros((9, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sp640.infer_device(d[0].data_ptr(), 1, 480, 640, slots[0].data_ptr())
    sp640.infer_device(d[1].data_ptr(), 8, 480, 640, slots[1].data_ptr())
    sp640.sync()
    pm.match_device_async([slots[j].data_ptr() for j in range(8)], [slots[j + 1].data_ptr() for j in range(8)], True)
    dev = pm.fetch(8)
    # the host API takes f64 features; SuperGlue narrows them to f32 like the slots do
    for j in range(8):
        assert dev[j] == pm.MatchingPoints(feats[j], feats[j + 1], True)
        assert len(dev[j]) > 300


def test_cpp_drop_in_api_equals_ctypes_path(U, F, sp_blob, sg_blob, pm, tmp_path):
    """the reference's C++ classes (include/*.h shims), driven like
    Tracking::ExtractFeatureAndMatch, give the same matches as the C-ABI path."""
    import ctypes as C
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "test_shim")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_shim.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ur-mvo_amd"), "-lurf_front", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "ur-mvo_amd")])
    L = U._lib.lib()
    spw, sgw = str(tmp_path / "sp.urfw"), str(tmp_path / "sg.urfw")
    assert L.urf_weights_save(spw.encode(), 1, sp_blob.ctypes.data_as(C.c_void_p), C.c_size_t(sp_blob.size)) == 0
    assert L.urf_weights_save(sgw.encode(), 2, sg_blob.ctypes.data_as(C.c_void_p), C.c_size_t(sg_blob.size)) == 0
    H, W = 240, 320
    fr = U.synth.shift_stream(1, 2, H, W)
    f0p, f1p = str(tmp_path / "f0.raw"), str(tmp_path / "f1.raw")
    fr[0].tofile(f0p)
    fr[1].tofile(f1p)
    vis = str(tmp_path / "keypoints")
    # the shims take the precision mode from a "#precision=N" suffix of engine_file (include/urf_shim.h), never from the environment
    run = lambda sfx, *extra: subprocess.check_output([exe, spw + sfx, sgw + sfx, f0p, f1p, str(H), str(W), *extra], text=True,  # noqa: E731
                                                      env=dict(os.environ, URF_PRECISION="1")).strip().split("\n")
    out = run("#precision=0", vis)
    # the DEFAULT of the drop-in headers is the strict-parity mode: the same keypoints in the same order and the same match
    # index list as the exact mode, line for line (distances within the fast matcher's error)
    strict = run("")
    assert strict[0] == out[0] and len(strict) == len(out)
    assert [l.split()[:2] for l in strict[1:]] == [l.split()[:2] for l in out[1:]]
    assert max(abs(float(a.split()[2]) - float(b.split()[2])) for a, b in zip(strict[1:], out[1:])) < 1e-3
    assert run("#precision=3") == strict
    # row a21 at the boundary: "#outlier=opencv42" behind engine_file makes MatchingPoints(..., true) run the reference's own
    # outlier call, cv::findFundamentalMat(FM_RANSAC, 3, 0.99) restated (urf_sg_config.outlier_stage = 1) -- the list a handle
    # configured that way through the C ABI returns, tuple for tuple; suffixes combine in any order
    cv_run = run("#outlier=opencv42")
    pm_cv = F.PointMatching(F.SuperGlueConfig(), precision=3, outlier_stage=1)
    assert pm_cv.build(sg_blob)
    sp_s = F.SuperPoint(F.SuperPointConfig(max_keypoints=400), max_height=H, max_width=W, precision=3)
    assert sp_s.build(sp_blob)
    g0, g1 = sp_s.infer(fr[0]), sp_s.infer(fr[1])
    want_cv = pm_cv.MatchingPoints(g0, g1, True)
    assert cv_run[0] == f"K0={g0.shape[0]} K1={g1.shape[0]} matches={len(want_cv)}"
    assert [tuple(l.split()[:2]) for l in cv_run[1:]] == [(str(q), str(t)) for q, t, _ in want_cv]
    assert len(cv_run) > 100
    assert run("#outlier=opencv42#precision=3") == cv_run and run("#precision=3#calibrate=0#outlier=opencv42") == cv_run
    assert run("#outlier=8point") == strict
    for sfx in ("#precision=1", "#precision=2"):
        fast = run(sfx)
        # same counts; keypoint INDICES may differ where near-tied scores swap places in the score-sorted list
        same = sum(a.split()[:2] == b.split()[:2] for a, b in zip(fast[1:], out[1:]))
        assert fast[0] == out[0] and same >= 0.95 * (len(out) - 1)
    # build() from the reference's configuration alone (src/super_point.cpp:18-102): engine files that do not exist yet, ONNX
    # files beside them -> the shims read the initialisers, build, and write the caches; the next start deserialises them
    import hashlib
    Wio = U.weights_io
    w_sp = U.synth.sp_weights(0)
    sd_sp = {}
    for name, (Wt, b_) in w_sp.items():
        sd_sp[name + ".weight"], sd_sp[name + ".bias"] = Wt, b_
    sp_onnx, sg_onnx = str(tmp_path / "superpoint_v1.onnx"), str(tmp_path / "superglue.onnx")
    Wio.write_onnx(sp_onnx, sd_sp, [("Conv", [f"x{i}", n + ".weight", n + ".bias"], [f"x{i + 1}"]) for i, (n, *_r) in enumerate(U.synth.SP_CONVS)])
    Wio.write_onnx(sg_onnx, Wio.superglue_to_state_dict(U.synth.sg_weights(0)), [("Conv", ["x", "final_proj.weight", "final_proj.bias"], ["y"])])
    spc, sgc = str(tmp_path / "cache_sp.engine"), str(tmp_path / "cache_sg.engine")
    first = subprocess.check_output([exe, spc, sgc, f0p, f1p, str(H), str(W), "-", sp_onnx, sg_onnx], text=True).strip().split("\n")
    assert first == strict
    sha = lambda pth: hashlib.sha256(open(pth, "rb").read()).hexdigest()    # noqa: E731
    assert sha(spc) == sha(spw) and sha(sgc) == sha(sgw)                      # save_engine(): the Python packer's container, byte for byte
    os.remove(sp_onnx); os.remove(sg_onnx)
    assert subprocess.check_output([exe, spc, sgc, f0p, f1p, str(H), str(W), "-", sp_onnx, sg_onnx], text=True).strip().split("\n") == strict
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=400, engine_file=spw), max_height=H, max_width=W)
    assert sp.build()                                     # engine_file path (deserialize_engine)
    f0, f1 = sp.infer(fr[0]), sp.infer(fr[1])
    m = pm.MatchingPoints(f0, f1, True)
    assert out[0] == f"K0={f0.shape[0]} K1={f1.shape[0]} matches={len(m)}"
    # SuperPoint::visualization (src/super_point.cpp:388-400; PPM without OpenCV): the frame with every keypoint in blue
    raw = open(vis + ".ppm", "rb").read()
    hdr = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(hdr)
    pic = np.frombuffer(raw[len(hdr):], np.uint8).reshape(H, W, 3)
    for r in f1:
        assert tuple(pic[int(r[2]), int(r[1])]) == (0, 0, 255)
    blue = (pic[..., 2] == 255) & (pic[..., 0] == 0) & (pic[..., 1] == 0)
    assert f1.shape[0] <= blue.sum() <= 5 * f1.shape[0]
    untouched = ~blue
    assert np.array_equal(pic[..., 0][untouched], fr[1][untouched])
    got = [tuple(l.split()) for l in out[1:]]
    assert [(int(a), int(b)) for a, b, _ in got] == [(q, t) for q, t, _ in m]
    assert np.allclose([float(c) for _, _, c in got], [d for _, _, d in m], rtol=0, atol=1e-7)


@pytest.mark.parametrize("extra,batch,port", [([], 8, "29533"), (["--resolution", "1241x376", "--batch-per-gpu", "4"], 4, "29537")])
def test_bench_multi_rank_control_flow_on_shared_gpu(tmp_path, extra, batch, port):
    """bench.py --gpus 2 with both ranks on cuda:0 (gloo): exercises the frame
    sharding, the slot all-gather and the cross-rank pairs in the strict-parity mode (bench.py's default).  Every pair of the
    single synthetic stream must still find its ~700 true matches.  Second leg: the geometry of BASELINE configs[3] per GPU
    (1241x376, four frames per rank and step)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, URF_BENCH_SHARED_GPU="1")
    out = subprocess.check_output(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
         "--warmup", "1", "--repeats", "1"] + extra, env=env, text=True, stderr=subprocess.DEVNULL)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * batch and d["scaling"] == "weak"
    assert d["config"]["precision"] == "strict parity"
    assert d["matches_per_step"] > batch * 600      # rank 0's pairs, incl. the pair that crosses the batch seam


def test_bench_event_ordered_rccl_exchange_in_a_one_rank_group():
    """URF_BENCH_FORCE_DIST=1: the N>1 exchange path of bench.py (RCCL all-gather on its own stream,
    ordered against the SuperPoint and matcher streams by events only) in a process group of one rank.
    The match lists must be the ones of the plain single-GPU run."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = []
    for force in ("0", "1"):
        env = dict(os.environ, URF_BENCH_FORCE_DIST=force, MASTER_PORT="29541")
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                                       "--repeats", "2", "--no-cpu-baseline", "--no-exact-check", "--no-secondary"], env=env, text=True, stderr=subprocess.DEVNULL)
        res.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
    assert res[0]["matches_per_step"] == res[1]["matches_per_step"] > 8 * 600
    # (a 4-step region: the three RCCL calls of a step and the end-of-region gathers weigh far more here than in a real run)
    assert res[1]["n_gpus"] == 1 and res[1]["value"] > 0.3 * res[0]["value"]
    # the match lists also travelled through the gather to rank 0 (urf_comm_gather on the matcher stream)
    assert res[0]["matches_last_step_all_ranks_at_rank0"] is None and res[1]["matches_last_step_all_ranks_at_rank0"] > 8 * 600


def test_bench_json_line_follows_the_contract():
    """one JSON line with the driver's keys, the roofline object of the dominant kernel and the cpu_baseline
    object; `traffic` comes from the committed PMC summary"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                                  text=True, stderr=subprocess.DEVNULL)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 3 * 8 / (d["ms_per_step"] * 3e-3)) / d["value"] < 0.01
    rp = d["repeats"]
    assert rp["regions"] == 5 and len(rp["frames_per_s"]) == 5 and rp["min"] <= d["value"] <= rp["max"]
    assert d["value"] == sorted(rp["frames_per_s"])[2]                 # the median region
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] in (8000.0, 2500.0, 157.3)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # traffic: bytes from the committed PMC summary of these very kernel sources, or null with the reason
    assert (isinstance(r["traffic"], int) and r["traffic"] > 1e7) or (r["traffic"] is None and "refused" in r["traffic_note"])
    lost = {k: v["ms_below_roof"] for k, v in r["all_kernels"].items()}
    assert r["kernel"] == max(lost, key=lost.get)                      # the family that loses the most time against its roof
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and 0 < c["value"] < 50 and 1 <= c["cores"] <= 32 and c["sample"]
    # the line times the strict-parity mode; its lists against the exact mode's on the same batches: every pair, index for index
    assert d["config"]["precision"] == "strict parity"
    x = d["exact_mode"]
    assert x["value"] > 100 and x["steps"] == 20
    same, tot = x["pairs_with_identical_index_list_and_distance_within_1e-3"].split("/")
    assert same == tot and int(tot) >= 8 * 10
    # ... and every other single-GPU configuration of BASELINE.json is measured in the same run
    sec = d["secondary"]
    for k in ("guarded_fast_640x480_batch8", "fast_unguarded_640x480_batch8", "strict_parity_1241x376_batch8", "strict_parity_1241x376_batch4",
              "guarded_fast_1241x376_batch8"):
        assert sec[k]["frames_per_s"] > 100 and len(sec[k]["regions_frames_per_s"]) == 3, k
    assert sec["strict_parity_1241x376_batch8"]["pairs_redone_exact"] == sec["strict_parity_1241x376_batch8"]["pairs_flagged"]
    for k in ("configs1_and_per_call_path_strict_parity_640x480", "configs1_and_per_call_path_guarded_fast_640x480"):
        assert 0.1 < sec[k]["superpoint_infer_ms_per_frame"] < 50 and 0.5 < sec[k]["matching_points_ms_per_pair"] < 100, k


@pytest.mark.parametrize("kw,its", [(dict(seed=0, noise=0.0, outliers=40), 200), (dict(seed=2, noise=0.3, outliers=40), 200),
                                    (dict(seed=1, planar=True, outliers=20), 200),
                                    (dict(seed=33, planar=True, outliers=10, noise=0.1), 5),     # homography branch
                                    (dict(seed=5, noise=0.1, outliers=60, motion=2.0), 200)])
def test_epipolar_reconstruct_bit_exact_vs_oracle(F, O, pm, kw, its):
    """EpipolarGeometry::reconstruct: GPU RANSAC searches + host tail vs the oracle"""
    from conftest import two_view_scene
    K, k1, k2, m, R, t = two_view_scene(**kw)
    eg = F.EpipolarGeometry(pm, K, 1.0, its, seed=0)
    ok, T, P, tri, model, sc = eg.reconstruct(k1, k2, m)
    ook, oT, oP, otri, omodel, osc = O.epi_reconstruct(K, k1, k2, m, iterations=its)
    if kw.get("seed") == 33:
        assert model == 0
    if kw.get("seed") == 0:
        assert ok and model == 1 and np.abs(T[:3, :3] - R).max() < 1e-3
    assert (ok, model, sc) == (ook, omodel, osc)
    assert np.array_equal(T, oT) and np.array_equal(tri, otri) and np.array_equal(P, oP)
    ok7, *_ = eg.reconstruct(k1[:7], k2, np.arange(7, dtype=np.int32))
    assert not ok7


@pytest.mark.parametrize("name", ["ransac_general.npz", "ransac_general2.npz", "ransac_planar.npz", "ransac_allmatched.npz"])
def test_reconstruct_over_the_reference_minimal_sets(F, O, pm, pm_sigma1, name):
    """the HIP searches + host tail over the REFERENCE's minimal sets (glibc srand(0)/rand(), restated in the
    product: urf_minimal_sets) -- against the numpy / LAPACK-SVD restatement of the reference (fixture) and, bit
    for bit, against the oracle; explicit sets and sampler = URF_SAMPLER_GLIBC are the same run"""
    from conftest import check_find_F_golden, check_reconstruct_golden
    g = golden(name)
    its, m = int(g["iterations"]), g["matches12"]
    nm = int((m >= 0).sum())
    assert np.array_equal(F.minimal_sets(1, 0, nm, its), g["sets"])
    eg = F.EpipolarGeometry(pm, g["K"], 1.0, its, seed=0, sampler=1)
    res = eg.reconstruct(g["keys1"], g["keys2"], m)
    check_reconstruct_golden(g, res)
    res2 = F.EpipolarGeometry(pm, g["K"], 1.0, its).reconstruct(g["keys1"], g["keys2"], m, sets=g["sets"])
    ores = O.epi_reconstruct(g["K"], g["keys1"], g["keys2"], m, iterations=its, sets=g["sets"])
    for a, b, c in zip(res, res2, ores):
        assert np.array_equal(np.asarray(a), np.asarray(b)) and np.array_equal(np.asarray(a), np.asarray(c))
    if name == "ransac_allmatched.npz":       # every keypoint matched, in order: _find_F alone sees the same problem
        s, inl, Fm = pm_sigma1.find_F_sets(g["keys1"], g["keys2"], g["sets"])
        check_find_F_golden(g, s, inl, Fm)
        so, io, Fo = O.ransac_find_F_sets(g["keys1"], g["keys2"], O.RansacConfig(its, 1.0, 0, 0.0), g["sets"])
        assert s == so and np.array_equal(inl, io) and np.array_equal(Fm, Fo)


# ------------------------------------------------ fast precision mode (split-f16)
def test_split_f16_gemm_is_fp32_accurate(F):
    rng = np.random.default_rng(0)
    for (M, N, K) in [(300, 128, 64), (1000, 256, 512), (2000, 768, 256)]:
        X = (rng.standard_normal((M, K)) * 3).astype(np.float32)
        W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        Y, _ = F.probe_h2gemm(X, W, b, reps=1)
        ref = X.astype(np.float64) @ W.astype(np.float64) + b
        assert np.abs(Y - ref).max() <= 4e-6 * np.abs(ref).max()      # same class as an fp32 GEMM


@pytest.mark.parametrize("n0,n1,seed", [(17, 130, 1), (300, 257, 3), (1000, 1024, 4)])
def test_fast_mode_superglue_matches_exact_mode(F, O, sg_blob, n0, n1, seed):
    """precision=1 (GNN on the f16 matrix core, split operands) against the exact
    mode: log-assignment within the north_star tolerance 1e-3, identical matches."""
    rng = np.random.default_rng(seed)
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=min(n0, n1) // 2)
    nf0, nf1 = O.sg_normalize(f0, 640, 512), O.sg_normalize(f1, 640, 512)
    ex = F.SuperGlue(F.SuperGlueConfig())
    fa = F.SuperGlue(F.SuperGlueConfig(), precision=1)
    assert ex.build(sg_blob) and fa.build(sg_blob)
    i0, i1, m0, m1, Z = ex.infer(nf0, nf1, want_scores=True)
    j0, j1, q0, q1, Zf = fa.infer(nf0, nf1, want_scores=True)
    assert np.abs(Zf - Z).max() < 1e-3
    conf = m0 > 0.6                                          # matches away from the 0.5 threshold
    assert np.array_equal(i0[conf], j0[conf])
    assert (i0 != j0).sum() <= 1 and (i1 != j1).sum() <= 1        # measured: 0
    assert np.abs(q0 - m0).max() < 1e-3


@pytest.mark.parametrize("H,W,seed", [(120, 160, 1), (250, 333, 3), (480, 640, 11), (376, 1241, 12)])
def test_fast_mode_superpoint_matches_exact_mode(U, F, O, sp_blob, H, W, seed):
    """precision=1 (3x3 convs on the f16 matrix core, split operands): dense maps
    agree with the exact mode like the torch run of the reference graph does
    (1e-5 / 1e-6); the keypoint set is the same up to near-ties at the top-k cut."""
    img = U.synth.shift_stream(seed, 1, H, W)[0]
    ex = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W)
    fa = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, precision=1)
    assert ex.build(sp_blob) and fa.build(sp_blob)
    fe, ff = ex.infer(img), fa.infer(img)
    Hc, Wc = H // 8, W // 8
    he, hf = ex.debug_tensor(1, (Hc * 8, Wc * 8)), fa.debug_tensor(1, (Hc * 8, Wc * 8))
    np.testing.assert_allclose(hf, he, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(fa.debug_tensor(2, (Hc, Wc, 256)), ex.debug_tensor(2, (Hc, Wc, 256)), rtol=1e-3, atol=1e-5)
    se, sf = ex.debug_tensor(0, (Hc * 8, Wc * 8)), fa.debug_tensor(0, (Hc * 8, Wc * 8))
    assert ((se != 0) != (sf != 0)).sum() <= 2                       # NMS support: at most a near-tie flips
    ke = {(int(r[1]), int(r[2])) for r in fe}
    kf = {(int(r[1]), int(r[2])) for r in ff}
    assert len(ke ^ kf) <= 2                                 # at most one swapped pair at the top-k cut (measured: 0)
    common = sorted(ke & kf)
    de = {(int(r[1]), int(r[2])): r for r in fe}
    df = {(int(r[1]), int(r[2])): r for r in ff}
    A = np.array([de[k] for k in common])
    Bm = np.array([df[k] for k in common])
    assert np.abs(A[:, 0] - Bm[:, 0]).max() < 2e-5          # scores
    assert np.abs(A[:, 3:] - Bm[:, 3:]).max() < 1e-3        # descriptors: north_star tolerance


# ------------------------------------------------------------------ camera
CAM_K = np.array([[420.5, 0, 318.2], [0, 419.1, 242.7], [0, 0, 1]])
CAM_P = np.array([[400.0, 0, 320, 0], [0, 400, 240, 0], [0, 0, 1, 0]])


@pytest.mark.parametrize("dist,kind", [([-0.28, 0.07, 1e-3, -2e-3, 0.01], 0), ([0.02, -0.01, 0.004, -0.001], 1),
                                       ([0, 0, 0, 0], 0), ([-0.3, 0.1, 0, 0, 0, 0.01, -0.02, 0.003], 0)])
def test_camera_maps_and_undistort_bit_exact_vs_oracle(F, O, dist, kind):
    W, H = 640, 480
    cam = F.Camera(W, H, CAM_K, dist, P=CAM_P, distortion_type=kind)
    m1, m2 = cam.maps()
    o1, o2 = O.cam_init_maps(O.cam_config(W, H, CAM_K, dist, P=CAM_P, distortion_type=kind))
    assert np.array_equal(m1, o1) and np.array_equal(m2, o2)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    assert np.array_equal(cam.UndistortImage(img), O.cam_remap(img, o1, o2))
    # a cv::Mat view with a row stride (ROI of a wider image)
    wide = rng.integers(0, 256, (H, W + 24)).astype(np.uint8)
    assert np.array_equal(cam.UndistortImage(wide[:, 8:8 + W]), O.cam_remap(np.ascontiguousarray(wide[:, 8:8 + W]), o1, o2))


def test_camera_from_maps_edges_and_ragged_sizes(F, O):
    """maps handed over as they are (the maintainer's OpenCV maps): out-of-image, non-finite and
    border-straddling coordinates, a source of a different size than the map, odd widths."""
    rng = np.random.default_rng(11)
    for (oh, ow, sh, sw) in [(33, 61, 40, 50), (480, 640, 480, 640), (1, 1, 7, 5), (17, 1030, 64, 1030)]:
        m1 = rng.uniform(-3, sw + 2, (oh, ow)).astype(np.float32)
        m2 = rng.uniform(-3, sh + 2, (oh, ow)).astype(np.float32)
        m1.flat[0] = np.nan; m2.flat[-1] = np.inf
        if ow > 4:
            m1[0, 1:4] = [sw - 1, sw - 0.5, -0.5]
        img = rng.integers(0, 256, (sh, sw)).astype(np.uint8)
        cam = F.Camera.from_maps(m1, m2)
        assert np.array_equal(cam.UndistortImage(img), O.cam_remap(img, m1, m2)), (oh, ow, sh, sw)


def test_undistort_in_front_of_superpoint_on_one_stream(U, F, O, sp_blob, sp640):
    """device-resident chain raw frames -> remap -> SuperPoint slots (only the raw u8 frame would
    cross PCIe) == oracle remap followed by the host-path SuperPoint"""
    import torch
    H, W, B = 480, 640, 3
    cam = F.Camera(W, H, CAM_K, [-0.28, 0.07, 1e-3, -2e-3, 0.01], P=CAM_P)
    raw = np.stack(U.synth.shift_stream(7, B, H, W))
    d_raw = torch.from_numpy(raw).cuda()
    d_und = torch.zeros_like(d_raw)
    slots = torch.zeros((B, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    cam.undistort_device(d_raw.data_ptr(), B, H, W, d_und.data_ptr(), superpoint=sp640)
    sp640.infer_device(d_und.data_ptr(), B, H, W, slots[0].data_ptr())
    sp640.sync()
    m1, m2 = cam.maps()
    for j in range(B):
        und = O.cam_remap(raw[j], m1, m2)
        assert np.array_equal(d_und[j].cpu().numpy(), und)
        feat = sp640.infer(und)
        # slots keep f32 (what SuperGlue consumes); the host API widens the same values to f64
        assert feat is not None and np.array_equal(F.slot_to_host(slots[j].data_ptr()).astype(np.float32), feat.astype(np.float32))


# ------------------------------------------------------------------ frame stream
def _as_tuples(m):
    return [(int(q), int(t), float(d)) for q, t, d in zip(m["queryIdx"], m["trainIdx"], m["distance"])]


@pytest.mark.parametrize("prec,depth", [(0, 3), (1, 3), (2, 3), (3, 3), (3, 7), (3, 1)])
def test_frame_stream_equals_the_per_frame_calls_of_the_reference(U, F, O, sp_blob, sg_blob, prec, depth):
    """urf_fe (batches, device-resident slots, 3 streams, ragged last batch) == the reference's loop
    SuperPoint::infer(frame) ; PointMatching::MatchingPoints(features_prev, features, matches, true)
    (src/tracking.cc:321-377) as the CPU ORACLE runs it: O.sp_infer / O.match_points on the same 21 frames.
    Exact mode: features and match lists bit for bit; fast modes: the same keypoint sets, and correspondences that may differ
    in a pair whose decisive matching scores are a near-tie (measured on this stream: one pair of twenty differs in two of its
    ~700 correspondences).  depth: batches the caller keeps in flight before it collects -- 7 = urf_fe_max_in_flight() is the pipelined loop
    bench.py times (SuperPoint two batches ahead of the matchers, fetches begun one step before they are ended), 1 = a collect
    right after every submit (integration/tracking.patch)."""
    from conftest import oracle_frames_and_pairs
    frames = np.stack(U.synth.shift_stream(17, 21, 480, 640))
    ofeats, olists = oracle_frames_and_pairs(list(frames), [(t - 1, t) for t in range(1, 21)])
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=1000), F.SuperGlueConfig(), batch=8, max_height=480, max_width=640,
                       precision=prec)
    assert fs.build(sp_blob, sg_blob)
    got_K, got_m, got_f = [], [], []
    for b0 in (0, 8, 16):
        fs.submit(frames[b0:b0 + 8])
        while fs.in_flight() >= depth:
            K, m, f = fs.collect(want_features=True)
            got_K += list(K); got_m += m; got_f += f
    while fs.in_flight():
        K, m, f = fs.collect(want_features=True)
        got_K += list(K); got_m += m; got_f += f
    assert len(got_K) == 21
    assert len(got_m[0]) == 0                                # no predecessor
    differing = 0
    coords = lambda lst, f0, f1: {(f0[q, 1], f0[q, 2], f1[t_, 1], f1[t_, 2]) for q, t_, _ in lst}   # noqa: E731
    for t in range(21):
        assert got_K[t] == ofeats[t].shape[0]
        if prec in (0, 3):
            assert np.array_equal(got_f[t].astype(np.float32), ofeats[t].astype(np.float32))
        else:
            assert {(r[1], r[2]) for r in got_f[t]} == {(r[1], r[2]) for r in ofeats[t]}, t
        if t > 0:
            ref = olists[t - 1]
            if prec == 0:
                assert _as_tuples(got_m[t]) == ref, t
            elif prec == 3:      # strict parity: the oracle's index list, position for position
                got = _as_tuples(got_m[t])
                assert [(q, t_) for q, t_, _ in got] == [(q, t_) for q, t_, _ in ref], t
                assert np.abs(np.array([d for _, _, d in got]) - np.array([d for _, _, d in ref])).max() < 1e-3, t
            else:
                a, b = coords(_as_tuples(got_m[t]), got_f[t - 1], got_f[t]), coords(ref, ofeats[t - 1], ofeats[t])
                assert len(a & b) >= 0.99 * len(a | b), t
                differing += int(a != b)
            assert len(ref) > 300
    # fast modes: at most one pair of the twenty differs from the oracle at all (measured: pair 19, where one keypoint has two
    # equally good partners -- scores 0.508240 and 0.508241 -- and the descriptor noise of the fast SuperPoint decides between
    # them; the guarded matcher flags that pair, tools/gpu_pairdiag.py)
    assert differing <= 1


@pytest.mark.parametrize("matchers,prec", [(1, 3), (1, 0), (3, 3), (4, 2)])
def test_frame_stream_at_full_depth_with_one_three_and_four_matchers(U, F, O, sp_blob, sg_blob, matchers, prec):
    """the in-flight limit is min(matchers + 5, 3 matchers + 2) (a handle holds two begun batches and one enqueued one): with ONE
    matcher every batch shares a handle, and round 5's loop began a third batch on it once six were in flight (deferred error at
    the next collect).  Fill the stream to urf_fe_max_in_flight(), check that one more submit is refused, and compare every
    list with the oracle's (strict / exact: index lists; guarded: pairs may differ in a near-tie)."""
    from conftest import oracle_frames_and_pairs
    frames = np.stack(U.synth.shift_stream(31, 18, 240, 320))
    ofeats, olists = oracle_frames_and_pairs(list(frames), [(t - 1, t) for t in range(1, 18)], max_kp=300)
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=300), F.SuperGlueConfig(), batch=2, max_height=240, max_width=320,
                       precision=prec, matchers=matchers)
    assert fs.build(sp_blob, sg_blob)
    cap = fs.max_in_flight()
    assert cap == min(matchers + 5, 3 * matchers + 2)
    got = []
    nb = 0
    for b0 in range(0, 18, 2):
        if fs.in_flight() == cap:
            with pytest.raises(RuntimeError, match="in flight"):
                fs.submit(frames[b0:b0 + 2])
            got += fs.collect()[1]
        fs.submit(frames[b0:b0 + 2]); nb += 1
    assert fs.in_flight() == min(cap, nb)
    while fs.in_flight():
        got += fs.collect()[1]
    assert len(got) == 18 and len(got[0]) == 0
    for t in range(1, 18):
        a, ref = _as_tuples(got[t]), olists[t - 1]
        if prec == 0:
            assert a == ref, t
        elif prec == 3:
            assert [(q, t_) for q, t_, _ in a] == [(q, t_) for q, t_, _ in ref], t
        else:
            assert len({(q, t_) for q, t_, _ in a} & {(q, t_) for q, t_, _ in ref}) >= 0.97 * len(ref), t


def test_frame_stream_keyframe_references_camera_and_errors(U, F, O, sp_blob, sg_blob, sp640, pm):
    """frames matched against a keyframe up to two batches back, undistortion in front, and the error
    paths: too many batches in flight, a reference that left the ring"""
    K = np.array([[420.5, 0, 318.2], [0, 419.1, 242.7], [0, 0, 1]])
    cam = F.Camera(640, 480, K, [-0.05, 0.01, 1e-4, -2e-4])
    m1, m2 = cam.maps()
    frames = np.stack(U.synth.shift_stream(23, 12, 480, 640, step=(0, 0)))     # a static scene: any pair matches
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=600), F.SuperGlueConfig(), batch=4, max_height=480, max_width=640)
    assert fs.build(sp_blob, sg_blob)
    fs.set_camera(cam)
    from conftest import oracle_frames_and_pairs
    refs = [None, 0, 1, 2, 0, 0, 5, 6, 0, 3, 7, 10]
    # the reference's loop as the CPU oracle runs it: cv::remap, SuperPoint::infer, MatchingPoints(keyframe, frame, true)
    om1, om2 = O.cam_init_maps(O.cam_config(640, 480, K, [-0.05, 0.01, 1e-4, -2e-4]))
    assert np.array_equal(om1, m1) and np.array_equal(om2, m2)
    feats, olists = oracle_frames_and_pairs([O.cam_remap(fr, om1, om2) for fr in frames], [(refs[t], t) for t in range(1, 12)] + [(9, 3)],
                                            max_kp=600)
    out = []
    fs.submit(frames[0:4])                      # predecessor chain
    fs.submit(frames[4:8], ref=[0, 0, 5, -1])   # keyframe 0 (previous batch), an earlier frame of this batch, predecessor
    fs.submit(frames[8:12], ref=[0, 3, 7, 10])  # keyframe two batches back; its match call is deferred
    assert fs.in_flight() == 3 and not fs.ready()   # (the oldest batch's fetch has not begun: a collect would wait)
    while fs.in_flight():
        out += fs.collect()[1]
    for t in range(1, 12):
        assert _as_tuples(out[t]) == olists[t - 1], t
    with pytest.raises(RuntimeError, match="left the ring"):
        fs.submit(frames[0:4], ref=[-1, -1, -1, 0])   # frame 0 is 3 batches back: outside the 2-batch window
    assert fs.in_flight() == 0                          # a rejected submit enqueues nothing
    fs.submit(frames[0:4], ref=[-1, -1, -1, 9])         # ... and leaves the stream usable
    assert _as_tuples(fs.collect()[1][3]) == olists[11]
    # matchers + 5 batches may be in flight (SuperPoint two ahead of the matchers, one batch begun, three waiting to be handed out)
    for _ in range(7):
        fs.submit(frames[4:8])
    with pytest.raises(RuntimeError, match="in flight"):
        fs.submit(frames[4:8])
    assert fs.in_flight() == 7
    n = 0
    while fs.in_flight():
        n += len(fs.collect()[1])
    assert n == 28


def test_frame_stream_ragged_submits_past_the_reference_window(U, F, sp_blob, sg_blob):
    """integration/tracking.patch's keyframe logic, restated: a live queue drains ONE frame per submit, the keyframe stays the
    same for a long time, and the caller asks urf_fe_frame_resident() whether it may still name it.  The ring is counted in
    submits (2 + history_batches = 4 here), not in frames: after four 1-frame submits the keyframe is gone although only four
    frames (half a batch) have passed.  No submit may fail, and the answer must flip exactly when submit would start refusing."""
    frames = np.stack(U.synth.shift_stream(29, 9, 240, 320, step=(0, 0)))
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=300), F.SuperGlueConfig(), batch=8, max_height=240, max_width=320,
                       history_batches=2)
    assert fs.build(sp_blob, sg_blob)
    assert not fs.frame_resident(0)                      # nothing submitted yet
    fs.submit(frames[0:1])                               # frame 0 = the keyframe
    fs.collect()
    seen = []
    for t in range(1, 9):
        ok = fs.frame_resident(0)
        seen.append(ok)
        if ok:
            fs.submit(frames[t:t + 1], ref=[0])
            K, m = fs.collect()
            assert len(m[0]) > 100                       # matched against the keyframe (a static scene)
        else:
            with pytest.raises(RuntimeError, match="left the ring"):
                fs.submit(frames[t:t + 1], ref=[0])      # what the frame-count arithmetic of round 3 would have done
            assert fs.in_flight() == 0
            fs.submit(frames[t:t + 1])                   # predecessor instead: the stream goes on
            fs.collect()
    assert seen == [True] * 4 + [False] * 4, seen
    assert not fs.frame_resident(-1) and not fs.frame_resident(9)


# ------------------------------------------------------------------ the reference's own outlier call (a21, opt-in)
def test_opencv42_outlier_stage_equals_the_oracle(U, F, O, sp_blob, sg_blob, sp640):
    """urf_sg_config.outlier_stage = 1: cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask) restated from
    OpenCV 4.2 (cvransac.hip) against the oracle's restatement of the same written arithmetic (oracle/cvransac_oracle.c),
    bit for bit: through MatchingPoints on frame pairs of a stream (host API and device batch).  (That checker restates the
    kernel's own text; the INDEPENDENT one is test_opencv42_kernel_vs_the_independent_numpy_golden below.)"""
    import torch
    frames = U.synth.shift_stream(41, 5, 480, 640)
    feats = [sp640.infer(f) for f in frames]
    pm_cv = F.PointMatching(F.SuperGlueConfig(), max_pairs=4, outlier_stage=1)
    assert pm_cv.build(sg_blob)
    rc = O.opencv42_ransac()
    want = [O.match_points(sg_blob, O.SGConfig(640, 512, 0.5, 100), rc, feats[j], feats[j + 1], True) for j in range(4)]
    plain = [O.match_points(sg_blob, O.SGConfig(640, 512, 0.5, 100), rc, feats[j], feats[j + 1], False) for j in range(4)]
    for j in range(4):
        assert pm_cv.MatchingPoints(feats[j], feats[j + 1], True) == want[j], j
        assert 300 < len(want[j]) <= len(plain[j])
    assert any(len(want[j]) < len(plain[j]) for j in range(4))          # the stage does reject something on this stream
    d = torch.from_numpy(np.stack(frames)).cuda()
    slots = torch.zeros((5, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sp640.infer_device(d[0].data_ptr(), 1, 480, 640, slots[0].data_ptr())
    sp640.infer_device(d[1].data_ptr(), 4, 480, 640, slots[1].data_ptr())
    sp640.sync()
    pm_cv.match_device_async([slots[j].data_ptr() for j in range(4)], [slots[j + 1].data_ptr() for j in range(4)], True)
    assert pm_cv.fetch(4) == want
    # few matches: the LMedS branch (8 .. 14), exactly 7, fewer than 7
    for n in (13, 7, 4):
        got = pm_cv.MatchingPoints(feats[0][:n], feats[0][:n], True)
        assert got == O.match_points(sg_blob, O.SGConfig(640, 512, 0.5, 100), rc, feats[0][:n], feats[0][:n], True), n
    with pytest.raises(RuntimeError, match="outlier_stage"):
        F.PointMatching(F.SuperGlueConfig(), outlier_stage=2)


@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e", "f"])
def test_opencv42_kernel_vs_the_independent_numpy_golden(U, F, O, name):
    """a21 with a checker that is NOT the kernel's twin: the golden scenes of tests/golden/make_cvransac_golden.py -- an
    independent numpy / LAPACK restatement of cv::findFundamentalMat(FM_RANSAC, 3, 0.99) of OpenCV 4.2 (numpy.linalg.svd for the
    null space, numpy.roots for the cubic, math.log for the iteration count; same cv::RNG stream) -- fed straight through
    cvransac.hip (urf_cv_find_fundamental = the kernel of outlier stage 1 on one correspondence list): the inlier mask, the
    number of hypothesis rounds (it depends on every accepted model's inlier count on the way) and the winning model F up to
    scale.  The kernel's numerics differ from numpy's (Gauss-Jordan, bisection, a multiplication chain), so a borderline
    correspondence could legitimately differ; none does on these scenes.  src/point_matching.cc:43-58.  PARITY UNPINNED
    against an OpenCV binary (none in this image)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"cvransac_{name}.npz"))
    mask, Fm, its = F.findFundamentalMat(g["m0"], g["m1"], 3.0, 0.99)
    assert mask.shape == g["mask"].shape and np.array_equal(mask, g["mask"]), (name, int(mask.sum()), int(g["mask"].sum()))
    assert its == int(g["iterations"]), (name, its, int(g["iterations"]))
    assert np.array_equal(mask, O.cv_find_fundamental_mask(g["m0"], g["m1"], 3.0, 0.99))     # ... and the C restatement agrees
    assert Fm is not None
    a, b = Fm / np.linalg.norm(Fm), g["F"] / np.linalg.norm(g["F"])
    if (a * b).sum() < 0:
        b = -b
    # RANSAC scenes: the same minimal set, so the same model to the solvers' rounding; the LMedS scenes (e, f: 14 points) pick
    # the least median among ~900 models, where two numerics may crown different near-equal models -- their F must at least
    # explain the same inliers (the mask equality above)
    if name in "abcd":
        assert np.abs(a - b).max() < 1e-6, (name, np.abs(a - b).max())
    # a rank-2 matrix either way (the 7-point cubic enforces det F = 0)
    assert abs(np.linalg.det(a)) < 1e-9


def test_opencv42_kernel_small_counts_and_the_oracle_on_random_scenes(U, F, O):
    """the dispatch on the point count (fewer than 7 / exactly 7: nothing rejected; 8..14: LMedS) and 20 random scenes between
    15 and 1024 correspondences against the C restatement, mask for mask"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cvransac_a.npz"))
    keep = np.nonzero(g["truth"])[0]
    for n in (0, 3, 7):
        m, Fm, its = F.findFundamentalMat(g["m0"][keep[:n]], g["m1"][keep[:n]])
        assert m.shape == (n,) and m.all() and Fm is None
    rng = np.random.default_rng(5)
    for t in range(20):
        n = int(rng.integers(8, 1025)) if t else 1024
        idx = rng.integers(0, len(g["m0"]), n)
        jit = rng.integers(-2, 3, (n, 2)).astype(np.float32)
        m0, m1 = g["m0"][idx] + jit, g["m1"][idx] + rng.integers(-1, 2, (n, 2)).astype(np.float32)
        bad = rng.random(n) < 0.3
        m1[bad] = rng.uniform(0, 480, (int(bad.sum()), 2)).astype(np.float32).round()
        m, _, _ = F.findFundamentalMat(m0, m1)
        assert np.array_equal(m, O.cv_find_fundamental_mask(m0, m1)), (t, n)
    with pytest.raises(RuntimeError, match="1024"):
        F.findFundamentalMat(np.zeros((1025, 2), np.float32), np.zeros((1025, 2), np.float32))


# ------------------------------------------------------------------ error behaviour of the boundary
def test_c_abi_errors_are_negative_codes_with_text_and_leave_outputs_untouched(U, F, sp_blob, sg_blob, pm):
    """reference contract: bool/count returns, no exceptions, outputs untouched on failure
    (src/tracking.cc:328-331,346-350)"""
    import ctypes as C
    L = U._lib.lib()
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=100), max_height=64, max_width=64)
    assert not sp.build(sp_blob[:-1]) and b"blob" in L.urf_last_error().lower()        # wrong container size
    assert sp.infer(np.zeros((32, 32), np.uint8)) is None                                 # not built
    assert sp.build(sp_blob)
    feat = np.full((8, 259), 7.0)
    K = C.c_int(-5)
    img = np.zeros((128, 128), np.uint8)
    rc = L.urf_sp_infer(sp._h, img.ctypes.data_as(C.c_void_p), 128, 128, C.c_size_t(128), None, C.c_size_t(0),
                        feat.ctypes.data_as(C.c_void_p), 8, C.byref(K))
    assert rc < 0 and K.value == -5 and (feat == 7.0).all()                               # larger than the arena
    assert len(L.urf_last_error()) > 0
    ok = np.zeros((64, 64), np.uint8); ok[20:40, 20:40] = 255
    assert sp.infer(ok) is not None                                                       # the handle survives
    # a frame with more keypoints than the caller's buffer: error, buffer untouched
    big = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=120, max_width=160)
    assert big.build(sp_blob)
    tex = U.synth.base_frame(3, 120, 160)
    n = big.infer(tex).shape[0]
    assert n > 8
    rc = L.urf_sp_infer(big._h, tex.ctypes.data_as(C.c_void_p), 120, 160, C.c_size_t(160), None, C.c_size_t(0),
                        feat.ctypes.data_as(C.c_void_p), 8, C.byref(K))
    assert rc < 0 and (feat == 7.0).all()
    # matcher: more than 1024 keypoints, zero keypoints, cap too small
    rng = np.random.default_rng(1)
    f0, f1 = make_features(rng, 50), make_features(rng, 60)
    out = (U._lib.DMatch * 4)()
    assert L.urf_match(pm._h, np.zeros((1025, 259)).ctypes.data_as(C.c_void_p), 1025, f1.ctypes.data_as(C.c_void_p), 60,
                       0, out, 4) < 0
    assert pm.MatchingPoints(np.zeros((0, 259)), f1, True) == []                          # nothing to match is not an error
    m = pm.MatchingPoints(f0, f0, False)
    assert len(m) > 4
    assert L.urf_match(pm._h, f0.ctypes.data_as(C.c_void_p), 50, f0.ctypes.data_as(C.c_void_p), 50, 0, out, 4) < 0
    assert b"cap" in L.urf_last_error()
    # a fetch without a batch in flight, or for another pair count than the batch in flight
    fresh = F.PointMatching(F.SuperGlueConfig(), max_pairs=2)
    assert fresh.build(sg_blob)
    with pytest.raises(RuntimeError, match="no batch in flight"):
        fresh.fetch(1)
    # configuration errors are reported at create time
    with pytest.raises(RuntimeError, match="negative"):
        F.SuperPoint(F.SuperPointConfig(keypoint_threshold=-1e-3), max_height=64, max_width=64)
    # urf_last_error() is process-wide: readable from another thread than the one that failed
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(L.urf_last_error()))
    t.start(); t.join()
    assert b"negative" in seen[0]
    # a camera whose maps do not have the size the caller states
    cam = F.Camera(320, 240, np.array([[300.0, 0, 160], [0, 300, 120], [0, 0, 1]]), [0.0, 0, 0, 0])
    fs = F.FrameStream(F.SuperPointConfig(), F.SuperGlueConfig(), batch=2, max_height=240, max_width=320)
    assert L.urf_fe_set_camera(fs._h, cam._h, 480, 640) < 0 and b"maps are 240 x 320" in L.urf_last_error()
    assert L.urf_fe_set_camera(fs._h, cam._h, 240, 320) == 0


# ------------------------------------------------------------------ map-point projection search
@pytest.mark.parametrize("seed,K,M,thr", [(1, 400, 300, 1), (2, 1000, 2000, 3), (3, 64, 5, 2), (4, 1, 1, 1)])
def test_search_by_projection_bit_exact_vs_oracle(U, F, O, seed, K, M, thr):
    from conftest import projection_scene
    sc = projection_scene(seed, K=K, M=M, duplicates=K > 150)
    cfg = O.sbp_config(*sc["cam"], *sc["size"], sc["pose"], thr)
    want = O.search_by_projection(cfg, sc["feat"], sc["pos"], sc["desc"], sc["occupied"], sc["valid"])
    got = F.SearchByProjection(sc["cam"], sc["size"], sc["pose"], sc["feat"], sc["pos"], sc["desc"], thr,
                               occupied=sc["occupied"], mappoint_valid=sc["valid"])
    assert np.array_equal(got, want)
    if K >= 400:
        assert (got >= 0).sum() > M // 10
    # no flags at all
    assert np.array_equal(F.SearchByProjection(sc["cam"], sc["size"], sc["pose"], sc["feat"], sc["pos"], sc["desc"], thr),
                          O.search_by_projection(cfg, sc["feat"], sc["pos"], sc["desc"]))


def test_search_by_projection_on_a_device_slot(U, F, O, sp_blob, sp640):
    """features read straight from the device slot SuperPoint wrote (f32, widened exactly) == the host
    feature matrix of the same frame"""
    import torch
    frame = U.synth.shift_stream(9, 1, 480, 640)[0]
    d = torch.from_numpy(frame).cuda()
    slot = torch.zeros(U._lib.lib().urf_slot_bytes() // 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sp640.infer_device(d.data_ptr(), 1, 480, 640, slot.data_ptr())
    sp640.sync()
    feat = F.slot_to_host(slot.data_ptr())
    K = feat.shape[0]
    rng = np.random.default_rng(0)
    src = rng.integers(0, K, 500)
    fx, fy, cx, cy = 420.0, 415.0, 321.5, 238.25
    depth = rng.uniform(2, 20, 500)
    pc = np.stack([(feat[src, 1] + rng.normal(0, 2, 500) - cx) / fx, (feat[src, 2] + rng.normal(0, 2, 500) - cy) / fy, np.ones(500)], 1) * depth[:, None]
    desc = feat[src, 3:] + rng.normal(0, 0.01, (500, 256))
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    pose = np.eye(4)
    cfg = O.sbp_config(fx, fy, cx, cy, 640, 480, pose, 1)
    want = O.search_by_projection(cfg, feat, pc, desc)
    got = F.SearchByProjection((fx, fy, cx, cy), (640, 480), pose, K, pc, desc, 1, d_slot=slot.data_ptr())
    assert np.array_equal(got, want) and (got >= 0).sum() > 100


def test_fast_mode_degenerate_keypoint_counts(F, sg_blob):
    """0, 1 and 65 keypoints (one full chunk + one masked key) through the fast matcher: no NaN leaks into the
    match lists, and a one-to-one planted pair still matches"""
    rng = np.random.default_rng(21)
    pmf = F.PointMatching(F.SuperGlueConfig(), max_pairs=1, precision=1)
    assert pmf.build(sg_blob)
    f1 = make_features(rng, 65)
    assert pmf.MatchingPoints(np.zeros((0, 259)), f1, True) == []
    assert pmf.MatchingPoints(f1, np.zeros((0, 259)), False) == []
    one = make_features(rng, 1)
    m = pmf.MatchingPoints(one, one, False)
    assert len(m) <= 1 and all(np.isfinite(d) for _, _, d in m)
    m = pmf.MatchingPoints(f1, f1, False)
    assert len(m) >= 60 and all(q == t and np.isfinite(d) for q, t, d in m)


def test_first_call_on_a_fresh_handle_is_not_racing_the_arena_memset(U, F, sp_blob, sg_blob):
    """regression: build() zeroes the arena with hipMemset on the null stream, the handle works on a non-blocking
    stream; without a device synchronisation at the end of build() the first call of a handle created while the
    GPU is busy could run before its buffers were cleared (observed: 0 matches for the first pair)"""
    frames = U.synth.shift_stream(100, 4, 480, 640)
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=480, max_width=640, precision=1)
    assert sp.build(sp_blob)
    f = [sp.infer(x) for x in frames]
    ref = None
    for rep in range(4):
        pm0 = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=8, precision=rep & 1)
        assert pm0.build(sg_blob)                       # big arena (8 pairs) right before the first call
        first = pm0.MatchingPoints(f[rep % 3], f[rep % 3 + 1], False)
        again = pm0.MatchingPoints(f[rep % 3], f[rep % 3 + 1], False)
        assert first == again and len(first) > 600, (rep, len(first), len(again))
        del pm0


def test_ransac_does_not_depend_on_the_order_of_the_matches(F, O, pm):
    """the sampler and every order-dependent float sum walk the correspondences in canonical (sorted) order:
    permuting the match list permutes the inlier flags and nothing else -- on the GPU and in the oracle"""
    from conftest import two_view_scene
    _, k1, k2, m12, _, _ = two_view_scene(seed=5, noise=0.4, outliers=60)
    sel = np.where(m12 >= 0)[0]
    p0, p1 = k1[sel], k2[m12[sel]]
    n = len(p0)
    cfg = O.ref_ransac()
    s_ref, inl_ref, F_ref = pm.find_F(p0, p1)
    so, io, Fo = O.ransac_find_F(p0, p1, cfg)
    assert s_ref == so and np.array_equal(inl_ref, io) and np.array_equal(F_ref, Fo.reshape(3, 3)) and 200 < inl_ref.sum() < n
    rng = np.random.default_rng(0)
    for _ in range(3):
        perm = rng.permutation(n)
        s, inl, Fm = pm.find_F(p0[perm], p1[perm])
        assert s == s_ref and np.array_equal(Fm, F_ref) and np.array_equal(inl, inl_ref[perm])


@pytest.mark.gpu
def test_attention_kernel_forms_of_the_experiments_build_are_bit_identical():
    """attn_h2_il_kernel (round 5's software-pipelined re-scheduling of the fast attention kernel, experiments build only,
    URF_ATTN_IL=1 eight waves / 2 four waves) must give the product kernel's bits: ragged sizes through SuperGlue::infer and
    the batched device path (tools/gpu_attn_il_check.py prints a sha256 per case; the knob is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "ur-mvo_amd", "liburf_front_exp.so")
    if not os.path.exists(so):
        pytest.skip("liburf_front_exp.so is not built")
    outs = {}
    for v in ("0", "1"):       # (URF_ATTN_IL=2, the four-wave form, is compared by tools/gpu_attn_il_check.sh: a third of this test's time)
        env = dict(os.environ, URF_LIB=so, URF_ATTN_IL=v)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_attn_il_check.py")], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        outs[v] = r.stdout
    assert len(outs["0"].splitlines()) >= 12
    assert outs["1"] == outs["0"]

