import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import the hyphenated package directory ur-mvo_amd/ as module ur_mvo_amd"""
    if "ur_mvo_amd" in sys.modules:
        return sys.modules["ur_mvo_amd"]
    d = os.path.join(ROOT, "ur-mvo_amd")
    spec = importlib.util.spec_from_file_location("ur_mvo_amd", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules["ur_mvo_amd"] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def U():
    return load_pkg()


def load_pkg_exp():
    """the same package a second time, bound to the experiments build (liburf_front_exp.so: fault injection, kernel A/B
    switches) -- its own module tree, its own library handle; None when that library was not built"""
    if "ur_mvo_amd_exp" in sys.modules:
        return sys.modules["ur_mvo_amd_exp"]
    d = os.path.join(ROOT, "ur-mvo_amd")
    so = os.path.join(d, "liburf_front_exp.so")
    if not os.path.exists(so):
        return None
    spec = importlib.util.spec_from_file_location("ur_mvo_amd_exp", os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules["ur_mvo_amd_exp"] = m
    old = os.environ.get("URF_LIB")
    os.environ["URF_LIB"] = so
    try:
        spec.loader.exec_module(m)
        m._lib.lib()
    finally:
        if old is None:
            del os.environ["URF_LIB"]
        else:
            os.environ["URF_LIB"] = old
    return m


@pytest.fixture(scope="session")
def Uexp():
    """the experiments build (test hooks); tests that need it skip when it is absent"""
    m = load_pkg_exp()
    if m is None:
        pytest.skip("liburf_front_exp.so is not built (`make -C ur-mvo_amd/csrc experiments`; __graft_entry__.build() does)")
    return m


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sp_blob(U):
    return U.synth.pack_sp(U.synth.sp_weights(0))


@pytest.fixture(scope="session")
def sg_blob(U):
    return U.synth.pack_sg(U.synth.sg_weights(0))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def make_features(rng, n, planted_from=None, m=0, shift=3):
    f = np.zeros((n, 259))
    f[:, 0] = rng.uniform(0.001, 1, n).astype(np.float32)
    f[:, 1] = rng.integers(4, 636, n)
    f[:, 2] = rng.integers(4, 476, n)
    d = rng.standard_normal((n, 256))
    f[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    if planted_from is not None and m:
        f[:m, 3:] = planted_from[:m, 3:]
        f[:m, 1:3] = planted_from[:m, 1:3] + shift
    return f


def two_view_scene(seed=0, n=400, noise=0.0, outliers=40, unmatched=30, planar=False, motion=1.0, tilt=0.0, jitter=0.02):
    """synthetic calibrated two-view scene: K, keys1, keys2 (shuffled), matches12."""
    rng = np.random.default_rng(seed)
    xs, ys = rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n)
    z = 6 + tilt * xs + jitter * rng.standard_normal(n) if planar else rng.uniform(4, 9, n)
    X = np.c_[xs, ys, z]
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1.0]], np.float32)
    th = 0.06 * motion
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([0.5, 0.03, 0.05]) * motion
    p0 = (K @ X.T).T
    p0 = p0[:, :2] / p0[:, 2:]
    p1 = (K @ (R @ X.T + t[:, None])).T
    p1 = p1[:, :2] / p1[:, 2:]
    p1 = p1 + rng.normal(0, noise, p1.shape) if noise else p1
    perm = rng.permutation(n)
    k2 = p1[perm]
    m = np.argsort(perm).astype(np.int32)
    if outliers:
        bad = rng.choice(n, outliers, replace=False)
        m[bad] = rng.integers(0, n, outliers)
    if unmatched:
        m[rng.choice(n, unmatched, replace=False)] = -1
    return K, p0.astype(np.float32), k2.astype(np.float32), m, R, t


def projection_scene(seed, K=400, M=300, W=640, H=480, duplicates=True):
    """a frame (features [K,259]) and map points for Mapping::SearchByProjection: most map points are
    keypoints back-projected along their ray with a noisy copy of the descriptor; some are behind the camera,
    outside the image or invalid; some keypoints are occupied; duplicated descriptors force exact ties."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 420.0, 415.0, 321.5, 238.25
    xy = np.stack([rng.integers(4, W - 4, K), rng.integers(4, H - 4, K)], 1).astype(np.float64)
    d = rng.standard_normal((K, 256)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if duplicates:                                  # identical descriptors a few pixels apart: exact distance ties
        for a_, b_ in ((5, 6), (40, 41), (100, 101)):
            d[b_] = d[a_]
            xy[b_] = xy[a_] + [6, -3]
    feat = np.zeros((K, 259))
    feat[:, 0] = rng.uniform(0.01, 1, K).astype(np.float32)
    feat[:, 1:3] = xy
    feat[:, 3:] = d
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.3, -0.1, 0.2])
    pose = np.eye(4); pose[:3, :3] = R; pose[:3, 3] = t
    src = rng.integers(0, K, M)
    depth = rng.uniform(2, 20, M)
    ray = np.stack([(xy[src, 0] + rng.normal(0, 3, M) - cx) / fx, (xy[src, 1] + rng.normal(0, 3, M) - cy) / fy, np.ones(M)], 1)
    pc = ray * depth[:, None]
    pc[::17, 2] *= -1                               # behind the camera
    pc[5::23, 0] += 1e4                             # far outside the image
    pos = pc @ R.T + t                              # pw = Rwc pc + twc
    md = d[src].astype(np.float64) + rng.normal(0, 0.02, (M, 256))
    md[7::11] = rng.standard_normal((len(md[7::11]), 256))          # unrelated descriptors: rejected by the 0.35 test
    md /= np.linalg.norm(md, axis=1, keepdims=True)
    md[::5] = d[src[::5]]                           # exact copies (distance of the true match ~ 0, ties with duplicates)
    valid = (rng.uniform(size=M) > 0.05).astype(np.uint8)
    occupied = (rng.uniform(size=K) < 0.1).astype(np.uint8)
    return dict(cam=(fx, fy, cx, cy), size=(W, H), pose=pose, feat=feat, pos=pos, desc=md, valid=valid, occupied=occupied)


def sg_golden_features(n, planted, seed):
    """the two feature sets of tests/golden/sg_n*.npz (shared with tests/golden/make_golden.py so that the larger
    fixtures only store the seed): random unit descriptors, `planted` true correspondences shifted by 5 px"""
    rng = np.random.default_rng(seed)

    def mk():
        f = np.zeros((n, 259))
        f[:, 0] = rng.uniform(0.001, 1, n).astype(np.float32)
        f[:, 1] = rng.integers(4, 636, n)
        f[:, 2] = rng.integers(4, 476, n)
        dd = rng.standard_normal((n, 256))
        f[:, 3:] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
        return f

    f0, f1 = mk(), mk()
    f1[:planted, 3:] = f0[:planted, 3:]
    f1[:planted, 1:3] = f0[:planted, 1:3] + 5
    return f0, f1


def check_sg_n1000(g, Z, i0, i1, m0, m1):
    """a SuperGlue result at n = 1000 against tests/golden/sg_n1000.npz (public architecture run by
    transformers, tests/golden/make_golden.py): every 8th row of the log-assignment and both dustbins within the
    north_star tolerance 1e-3, the decode of the FULL tensor (indices) identical, matching scores within 1e-3"""
    assert Z.shape == (1001, 1001)
    assert np.abs(Z[::8] - g["Zrows"]).max() < 1e-3
    assert np.abs(Z[-1] - g["Zbin_row"]).max() < 1e-3 and np.abs(Z[:, -1] - g["Zbin_col"]).max() < 1e-3
    assert np.abs(Z[:-1, :-1].max(1) - g["rowmax"]).max() < 1e-3 and np.abs(Z[:-1, :-1].max(0) - g["colmax"]).max() < 1e-3
    assert np.array_equal(i0, g["indices0"]) and np.array_equal(i1, g["indices1"])
    assert np.abs(m0 - g["mscores0"]).max() < 1e-3 and np.abs(m1 - g["mscores1"]).max() < 1e-3
    planted = int(g["planted"])
    assert (i0[:planted] == np.arange(planted)).sum() >= planted - 2 - planted // 50


RANSAC_GOLDEN = ["ransac_general.npz", "ransac_general2.npz", "ransac_planar.npz", "ransac_allmatched.npz"]


def check_reconstruct_golden(g, result):
    """an EpipolarGeometry::reconstruct result (ok, T21, P3D, tri, model, (SH, SF)) over the fixture's minimal sets
    against tests/golden/ransac_*.npz -- the numpy / LAPACK-SVD restatement of the reference's formulas
    (tests/golden/make_ransac_golden.py).  Tolerances are what f32 SVD vs f64 Jacobi allow."""
    ok, T, P, tri, model, (SH, SF) = result
    assert abs(SF - g["SF"]) <= 2e-4 * g["SF"] and abs(SH - g["SH"]) <= 2e-4 * g["SH"], (SH, SF, g["SH"], g["SF"])
    assert model == int(g["model"]) and ok == bool(g["ok"])
    if ok:
        assert np.abs(T - g["T21"]).max() < 1e-3
        gt = g["tri"].astype(bool)
        assert (tri.astype(bool) != gt).sum() <= max(1, len(gt) // 200)
        both = tri.astype(bool) & gt
        assert both.sum() > 50
        assert np.abs(P[both] - g["P3D"][both]).max() <= 2e-2 * np.abs(g["P3D"][both]).max()
        # and the motion is the scene's (rotation to 1e-2, translation direction to 5e-2)
        assert np.abs(T[:3, :3] - g["R_true"]).max() < 1e-2
        tn = g["t_true"] / np.linalg.norm(g["t_true"])
        assert np.abs(T[:3, 3] - tn).max() < 5e-2
    else:
        assert np.array_equal(T, np.eye(4, dtype=T.dtype)) and not tri.any()


def check_find_F_golden(g, score, inliers, F):
    """_find_F over the fixture's minimal sets (all keypoints matched, so the search sees exactly the reference's
    normalisation): best score, inlier mask (up to correspondences within 1e-2 of the chi-square gate) and the
    fundamental matrix up to scale and sign"""
    assert abs(score - g["SF"]) <= 2e-4 * g["SF"]
    diff = inliers.astype(bool) != g["inlF"]
    assert not (diff & (g["marginF"] > 1e-2)).any() and diff.sum() <= 3
    a, b = F.reshape(-1) / np.linalg.norm(F), g["F21"].reshape(-1) / np.linalg.norm(g["F21"])
    assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-3


def pose_scene(seed, n=300, noise=0.3, outliers=40, W=640, H=480):
    """map points seen by a camera at a known pose Twc: (cam(fx,fy,cx,cy), Xw [n,3], uv [n,2], Rwc, pwc, outlier mask)"""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 420.0, 415.0, 321.5, 238.25
    ax = rng.normal(0, 0.15, 3)
    th = np.linalg.norm(ax)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rwc = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
    pwc = rng.normal(0, 0.5, 3)
    uv = np.c_[rng.uniform(10, W - 10, n), rng.uniform(10, H - 10, n)]
    depth = rng.uniform(2, 15, n)
    pc = np.c_[(uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy, np.ones(n)] * depth[:, None]
    Xw = pc @ Rwc.T + pwc
    uv = uv + rng.normal(0, noise, uv.shape)
    bad = np.zeros(n, bool)
    if outliers:
        idx = rng.choice(n, outliers, replace=False)
        uv[idx] = np.c_[rng.uniform(0, W, outliers), rng.uniform(0, H, outliers)]
        bad[idx] = True
    return (fx, fy, cx, cy), Xw, uv, Rwc, pwc, bad


def quat_wxyz(R):
    """rotation matrix -> unit quaternion (w, x, y, z), w >= 0"""
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2 * np.sign(R[2, 1] - R[1, 2] or 1)
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2 * np.sign(R[0, 2] - R[2, 0] or 1)
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2 * np.sign(R[1, 0] - R[0, 1] or 1)
    return np.array([w, x, y, z])


# ------------------------------------------------------------------ the bench stream through the CPU oracle (GPU tests)
BENCH_STREAM_FRAMES = 40       # bench.py at one GPU: (matchers + 3) x batch = 5 x 8 frames of synth.shift_stream(100, 40, H, W)
SG_CFG = (640, 512, 0.5, 100)
_BENCH_ORACLE = {}


def run_oracle_jobs(kind, jobs, timeout=900):
    """CPU-oracle jobs spread over worker PROCESSES (tests/oracle_worker.py, plain subprocesses with .npz files in between:
    no pickled callables, no fork of a process that holds a GPU).  kind "sp": jobs = [(img, max_kp)] -> [features];
    kind "pm": jobs = [(f0, f1, "ref" | "sigma1")] -> [match list of (queryIdx, trainIdx, distance)]"""
    import subprocess
    import tempfile
    if not jobs:
        return []
    # URF_ORACLE_WORKERS: worker processes (default 1: the GPU boxes show 256 cores behind a cgroup quota of 16 -- six workers of
    # 32 OpenMP threads each took twenty times longer than one, tools/cpu_probe.py; oracle.threads() honours the quota)
    W = max(1, min(int(os.environ.get("URF_ORACLE_WORKERS", "1")), len(jobs)))
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for w in range(W):
            mine = list(range(w, len(jobs), W))
            d = {"kind": np.array(kind), "n": np.array(len(mine))}
            for i, j in enumerate(mine):
                if kind == "sp":
                    d[f"img_{i}"], d[f"max_kp_{i}"] = np.ascontiguousarray(jobs[j][0], np.uint8), np.array(jobs[j][1])
                else:
                    d[f"f0_{i}"], d[f"f1_{i}"], d[f"ransac_{i}"] = jobs[j][0], jobs[j][1], np.array(jobs[j][2])
            src, dst = os.path.join(tmp, f"in{w}.npz"), os.path.join(tmp, f"out{w}.npz")
            np.savez(src, **d)
            procs.append((mine, dst, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oracle_worker.py"), src, dst],
                                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
        out = [None] * len(jobs)
        for mine, dst, p in procs:
            try:
                _, err = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for _, _, q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, err.decode()[-2000:]
            r = np.load(dst)
            for i, j in enumerate(mine):
                if kind == "sp":
                    out[j] = r[f"feat_{i}"]
                else:
                    out[j] = [(int(q), int(t), float(np.float32(dd))) for q, t, dd in r[f"m_{i}"]]
        return out


def oracle_frames_and_pairs(frames, pairs, max_kp=1000):
    """O.sp_infer on every frame and O.match_points (default outlier stage, outlier rejection on) on every (first, second)
    index pair, spread over worker processes: (features, match lists)"""
    feats = run_oracle_jobs("sp", [(f, max_kp) for f in frames])
    lists = run_oracle_jobs("pm", [(feats[a], feats[b], "ref") for a, b in pairs])
    return feats, lists


def bench_stream_oracle(H, W, n_sigma1=8):
    """The stream bench.py times -- synth.shift_stream(100, 40, H, W) -- through the CPU oracle:
    frames, features of all 40 frames, the match list of every consecutive pair (j - 1, j) mod 40 with the handle's
    default outlier stage (the reference call's 3 px / 0.99, O.ref_ransac()) as lists["ref"][j], and the first `n_sigma1`
    pairs with EpipolarGeometry's own statement (sigma = 1, every hypothesis counts) as lists["sigma1"][j] (pair (j, j + 1)).
    Computed once per session, oracle jobs spread over worker processes."""
    if (H, W) in _BENCH_ORACLE:
        return _BENCH_ORACLE[(H, W)]
    U = load_pkg()
    frames = U.synth.shift_stream(100, BENCH_STREAM_FRAMES, H, W)
    n = BENCH_STREAM_FRAMES
    feats = run_oracle_jobs("sp", [(f, 1000) for f in frames])
    jobs = [(feats[(j - 1) % n], feats[j], "ref") for j in range(n)]
    jobs += [(feats[j], feats[j + 1], "sigma1") for j in range(n_sigma1)]
    res = run_oracle_jobs("pm", jobs)
    out = (frames, feats, {"ref": res[:n], "sigma1": res[n:]})
    _BENCH_ORACLE[(H, W)] = out
    return out


POSE_GOLDEN = ["pose_mono_a.npz", "pose_mono_b.npz", "pose_mono_few.npz", "pose_stereo_a.npz", "pose_stereo_b.npz"]


def check_pose_golden(g, frame_opt, pnp=None):
    """a FrameOptimization result (n - outliers, q_wc, p_wc, inlier flags) -- and, for the mono fixtures, a SolvePnPWithCV result
    (n_inliers, Twc, inlier flags) -- against tests/golden/pose_*.npz: the independent numpy restatement of the written
    specification (tests/golden/make_pose_golden.py: numpy.linalg.solve / svd, scipy rotations, numpy sums).  Poses to 1e-6,
    flags equal except for observations within 1e-6 of their gate."""
    k, q, p, inl = frame_opt
    assert abs(np.linalg.norm(q) - 1) < 1e-12
    dq = min(np.abs(q - g["q"]).max(), np.abs(q + g["q"]).max())
    assert dq < 1e-6 and np.abs(p - g["p"]).max() < 1e-6, (dq, np.abs(p - g["p"]).max())
    diff = inl.astype(bool) != g["inlier"].astype(bool)
    assert not (diff & (np.abs(g["margin"]) > 1e-6)).any()
    assert k == int(inl.sum())
    if pnp is not None:
        n_in, T, pinl = pnp
        d = pinl.astype(bool) != g["pnp_inlier"].astype(bool)
        assert not (d & (np.abs(g["pnp_margin"]) > 1e-6)).any() and n_in == int(pinl.sum())
        assert np.abs(T - g["pnp_pose"]).max() < 1e-6, np.abs(T - g["pnp_pose"]).max()
