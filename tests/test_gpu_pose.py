"""Pose stage on the GPU (-m gpu): urf_solve_pnp_ransac / urf_frame_optimization
through the C ABI against the CPU oracle, bit for bit (f64, canonical summation order), batched frames of
ragged sizes, and the edge cases of the reference's call sites (src/g2o_optimization.cc:352-353, :309-310)."""
import numpy as np
import pytest

from conftest import pose_scene, quat_wxyz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F(U):
    assert U._lib.lib().urf_device_count() >= 1, "GPU tests need an MI355X"
    return U.frontend


def _scenes():
    out = [pose_scene(s, n=n, noise=nz, outliers=o) for s, n, nz, o in
           [(0, 300, 0.0, 0), (1, 300, 0.3, 40), (2, 1000, 0.5, 300), (3, 64, 0.3, 5), (4, 9, 0.1, 0), (5, 7, 0.1, 0), (6, 500, 1.0, 200),
            (7, 8, 0.0, 0)]]
    return out


def test_pnp_ransac_batch_bit_exact_vs_oracle(F, O):
    sc = _scenes()
    cam = sc[0][0]
    ps = F.PoseStage(cam, max_batch=len(sc), capacity=1024)
    got = ps.SolvePnPWithCV([s[1] for s in sc], [s[2] for s in sc])
    for f, s in enumerate(sc):
        k, T, inl = O.solve_pnp_ransac(cam, s[1], s[2])
        assert got[f][0] == k and np.array_equal(got[f][1], T) and np.array_equal(got[f][2], inl), f
        if len(s[1]) >= 8 and k > 0:
            assert np.abs(T[:3, :3] - s[3]).max() < 2e-2 and inl[s[5]].sum() <= max(3, s[5].sum() // 6)
    assert got[5][0] == 0 and np.array_equal(got[5][1], np.eye(4))          # 7 correspondences: nothing (:352-353)
    # other parameters of the call: fewer hypotheses, a tighter gate, another seed
    g2 = ps.SolvePnPWithCV([sc[2][1]], [sc[2][2]], iterations=20, reprojection_error=4.0, confidence=0.9, seed=11)
    k, T, inl = O.solve_pnp_ransac(cam, sc[2][1], sc[2][2], iterations=20, reprojection_error=4.0, confidence=0.9, seed=11)
    assert g2[0][0] == k and np.array_equal(g2[0][1], T) and np.array_equal(g2[0][2], inl)


def test_frame_optimization_batch_bit_exact_vs_oracle(F, O):
    sc = _scenes()
    cam = sc[0][0]
    rng = np.random.default_rng(0)
    q0, p0 = [], []
    for s in sc:
        d = rng.normal(0, 0.01, 3)
        Kx = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        u, _, vt = np.linalg.svd(s[3] @ (np.eye(3) + Kx + 0.5 * Kx @ Kx))
        q0.append(quat_wxyz(u @ vt))
        p0.append(s[4] + rng.normal(0, 0.05, 3))
    ps = F.PoseStage(cam, max_batch=len(sc), capacity=1024)
    got = ps.FrameOptimization([s[1] for s in sc], [s[2] for s in sc], q0, p0)
    for f, s in enumerate(sc):
        k, q, p, inl = O.frame_optimization(cam, s[1], s[2], q0[f], p0[f])
        assert got[f][0] == k and np.array_equal(got[f][1], q) and np.array_equal(got[f][2], p) and np.array_equal(got[f][3], inl), f
    # the result is the scene's pose (frame 1: 300 points, 40 outliers)
    w, x, y, z = got[1][1]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    assert np.abs(R - sc[1][3]).max() < 2e-3 and np.abs(got[1][2] - sc[1][4]).max() < 3e-2
    assert got[1][3][sc[1][5]].sum() <= 4
    # another gate
    g2 = ps.FrameOptimization([sc[6][1]], [sc[6][2]], [q0[6]], [p0[6]], chi2_threshold=9.21)
    k, q, p, inl = O.frame_optimization(cam, sc[6][1], sc[6][2], q0[6], p0[6], chi2_threshold=9.21)
    assert g2[0][0] == k and np.array_equal(g2[0][1], q) and np.array_equal(g2[0][3], inl)


@pytest.mark.parametrize("name", __import__("conftest").POSE_GOLDEN)
def test_pose_stage_vs_the_independent_numpy_restatement(F, O, name):
    """the HIP pose stage against the numpy restatement's fixtures (tests/golden/make_pose_golden.py) and, for the stereo edges
    (EdgeStereoSE3ProjectXYZOnlyPose, src/g2o_optimization.cc:235-260), against the oracle bit for bit"""
    from conftest import check_pose_golden, golden
    g = golden(name)
    cam, n_mono, n = tuple(g["cam"]), int(g["n_mono"]), len(g["Xw"])
    gate = [float(v) for v in g["gate"]]
    ps = F.PoseStage(cam, max_batch=2, capacity=512)
    fs = ps.FrameOptimizationStereo(float(g["bf"]), [g["Xw"]], [g["obs"]], [n_mono], [g["q0"]], [g["p0"]], *gate)[0]
    of = O.frame_optimization_stereo(cam, float(g["bf"]), g["Xw"], g["obs"], n_mono, g["q0"], g["p0"], *gate)
    assert fs[0] == of[0] and np.array_equal(fs[1], of[1]) and np.array_equal(fs[2], of[2]) and np.array_equal(fs[3], of[3])
    pnp = None
    if n_mono == n:
        fm = ps.FrameOptimization([g["Xw"]], [g["obs"][:, :2]], [g["q0"]], [g["p0"]], chi2_threshold=gate[0])[0]
        assert fm[0] == fs[0] and np.array_equal(fm[1], fs[1]) and np.array_equal(fm[3], fs[3])      # the mono entry point is the same computation
        pnp = ps.SolvePnPWithCV([g["Xw"]], [g["obs"][:, :2]], seed=int(g["pnp_seed"]))[0]
    check_pose_golden(g, fs, pnp)


def test_frame_optimization_stereo_batch_of_ragged_frames_vs_oracle(F, O):
    """two frames of different mono / stereo mixes in one call (the kernel's row bookkeeping), bit-exact vs the oracle"""
    from conftest import golden
    a, b = golden("pose_stereo_a.npz"), golden("pose_stereo_b.npz")
    cam = tuple(a["cam"])
    ps = F.PoseStage(cam, max_batch=2, capacity=300)
    got = ps.FrameOptimizationStereo(float(a["bf"]), [a["Xw"], b["Xw"]], [a["obs"], b["obs"]], [int(a["n_mono"]), int(b["n_mono"])],
                                     [a["q0"], b["q0"]], [a["p0"], b["p0"]], 10.0, 75.0)       # configs/configs_aqua.yaml:42-43
    for f, g in enumerate((a, b)):
        of = O.frame_optimization_stereo(cam, float(g["bf"]), g["Xw"], g["obs"], int(g["n_mono"]), g["q0"], g["p0"], 10.0, 75.0)
        assert got[f][0] == of[0] and np.array_equal(got[f][1], of[1]) and np.array_equal(got[f][2], of[2]) and np.array_equal(got[f][3], of[3]), f


def test_pose_stage_argument_errors(F):
    ps = F.PoseStage((400.0, 400.0, 320.0, 240.0), max_batch=2, capacity=64)
    with pytest.raises(RuntimeError):
        ps.SolvePnPWithCV([np.zeros((10, 3))] * 3, [np.zeros((10, 2))] * 3)        # more frames than max_batch
    with pytest.raises(RuntimeError):
        ps.SolvePnPWithCV([np.zeros((65, 3))], [np.zeros((65, 2))])                # more observations than capacity
