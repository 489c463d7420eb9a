"""ctypes binding of the CPU oracle (liburf_oracle.so) -- TEST INFRASTRUCTURE.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Builds the library on first use if it is missing (gcc is in the image).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# URF_ORACLE_SO: the sanitizer build (`make -C oracle asan`), picked by the sanitizer test's subprocess only
_SO = os.environ.get("URF_ORACLE_SO") or os.path.join(_HERE, "liburf_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("sp_oracle.c", "sg_oracle.c", "ransac_oracle.c", "cvransac_oracle.c", "cam_oracle.c", "map_oracle.c", "pnp_oracle.c",
                                              "urf_oracle.h", "oracle_math.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    return _SO


class SPConfig(C.Structure):
    _fields_ = [("max_keypoints", C.c_int), ("keypoint_threshold", C.c_double), ("remove_borders", C.c_int)]


class SGConfig(C.Structure):
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int),
                ("matching_threshold", C.c_double), ("sinkhorn_iterations", C.c_int)]


class RansacConfig(C.Structure):
    """(iterations, sigma, seed, confidence): confidence <= 0 (default) = every hypothesis counts"""
    _fields_ = [("iterations", C.c_int), ("sigma", C.c_float), ("seed", C.c_uint32), ("confidence", C.c_float),
                ("stage", C.c_int)]      # stage 1 = the restatement of OpenCV 4.2's cv::findFundamentalMat (cvransac_oracle.c)


SIGMA_3PX = float(np.float32(3.0 / np.sqrt(3.841)))   # the 3 px gate of the reference's cv::findFundamentalMat call


def ref_ransac(iterations=200, seed=0):
    """the outlier stage's defaults = the parameters of the reference call
    cv::findFundamentalMat(..., cv::FM_RANSAC, 3, 0.99, mask), src/point_matching.cc:50"""
    return RansacConfig(iterations, SIGMA_3PX, seed, 0.99)


class DMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int), ("trainIdx", C.c_int), ("distance", C.c_float)]


_lib = None


def threads():
    """OpenMP threads the oracle uses: the cores this process may run on -- its affinity mask capped by the cgroup's CPU
    quota (a container can see 256 cores and own 16 of them) --, <= 32."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_NUM_THREADS", str(threads()))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        _lib = C.CDLL(_SO)
        _lib.o_exp.restype = C.c_float
        _lib.o_exp.argtypes = [C.c_float]
        _lib.o_log.restype = C.c_float
        _lib.o_log.argtypes = [C.c_float]
        _lib.o_wave_sum.restype = C.c_float
        _lib.oransac_find_F.restype = C.c_float
        _lib.oransac_find_F_sets.restype = C.c_float
        _lib.oransac_minimal_sets.restype = None
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


SP_CHANNELS = [64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256]
SP_SCALE = [1, 2, 2, 4, 4, 8, 8, 8, 8, 8, 8, 8]  # output is H/scale x W/scale


def sp_dense(blob, img, want_layers=False):
    """returns dict(scores, heat, desc[, layers]) for a u8 HxW image."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    Hc, Wc = H // 8, W // 8
    scores = np.zeros((Hc * 8, Wc * 8), np.float32)
    heat = np.zeros((Hc * 8, Wc * 8), np.float32)
    desc = np.zeros((Hc, Wc, 256), np.float32)
    layers = None
    lp = None
    if want_layers:
        layers = []
        h, w = H, W
        dims = []
        for i in range(12):
            if i in (1, 3, 5):
                h, w = h // 2, w // 2
            dims.append((h, w))
            layers.append(np.zeros((h, w, SP_CHANNELS[i]), np.float32))
        arr = (C.c_void_p * 12)(*[l.ctypes.data for l in layers])
        lp = arr
    rc = lib().osp_dense(_p(blob), _p(img), H, W, C.c_size_t(img.strides[0]), _p(scores), _p(heat), _p(desc), lp)
    assert rc == 0, rc
    out = dict(scores=scores, heat=heat, desc=desc)
    if want_layers:
        out["layers"] = layers
    return out


def sp_nms(heat):
    heat = np.ascontiguousarray(heat, np.float32)
    out = np.zeros_like(heat)
    lib().osp_simple_nms(_p(heat), heat.shape[0], heat.shape[1], _p(out))
    return out


def sp_postprocess(scores, desc, cfg, mask=None, cap=None):
    Hs, Ws = scores.shape
    Hc, Wc = desc.shape[:2]
    cap = cap or Hs * Ws
    feat = np.zeros((cap, 259), np.float64)  # column-major 259 x cap
    K = C.c_int(0)
    idx = np.zeros(cap, np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    rc = lib().osp_postprocess(_p(scores), Hs, Ws, _p(desc), Hc, Wc, _p(mask),
                               C.c_size_t(mask.strides[0] if mask is not None else 0),
                               C.byref(cfg), _p(feat), cap, C.byref(K), _p(idx))
    assert rc == 0, rc
    return feat[:K.value].copy(), idx[:K.value].copy()


def sp_infer(blob, cfg, img, mask=None, cap=4096):
    """SuperPoint::infer.  Returns feat [K][259] (row j = column j of the 259xK matrix)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    cap = max(cap, 1)
    if cfg.max_keypoints < 0:
        cap = (H // 8) * (W // 8) * 64
    feat = np.zeros((cap, 259), np.float64)
    K = C.c_int(0)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    rc = lib().osp_infer(_p(blob), C.byref(cfg), _p(img), H, W, C.c_size_t(img.strides[0]), _p(mask),
                         C.c_size_t(mask.strides[0] if mask is not None else 0), _p(feat), cap, C.byref(K))
    assert rc == 0, rc
    return feat[:K.value].copy()


def sg_normalize(feat, width, height):
    feat = np.ascontiguousarray(feat, np.float64)
    out = np.zeros_like(feat)
    lib().osg_normalize_keypoints(_p(feat), feat.shape[0], width, height, _p(out))
    return out


def sg_graph(blob, iters, f0, f1, want_final=False):
    f0 = np.ascontiguousarray(f0, np.float64)
    f1 = np.ascontiguousarray(f1, np.float64)
    n0, n1 = f0.shape[0], f1.shape[0]
    Z = np.zeros((n0 + 1, n1 + 1), np.float32)
    m0 = np.zeros((n0, 256), np.float32) if want_final else None
    m1 = np.zeros((n1, 256), np.float32) if want_final else None
    rc = lib().osg_graph(_p(blob), iters, _p(f0), n0, _p(f1), n1, _p(Z), _p(m0), _p(m1))
    assert rc == 0, rc
    return (Z, m0, m1) if want_final else Z


def sg_decode(Z, thresh):
    Z = np.ascontiguousarray(Z, np.float32)
    h, w = Z.shape
    i0 = np.zeros(h - 1, np.int32)
    i1 = np.zeros(w - 1, np.int32)
    m0 = np.zeros(h - 1, np.float64)
    m1 = np.zeros(w - 1, np.float64)
    lib().osg_decode(_p(Z), h, w, C.c_double(thresh), _p(i0), _p(i1), _p(m0), _p(m1))
    return i0, i1, m0, m1


def sg_infer(blob, cfg, f0, f1):
    f0 = np.ascontiguousarray(f0, np.float64)
    f1 = np.ascontiguousarray(f1, np.float64)
    n0, n1 = f0.shape[0], f1.shape[0]
    i0 = np.zeros(n0, np.int32)
    i1 = np.zeros(n1, np.int32)
    m0 = np.zeros(n0, np.float64)
    m1 = np.zeros(n1, np.float64)
    Z = np.zeros((n0 + 1, n1 + 1), np.float32)
    rc = lib().osg_infer(_p(blob), C.byref(cfg), _p(f0), n0, _p(f1), n1, _p(i0), _p(i1), _p(m0), _p(m1), _p(Z))
    assert rc == 0, rc
    return i0, i1, m0, m1, Z


def opencv42_ransac():
    """the outlier stage as the reference's own call: cv::findFundamentalMat(..., cv::FM_RANSAC, 3, 0.99, mask) restated"""
    return RansacConfig(0, 0.0, 0, 0.99, 1)


def cv_find_fundamental_mask(p0, p1, thresh=3.0, confidence=0.99):
    """OpenCV 4.2's findFundamentalMat(FM_RANSAC) restated: -> inlier mask [n] (uint8)"""
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    n = p0.shape[0]
    mask = np.zeros(max(n, 1), np.uint8)
    lib().ocv_find_fundamental_mask(_p(p0), _p(p1), n, C.c_double(thresh), C.c_double(confidence), _p(mask))
    return mask[:n]


def ransac_find_F(p0, p1, cfg):
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    n = p0.shape[0]
    inl = np.zeros(n, np.uint8)
    F = np.zeros(9, np.float32)
    s = lib().oransac_find_F(_p(p0), _p(p1), n, C.byref(cfg), _p(inl), _p(F))
    return float(s), inl, F.reshape(3, 3)


def minimal_sets(sampler, seed, n, iterations):
    """sampler 0: counter hash; 1: the C library's srand(seed)/rand() stream the reference draws from"""
    sets = np.zeros((iterations, 8), np.int32)
    lib().oransac_minimal_sets(int(sampler), C.c_uint32(seed), int(n), int(iterations), _p(sets))
    return sets


def ransac_find_F_sets(p0, p1, cfg, sets):
    p0 = np.ascontiguousarray(p0, np.float32)
    p1 = np.ascontiguousarray(p1, np.float32)
    sets = np.ascontiguousarray(sets, np.int32)
    assert sets.shape == (cfg.iterations, 8)
    n = p0.shape[0]
    inl = np.zeros(n, np.uint8)
    F = np.zeros(9, np.float32)
    s = lib().oransac_find_F_sets(_p(p0), _p(p1), n, C.byref(cfg), _p(sets), _p(inl), _p(F))
    return float(s), inl, F.reshape(3, 3)


def match_points(sg_blob, cfg, rcfg, f0, f1, outlier_rejection=True):
    f0 = np.ascontiguousarray(f0, np.float64)
    f1 = np.ascontiguousarray(f1, np.float64)
    n0, n1 = f0.shape[0], f1.shape[0]
    cap = max(n0, 1)
    out = (DMatch * cap)()
    n = lib().omatch_points(_p(sg_blob), C.byref(cfg), C.byref(rcfg), _p(f0), n0, _p(f1), n1,
                            int(bool(outlier_rejection)), out, cap)
    return [(out[i].queryIdx, out[i].trainIdx, out[i].distance) for i in range(n)]


def fma_gemm(A, B, C0=None):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    M, K = A.shape
    N = B.shape[1]
    out = np.zeros((M, N), np.float32)
    if C0 is not None:
        C0 = np.ascontiguousarray(C0, np.float32)
    lib().o_fma_gemm(_p(A), _p(B), _p(C0), M, N, K, _p(out))
    return out


class EpiConfig(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("sigma", C.c_float), ("iterations", C.c_int), ("seed", C.c_uint32),
                ("sampler", C.c_int)]


def epi_reconstruct(K, keys1, keys2, matches12, sigma=1.0, iterations=200, seed=0, sampler=0, sets=None):
    """EpipolarGeometry::reconstruct.  Returns (ok, T21[4,4], P3D[n1,3], tri[n1], model, (SH, SF)).
    sampler 1 = the reference's rand() stream; sets = explicit minimal sets [iterations, 8]."""
    cfg = EpiConfig((C.c_float * 9)(*np.asarray(K, np.float32).reshape(-1)), sigma, iterations, seed, sampler)
    k1 = np.ascontiguousarray(keys1, np.float32)
    k2 = np.ascontiguousarray(keys2, np.float32)
    m = np.ascontiguousarray(matches12, np.int32)
    n1, n2 = k1.shape[0], k2.shape[0]
    T = np.zeros(16, np.float32)
    P = np.zeros((n1, 3), np.float32)
    tri = np.zeros(n1, np.uint8)
    model = C.c_int(-1)
    sc = np.zeros(2, np.float32)
    if sets is not None:
        sets = np.ascontiguousarray(sets, np.int32)
        assert sets.shape == (iterations, 8)
        ok = lib().oepi_reconstruct_sets(C.byref(cfg), _p(k1), n1, _p(k2), n2, _p(m), _p(sets), _p(T), _p(P), _p(tri),
                                         C.byref(model), _p(sc))
    else:
        ok = lib().oepi_reconstruct(C.byref(cfg), _p(k1), n1, _p(k2), n2, _p(m), _p(T), _p(P), _p(tri), C.byref(model), _p(sc))
    return bool(ok), T.reshape(4, 4), P, tri, model.value, (float(sc[0]), float(sc[1]))


class CamConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("distortion_type", C.c_int), ("K", C.c_double * 9),
                ("D", C.c_double * 14), ("n_dist", C.c_int), ("R", C.c_double * 9), ("P", C.c_double * 9)]


def cam_config(width, height, K, D, P=None, R=None, distortion_type=0):
    c = CamConfig()
    c.width, c.height, c.distortion_type = int(width), int(height), int(distortion_type)
    K = np.asarray(K, np.float64).reshape(9)
    P = K if P is None else np.asarray(P, np.float64).reshape(3, -1)[:, :3].reshape(9)
    R = np.eye(3).reshape(9) if R is None else np.asarray(R, np.float64).reshape(9)
    D = np.asarray(D, np.float64).reshape(-1)
    for i in range(9):
        c.K[i], c.P[i], c.R[i] = K[i], P[i], R[i]
    for i in range(14):
        c.D[i] = D[i] if i < D.size else 0.0
    c.n_dist = int(min(D.size, 14))
    return c


def cam_init_maps(cfg):
    """src/camera.cc:69-85 -> (map1, map2) float32 [H, W]"""
    m1 = np.empty((cfg.height, cfg.width), np.float32)
    m2 = np.empty_like(m1)
    rc = lib().ocam_init_maps(C.byref(cfg), _p(m1), _p(m2))
    if rc != 0:
        raise ValueError("singular P*R")
    return m1, m2


def cam_remap(img, map1, map2):
    """src/camera.cc:116-118 (cv::remap, INTER_LINEAR, constant border 0)"""
    img = np.ascontiguousarray(img, np.uint8)
    map1 = np.ascontiguousarray(map1, np.float32)
    map2 = np.ascontiguousarray(map2, np.float32)
    oh, ow = map1.shape
    out = np.empty((oh, ow), np.uint8)
    lib().ocam_remap(_p(img), img.shape[0], img.shape[1], C.c_size_t(img.strides[0]), _p(map1), _p(map2), oh, ow,
                     _p(out), C.c_size_t(out.strides[0]))
    return out


class SbpConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("image_width", C.c_double), ("image_height", C.c_double), ("pose", C.c_double * 16), ("thr", C.c_int)]


def sbp_config(fx, fy, cx, cy, width, height, pose, thr):
    c = SbpConfig(fx, fy, cx, cy, width, height)
    for i, v in enumerate(np.asarray(pose, np.float64).reshape(16)):
        c.pose[i] = v
    c.thr = int(thr)
    return c


def search_by_projection(cfg, feat, mp_pos, mp_desc, occupied=None, mp_valid=None):
    """src/mapping.cc:667-735 -> best keypoint index per map point (-1 = rejected)"""
    f = np.ascontiguousarray(feat, np.float64)
    pos = np.ascontiguousarray(mp_pos, np.float64)
    desc = np.ascontiguousarray(mp_desc, np.float64)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    val = None if mp_valid is None else np.ascontiguousarray(mp_valid, np.uint8)
    out = np.full(pos.shape[0], -2, np.int32)
    lib().osbp_search(C.byref(cfg), _p(f), f.shape[0], _p(occ), _p(pos), _p(desc), _p(val), pos.shape[0], _p(out))
    return out


class PnpConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("iterations", C.c_int),
                ("reprojection_error", C.c_double), ("confidence", C.c_double), ("seed", C.c_uint32)]


def solve_pnp_ransac(cam, obj, img, iterations=100, reprojection_error=20.0, confidence=0.99, seed=0):
    """SolvePnPWithCV (src/g2o_optimization.cc:323-377) -> (n_inliers, Twc[4,4], inlier flags[n])"""
    cfg = PnpConfig(*[float(v) for v in cam], iterations, reprojection_error, confidence, seed)
    obj = np.ascontiguousarray(obj, np.float32)
    img = np.ascontiguousarray(img, np.float32)
    n = obj.shape[0]
    pose = np.zeros(16, np.float64)
    inl = np.zeros(max(n, 1), np.uint8)
    k = lib().opnp_solve_ransac(C.byref(cfg), _p(obj), _p(img), n, _p(pose), _p(inl))
    return int(k), pose.reshape(4, 4), inl[:n]


class PoseOptConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("chi2_threshold", C.c_double)]


def frame_optimization(cam, Xw, obs, q_wc, p_wc, inlier=None, chi2_threshold=5.991):
    """FrameOptimization (src/g2o_optimization.cc:179-321) -> (n - outliers, q_wc[4] (w,x,y,z), p_wc[3], inlier flags)"""
    cfg = PoseOptConfig(*[float(v) for v in cam], chi2_threshold)
    Xw = np.ascontiguousarray(Xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    n = Xw.shape[0]
    q = np.array(q_wc, np.float64).copy()
    p = np.array(p_wc, np.float64).copy()
    inl = np.ones(max(n, 1), np.uint8) if inlier is None else np.ascontiguousarray(inlier, np.uint8).copy()
    k = lib().oframe_optimization(C.byref(cfg), _p(Xw), _p(obs), n, _p(q), _p(p), _p(inl))
    return int(k), q, p, inl[:n]


class PoseOptStereoConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("chi2_mono", C.c_double), ("chi2_stereo", C.c_double)]


def frame_optimization_stereo(cam, bf, Xw, obs, n_mono, q_wc, p_wc, chi2_mono=5.991, chi2_stereo=7.815):
    """FrameOptimization with stereo edges (src/g2o_optimization.cc:179-321): rows [0, n_mono) of obs [n, 3] are mono
    observations (u, v, unused), the rest stereo (u, v, u_right) -> (n - outliers, q_wc, p_wc, inlier flags)"""
    cfg = PoseOptStereoConfig(*[float(v) for v in cam], float(bf), chi2_mono, chi2_stereo)
    Xw = np.ascontiguousarray(Xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    n = Xw.shape[0]
    assert obs.shape == (n, 3) and 0 <= n_mono <= n
    q = np.ascontiguousarray(q_wc, np.float64).copy()
    p = np.ascontiguousarray(p_wc, np.float64).copy()
    inl = np.ones(max(n, 1), np.uint8)
    lib().oframe_optimization_stereo.restype = C.c_int
    k = lib().oframe_optimization_stereo(C.byref(cfg), _p(Xw), _p(obs), int(n_mono), int(n - n_mono), _p(q), _p(p), _p(inl))
    return int(k), q, p, inl[:n]
