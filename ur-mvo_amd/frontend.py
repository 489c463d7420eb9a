"""Python mirror of the reference's C++ front-end classes over the C ABI.

Same class / method names and argument meaning as the reference headers
(include/super_point.h:20-33, include/super_glue.h:20-33,
include/point_matching.h:7-25 of UR-MVO) so the parity tests read like tests of
the reference.  numpy arrays stand in for cv::Mat / Eigen: a feature matrix is
an array of shape [K, 259] whose row j is column j of the reference's
Eigen::Matrix<double,259,Dynamic> (same memory layout).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DMatch, SGConfig, SPConfig, check

CAP = 1024
MATCH_DTYPE = np.dtype([("queryIdx", np.int32), ("trainIdx", np.int32), ("distance", np.float32)])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SuperPointConfig:
    """include/read_configs.h:9-18"""

    def __init__(self, max_keypoints=1000, keypoint_threshold=0.0005, remove_borders=4, onnx_file="",
                 engine_file="", dla_core=-1, input_tensor_names=("input",),
                 output_tensor_names=("scores", "descriptors")):
        self.max_keypoints = max_keypoints
        self.keypoint_threshold = keypoint_threshold
        self.remove_borders = remove_borders
        self.onnx_file = onnx_file
        self.engine_file = engine_file
        self.dla_core = dla_core
        self.input_tensor_names = list(input_tensor_names)
        self.output_tensor_names = list(output_tensor_names)


class SuperGlueConfig:
    """include/read_configs.h:20-29"""

    def __init__(self, image_width=640, image_height=512, matching_threshold=0.5, onnx_file="", engine_file="",
                 dla_core=-1):
        self.image_width = image_width
        self.image_height = image_height
        self.matching_threshold = matching_threshold
        self.onnx_file = onnx_file
        self.engine_file = engine_file
        self.dla_core = dla_core


class SuperPoint:
    """SuperPoint (include/super_point.h:20-33)."""

    def __init__(self, super_point_config, max_height=0, max_width=0, max_batch=1, device=0, precision=0,
                 guard_delta=0.0, guard_ulps=0.0):
        """precision: 0 exact, 1 fast, 2 guarded fast, 3 strict parity (= exact for SuperPoint); guard_delta / guard_ulps: the
        guarded mode's error model (urf_sp_config; 0 = the built-in constants)"""
        self.cfg = super_point_config
        self._c = SPConfig(super_point_config.max_keypoints, super_point_config.keypoint_threshold,
                           super_point_config.remove_borders, max_height, max_width, max_batch, device, precision,
                           guard_delta, guard_ulps)
        self._h = C.c_void_p()
        self._built = False
        check(_lib.lib().urf_sp_create(C.byref(self._c), C.byref(self._h)), "urf_sp_create")

    def build(self, blob=None):
        """build(): weights from `blob` (f32 container) or from cfg.engine_file."""
        L = _lib.lib()
        if blob is not None:
            blob = np.ascontiguousarray(blob, np.float32)
            rc = L.urf_sp_build(self._h, _p(blob), C.c_size_t(blob.size))
        else:      # the reference's flow: the cached engine_file if it exists, else onnx_file -> build -> write the cache
            rc = L.urf_sp_build_config(self._h, self.cfg.engine_file.encode(), self.cfg.onnx_file.encode())
        self._built = rc == 0
        return self._built

    def stream_ptr(self):
        """the handle's hipStream_t (for torch.cuda.ExternalStream / event ordering)"""
        return _lib.lib().urf_sp_stream(self._h)

    def result_stream_ptr(self):
        """the hipStream_t on which a call's slots become final (== stream_ptr() except in the guarded fast mode)"""
        return _lib.lib().urf_sp_result_stream(self._h)

    def infer(self, image, mask=None):
        """infer(image, mask, features) -> features [K,259] (or None on failure)."""
        image = np.asarray(image)
        assert image.dtype == np.uint8 and image.ndim == 2
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        feat = np.zeros((CAP, 259), np.float64)
        K = C.c_int(0)
        mp, ms = None, 0
        if mask is not None and mask.size:
            mask = np.ascontiguousarray(mask, np.uint8)
            mp, ms = _p(mask), mask.strides[0]
        rc = _lib.lib().urf_sp_infer(self._h, _p(image), image.shape[0], image.shape[1],
                                     C.c_size_t(image.strides[0]), mp, C.c_size_t(ms), _p(feat), CAP, C.byref(K))
        if rc != 0:
            return None
        return feat[:K.value].copy()

    def infer_batch(self, images):
        B = len(images)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        H, W = imgs[0].shape
        ptrs = (C.c_void_p * B)(*[i.ctypes.data for i in imgs])
        feat = np.zeros((B, CAP, 259), np.float64)
        K = (C.c_int * B)()
        check(_lib.lib().urf_sp_infer_batch(self._h, B, ptrs, H, W, C.c_size_t(W), _p(feat), CAP, K), "infer_batch")
        return [feat[b, :K[b]].copy() for b in range(B)]

    def infer_device(self, d_imgs_ptr, B, rows, cols, d_slots_ptr):
        check(_lib.lib().urf_sp_infer_device(self._h, B, C.c_void_p(d_imgs_ptr), rows, cols, C.c_void_p(d_slots_ptr)),
              "urf_sp_infer_device")

    def sync(self):
        check(_lib.lib().urf_sp_sync(self._h), "urf_sp_sync")

    def near_tie_reruns(self):
        """guarded fast mode, since build(): frames redone whole in the exact mode, frames processed, frames whose top-k cut was
        resolved per candidate (and how many candidates), frames flagged by the threshold band / an NMS near-tie / too many
        candidates at the cut (those are the ones redone whole)"""
        v = (C.c_ulonglong * 8)()
        check(_lib.lib().urf_sp_near_tie_reruns(self._h, v, 8), "urf_sp_near_tie_reruns")
        return dict(redone=int(v[0]), frames=int(v[1]), cut_resolved=int(v[2]), threshold=int(v[3]), nms=int(v[4]),
                    cut_overflow=int(v[5]), candidates=int(v[6]))

    def calibrate_guard(self, images=None, device_ptr=None, B=0, rows=0, cols=0):
        """guarded fast mode: check (and widen where needed) the guard's error model on representative frames -- a list of
        u8 images, or a device pointer to B frames.  Returns delta / c the frames needed and the constants now in use."""
        out = (C.c_double * 4)()
        if images is not None:
            imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
            H, W = imgs[0].shape
            ptrs = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
            check(_lib.lib().urf_sp_calibrate_guard(self._h, len(imgs), ptrs, H, W, C.c_size_t(W), out), "urf_sp_calibrate_guard")
        else:
            check(_lib.lib().urf_sp_calibrate_guard_device(self._h, B, C.c_void_p(device_ptr), rows, cols, out),
                  "urf_sp_calibrate_guard_device")
        return dict(delta_needed=out[0], c_needed=out[1], delta=out[2], c=out[3])

    def debug_tensor(self, which, shape):
        out = np.zeros(shape, np.float32)
        check(_lib.lib().urf_sp_debug_tensor(self._h, which, _p(out), C.c_size_t(out.size)), "debug_tensor")
        return out

    def stage_ms(self, previous=False, age=None):
        """HIP-event stage times of the call `age` calls ago (0 = latest, up to 3; it must
        have completed).  previous=True is age 1."""
        ms = (C.c_float * 32)()
        a = age if age is not None else (1 if previous else 0)
        n = check(_lib.lib().urf_sp_stage_ms_age(self._h, ms, 32, a), "stage_ms")
        return [ms[i] for i in range(n)]

    def save_engine(self, blob=None):
        if self.cfg.engine_file and blob is not None:
            blob = np.ascontiguousarray(blob, np.float32)
            check(_lib.lib().urf_weights_save(self.cfg.engine_file.encode(), 1, _p(blob), C.c_size_t(blob.size)))

    def deserialize_engine(self):
        return self.build(None)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().urf_sp_destroy(self._h)
                self._h = None
        except Exception:
            pass


SP_STAGES = ["upload", "conv1a+1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa|Da",
             "convPb", "convDb", "softmax", "nms", "select", "desc_norm", "sample", "download"]
PM_STAGES = ["prep", "kenc", "gnn", "final+score", "sinkhorn", "decode", "ransac", "attn(in gnn)", "exact redo of flagged pairs"]


class _PM:
    def __init__(self, cfg, max_pairs=1, device=0, sinkhorn_iterations=100, ransac_iterations=200,
                 ransac_sigma=0.0, ransac_seed=0, precision=0, ransac_threshold_px=0.0, ransac_confidence=0.0,
                 redo_flagged_pairs=0, guard_margin=0.0, outlier_stage=0, sinkhorn_residual_bound=0.0, calibrate_pairs=0, redo_merge=0, redo_shared_engine=0,
                 audit_period=0):
        """outlier stage: all-zero = the reference call's parameters (3 px, confidence 0.99, src/point_matching.cc:50);
        ransac_sigma > 0 states the gate like EpipolarGeometry does, ransac_confidence < 0 makes every hypothesis count.
        precision: 0 exact, 1 fast, 2 guarded fast (flagged pairs reported), 3 strict parity (flagged pairs redone in the
        exact mode); redo_flagged_pairs / guard_margin: urf_sg_config (0 = the mode's defaults); outlier_stage=1: OpenCV
        4.2's cv::findFundamentalMat(FM_RANSAC, 3, 0.99) restated instead of the in-tree 8-point search"""
        self.cfg = cfg
        self._c = SGConfig(cfg.image_width, cfg.image_height, cfg.matching_threshold, sinkhorn_iterations,
                           max_pairs, device, ransac_iterations, ransac_sigma, ransac_seed, precision,
                           ransac_threshold_px, ransac_confidence, redo_flagged_pairs, guard_margin, outlier_stage,
                           sinkhorn_residual_bound, calibrate_pairs, redo_merge, redo_shared_engine, audit_period)
        self._h = C.c_void_p()
        check(_lib.lib().urf_pm_create(C.byref(self._c), C.byref(self._h)), "urf_pm_create")

    def build(self, blob=None):
        L = _lib.lib()
        if blob is not None:
            blob = np.ascontiguousarray(blob, np.float32)
            rc = L.urf_pm_build(self._h, _p(blob), C.c_size_t(blob.size))
        else:
            rc = L.urf_pm_build_config(self._h, self.cfg.engine_file.encode(), self.cfg.onnx_file.encode())
        return rc == 0

    def sinkhorn_fallbacks(self):
        """how often the resident Sinkhorn launch of this handle gave up and the batch was redone with the streaming kernels"""
        return int(_lib.lib().urf_pm_sinkhorn_fallbacks(self._h))

    def sinkhorn_integrity(self):
        """integrity check of the fast Sinkhorn: dict(pairs redone, batches, bound, pairs processed) since build()"""
        v = (C.c_double * 4)()
        check(_lib.lib().urf_pm_sinkhorn_integrity(self._h, v, 4), "urf_pm_sinkhorn_integrity")
        return dict(pairs=int(v[0]), events=int(v[1]), bound=float(v[2]), seen=int(v[3]))

    def sinkhorn_residuals(self, P=1):
        """largest |column marginal - 1| of the plan the decode read, per pair of the batch handed out last"""
        f = (C.c_float * P)()
        check(_lib.lib().urf_pm_sinkhorn_residuals(self._h, f, P), "urf_pm_sinkhorn_residuals")
        return [float(f[p]) for p in range(P)]

    def near_tie_reruns(self):
        """guarded fast mode: dict(redone, pairs, threshold, runner_up) since build()"""
        v = (C.c_ulonglong * 8)()
        check(_lib.lib().urf_pm_near_tie_reruns(self._h, v, 8), "urf_pm_near_tie_reruns")
        return dict(redone=int(v[0]), pairs=int(v[1]), threshold=int(v[2]), runner_up=int(v[3]), flagged=int(v[4]))

    def calibrate_guard(self, slot_ptrs0, slot_ptrs1):
        """guarded fast mode: the fast matcher against the exact matcher on these pairs of device slots; widens the margin where
        needed.  Returns the measured difference of the log-assignments and the margin now in use."""
        P = len(slot_ptrs0)
        a0 = (C.c_void_p * P)(*slot_ptrs0)
        a1 = (C.c_void_p * P)(*slot_ptrs1)
        out = (C.c_double * 2)()
        check(_lib.lib().urf_pm_calibrate_guard(self._h, P, a0, a1, out), "urf_pm_calibrate_guard")
        return dict(z_difference=out[0], margin=out[1])

    def redo_engine_stats(self):
        """the (shared) redo engine of a strict handle: dict(passes, pairs, merged passes, handles sharing it)"""
        v = (C.c_double * 4)()
        check(_lib.lib().urf_pm_redo_engine_stats(self._h, v, 4), "urf_pm_redo_engine_stats")
        return dict(passes=int(v[0]), pairs=int(v[1]), merged=int(v[2]), sharers=int(v[3]))

    def guard_state(self):
        """dict(margin in use, largest calibrated fast-vs-exact difference, pairs the automatic calibration still wants, redo_all)"""
        v = (C.c_double * 11)()
        check(_lib.lib().urf_pm_guard_state(self._h, v, 11), "urf_pm_guard_state")
        # online_*: the by-product of every exact redo (largest fast-vs-exact difference seen on a redone pair, pairs sampled,
        # times the margin was raised for it, times it exceeded the margin its batch was guarded with); audits: unflagged pairs
        # sent through the exact engine / those whose exact index list differed from the fast one
        return dict(margin=float(v[0]), measured=float(v[1]), pairs_left=int(v[2]), redo_all=bool(v[3]),
                    online_worst=float(v[4]), online_pairs=int(v[5]), margin_raises=int(v[6]), online_violations=int(v[7]),
                    audits=int(v[8]), audit_mismatches=int(v[9]), exact_batches=int(v[10]))

    def near_tie_flags(self, P=1):
        """guard words of the pairs of the batch handed out last (0 = the pair's match set is the exact pipeline's)"""
        f = (C.c_int * P)()
        check(_lib.lib().urf_pm_near_tie_flags(self._h, f, P), "urf_pm_near_tie_flags")
        return [int(f[p]) for p in range(P)]

    def stage_ms(self):
        ms = (C.c_float * 16)()
        n = check(_lib.lib().urf_pm_stage_ms(self._h, ms, 16), "stage_ms")
        return [ms[i] for i in range(n)]

    def __del__(self):
        try:
            if self._h:
                _lib.lib().urf_pm_destroy(self._h)
                self._h = None
        except Exception:
            pass


class SuperGlue(_PM):
    """SuperGlue (include/super_glue.h:20-33)."""

    def infer(self, features0, features1, want_scores=False):
        """infer(features0, features1, indices0, indices1, mscores0, mscores1);
        features carry normalised keypoints.  Returns the four vectors (+ the
        (n0+1)x(n1+1) log-assignment when want_scores)."""
        f0 = np.ascontiguousarray(features0, np.float64)
        f1 = np.ascontiguousarray(features1, np.float64)
        n0, n1 = f0.shape[0], f1.shape[0]
        i0 = np.zeros(n0, np.int32)
        i1 = np.zeros(n1, np.int32)
        m0 = np.zeros(n0, np.float64)
        m1 = np.zeros(n1, np.float64)
        Z = np.zeros((n0 + 1, n1 + 1), np.float32) if want_scores else None
        rc = _lib.lib().urf_sg_infer(self._h, _p(f0), n0, _p(f1), n1, _p(i0), _p(i1), _p(m0), _p(m1), _p(Z))
        if rc != 0:
            return None
        return (i0, i1, m0, m1, Z) if want_scores else (i0, i1, m0, m1)


class PointMatching(_PM):
    """PointMatching (include/point_matching.h:7-25)."""

    def NormalizeKeypoints(self, features, width, height):
        f = np.ascontiguousarray(features, np.float64)
        out = np.zeros_like(f)
        _lib.lib().urf_normalize_keypoints(_p(f), f.shape[0], width, height, _p(out))
        return out

    def MatchingPoints(self, features0, features1, outlier_rejection=False):
        """-> list of (queryIdx, trainIdx, distance)."""
        f0 = np.ascontiguousarray(features0, np.float64)
        f1 = np.ascontiguousarray(features1, np.float64)
        out = (DMatch * CAP)()
        n = check(_lib.lib().urf_match(self._h, _p(f0), f0.shape[0], _p(f1), f1.shape[0],
                                       int(bool(outlier_rejection)), out, CAP), "urf_match")
        return [(out[i].queryIdx, out[i].trainIdx, out[i].distance) for i in range(n)]

    def stream_ptr(self):
        return _lib.lib().urf_pm_stream(self._h)

    def match_device_async(self, slot_ptrs0, slot_ptrs1, outlier_rejection=True):
        P = len(slot_ptrs0)
        a = (C.c_void_p * P)(*slot_ptrs0)
        b = (C.c_void_p * P)(*slot_ptrs1)
        check(_lib.lib().urf_match_device_async(self._h, P, a, b, int(bool(outlier_rejection))), "match_device_async")

    def fetch(self, P, as_arrays=False):
        """match lists of the last batch.  as_arrays=True returns one numpy structured
        array (queryIdx, trainIdx, distance) per pair without per-match Python objects."""
        out = np.zeros((P, CAP), dtype=MATCH_DTYPE)
        n = (C.c_int * P)()
        check(_lib.lib().urf_pm_fetch(self._h, P, _p(out), CAP, n), "urf_pm_fetch")
        if as_arrays:
            return [out[p, :n[p]] for p in range(P)]
        return [[(int(m[0]), int(m[1]), float(m[2])) for m in out[p, :n[p]]] for p in range(P)]

    def fetch_begin(self, P):
        """first half of fetch(): waits for the batch's fast pass and starts the exact redo of its flagged pairs (strict
        parity) on the redo engine's stream.  Returns 1 when a redo is running (the handle may take its next batch meanwhile),
        0 when the lists are final."""
        return check(_lib.lib().urf_pm_fetch_begin(self._h, P), "urf_pm_fetch_begin")

    def fetch_ready(self):
        """True when fetch_end() would not block (the oldest begun batch needed no redo, or its redo has delivered)"""
        return check(_lib.lib().urf_pm_fetch_ready(self._h), "urf_pm_fetch_ready") == 1

    def fetch_end(self, P, as_arrays=False):
        """second half of fetch(): waits for the redo of the oldest begun batch, if one runs, and returns its lists"""
        out = np.zeros((P, CAP), dtype=MATCH_DTYPE)
        n = (C.c_int * P)()
        check(_lib.lib().urf_pm_fetch_end(self._h, P, _p(out), CAP, n), "urf_pm_fetch_end")
        if as_arrays:
            return [out[p, :n[p]] for p in range(P)]
        return [[(int(m[0]), int(m[1]), float(m[2])) for m in out[p, :n[p]]] for p in range(P)]

    def sync(self):
        check(_lib.lib().urf_pm_sync(self._h), "urf_pm_sync")

    def device_results(self):
        """device pointers (matches [max_pairs][1024] x 12 B, counts [max_pairs] int) of the last batch"""
        m, n = C.c_void_p(), C.c_void_p()
        check(_lib.lib().urf_pm_device_results(self._h, C.byref(m), C.byref(n)), "urf_pm_device_results")
        return m.value, n.value

    def wait_for_sp(self, superpoint):
        check(_lib.lib().urf_pm_wait_for_sp(self._h, superpoint._h), "urf_pm_wait_for_sp")

    def let_sp_overlap_sinkhorn(self, superpoint):
        check(_lib.lib().urf_sp_wait_for_sinkhorn(superpoint._h, self._h), "urf_sp_wait_for_sinkhorn")

    def share_stream(self, superpoint):
        """run on the SuperPoint handle's HIP stream (in-order SP -> match pipeline)"""
        check(_lib.lib().urf_pm_share_stream(self._h, superpoint._h), "urf_pm_share_stream")
        self._sp_keepalive = superpoint

    def find_F(self, pts0, pts1):
        p0 = np.ascontiguousarray(pts0, np.float32)
        p1 = np.ascontiguousarray(pts1, np.float32)
        n = p0.shape[0]
        inl = np.zeros(max(n, 1), np.uint8)
        F = np.zeros(9, np.float32)
        s = C.c_float(0)
        check(_lib.lib().urf_ransac_find_F(self._h, _p(p0), _p(p1), n, _p(inl), _p(F), C.byref(s)), "find_F")
        return float(s.value), inl[:n], F.reshape(3, 3)

    def find_F_sets(self, pts0, pts1, sets):
        """_find_F over explicit minimal sets [iterations, 8] (the reference's _vSets), caller's order"""
        p0 = np.ascontiguousarray(pts0, np.float32)
        p1 = np.ascontiguousarray(pts1, np.float32)
        st = np.ascontiguousarray(sets, np.int32)
        n = p0.shape[0]
        inl = np.zeros(max(n, 1), np.uint8)
        F = np.zeros(9, np.float32)
        s = C.c_float(0)
        check(_lib.lib().urf_ransac_find_F_sets(self._h, _p(p0), _p(p1), n, _p(st), st.shape[0], _p(inl), _p(F), C.byref(s)),
              "find_F_sets")
        return float(s.value), inl[:n], F.reshape(3, 3)


class EpipolarGeometry:
    """EpipolarGeometry (include/epipolar_geometry.h:20-40 of UR-MVO): two-view
    initialisation.  Runs on the matcher's device/stream (pass a built
    PointMatching or SuperGlue object)."""

    def __init__(self, matcher, K, sigma=1.0, iterations=200, seed=0, sampler=0):
        """sampler 0: the build's counter hash; 1: the reference's rand() stream after srand(seed)"""
        self._m = matcher
        self._cfg = _lib.EpiConfig((C.c_float * 9)(*np.asarray(K, np.float32).reshape(-1)), sigma, iterations, seed, sampler)

    def reconstruct(self, vKeys1, vKeys2, vMatches12, sets=None):
        """-> (ok, T21[4,4], vP3D[n1,3], vbTriangulated[n1], model, (SH, SF)); sets: explicit minimal sets [iterations, 8]"""
        k1 = np.ascontiguousarray(vKeys1, np.float32)
        k2 = np.ascontiguousarray(vKeys2, np.float32)
        m = np.ascontiguousarray(vMatches12, np.int32)
        n1, n2 = k1.shape[0], k2.shape[0]
        T = np.zeros(16, np.float32)
        P = np.zeros((max(n1, 1), 3), np.float32)
        tri = np.zeros(max(n1, 1), np.uint8)
        model = C.c_int(-1)
        sc = np.zeros(2, np.float32)
        if sets is not None:
            st = np.ascontiguousarray(sets, np.int32)
            assert st.shape == (self._cfg.iterations, 8)
            rc = check(_lib.lib().urf_epipolar_reconstruct_sets(self._m._h, C.byref(self._cfg), _p(k1), n1, _p(k2), n2, _p(m),
                                                                _p(st), _p(T), _p(P), _p(tri), C.byref(model), _p(sc)),
                       "reconstruct_sets")
        else:
            rc = check(_lib.lib().urf_epipolar_reconstruct(self._m._h, C.byref(self._cfg), _p(k1), n1, _p(k2), n2, _p(m),
                                                           _p(T), _p(P), _p(tri), C.byref(model), _p(sc)), "reconstruct")
        return bool(rc), T.reshape(4, 4), P[:n1], tri[:n1], model.value, (float(sc[0]), float(sc[1]))


class Camera:
    """The undistortion part of Camera (include/camera.h, src/camera.cc:69-85, 116-118 of UR-MVO):
    maps built once, `UndistortImage` per frame on the GPU.

    Camera(width, height, K, D, P=None, R=None, distortion_type=0)  -- maps from the calibration, or
    Camera.from_maps(map1, map2)                                     -- OpenCV's own CV_32FC1 maps."""

    def __init__(self, width=0, height=0, K=None, D=(), P=None, R=None, distortion_type=0, device=0, _maps=None):
        self._h = C.c_void_p()
        if _maps is not None:
            m1 = np.ascontiguousarray(_maps[0], np.float32)
            m2 = np.ascontiguousarray(_maps[1], np.float32)
            assert m1.shape == m2.shape and m1.ndim == 2
            self.height, self.width = m1.shape
            check(_lib.lib().urf_cam_create_from_maps(_p(m1), _p(m2), self.width, self.height, device,
                                                      C.byref(self._h)), "urf_cam_create_from_maps")
            return
        c = _lib.CamConfig()
        c.width, c.height, c.distortion_type, c.device = int(width), int(height), int(distortion_type), int(device)
        Kf = np.asarray(K, np.float64).reshape(9)
        Pf = Kf if P is None else np.asarray(P, np.float64).reshape(3, -1)[:, :3].reshape(9)   # LEFT_P(0:3,0:3)
        Rf = np.eye(3).reshape(9) if R is None else np.asarray(R, np.float64).reshape(9)
        Df = np.asarray(D, np.float64).reshape(-1)
        for i in range(9):
            c.K[i], c.P[i], c.R[i] = Kf[i], Pf[i], Rf[i]
        for i in range(min(Df.size, 14)):
            c.D[i] = Df[i]
        c.n_dist = int(min(Df.size, 14))
        self.width, self.height = int(width), int(height)
        check(_lib.lib().urf_cam_create(C.byref(c), C.byref(self._h)), "urf_cam_create")

    @classmethod
    def from_maps(cls, map1, map2, device=0):
        return cls(device=device, _maps=(map1, map2))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:   # _lib is None at interpreter exit
            _lib.lib().urf_cam_destroy(self._h)
            self._h = C.c_void_p()

    def maps(self):
        m1 = np.empty((self.height, self.width), np.float32)
        m2 = np.empty_like(m1)
        check(_lib.lib().urf_cam_maps(self._h, _p(m1), _p(m2)), "urf_cam_maps")
        return m1, m2

    def UndistortImage(self, image):
        """src/camera.cc:116-118: u8 [rows, cols] -> u8 [height, width]"""
        img = np.asarray(image)
        assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
        out = np.empty((self.height, self.width), np.uint8)
        check(_lib.lib().urf_cam_undistort(self._h, _p(img), img.shape[0], img.shape[1], C.c_size_t(img.strides[0]),
                                           _p(out), C.c_size_t(out.strides[0])), "urf_cam_undistort")
        return out

    def undistort_device(self, d_imgs, n, rows, cols, d_out, superpoint=None):
        """n device-resident frames; with `superpoint` the remap is enqueued on that handle's stream, in
        order in front of its next infer_device()."""
        st = _lib.lib().urf_sp_stream(superpoint._h) if superpoint is not None else None
        check(_lib.lib().urf_cam_undistort_device(self._h, C.c_void_p(d_imgs), n, rows, cols, C.c_void_p(d_out),
                                                  C.c_void_p(st)), "urf_cam_undistort_device")

    def sync(self):
        check(_lib.lib().urf_cam_sync(self._h), "urf_cam_sync")


class FrameStream:
    """Batched caller of the path (urf_fe_*): what Tracking::ExtractFeatureAndMatch
    (src/tracking.cc:338-377) does per frame, for a stream of frames submitted in batches.

        fs = FrameStream(SuperPointConfig(), SuperGlueConfig(), batch=8, max_height=480, max_width=640)
        fs.build(sp_blob, sg_blob)
        fs.submit(frames[0:8]); fs.submit(frames[8:16])
        K, matches = fs.collect()          # batch 0: keypoint counts, one match array per frame
    """

    def __init__(self, sp_cfg, sg_cfg, batch=8, max_height=0, max_width=0, device=0, precision=0, matchers=2,
                 history_batches=0, outlier_rejection=True, sinkhorn_iterations=100, ransac_iterations=200,
                 ransac_sigma=0.0, ransac_seed=0, ransac_threshold_px=0.0, ransac_confidence=0.0,
                 redo_flagged_pairs=0, guard_margin=0.0, guard_delta=0.0, guard_ulps=0.0):
        c = _lib.FEConfig()
        c.sp = SPConfig(sp_cfg.max_keypoints, sp_cfg.keypoint_threshold, sp_cfg.remove_borders, max_height, max_width,
                        batch, device, precision, guard_delta, guard_ulps)
        c.sg = SGConfig(sg_cfg.image_width, sg_cfg.image_height, sg_cfg.matching_threshold, sinkhorn_iterations, batch,
                        device, ransac_iterations, ransac_sigma, ransac_seed, precision, ransac_threshold_px,
                        ransac_confidence, redo_flagged_pairs, guard_margin)
        c.batch, c.matchers, c.history_batches, c.outlier_rejection = batch, matchers, history_batches, int(outlier_rejection)
        self.batch = batch
        self._h = C.c_void_p()
        self._cam = None
        check(_lib.lib().urf_fe_create(C.byref(c), C.byref(self._h)), "urf_fe_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.lib().urf_fe_destroy(self._h)
            self._h = C.c_void_p()

    def build(self, sp_blob, sg_blob):
        a = np.ascontiguousarray(sp_blob, np.float32)
        b = np.ascontiguousarray(sg_blob, np.float32)
        return _lib.lib().urf_fe_build(self._h, _p(a), C.c_size_t(a.size), _p(b), C.c_size_t(b.size)) == 0

    def set_camera(self, camera):
        self._cam = camera     # borrowed by the native handle: keep it alive
        check(_lib.lib().urf_fe_set_camera(self._h, camera._h if camera is not None else None,
                                           camera.height if camera is not None else 0,
                                           camera.width if camera is not None else 0), "urf_fe_set_camera")

    def submit(self, frames, ref=None):
        """frames: [n, rows, cols] u8 (n <= batch); ref: None or n global frame indices (-1 = predecessor)"""
        fr = np.ascontiguousarray(frames, np.uint8)
        assert fr.ndim == 3
        r = None if ref is None else np.ascontiguousarray(ref, np.int64)
        check(_lib.lib().urf_fe_submit(self._h, _p(fr), fr.shape[0], fr.shape[1], fr.shape[2], C.c_size_t(fr.strides[1]),
                                       C.c_size_t(fr.strides[0]), _p(r)), "urf_fe_submit")

    def collect(self, want_features=False):
        """-> (K[n], [match array per frame]) or (K, matches, [features [K_j, 259] per frame])"""
        n = C.c_int(0)
        K = np.zeros(self.batch, np.int32)
        nm = np.zeros(self.batch, np.int32)
        m = np.zeros((self.batch, CAP), MATCH_DTYPE)
        feat = np.zeros((self.batch, CAP, 259), np.float64) if want_features else None
        check(_lib.lib().urf_fe_collect(self._h, C.byref(n), _p(K), _p(m), CAP, _p(nm), _p(feat)), "urf_fe_collect")
        out = [m[j, :nm[j]].copy() for j in range(n.value)]
        if want_features:
            return K[:n.value].copy(), out, [feat[j, :K[j]].copy() for j in range(n.value)]
        return K[:n.value].copy(), out

    def ready(self):
        """True when collect() would return without waiting for the GPU"""
        return check(_lib.lib().urf_fe_ready(self._h), "urf_fe_ready") == 1

    def in_flight(self):
        return _lib.lib().urf_fe_in_flight(self._h)

    def max_in_flight(self):
        """batches submit() accepts before one must be collected: min(matchers + 5, 3 matchers + 2)"""
        return _lib.lib().urf_fe_max_in_flight(self._h)

    def frame_resident(self, frame):
        """may the next submit() name global frame `frame` as a reference? (its slot is still in the ring)"""
        return check(_lib.lib().urf_fe_frame_resident(self._h, C.c_long(int(frame))), "urf_fe_frame_resident") == 1


def SearchByProjection(camera_fxfycxcy, image_size, pose, features, mappoint_positions, mappoint_descriptors, thr,
                       occupied=None, mappoint_valid=None, d_slot=None, device=0):
    """Mapping::SearchByProjection (src/mapping.cc:667-735): -> int32[M], accepted keypoint index or -1.
    features: [K, 259] f64, or pass d_slot (device slot pointer) with features = K to read them on the GPU."""
    c = _lib.SbpConfig(*[float(v) for v in camera_fxfycxcy], float(image_size[0]), float(image_size[1]))
    for i, v in enumerate(np.asarray(pose, np.float64).reshape(16)):
        c.pose[i] = v
    c.thr, c.device = int(thr), int(device)
    pos = np.ascontiguousarray(mappoint_positions, np.float64)
    desc = np.ascontiguousarray(mappoint_descriptors, np.float64)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    val = None if mappoint_valid is None else np.ascontiguousarray(mappoint_valid, np.uint8)
    out = np.full(pos.shape[0], -2, np.int32)
    if d_slot is not None:
        check(_lib.lib().urf_search_by_projection_slot(C.byref(c), C.c_void_p(d_slot), int(features), _p(occ), _p(pos), _p(desc),
                                                       _p(val), pos.shape[0], _p(out)), "urf_search_by_projection_slot")
    else:
        f = np.ascontiguousarray(features, np.float64)
        check(_lib.lib().urf_search_by_projection(C.byref(c), _p(f), f.shape[0], _p(occ), _p(pos), _p(desc), _p(val),
                                                  pos.shape[0], _p(out)), "urf_search_by_projection")
    return out


def findFundamentalMat(points0, points1, thresh=3.0, confidence=0.99, device=0):
    """cv::findFundamentalMat(points0, points1, cv::FM_RANSAC, 3, 0.99, mask) (src/point_matching.cc:50) as the outlier stage 1
    runs it on the GPU (cvransac.hip; OpenCV 4.2's published algorithm restated, parity unpinned):
    -> (mask uint8[n], F float64[3, 3] or None when no model was found, hypothesis rounds run)"""
    p0 = np.ascontiguousarray(points0, np.float32).reshape(-1, 2)
    p1 = np.ascontiguousarray(points1, np.float32).reshape(-1, 2)
    assert p0.shape == p1.shape
    n = p0.shape[0]
    mask = np.zeros(max(n, 1), np.uint8)
    Fm = np.zeros(9, np.float64)
    it = C.c_int(-1)
    check(_lib.lib().urf_cv_find_fundamental(_p(p0), _p(p1), n, C.c_double(thresh), C.c_double(confidence), _p(mask), _p(Fm),
                                             C.byref(it), int(device)), "urf_cv_find_fundamental")
    return mask[:n].copy(), (Fm.reshape(3, 3) if np.any(Fm != 0.0) else None), it.value


def slot_to_host(d_slot_ptr):
    feat = np.zeros((CAP, 259), np.float64)
    K = C.c_int(0)
    check(_lib.lib().urf_slot_to_host(C.c_void_p(d_slot_ptr), _p(feat), CAP, C.byref(K)), "slot_to_host")
    return feat[:K.value].copy()


def probe_fma_gemm(A, B, bias=None, device=0):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    M, K = A.shape
    N = B.shape[1]
    out = np.zeros((M, N), np.float32)
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
    check(_lib.lib().urf_probe_fma_gemm(_p(A), _p(B), _p(bias), M, N, K, _p(out), device), "probe_fma_gemm")
    return out


def probe_h2gemm(X, W, bias=None, reps=10, device=0):
    X = np.ascontiguousarray(X, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    M, K = X.shape
    N = W.shape[1]
    Y = np.zeros((M, N), np.float32)
    ms = C.c_float(0)
    if bias is not None:
        bias = np.ascontiguousarray(bias, np.float32)
    check(_lib.lib().urf_probe_h2gemm(_p(X), _p(W), _p(bias), M, N, K, _p(Y), reps, C.byref(ms), device), "probe_h2gemm")
    return Y, float(ms.value)


def probe_math(x, device=0):
    x = np.ascontiguousarray(x, np.float32)
    e = np.zeros_like(x)
    l = np.zeros_like(x)
    check(_lib.lib().urf_probe_math(_p(x), x.size, _p(e), _p(l), device), "probe_math")
    return e, l


def probe_divsqrt(a, b, device=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    q = np.zeros_like(a)
    s = np.zeros_like(a)
    qd = np.zeros(a.size, np.float64)
    sd = np.zeros(a.size, np.float64)
    check(_lib.lib().urf_probe_divsqrt(_p(a), _p(b), a.size, _p(q), _p(s), _p(qd), _p(sd), device), "probe_divsqrt")
    return q, s, qd, sd


def set_profiling(on):
    _lib.lib().urf_set_profiling(int(bool(on)))


def minimal_sets(sampler, seed, n, iterations):
    """urf_minimal_sets: sampler 0 = counter hash, 1 = the reference's glibc rand() stream after srand(seed)"""
    sets = np.zeros((iterations, 8), np.int32)
    check(_lib.lib().urf_minimal_sets(int(sampler), C.c_uint32(seed), int(n), int(iterations), _p(sets)), "urf_minimal_sets")
    return sets


class PoseStage:
    """SolvePnPWithCV and FrameOptimization (src/g2o_optimization.cc:323-377, 179-321) batched over frames
    (urf_pose_*): lists of per-frame arrays in, lists out."""

    def __init__(self, camera_fxfycxcy, max_batch=8, capacity=1024, device=0):
        self.cam = [float(v) for v in camera_fxfycxcy]
        self.max_batch, self.cap = max_batch, capacity
        self._h = C.c_void_p()
        check(_lib.lib().urf_pose_create(device, max_batch, capacity, C.byref(self._h)), "urf_pose_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.lib().urf_pose_destroy(self._h)
            self._h = C.c_void_p()

    def _pack(self, arrays, width, dtype):
        B = len(arrays)
        n = np.array([len(a) for a in arrays], np.int32)
        cap = max(int(n.max()) if B else 1, 1)
        out = np.zeros((B, cap, width), dtype)
        for f, a in enumerate(arrays):
            out[f, :len(a)] = np.asarray(a, dtype).reshape(-1, width)
        return n, cap, out

    def SolvePnPWithCV(self, object_points, image_points, iterations=100, reprojection_error=20.0, confidence=0.99, seed=0):
        """per frame: (n_inliers, Twc[4,4], inlier flags[n])"""
        n, cap, obj = self._pack(object_points, 3, np.float32)
        _, _, img = self._pack(image_points, 2, np.float32)
        B = len(n)
        cfg = _lib.PnpConfig(*self.cam, iterations, reprojection_error, confidence, seed)
        pose = np.zeros((B, 4, 4), np.float64)
        inl = np.zeros((B, cap), np.uint8)
        k = np.zeros(B, np.int32)
        check(_lib.lib().urf_solve_pnp_ransac(self._h, C.byref(cfg), B, _p(n), _p(obj), _p(img), cap, _p(pose), _p(inl), _p(k)),
              "urf_solve_pnp_ransac")
        return [(int(k[f]), pose[f].copy(), inl[f, :n[f]].copy()) for f in range(B)]

    def FrameOptimization(self, map_points, keypoints, q_wc, p_wc, chi2_threshold=5.991):
        """per frame: (n - outliers, q_wc (w,x,y,z), p_wc, inlier flags[n])"""
        n, cap, X = self._pack(map_points, 3, np.float64)
        _, _, obs = self._pack(keypoints, 2, np.float64)
        B = len(n)
        q = np.ascontiguousarray(np.asarray(q_wc, np.float64).reshape(B, 4)).copy()
        p = np.ascontiguousarray(np.asarray(p_wc, np.float64).reshape(B, 3)).copy()
        cfg = _lib.PoseOptConfig(*self.cam, chi2_threshold)
        inl = np.zeros((B, cap), np.uint8)
        k = np.zeros(B, np.int32)
        check(_lib.lib().urf_frame_optimization(self._h, C.byref(cfg), B, _p(n), _p(X), _p(obs), cap, _p(q), _p(p), _p(inl),
                                                _p(k)), "urf_frame_optimization")
        return [(int(k[f]), q[f].copy(), p[f].copy(), inl[f, :n[f]].copy()) for f in range(B)]


    def FrameOptimizationStereo(self, bf, map_points, observations, n_mono, q_wc, p_wc, chi2_mono=5.991, chi2_stereo=7.815):
        """FrameOptimization with stereo edges: per frame observations [n, 3] = (u, v, u_right), the first n_mono[f] rows mono
        (u_right unused) -> per frame (n - outliers, q_wc, p_wc, inlier flags[n])"""
        n, cap, X = self._pack(map_points, 3, np.float64)
        _, _, obs = self._pack(observations, 3, np.float64)
        B = len(n)
        nm = np.ascontiguousarray(n_mono, np.int32)
        ns = (n - nm).astype(np.int32)
        q = np.ascontiguousarray(np.asarray(q_wc, np.float64).reshape(B, 4)).copy()
        p = np.ascontiguousarray(np.asarray(p_wc, np.float64).reshape(B, 3)).copy()
        cfg = _lib.PoseOptStereoConfig(*self.cam, float(bf), chi2_mono, chi2_stereo)
        inl = np.zeros((B, cap), np.uint8)
        k = np.zeros(B, np.int32)
        check(_lib.lib().urf_frame_optimization_stereo(self._h, C.byref(cfg), B, _p(nm), _p(ns), _p(X), _p(obs), cap, _p(q), _p(p),
                                                       _p(inl), _p(k)), "urf_frame_optimization_stereo")
        return [(int(k[f]), q[f].copy(), p[f].copy(), inl[f, :n[f]].copy()) for f in range(B)]
