"""ctypes loader of liburf_front.so (the HIP product library).

Fails loudly when the library is missing or does not export the C ABI declared
in include/urf.h: there is no CPU or eager fallback in the product path.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# URF_LIB: tools/ point this at the experiments build (liburf_front_exp.so, `make -C ur-mvo_amd/csrc experiments`) for A/B runs
SO_PATH = os.environ.get("URF_LIB") or os.path.join(_HERE, "liburf_front.so")

# every symbol include/urf.h declares
SYMBOLS = [
    "urf_last_error", "urf_build_info", "urf_device_count", "urf_sp_create", "urf_sp_build", "urf_sp_build_file", "urf_sp_build_config", "urf_pm_build_config", "urf_onnx_import",
    "urf_weights_save", "urf_sp_destroy", "urf_sp_infer", "urf_sp_infer_batch", "urf_slot_bytes",
    "urf_sp_infer_device", "urf_sp_sync", "urf_slot_to_host", "urf_sp_debug_tensor", "urf_pm_create",
    "urf_pm_build", "urf_pm_build_file", "urf_pm_destroy", "urf_normalize_keypoints", "urf_sg_infer",
    "urf_match", "urf_match_device", "urf_match_device_async", "urf_pm_fetch", "urf_pm_fetch_begin", "urf_pm_fetch_ready", "urf_pm_fetch_end", "urf_pm_sync",
    "urf_ransac_find_F", "urf_sp_stage_ms", "urf_pm_stage_ms", "urf_set_profiling", "urf_probe_fma_gemm",
    "urf_probe_math", "urf_probe_divsqrt", "urf_pm_share_stream", "urf_sp_stream", "urf_sp_result_stream", "urf_epipolar_reconstruct", "urf_probe_h2gemm", "urf_pm_wait_for_sp", "urf_pm_wait_event", "urf_sp_stage_ms_age", "urf_sp_wait_for_sinkhorn",
    "urf_cam_create", "urf_cam_create_from_maps", "urf_cam_destroy", "urf_cam_maps", "urf_cam_undistort",
    "urf_cam_undistort_device", "urf_cam_sync", "urf_cam_size",
    "urf_fe_create", "urf_fe_build", "urf_fe_build_files", "urf_fe_destroy", "urf_fe_set_camera", "urf_fe_submit",
    "urf_fe_collect", "urf_fe_in_flight", "urf_fe_max_in_flight", "urf_fe_ready", "urf_fe_frame_resident", "urf_fe_superpoint", "urf_fe_matcher", "urf_pm_stream",
    "urf_search_by_projection", "urf_search_by_projection_slot", "urf_probe_mfma_f16",
    "urf_ransac_find_F_sets", "urf_minimal_sets", "urf_epipolar_reconstruct_sets",
    "urf_comm_unique_id", "urf_comm_init", "urf_sp_near_tie_reruns", "urf_sp_calibrate_guard", "urf_sp_calibrate_guard_device", "urf_pm_near_tie_reruns", "urf_pm_calibrate_guard", "urf_pm_guard_state", "urf_pm_redo_engine_stats", "urf_pm_near_tie_flags", "urf_comm_init_all", "urf_comm_group_start", "urf_comm_group_end", "urf_comm_init_loopback", "urf_comm_destroy", "urf_comm_world", "urf_comm_rank",
    "urf_comm_allgather_slots", "urf_comm_gather", "urf_comm_plan_pairs", "urf_pm_device_results", "urf_pm_sinkhorn_fallbacks", "urf_pm_sinkhorn_integrity", "urf_pm_sinkhorn_residuals",
    "urf_sg_debug_couplings", "urf_cv_find_fundamental", "urf_pose_create", "urf_pose_destroy", "urf_solve_pnp_ransac", "urf_frame_optimization", "urf_frame_optimization_stereo",
]
# exported by the experiments build only (#ifdef URF_EXPERIMENTS in include/urf.h): fault injection, kernel A/B switches, diagnostics
EXPERIMENT_SYMBOLS = [
    "urf_probe_h2gemm_variant", "urf_probe_h2gemm_xflags", "urf_probe_h2gemm_deep", "urf_probe_redo_fault", "urf_probe_linear_dma", "urf_probe_attn_exact_nqt", "urf_probe_attn_variant", "urf_probe_sinkhorn_stamps", "urf_probe_sinkhorn_fault",
    "urf_probe_sinkhorn_backoff", "urf_probe_sinkhorn_corrupt", "urf_probe_mfma_roof",
]


class SPConfig(C.Structure):
    _fields_ = [("max_keypoints", C.c_int), ("keypoint_threshold", C.c_double), ("remove_borders", C.c_int),
                ("max_height", C.c_int), ("max_width", C.c_int), ("max_batch", C.c_int), ("device", C.c_int),
                ("precision", C.c_int), ("guard_delta", C.c_float), ("guard_ulps", C.c_float)]


class SGConfig(C.Structure):
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int), ("matching_threshold", C.c_double),
                ("sinkhorn_iterations", C.c_int), ("max_pairs", C.c_int), ("device", C.c_int),
                ("ransac_iterations", C.c_int), ("ransac_sigma", C.c_float), ("ransac_seed", C.c_uint32),
                ("precision", C.c_int), ("ransac_threshold_px", C.c_float), ("ransac_confidence", C.c_float),
                ("redo_flagged_pairs", C.c_int), ("guard_margin", C.c_float), ("outlier_stage", C.c_int),
                ("sinkhorn_residual_bound", C.c_float), ("calibrate_pairs", C.c_int), ("redo_merge", C.c_int),
                ("redo_shared_engine", C.c_int), ("audit_period", C.c_int)]


class EpiConfig(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("sigma", C.c_float), ("iterations", C.c_int), ("seed", C.c_uint32),
                ("sampler", C.c_int)]


class CamConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("distortion_type", C.c_int), ("K", C.c_double * 9),
                ("D", C.c_double * 14), ("n_dist", C.c_int), ("R", C.c_double * 9), ("P", C.c_double * 9),
                ("device", C.c_int)]


class FEConfig(C.Structure):
    _fields_ = [("sp", SPConfig), ("sg", SGConfig), ("batch", C.c_int), ("matchers", C.c_int),
                ("history_batches", C.c_int), ("outlier_rejection", C.c_int)]


class SbpConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("image_width", C.c_double), ("image_height", C.c_double), ("pose", C.c_double * 16), ("thr", C.c_int),
                ("device", C.c_int)]


class PnpConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("iterations", C.c_int),
                ("reprojection_error", C.c_double), ("confidence", C.c_double), ("seed", C.c_uint32)]


class PoseOptConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("chi2_threshold", C.c_double)]


class PoseOptStereoConfig(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("chi2_mono", C.c_double), ("chi2_stereo", C.c_double)]


class DMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int), ("trainIdx", C.c_int), ("distance", C.c_float)]


def build(force=False):
    """compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python __graft_entry__.py` or `make -C ur-mvo_amd/csrc`. "
                "The front-end has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        missing = [s for s in SYMBOLS if not hasattr(L, s)]
        if missing:
            raise RuntimeError(f"liburf_front.so lacks C-ABI symbols: {missing}")
        L.urf_build_info.restype = C.c_char_p
        if b"EXPERIMENTS" in L.urf_build_info():
            missing = [s for s in EXPERIMENT_SYMBOLS if not hasattr(L, s)]
            if missing:
                raise RuntimeError(f"the experiments build lacks its test hooks: {missing}")
        L.urf_last_error.restype = C.c_char_p
        L.urf_build_info.restype = C.c_char_p
        L.urf_slot_bytes.restype = C.c_size_t
        L.urf_normalize_keypoints.restype = None
        L.urf_sp_destroy.restype = None
        L.urf_pm_destroy.restype = None
        L.urf_cam_destroy.restype = None
        L.urf_fe_destroy.restype = None
        L.urf_comm_destroy.restype = None
        L.urf_pose_destroy.restype = None
        L.urf_fe_superpoint.restype = C.c_void_p
        L.urf_fe_matcher.restype = C.c_void_p
        L.urf_sp_stream.restype = C.c_void_p
        L.urf_sp_result_stream.restype = C.c_void_p
        L.urf_pm_stream.restype = C.c_void_p
        _lib = L
    return _lib


def check(rc, what=""):
    if rc < 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().urf_last_error().decode()}")
    return rc
