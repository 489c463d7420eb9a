"""CPU tests of the boundary: the C-ABI library loads, exports every symbol
include/urf.h declares, fails loudly without a GPU, and the host-side pieces
(keypoint normalisation, weight files, sharding logic) behave like the
reference.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared(hdr):
    d = set(re.findall(r"\b(urf_[a-z0-9_]+)\s*\(", hdr, flags=re.I))
    return {x for x in d if not x.startswith("urf_sp_config") and x != "urf_dmatch"}


def test_library_exports_every_declared_symbol(U):
    hdr = open(os.path.join(ROOT, "include", "urf.h")).read()
    # the block under #ifdef URF_EXPERIMENTS (test hooks, kernel A/B switches) belongs to the experiments build alone
    m = re.search(r"#ifdef URF_EXPERIMENTS\n(.*?)#endif\n", hdr, flags=re.S)
    assert m, "include/urf.h lost its experiments block"
    hooks = _declared(m.group(1))
    declared = _declared(hdr.replace(m.group(0), ""))
    assert len(declared) >= 30 and len(hooks) >= 5 and not (declared & hooks)
    L = U._lib.lib()
    missing = [d for d in sorted(declared) if not hasattr(L, d)]
    assert not missing, missing
    assert set(U._lib.SYMBOLS) == declared
    assert set(U._lib.EXPERIMENT_SYMBOLS) == hooks
    # the product library carries no fault injection and no kernel switches
    leaked = [d for d in sorted(hooks) if hasattr(L, d)]
    assert not leaked, leaked


def test_experiments_build_exports_the_test_hooks():
    from conftest import load_pkg_exp
    X = load_pkg_exp()
    if X is None:
        pytest.skip("liburf_front_exp.so is not built")
    L = X._lib.lib()
    assert b"EXPERIMENTS" in L.urf_build_info()
    assert all(hasattr(L, d) for d in X._lib.SYMBOLS + X._lib.EXPERIMENT_SYMBOLS)


def test_no_gpu_fails_loudly_not_silently(U):
    L = U._lib.lib()
    if L.urf_device_count() > 0:
        pytest.skip("GPU present")
    cfg = U._lib.SPConfig(1000, 0.0005, 4, 480, 640, 1, 0, 0)
    h = C.c_void_p()
    rc = L.urf_sp_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and not h.value
    assert b"HIP" in L.urf_last_error() or b"device" in L.urf_last_error()
    with pytest.raises(RuntimeError):
        U.frontend.SuperPoint(U.frontend.SuperPointConfig())
    with pytest.raises(RuntimeError):
        U.frontend.PointMatching(U.frontend.SuperGlueConfig())


def test_product_never_touches_the_oracle():
    """the product path must not import, link or call anything under oracle/"""
    pk = os.path.join(ROOT, "ur-mvo_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt.replace("tools/calibrate", "") or f == "synth.py", (dp, f)
                assert "import oracle" not in txt and "from oracle" not in txt, (dp, f)
                assert "liburf_oracle" not in txt, (dp, f)


def test_normalize_keypoints_host_matches_oracle(U, O):
    rng = np.random.default_rng(0)
    f = rng.random((37, 259))
    f[:, 1] = rng.integers(0, 640, 37)
    f[:, 2] = rng.integers(0, 480, 37)
    pm_out = np.zeros_like(f)
    U._lib.lib().urf_normalize_keypoints(f.ctypes.data_as(C.c_void_p), 37, 640, 512, pm_out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(pm_out, O.sg_normalize(f, 640, 512))
    assert pm_out[0, 1] == (f[0, 1] - 320) / (640 * 0.7)      # src/point_matching.cc:71-74


def test_weight_file_roundtrip_and_rejects_garbage(U, tmp_path):
    L = U._lib.lib()
    blob = np.arange(100, dtype=np.float32)
    p = str(tmp_path / "w.urfw").encode()
    assert L.urf_weights_save(p, 1, blob.ctypes.data_as(C.c_void_p), C.c_size_t(100)) == 0
    raw = open(p, "rb").read()
    assert raw[:4] == b"URFW" and len(raw) == 16 + 400
    assert np.array_equal(np.frombuffer(raw[16:], np.float32), blob)


def test_synthetic_weights_are_bit_reproducible(U):
    import hashlib
    spb = U.synth.pack_sp(U.synth.sp_weights(0))
    sgb = U.synth.pack_sg(U.synth.sg_weights(0))
    assert spb.size == 1300865 and sgb.size == 12003905
    # digests pinned when the golden fixtures were generated (tests/golden/make_golden.py)
    d = open(os.path.join(ROOT, "tests", "golden", "weights.sha256")).read().split()
    assert hashlib.sha256(spb.tobytes()).hexdigest() == d[0]
    assert hashlib.sha256(sgb.tobytes()).hexdigest() == d[1]


def test_shard_and_pair_assignment(U):
    D = U.dist
    assert [D.shard_range(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    pairs = sum((D.pairs_for_rank(16, r, 4) for r in range(4)), [])
    assert pairs == [(t - 1, t) for t in range(16)]       # every pair exactly once, in order
    with pytest.raises(AssertionError):
        D.shard_range(10, 0, 4)


def test_cpp_shim_headers_compile_and_link(tmp_path):
    """include/super_point.h, super_glue.h, point_matching.h (the kept C++ API)
    compile against the C ABI and link with liburf_front.so."""
    import subprocess
    exe = str(tmp_path / "test_shim")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_shim.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ur-mvo_amd"), "-lurf_front", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "ur-mvo_amd")])
    assert os.path.exists(exe)


# ------------------------------------------------------------------ weight import (the data format in front of build())
def test_superpoint_state_dict_and_onnx_import_reproduce_the_packed_blob(U, tmp_path):
    """names of superpoint/SP/model.py:38-53 -> the SP blob; the ONNX reader finds the same tensors by
    name (raw_data and float_data encodings) and, for an exporter that renamed them, by Conv order"""
    W = U.weights_io
    w = U.synth.sp_weights(0)
    blob = U.synth.pack_sp(w)
    sd = {}
    for name, (Wt, b) in w.items():
        sd[name + ".weight"], sd[name + ".bias"] = Wt, b
    assert np.array_equal(W.superpoint_from_state_dict(sd), blob)
    nodes, prev = [], "input"
    for i, (name, *_r) in enumerate(U.synth.SP_CONVS):
        nodes.append(("Conv", [prev, name + ".weight", name + ".bias"], [f"t{i}"]))
        nodes.append(("Relu", [f"t{i}"], [f"r{i}"]))
        prev = f"r{i}"
    for raw in (True, False):
        p = str(tmp_path / f"sp_{raw}.onnx")
        W.write_onnx(p, sd, nodes, raw=raw)
        assert np.array_equal(W.superpoint_from_onnx(p), blob)
    anon = {f"onnx::Conv_{100 + 2 * i}": sd[n + ".weight"] for i, (n, *_r) in enumerate(U.synth.SP_CONVS)}
    anon.update({f"onnx::Conv_{101 + 2 * i}": sd[n + ".bias"] for i, (n, *_r) in enumerate(U.synth.SP_CONVS)})
    nodes2 = [("Conv", [f"x{i}", f"onnx::Conv_{100 + 2 * i}", f"onnx::Conv_{101 + 2 * i}"], [f"x{i + 1}"])
              for i in range(len(U.synth.SP_CONVS))]
    p = str(tmp_path / "sp_anon.onnx")
    W.write_onnx(p, anon, nodes2)
    assert np.array_equal(W.superpoint_from_onnx(p), blob)
    # container round trip = what urf_sp_build_file reads
    c = str(tmp_path / "sp.urfw")
    W.save_container(c, W.KIND_SP, blob)
    kind, back = W.load_container(c)
    assert kind == 1 and np.array_equal(back, blob)
    with pytest.raises(ValueError):
        W.save_container(c, W.KIND_SG, blob)


def test_superglue_state_dict_import_folds_batchnorm_and_reorders_heads(U, tmp_path):
    W = U.weights_io
    w = U.synth.sg_weights(0)
    blob = U.synth.pack_sg(w)
    sd = W.superglue_to_state_dict(w)
    assert sd["gnn.layers.3.attn.proj.1.weight"].shape == (256, 256, 1) and "kenc.encoder.10.running_var" in sd
    assert np.array_equal(W.superglue_from_state_dict(sd), blob)
    # an exported graph with BatchNorm folded and anonymous initialisers: both images walk the same
    # layers (every weight is used twice), the importer keeps first uses in graph order
    folded, nodes, k = {}, [], [0]

    def conv(Wt, b, src):
        name = f"onnx::Conv_{k[0]}"
        if not any(v is Wt for v in folded.values()):
            folded[name + "w"], folded[name + "b"] = Wt[:, :, None], b
            k[0] += 1
        key = [n for n, v in folded.items() if v.base is Wt or v is Wt or (v.shape[:2] == Wt.shape and np.shares_memory(v, Wt))][0]
        nodes.append(("Conv", [src, key, key[:-1] + "b"], [src + "'"]))

    seq = []
    for (Wt, b, bnp) in w["kenc"]:
        seq.append(U.synth._fold_bn(Wt, b, bnp))
    for L in w["layers"]:
        seq += [L["q"], L["k"], L["v"], L["merge"], U.synth._fold_bn(*L["mlp0"]), L["mlp1"]]
    seq.append(w["final"])
    for img in ("a", "b"):                 # second image: the same initialisers again
        for (Wt, b) in seq:
            conv(Wt, b, img)
    folded["bin_score"] = np.array(w["bin_score"], np.float32)
    p = str(tmp_path / "sg.onnx")
    W.write_onnx(p, folded, nodes)
    got = W.superglue_from_onnx(p)
    assert got.shape == blob.shape and np.array_equal(got, blob)
    # the same graph with a BatchNormalization node left in (renamed initialisers): the positional path would
    # silently drop it -- refused
    p2 = str(tmp_path / "sg_bn.onnx")
    W.write_onnx(p2, folded, nodes[:3] + [("BatchNormalization", ["a'", "s", "b", "m", "v"], ["a''"])] + nodes[3:])
    with pytest.raises(ValueError, match="BatchNormalization"):
        W.superglue_from_onnx(p2)


def test_cpp_onnx_import_equals_the_python_packer(U, tmp_path):
    """build() from the reference's configuration alone (src/super_point.cpp:18-102, src/super_glue.cpp:21-147): when
    engine_file does not exist the library reads the initialisers of onnx_file itself (urf_onnx_import: a protobuf
    wire-format reader in C++, no ONNX runtime) and packs them.  The blob must be the Python packer's, float for float
    -- by parameter name (raw_data and float_data encodings, BatchNorm folded and heads re-ordered for SuperGlue) and by
    Conv order for an exporter that renamed the initialisers -- and the container urf_weights_save writes from it the
    same bytes as weights_io.save_container's."""
    import ctypes as C
    import hashlib
    W, L = U.weights_io, U._lib.lib()

    def cpp(path, kind, n):
        out = np.zeros(n, np.float32)
        rc = L.urf_onnx_import(path.encode(), kind, out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        assert rc == 0, L.urf_last_error()
        return out

    # SuperPoint
    w = U.synth.sp_weights(0)
    blob = U.synth.pack_sp(w)
    sd = {}
    for name, (Wt, b) in w.items():
        sd[name + ".weight"], sd[name + ".bias"] = Wt, b
    nodes = [("Conv", [f"x{i}", n + ".weight", n + ".bias"], [f"x{i + 1}"]) for i, (n, *_r) in enumerate(U.synth.SP_CONVS)]
    for raw in (True, False):
        p = str(tmp_path / f"sp_{raw}.onnx")
        W.write_onnx(p, sd, nodes, raw=raw)
        assert np.array_equal(cpp(p, 1, blob.size), blob)
    anon = {f"onnx::Conv_{100 + 2 * i}": sd[n + ".weight"] for i, (n, *_r) in enumerate(U.synth.SP_CONVS)}
    anon.update({f"onnx::Conv_{101 + 2 * i}": sd[n + ".bias"] for i, (n, *_r) in enumerate(U.synth.SP_CONVS)})
    nodes2 = [("Conv", [f"x{i}", f"onnx::Conv_{100 + 2 * i}", f"onnx::Conv_{101 + 2 * i}"], [f"x{i + 1}"]) for i in range(12)]
    p = str(tmp_path / "sp_anon.onnx")
    W.write_onnx(p, anon, nodes2)
    assert np.array_equal(cpp(p, 1, blob.size), blob)
    a, b = str(tmp_path / "a.urfw"), str(tmp_path / "b.urfw")
    got = cpp(p, 1, blob.size)
    assert L.urf_weights_save(a.encode(), 1, got.ctypes.data_as(C.c_void_p), C.c_size_t(got.size)) == 0
    W.save_container(b, W.KIND_SP, blob)
    assert hashlib.sha256(open(a, "rb").read()).hexdigest() == hashlib.sha256(open(b, "rb").read()).hexdigest()
    # SuperGlue, parameter names kept (BatchNorm present as initialisers: folded by the importer)
    wg = U.synth.sg_weights(0)
    gblob = U.synth.pack_sg(wg)
    sdg = W.superglue_to_state_dict(wg)
    p = str(tmp_path / "sg_named.onnx")
    W.write_onnx(p, {k: v for k, v in sdg.items()}, [("Conv", ["x", "final_proj.weight", "final_proj.bias"], ["y"])])
    assert np.array_equal(W.superglue_from_onnx(p), gblob)
    assert np.array_equal(cpp(p, 2, gblob.size), gblob)
    # ... and folded + renamed: the distinct Conv weights in graph order, every weight used by both images
    seq = [U.synth._fold_bn(Wt, b_, bnp) for (Wt, b_, bnp) in wg["kenc"]]
    for Lr in wg["layers"]:
        seq += [Lr["q"], Lr["k"], Lr["v"], Lr["merge"], U.synth._fold_bn(*Lr["mlp0"]), Lr["mlp1"]]
    seq.append(wg["final"])
    folded, nodes3 = {}, []
    for i, (Wt, b_) in enumerate(seq):
        folded[f"onnx::Conv_{2 * i}"], folded[f"onnx::Conv_{2 * i + 1}"] = Wt[:, :, None], b_
    for img in ("a", "b"):
        nodes3 += [("Conv", [img, f"onnx::Conv_{2 * i}", f"onnx::Conv_{2 * i + 1}"], [img + "'"]) for i in range(len(seq))]
    folded["bin_score"] = np.array(wg["bin_score"], np.float32)
    p = str(tmp_path / "sg_folded.onnx")
    W.write_onnx(p, folded, nodes3)
    assert np.array_equal(cpp(p, 2, gblob.size), gblob)
    # errors: a BatchNormalization node with renamed initialisers, a wrong kind, a missing file
    p2 = str(tmp_path / "sg_bn.onnx")
    W.write_onnx(p2, folded, nodes3[:3] + [("BatchNormalization", ["a'", "s", "b", "m", "v"], ["a''"])] + nodes3[3:])
    out = np.zeros(gblob.size, np.float32)
    assert L.urf_onnx_import(p2.encode(), 2, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)) < 0 and b"BatchNormalization" in L.urf_last_error()
    assert L.urf_onnx_import(p.encode(), 1, out.ctypes.data_as(C.c_void_p), C.c_size_t(blob.size)) < 0
    assert L.urf_onnx_import(str(tmp_path / "none.onnx").encode(), 2, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)) < 0


def test_minimal_sets_glibc_stream_equals_the_c_library(U, O):
    """urf_minimal_sets(URF_SAMPLER_GLIBC): the product restates glibc's rand() (TYPE_3 additive feedback generator)
    to draw the reference's minimal sets (src/epipolar_geometry.cc:56-71,100-117); the oracle calls the C
    library's own srand()/rand().  Also the counter-hash sampler against the oracle's."""
    import ctypes as C
    F = U.frontend
    for (seed, n, its) in [(0, 8, 5), (0, 431, 200), (7, 1000, 200), (123456, 50, 37), (1, 9, 3)]:
        a, b = F.minimal_sets(1, seed, n, its), O.minimal_sets(1, seed, n, its)
        assert np.array_equal(a, b), (seed, n, its)
        assert a.min() >= 0 and a.max() < n and all(len(set(r)) == 8 for r in a)      # without replacement
        assert np.array_equal(F.minimal_sets(0, seed, n, its), O.minimal_sets(0, seed, n, its))
    # srand(0) is srand(1) in glibc, and the first draw follows Random::RandomInt
    assert np.array_equal(F.minimal_sets(1, 0, 100, 4), F.minimal_sets(1, 1, 100, 4))
    libc = C.CDLL("libc.so.6")
    libc.srand(0)
    r = libc.rand()
    assert F.minimal_sets(1, 0, 100, 1)[0, 0] == int((r / 2147483648.0) * 100)
    with pytest.raises(RuntimeError):
        F.minimal_sets(1, 0, 7, 1)              # fewer than 8 matches cannot seed a hypothesis


def test_register_and_lds_budgets_keep_the_pipeline_coresident(tmp_path):
    """in the three-stream pipeline one h2conv workgroup (one wave per SIMD) runs beside one attn_h2 workgroup
    (two waves per SIMD) on the same CU, which needs 2 x VGPR(attention) + VGPR(conv) <= 512 and LDS(attention) + LDS(conv)
    <= 160 KB.  A kernel edit that crosses the budget costs ~5 % of the step without showing in any serialised timing."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    usage = {}
    for name in ("attn_h2", "h2conv"):
        out = tmp_path / f"{name}.s"
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "ur-mvo_amd", "csrc", f"{name}.hip"), "-o", str(out)], check=True, capture_output=True)
        text = out.read_text()
        for m in re.finditer(r"^(_ZN3urf\w+):.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; LDSByteSize: (\d+)", text, re.S | re.M):
            usage[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    attn = [v for k, v in usage.items() if "attn_h2_kernelILi2ELi8E" in k]
    convs = [v for k, v in usage.items() if "h2conv_kernel" in k]
    assert len(attn) == 1 and len(convs) == 4, usage
    up8 = lambda n: (n + 7) // 8 * 8                   # VGPR allocation granule
    for vg, scratch, _ in convs:
        assert 2 * up8(attn[0][0]) + up8(vg) <= 512, (attn, convs)
        assert scratch <= 128, convs                    # a handful of spilled registers around the input staging at most (none in the tap loop)
    assert attn[0][1] == 0
    conv_lds = 2 * (2 * 2 * 64 * 64 + 2 * 10 * 18 * 64) + 12 * 20 * 4   # launch_h2conv's dynamic LDS, fused variant
    assert attn[0][2] + conv_lds <= 160 * 1024
    # the resident Sinkhorn of the product (sinkhorn_wide_kernel: 512 threads, 64 rows per workgroup): two waves per SIMD with
    # (next to) no scratch, a few KB of LDS (round 4's 110 KB of padding that kept the CU to itself is gone with the fault it was
    # there to avoid: DESIGN.md section 12), and no shared form in the product build
    out = tmp_path / "sinkhorn_resident.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "ur-mvo_amd", "csrc", "sinkhorn_resident.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    for m in re.finditer(r"^(_ZN3urf\w+):.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; LDSByteSize: (\d+)", text, re.S | re.M):
        usage[m.group(1)] = tuple(int(m.group(i)) for i in (2, 3, 4))
    wide = [v for k, v in usage.items() if "sinkhorn_wide_kernel" in k]
    assert len(wide) == 1, usage
    assert wide[0][0] <= 256 and wide[0][1] <= 512 and wide[0][2] <= 16 * 1024, wide
    assert not any("sinkhorn_regs_kernel" in k or "sinkhorn_resident_kernel" in k for k in usage), "the forms that share their CUs are experiments-build only"


def test_integration_patches_apply_to_the_reference(tmp_path):
    """integration/*.patch (the batched caller in Tracking and the pybind engine) apply cleanly to the
    reference tree they were diffed against; and every urf_* call they add is a symbol of the C ABI"""
    import shutil
    import subprocess
    ref = "/root/reference"
    patches = [os.path.join(ROOT, "integration", n) for n in ("tracking.patch", "main_py.patch")]
    text = "".join(open(p).read() for p in patches)
    added = set(re.findall(r"^\+.*?\b(urf_[a-z_]+)\(", text, re.M))
    assert {"urf_fe_create", "urf_fe_build_files", "urf_fe_submit", "urf_fe_collect", "urf_fe_destroy"} <= added
    from conftest import load_pkg
    assert added <= set(load_pkg()._lib.SYMBOLS)
    assert "urf_fe_frame_resident(" in text and "getenv" not in text
    assert "usleep(30000)" in text and "-        usleep(30000);" in text
    if not os.path.isdir(ref) or not shutil.which("patch"):
        pytest.skip("reference tree (or patch) not available here")
    for rel in ("src/tracking.cc", "include/tracking.h", "main_py.cpp"):
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        shutil.copy(os.path.join(ref, rel), tmp_path / rel)
    for p in patches:
        subprocess.run(["patch", "-p1", "--dry-run", "-i", p], cwd=tmp_path, check=True, capture_output=True)
        subprocess.run(["patch", "-p1", "-i", p], cwd=tmp_path, check=True, capture_output=True)
    out = open(tmp_path / "src" / "tracking.cc").read()
    assert "ExtractFeatureBatch" in out and "std::thread point_ectraction_thread" not in out
    assert "usleep(30000);" not in open(tmp_path / "main_py.cpp").read()


def test_shim_headers_opencv_branch_is_well_formed():
    """the shim headers choose their OpenCV / Eigen branch with __has_include; this image has neither library, so that branch
    (SuperPoint::visualization's cv::cvtColor / cv::circle / cv::imwrite, include/super_point.h) is otherwise never seen by a
    compiler.  Syntax check only (-fsyntax-only) against declaration stubs kept under tests/cpp/cv_stubs."""
    import subprocess
    src = "#include <super_point.h>\n#include <super_glue.h>\n#include <point_matching.h>\n#include <epipolar_geometry.h>\n" \
          "#ifndef URF_HAVE_CV\n#error the OpenCV branch was not selected\n#endif\n" \
          "void f(SuperPoint &sp, const cv::Mat &im) { sp.visualization(\"x\", im); }\n"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "cpp", "cv_stubs"),
                        "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_no_barrier_is_reached_with_an_lds_write_in_flight():
    """hipcc of ROCm 7.2 leaves out the `s_waitcnt lgkmcnt(0)` of __syncthreads() at a barrier that heads a loop when the
    pending LDS write sits on the back edge; the other waves then read the previous iteration's value whenever that write is
    still queued behind other workgroups' LDS traffic -- the run-to-run differences of the register-resident Sinkhorn that
    round 4 could only avoid.  The audit (tools/isa_barrier_audit.py) walks the basic-block graph of
    every kernel of the product build: no s_barrier may be reachable with one of the wave's own LDS writes un-waited.  It
    must also still see the fault where it is known to be: the shared forms of the experiments build without their fix."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_barrier_audit", os.path.join(ROOT, "tools", "isa_barrier_audit.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    # the checker on hand-written streams: a write on the back edge of a loop whose header is the barrier (the fault), the same
    # with the wait written out, a counted wait that does / does not cover the write, a scalar load in between (out of order)
    def fn(body):
        return "kern:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n"
    loop = "\tds_read_b32 v1, v0\n.LBB0_1:\n\ts_barrier\n\tds_read_b32 v2, v0\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 v0, v2\n%s\ts_cbranch_scc1 .LBB0_1"
    assert [ln for _, ln in A.audit_text(fn(loop % ""))] == [4]
    assert A.audit_text(fn(loop % "\ts_waitcnt lgkmcnt(0)\n")) == []
    assert A.audit_text(fn("\tds_write_b32 v0, v1\n\tds_read_b32 v2, v0\n\ts_waitcnt lgkmcnt(1)\n\ts_barrier")) == []
    assert len(A.audit_text(fn("\tds_write_b32 v0, v1\n\tds_read_b32 v2, v0\n\ts_waitcnt lgkmcnt(2)\n\ts_barrier"))) == 1
    assert len(A.audit_text(fn("\tds_write_b32 v0, v1\n\ts_load_dword s0, s[2:3], 0x0\n\ts_waitcnt lgkmcnt(1)\n\ts_barrier"))) == 1
    assert A.audit_text(fn("\tds_write_b32 v0, v1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier")) == []
    csrc = os.path.join(ROOT, "ur-mvo_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read() if f.endswith(".hip") else ""
        if "__syncthreads" in src or "s_barrier" in src:
            assert A.audit_text(A.compile_to_asm(os.path.join(csrc, f))) == [], f
    hits = A.audit_text(A.compile_to_asm(os.path.join(csrc, "sinkhorn_resident.hip"), ["-DURF_EXPERIMENTS"]))
    assert any("sinkhorn_regs_kernel" in name for name, _ in hits), "the audit no longer sees the known fault of the shared forms"
    fixed = A.audit_text(A.compile_to_asm(os.path.join(csrc, "sinkhorn_resident.hip"), ["-DURF_EXPERIMENTS", "-DURF_RS_LGKM_BARRIER"]))
    assert not any("sinkhorn_regs_kernel" in name for name, _ in fixed)
