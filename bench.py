"""bench.py -- VO front-end throughput on MI355X.

One "step" = one batch of 8 new 640x480 frames per GPU through the whole hot
path: SuperPoint on the 8 frames, SuperGlue + 8-point RANSAC on the 8 pairs
(frame t-1, t), inputs already resident in HBM, match lists delivered to the
host.  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), frames
block-sharded, one all-gather of feature slots per step (weak scaling: 8 frames
per GPU per step).

    python bench.py --gpus 1 --steps 20 --warmup 3

The K-step timed region (barrier + synchronize on both sides, max over ranks) is run --repeats times back to
back; `value` / `ms_per_step` are the MEDIAN region, `repeats` holds every region (boxes and clocks vary by
several per cent from run to run).  Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events
on the library's own stream; `cpu_baseline` times the CPU oracle on a bounded sample.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

H, W = 480, 640
BATCH = 8
# diagnostic only (what-if runs); any value but 100 is not the reference configuration and is flagged in the output
SINK_ITERS = int(os.environ.get("URF_BENCH_SINKHORN_ITERS", "100"))
MAX_KP = 1000
# algorithmic work, SURVEY.md 8(d): GFLOP per 640x480 frame / per pair at n=1000
GF_CONV1 = 22.649 + 0.354          # conv1b + fused conv1a
FP32_MFMA_PEAK_TF = 157.3          # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
F16_MFMA_PEAK_TF = 2500.0          # dense f16/bf16 MFMA peak (v_mfma_f32_16x16x32_f16)
HBM_PEAK_GBS = 8000.0              # HBM3E peak (MI355X_MICROARCH.md)


def sg_linear_gflop(n0, n1):
    per_pt = 2 * (3 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256 * 256)          # kenc
    per_pt += 18 * 2 * (3 * 256 * 256 + 256 * 256 + 512 * 512 + 512 * 256)      # qkv, merge, mlp
    per_pt += 2 * 256 * 256                                                      # final proj
    return per_pt * (n0 + n1) / 1e9


def sg_linear_gbytes(n0, n1, fast):
    """algorithmic HBM-side bytes of the linear layers of one pair: every activation tensor read or
    written once at 4 B per element (fp32, or two f16 planes in the fast mode), weights once per step
    (counted by the caller).  Per GNN layer and keypoint: qkv 256 in + 768 out, merge 256 + 256 (exact
    mode only: the fast mode folds the merge weights into mlp0 at build time), mlp0 512 + 512,
    mlp1 512 in + 256 residual + 256 out (the fast mode keeps the residual stream in its two f16 planes only)."""
    per_pt = 18 * (1024 + (0 if fast else 512) + 1024 + 1024)
    per_pt += (4 + 32) + (32 + 64) + (64 + 128) + (128 + 256) + (256 + 256 + 256) + (256 + 256)   # kenc, final proj
    return 4.0 * per_pt * (n0 + n1) / 1e9


def sg_attn_gbytes(n0, n1):
    """q, k, v in and the message out, once each, 256 channels x 4 B, 18 layers"""
    return 18 * 4 * 256 * 4.0 * (n0 + n1) / 1e9


def sg_attn_gflop(n0, n1):
    # per layer, per image: QK^T + PV = 2 * 2*nq*ns*256
    self_l = 4 * 256 * (n0 * n0 + n1 * n1)
    cross_l = 4 * 256 * (2 * n0 * n1)
    return 9 * (self_l + cross_l) / 1e9



def kernel_source_sha():
    """sha256 over the HIP sources of the library: a PMC summary is only quoted for the kernels it was taken on"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ur-mvo_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ur-mvo_amd", "csrc", "*.h"))):
        if os.path.basename(f) != "probes.hip":          # diagnostics only: no kernel of the path lives there
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_label, resolution, prec):
    """HBM-side bytes per step of a kernel family, from the committed summary of the two rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE; tools/pmc_summary.py) of this same command.  bench.py cannot collect counters about
    itself: the summary is the newest profiles/r*_pmc_hbm*.json for this resolution whose `source_sha` equals the
    sha of the kernel sources in this tree -- a summary taken on other kernels is refused (traffic = null)."""
    tag = ("" if resolution == "640x480" else "_" + resolution) + {0: "_exact", 1: "_fast", 2: "_guarded", 3: ""}[prec]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm{tag}.json")))
    if not files:
        return None, "no committed PMC summary for this resolution"
    d = json.load(open(files[-1]))
    if d.get("source_sha") != kernel_source_sha():
        return None, (f"{os.path.basename(files[-1])} was taken on other kernel sources (sha {d.get('source_sha')} != "
                      f"{kernel_source_sha()}): refused")
    fam = ("conv_mfma_kernel<9,pool,fuse1a>" if "conv_mfma_kernel<9" in kernel_label else "linear_exact" if "gemm128" in kernel_label
           else "attn_kernel" if "(attn_kernel)" in kernel_label
           else "h2gemm" if "h2gemm" in kernel_label else "attn_h2_kernel" if "attn_h2" in kernel_label
           else "sinkhorn_wide_kernel" if "sinkhorn_wide_kernel" in kernel_label
           else "sinkhorn_regs_kernel" if "sinkhorn_regs_kernel" in kernel_label
           else "sinkhorn_resident_kernel" if "sinkhorn_resident_kernel" in kernel_label
           else "sinkhorn_half_kernel" if "inkhorn" in kernel_label else "h2conv_kernel<pool,fuse1a>")
    k = d["kernels"].get(fam)
    if not k:
        return None, f"{os.path.basename(files[-1])} has no {fam}"
    calls = d["superpoint_calls"] if (fam.startswith("h2conv") or fam.startswith("conv_mfma")) else d["matcher_calls"]
    return int(k["bytes_total"] / calls), (f"bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the "
                                           f"{k['launches'] // calls} {fam} launches of a step; L2-miss traffic incl. "
                                           f"Infinity-Cache hits; {os.path.basename(files[-1])}")


def pmc_mfma_busy(kernel_label, resolution, prec):
    """the matrix pipe's busy fraction of a kernel family while it runs ALONE (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
    GRBM_GUI_ACTIVE serialises the dispatches), from the committed summary (tools/pmc_mfma_summary.py) taken on THESE kernel
    sources; None otherwise.  Counter evidence beside the computed `mfma_issue_frac`, not a replacement for it."""
    tag = ("" if resolution == "640x480" else "_" + resolution) + {0: "_exact", 1: "_fast", 2: "_guarded", 3: ""}[prec]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_mfma{tag}.json")))
    if not files:
        return None, "no committed matrix-pipe PMC summary for this resolution and mode"
    d = json.load(open(files[-1]))
    if d.get("source_sha") != kernel_source_sha():
        return None, f"{os.path.basename(files[-1])} was taken on other kernel sources: refused"
    fam = ("conv_mfma_kernel<9,pool,fuse1a>" if "conv_mfma_kernel<9" in kernel_label else "linear_exact" if "gemm128" in kernel_label
           else "attn_kernel" if "(attn_kernel)" in kernel_label
           else "h2gemm" if "h2gemm" in kernel_label else "attn_h2_kernel" if "attn_h2" in kernel_label
           else "h2conv_kernel<pool,fuse1a>" if "h2conv" in kernel_label else None)
    k = d["kernels"].get(fam) if fam else None
    if not k:
        return None, f"{os.path.basename(files[-1])} has no matrix-core kernel for {kernel_label}"
    return k["mfma_busy_frac"], (f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs) over the {k['launches']} {fam} launches of "
                                 f"{os.path.basename(files[-1])} (each kernel alone on the chip)")


def stream_run(U, spb, sgb, dev, device_index, prec, Hh, Ww, batch, steps, repeats, warmup=5, keep=False):
    """One more configuration through the SAME loop as the headline (ur-mvo_amd/pipeline.py, two matcher handles, SuperPoint two
    batches ahead): `repeats` timed regions of `steps` steps, median reported.  keep=True also returns every fetched list."""
    import torch
    F, synth, P = U.frontend, U.synth, U.pipeline
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=MAX_KP), max_height=Hh, max_width=Ww, max_batch=batch, device=device_index,
                      precision=prec)
    assert sp.build(spb), U._lib.lib().urf_last_error()
    pms = []
    for _ in range(2):
        m = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=batch, device=device_index, precision=prec)
        assert m.build(sgb), U._lib.lib().urf_last_error()
        pms.append(m)
    NB_ = 5
    fr = synth.shift_stream(100, NB_ * batch, Hh, Ww)
    d_fr = torch.from_numpy(np.stack(fr)).to(dev)
    pipe = P.SlotRingPipeline(sp, pms, d_fr, batch, Hh, Ww, device=dev)
    kept = {}
    rec = (lambda b, mt, res: kept.__setitem__(b, res)) if keep else None
    pipe.prologue()
    pipe.run(0, warmup)
    pipe.drain()
    sp.sync()
    nb, regions = warmup, []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(nb, steps, rec)
        pipe.drain(rec)
        sp.sync()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
        nb += steps
    dt = float(np.median(regions))
    g = [m.near_tie_reruns() for m in pms]
    out = {"frames_per_s": round(steps * batch / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "batch": batch, "steps": steps,
           "regions_frames_per_s": [round(steps * batch / r, 2) for r in regions],
           "pairs": sum(x["pairs"] for x in g), "pairs_flagged": sum(x["flagged"] for x in g), "pairs_redone_exact": sum(x["redone"] for x in g),
           "sinkhorn_fallbacks": sum(m.sinkhorn_fallbacks() for m in pms)}
    if prec == 3:
        gs = [m.guard_state() for m in pms]
        out["guard"] = {"margin_in_use": max(g_["margin"] for g_ in gs), "calibrated_difference": max(g_["measured"] for g_ in gs),
                        "online_largest_difference_on_a_redone_pair": max(g_["online_worst"] for g_ in gs),
                        "online_pairs_sampled": sum(g_["online_pairs"] for g_ in gs), "margin_raises": sum(g_["margin_raises"] for g_ in gs),
                        "online_violations": sum(g_["online_violations"] for g_ in gs), "audits": sum(g_["audits"] for g_ in gs),
                        "audit_mismatches": sum(g_["audit_mismatches"] for g_ in gs), "every_pair_redone": any(g_["redo_all"] for g_ in gs),
                        "batches_run_in_the_exact_mode_instead": sum(g_["exact_batches"] for g_ in gs)}
    del pipe, pms, sp, d_fr
    return (out, kept) if keep else out


def native_frame_stream_run(U, spb, sgb, device_index, prec, Hh, Ww, batch, steps, repeats, warmup=5, kept_ref=None):
    """The C++ caller a maintainer binds (urf_fe_submit / urf_fe_collect, integration/tracking.patch; src/tracking.cc:338-377):
    HOST u8 frames in (the copy to pinned memory and the PCIe transfer are inside the timed region), match lists out, the
    pipelined loop of ur-mvo_amd/csrc/fe_api.hip -- no Python in the data path but the two ctypes calls per batch.
    kept_ref: {batch: lists} of the headline run over the same stream: every list this run collects must equal it."""
    F, synth = U.frontend, U.synth
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=MAX_KP), F.SuperGlueConfig(image_width=640, image_height=512), batch=batch,
                       max_height=Hh, max_width=Ww, device=device_index, precision=prec, matchers=2)
    assert fs.build(spb, sgb), U._lib.lib().urf_last_error()
    NB_ = 5
    fr = np.stack(synth.shift_stream(100, NB_ * batch, Hh, Ww))
    depth = fs.max_in_flight() - 1       # (matchers + 4 = 6) batches stay in flight behind a submit (include/urf.h)
    got = {}
    nsub = [0]

    def pump(n, keep):
        for _ in range(n):
            k = nsub[0] % NB_
            fs.submit(fr[k * batch:(k + 1) * batch])
            nsub[0] += 1
            while fs.in_flight() > depth or (fs.in_flight() and fs.ready()):
                K_, m_ = fs.collect()
                if keep is not None:
                    keep.append(m_)

    def drain(keep):
        while fs.in_flight():
            K_, m_ = fs.collect()
            if keep is not None:
                keep.append(m_)

    pump(warmup, None)
    drain(None)
    regions, lists = [], []
    for _ in range(repeats):
        t0 = time.perf_counter()
        pump(steps, lists)
        drain(lists)
        regions.append(time.perf_counter() - t0)
    dt = float(np.median(regions))
    out = {"frames_per_s": round(steps * batch / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "batch": batch, "steps": steps,
           "regions_frames_per_s": [round(steps * batch / r, 2) for r in regions],
           "what": "urf_fe_submit (host u8 frames: pinned staging + PCIe inside the timed region) / urf_fe_collect (host DMatch lists), "
                   "two matcher handles, matchers + 4 batches in flight, batches handed out as soon as they are final"}
    if kept_ref is not None:
        # batch index of lists[i] in the stream: warm-up batches came first; the stream is periodic in NB_ batches (the first
        # frame of a batch is matched against the last frame of the previous one, so from batch 1 on the lists repeat)
        same = tot = 0
        for i, m_ in enumerate(lists):
            b = warmup + i
            refs = [kept_ref[r] for r in kept_ref if r % NB_ == b % NB_ and r >= 1]
            if not refs or b < 1:
                continue
            for a_, x_ in zip(m_, refs[0]):
                tot += 1
                same += int(len(a_) == len(x_) and np.array_equal(a_["queryIdx"], x_["queryIdx"]) and np.array_equal(a_["trainIdx"], x_["trainIdx"]))
        out["pairs_with_the_headline_runs_index_list"] = f"{same}/{tot}"
    del fs
    return out


def latency_runs(U, spb, sgb, device_index, prec, Hh, Ww, repeats=3):
    """BASELINE configs[1] (SuperPoint only, batch 1, <= 1024 keypoints) and the per-call path of the UNPATCHED reference caller
    (Tracking::ExtractFeatureAndMatch, src/tracking.cc:338-377: SuperPoint::infer on one frame, then MatchingPoints on host
    features, one pair) through the very entry points the C++ shims call (urf_sp_infer, urf_match): host buffers in and out."""
    F, synth = U.frontend, U.synth
    fr = synth.shift_stream(100, 12, Hh, Ww)
    sp1 = F.SuperPoint(F.SuperPointConfig(max_keypoints=MAX_KP), max_height=Hh, max_width=Ww, max_batch=1, device=device_index, precision=prec)
    pm1 = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=1, device=device_index, precision=prec)
    assert sp1.build(spb) and pm1.build(sgb)
    feats = [sp1.infer(f) for f in fr]                      # warm-up; the features of the pairs below
    pm1.MatchingPoints(feats[0], feats[1], True)
    t_sp, t_pm, t_call = [], [], []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for f in fr:
            sp1.infer(f)
        t_sp.append((time.perf_counter() - t0) / len(fr) * 1e3)
        t0 = time.perf_counter()
        for j in range(len(fr) - 1):
            t1 = time.perf_counter()
            pm1.MatchingPoints(feats[j], feats[j + 1], True)
            t_call.append((time.perf_counter() - t1) * 1e3)
        t_pm.append((time.perf_counter() - t0) / (len(fr) - 1) * 1e3)
    g = pm1.near_tie_reruns()
    return {"superpoint_infer_ms_per_frame": round(float(np.median(t_sp)), 3), "matching_points_ms_per_pair": round(float(np.median(t_pm)), 3),
            # (the mean above includes the calls whose pair the guard flagged: those run the exact redo before they return)
            "matching_points_ms_median_call": round(float(np.median(t_call)), 3), "matching_points_ms_slowest_call": round(float(np.max(t_call)), 3),
            "frames_per_s_one_frame_at_a_time": round(1e3 / (float(np.median(t_sp)) + float(np.median(t_pm))), 1),
            "keypoints": int(feats[0].shape[0]), "pairs_redone_exact": g["redone"], "pairs": g["pairs"],
            "what": "urf_sp_infer (u8 frame on the host -> 259 x K f64 on the host) and urf_match (two host feature matrices -> DMatch list)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 0.5 s per timed region: the drain of the last step is 1 % of it, not 5 %
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--batch-per-gpu", type=int, default=8,
                    help="frames (and pairs) per GPU per step: 8 = BASELINE configs[2]; 4 with --gpus 8 and --resolution "
                         "1241x376 = configs[3] (batch 32 sharded over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", type=int, default=3, choices=[0, 1, 2, 3],
                    help="3 = strict parity (default; the only mode besides 0 whose keypoints AND match index lists are the "
                         "oracle's): SuperPoint in exact fp32, the matcher on the f16 matrix core (split operands), every pair with "
                         "a decisive entry within the fast matcher's error of its alternative redone by the exact matcher inside the "
                         "library, inside the timed region; 2 = guarded fast: fast SuperPoint with the top-k cut resolved in exact "
                         "arithmetic (keypoint SET exact, order not), near-tied pairs flagged and NOT redone; 1 = fast without any "
                         "guard; 0 = exact fp32 everywhere (every tensor bit-identical to the oracle)")
    ap.add_argument("--no-exact-check", action="store_true", help="skip the exact-mode reference pass (N=1 only)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other single-GPU configurations measured after the headline (guarded fast mode, SuperPoint only at "
                         "batch 1, the per-call path, the 1241x376 stream at batch 8 and 4)")
    ap.add_argument("--no-guard-calibration", action="store_true",
                    help="skip the check of the guard's error model before the timed region (counter-collection runs: its "
                         "exact-mode launches would be tallied with the step's)")
    ap.add_argument("--matcher-gain", type=float, default=0.5,
                    help="residual gain of the seeded SuperGlue weights (synth.sg_weights(0, gnn_gain)): 0.5 = the headline's; larger gains "
                         "make the fast matcher's error, the strict handle's margin and the share of flagged pairs grow (what-if runs)")
    ap.add_argument("--resolution", default="640x480", choices=["640x480", "1241x376"],
                    help="frame size WxH: 640x480 (headline, BASELINE configs[2]) or the KITTI-size stream of configs[3]")
    args = ap.parse_args()
    global H, W, GF_CONV1, BATCH
    BATCH = args.batch_per_gpu
    assert 1 <= BATCH <= 64
    if args.resolution == "1241x376":
        H, W = 376, 1241
        GF_CONV1 = (22.649 + 0.354) * (376 * 1241) / (480 * 640)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the front-end has no CPU fallback")
    # URF_BENCH_SHARED_GPU=1 is a test rig: all ranks on cuda:0 with gloo, to
    # exercise the N>1 control flow on a 1-GPU box (numbers are meaningless).
    shared_gpu = os.environ.get("URF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # URF_BENCH_FORCE_DIST=1: run the N>1 exchange path (RCCL all-gather on its own stream, ordered
    # by events) in a process group of ONE rank -- the only way to execute that code on a 1-GPU box
    force_dist = world == 1 and os.environ.get("URF_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    exchange = world > 1 or force_dist          # feature slots go through the all-gather
    async_exchange = exchange and not shared_gpu

    U = load_pkg()
    F, synth, D, P = U.frontend, U.synth, U.dist, U.pipeline
    spb = synth.pack_sp(synth.sp_weights(0))
    sgb = synth.pack_sg(synth.sg_weights(0, gnn_gain=args.matcher_gain))
    PREC = args.precision
    FAST = PREC >= 1                  # the matcher runs on the f16 matrix core
    SP_FAST = PREC in (1, 2)          # ... and so does SuperPoint (strict parity keeps it in exact fp32)
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=MAX_KP), max_height=H, max_width=W, max_batch=BATCH,
                      device=local_rank, precision=PREC)
    assert sp.build(spb), U._lib.lib().urf_last_error()
    pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=BATCH, device=local_rank,
                         precision=PREC, sinkhorn_iterations=SINK_ITERS)
    assert pm.build(sgb), U._lib.lib().urf_last_error()

    # ONE synthetic stream, resident in HBM before the timed region.  Global batch k
    # = frames [k*8*world, (k+1)*8*world); rank g owns the block [g*8, g*8+8) of
    # it.  NB global batches are cycled.
    OVERLAP = int(os.environ.get("URF_BENCH_OVERLAP", "2"))
    MATCHERS = int(os.environ.get("URF_BENCH_MATCHERS", "2")) if OVERLAP == 2 else 1
    AHEAD = int(os.environ.get("URF_BENCH_SP_AHEAD", "2")) if OVERLAP == 2 else 1   # batches SuperPoint is enqueued ahead of the matcher (pipeline.py)
    NB = max(5, MATCHERS + 1 + AHEAD, int(os.environ.get("URF_BENCH_NB", "0")))      # the ring (and with it the 40-frame stream the parity tests hold oracle results for) stays at 5 batches for 2 matchers
    stream = synth.shift_stream(100, NB * BATCH * world, H, W)
    mine = [stream[(k * world + rank) * BATCH + j] for k in range(NB) for j in range(BATCH)]
    d_frames = torch.from_numpy(np.stack(mine)).to(dev)                       # [NB*8, H, W] u8
    del stream
    torch.cuda.synchronize()
    # guarded fast mode: the guard's error model is checked against the exact mode on this stream's own frames before anything is
    # timed (urf_sp_calibrate_guard_device: widens the constants if the frames need it; on these streams they hold as built)
    guard_model = None
    CALIBRATE = PREC >= 2 and not args.no_guard_calibration
    if CALIBRATE and PREC == 2:
        cal = [sp.calibrate_guard(device_ptr=d_frames[k * BATCH].data_ptr(), B=BATCH, rows=H, cols=W) for k in range(NB)]
        guard_model = {"delta_needed_by_the_stream": max(c_["delta_needed"] for c_ in cal),
                       "c_needed_by_the_stream": max(c_["c_needed"] for c_ in cal), "delta": cal[-1]["delta"], "c": cal[-1]["c"],
                       "frames_checked": NB * BATCH, "model": "|fast - exact| <= delta s (1 - s) + c eps s on the heat map"}
    F.set_profiling(True)
    # The step loop is ur-mvo_amd/pipeline.py (SlotRingPipeline; tests/test_gpu_fullsize.py runs the same object against the
    # CPU oracle).  URF_BENCH_OVERLAP: 0 = SuperPoint and the matcher on ONE in-order stream; 1 = SP(b+1) beside Sinkhorn(b)
    # (two streams); 2 (default) = three streams: two matchers alternate, so the Sinkhorn of batch b runs beside the
    # MFMA-bound GNN of batch b+1 and SuperPoint of b+2.  The host enqueues one step ahead and only waits (event) for the
    # match lists it reads.
    pms = [pm]
    for _ in range(MATCHERS - 1):
        pm_b = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=BATCH,
                               device=local_rank, precision=PREC, sinkhorn_iterations=SINK_ITERS)
        assert pm_b.build(sgb), U._lib.lib().urf_last_error()
        pms.append(pm_b)

    comm = None
    if async_exchange:
        # the exchange of include/urf.h (urf_comm_*): RCCL behind the C ABI, not torch.distributed.  ONE communicator and ONE
        # stream per rank; per step gather(b - M) then all-gather(b), the same order on every rank (pipeline.py)
        ids = [D.Comm.unique_id()] if rank == 0 else [None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        comm = D.Comm(world, rank, local_rank, ids[0])
    pipe = P.SlotRingPipeline(sp, pms, d_frames, BATCH, H, W, device=dev, rank=rank, world=world, comm=comm,
                              gloo=exchange and not async_exchange, overlap=OVERLAP, sp_ahead=AHEAD,
                              defer=int(os.environ.get("URF_BENCH_DEFER", "2")))
    ring = pipe.ring
    if CALIBRATE:
        # ... and the matcher's margin against the exact matcher, on the first pairs of the stream (urf_pm_calibrate_guard)
        guard_model = guard_model or {}
        pipe.sp_step(0); pipe.sp_step(1); sp.sync()
        c0 = [ring[0][j].data_ptr() for j in range(BATCH)]
        c1 = [ring[0][j + 1].data_ptr() for j in range(BATCH - 1)] + [ring[1][0].data_ptr()]
        try:
            calz = [m_.calibrate_guard(c0, c1) for m_ in pms]
            guard_model.update({"matcher_z_difference_on_the_stream": max(c_["z_difference"] for c_ in calz),
                                "matcher_margin": calz[-1]["margin"], "pairs_checked": BATCH})
        except RuntimeError as e_:      # (a resident Sinkhorn launch that gave up during the check: the built-in margin stays)
            guard_model.update({"matcher_check_skipped": str(e_)})

    sp_ms, conv1_ms, pm_ms, lin_ms, attn_ms, sink_ms, ransac_ms = [], [], [], [], [], [], []
    sp_stages = []
    n_matches = []
    kept = {}                      # batch index -> match lists (last steps), for the exact-mode cross-check

    def record(b, mt, res):
        age = (pipe.sp_calls - 1) - b          # SP(b) finished before match(b); how many SP calls ago?
        if 0 <= age <= 3:
            s = sp.stage_ms(age=age)
            sp_ms.append(sum(s[1:16])); conv1_ms.append(s[1]); sp_stages.append(s[:17])
        p = mt.stage_ms()
        pm_ms.append(sum(p[:7])); attn_ms.append(p[7]); lin_ms.append(p[1] + p[2] - p[7])
        sink_ms.append(p[4]); ransac_ms.append(p[6])
        n_matches.append(sum(len(r) for r in res))
        kept[b] = res

    def match_coords(res, cur, prev_last):
        """match lists as sets of pixel correspondences (x0,y0,x1,y1): keypoint ORDER may differ
        between precision modes (score-sorted, near-ties swap), coordinates do not"""
        feats = [F.slot_to_host(prev_last.data_ptr())] + [F.slot_to_host(cur[j].data_ptr()) for j in range(BATCH)]
        return [P.match_coords(res[j], feats[j], feats[j + 1]) for j in range(BATCH)]

    # prologue: SP(0) so that the loop body is exactly one match + one SP per step
    pipe.prologue()
    pipe.run(0, args.warmup)
    pipe.drain()
    sp.sync()
    # `repeats` timed regions of exactly --steps steps, each bracketed by barrier + synchronize, max over ranks
    region_s, region_local = [], []
    nb = args.warmup
    for rep in range(max(1, args.repeats)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(nb, args.steps, record)
        pipe.drain(record)
        pipe.finish_exchange(nb + args.steps - 1)     # the last batches' match lists reach rank 0 inside the timed region
        sp.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        region_local.append(time.perf_counter() - t0)
        region_s.append(D.max_over_ranks(region_local[-1], dev, world))
        nb += args.steps
    dt = float(np.median(region_s))
    args_last_batch = nb - 1
    total_frames = args.steps * BATCH * world
    fps = total_frames / dt
    gathered_total = pipe.gathered_matches_last      # the last step's matches of ALL ranks, as rank 0 received them
    # per-rank health, gathered to rank 0: a rank that fell back to the streaming Sinkhorn (or redid near-ties) is slower than
    # the others and would otherwise go unreported at N > 1
    sp_g = sp.near_tie_reruns()
    pm_g = [m.near_tie_reruns() for m in pms]
    health = {"rank": rank, "sinkhorn_fallbacks": sum(m.sinkhorn_fallbacks() for m in pms),
              "frames_redone_exact": sp_g["redone"], "frames": sp_g["frames"], "cuts_resolved": sp_g["cut_resolved"],
              "pairs_redone_exact": sum(g_["redone"] for g_ in pm_g), "pairs": sum(g_["pairs"] for g_ in pm_g),
              "pairs_flagged": sum(g_["flagged"] for g_ in pm_g),
              "comm_world": (U._lib.lib().urf_comm_world(comm._h) if comm is not None else None),   # ranks of this rank's RCCL communicator
              "region_s": [round(time_r, 4) for time_r in region_local],
              "superpoint_ms": round(float(np.mean(sp_ms)), 3) if sp_ms else None,
              "matching_ms": round(float(np.mean(pm_ms)), 3) if pm_ms else None}
    per_rank = [health]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, health)
        if async_exchange:     # every rank really sits in ONE RCCL communicator of `world` ranks
            assert all(h_["comm_world"] == world for h_ in per_rank), [h_["comm_world"] for h_ in per_rank]
    # SuperPoint stage by stage inside the timed region (guarded mode: the exact pass over the redo list and the resolution of
    # the top-k cuts run between "select" and "desc_norm" and are counted under "select")
    sp_stage_insitu = ({n_: round(float(v_), 3) for n_, v_ in zip(F.SP_STAGES, np.mean(np.array(sp_stages), axis=0))} if sp_stages else None)
    insitu = {"superpoint": float(np.mean(sp_ms)) if sp_ms else None, "matching": float(np.mean(pm_ms)),
              "linear": float(np.mean(lin_ms)), "attention": float(np.mean(attn_ms))}

    # ---- per-kernel HIP-event times for the roofline: a short SERIALISED pass (SP, then the
    # matcher, host-synchronised) right after the timed region.  In the timed region three
    # streams run concurrently, which stretches every kernel's begin->end time; rocprofv3
    # --kernel-trace serialises dispatches too, so these are the numbers it reports.
    del sp_ms[:], conv1_ms[:], pm_ms[:], lin_ms[:], attn_ms[:], sink_ms[:], ransac_ms[:]
    if rank == 0:
        bl = args_last_batch + 1
        for b in range(bl, bl + 5):
            pipe.sp_step(b)
            sp.sync()
            if exchange:
                pipe.gathered[b % NB] = ring[b % NB].repeat(world, 1)   # layout stand-in, no collective
            pms[0].wait_for_sp(sp) if OVERLAP else None
            pipe.pm_step(b, pms[0])
            pms[0].fetch(BATCH, as_arrays=True)
            s_ = sp.stage_ms(age=0)
            p_ = pms[0].stage_ms()
            sp_ms.append(sum(s_[1:16])); conv1_ms.append(s_[1])
            pm_ms.append(sum(p_[:7])); attn_ms.append(p_[7]); lin_ms.append(p_[1] + p_[2] - p_[7])
            sink_ms.append(p_[4]); ransac_ms.append(p_[6])

    if rank == 0:
        kp = [int(F.slot_to_host(ring[0][j].data_ptr()).shape[0]) for j in range(BATCH)]
        n_avg = float(np.mean(kp))
        # ---- roofline of the dominant kernel (largest summed device time per step)
        SG_WEIGHT_GB = 12003905 * 4 / 1e9                     # streamed once per step by the linear layers
        Hc_, Wc_ = H // 2, W // 2
        conv1_gb = BATCH * (H * W + Hc_ * Wc_ * 64 * 4) / 1e9   # u8 frame in, pooled 64-channel map out (4 B/element)
        per_step = {   # name: (ms, GFLOP, GB) per step
            ("conv1a+conv1b fused (h2conv_kernel<pool,fuse1a>)" if SP_FAST else "conv1a+conv1b fused (conv_mfma_kernel<9,pool,fuse1a>)"):
                (np.mean(conv1_ms), GF_CONV1 * BATCH, conv1_gb),
            ("SuperGlue linear layers (h2gemm_glds_kernel)" if FAST else "SuperGlue linear layers (gemm128 / conv_mfma_kernel<1>)"):
                (np.mean(lin_ms), sg_linear_gflop(n_avg, n_avg) * BATCH,
                 sg_linear_gbytes(n_avg, n_avg, FAST) * BATCH + SG_WEIGHT_GB),
            ("SuperGlue attention (attn_h2_kernel)" if FAST else "SuperGlue attention (attn_kernel)"):
                (np.mean(attn_ms), sg_attn_gflop(n_avg, n_avg) * BATCH, sg_attn_gbytes(n_avg, n_avg) * BATCH),
        }
        resident = FAST and os.environ.get("URF_SINKHORN_RESIDENT", "1") != "0"
        if resident:
            # chip-resident Sinkhorn (sinkhorn_resident.hip; plan tile in registers, or in LDS with URF_SINKHORN_REGS=0): the
            # couplings are read from HBM/L2 once per (re)absorption (initially and after iterations 1, 2, 4, ... 64: 8
            # times), 2 fma per element and iteration; no roof binds it -- an iteration is one inter-CU exchange (latency)
            # (the wide register-resident form: 64 rows per workgroup, one launch per batch, CUs to itself -- DESIGN.md 12)
            rs_name = "sinkhorn_wide_kernel, %d iterations in registers"
            per_step["Sinkhorn (" + rs_name % SINK_ITERS + ")"] = (
                np.mean(sink_ms), 2 * SINK_ITERS * BATCH * (n_avg + 1) ** 2 * 2 / 1e9, 8 * BATCH * (n_avg + 1) ** 2 * 4 / 1e9)
        else:
            # log-Sinkhorn: 2 passes per iteration, each streams one (n0+1) x (n1+1) f32 matrix (C or C^T) once
            # (SURVEY section 8d); ~6 flop per element (add, sub, exp, add, max)
            per_step["Sinkhorn (sinkhorn_half_kernel x %d)" % (2 * SINK_ITERS)] = (
                np.mean(sink_ms), 2 * SINK_ITERS * BATCH * (n_avg + 1) ** 2 * 6 / 1e9,
                2 * SINK_ITERS * BATCH * (n_avg + 1) ** 2 * 4 / 1e9)
        # fast mode: every product is 3 f16 MFMAs (hi*hi + hi*lo + lo*hi); the MFMA roof is priced on the
        # ALGORITHMIC flops (counted once), so its fraction is <= 1/3 by construction
        # Which roof: the dense contractions -- convolutions, linear layers, attention -- are priced on the
        # MFMA roof of the instruction they run on, with the ALGORITHMIC flops counted once (a split-f16 product is three
        # MFMAs, so those kernels cannot exceed 1/3: `mfma_issue_frac` is the same number times three); Sinkhorn on HBM.
        def on_f16(name):
            return ("h2" in name)

        def roofs(name, ms_, gf_, gb_):
            peak_tf_ = F16_MFMA_PEAK_TF if on_f16(name) else FP32_MFMA_PEAK_TF
            t_mfma, t_hbm = gf_ / peak_tf_, gb_ / HBM_PEAK_GBS * 1e3             # ms at each roof
            return ("hbm" if "inkhorn" in name else "mfma"), t_mfma, t_hbm, peak_tf_

        def frac_of(name, v):
            b_, tm_, th_, _ = roofs(name, *v)
            return (th_ if b_ == "hbm" else tm_) / v[0]       # minimum time at the family's roof / measured time

        # the kernel family reported = the one that loses the most time against its roof, ms x (1 - frac): stable from
        # run to run (the largest ms alone flips between two families that are within noise of each other)
        dom = max(per_step, key=lambda k: per_step[k][0] * (1.0 - frac_of(k, per_step[k])))
        ms, gf, gb = per_step[dom]
        bound, t_mfma, t_hbm, peak_tf = roofs(dom, ms, gf, gb)
        issue = 3 if on_f16(dom) else 1
        traffic, traffic_src = pmc_traffic(dom, args.resolution, PREC)
        mfma_busy, mfma_busy_src = pmc_mfma_busy(dom, args.resolution, PREC)
        if bound == "hbm":
            achieved, peak, unit = gb / ms * 1e3, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = gf / ms, peak_tf, "TFLOP/s"
        roofline = {"bound": bound, "kernel": dom, "achieved": round(achieved, 2), "peak": peak,
                    "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_note": traffic_src,
                    "mfma_busy_counter": mfma_busy, "mfma_busy_counter_note": mfma_busy_src,
                    "why_this_bound": f"SURVEY 8(d): dense contractions on the MFMA roof, Sinkhorn on HBM; for this family the minimum "
                                      f"time at the MFMA roof is {t_mfma:.3f} ms (algorithmic {gf:.1f} GFLOP counted once; x{issue} MFMA "
                                      f"instructions per product) and at the HBM roof {t_hbm:.3f} ms (byte model: {gb:.2f} GB per step)",
                    "tflops_logical": round(gf / ms, 2), "mfma_issue_frac": round(gf * issue / ms / peak_tf, 4),
                    "hbm_model_frac": round(gb / ms * 1e3 / HBM_PEAK_GBS, 4),
                    "launch_ms": round(float(ms), 4), "algorithmic_gflop_per_step": round(gf, 2),
                    "algorithmic_gbytes_per_step": round(gb, 3),
                    "measured": "HIP events on the library stream, serialised 5-step pass right after the timed region "
                                "(the timed region overlaps 3 streams; rocprofv3 --kernel-trace serialises as well)",
                    "in_timed_region_ms_per_step": {k: (round(v, 3) if v is not None else None) for k, v in insitu.items()},
                    "all_kernels": {k: {"ms_per_step": round(float(v[0]), 3), "tflops_logical": round(v[1] / v[0], 2),
                                        "gbytes_per_s": round(v[2] / v[0] * 1e3, 1), "bound": roofs(k, *v)[0],
                                        "frac": round(frac_of(k, v), 4), "ms_below_roof": round(float(v[0] * (1 - frac_of(k, v))), 3)}
                                    for k, v in per_step.items()}}
        if FAST and "h2gemm" in dom:
            # What paces this family: not the matrix pipe and not HBM but the STAGING of its operands into LDS.
            # Every workgroup of a launch fetches its weight tile and its activation tile (two f16 planes each = 4 B per element)
            # for the whole K; with 128 x 128 (128 x 64) tiles that is Cin x 1 KiB (768 B) per workgroup: per GNN layer and step
            tiles = int(np.ceil(n_avg / 128.0)) * 2 * BATCH
            fill = 18 * (tiles * 6 * 256 * 256 * 4 + tiles * 4 * 256 * 512 * 4 + 2 * tiles * 2 * 192 * 512 * 4) + 2 * tiles * 2 * 192 * 256 * 4
            roofline["lds_fill"] = {
                "gbytes_per_step": round(fill / 1e9, 2), "tbytes_per_s": round(fill / 1e9 / ms, 2),
                "note": "L2 -> LDS staging bytes of the 55 linear launches of a step (analytic: per workgroup (weight rows + activation rows) x "
                        "Cin x 4 B), over this family's measured time: the LDS-DMA stream of the chip sustains 6 - 7 TB/s "
                        "(MI355X_MICROARCH.md: ldsdma-fill, chip 6.4 TB/s), which is where these launches sit -- the split-f16 operands cost "
                        "4 B per element for 3 MFMAs of 16 cycles, so at 128 x 128 tiles the staging, not the matrix pipe, is the roof"}
        stage_means = {"superpoint": round(float(np.mean(sp_ms)), 3), "matching": round(float(np.mean(pm_ms)), 3),
                       "sinkhorn": round(float(np.mean(sink_ms)), 3), "ransac": round(float(np.mean(ransac_ms)), 3)}
        exact = None
        secondary = None
        if world == 1:
            # the headline's handles go NOW (explicitly: closures above still reference them, `del` alone would not destroy them):
            # their idle streams -- six in the strict mode -- otherwise share the runtime's four hardware queues with the
            # configurations measured next and slow those down by 10 - 20 %
            for m_ in pms:
                m_.__del__()
            sp.__del__()
        if world == 1 and not args.no_exact_check and NB == 5:
            # The exact fp32 mode (every tensor bit-identical to the oracle) through the same loop on the same stream, >= 20 steps
            # x 3 regions, and the headline's lists against its lists: in the strict-parity mode every pair's INDEX LIST must be
            # equal, position for position (the GPU tests assert the same against the CPU oracle).
            del pipe, pms, pm, sp
            ex_run, ex_lists = stream_run(U, spb, sgb, dev, local_rank, 0, H, W, BATCH, 20, 3, warmup=args.warmup, keep=True)
            cmp_b = [b for b in ex_lists if b in kept and b >= args.warmup]
            same = tot = same_idx = 0
            for b in cmp_b:
                for a_, x_ in zip(kept[b], ex_lists[b]):
                    tot += 1
                    same_idx += int(len(a_) == len(x_) and np.array_equal(a_["queryIdx"], x_["queryIdx"]) and np.array_equal(a_["trainIdx"], x_["trainIdx"]))
                    same += int(len(a_) == len(x_) and np.array_equal(a_["queryIdx"], x_["queryIdx"]) and np.array_equal(a_["trainIdx"], x_["trainIdx"])
                                and (len(a_) == 0 or float(np.abs(a_["distance"] - x_["distance"]).max()) < 1e-3))
            exact = {"value": ex_run["frames_per_s"], "unit": "frames/s", "dtype": "f32", "ms_per_step": ex_run["ms_per_step"],
                     "regions_frames_per_s": ex_run["regions_frames_per_s"], "steps": 20,
                     "pairs_with_identical_index_list": f"{same_idx}/{tot}",
                     "pairs_with_identical_index_list_and_distance_within_1e-3": f"{same}/{tot}"}
            if PREC == 3 and same != tot:
                print(f"bench.py: STRICT PARITY VIOLATED: {tot - same} of {tot} pairs differ from the exact mode", file=sys.stderr)
        if world == 1 and not args.no_secondary and args.resolution == "640x480" and BATCH == 8:
            # every other single-GPU configuration of BASELINE.json, in the run the driver makes (>= 3 regions each)
            secondary = {}
            for name_, prec_ in (("guarded_fast_640x480_batch8", 2), ("fast_unguarded_640x480_batch8", 1)):
                if prec_ != PREC:
                    secondary[name_] = stream_run(U, spb, sgb, dev, local_rank, prec_, 480, 640, 8, 30, 3)
            if PREC != 3:
                secondary["strict_parity_640x480_batch8"] = stream_run(U, spb, sgb, dev, local_rank, 3, 480, 640, 8, 30, 3)
            # The strict mode's rate as a function of what the guard flags.  The flag rate is a property of the matcher's WEIGHTS (how
            # far the graph layers move the descriptors, hence the split-f16 error and the margin the handle calibrates) and of
            # the data; the headline's seeded weights (synth.sg_weights: gnn_gain 0.5) are one point.  The same loop on weights
            # with 2x and 3x the residual gain, each against the exact mode on the same weights (index lists must be equal).
            curve = []
            for gain in (0.5, 1.0, 1.5):
                sgb_g = synth.pack_sg(synth.sg_weights(0, gnn_gain=gain))
                r_, lists_ = stream_run(U, spb, sgb_g, dev, local_rank, 3, 480, 640, 8, 20, 3, keep=True)
                x_, xl_ = stream_run(U, spb, sgb_g, dev, local_rank, 0, 480, 640, 8, 10, 1, keep=True)
                tot_ = same_ = 0
                for b_ in xl_:
                    if b_ in lists_ and b_ >= 5:
                        for a_, e_ in zip(lists_[b_], xl_[b_]):
                            tot_ += 1
                            same_ += int(len(a_) == len(e_) and np.array_equal(a_["queryIdx"], e_["queryIdx"]) and np.array_equal(a_["trainIdx"], e_["trainIdx"]))
                curve.append({"gnn_gain": gain, "frames_per_s": r_["frames_per_s"], "regions_frames_per_s": r_["regions_frames_per_s"],
                              "pairs": r_["pairs"], "pairs_flagged": r_["pairs_flagged"], "pairs_redone_exact": r_["pairs_redone_exact"],
                              "flag_rate": round(r_["pairs_flagged"] / max(1, r_["pairs"]), 4), "guard": r_["guard"],
                              "exact_mode_frames_per_s_same_weights": x_["frames_per_s"],
                              "pairs_with_the_exact_modes_index_list": f"{same_}/{tot_}"})
                if same_ != tot_:
                    print(f"bench.py: STRICT PARITY VIOLATED at gnn_gain {gain}: {tot_ - same_} of {tot_} pairs differ from the exact mode", file=sys.stderr)
            secondary["strict_parity_vs_flag_rate"] = {
                "what": "the strict loop (640x480, batch 8, 20 steps x 3 regions) on seeded matcher weights of growing residual gain "
                        "(synth.sg_weights(0, gnn_gain=g); 0.5 = the headline's): the margin each handle calibrates on its first pairs, "
                        "the pairs its guard flags, the rate, and the check of every list against the exact mode on the same weights",
                "curve": curve}
            secondary["native_frame_stream_strict_640x480"] = native_frame_stream_run(U, spb, sgb, local_rank, 3, 480, 640, 8, args.steps, 3,
                                                                                         kept_ref=kept if PREC == 3 else None)
            # (timed regions as long as the headline's: a region ends with a drain of the pipeline, 2 - 3 steps of latency)
            secondary["strict_parity_1241x376_batch8"] = stream_run(U, spb, sgb, dev, local_rank, 3, 376, 1241, 8, 20, 3)
            secondary["strict_parity_1241x376_batch4"] = stream_run(U, spb, sgb, dev, local_rank, 3, 376, 1241, 4, 20, 3)
            secondary["guarded_fast_1241x376_batch8"] = stream_run(U, spb, sgb, dev, local_rank, 2, 376, 1241, 8, 20, 3)
            secondary["configs1_and_per_call_path_strict_parity_640x480"] = latency_runs(U, spb, sgb, local_rank, 3, 480, 640)
            secondary["configs1_and_per_call_path_guarded_fast_640x480"] = latency_runs(U, spb, sgb, local_rank, 2, 480, 640)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            ncores = len(os.sched_getaffinity(0))
            NS = 5                                               # bounded sample: 5 frames, 4 pairs (~10-15 s)
            fr = synth.shift_stream(100, NS, H, W)
            ocfg = O.SPConfig(MAX_KP, 0.0005, 4)
            O.sp_infer(spb, ocfg, fr[0][:64, :64].copy())         # library load / thread pool start, untimed
            t = time.perf_counter()
            feats = [O.sp_infer(spb, ocfg, f) for f in fr]
            t_sp = (time.perf_counter() - t) / NS
            t = time.perf_counter()
            oms = [O.match_points(sgb, O.SGConfig(640, 512, 0.5, 100), O.ref_ransac(), feats[j], feats[j + 1], True)
                   for j in range(NS - 1)]
            t_pm = (time.perf_counter() - t) / (NS - 1)
            cpu = {"value": round(1.0 / (t_sp + t_pm), 4), "unit": "frames/s", "cores": O.threads(),
                   "kind": "port", "host_cores_visible": ncores,
                   "sample": f"{NS} frames {args.resolution} + {NS - 1} pairs of the bench stream: SuperPoint {t_sp:.2f} s/frame, "
                             f"SuperGlue+RANSAC {t_pm:.2f} s/pair (K={feats[0].shape[0]}, {len(oms[0])} matches in the first pair), "
                             f"C oracle with OpenMP"}
        out = {
            "metric": f"VO front-end frames/sec (SP+SG+RANSAC) @{args.resolution}", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "repeats": {"regions": len(region_s), "reported": "median",
                        "frames_per_s": [round(total_frames / r, 2) for r in region_s],
                        "min": round(total_frames / max(region_s), 2), "max": round(total_frames / min(region_s), 2)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (SuperPoint, fp32 MFMA) + f16x2-split on the f16 MFMA with fp32 accumulate (matcher; flagged pairs redone in f32)"
                      if PREC == 3 else "f16x2-split on the f16 MFMA, fp32 accumulate (fp32-equivalent; reference engine is TensorRT FP16)"
                      if FAST else "f32"), "data": "synthetic",
            "config": {"workload": f"{args.resolution} grayscale stream, SuperPoint + SuperGlue match + 8-pt RANSAC, "
                                   f"batch={BATCH} frames/pairs per GPU per step (BASELINE.json "
                                   f"{'configs[3]: batch 32 over 8 GPUs' if (BATCH * world == 32 and args.resolution == '1241x376') else 'configs[2]'})",
                       "resolution": args.resolution, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "max_keypoints": MAX_KP, "keypoints_per_frame": round(n_avg, 1),
                       "sinkhorn_iterations": SINK_ITERS, "ransac_iterations": 200, "precision": {0: "exact", 1: "fast", 2: "guarded fast", 3: "strict parity"}[PREC],
                       "weights": {"what": "seeded synthetic (the reference ships none): synth.sp_weights(0), synth.sg_weights(0)",
                                   "matcher_gnn_gain": args.matcher_gain,
                                   "strict_mode_pairs_flagged_and_redone": f"{sum(h_['pairs_redone_exact'] for h_ in per_rank)}/{sum(h_['pairs'] for h_ in per_rank)}",
                                   "note": "the strict mode's rate depends on the share of pairs its guard flags, a property of the weights and "
                                           "the data: secondary.strict_parity_vs_flag_rate holds the rate at other gains"},
                       "streams": {0: "one in-order stream", 1: "2 streams: SP(b+1) beside Sinkhorn(b)",
                                   2: f"{1 + MATCHERS} streams: SuperPoint enqueued {AHEAD} batches ahead, {MATCHERS} matcher handles in turn"}[OVERLAP],
                       "parallelism": f"dp{world}: frame shards + 1 RCCL all-gather of feature slots/step (urf_comm_*, C ABI) + "
                                      f"gather of the match lists to rank 0"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "exact_mode": exact,
            "secondary": secondary,
            "stage_ms_per_step": stage_means,
            "superpoint_stages_in_timed_region_ms": sp_stage_insitu,
            "matches_per_step": round(float(np.mean(n_matches)), 1),
            "matches_last_step_all_ranks_at_rank0": gathered_total,
            "sinkhorn_fallbacks": sum(h_["sinkhorn_fallbacks"] for h_ in per_rank),   # resident launches redone with the streaming kernels, all ranks
            # guarded fast mode: frames / pairs redone in the exact mode because a discrete decision sat within the fast mode's
            # error (all ranks, warm-up and timed regions; the reruns are part of the timed work)
            "near_tie_reruns": {"frames": sum(h_["frames_redone_exact"] for h_ in per_rank), "of_frames": sum(h_["frames"] for h_ in per_rank),
                                "pairs": sum(h_["pairs_redone_exact"] for h_ in per_rank), "of_pairs": sum(h_["pairs"] for h_ in per_rank),
                                "pairs_flagged_not_redone": sum(h_["pairs_flagged"] - h_["pairs_redone_exact"] for h_ in per_rank),
                                "frames_with_cut_resolved_per_candidate": sum(h_["cuts_resolved"] for h_ in per_rank),
                                "superpoint_causes": {k: sp_g[k] for k in ("cut_resolved", "candidates", "threshold", "nms", "cut_overflow")},
                                "matcher_causes": {k: sum(g_[k] for g_ in pm_g) for k in ("threshold", "runner_up")}},
            "guard_model": guard_model,
            "per_rank": per_rank,
            "library": U._lib.lib().urf_build_info().decode(),
        }
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
