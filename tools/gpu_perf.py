"""Isolated stage timings (no SP/PM overlap): python tools/gpu_perf.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg()
F, synth = U.frontend, U.synth
H, W, B = 480, 640, int(os.environ.get("URF_B", "8"))
spb = synth.pack_sp(synth.sp_weights(0))
sgb = synth.pack_sg(synth.sg_weights(0))
PREC = int(os.environ.get('URF_PRECISION', '0'))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=PREC)
assert sp.build(spb)
pm = F.PointMatching(F.SuperGlueConfig(), max_pairs=B, precision=PREC)
assert pm.build(sgb)
frames = synth.shift_stream(100, B + 1, H, W)
d = torch.from_numpy(np.stack(frames)).cuda()
slots = torch.zeros((B + 1, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
F.set_profiling(True)
sp.infer_device(d[0].data_ptr(), 1, H, W, slots[0].data_ptr())
sp.sync()
acc_s, acc_p = [], []
for it in range(6):
    sp.infer_device(d[1].data_ptr(), B, H, W, slots[1].data_ptr())
    sp.sync()
    acc_s.append(sp.stage_ms())
    pm.match_device_async([slots[j].data_ptr() for j in range(B)], [slots[j + 1].data_ptr() for j in range(B)], True)
    res = pm.fetch(B)
    acc_p.append(pm.stage_ms())
s = np.median(np.array(acc_s[1:]), axis=0)
p = np.median(np.array(acc_p[1:]), axis=0)
print("SP  total %.3f ms / %d frames:" % (s[1:16].sum(), B), ", ".join(f"{n}={v:.3f}" for n, v in zip(F.SP_STAGES, s)))
print("PM  total %.3f ms / %d pairs:" % (p[:7].sum(), B), ", ".join(f"{n}={v:.3f}" for n, v in zip(F.PM_STAGES, p)))
print("matches", [len(r) for r in res])
gf_conv1 = 23.003 * B
n = 1000
lin = (2 * (3 * 32 + 32 * 64 + 64 * 128 + 128 * 256 + 256 * 256) + 18 * 2 * (3 * 65536 + 65536 + 262144 + 131072) + 2 * 65536) * 2 * n / 1e9 * B
att = 9 * 4 * 256 * (4 * n * n) / 1e9 * B
print("TFLOP/s: conv1 %.1f  linear %.1f  attn %.1f" % (gf_conv1 / s[1], lin / (p[1] + p[2] - p[7]), att / p[7]))

# ---- Camera::UndistortImage stage: HBM-bound gather, 6 map bytes + 1 source + 1 output byte/pixel
import time
K = np.array([[420.5, 0, 318.2], [0, 419.1, 242.7], [0, 0, 1]])
cam = F.Camera(W, H, K, [-0.28, 0.07, 1e-3, -2e-3, 0.01])
d_und = torch.zeros_like(d[:B])
for reps in (3, 200):
    cam.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        cam.undistort_device(d[0].data_ptr(), B, H, W, d_und.data_ptr())
    cam.sync()
    dt = (time.perf_counter() - t0) / reps
print("undistort %d frames %dx%d: %.1f us/launch, %.0f GB/s algorithmic (8 B/pixel)" % (B, W, H, dt * 1e6, B * H * W * 8 / dt / 1e9))

# ---- native frame stream (urf_fe_*): host frames in, match lists out = the PCIe-inclusive rate
frames_h = np.stack(synth.shift_stream(100, 5 * B, H, W))
for with_cam in (False, True):
    fs = F.FrameStream(F.SuperPointConfig(max_keypoints=1000), F.SuperGlueConfig(), batch=B, max_height=H, max_width=W,
                       precision=PREC)
    assert fs.build(spb, sgb)
    if with_cam:
        fs.set_camera(cam)
    nb, t_sub, t_col = 40, 0.0, 0.0
    for b in range(nb + 2):
        if b == 2:
            t0 = time.perf_counter()
            t_sub = t_col = 0.0
        k = b % 5
        ta = time.perf_counter()
        fs.submit(frames_h[k * B:(k + 1) * B])
        tb = time.perf_counter()
        while fs.in_flight() > 6 or (fs.in_flight() and fs.ready()):     # the loop of include/urf.h: matchers + 4 stay in flight
            fs.collect()
        t_sub += tb - ta
        t_col += time.perf_counter() - tb
    while fs.in_flight():
        fs.collect()
    dt = time.perf_counter() - t0
    print("frame stream (host u8 frames -> match lists%s): %.0f frames/s; host per batch: submit %.2f ms, collect (wait + deferred enqueue) %.2f ms"
          % (", undistort in front" if with_cam else "", nb * B / dt, 1e3 * t_sub / nb, 1e3 * t_col / nb))
    del fs
