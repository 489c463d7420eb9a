"""Where an iteration of the LDS-resident Sinkhorn spends its cycles: s_memtime stamps of workgroup 0
(urf_probe_sinkhorn_stamps).    python tools/gpu_sinkhorn_stamps.py"""
import ctypes as C
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
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=1)
assert sp.build(spb)
pm = F.PointMatching(F.SuperGlueConfig(), max_pairs=B, precision=1)
assert pm.build(sgb)
frames = synth.shift_stream(100, B + 1, H, W)
d = torch.from_numpy(np.stack(frames)).cuda()
slots = torch.zeros((B + 1, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
sp.infer_device(d[0].data_ptr(), B + 1 if B < 8 else 1, H, W, slots[0].data_ptr())
if B >= 8:
    sp.infer_device(d[1].data_ptr(), B, H, W, slots[1].data_ptr())
sp.sync()
L = U._lib.lib()
for rep in range(3):
    if rep == 2:
        assert L.urf_probe_sinkhorn_stamps(1, 0, None) == 0
    pm.match_device_async([slots[j].data_ptr() for j in range(B)], [slots[j + 1].data_ptr() for j in range(B)], True)
    pm.fetch(B)
raw = np.zeros(100 * 10, np.int64)
assert L.urf_probe_sinkhorn_stamps(0, 100, raw.ctypes.data_as(C.c_void_p)) == 0
out = raw[:800].reshape(100, 8)
passes = raw[800:].reshape(100, 2)
names = ["row pass", "barrier", "a_dust + column pass + publish", "hop 1 (reduce, wave 0)", "hop 2 (sweep, wave 0)", "barrier", "b update",
         "to next iteration (absorb on 1,2,4,..)"]
dt = np.diff(np.concatenate([out, np.r_[out[1:, :1], out[-1:, -1:]]], axis=1), axis=1).astype(np.float64)
sel = [k for k in range(100) if (k + 1) not in (1, 2, 4, 8, 16, 32, 64) and k < 99]
print("s_memtime ticks (100 MHz constant clock on gfx950? compare with the total) per phase, median over plain iterations:")
for i, n in enumerate(names):
    print(f"  {n:42s} {np.median(dt[sel, i]):9.1f}   (absorb iterations: {np.median(dt[[0, 1, 3, 7, 15, 31, 63], i]):9.1f})")
print("  iteration total", np.median(dt[sel].sum(1)), " whole loop", out[-1, -1] - out[0, 0])
print("  poll passes per hop (median): hop 1", np.median(passes[:, 0]), " hop 2", np.median(passes[:, 1]))
