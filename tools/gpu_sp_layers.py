"""Per-layer time and fp32-MFMA efficiency of the exact SuperPoint at batch 8 (serialised, HIP events of the library).
    python tools/gpu_sp_layers.py [H W]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
B = 8
print(U._lib.lib().urf_build_info().decode())
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=0)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
d = torch.from_numpy(np.stack(synth.shift_stream(100, B, H, W))).cuda()
slots = torch.zeros((B, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
F.set_profiling(True)
acc = []
for rep in range(6):
    sp.infer_device(d.data_ptr(), B, H, W, slots.data_ptr()); sp.sync()
    if rep:
        acc.append(sp.stage_ms())
ms = np.mean(np.array(acc), axis=0)
s = (H * W) / (480 * 640)
gf = {"conv1a+1b": 23.003, "conv2a": 5.662, "conv2b": 5.662, "conv3a": 2.831, "conv3b": 5.662, "conv4a": 1.416, "conv4b": 1.416,
      "convPa|Da": 5.662, "convPb": 0.160, "convDb": 0.629}
tot = 0.0
for name, t in zip(F.SP_STAGES, ms):
    g = gf.get(name)
    tot += t
    print(f"{name:12s} {t * 1e3:8.1f} us" + (f"   {g * s * B / t:7.1f} TFLOP/s = {g * s * B / t / 157.3 * 100:5.1f} % of the fp32 MFMA peak" if g else ""))
print(f"sum {tot:.3f} ms; convolutions {sum(t for n, t in zip(F.SP_STAGES, ms) if n in gf):.3f} ms")
