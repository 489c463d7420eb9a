"""The dropped LDS wait at run time (experiments build): tools/repro/dropped_lds_wait.hip as hipcc compiles it (a bare s_barrier at the
loop header) and with the wait written out, alone on the chip and beside the exact SuperPoint (LDS-DMA weight stages, 16-byte
fragment reads on every CU).  Every workgroup computes the same deterministic recurrence; an output that differs from the reference
run is a workgroup in which some wave read the previous iteration's misc[0].
    python tools/gpu_back_edge_repro.py [rounds=30] [iters=4000] [wgs=1024]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
H, W, B = 376, 1241, 8
L = U._lib.lib()
print(L.urf_build_info().decode())
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=0)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(np.stack(synth.shift_stream(100, B, H, W))).to(dev)
slots = torch.zeros((B, L.urf_slot_bytes() // 4), dtype=torch.float32, device=dev)
x = torch.from_numpy(np.random.default_rng(5).uniform(0.5, 1.5, 256).astype(np.float32)).to(dev)
out = torch.zeros(wgs * 256, dtype=torch.float32, device=dev)
ws = torch.cuda.Stream(device=dev)
L.urf_probe_back_edge.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def run(fixed, beside):
    out.zero_()
    torch.cuda.synchronize()
    assert L.urf_probe_back_edge(fixed, wgs, iters, x.data_ptr(), out.data_ptr(), ws.cuda_stream) == 0
    if beside:
        for _ in range(6):
            sp.infer_device(d_frames.data_ptr(), B, H, W, slots.data_ptr())
        sp.sync()
    torch.cuda.synchronize()
    return out.view(wgs, 256).clone()


ref = run(1, False)
assert torch.equal(ref, ref[0:1].expand_as(ref)), "the reference run itself is not uniform over the workgroups"
for name, fixed, beside in (("as compiled, alone", 0, False), ("wait written out, alone", 1, False),
                            ("as compiled, beside the exact SuperPoint", 0, True), ("wait written out, beside the exact SuperPoint", 1, True)):
    bad = 0
    worst = 0.0
    for _ in range(rounds):
        o = run(fixed, beside)
        d = (o != ref).any(dim=1)
        bad += int(d.sum())
        if d.any():
            worst = max(worst, float((o - ref).abs().max()))
    print(f"{name}: {rounds} launches of {wgs} workgroups x {iters} iterations: {bad} workgroups with a different result (largest difference {worst:.3g})")
