"""Is an LDS value that one wave writes and publishes with a barrier always what the other waves read -- beside kernels of other
streams that fill their LDS by DMA?  (experiments build)  A hand-off kernel (256 workgroups of 4 waves, 4.5 KB of LDS) repeats the
LDS traffic of the register-resident Sinkhorn's iteration with self-describing values for `ms` milliseconds per round while the exact
SuperPoint (LDS-DMA weights) runs on another stream.     python tools/gpu_lds_handoff.py [ms=200] [rounds=20] [alone=0]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
ms = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
alone = len(sys.argv) > 3 and sys.argv[3] == "1"
H, W, B = 376, 1241, 8
L = U._lib.lib()
print(L.urf_build_info().decode())
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=0)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(np.stack(synth.shift_stream(100, B, H, W))).to(dev)
slots = torch.zeros((B, L.urf_slot_bytes() // 4), dtype=torch.float32, device=dev)
bad = torch.zeros(4, dtype=torch.int64, device=dev)
first = torch.zeros(8, dtype=torch.int32, device=dev)
poll = torch.arange(2048, dtype=torch.int64, device=dev)
ws = torch.cuda.Stream(device=dev)
L.urf_probe_lds_handoff.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
torch.cuda.synchronize()
for r in range(rounds):
    assert L.urf_probe_lds_handoff(0, 256, ms, ws.cuda_stream, bad.data_ptr(), first.data_ptr(), poll.data_ptr()) == 0
    if not alone:
        for _ in range(int(ms / 4.0) + 1):
            sp.infer_device(d_frames.data_ptr(), B, H, W, slots.data_ptr())
        sp.sync()
    torch.cuda.synchronize()
b = bad.cpu().numpy()
print(f"hand-off: {rounds} rounds of {ms} ms {'alone' if alone else 'beside the exact SuperPoint'}: {int(b[3])} iterations in all workgroups, "
      f"wrong words {int(b[0])} (of them the previous iteration's value: {int(b[1])}), workgroups hit {int(b[2])}, "
      f"first (workgroup, hand-off, index, iteration, got, want) {first.cpu().numpy().view(np.uint32)[:6].tolist()}")
