"""Does a co-resident kernel ever write into another workgroup's LDS?  (experiments build)  A watcher kernel -- 256 workgroups,
`lds` bytes of LDS each filled with a pattern and re-checked for `ms` milliseconds -- runs on one stream while the exact
SuperPoint (LDS-DMA weights) and, optionally, the strict matcher loop on others.    python tools/gpu_lds_watch.py [lds=6144] [ms=200] [rounds=20]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
lds = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, W, B = 376, 1241, 8
L = U._lib.lib()
print(L.urf_build_info().decode())
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=0)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(np.stack(synth.shift_stream(100, B, H, W))).to(dev)
slots = torch.zeros((B, L.urf_slot_bytes() // 4), dtype=torch.float32, device=dev)
bad = torch.zeros(2, dtype=torch.int64, device=dev)
first = torch.zeros(4, dtype=torch.int32, device=dev)
ws = torch.cuda.Stream(device=dev)
L.urf_probe_lds_watch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
torch.cuda.synchronize()
for r in range(rounds):
    assert L.urf_probe_lds_watch(0, 256, lds, ms, ws.cuda_stream, bad.data_ptr(), first.data_ptr()) == 0
    for _ in range(int(ms / 4.0) + 1):
        sp.infer_device(d_frames.data_ptr(), B, H, W, slots.data_ptr())
    sp.sync()
    torch.cuda.synchronize()
print(f"watcher: {rounds} rounds of {ms} ms, {lds} B of LDS per workgroup beside the exact SuperPoint: corrupted words {int(bad[0])}, workgroups hit {int(bad[1])}, first {first.cpu().numpy().view(np.uint32)}")
