"""Bit-equality of the attention kernel variants of the experiments build (URF_ATTN_IL=0: the product's, 1 / 2: software-pipelined):
runs the fast matcher (precision 1) on seeded feature pairs of ragged sizes and prints a sha256 of every pair's log-assignment
matrix and index lists.  Run per variant (URF_LIB=.../liburf_front_exp.so) and diff the output:
    tools/gpu_attn_il_check.sh"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402

U = load_pkg()
F, synth = U.frontend, U.synth
sgb = synth.pack_sg(synth.sg_weights(0))
sg = F.SuperGlue(F.SuperGlueConfig(), precision=1)
assert sg.build(sgb)
pm = F.PointMatching(F.SuperGlueConfig(), max_pairs=8, precision=1)
assert pm.build(sgb)
rng = np.random.default_rng(7)
sizes = [(1024, 1024), (1000, 1000), (1, 1), (63, 64), (64, 65), (65, 1024), (512, 300), (129, 127), (1024, 17), (960, 1023)]
for n0, n1 in sizes:
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=min(n0, n1) * 6 // 10)
    nf0 = F.PointMatching.NormalizeKeypoints(None, f0, 640, 512)
    nf1 = F.PointMatching.NormalizeKeypoints(None, f1, 640, 512)
    out = sg.infer(nf0, nf1)
    h = hashlib.sha256()
    for o in out:
        h.update(np.ascontiguousarray(o).tobytes())
    print(n0, n1, h.hexdigest()[:24], int((np.asarray(out[0]) >= 0).sum()))
# the batched path (16 images: the 2-tile kernel) on SuperPoint's slots of a synthetic stream
import torch  # noqa: E402
H, W, B = 480, 640, 8
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B + 1, precision=1)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
frames = synth.shift_stream(100, B + 1, H, W)
d = torch.from_numpy(np.stack(frames)).cuda()
slots = torch.zeros((B + 1, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
sp.infer_device(d[0].data_ptr(), B + 1, H, W, slots[0].data_ptr())
sp.sync()
for rep in range(2):
    pm.match_device_async([slots[j].data_ptr() for j in range(B)], [slots[j + 1].data_ptr() for j in range(B)], True)
    res = pm.fetch(B, as_arrays=True)
    h = hashlib.sha256()
    for r in res:
        h.update(np.ascontiguousarray(r).tobytes())
    print("batch", rep, h.hexdigest()[:24], [len(r) for r in res])
