"""Host-side enqueue cost of one matcher / SuperPoint call (is the pipeline launch-bound?)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
U = load_pkg(); F, synth = U.frontend, U.synth
H, W, B = 480, 640, 8
PREC = int(os.environ.get('URF_PRECISION', '1'))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=PREC)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
pm = F.PointMatching(F.SuperGlueConfig(), max_pairs=B, precision=PREC)
assert pm.build(synth.pack_sg(synth.sg_weights(0)))
frames = synth.shift_stream(100, B + 1, H, W)
d = torch.from_numpy(np.stack(frames)).cuda()
slots = torch.zeros((B + 1, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device="cuda")
sp.infer_device(d[0].data_ptr(), 1, H, W, slots[0].data_ptr()); sp.sync()
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.infer_device(d[1].data_ptr(), B, H, W, slots[1].data_ptr())
    t1 = time.perf_counter()
    sp.sync()
    t2 = time.perf_counter()
    pm.match_device_async([slots[j].data_ptr() for j in range(B)], [slots[j + 1].data_ptr() for j in range(B)], True)
    t3 = time.perf_counter()
    res = pm.fetch(B, as_arrays=True)
    t4 = time.perf_counter()
    print(f"SP enqueue {1e3*(t1-t0):.3f} ms (done after {1e3*(t2-t0):.3f});  matcher enqueue {1e3*(t3-t2):.3f} ms (done after {1e3*(t4-t2):.3f})")
