"""Does a configuration run slower when another one ran earlier in the same process?  (bench.py measures its secondary
configurations after the headline in one process.)    python tools/gpu_sequence_probe.py [order of precisions, e.g. 3213] [prof]"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); synth = U.synth
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
dev = torch.device("cuda", 0)
bench.MAX_KP = 1000
NAMES = {0: "exact", 1: "fast", 2: "guarded", 3: "strict"}
order = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "232132")]
if len(sys.argv) > 2 and sys.argv[2] == "prof":
    U.frontend.set_profiling(True)
# argv[3] = n: that many streams created (and kept, idle) first -- shifts the phase in which the HIP runtime deals the library's
# streams onto its hardware queues
dummies = [torch.cuda.Stream(device=dev) for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 0)]
for prec in order:
    name = NAMES[prec]
    r = bench.stream_run(U, spb, sgb, dev, 0, prec, 480, 640, 8, 30, 3)
    gc.collect()
    print(name, r["frames_per_s"], r["regions_frames_per_s"], flush=True)
