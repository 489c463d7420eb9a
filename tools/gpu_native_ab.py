import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, numpy as np
from __graft_entry__ import load_pkg
U = load_pkg()
spb, sgb = U.synth.pack_sp(U.synth.sp_weights(0)), U.synth.pack_sg(U.synth.sg_weights(0))
for steps in (30, 100, 30, 100):
    r = bench.native_frame_stream_run(U, spb, sgb, 0, 3, 480, 640, 8, steps, 3)
    print(steps, r["frames_per_s"], r["regions_frames_per_s"], flush=True)
import torch
dev = torch.device("cuda", 0)
for steps in (30, 100):
    r = bench.stream_run(U, spb, sgb, dev, 0, 3, 480, 640, 8, steps, 3)
    print("pipeline.py", steps, r["frames_per_s"], r["regions_frames_per_s"], flush=True)
