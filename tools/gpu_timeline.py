"""Where does a pipelined step go?  Host-side timestamps of the bench loop (ur-mvo_amd/pipeline.py) next to the GPU stage times
of every batch: how long the host spends enqueueing the matcher call / the SuperPoint call, how long it waits in fetch, and how
busy each stream is.    python tools/gpu_timeline.py [precision=3] [steps=60] [sp_ahead=2] [matchers=2] [defer=2]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth, P = U.frontend, U.synth, U.pipeline
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ahead = int(sys.argv[3]) if len(sys.argv) > 3 else 2
M = int(sys.argv[4]) if len(sys.argv) > 4 else 2
defer = int(sys.argv[5]) if len(sys.argv) > 5 else 2
H, W, B = 480, 640, 8
print(U._lib.lib().urf_build_info().decode())
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, precision=prec)
assert sp.build(spb)
pms = []
for _ in range(M):
    pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=B, precision=prec)
    assert pm.build(sgb)
    pms.append(pm)
NB = max(5, M + 1 + ahead)
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(np.stack(synth.shift_stream(100, NB * B, H, W))).to(dev)
F.set_profiling(True)
pipe = P.SlotRingPipeline(sp, pms, d_frames, B, H, W, device=dev, sp_ahead=ahead, defer=defer)
pipe.prologue()
pipe.run(0, 6)
rows = []
pm_stage, sp_stage, flagged = {}, {}, {}


def rec(b, mt, res):
    p = mt.stage_ms()
    pm_stage[b] = (sum(p[:7]), p[8] if len(p) > 8 else 0.0)
    flagged[b] = sum(1 for f in mt.near_tie_flags(B) if f)
    age = (pipe.sp_calls - 1) - b
    if 0 <= age <= 3:
        sp_stage[b] = sum(sp.stage_ms(age=age)[1:16])


wait = [0.0]
for m_ in pms:      # time spent blocked in the two halves of fetch
    for name in ("fetch_begin", "fetch_end"):
        def wrap(f):
            def g(*a_, **k_):
                t_ = time.perf_counter()
                r_ = f(*a_, **k_)
                wait[0] += time.perf_counter() - t_
                return r_
            return g
        setattr(m_, name, wrap(getattr(m_, name)))
torch.cuda.synchronize()
T0 = time.perf_counter()
for b in range(6, 6 + steps):
    t0 = time.perf_counter()
    wait[0] = 0.0
    pipe.one_step(b, rec)
    t3 = time.perf_counter()
    rows.append((b, (t3 - t0 - wait[0]) * 1e3, 0.0, wait[0] * 1e3, (t3 - T0) * 1e3))
pipe.drain(rec)
sp.sync()
torch.cuda.synchronize()
total = (time.perf_counter() - T0) * 1e3
print(f"precision {prec}, {steps} steps, sp_ahead {ahead}, {M} matchers, hand-out lag <= {defer}: {total / steps:.3f} ms/step = {steps * B / total * 1e3:.1f} frames/s")
print("step  enqueue  -  wait-in-fetch   t_end | GPU: SP(b) ms  match(b) fast ms  redo ms  flagged pairs")
for b, a, c, w, t in rows[:40]:
    ps = pm_stage.get(b, (0, 0))
    print(f"{b:4d}  {a:8.3f}  {c:8.3f}  {w:8.3f}  {t:8.2f} | {sp_stage.get(b, 0):6.2f}  {ps[0]:6.2f}  {ps[1]:6.2f}  {flagged.get(b, 0)}")
a = np.array([r[1:4] for r in rows])
print("mean host ms per step: enqueue %.3f, (unused %.3f), blocked in fetch %.3f" % tuple(a.mean(0)))
print("mean GPU ms per batch: SuperPoint %.2f, matcher fast pass %.2f, redo %.2f (batches with a flagged pair: %d of %d)" % (
    np.mean(list(sp_stage.values())), np.mean([v[0] for v in pm_stage.values()]), np.mean([v[1] for v in pm_stage.values()]),
    sum(1 for v in flagged.values() if v), len(flagged)))
rd = [pm_stage[b][1] for b in pm_stage if flagged.get(b)]
nr = [pm_stage[b][1] for b in pm_stage if not flagged.get(b)]
if rd:
    print("redo pass, batches with a flagged pair: mean %.2f ms (min %.2f max %.2f); without: mean %.2f ms" % (np.mean(rd), min(rd), max(rd), np.mean(nr) if nr else 0))
