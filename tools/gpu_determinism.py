"""Oracle-free determinism check of the benched loop: the stream is periodic (5 batches), so batch b and batch b + 5 must
produce identical match lists and identical slots, in every mode.  A mismatch is a race (or an order-dependent reduction).
    python tools/gpu_determinism.py [steps] [precisions, e.g. 3,2,1]        env: URF_SP_TWO_STREAMS, URF_SINKHORN_RESIDENT, ...
A long run (steps in the thousands) is the soak test of a mode: only the last five batches' lists are kept."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth, P = U.frontend, U.synth, U.pipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 26
precs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 2, 1]
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
print(U._lib.lib().urf_build_info().decode(), "URF_SP_TWO_STREAMS =", os.environ.get("URF_SP_TWO_STREAMS", "(unset: off)"))
sizes = [(376, 1241), (480, 640)] if len(sys.argv) <= 3 else [tuple(int(v) for v in sys.argv[3].split('x'))[::-1]]
for (H, W) in sizes:
    frames = synth.shift_stream(100, 40, H, W)
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    for prec in precs:
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=8, precision=prec)
        assert sp.build(spb)
        pms = []
        for _ in range(2):
            pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=8, precision=prec,
                                 sinkhorn_residual_bound=float(os.environ.get("URF_SOAK_RESID_BOUND", "0")),
                                 audit_period=int(os.environ.get("URF_SOAK_AUDIT", "0")))   # (0 = the default 256; < 0 = no audits)
            assert pm.build(sgb)
            pms.append(pm)
        pipe = P.SlotRingPipeline(sp, pms, d_frames, 8, H, W, device=dev, defer=int(os.environ.get('URF_BENCH_DEFER', '3')), sp_ahead=int(os.environ.get('URF_BENCH_SP_AHEAD', '2')))
        pipe.prologue()
        lists, slots = {}, {}
        bad = [0]
        bad_c = [0]
        audited = [0]

        cks = {}
        RSD = os.environ.get("URF_RS_DEBUG") == "1"        # experiments build: per-iteration bit sums inside the resident Sinkhorn
        dbg = {}
        CKS = os.environ.get("URF_CHECKSUMS") == "1"     # experiments build + URF_BENCH_DEFER=0: per-stage checksums of every batch
        CKN = ("encoded keypoints", "projected descriptors", "couplings", "u", "v")

        resid = {}
        rmax = [0.0]
        integ0 = [m.sinkhorn_integrity()["pairs"] for m in pms]

        def rec(b, mt, res):
            lists[b] = [r.copy() for r in res]
            resid[b] = mt.sinkhorn_residuals(8)
            rmax[0] = max(rmax[0], max(resid[b]))
            resid.pop(b - 10, None)
            if CKS:
                arr = (ctypes.c_ulonglong * 8)()
                U._lib.lib().urf_probe_pm_checksums(mt._h, arr)
                cks[b] = tuple(arr[:5])
                if b >= 6 and cks[b] != cks[b - 5] and RSD:
                    # fetch the per-iteration bit sums of this (suspect) launch only now: a copy per batch changes the timing enough to hide the fault
                    d = np.zeros(8 * 128 * 32 * 2, dtype=np.uint64)
                    U._lib.lib().urf_probe_pm_rs_debug(mt._h, ctypes.c_void_p(d.ctypes.data))
                    d = d.reshape(8, 128, 32, 2)
                    for pp in range(8):
                        cs = d[pp, :100, :, 1]
                        incons = np.nonzero((cs != cs[:, :1]).any(axis=1))[0]
                        if incons.size:
                            k0 = int(incons[0])
                            vals, counts = np.unique(cs[k0], return_counts=True)
                            odd = np.nonzero(cs[k0] != vals[np.argmax(counts)])[0]
                            print(f"   batch {b} pair {pp}: the 32 workgroups DISAGREE on the reduced column sums in {incons.size} iterations, first {k0 + 1}: workgroups {odd.tolist()} hold another value than the majority")
                    dbg[b] = d
                if b >= 6 and cks[b] != cks[b - 5]:
                    bad_c[0] += 1
                    if bad_c[0] <= 30:
                        print(f"   {W}x{H} precision {prec}: batch {b} checksums differ from batch {b - 5}: " + ", ".join(n for n, x, y in zip(CKN, cks[b], cks[b - 5]) if x != y))
                cks.pop(b - 10, None)
            if b >= 6:
                for j in range(8):
                    if not np.array_equal(lists[b][j], lists[b - 5][j]):
                        x, y = lists[b][j], lists[b - 5][j]
                        if prec == 3 and len(x) == len(y) and np.array_equal(x["queryIdx"], y["queryIdx"]) and np.array_equal(x["trainIdx"], y["trainIdx"]) \
                                and float(np.abs(x["distance"] - y["distance"]).max()) < 1e-3 and sum(m.guard_state()["audits"] for m in pms) > 0:
                            audited[0] += 1      # an audited pair (handed out from the exact engine): same index list, distances within 1e-3
                            continue
                        bad[0] += 1
                        if bad[0] <= 20:
                            x, y = lists[b][j], lists[b - 5][j]
                            same_idx = len(x) == len(y) and np.array_equal(x["queryIdx"], y["queryIdx"]) and np.array_equal(x["trainIdx"], y["trainIdx"])
                            dd = float(np.abs(x["distance"] - y["distance"]).max()) if same_idx and len(x) else -1.0
                            print(f"   {W}x{H} precision {prec}: batch {b} pair {j}: {len(x)} vs {len(y)} matches (batch {b - 5}); same index lists {same_idx}, max |distance difference| {dd:.3g}; "
                                  f"flags {mt.near_tie_flags(8)}, Sinkhorn fallbacks {[m.sinkhorn_fallbacks() for m in pms]}; "
                                  f"column-marginal residual {resid[b][j]:.3g} (batch {b}) / {resid[b - 5][j]:.3g} (batch {b - 5})")
            lists.pop(b - 10, None)

        for b in range(steps):
            pipe.one_step(b, rec)
            if os.environ.get("URF_DET_SYNC") == "1":
                # snapshot of the slots SuperPoint(b + 1) has just been asked to write (synchronises: changes the timing)
                sp.sync()
                slots[b + 1] = pipe.ring[(b + 1) % 5].clone()
        pipe.drain(rec)
        bad_l, bad_s = bad[0], 0
        for b in range(6, steps + 1):
            if b in slots and b - 5 in slots and not torch.equal(slots[b], slots[b - 5]):
                d = (slots[b] != slots[b - 5]).any(dim=1).nonzero().flatten().tolist()
                bad_s += 1
                for j in d:
                    x, y = slots[b][j].cpu().numpy(), slots[b - 5][j].cpu().numpy()
                    k = np.nonzero(x != y)[0]
                    print(f"   {W}x{H} precision {prec}: slots of batch {b} frame {j} differ from batch {b - 5} in {k.size} words, first at {k[:4]}"
                          f" (header {x[:4].view(np.int32)} vs {y[:4].view(np.int32)})")
        if hasattr(U._lib.lib(), "urf_probe_rs_verify"):
            n_ = ctypes.c_ulonglong(0)
            U._lib.lib().urf_probe_rs_verify(ctypes.byref(n_))
            print(f"   load verification inside the register Sinkhorn: {n_.value & 0xFFFFFFFF} couplings read differently by two loads, {n_.value >> 32} column sums read back from LDS differently than written (cumulative)")
        print(f"   largest column-marginal residual of any handed-out pair {rmax[0]:.3g}; pairs redone by the integrity check "
              f"{sum(m.sinkhorn_integrity()['pairs'] for m in pms) - sum(integ0)} (bound {pms[0].sinkhorn_integrity()['bound']:.3g})")
        if prec == 3:
            gs = [m.guard_state() for m in pms]
            print(f"   online guard check: margin {max(g['margin'] for g in gs):.3g}, largest fast-vs-exact difference on a redone pair {max(g['online_worst'] for g in gs):.3g} "
                  f"over {sum(g['online_pairs'] for g in gs)} pairs, {sum(g['margin_raises'] for g in gs)} raises, {sum(g['online_violations'] for g in gs)} violations; "
                  f"{sum(g['audits'] for g in gs)} unflagged pairs audited, {sum(g['audit_mismatches'] for g in gs)} with another index list than the fast pass "
                  f"({audited[0]} list comparisons differ in the distances of an audited pair only); batches diverted to the exact mode {sum(g['exact_batches'] for g in gs)}")
        print(f"{W}x{H} precision {prec}: {steps} steps, checksum mismatches {bad_c[0]}, list mismatches {bad_l}, slot-batch mismatches {bad_s}, guard {sp.near_tie_reruns()}, pairs redone {sum(m.near_tie_reruns()['redone'] for m in pms)}")
        del pipe, sp, pms
