"""How far is the fast mode from the exact mode at every DISCRETE decision of the path, and how often does a decision sit
closer to its threshold than that?  Input of the near-tie guard's constants (DESIGN.md "Guarded fast mode").

Per frame of the bench stream, both sizes: relative error of the heat map (pre-NMS) where it can matter (> threshold),
NMS support differences, gaps between consecutive candidates in score order (the top-k cut and the order inside the top k),
distance of candidates to the 0.0005 threshold.  Per pair: error of the log-assignment on entries that can become a match,
number of mutual maxima whose probability lies within delta of the 0.5 threshold.

    python tools/gpu_margins.py [N frames per size, default 24] > gpurun_out/margins.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
THR, K = 0.0005, 1000
out = {}
for (H, W) in ((480, 640), (376, 1241)):
    Hs, Ws = H // 8 * 8, W // 8 * 8
    frames = synth.shift_stream(100, N, H, W)
    sps = []
    for prec in (0, 1):
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=K), max_height=H, max_width=W, precision=prec)
        assert sp.build(spb)
        sps.append(sp)
    rec = {"heat_rel_err_max": [], "heat_rel_err_p999": [], "nms_support_diff": [], "cand": [], "cut_gap_rel": [],
           "min_adjacent_gap_rel_in_topk": [], "n_adjacent_gaps_below": {"1.2e-7": 0, "2.4e-7": 0, "5e-7": 0, "1e-6": 0, "3e-6": 0, "1e-5": 0},
           "cut_gap_below": {"1.2e-7": 0, "2.4e-7": 0, "5e-7": 0, "1e-6": 0, "3e-6": 0, "1e-5": 0},
           "topk_score_rel_err_max": [], "kp_set_diff": [], "kp_order_diff": [], "thr_margin_rel_min": []}
    feats = {0: [], 1: []}
    for fr in frames:
        heat, score, f = {}, {}, {}
        for prec in (0, 1):
            f[prec] = sps[prec].infer(fr)
            heat[prec] = sps[prec].debug_tensor(1, (Hs, Ws))
            score[prec] = sps[prec].debug_tensor(0, (Hs, Ws))
            feats[prec].append(f[prec])
        m = heat[0] > THR * 0.5
        rel = np.abs(heat[1][m] - heat[0][m]) / heat[0][m]
        rec["heat_rel_err_max"].append(float(rel.max()))
        rec["heat_rel_err_p999"].append(float(np.quantile(rel, 0.999)))
        rec["nms_support_diff"].append(int(((score[0] > 0) != (score[1] > 0)).sum()))
        b = 4
        inner = np.zeros_like(score[0], bool); inner[b:Hs - b, b:Ws - b] = True      # remove_borders on the (H, W) grid ~ the heat grid here
        c = np.sort(score[0][(score[0] > THR) & inner])[::-1].astype(np.float64)
        rec["cand"].append(int(c.size))
        if c.size > K:
            gaps = (c[:K] - c[1:K + 1]) / c[1:K + 1]
            rec["cut_gap_rel"].append(float(gaps[K - 1]))
            rec["min_adjacent_gap_rel_in_topk"].append(float(gaps[:K - 1].min()))
            for k_ in rec["n_adjacent_gaps_below"]:
                rec["n_adjacent_gaps_below"][k_] += int((gaps[:K - 1] < float(k_)).sum())
                rec["cut_gap_below"][k_] += int(gaps[K - 1] < float(k_))
        # error of the fast mode on the candidates around and above the cut, in units of the last place of the exact score
        top = np.argsort(-np.where(inner, score[0], 0).reshape(-1), kind="stable")[:K + 200]
        s0, s1 = score[0].reshape(-1)[top], score[1].reshape(-1)[top]
        both = s1 > 0
        ulps = np.abs(s1[both].astype(np.float64) - s0[both]) / np.spacing(s0[both])
        rec.setdefault("top_err_ulps_max", []).append(float(ulps.max()))
        rec.setdefault("top_err_ulps_mean", []).append(float(ulps.mean()))
        rec.setdefault("top_scores_range", []).append([float(s0[0]), float(s0[K - 1]), float(s0[-1])])
        rec.setdefault("exact_ties_in_topk", []).append(int((np.diff(c[:K + 1]) == 0).sum()) if c.size > K else 0)
        # softmax error model |s~ - s| <= delta * s * (1 - s) + c * ulp(s): the delta and c the data needs
        hm = heat[0] > THR * 0.5
        h0, h1 = heat[0][hm].astype(np.float64), heat[1][hm].astype(np.float64)
        rec.setdefault("heat_err_over_s_1ms_max", []).append(float((np.abs(h1 - h0) / (h0 * (1 - h0) + 1e-7)).max()))
        # the product's constants (sp_api.hip kGuardSpDelta / kGuardSpUlps): what is left of the error after the delta term, in ulps
        rec.setdefault("heat_err_minus_model_ulps_max", []).append(float(((np.abs(h1 - h0) - 1.6e-4 * h0 * (1 - h0)) / np.spacing(heat[0][hm])).max()))
        # NMS near-ties: pixels that are not the maximum of their 9x9 window but within eps (relative) of it, where it matters
        from scipy.ndimage import maximum_filter
        mx = maximum_filter(heat[0], size=9, mode="constant", cval=0.0)
        rel_ = (mx - heat[0]) / mx
        for e_ in ("1e-6", "1e-5", "1e-4", "1e-3"):
            rec.setdefault("nms_near_max_pixels", {}).setdefault(e_, []).append(int(((rel_ > 0) & (rel_ < float(e_)) & (mx > THR * 0.5)).sum()))
        eq = (heat[0] == mx) & (mx > THR * 0.5)
        cnt = maximum_filter(eq.astype(np.float32), size=9, mode="constant") * 0 + __import__("scipy.ndimage", fromlist=["uniform_filter"]).uniform_filter(eq.astype(np.float64), size=9, mode="constant") * 81
        rec.setdefault("nms_exact_tie_pixels", []).append(int((eq & (cnt > 1.5)).sum()))
        allc = score[0][(score[0] > 0) & inner].astype(np.float64)
        rec["thr_margin_rel_min"].append(float(np.abs(allc - THR).min() / THR))
        k0 = {(r[1], r[2]): r[0] for r in f[0]}; k1 = {(r[1], r[2]): r[0] for r in f[1]}
        rec["kp_set_diff"].append(len(set(k0) ^ set(k1)))
        rec["kp_order_diff"].append(int(f[0].shape != f[1].shape or (f[0][:, 1:3] != f[1][:, 1:3]).any(axis=1).sum()))
        common = set(k0) & set(k1)
        rec["topk_score_rel_err_max"].append(float(max(abs(k0[k] - k1[k]) / k0[k] for k in common)))
    # pairs: exact features into both matchers (isolates the matcher's own error)
    sgs = []
    for prec in (0, 1):
        sg = F.SuperGlue(F.SuperGlueConfig(image_width=640, image_height=512), precision=prec)
        assert sg.build(sgb)
        sgs.append(sg)
    pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512))
    prec_rec = {"Z_abs_err_max_where_p_gt_0.1": [], "Z_abs_err_max_all": [], "mscore_abs_err_max": [], "idx_diff": [],
                "matches": [], "near_half": {"1e-5": 0, "3e-5": 0, "1e-4": 0, "3e-4": 0, "1e-3": 0, "1e-2": 0},
                "pairs_with_near_half": {"1e-5": 0, "3e-5": 0, "1e-4": 0, "3e-4": 0, "1e-3": 0, "1e-2": 0}}
    for t in range(1, N):
        nf0 = pm.NormalizeKeypoints(feats[0][t - 1], 640, 512); nf1 = pm.NormalizeKeypoints(feats[0][t], 640, 512)
        r = [sgs[p].infer(nf0, nf1, want_scores=True) for p in (0, 1)]
        Z0, Z1 = r[0][4], r[1][4]
        big = Z0 > np.log(0.1)
        prec_rec["Z_abs_err_max_where_p_gt_0.1"].append(float(np.abs(Z1 - Z0)[big].max()) if big.any() else 0.0)
        prec_rec["Z_abs_err_max_all"].append(float(np.abs(Z1 - Z0).max()))
        eb = np.abs(Z1 - Z0)[big]
        prec_rec.setdefault("Z_abs_err_rms_where_p_gt_0.1", []).append(float(np.sqrt((eb ** 2).mean())))
        prec_rec.setdefault("Z_abs_err_p99_where_p_gt_0.1", []).append(float(np.quantile(eb, 0.99)))
        # couplings (the input of the Sinkhorn iterations): the GNN's own error, without the two Sinkhorn forms
        import ctypes as C_
        n0_, n1_ = nf0.shape[0], nf1.shape[0]
        Cs = []
        for p_ in (0, 1):
            sgs[p_].infer(nf0, nf1)
            Cm = np.zeros((n0_ + 1, n1_ + 1), np.float32)
            assert U._lib.lib().urf_sg_debug_couplings(sgs[p_]._h, n0_, n1_, Cm.ctypes.data_as(C_.c_void_p)) == 0
            Cs.append(Cm)
        prec_rec.setdefault("C_abs_err_max", []).append(float(np.abs(Cs[1] - Cs[0]).max()))
        prec_rec.setdefault("C_abs_err_max_where_p_gt_0.1", []).append(float(np.abs(Cs[1] - Cs[0])[big].max()) if big.any() else 0.0)
        prec_rec.setdefault("C_abs_max", []).append(float(np.abs(Cs[0]).max()))
        # input-induced: the EXACT matcher on the fast mode's features (descriptors differ at the 1e-6 level) vs on the exact ones,
        # rows / columns aligned by keypoint coordinates
        g0, g1 = feats[1][t - 1], feats[1][t]
        if g0.shape == feats[0][t - 1].shape and g1.shape == feats[0][t].shape:
            def perm(a, b):      # index of each keypoint of a in b, or None when the sets differ
                m_ = {(r[1], r[2]): i for i, r in enumerate(b)}
                idx = [m_.get((r[1], r[2]), -1) for r in a]
                return None if min(idx) < 0 else np.array(idx)
            p0_, p1_ = perm(feats[0][t - 1], g0), perm(feats[0][t], g1)
            if p0_ is not None and p1_ is not None:
                Zi = sgs[0].infer(pm.NormalizeKeypoints(g0, 640, 512), pm.NormalizeKeypoints(g1, 640, 512), want_scores=True)[4]
                Zi = Zi[:-1, :-1][np.ix_(p0_, p1_)]
                di = np.abs(Zi - Z0[:-1, :-1])[big[:-1, :-1]]
                prec_rec.setdefault("Z_input_induced_err_max_where_p_gt_0.1", []).append(float(di.max()) if di.size else 0.0)
                prec_rec.setdefault("Z_input_induced_err_rms_where_p_gt_0.1", []).append(float(np.sqrt((di ** 2).mean())) if di.size else 0.0)
        prec_rec["mscore_abs_err_max"].append(float(max(np.abs(r[0][2] - r[1][2]).max(), np.abs(r[0][3] - r[1][3]).max())))
        prec_rec["idx_diff"].append(int((r[0][0] != r[1][0]).sum() + (r[0][1] != r[1][1]).sum()))
        prec_rec["matches"].append(int((r[0][0] >= 0).sum()))
        # probability of every row's / column's best inner entry (the only ones that can cross the 0.5 threshold)
        pr = np.exp(Z0[:-1, :-1].max(1).astype(np.float64)); pc = np.exp(Z0[:-1, :-1].max(0).astype(np.float64))
        for k_ in prec_rec["near_half"]:
            nn = int((np.abs(pr - 0.5) < float(k_)).sum() + (np.abs(pc - 0.5) < float(k_)).sum())
            prec_rec["near_half"][k_] += nn
            prec_rec["pairs_with_near_half"][k_] += int(nn > 0)
    prec_rec["pairs"] = N - 1
    rec["frames"] = N
    out[f"{W}x{H}"] = {"superpoint": rec, "matcher": prec_rec}
print(json.dumps(out))
