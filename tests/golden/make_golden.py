"""Generates the golden parity fixtures in tests/golden/ (run in the build
container only: it imports the reference's own PyTorch SuperPoint graph from
/root/reference/superpoint/SP/model.py and, for SuperGlue, the public
architecture as implemented by transformers.models.superglue).

    python tests/golden/make_golden.py

Fixtures are DATA (inputs + expected outputs), never reference source:
  sp_dense_96x128.npz   image, post-NMS scores, dense descriptors (torch fp32)
  sp_sparse_240x320.npz image, keypoints/scores/descriptors after the
                        reference post-processing restated with torch ops
  sp_sparse_376x1241.npz same at the KITTI size (valid width 1240)
  sp_sparse_480x640.npz same at the size of BASELINE.json's headline configuration
  sp_bench_stream_480x640.npz, sp_bench_stream_376x1241.npz
                        frames 0..8 of the stream bench.py times (synth.shift_stream(100, 40, H, W)) through the
                        reference graph: per frame the top-1000 keypoints (x, y u16; score f32, reference order), the score
                        of the best candidate that did NOT make the cut and the candidate count.  The frames themselves are
                        regenerated from the seed.
  sg_n96.npz            two feature sets and the (n0+1)x(n1+1) log-assignment
  sg_n320.npz           same at n = 320 (features regenerated from the stored seed: conftest.sg_golden_features)
  sg_n1000.npz          n = 1000 (the bench size): every 8th row of the log-assignment in f32 (Zrows), the full
                        dustbin row/column, and the decode of the FULL tensor (indices0/1, mscores0/1 as
                        src/super_glue.cpp:303-430 computes them) -- 0.6 MB instead of 4 MB
The seeded synthetic weights are regenerated bit-exactly by synth.py.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference/superpoint/SP")
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "ur-mvo_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)
import model  # noqa: E402  (reference graph)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import sg_golden_features  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)
T = torch.from_numpy


def sp_model():
    w = synth.sp_weights(0)
    m = model.SuperPoint().eval()
    with torch.no_grad():
        for name, (W, b) in w.items():
            getattr(m, name).weight.copy_(T(W))
            getattr(m, name).bias.copy_(T(b))
    return m


def sp_run(m, img):
    # SuperPoint::process_input src/super_point.cpp:169-174
    x = T((img.astype(np.float64) / 255.0).astype(np.float32))[None, None]
    with torch.no_grad():
        s, d = m(x)
    return s[0].numpy(), d[0].permute(1, 2, 0).contiguous().numpy()


def sp_post(scores, desc, k=1000, thr=0.0005, border=4):
    """src/super_point.cpp:196-251,253-336 restated with numpy/torch ops."""
    Hs, Ws = scores.shape
    ys, xs = np.nonzero(scores.astype(np.float64) > thr)
    keep = (ys >= border) & (ys < Hs - border) & (xs >= border) & (xs < Ws - border)
    ys, xs = ys[keep], xs[keep]
    sc = scores[ys, xs]
    if k != -1 and k < len(sc):
        order = np.lexsort((ys * Ws + xs, -sc.astype(np.float64)))[:k]
        ys, xs, sc = ys[order], xs[order], sc[order]
    Hc, Wc = desc.shape[:2]
    kx = (xs - 4 + 0.5) / (Wc * 8 - 4 - 0.5) * 2 - 1
    ky = (ys - 4 + 0.5) / (Hc * 8 - 4 - 0.5) * 2 - 1
    grid = T(np.stack([kx, ky], -1)).double()[None, None]
    dm = T(desc).double().permute(2, 0, 1)[None]
    sm = torch.nn.functional.grid_sample(dm, grid, mode="bilinear", align_corners=True)[0, :, 0].T
    sm = torch.nn.functional.normalize(sm, p=2, dim=1).numpy()
    return xs.astype(np.int32), ys.astype(np.int32), sc.astype(np.float32), sm.astype(np.float32)


def main():
    m = sp_model()
    if not os.path.exists(os.path.join(OUT, "sp_dense_96x128.npz")) or "--force" in sys.argv:
        img = synth.shift_stream(3, 1, 96, 128)[0]
        s, d = sp_run(m, img)
        np.savez_compressed(os.path.join(OUT, "sp_dense_96x128.npz"), image=img, scores=s, desc=d.astype(np.float32))
    for (H, W, k, seed) in [(240, 320, 300, 4), (376, 1241, 1000, 5), (480, 640, 1000, 6)]:
        if os.path.exists(os.path.join(OUT, f"sp_sparse_{H}x{W}.npz")) and "--force" not in sys.argv:
            continue
        img = synth.shift_stream(seed, 1, H, W)[0]
        s, d = sp_run(m, img)
        xs, ys, sc, ds = sp_post(s, d, k=k)
        np.savez_compressed(os.path.join(OUT, f"sp_sparse_{H}x{W}.npz"), image=img, x=xs, y=ys, score=sc,
                            desc=ds.astype(np.float16), k=np.int32(k),
                            n_candidates=np.int32(int((s.astype(np.float64) > 0.0005).sum())))
        print(H, W, "K", len(xs), "cands", int((s > 0.0005).sum()))

    for (H, W) in [(480, 640), (376, 1241)]:
        name = os.path.join(OUT, f"sp_bench_stream_{H}x{W}.npz")
        if os.path.exists(name) and "--force" not in sys.argv:
            continue
        frames = synth.shift_stream(100, 40, H, W)[:9]          # bench.py: NB * batch * world = 40 frames at one GPU
        xs_, ys_, sc_, nxt, ncand = [], [], [], [], []
        for img in frames:
            s, d = sp_run(m, img)
            xs, ys, sc, _ = sp_post(s, d, k=1000)
            x2, y2, sc2, _ = sp_post(s, d[:, :, :], k=1001)
            assert len(xs) == 1000 and np.array_equal(xs, x2[:1000]) and np.array_equal(ys, y2[:1000])
            xs_.append(xs.astype(np.uint16)); ys_.append(ys.astype(np.uint16)); sc_.append(sc)
            nxt.append(sc2[1000]); ncand.append(len(sp_post(s, d, k=-1)[0]))
        np.savez_compressed(name, x=np.stack(xs_), y=np.stack(ys_), score=np.stack(sc_), first_cut_score=np.array(nxt, np.float32),
                            n_candidates=np.array(ncand, np.int32), seed=np.int32(100), stream_frames=np.int32(40))
        print(name, "cands", ncand, "cut margins (relative)", [(a[-1] - b) / b for a, b in zip(sc_, nxt)])

    sg_cases = [(96, 40, 7, "sg_n96.npz"), (320, 150, 8, "sg_n320.npz"), (1000, 600, 9, "sg_n1000.npz")]
    sg_cases = [c for c in sg_cases if "--force" in sys.argv or not os.path.exists(os.path.join(OUT, c[3]))]
    if not sg_cases:
        return
    # ---- SuperGlue: public architecture (transformers) with the synthetic weights
    from transformers.models.superglue import modeling_superglue as MS
    from transformers.models.superglue.configuration_superglue import SuperGlueConfig
    cfg = SuperGlueConfig()
    cfg._attn_implementation = "eager"
    w = synth.sg_weights(0)
    perm = synth.head_major_perm()
    kenc = MS.SuperGlueKeypointEncoder(cfg).eval()
    gnn = MS.SuperGlueAttentionalGNN(cfg).eval()
    fin = MS.SuperGlueFinalProjection(cfg).eval()

    def setbn(bn, p):
        g, b, mu, v = p
        bn.weight.copy_(T(g)); bn.bias.copy_(T(b)); bn.running_mean.copy_(T(mu)); bn.running_var.copy_(T(v))

    with torch.no_grad():
        for i, (W, b, bnp) in enumerate(w["kenc"]):
            L = kenc.encoder[i]
            if bnp is not None:
                L.linear.weight.copy_(T(W)); L.linear.bias.copy_(T(b)); setbn(L.batch_norm, bnp)
            else:
                L.weight.copy_(T(W)); L.bias.copy_(T(b))
        for li, L in enumerate(w["layers"]):
            g = gnn.layers[li]
            for nm, mod in (("q", g.attention.self.query), ("k", g.attention.self.key), ("v", g.attention.self.value)):
                W, b = L[nm]
                mod.weight.copy_(T(W[perm, :])); mod.bias.copy_(T(b[perm]))
            Wm, bm = L["merge"]
            g.attention.output.dense.weight.copy_(T(Wm[:, perm])); g.attention.output.dense.bias.copy_(T(bm))
            W0, b0, bn0 = L["mlp0"]
            g.mlp[0].linear.weight.copy_(T(W0)); g.mlp[0].linear.bias.copy_(T(b0)); setbn(g.mlp[0].batch_norm, bn0)
            W1, b1 = L["mlp1"]
            g.mlp[1].weight.copy_(T(W1)); g.mlp[1].bias.copy_(T(b1))
        Wf, bf = w["final"]
        fin.final_proj.weight.copy_(T(Wf)); fin.final_proj.bias.copy_(T(bf))
    for (n, planted, seed, name) in sg_cases:

        f0, f1 = sg_golden_features(n, planted, seed)

        def norm(f):  # src/point_matching.cc:63-76
            g = f.copy()
            g[:, 1] = (f[:, 1] - 640 // 2) / (640 * 0.7)
            g[:, 2] = (f[:, 2] - 512 // 2) / (640 * 0.7)
            return g

        nf0, nf1 = norm(f0), norm(f1)
        with torch.no_grad():
            kp = T(np.stack([nf0[:, 1:3], nf1[:, 1:3]]).astype(np.float32))
            sc = T(np.stack([nf0[:, 0], nf1[:, 0]]).astype(np.float32))
            ds = T(np.stack([nf0[:, 3:], nf1[:, 3:]]).astype(np.float32))
            enc, _ = kenc(kp, sc)
            x, _, _ = gnn(ds + enc, mask=None)
            pr = fin(x)
            S = pr[0:1] @ pr[1:2].transpose(1, 2) / 16.0
            Z = MS.log_optimal_transport(S, torch.tensor(float(w["bin_score"])), 100)[0].numpy()
        if n >= 1000:
            # decode of the full tensor, restated with numpy (src/super_glue.cpp:303-430): argmax over the inner block,
            # first maximum wins, mutual check, exp, threshold 0.5
            inner = Z[:-1, :-1]
            a0, a1 = inner.argmax(1), inner.argmax(0)
            mut0 = a1[a0] == np.arange(n)
            ms0 = np.where(mut0, np.exp(inner[np.arange(n), a0].astype(np.float64)), 0.0)
            i0 = np.where(mut0 & (ms0 > 0.5), a0, -1).astype(np.int32)
            mut1 = a0[a1] == np.arange(n)
            ms1 = np.where(mut1, ms0[a1], 0.0)
            i1 = np.where(mut1 & (i0[a1] >= 0), a1, -1).astype(np.int32)
            np.savez_compressed(os.path.join(OUT, name), n=np.int32(n), seed=np.int32(seed), planted=np.int32(planted),
                                Zrows=Z[::8].astype(np.float32), Zbin_row=Z[-1].astype(np.float32),
                                Zbin_col=Z[:, -1].astype(np.float32), indices0=i0, indices1=i1,
                                mscores0=ms0.astype(np.float32), mscores1=ms1.astype(np.float32),
                                rowmax=inner.max(1).astype(np.float32), colmax=inner.max(0).astype(np.float32))
            print(name, "matches", int((i0 >= 0).sum()), "Z range", Z.min(), Z.max())
            continue
        np.savez_compressed(os.path.join(OUT, name), **({} if n > 96 else dict(f0=f0, f1=f1)),   # larger: regenerated from the seed
                            n=np.int32(n), seed=np.int32(seed), Z=Z.astype(np.float32),
                            final0=pr[0].numpy().astype(np.float16 if n > 96 else np.float32),
                            final1=pr[1].numpy().astype(np.float16 if n > 96 else np.float32), planted=np.int32(planted))
        print(name, "Z range", Z.min(), Z.max())



if __name__ == "__main__":
    main()
