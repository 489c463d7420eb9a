"""Deterministic synthetic weights and image streams.

The reference ships neither trained weights nor the SuperGlue graph
(/root/reference/.MISSING_LARGE_BLOBS), so parity fixtures, tests and the bench
use seeded synthetic parameters.  Everything here is integer hashing plus
correctly rounded float ops, so the same seed yields bit-identical blobs on any
machine (no dependence on numpy's or torch's RNG streams).

Weight container layouts (shared, written spec in DESIGN.md):
  SP blob: for conv in [1a,1b,2a,2b,3a,3b,4a,4b,Pa,Pb,Da,Db]:
             W[kh*kw][cin][cout] f32, b[cout] f32          (1 300 865 floats)
  SG blob: kenc 5x(W[cin][cout], b) ; 18x(Wq,bq,Wk,bk,Wv,bv,Wm,bm,W1,b1,W2,b2) ;
           Wf,bf ; bin_score                                (12 003 905 floats)
           BatchNorm folded, attention channels head-major (c = h*64 + d).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

SP_CONVS = [  # name, cin, cout, k   (superpoint/SP/model.py:35-53)
    ("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 128, 256, 3), ("convPb", 256, 65, 1), ("convDa", 128, 256, 3), ("convDb", 256, 256, 1),
]
SP_BLOB_FLOATS = 1300865
SG_BLOB_FLOATS = 12003905
SG_LAYERS = 18
KENC_DIMS = [3, 32, 64, 128, 256, 256]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def u01(seed, stream, n):
    """n float64 uniforms in [0,1): splitmix64(counter) keyed by (seed, stream)."""
    with np.errstate(over="ignore"):
        key = _splitmix(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream))
        ctr = np.arange(n, dtype=np.uint64) + key
    z = _splitmix(ctr)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, stream, shape, bound):
    n = int(np.prod(shape))
    return ((u01(seed, stream, n) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------- SuperPoint
def sp_weights(seed=0, gain=3.0, calibrated=True):
    """dict name -> (W[cout,cin,k,k] f32, b[cout] f32) in torch OIHW layout.

    W ~ U(+-gain/sqrt(fan_in)), b ~ U(+-1/sqrt(fan_in)): default torch Conv2d
    init scaled by `gain` on the kernels -- gives a peaky, nearly tie-free heat
    map (SURVEY.md section 8c recipe).  With calibrated=True (seed 0 only) the
    convDb bias is replaced by data/sp_desc_bias_seed0.npy = -W*mean(relu(convDa))
    measured once on a calibration texture (tools/calibrate_sp_desc_bias.py), so
    the random descriptor head emits roughly zero-mean, discriminative
    descriptors instead of a near-constant vector."""
    out = {}
    for i, (name, cin, cout, k) in enumerate(SP_CONVS):
        fan_in = cin * k * k
        w = uniform(seed, 2 * i, (cout, cin, k, k), gain / np.sqrt(fan_in))
        b = uniform(seed, 2 * i + 1, (cout,), 1.0 / np.sqrt(fan_in))
        out[name] = (w, b)
    if calibrated and seed == 0 and gain == 3.0:
        out["convDb"] = (out["convDb"][0], np.load(os.path.join(_DATA, "sp_desc_bias_seed0.npy")).astype(np.float32))
    return out


def pack_sp(weights):
    """OIHW dict -> SP blob ([tap][cin][cout] + bias per conv)."""
    parts = []
    for name, cin, cout, k in SP_CONVS:
        w, b = weights[name]
        assert w.shape == (cout, cin, k, k)
        parts.append(np.ascontiguousarray(w.transpose(2, 3, 1, 0)).reshape(-1))  # ky,kx,cin,cout
        parts.append(b.reshape(-1))
    blob = np.concatenate(parts).astype(np.float32)
    assert blob.size == SP_BLOB_FLOATS
    return blob


# ------------------------------------------------------------------ SuperGlue
def sg_weights(seed=0, gnn_gain=0.5, final_gain=40.0, bin_score=2.3457):
    """Synthetic SuperGlue parameters in the ORIGINAL (Magic-Leap) layout:
    Conv1d weights [cout,cin]; attention channel c = d*4 + h; BatchNorm
    (gamma,beta,mean,var) on hidden MLP layers.

    The gains keep the residual stream descriptor-dominated and the final
    projection close to a scaled identity, so that true correspondences of a
    warped frame pair get high assignment scores even with random weights."""
    st = [1000]

    def nxt():
        st[0] += 1
        return st[0]

    def lin(cout, cin, g=1.0):
        return (uniform(seed, nxt(), (cout, cin), g / np.sqrt(cin)),
                uniform(seed, nxt(), (cout,), g * 0.1 / np.sqrt(cin)))

    def bn(c):
        gamma = 1.0 + uniform(seed, nxt(), (c,), 0.1)
        beta = uniform(seed, nxt(), (c,), 0.05)
        mean = uniform(seed, nxt(), (c,), 0.05)
        var = 1.0 + uniform(seed, nxt(), (c,), 0.1)
        return gamma, beta, mean, var

    w = {"kenc": [], "layers": []}
    for i in range(5):
        W, b = lin(KENC_DIMS[i + 1], KENC_DIMS[i], 1.0 if i < 4 else 0.1)
        w["kenc"].append((W, b, bn(KENC_DIMS[i + 1]) if i < 4 else None))
    for _ in range(SG_LAYERS):
        L = {}
        for nm in ("q", "k", "v"):
            L[nm] = lin(256, 256, 2.0)
        L["merge"] = lin(256, 256, 1.0)
        L["mlp0"] = lin(512, 512, 1.0) + (bn(512),)
        L["mlp1"] = lin(256, 512, gnn_gain)
        w["layers"].append(L)
    Wf, bf = lin(256, 256, 0.3)
    Wf = (Wf + final_gain * np.eye(256, dtype=np.float32)).astype(np.float32)
    w["final"] = (Wf, bf)
    w["bin_score"] = np.float32(bin_score)
    return w


def _fold_bn(W, b, bnp, eps=np.float32(1e-5)):
    if bnp is None:
        return W, b
    gamma, beta, mean, var = bnp
    s = (gamma / np.sqrt(var + eps)).astype(np.float32)
    return (W * s[:, None]).astype(np.float32), ((b - mean) * s + beta).astype(np.float32)


def head_major_perm():
    """perm[c_new] = c_orig, with c_new = h*64 + d and c_orig = d*4 + h."""
    c_new = np.arange(256)
    h, d = c_new // 64, c_new % 64
    return d * 4 + h


def pack_sg(w):
    """original-layout dict -> SG blob (BN folded, head-major, [cin][cout])."""
    perm = head_major_perm()
    parts = []

    def put(W, b):
        parts.append(np.ascontiguousarray(W.T).reshape(-1))
        parts.append(b.reshape(-1))

    for (W, b, bnp) in w["kenc"]:
        put(*_fold_bn(W, b, bnp))
    for L in w["layers"]:
        for nm in ("q", "k", "v"):
            W, b = L[nm]
            put(W[perm, :], b[perm])          # output channels reordered
        Wm, bm = L["merge"]
        put(Wm[:, perm], bm)                  # input channels reordered
        W0, b0, bn0 = L["mlp0"]
        put(*_fold_bn(W0, b0, bn0))
        put(*L["mlp1"])
    put(*w["final"])
    parts.append(np.array([w["bin_score"]], dtype=np.float32))
    blob = np.concatenate(parts).astype(np.float32)
    assert blob.size == SG_BLOB_FLOATS
    return blob


# -------------------------------------------------------------------- streams
def _box_blur(a, r):
    k = 2 * r + 1
    p = np.pad(a, ((r, r), (r, r)), mode="reflect")
    c = np.cumsum(np.pad(p, ((1, 0), (0, 0))), axis=0)
    p = (c[k:, :] - c[:-k, :]) / k
    c = np.cumsum(np.pad(p, ((0, 0), (1, 0))), axis=1)
    return (c[:, k:] - c[:, :-k]) / k


def base_frame(seed, H, W):
    """u8 texture: multi-scale blurred noise, full 0..255 range."""
    n = u01(seed, 7, H * W).reshape(H, W)
    img = 0.5 * _box_blur(n, 1) + 0.3 * _box_blur(n, 3) + 0.2 * _box_blur(u01(seed, 8, H * W).reshape(H, W), 8)
    img = (img - img.min()) / (img.max() - img.min())
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def homography(seed, t, H, W):
    """small per-step projective motion (rotation + shift + perspective)."""
    r = u01(seed, 100 + t, 8) * 2.0 - 1.0
    ang = 0.01 * r[0]
    c, s = np.cos(ang), np.sin(ang)
    Hm = np.array([[c * (1 + 0.01 * r[1]), -s, 4.0 * r[2]],
                   [s, c * (1 + 0.01 * r[3]), 3.0 * r[4]],
                   [2e-6 * r[5], 2e-6 * r[6], 1.0]])
    T = np.array([[1, 0, W / 2], [0, 1, H / 2], [0, 0, 1.0]])
    return T @ Hm @ np.linalg.inv(T)


def warp(img, Hm, seed=0, t=0, noise=2.0):
    """bilinear inverse warp of a u8 image by homography Hm (maps src->dst) plus
    uniform noise of +-noise grey levels."""
    H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    Hi = np.linalg.inv(Hm)
    d = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / d
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / d
    sx = np.clip(sx, 0, W - 1.001)
    sy = np.clip(sy, 0, H - 1.001)
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    f = img.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy)
           + f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
    out = out + (u01(seed, 500 + t, H * W).reshape(H, W) * 2 - 1) * noise
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def stream(seed, n_frames, H, W):
    """frames[0] = base texture; frames[t] = warp of frames[0] by the composed
    motion up to t (so consecutive frames share true correspondences)."""
    f0 = base_frame(seed, H, W)
    frames, Hs = [f0], [np.eye(3)]
    acc = np.eye(3)
    for t in range(1, n_frames):
        acc = homography(seed, t, H, W) @ acc
        frames.append(warp(f0, acc, seed, t))
        Hs.append(acc.copy())
    return frames, Hs


def shift_stream(seed, n_frames, H, W, step=(8, 8), noise=2.0):
    """frames cropped from one large texture at offsets t*step (multiples of the
    8-px SuperPoint cell, so even a random-weight detector is repeatable) plus
    +-noise grey levels.  True correspondence: p_{t+1} = p_t - step."""
    dx, dy = step
    big = base_frame(seed, H + dy * n_frames, W + dx * n_frames)
    frames = []
    for t in range(n_frames):
        f = big[dy * t:dy * t + H, dx * t:dx * t + W].astype(np.float64)
        f = f + (u01(seed, 500 + t, H * W).reshape(H, W) * 2 - 1) * noise
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    return frames
