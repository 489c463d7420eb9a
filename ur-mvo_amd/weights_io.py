"""Weight import: the reference's weight files -> URFW containers (`engine_file`).

The reference builds its engines from `superpoint_v1.onnx` and
`superglue_indoor_sim_int32.onnx` (configs/configs_aqua.yaml:15,31;
src/super_point.cpp:21-102, src/super_glue.cpp:21-147) and caches a TensorRT plan
in `engine_file`.  This back-end's `engine_file` is a URFW container
("URFW", u32 kind, u64 count, f32 payload; include/urf.h) holding the packed
weights of DESIGN.md section 3.  This module writes such containers from

  * a SuperPoint state dict (`superpoint/SP/model.py:38-53` names: conv1a ... convDb),
  * a SuperGlue state dict in the public Magic-Leap layout (kenc.encoder.N,
    gnn.layers.L.attn.proj.{0,1,2} / merge, gnn.layers.L.mlp.{0,1,3}, final_proj,
    bin_score) -- BatchNorm is folded, heads are re-ordered head-major,
  * the ONNX files themselves, read with a ~100-line protobuf wire-format reader
    (no `onnx` package in the image): initialisers by name when the exporter kept
    the parameter names, otherwise the Conv nodes in graph order.

Neither blob ships with the reference (`.MISSING_LARGE_BLOBS`), so the ONNX
path is exercised on files this module's own writer produces (tests) and is
UNVERIFIED against the real blobs; the state-dict path is pinned by the
SuperPoint module of the reference (tests/golden/make_golden.py uses the same
packing).

    python tools/import_weights.py --superpoint superpoint_v1.pth --out sp.urfw
"""
import struct

import numpy as np

from . import synth

KIND_SP, KIND_SG = 1, 2


# ----------------------------------------------------------------- containers
def save_container(path, kind, blob):
    blob = np.ascontiguousarray(blob, np.float32).reshape(-1)
    want = synth.SP_BLOB_FLOATS if kind == KIND_SP else synth.SG_BLOB_FLOATS
    if blob.size != want:
        raise ValueError(f"kind {kind} container holds {want} floats, got {blob.size}")
    with open(path, "wb") as f:
        f.write(b"URFW" + struct.pack("<IQ", kind, blob.size))
        f.write(blob.tobytes())


def load_container(path):
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) != 16 or head[:4] != b"URFW":
            raise ValueError(f"{path}: not a URFW container")
        kind, n = struct.unpack("<IQ", head[4:])
        blob = np.frombuffer(f.read(4 * n), np.float32)
    if blob.size != n:
        raise ValueError(f"{path}: truncated")
    return kind, blob.copy()


def _np(t):
    return np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, np.float32)


# ----------------------------------------------------------------- SuperPoint
def superpoint_from_state_dict(sd):
    """state dict of superpoint/SP/model.py's SuperPoint -> SP blob"""
    w = {}
    for name, cin, cout, k in synth.SP_CONVS:
        W, b = _np(sd[name + ".weight"]), _np(sd[name + ".bias"])
        if W.shape != (cout, cin, k, k) or b.shape != (cout,):
            raise ValueError(f"{name}: expected {(cout, cin, k, k)}, got {W.shape}")
        w[name] = (W, b)
    return synth.pack_sp(w)


# ------------------------------------------------------------------ SuperGlue
_KENC_CONV = (0, 3, 6, 9, 12)      # Conv1d positions in kenc.encoder (Conv, BN, ReLU triples; last Conv bare)
_KENC_BN = (1, 4, 7, 10)


def _conv1d(sd, key):
    W = _np(sd[key + ".weight"])
    if W.ndim == 3:
        W = W[:, :, 0]
    return W, _np(sd[key + ".bias"])


def _bn(sd, key):
    return (_np(sd[key + ".weight"]), _np(sd[key + ".bias"]), _np(sd[key + ".running_mean"]),
            _np(sd[key + ".running_var"]))


def superglue_from_state_dict(sd):
    """public SuperGlue state dict (Magic-Leap layout) -> SG blob (BN folded, head-major)"""
    w = {"kenc": [], "layers": []}
    for i, pos in enumerate(_KENC_CONV):
        W, b = _conv1d(sd, f"kenc.encoder.{pos}")
        w["kenc"].append((W, b, _bn(sd, f"kenc.encoder.{_KENC_BN[i]}") if i < 4 else None))
    for layer in range(synth.SG_LAYERS):
        p = f"gnn.layers.{layer}"
        L = {nm: _conv1d(sd, f"{p}.attn.proj.{j}") for j, nm in enumerate(("q", "k", "v"))}
        L["merge"] = _conv1d(sd, f"{p}.attn.merge")
        L["mlp0"] = _conv1d(sd, f"{p}.mlp.0") + (_bn(sd, f"{p}.mlp.1"),)
        L["mlp1"] = _conv1d(sd, f"{p}.mlp.3")
        w["layers"].append(L)
    w["final"] = _conv1d(sd, "final_proj")
    w["bin_score"] = np.float32(_np(sd["bin_score"]).reshape(-1)[0])
    return synth.pack_sg(w)


def superglue_to_state_dict(w):
    """inverse naming of superglue_from_state_dict for a synth.sg_weights() structure (tests, export)"""
    sd = {}

    def put_bn(key, bnp):
        sd[key + ".weight"], sd[key + ".bias"], sd[key + ".running_mean"], sd[key + ".running_var"] = bnp

    for i, pos in enumerate(_KENC_CONV):
        W, b, bnp = w["kenc"][i]
        sd[f"kenc.encoder.{pos}.weight"], sd[f"kenc.encoder.{pos}.bias"] = W[:, :, None], b
        if bnp is not None:
            put_bn(f"kenc.encoder.{_KENC_BN[i]}", bnp)
    for layer, L in enumerate(w["layers"]):
        p = f"gnn.layers.{layer}"
        for j, nm in enumerate(("q", "k", "v")):
            sd[f"{p}.attn.proj.{j}.weight"], sd[f"{p}.attn.proj.{j}.bias"] = L[nm][0][:, :, None], L[nm][1]
        sd[f"{p}.attn.merge.weight"], sd[f"{p}.attn.merge.bias"] = L["merge"][0][:, :, None], L["merge"][1]
        sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"] = L["mlp0"][0][:, :, None], L["mlp0"][1]
        put_bn(f"{p}.mlp.1", L["mlp0"][2])
        sd[f"{p}.mlp.3.weight"], sd[f"{p}.mlp.3.bias"] = L["mlp1"][0][:, :, None], L["mlp1"][1]
    sd["final_proj.weight"], sd["final_proj.bias"] = w["final"][0][:, :, None], w["final"][1]
    sd["bin_score"] = np.array(w["bin_score"], np.float32)
    return sd


# ------------------------------------------------- ONNX (protobuf wire format)
# ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5; NodeProto.input = 1, .output = 2,
# .op_type = 4; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9.
def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, i
        s += 7


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _tensor(buf):
    dims, dtype, name, raw, floats = [], 0, "", None, []
    for f, wt, v in _fields(buf):
        if f == 1:
            if wt == 0:
                dims.append(v)
            else:                      # packed repeated int64
                j = 0
                while j < len(v):
                    d, j = _varint(v, j)
                    dims.append(d)
        elif f == 2:
            dtype = v
        elif f == 4:
            floats.append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(v, "<f4", count=1))
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    if dtype != 1:                     # FLOAT only (int64 shape constants etc. are not weights)
        return name, None
    data = np.frombuffer(raw, "<f4") if raw is not None else (np.concatenate(floats) if floats else np.zeros(0, "<f4"))
    return name, data.astype(np.float32).reshape(dims if dims else ())


def read_onnx(path):
    """-> (initialisers: name -> f32 array, nodes: [(op_type, [inputs], [outputs])] in graph order)"""
    buf = memoryview(open(path, "rb").read())
    graph = None
    for f, wt, v in _fields(buf):
        if f == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no graph in the ONNX model")
    inits, nodes = {}, []
    for f, wt, v in _fields(graph):
        if f == 5 and wt == 2:
            name, arr = _tensor(v)
            if arr is not None:
                inits[name] = arr
        elif f == 1 and wt == 2:
            ins, outs, op = [], [], ""
            for nf, nwt, nv in _fields(v):
                if nf == 1:
                    ins.append(bytes(nv).decode())
                elif nf == 2:
                    outs.append(bytes(nv).decode())
                elif nf == 4:
                    op = bytes(nv).decode()
            nodes.append((op, ins, outs))
    return inits, nodes


def _convs_in_order(inits, nodes):
    """(W, b) of every Conv node whose weight is an initialiser, first use only, graph order"""
    seen, out = set(), []
    for op, ins, _ in nodes:
        if op == "Conv" and len(ins) >= 2 and ins[1] in inits and ins[1] not in seen:
            seen.add(ins[1])
            W = inits[ins[1]]
            b = inits[ins[2]] if len(ins) > 2 and ins[2] in inits else np.zeros(W.shape[0], np.float32)
            out.append((W, b))
    return out


def superpoint_from_onnx(path):
    inits, nodes = read_onnx(path)
    if all(n + ".weight" in inits for n, *_ in synth.SP_CONVS):
        return superpoint_from_state_dict(inits)
    convs = _convs_in_order(inits, nodes)
    # export order of model.py:58-86: conv1a..conv4b, convPa, convPb, convDa, convDb
    if len(convs) != len(synth.SP_CONVS):
        raise ValueError(f"{path}: {len(convs)} Conv nodes, SuperPoint has {len(synth.SP_CONVS)}")
    return superpoint_from_state_dict({**{n + ".weight": convs[i][0] for i, (n, *_r) in enumerate(synth.SP_CONVS)},
                                       **{n + ".bias": convs[i][1] for i, (n, *_r) in enumerate(synth.SP_CONVS)}})


def superglue_from_onnx(path):
    """By name when the exporter kept them.  Otherwise (constant folding renames the Conv weights and
    folds BatchNorm into them) the distinct Conv weights in graph order: 5 keypoint-encoder layers,
    then per GNN layer q, k, v, merge, mlp.0, mlp.3, then final_proj; bin_score is the only scalar."""
    inits, nodes = read_onnx(path)
    if "final_proj.weight" in inits and "gnn.layers.0.attn.proj.0.weight" in inits and "kenc.encoder.1.running_mean" in inits:
        return superglue_from_state_dict(inits)
    # the positional path assumes the exporter folded BatchNorm into the Conv weights: a graph that still carries
    # BatchNormalization nodes has the same Conv count and would silently lose them
    bn = [op for op, _ins, _outs in nodes if op == "BatchNormalization"]
    if bn:
        raise ValueError(f"{path}: {len(bn)} BatchNormalization nodes with renamed initialisers: export with constant "
                         f"folding (BatchNorm folded into the Conv weights) or keep the parameter names")
    convs = [(W[:, :, 0] if W.ndim == 3 else W, b) for W, b in _convs_in_order(inits, nodes)]
    want = 5 + 6 * synth.SG_LAYERS + 1
    if len(convs) != want:
        raise ValueError(f"{path}: {len(convs)} distinct Conv weights, SuperGlue has {want}")
    scal = [v for k, v in inits.items() if v.size == 1 and "bin_score" in k] or [v for v in inits.values() if v.size == 1]
    if len(scal) != 1:
        raise ValueError(f"{path}: cannot identify bin_score ({len(scal)} scalar initialisers)")
    it = iter(convs)
    w = {"kenc": [next(it) + (None,) for _ in range(5)], "layers": []}
    for _ in range(synth.SG_LAYERS):
        L = {nm: next(it) for nm in ("q", "k", "v")}
        L["merge"] = next(it)
        L["mlp0"] = next(it) + (None,)
        L["mlp1"] = next(it)
        w["layers"].append(L)
    w["final"] = next(it)
    w["bin_score"] = np.float32(scal[0].reshape(-1)[0])
    return synth.pack_sg(w)


# minimal writer, the inverse of read_onnx: used by the tests to make files the reader has never seen
def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _vi(field << 3 | 2) + _vi(len(payload)) + payload


def write_onnx(path, inits, nodes, raw=True):
    g = b""
    for op, ins, outs in nodes:
        n = b"".join(_ld(1, s.encode()) for s in ins) + b"".join(_ld(2, s.encode()) for s in outs) + _ld(4, op.encode())
        g += _ld(1, n)
    for name, arr in inits.items():
        arr = np.asarray(arr, np.float32)
        t = b"".join(_vi(1 << 3) + _vi(d) for d in arr.shape) + _vi(2 << 3) + _vi(1)
        t += _ld(9, arr.astype("<f4").tobytes()) if raw else _ld(4, arr.astype("<f4").tobytes())
        t += _ld(8, name.encode())
        g += _ld(5, t)
    with open(path, "wb") as f:
        f.write(_vi(1 << 3) + _vi(7) + _ld(7, g))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--superpoint", help=".pth state dict or .onnx of SuperPoint")
    ap.add_argument("--superglue", help=".pth state dict or .onnx of SuperGlue")
    ap.add_argument("--out", required=True, help="URFW container to write")
    a = ap.parse_args(argv)
    if bool(a.superpoint) == bool(a.superglue):
        ap.error("give exactly one of --superpoint / --superglue")
    src = a.superpoint or a.superglue
    if src.endswith(".onnx"):
        blob = superpoint_from_onnx(src) if a.superpoint else superglue_from_onnx(src)
    else:
        import torch
        sd = torch.load(src, map_location="cpu", weights_only=True)    # tensors only: a .pth is a pickle
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        blob = superpoint_from_state_dict(sd) if a.superpoint else superglue_from_state_dict(sd)
    save_container(a.out, KIND_SP if a.superpoint else KIND_SG, blob)
    print(f"wrote {a.out}: {blob.size} floats")


if __name__ == "__main__":
    main()
