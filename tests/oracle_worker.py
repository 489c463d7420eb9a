"""Worker process of tests/conftest.py::run_oracle_jobs: runs a shard of CPU-oracle jobs read from an .npz file and writes
the results to another (test infrastructure; the C oracle uses <= 32 OpenMP threads, a GPU box has many more cores).

    python tests/oracle_worker.py shard_in.npz shard_out.npz

shard_in: kind ("sp" | "pm"), n jobs; sp: img_i (u8 [H, W]), max_kp_i; pm: f0_i, f1_i ([K, 259] f64), ransac_i ("ref" | "sigma1")
shard_out: sp: feat_i; pm: m_i ([n, 3] f64: queryIdx, trainIdx, distance as the f32 it is)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import SG_CFG, load_pkg  # noqa: E402


def main():
    src, dst = sys.argv[1:3]
    d = np.load(src, allow_pickle=False)
    U = load_pkg()
    from oracle import oracle as O
    O.build()
    kind, n = str(d["kind"]), int(d["n"])
    out = {}
    if kind == "sp":
        blob = U.synth.pack_sp(U.synth.sp_weights(0))
        for i in range(n):
            out[f"feat_{i}"] = O.sp_infer(blob, O.SPConfig(int(d[f"max_kp_{i}"]), 0.0005, 4), d[f"img_{i}"])
    else:
        blob = U.synth.pack_sg(U.synth.sg_weights(0))
        for i in range(n):
            rc = O.ref_ransac() if str(d[f"ransac_{i}"]) == "ref" else O.RansacConfig(200, 1.0, 0)
            m = O.match_points(blob, O.SGConfig(*SG_CFG), rc, d[f"f0_{i}"], d[f"f1_{i}"], True)
            out[f"m_{i}"] = np.array(m, np.float64).reshape(-1, 3)
    np.savez(dst, **out)


if __name__ == "__main__":
    main()
