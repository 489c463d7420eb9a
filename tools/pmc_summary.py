"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o pmc -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch/pmc_results.db gpurun_out/pmc_write/pmc_results.db \
        profiles/r01_pmc_hbm.json "<the bench command>"

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3): both counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced reads at
64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores.  The counters sit on
the memory side of L2, i.e. Infinity-Cache hits are included: this is L2-miss traffic, an upper
bound on HBM bytes.
"""
import collections
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """must equal bench.kernel_source_sha(): bench.py refuses a summary taken on other kernel sources"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ur-mvo_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ur-mvo_amd", "csrc", "*.h"))):
        if os.path.basename(f) != "probes.hip":          # diagnostics only: no kernel of the path lives there
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def per_kernel(db):
    cur = sqlite3.connect(db).cursor()
    agg = collections.defaultdict(list)
    for name, val in cur.execute("select kernel_name, value from counters_collection"):
        agg[name].append(val)
    return agg


def family(name):
    if "h2conv_kernel<true, true" in name:
        return "h2conv_kernel<pool,fuse1a>"      # conv1a+conv1b fused: its own line (bench.py's conv1 kernel)
    if "conv_mfma_kernel<9, true, true" in name:
        return "conv_mfma_kernel<9,pool,fuse1a>"  # the same in the exact mode
    if "conv_mfma_kernel<1," in name or "gemm128_kernel" in name:
        return "linear_exact"                    # the exact mode's linear layers (SuperGlue + the two 1x1 heads of SuperPoint)
    for key in ("ransac_", "h2gemm", "h2mlp_kernel", "attn_h2_kernel", "h2conv_kernel", "sinkhorn_half_kernel", "sinkhorn_wide_kernel", "sinkhorn_resident_kernel", "sinkhorn_regs_kernel", "conv_mfma_kernel",
                "gemm128_kernel", "attn_kernel", "score_kernel", "nms_pass_kernel", "topk_kernel", "sample_kernel",
                "desc_norm_kernel", "softmax_d2s_kernel", "argmax_kernel", "decode_kernel", "split_kernel", "guard_compact_kernel", "nms_tie_kernel",
                "sg_prep_slots_kernel"):
        if key in name:
            return key
    return None


def main():
    fdb, wdb, out, cmd = sys.argv[1:5]
    f, w = per_kernel(fdb), per_kernel(wdb)
    fam = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
    for name, vals in f.items():
        k = family(name)
        if k:
            fam[k]["launches"] += len(vals)
            fam[k]["fetch_kib"] += float(sum(vals))
    for name, vals in w.items():
        k = family(name)
        if k:
            fam[k]["write_kib"] += float(sum(vals))
    res = {}
    for k, v in sorted(fam.items(), key=lambda kv: -(2 * kv[1]["fetch_kib"] + kv[1]["write_kib"])):
        n = max(v["launches"], 1)
        res[k] = {"launches": v["launches"], "bytes_total": int((2 * v["fetch_kib"] + v["write_kib"]) * 1024),
                  "FETCH_SIZE_KiB_per_launch_raw": round(v["fetch_kib"] / n, 1),
                  "WRITE_SIZE_KiB_per_launch": round(v["write_kib"] / n, 1),
                  "bytes_per_launch": int((2 * v["fetch_kib"] + v["write_kib"]) / n * 1024)}
    # matcher calls: one sg_prep_slots_kernel launch per urf_match_device_async (the redo engine of a strict handle never
    # launches it; the resident Sinkhorn would: one launch per call in its register form, two in the LDS form of the guarded modes)
    pm_calls = max(fam.get("sg_prep_slots_kernel", fam.get("decode_kernel", {"launches": 0}))["launches"], 1)
    # one topk_kernel launch per SuperPoint call -- two in the guarded fast mode (the gated redo pass), which also launches
    # one guard_compact_kernel per call
    sp_calls = max(fam.get("guard_compact_kernel", fam.get("topk_kernel", {"launches": 0}))["launches"], 1)
    json.dump({"command": cmd, "source_sha": kernel_source_sha(), "matcher_calls": pm_calls, "superpoint_calls": sp_calls, "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: wide reads tallied at half)",
               "scope": "L2-miss (fabric) traffic incl. Infinity-Cache hits; separate --pmc passes",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:24s} {v['launches']:6d} launches  {v['bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == "__main__":
    main()
