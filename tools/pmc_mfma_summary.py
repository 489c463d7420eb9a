"""Summarise one rocprofv3 PMC pass of bench.py -- the matrix pipe's busy cycles per kernel -- into profiles/.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 --kernel-trace \
        -d gpurun_out/pmc_mfma -o pmc -- python3 bench.py ...
    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma/pmc_results.db profiles/r05_pmc_mfma.json "<the bench command>"

Formula (rocprofv3 -L on gfx950, `MfmaUtil`): busy fraction = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE x SIMD_NUM), SIMD_NUM =
1024, GRBM_GUI_ACTIVE taken per XCD (rocprofv3 reports the sum over the 8 XCDs of a dispatch: divided by 8 here;
MI355X_MICROARCH.md, "DVFS give-back").  Under --pmc the dispatches are serialised, so this is each kernel ALONE on the chip --
the same situation as bench.py's roofline pass.  SQ_INSTS_VALU_MFMA_* are the issued instructions per dtype: with the known
cycles per instruction (v_mfma_f32_16x16x32_f16: 16, v_mfma_f32_16x16x4_f32: 32) they cross-check the busy cycles.
"""
import collections
import json
import sqlite3
import sys

from pmc_summary import family, kernel_source_sha

SIMD_NUM, XCDS = 1024, 8


def main():
    db, out, cmd = sys.argv[1:4]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    cname = "counter_name" if "counter_name" in cols else "name"
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    has_disp = "dispatch_id" in cols
    q = f"select kernel_name, {cname}, value" + (", dispatch_id" if has_disp else "") + " from counters_collection"
    for row in cur.execute(q):
        k = family(row[0])
        if not k:
            continue
        fam[k][row[1]] += float(row[2])
        if has_disp:
            disp[k].add(row[3])
        elif row[1] == "GRBM_GUI_ACTIVE":
            fam[k]["_rows"] += 1
    res = {}
    for k, v in fam.items():
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f16, f32 = v.get("SQ_INSTS_VALU_MFMA_F16", 0.0), v.get("SQ_INSTS_VALU_MFMA_F32", 0.0)
        n = len(disp[k]) if has_disp else int(v.get("_rows", 0))
        if gui <= 0 or busy <= 0:
            continue
        res[k] = {"launches": n, "mfma_busy_cycles": int(busy), "gui_active_cycles_per_xcd": int(gui),
                  "mfma_busy_frac": round(busy / (gui * SIMD_NUM), 4),
                  "mfma_instructions_f16": int(f16), "mfma_instructions_f32": int(f32),
                  "busy_cycles_per_instruction": round(busy / max(f16 + f32, 1.0), 2)}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]))
    json.dump({"command": cmd, "source_sha": kernel_source_sha(),
               "formula": "mfma_busy_frac = sum SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per kernel family, "
                          "dispatches serialised by the counter collection (each kernel alone on the chip)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:36s} {v['launches']:6d} launches  matrix pipe busy {100 * v['mfma_busy_frac']:5.1f} %  "
              f"({v['busy_cycles_per_instruction']:.1f} busy cycles per MFMA instruction)")


if __name__ == "__main__":
    main()
