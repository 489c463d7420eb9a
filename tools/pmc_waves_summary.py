"""Where the waves of each kernel family spend their cycles, and how much vector work runs beside the matrix pipe: one rocprofv3 PMC
pass of bench.py (eight SQ counters + GRBM_GUI_ACTIVE), summarised into profiles/.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES \
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_waves -o pmc -- python3 bench.py ...
    python tools/pmc_waves_summary.py gpurun_out/pmc_waves/pmc_results.db profiles/r05_pmc_waves.json "<the bench command>" \
        [profiles/r05_pmc_mfma.json]                                  (tools/gpu_pmc_waves.sh does both)

MI355X_MICROARCH.md, "rocprofv3 PMC slots": SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall
(MFMA dependency / pipe busy), SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to about SQ_WAVE_CYCLES (all four
in quad-cycles).  SQ_VALU_MFMA_COEXEC_CYCLES = cycles in which vector and matrix instructions execute together: against the
matrix pipe's busy cycles of the same launches (the optional fourth argument, tools/pmc_mfma_summary.py) it says how much of the
MFMA time has VALU work under it.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = the share of LDS cycles lost to bank conflicts.
"""
import collections
import json
import os
import sqlite3
import sys

from pmc_summary import family, kernel_source_sha


def main():
    db, out, cmd = sys.argv[1:4]
    mfma = json.load(open(sys.argv[4]))["kernels"] if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else {}
    cur = sqlite3.connect(db).cursor()
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for name, cname, val, did in cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        k = family(name)
        if k:
            fam[k][cname] += float(val)
            disp[k].add(did)
    res = {}
    for k, v in fam.items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        r = {"launches": len(disp[k]),
             "wave_cycles_parked_frac": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4),
             "wave_cycles_issue_stall_frac": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
             "wave_cycles_issuing_frac": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
             "wave_cycles_issuing_valu_frac": round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
             "lds_bank_conflict_frac": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
             "valu_mfma_coexec_cycles": int(v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0))}
        m = mfma.get(k)
        if m and m["launches"] == r["launches"] and m["mfma_busy_cycles"] > 0:
            r["coexec_over_mfma_busy"] = round(r["valu_mfma_coexec_cycles"] / m["mfma_busy_cycles"], 4)
        res[k] = r
    order = sorted(res, key=lambda k: -fam[k].get("SQ_WAVE_CYCLES", 0.0))
    res = {k: res[k] for k in order}
    json.dump({"command": cmd, "source_sha": kernel_source_sha(),
               "units": "fractions of SQ_WAVE_CYCLES (parked = s_waitcnt / barrier, issue stall = dependency / pipe busy, issuing); "
                        "coexec_over_mfma_busy = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES of the same launches",
               "kernels": res}, open(out, "w"), indent=1)
    for k, r in res.items():
        print(f"{k:34s} {r['launches']:5d} launches  parked {100 * r['wave_cycles_parked_frac']:5.1f} %  issue stall "
              f"{100 * r['wave_cycles_issue_stall_frac']:5.1f} %  issuing {100 * r['wave_cycles_issuing_frac']:5.1f} % (VALU "
              f"{100 * r['wave_cycles_issuing_valu_frac']:5.1f} %)  LDS conflicts {100 * r['lds_bank_conflict_frac']:4.1f} %  "
              f"VALU beside MFMA {100 * r.get('coexec_over_mfma_busy', float('nan')):5.1f} % of the matrix pipe's busy cycles")


if __name__ == "__main__":
    main()
