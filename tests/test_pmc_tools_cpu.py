"""tools/pmc_mfma_summary.py and tools/pmc_waves_summary.py on a synthetic rocprofv3 database: the arithmetic that turns counter
rows into the fractions quoted in DESIGN.md section 7 and in bench.py's `roofline.mfma_busy_counter` (no GPU needed)."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, rows):
    con = sqlite3.connect(path)
    con.execute("create table counters_collection (dispatch_id integer, kernel_name text, counter_name text, value real)")
    con.executemany("insert into counters_collection values (?, ?, ?, ?)", rows)
    con.commit()
    con.close()


def test_matrix_pipe_busy_fraction_and_cycles_per_instruction(tmp_path):
    gemm = "void urf::h2gemm_glds_kernel<0>(urf::H2Args)"
    conv = "void urf::conv_mfma_kernel<9, true, true, 4, true>(urf::ConvArgs)"
    rows = []
    for d in (1, 2):                              # two launches of the f16 GEMM: 1000 MFMAs each, 16 busy cycles per MFMA
        rows += [(d, gemm, "SQ_VALU_MFMA_BUSY_CYCLES", 16000.0), (d, gemm, "SQ_INSTS_VALU_MFMA_F16", 1000.0),
                 (d, gemm, "SQ_INSTS_VALU_MFMA_F32", 0.0), (d, gemm, "GRBM_GUI_ACTIVE", 8 * 100.0)]
    rows += [(3, conv, "SQ_VALU_MFMA_BUSY_CYCLES", 32.0 * 1024 * 50), (3, conv, "SQ_INSTS_VALU_MFMA_F32", 1024.0 * 50),
             (3, conv, "SQ_INSTS_VALU_MFMA_F16", 0.0), (3, conv, "GRBM_GUI_ACTIVE", 8 * 2000.0),
             (4, "some_other_kernel", "SQ_VALU_MFMA_BUSY_CYCLES", 5.0), (4, "some_other_kernel", "GRBM_GUI_ACTIVE", 8.0)]
    db, out = str(tmp_path / "pmc.db"), str(tmp_path / "mfma.json")
    _db(db, rows)
    subprocess.check_call([sys.executable, "pmc_mfma_summary.py", db, out, "cmd"], cwd=os.path.join(ROOT, "tools"))
    d = json.load(open(out))
    k = d["kernels"]
    assert set(k) == {"h2gemm", "conv_mfma_kernel<9,pool,fuse1a>"}
    assert k["h2gemm"]["launches"] == 2
    assert abs(k["h2gemm"]["mfma_busy_frac"] - 32000.0 / (200.0 * 1024)) < 1e-4
    assert k["h2gemm"]["busy_cycles_per_instruction"] == 16.0
    assert abs(k["conv_mfma_kernel<9,pool,fuse1a>"]["mfma_busy_frac"] - 0.8) < 1e-4      # 32 x 1024 x 50 / (2000 x 1024)
    assert k["conv_mfma_kernel<9,pool,fuse1a>"]["busy_cycles_per_instruction"] == 32.0
    sys.path.insert(0, ROOT)
    import bench
    assert d["source_sha"] == bench.kernel_source_sha()

    # the wave-state summary, with the matrix-pipe summary as its fourth argument
    rows = [(1, gemm, "SQ_WAVE_CYCLES", 1000.0), (1, gemm, "SQ_WAIT_ANY", 470.0), (1, gemm, "SQ_WAIT_INST_ANY", 395.0),
            (1, gemm, "SQ_ACTIVE_INST_ANY", 135.0), (1, gemm, "SQ_ACTIVE_INST_VALU", 68.0),
            (1, gemm, "SQ_VALU_MFMA_COEXEC_CYCLES", 1600.0), (1, gemm, "SQ_LDS_BANK_CONFLICT", 10.0),
            (1, gemm, "SQ_LDS_IDX_ACTIVE", 100.0), (1, gemm, "GRBM_GUI_ACTIVE", 800.0),
            (2, gemm, "SQ_WAVE_CYCLES", 1000.0), (2, gemm, "SQ_WAIT_ANY", 470.0), (2, gemm, "SQ_WAIT_INST_ANY", 395.0),
            (2, gemm, "SQ_ACTIVE_INST_ANY", 135.0), (2, gemm, "SQ_ACTIVE_INST_VALU", 68.0),
            (2, gemm, "SQ_VALU_MFMA_COEXEC_CYCLES", 1600.0), (2, gemm, "SQ_LDS_BANK_CONFLICT", 10.0),
            (2, gemm, "SQ_LDS_IDX_ACTIVE", 100.0), (2, gemm, "GRBM_GUI_ACTIVE", 800.0)]
    db2, out2 = str(tmp_path / "waves.db"), str(tmp_path / "waves.json")
    _db(db2, rows)
    subprocess.check_call([sys.executable, "pmc_waves_summary.py", db2, out2, "cmd", out], cwd=os.path.join(ROOT, "tools"),
                          stdout=subprocess.DEVNULL)
    w = json.load(open(out2))["kernels"]["h2gemm"]
    assert w["launches"] == 2
    assert (w["wave_cycles_parked_frac"], w["wave_cycles_issue_stall_frac"], w["wave_cycles_issuing_frac"]) == (0.47, 0.395, 0.135)
    assert w["lds_bank_conflict_frac"] == 0.1
    assert w["coexec_over_mfma_busy"] == 0.1          # 3200 / 32000
