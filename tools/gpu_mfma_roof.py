"""What the chip sustains on the split-f16 MFMA inner loop alone (urf_probe_mfma_roof): the ceiling to compare the
GEMM / conv / attention kernels with.    python tools/gpu_mfma_roof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

L = load_pkg()._lib.lib()
for waves in (4, 8, 16):
    for iters in (2000, 20000):
        pf, ghz = C.c_float(0), C.c_float(0)
        assert L.urf_probe_mfma_roof(0, waves, iters, C.byref(pf), C.byref(ghz)) == 0
        print(f"{waves:2d} waves/CU, {iters:6d} x 24 MFMA per wave: {pf.value:.3f} PFLOP/s of MFMA issue "
              f"({pf.value / 3:.3f} PFLOP/s logical at 3 MFMAs per product), in-kernel clock {ghz.value:.2f} GHz")
