"""What the chip sustains on the split-f16 MFMA inner loop alone (urf_probe_mfma_roof): the ceiling to compare the
GEMM / conv / attention kernels with.    python tools/gpu_mfma_roof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# urf_probe_mfma_roof is exported by the experiments build only (include/urf.h, #ifdef URF_EXPERIMENTS)
os.environ.setdefault("URF_LIB", os.path.join(ROOT, "ur-mvo_amd", "liburf_front_exp.so"))
from __graft_entry__ import load_pkg  # noqa: E402

L = load_pkg()._lib.lib()
MODES = ["registers only", "+ LDS fragment reads", "+ barrier per step", "+ LDS-DMA (L2 sources)", "+ LDS-DMA (HBM activations)"]
for mode in range(5):
    for waves in ((4, 8, 16) if mode == 0 else (8, 16)):
        for iters in (2000, 20000):
            pf, ghz = C.c_float(0), C.c_float(0)
            assert L.urf_probe_mfma_roof(0, waves, iters, mode, C.byref(pf), C.byref(ghz)) == 0, L.urf_last_error()
            print(f"mode {mode} ({MODES[mode]:28s}) {waves:2d} waves/CU, {iters:6d} x 24 MFMA per wave: {pf.value:.3f} PFLOP/s of MFMA "
                  f"issue ({pf.value / 3:.3f} logical), in-kernel clock {ghz.value:.2f} GHz", flush=True)
for waves in (4, 8, 16):
    for iters in (2000, 20000):
        pf, ghz = C.c_float(0), C.c_float(0)
        assert L.urf_probe_mfma_roof(0, waves, iters, 5, C.byref(pf), C.byref(ghz)) == 0, L.urf_last_error()
        print(f"mode 5 (fp32 MFMA 16x16x4, registers only) {waves:2d} waves/CU, {iters:6d} x 24 MFMA per wave: {pf.value * 1e3:.1f} TFLOP/s "
              f"(nominal peak 157.3), in-kernel clock {ghz.value:.2f} GHz", flush=True)
