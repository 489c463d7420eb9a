"""Regenerates ur-mvo_amd/data/sp_desc_bias_seed0.npy (uses the CPU oracle).

bias = -W_convDb @ mean_cells(relu(convDa(...))) on a 240x320 calibration
texture (synth.base_frame(99, 240, 320)) with the seed-0 synthetic weights.
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "ur-mvo_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)
from oracle import oracle as O  # noqa: E402

w = synth.sp_weights(0, calibrated=False)
o = O.sp_dense(synth.pack_sp(w), synth.base_frame(99, 240, 320), want_layers=True)
mu = o["layers"][10].reshape(-1, 256).mean(0)
b = -(w["convDb"][0].reshape(256, 256) @ mu).astype(np.float32)
np.save(os.path.join(ROOT, "ur-mvo_amd", "data", "sp_desc_bias_seed0.npy"), b)
print("saved", b.shape, float(np.abs(b).max()))
