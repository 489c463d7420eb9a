"""Convert the reference's weight files (.pth state dict or .onnx) into URFW containers:
    python tools/import_weights.py --superpoint superpoint_v1.pth --out superpoint_v1.urfw
    python tools/import_weights.py --superglue superglue_indoor_sim_int32.onnx --out superglue_indoor.urfw
then point superpoint.engine_file / superglue.engine_file (configs/configs_aqua.yaml:16,32) at them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

if __name__ == "__main__":
    load_pkg().weights_io.main()
