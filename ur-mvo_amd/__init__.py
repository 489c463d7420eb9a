"""ur-mvo_amd: MI355X-native front-end of UR-MVO (SuperPoint -> SuperGlue ->
epipolar RANSAC) as hand-written HIP kernels behind a C ABI (include/urf.h).

The directory name has a hyphen (fixed by the project layout), so import it with
    importlib: see load() in the repo-root helpers (tests/conftest.py, bench.py).
"""
from . import _lib, dist, frontend, pipeline, synth, weights_io  # noqa: F401
from .frontend import (PointMatching, SuperGlue, SuperGlueConfig, SuperPoint,  # noqa: F401
                       SuperPointConfig)
