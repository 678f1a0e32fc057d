"""vil_sensor_fusion_amd -- MI355X-native drop-in for the arithmetic behind gtsam_fusion's
GraphManager (reference: gtsam_fusion/include/gtsam_fusion/GraphManager.h:40-96).

The compute lives in libvilfusion.so (hand-written HIP for gfx950, C ABI in
include/vilfusion.h).  This package is the thin host-side mirror used by the tests, the
benchmark and Python callers; it never computes factor math on the CPU and raises if the
library or a GPU is missing.
"""
from ._lib import VilFusionError, lib, lib_path  # noqa: F401
from .engine import Engine, EngineOpts  # noqa: F401

__all__ = ["VilFusionError", "lib", "lib_path", "Engine", "EngineOpts"]
from .graph_manager import GraphManager  # noqa: F401,E402
from .sensor_manager import SensorManager  # noqa: F401,E402
