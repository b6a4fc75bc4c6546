"""gps_optimize_slam_amd -- MI355X-native GPS<->SLAM trajectory fusion.

Drop-in for the hot path of A2ureeE/GPS-optimize-SLAM's EKFGPSSLAM.py: the same module-level
functions (gps_optimize_slam_amd.ekfgpsslam) backed by hand-written gfx950 HIP kernels behind a C ABI
(include/gsf.h, libgsf.so), plus batched device entry points (gps_optimize_slam_amd.batch).

There is no CPU fallback: importing is cheap, but the first call that needs the kernels raises
GsfError if libgsf.so is missing or no MI355X is visible.
"""
from ._lib import GsfError, build_library, library_path  # noqa: F401

__all__ = ["GsfError", "build_library", "library_path"]
__version__ = "0.1.0"
