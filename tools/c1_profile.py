"""cProfile of the single-trajectory drop-in at the C1 shape (the committed fixtures; GPU box)."""
import cProfile
import json
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gps_optimize_slam_amd import ekfgpsslam as E  # noqa: E402

k, g = np.load(os.path.join(ROOT, "tests/golden/kat_bundled.npz")), np.load(os.path.join(ROOT, "tests/golden/c1_combined.npz"))
slam = {"timestamps": k["ts"], "positions": k["pos"], "quaternions": k["quat"]}
args = (slam, g["gps_t_raw"], g["lat"], g["lon"], g["alt"])
print(json.dumps(E.benchmark_c1(*args, repeats=10), indent=1))
pr = cProfile.Profile(); pr.enable(); E.benchmark_c1(*args, repeats=5); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
