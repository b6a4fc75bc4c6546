import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
from gps_optimize_slam_amd import ekfgpsslam as E
import json
print(json.dumps(E.benchmark_c1(10), indent=1))
pr = cProfile.Profile(); pr.enable(); E.benchmark_c1(5); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
