import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gps_optimize_slam_amd import batch as B
B.context().set_option("duo_kernel", 1)
exec(open("tools/chunk_timing.py").read().split("import torch\nfrom gps_optimize_slam_amd import batch as B\n")[1])
