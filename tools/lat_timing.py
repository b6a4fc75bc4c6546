"""Diagnostic: stamps of main wave / helper wave of trajectory 0 of the latency kernel (libgsf built with -DGSF_CHUNK_TIMING)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B
nb, n = int(sys.argv[1]), int(sys.argv[2])
bj = B.TrajectoryBatch.synthetic(nb, n, layout=0, seed=1)
oj = B.FusedPoses(0, nb, n, "cuda"); oj.status.zero_()
for _ in range(5):
    B.fuse_pipeline_batch(bj, out=oj)
torch.cuda.synchronize()
st = oj.status.cpu().numpy()[:64].reshape(-1, 2).astype("int64")
names = {0: "main entry", 2: "moments loop done", 3: "reductions done", 4: "umeyama done", 5: "pose 0 aligned", 6: "prelude done", 7: "barrier passed", 8: "finish done", 19: "helper entry"}
for k in range(20, 25): names[k] = f"gains chunk {k - 20} done"
for k in range(26, 31): names[k] = f"scans chunk {k - 26} done"
w0 = st[0][1]
for k in sorted(names):
    c, w = st[k]
    if w: print(f"{names[k]:22s} {(w - w0) / 100.0:7.2f} us")
