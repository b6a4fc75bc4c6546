"""Diagnostic: time stamps of wave 0 of the trajectory-major K4 / pipeline (needs libgsf.so built with -DGSF_CHUNK_TIMING).
usage: chunk_timing.py B N [pipeline]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B

nb, n = int(sys.argv[1]), int(sys.argv[2])
pipe = len(sys.argv) > 3
bj = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
oj = B.FusedPoses(bj.layout, nb, n, "cuda")
oj.status.zero_()
fn = (lambda: B.fuse_pipeline_batch(bj, out=oj, fit_rows=os.environ.get('FIT_ROWS', 'reference'))) if pipe else (lambda: B.ekf_fuse_batch(bj, out=oj))
for _ in range(5):
    fn()
torch.cuda.synchronize()
nch = (n + 63) // 64
raw = oj.status.cpu().numpy()
st = raw[:2 * (8 + nch)].reshape(-1, 2).astype("int64")
if raw[30] or raw[31]:
    print(f"(stamp 15, gap check of the last round done: cycles {int(raw[30]) - int(raw[0])}, {(int(raw[31]) - int(raw[1])) / 100.0:.2f} us after entry)")
names = ["entry", "rows of the (last) round arrived", "moments loop done", "reductions done", "umeyama_finalize done", "pose 0 aligned", "prelude done",
         "chunk 0 arrived"] + [f"chunk {k} done" for k in range(nch)]
c0, w0 = st[0]
prev = (c0, w0)
for nm, (c, w) in zip(names, st):
    if c == 0 and w == 0:
        continue
    print(f"{nm:22s} cycles {c - c0:8d} (+{c - prev[0]:6d})   {(w - w0) / 100.0:7.2f} us (+{(w - prev[1]) / 100.0:5.2f})")
    prev = (c, w)
print("raw status[:32] =", oj.status.cpu().numpy()[:32].tolist())
