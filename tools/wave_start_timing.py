"""Diagnostic: when does every wave of the trajectory-major kernel start and finish, relative to the first one?
Needs libgsf.so built with -DGSF_WAVE_START_TIMING.  usage: wave_start_timing.py B N [pipeline]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gps_optimize_slam_amd import batch as B

nb, n = int(sys.argv[1]), int(sys.argv[2])
pipe = len(sys.argv) > 3
B.context().set_option("duo_kernel", int(os.environ.get("DUO", "0")))
bj = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
oj = B.FusedPoses(bj.layout, nb, n, "cuda")
oj.status = torch.zeros(2 * nb, dtype=torch.int32, device="cuda")
fn = (lambda: B.fuse_pipeline_batch(bj, out=oj, fit_rows=os.environ.get('FIT_ROWS', 'reference'))) if pipe else (lambda: B.ekf_fuse_batch(bj, out=oj))
for _ in range(5):
    fn()
torch.cuda.synchronize()
st = oj.status.cpu().numpy().astype(np.int64)
start, end = st[:nb], st[nb:]
t0 = start.min()
s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 10, 50, 90, 99, 100)]
print("wave start  (us after the first wave; min/10/50/90/99/max):", q(s_us))
print("wave finish (us after the first wave started):             ", q(e_us))
print("wave duration (us):                                         ", q(e_us - s_us))
# which tracks are the slow ones?  status words from a second, normal context are not available in this build: classify by the
# generator's outage plan instead (valid mask of the batch)
v = bj.valid.cpu().numpy().reshape(nb, n).astype(bool)
dur = e_us - s_us
has_out = ~v.all(axis=1)
first_bad = np.where(has_out, (~v).argmax(axis=1), -1)
out_len = (~v).sum(axis=1)
ends_out = ~v[:, -1]
mid = has_out & ~ends_out & (first_bad > 0)
for name, m in (("no outage", ~has_out), ("mid outage (recovers -> RTS / sharp turn)", mid), ("starts in outage", has_out & (first_bad == 0)), ("ends in outage", ends_out)):
    if m.any():
        print(f"{name:45s} n={int(m.sum()):4d}  duration median {np.median(dur[m]):6.2f}  max {dur[m].max():6.2f} us")
if mid.any():
    chunks_spanned = (first_bad[mid] + out_len[mid]) // 64 - first_bad[mid] // 64
    for c in np.unique(chunks_spanned):
        mm = chunks_spanned == c
        print(f"   mid outage spanning {int(c)} chunk boundaries: n={int(mm.sum()):3d} median {np.median(dur[mid][mm]):6.2f} max {dur[mid][mm].max():6.2f} us")
sl = np.argsort(-dur)[:5]
print("slowest tracks:", [(int(i), round(float(dur[i]), 2), int(first_bad[i]), int(out_len[i])) for i in sl], "(index, us, first outage pose, outage length)")
