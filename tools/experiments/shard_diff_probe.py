import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gps_optimize_slam_amd import batch as B
nb, N = 3000, 200
full = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=11)
p0, q0, s0 = B.ekf_fuse_batch(full).host_traj_major()
parts = [B.ekf_fuse_batch(B.TrajectoryBatch.synthetic(1000, N, layout=0, seed=11, traj0=k * 1000)).host_traj_major() for k in range(3)]
q1 = np.concatenate([x[1] for x in parts]); p1 = np.concatenate([x[0] for x in parts])
d = np.argwhere(q1 != q0)
print("differing quaternion components:", len(d), " positions differing:", int((p1 != p0).sum()))
tr = np.unique(d[:, 0]); print("trajectories:", len(tr), tr[:10])
h = full.host_traj_major()
for b in tr[:6]:
    rows = np.unique(d[d[:, 0] == b][:, 1])
    v = h["valid"][b]
    print(f"traj {b}: status {s0[b]:#x} rows {rows[:12]} ... ({len(rows)})  valid around first: {v[max(0, rows[0]-3):rows[0]+4]}  chunk {rows[0] // 64} lane {rows[0] % 64}")
    print("   quat in:", h["quat"][b][rows[0]], " norm", np.linalg.norm(h["quat"][b][rows[0]]))
# ---- detail of the first few differing rows
np.set_printoptions(precision=17, linewidth=200)
for b in tr[:3]:
    rows = np.unique(d[d[:, 0] == b][:, 1])
    for r in rows[:2]:
        print(f"traj {b} row {r}: in {h['quat'][b][r]}")
        print(f"   big   {q0[b, r]}")
        print(f"   small {q1[b, r]}")
        print(f"   big==small per comp {q0[b, r] == q1[b, r]}; neighbours equal: {np.array_equal(q0[b, r-1], q1[b, r-1])} {np.array_equal(q0[b, r+1], q1[b, r+1]) if r + 1 < N else None}")
        print(f"   valid[r-2:r+3] {h['valid'][b][r-2:r+3]}  gps nan {np.isnan(h['gps'][b][r-2:r+3]).any(axis=1)}")
