"""A batch and its shards must give the same BITS: the big-batch build of the wave kernel (B > 2 048: rows fetched / stored as
slabs through LDS, its own translation unit and scheduler) against the small-batch build on independently generated shards, for K4 and
for the fused pipeline, several track lengths, both noise layouts and both synthetic workloads.  usage: stress_shards.py [SEEDS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gps_optimize_slam_amd import batch as B

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bad = 0; t0 = time.time(); runs = 0
for seed in range(seeds):
    for N in (65, 130, 200, 271, 300, 640, 777, 1000, 1025):
        for variant in (0, 1):
            nb = 2304 if N >= 640 else 3072
            parts = 3
            per = nb // parts
            full = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=100 + seed, variant=variant)
            shards = [B.TrajectoryBatch.synthetic(per, N, layout=0, seed=100 + seed, traj0=k * per, variant=variant) for k in range(parts)]
            for name, fn in (("K4", lambda b: B.ekf_fuse_batch(b).host_traj_major()), ("pipeline", lambda b: B.fuse_pipeline_batch(b)[0].host_traj_major())):
                pf, qf, sf = fn(full)
                ps = [fn(s) for s in shards]
                ok = (np.array_equal(np.concatenate([x[0] for x in ps]), pf, equal_nan=True) and np.array_equal(np.concatenate([x[1] for x in ps]), qf, equal_nan=True)
                      and np.array_equal(np.concatenate([x[2] for x in ps]), sf))
                runs += 1
                if not ok:
                    bad += 1
                    dq = int((np.concatenate([x[1] for x in ps]) != qf).sum()); dp = int((np.concatenate([x[0] for x in ps]) != pf).sum())
                    print(f"MISMATCH seed {seed} N {N} variant {variant} {name}: {dp} position / {dq} quaternion components differ", flush=True)
print(f"{runs} batch-vs-shards comparisons, {bad} with any differing bit, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
