"""Times K4 / the fused pipeline for each tuning variant on C3- and C2-shaped synthetic batches (GPU box only)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

shapes = [(100_000, 1000, 5), (1000, 271, 30), (400_000, 271, 5)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split("x")) + (5,) for s in sys.argv[1:]]
ctx = B.context()
for (nb, n, reps) in shapes:
    bt = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TIME_MAJOR, seed=1)
    o = B.FusedPoses(bt.layout, nb, n, "cuda")
    for v in range(5):
        ctx.set_option("ekf_variant", v)
        ms = timed(lambda: B.ekf_fuse_batch(bt, out=o), reps)
        print(json.dumps({"B": nb, "N": n, "variant": v, "ekf_ms": round(ms, 4), "Gposes_s": round(nb * n / ms / 1e6, 3), "alg_TBps": round(nb * n * 145 / ms / 1e9, 3)}), flush=True)
    ctx.set_option("ekf_variant", 0)
    ms = timed(lambda: B.fuse_pipeline_batch(bt, out=o), reps)
    print(json.dumps({"B": nb, "N": n, "pipeline_ms": round(ms, 4), "Gposes_s": round(nb * n / ms / 1e6, 3), "alg_TBps": round(nb * n * 145 / ms / 1e9, 3)}), flush=True)
    if nb * n <= 2e8:
        bj = bt.to_layout(B.LAYOUT_TRAJ_MAJOR)
        oj = B.FusedPoses(bj.layout, nb, n, "cuda")
        ms = timed(lambda: B.ekf_fuse_batch(bj, out=oj), reps)
        print(json.dumps({"B": nb, "N": n, "wave_ekf_ms": round(ms, 4), "Gposes_s": round(nb * n / ms / 1e6, 3), "alg_TBps": round(nb * n * 145 / ms / 1e9, 3)}), flush=True)
        ms = timed(lambda: B.fuse_pipeline_batch(bj, out=oj), reps)
        print(json.dumps({"B": nb, "N": n, "wave_pipeline_ms": round(ms, 4), "Gposes_s": round(nb * n / ms / 1e6, 3), "alg_TBps": round(nb * n * 145 / ms / 1e9, 3)}), flush=True)
        for v, nm in ((0, "wave_ppl1"), (5, "wave_ppl2"), (8, "block")):
            ctx.set_option("ekf_variant", v)
            ms_e = timed(lambda: B.ekf_fuse_batch(bj, out=oj), reps)
            ms_p = timed(lambda: B.fuse_pipeline_batch(bj, out=oj), reps)
            print(json.dumps({"B": nb, "N": n, "traj_major": nm, "ekf_ms": round(ms_e, 4), "pipeline_ms": round(ms_p, 4)}), flush=True)
        ctx.set_option("ekf_variant", 0)
        del bj, oj
    del bt, o
    torch.cuda.empty_cache()
