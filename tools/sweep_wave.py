"""Times the trajectory-major K4 / pipeline kernels per tuning variant (GPU box only).  usage: sweep_wave.py [BxN ...] [--variants 0,5,8]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


args = [x for x in sys.argv[1:] if not x.startswith("--")]
variants = [0, 1, 2, 3, 4, 5]
for x in sys.argv[1:]:
    if x.startswith("--variants="):
        variants = [int(v) for v in x.split("=")[1].split(",")]
shapes = [tuple(int(v) for v in s.split("x")) for s in args] or [(1000, 271), (250, 271), (4000, 271), (100_000, 1000)]
ctx = B.context()
for (nb, n) in shapes:
    reps = 200 if nb * n < 5e6 else 5
    bj = B.TrajectoryBatch.synthetic(nb, n, layout=B.LAYOUT_TRAJ_MAJOR, seed=1)
    oj = B.FusedPoses(bj.layout, nb, n, "cuda")
    for v in variants:
        # 0 = default (automatic); 1..5 = forced poses per lane; 8 = chunk-parallel block kernel
        ctx.set_option("ekf_variant", 8 if v == 8 else 0)
        ctx.set_option("wave_ppl", v if 1 <= v <= 5 else 0)
        try:
            ctx.set_option("duo_kernel", {6: 0, 7: 1}.get(v, -1))      # 6 = two-wave pipeline kernel off, 7 = forced on
        except Exception:
            pass                                                        # an older library under GSF_LIBRARY (A/B runs)
        ms_e = timed(lambda: B.ekf_fuse_batch(bj, out=oj), reps)
        ms_p = timed(lambda: B.fuse_pipeline_batch(bj, out=oj), reps)
        print(json.dumps({"B": nb, "N": n, "variant": v, "ekf_us": round(ms_e * 1e3, 2), "pipeline_us": round(ms_p * 1e3, 2),
                          "ekf_frac": round(nb * n * 145 / (ms_e * 1e-3) / 8e12, 4), "pipeline_frac": round(nb * n * 145 / (ms_p * 1e-3) / 8e12, 4)}), flush=True)
    ctx.set_option("ekf_variant", 0); ctx.set_option("wave_ppl", 0)
    try:
        ctx.set_option("duo_kernel", -1)
    except Exception:
        pass
    del bj, oj
    torch.cuda.empty_cache()
