#!/usr/bin/env python3
"""bench.py -- fused poses/sec of the GPS<->SLAM fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3] [--no-cpu-baseline]

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident trajectories: one launch of the fused
pipeline kernel (Umeyama fit on the valid rows -> Sim3 of pose 0 -> EKF predict/update + per-outage RTS), i.e. steps
3-5 of the reference's main_process_gui (EKFGPSSLAM.py:1002-1010) for every trajectory of the batch.  Two mappings exist
(DESIGN.md): wave-per-trajectory scans on the trajectory-major layout (default; fastest at C2 and at C3) and
lane-per-trajectory recursion on the time-major layout (--layout time).
Workloads (BASELINE.json configs):
  c2 (default, configs[1]) 1k synthetic KITTI-04-length (271-pose) trajectories per GPU
  c3 (configs[2])          100k synthetic 1k-pose trajectories per GPU (HBM-bound regime)
N > 1: one process per GPU (torch.distributed / RCCL), trajectories sharded by contiguous id blocks (weak scaling: the
per-GPU batch is fixed, no collective on the data path); the timed region is the K steps plus the ONE RCCL all-gather that
collects the fused poses (north star / SURVEY 8e).  The collect is also reported on its own ("collect": all-gather time,
received GB/s per rank, compute-only rate, and the rate when every step's result is gathered everywhere).
Prints ONE JSON line on rank 0 (contract in the round brief): metric/value/unit + roofline + cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_POSE = 145          # 89 B read (ts 8, pos 24, quat 32, gps 24, valid 1) + 56 B written
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
WORKLOADS = {
    "c2": dict(B=1000, N=271, name="C2: 1k synthetic KITTI-04-length (271-pose) trajectories per GPU (BASELINE configs[1])"),
    "c3": dict(B=100_000, N=1000, name="C3: 100k synthetic 1k-pose trajectories per GPU (BASELINE configs[2], HBM-bound regime)"),
}


def cpu_baseline(B_mod, torch, N, target_seconds=10.0):
    """The oracle (dense-7x7 C restatement of the reference path, single thread) timed on this box's host cores on a bounded
    sample of the same synthetic workload.  Checker code, timed here only as the reported CPU baseline."""
    import numpy as np
    from oracle import oracle as orc
    probe_B = 256
    def run(nb, seed):
        b = B_mod.TrajectoryBatch.synthetic(nb, N, layout=B_mod.LAYOUT_TRAJ_MAJOR, seed=seed)
        h = b.host_traj_major()
        del b
        t0 = time.perf_counter()                   # the same step as the GPU: Umeyama(valid rows) -> Sim3(pose 0) -> EKF+RTS
        orc.fuse_pipeline_batch(h["ts"], h["pos"], h["quat"], h["gps"], h["valid"])
        return time.perf_counter() - t0, nb * N
    dt, poses = run(probe_B, 99)
    rate = poses / dt
    nb = int(max(probe_B, min(target_seconds * rate / N, 3e9 / (N * 160))))    # ~target_seconds of work, <= ~3 GB host
    dt, poses = run(nb, 100)
    return {"value": poses / dt, "unit": "fused poses/s", "cores": 1, "kind": "port",
            "sample": f"{nb} synthetic {N}-pose trajectories ({poses} poses, {dt:.1f} s) through oracle/gsf_oracle.c "
                      f"(dense 7x7 EKF+RTS + Umeyama), 1 thread of {os.cpu_count()} host cores"}


def profiled_traffic(workload, kernel, grid_threads):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json, produced by
    tools/make_profiles.sh + tools/collect_profiles.py: separate --pmc runs of this same bench command; FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as read).  None when no profile of this kernel/grid is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1])).get(workload, {})
        key = f"{kernel} grid={grid_threads}"
        if key in d:
            return d[key]["hbm_bytes"], os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--ekf-variant", type=int, default=None, help="K4 tuning variant (gsf_set_option ekf_variant)")
    ap.add_argument("--kernel", choices=["pipeline", "ekf"], default="pipeline", help="step = fused pipeline (default) or K4 only")
    ap.add_argument("--layout", choices=["traj", "time"], default=None, help="traj = trajectory-major (wave-per-trajectory kernel), time = time-major (lane-per-trajectory kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra C3 / per-kernel measurements")
    ap.add_argument("--set-option", action="append", default=[], metavar="KEY=VALUE", help="gsf_set_option tuning knob (e.g. duo_kernel=0)")
    args = ap.parse_args()

    import torch
    from gps_optimize_slam_amd import batch as B
    from gps_optimize_slam_amd import distributed as D

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible -- the fusion path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    local = local % torch.cuda.device_count()   # one rank per GPU on a full node; ranks share a GPU only in a gloo rehearsal
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    wl = WORKLOADS[args.workload]
    Bn, N = wl["B"], wl["N"]
    steps = args.steps if args.steps is not None else (200 if args.workload == "c2" else 10)
    warmup = args.warmup if args.warmup is not None else (20 if args.workload == "c2" else 2)
    layout_name = args.layout or "traj"
    LAYOUT = B.LAYOUT_TRAJ_MAJOR if layout_name == "traj" else B.LAYOUT_TIME_MAJOR
    ctx = B.context()
    if args.ekf_variant is not None:
        ctx.set_option("ekf_variant", args.ekf_variant)
    for kv in args.set_option:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))

    def make_step(batch):
        out = B.FusedPoses(batch.layout, batch.B, batch.N, dev)
        if args.kernel == "pipeline":
            f = dict(dtype=torch.float64, device=dev)
            R, t, s = torch.empty((batch.B, 9), **f), torch.empty((batch.B, 3), **f), torch.empty((batch.B,), **f)
            import ctypes as C
            from gps_optimize_slam_amd import _lib
            cfg = _lib.EkfConfig.from_config(B.CONFIG)
            L, h, p = _lib.load(), ctx.handle, B._p
            def launch():
                _lib.check(L.gsf_fuse_pipeline_batch_dev(h, batch.layout, p(batch.ts), p(batch.pos), p(batch.quat), p(batch.gps), p(batch.valid),
                                                         C.byref(cfg), batch.B, batch.N, p(R), p(t), p(s), p(out.pos), p(out.quat), p(out.status)))
        else:
            def launch():
                B.ekf_fuse_batch(batch, out=out)
        return launch, out

    # ---- the timed workload: this rank's shard of world*B trajectories (ids [rank*B, (rank+1)*B))
    batch = B.TrajectoryBatch.synthetic(Bn, N, layout=LAYOUT, seed=20250523, traj0=rank * Bn)
    launch, out = make_step(batch)
    gather_bufs = None
    if world > 1:
        gather_bufs = (torch.empty((world * out.pos.shape[0],) + tuple(out.pos.shape[1:]), dtype=torch.float64, device=dev),
                       torch.empty((world * out.quat.shape[0],) + tuple(out.quat.shape[1:]), dtype=torch.float64, device=dev))

    def collect():                              # the job's ONE collect: fused poses of every shard to every GPU (RCCL over xGMI)
        D._gather(out.pos, world, gather_bufs[0])
        D._gather(out.quat, world, gather_bufs[1])

    for _ in range(warmup):
        launch()
    if world > 1:
        collect()                               # communicator set-up stays outside the timed region
    D.barrier(dev); torch.cuda.synchronize()
    # Timed region = the K fusion steps of this rank's shard (no collective on the data path: trajectories are independent) +
    # the single all-gather that collects the fused poses (north star / SURVEY 8e), bracketed by barrier + synchronize.
    # HIP events on torch's current stream == the stream the kernels are launched on (B.context()).  One pair around the K
    # launches: at C2 a launch is ~25 us, so per-launch event pairs would make the loop host-bound and pad the gaps.
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    ev0.record()
    for k in range(steps):
        launch()
    ev1.record()
    if world > 1:
        collect()
    ev2.record()
    torch.cuda.synchronize(); D.barrier(dev)
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    kern_ms = ev0.elapsed_time(ev1) / steps     # back-to-back launches of the one kernel: span / K = average launch duration
    collect_ms = ev1.elapsed_time(ev2) if world > 1 else 0.0
    poses_per_step = world * Bn * N
    value = poses_per_step * steps / elapsed
    alg_bytes = Bn * N * ALG_BYTES_PER_POSE
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    if layout_name == "traj":
        # <PIPELINE, SMALLBATCH>: up to 2 048 tracks the build with inlined cold blocks is launched (same arithmetic)
        kernel_name = "ekf_wave_kernel<%s, %s>" % ("true" if args.kernel == "pipeline" else "false", "true" if Bn <= 2048 else "false")
        grid_threads = Bn * 64
    else:
        kernel_name = "fuse_pipeline_kernel" if args.kernel == "pipeline" else "ekf_fuse_kernel"
        grid_threads = ((Bn + 63) // 64) * 64
    traffic, traffic_src = profiled_traffic(args.workload, kernel_name, grid_threads)
    result = {
        "metric": "fused poses/sec (whole node)", "value": value, "unit": "fused poses/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wl["name"], "trajectories_per_gpu": Bn, "poses_per_trajectory": N, "layout": "trajectory-major AoS (wave-per-trajectory scans)" if layout_name == "traj" else "time-major SoA (lane-per-trajectory)", "step": args.kernel,
                   "parallelism": f"trajectory-sharded x{world}" + (", one RCCL all-gather of the fused poses after the K steps (inside the timed region)" if world > 1 else "")},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms},
    }
    if world > 1:
        # the collect, reported on its own (compute-only and compute+gather separately): bytes each rank RECEIVES from the
        # others over xGMI / time of the single all-gather, and -- as a second, short leg -- the rate when EVERY step's result is
        # collected everywhere (all-gather issued after each step: link-bound by construction, 56 B/pose over xGMI against
        # 145 B/pose over HBM)
        recv = (world - 1) * Bn * N * 56
        k2 = max(1, min(steps, 20))
        D.barrier(dev); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(k2):
            launch(); collect()
        torch.cuda.synchronize(); D.barrier(dev)
        el2 = D.max_over_ranks(time.perf_counter() - t1, dev)
        result["collect"] = {"allgather_ms": collect_ms, "recv_bytes_per_rank": recv, "recv_GBps_per_rank": recv / (collect_ms * 1e-3) / 1e9 if collect_ms > 0 else None,
                             "compute_only_poses_per_s": world * Bn * N / (kern_ms * 1e-3),
                             "collect_every_step": {"steps": k2, "ms_per_step": el2 / k2 * 1e3, "poses_per_s": poses_per_step * k2 / el2}}
    # ---- accuracy gate on this run: ATE RMSE of the GPU result vs the CPU oracle on a sample of the timed batch
    if rank == 0:
        import numpy as np
        from oracle import oracle as orc
        nb = min(Bn, 64)
        sub = B.TrajectoryBatch.synthetic(nb, N, layout=LAYOUT, seed=20250523, traj0=0)
        so = B.ekf_fuse_batch(sub)
        h = sub.host_traj_major()
        p, q, st = so.host_traj_major()
        po, qo, sto = orc.fuse_batch(h["ts"], h["pos"], h["quat"], h["gps"], h["valid"], h["init_pos"], h["init_quat"])
        result["ate_rmse_vs_cpu_ref_m"] = float(np.sqrt(np.mean(np.sum((p - po) ** 2, axis=2))))
        result["max_abs_pos_err_m"] = float(np.abs(p - po).max())
        result["status_bits_equal"] = bool((st == sto).all())
    # ---- extras (rank 0, N=1): the HBM-regime config and the per-kernel figures
    if world == 1 and not args.no_extra:
        extra = {}
        del batch, out, launch
        torch.cuda.empty_cache()
        def timed(fn, reps):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        for name, (b3, n3, reps) in {"c3_100k_x_1k": (100_000, 1000, 5), "c2_1k_x_271": (1000, 271, 100)}.items():
            ab = b3 * n3 * ALG_BYTES_PER_POSE
            extra[name] = {}
            for lname, lay in (("time_major_lane_per_traj", B.LAYOUT_TIME_MAJOR), ("traj_major_wave_per_traj", B.LAYOUT_TRAJ_MAJOR)):
                bt = B.TrajectoryBatch.synthetic(b3, n3, layout=lay, seed=1)
                o = B.FusedPoses(bt.layout, b3, n3, dev)
                ms_e = timed(lambda: B.ekf_fuse_batch(bt, out=o), reps)
                ms_p = timed(lambda: B.fuse_pipeline_batch(bt, out=o), reps)
                extra[name][lname] = {"ekf_kernel_ms": ms_e, "ekf_poses_per_s": b3 * n3 / ms_e * 1e3, "ekf_alg_GBps": ab / ms_e / 1e6,
                                      "ekf_hbm_frac": ab / ms_e / 1e6 / HBM_PEAK_GBS, "pipeline_kernel_ms": ms_p,
                                      "pipeline_poses_per_s": b3 * n3 / ms_p * 1e3, "pipeline_alg_GBps": ab / ms_p / 1e6,
                                      "pipeline_hbm_frac": ab / ms_p / 1e6 / HBM_PEAK_GBS}
                del bt, o
                torch.cuda.empty_cache()
        result["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(B, torch, N)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
