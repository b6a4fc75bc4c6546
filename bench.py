#!/usr/bin/env python3
"""bench.py -- fused poses/sec of the GPS<->SLAM fusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5] [--fit-rows reference|all] [--no-cpu-baseline] [--no-extra]

A "step" is one pass of the hot path over one batch of synthetic, HBM-resident trajectories: one launch of the fused
pipeline kernel (Umeyama fit -> Sim3 of pose 0 -> EKF predict/update + per-outage RTS), i.e. steps 3-5 of the reference's
main_process_gui (EKFGPSSLAM.py:1002-1010) for every trajectory of the batch.  The rows of the fit are the ones main_process_gui
hands to its fit (--fit-rows reference, the default: first gap-free segment of the valid rows, <= 180 s, two fall-backs; ref :973-998)
or every valid row (--fit-rows all, the operator of rounds 1-3); the line reports both.
Workloads (BASELINE.json configs):
  c2 (default, configs[1]) 1k synthetic KITTI-04-length (271-pose) trajectories per GPU
  c3 (configs[2])          100k synthetic 1k-pose trajectories per GPU (HBM-bound regime)
  c5 (configs[4])          1.25M synthetic 1k-pose trajectories PER GPU (10M over 8 GPUs), fused chunk by chunk; every chunk's fused
                           poses are all-gathered over xGMI on a second stream while the next chunk is being fused
N > 1: one process per GPU.  `python bench.py --gpus N` starts its own N ranks (before anything touches the GPU in the parent);
under a launcher (torch.distributed.run: WORLD_SIZE set) it is one of the ranks.  Trajectories are sharded by contiguous id
blocks (weak scaling: fixed per-GPU batch, no collective on the data path); the timed region is the K steps plus the ONE RCCL
all-gather that collects the fused poses (north star / SURVEY 8e).  The collect is also reported on its own ("collect"), and a
C5-shaped leg ("c5") reports compute-only vs compute+gather rates and per-link GB/s.
Prints ONE JSON line on rank 0 (contract in the round brief): metric/value/unit + roofline + cpu_baseline.
"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_POSE = 145          # 89 B read (ts 8, pos 24, quat 32, gps 24, valid 1) + 56 B written
FIT_REREAD_BYTES_PER_POSE = 49    # the pipeline's Umeyama pass has to see pos (24) + gps (24) + valid (1) of every row before the filter starts
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
XGMI_LINK_GBS = 153.0             # per point-to-point link and direction; 7 links per GPU
WORKLOADS = {
    "c2": dict(B=1000, N=271, name="C2: 1k synthetic KITTI-04-length (271-pose) trajectories per GPU (BASELINE configs[1])"),
    "c3": dict(B=100_000, N=1000, name="C3: 100k synthetic 1k-pose trajectories per GPU (BASELINE configs[2], HBM-bound regime)"),
    "c5": dict(B=1_250_000, N=1000, name="C5: 1.25M synthetic 1k-pose trajectories per GPU = 10M over 8 GPUs (BASELINE configs[4]), chunked fuse + all-gather"),
}
SEED = 20250523


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: the parent starts the ranks and never touches the GPU (no torch import on this path)
# ----------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0 prints the line; every rank gets a deadline (a stalled communicator must not hang the parent: the ranks' own watchdog
    # fires first and exits non-zero, this is the backstop)
    limit = float(os.environ.get("GSF_BENCH_RANK_TIMEOUT_S", "900"))
    try:
        out0, _ = procs[0].communicate(timeout=limit)
        rc = procs[0].returncode
    except subprocess.TimeoutExpired:
        procs[0].kill()
        out0, _ = procs[0].communicate()
        rc = 124
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode or (124 if p.returncode is None else 0)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (checker code) timed on this box's host cores -- reported next to the GPU figure, never the target
# ----------------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """host cores this process may really use: the scheduler affinity and the cgroup CPU quota, not just os.cpu_count()"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return n


def reference_python_timing():
    """profiles/rNN_reference_timing.json: the REFERENCE ITSELF (EKFGPSSLAM.py, stubbed import) timed next to the oracle in the build container
    by tests/campaigns/time_reference.py -- the reference's files never travel to the GPU box, so this is a committed measurement, quoted."""
    try:
        d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_timing.json")))[-1]))
        h = d["headline"]
        return {"reference_python_poses_per_s": h["reference_python_poses_per_s"], "oracle_poses_per_s_same_inputs_same_box": h["oracle_poses_per_s_same_inputs"],
                "oracle_over_reference": h["oracle_over_reference"], "what": h["what"],
                "where": f"{d['where']}, {d['cpu']}, {d['cores_used']} core, {d['date']} (tests/campaigns/time_reference.py; the reference is single-threaded Python)",
                "steps_2_to_5_reference_poses_per_s": d["cases"]["synthetic_271_with_outage"]["steps_2_to_5"]["reference_poses_per_s"],
                "robust_fit_reference_ms": d["cases"]["synthetic_271_with_outage"]["compute_sim3_transform_robust"]["reference_ms"]}
    except Exception as e:
        return {"error": f"profiles/rNN_reference_timing.json unreadable: {e}"[:200]}


def cpu_baseline(B_mod, N, target_seconds=8.0, fit_rows="reference"):
    """The oracle (dense-7x7 C restatement of the reference path) on a bounded sample of the same synthetic workload: (i) ONE
    thread -- like for like with the single-threaded reference -- and (ii) trajectory-parallel over every host core (threads:
    ctypes releases the GIL and the C code has no shared state)."""
    import concurrent.futures as cf

    import numpy as np

    from oracle import oracle as orc

    def sample(nb, seed):
        b = B_mod.TrajectoryBatch.synthetic(nb, N, layout=B_mod.LAYOUT_TRAJ_MAJOR, seed=seed)
        h = b.host_traj_major()
        del b
        return h

    def run(h):                                    # the same step as the GPU: Umeyama(valid rows) -> Sim3(pose 0) -> EKF+RTS
        t0 = time.perf_counter()
        orc.fuse_pipeline_batch(h["ts"], h["pos"], h["quat"], h["gps"], h["valid"], fit_rows=fit_rows)
        return time.perf_counter() - t0

    h = sample(256, 99)
    rate = 256 * N / run(h)
    nb = int(max(256, min(target_seconds * rate / N, 3e9 / (N * 160))))        # ~target_seconds of work, <= ~3 GB of host arrays
    h = sample(nb, 100)
    dt1 = run(h)
    one = nb * N / dt1
    cores = usable_cores()
    res = {"value": one, "unit": "fused poses/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
           "usable_cores": cores,
           "sample": f"{nb} synthetic {N}-pose trajectories ({nb * N} poses, {dt1:.1f} s) through oracle/gsf_oracle.c "
                     f"(dense 7x7 EKF+RTS + Umeyama), 1 thread of {os.cpu_count()} host cores ({cores} usable by this process: affinity / cgroup quota)"}
    # every usable core: each worker fuses the same read-only sample slice `reps` times into its own outputs
    nbw = max(64, min(nb, int(1.0 * rate / N)))                                 # ~1 s of single-thread work per repetition
    hw = {k: np.ascontiguousarray(v[:nbw]) for k, v in h.items()}
    budget = target_seconds * 0.75
    t_end = [0.0]

    def worker(_):
        n = 0
        while True:                                # at least one repetition, then until the shared deadline
            orc.fuse_pipeline_batch(hw["ts"], hw["pos"], hw["quat"], hw["gps"], hw["valid"], fit_rows=fit_rows)
            n += 1
            if time.perf_counter() >= t_end[0]:
                return n

    t0 = time.perf_counter()
    t_end[0] = t0 + budget
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        done = sum(ex.map(worker, range(cores)))
    dta = time.perf_counter() - t0
    res["all_cores"] = {"value": done * nbw * N / dta, "unit": "fused poses/s", "cores": cores,
                        "sample": f"{cores} threads (every core usable by this process) fused {done} blocks of {nbw} trajectories x {N} poses in {dta:.1f} s"}
    # the port is a C restatement; the reference it restates is single-threaded Python, two orders of magnitude slower (measured in the
    # build container, where the reference can be imported)
    res["reference_python"] = reference_python_timing()
    res["fit_rows"] = fit_rows
    return res


# ----------------------------------------------------------------------------------------------------------------------
# profile bookkeeping
# ----------------------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    """sha256 over the kernel sources: a committed PMC profile is only quoted while the kernels it measured are the ones built"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gps_optimize_slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(workload, kernel, grid_threads):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json, produced by
    tools/make_profiles.sh + tools/collect_profiles.py: separate --pmc runs of this same bench command; FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as read).  The profile records the hash of the kernel sources it was
    taken from; a profile of other sources is not quoted (traffic = null, with the reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, "no committed PMC profile"
    try:
        doc = json.load(open(files[-1]))
        name = os.path.basename(files[-1])
        if doc.get("kernel_source_hash") != kernel_source_hash():
            return None, f"{name} was taken from other kernel sources ({doc.get('kernel_source_hash')}); not quoted"
        key = f"{kernel} grid={grid_threads}"
        d = doc.get(workload, {})
        if key in d and "hbm_bytes" in d[key]:
            return d[key]["hbm_bytes"], name
        return None, f"{name} holds no entry for {key}"
    except Exception as e:                                   # a malformed profile must not take the bench down
        return None, f"profile unreadable: {e}"


SIMDS, CLOCK_GHZ, VALU_ISSUE_CYCLES = 1024, 2.4, 4      # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz peak; one wave issues a VALU instruction per 4 cycles


def profiled_issue(workload, kernel, grid_threads, kernel_ms):
    """The issue-side roofline of `kernel` from the committed PMC passes (same file and hash gate as profiled_traffic): the time the
    launch's VALU wave-instructions need at one instruction per 4 cycles on every one of the 1 024 SIMDs, as a fraction of the
    measured kernel time, plus what the waves were doing (VALU active / waiting).  None when no matching profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        if doc.get("kernel_source_hash") != kernel_source_hash():
            return None
        iss = doc.get(workload, {}).get(f"{kernel} grid={grid_threads}", {}).get("issue")
        if not iss or not iss.get("SQ_INSTS_VALU"):
            return None
        floor_ms = iss["SQ_INSTS_VALU"] * VALU_ISSUE_CYCLES / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
        out = {"valu_wave_instructions_per_launch": iss["SQ_INSTS_VALU"], "salu_instructions_per_launch": iss.get("SQ_INSTS_SALU"),
               "cycles_per_instruction": VALU_ISSUE_CYCLES, "simds": SIMDS, "clock_GHz": CLOCK_GHZ, "floor_ms": floor_ms, "frac": floor_ms / kernel_ms,
               "source": os.path.basename(files[-1])}
        if iss.get("SQ_WAVE_CYCLES"):
            out["valu_active_frac_of_wave_cycles"] = iss.get("SQ_ACTIVE_INST_VALU", 0.0) / iss["SQ_WAVE_CYCLES"]
            out["waiting_frac_of_wave_cycles"] = iss.get("SQ_WAIT_ANY", 0.0) / iss["SQ_WAVE_CYCLES"]
            out["waves"] = iss.get("SQ_WAVES")
        return out
    except Exception:
        return None


def profiled_stream_rate():
    """Bytes per second the memory system moved for K3 (apply_sim3_kernel: the same 24 / 32-byte-stride pose rows read and written, no
    arithmetic to speak of) in the committed profile of the auxiliary kernels: the practical ceiling of this access pattern under a mixed
    read + write stream (the C3 timing probes).  None when the profile is not there."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_aux_by_kernel_and_grid.csv")))
    if not files:
        return None
    try:
        for r in csv.DictReader(open(files[-1])):
            if r["kernel"] in ("apply_sim3_slab_kernel", "apply_sim3_kernel") and int(r["grid_threads"]) == 25600000:       # 1e8 poses, 112 B each
                return 1e8 * 112 / (float(r["avg_us"]) * 1e-6), os.path.basename(files[-1])
    except Exception:
        pass
    return None


# ----------------------------------------------------------------------------------------------------------------------
# C5-shaped leg: fuse a big per-GPU shard chunk by chunk, all-gather every chunk's poses while the next chunk is fused
# ----------------------------------------------------------------------------------------------------------------------
STALL_EXIT_CODE = 3                # a rank whose collect leg stalled has printed what it had and leaves with this code


def run_c5(torch, B, D, rank, world, dev, rehearsal, traj_per_gpu, chunk_traj, N, passes=1, stall_cb=None, stall_seconds=240.0,
           leg_budget_s=150.0, inject_stall_s=0.0, fit_rows="reference"):
    import ctypes as C
    import threading

    from gps_optimize_slam_amd import _lib
    L = _lib.load()
    # size to what the GPU has free (symmetric over ranks): inputs 89 B/pose + outputs 56 B/pose + 2 receive buffers
    free_b, _ = torch.cuda.mem_get_info(dev)
    per_traj = N * ALG_BYTES_PER_POSE + 64
    chunk_traj = max(1, min(chunk_traj, traj_per_gpu))
    recv_bytes = 2 * world * chunk_traj * N * 56 if world > 1 else 0
    fit = int((free_b * 0.92 - recv_bytes) // per_traj)
    T = max(chunk_traj, min(traj_per_gpu, fit))
    if world > 1:
        T = int(D.all_reduce(torch.tensor([T], dtype=torch.int64, device=dev), op=torch.distributed.ReduceOp.MIN).item())
    T -= T % chunk_traj
    nchunk = T // chunk_traj
    M, P = chunk_traj, chunk_traj * N
    info = {"trajectories_per_gpu": T, "poses_per_trajectory": N, "chunk_trajectories": M, "chunks": nchunk,
            "requested_trajectories_per_gpu": traj_per_gpu, "input_GB_per_gpu": T * N * 89 / 1e9, "output_GB_per_gpu": T * N * 56 / 1e9}
    s_comp, s_comm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(s_comp):
        ctx = B.context()
        ctx.set_sim3_rows(fit_rows, B.CONFIG)
        info["fit_rows"] = fit_rows
        batch = B.TrajectoryBatch(B.LAYOUT_TRAJ_MAJOR, T, N, dev)
        t0 = time.perf_counter()
        for lo in range(0, T, 65536):                                      # generated in place, this rank's ids
            n = min(65536, T - lo)
            _lib.check(L.gsf_synth_batch_dev(ctx.handle, B.LAYOUT_TRAJ_MAJOR, C.c_uint64(SEED), rank * T + lo, n, N, B._p(batch.ts[lo:]), B._p(batch.pos[lo:]),
                                             B._p(batch.quat[lo:]), B._p(batch.gps[lo:]), B._p(batch.valid[lo:]), None, None))
        s_comp.synchronize()
        info["generate_s"] = time.perf_counter() - t0
        out = torch.empty((T * N * 7,), dtype=torch.float64, device=dev)   # chunk k = [pos (M,N,3) | quat (M,N,4)] at k*P*7
        f = dict(dtype=torch.float64, device=dev)
        R, t, s = torch.empty((T, 9), **f), torch.empty((T, 3), **f), torch.empty((T,), **f)
        status = torch.empty((T,), dtype=torch.int32, device=dev)
        cfg = _lib.EkfConfig.from_config(B.CONFIG)

    def fuse(k):                                                           # on s_comp
        lo = k * M
        o = out[k * P * 7:]
        _lib.check(L.gsf_fuse_pipeline_batch_dev(ctx.handle, B.LAYOUT_TRAJ_MAJOR, B._p(batch.ts[lo:]), B._p(batch.pos[lo:]), B._p(batch.quat[lo:]),
                                                 B._p(batch.gps[lo:]), B._p(batch.valid[lo:]), C.byref(cfg), M, N, B._p(R[lo:]), B._p(t[lo:]), B._p(s[lo:]),
                                                 B._p(o), B._p(o[P * 3:]), B._p(status[lo:])))

    # ---- compute only
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(s_comp):
        fuse(0)
        s_comp.synchronize()
        D.barrier(dev)
        ev[0].record()
        for k in range(nchunk):
            fuse(k)
        ev[1].record()
        s_comp.synchronize()
    comp_ms = D.max_over_ranks(ev[0].elapsed_time(ev[1]), dev)
    info["compute_only"] = {"ms": comp_ms, "poses_per_s": world * T * N / (comp_ms * 1e-3), "alg_GBps_per_gpu": T * N * ALG_BYTES_PER_POSE / comp_ms / 1e6,
                            "hbm_frac": T * N * ALG_BYTES_PER_POSE / comp_ms / 1e6 / HBM_PEAK_GBS}
    local_sum = out.view(torch.int64).sum()                                # integer checksum of this rank's fused poses (exact, order-free)
    # accuracy on device: the reference's error metric (Q15) of a sample of the fused tracks against the synthetic GNSS
    ns = min(256, M)
    st, _ = B.eval_errors_batch(batch.ts[:ns], out[:P * 3].view(M, N, 3)[:ns].contiguous(), batch.gps[:ns], batch.valid[:ns], 5.0)
    info["q15_rmse_vs_gnss_m_sample_mean"] = float(st[:, 3].nanmean().item())
    info["status_counts"] = {"had_outage": int((status & 1).ne(0).sum()), "rts_applied": int((status & 2).ne(0).sum()),
                             "sharp_turn": int((status & 4).ne(0).sum()), "ended_in_outage": int((status & 8).ne(0).sum()),
                             "fit_none": int(((status >> 8) & 1).ne(0).sum()), "fit_fallbacks": int(((status >> 8) & 16).ne(0).sum())}
    if world == 1:
        info["checksum_int64"] = int(local_sum.item())
        del batch, out
        torch.cuda.empty_cache()
        return info

    # ---- compute + gather: chunk k's poses are collected on s_comm while chunk k+1 is fused on s_comp
    recv = [torch.empty((world * P * 7,), dtype=torch.float64, device=dev) for _ in range(2)]
    total_sum = torch.zeros((), dtype=torch.int64, device=dev)
    exp_all = int(D.all_reduce(local_sum.clone().reshape(1)).item())      # wrapping int64 sum of every rank's checksum
    recv_per_rank = (world - 1) * T * N * 56
    legs = [("torch_all_gather", None)]
    if not rehearsal:
        legs += [("gsf_ncclAllGather", 0), ("gsf_direct_sendrecv", 1)]
    collector = None
    info["collect"] = {}
    info["backend"] = torch.distributed.get_backend()
    # a-priori budget: a leg moves recv_per_rank bytes into every GPU; at a pessimistic 30 GB/s per xGMI link (a fifth of the 153 GB/s
    # link rate) that is the time below.  Passes are cut so that one leg stays inside leg_budget_s; legs that would start after the
    # budget of all three is spent are skipped and say so.
    est_pass_s = recv_per_rank / (max(1, world - 1) * 30e9) + 2.0 * info["compute_only"]["ms"] * 1e-3
    passes = max(1, min(passes, int(leg_budget_s / max(est_pass_s, 1e-3))))
    info["time_budget"] = {"estimated_pass_s_at_30GBps_per_link": est_pass_s, "passes": passes, "leg_budget_s": leg_budget_s,
                           "watchdog_s": stall_seconds, "legs": [n for n, _ in legs]}
    t_legs = time.perf_counter()

    # Every leg runs under a watchdog: the library's own communicator and its direct exchange have never met a second GPU before an
    # 8-GPU node runs this (one GPU per build box), and a stalled collective cannot be cancelled from Python.  If a leg does not finish
    # within stall_seconds, every rank reports what it has (rank 0 prints the JSON line through stall_cb) and leaves with a NON-ZERO
    # exit code, so that the parent and the driver see the stall.
    def run_leg(name, mode):
        nonlocal collector
        def on_stall():
            info["stalled_leg"] = name
            info["own_communicator_error" if mode is not None else "collect_error"] = \
                f"watchdog: leg {name} did not finish within {stall_seconds:.0f} s"
            if stall_cb is not None:
                stall_cb(info)
            sys.stdout.flush()
            os._exit(STALL_EXIT_CODE)
        watchdog = threading.Timer(stall_seconds, on_stall)
        watchdog.daemon = True
        watchdog.start()
        try:
            if inject_stall_s > 0.0:                                       # test hook: a leg that hangs
                time.sleep(inject_stall_s)
            if mode is not None and collector is None:
                collector = D.PoseCollector(dev, s_comm)
                info["rccl_version"] = collector.rccl_version
            def gather(k):                                                 # on s_comm
                send = out[k * P * 7:(k + 1) * P * 7]
                if mode is None:
                    D.all_gather_flat(recv[k & 1], send)
                else:
                    collector.allgather(send, recv[k & 1], mode=mode, chunk_count=0)
                total_sum.add_(recv[k & 1].view(torch.int64).sum())        # the sink: checksum of the gathered chunk
            total_sum.zero_()
            with torch.cuda.stream(s_comm):
                gather(0)                                                  # communicator warm-up (outside the timed region)
            s_comm.synchronize()
            total_sum.zero_()
            D.barrier(dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(passes):
                s_comp.wait_stream(s_comm)                                 # a pass rewrites the rows the previous pass's gathers read
                for k in range(nchunk):
                    with torch.cuda.stream(s_comp):
                        fuse(k)
                        e = torch.cuda.Event()
                        e.record()
                    with torch.cuda.stream(s_comm):
                        s_comm.wait_event(e)
                        gather(k)
            torch.cuda.synchronize(dev)
            D.barrier(dev)
            el = D.max_over_ranks(time.perf_counter() - t0, dev) / passes
            ok = ((int(total_sum.item()) - exp_all * passes) % (1 << 64)) == 0
            return {"pass_ms": el * 1e3, "poses_per_s": world * T * N / el, "recv_GB_per_rank": recv_per_rank / 1e9,
                    "recv_GBps_per_rank": recv_per_rank / el / 1e9, "per_link_GBps": recv_per_rank / el / 1e9 / (world - 1),
                    "link_frac_of_153": recv_per_rank / el / 1e9 / (world - 1) / XGMI_LINK_GBS, "passes": passes,
                    "gathered_checksum_equals_sum_of_rank_checksums": bool(ok)}
        finally:
            watchdog.cancel()

    for name, mode in legs:
        if time.perf_counter() - t_legs > leg_budget_s * len(legs):
            info["collect"][name] = {"skipped": "time budget of the collect legs spent"}
            continue
        try:                                                               # one failed leg must not drop the others
            info["collect"][name] = run_leg(name, mode)
        except Exception as e:                                             # symmetric over ranks: every rank resolves the same library
            info["collect"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if mode is not None:
                info["own_communicator_error"] = info["collect"][name]["error"]
    if collector is not None:
        try:
            collector.close()
        except Exception as e:
            info["collector_close_error"] = str(e)[:200]
    del batch, out, recv
    torch.cuda.empty_cache()
    return info


def spread_sample(n_traj, n=64):
    """indices of a sample spread over a batch: its first, middle and last wave-sized slices (the gate must not look at the head only)"""
    import numpy as np
    if n_traj <= n:
        return np.arange(n_traj)
    a, b = n // 3, n // 3
    c = n - a - b
    mid = n_traj // 2
    return np.unique(np.concatenate([np.arange(a), np.arange(mid - b // 2, mid - b // 2 + b), np.arange(n_traj - c, n_traj)]))


def oracle_gate(np, torch, B, bt, pos_out, status, idx, fit_rows, R=None):
    """The TIMED outputs of trajectories `idx` of a trajectory-major batch against oracle.fuse_pipeline_batch under the same row rule:
    ATE RMSE, max |dp|, status words (the informational SVD-fallback bit aside)."""
    from oracle import oracle as orc
    ix = torch.as_tensor(idx, device=bt.ts.device)
    hh = {k: getattr(bt, k).index_select(0, ix).cpu().numpy() for k in ("ts", "pos", "quat", "gps", "valid")}
    po, qo, sto, Ro, _, _ = orc.fuse_pipeline_batch(hh["ts"], hh["pos"], hh["quat"], hh["gps"], hh["valid"], fit_rows=fit_rows)
    pg, sg = pos_out.index_select(0, ix).cpu().numpy(), status.index_select(0, ix).cpu().numpy()
    fin = np.isfinite(po).all(axis=(1, 2))
    res = {"sample": f"{len(idx)} trajectories spread over the batch (first / middle / last slice)", "fit_rows": fit_rows,
           "ate_rmse_vs_cpu_ref_m": float(np.sqrt(np.mean(np.sum((pg[fin] - po[fin]) ** 2, axis=2)))) if fin.any() else None,
           "max_abs_pos_err_m": float(np.abs(pg[fin] - po[fin]).max()) if fin.any() else None,
           "status_words_equal": bool(((sg & ~(16 << 8)) == sto).all()) and bool((np.isfinite(pg).all(axis=(1, 2)) == fin).all())}
    if R is not None:
        res["max_abs_sim3_R_err"] = float(np.nanmax(np.abs(R.index_select(0, ix).cpu().numpy() - Ro)))
    return res


# ----------------------------------------------------------------------------------------------------------------------
def setup(args):
    """Process-level set-up of one rank: imports, device, process group, the library context with the run's options, the fields every
    line carries.  Returns a namespace the per-workload functions share."""
    import types

    import numpy as np
    import torch

    from gps_optimize_slam_amd import _lib
    from gps_optimize_slam_amd import batch as B
    from gps_optimize_slam_amd import distributed as D

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible -- the fusion path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}", file=sys.stderr)
        sys.exit(2)
    ngpu = torch.cuda.device_count()
    rehearsal = world_env > ngpu                    # fewer GPUs than ranks (one-GPU box): gloo, ranks share the GPUs, tiny sizes
    local = int(os.environ.get("LOCAL_RANK", "0")) % ngpu
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ["LOCAL_RANK"] = str(local)
    rank, world, _ = D.init_from_env()
    backend = torch.distributed.get_backend() if world > 1 else None
    if world > 1 and not rehearsal and backend != "nccl":
        # one GPU per rank is there: the collect must be RCCL over xGMI, never a CPU-staged gloo rehearsal
        print(f"bench.py: {world} ranks on {ngpu} GPUs but the process group runs on '{backend}', not nccl (RCCL)", file=sys.stderr)
        sys.exit(2)
    wl = WORKLOADS[args.workload]
    Bn, N = wl["B"], wl["N"]
    if args.traj_per_gpu:
        Bn = args.traj_per_gpu
    if args.poses:
        N = args.poses
    if rehearsal and args.workload != "c2":
        Bn = min(Bn, 8192)
    steps = args.steps if args.steps is not None else {"c2": 2000, "c3": 10, "c5": 1}[args.workload]
    warmup = args.warmup if args.warmup is not None else {"c2": 50, "c3": 2, "c5": 0}[args.workload]
    ctx = B.context()
    for kv in args.set_option:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ctx.set_sim3_rows(args.fit_rows, B.CONFIG)                             # rows of the pipeline's fit on every raw launch below
    L = _lib.load()
    base = {"metric": "fused poses/sec (whole node)", "unit": "fused poses/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "backend": backend if world > 1 else "single process (no collective)", "gpus_visible": ngpu}
    # which build of the library this is: scheduler / -ffp-contract mode / compiler of every wave-level translation unit (gsf_version()),
    # and the Makefile's warning file if a translation unit took its fall-back path
    warn_path = os.path.join(ROOT, "gps_optimize_slam_amd", "BUILD_WARNINGS.txt")
    base["build"] = {"gsf_version": L.gsf_version().decode(), "library": os.path.basename(_lib.library_path()),
                     "build_warnings": open(warn_path).read().strip() if os.path.exists(warn_path) else None}
    if base["build"]["build_warnings"] and rank == 0:
        print("bench.py: BUILD WARNING -- " + base["build"]["build_warnings"], file=sys.stderr)
    x = types.SimpleNamespace(args=args, np=np, torch=torch, _lib=_lib, B=B, D=D, L=L, ctx=ctx, dev=dev, rank=rank, world=world, ngpu=ngpu,
                              rehearsal=rehearsal, wl=wl, Bn=Bn, N=N, steps=steps, warmup=warmup, base=base, partial={}, run_deadline=None)
    if world > 1:
        import threading

        def on_deadline():                         # what rank 0 prints if the run-wide deadline fires
            if rank == 0:
                print(json.dumps(dict(base, **x.partial, deadline_exceeded=f"{args.deadline_s:.0f} s", value=x.partial.get("value"))), flush=True)
            os._exit(STALL_EXIT_CODE)
        x.run_deadline = threading.Timer(args.deadline_s, on_deadline)
        x.run_deadline.daemon = True
        x.run_deadline.start()
    return x


def finish(x):
    if x.world > 1:
        x.run_deadline.cancel()
        x.torch.distributed.destroy_process_group()


def headline_c5(x):
    """--workload c5: the per-GPU shard of BASELINE configs[4] as the headline (chunked fuse + overlapped all-gather, run_c5)."""
    args, world, N, wl = x.args, x.world, x.N, x.wl
    chunk = args.chunk_traj or (32768 if not x.rehearsal else 1024)

    def emit_c5(info):
        if x.rank != 0:
            return
        done = [c for c in info.get("collect", {}).values() if "pass_ms" in c]
        best = min(done, key=lambda c: c["pass_ms"]) if world > 1 and done else None
        pass_ms = best["pass_ms"] if best else info["compute_only"]["ms"]
        T = info["trajectories_per_gpu"]
        co = info["compute_only"]
        result = dict(x.base, value=world * T * N / (pass_ms * 1e-3), ms_per_step=pass_ms,
                      config={"workload": wl["name"], "trajectories_per_gpu": T, "poses_per_trajectory": N, "chunk_trajectories": info["chunk_trajectories"],
                              "layout": "trajectory-major AoS (wave-per-trajectory scans)", "step": "one pass over the shard: per chunk fused pipeline"
                              + (" + all-gather of the chunk's poses (second stream, overlapped)" if world > 1 else ""),
                              "parallelism": f"trajectory-sharded x{world}" + (" (gloo rehearsal on shared GPUs)" if x.rehearsal else "")},
                      roofline={"bound": "hbm", "kernel": "ekf_wave_big_kernel<true, 1>", "achieved": co["alg_GBps_per_gpu"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": co["hbm_frac"], "traffic": None, "traffic_note": "no PMC pass at this size", "alg_bytes_per_launch": info["chunk_trajectories"] * N * ALG_BYTES_PER_POSE,
                                "kernel_ms": co["ms"] / info["chunks"]},
                      c5=info)
        print(json.dumps(result), flush=True)
    info = run_c5(x.torch, x.B, x.D, x.rank, world, x.dev, x.rehearsal, x.Bn, chunk, N, passes=max(1, x.steps), stall_cb=emit_c5,
                  stall_seconds=args.stall_seconds, inject_stall_s=args.inject_stall, fit_rows=args.fit_rows)
    emit_c5(info)


def timed_steps(x):
    """The timed region of the c2 / c3 headline: this rank's synthetic shard, W warm-up launches, then EXACTLY K launches of the step (one
    hipGraph replay of them where the capture is allowed) + for N > 1 the one all-gather of the fused poses, bracketed by barrier + synchronize
    on both sides.  Fills x with the batch, the outputs, the launcher, the elapsed times."""
    import ctypes as C
    args, torch, B, D, L, _lib, dev, world, Bn, N, steps = x.args, x.torch, x.B, x.D, x.L, x._lib, x.dev, x.world, x.Bn, x.N, x.steps
    x.time_major = args.layout == "time"
    batch = B.TrajectoryBatch.synthetic(Bn, N, layout=B.LAYOUT_TIME_MAJOR if x.time_major else B.LAYOUT_TRAJ_MAJOR, seed=SEED, traj0=x.rank * Bn)   # this rank's shard: ids [rank*B, (rank+1)*B)
    out = B.FusedPoses(batch.layout, Bn, N, dev)
    f = dict(dtype=torch.float64, device=dev)
    cfg = _lib.EkfConfig.from_config(B.CONFIG)
    h, p = x.ctx.handle, B._p
    x.R = None
    if args.kernel == "pipeline":
        R, t, s = torch.empty((Bn, 9), **f), torch.empty((Bn, 3), **f), torch.empty((Bn,), **f)
        x.R = R

        def launch(h=h):
            _lib.check(L.gsf_fuse_pipeline_batch_dev(h, batch.layout, p(batch.ts), p(batch.pos), p(batch.quat), p(batch.gps), p(batch.valid),
                                                     C.byref(cfg), Bn, N, p(R), p(t), p(s), p(out.pos), p(out.quat), p(out.status)))
    else:
        def launch(h=h):
            _lib.check(L.gsf_ekf_fuse_batch_dev(h, batch.layout, p(batch.ts), p(batch.pos), p(batch.quat), p(batch.gps), p(batch.valid), p(batch.init_pos),
                                                p(batch.init_quat), C.byref(cfg), Bn, N, p(out.pos), p(out.quat), p(out.status)))
    gathered = torch.empty((world * out.buf.numel(),), **f) if world > 1 else None

    def collect():                              # the job's ONE collect: fused poses of every shard to every GPU (RCCL over xGMI)
        D.all_gather_flat(gathered, out.buf)

    for _ in range(x.warmup):
        launch()
    if world > 1:
        collect()                               # communicator set-up stays outside the timed region
    # The K steps are one launch each of the same kernel: captured once into a hipGraph (on a side stream, with a context of the library
    # bound to that stream) and replayed in the timed region, so that a short run does not time the host's enqueue gaps.  Falls back
    # to eager launches if the capture is refused.
    graph, launch_mode = None, "eager launches"
    if not args.no_graph:
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                hc = B.context().handle                                   # the library launches on ITS context's stream: one bound to `side`
                for kv in args.set_option:
                    k_, v_ = kv.split("=")
                    B.context().set_option(k_, int(v_))
                B.context().set_sim3_rows(args.fit_rows, B.CONFIG)
                launch(hc)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # (thread-local error mode: another thread of the process -- torch's RCCL watchdog polls events -- must not invalidate the capture)
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for _ in range(steps):
                    launch(hc)
            torch.cuda.synchronize()
            g.replay()                                                    # one untimed replay (graph upload)
            torch.cuda.synchronize()
            graph, launch_mode = g, f"hipGraph: the {steps} launches captured once, one replay timed"
        except Exception as e:
            launch_mode = f"eager launches (graph capture refused: {type(e).__name__}: {e})"[:200]
            torch.cuda.synchronize()
    # Timed region = the K fusion steps of this rank's shard (no collective on the data path: trajectories are independent) +
    # the single all-gather that collects the fused poses (north star / SURVEY 8e), bracketed by barrier + synchronize.
    # HIP events on torch's current stream == the stream the kernels are launched on (B.context()).  One pair around the K
    # launches: at C2 a launch is ~20 us, so per-launch event pairs would make the loop host-bound and pad the gaps.
    def region():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()                          # (on an idle stream, right before the clock starts: the marker is for the kernel span, not part of the job)
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                launch()
        ev[1].record()
        if world > 1:
            collect()
            ev[2].record()
        else:
            ev[2] = ev[1]
        torch.cuda.synchronize(); D.barrier(dev)
        return time.perf_counter() - t0, ev
    # The driver times K = 20 steps: a region of 0.4 ms, in which the HOST side of the bracket (event records, the graph launch, the wake-up
    # from the synchronize) is 20 us when those code paths are hot and 35-100 us when the thread comes out of the capture / a barrier /
    # an idle spell (tools/experiments/k20_breakdown.py).  The region is therefore rehearsed, untimed, immediately before it is timed:
    # the same calls in the same order.  Not a change of the timed work -- the timed region below is exactly K steps.  (A spin-waiting
    # synchronize, hipDeviceScheduleSpin, changes nothing: tools/experiments/k20_overhead.sh.)
    # (only where it matters: at K > 200 the bracket is under 1 % of the region, and under rocprofv3 thousands of extra launches make the
    # profiler itself the bottleneck of the later ones)
    rehearsals = 2 if steps <= 200 else 0
    for _ in range(rehearsals):
        region()
    if rehearsals:
        launch_mode += f"; {rehearsals} untimed rehearsals of the region right before it"
    D.barrier(dev); torch.cuda.synchronize()
    el, (ev0, ev1, ev2) = region()
    x.elapsed = D.max_over_ranks(el, dev)
    x.kern_ms = ev0.elapsed_time(ev1) / steps     # back-to-back launches of the one kernel: span / K = average launch duration
    x.collect_ms = ev1.elapsed_time(ev2) if world > 1 else 0.0
    x.batch, x.out, x.launch, x.collect, x.gathered, x.launch_mode, x.f = batch, out, launch, collect, gathered, launch_mode, f


def launched_kernel(x):
    """Which build the library launches for this shape (gsf_ekf_wave.hip: launch_ekf_wave; all produce identical bits) -> its name and grid as
    rocprofv3 reports them, and the section of the committed PMC profile that holds it."""
    args, Bn, N = x.args, x.Bn, x.N
    pipe = args.kernel == "pipeline"
    blk = any(kv.replace(" ", "") == "block_kernel=1" for kv in args.set_option) and 64 < N <= 1024   # (under --fit-rows reference its launcher marks the rows with sim3_rows_kernel first)
    lane_route = x.time_major and Bn >= 32768 and not any(kv.replace(" ", "").startswith("lane_min_traj=") for kv in args.set_option)
    if lane_route:
        # lane per trajectory (gsf_ekf.hip): <LAYOUT, prefetch depth, waves per SIMD>
        kernel_name, grid_threads = ("fuse_pipeline_kernel<1, 2, 2>" if pipe else "ekf_fuse_kernel<1, 2, 2>"), ((Bn + 63) // 64) * 64
    elif blk:
        # the workgroup-per-trajectory kernel (opt-in): <PIPELINE, AXMODE, max threads, waves per SIMD, inlined cold blocks>
        Wv = (N + 63) // 64
        shape = "320, 5, true" if Wv <= 5 else ("512, 4, true" if Wv <= 8 else "1024, 4, false")
        kernel_name, grid_threads = "ekf_block_kernel<%s, 1, %s>" % ("true" if pipe else "false", shape), Bn * Wv * 64
    elif pipe and 64 < N <= 640 and Bn <= 256:
        kernel_name, grid_threads = "ekf_wave_duo_kernel<true, 1>", Bn * 128                     # a helper wave per trajectory
    else:
        # <PIPELINE, SMALLBATCH, AXMODE>: AXMODE 1 = x and y share their noise figures, z does not (the default CONFIG, compiled-in scans)
        # small batches: ekf_wave_kernel<PIPELINE, true, 1> (gsf_ekf_wave.hip); more than two waves per SIMD: ekf_wave_big_kernel<PIPELINE, 1>
        kernel_name = ("ekf_wave_kernel<%s, true, 1>" if Bn <= 2048 else "ekf_wave_big_kernel<%s, 1>") % ("true" if pipe else "false")
        grid_threads = Bn * 64
    wl_key = args.workload + ("" if pipe else "ekf") if not blk else args.workload + "block" + ("" if pipe else "ekf")   # section of the PMC profile
    if pipe and args.fit_rows == "all":
        wl_key += "all"                                                     # the same kernel fitting every valid row: its own section of the profile
    if lane_route:
        wl_key = args.workload + "lane" + ("" if pipe else "ekf")
    if args.poses:
        wl_key += f"_n{N}"                                                  # not a BASELINE configuration: never matches a committed section
    elif args.traj_per_gpu and args.workload == "c3" and Bn == 32768:
        wl_key = "c5chunk" + ("" if pipe else "ekf")                       # one chunk of the C5 shard: the grid the 38 launches of a pass have
    x.lane_route = lane_route
    return kernel_name, grid_threads, wl_key


def headline_result(x):
    """The contract line of the c2 / c3 headline from the timed region: value, config, roofline (algorithmic bytes per launch / the kernel's
    live average duration; counter traffic and the issue-side floor from the committed PMC profile of the same kernel sources)."""
    args, world, Bn, N, steps = x.args, x.world, x.Bn, x.N, x.steps
    kern_ms = x.kern_ms
    x.poses_per_step = world * Bn * N
    x.alg_bytes = alg_bytes = Bn * N * ALG_BYTES_PER_POSE
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    kernel_name, grid_threads, wl_key = launched_kernel(x)
    traffic, traffic_src = profiled_traffic(wl_key, kernel_name, grid_threads)
    result = dict(x.base, value=x.poses_per_step * steps / x.elapsed, ms_per_step=x.elapsed / steps * 1e3,
                  config={"workload": x.wl["name"], "trajectories_per_gpu": Bn, "poses_per_trajectory": N,
                          "layout": ("time-major SoA (" + ("lane per trajectory" if x.lane_route else "transposed, then wave per trajectory") + ")") if x.time_major
                          else "trajectory-major AoS (wave-per-trajectory scans)",
                          "step": args.kernel, "launch_mode": x.launch_mode, "parallelism": f"trajectory-sharded x{world}" + (", one RCCL all-gather of the fused poses after the K steps (inside the timed region)" if world > 1 else "")
                          + (" (gloo rehearsal on shared GPUs)" if x.rehearsal else "")},
                  roofline={"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                            "kernel_source_hash": kernel_source_hash()})
    # what the counters say bounds this kernel: the VALU issue floor next to the HBM figure (profile of the same sources, else absent)
    valu = profiled_issue(wl_key, kernel_name, grid_threads, kern_ms)
    if valu is not None:
        result["roofline"]["valu"] = valu
        if valu["frac"] > result["roofline"]["frac"]:
            result["roofline"]["bound"] = "valu"
            result["roofline"]["bound_note"] = ("counters: the launch's VALU wave-instructions at one per 4 cycles on all 1 024 SIMDs take a larger share of the "
                                                "kernel time than its algorithmic bytes at 8 TB/s; achieved / peak / frac stay the HBM figures")
    # many waves per SIMD (C3-sized batches): the counter traffic at the rate K3 reaches with the same rows is the first wall there -- the
    # timing probes of DESIGN.md section 5 (either stream alone hides behind the arithmetic, both together do not)
    stream = profiled_stream_rate()
    if stream is not None and traffic and Bn >= 2 * SIMDS:
        floor_ms = traffic / stream[0] * 1e3
        result["roofline"]["stream"] = {"counter_bytes_per_launch": traffic, "rate_of_K3_same_rows_GBps": stream[0] / 1e9, "floor_ms": floor_ms,
                                        "frac": floor_ms / kern_ms, "source": stream[1]}
        if floor_ms / kern_ms > 1.0:
            result["roofline"]["stream"]["note"] = ("above 1: part of this launch's counted traffic (the fit pass's second read) is served by the Infinity Cache, "
                                                    "which K3's streaming rate does not describe")
        elif floor_ms / kern_ms > max(result["roofline"]["frac"], (valu or {}).get("frac", 0.0)):
            result["roofline"]["bound"] = "hbm"
            result["roofline"]["bound_note"] = ("the bytes the counters saw, at the rate apply_sim3_kernel moves the same 24 / 32-byte-stride rows (mixed read + write), "
                                                "take a larger share of the kernel time than the VALU issue floor; achieved / peak / frac stay the algorithmic bytes at 8 TB/s")
    if args.kernel == "pipeline":
        x.torch.cuda.synchronize()
        # how many tracks of the timed batch took the Jacobi-SVD fallback of the fit (status bit GSF_SIM3_FLAG_SVD_FALLBACK << 8): one such
        # track costs its whole launch the SVD's time again, so the rate belongs next to the kernel time
        result["fit_fallbacks"] = int(((x.out.status >> 8) & 16).ne(0).sum().item())
        result["fit_none"] = int(((x.out.status >> 8) & 1).ne(0).sum().item())
        # the fused pipeline MOVES 194 B/pose (the fit pass re-reads pos/gps/valid: L2/Infinity-Cache hits at C2, real HBM traffic at C3)
        moved = Bn * N * (ALG_BYTES_PER_POSE + FIT_REREAD_BYTES_PER_POSE)
        result["roofline"]["moved_bytes_per_launch_incl_fit_pass"] = moved
        result["roofline"]["frac_of_peak_on_moved_bytes"] = moved / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return result


def collect_report(x):
    """N > 1: the job's one collect on its own (compute-only and compute+gather separately), the rate with a gather after every
    step, and that every gathered block equals its rank's own checksum."""
    torch, D, dev, world, Bn, N, steps = x.torch, x.D, x.dev, x.world, x.Bn, x.N, x.steps
    recv = (world - 1) * Bn * N * 56
    k2 = max(1, min(steps, 20))
    D.barrier(dev); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(k2):
        x.launch(); x.collect()
    torch.cuda.synchronize(); D.barrier(dev)
    el2 = D.max_over_ranks(time.perf_counter() - t1, dev)
    # gathered == the unsharded result?  rank r's block must equal what rank r computed (checksums travel by all_gather)
    mine = x.out.buf.view(torch.int64).sum().reshape(1)
    allsums = torch.empty((world,), dtype=torch.int64, device=dev)
    D.all_gather_flat(allsums, mine)
    blocks = x.gathered.view(world, -1).view(torch.int64).sum(dim=1)
    cms = x.collect_ms
    return {"allgather_ms": cms, "recv_bytes_per_rank": recv, "recv_GBps_per_rank": recv / (cms * 1e-3) / 1e9 if cms > 0 else None,
            "per_link_GBps": recv / (cms * 1e-3) / 1e9 / (world - 1) if cms > 0 else None,
            "compute_only_poses_per_s": world * Bn * N / (x.kern_ms * 1e-3),
            "collect_every_step": {"steps": k2, "ms_per_step": el2 / k2 * 1e3, "poses_per_s": x.poses_per_step * k2 / el2},
            "gathered_blocks_equal_rank_checksums": bool(torch.equal(blocks, allsums)),
            "backend": torch.distributed.get_backend(),
            "share_of_the_timed_region": cms / (cms + x.kern_ms * steps) if cms > 0 else None,
            "how_to_read_value": (f"the timed region is the {steps} steps ({x.kern_ms * steps:.3f} ms of kernels) plus this ONE all-gather ({cms:.3f} ms, "
                                  f"{recv / 1e6:.1f} MB received per rank): at small K the N > 1 `value` is the collect's time, not the kernels' -- "
                                  "compute_only_poses_per_s is the kernels alone, collect_every_step the rate with a gather after every step")}


def accuracy_gate(x, result):
    """The accuracy gate on THIS run (rank 0): the timed `out` buffers of a sample SPREAD over the timed batch against the CPU oracle on the same
    inputs -- ATE RMSE, max |dp|, status words -- and the reference's own error metric (Q15) of both results against the synthetic GNSS."""
    from oracle import oracle as orc
    args, np, torch, B, L, _lib, dev, Bn, N, out, f = x.args, x.np, x.torch, x.B, x.L, x._lib, x.dev, x.Bn, x.N, x.out, x.f
    p = B._p
    idx = spread_sample(Bn, 64)
    nb = len(idx)
    torch.cuda.synchronize()
    ix = torch.as_tensor(idx, device=dev)
    gb_ = x.batch.to_layout(B.LAYOUT_TRAJ_MAJOR)                          # (the gate reads trajectory-major views; a no-op for the default layout)
    if x.time_major:
        opos, oquat = torch.empty((Bn, N, 3), **f), torch.empty((Bn, N, 4), **f)
        _lib.check(L.gsf_transpose_to_traj_major_dev(x.ctx.handle, p(out.pos), p(opos), Bn, N, 3, 8))
        _lib.check(L.gsf_transpose_to_traj_major_dev(x.ctx.handle, p(out.quat), p(oquat), Bn, N, 4, 8))
        torch.cuda.synchronize()
    else:
        opos, oquat = out.pos, out.quat
    hh = {k: getattr(gb_, k).index_select(0, ix).cpu().numpy() for k in ("ts", "pos", "quat", "gps", "valid", "init_pos", "init_quat")}
    pg, qg, sg = opos.index_select(0, ix).cpu().numpy(), oquat.index_select(0, ix).cpu().numpy(), out.status.index_select(0, ix).cpu().numpy()
    result["gate_sample"] = f"{nb} trajectories of the timed batch: ids {int(idx[0])}..{int(idx[nb // 3 - 1])}, {int(idx[nb // 3])}..{int(idx[2 * (nb // 3) - 1])}, {int(idx[2 * (nb // 3)])}..{int(idx[-1])}"
    if args.kernel == "pipeline":
        po, qo, sto, Ro, to, so = orc.fuse_pipeline_batch(hh["ts"], hh["pos"], hh["quat"], hh["gps"], hh["valid"], fit_rows=args.fit_rows)
        rows_txt = ("the rows main_process_gui hands to its fit, EKFGPSSLAM.py:973-998" if args.fit_rows == "reference" else "every row with valid finite GNSS")
        result["gated"] = f"timed fused-pipeline outputs (Umeyama on {rows_txt} -> Sim3 of pose 0 -> EKF+RTS) vs oracle.fuse_pipeline_batch(fit_rows='{args.fit_rows}')"
        result["max_abs_sim3_R_err"] = float(np.nanmax(np.abs(x.R.index_select(0, ix).cpu().numpy() - Ro)))
        result["fit_status_words_equal"] = bool((((sg >> 8) & ~16) == (sto >> 8)).all())
    else:
        po, qo, sto = orc.fuse_batch(hh["ts"], hh["pos"], hh["quat"], hh["gps"], hh["valid"], hh["init_pos"], hh["init_quat"])
        result["gated"] = "timed K4 outputs vs oracle.fuse_batch"
    fin = np.isfinite(po).all(axis=(1, 2))
    result["ate_rmse_vs_cpu_ref_m"] = float(np.sqrt(np.mean(np.sum((pg[fin] - po[fin]) ** 2, axis=2))))
    result["max_abs_pos_err_m"] = float(np.abs(pg[fin] - po[fin]).max())
    result["max_abs_quat_err"] = float(np.abs(qg[fin] - qo[fin]).max())
    result["status_bits_equal"] = bool(((sg & 0xff) == (sto & 0xff)).all()) and bool((np.isfinite(pg).all(axis=(1, 2)) == fin).all())
    # the reference's own error metric (Q15, EKFGPSSLAM.py:1013-1033) of both results against the synthetic GNSS
    stats, _ = B.eval_errors_batch(gb_.ts.index_select(0, ix), opos.index_select(0, ix).contiguous(), gb_.gps.index_select(0, ix),
                                   gb_.valid.index_select(0, ix), 5.0)
    del gb_, opos, oquat
    g_rmse = stats[:, 3].cpu().numpy()
    c_rmse = np.array([orc.evaluate_trajectory_errors(hh["ts"][b], po[b], hh["gps"][b], hh["valid"][b])["rmse"] for b in range(nb)])
    both = np.isfinite(g_rmse) & np.isfinite(c_rmse)
    result["ref_style_error_q15"] = {"gpu_rmse_m_mean": float(g_rmse[both].mean()), "cpu_rmse_m_mean": float(c_rmse[both].mean()),
                                     "max_abs_diff_m": float(np.abs(g_rmse[both] - c_rmse[both]).max()), "trajectories": int(both.sum()),
                                     "definition": "min distance to any candidate fix after the first 5 s, RMSE per trajectory (EKFGPSSLAM.py:1013-1033)"}


def other_rows_report(x):
    """Both definitions of the fit's rows in one line: the headline is --fit-rows; here the other one, same batch, same box."""
    args, torch, B, world, Bn, N, out, R = x.args, x.torch, x.B, x.world, x.Bn, x.N, x.out, x.R
    other = "all" if args.fit_rows == "reference" else "reference"
    torch.cuda.synchronize()
    st_head = out.status.clone(); R_head = R.clone()
    x.ctx.set_sim3_rows(other, B.CONFIG)
    k3 = max(20, min(x.steps, 100))                                       # few launches: the kernel trace of this command averages over them too
    for _ in range(10):
        x.launch()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ea.record()
    for _ in range(k3):
        x.launch()
    eb.record(); torch.cuda.synchronize()
    ms_other = ea.elapsed_time(eb) / k3
    changed = int((R_head - R).abs().amax(dim=1).gt(0).sum().item())
    rows_bits = (st_head >> 8) if args.fit_rows == "reference" else (out.status >> 8)
    rep = {"headline": args.fit_rows,
           "definitions": {"reference": "the rows main_process_gui hands to its fit (EKFGPSSLAM.py:973-998): first gap-free segment of the valid rows, "
                                        "<= max_initial_duration, fall-backs :984-986 / :993-995",
                           "all": "every row with valid finite GNSS (the operator SURVEY 8(b)/(d) defined; rounds 1-3)"},
           "other": {"fit_rows": other, "kernel_ms": ms_other, "value": world * Bn * N / (ms_other * 1e-3), "launch_mode": f"{k3} eager launches",
                     "hbm_frac": x.alg_bytes / (ms_other * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "tracks_whose_fit_differs_between_the_two": changed, "tracks": Bn,
           "reference_rule_branches": {"first_segment_too_short_all_rows": int(((rows_bits & 64) != 0).sum().item()),
                                       "duration_limit_dropped_whole_segment": int(((rows_bits & 128) != 0).sum().item()),
                                       "too_few_rows_value_error": int(((rows_bits & 32) != 0).sum().item())}}
    x.ctx.set_sim3_rows(args.fit_rows, B.CONFIG)
    x.launch(); torch.cuda.synchronize()                                  # leave the headline definition's outputs in `out` for the extras
    return rep


def c5_leg(x, result):
    """N > 1: the C5-shaped leg next to the headline (BASELINE configs[4]): the per-GPU shard of 10M x 1k over 8 GPUs, chunked fuse + overlapped
    all-gather."""
    args = x.args
    tpg = args.traj_per_gpu or (WORKLOADS["c5"]["B"] if not x.rehearsal else 4096)
    chunk = args.chunk_traj or (32768 if not x.rehearsal else 1024)

    def emit_with_c5(info):
        if x.rank == 0:
            result["c5"] = info
            print(json.dumps(result), flush=True)
    try:
        info = run_c5(x.torch, x.B, x.D, x.rank, x.world, x.dev, x.rehearsal, tpg, chunk, WORKLOADS["c5"]["N"], stall_cb=emit_with_c5,
                      stall_seconds=args.stall_seconds, inject_stall_s=args.inject_stall, fit_rows=args.fit_rows)
    except Exception as e:                  # sizes are symmetric over ranks, so a failure (e.g. out of memory) is too
        info = {"error": f"{type(e).__name__}: {e}"[:300]}
    return info


def worker(args):
    x = setup(args)
    if args.workload == "c5":
        headline_c5(x)
        finish(x)
        return
    timed_steps(x)
    result = headline_result(x)
    if x.world > 1:
        result["collect"] = collect_report(x)
    x.partial.update(result)
    if x.rank == 0:
        accuracy_gate(x, result)
    if args.kernel == "pipeline" and not args.no_other_rows:
        result["fit_rows"] = other_rows_report(x)
    # extras (rank 0, N=1): PCIe-inclusive rate, the HBM-regime config, the robust and the whole-run chains, the drop-in's single-run latency
    if x.world == 1 and not args.no_extra and not x.time_major:
        result["extra"] = extras(x.torch, x.B, x.L, x.ctx, x.batch, x.out, x.launch, x.Bn, x.N, x.dev, args)
    x.batch = x.out = x.launch = x.collect = x.gathered = x.R = None
    x.torch.cuda.empty_cache()
    if x.world > 1 and not args.no_extra:
        result["c5"] = c5_leg(x, result)
    if x.rank == 0 and x.world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(x.B, x.N, fit_rows=args.fit_rows)
    if x.rank == 0:
        print(json.dumps(result))
    finish(x)


def extras(torch, B, L, ctx, batch, out, launch, Bn, N, dev, args):
    import numpy as np
    extra = {}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    # PCIe-inclusive: pinned host arrays -> device, the step, fused poses back (never `value`)
    names = ("ts", "pos", "quat", "gps", "valid")
    hin = {k: getattr(batch, k).cpu().pin_memory() for k in names}
    hout = torch.empty_like(out.buf, device="cpu").pin_memory()

    def e2e():
        for k in names:
            getattr(batch, k).copy_(hin[k], non_blocking=True)
        launch()
        hout.copy_(out.buf, non_blocking=True)
    reps = 20 if Bn * N < 1e7 else 3
    e2e()                                       # (untimed: the first transfer from / to a freshly pinned buffer maps it -- 90 ms once)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e2e()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    extra["pcie_inclusive"] = {"ms_per_step": dt * 1e3, "poses_per_s": Bn * N / dt, "bytes_over_pcie_per_step": Bn * N * ALG_BYTES_PER_POSE,
                               "GBps": Bn * N * ALG_BYTES_PER_POSE / dt / 1e9, "note": "pinned host buffers, H2D inputs + step + D2H fused poses, serial on one stream"}
    del hin, hout
    if args.workload == "c2" and args.kernel == "pipeline" and batch.layout == B.LAYOUT_TRAJ_MAJOR:
        # the HOST-POINTER entry of the boundary (include/gsf.h: gsf_fuse_pipeline_batch on plain host arrays -- staging arena + pinned mirror):
        # what a cgo / JNI / ctypes caller with its data in host memory gets per call, PCIe included
        import ctypes as C_
        from gps_optimize_slam_amd import _lib as _l
        hh = batch.host_traj_major()
        arrs = [np.ascontiguousarray(hh[k]) for k in ("ts", "pos", "quat", "gps", "valid")]
        Rh, th, sh = np.empty((Bn, 9)), np.empty((Bn, 3)), np.empty(Bn)
        ph, qh, sth = np.empty((Bn, N, 3)), np.empty((Bn, N, 4)), np.empty(Bn, np.int32)
        cfg_h = _l.EkfConfig.from_config(B.CONFIG)

        def host_call():
            _l.check(L.gsf_fuse_pipeline_batch(ctx.handle, 0, *[_l.hptr(a) for a in arrs], C_.byref(cfg_h), Bn, N, _l.hptr(Rh), _l.hptr(th), _l.hptr(sh),
                                               _l.hptr(ph), _l.hptr(qh), _l.hptr(sth)))
        try:
            host_call(); host_call()
            t0 = time.perf_counter()
            for _ in range(10):
                host_call()
            dth = (time.perf_counter() - t0) / 10
            extra["host_pointer_entry"] = {"entry": "gsf_fuse_pipeline_batch (host arrays in, host arrays out)", "ms_per_call": dth * 1e3, "poses_per_s": Bn * N / dth,
                                           "alg_bytes_through_the_call_GBps": Bn * N * ALG_BYTES_PER_POSE / dth / 1e9}
        except Exception as e:
            extra["host_pointer_entry"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        del arrs, ph, qh
    if args.workload == "c2":
        for name, (b3, n3, reps) in {"c3_100k_x_1k": (100_000, 1000, 5)}.items():
            ab = b3 * n3 * ALG_BYTES_PER_POSE
            extra[name] = {}
            for lname, lay in (("traj_major_wave_per_traj", B.LAYOUT_TRAJ_MAJOR), ("time_major_lane_per_traj", B.LAYOUT_TIME_MAJOR)):
                bt = B.TrajectoryBatch.synthetic(b3, n3, layout=lay, seed=1)
                o = B.FusedPoses(bt.layout, b3, n3, dev)
                ms_e = timed(lambda: B.ekf_fuse_batch(bt, out=o), reps)
                Rk = [None]

                def pipe():
                    Rk[0] = B.fuse_pipeline_batch(bt, out=o, fit_rows=args.fit_rows)[1]
                ms_p = timed(pipe, reps)
                ms_pa = timed(lambda: B.fuse_pipeline_batch(bt, out=o, fit_rows="all" if args.fit_rows == "reference" else "reference"), reps)
                gate = None
                if lay == B.LAYOUT_TRAJ_MAJOR:                              # the timed outputs of the headline definition, a sample spread over the batch
                    pipe(); torch.cuda.synchronize()
                    try:
                        gate = oracle_gate(np, torch, B, bt, o.pos, o.status, spread_sample(b3, 64), args.fit_rows, Rk[0])
                    except Exception as e:
                        gate = {"error": f"{type(e).__name__}: {e}"[:200]}
                extra[name][lname] = {"fit_rows": args.fit_rows, "pipeline_kernel_ms_other_fit_rows": ms_pa, "gate_vs_oracle": gate,"ekf_kernel_ms": ms_e, "ekf_poses_per_s": b3 * n3 / ms_e * 1e3, "ekf_alg_GBps": ab / ms_e / 1e6,
                                      "ekf_hbm_frac": ab / ms_e / 1e6 / HBM_PEAK_GBS, "pipeline_kernel_ms": ms_p,
                                      "pipeline_poses_per_s": b3 * n3 / ms_p * 1e3, "pipeline_alg_GBps": ab / ms_p / 1e6,
                                      "pipeline_hbm_frac": ab / ms_p / 1e6 / HBM_PEAK_GBS,
                                      "pipeline_frac_on_moved_194B": (ab / ALG_BYTES_PER_POSE * (ALG_BYTES_PER_POSE + FIT_REREAD_BYTES_PER_POSE)) / ms_p / 1e6 / HBM_PEAK_GBS}
                del bt, o
                torch.cuda.empty_cache()
        # a SMALL time-major batch (C2 shape): routed through transpose -> wave kernel -> transpose by the library
        bt = B.TrajectoryBatch.synthetic(1000, 271, layout=B.LAYOUT_TIME_MAJOR, seed=1)
        o = B.FusedPoses(bt.layout, 1000, 271, dev)
        extra["c2_time_major_pipeline_ms"] = timed(lambda: B.fuse_pipeline_batch(bt, out=o), 50)
        del bt, o
        # KITTI-04-length tracks in batches of other sizes: the headline batch (1 000 tracks = one wave per SIMD) is the latency-bound end of
        # the curve, the chip's throughput on 271-pose tracks is what the large batches show (same kernels, same bits per track)
        sweep = {}
        for nb_ in (256, 1000, 2048, 4096, 16384, 131_072):          # (131 072, not 100 000: the C3 launches have that grid, and the committed trace is keyed by kernel and grid)
            bt = B.TrajectoryBatch.synthetic(nb_, 271, layout=B.LAYOUT_TRAJ_MAJOR, seed=SEED)
            o = B.FusedPoses(bt.layout, nb_, 271, dev)
            ms_ = timed(lambda: B.fuse_pipeline_batch(bt, out=o, fit_rows=args.fit_rows), 50 if nb_ <= 16384 else 10)
            sweep[str(nb_)] = {"kernel_ms_eager_launches": ms_, "poses_per_s": nb_ * 271 / ms_ * 1e3, "hbm_frac": nb_ * 271 * ALG_BYTES_PER_POSE / ms_ / 1e6 / HBM_PEAK_GBS}
            del bt, o
        extra["batch_size_sweep_271_pose_tracks"] = sweep
        torch.cuda.empty_cache()
        # the path from the geodetic GNSS log (K1 -> time alignment -> fit -> EKF, one device chain) and the robust chain (device-side
        # legacy-MT19937 draws -> RANSAC -> inlier refit -> Sim3 of pose 0 -> EKF), both at the C2 shape
        gb = B.GeodeticBatch.synthetic(1000, 271, seed=SEED)
        og = B.FusedPoses(B.LAYOUT_TRAJ_MAJOR, 1000, 271, dev)
        ms_g = timed(lambda: B.fuse_from_geodetic(gb, out=og), 50)
        extra["geodetic_chain_c2"] = {"ms": ms_g, "poses_per_s": 271e3 / ms_g * 1e3, "gnss_fixes": int(gb.gps_t.numel()),
                                      "stages": "gsf_gps_rows_to_utm_batch_dev -> gsf_time_align_batch_dev -> gsf_fuse_pipeline_batch_dev"}
        del gb, og
        extra["full_chain_c2"] = full_chain(torch, np, B, timed, dev, 1000, 271)
        extra["robust_chain_c2"] = robust_chain(torch, np, B, timed, dev, 1000, 271, args.fit_rows, parity=32)
        extra["robust_chain_8192"] = robust_chain(torch, np, B, timed, dev, 8192, 271, args.fit_rows, parity=0)
        a_, b_ = extra["robust_chain_c2"], extra["robust_chain_8192"]
        if "ms" in a_ and "ms" in b_:
            b_["per_stream_cost_vs_1000_streams"] = (b_["ms"] / 8192) / (a_["ms"] / 1000)
            b_["draws_per_stream_cost_vs_1000_streams"] = (b_["draws_ms"] / 8192) / (a_["draws_ms"] / 1000)
            b_["all_trials_per_stream_cost_vs_1000_streams"] = (b_["all_trials"]["ms"] / 8192) / (a_["all_trials"]["ms"] / 1000)
        extra["c1_drop_in"] = c1_latency(np)
        extra["fit_distributions_c2"] = fit_distributions(torch, np, B, timed, dev)
        extra["c4_1M_x_50"] = c4_windows(torch, B, timed)
        torch.cuda.empty_cache()
        extra["c5_shard_1gpu"] = c5_shard(torch, B, L, dev, timed, fit_rows=args.fit_rows)
    return extra


def robust_chain(torch, np, B, timed, dev, nb, N, fit_rows, parity=0):
    """Steps 3-5 with the reference's ROBUST fit as one device chain at `nb` streams (EKFGPSSLAM.py:1002-1010, :389-426), in both modes: every
    trajectory drawing all max_trials (its generator ends where np.random ends in the reference), and stopping at the first trial that counts
    every row (ref :413; `ms`, the mode batch.py uses by default) -- same R / t / s / masks / poses bit for bit, checked here on the timed
    outputs.  With parity > 0 a sample of the TIMED run against the oracle fed with NumPy's own draws for the same seeds."""
    from oracle import oracle as orc
    try:
        bt = B.TrajectoryBatch.synthetic(nb, N, layout=B.LAYOUT_TRAJ_MAJOR, seed=SEED)
        o = B.FusedPoses(bt.layout, nb, N, dev)
        o_full = B.FusedPoses(bt.layout, nb, N, dev)
        st0 = B.mt19937_seed(np.arange(nb))
        sc = B.CONFIG["sim3_ransac"]
        keep, keep_full = [None], [None]

        def chain():
            keep[0] = B.fuse_pipeline_robust_batch(bt, st0.clone(), out=o, want_mask=False, fit_rows=fit_rows, early_exit=True, return_info=True)

        def chain_full():
            keep_full[0] = B.fuse_pipeline_robust_batch(bt, st0.clone(), out=o_full, want_mask=False, fit_rows=fit_rows, early_exit=False, return_info=True)
        ms_full = timed(chain_full, 5)
        ms_r = timed(chain, 20)
        npop = [N] * nb                                                    # (host values: the library then knows the largest population)
        ms_d = timed(lambda: B.mt19937_choice_batch(st0.clone(), npop, sc["max_trials"], sc["min_samples"]), 5)
        torch.cuda.synchronize()
        (oe, Re, te, se, nine, _, infoe), (of, Rf, tf, sf, ninf, _, infof) = keep[0], keep_full[0]
        sat = ((oe.status >> 8) & 256) != 0
        same = all(torch.equal(torch.nan_to_num(a, nan=-1.0).view(torch.int64), torch.nan_to_num(b_, nan=-1.0).view(torch.int64))
                   for a, b_ in ((oe.pos, of.pos), (oe.quat, of.quat), (Re, Rf), (te, tf), (se, sf)))
        same = same and bool(torch.equal(nine, ninf)) and bool(torch.equal(oe.status & ~(256 << 8), of.status)) and bool(torch.equal(infoe[:, 0], infof[:, 0]))
        drawn = infoe[:, 1].long()
        hist = torch.bincount(drawn[sat]) if bool(sat.any()) else torch.zeros(1, dtype=torch.long)
        res = {"ms": ms_r, "poses_per_s": nb * N / ms_r * 1e3, "streams": nb, "max_trials": sc["max_trials"], "fit_rows": fit_rows,
               "mode": "ransac_early_exit = 1 (batch.py's default): a trajectory stops at the first trial that counts every row (ref :413)",
               "ms_per_1000_streams": ms_r / nb * 1000,
               "all_trials": {"ms": ms_full, "poses_per_s": nb * N / ms_full * 1e3, "ms_per_1000_streams": ms_full / nb * 1000,
                              "mode": "ransac_early_exit = 0: every trajectory draws max_trials; generators end where the reference leaves np.random"},
               "speedup_over_all_trials": ms_full / ms_r,
               "outputs_identical_in_both_modes": bool(same),
               "saturated_trajectories": int(sat.sum().item()), "saturated_share": float(sat.double().mean().item()),
               "trials_drawn_by_saturated_trajectories_histogram": {str(k): int(v) for k, v in enumerate(hist.tolist()) if v},
               "unsaturated_trajectories_draw": sc["max_trials"],
               "deciding_trial_max_among_saturated": int(infoe[sat, 0].max().item()) if bool(sat.any()) else None,
               "draws_ms": ms_d, "draws_wall_us_per_trial_of_every_stream": ms_d * 1e3 / sc["max_trials"],
               "draws_wall_ns_per_stream_and_trial": ms_d * 1e6 / (nb * sc["max_trials"]),
               "stages": "row choice -> compact -> early-exit probe (rounds of 8 drawn-and-scored trials per trajectory) -> mt19937 choice + K2b for "
                         "the trials left of undecided trajectories -> Sim3(pose 0) -> K4"}
        if parity > 0:
            chain(); torch.cuda.synchronize()
            out, R, t, s, nin, _, _ = keep[0]
            idx = spread_sample(nb, parity)
            ix = torch.as_tensor(idx, device=dev)
            hh = {k_: getattr(bt, k_).index_select(0, ix).cpu().numpy() for k_ in ("ts", "pos", "quat", "gps", "valid")}
            pg, sg = out.pos.index_select(0, ix).cpu().numpy(), out.status.index_select(0, ix).cpu().numpy()
            Rg, ng = R.index_select(0, ix).cpu().numpy(), nin.index_select(0, ix).cpu().numpy()
            ta, tg = B.CONFIG["time_alignment"], B.CONFIG["sim3_ransac"]
            worst, rworst, ok_status, ok_count = 0.0, 0.0, True, True
            for k_, b_ in enumerate(idx):
                okr = (hh["valid"][k_] != 0) & ~np.isnan(hh["gps"][k_]).any(axis=1)
                rows = np.where(okr)[0] if fit_rows == "all" else orc.pick_sim3_rows(hh["ts"][k_], okr, tg["min_samples"], ta["max_gps_gap_threshold"], tg["max_initial_duration"])
                np.random.seed(int(b_))
                draws = np.stack([np.random.choice(len(rows), sc["min_samples"], replace=False) for _ in range(sc["max_trials"])]).astype(np.int32)
                Ro, to, so, mo = orc.compute_sim3_transform_robust(hh["pos"][k_][rows], hh["gps"][k_][rows], sc["min_samples"], sc["residual_threshold"], sc["max_trials"],
                                                                   sc["min_inliers_needed"], sample_idx=draws, return_mask=True)
                sp, sq = orc.transform_trajectory(hh["pos"][k_][:1], hh["quat"][k_][:1], Ro, to, so)
                po, qo, sto = orc.apply_ekf_correction_aligned(hh["ts"][k_], hh["pos"][k_], hh["quat"][k_], hh["gps"][k_], hh["valid"][k_], sp[0], sq[0], return_status=True)
                worst = max(worst, float(np.abs(pg[k_] - po).max())); rworst = max(rworst, float(np.abs(Rg[k_].reshape(3, 3) - Ro).max()))
                ok_status &= bool((sg[k_] & 0xff) == sto); ok_count &= bool(ng[k_] == mo.sum())
            res["parity_sample_vs_oracle"] = {"trajectories": int(len(idx)), "of_the_timed_run": True, "max_abs_pos_err_m": worst, "max_abs_sim3_R_err": rworst,
                                              "status_bits_equal": ok_status, "inlier_counts_equal": ok_count,
                                              "how": "oracle.compute_sim3_transform_robust fed np.random.choice draws of np.random.seed(trajectory id), rows by the same rule"}
        del bt, o, o_full, st0
        torch.cuda.empty_cache()
        return res
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def full_chain(torch, np, B, timed, dev, nb, N):
    """Steps 1-6 of main_process_gui (EKFGPSSLAM.py:959-1033) for `nb` trajectories as ONE device chain (gsf_run_fusion_batch_dev): geodesy slice
    -> GPS RANSAC pre-filter (windows walked on the device) -> time alignment -> row choice -> robust Sim3 (early exit on) -> apply -> EKF + RTS
    -> error metric; the reference-complete throughput figure next to the plain-fit headline.  Stage split: the stages timed on their own on the
    same data (their sum is not the chain's time: launches overlap their tails)."""
    import ctypes as C
    from gps_optimize_slam_amd import _lib
    try:
        L, ctx = _lib.load(), B.context()
        gb = B.GeodeticBatch.synthetic(nb, N, seed=SEED)
        st0 = B.mt19937_seed(np.arange(nb) + 1)
        keep = [None]

        def chain():
            keep[0] = B.run_fusion_batch(gb, st0.clone(), early_exit=True, want_mask=False)
        ms = timed(chain, 10)
        ms_full = timed(lambda: B.run_fusion_batch(gb, st0.clone(), early_exit=False, want_mask=False), 3)
        torch.cuda.synchronize()
        r = keep[0]
        total = int(gb.gps_t.numel())
        res = {"ms": ms, "poses_per_s": nb * N / ms * 1e3, "trajectories": nb, "poses": N, "gnss_fixes": total,
               "all_trials_ms": ms_full, "all_trials_poses_per_s": nb * N / ms_full * 1e3,
               "stages": "gsf_gps_rows_to_utm -> loader compaction -> gps_prefilter_chain (device-walked windows) -> filtered rows -> time_align -> sim3 rows -> "
                         "compact -> early-exit probe (+ mt19937 choice + K2b for undecided tracks) -> Sim3(pose 0) -> K4 -> apply Sim3 (all poses) -> 3 x eval_errors",
               "run_status_nonzero": int((r.run_status != 0).sum().item()), "saturated_share": float((((r.fused.status >> 8) & 256) != 0).double().mean().item()),
               "fixes_kept_share": float(r.gps_keep.double().mean().item()),
               "step6_rmse_m_mean": {"raw_slam": float(r.err_stats[0, :, 3].nanmean().item()), "sim3": float(r.err_stats[1, :, 3].nanmean().item()),
                                     "ekf": float(r.err_stats[2, :, 3].nanmean().item())}}
        # the same chain on logs with 2 % of the fixes thrown 40 m off: the pre-filter has fixes to drop, its windows need more than one trial
        gbo = gb.with_outliers(0.02, 40.0, seed=7)
        ko = [None]

        def chain_o():
            ko[0] = B.run_fusion_batch(gbo, st0.clone(), early_exit=True, want_mask=False)
        ms_o = timed(chain_o, 10)
        res["with_2pct_outlier_fixes"] = {"ms": ms_o, "poses_per_s": nb * N / ms_o * 1e3, "fixes_kept_share": float(ko[0].gps_keep.double().mean().item()),
                                          "run_status_nonzero": int((ko[0].run_status != 0).sum().item()),
                                          "step6_rmse_m_mean_ekf": float(ko[0].err_stats[2, :, 3].nanmean().item())}
        del gbo
        # stage split, each on its own
        f = dict(dtype=torch.float64, device=dev)
        utm = torch.empty_like(gb.gps_llh); zone = torch.empty(nb, dtype=torch.int32, device=dev); south = torch.empty_like(zone)
        split = {}
        split["geodesy_slice_ms"] = timed(lambda: _lib.check(L.gsf_gps_rows_to_utm_batch_dev(ctx.handle, B._p(gb.gps_llh), B._p(gb.gps_offsets), nb, B._p(utm), B._p(zone), B._p(south))), 20)
        pc = _lib.PrefilterConfig.from_config(B.CONFIG["gps_filtering_ransac"])
        keepm = torch.empty(total, dtype=torch.uint8, device=dev); ls = torch.empty(nb, dtype=torch.int32, device=dev)
        split["prefilter_ms"] = timed(lambda: _lib.check(L.gsf_gps_prefilter_auto_dev(ctx.handle, B._p(gb.gps_t), B._p(utm), B._p(gb.gps_offsets), nb, int(gb.max_fixes), C.byref(pc),
                                                                                    B._p(st0.clone()), B._p(keepm), B._p(ls), None)), 10)
        al = torch.empty((nb, N, 3), **f); va = torch.empty((nb, N), dtype=torch.uint8, device=dev)
        split["time_alignment_ms"] = timed(lambda: _lib.check(L.gsf_time_align_loaded_rows_batch_dev(ctx.handle, B._p(gb.ts), B._p(gb.slam_offsets), B._p(gb.gps_t), B._p(utm),
                                                                                                   B._p(gb.gps_offsets), nb, max(2, int(gb.max_fixes)), 5.0, B._p(al), B._p(va), None)), 20)
        tb = B.TrajectoryBatch(B.LAYOUT_TRAJ_MAJOR, nb, N, dev)
        tb.ts, tb.pos, tb.quat, tb.gps, tb.valid = gb.ts, gb.pos, gb.quat, r.aligned, r.valid
        o = B.FusedPoses(B.LAYOUT_TRAJ_MAJOR, nb, N, dev)
        split["robust_steps_3_to_5_ms"] = timed(lambda: B.fuse_pipeline_robust_batch(tb, st0.clone(), out=o, want_mask=False), 10)
        split["apply_sim3_all_poses_ms"] = timed(lambda: B.apply_sim3_batch(gb.pos.view(-1, 3), gb.quat.view(-1, 4), gb.slam_offsets, r.R, r.t, r.s), 20)
        split["error_metric_3_tracks_ms"] = 3 * timed(lambda: B.eval_errors_batch(gb.ts, r.fused.pos, r.aligned, r.valid, 5.0), 20)
        res["stage_split"] = split
        res["stage_split_sum_ms"] = sum(split.values())
        del gb, r, o, tb, ko
        torch.cuda.empty_cache()
        return res
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def fit_distributions(torch, np, B, timed, dev, nb=1000, N=271):
    """The fused pipeline at the C2 shape on three input distributions -- the fit's rotation comes from a Newton polar iteration with a
    Jacobi-SVD fallback, so its speed must not hinge on the generator: (a) the default workload (white 2 cm SLAM noise), (b) SURVEY 8d
    to the letter (random-walk drift of 2 cm per pose, bursts on 5 % of the tracks), (c) the real C1 track (271 poses of KITTI 04 and
    its time-aligned GNSS, committed fixtures) replicated with independent 0.45 m GNSS noise per copy.  Per distribution: kernel time,
    fallback count, and the timed outputs against the CPU oracle on a sample."""
    from oracle import oracle as orc
    res = {}
    gold = os.path.join(ROOT, "tests", "golden")

    def c1_batch():
        k, g = np.load(os.path.join(gold, "kat_bundled.npz")), np.load(os.path.join(gold, "c1_combined.npz"))
        return B.TrajectoryBatch.replicated(k["ts"], k["pos"], k["quat"], g["aligned"], g["valid"], nb, 0.45, seed=1)
    makers = (("synthetic_white_noise_default", lambda: B.TrajectoryBatch.synthetic(nb, N, layout=B.LAYOUT_TRAJ_MAJOR, seed=SEED)),
              ("synthetic_random_walk_drift_8d", lambda: B.TrajectoryBatch.synthetic(nb, N, layout=B.LAYOUT_TRAJ_MAJOR, seed=SEED, variant=1)),
              ("c1_kitti04_track_replicated", c1_batch))
    for name, make in makers:
        try:
            bt = make()
            o = B.FusedPoses(bt.layout, bt.B, bt.N, dev)
            ms = timed(lambda: B.fuse_pipeline_batch(bt, out=o), 200)
            torch.cuda.synchronize()
            st = o.status.cpu().numpy()
            ns = 32
            hh = {k_: getattr(bt, k_)[:ns].cpu().numpy() for k_ in ("ts", "pos", "quat", "gps", "valid")}
            po, qo, sto, _, _, _ = orc.fuse_pipeline_batch(hh["ts"], hh["pos"], hh["quat"], hh["gps"], hh["valid"])
            fin = np.isfinite(po).all(axis=(1, 2))
            pg = o.pos[:ns].cpu().numpy()
            res[name] = {"kernel_us": ms * 1e3, "fit_fallbacks": int(((st >> 8) & 16 != 0).sum()), "fit_none": int(((st >> 8) & 1 != 0).sum()),
                         "trajectories": int(bt.B), "poses": int(bt.N), "max_abs_pos_err_vs_oracle_m": float(np.abs(pg[fin] - po[fin]).max()),
                         "status_bits_equal": bool(((st[:ns] & 0xff) == (sto & 0xff)).all()),
                         "had_outage": int((st & 1 != 0).sum()), "sharp_turn": int((st & 4 != 0).sum())}
            del bt, o
        except Exception as e:
            res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def planted_windows(torch, nw, W, seed, dev="cuda"):
    """nw windows of W point pairs with a planted (R, t, s) each: dst = s R src + t + 2 cm noise at UTM magnitudes (config C4's input)"""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    f = dict(dtype=torch.float64, device=dev)
    step = torch.randn(nw, W, 3, generator=g, **f) * torch.tensor([0.05, 0.03, 0.1], **f) + torch.tensor([0.0, 0.0, 1.4], **f)
    src = torch.cumsum(step, dim=1)                                        # a short drive per window: 1.4 m per pose along camera z
    q = torch.randn(nw, 4, generator=g, **f); q = q / q.norm(dim=1, keepdim=True)
    x, y, z, w = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1).view(nw, 3, 3)
    sc = 0.9 + 0.2 * torch.rand(nw, generator=g, **f)
    t = torch.tensor([4.5e5, 5.4e6, 110.0], **f) + 50.0 * torch.randn(nw, 3, generator=g, **f)
    dst = sc[:, None, None] * torch.einsum("bij,bwj->bwi", R, src) + t[:, None, :] + 0.02 * torch.randn(nw, W, 3, generator=g, **f)
    return src.contiguous(), dst.contiguous(), R.reshape(nw, 9), t, sc


def c4_windows(torch, B, timed, nw=1_000_000, W=50):
    """BASELINE configs[3]: sliding-window Sim3 re-alignment, 1M windows x 50 point pairs (Umeyama only):
    gsf_sim3_umeyama_windows_dev, algorithmic traffic 48 W + 104 = 2 504 B per window."""
    src, dst, Rp, tp, sp = planted_windows(torch, nw, W, 7)
    R, t, s, st = B.sim3_umeyama_batch(src, dst)
    ms = timed(lambda: B.sim3_umeyama_batch(src, dst), 10)
    alg = nw * (48 * W + 104)
    # the planted transform is recovered to the noise level: the fitted map explains the points (the rotation ABOUT the drive axis is only
    # as well determined as the window's lateral extent allows, so the residual is the meaningful figure, not R - R_planted)
    resid = (s[:, None, None] * torch.einsum("bij,bwj->bwi", R.view(nw, 3, 3), src) + t[:, None, :] - dst).norm(dim=2)
    res = {"ms": ms, "windows_per_s": nw / ms * 1e3, "alg_bytes_per_window": 48 * W + 104, "alg_GBps": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
           "kernels": "windows_fused_kernel (moments in LDS + lane-per-window closed form, one launch)", "status_nonzero": int((st != 0).sum().item()),
           "planted_transform_recovered": {"max_abs_scale_err": float((s - sp).abs().max().item()), "mean_abs_scale_err": float((s - sp).abs().mean().item()),
                                           "fit_residual_rms_m": float(resid.pow(2).mean().sqrt().item()), "fit_residual_max_m": float(resid.max().item()),
                                           "planted_noise_sigma_m_per_axis": 0.02}}
    del src, dst, resid
    return res


def c5_shard(torch, B, L, dev, timed, traj=1_245_184, N=1000, chunk=32768, fit_rows="reference"):
    """BASELINE configs[4] on ONE GPU: the per-GPU shard of 10M x 1k trajectories over 8 GPUs (1 245 184 = 38 chunks of 32 768, sized
    down symmetrically to what is free), compute-only pass of the fused pipeline, in BOTH layouts: trajectory-major (wave per trajectory,
    chunk by chunk as the 8-GPU run does) and time-major (lane per trajectory, the whole shard in one launch)."""
    import ctypes as C

    from gps_optimize_slam_amd import _lib
    free_b, _ = torch.cuda.mem_get_info(dev)
    T = int(min(traj, (free_b * 0.92) // (N * ALG_BYTES_PER_POSE + 64)))
    T -= T % chunk
    res = {"trajectories": T, "poses_per_trajectory": N, "chunk_trajectories": chunk, "requested_trajectories": traj}
    if T <= 0:
        res["error"] = "not enough free memory for one chunk"
        return res
    import numpy as np
    ctx = B.context()
    ctx.set_sim3_rows(fit_rows, B.CONFIG)
    res["fit_rows"] = fit_rows
    cfg = _lib.EkfConfig.from_config(B.CONFIG)
    f = dict(dtype=torch.float64, device=dev)
    alg = T * N * ALG_BYTES_PER_POSE
    for lname, lay in (("traj_major_wave_per_traj", B.LAYOUT_TRAJ_MAJOR), ("time_major_lane_per_traj", B.LAYOUT_TIME_MAJOR)):
        try:
            bt = B.TrajectoryBatch(lay, T, N, dev)
            if lay == B.LAYOUT_TRAJ_MAJOR:
                for lo in range(0, T, 65536):                              # generated in place, 64 Ki trajectories per launch
                    n = min(65536, T - lo)
                    _lib.check(L.gsf_synth_batch_dev(ctx.handle, lay, C.c_uint64(SEED), lo, n, N, B._p(bt.ts[lo:]), B._p(bt.pos[lo:]), B._p(bt.quat[lo:]),
                                                     B._p(bt.gps[lo:]), B._p(bt.valid[lo:]), None, None))
            else:
                _lib.check(L.gsf_synth_batch_dev(ctx.handle, lay, C.c_uint64(SEED), 0, T, N, B._p(bt.ts), B._p(bt.pos), B._p(bt.quat), B._p(bt.gps),
                                                 B._p(bt.valid), None, None))
            out = torch.empty((T * N * 7,), **f)
            R, t, s = torch.empty((T, 9), **f), torch.empty((T, 3), **f), torch.empty((T,), **f)
            status = torch.empty((T,), dtype=torch.int32, device=dev)
            P = chunk * N

            def one_pass():
                if lay == B.LAYOUT_TRAJ_MAJOR:
                    for k in range(T // chunk):
                        lo, o = k * chunk, out[k * P * 7:]
                        _lib.check(L.gsf_fuse_pipeline_batch_dev(ctx.handle, lay, B._p(bt.ts[lo:]), B._p(bt.pos[lo:]), B._p(bt.quat[lo:]), B._p(bt.gps[lo:]),
                                                                 B._p(bt.valid[lo:]), C.byref(cfg), chunk, N, B._p(R[lo:]), B._p(t[lo:]), B._p(s[lo:]),
                                                                 B._p(o), B._p(o[P * 3:]), B._p(status[lo:])))
                else:
                    _lib.check(L.gsf_fuse_pipeline_batch_dev(ctx.handle, lay, B._p(bt.ts), B._p(bt.pos), B._p(bt.quat), B._p(bt.gps), B._p(bt.valid),
                                                             C.byref(cfg), T, N, B._p(R), B._p(t), B._p(s), B._p(out), B._p(out[T * N * 3:]), B._p(status)))
            ms = timed(one_pass, 3)
            res[lname] = {"pass_ms": ms, "poses_per_s": T * N / ms * 1e3, "alg_GBps": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
                          "launches_per_pass": T // chunk if lay == B.LAYOUT_TRAJ_MAJOR else 1,
                          "fit_none": int(((status >> 8) & 1).ne(0).sum().item()), "had_outage": int((status & 1).ne(0).sum().item()),
                          "rts_applied": int((status & 2).ne(0).sum().item()), "sharp_turn": int((status & 4).ne(0).sum().item())}
            if lay == B.LAYOUT_TRAJ_MAJOR:
                # the timed outputs against the oracle: 64 trajectories from the first, a middle and the last chunk of the pass
                try:
                    nck = T // chunk
                    gates = []
                    for k in sorted({0, nck // 2, nck - 1}):
                        lo = k * chunk
                        view = B.TrajectoryBatch.__new__(B.TrajectoryBatch)
                        for nm in ("ts", "pos", "quat", "gps", "valid"):
                            setattr(view, nm, getattr(bt, nm)[lo:lo + chunk])
                        pos_k = out[k * P * 7:k * P * 7 + P * 3].view(chunk, N, 3)
                        gates.append(oracle_gate(np, torch, B, view, pos_k, status[lo:lo + chunk], spread_sample(chunk, 22), fit_rows, R[lo:lo + chunk]))
                    res[lname]["gate_vs_oracle"] = {"chunks": sorted({0, nck // 2, nck - 1}), "trajectories": sum(int(g["sample"].split()[0]) for g in gates), "fit_rows": fit_rows,
                                                    "ate_rmse_vs_cpu_ref_m": float(np.sqrt(np.mean([g["ate_rmse_vs_cpu_ref_m"] ** 2 for g in gates]))),
                                                    "max_abs_pos_err_m": max(g["max_abs_pos_err_m"] for g in gates),
                                                    "max_abs_sim3_R_err": max(g["max_abs_sim3_R_err"] for g in gates),
                                                    "status_words_equal": all(g["status_words_equal"] for g in gates)}
                except Exception as e:
                    res[lname]["gate_vs_oracle"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                view = pos_k = gates = None                              # views of the shard: dropped with it, or the other layout has no room
            del bt, out, R, t, s, status
        except Exception as e:                                             # e.g. out of memory on a smaller card: reported, not fatal
            res[lname] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
    if all("pass_ms" in res.get(k, {}) for k in ("traj_major_wave_per_traj", "time_major_lane_per_traj")):
        res["faster_layout"] = min(("traj_major_wave_per_traj", "time_major_lane_per_traj"), key=lambda k: res[k]["pass_ms"])
    return res


def c1_latency(np):
    """Wall time of the drop-in's single-trajectory functions at the C1 shape (BASELINE configs[0]): the 271-pose KITTI-04 track and its
    279 raw GNSS fixes from the committed fixtures tests/golden/{kat_bundled,c1_combined}.npz (data only; the package itself reads no
    test file): steps 2-5 of main_process_gui (EKFGPSSLAM.py:971-1010) next to the reference's ~0.11 s."""
    from gps_optimize_slam_amd import ekfgpsslam as E
    try:
        gold = os.path.join(ROOT, "tests", "golden")
        k, g = np.load(os.path.join(gold, "kat_bundled.npz")), np.load(os.path.join(gold, "c1_combined.npz"))
        slam = {"timestamps": k["ts"], "positions": k["pos"], "quaternions": k["quat"]}
        res = E.benchmark_c1(slam, g["gps_t_raw"], g["lat"], g["lon"], g["alt"], repeats=20)
        try:                                       # the reference on the same track, timed in the build container (tests/campaigns/time_reference.py)
            prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_timing.json")))[-1]
            c = json.load(open(prof))["cases"]["kitti04_vs_combined_gnss"]
            res["reference_cpu_ms"] = {"steps_2_to_5": c["steps_2_to_5"]["reference_ms"], "compute_sim3_transform_robust": c["compute_sim3_transform_robust"]["reference_ms"],
                                       "apply_ekf_correction": c["apply_ekf_correction"]["reference_ms"], "source": os.path.basename(prof) + " (the reference itself, 1 core)"}
        except Exception as e:
            res["reference_cpu_ms"] = {"error": str(e)[:120]}
        return res
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--kernel", choices=["pipeline", "ekf"], default="pipeline", help="step = fused pipeline (default) or K4 only")
    ap.add_argument("--layout", choices=["traj", "time"], default="traj",
                    help="batch layout of the c2 / c3 step: trajectory-major (wave per trajectory, default) or time-major (lane per trajectory above "
                         "32 768 tracks, else transposed and run by the wave kernel)")
    ap.add_argument("--fit-rows", choices=["reference", "all"], default="reference",
                    help="rows of the pipeline's Sim3 fit: what main_process_gui picks (EKFGPSSLAM.py:973-998, default) or every valid row")
    ap.add_argument("--traj-per-gpu", type=int, default=None, help="override the workload's trajectories per GPU (c5: default 1 250 000, sized down to what fits)")
    ap.add_argument("--chunk-traj", type=int, default=None, help="c5: trajectories per chunk (default 32 768)")
    ap.add_argument("--poses", type=int, default=None, help="override the workload's poses per trajectory (counter studies: two track lengths 64 poses apart differ by exactly one chunk of the wave kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-rows", action="store_true", help="do not time the other --fit-rows definition after the headline (counter passes: every launch of the run is then the headline's)")
    ap.add_argument("--no-graph", action="store_true", help="time K eager launches instead of one hipGraph replay of them")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (C3 figures, PCIe-inclusive rate, C1 latency; N>1: the C5 leg)")
    ap.add_argument("--stall-seconds", type=float, default=240.0, help="N>1: watchdog per collect leg of the C5-shaped run (stall -> partial line, exit code 3)")
    ap.add_argument("--deadline-s", type=float, default=540.0, help="N>1: run-wide deadline; when it fires rank 0 prints what it has and every rank exits with code 3")
    ap.add_argument("--inject-stall", type=float, default=0.0, help="test hook: every collect leg of the C5-shaped run sleeps this long first")
    ap.add_argument("--set-option", action="append", default=[], metavar="KEY=VALUE", help="gsf_set_option tuning knob (e.g. duo_kernel=0)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    worker(args)


if __name__ == "__main__":
    main()
