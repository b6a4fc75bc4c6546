"""The drop-in boundary used from plain C (examples/fuse_batch.c): what a cgo / JNI / FFI binding of another host language sees.
CPU tier: include/gsf.h is valid C99 and C++11, the consumer builds warning-free against libgsf.so and fails loudly without a device.
GPU tier: its outputs equal the Python route's bit for bit (same library, same kernels, no Python or torch in the process)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gps_optimize_slam_amd")
BUILD = os.path.join(ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def exe():
    from gps_optimize_slam_amd import _lib
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "fuse_batch")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "fuse_batch.c"), "-o", out, "-L" + PKG, "-lgsf", "-Wl,-rpath," + PKG])
    return out


@pytest.fixture(scope="module")
def exe_run():
    from gps_optimize_slam_amd import _lib
    if not os.path.exists(_lib.library_path()):
        _lib.build_library()
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "run_fusion")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "run_fusion.c"), "-o", out, "-L" + PKG, "-lgsf", "-Wl,-rpath," + PKG])
    return out


def test_run_fusion_consumer_builds_and_fails_loudly_without_a_device(exe_run, tmp_path):
    from gps_optimize_slam_amd import _lib
    if _lib.load().gsf_device_count() > 0:
        pytest.skip("a HIP device is present")
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([1, 4, 2], dtype=np.int64).tofile(f)
        for n in (4, 12, 16):
            np.zeros(n).tofile(f)
        np.array([0, 2], dtype=np.int64).tofile(f); np.zeros(2).tofile(f); np.zeros(6).tofile(f)
    r = subprocess.run([exe_run, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not (tmp_path / "out.bin").exists()


@pytest.mark.gpu
def test_run_fusion_consumer_equals_the_python_route(exe_run, tmp_path):
    """examples/run_fusion.c (gsf_run_fusion_batch on host arrays, its generators seeded in C like np.random.seed) against batch.run_fusion_batch on
    the same synthetic geodetic logs: every output word."""
    import torch
    from gps_optimize_slam_amd import batch as B
    nb, N = 40, 271
    gb = B.GeodeticBatch.synthetic(nb, N, seed=5)
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([nb, N, gb.gps_t.numel()], dtype=np.int64).tofile(f)
        for a in (gb.ts, gb.pos, gb.quat, gb.gps_offsets, gb.gps_t, gb.gps_llh):
            a.cpu().numpy().tofile(f)
    r = subprocess.run([exe_run, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "7"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"ran {nb} trajectories x {N} poses" in r.stdout and "0 runs the reference would have aborted" in r.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    P, T = nb * N, gb.gps_t.numel()
    nd = nb * 13 + P * 7 + nb * 12
    fl = np.frombuffer(raw[:8 * nd], dtype=np.float64)
    it = np.frombuffer(raw[8 * nd:8 * nd + 16 * nb], dtype=np.int32).reshape(4, nb)
    keep = np.frombuffer(raw[8 * nd + 16 * nb:], dtype=np.uint8)
    st = B.mt19937_seed(np.arange(nb) + 7)
    res = B.run_fusion_batch(gb, st, early_exit=False)
    p, q, status = res.fused.host_traj_major()
    o = 0
    for ref in (res.R.cpu().numpy(), res.t.cpu().numpy(), res.s.cpu().numpy(), p, q, res.err_stats.cpu().numpy()):
        np.testing.assert_array_equal(np.nan_to_num(fl[o:o + ref.size], nan=-1.0), np.nan_to_num(ref.ravel(), nan=-1.0)); o += ref.size
    for k, ref in enumerate((status, res.n_inliers.cpu().numpy(), res.zone.cpu().numpy(), res.run_status.cpu().numpy())):
        np.testing.assert_array_equal(it[k], ref)
    np.testing.assert_array_equal(keep, res.gps_keep.cpu().numpy())
    assert len(keep) == T


def test_header_is_plain_c_and_cxx():
    h = os.path.join(ROOT, "include", "gsf.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", h])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c++", h])
    src = open(h).read()
    assert "torch" not in src.replace("torch's current stream", "") and "at::" not in src and "std::" not in src      # plain pointers and sizes only


def _write_input(path, ts, pos, quat, gps, valid):
    with open(path, "wb") as f:
        np.array(ts.shape, dtype=np.int64).tofile(f)
        for a in (ts, pos, quat, gps):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
        np.ascontiguousarray(valid, dtype=np.uint8).tofile(f)


def test_consumer_fails_loudly_without_a_device(exe, tmp_path):
    from gps_optimize_slam_amd import _lib
    if _lib.load().gsf_device_count() > 0:
        pytest.skip("a HIP device is present")
    ts = np.zeros((1, 4)); _write_input(tmp_path / "in.bin", ts, np.zeros((1, 4, 3)), np.zeros((1, 4, 4)), np.zeros((1, 4, 3)), np.ones((1, 4), np.uint8))
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not (tmp_path / "out.bin").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("rows", ["reference", "all"])
def test_c_consumer_equals_the_python_route(exe, tmp_path, rows):
    """rows="reference": the C process sets NOTHING on its fresh context -- the library's default row rule must be the reference's flow
    (main_process_gui, EKFGPSSLAM.py:973-998) and equal the Python route's fit_rows="reference" bit for bit; "all": it switches to mode 0."""
    import torch
    from gps_optimize_slam_amd import batch as B
    nb, N = 96, 271
    bt = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=11)
    h = bt.host_traj_major()
    _write_input(tmp_path / "in.bin", h["ts"], h["pos"], h["quat"], h["gps"], h["valid"])
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + (["all"] if rows == "all" else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "gfx950" in r.stdout and f"fused {nb} trajectories x {N} poses" in r.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    P = nb * N
    f = np.frombuffer(raw[:8 * (nb * 13 + P * 7)], dtype=np.float64)
    st = np.frombuffer(raw[8 * (nb * 13 + P * 7):], dtype=np.int32)
    Rc, tc, sc = f[:nb * 9].reshape(nb, 9), f[nb * 9:nb * 12].reshape(nb, 3), f[nb * 12:nb * 13]
    pc, qc = f[nb * 13:nb * 13 + P * 3].reshape(nb, N, 3), f[nb * 13 + P * 3:].reshape(nb, N, 4)
    ctx = B.context()
    was = ctx.options.get("block_kernel", -1)              # (a test tier may route this process through the opt-in block kernel by environment;
    ctx.set_option("duo_kernel", -1); ctx.set_option("block_kernel", -1)   #  the C process has the library's defaults)
    try:
        out, R, t, s = B.fuse_pipeline_batch(bt, fit_rows=rows)
        p, q, stp = out.host_traj_major()
    finally:
        ctx.set_option("block_kernel", was)
    np.testing.assert_array_equal(st, stp)
    for a, b in ((Rc, R.cpu().numpy()), (tc, t.cpu().numpy()), (sc, s.cpu().numpy()), (pc, p), (qc, q)):
        np.testing.assert_array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))
    assert (st & 1).any() and np.isfinite(pc).all()
