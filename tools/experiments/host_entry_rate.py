"""Rate of the HOST-POINTER entry of the boundary (gsf_fuse_pipeline_batch on plain NumPy arrays: staging arena + pinned mirror, include/gsf.h)
at the C2 shape and at 16 384 x 271: what a cgo / JNI / ctypes caller that holds its data in host memory gets per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
from gps_optimize_slam_amd import _lib, batch as B
L = _lib.load(); ctx = B.context(); ctx.set_sim3_rows("reference", B.CONFIG)
cfg = _lib.EkfConfig.from_config(B.CONFIG)
for nb, N in ((1000, 271), (16384, 271), (1000, 1000)):
    h = B.TrajectoryBatch.synthetic(nb, N, layout=0, seed=1).host_traj_major()
    ts, pos, quat, gps, valid = (np.ascontiguousarray(h[k]) for k in ("ts", "pos", "quat", "gps", "valid"))
    R, t, s = np.empty((nb, 9)), np.empty((nb, 3)), np.empty(nb)
    po, qo, st = np.empty((nb, N, 3)), np.empty((nb, N, 4)), np.empty(nb, np.int32)
    call = lambda: _lib.check(L.gsf_fuse_pipeline_batch(ctx.handle, 0, _lib.hptr(ts), _lib.hptr(pos), _lib.hptr(quat), _lib.hptr(gps), _lib.hptr(valid), C.byref(cfg), nb, N,
                                                        _lib.hptr(R), _lib.hptr(t), _lib.hptr(s), _lib.hptr(po), _lib.hptr(qo), _lib.hptr(st)))
    call(); call()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps): call()
    dt = (time.perf_counter() - t0) / reps
    byt = nb * N * 145
    print(f"{nb} x {N}: {dt * 1e3:.3f} ms per call, {nb * N / dt / 1e6:.1f} M poses/s, {byt / dt / 1e9:.2f} GB/s of algorithmic bytes through the call")
