"""K3 (apply Sim3): slab build against the per-pose build (gsf_set_option "ekf_variant" 11) -- identical bits on ragged sets with odd
lengths, short tracks, invalid quaternions; time at 1e8 poses (1e5 x 1 000) and at the C2 / C1 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gps_optimize_slam_amd import batch as B


def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


g = torch.Generator(device="cuda"); g.manual_seed(3)
dev = "cuda"
ok = True


def both(pos, quat, offs, R, t, s):
    out = {}
    for var in (0, 11):
        B.context().set_option("ekf_variant", var)
        out[var] = [o.cpu().numpy() for o in B.apply_sim3_batch(pos, quat, offs, R, t, s)]
    B.context().set_option("ekf_variant", 0)
    return all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[0], out[11]))


rng = np.random.default_rng(1)
for trial in range(6):
    lens = rng.integers(1, 400, size=int(rng.integers(1, 300)))
    if trial == 0: lens = np.array([1, 2, 3, 63, 64, 65, 127, 128, 129, 271, 1000, 1025])
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=dev)
    n = int(offs[-1]); nb = len(lens)
    pos = torch.randn(n, 3, dtype=torch.float64, device=dev, generator=g) * 100; quat = torch.randn(n, 4, dtype=torch.float64, device=dev, generator=g)
    quat[rng.integers(0, n, size=max(1, n // 500))] = 0.0                 # invalid quaternions -> NaN rows + flag
    A = torch.linalg.qr(torch.randn(nb, 3, 3, dtype=torch.float64, device=dev, generator=g))[0]
    R = (A * torch.sign(torch.linalg.det(A)).reshape(nb, 1, 1)).reshape(nb, 9).contiguous()
    t = torch.randn(nb, 3, dtype=torch.float64, device=dev, generator=g) * 1e5; s = torch.rand(nb, dtype=torch.float64, device=dev, generator=g) + 0.5
    same = both(pos, quat, offs, R, t, s); ok &= same
    print(f"ragged case {trial}: {nb} tracks, {n} poses: identical {same}", flush=True)

for nb, n in ((100_000, 1000), (1000, 271), (1, 271)):
    offs = torch.arange(0, nb * n + 1, n, dtype=torch.int64, device=dev)
    pos = torch.randn(nb * n, 3, dtype=torch.float64, device=dev, generator=g); quat = torch.randn(nb * n, 4, dtype=torch.float64, device=dev, generator=g)
    R = torch.eye(3, dtype=torch.float64, device=dev).reshape(1, 9).repeat(nb, 1).contiguous(); t = torch.zeros(nb, 3, dtype=torch.float64, device=dev); s = torch.ones(nb, dtype=torch.float64, device=dev)
    same = both(pos, quat, offs, R, t, s); ok &= same
    ms = {}
    for rep in range(2):
        for var in (11, 0):
            B.context().set_option("ekf_variant", var)
            ms.setdefault(var, []).append(timed(lambda: B.apply_sim3_batch(pos, quat, offs, R, t, s)))
    B.context().set_option("ekf_variant", 0)
    print(f"{nb} x {n}: per-pose {min(ms[11]) * 1e3:9.1f} us ({nb * n * 112 / min(ms[11]) / 1e6:7.0f} GB/s)   slabs {min(ms[0]) * 1e3:9.1f} us ({nb * n * 112 / min(ms[0]) / 1e6:7.0f} GB/s)   identical {same}", flush=True)
    del pos, quat
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
