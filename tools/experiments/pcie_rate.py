import torch, time
for mb in (2, 8, 32, 256):
    n = mb * 1024 * 1024 // 8
    h = torch.empty(n, dtype=torch.float64).pin_memory(); d = torch.empty(n, dtype=torch.float64, device="cuda")
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {mb:4d} MB: {dt*1e3:7.3f} ms  {mb/1024/dt:6.1f} GB/s")
