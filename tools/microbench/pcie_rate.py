import torch, time
x = torch.empty(650*1024*1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (48<<20, 650<<20):
    for rep in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        d[:n].copy_(x[:n], non_blocking=True); torch.cuda.synchronize()
        dt=time.perf_counter()-t0
    print("H2D", n>>20, "MB", round(dt*1e3,2), "ms", round(n/dt/1e9,1), "GB/s")
    for rep in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        x[:n].copy_(d[:n], non_blocking=True); torch.cuda.synchronize()
        dt=time.perf_counter()-t0
    print("D2H", n>>20, "MB", round(dt*1e3,2), "ms", round(n/dt/1e9,1), "GB/s")
