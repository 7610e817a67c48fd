import torch, time
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for gb in (1, 8, 32, 48, 64):
    t = time.perf_counter(); x = torch.empty(gb << 30, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    t = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t3 = time.perf_counter() - t
    t = time.perf_counter(); del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t4 = time.perf_counter() - t
    print(f"{gb} GiB: malloc {t1*1e3:.1f} ms, first touch {t2*1e3:.1f} ms, second touch {t3*1e3:.1f} ms, free {t4*1e3:.1f} ms")
