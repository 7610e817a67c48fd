#!/usr/bin/env python3
"""Device bandwidth probes beside the bench numbers: fill (write only), copy (read + write), reduce (read only)."""
import json, torch
dev = torch.device("cuda:0")
n = 1 << 32
a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tf = t(lambda: a.zero_()); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.view(torch.int32).sum())
print(json.dumps({"bytes": n, "fill_GBps": round(n / tf / 1e9, 1), "copy_GBps_read_plus_write": round(2 * n / tc / 1e9, 1), "read_sum_GBps": round(n / tr / 1e9, 1)}))
