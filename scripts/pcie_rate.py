"""Host <-> device copy rates of the box through pinned memory (what the one-shot calls' uploads and read-backs are bound by):
python scripts/pcie_rate.py -> one JSON line. Sizes: the 10 MB of BASELINE configs[2]'s observations, 64 and 96 MB (configs[4]:
per-observation costs down, regrouped observations up)."""
import json, time
import torch
out = {}
for mb in (10, 64, 96):
    n = mb * 1000 * 1000
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, (dst, src) in {"h2d": (d, h), "d2h": (h, d)}.items():
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t = time.perf_counter(); dst.copy_(src, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ts.sort()
        out[f"{name}_{mb}MB_GBps"] = round(n / ts[len(ts) // 2] / 1e9, 1)
        out[f"{name}_{mb}MB_ms"] = round(ts[len(ts) // 2] * 1e3, 3)
print(json.dumps(out))
