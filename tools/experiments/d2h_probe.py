#!/usr/bin/env python3
"""Experiment: device-to-host copy of a gathered row block (12.8 MB, BASELINE config 5's shard) - pageable `.cpu()` against a
cached pinned buffer.  usage (GPU box): python3 tools/experiments/d2h_probe.py"""
import time
import torch

x = torch.randint(0, 255, (12838750,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for name in ("pageable .cpu()", "pinned, cached", "pinned, allocated per call"):
    pinned = torch.empty(x.shape, dtype=torch.uint8, pin_memory=True) if name == "pinned, cached" else None
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "pageable .cpu()":
            h = x.cpu()
        elif name == "pinned, cached":
            pinned.copy_(x, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            h = pinned
        else:
            p = torch.empty(x.shape, dtype=torch.uint8, pin_memory=True)
            p.copy_(x, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            h = p
        n = h.numpy()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-28s %.3f ms (min %.3f)" % (name, sorted(ts)[len(ts) // 2], min(ts)))
