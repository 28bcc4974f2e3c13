#!/usr/bin/env python3
"""What the one-time allocations of a cold file run cost: hipMalloc of work-set sized buffers (kept alive, one after the
other; then with a pause between them; then while a kernel runs) and page-locked host buffers.  torch's allocator passes
requests of this size straight to hipMalloc / hipHostMalloc."""
import sys
import time
import torch

torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
keep = []
if mode == "busy":
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
for k in range(6):
    if mode == "pause":
        time.sleep(0.2)
    if mode == "busy":
        for _ in range(50):
            a.mul_(1.0001)
    t0 = time.perf_counter()
    keep.append(torch.empty(24 << 30, dtype=torch.uint8, device="cuda"))
    t1 = time.perf_counter()
    print(f"{mode}: hipMalloc 24 GB number {k}: {(t1 - t0) * 1e3:8.2f} ms", flush=True)
torch.cuda.synchronize()
if mode == "plain":
    for mb in (16, 128, 128, 1024):
        t0 = time.perf_counter()
        h = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
        t1 = time.perf_counter()
        keep.append(h)
        print(f"hipHostMalloc {mb:5d} MB: {(t1 - t0) * 1e3:8.2f} ms", flush=True)
