#!/usr/bin/env python3
"""Do host<->device copies overlap with compute on this box?  A long kernel sequence on one stream, a 60 MB pinned
H2D + 42 MB D2H on another: together vs alone."""
import time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.float32)
h = torch.empty(60_000_000, dtype=torch.uint8).pin_memory()
h2 = torch.empty(42_000_000, dtype=torch.uint8).pin_memory()
d = torch.empty(60_000_000, dtype=torch.uint8, device=dev)
d2 = torch.empty(42_000_000, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def compute():
    with torch.cuda.stream(s1):
        for _ in range(12):
            torch.mm(a, a)
def copies():
    with torch.cuda.stream(s2):
        d.copy_(h, non_blocking=True)
        h2.copy_(d2, non_blocking=True)
def t(fn):
    torch.cuda.synchronize(dev); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev); return (time.perf_counter() - t0) * 1e3
for _ in range(3):
    compute(); copies()
print("compute alone %.2f ms, copies alone %.2f ms, both %.2f ms" % (min(t(compute) for _ in range(5)), min(t(copies) for _ in range(5)),
      min(t(lambda: (compute(), copies())) for _ in range(5))))
