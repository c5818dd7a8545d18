#!/usr/bin/env python3
"""HBM ceilings on the GPU box: pure write (fill), copy (read+write) -- to place the engine's
write-dominated stream (2041 of 2053 algorithmic bytes per env-step are writes) on the right roofline."""
import torch
import time
dev = "cuda"
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, bytes_moved in (("fill (write only)", lambda: x.fill_(1.0), n * 4),
                                  ("zero_ (write only)", lambda: x.zero_(), n * 4),
                                  ("copy (read+write)", lambda: y.copy_(x), n * 8)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%5d MB %-20s %8.3f ms  %7.1f GB/s" % (mb, name, ms, bytes_moved / ms / 1e6), flush=True)
