#!/usr/bin/env python3
"""H2D copy rate from pinned host memory vs copy size and number of streams (torch; calibration for DESIGN.md)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (0.3, 0.92, 2.15, 8, 32, 128):
    n = int(mb * 1e6)
    cnt = max(8, min(512, int(1024e6 / n)))
    src = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(min(cnt, 16))]
    dst = torch.empty(cnt * n, dtype=torch.uint8, device=dev)
    for ns in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(cnt):
                with torch.cuda.stream(streams[i % ns]):
                    dst[i * n:(i + 1) * n].copy_(src[i % len(src)], non_blocking=True)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("size %7.2f MB x %3d, %d stream(s): %6.1f GB/s, %6.1f us per copy (host enqueue %5.1f us per copy)" % (mb, cnt, ns, cnt * n / dt / 1e9, dt / cnt * 1e6, t_enq / cnt * 1e6), flush=True)
