#!/usr/bin/env python3
"""the link the job is bound by: device-to-host copies into page-locked memory, sizes of a batch's result arrays, 1..6 streams at once"""
import time, torch
dev = torch.device("cuda", 0)
for mb in (4, 10, 32, 128):
    n = mb << 20
    for ns in (1, 2, 6):
        src = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(ns)]
        dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(ns)]
        st = [torch.cuda.Stream() for _ in range(ns)]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(20):
                for i in range(ns):
                    with torch.cuda.stream(st[i]):
                        dst[i].copy_(src[i], non_blocking=True)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("D2H %4d MB x %d streams: %.1f GB/s" % (mb, ns, 20 * ns * n / el / 1e9))
