#!/usr/bin/env python3
"""is the link full duplex for this library's copies?  device-to-host (a batch's results, 40 MB) and host-to-device (a batch's text, 16 MB)
copies between page-locked and device memory, each direction alone and both at once on separate streams"""
import time, torch
dev = torch.device("cuda", 0)
D2H, H2D, IT = 40 << 20, 16 << 20, 40
def bufs(n, ns): return [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(ns)], [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(ns)]
for ns, nu in ((1, 1), (2, 1), (3, 1), (3, 3), (6, 6)):
    dd, dh = bufs(D2H, ns)
    ud, uh = bufs(H2D, nu)
    sd = [torch.cuda.Stream() for _ in range(ns)]
    su = [torch.cuda.Stream() for _ in range(nu)]
    def run(down, up):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for it in range(IT):
                for i in range(max(ns, nu)):
                    if down:
                        with torch.cuda.stream(sd[i % ns]): dh[i % ns].copy_(dd[i % ns], non_blocking=True)
                    if up:
                        with torch.cuda.stream(su[i % nu]): ud[i % nu].copy_(uh[i % nu], non_blocking=True)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        return el
    e1, e2, e3 = run(True, False), run(False, True), run(True, True)
    m = max(ns, nu)
    print("%d D2H stream(s), %d H2D stream(s): D2H alone %.1f GB/s; H2D alone %.1f GB/s; both at once: D2H %.1f + H2D %.1f GB/s (%.2f ms per 40 MB + 16 MB pair; alone %.2f and %.2f ms)"
          % (ns, nu, IT * m * D2H / e1 / 1e9, IT * m * H2D / e2 / 1e9, IT * m * D2H / e3 / 1e9, IT * m * H2D / e3 / 1e9, e3 / IT / m * 1e3, e1 / IT / m * 1e3, e2 / IT / m * 1e3))
