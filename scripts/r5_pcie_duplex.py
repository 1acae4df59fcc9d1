#!/usr/bin/env python3
"""round 5: is the link full duplex for this runtime's copies?  D2H copies of 41 MB (a batch's results) on three streams alone, then with H2D copies of
13 MB (a batch's text block) on three other streams at the same time.  GB/s of each direction."""
import time, json
import torch
dev = torch.device("cuda", 0)
D = [torch.empty(41 << 20, dtype=torch.uint8, device=dev) for _ in range(3)]
H = [torch.empty(41 << 20, dtype=torch.uint8).pin_memory() for _ in range(3)]
U = [torch.empty(13 << 20, dtype=torch.uint8).pin_memory() for _ in range(3)]
V = [torch.empty(13 << 20, dtype=torch.uint8, device=dev) for _ in range(3)]
sd = [torch.cuda.Stream() for _ in range(3)]
su = [torch.cuda.Stream() for _ in range(3)]
def run(n_down, n_up, reps=60):
    """3 copies down and 3 up per repetition; the uploads on n_up streams (1: one at a time, what the library does; 3: each on its own)"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(reps):
        for i in range(3):
            if n_down:
                with torch.cuda.stream(sd[i]): H[i].copy_(D[i], non_blocking=True)
            if n_up:
                with torch.cuda.stream(su[i % n_up]): V[i].copy_(U[i], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return {"seconds": round(dt, 4), "d2h_GBps": round(3 * reps * (41 << 20) / dt / 1e9, 2) if n_down else 0, "h2d_GBps": round(3 * reps * (13 << 20) / dt / 1e9, 2) if n_up else 0}
run(1, 1, 5)
out = {"d2h_alone": run(1, 0), "h2d_alone_one_stream": run(0, 1), "h2d_alone_three_streams": run(0, 3), "both_uploads_on_one_stream": run(1, 1), "both_uploads_on_three_streams": run(1, 3)}
print(json.dumps(out))
