#!/usr/bin/env python3
"""the write ceiling the tool's output meets: N threads pwrite 8 MB pieces of page-locked-like memory into one file on the memory-backed file
system (/dev/shm), as the tool's writer threads do; GB/s by number of threads, under the CPU quota of the container."""
import os, sys, threading, time, tempfile
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
piece = bytes(bytearray(os.urandom(1 << 20)) * 8)
total = 2 << 30
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", " file system:", base or tempfile.gettempdir())
for nt in (1, 2, 4, 8, 12, 16, 24):
    with tempfile.NamedTemporaryFile(dir=base) as f:
        fd = f.fileno()
        npieces = total // len(piece)
        nxt = [0]
        lock = threading.Lock()
        def work():
            while True:
                with lock:
                    i = nxt[0]; nxt[0] += 1
                if i >= npieces:
                    return
                os.pwrite(fd, piece, i * len(piece))
        ts = [threading.Thread(target=work) for _ in range(nt)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        el = time.perf_counter() - t0
        print("%2d writer threads: %.2f GB/s (%.1f GB in %.2f s)" % (nt, total / el / 1e9, total / 1e9, el), flush=True)
