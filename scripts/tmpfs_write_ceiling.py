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
# several files, one writer thread each (what the tool does since round 4: a file's lock does not serialise writes to another file)
for nf in (2, 3):
    files = [tempfile.NamedTemporaryFile(dir=base) for _ in range(nf)]
    per = total // nf // len(piece)
    def one(fd):
        for i in range(per):
            os.pwrite(fd, piece, i * len(piece))
    ts = [threading.Thread(target=one, args=(f.fileno(),)) for f in files]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    el = time.perf_counter() - t0
    print("%d files, one writer thread each: %.2f GB/s in all (%.1f GB in %.2f s)" % (nf, nf * per * len(piece) / el / 1e9, nf * per * len(piece) / 1e9, el), flush=True)
    for f in files: f.close()
