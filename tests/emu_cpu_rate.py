"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): the product's own algorithm -- the device code compiled for the host, tests/emu,
WITHOUT the emulation build's cross-checks (-DMTG_NO_XCHECK) -- timed on the host cores, as the `same_algorithm_value` next to the oracle's
number.  The oracle restates the REFERENCE algorithm (hash probes per nucleotide); this is what a CPU does with the unitig-store walk the GPU
runs, so the GPU/CPU ratio of the kernels can be read without the algorithmic gain mixed in.

usage: python tests/emu_cpu_rate.py <kmers.npy> <counts.npy> <gaps.json> <processes>   (one process per core: each with its own copy of the index and a slice of the sites)
prints one JSON line {"sites": n, "seconds": s, "processes": p, "sha256": digest of the sorted filled sequences}"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAST_SO = os.path.join(ROOT, "tests", "emu", "libmtgfill_emu_fast.so")


def build_fast():
    """tests/emu/libmtgfill_emu_fast.so: the emulation backend + the product's host code, cross-checks compiled out, -O3"""
    from tests import emu_lib
    csrc = os.path.join(ROOT, "mindthegap_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "emu", "emu_backend.cpp"), os.path.join(csrc, "mtg_host.cpp"), os.path.join(csrc, "mtg_cli.cpp")]
    deps = srcs + emu_lib.HDRS + [os.path.join(csrc, "mtg_internal.h"), os.path.join(ROOT, "include", "mtg_fill.h")]
    if not os.path.exists(FAST_SO) or any(os.path.getmtime(d) > os.path.getmtime(FAST_SO) for d in deps):
        tmp = FAST_SO + ".%d.tmp" % os.getpid()
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DMTG_NO_XCHECK", "-o", tmp] + srcs + ["-lz"])
        os.replace(tmp, FAST_SO)
    return FAST_SO


def child(km_path, ct_path, gaps_json, i, n_proc, sync_dir):
    """one process of the measurement: its own copy of the index, slice i of the sites; waits for the others before the timed pass"""
    km, ct = np.load(km_path), np.load(ct_path)
    os.environ["MTG_POOL_THREADS"] = "1"  # the processes are the parallelism
    os.environ.pop("MTG_ROUNDS", None)
    os.environ["MTG_EMU_NO_POISON"] = "1"  # the emulator's read-before-write check off
    import mindthegap_amd as mtg
    from mindthegap_amd import lib as L
    L._lib = L._bind(C.CDLL(FAST_SO))
    idx = mtg.Index.from_kmers(km, ct, 31)
    sites = json.load(open(gaps_json))
    n = len(sites)
    mine = sites[i * n // n_proc:(i + 1) * n // n_proc]
    reps = max(1, -(-3000 // max(len(mine), 1)))  # a call long enough that its fixed cost (a few ms of allocations) does not set the rate: the slice several times over
    prepared = mtg.Index.prepare_gaps([mtg.Gap(l, r, [(r, name, False)]) for (l, r, name) in mine] * reps)  # ctypes arrays built outside the timed region
    h, _nf, _buf = idx.fill_prepared(prepared)  # untimed: scratch allocated, pages touched
    idx.free_results(h)
    open(os.path.join(sync_dir, "ready.%d" % i), "w").close()
    while not os.path.exists(os.path.join(sync_dir, "go")):
        time.sleep(0.002)
    t0 = time.time()
    h, _nf, buf = idx.fill_prepared(prepared)
    t1 = time.time()
    seqs = [l for l in buf.tobytes().decode().split("\n") if l]
    if seqs != seqs[: len(seqs) // reps] * reps:
        sys.exit("repetitions differ")
    seqs = seqs[: len(seqs) // reps]
    json.dump({"t0": t0, "t1": t1, "sites": len(mine) * reps, "distinct": len(mine), "seqs": seqs}, open(os.path.join(sync_dir, "out.%d" % i), "w"))


def main():
    if len(sys.argv) > 5:
        return child(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[5]), int(sys.argv[4]), sys.argv[6])
    import tempfile
    km_path, ct_path, gaps_json, n_proc = sys.argv[1], sys.argv[2], sys.argv[3], max(1, int(sys.argv[4]))
    build_fast()
    with tempfile.TemporaryDirectory() as d:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), km_path, ct_path, gaps_json, str(n_proc), str(i), d]) for i in range(n_proc)]
        while not all(os.path.exists(os.path.join(d, "ready.%d" % i)) for i in range(n_proc)):
            if any(p.poll() not in (None, 0) for p in ps):
                sys.exit("a worker failed")
            time.sleep(0.01)
        open(os.path.join(d, "go"), "w").close()
        for p in ps:
            if p.wait() != 0:
                sys.exit("a worker failed")
        outs = [json.load(open(os.path.join(d, "out.%d" % i))) for i in range(n_proc)]
    dt = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
    seqs = sorted(l for o in outs for l in o["seqs"])
    print(json.dumps({"sites": sum(o["sites"] for o in outs), "distinct_sites": sum(o["distinct"] for o in outs), "seconds": dt, "processes": n_proc, "sha256": hashlib.sha256("\n".join(seqs).encode()).hexdigest()}))


if __name__ == "__main__":
    main()
