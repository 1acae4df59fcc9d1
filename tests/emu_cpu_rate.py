"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): the product's own algorithm -- the device code compiled for the host, tests/emu --
timed on the host cores, as the `same_algorithm_value` next to the oracle's number.  The oracle restates the REFERENCE algorithm (hash probes
per nucleotide); this is what a CPU does with the unitig-store walk the GPU runs, so the GPU/CPU ratio of the kernels can be read without the
algorithmic gain mixed in.

usage: python tests/emu_cpu_rate.py <kmers.npy> <counts.npy> <gaps.json> <threads>
prints one JSON line {"sites": n, "seconds": s, "threads": t, "sha256": digest of the sorted filled sequences}"""
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    km, ct, gaps_json, nth = np.load(sys.argv[1]), np.load(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    os.environ.setdefault("MTG_POOL_THREADS", "1")  # the slices are the parallelism
    os.environ.pop("MTG_ROUNDS", None)
    os.environ["MTG_EMU_NO_POISON"] = "1"  # the emulator's read-before-write check off: it costs more than the walk
    from tests import emu_lib
    mtg = emu_lib.product_on_emulator()
    idx = mtg.Index.from_kmers(km, ct, 31)
    sites = json.load(open(gaps_json))
    gaps = [mtg.Gap(l, r, [(r, name, False)]) for (l, r, name) in sites]
    n = len(gaps)
    nth = max(1, min(nth, n))
    parts = [gaps[i * n // nth:(i + 1) * n // nth] for i in range(nth)]
    prepared = [mtg.Index.prepare_gaps(p) for p in parts]  # ctypes arrays built outside the timed region
    out = [None] * nth

    def work(i):  # the C call of the slice (ctypes releases the interpreter lock) and the copy of its sequences
        out[i] = idx.fill_prepared(prepared[i])

    def run():
        th = [threading.Thread(target=work, args=(i,)) for i in range(nth)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    run()  # untimed: every workspace allocates its scratch on first use
    for (h, _nf, _buf) in out:
        idx.free_results(h)
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    seqs = sorted(l for (_h, _nf, buf) in out for l in buf.tobytes().decode().split("\n") if l)
    print(json.dumps({"sites": n, "seconds": dt, "threads": nth, "sha256": hashlib.sha256("\n".join(seqs).encode()).hexdigest()}))


if __name__ == "__main__":
    main()
