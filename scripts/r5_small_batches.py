#!/usr/bin/env python3
"""round 5: what a SMALL batch costs (BASELINE config 5 literally: 100 000 sites over 8 ranks = 12 500 per rank and step).  Prepared batches
of --sites sites, --in-flight caller threads; prints ms per step, and with MTG_DEBUG_TIMERS=1 in the environment the library's own phases."""
import argparse, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

ap = argparse.ArgumentParser()
ap.add_argument("--nseq", type=int, default=100000)
ap.add_argument("--sites", type=int, nargs="+", default=[12500, 100000])
ap.add_argument("--in-flight", type=int, nargs="+", default=[6, 1])
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
S = SynthSet(nseq=a.nseq, n_sites=min(a.nseq, max(a.sites)), seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
out = {}
for ns in a.sites:
    ns = min(ns, S.n_sites)
    gaps = []
    for i in range(ns):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    batch = idx.prepare_batch(mtg.Index.prepare_gaps(gaps), params)
    for nf in a.in_flight:
        def run(count):
            it = iter(range(count)); lk = threading.Lock()
            def wk():
                torch.cuda.set_device(0)
                while True:
                    with lk:
                        if next(it, None) is None:
                            return
                    h, _nf, _ = idx.fill_prepared(batch, params, want_seqs=False)
                    idx.free_results(h)
            ts = [threading.Thread(target=wk) for _ in range(nf)]
            for t in ts: t.start()
            for t in ts: t.join()
        run(24)
        reps = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(a.steps); torch.cuda.synchronize(); reps.append((time.perf_counter() - t0) / a.steps)
        mtg.tuning_set("KERNEL_TIMERS", "1")
        h, _nf, _ = idx.fill_prepared(batch, params, want_seqs=False)
        st = mtg.last_batch_stats(); idx.free_results(h)
        mtg.tuning_set("KERNEL_TIMERS", None)
        out["%d sites, %d in flight" % (ns, nf)] = {"ms_per_step": round(float(np.median(reps)) * 1e3, 4), "M_sites_per_s": round(ns / float(np.median(reps)) / 1e6, 2),
                                                   "alone_ms": {k2: round(st[k2], 4) for k2 in ("kernel_ms", "finish_kernel_ms", "lean_kernel_ms", "copy_kernel_ms", "post_kernel_ms", "emit_kernel_ms", "device_span_ms", "d2h_ms", "host_ms", "total_ms") if k2 in st}}
        print(json.dumps({k2: v for k2, v in out.items() if k2.startswith("%d sites, %d" % (ns, nf))}), flush=True)
    batch.close()
print(json.dumps(out), flush=True)
