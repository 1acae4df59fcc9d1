#!/usr/bin/env python3
"""round 5: the text entry (mtg_fill_text) next to the prepared one on the bench workload: ms per step with N caller threads, and the phases of one
call alone (the library's statistics).  python3 scripts/r5_text_entry.py [--in-flight 6 9 12]"""
import argparse, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

ap = argparse.ArgumentParser()
ap.add_argument("--nseq", type=int, default=600000)
ap.add_argument("--sites", type=int, default=100000)
ap.add_argument("--in-flight", type=int, nargs="+", default=[6, 9])
ap.add_argument("--steps", type=int, default=120)
ap.add_argument("--batches", type=int, default=4)
a = ap.parse_args()
S = SynthSet(nseq=a.nseq, n_sites=min(a.nseq, a.sites * a.batches), seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
B = []
for b in range(a.batches):
    gaps = []
    for i in range(b * a.sites, (b + 1) * a.sites):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    B.append({"prepared": idx.prepare_batch(mtg.Index.prepare_gaps(gaps), params), "text": mtg.TextGaps(gaps)})
out = {}
def measure(kind, nf):
    def run(count):
        it = iter(range(count)); lk = threading.Lock()
        def wk():
            torch.cuda.set_device(0)
            while True:
                with lk:
                    i = next(it, None)
                if i is None:
                    return
                h, _nf, _ = idx.fill_prepared(B[i % len(B)][kind], params, want_seqs=False)
                idx.free_results(h)
        ts = [threading.Thread(target=wk) for _ in range(nf)]
        for t in ts: t.start()
        for t in ts: t.join()
    run(3 * nf)
    reps = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(a.steps); torch.cuda.synchronize(); reps.append((time.perf_counter() - t0) / a.steps)
    h, _nf, _ = idx.fill_prepared(B[0][kind], params, want_seqs=False)
    st = mtg.last_batch_stats(); idx.free_results(h)
    return {"ms_per_step": round(float(np.median(reps)) * 1e3, 4), "M_sites_per_s": round(a.sites / float(np.median(reps)) / 1e6, 2),
            "one_call_alone_ms": {k2: round(st[k2], 3) for k2 in ("input_ms", "upload_ms", "device_span_ms", "d2h_ms", "host_ms", "total_ms") if k2 in st}}
for nf in a.in_flight:
    for kind in ("prepared", "text"):
        out["%s, %d in flight" % (kind, nf)] = measure(kind, nf)
        print(kind, nf, json.dumps(out["%s, %d in flight" % (kind, nf)]), flush=True)
for b in B:
    b["text"].register()
for nf in a.in_flight:
    out["registered text, %d in flight" % nf] = measure("text", nf)
    print("registered", nf, json.dumps(out["registered text, %d in flight" % nf]), flush=True)
print(json.dumps(out), flush=True)
