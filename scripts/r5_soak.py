#!/usr/bin/env python3
"""round 5: a soak of the batch entries -- N caller threads fill rotating batches (prepared and text, haploid and a bubble-rich index alternating on
the same device) for --seconds, and EVERY result set is compared with the first one its batch produced (records' filled counts and the sequence arena).
Looks for what only shows under concurrency: workspaces, the launch policy's shares (light / full walk kernel), events, result blocks."""
import argparse, hashlib, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--threads", type=int, default=6)
ap.add_argument("--nseq", type=int, default=120000)
ap.add_argument("--sites", type=int, default=20000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
work = []
for het in (0, 4):
    S = SynthSet(nseq=a.nseq, n_sites=min(a.nseq // (2 if het else 1), a.sites * 3), seed=3 + het, k=31, het_snps=het) if het else SynthSet(nseq=a.nseq, n_sites=a.sites * 3, seed=3, k=31)
    w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
    for b in range(3):
        gaps = []
        for i in range(b * a.sites, min((b + 1) * a.sites, S.n_sites)):
            l, r, _ = S.site(i)
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        if not gaps:
            continue
        work.append({"idx": idx, "prepared": idx.prepare_batch(mtg.Index.prepare_gaps(gaps), params), "text": mtg.TextGaps(gaps), "ref": {}, "n": len(gaps), "kind": "het" if het else "haploid"})
lock = threading.Lock()
stop = time.time() + a.seconds
calls = [0]
bad = []
def wk(t):
    torch.cuda.set_device(0)
    i = t
    while time.time() < stop and not bad:
        wb = work[i % len(work)]
        entry = "prepared" if (i // len(work)) % 2 == 0 else "text"
        h, nf, seqs = wb["idx"].fill_prepared(wb[entry], params, want_seqs=True)
        dig = hashlib.sha256(np.asarray(nf).tobytes() + seqs.tobytes()).hexdigest()
        wb["idx"].free_results(h)
        with lock:
            calls[0] += 1
            ref = wb["ref"].setdefault("d", dig)
            if ref != dig:
                bad.append((t, i, wb["kind"], entry))
        i += a.threads
ts = [threading.Thread(target=wk, args=(t,)) for t in range(a.threads)]
for t in ts: t.start()
for t in ts: t.join()
st = mtg.last_batch_stats()
print(json.dumps({"seconds": a.seconds, "threads": a.threads, "calls": calls[0], "batches": [(w["kind"], w["n"]) for w in work], "mismatches": bad[:5], "ok": not bad}))
sys.exit(1 if bad else 0)
