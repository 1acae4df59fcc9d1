#!/usr/bin/env python3
"""where a batch's time goes through mtg_fill_text (one batch alone, MTG_DEBUG_TIMERS=1 in the environment) next to the prepared entry"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet
S = SynthSet(nseq=600000, n_sites=100000, seed=1, k=31)
dev = torch.device("cuda", 0)
pw, po, pl, pn = S.packed()
w = torch.from_numpy(pw.view(np.int64)).to(dev); wo = torch.from_numpy(po.view(np.int64)).to(dev); ln = torch.from_numpy(pl.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), pn, S.total_kmers_upper_bound, 31, 3, 0)
del w, wo, ln
params = mtg.FillParams(max_nodes=100, max_depth=10000)
gaps = []
for i in range(100000):
    l, r, _ = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
text = mtg.TextGaps(gaps)
prep = idx.prepare_batch(mtg.Index.prepare_gaps(gaps), params)
for name, obj in (("prepared", prep), ("text", text), ("text", text), ("prepared", prep), ("text", text)):
    t0 = time.perf_counter()
    h, nf, _ = idx.fill_prepared(obj, params, want_seqs=False)
    el = time.perf_counter() - t0
    st = mtg.last_batch_stats(); idx.free_results(h)
    print("== %s: %.2f ms; h2d %.2f d2h %.2f host %.2f total %.2f" % (name, el * 1e3, st["h2d_ms"], st["d2h_ms"], st["host_ms"], st["total_ms"]), flush=True)
