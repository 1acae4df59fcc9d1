#!/usr/bin/env python3
"""diagnostic: one site of the bench's site set alone on the device (with a -DMTG_STAMPS build the library prints the walk's per-phase times)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

sites = [int(x) for x in sys.argv[1:]] or [118834]
S = SynthSet(nseq=600000, n_sites=int(os.environ.get("NSITES", 400000)), seed=1, k=31, het_snps=4 if os.environ.get("HET") else 0)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
for g in sites:
    l, r, ins = S.site(g)
    gaps = [mtg.Gap(l, r, [(r, "x", False)])]
    for rep in range(3):
        res = idx.fill_batch(gaps, params)
        st = mtg.last_batch_stats()
    c = idx.stage_a([l], [r])[0]
    print("site", g, "k_stage_a %.3f ms" % st["kernel_ms"], "lines", st["index_lines"], "store reads", st["store_runs"], "contigs", [len(x) for x in c], "nb_nodes", res[0]["nb_nodes"], "filled", len(res[0]["filled"]))
