#!/usr/bin/env python3
"""diagnostic: traversal time against the number of gaps in the launch (one wave ... the whole batch), haploid or diploid (HET=1) bench set"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

het = 4 if os.environ.get("HET") else 0
S = SynthSet(nseq=600000, n_sites=100000, seed=1, k=31, het_snps=het)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
gaps = []
for i in range(100000):
    l, r, ins = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, "x", False)]))
for n in (64, 1024, 4096, 16384, 32768, 65536, 100000):
    p = mtg.Index.prepare_gaps(gaps[:n])
    best = None
    for rep in range(4):
        h, nf, _ = idx.fill_prepared(p, params, want_seqs=False)
        st = mtg.last_batch_stats()
        idx.free_results(h)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    print("gaps %6d waves %5d  k_stage_a %.3f  k_copy %.3f  k_post+scans %.3f  k_emit %.3f ms" % (n, (n + 63) // 64, best["kernel_ms"], best["copy_kernel_ms"], best["post_kernel_ms"], best["emit_kernel_ms"]))
