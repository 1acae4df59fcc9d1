#!/usr/bin/env python3
"""diagnostic: the gaps of the diploid bench set whose walk crosses two SNPs exactly k apart (the last nodes of the first bubble's branches both lead
to both alleles of the second: four paths of 2k + 1 nodes, the general bubble code), as one batch; with a -DMTG_STAMPS build the per-phase
times of exactly these walks are printed by the library."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

S = SynthSet(nseq=600000, n_sites=100000, seed=1, k=31, het_snps=4)
nloci = S.nseq // 2
x = S.words[:100000] ^ S.words[nloci:nloci + 100000]
rows, wi = np.nonzero(x)
sel = []
by = {}
for r, w_ in zip(rows.tolist(), wi.tolist()):
    v = int(x[r, w_])
    for b in range(32):
        if (v >> (2 * b)) & 3:
            by.setdefault(r, []).append(w_ * 32 + b)
for r, ps in by.items():
    ps.sort()
    right = [p for p in ps if p > int(S.pos[r])]
    if len(right) == 2 and right[1] - right[0] == 31:
        sel.append(r)
print("gaps whose two SNPs behind the site are exactly k apart:", len(sel))
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
gaps = []
for i in sel[:64]:
    l, r, ins = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, "x", False)]))
p = mtg.Index.prepare_gaps(gaps)
for rep in range(3):
    h, nf, _ = idx.fill_prepared(p, params, want_seqs=False)
    st = mtg.last_batch_stats()
    idx.free_results(h)
print("one wave of them: k_stage_a %.3f ms" % st["kernel_ms"], "filled", int((nf > 0).sum()), "of", len(gaps))
