#!/usr/bin/env python3
"""diagnostic: which gaps of the diploid bench set walk slowly.  The batch is run in pieces of 64 gaps (one wave each), the slowest pieces gap by
gap; for the slow gaps the positions of the locus' SNPs relative to the site are printed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # first site looked at
het = 4 if os.environ.get("HET", "1") != "0" else 0  # HET=0: the haploid set
S = SynthSet(nseq=600000, n_sites=int(os.environ.get("NSITES", max(100000, first + n))), seed=1, k=31, het_snps=het)  # NSITES=400000: the site set of bench.py
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
gaps = []
for i in range(first, first + n):
    l, r, ins = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, "x", False)]))

def t_of(sub):
    p = mtg.Index.prepare_gaps(sub)
    best = 1e9
    for rep in range(2):
        h, nf, _ = idx.fill_prepared(p, params, want_seqs=False)
        st = mtg.last_batch_stats()
        idx.free_results(h)
        best = min(best, st["kernel_ms"])
    return best, st

times = np.array([t_of(gaps[a:a + 64])[0] for a in range(0, n, 64)])
med = float(np.median(times))
print("pieces of 64 gaps: median %.3f ms, max %.3f ms, > 3 x median: %d of %d" % (med, times.max(), int((times > 3 * med).sum()), len(times)))
nloci = S.nseq // 2
shown = 0
for pi in np.argsort(-times)[:6]:
    a = int(pi) * 64
    single = [(t_of(gaps[g:g + 1])[0], g) for g in range(a, min(a + 64, n))]
    single.sort(reverse=True)
    ms, g = single[0]
    st = t_of(gaps[g:g + 1])[1]
    gs = first + g  # site number
    snps = np.nonzero(S.codes(gs).astype(np.int16) != S.codes(gs + nloci).astype(np.int16))[0] if het else np.zeros(0, dtype=np.int64)
    p, L = int(S.pos[gs]), int(S.ins_len[gs])
    print("piece %d: %.3f ms; slowest gap %d alone: %.3f ms (next %.3f); site pos %d ins %d len %d; SNPs at %s (relative to the end of the insertion: %s); lines %d store_runs %d contig_nt %d"
          % (pi, times[pi], gs, ms, single[1][0], p, L, int(S.lens[gs]), snps.tolist(), (snps - (p + L)).tolist(), st["index_lines"], st["store_runs"], st["contig_nt"]))
