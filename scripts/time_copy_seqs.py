"""cost of serialising a batch's sequences (mtg_results_copy_seqs) into pageable and page-locked memory, human workload"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet
S = SynthSet(nseq=100000, n_sites=100000, seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 40)
gaps = []
for i in range(S.n_sites):
    l, r, _ = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
prep = mtg.Index.prepare_gaps(gaps)
pin = torch.empty(80 << 20, dtype=torch.uint8).pin_memory().numpy()
for mode in ("none", "pageable", "pinned", "none", "pinned"):
    ts = []
    for it in range(12):
        t0 = time.perf_counter()
        if mode == "none":
            h, nf, _ = idx.fill_prepared(prep, want_seqs=False)
        elif mode == "pageable":
            h, nf, s = idx.fill_prepared(prep, want_seqs=True)
        else:
            h, nf, s = idx.fill_prepared(prep, want_seqs=True, out=pin)
        idx.free_results(h)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "ms/step: median %.2f min %.2f" % (sorted(ts)[len(ts) // 2], min(ts)))
ts = []
for it in range(12):
    t0 = time.perf_counter()
    h, nf, nb = idx.fill_prepared_serial(prep, pin)
    idx.free_results(h)
    ts.append((time.perf_counter() - t0) * 1e3)
print("serial (in place, pinned)", "ms/step: median %.2f min %.2f" % (sorted(ts)[len(ts) // 2], min(ts)))
