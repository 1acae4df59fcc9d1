"""soak: many batches on one index (alternating sizes; host strings, prepared batches and text blocks; plain, serialised, device-resident and relocatable results); prints host
RSS and device memory at intervals"""
import os, sys, time, resource
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
preps = [mtg.Index.prepare_gaps(gaps[:n]) for n in (100000, 1000, 37000, 64, 100000)]
sizes = [100000, 1000, 37000, 64, 100000]
out = np.empty(80 << 20, dtype=np.uint8)
dbuf = torch.empty(80 << 20, dtype=torch.uint8, device=dev)
batches = [idx.prepare_batch(p) for p in preps]  # prepared (device-resident) forms of the same batches
texts = [mtg.TextGaps(gaps[:n]) for n in sizes]    # the same batches as blocks of text (mtg_fill_text)
def rss():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20
t0 = time.time()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2000):
    j = it % len(preps)
    if it % 11 == 0:  # one block of text, marshalled on the device
        h, nf, _ = idx.fill_prepared(texts[j], want_seqs=False)
    elif it % 13 == 0:  # the relocatable form, written by the result kernel into a device buffer
        h, nf, nb = idx.fill_prepared_wire_device(batches[j], it, dbuf.data_ptr(), dbuf.numel())
    elif it % 3 == 0:
        h, nf, nb = idx.fill_prepared_serial(preps[j], out)
    elif it % 5 == 0:
        h, nf, nb = idx.fill_prepared_serial_device(batches[j], dbuf.data_ptr(), dbuf.numel())
    elif it % 5 == 1:
        h, nf, _ = idx.fill_prepared(batches[j], want_seqs=False)
    else:
        h, nf, _ = idx.fill_prepared(preps[j], want_seqs=(it % 7 == 0))
    assert int((nf > 0).sum()) == sizes[j]
    idx.free_results(h)
    if it % 400 == 0:
        free, total = torch.cuda.mem_get_info()
        print("iter %5d  %.1f s  host RSS %.0f MB  device used %.2f GB" % (it, time.time() - t0, rss(), (total - free) / 2**30), flush=True)
print("done", time.time() - t0)
