#!/usr/bin/env python3
"""human-scale index through its container: built on the device (3.1e9 k-mers), written from the device tables, loaded again piece by piece
straight from the file, and used for a batch of fills.  Needs ~40 GB of disk under $TMPDIR."""
import os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
d = tempfile.mkdtemp()
print("free disk under", d, ": %.1f GB" % (shutil.disk_usage(d).free / 1e9), flush=True)
S = SynthSet(nseq=nseq, n_sites=20000, seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
del w, wo, ln
torch.cuda.empty_cache()
info = idx.info()
print("built:", info["nb_solid_kmers"], "k-mers,", info["nb_unitigs"], "unitigs", flush=True)
p = os.path.join(d, "human.mtgidx")
t0 = time.time(); idx.save(p); t_save = time.time() - t0
size = os.path.getsize(p)
print("saved: %.1f GB in %.1f s (%.2f GB/s)" % (size / 1e9, t_save, size / 1e9 / t_save), flush=True)
rng = np.random.default_rng(3)
gaps, truth = [], []
for i in range(20000):
    l, r, ins = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, "x", False)])); truth.append(ins)
res0 = idx.fill_batch(gaps)
idx.close()
t0 = time.time(); g = mtg.Index.load(p); t_load = time.time() - t0
info2 = g.info()
print("loaded in %.1f s (%.2f GB/s): %d k-mers, %d unitigs" % (t_load, size / 1e9 / t_load, info2["nb_solid_kmers"], info2["nb_unitigs"]), flush=True)
assert info2["nb_solid_kmers"] == info["nb_solid_kmers"] and info2["nb_unitigs"] == info["nb_unitigs"]
res1 = g.fill_batch(gaps)
assert res0 == res1
assert [r["filled"][0]["seq"] for r in res1] == truth
print("fills on the loaded index identical to those on the built one and to the truth:", len(res1))
g.close()
os.remove(p); os.rmdir(d)
