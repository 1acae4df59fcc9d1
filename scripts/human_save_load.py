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
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 20000  # sites (the tool run at the end fills all of them)
S = SynthSet(nseq=nseq, n_sites=NS, seed=1, k=31)
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
for i in range(min(NS, 20000)):
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
# the tool itself on the saved index: MindTheGap fill -graph ... -bkpt ... (load, fill of all sites in batches, FASTA / VCF / info files)
bk = os.path.join(d, "s.breakpoints")
S.write_breakpoints(bk)
t0 = time.time()
rc = mtg.fill_main(["-graph", p, "-bkpt", bk, "-out", os.path.join(d, "tool")])
t_tool = time.time() - t0
seqs = [l.rstrip("\n") for l in open(os.path.join(d, "tool.insertions.fasta")) if not l.startswith(">")]
print("MindTheGap fill -graph (36 GB container) -bkpt (%d sites): %.1f s in all, rc %d; %d sequences, all equal to the inserted ones: %s; output %.0f MB"
      % (S.n_sites, t_tool, rc, len(seqs), seqs == [S.site(i)[2] for i in range(S.n_sites)], sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("tool.")) / 1e6), flush=True)
for fn in os.listdir(d):
    os.remove(os.path.join(d, fn))
os.rmdir(d)
