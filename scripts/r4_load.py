#!/usr/bin/env python3
"""round 4: the config-4 index through its container: built on the device, saved (container v3, 3.9 GB), loaded again -- the phases of the load
(words up, abundance bytes from the file to HBM in page-locked pieces while the tables are derived), fills against the truth."""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
NS = 20000
S = SynthSet(nseq=nseq, n_sites=NS, seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
del w, wo, ln
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
with tempfile.TemporaryDirectory(dir=base) as d:
    p = os.path.join(d, "h.mtgidx")
    t0 = time.time(); idx.save(p); t_save = time.time() - t0
    info = idx.info()
    idx.close()
    for rep in range(2):
        t0 = time.time(); g = mtg.Index.load(p); t_load = time.time() - t0
        prof = g.build_profile()
        print("load %d: %.2f s wall (container %.2f GB, saved in %.2f s)" % (rep, t_load, os.path.getsize(p) / 1e9, t_save))
        for ph in prof["phases"]:
            print("  %-26s %9.2f ms  %8.2f GB" % (ph["name"], ph["ms"], ph["bytes"] / 1e9))
        print("  library total %.2f s, peak %.1f GB" % (prof["total_ms"] / 1e3, prof["peak_device_bytes"] / 1e9), flush=True)
        if rep == 0:
            g.close()
    i2 = g.info()
    assert i2["nb_solid_kmers"] == info["nb_solid_kmers"] and i2["nb_unitigs"] == info["nb_unitigs"]
    gaps, truth = [], []
    for i in range(NS):
        l, r, ins = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, "x", False)])); truth.append(ins)
    res = g.fill_batch(gaps)
    assert [r["filled"][0]["seq"] for r in res] == truth
    print("fills on the loaded index identical to the truth:", len(res))
