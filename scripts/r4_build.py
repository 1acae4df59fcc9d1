#!/usr/bin/env python3
"""round 4: the construction of the config-4 index (3 Gbp donor, 3.1e9 k-mers) -- phases, device times, peak device memory -- with the lean
build (junction table) and, with MTG_LEGACY_BUILD=1 in the environment, the build of rounds 1-3; then a batch of fills against the truth."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
t0 = time.time()
S = SynthSet(nseq=nseq, n_sites=NS, seed=1, k=31)
print("donor generated in %.1f s" % (time.time() - t0), flush=True)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
t0 = time.time()
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
wall = time.time() - t0
info = idx.info()
prof = idx.build_profile()
out = {"mode": "legacy" if os.environ.get("MTG_LEGACY_BUILD") else "lean", "nseq": nseq, "wall_s": wall, "info": info, "profile": prof}
print(json.dumps(out), flush=True)
for ph in prof["phases"]:
    print("  %-24s %9.2f ms  %8.1f GB  %6.2f TB/s  units %d" % (ph["name"], ph["ms"], ph["bytes"] / 1e9, ph["bytes"] / 1e9 / max(ph["ms"], 1e-9), ph["units"]))
print("  total %.2f s, peak %.1f GB, resident %.1f GB" % (prof["total_ms"] / 1e3, prof["peak_device_bytes"] / 1e9, info["device_bytes"] / 1e9), flush=True)
gaps, truth = [], []
for i in range(min(NS, 20000)):
    l, r, ins = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, "x", False)])); truth.append(ins)
res = idx.fill_batch(gaps)
ok = [r["filled"][0]["seq"] if r["filled"] else None for r in res] == truth
print("fills identical to the truth:", ok, len(res), flush=True)
assert ok
