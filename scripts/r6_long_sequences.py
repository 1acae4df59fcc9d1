#!/usr/bin/env python3
"""round 6: the index construction from a donor given as a FEW LONG sequences (24 x 125 Mbp, i.i.d.) -- as the entry takes them, in pieces of 65 536
nucleotides that share k - 1 (the default), and with NO_SPLIT_LONG=1 (a workgroup a sequence: 24 of the device's workgroups have work)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
wps = (L + 31) // 32 + 1
w = torch.randint(-2**63, 2**63 - 1, (nseq * wps,), dtype=torch.int64, device=dev, generator=g)
if len(sys.argv) > 3 and sys.argv[3] == "repeats":  # the same 64 nucleotides every 313 words (10 kb): the chains end there, as they do at the repeats of a real genome
    v = w.view(nseq, wps)
    v[:, 5::313] = 0x1B27E4D8C6A59F03
    v[:, 6::313] = 0x5AC3F10E9B7D2648
wo = (torch.arange(nseq, dtype=torch.int64, device=dev) * wps)
ln = torch.full((nseq,), L, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for whole in ("1", None):  # (the unsplit form first: 21 s)
    mtg.tuning_set("NO_SPLIT_LONG", whole)
    t0 = time.time()
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), nseq, nseq * (L - 30), 31, 3, 0)
    wall = time.time() - t0
    mtg.tuning_set("NO_SPLIT_LONG", None)
    info, prof = idx.info(), idx.build_profile()
    ph = {}
    for p in prof["phases"]:
        ph[p["name"]] = ph.get(p["name"], 0.0) + p["ms"]
    print("NO_SPLIT_LONG=%s: wall %.2f s, device %.3f s; solid %d, unitigs %d; phases %s" % (whole, wall, prof["total_ms"] / 1e3, info["nb_solid_kmers"], info["nb_unitigs"],
          {k: round(v, 1) for k, v in ph.items() if v >= 1.0}), flush=True)
    idx.close()
