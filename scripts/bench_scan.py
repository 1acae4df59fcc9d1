"""Throughput of the sequence-scan kernel (k_scan: rolling k-mer + LDS-staged minimizer-blocked Bloom + exact confirmation)
on the synthetic human-scale donor: every position of the donor (all members) and of an unrelated random genome (no members)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
mtg.load_library()
dev = torch.device("cuda", 0)
torch.cuda.init()
S = SynthSet(nseq=nseq, n_sites=min(100000, nseq), seed=1)
w = torch.from_numpy(S.words.view(np.int64)).to(dev)
wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev)
ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 40)
info = idx.info()
out = torch.zeros_like(w)
res = {"nb_solid_kmers": int(info["nb_solid_kmers"]), "bloom_bytes": int(info["bloom_blocks"]) * 64, "bloom_minimizer": int(info["bloom_minimizer"])}
for name, words in (("members", w), ("non_members", torch.from_numpy(np.random.default_rng(99).integers(0, 2**63, size=S.words.shape, dtype=np.int64)).to(dev))):
    for exact in (False, True):
        st = idx.scan_packed_device(words.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, out.data_ptr(), exact=exact)  # warm-up
        st = idx.scan_packed_device(words.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, out.data_ptr(), exact=exact)
        nk = st["n_kmers"]
        hbm = st["blocks_staged"] * 64 + (st["bloom_positive"] * info["abnd_bucket_bytes"] if exact else 0) + nk / 4 + nk / 8
        res["%s_%s" % (name, "exact" if exact else "bloom")] = {
            "kmers": nk, "kernel_ms": st["kernel_ms"], "Gkmers_per_s": nk / st["kernel_ms"] / 1e6, "bloom_positive": st["bloom_positive"], "confirmed": st["confirmed"],
            "blocks_staged": st["blocks_staged"], "kmers_per_block": nk / max(st["blocks_staged"], 1), "est_hbm_GBps": hbm / st["kernel_ms"] / 1e6,
            "naive_64B_per_kmer_GBps_equivalent": nk * 64 / st["kernel_ms"] / 1e6}
print(json.dumps(res, indent=1))
