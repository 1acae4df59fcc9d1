#!/usr/bin/env python3
"""MTG_MAX_CHUNK=<n> python scripts/r4_chunk_diag.py [nloci]: the diploid test case through the tool with several launches per batch; which
records differ from the oracle's files"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet
from tests import oracle_lib
from tests.test_emu_parity import _write_idx
nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = SynthSet(nseq=2 * nloci, n_sites=nloci, seed=13, het_snps=4)
o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
km, ct = o.export()
d = pathlib.Path(tempfile.mkdtemp())
_write_idx(str(d / "d.mtgidx"), km, ct)
S.write_breakpoints(str(d / "d.breakpoints"))
o.fill_files("bkpt", str(d / "d.breakpoints"), str(d / "cpu"))
rc = mtg.Filler().run(["-graph", str(d / "d.mtgidx"), "-bkpt", str(d / "d.breakpoints"), "-out", str(d / "hip")])
print("exit code", rc, "stats", {k: v for k, v in mtg.last_batch_stats().items() if k in ("n_launches", "n_parked_gaps", "n_retried_gaps", "n_lean_gaps")})
def recs(p):
    out, name = {}, None
    for l in open(p):
        l = l.rstrip("\n")
        if l.startswith(">"): name = l; out.setdefault(name.split("_len_")[0], []).append([l, ""])
        else: out[name.split("_len_")[0]][-1][1] += l
    return out
a, b = recs(str(d / "cpu.insertions.fasta")), recs(str(d / "hip.insertions.fasta"))
bad = [k for k in a if a.get(k) != b.get(k)] + [k for k in b if k not in a]
print("records", len(a), len(b), "differing sites", len(bad))
for k in bad[:8]:
    print(" site", k)
    for tag, x in (("cpu", a.get(k)), ("hip", b.get(k))):
        print("   ", tag, [(h[h.find("_len_"):][:70], len(s)) for h, s in (x or [])])
