#!/usr/bin/env python3
"""TEST-ONLY: wall time of the host-side input passes of mtg_fill_batch (sizes + layout, input set, marshal) on this machine's CPU,
with the device part replaced by the emulation build (tests/emu).  Only the phases before the device work are representative: the
emulator hands results back gap by gap.   MTG_DEBUG_TIMERS=1 python scripts/host_phases_emu.py [n_gaps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTG_DEBUG_TIMERS", "1")
import numpy as np  # noqa: E402

from tests import emu_lib  # noqa: E402

mtg = emu_lib.product_on_emulator()
from mindthegap_amd.synth import SynthSet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = SynthSet(nseq=400, n_sites=400, seq_len=2100, seed=1, ins_min=int(os.environ.get("INS_MIN", "40")), ins_max=int(os.environ.get("INS_MAX", "60")), k=31)
seqs = [S.ascii(j) for j in range(S.nseq)]
from tests import oracle_lib  # noqa: E402  (k-mer extraction only)
o = oracle_lib.Index.from_sequences(seqs, 31, 3, 40)
km, ab = o.export()
idx = mtg.Index.from_kmers(km, ab, 31)
gaps = []
for i in range(n):
    l, r, _ = S.site(i % 400)
    gaps.append(mtg.Gap(l, r, [(r, S.site_name(i % 400), False)]))
prep = mtg.Index.prepare_gaps(gaps)
for _ in range(3):
    h, nf, _ = idx.fill_prepared(prep, mtg.FillParams(max_nodes=100, max_depth=10000), want_seqs=False)
    idx.free_results(h)
print("filled", int((nf > 0).sum()), "of", n)
