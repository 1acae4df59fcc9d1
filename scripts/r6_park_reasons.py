"""round 6: which nodes park the gaps of the indel set?  A scaled-down human-indel set on the TEST-ONLY emulation of the product
(MTG_EMU_PARK_STATS=1 makes the walk kernel's park site tally the shape of the node: mtg_traverse.h, emu_park_note)."""
import os
import sys
os.environ["MTG_EMU_PARK_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mindthegap_amd.synth import SynthSet
from tests import emu_lib, oracle_lib

nloci = int(sys.argv[1]) if len(sys.argv) > 1 else 600
mtg = emu_lib.product_on_emulator()
S = SynthSet(nseq=2 * nloci, n_sites=nloci, seed=1, k=31, het_snps=4, het_indels=2)
o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 0)
km, ct = o.export()
idx = mtg.Index.from_kmers(km, ct, 31)
gaps = []
for i in range(nloci):
    l, r, _ = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
res = idx.fill_batch(gaps, mtg.FillParams(max_nodes=100, max_depth=10000))
st = mtg.last_batch_stats()
print("sites", nloci, "filled", sum(1 for r in res if r["filled"]), "parked", st.get("n_parked_gaps"))
