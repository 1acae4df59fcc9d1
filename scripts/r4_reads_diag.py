#!/usr/bin/env python3
"""where a batch's time goes on the reads-built graph (scripts/r4_reads_workload.py): batch statistics and the library's debug timers of one batch"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet, NT
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = SynthSet(nseq=nseq, n_sites=nseq, seed=2, k=31)
d = tempfile.mkdtemp(); reads = os.path.join(d, "reads.fasta")
comp = np.array([2, 3, 0, 1], dtype=np.uint8)
with open(reads, "wb") as f:
    for j in range(S.nseq):
        rng = np.random.default_rng(1000003 * 7 + j)
        c = S.codes(j); L = len(c); nr = int(round(30 * L / 150))
        st = rng.integers(0, L - 150 + 1, nr)
        m = c[st[:, None] + np.arange(150)[None, :]]
        rev = rng.integers(0, 2, nr).astype(bool)
        m[rev] = comp[m[rev][:, ::-1]]
        e = rng.random(m.shape) < 0.005
        m[e] = (m[e] + rng.integers(1, 4, int(e.sum())).astype(np.uint8)) & 3
        out = np.empty((nr, 154), dtype=np.uint8)
        out[:, 0] = ord(">"); out[:, 1] = ord("r"); out[:, 2] = 10; out[:, 153] = 10
        out[:, 3:153] = NT[m]
        f.write(out.tobytes())
idx = mtg.Index.from_reads([reads], 31, 3, 0)
os.remove(reads); os.rmdir(d)
params = mtg.FillParams(max_nodes=100, max_depth=10000)
gaps = []
for i in range(S.n_sites):
    l, r, _ = S.site(i)
    gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
pb = idx.prepare_batch(mtg.Index.prepare_gaps(gaps), params)
for rep in range(3):
    t0 = time.perf_counter()
    h, nf, _ = idx.fill_prepared(pb, params, want_seqs=False)
    el = time.perf_counter() - t0
    st = mtg.last_batch_stats(); idx.free_results(h)
    print("rep %d: %.2f ms for %d sites; stats:" % (rep, el * 1e3, S.n_sites), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k in ("kernel_ms", "finish_kernel_ms", "copy_kernel_ms", "post_kernel_ms", "emit_kernel_ms", "host_ms", "d2h_ms", "h2d_ms", "total_ms", "n_parked_gaps", "n_lean_gaps", "dense_words", "n_launches", "n_retried_gaps", "n_rounds")}, flush=True)
os.environ["MTG_DEBUG_TIMERS"] = "1"
