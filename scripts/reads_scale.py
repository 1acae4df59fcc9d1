#!/usr/bin/env python3
"""`MindTheGap fill -in <reads>` beyond the unit-test sizes: a donor of nseq x 5 kb, 30x error-free 150-nt reads written to a FASTA file
(vectorised generator), k-mer counting on the device from the streamed file, fill of every site; every fill must be the inserted sequence."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet, NT

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S = SynthSet(nseq=nseq, n_sites=nseq, seed=2, k=31)
d = tempfile.mkdtemp()
reads, bk = os.path.join(d, "reads.fasta"), os.path.join(d, "s.breakpoints")
rng = np.random.default_rng(7)
comp = np.array([2, 3, 0, 1], dtype=np.uint8)
t0 = time.time()
nreads = 0
with open(reads, "wb") as f:
    for j in range(S.nseq):
        c = S.codes(j)
        L = len(c)
        nr = int(round(30 * L / 150))
        st = rng.integers(0, L - 150 + 1, nr)
        m = c[st[:, None] + np.arange(150)[None, :]]
        rev = rng.integers(0, 2, nr).astype(bool)
        m[rev] = comp[m[rev][:, ::-1]]
        out = np.empty((nr, 154), dtype=np.uint8)
        out[:, 0] = ord(">"); out[:, 1] = ord("r"); out[:, 2] = 10; out[:, 153] = 10
        out[:, 3:153] = NT[m]
        f.write(out.tobytes())
        nreads += nr
S.write_breakpoints(bk)
print("donor %.0f Mbp, %d reads, %.2f GB of FASTA written in %.0f s" % (S.lens.sum() / 1e6, nreads, os.path.getsize(reads) / 1e9, time.time() - t0), flush=True)
t0 = time.time()
assert mtg.fill_main(["-in", reads, "-bkpt", bk, "-abundance-min", "3", "-out", os.path.join(d, "hip")]) == 0
t_all = time.time() - t0
seqs = [l.rstrip("\n") for l in open(os.path.join(d, "hip.insertions.fasta")) if not l.startswith(">")]
ok = seqs == [S.site(i)[2] for i in range(S.n_sites)]
g = mtg.Index.load(os.path.join(d, "hip.mtgidx"))
print("MindTheGap fill -in: %.1f s in all (counting, index, %d fills, files); solid k-mers %d; every fill equals the inserted sequence: %s" % (t_all, len(seqs), g.info()["nb_solid_kmers"], ok), flush=True)
g.close()
for fn in os.listdir(d):
    os.remove(os.path.join(d, fn))
os.rmdir(d)
assert ok
