"""N > 1 path on CPU: world_size-2 and world_size-8 gloo jobs; shards + gather must reproduce the single-process result."""
import json
import os
import subprocess
import sys

import pytest

from mindthegap_amd.shard import shard_range
from mindthegap_amd.synth import SynthSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 100, 100000):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_job_matches_truth(tmp_path, world):
    from tests import emu_lib, oracle_lib
    emu_lib.build_full()  # build once here: the two ranks must not race on the compiler
    oracle_lib.build()
    out = str(tmp_path / "gathered.bin")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    # bench.py's N > 1 path (tests/dist_worker.py: bench_shaped_job): every payload of every rank and step verified on rank 0; at eight ranks some
    # ranks pad their collectives, and the last set runs through the slotted gather
    v = json.load(open(out + ".bench_shaped.json"))
    assert [x["batches"] for x in v] == [sum(x["per_rank"]) for x in v] and all(len(x["per_rank"]) == world for x in v)
    assert any(x["padded_ranks"] > 0 for x in v) and any(x["slots"] > 1 for x in v), v
    S = SynthSet(nseq=24, n_sites=20, seed=3)
    expected = "".join(S.site(i)[2] + "\n" for i in range(S.n_sites)).encode()
    assert open(out, "rb").read() == expected  # rank order == site order, every fill identical to the inserted sequence
    # the sharded tool run of the two ranks (7 batches of 3 sites dealt in turn: rank 1 has no batch in the last round) against the single-process tool
    # on the same sites: FASTA, info and extension files byte for byte, the VCF below its dated header
    mtg = emu_lib.product_on_emulator()
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    import struct
    idxf = str(tmp_path / "s.mtgidx")
    with open(idxf, "wb") as f:
        f.write(b"MTGIDX1\0" + struct.pack("<iiii", 31, 3, -1, 0) + struct.pack("<Q", len(km)) + km.tobytes() + ct.tobytes())
    bk = str(tmp_path / "s.breakpoints")
    with open(bk, "w") as f:
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            if i % 5 == 2:
                r = l[::-1].translate(str.maketrans("ACGT", "TGCA"))
            if i % 7 == 3:
                l = l[:10] + "N" + l[11:]
            f.write(">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (S.site_name(i), l, S.site_name(i), r))
    assert mtg.Filler().run(["-graph", idxf, "-bkpt", bk, "-out", str(tmp_path / "single"), "-extend"]) == 0
    for ext in (".insertions.fasta", ".info.txt", ".extensions.fasta"):
        a, b = open(out + ".sharded" + ext, "rb").read(), open(str(tmp_path / "single") + ext, "rb").read()
        assert a == b and (len(a) > 0 or ext == ".extensions.fasta"), ext
    body = lambda p: [l for l in open(p).read().splitlines() if not l.startswith("##")]
    assert body(out + ".sharded.insertions.vcf") == body(str(tmp_path / "single.insertions.vcf"))
    o.close()
