"""N > 1 path on CPU: world_size-2 gloo job; shards + gather must reproduce the single-process result."""
import os
import subprocess
import sys

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


def test_world_size_2_gloo_matches_truth(tmp_path):
    from tests import emu_lib, oracle_lib
    emu_lib.build_full()  # build once here: the two ranks must not race on the compiler
    oracle_lib.build()
    out = str(tmp_path / "gathered.bin")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    import socket
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "dist_worker.py"), out]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    S = SynthSet(nseq=24, n_sites=20, seed=3)
    expected = "".join(S.site(i)[2] + "\n" for i in range(S.n_sites)).encode()
    assert open(out, "rb").read() == expected  # rank order == site order, every fill identical to the inserted sequence
