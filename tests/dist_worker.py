"""worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo job.  Fills its shard of a synthetic set on the
TEST-ONLY emulation build (no GPU here) and gathers the filled sequences on rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import torch.distributed as dist
    from mindthegap_amd.shard import PipelinedGather, gather_bytes, shard_range
    from mindthegap_amd.synth import SynthSet
    from tests import emu_lib, oracle_lib
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mtg = emu_lib.product_on_emulator()
    S = SynthSet(nseq=24, n_sites=20, seed=3)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    idx = mtg.Index.from_kmers(km, ct, 31)  # replicated index
    lo, hi = shard_range(S.n_sites, rank, world)
    gaps = []
    for i in range(lo, hi):
        l, r, _ = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
    h, nf, seqs = idx.fill_prepared(mtg.Index.prepare_gaps(gaps))
    idx.free_results(h)
    parts = gather_bytes(np.asarray(seqs), dst=0)
    # the per-step pipelined gather of bench.py: three steps in flight through two buffers, payloads written in place
    import torch
    cap = torch.tensor([len(seqs)], dtype=torch.int64)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)  # every rank must use the same capacity
    pg = PipelinedGather(int(cap.item()) * 2 + 64, dst=0)
    prepared = mtg.Index.prepare_gaps(gaps)
    for step in range(3):
        h, nf, nbytes = idx.fill_prepared_serial(prepared, pg.buffer())  # NUL-terminated sequences, written into the gather buffer
        idx.free_results(h)
        pg.submit(nbytes)
    pg.drain()
    piped = pg.last()
    if rank == 0:
        assert [p.tobytes().replace(b"\0", b"\n") for p in piped] == [p.tobytes() for p in parts]
    # bench.py's default: several steps in flight, issued from caller threads that share one PipelinedGather (acquire / submit(j))
    import threading
    pg2 = PipelinedGather(int(cap.item()) * 2 + 64, dst=0, depth=3)
    todo, lock, errors = iter(range(6)), threading.Lock(), []

    def worker():
        try:
            while True:
                with lock:
                    if next(todo, None) is None:
                        return
                j, buf = pg2.acquire()
                h2, _, nb = idx.fill_prepared_serial(prepared, buf)
                idx.free_results(h2)
                pg2.submit(nb, j)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    pg2.drain()
    piped2 = pg2.last()
    if rank == 0:  # every step carries the same payload, so whichever step came last must equal the blocking gather
        assert [p.tobytes().replace(b"\0", b"\n") for p in piped2] == [p.tobytes() for p in parts]
        with open(out_path, "wb") as f:
            for p in parts:
                f.write(p.tobytes())
    # several payloads per collective (SlottedGather: what bench.py uses when a rank has several batches per step): seven tagged payloads through buffers of
    # two slots from two caller threads; rank 0 must see every (rank, tag) once, with the payload that rank sent under that tag
    from mindthegap_amd.shard import SlottedGather
    seen3, lock3 = {}, threading.Lock()

    def arrive3(items):
        with lock3:
            for it in items:
                r, tag, t = it
                if tag >= 0:
                    assert (r, tag) not in seen3
                    seen3[(r, tag)] = t.cpu().numpy().tobytes()

    sg = SlottedGather(256, slots=2, dst=0, depth=3, on_arrival=arrive3 if rank == 0 else None)
    todo3 = iter(range(7))

    def worker3():
        try:
            while True:
                with lock:
                    t = next(todo3, None)
                if t is None:
                    return
                h3, buf3 = sg.acquire()
                msg = ("rank %d payload %d " % (rank, t)).encode() * (1 + t)
                buf3[: len(msg)] = np.frombuffer(msg, dtype=np.uint8)
                sg.submit(len(msg), h3, tag=t)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker3) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    sg.drain()
    if rank == 0:
        assert sorted(seen3) == [(r, t) for r in range(world) for t in range(7)], sorted(seen3)
        for (r, t), b in seen3.items():
            assert b == ("rank %d payload %d " % (r, t)).encode() * (1 + t)
    # the whole path a multi-GPU run of the tool takes: shards of batches, results (records AND sequences, tagged with their batch index) gathered on
    # rank 0, put back in input order and written by the tool's own writers -- the files must equal those of the single-process tool
    from mindthegap_amd.shard import fill_bkpt_sharded
    sites = []
    for i in range(S.n_sites):
        l, r, _ = S.site(i)
        if i % 5 == 2:
            r = l[::-1].translate(str.maketrans("ACGT", "TGCA"))  # an anchor pair with nothing in between: reverse attempt, no fill
        if i % 7 == 3:
            l = l[:10] + "N" + l[11:]  # a character revcomp_sequence drops (src/Utils.cpp:44-77): the reverse attempt's target is one shorter
        sites.append((S.site_name(i), S.site_name(i), l, r))
    fill_bkpt_sharded(idx, sites, out_path + ".sharded", batch_sites=3, extend=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
