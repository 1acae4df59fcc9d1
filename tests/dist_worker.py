"""worker of tests/test_distributed_cpu.py: one rank of a world_size-N gloo job.  Fills its shard of a synthetic set on the
TEST-ONLY emulation build (no GPU here) and gathers the filled sequences on rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_shaped_job(mtg, idx, S, total_sites, batch_sites, in_flight=3, steps=2):
    """The N > 1 result path of bench.py on the emulator: the strong-scaling plan (shard.py: strong_plan), one payload per batch in relocatable
    form (mtg_results_to_wire), gathered on rank 0 through the gather bench.py would pick for that plan (PipelinedGather, or SlottedGather once a
    rank has three batches per step), ranks with fewer batches than the others padding their collectives, caller threads racing; rank 0 checks
    EVERY payload of EVERY rank of EVERY step where it arrives (wire_check with the checksum), rebuilds it (mtg_results_from_wire) and compares
    its sequences with the digest the owning rank computed from the truth.  Returns rank 0's verdict (None elsewhere)."""
    import hashlib
    import threading
    import torch
    import torch.distributed as dist
    from mindthegap_amd import lib as L
    from mindthegap_amd.shard import PipelinedGather, SlottedGather, gather_slots_for, strong_plan, wire_check
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = strong_plan(total_sites, batch_sites, rank, world)
    # the plans of all ranks tile the site set in rank order, and the global batch indices are 0 .. n - 1 in site order
    every = [strong_plan(total_sites, batch_sites, r, world) for r in range(world)]
    flat = [b for pl in every for b in pl["my"]]
    assert [g for g, _, _ in flat] == list(range(plan["n_batches_job"]))
    assert flat[0][1] == 0 and flat[-1][2] == total_sites and all(flat[i][2] == flat[i + 1][1] for i in range(len(flat) - 1))
    assert [len(pl["my"]) for pl in every] == plan["per_rank"] and max(plan["per_rank"]) == plan["max_per_rank"]
    batches = []
    for gidx, s0, s1 in plan["my"]:
        gaps, expected = [], []
        for i in range(s0, s1):
            l, r, ins = S.site(i)
            gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
            expected.append(ins)
        batches.append(dict(gidx=gidx, n=s1 - s0, prepared=mtg.Index.prepare_gaps(gaps), digest=hashlib.sha256(("\0".join(expected) + "\0").encode()).hexdigest()))
    need = torch.tensor([max([0] + [96 * b["n"] + 1100 * b["n"] + 4096 for b in batches])], dtype=torch.int64)
    dist.all_reduce(need, op=dist.ReduceOp.MAX)
    slots, depth = gather_slots_for(plan["max_per_rank"], in_flight)
    seen, bad, keep, lock = {}, [], {}, threading.Lock()

    def on_arrival(items):
        for it in items:
            r, tag, t = it
            if tag < 0:
                continue
            h = wire_check(t, checksum=True, head=getattr(it, "head", None))
            with lock:
                if not h["ok"] or h["tag"] != tag:
                    bad.append((r, tag, h.get("why")))
                seen[(r, tag)] = seen.get((r, tag), 0) + 1
                keep[tag] = t.cpu().numpy().copy()

    if slots > 1:
        pg = SlottedGather(int(need.item()), slots=slots, dst=0, depth=depth, on_arrival=on_arrival if rank == 0 else None)
    else:
        pg = PipelinedGather(int(need.item()), dst=0, depth=depth, on_arrival=on_arrival if rank == 0 else None)
    errors = []

    def fill(b):
        try:
            j, buf = pg.acquire()
            h, _nf, _ = idx.fill_prepared(b["prepared"], want_seqs=False)
            payload = L.results_to_wire(h, b["gidx"], buf)
            idx.free_results(h)
            pg.submit(payload.size, j, tag=b["gidx"])
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    for _step in range(steps):
        work, wl = list(batches), threading.Lock()

        def caller():
            while True:
                with wl:
                    if not work:
                        return
                    b = work.pop(0)
                fill(b)

        ts = [threading.Thread(target=caller) for _ in range(in_flight)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
        for _ in range(plan["max_per_rank"] - len(batches)):  # bench.py: pad_gathers -- every rank issues the same number of collectives
            j, _buf = pg.acquire()
            pg.submit(0, j, tag=-1)
    pg.drain()
    mine = [(b["gidx"], b["digest"], b["n"]) for b in batches]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank != 0:
        return None
    want = {g: (d, n, r) for r, lst in enumerate(everyone) for (g, d, n) in lst}
    assert not bad, bad
    assert len(want) == plan["n_batches_job"]
    assert sorted(seen) == sorted((r, g) for g, (_d, _n, r) in want.items()), (sorted(seen), want)
    assert all(c == steps for c in seen.values()), seen  # every payload of every step arrived exactly once
    for g, (d, n, _r) in want.items():
        p = keep[g]
        hd = L.wire_header(p)
        wr = L.WireResults(p)
        o_s = 64 + 40 * hd["n_gaps"] + 40 * hd["n_filled"]
        assert hd["n_gaps"] == n and hd["tag"] == g and wr.tag == g
        assert hashlib.sha256(p[o_s: o_s + hd["seq_bytes"]].tobytes()).hexdigest() == d, g
        wr.close()
    return dict(batches=len(want), per_rank=plan["per_rank"], slots=slots, padded_ranks=sum(1 for c in plan["per_rank"] if c < plan["max_per_rank"]))


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
    # bench.py's own N > 1 path, scaled down: the "every donor sequence carries a site" set (several batches per rank at two ranks, one at eight,
    # uneven: some ranks pad) and BASELINE config 5's literal shape (one small batch per rank); then a set cut so that batch counts differ between
    # ranks at both world sizes and the slotted gather is used (three or more batches per step on the fullest rank)
    verdicts = []
    for total, bs in ((20, 4), (13, 4), (19, 1)):
        verdicts.append(bench_shaped_job(mtg, idx, S, total, bs))
    if rank == 0:
        import json
        with open(out_path + ".bench_shaped.json", "w") as f:
            json.dump(verdicts, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
