#!/usr/bin/env python3
"""bench.py -- breakpoints filled / second on the synthetic human-scale set (BASELINE.json configs[3]; configs[4] for N > 1).

One "step" = one pass of the fill hot path (Filler::gapFillFromSource for every site of a batch, reverse attempt for the unfilled ones)
through the C ABI of libmtgfill.so.  The index is built once, before the timed region, and stays resident in HBM; the batches of
sites are marshalled once and resident too (mtg_batch_prepare), so a step starts with its input in HBM; the results (C-ABI records and
ASCII sequences) are in host memory when it ends.  Several distinct batches of sites are rotated through the steps.

N = 1: every step fills one batch of 100 000 sites.  N > 1 (one process per GPU, torchrun): STRONG scaling by default -- one fixed set
of sites (600 000: every donor sequence carries one) is sharded over the ranks with shard_range, a step fills the whole set, the index is
replicated, there is no data-path collective, and every batch's sequences are gathered on rank 0 over RCCL while the next ones are filled.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=os.environ.get("MTG_BENCH_WORKLOAD", "human"), choices=["human", "ecoli", "tiny", "human-het", "human-indel"])
    ap.add_argument("--sites", type=int, default=0, help="sites per batch (default: the workload's)")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches of sites rotated through the steps (N = 1 and weak scaling)")
    ap.add_argument("--nseq", type=int, default=0)
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"], help="N > 1: strong (default) = one fixed site set sharded over the ranks; weak = one batch per rank and step")
    ap.add_argument("--strong-sites", type=int, default=0, help="size of the sharded site set (default: every donor sequence carries a site: 600000)")
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks of --steps steps (0: at least 5 and at least ~1 s of timed work); the median block is reported")
    ap.add_argument("--cpu-sites", type=int, default=30000, help="sites of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-index-seqs", type=int, default=30000, help="donor sequences in the CPU baseline's index")
    ap.add_argument("--host-threads", type=int, default=-1, help="host threads per rank for the per-gap passes (default: the library's pool = CPU budget of the process, shared between the ranks)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MTG_BENCH_IN_FLIGHT", "6")),
                    help="batches in flight: caller threads issuing batches on the one index (the library runs up to six batches of an index side by side)")
    ap.add_argument("--host-strings", action="store_true", help="marshal the sites from host strings inside every step (mtg_fill_batch) instead of filling prepared, device-resident batches")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the random-line ceiling micro-benchmark")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (host-string input, BASELINE configs[4] literal)")
    return ap.parse_args()


WORKLOADS = {
    # name: (donor sequences, sites per batch, description)
    "human": (600000, 100000, "synthetic human-scale: 3 Gbp i.i.d. donor as 600000 x 5 kb sequences, 100000 insertion sites (50-1000 nt) per batch, k=31, max-nodes 100"),
    "ecoli": (1000, 1000, "synthetic E.coli-scale: 5 Mbp donor as 1000 x 5 kb sequences, 1000 insertion sites, k=31"),
    "human-het": (600000, 100000, "secondary, divergence-heavy: diploid donor, 300000 loci x 2 haplotypes x 5 kb with 4 heterozygous SNPs per locus, 100000 insertion sites per batch"),
    "human-indel": (600000, 100000, "diagnostic, general bubble code: the diploid donor of human-het with two deletions of 1-3 nt per locus in the second haplotype besides the 4 SNPs (bubbles with branches of different lengths), 100000 insertion sites per batch"),
    "tiny": (400, 256, "tiny smoke workload"),
}


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def reference_binary():
    """a real MindTheGap install on the box, if any (never this repo's own executable)"""
    exe = shutil.which("MindTheGap")
    if not exe:
        return None
    try:
        out = subprocess.run([exe, "-version"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None
    return None if "mindthegap_amd" in out else {"path": exe, "version": out.strip()}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import mindthegap_amd as mtg
    from mindthegap_amd.shard import PipelinedGather, shard_range
    from mindthegap_amd.synth import SynthSet

    # the ranks of a node share its CPUs (and, in a container, one CFS quota): each rank's worker pool gets its share
    if world > 1:
        os.environ.setdefault("MTG_POOL_THREADS", str(max(2, mtg.cpu_budget() // world)))
    lib = mtg.load_library()
    if not torch.cuda.is_available() or mtg.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # test hooks (single-GPU dry run of the N > 1 path): MTG_BENCH_ONE_DEVICE=1 puts every rank on device 0, MTG_BENCH_BACKEND=gloo
    # replaces RCCL by gloo with host tensors.  The driver's runs use neither.
    if os.environ.get("MTG_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("MTG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    lib.mtg_set_device(local_rank)
    # test hook: MTG_BENCH_FORCE_GATHER=1 runs the N > 1 result path (process group, pipelined gather of every batch's sequences) in a
    # world of one rank, the only RCCL configuration a one-GPU box offers
    dist_on = world > 1 or bool(os.environ.get("MTG_BENCH_FORCE_GATHER"))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif dist_on:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    nseq0, sites0, desc = WORKLOADS[a.workload]
    batch_sites = a.sites or sites0
    het = 4 if a.workload in ("human-het", "human-indel") else 0
    indels = 2 if a.workload == "human-indel" else 0
    nloci0 = (a.nseq or nseq0) // (2 if het else 1)
    scaling = a.scaling if a.scaling != "auto" else ("strong" if world > 1 else "weak")
    if world == 1:
        scaling = "weak"  # one GPU: a step is one batch
    k = 31

    # ---------------------------------------------------------------- the site set and how it is dealt out
    # site i = the insertion of donor sequence i.  batches[b] = range of sites one fill call takes; my_batches = those of this rank, in the
    # order in which it issues them within a step; a step of the whole job = steps_sites sites
    if scaling == "strong":
        total_sites = a.strong_sites or min(nloci0, 6 * batch_sites)
        lo, hi = shard_range(total_sites, rank, world)
        my_batches = [(s, min(s + batch_sites, hi)) for s in range(lo, hi, batch_sites)]
        step_sites = total_sites
        n_sites_total = total_sites
        rotate = False  # every step runs all of the rank's batches
    else:
        nb = max(1, min(a.batches, nloci0 // (batch_sites * world))) if world > 1 else max(1, min(a.batches, nloci0 // batch_sites))
        my_batches = [((rank * nb + b) * batch_sites, (rank * nb + b + 1) * batch_sites) for b in range(nb)]
        step_sites = batch_sites * world
        n_sites_total = world * nb * batch_sites
        rotate = True  # step s runs batch s % nb
    nseq = max(a.nseq or nseq0, n_sites_total * (2 if het else 1))

    # ---------------------------------------------------------------- synthetic donor genome + index (not timed as "fill")
    t0 = time.time()
    S = SynthSet(nseq=nseq, n_sites=n_sites_total, seed=1, k=k, het_snps=het, het_indels=indels)
    t_gen = time.time() - t0
    t0 = time.time()
    w = torch.from_numpy(S.words.view(np.int64)).to(dev)
    wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev)
    ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, k, 3, 0)
    torch.cuda.synchronize()
    t_index = time.time() - t0
    info = idx.info()
    del w, wo, ln
    torch.cuda.empty_cache()

    # ---------------------------------------------------------------- this rank's batches: host strings, prepared (device-resident) form, truth
    host_threads = a.host_threads if a.host_threads >= 0 else 0  # 0 = the library's worker pool as it sized itself
    params = mtg.FillParams(max_nodes=100, max_depth=10000, nb_host_threads=host_threads)

    class B:
        pass

    batches = []
    for (s0, s1) in my_batches:
        b = B()
        b.gaps, b.expected = [], []
        for i in range(s0, s1):
            l, r, ins = S.site(i)
            b.gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
            b.expected.append(ins)
        b.strings = mtg.Index.prepare_gaps(b.gaps)
        b.prepared = b.strings if a.host_strings else idx.prepare_batch(b.strings, params)
        b.digest = hashlib.sha256(("\0".join(b.expected) + "\0").encode()).hexdigest() if b.expected else hashlib.sha256(b"").hexdigest()
        b.n = s1 - s0
        batches.append(b)

    pg = None  # N > 1: pipelined gather of every batch's sequences on rank 0 (created after the first, untimed, pass)
    hbm_pool = None  # secondary measurement: device buffers the sequences are left in (queue of torch tensors)
    acc = dict(kernel_ms=0.0, post_kernel_ms=0.0, emit_kernel_ms=0.0, index_lines=0, contig_nt=0, n_launches=0, host_ms=0.0, d2h_ms=0.0, total_ms=0.0, store_runs=0, run_nt=0,
               post_lines=0, contig_words=0, coverage_kmers=0, dense_words=0, seq_bytes=0, copy_kernel_ms=0.0, copy_words=0, copy_cmds=0, coverage_direct_kmers=0, finish_kernel_ms=0.0, n_parked_gaps=0, gaps=0)
    acc_lock = threading.Lock()

    def fill(b, prepared=None, want_seqs=False, record=False):
        """one batch: forward attempt for every site, reverse attempt (src/Filler.cpp:669-680) for the unfilled ones.  The results are
        host records and ASCII sequences when the call returns.  N > 1: the sequences are written into a page-locked buffer and
        gathered on rank 0 over RCCL / xGMI while the next batches run."""
        prepared = b.prepared if prepared is None else prepared
        seqs = None
        if pg is not None and not want_seqs:
            j, buf = pg.acquire()
            if pg.on_gpu and isinstance(prepared, mtg.Batch):
                # the result kernel writes the sequences into the gather's device buffer: RCCL takes them from there (no trip to the host and
                # back), and the batch's own stream copies them to the rank's page-locked buffer as well, so that every rank holds its records
                # AND sequences in host memory exactly as a single GPU does -- `value` means the same at every N
                ptr, cap_b = pg.device_area(j)
                h, nf, nbytes = idx.fill_prepared_serial_device(prepared, ptr, cap_b, params, host_out=buf)
                pg.submit(nbytes, j, on_device=True)
            else:
                h, nf, nbytes = idx.fill_prepared_serial(prepared, buf, params)  # written straight into the page-locked gather buffer
                pg.submit(nbytes, j)
        elif hbm_pool is not None and not want_seqs and isinstance(prepared, mtg.Batch):
            t = hbm_pool.get()
            h, nf, nbytes = idx.fill_prepared_serial_device(prepared, t.data_ptr(), t.numel(), params)
            hbm_pool.put(t)
        else:
            h, nf, seqs = idx.fill_prepared(prepared, params, want_seqs=want_seqs)
        st = mtg.last_batch_stats()
        idx.free_results(h)
        unfilled = np.nonzero(nf == 0)[0]
        n_filled = int((nf > 0).sum())
        if len(unfilled):
            rg = [mtg.Gap(rc(b.gaps[j].target), rc(b.gaps[j].source), [(rc(b.gaps[j].source), "rev", False)], reverse=True) for j in unfilled]
            h2, nf2, _ = idx.fill_prepared(mtg.Index.prepare_gaps(rg), params, want_seqs=False)
            st2 = mtg.last_batch_stats()
            idx.free_results(h2)
            n_filled += int((nf2 > 0).sum())
            for key in acc:
                if key in st2:
                    st[key] += st2[key]
        if record:
            with acc_lock:
                for key in acc:
                    if key in st:
                        acc[key] += st[key]
                acc["gaps"] += b.n
        return n_filled, seqs

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def run_block(count, record, first_step=0):
        """`count` steps, a.in_flight batches in flight: caller threads take the next batch off a shared list, like the reference's
        Dispatcher threads take the next group of records; every batch is complete when this returns"""
        work = []
        for s in range(first_step, first_step + count):
            work += [batches[s % len(batches)]] if rotate else list(batches)
        it = iter(work)
        errors = []

        def worker():
            try:
                torch.cuda.set_device(local_rank)  # the current device is a per-thread setting
                while True:
                    with acc_lock:
                        b = next(it, None)
                    if b is None:
                        return
                    fill(b, record=record)
            except BaseException as e:  # surfaced on the main thread
                errors.append(e)

        nthreads = max(1, min(a.in_flight, len(work)))
        if nthreads <= 1:
            worker()
        else:
            ts = [threading.Thread(target=worker) for _ in range(nthreads)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]

    # untimed set-up, not a warm-up step: every caller thread enters the library at the same moment, so that each of the index's
    # workspaces (scratch, page-locked staging blocks, streams) and result objects has been allocated once before anything is timed
    if batches:
        gate = threading.Barrier(max(1, a.in_flight))

        def prime(t):
            torch.cuda.set_device(local_rank)
            gate.wait()
            fill(batches[t % len(batches)])

        ts = [threading.Thread(target=prime, args=(t,)) for t in range(max(1, a.in_flight))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if dist_on:
        cap = torch.tensor([max([idx.last_seq_bytes] + [sum(len(e) + 1 for e in b.expected) for b in batches])], dtype=torch.int64, device=cdev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        pg = PipelinedGather(int(cap.item()) * 51 // 50 + (1 << 16), dst=0, device=cdev, depth=max(a.in_flight, 1) + 1)  # the gather moves whole buffers: little slack (the batches and their sizes are known)
    run_block(a.warmup, False)

    # ---------------------------------------------------------------- the timed blocks: each EXACTLY a.steps steps between barrier + synchronize
    def timed_block(first_step):
        barrier()
        t0 = time.perf_counter()
        run_block(a.steps, True, first_step)
        if pg is not None:
            pg.drain()  # the gathers still in flight belong to the timed steps
        barrier()
        el = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([el], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    times = [timed_block(0)]
    repeats = a.repeats if a.repeats > 0 else int(min(60, max(5, np.ceil(1.0 / max(times[0], 1e-6)))))
    if dist_on:  # every rank must run the same number of blocks
        rt = torch.tensor([repeats], device=cdev, dtype=torch.int64)
        dist.broadcast(rt, src=0)
        repeats = int(rt.item())
    for r in range(1, repeats):
        times.append(timed_block(r * a.steps))
    elapsed = float(np.median(times))
    n_blocks = len(times)

    # ---------------------------------------------------------------- verification (untimed): every distinct batch once, sequences against the truth
    gathered_ok = None
    if pg is not None:
        last = pg.last()
        if rank == 0 and last is not None:  # one pipelined payload per rank: it must be one of that rank's batches (which one depends on the interleaving)
            mine = {b.digest for b in batches}
            gathered_ok = hashlib.sha256(last[0].tobytes()).hexdigest() in mine and all(len(p) > 0 for p in last)
    pg_saved, pg = pg, None
    n_filled_rank, identical = 0, True
    for b in batches:
        nf_b, seqs = fill(b, want_seqs=True)
        n_filled_rank += nf_b
        if not het:  # size-independent parity property at full size: every site is filled with exactly its inserted sequence
            identical = identical and hashlib.sha256(seqs.tobytes().replace(b"\n", b"\0")).hexdigest() == b.digest
    if het:
        identical = None  # diploid: the truth is a haplotype mix, checked against the oracle below
    st_alone = mtg.last_batch_stats() if batches else None  # one batch alone on the device
    n_sites_rank = sum(b.n for b in batches)
    if dist_on:
        v = torch.tensor([n_filled_rank, n_sites_rank, 1 if identical in (True, None) else 0], device=cdev, dtype=torch.int64)
        dist.all_reduce(v)
        n_filled_all, n_sites_all, ident_all = int(v[0].item()), int(v[1].item()), int(v[2].item()) == world
        if identical is not None:
            identical = ident_all
    else:
        n_filled_all, n_sites_all = n_filled_rank, n_sites_rank

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    value = step_sites * a.steps / elapsed
    sites_per_rank_step = sum(b.n for b in batches) if not rotate else batch_sites
    # ---------------------------------------------------------------- secondary: the same steps from host strings (marshalling + upload inside the step)
    secondary = {}
    if world == 1 and not a.no_secondary and not a.host_strings and batches:
        for b in batches:
            fill(b, prepared=b.strings)
        t0 = time.perf_counter()
        work = [batches[s % len(batches)] for s in range(a.steps)]
        it = iter(work)

        def hs_worker():
            torch.cuda.set_device(local_rank)
            while True:
                with acc_lock:
                    b = next(it, None)
                if b is None:
                    return
                fill(b, prepared=b.strings)

        ts = [threading.Thread(target=hs_worker) for _ in range(max(1, a.in_flight))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        secondary["value_from_host_strings"] = batch_sites * a.steps / (time.perf_counter() - t0)
        secondary["value_from_host_strings_note"] = "same steps through mtg_fill_batch: the sites are marshalled from the caller's strings and uploaded inside every step"

    # ---------------------------------------------------------------- secondary: the sequences left in HBM (a consumer on the device, or the send buffer of a gather):
    # records still come to the host, the ASCII -- three quarters of the result bytes -- does not cross PCIe
    if world == 1 and not a.no_secondary and not a.host_strings and batches:
        import queue
        cap_b = max(sum(len(e) + 1 for e in b.expected) for b in batches) * 5 // 4 + (1 << 20)
        hbm_pool = queue.Queue()
        for _ in range(max(1, a.in_flight)):
            hbm_pool.put(torch.empty(cap_b, dtype=torch.uint8, device=dev))
        run_block(max(a.warmup, 1), False)
        ts = []
        for r in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_block(a.steps, False, r * a.steps)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = hbm_pool.get()
        b0 = batches[0]
        h0, nf0, nb0 = idx.fill_prepared_serial_device(b0.prepared, t.data_ptr(), t.numel(), params)
        idx.free_results(h0)
        ok_hbm = hashlib.sha256(t[:nb0].cpu().numpy().tobytes()).hexdigest() == b0.digest if not het else None
        hbm_pool = None
        secondary["value_sequences_left_in_hbm"] = batch_sites * a.steps / float(np.median(ts))
        secondary["value_sequences_left_in_hbm_note"] = ("same steps through mtg_fill_prepared_serial_device: records to the host, ASCII sequences into a device buffer of the caller "
                                                         "(median of 5 blocks; one buffer verified against the truth: %s)" % ok_hbm)

    # ---------------------------------------------------------------- CPU baseline (oracle = "port"), bounded sample, rank 0, N = 1 only
    cpu = None
    probes_per_nt = 8.0
    if world == 1 and a.cpu_sites > 0 and batches:
        from tests import oracle_lib
        b0 = batches[0]
        ns = min(a.cpu_sites, b0.n)
        nidx = max(min(a.cpu_index_seqs, S.nseq), ns)
        cores = mtg.cpu_budget()  # the threads the container may actually run (CFS quota), not the hardware threads of the host
        if het:  # both haplotypes of the sampled loci
            nl = S.nseq // 2
            nidx = max(min(a.cpu_index_seqs // 2, nl), ns)
            seqs_ascii = [S.ascii(j) for j in range(nidx)] + [S.ascii(nl + j) for j in range(nidx)]
        else:
            seqs_ascii = [S.ascii(j) for j in range(nidx)]
        oidx = oracle_lib.Index.from_sequences(seqs_ascii, k, 3, 0)
        with tempfile.TemporaryDirectory() as d:
            bk = os.path.join(d, "s.breakpoints")
            S.write_breakpoints(bk, range(ns))
            ost = oidx.fill_files("bkpt", bk, os.path.join(d, "cpu"), params=oracle_lib.default_params(nb_cores=cores))
            cpu_fa = open(os.path.join(d, "cpu.insertions.fasta")).read()
        # the oracle's worker threads write records in completion order: compare as multisets.  The HIP side of the comparison is what the
        # library returned for these very sites, never the truth
        cpu_seqs = sorted(l for l in cpu_fa.splitlines() if not l.startswith(">"))
        res = idx.fill_batch(b0.gaps[:ns], params)
        hip_seqs = sorted(f["seq"] for r in res for f in r["filled"])
        # algorithmic probes per contig nucleotide of the REFERENCE algorithm, counted by the oracle on the sample
        sample_nt = sum(S.seq_len + int(S.ins_len[i]) - int(S.pos[i]) + k for i in range(ns))
        probes_per_nt = ost["probes"] / max(sample_nt, 1)
        cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
        cpu = {"value": ns / ost["seconds"], "unit": "breakpoints/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
               "sample": "%d of the %d sites of batch 0, index over the first %d donor sequences (%d k-mers), CPU restatement of the reference Filler (gatb-core unavailable)"
                         % (ns, b0.n, nidx, len(oidx)),
               "seconds": ost["seconds"], "identical_to_hip": cpu_seqs == hip_seqs and len(hip_seqs) > 0,
               "reference_binary": reference_binary() or "no MindTheGap install on this machine (BASELINE.md section 2): the port is timed"}
        oidx.close()

    # ---------------------------------------------------------------- roofline: bytes the implemented layout has to move, per launch, for every kernel of a step
    L = max(acc["n_launches"], 1)
    gaps_l = acc["gaps"] / L
    bucket = int(info["adj_bucket_bytes"])

    def kern(name, ms, bytes_per_launch, parts):
        avg_s = ms / L * 1e-3
        ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "avg_kernel_ms": ms / L, "bytes_per_launch": bytes_per_launch, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "bytes_breakdown": parts}

    # k_stage_a: one 8-byte word per counted store read (unitig headers, the short reads around a long run), the nucleotides the lanes take out of the
    # store themselves, the contig words the lanes write, one 16-byte command per run left to k_copy; k_copy: every word once in, once out
    lane_nt = max(acc["run_nt"] - 32 * acc["copy_words"], 0)
    sa_parts = {"adj_bucket_reads_x_%dB" % bucket: acc["index_lines"] / L * bucket, "unitig_store_word_reads_x_8B": acc["store_runs"] / L * 8, "unitig_sequence_2bit_taken_by_lanes": lane_nt / L / 4,
                "contig_words_written_by_lanes_x_8B": (acc["contig_words"] - acc["copy_words"]) / L * 8, "copy_commands_x_24B": acc["copy_cmds"] / L * 24,
                "per_gap_input_and_record": gaps_l * (8 + 8 + 4 + 4 + 16 + 36)}
    cp_parts = {"unitig_words_read_x_8B": (acc["copy_words"] + acc["copy_cmds"]) / L * 8, "contig_words_written_x_8B": acc["copy_words"] / L * 8, "copy_commands_x_24B": acc["copy_cmds"] / L * 24,
                "per_gap_record": gaps_l * 36}
    po_parts = {"contig_words_scanned_x_8B": acc["contig_words"] / L * 8, "bucket_reads_x_32B": acc["post_lines"] / L * 32,
                "coverage_abundance_bytes": acc["coverage_kmers"] / L, "coverage_kmer_check_of_looked_up_blocks": (acc["coverage_kmers"] - acc["coverage_direct_kmers"]) / L * 0.25 * (1 + k / 64.0),
                "per_gap_record_and_targets": gaps_l * (36 + 144 + 16 + 2 * 128 + 24)}
    em_parts = {"ascii_written": acc["seq_bytes"] / L, "contig_2bit_read": acc["seq_bytes"] / L / 4, "per_gap_records": gaps_l * (144 + 56 + 40), "dense_contigs_x_16B": acc["dense_words"] / L * 16}
    kerns = [kern("k_stage_a", acc["kernel_ms"], sum(sa_parts.values()), sa_parts), kern("k_copy", acc["copy_kernel_ms"], sum(cp_parts.values()), cp_parts),
             kern("k_post(+k_scan1,k_scan2)", acc["post_kernel_ms"], sum(po_parts.values()), po_parts), kern("k_emit", acc["emit_kernel_ms"], sum(em_parts.values()), em_parts)]
    dom = max(kerns, key=lambda x: x["avg_kernel_ms"])
    roof = dict(dom)
    roof["kernels"] = kerns
    roof["launches"] = int(acc["n_launches"])
    roof["traffic"], roof["traffic_source"] = None, None
    pmc = os.path.join(ROOT, "profiles", "r02_pmc.json")
    if a.workload == "human" and batch_sites == 100000 and os.path.exists(pmc):
        pj = json.load(open(pmc))
        key = "mtgi::" + dom["kernel"].split("(")[0]
        if key in pj.get("kernels", {}):
            roof["traffic"] = pj["kernels"][key].get("hbm_bytes_per_launch")
            roof["traffic_source"] = "profiles/r02_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same command, HEAD %s)" % pj.get("head", "?")
    # what the reference's algorithm would have moved for the same contigs (64 B per membership probe, probes counted by the oracle):
    # kept for comparison only -- the unitig layout reads one bucket and one stretch of 2-bit sequence where gatb probes 8 Bloom blocks per nucleotide
    roof["reference_algorithm_equivalent"] = {"bytes_per_launch": 64.0 * probes_per_nt * acc["contig_nt"] / L, "probes_per_contig_nt": probes_per_nt,
                                              "equivalent_GBps_of_k_stage_a": 64.0 * probes_per_nt * acc["contig_nt"] / L / max(acc["kernel_ms"] / L * 1e-3, 1e-12) / 1e9}
    roof["pcie"] = {"result_bytes_per_launch": acc["seq_bytes"] / L + gaps_l * (56 + 40), "note": "records and ASCII sequences are copied to page-locked host memory inside every step"}
    if st_alone:
        roof["one_batch_alone_ms"] = {"k_stage_a": st_alone["kernel_ms"] / max(st_alone["n_launches"], 1), "k_copy": st_alone["copy_kernel_ms"] / max(st_alone["n_launches"], 1), "k_post+scans": st_alone["post_kernel_ms"] / max(st_alone["n_launches"], 1),
                                      "k_emit": st_alone["emit_kernel_ms"] / max(st_alone["n_launches"], 1)}
    if not a.no_ceiling:
        tb = min(int(info["device_bytes"] // 2), 16 << 30)  # the ceiling is flat beyond ~16 GB (profiles/r01_random_line_ceiling.txt)
        ms, gbps = mtg.random_line_ceiling(max(tb, 1 << 26), batch_sites, 512, bucket)
        roof["random_read_ceiling_reads_per_s"] = gbps * 1e9 / bucket
        # the traversal against the HBM-random-read roofline (north star): its dependent random reads (ADJ buckets + short reads of the unitig store)
        # per second of kernel time, with six batches in flight and for one batch alone, over the measured ceiling of dependent random reads
        rr = (acc["index_lines"] + acc["store_runs"]) / L
        avg_s = acc["kernel_ms"] / L * 1e-3
        roof["random_reads_of_k_stage_a"] = {"per_launch": rr, "reads_per_s": rr / avg_s if avg_s > 0 else 0.0, "frac_of_ceiling": rr / avg_s / roof["random_read_ceiling_reads_per_s"] if avg_s > 0 else 0.0}
        if st_alone and st_alone["kernel_ms"] > 0:
            rr1 = (st_alone["index_lines"] + st_alone["store_runs"]) / max(st_alone["n_launches"], 1)
            t1 = st_alone["kernel_ms"] / max(st_alone["n_launches"], 1) * 1e-3
            roof["random_reads_of_k_stage_a"]["alone"] = {"per_launch": rr1, "reads_per_s": rr1 / t1, "frac_of_ceiling": rr1 / t1 / roof["random_read_ceiling_reads_per_s"]}

    out = {"metric": "breakpoints filled/sec", "value": value, "unit": "breakpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": desc, "sites_per_step": step_sites, "sites_per_batch": batch_sites, "sites_per_rank_and_step": sites_per_rank_step, "distinct_batches_per_rank": len(batches),
                      "site_set": n_sites_total, "donor_sequences": S.nseq, "k": k, "max_nodes": 100, "max_length": 10000,
                      "index": "exact k-mer set of the donor, abundance of a k-mer = Poisson(24) drawn from its hash, at least 3 (SURVEY 8d; no reads simulated)", "nb_solid_kmers": int(info["nb_solid_kmers"]),
                      "nb_unitigs": int(info["nb_unitigs"]), "index_bytes": int(info["device_bytes"]), "unitig_store_bytes": int(info["unitig_bytes"]), "index_build_s": t_index,
                      "genome_gen_s": t_gen, "batches_in_flight": a.in_flight, "input": "host strings, marshalled in every step" if a.host_strings else "prepared batches, resident in HBM",
                      "output": "C-ABI records + ASCII sequences in page-locked host memory"},
           "timed_blocks": {"blocks": n_blocks, "steps_per_block": a.steps, "reported": "median", "ms_per_step_min": min(times) / a.steps * 1e3,
                            "ms_per_step_median": elapsed / a.steps * 1e3, "ms_per_step_max": max(times) / a.steps * 1e3, "timed_seconds_total": sum(times)},
           "filled": n_filled_all, "sites_verified": n_sites_all, "filled_per_s": value * n_filled_all / max(n_sites_all, 1),
           "filled_sequences_identical_to_truth": identical, "gathered_payload_verified": gathered_ok,
           "stage_ms_per_batch": {"k_stage_a+k_finish": acc["kernel_ms"] / L, "k_finish": acc["finish_kernel_ms"] / L, "parked_gaps": acc["n_parked_gaps"] / L, "k_copy": acc["copy_kernel_ms"] / L, "k_post+scans": acc["post_kernel_ms"] / L, "k_emit": acc["emit_kernel_ms"] / L, "d2h": acc["d2h_ms"] / L,
                                  "host": acc["host_ms"] / L, "c_call": acc["total_ms"] / L},
           "roofline": roof, "cpu_baseline": cpu}
    out.update(secondary)
    # ---------------------------------------------------------------- secondary line: the diploid workload (walks cross SNP bubbles), as a child
    # process once this one has given the device back (the two indexes do not fit the HBM together)
    if world == 1 and not a.no_secondary and a.workload == "human" and not a.host_strings:
        for b in batches:
            if hasattr(b.prepared, "close"):
                b.prepared.close()
        batches.clear()
        idx.close()
        torch.cuda.empty_cache()
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "human-het", "--batches", "3", "--cpu-sites", "0", "--no-ceiling", "--no-secondary",
                                 "--steps", str(a.steps), "--warmup", str(a.warmup), "--in-flight", str(a.in_flight)], capture_output=True, text=True, timeout=400)
            d = json.loads(cp.stdout.strip().splitlines()[-1])
            out["secondary_diploid"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "timed_blocks": d["timed_blocks"],
                                        "filled": d["filled"], "sites_verified": d["sites_verified"], "one_batch_alone_ms": d["roofline"].get("one_batch_alone_ms"),
                                        "ratio_to_headline": d["value"] / value if value else None}
        except Exception as e:  # the headline line does not depend on it
            out["secondary_diploid"] = {"error": repr(e)[:300]}
    if dist_on:
        pg_saved.drain()
        dist.destroy_process_group()
    # the JSON line is the last thing on stdout: whatever native libraries (RCCL's version banner) left in the C stdio buffer goes first
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
