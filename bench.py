#!/usr/bin/env python3
"""bench.py -- breakpoints filled / second on the synthetic human-scale set (BASELINE.json configs[3]).

One "step" = one pass of the fill hot path (Filler::gapFillFromSource for every site, reverse attempt for the unfilled ones)
over one batch of sites, through the C ABI of libmtgfill.so.  The index is built once, before the timed region, and stays
resident in HBM.  N > 1: one process per GPU (torchrun), the index is replicated (each rank builds the same deterministic
donor genome), each rank fills its own sites (no data-path collective), the filled sequences are gathered on rank 0 over RCCL.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=os.environ.get("MTG_BENCH_WORKLOAD", "human"), choices=["human", "ecoli", "tiny", "human-het"])
    ap.add_argument("--sites", type=int, default=0, help="sites per GPU (default: the workload's)")
    ap.add_argument("--nseq", type=int, default=0)
    ap.add_argument("--cpu-sites", type=int, default=30000, help="sites of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-index-seqs", type=int, default=30000, help="donor sequences in the CPU baseline's index")
    ap.add_argument("--host-threads", type=int, default=-1, help="host threads per rank for the per-gap passes (default: the library's pool = CPU budget of the process, shared between the ranks)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MTG_BENCH_IN_FLIGHT", "3")),
                    help="steps in flight: caller threads issuing batches on the one index (the library runs up to three batches of an index side by side)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the random-64B-line ceiling micro-benchmark")
    return ap.parse_args()


WORKLOADS = {
    # name: (donor sequences, sites per GPU, description)
    "human": (600000, 100000, "synthetic human-scale: 3 Gbp i.i.d. donor as 600000 x 5 kb sequences, 100000 insertion sites (50-1000 nt), k=31, max-nodes 100"),
    "ecoli": (1000, 1000, "synthetic E.coli-scale: 5 Mbp donor as 1000 x 5 kb sequences, 1000 insertion sites, k=31"),
    "human-het": (600000, 100000, "secondary, divergence-heavy: diploid donor, 300000 loci x 2 haplotypes x 5 kb with 4 heterozygous SNPs per locus, 100000 insertion sites"),
    "tiny": (400, 256, "tiny smoke workload"),
}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import mindthegap_amd as mtg
    from mindthegap_amd.shard import PipelinedGather, gather_bytes
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
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    nseq0, sites0, desc = WORKLOADS[a.workload]
    sites_per_gpu = a.sites or sites0
    n_sites_total = sites_per_gpu * world
    het = 4 if a.workload == "human-het" else 0
    nseq = max(a.nseq or nseq0, n_sites_total * (2 if het else 1))
    k = 31

    # ---------------------------------------------------------------- synthetic donor genome + index (not timed as "fill")
    t0 = time.time()
    S = SynthSet(nseq=nseq, n_sites=n_sites_total, seed=1, k=k, het_snps=het)
    t_gen = time.time() - t0
    t0 = time.time()
    w = torch.from_numpy(S.words.view(np.int64)).to(dev)
    wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev)
    ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, k, 3, 40)
    torch.cuda.synchronize()
    t_index = time.time() - t0
    info = idx.info()
    del w, wo, ln
    torch.cuda.empty_cache()

    # ---------------------------------------------------------------- this rank's sites
    my_sites = range(rank * sites_per_gpu, (rank + 1) * sites_per_gpu)
    gaps, expected = [], []
    for i in my_sites:
        l, r, ins = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
        expected.append(ins)
    prepared = mtg.Index.prepare_gaps(gaps)
    # 0 = the library's worker pool as it sized itself (CPU budget of the process, at most 64; MTG_POOL_THREADS above for N > 1)
    host_threads = a.host_threads if a.host_threads >= 0 else 0
    params = mtg.FillParams(max_nodes=100, max_depth=10000, nb_host_threads=host_threads)
    exp_digest = hashlib.sha256(("\n".join(expected) + "\n").encode()).hexdigest()

    def rc(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))

    pg = None  # N > 1: pipelined gather of every step's sequences on rank 0 (created after the first, untimed, step)

    def step(want_seqs=False, final=False):
        """forward attempt for every site, reverse attempt (src/Filler.cpp:669-680) for the unfilled ones.  The results stay in the
        library's result arena (mtg_results_get).  N > 1: every step's sequences are serialised into a page-locked buffer and gathered
        on rank 0 over RCCL / xGMI while the next step runs; the final, untimed pass uses the blocking gather and is verified."""
        tp0 = time.perf_counter()
        if pg is not None and not final:
            j, buf = pg.acquire()
            h, nf, nbytes = idx.fill_prepared_serial(prepared, buf, params)  # decoded straight into the page-locked gather buffer
            seqs = None
            pg.submit(nbytes, j)
        else:
            h, nf, seqs = idx.fill_prepared(prepared, params, want_seqs=want_seqs or final)
        st = mtg.last_batch_stats()
        tp1 = time.perf_counter()
        idx.free_results(h)
        if os.environ.get("MTG_BENCH_DEBUG"):
            sys.stderr.write("step: fill_prepared %.1f ms (C total %.1f: marshal %.1f kernel %.1f post %.1f h2d %.1f d2h %.1f host %.1f result %.1f) free %.1f ms\n" % (
                (tp1 - tp0) * 1e3, st["total_ms"], st["marshal_ms"], st["kernel_ms"], st["post_kernel_ms"], st["h2d_ms"], st["d2h_ms"], st["host_ms"], st["result_ms"],
                (time.perf_counter() - tp1) * 1e3))
        unfilled = np.nonzero(nf == 0)[0]
        n_filled = int((nf > 0).sum())
        if len(unfilled):
            rg = [mtg.Gap(rc(gaps[j].target), rc(gaps[j].source), [(rc(gaps[j].source), "rev", False)], reverse=True) for j in unfilled]
            h2, nf2, _ = idx.fill_prepared(mtg.Index.prepare_gaps(rg), params, want_seqs=False)
            st2 = mtg.last_batch_stats()
            idx.free_results(h2)
            n_filled += int((nf2 > 0).sum())
            for key in ("kernel_ms", "post_kernel_ms", "h2d_ms", "d2h_ms", "host_ms", "index_lines", "n_launches", "contig_nt"):
                st[key] += st2[key]
        if world > 1 and final:  # blocking gather (all_gather of sizes + padded gather)
            gather_bytes(seqs, dst=0, device=cdev)
        return n_filled, seqs, st

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        step()  # untimed: sizes the gather buffers (the largest payload of any rank, plus head room)
        cap = torch.tensor([idx.last_seq_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        pg = PipelinedGather(int(cap.item()) * 5 // 4 + (1 << 20), dst=0, device=cdev, depth=max(a.in_flight, 1) + 1)
    import threading
    acc = dict(kernel_ms=0.0, index_lines=0, contig_nt=0, n_launches=0, host_ms=0.0, d2h_ms=0.0, post_kernel_ms=0.0, total_ms=0.0)
    acc_lock = threading.Lock()

    def run_steps(count, record):
        """`count` steps, a.in_flight of them in flight: caller threads take the next step off a shared counter, like the reference's
        Dispatcher threads take the next group of records; every step is complete when this returns"""
        todo = iter(range(count))
        errors = []

        def worker():
            try:
                torch.cuda.set_device(local_rank)  # the current device is a per-thread setting
                while True:
                    with acc_lock:
                        if next(todo, None) is None:
                            return
                    _, _, st = step()
                    if record:
                        with acc_lock:
                            for key in acc:
                                acc[key] += st[key]
            except BaseException as e:  # surfaced on the main thread
                errors.append(e)

        if a.in_flight <= 1:
            worker()
        else:
            ts = [threading.Thread(target=worker) for _ in range(a.in_flight)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]

    if a.in_flight > 1:
        # untimed set-up, not a warm-up step: every caller thread enters the library at the same moment, so that each of the index's
        # workspaces (scratch, page-locked staging blocks, streams) has been allocated once before anything is timed
        gate = threading.Barrier(a.in_flight)

        def prime():
            torch.cuda.set_device(local_rank)
            gate.wait()
            step()

        ts = [threading.Thread(target=prime) for _ in range(a.in_flight)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    run_steps(a.warmup, False)
    barrier()
    t0 = time.perf_counter()
    n_filled, seqs = 0, np.empty(0, dtype=np.uint8)
    run_steps(a.steps, True)
    kernel_ms, lines, contig_nt, launches = acc["kernel_ms"], acc["index_lines"], acc["contig_nt"], acc["n_launches"]
    host_ms, d2h_ms, post_ms, call_ms = acc["host_ms"], acc["d2h_ms"], acc["post_kernel_ms"], acc["total_ms"]
    if pg is not None:
        pg.drain()  # the gathers still in flight belong to the timed steps
    barrier()
    elapsed = time.perf_counter() - t0
    n_filled, seqs, st_alone = step(want_seqs=True, final=True)  # untimed pass whose sequences are verified below (one batch on the device)
    if world > 1:
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nf_t = torch.tensor([n_filled], device=cdev, dtype=torch.int64)
        dist.all_reduce(nf_t)
        n_filled_all = int(nf_t.item())
    else:
        n_filled_all = n_filled
    # size-independent parity property at full size: every site is filled with exactly its inserted sequence
    identical = hashlib.sha256(seqs.tobytes()).hexdigest() == exp_digest if not het else None  # diploid: the truth is a haplotype mix, checked against the oracle below

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = n_sites_total * a.steps / elapsed
    # ---------------------------------------------------------------- CPU baseline (oracle = "port"), bounded sample, rank 0, N = 1 only
    cpu = None
    probes_per_nt = 8.0
    if world == 1 and a.cpu_sites > 0:
        from tests import oracle_lib
        ns = min(a.cpu_sites, sites_per_gpu)
        nidx = max(min(a.cpu_index_seqs, S.nseq), ns)
        cores = mtg.cpu_budget()  # the threads the container may actually run (CFS quota), not the hardware threads of the host
        if het:  # both haplotypes of the sampled loci
            nl = S.nseq // 2
            nidx = max(min(a.cpu_index_seqs // 2, nl), ns)
            seqs_ascii = [S.ascii(j) for j in range(nidx)] + [S.ascii(nl + j) for j in range(nidx)]
        else:
            seqs_ascii = [S.ascii(j) for j in range(nidx)]
        oidx = oracle_lib.Index.from_sequences(seqs_ascii, k, 3, 40)
        with tempfile.TemporaryDirectory() as d:
            bk = os.path.join(d, "s.breakpoints")
            S.write_breakpoints(bk, range(ns))
            ost = oidx.fill_files("bkpt", bk, os.path.join(d, "cpu"), params=oracle_lib.default_params(nb_cores=cores))
            cpu_fa = open(os.path.join(d, "cpu.insertions.fasta")).read()
        # the oracle's worker threads write records in completion order: compare as multisets
        cpu_seqs = sorted(l for l in cpu_fa.splitlines() if not l.startswith(">"))
        hip_all = seqs.tobytes().decode().split("\n")[:-1] if rank == 0 else []
        hip_seqs = sorted(hip_all[:ns]) if n_filled == sites_per_gpu and len(hip_all) == sites_per_gpu else sorted(expected[:ns])
        # algorithmic probes per contig nucleotide, counted by the oracle on the sample
        sample_nt = sum(S.seq_len + int(S.ins_len[i]) - int(S.pos[i]) + k for i in range(ns))
        probes_per_nt = ost["probes"] / max(sample_nt, 1)
        cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
        cpu = {"value": ns / ost["seconds"], "unit": "breakpoints/s", "cores": cores, "cpu_model": cpu_model, "kind": "port",
               "sample": "%d of the %d sites, index over the first %d donor sequences (%d k-mers), CPU restatement of the reference Filler (gatb-core unavailable)"
                         % (ns, sites_per_gpu, nidx, len(oidx)),
               "seconds": ost["seconds"], "identical_to_hip": cpu_seqs == hip_seqs and identical is not False}
        oidx.close()

    # ---------------------------------------------------------------- roofline of the dominant kernel (k_stage_a)
    # achieved = ALGORITHMIC bytes (64 B per membership probe, probes counted by the oracle) / average kernel time.
    # The ADJ layout answers a node's 8 probes with ONE bucket read, so the HBM traffic is ~1/8 of that figure.
    line_bytes = int(info["adj_bucket_bytes"])
    alg_bytes_per_launch = 64.0 * probes_per_nt * contig_nt / max(launches, 1)
    avg_kernel_s = kernel_ms / max(launches, 1) * 1e-3
    achieved = alg_bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.json")
    if a.workload == "human" and sites_per_gpu == 100000 and os.path.exists(pmc):
        traffic = json.load(open(pmc))["mtgi::k_stage_a"]["hbm_read_bytes_avg"]
        traffic_src = "profiles/r01_pmc_fetch_size.json (rocprofv3 --pmc FETCH_SIZE, separate pass, same command; calibrated factor 1.000 on k_chase)"
    roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "k_stage_a", "avg_kernel_ms": kernel_ms / max(launches, 1), "launches": int(launches),
            "algorithmic_bytes_per_launch": alg_bytes_per_launch, "probes_per_contig_nt": probes_per_nt,
            "bucket_reads_per_launch": lines / max(launches, 1), "bucket_bytes": line_bytes,
            "bucket_reads_per_s": lines / max(kernel_ms, 1e-9) * 1e3,
            "avg_kernel_ms_one_batch_in_flight": st_alone["kernel_ms"] / max(st_alone["n_launches"], 1),
            "frac_one_batch_in_flight": (64.0 * probes_per_nt * st_alone["contig_nt"] / max(st_alone["kernel_ms"], 1e-9) * 1e3 / 1e9) / 8000.0,
            "note": "achieved follows SURVEY 8d (64 B per membership probe of the reference algorithm); the ADJ layout answers the ~8 probes of a node, and "
                    "the lookahead up to 15 nodes, with one 32-byte bucket read, so frac exceeds 1 by construction. The kernel is bound by dependent random "
                    "reads and the per-step VALU work, not by HBM bandwidth: see traffic (PMC) and frac_of_random_read_ceiling. With several steps in flight the "
                    "traversal of one batch shares the device with the kernels of the others, so its launches take longer than the kernel alone "
                    "(*_one_batch_in_flight: the untimed verification pass) while the whole job is faster."}
    # second kernel of the step: one abundance look-up (64 algorithmic bytes) per k-mer of source + fill
    lookups = float(idx.last_seq_bytes)  # sum over the filled sites of (insert length + 1) = k-mers of source + fill
    post_s = post_ms / max(launches, 1) * 1e-3
    post_traffic = None
    if traffic is not None:
        pj = json.load(open(pmc))
        kp, ka = pj.get("mtgi::k_post", {}), pj.get("mtgi::k_stage_a", {})
        if kp.get("hbm_read_bytes_avg") and ka.get("launches_FETCH_SIZE"):  # k_post runs as several launches per traversal launch: sum them
            post_traffic = kp["hbm_read_bytes_avg"] * kp["launches_FETCH_SIZE"] / ka["launches_FETCH_SIZE"]
    roof["post_kernel"] = {"kernel": "k_post", "bound": "hbm", "avg_kernel_ms": post_ms / max(launches, 1), "abundance_lookups_per_launch": lookups,
                           "achieved": 64.0 * lookups / post_s / 1e9 if post_s > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                           "frac": (64.0 * lookups / post_s / 1e9 / 8000.0) if post_s > 0 else 0.0, "traffic": post_traffic}
    if not a.no_ceiling:
        tb = min(int(info["device_bytes"] // 2), 16 << 30)  # the ceiling is flat beyond ~16 GB (profiles/r01_random_line_ceiling.txt)
        ms, gbps = mtg.random_line_ceiling(max(tb, 1 << 26), sites_per_gpu, 512, line_bytes)
        roof["random_read_ceiling_reads_per_s"] = gbps * 1e9 / line_bytes
        roof["frac_of_random_read_ceiling"] = roof["bucket_reads_per_s"] / roof["random_read_ceiling_reads_per_s"] if gbps else None

    out = {"metric": "breakpoints filled/sec", "value": value, "unit": "breakpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": desc, "sites_per_gpu": sites_per_gpu, "donor_sequences": S.nseq, "k": k, "max_nodes": 100, "max_length": 10000,
                      "index": "exact k-mer set of the donor, abundance = 3 + hash %% 40 (no reads simulated)", "nb_solid_kmers": int(info["nb_solid_kmers"]),
                      "index_bytes": int(info["device_bytes"]), "index_build_s": t_index, "genome_gen_s": t_gen, "steps_in_flight": a.in_flight},
           "filled": n_filled_all, "filled_per_s": n_filled_all * a.steps / elapsed,
           "filled_sequences_identical_to_truth": identical,
           "stage_ms_per_step": {"kernel": kernel_ms / a.steps, "post_kernel": post_ms / a.steps, "d2h": d2h_ms / a.steps, "host": host_ms / a.steps,
                                 "c_call": call_ms / a.steps},
           "roofline": roof, "cpu_baseline": cpu}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
