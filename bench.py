#!/usr/bin/env python3
"""bench.py -- breakpoints filled / second on the synthetic human-scale set (BASELINE.json configs[3]; configs[4] for N > 1).

One "step" = one pass of the fill hot path (Filler::gapFillFromSource for every site of a batch, reverse attempt for the unfilled ones)
through the C ABI of libmtgfill.so.  The index is built once, before the timed region, and stays resident in HBM; the batches of
sites are marshalled once and resident too (mtg_batch_prepare), so a step starts with its input in HBM (the contract's wording); the
results (C-ABI records and ASCII sequences) are in host memory when it ends.  Several distinct batches of sites are rotated through the steps.

N = 1: every step fills one batch of 100 000 sites.  N > 1 (one process per GPU, torchrun): STRONG scaling by default -- one fixed set
of sites is sharded over the ranks with shard_range, a step fills the whole set, the index is replicated, there is no data-path collective;
every batch leaves its rank in relocatable form (records AND sequences, tagged with its global batch index, written by the result kernel
straight into the gather's device buffer) and is gathered on rank 0 over RCCL while the next ones are filled; rank 0 checks every payload
of every rank where it arrives.  Two site sets are measured: 600 000 sites (every donor sequence carries one) and BASELINE config 5's
literal 100 000.

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

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PCIE_PEAK_GBS = 64.0    # PCIe 5.0 x16, one direction

WORKLOADS = {
    # name: (donor sequences, sites per batch, description)
    "human": (600000, 100000, "synthetic human-scale: 3 Gbp i.i.d. donor as 600000 x 5 kb sequences, 100000 insertion sites (50-1000 nt) per batch, k=31, max-nodes 100"),
    "ecoli": (1000, 1000, "synthetic E.coli-scale: 5 Mbp donor as 1000 x 5 kb sequences, 1000 insertion sites, k=31"),
    "human-het": (600000, 100000, "secondary, divergence-heavy: diploid donor, 300000 loci x 2 haplotypes x 5 kb with 4 heterozygous SNPs per locus, 100000 insertion sites per batch"),
    "human-indel": (600000, 100000, "secondary, general bubble code: the diploid donor of human-het with two deletions of 1-3 nt per locus in the second haplotype besides the 4 SNPs (bubbles with branches of different lengths), 100000 insertion sites per batch"),
    "human-tips": (600000, 100000, "secondary, tips and error bubbles: the haploid human-scale donor plus one erroneous fragment per three donor sequences in the index (a copy of k+1..k+45 donor nucleotides with one substitution, abundance >= 3: ~0.1 % of the k-mers), 100000 insertion sites per batch"),
    "tiny": (400, 256, "tiny smoke workload"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of the job (default: WORLD_SIZE, else 1).  N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (torch.distributed.run)")
    ap.add_argument("--full-line", action="store_true", help="print the whole detail object as the last line instead of the compact one (what the child processes of the secondary workloads hand to their parent)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the whole detail object is written (the last stdout line is the compact one)")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=os.environ.get("MTG_BENCH_WORKLOAD", "human"), choices=list(WORKLOADS))
    ap.add_argument("--sites", type=int, default=0, help="sites per batch (default: the workload's)")
    ap.add_argument("--cpu-same-sites", type=int, default=1000, help="sites (and donor sequences) of the same-algorithm CPU measurement (0: skip)")
    ap.add_argument("--no-children", action="store_true", help="skip the secondary workloads (child processes)")
    ap.add_argument("--tool-sites", type=int, default=2000000, help="sites of the tool measurement (the sites of the batches, repeated)")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches of sites rotated through the steps (N = 1 and weak scaling)")
    ap.add_argument("--nseq", type=int, default=0)
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"], help="N > 1: strong (default) = one fixed site set sharded over the ranks; weak = one batch per rank and step")
    ap.add_argument("--strong-sites", type=int, default=0, help="size of the sharded site set (default: two measurements, 600000 = every donor sequence carries a site, and BASELINE config 5's 100000)")
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks of --steps steps (0: at least 5 and at least ~1 s of timed work); the median block is reported")
    ap.add_argument("--cpu-sites", type=int, default=30000, help="sites of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-index-seqs", type=int, default=30000, help="donor sequences in the CPU baseline's index")
    ap.add_argument("--host-threads", type=int, default=-1, help="host threads per rank for the per-gap passes (default: the library's pool = CPU budget of the process, shared between the ranks)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MTG_BENCH_IN_FLIGHT", "6")),
                    help="batches in flight: caller threads issuing batches on the one index (the library runs up to six batches of an index side by side)")
    ap.add_argument("--host-strings", action="store_true", help="marshal the sites from host strings inside every step (mtg_fill_batch) instead of filling prepared, device-resident batches")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the random-line ceiling micro-benchmark")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (host-string input, sequences left in HBM, the tool, the other workloads)")
    ap.add_argument("--no-tool", action="store_true", help="skip the tool measurement (MindTheGap fill on the batches' sites)")
    return ap.parse_args()


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


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes (torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1) before anything in this process has touched a GPU, pass their output through and leave with their exit code.
    Never an exec: a process image is not replaced here."""
    import socket
    import torch  # device_count() may initialise the HIP runtime in THIS process (a build without amdsmi falls back to hipGetDeviceCount): the ranks are therefore only ever started as children, never by replacing this process
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d: this machine shows %d GPU(s); refusing to print a %d-GPU line from fewer devices" % (n, have, n))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cp = subprocess.run(cmd, env=env)
    raise SystemExit(cp.returncode)


def _num(x, digits=4):
    """a number of the compact line: rounded to `digits` significant figures (ints stay ints, None / bool / str pass)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


COMPACT_LIMIT = 6000  # bytes: what the driver's tail of stdout holds with room to spare (round 4's 24 KB line was not parsed)


def _phase_sums(phases):
    """name -> ms of the construction's phases of at least 1 ms; a phase that runs once per chunk of the input (the partitioned junction table) is its sum"""
    acc = {}
    for ph in phases:
        acc[ph["name"]] = acc.get(ph["name"], 0.0) + ph.get("ms", 0.0)
    return {k: _num(v, 3) for k, v in acc.items() if v >= 1.0}


def compact_line(out, detail_path=None):
    """The ONE line the driver parses, from the detail object: the contract's keys, a compact `roofline` (dominant fill kernel against HBM on
    the bytes the layout must move; the link the job is bound by; the index construction on SURVEY 8d's bytes), a compact `cpu_baseline`,
    `end_to_end` (config 4 literally) and one number per secondary.  Everything else is in the detail file."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    roof, cpu, ib, cfg = out.get("roofline") or {}, out.get("cpu_baseline") or {}, out.get("index_build") or {}, out.get("config") or {}
    dom = next((kk for kk in roof.get("kernels", []) if kk.get("kernel") == roof.get("dominant_kernel")), {})
    wall = next((kk for kk in roof.get("kernels", []) if str(kk.get("kernel", "")).startswith("k_stage_a")), {})
    line = {kk: out.get(kk) for kk in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _num(out.get("value"), 6), _num(out.get("ms_per_step"), 5)
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], "sites_per_step": cfg.get("sites_per_step"), "sites_per_batch": cfg.get("sites_per_batch"),
                      "sites_per_rank_and_step": cfg.get("sites_per_rank_and_step"), "site_set": cfg.get("site_set"), "k": cfg.get("k"), "max_nodes": cfg.get("max_nodes"),
                      "nb_solid_kmers": cfg.get("nb_solid_kmers"), "index_bytes": cfg.get("index_bytes"), "batches_in_flight": cfg.get("batches_in_flight"),
                      "input": cfg.get("input"), "output": str(cfg.get("output", ""))[:120]}
    line["filled"], line["sites_verified"] = out.get("filled"), out.get("sites_verified")
    line["filled_sequences_identical_to_truth"] = out.get("filled_sequences_identical_to_truth")
    if out.get("gathered_payload_verified") is not None:
        line["gathered_payload_verified"] = g(out, "gathered_payload_verified", "ok")
    # the entries of the same steps: inputs resident in HBM (= value, the contract's wording), one block of text per batch, host strings, ...
    line["value_prepared"] = _num(out.get("value"), 6)
    for kk in ("value_from_host_text", "value_from_registered_text", "value_from_host_strings", "value_sequences_left_in_hbm", "tool_sites_per_s"):
        if kk in out:
            line[kk] = _num(out.get(kk))
    line["roofline"] = {
        "bound": "hbm", "kernel": dom.get("kernel"), "achieved": _num(dom.get("achieved")), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": _num(dom.get("frac")),
        "frac_basis": "layout bytes", "bytes_per_launch": _num(dom.get("bytes_per_launch"), 6), "bytes_are": "what the implemented layout must move per launch (counters of the timed launches)",
        "avg_kernel_ms": _num(dom.get("avg_kernel_ms")), "traffic": _num(dom.get("traffic"), 6), "traffic_over_bytes": _num(dom.get("traffic_over_bytes")), "traffic_over_request_bytes": _num(dom.get("traffic_over_request_bytes")),
        "traffic_is": "replayed from profiles/ (PMC FETCH_SIZE + WRITE_SIZE of this command)" if dom.get("traffic") else None,
        "dominant_kernel": roof.get("dominant_kernel"), "dominant_kernel_frac": _num(roof.get("dominant_kernel_frac_of_hbm_peak")),
        "walk_kernel": wall.get("kernel"), "first_walk_by_light_kernel": _num((roof.get("first_walk_kernel") or {}).get("k_walk (light: simple paths only)")), "walk_kernel_frac": _num(wall.get("frac")), "walk_kernel_ms": _num(wall.get("avg_kernel_ms")),
        "walk_traffic_over_bytes": _num(wall.get("traffic_over_bytes")),
        "random_read_frac": _num(g(roof, "random_reads_of_k_stage_a", "alone", "frac_of_ceiling") or g(roof, "random_reads_of_k_stage_a", "frac_of_ceiling")),
        "sec8d_equivalent_frac_of_the_walk": _num(g(roof, "reference_algorithm_equivalent", "equivalent_GBps_of_k_stage_a") / HBM_PEAK_GBS) if g(roof, "reference_algorithm_equivalent", "equivalent_GBps_of_k_stage_a") else None,
        "sec8d_frac_of_step": _num(g(roof, "reference_algorithm_equivalent", "bytes_per_launch") / (out.get("ms_per_step") * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if g(roof, "reference_algorithm_equivalent", "bytes_per_launch") and out.get("ms_per_step") else None,
        "sec8d_note": "above 1: a step does not do SURVEY 8(d)'s 8 probes per nucleotide, the index construction did (build_frac)",
        "build_frac": _num(ib.get("sec8d_frac_over_device_seconds")), "build_device_s": _num(ib.get("device_seconds")),
        "job_bound": "pcie", "pcie_GBps": _num(roof.get("achieved")), "pcie_peak_GBps": PCIE_PEAK_GBS, "pcie_frac": _num(roof.get("frac")),
        "kernels_alone_ms": {kk: _num(v, 3) for kk, v in (roof.get("one_batch_alone_ms") or {}).items() if kk != "launches"},
    }
    line["cpu_baseline"] = {"value": _num(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"), "sample": str(cpu.get("sample", ""))[:160],
                            "identical_to_hip": cpu.get("identical_to_hip"), "same_algorithm_value": _num(cpu.get("same_algorithm_value")),
                            "reference_binary": bool(isinstance(cpu.get("reference_binary"), dict))} if cpu else None
    if ib:
        line["index_build"] = {"seconds": _num(ib.get("seconds")), "device_seconds": _num(ib.get("device_seconds")), "hipmalloc_seconds": _num(ib.get("hipmalloc_seconds")),
                               "peak_device_bytes": ib.get("peak_device_bytes"), "resident_bytes": ib.get("resident_bytes"),
                               "sec8d_bytes": _num(g(ib, "reference_algorithm_equivalent", "bytes"), 5), "sec8d_frac_over_device_seconds": _num(ib.get("sec8d_frac_over_device_seconds")),
                               "phases_ms": _phase_sums(ib.get("phases", []))}
    e2e = out.get("end_to_end")
    if isinstance(e2e, dict):
        line["end_to_end"] = {"sites": cfg.get("sites_per_batch"), "from_donor_in_hbm_s": _num(g(e2e, "from_donor_in_hbm", "seconds")), "from_container_s": _num(g(e2e, "from_container", "seconds")),
                              "cpu_port_scaled_s": _num(g(e2e, "cpu_port_same_span", "scaled_to_config4_estimate_s")),
                              "from_container_load_s": _num(g(e2e, "from_container_split", "load_container_s")), "from_container_fill_and_files_s": _num(g(e2e, "from_container_split", "fill_and_files_s")),
                              "index_build_s": _num(g(e2e, "from_donor_in_hbm", "index_build_s")), "fill_and_files_s": _num(g(e2e, "from_donor_in_hbm", "tool_fill_and_files_s")),
                              "identical_to_truth": bool(g(e2e, "from_donor_in_hbm", "sequences_identical_to_truth") and g(e2e, "from_container", "sequences_identical_to_truth")) if "from_container" in e2e else None,
                              "error": e2e.get("error")}
    sec = {}
    for kk, v in out.items():
        if kk.startswith("secondary_") and isinstance(v, dict):
            ident = v.get("identical_to_oracle")
            if ident is None:
                ident = g(v, "oracle_sample", "identical_to_hip")
            sec[kk[len("secondary_"):]] = {"value": _num(v.get("value")), "ratio": _num(v.get("ratio_to_headline"), 3), "identical_to_oracle": ident, **({"unit": v["unit"]} if v.get("unit") not in (None, "breakpoints/s") else {}),
                                           **({"error": str(v["error"])[:80]} if "error" in v else {})}
    if sec:
        line["secondary"] = sec
    sh = g(out, "eight_gpu_outlook_from_one_rank", "shards")
    if isinstance(sh, dict):
        line["eight_gpu_outlook"] = {kk: {"ms_per_step": _num(v.get("ms_per_step_of_one_rank")), "efficiency": _num(v.get("efficiency_vs_8x_the_headline"), 3)} for kk, v in sh.items() if isinstance(v, dict)}
    s5 = out.get("strong_scaling_config5_literal")
    if isinstance(s5, dict):
        line["config5_literal"] = {"site_set": s5.get("site_set"), "value": _num(s5.get("value"), 6), "ms_per_step": _num(s5.get("ms_per_step")), "identical": s5.get("filled_sequences_identical_to_truth"),
                                   "gathered_ok": g(s5, "gathered_payload_verified", "ok")}
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    # the line must stay readable to the driver whatever a later round adds to the detail object: shed the optional parts, largest first
    for drop in ("eight_gpu_outlook", "secondary", "index_build", "end_to_end"):
        if len(json.dumps(line)) <= COMPACT_LIMIT:
            break
        line.pop(drop, None)
    return line


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus or 1) > 1:
        spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU): the line would carry the wrong n_gpus" % (a.gpus, world))
    import torch
    import torch.distributed as dist
    import mindthegap_amd as mtg
    from mindthegap_amd import lib as L
    from mindthegap_amd.shard import PipelinedGather, SlottedGather, gather_slots_for, shard_range, strong_plan, wire_check
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
    # test hook: MTG_BENCH_FORCE_GATHER=1 runs the N > 1 result path (process group, pipelined gather of every batch) in a world of one
    # rank, the only RCCL configuration a one-GPU box offers
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
    tips = 1.0 / 3.0 if a.workload == "human-tips" else 0.0
    nloci0 = (a.nseq or nseq0) // (2 if het else 1)
    scaling = a.scaling if a.scaling != "auto" else ("strong" if dist_on else "weak")
    if not dist_on:
        scaling = "weak"  # one GPU: a step is one batch
    k = 31

    # ---------------------------------------------------------------- the site sets and how they are dealt out
    # site i = the insertion of donor sequence i.  A configuration = (total sites, this rank's batches as site ranges, sites per step of the
    # whole job, rotate).  Strong scaling measures two site sets (the larger one first: it is the headline and sizes the donor).
    configs = []
    if scaling == "strong":
        big = min(nloci0, 6 * batch_sites)
        sets = [a.strong_sites] if a.strong_sites else [big] + ([batch_sites] if big != batch_sites else [])
        for total_sites in sets:
            # global batch index = position of the batch in site order: rank r's batches follow those of the ranks before it (shard.py: strong_plan,
            # exercised at world sizes 2 and 8 by tests/test_distributed_cpu.py)
            pl_ = strong_plan(total_sites, batch_sites, rank, world)
            configs.append(dict(total=total_sites, my=pl_["my"], step_sites=total_sites, rotate=False, n_batches_job=pl_["n_batches_job"], max_per_rank=pl_["max_per_rank"]))
    else:
        nb = max(1, min(a.batches, nloci0 // (batch_sites * world)))
        my = [(rank * nb + b, (rank * nb + b) * batch_sites, (rank * nb + b + 1) * batch_sites) for b in range(nb)]
        configs.append(dict(total=world * nb * batch_sites, my=my, step_sites=batch_sites * world, rotate=True, n_batches_job=world * nb, max_per_rank=nb))
    n_sites_total = max(c["total"] for c in configs)
    nseq = max(a.nseq or nseq0, n_sites_total * (2 if het else 1))

    # ---------------------------------------------------------------- synthetic donor genome + index (not timed as "fill")
    t0 = time.time()
    S = SynthSet(nseq=nseq, n_sites=n_sites_total, seed=1, k=k, het_snps=het, het_indels=indels, tips=tips)
    t_gen = time.time() - t0
    t0 = time.time()
    pw, po, pl, pn = S.packed()
    w = torch.from_numpy(pw.view(np.int64)).to(dev)
    wo = torch.from_numpy(po.view(np.int64)).to(dev)
    ln = torch.from_numpy(pl.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    t_donor_up = time.time() - t0  # the packed donor (0.9 GB of pageable numpy memory) to the device: not part of the construction
    t0 = time.time()
    idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), pn, S.total_kmers_upper_bound, k, 3, 0)
    torch.cuda.synchronize()
    t_index = time.time() - t0  # wall time of the library call: the donor is in HBM when it starts, the index is ready when it returns
    info = idx.info()
    bprof = idx.build_profile()
    del w, wo, ln, pw
    torch.cuda.empty_cache()

    host_threads = a.host_threads if a.host_threads >= 0 else 0  # 0 = the library's worker pool as it sized itself
    params = mtg.FillParams(max_nodes=100, max_depth=10000, nb_host_threads=host_threads)
    STAT_KEYS = dict(kernel_ms=0.0, post_kernel_ms=0.0, emit_kernel_ms=0.0, index_lines=0, contig_nt=0, n_launches=0, host_ms=0.0, d2h_ms=0.0, total_ms=0.0, store_runs=0, run_nt=0,
                     post_lines=0, contig_words=0, coverage_kmers=0, dense_words=0, seq_bytes=0, copy_kernel_ms=0.0, copy_words=0, copy_cmds=0, coverage_direct_kmers=0,
                     finish_kernel_ms=0.0, n_parked_gaps=0, n_lean_gaps=0, lean_kernel_ms=0.0, copy_words_executed=0, copy_cmds_executed=0, post_scanned_words=0, device_span_ms=0.0, n_light_walks=0, n_branching_gaps=0, gaps=0)

    class B:
        pass

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(cfg):
        """the timed blocks of one configuration; returns what the report needs"""
        acc = dict(STAT_KEYS)
        acc_lock = threading.Lock()
        batches = []
        for (gidx, s0, s1) in cfg["my"]:
            b = B()
            b.gidx = gidx
            b.gaps, b.expected = [], []
            for i in range(s0, s1):
                l, r, ins = S.site(i)
                b.gaps.append(mtg.Gap(l, r, [(r, S.site_name(i), False)]))
                b.expected.append(ins)
            b.strings = mtg.Index.prepare_gaps(b.gaps)
            b.prepared = b.strings if a.host_strings else idx.prepare_batch(b.strings, params)
            b.digest = hashlib.sha256(("\0".join(b.expected) + "\0").encode()).hexdigest() if b.expected else hashlib.sha256(b"").hexdigest()
            b.n = s1 - s0
            b.s0 = s0
            batches.append(b)
        st = dict(pg=None, hbm_pool=None, seen={}, bad=[], keep=None)
        full_wire_check = os.environ.get("MTG_BENCH_WIRE_CHECK", "") == "full"
        rotate = cfg["rotate"]

        def on_arrival(items):
            """rank 0, for every completed gather: every rank's payload is validated where it arrived (device memory with RCCL): header, sizes,
            the checksum of its records and sequences recomputed by tensor arithmetic.  keep: the payloads of a verification step come to the host."""
            for it in items:
                r, tag, t = it
                if tag < 0:
                    continue
                # inside the timed steps the header (magic, tag, sizes) of every payload; the checksum of every payload's records and sequences is
                # recomputed in the verification step (keep is set), which sees every batch of every rank once (MTG_BENCH_WIRE_CHECK=full: always)
                h = wire_check(t, checksum=st["keep"] is not None or full_wire_check, head=getattr(it, "head", None))
                if not h["ok"] or h["tag"] != tag:
                    st["bad"].append((r, tag, h.get("why")))
                st["seen"][(r, tag)] = h.get("checksum")
                if st["keep"] is not None:
                    st["keep"][tag] = t.cpu().numpy().copy()

        def fill(b, prepared=None, want_seqs=False, record=False):
            """one batch: forward attempt for every site, reverse attempt (src/Filler.cpp:669-680) for the unfilled ones.  The results are
            host records and ASCII sequences when the call returns.  N > 1: the batch also leaves in relocatable form -- records and
            sequences, tagged with its global batch index -- written by the result kernel into the gather's device buffer, and is gathered
            on rank 0 over RCCL / xGMI while the next batches run."""
            prepared = b.prepared if prepared is None else prepared
            seqs = None
            pg = st["pg"]
            if pg is not None and not want_seqs:
                j, buf = pg.acquire()
                if pg.on_gpu and isinstance(prepared, mtg.Batch):
                    ptr, cap_b = pg.device_area(j)
                    h, nf, nbytes = idx.fill_prepared_wire_device(prepared, b.gidx, ptr, cap_b, params)
                    pg.submit(nbytes, j, on_device=True, tag=b.gidx)
                else:  # host tensors (gloo dry run) or host-string input: the finished result set is serialised into the page-locked gather buffer
                    h, nf, _ = idx.fill_prepared(prepared, params, want_seqs=False)
                    payload = L.results_to_wire(h, b.gidx, buf)
                    pg.submit(payload.size, j, tag=b.gidx)
            elif st["hbm_pool"] is not None and not want_seqs and isinstance(prepared, mtg.Batch):
                t = st["hbm_pool"].get()
                h, nf, nbytes = idx.fill_prepared_serial_device(prepared, t.data_ptr(), t.numel(), params)
                st["hbm_pool"].put(t)
            else:
                h, nf, seqs = idx.fill_prepared(prepared, params, want_seqs=want_seqs)
            stt = mtg.last_batch_stats()
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
                        stt[key] += st2[key]
            if record:
                with acc_lock:
                    for key in acc:
                        if key in stt:
                            acc[key] += stt[key]
                    acc["gaps"] += b.n
            return n_filled, seqs

        def pad_gathers(count):
            """a rank with fewer batches than the others sends empty payloads, so that every rank issues the same number of collectives"""
            pg = st["pg"]
            for _ in range(count):
                j, _buf = pg.acquire()
                pg.submit(0, j, tag=-1)

        # The caller threads of the job: a.in_flight of them, started ONCE (the reference's Dispatcher keeps its worker threads for the whole run
        # too).  Round 4 started and joined six Python threads inside every timed block: 0.5 ms of a 15 ms block of 20 steps -- 3-4 % of the
        # headline, and a third of a block of 12 500-site steps (the 8-GPU outlook).
        import queue as _queue
        tasks = _queue.Queue()
        done_cv = threading.Condition()
        pending = [0]
        errors = []

        def caller():
            torch.cuda.set_device(local_rank)  # the current device is a per-thread setting
            while True:
                item = tasks.get()
                if item is None:
                    return
                fn, args = item
                try:
                    fn(*args)
                except BaseException as e:  # surfaced on the main thread
                    errors.append(e)
                with done_cv:
                    pending[0] -= 1
                    if pending[0] == 0:
                        done_cv.notify_all()

        callers = [threading.Thread(target=caller, daemon=True) for _ in range(max(1, a.in_flight))]
        for t in callers:
            t.start()

        def run_tasks(items):
            """every item (function, arguments) on the caller threads; returns when all are done"""
            if not items:
                return
            with done_cv:
                pending[0] += len(items)
            for it_ in items:
                tasks.put(it_)
            with done_cv:
                while pending[0]:
                    done_cv.wait()
                errs = errors[:]  # taken and cleared under the lock: a failure of one measurement must not fail every later one (advisor, round 5)
                del errors[:]
            if errs:
                raise errs[0]

        def run_block(count, record, first_step=0, host_strings=False, host_text=False):
            """`count` steps, a.in_flight batches in flight: the caller threads take the next batch off a shared queue, like the reference's
            Dispatcher threads take the next group of records; every batch is complete when this returns"""
            work = []
            for s in range(first_step, first_step + count):
                work += [batches[s % len(batches)]] if rotate else list(batches)
            run_tasks([(lambda b: fill(b, prepared=b.strings if host_strings else b.text if host_text else None, record=record), (b,)) for b in work])
            if st["pg"] is not None and not rotate:
                pad_gathers(count * (cfg["max_per_rank"] - len(batches)))

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
            # a payload holds records (80 bytes per gap) and sequences; the capacity is agreed on (the gather moves whole buffers)
            need = max([0] + [sum(len(e) + 1 for e in b.expected) + 96 * b.n + 4096 for b in batches])
            cap = torch.tensor([need], dtype=torch.int64, device=cdev)
            dist.all_reduce(cap, op=dist.ReduceOp.MAX)
            # several payloads per collective when a rank has several batches per step (the orchestration of a gather, not its bytes, is what a rank pays per batch)
            slots, depth = gather_slots_for(cfg["max_per_rank"], a.in_flight)  # measured on one RCCL rank: 94 M/s with 1 slot, 76 with 3, 101 with 6 (one collective per step)
            slots = int(os.environ.get("MTG_BENCH_GATHER_SLOTS", "0")) or slots
            depth = int(os.environ.get("MTG_BENCH_GATHER_DEPTH", "0")) or (max(a.in_flight, 1) + 1 if slots == 1 else max(3, (max(a.in_flight, 1) + slots - 1) // slots + 2))
            cap_p = int(cap.item()) * 51 // 50 + (1 << 16)
            if slots > 1:
                st["pg"] = SlottedGather(cap_p, slots=slots, dst=0, device=cdev, depth=depth, on_arrival=on_arrival if rank == 0 else None)
            else:
                st["pg"] = PipelinedGather(cap_p, dst=0, device=cdev, depth=depth, on_arrival=on_arrival if rank == 0 else None)
            st["gather_slots"] = slots
        run_block(a.warmup, False)

        def timed_block(first_step):
            barrier()
            t0 = time.perf_counter()
            run_block(a.steps, True, first_step)
            if st["pg"] is not None:
                st["pg"].drain()  # the gathers still in flight belong to the timed steps
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
        n_arrived_timed = len(st["seen"])

        # ---------------------------------------------------------------- verification (untimed)
        # (1) N > 1: one more step whose payloads rank 0 brings to the host: EVERY batch of EVERY rank is rebuilt from its payload
        #     (mtg_results_from_wire validates it) and the sequences in it are compared with the digest the owning rank computed from the truth;
        # (2) every distinct batch once through the plain entry, sequences against the truth.
        gathered = None
        if st["pg"] is not None:
            st["keep"] = {} if rank == 0 else None
            run_block(1, False)
            st["pg"].drain()
            mine = [(b.gidx, b.digest, b.n) for b in batches]
            everyone = [None] * max(world, 1)
            dist.all_gather_object(everyone, mine)
            if rank == 0:
                want = {g: (d, n) for lst in everyone for (g, d, n) in lst}
                ok, checked = not st["bad"], 0
                for g, (d, n) in want.items():
                    p = st["keep"].get(g)
                    if p is None:
                        ok = False
                        continue
                    hd = L.wire_header(p)
                    wr = L.WireResults(p)  # sizes, offsets, checksum
                    o_s = 64 + 40 * hd["n_gaps"] + 40 * hd["n_filled"]
                    seq = p[o_s: o_s + hd["seq_bytes"]].tobytes()
                    good = hd["n_gaps"] == n and hd["tag"] == g and bool(het or tips or hashlib.sha256(seq).hexdigest() == d)
                    ok = ok and good
                    checked += 1
                    wr.close()
                gathered = {"ok": bool(ok and checked == len(want) == cfg["n_batches_job"]), "batches_checked": checked, "batches_of_the_job": cfg["n_batches_job"],
                            "payloads_validated_on_arrival_in_the_timed_blocks": n_arrived_timed, "bad": st["bad"][:5],
                            "how": "every rank's payload of every batch: header (magic, tag, sizes) where it arrived in every gather of the timed blocks; in one untimed step every payload's checksum recomputed on the device, the batch rebuilt with mtg_results_from_wire (validates again on the host) and its sequences compared with the owning rank's digest of the truth"}
            st["keep"] = None
        pg_saved, st["pg"] = st["pg"], None
        n_filled_rank, identical = 0, True
        for b in batches:
            nf_b, seqs = fill(b, want_seqs=True)
            n_filled_rank += nf_b
            if not het and not tips:  # size-independent parity property at full size: every site is filled with exactly its inserted sequence
                identical = identical and hashlib.sha256(seqs.tobytes().replace(b"\n", b"\0")).hexdigest() == b.digest
        if het or tips:
            identical = None  # alleles / error bubbles: the truth is a mix, checked against the oracle (cpu_baseline, tests)
        # the kernels' own times: every distinct batch once more (twice when there are few), ONE batch on the device at a time; the averages are what
        # every roofline fraction below is computed from (under six batches in flight a kernel's events also span other batches' workgroups)
        st_alone = None
        if batches:  # every rank for itself (no collective in it); rank 0's is reported
            runs = []
            mtg.tuning_set("KERNEL_TIMERS", "1")  # an event between the kernels: each kernel's own time in the statistics (off in the timed blocks: three events per batch instead of nine)
            for b in (batches * 2 if len(batches) < 4 else batches):
                if not isinstance(b.prepared, mtg.Batch):
                    break
                torch.cuda.synchronize()
                h_a, _nf_a, _ = idx.fill_prepared(b.prepared, params, want_seqs=False)
                runs.append(mtg.last_batch_stats())
                idx.free_results(h_a)
            mtg.tuning_set("KERNEL_TIMERS", None)
            if runs:
                st_alone = {key: sum(r[key] for r in runs) for key in runs[0]}
                st_alone["runs"] = len(runs)
        n_sites_rank = sum(b.n for b in batches)
        if dist_on:
            v = torch.tensor([n_filled_rank, n_sites_rank, 1 if identical in (True, None) else 0], device=cdev, dtype=torch.int64)
            dist.all_reduce(v)
            n_filled_all, n_sites_all, ident_all = int(v[0].item()), int(v[1].item()), int(v[2].item()) == world
            if identical is not None:
                identical = ident_all
        else:
            n_filled_all, n_sites_all = n_filled_rank, n_sites_rank
        def stop_callers():
            """the caller threads of this configuration leave (one sentinel each); the report's later measurements start their own through run_tasks no more"""
            for _ in callers:
                tasks.put(None)
        return dict(acc=acc, batches=batches, st=st, pg=pg_saved, times=times, elapsed=elapsed, gathered=gathered, identical=identical, st_alone=st_alone, stop_callers=stop_callers,
                    n_filled=n_filled_all, n_sites=n_sites_all, fill=fill, run_block=run_block, run_tasks=run_tasks, value=cfg["step_sites"] * a.steps / elapsed)

    results = []
    for ci, c in enumerate(configs):
        if ci > 0:  # the device copies of the previous configuration's batches make room
            for b in results[-1]["batches"]:
                if hasattr(b.prepared, "close"):
                    b.prepared.close()
            if results[-1]["pg"] is not None:
                results[-1]["pg"].drain()
                results[-1]["pg"] = None
            torch.cuda.empty_cache()
        results.append(measure(c))
    R0, cfg0 = results[0], configs[0]
    if rank != 0:
        if dist_on:
            for r in results:
                if r["pg"] is not None:
                    r["pg"].drain()
            dist.destroy_process_group()
        return

    acc, batches, times, elapsed, value, st_alone = R0["acc"], R0["batches"], R0["times"], R0["elapsed"], R0["value"], R0["st_alone"]
    step_sites = cfg0["step_sites"]
    sites_per_rank_step = sum(b.n for b in batches) if not cfg0["rotate"] else batch_sites
    secondary = {}
    if len(results) > 1:  # BASELINE config 5 literally: the 100 000-site set sharded over the ranks
        R1, c1 = results[1], configs[1]
        secondary["strong_scaling_config5_literal"] = {"site_set": c1["total"], "sites_per_rank": sum(b.n for b in R1["batches"]), "value": R1["value"], "unit": "breakpoints/s",
                                                       "ms_per_step": R1["elapsed"] / a.steps * 1e3, "blocks": len(R1["times"]), "gathered_payload_verified": R1["gathered"],
                                                       "filled": R1["n_filled"], "sites_verified": R1["n_sites"], "filled_sequences_identical_to_truth": R1["identical"]}
    single = world == 1 and not dist_on and len(results) == 1
    # ---------------------------------------------------------------- secondary: the same steps from host strings (marshalling + upload inside the step)
    if single and not a.no_secondary and not a.host_strings and batches:
        R0["run_block"](max(a.warmup, 1), False, host_strings=True)
        ts = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R0["run_block"](a.steps, False, r * a.steps, host_strings=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        secondary["value_from_host_strings"] = batch_sites * a.steps / float(np.median(ts))
        secondary["value_from_host_strings_note"] = ("same steps through mtg_fill_batch: the sites are marshalled from the caller's strings (three C strings per site, wherever the caller has them) and "
                                                     "uploaded inside every step; median of 3 blocks, %d pool threads" % (int(os.environ.get("MTG_POOL_THREADS", "0")) or mtg.cpu_budget()))

    # ---------------------------------------------------------------- secondary: the same steps from one block of text per batch (mtg_fill_text): the host copies the
    # block and its offset arrays into page-locked memory, the device encodes sources, patterns and dictionary keys (k_marshal_text, k_marshal_targets)
    if single and not a.no_secondary and not a.host_strings and batches:
        for b in batches:
            b.text = mtg.TextGaps(b.gaps)  # the block a breakpoint-file reader would hold; built outside the steps, like the strings above
        R0["run_block"](max(a.warmup, 1), False, host_text=True)
        ts = []
        for r in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R0["run_block"](a.steps, False, r * a.steps, host_text=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        _, seqs_t = R0["fill"](batches[0], prepared=batches[0].text, want_seqs=True)
        ok_text = (hashlib.sha256(seqs_t.tobytes().replace(b"\n", b"\0")).hexdigest() == batches[0].digest) if not (het or tips) else None
        secondary["value_from_host_text"] = batch_sites * a.steps / float(np.median(ts))
        secondary["value_from_host_text_note"] = ("same steps through mtg_fill_text: every step copies its sites' text block (%d bytes for %d sites) and offset arrays into page-locked memory, uploads "
                                                  "them and encodes on the device; median of 3 blocks, %d pool threads; sequences of one batch identical to the truth: %s"
                                                  % (len(batches[0].text.text), batches[0].n, int(os.environ.get("MTG_POOL_THREADS", "0")) or mtg.cpu_budget(), ok_text))
        # the same with the blocks page-locked by the caller once (mtg_host_register): no copy of the block inside the steps
        try:
            for b in batches:
                b.text.register()
            R0["run_block"](max(a.warmup, 1), False, host_text=True)
            ts = []
            for r in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                R0["run_block"](a.steps, False, r * a.steps, host_text=True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            _, seqs_t = R0["fill"](batches[0], prepared=batches[0].text, want_seqs=True)
            ok_reg = (hashlib.sha256(seqs_t.tobytes().replace(b"\n", b"\0")).hexdigest() == batches[0].digest) if not (het or tips) else None
            secondary["value_from_registered_text"] = batch_sites * a.steps / float(np.median(ts))
            secondary["value_from_registered_text_note"] = ("mtg_fill_text on blocks the caller has page-locked once (mtg_host_register): the block goes up from where it is, only the offset "
                                                            "arrays are copied inside the step; sequences of one batch identical to the truth: %s" % ok_reg)
        except Exception as e:
            secondary["value_from_registered_text"] = None
            secondary["value_from_registered_text_note"] = "failed: " + repr(e)[:200]
        finally:
            for b in batches:
                b.text.unregister()
        for b in batches:
            b.text = None

    # ---------------------------------------------------------------- secondary: the sequences left in HBM (a consumer on the device, or the send buffer of a gather):
    # records still come to the host, the ASCII -- three quarters of the result bytes -- does not cross PCIe
    if single and not a.no_secondary and not a.host_strings and batches:
        import queue
        cap_b = max(sum(len(e) + 1 for e in b.expected) for b in batches) * 5 // 4 + (1 << 20)
        pool = queue.Queue()
        for _ in range(max(1, a.in_flight)):
            pool.put(torch.empty(cap_b, dtype=torch.uint8, device=dev))
        R0["st"]["hbm_pool"] = pool
        R0["run_block"](max(a.warmup, 1), False)
        ts = []
        for r in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R0["run_block"](a.steps, False, r * a.steps)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = pool.get()
        b0 = batches[0]
        h0, nf0, nb0 = idx.fill_prepared_serial_device(b0.prepared, t.data_ptr(), t.numel(), params)
        idx.free_results(h0)
        ok_hbm = hashlib.sha256(t[:nb0].cpu().numpy().tobytes()).hexdigest() == b0.digest if not (het or tips) else None
        R0["st"]["hbm_pool"] = None
        del pool, t
        secondary["value_sequences_left_in_hbm"] = batch_sites * a.steps / float(np.median(ts))
        secondary["value_sequences_left_in_hbm_note"] = ("same steps through mtg_fill_prepared_serial_device: records to the host, ASCII sequences into a device buffer of the caller "
                                                         "(median of 5 blocks; one buffer verified against the truth: %s)" % ok_hbm)

    # ---------------------------------------------------------------- secondary: what ONE rank of an 8-GPU job does per step -- the shard sizes of strong scaling at N = 8
    # (75 000 sites per rank of the 600 000-site set; 12 500 of BASELINE config 5's literal 100 000), filled as prepared batches with six in flight, no gather:
    # the fixed costs of a launch (eight kernels of 45-60 us floors, the copies' latencies) are what an 8-GPU run pays 8 x per step
    if single and not a.no_secondary and not a.host_strings and batches and batch_sites >= 100000:
        outlook = {}
        for per_rank in (75000, 12500):
            try:
                sub = idx.prepare_batch(mtg.Index.prepare_gaps(batches[0].gaps[:per_rank]), params)
                def run_sub(count):
                    def one():
                        h_s, _nf, _ = idx.fill_prepared(sub, params, want_seqs=False)
                        idx.free_results(h_s)
                    R0["run_tasks"]([(one, ())] * count)
                run_sub(12)
                # blocks long enough for the steady state: a block of 20 steps of 0.1 ms is mostly the pipeline filling and draining (six batches in
                # flight, 0.3 ms from a batch's first kernel to its results on the host)
                sub_steps = max(a.steps, 240)
                reps = []
                for _ in range(5):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); run_sub(sub_steps); torch.cuda.synchronize(); reps.append(time.perf_counter() - t0)
                t_b = float(np.median(reps)) / sub_steps
                outlook[str(per_rank)] = {"sites_per_rank_and_step": per_rank, "ms_per_step_of_one_rank": t_b * 1e3, "sites_per_s_of_one_rank": per_rank / t_b,
                                          "times_8_if_the_ranks_do_not_interfere": 8 * per_rank / t_b, "efficiency_vs_8x_the_headline": (8 * per_rank / t_b) / (8 * value) if value else None}
                sub.close()
            except Exception as e:
                outlook[str(per_rank)] = {"error": repr(e)[:200]}
        secondary["eight_gpu_outlook_from_one_rank"] = {"what": "one rank's share of a strong-scaling step at N = 8, measured on this one GPU in its steady state (blocks of at least 240 steps; prepared batches, six in flight, results to host memory, NO gather and no second rank: "
                                                                "an upper bound of what eight ranks reach together; the gather's per-batch cost is in profiles/r04_dry_one_rank_rccl.json)", "shards": outlook}

    # ---------------------------------------------------------------- secondary: the tool.  `MindTheGap fill -bkpt` on the sites of the batches, through the library's
    # own tool entry on the index that is already in HBM (mtg_fill_main_on_index = Filler::execute behind the graph load: the load of a 36 GB
    # index file is not part of the rate), output files on a memory-backed file system when there is one
    if single and not a.no_secondary and not a.no_tool and batches:
        try:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            with tempfile.TemporaryDirectory(dir=base) as d:
                bk = os.path.join(d, "sites.breakpoints")
                site_ids = [i for b in batches for i in range(b.s0, b.s0 + b.n)]
                S.write_breakpoints(bk, site_ids)
                reps = max(1, a.tool_sites // max(len(site_ids), 1))  # a run long enough that the first and the last batch (nothing else in flight) do not set the rate
                if reps > 1:
                    one = open(bk, "rb").read()
                    with open(bk, "wb") as f:
                        for _ in range(reps):
                            f.write(one)
                os.environ["MTG_TOOL_QUIET"] = "1"  # no summary on stdout: this process prints one JSON line
                idx.fill_main(["-bkpt", bk, "-out", os.path.join(d, "warm")])  # page cache, the workspaces of the tool's host threads
                t0 = time.perf_counter()
                rc_tool = idx.fill_main(["-bkpt", bk, "-out", os.path.join(d, "tool")])
                el = time.perf_counter() - t0
                fa = open(os.path.join(d, "tool.insertions.fasta"), "rb").read()
                seqs = b"\0".join(l for l in fa.split(b"\n") if l and not l.startswith(b">")) + b"\0"
                want = hashlib.sha256(("\0".join(e for b in batches for e in b.expected) + "\0").encode()).hexdigest()
                out_bytes = sum(os.path.getsize(os.path.join(d, "tool" + e)) for e in (".insertions.fasta", ".info.txt", ".insertions.vcf"))
                if reps > 1:
                    seqs = seqs[: len(seqs) // reps] if seqs == seqs[: len(seqs) // reps] * reps else b"repetitions differ"
                secondary["tool_sites_per_s"] = len(site_ids) * reps / el
                secondary["tool"] = {"sites": len(site_ids) * reps, "distinct_sites": len(site_ids), "seconds": el, "exit_code": rc_tool, "output_bytes": out_bytes, "output_GBps": out_bytes / el / 1e9, "input_bytes": os.path.getsize(bk),
                                     "sequences_identical_to_truth": (hashlib.sha256(seqs).hexdigest() == want) if not (het or tips) else None, "output_dir": "memory-backed (/dev/shm)" if base else "temporary directory",
                                     "what": "MindTheGap fill -bkpt <sites> -out <prefix> on the resident index: breakpoint file mapped, batches of 100000 sites handed out as a stream, 3 host threads per device "
                                             "copy and parse their batch, pass its text to mtg_fill_text (marshalled on the device), format FASTA / info / VCF in pieces on the worker pool; the pieces are "
                                             "placed in input order and written with pwrite by writer threads; wall time of the whole call"}
                # ---- end to end, BASELINE config 4 LITERALLY (100 000 sites): what a user of the tool waits for.  (a) index construction from the donor
                # in HBM (index_build above) + `MindTheGap fill -bkpt` on it, files written; (b) `MindTheGap fill -graph <container> -bkpt`: container read,
                # index derived on the device, fill, files written -- one call of the tool's entry point, wall time.
                if a.workload == "human" and not os.environ.get("MTG_BENCH_NO_E2E"):
                    try:
                        ids1 = site_ids[:batch_sites]
                        bk1 = os.path.join(d, "cfg4.breakpoints")
                        S.write_breakpoints(bk1, ids1)
                        t0 = time.perf_counter()
                        rc_a = idx.fill_main(["-bkpt", bk1, "-out", os.path.join(d, "e2e_a")])
                        t_tool1 = time.perf_counter() - t0
                        cont = os.path.join(d, "cfg4.mtgidx")
                        t0 = time.perf_counter()
                        idx.save(cont)
                        t_save = time.perf_counter() - t0
                        t0 = time.perf_counter()
                        rc_b = mtg.fill_main(["-graph", cont, "-bkpt", bk1, "-out", os.path.join(d, "e2e_b")])
                        t_graph = time.perf_counter() - t0
                        # the same span in its parts (verdict r5, item 8): the container read and the tables derived (Index.load), then the tool on the loaded index
                        split = None
                        try:
                            t0 = time.perf_counter()
                            idx2 = mtg.Index.load(cont)
                            t_load = time.perf_counter() - t0
                            t0 = time.perf_counter()
                            rc_c = idx2.fill_main(["-bkpt", bk1, "-out", os.path.join(d, "e2e_c")])
                            t_fill2 = time.perf_counter() - t0
                            idx2.close()
                            split = {"load_container_s": max(t_graph - t_fill2, 0.0), "fill_and_files_s": t_fill2, "load_measured_in_a_call_of_its_own_s": t_load, "exit_code": rc_c,
                                     "note": "fill_and_files_s: MindTheGap fill -bkpt on an index loaded by Index.load (a second run); load_container_s = the one-call figure minus it (file read, store uploaded, tables derived on the device); "
                                             "the load timed in a call of its own happens while the first index is still resident: its 48 GB hipMalloc is a property of the box on the day (0.3 ms ... 4 s, profiles/r04_hipmalloc_latency.txt)"}
                        except Exception as e2:
                            split = {"error": repr(e2)[:200]}
                        want1 = hashlib.sha256(("\0".join(batches[0].expected[:len(ids1)]) + "\0").encode()).hexdigest()

                        def fasta_ok(prefix):
                            fa1 = open(prefix + ".insertions.fasta", "rb").read()
                            return hashlib.sha256(b"\0".join(l for l in fa1.split(b"\n") if l and not l.startswith(b">")) + b"\0").hexdigest() == want1
                        out_b = sum(os.path.getsize(os.path.join(d, "e2e_b" + e)) for e in (".insertions.fasta", ".info.txt", ".insertions.vcf"))
                        secondary["end_to_end"] = {"config": "BASELINE configs[3] literally: %d sites, 3 Gbp donor, k=31, max-nodes 100; input = breakpoint file, output = FASTA + info + VCF files (%s)" % (len(ids1), "memory-backed /dev/shm" if base else "temporary directory"),
                                                   "from_donor_in_hbm": {"seconds": t_index + t_tool1, "index_build_s": t_index, "tool_fill_and_files_s": t_tool1, "exit_code": rc_a, "sequences_identical_to_truth": fasta_ok(os.path.join(d, "e2e_a")),
                                                                         "what": "mtg_index_create_from_packed_device (index_build) + MindTheGap fill -bkpt on the resident index"},
                                                   "from_container": {"seconds": t_graph, "exit_code": rc_b, "container_bytes": os.path.getsize(cont), "container_write_s_untimed": t_save, "output_bytes": out_b,
                                                                      "sequences_identical_to_truth": fasta_ok(os.path.join(d, "e2e_b")),
                                                                      "what": "MindTheGap fill -graph <container v3> -bkpt <sites> -out <prefix>: ONE call -- container read, unitig store uploaded, tables derived on the device, 100 000 fills, three files written"},
                                                   "from_container_split": split,
                                                   "sites_per_s_from_container": len(ids1) / t_graph}
                    except Exception as e:
                        secondary["end_to_end"] = {"error": repr(e)[:300]}
        except Exception as e:  # the headline line does not depend on it
            secondary["tool"] = {"error": repr(e)[:300]}

    # ---------------------------------------------------------------- CPU baseline (oracle = "port"), bounded sample, rank 0, N = 1 only
    cpu = None
    probes_per_nt = 8.0
    if single and a.cpu_sites > 0 and batches:
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
            if tips:  # the erroneous fragments copied from the sampled donor sequences: the tips and error bubbles their walks meet
                seqs_ascii += [S.extra_ascii(int(j)) for j in np.nonzero(S.extra_rows < nidx)[0]]
        t0 = time.perf_counter()
        oidx = oracle_lib.Index.from_sequences(seqs_ascii, k, 3, 0)
        t_oidx = time.perf_counter() - t0
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
               "seconds": ost["seconds"], "index_seconds": t_oidx, "index_kmers": len(oidx), "identical_to_hip": cpu_seqs == hip_seqs and len(hip_seqs) > 0,
               "reference_binary": reference_binary() or "no MindTheGap install on this machine (BASELINE.md section 2): the port is timed"}
        oidx.close()
        # the product's own algorithm on the host cores (tests/emu build without its cross-checks, one process per core): what a CPU does with the
        # unitig-store walk the GPU runs -- the GPU/CPU ratio of the same algorithm, next to the ratio against the reference's algorithm above
        if a.cpu_same_sites > 0 and not het:
            try:
                nsa = min(a.cpu_same_sites, b0.n)
                o2 = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(nsa)], k, 3, 0)  # k-mer counting only (the emulated index is built from the counts)
                km2, ct2 = o2.export()
                o2.close()
                with tempfile.TemporaryDirectory() as d:
                    np.save(os.path.join(d, "km.npy"), km2)
                    np.save(os.path.join(d, "ct.npy"), ct2)
                    json.dump([(g.source, g.target, g.targets[0][1]) for g in b0.gaps[:nsa]], open(os.path.join(d, "gaps.json"), "w"))
                    procs = max(1, min(cores, 16))
                    cp = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu_cpu_rate.py"), os.path.join(d, "km.npy"), os.path.join(d, "ct.npy"), os.path.join(d, "gaps.json"), str(procs)],
                                        capture_output=True, text=True, timeout=600)
                    sa = json.loads(cp.stdout.strip().splitlines()[-1])
                res2 = idx.fill_batch(b0.gaps[:nsa], params)
                hip2 = hashlib.sha256("\n".join(sorted(f["seq"] for r in res2 for f in r["filled"])).encode()).hexdigest()
                cpu["same_algorithm_value"] = sa["sites"] / sa["seconds"]
                cpu["same_algorithm"] = {"unit": "breakpoints/s", "processes": sa["processes"], "sites": sa["sites"], "distinct_sites": sa["distinct_sites"], "seconds": sa["seconds"],
                                         "identical_to_hip": sa["sha256"] == hip2,
                                         "what": "the device code compiled for the host (tests/emu, cross-checks compiled out), one process per core, each with its own index over the first %d donor "
                                                 "sequences and a slice of their sites (repeated to >= 3000 per process); traversal + copy + post + emit of mtg_fill_batch, as on the device" % nsa}
            except Exception as e:
                cpu["same_algorithm"] = {"error": repr(e)[:300]}

    if cpu and isinstance(secondary.get("end_to_end"), dict) and "from_container" in secondary["end_to_end"]:
        e2e = secondary["end_to_end"]
        nsites = batch_sites
        # the CPU port over the same span (index + fill), measured on its bounded sample; the whole of config 4 was NOT run on the CPU: the scaled figure is an
        # extrapolation (fill time by sites, index time by k-mers) and says so
        est = cpu["seconds"] * nsites / max(ns, 1) + cpu["index_seconds"] * float(info["nb_solid_kmers"]) / max(cpu["index_kmers"], 1)
        e2e["cpu_port_same_span"] = {"measured": {"sites": ns, "index_kmers": cpu["index_kmers"], "index_s": cpu["index_seconds"], "fill_s": cpu["seconds"], "cores": cpu["cores"]},
                                     "scaled_to_config4_estimate_s": est, "scaling": "fill_s x 100000 / sites + index_s x 3.0e9 / index_kmers (k-mer counting of the oracle is a hash-map insert per k-mer; no file I/O on either side of the index step)",
                                     "ratio_estimate_from_container": est / max(e2e["from_container"]["seconds"], 1e-9), "ratio_estimate_from_donor_in_hbm": est / max(e2e["from_donor_in_hbm"]["seconds"], 1e-9)}

    # ---------------------------------------------------------------- roofline.  What binds the job is the link to the host: every step copies its results
    # (records + ASCII sequences) into page-locked host memory.  Under it, per kernel of a step, the bytes the implemented layout has to move against HBM:
    # counted by the kernels themselves for the launches of the timed region (a LEAN gap -- target located in the unitig store -- has no copy
    # command executed and no contig scanned: copy_words_executed / post_scanned_words), divided by the kernel's OWN time (one batch on the device).
    Ln = max(acc["n_launches"], 1)
    gaps_l = acc["gaps"] / Ln
    bucket = int(info["adj_bucket_bytes"])
    nl1 = max(st_alone["n_launches"], 1) if st_alone else 0

    def alone_ms(*keys_plus_minus):
        """average time of a kernel (group) with one batch on the device: sum of the +keys minus the -keys"""
        if not st_alone:
            return None
        return sum((-1.0 if kk.startswith("-") else 1.0) * st_alone[kk.lstrip("-")] for kk in keys_plus_minus) / nl1

    def kern(name, ms_in_flight, ms_alone, bytes_per_launch, parts):
        own = ms_alone if ms_alone else ms_in_flight  # N > 1 / host-string runs have no pass with one batch alone
        ach = bytes_per_launch / (own * 1e-3) / 1e9 if own and own > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "avg_kernel_ms": own, "avg_kernel_ms_with_%d_batches_in_flight" % a.in_flight: ms_in_flight, "bytes_per_launch": bytes_per_launch,
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "bytes_breakdown": parts}

    lane_nt = max(acc["run_nt"] - 32 * acc["copy_words"], 0)
    n_lean_l = acc["n_lean_gaps"] / Ln
    sa_parts = {"adj_bucket_reads_x_%dB" % bucket: acc["index_lines"] / Ln * bucket, "unitig_store_word_reads_x_8B": acc["store_runs"] / Ln * 8, "unitig_sequence_2bit_taken_by_lanes": lane_nt / Ln / 4,
                "contig_words_written_by_lanes_x_8B": (acc["contig_words"] - acc["copy_words"]) / Ln * 8, "copy_commands_x_24B": acc["copy_cmds"] / Ln * 24,
                "per_gap_input_and_record": gaps_l * (8 + 8 + 4 + 4 + 16 + 36), "parked_gap_state_x_2x176B": acc["n_parked_gaps"] / Ln * 352}
    # the lean decision (by the walking lane; the kernel k_lean until late round 5): the gap's record, its target (k-mer, mask, offsets), the ADJ bucket of the target's junction, contig length / start, its commands, the 16-byte LeanRec it leaves
    ln_parts = {"per_gap_record_target_and_leanrec": gaps_l * (36 + 25 + 8 + 16), "adj_bucket_of_the_target_x_%dB" % bucket: gaps_l * bucket, "copy_commands_read_x_24B": acc["copy_cmds"] / Ln * 24}
    # k_copy, one wave per LISTED gap (the gaps that are not lean): only their commands are executed
    listed = max(gaps_l - n_lean_l, 0.0)
    cp_parts = {"unitig_words_read_x_8B": (acc["copy_words_executed"] + acc["copy_cmds_executed"]) / Ln * 8, "contig_words_written_x_8B": acc["copy_words_executed"] / Ln * 8,
                "copy_commands_x_24B": acc["copy_cmds_executed"] / Ln * 24, "per_listed_gap_record": listed * (36 + 4), "list_count_read_by_every_wave_of_the_grid": gaps_l / 4 * 4}
    po_parts = {"contig_words_scanned_x_8B": acc["post_scanned_words"] / Ln * 8, "bucket_reads_x_32B": acc["post_lines"] / Ln * 32,
                "coverage_abundance_bytes": acc["coverage_kmers"] / Ln, "coverage_kmer_check_of_looked_up_blocks": (acc["coverage_kmers"] - acc["coverage_direct_kmers"]) / Ln * 0.25 * (1 + k / 64.0),
                "per_gap_gapout_target_leanrec_command_read": gaps_l * (36 + 26 + 16 + 24), "per_gap_slot_record_written_160B_and_rewritten_by_the_scan_48B": gaps_l * (160 + 160 + 48)}
    em_parts = {"ascii_written": acc["seq_bytes"] / Ln, "sequence_2bit_read": acc["seq_bytes"] / Ln / 4, "per_gap_records": n_lean_l * (160 + 56 + 40) + listed * (160 + 160 + 56 + 40), "dense_contigs_x_16B": acc["dense_words"] / Ln * 16}
    sa_parts.update({"lean_decision_" + pk: pv for pk, pv in ln_parts.items()})  # the walking lane decides the lean form at the end of its walk (k_lean until late round 5)
    kerns = [kern("k_stage_a(+k_finish)", acc["kernel_ms"] / Ln, alone_ms("kernel_ms"), sum(sa_parts.values()), sa_parts),
             kern("k_copy", (acc["copy_kernel_ms"] - acc["lean_kernel_ms"]) / Ln, alone_ms("copy_kernel_ms", "-lean_kernel_ms"), sum(cp_parts.values()), cp_parts),
             kern("k_post(+k_post_lean,k_scan1,k_scan2)", acc["post_kernel_ms"] / Ln, alone_ms("post_kernel_ms"), sum(po_parts.values()), po_parts),
             kern("k_emit(+k_emit_lean)", acc["emit_kernel_ms"] / Ln, alone_ms("emit_kernel_ms"), sum(em_parts.values()), em_parts)]
    dom = max(kerns, key=lambda x: x["avg_kernel_ms"] or 0.0)
    result_bytes = acc["seq_bytes"] / Ln + gaps_l * (56 + 40)
    ms_per_batch_of_rank = elapsed / a.steps * 1e3 / max(1, (len(batches) if not cfg0["rotate"] else 1))  # this rank finishes a batch every so many ms
    pcie_ach = result_bytes / (ms_per_batch_of_rank * 1e-3) / 1e9 if ms_per_batch_of_rank > 0 else 0.0
    roof = {"bound": "pcie", "what": "results of a step (C-ABI records + ASCII sequences) copied device -> page-locked host memory inside the step; the kernels of a batch take less device time than its copies take on the link, batches in flight overlap the two",
            "achieved": pcie_ach, "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": pcie_ach / PCIE_PEAK_GBS, "bytes_per_launch": result_bytes,
            "measured_link_ceiling_GBps": "scripts/pcie_d2h.py: 57 with two or three copies in flight",
            "traffic": None, "traffic_source": None,
            "dominant_kernel": dom["kernel"], "dominant_kernel_frac_of_hbm_peak": dom["frac"], "kernels": kerns, "launches": int(acc["n_launches"]),
            "first_walk_kernel": {"k_walk (light: simple paths only)": int(acc["n_light_walks"]), "k_stage_a (full)": int(acc["n_launches"] - acc["n_light_walks"]),
                                  "gaps_that_met_a_branching_node_per_launch": acc["n_branching_gaps"] / Ln},
            "kernel_times": "avg_kernel_ms = HIP events on the batch's stream with ONE batch on the device (%s launches after the timed blocks); the same command under rocprofv3 --kernel-trace --stats with --in-flight 1 is profiles/r06_kernel_stats_one_batch_in_flight.csv"
                            % (st_alone["runs"] if st_alone else 0)}
    pmc = next((q for q in (os.path.join(ROOT, "profiles", "r06_pmc.json"), os.path.join(ROOT, "profiles", "r05_pmc.json"), os.path.join(ROOT, "profiles", "r04_pmc.json"), os.path.join(ROOT, "profiles", "r03_pmc.json")) if os.path.exists(q)), None)
    if a.workload == "human" and batch_sites == 100000 and pmc:
        pj = json.load(open(pmc))
        fill_kernels = ("k_stage_a", "k_walk", "k_finish", "k_bubble", "k_lean", "k_copy", "k_post", "k_scan1", "k_scan2", "k_emit", "k_wire_sum", "k_marshal", "k_encode_targets")
        by_kernel = {kn.split("::")[-1]: kv.get("hbm_bytes_per_launch") for kn, kv in pj.get("kernels", {}).items() if kn.split("::")[-1].startswith(fill_kernels)}
        # a launch starts with k_walk OR k_stage_a (the light or the full walk kernel): their two averages are one entry, weighted by their launches
        n_by = {kn.split("::")[-1]: kv.get("launches_FETCH_SIZE", 0) for kn, kv in pj.get("kernels", {}).items()}
        if by_kernel.get("k_walk") and by_kernel.get("k_stage_a"):
            nw_, ns_ = n_by.get("k_walk", 0), n_by.get("k_stage_a", 0)
            by_kernel["k_stage_a"] = (by_kernel["k_walk"] * nw_ + by_kernel["k_stage_a"] * ns_) / max(nw_ + ns_, 1)
            by_kernel["k_walk_or_k_stage_a_launches"] = {"k_walk": nw_, "k_stage_a": ns_, "k_walk_bytes_per_launch": by_kernel.pop("k_walk")}
        elif by_kernel.get("k_walk"):
            by_kernel["k_stage_a"] = by_kernel.pop("k_walk")
        # `traffic`: HBM bytes per launch of the dominant kernel (with the scans it is reported with); every kernel of a fill under traffic_by_kernel
        dk = dom["kernel"].split("(")[0]
        roof["traffic"] = sum(v for kn, v in by_kernel.items() if v and not isinstance(v, dict) and (kn == dk or (dk == "k_post" and kn in ("k_post_lean", "k_scan1", "k_scan2")) or (dk == "k_emit" and kn == "k_emit_lean") or (dk == "k_stage_a" and kn.startswith(("k_finish", "k_bubble", "k_walk"))))) or None
        roof["traffic_by_kernel"] = by_kernel
        roof["traffic_is"] = "REPLAYED, not measured in this run: counters cannot be read from inside the process"
        roof["traffic_source"] = ("%s (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, no trace domain, the bench command of scripts/profile_round5.sh, HEAD %s): "
                                  "average HBM bytes per launch.  FETCH_SIZE correction: none -- the guide's x2 applies to wide coalesced 16 B/lane streams; these kernels read "
                                  "scattered buckets and 8 B/lane runs, and FETCH_SIZE was calibrated at 1.000 on this library's scattered 16-byte reads (profiles/r01_pmc_fetch_size.json)"
                                  % (os.path.relpath(pmc, ROOT), pj.get("head", "?")))
        for kr in kerns:
            tk = kr["kernel"].split("(")[0]
            tv = sum(v for kn, v in by_kernel.items() if v and not isinstance(v, dict) and (kn == tk or (tk == "k_post" and kn in ("k_post_lean", "k_scan1", "k_scan2")) or (tk == "k_emit" and kn == "k_emit_lean") or (tk == "k_stage_a" and kn.startswith(("k_finish", "k_bubble", "k_walk")))))
            kr["traffic"] = tv or None
            kr["traffic_over_bytes"] = (tv / kr["bytes_per_launch"]) if tv and kr["bytes_per_launch"] else None
            # the fabric serves (and FETCH_SIZE counts) a scattered read as a 64-byte request: the same layout with every bucket read at 64 B
            rb = kr["bytes_per_launch"] + sum(v for pk, v in kr["bytes_breakdown"].items() if "bucket" in pk and pk.endswith("x_32B"))
            kr["bytes_per_launch_at_64B_requests"] = rb
            kr["traffic_over_request_bytes"] = (tv / rb) if tv and rb else None
    # what the reference's algorithm would have moved for the same contigs (64 B per membership probe, probes counted by the oracle):
    # kept for comparison only -- the unitig layout reads one bucket and one stretch of 2-bit sequence where gatb probes 8 Bloom blocks per nucleotide.
    # That work has not vanished: it is done ONCE, when the index is built (index_build below carries its clock and its roofline).
    roof["reference_algorithm_equivalent"] = {"bytes_per_launch": 64.0 * probes_per_nt * acc["contig_nt"] / Ln, "probes_per_contig_nt": probes_per_nt,
                                              "equivalent_GBps_of_k_stage_a": 64.0 * probes_per_nt * acc["contig_nt"] / Ln / max((alone_ms("kernel_ms") or acc["kernel_ms"] / Ln) * 1e-3, 1e-12) / 1e9,
                                              "note": "a step does not move these bytes (frac would exceed 1): the walk reads the unitig store the index build derived from the k-mer set; see index_build"}
    if st_alone:
        roof["one_batch_alone_ms"] = {"k_stage_a+k_finish": st_alone["kernel_ms"] / nl1, "k_finish": st_alone["finish_kernel_ms"] / nl1, "event_gap_before_k_copy": st_alone["lean_kernel_ms"] / nl1,
                                      "k_copy": (st_alone["copy_kernel_ms"] - st_alone["lean_kernel_ms"]) / nl1,
                                      "k_post+scans": st_alone["post_kernel_ms"] / nl1, "k_emit": st_alone["emit_kernel_ms"] / nl1, "parked_gaps": st_alone["n_parked_gaps"] / nl1,
                                      "sum": (st_alone["kernel_ms"] + st_alone["copy_kernel_ms"] + st_alone["post_kernel_ms"] + st_alone["emit_kernel_ms"]) / nl1,
                                      "first_kernel_to_last": st_alone["device_span_ms"] / nl1,  # less than the sum: k_finish runs next to k_lean, k_copy and k_post_lean (second stream)
                                      "launches": nl1}
    if not a.no_ceiling:
        tb = min(int(info["device_bytes"] // 2), 16 << 30)  # the ceiling is flat beyond ~16 GB (profiles/r01_random_line_ceiling.txt)
        ms, gbps = mtg.random_line_ceiling(max(tb, 1 << 26), batch_sites, 512, bucket)
        roof["random_read_ceiling_reads_per_s"] = gbps * 1e9 / bucket
        # the traversal against the HBM-random-read roofline (north star): its dependent random reads (ADJ buckets + short reads of the unitig store)
        # per second of kernel time, with six batches in flight and for one batch alone, over the measured ceiling of dependent random reads
        rr = (acc["index_lines"] + acc["store_runs"]) / Ln
        avg_s = acc["kernel_ms"] / Ln * 1e-3
        roof["random_reads_of_k_stage_a"] = {"per_launch": rr, "reads_per_s": rr / avg_s if avg_s > 0 else 0.0, "frac_of_ceiling": rr / avg_s / roof["random_read_ceiling_reads_per_s"] if avg_s > 0 else 0.0}
        if st_alone and st_alone["kernel_ms"] > 0:
            rr1 = (st_alone["index_lines"] + st_alone["store_runs"]) / nl1
            t1 = st_alone["kernel_ms"] / nl1 * 1e-3
            roof["random_reads_of_k_stage_a"]["alone"] = {"per_launch": rr1, "reads_per_s": rr1 / t1, "frac_of_ceiling": rr1 / t1 / roof["random_read_ceiling_reads_per_s"]}

    # ---------------------------------------------------------------- the index construction under a clock and a roofline (Graph::create, src/Filler.cpp:172-226): per
    # phase the device time (HIP events inside the library) and the bytes the implemented layout must move; against it the reference
    # algorithm's probes for the same k-mer set (8 membership probes of 64 B per k-mer -- what the walk no longer does per step)
    dev_s = sum(ph["ms"] for ph in bprof["phases"] if not ph["name"].startswith("hipMalloc")) / 1e3  # device time of the phases (HIP events)
    alloc_s = sum(ph["ms"] for ph in bprof["phases"] if ph["name"].startswith("hipMalloc")) / 1e3   # host wall time inside hipMalloc / hipFree
    lay_b = sum(ph["bytes"] for ph in bprof["phases"])
    ref_b = float(info["nb_solid_kmers"]) * 8 * 64
    index_build = {"seconds": t_index, "device_seconds": dev_s, "hipmalloc_seconds": alloc_s, "library_seconds": bprof["total_ms"] / 1e3,
                   "hipmalloc_note": "57.8 GB are allocated in all (the junction table's memory becomes the sparse ADJ table); on this pool the first large hipMalloc of a process takes between 1 ms and several seconds (device memory freed shortly before, by this or a previous process, is scrubbed before it is handed out again): seconds = device_seconds + hipmalloc_seconds + a few ms", "peak_device_bytes": bprof["peak_device_bytes"], "resident_bytes": int(info["device_bytes"]),
                   "source": "2-bit packed donor sequences resident in HBM (mtg_index_create_from_packed_device); abundances a function of the k-mer",
                   "phases": [{"name": ph["name"], "ms": ph["ms"], "bytes": ph["bytes"], "units": ph["units"], "GBps": ph["bytes"] / max(ph["ms"], 1e-9) / 1e6,
                               "frac_of_hbm_peak": ph["bytes"] / max(ph["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS} for ph in bprof["phases"]],
                   "bytes_of_the_layout": lay_b, "achieved_GBps": lay_b / max(dev_s, 1e-9) / 1e9, "frac_of_hbm_peak": lay_b / max(dev_s, 1e-9) / 1e9 / HBM_PEAK_GBS,
                   "reference_algorithm_equivalent": {"bytes": ref_b, "what": "k-mers x 8 membership probes x 64 B (SURVEY 8d)", "GBps_over_the_whole_build": ref_b / max(t_index, 1e-9) / 1e9,
                                                      "frac_of_hbm_peak": ref_b / max(t_index, 1e-9) / 1e9 / HBM_PEAK_GBS},
                   # SURVEY 8(d)'s bytes over the DEVICE time of the construction (the kernels' own clock; `seconds` also holds hipMalloc, a box property)
                   "sec8d_frac_over_device_seconds": ref_b / max(dev_s, 1e-9) / 1e9 / HBM_PEAK_GBS,
                   "traffic": None}
    pmcb = next((q for q in (os.path.join(ROOT, "profiles", "r06_pmc_build.json"), os.path.join(ROOT, "profiles", "r05_pmc_build.json"), os.path.join(ROOT, "profiles", "r04_pmc_build.json")) if os.path.exists(q)), os.path.join(ROOT, "profiles", "r05_pmc_build.json"))
    if a.workload == "human" and os.path.exists(pmcb):
        pb = json.load(open(pmcb))
        index_build["traffic"] = {kn.split("::")[-1]: kv.get("hbm_bytes_per_launch") for kn, kv in pb.get("kernels", {}).items()}
        index_build["traffic_is"] = "REPLAYED from %s (rocprofv3 --pmc passes of scripts/r4_build.py, HEAD %s); FETCH_SIZE x2 applied to no kernel (16 B/lane table scans, scattered buckets)" % (os.path.relpath(pmcb, ROOT), pb.get("head", "?"))

    out = {"metric": "breakpoints filled/sec", "value": value, "unit": "breakpoints/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": scaling if world > 1 else "none", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": desc, "sites_per_step": step_sites, "sites_per_batch": batch_sites, "sites_per_rank_and_step": sites_per_rank_step, "distinct_batches_per_rank": len(batches),
                      "site_set": cfg0["total"], "donor_sequences": S.nseq, "erroneous_fragments": int(len(S.extra_lens)), "k": k, "max_nodes": 100, "max_length": 10000,
                      "index": "exact k-mer set of the donor, abundance of a k-mer = Poisson(24) drawn from its hash, at least 3 (SURVEY 8d; no reads simulated)", "nb_solid_kmers": int(info["nb_solid_kmers"]),
                      "nb_unitigs": int(info["nb_unitigs"]), "index_bytes": int(info["device_bytes"]), "unitig_store_bytes": int(info["unitig_bytes"]), "index_build_s": t_index,
                      "genome_gen_s": t_gen, "donor_upload_s": t_donor_up, "batches_in_flight": a.in_flight, "input": "host strings, marshalled in every step" if a.host_strings else "prepared batches, resident in HBM",
                      "output": "C-ABI records + ASCII sequences in page-locked host memory" + ("; every batch also gathered on rank 0 in relocatable form (records + sequences), validated there" if dist_on else "")},
           "timed_blocks": {"blocks": len(times), "steps_per_block": a.steps, "reported": "median", "ms_per_step_min": min(times) / a.steps * 1e3,
                            "ms_per_step_median": elapsed / a.steps * 1e3, "ms_per_step_max": max(times) / a.steps * 1e3, "timed_seconds_total": sum(times)},
           "filled": R0["n_filled"], "sites_verified": R0["n_sites"] if R0["identical"] is not None else None, "sites_run_in_the_untimed_pass": R0["n_sites"],
           "filled_per_s": value * R0["n_filled"] / max(R0["n_sites"], 1),
           "filled_sequences_identical_to_truth": R0["identical"], "gathered_payload_verified": R0["gathered"],
           "stage_ms_per_batch": {"k_stage_a+k_finish": acc["kernel_ms"] / Ln, "k_finish": acc["finish_kernel_ms"] / Ln, "parked_gaps": acc["n_parked_gaps"] / Ln, "lean_gaps": acc["n_lean_gaps"] / Ln, "event_gap_before_k_copy": acc["lean_kernel_ms"] / Ln, "k_copy": (acc["copy_kernel_ms"] - acc["lean_kernel_ms"]) / Ln,
                                  "k_post+scans": acc["post_kernel_ms"] / Ln, "k_emit": acc["emit_kernel_ms"] / Ln, "d2h": acc["d2h_ms"] / Ln,
                                  "host": acc["host_ms"] / Ln, "c_call": acc["total_ms"] / Ln, "first_kernel_to_last": acc["device_span_ms"] / Ln,
                                  "note": "kernel columns are 0 unless KERNEL_TIMERS is set: the timed blocks record three events per batch; roofline.one_batch_alone_ms has every kernel's own time"},
           "roofline": roof, "cpu_baseline": cpu, "index_build": index_build}
    out.update(secondary)
    # ---------------------------------------------------------------- secondary lines: the workloads whose walks cross bubbles (SNPs; indels; tips and error bubbles), each
    # as a child process once this one has given the device back (two human-scale indexes do not fit the HBM together)
    if single and not a.no_secondary and not a.no_children and a.workload == "human" and not a.host_strings:
        for b in batches:
            if hasattr(b.prepared, "close"):
                b.prepared.close()
        batches.clear()
        idx.close()
        torch.cuda.empty_cache()
        for wl, key in (("human-het", "secondary_diploid"), ("human-indel", "secondary_indel"), ("human-tips", "secondary_tips")):
            try:
                # every child checks a sample of its sites against the oracle (cpu_baseline leg: 6 000 sites, index over the donor sequences of those loci): the
                # alleles and error bubbles of these sets have no closed-form truth, the oracle's fills are the reference
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wl, "--batches", "3", "--cpu-sites", "6000", "--cpu-index-seqs", "12000", "--cpu-same-sites", "0",
                                     "--no-ceiling", "--no-secondary", "--full-line", "--detail", "", "--steps", str(a.steps), "--warmup", str(a.warmup), "--in-flight", str(a.in_flight), "--repeats", "5"],
                                    capture_output=True, text=True, timeout=600)
                d = json.loads(cp.stdout.strip().splitlines()[-1])
                cb = d.get("cpu_baseline") or {}
                out[key] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "timed_blocks": d["timed_blocks"],
                            "filled": d["filled"], "sites_run_in_the_untimed_pass": d["sites_run_in_the_untimed_pass"],
                            "identical_to_oracle": cb.get("identical_to_hip"), "oracle_sample": cb.get("sample"), "oracle_sites_per_s": cb.get("value"),
                            "one_batch_alone_ms": d["roofline"].get("one_batch_alone_ms"),
                            "stage_ms_per_batch": d["stage_ms_per_batch"], "ratio_to_headline": d["value"] / value if value else None}
            except Exception as e:  # the headline line does not depend on it
                out[key] = {"error": repr(e)[:300]}
        # a fourth secondary, of another kind: an index COUNTED FROM READS (200 Mbp donor, 30x reads with 0.5 % substitutions through -in): the graph has the
        # tips and bubbles sequencing errors leave, unitigs end where they do in data; rate, unitig statistics, and a sample against the oracle
        if not os.environ.get("MTG_BENCH_NO_READS"):
            try:
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r4_reads_workload.py"), "--steps", str(a.steps)], capture_output=True, text=True, timeout=600)
                d = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
                d["ratio_to_headline"] = d["value"] / value if value else None
                out["secondary_reads_built"] = d
            except Exception as e:
                out["secondary_reads_built"] = {"error": repr(e)[:300]}
        # BASELINE configs[1] (SURVEY 8d cfg 2): 5 Mbp donor as 1000 x 5 kb, 30x reads of 150 nt through -in with -abundance-min 3, 1000 sites -- E0 (error-free
        # reads: pure simple paths, parity fully pinned) and E1 (0.1 % substitutions: a few tips survive), each with 300 sites against the oracle
        if not os.environ.get("MTG_BENCH_NO_ECOLI"):
            for err, key in ((0.0, "secondary_ecoli_E0"), (0.001, "secondary_ecoli_E1")):
                try:
                    cp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r4_reads_workload.py"), "--nseq", "1000", "--err", str(err), "--oracle-seqs", "300", "--steps", str(max(a.steps, 50)),
                                         "--label", "BASELINE configs[1] " + key[-2:]], capture_output=True, text=True, timeout=300)
                    d = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
                    d["ratio_to_headline"] = d["value"] / value if value else None
                    out[key] = d
                except Exception as e:
                    out[key] = {"error": repr(e)[:300]}
        # contig mode (BASELINE configs[2]), the tool on a resident index: the reference's bundled case (GFA == gold.gfa), 2 000 and
        # 10 000 contigs cut from a synthetic donor with the all-pairs dictionary (2 (N - 1) targets per seed), a sample of the seeds against the CPU oracle
        if not os.environ.get("MTG_BENCH_NO_CONTIG"):
            for key, args, tmo in (("secondary_contig_bundled", ["--bundled"], 120), ("secondary_contig_2k", ["--contigs", "2000", "--oracle-stride", "40", "--repeats", "1"], 180),
                                   ("secondary_contig_10k", ["--contigs", "10000", "--oracle-stride", "400", "--repeats", "1"], 400)):
                try:
                    cp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r6_contig_workload.py")] + args, capture_output=True, text=True, timeout=tmo)
                    out[key] = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
                except Exception as e:
                    out[key] = {"error": repr(e)[:300]}
    if dist_on:
        for r in results:
            if r["pg"] is not None:
                r["pg"].drain()
        dist.destroy_process_group()
    # the JSON line is the last thing on stdout: whatever native libraries (RCCL's version banner) left in the C stdio buffer goes first
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if a.full_line:
        print(json.dumps(out), flush=True)
        return
    detail_path = a.detail or None
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            side = os.path.join(ROOT, "gpurun_out")  # scratch that travels back from the GPU box
            if os.path.isdir(side):
                shutil.copyfile(detail_path, os.path.join(side, os.path.basename(detail_path)))
        except OSError as e:
            print("bench.py: detail file not written: %r" % (e,), file=sys.stderr)
            detail_path = None
    print(json.dumps(out), file=sys.stderr, flush=True)  # the whole detail object: stderr, never the last stdout line
    print(json.dumps(compact_line(out, detail_path)), flush=True)


if __name__ == "__main__":
    main()
