"""BASELINE config 2 at full size: 5 Mbp donor as 1000 x 5 kb sequences, 1000 insertion sites, 30x 150-bp simulated reads,
variants E0 (error-free) and E1 (0.1 % substitutions), index built from the reads with -abundance-min 3 (the -in path).
Runs `MindTheGap fill` through the HIP library and the CPU oracle, compares the output files byte for byte."""
import hashlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import torch
torch.cuda.init()
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet, simulate_reads
from tests import oracle_lib

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
res = {}
with tempfile.TemporaryDirectory() as d:
    S = SynthSet(nseq=nseq, n_sites=nseq, seed=1)
    bk = os.path.join(d, "s.breakpoints")
    S.write_breakpoints(bk)
    for name, err in (("E0", 0.0), ("E1", 0.001)):
        reads = os.path.join(d, name + ".fa")
        t0 = time.time(); n = simulate_reads(S, reads, 30, 150, err, seed=5); t_sim = time.time() - t0
        t0 = time.time(); rc = mtg.fill_main(["-in", reads, "-bkpt", bk, "-abundance-min", "3", "-out", os.path.join(d, "hip_" + name)]); t_hip = time.time() - t0
        assert rc == 0
        t0 = time.time(); o = oracle_lib.Index.from_files([reads], 31, 3); t_oidx = time.time() - t0
        st = o.fill_files("bkpt", bk, os.path.join(d, "cpu_" + name), params=oracle_lib.default_params(nb_cores=1))
        same = {}
        for ext in (".insertions.fasta", ".info.txt"):
            a = open(os.path.join(d, "hip_" + name + ext), "rb").read(); b = open(os.path.join(d, "cpu_" + name + ext), "rb").read()
            same[ext] = a == b
        va = [l for l in open(os.path.join(d, "hip_" + name + ".insertions.vcf")) if not l.startswith("##")]
        vb = [l for l in open(os.path.join(d, "cpu_" + name + ".insertions.vcf")) if not l.startswith("##")]
        same[".vcf"] = va == vb
        info = [l.split("\t") for l in open(os.path.join(d, "cpu_" + name + ".info.txt"))]
        multi = sum(1 for r in info if int(r[2]) > 1)
        res[name] = {"reads": n, "solid_kmers": len(o), "filled_cpu": st["filled"], "sites": st["records"], "multi_contig_sites": multi, "identical": same,
                     "sha256_fasta": hashlib.sha256(open(os.path.join(d, "hip_" + name + ".insertions.fasta"), "rb").read()).hexdigest(),
                     "oracle_fill_s_1core": st["seconds"], "oracle_probes": st["probes"], "hip_cli_total_s_incl_index": t_hip, "oracle_index_s": t_oidx, "simulate_s": t_sim}
        o.close()
print(json.dumps(res, indent=1))
