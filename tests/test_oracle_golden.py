"""Pins the CPU oracle against every golden vector the reference's own tests hold for the fill path
(SURVEY.md 8c): test/full_test (FASTA + VCF, headers included), test/contig_test (GFA full diff,
FASTA, seed dictionary) and the graph KATs of both golden logs."""
import os

import numpy as np

from tests import oracle_lib


def _read(p):
    with open(p) as f:
        return f.read()


def _vcf_body(p):
    return [l for l in _read(p).splitlines() if not l.startswith("##")]


def test_full_test_bkpt_mode(oracle, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "data")
    idx = oracle_lib.Index.from_files([os.path.join(d, "reads_r1.fastq"), os.path.join(d, "reads_r2.fastq")], k=31, abundance_min=-1)
    # KATs: test/full_test/gold_fill.output:11-14
    assert oracle.mtgo_index_auto_cutoff(idx.h) == 7
    assert idx.stats() == (7419, 36)
    st = idx.fill_files("bkpt", os.path.join(golden_dir, "full_test", "gold.breakpoints"), str(tmp_path / "full"))
    assert (st["records"], st["filled"], st["multiple"]) == (8, 8, 0)  # gold_fill.output:18-21
    assert _read(tmp_path / "full.insertions.fasta") == _read(os.path.join(golden_dir, "full_test", "gold.insertions.fasta"))
    assert _vcf_body(tmp_path / "full.insertions.vcf") == _vcf_body(os.path.join(golden_dir, "full_test", "gold.insertions.vcf"))
    idx.close()


def test_contig_test_contig_mode(oracle, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "data")
    g = os.path.join(golden_dir, "contig_test")
    idx = oracle_lib.Index.from_files([os.path.join(d, "contig-reads.fasta.gz")], k=31, abundance_min=3)
    assert idx.stats() == (10194, 46)  # test/contig_test/gold.log:12-13
    st = idx.fill_files("contig", os.path.join(d, "contigs.fasta"), str(tmp_path / "ctg"))
    # gold.log:17-25: 10 contigs, 9 used, 18 seeds, 13 filled, 4 multiple
    assert (st["nb_contigs"], st["nb_used_contigs"], st["records"], st["filled"], st["multiple"]) == (10, 9, 18, 13, 4)
    assert _read(tmp_path / "ctg.gfa") == _read(os.path.join(g, "gold.gfa"))  # full diff, test/simple_full_test.sh:188
    assert _read(tmp_path / "ctg.insertions.fasta") == _read(os.path.join(g, "gold.insertions.fasta"))
    assert _read(tmp_path / "ctg_seed_dictionary.fasta") == _read(os.path.join(g, "gold_seed_dictionary.fasta"))
    # gold.info.txt is stale for multi-node seeds; its single-node rows are pinned
    mine = dict(l.split("\t", 1) for l in _read(tmp_path / "ctg.info.txt").splitlines())
    gold = dict(l.split("\t", 1) for l in _read(os.path.join(g, "gold.info.txt")).splitlines())
    single = [n for n, v in gold.items() if v.split("\t")[1] == "1"]
    assert len(single) == 10
    for n in single:
        assert mine[n] == gold[n], n
    idx.close()


def test_needleman_wunsch_kat(oracle):
    # src/Utils.cpp:87-189: identity = matches on the traceback / max(len)
    assert oracle.mtgo_needleman_wunsch(b"ACGTACGT", b"ACGTACGT") == 1.0
    assert abs(oracle.mtgo_needleman_wunsch(b"ACGTACGT", b"ACGAACGT") - 7 / 8) < 1e-7
    assert abs(oracle.mtgo_needleman_wunsch(b"ACGTACGTAA", b"ACGTACGT") - 0.8) < 1e-7


def test_end_rule_switch_only_changes_info(oracle, golden_dir, tmp_path):
    """the alternative end-of-branching rule changes contig boundaries but not the outputs."""
    d = os.path.join(golden_dir, "data")
    idx = oracle_lib.Index.from_files([os.path.join(d, "contig-reads.fasta.gz")], k=31, abundance_min=3)
    p = oracle_lib.default_params(end_rule_nonbranching=1)
    idx.fill_files("contig", os.path.join(d, "contigs.fasta"), str(tmp_path / "alt"), params=p)
    assert _read(tmp_path / "alt.gfa") == _read(os.path.join(golden_dir, "contig_test", "gold.gfa"))
    idx.close()


def test_synthetic_abundance_is_poisson_24():
    """the abundance model of the synthetic benchmark sets (donor k-mers with Poisson(24) abundance): drawn by inversion from a
    64-bit hash of the k-mer; mean and variance of 24, the frequencies follow the distribution, never below the solidity threshold"""
    import math
    import random
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(60000))
    idx = oracle_lib.Index.from_sequences([g], 31, 3, 0)
    km, ct = idx.export()
    idx.close()
    ct = np.asarray(ct, dtype=np.float64)
    assert len(ct) > 59000 and ct.min() >= 3
    assert abs(ct.mean() - 24.0) < 0.1 and abs(ct.var() - 24.0) < 0.6
    for v in (15, 20, 24, 28, 35):
        p = math.exp(-24.0) * 24.0 ** v / math.factorial(v)
        n = float((ct == v).sum())
        assert abs(n - p * len(ct)) < 5 * math.sqrt(p * len(ct)), (v, n, p * len(ct))
