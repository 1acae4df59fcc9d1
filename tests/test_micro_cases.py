"""The reference's own micro-cases (test/simple_test.sh: find + fill on a few hundred to ten thousand reads, the filled sequences compared
with test/truths/*.fasta), as far as the fill side goes: the breakpoint k-mers `find` would emit are DERIVED INPUTS here (find is out of
scope): left = the k reference nucleotides before the site, right = the k after it.  The site positions come from the truth headers
("..._pos_<p>_repeat_<r>_...", insert_ref10K) or are the one position where the reference differs from the genome the reads were drawn
from (52 on deleted.fasta, 120 on deleted_before_SNP.fasta); with this rule all 13 + 2 truths come out exactly, fuzzy sites included.

What these cases pin beyond the two golden sets: the [MEM] traversal on 15 more gaps whose expected sequences are the reference's, and the
automatic solidity cut-off (gatb's Histogram heuristic): the fill of "clean-insert" only succeeds when the cut-off inferred from
master.fasta -- a sparse histogram with counts 0..3 -- is at most 33, which the restatement only yields with gatb's integer (truncated)
smoothed histogram (the golden value 7 of test/full_test does not tell the two apart)."""
import os
import re

import pytest

from tests import oracle_lib

K = 31
MICRO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "micro")


def _fa(path):
    recs = []
    for l in open(path):
        l = l.rstrip("\n")
        if l.startswith(">"):
            recs.append([l[1:], ""])
        elif recs:
            recs[-1][1] += l
    return recs


def _cases():
    """(name, read files, breakpoint file text, expected sequence lines)"""
    out = []
    ref = _fa(os.path.join(MICRO, "ref_g10K_del.fasta"))[0][1]
    truth = _fa(os.path.join(MICRO, "truth_insert_ref10K.fasta"))
    bk, want = "", []
    for name, seq in truth:
        m = re.search(r"bkpt(\d+)_left_kmer_(\w+)_pos_(\d+)_repeat_(\d+)_(\w+)", name)
        i, pos, rep, gt = m.group(1), int(m.group(3)), int(m.group(4)), m.group(5)
        hdr = "bkpt%s_%s_pos_%d_fuzzy_%d_%s" % (i, m.group(2), pos, rep, gt)
        bk += ">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (hdr, ref[pos - K:pos], hdr, ref[pos:pos + K])
        want.append(seq)
    out.append(("13-inserts-ref10k", ["readref10K.fasta"], bk, want))
    for case, reads, refn, truthn, pos in (("clean-insert", ["master.fasta"], "ref_deleted.fasta", "truth_insertion.fasta", 52),
                                           ("snp-before-clean-insert", ["master.fasta"], "ref_deleted_before_SNP.fasta", "truth_insertion_before_SNP.fasta", 120),
                                           ("hetero-insert", ["deleted.fasta", "master.fasta"], "ref_deleted.fasta", "truth_insertion.fasta", 52)):
        ref = _fa(os.path.join(MICRO, refn))[0][1]
        hdr = "bkpt1_ref_pos_%d_fuzzy_0_HOM" % pos
        bk = ">%s left_kmer\n%s\n>%s right_kmer\n%s\n" % (hdr, ref[pos - K:pos], hdr, ref[pos:pos + K])
        out.append((case, reads, bk, [_fa(os.path.join(MICRO, truthn))[0][1]]))
    return out


def _seq_lines(path):
    return [l.rstrip("\n") for l in open(path) if not l.startswith(">")]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_oracle_reproduces_the_reference_truths(case, tmp_path):
    name, reads, bk_text, want = case
    bk = str(tmp_path / "derived.breakpoints")
    open(bk, "w").write(bk_text)
    o = oracle_lib.Index.from_files([os.path.join(MICRO, r) for r in reads], K, -1)  # -abundance-min auto, like the script
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    assert _seq_lines(str(tmp_path / "cpu.insertions.fasta")) == want
    o.close()


def test_auto_cutoff_datapoints(tmp_path):
    """7 on the reads of test/full_test (gold_fill.output:11); at most 33 on master.fasta (see the module docstring): the restatement says 3"""
    g = os.path.join(os.path.dirname(MICRO), "data")
    o = oracle_lib.Index.from_files([os.path.join(g, "reads_r1.fastq"), os.path.join(g, "reads_r2.fastq")], K, -1)
    assert len(o) == 7419  # abundance_min 7
    o.close()
    o = oracle_lib.Index.from_files([os.path.join(MICRO, "master.fasta")], K, -1)
    o3 = oracle_lib.Index.from_files([os.path.join(MICRO, "master.fasta")], K, 3)
    assert len(o) == len(o3) == 167
    o.close(); o3.close()


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_product_on_emulator_reproduces_the_reference_truths(case, tmp_path):
    """the product's CLI (host code + device code under emulation): `fill -in <reads> -bkpt <derived>` with the automatic cut-off"""
    from tests import emu_lib
    from mindthegap_amd import lib as L
    saved = L._lib
    try:
        mtg = emu_lib.product_on_emulator()
        name, reads, bk_text, want = case
        bk = str(tmp_path / "derived.breakpoints")
        open(bk, "w").write(bk_text)
        assert mtg.fill_main(["-in", ",".join(os.path.join(MICRO, r) for r in reads), "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
        assert _seq_lines(str(tmp_path / "hip.insertions.fasta")) == want
    finally:
        L._lib = saved


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_hip_reproduces_the_reference_truths(case, tmp_path):
    """the same through libmtgfill.so on the MI355X (k-mer counting, index, traversal, records on the device), next to the oracle's files"""
    import torch
    torch.cuda.init()
    import mindthegap_amd as mtg
    mtg.load_library()
    name, reads, bk_text, want = case
    bk = str(tmp_path / "derived.breakpoints")
    open(bk, "w").write(bk_text)
    paths = [os.path.join(MICRO, r) for r in reads]
    assert mtg.fill_main(["-in", ",".join(paths), "-bkpt", bk, "-out", str(tmp_path / "hip")]) == 0
    assert _seq_lines(str(tmp_path / "hip.insertions.fasta")) == want
    o = oracle_lib.Index.from_files(paths, K, -1)
    o.fill_files("bkpt", bk, str(tmp_path / "cpu"))
    for ext in (".insertions.fasta", ".info.txt"):
        assert open(str(tmp_path / "hip") + ext).read() == open(str(tmp_path / "cpu") + ext).read(), ext
    o.close()
