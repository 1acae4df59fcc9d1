#!/usr/bin/env python3
"""diagnostic: per-batch statistics of the traversal on the bench's site set (which batches walk slowly, and why)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
het = 4 if os.environ.get("HET") else 0
S = SynthSet(nseq=600000, n_sites=nb * 100000, seed=1, k=31, het_snps=het, het_indels=2 if os.environ.get("INDEL") else 0)  # HET=1 INDEL=1: deletions besides the SNPs
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
print(idx.info())
params = mtg.FillParams(max_nodes=100, max_depth=10000)
for b in range(nb):
    gaps = []
    for i in range(b * 100000, (b + 1) * 100000):
        l, r, ins = S.site(i)
        gaps.append(mtg.Gap(l, r, [(r, "x", False)]))
    prep = mtg.Index.prepare_gaps(gaps)
    for rep in range(3):
        h, nf, _ = idx.fill_prepared(prep, params, want_seqs=False)
        st = mtg.last_batch_stats()
        idx.free_results(h)
    print("batch", b, "k_stage_a %.3f copy %.3f post %.3f emit %.3f host %.3f total %.3f" % (st["kernel_ms"], st["copy_kernel_ms"], st["post_kernel_ms"], st["emit_kernel_ms"], st["host_ms"], st["total_ms"]),
          {k: st[k] for k in ("index_lines", "store_runs", "run_nt", "contig_nt", "n_launches", "n_retried_gaps", "post_lines", "coverage_kmers", "dense_words", "copy_words", "copy_cmds")}, "filled", int((nf > 0).sum()))
    # the slowest walks: split the batch in 16 pieces and time each traversal alone
    if len(sys.argv) > 2:
        for piece in range(16):
            sub = mtg.Index.prepare_gaps(gaps[piece * 6250:(piece + 1) * 6250])
            h, nf, _ = idx.fill_prepared(sub, params, want_seqs=False)
            st = mtg.last_batch_stats()
            idx.free_results(h)
            print("   piece %2d k_stage_a %.3f lines %d runs %d run_nt %d contig_nt %d dense %d host %.2f" % (piece, st["kernel_ms"], st["index_lines"], st["store_runs"], st["run_nt"], st["contig_nt"], st["dense_words"], st["host_ms"]))
