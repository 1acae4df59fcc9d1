#!/bin/bash
# one stream per batch (default), the general k_post on a second stream, the finishing kernel there as well: value, sequences-left-in-HBM rate
# (the kernel-bound one) and one batch alone, per workload.   gpurun -- 'bash scripts/r4_streams.sh <workload> ...'
out=gpurun_out/r4streams; mkdir -p $out
for W in "$@"; do for v in MTG_TUNING= MTG_POST_SECOND_STREAM=1 MTG_FINISH_OVERLAP=1; do
    env MTG_BENCH_NO_READS=1 MTG_BENCH_NO_E2E=1 $v python bench.py --workload $W --cpu-sites 0 --cpu-same-sites 0 --no-children --no-tool --no-ceiling > $out/b.json 2> $out/b.err
    python - $W "$v" $out/b.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
a = d["roofline"]["one_batch_alone_ms"]
print("%s %-22s value %.1f M/s, left in HBM %.1f; alone: sum %.3f first-to-last %.3f k_finish %.3f" % (sys.argv[1], sys.argv[2] or "default", d["value"] / 1e6, d.get("value_sequences_left_in_hbm", 0) / 1e6, a["sum"], a.get("first_kernel_to_last", 0), a["k_finish"]))
PY
done; done 2>&1 | tee $out/summary.txt
