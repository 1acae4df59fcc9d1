#!/bin/bash
# The secondary workloads one by one (bench.py --workload W: value, the sample's identity with the oracle, parked gaps, kernels alone) and
# the reads-built one; writes gpurun_out/r4sec/summary.txt.  Run on the GPU box: gpurun --timeout 1500 -- 'bash scripts/r4_secondary.sh'
out=gpurun_out/r4sec
mkdir -p $out
: > $out/summary.txt
for w in ${WORKLOADS:-human-tips human-het human-indel}; do
    MTG_BENCH_NO_READS=1 MTG_BENCH_NO_E2E=1 python bench.py --workload $w --cpu-sites 6000 --cpu-index-seqs 12000 --cpu-same-sites 0 --no-children --no-tool --no-ceiling > $out/$w.json 2> $out/$w.err
    python - $w $out/$w.json >> $out/summary.txt <<'PY'
import json, sys
w, p = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    k = {x["kernel"]: round(x["avg_kernel_ms"] or 0.0, 3) for x in d["roofline"].get("kernels", [])}
    print(w, round(d["value"] / 1e6, 1), "M/s; identical to the oracle sample:", d["cpu_baseline"].get("identical_to_hip"), "; kernels alone (ms):", k)
except Exception as e:
    print(w, "FAILED", e)
PY
done
python scripts/r4_reads_workload.py > $out/reads.json 2> $out/reads.err
tail -1 $out/reads.json >> $out/summary.txt
cat $out/summary.txt
