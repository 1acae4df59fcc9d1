#!/bin/bash
# the headline workload without its secondary measurements: value, the kernels of one batch alone, left-in-HBM rate.  gpurun -- 'bash scripts/r4_headline_quick.sh [ENV=..]'
out=gpurun_out/r4quick; mkdir -p $out
env MTG_BENCH_NO_READS=1 MTG_BENCH_NO_E2E=1 "$@" python bench.py --no-children --no-tool --no-ceiling --cpu-sites 0 --cpu-same-sites 0 > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.1f M/s; sequences left in HBM %.1f; strings %.1f; text %.1f; identical to truth: %s" % (d["value"] / 1e6, d.get("value_sequences_left_in_hbm", 0) / 1e6, d.get("value_from_host_strings", 0) / 1e6, d.get("value_from_host_text", 0) / 1e6, d.get("filled_sequences_identical_to_truth")))
print("one batch alone (ms):", {k: round(v, 4) for k, v in d["roofline"]["one_batch_alone_ms"].items()})
PY
