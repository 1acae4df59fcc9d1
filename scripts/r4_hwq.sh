#!/bin/bash
# hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES, default 4) against the streams of six batches in flight: value of a workload per setting
#   gpurun -- 'bash scripts/r4_hwq.sh <workload> <settings...>'
out=gpurun_out/r4hwq; mkdir -p $out
W=$1; shift
for q in "$@"; do
    env MTG_BENCH_NO_READS=1 MTG_BENCH_NO_E2E=1 GPU_MAX_HW_QUEUES=$q python bench.py --workload $W --cpu-sites 0 --cpu-same-sites 0 --no-children --no-tool --no-ceiling > $out/$W.$q.json 2> $out/$W.$q.err
    python - $W $q $out/$W.$q.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("%s GPU_MAX_HW_QUEUES=%s: value %.1f M/s, text %.1f, strings %.1f, sequences left in HBM %.1f" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d.get("value_from_host_text", 0) / 1e6, d.get("value_from_host_strings", 0) / 1e6, d.get("value_sequences_left_in_hbm", 0) / 1e6))
PY
done 2>&1 | tee -a $out/summary.txt
