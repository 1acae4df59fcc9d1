#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2f}; mkdir -p $O
for w in human human-het; do
 for f in 3 4 6 8; do
  timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --workload $w --in-flight $f > $O/b_${w}_$f.json 2> $O/b_${w}_$f.err
  python3 - $O/b_${w}_$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "value %.4g  ms/step %.3f"%(d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["stage_ms_per_batch"].items()}, d["filled_sequences_identical_to_truth"])
PY
 done
done
