#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2h}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
python3 scripts/diag_batches.py 4 2>&1 | grep -v amdgpu.ids | grep "^batch" | cut -c1-150
HET=1 python3 scripts/diag_batches.py 2 2>&1 | grep -v amdgpu.ids | grep "^batch" | cut -c1-150
for w in human human-het; do
  timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --workload $w > $O/b_$w.json 2> $O/b_$w.err
  python3 - $O/b_$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "value %.4g  ms/step %.3f"%(d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["stage_ms_per_batch"].items()}, d["filled_sequences_identical_to_truth"], d["roofline"].get("one_batch_alone_ms"))
PY
done
