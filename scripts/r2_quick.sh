#!/bin/bash
# quick look: adversarial + diploid GPU tests, then haploid / diploid bench lines with one and three batches in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2q}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for w in human human-het; do
 for f in 1 3; do
  timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --workload $w --in-flight $f --batches ${2:-4} > $O/b_${w}_$f.json 2> $O/b_${w}_$f.err
  python3 - $O/b_${w}_$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "value %.4g  ms/step %.3f"%(d["value"], d["ms_per_step"]), {k:round(v,3) for k,v in d["stage_ms_per_batch"].items()}, d["filled"], d["filled_sequences_identical_to_truth"], d["roofline"].get("one_batch_alone_ms"))
PY
 done
done
