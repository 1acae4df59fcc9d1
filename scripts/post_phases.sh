#!/bin/bash
# k_post with phases knocked out (MTG_POST_DBG: 1 no coverage pass, 2 no terminal search, 4 no dense copy): where its time goes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-postph}; mkdir -p $O
for dbg in 0 1 4 5 2; do
  MTG_POST_DBG=$dbg timeout 300 python bench.py --cpu-sites 0 --no-ceiling --in-flight 1 --steps 10 --warmup 2 > $O/dbg$dbg.json 2> $O/dbg$dbg.err
  python - $O/dbg$dbg.json $dbg <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("MTG_POST_DBG=%s k_stage_a %.3f k_post %.3f ms/step %.2f" % (sys.argv[2], r["avg_kernel_ms"], r["post_kernel"]["avg_kernel_ms"], d["ms_per_step"]))
PY
done
