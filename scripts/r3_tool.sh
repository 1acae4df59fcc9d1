#!/bin/bash
# round 3: the tool's rate on the resident human-scale index (bench.py's tool leg alone), for several numbers of workers per device
out=gpurun_out/${1:-r3_tool}; mkdir -p $out
for f in ${FLIGHTS:-3 6}; do
  MTG_CLI_IN_FLIGHT=$f MTG_TOOL_TIMERS=1 timeout 900 python bench.py --no-children --no-ceiling --cpu-sites 0 --steps 10 --warmup 3 > $out/bench_f$f.json 2> $out/bench_f$f.err
  grep "\[tool\]" $out/bench_f$f.err | tail -n 1
  python - $out/bench_f$f.json $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d.get("tool",{})
print("workers per device", sys.argv[2], "| tool %.2f M/s" % (d.get("tool_sites_per_s",0)/1e6), "seconds", t.get("seconds"), "out GB/s", t.get("output_GBps"), "identical", t.get("sequences_identical_to_truth"))
PY
done
