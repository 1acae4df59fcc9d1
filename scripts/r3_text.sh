#!/bin/bash
# round 3: the text-block entry (mtg_fill_text) and the tool on it: parity tests, then the bench's input-side rates with two pool threads and with all
out=gpurun_out/${1:-r3_text}; mkdir -p $out
timeout 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "text or cli or reverse or replica or edge" 2>&1 | grep -E "passed|failed|Error|assert|^tests/" | tail -n 15 | tee $out/pytest.txt
for th in 2 0; do
  export MTG_POOL_THREADS=$th; [ $th = 0 ] && unset MTG_POOL_THREADS
  MTG_TOOL_TIMERS=1 timeout 900 python bench.py --no-children --no-ceiling --cpu-sites 0 --steps 20 --warmup 5 > $out/bench_th$th.json 2> $out/bench_th$th.err
  grep "\[tool\]" $out/bench_th$th.err | tail -n 2
  python - $out/bench_th$th.json $th <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("pool threads", sys.argv[2] or "all", "| prepared %.1f M/s | host strings %.1f | host text %.1f | tool %.2f M/s" % (d["value"]/1e6, d.get("value_from_host_strings",0)/1e6, d.get("value_from_host_text",0)/1e6, d.get("tool_sites_per_s",0)/1e6), d.get("tool"))
except Exception as e:
    print("bench FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
