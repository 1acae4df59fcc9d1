#!/bin/bash
# round 3: the GPU parity suite (small tests first, the full-size ones under their own timeout) and a short bench line
out=gpurun_out/${1:-r3_check}
mkdir -p $out
timeout 700 python -u -m pytest tests/test_gpu_parity.py tests/test_micro_cases.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^tests/" | tail -n 15 > $out/pytest.txt; cat $out/pytest.txt
timeout ${2:-1200} python -u -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^tests/|human-scale index|config 4" | tail -n 15 > $out/pytest_full.txt; cat $out/pytest_full.txt
timeout 600 python bench.py --no-secondary --cpu-sites 0 --steps 20 --warmup 5 > $out/bench_human.json 2> $out/bench_human.err
python - $out/bench_human.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("human value %.1f M/s" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], "index GB %.1f" % (d["config"]["index_bytes"]/1e9), "build s %.1f" % d["config"]["index_build_s"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["stage_ms_per_batch"].items()}, "alone", d["roofline"].get("one_batch_alone_ms"), "identical", d.get("filled_sequences_identical_to_truth"), "pcie frac %.2f" % d["roofline"]["frac"])
except Exception as e:
    print("bench FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
