#!/bin/bash
# round 3: the bubble kernels against the oracle -- the small GPU suites under every launch sequence, and the secondary workloads at full size
# with the oracle's sample (bench.py --cpu-sites: `identical_to_hip` compares the CPU port's sequences with what the library returned)
out=gpurun_out/${1:-r3_verify}; mkdir -p $out
for env in "MTG_ROUNDS=6" "MTG_ROUNDS=3 MTG_BUBBLE_GROUPS=1" "MTG_ROUNDS=0 MTG_FINISH_G=64" "MTG_FINISH_G=1" "MTG_CLASSIC_WALK=1"; do
  echo "== $env: $(env $env timeout 900 python -u -m pytest tests/test_gpu_parity.py tests/test_micro_cases.py -m gpu -x -q -p no:cacheprovider -k 'fuzz or snp or diploid or allelic or tier or micro or reference or golden or synthetic' 2>&1 | tail -n 1)"
done 2>&1 | tee $out/launch_sequences.txt
for w in human-indel human-tips human-het; do
  timeout 900 python bench.py --workload $w --batches 3 --cpu-sites 6000 --cpu-same-sites 0 --no-ceiling --no-secondary --steps 10 --warmup 3 > $out/bench_$w.json 2> $out/bench_$w.err
  python - $out/bench_$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["cpu_baseline"]
    print(sys.argv[2], "value %.1f M/s" % (d["value"]/1e6), "| oracle on", c["sample"][:60], "...: identical_to_hip", c["identical_to_hip"], "cpu %.0f/s" % c["value"], "| parked per batch", d["stage_ms_per_batch"]["parked_gaps"])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done 2>&1 | tee $out/secondary_against_oracle.txt
