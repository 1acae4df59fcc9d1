#!/bin/bash
# round 3: the finishing kernel (parked gaps served by groups of lanes, bubbles from LDS) against the one-lane form, on the three bench workloads
# usage (GPU box, repository root): scripts/r3_finish.sh <tag> [workloads...]
tag=${1:-r3_finish}; shift
wl=${@:-human-indel human-het human}
out=gpurun_out/$tag
mkdir -p $out
for w in $wl; do
  for mode in ${MODES:-classic g16 g64}; do
    case $mode in
      classic) export MTG_CLASSIC_WALK=1; unset MTG_FINISH_G MTG_ROUNDS;;
      g16) unset MTG_CLASSIC_WALK; export MTG_FINISH_G=16 MTG_ROUNDS=0;;
      g64) unset MTG_CLASSIC_WALK; export MTG_FINISH_G=64 MTG_ROUNDS=0;;
      auto) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS MTG_PARK_SNP;;
    snp0) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS; export MTG_PARK_SNP=0;;
    snp1) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS; export MTG_PARK_SNP=1;;
      o*r*) unset MTG_CLASSIC_WALK; unset MTG_BUBBLE_GROUPS; export MTG_FINISH_G=16 MTG_ROUNDS=${mode#*r};;
      g*r*) unset MTG_CLASSIC_WALK; export MTG_BUBBLE_GROUPS=1; g=${mode#g}; export MTG_FINISH_G=${g%r*} MTG_ROUNDS=${mode#*r};;
    esac
    timeout 600 python bench.py --workload $w --batches ${NBATCH:-3} --in-flight ${INFLIGHT:-6} --cpu-sites 0 --no-ceiling --no-secondary --steps 20 --warmup 5 > $out/bench_${w}_${mode}.json 2> $out/bench_${w}_${mode}.err
    python - $out/bench_${w}_${mode}.json $w $mode <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], "value %.1f M/s" % (d["value"]/1e6), "ms/step %.3f" % d["ms_per_step"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["stage_ms_per_batch"].items()}, "alone", d["roofline"].get("one_batch_alone_ms"), "identical", d.get("filled_sequences_identical_to_truth"), "filled", d.get("filled"), "/", d.get("sites_verified"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done 2>&1 | tee $out/summary.txt
