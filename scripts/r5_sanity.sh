#!/bin/bash
# round 5: the less-travelled paths after this round's changes -- bench.py on the small workloads and entries, and the GPU tests whose walks cross
# bubbles / whose gaps are multi-contig under the variants of the tuning table that change who does what: the light walk kernel forced on and off,
# the host's general path, no lean gaps, several launches per batch (the head region and the work lists across launches), the tool's small batches
cd $GRAFT_REPO_ROOT
for args in "--workload tiny --host-strings --cpu-sites 0 --no-ceiling" "--workload ecoli --steps 10 --warmup 2" "--workload tiny --cpu-sites 100"; do echo "== bench.py $args"; MTG_BENCH_NO_READS=1 timeout 500 python bench.py $args 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','value_from_host_strings','value_from_host_text','tool_sites_per_s','filled_sequences_identical_to_truth')}, (d.get('cpu_baseline') or {}).get('identical_to_hip'))" || tail -5 /tmp/err.txt; done
for v in "MTG_LIGHT_WALK=1" "MTG_LIGHT_WALK=0" "MTG_LIGHT_WALK=1 MTG_MAX_CHUNK=37" "MTG_HOST_GENERAL=1" "MTG_NO_LEAN=1 MTG_LIGHT_WALK=1" "MTG_MAX_CHUNK=5 MTG_CLI_BATCH=50" "MTG_TUNING=FINISH_G=16,ROUNDS=3,MAX_CHUNK=64,LIGHT_WALK=1" "MTG_CLI_BATCH=7 MTG_LIGHT_WALK=1" "MTG_HOST_FORMAT=1 MTG_MAX_CHUNK=37" "MTG_TUNING=NO_DEFER=1,HOST_PATHS=1"; do
    echo "== tests under: $v"
    env $v timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_micro_cases.py -q -x -p no:cacheprovider -m gpu -k "cli or synthetic_sites or short_fills or reverse_attempt or diploid or allelic or replica or several or contig or fuzz_on_device or adversarial or text_batches or golden" 2>&1 | grep -E "passed|failed|rror" | tail -2
done
