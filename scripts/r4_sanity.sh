#!/bin/bash
# the less-travelled paths after round 4's kernel changes: bench.py on the small workloads and entries, and the synthetic / short-fill / reverse
# GPU tests under the index and launch variants of the tuning table (no lean gaps: every gap through the general k_post / k_emit over the
# list; no unitig store; dense tables; the legacy build; several launches per batch; one stream for k_post; own upload streams)
for args in "--workload tiny --host-strings --cpu-sites 0 --no-ceiling" "--workload ecoli --steps 10 --warmup 2" "--workload tiny --cpu-sites 100"; do echo "== bench.py $args"; MTG_BENCH_NO_READS=1 timeout 500 python bench.py $args 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','value_from_host_strings','value_from_host_text','tool_sites_per_s','filled_sequences_identical_to_truth')}, (d.get('cpu_baseline') or {}).get('identical_to_hip'), (d.get('tool') or {}).get('error'))" || tail -5 /tmp/err.txt; done
for v in "" "MTG_NO_LEAN=1" "MTG_MAX_CHUNK=37" "MTG_UPLOAD_OWN_STREAM=1" "MTG_TUNING=NO_DEFER=1,HOST_PATHS=1" "MTG_HOST_FORMAT=1"; do
    echo "== tests under: ${v:-defaults}"
    env $v timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "synthetic_sites or short_fills or reverse_attempt or diploid_bubbles or allelic or text_batches" 2>&1 | grep -E "passed|failed|rror" | tail -2
done
