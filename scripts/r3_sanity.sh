for args in "--workload tiny --host-strings --cpu-sites 0 --no-ceiling" "--workload ecoli --steps 10 --warmup 2" "--workload tiny --cpu-sites 100"; do echo "== bench.py $args"; timeout 500 python bench.py $args 2>/tmp/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','value_from_host_strings','value_from_host_text','tool_sites_per_s','filled_sequences_identical_to_truth')}, (d.get('cpu_baseline') or {}).get('identical_to_hip'), (d.get('cpu_baseline') or {}).get('same_algorithm_value'), d.get('tool',{}).get('error'))" || tail -5 /tmp/err.txt; done
