#!/bin/bash
# host threads of the pool x steps in flight on the headline workload; prints throughput and how often the cgroup was throttled
cd $GRAFT_REPO_ROOT
for t in ${THREADS:-8 12 14 16 24}; do for f in ${FLIGHT:-1 2}; do
  a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
  MTG_POOL_THREADS=$t python bench.py --workload ${WORKLOAD:-human} --cpu-sites 0 --no-ceiling --steps ${STEPS:-100} --warmup 6 --in-flight $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pool', $t, 'in_flight', $f, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms', d['filled_sequences_identical_to_truth'], d['stage_ms_per_step'])"
  b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); echo "   throttled periods during the run (incl. set-up): $((b-a))"
done; done
