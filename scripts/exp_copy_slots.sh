#!/bin/bash
# experiment: how many batches may copy their results to the host at once (MTG_COPY_SLOTS), default bench workload
cd $GRAFT_REPO_ROOT
for sl in ${SLOTS:-0 1 2 3}; do
  for fl in ${FLIGHT:-6}; do
    MTG_COPY_SLOTS=$sl python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --in-flight $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('copy slots $sl in flight $fl: value %.4g ms/step %.3f (min %.3f max %.3f)'%(d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], d['timed_blocks']['ms_per_step_max']))"
  done
done
