#!/bin/bash
# round 4: the tool's rate with the text formatted on the device and on the host (MTG_HOST_FORMAT=1), pool sizes, where the time goes; the
# write ceiling of the memory-backed file system; the container load in a fresh process.   bash scripts/r4_tool.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r4t}; O=gpurun_out/$T; mkdir -p $O
python3 scripts/tmpfs_write_ceiling.py > $O/tmpfs_write_ceiling.txt 2>&1
B="--cpu-sites 0 --no-ceiling --no-children"
{
 for mode in device host; do
  for th in default 2; do
   if [ $mode = host ]; then export MTG_HOST_FORMAT=1; else unset MTG_HOST_FORMAT; fi
   if [ $th = default ]; then unset MTG_POOL_THREADS; else export MTG_POOL_THREADS=$th; fi
   MTG_BENCH_NO_E2E=1 MTG_TOOL_TIMERS=1 python3 bench.py $B 2> $O/tool_${mode}_${th}.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tool']; print('format on the $mode, MTG_POOL_THREADS=$th: tool %.2f M sites/s (%.3f s, %.2f GB/s out, identical %s)  prepared %.1f  text %.1f' % (d['tool_sites_per_s']/1e6, t['seconds'], t['output_GBps'], t['sequences_identical_to_truth'], d['value']/1e6, d['value_from_host_text']/1e6))"
   grep "\[tool\]" $O/tool_${mode}_${th}.err | tail -2
  done
 done
 unset MTG_HOST_FORMAT MTG_POOL_THREADS
} > $O/tool_rates.txt 2>&1
cat $O/tmpfs_write_ceiling.txt $O/tool_rates.txt
