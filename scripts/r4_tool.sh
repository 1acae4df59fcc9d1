#!/bin/bash
# round 4: the tool's rate with the text formatted on the device and on the host (MTG_HOST_FORMAT=1), pool sizes, workers per device, writer
# threads per output file, where the time goes; the write ceiling of the memory-backed file system.   bash scripts/r4_tool.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r4t}; O=gpurun_out/$T; mkdir -p $O
python3 scripts/tmpfs_write_ceiling.py > $O/tmpfs_write_ceiling.txt 2>&1
B="--cpu-sites 0 --no-ceiling --no-children"
run() { # mode writers-per-file threads inflight
  if [ $1 = host ]; then export MTG_HOST_FORMAT=1; else unset MTG_HOST_FORMAT; fi
  if [ $2 = default ]; then unset MTG_CLI_WRITERS; else export MTG_CLI_WRITERS=$2; fi
  if [ $3 = default ]; then unset MTG_POOL_THREADS; else export MTG_POOL_THREADS=$3; fi
  if [ $4 = default ]; then unset MTG_CLI_IN_FLIGHT; else export MTG_CLI_IN_FLIGHT=$4; fi
  MTG_BENCH_NO_E2E=1 MTG_BENCH_NO_READS=1 MTG_TOOL_TIMERS=1 python3 bench.py $B 2> $O/tool.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tool']; print('format on the $1, writers per file $2, MTG_POOL_THREADS=$3, workers per device $4: tool %.2f M sites/s (%.3f s, %.2f GB/s out, identical %s)  prepared %.1f  text %.1f' % (d['tool_sites_per_s']/1e6, t['seconds'], t['output_GBps'], t['sequences_identical_to_truth'], d['value']/1e6, d['value_from_host_text']/1e6))"
  grep "\[tool\]" $O/tool.err | tail -2
}
{
 run device default default default
 run device 2 default default
 run device default 2 default
 run device default default 4
 run device default default 6
 run host default default default
 run host default 2 default
 unset MTG_HOST_FORMAT MTG_POOL_THREADS MTG_CLI_WRITERS MTG_CLI_IN_FLIGHT
} > $O/tool_rates.txt 2>&1
rm -f $O/tool.err
cat $O/tmpfs_write_ceiling.txt $O/tool_rates.txt
