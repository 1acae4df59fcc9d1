#!/bin/bash
# round 6: the kernels of the indel (and het) secondary with ONE batch in flight under rocprofv3 --kernel-trace --stats (every kernel's own time,
# k_paths / k_general included), and the same with six in flight.   bash scripts/r6_indel_profile.sh <tag> [workload]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6i}; W=${2:-human-indel}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="--cpu-sites 0 --no-ceiling --no-secondary --no-children --workload $W"
export MTG_KERNEL_TIMERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o stats -- python3 bench.py $B --in-flight 1 --steps 12 --warmup 4 --repeats 2 --detail $O/detail1.json > $O/bench_one.json 2> $O/stats1.err
python3 scripts/aggregate_profiles.py stats $O/stats1 $O/kernel_stats_one_batch_in_flight.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats6 -o stats -- python3 bench.py $B --steps 20 --warmup 4 --repeats 3 --detail $O/detail6.json > $O/bench_six.json 2> $O/stats6.err
python3 scripts/aggregate_profiles.py stats $O/stats6 $O/kernel_stats.csv
rm -rf $O/stats1 $O/stats6
cut -c1-160 $O/kernel_stats_one_batch_in_flight.csv | head -30; echo; cut -c1-160 $O/kernel_stats.csv | head -30; tail -c 400 $O/bench_six.json
