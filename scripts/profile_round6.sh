#!/bin/bash
# the evidence kept under profiles/ for round 6.  Run on the GPU box:  bash scripts/profile_round5.sh <tag>   (then copy gpurun_out/<tag>/* to profiles/r06_*)
#   (MTG_HEAD=<commit> in the environment names the code in the pmc files)
#   fill kernels: rocprofv3 --kernel-trace --stats of the bench command with six batches in flight and with ONE (the kernels' own times), the two PMC
#   passes (each on its own, no trace domain); index construction: the same three for scripts/r4_build.py (config-4 index from the donor in HBM);
#   the sequence scan (k_scan) at config-4 size: kernel statistics and the two PMC passes; small batches (config 5's share of one rank); then the
#   default bench line (it replays the PMC files of this very run).  Every rocprofv3 command has the program itself after `--`.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6p}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="--cpu-sites 0 --no-ceiling --no-secondary"
export MTG_KERNEL_TIMERS=1   # the profiled runs record an event between the kernels, as round 4's did (the bench's timed blocks below do not)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py $B --repeats 3 > $O/bench_under_rocprof.json 2> $O/stats.err
python3 scripts/aggregate_profiles.py stats $O/stats $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o stats -- python3 bench.py $B --in-flight 1 --steps 20 --warmup 4 --repeats 2 > $O/bench_one_batch_in_flight.json 2> $O/stats1.err
python3 scripts/aggregate_profiles.py stats $O/stats1 $O/kernel_stats_one_batch_in_flight.csv
unset MTG_KERNEL_TIMERS
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -o stats -- python3 scripts/r4_build.py > $O/build_under_rocprof.txt 2> $O/bstats.err
python3 scripts/aggregate_profiles.py stats $O/bstats $O/build_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bpmc_fetch -o pmc -- python3 scripts/r4_build.py > /dev/null 2> $O/bpmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bpmc_write -o pmc -- python3 scripts/r4_build.py > /dev/null 2> $O/bpmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/bpmc_fetch $O/bpmc_write $O/pmc_build.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sstats -o stats -- python3 scripts/bench_scan.py > $O/scan_kernel.json 2> $O/sstats.err
python3 scripts/aggregate_profiles.py stats $O/sstats $O/scan_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/spmc_fetch -o pmc -- python3 scripts/bench_scan.py > /dev/null 2> $O/spmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/spmc_write -o pmc -- python3 scripts/bench_scan.py > /dev/null 2> $O/spmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/spmc_fetch $O/spmc_write $O/pmc_scan.json
rm -rf $O/stats $O/stats1 $O/pmc_fetch $O/pmc_write $O/bstats $O/bpmc_fetch $O/bpmc_write $O/sstats $O/spmc_fetch $O/spmc_write
cp $O/pmc.json profiles/r06_pmc.json; cp $O/pmc_build.json profiles/r06_pmc_build.json   # on the box: the bench line below carries the traffic of this very code
python3 scripts/r4_build.py > $O/build_lean.txt 2>&1
python3 scripts/r5_small_batches.py --nseq 600000 --sites 12500 100000 --in-flight 6 1 > $O/small_batches.txt 2>&1
# the indel secondary's kernels on their own (one batch in flight) and its request counters
bash scripts/r6_indel_profile.sh $T/indel human-indel > /dev/null 2>&1
cp gpurun_out/$T/indel/kernel_stats_one_batch_in_flight.csv $O/kernel_stats_indel_one_batch_in_flight.csv 2>/dev/null; cp gpurun_out/$T/indel/kernel_stats.csv $O/kernel_stats_indel.csv 2>/dev/null
bash scripts/r6_pmc_workload.sh $T/pmc_indel human-indel FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU > /dev/null 2>&1
cp gpurun_out/$T/pmc_indel/pmc.json $O/pmc_indel.json 2>/dev/null
sleep 5
S=$(date +%s); python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench.py: $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
cp bench_detail.json $O/bench_default_detail.json 2>/dev/null
tail -c 600 $O/bench_default.json; echo; cat $O/bench_default_wall.txt; grep -E "k_stage_a|k_finish|k_lean|k_copy|k_post|k_emit|k_scan|k_general|k_paths|copyBuffer|fillBuffer" $O/kernel_stats_one_batch_in_flight.csv | cut -c1-150; cut -c1-150 $O/build_kernel_stats.csv | head -14; grep -E "k_scan|copyBuffer" $O/scan_kernel_stats.csv | cut -c1-150
