#!/bin/bash
# the evidence kept under profiles/ for round 2: default bench line, diploid line, rocprofv3 kernel statistics of the same command, the two
# PMC passes (each on its own, no trace domain), host-thread sweep, per-batch straggler diagnosis.  Run on the GPU box:
#   MTG_HEAD=<commit> bash scripts/profile_round2.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r2p}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --repeats 3 > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py stats $O/stats $O/kernel_stats.csv
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
cp $O/pmc.json profiles/r02_pmc.json   # on the box: the bench lines below carry the traffic of this very code
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --workload human-het --batches 3 --cpu-sites 10000 > $O/bench_het.json 2> $O/bench_het.err
bash scripts/r2_alone.sh $T/alone_run > /dev/null 2>&1; cp $O/alone_run/alone_kernel_stats.txt $O/kernel_stats_one_batch_alone.txt; rm -rf $O/alone_run
bash scripts/r2_pmc_insts.sh $T/insts_run > /dev/null 2>&1; cp $O/insts_run/insts.txt $O/instructions_per_wave.txt; rm -rf $O/insts_run
HET=1 bash scripts/r2_pmc_insts.sh $T/insts_run > /dev/null 2>&1; { echo "# diploid set"; cat $O/insts_run/insts.txt; } >> $O/instructions_per_wave.txt; rm -rf $O/insts_run
{ python3 scripts/diag_scaling.py 2>&1 | grep "^gaps"; echo "# diploid set"; HET=1 python3 scripts/diag_scaling.py 2>&1 | grep "^gaps"; } > $O/kernel_time_by_launch_size.txt
{
 echo "# bench.py (prepared batches / host strings) with the library's worker pool at 2 threads and at its default (CPU budget of the box)"
 for th in 2 16; do for mode in "" "--host-strings"; do
   MTG_POOL_THREADS=$th python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MTG_POOL_THREADS=$th $mode value %.4g breakpoints/s  ms/step %.3f  (min %.3f max %.3f, %d blocks)' % (d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step_min'], d['timed_blocks']['ms_per_step_max'], d['timed_blocks']['blocks']))"
 done; done
 grep -E "nr_throttled|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null
 cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > $O/host_threads.txt 2>&1
python3 scripts/diag_batches.py 4 pieces 2>&1 | grep -v amdgpu.ids > $O/straggler_batches.txt
tail -c 400 $O/bench_default.json; cat $O/host_threads.txt; grep -E "k_stage_a|k_copy|k_post|k_emit|k_scan" $O/kernel_stats.csv | cut -c1-200
