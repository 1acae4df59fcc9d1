#!/bin/bash
# the evidence kept under profiles/: default bench line, rocprofv3 kernel statistics of the same command, the two PMC passes (each on its
# own, no trace domain), the diploid workload.  Run on the GPU box: bash scripts/profile_round.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r1}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --workload human-het > $O/bench_het.json 2> $O/bench_het.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py --cpu-sites 0 --no-ceiling > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py --cpu-sites 0 --no-ceiling --steps 4 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py --cpu-sites 0 --no-ceiling --steps 4 --warmup 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py stats $O/stats $O/kernel_stats.csv
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
tail -c 300 $O/bench_default.json; grep -E "k_stage_a|k_post" $O/kernel_stats.csv | cut -c1-40,200-400
