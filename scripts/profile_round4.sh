#!/bin/bash
# the evidence kept under profiles/ for round 4.  Run on the GPU box:  bash scripts/profile_round4.sh <tag>   (then copy gpurun_out/<tag>/* to profiles/r04_*)
#   (MTG_HEAD=<commit> in the environment names the code in the pmc files)
#   fill kernels: rocprofv3 --kernel-trace --stats of the bench command with six batches in flight and with ONE (the kernels' own times), the two PMC
#   passes (each on its own, no trace domain); index construction: the same three for scripts/r4_build.py (config-4 index from the donor in HBM);
#   the default bench line (replays the PMC files of this very run); the N > 1 result path on one GPU; tool rates and the write ceiling of the
#   memory-backed file system; container load; the reads-built workload; the GPU test suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r4p}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="--cpu-sites 0 --no-ceiling --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py $B --repeats 3 > $O/bench_under_rocprof.json 2> $O/stats.err
python3 scripts/aggregate_profiles.py stats $O/stats $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o stats -- python3 bench.py $B --in-flight 1 --steps 20 --warmup 4 --repeats 2 > $O/bench_one_batch_in_flight.json 2> $O/stats1.err
python3 scripts/aggregate_profiles.py stats $O/stats1 $O/kernel_stats_one_batch_in_flight.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -o stats -- python3 scripts/r4_build.py > $O/build_under_rocprof.txt 2> $O/bstats.err
python3 scripts/aggregate_profiles.py stats $O/bstats $O/build_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bpmc_fetch -o pmc -- python3 scripts/r4_build.py > /dev/null 2> $O/bpmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bpmc_write -o pmc -- python3 scripts/r4_build.py > /dev/null 2> $O/bpmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/bpmc_fetch $O/bpmc_write $O/pmc_build.json
rm -rf $O/stats $O/stats1 $O/pmc_fetch $O/pmc_write $O/bstats $O/bpmc_fetch $O/bpmc_write
cp $O/pmc.json profiles/r04_pmc.json; cp $O/pmc_build.json profiles/r04_pmc_build.json   # on the box: the bench line below carries the traffic of this very code
# the default bench BEFORE the legacy build: a process that exits with 200 GB allocated leaves the driver wiping that memory, and the next
# process's first large hipMalloc waits for it (scripts/hipmalloc_latency.py: 0.3 ms on a quiet device, 6.6 s behind 49 GB of freed pieces) --
# round 4's earlier bench lines carried 4 s of "hipmalloc_seconds" for that reason alone
sleep 5
S=$(date +%s); python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench.py: $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
python3 scripts/r4_build.py > $O/build_lean.txt 2>&1
MTG_LEGACY_BUILD=1 python3 scripts/r4_build.py > $O/build_legacy.txt 2>&1
sleep 5
python3 scripts/r4_load.py > $O/container_load.txt 2>&1
bash scripts/r4_load_sweep.sh > $O/container_load_fresh_processes.txt 2>&1
MTG_BENCH_ONE_DEVICE=1 MTG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --nseq 120000 --sites 20000 --cpu-sites 0 --no-ceiling > $O/dry_two_ranks_gloo.json 2> $O/dry_two_ranks_gloo.err
MTG_BENCH_FORCE_GATHER=1 timeout 600 python bench.py $B > $O/dry_one_rank_rccl.json 2> $O/dry_one_rank_rccl.err
bash scripts/r4_tool.sh $T/tool > /dev/null 2>&1; cp $O/tool/tool_rates.txt $O/tool_rates.txt; cp $O/tool/tmpfs_write_ceiling.txt $O/tmpfs_write_ceiling.txt; rm -rf $O/tool
grep "^{" $O/bench_default.json | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); json.dump(d.get('secondary_reads_built'), open('$O/reads_built_workload.json','w'), indent=1)"
timeout 2700 python -u -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -n 4 > $O/gpu_tests.txt
tail -c 400 $O/bench_default.json; echo; cat $O/bench_default_wall.txt $O/tool_rates.txt $O/gpu_tests.txt; grep -E "k_stage_a|k_finish|k_lean|k_copy|k_post|k_emit|k_scan|k_fmt" $O/kernel_stats_one_batch_in_flight.csv | cut -c1-160; cut -c1-160 $O/build_kernel_stats.csv | head -20
