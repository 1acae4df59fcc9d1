#!/bin/bash
# the evidence kept under profiles/ for round 3.  Run on the GPU box:  bash scripts/profile_round3.sh <tag>   (then copy gpurun_out/<tag>/* to profiles/r03_*)
#   (MTG_HEAD=<commit> in the environment names the code in pmc.json)
#   kernel statistics of the bench command (rocprofv3 --kernel-trace --stats), with six batches in flight and with ONE (the kernels' own times);
#   the two PMC passes (each on its own, no trace domain); the default bench line (PMC traffic of this very code in it); the N > 1 result
#   path on one GPU (two gloo ranks; one RCCL rank); the input-side rates with two pool threads; the walk modes on the secondary workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r3p}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="--cpu-sites 0 --no-ceiling --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 bench.py $B --repeats 3 > $O/bench_under_rocprof.json 2> $O/stats.err
python3 scripts/aggregate_profiles.py stats $O/stats $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o stats -- python3 bench.py $B --in-flight 1 --steps 20 --warmup 4 --repeats 2 > $O/bench_one_batch_in_flight.json 2> $O/stats1.err
python3 scripts/aggregate_profiles.py stats $O/stats1 $O/kernel_stats_one_batch_in_flight.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rm -rf $O/stats $O/stats1 $O/pmc_fetch $O/pmc_write
cp $O/pmc.json profiles/r03_pmc.json   # on the box: the bench line below carries the traffic of this very code
S=$(date +%s); python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench.py: $(( $(date +%s) - S )) s wall" > $O/bench_default_wall.txt
python3 scripts/pcie_d2h.py > $O/pcie_d2h.txt 2>&1
# the N > 1 result path: two gloo ranks on the one device (small donor), strong scaling with both site sets; one RCCL rank at the full workload
MTG_BENCH_ONE_DEVICE=1 MTG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --nseq 120000 --sites 20000 --cpu-sites 0 --no-ceiling > $O/dry_two_ranks_gloo.json 2> $O/dry_two_ranks_gloo.err
MTG_BENCH_FORCE_GATHER=1 timeout 600 python bench.py $B > $O/dry_one_rank_rccl.json 2> $O/dry_one_rank_rccl.err
{
 echo "# bench.py input-side rates (M breakpoints/s) with the library's worker pool at 2 threads and at its default: prepared batches (value), host strings (mtg_fill_batch), one text block per batch (mtg_fill_text), the tool"
 for th in 2 0; do
   if [ $th = 0 ]; then unset MTG_POOL_THREADS; else export MTG_POOL_THREADS=$th; fi
   python3 bench.py --cpu-sites 0 --no-ceiling --no-children 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MTG_POOL_THREADS=%s prepared %.1f  host strings %.1f  host text %.1f  sequences left in HBM %.1f  tool %.2f' % ('$th' if '$th' != '0' else 'default', d['value']/1e6, d['value_from_host_strings']/1e6, d['value_from_host_text']/1e6, d['value_sequences_left_in_hbm']/1e6, d['tool_sites_per_s']/1e6))"
 done
 unset MTG_POOL_THREADS
 cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > $O/host_threads.txt 2>&1
timeout 2400 python -u -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|human-scale index|config 4" | tail -n 8 > $O/gpu_tests.txt
tail -c 300 $O/bench_default.json; echo; cat $O/bench_default_wall.txt $O/host_threads.txt; grep -E "k_stage_a|k_finish|k_bubble|k_copy|k_post|k_emit|k_scan|k_marshal" $O/kernel_stats_one_batch_in_flight.csv | cut -c1-160
