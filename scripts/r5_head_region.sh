cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/hd; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
B="--cpu-sites 0 --no-ceiling --no-secondary"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python3 bench.py $B --steps 8 --warmup 2 --repeats 1 > /dev/null 2> $O/pmc_write.err
python3 scripts/aggregate_profiles.py pmc $O/pmc_fetch $O/pmc_write $O/pmc.json
rm -rf $O/pmc_fetch $O/pmc_write
python3 bench.py $B --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json
cat $O/tests.txt; tail -c 1500 $O/bench.json
