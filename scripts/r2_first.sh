#!/bin/bash
# first GPU look of round 2: parity suite, then the bench in three shapes (default, one step in flight, diploid)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2a}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py --cpu-sites 0 --no-ceiling > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
MTG_DEBUG_TIMERS=1 timeout 300 python bench.py --cpu-sites 0 --no-ceiling --in-flight 1 --steps 6 --warmup 2 > $O/bench_if1.json 2> $O/bench_if1.err; tail -c 700 $O/bench_if1.json; tail -40 $O/bench_if1.err
timeout 300 python bench.py --cpu-sites 0 --no-ceiling --workload human-het > $O/bench_het.json 2> $O/bench_het.err; tail -c 700 $O/bench_het.json
