#!/bin/bash
# parity suite, tiny bench (script errors show up in seconds), default bench, one batch in flight, diploid
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2d}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py --workload tiny --steps 4 --warmup 1 > $O/bench_tiny.json 2> $O/bench_tiny.err || tail -20 $O/bench_tiny.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -3 $O/bench.err
MTG_DEBUG_TIMERS=1 timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --in-flight 1 --steps 6 --warmup 2 --repeats 2 --batches 2 > $O/bench_if1.json 2> $O/bench_if1.err; tail -c 300 $O/bench_if1.json
timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --workload human-het --batches 2 > $O/bench_het.json 2> $O/bench_het.err; tail -c 300 $O/bench_het.json
