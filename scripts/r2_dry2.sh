#!/bin/bash
# dry run of the N > 1 path of bench.py on ONE GPU: two ranks share device 0 (small donor), gloo instead of RCCL; strong and weak scaling
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2dry}; mkdir -p $O
for sc in strong weak; do
MTG_BENCH_ONE_DEVICE=1 MTG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --nseq 120000 --sites 20000 --scaling $sc --cpu-sites 0 --no-ceiling > $O/dry_$sc.json 2> $O/dry_$sc.err
tail -c 900 $O/dry_$sc.json; echo; tail -5 $O/dry_$sc.err
done
# and with RCCL on a single rank's device pair is impossible here; the one-rank RCCL gather is covered by the pytest suite
