#!/bin/bash
# round 5: the N > 1 paths of bench.py on a one-GPU box: two gloo ranks on device 0 (host tensors), one RCCL rank with the gather forced
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/dry; rm -rf $O; mkdir -p $O
MTG_BENCH_ONE_DEVICE=1 MTG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --nseq 120000 --sites 20000 --cpu-sites 0 --no-ceiling > $O/dry_two_ranks_gloo.json 2> $O/dry_two_ranks_gloo.err
echo "gloo rc $?"; tail -c 300 $O/dry_two_ranks_gloo.json
MTG_BENCH_FORCE_GATHER=1 timeout 900 python3 bench.py --steps 20 --warmup 5 --cpu-sites 0 --no-ceiling --no-secondary > $O/dry_one_rank_rccl.json 2> $O/dry_one_rank_rccl.err
echo "rccl rc $?"; tail -c 300 $O/dry_one_rank_rccl.json
python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/gpus2_on_one_gpu.txt 2>&1; echo "--gpus 2 on a one-GPU box: rc $?"; tail -2 $O/gpus2_on_one_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
