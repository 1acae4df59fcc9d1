#!/bin/bash
# dry run of the N > 1 path of bench.py on ONE GPU: two ranks share device 0 (small donor), gloo instead of RCCL; strong and weak scaling
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2dry}; mkdir -p $O
for sc in strong weak; do
MTG_BENCH_ONE_DEVICE=1 MTG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --nseq 120000 --sites 20000 --scaling $sc --cpu-sites 0 --no-ceiling > $O/dry_$sc.json 2> $O/dry_$sc.err
tail -c 900 $O/dry_$sc.json; echo; tail -5 $O/dry_$sc.err
done
# the same result path over RCCL in a world of one rank (two ranks cannot share a device under RCCL): process group, the fill writing its
# sequences into the gather's device buffer (mtg_fill_prepared_serial_device), asynchronous gather per batch, at the full workload
MTG_BENCH_FORCE_GATHER=1 timeout 600 python bench.py --cpu-sites 0 --no-ceiling --no-secondary > $O/dry_rccl1.json 2> $O/dry_rccl1.err
python3 -c "
import json
d=json.loads(open('$O/dry_rccl1.json').read().strip().splitlines()[-1])
print('one rank over RCCL, sequences gathered from HBM: value %.4g ms/step %.3f gathered payload verified %s identical to truth %s' % (d['value'], d['ms_per_step'], d['gathered_payload_verified'], d['filled_sequences_identical_to_truth']))"
