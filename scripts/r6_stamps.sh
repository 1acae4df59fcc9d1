#!/bin/bash
# round 6: the diagnostic build with shader-clock stamps (mindthegap_amd/lib_stamps, built here with `make OUT=../lib_stamps EXTRA=-DMTG_STAMPS`,
# never the product) on the secondary workloads: where a lane's life goes in the walk kernel.   bash scripts/r6_stamps.sh <out-file> [workloads...]
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r6_walk_stamps.txt}; shift
export MTG_LIBRARY_PATH=$GRAFT_REPO_ROOT/mindthegap_amd/lib_stamps/libmtgfill.so
: > $O
for W in ${@:-human-indel human-het human}; do
  echo "== $W" >> $O
  timeout 600 python3 bench.py --cpu-sites 0 --no-ceiling --workload $W --steps 3 --warmup 1 --repeats 1 --batches 1 --no-secondary --no-children --in-flight 1 --detail gpurun_out/stamps_detail.json 2>&1 | grep -E "stamps" | tail -5 >> $O
done
cat $O
