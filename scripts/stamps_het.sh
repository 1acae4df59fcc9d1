#!/bin/bash
# diagnostic: per-phase shader-clock stamps of k_stage_a on the diploid workload (build with -DMTG_STAMPS, then restore the product build)
cd $GRAFT_REPO_ROOT
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc EXTRA="-DMTG_STAMPS" 2>&1 | grep -E "error"
timeout 600 python bench.py --cpu-sites 0 --no-ceiling --workload ${1:-human-het} --steps 3 --warmup 1 --repeats 1 --batches 1 --no-secondary --in-flight 1 2>&1 | grep -E "stamps" | tail -4
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E "error"
