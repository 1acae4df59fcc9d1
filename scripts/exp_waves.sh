#!/bin/bash
# experiment: k_stage_a compiled for 3 waves per SIMD (168 VGPRs, 144 bytes more scratch per lane) against the default (210 VGPRs, 2 waves)
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-0 3 0 3}; do
  make -C mindthegap_amd/csrc clean >/dev/null
  if [ $v = 0 ]; then make -C mindthegap_amd/csrc 2>&1 | grep -E " error"; else make -C mindthegap_amd/csrc EXTRA="-DMTG_STAGE_A_WAVES=$v" 2>&1 | grep -E " error"; fi
  echo "STAGE_A_WAVES=$v"
  for w in human human-het; do WORKLOAD=$w THREADS="16" FLIGHT="1 3" STEPS=${STEPS:-200} bash scripts/sweep_pool.sh | grep pool | cut -c1-110; done
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E " error"
