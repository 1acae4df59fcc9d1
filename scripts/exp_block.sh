#!/bin/bash
# experiment: gaps per block of the chunk pass (RESULT_BLOCK) -- rebuilds the library on the GPU box for each value, restores the default
cd $GRAFT_REPO_ROOT
for b in ${BLOCKS:-512 256 128 512 256}; do
  make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc EXTRA="-DMTG_RESULT_BLOCK_V=$b" 2>&1 | grep -E " error" 
  echo "RESULT_BLOCK=$b"; THREADS="16" FLIGHT="3" STEPS=300 bash scripts/sweep_pool.sh | head -1 | cut -c1-60
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E " error"
