#!/bin/bash
# experiment: what the parts of k_post cost -- builds with the coverage pass (bit 0), the terminal search (bit 1) or both switched off
# (-DMTG_POST_DBG=n: results are wrong, only times and instruction counts mean anything); the default build is restored at the end
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 1 2 3; do
  make -C mindthegap_amd/csrc clean >/dev/null
  if [ $v = 0 ]; then make -C mindthegap_amd/csrc 2>&1 | grep -E " error"; else make -C mindthegap_amd/csrc EXTRA="-DMTG_POST_DBG=$v" 2>&1 | grep -E " error"; fi
  echo "MTG_POST_DBG=$v"
  bash scripts/r2_pmc_insts.sh exp_post_$v | grep k_post
  bash scripts/r2_alone.sh exp_post_$v | grep k_post
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E " error"
