#!/bin/bash
# experiment: the library built with extra compiler flags (one variant per argument, "" = the default build), per-batch kernel times of
# the haploid and the diploid bench set for each; the default build is restored at the end
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  make -C mindthegap_amd/csrc clean >/dev/null
  make -C mindthegap_amd/csrc EXTRA="$v" 2>&1 | grep -E " error"
  echo "EXTRA=$v"
  python3 scripts/diag_batches.py 2 2>&1 | grep "^batch" | cut -c1-90
  HET=1 python3 scripts/diag_batches.py 1 2>&1 | grep "^batch" | cut -c1-90
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E " error"
