#!/bin/bash
# round 6: the binning kernels of the partitioned junction table with a tile's records put in bin order in LDS before they are written (the library as built)
# and in the order they come (mindthegap_amd/lib_unsorted: make OUT=../lib_unsorted EXTRA=-DMTG_BIN_SORTED=0).   bash scripts/r6_binsort_ab.sh [nseq] [sites]
cd $GRAFT_REPO_ROOT
N=${1:-600000}; S=${2:-20000}
for L in lib_unsorted lib; do
  echo "== $L nseq=$N"
  MTG_LIBRARY_PATH=$GRAFT_REPO_ROOT/mindthegap_amd/$L/libmtgfill.so timeout 600 python3 scripts/r4_build.py $N $S 2>&1 | grep -vE "^\{|device_run|fill_batch" | cut -c1-200
done
