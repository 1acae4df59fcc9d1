#!/bin/bash
# round 6: the index construction with every chain walked on the junction table (BUILD_POSITIONAL=0) and with the chains that lie whole in one sequence
# found by position (=1): phases, statistics of the graph, fills against the truth.   bash scripts/r6_positional_ab.sh [nseq] [sites]
cd $GRAFT_REPO_ROOT
N=${1:-600000}; S=${2:-20000}
for P in 0 1; do
  echo "== BUILD_POSITIONAL=$P nseq=$N"
  MTG_BUILD_POSITIONAL=$P MTG_DEBUG_TIMERS=1 timeout 600 python3 scripts/r4_build.py $N $S 2>&1 | grep -vE "^\{" | cut -c1-200
  MTG_BUILD_POSITIONAL=$P timeout 600 python3 scripts/r4_build.py $N $S 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['info']; print('   info:', {k:i[k] for k in ('nb_solid_kmers','nb_branching','nb_unitigs','nb_kmers_outside_unitigs','unitig_bytes','device_bytes') if k in i}, 'total_ms', round(d['profile']['total_ms'],1))"
done
