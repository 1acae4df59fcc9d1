#!/bin/bash
# round 6: the index construction with the junction table built by scattered insertion (BUILD_PARTITIONED=0) and partition by partition (=1):
# phases, statistics of the graph, fills against the truth.   bash scripts/r6_build_ab.sh [nseq] [sites]
cd $GRAFT_REPO_ROOT
N=${1:-600000}; S=${2:-20000}
for P in 0 1; do
  echo "== BUILD_PARTITIONED=$P nseq=$N"
  MTG_BUILD_PARTITIONED=$P timeout 600 python3 scripts/r4_build.py $N $S 2>&1 | grep -vE "^\{" | cut -c1-200
  MTG_BUILD_PARTITIONED=$P timeout 600 python3 scripts/r4_build.py $N $S 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['info']; print('   info:', {k:i[k] for k in ('nb_solid_kmers','nb_branching','nb_unitigs','device_bytes') if k in i})"
done
