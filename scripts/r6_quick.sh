#!/bin/bash
# round 6: one workload, quickly: the kernels of one batch alone and the rate with six in flight, for one or several builds of the library
#   bash scripts/r6_quick.sh <workload> [lib dir under mindthegap_amd ...]      (default: lib)
cd $GRAFT_REPO_ROOT
W=${1:-human-indel}; shift
for L in ${@:-lib}; do
  export MTG_LIBRARY_PATH=$GRAFT_REPO_ROOT/mindthegap_amd/$L/libmtgfill.so
  timeout 900 python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --no-children --workload $W --steps 20 --warmup 4 --repeats 3 --detail gpurun_out/quick_${W}_$L.json > /dev/null 2> gpurun_out/quick_${W}_$L.err
  python3 - "$W" "$L" <<'PY'
import json, sys
w, l = sys.argv[1:3]
d = json.load(open("gpurun_out/quick_%s_%s.json" % (w, l)))
a = d["roofline"].get("one_batch_alone_ms") or {}
print("%-12s %-10s value %.1f M/s  ms/step %.3f | alone: walk+finish %.3f (finish %.3f) copy %.3f post %.3f emit %.3f sum %.3f parked %s identical %s" % (
    w, l, d["value"] / 1e6, d["ms_per_step"], a.get("k_stage_a+k_finish", 0), a.get("k_finish", 0), a.get("k_copy", 0), a.get("k_post+scans", 0), a.get("k_emit", 0), a.get("sum", 0), a.get("parked_gaps"), d.get("filled_sequences_identical_to_truth")))
PY
done
