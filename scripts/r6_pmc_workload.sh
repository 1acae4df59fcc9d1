#!/bin/bash
# round 6: memory-request counters of one workload's fill kernels, one batch in flight, each counter in a pass of its own (no trace domain)
#   bash scripts/r6_pmc_workload.sh <tag> <workload> [counters...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6pmc}; W=${2:-human-indel}; shift; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
B="--cpu-sites 0 --no-ceiling --no-secondary --no-children --workload $W --in-flight 1 --steps 6 --warmup 2 --repeats 1 --batches 3"
for C in ${@:-FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum}; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o pmc -- python3 bench.py $B --detail $O/detail_$C.json > /dev/null 2> $O/pmc_$C.err
done
python3 - "$O" <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mtgi::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(os.path.join(O, "pmc.json"), "w"), indent=1)
for k, cs in sorted(out.items()):
    if any(x in k for x in ("k_stage_a", "k_walk", "k_finish", "k_post", "k_emit", "k_copy")):
        print(k, {c: round(v["avg"], 1) for c, v in cs.items()})
PY
rm -rf $O/pmc_*/
