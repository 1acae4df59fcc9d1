#!/bin/bash
# round 6: counters of the index construction's kernels (scripts/r4_build.py), each counter in a pass of its own
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6pmcb; rm -rf $O; mkdir -p $O
for C in ${@:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS TCP_TCC_WRITE_REQ_sum}; do
  rocprofv3 --pmc $C --output-format csv -d $O/p_$C -o pmc -- python3 scripts/r4_build.py 600000 200 > /dev/null 2> $O/p_$C.err
done
python3 - "$O" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(O, "p_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(w in k for w in ("k_bin", "k_build_seg", "k_jt_walk", "k_sparse_link", "k_pos_plan", "k_jt_scan", "k_us_ab")):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, cs in sorted(acc.items()):
    print(k, {c: "%.3g" % v for c, v in sorted(cs.items())})
PY
rm -rf $O/p_*/
