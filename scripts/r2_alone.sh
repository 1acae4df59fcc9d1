#!/bin/bash
# kernel statistics of single batches, one at a time (nothing else on the device): rocprofv3 --kernel-trace --stats of scripts/diag_batches.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2m}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone -o alone -- python3 scripts/diag_batches.py 2 > /dev/null 2> $O/alone.err
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/alone/**/*kernel_stats.csv", recursive=True)[0]
out = open(sys.argv[1] + "/alone_kernel_stats.txt", "w")
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if any(x in n for x in ("k_stage_a", "k_copy", "k_post", "k_scan", "k_emit", "k_encode", "copyBuffer", "fillBuffer")):
        line = "%-36s calls %4s  avg %8.1f us  min %8.1f  max %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
        print(line); out.write(line + "\n")
PY
