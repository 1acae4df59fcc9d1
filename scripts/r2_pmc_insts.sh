#!/bin/bash
# instruction counts per kernel (one pass of SQ counters) on single batches: rocprofv3 --pmc of scripts/diag_batches.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2n}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/pmc -o pmc -- python3 scripts/diag_batches.py 1 ${2:-} > /dev/null 2> $O/pmc.err
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if not any(x in n for x in ("k_stage_a", "k_copy", "k_post", "k_scan", "k_emit")): continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": calls[n] += 1
out = open(sys.argv[1] + "/insts.txt", "w")
for n in acc:
    w = acc[n]["SQ_WAVES"] or 1
    line = "%-18s launches %d waves/launch %.0f  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f SMEM %.0f" % (n, calls[n], w / calls[n], acc[n]["SQ_INSTS_VALU"] / w, acc[n]["SQ_INSTS_SALU"] / w, acc[n]["SQ_INSTS_LDS"] / w, acc[n]["SQ_INSTS_VMEM_RD"] / w, acc[n]["SQ_INSTS_VMEM_WR"] / w, acc[n]["SQ_INSTS_SMEM"] / w)
    print(line); out.write(line + "\n")
PY
