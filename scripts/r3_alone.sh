#!/bin/bash
# round 3: per-kernel time of single batches, one at a time on the device (rocprofv3 --kernel-trace --stats of scripts/diag_batches.py), per walk mode
# usage: [HET=1 INDEL=1] MODES="classic g16r0 g16r3" scripts/r3_alone.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r3_alone}; mkdir -p $O
for mode in ${MODES:-classic g16r0 g16r3}; do
  case $mode in
    classic) export MTG_CLASSIC_WALK=1; unset MTG_FINISH_G MTG_ROUNDS;;
    auto) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS MTG_PARK_SNP;;
    snp0) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS; export MTG_PARK_SNP=0;;
    snp1) unset MTG_CLASSIC_WALK MTG_FINISH_G MTG_ROUNDS MTG_BUBBLE_GROUPS; export MTG_PARK_SNP=1;;
    o*r*) unset MTG_CLASSIC_WALK; unset MTG_BUBBLE_GROUPS; export MTG_FINISH_G=16 MTG_ROUNDS=${mode#*r};;
    g*r*) unset MTG_CLASSIC_WALK; export MTG_BUBBLE_GROUPS=1; g=${mode#g}; export MTG_FINISH_G=${g%r*} MTG_ROUNDS=${mode#*r};;
  esac
  rm -rf $O/alone_$mode
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone_$mode -o alone -- python3 scripts/diag_batches.py ${NB:-2} > $O/alone_$mode.out 2> $O/alone_$mode.err
  python3 - $O $mode <<'PY'
import csv, glob, sys
O, mode = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/alone_" + mode + "/**/*kernel_stats.csv", recursive=True)
out = open(O + "/alone_%s_kernel_stats.txt" % mode, "w")
print("==", mode)
if not f:
    print("no stats"); sys.exit(0)
open(O + "/alone_%s_kernel_stats.csv" % mode, "w").write(open(f[0]).read())
for r in csv.DictReader(open(f[0])):
    n = r["Name"].split("(")[0]
    if any(x in n for x in ("k_stage_a", "k_bubble", "k_finish", "k_copy", "k_post", "k_scan", "k_emit", "k_wire")):
        line = "%-40s calls %4s  total %9.1f us  avg %8.1f us  min %8.1f  max %8.1f" % (n[:40], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
        print(line); out.write(line + "\n")
PY
  grep "^batch" $O/alone_$mode.out | cut -c1-200
  rm -rf $O/alone_$mode
done 2>&1 | tee $O/alone_summary.txt
