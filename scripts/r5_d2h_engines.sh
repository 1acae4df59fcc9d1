#!/bin/bash
# round 5: which engine carries the result copies?  Device-to-host copies into page-locked memory (scripts/pcie_d2h.py) under a kernel trace --
# __amd_rocclr_copyBuffer in the list means the copies ran as blit kernels on the CUs, its absence that a copy engine (SDMA) took them -- with the
# runtime's default and with the environment switches that steer the choice.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r5d2h}; mkdir -p $O
for v in "default" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_SDMA=1" "HIP_FORCE_DEV_KERNARG=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=65536"; do
  echo "=== $v"
  rm -rf $O/t
  if [ "$v" = "default" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python3 scripts/pcie_d2h.py 2> /dev/null | grep -E " 32 MB| 4 MB x 6"
  else export $v; rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o s -- python3 scripts/pcie_d2h.py 2> /dev/null | grep -E " 32 MB| 4 MB x 6"; unset ${v%%=*}; fi
  f=$(find $O/t -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then grep -E "copyBuffer|Name" $f | cut -c1-120; else echo "(no kernel ran: the copies went to a copy engine)"; fi
done
rm -rf $O/t
