#!/bin/bash
# experiment: k_stage_a with its registers capped for N waves per SIMD (-DMTG_STAGE_A_WAVES=N; 0 = what the default build does)
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-0 2}; do
  make -C mindthegap_amd/csrc clean >/dev/null
  if [ $v = 0 ]; then make -C mindthegap_amd/csrc 2>&1 | grep -E " error"; else make -C mindthegap_amd/csrc EXTRA="-DMTG_STAGE_A_WAVES=$v" 2>&1 | grep -E " error"; fi
  echo "STAGE_A_WAVES=$v"
  python3 scripts/diag_batches.py 2 2>&1 | grep "^batch" | cut -c1-100
  HET=1 python3 scripts/diag_batches.py 1 2>&1 | grep "^batch" | cut -c1-100
  for w in human human-het; do
    timeout 300 python bench.py --cpu-sites 0 --no-ceiling --no-secondary --workload $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'value %.4g ms/step %.3f'%(d['value'], d['ms_per_step']), {k:round(v,3) for k,v in d['stage_ms_per_batch'].items()})"
  done
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E " error"
