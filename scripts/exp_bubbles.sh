#!/bin/bash
# timing experiments on both workloads (GPU box): noinline vs forceinline bubble code
for cfg in "" "-DMTG_INLINE_ALL"; do
  make -C mindthegap_amd/csrc -B EXTRA="$cfg" > /dev/null 2>&1
  echo "=== EXTRA=$cfg"
  for w in human human-het; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --cpu-sites 0 --no-ceiling 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w kernel_ms %.2f reads %.3e value %.0f' % (d['roofline']['avg_kernel_ms'], d['roofline']['bucket_reads_per_launch'], d['value']))"
  done
done
