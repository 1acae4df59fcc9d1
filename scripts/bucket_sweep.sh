#!/bin/bash
# builds the library with different ADJ bucket sizes / load factors and runs the human-scale bench (GPU box)
for cfg in "2 0.5" "2 0.35" "4 0.5" "4 0.35" "8 0.5"; do
  set -- $cfg
  make -C mindthegap_amd/csrc -B EXTRA="-DMTG_ADJ_SLOTS=$1" > /dev/null 2>&1
  echo "=== ADJ_SLOTS=$1 load=$2"
  MTG_INDEX_LOAD=$2 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sites 0 --no-ceiling 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; c=d['config']
print('value %.0f ms/step %.1f kernel_ms %.2f lines/launch %.3e post %.2f index_GB %.1f build_s %.1f identical %s' % (d['value'], d['ms_per_step'], r['avg_kernel_ms'], r['index_lines_per_launch'], d['stage_ms_per_step']['post_kernel'], c['index_bytes']/1e9, c['index_build_s'], d['filled_sequences_identical_to_truth']))"
done
