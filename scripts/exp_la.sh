#!/bin/bash
# timing experiment: k_stage_a against the lookahead length (steps per walk); restores the product build at the end
cd $GRAFT_REPO_ROOT
for la in 4 7 10 14; do
  make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc EXTRA="-DMTG_LA_MAX_V=$la" 2>&1 | grep -E "error"
  echo -n "LA=$la "
  timeout 600 python bench.py --cpu-sites 0 --no-ceiling --steps 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['stage_ms_per_step']['kernel'],3), 'reads/launch', r['bucket_reads_per_launch'], 'reads/s', round(r['bucket_reads_per_s']/1e9,2), d['filled_sequences_identical_to_truth'])"
done
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E "error"
