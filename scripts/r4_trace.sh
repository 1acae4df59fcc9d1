#!/bin/bash
# kernel + copy timeline of a workload under six batches in flight: how much of the time kernels of different batches run side by side.
#   gpurun -- 'bash scripts/r4_trace.sh <tag> <workload>'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r4tr}; W=${2:-human-het}; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o t -- python3 bench.py --workload $W --cpu-sites 0 --cpu-same-sites 0 --no-ceiling --no-secondary --steps 30 --warmup 4 --repeats 1 > $O/tr.json 2> $O/tr.err
python3 scripts/gpu_timeline.py $O/tr 24 0 | head -60 | tee $O/timeline_$W.txt
python3 -c "import json;d=json.loads([l for l in open('$O/tr.json') if l.startswith('{')][-1]);print('   bench value %.4g ms/step %.3f'%(d['value'],d['ms_per_step']))" | tee -a $O/timeline_$W.txt
rm -rf $O/tr
