#!/bin/bash
# kernel + copy timeline of the default bench and variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r2t}; mkdir -p $O
for v in "default:" "b1:--batches 1" "hs:--host-strings"; do
  tag=${v%%:*}; args=${v#*:}
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_$tag -o t -- python3 bench.py --cpu-sites 0 --no-ceiling --no-secondary --steps 30 --warmup 4 --repeats 1 $args > $O/tr_$tag.json 2> $O/tr_$tag.err
  echo "== $tag"; python3 scripts/gpu_timeline.py $O/tr_$tag 24 ${2:-0} | head -${3:-40}
  python3 -c "import json;d=json.load(open('$O/tr_$tag.json'));print('   bench value %.4g ms/step %.3f'%(d['value'],d['ms_per_step']))"
  rm -rf $O/tr_$tag
done
