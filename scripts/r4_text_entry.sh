#!/bin/bash
# value / value_from_host_strings / value_from_host_text of the headline workload under a few settings of the copy path (which engine moves
# the bytes, how many batches copy at once): where the text entry's missing 0.3 ms per batch go.  gpurun --timeout 1500 -- 'bash scripts/r4_text_entry.sh'
out=gpurun_out/r4text
mkdir -p $out
run() {
    name=$1; shift
    env MTG_BENCH_NO_READS=1 MTG_BENCH_NO_E2E=1 "$@" python bench.py --no-children --no-tool --no-ceiling --cpu-sites 0 --cpu-same-sites 0 > $out/$name.json 2> $out/$name.err
    python - $name $out/$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], "prepared %.1f strings %.1f text %.1f registered text %.1f M/s" % (d["value"] / 1e6, d.get("value_from_host_strings", 0) / 1e6, d.get("value_from_host_text", 0) / 1e6, (d.get("value_from_registered_text") or 0) / 1e6))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for v in "$@"; do
    case $v in
        default) run default ;;
        nosdma) run nosdma HSA_ENABLE_SDMA=0 ;;
        slots0) run slots0 MTG_COPY_SLOTS=0 ;;
        slots6) run slots6 MTG_COPY_SLOTS=6 ;;
        *) run "$v" $v ;;
    esac
done 2>&1 | tee $out/summary.txt
