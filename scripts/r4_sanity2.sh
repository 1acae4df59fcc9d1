#!/bin/bash
# second sweep: the tool's drivers and the host side under small batches, many workers, one pool thread, one copy slot, and combinations with
# several launches per batch; the CLI / synthetic / short-fill / reverse / diploid / allelic / replica GPU tests each time
for v in "MTG_CLI_BATCH=7" "MTG_CLI_BATCH=1000 MTG_CLI_IN_FLIGHT=6" "MTG_POOL_THREADS=1" "MTG_COPY_SLOTS=1" "MTG_MAX_CHUNK=37 MTG_NO_LEAN=1" "MTG_MAX_CHUNK=5 MTG_CLI_BATCH=50" "MTG_TUNING=FINISH_G=16,ROUNDS=3,MAX_CHUNK=64" "MTG_CLI_NO_MMAP=1 MTG_CLI_BATCH=3" "MTG_HOST_FORMAT=1 MTG_MAX_CHUNK=37"; do
    echo "== tests under: $v"
    env $v timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "cli or synthetic_sites or short_fills or reverse_attempt or diploid_bubbles or allelic or replica or several_launches" 2>&1 | grep -E "passed|failed|rror" | tail -2
done
