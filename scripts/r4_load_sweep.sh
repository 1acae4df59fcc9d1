#!/bin/bash
# round 4: `Graph::load` of the config-4 container in FRESH processes, by number of uploader threads
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mindthegap_amd as mtg
from mindthegap_amd.synth import SynthSet
S = SynthSet(nseq=600000, n_sites=1000, seed=1, k=31)
dev = torch.device("cuda", 0)
w = torch.from_numpy(S.words.view(np.int64)).to(dev); wo = torch.from_numpy(S.word_off.view(np.int64)).to(dev); ln = torch.from_numpy(S.lens.view(np.int32)).to(dev)
idx = mtg.Index.from_packed_device(w.data_ptr(), wo.data_ptr(), ln.data_ptr(), S.nseq, S.total_kmers_upper_bound, 31, 3, 0)
t0 = time.time(); idx.save("/dev/shm/h.mtgidx"); print("saved %.2f GB in %.2f s" % (os.path.getsize("/dev/shm/h.mtgidx") / 1e9, time.time() - t0), flush=True)
PY
for th in 1 2 4 8 16; do MTG_LOAD_THREADS=$th MTG_DEBUG_TIMERS=1 python3 scripts/r4_load_cold.py /dev/shm/h.mtgidx 2>&1 | grep -E "cold load|again|alloc\]"; done
rm -f /dev/shm/h.mtgidx
