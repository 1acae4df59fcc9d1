#!/usr/bin/env python3
"""the container load in a FRESH process (what `MindTheGap fill -graph` pays): usage r4_load_cold.py <container>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindthegap_amd as mtg
t0 = time.time(); g = mtg.Index.load(sys.argv[1]); t = time.time() - t0
prof = g.build_profile()
print("cold load (MTG_LOAD_THREADS=%s): %.2f s wall; " % (os.environ.get("MTG_LOAD_THREADS", "default"), t) + ", ".join("%s %.0f ms" % (ph["name"], ph["ms"]) for ph in prof["phases"]), flush=True)
t0 = time.time(); g.close(); g = mtg.Index.load(sys.argv[1]); t = time.time() - t0
prof = g.build_profile()
print("  again, same process: %.2f s wall; " % t + ", ".join("%s %.0f ms" % (ph["name"], ph["ms"]) for ph in prof["phases"]), flush=True)
