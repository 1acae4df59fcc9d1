import sys
sys.path.insert(0, '.')
import mindthegap_amd as m
m.load_library()
for chains in (64, 1000, 10000, 30000, 60000, 100000, 200000):
    ms, g = m.random_line_ceiling(100 << 30, chains, 1024, 16)
    print("chains %7d : %8.3f ms  %.3f us/step  %.2f Greads/s" % (chains, ms, ms * 1e3 / 1024, g / 16), flush=True)
