"""Random-line read ceiling of the MI355X for the simple-path walk's access pattern (dependent chains, one line per step)."""
import sys
sys.path.insert(0, '.')
import mindthegap_amd as m
m.load_library()
for tb in (1 << 28, 1 << 34, 100 << 30):
    for line in (16, 32, 64, 128):
        for chains in (100000, 800000):
            ms, g = m.random_line_ceiling(tb, chains, 256, line)
            print("table %6.1f GB line %3d B chains %7d : %8.3f ms %8.1f GB/s %7.2f Glines/s" % (tb / 2**30, line, chains, ms, g, g / line), flush=True)
