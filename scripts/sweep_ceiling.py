import sys; sys.path.insert(0,'.')
import mindthegap_amd as m
m.load_library()
for tb in (1<<28, 1<<32, 1<<34, 1<<36, 140<<30):
    for chains in (100000, 400000, 1600000):
        ms,g = m.random_line_ceiling(tb, chains, 256)
        print("table %6.1f GB chains %8d : %8.3f ms %8.1f GB/s  %.2f us/step" % (tb/2**30, chains, ms, g, ms*1e3/256), flush=True)
