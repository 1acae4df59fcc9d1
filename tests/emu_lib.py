"""ctypes binding of the TEST-ONLY host emulation of the device code (tests/emu/emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "emu.cpp")
SO = os.path.join(ROOT, "tests", "emu", "libmtg_emu.so")
HDRS = [os.path.join(ROOT, "mindthegap_amd", "csrc", h) for h in ("mtg_dev.h", "mtg_build.h", "mtg_traverse.h", "mtg_bubble.h", "mtg_post.h", "mtg_paths.h", "mtg_emit.h", "mtg_copy.h", "mtg_marshal.h", "mtg_format.h", "mtg_tuning.h", "mtg_hostutil.h")] + [os.path.join(ROOT, "tests", "emu", h) for h in ("emu_us.h", "emu_walk.h")]

# MTG_EMU_SANITIZE=1: build the emulation libraries with AddressSanitizer + UBSan (run pytest with LD_PRELOAD=$(gcc -print-file-name=libasan.so))
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-O1"] if os.environ.get("MTG_EMU_SANITIZE") else []
if SAN:
    SO = SO.replace(".so", "_san.so")

_lib = None
_lanes_libs = {}


def load(lanes=0, tsan=False):
    """lanes = N > 0: the build whose group form of the bubble code runs on N lanes in lock step (one host thread per lane, rendezvous at
    the group's collectives: mtg_bubble.h, MTG_EMU_LANES); tsan: that build under ThreadSanitizer (a separate process must load it)"""
    global _lib
    if lanes == 0 and _lib is not None:
        return _lib
    if lanes and (lanes, tsan) in _lanes_libs:
        return _lanes_libs[(lanes, tsan)]
    so = SO if not lanes else SO.replace(".so", "_lanes%d%s.so" % (lanes, "_tsan" if tsan else ""))
    extra = [] if not lanes else ["-DMTG_EMU_LANES=%d" % lanes, "-pthread"] + (["-fsanitize=thread", "-O1"] if tsan else [])
    deps = [SRC] + HDRS
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        tmp = so + ".%d.tmp" % os.getpid()  # several pytest workers may build at once: each writes its own file, the rename is atomic
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall"] + SAN + extra + ["-o", tmp, SRC])
        os.replace(tmp, so)
    lib = C.CDLL(so)
    P = C.POINTER
    lib.emu_index_create.restype = C.c_void_p
    lib.emu_index_create.argtypes = [P(C.c_uint64), P(C.c_uint32), C.c_size_t, C.c_int, C.c_double]
    lib.emu_index_free.argtypes = [C.c_void_p]
    lib.emu_query.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint32), P(C.c_uint8), P(C.c_uint8)]
    lib.emu_stage_a.restype = C.c_void_p
    lib.emu_stage_a.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    lib.emu_free.argtypes = [C.c_void_p]
    lib.emu_nw_matches.argtypes = [C.c_char_p, C.c_char_p]
    lib.emu_coop_counts.argtypes = [P(C.c_ulong)]
    lib.emu_walk_counts.argtypes = [P(C.c_ulong)]
    if lanes:
        _lanes_libs[(lanes, tsan)] = lib
    else:
        _lib = lib
    return lib


class EmuIndex:
    def __init__(self, kmers, counts, k, load_factor=0.6, lanes=0):
        self.lib = load(lanes)
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        self.k = k
        self.h = C.c_void_p(self.lib.emu_index_create(kmers.ctypes.data_as(C.POINTER(C.c_uint64)), counts.ctypes.data_as(C.POINTER(C.c_uint32)), len(kmers), k, load_factor))

    def query(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        n = len(kmers)
        ab = np.zeros(n, dtype=np.uint32)
        su = np.zeros(n, dtype=np.uint8)
        pr = np.zeros(n, dtype=np.uint8)
        self.lib.emu_query(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), n, ab.ctypes.data_as(C.POINTER(C.c_uint32)),
                           su.ctypes.data_as(C.POINTER(C.c_uint8)), pr.ctypes.data_as(C.POINTER(C.c_uint8)))
        return ab, su, pr

    def stage_a(self, source, target, max_nodes=100, max_depth=10000, end_rule=0, tier=-1):
        st, ln, tu = C.c_uint32(), C.c_uint32(), C.c_uint32()
        ptr = self.lib.emu_stage_a(self.h, max_nodes, max_depth, end_rule, source.encode(), target.encode(), tier, C.byref(st), C.byref(ln), C.byref(tu))
        s = C.string_at(ptr).decode()
        self.lib.emu_free(ptr)
        return (s.split("\n") if s else []), st.value, ln.value, tu.value

    def close(self):
        if self.h:
            self.lib.emu_index_free(self.h)
            self.h = None


# ---- the product's host code + CLI linked against the emulated device backend (tests/emu/emu_backend.cpp) ----
FULL_SO = os.path.join(ROOT, "tests", "emu", "libmtgfill_emu_san.so" if SAN else "libmtgfill_emu.so")


def build_full():
    """(re)build tests/emu/libmtgfill_emu.so when a source is newer"""
    csrc = os.path.join(ROOT, "mindthegap_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "emu", "emu_backend.cpp"), os.path.join(csrc, "mtg_host.cpp"), os.path.join(csrc, "mtg_cli.cpp")]
    deps = srcs + HDRS + [os.path.join(csrc, "mtg_internal.h"), os.path.join(ROOT, "include", "mtg_fill.h")]
    if not os.path.exists(FULL_SO) or any(os.path.getmtime(d) > os.path.getmtime(FULL_SO) for d in deps):
        tmp = FULL_SO + ".%d.tmp" % os.getpid()
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread"] + SAN + ["-o", tmp] + srcs + ["-lz"])
        os.replace(tmp, FULL_SO)
    return FULL_SO


def product_on_emulator():
    """Returns the mindthegap_amd package with its ctypes handle pointed at the TEST-ONLY emulation build, so the CPU suite can
    drive the real host code / CLI.  The product itself has no such switch."""
    build_full()
    import mindthegap_amd
    from mindthegap_amd import lib as L
    L._lib = L._bind(C.CDLL(FULL_SO))
    return mindthegap_amd


def gen_counts():
    """(multi-contig gaps with several reached targets the device function finished, multi-target gaps it left to the host) of the product on the emulator"""
    a = (C.c_ulong * 2)()
    lib = C.CDLL(build_full())
    lib.emu_gen_counts.argtypes = [C.POINTER(C.c_ulong)]
    lib.emu_gen_counts(a)
    return int(a[0]), int(a[1])


def walk_counts(full=False):
    """(stored unitigs whose two walkers met in the middle, stored unitigs whose owner walked the whole chain) over the lean builds so far, of
    tests/emu/libemu or (full) of the product on the emulator"""
    a = (C.c_ulong * 2)()
    if full:
        lib = C.CDLL(build_full())
        lib.emu_walk_counts.argtypes = [C.POINTER(C.c_ulong)]
        lib.emu_walk_counts(a)
    else:
        load().emu_walk_counts(a)
    return int(a[0]), int(a[1])


def coop_counts(lanes=0):
    """(answered with a consensus, rejected, passed on as too big, tips answered by the fast path) by the group form of the bubble code in the emulation library so far"""
    a = (C.c_ulong * 9)()
    load(lanes).emu_coop_counts(a)
    return tuple(int(x) for x in a)  # the fourth: tips answered by the walking lane's fast path (tip_fast); the fifth: unequal-length bubbles (indel_bulk); the sixth: nodes whose one successor has a second predecessor (merge_fast); then the refusals the cross-checking build has verified: one successor / a marked successor / the SNP pattern onto a marked node
