"""ctypes binding of oracle/_build/libmtg_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "libmtg_oracle.so")


class Params(C.Structure):
    _fields_ = [("max_nodes", C.c_int), ("max_depth", C.c_int), ("nb_mis_allowed", C.c_int), ("overlap", C.c_int),
                ("fwd_only", C.c_int), ("filter", C.c_int), ("extend", C.c_int), ("nb_cores", C.c_int),
                ("end_rule_nonbranching", C.c_int), ("seed_stride", C.c_int)]


def build():
    srcs = [os.path.join(ODIR, f) for f in ("mtg_oracle.cpp", "mtg_oracle.h", "mtg_oracle_main.cpp", "Makefile")]
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["make", "-C", ODIR], stdout=subprocess.DEVNULL)
    return SO


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    P = C.POINTER
    lib.mtgo_default_params.argtypes = [P(Params)]
    lib.mtgo_index_from_files.restype = C.c_void_p
    lib.mtgo_index_from_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.mtgo_index_from_kmers.restype = C.c_void_p
    lib.mtgo_index_from_kmers.argtypes = [P(C.c_uint64), P(C.c_uint32), C.c_size_t, C.c_int]
    lib.mtgo_index_from_sequences.restype = C.c_void_p
    lib.mtgo_index_from_sequences.argtypes = [P(C.c_char_p), C.c_size_t, C.c_int, C.c_uint32, C.c_uint32]
    lib.mtgo_index_free.argtypes = [C.c_void_p]
    lib.mtgo_index_k.argtypes = [C.c_void_p]
    lib.mtgo_index_size.restype = C.c_size_t
    lib.mtgo_index_size.argtypes = [C.c_void_p]
    lib.mtgo_index_abundance_min.argtypes = [C.c_void_p]
    lib.mtgo_index_auto_cutoff.argtypes = [C.c_void_p]
    lib.mtgo_index_export.restype = C.c_size_t
    lib.mtgo_index_export.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint32), C.c_size_t]
    lib.mtgo_index_stats.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]
    lib.mtgo_contains_batch.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint8)]
    lib.mtgo_abundance_batch.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint32)]
    lib.mtgo_stage_a.restype = C.c_void_p
    lib.mtgo_stage_a.argtypes = [C.c_void_p, P(Params), C.c_char_p, C.c_char_p, P(C.c_uint64)]
    lib.mtgo_fill_files.argtypes = [C.c_void_p, P(Params), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, P(C.c_uint64), P(C.c_double)]
    lib.mtgo_free.argtypes = [C.c_void_p]
    lib.mtgo_needleman_wunsch.restype = C.c_float
    lib.mtgo_needleman_wunsch.argtypes = [C.c_char_p, C.c_char_p]
    _lib = lib
    return lib


def default_params(**kw):
    p = Params()
    load().mtgo_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Index:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index construction failed")
        self.h = C.c_void_p(handle)
        self.lib = load()

    @classmethod
    def from_files(cls, paths, k=31, abundance_min=-1, abundance_max=0):
        return cls(load().mtgo_index_from_files(",".join(paths).encode(), k, abundance_min, abundance_max))

    @classmethod
    def from_sequences(cls, seqs, k=31, abund_lo=1, abund_span=40):
        arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
        return cls(load().mtgo_index_from_sequences(arr, len(seqs), k, abund_lo, abund_span))

    @classmethod
    def from_kmers(cls, kmers, counts, k=31):
        import numpy as np
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        return cls(load().mtgo_index_from_kmers(kmers.ctypes.data_as(C.POINTER(C.c_uint64)), counts.ctypes.data_as(C.POINTER(C.c_uint32)), len(kmers), k))

    @property
    def k(self):
        return self.lib.mtgo_index_k(self.h)

    def __len__(self):
        return self.lib.mtgo_index_size(self.h)

    def export(self):
        import numpy as np
        n = len(self)
        km = np.zeros(n, dtype=np.uint64)
        ct = np.zeros(n, dtype=np.uint32)
        self.lib.mtgo_index_export(self.h, km.ctypes.data_as(C.POINTER(C.c_uint64)), ct.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return km, ct

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.mtgo_index_stats(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def contains(self, kmers):
        import numpy as np
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.uint8)
        self.lib.mtgo_contains_batch(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def abundance(self, kmers):
        import numpy as np
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.uint32)
        self.lib.mtgo_abundance_batch(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def stage_a(self, source, target, params=None):
        params = params or default_params()
        probes = C.c_uint64()
        ptr = self.lib.mtgo_stage_a(self.h, C.byref(params), source.encode(), target.encode(), C.byref(probes))
        s = C.string_at(ptr).decode()
        self.lib.mtgo_free(ptr)
        return (s.split("\n") if s else []), probes.value

    def fill_files(self, mode, input_path, out_prefix, params=None, sample="oracle"):
        params = params or default_params()
        stats = (C.c_uint64 * 8)()
        secs = C.c_double()
        rc = self.lib.mtgo_fill_files(self.h, C.byref(params), 0 if mode == "bkpt" else 1, input_path.encode(), out_prefix.encode(), sample.encode(), stats, C.byref(secs))
        if rc != 0:
            raise RuntimeError("oracle fill failed rc=%d" % rc)
        keys = ["records", "filled", "multiple", "probes", "abundance_lookups", "nb_contigs", "nb_used_contigs"]
        d = {k: int(stats[i]) for i, k in enumerate(keys)}
        d["seconds"] = secs.value
        return d

    def close(self):
        if self.h:
            self.lib.mtgo_index_free(self.h)
            self.h = None
