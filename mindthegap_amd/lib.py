"""ctypes binding of libmtgfill.so (include/mtg_fill.h)."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
SO = os.path.join(LIBDIR, "libmtgfill.so")
if os.environ.get("MTG_LIBRARY_PATH"):  # diagnostics: another build of the same HIP library (e.g. with device-side timing compiled in)
    SO = os.environ["MTG_LIBRARY_PATH"]

ERRORS = {1: "MTG_ERR_NO_DEVICE", 2: "MTG_ERR_ARG", 3: "MTG_ERR_IO", 4: "MTG_ERR_NOMEM", 5: "MTG_ERR_OVERFLOW", 6: "MTG_ERR_FORMAT"}



def cpu_budget():
    """CPUs this process may keep busy: hardware threads, affinity mask and the CFS bandwidth limit of its cgroup (what the library's
    worker pool sizes itself by, Pool::cpu_budget in csrc/mtg_internal.h)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota, period = -1, 0
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, period = int(q), int(per)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota > 0 and period > 0:
        n = min(n, max(1, quota // period))
    return max(1, n)

class MtgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code


def library_path():
    return SO


def build_library(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "mtg_fill.h")]
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs if os.path.isfile(s))
    if stale:
        subprocess.check_call(["make", "-C", CSRC] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return SO


class Params(C.Structure):
    _fields_ = [("max_nodes", C.c_int), ("max_depth", C.c_int), ("nb_mis_allowed", C.c_int), ("end_rule_nonbranching", C.c_int),
                ("nb_host_threads", C.c_int)]


class IndexInfo(C.Structure):
    _fields_ = [("k", C.c_int), ("abundance_min", C.c_int), ("abundance_auto", C.c_int), ("nb_solid_kmers", C.c_uint64),
                ("nb_branching", C.c_uint64), ("device_bytes", C.c_uint64), ("adj_buckets", C.c_uint64), ("abnd_buckets", C.c_uint64), ("adj_bucket_bytes", C.c_uint32), ("abnd_bucket_bytes", C.c_uint32), ("bloom_blocks", C.c_uint64), ("bloom_minimizer", C.c_uint32),
                ("nb_unitigs", C.c_uint64), ("unitig_bytes", C.c_uint64), ("nb_saturated", C.c_uint64), ("sparse", C.c_uint32), ("pad_", C.c_uint32),
                ("nb_kmers_outside_unitigs", C.c_uint64)]


class CGap(C.Structure):
    _fields_ = [("source", C.c_char_p), ("target", C.c_char_p), ("n_targets", C.c_int), ("target_seqs", C.POINTER(C.c_char_p)),
                ("target_names", C.POINTER(C.c_char_p)), ("target_is_rc", C.POINTER(C.c_uint8)), ("is_anchor_repeated", C.c_int),
                ("reverse", C.c_int)]


class CFilled(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("nb_errors_in_anchor", C.c_int), ("target_index", C.c_int), ("avg_coverage", C.c_float),
                ("median_coverage", C.c_float), ("qual", C.c_int), ("solution_count", C.c_int), ("solution_rank", C.c_int)]


class CGapResult(C.Structure):
    _fields_ = [("nb_nodes", C.c_int), ("total_nt", C.c_int), ("nb_terminal", C.c_int), ("has_solution_counts", C.c_int),
                ("nb_total_filled", C.c_int), ("nb_reported", C.c_int), ("n_filled", C.c_int), ("filled", C.POINTER(CFilled)),
                ("extension", C.c_char_p)]


class ScanStats(C.Structure):
    _fields_ = [("n_kmers", C.c_uint64), ("bloom_positive", C.c_uint64), ("confirmed", C.c_uint64), ("blocks_staged", C.c_uint64), ("kernel_ms", C.c_double)]


class BatchStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("post_kernel_ms", C.c_double), ("total_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("host_ms", C.c_double), ("marshal_ms", C.c_double), ("result_ms", C.c_double),
                ("index_lines", C.c_uint64), ("n_launches", C.c_uint64), ("n_retried_gaps", C.c_uint64), ("contig_nt", C.c_uint64),
                ("store_runs", C.c_uint64), ("run_nt", C.c_uint64), ("post_lines", C.c_uint64), ("contig_words", C.c_uint64), ("coverage_kmers", C.c_uint64),
                ("dense_words", C.c_uint64), ("emit_kernel_ms", C.c_double), ("seq_bytes", C.c_uint64), ("copy_kernel_ms", C.c_double), ("copy_words", C.c_uint64),
                ("copy_cmds", C.c_uint64), ("coverage_direct_kmers", C.c_uint64), ("finish_kernel_ms", C.c_double), ("n_parked_gaps", C.c_uint64), ("n_rounds", C.c_uint64), ("n_lean_gaps", C.c_uint64),
                ("lean_kernel_ms", C.c_double), ("copy_words_executed", C.c_uint64), ("copy_cmds_executed", C.c_uint64), ("post_scanned_words", C.c_uint64), ("device_span_ms", C.c_double), ("n_general_device", C.c_uint64), ("n_general_host", C.c_uint64), ("n_light_walks", C.c_uint64),
                ("n_branching_gaps", C.c_uint64)]


class BuildPhase(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_double), ("bytes", C.c_uint64), ("units", C.c_uint64)]


class WireHeader(C.Structure):
    _fields_ = [("magic", C.c_uint64), ("tag", C.c_uint64), ("n_gaps", C.c_uint64), ("n_filled", C.c_uint64), ("seq_bytes", C.c_uint64), ("ext_bytes", C.c_uint64),
                ("total_bytes", C.c_uint64), ("checksum", C.c_uint64)]


class CSite(C.Structure):
    _fields_ = [("name", C.c_char_p), ("name_r", C.c_char_p), ("source", C.c_char_p), ("target", C.c_char_p)]


class CText(C.Structure):
    _fields_ = [("fasta", C.c_void_p), ("info", C.c_void_p), ("vcf", C.c_void_p), ("ext", C.c_void_p),
                ("fasta_bytes", C.c_uint64), ("info_bytes", C.c_uint64), ("vcf_bytes", C.c_uint64), ("ext_bytes", C.c_uint64)]


WIRE_HEADER_BYTES = C.sizeof(WireHeader)

_lib = None


def load_library():
    """Load libmtgfill.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise MtgError(1, "libmtgfill.so is not built (%s); run __graft_entry__.build() or make -C mindthegap_amd/csrc" % SO)
    _lib = _bind(C.CDLL(SO))
    return _lib


def _bind(lib):
    P = C.POINTER
    lib.mtg_last_error.restype = C.c_char_p
    lib.mtg_index_create_from_reads.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, P(C.c_void_p)]
    lib.mtg_index_create_from_kmers.argtypes = [P(C.c_uint64), P(C.c_uint32), C.c_size_t, C.c_int, P(C.c_void_p)]
    lib.mtg_index_create_from_packed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, P(C.c_void_p)]
    lib.mtg_fill_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_void_p)]
    lib.mtg_fill_text_serial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint64, P(C.c_uint64), P(C.c_void_p)]
    lib.mtg_index_load.argtypes = [C.c_char_p, P(C.c_void_p)]
    lib.mtg_index_save.argtypes = [C.c_void_p, C.c_char_p]
    lib.mtg_index_replicate.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p)]
    lib.mtg_index_get_info.argtypes = [C.c_void_p, P(IndexInfo)]
    lib.mtg_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.mtg_host_unregister.argtypes = [C.c_void_p]
    lib.mtg_tuning_count.restype = C.c_size_t
    lib.mtg_tuning_count.argtypes = []
    lib.mtg_tuning_describe.argtypes = [C.c_size_t, P(C.c_char_p), P(C.c_char_p), P(C.c_char_p), P(C.c_char_p)]
    lib.mtg_tuning_get.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.mtg_tuning_set.argtypes = [C.c_char_p, C.c_char_p]
    lib.mtg_index_build_profile.argtypes = [C.c_void_p, P(BuildPhase), C.c_size_t, P(C.c_size_t), P(C.c_uint64), P(C.c_double)]
    lib.mtg_index_free.argtypes = [C.c_void_p]
    lib.mtg_index_free.restype = None
    lib.mtg_index_contains.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint8)]
    lib.mtg_index_abundance.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint32)]
    lib.mtg_index_neighbors.argtypes = [C.c_void_p, P(C.c_uint64), C.c_size_t, P(C.c_uint8), P(C.c_uint8)]
    lib.mtg_index_scan_sequences.argtypes = [C.c_void_p, P(C.c_char_p), C.c_size_t, C.c_int, P(P(C.c_uint8)), P(ScanStats)]
    lib.mtg_index_scan_packed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, P(ScanStats)]
    lib.mtg_default_params.argtypes = [P(Params)]
    lib.mtg_default_params.restype = None
    lib.mtg_fill_batch.argtypes = [C.c_void_p, P(Params), P(CGap), C.c_size_t, P(C.c_void_p)]
    lib.mtg_fill_batch_serial.argtypes = [C.c_void_p, P(Params), P(CGap), C.c_size_t, C.c_void_p, C.c_uint64, P(C.c_uint64), P(C.c_void_p)]
    lib.mtg_batch_prepare.argtypes = [C.c_void_p, P(Params), P(CGap), C.c_size_t, P(C.c_void_p)]
    lib.mtg_batch_free.argtypes = [C.c_void_p]
    lib.mtg_batch_free.restype = None
    lib.mtg_fill_prepared.argtypes = [C.c_void_p, P(Params), C.c_void_p, P(C.c_void_p)]
    lib.mtg_fill_prepared_serial.argtypes = [C.c_void_p, P(Params), C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64), P(C.c_void_p)]
    lib.mtg_fill_prepared_serial_device.argtypes = [C.c_void_p, P(Params), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, P(C.c_uint64), P(C.c_void_p)]
    lib.mtg_results_get.argtypes = [C.c_void_p, C.c_size_t]
    lib.mtg_results_get.restype = P(CGapResult)
    lib.mtg_results_free.argtypes = [C.c_void_p]
    lib.mtg_results_free.restype = None
    lib.mtg_results_summary.argtypes = [C.c_void_p, P(C.c_uint32), P(C.c_uint64), P(C.c_uint64)]
    lib.mtg_results_copy_seqs.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    lib.mtg_stage_a_batch.argtypes = [C.c_void_p, P(Params), P(C.c_char_p), P(C.c_char_p), C.c_size_t, P(C.c_void_p)]
    lib.mtg_contigs_count.argtypes = [C.c_void_p, C.c_size_t]
    lib.mtg_contigs_count.restype = C.c_size_t
    lib.mtg_contigs_get.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    lib.mtg_contigs_get.restype = C.c_char_p
    lib.mtg_contigs_free.argtypes = [C.c_void_p]
    lib.mtg_contigs_free.restype = None
    lib.mtg_fill_prepared_wire_device.argtypes = [C.c_void_p, P(Params), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, P(C.c_uint64), P(C.c_void_p)]
    lib.mtg_results_wire_size.argtypes = [C.c_void_p, P(C.c_uint64)]
    lib.mtg_results_to_wire.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, P(C.c_uint64)]
    lib.mtg_results_from_wire.argtypes = [C.c_void_p, C.c_uint64, P(C.c_void_p), P(C.c_uint64)]
    lib.mtg_format_bkpt.argtypes = [P(CSite), C.c_size_t, C.c_void_p, C.c_void_p, P(C.c_int64), C.c_int, C.c_int, P(CText)]
    lib.mtg_format_vcf_header.argtypes = [C.c_char_p, C.c_char_p, P(CText)]
    lib.mtg_text_free.argtypes = [P(CText)]
    lib.mtg_text_free.restype = None
    lib.mtg_last_batch_stats.argtypes = [P(BatchStats)]
    lib.mtg_fill_main.argtypes = [C.c_int, P(C.c_char_p)]
    lib.mtg_fill_main_on_index.argtypes = [C.c_void_p, C.c_int, P(C.c_char_p)]
    lib.mtg_nw_matches.argtypes = [P(C.c_char_p), P(C.c_char_p), C.c_size_t, P(C.c_uint32)]
    lib.mtg_bench_random_lines.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, P(C.c_double), P(C.c_double)]
    return lib


def _check(rc):
    if rc != 0:
        raise MtgError(rc, load_library().mtg_last_error().decode(errors="replace"))


def device_count():
    return load_library().mtg_device_count()


class FillParams:
    def __init__(self, max_nodes=100, max_depth=10000, nb_mis_allowed=2, end_rule_nonbranching=0, nb_host_threads=0):
        self.c = Params(max_nodes, max_depth, nb_mis_allowed, end_rule_nonbranching, nb_host_threads)


class Gap:
    """One Filler::gapFillFromSource call (src/Filler.hpp:188-189)."""

    def __init__(self, source, target, targets, is_anchor_repeated=False, reverse=False):
        self.source, self.target = source, target
        self.targets = list(targets)  # [(kmer, name, is_rc)] in dictionary iteration order
        self.is_anchor_repeated, self.reverse = is_anchor_repeated, reverse


class CTextGaps(C.Structure):
    _fields_ = [("text", C.c_void_p), ("text_bytes", C.c_uint64), ("n", C.c_uint64), ("source_off", C.c_void_p), ("source_len", C.c_void_p), ("pattern_off", C.c_void_p),
                ("pattern_len", C.c_void_p), ("dict_first", C.c_void_p), ("dict_seq_off", C.c_void_p), ("dict_seq_len", C.c_void_p), ("dict_name_off", C.c_void_p),
                ("dict_name_len", C.c_void_p), ("dict_is_rc", C.c_void_p), ("gap_flags", C.c_void_p)]


class TextGaps:
    """mtg_text_gaps: a batch whose strings are (offset, length) pairs into one block of text (what mtg_fill_text encodes on the device).
    Built here from a list of Gap by laying their strings out one after the other; a reader of a breakpoint file would point into its buffer."""

    def __init__(self, gaps):
        parts, off = [], 0

        def put(s):
            nonlocal off
            b = s.encode() if isinstance(s, str) else s
            parts.append(b)
            o = off
            off += len(b)
            return o, len(b)

        n = len(gaps)
        so, sl, po, pl = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
        first = np.zeros(n + 1, np.uint32)
        flags = np.zeros(max(n, 1), np.uint8)
        do, dl, no, nl, rc = [], [], [], [], []
        for i, g in enumerate(gaps):
            so[i], sl[i] = put(g.source)
            po[i], pl[i] = put(g.target)
            for (kmer, name, is_rc) in g.targets:
                # a one-entry dictionary whose sequence is the pattern itself (breakpoint mode) points at the pattern's bytes, as the tool's reader of a
                # breakpoint file does (mtg_cli.cpp): the right k-mer stands once in the block
                a, b = (int(po[i]), int(pl[i])) if kmer == g.target else put(kmer)
                do.append(a)
                dl.append(b)
                a, b = put(name)
                no.append(a)
                nl.append(b)
                rc.append(1 if is_rc else 0)
            first[i + 1] = len(do)
            flags[i] = (1 if g.is_anchor_repeated else 0) | (2 if g.reverse else 0)
        self.n = n
        self.text = np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8).copy()
        self.arrays = dict(source_off=so, source_len=sl, pattern_off=po, pattern_len=pl, dict_first=first, dict_seq_off=np.array(do + [0], np.uint64), dict_seq_len=np.array(dl + [0], np.uint32),
                           dict_name_off=np.array(no + [0], np.uint64), dict_name_len=np.array(nl + [0], np.uint32), dict_is_rc=np.array(rc + [0], np.uint8), gap_flags=flags)
        self.c = CTextGaps(self.text.ctypes.data, len(self.text) - 1, n, *[self.arrays[k].ctypes.data for k in ("source_off", "source_len", "pattern_off", "pattern_len", "dict_first", "dict_seq_off",
                                                                                                            "dict_seq_len", "dict_name_off", "dict_name_len", "dict_is_rc", "gap_flags")])
        self.registered = False

    def register(self):
        """page-lock the block (mtg_host_register): mtg_fill_text then uploads it from here instead of copying it into its own page-locked block"""
        if not self.registered:
            _check(load_library().mtg_host_register(self.text.ctypes.data, len(self.text)))
            self.registered = True
        return self

    def unregister(self):
        if self.registered:
            _check(load_library().mtg_host_unregister(self.text.ctypes.data))
            self.registered = False

    def __del__(self):
        try:
            self.unregister()
        except Exception:
            pass


class Batch:
    """A batch of gaps marshalled once and kept in device memory (mtg_batch_prepare); fill it with Index.fill_prepared / fill_prepared_serial."""

    def __init__(self, handle, n, keep):
        self.h, self.n, self._keep = handle, n, keep  # _keep: the ctypes arrays the batch's strings live in

    def close(self):
        if self.h:
            load_library().mtg_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """Device-resident de Bruijn graph index (the gatb Graph of src/Filler.cpp:210,222)."""

    def __init__(self, handle):
        self.h = handle
        self.lib = load_library()
        self.last_seq_bytes = 0

    @classmethod
    def from_reads(cls, paths, k=31, abundance_min=-1, abundance_max=0):
        h = C.c_void_p()
        _check(load_library().mtg_index_create_from_reads(",".join(paths).encode(), k, abundance_min, abundance_max, C.byref(h)))
        return cls(h)

    @classmethod
    def from_kmers(cls, kmers, abundance, k=31):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        abundance = np.ascontiguousarray(abundance, dtype=np.uint32)
        h = C.c_void_p()
        _check(load_library().mtg_index_create_from_kmers(kmers.ctypes.data_as(C.POINTER(C.c_uint64)), abundance.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                          len(kmers), k, C.byref(h)))
        return cls(h)

    @classmethod
    def from_packed_device(cls, words_ptr, word_off_ptr, len_ptr, nseq, total_kmers_upper_bound, k=31, abund_lo=3, abund_span=40):
        """Device pointers (e.g. torch tensor .data_ptr()) of 2-bit packed sequences."""
        h = C.c_void_p()
        _check(load_library().mtg_index_create_from_packed_device(words_ptr, word_off_ptr, len_ptr, nseq, total_kmers_upper_bound, k, abund_lo, abund_span, C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        _check(load_library().mtg_index_load(path.encode(), C.byref(h)))
        return cls(h)

    def save(self, path):
        _check(self.lib.mtg_index_save(self.h, path.encode()))

    def info(self):
        i = IndexInfo()
        _check(self.lib.mtg_index_get_info(self.h, C.byref(i)))
        return {f[0]: getattr(i, f[0]) for f in IndexInfo._fields_}

    def build_profile(self):
        """how the index was constructed: {"phases": [{name, ms, bytes, units}], "peak_device_bytes", "total_ms"}"""
        n, peak, total = C.c_size_t(), C.c_uint64(), C.c_double()
        _check(self.lib.mtg_index_build_profile(self.h, None, 0, C.byref(n), C.byref(peak), C.byref(total)))
        arr = (BuildPhase * max(n.value, 1))()
        _check(self.lib.mtg_index_build_profile(self.h, arr, n.value, C.byref(n), C.byref(peak), C.byref(total)))
        return {"phases": [{"name": arr[i].name.decode(), "ms": arr[i].ms, "bytes": arr[i].bytes, "units": arr[i].units} for i in range(n.value)],
                "peak_device_bytes": peak.value, "total_ms": total.value}

    def contains(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.uint8)
        _check(self.lib.mtg_index_contains(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def abundance(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.uint32)
        _check(self.lib.mtg_index_abundance(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers), out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def neighbors(self, kmers):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        s = np.zeros(len(kmers), dtype=np.uint8)
        p = np.zeros(len(kmers), dtype=np.uint8)
        _check(self.lib.mtg_index_neighbors(self.h, kmers.ctypes.data_as(C.POINTER(C.c_uint64)), len(kmers), s.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            p.ctypes.data_as(C.POINTER(C.c_uint8))))
        return s, p

    def scan_sequences(self, seqs, exact=True):
        """membership (0/1) of the k-mer starting at every position of every sequence; returns (list of uint8 arrays, stats dict)"""
        k = self.info()["k"]
        n = len(seqs)
        arr = (C.c_char_p * n)(*[s.encode() for s in seqs])
        outs = [np.zeros(max(len(s) - k + 1, 0), dtype=np.uint8) for s in seqs]
        ptrs = (C.POINTER(C.c_uint8) * n)(*[o.ctypes.data_as(C.POINTER(C.c_uint8)) for o in outs])
        st = ScanStats()
        _check(self.lib.mtg_index_scan_sequences(self.h, arr, n, 1 if exact else 0, ptrs, C.byref(st)))
        return outs, {f[0]: getattr(st, f[0]) for f in ScanStats._fields_}

    def scan_packed_device(self, words_ptr, word_off_ptr, len_ptr, nseq, out_bits_ptr, exact=True):
        st = ScanStats()
        _check(self.lib.mtg_index_scan_packed_device(self.h, words_ptr, word_off_ptr, len_ptr, nseq, 1 if exact else 0, out_bits_ptr, C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in ScanStats._fields_}

    def stage_a(self, sources, targets, params=None):
        """Contigs of every gap (gatb IterativeExtensions::construct_linear_seqs, src/Filler.cpp:884)."""
        params = params or FillParams()
        n = len(sources)
        sa = (C.c_char_p * n)(*[s.encode() for s in sources])
        ta = (C.c_char_p * n)(*[t.encode() for t in targets])
        h = C.c_void_p()
        _check(self.lib.mtg_stage_a_batch(self.h, C.byref(params.c), sa, ta, n, C.byref(h)))
        out = []
        for g in range(n):
            out.append([self.lib.mtg_contigs_get(h, g, i).decode() for i in range(self.lib.mtg_contigs_count(h, g))])
        self.lib.mtg_contigs_free(h)
        return out

    @staticmethod
    def prepare_gaps(gaps):
        """marshal a list of Gap once; the returned object can be passed to fill_prepared repeatedly"""
        n = len(gaps)
        arr = (CGap * n)()
        keep = []
        for i, g in enumerate(gaps):
            m = len(g.targets)
            seqs = (C.c_char_p * max(m, 1))(*[t[0].encode() for t in g.targets])
            names = (C.c_char_p * max(m, 1))(*[t[1].encode() for t in g.targets])
            rcs = (C.c_uint8 * max(m, 1))(*[1 if t[2] else 0 for t in g.targets])
            keep.append((seqs, names, rcs))
            arr[i] = CGap(g.source.encode(), g.target.encode(), m, seqs, names, rcs, int(g.is_anchor_repeated), int(g.reverse))
        return (arr, n, keep)

    def prepare_batch(self, gaps, params=None):
        """marshal a list of Gap (or the result of prepare_gaps) into a device-resident Batch"""
        params = params or FillParams()
        arr, n, keep = gaps if isinstance(gaps, tuple) else self.prepare_gaps(gaps)
        h = C.c_void_p()
        _check(self.lib.mtg_batch_prepare(self.h, C.byref(params.c), arr, n, C.byref(h)))
        return Batch(h, n, (arr, keep))

    def fill_main(self, argv):
        """mtg_fill_main_on_index: `MindTheGap fill <argv>` with this resident index as the graph; returns the exit code"""
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        return self.lib.mtg_fill_main_on_index(self.h, len(argv), arr)

    def fill_prepared_wire_device(self, batch, tag, d_ptr, cap, params=None):
        """mtg_fill_prepared_wire_device: fills the prepared Batch; the whole batch in relocatable form (records AND sequences, tagged) is
        written by the result kernel into the device buffer at d_ptr (cap bytes).  Returns (results handle, n_filled per gap, payload bytes)."""
        params = params or FillParams()
        h = C.c_void_p()
        nb = C.c_uint64()
        _check(self.lib.mtg_fill_prepared_wire_device(self.h, C.byref(params.c), batch.h, int(tag), C.c_void_p(d_ptr), int(cap), C.byref(nb), C.byref(h)))
        nf = np.empty(batch.n, dtype=np.uint32)
        sb = C.c_uint64()
        _check(self.lib.mtg_results_summary(h, nf.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(sb), None))
        self.last_seq_bytes = int(sb.value)
        return h, nf, int(nb.value)

    def fill_prepared(self, prepared, params=None, want_seqs=True, out=None):
        """one mtg_fill_batch (or, for a Batch, mtg_fill_prepared) call; returns (results handle, n_filled per gap, uint8 array of the packed
        "seq\\n" bytes).  Free with free_results.
        out: optional uint8 array that receives the bytes (e.g. a page-locked buffer); self.last_seq_bytes = their number either way."""
        params = params or FillParams()
        h = C.c_void_p()
        t0 = time.perf_counter()
        if isinstance(prepared, Batch):
            n = prepared.n
            _check(self.lib.mtg_fill_prepared(self.h, C.byref(params.c), prepared.h, C.byref(h)))
        elif isinstance(prepared, TextGaps):
            n = prepared.n
            _check(self.lib.mtg_fill_text(self.h, C.byref(params.c), C.byref(prepared.c), C.byref(h)))
        else:
            arr, n, _ = prepared
            _check(self.lib.mtg_fill_batch(self.h, C.byref(params.c), arr, n, C.byref(h)))
        t1 = time.perf_counter()
        nf = np.empty(n, dtype=np.uint32)
        nb, ng = C.c_uint64(), C.c_uint64()
        _check(self.lib.mtg_results_summary(h, nf.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nb), C.byref(ng)))
        if os.environ.get("MTG_BENCH_DEBUG"):
            sys.stderr.write("  [py] mtg_fill_batch %.1f ms, summary %.1f ms\n" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
        self.last_seq_bytes = int(nb.value)
        if not want_seqs:
            return h, nf, None
        if out is not None:
            if out.size < nb.value:
                raise MtgError(2, "output buffer of %d bytes for %d bytes of sequences" % (out.size, nb.value))
            buf = out
        else:
            buf = np.empty(max(int(nb.value), 1), dtype=np.uint8)
        _check(self.lib.mtg_results_copy_seqs(h, buf.ctypes.data_as(C.c_char_p), nb.value))
        return h, nf, buf[: nb.value]

    def fill_prepared_serial(self, prepared, out, params=None):
        """mtg_fill_batch_serial: the batch's filled sequences end up in the uint8 array `out` (NUL-terminated, gap order), decoded there
        directly in the common case; returns (results handle, n_filled per gap, number of bytes).  `out` must stay untouched until
        free_results."""
        params = params or FillParams()
        h = C.c_void_p()
        nb = C.c_uint64()
        if isinstance(prepared, Batch):
            n = prepared.n
            _check(self.lib.mtg_fill_prepared_serial(self.h, C.byref(params.c), prepared.h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(nb), C.byref(h)))
        else:
            arr, n, _ = prepared
            _check(self.lib.mtg_fill_batch_serial(self.h, C.byref(params.c), arr, n, out.ctypes.data_as(C.c_void_p), out.size, C.byref(nb), C.byref(h)))
        nf = np.empty(n, dtype=np.uint32)
        _check(self.lib.mtg_results_summary(h, nf.ctypes.data_as(C.POINTER(C.c_uint32)), None, None))
        self.last_seq_bytes = int(nb.value)
        return h, nf, int(nb.value)

    def fill_prepared_serial_device(self, batch, dev_ptr, cap, params=None, host_out=None):
        """mtg_fill_prepared_serial_device: the filled sequences of a prepared Batch are produced in DEVICE memory, in the caller's buffer of `cap`
        bytes at address dev_ptr on the index's device (NUL-terminated, gap order); host_out (uint8 array of at least cap bytes, e.g. page-locked):
        they are copied there as well.  Returns (results handle, n_filled per gap, number of bytes)."""
        params = params or FillParams()
        h = C.c_void_p()
        nb = C.c_uint64()
        if host_out is not None and host_out.size < cap:
            raise MtgError(2, "host copy of %d bytes for a device buffer of %d" % (host_out.size, cap))
        hp = host_out.ctypes.data_as(C.c_void_p) if host_out is not None else None
        _check(self.lib.mtg_fill_prepared_serial_device(self.h, C.byref(params.c), batch.h, C.c_void_p(dev_ptr), cap, hp, C.byref(nb), C.byref(h)))
        nf = np.empty(batch.n, dtype=np.uint32)
        _check(self.lib.mtg_results_summary(h, nf.ctypes.data_as(C.POINTER(C.c_uint32)), None, None))
        self.last_seq_bytes = int(nb.value)
        return h, nf, int(nb.value)

    def free_results(self, h):
        self.lib.mtg_results_free(h)

    def fill_batch(self, gaps, params=None):
        """Filler::gapFillFromSource for every gap; returns one dict per gap."""
        params = params or FillParams()
        arr, n, _keep = self.prepare_gaps(gaps)
        h = C.c_void_p()
        _check(self.lib.mtg_fill_batch(self.h, C.byref(params.c), arr, n, C.byref(h)))
        res = []
        for i in range(n):
            r = self.lib.mtg_results_get(h, i).contents
            filled = []
            for j in range(r.n_filled):
                f = r.filled[j]
                filled.append(dict(seq=f.seq.decode(), nb_errors_in_anchor=f.nb_errors_in_anchor, target_index=f.target_index, avg_coverage=f.avg_coverage,
                                   median_coverage=f.median_coverage, qual=f.qual, solution_count=f.solution_count, solution_rank=f.solution_rank))
            res.append(dict(nb_nodes=r.nb_nodes, total_nt=r.total_nt, nb_terminal=r.nb_terminal, has_solution_counts=bool(r.has_solution_counts),
                            nb_total_filled=r.nb_total_filled, nb_reported=r.nb_reported, filled=filled, extension=r.extension.decode()))
        self.lib.mtg_results_free(h)
        return res

    def close(self):
        if self.h:
            self.lib.mtg_index_free(self.h)
            self.h = None


def results_to_wire(h, tag, out=None):
    """mtg_results_to_wire: the result set h (records AND sequences) in relocatable form, tagged; written into the uint8 array `out`
    (e.g. the page-locked buffer of a gather) or a new array.  Returns the payload (a view of `out`)."""
    lib = load_library()
    need = C.c_uint64()
    _check(lib.mtg_results_wire_size(h, C.byref(need)))
    if out is None:
        out = np.empty(int(need.value), dtype=np.uint8)
    if out.size < need.value:
        raise MtgError(2, "wire buffer of %d bytes for a payload of %d" % (out.size, need.value))
    got = C.c_uint64()
    _check(lib.mtg_results_to_wire(h, int(tag), out.ctypes.data_as(C.c_void_p), out.size, C.byref(got)))
    return out[: int(got.value)]


def wire_header(payload):
    """the mtg_wire_header of a payload (uint8 array) as a dict"""
    h = WireHeader.from_buffer_copy(np.ascontiguousarray(payload[:WIRE_HEADER_BYTES]).tobytes())
    return {f[0]: int(getattr(h, f[0])) for f in WireHeader._fields_}


class WireResults:
    """mtg_results_from_wire: a result set rebuilt from a payload (validated: sizes, offsets, checksum); the payload is kept alive."""

    def __init__(self, payload):
        self.lib = load_library()
        self.payload = np.ascontiguousarray(payload, dtype=np.uint8)
        self.h = C.c_void_p()
        tag = C.c_uint64()
        _check(self.lib.mtg_results_from_wire(self.payload.ctypes.data_as(C.c_void_p), self.payload.size, C.byref(self.h), C.byref(tag)))
        self.tag = int(tag.value)
        self.n = wire_header(self.payload)["n_gaps"]

    def close(self):
        if self.h:
            self.lib.mtg_results_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def format_bkpt(sites, fwd, rev=None, rev_index=None, filter=False, extend=False):
    """mtg_format_bkpt: the text a run of breakpoint sites adds to the tool's output files.  sites: [(name, name_r, source, target)];
    fwd / rev: result handles (c_void_p or WireResults); rev_index: per site, index into rev or -1.  Returns dict of bytes."""
    lib = load_library()
    n = len(sites)
    arr = (CSite * max(n, 1))()
    keep = []
    for i, (a, b, c, d) in enumerate(sites):
        vals = [x.encode() if isinstance(x, str) else x for x in (a, b, c, d)]
        keep.append(vals)
        arr[i].name, arr[i].name_r, arr[i].source, arr[i].target = vals
    hf = fwd.h if isinstance(fwd, WireResults) else fwd
    hr = rev.h if isinstance(rev, WireResults) else rev
    ri = None
    if rev_index is not None:
        ri = np.ascontiguousarray(rev_index, dtype=np.int64)
    t = CText()
    _check(lib.mtg_format_bkpt(arr, n, hf, hr, ri.ctypes.data_as(C.POINTER(C.c_int64)) if ri is not None else None, int(filter), int(extend), C.byref(t)))
    out = {k: C.string_at(getattr(t, k), getattr(t, k + "_bytes")) if getattr(t, k + "_bytes") else b"" for k in ("fasta", "info", "vcf", "ext")}
    lib.mtg_text_free(C.byref(t))
    return out


def vcf_header(sample, prefix):
    """mtg_format_vcf_header: the header the tool writes at the top of <prefix>.insertions.vcf"""
    lib = load_library()
    t = CText()
    _check(lib.mtg_format_vcf_header(sample.encode(), prefix.encode(), C.byref(t)))
    out = C.string_at(t.vcf, t.vcf_bytes)
    lib.mtg_text_free(C.byref(t))
    return out


def tuning():
    """every switch of the library (mtg_tuning_describe / mtg_tuning_get): [{name, default, kind, what, value}]; entry X starts from MTG_X in the environment"""
    lib = load_library()
    out = []
    for i in range(lib.mtg_tuning_count()):
        name, dflt, kind, what = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        _check(lib.mtg_tuning_describe(i, C.byref(name), C.byref(dflt), C.byref(kind), C.byref(what)))
        buf = C.create_string_buffer(64)
        _check(lib.mtg_tuning_get(name.value, buf, 64))
        out.append({"name": name.value.decode(), "default": dflt.value.decode(), "kind": kind.value.decode(), "what": what.value.decode(), "value": buf.value.decode()})
    return out


def tuning_set(name, value):
    """mtg_tuning_set: holds for the calls that begin after it; None or "" = not set"""
    _check(load_library().mtg_tuning_set(name.encode(), None if value is None else str(value).encode()))


def last_batch_stats():
    s = BatchStats()
    _check(load_library().mtg_last_batch_stats(C.byref(s)))
    return {f[0]: getattr(s, f[0]) for f in BatchStats._fields_}


def fill_main(argv):
    """`MindTheGap fill <argv>` (Filler::run, src/main.cpp:105-120); returns the exit code."""
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    return load_library().mtg_fill_main(len(argv), arr)


def nw_matches(pairs):
    """needleman_wunsch match counts (src/Utils.cpp:87-189) of [(a, b), ...] on the device; identity = matches / max(len(a), len(b))"""
    n = len(pairs)
    A = (C.c_char_p * n)(*[p[0].encode() for p in pairs])
    B = (C.c_char_p * n)(*[p[1].encode() for p in pairs])
    out = np.zeros(n, dtype=np.uint32)
    _check(load_library().mtg_nw_matches(A, B, n, out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def random_line_ceiling(table_bytes, n_chains, chain_len, line_bytes=64):
    ms, gbps = C.c_double(), C.c_double()
    _check(load_library().mtg_bench_random_lines(table_bytes, n_chains, chain_len, line_bytes, C.byref(ms), C.byref(gbps)))
    return ms.value, gbps.value


class Filler:
    """The reference's tool object (src/Filler.hpp) as the tests and the Python callers see it: `run(argv)` takes the reference's own option
    strings (src/Filler.cpp:76-113) and writes the same files; `gap_fill_from_source` is Filler::gapFillFromSource (src/Filler.hpp:188-189)
    for a batch of gaps."""
    STR_URI_BKPT, STR_URI_CONTIG, STR_URI_GRAPH, STR_URI_INPUT, STR_URI_OUTPUT = "-bkpt", "-contig", "-graph", "-in", "-out"
    STR_MAX_DEPTH, STR_MAX_NODES, STR_CONTIG_OVERLAP, STR_FILTER, STR_FWD_ONLY, STR_EXTEND = "-max-length", "-max-nodes", "-overlap", "-filter", "-fwd-only", "-extend"
    nb_mis_allowed = 2  # src/Filler.cpp:56

    def run(self, argv):
        """`MindTheGap fill <argv>`; returns the exit code (0 / 1)"""
        return fill_main(list(argv))

    def gap_fill_from_source(self, index, gaps, max_nodes=100, max_depth=10000):
        """gaps: iterable of Gap; returns the per-gap result dicts of Index.fill_batch"""
        return index.fill_batch(list(gaps), FillParams(max_nodes=max_nodes, max_depth=max_depth, nb_mis_allowed=self.nb_mis_allowed))
