"""The C-ABI library loads on a machine without GPU and exports every symbol include/mtg_fill.h declares; without a device the
entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    import mindthegap_amd
    so = mindthegap_amd.build_library()
    lib = C.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "mtg_fill.h")).read()
    names = set(re.findall(r"\b(mtg_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), "missing export: " + n


def test_no_cpu_fallback_without_device():
    import mindthegap_amd
    from mindthegap_amd import lib as L
    L._lib = None  # make sure the real library is bound (emulator tests re-point the handle)
    mindthegap_amd.load_library()
    if mindthegap_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(mindthegap_amd.MtgError) as e:
        mindthegap_amd.Index.from_kmers(np.array([1, 2], dtype=np.uint64), np.array([1, 1], dtype=np.uint32), 31)
    assert e.value.code == 1  # MTG_ERR_NO_DEVICE
    assert mindthegap_amd.Filler().run(["-in", os.path.join(ROOT, "tests", "golden", "data", "contigs.fasta"), "-bkpt", "x", "-out", "/tmp/mtg_nodev"]) == 1


def test_product_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mindthegap_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower().replace("oracle/ is never", ""), os.path.join(dirpath, f)
