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


def test_tuning_table_lists_reads_and_sets_every_switch():
    """every switch of the library is an entry of one table (mtg_tuning.h) the C-ABI lists, reads and sets; the product asks the table, not
    the environment: no getenv("MTG_...") is left in mindthegap_amd/ except the table's own two (MTG_<NAME>, MTG_TUNING)"""
    import subprocess
    import sys
    import mindthegap_amd
    from mindthegap_amd import lib as L
    L._lib = None
    t = mindthegap_amd.tuning()
    names = [e["name"] for e in t]
    assert len(names) == len(set(names)) >= 30
    assert all(e["kind"] in ("cap", "ab", "test", "diag") and len(e["what"]) > 10 for e in t)
    by = {e["name"]: e for e in t}
    assert by["COPY_SLOTS"]["default"] == "3" and by["NO_LEAN"]["default"] == ""
    mindthegap_amd.tuning_set("COPY_SLOTS", 5)
    mindthegap_amd.tuning_set("MTG_NO_LEAN", 1)  # the prefix is accepted
    by = {e["name"]: e["value"] for e in mindthegap_amd.tuning()}
    assert by["COPY_SLOTS"] == "5" and by["NO_LEAN"] == "1"
    mindthegap_amd.tuning_set("COPY_SLOTS", "3")
    mindthegap_amd.tuning_set("NO_LEAN", None)
    assert {e["name"]: e["value"] for e in mindthegap_amd.tuning()}["NO_LEAN"] == ""
    with pytest.raises(mindthegap_amd.MtgError):
        mindthegap_amd.tuning_set("NO_SUCH_SWITCH", 1)
    with pytest.raises(mindthegap_amd.MtgError):
        mindthegap_amd.tuning_set("COPY_SLOTS", "9" * 100)
    # the environment: MTG_<NAME> and MTG_TUNING, read when the table is first asked (a fresh process)
    code = "import mindthegap_amd as m; print({e['name']: e['value'] for e in m.tuning()})"
    env = dict(os.environ, MTG_FINISH_G="8", MTG_TUNING="ROUNDS=2,HOST_PATHS", PYTHONPATH=ROOT)
    out = eval(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert out["FINISH_G"] == "8" and out["ROUNDS"] == "2" and out["HOST_PATHS"] == "1" and out["COPY_SLOTS"] == "3"
    stray = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mindthegap_amd")):
        for f in files:
            if f.endswith((".cpp", ".h", ".hip")):
                for m in re.finditer(r'getenv\("(MTG_[A-Z0-9_]+)"\)', open(os.path.join(dirpath, f), errors="replace").read()):
                    if m.group(1) not in ("MTG_TUNING", "MTG_EMU_COOP_STATS", "MTG_EMU_PARK_STATS"):  # the last two: TEST-ONLY emulation build (#ifdef MTG_EMU)
                        stray.append((f, m.group(1)))
    assert not stray, stray
