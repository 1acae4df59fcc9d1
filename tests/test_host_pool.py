"""the host worker pool of libmtgfill.so (mtg_internal.h) and the block patterns built on it, under ThreadSanitizer"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_patterns_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "pool_stress")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe, os.path.join(ROOT, "tests", "emu", "pool_stress.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_concurrent_batches_on_one_index_under_thread_sanitizer(tmp_path):
    """three threads x six batches on one (emulated) index: workspace hand-out, shared pool, result cache and the per-batch host state are
    free of data races, and every batch equals the single-threaded one"""
    exe = str(tmp_path / "concurrent_batches")
    csrc = os.path.join(ROOT, "mindthegap_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("concurrent_batches.cpp", "emu_backend.cpp")] + [os.path.join(csrc, f) for f in ("mtg_host.cpp", "mtg_cli.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe] + srcs + ["-lz"])
    env = dict(os.environ, MTG_POOL_THREADS="4", MTG_EMU_PART="256")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-500:], r.stderr[-2000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]


def test_vectorised_host_helpers_match_scalar_code(tmp_path):
    """decode_slice (pdep + vpshufb) for every alignment / length / direction and FillInput::set_common (pext packing, k-mer by bit
    reversal, validity tests) against per-character code, with and without the vector paths (MTG_NO_VEC)"""
    exe = str(tmp_path / "host_units")
    csrc = os.path.join(ROOT, "mindthegap_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("host_units.cpp", "emu_backend.cpp")] + [os.path.join(csrc, "mtg_cli.cpp")]
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-pthread", "-o", exe] + srcs + ["-lz"])
    for extra in ({}, {"MTG_NO_VEC": "1"}):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (extra, r.stdout[-300:], r.stderr[-1000:])


def test_contig_mode_dictionary_order_equals_the_literal_construction(tmp_path):
    """contig mode's per-seed target dictionary (src/Filler.cpp:522-533): the order mtg_dict_order.h derives from the keys' hash codes (a map of target
    numbers on a reused block, round 6) equals the iteration order of the reference's literally constructed std::unordered_map<std::string, ...>, for
    400 random dictionaries of up to 6000 keys with none / one / some / all entries left out"""
    exe = str(tmp_path / "dict_order")
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "emu", "dict_order.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-300:], r.stderr[-1000:])
