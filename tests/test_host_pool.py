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
