"""csrc/rpt_workers.hpp: the helper threads rpt_set_objects shares a batch of screen bounds with.  No GPU: a C++ stress
(tests/native/workers_stress.cpp) built with g++, plain and under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "workers_stress.cpp")


def _build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    p = subprocess.run(["g++", "-std=c++17", "-pthread", *extra, "-o", exe, SRC], capture_output=True, text=True, timeout=300)
    return exe, p


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_every_item_once_from_two_submitting_threads(tmp_path):
    exe, p = _build(tmp_path, "workers_stress", ["-O2"])
    assert p.returncode == 0, p.stderr[-2000:]
    for env_threads in (None, "0", "2"):        # default, no helpers (the caller does everything), two helpers
        env = dict(os.environ)
        if env_threads is not None:
            env["RPT_HOST_THREADS"] = env_threads
        r = subprocess.run([exe, "8000"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and " bad 0" in r.stdout, (env_threads, r.stdout, r.stderr[-2000:])
        if env_threads == "0":
            assert r.stdout.startswith("threads 0 ")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_a_forked_child_computes_its_batches_alone(tmp_path):
    """fork() in a process whose pool is busy: the child inherits the counters, not the threads; the pthread_atfork child
    handler empties the pool, every batch of the child completes on its calling thread (before, a child could
    deadlock on the pool's mutex or wait for holders that do not exist)."""
    exe, p = _build(tmp_path, "workers_stress", ["-O2"])
    assert p.returncode == 0, p.stderr[-2000:]
    r = subprocess.run([exe, "--fork", "200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " bad 0" in r.stdout, (r.stdout, r.stderr[-2000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_thread_sanitizer_finds_no_race(tmp_path):
    exe, p = _build(tmp_path, "workers_tsan", ["-O1", "-g", "-fsanitize=thread"])
    if p.returncode != 0:
        pytest.skip("g++ cannot link -fsanitize=thread here: " + p.stderr[-300:])
    r = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and " bad 0" in r.stdout, (r.stdout, r.stderr[-2000:])
