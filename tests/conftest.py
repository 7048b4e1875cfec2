import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the in-tree libraries once if they are missing (hipcc cross-compiles without a GPU)."""
    pkg = os.path.join(ROOT, "relativitypathtracer_amd")
    need = [os.path.join(pkg, "librpt_scene.so"), os.path.join(pkg, "librpt_hip.so"),
            os.path.join(ROOT, "oracle", "librpt_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py")], check=True, cwd=ROOT)
    yield


# The benchmark configurations of BASELINE.json / BASELINE.md §3: scene, camera velocity, camera time.
CONFIGS = {
    "cube": dict(scene="cube", v=(0.0, 0.0, 0.0), t=0.0),
    "arch": dict(scene="arch", v=(0.0, 0.0, 0.95), t=5.25),
    "arch_t0": dict(scene="arch", v=(0.0, 0.0, 0.95), t=0.0),
    "bunny": dict(scene="bunny", v=(0.0, 0.0, 0.0), t=0.0),
    "shadows": dict(scene="shadows", v=(0.0, 0.0, 0.0), t=16.0),
    "cubes": dict(scene="cubes", v=(0.3, 0.0, 0.1), t=3.0),
    "rulers": dict(scene="rulers", v=(0.0, 0.0, 0.0), t=2.5),
    "ladder": dict(scene="ladder_paradox", v=(0.0, 0.0, 0.0), t=1.0),
    "soccer": dict(scene="soccer", v=(0.0, 0.0, 0.0), t=2.0),
}


def load_config(name):
    from relativitypathtracer_amd import Scene
    c = CONFIGS[name]
    s = Scene.from_file(c["scene"])
    s.set_camera(c["v"], c["t"])
    s.update_objects()
    return s
