"""The C++ example host (examples/rpt_render_main.cpp) — the reference's main()/render() sequence written
against the C-ABI only — compiles with g++, and on a GPU renders shadows.txt identically to the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "relativitypathtracer_amd")
ASSETS = os.path.join(ROOT, "assets", "reference")


def build(tmp_path):
    exe = str(tmp_path / "rpt_render")
    cmd = ["g++", "-O2", "-std=c++17", f"-I{ROOT}/include", f"{ROOT}/examples/rpt_render_main.cpp", "-o", exe,
           f"-L{PKG}", "-lrpt_hip", "-lrpt_scene", f"-Wl,-rpath,{PKG}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_example_host_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, "64", "48", str(tmp_path / "o.ppm")], stdin=f, capture_output=True, text=True,
                           env={**os.environ, "RPT_ASSETS": ASSETS})
    assert p.returncode == 1 and "no usable gfx950 device" in p.stderr
    # scene errors are reported, not fatal crashes
    with open(os.path.join(ASSETS, "Scenes", "arch.txt")) as f:      # needs a JPEG decoder the C++ example does not have
        p = subprocess.run([exe, "64", "48", str(tmp_path / "o.ppm")], stdin=f, capture_output=True, text=True,
                           env={**os.environ, "RPT_ASSETS": ASSETS})
    assert p.returncode == 1 and "ReadTexture" in p.stderr


@pytest.mark.gpu
def test_example_host_matches_oracle(tmp_path):
    import oracle_ffi
    from relativitypathtracer_amd import Scene
    exe = build(tmp_path)
    out = tmp_path / "shadows.ppm"
    W, H = 320, 184
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, str(W), str(H), str(out), "0", "0", "0", "16"], stdin=f, capture_output=True, text=True,
                           env={**os.environ, "RPT_ASSETS": ASSETS})
    assert p.returncode == 0, p.stderr
    data = out.read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    assert data.startswith(header)
    img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    s = Scene.from_file("shadows")
    s.set_camera((0, 0, 0), 16.0)
    s.update_objects()
    opx, _, _ = oracle_ffi.render(s, W, H, want_rgb=False)
    want = opx["rgba"].reshape(H, W, 4)[::-1, :, :3]
    assert np.array_equal(img, want)


@pytest.mark.gpu
def test_example_host_frames_in_flight(tmp_path):
    """40 frames with the clock running, 3 in flight over rpt_share_scene: the PPM holds the last frame."""
    import oracle_ffi
    from relativitypathtracer_amd import Scene
    exe = build(tmp_path)
    out = tmp_path / "anim.ppm"
    W, H, frames = 320, 184, 40
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, str(W), str(H), str(out), "0", "0", "0", "16", str(frames), "3"], stdin=f, capture_output=True,
                           text=True, env={**os.environ, "RPT_ASSETS": ASSETS})
    assert p.returncode == 0, p.stderr
    assert f"{frames} frames, 3 in flight" in p.stderr
    data = out.read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    s = Scene.from_file("shadows")
    s.set_camera((0, 0, 0), 16.0)
    s.set_paused(False)
    for _ in range(frames):
        s.advance_time(16)
    s.update_objects()
    opx, _, _ = oracle_ffi.render(s, W, H, want_rgb=False)
    assert np.array_equal(img, opx["rgba"].reshape(H, W, 4)[::-1, :, :3])


@pytest.mark.gpu
def test_frame_ring_acquire_hands_out_the_frame_of_n_submits_ago(tmp_path):
    """rpt::FrameRing::acquire() returns the OLDEST frame in flight, complete and not yet overwritten: the example host
    copies frame 17 of 40 back right after acquire() (while two newer frames are in flight) and the pixels are the
    oracle's frame 17 — not frame 20, which the same slot receives in the following enqueue()."""
    import oracle_ffi
    from relativitypathtracer_amd import Scene
    exe = build(tmp_path)
    out, dump = tmp_path / "anim.ppm", tmp_path / "frame17.ppm"
    W, H, frames, k = 320, 184, 40, 17
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, str(W), str(H), str(out), "0", "0", "0", "16", str(frames), "3"], stdin=f, capture_output=True,
                           text=True, env={**os.environ, "RPT_ASSETS": ASSETS, "RPT_DUMP_FRAME": str(k), "RPT_DUMP_PATH": str(dump)})
    assert p.returncode == 0, p.stderr
    header = f"P6\n{W} {H}\n255\n".encode()
    img = np.frombuffer(dump.read_bytes()[len(header):], np.uint8).reshape(H, W, 3)
    frames_by_clock = {}
    for n in (k, k + 3):
        s = Scene.from_file("shadows")
        s.set_camera((0, 0, 0), 16.0)
        s.set_paused(False)
        for _ in range(n + 1):
            s.advance_time(16)
        s.update_objects()
        opx, _, _ = oracle_ffi.render(s, W, H, want_rgb=False)
        frames_by_clock[n] = opx["rgba"].reshape(H, W, 4)[::-1, :, :3]
    assert np.array_equal(img, frames_by_clock[k])
    assert not np.array_equal(frames_by_clock[k], frames_by_clock[k + 3])


def build_multi_gpu(tmp_path):
    exe = str(tmp_path / "rpt_multi_gpu")
    cmd = ["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT}/include", "-I/opt/rocm/include",
           f"{ROOT}/examples/rpt_multi_gpu_main.cpp", "-o", exe, f"-L{PKG}", "-lrpt_hip", "-lrpt_scene", "-L/opt/rocm/lib",
           "-lamdhip64", "-lrccl", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_multi_gpu_example_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_multi_gpu(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, "64", "48", str(tmp_path / "o.ppm"), "3"], stdin=f, capture_output=True, text=True,
                           env={**os.environ, "RPT_ASSETS": ASSETS})
    assert p.returncode == 1 and "no usable gfx950 device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("root_run", ["0", "4"])
def test_multi_gpu_example_matches_oracle(tmp_path, root_run):
    """The native C++ multi-GPU host (row tiles -> colour planes -> ONE ncclGather -> root reassembly, three frames
    in flight) on the GPUs this box has (one: the gather is then GPU 0 to itself): 20 animated frames, the PPM
    holds the last."""
    import oracle_ffi
    from relativitypathtracer_amd import Scene
    exe = build_multi_gpu(tmp_path)
    out = tmp_path / "multi.ppm"
    W, H, frames = 320, 184, 20
    with open(os.path.join(ASSETS, "Scenes", "shadows.txt")) as f:
        p = subprocess.run([exe, str(W), str(H), str(out), str(frames), "1", root_run], stdin=f, capture_output=True, text=True,
                           env={**os.environ, "RPT_ASSETS": ASSETS, "RPT_T0": "16"}, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert f"{frames} frames of {W}x{H}, 3 in flight" in p.stderr
    data = out.read_bytes()
    header = f"P6\n{W} {H}\n255\n".encode()
    img = np.frombuffer(data[len(header):], np.uint8).reshape(H, W, 3)
    s = Scene.from_file("shadows")
    s.set_camera((0, 0, 0), 16.0)
    s.set_paused(False)
    for _ in range(frames):
        s.advance_time(16)
    s.update_objects()
    opx, _, _ = oracle_ffi.render(s, W, H, want_rgb=False)
    assert np.array_equal(img, opx["rgba"].reshape(H, W, 4)[::-1, :, :3])


def test_frame_ring_header_is_plain_cxx11(tmp_path):
    """include/rpt_frames.hpp is header-only C++11 over the C-ABI: it must compile on its own, warnings as errors."""
    src = tmp_path / "ring.cpp"
    src.write_text('#include "rpt_frames.hpp"\nint main() { rpt::FrameRing *r = nullptr; (void)r; return 0; }\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Wpedantic", "-Werror", f"-I{ROOT}/include", "-fsyntax-only", str(src)],
                   check=True, capture_output=True)
