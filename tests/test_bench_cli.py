"""bench.py's command line: `python bench.py --gpus N` works BARE (it starts its own ranks through torch.distributed.run
before touching a device and relays rank 0's line), the N > 1 plumbing — rendezvous on 127.0.0.1, barriers, max over
ranks, one JSON line — is exercised without a device by --dry-run, and without a GPU the real run fails loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config"}


def run_bench(*args, env=None, timeout=600):
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT,
                          env={**clean, **(env or {})}, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_bare_gpus_n_launches_its_own_ranks(n):
    p = run_bench("--gpus", str(n), "--steps", "4", "--warmup", "1", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 alone prints, once
    out = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(out)
    assert out["n_gpus"] == n and out["steps"] == 4 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["comm"]["ranks_in_group"] == n and out["comm"]["world_size"] == n
    assert "7680x4320" in out["comm"]["config5"]["workload"]          # the N > 1 line carries config 5 (timed in a real run)
    assert out["config"]["workload"].startswith("Scenes/bunny.txt 3840x2160") and "model" not in out["config"]


def test_dry_run_accepts_config_5():
    p = run_bench("--gpus", "2", "--dry-run", "--width", "7680", "--height", "4320")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert "7680x4320" in out["config"]["workload"]


def test_real_run_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for n in ("1", "2"):
        p = run_bench("--gpus", n, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
        assert p.returncode != 0
        assert "needs a GPU" in (p.stderr + p.stdout)
        assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]      # no line, no silent CPU fallback


@pytest.mark.gpu
def test_bare_single_gpu_line():
    """The driver's own N = 1 command form on the GPU box: one line, the contract's keys, roofline + regimes."""
    p = run_bench("--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--workload", "shadows", "--width", "1280", "--height", "720", "--check")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert CONTRACT_KEYS <= set(out) and out["n_gpus"] == 1
    assert out["check"] == "framebuffer rows identical to the oracle"
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and 0 < r["frac_blocking"] < 1
    # roofline.frac is the dominant kernel's own fraction (a launch that has the device to itself); the device-level figure with
    # four launches overlapping is reported under a name that says so
    assert r["frac"] == r["frac_kernel_alone"] and r["launches_overlapped"] == 1.0 and "ballot_first" in r["kernel"]
    assert 0 < r["frac_device_4_in_flight"] < 1 and r["device_in_flight"]["launches_overlapped"] >= 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 0.02 * r["achieved"]
    assert out["ms_per_frame_blocking"] > 0 and out["frame_ms_blocking"]["median"] > 0 and out["launch_ms_in_flight"]["frames"] == 10
    assert "traffic" in r       # null unless a PMC summary of this very build is committed


# ---- figures quoted from committed counter summaries --------------------------------------------------------------

def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counter_figures_are_quoted_only_for_the_build_they_were_taken_on(monkeypatch):
    """roofline.traffic / roofline.valu come from profiles/*_pmc_summary.json, per kernel, and only when the summary's recorded
    library hash is the running library's: a number from another build is stale and the line then says null."""
    import json
    bench = _load_bench()
    summary = json.load(open(os.path.join(ROOT, "profiles", "r04_bunny_3840x2160_pmc_summary.json")))
    recorded = summary["build"]["librpt_hip_sha256"]
    k43, k41 = "rpt_render_kernel_ballot_first_w5", "rpt_render_kernel_ballot_w5"

    monkeypatch.setattr(bench, "library_sha256", lambda: recorded)
    t43, src = bench.measured_traffic("bunny", 3840, 2160, k43)
    t41, _ = bench.measured_traffic("bunny", 3840, 2160, k41)
    assert src == os.path.join("profiles", "r04_bunny_3840x2160_pmc_summary.json")
    d43, d41 = summary["kernels"][k43]["derived"], summary["kernels"][k41]["derived"]
    assert t43 == int(sum(v for k, v in d43.items() if k.startswith("hbm_"))) and t41 == int(sum(v for k, v in d41.items() if k.startswith("hbm_")))
    assert t43 != t41                                              # one block per kernel, not a blend
    assert 3840 * 2160 * 16 <= t41 < 1.5 * 3840 * 2160 * 16         # the frame is written once and little else moves

    v = bench.measured_valu("bunny", 3840, 2160, k41, 0.1e-3)
    n = summary["kernels"][k41]["SQ_INSTS_VALU"]["mean"]
    assert v["wave_instructions_per_launch"] == int(n) and v["kernel"] == k41
    assert v["peak_wave_instructions_per_s"] == 256 * 4 * 2.4e9 / 2          # a wave64 instruction holds its SIMD-32 for two cycles
    assert abs(v["frac"] - n / (v["peak_wave_instructions_per_s"] * 0.1e-3)) < 1e-4
    assert 1.0 <= v["lanes_active_of_64"] <= 65.0
    assert bench.measured_valu("bunny", 3840, 2160, k41, None) is None
    assert bench.measured_traffic("bunny", 3840, 2160, "rpt_render_kernel_nonexistent") == (None, None)

    monkeypatch.setattr(bench, "library_sha256", lambda: "0" * 64)          # another build: nothing is quoted
    assert bench.measured_traffic("bunny", 3840, 2160, k43) == (None, None)
    assert bench.measured_valu("bunny", 3840, 2160, k43, 0.1e-3) is None
    monkeypatch.setattr(bench, "library_sha256", lambda: None)              # no library at all
    assert bench.measured_traffic("bunny", 3840, 2160, k43) == (None, None)
