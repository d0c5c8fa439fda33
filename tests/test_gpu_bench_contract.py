"""The bench line's contract (driver's parser): one JSON line on stdout with the keys the round instructions name, on a small
workload so that the test takes seconds.  Runs bench.py as a child process (one GPU process besides pytest)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra, gpus=1, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2", "--repeats", "2", *extra]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_small_mesh():
    d = _run("--workload", "dir100k", "--nodes", "3000", "--cpu-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["edges_nonself"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert len(d["ms_per_step_all"]) == 2 and d["ms_per_step_spread"][0] <= d["ms_per_step"] <= d["ms_per_step_spread"][1]


def test_bench_batched_workload_line():
    d = _run("--workload", "batch50k", "--nodes", "2000", "--no-cpu-baseline")
    assert d["config"]["meshes_per_gpu"] == 8 and "batched" in d["config"]["parallelism"]
    assert d["iters_per_sec"] > 0 and "roofline" in d and "cpu_baseline" not in d


def test_bench_kernel_records_come_from_the_library():
    """`roofline` / `kernels` are built from the library's launch records (durations AND algorithmic bytes stated at the launch
    sites): every sweep row carries bytes, the iteration's total equals the sum of the rows, and the dominant kernel is one of them."""
    d = _run("--workload", "dir100k", "--nodes", "40000", "--no-cpu-baseline")
    rows = {r["kernel"]: r for r in d["kernels"]}
    assert "k_f_tile_fused" in rows and rows["k_f_tile_fused"]["launches"] == 6
    M4 = d["config"]["nodes"] * 10 * 4
    bf = 89 * d["config"]["nodes"] + 20 * d["config"]["edges_nonself"]
    assert abs(rows["k_f_tile_fused"]["alg_bytes_per_launch"] - (bf + 2 * M4)) < 1
    sweep_bytes = sum(r["alg_bytes_per_launch"] * r["work_launches"] for r in d["kernels"] if "alg_bytes_per_launch" in r)
    assert abs(sweep_bytes - d["roofline_iter"]["alg_bytes"]) < 1e-6 * sweep_bytes
    assert d["roofline"]["kernel"] in rows and d["roofline"]["alg_bytes_per_launch"] > 0
    # the Newton-Krylov leg: the stored linearisation is what it applies, and its 30 Arnoldi steps do reduce the linear residual
    nk, rj = d["newton_krylov"], d["roofline_jvp"]
    assert rj["kernel"] == "k_jvp_lin" and rj["avg_launch_us"] < d["roofline_jvp_direct"]["avg_launch_us"]
    assert nk["arnoldi_steps"] == 30 and 0.0 < nk["linear_residual_after_m"] < nk["linear_residual_first"] <= 1.0 + 1e-6
    assert d["ms_per_step_first"] > 0 and "warmup_extra" in d and d["ranks"]["world_size"] == 1


def test_bench_two_ranks_self_launched_on_one_card():
    """``bench.py --gpus 2`` started bare launches its two ranks itself (here both on the box's one card, rendezvous over gloo --
    RCCL refuses two ranks on one device): n_gpus = 2 in the line, twice the meshes, value = sum over ranks."""
    env = dict(os.environ, PSIGNN_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    d = _run("--workload", "batch50k", "--nodes", "3000", "--no-cpu-baseline", gpus=2, env=env)
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
    assert d["config"]["meshes_per_gpu"] == 8 and "x16" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 8 * d["config"]["edges_nonself"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
