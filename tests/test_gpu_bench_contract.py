"""GPU: the one JSON line of ``python bench.py`` at N = 1 -- every key of the driver's contract, the ``roofline`` and
``cpu_baseline`` objects, and the arithmetic between them -- on a shrunken workload (the driver runs the full one)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_line_carries_the_contract(tmp_path):
    env = dict(os.environ, VOLTRIX_TUNE_SPACE="none")
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                          "--scale", "0.05", "--tune", "none"], capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # ONE line
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "spmm_gflops" and line["unit"] == "GFLOP/s" and line["higher_is_better"] is True
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2 and line["vs_baseline"] is None
    assert line["dtype"] == "f16" and line["data"] == "synthetic" and "workload" in line["config"]
    cfg, roof, cpu = line["config"], line["roofline"], line["cpu_baseline"]
    # value = 2 nnz F / time
    assert abs(line["value"] - 2 * cfg["nnz"] * cfg["feat"] / (line["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * line["value"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # achieved = algorithmic bytes / measured kernel time (HIP events on the launch stream)
    assert abs(roof["achieved"] - roof["algorithmic_bytes"] / (roof["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    want_bytes = 4 * (cfg["nnz"] + cfg["num_nodes"] + 1) + cfg["num_nodes"] * cfg["feat"] * (2 + 4)
    assert roof["algorithmic_bytes"] == want_bytes
    assert "traffic" in roof                                     # null off the full-size configurations with PMC passes
    assert "gather_model" in roof                                 # a model beside the roofline; null without counters
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "GFLOP/s"
    assert isinstance(cpu["sample"], str) and cfg["rowsum_check_max_rel_err"] < 1e-4
    assert cfg["tile"]["launches_per_step"] == 1 and cfg["first_call_ms"] > 0


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` outside torch.distributed.run spawns its two ranks as child processes (before touching the
    GPU) and relays rank 0's ONE line; the timed step is the dependent one (gather, then the product that consumes it), the
    overlapped figure rides beside it.  Two gloo ranks on the one GPU of the box."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VOLTRIX_TUNE_SPACE"] = "none"
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                          "--scale", "0.01", "--workload", "reddit_like", "--steps", "3", "--warmup", "1", "--tune", "none",
                          "--gather", "collective", "--config5-scale", "0.002"], capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert run.stdout.strip() == lines[0], "only the JSON line on stdout (round 6: gloo / RCCL banners go to stderr)"
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    for key in ("allgather_ms", "local_spmm_ms", "predicted_ms", "step_independent_ms", "timed_step"):
        assert key in cfg, key
    assert cfg["timed_step"].startswith("dependent") and cfg["rowsum_check_max_rel_err"] < 1e-4
    # round 6: BASELINE configs[4] (papers-like sharded over the ranks) measured beside the headline steps of every N > 1 run
    c5 = cfg["config5_papers_like"]
    assert "error" not in c5, c5
    assert c5["workload"].startswith("papers_like") and c5["degree_check_max_rel_err_rank0"] == 0.0
    assert c5["step_ms"] >= 0.5 * max(c5["allgather_ms"], c5["local_spmm_ms"]) and c5["gflops"] > 0
    assert cfg["allgather_ms"] > 0 and cfg["local_spmm_ms"] > 0 and cfg["step_independent_ms"] > 0
    # the dependent step cannot be shorter than its two halves one after the other allow
    assert line["ms_per_step"] >= 0.5 * max(cfg["allgather_ms"], cfg["local_spmm_ms"])
