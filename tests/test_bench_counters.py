"""CPU: bench.py replays the PMC-derived counters of profiles/traffic.json only for the exact configuration AND only while the
entry's ``sources_hash`` is the hash of the kernel sources in the tree (a kernel edit without new PMC passes must not ship
stale counters under a fresh step time)."""
import json
import os

import bench
from voltrix.jit.compiler import get_kernel_sources_version


def test_counters_need_matching_kernel_sources(tmp_path, monkeypatch):
    os.makedirs(tmp_path / "profiles")
    now = get_kernel_sources_version()
    runs = {"fresh": {"traffic_bytes": 1, "source": "a", "sources_hash": now},
            "stale": {"traffic_bytes": 2, "source": "b", "sources_hash": "0123456789ab"},
            "unhashed": {"traffic_bytes": 3, "source": "c"}}
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps({"runs": runs}))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    entry, why = bench.measured_counters("fresh", now)
    assert entry["traffic_bytes"] == 1 and why is None
    for key in ("stale", "unhashed"):
        entry, why = bench.measured_counters(key, now)
        assert entry is None and why.startswith("stale:") and now in why
    entry, why = bench.measured_counters("absent", now)
    assert entry is None and why.startswith("none for this exact configuration")


def test_kernel_sources_hash_follows_the_sources(tmp_path, monkeypatch):
    """Twelve hex digits over every header under include/voltrix and every source under csrc; stable across calls."""
    h = get_kernel_sources_version()
    assert len(h) == 12 and int(h, 16) >= 0 and h == get_kernel_sources_version()


def test_multi_gpu_defaults_are_one_workload_and_the_collective(monkeypatch):
    """Round 6: the driver derives its scaling curve from the per-N values of `bench.py --gpus N --steps K --warmup W`, so every N must
    time the SAME workload (the headline graph; papers-like is measured beside it at N > 1), and the default exchange is the RCCL
    collective -- the measured choice with its point-to-point candidate is opt-in."""
    import sys

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    args = bench.parse_args()
    assert args.workload is None and args.gather == "collective" and not args.no_config5 and args.config5_scale == 1.0
    src = open(bench.__file__).read()
    assert 'workload = args.workload or "reddit_like"' in src          # N = 1 and N > 1 alike
    assert 'extras["config5_papers_like"] = config5_leg(' in src
