"""GPU: the row-sharded operator on the HIP path -- world size 1 in-process, and world size 2 as two fresh child processes
that share cuda:0 (gloo all-gather): rectangular shards, column remap into the padded gather buffer, the two-level
side-car on a shard.  The N > 1 plumbing is also covered by tests/test_dist_gloo.py on CPU; the driver runs bench.py
--gpus N on a real 8-GPU node for the multi-GPU numbers (none was measured in rounds 1-2)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth_graphs
from oracle import torch_ref

pytestmark = pytest.mark.gpu

from conftest import REPO  # noqa: E402


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_row_sharded_world1_hip_path(cuda_device, monkeypatch):
    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.dist import RowShardedSpMM

    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.01)
    n = indptr.numel() - 1
    feat = torch.randn(n, 64).half()
    op = RowShardedSpMM(indptr, indices, n, hash_tag="dist_test")
    out = op(feat.cuda())
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert out.is_cuda and out.shape == (n, 64)
    assert float((out.cpu() - ref).norm() / ref.norm()) < 1e-6


def test_row_sharded_operator_with_two_level_format(cuda_device, monkeypatch):
    """VOLTRIX_HYBRID=1 gives the sharded operator's local handles the two-level side-car (column ids index the
    gathered buffer: num_cols = world * rows_padded)."""
    import numpy as np

    import synth_graphs
    from oracle import torch_ref
    from voltrix import dist as vdist
    from voltrix import two_level_of as voltrix_two_level_of

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    monkeypatch.setenv("VOLTRIX_HYBRID", "1")
    monkeypatch.setenv("VOLTRIX_HYBRID_MIN_SHARE", "0")
    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.006)
    n = indptr.numel() - 1
    op = vdist.RowShardedSpMM(indptr, indices, n, device=cuda_device, hash_tag="dist_two_level")
    hint = voltrix_two_level_of(op.handle[1])
    assert hint is not None and hint.plan.num_ksteps > 0
    feat = torch.randn(n, 64).half()
    out = op(feat.cuda())
    ref = torch_ref.spmm(indptr, indices, feat.float(), n)
    assert float((out.cpu() - ref).norm() / ref.norm()) < 1e-5


def test_slab_pipeline_first_call_is_right(cuda_device, monkeypatch):
    """with slabs > 1 the gather buffers used to be allocated (and zero-filled on the main stream) AFTER the fork
    to the communication stream, unordered against the first gather into them -- so the FIRST product of a fresh operator (or
    after a width / dtype change) could see wiped rows.  One-rank group with the exchange forced on (the all-gather and both
    streams really run): the first call, a second width, and the first width again, each against the oracle."""
    import torch.distributed as dist

    monkeypatch.setenv("VOLTRIX_TUNE_SPACE", "none")
    from voltrix.dist import RowShardedSpMM

    indptr, indices, _ = synth_graphs.generate("reddit_like", scale=0.02)
    n = indptr.numel() - 1
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        op = RowShardedSpMM(indptr.cuda(), indices.cuda(), n, hash_tag="dist_slab_first", slabs=3, exchange_at_world_1=True)
        for width in (96, 40, 96):
            feat = torch.randn(n, width).half()
            ref = torch_ref.spmm(indptr, indices, feat.float(), n)
            for _ in range(2):      # the first call of this width, then a call on the reused buffers
                out = op(feat.cuda())
                torch.cuda.synchronize()
                assert float((out.cpu() - ref).norm() / ref.norm()) < 1e-5, width
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["window", "two-level"])
def test_row_sharded_world2_two_processes_hip_path(cuda_device, tmp_path, mode):
    """Two ranks = two child processes on cuda:0 (3 GPU processes with this one): RowShardedSpMM end to end on the HIP
    extension with world-size-2 semantics, each rank checked against the oracle on its own rows."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    procs, outs = [], []
    for rank in range(2):
        out = str(tmp_path / f"rank{rank}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, worker, str(rank), "2", port, out] +
                                      (["two-level"] if mode == "two-level" else []),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append(p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(log[-2000:] for log in logs)
    results = [json.load(open(o)) for o in outs]
    assert all(r["ok"] for r in results), results
    assert results[0]["rows"][1] == results[1]["rows"][0] and results[0]["rows"][0] == 0
    assert all(r["two_level"] == (mode == "two-level") for r in results)
    if mode == "two-level":
        assert all(r["shared_edges"] > 0 for r in results)


@pytest.mark.parametrize("extra", [[], ["--gather", "auto"], ["--gather", "p2p"], ["--slabs", "2"]])
def test_bench_n_gt_1_branch_with_a_one_rank_rccl_group(tmp_path, extra):
    """bench.py's N > 1 code path -- init_process_group("nccl", device_id=...), voltrix.dist.RowShardedSpMM.from_shard on a
    device-resident shard, the in-place all_gather_into_tensor (or the batched point-to-point form, or the feature-slab
    pipeline), the barriers, the all-reduced timings and the allgather_ms leg -- in a fresh child process with a ONE-rank RCCL
    group: the calls the 8-GPU scaling run will make, rehearsed on the one GPU there is."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", VOLTRIX_TUNE_SPACE="none")
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-dist", "--workload",
                          "papers_like", "--scale", "0.002", "--steps", "3", "--warmup", "1", "--tune", "none",
                          "--no-cpu-baseline", *extra], capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0
    cfg = line["config"]
    assert cfg["allgather_ms"] > 0 and cfg["local_spmm_ms"] > 0 and cfg["rowsum_check_max_rel_err"] < 1e-4
    assert "exchange:" in cfg["parallelism"] and cfg["predicted_ms"]["step_direct_ms"] > 0
    assert cfg["first_call_ms"] > 0 and cfg["handle_bytes"]["reference_handle"] > 0
    if not extra:   # the default since round 6: the RCCL collective, no measured choice
        assert "exchange_choice" not in cfg and "exchange: collective" in cfg["parallelism"]
    if extra == ["--gather", "auto"]:   # both schedules were timed over RCCL and one of them was kept
        choice = cfg["exchange_choice"]
        assert set(choice["candidates_ms"]) == {"collective", "p2p"} and choice["picked"] in ("collective", "p2p")
        assert all(v > 0 for v in choice["candidates_ms"].values())


@pytest.mark.parametrize("extra", [[], ["--gather", "auto"], ["--gather", "p2p"], ["--slabs", "2"], ["--gather", "rows"],
                                   ["--gather", "auto", "--workload", "papers_like", "--scale", "0.002", "--rows-below", "1.01"]])
def test_bench_two_ranks_share_one_device_over_gloo(extra):
    """The N = 2 data path of bench.py end to end -- two ranks generate their own shards, ``RowShardedSpMM.from_shard``, the
    exchange (collective / point-to-point / slab pipeline), the product, and the row-sum check of every rank's result against
    torch ops on the gathered B (padded-shard layout: the check remaps the shard's global column ids like the operator does)
    -- as two processes on cuda:0 with gloo, the only two-rank form a one-GPU box allows."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VOLTRIX_TUNE_SPACE="none")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2",
                          "--one-device", "--backend", "gloo", "--workload", "reddit_like", "--scale", "0.1", "--steps", "2",
                          "--warmup", "1", "--tune", "none", "--config5-scale", "0.001", *extra], capture_output=True, text=True,
                         env=env, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rowsum_check_max_rel_err"] < 1e-4 and line["config"]["allgather_ms"] > 0
    if "--workload" not in extra:            # headline workload at N > 1: BASELINE configs[4] is measured beside it (round 6)
        assert "error" not in line["config"]["config5_papers_like"], line["config"]["config5_papers_like"]
    if "auto" in extra:   # --gather auto: measured choice, the same on both ranks (rank 0 prints it)
        choice = line["config"]["exchange_choice"]
        assert choice["picked"] in ("collective", "p2p", "rows") and set(choice["candidates_ms"]) == {"collective", "p2p"}
        if "--workload" in extra:            # the referenced-rows operator is built beside the all-gather one and both are
            assert 0 < choice["referenced_fraction_of_remote_rows"] <= 1.0   # (threshold raised: the candidate is always built)
            assert set(choice["whole_step_ms"]) == {"allgather", "rows"}     # timed, MAX over ranks
